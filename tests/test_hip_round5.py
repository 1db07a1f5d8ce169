"""Round 5 on the GPU.

  * ABI v5: gather sources channel-last, EACH IN ITS OWN ALLOCATION -- the tensors actual torch convolutions write in the channels_last /
    channels_last_3d memory formats (the producers of network/mvs_models.py:624-646, new every step in train.py:136-163) -- read in place,
    recognised per source by stride; forward bit-identical to the repacked route, gradients returned with the inputs' strides;
  * round 4's advisor findings: a graphed step replays on NEW source values; FlatAdam leaves frozen tensors alone; the "versions" weight cache
    sees FlatAdam steps.
"""
import types

import pytest
import torch

from conftest import load_golden
from test_hip_configs import dev
from test_hip_round4 import _call, _mods, _net, _qfn

pytestmark = pytest.mark.gpu


def _same_strides(a, b):
    """Equal strides on every dimension that has more than one element (torch leaves the others arbitrary)."""
    return all(n == 1 or x == y for n, x, y in zip(a.shape, a.stride(), b.stride()))
DEV = "cuda:0"


def _conv_sources(scene, seed=0):
    """Sources as nn.Conv3d / nn.Conv2d running in channels_last(_3d) write them on this GPU (what bench.py's conv_produced_sources does)."""
    g = torch.Generator().manual_seed(seed)
    c3 = torch.nn.Conv3d(8, 8, 3, padding=1).to(DEV).to(memory_format=torch.channels_last_3d)
    c2 = torch.nn.Conv2d(8, 8, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
    with torch.no_grad():
        for m in (c3, c2):
            m.weight.copy_(dev(torch.randn(m.weight.shape, generator=g)) * 0.1)
            m.bias.copy_(dev(torch.randn(m.bias.shape, generator=g)) * 0.1)
        vols = [c3(v.contiguous(memory_format=torch.channels_last_3d)) for v in scene["vols"]]
        feat = c2(scene["img_feat"][:, 0].contiguous(memory_format=torch.channels_last)).unsqueeze(1)
        imgs = scene["imgs"][0].contiguous(memory_format=torch.channels_last).unsqueeze(0)
    return vols, imgs, feat


def _scene():
    from uc_nerf_amd.synthetic import make_scene, scene_to
    return scene_to(make_scene(seed=0), torch.device(DEV))


def test_conv_outputs_in_channels_last_formats_are_read_in_place():
    """What MIOpen really returns: a Conv3d in channels_last_3d writes [D,h,w,8] memory, a Conv2d in channels_last [V,H,W,8]; five separate
    allocations; every one recognised and read in place -- no repack buffer exists -- and the render equals the repacked route of the same
    values bit for bit on every kernel route (network/mvs_models.py:624-646, utils/utils.py:742-799,833-893)."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, random_pixels
    scene = _scene()
    vols, imgs, feat = _conv_sources(scene)
    for v in vols:
        assert v.is_contiguous(memory_format=torch.channels_last_3d) and ops.cl_volume_view(v) is not None, v.stride()
    assert ops.cl_img_feat_view(feat) is not None, feat.stride()
    assert ops.cl_imgs_view(imgs) is not None and ops.cl_imgs_view(imgs)[1] == 3, imgs.stride()
    assert len({t.untyped_storage().data_ptr() for t in vols + [imgs, feat]}) == 5
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    flat = flat_params_of(sd).to(DEV)
    major = dict(scene, vols=[v.contiguous() for v in vols], imgs=imgs.contiguous(), img_feat=feat.contiguous())
    for n in (512, 700, 2048):
        xs, ys = random_pixels(n, 256, 320, seed=n)
        noise = dev(torch.rand(n, 64, generator=torch.Generator().manual_seed(100)))
        for prec in ("bf16x3_fused", "bf16x3", "f32"):
            ra = CoarseFineRenderer(major, flat, 64, 128, precision=prec)
            rb = CoarseFineRenderer(dict(scene, vols=vols, imgs=imgs, img_feat=feat), flat, 64, 128, precision=prec)
            assert rb.src.zero_copy and rb.src.inplace == [True] * 5 and not any(ra.src.inplace)
            a = ra.render(dev(xs), dev(ys), perturb=1.0, noise=noise)
            b = rb.render(dev(xs), dev(ys), perturb=1.0, noise=noise)
            assert rb.src._cl is None                                   # nothing was repacked
            for k in ("rgb", "depth", "acc", "weights", "z_fine"):
                assert torch.equal(a[k], b[k]), (n, prec, k)


@pytest.mark.parametrize("which", [(True, False, True, False, True), (False, True, False, True, False), (True, True, True, True, False), (False, False, False, False, True)])
def test_sources_are_recognised_one_by_one(which):
    """Per source: the channel-last ones are read in place, the others repacked into a buffer that holds only them (ucnerf_gather_repack keeps the
    entries of ucnerf_render_params.cl that are set); same bits as the all-repacked route.  Padded [V,H,W,4] colours are read in place too."""
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, random_pixels
    scene = _scene()
    vols, imgs, feat = _conv_sources(scene, seed=1)
    mix_v = [v if w else v.contiguous() for v, w in zip(vols, which[:3])]
    mix_f = feat if which[3] else feat.contiguous()
    if which[4]:                                                                  # colours as padded 16-byte pixels
        buf = torch.zeros(imgs.shape[1], 256, 320, 4, device=DEV)
        buf[..., :3] = imgs[0].permute(0, 2, 3, 1)
        mix_i = buf[..., :3].permute(0, 3, 1, 2).unsqueeze(0)
    else:
        mix_i = imgs.contiguous()
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    flat = flat_params_of(sd).to(DEV)
    major = dict(scene, vols=[v.contiguous() for v in vols], imgs=imgs.contiguous(), img_feat=feat.contiguous())
    xs, ys = random_pixels(600, 256, 320, seed=2)
    for prec in ("bf16x3_fused", "f32"):
        ra = CoarseFineRenderer(major, flat, 64, 128, precision=prec)
        rb = CoarseFineRenderer(dict(scene, vols=mix_v, imgs=mix_i, img_feat=mix_f), flat, 64, 128, precision=prec)
        assert rb.src.inplace == list(which) and not rb.src.zero_copy
        assert not which[4] or rb.src.rgb_stride == 4
        a, b = ra.render(dev(xs), dev(ys)), rb.render(dev(xs), dev(ys))
        for k in ("rgb", "depth", "acc", "weights"):
            assert torch.equal(a[k], b[k]), (prec, k)
        # the repack buffer holds exactly the sources that are not in place
        n_vox = [v.numel() for v in vols]
        want = sum(n for n, w in zip(n_vox, which[:3]) if not w) + (0 if which[3] else feat.numel()) + (0 if which[4] else 4 * imgs.numel() // 3)
        assert rb.src._cl.numel() == want
        p = rb.pass_.p
        for k in range(3):
            assert (p.cl.vol[k] == mix_v[k].data_ptr()) == which[k]
        assert (p.cl.img_feat == mix_f.data_ptr()) == which[3] and (p.cl.imgs == mix_i.data_ptr()) == which[4]
    assert L.lib().ucnerf_abi_version() == 6


@pytest.mark.parametrize("which", [(True,) * 4, (True, False, True, False), (False, True, False, True)])
def test_training_through_conv_produced_sources(which, sd_v7):
    """rendering() under autograd on sources that convolutions wrote channel-last (train.py:136-163): forward bit-identical to the channel-major
    route; the volume / image-feature gradients arrive with the inputs' strides (accumulated channel-last by the gather backward: no scratch, no
    transposing add for them) and equal the channel-major route's up to float-atomic order -- also when only some sources are channel-last."""
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    gen = torch.Generator().manual_seed(7)
    r3, r1 = dev(torch.randn(g["z"].shape[0], 3, generator=gen)), dev(torch.randn(g["z"].shape[0], generator=gen))
    c3 = torch.nn.Conv3d(8, 8, 3, padding=1).to(DEV).to(memory_format=torch.channels_last_3d)
    c2 = torch.nn.Conv2d(8, 8, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
    with torch.no_grad():
        vols_cl = [c3(dev(g["vol%d" % k]).contiguous(memory_format=torch.channels_last_3d)) for k in (1, 2, 3)]
        feat_cl = c2(dev(g["img_feat"])[:, 0].contiguous(memory_format=torch.channels_last)).unsqueeze(1)
    seen = {}

    def run(cl):
        net = _net(mods, 7, sd_v7)
        vols = [(v if (cl and w) else v.contiguous()).detach().requires_grad_(True) for v, w in zip(vols_cl, which[:3])]
        img_feat = (feat_cl if (cl and which[3]) else feat_cl.contiguous()).detach().requires_grad_(True)
        if cl:
            img_feat.register_hook(lambda gr: seen.update(stride=gr.stride()))
        conf = dev(g["conf"]).requires_grad_(True)
        rgb, depth = _call(mods, g, net, qfn, vols=vols, img_feat=img_feat, conf=conf)
        ((rgb * r3).sum() + (depth * r1).sum()).backward()
        from uc_nerf_amd import dropin
        return rgb.detach(), depth.detach(), vols, img_feat, conf, net, dropin.session_of(net)

    rgb_a, d_a, vols_a, if_a, conf_a, net_a, _ = run(False)
    rgb_b, d_b, vols_b, if_b, conf_b, net_b, sess = run(True)
    assert sess.src.inplace[:4] == list(which)
    assert torch.equal(rgb_a, rgb_b) and torch.equal(d_a, d_b)
    for k, (v, w) in enumerate(zip(vols_b, vols_a)):
        assert v.grad.shape == v.shape and _same_strides(v.grad, v), (k, v.grad.stride(), v.stride())
        torch.testing.assert_close(v.grad, w.grad, atol=2e-5 * max(w.grad.abs().max().item(), 1e-6), rtol=1e-4)
    assert all(n == 1 or a == b for n, a, b in zip(if_b.shape, seen["stride"], if_b.stride())), (seen["stride"], if_b.stride())
    torch.testing.assert_close(if_b.grad, if_a.grad, atol=2e-5 * max(if_a.grad.abs().max().item(), 1e-6), rtol=1e-4)
    torch.testing.assert_close(conf_b.grad, conf_a.grad, atol=2e-5 * max(conf_a.grad.abs().max().item(), 1e-6), rtol=1e-4)
    for (name, p), q in zip(net_a.named_parameters(), net_b.parameters()):
        if p.grad is None:
            assert q.grad is None, name
        else:
            torch.testing.assert_close(q.grad, p.grad, atol=2e-5 * max(p.grad.abs().max().item(), 1e-6), rtol=1e-4, msg=lambda s_: name + ": " + s_)


def test_four_view_scene_through_channel_last_sources():
    """The Hamlyn-style configuration (view_num 4: three source views, configs[3]) on channel-last sources in place: the no-grad render equals the
    channel-major one bit for bit and the golden vectors to 1e-4; under autograd the source gradients agree with the channel-major route."""
    from test_oracle_golden import sd_v4_for_g16
    from uc_nerf_amd import dropin
    mods = _mods()
    g = load_golden("g16_rendering_v4")
    sd = sd_v4_for_g16(g)
    qfn = _qfn(mods)
    vols_cl = [dev(g["vol%d" % k]).contiguous(memory_format=torch.channels_last_3d) for k in (1, 2, 3)]
    feat = dev(g["img_feat"])
    feat_cl = feat[:, 0].contiguous(memory_format=torch.channels_last).unsqueeze(1)
    imgs = dev(g["imgs"])
    imgs_cl = imgs[0].contiguous(memory_format=torch.channels_last).unsqueeze(0)
    net = _net(mods, g["V"], sd)
    with torch.no_grad():
        rgb_a, d_a = _call(mods, g, net, qfn)
        rgb_b, d_b = _call(mods, g, net, qfn, vols=vols_cl, img_feat=feat_cl, imgs=imgs_cl)
    sess = dropin.session_of(net)
    assert sess.src.zero_copy and sess.src.inplace == [True] * 5
    assert torch.equal(rgb_a, rgb_b) and torch.equal(d_a, d_b)
    from test_hip_configs import close
    close(rgb_b, g["rgb_first" if "rgb_first" in g else "rgb"], 1e-4); close(d_b, g["depth_first" if "depth_first" in g else "depth"], 1e-4)
    gen = torch.Generator().manual_seed(3)
    r3, r1 = dev(torch.randn(g["z"].shape[0], 3, generator=gen)), dev(torch.randn(g["z"].shape[0], generator=gen))

    def run(cl):
        net = _net(mods, g["V"], sd)
        vols = [(v if cl else v.contiguous()).detach().requires_grad_(True) for v in vols_cl]
        f = (feat_cl if cl else feat).detach().requires_grad_(True)
        rgb, depth = _call(mods, g, net, qfn, vols=vols, img_feat=f, imgs=imgs_cl if cl else imgs)
        ((rgb * r3).sum() + (depth * r1).sum()).backward()
        return vols, f

    va, fa = run(False)
    vb, fb = run(True)
    for a, b in zip(va + [fa], vb + [fb]):
        assert _same_strides(b.grad, b)
        torch.testing.assert_close(b.grad, a.grad, atol=2e-5 * max(a.grad.abs().max().item(), 1e-6), rtol=1e-4)


def test_frozen_sources_cost_no_gradient_buffer(sd_v7):
    """need = all False for the heavy sources (a frozen MVS network, detached volumes): the backward allocates no source-gradient segment and hands the
    gather backward no channel-last gradient array (round 4's advisor finding on the zero-copy route)."""
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    net = _net(mods, 7, sd_v7)
    vols = [dev(g["vol%d" % k]).contiguous(memory_format=torch.channels_last_3d) for k in (1, 2, 3)]
    torch.cuda.reset_peak_memory_stats()
    rgb, depth = _call(mods, g, net, qfn, vols=vols)
    base = torch.cuda.memory_allocated()
    (rgb.sum() + depth.sum()).backward()
    torch.cuda.synchronize()
    assert all(v.grad is None for v in vols)
    pool = sum(p.grad.numel() for p in net.parameters() if p.grad is not None)
    assert pool > 0 and torch.cuda.max_memory_allocated() - base < 64 << 20        # (a source-gradient pool alone would be 72 MB on the bench scene; here the scene is tiny: the bound is on what the backward adds)


# ---------------------------------------------------------------------------------------------- round 4's advisor findings
def test_graphed_step_gathers_from_the_sources_of_the_replay(sd_v7):
    """train_step.GraphedStep: new volumes / image features COPIED INTO the static source tensors between replays are what the replay trains on --
    the capture records the source repack although the tensors' identity and version said "unchanged" while capturing.  Channel-last (in place)
    sources need no repack at all.  Compared with the eager step on the same values."""
    from uc_nerf_amd.flat import FlatAdam
    from uc_nerf_amd.train_step import GraphedStep
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    gen = torch.Generator().manual_seed(5)
    target = dev(torch.rand(g["z"].shape[0], 3, generator=gen))
    V = g["V"]
    args = types.SimpleNamespace(view_num=V, feat_dim=24 + 12 * (V - 1) + 1, img_downscale=1.0, use_color_volume=False, net_type="v2")
    w2cs, Ks = dev(g["w2cs"]), dev(g["K"]).repeat(V, 1, 1)
    ndc = {"stage1": dev(g["ndc1"]), "stage2": dev(g["ndc2"]), "stage3": dev(g["ndc3"]), "ndc": dev(g["ndc"])}
    pts, z, rays_d, imgs, conf = dev(g["pts"]), dev(g["z"]), dev(g["rays_d"]), dev(g["imgs"]), dev(g["conf"])
    new_vols = [dev(g["vol%d" % k]) * 0.5 + 0.1 for k in (1, 2, 3)]
    new_feat = dev(g["img_feat"]) * 0.7 - 0.05

    for channel_last in (False, True):
        fmt3 = torch.channels_last_3d if channel_last else torch.contiguous_format
        vols_e = [dev(g["vol%d" % k]).contiguous(memory_format=fmt3) for k in (1, 2, 3)]
        vols_g = [v.clone(memory_format=torch.preserve_format) for v in vols_e]
        feat_e, feat_g = dev(g["img_feat"]), dev(g["img_feat"])

        def make(net, opt, vols, feat):
            vf = {"stage%d" % (k + 1): {"volume_feature_no_ref": vols[k]} for k in range(3)}

            def step():
                opt.zero_grad(set_to_none=True)
                rgb, depth = mods.renderer.rendering(args, {"w2cs": w2cs, "intrinsics": Ks}, pts, ndc, z, rays_d, vf, imgs, network_fn=net, img_feat=feat,
                                                     network_query_fn=qfn, confidence=conf)
                loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
                loss.backward()
                opt.step()
                return loss
            return step

        net_e, net_g = _net(mods, 7, sd_v7), _net(mods, 7, sd_v7)
        step_e = make(net_e, FlatAdam(net_e, lr=5e-4), vols_e, feat_e)
        graphed = GraphedStep(make(net_g, FlatAdam(net_g, lr=5e-4, capturable=True), vols_g, feat_g), warmup=2)
        for _ in range(2):
            step_e()
        # a new batch of sources: written into the tensors both steps read
        with torch.no_grad():
            for dst_e, dst_g, src in zip(vols_e, vols_g, new_vols):
                dst_e.copy_(src)
                dst_g.copy_(src)
            feat_e.copy_(new_feat)
            feat_g.copy_(new_feat)
        for _ in range(3):
            le, lg = step_e(), graphed.replay()
            assert abs(lg.item() - le.item()) < 1e-4 * max(1.0, abs(le.item())), channel_last
            for (name, a), b in zip(net_e.named_parameters(), net_g.parameters()):
                torch.testing.assert_close(a, b, atol=2e-5, rtol=0, msg=lambda s_: name + ": " + s_)


def test_flat_adam_leaves_frozen_tensors_alone(sd_v7):
    """A partly frozen network (requires_grad = False on some tensors) through the REAL backward: the kernels write a weight gradient for every tensor
    they reach, but the frozen segments of the flat gradient are zeroed before anybody can step or reduce them -- FlatAdam makes the step
    torch.optim.Adam(grad_vars) makes (train.py:85-92 hands Adam whatever requires a gradient)."""
    from uc_nerf_amd.flat import FlatAdam, FlatStore
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    nets = [_net(mods, 7, sd_v7), _net(mods, 7, sd_v7)]
    frozen = ("pts_linears.0.", "pts_linears.3.", "views_linears.0.weight", "rgb_linear.bias")
    for net in nets:
        for name, p in net.named_parameters():
            if any(t in name for t in frozen):
                p.requires_grad_(False)
    before = {n: p.detach().clone() for n, p in nets[0].named_parameters()}
    opts = [torch.optim.Adam([p for p in nets[0].parameters() if p.requires_grad], lr=1e-3), FlatAdam(nets[1], lr=1e-3)]
    for _ in range(3):
        for net, opt in zip(nets, opts):
            opt.zero_grad(set_to_none=True)
            rgb, depth = _call(mods, g, net, qfn)
            (rgb.sum() + 0.3 * depth.sum()).backward()
            opt.step()
    assert FlatStore.of(nets[1]).flat_grad() is not None                       # (the flat route was taken)
    moved = 0
    for (name, a), b in zip(nets[0].named_parameters(), nets[1].parameters()):
        if any(t in name for t in frozen):
            assert torch.equal(b, before[name]) and torch.equal(a, before[name]), name
        else:
            torch.testing.assert_close(b, a, atol=1e-5, rtol=1e-4, msg=lambda s_: name + ": " + s_)      # (three Adam steps on gradients that differ by float-atomic order)
            moved += int(not torch.equal(a, before[name]))
    assert moved >= 20


def test_versions_weight_cache_sees_flat_adam_steps(sd_v7):
    """set_weight_cache("versions"): FlatAdam steps a Parameter that shares the flat buffer's storage and version counter, not the per-tensor ones;
    the cache key includes the buffer's own counter, so an evaluation call after a step renders with the new weights."""
    import uc_nerf_amd
    from uc_nerf_amd.flat import FlatAdam
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    net = _net(mods, 7, sd_v7)
    opt = FlatAdam(net, lr=1e-2)
    uc_nerf_amd.set_weight_cache("versions")
    try:
        with torch.no_grad():
            r0, _ = _call(mods, g, net, qfn)
            r0b, _ = _call(mods, g, net, qfn)
        assert torch.equal(r0, r0b)
        opt.zero_grad(set_to_none=True)
        rgb, depth = _call(mods, g, net, qfn)
        (rgb.sum() + depth.sum()).backward()
        opt.step()
        with torch.no_grad():
            r1, _ = _call(mods, g, net, qfn)
        uc_nerf_amd.set_weight_cache("verify")
        with torch.no_grad():
            r1v, _ = _call(mods, g, net, qfn)
        assert torch.equal(r1, r1v) and (r1 - r0).abs().max() > 1e-4
    finally:
        uc_nerf_amd.set_weight_cache("verify")


# ---------------------------------------------------------------------------------------------- the evaluation loop's ray builder in one launch
@pytest.mark.parametrize("chunk_idx,S", [(0, 90), (7, 90), (79, 90), (3, 30), (11, 192)])
def test_build_rays_test_in_one_launch_equals_the_three_launches(chunk_idx, S):
    """ucnerf_build_rays_test (utils/utils.py:600-739 in one launch, matrices read from device memory) against the composition it replaces --
    ucnerf_ray_gen (grid mode) -> torch indexing of the depth hypotheses -> ucnerf_sample_cascade -> ucnerf_ndc_project -- bit for bit."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.synthetic import cascade_outputs
    from uc_nerf_amd.utils import utils as U
    scene = _scene()
    outputs = cascade_outputs(scene)
    H, W, chunk = scene["H"], scene["W"], 1024
    start = chunk_idx * chunk
    n = min(chunk, H * W - start)
    t_rand = dev(torch.rand(n, S, generator=torch.Generator().manual_seed(chunk_idx)))
    nf_ref = torch.tensor([scene["near"], scene["far"]], device=DEV)
    got = ops.build_rays_test(H, W, start, n, S, scene["K"], scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], nf_ref,
                              [outputs["stage%d" % k]["depth_values"] for k in (1, 2, 3)], t_rand, want_ranges=True)
    rays_d, _, pix = ops.ray_gen(scene["K"], scene["c2w"], H=H, W=W, grid_start=start, n=n, device=torch.device(DEV), want_pix=True)
    ranges = U._stage_ranges(outputs, pix.long())
    z, pts = ops.sample_cascade(ranges, S, t_rand, scene["c2w"][:3, 3].contiguous(), rays_d)
    ndc = ops.ndc_project(pts, scene["w2cs"][0], scene["intrinsics"][0], [W - 1, H - 1], U._near_far_dict(ranges, S, scene["near"], scene["far"]))
    assert torch.equal(got["rays_d"], rays_d) and torch.equal(got["ranges"], ranges) and torch.equal(got["z"], z) and torch.equal(got["pts"], pts)
    for k in ("stage1", "stage2", "stage3", "ndc"):
        assert torch.equal(got[k], ndc[k]), k
    assert torch.equal(got["rays_o"], scene["c2w"][:3, 3])
    # ... and through the mirror of the reference's function: same tuple, same draws
    torch.manual_seed(3)
    a = U.build_rays_test(H, W, scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], nf_ref.view(1, 2), nf_ref, S, chunk=chunk, idx=chunk_idx, outputs=outputs)
    torch.manual_seed(3)
    t2 = torch.rand((n, S), device=DEV)
    b = ops.build_rays_test(H, W, start, n, S, scene["K"], scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], nf_ref, [outputs["stage%d" % k]["depth_values"] for k in (1, 2, 3)], t2)
    assert torch.equal(a[0], b["pts"]) and torch.equal(a[1], b["rays_d"]) and torch.equal(a[3], b["z"]) and tuple(a[4].shape) == (n, 3)
    assert all(torch.equal(a[2][k], b[k]) for k in ("stage1", "stage2", "stage3", "ndc")) and a[5]["pad"] == 0


# ---------------------------------------------------------------------------------------------- rendering()'s chunks in ONE launch
@pytest.mark.parametrize("n,S,given", [(1024, 90, True), (1000, 90, True), (1024, 90, False), (700, 45, False), (512, 96, True), (512, 33, True), (300, 90, True)])
def test_tail_route_serves_given_coordinates_and_any_sample_count_bit_identically(n, S, given):
    """Round 5: the gather-fused launch composites its rays in its own tail also for GIVEN coordinates (what rendering() is handed,
    network/renderer.py:215-255) and for sample counts that are no multiple of 32 (the 90 cascade samples): a block's tiles start at its own first
    sample.  And it makes the view-direction features itself from the rotation on the device.  Outputs equal the separate launches' bit for bit."""
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import cascade_outputs, init_ucnerf_state_dict, live_path_batch
    scene = _scene()
    outputs = cascade_outputs(scene)
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    src = ops.GatherSources(scene["vols"], scene["confidence"], scene["imgs"], scene["img_feat"], scene["w2cs"][1:], scene["intrinsics"][1:])
    pw = ops.PackedWeights.get(6, 0, torch.device(DEV), "bf16x3_fused")
    rp = ops.RenderPass(src, pw, pw.pack(flat_params_of(sd).to(DEV)), scene["c2w"][:3, 3].to(DEV), scene["w2cs"][0], scene["intrinsics"][0], scene["w2cs"][0],
                        scene["near"], scene["far"])
    rp.repack_sources()
    b = live_path_batch(scene, outputs, n, S, seed=n + S)
    coords = {"pts": b["rays_pts"], "stage1": b["rays_ndc"]["stage1"], "stage2": b["rays_ndc"]["stage2"], "stage3": b["rays_ndc"]["stage3"], "ndc": b["rays_ndc"]["ndc"]} if given else None
    nf = None if given else b["ranges"]
    ang, _ = ops.dir_feature(b["rays_dir"], scene["w2cs"][0])
    count = L.lib().ucnerf_fused_tail_launches
    fits = bool(L.lib().ucnerf_fused_tail_fits(n, S))
    outs = []
    for tail, dev_rot in ((0, False), (1, False), (1, True)):
        L.lib().ucnerf_set_fused_tail(tail)
        try:
            c0 = count()
            kw = dict(w2c_dir_dev=scene["w2cs"][0]) if dev_rot else dict(dir_feat=ang)
            o = rp(b["rays_dir"], b["depth_candidates"], near_far=nf, want=("acc", "weights", "var"), keep=("raw",), coords=coords, **kw)
            assert count() - c0 == int(bool(tail) and fits), (tail, fits)
            outs.append({k: v.clone() for k, v in o.items() if torch.is_tensor(v)})
        finally:
            L.lib().ucnerf_set_fused_tail(1)
    assert fits == (n != 300)                                # (300 rays spread over 150 of the 256 CUs: the separate launches serve them)
    for other in outs[1:]:
        for k in ("rgb", "depth", "acc", "weights", "var", "raw"):
            assert torch.equal(outs[0][k], other[k]), k


def test_rendering_under_no_grad_is_one_launch_per_chunk_and_unchanged():
    """rendering() on an evaluation chunk (train.py:254-272: 1024 pixels x 90 samples, coordinates from build_rays_test): the default route is now
    ONE launch -- view directions, gather, encoding, MLP, compositing -- with the same bits as the separate launches, and within 2e-5 of the exact kernel."""
    import uc_nerf_amd
    from uc_nerf_amd import _lib as L
    import utils.utils as U
    from uc_nerf_amd.synthetic import cascade_outputs, init_ucnerf_state_dict
    mods = _mods()
    scene = _scene()
    outputs = cascade_outputs(scene)
    net = _net(mods, 7, init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05))
    qfn = _qfn(mods)
    a = types.SimpleNamespace(view_num=7, feat_dim=97, img_downscale=1.0, use_color_volume=False, net_type="v2")
    nf = torch.tensor([[scene["near"], scene["far"]]] * 7, device=DEV)
    torch.manual_seed(11)
    pts, rd, ndc, z, ro, ndcp = U.build_rays_test(256, 320, scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], nf, nf[-1], 90, chunk=1024, idx=33, outputs=outputs)

    def call():
        pose = {"w2cs": scene["w2cs"].clone(), "intrinsics": scene["intrinsics"].clone()}
        return mods.renderer.rendering(a, pose, pts, ndc, z, rd, outputs, scene["imgs"], img_feat=scene["img_feat"], confidence=scene["confidence"],
                                       network_fn=net, network_query_fn=qfn, ndc_parameters=ndcp)
    count = L.lib().ucnerf_fused_tail_launches
    with torch.no_grad():
        c0 = count()
        rgb1, d1 = call()
        assert count() - c0 == 1
        L.lib().ucnerf_set_fused_tail(0)
        try:
            rgb0, d0 = call()
        finally:
            L.lib().ucnerf_set_fused_tail(1)
        assert torch.equal(rgb0, rgb1) and torch.equal(d0, d1)
        uc_nerf_amd.set_inference_precision("f32")
        try:
            rgbx, dx = call()
        finally:
            uc_nerf_amd.set_inference_precision("bf16x3_fused")
    assert (rgb1 - rgbx).abs().max().item() < 2e-5 and (d1 - dx).abs().max().item() < 4e-5


def test_differential_fuzz_of_the_render_pass_against_the_oracle():
    """tests/fuzz_render.py on a fixed set of 24 random cases (1..7 source views, ragged ray counts, 2..200 samples, cameras that send many samples
    outside the volumes and images; network/renderer.py:215-255 restated by the oracle).  The exact-f32 route holds the 1e-4 bar on every ray;
    channel-last sources in place and the one-launch tail route equal their counterparts bit for bit; the split-bf16 default route holds the bar
    on all but a few rays in 10^4 and stays within 3e-4 (its measured level with an unscaled density head: DESIGN.md section 5)."""
    import fuzz_render
    s = fuzz_render.run(cases=24, seed=3, verbose=False)
    assert s["identity_failures"] == 0, [r["identity_bad"] for r in s["rows"] if r["identity_bad"]]
    assert s["rays_above_bar"]["f32"] == 0 and max(s["worst"]["f32_rgb"], s["worst"]["f32_depth"], s["worst"]["f32_acc"]) <= 2e-5, s["worst"]
    assert s["tail_cases"] >= 5 and s["rays"] >= 5000
    for tag in ("fused", "given"):
        assert s["rays_above_bar"][tag] <= 1e-3 * s["rays"], s["rays_above_bar"]
        assert max(s["worst"][tag + "_rgb"], s["worst"][tag + "_depth"], s["worst"][tag + "_acc"]) <= 3e-4, s["worst"]


def test_differential_fuzz_of_the_resampling_launch():
    """tests/fuzz_sampling.py on a fixed set of 80 random cases (2..700 bin edges, 1..300 draws, degenerate weight rows, draws on cdf entries, ties in
    the merge, the from_coarse form): cdf, int64 indices, samples, the sorted merge and its rank, all bit for bit against the oracle's restatement of
    torch-CPU's arithmetic (data/ray_utils.py:98-141,216-219) -- and, where the cdf is not monotone, against torch-CPU's own binary search."""
    import fuzz_sampling
    s = fuzz_sampling.run(cases=80, seed=5, verbose=False)
    assert not s["failures"], s["failures"]
    assert s["rows"] > 3000


def test_differential_fuzz_of_the_backward_against_oracle_autograd():
    """tests/fuzz_grads.py on a fixed set of 12 random cases: ucnerf_render_fused_bwd (both training precisions, sources channel-major and channel-last
    in place, coordinates derived and given) against torch autograd through the oracle's rendering (network/renderer.py:215-255) RE-RUN ON THE DEVICE'S
    SIDES OF EVERY RELU (read back from the training forward's kept activations): every element of the 30 network gradients, the three volumes', the
    image features' and the confidence map's within the tests' bar with no outlier allowance, and every side that differs from the float64 oracle's
    sits at an argument within rounding noise of zero."""
    import fuzz_grads
    s = fuzz_grads.run(cases=12, seed=2, verbose=False)
    assert s["failures"] == 0, [r["bad"] for r in s["rows"] if r["bad"]]
    assert s["outside"] == {"f32": 0, "bf16x3": 0} and s["far_flips"] == {"f32": 0, "bf16x3": 0}
    assert s["worst"]["f32"] <= 3e-4 and s["elements"] > 5e6


def test_differential_fuzz_of_the_evaluation_ray_builder():
    """tests/fuzz_builders.py on a fixed set of 40 random cases: ucnerf_build_rays_test (one launch) against the oracle's composition of the reference's
    steps (utils/utils.py:248-271,600-739) -- random image sizes, chunk positions (the ragged last chunk included), sample counts, hypothesis counts
    and ranges, cameras; a stage with one hypothesis (near == far) must give the reference's infinities and NaNs in the same places."""
    import fuzz_builders
    s = fuzz_builders.run(cases=40, seed=4, verbose=False)
    assert not s["failures"], s["failures"]
    assert s["rays"] > 5000


def test_differential_fuzz_of_the_cost_volume_and_depth_regression():
    """tests/fuzz_mvs.py on a fixed set of 15 random cases (network/mvs_models.py:599-646): 1..8 source views, 1..32 channels, odd map
    sizes, 1..48 hypotheses, padding, cameras from nearly identical to far apart -- variance volume, in-view count, the gradient into the feature maps,
    probabilities, depth, confidence and the gradient at the logits against oracle/mvs_oracle.py, with the nearest-neighbour allowance of the fixture tests."""
    import fuzz_mvs
    s = fuzz_mvs.run(cases=15, seed=6, verbose=False)
    assert not s["failures"], s["failures"]


def test_pending_backward_keeps_its_forward_time_weight_stream(sd_v7):
    """A training forward, then the parameters change in place (an optimizer step of another loss) and a no_grad render of the SAME precision runs, then
    the first call's backward: it must differentiate through the weights its forward used.  (The no_grad route re-packs its cached stream in place; a
    training forward's stream therefore never enters that cache.)"""
    import uc_nerf_amd
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    gen = torch.Generator().manual_seed(11)
    r3, r1 = dev(torch.randn(g["z"].shape[0], 3, generator=gen)), dev(torch.randn(g["z"].shape[0], generator=gen))

    def grads(disturb):
        net = _net(mods, 7, sd_v7)
        rgb, depth = _call(mods, g, net, qfn)
        loss = (rgb * r3).sum() + (depth * r1).sum()
        if disturb:
            with torch.no_grad():
                for p in net.parameters():
                    p.mul_(1.25)
                _call(mods, g, net, qfn)
        loss.backward()
        return [p.grad.clone() for p in net.parameters() if p.grad is not None]

    uc_nerf_amd.install_dropin(precision="f32")
    try:
        a, b = grads(False), grads(True)
    finally:
        uc_nerf_amd.install_dropin()
    assert len(a) == len(b) == 30
    for x, y in zip(a, b):
        torch.testing.assert_close(y, x, atol=2e-5 * max(x.abs().max().item(), 1e-6), rtol=1e-4)


def test_stateful_fuzz_of_the_dropin_against_fresh_networks():
    """tests/fuzz_dropin.py: one long-lived network + session driven through random sequences of renders, training steps (Adam / FlatAdam), delayed
    backwards, in-place weight writes, load_state_dict, CPU round trips, source writes / replacements / layout switches and knob changes must answer
    every call like a freshly built network.  The first case is the one that found a stale "versions" weight cache in round 5 (a re-flattened
    parameter buffer at the old buffer's address with the old version count): seed 1 case 361."""
    import fuzz_dropin
    s = fuzz_dropin.run(steps=60, seed=1, verbose=False, only=361)
    assert not s["failures"], s["failures"]
    s = fuzz_dropin.run(cases=25, steps=50, seed=7, verbose=False)
    assert not s["failures"], s["failures"]
    assert s["renders"] > 300 and s["trains"] > 150


def test_stateful_fuzz_of_the_coarse_fine_pipeline_and_its_route_equivalences():
    """tests/fuzz_pipeline.py: one CoarseFineRenderer per case through random renders (1..4096 rays, jitter, random draws, reuse_coarse), new weights,
    sources written in place, route knobs (folded launches, in-kernel rays, tail route) and source layouts -- every render bit-identical to a freshly
    built renderer with the same settings AND to the plain route (nothing folded, no tail, channel-major sources, fine pass re-evaluating all depths)."""
    import fuzz_pipeline
    s = fuzz_pipeline.run(cases=60, steps=25, seed=9, verbose=False)
    assert not s["failures"], s["failures"]
    assert s["renders"] > 500


def test_empty_batches_go_through_every_route():
    """Zero rays (the empty last chunk of a script that slices its pixels, an empty shard of a sharded batch): rendering() under no_grad in every
    precision and under autograd, and the evaluation ray builder, return empty outputs -- and the backward of an empty batch leaves zero gradients --
    as torch hands empty tensors through the reference's own functions (network/renderer.py:215-255)."""
    import uc_nerf_amd
    import fuzz_dropin as FD
    from uc_nerf_amd import ops
    from uc_nerf_amd.synthetic import cascade_outputs, init_ucnerf_state_dict, live_path_batch, make_scene, scene_to
    mods = FD._mods()
    V = 4
    scene = scene_to(make_scene(seed=0, H=64, W=80, V=V, small_volumes=True), torch.device(DEV))
    outputs = cascade_outputs(scene)
    net = FD._net(mods, V, init_ucnerf_state_dict(seed=0, n_src=V - 1, sigma_scale=0.05, sigma_bias=0.05))
    qfn = _qfn(mods)
    args = types.SimpleNamespace(view_num=V, feat_dim=24 + 12 * (V - 1) + 1, img_downscale=1.0, use_color_volume=False, net_type="v2")
    b = live_path_batch(scene, outputs, 8, 30, seed=1)
    e = dict(rays_pts=b["rays_pts"][:0], rays_ndc={k: v[:0] for k, v in b["rays_ndc"].items()}, depth_candidates=b["depth_candidates"][:0], rays_dir=b["rays_dir"][:0])

    def call(grad):
        vf = {"stage%d" % (i + 1): {"volume_feature_no_ref": scene["vols"][i]} for i in range(3)}
        pose = {"w2cs": scene["w2cs"].clone(), "intrinsics": scene["intrinsics"].clone()}
        with torch.enable_grad() if grad else torch.no_grad():
            return mods.renderer.rendering(args, pose, e["rays_pts"], e["rays_ndc"], e["depth_candidates"], e["rays_dir"], vf, scene["imgs"], network_fn=net,
                                           img_feat=scene["img_feat"], network_query_fn=qfn, confidence=scene["confidence"])
    try:
        for prec in ("bf16x3_fused", "bf16x3", "f32"):
            uc_nerf_amd.set_inference_precision(prec)
            rgb, d = call(False)
            assert tuple(rgb.shape) == (0, 3) and tuple(d.shape) == (0,)
    finally:
        uc_nerf_amd.set_inference_precision("bf16x3_fused")
    rgb, d = call(True)
    assert tuple(rgb.shape) == (0, 3) and tuple(d.shape) == (0,)
    (rgb.sum() + d.sum()).backward()
    assert all(p.grad is None or not torch.count_nonzero(p.grad) for p in net.parameters())
    out = ops.build_rays_test(64, 80, 0, 0, 30, scene["K"], scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], torch.tensor([1., 4.], device=DEV),
                              [outputs["stage%d" % k]["depth_values"] for k in (1, 2, 3)])
    assert tuple(out["z"].shape) == (0, 30) and tuple(out["pts"].shape) == (0, 30, 3)


def test_differential_fuzz_of_the_stand_alone_network_launch():
    """tests/fuzz_mlp.py on a fixed set of 40 random cases: ucnerf_mlp_fwd and ucnerf_embed against the oracle (network/models.py:50-54,138-184;
    utils/run_nerf_helpers.py:52-53) -- 1..7 source views, ragged sample counts, both encoding layouts, directions per ray or per sample, features
    row-major or tiled, all three arithmetics; bars scaled by the float32 oracle's own distance from the float64 one where the input is ill-conditioned."""
    import fuzz_mlp
    s = fuzz_mlp.run(cases=40, seed=8, verbose=False)
    assert not s["failures"], s["failures"]
    assert s["worst"]["embed"] <= 2e-6 and s["samples"] > 20000


@pytest.mark.parametrize("fixture", ["g10_rendering", "g16_rendering_v4"])
def test_fp16_terms_render_the_reference_vectors_at_float32_level(fixture, sd_v7):
    """ucnerf_mlp_config.operand = 1 (ABI v6): the split precisions with fp16 terms -- the same three matrix-core products per algorithmic one, 11
    significant bits per term instead of 8.  rendering() under no_grad on G10 / G16 (network/renderer.py:215-255): within float32 rounding of the reference's
    vectors to 4e-6 / 1e-5 (the bf16 terms are held to 1e-4 and measure 2e-5), on every route (gather fused incl. the one-launch tail, two-kernel pass), and the
    training forward still packs bf16 terms."""
    import uc_nerf_amd
    from uc_nerf_amd import dropin, ops
    from test_hip_configs import close
    from test_oracle_golden import sd_v4_for_g16
    mods = _mods()
    g = load_golden(fixture)
    sd = sd_v7 if fixture == "g10_rendering" else sd_v4_for_g16(g)
    net, qfn = _net(mods, g["V"], sd), _qfn(mods)
    want_rgb, want_d = g["rgb_first" if "rgb_first" in g else "rgb"], g["depth_first" if "depth_first" in g else "depth"]
    try:
        uc_nerf_amd.set_split_operand("fp16")
        assert ops.split_operand() == "fp16"
        for prec in ("bf16x3_fused", "bf16x3"):
            uc_nerf_amd.set_inference_precision(prec)
            with torch.no_grad():
                rgb, depth = _call(mods, g, net, qfn)
            sess = dropin.session_of(net)
            assert sess.passes[(prec, 0)].pw.operand == 1
            close(rgb, want_rgb, 4e-6); close(depth, want_d, 1e-5)
        # a training call: f32 by default -- and the opt-in split training forward keeps bf16 terms whatever the setting
        uc_nerf_amd.set_training_precision("bf16x3")
        rgb, depth = _call(mods, g, net, qfn)
        (rgb.sum() + depth.sum()).backward()
        assert dropin.session_of(net).passes[("bf16x3", 0)].pw.operand == 0
        close(rgb.detach(), want_rgb, 1e-4)
    finally:
        uc_nerf_amd.set_split_operand("bf16")
        uc_nerf_amd.set_inference_precision("bf16x3_fused")
        uc_nerf_amd.set_training_precision("f32")
    with torch.no_grad():
        _call(mods, g, net, qfn)
    assert dropin.session_of(net).passes[("bf16x3_fused", 0)].pw.operand == 0


def test_fp16_terms_keep_every_bit_identity_and_refuse_what_they_cannot_serve():
    """With fp16 terms the route equivalences hold as with bf16 ones (channel-last sources in place == repacked, one-launch tail == two launches, on
    derived and on given coordinates: tests/fuzz_render.py on 16 random cases under set_split_operand("fp16")), the render fuzz's bar is met with a
    margin of 5x on every ray (well-conditioned inputs), and ucnerf_mlp_fwd_train refuses the operand loudly."""
    import uc_nerf_amd
    import fuzz_render
    from uc_nerf_amd import ops
    try:
        uc_nerf_amd.set_split_operand("fp16")
        s = fuzz_render.run(cases=16, seed=5, verbose=False)
        assert s["identity_failures"] == 0 and s["failures"] == 0, [r["bad"] for r in s["rows"] if r["bad"]]
        assert max(s["worst"]["fused_rgb"], s["worst"]["fused_depth"], s["worst"]["fused_acc"]) <= 2e-5, s["worst"]
        pw = ops.PackedWeights.get(3, 0, torch.device(DEV), "bf16x3")
        assert pw.operand == 1
        m = 64
        flat = torch.zeros(pw.n_params, device=DEV)
        with pytest.raises(RuntimeError, match="bf16 terms"):
            ops.mlp_fwd_train(pw, pw.pack(flat), torch.zeros(m, 3, device=DEV), torch.zeros(m, 3, device=DEV), torch.zeros(m, 24 + 36 + 1, device=DEV), 1)
    finally:
        uc_nerf_amd.set_split_operand("bf16")
