"""Round 4 on the GPU.

  * what an UNMODIFIED caller of rendering() gets under no_grad (train.py:254-272): the gather-fused split-bf16 kernel, inside the 1e-4 bar on the
    reference's own rendering() fixtures G10 / G16;
  * flat parameter / gradient storage (uc_nerf_amd/flat.py) through the real backward: the p.grad of a rendering() step are views of ONE vector the
    kernels wrote, the six tensors the reference never differentiates keep None, FlatGradBucket reduces that vector in place, FlatAdam and
    torch.optim.Adam(grad_vars) (train.py:85-92) make the same step;
  * a frozen network with differentiable inputs (run_network_mvs route, network/renderer.py:78-106) against oracle autograd.
"""
import types

import pytest
import torch

from conftest import load_golden
from oracle import ucnerf_oracle as O
from test_hip_configs import close, dev
from test_oracle_golden import sd_v4_for_g16

pytestmark = pytest.mark.gpu


def _same_strides(a, b):
    """Equal strides on every dimension that has more than one element (torch leaves the others arbitrary)."""
    return all(n == 1 or x == y for n, x, y in zip(a.shape, a.stride(), b.stride()))
DEV = "cuda:0"
NO_GRAD = ("pts_bias_confidence_1.", "feature_linear_1.", "confi_linear.")


def _mods():
    import uc_nerf_amd
    uc_nerf_amd.install_dropin()
    import network.models as models
    import network.renderer as renderer
    return types.SimpleNamespace(models=models, renderer=renderer)


def _net(mods, V, sd):
    m = mods.models.UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=24 + 12 * (V - 1) + 1, skips=[4], view_num=V)
    m.load_state_dict(sd)
    return m.to(DEV)


def _qfn(mods):
    e_p, _ = mods.models.get_embedder(10, 0)
    e_d, _ = mods.models.get_embedder(4, 0)
    return lambda pts, vd, f, fn: mods.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d)


def _call(mods, g, net, qfn, vols=None, img_feat=None, conf=None, imgs=None):
    V = g["V"]
    args = types.SimpleNamespace(view_num=V, feat_dim=24 + 12 * (V - 1) + 1, img_downscale=1.0, use_color_volume=False, net_type="v2")
    vols = vols if vols is not None else [dev(g["vol%d" % k]) for k in (1, 2, 3)]
    vf = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
    pose = {"w2cs": dev(g["w2cs"]).clone(), "intrinsics": dev(g["K"]).repeat(V, 1, 1)}
    ndc = {"stage1": dev(g["ndc1"]), "stage2": dev(g["ndc2"]), "stage3": dev(g["ndc3"]), "ndc": dev(g["ndc"])}
    return mods.renderer.rendering(args, pose, dev(g["pts"]), ndc, dev(g["z"]), dev(g["rays_d"]), vf, imgs if imgs is not None else dev(g["imgs"]), network_fn=net,
                                   img_feat=img_feat if img_feat is not None else dev(g["img_feat"]), network_query_fn=qfn,
                                   confidence=conf if conf is not None else dev(g["conf"]))


# ---------------------------------------------------------------------------------------------- the default route
@pytest.mark.parametrize("fixture", ["g10_rendering", "g16_rendering_v4"])
def test_unmodified_caller_gets_the_gather_fused_kernel_within_the_parity_bar(fixture, sd_v7):
    """install_dropin() and nothing else: rendering() under no_grad runs on the kernel the headline benchmark is quoted on."""
    from uc_nerf_amd import dropin
    mods = _mods()
    assert dropin.inference_precision(None) == "bf16x3_fused"
    g = load_golden(fixture)
    sd = sd_v7 if fixture == "g10_rendering" else sd_v4_for_g16(g)
    net, qfn = _net(mods, g["V"], sd), _qfn(mods)
    with torch.no_grad():
        rgb, depth = _call(mods, g, net, qfn)
    sess = dropin.session_of(net)
    assert list(sess.passes) == [("bf16x3_fused", 0)]
    close(rgb, g["rgb_first" if "rgb_first" in g else "rgb"], 1e-4); close(depth, g["depth_first" if "depth_first" in g else "depth"], 1e-4)
    # the opt-out: the exact-f32 kernel through the same call
    import uc_nerf_amd
    uc_nerf_amd.install_dropin(precision="f32")
    try:
        with torch.no_grad():
            rgb32, d32 = _call(mods, g, net, qfn)
        assert ("f32", 0) in sess.passes
    finally:
        uc_nerf_amd.install_dropin(precision="bf16x3_fused")
    close(rgb, rgb32, 2e-5); close(depth, d32, 4e-5)


# ---------------------------------------------------------------------------------------------- flat storage
def test_rendering_backward_writes_one_flat_gradient_vector_that_bucket_and_optimizer_use_in_place(sd_v7):
    from uc_nerf_amd import parallel as P
    from uc_nerf_amd.flat import FlatAdam, FlatStore
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    gen = torch.Generator().manual_seed(3)
    r3, r1 = dev(torch.randn(g["z"].shape[0], 3, generator=gen)), dev(torch.randn(g["z"].shape[0], generator=gen))

    def step(net, opt, bucket=None):
        opt.zero_grad(set_to_none=True)
        rgb, depth = _call(mods, g, net, qfn)
        ((rgb * r3).sum() + (depth * r1).sum()).backward()
        if bucket is not None:
            red = bucket.allreduce(1.0, [rgb.detach().sum()])
            assert bucket.last_path == "in_place" and torch.equal(red[0], rgb.detach().sum())
        opt.step()

    net_a, net_b = _net(mods, 7, sd_v7), _net(mods, 7, sd_v7)
    st = FlatStore.of(net_a)
    assert st.is_flat() and st.flat.is_cuda and st.n == 181642
    opt_a = torch.optim.Adam(list(net_a.parameters()), lr=5e-4, betas=(0.9, 0.999))          # what train.py builds (train.py:85-92)
    opt_b = FlatAdam(net_b, lr=5e-4, betas=(0.9, 0.999))
    bucket = P.FlatGradBucket(list(net_a.parameters()), n_scalars=1)
    for it in range(3):
        step(net_a, opt_a, bucket)
        # the gradients of the step: views of ONE vector at the parameters' own offsets; untouched tensors keep None
        gf = st.flat_grad()
        assert gf is not None
        for (name, p), o in zip(net_a.named_parameters(), st.offsets):
            if any(t in name for t in NO_GRAD):
                assert p.grad is None, name
            else:
                assert p.grad is not None and p.grad.data_ptr() == gf.data_ptr() + 4 * o, name
        step(net_b, opt_b)
        for (name, a), b in zip(net_a.named_parameters(), net_b.parameters()):
            # (the two networks' gradients come from float atomics in another order: 1e-7-level noise, which Adam's normalisation turns into up to
            #  ~1 % of a step where |g| is near eps; a different optimizer would be off by the step itself, lr = 5e-4)
            torch.testing.assert_close(a, b, atol=5e-6, rtol=0, msg=lambda s_: "%s after step %d: %s" % (name, it, s_))
    assert st.is_flat() and FlatStore.of(net_b).is_flat()
    # the gradient values themselves against the reference's autograd (G10 carries them for the plain sum loss; here: oracle autograd on the same loss)
    p = {k: v.clone().requires_grad_(True) for k, v in sd_v7.items()}
    V = g["V"]
    rgb_o, depth_o = O.rendering(p, {"w2cs": g["w2cs"].clone(), "intrinsics": g["K"].repeat(V, 1, 1)}, g["pts"],
                                 {"stage1": g["ndc1"], "stage2": g["ndc2"], "stage3": g["ndc3"], "ndc": g["ndc"]}, g["z"], g["rays_d"],
                                 [g["vol1"], g["vol2"], g["vol3"]], g["imgs"], g["img_feat"], g["conf"], V)
    ((rgb_o * r3.cpu()).sum() + (depth_o * r1.cpu()).sum()).backward()
    net_c = _net(mods, 7, sd_v7)
    rgb, depth = _call(mods, g, net_c, qfn)
    ((rgb * r3).sum() + (depth * r1).sum()).backward()
    for name, q in net_c.named_parameters():
        if p[name].grad is None:
            assert q.grad is None, name
        else:
            w = p[name].grad
            torch.testing.assert_close(q.grad.cpu(), w, atol=3e-4 * w.abs().max().item() + 1e-8, rtol=3e-3, msg=lambda s_: name + ": " + s_)


def test_device_move_and_data_writes_keep_the_flat_buffer_the_parameters(sd_v7):
    from uc_nerf_amd import dropin
    from uc_nerf_amd.flat import FlatStore
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    net = mods.models.UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=97, skips=[4], view_num=7)
    net.load_state_dict(sd_v7)
    st = FlatStore.of(net)
    assert not st.flat.is_cuda
    net.to(DEV)                                            # re-flattened on the device by UCNeRF._apply
    assert st.is_flat() and st.flat.is_cuda
    with torch.no_grad():
        rgb0, _ = _call(mods, g, net, qfn)
        net.nerf.rgb_linear.bias.data.add_(0.3)            # a `.data` write (weights_init style): lands in the flat buffer, the next call packs from it
        rgb1, _ = _call(mods, g, net, qfn)
        assert (rgb1 - rgb0).abs().max() > 1e-3
        net.nerf.rgb_linear.bias.data = net.nerf.rgb_linear.bias.data - 0.3      # re-pointed: sync() gathers it back
        rgb2, _ = _call(mods, g, net, qfn)
        assert st.is_flat()
        close(rgb2, rgb0, 1e-6)
    assert dropin.session_of(net).store is st


# ---------------------------------------------------------------------------------------------- frozen network, differentiable inputs
def test_frozen_network_with_differentiable_features_matches_oracle_autograd(sd_v7):
    """run_network_mvs on a network whose parameters do not require grad, gradients wanted for the features: the backward takes its data
    gradients from the flat parameter vector saved by the forward, which must be the real parameters (round 3's advisor finding)."""
    mods = _mods()
    net = _net(mods, 7, sd_v7)
    net.requires_grad_(False)
    gen = torch.Generator().manual_seed(21)
    N, S, F = 24, 10, 97
    pts, feats = torch.rand(N, S, 3, generator=gen), torch.randn(N, S, F, generator=gen)
    feats[..., -1] = torch.rand(N, S, generator=gen)
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1)
    r = torch.randn(N, S, 4, generator=gen)
    qfn = _qfn(mods)
    for trial in range(2):                                  # twice: the second call must not see a stale vector either
        if trial == 1:
            with torch.no_grad():
                net.nerf.pts_linears[1].weight.mul_(1.5)
        fd = dev(feats).requires_grad_(True)
        out = qfn(dev(pts), dev(dirs), fd, net)
        (out * dev(r)).sum().backward()
        sd_now = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        fo = feats.clone().requires_grad_(True)
        ref = O.run_network_mvs(sd_now, pts, dirs, fo, n_src=6)
        (ref * r).sum().backward()
        close(out.detach(), ref.detach(), 1e-5 * max(1.0, ref.abs().max().item()), 1e-4)
        gs = fo.grad.abs().max().item()
        close(fd.grad, fo.grad, 2e-4 * gs, 2e-3)
        assert all(p.grad is None for p in net.parameters())


# ---------------------------------------------------------------------------------------------- the launches folded into the coarse pass
@pytest.mark.parametrize("S,M,u_kind", [(64, 128, "linspace"), (64, 128, "random"), (90, 45, "random"), (3, 7, "random"), (129, 300, "linspace"),
                                        (200, 128, "random"), (300, 1024, "linspace")])
def test_compositing_fused_with_the_resampling_equals_the_two_launches_bit_for_bit(S, M, u_kind, sd_v7):
    """ABI v4 `ucnerf_render_params.resample` / `ucnerf_composite_sample_pdf`: network/renderer.py:109-140 followed by data/ray_utils.py:216-219 in ONE
    launch.  Same weights (the lane split of the compositing kernel), hence the same cdf, the same searchsorted indices, the same depths."""
    from uc_nerf_amd import ops
    from test_hip_configs import render_pass_for
    g = load_golden("g10_rendering")
    rp, _ = render_pass_for(g, sd_v7, precision="bf16x3_fused")
    rp.repack_sources()
    gen = torch.Generator().manual_seed(S * 1000 + M)
    n = 37
    rays_d = dev(g["rays_d"][:n])
    z = dev(torch.sort(1.0 + 3.0 * torch.rand(n, S, generator=gen), -1)[0])
    u = torch.linspace(0., 1., M) if u_kind == "linspace" else torch.rand(n, M, generator=gen)
    u = dev(u)
    for want_rank in (False, True):
        two = rp(rays_d, z, want=("acc", "weights", "var"))
        hs = ops.sample_pdf(None, two["weights"], u, z_merge=z, want_inds=False, from_coarse=True, want_rank=want_rank)
        one = rp(rays_d, z, want=("acc", "weights", "var"), resample={"u": u, "want_rank": want_rank})
        for k in ("rgb", "depth", "acc", "weights", "var"):
            assert torch.equal(one[k], two[k]), k
        for k in ("samples", "z_sorted") + (("merge_rank",) if want_rank else ()):
            assert torch.equal(one[k], hs[k]), k
    assert one["z_sorted"].shape == (n, S + M) and bool((one["z_sorted"][:, 1:] >= one["z_sorted"][:, :-1]).all())


@pytest.mark.parametrize("n,perturb", [(512, 1.0), (4096, 1.0), (37, 0.0), (1, 1.0)])
def test_step_with_the_small_launches_folded_in_equals_the_old_launch_structure_bit_for_bit(n, perturb):
    """ABI v4 gen_rays / gen_depths + resample: the 64 + 128 step as four launches (the gather-fused coarse launch generates its own rays, depths and
    view-direction features; its compositing launch re-samples) against ray_gen_sample -> coarse -> composite -> sample_pdf -> fine -> composite."""
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
    scene = scene_to(make_scene(seed=0), torch.device(DEV))
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    r = CoarseFineRenderer(scene, flat_params_of(sd).to(DEV), 64, 128, precision="bf16x3_fused")
    xs, ys = random_pixels(4096, 256, 320, seed=0)
    xs, ys = dev(xs[:n].contiguous()), dev(ys[:n].contiguous())
    noise = dev(torch.rand(n, 64, generator=torch.Generator().manual_seed(100)))
    assert r.fold_launches and r.fold_rays is None         # the default: compositing + re-sampling folded; rays a launch of their own (measured faster) unless the pass takes the tail route
    r.fold_launches = r.fold_rays = False
    old = r.render(xs, ys, perturb=perturb, noise=noise if perturb > 0 else None)
    old = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in old.items()}
    for fl, fr in ((True, False), (False, True), (True, True)):
        r.fold_launches, r.fold_rays = fl, fr
        new = r.render(xs, ys, perturb=perturb, noise=noise if perturb > 0 else None)
        for k in ("rays_d", "z_coarse", "z_samples", "z_fine", "rgb", "depth", "acc", "weights", "var"):
            assert torch.equal(new[k], old[k]), (fl, fr, k)
        for k in ("rgb", "depth", "weights"):
            assert torch.equal(new["coarse"][k], old["coarse"][k]), (fl, fr, "coarse " + k)
        # reuse_coarse goes through the same folded launches (merge_rank from the fused compositing launch)
        ru = r.render(xs, ys, perturb=perturb, noise=noise if perturb > 0 else None, reuse_coarse=True)
        assert torch.equal(ru["rgb"], old["rgb"]) and torch.equal(ru["depth"], old["depth"])


@pytest.mark.parametrize("n,coarse_in_tail,fine_in_tail", [(512, True, True), (500, True, True), (700, True, True), (1024, True, True), (300, False, False), (37, False, False),
                                                         (4096, False, False)])
def test_compositing_and_resampling_in_the_tail_of_the_fused_launch_are_bit_identical(n, coarse_in_tail, fine_in_tail, monkeypatch):
    """Row f1 to the letter for small passes: with at most three rounds of tiles and (nearly) every CU owning whole rays, the gather-fused launch
    composites its rays itself and -- coarse pass -- draws the fine depths from them (K3 .. K9 in one launch, csrc/mlp_bf16.hip TAIL instantiation);
    larger or badly balanced passes keep the separate launches.  Same device functions: every output equal bit for bit; the library's counter says
    which passes took the route."""
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
    scene = scene_to(make_scene(seed=0), torch.device(DEV))
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    r = CoarseFineRenderer(scene, flat_params_of(sd).to(DEV), 64, 128, precision="bf16x3_fused")
    xs, ys = random_pixels(4096, 256, 320, seed=0)
    xs, ys = dev(xs[:n].contiguous()), dev(ys[:n].contiguous())
    noise = dev(torch.rand(n, 64, generator=torch.Generator().manual_seed(100)))
    count = L.lib().ucnerf_fused_tail_launches
    L.lib().ucnerf_set_fused_tail(0)
    c0 = count()
    old = r.render(xs, ys, perturb=1.0, noise=noise)
    old = {k: (v.clone() if torch.is_tensor(v) else {kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in old.items()}
    assert count() == c0                                   # switched off: the separate launches
    L.lib().ucnerf_set_fused_tail(1)
    r.fold_rays = False                                    # rays from their own launch
    new = r.render(xs, ys, perturb=1.0, noise=noise)
    assert count() - c0 == int(coarse_in_tail) + int(fine_in_tail)
    for k in ("rays_d", "z_coarse", "z_fine", "rgb", "depth", "weights"):
        assert torch.equal(new[k], old[k]), k
    r.fold_rays = None                                     # the default: on the tail route the coarse launch's blocks generate their own rays as well
    c1 = count()
    new = r.render(xs, ys, perturb=1.0, noise=noise)
    assert count() - c1 == int(coarse_in_tail) + int(fine_in_tail)
    for k in ("rays_d", "z_coarse", "z_samples", "z_fine", "rgb", "depth", "acc", "weights", "var"):
        assert torch.equal(new[k], old[k]), k
    for k in ("rgb", "depth", "weights"):
        assert torch.equal(new["coarse"][k], old["coarse"][k]), "coarse " + k
    ru = r.render(xs, ys, perturb=1.0, noise=noise, reuse_coarse=True)
    assert torch.equal(ru["rgb"], old["rgb"]) and torch.equal(ru["depth"], old["depth"])


@pytest.mark.parametrize("n,S,M,white", [(768, 32, 64, False), (512, 96, 32, True), (256, 128, 0, False), (256, 256, 0, True), (512, 64, 128, True)])
def test_tail_route_of_a_single_pass_covers_the_sample_counts_and_outputs_of_the_separate_launches(n, S, M, white, monkeypatch):
    """One RenderPass call per case, with and without the tail route: S = 32 .. 256 (one to four samples per lane in the compositing wave), the white
    background, every optional output, re-sampling to M new depths with uniform and random draws.  Bit-identical; the counter proves the route ran."""
    from uc_nerf_amd import _lib as L, ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
    scene = scene_to(make_scene(seed=0), torch.device(DEV))
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    src = ops.GatherSources(scene["vols"], scene["confidence"], scene["imgs"], scene["img_feat"], scene["w2cs"][1:], scene["intrinsics"][1:])
    pw = ops.PackedWeights.get(6, 0, torch.device(DEV), "bf16x3_fused")
    rp = ops.RenderPass(src, pw, pw.pack(dev(flat_params_of(sd))), dev(scene["c2w"][:3, 3]), scene["w2cs"][0], scene["intrinsics"][0], scene["w2cs"][0],
                        scene["near"], scene["far"], white)
    rp.repack_sources()
    xs, ys = random_pixels(n, 256, 320, seed=5)
    rays_d, _, _ = ops.ray_gen(scene["K"].cpu(), scene["c2w"].cpu(), xs=dev(xs), ys=dev(ys))
    z, _ = ops.sample_stratified(None, S, n=n, near=float(scene["near"]), far=float(scene["far"]), device=torch.device(DEV), perturb=1.0,
                                 noise=dev(torch.rand(n, S, generator=torch.Generator().manual_seed(S))))
    ang, _ = ops.dir_feature(rays_d, scene["w2cs"][0])
    count = L.lib().ucnerf_fused_tail_launches
    assert L.lib().ucnerf_fused_tail_fits(n, S) == 1
    for u in ((None,) if M == 0 else (torch.linspace(0., 1., M, device=DEV), dev(torch.rand(n, M, generator=torch.Generator().manual_seed(M))))):
        res = None if u is None else {"u": u, "want_rank": True}
        outs = []
        for tail in (False, True):
            L.lib().ucnerf_set_fused_tail(int(tail))
            c0 = count()
            o = rp(rays_d, z, want=("acc", "weights", "var"), keep=("raw",), dir_feat=ang, resample=res)
            assert count() - c0 == int(tail)
            outs.append({k: v.clone() for k, v in o.items() if torch.is_tensor(v)})
        assert set(outs[0]) == set(outs[1]) and {"rgb", "depth", "acc", "weights", "var", "raw"} <= set(outs[0])
        if res is not None:
            assert {"samples", "z_sorted", "merge_rank"} <= set(outs[0])
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), (k, "uniform" if u is not None and u.dim() == 1 else "random")


# ---------------------------------------------------------------------------------------------- sources handed over channel-last, zero copy
def _bench_scene_channel_last(seed=0):
    from uc_nerf_amd import ops
    from uc_nerf_amd.synthetic import make_scene, scene_to
    scene = scene_to(make_scene(seed=seed), torch.device(DEV))
    cl = ops.ChannelLastSources.from_reference_layout(scene["vols"], scene["imgs"], scene["img_feat"])
    scene_cl = dict(scene, vols=cl.vols, imgs=cl.imgs, img_feat=cl.img_feat)
    return scene, scene_cl, cl


def test_channel_last_sources_are_read_zero_copy_and_render_bit_identically():
    """ops.ChannelLastSources: the three cascade volumes as [1,8,D,h,w] tensors over [D,h,w,8] memory (torch's channels_last_3d), the source images /
    image features as channels_last [V,3,H,W] / [V,8,H,W] -- each its OWN allocation (ABI v5), recognised by stride.  A pass bound to them reads them
    in place (utils/utils.py:742-799,833-893 sample the same values): no repack launch, bit-identical renders."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, random_pixels
    scene, scene_cl, cl = _bench_scene_channel_last()
    for a, b in zip(cl.vols + [cl.imgs, cl.img_feat], scene["vols"] + [scene["imgs"], scene["img_feat"]]):
        assert a.shape == b.reshape(a.shape).shape and torch.equal(a, b.reshape(a.shape))
    assert cl.vols[0].is_contiguous(memory_format=torch.channels_last_3d)
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    flat = flat_params_of(sd).to(DEV)
    xs, ys = random_pixels(700, 256, 320, seed=0)
    noise = dev(torch.rand(700, 64, generator=torch.Generator().manual_seed(100)))
    for prec in ("bf16x3_fused", "bf16x3", "f32"):
        ra = CoarseFineRenderer(scene, flat, 64, 128, precision=prec)
        rb = CoarseFineRenderer(scene_cl, flat, 64, 128, precision=prec)
        assert rb.src.zero_copy and not any(ra.src.inplace) and rb.src._cl is None
        assert [rb.pass_.p.cl.vol[k] for k in range(3)] == [v.data_ptr() for v in cl.vols] and rb.pass_.p.cl.rgb_stride == 3
        assert rb.pass_.p.cl.img_feat == cl.img_feat.data_ptr() and rb.pass_.p.cl.imgs == cl.imgs.data_ptr()
        before = [t.clone() for t in cl.vols + [cl.imgs, cl.img_feat]]
        a = ra.render(dev(xs), dev(ys), perturb=1.0, noise=noise)
        b = rb.render(dev(xs), dev(ys), perturb=1.0, noise=noise, repack=True)       # (repack=True is a no-op for them: the arrays are the caller's)
        assert rb.src._cl is None and all(torch.equal(t, u) for t, u in zip(cl.vols + [cl.imgs, cl.img_feat], before))
        for k in ("rgb", "depth", "acc", "weights", "z_fine"):
            assert torch.equal(a[k], b[k]), (prec, k)
    # an entry point that reads the reference's channel-major layout refuses them (their pointers are withheld) instead of reading the wrong layout
    src = ops.GatherSources(cl.vols, scene["confidence"], cl.imgs, cl.img_feat, scene["w2cs"][1:], scene["intrinsics"][1:])
    assert src.zero_copy
    pts = torch.rand(8, 4, 3, device=DEV)
    with pytest.raises(RuntimeError, match="null"):
        ops.feat_gather_fwd(src, pts, pts, pts, pts)
    # every source is recognised on its own: a tensor that is not channel-last takes the repack route, the others stay in place
    moved = ops.GatherSources(cl.vols, scene["confidence"], cl.imgs.contiguous(), cl.img_feat, scene["w2cs"][1:], scene["intrinsics"][1:])
    assert not moved.zero_copy and moved.inplace == [True, True, True, True, False]


def test_training_through_channel_last_sources_returns_gradients_in_their_layout(sd_v7):
    """rendering() under autograd with the sources handed over channel-last (ChannelLastSources): the forward equals the repack route bit for bit, and
    the gradients of the volumes / image features arrive with the inputs' shapes AND strides -- accumulated channel-last by the gather backward, no
    scratch, no transposing pass -- equal to the channel-major route's (float-atomic order aside)."""
    from uc_nerf_amd import ops
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    gen = torch.Generator().manual_seed(7)
    r3, r1 = dev(torch.randn(g["z"].shape[0], 3, generator=gen)), dev(torch.randn(g["z"].shape[0], generator=gen))
    vols_ref = [dev(g["vol%d" % k]) for k in (1, 2, 3)]
    cl = ops.ChannelLastSources.from_reference_layout(vols_ref, dev(g["imgs"]), dev(g["img_feat"]))

    seen = {}

    def run(zero_copy):
        net = _net(mods, 7, sd_v7)
        if zero_copy:
            vols = [v.detach().requires_grad_(True) for v in cl.vols]
            img_feat = cl.img_feat.detach().requires_grad_(True)
            img_feat.register_hook(lambda gr: seen.update(stride=gr.stride(), ptr=gr.data_ptr()))      # the gradient as the backward hands it upstream
            imgs = cl.imgs
        else:
            vols = [v.clone().requires_grad_(True) for v in vols_ref]
            img_feat = dev(g["img_feat"]).requires_grad_(True)
            imgs = dev(g["imgs"])
        conf = dev(g["conf"]).requires_grad_(True)
        V = g["V"]
        args = types.SimpleNamespace(view_num=V, feat_dim=97, img_downscale=1.0, use_color_volume=False, net_type="v2")
        vf = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
        pose = {"w2cs": dev(g["w2cs"]).clone(), "intrinsics": dev(g["K"]).repeat(V, 1, 1)}
        ndc = {"stage1": dev(g["ndc1"]), "stage2": dev(g["ndc2"]), "stage3": dev(g["ndc3"]), "ndc": dev(g["ndc"])}
        rgb, depth = mods.renderer.rendering(args, pose, dev(g["pts"]), ndc, dev(g["z"]), dev(g["rays_d"]), vf, imgs, network_fn=net,
                                             img_feat=img_feat, network_query_fn=qfn, confidence=conf)
        ((rgb * r3).sum() + (depth * r1).sum()).backward()
        from uc_nerf_amd import dropin
        return rgb.detach(), depth.detach(), vols, img_feat, conf, net, dropin.session_of(net)

    rgb_a, d_a, vols_a, if_a, conf_a, net_a, _ = run(False)
    rgb_b, d_b, vols_b, if_b, conf_b, net_b, sess = run(True)
    assert sess.src.zero_copy and sess.src._cl is None
    assert torch.equal(rgb_a, rgb_b) and torch.equal(d_a, d_b)
    # gradients: same shapes AND strides as the inputs (channel-last memory), segments of the step's one zero-filled gradient pool
    for k, (v, w) in enumerate(zip(vols_b, vols_a)):
        assert v.grad.shape == v.shape and _same_strides(v.grad, v), (k, v.grad.stride(), v.stride())
        torch.testing.assert_close(v.grad, w.grad.reshape(v.shape), atol=2e-5 * max(w.grad.abs().max().item(), 1e-6), rtol=1e-4)
    assert vols_b[1].grad.data_ptr() > vols_b[0].grad.data_ptr() and vols_b[1].grad.data_ptr() - vols_b[0].grad.data_ptr() == 4 * vols_b[0].numel()
    # the image features' gradient goes upstream (to the feature network's backward) with the input's strides
    assert seen["stride"] == if_b.stride() or all(n == 1 or a == b for n, a, b in zip(if_b.shape, seen["stride"], if_b.stride()))
    torch.testing.assert_close(if_b.grad, if_a.grad.reshape(if_b.shape), atol=2e-5 * max(if_a.grad.abs().max().item(), 1e-6), rtol=1e-4)
    torch.testing.assert_close(conf_b.grad, conf_a.grad, atol=2e-5 * max(conf_a.grad.abs().max().item(), 1e-6), rtol=1e-4)
    for (name, p), q in zip(net_a.named_parameters(), net_b.parameters()):
        if p.grad is None:
            assert q.grad is None, name
        else:
            torch.testing.assert_close(q.grad, p.grad, atol=2e-5 * max(p.grad.abs().max().item(), 1e-6), rtol=1e-4, msg=lambda s_: name + ": " + s_)


# ---------------------------------------------------------------------------------------------- the training step as one HIP-graph replay
def test_graphed_training_step_follows_the_eager_one(sd_v7):
    """train_step.GraphedStep: zero_grad + rendering() + loss + backward + FlatAdam captured once, replayed per step -- the parameters follow the
    eager step's (train.py:147-188 issues the same work launch by launch) up to the order of the gradients' float atomics."""
    from uc_nerf_amd.flat import FlatAdam
    from uc_nerf_amd.train_step import GraphedStep
    mods = _mods()
    g = load_golden("g10_rendering")
    qfn = _qfn(mods)
    gen = torch.Generator().manual_seed(5)
    target = dev(torch.rand(g["z"].shape[0], 3, generator=gen))

    # everything the step reads is on the device already: a capture admits no host-to-device copy
    V = g["V"]
    args = types.SimpleNamespace(view_num=V, feat_dim=24 + 12 * (V - 1) + 1, img_downscale=1.0, use_color_volume=False, net_type="v2")
    vf = {"stage%d" % k: {"volume_feature_no_ref": dev(g["vol%d" % k])} for k in (1, 2, 3)}
    w2cs, Ks = dev(g["w2cs"]), dev(g["K"]).repeat(V, 1, 1)
    ndc = {"stage1": dev(g["ndc1"]), "stage2": dev(g["ndc2"]), "stage3": dev(g["ndc3"]), "ndc": dev(g["ndc"])}
    pts, z, rays_d, imgs, img_feat, conf = dev(g["pts"]), dev(g["z"]), dev(g["rays_d"]), dev(g["imgs"]), dev(g["img_feat"]), dev(g["conf"])

    def make(net, opt):
        def step():
            opt.zero_grad(set_to_none=True)
            rgb, depth = mods.renderer.rendering(args, {"w2cs": w2cs, "intrinsics": Ks}, pts, ndc, z, rays_d, vf, imgs, network_fn=net, img_feat=img_feat,
                                                 network_query_fn=qfn, confidence=conf)
            loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
            loss.backward()
            opt.step()
            return loss
        return step

    net_e, net_g = _net(mods, 7, sd_v7), _net(mods, 7, sd_v7)
    step_e = make(net_e, FlatAdam(net_e, lr=5e-4))
    graphed = GraphedStep(make(net_g, FlatAdam(net_g, lr=5e-4, capturable=True)), warmup=2)     # (the 2 warm-up steps have stepped net_g; the capture itself executes nothing)
    for _ in range(2):
        loss_e = step_e()
    losses = [loss_e.item()]
    for _ in range(4):
        losses.append(step_e().item())
        lg = graphed.replay()
        for (name, a), b in zip(net_e.named_parameters(), net_g.parameters()):
            torch.testing.assert_close(a, b, atol=2e-5, rtol=0, msg=lambda s_: name + ": " + s_)
        assert abs(lg.item() - losses[-1]) < 1e-4 * max(1.0, abs(losses[-1]))
    assert losses[-1] < losses[0]                            # (and it trains)
