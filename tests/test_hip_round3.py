"""Round 3: the two BASELINE.json configurations that had never run at their workload on the HIP path, and the routes added this round.

  configs[2]  32 768 rays in eight contiguous 4096-ray shards (parallel.ShardedRenderer), rendered back to back on one GPU:
              bit-identical to the unsharded render + an oracle spot check
  configs[4]  one training step (TrainStep: render -> the reference's loss mix -> backward -> bucket -> optimizer) on a live-shape
              batch of 2000 rays x 90 cascade samples; gradients against autograd through the CPU oracle, with the number of
              gradient elements outside 2e-4 recorded
  rendering()'s coordinates through the gather-fused kernel (the COORDS instantiation) against the reference fixtures G10 / G16
  the two backward modes (register-resident chain / layer-by-layer) against each other
"""
import os
import types

import pytest
import torch

from conftest import load_golden
from oracle import ucnerf_oracle as O
from test_hip_configs import close, coords_of, dev, record, render_pass_for
from test_oracle_golden import scaled_sd, sd_v4_for_g16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ---------------------------------------------------------------------------------------------- configs[2]
def test_configs2_32768_rays_in_eight_contiguous_shards_equal_the_unsharded_render():
    from uc_nerf_amd import parallel as P
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    scene = make_scene(seed=0)
    sd = init_ucnerf_state_dict(seed=0, sigma_scale=0.05, sigma_bias=0.05)
    n, world = 32768, 8
    xs, ys = random_pixels(n, 256, 320, seed=0)
    xs_d, ys_d = dev(xs), dev(ys)
    r = CoarseFineRenderer(scene_to(scene, torch.device(DEV)), flat_params_of(sd).to(DEV), 64, 128, precision="bf16x3_fused")
    full = r.render(xs_d, ys_d)
    keys = ("rgb", "depth", "acc", "z_fine")
    parts = []
    for rank in range(world):                            # the eight ranks of configs[2], one after the other on this GPU
        a, b = P.shard_range(n, rank, world)
        assert b - a == 4096 and a == 4096 * rank        # contiguous 4096-ray blocks
        out = P.ShardedRenderer(r.render, rank=rank, world=world).render(xs_d, ys_d, gather=())
        parts.append({k: out[k].clone() for k in keys})
    for k in keys:
        assert torch.equal(torch.cat([p[k] for p in parts], 0), full[k]), k      # a ray's result does not depend on its batch
    # oracle spot check: eight rays of every shard, teacher-forced on the device's own fine depths (rows 0 / H-1 are the
    # in-mask knife edge of these x-translated source views, test_hip_configs.py)
    idx = torch.cat([torch.nonzero((ys[4096 * k:4096 * (k + 1)] > 0) & (ys[4096 * k:4096 * (k + 1)] < 255))[:8, 0] + 4096 * k for k in range(world)])
    assert idx.numel() == 64
    with torch.no_grad():
        ref = O.render_coarse_fine(sd, scene, xs[idx], ys[idx], 64, 128, z_fine_override=full["z_fine"][idx].cpu())
    close(full["rgb"][idx], ref["rgb"], 1e-4); close(full["depth"][idx], ref["depth"], 1e-4); close(full["acc"][idx], ref["acc"], 1e-4)
    # the exact-f32 kernel on the same shards (its two-kernel pass holds 4096 x 192 feature rows at a time: shard by shard only)
    r32 = CoarseFineRenderer(scene_to(scene, torch.device(DEV)), flat_params_of(sd).to(DEV), 64, 128, precision="f32")
    for rank in (0, 7):
        out = P.ShardedRenderer(r32.render, rank=rank, world=world).render(xs_d, ys_d, gather=())
        sel = idx[(idx >= 4096 * rank) & (idx < 4096 * (rank + 1))]
        with torch.no_grad():
            ref32 = O.render_coarse_fine(sd, scene, xs[sel], ys[sel], 64, 128, z_fine_override=out["z_fine"][sel - 4096 * rank].cpu())
        close(out["rgb"][sel - 4096 * rank], ref32["rgb"], 1e-4); close(out["depth"][sel - 4096 * rank], ref32["depth"], 1e-4)
    record("configs2_32768_rays_8_shards", shards=world, rays=n, equal_to_unsharded=True,
           max_abs_rgb_spot=(full["rgb"][idx].cpu() - ref["rgb"]).abs().max().item(),
           max_abs_depth_spot=(full["depth"][idx].cpu() - ref["depth"]).abs().max().item())


# ---------------------------------------------------------------------------------------------- configs[4]
def _install():
    import uc_nerf_amd
    uc_nerf_amd.install_dropin()
    import network.models as models
    import network.renderer as renderer
    return models, renderer


@pytest.mark.parametrize("mode", ["chain", "layerwise"])
def test_configs4_training_step_on_a_live_shape_batch_against_oracle_autograd(mode):
    """2000 rays x 90 cascade samples (train.py:147-188): [50 patches of 6 x 6 | 150 other rays | 50 sparse-depth rays], the
    reference's loss mix (utils/loss.py via TrainStep), backward into the MLP parameters and the gather sources."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.synthetic import cascade_outputs, init_ucnerf_state_dict, live_path_batch, make_scene, scene_to
    from uc_nerf_amd.train_step import TrainStep
    from uc_nerf_amd.utils import loss as L
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    models, renderer = _install()
    scene_cpu = make_scene(seed=0)
    scene = scene_to(scene_cpu, torch.device(DEV))
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    a = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=6, netwidth=128, feat_dim=97, net_type="v2", view_num=7,
                              netchunk=1024, perturb=1.0, N_samples=90, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0, ckpt=None,
                              device=DEV, img_downscale=1.0, use_color_volume=False)
    kw, _, _, grad_vars = models.create_ucnerf(a, dir_embedder=True, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(sd)
    outputs = cascade_outputs(scene)
    N, S, patch_num, ps, n_rays = 2000, 90, 50, 6, 1950
    tr = live_path_batch(scene, outputs, N, S, seed=4)
    vols = [v.detach().clone().requires_grad_(True) for v in scene["vols"]]
    img_feat = scene["img_feat"].detach().clone().requires_grad_(True)
    conf = scene["confidence"].detach().clone().requires_grad_(True)
    vf = {"stage%d" % (k + 1): {"volume_feature_no_ref": vols[k]} for k in range(3)}
    gen = torch.Generator().manual_seed(5)
    target_s = torch.rand(N, 3, generator=gen)
    t_depth, t_w = 1 + 3 * torch.rand(N - n_rays, generator=gen), 2 * torch.rand(N - n_rays, generator=gen)
    dpt = torch.rand(patch_num, ps, ps, 1, generator=gen)
    mvs_out = {"stage%d" % k: {"depth": 1 + 3 * torch.rand(1, 64 * 2 ** (k - 1), 80 * 2 ** (k - 1), generator=gen)} for k in (1, 2, 3)}
    gt = {k: torch.where(torch.rand(v["depth"].shape, generator=gen) < 0.4, v["depth"] + 0.1, torch.zeros_like(v["depth"])) for k, v in mvs_out.items()}
    w = {k: (v > 0).float() * 0.7 for k, v in gt.items()}

    def render(idx):
        pose = {"w2cs": tr["pose_ref"]["w2cs"].clone(), "intrinsics": tr["pose_ref"]["intrinsics"].clone()}
        return renderer.rendering(a, pose, tr["rays_pts"][idx], {k: v[idx] for k, v in tr["rays_ndc"].items()}, tr["depth_candidates"][idx],
                                  tr["rays_dir"][idx], vf, scene["imgs"], img_feat=img_feat, confidence=conf, network_fn=net,
                                  network_query_fn=kw["network_query_fn"], white_bkgd=False)

    ops.set_backward_mode(mode)
    try:
        step = TrainStep(render, grad_vars, torch.optim.SGD(grad_vars, lr=0.0), n_rays, patch_num, ps)      # lr 0: the gradients stay readable
        out = step(dev(target_s), dev(t_depth), dev(t_w), dev(dpt), {k: {"depth": dev(v["depth"])} for k, v in mvs_out.items()}, gt, w)
    finally:
        ops.set_backward_mode("chain")
    torch.cuda.synchronize()

    # the oracle: the same batch through the CPU restatement, autograd all the way
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ov = [v.clone().requires_grad_(True) for v in scene_cpu["vols"]]
    oi, oc = scene_cpu["img_feat"].clone().requires_grad_(True), scene_cpu["confidence"].clone().requires_grad_(True)
    ndc = {k: v.cpu() for k, v in tr["rays_ndc"].items()}
    orgb, odepth = O.rendering(p, {"w2cs": scene_cpu["w2cs"].clone(), "intrinsics": scene_cpu["intrinsics"].clone()}, tr["rays_pts"].cpu(), ndc,
                               tr["depth_candidates"].cpu(), tr["rays_dir"].cpu(), ov, scene_cpu["imgs"], oi, oc, 7)
    want, parts = L.training_loss(orgb, odepth, target_s, t_depth, t_w, dpt, mvs_out, gt, w, n_rays=n_rays, patch_num=patch_num, patch_size=ps)
    want.backward()
    assert abs(float(out["loss"]) - want.item()) < 2e-4 * max(1.0, abs(want.item()))
    for k in TrainStep.TERMS:
        assert abs(float(out[k]) - parts[k].item()) < 2e-4 * max(1.0, abs(parts[k].item())), k

    # Parameter gradients are sums over 180 000 samples: element-wise at the whole-pass bar.  A gradient into a SOURCE voxel / pixel is
    # the sum over the handful of samples that touch it, and with ~1.6e8 unit evaluations per step a few pre-activations lie within
    # rounding of zero and fall on the other side of their relu in the two evaluations (different summation orders; DESIGN.md 5,
    # conditioning fact 3): those samples' contributions legitimately differ.  Sources: at most 1e-5 of the elements may sit outside the
    # bar, none by more than 1 % of the largest gradient; everything is counted and recorded.
    outside, total, worst, isolated = 0, 0, 0.0, {}
    def check(got, wg, name, source=False):
        nonlocal outside, total, worst
        gm = wg.abs().max().item()
        err = (got.cpu() - wg).abs()
        n_out = int((err > 2e-4 * gm + 2e-3 * wg.abs()).sum())
        outside += n_out
        total += wg.numel()
        worst = max(worst, (err.max() / max(gm, 1e-30)).item())
        if source:
            bad = err > 3e-4 * gm + 3e-3 * wg.abs() + 1e-9
            isolated[name] = int(bad.sum())
            assert int(bad.sum()) <= max(32, wg.numel() // 100000), "%s: %d elements outside the bar" % (name, int(bad.sum()))
            assert err.max().item() <= 1e-2 * gm, "%s: error %g against |g|max %g" % (name, err.max().item(), gm)
        else:
            torch.testing.assert_close(got.cpu(), wg, atol=3e-4 * gm + 1e-9, rtol=3e-3, msg=lambda s_: name + ": " + s_)
    for name, q in net.named_parameters():
        if p[name].grad is None:
            assert q.grad is None, name                   # the six tensors the reference's autograd never reaches
        else:
            check(q.grad, p[name].grad, name)
    for got, wg, name in zip(vols + [img_feat, conf], ov + [oi, oc], ("vol1", "vol2", "vol3", "img_feat", "confidence")):
        check(got.grad.reshape(wg.grad.shape), wg.grad, name, source=True)
    record("configs4_train_step_2000x90_%s" % mode, rays=N, samples_per_ray=S, loss=float(out["loss"]), gradient_elements=total,
           elements_outside_2e4_abs_2e3_rel=outside, worst_abs_error_over_gmax=worst, source_elements_outside_3e4_3e3=isolated)


# ---------------------------------------------------------------------------------------------- rendering()'s coordinates, gather fused
@pytest.mark.parametrize("fixture", ["g10_rendering", "g16_rendering_v4"])
def test_given_coordinates_through_the_gather_fused_kernel_match_the_reference_fixture(fixture, sd_v7):
    """The COORDS instantiation of the gather-fused kernel reads the world points and the stage / encoding copies rendering() is
    handed (per-ray cascade ranges, clamps and all), not (ray, depth): pinned by the reference's own rendering() outputs."""
    g = load_golden(fixture)
    sd = sd_v7 if fixture == "g10_rendering" else sd_v4_for_g16(g)
    rp, _ = render_pass_for(g, sd, precision="bf16x3_fused")
    rp.repack_sources()
    out = rp(dev(g["rays_d"]), dev(g["z"]), coords=coords_of(g), want=("acc", "weights"))
    first = "rgb_first" if "rgb_first" in g else "rgb"
    close(out["rgb"], g[first], 1e-4); close(out["depth"], g["depth_first" if "depth_first" in g else "depth"], 1e-4)
    rp3, _ = render_pass_for(g, sd, precision="bf16x3")
    rp3.repack_sources()
    two = rp3(dev(g["rays_d"]), dev(g["z"]), coords=coords_of(g), want=("acc", "weights"))
    close(out["rgb"], two["rgb"], 2e-5); close(out["depth"], two["depth"], 4e-5); close(out["weights"], two["weights"], 2e-5)


# ---------------------------------------------------------------------------------------------- the two backward modes
@pytest.mark.parametrize("m", [10, 70, 640, 650, 5000, 20480, 45060])
def test_backward_modes_agree(sd_v7, m):
    """The one-launch backward (gradient chain + weight-gradient launch, bwd_mode 0) against the layer-by-layer backward (bwd_mode 1).  The sizes walk
    the weight-gradient launch's scheduler through its corners: one stage per pair (fewer chunks than the look-ahead, END found before the first
    barrier), a ragged last stage, fewer chunks than blocks, more blocks homed on a pair than it has chunks, and a size where the counters do
    most of the hand-out."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of as flat_params
    gen = torch.Generator().manual_seed(12)
    S, F = 10, 97
    pts, feats = torch.rand(m, 3, generator=gen), torch.randn(m, F, generator=gen)
    feats[:, -1] = torch.rand(m, generator=gen)
    dirs = torch.nn.functional.normalize(torch.randn(m // S, 3, generator=gen), dim=-1)
    r = torch.randn(m, 4, generator=gen)
    pw = ops.PackedWeights.get(6, 0, torch.device(DEV))
    res = {}
    for mode in ("chain", "layerwise"):
        ops.set_backward_mode(mode)
        try:
            flat = dev(flat_params(sd_v7)).requires_grad_(True)
            fd = dev(feats).requires_grad_(True)
            (ops.mlp(flat, fd, dev(pts), dev(dirs), pw, S) * dev(r)).sum().backward()
            res[mode] = (flat.grad.clone(), fd.grad.clone())
        finally:
            ops.set_backward_mode("chain")
    for a_, b_ in zip(res["chain"], res["layerwise"]):
        torch.testing.assert_close(a_, b_, atol=5e-5 * b_.abs().max().item(), rtol=1e-3)


# ---------------------------------------------------------------------------------------------- configs[4]: bf16 features (channel-last source copies in bf16)
def _rounded(scene):
    """The scene with its heavy gather sources rounded to bf16 (what the bf16 channel-last copies hold), still fp32 tensors."""
    s = dict(scene)
    s["vols"] = [v.bfloat16().float() for v in scene["vols"]]
    s["imgs"] = scene["imgs"].bfloat16().float()
    s["img_feat"] = scene["img_feat"].bfloat16().float()
    return s


@pytest.mark.parametrize("precision", ["bf16x3_fused", "bf16x3", "f32"])
def test_bf16_source_copies_equal_fp32_copies_of_bf16_rounded_sources_and_stay_above_50_db(precision):
    """configs[4] "fp32 MLP / bf16 features": the gather reads bf16 channel-last copies.  (a) Nothing but the rounding of the sources changes:
    the render is BIT-IDENTICAL to the fp32-copy render of sources rounded to bf16 beforehand -- same corner order, same fp32 arithmetic --
    on the two-kernel pass and inside the gather-fused kernel; (b) quality gate: >= 50 dB against the fp32-source render."""
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
    scene = make_scene(seed=0)
    sd = init_ucnerf_state_dict(seed=0, sigma_scale=0.05, sigma_bias=0.05)
    xs, ys = random_pixels(2048, 256, 320, seed=3)
    xs_d, ys_d = dev(xs), dev(ys)
    flat = flat_params_of(sd).to(DEV)
    d = torch.device(DEV)
    full = CoarseFineRenderer(scene_to(scene, d), flat, 64, 128, precision=precision).render(xs_d, ys_d)
    full = {k: full[k].clone() for k in ("rgb", "depth", "acc")}
    r16 = CoarseFineRenderer(scene_to(scene, d), flat, 64, 128, precision=precision, sources_bf16=True)
    got = r16.render(xs_d, ys_d)
    assert r16.pass_.p.cl.bf16 == 1
    got = {k: got[k].clone() for k in ("rgb", "depth", "acc")}
    ref = CoarseFineRenderer(scene_to(_rounded(scene), d), flat, 64, 128, precision=precision).render(xs_d, ys_d)
    for k in ("rgb", "depth", "acc"):
        assert torch.equal(got[k], ref[k]), k
    mse = ((got["rgb"] - full["rgb"]) ** 2).mean().item()
    psnr = 99.0 if mse == 0 else -10.0 * torch.log10(torch.tensor(mse)).item()
    assert psnr >= 50.0, psnr
    record("configs4_bf16_source_copies_%s" % precision, rays=2048, psnr_db_vs_fp32_sources=psnr,
           max_abs_rgb=(got["rgb"] - full["rgb"]).abs().max().item(), max_abs_depth=(got["depth"] - full["depth"]).abs().max().item(),
           copy_bytes=int(r16.src._cl.numel() * 4))


def test_bf16_source_copies_through_the_rendering_mirror_forward_and_backward(sd_v7):
    """set_source_precision("bf16") under rendering(): the forward (given coordinates) and the training step's gradients equal those of
    the fp32 copies of sources rounded to bf16 beforehand; the source gradients arrive in fp32 for the fp32 tensors handed in."""
    import uc_nerf_amd
    models, renderer = _install()
    g = load_golden("g10_rendering")
    V = g["V"]
    e_p, _ = models.get_embedder(10, 0)
    e_d, _ = models.get_embedder(4, 0)
    qfn = lambda pts, vd, f, fn: renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d)
    args = types.SimpleNamespace(view_num=V, feat_dim=97, img_downscale=1.0, use_color_volume=False, net_type="v2")
    ndc = {"stage1": dev(g["ndc1"]), "stage2": dev(g["ndc2"]), "stage3": dev(g["ndc3"]), "ndc": dev(g["ndc"])}

    def run(rounded, prec):
        rnd = (lambda t: t.bfloat16().float()) if rounded else (lambda t: t)
        uc_nerf_amd.set_source_precision(prec)
        uc_nerf_amd.set_inference_precision("f32")      # one kernel for both runs (the default, bf16x3_fused, serves given coordinates from fp32 copies only
        try:                                            #  and hands bf16 copies to the two-kernel bf16x3 pass: another summation order in the bias nets)
            net = models.UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=97, view_num=V).to(DEV)
            net.load_state_dict({k: dev(v) for k, v in sd_v7.items()})
            vols = [dev(rnd(g["vol%d" % k])).requires_grad_(True) for k in (1, 2, 3)]
            img_feat, conf = dev(rnd(g["img_feat"])).requires_grad_(True), dev(g["conf"]).requires_grad_(True)
            vf = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
            pose = {"w2cs": dev(g["w2cs"]).clone(), "intrinsics": dev(g["K"]).repeat(V, 1, 1)}
            rgb, depth = renderer.rendering(args, pose, dev(g["pts"]), ndc, dev(g["z"]), dev(g["rays_d"]), vf, dev(rnd(g["imgs"])),
                                            network_fn=net, img_feat=img_feat, network_query_fn=qfn, confidence=conf)
            (rgb.sum() + 0.3 * depth.sum()).backward()
            with torch.no_grad():
                pose = {"w2cs": dev(g["w2cs"]).clone(), "intrinsics": dev(g["K"]).repeat(V, 1, 1)}
                ev = renderer.rendering(args, pose, dev(g["pts"]), ndc, dev(g["z"]), dev(g["rays_d"]), vf, dev(rnd(g["imgs"])),
                                        network_fn=net, img_feat=img_feat, network_query_fn=qfn, confidence=conf)
            return (rgb.detach(), depth.detach(), ev[0], ev[1],
                    [q.grad.clone() for q in net.parameters() if q.grad is not None] + [t.grad.clone() for t in vols + [img_feat, conf]])
        finally:
            uc_nerf_amd.set_source_precision("f32")
            uc_nerf_amd.set_inference_precision("bf16x3_fused")

    a = run(False, "bf16")
    b = run(True, "f32")
    for i in range(4):
        assert torch.equal(a[i], b[i]), i
    assert all(t.dtype == torch.float32 for t in a[4])
    for x, y in zip(a[4], b[4]):
        torch.testing.assert_close(x, y, atol=2e-5 * max(y.abs().max().item(), 1e-6), rtol=1e-4)      # (float atomics: the accumulation order differs run to run)
    # ... and the rounding is really there: against the unrounded fp32 sources the render moves
    c = run(False, "f32")
    assert not torch.equal(a[0], c[0]) and (a[0] - c[0]).abs().max().item() < 2e-2


# ---------------------------------------------------------------------------------------------- the 24-bit sets of the training step
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_kept_activation_sets_as_24_bit_floats_equal_the_fp32_sets_to_16_significant_bits(precision, sd_v7):
    """ucnerf_mlp_fwd_train keeps the ten [m,128] activation sets as 24-bit floats for the gradient chain (bwd_mode 0) and as fp32 for the
    layer-by-layer backward (bwd_mode 1): decoded, the former are the latter rounded to the top 24 bits (<= 2^-16 relative, half up) -- every
    element, both forward kernels, a sample count that is not a multiple of the 32-sample tile."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of as flat_params
    gen = torch.Generator().manual_seed(21)
    m, S, F = 4100, 10, 97
    pts, feats = torch.rand(m, 3, generator=gen), torch.randn(m, F, generator=gen)
    feats[:, -1] = torch.rand(m, generator=gen)
    dirs = torch.nn.functional.normalize(torch.randn(m // S, 3, generator=gen), dim=-1)
    pw = ops.PackedWeights.get(6, 0, torch.device(DEV), precision)
    ws = pw.pack(dev(flat_params(sd_v7)))
    raw24, s24 = ops.mlp_fwd_train(pw, ws, dev(pts), dev(dirs), dev(feats), S, "chain")
    raw32, s32 = ops.mlp_fwd_train(pw, ws, dev(pts), dev(dirs), dev(feats), S, "layerwise")
    assert torch.equal(raw24, raw32)
    for name in ops.KEPT_SETS:
        a, b = s24[name], s32[name]
        want = ((b.view(torch.int32) + 0x80) & ~0xff).view(torch.float32)          # the top 24 bits, dropped byte rounded half up
        assert torch.equal(a, want), name
        assert ((a - b).abs() <= b.abs() * 2.0 ** -16 + 1e-38).all(), name


# ---------------------------------------------------------------------------------------------- the gradient bucket over RCCL (one rank)
def test_flat_grad_bucket_and_barrier_over_the_rccl_backend_with_one_rank():
    """No multi-GPU node exists for the builder, so the data-parallel path has only crossed gloo.  This runs the same calls over the backend
    the driver's scaling series will use (`nccl` = RCCL) with a world of ONE rank on this GPU: process-group creation bound to the device,
    a device barrier, the max-over-ranks reduction of the bench and FlatGradBucket.allreduce on device tensors (a CPU tensor, a wrong dtype or a
    missing device binding fails here and not in the first 8-GPU run)."""
    import socket
    import torch.distributed as dist
    from uc_nerf_amd import parallel as P
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        dist.barrier()
        t = torch.tensor([1.25], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == 1.25
        gen = torch.Generator().manual_seed(3)
        params = [torch.nn.Parameter(torch.randn(n, generator=gen).to(dev)) for n in (7, 128 * 63, 1, 33)]
        grads = [torch.randn(p.shape, generator=gen).to(dev) for p in params]
        for p, g in zip(params[:3], grads[:3]):
            p.grad = g.clone()                       # the last parameter gets no gradient (the reference leaves six tensors without one)
        bucket = P.FlatGradBucket(params, n_scalars=2)
        out = bucket.allreduce(1.0, scalars=(torch.tensor(2.0, device=dev), torch.tensor(-1.5, device=dev)))
        torch.cuda.synchronize()
        assert out.device.type == "cuda" and out.tolist() == [2.0, -1.5]
        for p, g in zip(params[:3], grads[:3]):
            assert torch.equal(p.grad, g)
        assert params[3].grad is None
        for p, g in zip(params[:3], grads[:3]):
            p.grad = 2 * g
        bucket.allreduce(0.5, scalars=(torch.tensor(0.0, device=dev), torch.tensor(0.0, device=dev)))      # steady state: no read-back
        for p, g in zip(params[:3], grads[:3]):
            assert torch.equal(p.grad, g)
    finally:
        dist.destroy_process_group()
