"""Row f1 of SURVEY.md 8(f): the render pass with the feature gather INSIDE the bf16x3 MLP kernel (precision "bf16x3_fused")
against the oracle and against the two-kernel pass it replaces (network/renderer.py:215-255)."""
import numpy as np
import pytest
import torch

from oracle import ucnerf_oracle as O
from test_hip_pipeline import close, oracle_pass, to_dev

DEV = "cuda:0"


def test_fused_pack_index_is_a_reordering_of_the_bias_net_sections():
    """CPU: precision 3 permutes the operands of the two bias nets and nothing else (same entries, same zero padding per row)."""
    import ctypes as C
    from uc_nerf_amd import _lib as L
    lib = L.lib()
    for v in (1, 2, 3, 4, 5, 6, 7, 8):
        idx = {}
        for prec in (1, 3):
            cfg = L.MlpConfig(v, 0, prec)
            n = lib.ucnerf_mlp_index_count(C.byref(cfg))
            buf = (C.c_int32 * n)()
            assert lib.ucnerf_mlp_pack_index(C.byref(cfg), buf) == 0
            idx[prec] = np.frombuffer(buf, dtype=np.int32).copy()
        a, b = idx[1], idx[3]
        assert a.shape == b.shape
        kd, kc = (24 + 4 * v + 15) // 16, (8 * v + 15) // 16
        slot = 4096                                                     # bf16 elements per k16-step (two half-steps)
        bd = slice(0, kd * slot)
        bc0 = (kd + 4 + 32 + 12) * slot
        bc = slice(bc0, bc0 + kc * slot)
        rest = np.ones(a.size, bool); rest[bd] = False; rest[bc] = False
        assert np.array_equal(a[rest], b[rest])
        for sec in (bd, bc):
            assert np.array_equal(np.sort(a[sec][a[sec] >= 0]), np.sort(b[sec][b[sec] >= 0]))
        assert not np.array_equal(a[bd], b[bd])


def _scene(V, seed=11):
    from uc_nerf_amd.synthetic import make_scene
    scene = make_scene(seed=seed, H=32, W=40, V=V + 1, small_volumes=True)
    scene["c2w"][:3, 3] = torch.tensor([0.05, -0.02, 0.01])
    scene["w2cs"][0, :3, :3] = torch.tensor([[0.9988, -0.0300, 0.0392], [0.0311, 0.9991, -0.0287], [-0.0383, 0.0299, 0.9988]])
    if V >= 2:
        scene["intrinsics"][2, 0, 0] *= 1.05
    return scene


@pytest.mark.gpu
@pytest.mark.parametrize("V,N,S,per_ray_ranges", [(6, 48, 20, False), (6, 7, 90, True), (6, 33, 64, False), (6, 1, 1, False),
                                                  (4, 21, 33, True), (3, 40, 16, False), (5, 9, 70, False), (2, 17, 8, True), (1, 5, 40, False),
                                                  (7, 30, 50, True), (8, 19, 64, False), (8, 3, 200, True)])
def test_gather_fused_pass_matches_oracle_and_the_two_kernel_pass(V, N, S, per_ray_ranges):
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    scene = _scene(V)
    g = torch.Generator().manual_seed(N * 1000 + S + V)
    sd = init_ucnerf_state_dict(seed=5, n_src=V, sigma_scale=0.1, sigma_bias=0.02)
    xs, ys = torch.randint(0, 40, (N,), generator=g).float(), torch.randint(0, 32, (N,), generator=g).float()
    _, rays_d, _ = O.get_rays_mvs_pixels(xs, ys, scene["K"], scene["c2w"])
    z = torch.sort(1.0 + 3.0 * torch.rand(N, S, generator=g), -1)[0]
    near_far = None
    if per_ray_ranges:
        lo = 0.8 + 0.4 * torch.rand(N, 3, generator=g)
        near_far = torch.stack([lo[:, 0], lo[:, 0] + 3, lo[:, 1], lo[:, 1] + 2.5, lo[:, 2], lo[:, 2] + 3.5], -1)
    want = oracle_pass(sd, scene, rays_d, z, near_far)

    sc = to_dev(scene)
    src = ops.GatherSources(sc["vols"], sc["confidence"], sc["imgs"], sc["img_feat"], sc["w2cs"][1:], sc["intrinsics"][1:])
    flat = flat_params_of(sd).to(DEV)
    nf = None if near_far is None else near_far.to(DEV)
    passes = {}
    for prec in ("bf16x3", "bf16x3_fused"):
        pw = ops.PackedWeights.get(src.V, 0, torch.device(DEV), prec)
        rp = ops.RenderPass(src, pw, pw.pack(flat), sc["c2w"][:3, 3], sc["w2cs"][0], sc["intrinsics"][0], sc["w2cs"][0],
                            scene["near"], scene["far"])
        rp.repack_sources()
        passes[prec] = rp
    two = passes["bf16x3"](rays_d.to(DEV), z.to(DEV), near_far=nf, keep=("raw",))
    one = passes["bf16x3_fused"](rays_d.to(DEV), z.to(DEV), near_far=nf, keep=("raw",))
    # against the oracle: the bars of the two-kernel bf16x3 pass
    scale = max(1.0, want["raw"][..., 3].abs().max().item())
    close(one["raw"][..., :3], want["raw"][..., :3], 5e-5)
    close(one["raw"][..., 3], want["raw"][..., 3], 5e-5 * scale, 1e-5)
    close(one["rgb"], want["rgb"], 1e-4); close(one["depth"], want["depth"], 1e-4)
    close(one["acc"], want["acc"], 1e-4); close(one["weights"], want["weights"], 5e-5, 1e-4)
    # against the pass it replaces: same features, same arithmetic but for the summation order inside the two bias nets
    close(one["raw"], two["raw"], 2e-5 * scale, 1e-5)
    close(one["rgb"], two["rgb"], 1e-5); close(one["depth"], two["depth"], 2e-5)
    # a precomputed view-direction feature gives the same render
    ang, _ = ops.dir_feature(rays_d.to(DEV), sc["w2cs"][0])
    pre = passes["bf16x3_fused"](rays_d.to(DEV), z.to(DEV), near_far=nf, dir_feat=ang)
    assert torch.equal(pre["rgb"], one["rgb"]) and torch.equal(pre["depth"], one["depth"])


@pytest.mark.gpu
def test_gather_fused_precision_refuses_what_it_cannot_serve():
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    scene = _scene(6)
    sd = init_ucnerf_state_dict(seed=5, sigma_scale=0.1, sigma_bias=0.02)
    sc = to_dev(scene)
    src = ops.GatherSources(sc["vols"], sc["confidence"], sc["imgs"], sc["img_feat"], sc["w2cs"][1:], sc["intrinsics"][1:])
    pw = ops.PackedWeights.get(src.V, 0, torch.device(DEV), "bf16x3_fused")
    ws = pw.pack(flat_params_of(sd).to(DEV))
    rp = ops.RenderPass(src, pw, ws, sc["c2w"][:3, 3], sc["w2cs"][0], sc["intrinsics"][0], sc["w2cs"][0], scene["near"], scene["far"])
    rays_d = torch.randn(4, 3, device=DEV)
    z = torch.sort(1 + 3 * torch.rand(4, 8, device=DEV), -1)[0]
    with pytest.raises(RuntimeError, match="channel-last"):        # sources not repacked
        rp(rays_d, z)
    rp.repack_sources()
    with pytest.raises(RuntimeError, match="keeps no features"):
        rp(rays_d, z, keep=("raw", "feats"))
    with pytest.raises(RuntimeError, match="uncertainty"):
        rp(rays_d, z, want=("u",))
    # the stand-alone MLP entry has no use for a stream in the fused operand order
    with pytest.raises(RuntimeError, match="precision 3"):
        ops.mlp_fwd(pw, ws, torch.zeros(8, 3, device=DEV), torch.zeros(8, 3, device=DEV), torch.zeros(8, src.F, device=DEV), 1)


@pytest.mark.gpu
def test_gather_fused_pipeline_at_the_headline_shape():
    """4096 rays x 64 + 128: the fused route against the two-kernel bf16x3 route on the bench's scene."""
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels
    scene = make_scene(seed=0)
    sd = init_ucnerf_state_dict(seed=1, sigma_scale=0.1, sigma_bias=0.02)
    H, W = scene["imgs"].shape[-2:]
    xs, ys = random_pixels(4096, H, W, seed=3)
    flat = flat_params_of(sd).to(DEV)
    two = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3").render(xs.to(DEV), ys.to(DEV))
    one = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3_fused").render(xs.to(DEV), ys.to(DEV))
    # The coarse passes agree to rounding; that rounding moves the fine depths by ~1e-6, and a sample that sits on the border of a source
    # view flips its in-view mask (a 0/1 input of the depth-bias net, utils/utils.py:742-799) with it: rays holding such a sample are
    # knife-edge rays of the reference's own algorithm (16 of 4096 here), every other ray must agree.
    close(one["coarse"]["weights"], two["coarse"]["weights"], 1e-5, 1e-4)
    close(one["coarse"]["rgb"], two["coarse"]["rgb"], 1e-5)
    same = (one["z_fine"] - two["z_fine"]).abs().amax(-1) < 1e-4
    assert same.float().mean().item() > 0.95          # (measured 0.968: the same share as bf16x3 against f32 in test_hip_pipeline)
    d_rgb = (one["rgb"] - two["rgb"]).abs().amax(-1)[same]
    d_depth = (one["depth"] - two["depth"]).abs()[same]
    assert (d_rgb < 2e-5).float().mean().item() > 0.99 and (d_depth < 5e-5).float().mean().item() > 0.99
    # ... and on IDENTICAL fine depths the two routes agree on every ray
    a = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3")
    b = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3_fused")
    a.pass_.repack_sources(); b.pass_.repack_sources()
    fa = a.pass_(two["rays_d"], two["z_fine"])
    fb = b.pass_(two["rays_d"], two["z_fine"])
    close(fb["rgb"], fa["rgb"], 1e-5); close(fb["depth"], fa["depth"], 2e-5)


@pytest.mark.gpu
def test_small_passes_take_the_two_kernel_route():
    """With fused_min_rounds > 0 a 'bf16x3_fused' renderer serves passes below that many tiles per wave with the two-kernel route (same
    parameters, second stream): bit-identical to a 'bf16x3' renderer; the default (0) is the fused kernel at every size."""
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, random_pixels
    scene = _scene(6)
    sd = init_ucnerf_state_dict(seed=5, sigma_scale=0.1, sigma_bias=0.02)
    flat = flat_params_of(sd).to(DEV)
    xs, ys = random_pixels(64, 32, 40, seed=1)
    two = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3").render(xs.to(DEV), ys.to(DEV))
    auto = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3_fused", fused_min_rounds=4)
    assert auto.fused_min_samples == 4 * torch.cuda.get_device_properties(0).multi_processor_count * 256
    a = auto.render(xs.to(DEV), ys.to(DEV))
    assert torch.equal(a["rgb"], two["rgb"]) and torch.equal(a["depth"], two["depth"])
    forced = CoarseFineRenderer(to_dev(scene), flat, 64, 128, precision="bf16x3_fused")
    assert forced.pass_small is None
    f = forced.render(xs.to(DEV), ys.to(DEV))
    same = (f["z_fine"] - two["z_fine"]).abs().amax(-1) < 1e-4
    assert ((f["rgb"] - two["rgb"]).abs().amax(-1)[same] < 2e-5).float().mean().item() > 0.9       # (border rows: in-mask knife edges)
    # new parameters reach both streams
    sd2 = init_ucnerf_state_dict(seed=6, sigma_scale=0.1, sigma_bias=0.02)
    auto.set_params(flat_params_of(sd2).to(DEV))
    b = auto.render(xs.to(DEV), ys.to(DEV))
    two2 = CoarseFineRenderer(to_dev(scene), flat_params_of(sd2).to(DEV), 64, 128, precision="bf16x3").render(xs.to(DEV), ys.to(DEV))
    assert torch.equal(b["rgb"], two2["rgb"])


@pytest.mark.gpu
def test_gather_fused_pass_on_randomised_shapes():
    """Seeded sweep over view counts, ray / sample counts (ragged tiles, more tiles than the grid holds with max_blocks = 1), depth ranges
    and background: the fused pass against the two-kernel pass on the same inputs."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    rng = np.random.default_rng(7)
    for case in range(14):
        V = int(rng.integers(1, 9))
        N, S = int(rng.integers(1, 80)), int(rng.integers(1, 130))
        ranges, white, max_blocks = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.choice([0, 0, 1, 3]))
        scene = _scene(V, seed=20 + case)
        g = torch.Generator().manual_seed(1000 + case)
        sd = init_ucnerf_state_dict(seed=case, n_src=V, sigma_scale=0.1, sigma_bias=0.02)
        xs, ys = 39 * torch.rand(N, generator=g), 1 + 29 * torch.rand(N, generator=g)      # (off the border rows: no in-mask knife edges)
        _, rays_d, _ = O.get_rays_mvs_pixels(xs, ys, scene["K"], scene["c2w"])
        z = torch.sort(1.0 + 3.0 * torch.rand(N, S, generator=g), -1)[0]
        nf = None
        if ranges:
            lo = 0.8 + 0.4 * torch.rand(N, 3, generator=g)
            nf = torch.stack([lo[:, 0], lo[:, 0] + 3, lo[:, 1], lo[:, 1] + 2.5, lo[:, 2], lo[:, 2] + 3.5], -1).to(DEV)
        sc = to_dev(scene)
        src = ops.GatherSources(sc["vols"], sc["confidence"], sc["imgs"], sc["img_feat"], sc["w2cs"][1:], sc["intrinsics"][1:])
        flat = flat_params_of(sd).to(DEV)
        outs = {}
        for prec in ("bf16x3", "bf16x3_fused"):
            pw = ops.PackedWeights.get(src.V, 0, torch.device(DEV), prec)
            rp = ops.RenderPass(src, pw, pw.pack(flat), sc["c2w"][:3, 3], sc["w2cs"][0], sc["intrinsics"][0], sc["w2cs"][0],
                                scene["near"], scene["far"], white_bkgd=white, max_blocks=max_blocks)
            rp.repack_sources()
            outs[prec] = rp(rays_d.to(DEV), z.to(DEV), near_far=nf, keep=("raw",))
        one, two = outs["bf16x3_fused"], outs["bf16x3"]
        what = "case %d: V=%d N=%d S=%d ranges=%s white=%s max_blocks=%d" % (case, V, N, S, ranges, white, max_blocks)
        scale = max(1.0, two["raw"][..., 3].abs().max().item())
        assert torch.isfinite(one["raw"]).all(), what
        assert (one["raw"] - two["raw"]).abs().max().item() < 3e-5 * scale, what
        assert (one["rgb"] - two["rgb"]).abs().max().item() < 2e-5 and (one["depth"] - two["depth"]).abs().max().item() < 3e-5, what
        assert (one["weights"] - two["weights"]).abs().max().item() < 2e-5, what
