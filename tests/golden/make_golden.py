"""Generates tests/golden/*.npz by running the UC-NeRF reference's own Python on seeded inputs.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference has no tests/fixtures of its own; these captured vectors are what
pins the oracle (oracle/ucnerf_oracle.py) and, through it, the HIP kernels.  Fixtures hold data only:
inputs (including every random draw, captured by re-seeding) and the reference's outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

ref = _ref_import.load()
torch.set_num_threads(1)   # fixtures must not depend on intra-op partitioning


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024, len(out)))


def rand_pose(g, scale=0.15):
    """Small random rotation (Rodrigues) + translation -> 4x4."""
    w = (torch.rand(3, generator=g) - 0.5) * 2 * scale
    th = w.norm()
    k = w / th
    Kx = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = torch.eye(3) + torch.sin(th) * Kx + (1 - torch.cos(th)) * (Kx @ Kx)
    M = torch.eye(4)
    M[:3, :3] = R
    M[:3, 3] = (torch.rand(3, generator=g) - 0.5) * 0.2
    return M


# ------------------------------------------------------------------ G1 ray generation
def g1():
    g = torch.Generator().manual_seed(101)
    H, W = 16, 20
    K = torch.tensor([[21.5, 0, 9.25], [0, 19.75, 8.5], [0, 0, 1]])
    c2w = rand_pose(g)
    dirs = ref.ray_utils.get_ray_directions(H, W, K)
    ro, rd = ref.ray_utils.get_rays(dirs, c2w[:3, :4])
    mo, md, mpix = ref.utils.get_rays_mvs(H, W, K, c2w, isRandom=False, chunk=64, idx=2)
    fo, fd, fpix = ref.utils.get_rays_mvs(H, W, K, c2w, isRandom=False)
    ho, hd = ref.helpers.get_rays(H, W, 23.0, c2w[:3, :4])
    # 400x400 plumbing config: corner crop only (c1)
    K4 = torch.tensor([[500., 0, 200], [0, 500, 200], [0, 0, 1]])
    d4 = ref.ray_utils.get_ray_directions(400, 400, K4)
    o4, r4 = ref.ray_utils.get_rays(d4, c2w[:3, :4])
    sel = torch.tensor([0, 1, 399, 400, 79999, 159999])
    save("g1_raygen", H=H, W=W, K=K, c2w=c2w, dirs=dirs, rays_o=ro, rays_d=rd, mvs_o=mo, mvs_d=md, mvs_pix=mpix,
         mvs_full_d=fd, mvs_full_pix=fpix, gl_focal=23.0, gl_o=ho, gl_d=hd, K400=K4, sel400=sel, d400=r4[sel],
         o400=o4[sel])


# ------------------------------------------------------------------ G2 NDC ray warp
def g2():
    g = torch.Generator().manual_seed(102)
    N = 64
    ro = torch.randn(N, 3, generator=g) * 0.3
    rd = torch.randn(N, 3, generator=g)
    rd[:, 2] = -rd[:, 2].abs() - 0.2
    o1, d1 = ref.ray_utils.get_ndc_rays(48, 64, [55.0, 57.5], 1.0, ro, rd)
    o2, d2 = ref.helpers.ndc_rays(48, 64, 55.0, 1.0, ro, rd)
    save("g2_ndc_rays", H=48, W=64, focal2=[55.0, 57.5], focal=55.0, near=1.0, rays_o=ro, rays_d=rd, o_ru=o1, d_ru=d1,
         o_h=o2, d_h=d2)


# ------------------------------------------------------------------ G3 depth sampling
def g3():
    g = torch.Generator().manual_seed(103)
    N, S = 48, 64
    rays = torch.cat([torch.randn(N, 3, generator=g) * 0.1, torch.randn(N, 3, generator=g),
                      1.0 + torch.rand(N, 1, generator=g), 3.0 + torch.rand(N, 1, generator=g)], -1)
    p0, _, _, z0 = ref.ray_utils.ray_marcher(rays, S, perturb=0)
    pl, _, _, zl = ref.ray_utils.ray_marcher(rays, S, lindisp=True, perturb=0)
    torch.manual_seed(7)
    noise = torch.rand(N, S)
    torch.manual_seed(7)
    p1, _, _, z1 = ref.ray_utils.ray_marcher(rays, S, perturb=1.0)
    torch.manual_seed(7)
    p5, _, _, z5 = ref.ray_utils.ray_marcher(rays, S, perturb=0.5)
    # live cascade sampler (utils/utils.py:684-706) via build_rays_test on a tiny scene
    H, W, NS = 8, 12, 9
    outputs = {}
    for k, d in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
        lo = 1.0 + torch.rand(1, 1, H // d, W // d, generator=g)
        outputs[k] = {"depth_values": torch.cat([lo, lo + 0.3, lo + 0.5 + torch.rand(1, 1, H // d, W // d, generator=g)], 1)}
    K = torch.tensor([[10., 0, 6], [0, 10, 4], [0, 0, 1]])
    c2w, w2c = rand_pose(g), rand_pose(g)
    near_fars = torch.tensor([[1.0, 4.0], [1.0, 4.0]])
    torch.manual_seed(11)
    t_rand = torch.rand(32, NS)
    torch.manual_seed(11)
    pts, rdir, ndc, zc, ro, _ = ref.utils.build_rays_test(H, W, c2w, w2c, K, near_fars, near_fars[-1], NS, chunk=32,
                                                          idx=1, outputs=outputs)
    save("g3_sampling", rays=rays, S=S, z_det=z0, pts_det=p0, z_lindisp=zl, noise=noise, z_p1=z1, pts_p1=p1, z_p05=z5,
         bt_H=H, bt_W=W, bt_NS=NS, bt_K=K, bt_c2w=c2w, bt_w2c=w2c, bt_near_fars=near_fars, bt_t_rand=t_rand,
         bt_dv1=outputs["stage1"]["depth_values"], bt_dv2=outputs["stage2"]["depth_values"],
         bt_dv3=outputs["stage3"]["depth_values"], bt_pts=pts, bt_dir=rdir, bt_z=zc, bt_o=ro,
         bt_ndc1=ndc["stage1"], bt_ndc2=ndc["stage2"], bt_ndc3=ndc["stage3"], bt_ndc=ndc["ndc"])


# ------------------------------------------------------------------ G4 projection to reference view
def g4():
    g = torch.Generator().manual_seed(104)
    N, S = 24, 10
    w2c = rand_pose(g)
    K = torch.tensor([[31., 0.3, 20.], [0, 29., 16.], [0, 0, 1]])
    pts = torch.randn(N, S, 3, generator=g) * 0.5 + torch.tensor([0., 0., 2.0])
    # rows that hit the |z| < 1e-4 clamp: put camera-space z at +-5e-5 / 0
    Rinv, t = w2c[:3, :3].T, w2c[:3, 3]
    for i, zc in enumerate((5e-5, -5e-5, 0.0, 9.9e-5)):
        pc = torch.tensor([0.3 * (i + 1), -0.2, zc])
        pts[0, i] = (pc - t) @ Rinv.T
    inv_scale = torch.tensor([39, 31])
    nf = {}
    for k in ("1", "2", "3"):
        nf["near_" + k] = 0.5 + torch.rand(N, S, 1, generator=g)
        nf["far_" + k] = 3.0 + torch.rand(N, S, 1, generator=g)
    nf["near"], nf["far"] = torch.tensor(0.8), torch.tensor(4.5)
    out = ref.utils.get_ndc_coordinate(w2c, K, pts.clone(), inv_scale, nf)
    q2d = ref.utils.get_ndc_coordinate(w2c, K, pts.clone(), inv_scale, sample_2d=True)
    save("g4_ndc_coord", w2c=w2c, K=K, pts=pts, inv_scale=inv_scale, q2d=q2d,
         **{k: v for k, v in nf.items()}, **{"out_" + k: v for k, v in out.items()})


# ------------------------------------------------------------------ G5 positional encodings
def g5():
    g = torch.Generator().manual_seed(105)
    x = torch.cat([torch.rand(40, 3, generator=g), (torch.rand(24, 3, generator=g) - 0.5) * 1200.0])
    e10, d10 = ref.models.get_embedder(10, 0)
    e4, d4 = ref.models.get_embedder(4, 0)
    h10, _ = ref.helpers.get_embedder(10, 0)
    h4, _ = ref.helpers.get_embedder(4, 0)
    x3 = x.reshape(8, 8, 3)
    save("g5_embed", x=x, live10=e10(x), live4=e4(x), live10_3d=e10(x3), inter10=h10(x), inter4=h4(x), d10=d10, d4=d4)


# ------------------------------------------------------------------ G6 the MLP
def make_net(V):
    """Seeded reference network; the V=7 one is shared by g6/g10/g11 (state_dict stored once, sd_v7.npz)."""
    torch.manual_seed(600 + V)
    return ref.models.UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=24 + (V - 1) * 12 + 1,
                             skips=[4], view_num=V)


def g6():
    save("sd_v7", **{k: v for k, v in make_net(7).state_dict().items()})
    for V, tag in ((7, "v7"), (4, "v4")):
        F = 24 + (V - 1) * 12 + 1
        net = make_net(V)
        g = torch.Generator().manual_seed(106 + V)
        M = 96
        x = torch.cat([torch.randn(M, 63, generator=g).clamp(-1, 1), torch.randn(M, F - 1, generator=g),
                       torch.rand(M, 1, generator=g), torch.randn(M, 27, generator=g).clamp(-1, 1)], -1)
        x.requires_grad_(True)
        r = torch.randn(M, 4, generator=g)
        out = net(x)
        (out * r).sum().backward()
        extra = {}
        if V == 7:   # parameter grads for the shared network; names of params the reference leaves without grad
            extra = {"grad." + k: p.grad for k, p in net.named_parameters() if p.grad is not None}
            extra["no_grad_names"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
        else:
            extra = {"sd." + k: v for k, v in net.state_dict().items()}
        save("g6_mlp_" + tag, V=V, x=x.detach(), r=r, out=out, grad_x=x.grad, **extra)


# ------------------------------------------------------------------ G7 feature gather
def tiny_scene(g, V=7, H=32, W=40, D=(6, 4, 2)):
    vols = [torch.randn(1, 8, D[0], H // 4, W // 4, generator=g), torch.randn(1, 8, D[1], H // 2, W // 2, generator=g),
            torch.randn(1, 8, D[2], H, W, generator=g)]
    imgs = torch.rand(1, V - 1, 3, H, W, generator=g)
    img_feat = torch.randn(V - 1, 1, 8, H, W, generator=g)
    conf = torch.rand(H, W, generator=g).clamp(1e-3, 1)
    K = torch.tensor([[30., 0, W / 2], [0, 30., H / 2], [0, 0, 1]])
    w2cs = torch.stack([rand_pose(g, 0.05) for _ in range(V)])
    return vols, imgs, img_feat, conf, K, w2cs


def g7():
    g = torch.Generator().manual_seed(107)
    V, H, W = 7, 32, 40
    vols, imgs, img_feat, conf, K, w2cs = tiny_scene(g, V, H, W)
    N, S = 20, 12
    pts = torch.randn(N, S, 3, generator=g) * torch.tensor([1.2, 1.0, 0.8]) + torch.tensor([0., 0., 2.2])
    pts[0, :3] = torch.tensor([[50., 0, 1.0], [0, -50., 1.0], [0, 0, -1.0]])     # far outside / behind
    inv_scale = torch.tensor([W - 1, H - 1])
    nf = {}
    for k in ("1", "2", "3"):
        nf["near_" + k] = 1.0 + 0.5 * torch.rand(N, S, 1, generator=g)
        nf["far_" + k] = 3.0 + 0.5 * torch.rand(N, S, 1, generator=g)
    nf["near"], nf["far"] = torch.tensor(1.0), torch.tensor(4.0)
    ndc = ref.utils.get_ndc_coordinate(w2cs[0], K, pts.clone(), inv_scale, nf)
    # exact pixel centres / image edges in the stage grids
    ndc["stage3"][1, 0] = torch.tensor([0.0, 0.0, 0.0])
    ndc["stage3"][1, 1] = torch.tensor([1.0, 1.0, 1.0])
    ndc["stage1"][1, 2] = torch.tensor([0.5, 0.5, 0.5])
    for v in vols + [img_feat, conf]:
        v.requires_grad_(True)
    volume_feature = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
    pose = {"w2cs": w2cs[1:].clone(), "intrinsics": K.repeat(V - 1, 1, 1)}
    rf, cs = ref.utils.index_point_feature(volume_feature, conf.reshape(1, 1, 1, H, W), ndc)
    cv = ref.utils.build_color_volume(pts, pose, imgs, None, img_feat, with_mask=True)
    feats = ref.renderer.gen_pts_feats(imgs, volume_feature, pts, pose, ndc, 24 + 12 * (V - 1) + 1, img_feat,
                                       confidence=conf)
    r = torch.randn(feats.shape, generator=g)
    (feats * r).sum().backward()
    save("g7_gather", V=V, H=H, W=W, K=K, w2cs=w2cs, pts=pts, vol1=vols[0], vol2=vols[1], vol3=vols[2], imgs=imgs,
         img_feat=img_feat, conf=conf, ndc1=ndc["stage1"], ndc2=ndc["stage2"], ndc3=ndc["stage3"], ndc=ndc["ndc"],
         ray_feats=rf, conf_sampled=cs, color_volume=cv, feats=feats, r=r, g_vol1=vols[0].grad, g_vol2=vols[1].grad,
         g_vol3=vols[2].grad, g_img_feat=img_feat.grad, g_conf=conf.grad)


# ------------------------------------------------------------------ G8 sample_pdf
def g8():
    g = torch.Generator().manual_seed(108)
    N, L, M = 160, 63, 128
    z = torch.sort(1.0 + 3.0 * torch.rand(N, L + 1, generator=g), -1)[0]
    bins = .5 * (z[:, 1:] + z[:, :-1])
    w = torch.rand(N, L - 1, generator=g) ** 4
    w[0] = 0.0                                  # all-zero weights
    w[1] = 0.0; w[1, 17] = 1.0                  # one-hot
    w[2] = 1e-9 * torch.rand(L - 1, generator=g)  # tiny totals
    w[3] = 0.0; w[3, 0] = 5.0; w[3, -1] = 5.0   # mass at both ends
    w[4] = 1e4 * torch.rand(L - 1, generator=g)
    det = ref.ray_utils.sample_pdf(bins, w, M, det=True)
    det_h = ref.helpers.sample_pdf(bins, w, M, det=True)
    torch.manual_seed(21)
    u = torch.rand(N, M)
    # u exactly on cdf edges for some rows: recompute cdf the way the reference does
    ww = w + 1e-5
    cdf = torch.cat([torch.zeros(N, 1), torch.cumsum(ww / torch.sum(ww, -1, keepdim=True), -1)], -1)
    u[5:40, :L] = cdf[5:40]
    u[5:40, L:L + 8] = torch.tensor([0.0, 1.0, 0.99999994, 1e-8, 0.5, 0.25, 0.75, 1.0])
    # monkeypatch torch.rand inside the reference call to return our u (captures the random branch)
    orig = torch.rand
    torch.rand = lambda *a, **k: u.clone()
    try:
        rnd = ref.ray_utils.sample_pdf(bins, w, M, det=False)
    finally:
        torch.rand = orig
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    inds_det = torch.searchsorted(cdf, torch.linspace(0., 1., M).expand(N, M).contiguous(), right=True)
    pyt = ref.ray_utils.sample_pdf(bins[:16], w[:16], M, det=False, pytest=True)
    np.random.seed(0)
    u_py = np.random.rand(16, M)
    save("g8_sample_pdf", bins=bins, weights=w, M=M, samples_det=det, samples_det_helpers=det_h, u=u, samples_u=rnd,
         cdf=cdf, inds_u=inds, inds_det=inds_det, samples_pytest=pyt, u_pytest=torch.Tensor(u_py))
    # the ragged-length rowsum pin: sums of rows of every length 1..70
    rows = {}
    gg = torch.Generator().manual_seed(1108)
    for n in list(range(1, 71)) + [127, 128, 190, 191]:
        x = torch.rand(12, n, generator=gg) ** 3
        rows["x%d" % n] = x
        rows["s%d" % n] = torch.sum(x, -1)
        rows["c%d" % n] = torch.cumsum(x, -1)
    save("g8_rowsum", **rows)


# ------------------------------------------------------------------ G9 compositing
def g9():
    g = torch.Generator().manual_seed(109)
    out = {}
    for S in (64, 192, 90):
        N = 40
        raw = torch.cat([torch.rand(N, S, 3, generator=g), torch.relu(torch.randn(N, S, 1, generator=g)) * 2], -1)
        raw[0, :, 3] = 0.0                       # empty ray
        raw[1, :, 3] = 50.0                      # saturated from the first sample
        raw[2, :, 3] = 0.0; raw[2, S // 2, 3] = 30.0
        z = torch.sort(1.0 + 3 * torch.rand(N, S, generator=g), -1)[0]
        raw.requires_grad_(True)
        for wb in (False, True):
            res = ref.renderer.raw2outputs(raw, z, None, wb, "v2")
            tag = "S%d_wb%d_" % (S, wb)
            for name, t in zip(("rgb", "disp", "acc", "weights", "depth", "alpha", "var"), res):
                out[tag + name] = t
        r3, r1 = torch.randn(N, 3, generator=g), torch.randn(N, generator=g)
        res = ref.renderer.raw2outputs(raw, z, None, False, "v2")
        (res[0] * r3).sum().add((res[4] * r1).sum()).backward()
        out.update({"S%d_raw" % S: raw.detach(), "S%d_z" % S: z, "S%d_r3" % S: r3, "S%d_r1" % S: r1,
                    "S%d_graw" % S: raw.grad})
        # nerf-pytorch style variant
        rays_d = torch.randn(N, 3, generator=g)
        rawh = torch.randn(N, S, 4, generator=g)
        resh = ref.helpers.raw2outputs(rawh, z, rays_d, 0, False)
        resw = ref.helpers.raw2outputs(rawh, z, rays_d, 0, True)
        for name, t in zip(("rgb", "disp", "acc", "weights", "depth"), resh):
            out["S%d_h_" % S + name] = t
        out.update({"S%d_h_raw" % S: rawh, "S%d_h_rays_d" % S: rays_d, "S%d_h_rgb_wb" % S: resw[0]})
        cosang = rays_d.norm(dim=-1)
        out["S%d_dists" % S] = ref.renderer.depth2dist(z, cosang)
    save("g9_composite", **out)


# ------------------------------------------------------------------ G10 rendering(), first and second call
class Args:
    pass


def mk_args(V):
    a = Args()
    a.view_num, a.feat_dim, a.img_downscale, a.use_color_volume, a.net_type = V, 24 + 12 * (V - 1) + 1, 1.0, False, "v2"
    a.netchunk, a.multires, a.multires_views, a.i_embed = 1024, 10, 4, 0
    return a


def g10():
    g = torch.Generator().manual_seed(110)
    V, H, W = 7, 32, 40
    vols, imgs, img_feat, conf, K, w2cs = tiny_scene(g, V, H, W)
    net = make_net(V)
    e_p, _ = ref.models.get_embedder(10, 0)
    e_d, _ = ref.models.get_embedder(4, 0)
    qfn = lambda pts, vd, f, fn: ref.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d, netchunk=8)
    N, S = 24, 12
    c2w = rand_pose(g, 0.05)
    xs, ys = torch.randint(0, W, (N,), generator=g).float(), torch.randint(0, H, (N,), generator=g).float()
    dirs = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], -1)
    rays_d = dirs @ c2w[:3, :3].T
    z = torch.sort(1.0 + 3.0 * torch.rand(N, S, generator=g), -1)[0]
    pts = c2w[:3, 3].reshape(1, 1, 3) + rays_d[:, None] * z[..., None]
    inv_scale = torch.tensor([W - 1, H - 1])
    nf = {}
    for k in ("1", "2", "3"):
        nf["near_" + k] = torch.full((N, S, 1), 1.0)
        nf["far_" + k] = torch.full((N, S, 1), 4.0)
    nf["near"], nf["far"] = torch.tensor(1.0), torch.tensor(4.0)
    ndc = ref.utils.get_ndc_coordinate(w2cs[0], K, pts.clone(), inv_scale, nf)
    volume_feature = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
    pose = {"w2cs": w2cs.clone(), "intrinsics": K.repeat(V, 1, 1)}
    a = mk_args(V)
    with torch.no_grad():
        rgb1, d1 = ref.renderer.rendering(a, pose, pts, ndc, z, rays_d, volume_feature, imgs, network_fn=net,
                                          img_feat=img_feat, network_query_fn=qfn, confidence=conf)
        n_after = pose["w2cs"].shape[0]
        rgb2, d2 = ref.renderer.rendering(a, pose, pts, ndc, z, rays_d, volume_feature, imgs, network_fn=net,
                                          img_feat=img_feat, network_query_fn=qfn, confidence=conf)
        rgbw, _ = ref.renderer.rendering(a, {"w2cs": w2cs.clone(), "intrinsics": K.repeat(V, 1, 1)}, pts, ndc, z,
                                         rays_d, volume_feature, imgs, network_fn=net, img_feat=img_feat,
                                         network_query_fn=qfn, white_bkgd=True, confidence=conf)
    save("g10_rendering", V=V, H=H, W=W, K=K, w2cs=w2cs, c2w=c2w, xs=xs, ys=ys, z=z, pts=pts, rays_d=rays_d,
         vol1=vols[0], vol2=vols[1], vol3=vols[2], imgs=imgs, img_feat=img_feat, conf=conf,
         ndc1=ndc["stage1"], ndc2=ndc["stage2"], ndc3=ndc["stage3"], ndc=ndc["ndc"], rgb_first=rgb1, depth_first=d1,
         rgb_second=rgb2, depth_second=d2, rgb_white=rgbw, n_w2cs_after_first=n_after)


# ------------------------------------------------------------------ G11 coarse -> fine composition
def g11():
    g = torch.Generator().manual_seed(111)
    V, H, W = 7, 32, 40
    vols, imgs, img_feat, conf, K, _ = tiny_scene(g, V, H, W)
    w2cs = torch.eye(4).repeat(V, 1, 1)
    w2cs[:, 0, 3] = 0.01 * torch.arange(V)
    net = make_net(V)
    # scale sigma head so that weights are non-trivial along the ray
    with torch.no_grad():
        net.nerf.alpha_linear.weight.mul_(0.05); net.nerf.alpha_linear_1.weight.mul_(0.05)
        net.nerf.alpha_linear.bias.add_(0.05); net.nerf.alpha_linear_1.bias.add_(0.05)
    e_p, _ = ref.models.get_embedder(10, 0)
    e_d, _ = ref.models.get_embedder(4, 0)
    qfn = lambda pts, vd, f, fn: ref.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d, netchunk=64)
    N, NC, NF = 128, 64, 128
    c2w = torch.eye(4)
    xs, ys = torch.randint(0, W, (N,), generator=g).float(), torch.randint(0, H, (N,), generator=g).float()
    ro, rd, _ = None, None, None
    dirs = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], -1)
    rd = dirs @ c2w[:3, :3].T
    rays = torch.cat([c2w[:3, 3].expand(N, 3), rd, torch.full((N, 1), 1.0), torch.full((N, 1), 4.0)], -1)
    volume_feature = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
    inv_scale = torch.tensor([W - 1, H - 1])
    a = mk_args(V)

    def one_pass(z):
        S = z.shape[1]
        pts = rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]
        nf = {}
        for k in ("1", "2", "3"):
            nf["near_" + k] = torch.full((N, S, 1), 1.0); nf["far_" + k] = torch.full((N, S, 1), 4.0)
        nf["near"], nf["far"] = torch.tensor(1.0), torch.tensor(4.0)
        ndc = ref.utils.get_ndc_coordinate(w2cs[0], K, pts.clone(), inv_scale, nf)
        pose = {"w2cs": w2cs.clone(), "intrinsics": K.repeat(V, 1, 1)}
        # rendering() returns only (rgb, depth); recompose its body to also get weights/var
        cos_angle = torch.norm(rd, dim=-1)
        angle = ref.renderer.gen_dir_feature(pose["w2cs"][0], rd / cos_angle.unsqueeze(-1))
        pose["w2cs"], pose["intrinsics"] = pose["w2cs"][1:], pose["intrinsics"][1:]
        feats = ref.renderer.gen_pts_feats(imgs, volume_feature, pts, pose, ndc, a.feat_dim, img_feat, confidence=conf)
        raw = qfn(ndc["ndc"], angle, feats, net)
        rgb_map, disp, acc, w, depth, _, var = ref.renderer.raw2outputs(raw, z, None, False, "v2")
        pose2 = {"w2cs": w2cs.clone(), "intrinsics": K.repeat(V, 1, 1)}
        rgb_chk, depth_chk = ref.renderer.rendering(a, pose2, pts, ndc, z, rd, volume_feature, imgs, network_fn=net,
                                                    img_feat=img_feat, network_query_fn=qfn, confidence=conf)
        assert torch.equal(rgb_chk, rgb_map) and torch.equal(depth_chk, depth)
        return rgb_map, depth, acc, w, var, raw

    with torch.no_grad():
        _, _, _, z_c = ref.ray_utils.ray_marcher(rays, NC, perturb=0)
        c_rgb, c_depth, c_acc, c_w, c_var, c_raw = one_pass(z_c)
        z_mid = .5 * (z_c[:, :-1] + z_c[:, 1:])
        z_s = ref.ray_utils.sample_pdf(z_mid, c_w[:, 1:-1], NF, det=True)
        z_f = torch.sort(torch.cat([z_s, z_c], -1), -1)[0]
        f_rgb, f_depth, f_acc, f_w, f_var, f_raw = one_pass(z_f)
    save("g11_coarse_fine", V=V, H=H, W=W, K=K, w2cs=w2cs, xs=xs, ys=ys, near=1.0, far=4.0, vol1=vols[0], vol2=vols[1],
         vol3=vols[2], imgs=imgs, img_feat=img_feat, conf=conf, z_coarse=z_c, c_rgb=c_rgb, c_depth=c_depth,
         c_weights=c_w, z_samples=z_s, z_fine=z_f, f_rgb=f_rgb, f_depth=f_depth, f_acc=f_acc, f_var=f_var,
         f_weights=f_w, c_sigma=c_raw[..., 3], sigma_head_scale=0.05, sigma_head_bias=0.05)


# ------------------------------------------------------------------ G16 rendering() in the Hamlyn configuration (V = 4)
def recompose_rendering(a, pose, pts, ndc, z, rays_d, volume_feature, imgs, net, img_feat, conf, qfn):
    """The body of rendering() (network/renderer.py:232-252) step by step, to capture what it discards: features,
    raw network output, weights, acc, var and the per-sample uncertainty u = 1 - confidence (network/models.py:149).
    Checked bit for bit against rendering() itself by the callers."""
    cos_angle = torch.norm(rays_d, dim=-1)
    angle = ref.renderer.gen_dir_feature(pose["w2cs"][0], rays_d / cos_angle.unsqueeze(-1))
    src = {"w2cs": pose["w2cs"][1:], "intrinsics": pose["intrinsics"][1:]}
    feats = ref.renderer.gen_pts_feats(imgs, volume_feature, pts, src, ndc, a.feat_dim, img_feat, confidence=conf)
    raw = qfn(ndc["ndc"], angle, feats, net)
    rgb_map, disp, acc, w, depth, _, var = ref.renderer.raw2outputs(raw, z, None, False, "v2")
    u = 1 - feats[..., -1]
    return dict(feats=feats, raw=raw, rgb=rgb_map, depth=depth, acc=acc, weights=w, var=var, u=u, wu=(w * u).sum(-1))


def g16():
    """configs[3] (Hamlyn, 3 source views -> view_num 4, feat_dim 61, data/hamlyn.py:208): rendering() first and second
    call, everything its body computes, and the gradients autograd sends into the network and the gather sources.  The
    stage coordinates carry per-ray cascade ranges (as build_rays produces them), the target camera differs from the
    reference view."""
    g = torch.Generator().manual_seed(116)
    V, H, W = 4, 32, 40
    vols, imgs, img_feat, conf, K, w2cs = tiny_scene(g, V, H, W)
    net = make_net(V)
    with torch.no_grad():      # non-trivial weights along the ray
        net.nerf.alpha_linear.weight.mul_(0.1); net.nerf.alpha_linear_1.weight.mul_(0.1)
        net.nerf.alpha_linear.bias.add_(0.05); net.nerf.alpha_linear_1.bias.add_(0.05)
    e_p, _ = ref.models.get_embedder(10, 0)
    e_d, _ = ref.models.get_embedder(4, 0)
    qfn = lambda pts, vd, f, fn: ref.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d, netchunk=16)
    N, S = 40, 18
    c2w = rand_pose(g, 0.05)
    xs, ys = torch.randint(0, W, (N,), generator=g).float(), torch.randint(0, H, (N,), generator=g).float()
    dirs = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], -1)
    rays_d = dirs @ c2w[:3, :3].T
    z = torch.sort(1.0 + 3.0 * torch.rand(N, S, generator=g), -1)[0]
    pts = c2w[:3, 3].reshape(1, 1, 3) + rays_d[:, None] * z[..., None]
    inv_scale = torch.tensor([W - 1, H - 1])
    lo = 0.8 + 0.4 * torch.rand(N, 3, generator=g)
    ranges = torch.stack([lo[:, 0], lo[:, 0] + 3.0, lo[:, 1], lo[:, 1] + 2.5, lo[:, 2], lo[:, 2] + 3.5], -1)     # [N,6]
    nf = {}
    for i, k in enumerate(("near_1", "far_1", "near_2", "far_2", "near_3", "far_3")):
        nf[k] = ranges[:, i].reshape(N, 1, 1).expand(N, S, 1)
    nf["near"], nf["far"] = torch.tensor(1.0), torch.tensor(4.0)
    ndc = ref.utils.get_ndc_coordinate(w2cs[0], K, pts.clone(), inv_scale, nf)
    for v in vols + [img_feat, conf]:
        v.requires_grad_(True)
    volume_feature = {"stage%d" % (i + 1): {"volume_feature_no_ref": vols[i]} for i in range(3)}
    a = mk_args(V)
    pose = {"w2cs": w2cs.clone(), "intrinsics": K.repeat(V, 1, 1)}
    rgb1, d1 = ref.renderer.rendering(a, pose, pts, ndc, z, rays_d, volume_feature, imgs, network_fn=net, img_feat=img_feat,
                                      network_query_fn=qfn, confidence=conf)
    n_after = pose["w2cs"].shape[0]
    r3, r1 = torch.randn(N, 3, generator=g), torch.randn(N, generator=g)
    ((rgb1 * r3).sum() + (d1 * r1).sum()).backward()
    grads = {"g_vol1": vols[0].grad, "g_vol2": vols[1].grad, "g_vol3": vols[2].grad, "g_img_feat": img_feat.grad, "g_conf": conf.grad}
    grads.update({"grad." + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
    with torch.no_grad():
        rgb2, d2 = ref.renderer.rendering(a, pose, pts, ndc, z, rays_d, volume_feature, imgs, network_fn=net, img_feat=img_feat,
                                          network_query_fn=qfn, confidence=conf)
        full = recompose_rendering(a, {"w2cs": w2cs.clone(), "intrinsics": K.repeat(V, 1, 1)}, pts, ndc, z, rays_d, volume_feature,
                                   imgs, net, img_feat, conf, qfn)
    assert torch.equal(full["rgb"], rgb1.detach()) and torch.equal(full["depth"], d1.detach())
    save("g16_rendering_v4", V=V, H=H, W=W, K=K, w2cs=w2cs, c2w=c2w, xs=xs, ys=ys, z=z, pts=pts, rays_d=rays_d, ranges=ranges,
         vol1=vols[0], vol2=vols[1], vol3=vols[2], imgs=imgs, img_feat=img_feat, conf=conf,
         ndc1=ndc["stage1"], ndc2=ndc["stage2"], ndc3=ndc["stage3"], ndc=ndc["ndc"], rgb_first=rgb1, depth_first=d1,
         rgb_second=rgb2, depth_second=d2, n_w2cs_after_first=n_after, r3=r3, r1=r1,
         feats=full["feats"], raw=full["raw"], acc=full["acc"], weights=full["weights"], var=full["var"], u_sampled=full["u"],
         wu=full["wu"], sigma_head_scale=0.1, sigma_head_bias=0.05, **grads)     # network = g6_mlp_v4's "sd." entries, sigma heads rescaled


# ------------------------------------------------------------------ G17 the opt-in uncertainty outputs on the G10 / G11 scenes
def g17():
    """What rendering() computes and discards on the (already committed) G10 and G11 inputs: per-sample uncertainty
    u = 1 - sampled confidence (network/models.py:149), weights / acc / var (network/renderer.py:129) and the composite
    sum_i w_i u_i.  The G10 / G11 fixtures themselves are read, not regenerated; the recomposed body is checked bit for bit
    against their stored renders."""
    def load(name):
        with np.load(os.path.join(HERE, name + ".npz")) as f:
            return {k: (torch.from_numpy(f[k].copy()) if f[k].ndim else f[k].item()) for k in f.files if f[k].dtype.kind not in "US"}
    sd = load("sd_v7")
    e_p, _ = ref.models.get_embedder(10, 0)
    e_d, _ = ref.models.get_embedder(4, 0)
    out = {}
    with torch.no_grad():
        g = load("g10_rendering")
        V = g["V"]
        net = make_net(V)
        net.load_state_dict(sd)
        qfn = lambda pts, vd, f, fn: ref.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d, netchunk=8)
        vf = {"stage%d" % k: {"volume_feature_no_ref": g["vol%d" % k]} for k in (1, 2, 3)}
        ndc = {"stage1": g["ndc1"], "stage2": g["ndc2"], "stage3": g["ndc3"], "ndc": g["ndc"]}
        full = recompose_rendering(mk_args(V), {"w2cs": g["w2cs"].clone(), "intrinsics": g["K"].repeat(V, 1, 1)}, g["pts"], ndc, g["z"],
                                   g["rays_d"], vf, g["imgs"], net, g["img_feat"], g["conf"], qfn)
        assert torch.equal(full["rgb"], g["rgb_first"]) and torch.equal(full["depth"], g["depth_first"])
        out.update({"g10_" + k: full[k] for k in ("u", "wu", "weights", "acc", "var")})

        g = load("g11_coarse_fine")
        V, H, W = g["V"], g["H"], g["W"]
        with torch.no_grad():
            net.nerf.alpha_linear.weight.mul_(g["sigma_head_scale"]); net.nerf.alpha_linear_1.weight.mul_(g["sigma_head_scale"])
            net.nerf.alpha_linear.bias.add_(g["sigma_head_bias"]); net.nerf.alpha_linear_1.bias.add_(g["sigma_head_bias"])
        qfn = lambda pts, vd, f, fn: ref.renderer.run_network_mvs(pts, vd, f, fn, embed_fn=e_p, embeddirs_fn=e_d, netchunk=64)
        vf = {"stage%d" % k: {"volume_feature_no_ref": g["vol%d" % k]} for k in (1, 2, 3)}
        K = g["K"]
        dirs = torch.stack([(g["xs"] - K[0, 2]) / K[0, 0], (g["ys"] - K[1, 2]) / K[1, 1], torch.ones_like(g["xs"])], -1)
        rd = dirs @ torch.eye(3)
        for tag, z in (("c", g["z_coarse"]), ("f", g["z_fine"])):
            N, S = z.shape
            pts = rd[:, None] * z[..., None]
            nf = {}
            for k in ("1", "2", "3"):
                nf["near_" + k] = torch.full((N, S, 1), 1.0); nf["far_" + k] = torch.full((N, S, 1), 4.0)
            nf["near"], nf["far"] = torch.tensor(1.0), torch.tensor(4.0)
            ndc = ref.utils.get_ndc_coordinate(g["w2cs"][0], K, pts.clone(), torch.tensor([W - 1, H - 1]), nf)
            full = recompose_rendering(mk_args(V), {"w2cs": g["w2cs"].clone(), "intrinsics": K.repeat(V, 1, 1)}, pts, ndc, z, rd, vf,
                                       g["imgs"], net, g["img_feat"], g["conf"], qfn)
            assert torch.equal(full["rgb"], g[tag + "_rgb"]) and torch.equal(full["weights"], g[tag + "_weights"])
            out.update({"g11_%s_%s" % (tag, k): full[k] for k in ("u", "wu")})
    save("g17_uncertainty", **out)


# ------------------------------------------------------------------ G12 / G13 cost volume + depth regression (row f2)
def mvs_setup(g, V, C, H, W, D, pad):
    """Source feature maps, (src_proj @ ref_proj_inv)[:3] per source view, per-pixel depth hypotheses."""
    K = torch.tensor([[0.9 * W, 0, 0.5 * W - 0.3], [0, 0.95 * W, 0.5 * H + 0.2], [0, 0, 1]])
    K4 = torch.eye(4); K4[:3, :3] = K
    w2c = [rand_pose(g, 0.05) for _ in range(V + 1)]                     # [0] = the target ("reference") view
    affine = torch.stack([K4 @ m for m in w2c])                          # data/scared.py builds K @ w2c per view
    affine_inv = torch.inverse(affine)
    feats = [torch.randn(1, C, H, W, generator=g) for _ in range(V)]
    imgs = torch.rand(1, V, 3, 2 * H, 2 * W, generator=g)
    Hp, Wp = H + 2 * pad, W + 2 * pad
    base = torch.linspace(1.2, 3.4, D).view(1, D, 1, 1)
    depth_values = (base + 0.15 * torch.rand(1, D, H, W, generator=g)).contiguous()   # padded inside DepthNet (replicate)
    return feats, imgs, affine, affine_inv, depth_values


def g12_g13():
    out12, out13 = {}, {}
    for tag, (V, C, H, W, D, pad) in {"a": (3, 8, 16, 20, 6, 0), "b": (6, 16, 12, 14, 5, 2)}.items():
        g = torch.Generator().manual_seed(1200 + len(tag) + V)
        feats, imgs, affine, affine_inv, depth_values = mvs_setup(g, V, C, H, W, D, pad)
        captured = {}
        for f_ in feats:
            f_.requires_grad_(True)

        def cost_reg(v, captured=captured, g=g):
            # stands where the 3D regularisation CNN is passed in (an argument of DepthNet.forward): records the variance
            # volume it is given and returns seeded (features, logits); the logits are a leaf so that their gradient --
            # what the real network would receive -- can be captured
            captured["v"] = v
            captured["variance"] = v.detach().clone()
            prob = (torch.randn(v.shape[0], 1, *v.shape[2:], generator=g) * 2.0).requires_grad_(True)
            captured["prob"] = prob
            captured["prob_pre"] = prob.detach().squeeze(1).clone()
            return v[:, :8] * 1.0, prob * 1.0          # (non-leaf: DepthNet adds prob_volume_init in place)

        prob_init = torch.randn(1, D, H + 2 * pad, W + 2 * pad, generator=g) * 0.5 if tag == "b" else None
        net = ref.mvs.DepthNet()
        o = net(feats, affine, affine_inv, depth_values, D, cost_reg, imgs, pad=pad,
                prob_volume_init=None if prob_init is None else prob_init.clone())
        # gradients the reference's autograd sends back: into the feature maps through the variance volume, into the
        # logits through depth and photometric confidence
        r_var = torch.randn(captured["v"].shape, generator=g)
        r_depth = torch.randn(o["depth"].shape, generator=g)
        r_conf = torch.randn(o["photometric_confidence"].shape, generator=g)
        ((captured["v"] * r_var).sum() + (o["depth"] * r_depth).sum() + (o["photometric_confidence"] * r_conf).sum()).backward()
        g_feats = torch.cat([f_.grad for f_ in feats])
        g_prob = captured["prob"].grad[0, 0]
        feats = [f_.detach() for f_ in feats]
        o = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in o.items()}
        # the same warps through homo_warp directly (the function the kernel replaces)
        dv_pad = torch.nn.functional.pad(depth_values, (pad, pad, pad, pad), "replicate") if pad > 0 else depth_values
        proj = torch.stack([(affine[i + 1:i + 2] @ affine_inv[0:1])[0, :3] for i in range(V)])
        warped0, grid0 = ref.utils.homo_warp(feats[0], proj[0:1], dv_pad, pad=pad)
        out12.update({tag + "_" + k: v for k, v in dict(
            V=V, C=C, H=H, W=W, D=D, pad=pad, feats=torch.cat(feats), proj=proj, depth_values=dv_pad[0],
            variance=captured["variance"][0], warped0=warped0[0], grid0=grid0.reshape(D, -1, 2), r_var=r_var[0],
            g_feats=g_feats).items()})
        out13.update({tag + "_" + k: v for k, v in dict(
            D=D, pad=pad, prob_pre=captured["prob_pre"][0], depth_values=o["depth_values"][0],
            prob_init=torch.zeros(0) if prob_init is None else prob_init[0], prob_volume=o["prob_volume"][0],
            depth=o["depth"][0], confidence=o["photometric_confidence"][0], r_depth=r_depth[0], r_conf=r_conf[0],
            g_prob_pre=g_prob).items()})
    save("g12_cost_volume", **out12)
    save("g13_depth_regress", **out13)


# ------------------------------------------------------------------ G14 on-disk formats (row f4)
def g14():
    """A tiny COLMAP model written with the reference's own writer (colmapUtils/read_write_model.py), an LLFF
    poses_bounds array, and what the reference's readers / dataset code make of them."""
    import shutil
    import tempfile
    import types
    import colmapUtils.read_write_model as rw
    if "imageio" not in sys.modules:
        try:
            import imageio  # noqa: F401
        except ImportError:
            sys.modules["imageio"] = types.ModuleType("imageio")          # only imported, never called here
    if "data" not in sys.modules:                                          # data/__init__.py pulls in every dataset
        pkg = types.ModuleType("data"); pkg.__path__ = [os.path.join(_ref_import.REF, "data")]; sys.modules["data"] = pkg
    import data.scared as sc

    rng = np.random.RandomState(14)
    n_img, n_pts, W, H = 5, 60, 64, 48
    cams = {1: rw.Camera(id=1, model="PINHOLE", width=W, height=H, params=np.array([50.0, 52.0, 31.5, 24.25]))}
    pts_xyz = np.column_stack([rng.uniform(-1, 1, n_pts), rng.uniform(-0.8, 0.8, n_pts), rng.uniform(2.0, 6.0, n_pts)])
    images, tracks = {}, {k + 1: [] for k in range(n_pts)}
    names = ["frame_%03d.png" % k for k in (3, 0, 4, 1, 2)]                # file order != name order
    for i in range(n_img):
        ang = 0.05 * (i - 2)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([0.1 * (i - 2), 0.02 * i, 0.0])
        cam = pts_xyz @ R.T + t
        xy = np.column_stack([50.0 * cam[:, 0] / cam[:, 2] + 31.5, 52.0 * cam[:, 1] / cam[:, 2] + 24.25])
        sel = rng.permutation(n_pts)[:40]
        ids = np.where(rng.rand(40) < 0.15, -1, sel + 1).astype(np.int64)  # some untriangulated keypoints
        for j, pid in enumerate(ids):
            if pid != -1:
                tracks[int(pid)].append((i + 1, j))
        images[i + 1] = rw.Image(id=i + 1, qvec=rw.rotmat2qvec(R), tvec=t, camera_id=1, name=names[i], xys=xy[sel], point3D_ids=ids)
    points = {k: rw.Point3D(id=k, xyz=pts_xyz[k - 1], rgb=rng.randint(0, 255, 3), error=np.array(rng.uniform(0.2, 2.0)),
                            image_ids=np.array([a for a, _ in tracks[k]], dtype=np.int64),
                            point2D_idxs=np.array([b for _, b in tracks[k]], dtype=np.int64)) for k in tracks}
    out_dir = os.path.join(HERE, "colmap_tiny", "sparse", "0")
    shutil.rmtree(os.path.join(HERE, "colmap_tiny"), ignore_errors=True)
    os.makedirs(out_dir)
    rw.write_model(cams, images, points, out_dir, ext=".bin")

    # LLFF poses_bounds: [N, 3x5 (rotation | translation | H, W, focal) + near, far]
    pb = np.zeros((n_img, 17))
    for i in range(n_img):
        c2w = sc.ScaredDataset.get_poses(None, {1: images[i + 1]})[0]
        m = np.concatenate([c2w[:3, :4], np.array([[H], [W], [50.0]])], 1)
        pb[i, :15] = m.reshape(-1)
        pb[i, 15:] = [1.5 + 0.1 * i, 7.0 - 0.2 * i]
    np.save(os.path.join(HERE, "colmap_tiny", "poses_bounds.npy"), pb)

    # what the reference reads back
    r_cams = rw.read_cameras_binary(os.path.join(out_dir, "cameras.bin"))
    r_imgs = rw.read_images_binary(os.path.join(out_dir, "images.bin"))
    r_pts = rw.read_points3d_binary(os.path.join(out_dir, "points3D.bin"))
    arrs = {"cam_params": r_cams[1].params, "cam_wh": np.array([r_cams[1].width, r_cams[1].height])}
    for i in r_imgs:
        arrs["img%d_qvec" % i] = r_imgs[i].qvec; arrs["img%d_tvec" % i] = r_imgs[i].tvec
        arrs["img%d_xys" % i] = r_imgs[i].xys; arrs["img%d_ids" % i] = r_imgs[i].point3D_ids
        arrs["img%d_R" % i] = r_imgs[i].qvec2rotmat()
    arrs["img_names"] = np.array([r_imgs[i].name for i in r_imgs])
    arrs["pt_ids"] = np.array(sorted(r_pts))
    arrs["pt_xyz"] = np.stack([r_pts[k].xyz for k in sorted(r_pts)])
    arrs["pt_err"] = np.array([float(r_pts[k].error) for k in sorted(r_pts)])
    arrs["pt_rgb"] = np.stack([r_pts[k].rgb for k in sorted(r_pts)])
    arrs["pt7_track"] = np.stack([r_pts[7].image_ids, r_pts[7].point2D_idxs])

    # pose normalisation exactly as data/scared.py:222-244 applies it, with the reference's center_poses
    poses = pb[:, :15].reshape(-1, 3, 5)
    bounds = pb[:, -2:].copy()
    blender2opencv = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    p2 = np.concatenate([poses[..., 1:2], -poses[..., :1], poses[..., 2:4]], -1)
    scale = 1. / (bounds.min() * 0.75)
    bounds *= scale
    p2[..., 3] *= scale
    centred, pose_avg = sc.center_poses(p2, blender2opencv)
    arrs.update(poses_centred=centred, pose_avg=pose_avg, bounds_scaled=bounds, scale_factor=scale,
                nearest=sc.get_nearest_pose_ids(centred[2], centred[[0, 1, 3, 4]], 3))

    # sparse depth supervision (data/scared.py:285-344) through the reference's method on a stand-in `self`
    tmp = tempfile.mkdtemp()
    shutil.copytree(os.path.join(HERE, "colmap_tiny", "sparse"), os.path.join(tmp, "sparse"))
    stub = types.SimpleNamespace(img_wh=(32, 24))
    stub.get_poses = types.MethodType(sc.ScaredDataset.get_poses, stub)
    lst = sc.ScaredDataset.load_colmap_depth(stub, tmp, factor=2.0, bd_factor=.75, bds_raw=pb[:, -2:].transpose([1, 0]))
    shutil.rmtree(tmp)
    arrs["sd_names"] = np.array([d["name"] for d in lst])
    for k, d in enumerate(lst):
        arrs["sd%d_depth" % k] = d["depth"]; arrs["sd%d_coord" % k] = d["coord"]; arrs["sd%d_weight" % k] = d["weight"]
        arrs["sd%d_depth_img" % k] = d["depth_img"]; arrs["sd%d_weight_img" % k] = d["weight_img"]
    save("g14_formats", **arrs)


# ------------------------------------------------------------------ G15 training losses (row f3)
def g15():
    """The trainer's loss terms (train.py:164-188) from the reference's own loss classes on seeded inputs, with the
    gradients they send back into the rendered rgb / depth and the cascade depths."""
    import importlib
    L = importlib.import_module("utils.loss")
    g = torch.Generator().manual_seed(15)
    patch_num, ps, n_rays, n_depth = 6, 4, 6 * 16 + 40, 25
    N = n_rays + n_depth
    rgb = torch.rand(N, 3, generator=g).requires_grad_(True)
    depth_pred = (1.0 + 3.0 * torch.rand(N, generator=g)).requires_grad_(True)
    target_s = torch.rand(N, 3, generator=g)
    target_depths = 1.0 + 3.0 * torch.rand(n_depth, generator=g)
    target_weights = 2.0 * torch.rand(n_depth, generator=g)
    patch_dpt = torch.rand(patch_num, ps, ps, 1, generator=g)
    outputs, gt_ms, w_ms = {}, {}, {}
    for k, (h, w) in {"stage1": (6, 8), "stage2": (12, 16), "stage3": (24, 32)}.items():
        outputs[k] = {"depth": (1.0 + 3.0 * torch.rand(1, h, w, generator=g)).requires_grad_(True)}
        m = torch.rand(1, h, w, generator=g) < 0.3
        gt_ms[k] = torch.where(m, 1.0 + 3.0 * torch.rand(1, h, w, generator=g), torch.zeros(1, h, w))
        w_ms[k] = torch.where(m, 0.1 + 1.9 * torch.rand(1, h, w, generator=g), torch.zeros(1, h, w))
    outputs["depth"] = outputs["stage3"]["depth"]                       # CascadeMVSNet.forward also spreads the last stage
    smooth, edge = ref.mvs.EdgePreservingSmoothnessLoss(), L.GradientLoss()      # (train.py:12,17,42-43)
    patch_pts = patch_num * ps * ps
    loss_mvs, _ = ref.mvs.cas_mvsnet_loss(outputs, gt_ms, w_ms)
    patch_depth = depth_pred[:patch_pts].reshape(-1, ps, ps)
    smooth_loss = smooth(patch_depth[:patch_num // 2, ...], patch_dpt[:patch_num // 2, ...])
    loss_nerf_depth = torch.mean(((depth_pred[n_rays:] - target_depths) ** 2) * target_weights)
    loss_si = edge(patch_depth[patch_num // 2:, ...], patch_dpt[patch_num // 2:, ...].squeeze(-1),
                   torch.ones_like(patch_depth[patch_num // 2:, ...]))
    depth_loss = loss_nerf_depth * 0.05 + loss_mvs * 0.05 + smooth_loss * 0.05 + loss_si * 0.008
    img_loss = L.img2mse(rgb, target_s)
    loss = depth_loss + img_loss * 5.0
    loss.backward()
    arrs = dict(patch_num=patch_num, patch_size=ps, n_rays=n_rays, rgb=rgb, depth_pred=depth_pred, target_s=target_s,
                target_depths=target_depths, target_weights=target_weights, patch_dpt=patch_dpt, loss=loss, img_loss=img_loss,
                loss_mvs=loss_mvs, smooth_loss=smooth_loss, loss_nerf_depth=loss_nerf_depth, loss_scaleinvariant=loss_si,
                g_rgb=rgb.grad, g_depth=depth_pred.grad)
    for k in ("stage1", "stage2", "stage3"):
        arrs[k + "_depth"] = outputs[k]["depth"]; arrs[k + "_gt"] = gt_ms[k]; arrs[k + "_w"] = w_ms[k]
        arrs[k + "_g"] = outputs[k]["depth"].grad
    save("g15_losses", **arrs)


# ------------------------------------------------------------------ G18 the training ray builder (utils/utils.py:400-597)
def g18():
    """build_rays as train.py:147-163 calls it (with_depth=True), on a tiny scene, with every random draw RECORDED in call order -- torch.multinomial
    (the two confidence-weighted patch picks), np.random.randint (the patch shifts), torch.randint (the uniform pixels), torch.rand (the jitter) -- so that
    a replay of the same draws through another implementation must reproduce every output, whatever device its generator lives on."""
    import types
    g = torch.Generator().manual_seed(118)
    H, W, V, NS = 24, 32, 3, 12
    patch_num, ps, n_rays = 4, 4, 90
    args = types.SimpleNamespace(patch_num=patch_num, patch_size=ps)
    imgs = torch.rand(1, V, 3, H, W, generator=g)
    conf = torch.rand(H, W, generator=g).clamp(1e-3, 1 - 1e-3)
    outputs = {}
    for k, (d, D) in (("stage1", (4, 5)), ("stage2", (2, 4)), ("stage3", (1, 3))):
        lo = 1.0 + torch.rand(1, 1, H // d, W // d, generator=g)
        steps = torch.linspace(0, 1, D).view(1, D, 1, 1)
        hi = lo + 0.4 + torch.rand(1, 1, H // d, W // d, generator=g)
        outputs[k] = {"depth_values": (lo * (1 - steps) + hi * steps).contiguous(), "prob_volume": torch.softmax(torch.randn(1, D, H // d, W // d, generator=g), 1)}
    K = torch.tensor([[30., 0, 16], [0, 30, 12], [0, 0, 1]])
    c2ws = torch.stack([rand_pose(g) for _ in range(V)])
    w2cs = torch.inverse(c2ws)
    intrinsics = K.repeat(V, 1, 1)
    near_fars = torch.tensor([[1.0, 4.0]] * V)
    pose_ref = {"w2cs": w2cs.clone(), "intrinsics": intrinsics.clone(), "near_fars": near_fars}
    n_sparse = 7
    coords = torch.stack([torch.randint(0, H, (n_sparse,), generator=g), torch.randint(0, W, (n_sparse,), generator=g)], -1).float()     # (row, col)
    sparse = torch.zeros(H, W)
    sparse[coords[:, 0].long(), coords[:, 1].long()] = 1.0 + 2.0 * torch.rand(n_sparse, generator=g)
    draws = []
    orig = (torch.multinomial, np.random.randint, torch.randint, torch.rand)

    def rec(kind, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            draws.append((kind, torch.as_tensor(out).clone()))
            return out
        return wrapped
    torch.multinomial, np.random.randint, torch.randint, torch.rand = rec("multinomial", orig[0]), rec("np_randint", orig[1]), rec("randint", orig[2]), rec("rand", orig[3])
    # get_rays_with_random_patches (utils/utils.py:196-199) slices a transposed mesh grid, moves the slice `.to(device)` and calls `.view(-1)` on it: on the
    # GPU the reference runs on, the move is a copy and the copy is contiguous; on this container's CPU the move is the identity and `.view` raises.  The
    # capture therefore lets `.view` fall back to `.reshape` (the same values the GPU path sees) for the duration of the call.
    view = torch.Tensor.view

    def view_or_reshape(self, *shape, **kw):
        try:
            return view(self, *shape, **kw)
        except RuntimeError:
            return self.reshape(*shape)
    torch.Tensor.view = view_or_reshape
    try:
        torch.manual_seed(18)
        np.random.seed(18)
        out = ref.utils.build_rays(args, imgs, conf, sparse, coords, pose_ref, w2cs, c2ws, intrinsics, n_rays, NS, with_depth=True, outputs=outputs)
    finally:
        torch.multinomial, np.random.randint, torch.randint, torch.rand = orig
        torch.Tensor.view = view
    pts, rays_d, colors, ndc, z, rays_o, rays_depths, par, pix = out
    assert rays_depths is None
    arrs = dict(H=H, W=W, V=V, NS=NS, patch_num=patch_num, patch_size=ps, n_rays=n_rays, imgs=imgs, conf=conf, K=K, c2ws=c2ws, w2cs=w2cs, near_fars=near_fars,
                coords=coords, sparse=sparse, pts=pts, rays_d=rays_d, colors=colors, z=z, rays_o=rays_o, pix=pix,
                ndc1=ndc["stage1"], ndc2=ndc["stage2"], ndc3=ndc["stage3"], ndc=ndc["ndc"], n_draws=len(draws),
                draw_kinds=np.array([k for k, _ in draws]))
    for i, (_, v) in enumerate(draws):
        arrs["draw_%03d" % i] = v
    for k in ("stage1", "stage2", "stage3"):
        arrs[k + "_depth_values"] = outputs[k]["depth_values"]
        arrs[k + "_prob_volume"] = outputs[k]["prob_volume"]
    save("g18_build_rays", **arrs)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12_g13, g14, g15, g16, g17, g18):
        if not only or fn.__name__ in only:
            fn()
