"""CPU oracle for the UC-NeRF ray-marching volume-render hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product package (``uc_nerf_amd``)
never imports it and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors captured in the build container by importing the reference's own Python
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).  The reference ships no tests or
fixtures of its own, so those captured vectors are the pin.

This is a restatement written from the reference's behaviour, in plain torch-CPU / numpy, with
explicit index arithmetic where the reference leans on library ops (``F.grid_sample``,
``searchsorted``, ``cumprod``) so that it documents exactly what the HIP kernels have to compute.
All functions are dtype-generic: feed float64 tensors to get a high-precision yardstick.

Reference locations restated (paths relative to the reference tree):
  a1  data/ray_utils.py:12-53, utils/run_nerf_helpers.py:248-257, utils/utils.py:217-271
  a2  data/ray_utils.py:56-94, utils/run_nerf_helpers.py:277-294
  a3  data/ray_utils.py:152-197, utils/utils.py:393-397 and :698-717
  a4  utils/utils.py:323-373
  a5  network/models.py:20-71, utils/run_nerf_helpers.py:23-71
  a6  network/models.py:74-207, network/renderer.py:57-106
  a7  network/renderer.py:163-212, utils/utils.py:742-799 and :833-893
  a8  data/ray_utils.py:98-141 == utils/run_nerf_helpers.py:298-341
  a9  network/renderer.py:6-36 and :109-140, utils/run_nerf_helpers.py:343-390
  a10 network/renderer.py:215-255, data/ray_utils.py:199-224 (coarse->fine recipe)
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------------------------
# a1  ray generation
# --------------------------------------------------------------------------------------------


def pixel_grid(H, W, dtype=torch.float32):
    """Row-major (ij) pixel grid flattened: returns ys[H*W], xs[H*W] (data/ray_utils.py:22-23)."""
    ys = torch.arange(H, dtype=dtype).repeat_interleave(W)
    xs = torch.arange(W, dtype=dtype).repeat(H)
    return ys, xs


def camera_dirs(xs, ys, K):
    """d = ((x-cx)/fx, (y-cy)/fy, 1) -- no half-pixel offset (data/ray_utils.py:27, utils/utils.py:259-261)."""
    return torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], -1)


def get_ray_directions(H, W, K):
    ys, xs = pixel_grid(H, W, K.dtype)
    return camera_dirs(xs, ys, K)


def get_rays(directions, c2w):
    """rays_d = dirs @ R^T (not normalised), rays_o = t broadcast (data/ray_utils.py:45-51)."""
    rays_d = (directions.reshape(-1, 3) @ c2w[:3, :3].T)
    rays_o = c2w[:3, 3].expand_as(rays_d)
    return rays_o, rays_d


def get_rays_mvs_grid(H, W, K, c2w, chunk=-1, idx=-1):
    """Deterministic branch of get_rays_mvs (utils/utils.py:248-271): returns rays_o[3], rays_d[N,3],
    pixel_coordinates[2,N] as (row, col)."""
    ys, xs = pixel_grid(H, W, c2w.dtype)
    if chunk > 0:
        ys, xs = ys[idx * chunk:(idx + 1) * chunk], xs[idx * chunk:(idx + 1) * chunk]
    return get_rays_mvs_pixels(xs, ys, K, c2w)


def get_rays_mvs_pixels(xs, ys, K, c2w):
    """get_rays_mvs on given pixel coordinates (the random branches differ only in how xs, ys are drawn)."""
    rays_d = camera_dirs(xs, ys, K) @ c2w[:3, :3].T
    rays_o = c2w[:3, -1].clone()
    return rays_o, rays_d, torch.stack((ys, xs))


def get_rays_opengl(H, W, focal, c2w):
    """utils/run_nerf_helpers.py:248-257: single focal, image-centre principal point, -z forward, y up.
    Returns [H,W,3] tensors."""
    j, i = torch.meshgrid(torch.arange(H, dtype=c2w.dtype), torch.arange(W, dtype=c2w.dtype), indexing="ij")
    dirs = torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -torch.ones_like(i)], -1)
    rays_d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


# --------------------------------------------------------------------------------------------
# a2  NDC ray warp
# --------------------------------------------------------------------------------------------


def get_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """data/ray_utils.py:56-94 (focal = [fx, fy]; d2 = 1 - o2)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    ox, oy = o[..., 0] / o[..., 2], o[..., 1] / o[..., 2]
    sx, sy = -1. / (W / (2. * focal[0])), -1. / (H / (2. * focal[1]))
    o2 = 1. + 2. * near / o[..., 2]
    out_o = torch.stack([sx * ox, sy * oy, o2], -1)
    out_d = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - ox),
                         sy * (rays_d[..., 1] / rays_d[..., 2] - oy), 1 - o2], -1)
    return out_o, out_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """utils/run_nerf_helpers.py:277-294 (scalar focal; d2 = -2 near / o_z)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx, sy = -1. / (W / (2. * focal)), -1. / (H / (2. * focal))
    out_o = torch.stack([sx * o[..., 0] / o[..., 2], sy * o[..., 1] / o[..., 2], 1. + 2. * near / o[..., 2]], -1)
    out_d = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2]),
                         sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2]),
                         -2. * near / o[..., 2]], -1)
    return out_o, out_d


# --------------------------------------------------------------------------------------------
# a3  depth sampling along rays
# --------------------------------------------------------------------------------------------


def stratify(z, t_rand, scale=1.0):
    """Jitter inside [lower, upper] where the bounds are mid-points between neighbours
    (data/ray_utils.py:184-191 with scale=perturb; utils/utils.py:701-706 with scale=1)."""
    mids = .5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * (scale * t_rand)


def ray_marcher(rays, N_samples=64, lindisp=False, perturb=0, noise=None):
    """rays [N,8] = (o, d, near, far).  `noise` in [0,1) replaces torch.rand (data/ray_utils.py:152-197)."""
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    near, far = rays[:, 6:7], rays[:, 7:8]
    t = torch.linspace(0, 1, N_samples, dtype=rays.dtype)
    if not lindisp:
        z = near * (1 - t) + far * t
    else:
        z = 1 / (1 / near * (1 - t) + 1 / far * t)
    z = z.expand(rays.shape[0], N_samples)
    if perturb > 0:
        z = stratify(z, noise, perturb)
    pts = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(2)
    return pts, rays_o, rays_d, z


def sample_points_uniform(near, far, num_samples):
    """near/far [N,1] -> [N,num_samples] (utils/utils.py:393-397)."""
    t = torch.linspace(0., 1., steps=num_samples, dtype=near.dtype).view(1, num_samples)
    return near * (1.0 - t) + far * t


def cascade_depth_candidates(near_1, far_1, near_2, far_2, near_3, far_3, N_samples, t_rand):
    """Live sampler of build_rays / build_rays_test: three N_samples//3 uniform sets, concatenated,
    sorted, stratified with scale 1 (utils/utils.py:684-706)."""
    n = N_samples // 3
    z = torch.cat([sample_points_uniform(near_1, far_1, n), sample_points_uniform(near_2, far_2, n),
                   sample_points_uniform(near_3, far_3, n)], -1)
    z, _ = torch.sort(z, dim=-1)
    return stratify(z, t_rand, 1.0)


# --------------------------------------------------------------------------------------------
# a4  world -> reference-camera normalised coordinates
# --------------------------------------------------------------------------------------------


def project_points(w2c, K, pts, inv_scale):
    """Core of get_ndc_coordinate (utils/utils.py:333-347): returns [..., 3] = (x/(W-1), y/(H-1), z_cam).
    The |z|<1e-4 clamp happens in camera space before the intrinsics are applied."""
    shp = pts.shape
    p = pts.reshape(-1, 3)
    if w2c is not None:
        p = p @ w2c[:3, :3].T + w2c[:3, 3].reshape(1, 3)
        z = torch.where(p[:, 2].abs() < 1e-4, torch.full_like(p[:, 2], 1e-4), p[:, 2])
        p = torch.stack([p[:, 0], p[:, 1], z], -1)
    q = p @ K.T
    xy = (q[:, :2] / q[:, 2:] + 0.0) / inv_scale.reshape(1, 2).to(q.dtype)
    return torch.cat([xy, q[:, 2:]], -1).reshape(shp)


def get_ndc_coordinate(w2c, K, pts, inv_scale, near_far=None, sample_2d=False):
    q = project_points(w2c, K, pts, inv_scale)
    if sample_2d:
        return q
    out = {}
    for key, nk, fk in (("stage1", "near_1", "far_1"), ("stage2", "near_2", "far_2"),
                        ("stage3", "near_3", "far_3"), ("ndc", "near", "far")):
        z = (q[..., 2:] - near_far[nk]) / (near_far[fk] - near_far[nk])
        out[key] = torch.cat([q[..., :2], z], -1)
    return out


# --------------------------------------------------------------------------------------------
# a5  positional encodings
# --------------------------------------------------------------------------------------------


def embed_live(x, n_freqs):
    """network/models.py:50-54: [x | sin(f0 x) sin(f1 x) ... | cos(f0 x) ...], f_k = 2^k, x = 3-vector."""
    f = (2. ** torch.arange(n_freqs, dtype=x.dtype)).view(*([1] * (x.dim() - 1)), -1, 1)
    xs = (x.unsqueeze(-2) * f).reshape(*x.shape[:-1], -1)
    return torch.cat((x, torch.sin(xs), torch.cos(xs)), -1)


def embed_interleaved(x, n_freqs):
    """utils/run_nerf_helpers.py:52-53: [x | sin(f0 x) | cos(f0 x) | sin(f1 x) | cos(f1 x) ...]."""
    parts = [x]
    for k in range(n_freqs):
        parts += [torch.sin(x * (2. ** k)), torch.cos(x * (2. ** k))]
    return torch.cat(parts, -1)


# --------------------------------------------------------------------------------------------
# a6  the uncertainty-conditioned MLP
# --------------------------------------------------------------------------------------------


def ucnerf_param_shapes(W=128, D=6, in_pts=63, in_views=27, n_src=6, skips=(4,)):
    """Ordered (name, shape) list == reference state_dict order (network/models.py:87-104)."""
    out = []
    for i in range(D):
        k = in_pts if i == 0 else (W + in_pts if (i - 1) in skips else W)
        out += [(f"nerf.pts_linears.{i}.weight", (W, k)), (f"nerf.pts_linears.{i}.bias", (W,))]
    out += [("nerf.pts_bias_depth_fine.weight", (W, 24 + 4 * n_src)), ("nerf.pts_bias_depth_fine.bias", (W,)),
            ("nerf.pts_bias_confidence.weight", (W, 8 * n_src)), ("nerf.pts_bias_confidence.bias", (W,)),
            ("nerf.pts_bias_confidence_1.weight", (1, 1)), ("nerf.pts_bias_confidence_1.bias", (1,)),
            ("nerf.views_linears.0.weight", (W // 2, in_views + W)), ("nerf.views_linears.0.bias", (W // 2,)),
            ("nerf.view_confi_linears.0.weight", (W // 2, in_views + W)), ("nerf.view_confi_linears.0.bias", (W // 2,)),
            ("nerf.feature_linear.weight", (W, W)), ("nerf.feature_linear.bias", (W,)),
            ("nerf.feature_linear_1.weight", (W, W)), ("nerf.feature_linear_1.bias", (W,)),
            ("nerf.confi_linear.weight", (W, W)), ("nerf.confi_linear.bias", (W,)),
            ("nerf.alpha_linear.weight", (1, W // 2)), ("nerf.alpha_linear.bias", (1,)),
            ("nerf.alpha_linear_1.weight", (1, W)), ("nerf.alpha_linear_1.bias", (1,)),
            ("nerf.rgb_linear.weight", (3, W // 2)), ("nerf.rgb_linear.bias", (3,)),
            ("nerf.confi_rgb_linear.weight", (3, W)), ("nerf.confi_rgb_linear.bias", (3,))]
    return out


def init_ucnerf_params(seed=0, dtype=torch.float32, **cfg):
    """Kaiming-normal weights / zero biases (network/models.py:13-17,107-118).  The oracle initialises
    *every* Linear that way; which layers the reference leaves at nn.Linear's default init is
    irrelevant to parity because fixtures carry the state_dict itself."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in ucnerf_param_shapes(**cfg):
        if name.endswith(".weight"):
            p[name] = (torch.randn(shape, generator=g, dtype=torch.float64) * math.sqrt(2.0 / shape[1])).to(dtype)
        else:
            p[name] = torch.zeros(shape, dtype=dtype)
    return p


def _lin(p, name, x):
    return x @ p[name + ".weight"].T + p[name + ".bias"]


def ucnerf_mlp(p, x, in_pts=63, in_views=27, n_src=6, D=6, skips=(4,)):
    """BaseAdapt_Renderer.forward (network/models.py:138-184).  x [..., in_pts + F + in_views],
    F = 24 + 12*n_src + 1.  Returns [..., 4] = (rgb after sigmoid, sigma after relu)."""
    F = x.shape[-1] - in_pts - in_views
    pts, feats, views = x[..., :in_pts], x[..., in_pts:in_pts + F], x[..., in_pts + F:]
    n_mvs = 24 + 4 * n_src
    mvs, img = feats[..., :n_mvs], feats[..., n_mvs:n_mvs + 8 * n_src]
    u = 1 - feats[..., -1:]                                  # confidence -> uncertainty
    b_depth = _lin(p, "nerf.pts_bias_depth_fine", mvs)
    b_conf = _lin(p, "nerf.pts_bias_confidence", img)
    h = pts
    for i in range(D):
        h = torch.relu(_lin(p, f"nerf.pts_linears.{i}", h) * b_depth)
        if i in skips:
            h = torch.cat([pts, h], -1)
    base_rgb = _lin(p, "nerf.confi_rgb_linear", h)
    base_sigma = _lin(p, "nerf.alpha_linear_1", h)
    h1 = torch.cat([_lin(p, "nerf.feature_linear", h * b_conf), views], -1)
    adapt_rgb = _lin(p, "nerf.rgb_linear", torch.relu(_lin(p, "nerf.views_linears.0", h1)))
    adapt_sigma = _lin(p, "nerf.alpha_linear", torch.relu(_lin(p, "nerf.view_confi_linears.0", h1)))
    rgb = torch.sigmoid(base_rgb * (1 - u) + adapt_rgb * u)
    sigma = torch.relu(adapt_sigma * (1 - u) + base_sigma * u)
    return torch.cat([rgb, sigma], -1)


def run_network_mvs(p, pts, viewdirs, feats, multires=10, multires_views=4, n_src=6, layout="live", **kw):
    """network/renderer.py:78-106 with the embedders bound: embed(pts) | feats | embed(dirs per sample)."""
    emb = embed_live if layout == "live" else embed_interleaved
    e_pts = emb(pts, multires)
    if viewdirs.dim() != 3:
        viewdirs = viewdirs[:, None].expand(-1, pts.shape[1], -1)
    e_dir = emb(viewdirs, multires_views)
    x = torch.cat([e_pts, feats, e_dir], -1)
    return ucnerf_mlp(p, x, in_pts=e_pts.shape[-1], in_views=e_dir.shape[-1], n_src=n_src, **kw)


# --------------------------------------------------------------------------------------------
# a7  feature gather (explicit restatement of grid_sample semantics)
# --------------------------------------------------------------------------------------------


def _unnorm(g, size, align_corners):
    """[-1,1] grid coordinate -> pixel index, then `border` clamp (ATen grid_sampler semantics)."""
    if align_corners:
        i = (g + 1) / 2 * (size - 1)
    else:
        i = ((g + 1) * size - 1) / 2
    return i.clamp(0, size - 1)


def _corner(vol, idx_list, sizes):
    """vol [C, *sizes]; idx_list per-dim integer index tensors; out-of-range corners contribute 0."""
    ok = torch.ones_like(idx_list[0], dtype=torch.bool)
    safe = []
    for ix, s in zip(idx_list, sizes):
        ok &= (ix >= 0) & (ix < s)
        safe.append(ix.clamp(0, s - 1))
    v = vol[(slice(None),) + tuple(safe)]                    # [C, ...]
    return v * ok.to(vol.dtype)


def sample_bilinear(img, gx, gy, align_corners):
    """img [C,H,W]; gx, gy any shape in [-1,1] -> [C, *shape] (bilinear, border padding)."""
    C, H, W = img.shape
    ix, iy = _unnorm(gx, W, align_corners), _unnorm(gy, H, align_corners)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    tx, ty = ix - x0, iy - y0
    x0, y0 = x0.long(), y0.long()
    out = 0
    for dy, wy in ((0, 1 - ty), (1, ty)):
        for dx, wx in ((0, 1 - tx), (1, tx)):
            out = out + _corner(img, [y0 + dy, x0 + dx], (H, W)) * (wy * wx)
    return out


def sample_trilinear(vol, gx, gy, gz, align_corners=False):
    """vol [C,D,H,W]; grid (x,y,z) <-> (W,H,D) -> [C, *shape] (trilinear, border padding)."""
    C, D, H, W = vol.shape
    ix, iy, iz = _unnorm(gx, W, align_corners), _unnorm(gy, H, align_corners), _unnorm(gz, D, align_corners)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    tx, ty, tz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    out = 0
    for dz, wz in ((0, 1 - tz), (1, tz)):
        for dy, wy in ((0, 1 - ty), (1, ty)):
            for dx, wx in ((0, 1 - tx), (1, tx)):
                out = out + _corner(vol, [z0 + dz, y0 + dy, x0 + dx], (D, H, W)) * (wz * wy * wx)
    return out


def index_point_feature(vols, confidence, ndc):
    """utils/utils.py:833-893.  vols = 3 tensors [1,8,D,h,w]; confidence [H,W] (any leading 1s);
    ndc = dict of [N,S,3] in ~[0,1].  Returns ([N,S,24], [N,S])."""
    feats = []
    for key, vol in zip(("stage1", "stage2", "stage3"), vols):
        g = ndc[key] * 2 - 1.0
        feats.append(sample_trilinear(vol[0], g[..., 0], g[..., 1], g[..., 2]).permute(1, 2, 0))
    g = ndc["stage3"] * 2 - 1.0
    H, W = confidence.shape[-2:]
    conf = sample_bilinear(confidence.reshape(1, H, W), g[..., 0], g[..., 1], align_corners=False)[0]
    return torch.cat(feats, -1), conf


def build_color_volume(pts, w2cs, intrinsics, imgs, img_feat=None):
    """utils/utils.py:742-799 with with_mask=True.  imgs [1,V,3,H,W]; img_feat [V,1,8,H,W];
    w2cs [V,4,4], intrinsics [V,3,3] (source views only).  Returns [N,S,4V (+8V)]."""
    _, V, C, H, W = imgs.shape
    inv_scale = torch.tensor([W - 1, H - 1], dtype=pts.dtype)
    cols, extra = [], []
    for i in range(V):
        q = project_points(w2cs[i], intrinsics[i], pts, inv_scale)
        gx, gy = q[..., 0] * 2.0 - 1.0, q[..., 1] * 2.0 - 1.0
        rgb = sample_bilinear(imgs[0, i], gx, gy, align_corners=True).permute(1, 2, 0)
        mask = ((gx > -1.0) & (gx < 1.0) & (gy > -1.0) & (gy < 1.0)).to(pts.dtype)
        cols += [rgb, mask.unsqueeze(-1)]
        if img_feat is not None:
            extra.append(sample_bilinear(img_feat[i, 0], gx, gy, align_corners=True).permute(1, 2, 0))
    return torch.cat(cols + extra, -1)


def gen_pts_feats(imgs, vols, pts, w2cs, intrinsics, ndc, img_feat, confidence):
    """network/renderer.py:177-212 (use_color_volume=False): [vol 24 | (rgb,mask) x V | img_feat 8 x V | conf]."""
    ray_feats, conf = index_point_feature(vols, confidence, ndc)
    colors = build_color_volume(pts, w2cs, intrinsics, imgs, img_feat)
    return torch.cat([ray_feats, colors, conf.unsqueeze(-1)], -1)


def gen_dir_feature(w2c_ref, rays_dir):
    """network/renderer.py:163-174."""
    return rays_dir @ w2c_ref[:3, :3].T


# --------------------------------------------------------------------------------------------
# a8  inverse-CDF sampling -- with torch-CPU's exact accumulation orders written out
# --------------------------------------------------------------------------------------------


def _f32(a):
    return a.astype(np.float32)


def _multi_row_sum(items, n_groups):
    """ATen cascade accumulation over `n_groups` groups of 4 items (items [R, n, L]): 4 parallel partial
    sums; every 2^p groups (p = max(4, ceil(log2 n_groups)/4)) level-0 spills into level 1, and so on."""
    R, _, L = items.shape
    ceil_log2 = 0 if n_groups <= 1 else int(n_groups - 1).bit_length()
    power = max(4, ceil_log2 // 4)
    step, mask = 1 << power, (1 << power) - 1
    acc = np.zeros((4, 4, R, L), np.float32)            # [level][ilp slot]
    i = 0
    while i + step <= n_groups:
        for _ in range(step):
            for k in range(4):
                acc[0, k] = _f32(acc[0, k] + items[:, 4 * i + k])
            i += 1
        for j in range(1, 4):
            acc[j] = _f32(acc[j] + acc[j - 1])
            acc[j - 1] = 0
            if i & (mask << (j * power)):
                break
    while i < n_groups:
        for k in range(4):
            acc[0, k] = _f32(acc[0, k] + items[:, 4 * i + k])
        i += 1
    for j in range(1, 4):
        acc[0] = _f32(acc[0] + acc[j])
    return acc[0]                                        # [4, R, L]


def _row_sum(items):
    """ATen row_sum: groups of 4 items -> 4 partials (cascade), leftover items into partial 0, fold 1,2,3."""
    n = items.shape[1]
    part = _multi_row_sum(items, n // 4)
    for i in range((n // 4) * 4, n):
        part[0] = _f32(part[0] + items[:, i])
    for k in (1, 2, 3):
        part[0] = _f32(part[0] + part[k])
    return part[0]                                       # [R, L]


def torch_cpu_rowsum_f32(x, lanes=8):
    """Bit pattern of torch-CPU ``x.sum(-1)`` for contiguous float32 rows of length n (ATen SumKernel's
    cascade_sum, restated; pinned for n = 1..70, 127, 128, 190, 191 by tests/golden/g8_rowsum.npz):
      n <  lanes: scalar path -- _row_sum over the n scalars;
      n >= lanes: the row is cut into n//lanes vectors, _row_sum over those vectors gives one vector of
                  partial sums; the n%lanes tail elements are summed first into a scalar (left to right),
                  then the lanes of the partial vector are added to it left to right.
    `lanes` is the float vector width of the ATen build that produced the fixtures (8)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    n = x.shape[-1]
    rows = x.reshape(-1, n)
    if n < lanes:
        out = _row_sum(rows[:, :, None])[:, 0]
    else:
        nv = n // lanes
        vacc = _row_sum(rows[:, :nv * lanes].reshape(-1, nv, lanes))
        out = np.zeros(rows.shape[0], np.float32)
        for k in range(nv * lanes, n):
            out = _f32(out + rows[:, k])
        for lane in range(lanes):
            out = _f32(out + vacc[:, lane])
    return out.reshape(x.shape[:-1])


def torch_cpu_cumsum_f32(x):
    """torch-CPU cumsum of float32 rows: sequential float64 accumulator, each output rounded to f32."""
    return np.cumsum(np.asarray(x, np.float32).astype(np.float64), -1).astype(np.float32)


def sample_pdf(bins, weights, u, exact_f32=True):
    """data/ray_utils.py:98-141 with the uniform draws `u` [N,M] given.  Returns (samples, inds, cdf).
    `inds` = searchsorted(cdf, u, right=True) = number of cdf entries <= u (int64).
    With exact_f32 the float32 path reproduces torch-CPU's accumulation order bit for bit."""
    weights = weights.detach()     # the hierarchy detaches the fine depths (data/ray_utils.py:217)
    if exact_f32 and weights.dtype == torch.float32:
        w = (weights + 1e-5).contiguous()
        tot = torch.from_numpy(torch_cpu_rowsum_f32(w.numpy())).unsqueeze(-1)
        pdf = w / tot
        cdf = torch.from_numpy(torch_cpu_cumsum_f32(pdf.numpy()))
    else:
        w = weights + 1e-5
        pdf = w / w.sum(-1, keepdim=True)
        cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    L = cdf.shape[-1]
    inds = (cdf.unsqueeze(-2) <= u.unsqueeze(-1)).sum(-1)           # right=True
    below, above = (inds - 1).clamp(min=0), inds.clamp(max=L - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    return b0 + t * (b1 - b0), inds, cdf


def merge_sorted(z_a, z_b):
    """sort(cat([z_a, z_b])) (data/ray_utils.py:219) -- values only, so ties need no rule."""
    return torch.sort(torch.cat([z_a, z_b], -1), -1)[0]


# --------------------------------------------------------------------------------------------
# a9  alpha compositing
# --------------------------------------------------------------------------------------------


def depth2dist(z, cos_angle):
    """network/renderer.py:6-15 (computed by rendering() but unused by raw2alpha)."""
    d = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], -1)
    return d * cos_angle.unsqueeze(-1)


def composite_weights(alpha):
    """w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-10)."""
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    return alpha * T


def raw2outputs_live(raw, z, white_bkgd=False):
    """network/renderer.py:25-36,109-140: alpha = 1-exp(-sigma) (dists ignored; rgb/sigma already activated).
    Returns (rgb_map, disp_map, acc_map, weights, depth_map, weights, var)."""
    rgb, sigma = raw[..., :3], raw[..., 3]
    w = composite_weights(1. - torch.exp(-sigma))
    var = torch.var(w, dim=1, unbiased=True)
    rgb_map = (w[..., None] * rgb).sum(-2)
    depth_map = (w * z).sum(-1)
    acc = w.sum(-1)
    disp = 1. / torch.maximum(torch.full_like(depth_map, 1e-10), depth_map / acc)   # NaN (0/0) propagates
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc[..., None])
    return rgb_map, disp, acc, w, depth_map, w, var


def raw2outputs_helpers(raw, z, rays_d, noise=None, white_bkgd=False):
    """utils/run_nerf_helpers.py:343-390: alpha = 1-exp(-relu(sigma+noise)*dists), rgb = sigmoid(raw)."""
    dists = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    s = raw[..., 3] if noise is None else raw[..., 3] + noise
    w = composite_weights(1. - torch.exp(-torch.relu(s) * dists))
    rgb_map = (w[..., None] * rgb).sum(-2)
    depth_map = (w * z).sum(-1)
    acc = w.sum(-1)
    disp = 1. / torch.maximum(torch.full_like(depth_map, 1e-10), depth_map / acc)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc[..., None])
    return rgb_map, disp, acc, w, depth_map


# --------------------------------------------------------------------------------------------
# a10  orchestration
# --------------------------------------------------------------------------------------------


def rendering(p, pose_ref, rays_pts, rays_ndc, depth_candidates, rays_dir, vols, imgs, img_feat, confidence,
              view_num, white_bkgd=False, full=False, **mlp_kw):
    """network/renderer.py:215-255, including the in-place trim of pose_ref on the first call."""
    cos_angle = torch.norm(rays_dir, dim=-1)
    angle = gen_dir_feature(pose_ref["w2cs"][0], rays_dir / cos_angle.unsqueeze(-1))
    if pose_ref["w2cs"].shape[0] == view_num:
        pose_ref["w2cs"] = pose_ref["w2cs"][1:]
        pose_ref["intrinsics"] = pose_ref["intrinsics"][1:]
    feats = gen_pts_feats(imgs, vols, rays_pts, pose_ref["w2cs"], pose_ref["intrinsics"], rays_ndc, img_feat,
                          confidence)
    raw = run_network_mvs(p, rays_ndc["ndc"], angle, feats, n_src=imgs.shape[1], **mlp_kw)
    rgb_map, disp, acc, w, depth_map, _, var = raw2outputs_live(raw, depth_candidates, white_bkgd)
    if full:
        # opt-in uncertainty outputs (SURVEY.md 8(a) note): u = 1 - sampled confidence is what the network blends its
        # heads with (network/models.py:149,177-178); its composite sum_i w_i u_i is a build extra (no reference line)
        u = 1 - feats[..., -1]
        return dict(rgb=rgb_map, depth=depth_map, acc=acc, weights=w, var=var, disp=disp, raw=raw, feats=feats,
                    u_sampled=u, wu=(w * u).sum(-1))
    return rgb_map, depth_map


def scene_near_far(N, S, near, far, dtype=torch.float32):
    """near_far dict with constant per-stage ranges (the synthetic-scene convention of SURVEY 8(d))."""
    nf = {}
    for k in ("1", "2", "3"):
        nf["near_" + k] = torch.full((N, S, 1), near, dtype=dtype)
        nf["far_" + k] = torch.full((N, S, 1), far, dtype=dtype)
    nf["near"], nf["far"] = torch.tensor(near, dtype=dtype), torch.tensor(far, dtype=dtype)
    return nf


def render_coarse_fine(p, scene, xs, ys, n_coarse=64, n_fine=128, u=None, noise=None, perturb=0.0,
                       white_bkgd=False, z_fine_override=None, **mlp_kw):
    """The 64+128 composition of SURVEY 3.3 from the reference's pieces (data/ray_utils.py:199-224 recipe):
    rays -> ray_marcher(n_coarse) -> render -> sample_pdf(mid-points, w[1:-1], n_fine) -> sorted merge ->
    render(n_coarse+n_fine).  `scene` keys: K[3,3] (target), c2w[4,4] (target), w2cs[V,4,4] and
    intrinsics[V,3,3] (index 0 = reference view), near, far, vols (3), imgs [1,V-1,3,H,W],
    img_feat [V-1,1,8,H,W], confidence [H,W]."""
    dt = scene["c2w"].dtype
    N = xs.shape[0]
    H, W = scene["imgs"].shape[-2:]
    inv_scale = torch.tensor([W - 1, H - 1], dtype=dt)
    rays_o, rays_d, _ = get_rays_mvs_pixels(xs.to(dt), ys.to(dt), scene["K"], scene["c2w"])
    rays = torch.cat([rays_o.expand(N, 3), rays_d, torch.full((N, 1), scene["near"], dtype=dt),
                      torch.full((N, 1), scene["far"], dtype=dt)], -1)
    V = scene["w2cs"].shape[0]

    def one_pass(z):
        S = z.shape[1]
        pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
        nf = scene_near_far(N, S, scene["near"], scene["far"], dt)
        ndc = get_ndc_coordinate(scene["w2cs"][0], scene["intrinsics"][0], pts, inv_scale, nf)
        pose = {"w2cs": scene["w2cs"].clone(), "intrinsics": scene["intrinsics"].clone()}
        return rendering(p, pose, pts, ndc, z, rays_d, scene["vols"], scene["imgs"], scene["img_feat"],
                         scene["confidence"], V, white_bkgd, full=True, **mlp_kw)

    _, _, _, z_c = ray_marcher(rays, n_coarse, perturb=perturb, noise=noise)
    coarse = one_pass(z_c)
    z_mid = .5 * (z_c[:, :-1] + z_c[:, 1:])
    if u is None:
        u = torch.linspace(0., 1., n_fine, dtype=dt).expand(N, n_fine)
    z_s, inds, _ = sample_pdf(z_mid, coarse["weights"][:, 1:-1].contiguous(), u.contiguous())
    z_f = merge_sorted(z_s, z_c) if z_fine_override is None else z_fine_override
    fine = one_pass(z_f)
    fine.update(z_coarse=z_c, z_fine=z_f, z_samples=z_s, inds=inds, coarse=coarse)
    return fine


def make_synthetic_scene(seed=0, H=256, W=320, V=7, near=1.0, far=4.0, dtype=torch.float32, small=False):
    """Synthetic scene of SURVEY 8(d): K = [[250,0,160],[0,250,128],[0,0,1]] scaled to (H, W),
    w2c_i = I with t_x = 0.01 i, randn volumes / rand images / randn image features / clamped rand confidence.
    `small` shrinks the cascade volumes (for fixtures)."""
    g = torch.Generator().manual_seed(seed)
    fx = 250.0 * W / 320.0
    K = torch.tensor([[fx, 0, W / 2.0], [0, fx, H / 2.0], [0, 0, 1]], dtype=dtype)
    w2cs = torch.eye(4, dtype=dtype).repeat(V, 1, 1)
    w2cs[:, 0, 3] = 0.01 * torch.arange(V, dtype=dtype)
    D = (6, 4, 2) if small else (48, 32, 8)
    vols = [torch.randn(1, 8, D[0], H // 4, W // 4, generator=g).to(dtype),
            torch.randn(1, 8, D[1], H // 2, W // 2, generator=g).to(dtype),
            torch.randn(1, 8, D[2], H, W, generator=g).to(dtype)]
    imgs = torch.rand(1, V - 1, 3, H, W, generator=g).to(dtype)
    img_feat = torch.randn(V - 1, 1, 8, H, W, generator=g).to(dtype)
    conf = torch.rand(H, W, generator=g).clamp(1e-3, 1).to(dtype)
    return dict(K=K, c2w=torch.eye(4, dtype=dtype), w2cs=w2cs, intrinsics=K.repeat(V, 1, 1), near=near, far=far,
                vols=vols, imgs=imgs, img_feat=img_feat, confidence=conf, H=H, W=W)
