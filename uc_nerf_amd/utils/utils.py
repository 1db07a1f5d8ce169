"""Drop-in for the hot-path functions of the reference's utils/utils.py: ray generation (`get_rays_mvs`,
`get_rays_mvs_coord`, `get_rays_with_random_patches`), `get_ndc_coordinate`, `sample_points_uniform`, the two ray
builders (`build_rays`, `build_rays_test`), the two halves of the feature gather (`index_point_feature`,
`build_color_volume`) and the small helpers train.py touches (`filter_keys`, `img2mse`, `mse2psnr2`, `init_log`,
`sub_selete_data`).  Visualisation, PFM/IO, pose-path generators, `homo_warp` and schedulers are out of scope
(SURVEY.md 2.1 row 6).

Random pixel selection (randint / multinomial / numpy shifts) stays host-side torch/numpy logic exactly where the
reference has it, in the same call order; everything per-ray or per-sample after that is a HIP kernel.
"""
import numpy as np
import torch

from .. import ops

img2mse = lambda x, y: torch.mean((x - y) ** 2)                     # noqa: E731
mse2psnr2 = lambda x: -10. * np.log(x) / np.log(10.)                # noqa: E731
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)          # noqa: E731


def init_log(log, keys):
    for key in keys:
        log[key] = torch.tensor([0.0], dtype=float)
    return log


def filter_keys(dict):
    """utils/utils.py:929-935."""
    dict.pop('N_samples')
    for k in ('ndc', 'lindisp'):
        if k in dict.keys():
            dict.pop(k)
    return dict


def sub_selete_data(data_batch, device, idx, filtKey=[], filtIndex=()):
    """utils/utils.py:938-956: the batch is cast to float and moved to `device` (the index filter of the reference
    never triggers because it tests the key string, not the tensor)."""
    return {k: v.float().to(device) for k, v in data_batch.items()}


# ------------------------------------------------------------------------------------------------ rays
def _rays_from_pixels(xs, ys, intrinsic, c2w):
    rays_d, _, _ = ops.ray_gen(intrinsic, c2w, xs=xs.float().contiguous(), ys=ys.float().contiguous())
    rays_o = c2w[:3, -1].clone()
    return rays_o, rays_d, torch.stack((ys.float(), xs.float()))    # pixel_coordinates = (row, col)


def get_rays_mvs(H, W, intrinsic, c2w, N=1024, isRandom=True, is_precrop_iters=False, chunk=-1, idx=-1,
                 with_mask=False, photo_confidence=None):
    """utils/utils.py:217-271 -> (rays_o [3], rays_d [N,3], pixel_coordinates [2,N] as (row, col))."""
    device = c2w.device
    if not isRandom:
        total = H * W
        start, n = (idx * chunk, max(0, min(chunk, total - idx * chunk))) if chunk > 0 else (0, total)
        rays_d, _, pix = ops.ray_gen(intrinsic, c2w, H=H, W=W, grid_start=start, n=n, device=device, want_pix=True)
        return c2w[:3, -1].clone(), rays_d, pix
    if with_mask:
        select = torch.multinomial(photo_confidence.reshape(-1), N)
        xs, ys = (select % W).float().to(device), (select // W).float().to(device)
    elif is_precrop_iters and torch.rand((1,)) > 0.3:
        xs = torch.randint(W // 6, W - W // 6, (N,)).float().to(device)
        ys = torch.randint(H // 6, H - H // 6, (N,)).float().to(device)
    else:
        xs = torch.randint(0, W, (N,)).float().to(device)
        ys = torch.randint(0, H, (N,)).float().to(device)
    return _rays_from_pixels(xs, ys, intrinsic, c2w)


def get_rays_mvs_coord(H, W, intrinsic, c2w, coords, sparse_depths=None, N=1024, with_mask=False, photo_confidence=None):
    """utils/utils.py:274-320: rays through given (row, col) coordinates, or confidence-weighted picks among the
    pixels with a sparse depth."""
    device = c2w.device
    xs, ys = coords[:, 1], coords[:, 0]
    if with_mask:
        conf_mask = photo_confidence * (sparse_depths > 0)
        if torch.count_nonzero(conf_mask) >= N:
            select = torch.multinomial(conf_mask.reshape(-1), N)
            xs, ys = (select % W).float(), (select // W).float()
    return _rays_from_pixels(xs.to(device), ys.to(device), intrinsic, c2w.float())


def get_rays_with_random_patches(H, W, intrinsic, c2w, patch_size=30, num_patches=1, random=True, confidence=None):
    """utils/utils.py:169-215: `num_patches` square patches (uniform, or drawn from `confidence`), each shifted by a
    numpy-random offset inside its cell.  -> (rays_o [3*num_patches], rays_d [P,3], pixel_coordinates [2,P])."""
    device = c2w.device
    ps = patch_size
    if random:
        cell_r = torch.randint(0, H // ps, size=(num_patches,))
        cell_c = torch.randint(0, W // ps, size=(num_patches,))
    else:
        select = torch.multinomial(confidence.reshape(-1), num_patches)
        cell_r = torch.clamp((select // W) // ps, min=0, max=H // ps - 2).long()
        cell_c = torch.clamp((select % W) // ps, min=0, max=W // ps - 2).long()
    xs, ys = [], []
    for k in range(num_patches):
        r0 = int(cell_r[k]) * ps + np.random.randint(0, ps)
        c0 = int(cell_c[k]) * ps + np.random.randint(0, ps)
        rows = torch.arange(r0, min(r0 + ps, H), dtype=torch.float32)       # slicing past the border truncates
        cols = torch.arange(c0, min(c0 + ps, W), dtype=torch.float32)
        ys.append(rows.repeat_interleave(cols.numel()))
        xs.append(cols.repeat(rows.numel()))
    xs, ys = torch.cat(xs).to(device), torch.cat(ys).to(device)
    _, rays_d, pix = _rays_from_pixels(xs, ys, intrinsic, c2w)
    return c2w[:3, -1].clone().repeat(num_patches), rays_d, pix


# ------------------------------------------------------------------------------------------------ projection
def get_ndc_coordinate(w2c_ref, intrinsic_ref, point_samples, inv_scale, near_far=None, sample_2d=False):
    """utils/utils.py:323-373."""
    return ops.ndc_project(point_samples, w2c_ref, intrinsic_ref, inv_scale, near_far, sample_2d)


def sample_points_uniform(device, near, far, num_samples, num_rays):
    """utils/utils.py:393-397 (tiny broadcast; the live path uses the fused cascade sampler instead)."""
    t = torch.linspace(0., 1., steps=num_samples).view(1, num_samples).to(near.device).expand(num_rays, num_samples)
    return near * (1.0 - t) + far * t


def _stage_ranges(outputs, pix):
    """Per-ray (near_1, far_1, near_2, far_2, near_3, far_3) from the cascade depth hypotheses at the ray's pixel
    (stage 1 at 1/4 resolution, stage 2 at 1/2): utils/utils.py:659-683."""
    cols = []
    for key, div in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
        dv = outputs[key]['depth_values']
        r = torch.div(pix[0], div, rounding_mode='trunc').long()
        c = torch.div(pix[1], div, rounding_mode='trunc').long()
        cols += [dv[0, 0, r, c], dv[0, -1, r, c]]
    return torch.stack(cols, -1).float().contiguous()


def _near_far_dict(ranges, N_samples, near_ref, far_ref):
    nf = {}
    for i, k in enumerate(("near_1", "far_1", "near_2", "far_2", "near_3", "far_3")):
        nf[k] = ranges[:, i:i + 1].expand(-1, N_samples).unsqueeze(-1)
    nf['near'], nf['far'] = near_ref, far_ref
    return nf


_INV_SCALE = {}


def _inv_scale(W, H, device):
    """torch.tensor([W - 1, H - 1]) on `device` (utils/utils.py:633), made ONCE per image size and device: the reference uploads it in every call
    -- a host-to-device copy per 1024-pixel chunk; callers only read it."""
    key = (int(W), int(H), str(device))
    t = _INV_SCALE.get(key)
    if t is None:
        t = _INV_SCALE[key] = torch.tensor([W - 1, H - 1]).to(device)
    return t


def build_rays_test(H, W, tgt_to_world, world_to_ref, intrinsic, near_fars_ref, near_fars, N_samples, pad=0, ref_idx=0,
                    use_cpu=False, chunk=-1, idx=-1, outputs=None):
    """utils/utils.py:600-739 -> (pts [R,S,3], rays_dir [R,3], ndc dict, depth_candidates [R,S], rays_o [R,3],
    ndc_parameters).  ONE launch per call (ucnerf_build_rays_test) behind torch's draw of the jitter: pixel -> ray, the three cascade ranges
    read at px // 4, px // 2, px, the 3 x N_samples/3 depths sorted and jittered, the world points and their four normalised copies; the
    camera matrices are read from the device tensors handed in (the reference's ops do the same: nothing is read back per chunk)."""
    if use_cpu:
        raise RuntimeError("uc_nerf_amd.build_rays_test: use_cpu=True is not available (GPU-only implementation)")
    device = tgt_to_world.device
    inv_scale = _inv_scale(W, H, device)
    intrinsic_render = intrinsic if intrinsic.dim() == 2 else intrinsic.mean(0)
    total = H * W
    start, n = (idx * chunk, max(0, min(chunk, total - idx * chunk))) if chunk > 0 else (0, total)
    near_ref, far_ref = near_fars_ref[ref_idx, 0], near_fars_ref[ref_idx, 1]
    t_rand = torch.rand((n, N_samples), device=device)
    if intrinsic.dim() != 2 or world_to_ref.dim() != 2:
        raise RuntimeError("uc_nerf_amd.build_rays_test: the reference view's intrinsic must be [3,3] and world_to_ref [4,4]")
    o = ops.build_rays_test(H, W, start, n, N_samples, intrinsic_render, tgt_to_world, world_to_ref, intrinsic, near_fars_ref[ref_idx],
                            [outputs["stage%d" % k]['depth_values'] for k in (1, 2, 3)], t_rand)
    ndc = {"stage1": o["stage1"], "stage2": o["stage2"], "stage3": o["stage3"], "ndc": o["ndc"]}
    ndc_parameters = {'w2c_ref': world_to_ref, 'intrinsic_ref': intrinsic, 'inv_scale': inv_scale, 'near': near_ref,
                      'far': far_ref, 'pad': pad}
    return o["pts"], o["rays_d"], ndc, o["z"], o["rays_o"].reshape(1, 3).expand(n, -1), ndc_parameters


def build_rays(args, imgs, mvs_confidence, sparse_depths, coords, pose_ref, w2cs, c2ws, intrinsics, N_rays, N_samples,
               pad=0, is_precrop_iters=False, ref_idx=0, with_depth=False, outputs=None):
    """utils/utils.py:400-597, the training sampler: patch_num/2 patches drawn from the confidence map, patch_num/2
    from the uncertainty map, uniform rays up to N_rays, then one ray per sparse-depth pixel.  -> the reference's
    9-tuple (pts, rays_dir, colours, ndc dict, depth_candidates, rays_o, None, ndc_parameters, pixel coords [2,R])."""
    if not with_depth:
        raise UnboundLocalError("build_rays(with_depth=False) leaves depth_candidate undefined in the reference "
                                "(utils/utils.py:572); train.py always passes with_depth=True")
    device = imgs.device
    N, V, C, H, W = imgs.shape
    w2c_ref, intrinsic_ref = pose_ref['w2cs'][ref_idx], pose_ref['intrinsics'][ref_idx]
    inv_scale = torch.tensor([W - 1, H - 1]).to(device)
    near_ref, far_ref = pose_ref['near_fars'][ref_idx, 0], pose_ref['near_fars'][ref_idx, 1]
    intrinsic, c2w = intrinsics[0], c2ws[0].clone()
    half = args.patch_num // 2
    _, d_p0, pix_p0 = get_rays_with_random_patches(H, W, intrinsic, c2w, patch_size=args.patch_size, num_patches=half,
                                                   random=False, confidence=mvs_confidence)
    _, d_p1, pix_p1 = get_rays_with_random_patches(H, W, intrinsic, c2w, patch_size=args.patch_size, num_patches=half,
                                                   random=False, confidence=1 - mvs_confidence)
    rays_o, d_u, pix_u = get_rays_mvs(H, W, intrinsic, c2w, N_rays - (len(d_p0) + len(d_p1)),
                                      is_precrop_iters=is_precrop_iters)
    _, d_s, pix_s = get_rays_mvs_coord(H, W, intrinsic, c2w, coords, sparse_depths=sparse_depths)
    rays_d = torch.cat([d_p0, d_p1, d_u, d_s], dim=0)
    pix = torch.cat([pix_p0.long(), pix_p1.long(), pix_u.long(), pix_s.long()], dim=1)
    n = rays_d.shape[0]
    colors = imgs[0, 0, :, pix[0], pix[1]].permute(1, 0)
    ranges = _stage_ranges(outputs, pix)
    t_rand = torch.rand((n, N_samples), device=device)
    z, pts = ops.sample_cascade(ranges, N_samples, t_rand, rays_o, rays_d)
    near_far = _near_far_dict(ranges, N_samples, near_ref, far_ref)
    points_all = get_ndc_coordinate(w2c_ref, intrinsic_ref, pts, inv_scale, near_far)
    ndc_parameters = {'w2c_ref': w2c_ref, 'intrinsic_ref': intrinsic_ref, 'inv_scale': inv_scale, 'near': near_ref,
                      'far': far_ref, 'pad': pad}
    return pts, rays_d, colors, points_all, z, rays_o.reshape(1, 3).expand(n, -1), None, ndc_parameters, pix


# ------------------------------------------------------------------------------------------------ gather halves
def index_point_feature(volume_feature, volume_confidence, ray_coordinate_ref, chunk=-1):
    """utils/utils.py:833-893 -> ([N,S,24] cascade features, [N,S] confidence), squeezed like the reference."""
    vols = [volume_feature["stage%d" % k]["volume_feature_no_ref"] for k in (1, 2, 3)]
    feats = ops._FeatGather.apply(vols[0], vols[1], vols[2], volume_confidence, None, None, None, None, None,
                                  ray_coordinate_ref['stage1'], ray_coordinate_ref['stage2'], ray_coordinate_ref['stage3'])
    return feats[..., :24].squeeze(), feats[..., -1].squeeze()


def build_color_volume(point_samples, pose_ref, imgs, near_far=None, img_feat=None, confidence=None, downscale=1.0,
                       with_mask=False):
    """utils/utils.py:742-799 -> [N,S, 4V (+ 8V)]: per source view (r,g,b,in-mask), then the image features."""
    if not with_mask:
        raise NotImplementedError("uc_nerf_amd.build_color_volume: the reference only ever calls it with with_mask=True")
    V = imgs.shape[1]
    feats = ops._FeatGather.apply(None, None, None, None, img_feat, imgs, pose_ref['w2cs'], pose_ref['intrinsics'],
                                  point_samples, None, None, None)
    return feats[..., 24:24 + (12 if img_feat is not None else 4) * V]
