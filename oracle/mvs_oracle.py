"""CPU oracle for the step in front of the ray-marching path: cost-volume assembly and depth regression of the
cascade MVS stage.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***  Same rules as ``oracle/ucnerf_oracle.py``: only ``tests/`` may import it;
the product package never does.

Parity status: PINNED by ``tests/golden/g12_cost_volume.npz`` and ``g13_depth_regress.npz``, captured by running the
reference's own ``homo_warp`` (utils/utils.py:1105-1172) and ``DepthNet.forward`` (network/mvs_models.py:589-646) in
the build container (``tests/golden/make_golden.py``).  The reference's ``homo_warp`` calls
``kornia.utils.create_meshgrid`` (kornia >= 0.6.12, requirements.txt:12), which is absent from this image; the capture
script supplies that one function restated from its published behaviour (pixel grid, x then y, shape [1,H,W,2]) --
``create_meshgrid_pixels`` below is the same restatement.

Restated here with explicit index arithmetic where the reference leans on ``F.grid_sample`` so that the HIP kernels have
an exact description of what to compute.
"""
import torch


def create_meshgrid_pixels(H, W, dtype=torch.float32):
    """kornia.utils.create_meshgrid(H, W, normalized_coordinates=False): [1,H,W,2], last dim = (x, y)."""
    xs = torch.linspace(0, W - 1, W, dtype=dtype)
    ys = torch.linspace(0, H - 1, H, dtype=dtype)
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")                       # [W,H]
    return torch.stack([gx, gy], -1).permute(1, 0, 2).unsqueeze(0)       # [1,H,W,2]


def homo_warp_grid(proj_mat, depth_values, H, W, pad=0):
    """utils/utils.py:1113-1156.  proj_mat [3,4] = (src_proj @ ref_proj_inv)[:3]; depth_values [D,Hp,Wp] (Hp = H + 2 pad).
    Returns the normalised sampling grid [D, Hp*Wp, 2] (x, y in [-1,1] of the SOURCE map of size H x W)."""
    D, Hp, Wp = depth_values.shape
    assert Hp == H + 2 * pad and Wp == W + 2 * pad
    R, T = proj_mat[:, :3], proj_mat[:, 3:]
    grid = create_meshgrid_pixels(Hp, Wp, depth_values.dtype)
    if pad > 0:
        grid = grid - pad
    g = grid.permute(0, 3, 1, 2).reshape(1, 2, Hp * Wp)
    g = torch.cat((g, torch.ones_like(g[:, :1])), 1)                     # [1,3,Hp*Wp]
    g = g.repeat(1, 1, D)                                                # [1,3,D*Hp*Wp]
    q = R.unsqueeze(0) @ g + T.unsqueeze(0) / depth_values.reshape(1, 1, D * Hp * Wp)
    xy = q[:, :2] / q[:, 2:]
    x = xy[:, 0] / ((W - 1) / 2) - 1
    y = xy[:, 1] / ((H - 1) / 2) - 1
    return torch.stack([x, y], -1).reshape(D, Hp * Wp, 2)


def sample_nearest_border(src, grid):
    """F.grid_sample(src[None], grid[None], mode='nearest', padding_mode='border', align_corners=True) with the index
    arithmetic spelled out (ATen GridSampler: unnormalise, clip, nearbyint).  src [C,H,W], grid [..., 2] -> [C, ...]."""
    C, H, W = src.shape
    ix = ((grid[..., 0] + 1) / 2) * (W - 1)
    iy = ((grid[..., 1] + 1) / 2) * (H - 1)
    ix = torch.clamp(ix, 0, W - 1)
    iy = torch.clamp(iy, 0, H - 1)
    ix = torch.round(ix).long()                                          # round-half-even == std::nearbyint
    iy = torch.round(iy).long()
    return src[:, iy, ix]


def cost_volume_variance(feats, proj_mats, depth_values, pad=0):
    """network/mvs_models.py:599-626 without the (unused there) warped-image volume.
    feats [V,C,H,W] source-view feature maps, proj_mats [V,3,4], depth_values [D,Hp,Wp].
    Returns (variance [C,D,Hp,Wp], count [D,Hp,Wp]).  The mask stack has V+1 slots initialised to one and only V of
    them are overwritten (mvs_models.py:607,618), so count = 1 / (1 + number of views that see the voxel)."""
    V, C, H, W = feats.shape
    D, Hp, Wp = depth_values.shape
    vsum = torch.zeros(C, D, Hp * Wp, dtype=feats.dtype)
    vsq = torch.zeros_like(vsum)
    msum = torch.ones(D, Hp * Wp, dtype=feats.dtype)
    for i in range(V):
        grid = homo_warp_grid(proj_mats[i], depth_values, H, W, pad)
        warped = sample_nearest_border(feats[i], grid)                   # [C,D,Hp*Wp]
        inside = ((grid > -1.0) * (grid < 1.0))
        msum = msum + (inside[..., 0] * inside[..., 1]).to(feats.dtype)
        vsum = vsum + warped
        vsq = vsq + warped ** 2
    count = 1.0 / msum
    var = vsq * count - (vsum * count) ** 2
    return var.reshape(C, D, Hp, Wp), count.reshape(D, Hp, Wp)


def depth_regress(prob_pre, depth_values, prob_init=None, pad=0):
    """network/mvs_models.py:629-646.  prob_pre [D,Hp,Wp] (the regularisation net's output), depth_values [D,Hp,Wp].
    Returns (prob_volume [D,Hp,Wp], depth [H,W], photometric_confidence [H,W])."""
    D = prob_pre.shape[0]
    if prob_init is not None:
        prob_pre = prob_pre + prob_init
    p = torch.softmax(prob_pre, dim=0)
    depth = torch.sum(p * depth_values, 0)
    # 4 * avg_pool3d over a window of 4 along depth, padded (1, 2): sum of p[d-1 .. d+2]
    pp = torch.cat([p.new_zeros((1,) + p.shape[1:]), p, p.new_zeros((2,) + p.shape[1:])], 0)      # (explicit shapes: D = 1 has no p[:2])
    sum4 = 4 * ((pp[0:D] + pp[1:D + 1] + pp[2:D + 2] + pp[3:D + 3]) / 4)
    idx = torch.sum(p * torch.arange(D, dtype=p.dtype).view(D, 1, 1), 0).long().clamp(0, D - 1)
    conf = torch.gather(sum4, 0, idx.unsqueeze(0)).squeeze(0).clamp(0, 1)
    if pad > 0:
        depth = depth[pad:-pad, pad:-pad]
        conf = conf[pad:-pad, pad:-pad]
    return p, depth, conf
