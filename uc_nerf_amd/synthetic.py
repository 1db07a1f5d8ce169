"""Synthetic scenes and reference-style parameter initialisation: there is no network for
datasets or checkpoints, so benches, smoke tests and examples run on seeded random inputs of the real shapes."""
import math
from collections import OrderedDict

import torch


def ucnerf_param_shapes(n_src=6, W=128, D=6, in_pts=63, in_views=27, skips=(4,)):
    """(name, shape) in the reference's state_dict order (network/models.py:87-104)."""
    out = []
    for i in range(D):
        k = in_pts if i == 0 else (W + in_pts if (i - 1) in skips else W)
        out += [("nerf.pts_linears.%d.weight" % i, (W, k)), ("nerf.pts_linears.%d.bias" % i, (W,))]
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


# Linears the reference re-initialises with kaiming-normal weights / zero bias (network/models.py:107-118);
# the others (pts_bias_confidence, alpha_linear_1) keep nn.Linear's default init.
KAIMING = ("pts_bias_depth_fine", "pts_linears", "views_linears", "view_confi_linears", "confi_linear",
           "pts_bias_confidence_1", "feature_linear", "feature_linear_1", "alpha_linear.", "rgb_linear",
           "confi_rgb_linear")


def init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=1.0, sigma_bias=0.0):
    """Random parameters with the reference's init policy.  sigma_scale / sigma_bias rescale the two density
    heads so that a random network produces non-degenerate weights along a ray (benches, smoke test)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in ucnerf_param_shapes(n_src):
        layer = name.rsplit(".", 1)[0]
        fan_in = shape[1] if len(shape) == 2 else dict(ucnerf_param_shapes(n_src))[layer + ".weight"][1]
        kaiming = any(("nerf." + k) in (layer + ".") for k in KAIMING)
        if name.endswith(".weight"):
            if kaiming:
                sd[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            else:   # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:
            sd[name] = torch.zeros(shape) if kaiming else (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    for n in ("nerf.alpha_linear", "nerf.alpha_linear_1"):
        sd[n + ".weight"] *= sigma_scale
        sd[n + ".bias"] += sigma_bias
    return sd


def make_scene(seed=0, H=256, W=320, V=7, near=1.0, far=4.0, small_volumes=False):
    """K = [[250,0,160],[0,250,128],[0,0,1]] scaled to (H, W); w2c_i = I with t_x = 0.01 i (view 0 = reference =
    target); randn cascade volumes [1,8,48,H/4,W/4], [1,8,32,H/2,W/2], [1,8,8,H,W]; rand images; randn image
    features; confidence rand.clamp(1e-3, 1).  CPU tensors; move with scene_to()."""
    g = torch.Generator().manual_seed(seed)
    fx = 250.0 * W / 320.0
    K = torch.tensor([[fx, 0, W / 2.0], [0, fx, H / 2.0], [0, 0, 1]])
    w2cs = torch.eye(4).repeat(V, 1, 1)
    w2cs[:, 0, 3] = 0.01 * torch.arange(V)
    D = (6, 4, 2) if small_volumes else (48, 32, 8)
    vols = [torch.randn(1, 8, D[0], H // 4, W // 4, generator=g), torch.randn(1, 8, D[1], H // 2, W // 2, generator=g),
            torch.randn(1, 8, D[2], H, W, generator=g)]
    imgs = torch.rand(1, V - 1, 3, H, W, generator=g)
    img_feat = torch.randn(V - 1, 1, 8, H, W, generator=g)
    conf = torch.rand(H, W, generator=g).clamp(1e-3, 1)
    return dict(K=K, c2w=torch.eye(4), w2cs=w2cs, intrinsics=K.repeat(V, 1, 1), near=near, far=far, vols=vols, imgs=imgs,
                img_feat=img_feat, confidence=conf, H=H, W=W)


def scene_to(scene, device):
    out = {}
    for k, v in scene.items():
        if torch.is_tensor(v):
            out[k] = v.to(device)
        elif isinstance(v, list):
            out[k] = [t.to(device) for t in v]
        else:
            out[k] = v
    return out


def random_pixels(n, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    ys = torch.randint(0, H, (n,), generator=g).float()
    xs = torch.randint(0, W, (n,), generator=g).float()
    return xs, ys


def cascade_outputs(scene, seed=0):
    """Stand-in for the consistency learner's per-stage outputs that the ray builders read (utils/utils.py:659-683): depth
    hypotheses `depth_values` [1,D,h,w] per cascade stage (48 / 32 / 8 planes, nested ranges around a smooth depth map) and the
    three `volume_feature_no_ref` volumes of the scene.  Tensors on the scene's device."""
    dev = scene["confidence"].device
    g = torch.Generator().manual_seed(seed)
    H, W, near, far = scene["H"], scene["W"], scene["near"], scene["far"]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    depth = near + (far - near) * (0.35 + 0.3 * torch.sin(3 * xx) * torch.cos(2 * yy) + 0.05 * torch.rand(H, W, generator=g))
    out = {}
    for k, (div, D, half) in enumerate(((4, 48, 1.0), (2, 32, 0.5), (1, 8, 0.15))):
        d = depth[::div, ::div]
        lo, hi = (d - half * (far - near) / 2).clamp(min=near), (d + half * (far - near) / 2).clamp(max=far)
        t = torch.linspace(0, 1, D).view(D, 1, 1)
        out["stage%d" % (k + 1)] = {"depth_values": (lo[None] * (1 - t) + hi[None] * t)[None].contiguous().to(dev),
                                    "volume_feature_no_ref": scene["vols"][k]}
    return out


def live_path_batch(scene, outputs, n_rays, n_samples=90, seed=0, chunk_idx=None):
    """Inputs of one `rendering()` call shaped as the reference's live path builds them (train.py:147-163: `build_rays`,
    2000 rays x 90 cascade-guided samples; train.py:254-272: `build_rays_test`, chunks of 1024 pixels x 90): rays through
    random pixels (chunk_idx None) or through chunk `chunk_idx` of the image, per-ray cascade ranges from `outputs`, sorted +
    jittered depths, world points and the four normalised coordinate sets.  Returns a dict of device tensors + `pose_ref`."""
    from . import ops
    from .utils import utils as U
    dev = scene["confidence"].device
    H, W = scene["H"], scene["W"]
    torch.manual_seed(seed)
    if chunk_idx is None:
        xs, ys = random_pixels(n_rays, H, W, seed=seed)
        rays_d, _, _ = ops.ray_gen(scene["K"], scene["c2w"], xs=xs.to(dev), ys=ys.to(dev))
        pix = torch.stack([ys, xs]).long().to(dev)
    else:
        rays_d, _, pixf = ops.ray_gen(scene["K"], scene["c2w"], H=H, W=W, grid_start=chunk_idx * n_rays, n=n_rays, device=dev, want_pix=True)
        pix = pixf.long()
    rays_o = scene["c2w"][:3, 3].to(dev)
    ranges = U._stage_ranges(outputs, pix)
    t_rand = torch.rand(n_rays, n_samples, device=dev)
    z, pts = ops.sample_cascade(ranges, n_samples, t_rand, rays_o, rays_d)
    near_far = U._near_far_dict(ranges, n_samples, scene["near"], scene["far"])
    inv_scale = torch.tensor([W - 1, H - 1])
    ndc = U.get_ndc_coordinate(scene["w2cs"][0], scene["intrinsics"][0], pts, inv_scale, near_far)
    pose_ref = {"w2cs": scene["w2cs"].to(dev).clone(), "intrinsics": scene["intrinsics"].to(dev).clone()}
    return dict(rays_pts=pts, rays_ndc=ndc, depth_candidates=z, rays_dir=rays_d, ranges=ranges, pose_ref=pose_ref)
