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
