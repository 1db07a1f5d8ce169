"""Times one training-style step of the fused render path: forward (keeps raw + row-major features) and backward
(parameter, volume, image-feature and confidence gradients) on 1024 rays x 128 depths of the synthetic scene."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to  # noqa: E402

dev = torch.device("cuda:0")
n, S = int(os.environ.get("RAYS", "1024")), int(os.environ.get("SAMPLES", "128"))
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
flat = flat_params_of(sd).to(dev)
zc = os.environ.get("ZC", "0") == "1"      # sources handed over channel-last (ops.ChannelLastSources): no repack, channel-last source gradients
if zc:
    cl = ops.ChannelLastSources.from_reference_layout(scene["vols"], scene["imgs"], scene["img_feat"])
    scene = dict(scene, vols=cl.vols, imgs=cl.imgs, img_feat=cl.img_feat)
r = CoarseFineRenderer(scene, flat, 64, 128)          # f32: the training path
xs, ys = random_pixels(n, scene["H"], scene["W"], seed=0)
rays_d, _, _ = ops.ray_gen(r.K_host, r.c2w_host, xs=xs.to(dev), ys=ys.to(dev))
z, _ = ops.sample_stratified(None, S, n=n, near=r.near_host, far=r.far_host, device=dev)
g_rgb, g_depth = torch.randn(n, 3, device=dev), torch.randn(n, device=dev)


def step():
    r.pass_.repack_sources()                  # the MVS net produces new volumes / feature maps every training step
    out = r.pass_(rays_d, z, keep=("raw", "feats"))
    return r.pass_.backward(rays_d, z, out, g_rgb, g_depth, flat)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
print("train-style step (fwd + bwd), %d rays x %d samples, channel-last sources=%d: %.3f ms = %.0f rays/s" % (n, S, zc, ms, n / ms * 1e3))
