"""The 64 + 128 render step at a given ray count (the strong-scaling shard: RAYS=512), for rocprofv3 traces and A/B timing.
  RAYS=512 REPACK=1 STEPS=300 PREC=bf16x3_fused python3 scripts/time_strong.py
REPACK=1 rebuilds the channel-last source copies inside every step (the headline rule), 0 hoists it (sources constant across batches)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to  # noqa: E402

dev = torch.device("cuda:0")
n = int(os.environ.get("RAYS", "512"))
repack = os.environ.get("REPACK", "1") == "1"
steps = int(os.environ.get("STEPS", "300"))
scene_cpu = make_scene(seed=0)
scene = scene_to(scene_cpu, dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
if os.environ.get("ZC", "0") == "1":       # sources handed over channel-last: read in place, no repack to hoist
    from uc_nerf_amd import ops
    cl = ops.ChannelLastSources.from_reference_layout(scene["vols"], scene["imgs"], scene["img_feat"])
    scene = dict(scene, vols=cl.vols, imgs=cl.imgs, img_feat=cl.img_feat)
r = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), 64, 128, precision=os.environ.get("PREC", "bf16x3_fused"))
xs, ys = random_pixels(4096, scene_cpu["H"], scene_cpu["W"], seed=0)
xs, ys = xs[:n].to(dev).contiguous(), ys[:n].to(dev).contiguous()
noise = torch.rand(4096, 64, generator=torch.Generator().manual_seed(100))[:n].to(dev).contiguous()
fold = int(os.environ.get("FOLD", "1"))      # bit 0: coarse compositing + re-sampling in one launch; bit 1: rays generated inside the coarse MLP launch; 0: the launch structure of rounds 1-3
r.fold_launches, r.fold_rays = bool(fold & 1), bool(fold & 2)
r.pass_.repack_sources()
for _ in range(max(150, steps // 2)):
    r.render(xs, ys, perturb=1.0, noise=noise, repack=repack)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    r.render(xs, ys, perturb=1.0, noise=noise, repack=repack)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print("render step %d rays x (64+128), repack=%d fold=%d: %.4f ms = %.3f M rays/s" % (n, repack, fold, ms, n / ms / 1e3))
