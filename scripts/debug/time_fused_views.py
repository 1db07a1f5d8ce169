"""Step time of the 64 + 128 hierarchy on a scene with V source views (bench.py is fixed at six): V=4 RAYS=4096 PREC=bf16x3_fused python scripts/debug/time_fused_views.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to  # noqa: E402

dev = torch.device("cuda:0")
V, n = int(os.environ.get("V", "4")), int(os.environ.get("RAYS", "4096"))
prec = os.environ.get("PREC", "bf16x3_fused")
scene = make_scene(seed=0, V=V + 1)
sd = init_ucnerf_state_dict(seed=0, n_src=V, sigma_scale=0.05, sigma_bias=0.05)
r = CoarseFineRenderer(scene_to(scene, dev), flat_params_of(sd).to(dev), 64, 128, precision=prec)
xs, ys = random_pixels(n, 256, 320, seed=0)
xs, ys = xs.to(dev), ys.to(dev)
noise = torch.rand(n, 64, device=dev)
for _ in range(150):
    out = r.render(xs, ys, perturb=1.0, noise=noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    out = r.render(xs, ys, perturb=1.0, noise=noise)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print("V=%d %s rays %d: %.4f ms/step = %.3f M rays/s (lib %s)" % (V, prec, n, dt * 1e3, n / dt / 1e6, os.environ.get("UCNERF_LIB", "default")))
