"""Host-side cost of the rendering() TRAINING step (train.py:147-188: forward + loss + backward + Adam) at RAYS rays x 90 samples: issue time
against completion time, and a cProfile of the issuing thread (GPU box).  RAYS=250 is the per-GPU share of the 2000-ray batch at 8 GPUs."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, live_path_batch, make_scene, scene_to  # noqa: E402

dev = torch.device("cuda:0")
n = int(os.environ.get("RAYS", "250"))
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
a, kw, net, grad_vars, outputs = bench.dropin_setup(scene, sd, dev)
tr = live_path_batch(scene, outputs, n, 90, seed=4)
vols = [v.detach().clone().requires_grad_(True) for v in scene["vols"]]
img_feat = scene["img_feat"].detach().clone().requires_grad_(True)
conf = scene["confidence"].detach().clone().requires_grad_(True)
target = torch.rand(n, 3, device=dev)
if os.environ.get("FLAT_ADAM", "0") == "1":
    from uc_nerf_amd.flat import FlatAdam
    opt = FlatAdam(net, lr=5e-4, betas=(0.9, 0.999))
else:
    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))


def step():
    opt.zero_grad(set_to_none=True)
    for t in vols + [img_feat, conf]:
        t.grad = None
    rgb, depth = bench.dropin_call(a, kw, scene, outputs, tr, confidence=conf, vols=vols, img_feat=img_feat)
    loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
    loss.backward()
    opt.step()
    return loss


for _ in range(30):
    step()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("%d rays: per step: issue %.1f us, complete %.1f us" % (n, t_issue / K * 1e6, t_all / K * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
