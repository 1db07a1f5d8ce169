"""Host-side cost of the rendering() drop-in on the evaluation chunk (1024 x 90, bf16x3): cProfile over 300 calls (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import uc_nerf_amd  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, live_path_batch, make_scene, scene_to  # noqa: E402

dev = torch.device("cuda:0")
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
a, kw, net, grad_vars, outputs = bench.dropin_setup(scene, sd, dev)
uc_nerf_amd.set_inference_precision(os.environ.get("PREC", "bf16x3"))
ev = live_path_batch(scene, outputs, 1024, 90, seed=3, chunk_idx=7)


def step():
    with torch.no_grad():
        return bench.dropin_call(a, kw, scene, outputs, ev)


for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per call: issue %.1f us, complete %.1f us" % (t_issue / 300 * 1e6, t_all / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
