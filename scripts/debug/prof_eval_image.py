"""Host-side cost of the reference's evaluation loop (train.py:251-275) through the drop-in modules: 80 x (build_rays_test -> rendering -> .cpu() x 2) per
image; cProfile over two images + the wall split per call (GPU box).  IMAGES=3 python scripts/debug/prof_eval_image.py"""
import cProfile
import os
import pstats
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, scene_to  # noqa: E402

dev = torch.device("cuda:0")
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
a, kw, net, grad_vars, outputs = bench.dropin_setup(scene, sd, dev)
ctx = types.SimpleNamespace(dev=dev)
for rep in range(2):
    r = bench.bench_eval_image(ctx, scene, a, kw, outputs, 0.0, images=int(os.environ.get("IMAGES", "3")))
    print({k: v for k, v in r.items() if k != "note"})
pr = cProfile.Profile()
pr.enable()
bench.bench_eval_image(ctx, scene, a, kw, outputs, 0.0, images=1)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
# a trivial launch + sync after the GPU went idle (what a drained stream costs the next chunk)
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    time.sleep(0.0002)
    t0 = time.perf_counter()
    x.add_(1.0)
    y = x[:4].cpu()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("add_ + 16-byte .cpu() after 0.2 ms idle: median %.1f us, p90 %.1f us" % (ts[100] * 1e6, ts[180] * 1e6))
# Is the spread between images Python's cyclic garbage collector (a full collection walks every object torch has imported)?
import gc
print("gc counts", gc.get_count(), "stats", gc.get_stats())
for mode in ("gc as it is", "gc.freeze()", "gc.disable()"):
    if mode == "gc.freeze()":
        gc.collect(); gc.freeze()
    if mode == "gc.disable()":
        gc.disable()
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        bench.bench_eval_image(ctx, scene, a, kw, outputs, 0.0, images=1)
        ts.append((time.perf_counter() - t0) / 4 * 1e3)          # (2 warm-up images + 1 timed + 1 split image per call)
    print("%-14s ms per image over 6 x 4 images: %s" % (mode, " ".join("%.1f" % t for t in ts)), "gen2 collections so far:", gc.get_stats()[2]["collections"])
