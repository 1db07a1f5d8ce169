"""rendering() under no_grad on the 1024 x 90 evaluation chunk: host issue time against completion time (is the drop-in host-bound?).  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, live_path_batch, make_scene, scene_to
dev = torch.device("cuda:0")
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
a, kw, net, grad_vars, outputs = bench.dropin_setup(scene, sd, dev)
ev = live_path_batch(scene, outputs, 1024, 90, seed=3, chunk_idx=7)
with torch.no_grad():
    for _ in range(200): bench.dropin_call(a, kw, scene, outputs, ev)
    torch.cuda.synchronize()
    for rep in range(3):
        K = 1000
        t0 = time.perf_counter()
        for _ in range(K): bench.dropin_call(a, kw, scene, outputs, ev)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("eval chunk 1024 x 90: issue %.1f us, complete %.1f us per call" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
