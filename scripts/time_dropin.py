"""The rendering() drop-in on the live-path shapes (train.py:147-188: 2000 rays x 90 samples, forward + loss + backward + Adam;
train.py:254-272: 1024 x 90 chunk under no_grad) -- for per-kernel profiles (rocprofv3 --kernel-trace --stats) and A/B timing.
MODE=train|eval  STEPS=..  PREC=f32|bf16x3 (eval)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, live_path_batch, make_scene, scene_to  # noqa: E402

dev = torch.device("cuda:0")
mode, steps = os.environ.get("MODE", "train"), int(os.environ.get("STEPS", "40"))
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
a, kw, net, grad_vars, outputs = bench.dropin_setup(scene, sd, dev)
if mode == "eval":
    import uc_nerf_amd
    uc_nerf_amd.set_inference_precision(os.environ.get("PREC", "f32"))
    ev = live_path_batch(scene, outputs, 1024, 90, seed=3, chunk_idx=7)

    def step():
        with torch.no_grad():
            return bench.dropin_call(a, kw, scene, outputs, ev)
else:
    n = int(os.environ.get("RAYS", "2000"))
    tr = live_path_batch(scene, outputs, n, 90, seed=4)
    vols = [v.detach().clone().requires_grad_(True) for v in scene["vols"]]
    img_feat = scene["img_feat"].detach().clone().requires_grad_(True)
    conf = scene["confidence"].detach().clone().requires_grad_(True)
    target = torch.rand(n, 3, device=dev)
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

for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("%s: %.4f ms per step (%d steps)" % (mode, (time.perf_counter() - t0) / steps * 1e3, steps))
