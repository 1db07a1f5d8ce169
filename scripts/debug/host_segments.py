"""Host issue time of the rendering() training step by segment (no synchronisation inside): zero_grad, forward, loss, backward, optimizer.  RAYS=250 (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from uc_nerf_amd.flat import FlatAdam
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, live_path_batch, make_scene, scene_to

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
opt = FlatAdam(net, lr=5e-4, betas=(0.9, 0.999)) if os.environ.get("FLAT_ADAM", "1") == "1" else torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))
seg = [0.0] * 5
def step(rec):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    for t in vols + [img_feat, conf]:
        t.grad = None
    t1 = time.perf_counter()
    rgb, depth = bench.dropin_call(a, kw, scene, outputs, tr, confidence=conf, vols=vols, img_feat=img_feat)
    t2 = time.perf_counter()
    loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            seg[i] += d
for _ in range(100): step(False)
torch.cuda.synchronize()
K = 500
t0 = time.perf_counter()
for _ in range(K): step(True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("%d rays: issue %.1f us, complete %.1f us per step" % (n, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
print("  zero_grad %.1f | forward %.1f | loss %.1f | backward %.1f | optimizer %.1f us" % tuple(s / K * 1e6 for s in seg))
