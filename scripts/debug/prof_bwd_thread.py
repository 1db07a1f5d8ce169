"""cProfile of the Python that runs INSIDE autograd's worker thread during loss.backward() of the rendering() training step (the main thread's
profile shows it only as run_backward's wait): _FusedRender.backward and what it calls.  RAYS=250 (GPU box)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from uc_nerf_amd import dropin
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
opt = FlatAdam(net, lr=5e-4, betas=(0.9, 0.999))
pr = cProfile.Profile()
orig = dropin._FusedRender.backward
acc = [0.0, 0]
def timed(ctx, *g):
    t0 = time.perf_counter()
    pr.enable()
    try:
        return orig(ctx, *g)
    finally:
        pr.disable()
        acc[0] += time.perf_counter() - t0; acc[1] += 1
dropin._FusedRender.backward = staticmethod(timed)

def step():
    opt.zero_grad(set_to_none=True)
    for t in vols + [img_feat, conf]:
        t.grad = None
    rgb, depth = bench.dropin_call(a, kw, scene, outputs, tr, confidence=conf, vols=vols, img_feat=img_feat)
    loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
    loss.backward()
    opt.step()

for _ in range(30):
    step()
torch.cuda.synchronize()
acc[0], acc[1] = 0.0, 0
pr = cProfile.Profile()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("%d rays: %.1f us per step; _FusedRender.backward (host, in the autograd thread, under cProfile): %.1f us per call" % (n, (time.perf_counter() - t0) / 200 * 1e6, acc[0] / acc[1] * 1e6))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
