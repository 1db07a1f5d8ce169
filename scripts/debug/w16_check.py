"""Development check of the 16-sample-tile kernel (precision "bf16x3_fused_w16"): one pass against the 32-sample fused kernel on equal depths, then the
64 + 128 step at RAYS rays with every pass / only the partly filled passes on 16-sample tiles (ROUNDS = fused_min_rounds)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to  # noqa: E402

dev = torch.device("cuda:0")
n = int(os.environ.get("RAYS", "512"))
scene_cpu = make_scene(seed=0)
scene = scene_to(scene_cpu, dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
flat = flat_params_of(sd).to(dev)
xs, ys = random_pixels(4096, scene_cpu["H"], scene_cpu["W"], seed=0)
xs, ys = xs[:n].to(dev).contiguous(), ys[:n].to(dev).contiguous()
noise = torch.rand(4096, 64, generator=torch.Generator().manual_seed(100))[:n].to(dev).contiguous()
ra = CoarseFineRenderer(scene, flat, 64, 128, precision="bf16x3_fused")
rb = CoarseFineRenderer(scene, flat, 64, 128, precision="bf16x3_fused", fused_min_rounds=float(os.environ.get("ROUNDS", "1000")))
a = ra.render(xs, ys, perturb=1.0, noise=noise)
# one pass on equal depths
pa = ra.pass_(a["rays_d"], a["z_fine"], want=("acc", "weights"), keep=("raw",))
rb.pass_small.repack_sources(force=False)
ang, _ = ops.dir_feature(a["rays_d"], scene["w2cs"][0])
pb = rb.pass_small(a["rays_d"], a["z_fine"], want=("acc", "weights"), keep=("raw",), dir_feat=ang)
torch.cuda.synchronize()
for k in ("raw", "rgb", "depth", "weights"):
    d = (pa[k] - pb[k]).abs()
    print("%-8s max |diff| %.3e  (max |value| %.3e)  nan %d" % (k, d.max().item(), pa[k].abs().max().item(), int(torch.isnan(pb[k]).sum())))
b = rb.render(xs, ys, perturb=1.0, noise=noise)
same = (a["z_fine"] - b["z_fine"]).abs().amax(-1) < 1e-4
print("step: rays with equal fine depths %.4f, max |d rgb| on them %.3e" % (same.float().mean().item(), (a["rgb"] - b["rgb"]).abs().amax(-1)[same].max().item()))
for name, r in (("32-sample tiles", ra), ("16-sample tiles where rounds < %s" % os.environ.get("ROUNDS", "1000"), rb)):
    for _ in range(200):
        r.render(xs, ys, perturb=1.0, noise=noise, repack=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        r.render(xs, ys, perturb=1.0, noise=noise, repack=False)
    torch.cuda.synchronize()
    print("%-44s %d rays: %.4f ms per step" % (name, n, (time.perf_counter() - t0) / 400 * 1e3))
