"""Compositing and re-sampling in the tail of the gather-fused launch (small passes) against the separate launches: bit identity and time (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uc_nerf_amd import _lib as _L  # noqa: E402
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to

dev = torch.device("cuda:0")
scene_cpu = make_scene(seed=0)
scene = scene_to(scene_cpu, dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
r = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), 64, 128, precision="bf16x3_fused")
r.pass_.repack_sources()
for n in (512, 500, 700):
    xs, ys = random_pixels(4096, scene_cpu["H"], scene_cpu["W"], seed=0)
    xs, ys = xs[:n].to(dev).contiguous(), ys[:n].to(dev).contiguous()
    noise = torch.rand(4096, 64, generator=torch.Generator().manual_seed(100))[:n].to(dev).contiguous()
    outs = {}
    for tail in ("0", "1"):
        _L.lib().ucnerf_set_fused_tail(int(tail))
        r.fold_rays = False if tail == "0" else None
        o = r.render(xs, ys, perturb=1.0, noise=noise, repack=False)
        torch.cuda.synchronize()
        outs[tail] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
    bad = [k for k in outs["0"] if not torch.equal(outs["0"][k], outs["1"][k])]
    print("%5d rays: %d outputs compared (%s), differing: %s" % (n, len(outs["0"]), ", ".join(sorted(outs["0"])), bad or "none"))
    for tail, fr in (("0", False), ("1", False), ("1", None), ("0", False), ("1", False), ("1", None)):
        _L.lib().ucnerf_set_fused_tail(int(tail))
        r.fold_rays = fr
        for _ in range(100): r.render(xs, ys, perturb=1.0, noise=noise, repack=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): r.render(xs, ys, perturb=1.0, noise=noise, repack=False)
        torch.cuda.synchronize()
        print("      tail=%s fold_rays=%s: %.4f ms per step" % (tail, fr, (time.perf_counter() - t0) / 300 * 1e3))
