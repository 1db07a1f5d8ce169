"""Stateful fuzzer of the coarse + fine pipeline (uc_nerf_amd.pipeline.CoarseFineRenderer: the benchmark's step, data/ray_utils.py:199-224): ONE renderer
object per case driven through random renders and state changes, every render compared BIT FOR BIT with

  (a) a freshly built renderer with the same settings, and
  (b) a freshly built renderer on the PLAIN route -- no folded launches, no rays generated in-kernel, no tail route, fine pass re-evaluating every
      depth, sources channel-major -- which DESIGN.md states every other route to be bit-identical to.

    python tests/fuzz_pipeline.py [--cases 40] [--steps 30] [--seed 0]

Per case: random scene size, 1..6 source views, (n_coarse, n_fine) from (8,16) .. (64,128), precision bf16x3_fused / bf16x3 / f32, white background.
Steps: render (1..4096 rays, jitter on / off, deterministic or random draws, reuse_coarse on / off), set_params (new weights), sources written in place
(+ repack), route knobs flipped (fold_launches, fold_rays, tail route), sources switched between channel-major and channel-last layouts.
No oracle: this checks state and the route equivalences; tests/fuzz_render.py checks the arithmetic.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEV = "cuda:0"
KEYS = ("rgb", "depth", "acc", "weights", "z_fine", "z_samples")


def run(cases=40, steps=30, seed=0, verbose=True, only=None):
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
    dev = torch.device(DEV)
    failures, renders = [], 0
    t0 = time.time()
    for c in (range(cases) if only is None else [only]):
        s = seed * 1000 + c
        rng = np.random.RandomState(s)
        V = int(rng.randint(2, 8))
        H, W = 4 * int(rng.randint(8, 33)), 4 * int(rng.randint(8, 41))
        nc, nf = [(8, 16), (16, 32), (32, 64), (64, 128), (64, 64), (33, 45)][int(rng.randint(0, 6))]
        prec = str(rng.choice(["bf16x3_fused", "bf16x3_fused", "bf16x3", "f32"]))
        white = bool(rng.rand() < 0.3)
        base = scene_to(make_scene(seed=s, H=H, W=W, V=V, small_volumes=bool(rng.rand() < 0.5)), dev)
        cl = [False] * 5
        flat = flat_params_of(init_ucnerf_state_dict(seed=s, n_src=V - 1, sigma_scale=0.05, sigma_bias=0.05)).to(dev)

        def laid_out(scene, cl_):
            v = [t.contiguous(memory_format=torch.channels_last_3d) if k else t.contiguous() for t, k in zip(scene["vols"], cl_[:3])]
            f = scene["img_feat"][:, 0].contiguous(memory_format=torch.channels_last).unsqueeze(1) if cl_[3] else scene["img_feat"].contiguous()
            i = scene["imgs"][0].contiguous(memory_format=torch.channels_last).unsqueeze(0) if cl_[4] else scene["imgs"].contiguous()
            return dict(scene, vols=v, img_feat=f, imgs=i)

        def clone_scene(scene):
            return {k: ([t.clone() for t in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v)) for k, v in scene.items()}

        scene = laid_out(clone_scene(base), cl)
        r = CoarseFineRenderer(scene, flat, nc, nf, white_bkgd=white, precision=prec)
        knobs = dict(fold_launches=True, fold_rays=None, tail=1)
        log = []
        try:
            for k in range(steps):
                op = rng.choice(["render", "render", "render", "params", "sources", "knobs", "layout"])
                if op == "render":
                    n = int(rng.choice([1, 31, 64, 100, 257, 512, 700, 1024, 2048, 4096]))
                    xs, ys = random_pixels(n, H, W, seed=s * 100 + k)
                    xs, ys = xs.to(dev), ys.to(dev)
                    g = torch.Generator().manual_seed(s * 100 + k)
                    perturb = float(rng.choice([0.0, 1.0]))
                    noise = torch.rand(n, nc, generator=g).to(dev) if perturb else None
                    u = torch.rand(n, nf, generator=g).to(dev) if rng.rand() < 0.4 else None
                    reuse = bool(rng.rand() < 0.3)
                    L.lib().ucnerf_set_fused_tail(knobs["tail"])
                    r.fold_launches, r.fold_rays = knobs["fold_launches"], knobs["fold_rays"]
                    got = r.render(xs, ys, perturb=perturb, noise=noise, u=u, reuse_coarse=reuse)
                    # (a) fresh, same settings
                    ra = CoarseFineRenderer(laid_out(clone_scene(scene), cl), flat.clone(), nc, nf, white_bkgd=white, precision=prec)
                    ra.fold_launches, ra.fold_rays = knobs["fold_launches"], knobs["fold_rays"]
                    wa = ra.render(xs, ys, perturb=perturb, noise=noise, u=u, reuse_coarse=reuse)
                    # (b) fresh, the plain route
                    L.lib().ucnerf_set_fused_tail(0)
                    rb = CoarseFineRenderer(laid_out(clone_scene(scene), [False] * 5), flat.clone(), nc, nf, white_bkgd=white, precision=prec)
                    rb.fold_launches, rb.fold_rays = False, False
                    wb = rb.render(xs, ys, perturb=perturb, noise=noise, u=u, reuse_coarse=False)
                    L.lib().ucnerf_set_fused_tail(knobs["tail"])
                    renders += 1
                    log.append("render n=%d perturb=%g u=%d reuse=%d" % (n, perturb, u is not None, reuse))
                    for key in KEYS:
                        if key in got and key in wa and not torch.equal(got[key], wa[key]):
                            raise AssertionError("%s differs from a fresh renderer's (same settings): max %.3g" % (key, float((got[key] - wa[key]).abs().max())))
                    for key in ("rgb", "depth", "z_fine", "z_samples") + (() if reuse else ("acc", "weights")):
                        if key in got and key in wb and not torch.equal(got[key], wb[key]):
                            raise AssertionError("%s differs from the plain route's: max %.3g" % (key, float((got[key] - wb[key]).abs().max())))
                elif op == "params":
                    flat = flat * (1.0 + 0.02 * float(rng.randn()))
                    r.set_params(flat)
                    log.append("params")
                elif op == "sources":
                    i = int(rng.randint(0, 6))
                    with torch.no_grad():
                        if i < 3:
                            scene["vols"][i].mul_(0.9)
                        elif i == 3:
                            scene["img_feat"].mul_(0.9)
                        elif i == 4:
                            scene["imgs"].mul_(0.95)
                        else:
                            scene["confidence"].mul_(0.95)
                    log.append("sources %d written in place" % i)
                elif op == "layout":
                    cl = [bool(rng.rand() < 0.5) for _ in range(5)]
                    scene = laid_out(clone_scene(scene), cl)
                    r = CoarseFineRenderer(scene, flat, nc, nf, white_bkgd=white, precision=prec)     # (new source tensors: a new renderer, as a script would build)
                    log.append("layout %s" % "".join(str(int(x)) for x in cl))
                else:
                    knobs["fold_launches"] = bool(rng.rand() < 0.7)
                    knobs["fold_rays"] = [None, True, False][int(rng.randint(0, 3))]
                    knobs["tail"] = int(rng.rand() < 0.7)
                    log.append("knobs %s" % knobs)
        except AssertionError as e:
            failures.append("case %d (seed %d) V=%d %dx%d %d+%d %s: %s; steps: %s" % (c, s, V, H, W, nc, nf, prec, e, log[-8:] if only is None else log))
        finally:
            L.lib().ucnerf_set_fused_tail(1)
        if verbose:
            print("case %d (seed %d) V=%d %dx%d %d+%d %s: %d steps %s" % (c, s, V, H, W, nc, nf, prec, len(log),
                                                                        "FAIL" if failures and failures[-1].startswith("case %d " % c) else "ok"), flush=True)
    return dict(cases=cases, steps=steps, seed=seed, renders=renders, failures=failures, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--case", type=int, default=None)
    a = ap.parse_args()
    s = run(a.cases, a.steps, a.seed, only=a.case)
    print("fuzz_pipeline: %d cases x %d steps, %d renders checked against fresh renderers and the plain route, %d failures, %.0f s" % (
        s["cases"], s["steps"], s["renders"], len(s["failures"]), s["seconds"]))
    for f in s["failures"]:
        print("  " + f)
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
