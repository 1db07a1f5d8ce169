"""Differential fuzzer of the evaluation loop's ray builder (ucnerf_build_rays_test: rays, cascade ranges, sorted + jittered depths, world points and
the four normalised coordinate sets in ONE launch) against the oracle's composition of the reference's steps (utils/utils.py:248-271, 600-739).

    python tests/fuzz_builders.py [--cases 200] [--seed 0]

Per case: image 8..256 x 8..320 (multiples of 4), samples per ray 3..192 (multiples of 3, as N_samples // 3 * 3), 1..48 depth hypotheses per stage with
random per-pixel ranges (near < far, overlapping or nested or disjoint between stages), a random chunk of the pixel grid (start anywhere, 1..1500 rays,
the ragged last chunk included), random target camera / reference view / intrinsics, the reference's jitter draws given.
Bars (those of tests/test_hip_parity.py): depths 1e-6 abs + 1e-6 rel, points 1e-5 + 1e-5, normalised coordinates 2e-5 + 2e-5, directions 1e-6;
origin exact.  Test infrastructure (imports oracle/): lives under tests/.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_render as FR                                                         # noqa: E402
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


def worst(got, want, atol, rtol, exact=None):
    """max over elements of |got - want| / (atol + rtol |want| + 32 |want - exact|): <= 1 passes.  `exact`: the same quantity from the oracle run in
    float64 -- where the float32 oracle itself is far from it (a sample next to the reference camera's plane: the projection divides by a depth that is
    the small difference of large terms) no float32 evaluation can be held to the plain bar."""
    got, want = got.detach().cpu().double(), want.double()
    if not want.numel():
        return 0.0
    slack = 32 * (want - exact.double()).abs() if exact is not None else torch.zeros_like(want)
    slack = torch.where(torch.isfinite(slack), slack, torch.zeros_like(slack))
    fin = torch.isfinite(want)
    # a stage with ONE depth hypothesis has near == far: the reference divides by zero there (utils/utils.py:333-347) and the kernel must give the
    # same infinities and NaNs in the same places
    same_kind = (torch.isnan(want) & torch.isnan(got)) | (torch.isinf(want) & (got == want))
    if not bool((fin | same_kind).all()) or not bool(torch.isfinite(got[fin]).all()):
        return float("inf")
    return float(((got[fin] - want[fin]).abs() / (atol + rtol * want[fin].abs() + slack[fin])).max()) if fin.any() else 0.0


def run(cases=200, seed=0, verbose=True):
    from uc_nerf_amd import ops
    dev = torch.device(DEV)
    bad, rays = [], 0
    worst_all = {}
    t0 = time.time()
    for c in range(cases):
        s = seed * 100000 + c
        rng = np.random.RandomState(s)
        g = torch.Generator().manual_seed(s)
        H, W = 4 * int(rng.randint(2, 65)), 4 * int(rng.randint(2, 81))
        S = 3 * int(rng.choice([1, 2, 5, 10, 11, 21, 30, 32, 43, 64]))
        n_pix = H * W
        n = int(min(rng.choice([1, 2, 31, 64, 100, 257, 1024, 1500]), n_pix))
        start = int(rng.randint(0, n_pix - n + 1)) if rng.rand() < 0.7 else n_pix - n          # (the last, ragged chunk of the image)
        near, far = float(rng.uniform(0.3, 2.0)), 0.0
        far = near + float(rng.uniform(0.5, 6.0))
        K = torch.tensor([[float(rng.uniform(0.6, 1.6)) * W, 0, W / 2.0 + rng.uniform(-3, 3)], [0, float(rng.uniform(0.6, 1.6)) * W, H / 2.0 + rng.uniform(-3, 3)],
                          [0, 0, 1]], dtype=torch.float32)
        K_ref = K.clone()
        K_ref[0, 0] *= float(rng.uniform(0.9, 1.1))
        amp = 0.5 if rng.rand() < 0.3 else 0.05
        c2w, w2c_ref = FR.pose(rng, amp, amp), FR.pose(rng, amp, amp)
        dvs = []
        for k, div in enumerate((4, 2, 1)):
            D = int(rng.randint(1, 49))
            lo = near + (far - near) * torch.rand(1, 1, H // div, W // div, generator=g) * 0.6
            hi = lo + (far - near) * (0.02 + 0.4 * torch.rand(1, 1, H // div, W // div, generator=g))
            t = torch.linspace(0, 1, D).view(1, D, 1, 1) if D > 1 else torch.zeros(1, 1, 1, 1)
            dvs.append((lo * (1 - t) + hi * t).contiguous())
        t_rand = torch.rand(n, S, generator=g)
        nf_ref = torch.tensor([near, far])
        # ---- oracle composition
        ys, xs = O.pixel_grid(H, W, torch.float32)
        ro, rd, pix = O.get_rays_mvs_pixels(xs[start:start + n], ys[start:start + n], K, c2w)
        p = pix.long()
        nf = {}
        for k, div in (("1", 4), ("2", 2), ("3", 1)):
            dv = dvs[int(k) - 1]
            pr, pc = torch.div(p[0], div, rounding_mode="trunc"), torch.div(p[1], div, rounding_mode="trunc")
            nf["near_" + k], nf["far_" + k] = dv[0, 0, pr, pc].unsqueeze(1), dv[0, -1, pr, pc].unsqueeze(1)
        z = O.cascade_depth_candidates(nf["near_1"], nf["far_1"], nf["near_2"], nf["far_2"], nf["near_3"], nf["far_3"], S, t_rand)
        pts = ro.reshape(1, 1, 3) + z.unsqueeze(-1) * rd.unsqueeze(1)
        nfe = {k: v.expand(-1, S).unsqueeze(-1) for k, v in nf.items()}
        nfe["near"], nfe["far"] = nf_ref[0], nf_ref[1]
        ndc = O.get_ndc_coordinate(w2c_ref, K_ref, pts, torch.tensor([W - 1, H - 1]), nfe)
        ndc64 = O.get_ndc_coordinate(w2c_ref.double(), K_ref.double(), pts.double(), torch.tensor([W - 1, H - 1], dtype=torch.float64),
                                     {k_: (v.double() if torch.is_tensor(v) else v) for k_, v in nfe.items()})
        # ---- device: one launch
        got = ops.build_rays_test(H, W, start, n, S, K.to(dev), c2w.to(dev), w2c_ref.to(dev), K_ref.to(dev), nf_ref.to(dev), [d.to(dev) for d in dvs],
                                  t_rand.to(dev), want_ranges=True)
        ranges = torch.cat([nf[k] for k in ("near_1", "far_1", "near_2", "far_2", "near_3", "far_3")], -1)
        w = {"rays_d": worst(got["rays_d"], rd, 1e-6, 1e-6), "z": worst(got["z"], z, 1e-6, 1e-6), "pts": worst(got["pts"], pts, 1e-5, 1e-5),
             "ranges": 0.0 if torch.equal(got["ranges"].cpu(), ranges) else float("inf"), "rays_o": 0.0 if torch.equal(got["rays_o"].cpu(), ro) else float("inf")}
        for k in ("stage1", "stage2", "stage3", "ndc"):
            w[k] = worst(got[k], ndc[k], 2e-5, 2e-5, exact=ndc64[k])
        for k, v in w.items():
            worst_all[k] = max(worst_all.get(k, 0.0), v)
        rays += n
        fails = ["%s %.3g" % (k, v) for k, v in w.items() if not v <= 1.0]
        tag = "case %d (seed %d): %dx%d S=%d start=%d n=%d D=%s" % (c, s, H, W, S, start, n, [d.shape[1] for d in dvs])
        if fails:
            bad.append(tag + ": " + ", ".join(fails))
        if verbose:
            print(tag + ("  FAIL " + ", ".join(fails) if fails else "  ok (worst z %.2f pts %.2f ndc %.2f of the bar)" % (w["z"], w["pts"], w["ndc"])), flush=True)
    return dict(cases=cases, seed=seed, rays=rays, failures=bad, worst_fraction_of_bar=worst_all, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    s = run(a.cases, a.seed)
    print("fuzz_builders: %d cases, %d rays, %d failures, worst error as a fraction of its bar %s, %.0f s" % (
        s["cases"], s["rays"], len(s["failures"]), {k: round(v, 3) for k, v in s["worst_fraction_of_bar"].items()}, s["seconds"]))
    for b in s["failures"]:
        print("  " + b)
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
