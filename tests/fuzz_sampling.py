"""Differential fuzzer of the hierarchical re-sampling launch (ucnerf_sample_pdf) against the CPU oracle AND against torch-CPU itself.

    python tests/fuzz_sampling.py [--cases 300] [--seed 0]

Per case: 1..300 rays, 2..700 bin edges, 1..300 draws, weight rows drawn from eight families (uniform, cubed, one spike, sparse with exact zeros,
all zero, tiny, huge dynamic range, a few negative), draws from five (uniform, sorted linspace shared by all rays, exact cdf entries, 0 and
1 - 2^-24 at the ends, duplicated), optionally merged with a second sorted list that shares values with the samples (ties), optionally in the
`from_coarse` form the fused compositing launch uses (mid-point bins of z, w[1:-1]).

Checked, all bit for bit (data/ray_utils.py:98-141,216-219):
  cdf, inds (int64), samples  ==  oracle.sample_pdf            (the restatement of torch-CPU's accumulation order)
  oracle.sample_pdf           ==  the same lines run by torch on the CPU of this box, whenever the row sums take ATen's vectorised path
                                  identically (they do on every AVX2/AVX-512 build: the fixtures pin lanes = 8) -- reported, not required
  with negative weights (a cdf that is not monotone; cannot occur in the reference) inds / samples  ==  torch-CPU's searchsorted, whose
                                  binary search the device repeats step for step (the oracle counts entries <= u instead: equal only when monotone)
  z_sorted                    ==  torch.sort(cat(samples, z_merge))
  merge_rank                  is a permutation that places cat(samples, z_merge) into z_sorted
Test infrastructure (imports oracle/): lives under tests/; tests/test_hip_round5.py runs a fixed set of its cases.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


def weights_of(rng, g, n, L, family):
    w = torch.rand(n, L, generator=g)
    if family == 1:
        w = w ** 3
    elif family == 2:
        w = w * 1e-3
        w[torch.arange(n), torch.randint(0, L, (n,), generator=g)] = 5.0
    elif family == 3:
        w = torch.where(torch.rand(n, L, generator=g) < 0.7, torch.zeros(()), w)
    elif family == 4:
        w = torch.zeros(n, L)
    elif family == 5:
        w = w * 1e-7
    elif family == 6:
        w = w * 1e-6
        w[:, int(rng.randint(0, L))] = 3e5 * (1 + torch.rand(n, generator=g))
    elif family == 7:
        w[::2, int(rng.randint(0, L))] = -0.25
    return w.contiguous()


def torch_cpu_sample_pdf(bins, weights, u):
    """data/ray_utils.py:98-141 as torch runs it on this CPU."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    g0, g1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = g1 - g0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - g0) / denom * (b1 - b0), inds, cdf


def run(cases=300, seed=0, verbose=True):
    from uc_nerf_amd import ops
    bad, rows, torch_agrees = [], 0, 0
    t0 = time.time()
    for c in range(cases):
        s = seed * 100000 + c
        rng = np.random.RandomState(s)
        g = torch.Generator().manual_seed(s)
        n = int(rng.choice([1, 2, 3, 17, 64, 100, 300]))
        Lb = int(rng.choice([2, 3, 5, 8, 9, 16, 33, 63, 64, 65, 128, 129, 191, 192, 256, 513, 700]))          # bin edges
        M = int(rng.choice([1, 2, 7, 32, 33, 64, 128, 129, 192, 300]))
        fam, ufam = int(rng.randint(0, 8)), int(rng.randint(0, 5))
        from_coarse = bool(rng.rand() < 0.3) and Lb >= 3
        merge = from_coarse or bool(rng.rand() < 0.5)
        if from_coarse:
            S = Lb + 1                                                    # z [n,S] -> S-1 mid-points = bin edges, w[1:-1] = S-2 weights
            z = torch.sort(0.5 + 4.0 * torch.rand(n, S, generator=g), -1)[0]
            if rng.rand() < 0.3:
                z[:, S // 3] = z[:, S // 3 + 1]                           # a zero-length interval
            w_full = weights_of(rng, g, n, S, fam)
            bins = 0.5 * (z[:, :-1] + z[:, 1:])
            w = w_full[:, 1:-1].contiguous()
        else:
            bins = torch.sort(torch.rand(n, Lb, generator=g) * float(rng.choice([1.0, 100.0])), -1)[0]
            w = weights_of(rng, g, n, Lb - 1, fam)
        want_cdf = O.sample_pdf(bins, w, torch.zeros(n, 1))[2]
        shared = False
        if ufam == 0:
            u = torch.rand(n, M, generator=g)
        elif ufam == 1:
            u, shared = torch.linspace(0., 1., M), True
        elif ufam == 2:                                                   # draws that sit exactly on cdf entries
            idx = torch.randint(0, Lb, (n, M), generator=g)
            u = torch.gather(want_cdf, -1, idx).clamp(0, 1)
        elif ufam == 3:
            u = torch.rand(n, M, generator=g)
            u[:, 0] = 0.0
            u[:, -1] = 1.0 - 2.0 ** -24
        else:
            u = torch.rand(n, 1, generator=g).expand(n, M).contiguous()
        u_full = u.expand(n, M).contiguous() if shared else u
        want_s, want_i, want_c = O.sample_pdf(bins, w, u_full)
        ts, ti, tc = torch_cpu_sample_pdf(bins, w, u_full)
        torch_agrees += int(torch.equal(tc, want_c) and torch.equal(ti, want_i) and torch.equal(ts, want_s))
        if fam == 7 and torch.equal(tc, want_c):
            # a cdf that is not monotone (negative weights: cannot occur in the reference, whose weights are alpha * T >= 0): searchsorted's answer
            # is then whatever its binary search visits, and THAT is the definition -- the oracle's counting form is not.  The device runs the same
            # search (sample_pdf_device.h), so it is held to torch-CPU's own result here
            want_s, want_i = ts, ti
        tag = "case %d (seed %d): n=%d bins=%d M=%d weights %d draws %d from_coarse=%d merge=%d" % (c, s, n, Lb, M, fam, ufam, from_coarse, merge)
        zm = None
        if merge and not from_coarse:
            zm = torch.sort(torch.rand(n, int(rng.choice([1, 31, 64, 90])), generator=g) * bins.max(), -1)[0]
            k = min(zm.shape[1], M)
            if rng.rand() < 0.5:
                zm[:, :k] = torch.sort(want_s[:, :k], -1)[0]             # ties across the two lists
                zm = torch.sort(zm, -1)[0]
        if from_coarse:
            out = ops.sample_pdf(None, w_full.to(DEV), u.to(DEV), z_merge=z.to(DEV), from_coarse=True, want_cdf=True, want_rank=True)
            zm = z
        else:
            out = ops.sample_pdf(bins.to(DEV), w.to(DEV), u.to(DEV), z_merge=None if zm is None else zm.to(DEV), want_cdf=True, want_rank=zm is not None)
        errs = []
        if not torch.equal(out["cdf"].cpu(), want_c):
            errs.append("cdf")
        if not torch.equal(out["inds"].cpu(), want_i):
            errs.append("inds")
        if not torch.equal(out["samples"].cpu(), want_s):
            # NaN-aware: a NaN sample (0/0 never occurs here by construction) would compare unequal
            errs.append("samples (max diff %.3g)" % float((out["samples"].cpu() - want_s).abs().max()))
        if zm is not None:
            cat = torch.cat([want_s, zm], -1)
            if not torch.equal(out["z_sorted"].cpu(), torch.sort(cat, -1)[0]):
                errs.append("z_sorted")
            rank = out["merge_rank"].cpu().long()
            if not torch.equal(torch.sort(rank, -1)[0], torch.arange(cat.shape[1]).expand_as(rank)):
                errs.append("merge_rank is not a permutation")
            elif not torch.equal(torch.zeros_like(cat).scatter_(-1, rank, cat), out["z_sorted"].cpu()):
                errs.append("merge_rank does not place cat into z_sorted")
        rows += n
        if errs:
            bad.append(tag + ": " + ", ".join(errs))
        if verbose:
            print(tag + ("  FAIL " + ", ".join(errs) if errs else "  ok"), flush=True)
    return dict(cases=cases, seed=seed, rows=rows, failures=bad, torch_cpu_agrees_with_oracle=torch_agrees, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    s = run(a.cases, a.seed)
    print("fuzz_sampling: %d cases, %d rows, %d failures, torch-CPU == oracle in %d cases, %.0f s" % (
        s["cases"], s["rows"], len(s["failures"]), s["torch_cpu_agrees_with_oracle"], s["seconds"]))
    for b in s["failures"]:
        print("  " + b)
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
