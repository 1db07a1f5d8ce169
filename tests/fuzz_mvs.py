"""Differential fuzzer of the step in front of the ray marcher: ucnerf_cost_volume and ucnerf_depth_regress, forward and
backward, against oracle/mvs_oracle.py (network/mvs_models.py:599-646, utils/utils.py:1105-1172).

    python tests/fuzz_mvs.py [--cases 120] [--seed 0]

Per case: 1..8 source views, 1..32 channels, maps 2..96 x 2..120, 1..48 hypotheses, padding 0..3, source cameras from nearly identical to far apart (so that
warps leave the source maps: the border clamp and the in-view count), hypothesis maps per pixel.  The cost volume samples NEAREST: a warped coordinate within
rounding of x.5 may pick the neighbour (the reference's projection is a BLAS product whose summation order is no part of its contract), so -- as in
tests/test_hip_parity.py -- up to 2e-3 of the voxels may differ and the rest must agree to 1e-5 + 1e-5 relative; the count likewise; the gradient into the
feature maps is compared on the maps the differing voxels do not touch -- in practice: within 2e-6 |g|max + 1e-4 relative on all but a 5e-3 share of elements.
Depth regression: probabilities 1e-6 + 1e-5, depth 2e-6 + 2e-6, confidence (a window at floor(E[d])) up to 5e-3 of the pixels differing, gradient likewise.
Test infrastructure (imports oracle/): lives under tests/.
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
from oracle import mvs_oracle as M                                               # noqa: E402

DEV = "cuda:0"


def share_outside(got, want, atol, rtol, reduce_dim=None):
    bad = (got.detach().cpu() - want).abs() > atol + rtol * want.abs()
    if reduce_dim is not None:
        bad = bad.any(dim=reduce_dim)
    return bad, float(bad.float().mean()) if bad.numel() else 0.0


def run(cases=120, seed=0, verbose=True):
    from uc_nerf_amd import ops
    dev = torch.device(DEV)
    failures = []
    worst = {"variance_share": 0.0, "count_share": 0.0, "g_feats_share": 0.0, "conf_share": 0.0, "g_logits_share": 0.0}
    t0 = time.time()
    for c in range(cases):
        s = seed * 100000 + c
        rng = np.random.RandomState(s)
        g = torch.Generator().manual_seed(s)
        V, C = int(rng.randint(1, 9)), int(rng.choice([1, 3, 8, 8, 8, 16, 32]))
        H, W = int(rng.randint(2, 97)), int(rng.randint(2, 121))
        D, pad = int(rng.randint(1, 49)), int(rng.choice([0, 0, 1, 3]))
        if C * D * (H + 2 * pad) * (W + 2 * pad) > 6e6:
            D = max(1, int(6e6 // (C * (H + 2 * pad) * (W + 2 * pad))))
        Hp, Wp = H + 2 * pad, W + 2 * pad
        f = float(rng.uniform(0.5, 1.5)) * W
        K4 = torch.eye(4)
        K4[0, 0], K4[1, 1], K4[0, 2], K4[1, 2] = f, f * float(rng.uniform(0.9, 1.1)), W / 2.0, H / 2.0
        amp = float(rng.choice([0.01, 0.05, 0.3]))
        w2c = torch.stack([torch.eye(4)] + [FR.pose(rng, amp, amp) for _ in range(V)])
        affine = K4 @ w2c
        proj = (affine[1:] @ torch.inverse(affine)[0:1])[:, :3].contiguous()
        feats = torch.randn(V, C, H, W, generator=g)
        near = float(rng.uniform(0.5, 2.0))
        dv = (near + torch.linspace(0, float(rng.uniform(0.5, 4.0)), D).view(D, 1, 1) + 0.05 * torch.rand(D, Hp, Wp, generator=g)).contiguous()
        bad = []
        # ---- cost volume, forward
        want, cnt_want = M.cost_volume_variance(feats, proj, dv, pad)
        fd = feats.to(dev).requires_grad_(True)
        var_c, cnt = ops.cost_volume(fd, proj.to(dev), dv.to(dev), pad=pad, want_count=True)
        var = ops.cost_volume(fd, proj.to(dev), dv.to(dev), pad=pad)                  # (the differentiable form)
        if not torch.equal(var.detach(), var_c):
            bad.append("variance with and without the count differ")
        vox, sh = share_outside(var, want, 1e-5, 1e-5, reduce_dim=0)
        worst["variance_share"] = max(worst["variance_share"], sh)
        # (a share, so small volumes get an absolute allowance of two voxels)
        if vox.sum() > max(2, 2e-3 * vox.numel()):
            bad.append("variance: %.4f of the voxels differ" % sh)
        cb, csh = share_outside(cnt, cnt_want, 0, 0)
        worst["count_share"] = max(worst["count_share"], csh)
        if cb.sum() > max(2, 2e-3 * cb.numel()):
            bad.append("count: %.4f differ" % csh)
        # ---- cost volume, backward
        r = torch.randn(want.shape, generator=g)
        (var * r.to(dev)).sum().backward()
        fo = feats.clone().requires_grad_(True)
        (M.cost_volume_variance(fo, proj, dv, pad)[0] * r).sum().backward()
        gs = fo.grad.abs().max().item()
        gb, gsh = share_outside(fd.grad, fo.grad, 2e-6 * gs + 1e-12, 1e-4)
        worst["g_feats_share"] = max(worst["g_feats_share"], gsh)
        # a voxel that picked the neighbouring pixel moves C gradient elements from one pixel to the other
        if gb.sum() > max(4 * C, 2 * C * int(vox.sum()) + 5e-3 * gb.numel()):
            bad.append("g_feats: %.4f of the elements differ (%d voxels differed)" % (gsh, int(vox.sum())))
        # ---- depth regression, forward + backward
        Dr = min(D, 128)
        logits = torch.randn(Dr, Hp, Wp, generator=g) * float(rng.choice([0.3, 1.0, 4.0]))
        init = torch.randn(Dr, Hp, Wp, generator=g) if rng.rand() < 0.3 else None
        p_w, d_w, c_w = M.depth_regress(logits, dv[:Dr], init, pad)
        x = logits.to(dev).requires_grad_(True)
        prob, depth, conf = ops.depth_regress(x, dv[:Dr].to(dev), None if init is None else init.to(dev), pad=pad)
        if share_outside(prob, p_w, 1e-6, 1e-5)[1] > 0:
            bad.append("prob_volume differs (max %.3g)" % float((prob.cpu() - p_w).abs().max()))
        if share_outside(depth, d_w, 2e-6, 2e-6)[1] > 0:
            bad.append("depth differs (max %.3g)" % float((depth.detach().cpu() - d_w).abs().max()))
        cfb, cfs = share_outside(conf, c_w, 2e-6, 0)
        worst["conf_share"] = max(worst["conf_share"], cfs)
        if cfb.sum() > max(2, 5e-3 * cfb.numel()):
            bad.append("confidence: %.4f of the pixels differ" % cfs)
        r1, r2 = torch.randn(d_w.shape, generator=g), torch.randn(c_w.shape, generator=g)
        ((depth * r1.to(dev)).sum() + (conf * r2.to(dev)).sum()).backward()
        lo = logits.clone().requires_grad_(True)
        _, d_o, c_o = M.depth_regress(lo, dv[:Dr], init, pad)
        ((d_o * r1).sum() + (c_o * r2).sum()).backward()
        lb, ls = share_outside(x.grad, lo.grad, 2e-6 + 2e-6 * lo.grad.abs().max().item(), 1e-4, reduce_dim=0)
        worst["g_logits_share"] = max(worst["g_logits_share"], ls)
        if lb.sum() > max(2, int(cfb.sum()) + 5e-3 * lb.numel()):
            bad.append("g_logits: %.4f of the pixels differ" % ls)
        tag = "case %d (seed %d): V=%d C=%d %dx%d D=%d pad=%d amp=%g" % (c, s, V, C, H, W, D, pad, amp)
        if bad:
            failures.append(tag + ": " + "; ".join(bad))
        if verbose:
            print(tag + ("  FAIL " + "; ".join(bad) if bad else "  ok (voxels differing %.1e, gradient elements %.1e, confidence pixels %.1e)" % (sh, gsh, cfs)), flush=True)
    return dict(cases=cases, seed=seed, failures=failures, worst_shares=worst, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    s = run(a.cases, a.seed)
    print("fuzz_mvs: %d cases, %d failures, largest differing shares %s, %.0f s" % (s["cases"], len(s["failures"]), {k: round(v, 5) for k, v in s["worst_shares"].items()}, s["seconds"]))
    for b in s["failures"]:
        print("  " + b)
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
