"""Differential fuzzer of the stand-alone network launch (ucnerf_mlp_fwd: positional encodings + the uncertainty-conditioned MLP, network/models.py:50-54,
138-184 and utils/run_nerf_helpers.py:52-53) against the oracle, and of the stand-alone encoders (ucnerf_embed).

    python tests/fuzz_mlp.py [--cases 200] [--seed 0]

Per case: 1..7 source views, 1..6000 samples (ragged: tiles of 32 never fill), both positional-encoding layouts (the live network's and the
nerf-pytorch helpers'), directions per ray or per sample, features row-major or in the kernel's tile layout, coordinates up to +-4 (arguments of the
highest frequency ~2000 rad) and features up to a few units, density-head scale 0.02 / 0.1 / 1, every arithmetic: exact f32, split-bf16, plain bf16.
Bars: raw rgb 2e-5 / density 2e-5 max(1, |sigma|max) for f32 (tests/test_hip_pipeline.py) PLUS 32x the float32 oracle's own distance from the float64
oracle on that input (feature rows of a few units drive the trunk's activations to 1e3 and beyond -- every layer multiplies by the depth-bias net's
output -- and no float32 evaluation is good to 2e-5 there); split-bf16: 16x the fixed part + 128 x 8 x that distance (2^-17 against 2^-24 arithmetic; the distance is the
maximum over a case's few thousand outputs of ONE float32 evaluation's error: a noisy yardstick, hence the generous factors -- 1 500 cases put three
over an 8x / 4x version of these bars, by at most 2.2x);
plain bf16 is reported only (it is outside the parity bar by design).  Encoders: 2e-6 absolute.
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
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


def run(cases=200, seed=0, verbose=True):
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    dev = torch.device(DEV)
    failures, samples = [], 0
    worst = {"f32": 0.0, "bf16x3": 0.0, "bf16": 0.0, "embed": 0.0}
    t0 = time.time()
    for c in range(cases):
        s = seed * 100000 + c
        rng = np.random.RandomState(s)
        g = torch.Generator().manual_seed(s)
        n_src = int(rng.randint(1, 8))
        S = int(rng.choice([1, 3, 30, 64, 90]))
        n = int(rng.choice([1, 2, 5, 11, 33, 64]))
        if rng.rand() < 0.3:
            n = int(rng.randint(1, max(2, 6000 // S)))
        m = n * S
        layout = int(rng.randint(0, 2))
        per_sample = bool(rng.rand() < 0.4)
        F = 24 + 12 * n_src + 1
        amp = float(rng.choice([1.0, 1.0, 4.0]))
        pts = (torch.rand(n, S, 3, generator=g) * 2 - 1) * amp
        dirs = torch.randn(n, S, 3, generator=g) if per_sample else torch.randn(n, 3, generator=g)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        feats = torch.randn(n, S, F, generator=g) * float(rng.choice([0.3, 1.0, 2.0]))
        feats[..., -1] = torch.rand(n, S, generator=g)                       # the confidence column
        if rng.rand() < 0.3:
            feats[:, S // 2:, :] = 0                                        # samples outside every source: all-zero feature rows
        sig = float(rng.choice([0.02, 0.1, 1.0]))
        sd = init_ucnerf_state_dict(seed=s % 11, n_src=n_src, sigma_scale=sig, sigma_bias=0.02)
        lay = "live" if layout == 0 else "interleaved"
        want = O.run_network_mvs(sd, pts, dirs, feats, n_src=n_src, layout=lay)
        smax = max(1.0, float(want[..., 3].abs().max()))
        # how well conditioned is this input?  The trunk multiplies by the depth-bias net's output in every layer ((W h + b) * b_d), so feature rows of a
        # few units drive the activations to 1e3 and beyond, and a float32 evaluation is then good to 1e-4, not 1e-7: the float32 oracle's own distance
        # from the float64 oracle measures it, and the bars scale with it
        w64 = O.run_network_mvs({k_: v.double() for k_, v in sd.items()}, pts.double(), dirs.double(), feats.double(), n_src=n_src, layout=lay)
        noise = max(float((want[..., :3].double() - w64[..., :3]).abs().max()), float((want[..., 3].double() - w64[..., 3]).abs().max()) / smax)
        worst["oracle_f32_vs_f64"] = max(worst.get("oracle_f32_vs_f64", 0.0), noise)
        flat = flat_params_of(sd).to(dev)
        bad = []
        tiled = bool(rng.rand() < 0.5)
        fd = feats.reshape(m, F).to(dev)
        if tiled:                                                            # [ceil(m/32)][F][32]
            pad = (m + 31) // 32 * 32
            ft = torch.zeros(pad, F, device=dev)
            ft[:m] = fd
            fd_in = ft.view(-1, 32, F).permute(0, 2, 1).contiguous().reshape(-1)
        else:
            fd_in = fd
        for prec in ("f32", "bf16x3", "bf16"):
            pw = ops.PackedWeights.get(n_src, layout, dev, prec)
            raw = ops.mlp_fwd(pw, pw.pack(flat), pts.reshape(m, 3).to(dev), dirs.reshape(-1, 3).to(dev), fd_in, S, feats_tiled=tiled).cpu().view(n, S, 4)
            e_rgb = float((raw[..., :3] - want[..., :3]).abs().max())
            e_sig = float((raw[..., 3] - want[..., 3]).abs().max()) / smax
            e = max(e_rgb, e_sig)
            worst[prec] = max(worst[prec], e)
            bar = {"f32": 2e-5 + 32 * noise, "bf16x3": 16 * 2e-5 + 128 * 8 * noise, "bf16": float("inf")}[prec]
            if not e <= bar:
                bad.append("%s: rgb %.3g, sigma %.3g of max(1,|sigma|max)" % (prec, e_rgb, e_sig))
        # the encoders on their own
        for nf in (10, 4):
            got = ops.embed(pts.reshape(m, 3).to(dev), nf, layout).cpu()
            ref = (O.embed_live if layout == 0 else O.embed_interleaved)(pts.reshape(m, 3), nf)
            e = float((got - ref).abs().max())
            worst["embed"] = max(worst["embed"], e)
            if not e <= 2e-6:
                bad.append("embed n_freqs=%d: %.3g" % (nf, e))
        samples += m
        tag = "case %d (seed %d): n_src=%d n=%d S=%d layout=%d dirs_per_sample=%d tiled=%d amp=%g sigma=%g" % (c, s, n_src, n, S, layout, per_sample, tiled, amp, sig)
        if bad:
            failures.append(tag + ": " + "; ".join(bad))
        if verbose:
            print(tag + ("  FAIL " + "; ".join(bad) if bad else "  ok"), flush=True)
    return dict(cases=cases, seed=seed, samples=samples, failures=failures, worst=worst, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    s = run(a.cases, a.seed)
    print("fuzz_mlp: %d cases, %d samples, %d failures, worst errors %s, %.0f s" % (s["cases"], s["samples"], len(s["failures"]),
                                                                                  {k: float("%.3g" % v) for k, v in s["worst"].items()}, s["seconds"]))
    for b in s["failures"]:
        print("  " + b)
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
