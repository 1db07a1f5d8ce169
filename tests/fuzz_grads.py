"""Differential fuzzer of the render pass's BACKWARD (ucnerf_render_fused_bwd) against autograd through the CPU oracle.

    python tests/fuzz_grads.py [--cases 40] [--seed 0]

The cases are tests/fuzz_render.py's (random view counts, image / volume sizes, cameras, ragged ray counts; capped at 12 000 samples so that the oracle's
autograd stays at seconds per case).  The scalar sum(rgb r3) + sum(depth r1), r random, is differentiated
  * by torch autograd through oracle.rendering (the restatement of network/renderer.py:215-255 -- what the reference's autograd does), and
  * by the training forward + ucnerf_render_fused_bwd, in both training precisions ("f32": exact forward; "bf16x3": split-bf16 forward keeping 24-bit
    activations), default backward mode (the gradient chain), with the sources channel-major and -- every other case -- channel-last in place,
and compared per tensor: the 36 network tensors (30 with gradients, 6 the reference never reaches: exactly zero), the three volumes, the image
features, the confidence map.

Comparing gradients of a relu network element by element needs one more ingredient.  Pre-activations carry float32 noise (here ~1e-5 of a layer's
rms: a trunk layer is (W h + b) * b_d, a product of two sums that cancel), so every ray has a few units whose argument is within that noise of zero; there
the derivative is one-sided, and which side an evaluation lands on is decided by its rounding order -- which the oracle shares with torch-CPU, not with a
GPU.  A flipped unit changes nothing in the forward (the value is ~0 on either side) and one sample's contribution to the gradient; on a ray with
leverage that is visible (seed 0 case 52: one unit of one sample of one ray, argument +8.6e-6 in float64, moved the batch gradient by 1e-2 of its
maximum).  So the comparison is made twice:
  * RAW: device against the oracle as it is -- reported (elements outside the bar), not required;
  * SIDES GIVEN: the device's kept activations (the training forward's own record, read back from the backward workspace: trunk h0..h5, the two view
    branches, the density) say on which side of every relu the device was; the oracle is re-run with exactly those sides (relu(x) := x * side) and the
    device is held to an element-wise bar against THAT, with no outlier allowance: the tests' |got - want| <= 3e-4 max|want| + 3e-3 |want| per tensor
    for "f32"; 2e-3 / 1e-2 for "bf16x3", whose forward and kept activations are good to 2^-16;
  * and the sides themselves are checked: wherever the device's side differs from the float64 oracle's, the float64 argument must be within noise of
    zero -- |x| <= 64 x max(|float32 - float64 oracle| on that element, 3e-8 of the row's rms) for "f32", 128 times that for "bf16x3" (unit roundoff
    2^-17 against 2^-24); the worst |x| / rms met is reported.
A tensor for which the float32 oracle itself (its distance from the float64 oracle scaled by the same factor for "bf16x3") misses the bar -- a sum that
cancels: the one-element gradient of a head's bias on a batch of two rays -- is counted, not failed.
Test infrastructure (imports oracle/): lives under tests/.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_render as FR                                                         # noqa: E402
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


BAR = {"f32": (3e-4, 3e-3), "bf16x3": (2e-3, 1e-2)}        # per element: |got - want| <= a max|want| + b |want|
COARSER = {"f32": 1, "bf16x3": 128}                        # the arithmetic's unit roundoff in units of float32's (2^-24; split-bf16: 2^-17)


def outside(got, want, prec="f32", scale=0.0):
    """(number of elements outside the element-wise bar, worst |error| / max|want|).  scale: a floor for max|want| -- a bias gradient is the plain sum
    of what its weight gradient sums with activations as factors, and cancels where that does not: it is held to the larger of the two maxima."""
    got, want = got.detach().cpu().reshape(want.shape), want.detach()
    m = max(want.abs().max().item(), float(scale))
    err = (got - want.to(got.dtype)).abs()
    a, b = BAR[prec]
    out = err > (a * m + 1e-8 + b * want.abs())
    return int(out.sum()), float(err.max() / max(m, 1e-30))


NAMES5 = ("vol1", "vol2", "vol3", "img_feat", "confidence")


def oracle_grads(sd, sc, rays_d, z, r3, r1, white, dtype=torch.float32, sides=None):
    """{tensor name: gradient} of sum(rgb r3) + sum(depth r1) by autograd through the oracle, computed in `dtype` (sides: see Relus); the oracle's
    coordinates; its relu arguments."""
    cv = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t              # noqa: E731
    p = {k: cv(v).clone().requires_grad_(True) for k, v in sd.items()}
    leaves = dict(vols=[cv(v).clone().requires_grad_(True) for v in sc["vols"]], img_feat=cv(sc["img_feat"]).clone().requires_grad_(True),
                  confidence=cv(sc["confidence"]).clone().requires_grad_(True))
    scene = {k: ([cv(x) for x in v] if isinstance(v, list) else cv(v)) for k, v in sc.items()}
    scene.update(leaves)
    with Relus(sides) as rl:
        want, pts, ndc = FR.oracle_pass(p, scene, cv(rays_d), cv(z), white)
    ((want["rgb"] * cv(r3)).sum() + (want["depth"] * cv(r1)).sum()).backward()
    out = {n_: (t.grad if t.grad is not None else torch.zeros_like(t)) for n_, t in zip(NAMES5, leaves["vols"] + [leaves["img_feat"], leaves["confidence"]])}
    out.update({k: v.grad for k, v in p.items()})                                                # (None: the reference's autograd never reaches it)
    return out, pts.detach(), {k: v.detach() for k, v in ndc.items()}, rl.args


RELU_UNITS = (128, 128, 128, 128, 128, 128, 64, 64, 1)      # the oracle's relu calls in order: trunk 0..5, views branch, confidence branch, density


class Relus:
    """While active, records every relu argument of the oracle's forward and, with `sides` (one bool tensor per call), replaces relu(x) by x * side."""

    def __init__(self, sides=None):
        self.sides, self.args = sides, []

    def __enter__(self):
        self.orig = torch.relu
        torch.relu = self._relu
        return self

    def __exit__(self, *exc):
        torch.relu = self.orig

    def _relu(self, x):
        i = len(self.args)
        self.args.append(x.detach())
        if x.shape[-1] != RELU_UNITS[i]:
            raise RuntimeError("fuzz_grads: relu call %d of the oracle has %d units, expected %d" % (i, x.shape[-1], RELU_UNITS[i]))
        if self.sides is None:
            return self.orig(x)
        return x * self.sides[i].reshape(x.shape).to(x.dtype)


class DevicePass:
    def __init__(self, sdv, sd, flat, sc, white, prec):
        from uc_nerf_amd import ops
        self.ops, self.sd, self.flat = ops, sd, flat
        dev = flat.device
        src = ops.GatherSources(sdv["vols"], sdv["confidence"], sdv["imgs"], sdv["img_feat"], sdv["w2cs"][1:], sdv["intrinsics"][1:])
        pw = ops.PackedWeights.get(src.V, 0, dev, prec)
        self.rp = ops.RenderPass(src, pw, pw.pack(flat), sdv["c2w"][:3, 3], sdv["w2cs"][0], sdv["intrinsics"][0], sdv["w2cs"][0], sc["near"], sc["far"], white)
        self.rp.repack_sources()
        pw32 = ops.PackedWeights.get(src.V, 0, dev, "f32")
        self.f32w = (pw32, pw32.pack(flat))

    def grads(self, rays_d, z, r3, r1, coords=None):
        """{tensor name: gradient}, and the side of every relu the training forward was on ([n,S,units] bool, in the oracle's call order)."""
        from uc_nerf_amd import _lib as L
        ops = self.ops
        dev = self.flat.device
        rd, zz = rays_d.to(dev).contiguous(), z.to(dev).contiguous()
        n, S = zz.shape
        out = self.rp(rd, zz, keep=("raw", "feats"), coords=coords)
        # the kept activation sets sit at the head of the MLP part of the backward workspace: [forward-style carve][g_raw 4M][g_feats M F][sets ...]
        # (csrc/render.hip carve_bwd_render; ops.mlp_fwd_train reads the same layout)
        M, V = n * S, self.rp.src.V
        a4 = lambda k: (k + 3) // 4 * 4                                                            # noqa: E731
        off = L.lib().ucnerf_render_workspace_floats(n, S, V) + a4(4 * M) + a4(M * (24 + 12 * V + 1))
        per = a4(M * 128)
        ws = self.rp._bwd_ws
        kept = {}
        for i, name in enumerate(ops.KEPT_SETS):
            chunk = ws[off + i * per: off + (i + 1) * per]
            kept[name] = (ops.decode_p24(chunk, M) if ops._backward_mode == 0 else chunk[: M * 128].view(M, 128)).cpu()
        sides = [(kept["h%d" % k] > 0).view(n, S, 128) for k in range(6)]
        sides += [(kept["vc"][:, :64] > 0).view(n, S, 64), (kept["vc"][:, 64:] > 0).view(n, S, 64), (out["raw"][..., 3:4] > 0).cpu()]
        res = self.rp.backward(rd, zz, out, r3.to(dev).contiguous(), r1.to(dev).contiguous(), self.flat, coords=coords, f32_weights=self.f32w)
        got = dict(zip(NAMES5, (res[1], res[2], res[3], res[5], res[4])))
        o = 0
        for k, v in self.sd.items():
            got[k] = res[0][o:o + v.numel()].view(v.shape)
            o += v.numel()
        return got, sides


def run(cases=40, seed=0, verbose=True):
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, scene_to
    dev = torch.device(DEV)
    failures, rows = [], []
    t0 = time.time()
    for c in range(cases):
        s = seed * 100000 + c
        case = FR.make_case(s)
        sc, V, white = case["scene"], case["V"], case["white"]
        n, S = case["n"], case["S"]
        if n * S > 12000:
            n = max(1, 12000 // S)
        xs, ys, z = case["xs"][:n], case["ys"][:n], case["z"][:n].contiguous()
        sd = init_ucnerf_state_dict(seed=s % 7, n_src=V - 1, sigma_scale=case["sigma"], sigma_bias=0.02)
        _, rays_d, _ = O.get_rays_mvs_pixels(xs, ys, sc["K"], sc["c2w"])
        g = torch.Generator().manual_seed(s + 1)
        r3, r1 = torch.randn(n, 3, generator=g), torch.randn(n, generator=g)
        want, pts, ndc, x32 = oracle_grads(sd, sc, rays_d, z, r3, r1, white)
        with Relus() as r64, torch.no_grad():
            cv = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t         # noqa: E731
            FR.oracle_pass({k: cv(v) for k, v in sd.items()}, {k: ([cv(x) for x in v] if isinstance(v, list) else cv(v)) for k, v in sc.items()},
                           cv(rays_d), cv(z), white)
        x64 = r64.args
        cl = bool(c % 2)
        sdv = scene_to(sc, dev)
        if cl:
            sdv = dict(sdv, vols=[v.contiguous(memory_format=torch.channels_last_3d) for v in sdv["vols"]],
                       imgs=sdv["imgs"][0].contiguous(memory_format=torch.channels_last).unsqueeze(0),
                       img_feat=sdv["img_feat"][:, 0].contiguous(memory_format=torch.channels_last).unsqueeze(1))
        flat = flat_params_of(sd).to(dev)
        coords = None
        if (c // 2) % 2:
            coords = {"pts": pts.to(dev), "stage1": ndc["stage1"].to(dev), "stage2": ndc["stage2"].to(dev), "stage3": ndc["stage3"].to(dev), "ndc": ndc["ndc"].to(dev)}
        bad = []
        row = dict(case=c, seed=s, V=V, n=n, S=S, channel_last=cl, given_coords=coords is not None, sigma=case["sigma"], cancelling_tensors=0)
        w64 = None
        for prec in ("f32", "bf16x3"):
            dp = DevicePass(sdv, sd, flat, sc, white, prec)
            got, sides = dp.grads(rays_d, z, r3, r1, coords)
            # RAW: against the oracle as it is (reported)
            raw_out = [outside(got[k], w_, prec) for k, w_ in want.items() if w_ is not None]
            row[prec + "_raw_outside"], row[prec + "_raw_worst"] = sum(o_[0] for o_ in raw_out), max(o_[1] for o_ in raw_out)
            # the sides: where they differ from the float64 oracle's, the argument must be within noise of zero
            flips, far_flips, worst_ratio = 0, 0, 0.0
            for sd_, a32, a64 in zip(sides, x32, x64):
                a64 = a64.reshape(sd_.shape)
                diff = sd_ != (a64 > 0)
                if diff.any():
                    # the scale of a row of units: its rms (the single density unit: the rms over the batch)
                    rms = (a64.pow(2).mean(-1, keepdim=True).sqrt() if a64.shape[-1] > 1 else a64.pow(2).mean().sqrt().reshape(1, 1, 1)).expand_as(a64)
                    tol = 64 * COARSER[prec] * torch.maximum((a32.reshape(sd_.shape).double() - a64).abs(), 3e-8 * rms)
                    flips += int(diff.sum())
                    far_flips += int((diff & (a64.abs() > tol)).sum())
                    worst_ratio = max(worst_ratio, float((a64.abs() / rms)[diff].max()))
            row[prec + "_flips"], row[prec + "_far_flips"], row[prec + "_flip_worst"] = flips, far_flips, worst_ratio
            b = []
            if far_flips:
                b.append("%d relu sides differ from the float64 oracle away from a tie (worst |x| / rms %.3g)" % (far_flips, worst_ratio))
            # SIDES GIVEN: the bar, against the oracle on the device's sides
            want_s = oracle_grads(sd, sc, rays_d, z, r3, r1, white, sides=sides)[0] if flips else want
            n_out, worst, elements = 0, 0.0, 0
            for k, w_ in want_s.items():
                if w_ is None:
                    if torch.count_nonzero(got[k]):
                        b.append(k + ": gradient where the reference has none")
                    continue
                wk = k[:-4] + "weight" if k.endswith(".bias") else None
                floor_ = float(want_s[wk].abs().max()) if wk is not None and want_s.get(wk) is not None else 0.0
                o_ = outside(got[k], w_, prec, floor_)
                elements += w_.numel()
                if o_[0]:
                    if w64 is None:
                        w64 = {}
                    key = (prec, k)
                    if key not in w64:
                        w64[key] = oracle_grads(sd, sc, rays_d, z, r3, r1, white, torch.float64, sides=sides)[0][k]
                    # a sum that cancels: the float32 oracle's own distance from exact arithmetic, scaled to this precision's roundoff, misses the bar too
                    w64k = w64[key].float()
                    if outside(w64k + COARSER[prec] * (w_ - w64k), w64k, prec)[0]:
                        row["cancelling_tensors"] += 1
                        continue
                    b.append("%s: %d of %d outside the bar with the sides given, worst %.3g of max|g|" % (k, o_[0], w_.numel(), o_[1]))
                    n_out += o_[0]
                worst = max(worst, o_[1])
            row[prec + "_outside"], row[prec + "_worst"], row[prec + "_elements"] = n_out, worst, elements
            bad += [prec + " " + x for x in b]
        row["bad"] = bad
        rows.append(row)
        if bad:
            failures.append(row)
        if verbose:
            print("case %3d seed %d V=%d n=%d S=%d cl=%d coords=%d sigma=%g |  f32: raw %d outside (worst %.1e), %d flips, sides given %d outside (worst %.1e)"
                  " |  bf16x3: raw %d outside (worst %.1e), %d flips (|x|/rms <= %.1e), sides given %d outside (worst %.1e)  %s" % (
                      c, s, V, n, S, cl, coords is not None, case["sigma"], row["f32_raw_outside"], row["f32_raw_worst"], row["f32_flips"], row["f32_outside"],
                      row["f32_worst"], row["bf16x3_raw_outside"], row["bf16x3_raw_worst"], row["bf16x3_flips"], row["bf16x3_flip_worst"], row["bf16x3_outside"],
                      row["bf16x3_worst"], "FAIL " + "; ".join(bad[:4]) if bad else "ok"), flush=True)
    tot = lambda key: {p_: sum(r[p_ + key] for r in rows) for p_ in ("f32", "bf16x3")}             # noqa: E731
    return dict(cases=cases, seed=seed, failures=len(failures), seconds=round(time.time() - t0, 1), elements=sum(r["f32_elements"] for r in rows),
                rays=sum(r["n"] for r in rows), samples=sum(r["n"] * r["S"] for r in rows), raw_outside=tot("_raw_outside"), flips=tot("_flips"),
                far_flips=tot("_far_flips"), outside=tot("_outside"), worst={p_: max(r[p_ + "_worst"] for r in rows) for p_ in ("f32", "bf16x3")},
                raw_worst={p_: max(r[p_ + "_raw_worst"] for r in rows) for p_ in ("f32", "bf16x3")},
                cancelling_tensors=sum(r["cancelling_tensors"] for r in rows), rows=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    s = run(a.cases, a.seed)
    print("fuzz_grads: %d cases (%d rays, %d samples, %d gradient elements per precision), %d failures, %.0f s" % (
        s["cases"], s["rays"], s["samples"], s["elements"], s["failures"], s["seconds"]))
    print("  against the oracle as it is: elements outside the bar %s, worst error / max|g| %s" % (s["raw_outside"], s["raw_worst"]))
    print("  relu sides that differ from the float64 oracle: %s, of them away from a tie: %s" % (s["flips"], s["far_flips"]))
    print("  against the oracle on the device's sides: elements outside the bar %s, worst error / max|g| %s; %d tensors where the float32 oracle itself "
          "misses the bar against float64" % (s["outside"], s["worst"], s["cancelling_tensors"]))
    return 1 if s["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
