"""Differential fuzzer of the render pass (ucnerf_render_fused_fwd) against the CPU oracle, over random scene shapes and cameras.

    python tests/fuzz_render.py [--cases 60] [--seed 0] [--out gpurun_out/fuzz_render.json]

Per case: a random number of source views (1..7), image size, cascade depths, ray count (1..1500, ragged), samples per ray (2..200), near/far range,
focal length, target camera / reference view / source views with random rotations and offsets (so that many samples leave the volumes and the images:
masks, borders and empty rays are exercised), random pixel set.  Compared with the oracle's render of the same rays and depths
(oracle/ucnerf_oracle.py `rendering`, the restatement of network/renderer.py:215-255):

  f32       the exact-f32 kernels, coordinates derived from (ray, depth)
  fused     the gather-fused split-bf16 kernel (the default inference route), derived coordinates
  given     the same with the oracle's coordinates handed over (what rendering() receives)
and, with no tolerance at all,
  in place  all five sources channel-last in their own allocations == the re-packed copies, bit for bit
  tail      the one-launch tail route == the two-launch route, bit for bit (where the pass is small enough to take it)

A case FAILS when rgb / acc differ from the oracle by more than 1e-4, depth by more than 1e-4 x far (the parity bar), or a bit-identity does not
hold.  This is test infrastructure and lives under tests/ for that reason (it imports oracle/); tests/test_hip_round5.py runs a fixed set of its
cases, the command line searches further.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


def rot(rng, amp):
    """Rotation by a random axis-angle vector with |angle| <= amp."""
    v = rng.uniform(-1, 1, 3)
    a = rng.uniform(0, amp)
    v = v / (np.linalg.norm(v) + 1e-9) * a
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    th = np.linalg.norm(v)
    R = np.eye(3) if th < 1e-9 else np.eye(3) + math.sin(th) / th * K + (1 - math.cos(th)) / th ** 2 * (K @ K)
    return torch.tensor(R, dtype=torch.float32)


def pose(rng, amp_r, amp_t):
    m = torch.eye(4)
    m[:3, :3] = rot(rng, amp_r)
    m[:3, 3] = torch.tensor(rng.uniform(-amp_t, amp_t, 3), dtype=torch.float32)
    return m


def make_case(seed):
    rng = np.random.RandomState(seed)
    V = int(rng.randint(2, 9))                        # views including the reference: 1..7 source views
    H, W = 4 * int(rng.randint(3, 33)), 4 * int(rng.randint(3, 41))
    D = [int(rng.randint(1, 49)), int(rng.randint(1, 33)), int(rng.randint(1, 9))]
    n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 100, 257, 500, 700, 1024, 1500]))
    S = int(rng.choice([2, 3, 8, 31, 32, 33, 45, 64, 90, 96, 128, 150, 192, 200]))
    if n * S > 120000:
        n = max(1, 120000 // S)
    near = float(rng.uniform(0.5, 2.0))
    far = near + float(rng.uniform(0.5, 5.0))
    wild = rng.rand() < 0.3                           # a third of the cases: cameras far enough apart that most samples miss something
    ar, at = (0.5, 0.5) if wild else (0.08, 0.08)
    g = torch.Generator().manual_seed(seed)
    fx = float(rng.uniform(0.6, 1.6)) * W
    K = torch.tensor([[fx, 0, W / 2.0 + rng.uniform(-3, 3)], [0, fx * rng.uniform(0.9, 1.1), H / 2.0 + rng.uniform(-3, 3)], [0, 0, 1]], dtype=torch.float32)
    w2cs = torch.stack([pose(rng, ar, at) for _ in range(V)])
    intr = K.repeat(V, 1, 1).clone()
    intr[1:, 0, 0] *= torch.tensor(rng.uniform(0.9, 1.1, V - 1), dtype=torch.float32)
    c2w = pose(rng, ar, at)
    scene = dict(K=K, c2w=c2w, w2cs=w2cs, intrinsics=intr, near=near, far=far, H=H, W=W,
                 vols=[torch.randn(1, 8, D[0], H // 4, W // 4, generator=g), torch.randn(1, 8, D[1], H // 2, W // 2, generator=g),
                       torch.randn(1, 8, D[2], H, W, generator=g)],
                 imgs=torch.rand(1, V - 1, 3, H, W, generator=g), img_feat=torch.randn(V - 1, 1, 8, H, W, generator=g),
                 confidence=torch.rand(H, W, generator=g).clamp(1e-3, 1))
    # pixels: random, off-grid for half of the cases (the evaluation loop's are integers; training jitters nothing either, but the kernels take floats)
    xs = torch.randint(0, W, (n,), generator=g).float()
    ys = torch.randint(0, H, (n,), generator=g).float()
    if rng.rand() < 0.5:
        xs, ys = xs + torch.rand(n, generator=g) - 0.5, ys + torch.rand(n, generator=g) - 0.5
    z = torch.sort(near + (far - near) * torch.rand(n, S, generator=g), -1)[0]
    if rng.rand() < 0.2:                              # repeated depths (zero-length intervals)
        z[:, S // 2:] = z[:, S // 2:S // 2 + 1]
    sig = float(rng.choice([0.02, 0.1, 1.0]))
    return dict(scene=scene, xs=xs, ys=ys, z=z, V=V, n=n, S=S, D=D, wild=bool(wild), sigma=sig, white=bool(rng.rand() < 0.25))


def oracle_pass(sd, scene, rays_d, z, white):
    N, S = z.shape
    o = scene["c2w"][:3, 3]
    pts = o.reshape(1, 1, 3) + rays_d[:, None] * z[..., None]
    H, W = scene["imgs"].shape[-2:]
    nf = O.scene_near_far(N, S, scene["near"], scene["far"])
    ndc = O.get_ndc_coordinate(scene["w2cs"][0], scene["intrinsics"][0], pts, torch.tensor([W - 1, H - 1]), nf)
    pose_ref = {"w2cs": scene["w2cs"].clone(), "intrinsics": scene["intrinsics"].clone()}
    out = O.rendering(sd, pose_ref, pts, ndc, z, rays_d, scene["vols"], scene["imgs"], scene["img_feat"], scene["confidence"],
                      scene["w2cs"].shape[0], white, full=True)
    return out, pts, ndc


def run(cases=60, seed=0, case=-1, verbose=True):
    """Runs the cases; returns the summary dict (main() writes it out)."""
    import types
    a = types.SimpleNamespace(cases=cases, seed=seed, case=case)
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, scene_to
    dev = torch.device(DEV)
    rows, failures = [], []
    t0 = time.time()
    for c in (range(a.cases) if a.case < 0 else [a.case]):
        seed = a.seed * 100000 + c
        case = make_case(seed)
        sc, z, V, white = case["scene"], case["z"], case["V"], case["white"]
        sd = init_ucnerf_state_dict(seed=seed % 7, n_src=V - 1, sigma_scale=case["sigma"], sigma_bias=0.02)
        _, rays_d, _ = O.get_rays_mvs_pixels(case["xs"], case["ys"], sc["K"], sc["c2w"])
        want, pts, ndc = oracle_pass(sd, sc, rays_d, z, white)
        far = sc["far"]
        sd_dev = scene_to(sc, dev)
        flat = flat_params_of(sd).to(dev)
        rd, zd = rays_d.to(dev), z.to(dev)
        coords = {"pts": pts.to(dev), "stage1": ndc["stage1"].to(dev), "stage2": ndc["stage2"].to(dev), "stage3": ndc["stage3"].to(dev),
                  "ndc": ndc["ndc"].to(dev)}
        row = dict(case=c, seed=seed, V=V, H=sc["H"], W=sc["W"], D=case["D"], n=case["n"], S=case["S"], wild=case["wild"], sigma=case["sigma"],
                   white=white, in_volume=float((ndc["ndc"].abs() <= 1).all(-1).float().mean()))
        bad = []

        def make_pass(prec, scene_dev):
            src = ops.GatherSources(scene_dev["vols"], scene_dev["confidence"], scene_dev["imgs"], scene_dev["img_feat"], scene_dev["w2cs"][1:],
                                    scene_dev["intrinsics"][1:])
            pw = ops.PackedWeights.get(src.V, 0, dev, prec)
            rp = ops.RenderPass(src, pw, pw.pack(flat), scene_dev["c2w"][:3, 3], scene_dev["w2cs"][0], scene_dev["intrinsics"][0], scene_dev["w2cs"][0],
                                sc["near"], sc["far"], white)
            rp.repack_sources()
            return rp

        def err(out, tag):
            e = {"rgb": (out["rgb"].cpu() - want["rgb"]).abs().max().item(), "depth": (out["depth"].cpu() - want["depth"]).abs().max().item() / far,
                 "acc": (out["acc"].cpu() - want["acc"]).abs().max().item()}
            for k, v in e.items():
                row["%s_%s" % (tag, k)] = v
                if not (v <= 1e-4):
                    bad.append("%s %s %.3g" % (tag, k, v))
            per_ray = torch.maximum(torch.maximum((out["rgb"].cpu() - want["rgb"]).abs().max(-1)[0], (out["acc"].cpu() - want["acc"]).abs()),
                                    (out["depth"].cpu() - want["depth"]).abs() / far)
            row[tag + "_rays_above_bar"] = int((per_ray > 1e-4).sum())

        rp32 = make_pass("f32", sd_dev)
        err(rp32(rd, zd), "f32")
        rpf = make_pass("bf16x3_fused", sd_dev)
        fused = rpf(rd, zd)
        err(fused, "fused")
        if a.case >= 0:
            e = (fused["rgb"].cpu() - want["rgb"]).abs().max(-1)[0]
            ea = (fused["acc"].cpu() - want["acc"]).abs()
            top = torch.argsort(torch.maximum(e, ea), descending=True)[:8]
            sig = want["raw"][..., 3]
            alpha = 1 - torch.exp(-torch.relu(sig[:, :-1]) * (z[:, 1:] - z[:, :-1]) * rays_d.norm(dim=-1, keepdim=True))
            T_last = torch.cumprod(1 - alpha + 1e-10, -1)[:, -1]
            print("rays above the bar: rgb %d, acc %d of %d;  error quantiles rgb (50/90/99/max): %s" % (
                int((e > 1e-4).sum()), int((ea > 1e-4).sum()), e.numel(), ["%.2e" % float(torch.quantile(e, q)) for q in (0.5, 0.9, 0.99, 1.0)]))
            for i in top.tolist():
                print("  ray %4d  rgb err %.2e  acc err %.2e  acc %.6f  sigma range [%.3g, %.3g]  sigma_last %.3e  T before last %.3e  max |sigma| %.3g" % (
                    i, e[i], ea[i], want["acc"][i], sig[i].min(), sig[i].max(), sig[i, -1], T_last[i], sig[i].abs().max()))
        given = rpf(rd, zd, coords=coords)
        err(given, "given")
        # channel-last sources in their own allocations, read in place
        cl = dict(sd_dev, vols=[v.contiguous(memory_format=torch.channels_last_3d) for v in sd_dev["vols"]],
                  imgs=sd_dev["imgs"][0].contiguous(memory_format=torch.channels_last).unsqueeze(0),
                  img_feat=sd_dev["img_feat"][:, 0].contiguous(memory_format=torch.channels_last).unsqueeze(1))
        rpc = make_pass("bf16x3_fused", cl)
        row["in_place"] = bool(rpc.src.zero_copy)
        if V - 1 > 1 or True:
            inpl = rpc(rd, zd)
            for k in ("rgb", "depth", "acc", "weights"):
                if not torch.equal(inpl[k], fused[k]):
                    bad.append("in place != repacked: " + k)
        # tail route on / off
        lib = L.lib()
        row["tail"] = bool(lib.ucnerf_fused_tail_fits(case["n"], case["S"]))
        lib.ucnerf_set_fused_tail(0)
        try:
            two = rpf(rd, zd)
            two_g = rpf(rd, zd, coords=coords)
        finally:
            lib.ucnerf_set_fused_tail(1)
        for k in ("rgb", "depth", "acc", "weights"):
            if not torch.equal(two[k], fused[k]):
                bad.append("tail != two launches (derived): " + k)
            if not torch.equal(two_g[k], given[k]):
                bad.append("tail != two launches (given): " + k)
        row["bad"] = bad
        row["identity_bad"] = [b for b in bad if "!=" in b]
        rows.append(row)
        if bad:
            failures.append(row)
        if verbose:
            print("case %3d seed %d V=%d %dx%d D=%s n=%d S=%d wild=%d white=%d inside=%.2f  f32 %.1e/%.1e  fused %.1e/%.1e  given %.1e/%.1e  %s" % (
                c, seed, V, sc["H"], sc["W"], case["D"], case["n"], case["S"], case["wild"], white, row["in_volume"], row["f32_rgb"], row["f32_depth"],
                row["fused_rgb"], row["fused_depth"], row["given_rgb"], row["given_depth"], "FAIL " + "; ".join(bad) if bad else "ok"), flush=True)
    summary = dict(cases=len(rows), seed=a.seed, failures=len(failures), seconds=round(time.time() - t0, 1),
                   worst={k: max(r[k] for r in rows) for k in ("f32_rgb", "f32_depth", "f32_acc", "fused_rgb", "fused_depth", "fused_acc", "given_rgb",
                                                             "given_depth", "given_acc")},
                   tail_cases=sum(r["tail"] for r in rows), rays=sum(r["n"] for r in rows),
                   rays_above_bar={t: sum(r[t + "_rays_above_bar"] for r in rows) for t in ("f32", "fused", "given")},
                   identity_failures=sum(bool(r["identity_bad"]) for r in rows), rows=rows)
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_render.json"))
    ap.add_argument("--case", type=int, default=-1, help="run this one case and print where its largest errors sit")
    a = ap.parse_args()
    summary = run(a.cases, a.seed, a.case)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(summary, f, indent=1)
    print("fuzz_render: %d cases, %d failures, %.0f s; rays above the bar %s of %d; worst %s" % (
        summary["cases"], summary["failures"], summary["seconds"], json.dumps(summary["rays_above_bar"]), summary["rays"], json.dumps(summary["worst"])))
    return 1 if summary["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
