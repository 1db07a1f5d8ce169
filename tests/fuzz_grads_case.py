"""One case of tests/fuzz_grads.py taken apart ray by ray: which rays carry the gradient differences, in which backward mode.

    python tests/fuzz_grads_case.py --case 52 [--precision f32]
(test infrastructure: imports oracle/)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_render as FR                                                         # noqa: E402
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", type=int, required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", default="f32")
    a = ap.parse_args()
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, scene_to
    dev = torch.device(DEV)
    s = a.seed * 100000 + a.case
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
    sdv = scene_to(sc, dev)
    flat = flat_params_of(sd).to(dev)
    key = "nerf.pts_linears.0.weight"
    off = 0
    for k, v in sd.items():
        if k == key:
            break
        off += v.numel()
    m = sd[key].numel()

    def oracle_grad(sel):
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        want, pts, ndc = FR.oracle_pass(p, sc, rays_d[sel], z[sel], white)
        ((want["rgb"] * r3[sel]).sum() + (want["depth"] * r1[sel]).sum()).backward()
        return p[key].grad.reshape(-1), want

    def device_grad(sel, mode):
        ops.set_backward_mode(mode)
        src = ops.GatherSources(sdv["vols"], sdv["confidence"], sdv["imgs"], sdv["img_feat"], sdv["w2cs"][1:], sdv["intrinsics"][1:])
        pw = ops.PackedWeights.get(src.V, 0, dev, a.precision)
        rp = ops.RenderPass(src, pw, pw.pack(flat), sdv["c2w"][:3, 3], sdv["w2cs"][0], sdv["intrinsics"][0], sdv["w2cs"][0], sc["near"], sc["far"], white)
        rp.repack_sources()
        rd, zz = rays_d[sel].to(dev).contiguous(), z[sel].to(dev).contiguous()
        out = rp(rd, zz, keep=("raw", "feats"))
        pw32 = ops.PackedWeights.get(src.V, 0, dev, "f32")
        gf = rp.backward(rd, zz, out, r3[sel].to(dev).contiguous(), r1[sel].to(dev).contiguous(), flat, f32_weights=(pw32, pw32.pack(flat)))[0]
        return gf[off:off + m].cpu(), out

    every = torch.arange(n)
    want, _ = oracle_grad(every)
    for mode in ("chain", "layerwise"):
        got, _ = device_grad(every, mode)
        print("all %d rays, backward mode %-9s: max |diff| / max |g| = %.3e" % (n, mode, float((got - want).abs().max() / want.abs().max())))
    gmax = float(want.abs().max())
    worst = []
    for i in range(n):
        sel = torch.tensor([i])
        w_i, o = oracle_grad(sel)
        c_i, d = device_grad(sel, "chain")
        l_i, _ = device_grad(sel, "layerwise")
        worst.append((float((c_i - w_i).abs().max()) / gmax, float((l_i - w_i).abs().max()) / gmax, i, float(w_i.abs().max()) / gmax,
                      float((d["raw"].cpu() - o["raw"]).abs().max())))
    worst.sort(reverse=True)
    print("rays by their share of the difference (chain err / max|g| of the batch, layers err, ray, the ray's own max|g| share, max raw diff):")
    for w_ in worst[:8]:
        print("  %.3e  %.3e  ray %d  own %.3e  raw diff %.2e" % w_)
    ops.set_backward_mode("chain")


if __name__ == "__main__":
    main()
