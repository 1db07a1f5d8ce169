"""One ray of a tests/fuzz_grads.py case: the backward of the compositing alone (device vs torch autograd on the oracle's raw), and where along the ray the
gradients with respect to raw differ.   python tests/fuzz_grads_ray.py --case 52 --ray 17      (test infrastructure: imports oracle/)"""
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
    ap.add_argument("--ray", type=int, required=True)
    a = ap.parse_args()
    from uc_nerf_amd import ops
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    s = a.case
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
    i = a.ray
    sel = slice(i, i + 1)
    want, _, _ = FR.oracle_pass(sd, sc, rays_d[sel], z[sel], white)
    raw = want["raw"].detach()
    for dt in (torch.float32, torch.float64):
        r = raw.to(dt).clone().requires_grad_(True)
        rgb, disp, acc, w, depth, _, var = O.raw2outputs_live(r, z[sel].to(dt), white)
        ((rgb * r3[sel].to(dt)).sum() + (depth * r1[sel].to(dt)).sum()).backward()
        if dt == torch.float32:
            g32 = r.grad.clone()
        else:
            g64 = r.grad.clone()
    gd = ops.composite_bwd(raw.to(DEV), z[sel].to(DEV), g_rgb=r3[sel].to(DEV).contiguous(), g_depth=r1[sel].to(DEV).contiguous(), white_bkgd=white).cpu()
    m = g64.abs().max().item()
    print("ray %d of case %d: S = %d, sigma in [%.3g, %.3g], acc %.6f" % (i, s, S, raw[..., 3].min(), raw[..., 3].max(), want["acc"].item()))
    print("gradient w.r.t. raw, max|g| (float64 autograd) = %.4g" % m)
    print("  device  vs float64: max diff %.3e (%.2e of max)" % ((gd - g64).abs().max().item(), (gd - g64).abs().max().item() / m))
    print("  float32 vs float64: max diff %.3e (%.2e of max)" % ((g32 - g64).abs().max().item(), (g32 - g64).abs().max().item() / m))
    d = (gd[0] - g64[0].float()).abs().max(-1)[0]
    top = torch.argsort(d, descending=True)[:6]
    alpha = 1 - torch.exp(-raw[0, :, 3].double())
    T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - alpha + 1e-10]), 0)[:-1]
    for j in top.tolist():
        print("  sample %3d: device %s   float32 %s   float64 %s   sigma %.4g  alpha %.6g  T %.4g" % (
            j, ["%.5g" % v for v in gd[0, j].tolist()], ["%.5g" % v for v in g32[0, j].tolist()], ["%.5g" % v for v in g64[0, j].tolist()],
            raw[0, j, 3], alpha[j], T[j]))


def sigma_ties(case_id, ray):
    """Arguments of the oracle's density relu along the ray (float32 and float64) next to the device's density."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, scene_to
    s = case_id
    case = FR.make_case(s)
    sc, V, white = case["scene"], case["V"], case["white"]
    n = case["n"] if case["n"] * case["S"] <= 12000 else max(1, 12000 // case["S"])
    xs, ys, z = case["xs"][:n], case["ys"][:n], case["z"][:n].contiguous()
    sd = init_ucnerf_state_dict(seed=s % 7, n_src=V - 1, sigma_scale=case["sigma"], sigma_bias=0.02)
    _, rays_d, _ = O.get_rays_mvs_pixels(xs, ys, sc["K"], sc["c2w"])
    sel = slice(ray, ray + 1)
    pre = {}
    orig = torch.relu
    for dt in (torch.float32, torch.float64):
        seen = []
        torch.relu = lambda x: (seen.append(x.detach()), orig(x))[1]
        try:
            cv = lambda t: t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t      # noqa: E731
            scene = {k: ([cv(x) for x in v] if isinstance(v, list) else cv(v)) for k, v in sc.items()}
            FR.oracle_pass({k: cv(v) for k, v in sd.items()}, scene, cv(rays_d[sel]), cv(z[sel]), white)
        finally:
            torch.relu = orig
        pre[dt] = seen[-1].reshape(-1)                     # the last relu of the forward: the density
    dev = torch.device(DEV)
    sdv = scene_to(sc, dev)
    src = ops.GatherSources(sdv["vols"], sdv["confidence"], sdv["imgs"], sdv["img_feat"], sdv["w2cs"][1:], sdv["intrinsics"][1:])
    pw = ops.PackedWeights.get(src.V, 0, dev, "f32")
    rp = ops.RenderPass(src, pw, pw.pack(flat_params_of(sd).to(dev)), sdv["c2w"][:3, 3], sdv["w2cs"][0], sdv["intrinsics"][0], sdv["w2cs"][0], sc["near"], sc["far"], white)
    rp.repack_sources()
    out = rp(rays_d[sel].to(dev).contiguous(), z[sel].to(dev).contiguous(), keep=("raw", "feats"))
    sig = out["raw"][0, :, 3].cpu()
    print("density relu along ray %d: argument in float32 / float64 (oracle), density on the device" % ray)
    order = torch.argsort(pre[torch.float64].abs())[:8]
    for j in order.tolist():
        print("  sample %3d: %+.4e  %+.4e   device sigma %.4e" % (j, pre[torch.float32][j], pre[torch.float64][j], sig[j]))


def relu_masks(case_id, ray):
    """Every relu of the network along the ray: where the device's kept activation and the oracle's argument (float32, float64) sit on different sides."""
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    s = case_id
    case = FR.make_case(s)
    sc, V, white = case["scene"], case["V"], case["white"]
    n = case["n"] if case["n"] * case["S"] <= 12000 else max(1, 12000 // case["S"])
    xs, ys, z = case["xs"][:n], case["ys"][:n], case["z"][:n].contiguous()
    sd = init_ucnerf_state_dict(seed=s % 7, n_src=V - 1, sigma_scale=case["sigma"], sigma_bias=0.02)
    _, rays_d, _ = O.get_rays_mvs_pixels(xs, ys, sc["K"], sc["c2w"])
    sel = slice(ray, ray + 1)
    pre, feats, ndc = {}, None, None
    orig = torch.relu
    for dt in (torch.float32, torch.float64):
        seen = []
        torch.relu = lambda x: (seen.append(x.detach()), orig(x))[1]
        try:
            cv = lambda t: t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t      # noqa: E731
            scene = {k: ([cv(x) for x in v] if isinstance(v, list) else cv(v)) for k, v in sc.items()}
            want, pts, nd = FR.oracle_pass({k: cv(v) for k, v in sd.items()}, scene, cv(rays_d[sel]), cv(z[sel]), white)
        finally:
            torch.relu = orig
        pre[dt] = [t.reshape(-1, t.shape[-1]) for t in seen if t.dim() >= 2 and t.shape[-1] in (128, 64)]
        if dt == torch.float32:
            feats, ndc = want["feats"].detach(), nd["ndc"].detach()
    dev = torch.device(DEV)
    pw = ops.PackedWeights.get(V - 1, 0, dev, "f32")
    flat = flat_params_of(sd).to(dev)
    S = z.shape[1]
    cos = rays_d[sel].norm(dim=-1, keepdim=True)
    raw, sets = ops.mlp_fwd_train(pw, pw.pack(flat), ndc.reshape(-1, 3).to(dev), (rays_d[sel] / cos).to(dev), feats.reshape(S, -1).to(dev).contiguous(), S)
    print("raw: device vs oracle max diff %.2e" % float((raw.cpu() - want["raw"].reshape(-1, 4)).abs().max()))
    names = ["h0", "h1", "h2", "h3", "h4", "h5"]
    print("relu arguments captured from the oracle: %s" % [tuple(t.shape) for t in pre[torch.float32]])
    for li, name in enumerate(names):
        a32, a64, h = pre[torch.float32][li], pre[torch.float64][li], sets[name].cpu()
        flip = ((a64 > 0) != (h > 0))
        print("layer %s: %d of %d units differ in sign between the float64 oracle and the device's kept activation; float32 oracle vs float64: %d" % (
            name, int(flip.sum()), flip.numel(), int(((a64 > 0) != (a32 > 0)).sum())))
        for j, k in flip.nonzero().tolist()[:6]:
            print("    sample %d unit %d: oracle float32 %+.3e float64 %+.3e, device kept %+.3e (layer rms %.3g)" % (j, k, a32[j, k], a64[j, k], h[j, k], a64.pow(2).mean().sqrt()))


if __name__ == "__main__":
    if "--masks" in sys.argv:
        sys.argv.remove("--masks")
        ap = argparse.ArgumentParser()
        ap.add_argument("--case", type=int, required=True)
        ap.add_argument("--ray", type=int, required=True)
        a_ = ap.parse_args()
        relu_masks(a_.case, a_.ray)
        sys.exit(0)
    if "--sigma" in sys.argv:
        sys.argv.remove("--sigma")
        ap = argparse.ArgumentParser()
        ap.add_argument("--case", type=int, required=True)
        ap.add_argument("--ray", type=int, required=True)
        a_ = ap.parse_args()
        sigma_ties(a_.case, a_.ray)
        sys.exit(0)
    main()
