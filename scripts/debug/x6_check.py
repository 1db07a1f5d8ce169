"""bf16x6 training forward against the exact-f32 one: raw outputs and the kept 24-bit sets (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uc_nerf_amd import ops
from uc_nerf_amd.pipeline import flat_params_of
from uc_nerf_amd.synthetic import init_ucnerf_state_dict

dev = torch.device("cuda:0")
for n_src, m, S in ((6, 4100, 10), (3, 777, 7), (8, 2048, 1), (1, 300, 3), (6, 131072, 128)):
    sd = init_ucnerf_state_dict(seed=3, n_src=n_src)
    flat = flat_params_of(sd).to(dev)
    gen = torch.Generator().manual_seed(m)
    F = 24 + 12 * n_src + 1
    pts, feats = torch.rand(m, 3, generator=gen), torch.randn(m, F, generator=gen)
    feats[:, -1] = torch.rand(m, generator=gen)
    dirs = torch.nn.functional.normalize(torch.randn(m // S, 3, generator=gen), dim=-1)
    pts, feats, dirs = pts.to(dev), feats.to(dev), dirs.to(dev)
    res = {}
    for prec in ("f32", "bf16x6"):
        pw = ops.PackedWeights.get(n_src, 0, dev, prec)
        ws = pw.pack(flat)
        raw, sets = ops.mlp_fwd_train(pw, ws, pts, dirs, feats, S, "chain")
        torch.cuda.synchronize()
        res[prec] = (raw, sets)
    r0, r1 = res["f32"][0], res["bf16x6"][0]
    n_mvs = 24 + 4 * n_src
    W, b = sd["nerf.pts_bias_depth_fine.weight"].double().to(dev), sd["nerf.pts_bias_depth_fine.bias"].double().to(dev)
    bd_true = feats[:, :n_mvs].double() @ W.T + b
    terms = feats[:, :n_mvs].double().abs() @ W.abs().T
    for prec in ("f32", "bf16x6"):
        e = (res[prec][1]["bd"].double() - bd_true).abs()
        print("   bd of %-6s against float64: max |err| %.3e, max |err| / sum |terms| %.3e, mean %.3e" % (prec, e.max().item(), (e / terms).max().item(), (e / terms).mean().item()))
    print("n_src %d m %d: raw max |diff| %.3e (max |raw| %.3e) nan %d" % (n_src, m, (r0 - r1).abs().max().item(), r0.abs().max().item(), int(torch.isnan(r1).sum())))
    for name in res["f32"][1]:
        a, b = res["f32"][1][name], res["bf16x6"][1][name]
        d = (a - b).abs()
        rel = (d / (a.abs().max() + 1e-30)).max().item()
        print("   %-3s max |diff| %.3e  (scale %.3e, rel-to-max %.2e)  unequal %.4f" % (name, d.max().item(), a.abs().max().item(), rel, (a != b).float().mean().item()))
    if m >= 100000:
        for prec in ("f32", "bf16x6"):
            pw = ops.PackedWeights.get(n_src, 0, dev, prec)
            ws = pw.pack(flat)
            for _ in range(3): ops.mlp_fwd_train(pw, ws, pts, dirs, feats, S, "chain")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ops.mlp_fwd_train(pw, ws, pts, dirs, feats, S, "chain")
            torch.cuda.synchronize()
            print("   %s training forward, %d samples: %.1f us (incl. the sets' decode on the host side of the wrapper)" % (prec, m, (time.perf_counter() - t0) / 10 * 1e6))
