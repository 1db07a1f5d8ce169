"""Parameter gradients of the one-launch weight-gradient kernel (bwd_mode 0) against the layer-by-layer backward (bwd_mode 1), tensor by
tensor, at a given number of samples:  python scripts/debug/wgrad_modes.py [m] [S]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from tests.conftest import load_golden
from uc_nerf_amd import ops

m, S, F = int(sys.argv[1]) if len(sys.argv) > 1 else 180000, int(sys.argv[2]) if len(sys.argv) > 2 else 90, 97
sd = load_golden("sd_v7")
gen = torch.Generator().manual_seed(12)
pts, feats = torch.rand(m, 3, generator=gen), torch.randn(m, F, generator=gen)
feats[:, -1] = torch.rand(m, generator=gen)
dirs = torch.nn.functional.normalize(torch.randn(m // S, 3, generator=gen), dim=-1)
r = torch.randn(m, 4, generator=gen)
dev = torch.device("cuda:0")
pw = ops.PackedWeights.get(6, 0, dev)
flat0 = torch.cat([v.reshape(-1) for v in sd.values()]).float()
res = {}
for mode in ("chain", "layerwise"):
    ops.set_backward_mode(mode)
    flat = flat0.to(dev).requires_grad_(True)
    fd = feats.to(dev).requires_grad_(True)
    (ops.mlp(flat, fd, pts.to(dev), dirs.to(dev), pw, S) * r.to(dev)).sum().backward()
    res[mode] = flat.grad.cpu().clone()
ops.set_backward_mode("chain")
o = 0
for k, v in sd.items():
    n = v.numel()
    a, b = res["chain"][o:o + n], res["layerwise"][o:o + n]
    gm = b.abs().max().item()
    print("%-40s %-14s |g|max %.3e  max err / |g|max %.2e  zeros %d / %d" % (k, tuple(v.shape), gm, (a - b).abs().max().item() / max(gm, 1e-30), int((a == 0).sum()), n))
    o += n
