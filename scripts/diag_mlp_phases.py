"""Diagnostic (UCNERF_MLP_DIAG build): where one tile (the 6th of every wave) spends its cycles, section by section."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
W = int(os.environ.get("DIAG_WAVES", "8"))
dbg = torch.zeros(256 * W * 16, dtype=torch.int64, device=dev)
os.environ["UCNERF_MLP_DIAG_PTR"] = str(dbg.data_ptr())
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.pipeline import flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict  # noqa: E402

m, S, F = 4096 * 192, 192, 97
pw = ops.PackedWeights.get(6, 0, dev)
ws = pw.pack(flat_params_of(init_ucnerf_state_dict(0)).to(dev))
g = torch.Generator().manual_seed(0)
pts = torch.rand(m, 3, generator=g).to(dev)
dirs = torch.randn(m, 3, generator=g).to(dev)
feats = torch.randn(((m + 31) // 32) * 32 * F, generator=g).to(dev)
ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=True)
torch.cuda.synchronize()
dbg.zero_()
torch.cuda.synchronize()
ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=True)
torch.cuda.synchronize()
t = dbg.cpu().view(256 * W, 16)[:, :14].double()
t = t[t[:, 13] > 0]
d = t[:, 1:] - t[:, :-1]
names = ["feat-issue+PE+stash", "bd GEMM (24 ks)", "L0 GEMM (32 ks)", "L0 epilogue", "L1-4 (256 ks + 4 epi)", "L5 GEMM (96 ks)",
         "L5 epi + bc loads + base heads", "bc GEMM (24 ks)", "h*bc + dir load", "FT GEMM (64 ks)", "VC GEMM (80 ks)", "relu + adapt heads",
         "blend + store"]
ks = [0, 24, 32, 0, 256, 96, 0, 24, 0, 64, 80, 0, 0]
tot = (t[:, 13] - t[:, 0]).mean()
print("tile total %.0f cycles (two waves share a SIMD: ideal GEMM time = k-steps * 64 * waves)" % tot)
exc = 0
for n, k, col in zip(names, ks, d.t()):
    print("%-34s mean %8.0f  ideal %7d  excess %7.0f" % (n, col.mean(), k * 64 * W, col.mean() - k * 64 * W))
    exc += col.mean() - k * 64 * W
print("sum of excess: %.0f (%.1f%% of the tile)" % (exc, 100 * exc / tot))
