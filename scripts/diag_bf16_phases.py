"""Diagnostic (UCNERF_MLP_DIAG build of mlp_bf16.hip): where one tile (the 6th of every wave) spends its cycles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
W, NB = 4, 512
MB = int(os.environ.get("DIAG_MAX_BLOCKS", "0"))     # 256 = one wave per SIMD (solo phase times)
dbg = torch.zeros(NB * W * 16, dtype=torch.int64, device=dev)
os.environ["UCNERF_MLP_DIAG_PTR"] = str(dbg.data_ptr())
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.pipeline import flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict  # noqa: E402

m, S, F = 4096 * 192, 192, 97
pw = ops.PackedWeights.get(6, 0, dev, precision="bf16x3")
ws = pw.pack(flat_params_of(init_ucnerf_state_dict(0)).to(dev))
g = torch.Generator().manual_seed(0)
pts = torch.rand(m, 3, generator=g).to(dev)
dirs = torch.randn(m, 3, generator=g).to(dev)
feats = torch.randn(((m + 31) // 32) * 32 * F, generator=g).to(dev)
ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=True, max_blocks=MB)
torch.cuda.synchronize()
dbg.zero_()
torch.cuda.synchronize()
ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=True, max_blocks=MB)
torch.cuda.synchronize()
raw = dbg.cpu().view(NB * W, 16).double()
raw = raw[raw[:, 13] > 0]
print("tile start -> inputs arrived %.0f | sincos %.0f | split + stash %.0f" % ((raw[:, 4] - raw[:, 0]).mean(), (raw[:, 14] - raw[:, 4]).mean(), (raw[:, 1] - raw[:, 14]).mean()))
t = dbg.cpu().view(NB * W, 16)[:, :14].double()
t[:, 4] = t[:, 3]
t = t[t[:, 13] > 0]
d = t[:, 1:] - t[:, :-1]
names = ["feat loads + PE + stash", "bd GEMM (3 st)", "L0 GEMM (4 st)", "(unused)", "L1-4 (32 st + 4 epi)", "L5 GEMM (12 st)",
         "L5 epi + bc loads + base heads", "bc GEMM (3 st)", "h*bc -> frags", "FT GEMM (8 st) + frags", "VC GEMM (10 st) + dir PE", "relu + adapt heads",
         "blend + store"]
st = [0, 3, 4, 0, 32, 12, 0, 3, 0, 8, 10, 0, 0]
tot = (t[:, 13] - t[:, 0]).mean()
print("waves stamped %d; tile total %.0f counter ticks (two waves share a SIMD: ideal GEMM time = steps * 12 MFMA * 32 cyc * 2)" % (t.shape[0], tot))
exc = 0
for n, k, col in zip(names, st, d.t()):
    print("%-34s mean %8.0f  ideal %7d  excess %7.0f" % (n, col.mean(), k * 768, col.mean() - k * 768))
    exc += col.mean() - k * 768
print("sum of excess: %.0f (%.1f%% of the tile)" % (exc, 100 * exc / tot))
