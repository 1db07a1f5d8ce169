"""Diagnostic (UCNERF_MLP_DIAG build only): per-wave tile start/end clocks of mlp_fwd_kernel on the bench's fine pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(256 * 8 * 16 * 2, dtype=torch.int64, device=dev)
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
for _ in range(3):
    dbg.zero_()
    raw = ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=True)
torch.cuda.synchronize()
t = dbg.cpu().view(256, 8, 16, 2)
t0 = t[..., 0][t[..., 0] > 0].min()
dur = (t[..., 1] - t[..., 0])[:, :, :12].float()
start = (t[..., 0] - t0)[:, :, :12].float()
end = (t[..., 1] - t0)[:, :, :12].float()
print("tile duration (cycles): mean %.0f  min %.0f  max %.0f" % (dur.mean(), dur.min(), dur.max()))
print("waves 0-3 mean tile %.0f, waves 4-7 mean tile %.0f" % (dur[:, :4].mean(), dur[:, 4:].mean()))
print("kernel span (cycles) %.0f; per-wave finish: min %.0f max %.0f mean %.0f" % (end.max(), end[:, :, -1].min(), end[:, :, -1].max(), end[:, :, -1].mean()))
print("first-tile start spread: min %.0f max %.0f" % (start[:, :, 0].min(), start[:, :, 0].max()))
b = 7
for w in (0, 4):
    print("block %d wave %d starts:" % (b, w), [int(x) for x in start[b, w]])
    print("block %d wave %d durs  :" % (b, w), [int(x) for x in dur[b, w]])
fin = end[:, :, -1].max(1)[0]
print("per-block finish quantiles:", [int(torch.quantile(fin, q)) for q in (0.0, 0.1, 0.5, 0.9, 1.0)])
