"""Times ucnerf_mlp_fwd alone (fine-pass shape, tiled features) -- for A/B of library variants, results not checked."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.pipeline import flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict  # noqa: E402

dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x3")
m, S, F = 4096 * 192, 192, 97
pw = ops.PackedWeights.get(6, 0, dev, precision=prec)
ws = pw.pack(flat_params_of(init_ucnerf_state_dict(0)).to(dev))
g = torch.Generator().manual_seed(0)
mode = os.environ.get("DATA", "random")        # random | zeros | small : operand toggling (the kernel is power-bound)
pts = torch.rand(m, 3, generator=g).to(dev)
dirs = torch.randn(m, 3, generator=g).to(dev)
TILED = os.environ.get("ROWMAJOR", "0") != "1"   # ROWMAJOR=1: features [m,F] row-major (the public API / training layout)
feats = torch.randn(((m + 31) // 32) * 32 * F if TILED else m * F, generator=g).to(dev)
if mode == "zeros":
    pts, dirs, feats = torch.zeros_like(pts), torch.zeros_like(dirs), torch.zeros_like(feats)
elif mode == "small":
    feats = feats * 0.05
MB = int(os.environ.get("MAXB", "0"))         # 256 = one wave per SIMD (how long a wave's own chain is without a partner)
for _ in range(3):
    ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=TILED, max_blocks=MB)
a, b = ops.Event(), ops.Event()
torch.cuda.synchronize()
a.record()
K = 10
for _ in range(K):
    ops.mlp_fwd(pw, ws, pts, dirs, feats, S, feats_tiled=TILED, max_blocks=MB)
b.record()
torch.cuda.synchronize()
print("%s %s data=%s max_blocks=%d: %.4f ms per fine-pass launch (786432 samples)" % (os.environ.get("UCNERF_LIB", "default").split("_")[-1], prec, mode, MB, a.elapsed_ms(b) / K))
