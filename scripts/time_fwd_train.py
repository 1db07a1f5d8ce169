"""Times the training forward (ucnerf_mlp_fwd_train: activations kept) against the plain forward on row-major features, both precisions:
what keeping the ten activation sets costs.  M=180000 S=90 by default."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uc_nerf_amd import _lib as L, ops  # noqa: E402
from uc_nerf_amd.pipeline import flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict  # noqa: E402

dev = torch.device("cuda:0")
m, S, F = int(os.environ.get("M", "180000")), int(os.environ.get("S", "90")), 97
g = torch.Generator().manual_seed(0)
pts = torch.rand(m, 3, generator=g).to(dev)
dirs = torch.randn(m // S, 3, generator=g).to(dev)
feats = torch.randn(m, F, generator=g).to(dev)
flat = flat_params_of(init_ucnerf_state_dict(0)).to(dev)
raw = torch.empty(m, 4, device=dev)
for prec in ("f32", "bf16x3"):
    pw = ops.PackedWeights.get(6, 0, dev, precision=prec)
    ws = pw.pack(flat)
    p = L.MlpParams()
    p.cfg = pw.cfg
    p.m, p.S = m, S
    p.pts, p.dirs, p.feats, p.wstream, p.raw = (ops._ptr(t) for t in (pts, dirs, feats, ws, raw))
    need = L.lib().ucnerf_mlp_bwd_workspace_floats(C.addressof(pw.cfg), m)
    work = torch.empty(need, device=dev)
    st = ops._stream()

    def fwd():
        L.check(L.lib().ucnerf_mlp_fwd(C.addressof(p), st), "fwd")

    def fwd_train():
        L.check(L.lib().ucnerf_mlp_fwd_train(C.addressof(p), ops._ptr(work), ops._backward_mode, st), "fwd_train")

    for name, fn in (("forward", fwd), ("training forward (10 sets kept)", fwd_train)):
        for _ in range(3):
            fn()
        a, b = ops.Event(), ops.Event()
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        print("%-7s %-34s %8.1f us per launch (%d samples)" % (prec, name, a.elapsed_ms(b) / 10 * 1e3, m))
