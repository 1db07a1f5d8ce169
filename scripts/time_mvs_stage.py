"""Times ucnerf_cost_volume / ucnerf_depth_regress (forward and backward) at the three cascade-stage shapes (6 source views,
256x320 images) and reports them against the HBM roofline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uc_nerf_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
V = 6
stages = [("stage1", 32, 64, 80, 48), ("stage2", 16, 128, 160, 32), ("stage3", 8, 256, 320, 8)]

for name, C, H, W, D in stages:
    f = 250.0 * W / 320
    K4 = torch.eye(4); K4[0, 0] = K4[1, 1] = f; K4[0, 2] = W / 2; K4[1, 2] = H / 2
    w2c = torch.eye(4).repeat(V + 1, 1, 1)
    w2c[1:, 0, 3] = 0.01 * torch.arange(1, V + 1)
    affine = K4 @ w2c
    proj = (affine[1:] @ torch.inverse(affine)[0:1])[:, :3].contiguous()
    feats = torch.randn(V, C, H, W, generator=gen)
    # hypothesis bands as the cascade produces them (mvs_models.py:699-735: 48 planes over [near, far] = [1, 4]; then
    # 32 planes at twice, 8 planes at once the base interval (far - near) / 48 around the previous stage's depth)
    lo, hi = (1.0, 4.0) if D >= 32 else (2.5 - 4 * 3.0 / 48, 2.5 + 4 * 3.0 / 48)
    dv = (torch.linspace(lo, hi, D).view(D, 1, 1) + 0.02 * torch.rand(D, H, W, generator=gen)).contiguous()
    logits = torch.randn(D, H, W, generator=gen)
    fd, pd, dd, ld = feats.to(dev), proj.to(dev), dv.to(dev), logits.to(dev)
    for _ in range(3):
        var = ops.cost_volume(fd, pd, dd)
        out = ops.depth_regress(ld, dd)
    a, b, c = ops.Event(), ops.Event(), ops.Event()
    torch.cuda.synchronize()
    K = 20
    a.record()
    for _ in range(K):
        var = ops.cost_volume(fd, pd, dd)
    b.record()
    for _ in range(K):
        out = ops.depth_regress(ld, dd)
    c.record()
    torch.cuda.synchronize()
    t_cv, t_dr = a.elapsed_ms(b) / K, b.elapsed_ms(c) / K
    by_cv = 4 * (C * D * H * W + D * H * W + V * C * H * W)                 # volume written + hypotheses + source maps once
    by_dr = 4 * (3 * D * H * W + 2 * H * W)                                  # logits, hypotheses read; probabilities written; 2 maps
    # backward kernels (through the autograd wrappers: includes zeroing the feature-gradient maps)
    fg, lg = fd.clone().requires_grad_(True), ld.clone().requires_grad_(True)
    r_v = torch.randn_like(var)
    r_d, r_c = torch.randn_like(out[1]), torch.randn_like(out[2])

    def bwd_cv():
        fg.grad = None
        ops.cost_volume(fg, pd, dd).backward(r_v)

    def bwd_dr():
        lg.grad = None
        _, d_, c_ = ops.depth_regress(lg, dd)
        torch.autograd.backward([d_, c_], [r_d, r_c])

    for fn in (bwd_cv, bwd_dr):
        fn()
    e0, e1, e2 = ops.Event(), ops.Event(), ops.Event()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(K):
        bwd_cv()
    e1.record()
    for _ in range(K):
        bwd_dr()
    e2.record()
    torch.cuda.synchronize()
    t_cvb, t_drb = e0.elapsed_ms(e1) / K - t_cv, e1.elapsed_ms(e2) / K - t_dr      # forward + backward minus forward
    line = "%s C=%d D=%d %dx%d: cost_volume %.1f us = %.2f TB/s (%.2f of 8), depth_regress %.1f us = %.2f TB/s (%.2f of 8)" % (
        name, C, D, H, W, t_cv * 1e3, by_cv / t_cv / 1e9, by_cv / t_cv / 1e9 / 8, t_dr * 1e3, by_dr / t_dr / 1e9, by_dr / t_dr / 1e9 / 8)
    line += "; backward %.1f us / %.1f us" % (t_cvb * 1e3, t_drb * 1e3)
    print(line)
