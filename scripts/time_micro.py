import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from uc_nerf_amd import ops
from uc_nerf_amd.pipeline import flat_params_of
from uc_nerf_amd.synthetic import init_ucnerf_state_dict
dev = torch.device("cuda:0")
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
pw = ops.PackedWeights.get(6, 0, dev, "bf16x3"); ws = pw.pack(flat_params_of(sd).to(dev))
n, S = 4096, 128
g = torch.Generator().manual_seed(5)
feats = torch.randn(n * S, 97, generator=g); feats[:, -1] = torch.rand(n * S, generator=g); feats = feats.to(dev)
pts = torch.rand(n * S, 3, generator=g).to(dev)
dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
z = torch.sort(1 + 3 * torch.rand(n, S, generator=g), -1)[0].to(dev)
def t(fn, k=40):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
raw = ops.mlp_fwd(pw, ws, pts, dirs, feats, S)
print("mlp per-ray dirs   %.4f ms" % t(lambda: ops.mlp_fwd(pw, ws, pts, dirs, feats, S)))
dd = dirs[:, None].expand(n, S, 3).reshape(-1, 3).contiguous()
print("mlp per-sample dirs %.4f ms" % t(lambda: ops.mlp_fwd(pw, ws, pts, dd, feats, S)))
print("composite          %.4f ms" % t(lambda: ops.composite_fwd(raw.view(n, S, 4), z)))
