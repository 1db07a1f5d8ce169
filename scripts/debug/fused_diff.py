"""Where does the gather-fused pass differ from the two-kernel pass on the bench scene?  (GPU)"""
import torch
from uc_nerf_amd import ops
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to
DEV = torch.device("cuda:0")
scene = make_scene(seed=0)
sd = init_ucnerf_state_dict(seed=1, sigma_scale=0.1, sigma_bias=0.02)
H, W = scene["imgs"].shape[-2:]
xs, ys = random_pixels(4096, H, W, seed=3)
flat = flat_params_of(sd).to(DEV)
sc = scene_to(scene, DEV)
two = CoarseFineRenderer(sc, flat, 64, 128, precision="bf16x3")
one = CoarseFineRenderer(sc, flat, 64, 128, precision="bf16x3_fused")
a = two.render(xs.to(DEV), ys.to(DEV))
two.pass_.repack_sources(); one.pass_.repack_sources()
rays_d, angle, z_c = two.sampler(xs.to(DEV), ys.to(DEV), 0.0, None)
for name, z in (("coarse", z_c), ("fine", a["z_fine"])):
    ra = two.pass_(rays_d, z, keep=("raw",), dir_feat=angle)
    rb = one.pass_(rays_d, z, keep=("raw",), dir_feat=angle)
    d = (ra["raw"] - rb["raw"]).abs()
    print(name, "max raw diff", d.max().item(), "samples > 1e-4:", (d.amax(-1) > 1e-4).sum().item(), "of", d.shape[0] * d.shape[1])
    bad = (d.amax(-1) > 1e-4).nonzero()
    print("  first bad (ray, sample):", bad[:12].tolist())
    if len(bad):
        r, s = bad[0].tolist()
        print("  raw two", ra["raw"][r, s].tolist(), "one", rb["raw"][r, s].tolist(), "z", z[r, s].item())
        print("  rays with bad samples:", torch.unique(bad[:, 0]).numel(), " sample idx histogram:", torch.bincount(bad[:, 1], minlength=z.shape[1]).tolist())
    print("  rgb diff", (ra["rgb"] - rb["rgb"]).abs().max().item())

for trial in range(3):
    a = two.render(xs.to(DEV), ys.to(DEV))
    b = one.render(xs.to(DEV), ys.to(DEV))
    same = (a["z_fine"] - b["z_fine"]).abs().amax(-1) < 1e-4
    d = (a["rgb"] - b["rgb"]).abs().amax(-1)
    bad = (same & (d > 1e-4)).nonzero().flatten()
    print("trial", trial, "same", same.float().mean().item(), "bad rays among same:", bad.numel(), bad[:40].tolist())
    dw = (a["coarse"]["weights"] - b["coarse"]["weights"]).abs().amax(-1)
    print("   coarse weights diff > 1e-4 rays:", (dw > 1e-4).sum().item(), " max", dw.max().item())
    if bad.numel():
        r = bad[0].item()
        print("   ray", r, "rgb", a["rgb"][r].tolist(), b["rgb"][r].tolist(), "max dz", (a["z_fine"][r] - b["z_fine"][r]).abs().max().item())
        fa = two.pass_(a["rays_d"], a["z_fine"], keep=("raw",))
        fb = one.pass_(a["rays_d"], a["z_fine"], keep=("raw",))
        print("   same-z pass: rgb diff at ray", (fa["rgb"][r] - fb["rgb"][r]).abs().max().item(), " two-vs-render", (fa["rgb"][r] - a["rgb"][r]).abs().max().item(),
              " one-vs-render", (fb["rgb"][r] - b["rgb"][r]).abs().max().item())
