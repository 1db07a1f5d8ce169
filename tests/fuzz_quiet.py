"""How do the split's two kinds of terms behave on a QUIET network?  The trunk of the reference's MLP is positively homogeneous in its first layer
(kaiming init: zero biases), so scaling pts_linears.0 by alpha scales every trunk activation by alpha: the render pass of such networks, alpha from 1 to
1e-5, with bf16 and with fp16 terms against the oracle (fuzz_render's scenes).   python tests/fuzz_quiet.py      (test infrastructure: imports oracle/)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_render as FR                                                         # noqa: E402
from oracle import ucnerf_oracle as O                                            # noqa: E402

DEV = "cuda:0"


def main():
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, scene_to
    dev = torch.device(DEV)
    print("alpha    max |activation| of the trunk (oracle)   bf16 terms: max err rgb / acc      fp16 terms: max err rgb / acc     (worst over 6 scenes)")
    for alpha in (1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
        worst = {0: [0.0, 0.0], 1: [0.0, 0.0]}
        amax = 0.0
        for seed in (3, 5, 9, 12, 20, 31):
            case = FR.make_case(seed)
            sc, V = case["scene"], case["V"]
            sd = init_ucnerf_state_dict(seed=seed % 7, n_src=V - 1, sigma_scale=1.0, sigma_bias=0.02)
            sd["nerf.pts_linears.0.weight"] = sd["nerf.pts_linears.0.weight"] * alpha
            # (a density head that undoes the scale, so that the render stays a render: sigma of order one)
            sd["nerf.alpha_linear_1.weight"] = sd["nerf.alpha_linear_1.weight"] / alpha
            _, rays_d, _ = O.get_rays_mvs_pixels(case["xs"], case["ys"], sc["K"], sc["c2w"])
            seen, orig = [], torch.relu
            torch.relu = lambda x: (seen.append(float(x.detach().abs().max())), orig(x))[1]
            try:
                want, _, _ = FR.oracle_pass(sd, sc, rays_d, case["z"], False)
            finally:
                torch.relu = orig
            amax = max(amax, max(seen[:6]))
            sdv = scene_to(sc, dev)
            flat = flat_params_of(sd).to(dev)
            for op in (0, 1):
                src = ops.GatherSources(sdv["vols"], sdv["confidence"], sdv["imgs"], sdv["img_feat"], sdv["w2cs"][1:], sdv["intrinsics"][1:])
                pw = ops.PackedWeights.get(src.V, 0, dev, "bf16x3_fused", operand=op)
                rp = ops.RenderPass(src, pw, pw.pack(flat), sdv["c2w"][:3, 3], sdv["w2cs"][0], sdv["intrinsics"][0], sdv["w2cs"][0], sc["near"], sc["far"], False)
                rp.repack_sources()
                out = rp(rays_d.to(dev), case["z"].to(dev))
                worst[op][0] = max(worst[op][0], float((out["rgb"].cpu() - want["rgb"]).abs().max()))
                worst[op][1] = max(worst[op][1], float((out["acc"].cpu() - want["acc"]).abs().max()))
        print("%-8g %-40.3g %.2e / %.2e               %.2e / %.2e" % (alpha, amax, worst[0][0], worst[0][1], worst[1][0], worst[1][1]), flush=True)


if __name__ == "__main__":
    main()
