"""Pure host time (no drain between calls) of the two calls of one evaluation chunk -- build_rays_test and rendering() -- and of their parts."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, scene_to  # noqa: E402

dev = torch.device("cuda:0")
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
a, kw, net, grad_vars, outputs = bench.dropin_setup(scene, sd, dev)
import utils.utils as U  # noqa: E402
import network.renderer as renderer  # noqa: E402
H, W, chunk = 256, 320, 1024
near_fars = torch.tensor([[scene["near"], scene["far"]]] * 7, device=dev)
pose_ref = {"w2cs": scene["w2cs"].clone(), "intrinsics": scene["intrinsics"].clone()}


def t_us(fn, n=400, sync_every=40):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for k in range(n):
        t0 = time.perf_counter()
        fn()
        tot += time.perf_counter() - t0
        if k % sync_every == sync_every - 1:
            torch.cuda.synchronize()            # (keeps the queue from filling: a full queue would block the issuing thread)
    return tot / n * 1e6


with torch.no_grad():
    b = lambda: U.build_rays_test(H, W, scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], near_fars, near_fars[-1], 90, chunk=chunk, idx=7, outputs=outputs)      # noqa: E731
    pts, rd, ndc, z, ro, ndcp = b()
    r = lambda: renderer.rendering(a, pose_ref, pts, ndc, z, rd, outputs, scene["imgs"], near_fars=near_fars[0], img_feat=scene["img_feat"],      # noqa: E731
                                   confidence=scene["confidence"], ndc_parameters=ndcp, network_fn=kw["network_fn"], network_query_fn=kw["network_query_fn"], white_bkgd=False)
    r()
    print("build_rays_test            %6.1f us" % t_us(b))
    print("  torch.rand               %6.1f us" % t_us(lambda: torch.rand((1024, 90), device=dev)))
    print("  torch.empty (6 MB)       %6.1f us" % t_us(lambda: torch.empty(1500000, device=dev)))
    dv = [outputs["stage%d" % k]["depth_values"] for k in (1, 2, 3)]
    tr = torch.rand((1024, 90), device=dev)
    print("  ops.build_rays_test      %6.1f us" % t_us(lambda: ops.build_rays_test(H, W, 7168, 1024, 90, scene["intrinsics"][0], scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0], near_fars[0], dv, tr)))
    print("rendering()                %6.1f us" % t_us(r))
    from uc_nerf_amd import dropin
    sess = dropin.session_of(kw["network_fn"])
    vols = [outputs["stage%d" % k]["volume_feature_no_ref"] for k in (1, 2, 3)]
    print("  session_of               %6.1f us" % t_us(lambda: dropin.session_of(kw["network_fn"])))
    print("  sess.sources (hit)       %6.1f us" % t_us(lambda: sess.sources(vols, scene["confidence"], scene["imgs"], scene["img_feat"], pose_ref["w2cs"], pose_ref["intrinsics"])))
    src = sess.src
    print("  sess.render_pass         %6.1f us" % t_us(lambda: sess.render_pass("bf16x3_fused", 0, src, False)))
    print("    store.sync             %6.1f us" % t_us(lambda: sess.store.sync()))
    rp, _ = sess.render_pass("bf16x3_fused", 0, src, False)
    coords = {"pts": pts, "stage1": ndc["stage1"], "stage2": ndc["stage2"], "stage3": ndc["stage3"], "ndc": ndc["ndc"]}
    print("  RenderPass.__call__      %6.1f us" % t_us(lambda: rp(rd, z, want=(), coords=coords, w2c_dir_dev=pose_ref["w2cs"][0])))
    print("  query_layout             %6.1f us" % t_us(lambda: dropin.query_layout(kw["network_query_fn"])))
    x = torch.zeros(1024, 3, device=dev)
    print("x.cpu() of 12 KB after sync %5.1f us" % t_us(lambda: x.cpu(), n=200, sync_every=1))
