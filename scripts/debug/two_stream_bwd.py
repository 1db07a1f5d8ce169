"""VERDICT r04 item 4, the experiment: the training-style step's batch split in two halves on two streams -- chain(h1) -> [wgrad(h1) || chain(h2)]
-> wgrad(h2) falls out of stream order when half 2 starts behind half 1 -- against the same two halves on ONE stream and against the unsplit batch.
Two RenderPass objects (own workspaces), channel-last sources (no repack / transposing pass in the way), exact-f32 training forward.

  RAYS=2000 SAMPLES=90 python scripts/debug/two_stream_bwd.py           (under rocprofv3 --kernel-trace for the overlap picture)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uc_nerf_amd import ops  # noqa: E402
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to  # noqa: E402

dev = torch.device("cuda:0")
n, S = int(os.environ.get("RAYS", "2000")), int(os.environ.get("SAMPLES", "90"))
scene = scene_to(make_scene(seed=0), dev)
cl = ops.ChannelLastSources.from_reference_layout(scene["vols"], scene["imgs"], scene["img_feat"])
scene = dict(scene, vols=cl.vols, imgs=cl.imgs, img_feat=cl.img_feat)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
flat = flat_params_of(sd).to(dev)
full, ha, hb = (CoarseFineRenderer(scene, flat, 64, 128) for _ in range(3))
xs, ys = random_pixels(n, scene["H"], scene["W"], seed=0)
rays_d, _, _ = ops.ray_gen(full.K_host, full.c2w_host, xs=xs.to(dev), ys=ys.to(dev))
z, _ = ops.sample_stratified(None, S, n=n, near=full.near_host, far=full.far_host, device=dev)
g_rgb, g_depth = torch.randn(n, 3, device=dev), torch.randn(n, device=dev)
h = n // 2
parts = [(ha, slice(0, h)), (hb, slice(h, n))]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def unsplit():
    out = full.pass_(rays_d, z, keep=("raw", "feats"))
    return full.pass_.backward(rays_d, z, out, g_rgb, g_depth, flat)


def halves(streams):
    cur = torch.cuda.current_stream()
    res = []
    for (r, sl), st in zip(parts, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            rd, zz = rays_d[sl].contiguous(), z[sl].contiguous()
            out = r.pass_(rd, zz, keep=("raw", "feats"))
            res.append(r.pass_.backward(rd, zz, out, g_rgb[sl].contiguous(), g_depth[sl].contiguous(), flat))
    for st in set(streams):
        cur.wait_stream(st)
    return res


def timed(fn, k=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


a = timed(unsplit)
b = timed(lambda: halves((sa, sa)))
c = timed(lambda: halves((sa, sb)))
# the split is exact: the two halves' parameter gradients add up to the unsplit batch's (float-atomic order aside)
gu = unsplit()[0]
gh = halves((sa, sb))
torch.cuda.synchronize()
err = ((gh[0][0] + gh[1][0])[:gu.numel()] - gu).abs().max().item() / gu.abs().max().item()
print("train-style step %d x %d: unsplit %.3f ms | two halves, one stream %.3f ms | two halves, two streams %.3f ms | rel. gradient difference %.1e" % (n, S, a, b, c, err))
