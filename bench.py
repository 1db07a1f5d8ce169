#!/usr/bin/env python3
"""Headline benchmark: rendered rays/sec, 64 coarse + 128 fine samples (the fine pass evaluates the merged
192), 4096 synthetic rays per GPU (BASELINE.json configs[1]).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N ...                 (WORLD_SIZE unset: starts the N ranks itself -- see self_launch_command -- and relays rank 0's line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W [--scaling weak] [--mode train]

One process per GPU.  Rays are independent, so the batch is sharded across ranks with no data-path collective:
  --scaling strong (default) a 4096-ray global batch split N ways (512 rays per GPU at N = 8: the north-star's strong-scaling target);
                   at N > 1 the weak-scaling step is measured too and reported as the secondary field `weak`
  --scaling weak   4096 rays per GPU, global batch 4096 N
The only collectives of the render mode are the barrier and the MAX of the timed interval.  Rank 0 prints ONE JSON line.
  --mode train     the line's metric becomes training rays/sec: rendering() drop-in forward + backward on this rank's
                   shard of a live-path batch (2000 rays x 90 cascade samples per GPU), ONE flat-bucket all-reduce of the
                   gradients over RCCL (parallel.FlatGradBucket, 0.73 MB), Adam step.
In render mode with N > 1 a short data-parallel training run is appended as the secondary field `train_dp`, so that the
RCCL bucket all-reduce is exercised and timed whenever the driver runs the scaling series.
"""
import argparse
import json
import math
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 2 * 147328          # MLP multiply-accumulates per sample evaluation
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
# bf16x3: every algorithmic multiply-accumulate is three bf16 MFMA multiply-accumulates on K padded to 16 per step
BF16X3_EXECUTED_FLOP_PER_SAMPLE = 3 * 2 * 72 * 16 * 128


def cpu_baseline(scene_cpu, sd, n_rays, n_coarse, n_fine, budget_s=45.0):
    """The CPU oracle (torch-CPU restatement of the reference path, all host cores) on a bounded sample of the
    same workload.  Test infrastructure used ONLY as the reported baseline."""
    from oracle import ucnerf_oracle as O
    from uc_nerf_amd.synthetic import random_pixels
    # the GPU box gives one GPU tenant a 16-core share of the host; more intra-op threads than that only
    # oversubscribes (measured: 256 threads -> 1.5 rays/s)
    cores = min(os.cpu_count() or 1, int(os.environ.get("UCNERF_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    xs, ys = random_pixels(4096, scene_cpu["H"], scene_cpu["W"], seed=0)          # the GPU's own seed-0 batch ("same inputs")
    xs, ys = xs[:n_rays], ys[:n_rays]
    noise = torch.rand(n_rays, n_coarse, generator=torch.Generator().manual_seed(100))
    times = []
    with torch.no_grad():
        O.render_coarse_fine(sd, scene_cpu, xs[:64], ys[:64], n_coarse, n_fine, noise=noise[:64], perturb=1.0)   # warm-up
        t_all = time.perf_counter()
        while len(times) < 5 and (time.perf_counter() - t_all) < budget_s:
            t0 = time.perf_counter()
            O.render_coarse_fine(sd, scene_cpu, xs, ys, n_coarse, n_fine, noise=noise, perturb=1.0)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": n_rays / med, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "the first %d rays of the GPU's seed-0 batch x (%d+%d) samples, same synthetic scene, torch-CPU fp32 oracle, "
                      "median of %d runs" % (n_rays, n_coarse, n_fine, len(times))}


def parity_vs_oracle(scene_cpu, sd, out, xs_cpu, ys_cpu, noise_cpu, n_coarse, n_fine, n_check=256):
    """"PSNR vs ref" of the metric, on a bounded sample: the first `n_check` rays of the timed batch re-rendered by the CPU oracle (the pinned
    restatement of the reference path) teacher-forced on the device's own fine depths (the free-running hierarchy is only statistically
    comparable: a draw landing in the neighbouring cdf bin moves a fine depth), against the device's render.  The checker, never the product."""
    from oracle import ucnerf_oracle as O
    n = min(n_check, int(xs_cpu.shape[0]))
    with torch.no_grad():
        ref = O.render_coarse_fine(sd, scene_cpu, xs_cpu[:n], ys_cpu[:n], n_coarse, n_fine, noise=noise_cpu[:n], perturb=1.0,
                                   z_fine_override=out["z_fine"][:n].cpu())
    d_rgb, d_depth = out["rgb"][:n].cpu() - ref["rgb"], out["depth"][:n].cpu() - ref["depth"]
    mse = torch.mean(d_rgb ** 2).item()
    return {"rays": n, "psnr_db": -10.0 * math.log10(max(mse, 1e-20)), "max_abs_rgb": d_rgb.abs().max().item(), "max_abs_depth": d_depth.abs().max().item(),
            "max_abs_coarse_weight": (out["coarse"]["weights"][:n].cpu() - ref["coarse"]["weights"]).abs().max().item(),
            "bar": 1e-4, "note": "the device's render of the first %d rays of the timed batch against the CPU oracle (fp32, the pinned restatement of the reference "
                                 "path) teacher-forced on the device's own fine depths; all rays counted (the batch's knife-edge rays included)" % n}


def self_launch_command(gpus, argv, env, port=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE / RANK unset): the command that starts the N ranks --
    one process per GPU through torch.distributed.run, rendezvous on 127.0.0.1 -- or None when this process IS a rank (or N = 1).
    The parent never touches the GPU (nothing here initialises HIP) and never replaces itself: the ranks are children."""
    if gpus <= 1 or "WORLD_SIZE" in env or "RANK" in env:
        return None
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus, "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    cmd = self_launch_command(gpus, argv, os.environ)
    if cmd is None:
        return
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)      # stdout is inherited: rank 0's JSON line passes through


class Ctx:
    """Process-wide state of one bench run (rank / world / device / collectives)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
        # one rank per GPU; UCNERF_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks (ranks then
        # share devices round-robin and the collectives go through gloo) -- the driver's runs use RCCL ("nccl")
        self.backend = os.environ.get("UCNERF_BENCH_BACKEND", "nccl")
        n_dev = torch.cuda.device_count()
        if local >= n_dev and self.backend == "nccl":
            raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs visible" % (local, n_dev))
        torch.cuda.set_device(local % n_dev)
        self.dev = torch.device("cuda", local % n_dev)
        self.dist = None
        # UCNERF_BENCH_GROUP_AT_1=1: create the process group with ONE rank too (and run `train_dp` through it): the only way to put the RCCL
        # calls of the data-parallel step on a box with one GPU -- a rehearsal of the code path, not a scaling measurement
        if self.world > 1 or os.environ.get("UCNERF_BENCH_GROUP_AT_1") == "1":
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.dist is None:
            return dt
        t = torch.tensor([dt], device=self.dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def timed(self, fn, steps, warmup):
        """W untimed calls, barrier + synchronize, K timed calls, barrier + synchronize; seconds per call, MAX over ranks."""
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0) / steps


def guarded(fn, key=None):
    """Secondary measurements never take the headline line down with them.  With `key` the failure is reported as
    {key: {"error": ...}} (for measurements that contribute several fields via dict.update)."""
    try:
        return fn()
    except Exception as e:       # noqa: BLE001
        err = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        return {key: err} if key else err


# ------------------------------------------------------------------------------------------------ sources as a real producer writes them
def conv_produced_sources(scene, dev, seed=0):
    """Gather sources that actual torch convolutions wrote in torch's channels_last memory formats -- what the networks in front of the path
    produce when they run in those formats on this GPU (network/mvs_models.py:624-646: cost_regularization -> volume_feature_no_ref per stage,
    FeatureNet -> img_feats): one nn.Conv3d(8, 8, 3) per cascade volume in channels_last_3d, one nn.Conv2d(8, 8, 3) over the V feature maps in
    channels_last, the images as a channels_last [V,3,H,W] stack.  Each is its OWN allocation with whatever strides MIOpen returns (recorded).
    Returns (dict(vols, imgs, img_feat), record)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    c3 = torch.nn.Conv3d(8, 8, 3, padding=1).to(dev).to(memory_format=torch.channels_last_3d)
    c2 = torch.nn.Conv2d(8, 8, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        for m in (c3, c2):
            m.weight.copy_(torch.randn(m.weight.shape, generator=g).to(dev) * 0.1)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g).to(dev) * 0.1)
        vols = [c3(v.contiguous(memory_format=torch.channels_last_3d)) for v in scene["vols"]]
        feat = c2(scene["img_feat"][:, 0].contiguous(memory_format=torch.channels_last)).unsqueeze(1)       # [V,1,8,H,W], as the reference shapes it
        imgs = scene["imgs"][0].contiguous(memory_format=torch.channels_last).unsqueeze(0)                 # [1,V,3,H,W]
    rec = {"vol_strides": [list(v.stride()) for v in vols], "img_feat_stride": list(feat.stride()), "imgs_stride": list(imgs.stride()),
           "vols_channels_last_3d": [bool(v.is_contiguous(memory_format=torch.channels_last_3d)) for v in vols],
           "separate_allocations": len({t.untyped_storage().data_ptr() for t in vols + [feat, imgs]}) == 5}
    return {"vols": vols, "imgs": imgs, "img_feat": feat}, rec


# ------------------------------------------------------------------------------------------------ live-path (drop-in) workloads
def dropin_setup(scene, sd, dev):
    import uc_nerf_amd
    uc_nerf_amd.install_dropin()
    import network.models as models
    from uc_nerf_amd.synthetic import cascade_outputs
    a = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=6, netwidth=128, feat_dim=97, net_type="v2", view_num=7,
                              netchunk=1024, perturb=1.0, N_samples=90, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0, ckpt=None,
                              device=str(dev), img_downscale=1.0, use_color_volume=False)
    kw_train, _, _, grad_vars = models.create_ucnerf(a, dir_embedder=True, pts_embedder=True)      # as train.py:36-37
    net = kw_train["network_fn"]
    net.load_state_dict(sd)
    outputs = cascade_outputs(scene)
    return a, kw_train, net, grad_vars, outputs


def dropin_call(a, kw, scene, outputs, batch, confidence=None, vols=None, img_feat=None):
    import network.renderer as renderer
    vf = outputs if vols is None else {"stage%d" % (k + 1): {"volume_feature_no_ref": vols[k]} for k in range(3)}
    pose = batch["pose_ref"]                                  # trimmed in place by the first call, as in train.py
    return renderer.rendering(a, pose, batch["rays_pts"], batch["rays_ndc"], batch["depth_candidates"], batch["rays_dir"], vf, scene["imgs"],
                              img_feat=scene["img_feat"] if img_feat is None else img_feat,
                              confidence=scene["confidence"] if confidence is None else confidence,
                              network_fn=kw["network_fn"], network_query_fn=kw["network_query_fn"], white_bkgd=kw["white_bkgd"])


def bench_eval_image(ctx, scene, a, kw, outputs, ms_per_call_default, images=7, chunk=1024):
    """The reference's evaluation loop, literally (train.py:251-275): per 256 x 320 image, 80 chunks of build_rays_test -> rendering -> .cpu() (x2),
    then the cat / reshape / clamp of the image -- through the drop-in modules, on the synthetic scene.  Every .cpu() drains the stream, so
    the host's issue time of a chunk is NOT hidden behind the GPU: this is what an unmodified caller waits for."""
    import utils.utils as U
    import network.renderer as renderer
    dev = ctx.dev
    H, W = scene["H"], scene["W"]
    near_fars = torch.tensor([[scene["near"], scene["far"]]] * scene["w2cs"].shape[0], device=dev)
    tgt_to_world, world_to_ref, intrinsic = scene["c2w"], scene["w2cs"][0], scene["intrinsics"][0]
    n_chunks = H * W // chunk + int(H * W % chunk > 0)
    split = {"build_rays_test": 0.0, "rendering": 0.0, "cpu": 0.0}

    def image(timing=False):
        pose_ref = {"w2cs": scene["w2cs"].clone(), "intrinsics": scene["intrinsics"].clone()}
        rgbs, depth_preds = [], []
        for chunk_idx in range(n_chunks):
            t0 = time.perf_counter()
            rays_pts, rays_dir, rays_NDC, depth_candidates, rays_o, ndc_parameters = U.build_rays_test(
                H, W, tgt_to_world, world_to_ref, intrinsic, near_fars, near_fars[-1], a.N_samples, pad=0, chunk=chunk, idx=chunk_idx, outputs=outputs)
            t1 = time.perf_counter()
            rgb, depth_pred = renderer.rendering(a, pose_ref, rays_pts, rays_NDC, depth_candidates, rays_dir, outputs, scene["imgs"],
                                                 near_fars=near_fars[0], img_feat=scene["img_feat"], confidence=scene["confidence"],
                                                 ndc_parameters=ndc_parameters, network_fn=kw["network_fn"], network_query_fn=kw["network_query_fn"],
                                                 white_bkgd=kw["white_bkgd"])
            t2 = time.perf_counter()
            rgbs.append(rgb.cpu())
            depth_preds.append(depth_pred.cpu())
            if timing:
                t3 = time.perf_counter()
                split["build_rays_test"] += t1 - t0; split["rendering"] += t2 - t1; split["cpu"] += t3 - t2
        return torch.clamp(torch.cat(rgbs).reshape(H, W, 3).permute(2, 0, 1), 0, 1), torch.cat(depth_preds).reshape(H, W)

    with torch.no_grad():
        for _ in range(2):
            rgb_img, depth_img = image()
        torch.cuda.synchronize()
        per = []
        for _ in range(images):
            t0 = time.perf_counter()
            rgb_img, depth_img = image()
            per.append(time.perf_counter() - t0)
        image(timing=True)
    assert tuple(rgb_img.shape) == (3, H, W) and torch.isfinite(rgb_img).all() and torch.isfinite(depth_img).all()
    per.sort()
    dt = per[len(per) // 2]                       # median: the loop drains the stream 160 times per image and follows the host's scheduling noise
    return {"ms_per_image": dt * 1e3, "ms_per_image_min": per[0] * 1e3, "ms_per_image_max": per[-1] * 1e3, "images_timed": images,
            "value": H * W / dt, "unit": "rays/s", "chunks": n_chunks, "rays_per_chunk": chunk, "samples_per_ray": a.N_samples,
            "ms_rendering_calls_alone": n_chunks * ms_per_call_default if chunk == 1024 else None,
            "host_wall_ms_per_image_by_call": {k: v * 1e3 for k, v in split.items()},
            "note": "the loop of train.py:251-275 through the drop-in modules, unmodified: build_rays_test -> rendering -> rgb.cpu(), depth.cpu() per %d-pixel "
                    "chunk (args.chunk of the reference's own command line); median of %d images; host_wall_ms_per_image_by_call = wall time inside each call "
                    "during one more image (the .cpu() wait contains the chunk's GPU time)" % (chunk, images)}


def bench_dropin(ctx, scene, sd, steps=40, warmup=10):
    """rendering() -- the reference's call surface -- on the shapes train.py runs it on, next to the library's own RenderPass."""
    import uc_nerf_amd
    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import flat_params_of
    from uc_nerf_amd.synthetic import live_path_batch
    dev = ctx.dev
    a, kw, net, grad_vars, outputs = dropin_setup(scene, sd, dev)
    out = {}
    # evaluation: one 1024-pixel chunk x 90 samples under no_grad (train.py:254-272)
    ev = live_path_batch(scene, outputs, 1024, 90, seed=3, chunk_idx=7)
    precs = ("f32", "bf16x3", "bf16x3_fused")      # bf16x3_fused: the headline kernel (gather inside the MLP kernel) on the coordinates rendering() is handed
    with torch.no_grad():
        # what an UNMODIFIED caller gets: install_dropin() and nothing else (train.py:254-272 calls rendering() under no_grad)
        from uc_nerf_amd import dropin as _dropin
        dt = ctx.timed(lambda: dropin_call(a, kw, scene, outputs, ev), steps * 4, warmup * 16)      # (first leg of this section: the longer warm-up lets the clock settle on it)
        out["dropin_eval_default"] = {"ms_per_call": dt * 1e3, "value": 1024 / dt, "unit": "rays/s", "rays": 1024, "samples_per_ray": 90,
                                      "precision": _dropin.inference_precision(a), "weight_cache": "verify (re-packed from the live parameters in every call)",
                                      "note": "rendering() after install_dropin() alone: no precision knob touched"}
        for prec in precs:
            uc_nerf_amd.set_inference_precision(prec)
            dt = ctx.timed(lambda: dropin_call(a, kw, scene, outputs, ev), steps * 4, warmup * 4)
            out["dropin_eval_" + prec] = {"ms_per_call": dt * 1e3, "value": 1024 / dt, "unit": "rays/s", "rays": 1024, "samples_per_ray": 90,
                                          "weight_cache": "verify (re-packed from the live parameters in every call)"}
            # the opt-in version-keyed weight cache (one small launch less per call; blind to `.data` writes, see dropin.set_weight_cache)
            uc_nerf_amd.set_weight_cache("versions")
            try:
                dtv = ctx.timed(lambda: dropin_call(a, kw, scene, outputs, ev), steps * 4, warmup * 4)
            finally:
                uc_nerf_amd.set_weight_cache("verify")
            out["dropin_eval_" + prec]["ms_per_call_versions_cache"] = dtv * 1e3
        uc_nerf_amd.set_inference_precision("bf16x3_fused")       # (back to the default)
        # the same chunk through the library's own RenderPass (coordinates derived in-kernel from ray + depth + cascade ranges)
        src = ops.GatherSources(scene["vols"], scene["confidence"], scene["imgs"], scene["img_feat"], scene["w2cs"][1:], scene["intrinsics"][1:])
        for prec in precs:
            pw = ops.PackedWeights.get(6, 0, dev, prec)
            rp = ops.RenderPass(src, pw, pw.pack(flat_params_of(sd).to(dev)), scene["c2w"][:3, 3].to(dev), scene["w2cs"][0], scene["intrinsics"][0],
                                scene["w2cs"][0], scene["near"], scene["far"])
            rp.repack_sources()
            ang, _ = ops.dir_feature(ev["rays_dir"], scene["w2cs"][0])
            dt = ctx.timed(lambda: rp(ev["rays_dir"], ev["depth_candidates"], near_far=ev["ranges"], want=(), dir_feat=ang), steps * 4, warmup * 4)
            out["render_pass_1024x90_" + prec] = {"ms_per_call": dt * 1e3, "value": 1024 / dt, "unit": "rays/s"}
        for prec in precs:
            rp_ms = out["render_pass_1024x90_" + prec]["ms_per_call"]
            out["dropin_eval_" + prec]["speed_vs_render_pass"] = rp_ms / out["dropin_eval_" + prec]["ms_per_call"]
            out["dropin_eval_" + prec]["speed_vs_render_pass_versions_cache"] = rp_ms / out["dropin_eval_" + prec]["ms_per_call_versions_cache"]
    out["dropin_eval_image"] = guarded(lambda: bench_eval_image(ctx, scene, a, kw, outputs, out["dropin_eval_default"]["ms_per_call"]))
    # the same loop with the reference's own --chunk option set to the image size: ONE build_rays_test + ONE rendering per image (the fused pass keeps
    # no per-sample features, so nothing but the coordinates -- 0.44 GB -- scales with the chunk)
    out["dropin_eval_image_one_chunk"] = guarded(lambda: bench_eval_image(ctx, scene, a, kw, outputs, 0.0, chunk=scene["H"] * scene["W"]))
    # training: 2000 rays x 90 samples, forward + loss + backward into the network AND the gather sources + Adam (train.py:147-188, 85-92)
    vols = [v.detach().clone().requires_grad_(True) for v in scene["vols"]]
    img_feat = scene["img_feat"].detach().clone().requires_grad_(True)
    conf = scene["confidence"].detach().clone().requires_grad_(True)

    def make_step(n_rays, opt, seed, vols=vols, img_feat=img_feat, scene=scene):
        tr = live_path_batch(scene, outputs, n_rays, 90, seed=seed)
        target = torch.rand(n_rays, 3, device=dev)

        def train_step():
            opt.zero_grad(set_to_none=True)
            for t in vols + [img_feat, conf]:
                t.grad = None
            rgb, depth = dropin_call(a, kw, scene, outputs, tr, confidence=conf, vols=vols, img_feat=img_feat)
            loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
            loss.backward()
            opt.step()
            return loss
        return train_step

    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))              # exactly what train.py builds (train.py:85-92)
    train_step = make_step(2000, opt, 4)
    dt = ctx.timed(train_step, steps, warmup)
    assert torch.isfinite(train_step()).all()
    uc_nerf_amd.set_training_precision("bf16x3")
    try:
        dt_b = ctx.timed(train_step, steps, warmup)
    finally:
        uc_nerf_amd.set_training_precision("f32")
    out["dropin_train_bf16x3_forward"] = {"ms_per_step": dt_b * 1e3, "value": 2000 / dt_b, "unit": "rays/s",
                                          "note": "the same step with the opt-in split-bf16 training forward (set_training_precision('bf16x3'))"}
    out["dropin_train"] = {"ms_per_step": dt * 1e3, "value": 2000 / dt, "unit": "rays/s", "rays": 2000, "samples_per_ray": 90, "dtype": "f32",
                           "optimizer": "torch.optim.Adam(grad_vars) as train.py:85-92 builds it",
                           "note": "rendering() forward (activations kept) + img/depth loss + backward into MLP parameters, cascade volumes, "
                                   "img_feats and confidence + Adam step; parameters and gradients live flat (uc_nerf_amd/flat.py): no concatenation, "
                                   "no per-tensor gradient copies; the weight stream is repacked once per step"}
    # the same step with the sources as a real producer's convolutions write them (channels_last_3d volumes, channels_last feature maps: read in place,
    # their gradients accumulated channel-last with the inputs' strides -- no scratch memset, no transposing add)
    def train_channels_last():
        prod, rec = conv_produced_sources(scene, dev)
        vols_c = [v.detach().requires_grad_(True) for v in prod["vols"]]
        feat_c = prod["img_feat"].detach().requires_grad_(True)
        sc = dict(scene, imgs=prod["imgs"])
        res = {}
        for n_rays, seed in ((2000, 4), (250, 5)):
            tr = live_path_batch(scene, outputs, n_rays, 90, seed=seed)
            target = torch.rand(n_rays, 3, device=dev)

            def step_c():
                opt.zero_grad(set_to_none=True)
                for t in vols_c + [feat_c, conf]:
                    t.grad = None
                rgb, depth = dropin_call(a, kw, sc, outputs, tr, confidence=conf, vols=vols_c, img_feat=feat_c)
                loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
                loss.backward()
                opt.step()
                return loss
            res[n_rays] = ctx.timed(step_c, steps * (1 if n_rays == 2000 else 2), warmup * (1 if n_rays == 2000 else 2))
            assert torch.isfinite(step_c()).all()
        from uc_nerf_amd import dropin as _dr
        ok = all(v.grad is not None and all(n == 1 or x == y for n, x, y in zip(v.shape, v.grad.stride(), v.stride())) for v in vols_c) and feat_c.grad is not None
        return dict(rec, ms_per_step=res[2000] * 1e3, ms_per_step_250=res[250] * 1e3, read_in_place=list(_dr.session_of(net).src.inplace),
                    gradients_have_the_inputs_strides=bool(ok),
                    note="dropin_train with conv-produced channels_last(_3d) sources handed to rendering() as they are")
    out["dropin_train_channels_last"] = guarded(train_channels_last)
    # the training side of the strong-scaling point: configs[4]'s 2000-ray batch over 8 GPUs = 250 rays per rank, measured on this one GPU
    step250 = make_step(250, opt, 5)
    dt250 = ctx.timed(step250, steps * 2, warmup * 2)
    out["strong_train_250"] = {"ms_per_step": dt250 * 1e3, "rays": 250, "samples_per_ray": 90, "ms_per_step_2000": dt * 1e3,
                               "projected_speedup_at_8_gpus": dt / dt250,
                               "note": "the rendering() training step on 250 rays x 90 (the per-GPU share of the 2000-ray batch at 8 GPUs) against the "
                                       "2000-ray step of the same run (source tensors unchanged between steps: their channel-last copies are reused; "
                                       "a producer that writes new volumes every step, train.py:136-163, adds the 25-us repack to both, or nothing "
                                       "with ops.ChannelLastSources); the gradient all-reduce (0.73 MB; 0.02 ms through a one-rank RCCL group) is NOT "
                                       "in either figure; the 250-ray step is HOST-bound (Python issue time), see ms_per_step_flat_adam; NOT a scaling measurement"}
    # the same steps with the flat optimizer (uc_nerf_amd.flat.FlatAdam: one fused launch over the flat buffer and the flat gradient).  The
    # 250-ray step is HOST-bound with torch's multi-tensor Adam over 30 tensors (~0.25 ms of Python per step against ~0.55 ms of GPU work,
    # profiles/r04_experiments.md): the optimizer is what the shard's step time hangs on
    from uc_nerf_amd.flat import FlatAdam
    net2 = kw["network_fn"]
    optf = FlatAdam(net2, lr=5e-4, betas=(0.9, 0.999))
    stepf = make_step(2000, optf, 4)
    dtf = ctx.timed(stepf, steps, warmup)
    assert torch.isfinite(stepf()).all()
    out["dropin_train_flat_adam"] = {"ms_per_step": dtf * 1e3, "value": 2000 / dtf, "unit": "rays/s",
                                     "note": "the same step with uc_nerf_amd.flat.FlatAdam(network_fn) in place of torch.optim.Adam(grad_vars): an opt-in, "
                                             "element-wise the same update"}
    stepf250 = make_step(250, optf, 5)
    dtf250 = ctx.timed(stepf250, steps * 2, warmup * 2)
    out["strong_train_250"].update(ms_per_step_flat_adam=dtf250 * 1e3, ms_per_step_2000_flat_adam=dtf * 1e3,
                                   projected_speedup_at_8_gpus_flat_adam=dtf / dtf250)

    # ... and as ONE HIP-graph replay per step (train_step.GraphedStep: zero_grad + rendering() + loss + backward + FlatAdam captured once): what is
    # left is the GPU time of the step
    def graphed():
        from uc_nerf_amd.train_step import GraphedStep
        res = {}
        for n_rays, seed in ((2000, 4), (250, 5)):
            optg = FlatAdam(net2, lr=5e-4, betas=(0.9, 0.999), capturable=True)
            g = GraphedStep(make_step(n_rays, optg, seed))
            dtg = ctx.timed(g.replay, steps * 2, warmup * 2)
            assert torch.isfinite(g.replay()).all()
            res[n_rays] = dtg
        # ... and on sources as a producer's convolutions write them (channels_last / channels_last_3d, read in place: no repack, no scratch fill, no
        # transposing add in the captured step)
        prod, _ = conv_produced_sources(scene, dev)
        vols_c = [v.detach().requires_grad_(True) for v in prod["vols"]]
        feat_c = prod["img_feat"].detach().requires_grad_(True)
        for n_rays, seed in ((2000, 4), (250, 5)):
            optg = FlatAdam(net2, lr=5e-4, betas=(0.9, 0.999), capturable=True)
            g = GraphedStep(make_step(n_rays, optg, seed, vols=vols_c, img_feat=feat_c, scene=dict(scene, imgs=prod["imgs"])))
            res[("cl", n_rays)] = ctx.timed(g.replay, steps * 2, warmup * 2)
            assert torch.isfinite(g.replay()).all()
        return {"ms_per_step_2000": res[2000] * 1e3, "ms_per_step_250": res[250] * 1e3, "projected_speedup_at_8_gpus": res[2000] / res[250],
                "channels_last_sources": {"ms_per_step_2000": res[("cl", 2000)] * 1e3, "ms_per_step_250": res[("cl", 250)] * 1e3,
                                          "projected_speedup_at_8_gpus": res[("cl", 2000)] / res[("cl", 250)]},
                "note": "the rendering() training step (forward + loss + backward + FlatAdam) captured into one HIP graph and replayed "
                        "(uc_nerf_amd.train_step.GraphedStep): an opt-in for static batch shapes; NOT a scaling measurement"}
    out["dropin_train_graphed"] = guarded(graphed)
    return out


def bench_train_dp(ctx, scene, sd, rays_per_rank, steps=30, warmup=8):
    """Data-parallel training steps: rendering() drop-in on this rank's rays, backward, ONE flat-bucket all-reduce over the
    process group (RCCL when backend = nccl), Adam.  Returns per-step time and the all-reduce's share."""
    from uc_nerf_amd import parallel as P
    from uc_nerf_amd.synthetic import live_path_batch
    dev = ctx.dev
    a, kw, net, grad_vars, outputs = dropin_setup(scene, sd, dev)
    tr = live_path_batch(scene, outputs, rays_per_rank, 90, seed=10 + ctx.rank)
    target = torch.rand(rays_per_rank, 3, device=dev)
    opt = torch.optim.Adam(grad_vars, lr=5e-4, betas=(0.9, 0.999))
    bucket = P.FlatGradBucket(grad_vars, n_scalars=1, collective_at_one=True)      # (one-rank rehearsal: the RCCL call is issued all the same)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    k = [-warmup]

    def step():
        opt.zero_grad(set_to_none=True)
        rgb, depth = dropin_call(a, kw, scene, outputs, tr)
        loss = torch.mean((rgb - target) ** 2) * 5.0 + 0.05 * torch.mean((depth - 2.0) ** 2)
        loss.backward()
        if k[0] >= 0:
            ev[k[0]][0].record()
        bucket.allreduce(1.0 / ctx.world, [loss.detach()])
        if k[0] >= 0:
            ev[k[0]][1].record()
        opt.step()
        k[0] += 1

    dt = ctx.timed(step, steps, warmup)
    ar_ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)[len(ev) // 2]
    return {"ms_per_step": dt * 1e3, "value": rays_per_rank * ctx.world / dt, "unit": "rays/s", "rays_per_gpu": rays_per_rank, "samples_per_ray": 90,
            "ranks_seen": ctx.dist.get_world_size() if ctx.dist is not None else 1, "backend": ctx.backend if ctx.dist is not None else "none",
            "allreduce_ms_median": ar_ms, "bucket_bytes": bucket.numel * 4, "bucket_path": bucket.last_path,
            "note": "rendering() fwd + bwd + one flat-bucket gradient all-reduce + Adam; allreduce_ms_median = everything FlatGradBucket.allreduce "
                    "does (bucket_path in_place: the gradients ARE the bucket -- scalar + flag copies, one scale, the collective, nothing after it)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # ~0.2 s of GPU time: the clocks need ~100 steps to settle
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step (weak) / global rays per step (strong)")
    ap.add_argument("--coarse", type=int, default=64)
    ap.add_argument("--fine", type=int, default=128)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): --rays is the GLOBAL batch, split over the GPUs (the north-star's 4096-ray strong-scaling target); "
                         "weak: --rays per GPU.  Identical at N = 1.")
    ap.add_argument("--mode", choices=["render", "train"], default="render")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="size of the CPU-baseline sample (0 = skip): the first N rays of the GPU's own seed-0 "
                    "batch; 4096 = the whole stated batch, five runs of ~5 s on 16 host threads")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--fused-min-rounds", type=int, default=0,
                    help="bf16x3_fused: passes with fewer tiles per wave than this take the two-kernel route (0: always the fused kernel)")
    ap.add_argument("--no-reuse", "--headline-only", dest="no_reuse", action="store_true",
                    help="skip the secondary measurements: profiling runs")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph (no per-kernel HIP events)")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16", "bf16x3_fused"], default="bf16x3_fused",
                    help="MLP arithmetic: exact fp32 MFMA; split-bf16 (3 bf16 MFMAs per product, fp32 accumulate, within the "
                         "1e-4 parity bar); plain bf16 (1 MFMA per product, NOT within the parity bar: PSNR reported); bf16x3_fused (default): "
                         "split-bf16 with the feature gather inside the MLP kernel -- one launch per pass, no feature buffer")
    args = ap.parse_args()
    self_launch(args.gpus, sys.argv[1:])          # N > 1 without a launcher: start the ranks (this process then only relays and exits)
    ctx = Ctx(args)
    rank, world, dev = ctx.rank, ctx.world, ctx.dev

    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to

    scene_cpu = make_scene(seed=0)
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    scene = scene_to(scene_cpu, dev)

    if args.mode == "train":
        rays = 2000 if args.rays == 4096 else args.rays
        per_rank = rays if args.scaling == "weak" else max(1, rays // world)
        res = bench_train_dp(ctx, scene, sd, per_rank, steps=args.steps, warmup=max(args.warmup, 5))
        if rank == 0:
            print(json.dumps({"metric": "training rays/sec (rendering() fwd+bwd, 90 samples, flat-bucket all-reduce, Adam)", "value": res["value"],
                              "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 5), "ms_per_step": res["ms_per_step"],
                              "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "live path: %d rays/GPU x 90 cascade samples, V=7 views 256x320" % per_rank,
                                         "parallelism": "ray-sharded x%d, one flat gradient bucket all-reduced per step" % world}, "train": res}), flush=True)
        if ctx.dist is not None:
            ctx.dist.barrier()
            ctx.dist.destroy_process_group()
        return

    rays = args.rays if args.scaling == "weak" else max(1, args.rays // world)           # per rank
    global_rays = rays * world
    renderer = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks,
                                  precision=args.precision, fused_min_rounds=args.fused_min_rounds)
    # this rank's shard of the global batch: contiguous block split
    xs_all, ys_all = random_pixels(global_rays, scene_cpu["H"], scene_cpu["W"], seed=0)
    xs = xs_all[rank * rays:(rank + 1) * rays].to(dev)
    ys = ys_all[rank * rays:(rank + 1) * rays].to(dev)
    noise = torch.rand(rays, args.coarse, generator=torch.Generator().manual_seed(100 + rank)).to(dev)
    barrier = ctx.barrier

    # The chip needs ~100 ms of load before its clocks settle (a 20-step run reads 7 % lower than a 200-step one): whatever
    # W is, at least 120 untimed steps run before the timed region; the extra ones are reported as `clock_settle_steps`.
    settle = max(0, 120 - args.warmup)
    for _ in range(settle + args.warmup):
        out = renderer.render(xs, ys, perturb=1.0, noise=noise)
    events = [[(ops.Event(), ops.Event()), (ops.Event(), ops.Event())] for _ in range(args.steps)]
    # The host only issues launches (0.15-0.4 ms of Python per 1 ms step) and must stay ahead of the GPU's queue: a
    # generational GC pass over the set-up objects (the CPU scene, the event list) stalls it for tens of milliseconds,
    # which the queue cannot cover.  Park everything allocated so far in the permanent generation.
    import gc
    gc.collect()
    gc.freeze()
    graph_ms = None
    if args.graph:                       # extra measurement: the same K steps as replays of one captured HIP graph
        g = renderer.capture(rays, perturb=1.0)
        graph_ms = ctx.timed(lambda: g(xs, ys, noise), args.steps, args.warmup) * 1e3
        og = g(xs, ys, noise)
        ref = renderer.render(xs, ys, perturb=1.0, noise=noise)
        assert torch.equal(og["rgb"], ref["rgb"]) and torch.equal(og["depth"], ref["depth"])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = renderer.render(xs, ys, perturb=1.0, noise=noise, events=events[k])
    barrier()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(out["rgb"]).all() and torch.isfinite(out["depth"]).all()

    # dominant kernel (mlp_fwd): HIP events recorded around its two launches per step, on the launch stream
    mlp_ms = sum(a.elapsed_ms(b) for step in events for a, b in step)
    launches = 2 * args.steps
    samples_per_step = rays * (args.coarse + args.coarse + args.fine)
    achieved = samples_per_step * args.steps * FLOP_PER_SAMPLE / (mlp_ms * 1e-3) / 1e12
    bf16 = args.precision != "f32"

    import threading
    line_once = threading.Lock()

    def finish(extra_, final=True):
        """Builds the ONE JSON line and prints it (rank 0).  final=False: called by the watchdog below when the secondary measurements of an N > 1 run
        did not come back -- the headline, measured before them, still goes out (nothing here then touches the GPU or a collective)."""
        if not line_once.acquire(blocking=False):      # (the watchdog and the main thread never both print)
            return
        psnr_vs_f32 = None
        if args.precision == "bf16" and final:         # outside the parity bar by construction: report the distance to the exact render
            r32 = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, precision="f32")
            ref = r32.pass_(out["rays_d"], out["z_fine"])            # same fine depths: isolates the network arithmetic
            mse = torch.mean((out["rgb"] - ref["rgb"]) ** 2).item()
            psnr_vs_f32 = {"psnr_db": -10.0 * math.log10(max(mse, 1e-20)),
                           "max_abs_rgb": (out["rgb"] - ref["rgb"]).abs().max().item(),
                           "max_abs_depth": (out["depth"] - ref["depth"]).abs().max().item()}
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, scripts/pmc_mlp.sh), which
        # cannot run inside this process: the figure is READ BACK from the committed profile and labelled as such
        traffic, traffic_source = None, None
        for name in (("r05_mlp_bf16_fused_hbm_traffic.json", "r04_mlp_bf16_fused_hbm_traffic.json", "r03_mlp_bf16_fused_hbm_traffic.json", "r02_mlp_bf16_fused_hbm_traffic.json") if args.precision == "bf16x3_fused" else
                     ("r02_mlp_bf16_hbm_traffic.json", "r01_mlp_bf16_hbm_traffic.json") if bf16 else ("r02_mlp_fwd_hbm_traffic.json", "r01_mlp_fwd_hbm_traffic.json")):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("bytes_per_launch")
                traffic_source = "profiles/%s (rocprofv3 PMC passes of this command on an earlier box; not measured in this run)" % name
                break

        if rank == 0:
            roof = {"bound": "mfma", "kernel": ("mlp_fwd_bf16_kernel (FUSED instantiation: feature gather + encoding + MLP)" if args.precision == "bf16x3_fused"
                                                else "mlp_fwd_bf16_kernel" if bf16 else "mlp_fwd_kernel"), "achieved": achieved,
                    "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_ms": mlp_ms / launches, "flop_per_launch_avg": samples_per_step * FLOP_PER_SAMPLE / 2}
            if bf16:
                # achieved / frac count ALGORITHMIC flops (one multiply-accumulate per weight and sample) against the dense bf16
                # peak; the split evaluation executes three bf16 MFMA products per algorithmic one (fp32-grade accuracy), so the
                # matrix pipe itself runs at `executed` TFLOP/s
                terms = 3 if args.precision in ("bf16x3", "bf16x3_fused") else 1
                ex = achieved * BF16X3_EXECUTED_FLOP_PER_SAMPLE / 3 * terms / FLOP_PER_SAMPLE
                roof.update(executed=ex, executed_frac=ex / peak,
                            note=("bf16x3: 3 bf16 MFMAs per algorithmic MAC; algorithmic ceiling = peak/3.003 = 832 TFLOP/s"
                                  + ("; this launch also does the pass's feature gather (no separate gather kernel, no feature buffer): `achieved` "
                                     "counts the MLP's flops only -- the kernel-only figure is two_kernel_pass.roofline"
                                     if args.precision == "bf16x3_fused" else ""))
                            if terms == 3 else "plain bf16: NOT within the 1e-4 parity bar, see parity_vs_f32")
                if terms == 3:
                    # what the three-MFMA formulation reaches on an MI355X board with this kernel's own instruction mix and nothing else (no gather, no
                    # encoding, no DMA, no barrier): measured by scripts/micro/headline_ceiling.hip on one box next to the kernel, READ BACK from the
                    # committed artifact (the micro-benchmark takes 30 s of its own; it is not run here)
                    cpath = os.path.join(ROOT, "profiles", "r05_headline_ceiling.json")
                    if os.path.exists(cpath):
                        with open(cpath) as f:
                            ceil = json.load(f)["ceiling_executed_tflops"]
                        roof.update(frac_of_measured_ceiling=ex / ceil["iii_plus_valu_5_per_mfma"],
                                    measured_ceiling={"executed_tflops": ceil["iii_plus_valu_5_per_mfma"], "with_one_barrier_per_ring_slot": ceil["iv_plus_barrier_per_slot"],
                                                      "bare_3_mfma_loop": ceil["i_bare_3mfma"], "as_frac_of_peak_algorithmic": ceil["iii_plus_valu_5_per_mfma"] / 3.003 / peak,
                                                      "source": "profiles/r05_headline_ceiling.json (.md): power-limited loops with the kernel's instruction ratios, "
                                                                "same geometry, measured on an earlier box of this round; not measured in this run"})
            line = {
                "metric": "rendered rays/sec (coarse+fine, 64+128 samples)",
                "value": global_rays * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "clock_settle_steps": settle, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                "scaling": args.scaling, "vs_baseline": None,
                "dtype": {"f32": "f32", "bf16x3": "f32 via split-bf16 (bf16x3) MFMA, f32 accumulate",
                          "bf16x3_fused": "f32 via split-bf16 (bf16x3) MFMA, f32 accumulate (feature gather fused into the MLP kernel)",
                          "bf16": "bf16 operands, f32 accumulate"}[args.precision], "data": "synthetic",
                "config": {"workload": "configs[1] shapes: %d rays/GPU x (%d coarse + %d fine -> %d merged) samples, V=7 views "
                                       "256x320, cascade volumes 48x64x80/32x128x160/8x256x320, UCNeRF D=6 W=128 random init"
                                       % (rays, args.coarse, args.fine, args.coarse + args.fine),
                           "global_rays": global_rays, "parallelism": "ray-sharded x%d, no data-path collective" % world,
                           "precision": args.precision},
                "roofline": roof,
                "mlp_share_of_step": mlp_ms / (dt * 1e3),
                "ranks_seen": ctx.dist.get_world_size() if ctx.dist is not None else 1,
                **({"hip_graph_ms_per_step": graph_ms} if graph_ms is not None else {}),
            }
            if psnr_vs_f32 is not None:
                line["parity_vs_f32"] = psnr_vs_f32
            line.update(extra_)
            if args.cpu_rays > 0 and world == 1 and final:      # (N = 1 only: the CPU legs take 10-30 s, during which other ranks would sit at the closing barrier)
                line["parity_vs_oracle"] = guarded(lambda: parity_vs_oracle(scene_cpu, sd, out, xs_all[:rays], ys_all[:rays], noise.cpu(), args.coarse, args.fine))
                line["cpu_baseline"] = cpu_baseline(scene_cpu, sd, args.cpu_rays, args.coarse, args.fine)
            print(json.dumps(line), flush=True)

    # N > 1: the secondary measurements below contain collectives (barriers, the gradient all-reduce).  Should one of them not come back -- a rank that
    # failed alone leaves the others waiting in a barrier -- the headline measured above must still reach the driver: a watchdog thread (it runs while
    # the main thread sits in a blocking HIP / RCCL call: those release the GIL) prints the line with what there is and ends the process.
    secondaries_done = threading.Event()
    if world > 1:
        limit = float(os.environ.get("UCNERF_BENCH_SECONDARY_LIMIT_S", "240"))

        def watchdog():
            if not secondaries_done.wait(limit):
                try:
                    finish({"secondaries": {"error": "did not finish within %.0f s on rank %d: the line carries the headline only" % (limit, rank)}}, final=False)
                finally:
                    os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()

    extra = {}
    secondary = not args.no_reuse
    if secondary:
        # reuse_coarse: the fine pass evaluates only the 128 new depths (bit-identical renders, tests/test_hip_pipeline.py)
        def reuse():
            dt2 = ctx.timed(lambda: renderer.render(xs, ys, perturb=1.0, noise=noise, reuse_coarse=True), args.steps, args.warmup)
            out2 = renderer.render(xs, ys, perturb=1.0, noise=noise, reuse_coarse=True)
            assert torch.equal(out2["rgb"], out["rgb"]) and torch.equal(out2["depth"], out["depth"])
            return {"value": global_rays / dt2, "unit": "rays/s", "ms_per_step": dt2 * 1e3,
                    "note": "NOT the headline: fine pass evaluates only the 128 new depths and re-uses the coarse pass's outputs for the "
                            "64 coarse ones; renders verified bit-identical in this run"}
        extra["reuse_coarse"] = guarded(reuse)

        def other_precisions():
            res = {}
            r32 = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, precision="f32")
            if args.precision != "f32":          # the exact-fp32 kernel, driver-timed in the same line (roofline of ITS dominant kernel)
                ev32 = [[(ops.Event(), ops.Event()), (ops.Event(), ops.Event())] for _ in range(40)]
                k32 = [-10]

                def step32():
                    r32.render(xs, ys, perturb=1.0, noise=noise, events=ev32[k32[0]] if k32[0] >= 0 else None)
                    k32[0] += 1
                dt32 = ctx.timed(step32, 40, 10)
                ms32 = sum(a.elapsed_ms(b) for st in ev32 for a, b in st)
                ach32 = samples_per_step * 40 * FLOP_PER_SAMPLE / (ms32 * 1e-3) / 1e12
                res["exact_f32"] = {"value": global_rays / dt32, "unit": "rays/s", "ms_per_step": dt32 * 1e3,
                                    "roofline": {"bound": "mfma", "kernel": "mlp_fwd_kernel", "achieved": ach32, "peak": PEAK_F32_MFMA_TFLOPS,
                                                 "unit": "TFLOP/s", "frac": ach32 / PEAK_F32_MFMA_TFLOPS, "avg_launch_ms": ms32 / 80},
                                    "note": "NOT the headline: precision='f32' (v_mfma_f32_32x32x2_f32, exact fp32 products)"}
            if args.precision in ("bf16x3", "bf16x3_fused"):       # plain-bf16 operands (the dtype configs[1] names): NOT within the 1e-4 parity bar
                rp = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks, precision="bf16")
                dt3 = ctx.timed(lambda: rp.render(xs, ys, perturb=1.0, noise=noise), args.steps, args.warmup)
                op = rp.render(xs, ys, perturb=1.0, noise=noise)
                ref = r32.pass_(op["rays_d"], op["z_fine"])
                mse = torch.mean((op["rgb"] - ref["rgb"]) ** 2).item()
                res["plain_bf16"] = {"value": global_rays / dt3, "unit": "rays/s", "ms_per_step": dt3 * 1e3,
                                     "psnr_db_vs_f32": -10.0 * math.log10(max(mse, 1e-20)),
                                     "max_abs_rgb_vs_f32": (op["rgb"] - ref["rgb"]).abs().max().item(),
                                     "note": "NOT the headline and NOT within the 1e-4 parity bar: one bf16 MFMA per product (precision='bf16')"}
                # the same split with fp16 terms (ucnerf_mlp_config.operand = 1, ABI v6): three f16 MFMAs per product, 11 significant bits per term
                if ops.split_operand() == "bf16":
                    ops.set_split_operand("fp16")
                    try:
                        rh = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks, precision=args.precision)
                    finally:
                        ops.set_split_operand("bf16")
                    evh = [[(ops.Event(), ops.Event()), (ops.Event(), ops.Event())] for _ in range(args.steps)]
                    kh = [-args.warmup]

                    def steph():
                        rh.render(xs, ys, perturb=1.0, noise=noise, events=evh[kh[0]] if kh[0] >= 0 else None)
                        kh[0] += 1
                    dth = ctx.timed(steph, args.steps, args.warmup)
                    msh = sum(a.elapsed_ms(b) for st in evh for a, b in st)
                    oh = rh.render(xs, ys, perturb=1.0, noise=noise)
                    refh = r32.pass_(oh["rays_d"], oh["z_fine"])
                    o3 = renderer.render(xs, ys, perturb=1.0, noise=noise)
                    ref3 = r32.pass_(o3["rays_d"], o3["z_fine"])
                    res["fp16_terms"] = {"value": global_rays / dth, "unit": "rays/s", "ms_per_step": dth * 1e3, "avg_launch_ms": msh / (2 * args.steps),
                                         "max_abs_rgb_vs_f32": (oh["rgb"] - refh["rgb"]).abs().max().item(),
                                         "max_abs_depth_vs_f32": (oh["depth"] - refh["depth"]).abs().max().item(),
                                         "headline_max_abs_rgb_vs_f32": (o3["rgb"] - ref3["rgb"]).abs().max().item(),
                                         "headline_max_abs_depth_vs_f32": (o3["depth"] - ref3["depth"]).abs().max().item(),
                                         "note": "NOT the headline: the headline's route and precision with fp16 instead of bf16 terms (set_split_operand('fp16'): "
                                                 "~21-bit products at the same matrix-core rate; fp16's range -- activations beyond 131 008 are clamped, never infinite). Distances: to the "
                                                 "exact-f32 kernel on the same fine depths, this render's and the headline's"}
            return res
        def other_route():
            """Row f1: the same step on the other route -- the gather inside the MLP kernel (one launch per pass, the default) against
            the two-kernel pass (gather -> feature buffer -> MLP)."""
            fused_is_headline = args.precision == "bf16x3_fused"
            prec = "bf16x3" if fused_is_headline else "bf16x3_fused"
            rf = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks, precision=prec)
            evf = [[(ops.Event(), ops.Event()), (ops.Event(), ops.Event())] for _ in range(args.steps)]
            kf = [-args.warmup]

            def stepf():
                rf.render(xs, ys, perturb=1.0, noise=noise, events=evf[kf[0]] if kf[0] >= 0 else None)
                kf[0] += 1
            dtf = ctx.timed(stepf, args.steps, args.warmup)
            msf = sum(a.elapsed_ms(b) for st in evf for a, b in st)
            achf = samples_per_step * args.steps * FLOP_PER_SAMPLE / (msf * 1e-3) / 1e12
            of = rf.render(xs, ys, perturb=1.0, noise=noise)
            # (bit-equal fine depths: a depth that moved by 1e-5 can carry a sample across a source image's border -- a mask flip worth 1e-2 of rgb on
            #  this synthetic scene -- so "nearly equal" depths say nothing about the two kernels' arithmetic)
            same = (of["z_fine"] == out["z_fine"]).all(-1)
            exf = achf * BF16X3_EXECUTED_FLOP_PER_SAMPLE / FLOP_PER_SAMPLE
            return {"value": global_rays / dtf, "unit": "rays/s", "ms_per_step": dtf * 1e3, "precision": prec,
                    "roofline": {"bound": "mfma", "kernel": "mlp_fwd_bf16_kernel" + ("" if fused_is_headline else " (gather fused)"),
                                 "achieved": achf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achf / PEAK_BF16_MFMA_TFLOPS,
                                 "executed": exf, "executed_frac": exf / PEAK_BF16_MFMA_TFLOPS, "avg_launch_ms": msf / (2 * args.steps)},
                    "max_abs_rgb_vs_headline_on_rays_with_equal_depths": (of["rgb"] - out["rgb"]).abs().amax(-1)[same].max().item() if bool(same.any()) else None,
                    "rays_with_equal_fine_depths": same.float().mean().item(),
                    "note": ("NOT the headline: the two-kernel pass (feat_gather_cl -> tiled feature buffer -> mlp_fwd_bf16); its MLP launch does "
                             "no gather, so its roofline fraction is the kernel-only figure" if fused_is_headline else
                             "NOT the headline: the gather inside the MLP kernel; `achieved` counts the MLP's algorithmic flops only")}
        if world == 1:        # (N > 1: the scaling series needs `value`; the single-GPU fields are not repeated on every rank)
            extra.update(guarded(other_precisions, "other_precisions"))
            if args.precision in ("bf16x3", "bf16x3_fused"):
                extra["two_kernel_pass" if args.precision == "bf16x3_fused" else "gather_fused"] = guarded(other_route)

        def micro():
            """SURVEY.md 8(d) secondary micro-bench (north_star's literal '4096-ray x 128-sample'): single pass, S = 128,
            PRECOMPUTED input_feat ~ randn [4096,128,97] -> positional encoding + MLP + compositing only."""
            n, S = 4096, 128
            g = torch.Generator().manual_seed(5)
            feats = torch.randn(n * S, 97, generator=g)
            feats[:, -1] = torch.rand(n * S, generator=g)
            feats = feats.to(dev)
            pts = torch.rand(n * S, 3, generator=g).to(dev)
            dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
            z = torch.sort(1 + 3 * torch.rand(n, S, generator=g), -1)[0].to(dev)
            pw, ws = renderer.pw, renderer.wstream
            if args.precision == "bf16x3_fused":          # precomputed features: the stand-alone MLP entry, a stream in the two-kernel operand order
                pw = ops.PackedWeights.get(pw.cfg.n_src, 0, dev, "bf16x3")
                ws = pw.pack(flat_params_of(sd).to(dev))

            def step():
                raw = ops.mlp_fwd(pw, ws, pts, dirs, feats, S)
                return ops.composite_fwd(raw.view(n, S, 4), z)
            torch.cuda.synchronize()                       # the 203 MB of uploads are done before anything is timed
            dtm = min(ctx.timed(step, 60, 20), ctx.timed(step, 60, 0))       # (one run in four read 3x high on its first block)
            return {"value": n * world / dtm, "unit": "rays/s", "ms_per_step": dtm * 1e3, "rays": n, "samples_per_ray": S, "precision": args.precision,
                    "tflops_algorithmic": n * S * FLOP_PER_SAMPLE / dtm / 1e12,
                    "note": "single pass on precomputed [4096,128,97] features: PE + MLP + composite (no gather, no sampling)"}
        if world == 1:
            extra["micro_4096x128_precomputed_feats"] = guarded(micro)

        def fused_train(fwd_precision="f32"):
            """A training-style step of the library's fused pass: source repack + forward (activations kept) + full backward
            (parameters, volumes, image features, confidence), 1024 rays x 128 depths; forward in exact f32 or split-bf16."""
            n_t, s_t = min(1024, int(xs.shape[0])), 128
            rt = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, precision=fwd_precision)
            flat_t = flat_params_of(sd).to(dev)
            rd_t, _, _ = ops.ray_gen(rt.K_host, rt.c2w_host, xs=xs[:n_t], ys=ys[:n_t])
            z_t, _ = ops.sample_stratified(None, s_t, n=n_t, near=rt.near_host, far=rt.far_host, device=dev)
            g_rgb, g_depth = torch.randn(n_t, 3, device=dev), torch.randn(n_t, device=dev)

            def train_step():
                rt.pass_.repack_sources()
                kept = rt.pass_(rd_t, z_t, keep=("raw", "feats"))
                return rt.pass_.backward(rd_t, z_t, kept, g_rgb, g_depth, flat_t)
            dt4 = ctx.timed(train_step, 30, 5)
            assert all(torch.isfinite(t).all() for t in train_step() if t is not None)
            return {"ms_per_step": dt4 * 1e3, "value": n_t / dt4, "unit": "rays/s", "rays": n_t, "samples_per_ray": s_t,
                    "dtype": "f32 forward (activations kept as 24-bit floats), bf16x3 backward" if fwd_precision == "f32" else "forward bf16x3 (activations kept as 24-bit floats), backward bf16x3 gradient chain + weight-gradient launch",
                    "note": "NOT the headline: source repack + forward + full backward of one fused render pass on one GPU "
                            "(no optimizer, no collective)"}
        def fused_train_channel_last():
            """The same training-style step with the sources handed over channel-last (ops.ChannelLastSources): no repack, and the source gradients
            accumulated in a buffer of the same layout -- no scratch memset, no transposing add."""
            n_t, s_t = min(1024, int(xs.shape[0])), 128
            cl = ops.ChannelLastSources.from_reference_layout(scene["vols"], scene["imgs"], scene["img_feat"])
            rt = CoarseFineRenderer(dict(scene, vols=cl.vols, imgs=cl.imgs, img_feat=cl.img_feat), flat_params_of(sd).to(dev), args.coarse, args.fine, precision="f32")
            assert rt.src.zero_copy
            flat_t = flat_params_of(sd).to(dev)
            rd_t, _, _ = ops.ray_gen(rt.K_host, rt.c2w_host, xs=xs[:n_t], ys=ys[:n_t])
            z_t, _ = ops.sample_stratified(None, s_t, n=n_t, near=rt.near_host, far=rt.far_host, device=dev)
            g_rgb, g_depth = torch.randn(n_t, 3, device=dev), torch.randn(n_t, device=dev)

            def train_step():
                kept = rt.pass_(rd_t, z_t, keep=("raw", "feats"))
                return rt.pass_.backward(rd_t, z_t, kept, g_rgb, g_depth, flat_t)
            dt4 = ctx.timed(train_step, 30, 5)
            assert all(torch.isfinite(t).all() for t in train_step() if t is not None)
            return {"ms_per_step": dt4 * 1e3, "value": n_t / dt4, "unit": "rays/s", "rays": n_t, "samples_per_ray": s_t,
                    "note": "NOT the headline: train_step with zero-copy channel-last sources and channel-last source gradients (no repack, no transposing add)"}
        if world == 1:
            extra["train_step"] = guarded(fused_train)
            extra["train_step_bf16x3_forward"] = guarded(lambda: fused_train("bf16x3"))
            extra["train_step_channel_last_sources"] = guarded(fused_train_channel_last)

        if world == 1:
            extra.update(guarded(lambda: bench_dropin(ctx, scene, sd), "dropin"))

            def strong_512():
                """The per-GPU share of the north-star's strong-scaling point (4096 rays over 8 GPUs) measured on this one GPU, LIKE FOR LIKE:
                each ratio divides a 4096-ray step by a 512-ray step under the SAME source rule (repacked in every step / constant)."""
                n = 512
                x5, y5, nz5 = xs[:n].contiguous(), ys[:n].contiguous(), noise[:n].contiguous()
                dt5 = ctx.timed(lambda: renderer.render(x5, y5, perturb=1.0, noise=nz5, repack=False), 300, 60)
                dt5r = ctx.timed(lambda: renderer.render(x5, y5, perturb=1.0, noise=nz5, repack=True), 300, 60)
                dt40 = ctx.timed(lambda: renderer.render(xs, ys, perturb=1.0, noise=noise, repack=False), 100, 20)
                dt40r = ctx.timed(lambda: renderer.render(xs, ys, perturb=1.0, noise=noise, repack=True), 100, 20)
                # the same two steps as replays of ONE captured HIP graph each; the caller writes into the graph's static inputs once, a replay
                # is then a single hipGraphLaunch -- no per-replay copies
                g5, g5r = renderer.capture(n, perturb=1.0, repack=False), renderer.capture(n, perturb=1.0, repack=True)
                for g in (g5, g5r):
                    g.inputs["xs"].copy_(x5); g.inputs["ys"].copy_(y5); g.inputs["noise"].copy_(nz5)
                dt5g, dt5gr = ctx.timed(lambda: g5(), 300, 60), ctx.timed(lambda: g5r(), 300, 60)
                og, ref = g5(), renderer.render(x5, y5, perturb=1.0, noise=nz5, repack=False)
                assert torch.equal(og["rgb"], ref["rgb"]) and torch.equal(og["depth"], ref["depth"])
                # the zero-copy route: the sources handed over channel-last (ops.ChannelLastSources views, what a producer writing that layout gives):
                # the pass reads them in place -- there is no repack to hoist -- and renders bit-identically (checked here)
                cl = ops.ChannelLastSources.from_reference_layout(scene["vols"], scene["imgs"], scene["img_feat"])
                rz = CoarseFineRenderer(dict(scene, vols=cl.vols, imgs=cl.imgs, img_feat=cl.img_feat), flat_params_of(sd).to(dev), args.coarse, args.fine,
                                        max_blocks=args.max_blocks, precision=args.precision)
                assert rz.src.zero_copy
                dt5z = ctx.timed(lambda: rz.render(x5, y5, perturb=1.0, noise=nz5), 300, 60)
                dt40z = ctx.timed(lambda: rz.render(xs, ys, perturb=1.0, noise=noise), 100, 20)
                oz = rz.render(x5, y5, perturb=1.0, noise=nz5)
                assert torch.equal(oz["rgb"], ref["rgb"]) and torch.equal(oz["depth"], ref["depth"])
                # ... and with the sources as REAL torch convolutions write them (nn.Conv3d in channels_last_3d, nn.Conv2d in channels_last: each tensor its
                # own allocation, recognised by stride, ABI v5) against the SAME values handed over channel-major and repacked in every step
                prod, prec_ = conv_produced_sources(scene, dev)
                rt_ = CoarseFineRenderer(dict(scene, **prod), flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks, precision=args.precision)
                major = {"vols": [v.contiguous() for v in prod["vols"]], "imgs": prod["imgs"].contiguous(), "img_feat": prod["img_feat"].contiguous()}
                rm_ = CoarseFineRenderer(dict(scene, **major), flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks, precision=args.precision)
                dt5t = ctx.timed(lambda: rt_.render(x5, y5, perturb=1.0, noise=nz5), 300, 60)
                dt40t = ctx.timed(lambda: rt_.render(xs, ys, perturb=1.0, noise=noise), 100, 20)
                dt5m = ctx.timed(lambda: rm_.render(x5, y5, perturb=1.0, noise=nz5, repack=True), 300, 60)
                ot, om = rt_.render(x5, y5, perturb=1.0, noise=nz5), rm_.render(x5, y5, perturb=1.0, noise=nz5, repack=True)
                tcl = dict(prec_, read_in_place=list(rt_.src.inplace), zero_copy=bool(rt_.src.zero_copy), ms_per_step=dt5t * 1e3, ms_per_step_4096=dt40t * 1e3,
                           ms_per_step_same_values_repacked=dt5m * 1e3, projected_speedup_at_8_gpus=dt40t / dt5t if rays == 4096 else None,
                           equals_repacked_route=bool(torch.equal(ot["rgb"], om["rgb"]) and torch.equal(ot["depth"], om["depth"])),
                           note="sources produced by nn.Conv3d(...).to(memory_format=channels_last_3d) / nn.Conv2d(...).to(memory_format=channels_last) on this GPU, "
                                "handed to the renderer as they are: read in place when the strides are the channel-last ones (read_in_place: vol1, vol2, vol3, img_feat, "
                                "imgs), repacked per source otherwise; projection = this run's 4096-ray step over its 512-ray step under the same rule")
                # how many passes of one step composite / re-sample / generate their rays inside the gather-fused launch (the library's own counter)
                from uc_nerf_amd import _lib as _L
                c0 = _L.lib().ucnerf_fused_tail_launches()
                renderer.render(x5, y5, perturb=1.0, noise=nz5, repack=False)
                tail5 = int(_L.lib().ucnerf_fused_tail_launches() - c0)
                renderer.render(xs, ys, perturb=1.0, noise=noise, repack=False)
                tail40 = int(_L.lib().ucnerf_fused_tail_launches() - c0) - tail5
                full = rays == 4096
                return {"ms_per_step": dt5 * 1e3, "ms_per_step_with_repack": dt5r * 1e3, "ms_per_step_zero_copy_sources": dt5z * 1e3,
                        "ms_per_step_4096_zero_copy_sources": dt40z * 1e3, "zero_copy_equals_repacked_route": True, "rays": n,
                        "ms_per_step_4096_constant_sources": dt40 * 1e3, "ms_per_step_4096_with_repack": dt40r * 1e3,
                        "hip_graph_ms_per_step": dt5g * 1e3, "hip_graph_ms_per_step_with_repack": dt5gr * 1e3,
                        "projected_speedup_at_8_gpus": {"with_repack": dt40r / dt5r if full else None, "constant_sources": dt40 / dt5 if full else None,
                                                        "zero_copy_sources": dt40z / dt5z if full else None},
                        "target_ms_per_step_for_6x": {"with_repack": dt40r / 6 * 1e3, "constant_sources": dt40 / 6 * 1e3, "zero_copy_sources": dt40z / 6 * 1e3},
                        "passes_on_the_tail_route": {"512_rays": tail5, "4096_rays": tail40}, "torch_channels_last": tcl,
                        "note": "512 rays x (64+128) on ONE GPU; each projection divides this run's 4096-ray step by the 512-ray step under the same "
                                "source rule (with_repack: channel-last source copies rebuilt inside both steps -- the headline rule; constant_sources: "
                                "rebuilt in neither; zero_copy_sources: sources handed over channel-last, ops.ChannelLastSources, read in place); no collective is in either figure (rendering needs none; the gather of 80 KB of outputs is ignored); "
                                "the 512-ray step is two launches (its passes generate, composite and re-sample their rays inside the gather-fused launch: "
                                "ucnerf_fused_tail_launches, DESIGN.md 4.4), the 4096-ray step five; NOT a scaling measurement"}
            extra["strong_512"] = guarded(strong_512)

            def constant_sources():
                """The headline step when the sources do not change between batches -- the batches of ONE image, as in the reference's
                evaluation loop (train.py:254-272: the feature volumes are built once per image, rendering() is called per ray chunk)."""
                per_image = max(1, -(-scene_cpu["H"] * scene_cpu["W"] // global_rays))
                k = [0]

                def step():
                    renderer.render(xs, ys, perturb=1.0, noise=noise, repack=(k[0] % per_image == 0))
                    k[0] += 1
                dt6 = ctx.timed(step, args.steps, args.warmup)
                return {"value": global_rays / dt6, "unit": "rays/s", "ms_per_step": dt6 * 1e3, "batches_per_image": per_image,
                        "note": "NOT the headline (which repacks the sources in every step): channel-last source copies rebuilt once per "
                                "%d x %d image = every %d batches" % (scene_cpu["H"], scene_cpu["W"], per_image)}
            extra["constant_sources"] = guarded(constant_sources)

            def bf16_sources():
                """configs[4]'s "fp32 MLP / bf16 features" taken literally: the same step with the channel-last source copies held in bf16
                (half the bytes per gather corner and per repack).  Quality against the fp32-source render of the same rays and depths."""
                rb = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks,
                                        precision=args.precision, sources_bf16=True)
                dt7 = ctx.timed(lambda: rb.render(xs, ys, perturb=1.0, noise=noise), args.steps, args.warmup)
                ob = rb.render(xs, ys, perturb=0.0)
                of = renderer.render(xs, ys, perturb=0.0)
                mse = torch.mean((ob["rgb"] - of["rgb"]) ** 2).item()
                return {"value": global_rays / dt7, "unit": "rays/s", "ms_per_step": dt7 * 1e3,
                        "psnr_db_vs_fp32_sources": -10.0 * math.log10(max(mse, 1e-20)),
                        "max_abs_rgb_vs_fp32_sources": (ob["rgb"] - of["rgb"]).abs().max().item(),
                        "note": "NOT the headline and NOT within the 1e-4 parity bar: gather sources rounded to bf16 in their channel-last copies "
                                "(GatherSources(cl_bf16=True) / set_source_precision('bf16')); MLP and compositing unchanged"}
            extra["bf16_sources"] = guarded(bf16_sources)

            def hip_graph_replay():
                """The headline step (source repack included) captured once into a HIP graph and replayed: the same launches without the
                host's per-launch work (`CoarseFineRenderer.capture`, replay asserted bit-identical to the eager step)."""
                g = renderer.capture(rays, perturb=1.0)
                og = g(xs, ys, noise)                       # (fills the graph's static inputs once; the timed replays copy nothing)
                dtg = ctx.timed(lambda: g(), args.steps, args.warmup)
                og = g()
                ref = renderer.render(xs, ys, perturb=1.0, noise=noise)
                assert torch.equal(og["rgb"], ref["rgb"]) and torch.equal(og["depth"], ref["depth"])
                return {"value": global_rays / dtg, "unit": "rays/s", "ms_per_step": dtg * 1e3,
                        "note": "NOT the headline (which issues its launches eagerly, with HIP events around the MLP launches): one graph replay per step"}
            if not args.graph:
                extra["hip_graph_replay"] = guarded(hip_graph_replay)
        else:
            def other_scaling():
                """The step under the OTHER scaling rule, so that one N > 1 line carries both: weak = --rays per GPU, strong = --rays in all."""
                o_rays = args.rays if args.scaling == "strong" else max(1, args.rays // world)          # per rank
                xo_all, yo_all = random_pixels(o_rays * world, scene_cpu["H"], scene_cpu["W"], seed=0)
                xo, yo = xo_all[rank * o_rays:(rank + 1) * o_rays].to(dev), yo_all[rank * o_rays:(rank + 1) * o_rays].to(dev)
                no = torch.rand(o_rays, args.coarse, generator=torch.Generator().manual_seed(100 + rank)).to(dev)
                dto = ctx.timed(lambda: renderer.render(xo, yo, perturb=1.0, noise=no), args.steps, args.warmup)
                return {"value": o_rays * world / dto, "unit": "rays/s", "ms_per_step": dto * 1e3, "rays_per_gpu": o_rays, "global_rays": o_rays * world,
                        "scaling": "weak" if args.scaling == "strong" else "strong",
                        "note": "NOT the headline: the same step under the other scaling rule (barrier + MAX over ranks as the headline)"}
            extra["weak" if args.scaling == "strong" else "strong"] = guarded(other_scaling)
            extra["train_dp"] = guarded(lambda: bench_train_dp(ctx, scene, sd, 2000 if args.scaling == "weak" else max(1, 2000 // world)))
        if world == 1 and ctx.dist is not None:       # UCNERF_BENCH_GROUP_AT_1: the data-parallel step through a one-rank process group
            extra["train_dp"] = guarded(lambda: bench_train_dp(ctx, scene, sd, 2000))

    secondaries_done.set()
    finish(extra)
    if ctx.dist is not None:
        if world > 1:                # the line is out: a closing barrier that does not come back must not keep the launcher waiting
            t_ = threading.Timer(60.0, lambda: os._exit(0))
            t_.daemon = True
            t_.start()
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
