#!/usr/bin/env python3
"""Headline benchmark: rendered rays/sec, 64 coarse + 128 fine samples (the fine pass evaluates the merged
192), 4096 synthetic rays per GPU (BASELINE.json configs[1]).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU.  Rays are independent, so the batch is sharded across ranks with no data-path
collective (weak scaling: 4096 rays per GPU); the only collectives are the barrier and the MAX of the
timed interval.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 2 * 147328          # MLP multiply-accumulates per sample evaluation
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
# bf16x3: every algorithmic multiply-accumulate is three bf16 MFMA multiply-accumulates on K padded to 16 per step
BF16X3_EXECUTED_FLOP_PER_SAMPLE = 3 * 2 * 72 * 16 * 128


def cpu_baseline(scene_cpu, sd, n_rays, n_coarse, n_fine, budget_s=20.0):
    """The CPU oracle (torch-CPU restatement of the reference path, all host cores) on a bounded sample of the
    same workload.  Test infrastructure used ONLY as the reported baseline."""
    from oracle import ucnerf_oracle as O
    from uc_nerf_amd.synthetic import random_pixels
    # the GPU box gives one GPU tenant a 16-core share of the host; more intra-op threads than that only
    # oversubscribes (measured: 256 threads -> 1.5 rays/s)
    cores = min(os.cpu_count() or 1, int(os.environ.get("UCNERF_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    xs, ys = random_pixels(n_rays, scene_cpu["H"], scene_cpu["W"], seed=1)
    noise = torch.rand(n_rays, n_coarse, generator=torch.Generator().manual_seed(2))
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
            "sample": "%d rays x (%d+%d) samples of the same synthetic scene, torch-CPU fp32 oracle, median of %d runs"
                      % (n_rays, n_coarse, n_fine, len(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # ~0.2 s of GPU time: the clocks need ~100 steps to settle
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step")
    ap.add_argument("--coarse", type=int, default=64)
    ap.add_argument("--fine", type=int, default=128)
    ap.add_argument("--cpu-rays", type=int, default=512, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--no-reuse", action="store_true", help="skip the secondary measurements (reuse_coarse, plain bf16): profiling runs")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph (no per-kernel HIP events)")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16"], default="bf16x3",
                    help="MLP arithmetic: exact fp32 MFMA; split-bf16 (3 bf16 MFMAs per product, fp32 accumulate, within the "
                         "1e-4 parity bar); plain bf16 (1 MFMA per product, NOT within the parity bar: PSNR reported)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    # one rank per GPU; UCNERF_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks (ranks then
    # share devices round-robin and the barrier / MAX go through gloo) -- the driver's runs use RCCL ("nccl")
    backend = os.environ.get("UCNERF_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if local >= n_dev and backend == "nccl":
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs visible" % (local, n_dev))
    local_dev = local % n_dev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from uc_nerf_amd import ops
    from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to

    scene_cpu = make_scene(seed=0)
    sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
    scene = scene_to(scene_cpu, dev)
    renderer = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks,
                                  precision=args.precision)
    # this rank's shard of the global batch of rays*world pixels: contiguous block split
    xs_all, ys_all = random_pixels(args.rays * world, scene_cpu["H"], scene_cpu["W"], seed=0)
    xs = xs_all[rank * args.rays:(rank + 1) * args.rays].to(dev)
    ys = ys_all[rank * args.rays:(rank + 1) * args.rays].to(dev)
    noise = torch.rand(args.rays, args.coarse, generator=torch.Generator().manual_seed(100 + rank)).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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
        g = renderer.capture(args.rays, perturb=1.0)
        for _ in range(args.warmup):
            og = g(xs, ys, noise)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            og = g(xs, ys, noise)
        barrier()
        graph_ms = (time.perf_counter() - t0) / args.steps * 1e3
        ref = renderer.render(xs, ys, perturb=1.0, noise=noise)
        assert torch.equal(og["rgb"], ref["rgb"]) and torch.equal(og["depth"], ref["depth"])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = renderer.render(xs, ys, perturb=1.0, noise=noise, events=events[k])
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(out["rgb"]).all() and torch.isfinite(out["depth"]).all()

    # secondary number (not `value`): the same steps with the fine pass re-using the coarse pass's network outputs for the
    # 64 coarse depths (bit-identical renders, tests/test_hip_pipeline.py) -- 192 instead of 256 evaluations per ray
    dt2 = None
    if not args.no_reuse:
        for _ in range(args.warmup):
            out2 = renderer.render(xs, ys, perturb=1.0, noise=noise, reuse_coarse=True)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            out2 = renderer.render(xs, ys, perturb=1.0, noise=noise, reuse_coarse=True)
        barrier()
        dt2 = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = t.item()
        assert torch.equal(out2["rgb"], out["rgb"]) and torch.equal(out2["depth"], out["depth"])

    # second secondary number: plain-bf16 operands (the dtype BASELINE.json's configs[1] names), which is NOT within the 1e-4
    # parity bar -- reported with its PSNR against the exact-f32 render of the same depths
    plain = None
    if not args.no_reuse and args.precision == "bf16x3":
        rp = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, max_blocks=args.max_blocks, precision="bf16")
        for _ in range(args.warmup):
            op = rp.render(xs, ys, perturb=1.0, noise=noise)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            op = rp.render(xs, ys, perturb=1.0, noise=noise)
        barrier()
        dt3 = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt3], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt3 = t.item()
        r32 = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, precision="f32")
        ref = r32.pass_(op["rays_d"], op["z_fine"])
        mse = torch.mean((op["rgb"] - ref["rgb"]) ** 2).item()
        plain = {"value": args.rays * world * args.steps / dt3, "unit": "rays/s", "ms_per_step": dt3 / args.steps * 1e3,
                 "psnr_db_vs_f32": -10.0 * __import__("math").log10(max(mse, 1e-20)),
                 "max_abs_rgb_vs_f32": (op["rgb"] - ref["rgb"]).abs().max().item(),
                 "note": "NOT the headline and NOT within the 1e-4 parity bar: one bf16 MFMA per product (precision='bf16')"}

    # third secondary number: a training-style step on this GPU -- source repack + forward + full backward (parameters,
    # volumes, image features, confidence) of one fused render pass, 1024 rays x 128 depths, exact-f32 path
    train = None
    if not args.no_reuse:
        n_t, s_t = min(1024, int(xs.shape[0])), 128
        rt = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, precision="f32")
        flat_t = flat_params_of(sd).to(dev)
        rd_t, _, _ = ops.ray_gen(rt.K_host, rt.c2w_host, xs=xs[:n_t], ys=ys[:n_t])
        z_t, _ = ops.sample_stratified(None, s_t, n=n_t, near=rt.near_host, far=rt.far_host, device=dev)
        g_rgb, g_depth = torch.randn(n_t, 3, device=dev), torch.randn(n_t, device=dev)

        def train_step():
            rt.pass_.repack_sources()
            kept = rt.pass_(rd_t, z_t, keep=("raw", "feats"))
            return rt.pass_.backward(rd_t, z_t, kept, g_rgb, g_depth, flat_t)

        for _ in range(5):
            grads = train_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(30):
            grads = train_step()
        barrier()
        dt4 = (time.perf_counter() - t0) / 30
        assert all(torch.isfinite(t).all() for t in grads if t is not None)
        train = {"ms_per_step": dt4 * 1e3, "value": n_t / dt4, "unit": "rays/s", "rays": n_t, "samples_per_ray": s_t, "dtype": "f32",
                 "note": "NOT the headline: source repack + forward + full backward of one fused render pass on one GPU "
                         "(no optimizer, no collective)"}

    # dominant kernel (mlp_fwd): HIP events recorded around its two launches per step, on the launch stream
    mlp_ms = sum(a.elapsed_ms(b) for step in events for a, b in step)
    launches = 2 * args.steps
    samples_per_step = args.rays * (args.coarse + args.coarse + args.fine)
    achieved = samples_per_step * args.steps * FLOP_PER_SAMPLE / (mlp_ms * 1e-3) / 1e12
    bf16 = args.precision != "f32"
    psnr_vs_f32 = None
    if args.precision == "bf16":         # outside the parity bar by construction: report the distance to the exact render
        r32 = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), args.coarse, args.fine, precision="f32")
        ref = r32.pass_(out["rays_d"], out["z_fine"])            # same fine depths: isolates the network arithmetic
        mse = torch.mean((out["rgb"] - ref["rgb"]) ** 2).item()
        psnr_vs_f32 = {"psnr_db": -10.0 * __import__("math").log10(max(mse, 1e-20)),
                       "max_abs_rgb": (out["rgb"] - ref["rgb"]).abs().max().item(),
                       "max_abs_depth": (out["depth"] - ref["depth"]).abs().max().item()}
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, scripts/pmc_mlp.sh), which
    # cannot run inside this process: the figure is READ BACK from the committed profile and labelled as such
    traffic, traffic_source = None, None
    for name in (("r02_mlp_bf16_hbm_traffic.json", "r01_mlp_bf16_hbm_traffic.json") if bf16 else ("r02_mlp_fwd_hbm_traffic.json", "r01_mlp_fwd_hbm_traffic.json")):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("bytes_per_launch")
            traffic_source = "profiles/%s (rocprofv3 PMC passes of this command on an earlier box; not measured in this run)" % name
            break

    if rank == 0:
        roof = {"bound": "mfma", "kernel": "mlp_fwd_bf16_kernel" if bf16 else "mlp_fwd_kernel", "achieved": achieved,
                "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": mlp_ms / launches, "flop_per_launch_avg": samples_per_step * FLOP_PER_SAMPLE / 2}
        if bf16:
            # achieved / frac count ALGORITHMIC flops (one multiply-accumulate per weight and sample) against the dense bf16
            # peak; the split evaluation executes three bf16 MFMA products per algorithmic one (fp32-grade accuracy), so the
            # matrix pipe itself runs at `executed` TFLOP/s
            terms = 3 if args.precision == "bf16x3" else 1
            ex = achieved * BF16X3_EXECUTED_FLOP_PER_SAMPLE / 3 * terms / FLOP_PER_SAMPLE
            roof.update(executed=ex, executed_frac=ex / peak,
                        note="bf16x3: 3 bf16 MFMAs per algorithmic MAC; algorithmic ceiling = peak/3.003 = 832 TFLOP/s"
                        if terms == 3 else "plain bf16: NOT within the 1e-4 parity bar, see parity_vs_f32")
        line = {
            "metric": "rendered rays/sec (coarse+fine, 64+128 samples)",
            "value": args.rays * world * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "clock_settle_steps": settle, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 via split-bf16 (bf16x3) MFMA, f32 accumulate",
                      "bf16": "bf16 operands, f32 accumulate"}[args.precision], "data": "synthetic",
            "config": {"workload": "configs[1] shapes: %d rays/GPU x (%d coarse + %d fine -> %d merged) samples, V=7 views "
                                   "256x320, cascade volumes 48x64x80/32x128x160/8x256x320, UCNeRF D=6 W=128 random init"
                                   % (args.rays, args.coarse, args.fine, args.coarse + args.fine),
                       "global_rays": args.rays * world, "parallelism": "ray-sharded x%d, no data-path collective" % world,
                       "precision": args.precision},
            "roofline": roof,
            "mlp_share_of_step": mlp_ms / (dt * 1e3),
            **({"hip_graph_ms_per_step": graph_ms} if graph_ms is not None else {}),
        }
        if psnr_vs_f32 is not None:
            line["parity_vs_f32"] = psnr_vs_f32
        if plain is not None:
            line["plain_bf16"] = plain
        if train is not None:
            line["train_step"] = train
        if dt2 is not None:
            line["reuse_coarse"] = {"value": args.rays * world * args.steps / dt2, "unit": "rays/s", "ms_per_step": dt2 / args.steps * 1e3,
                                    "note": "NOT the headline: fine pass evaluates only the 128 new depths and re-uses the coarse "
                                            "pass's outputs for the 64 coarse ones; renders verified bit-identical in this run"}
        if world == 1 and args.cpu_rays > 0:
            line["cpu_baseline"] = cpu_baseline(scene_cpu, sd, args.cpu_rays, args.coarse, args.fine)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
