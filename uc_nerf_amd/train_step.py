"""The reference's optimisation step (train.py:110-211) on a prepared ray batch, under ray-sharded data parallelism
(SURVEY.md section 8, row f3).

A training batch is laid out as the reference builds it (utils/utils.py:400-597, train.py:164-188):

    [ patch rays: patch_num x patch_size^2 | other rays up to N_rays | sparse-depth rays from N_rays on ]

and its loss mixes four kinds of means: over all rays (image term), over the sparse-depth rays, over the first half of
the patches (edge-preserving smoothness against the DPT prior) and over the second half (scale-and-shift invariant
gradient term), plus the cascade depth loss on the (replicated) MVS outputs.  Sharding therefore deals out each region
separately -- whole patches, never cut -- and every rank scales each term by its share of that term's population, so
that the SUM of the ranks' gradients is exactly the gradient of the single-process loss: one all-reduce of one flat
bucket per step (`parallel.FlatGradBucket`), no other collective.  The reference has no distributed code (train.py:433-438
hard-wires one GPU); with world = 1 this is the reference's step.
"""
import numpy as np
import torch

from . import parallel as P
from .utils import loss as L


class BatchShard:
    """This rank's part of a training batch: `index` (into the global ray batch, ordered [A patches | B patches | other
    rays | sparse-depth rays]) and the local / global population of every loss term."""

    def __init__(self, n_rays, n_total, patch_num, patch_size, rank=0, world=1):
        pe = patch_size * patch_size
        n_a = patch_num // 2
        n_b = patch_num - n_a
        if patch_num * pe > n_rays or n_rays > n_total:
            raise ValueError("batch layout: %d patch rays, N_rays %d, %d rays in total" % (patch_num * pe, n_rays, n_total))
        a0, a1 = P.shard_range(n_a, rank, world)
        b0, b1 = P.shard_range(n_b, rank, world)
        o0, o1 = P.shard_range(n_rays - patch_num * pe, rank, world)
        d0, d1 = P.shard_range(n_total - n_rays, rank, world)
        self.patch_size = patch_size
        self.local = {"A": a1 - a0, "B": b1 - b0, "other": o1 - o0, "depth": d1 - d0}
        self.total = {"A": n_a, "B": n_b, "rays": n_total, "depth": n_total - n_rays}
        self.patch_ids = torch.cat([torch.arange(a0, a1), n_a + torch.arange(b0, b1)])         # rows of the patch prior
        self.index = torch.cat([torch.arange(a0 * pe, a1 * pe), n_a * pe + torch.arange(b0 * pe, b1 * pe),
                                patch_num * pe + torch.arange(o0, o1), n_rays + torch.arange(d0, d1)])
        self.depth_ids = torch.arange(d0, d1)                                                   # rows of target depths / weights
        self.world = world

    @property
    def n_local(self):
        return int(self.index.numel())


def sharded_training_loss(rgb, depth_pred, target_s, target_depths, target_weights, patch_dpt, mvs_outputs, depth_sparse_ms, weight_ms,
                          shard, smooth_loss=None, edge_loss=None, mvs_term=None):
    """This rank's share of the loss of train.py:164-188; all arguments are the rank's LOCAL rows in `shard.index` order
    (`patch_dpt`: rows `shard.patch_ids`, `target_depths/weights`: rows `shard.depth_ids`).  Summed over the ranks it is the
    single-process loss, and so are its gradients.  Returns (weighted local loss, dict of weighted local terms)."""
    smooth_loss = smooth_loss or L.EdgePreservingSmoothnessLoss()
    edge_loss = edge_loss or L.GradientLoss()
    ps, loc, tot = shard.patch_size, shard.local, shard.total
    pe = ps * ps
    na, nb = loc["A"], loc["B"]
    zero = rgb.sum() * 0.0
    patch_depth = depth_pred[:(na + nb) * pe].reshape(-1, ps, ps)
    smooth = smooth_loss(patch_depth[:na], patch_dpt[:na]) * (na / tot["A"]) if na else zero
    scale_inv = edge_loss(patch_depth[na:], patch_dpt[na:].squeeze(-1), torch.ones_like(patch_depth[na:])) * (nb / tot["B"]) if nb else zero
    nd = loc["depth"]
    n_loc = rgb.shape[0]
    nerf_depth = (torch.mean(((depth_pred[n_loc - nd:] - target_depths) ** 2) * target_weights) * (nd / tot["depth"])) if nd else zero
    img = L.img2mse(rgb, target_s) * (n_loc / tot["rays"]) if n_loc else zero
    # (mvs_term: the cascade depth loss already evaluated -- its boolean-mask indexing reads sizes back to the host, which a graph capture does
    #  not admit; it depends on nothing the renderer produces)
    mvs = mvs_term if mvs_term is not None else L.cas_mvsnet_loss(mvs_outputs, depth_sparse_ms, weight_ms)[0]
    mvs = mvs / shard.world                                                  # replicated term: every rank holds all of it
    loss = nerf_depth * 0.05 + mvs * 0.05 + smooth * 0.05 + scale_inv * 0.008 + img * 5.0
    return loss, {"img_loss": img, "loss_nerf_depth": nerf_depth, "loss_mvs": mvs, "smooth_loss": smooth,
                  "loss_scaleinvariant": scale_inv}


class TrainStep:
    """`render_fn(index) -> (rgb [n,3], depth_pred [n])` renders the given rays of the global batch (on the GPU: the
    `network.renderer.rendering` mirror on the rank's rows); `params` are the tensors the optimizer updates.

    (Round 5 built an automatic per-batch-shape HIP-graph capture into this class -- eager for a few steps, then one replay per step on static
    copies of the batch -- and took it out again: on ROCm 7.2 `hipStreamEndCapture` crashed inside `hip::Stream::EndCapture` for this step's
    capture, in a fresh process as in the test suite, whatever was stripped from the captured function; `GraphedStep` below, the explicit
    opt-in on a caller-built step function, is unaffected.  The experiment, the rocgdb backtrace and the variants tried are in
    profiles/r05_experiments.md.)"""

    TERMS = ("img_loss", "loss_nerf_depth", "loss_mvs", "smooth_loss", "loss_scaleinvariant")

    def __init__(self, render_fn, params, optimizer, n_rays, patch_num, patch_size, rank=0, world=1, group=None):
        self.render_fn, self.optimizer = render_fn, optimizer
        self.n_rays, self.patch_num, self.patch_size = n_rays, patch_num, patch_size
        self.rank, self.world, self.group = rank, world, group
        self.bucket = P.FlatGradBucket(list(params), n_scalars=1 + len(self.TERMS))

    def __call__(self, target_s, target_depths, target_weights, patch_dpt, mvs_outputs, depth_sparse_ms, weight_ms):
        """All arguments are the GLOBAL batch tensors (every rank holds them: they are tiny); returns the global loss terms."""
        n_total = target_s.shape[0]
        sh = BatchShard(self.n_rays, n_total, self.patch_num, self.patch_size, self.rank, self.world)
        idx = sh.index.to(target_s.device)
        rgb, depth_pred = self.render_fn(idx)
        loss, parts = sharded_training_loss(rgb, depth_pred, target_s[idx], target_depths[sh.depth_ids.to(target_depths.device)],
                                            target_weights[sh.depth_ids.to(target_weights.device)],
                                            patch_dpt[sh.patch_ids.to(patch_dpt.device)], mvs_outputs, depth_sparse_ms, weight_ms, sh)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        # the terms are already weighted by population shares: the plain sum over ranks is the global gradient / loss
        red = self.bucket.allreduce(1.0, [loss.detach()] + [parts[k].detach() for k in self.TERMS], group=self.group)
        self.optimizer.step()
        out = {"loss": red[0]}
        out.update({k: red[1 + i] for i, k in enumerate(self.TERMS)})
        out["img_mse"] = out["img_loss"]
        out["psnr"] = -10.0 * np.log(max(float(out["img_loss"]), 1e-20)) / np.log(10.0)        # mse2psnr2 (utils/utils.py:14)
        return out


_CAPTURE_STREAMS = []        # streams captures have run on: kept for the life of the process (see GraphedStep.__init__)


class GraphedStep:
    """One optimisation step as ONE HIP-graph replay (SURVEY.md 8(f) f1 "HIP-graph capture", for the regime it pays in: a data-parallel shard of a few
    hundred rays, where the step is bound by the host -- the rendering() training step on 250 rays x 90 issues ~70 launches in ~0.8-1.0 ms of Python
    against ~0.5 ms of GPU work; profiles/r04_experiments.md).

    `fn()` is the whole step on STATIC tensors -- `optimizer.zero_grad(set_to_none=True)`, `rendering()`, the loss, `backward()`, optionally
    `FlatGradBucket.allreduce(...)` (RCCL collectives capture), `optimizer.step()` -- and returns a tensor or tuple of tensors (e.g. the loss).
    Every library launch goes to torch's current stream and every allocation through torch's caching allocator, so the capture is a plain stream
    capture.  What the caller owes: the optimizer must be capturable (`torch.optim.Adam(..., capturable=True)` or `flat.FlatAdam(...,
    capturable=True)`); a new batch is COPIED INTO the tensors `fn` reads (same shapes) before `replay()`; nothing inside `fn` may read device
    memory back (`.item()`, `float()`; a FlatGradBucket must be built with `verify_every=0`)."""

    def __init__(self, fn, warmup=3, warmup_fn=None):
        import torch
        dev = torch.cuda.current_device()
        # ONE stream of this graph's own for the warm-up AND the capture, never destroyed (round 5).  Same stream: the parameters' gradient
        # accumulators are created by the warm-up's forward on the stream the capture runs on, so autograd need not bridge two streams with
        # short-lived events inside the capture (torch warned about exactly that mismatch in round 4's captures).  Never destroyed, and not
        # torch's shared default capture stream: the HIP runtime keeps raw pointers to the streams and events a capture touched
        # (hip::Stream::EndCapture walks them; a crash there is on record, profiles/r05_experiments.md) -- nothing it may point at is freed.
        cap = torch.cuda.Stream(device=dev)
        _CAPTURE_STREAMS.append(cap)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):                         # warm-up: one-time attribute calls, workspaces, optimizer state
            for _ in range(max(0, warmup)):                  # (warmup_fn: a pass that warms the same ops up without stepping the optimizer)
                (warmup_fn or fn)()
        torch.cuda.current_stream(dev).wait_stream(cap)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=cap):
            self.out = fn()

    def replay(self):
        self.graph.replay()
        return self.out

    __call__ = replay
