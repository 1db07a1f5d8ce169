"""Ray-sharded data parallelism: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

Rays are independent units (no op on the hot path mixes rays), so rendering shards with NO data-path collective:
rank r renders a contiguous block of the ray batch against replicated weights and gather sources.  Training adds
exactly one collective per step: an all-reduce of ONE flat fp32 bucket holding every parameter gradient plus the
loss scalars (0.73 MB for the MLP -- latency-bound over xGMI, so one bucket, not one per tensor).
The reference has no distributed code at all (train.py:433-438 hard-wires one GPU); this module is new design.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialises the default process group from the torchrun environment; returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous block split of n units: the first n % world ranks get one extra.  Returns (start, stop)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_rays(tensors, rank, world, dim=0):
    """Slices every tensor of a list along the ray dimension to this rank's block."""
    n = tensors[0].shape[dim]
    a, b = shard_range(n, rank, world)
    return [t.narrow(dim, a, b - a) for t in tensors]


def patch_aligned_ranges(n_rays, n_patch_rays, patch_elems, world):
    """Training batches of the reference start with whole patches (patch_num x patch_size^2 rays whose depth maps
    feed patch-wise smoothness losses, train.py:172-185).  Shards must not cut a patch: patches are dealt out in
    whole units, the remaining rays are block-split.  Returns per-rank lists of (start, stop) ranges."""
    if n_patch_rays % patch_elems or n_patch_rays > n_rays:
        raise ValueError("patch region of %d rays is not a whole number of %d-ray patches" % (n_patch_rays, patch_elems))
    n_patches = n_patch_rays // patch_elems
    out = []
    for r in range(world):
        pa, pb = shard_range(n_patches, r, world)
        ra, rb = shard_range(n_rays - n_patch_rays, r, world)
        ranges = []
        if pb > pa:
            ranges.append((pa * patch_elems, pb * patch_elems))
        if rb > ra:
            ranges.append((n_patch_rays + ra, n_patch_rays + rb))
        out.append(ranges)
    return out


class FlatGradBucket:
    """All-reduces every parameter gradient (+ optional scalars) as ONE flat fp32 tensor.

    Shards of unequal size are handled by weighting: each rank's gradient is that of the MEAN loss over its own
    rays, so the global mean-loss gradient is sum_r (n_r / n) g_r -- pass `weight = n_r / n` (1/world for equal
    shards, which reproduces plain gradient averaging).

    Which parameters receive a gradient is a property of the model and the loss, not of the step (the reference leaves
    six tensors without one) -- except on a rank whose shard is empty, where NO parameter gets one.  Every
    rank must still end up with the same gradients, so the bucket also carries one has-gradient flag per parameter; the
    reduced flags are read back ONCE (the only host synchronisation this class ever makes) and cached: parameters some
    rank differentiated get their p.grad materialised from the reduced segment on every rank, parameters no rank touched
    keep grad = None as in the reference.  Nothing else reads device memory: scalars are written into the bucket with a
    device-side copy.

    The cached flags are refreshed (one more read-back) whenever they could have changed: when THIS rank's own pattern of
    None / not-None gradients differs from the previous step's (a loss term switched on by a schedule, layers unfrozen -- every
    rank runs the same model and loss, so every rank with rays sees the change and refreshes in the same step), and in every
    step on a rank that differentiated nothing at all (an empty shard: it cannot see such a change locally, and its step time
    does not matter)."""

    def __init__(self, params, n_scalars=0):
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.n_scalars = n_scalars
        self.n_grads = sum(self.sizes)
        self.numel = self.n_grads + n_scalars + len(self.params)
        self.flat = None
        self.has_grad = None         # cached reduced flags: list of bool, one per parameter
        self.local_flags = None      # this rank's own None / not-None pattern at the step has_grad was read

    def allreduce(self, weight, scalars=(), group=None):
        """In place on p.grad.  Returns the reduced scalars (e.g. global loss terms) as a device tensor."""
        if len(scalars) != self.n_scalars:
            raise ValueError("expected %d scalars, got %d" % (self.n_scalars, len(scalars)))
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.params, self.sizes):
            seg = self.flat[off:off + n]
            if p.grad is None:
                seg.zero_()
            else:
                torch.mul(p.grad.reshape(-1), weight, out=seg)
            off += n
        if self.n_scalars:
            vals = torch.stack([torch.as_tensor(s, dtype=torch.float32, device=dev).reshape(()) for s in scalars])
            torch.mul(vals, weight, out=self.flat[off:off + self.n_scalars])
        off += self.n_scalars
        local = tuple(p.grad is not None for p in self.params)                     # (host-side: no device access)
        refresh = self.has_grad is None or local != self.local_flags or not any(local)
        # the flags always travel (every rank must contribute the same bucket layout); they are only READ when a refresh is due
        self.flat[off:].copy_(torch.tensor([1.0 if f else 0.0 for f in local]), non_blocking=True)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if refresh:
            self.has_grad = [bool(v > 0) for v in self.flat[off:].tolist()]       # the read-back: first step, pattern change, empty shard
            self.local_flags = local
        off = 0
        for p, n, has in zip(self.params, self.sizes, self.has_grad):
            if p.grad is not None:
                p.grad.copy_(self.flat[off:off + n].view_as(p.grad))
            elif has:                                                              # e.g. an empty shard on this rank
                p.grad = self.flat[off:off + n].view_as(p).clone()
            off += n
        return self.flat[self.n_grads:self.n_grads + self.n_scalars].clone()


def all_gather_rays(local, n_total, rank, world, group=None):
    """Gathers per-rank ray outputs [n_r, ...] (block split of n_total) back into [n_total, ...] on every rank."""
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    longest = max(b - a for a, b in sizes)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][:b - a] for r, (a, b) in enumerate(sizes)], 0)


class ShardedRenderer:
    """Renders this rank's block of a pixel list with `render_fn(xs, ys, **kw) -> dict of [n_r, ...] tensors`
    (e.g. CoarseFineRenderer.render) and, if asked, reassembles the full image on every rank."""

    def __init__(self, render_fn, rank=None, world=None, group=None):
        self.render_fn = render_fn
        self.rank = dist.get_rank(group) if rank is None and dist.is_initialized() else (rank or 0)
        self.world = dist.get_world_size(group) if world is None and dist.is_initialized() else (world or 1)
        self.group = group

    def render(self, xs, ys, gather=("rgb", "depth"), **kw):
        n = xs.shape[0]
        a, b = shard_range(n, self.rank, self.world)
        out = self.render_fn(xs[a:b], ys[a:b], **kw)
        if gather:
            for k in gather:
                out[k] = all_gather_rays(out[k], n, self.rank, self.world, self.group)
        return out
