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

    **No pack, no unpack for the renderer network.**  Its parameters live in one flat buffer and the backward of `rendering()` writes one
    flat gradient vector that the `p.grad`s are views of (flat.FlatStore); the bucket IS that vector -- layout
    `[network gradients (in place) | other parameters' gradients | scalars | has-gradient flags]`, the last three in the room the backward
    left behind the gradients -- so a step costs: one multi-tensor copy for parameters of other modules (none when only the renderer
    trains), one copy per scalar, one for the flags, one scale, ONE all-reduce, and nothing afterwards.  Gradients that are not views of one
    buffer (the op-by-op route, accumulation over several backward passes) take the generic path: the same bucket filled and emptied
    with two multi-tensor copies.

    Which parameters receive a gradient is a property of the model and the loss, not of the step (the reference leaves
    six tensors without one) -- except on a rank whose shard is empty, where NO parameter gets one.  Every
    rank must still end up with the same gradients, so the bucket also carries one has-gradient flag per parameter; the
    reduced flags are read back (the only host synchronisation this class makes) and cached: parameters some
    rank differentiated get their p.grad materialised from the reduced segment on every rank, parameters no rank touched
    keep grad = None as in the reference.  Nothing else reads device memory: scalars are written into the bucket with a
    device-side copy.

    **When the reduced flags are read** (`verify_every`).  A pattern change that only ANOTHER rank can see (a data-dependent branch, a per-rank
    loss term) is invisible locally, and an optimizer step taken on one rank and skipped on another is permanent divergence -- so the default,
    `verify_every=1`, reads the reduced flags back in EVERY step (144 bytes behind the all-reduce; the one host synchronisation of the step):
    every rank materialises exactly the gradients some rank produced, in the same step, always.  `verify_every=N > 1` reads them every N steps,
    when this rank's own None / not-None pattern changed, and in every step on a rank that differentiated nothing; in between, the reduced flags
    are compared with the cached pattern ON THE DEVICE and the mismatch count travels to pinned host memory behind an event that the NEXT call
    tests without waiting: a change is then acted on one step late (`late_detections` counts them, a warning is issued) -- for callers that
    accept that in exchange for a step without a read-back.  `verify_every=0`: never after the first step (a captured HIP graph admits no
    read-back: train_step.GraphedStep); exact for optimizers that step the flat buffer as ONE parameter (flat.FlatAdam: every rank applies the
    whole reduced vector, whatever the flags say)."""

    def __init__(self, params, n_scalars=0, verify_every=1, collective_at_one=False):
        from .flat import store_of_param
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.n_scalars = n_scalars
        self.verify_every = int(verify_every)
        self.collective_at_one = bool(collective_at_one)      # issue the all-reduce through a ONE-rank group too (rehearsal of the RCCL call on a one-GPU box)
        # parameters that live in a flat store (the renderer network) keep their place in the store's gradient vector; the others are
        # packed behind it
        self.store, self.seg = None, []                # seg[i] = offset of parameter i inside the bucket
        placed = [store_of_param(p) for p in self.params]
        stores = {id(st): st for st, _ in placed if st is not None}
        if len(stores) == 1:
            self.store = next(iter(stores.values()))
        base = self.store.n if self.store is not None else 0
        off = base
        for p, n, (st, i) in zip(self.params, self.sizes, placed):
            if st is not None and st is self.store:
                self.seg.append(st.offsets[i])
            else:
                self.seg.append(off)
                off += n
        self.n_grads = off                             # floats in front of the scalars
        self.numel = self.n_grads + n_scalars + len(self.params)
        self.in_store = [st is not None and st is self.store for st, _ in placed]
        if self.store is not None:
            from .flat import TAIL_FLOATS
            self.store.reserve(max(0, self.numel - self.store.n - TAIL_FLOATS))
        self.flat = None             # the generic path's own buffer
        self.has_grad = None         # cached reduced flags: list of bool, one per parameter
        self.local_flags = None      # this rank's own None / not-None pattern at the step has_grad was read
        self._flag_cache = {}        # local pattern -> device tensor of 0 / 1 flags
        self._step = 0
        self.last_path = None        # "in_place" / "generic": which route the last allreduce took (tests, bench)
        self.readbacks = 0           # how many steps read the reduced flags back (tests, bench)
        self.late_detections = 0     # verify_every > 1: pattern changes of another rank found one step late by the device-side check
        self._pending = None         # (pinned mismatch count, event or None) of the last step's device-side check
        self._cached_dev = None      # the cached reduced pattern as a device tensor

    def _flags_tensor(self, local, dev):
        t = self._flag_cache.get((local, dev))
        if t is None:
            t = self._flag_cache[(local, dev)] = torch.tensor([1.0 if f else 0.0 for f in local], dtype=torch.float32, device=dev)
        return t

    def allreduce(self, weight, scalars=(), group=None):
        """In place on p.grad.  Returns the reduced scalars (e.g. global loss terms) as a device tensor."""
        if len(scalars) != self.n_scalars:
            raise ValueError("expected %d scalars, got %d" % (self.n_scalars, len(scalars)))
        dev = self.params[0].device
        local = tuple(p.grad is not None for p in self.params)                     # (host-side: no device access)
        # ---- the bucket: the store's own gradient vector when this step's gradients are views of one, else our buffer
        buf = self.store.flat_grad(room=self.numel) if self.store is not None else None
        in_place = buf is not None
        if not in_place:
            if self.flat is None or self.flat.device != dev:
                self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
            buf = self.flat
            buf[:self.n_grads].zero_()
        self.last_path = "in_place" if in_place else "generic"
        views = [buf[o:o + n] for o, n in zip(self.seg, self.sizes)]
        fill = [(v, p.grad.reshape(-1)) for v, p, ins in zip(views, self.params, self.in_store) if p.grad is not None and not (in_place and ins)]
        if fill:
            torch._foreach_copy_([a for a, _ in fill], [b for _, b in fill])
        off = self.n_grads
        for k, sc in enumerate(scalars):
            buf[off + k].copy_(torch.as_tensor(sc, dtype=torch.float32).reshape(()), non_blocking=True)
        off += self.n_scalars
        # the flags always travel (every rank must contribute the same bucket layout); they are only READ when a refresh is due
        buf[off:off + len(self.params)].copy_(self._flags_tensor(local, dev), non_blocking=True)
        if weight != 1.0:
            buf[:self.numel].mul_(weight)
        if dist.is_initialized() and (dist.get_world_size(group) > 1 or self.collective_at_one):
            dist.all_reduce(buf[:self.numel], op=dist.ReduceOp.SUM, group=group)
        self._step += 1
        flags = buf[off:off + len(self.params)]
        if self._pending is not None:                      # last step's device-side comparison, if it has arrived (never waited for)
            pin, ev = self._pending
            if ev is None or ev.query():
                self._pending = None
                if float(pin) != 0.0:
                    self.late_detections += 1
                    self.has_grad = None                   # another rank's pattern changed last step: refresh now
                    import warnings
                    warnings.warn("uc_nerf_amd.FlatGradBucket: another rank's gradient pattern changed and was noticed one step late "
                                  "(verify_every=%d); use verify_every=1 for same-step agreement" % self.verify_every)
        refresh = (self.has_grad is None or local != self.local_flags or not any(local)
                   or (self.verify_every > 0 and self._step % self.verify_every == 0))
        if refresh:
            self.has_grad = [bool(v > 0) for v in flags.tolist()]                  # the read-back
            self.local_flags = local
            self.readbacks += 1
            self._cached_dev = None
        elif self.verify_every > 1 and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            # no read-back this step: compare the reduced flags with the cached pattern on the device; the count reaches the host behind an event
            if self._cached_dev is None or self._cached_dev.device != dev:
                self._cached_dev = torch.tensor([1.0 if h else 0.0 for h in self.has_grad], dtype=torch.float32, device=dev)
            mis = ((flags > 0).to(torch.float32) - self._cached_dev).abs().sum()
            if dev.type == "cuda":
                pin = torch.empty((), dtype=torch.float32, pin_memory=True)
                pin.copy_(mis, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (pin, ev)
            else:
                self._pending = (mis, None)
        back = []
        for p, v, has, ins in zip(self.params, views, self.has_grad, self.in_store):
            if p.grad is not None:
                if not (in_place and ins):
                    back.append((p.grad, v.view_as(p.grad)))
            elif has:                                                              # e.g. an empty shard on this rank
                p.grad = v.view_as(p).clone()
        if back:
            torch._foreach_copy_([a for a, _ in back], [b for _, b in back])
        return buf[self.n_grads:self.n_grads + self.n_scalars].clone()


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
