"""Flat parameter and gradient storage for the renderer network ("one flat bucket").

The reference hands `grad_vars` -- 36 separate tensors -- to Adam and steps once per batch (train.py:85-92,147-188); its
autograd produces 30 separate gradient tensors.  The kernels here read ONE flat parameter vector (the concatenation of the
state_dict tensors in state_dict order) and write ONE flat gradient vector, so the storage that makes both free is:

  * every parameter of the module is a VIEW of one flat fp32 buffer (`FlatStore.flat`).  Names, shapes, `state_dict()` and
    `load_state_dict()` are unchanged (load copies in place); `module.to(device)` re-flattens on the new device
    (`UCNeRF._apply`), and `sync()` re-flattens whenever some `p.data` was re-pointed behind our back;
  * the backward of `rendering()` writes ONE flat gradient vector (the head of the step's zero-filled gradient pool, a fresh allocation
    per backward) and returns views of it, which autograd installs as `p.grad` without a copy when `p.grad` was None (the default
    `zero_grad(set_to_none=True)`); an existing `p.grad` is accumulated into as usual.  A step's gradients are therefore views of one
    buffer that nobody overwrites later -- no aliasing between steps;
  * `parallel.FlatGradBucket` all-reduces that very buffer, with the loss scalars and has-gradient flags in its tail: no pack,
    no unpack;
  * `FlatAdam` steps the flat buffer with the flat gradient as ONE fused launch.
The six tensors the reference's autograd never reaches keep `grad = None`: their segments of the gradient
buffer are simply never handed out.
"""
import weakref

import torch

TAIL_FLOATS = 256          # spare floats behind the gradients of a buffer: a bucket's loss scalars and has-gradient flags travel there


class FlatStore:
    """Flat storage of `module`'s parameters (registration = state_dict order).  `FlatStore.of(module)` returns the module's store,
    creating it on first use."""

    def __init__(self, module):
        self.params = list(module.parameters())
        if not self.params:
            raise ValueError("uc_nerf_amd.FlatStore: module has no parameters")
        self.sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        self.n = self.offsets[-1]
        self.extra = 0             # floats a bucket reserved behind the gradients (other modules' gradients), see reserve()
        self.flat = None
        me = weakref.ref(self)
        for p in self.params:
            p._ucnerf_flat_owner = me
        self.flatten()

    @classmethod
    def of(cls, module):
        params = list(module.parameters())
        same = lambda s_: s_ is not None and len(s_.params) == len(params) and all(a is b for a, b in zip(s_.params, params))      # noqa: E731
        st = module.__dict__.get("_ucnerf_flat_store")
        if not same(st):
            # (a wrapper and the module it wraps -- UCNeRF and its .nerf -- hold the same parameters: ONE store between them)
            owner = getattr(params[0], "_ucnerf_flat_owner", None) if params else None
            st = owner() if owner is not None else None
            if not same(st):
                st = cls(module)
            module.__dict__["_ucnerf_flat_store"] = st
        return st

    # ---- parameters
    def is_flat(self):
        f = self.flat
        if f is None:
            return False
        base, dev, dt = f.data_ptr(), f.device, f.dtype
        es = f.element_size()
        return all(p.device == dev and p.dtype == dt and p.data_ptr() == base + es * o and p.is_contiguous()
                   for p, o in zip(self.params, self.offsets))

    def flatten(self):
        """(Re)builds the flat buffer from the parameters as they are now and re-points every `p.data` at its segment."""
        p0 = self.params[0]
        if any(p.device != p0.device or p.dtype != p0.dtype for p in self.params):
            raise RuntimeError("uc_nerf_amd.FlatStore: parameters must share one device and dtype")
        flat = torch.empty(self.n, dtype=p0.dtype, device=p0.device)
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                seg = flat[o:o + n].view(p.shape)
                seg.copy_(p.data)
                p.data = seg
        self.flat = flat
        self._ptrs = [p.data_ptr() for p in self.params]      # (sync()'s fast check)
        # which flat buffer this is, counted up and never reused: an ADDRESS is no identity across re-flattenings (the old buffer is freed and the
        # allocator may hand the same address out again -- with the same version count, which starts from the 36 copies above every time)
        self.generation = getattr(self, "generation", 0) + 1
        return flat

    def sync(self):
        """The flat parameter buffer, guaranteed to BE the parameters' storage (re-flattens after `.to()`, `.float()`,
        `load_state_dict(assign=True)` or any other re-pointing of `p.data`).  The common case -- nothing moved -- is one pass over the 36
        data pointers against the list recorded when the buffer was built (a per-chunk evaluation call pays it 80 times per image)."""
        f = self.flat
        if not (f is not None and getattr(self, "_ptrs", None) is not None and [p.data_ptr() for p in self.params] == self._ptrs
                and self.params[0].dtype == f.dtype and self.params[0].device == f.device) and not self.is_flat():
            self.flatten()
        return self.flat

    # ---- gradients
    def reserve(self, extra):
        """Room for `extra` more floats between the gradients and the tail of every gradient buffer handed out from now on (a bucket that
        also carries other modules' gradients)."""
        if extra > self.extra:
            self.extra = int(extra)

    @property
    def grad_room(self):
        """Floats a backward allocates for the flat gradient: the gradients, a bucket's reserved room, the tail."""
        return self.n + self.extra + TAIL_FLOATS

    def grad_views(self, buf, wanted):
        """Fresh views of the flat gradient `buf`, one per parameter (None where `wanted` is False): what a backward returns to autograd.
        (Fresh on purpose: autograd installs a returned gradient as `p.grad` WITHOUT a copy only when nobody else holds the tensor.)"""
        self._issued = (buf.untyped_storage().data_ptr(), buf.storage_offset(), buf.numel())      # what flat_grad() may hand to a bucket
        # (one split for all segments, a view only where a parameter is not a vector: 19 tensor operations instead of 72 -- the 250-ray data-parallel
        #  step is bound by the host's issue time)
        segs = buf[:self.n].split(self.sizes)
        return [(g if p.dim() == 1 else g.view(p.shape)) if w else None for g, p, w in zip(segs, self.params, wanted)]

    def zero_segments(self, buf, which):
        """Zeroes the segments of the flat gradient `buf` selected by `which` (one multi-tensor launch): the backward kernels write a weight
        gradient for EVERY tensor they reach, and a tensor the caller froze (requires_grad = False) must not be stepped by an optimizer
        (FlatAdam) or reduced by a bucket that takes the buffer whole -- round 4's advisor finding."""
        segs = [g for g, w in zip(buf[:self.n].split(self.sizes), which) if w]
        if segs:
            torch._foreach_zero_(segs)

    def flat_grad(self, room=None):
        """The flat fp32 tensor [>= n + extra + tail] every non-None `p.grad` of this store is a view of (each at its own offset), or None:
        gradients produced some other way (the op-by-op route, accumulation into existing grads, two passes summed by autograd)."""
        first = next(((p.grad, o) for p, o in zip(self.params, self.offsets) if p.grad is not None), None)
        if first is None:
            return None
        g0, o0 = first
        if g0.dtype != torch.float32:
            return None
        stg = g0.untyped_storage()
        base = g0.storage_offset() - o0                    # element offset of the flat gradient inside the storage
        room = self.grad_room if room is None else room
        sp = stg.data_ptr()
        # only the vector the LAST backward handed out, and only as much room as it left behind the gradients (what follows in the storage
        # belongs to somebody else: the source gradients of the same pool)
        if getattr(self, "_issued", None) is None or (sp, base) != self._issued[:2] or room > self._issued[2]:
            return None
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != sp or g.storage_offset() != base + o:
                return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(stg, base, (room,))


def store_of_param(p):
    """The FlatStore whose buffer `p` lives in (and p's index there), or (None, -1)."""
    owner = getattr(p, "_ucnerf_flat_owner", None)
    st = owner() if owner is not None else None
    if st is None:
        return None, -1
    for i, q in enumerate(st.params):
        if q is p:
            return st, i
    return None, -1


class FlatAdam(torch.optim.Adam):
    """Adam over a FlatStore-backed module as ONE parameter: the flat buffer, with the flat gradient the backward wrote -- one fused
    launch per step instead of a multi-tensor pass over 30 tensors.  Element-wise identical to `torch.optim.Adam(module.parameters())`
    (Adam is element-wise; the six tensors without gradients see g = 0, m = v = 0: no update) as long as `weight_decay == 0` -- refused
    otherwise -- and every differentiated tensor gets its first gradient in the same step (one step counter for the whole buffer).
    `zero_grad()` clears the module's `p.grad`s."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        if weight_decay != 0:
            raise ValueError("uc_nerf_amd.FlatAdam: weight decay would also decay the tensors the reference leaves without a gradient")
        self.store = FlatStore.of(module)
        self.fp = torch.nn.Parameter(self.store.sync())
        kw.setdefault("fused", self.fp.is_cuda)               # (capturable=True: usable inside a HIP-graph capture, train_step.GraphedStep)
        super().__init__([self.fp], lr=lr, betas=betas, eps=eps, weight_decay=0, **kw)

    def _bind(self):
        st = self.store
        flat = st.sync()
        if self.fp.data_ptr() != flat.data_ptr():
            self.fp.data = flat
        g = st.flat_grad(room=st.n)
        if g is None:
            # gradients that are not views of one buffer (another backward route): gather them
            g = torch.zeros(st.n, dtype=torch.float32, device=flat.device)
            have = [(g[o:o + n].view(p.shape), p.grad) for p, o, n in zip(st.params, st.offsets, st.sizes) if p.grad is not None]
            if have:
                torch._foreach_copy_([a for a, _ in have], [b for _, b in have])
        self.fp.grad = g[:st.n]

    @torch.no_grad()
    def step(self, closure=None):
        if all(p.grad is None for p in self.store.params):
            return None                            # (nothing was differentiated: torch's Adam skips parameters without a gradient too)
        self._bind()
        out = super().step(closure)
        # torch's FUSED Adam kernel writes the parameters without bumping their version counter (verified on torch 2.10: `_version` stays put), so
        # nothing keyed on versions -- dropin.set_weight_cache("versions") -- could see this step: bump the flat buffer's counter by hand (no launch)
        torch.autograd.graph.increment_version(self.store.flat)
        return out

    def zero_grad(self, set_to_none=True):
        for p in self.store.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        self.fp.grad = None
