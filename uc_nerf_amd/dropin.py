"""Host-side engine behind the reference-named `rendering()` (network/renderer.py:215-255): routes the call surface
train.py uses onto the kernels the benchmark is quoted on.

What one `rendering()` call becomes on the device (training, grad enabled):

    dir_feature -> [channel-last repack of the sources, only if they changed] -> channel-last gather on the GIVEN
    coordinates (rays_pts / rays_ndc as build_rays produced them) -> PE + MLP (training forward: activations kept) ->
    composite;   backward: composite_bwd -> layer-by-layer MLP backward from the kept activations -> gather backward.

and under `torch.no_grad()` (evaluation, train.py:254-272) the lean pass: tiled features, nothing kept, optionally the
split-bf16 matrix-core MLP (`set_inference_precision("bf16x3")`, or an `inference_precision` attribute on `args`).

State that survives between calls lives in one `FusedSession` per network module:
  * the packed weight stream per precision, rebuilt only when a parameter's version counter (or storage) changed, i.e.
    once per optimizer step, not once per call;
  * the `GatherSources` + their channel-last copies, rebuilt only when a source tensor object / version changed, i.e. once
    per image in evaluation (80 chunks share them) and once per step in training;
  * the render / backward workspaces.
"""
import weakref

import torch

from . import ops
from .flat import FlatStore

_INFERENCE_PRECISION = "bf16x3_fused"     # what an unmodified train.py gets under torch.no_grad(): the kernel the headline benchmark is quoted on
_TRAINING_PRECISION = "f32"         # arithmetic of the training FORWARD (the kept activations have the format the backward mode reads either way)
_INFERENCE_PRECISIONS = ("f32", "bf16x3", "bf16", "bf16x3_fused")
_WEIGHT_CACHE = "verify"            # how an inference call gets its packed weight stream, see set_weight_cache
_SOURCE_PRECISION = "f32"           # element type of the channel-last source copies the gather reads, see set_source_precision
_MAX_FEATURE_BYTES = (1 << 31) - (1 << 20)       # the MLP kernels address a pass's feature buffer with 32-bit byte offsets


def set_inference_precision(precision):
    """MLP arithmetic of `rendering()` under torch.no_grad() (evaluation, train.py:254-272): "bf16x3_fused" (default since round 4:
    split-bf16 matrix cores -- every product as hi*hi + hi*lo + lo*hi, fp32 accumulate, rendered outputs within 2e-5 of the exact ones,
    the parity bar is 1e-4 -- with the feature gather INSIDE the MLP kernel: the kernel the headline benchmark is quoted on, reading the
    coordinates rendering() was handed; per-sample uncertainty extras and bf16 source copies fall back to "bf16x3"), "bf16x3" (the same
    arithmetic as gather -> feature buffer -> MLP), "f32" (exact fp32 MFMA: the opt-out, 2.7x slower) or "bf16" (plain bf16 operands, NOT
    within the bar).  Training runs f32 unless set_training_precision says otherwise."""
    global _INFERENCE_PRECISION
    if precision not in _INFERENCE_PRECISIONS:
        raise ValueError("uc_nerf_amd: inference precision must be one of %s, got %r" % (", ".join(_INFERENCE_PRECISIONS), precision))
    _INFERENCE_PRECISION = precision


def set_weight_cache(policy):
    """How a no_grad `rendering()` call obtains the packed weight stream of its network:
      "verify"   (default) re-packed from the LIVE parameter tensors in every call -- one small launch (ucnerf_mlp_pack_tensors, ~3 us
                 of GPU time).  Nothing can go stale: in-place writes through `.data` (the reference's own weights_init, network/models.py:15-17;
                 EMA / clamping code; older optimizers) do not bump a tensor's version counter, so no cache key can see them.
      "versions" the stream is cached and rebuilt only when a parameter's version counter or storage changed (optimizer steps,
                 load_state_dict, ordinary in-place ops).  One launch less per call; for evaluation loops that do not write parameters
                 through `.data` -- or that call `uc_nerf_amd.dropin.session_of(net).invalidate()` after doing so.  (torch's FUSED optimizer
                 kernels -- `torch.optim.Adam(..., fused=True)` -- do not bump version counters either; `flat.FlatAdam` bumps the flat
                 buffer's by hand after its step, the default foreach Adam of train.py:85-92 bumps them itself.)
    Calls under autograd always pack afresh (once per training step; the backward needs the flat vector of that moment anyway)."""
    global _WEIGHT_CACHE
    if policy not in ("verify", "versions"):
        raise ValueError("uc_nerf_amd: weight cache policy must be 'verify' or 'versions', got %r" % (policy,))
    _WEIGHT_CACHE = policy


def set_source_precision(precision):
    """Element type of the channel-last copies of the gather sources (cascade volumes, source images, image features) that `rendering()`
    reads: "f32" (default: features bit-identical to sampling the tensors handed in) or "bf16" (SURVEY.md 8 configs[4] "fp32 MLP / bf16
    features": the copies are rounded to bf16 -- half the bytes of every gather corner; the MLP and the compositing stay as they are, and
    the backward still accumulates fp32 gradients for the fp32 tensors).  Not within the 1e-4 parity bar: an opt-in quality / speed trade
    (rendered images ~55 dB from the fp32-source render on the benchmark scene, tests/test_hip_round3.py)."""
    global _SOURCE_PRECISION
    if precision not in ("f32", "bf16"):
        raise ValueError("uc_nerf_amd: source precision must be 'f32' or 'bf16', got %r" % (precision,))
    _SOURCE_PRECISION = precision


def set_training_precision(precision):
    """MLP arithmetic of the forward of `rendering()` under autograd: "f32" (default: exact fp32 MFMA; gradients within 2e-4 of
    the reference's, element by element) or "bf16x3" (split-bf16 matrix cores: the forward launch takes about half the time;
    outputs and kept activations within ~1e-5 of the exact ones -- enough to put a pre-activation that lies within 1e-5 of zero
    on the other side of its relu, so gradients match in direction and norm (cosine > 0.9995), not element by element)."""
    global _TRAINING_PRECISION
    if precision not in ("f32", "bf16x3"):
        raise ValueError("uc_nerf_amd: training forward precision must be 'f32' or 'bf16x3', got %r" % (precision,))
    _TRAINING_PRECISION = precision


def inference_precision(args=None):
    prec = getattr(args, "inference_precision", None) or _INFERENCE_PRECISION
    if prec not in _INFERENCE_PRECISIONS:
        raise ValueError("uc_nerf_amd: args.inference_precision must be one of %s, got %r" % (", ".join(_INFERENCE_PRECISIONS), prec))
    return prec


# Parameters the reference's autograd never reaches: they keep grad = None here too.
_NO_GRAD = ("pts_bias_confidence_1.", "feature_linear_1.", "confi_linear.")


def _tensor_sig(t):
    """Identity of a tensor's CONTENT as far as torch can vouch for it: the base tensor object (by id, kept honest by a
    weakref), its version counter (bumped by every in-place write), and the view geometry."""
    if t is None:
        return None, None
    base = t._base if t._base is not None else t
    return (id(base), base._version, t.storage_offset(), tuple(t.shape), tuple(t.stride())), weakref.ref(base)


class FusedSession:
    def __init__(self, net):
        nerf = getattr(net, "nerf", net)
        self.n_src = nerf.view_num
        named = list(net.named_parameters())
        self.store = FlatStore.of(net)       # the parameters are views of ONE flat buffer (flat.py): no concatenation per step, one gradient buffer
        self.params = [p for _, p in named]
        self.grad_mask = [not any(tag in name for tag in _NO_GRAD) for name, _ in named]
        self.sizes = [p.numel() for p in self.params]
        self.weights = {}            # (precision, layout) -> (signature, flat, pw, wstream)
        self.passes = {}             # (precision, layout) -> ops.RenderPass
        self.src, self.src_sig, self.src_refs = None, None, None

    # ---- caches
    def invalidate(self):
        """Forget every packed weight stream and source copy (for callers that write parameters or sources behind torch's back while the
        "versions" weight cache is selected)."""
        self.weights.clear()
        self.src, self.src_sig, self.src_refs = None, None, None
        self._src_quick = self._src_keep = None

    def packed(self, precision, layout, fresh=False):
        """(flat parameter vector, packer, packed stream) of the network's CURRENT parameters.  The parameters live in one flat buffer
        (flat.FlatStore), so nothing is ever concatenated.
        fresh=True (calls under autograd): packed now, and `flat` is a SNAPSHOT of the buffer (one 0.7-MB copy) -- the backward takes its
        weight gradients against the parameters of this very forward, whatever an optimizer does in between.
        Otherwise the stream follows the weight-cache policy (set_weight_cache) and `flat` is the live buffer."""
        live = self.store.sync()
        dev = live.device
        if live.dtype != torch.float32:
            raise RuntimeError("uc_nerf_amd: network parameters must be float32")
        # (the 16-bit terms of the split precisions, ops.set_split_operand: a training forward keeps activations for a backward that splits them into
        #  bf16 terms, so it is always packed for those)
        pw = ops.PackedWeights.get(self.n_src, layout, dev, precision, operand=0 if fresh else None)
        key = (precision, layout)
        ent = self.weights.get(key)
        if ent is not None and ent.get("operand", 0) != pw.operand:       # (the terms changed since: another stream format)
            ent = None
        if fresh:
            # (the stream belongs to this call's autograd context until its backward has run: it is never entered in the cache, where a later
            #  no_grad call of the same precision would re-pack it in place -- with other values, had an optimizer stepped in between)
            flat = live.clone()
            return flat, pw, pw.pack(flat)
        if _WEIGHT_CACHE == "versions":
            # (the flat buffer's own counter too: flat.FlatAdam steps a Parameter that shares the buffer's storage and version counter, not the
            #  per-tensor ones -- round 4's advisor finding)
            # (and the store's generation, not the buffer's address: after a re-flattening -- module.to(), .float(), a CPU round trip -- the new
            #  buffer may sit where the old one did, with the same version count; round 5's stateful fuzzer, tests/fuzz_dropin.py, found exactly that)
            sig = (self.store.generation, live._version) + tuple(p._version for p in self.params)
            if ent is None or ent["sig"] != sig:
                ent = self.weights[key] = {"sig": sig, "ws": pw.pack(live), "operand": pw.operand}
            return live, pw, ent["ws"]
        # "verify": one launch re-packs the stream IN PLACE from the live buffer
        if ent is None or ent["ws"].device != dev:
            ent = self.weights[key] = {"sig": None, "ws": torch.empty(pw.n_stream, device=dev), "operand": pw.operand}
        pw.pack(live, out=ent["ws"])
        ent["sig"] = None
        return live, pw, ent["ws"]

    def sources(self, vols, conf, imgs, img_feat, w2cs, intrinsics):
        """GatherSources for these tensors.  Two signatures: the HEAVY sources (volumes, images, image features -- what the
        channel-last repack copies, ~150 MB of traffic) and the light ones (confidence map, poses).  Same heavy + same light:
        the cached object; same heavy only: a new object that inherits the channel-last copies."""
        # fast path (the 80 chunks of one image hand over the very same tensor objects): identities and version counters, no signatures
        quick = (tuple(map(id, vols)), id(imgs), id(img_feat), id(conf), id(w2cs), id(intrinsics),
                 tuple(t._version for t in vols), imgs._version, img_feat._version, conf._version, w2cs._version, intrinsics._version, _SOURCE_PRECISION)
        if self.src is not None and quick == getattr(self, "_src_quick", None):
            return self.src
        heavy = [_tensor_sig(t) for t in list(vols) + [imgs, img_feat]]
        light = [_tensor_sig(t) for t in (conf, w2cs, intrinsics)]
        hsig, lsig = tuple(s for s, _ in heavy), tuple(s for s, _ in light)
        old = self.src
        bf16 = _SOURCE_PRECISION == "bf16"
        heavy_ok = (old is not None and old.cl_bf16 == bf16 and hsig == self.src_sig[0]
                    and all(r is None or r() is not None for r in self.src_refs[0]))
        if heavy_ok and lsig == self.src_sig[1] and all(r is None or r() is not None for r in self.src_refs[1]):
            self._src_quick, self._src_keep = quick, (list(vols), imgs, img_feat, conf, w2cs, intrinsics)      # (kept alive: an id is only an identity while its object lives)
            return old
        self.src = ops.GatherSources(vols, conf, imgs, img_feat, w2cs, intrinsics, cl_bf16=bf16)
        if heavy_ok:
            self.src._cl, self.src.cl_all = old._cl, old.cl_all
        self.src_sig, self.src_refs = (hsig, lsig), ([r for _, r in heavy], [r for _, r in light])
        self._src_quick, self._src_keep = quick, (list(vols), imgs, img_feat, conf, w2cs, intrinsics)
        return self.src

    def render_pass(self, precision, layout, src, white_bkgd, fresh=False):
        flat, pw, ws = self.packed(precision, layout, fresh)
        rp = self.passes.get((precision, layout))
        if rp is None:
            rp = self.passes[(precision, layout)] = ops.RenderPass(src, pw, ws, white_bkgd=white_bkgd)
        if rp.src is not src:
            rp.set_sources(src)
        if rp.wstream is not ws:
            rp.set_weights(pw, ws)
        rp.set_white_bkgd(white_bkgd)
        if not rp.use_cl:
            rp.repack_sources(force=False)           # a copy made for these very sources through another precision's pass is reused
        if not src.zero_copy and torch.cuda.is_current_stream_capturing():
            # A step being captured into a HIP graph (train_step.GraphedStep) is replayed on whatever the caller copies into the static source
            # tensors: the repack must be IN the graph even though the tensors' identity and version say "unchanged" now (round 4's advisor
            # finding: a replay would otherwise gather from the warm-up's copies).  Sources read in place need nothing.
            rp.repack_sources(force=True)
        return rp, flat


_PRIVATE_SESSIONS = None        # set while a step is being captured into a HIP graph (private_sessions)


class private_sessions:
    """Context: every network rendered inside gets a FRESH FusedSession, returned in `self.sessions` -- for the capture of a step into a HIP graph
    (train_step.TrainStep).  A graph replays on the buffers it was captured with; the regular session's caches (workspaces, source copies, pose
    tensors) are re-allocated and dropped as eager calls with other shapes or sources come by, which would leave the graph with dangling
    pointers.  The private session is created inside the capture (its buffers come from the graph's own memory pool), used by nothing else, and
    kept alive by whoever keeps the graph.  Parameters and their flat store are shared with the regular session: they are the same tensors."""

    def __enter__(self):
        global _PRIVATE_SESSIONS
        self._outer, self.sessions = _PRIVATE_SESSIONS, {}
        _PRIVATE_SESSIONS = self.sessions
        return self

    def __exit__(self, *exc):
        global _PRIVATE_SESSIONS
        _PRIVATE_SESSIONS = self._outer
        return False


def session_of(net):
    if _PRIVATE_SESSIONS is not None:
        s = _PRIVATE_SESSIONS.get(id(net))
        if s is None:
            s = _PRIVATE_SESSIONS[id(net)] = FusedSession(net)
        return s
    s = net.__dict__.get("_ucnerf_session")
    if s is None or s.params[0] is not next(net.parameters()):
        s = FusedSession(net)
        net.__dict__["_ucnerf_session"] = s
    return s


class _FusedRender(torch.autograd.Function):
    """One rendering() call under autograd.  Differentiable inputs: the three cascade volumes, confidence, img_feat and the
    network parameters (never positions: SURVEY.md 3.2)."""

    @staticmethod
    def forward(ctx, sess, layout, white_bkgd, coords, z, rays_dir, angle, imgs, w2cs, intrinsics, vol1, vol2, vol3, conf, img_feat, *params):
        src = sess.sources([vol1, vol2, vol3], conf, imgs, img_feat, w2cs, intrinsics)
        rp, flat = sess.render_pass(_TRAINING_PRECISION, layout, src, white_bkgd, fresh=True)       # packed from the parameters as they are NOW
        out = rp(rays_dir, z, want=(), keep=("raw", "feats"), dir_feat=angle, coords=coords)
        ctx.sess, ctx.rp, ctx.src, ctx.pw, ctx.ws, ctx.flat, ctx.white_bkgd = sess, rp, src, rp.pw, rp.wstream, flat, white_bkgd
        ctx.layout = layout
        ctx.coords, ctx.kept = coords, {"raw": out["raw"], "feats": out["feats"], "feats_tiled": out.get("feats_tiled", False)}
        ctx.geom = (z, rays_dir, angle)
        ctx.shapes = tuple(t.shape for t in (vol1, vol2, vol3, conf, img_feat))
        return out["rgb"], out["depth"]

    @staticmethod
    def backward(ctx, g_rgb, g_depth):
        rp, sess = ctx.rp, ctx.sess
        z, rays_dir, angle = ctx.geom
        if rp.src is not ctx.src:                    # another forward re-bound the pass since: bind this call's state again
            rp.set_sources(ctx.src)
        if rp.wstream is not ctx.ws:
            rp.set_weights(ctx.pw, ctx.ws)
        rp.set_white_bkgd(ctx.white_bkgd)
        if g_rgb is None:
            g_rgb = torch.zeros(z.shape[0], 3, device=z.device)
        need = tuple(ctx.needs_input_grad[10:15])                            # vol1, vol2, vol3, conf, img_feat
        need = (need[0], need[1], need[2], need[3], need[4])
        f32w = None
        if rp.pw.cfg.precision != 0 and not rp.saved_matches(z.shape[0], z.shape[1], ctx.kept["raw"]):
            # another forward overwrote the kept activations: recompute them exactly -- from the parameters of THIS call's forward
            # (ctx.flat, which the weight gradients below are taken against), not from whatever the parameters hold by now
            pw32 = ops.PackedWeights.get(sess.n_src, ctx.layout, z.device, "f32")
            f32w = (pw32, pw32.pack(ctx.flat))
        store = sess.store
        g_flat, gv1, gv2, gv3, gc, gi = rp.backward(rays_dir, z, ctx.kept, g_rgb.contiguous(), g_depth, ctx.flat, need=need,
                                                    coords=ctx.coords, dir_feat=angle, f32_weights=f32w, flat_room=store.grad_room)
        grads = [g.reshape(s) if g is not None else None for g, s in zip((gv1, gv2, gv3, gc, gi), ctx.shapes)]
        # views of the ONE flat gradient vector the kernels wrote: autograd installs them as p.grad without a copy (flat.py), so the
        # step's gradients stay one buffer -- what FlatGradBucket all-reduces in place and FlatAdam steps in one launch
        req = ctx.needs_input_grad[15:]
        if not all(req):                             # frozen tensors: their segments of the flat gradient stay zero (nobody steps or reduces them)
            store.zero_segments(g_flat, [has and not r for has, r in zip(sess.grad_mask, req)])
        g_params = store.grad_views(g_flat, [has and r for has, r in zip(sess.grad_mask, req)])
        # (ctx.kept stays: a second backward over the same graph -- retain_graph=True -- finds the kept activations overwritten and
        #  recomputes them through the route above)
        return (None,) * 10 + tuple(grads) + tuple(g_params)


def fused_rendering(net, layout, args, w2c_dir, rays_pts, rays_ndc, z, rays_dir, vols, imgs, img_feat, conf, w2cs, intrinsics,
                    white_bkgd=False, extras=()):
    """The body of rendering() on the fast kernels.  Returns (rgb [N,3], depth [N]) or, with `extras` (any of "acc",
    "weights", "var", "u", "wu"; inference only), (rgb, depth, dict)."""
    sess = session_of(net)
    N, S = z.shape[0], z.shape[1]
    F = 24 + 12 * sess.n_src + 1
    max_rays = max(1, _MAX_FEATURE_BYTES // (((S + 31) // 32 * 32) * F * 4))
    if N > max_rays:                                 # the reference's netchunk loop, only where the address range forces it
        parts = [fused_rendering(net, layout, args, w2c_dir, rays_pts[a:a + max_rays], {k: v[a:a + max_rays] for k, v in rays_ndc.items()},
                                 z[a:a + max_rays], rays_dir[a:a + max_rays], vols, imgs, img_feat, conf, w2cs, intrinsics, white_bkgd, extras)
                 for a in range(0, N, max_rays)]
        if extras:
            return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]),
                    {k: torch.cat([p[2][k] for p in parts]) for k in parts[0][2]})
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    coords = {"pts": rays_pts, "stage1": rays_ndc["stage1"], "stage2": rays_ndc["stage2"], "stage3": rays_ndc["stage3"], "ndc": rays_ndc["ndc"]}
    diff = [t for t in list(vols) + [conf, img_feat] if t is not None and t.requires_grad]
    train = torch.is_grad_enabled() and (bool(diff) or any(p.requires_grad for p in sess.params))
    if train:
        if extras:
            raise RuntimeError("uc_nerf_amd.rendering: extras are an inference-time option (call under torch.no_grad())")
        angle, _ = ops.dir_feature(rays_dir, w2c_dir)
        return _FusedRender.apply(sess, layout, bool(white_bkgd), coords, z, rays_dir, angle, imgs, w2cs, intrinsics,
                                  vols[0], vols[1], vols[2], conf, img_feat, *sess.params)
    src = sess.sources(vols, conf, imgs, img_feat, w2cs, intrinsics)
    prec = inference_precision(args)
    if prec == "bf16x3_fused" and ("u" in extras or "wu" in extras or _SOURCE_PRECISION == "bf16" or not src.full):
        # the gather-fused kernel keeps nothing per sample, reads fp32 copies when it is handed coordinates and gathers every unit:
        # the same arithmetic on the two-kernel pass otherwise
        prec = "bf16x3"
    rp, _ = sess.render_pass(prec, layout, src, white_bkgd)
    # the view-direction feature: the gather-fused pass takes the rotation as the device tensor it is (round 5: its tail route -- chunks of up to
    # three rounds of tiles, e.g. the 1024 x 90 of train.py:254-272 -- makes the features, gathers, evaluates and composites in ONE launch);
    # the other passes get them from ucnerf_dir_feature as before
    on_dev = (prec == "bf16x3_fused" and torch.is_tensor(w2c_dir) and w2c_dir.is_cuda and w2c_dir.dtype == torch.float32 and w2c_dir.dim() == 2
              and w2c_dir.shape[0] >= 3 and w2c_dir.shape[1] == 4 and w2c_dir.is_contiguous())
    if on_dev:
        out = rp(rays_dir, z, want=tuple(extras), coords=coords, w2c_dir_dev=w2c_dir)
    else:
        angle, _ = ops.dir_feature(rays_dir, w2c_dir)
        out = rp(rays_dir, z, want=tuple(extras), dir_feat=angle, coords=coords)
    if extras:
        return out["rgb"], out["depth"], {k: out[k] for k in extras if k in out}
    return out["rgb"], out["depth"]


# ------------------------------------------------------------------------------------------------ query-function probe
class _Probed(Exception):
    pass


class _FusionProbe:
    """Stands where the network goes in one dry call of a `network_query_fn`: if the query function is this package's
    run_network_mvs with fusable embedders it calls forward_raw(pts, viewdirs, feats, layout) -- which records the layout."""

    def __init__(self):
        self.layout = None

    def forward_raw(self, pts, viewdirs, feats, pe_layout=0):
        self.layout = pe_layout
        raise _Probed()


_probe_cache = weakref.WeakKeyDictionary()


def query_layout(network_query_fn):
    """PE layout (0 / 1) if `network_query_fn(pts, viewdirs, feats, fn)` is the fusable encode -> cat -> MLP of this package's
    run_network_mvs (what create_ucnerf builds, network/models.py:233-236), else None.  Found by one dry call with sentinels."""
    try:
        return _probe_cache[network_query_fn]
    except (KeyError, TypeError):
        pass
    probe, layout = _FusionProbe(), None
    try:
        network_query_fn(object(), object(), object(), probe)
    except _Probed:
        layout = probe.layout
    except Exception:
        layout = None
    try:
        _probe_cache[network_query_fn] = layout
    except TypeError:
        pass
    return layout
