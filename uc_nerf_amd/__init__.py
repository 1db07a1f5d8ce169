"""uc_nerf_amd -- MI355X (gfx950) implementation of UC-NeRF's ray-marching volume-render hot path.

Layers:
  csrc/ + include/ucnerf_hip.h   hand-written HIP kernels behind a C ABI (libucnerf_hip.so)
  _lib.py, ops.py                ctypes binding and torch-tensor wrappers (plumbing)
  network/, utils/, data/        mirrors of the reference's Python call surface for this path
  pipeline.py, parallel.py       the fused coarse+fine renderer and ray-sharded multi-GPU driver

GPU only: there is no CPU or PyTorch fallback; calls on CPU tensors raise.
"""
__version__ = "0.5.0"


def set_inference_precision(precision):
    """MLP arithmetic of the `rendering()` drop-in under torch.no_grad(): "bf16x3_fused" (default: the headline kernel, feature gather inside
    the split-bf16 MLP kernel, within the 1e-4 parity bar), "bf16x3", "f32" (exact fp32 MFMA, the opt-out) or "bf16" (dropin.py)."""
    from . import dropin
    dropin.set_inference_precision(precision)


def set_weight_cache(policy):
    """How a no_grad `rendering()` call obtains its packed weight stream: "verify" (default: re-packed from the live parameters in every
    call, nothing can go stale) or "versions" (cached per parameter version counter: one launch less per call) -- dropin.set_weight_cache."""
    from . import dropin
    dropin.set_weight_cache(policy)


def set_source_precision(precision):
    """Element type of the channel-last source copies `rendering()` gathers from: "f32" (default) or "bf16" (configs[4]'s "bf16 features":
    an opt-in quality / speed trade, dropin.set_source_precision)."""
    from . import dropin
    dropin.set_source_precision(precision)


def set_training_precision(precision):
    """MLP arithmetic of the training forward of the `rendering()` drop-in: "f32" (default) or "bf16x3" (dropin.py)."""
    from . import dropin
    dropin.set_training_precision(precision)


def set_split_operand(kind):
    """The 16-bit terms of the split precisions: "bf16" (default: 8 significant bits per term, float32's range) or "fp16" (11 bits per term at the
    same matrix-core rate: renders at float32 level, no measurable kernel time (+0.1 %); fp16's range -- an activation beyond 65 504 overflows) -- ops.set_split_operand."""
    from . import ops
    ops.set_split_operand(kind)


def set_backward_mode(mode):
    """Backward of the network: "chain" (default: one register-resident gradient-chain launch + one weight-gradient launch) or "layerwise" (the
    exact-fp32 layer-by-layer kernels of rounds 1-2, kept as the cross-check) -- ops.set_backward_mode."""
    from . import ops
    ops.set_backward_mode(mode)


def install_dropin(precision=None, training_precision=None, split_operand=None):
    """Registers this package's mirrors under the reference's module names (`network.renderer`,
    `network.models`, `utils.utils`, `utils.run_nerf_helpers`, `data.ray_utils`) so that the reference's
    train.py imports resolve here unchanged.  See INTEGRATION.md.
    precision: MLP arithmetic of `rendering()` under torch.no_grad() (None keeps the default, "bf16x3_fused"; "f32" = exact fp32 MFMA);
    training_precision: of the training forward (None keeps "f32");
    split_operand: the 16-bit terms of the split precisions under no_grad, "bf16" or "fp16" (None keeps the current setting: "bf16" unless
    UCNERF_SPLIT_OPERAND says otherwise) -- set_split_operand."""
    import importlib
    if precision is not None:
        set_inference_precision(precision)
    if split_operand is not None:
        set_split_operand(split_operand)
    if training_precision is not None:
        set_training_precision(training_precision)
    import sys
    import types
    for pkg in ("network", "utils", "data"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    for name in ("network.renderer", "network.models", "utils.utils", "utils.run_nerf_helpers", "data.ray_utils"):
        mod = importlib.import_module("uc_nerf_amd." + name)
        sys.modules[name] = mod
        setattr(sys.modules[name.split(".")[0]], name.split(".")[1], mod)
