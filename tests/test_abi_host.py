"""CPU-side checks of the C ABI and host logic (no GPU, no compute calls): the library loads, exports every symbol
include/ucnerf_hip.h declares, the ctypes mirrors match, argument validation reports errors, and the host-built
weight-pack index is a faithful rearrangement of the flat parameter vector."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from uc_nerf_amd.build import build
    build()
    from uc_nerf_amd import _lib
    _lib.lib()
    return _lib


def test_every_declared_symbol_is_exported_and_bound(L):
    hdr = open(os.path.join(ROOT, "include", "ucnerf_hip.h")).read()
    declared = set(re.findall(r"\b(ucnerf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    raw = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "library does not export " + name
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    structs = set(re.findall(r"\}\s*(ucnerf_[a-z0-9_]+)\s*;", hdr))
    assert structs == set(L.STRUCTS), structs ^ set(L.STRUCTS)
    for cname, cls in L.STRUCTS.items():
        assert L.lib().ucnerf_sizeof(cname.encode()) == C.sizeof(cls), cname
    assert L.lib().ucnerf_sizeof(b"nope") == -1


# switches that select between CORRECT variants, with the value a production build carries; everything else (…_EXP, …_DIAG_ON, NO_PK, SPLIT_DOT,
# FOOT_UNDER_GEMM, GATHER_RUN) is an experiment and must read 0
PRODUCTION_FLAGS = {"UCNERF_BF16_BW": "8", "UCNERF_BF16_NBUF": "4", "UCNERF_BF16_WPS": "2", "UCNERF_BF16_HINT_V": "5", "UCNERF_BF16_IDLE_SKIP": "1",
                    "UCNERF_BF16_WAVE_MAJOR": "1", "UCNERF_MLP_WAVES": "8", "UCNERF_MLP_PRIO": "3", "UCNERF_MLP_RING": "4", "UCNERF_TN_BF16X3": "1",
                    "UCNERF_TN_DEPTH2": "2", "UCNERF_GATHER_WAVES": "1", "UCNERF_CHAIN_WAVES": "4", "UCNERF_MLP_SAVE_NT": "1"}


def test_library_was_built_with_production_switches(L):
    """ucnerf_build_flags() names every compile-time switch of the kernels: no wrong-result experiment may be live in the shipped binary."""
    flags = L.lib().ucnerf_build_flags().decode()
    pairs = re.findall(r"(UCNERF_[A-Z0-9_]+)=(\S+)", flags)
    assert len(pairs) >= 18 and {"mlp_bf16x3", "mlp_bf16_plain", "mlp_f32", "mlp_bwd", "gather_cl"} <= set(re.findall(r"(\w+):", flags)), flags
    for name, value in pairs:
        assert value == PRODUCTION_FLAGS.get(name, "0"), "%s=%s in the shipped library (%s)" % (name, value, flags)


def test_library_belongs_to_the_sources_in_this_tree(L):
    """The digest linked into the binary (uc_nerf_amd/build.py) equals the digest of the sources beside it: a stale or foreign .so is caught
    here instead of producing results of some other code (on the GPU box the tree travels with its prebuilt library)."""
    from uc_nerf_amd import build as B
    if os.environ.get("UCNERF_LIB"):
        pytest.skip("an A/B variant library is selected")
    assert L.lib().ucnerf_source_hash().decode() == B.source_hash()
    assert isinstance(B.built_here(), bool)


def test_argument_validation_reports_instead_of_launching(L):
    lib = L.lib()
    assert lib.ucnerf_ray_gen(None, None) == -1 and b"null params" in lib.ucnerf_last_error()
    p = L.SamplePdfParams()
    p.n, p.n_bins, p.n_samples = 4, 5000, 8
    p.weights = p.u = p.bins = 8          # non-null dummies; never dereferenced because validation fails first
    assert lib.ucnerf_sample_pdf(C.addressof(p), None) == -1 and b"n_bins" in lib.ucnerf_last_error()
    q = L.CompositeParams()
    q.n, q.S, q.raw, q.z, q.rgb_map, q.depth_map = 2, 5000, 16, 16, 16, 16
    assert lib.ucnerf_composite_fwd(C.addressof(q), None) == -1 and b"outside 1..1024" in lib.ucnerf_last_error()
    cfg = L.MlpConfig(12, 0)
    assert lib.ucnerf_mlp_param_count(C.addressof(cfg)) < 0 and b"n_src" in lib.ucnerf_last_error()
    e = L.RayGenParams()                   # empty batch: succeeds without touching any pointer or the device
    assert lib.ucnerf_ray_gen(C.addressof(e), None) == 0
    assert lib.ucnerf_render_workspace_floats(4096, 192, 6) > 4096 * 192 * 97
    assert lib.ucnerf_render_workspace_floats(1, 0, 6) < 0
    # ABI v4 entry points and fields: arguments are checked before anything is launched
    assert lib.ucnerf_composite_sample_pdf(None, None, None) == -1 and b"null params" in lib.ucnerf_last_error()
    c = L.CompositeParams()
    c.n, c.S, c.raw, c.z, c.rgb_map, c.depth_map = 4, 64, 16, 16, 16, 16
    s = L.SamplePdfParams()
    s.n, s.n_bins, s.n_samples, s.n_merge, s.from_coarse, s.u, s.samples = 4, 62, 128, 64, 1, 16, 16            # n_bins must be S - 1
    assert lib.ucnerf_composite_sample_pdf(C.addressof(c), C.addressof(s), None) == -1 and b"from_coarse form over the composited pass" in lib.ucnerf_last_error()
    s.n_bins, s.from_coarse = 63, 0
    assert lib.ucnerf_composite_sample_pdf(C.addressof(c), C.addressof(s), None) == -1
    s.from_coarse, s.z_merge = 1, 32                                                                              # z_merge, when given, must be the pass's own depths
    assert lib.ucnerf_composite_sample_pdf(C.addressof(c), C.addressof(s), None) == -1 and b"z_merge" in lib.ucnerf_last_error()
    r = L.RenderParams()
    r.n, r.S, r.cfg = 4, 64, L.MlpConfig(6, 0, 3)
    r.rays_o = r.rays_d = r.z = r.workspace = r.wstream = r.rgb_map = r.depth_map = 16
    rg = L.RayGenParams()
    r.gen_rays = C.addressof(rg)                                                                                  # gen_rays without gen_depths
    assert lib.ucnerf_render_fused_fwd(C.addressof(r), None) == -1 and b"go together" in lib.ucnerf_last_error()
    ss = L.SampleStratifiedParams()
    r.gen_depths = C.addressof(ss)
    rg.n, ss.n, ss.S = 4, 4, 32                                                                                   # depths for another S
    assert lib.ucnerf_render_fused_fwd(C.addressof(r), None) == -1 and b"differ from the pass" in lib.ucnerf_last_error()
    r.cfg = L.MlpConfig(6, 0, 1)                                                                                   # generated rays need the gather-fused kernel
    assert lib.ucnerf_render_fused_fwd(C.addressof(r), None) == -1 and b"gather-fused kernel" in lib.ucnerf_last_error()


@pytest.mark.parametrize("n_src,layout", [(6, 0), (3, 0), (6, 1), (1, 0), (8, 1)])
def test_pack_index_is_a_rearrangement_of_the_parameters(L, n_src, layout):
    from uc_nerf_amd.synthetic import ucnerf_param_shapes
    lib = L.lib()
    cfg = L.MlpConfig(n_src, layout)
    n_params, n_stream = lib.ucnerf_mlp_param_count(C.addressof(cfg)), lib.ucnerf_mlp_stream_count(C.addressof(cfg))
    shapes = ucnerf_param_shapes(n_src)
    assert n_params == sum(int(np.prod(s)) for _, s in shapes)
    idx = np.empty(n_stream, np.int32)
    assert lib.ucnerf_mlp_pack_index(C.addressof(cfg), idx.ctypes.data) == 0
    assert idx.min() == -1 and idx.max() < n_params
    used = idx[idx >= 0]
    counts = np.bincount(used, minlength=n_params)
    off, unused = 0, {"nerf.pts_bias_confidence_1", "nerf.feature_linear_1", "nerf.confi_linear"}
    for name, shp in shapes:
        n = int(np.prod(shp))
        c = counts[off:off + n]
        if name.rsplit(".", 1)[0] in unused:
            assert not c.any(), name                       # present for checkpoint compatibility, never read
        else:
            assert c.min() >= 1, name                      # every live weight reaches the kernel
            assert c.max() <= 2, name                      # at most one extra copy (the ring's wrap-around k-steps)
        off += n
    # k-step region: 256 floats per k-step, all sections multiples of the 4-deep prefetch ring
    ring = 4
    ks = ((24 + 4 * n_src) // 2 + ring - 1) // ring * ring + (4 * n_src + ring - 1) // ring * ring + 32 + 4 * 64 + (32 + 64) + 64 + (64 + 16)
    assert n_stream == (ks + ring) * 256 + 10 * 128 + 2 * 516
    assert np.array_equal(idx[:ring * 256], idx[ks * 256:(ks + ring) * 256])      # wrap copy of the first ring
    # the two encoding layouts permute the same columns of the first layer
    if layout == 1:
        cfg0 = L.MlpConfig(n_src, 0)
        idx0 = np.empty(n_stream, np.int32)
        lib.ucnerf_mlp_pack_index(C.addressof(cfg0), idx0.ctypes.data)
        assert not np.array_equal(idx, idx0) and np.array_equal(np.sort(idx), np.sort(idx0))


def test_tail_route_diagnostics_answer_without_a_device(L):
    """ucnerf_fused_tail_fits / ucnerf_fused_tail_launches (the route on which a small pass composites, re-samples and generates its rays inside the
    gather-fused launch): callable on a host without a GPU -- no device, no route, nothing counted, no crash."""
    lib = L.lib()
    assert lib.ucnerf_fused_tail_launches() == 0
    if not torch.cuda.is_available():
        assert lib.ucnerf_fused_tail_fits(512, 64) == 0
    assert lib.ucnerf_fused_tail_fits(512, 63) == 0 and lib.ucnerf_fused_tail_fits(0, 64) == 0 and lib.ucnerf_fused_tail_fits(512, 512) == 0


def test_product_package_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import uc_nerf_amd, uc_nerf_amd.ops, uc_nerf_amd.pipeline, uc_nerf_amd.parallel;"
            "uc_nerf_amd.install_dropin(); import network.renderer, network.models, utils.utils, utils.run_nerf_helpers, data.ray_utils;"
            "assert not any(m.startswith('oracle') for m in sys.modules), 'oracle leaked into the product'") % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "uc_nerf_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read(), f


def test_cpu_tensors_are_refused_loudly():
    from uc_nerf_amd import ops
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.embed(torch.zeros(4, 3), 10)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.composite_fwd(torch.zeros(2, 8, 4), torch.zeros(2, 8))


def test_gradient_chain_counted_waits_never_exceed_the_operations_actually_issued(tmp_path):
    """mlp_bwd_chain_kernel waits for a slot of its LDS weight ring with `s_waitcnt vmcnt(6 + E)`, E = a compile-time count of the loads and
    stores every wave has issued since the awaited copy (the compiler does not see the asm copies, so nothing else orders them).  An E larger
    than what the compiled code really issues would let a wave read a slot before its copy has landed -- silently.  This compiles the
    translation unit to ISA and replays the kernel's vector-memory stream: for every one of the 128 waits of the tile body, the number of
    vector-memory instructions issued behind the awaited copy must be at least the count waited for."""
    import shutil
    import subprocess
    from uc_nerf_amd import build as B
    hipcc = B._hipcc()
    if not (os.path.isabs(hipcc) and os.path.exists(hipcc)) and shutil.which(hipcc) is None:
        pytest.skip("hipcc not available")
    out = tmp_path / "chain.s"
    cmd = [hipcc] + B.FLAGS + ["--cuda-device-only", "-S", os.path.join(B.CSRC, "mlp_bwd_chain.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN6ucnerf20mlp_bwd_chain_kernel"))
    events = []                                   # ("dma" | "vm" | "adv", vmcnt)
    pending_wait = None
    for l in lines[start:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if t.startswith("s_waitcnt") and "vmcnt(" in t and "lgkmcnt(0)" in t:
            pending_wait = int(t.split("vmcnt(")[1].split(")")[0])
            continue
        if t.startswith("s_barrier"):
            if pending_wait is not None:
                events.append(("adv", pending_wait))
            pending_wait = None
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        pending_wait = None if not t.startswith("s_") and not t.startswith("v_") else pending_wait
        if t.startswith("global_load_lds"):
            events.append(("dma", 0))
        elif t.startswith(("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "flat_load", "flat_store", "scratch_")):
            events.append(("vm", 0))
    n_adv = sum(1 for e in events if e[0] == "adv")
    n_dma = sum(1 for e in events if e[0] == "dma")
    assert n_adv == 128 and n_dma == 128 + 8, (n_adv, n_dma)          # 128 half-steps per tile; eight copies in the prologue
    eighth = [i for i, e in enumerate(events) if e[0] == "dma"][7]     # the prologue's last copy: the tile loop's body starts behind it
    body = events[eighth + 1:]
    seq = body + body                                                  # two iterations of the tile loop
    dma_idx = [i for i, e in enumerate(seq) if e[0] == "dma"]
    adv_idx = [i for i, e in enumerate(seq) if e[0] == "adv"]
    assert len(dma_idx) == 256 and len(adv_idx) == 256
    checked = 0
    for j in range(128):                                               # the advances of the second iteration
        a_i = adv_idx[128 + j]
        awaited = dma_idx[128 + j + 1 - 8]                             # the copy of half-step j + 1, issued seven advances earlier
        younger = sum(1 for e in seq[awaited + 1:a_i] if e[0] in ("dma", "vm"))
        assert seq[a_i][1] <= younger, "half-step %d waits for vmcnt(%d) but only %d operations follow the awaited copy" % (j, seq[a_i][1], younger)
        assert seq[a_i][1] >= 6
        checked += 1
    assert checked == 128


def test_entry_points_refuse_null_and_empty_arguments_without_crashing():
    """Every (params*, stream) entry point of include/ucnerf_hip.h, in a child process (a crash must not take the test run with it): NULL params, an all-zero
    struct, positive sizes with NULL arrays, negative and absurd sizes with dummy arrays -- an error code and a message every time (an all-zero struct
    may also be an empty batch: success), never a crash, never a launch.  110 calls; runs without a GPU."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_null_probe.py")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, "the probe died (exit %d): %s" % (r.returncode, r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["calls"] >= 100 and not out["problems"], out["problems"]


def test_split_operand_setting_is_part_of_a_packer_identity():
    """ucnerf_mlp_config.operand (ABI v6) on the host side: the module setting selects the packer (and with it the stream format), "f32" ignores it,
    an unknown kind is refused, and the struct mirror carries the field where the header has it."""
    from uc_nerf_amd import _lib as L
    from uc_nerf_amd import ops
    assert [f[0] for f in L.MlpConfig._fields_] == ["n_src", "pe_layout", "precision", "operand"]
    assert L.lib().ucnerf_sizeof(b"ucnerf_mlp_config") == C.sizeof(L.MlpConfig) == 16
    dev = torch.device("cpu")
    try:
        a = ops.PackedWeights.get(6, 0, dev, "bf16x3_fused")
        ops.set_split_operand("fp16")
        b = ops.PackedWeights.get(6, 0, dev, "bf16x3_fused")
        assert a is not b and (a.operand, b.operand) == (0, 1) and a.n_stream == b.n_stream
        assert ops.PackedWeights.get(6, 0, dev, "f32").operand == 0
        assert ops.PackedWeights.get(6, 0, dev, "bf16x3_fused", operand="bf16") is a
        with pytest.raises(ValueError):
            ops.set_split_operand("tf32")
    finally:
        ops.set_split_operand("bf16")
    assert ops.PackedWeights.get(6, 0, dev, "bf16x3_fused") is a
