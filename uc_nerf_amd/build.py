"""Builds libucnerf_hip.so (gfx950 only) in-tree with hipcc.  `python -m uc_nerf_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the working tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libucnerf_hip.so")
SOURCES = ["rays.hip", "gather.hip", "gather_cl.hip", "mlp.hip", "mlp_bf16.hip", "mlp_bwd.hip", "mlp_bwd_chain.hip", "mlp_wgrad.hip", "composite.hip", "sample_pdf.hip", "render.hip", "mvs.hip"]
# (source, object name, extra flags): translation units built more than once with different switches
VARIANTS = [("mlp_bf16.hip", "mlp_bf16_plain.o", ["-DUCNERF_BF16_BUILD_TERMS=1"])]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mlp_layout.h"), os.path.join(CSRC, "sincos_cw.h"), os.path.join(CSRC, "gather_cl_device.h"), os.path.join(CSRC, "mlp_bwd_parts.h"),
           os.path.join(HERE, "..", "include", "ucnerf_hip.h")]
# -ffp-contract=off: the sample_pdf / sampling kernels reproduce torch-CPU roundings (separate mul and add)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    for s, oname, extra in VARIANTS:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, oname)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=6) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                print(err)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs] + [os.path.join(OBJ, o) for _, o, _ in VARIANTS]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
