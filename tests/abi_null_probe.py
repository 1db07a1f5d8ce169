"""Child process of tests/test_abi_host.py::test_entry_points_refuse_null_and_empty_arguments_without_crashing: calls every (params*, stream) entry point
of the C ABI with (1) a NULL params pointer, (2) an all-zero struct, (3) a struct whose integer fields hold small positive sizes and whose pointers
are all NULL, (4) the same with every pointer a small non-null dummy and ONE size field out of range.  None may crash; (1) and (3) must return an
error with a message; (2) returns success (an empty batch) or an error.  Runs without a GPU: a call that passes validation would need a device, so a
non-error return from (3) / (4) is reported too.  Prints one JSON line."""
import ctypes as C
import json
import sys

from uc_nerf_amd import _lib as L

ENTRY = {
    "ucnerf_ray_gen": L.RayGenParams, "ucnerf_ndc_rays": L.NdcRaysParams, "ucnerf_dir_feature": L.DirFeatureParams,
    "ucnerf_sample_stratified": L.SampleStratifiedParams, "ucnerf_sample_cascade": L.SampleCascadeParams, "ucnerf_ndc_project": L.NdcProjectParams,
    "ucnerf_build_rays_test": L.BuildRaysTestParams, "ucnerf_embed": L.EmbedParams, "ucnerf_feat_gather_fwd": L.FeatGatherParams,
    "ucnerf_feat_gather_bwd": L.FeatGatherBwdParams, "ucnerf_mlp_fwd": L.MlpParams, "ucnerf_mlp_bwd": L.MlpBwdParams,
    "ucnerf_composite_fwd": L.CompositeParams, "ucnerf_composite_bwd": L.CompositeBwdParams, "ucnerf_sample_pdf": L.SamplePdfParams,
    "ucnerf_merge_rows": L.MergeRowsParams, "ucnerf_cost_volume": L.CostVolumeParams, "ucnerf_depth_regress": L.DepthRegressParams,
    "ucnerf_cost_volume_bwd": L.CostVolumeBwdParams, "ucnerf_depth_regress_bwd": L.DepthRegressBwdParams, "ucnerf_render_fused_fwd": L.RenderParams,
    "ucnerf_render_fused_bwd": L.RenderBwdParams,
}
INTS = (C.c_int, C.c_int32, C.c_int64, C.c_uint32, C.c_long)


def fill(obj, size, ptr):
    """Every integer field := size, every pointer-like field := ptr, nested structs and arrays recursively (floats stay 0)."""
    for name, tp in obj._fields_:
        if issubclass(tp, C.Structure):
            fill(getattr(obj, name), size, ptr)
        elif issubclass(tp, C.Array):
            arr = getattr(obj, name)
            et = tp._type_
            for i in range(tp._length_):
                if issubclass(et, C.Structure):
                    fill(arr[i], size, ptr)
                elif et in INTS:
                    arr[i] = size
                elif et in (C.c_void_p, C.c_uint64, C.c_size_t):
                    arr[i] = ptr
        elif tp in INTS:
            setattr(obj, name, size)
        elif tp in (C.c_void_p, C.c_uint64, C.c_size_t):
            setattr(obj, name, ptr)


def main():
    lib = L.lib()
    out = {"calls": 0, "problems": []}
    for name, cls in ENTRY.items():
        fn = getattr(lib, name)
        rc = fn(None, None)
        out["calls"] += 1
        if rc == 0 or not lib.ucnerf_last_error():
            out["problems"].append("%s(NULL) returned %d" % (name, rc))
        z = cls()
        rc = fn(C.addressof(z), None)                                    # all zero: an empty batch or an error, never a crash
        out["calls"] += 1
        p = cls()
        fill(p, 4, 0)
        rc = fn(C.addressof(p), None)
        out["calls"] += 1
        if rc == 0:
            out["problems"].append("%s with sizes 4 and NULL pointers returned success" % name)
        for bad in (-1, 1 << 30):                                        # every size negative / absurd, pointers non-null dummies (never dereferenced)
            q = cls()
            fill(q, bad, 64)
            rc = fn(C.addressof(q), None)
            out["calls"] += 1
            if rc == 0:
                out["problems"].append("%s with every size %d returned success" % (name, bad))
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
