"""ctypes binding of libucnerf_hip.so (the C ABI declared in include/ucnerf_hip.h).

There is no CPU fallback anywhere in this package: if the library is missing or a call fails, a
RuntimeError is raised.  Struct mirrors are checked against the library's own sizeof() table at load time.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UCNERF_LIB") or os.path.join(_HERE, "libucnerf_hip.so")   # override: A/B builds

ABI_VERSION = 6    # UCNERF_ABI_VERSION of include/ucnerf_hip.h this binding mirrors

fp = C.POINTER(C.c_float)
i32 = C.c_int32
f32 = C.c_float
vp = C.c_void_p     # device pointers travel as integers


class RayGenParams(C.Structure):
    _fields_ = [("n", i32), ("H", i32), ("W", i32), ("grid_start", i32), ("opengl", i32), ("K", f32 * 9),
                ("c2w", f32 * 12), ("xs", vp), ("ys", vp), ("rays_d", vp), ("rays_o", vp), ("pix", vp), ("w2c_dir", f32 * 12), ("angle", vp)]


class NdcRaysParams(C.Structure):
    _fields_ = [("n", i32), ("H", i32), ("W", i32), ("variant", i32), ("focal_x", f32), ("focal_y", f32),
                ("near", f32), ("rays_o", vp), ("rays_d", vp), ("out_o", vp), ("out_d", vp)]


class DirFeatureParams(C.Structure):
    _fields_ = [("n", i32), ("has_ref", i32), ("repeat", i32), ("w2c_ref", f32 * 12), ("rays_d", vp), ("angle", vp),
                ("cos_angle", vp), ("w2c_ref_dev", vp)]


class SampleStratifiedParams(C.Structure):
    _fields_ = [("n", i32), ("S", i32), ("lindisp", i32), ("perturb", f32), ("near", f32), ("far", f32), ("rays", vp), ("noise", vp), ("z", vp),
                ("pts", vp)]


class SampleCascadeParams(C.Structure):
    _fields_ = [("n", i32), ("S", i32), ("near_far", vp), ("t_rand", vp), ("rays_o", vp), ("rays_d", vp), ("z", vp),
                ("pts", vp)]


class NdcProjectParams(C.Structure):
    _fields_ = [("m", i32), ("has_w2c", i32), ("sample_2d", i32), ("nf_stride", i32), ("w2c", f32 * 12), ("K", f32 * 9),
                ("inv_scale", f32 * 2), ("near", f32), ("far", f32), ("pts", vp), ("near_1", vp), ("far_1", vp),
                ("near_2", vp), ("far_2", vp), ("near_3", vp), ("far_3", vp), ("out_stage1", vp), ("out_stage2", vp),
                ("out_stage3", vp), ("out_ndc", vp)]


class BuildRaysTestParams(C.Structure):
    _fields_ = [("n", i32), ("S", i32), ("H", i32), ("W", i32), ("grid_start", i32), ("dv_d", i32 * 3), ("dv_h", i32 * 3), ("dv_w", i32 * 3),
                ("K", vp), ("c2w", vp), ("w2c_ref", vp), ("K_ref", vp), ("near_far_ref", vp), ("depth_values", vp * 3), ("t_rand", vp),
                ("rays_o", vp), ("rays_d", vp), ("near_far", vp), ("z", vp), ("pts", vp), ("ndc1", vp), ("ndc2", vp), ("ndc3", vp), ("ndc", vp)]


class EmbedParams(C.Structure):
    _fields_ = [("m", i32), ("n_freqs", i32), ("layout", i32), ("x", vp), ("out", vp)]


class FeatGatherParams(C.Structure):
    _fields_ = [("m", i32), ("V", i32), ("H", i32), ("W", i32), ("vol_d", i32 * 3), ("vol_h", i32 * 3),
                ("vol_w", i32 * 3), ("out_tiled", i32), ("unit_mask", i32), ("pts", vp), ("ndc1", vp), ("ndc2", vp), ("ndc3", vp),
                ("vol", vp * 3), ("conf", vp), ("imgs", vp), ("img_feat", vp), ("w2cs", vp), ("intrinsics", vp),
                ("feats", vp), ("u_out", vp)]


class ClSources(C.Structure):
    """ucnerf_cl_sources (ABI v5): the gather sources channel-last, each in its own allocation."""
    _fields_ = [("vol", vp * 3), ("img_feat", vp), ("imgs", vp), ("rgb_stride", i32), ("bf16", i32)]


class ClGrads(C.Structure):
    _fields_ = [("vol", vp * 3), ("img_feat", vp)]


class FeatGatherBwdParams(C.Structure):
    _fields_ = [("fwd", FeatGatherParams), ("g_feats", vp), ("g_vol", vp * 3), ("g_conf", vp), ("g_img_feat", vp),
                ("scratch", vp), ("g_cl", ClGrads)]


class MlpConfig(C.Structure):
    _fields_ = [("n_src", i32), ("pe_layout", i32), ("precision", i32), ("operand", i32)]


class MlpParams(C.Structure):
    _fields_ = [("cfg", MlpConfig), ("m", i32), ("S", i32), ("dirs_per_sample", i32), ("feats_tiled", i32),
                ("max_blocks", i32), ("encoded", i32), ("pts_stride", i32), ("dirs_stride", i32), ("feat_stride", i32),
                ("pts", vp), ("dirs", vp), ("feats", vp), ("wstream", vp), ("raw", vp)]


class MlpBwdParams(C.Structure):
    _fields_ = [("fwd", MlpParams), ("g_raw", vp), ("flat_params", vp), ("g_feats", vp), ("g_feat_stride", i32),
                ("g_flat", vp), ("workspace", vp), ("saved_valid", i32), ("bwd_mode", i32)]


class CompositeParams(C.Structure):
    _fields_ = [("n", i32), ("S", i32), ("variant", i32), ("white_bkgd", i32), ("raw", vp), ("z", vp), ("rays_d", vp),
                ("noise", vp), ("rgb_map", vp), ("depth_map", vp), ("acc_map", vp), ("disp_map", vp), ("weights", vp),
                ("var", vp), ("u", vp), ("wu", vp)]


class CompositeBwdParams(C.Structure):
    _fields_ = [("fwd", CompositeParams), ("g_rgb", vp), ("g_depth", vp), ("g_acc", vp), ("g_weights", vp),
                ("g_raw", vp)]


class SamplePdfParams(C.Structure):
    _fields_ = [("n", i32), ("n_bins", i32), ("n_samples", i32), ("u_stride", i32), ("n_merge", i32), ("from_coarse", i32),
                ("bins", vp),
                ("weights", vp), ("u", vp), ("z_merge", vp), ("samples", vp), ("inds", vp), ("cdf", vp),
                ("z_sorted", vp), ("merge_rank", vp)]


class MergeRowsParams(C.Structure):
    _fields_ = [("n", i32), ("na", i32), ("nb", i32), ("width", i32), ("a", vp), ("b", vp), ("rank", vp), ("out", vp)]


class CostVolumeParams(C.Structure):
    _fields_ = [("V", i32), ("C", i32), ("H", i32), ("W", i32), ("D", i32), ("pad", i32), ("feats", vp), ("proj", vp),
                ("depth_values", vp), ("variance", vp), ("count", vp)]


class DepthRegressParams(C.Structure):
    _fields_ = [("D", i32), ("Hp", i32), ("Wp", i32), ("pad", i32), ("prob_pre", vp), ("prob_init", vp), ("depth_values", vp),
                ("prob_volume", vp), ("depth", vp), ("confidence", vp)]


class CostVolumeBwdParams(C.Structure):
    _fields_ = [("fwd", CostVolumeParams), ("g_variance", vp), ("g_feats", vp)]


class DepthRegressBwdParams(C.Structure):
    _fields_ = [("fwd", DepthRegressParams), ("g_depth", vp), ("g_confidence", vp), ("g_prob_pre", vp)]


class RenderParams(C.Structure):
    _fields_ = [("n", i32), ("S", i32), ("white_bkgd", i32), ("max_blocks", i32), ("cfg", MlpConfig), ("rays_o", vp),
                ("rays_d", vp), ("z", vp), ("w2c_ref", f32 * 12), ("K_ref", f32 * 9), ("w2c_dir", f32 * 12),
                ("near", f32), ("far", f32), ("near_far", vp), ("H", i32), ("W", i32), ("vol_d", i32 * 3),
                ("vol_h", i32 * 3), ("vol_w", i32 * 3), ("vol", vp * 3), ("conf", vp), ("imgs", vp), ("img_feat", vp),
                ("w2cs", vp), ("intrinsics", vp), ("wstream", vp), ("cl", ClSources), ("workspace", vp), ("rgb_map", vp),
                ("depth_map", vp), ("acc_map", vp), ("weights", vp), ("var", vp), ("raw", vp), ("feats", vp),
                ("ev_mlp_start", vp), ("ev_mlp_stop", vp), ("train_workspace", vp), ("dir_feat", vp), ("u_sampled", vp),
                ("wu_map", vp), ("pts_in", vp), ("ndc1_in", vp), ("ndc2_in", vp), ("ndc3_in", vp), ("ndc_in", vp), ("feats_tiled", i32), ("train_bwd_mode", i32),
                ("resample", vp), ("w2c_dir_dev", vp), ("gen_rays", vp), ("gen_depths", vp)]


class RenderBwdParams(C.Structure):
    _fields_ = [("fwd", RenderParams), ("g_rgb", vp), ("g_depth", vp), ("flat_params", vp), ("g_flat", vp),
                ("g_vol", vp * 3), ("g_conf", vp), ("g_img_feat", vp), ("workspace", vp), ("gather_scratch", vp),
                ("saved_valid", i32), ("bwd_mode", i32), ("g_cl", ClGrads)]


STRUCTS = {
    "ucnerf_ray_gen_params": RayGenParams, "ucnerf_ndc_rays_params": NdcRaysParams,
    "ucnerf_dir_feature_params": DirFeatureParams, "ucnerf_sample_stratified_params": SampleStratifiedParams,
    "ucnerf_sample_cascade_params": SampleCascadeParams, "ucnerf_ndc_project_params": NdcProjectParams,
    "ucnerf_embed_params": EmbedParams, "ucnerf_feat_gather_params": FeatGatherParams,
    "ucnerf_feat_gather_bwd_params": FeatGatherBwdParams, "ucnerf_mlp_config": MlpConfig,
    "ucnerf_mlp_params": MlpParams, "ucnerf_mlp_bwd_params": MlpBwdParams, "ucnerf_composite_params": CompositeParams,
    "ucnerf_composite_bwd_params": CompositeBwdParams, "ucnerf_sample_pdf_params": SamplePdfParams,
    "ucnerf_render_params": RenderParams, "ucnerf_render_bwd_params": RenderBwdParams,
    "ucnerf_merge_rows_params": MergeRowsParams,
    "ucnerf_cost_volume_params": CostVolumeParams, "ucnerf_depth_regress_params": DepthRegressParams,
    "ucnerf_cost_volume_bwd_params": CostVolumeBwdParams, "ucnerf_depth_regress_bwd_params": DepthRegressBwdParams,
    "ucnerf_cl_sources": ClSources, "ucnerf_cl_grads": ClGrads, "ucnerf_build_rays_test_params": BuildRaysTestParams,
}

# every symbol include/ucnerf_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ucnerf_last_error": (C.c_char_p, []),
    "ucnerf_abi_version": (C.c_int, []),
    "ucnerf_sizeof": (C.c_int, [C.c_char_p]),
    "ucnerf_device_cus": (C.c_int, []),
    "ucnerf_build_flags": (C.c_char_p, []),
    "ucnerf_fused_tail_launches": (C.c_int64, []),
    "ucnerf_set_fused_tail": (C.c_int32, [C.c_int32]),
    "ucnerf_fused_tail_fits": (C.c_int32, [C.c_int32, C.c_int32]),
    "ucnerf_fused_tail_fits_resample": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "ucnerf_source_hash": (C.c_char_p, []),
    "ucnerf_event_create": (C.c_void_p, []),
    "ucnerf_event_record": (C.c_int, [_P, _P]),
    "ucnerf_event_elapsed_ms": (C.c_int, [_P, _P, _P]),
    "ucnerf_event_destroy": (C.c_int, [_P]),
    "ucnerf_ray_gen": (C.c_int, [_P, _P]),
    "ucnerf_ndc_rays": (C.c_int, [_P, _P]),
    "ucnerf_dir_feature": (C.c_int, [_P, _P]),
    "ucnerf_sample_stratified": (C.c_int, [_P, _P]),
    "ucnerf_ray_gen_sample": (C.c_int, [_P, _P, _P]),
    "ucnerf_sample_cascade": (C.c_int, [_P, _P]),
    "ucnerf_ndc_project": (C.c_int, [_P, _P]),
    "ucnerf_build_rays_test": (C.c_int, [_P, _P]),
    "ucnerf_embed": (C.c_int, [_P, _P]),
    "ucnerf_feat_gather_fwd": (C.c_int, [_P, _P]),
    "ucnerf_feat_gather_bwd": (C.c_int, [_P, _P]),
    "ucnerf_feat_gather_bwd_scratch_floats": (C.c_int64, [_P]),
    "ucnerf_mlp_param_count": (C.c_int64, [_P]),
    "ucnerf_mlp_stream_count": (C.c_int64, [_P]),
    "ucnerf_mlp_index_count": (C.c_int64, [_P]),
    "ucnerf_mlp_pack_index": (C.c_int, [_P, _P]),
    "ucnerf_mlp_pack": (C.c_int, [_P, _P, _P, _P, _P]),
    "ucnerf_mlp_pack_tensors": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "ucnerf_mlp_unpack_grad": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "ucnerf_mlp_fwd": (C.c_int, [_P, _P]),
    "ucnerf_mlp_bwd_workspace_floats": (C.c_int64, [_P, C.c_int32]),
    "ucnerf_mlp_bwd": (C.c_int, [_P, _P]),
    "ucnerf_mlp_fwd_train": (C.c_int, [_P, _P, C.c_int32, _P]),
    "ucnerf_composite_fwd": (C.c_int, [_P, _P]),
    "ucnerf_composite_bwd": (C.c_int, [_P, _P]),
    "ucnerf_sample_pdf": (C.c_int, [_P, _P]),
    "ucnerf_composite_sample_pdf": (C.c_int, [_P, _P, _P]),
    "ucnerf_merge_rows": (C.c_int, [_P, _P]),
    "ucnerf_cost_volume": (C.c_int, [_P, _P]),
    "ucnerf_depth_regress": (C.c_int, [_P, _P]),
    "ucnerf_cost_volume_bwd": (C.c_int, [_P, _P]),
    "ucnerf_depth_regress_bwd": (C.c_int, [_P, _P]),
    "ucnerf_render_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "ucnerf_render_fused_fwd": (C.c_int, [_P, _P]),
    "ucnerf_gather_repack_floats": (C.c_int64, [_P]),
    "ucnerf_gather_repack": (C.c_int, [_P, _P, _P, _P]),
    "ucnerf_render_bwd_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "ucnerf_render_fused_bwd": (C.c_int, [_P, _P]),
}

_lib = None


def lib():
    """Loads the library once; raises if it is absent or its ABI does not match this binding."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64 (same SONAME as the system one).  Device pointers and streams
    # handed over by torch are only meaningful inside THAT runtime instance, so it has to be the one our
    # library binds to: make sure it is resident before dlopen() resolves libucnerf_hip.so's dependency.
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("uc_nerf_amd: %s is missing -- build it with `python -m uc_nerf_amd.build` "
                           "(there is no CPU or PyTorch fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise RuntimeError("uc_nerf_amd: %s does not export %s" % (LIB_PATH, name)) from e
        fn.restype, fn.argtypes = res, args
    if L.ucnerf_abi_version() != ABI_VERSION:
        raise RuntimeError("uc_nerf_amd: ABI version mismatch")
    for cname, cls in STRUCTS.items():
        got = L.ucnerf_sizeof(cname.encode())
        if got != C.sizeof(cls):
            raise RuntimeError("uc_nerf_amd: struct %s is %d bytes in the library, %d in the binding"
                               % (cname, got, C.sizeof(cls)))
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError("uc_nerf_amd: %s failed (%d): %s" % (what, rc, lib().ucnerf_last_error().decode()))


def call(name, params, stream):
    """Invokes an `int f(const params*, stream)` entry point."""
    check(getattr(lib(), name)(C.addressof(params), C.c_void_p(stream)), name)
