/*
 * ucnerf_hip.h -- C ABI of libucnerf_hip.so: the MI355X (gfx950) implementation of UC-NeRF's
 * ray-marching volume-render hot path.
 *
 * The reference (wrld/UC-NeRF) is pure Python/PyTorch and has no FFI of its own; each entry point below
 * replaces one torch-op sequence of the reference's hot path (file:line cited per function, paths relative
 * to the reference tree).  A binding for any host language only needs this header: plain pointers and
 * sizes, no torch types.  The Python host side in uc_nerf_amd/ binds it with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the field name ends in _host;
 *   - all arrays are dense, row-major, float32 unless stated; index outputs are int64 (as torch's);
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing,
 *     is re-entrant, and returns 0 on success or a negative UCNERF_E* code; ucnerf_last_error() returns
 *     a thread-local description of the last failure;
 *   - arguments are checked before anything is launched: a NULL params pointer, a NULL required array, a size outside the documented range and a
 *     NEGATIVE count are UCNERF_EINVAL; a count of ZERO rays / samples / rows is an empty batch -- success, nothing launched, no pointer read
 *     (torch hands empty tensors to the reference's functions the same way);
 *   - "sample" = one depth sample on one ray; M = N_rays * S samples, ray-major (sample s of ray r at
 *     index r*S + s).
 */
#ifndef UCNERF_HIP_H
#define UCNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCNERF_ABI_VERSION 6

#define UCNERF_OK 0
#define UCNERF_EINVAL (-1)   /* bad argument (null pointer, unsupported size/config) */
#define UCNERF_EHIP (-2)     /* HIP runtime error at launch */

const char* ucnerf_last_error(void);
int ucnerf_abi_version(void);
/* sizeof() of a params struct by name ("ucnerf_mlp_params", ...), -1 if unknown: lets a binding check its mirror. */
int ucnerf_sizeof(const char* struct_name);
/* Number of compute units of the current device (grid sizing for persistent kernels); <0 on error. */
int ucnerf_device_cus(void);
/* The compile-time parameters this binary was built with, as "NAME=value NAME=value ..." (structural parameters of the kernels:
 * UCNERF_BF16_BW, UCNERF_BF16_NBUF, UCNERF_MLP_WAVES, UCNERF_CHAIN_WAVES, ...; the wrong-result timing switches of rounds 1-3 were taken out of
 * the sources in round 4).  The host side checks the production values (tests/test_abi_host.py). */
const char* ucnerf_build_flags(void);
/* Diagnostics: how many ucnerf_render_fused_fwd calls of this process let the gather-fused launch composite (and re-sample) its rays in its own
 * tail instead of launching K7 / K8 / K9 behind it -- passes of at most three rounds of 32-sample tiles, derived or given coordinates, any number of
 * samples per ray up to 256 (csrc/render.hip: tail_fits; the
 * outputs are bit-identical either way).  The environment variable UCNERF_FUSED_TAIL=0 (read once, at the first use) starts the process with
 * that route off; ucnerf_set_fused_tail(0 / 1) switches it at run time and returns the previous setting (thread-safe). */
int64_t ucnerf_fused_tail_launches(void);
int32_t ucnerf_set_fused_tail(int32_t on);
/* 1 when a pass of n rays x S samples is of the size that takes that route on the current device (the other conditions are the caller's to know:
 * gather-fused precision, fp32 channel-last sources, no max_blocks).  A host uses it to decide whether to hand the pass its ray
 * generation too (gen_rays / gen_depths): on the tail route the blocks generate their own rays at no measurable cost. */
int32_t ucnerf_fused_tail_fits(int32_t n, int32_t S);
/* The same for a COARSE pass that also draws the next pass's n_samples depths (ucnerf_render_params.resample): the library's own decision for
 * that pass (size AND the re-sampling limits: S - 1 <= 128 bins, S + n_samples <= 512), so that a host that folds the ray generation into the
 * pass exactly when it takes the tail route decides as the library does (round 4's advisor finding). */
int32_t ucnerf_fused_tail_fits_resample(int32_t n, int32_t S, int32_t n_samples);
/* Digest of the sources, headers and flags this binary was linked from (uc_nerf_amd/build.py: source_hash()): a host can tell a library
 * that does not belong to the tree it sits in (tests/test_abi_host.py), and __graft_entry__.build() rebuilds one that was not linked on
 * the machine it runs on. */
const char* ucnerf_source_hash(void);
/* HIP timing events for measuring kernels inside a call chain (bench harness): create / record on a stream /
 * elapsed milliseconds between two recorded events (waits for `stop`) / destroy. */
void* ucnerf_event_create(void);
int ucnerf_event_record(void* event, void* stream);
int ucnerf_event_elapsed_ms(void* start, void* stop, float* ms_host);
int ucnerf_event_destroy(void* event);

/* ------------------------------------------------------------------------------------------------
 * a1  ray generation -- data/ray_utils.py:12-53, utils/utils.py:217-271 (get_rays_mvs, deterministic and
 *     given-pixel branches), utils/run_nerf_helpers.py:248-257 (OpenGL variant).
 *     d_cam = ((x-K02)/K00, (y-K12)/K11, 1)  [opengl: ((x-W/2)/f, -(y-H/2)/f, -1)],  rays_d = d_cam @ R^T.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n;            /* rays to generate */
    int32_t H, W;         /* image size (grid mode / opengl) */
    int32_t grid_start;   /* grid mode (xs == NULL): first flattened row-major pixel index */
    int32_t opengl;       /* 0: intrinsic-matrix +z convention, 1: single-focal -z convention */
    float K[9];           /* 3x3 intrinsics, row-major (opengl: K[0] = focal) */
    float c2w[12];        /* 3x4 camera-to-world, row-major */
    const float* xs;      /* [n] pixel columns, or NULL for grid mode */
    const float* ys;      /* [n] pixel rows */
    float* rays_d;        /* [n,3] out */
    float* rays_o;        /* [n,3] out or NULL (origin is c2w[:,3] for every ray) */
    float* pix;           /* [2,n] out (row, col) or NULL */
    float w2c_dir[12];    /* with `angle`: rotation of the view-direction feature (see ucnerf_dir_feature) */
    float* angle;         /* [n,3] out or NULL: (d/|d|) @ R_dir^T of the generated rays, same values as ucnerf_dir_feature
                             on rays_d with has_ref = 1 -- saves that launch in the render passes (ucnerf_render_params.dir_feat) */
} ucnerf_ray_gen_params;
int ucnerf_ray_gen(const ucnerf_ray_gen_params* p, void* stream);

/* a2  LLFF NDC warp of rays -- data/ray_utils.py:56-94 (variant 0: focal[2], d2 = 1-o2),
 *     utils/run_nerf_helpers.py:277-294 (variant 1: scalar focal, d2 = -2 near/o_z). */
typedef struct {
    int32_t n, H, W, variant;
    float focal_x, focal_y, near;
    const float* rays_o;  /* [n,3] */
    const float* rays_d;  /* [n,3] */
    float* out_o;         /* [n,3] */
    float* out_d;         /* [n,3] */
} ucnerf_ndc_rays_params;
int ucnerf_ndc_rays(const ucnerf_ndc_rays_params* p, void* stream);

/* view-direction feature of rendering() -- network/renderer.py:232-238,163-174:
 *   cos = |d|,  angle = (d/cos) @ R_ref^T   (R_ref == NULL: angle = d/cos). */
typedef struct {
    int32_t n;
    int32_t has_ref;
    int32_t repeat;       /* 0/1: angle is [n,3]; S > 1: every ray's row is written S times -> [n*S,3] (per-sample form) */
    float w2c_ref[12];
    const float* rays_d;  /* [n,3] */
    float* angle;         /* [n,3] (or [n*repeat,3]) out */
    float* cos_angle;     /* [n] out or NULL */
    const float* w2c_ref_dev; /* optional DEVICE pointer to the same matrix, row-major with 4 columns (a 3x4 or 4x4 tensor): used
                                 instead of w2c_ref when non-NULL (has_ref is then implied), so that a caller holding the pose
                                 on the device -- rendering() gets pose_ref['w2cs'] as a device tensor -- need not read it back */
} ucnerf_dir_feature_params;
int ucnerf_dir_feature(const ucnerf_dir_feature_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a3  depth sampling -- data/ray_utils.py:152-197 (ray_marcher) and the live cascade sampler
 *     utils/utils.py:393-397,684-706 (three uniform sets, sorted, stratified).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, S;
    int32_t lindisp;
    float perturb;        /* > 0: z = lower + (upper-lower) * perturb * noise */
    float near, far;      /* used when rays == NULL: every ray samples [near, far] (pts must be NULL) */
    const float* rays;    /* [n,8] = (o, d, near, far), or NULL */
    const float* noise;   /* [n,S] uniform [0,1) draws, required when perturb > 0 */
    float* z;             /* [n,S] out */
    float* pts;           /* [n,S,3] out or NULL */
} ucnerf_sample_stratified_params;
int ucnerf_sample_stratified(const ucnerf_sample_stratified_params* p, void* stream);

/* a1 + a3 in one launch: the rays of ucnerf_ray_gen and their stratified depths (scalar near / far form: rays = pts = NULL) --
 * data/ray_utils.py:32-53 followed by :176-194, as ray_marcher chains them.  Same results as the two calls. */
int ucnerf_ray_gen_sample(const ucnerf_ray_gen_params* rays, const ucnerf_sample_stratified_params* depths, void* stream);

typedef struct {
    int32_t n, S;         /* S must be a multiple of 3, S <= 768 */
    const float* near_far;/* [n,6] = (near_1, far_1, near_2, far_2, near_3, far_3) per ray */
    const float* t_rand;  /* [n,S] uniform draws, or NULL for no jitter */
    const float* rays_o;  /* [3] shared origin, used when pts != NULL */
    const float* rays_d;  /* [n,3] */
    float* z;             /* [n,S] out (sorted then jittered) */
    float* pts;           /* [n,S,3] out or NULL */
} ucnerf_sample_cascade_params;
int ucnerf_sample_cascade(const ucnerf_sample_cascade_params* p, void* stream);

/* a1 + a3 + a4 of the EVALUATION loop in one launch (ABI v5) -- utils/utils.py:600-739 (build_rays_test), called once per 1024-pixel chunk by
 * train.py:251-256: rays through pixels grid_start .. grid_start + n - 1 of the flattened HxW grid (get_rays_mvs, isRandom=False), the three
 * per-ray cascade ranges read from the stages' depth hypotheses at (row, col) // (4, 2, 1) (first and last plane), 3 x S/3 uniform depths
 * sorted and stratified-jittered, the world points and their four normalised copies (get_ndc_coordinate).  Same values as ucnerf_ray_gen ->
 * ucnerf_sample_cascade -> ucnerf_ndc_project, bit for bit.  The camera matrices and the scene range are read from DEVICE memory -- the caller
 * holds them as device tensors, nothing is read back to the host per chunk. */
typedef struct {
    int32_t n, S;              /* rays of the chunk; samples per ray (multiple of 3, <= 768) */
    int32_t H, W, grid_start;
    int32_t dv_d[3], dv_h[3], dv_w[3];   /* sizes of depth_values[k] */
    const float* K;            /* [3,3] intrinsics of the rendered view (device) */
    const float* c2w;          /* [>=3,4] camera-to-world of the rendered view (device, rows 4 floats apart) */
    const float* w2c_ref;      /* [>=3,4] reference view (device) */
    const float* K_ref;        /* [3,3] (device) */
    const float* near_far_ref; /* [2] scene range (near, far) (device) */
    const float* depth_values[3];  /* [D_k,h_k,w_k] hypotheses of cascade stage k + 1 (device) */
    const float* t_rand;       /* [n,S] uniform draws, or NULL for no jitter */
    float* rays_o;             /* [3] out or NULL: c2w[:3,3] */
    float* rays_d;             /* [n,3] out */
    float* near_far;           /* [n,6] out or NULL: (near_1, far_1, near_2, far_2, near_3, far_3) per ray */
    float* z;                  /* [n,S] out */
    float* pts;                /* [n,S,3] out */
    float* ndc1;               /* [n,S,3] out: stage copies and the scene-normalised copy */
    float* ndc2;
    float* ndc3;
    float* ndc;
} ucnerf_build_rays_test_params;
int ucnerf_build_rays_test(const ucnerf_build_rays_test_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a4  world -> reference-view normalised coordinates -- utils/utils.py:323-373 (get_ndc_coordinate).
 *     p_cam = p R^T + T; |z|<1e-4 -> 1e-4; q = p_cam K^T; xy = q.xy/q.z / inv_scale;
 *     z_k = (q.z - near_k)/(far_k - near_k) for k in {stage1, stage2, stage3, ndc}.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t m;            /* points */
    int32_t has_w2c;
    int32_t sample_2d;    /* 1: write only out_ndc = (x, y, raw z) */
    int32_t nf_stride;    /* elements between consecutive points in near/far arrays: 1 (per point) or 0 (broadcast) */
    float w2c[12];
    float K[9];
    float inv_scale[2];   /* (W-1, H-1) */
    float near, far;      /* scene range for the 'ndc' copy */
    const float* pts;     /* [m,3] */
    const float* near_1; const float* far_1;   /* [m] (or [1] when nf_stride == 0) */
    const float* near_2; const float* far_2;
    const float* near_3; const float* far_3;
    float* out_stage1;    /* [m,3] */
    float* out_stage2;
    float* out_stage3;
    float* out_ndc;
} ucnerf_ndc_project_params;
int ucnerf_ndc_project(const ucnerf_ndc_project_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a5  positional encoding -- network/models.py:20-71 (layout 0, "live": [x | sin(2^k x) k<L | cos ...]),
 *     utils/run_nerf_helpers.py:23-71 (layout 1, interleaved [x | sin f0 | cos f0 | sin f1 ...]).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t m;            /* 3-vectors */
    int32_t n_freqs;
    int32_t layout;
    const float* x;       /* [m,3] */
    float* out;           /* [m, 3 + 6*n_freqs] */
} ucnerf_embed_params;
int ucnerf_embed(const ucnerf_embed_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ABI v5: the gather sources in the layout the fast kernels read ("channel-last": one voxel / pixel = contiguous channels), EACH IN ITS OWN
 * ALLOCATION -- what a producer's convolutions write when they run in torch's channels_last / channels_last_3d memory formats
 * (network/mvs_models.py:624-646: cost_regularization -> volume_feature_no_ref per stage, FeatureNet -> img_feats; new tensors every training
 * step, train.py:136-163), read IN PLACE.  ucnerf_gather_repack builds the same layouts from the reference's channel-major tensors for the
 * sources that are not handed over this way.
 *   vol[k]    [D,h,w,8]           = the memory of a [1,8,D,h,w] tensor in channels_last_3d            16-byte aligned
 *   img_feat  [V,H,W,8]           = a [V,8,H,W] tensor in channels_last                                16-byte aligned
 *   imgs      [V,H,W,rgb_stride]  rgb_stride = 3 (a [V,3,H,W] tensor in channels_last; 4-byte aligned) or 4 (r,g,b,pad: what the repack writes)
 * bf16 = 1: every array holds bf16 instead of fp32 (rgb_stride must be 4; SURVEY.md 8 configs[4] "bf16 features").
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const void* vol[3];
    const void* img_feat;
    const void* imgs;
    int32_t rgb_stride;
    int32_t bf16;
} ucnerf_cl_sources;
/* Gradients of the differentiable ones among them, in the same layouts (fp32, 16-byte aligned, ACCUMULATED into: zero them first); any may be NULL */
typedef struct {
    float* vol[3];      /* [D,h,w,8] */
    float* img_feat;    /* [V,H,W,8] */
} ucnerf_cl_grads;

/* ------------------------------------------------------------------------------------------------
 * a7  feature gather -- network/renderer.py:177-212 (gen_pts_feats) = utils/utils.py:833-893
 *     (index_point_feature: 3 trilinear volume lookups + bilinear confidence, align_corners=False, border)
 *     + utils/utils.py:742-799 (build_color_volume: per source view bilinear rgb + in-mask + 8-ch image
 *     features, align_corners=True, border).  Feature vector per sample, F = 24 + 12*V + 1:
 *       [0:8] stage1 | [8:16] stage2 | [16:24] stage3 | [24:24+4V] (r,g,b,mask) x V | [..+8V] img_feat x V | conf
 *     Sources are read in the reference's own layouts.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t m;                 /* samples */
    int32_t V;                 /* source views (reference view_num - 1), 1..8 */
    int32_t H, W;              /* image / confidence / img_feat size */
    int32_t vol_d[3], vol_h[3], vol_w[3];   /* cascade volume sizes, 8 channels each */
    int32_t out_tiled;         /* 0: feats [m,F] row-major; 1: [ceil(m/32)][F][32] (MLP tile layout) */
    int32_t unit_mask;         /* 0 = everything; else bit k selects unit k: 0..2 cascade volumes, 3 confidence,
                                  4+i source view i (rgb+mask and image features).  Sources of unselected units may be NULL. */
    const float* pts;          /* [m,3] world points */
    const float* ndc1;         /* [m,3] stage coordinates in ~[0,1] (ucnerf_ndc_project outputs) */
    const float* ndc2;
    const float* ndc3;
    const float* vol[3];       /* [8,D,h,w] */
    const float* conf;         /* [H,W] */
    const float* imgs;         /* [V,3,H,W] */
    const float* img_feat;     /* [V,8,H,W] */
    const float* w2cs;         /* [V,12] rows 0..2 of each 4x4 */
    const float* intrinsics;   /* [V,9] */
    float* feats;              /* out */
    float* u_out;              /* optional [m] out: the per-sample uncertainty u = 1 - (sampled confidence) that the MLP
                                  blends its two heads with (network/models.py:149); written by the confidence unit */
} ucnerf_feat_gather_params;
int ucnerf_feat_gather_fwd(const ucnerf_feat_gather_params* p, void* stream);

/* Backward of the gather w.r.t. its differentiable sources (the reference's autograd reaches the three
 * volumes, img_feat and confidence through grid_sample; positions and images get no gradient,
 * SURVEY.md 3.2).  Gradients are ACCUMULATED (atomic add) into the g_* buffers: zero them first. */
typedef struct {
    ucnerf_feat_gather_params fwd;   /* same geometry/inputs as the forward call (feats unused) */
    const float* g_feats;            /* [m,F] row-major upstream gradient */
    float* g_vol[3];                 /* [8,D,h,w] or NULL to skip */
    float* g_conf;                   /* [H,W] or NULL */
    float* g_img_feat;               /* [V,8,H,W] or NULL */
    float* scratch;                  /* optional, ucnerf_feat_gather_bwd_scratch_floats() floats, 16-byte aligned: the
                                        volume / image-feature gradients are then accumulated channel-LAST in it (eight
                                        lanes add the 32 contiguous bytes of one corner instead of eight scattered
                                        cache lines) and added to g_vol / g_img_feat by a transposing pass.  Contents
                                        are overwritten.  NULL: atomics go straight to the channel-major outputs. */
    ucnerf_cl_grads g_cl;            /* ABI v5, per source: a gradient array in the CHANNEL-LAST layout of ucnerf_cl_sources.  A source whose entry is
                                        set has its gradient ACCUMULATED there (zero it first) and its g_vol[k] / g_img_feat entry is ignored: no
                                        scratch, no memset, no transposing pass for it -- for callers whose sources live channel-last
                                        (ucnerf_render_params.cl handed over in place).  Entries left NULL take the route above. */
} ucnerf_feat_gather_bwd_params;
int ucnerf_feat_gather_bwd(const ucnerf_feat_gather_bwd_params* p, void* stream);
int64_t ucnerf_feat_gather_bwd_scratch_floats(const ucnerf_feat_gather_params* p);

/* ------------------------------------------------------------------------------------------------
 * a6 (+a5 fused)  the uncertainty-conditioned MLP -- network/models.py:138-184 driven by
 *     network/renderer.py:78-106 (run_network_mvs; batchify's netchunk loop is unnecessary here).
 *     Fixed architecture: D=6, W=128, skip after layer 4, multires 10 / 4 (63 + F + 27 inputs).
 *     Weights are consumed as a pre-packed MFMA operand stream built by ucnerf_mlp_pack from the flat
 *     parameter vector (concatenation of the reference state_dict tensors in state_dict order).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_src;        /* V = view_num - 1 */
    int32_t pe_layout;    /* 0 live (network/models.py), 1 interleaved (run_nerf_helpers.py) */
    int32_t precision;    /* 0: f32 -- exact fp32 MFMA (v_mfma_f32_32x32x2_f32); forward, backward, every input mode.
                             1: bf16x3 -- every product as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the bf16 matrix cores
                                (v_mfma_f32_32x32x16_bf16, fp32 accumulate): inference forward only; rendered outputs
                                stay within 1e-5 of fp64 (the parity bar is 1e-4), ~2.9x the f32 throughput.
                             2: bf16 -- the hi*hi term only (plain bf16 operands, fp32 accumulate; same packed stream as 1):
                                inference forward only, NOT within the 1e-4 parity bar (rendered error ~3e-3, > 50 dB PSNR
                                against the f32 render); offered because the reference configuration names bf16.
                             3: bf16x3 with the feature gather INSIDE the MLP kernel: the stream is packed in the
                                order that kernel's lane halves produce the bias nets' operands and serves ucnerf_render_fused_fwd
                                only (channel-last sources, coordinates derived from (ray, depth) or handed over -- pts_in ... ndc_in --,
                                no kept features, no per-sample uncertainty); ucnerf_mlp_fwd refuses it.  Same arithmetic as 1. */
    int32_t operand;      /* (ABI v6) the 16-bit terms of precisions 1 .. 3 -- the packed stream and the kernels that read it must agree:
                             0: bf16 (8 significant bits per term; any float32 range).  The default.
                             1: fp16 (v_mfma_f32_32x32x16_f16: 11 significant bits per term at the same matrix-core rate, so the three-product
                                split holds ~22 bits -- rendered outputs at float32 level where the bf16 split is at 16 .. 17 bits).  fp16's RANGE
                                is the price: the conversions round toward zero, so an activation beyond 131 008 is clamped -- finite, but wrong
                                (float32 itself resolves no more than 1e-2 of it) -- and a term below 6e-5 is held with an absolute resolution of 3e-8.  Inference forward
                                only (ucnerf_mlp_fwd_train refuses it).  Ignored by precision 0. */
} ucnerf_mlp_config;

/* Sizes: floats in the flat parameter vector, floats (4-byte units) of the packed stream, int32 entries of the pack
 * index (<0 on bad config).  Stream and index depend on cfg.precision. */
int64_t ucnerf_mlp_param_count(const ucnerf_mlp_config* cfg);
int64_t ucnerf_mlp_stream_count(const ucnerf_mlp_config* cfg);
int64_t ucnerf_mlp_index_count(const ucnerf_mlp_config* cfg);
/* Host-side: fills idx_host[index_count] with the flat-parameter index feeding each stream element (-1 = zero pad;
 * bf16x3: bit 30 selects the low part of the split). */
int ucnerf_mlp_pack_index(const ucnerf_mlp_config* cfg, int32_t* idx_host);
/* Device-side: builds the packed stream from the flat parameters; idx is the device copy of the table above. */
int ucnerf_mlp_pack(const ucnerf_mlp_config* cfg, const float* flat_params, const int32_t* idx, float* stream_out, void* stream);
/* The same from parameters that live in SEPARATE device tensors (a torch module's, in state_dict order: their concatenation is the flat
 * vector), without first concatenating them: tensor_ptrs_host / tensor_numel_host are HOST arrays of n_tensors (<= 48) device pointers
 * and element counts.  Lets a drop-in re-pack from the live parameters in every call -- in-place writes through `.data` (the reference's
 * own weights_init, network/models.py:15-17) leave no trace a cache could key on. */
int ucnerf_mlp_pack_tensors(const ucnerf_mlp_config* cfg, int32_t n_tensors, const void* const* tensor_ptrs_host, const int64_t* tensor_numel_host,
                            const int32_t* idx, float* stream_out, void* stream);
/* Transpose of the pack for gradients: g_flat[idx[i]] += g_stream[i] (g_flat zeroed by the caller). */
int ucnerf_mlp_unpack_grad(const float* g_stream, const int32_t* idx, float* g_flat, int64_t n, void* stream);

typedef struct {
    ucnerf_mlp_config cfg;
    int32_t m;                 /* samples */
    int32_t S;                 /* samples per ray (view dir of sample s is dirs[s / S]); dirs_per_sample: ignored */
    int32_t dirs_per_sample;   /* 1: dirs is [m,3] */
    int32_t feats_tiled;       /* feats layout, see ucnerf_feat_gather_params.out_tiled */
    int32_t max_blocks;        /* 0 = auto (persistent grid sized from the CU count) */
    int32_t encoded;           /* 1: pts / dirs rows already hold the 63 / 27 encoded values (cfg.pe_layout order), as in
                                  UCNeRF.forward(x) of the reference; implies dirs_per_sample, row-major feats */
    int32_t pts_stride;        /* floats between consecutive rows of pts / dirs / row-major feats; 0 = dense (3 or 63, */
    int32_t dirs_stride;       /*   3 or 27, F).  Lets all three point into one [m, 63+F+27] matrix. */
    int32_t feat_stride;
    const float* pts;          /* [m,3] the 'ndc' coordinates fed to the positional encoding */
    const float* dirs;         /* [m/S,3] or [m,3] view-direction feature */
    const float* feats;        /* [m,F] or tiled */
    const float* wstream;      /* packed weights (ucnerf_mlp_pack) */
    float* raw;                /* [m,4] out: rgb (after sigmoid), sigma (after relu) */
} ucnerf_mlp_params;
int ucnerf_mlp_fwd(const ucnerf_mlp_params* p, void* stream);

/* Backward (autograd of network/models.py:138-184): re-runs the forward keeping the per-layer activations (unless
 * saved_valid), then walks the layers backwards (bwd_mode).  Produces d(feats) and ACCUMULATES the parameter gradients
 * into g_flat, a vector laid out exactly like the flat parameter vector (zero it first).  The three parameter
 * sets the reference never uses (pts_bias_confidence_1, feature_linear_1, confi_linear) get no contribution.
 * Positions and view directions receive no gradient (the reference path is non-differentiable there). */
typedef struct {
    ucnerf_mlp_params fwd;     /* forward arguments (feats row-major [m,F], or tiled with bwd_mode 0); raw is not written */
    const float* g_raw;        /* [m,4] upstream gradient */
    const float* flat_params;  /* [param_count] the parameters the wstream was packed from */
    float* g_feats;            /* [m,F] out, rows g_feat_stride floats apart (every one of the F columns written) */
    int32_t g_feat_stride;     /* 0 = F */
    float* g_flat;             /* [param_count] accumulated */
    float* workspace;          /* scratch, ucnerf_mlp_bwd_workspace_floats() floats, 16-byte aligned (kept activations, gradient sets, and --
                                  bwd_mode 0 -- the weight-gradient launch's 16 chunk counters, which the call zeroes itself: one workspace per
                                  call in flight) */
    int32_t saved_valid;       /* 1: `workspace` already holds the activations of THIS forward, written by
                                  ucnerf_mlp_fwd_train with the same arguments -- the backward then skips its own forward */
    int32_t bwd_mode;          /* 0: register-resident gradient chain (ONE kernel walks the network backwards per 32-sample tile, data
                                  gradients as split-bf16 products on the matrix cores, fp32 accumulate: 2^-16 relative) + one
                                  weight-gradient GEMM per layer;  1: layer-by-layer exact-fp32 data-gradient GEMMs (round 1/2 path) */
} ucnerf_mlp_bwd_params;
int64_t ucnerf_mlp_bwd_workspace_floats(const ucnerf_mlp_config* cfg, int32_t m);
int ucnerf_mlp_bwd(const ucnerf_mlp_bwd_params* p, void* stream);
/* Training forward: ucnerf_mlp_fwd (f32 or bf16x3 precision, features row-major or tiled) that also keeps the per-layer activations in `bwd_workspace`
 * (ucnerf_mlp_bwd_workspace_floats floats), for a following ucnerf_mlp_bwd with saved_valid = 1 and THE SAME bwd_mode: for bwd_mode 0 the ten
 * [m,128] sets are kept as 24-bit floats (sign, exponent, 15 mantissa bits: the 16 significant bits the chain's and the weight-gradient launch's
 * split-bf16 operands carry -- three quarters of the bytes every kernel of the step moves), for bwd_mode 1 as fp32. */
int ucnerf_mlp_fwd_train(const ucnerf_mlp_params* p, float* bwd_workspace, int32_t bwd_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a9  alpha compositing -- network/renderer.py:25-36,109-140 (variant 0 "live": alpha = 1-exp(-sigma),
 *     rgb/sigma already activated, dists ignored) and utils/run_nerf_helpers.py:343-390 (variant 1:
 *     alpha = 1-exp(-relu(sigma+noise)*dists), rgb = sigmoid(raw), dists = dz*|d|, last = 1e10).
 *     T_i = prod_{j<i} (1-alpha_j+1e-10), w = alpha*T.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, S;
    int32_t variant;
    int32_t white_bkgd;
    const float* raw;      /* [n,S,4] */
    const float* z;        /* [n,S] */
    const float* rays_d;   /* [n,3]  (variant 1) */
    const float* noise;    /* [n,S] or NULL (variant 1) */
    float* rgb_map;        /* [n,3] */
    float* depth_map;      /* [n] */
    float* acc_map;        /* [n] or NULL */
    float* disp_map;       /* [n] or NULL */
    float* weights;        /* [n,S] or NULL */
    float* var;            /* [n] unbiased variance of the weights, or NULL (variant 0 only) */
    const float* u;        /* optional [n,S] per-sample uncertainty (ucnerf_feat_gather_params.u_out) ... */
    float* wu;             /* ... and [n] out: sum_i w_i u_i, the composited uncertainty of the ray (a build extra: the
                              reference keeps u per sample only, network/models.py:149,177-178) */
} ucnerf_composite_params;
int ucnerf_composite_fwd(const ucnerf_composite_params* p, void* stream);

typedef struct {
    ucnerf_composite_params fwd;   /* inputs as in the forward call; outputs unused */
    const float* g_rgb;        /* [n,3] or NULL */
    const float* g_depth;      /* [n] or NULL */
    const float* g_acc;        /* [n] or NULL */
    const float* g_weights;    /* [n,S] or NULL */
    float* g_raw;              /* [n,S,4] out */
} ucnerf_composite_bwd_params;
int ucnerf_composite_bwd(const ucnerf_composite_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a8  inverse-CDF sampling -- data/ray_utils.py:98-141 == utils/run_nerf_helpers.py:298-341, with the
 *     uniform draws u given, and the sorted merge of data/ray_utils.py:219.
 *     Reproduces torch-CPU's accumulation orders (vectorised row sum, float64 cumsum) so that
 *     inds == torch.searchsorted(cdf, u, right=True) bit for bit.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n;
    int32_t n_bins;        /* L = bins per ray; weights per ray = L-1; 2 <= L <= 1024 */
    int32_t n_samples;     /* M draws per ray, M <= 1024 */
    int32_t u_stride;      /* n_samples (per-ray draws) or 0 (one shared row, e.g. linspace) */
    int32_t n_merge;       /* z_merge entries per ray (0: no merge), n_merge + M <= 2048 */
    int32_t from_coarse;   /* 1: the hierarchical recipe of data/ray_utils.py:216-217 in one call -- z_merge holds the
                              coarse depths z[n,S], weights the coarse weights w[n,S]; bins = .5*(z[:-1]+z[1:]) (so
                              n_bins = S-1) and the pdf uses w[:, 1:-1].  bins may then be NULL. */
    const float* bins;     /* [n,L] */
    const float* weights;  /* [n,L-1]  (from_coarse: [n,S]) */
    const float* u;        /* [n,M] or [M] */
    const float* z_merge;  /* [n,n_merge] values to merge with the samples, or NULL */
    float* samples;        /* [n,M] out or NULL */
    int64_t* inds;         /* [n,M] out or NULL */
    float* cdf;            /* [n,L] out or NULL */
    float* z_sorted;       /* [n,n_merge+M] out: sort(cat(samples, z_merge)), or NULL */
    int32_t* merge_rank;   /* [n,M+n_merge] out or NULL (needs z_sorted): position in z_sorted of element i of
                              cat(samples, z_merge) -- the permutation of that sort, for ucnerf_merge_rows */
} ucnerf_sample_pdf_params;
int ucnerf_sample_pdf(const ucnerf_sample_pdf_params* p, void* stream);

/* a9 of the coarse pass + a8 in ONE launch -- network/renderer.py:109-140 followed by data/ray_utils.py:216-219, as the hierarchical renderer
 * chains them (data/ray_utils.py:199-224): the wave that composites a ray re-samples it from the weights it has just computed.  `c` is the
 * compositing call (live variant, no uncertainty inputs), `s` the re-sampling in its from_coarse form with n == c->n, n_merge == c->S,
 * n_bins == c->S - 1; s->weights is ignored (the composite's weights, also written to c->weights when that is non-NULL) and s->z_merge, when
 * given, must equal c->z.  Results are those of ucnerf_composite_fwd(c) followed by ucnerf_sample_pdf(s with weights = c->weights), bit for bit. */
int ucnerf_composite_sample_pdf(const ucnerf_composite_params* c, const ucnerf_sample_pdf_params* s, void* stream);

/* Applies the permutation of a sorted merge to per-sample rows: out[r][rank[r][i]] = cat(a[r], b[r])[i].
 * Lets the fine pass of the hierarchical renderer (data/ray_utils.py:199-224) evaluate the network on the NEW depths
 * only and take the coarse depths' outputs from the coarse pass: the merged rows are bit-identical to re-evaluating
 * all of them, because a sample's network output depends on nothing but that sample. */
typedef struct {
    int32_t n, na, nb, width;  /* rows per ray in a / b, floats per row (1..8) */
    const float* a;            /* [n,na,width] rows of the first na elements of the concatenation */
    const float* b;            /* [n,nb,width] */
    const int32_t* rank;       /* [n,na+nb] */
    float* out;                /* [n,na+nb,width] */
} ucnerf_merge_rows_params;
int ucnerf_merge_rows(const ucnerf_merge_rows_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * f2   the step in front of the path: cost-volume assembly and depth regression of one cascade stage.
 *      ucnerf_cost_volume replaces the loop of network/mvs_models.py:609-626: homo_warp (utils/utils.py:1105-1172,
 *      nearest-neighbour grid_sample, border padding, align_corners=True) of every source view's feature map onto the
 *      D x Hp x Wp hypotheses of the target view, the in-frustum count and the variance volume handed to the
 *      regularisation network.  (The warped-image volume of :614,617 is never used by the reference and is not built.)
 *      ucnerf_depth_regress replaces network/mvs_models.py:629-646: softmax over depth (+ optional initial logits),
 *      expected depth, 4-tap photometric confidence, cropped by `pad`.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t V, C, H, W;        /* source feature maps [V,C,H,W] */
    int32_t D, pad;            /* hypotheses on the padded target grid: Hp = H + 2 pad, Wp = W + 2 pad */
    const float* feats;        /* [V,C,H,W] */
    const float* proj;         /* [V,12]: (src_proj @ ref_proj_inv)[:3], row-major 3x4 */
    const float* depth_values; /* [D,Hp,Wp] */
    float* variance;           /* [C,D,Hp,Wp] */
    float* count;              /* [D,Hp,Wp] = 1 / (1 + views that see the voxel), or NULL */
} ucnerf_cost_volume_params;
int ucnerf_cost_volume(const ucnerf_cost_volume_params* p, void* stream);

typedef struct {
    int32_t D, Hp, Wp, pad;
    const float* prob_pre;     /* [D,Hp,Wp] logits from the regularisation network */
    const float* prob_init;    /* [D,Hp,Wp] added to the logits, or NULL */
    const float* depth_values; /* [D,Hp,Wp] */
    float* prob_volume;        /* [D,Hp,Wp] softmax over D */
    float* depth;              /* [Hp - 2 pad, Wp - 2 pad] */
    float* confidence;         /* [Hp - 2 pad, Wp - 2 pad] */
} ucnerf_depth_regress_params;
int ucnerf_depth_regress(const ucnerf_depth_regress_params* p, void* stream);

/* Backward of the two kernels above (what autograd sends through them in the reference: into the source feature maps
 * through the variance volume -- the sampling grid and the mask count carry no gradient, nearest-neighbour lookup -- and
 * into the regularisation network's logits through depth and photometric confidence; floor(E[d]) and the clamp's
 * inactive side pass nothing). */
typedef struct {
    ucnerf_cost_volume_params fwd;   /* variance / count are not used */
    const float* g_variance;         /* [C,D,Hp,Wp] */
    float* g_feats;                  /* [V,C,H,W], accumulated into (float atomics) */
} ucnerf_cost_volume_bwd_params;
int ucnerf_cost_volume_bwd(const ucnerf_cost_volume_bwd_params* p, void* stream);

typedef struct {
    ucnerf_depth_regress_params fwd; /* prob_volume = the forward's output (read); depth / confidence are not used */
    const float* g_depth;            /* [Hp - 2 pad, Wp - 2 pad] or NULL */
    const float* g_confidence;       /* [Hp - 2 pad, Wp - 2 pad] or NULL */
    float* g_prob_pre;               /* [D,Hp,Wp] (also the gradient of prob_init) */
} ucnerf_depth_regress_bwd_params;
int ucnerf_depth_regress_bwd(const ucnerf_depth_regress_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a10  one fused render pass -- network/renderer.py:215-255 (rendering) with the projection of
 *      utils/utils.py:716-724 in front: rays + depths -> world points -> stage coordinates -> features ->
 *      PE + MLP -> composite.  Source views = pose entries 1..V of the reference's pose_ref
 *      (the in-place trim of renderer.py:241-243 is host-side behaviour and stays in the Python mirror).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, S;
    int32_t white_bkgd;
    int32_t max_blocks;        /* MLP persistent grid cap, 0 = auto */
    ucnerf_mlp_config cfg;
    /* geometry */
    const float* rays_o;       /* [3] shared origin */
    const float* rays_d;       /* [n,3] */
    const float* z;            /* [n,S] */
    float w2c_ref[12];         /* reference view: projection for the stage/ndc coordinates */
    float K_ref[9];
    float w2c_dir[12];         /* rotation used for the view-direction feature (pose_ref['w2cs'][0] at call time) */
    float near, far;           /* scene range */
    const float* near_far;     /* [n,6] per-ray cascade ranges, or NULL: all stages use (near, far) */
    /* gather sources, see ucnerf_feat_gather_params */
    int32_t H, W;
    int32_t vol_d[3], vol_h[3], vol_w[3];
    const float* vol[3];
    const float* conf;
    const float* imgs;
    const float* img_feat;
    const float* w2cs;         /* [V,12] source views */
    const float* intrinsics;   /* [V,9] */
    const float* wstream;
    ucnerf_cl_sources cl;      /* optional (ABI v5; all five pointers or none): the sources channel-last, each in its own allocation -- handed over
                                  in place by a producer that writes that layout (vol / imgs / img_feat above may then be NULL) and / or built by
                                  ucnerf_gather_repack.  When given and `feats` is NULL, the pass uses the fast gather that reads them and derives the
                                  sample coordinates itself; vol/imgs/img_feat are then not touched.  REQUIRED when
                                  cfg.precision == 3 (the gather then runs inside the MLP kernel: no feature buffer).
                                  cl.bf16 = 1 (bf16 copies, ucnerf_gather_repack with the same flag: 16-byte voxels and feature pixels, 8-byte colours,
                                  values rounded to nearest even) -- SURVEY.md 8 configs[4] "fp32 MLP / bf16 features": half the bytes of every
                                  gather corner, features within bf16 rounding of the fp32 sources'.  Forward only reads differ; the backward
                                  accumulates source gradients in fp32 as before.  With cfg.precision == 3: derived coordinates only */
    /* workspace: ucnerf_render_workspace_floats(n, S, V) floats */
    float* workspace;
    /* outputs */
    float* rgb_map;            /* [n,3] */
    float* depth_map;          /* [n] */
    float* acc_map;            /* [n] or NULL */
    float* weights;            /* [n,S] or NULL */
    float* var;                /* [n] or NULL */
    float* raw;                /* [n,S,4] or NULL (kept for backward) */
    float* feats;              /* [n*S,F] row-major or NULL (kept for backward) */
    /* optional timing hooks: events (ucnerf_event_create) recorded on `stream` right before / after the MLP kernel */
    void* ev_mlp_start;
    void* ev_mlp_stop;
    float* train_workspace;    /* optional (training forward; needs raw and feats): the workspace of the coming
                                  ucnerf_render_fused_bwd call -- the MLP activations are kept there, see
                                  ucnerf_render_bwd_params.saved_valid */
    const float* dir_feat;     /* optional [n,3]: the view-direction feature already computed (ucnerf_ray_gen.angle or
                                  ucnerf_dir_feature with w2c_dir); NULL: the pass computes it itself */
    /* opt-in uncertainty outputs (SURVEY.md 8(a) note): */
    float* u_sampled;          /* [n,S] or NULL: u = 1 - confidence sampled at every depth (network/models.py:149) */
    float* wu_map;             /* [n] or NULL: sum_i w_i u_i */
    /* Coordinates GIVEN by the caller instead of derived from (rays_d, z): what rendering() of the reference receives
     * from build_rays / build_rays_test (network/renderer.py:215-255: rays_pts, rays_ndc).  All five or none; when set,
     * rays_o / rays_d / z are used for the view direction and the compositing only and near_far is ignored. */
    const float* pts_in;       /* [n*S,3] world points */
    const float* ndc1_in;      /* [n*S,3] stage coordinates in ~[0,1] */
    const float* ndc2_in;
    const float* ndc3_in;
    const float* ndc_in;       /* [n*S,3] scene-normalised copy fed to the positional encoding */
    int32_t feats_tiled;       /* layout of `feats` when it is kept: 0 row-major [n*S,F]; 1 the MLP's tile layout [ceil(n*S/32)][F][32]
                                  (ceil(n*S/32)*32*F floats) -- what the training forward reads three times faster (coalesced 128-byte
                                  rows instead of 4-byte pieces of 388-byte rows) and ucnerf_render_fused_bwd (bwd_mode 0) accepts */
    int32_t train_bwd_mode;    /* with train_workspace: the bwd_mode of the coming ucnerf_render_fused_bwd call (it fixes the format the
                                  activations are kept in, see ucnerf_mlp_fwd_train) */
    /* ABI v4: the launches around a coarse pass folded into it (data/ray_utils.py:199-224 chains them; at 512 rays per GPU each ~5-us launch
     * is 4 % of the step) */
    const ucnerf_sample_pdf_params* resample;   /* optional (HOST pointer): the pass's compositing launch also draws the NEXT pass's depths from its
                                  weights -- ucnerf_composite_sample_pdf(this pass's compositing, *resample) instead of ucnerf_composite_fwd;
                                  resample->weights / z_merge may be NULL (implied: this pass's weights and z) */
    const float* w2c_dir_dev;  /* ABI v5, optional DEVICE pointer to the rotation of the view-direction feature, row-major with 4 columns (a [3,4] or
                                  [4,4] tensor): used instead of w2c_dir when dir_feat is NULL and this is set -- rendering() holds
                                  pose_ref['w2cs'][0] on the device and need not read it back.  On the tail route the features are then made
                                  in the blocks' prologues (no ucnerf_dir_feature launch) */
    const ucnerf_ray_gen_params* gen_rays;      /* optional (HOST pointers, both or none; cfg.precision == 3, derived coordinates, xs / ys pixel lists): the */
    const ucnerf_sample_stratified_params* gen_depths;  /* fused launch generates its own rays and stratified depths -- ucnerf_ray_gen_sample folded into the
                                  kernel's per-tile prologue.  rays_d / z / dir_feat of THIS struct are then buffers the launch FILLS (they must
                                  equal gen_rays->rays_d / gen_depths->z / gen_rays->angle; later passes and the compositing read them);
                                  same values as ucnerf_ray_gen_sample, bit for bit */
} ucnerf_render_params;
int64_t ucnerf_render_workspace_floats(int32_t n, int32_t S, int32_t V);
int ucnerf_render_fused_fwd(const ucnerf_render_params* p, void* stream);
/* Channel-last repack of the gather sources named in `p` (vol[3] -> [D,h,w,8] each, img_feat -> [V,H,W,8], imgs -> [V,H,W,4] = (r,g,b,0); fp32, or
 * bf16 when p->cl.bf16) into `dst` (ucnerf_gather_repack_floats(p) floats, 16-byte aligned), and the pointers of the result into `out` (rgb_stride 4,
 * bf16 as asked).  PER SOURCE: an entry of p->cl that is already set -- a source handed over in place -- is kept as it is (copied to `out`, nothing
 * read or written for it, no room taken in dst; with p->cl.bf16 the in-place entries must hold bf16 too); its reference-layout pointer may be NULL.
 * Redo it whenever the repacked sources change (once per image in evaluation, once per step in training): ~150 MB of traffic for all five. */
int64_t ucnerf_gather_repack_floats(const ucnerf_render_params* p);
int ucnerf_gather_repack(const ucnerf_render_params* p, float* dst, ucnerf_cl_sources* out, void* stream);

/* Backward of one render pass: d(rgb_map, depth_map) -> d(parameters), d(volumes, img_feat, confidence).
 * fwd.raw and fwd.feats must point at the buffers the forward call filled (keep them); all g_* outputs are
 * ACCUMULATED (zero them first). */
typedef struct {
    ucnerf_render_params fwd;
    const float* g_rgb;            /* [n,3] */
    const float* g_depth;          /* [n] or NULL */
    const float* flat_params;      /* [param_count] */
    float* g_flat;                 /* [param_count] accumulated */
    float* g_vol[3];               /* accumulated or NULL */
    float* g_conf;
    float* g_img_feat;
    float* workspace;              /* ucnerf_render_bwd_workspace_floats(n, S, V) floats, 16-byte aligned */
    float* gather_scratch;         /* optional: ucnerf_feat_gather_bwd_params.scratch for the gather backward */
    int32_t saved_valid;           /* 1: the forward call was given this `workspace` as fwd.train_workspace (it kept the
                                      MLP activations there), so the backward does not repeat the network forward */
    int32_t bwd_mode;              /* ucnerf_mlp_bwd_params.bwd_mode */
    ucnerf_cl_grads g_cl;          /* ABI v5, per source: ucnerf_feat_gather_bwd_params.g_cl -- source gradients accumulated in the channel-last
                                      layout of the in-place sources (that source's g_vol / g_img_feat entry is then ignored; g_conf as before) */
} ucnerf_render_bwd_params;
int64_t ucnerf_render_bwd_workspace_floats(int32_t n, int32_t S, int32_t V);
int ucnerf_render_fused_bwd(const ucnerf_render_bwd_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UCNERF_HIP_H */
