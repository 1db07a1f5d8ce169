// K4: per-sample feature gather (network/renderer.py:177-212 = utils/utils.py:833-893 + :742-799 of the
// reference), forward and backward.
//
// Work is split into "units": 0,1,2 = the three cascade volumes (8 channels, trilinear, align_corners=False),
// 3 = confidence (bilinear, align_corners=False, sampled with the stage-3 grid), 4.. = one per source view
// (world point -> view pixel; bilinear rgb + in-mask + 8 image-feature channels, align_corners=True).
// blockIdx.y selects the unit, so every wave runs one code path and consecutive lanes are consecutive
// samples of a ray (neighbouring voxels -> L2 locality).  grid_sample semantics restated:
//   align_corners=False: i = ((g+1)*size - 1)/2;  align_corners=True: i = (g+1)/2*(size-1);
//   border padding: clamp i to [0, size-1] before floor/frac; a corner index == size gets weight 0.
#include "common.h"

namespace ucnerf {

__device__ __forceinline__ float unnorm(float g, int size, bool align) {
    float i = align ? (g + 1.f) / 2.f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) / 2.f;
    return fminf(fmaxf(i, 0.f), (float)(size - 1));
}

struct Lerp {        // one axis of a (bi|tri)linear footprint
    int i0, i1;      // corner indices (i1 clamped into range; its weight is 0 when it was out of range)
    float w0, w1;
};

__device__ __forceinline__ Lerp axis(float g, int size, bool align) {
    const float x = unnorm(g, size, align);
    const float f = floorf(x);
    Lerp a;
    a.i0 = (int)f;
    a.w1 = x - f;
    a.w0 = 1.f - a.w1;
    a.i1 = a.i0 + 1;
    if (a.i1 > size - 1) { a.i1 = size - 1; a.w1 = 0.f; }
    return a;
}

// feature f of sample s in the output tensor
__device__ __forceinline__ size_t out_index(const ucnerf_feat_gather_params& p, int F, int s, int f) {
    return p.out_tiled ? ((size_t)(s >> 5) * F + f) * 32 + (s & 31) : (size_t)s * F + f;
}

// world point -> (gx, gy) in [-1,1] of source view `v` (get_ndc_coordinate with sample_2d=True, then *2-1)
__device__ __forceinline__ void project_view(const ucnerf_feat_gather_params& p, int v, int s, float* gx, float* gy) {
    const float* M = p.w2cs + 12 * v;
    const float* K = p.intrinsics + 9 * v;
    const float x = p.pts[3 * (size_t)s], y = p.pts[3 * (size_t)s + 1], z = p.pts[3 * (size_t)s + 2];
    const float cx = x * M[0] + y * M[1] + z * M[2] + M[3];
    const float cy = x * M[4] + y * M[5] + z * M[6] + M[7];
    float cz = x * M[8] + y * M[9] + z * M[10] + M[11];
    if (fabsf(cz) < 1e-4f) cz = 1e-4f;
    const float qx = cx * K[0] + cy * K[1] + cz * K[2];
    const float qy = cx * K[3] + cy * K[4] + cz * K[5];
    const float qz = cx * K[6] + cy * K[7] + cz * K[8];
    *gx = (qx / qz + 0.0f) / (float)(p.W - 1) * 2.0f - 1.0f;
    *gy = (qy / qz + 0.0f) / (float)(p.H - 1) * 2.0f - 1.0f;
}

__global__ void __launch_bounds__(256) feat_gather_fwd_kernel(ucnerf_feat_gather_params p) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.m) return;
    const int unit = blockIdx.y;
    const int F = 24 + 12 * p.V + 1;
    if (p.unit_mask && !((p.unit_mask >> unit) & 1)) return;
    if (unit < 3) {
        const float* g = (unit == 0 ? p.ndc1 : unit == 1 ? p.ndc2 : p.ndc3) + 3 * (size_t)s;
        const int D = p.vol_d[unit], h = p.vol_h[unit], w = p.vol_w[unit];
        const Lerp ax = axis(g[0] * 2.f - 1.0f, w, false), ay = axis(g[1] * 2.f - 1.0f, h, false),
                   az = axis(g[2] * 2.f - 1.0f, D, false);
        const float* vol = p.vol[unit];
        const size_t cs = (size_t)D * h * w;
        const size_t o00 = ((size_t)az.i0 * h + ay.i0) * w, o01 = ((size_t)az.i0 * h + ay.i1) * w,
                     o10 = ((size_t)az.i1 * h + ay.i0) * w, o11 = ((size_t)az.i1 * h + ay.i1) * w;
        const float w00 = az.w0 * ay.w0, w01 = az.w0 * ay.w1, w10 = az.w1 * ay.w0, w11 = az.w1 * ay.w1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* vc = vol + c * cs;
            float acc = vc[o00 + ax.i0] * (w00 * ax.w0);
            acc += vc[o00 + ax.i1] * (w00 * ax.w1);
            acc += vc[o01 + ax.i0] * (w01 * ax.w0);
            acc += vc[o01 + ax.i1] * (w01 * ax.w1);
            acc += vc[o10 + ax.i0] * (w10 * ax.w0);
            acc += vc[o10 + ax.i1] * (w10 * ax.w1);
            acc += vc[o11 + ax.i0] * (w11 * ax.w0);
            acc += vc[o11 + ax.i1] * (w11 * ax.w1);
            p.feats[out_index(p, F, s, 8 * unit + c)] = acc;
        }
    } else if (unit == 3) {
        const float* g = p.ndc3 + 3 * (size_t)s;
        const Lerp ax = axis(g[0] * 2.f - 1.0f, p.W, false), ay = axis(g[1] * 2.f - 1.0f, p.H, false);
        const float* c = p.conf;
        float acc = c[(size_t)ay.i0 * p.W + ax.i0] * (ay.w0 * ax.w0);
        acc += c[(size_t)ay.i0 * p.W + ax.i1] * (ay.w0 * ax.w1);
        acc += c[(size_t)ay.i1 * p.W + ax.i0] * (ay.w1 * ax.w0);
        acc += c[(size_t)ay.i1 * p.W + ax.i1] * (ay.w1 * ax.w1);
        p.feats[out_index(p, F, s, F - 1)] = acc;
        if (p.u_out) p.u_out[s] = 1.f - acc;        // network/models.py:149
    } else {
        const int v = unit - 4;
        float gx, gy;
        project_view(p, v, s, &gx, &gy);
        const Lerp ax = axis(gx, p.W, true), ay = axis(gy, p.H, true);
        const size_t hw = (size_t)p.H * p.W;
        const size_t o00 = (size_t)ay.i0 * p.W + ax.i0, o01 = (size_t)ay.i0 * p.W + ax.i1,
                     o10 = (size_t)ay.i1 * p.W + ax.i0, o11 = (size_t)ay.i1 * p.W + ax.i1;
        const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
        const float* img = p.imgs + (size_t)v * 3 * hw;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* ic = img + c * hw;
            float acc = ic[o00] * w00; acc += ic[o01] * w01; acc += ic[o10] * w10; acc += ic[o11] * w11;
            p.feats[out_index(p, F, s, 24 + 4 * v + c)] = acc;
        }
        const bool in = gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f;      // utils/utils.py:791-792
        p.feats[out_index(p, F, s, 24 + 4 * v + 3)] = in ? 1.f : 0.f;
        const float* ft = p.img_feat + (size_t)v * 8 * hw;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* fc = ft + c * hw;
            float acc = fc[o00] * w00; acc += fc[o01] * w01; acc += fc[o10] * w10; acc += fc[o11] * w11;
            p.feats[out_index(p, F, s, 24 + 4 * p.V + 8 * v + c)] = acc;
        }
    }
}

// Backward: scatter g_feats through the same footprints (float atomics; accumulates into the g_* buffers).
__global__ void __launch_bounds__(256) feat_gather_bwd_kernel(ucnerf_feat_gather_bwd_params bp) {
    const ucnerf_feat_gather_params& p = bp.fwd;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.m) return;
    const int unit = blockIdx.y;
    const int F = 24 + 12 * p.V + 1;
    if (p.unit_mask && !((p.unit_mask >> unit) & 1)) return;
    const float* gf = bp.g_feats + (size_t)s * F;
    if (unit < 3) {
        float* gv = bp.g_vol[unit];
        if (!gv) return;
        const float* g = (unit == 0 ? p.ndc1 : unit == 1 ? p.ndc2 : p.ndc3) + 3 * (size_t)s;
        const int D = p.vol_d[unit], h = p.vol_h[unit], w = p.vol_w[unit];
        const Lerp ax = axis(g[0] * 2.f - 1.0f, w, false), ay = axis(g[1] * 2.f - 1.0f, h, false),
                   az = axis(g[2] * 2.f - 1.0f, D, false);
        const size_t cs = (size_t)D * h * w;
        const size_t o00 = ((size_t)az.i0 * h + ay.i0) * w, o01 = ((size_t)az.i0 * h + ay.i1) * w,
                     o10 = ((size_t)az.i1 * h + ay.i0) * w, o11 = ((size_t)az.i1 * h + ay.i1) * w;
        const float w00 = az.w0 * ay.w0, w01 = az.w0 * ay.w1, w10 = az.w1 * ay.w0, w11 = az.w1 * ay.w1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float* vc = gv + c * cs;
            const float gc = gf[8 * unit + c];
            atomicAdd(vc + o00 + ax.i0, gc * (w00 * ax.w0));
            if (ax.w1 != 0.f) atomicAdd(vc + o00 + ax.i1, gc * (w00 * ax.w1));
            if (w01 != 0.f) {
                atomicAdd(vc + o01 + ax.i0, gc * (w01 * ax.w0));
                if (ax.w1 != 0.f) atomicAdd(vc + o01 + ax.i1, gc * (w01 * ax.w1));
            }
            if (w10 != 0.f) {
                atomicAdd(vc + o10 + ax.i0, gc * (w10 * ax.w0));
                if (ax.w1 != 0.f) atomicAdd(vc + o10 + ax.i1, gc * (w10 * ax.w1));
            }
            if (w11 != 0.f) {
                atomicAdd(vc + o11 + ax.i0, gc * (w11 * ax.w0));
                if (ax.w1 != 0.f) atomicAdd(vc + o11 + ax.i1, gc * (w11 * ax.w1));
            }
        }
    } else if (unit == 3) {
        if (!bp.g_conf) return;
        const float* g = p.ndc3 + 3 * (size_t)s;
        const Lerp ax = axis(g[0] * 2.f - 1.0f, p.W, false), ay = axis(g[1] * 2.f - 1.0f, p.H, false);
        const float gc = gf[F - 1];
        atomicAdd(bp.g_conf + (size_t)ay.i0 * p.W + ax.i0, gc * (ay.w0 * ax.w0));
        atomicAdd(bp.g_conf + (size_t)ay.i0 * p.W + ax.i1, gc * (ay.w0 * ax.w1));
        atomicAdd(bp.g_conf + (size_t)ay.i1 * p.W + ax.i0, gc * (ay.w1 * ax.w0));
        atomicAdd(bp.g_conf + (size_t)ay.i1 * p.W + ax.i1, gc * (ay.w1 * ax.w1));
    } else {
        if (!bp.g_img_feat) return;
        const int v = unit - 4;
        float gx, gy;
        project_view(p, v, s, &gx, &gy);
        const Lerp ax = axis(gx, p.W, true), ay = axis(gy, p.H, true);
        const size_t hw = (size_t)p.H * p.W;
        const size_t o00 = (size_t)ay.i0 * p.W + ax.i0, o01 = (size_t)ay.i0 * p.W + ax.i1,
                     o10 = (size_t)ay.i1 * p.W + ax.i0, o11 = (size_t)ay.i1 * p.W + ax.i1;
        const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
        float* ft = bp.g_img_feat + (size_t)v * 8 * hw;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float* fc = ft + c * hw;
            const float gc = gf[24 + 4 * p.V + 8 * v + c];
            atomicAdd(fc + o00, gc * w00); atomicAdd(fc + o01, gc * w01);
            atomicAdd(fc + o10, gc * w10); atomicAdd(fc + o11, gc * w11);
        }
    }
}

// ---- channel-last accumulation (scratch given): thread = (sample, channel); the eight lanes of a sample add the 32
// contiguous bytes of one corner, so an atomic wave-instruction touches 8 cache lines instead of 64.
// A source whose gradient array is handed over channel-last (ucnerf_feat_gather_bwd_params.g_cl: the layouts of ucnerf_cl_sources) is accumulated
// THERE instead of in its piece of the scratch buffer -- same kernel, another base pointer.
struct ScratchLayout { size_t vol[3], img, total; };
__host__ __device__ inline ScratchLayout scratch_layout(const ucnerf_feat_gather_params& p) {
    ScratchLayout L;
    size_t o = 0;
    for (int k = 0; k < 3; ++k) { L.vol[k] = o; o += 8 * (size_t)p.vol_d[k] * p.vol_h[k] * p.vol_w[k]; }
    L.img = o; o += (size_t)8 * p.V * p.H * p.W;
    L.total = o;
    return L;
}

// Neighbouring samples of a ray fall into the same cell again and again (a ray crosses a source view along a short
// epipolar segment and the reference frustum along one column), so most atomics of a wave would hit addresses another
// lane of the same wave also hits.  The eight samples of a wave are therefore combined first: a segmented suffix sum
// over runs of equal cell offsets along the sample dimension (`run_atomic_add`, common.h), and only the first lane of a
// run issues the atomic.

// confidence map: thread = sample; the samples of a ray share their four pixels in the reference view
__global__ void __launch_bounds__(256) conf_bwd_kernel(ucnerf_feat_gather_bwd_params bp) {
    const ucnerf_feat_gather_params& p = bp.fwd;
    const int s_raw = blockIdx.x * 256 + threadIdx.x;
    const bool live = s_raw < p.m;
    const int s = live ? s_raw : p.m - 1;
    const int F = 24 + 12 * p.V + 1, pos = threadIdx.x & 63;
    const float* g = p.ndc3 + 3 * (size_t)s;
    const Lerp ax = axis(g[0] * 2.f - 1.0f, p.W, false), ay = axis(g[1] * 2.f - 1.0f, p.H, false);
    const float gc = live ? bp.g_feats[(size_t)s * F + F - 1] : 0.f;
    const int dead = live ? 0 : -1;
    run_atomic_add<1>(bp.g_conf, dead | (ay.i0 * p.W + ax.i0), gc * (ay.w0 * ax.w0), pos);
    run_atomic_add<1>(bp.g_conf, dead | (ay.i0 * p.W + ax.i1), gc * (ay.w0 * ax.w1), pos);
    run_atomic_add<1>(bp.g_conf, dead | (ay.i1 * p.W + ax.i0), gc * (ay.w1 * ax.w0), pos);
    run_atomic_add<1>(bp.g_conf, dead | (ay.i1 * p.W + ax.i1), gc * (ay.w1 * ax.w1), pos);
}

// The corner keys (element offsets of channel 0 of the cell, -1 past the end) and weighted gradients of one sample: the 8 corners of a volume's cell, the
// 4 of a source view's pixel.  (Two functions and two branches in the kernel, not one function returning either: arrays merged across a branch are
// not scalarised -- the compiler parked them in LDS.)
struct VolCorners { int key[8]; float val[8]; };
struct ViewCorners { int key[4]; float val[4]; };

__device__ __forceinline__ VolCorners vol_corners(const ucnerf_feat_gather_bwd_params& bp, int unit, long long s_raw, int c) {
    const ucnerf_feat_gather_params& p = bp.fwd;
    const bool live = s_raw < p.m;
    const size_t s = live ? (size_t)s_raw : (size_t)p.m - 1;
    const int F = 24 + 12 * p.V + 1;
    const int dead = live ? 0 : -1;                                     // keys of lanes past the end
    const float* g = (unit == 0 ? p.ndc1 : unit == 1 ? p.ndc2 : p.ndc3) + 3 * s;
    const int D = p.vol_d[unit], h = p.vol_h[unit], w = p.vol_w[unit];
    const Lerp ax = axis(g[0] * 2.f - 1.0f, w, false), ay = axis(g[1] * 2.f - 1.0f, h, false),
               az = axis(g[2] * 2.f - 1.0f, D, false);
    const int o00 = (az.i0 * h + ay.i0) * w, o01 = (az.i0 * h + ay.i1) * w,
              o10 = (az.i1 * h + ay.i0) * w, o11 = (az.i1 * h + ay.i1) * w;
    const float w00 = az.w0 * ay.w0, w01 = az.w0 * ay.w1, w10 = az.w1 * ay.w0, w11 = az.w1 * ay.w1;
    const float gc = live ? bp.g_feats[s * F + 8 * unit + c] : 0.f;
    // same products as the direct path (weights folded in the same order); zero-weight corners add nothing
    VolCorners r;
    r.key[0] = dead | (8 * (o00 + ax.i0)); r.val[0] = gc * (w00 * ax.w0);
    r.key[1] = dead | (8 * (o00 + ax.i1)); r.val[1] = gc * (w00 * ax.w1);
    r.key[2] = dead | (8 * (o01 + ax.i0)); r.val[2] = gc * (w01 * ax.w0);
    r.key[3] = dead | (8 * (o01 + ax.i1)); r.val[3] = gc * (w01 * ax.w1);
    r.key[4] = dead | (8 * (o10 + ax.i0)); r.val[4] = gc * (w10 * ax.w0);
    r.key[5] = dead | (8 * (o10 + ax.i1)); r.val[5] = gc * (w10 * ax.w1);
    r.key[6] = dead | (8 * (o11 + ax.i0)); r.val[6] = gc * (w11 * ax.w0);
    r.key[7] = dead | (8 * (o11 + ax.i1)); r.val[7] = gc * (w11 * ax.w1);
    return r;
}

__device__ __forceinline__ ViewCorners view_corners(const ucnerf_feat_gather_bwd_params& bp, int v, long long s_raw, int c) {
    const ucnerf_feat_gather_params& p = bp.fwd;
    const bool live = s_raw < p.m;
    const int s = live ? (int)s_raw : p.m - 1;
    const int F = 24 + 12 * p.V + 1;
    const int dead = live ? 0 : -1;
    float gx, gy;
    project_view(p, v, s, &gx, &gy);
    const Lerp ax = axis(gx, p.W, true), ay = axis(gy, p.H, true);
    const float gc = live ? bp.g_feats[(size_t)s * F + 24 + 4 * p.V + 8 * v + c] : 0.f;
    ViewCorners r;
    r.key[0] = dead | (8 * (ay.i0 * p.W + ax.i0)); r.val[0] = gc * (ay.w0 * ax.w0);
    r.key[1] = dead | (8 * (ay.i0 * p.W + ax.i1)); r.val[1] = gc * (ay.w0 * ax.w1);
    r.key[2] = dead | (8 * (ay.i1 * p.W + ax.i0)); r.val[2] = gc * (ay.w1 * ax.w0);
    r.key[3] = dead | (8 * (ay.i1 * p.W + ax.i1)); r.val[3] = gc * (ay.w1 * ax.w1);
    return r;
}

// thread = (sample pair, channel): a wave covers SIXTEEN consecutive samples -- lane (sl, c) holds samples 16 w + sl and 16 w + 8 + sl of channel c -- and
// combines runs of equal cells over all sixteen before any atomic is issued (run_atomic_add2: round 5; it was eight samples per wave)
__global__ void __launch_bounds__(256) feat_gather_bwd_cl_kernel(ucnerf_feat_gather_bwd_params bp) {
    const ucnerf_feat_gather_params& p = bp.fwd;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int c = threadIdx.x & 7, sl = (threadIdx.x >> 3) & 7;       // channel, sample within the batch of eight
    const long long s_a = wave * 16 + sl, s_b = s_a + 8;              // (past the end: clamped inside the corner functions, keys -1)
    const int unit = blockIdx.y < 3 ? blockIdx.y : blockIdx.y + 1;      // units 0..2 volumes, 4.. views (conf stays on the direct path)
    if (p.unit_mask && !((p.unit_mask >> unit) & 1)) return;            // (uniform per block)
    const ScratchLayout L = scratch_layout(p);
    if (unit < 3) {
        if (!bp.g_cl.vol[unit] && !bp.g_vol[unit]) return;
        float* gv = (bp.g_cl.vol[unit] ? bp.g_cl.vol[unit] : bp.scratch + L.vol[unit]) + c;
        const VolCorners A = vol_corners(bp, unit, s_a, c), B = vol_corners(bp, unit, s_b, c);
#pragma unroll
        for (int k = 0; k < 8; ++k) run_atomic_add2<8>(gv, A.key[k], A.val[k], B.key[k], B.val[k], sl);
    } else {
        if (!bp.g_cl.img_feat && !bp.g_img_feat) return;
        const int v = unit - 4;
        float* ft = (bp.g_cl.img_feat ? bp.g_cl.img_feat : bp.scratch + L.img) + (size_t)8 * v * ((size_t)p.H * p.W) + c;
        const ViewCorners A = view_corners(bp, v, s_a, c), B = view_corners(bp, v, s_b, c);
#pragma unroll
        for (int k = 0; k < 4; ++k) run_atomic_add2<8>(ft, A.key[k], A.val[k], B.key[k], B.val[k], sl);
    }
}

// g[c][i] += scratch[i][c] for every source in one launch (blockIdx.y = source; single writer per element)
struct AddTransposedArgs {
    int count;
    const float4* src[11];       // 3 volumes + up to 8 views
    float* dst[11];
    size_t n[11];                // positions per source
};

__global__ void __launch_bounds__(256) add_transposed_kernel(AddTransposedArgs a) {
    const int k = blockIdx.y;
    const size_t n = a.n[k];
    const float4* __restrict__ src = a.src[k];
    float* __restrict__ dst = a.dst[k];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 u = src[2 * i], v = src[2 * i + 1];
        dst[i] += u.x; dst[n + i] += u.y; dst[2 * n + i] += u.z; dst[3 * n + i] += u.w;
        dst[4 * n + i] += v.x; dst[5 * n + i] += v.y; dst[6 * n + i] += v.z; dst[7 * n + i] += v.w;
    }
}

static int check_geometry(const ucnerf_feat_gather_params* p, const char* who) {
    const int mask = p->unit_mask ? p->unit_mask : ~0;
    UCNERF_REQUIRE(p->V >= 1 && p->V <= 8, "%s: V = %d outside 1..8", who, p->V);
    UCNERF_REQUIRE(p->H >= 2 && p->W >= 2, "%s: image size %dx%d", who, p->H, p->W);
    const float* nd[3] = {p->ndc1, p->ndc2, p->ndc3};
    for (int k = 0; k < 3; ++k)
        if (mask & (1 << k)) {
            UCNERF_REQUIRE(nd[k], "%s: null stage-%d coordinates", who, k + 1);
            UCNERF_REQUIRE(p->vol_d[k] >= 1 && p->vol_h[k] >= 1 && p->vol_w[k] >= 1, "%s: volume %d has an empty dimension", who, k);
        }
    if (mask & 8) UCNERF_REQUIRE(p->ndc3, "%s: the confidence lookup needs stage-3 coordinates", who);
    if (mask & (((1 << p->V) - 1) << 4)) UCNERF_REQUIRE(p->pts && p->w2cs && p->intrinsics, "%s: null points / camera arrays", who);
    return UCNERF_OK;
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int ucnerf_feat_gather_fwd(const ucnerf_feat_gather_params* p, void* stream) {
    UCNERF_REQUIRE(p, "feat_gather_fwd: null params");
    UCNERF_COUNT(p->m);
    int rc = check_geometry(p, "feat_gather_fwd");
    if (rc) return rc;
    const int mask = p->unit_mask ? p->unit_mask : ~0;
    for (int k = 0; k < 3; ++k) UCNERF_REQUIRE(!(mask & (1 << k)) || p->vol[k], "feat_gather_fwd: null volume %d", k);
    UCNERF_REQUIRE(!(mask & 8) || p->conf, "feat_gather_fwd: null confidence");
    UCNERF_REQUIRE(!(mask & (((1 << p->V) - 1) << 4)) || (p->imgs && p->img_feat), "feat_gather_fwd: null images / image features");
    UCNERF_REQUIRE(p->feats, "feat_gather_fwd: null output");
    UCNERF_COUNT(p->m);
    hipLaunchKernelGGL(feat_gather_fwd_kernel, dim3(cdiv(p->m, 256), 4 + p->V), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("feat_gather_fwd");
}

int ucnerf_feat_gather_bwd(const ucnerf_feat_gather_bwd_params* bp, void* stream) {
    UCNERF_REQUIRE(bp, "feat_gather_bwd: null params");
    UCNERF_COUNT(bp->fwd.m);
    UCNERF_REQUIRE(bp->g_feats, "feat_gather_bwd: null g_feats");
    int rc = check_geometry(&bp->fwd, "feat_gather_bwd");
    if (rc) return rc;
    UCNERF_COUNT(bp->fwd.m);
    hipStream_t st = (hipStream_t)stream;
    const ucnerf_feat_gather_params& f = bp->fwd;
    // per source: accumulate channel-last straight into the caller's gradient array (g_cl, ABI v5), or in the scratch buffer and add transposed
    const ucnerf_cl_grads& gc = bp->g_cl;
    const bool any_cl = gc.vol[0] || gc.vol[1] || gc.vol[2] || gc.img_feat;
    const bool via_scratch = (bp->g_vol[0] && !gc.vol[0]) || (bp->g_vol[1] && !gc.vol[1]) || (bp->g_vol[2] && !gc.vol[2]) || (bp->g_img_feat && !gc.img_feat);
    if (!bp->scratch && !any_cl) {
        hipLaunchKernelGGL(feat_gather_bwd_kernel, dim3(cdiv(f.m, 256), 4 + f.V), dim3(256), 0, st, *bp);
        return check_launch("feat_gather_bwd");
    }
    UCNERF_REQUIRE(!via_scratch || bp->scratch, "feat_gather_bwd: channel-last gradients for some sources only: the others need the scratch buffer");
    UCNERF_REQUIRE(((uintptr_t)bp->scratch & 15) == 0, "feat_gather_bwd: scratch must be 16-byte aligned");
    UCNERF_REQUIRE((((uintptr_t)gc.vol[0] | (uintptr_t)gc.vol[1] | (uintptr_t)gc.vol[2] | (uintptr_t)gc.img_feat) & 15) == 0, "feat_gather_bwd: g_cl arrays must be 16-byte aligned");
    const ScratchLayout L = scratch_layout(f);
    if (via_scratch) {      // zero what the kernel accumulates in: the used pieces of the scratch buffer (contiguous when all four take this route)
        const size_t vox[3] = {8 * (size_t)f.vol_d[0] * f.vol_h[0] * f.vol_w[0], 8 * (size_t)f.vol_d[1] * f.vol_h[1] * f.vol_w[1], 8 * (size_t)f.vol_d[2] * f.vol_h[2] * f.vol_w[2]};
        if (!any_cl) {
            if (hipMemsetAsync(bp->scratch, 0, L.total * sizeof(float), st) != hipSuccess) return fail(UCNERF_EHIP, "feat_gather_bwd: memset failed");
        } else {
            for (int k = 0; k < 3; ++k)
                if (bp->g_vol[k] && !gc.vol[k] && hipMemsetAsync(bp->scratch + L.vol[k], 0, vox[k] * sizeof(float), st) != hipSuccess) return fail(UCNERF_EHIP, "feat_gather_bwd: memset failed");
            if (bp->g_img_feat && !gc.img_feat && hipMemsetAsync(bp->scratch + L.img, 0, (L.total - L.img) * sizeof(float), st) != hipSuccess) return fail(UCNERF_EHIP, "feat_gather_bwd: memset failed");
        }
    }
    hipLaunchKernelGGL(feat_gather_bwd_cl_kernel, dim3(cdiv((long long)f.m * 4, 256), 3 + f.V), dim3(256), 0, st, *bp);      // (a thread serves two samples)
    if (bp->g_conf && (8 & (f.unit_mask ? f.unit_mask : ~0)))      // confidence: straight into the map (one channel), runs combined per wave
        hipLaunchKernelGGL(conf_bwd_kernel, dim3(cdiv(f.m, 256)), dim3(256), 0, st, *bp);
    if (!via_scratch) return check_launch("feat_gather_bwd (channel-last gradients)");
    const int mask = f.unit_mask ? f.unit_mask : ~0;
    AddTransposedArgs at;
    at.count = 0;
    size_t n_max = 0;
    auto add = [&](const float* src, float* dst, size_t n) {
        at.src[at.count] = (const float4*)src; at.dst[at.count] = dst; at.n[at.count] = n; ++at.count;
        if (n > n_max) n_max = n;
    };
    for (int k = 0; k < 3; ++k)
        if (bp->g_vol[k] && !gc.vol[k] && (mask & (1 << k))) add(bp->scratch + L.vol[k], bp->g_vol[k], (size_t)f.vol_d[k] * f.vol_h[k] * f.vol_w[k]);
    if (bp->g_img_feat && !gc.img_feat) {
        const size_t hw = (size_t)f.H * f.W;
        for (int v = 0; v < f.V; ++v)
            if (mask & (1 << (4 + v))) add(bp->scratch + L.img + 8 * (size_t)v * hw, bp->g_img_feat + 8 * (size_t)v * hw, hw);
    }
    if (at.count) {
        int bx = cdiv(n_max, 256);
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(add_transposed_kernel, dim3(bx, at.count), dim3(256), 0, st, at);
    }
    return check_launch("feat_gather_bwd");
}

int64_t ucnerf_feat_gather_bwd_scratch_floats(const ucnerf_feat_gather_params* p) {
    if (!p) return fail(UCNERF_EINVAL, "feat_gather_bwd_scratch_floats: null params");
    return (int64_t)scratch_layout(*p).total;
}

}  // extern "C"
