// f2: the step in front of the ray-marching path -- cost-volume assembly and depth regression of one cascade MVS stage,
// forward and backward
// (network/mvs_models.py:589-646, utils/utils.py:1105-1172).  The source maps (a few MB) stay in L2; the HBM traffic is
// the volume that is written (cost volume) or read and written (regression) -- 3 to 50 MB at the reference's stage sizes,
// i.e. tens of microseconds: launch latency and (for the cost volume) the projection arithmetic matter more than bandwidth.
#include "common.h"

namespace ucnerf {

// ---- cost volume.  What bounds it is not HBM but the projection: seven correctly rounded divisions per voxel and source
// view (they decide which pixel a nearest-neighbour lookup lands on, so they are kept exact).  Two layouts:
//  * wide maps (C > 16): block = 32 voxels x 8 lanes; lane i < V projects the voxel into view i once, shared through LDS,
//    then lane l streams channels l, l + 8, ...
//  * C <= 16: thread = voxel, all views and channels in registers (no LDS round, fewer threads per voxel; measured
//    faster up to 16 channels).
// Output [C][D][Hp*Wp]: for one channel, consecutive voxels are consecutive addresses.
constexpr int CV_MAX_VIEWS = 8, CV_VOX = 32;

struct CvHit { int idx; float mask; };

// Round 5 (counters: the kernels are bound by vector-ALU issue, profiles/r05_mvs_cost_volume_pmc.json): every gather and store through a 32-bit BYTE
// offset from a block-uniform base -- the address is one scalar pair + one VGPR, one vector add per access instead of a 64-bit shift-and-add chain
// (the host checks that the source maps and the volume stay below 4 GB).
__device__ __forceinline__ float cv_ld(const float* base, unsigned byte_off) { return *(const float*)((const char*)base + byte_off); }
__device__ __forceinline__ void cv_st(float* base, unsigned byte_off, float v) { *(float*)((char*)base + byte_off) = v; }

__device__ __forceinline__ CvHit cv_project(const ucnerf_cost_volume_params& p, int view, int pix, int Wp, float depth) {
    const float x = (float)(pix % Wp - p.pad), y = (float)(pix / Wp - p.pad);          // create_meshgrid - pad  (utils.py:1128-1134)
    const float sx = (float)(p.W - 1) / 2.f, sy = (float)(p.H - 1) / 2.f;
    const float* M = p.proj + 12 * view;
    // R @ (x, y, 1) + T / depth   (utils.py:1143-1144)
    const float qx = ((M[0] * x + M[1] * y) + M[2]) + M[3] / depth;
    const float qy = ((M[4] * x + M[5] * y) + M[6]) + M[7] / depth;
    const float qz = ((M[8] * x + M[9] * y) + M[10]) + M[11] / depth;
    const float gx = (qx / qz) / sx - 1.f, gy = (qy / qz) / sy - 1.f;                  // :1152-1157
    // grid_sample(nearest, border, align_corners=True): unnormalise, clip, round half to even
    float ix = ((gx + 1.f) / 2.f) * (float)(p.W - 1), iy = ((gy + 1.f) / 2.f) * (float)(p.H - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(p.W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(p.H - 1));
    CvHit h;
    h.idx = (int)rintf(iy) * p.W + (int)rintf(ix);
    h.mask = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.f : 0.f;         // mvs_models.py:616-617
    return h;
}

__global__ void __launch_bounds__(CV_VOX * CV_MAX_VIEWS) cost_volume_wide_kernel(ucnerf_cost_volume_params p) {
    __shared__ int sidx[CV_MAX_VIEWS][CV_VOX];
    __shared__ float smask[CV_MAX_VIEWS][CV_VOX];
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;
    const unsigned plane = (unsigned)Hp * Wp, total = plane * p.D;                     // (host checks total < 2^31)
    const int vox = threadIdx.x % CV_VOX, ln = threadIdx.x / CV_VOX;
    const unsigned t_raw = blockIdx.x * CV_VOX + vox;
    const bool live = t_raw < total;
    const unsigned t = live ? t_raw : total - 1;
    if (ln < p.V) {
        const CvHit h = cv_project(p, ln, (int)(t % plane), Wp, p.depth_values[t]);
        sidx[ln][vox] = h.idx; smask[ln][vox] = h.mask;
    }
    __syncthreads();
    float msum = 1.f;                                                                  // the mask stack's untouched slot (mvs_models.py:607)
    for (int i = 0; i < p.V; ++i) msum += smask[i][vox];
    const float count = 1.0f / msum;
    if (p.count && ln == 0 && live) p.count[t] = count;
    const unsigned hw4 = (unsigned)p.H * p.W * 4u, chw4 = hw4 * p.C;                   // bytes (host checks V*C*H*W*4 < 2^32)
    for (int c = ln; c < p.C; c += CV_MAX_VIEWS) {
        float s = 0.f, q = 0.f;
        for (int i = 0; i < p.V; ++i) {
            const float v = cv_ld(p.feats, i * chw4 + c * hw4 + 4u * (unsigned)sidx[i][vox]);
            s = s + v;                                                                 // volume_sum, volume_sq_sum in view order (:620-621)
            q = q + v * v;
        }
        const float mean = s * count;
        if (live) cv_st(p.variance, (unsigned)c * (total * 4u) + 4u * t, q * count - mean * mean);         // :624
    }
}

__global__ void __launch_bounds__(256) cost_volume_narrow_kernel(ucnerf_cost_volume_params p) {
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;
    const unsigned plane = (unsigned)Hp * Wp, total = plane * p.D;
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const float depth = p.depth_values[t];
    unsigned idx[CV_MAX_VIEWS];                                                        // BYTE offsets of the voxel's pixel in every view's channel 0
    float msum = 1.f;
    const unsigned hw4 = (unsigned)p.H * p.W * 4u, chw4 = hw4 * p.C;
#pragma unroll
    for (int i = 0; i < CV_MAX_VIEWS; ++i)
        if (i < p.V) {
            const CvHit h = cv_project(p, i, (int)(t % plane), Wp, depth);
            idx[i] = i * chw4 + 4u * (unsigned)h.idx;
            msum += h.mask;
        }
    const float count = 1.0f / msum;
    if (p.count) p.count[t] = count;
    for (int c = 0; c < p.C; ++c) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < CV_MAX_VIEWS; ++i)
            if (i < p.V) {
                const float v = cv_ld(p.feats, idx[i] + c * hw4);
                s = s + v;
                q = q + v * v;
            }
        const float mean = s * count;
        cv_st(p.variance, (unsigned)c * (total * 4u) + 4u * t, q * count - mean * mean);
    }
}

// ---- depth regression: block = 32 pixels x 8 depth lanes (lane dl owns depths dl, dl + 8, ...; for a fixed depth the 32
// pixels of a row of the block are 128 contiguous bytes).  Maximum, normaliser and the two expectations are combined
// across the depth lanes through LDS; the probabilities are written once; the confidence's 4-tap window takes them from an LDS copy.
constexpr int DR_PIX = 32, DR_DL = 8, DR_MAX_D = 128;

__global__ void __launch_bounds__(DR_PIX * DR_DL) depth_regress_kernel(ucnerf_depth_regress_params p) {
    __shared__ float red[3][DR_DL][DR_PIX];
    __shared__ float prs[DR_MAX_D][DR_PIX];              // the block's probabilities, for the confidence window
    const long long plane = (long long)p.Hp * p.Wp;
    const int px = threadIdx.x % DR_PIX, dl = threadIdx.x / DR_PIX;
    const long long t_raw = (long long)blockIdx.x * DR_PIX + px;
    const bool live = t_raw < plane;
    const long long t = live ? t_raw : plane - 1;
    auto logit = [&](int d) {
        const float v = p.prob_pre[(size_t)d * plane + t];
        return p.prob_init ? v + p.prob_init[(size_t)d * plane + t] : v;
    };
    float mx = -INFINITY;
    for (int d = dl; d < p.D; d += DR_DL) mx = fmaxf(mx, logit(d));
    red[0][dl][px] = mx;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DR_DL; ++k) mx = fmaxf(mx, red[0][k][px]);
    float den = 0.f;
    for (int d = dl; d < p.D; d += DR_DL) den += expf(logit(d) - mx);
    __syncthreads();
    red[0][dl][px] = den;
    __syncthreads();
    den = 0.f;
#pragma unroll
    for (int k = 0; k < DR_DL; ++k) den += red[0][k][px];
    float depth = 0.f, didx = 0.f;
    for (int d = dl; d < p.D; d += DR_DL) {
        const float pr = expf(logit(d) - mx) / den;
        if (live) p.prob_volume[(size_t)d * plane + t] = pr;
        prs[d][px] = pr;
        depth += pr * p.depth_values[(size_t)d * plane + t];
        didx += pr * (float)d;
    }
    red[1][dl][px] = depth; red[2][dl][px] = didx;
    __syncthreads();
    if (dl != 0 || !live) return;
    depth = 0.f; didx = 0.f;
#pragma unroll
    for (int k = 0; k < DR_DL; ++k) { depth += red[1][k][px]; didx += red[2][k][px]; }
    int di = (int)didx;                                  // .long(): truncation (the value is non-negative)
    di = di < 0 ? 0 : di > p.D - 1 ? p.D - 1 : di;
    // 4 * avg_pool3d over depths di-1 .. di+2 of the zero-padded volume, summed in that order (:635-636)
    float s4 = 0.f;
#pragma unroll
    for (int k = -1; k <= 2; ++k) {
        const int d = di + k;
        s4 += (d >= 0 && d < p.D) ? prs[d][px] : 0.f;
    }
    s4 = 4.f * (s4 / 4.f);
    s4 = fminf(fmaxf(s4, 0.f), 1.f);
    const int yy = (int)(t / p.Wp) - p.pad, xx = (int)(t % p.Wp) - p.pad;
    const int H = p.Hp - 2 * p.pad, W = p.Wp - 2 * p.pad;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        p.depth[(size_t)yy * W + xx] = depth;
        p.confidence[(size_t)yy * W + xx] = s4;
    }
}

// ---- backward of the cost volume: var = q * count - (s * count)^2  =>  d var / d v_i = 2 * count * (v_i - mean).
// The hypotheses of one target pixel project onto a short epipolar segment of every source view, so neighbouring depths
// keep landing on the same source pixel (stage 1: ~10 depths per pixel) and a scatter per voxel queues that many atomics
// on one address.  A wave therefore holds 8 pixels x 8 consecutive depths (lane = 8 * depth position + pixel): per
// channel and view, runs of equal source pixels along the depth positions are summed with a segmented suffix scan
// (`run_atomic_add<8>`, common.h: three shuffle steps) and only the first lane of a run issues the atomic.
__global__ void __launch_bounds__(256) cost_volume_bwd_kernel(ucnerf_cost_volume_bwd_params bp) {
    const ucnerf_cost_volume_params& p = bp.fwd;
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;
    const unsigned plane = (unsigned)Hp * Wp, total = plane * p.D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 7, dpos = lane >> 3;
    if ((blockIdx.y * 4 + wave) * 8 >= p.D) return;                      // (whole wave past the last depth; no block-level sync below)
    const unsigned pix_raw = blockIdx.x * 8 + pl;
    const int d_raw = (blockIdx.y * 4 + wave) * 8 + dpos;
    const bool live = pix_raw < plane && d_raw < p.D;
    const unsigned pix = pix_raw < plane ? pix_raw : plane - 1;
    const int d = d_raw < p.D ? d_raw : p.D - 1;
    const unsigned t = d * plane + pix;
    const unsigned hw = (unsigned)p.H * p.W, chw = hw * p.C;
    const float depth = p.depth_values[t];
    int idx[CV_MAX_VIEWS];
    float msum = 1.f;
#pragma unroll
    for (int i = 0; i < CV_MAX_VIEWS; ++i)
        if (i < p.V) {
            const CvHit h = cv_project(p, i, (int)pix, Wp, depth);
            idx[i] = h.idx;
            msum += h.mask;
        }
    const float count = 1.0f / msum;
    for (int c = 0; c < p.C; ++c) {
        float v[CV_MAX_VIEWS], s = 0.f;
#pragma unroll
        for (int i = 0; i < CV_MAX_VIEWS; ++i)
            if (i < p.V) { v[i] = p.feats[i * chw + c * hw + (unsigned)idx[i]]; s = s + v[i]; }
        const float mean = s * count;
        const float g2 = live ? 2.f * count * bp.g_variance[(size_t)c * total + t] : 0.f;
#pragma unroll
        for (int i = 0; i < CV_MAX_VIEWS; ++i)
            if (i < p.V) run_atomic_add<8>(bp.g_feats + (i * chw + c * hw), live ? idx[i] : -1, g2 * (v[i] - mean), dpos);
    }
}

// ---- backward of the depth regression (same block shape as the forward, so that floor(E[d]) is recomputed from the very
// same partial sums): g_p[d] = g_depth * depth_values[d] + g_conf * [d in the window and the clamp passes];
// g_x = p * (g_p - sum_d p g_p).
__global__ void __launch_bounds__(DR_PIX * DR_DL) depth_regress_bwd_kernel(ucnerf_depth_regress_bwd_params bp) {
    const ucnerf_depth_regress_params& p = bp.fwd;
    __shared__ float red[2][DR_DL][DR_PIX];
    __shared__ float prs[DR_MAX_D][DR_PIX];
    const long long plane = (long long)p.Hp * p.Wp;
    const int px = threadIdx.x % DR_PIX, dl = threadIdx.x / DR_PIX;
    const long long t_raw = (long long)blockIdx.x * DR_PIX + px;
    const bool live = t_raw < plane;
    const long long t = live ? t_raw : plane - 1;
    float didx = 0.f;
    for (int d = dl; d < p.D; d += DR_DL) {
        const float pr = p.prob_volume[(size_t)d * plane + t];
        prs[d][px] = pr;
        didx += pr * (float)d;
    }
    red[0][dl][px] = didx;
    __syncthreads();
    didx = 0.f;
#pragma unroll
    for (int k = 0; k < DR_DL; ++k) didx += red[0][k][px];
    int di = (int)didx;
    di = di < 0 ? 0 : di > p.D - 1 ? p.D - 1 : di;
    float s4 = 0.f;
#pragma unroll
    for (int k = -1; k <= 2; ++k) {
        const int d = di + k;
        s4 += (d >= 0 && d < p.D) ? prs[d][px] : 0.f;
    }
    s4 = 4.f * (s4 / 4.f);
    const int yy = (int)(t / p.Wp) - p.pad, xx = (int)(t % p.Wp) - p.pad;
    const int H = p.Hp - 2 * p.pad, W = p.Wp - 2 * p.pad;
    const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float gd = inside && bp.g_depth ? bp.g_depth[(size_t)yy * W + xx] : 0.f;
    float gc = inside && bp.g_confidence ? bp.g_confidence[(size_t)yy * W + xx] : 0.f;
    if (!(s4 >= 0.f && s4 <= 1.f)) gc = 0.f;                              // clamp(0, 1) passes the gradient inside the range only
    float dot = 0.f;
    for (int d = dl; d < p.D; d += DR_DL) {
        const float gp = gd * p.depth_values[(size_t)d * plane + t] + ((d >= di - 1 && d <= di + 2) ? gc : 0.f);
        dot += prs[d][px] * gp;
    }
    red[1][dl][px] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int k = 0; k < DR_DL; ++k) dot += red[1][k][px];
    if (!live) return;
    for (int d = dl; d < p.D; d += DR_DL) {
        const float gp = gd * p.depth_values[(size_t)d * plane + t] + ((d >= di - 1 && d <= di + 2) ? gc : 0.f);
        bp.g_prob_pre[(size_t)d * plane + t] = prs[d][px] * (gp - dot);
    }
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int ucnerf_cost_volume(const ucnerf_cost_volume_params* p, void* stream) {
    UCNERF_REQUIRE(p, "cost_volume: null params");
    UCNERF_REQUIRE(p->V >= 1 && p->V <= CV_MAX_VIEWS, "cost_volume: V = %d outside 1..%d", p->V, CV_MAX_VIEWS);
    UCNERF_REQUIRE(p->C >= 1 && p->H >= 2 && p->W >= 2 && p->D >= 1 && p->pad >= 0, "cost_volume: bad sizes C=%d H=%d W=%d D=%d pad=%d",
                   p->C, p->H, p->W, p->D, p->pad);
    UCNERF_REQUIRE(p->feats && p->proj && p->depth_values && p->variance, "cost_volume: null pointer");
    const long long total = (long long)p->D * (p->H + 2 * p->pad) * (p->W + 2 * p->pad);
    UCNERF_REQUIRE(total * p->C < (1ll << 30) && (long long)p->V * p->C * p->H * p->W < (1ll << 30), "cost_volume: volume / maps of 4 GB and more (32-bit byte offsets)");
    if (p->C > 16) hipLaunchKernelGGL(cost_volume_wide_kernel, dim3(cdiv(total, CV_VOX)), dim3(CV_VOX * CV_MAX_VIEWS), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(cost_volume_narrow_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("cost_volume");
}

int ucnerf_depth_regress(const ucnerf_depth_regress_params* p, void* stream) {
    UCNERF_REQUIRE(p, "depth_regress: null params");
    UCNERF_REQUIRE(p->D >= 1 && p->D <= DR_MAX_D && p->pad >= 0 && p->Hp > 2 * p->pad && p->Wp > 2 * p->pad, "depth_regress: bad sizes D=%d (max 128) Hp=%d Wp=%d pad=%d",
                   p->D, p->Hp, p->Wp, p->pad);
    UCNERF_REQUIRE(p->prob_pre && p->depth_values && p->prob_volume && p->depth && p->confidence, "depth_regress: null pointer");
    const long long plane = (long long)p->Hp * p->Wp;
    hipLaunchKernelGGL(depth_regress_kernel, dim3(cdiv(plane, DR_PIX)), dim3(DR_PIX * DR_DL), 0, (hipStream_t)stream, *p);
    return check_launch("depth_regress");
}

int ucnerf_cost_volume_bwd(const ucnerf_cost_volume_bwd_params* bp, void* stream) {
    UCNERF_REQUIRE(bp, "cost_volume_bwd: null params");
    const ucnerf_cost_volume_params* p = &bp->fwd;
    UCNERF_REQUIRE(p->V >= 1 && p->V <= CV_MAX_VIEWS, "cost_volume_bwd: V = %d outside 1..%d", p->V, CV_MAX_VIEWS);
    UCNERF_REQUIRE(p->C >= 1 && p->H >= 2 && p->W >= 2 && p->D >= 1 && p->pad >= 0, "cost_volume_bwd: bad sizes");
    UCNERF_REQUIRE(p->feats && p->proj && p->depth_values && bp->g_variance && bp->g_feats, "cost_volume_bwd: null pointer");
    const long long total = (long long)p->D * (p->H + 2 * p->pad) * (p->W + 2 * p->pad);
    UCNERF_REQUIRE(total < (1ll << 31) && (long long)p->V * p->C * p->H * p->W < (1ll << 31), "cost_volume_bwd: volume / maps too large for 32-bit indices");
    hipLaunchKernelGGL(cost_volume_bwd_kernel, dim3(cdiv(total / p->D, 8), cdiv(p->D, 32)), dim3(256), 0, (hipStream_t)stream, *bp);
    return check_launch("cost_volume_bwd");
}

int ucnerf_depth_regress_bwd(const ucnerf_depth_regress_bwd_params* bp, void* stream) {
    UCNERF_REQUIRE(bp, "depth_regress_bwd: null params");
    const ucnerf_depth_regress_params* p = &bp->fwd;
    UCNERF_REQUIRE(p->D >= 1 && p->D <= DR_MAX_D && p->pad >= 0 && p->Hp > 2 * p->pad && p->Wp > 2 * p->pad, "depth_regress_bwd: bad sizes");
    UCNERF_REQUIRE(p->prob_volume && p->depth_values && bp->g_prob_pre, "depth_regress_bwd: null pointer");
    const long long plane = (long long)p->Hp * p->Wp;
    hipLaunchKernelGGL(depth_regress_bwd_kernel, dim3(cdiv(plane, DR_PIX)), dim3(DR_PIX * DR_DL), 0, (hipStream_t)stream, *bp);
    return check_launch("depth_regress_bwd");
}

}  // extern "C"
