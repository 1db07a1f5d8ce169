// K3+K4 fast path for the fused render pass: the gather reads CHANNEL-LAST copies of its sources and computes its
// own sample coordinates from (ray, depth), so nothing per-sample is read except z.
//
//   volumes  [8,D,h,w]  -> [D,h,w,8]      one 32-byte voxel = two float4 loads (was 8 scattered dwords)
//   images + image features [V,3,H,W] + [V,8,H,W] -> [V,H,W,12] = (r,g,b,f0..f7,0): one 48-byte pixel
//   confidence stays [H,W]
// A trilinear footprint is 8 voxels in 4 x-adjacent pairs: 4 x 64 contiguous bytes instead of 64 scattered
// 4-byte reads.  Semantics are those of gather.hip (grid_sample restated; reference lines cited there).
#include "common.h"
#include "gather_cl_device.h"

namespace ucnerf {

// ------------------------------------------------------------------------------------------------ repack
// One launch for all four sources (blockIdx.y = source; blocks past a source's size exit): the three volumes and the
// image stack are 5-13 us of copying each, so four launches were mostly launch latency.
struct RepackArgs {
    const float* vol[3]; float4* vol_dst[3]; size_t n_vox[3];
    const float* imgs; const float* feat; float4* img_dst; int V; size_t hw;
};

// two floats -> one dword of two bf16 (round to nearest even), first value in the low half
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
    const bf2_ v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// S16: the copies hold bf16 (SURVEY.md 8 configs[4] "bf16 features"): a voxel is 16 bytes, a pixel 24 -- half the bytes of every corner
template <bool S16>
__global__ void __launch_bounds__(256) repack_sources_kernel(RepackArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (k < 3) {
        const size_t n_vox = a.n_vox[k];
        if (i >= n_vox) return;
        const float* __restrict__ src = a.vol[k];
        float c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = src[(size_t)q * n_vox + i];
        if (S16) {
            reinterpret_cast<uint4*>(a.vol_dst[k])[i] = make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
        } else {
            a.vol_dst[k][2 * i] = make_float4(c[0], c[1], c[2], c[3]);
            a.vol_dst[k][2 * i + 1] = make_float4(c[4], c[5], c[6], c[7]);
        }
    } else {
        const size_t hw = a.hw;
        if (i >= (size_t)a.V * hw) return;
        const size_t v = i / hw, px = i % hw;
        const float* im = a.imgs + v * 3 * hw + px;
        const float* ft = a.feat + v * 8 * hw + px;
        if (S16) {
            uint2* d = reinterpret_cast<uint2*>(a.img_dst) + 3 * i;         // 24 bytes per pixel
            d[0] = make_uint2(pack_bf16x2(im[0], im[hw]), pack_bf16x2(im[2 * hw], ft[0]));
            d[1] = make_uint2(pack_bf16x2(ft[hw], ft[2 * hw]), pack_bf16x2(ft[3 * hw], ft[4 * hw]));
            d[2] = make_uint2(pack_bf16x2(ft[5 * hw], ft[6 * hw]), pack_bf16x2(ft[7 * hw], 0.f));
        } else {
            a.img_dst[3 * i] = make_float4(im[0], im[hw], im[2 * hw], ft[0]);
            a.img_dst[3 * i + 1] = make_float4(ft[hw], ft[2 * hw], ft[3 * hw], ft[4 * hw]);
            a.img_dst[3 * i + 2] = make_float4(ft[5 * hw], ft[6 * hw], ft[7 * hw], 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ gather
struct GatherClArgs {
    int n, S, V, H, W;
    int vol_d[3], vol_h[3], vol_w[3];
    const float4* vol[3];      // channel-last volumes
    const float* conf;
    const float4* img;         // [V,H,W,12]
    const float* rays_o;       // [3]
    const float* rays_d;       // [n,3]
    const float* z;            // [n,S]
    const float* near_far;     // [n,6] or NULL
    float near, far;
    float w2c_ref[12], K_ref[9];
    const float* w2cs;         // [V,12] device (uniform per block -> scalar loads)
    const float* Ks;           // [V,9]
    float* feats;              // tiled [ceil(M/32)][F][32], or row-major [M][F] (tiled = 0: the training forward keeps them)
    int tiled;
    float* ndc;                // [M,3] (x, y, scene-normalised z) for the MLP's positional encoding, or NULL
    float* u_out;              // [M] 1 - sampled confidence (network/models.py:149), or NULL
    // GIVEN coordinates (rendering() hands over rays_pts / rays_ndc, network/renderer.py:215-255) instead of (ray, depth)
    const float* pts_in;       // [M,3] world points
    const float* ndc_in[3];    // [M,3] stage coordinates
    unsigned M, div_m, div_sh; // n * S; idx / S = umulhi(idx, div_m) >> div_sh for idx < 2^31 (S >= 2)
};

struct SampleIn { unsigned r; float z, dx, dy, dz; };        // what a sample reads before it can compute anything

__device__ __forceinline__ SampleIn sample_in(const GatherClArgs& a, unsigned idx) {
    SampleIn s;
    s.r = a.S == 1 ? idx : (__umulhi(idx, a.div_m) >> a.div_sh);                     // idx / S (host-side magic, idx < 2^31)
    s.z = a.z[idx];
    s.dx = a.rays_d[3 * s.r]; s.dy = a.rays_d[3 * s.r + 1]; s.dz = a.rays_d[3 * s.r + 2];
    return s;
}

template <bool TILED, bool GIVEN, bool S16>
__device__ __forceinline__ void gather_cl_unit(const GatherClArgs& a, unsigned idx, int unit, const SampleIn& in) {
    constexpr unsigned VOX = S16 ? 16u : 32u, PIX = S16 ? 24u : 48u;     // bytes per voxel / pixel of the channel-last copies
    const int F = 24 + 12 * a.V + 1;
    const unsigned r = in.r;
    const float z = in.z;
    float x, y, w;                                                          // world point
    if (GIVEN) {
        if (unit >= 4) { const float* q = a.pts_in + 3 * (size_t)idx; x = q[0]; y = q[1]; w = q[2]; }
        else x = y = w = 0.f;
    } else { x = a.rays_o[0] + z * in.dx; y = a.rays_o[1] + z * in.dy; w = a.rays_o[2] + z * in.dz; }
    constexpr int fs = TILED ? 32 : 1;                                      // feature f at out[f * fs]
    // Tiled rows are full 128-byte lines written once and read once by the MLP: stream them past the L2 so that the sources
    // stay there (-19 us per step).  Row-major rows are completed piecewise by one thread and want the L2.
#define PUT(IDX, VAL) do { if (TILED) __builtin_nontemporal_store((float)(VAL), &out[(IDX)]); else out[(IDX)] = (VAL); } while (0)
    float* out = TILED ? a.feats + ((size_t)(idx >> 5) * F) * 32 + (idx & 31) : a.feats + (size_t)idx * F;
    if (unit < 4) {
        float u, v, qz = 0.f, zn_given = 0.f;
        if (GIVEN) {                                                        // unit 3 (confidence) samples with the stage-3 grid
            const float* g = a.ndc_in[unit == 3 ? 2 : unit] + 3 * (size_t)idx;
            u = g[0]; v = g[1]; zn_given = g[2];
        } else {
            float qx, qy;
            project_cl(a.w2c_ref, a.K_ref, x, y, w, &qx, &qy, &qz);
            u = (qx / qz + 0.0f) / (float)(a.W - 1); v = (qy / qz + 0.0f) / (float)(a.H - 1);
        }
        if (unit == 3) {
            if (!GIVEN && a.ndc) {
                a.ndc[3 * (size_t)idx] = u; a.ndc[3 * (size_t)idx + 1] = v; a.ndc[3 * (size_t)idx + 2] = (qz - a.near) / (a.far - a.near);
            }
            const LerpCl ax = axis_cl(u * 2.f - 1.0f, a.W, false), ay = axis_cl(v * 2.f - 1.0f, a.H, false);
            const float* c = a.conf;
            float acc = c[(size_t)ay.i0 * a.W + ax.i0] * (ay.w0 * ax.w0);
            acc += c[(size_t)ay.i0 * a.W + ax.i1] * (ay.w0 * ax.w1);
            acc += c[(size_t)ay.i1 * a.W + ax.i0] * (ay.w1 * ax.w0);
            acc += c[(size_t)ay.i1 * a.W + ax.i1] * (ay.w1 * ax.w1);
            PUT((F - 1) * fs, acc);
            if (a.u_out) a.u_out[idx] = 1.f - acc;
            return;
        }
        float zn = zn_given;
        if (!GIVEN) {
            float nk = a.near, fk = a.far;
            if (a.near_far) { nk = a.near_far[6 * (size_t)r + 2 * unit]; fk = a.near_far[6 * (size_t)r + 2 * unit + 1]; }
            zn = (qz - nk) / (fk - nk);
        }
        const int D = a.vol_d[unit], hh = a.vol_h[unit], ww = a.vol_w[unit];
        const LerpCl ax = axis_cl(u * 2.f - 1.0f, ww, false), ay = axis_cl(v * 2.f - 1.0f, hh, false),
                     az = axis_cl(zn * 2.f - 1.0f, D, false);
        const char* vol = (const char*)a.vol[unit];
        // byte offsets of the four (z, y) rows at x0, and the step to x1 (0 at the clamped border)
        const unsigned o00 = (unsigned)((az.i0 * hh + ay.i0) * ww + ax.i0) * VOX, o01 = (unsigned)((az.i0 * hh + ay.i1) * ww + ax.i0) * VOX,
                       o10 = (unsigned)((az.i1 * hh + ay.i0) * ww + ax.i0) * VOX, o11 = (unsigned)((az.i1 * hh + ay.i1) * ww + ax.i0) * VOX;
        const unsigned dx = (unsigned)(ax.i1 - ax.i0) * VOX;
        const float w00 = az.w0 * ay.w0, w01 = az.w0 * ay.w1, w10 = az.w1 * ay.w0, w11 = az.w1 * ay.w1;
        gf2 lo[2] = {{0, 0}, {0, 0}}, hi[2] = {{0, 0}, {0, 0}};
        // same accumulation order as gather.hip: (z0,y0), (z0,y1), (z1,y0), (z1,y1), x0 before x1
#define CORNER8(O, WT) { const float w__ = (WT);                                                                      \
        if (S16) { const float4 r_ = ld16(vol, (O)); FMA4(lo, bf16x4_lo(r_), w__) FMA4(hi, bf16x4_hi(r_), w__) }          \
        else { FMA4(lo, ld16(vol, (O)), w__) FMA4(hi, ld16(vol, (O) + 16u), w__) } }
        CORNER8(o00, w00 * ax.w0) CORNER8(o00 + dx, w00 * ax.w1)
        CORNER8(o01, w01 * ax.w0) CORNER8(o01 + dx, w01 * ax.w1)
        CORNER8(o10, w10 * ax.w0) CORNER8(o10 + dx, w10 * ax.w1)
        CORNER8(o11, w11 * ax.w0) CORNER8(o11 + dx, w11 * ax.w1)
#undef CORNER8
        PUT((8 * unit + 0) * fs, lo[0].x); PUT((8 * unit + 1) * fs, lo[0].y); PUT((8 * unit + 2) * fs, lo[1].x); PUT((8 * unit + 3) * fs, lo[1].y);
        PUT((8 * unit + 4) * fs, hi[0].x); PUT((8 * unit + 5) * fs, hi[0].y); PUT((8 * unit + 6) * fs, hi[1].x); PUT((8 * unit + 7) * fs, hi[1].y);
    } else {
        const int vi = unit - 4;
        float qx, qy, qz;
        project_cl(a.w2cs + 12 * vi, a.Ks + 9 * vi, x, y, w, &qx, &qy, &qz);
        const float gx = (qx / qz + 0.0f) / (float)(a.W - 1) * 2.0f - 1.0f, gy = (qy / qz + 0.0f) / (float)(a.H - 1) * 2.0f - 1.0f;
        const LerpCl ax = axis_cl(gx, a.W, true), ay = axis_cl(gy, a.H, true);
        const char* img = (const char*)a.img + (size_t)vi * a.H * a.W * PIX;
        const unsigned p00 = (unsigned)(ay.i0 * a.W + ax.i0) * PIX, p10 = (unsigned)(ay.i1 * a.W + ax.i0) * PIX;
        const unsigned dx = (unsigned)(ax.i1 - ax.i0) * PIX;
        const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
        gf2 c0[2] = {{0, 0}, {0, 0}}, c1[2] = {{0, 0}, {0, 0}}, c2[2] = {{0, 0}, {0, 0}};      // (r g b f0) (f1..f4) (f5 f6 f7 -)
#define CORNER12(P, WT) { const float w__ = (WT);                                                                                        \
        if (S16) { const float4 r_ = ld16(img, (P)); const float2 t_ = ld8(img, (P) + 16u);                                              \
                   FMA4(c0, bf16x4_lo(r_), w__) FMA4(c1, bf16x4_hi(r_), w__) FMA4(c2, bf16x4_lo(make_float4(t_.x, t_.y, 0.f, 0.f)), w__) } \
        else { FMA4(c0, ld16(img, (P)), w__) FMA4(c1, ld16(img, (P) + 16u), w__) FMA4(c2, ld16(img, (P) + 32u), w__) } }
        CORNER12(p00, w00) CORNER12(p00 + dx, w01) CORNER12(p10, w10) CORNER12(p10 + dx, w11)
#undef CORNER12
        PUT((24 + 4 * vi) * fs, c0[0].x);
        PUT((24 + 4 * vi + 1) * fs, c0[0].y);
        PUT((24 + 4 * vi + 2) * fs, c0[1].x);
        PUT((24 + 4 * vi + 3) * fs, (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.f : 0.f);
        const float f8[8] = {c0[1].y, c1[0].x, c1[0].y, c1[1].x, c1[1].y, c2[0].x, c2[0].y, c2[1].x};
#pragma unroll
        for (int c = 0; c < 8; ++c) PUT((24 + 4 * a.V + 8 * vi + c) * fs, f8[c]);
    }
#undef PUT
}

// Tiled output (the inference path): grid.y = unit, every (tile, feature) row is one full 128-byte line whoever writes it.
// Row-major output (the training forward keeps the features): a sample's 388-byte row shares cache lines with its
// neighbours', and units running as separate sweeps leave every line partially written when it is evicted (measured:
// 142 MB of HBM writes for 51 MB of features) -- there one thread walks all units so that the row completes in L2.
#ifndef UCNERF_GATHER_WAVES
#define UCNERF_GATHER_WAVES 1     // min waves per SIMD asked of the compiler (8 = 64 VGPRs: measured no faster than the 7 it gets by itself)
#endif
template <bool TILED, bool GIVEN, bool S16 = false>
__global__ void __launch_bounds__(256, TILED ? UCNERF_GATHER_WAVES : 1) feat_gather_cl_kernel(GatherClArgs a) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= a.M) return;
    SampleIn in = {0u, 0.f, 0.f, 0.f, 0.f};
    if (!GIVEN) in = sample_in(a, idx);
    if (TILED) gather_cl_unit<true, GIVEN, S16>(a, idx, blockIdx.y, in);
    else
        for (int unit = 0; unit < 4 + a.V; ++unit) gather_cl_unit<false, GIVEN, S16>(a, idx, unit, in);
}


// ------------------------------------------------------------------------------------------------ gather with corner reuse
// The per-sample kernel above is bound by the bytes its corner reads pull through the CU's vector-memory path (2.1 KB per
// sample, TA busy 0.83) although consecutive samples of a ray hardly move: a ray crosses a source view along a short epipolar
// segment (~0.06 px per sample at the bench's poses) and the reference frustum along one (x, y) column.  Here a thread owns a
// RUN of four consecutive samples, works out their footprints first and then sweeps the sources one 16-byte channel group at
// a time, keeping the previous sample's corners in registers: it reloads everything when (x0, y0) moved, one z-plane when the
// depth index advanced by one, nothing otherwise.  Four consecutive samples are 16 contiguous bytes of every tiled feature
// row, so each feature leaves as one float4 store per thread (eight threads complete a 128-byte line).  blockIdx.y = source view.
// Arithmetic (projection, weights, accumulation order per channel) is the per-sample kernel's: the features are bit-identical.
// The four units of the reference frustum (three volumes + confidence) stay on the per-sample kernel: the same scheme for them
// (one projection per sample, z-plane sliding) was built and needs ~170 live registers per thread -- hipcc spills 350-600 bytes
// per lane at every occupancy tried, and the source views carry 60 % of the gathered bytes anyway.
constexpr int RUN = 4;
constexpr int RUN_THREADS = 256;
typedef float gf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gf4 ld16v(const char* base, unsigned off) { return *(const gf4*)(base + off); }
#define FMA4V(O, A, WT)                                                                          \
    { const gf4 a_ = (A); const gf2 w_ = {(WT), (WT)};                                           \
      O[0] = __builtin_elementwise_fma(a_.xy, w_, O[0]);                                         \
      O[1] = __builtin_elementwise_fma(a_.zw, w_, O[1]); }

template <bool GIVEN>
__global__ void __launch_bounds__(RUN_THREADS, 4) feat_gather_run_kernel(GatherClArgs a) {
    const unsigned base = (blockIdx.x * RUN_THREADS + threadIdx.x) * RUN;
    if (base >= a.M) return;
    const int F = 24 + 12 * a.V + 1;
    float* row = a.feats + ((size_t)(base >> 5) * F) * 32 + (base & 31);       // feature f of samples base..base+3 at row[f * 32 .. +3]
#define PUT4(FEAT, O) __builtin_nontemporal_store((gf4){(O)[0], (O)[1], (O)[2], (O)[3]}, reinterpret_cast<gf4*>(row + (size_t)(FEAT) * 32));
    {
        const int vi = blockIdx.y;
        const char* img = (const char*)(a.img + (size_t)vi * a.H * a.W * 3);
        unsigned p00[RUN], p10[RUN], dx[RUN];
        float w00[RUN], w01[RUN], w10[RUN], w11[RUN], mask[RUN];
#pragma unroll
        for (int j = 0; j < RUN; ++j) {
            const unsigned idx = min(base + j, a.M - 1);
            float x, y, w;
            if (GIVEN) { const float* q = a.pts_in + 3 * (size_t)idx; x = q[0]; y = q[1]; w = q[2]; }
            else { const SampleIn in = sample_in(a, idx); x = a.rays_o[0] + in.z * in.dx; y = a.rays_o[1] + in.z * in.dy; w = a.rays_o[2] + in.z * in.dz; }
            float qx, qy, qz;
            project_cl(a.w2cs + 12 * vi, a.Ks + 9 * vi, x, y, w, &qx, &qy, &qz);
            const float gx = (qx / qz + 0.0f) / (float)(a.W - 1) * 2.0f - 1.0f, gy = (qy / qz + 0.0f) / (float)(a.H - 1) * 2.0f - 1.0f;
            const LerpCl ax = axis_cl(gx, a.W, true), ay = axis_cl(gy, a.H, true);
            p00[j] = (unsigned)(ay.i0 * a.W + ax.i0) * 48u; p10[j] = (unsigned)(ay.i1 * a.W + ax.i0) * 48u;
            dx[j] = (unsigned)(ax.i1 - ax.i0) * 48u;
            w00[j] = ay.w0 * ax.w0; w01[j] = ay.w0 * ax.w1; w10[j] = ay.w1 * ax.w0; w11[j] = ay.w1 * ax.w1;
            mask[j] = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.f : 0.f;
        }
        const int fc = 24 + 4 * vi, fb = 24 + 4 * a.V + 8 * vi;
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {            // (r g b f0) (f1 f2 f3 f4) (f5 f6 f7 -)
            const unsigned co = 16u * comp;
            gf4 p[4];
            float out[4][RUN];
#pragma unroll
            for (int j = 0; j < RUN; ++j) {
                if (!(j > 0 && p00[j] == p00[j - 1])) {   // (p00 fixes x0, y0 and with them x1, y1)
                    p[0] = ld16v(img, p00[j] + co); p[1] = ld16v(img, p00[j] + dx[j] + co);
                    p[2] = ld16v(img, p10[j] + co); p[3] = ld16v(img, p10[j] + dx[j] + co);
                }
                gf2 acc[2] = {{0, 0}, {0, 0}};
                FMA4V(acc, p[0], w00[j]) FMA4V(acc, p[1], w01[j]) FMA4V(acc, p[2], w10[j]) FMA4V(acc, p[3], w11[j])
                out[0][j] = acc[0].x; out[1][j] = acc[0].y; out[2][j] = acc[1].x; out[3][j] = acc[1].y;
            }
            if (comp == 0) { PUT4(fc + 0, out[0]) PUT4(fc + 1, out[1]) PUT4(fc + 2, out[2]) PUT4(fc + 3, mask) PUT4(fb + 0, out[3]) }
            else if (comp == 1) { PUT4(fb + 1, out[0]) PUT4(fb + 2, out[1]) PUT4(fb + 3, out[2]) PUT4(fb + 4, out[3]) }
            else { PUT4(fb + 5, out[0]) PUT4(fb + 6, out[1]) PUT4(fb + 7, out[2]) }
        }
    }
#undef PUT4
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int64_t ucnerf_gather_repack_floats(const ucnerf_render_params* p) {
    if (!p) return fail(UCNERF_EINVAL, "gather_repack_floats: null params");
    int64_t n = 0;
    for (int k = 0; k < 3; ++k) n += 8ll * p->vol_d[k] * p->vol_h[k] * p->vol_w[k];
    n += 12ll * p->cfg.n_src * p->H * p->W;
    return p->sources_cl_bf16 ? (n + 1) / 2 : n;           // (bf16 copies: two values per float; every volume is a multiple of 16 bytes)
}

int ucnerf_gather_repack(const ucnerf_render_params* p, float* dst, void* stream) {
    UCNERF_REQUIRE(p && dst, "gather_repack: null pointer");
    UCNERF_REQUIRE(p->vol[0] && p->vol[1] && p->vol[2] && p->imgs && p->img_feat, "gather_repack: null source");
    UCNERF_REQUIRE(((uintptr_t)dst & 15) == 0, "gather_repack: destination must be 16-byte aligned");
    RepackArgs a;
    float* o = dst;
    size_t n_max = 0;
    for (int k = 0; k < 3; ++k) {
        a.n_vox[k] = (size_t)p->vol_d[k] * p->vol_h[k] * p->vol_w[k];
        a.vol[k] = p->vol[k]; a.vol_dst[k] = (float4*)o;
        o += (p->sources_cl_bf16 ? 4 : 8) * a.n_vox[k];
        if (a.n_vox[k] > n_max) n_max = a.n_vox[k];
    }
    a.imgs = p->imgs; a.feat = p->img_feat; a.img_dst = (float4*)o; a.V = p->cfg.n_src; a.hw = (size_t)p->H * p->W;
    if (a.hw * a.V > n_max) n_max = a.hw * a.V;
    if (p->sources_cl_bf16) hipLaunchKernelGGL(repack_sources_kernel<true>, dim3(cdiv(n_max, 256), 4), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(repack_sources_kernel<false>, dim3(cdiv(n_max, 256), 4), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("gather_repack");
}

}  // extern "C"

namespace ucnerf {

// called by render.hip: gather (+ ndc) for one pass from the repacked sources
int launch_gather_cl(const ucnerf_render_params* p, const float* repacked, float* feats, int tiled, float* ndc, hipStream_t st) {
    const bool given = p->pts_in != nullptr;
    GatherClArgs a;
    memset(&a, 0, sizeof(a));
    a.n = p->n; a.S = p->S; a.V = p->cfg.n_src; a.H = p->H; a.W = p->W;
    const float* o = repacked;
    for (int k = 0; k < 3; ++k) {
        a.vol_d[k] = p->vol_d[k]; a.vol_h[k] = p->vol_h[k]; a.vol_w[k] = p->vol_w[k];
        a.vol[k] = (const float4*)o;
        o += (p->sources_cl_bf16 ? 4ull : 8ull) * p->vol_d[k] * p->vol_h[k] * p->vol_w[k];
    }
    a.img = (const float4*)o;
    a.conf = p->conf; a.rays_o = p->rays_o; a.rays_d = p->rays_d; a.z = p->z; a.near_far = p->near_far;
    a.near = p->near; a.far = p->far;
    memcpy(a.w2c_ref, p->w2c_ref, sizeof(a.w2c_ref));
    memcpy(a.K_ref, p->K_ref, sizeof(a.K_ref));
    a.w2cs = p->w2cs; a.Ks = p->intrinsics;
    a.feats = feats; a.tiled = tiled; a.ndc = ndc; a.u_out = p->u_sampled;
    a.pts_in = p->pts_in; a.ndc_in[0] = p->ndc1_in; a.ndc_in[1] = p->ndc2_in; a.ndc_in[2] = p->ndc3_in;
    UCNERF_REQUIRE(a.V >= 1 && a.V <= 8, "gather_cl: V = %d outside 1..8", a.V);
    const long long M = (long long)p->n * p->S;
    UCNERF_REQUIRE(M < (1ll << 31), "gather_cl: %lld samples in one pass (limit 2^31 - 1)", M);
    a.M = (unsigned)M;
    {   // magic for idx / S: l = ceil(log2 S), m = ceil(2^(31+l) / S) < 2^32, exact for idx < 2^31
        unsigned l = 1;
        while ((1u << l) < (unsigned)p->S) ++l;
        a.div_m = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)p->S - 1) / (unsigned)p->S);
        a.div_sh = l - 1;
    }
    const dim3 grid(cdiv(M, 256), tiled ? 4 + a.V : 1), block(256);
#ifndef UCNERF_GATHER_RUN
#define UCNERF_GATHER_RUN 0       // tiled layout: 1 = source views on the corner-reuse kernel (a run of four samples per thread), 0 = one sample per
                                  // thread for every unit.  Measured (profiles/r02_gather_experiments.md): 3x fewer corner loads, but 36.5 us for the
                                  // six views against ~32 us on the per-sample kernel at 7 waves per SIMD -- the default stays 0
#endif
    if (p->sources_cl_bf16) {
        UCNERF_REQUIRE(!UCNERF_GATHER_RUN, "gather_cl: the corner-reuse variant reads fp32 copies only");
        if (tiled && !given) hipLaunchKernelGGL((feat_gather_cl_kernel<true, false, true>), grid, block, 0, st, a);
        else if (tiled) hipLaunchKernelGGL((feat_gather_cl_kernel<true, true, true>), grid, block, 0, st, a);
        else if (!given) hipLaunchKernelGGL((feat_gather_cl_kernel<false, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((feat_gather_cl_kernel<false, true, true>), grid, block, 0, st, a);
    }
    else if (tiled && UCNERF_GATHER_RUN) {
        const dim3 grid_ref(cdiv(M, 256), 4), grid_r(cdiv(cdiv(M, RUN), RUN_THREADS), a.V), block_r(RUN_THREADS);
        if (!given) {
            hipLaunchKernelGGL((feat_gather_cl_kernel<true, false>), grid_ref, block, 0, st, a);        // volumes + confidence (+ ndc, u)
            hipLaunchKernelGGL((feat_gather_run_kernel<false>), grid_r, block_r, 0, st, a);            // source views
        } else {
            hipLaunchKernelGGL((feat_gather_cl_kernel<true, true>), grid_ref, block, 0, st, a);
            hipLaunchKernelGGL((feat_gather_run_kernel<true>), grid_r, block_r, 0, st, a);
        }
    }
    else if (tiled && !given) hipLaunchKernelGGL((feat_gather_cl_kernel<true, false>), grid, block, 0, st, a);
    else if (tiled) hipLaunchKernelGGL((feat_gather_cl_kernel<true, true>), grid, block, 0, st, a);
    else if (!given) hipLaunchKernelGGL((feat_gather_cl_kernel<false, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((feat_gather_cl_kernel<false, true>), grid, block, 0, st, a);
    return check_launch("feat_gather_cl");
}

const char* build_flags_gather_cl() { return "gather_cl: " UCNERF_FLAG(UCNERF_GATHER_RUN) UCNERF_FLAG(UCNERF_GATHER_WAVES); }

}  // namespace ucnerf
