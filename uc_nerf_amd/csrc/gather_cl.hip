// K3+K4 fast path for the fused render pass: the gather reads CHANNEL-LAST copies of its sources and computes its
// own sample coordinates from (ray, depth), so nothing per-sample is read except z.
//
//   volumes         [8,D,h,w] -> [D,h,w,8]      one 32-byte voxel = two float4 loads (was 8 scattered dwords)
//   image features  [V,8,H,W] -> [V,H,W,8]      one 32-byte pixel
//   images          [V,3,H,W] -> [V,H,W,3 | 4]  one 12-byte load (the repack pads to 16)
//   confidence stays [H,W]
// Every source is its own array (ABI v5, ucnerf_cl_sources): the memory of torch's channels_last / channels_last_3d tensors, read in place when
// the producer writes them, built by repack_sources_kernel otherwise.
// A trilinear footprint is 8 voxels in 4 x-adjacent pairs: 4 x 64 contiguous bytes instead of 64 scattered
// 4-byte reads.  Semantics are those of gather.hip (grid_sample restated; reference lines cited there).
#include "common.h"
#include "gather_cl_device.h"

namespace ucnerf {

// ------------------------------------------------------------------------------------------------ repack
// One launch for all four sources (blockIdx.y = source; blocks past a source's size exit): the three volumes and the
// image stack are 5-13 us of copying each, so four launches were mostly launch latency.
struct RepackArgs {
    const float* vol[3]; float4* vol_dst[3]; size_t n_vox[3];       // (a source with a null destination is skipped: it is read in place)
    const float* feat; float4* feat_dst; const float* imgs; float4* col_dst; int V; size_t hw;
};

// two floats -> one dword of two bf16 (round to nearest even), first value in the low half
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
    const bf2_ v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// S16: the copies hold bf16 (SURVEY.md 8 configs[4] "bf16 features"): a voxel / feature pixel is 16 bytes, a colour 8 -- half the bytes of every corner
// blockIdx.y: 0..2 volumes, 3 image features, 4 colours
template <bool S16>
__global__ void __launch_bounds__(256) repack_sources_kernel(RepackArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (k < 4) {
        const size_t n_vox = k < 3 ? a.n_vox[k] : a.hw;              // positions per channel plane
        const size_t n_all = k < 3 ? n_vox : (size_t)a.V * a.hw;
        float4* dst = k < 3 ? a.vol_dst[k] : a.feat_dst;
        if (!dst || i >= n_all) return;
        const float* __restrict__ src = k < 3 ? a.vol[k] + i : a.feat + (i / n_vox) * 8 * n_vox + i % n_vox;
        float c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = src[(size_t)q * n_vox];
        if (S16) {
            reinterpret_cast<uint4*>(dst)[i] = make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
        } else {
            dst[2 * i] = make_float4(c[0], c[1], c[2], c[3]);
            dst[2 * i + 1] = make_float4(c[4], c[5], c[6], c[7]);
        }
    } else {
        const size_t hw = a.hw;
        if (!a.col_dst || i >= (size_t)a.V * hw) return;
        const float* im = a.imgs + (i / hw) * 3 * hw + i % hw;
        if (S16) reinterpret_cast<uint2*>(a.col_dst)[i] = make_uint2(pack_bf16x2(im[0], im[hw]), pack_bf16x2(im[2 * hw], 0.f));
        else a.col_dst[i] = make_float4(im[0], im[hw], im[2 * hw], 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ gather
struct GatherClArgs {
    int n, S, V, H, W;
    int vol_d[3], vol_h[3], vol_w[3];
    const float4* vol[3];      // channel-last volumes [D,h,w,8]
    const float* conf;
    const char* feat;          // [V,H,W,8] image features
    const char* col;           // [V,H,W,col_px / 4] colours (fp32: 12 or 16 bytes per pixel; bf16: 8)
    unsigned col_px;
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
    constexpr unsigned VOX = S16 ? 16u : 32u;                             // bytes per voxel / feature pixel of the channel-last arrays
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
        // pixel indices (view included: the V maps of a source are one array) of the two rows at x0, and the step to x1 (0 at the clamped border)
        const unsigned i00 = (unsigned)((vi * a.H + ay.i0) * a.W + ax.i0), i10 = (unsigned)((vi * a.H + ay.i1) * a.W + ax.i0);
        const unsigned dx = (unsigned)(ax.i1 - ax.i0);
        const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
        gf2 rg = {0, 0}, c1[2] = {{0, 0}, {0, 0}}, c2[2] = {{0, 0}, {0, 0}};      // (r g) b | (f0..f3) (f4..f7)
        float bl = 0.f;
        // same accumulation per channel as ever: corners (y0,x0) (y0,x1) (y1,x0) (y1,x1), one fused multiply-add each
#define CORNER11(I, WT) { const float w__ = (WT); const unsigned i__ = (I);                                                              \
        if (S16) { const float4 r_ = ld16(a.feat, i__ * VOX); const float2 t_ = ld8(a.col, i__ * 8u);                                    \
                   const float4 k_ = bf16x4_lo(make_float4(t_.x, t_.y, 0.f, 0.f));                                                      \
                   FMA4(c1, bf16x4_lo(r_), w__) FMA4(c2, bf16x4_hi(r_), w__)                                                             \
                   rg = __builtin_elementwise_fma((gf2){k_.x, k_.y}, (gf2){w__, w__}, rg); bl = __builtin_fmaf(k_.z, w__, bl); }         \
        else { const gf3u k_ = ld12(a.col, __umul24(i__, a.col_px));                                                                     \
               FMA4(c1, ld16(a.feat, i__ * VOX), w__) FMA4(c2, ld16(a.feat, i__ * VOX + 16u), w__)                                       \
               rg = __builtin_elementwise_fma((gf2){k_.x, k_.y}, (gf2){w__, w__}, rg); bl = __builtin_fmaf(k_.z, w__, bl); } }
        CORNER11(i00, w00) CORNER11(i00 + dx, w01) CORNER11(i10, w10) CORNER11(i10 + dx, w11)
#undef CORNER11
        PUT((24 + 4 * vi) * fs, rg.x);
        PUT((24 + 4 * vi + 1) * fs, rg.y);
        PUT((24 + 4 * vi + 2) * fs, bl);
        PUT((24 + 4 * vi + 3) * fs, (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.f : 0.f);
        const float f8[8] = {c1[0].x, c1[0].y, c1[1].x, c1[1].y, c2[0].x, c2[0].y, c2[1].x, c2[1].y};
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


}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

// floats of `dst` a source takes when it is repacked (0 when it is handed over in place)
static void repack_sizes(const ucnerf_render_params* p, size_t n[5]) {
    const size_t half = p->cl.bf16 ? 2 : 1;
    for (int k = 0; k < 3; ++k) n[k] = p->cl.vol[k] ? 0 : (8 * (size_t)p->vol_d[k] * p->vol_h[k] * p->vol_w[k] / half + 3) / 4 * 4;
    const size_t px = (size_t)p->cfg.n_src * p->H * p->W;
    n[3] = p->cl.img_feat ? 0 : (8 * px / half + 3) / 4 * 4;
    n[4] = p->cl.imgs ? 0 : (4 * px / half + 3) / 4 * 4;
}

int64_t ucnerf_gather_repack_floats(const ucnerf_render_params* p) {
    if (!p) return fail(UCNERF_EINVAL, "gather_repack_floats: null params");
    size_t n[5];
    repack_sizes(p, n);
    return (int64_t)(n[0] + n[1] + n[2] + n[3] + n[4]);
}

int ucnerf_gather_repack(const ucnerf_render_params* p, float* dst, ucnerf_cl_sources* out, void* stream) {
    UCNERF_REQUIRE(p && out, "gather_repack: null pointer");
    size_t n[5];
    repack_sizes(p, n);
    const bool any = n[0] + n[1] + n[2] + n[3] + n[4] > 0;
    UCNERF_REQUIRE(!any || dst, "gather_repack: null destination");
    UCNERF_REQUIRE(((uintptr_t)dst & 15) == 0, "gather_repack: destination must be 16-byte aligned");
    UCNERF_REQUIRE(!p->cl.bf16 || !p->cl.imgs || p->cl.rgb_stride == 4, "gather_repack: bf16 colours handed over in place must be padded to 4 values per pixel");
    RepackArgs a;
    memset(&a, 0, sizeof(a));
    ucnerf_cl_sources o = p->cl;
    o.rgb_stride = p->cl.imgs ? p->cl.rgb_stride : 4;
    float* q = dst;
    size_t n_max = 0;
    for (int k = 0; k < 3; ++k) {
        a.n_vox[k] = (size_t)p->vol_d[k] * p->vol_h[k] * p->vol_w[k];
        if (!n[k]) continue;
        UCNERF_REQUIRE(p->vol[k], "gather_repack: null volume %d", k);
        a.vol[k] = p->vol[k]; a.vol_dst[k] = (float4*)q; o.vol[k] = q; q += n[k];
        if (a.n_vox[k] > n_max) n_max = a.n_vox[k];
    }
    a.V = p->cfg.n_src; a.hw = (size_t)p->H * p->W;
    if (n[3]) { UCNERF_REQUIRE(p->img_feat, "gather_repack: null image features"); a.feat = p->img_feat; a.feat_dst = (float4*)q; o.img_feat = q; q += n[3]; }
    if (n[4]) { UCNERF_REQUIRE(p->imgs, "gather_repack: null images"); a.imgs = p->imgs; a.col_dst = (float4*)q; o.imgs = q; q += n[4]; }
    if ((n[3] || n[4]) && a.hw * a.V > n_max) n_max = a.hw * a.V;
    *out = o;
    if (!any) return UCNERF_OK;
    if (p->cl.bf16) hipLaunchKernelGGL(repack_sources_kernel<true>, dim3(cdiv(n_max, 256), 5), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(repack_sources_kernel<false>, dim3(cdiv(n_max, 256), 5), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("gather_repack");
}

}  // extern "C"

namespace ucnerf {

// The channel-last sources of a pass, validated (all five arrays, alignment, the sizes the kernels' 32-bit offsets assume)
int check_cl_sources(const ucnerf_render_params* p, const char* who) {
    const ucnerf_cl_sources& c = p->cl;
    UCNERF_REQUIRE(c.vol[0] && c.vol[1] && c.vol[2] && c.img_feat && c.imgs, "%s: the channel-last sources must all be given (ucnerf_gather_repack completes them)", who);
    for (int k = 0; k < 3; ++k) {
        UCNERF_REQUIRE(((uintptr_t)c.vol[k] & 15) == 0, "%s: channel-last volume %d must be 16-byte aligned", who, k);
        UCNERF_REQUIRE(32ull * p->vol_d[k] * p->vol_h[k] * p->vol_w[k] < (1ull << 32), "%s: volume %d exceeds 4 GB", who, k);
    }
    UCNERF_REQUIRE(((uintptr_t)c.img_feat & 15) == 0, "%s: channel-last image features must be 16-byte aligned", who);
    UCNERF_REQUIRE(c.bf16 ? c.rgb_stride == 4 : (c.rgb_stride == 3 || c.rgb_stride == 4), "%s: rgb_stride %d (3 or 4; bf16: 4)", who, c.rgb_stride);
    UCNERF_REQUIRE(((uintptr_t)c.imgs & (c.bf16 ? 7 : c.rgb_stride == 4 ? 15 : 3)) == 0, "%s: channel-last images are misaligned", who);
    UCNERF_REQUIRE((unsigned long long)p->cfg.n_src * p->H * p->W < (1ull << 24), "%s: %d x %d x %d pixels (limit 2^24)", who, p->cfg.n_src, p->H, p->W);
    return UCNERF_OK;
}

// called by render.hip: gather (+ ndc) for one pass from the channel-last sources
int launch_gather_cl(const ucnerf_render_params* p, float* feats, int tiled, float* ndc, hipStream_t st) {
    if (int rc = check_cl_sources(p, "gather_cl")) return rc;
    const bool given = p->pts_in != nullptr;
    GatherClArgs a;
    memset(&a, 0, sizeof(a));
    a.n = p->n; a.S = p->S; a.V = p->cfg.n_src; a.H = p->H; a.W = p->W;
    for (int k = 0; k < 3; ++k) {
        a.vol_d[k] = p->vol_d[k]; a.vol_h[k] = p->vol_h[k]; a.vol_w[k] = p->vol_w[k];
        a.vol[k] = (const float4*)p->cl.vol[k];
    }
    a.feat = (const char*)p->cl.img_feat; a.col = (const char*)p->cl.imgs;
    a.col_px = p->cl.bf16 ? 8u : 4u * (unsigned)p->cl.rgb_stride;
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
    if (p->cl.bf16) {
        if (tiled && !given) hipLaunchKernelGGL((feat_gather_cl_kernel<true, false, true>), grid, block, 0, st, a);
        else if (tiled) hipLaunchKernelGGL((feat_gather_cl_kernel<true, true, true>), grid, block, 0, st, a);
        else if (!given) hipLaunchKernelGGL((feat_gather_cl_kernel<false, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((feat_gather_cl_kernel<false, true, true>), grid, block, 0, st, a);
    }
    else if (tiled && !given) hipLaunchKernelGGL((feat_gather_cl_kernel<true, false>), grid, block, 0, st, a);
    else if (tiled) hipLaunchKernelGGL((feat_gather_cl_kernel<true, true>), grid, block, 0, st, a);
    else if (!given) hipLaunchKernelGGL((feat_gather_cl_kernel<false, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((feat_gather_cl_kernel<false, true>), grid, block, 0, st, a);
    return check_launch("feat_gather_cl");
}

const char* build_flags_gather_cl() { return "gather_cl: " UCNERF_FLAG(UCNERF_GATHER_WAVES); }

}  // namespace ucnerf
