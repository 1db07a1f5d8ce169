// The split-bf16 MLP forward of mlp_bf16.hip on the 16x16x32 MFMA shape (v_mfma_f32_16x16x32_bf16).
//
// Why a second shape: the kernel is power-bound, and under the chip's power limit the 16x16x32 instruction sustains
// 1.1-1.3x the FLOP/s of 32x32x16 (scripts/micro/mfma_shapes.hip: 1.9-2.1 vs 1.4-1.7 PFLOP/s with operands re-read from
// LDS, two waves per SIMD).  Same algorithm, same ring / DMA / barrier scheme, same pair-split phases; what changes is
// the register-resident layout:
//   * a wave still owns 32 samples, as TWO column tiles c = 0, 1 of 16: lane l = (g = l >> 4, i = l & 15) works for the
//     samples 16c + i; an accumulator tile acc[rt][c] (16 output rows x 16 samples) is 4 registers per lane holding
//     rows 16 rt + 4 g + r;
//   * a k32-step q contracts 32 input features.  For hidden layers its B fragment for column tile c is built from the
//     accumulators of row tiles 2q and 2q + 1: element j < 4 = feature 32q + 4g + j, j >= 4 = feature 32q + 16 + 4g + j - 4
//     -- again no data movement between a layer's output and the next layer's operand;
//   * a half-step = one k32-step for one PAIR of row tiles (32 rows): 4 KB of weights [hi0 | lo0 | hi1 | lo1], twelve
//     MFMAs (2 row tiles x 2 column tiles x 3 terms, 16 cycles each) = the MFMA time of a half-step of mlp_bf16.hip.
// Phase A of a 128-wide layer = row tiles 0..3 (pairs 0, 1) for all k32-steps, phase B = row tiles 4..7; the epilogue of
// phase A's rows is issued under phase B's MFMAs, that of phase B's under the next layer's phase A.
#include "common.h"
#include "mlp_layout.h"
#include "sincos_cw.h"

#include <cstdlib>
#include <vector>

namespace ucnerf {
namespace s16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define SB0 __builtin_amdgcn_sched_barrier(0)

constexpr int BW = 4;                 // waves per block (one per SIMD; two blocks per CU)
constexpr int SLOT_BYTES = 8192;      // two half-steps
constexpr int HALF_BYTES = 4096;      // [hi0, lo0, hi1, lo1][64 lanes][16 B]
constexpr int NBUF = 4;
constexpr int DMA_PER_SLOT = SLOT_BYTES / 1024 / BW;

struct Layout {
    int v, F, kd32, kc32, halfs;      // half-steps per tile
    int64_t const_off_bytes, total_bytes;
};

inline bool layout(int v, Layout* B) {
    if (v < 1 || v > 8) return false;
    B->v = v; B->F = 24 + 12 * v + 1;
    B->kd32 = (24 + 4 * v + 31) / 32; B->kc32 = (8 * v + 31) / 32;
    // k32-steps x 4 row-tile pairs: bd | L0 (2) | L1..L4 (4 each) | L5 (6) | bc | ft (4) | vc (5)
    B->halfs = 4 * (B->kd32 + 2 + 16 + 6 + B->kc32 + 4 + 5);
    B->const_off_bytes = (int64_t)B->halfs * HALF_BYTES;
    B->total_bytes = B->const_off_bytes + (int64_t)CONST_FLOATS * 4;
    return true;
}

// input feature of element j of lane group g in k32-step q
__host__ __device__ inline int hid_feat(int q, int g, int j) { return 32 * q + 16 * (j >> 2) + 4 * g + (j & 3); }
__host__ __device__ inline int nat_feat(int q, int g, int j) { return 32 * q + 8 * g + j; }
// output feature of register r of row tile rt on lane group g
__host__ __device__ inline int out_feat(int rt, int g, int r) { return 16 * rt + 4 * g + r; }

// point-encoding slot (two k32-steps): argument a = 8g + j; step 0 = sin(a) | x, y ; step 1 = cos(a) | z, pad
inline void pts_slot(int q, int g, int j, int* kind, int* a) {
    const int n = 8 * g + j;
    if (n < 30) { *kind = q; *a = n; }
    else if (n == 30) { *kind = 2; *a = q ? 2 : 0; }
    else if (q == 0) { *kind = 2; *a = 1; }
    else { *kind = 3; *a = 0; }
}
// direction-encoding slot (one k32-step): n = 8g + j: sin 0..11 | cos 0..11 | x, y, z | pad
inline void dir_slot(int g, int j, int* kind, int* a) {
    const int n = 8 * g + j;
    if (n < 12) { *kind = 0; *a = n; }
    else if (n < 24) { *kind = 1; *a = n - 12; }
    else if (n < 27) { *kind = 2; *a = n - 24; }
    else { *kind = 3; *a = 0; }
}

}  // namespace s16

// ------------------------------------------------------------------------------------------------ host: pack index
// Same element format as build_pack_index_bf16 (flat parameter index | part << 30, -1 = zero), half-steps in consumption
// order, then the fp32 constants in THIS kernel's register order: bias blocks [sec][g][rt][r], head blocks [g][rt][r][4] + 4.
int build_pack_index_bf16_s16(const ucnerf_mlp_config* cfg, int32_t* idx) {
    using namespace s16;
    Layout B;
    MlpLayout L;
    if (!layout(cfg->n_src, &B) || !mlp_layout(cfg->n_src, &L)) return -1;
    const int v = B.v, W = MLP_W;
    const int64_t n16 = (int64_t)B.halfs * (HALF_BYTES / 2);
    for (int64_t i = 0; i < n16 + CONST_FLOATS; ++i) idx[i] = -1;
    int64_t hidx = 0;
    auto put_half = [&](const std::vector<int64_t>& row_base, const int (&col)[4][8], int pair) {
        for (int t = 0; t < 2; ++t)
            for (int part = 0; part < 2; ++part)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int c = col[lane >> 4][j];
                        const int64_t e = (((hidx * 4 + t * 2 + part) * 64) + lane) * 8 + j;
                        idx[e] = c < 0 ? -1 : (int32_t)((row_base[16 * (2 * pair + t) + (lane & 15)] + c) | ((int64_t)part << 30));
                    }
        ++hidx;
    };
    auto rows = [&](int64_t base, int K) { std::vector<int64_t> rb(128); for (int n = 0; n < 128; ++n) rb[n] = base + (int64_t)n * K; return rb; };
    auto nat = [&](int q, int K, int (&col)[4][8]) { for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) { const int f = nat_feat(q, g, j); col[g][j] = f < K ? f : -1; } };
    auto hid = [&](int q, int base, int (&col)[4][8]) { for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) col[g][j] = base + hid_feat(q, g, j); };
    auto pts = [&](int q, int base, int (&col)[4][8]) {
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) {
                int kind, a;
                pts_slot(q, g, j, &kind, &a);
                const int c = pe_column(kind, a, 10, cfg->pe_layout);
                col[g][j] = c < 0 ? -1 : base + c;
            }
    };
    auto dir = [&](int base, int (&col)[4][8]) {
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) {
                int kind, a;
                dir_slot(g, j, &kind, &a);
                const int c = pe_column(kind, a, 4, cfg->pe_layout);
                col[g][j] = c < 0 ? -1 : base + c;
            }
    };
    int col[4][8];
    {   // bd, step-major over the four pairs
        const auto rb = rows(L.p_bdw, 24 + 4 * v);
        for (int q = 0; q < B.kd32; ++q) { nat(q, 24 + 4 * v, col); for (int p = 0; p < 4; ++p) put_half(rb, col, p); }
    }
    {   // L0
        const auto rb = rows(L.p_lw[0], MLP_PE_PTS);
        for (int ph = 0; ph < 2; ++ph) for (int q = 0; q < 2; ++q) { pts(q, 0, col); for (int pp = 0; pp < 2; ++pp) put_half(rb, col, 2 * ph + pp); }
    }
    for (int l = 1; l < 5; ++l) {
        const auto rb = rows(L.p_lw[l], W);
        for (int ph = 0; ph < 2; ++ph) for (int q = 0; q < 4; ++q) { hid(q, 0, col); for (int pp = 0; pp < 2; ++pp) put_half(rb, col, 2 * ph + pp); }
    }
    {   // L5 on [pe | h]: k order h0, h1, pe0, pe1, h2, h3
        const auto rb = rows(L.p_lw[5], W + MLP_PE_PTS);
        for (int ph = 0; ph < 2; ++ph)
            for (int s = 0; s < 6; ++s) {
                if (s < 2) hid(s, MLP_PE_PTS, col); else if (s < 4) pts(s - 2, 0, col); else hid(s - 2, MLP_PE_PTS, col);
                for (int pp = 0; pp < 2; ++pp) put_half(rb, col, 2 * ph + pp);
            }
    }
    {   // bc, step-major
        const auto rb = rows(L.p_bcw, 8 * v);
        for (int q = 0; q < B.kc32; ++q) { nat(q, 8 * v, col); for (int p = 0; p < 4; ++p) put_half(rb, col, p); }
    }
    {   // feature_linear
        const auto rb = rows(L.p_fw, W);
        for (int ph = 0; ph < 2; ++ph) for (int q = 0; q < 4; ++q) { hid(q, 0, col); for (int pp = 0; pp < 2; ++pp) put_half(rb, col, 2 * ph + pp); }
    }
    std::vector<int64_t> rvc(128);
    for (int n = 0; n < 64; ++n) { rvc[n] = L.p_vw + (int64_t)n * (W + MLP_PE_DIR); rvc[64 + n] = L.p_vcw + (int64_t)n * (W + MLP_PE_DIR); }
    for (int ph = 0; ph < 2; ++ph)
        for (int s = 0; s < 5; ++s) {
            if (s < 4) hid(s, 0, col); else dir(W, col);
            for (int pp = 0; pp < 2; ++pp) put_half(rvc, col, 2 * ph + pp);
        }
    if (hidx != B.halfs) return -1;

    // constants
    int32_t* cst = idx + n16;
    auto bias = [&](int sec, int64_t p_lo, int64_t p_hi) {     // rows 0..63 from p_lo (or all 128 when p_hi < 0), 64..127 from p_hi
        for (int g = 0; g < 4; ++g)
            for (int rt = 0; rt < 8; ++rt)
                for (int r = 0; r < 4; ++r) {
                    const int n = out_feat(rt, g, r);
                    cst[sec * 128 + g * 32 + rt * 4 + r] = (int32_t)(p_hi < 0 || n < 64 ? p_lo + n : p_hi + n - 64);
                }
    };
    bias(SEC_BD, L.p_bdb, -1); bias(SEC_BC, L.p_bcb, -1); bias(SEC_FT, L.p_fb, -1); bias(SEC_VC, L.p_vb, L.p_vcb);
    for (int l = 0; l < 6; ++l) bias(SEC_L0 + l, L.p_lb[l], -1);
    int32_t* hb = cst + N_SEC * 128;
    int32_t* ha = hb + 516;
    for (int g = 0; g < 4; ++g)
        for (int rt = 0; rt < 8; ++rt)
            for (int r = 0; r < 4; ++r) {
                const int f = out_feat(rt, g, r);
                const int o = ((g * 8 + rt) * 4 + r) * 4;
                for (int c = 0; c < 3; ++c) hb[o + c] = (int32_t)(L.p_crw + c * W + f);        // confi_rgb_linear
                hb[o + 3] = (int32_t)(L.p_a1w + f);                                              // alpha_linear_1
                if (f < 64) for (int c = 0; c < 3; ++c) ha[o + c] = (int32_t)(L.p_rw + c * 64 + f);   // rgb_linear
                else ha[o + 3] = (int32_t)(L.p_aw + f - 64);                                      // alpha_linear
            }
    for (int c = 0; c < 3; ++c) { hb[512 + c] = (int32_t)(L.p_crb + c); ha[512 + c] = (int32_t)(L.p_rb + c); }
    hb[512 + 3] = (int32_t)L.p_a1b;
    ha[512 + 3] = (int32_t)L.p_ab;
    return 0;
}

int64_t bf16_s16_index_count(const ucnerf_mlp_config* cfg) {
    s16::Layout B;
    if (!s16::layout(cfg->n_src, &B)) return -1;
    return (int64_t)B.halfs * (s16::HALF_BYTES / 2) + CONST_FLOATS;
}

int64_t bf16_s16_stream_floats(const ucnerf_mlp_config* cfg) {
    s16::Layout B;
    if (!s16::layout(cfg->n_src, &B)) return -1;
    return B.total_bytes / 4;
}

namespace s16 {

// ------------------------------------------------------------------------------------------------ device helpers
struct Frag { bf16x8 hi, lo; };

// (hi, lo) split: hi = top 16 bits (truncated bf16), lo = bf16_rne(x - hi) -- see mlp_bf16.hip
__device__ __forceinline__ Frag split8(const float (&x)[8]) {
    u32x4 hi;
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[j]), b1 = __builtin_bit_cast(unsigned, x[j + 1]);
        hi[j >> 1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const f32x2 h = {__builtin_bit_cast(float, b0 & 0xffff0000u), __builtin_bit_cast(float, b1 & 0xffff0000u)};
        const f32x2 l = (f32x2){x[j], x[j + 1]} - h;
        f.lo[j] = (__bf16)l.x;
        f.lo[j + 1] = (__bf16)l.y;
    }
    f.hi = __builtin_bit_cast(bf16x8, hi);
    return f;
}

__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
template <class T> __device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }

// fragment of k32-step q for one column tile: rows of row tiles 2q (a0) and 2q + 1 (a1); MODE 0 plain, 1 times m, 2 relu(times m)
template <int MODE>
__device__ __forceinline__ Frag frag_of(const f32x4& a0, const f32x4& a1, const f32x4& m0, const f32x4& m1) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v0 = a0[j], v1 = a1[j];
        if (MODE >= 1) { v0 *= m0[j]; v1 *= m1[j]; }
        t[j] = MODE == 2 ? fmaxf(v0, 0.f) : v0;
        t[4 + j] = MODE == 2 ? fmaxf(v1, 0.f) : v1;
    }
    Frag f = split8(t);
    pin(f.hi); pin(f.lo);
    return f;
}

struct Geom {
    int F, kd32, kc32, f_img, slots, feat_stride;
    int const_off_bytes;
};

struct AF { bf16x8 h0, l0, h1, l1; };

struct Pipe {
    const char* __restrict__ gsrc;
    char* ring;
    unsigned ring_lds;
    const char* buf;
    int wave, gpos, next_src, slots;
};

// (inline asm on purpose: see mlp_bf16.hip)
__device__ __forceinline__ void issue_dma(Pipe& P, int pos) {
    const char* src = P.gsrc + (size_t)P.next_src * SLOT_BYTES;
    const unsigned dst = P.ring_lds + (pos & (NBUF - 1)) * SLOT_BYTES + P.wave * (DMA_PER_SLOT * 1024);
#pragma unroll
    for (int i = 0; i < DMA_PER_SLOT; ++i)
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(src), "s"(dst), "n"(i * 1024) : "memory", "m0");
    P.next_src = P.next_src + 1 == P.slots ? 0 : P.next_src + 1;
}

__device__ __forceinline__ AF read_half(const char* buf, int lane, int half) {
    const bf16x8* a = reinterpret_cast<const bf16x8*>(buf + half * HALF_BYTES) + lane;
    AF f;
    f.h0 = a[0]; f.l0 = a[64]; f.h1 = a[128]; f.l1 = a[192];
    return f;
}

__device__ __forceinline__ void advance(Pipe& P) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * DMA_PER_SLOT) : "memory");
    __builtin_amdgcn_s_barrier();
    issue_dma(P, P.gpos);
    ++P.gpos;
    P.buf = P.ring + (P.gpos & (NBUF - 1)) * SLOT_BYTES;
}

// accumulators of one row-tile pair: [t][c]
struct Pair { f32x4 a[2][2]; };

// One half-step: for row tiles t = 0, 1 of the pair and column tiles c = 0, 1:  a[t][c] += A_t.hi B_c.hi + A_t.hi B_c.lo + A_t.lo B_c.hi
template <class F>
__device__ __forceinline__ void half_step(const int ODD, Pipe& P, AF& cur, int lane, const Frag& b0, const Frag& b1, Pair& acc, F&& fill) {
    if (ODD) advance(P);
    const AF nxt = read_half(P.buf, lane, ODD ? 0 : 1);
    SB0;
    acc.a[0][0] = MFMA16S(cur.h0, b0.hi, acc.a[0][0]); acc.a[0][0] = MFMA16S(cur.h0, b0.lo, acc.a[0][0]); acc.a[0][0] = MFMA16S(cur.l0, b0.hi, acc.a[0][0]);
    acc.a[0][1] = MFMA16S(cur.h0, b1.hi, acc.a[0][1]); acc.a[0][1] = MFMA16S(cur.h0, b1.lo, acc.a[0][1]); acc.a[0][1] = MFMA16S(cur.l0, b1.hi, acc.a[0][1]);
    acc.a[1][0] = MFMA16S(cur.h1, b0.hi, acc.a[1][0]); acc.a[1][0] = MFMA16S(cur.h1, b0.lo, acc.a[1][0]); acc.a[1][0] = MFMA16S(cur.l1, b0.hi, acc.a[1][0]);
    acc.a[1][1] = MFMA16S(cur.h1, b1.hi, acc.a[1][1]); acc.a[1][1] = MFMA16S(cur.h1, b1.lo, acc.a[1][1]); acc.a[1][1] = MFMA16S(cur.l1, b1.hi, acc.a[1][1]);
    fill();
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
    SB0;
    cur = nxt;
}
#define HS(ODD, B0, B1, ACC, ...) half_step((ODD), P, cur, lane, (B0), (B1), (ACC), [&]() { __VA_ARGS__; })

// bias of section `sec` -> both column tiles of one row-tile pair   (constants [sec][g][rt][r])
__device__ __forceinline__ void init_bias_pair(const float* cst, int sec, int g, int pair, Pair& acc) {
    const f32x4* b = reinterpret_cast<const f32x4*>(cst + sec * 128 + opaque(g * 32)) + 2 * pair;
#pragma unroll
    for (int t = 0; t < 2; ++t) { const f32x4 v = b[t]; acc.a[t][0] = v; acc.a[t][1] = v; }
}

struct HeadAcc { f32x2 s01, s23; };
// head partial sums over the four registers of row tile rt, for both column tiles (weights [g][rt][r][4] in LDS, read once)
template <class Map>
__device__ __forceinline__ void head_part(HeadAcc (&h)[2], const float* hd, int g, int rt, const f32x4& x0, const f32x4& x1, Map map) {
    const f32x4* w = reinterpret_cast<const f32x4*>(hd) + opaque(g * 32) + rt * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 wv = w[r];
        const float v0 = map(x0[r]), v1 = map(x1[r]);
        h[0].s01 = __builtin_elementwise_fma((f32x2){v0, v0}, (f32x2){wv.x, wv.y}, h[0].s01);
        h[0].s23 = __builtin_elementwise_fma((f32x2){v0, v0}, (f32x2){wv.z, wv.w}, h[0].s23);
        h[1].s01 = __builtin_elementwise_fma((f32x2){v1, v1}, (f32x2){wv.x, wv.y}, h[1].s01);
        h[1].s23 = __builtin_elementwise_fma((f32x2){v1, v1}, (f32x2){wv.z, wv.w}, h[1].s23);
    }
    pin(h[0].s01); pin(h[0].s23); pin(h[1].s01); pin(h[1].s23);
}
// sum over the four lane groups (x[l] + x[l^16] + x[l^32] + x[l^48]) with the two row-swap permutes, plus the head bias
__device__ __forceinline__ f32x4 head_finish(const HeadAcc& a, const float* hd) {
    auto fold = [](float v) {
        float p = v, q = v;
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
        float s = p + q, t = s;
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(s), "+v"(t));
        return s + t;
    };
    f32x4 s = {fold(a.s01.x), fold(a.s01.y), fold(a.s23.x), fold(a.s23.y)};
    const f32x4 b = *reinterpret_cast<const f32x4*>(hd + 512);
    return s + b;
}

// sin, cos of encoding argument `a` (per-lane, < 32): frequency a / 3, coordinate a % 3, on coordinates in revolutions.
// (scalars in, selects on values: an array indexed by a % 3 becomes a stack array, and `a` is laundered by the caller so
//  that the per-lane scales and selectors are recomputed per tile instead of living in registers across the tile loop)
__device__ __forceinline__ void encode_arg(float h0, float l0, float h1, float l1, float h2, float l2, int a, float* s, float* c) {
    const int fr = (a * 11) >> 5, co = a - 3 * fr;                 // a / 3 for a < 32
    const bool c0 = co == 0, c1 = co == 1;
    const float hi = c0 ? h0 : (c1 ? h1 : h2), lo = c0 ? l0 : (c1 ? l1 : l2);
    sincos_rev(hi, lo, __builtin_bit_cast(float, (127 + fr) << 23), s, c);
}

template <bool TILED, int NSRC>
__global__ void __launch_bounds__(64 * BW, 2) mlp_fwd_bf16_s16_kernel(ucnerf_mlp_params p, Geom gm, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [ring][constants][pe stash][u]
    char* ring = smem;
    float* cst = reinterpret_cast<float*>(smem + NBUF * SLOT_BYTES);
    Frag* stash_all = reinterpret_cast<Frag*>(smem + NBUF * SLOT_BYTES + ((CONST_FLOATS * 4 + 15) & ~15));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    constexpr int KD_S = (24 + 4 * NSRC + 31) / 32, KC_S = (8 * NSRC + 31) / 32;
    const int kd32 = NSRC ? KD_S : gm.kd32, kc32 = NSRC ? KC_S : gm.kc32;
    const int F = NSRC ? 24 + 12 * NSRC + 1 : gm.F, f_img = NSRC ? 24 + 4 * NSRC : gm.f_img;
    const char* __restrict__ ws = reinterpret_cast<const char*>(p.wstream);

    {
        const float* csrc = reinterpret_cast<const float*>(ws + gm.const_off_bytes);
        for (int i = threadIdx.x; i < CONST_FLOATS; i += 64 * BW) cst[i] = csrc[i];
    }
    __syncthreads();
    const float* hb = cst + N_SEC * 128;
    const float* ha = hb + 516;
    Frag* stash = stash_all + (size_t)wave * (4 * 64) + lane;                   // fragment (q, c) at stash[(2q + c) * 64]
    float* ustash_w = reinterpret_cast<float*>(stash_all + (size_t)BW * 4 * 64) + wave * 128;   // u of (c, lane) at [c * 64 + lane]

    Pipe P;
    P.gsrc = ws + wave * (DMA_PER_SLOT * 1024) + lane * 16;
    P.ring = ring; P.ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring; P.buf = ring;
    P.wave = wave; P.gpos = 0; P.next_src = 0; P.slots = gm.slots;
#pragma unroll
    for (int i = 0; i < NBUF; ++i) issue_dma(P, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * DMA_PER_SLOT) : "memory");
    __builtin_amdgcn_s_barrier();
    AF cur = read_half(P.buf, lane, 0);

    const int tiles_per_round = gridDim.x * BW;
    const int n_rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;

    // sample of (tile, column tile c) for this lane, clamped into the batch
    auto sample_of = [&](int tile, int c) { const int s_raw = tile * 32 + 16 * c + (opaque(lane) & 15); return s_raw < p.m ? s_raw : p.m - 1; };
    auto feat_base = [&](int s) { return TILED ? p.feats + ((size_t)(s >> 5) * F * 32 + (s & 31)) : p.feats + (size_t)s * gm.feat_stride; };
    constexpr int fstride = TILED ? 32 : 1;
    // natural-order operands of a bias net: element j of k32-step q = feature f0 + 32q + 8g + j (clamped into the row: columns
    // past the section's width meet zero weights)
    auto load_nat = [&](const float* fb, int f0, int q, float (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = fb[(size_t)min(f0 + 32 * q + 8 * opaque(g) + j, F - 1) * fstride];
    };
    float nfs[2][2][8], nconf[2], npx[2][3];              // [c][q]: depth-bias-net operands of the next tile, one tile ahead
    auto fetch = [&](int tile) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int s = sample_of(tile, c);
            const float* fb = feat_base(s);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (q < kd32) load_nat(fb, 0, q, nfs[c][q]);
            nconf[c] = fb[(size_t)(F - 1) * fstride];
            const float* prow = p.pts + (size_t)s * 3;
            npx[c][0] = prow[0]; npx[c][1] = prow[1]; npx[c][2] = prow[2];
        }
    };
    fetch(blockIdx.x * BW + wave);

    for (int round = 0; round < n_rounds; ++round) {
        const int tile = round * tiles_per_round + blockIdx.x * BW + wave;
        Pair bd[4], acc[4];                               // [row-tile pair]
        Frag X[4][2], Y[4][2];                            // [k32-step][column tile]
        float fsec[2][2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) fsec[c][q][e] = nfs[c][q][e];
            ustash_w[c * 64 + opaque(lane)] = 1.f - nconf[c];
        }

        // ---- point encoding -> fragments in LDS: this lane group's arguments a = 8g + j, sin in step 0, cos in step 1
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const Rev2 t0 = to_revolutions(npx[c][0]), t1 = to_revolutions(npx[c][1]), t2 = to_revolutions(npx[c][2]);
            const int a0 = 8 * opaque(g);
            float sn[8], cs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) encode_arg(t0.hi, t0.lo, t1.hi, t1.lo, t2.hi, t2.lo, a0 + j, &sn[j], &cs[j]);
            if (g == 3) { sn[6] = npx[c][0]; sn[7] = npx[c][1]; cs[6] = npx[c][2]; cs[7] = 0.f; }      // arguments 30, 31: raw coordinates
            stash[(0 + c) * 64] = split8(sn);
            stash[(2 + c) * 64] = split8(cs);
        }

        // ---- depth-bias net (step-major over the four pairs)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) init_bias_pair(cst, SEC_BD, g, pp, bd[pp]);
        {
            Frag f0 = split8(fsec[0][0]), f1 = split8(fsec[1][0]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (q < kd32) {
                    Frag n0, n1;
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        HS(pp & 1, f0, f1, bd[pp],
                           if (pp == 0 && q + 1 < 2 && q + 1 < kd32) { n0 = split8(fsec[0][q + 1]); pin(n0.hi); pin(n0.lo); }
                           if (pp == 1 && q + 1 < 2 && q + 1 < kd32) { n1 = split8(fsec[1][q + 1]); pin(n1.hi); pin(n1.lo); }
                           if (q + 1 >= kd32 && pp >= 2) init_bias_pair(cst, SEC_L0, g, pp - 2, acc[pp - 2]));
                    if (q + 1 < 2 && q + 1 < kd32) { f0 = n0; f1 = n1; }
                }
        }

        // ---- layer 0 (pair-split) on the point encoding
        {
#pragma unroll
            for (int q = 0; q < 2; ++q) {                                     // phase A -> pairs 0, 1
                const Frag b0 = stash[(2 * q) * 64], b1 = stash[(2 * q + 1) * 64];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    HS(pp, b0, b1, acc[pp], if (q == 1) init_bias_pair(cst, SEC_L0, g, 2 + pp, acc[2 + pp]));
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {                                     // phase B -> pairs 2, 3; epilogue of pairs 0, 1 underneath
                const Frag b0 = stash[(2 * q) * 64], b1 = stash[(2 * q + 1) * 64];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    HS(pp, b0, b1, acc[2 + pp],
                       X[q][pp] = frag_of<2>(acc[q].a[0][pp], acc[q].a[1][pp], bd[q].a[0][pp], bd[q].a[1][pp]);
                       if (q == 1 && pp == 1) { init_bias_pair(cst, SEC_L0 + 1, g, 0, acc[0]); init_bias_pair(cst, SEC_L0 + 1, g, 1, acc[1]); });
            }
        }

        // ---- layers 1..4: in[0..1] + pairs 2, 3 of the previous layer in acc (their epilogue -> in[2..3] runs under phase A)
        auto layer128 = [&](Frag (&in)[4][2], Frag (&out)[4][2], int sec) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    HS(pp, in[q][0], in[q][1], acc[pp],
                       if (q < 2) in[2 + q][pp] = frag_of<2>(acc[2 + q].a[0][pp], acc[2 + q].a[1][pp], bd[2 + q].a[0][pp], bd[2 + q].a[1][pp]);
                       if (q == 2) init_bias_pair(cst, sec, g, 2 + pp, acc[2 + pp]));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    HS(pp, in[q][0], in[q][1], acc[2 + pp],
                       if (!(q & 1)) out[q >> 1][pp] = frag_of<2>(acc[q >> 1].a[0][pp], acc[q >> 1].a[1][pp], bd[q >> 1].a[0][pp], bd[q >> 1].a[1][pp]);
                       if (q == 3) init_bias_pair(cst, sec + 1, g, pp, acc[pp]));
        };
#pragma unroll 1
        for (int l = 1; l < 5; l += 2) {
            layer128(X, Y, SEC_L0 + l);
            layer128(Y, X, SEC_L0 + l + 1);
        }

        // ---- layer 5 on [h | pe] in k order h0, h1, pe0, pe1, h2, h3; h5 = relu(. * bd) stays fp32 in acc
        const float* fb0 = feat_base(sample_of(tile, 0));
        const float* fb1 = feat_base(sample_of(tile, 1));
        float dv[2][3];
#pragma unroll
        for (int s = 0; s < 6; ++s) {                                         // phase A
            const bool pe = s == 2 || s == 3;
            const int q = s < 2 ? s : s - 2;
            const Frag b0 = pe ? stash[(2 * (s - 2)) * 64] : X[q][0], b1 = pe ? stash[(2 * (s - 2) + 1) * 64] : X[q][1];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                HS(pp, b0, b1, acc[pp],
                   if (s < 2) X[2 + s][pp] = frag_of<2>(acc[2 + s].a[0][pp], acc[2 + s].a[1][pp], bd[2 + s].a[0][pp], bd[2 + s].a[1][pp]);
                   if (s == 2) init_bias_pair(cst, SEC_L0 + 5, g, 2 + pp, acc[2 + pp]));
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {                                         // phase B; h5 of pairs 0, 1 underneath
            const bool pe = s == 2 || s == 3;
            const int q = s < 2 ? s : s - 2;
            const Frag b0 = pe ? stash[(2 * (s - 2)) * 64] : X[q][0], b1 = pe ? stash[(2 * (s - 2) + 1) * 64] : X[q][1];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                HS(pp, b0, b1, acc[2 + pp],
                   if (s == 0) {
                       _Pragma("unroll")
                       for (int t = 0; t < 2; ++t)
                           _Pragma("unroll")
                           for (int c = 0; c < 2; ++c) {
                               _Pragma("unroll")
                               for (int r = 0; r < 4; ++r) acc[pp].a[t][c][r] = fmaxf(acc[pp].a[t][c][r] * bd[pp].a[t][c][r], 0.f);
                               pin(acc[pp].a[t][c]);
                           }
                   }
                   );
        }
        // operands of the confidence-bias net + view direction: one batch of plain loads, issued once the trunk's fragments are dead
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* fb = c ? fb1 : fb0;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
                if (qq < kc32) load_nat(fb, f_img, qq, fsec[c][qq]);
            const int sd = sample_of(tile, c);
            int ray = sd;
            if (!p.dirs_per_sample) { int S = p.S; asm volatile("" : "+s"(S)); ray = sd / S; }
            const float* drow = p.dirs + (size_t)ray * 3;
            dv[c][0] = drow[0]; dv[c][1] = drow[1]; dv[c][2] = drow[2];
        }
#pragma unroll
        for (int pp = 2; pp < 4; ++pp)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[pp].a[t][c][r] = fmaxf(acc[pp].a[t][c][r] * bd[pp].a[t][c][r], 0.f);

        // ---- confidence-bias net (step-major) -> bd; base heads of row tiles 0..3 underneath
        HeadAcc hbase[2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
        auto ident = [](float v) { return v; };
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) init_bias_pair(cst, SEC_BC, g, pp, bd[pp]);
        {
            Frag f0 = split8(fsec[0][0]), f1 = split8(fsec[1][0]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (q < kc32) {
                    Frag n0, n1;
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        HS(pp & 1, f0, f1, bd[pp],
                           if (pp == 0 && q + 1 < 2 && q + 1 < kc32) { n0 = split8(fsec[0][q + 1]); pin(n0.hi); pin(n0.lo); }
                           if (pp == 1 && q + 1 < 2 && q + 1 < kc32) { n1 = split8(fsec[1][q + 1]); pin(n1.hi); pin(n1.lo); }
                           if (q == 0) head_part(hbase, hb, g, pp, acc[pp >> 1].a[pp & 1][0], acc[pp >> 1].a[pp & 1][1], ident));
                    if (q + 1 < 2 && q + 1 < kc32) { f0 = n0; f1 = n1; }
                }
        }
        // g = h5 * b_c: fragments of pairs 0, 1 now, of pairs 2, 3 under feature_linear's phase A
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) X[q][c] = frag_of<1>(acc[q].a[0][c], acc[q].a[1][c], bd[q].a[0][c], bd[q].a[1][c]);
        init_bias_pair(cst, SEC_FT, g, 0, acc[0]);
        init_bias_pair(cst, SEC_FT, g, 1, acc[1]);

        // ---- feature_linear (pair-split); base heads of row tiles 4..7 underneath
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                HS(pp, X[q][0], X[q][1], acc[pp],
                   if (q < 2) X[2 + q][pp] = frag_of<1>(acc[2 + q].a[0][pp], acc[2 + q].a[1][pp], bd[2 + q].a[0][pp], bd[2 + q].a[1][pp]);
                   if (q >= 2) head_part(hbase, hb, g, 4 + 2 * (q - 2) + pp, acc[2 + (q - 2)].a[pp][0], acc[2 + (q - 2)].a[pp][1], ident);
                   if (q == 3 && pp == 1) { init_bias_pair(cst, SEC_FT, g, 2, acc[2]); init_bias_pair(cst, SEC_FT, g, 3, acc[3]); });
        const f32x4 base0 = head_finish(hbase[0], hb), base1 = head_finish(hbase[1], hb);
        Frag D[2];                                                              // direction-encoding fragments of the two column tiles
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                HS(pp, X[q][0], X[q][1], acc[2 + pp],
                   if (!(q & 1)) Y[q >> 1][pp] = frag_of<0>(acc[q >> 1].a[0][pp], acc[q >> 1].a[1][pp], acc[q >> 1].a[0][pp], acc[q >> 1].a[1][pp]);
                   if (q & 1) {
                       if (q == 1) {                                          // direction encoding of column tile pp: slot n = 8g + j
                           const Rev2 t0 = to_revolutions(dv[pp][0]); const Rev2 t1 = to_revolutions(dv[pp][1]); const Rev2 t2 = to_revolutions(dv[pp][2]);
                           const int n0 = 8 * opaque(g);
                           float e[8];
                           _Pragma("unroll")
                           for (int j = 0; j < 8; ++j) {
                               const int n = n0 + j;
                               float sn; float cs;
                               encode_arg(t0.hi, t0.lo, t1.hi, t1.lo, t2.hi, t2.lo, n < 12 ? n : (n < 24 ? n - 12 : 0), &sn, &cs);
                               const float raw = n == 24 ? dv[pp][0] : (n == 25 ? dv[pp][1] : (n == 26 ? dv[pp][2] : 0.f));
                               e[j] = n < 12 ? sn : (n < 24 ? cs : raw);
                           }
                           D[pp] = split8(e); pin(D[pp].hi); pin(D[pp].lo);
                       } else init_bias_pair(cst, SEC_VC, g, pp, acc[pp]);
                   });

        // ---- views_linears | view_confi_linears on [feature | dir encoding] (pair-split), relu
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const Frag b0 = s < 4 ? Y[s][0] : D[0], b1 = s < 4 ? Y[s][1] : D[1];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                HS(pp, b0, b1, acc[pp],
                   if (s < 2) Y[2 + s][pp] = frag_of<0>(acc[2 + s].a[0][pp], acc[2 + s].a[1][pp], acc[2 + s].a[0][pp], acc[2 + s].a[1][pp]);
                   if (s == 2) init_bias_pair(cst, SEC_VC, g, 2 + pp, acc[2 + pp]));
        }
        HeadAcc hadapt[2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
        auto relu = [](float v) { return fmaxf(v, 0.f); };
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const Frag b0 = s < 4 ? Y[s][0] : D[0], b1 = s < 4 ? Y[s][1] : D[1];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                HS(pp, b0, b1, acc[2 + pp],
                   if (s < 2) head_part(hadapt, ha, g, 2 * s + pp, acc[s].a[pp][0], acc[s].a[pp][1], relu));
        }
        fetch(tile + tiles_per_round);
        // ---- adapt heads of row tiles 4..7, uncertainty blend
#pragma unroll
        for (int rt = 4; rt < 8; ++rt) head_part(hadapt, ha, g, rt, acc[rt >> 1].a[rt & 1][0], acc[rt >> 1].a[rt & 1][1], relu);
        const f32x4 adapt0 = head_finish(hadapt[0], ha), adapt1 = head_finish(hadapt[1], ha);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 base = c ? base1 : base0, adapt = c ? adapt1 : adapt0;
            const float u = ustash_w[c * 64 + opaque(lane)], omu = 1.f - u;
            const int s_raw = tile * 32 + 16 * c + (opaque(lane) & 15);
            f32x4 out;
            out.x = 1.f / (1.f + expf(-(base.x * omu + adapt.x * u)));
            out.y = 1.f / (1.f + expf(-(base.y * omu + adapt.y * u)));
            out.z = 1.f / (1.f + expf(-(base.z * omu + adapt.z * u)));
            out.w = fmaxf(adapt.w * omu + base.w * u, 0.f);
            if (g == 0 && s_raw < p.m) reinterpret_cast<f32x4*>(p.raw)[s_raw] = out;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr size_t smem_bytes() {
    return (size_t)NBUF * SLOT_BYTES + ((CONST_FLOATS * 4 + 15) & ~15) + (size_t)BW * 4 * 64 * sizeof(Frag) + (size_t)BW * 128 * sizeof(float);
}

}  // namespace s16

int launch_mlp_fwd_bf16_s16(const ucnerf_mlp_params* p, hipStream_t st) {
    using namespace s16;
    UCNERF_REQUIRE(p, "mlp_fwd: null params");
    if (p->m == 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->pts && p->dirs && p->feats && p->wstream && p->raw, "mlp_fwd: null pointer");
    UCNERF_REQUIRE(!p->encoded && !p->pts_stride && !p->dirs_stride, "mlp_fwd (bf16x3): encoded / strided inputs are only available in f32 precision");
    UCNERF_REQUIRE(p->dirs_per_sample || p->S > 0, "mlp_fwd: S must be > 0 when dirs are per ray");
    UCNERF_REQUIRE(((uintptr_t)p->wstream & 15) == 0 && ((uintptr_t)p->raw & 15) == 0, "mlp_fwd: wstream/raw must be 16-byte aligned");
    Layout B;
    UCNERF_REQUIRE(layout(p->cfg.n_src, &B), "mlp_fwd: n_src %d outside 1..8", p->cfg.n_src);
    UCNERF_REQUIRE(B.kd32 <= 2 && B.kc32 <= 2, "mlp_fwd (16x16x32): bias-net width");
    const int n_tiles = cdiv(p->m, 32);
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_fwd: no device");
    int blocks = cdiv(n_tiles, BW);
    const int cap = p->max_blocks > 0 ? p->max_blocks : 2 * cus;
    if (blocks > cap) blocks = cap;
    Geom g;
    g.F = B.F; g.kd32 = B.kd32; g.kc32 = B.kc32; g.f_img = 24 + 4 * B.v; g.slots = B.halfs / 2;
    g.feat_stride = p->feat_stride ? p->feat_stride : B.F;
    g.const_off_bytes = (int)B.const_off_bytes;
    static bool attr_set = false;
    const size_t smem = smem_bytes();
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_s16_kernel<true, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_s16_kernel<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_s16_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_s16_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    dim3 grid(blocks), block(64 * BW);
    if (B.v == 6) {
        if (p->feats_tiled) hipLaunchKernelGGL((mlp_fwd_bf16_s16_kernel<true, 6>), grid, block, smem, st, *p, g, n_tiles);
        else hipLaunchKernelGGL((mlp_fwd_bf16_s16_kernel<false, 6>), grid, block, smem, st, *p, g, n_tiles);
    } else {
        if (p->feats_tiled) hipLaunchKernelGGL((mlp_fwd_bf16_s16_kernel<true, 0>), grid, block, smem, st, *p, g, n_tiles);
        else hipLaunchKernelGGL((mlp_fwd_bf16_s16_kernel<false, 0>), grid, block, smem, st, *p, g, n_tiles);
    }
    return check_launch("mlp_fwd_bf16_s16");
}

}  // namespace ucnerf
