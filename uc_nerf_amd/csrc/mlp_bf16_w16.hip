// The gather-fused split-bf16 MLP forward of mlp_bf16.hip on 16-SAMPLE tiles (v_mfma_f32_16x16x32_bf16): the kernel for PARTLY FILLED rounds.
//
// Why a second tile size: a data-parallel shard of a few hundred rays (512 rays x 64 + 192 samples = 0.5 + 1.5 rounds of the chip's 2 048 wave
// slots with 32-sample tiles) is bound by the latency of ONE tile, not by throughput: a lone wave on a SIMD keeps the matrix pipe busy 47 % of
// its tile's 29 us.  With 16 samples per wave the same pass is 1 + 3 FULL rounds of waves that each carry half the matrix and element-wise work,
// two per SIMD, one filling the other's gaps.  (Per sample the weights cross LDS twice as often -- the price of a small tile, which is why
// full rounds stay on the 32-sample kernel: the launcher in render.hip picks per pass.)
//
// Same algorithm, arithmetic and ring / DMA / barrier scheme as mlp_bf16.hip (precision 3); what changes is the register-resident layout
// (first built as round 1's 16x16x32 experiment, which kept 32 samples per wave):
//   * lane l = (g = l >> 4, i = l & 15) works for sample 16 tile + i; an accumulator row tile (16 output rows x 16 samples) is 4 registers per
//     lane holding rows 16 rt + 4 g + r: eight row tiles per 128-wide layer;
//   * a k32-step q contracts 32 input features; for hidden layers its B fragment is built from the accumulators of row tiles 2q and 2q + 1:
//     element j < 4 = feature 32 q + 4 g + j, j >= 4 = feature 32 q + 16 + 4 g + j - 4 -- no data movement between a layer's output and the next
//     layer's operand;
//   * a half-step = one k32-step for one PAIR of row tiles (32 rows): 4 KB of weights [hi0 | lo0 | hi1 | lo1], six MFMAs of 16 cycles.  Phase A of
//     a 128-wide layer = row-tile pairs 0, 1 for all k32-steps, phase B = pairs 2, 3; the epilogue of phase A's rows is issued under phase B's
//     MFMAs, that of phase B's under the next layer's phase A;
//   * the gather: lane group g gathers cascade volume g (g < 3; g = 3 the confidence map) and the source views 2 pr, 2 pr + 1 of pair
//     pr = (g + 1) & 3 -- which is exactly the NATURAL column order of the two bias nets' k32-steps (element j of group g in step q = column
//     32 q + 8 g + j), so this kernel's stream needs no operand permutation; a view's eight image features go through LDS to the lane group whose
//     confidence-net operand they are.  Feature arithmetic and accumulation order per feature are gather_cl.hip's.
#include "common.h"
#include "mlp_layout.h"
#include "sincos_cw.h"
#include "gather_cl_device.h"

// the LDS-DMA asm below names m0 as a clobber on purpose (it loads the LDS base into it)
#pragma clang diagnostic ignored "-Winline-asm"

#include <vector>

namespace ucnerf {
namespace w16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define WSB0 __builtin_amdgcn_sched_barrier(0)

constexpr int BW = 8;                 // waves per block = per CU, sharing one weight ring (as mlp_bf16.hip)
constexpr int SLOT_BYTES = 8192;      // two half-steps
constexpr int HALF_BYTES = 4096;      // [hi0, lo0, hi1, lo1][64 lanes][16 B]
constexpr int NBUF = 4;
constexpr int DMA_PER_SLOT = SLOT_BYTES / 1024 / BW;
constexpr int TILE = 16;              // samples per wave
constexpr int VIEW_TAB = 24;          // floats per source view in the LDS table: w2c (12), K (9), pad

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

}  // namespace w16

// ------------------------------------------------------------------------------------------------ host: pack index (cfg.precision 4)
// Element format of build_pack_index_bf16 (flat parameter index | part << 30, -1 = zero), half-steps in consumption order, then the fp32
// constants in THIS kernel's register order: bias blocks [sec][g][rt][r], head blocks [g][rt][r][4] + 4.
int build_pack_index_bf16_w16(const ucnerf_mlp_config* cfg, int32_t* idx) {
    using namespace w16;
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
    {   // bd, step-major over the four pairs (natural column order = the gather's operand order, see the file comment)
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

int64_t bf16_w16_index_count(const ucnerf_mlp_config* cfg) {
    w16::Layout B;
    if (!w16::layout(cfg->n_src, &B)) return -1;
    return (int64_t)B.halfs * (w16::HALF_BYTES / 2) + CONST_FLOATS;
}

int64_t bf16_w16_stream_floats(const ucnerf_mlp_config* cfg) {
    w16::Layout B;
    if (!w16::layout(cfg->n_src, &B)) return -1;
    return B.total_bytes / 4;
}

int64_t bf16_w16_const_off_bytes(const ucnerf_mlp_config* cfg) {
    w16::Layout B;
    if (!w16::layout(cfg->n_src, &B)) return -1;
    return B.const_off_bytes;
}

namespace w16 {

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

// fragment of k32-step q: rows of row tiles 2q (a0) and 2q + 1 (a1); MODE 0 plain, 1 times m, 2 relu(times m)
template <int MODE>
__device__ __forceinline__ Frag frag_of(const f32x4& a0, const f32x4& a1, const f32x4& m0, const f32x4& m1) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        f32x2 v0 = {a0[j], a0[j + 1]}, v1 = {a1[j], a1[j + 1]};
        if (MODE >= 1) { v0 = v0 * (f32x2){m0[j], m0[j + 1]}; v1 = v1 * (f32x2){m1[j], m1[j + 1]}; }
        t[j] = MODE == 2 ? fmaxf(v0.x, 0.f) : v0.x; t[j + 1] = MODE == 2 ? fmaxf(v0.y, 0.f) : v0.y;
        t[4 + j] = MODE == 2 ? fmaxf(v1.x, 0.f) : v1.x; t[5 + j] = MODE == 2 ? fmaxf(v1.y, 0.f) : v1.y;
    }
    Frag f = split8(t);
    pin(f.hi); pin(f.lo);
    return f;
}

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
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NBUF - 2) * DMA_PER_SLOT) : "memory");
    __builtin_amdgcn_s_barrier();
    issue_dma(P, P.gpos);
    ++P.gpos;
    P.buf = P.ring + (P.gpos & (NBUF - 1)) * SLOT_BYTES;
}

// One half-step: for row tiles t = 0, 1 of pair pp:  acc[2 pp + t] += A_t.hi B.hi + A_t.hi B.lo + A_t.lo B.hi
template <class F>
__device__ __forceinline__ void half_step(const int ODD, Pipe& P, AF& cur, int lane, const Frag& b, f32x4& c0, f32x4& c1, F&& fill) {
    if (ODD) advance(P);
    const AF nxt = read_half(P.buf, lane, ODD ? 0 : 1);
    WSB0;
    c0 = MFMA16S(cur.h0, b.hi, c0); c0 = MFMA16S(cur.h0, b.lo, c0); c0 = MFMA16S(cur.l0, b.hi, c0);
    c1 = MFMA16S(cur.h1, b.hi, c1); c1 = MFMA16S(cur.h1, b.lo, c1); c1 = MFMA16S(cur.l1, b.hi, c1);
    fill();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    WSB0;
    cur = nxt;
}
#define WHS(ODD, B, PP, ACC, ...) half_step((ODD), P, cur, lane, (B), (ACC)[2 * (PP)], (ACC)[2 * (PP) + 1], [&]() { __VA_ARGS__; })

// bias of section `sec` -> the two row tiles of one pair   (constants [sec][g][rt][r])
__device__ __forceinline__ void init_bias_pair(const float* cst, int sec, int g, int pair, f32x4 (&acc)[8]) {
    const f32x4* b = reinterpret_cast<const f32x4*>(cst + sec * 128 + opaque(g * 32)) + 2 * pair;
    acc[2 * pair] = b[0];
    acc[2 * pair + 1] = b[1];
}

struct HeadAcc { f32x2 s01, s23; };
// head partial sums over the four registers of row tile rt (weights [g][rt][r][4] in LDS)
template <class Map>
__device__ __forceinline__ void head_part(HeadAcc& h, const float* hd, int g, int rt, const f32x4& x, Map map) {
    const f32x4* w = reinterpret_cast<const f32x4*>(hd) + opaque(g * 32) + rt * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 wv = w[r];
        const float v = map(x[r]);
        h.s01 = __builtin_elementwise_fma((f32x2){v, v}, (f32x2){wv.x, wv.y}, h.s01);
        h.s23 = __builtin_elementwise_fma((f32x2){v, v}, (f32x2){wv.z, wv.w}, h.s23);
    }
    pin(h.s01); pin(h.s23);
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
__device__ __forceinline__ void encode_arg(float h0, float l0, float h1, float l1, float h2, float l2, int a, float* s, float* c) {
    const int fr = (a * 11) >> 5, co = a - 3 * fr;                 // a / 3 for a < 32
    const bool c0 = co == 0, c1 = co == 1;
    const float hi = c0 ? h0 : (c1 ? h1 : h2), lo = c0 ? l0 : (c1 ? l1 : l2);
    sincos_rev(hi, lo, __builtin_bit_cast(float, (127 + fr) << 23), s, c);
}

// what the launch needs of the render pass (the FusedGather of mlp_bf16.hip, derived coordinates only)
struct Gather {
    int m, S, V, H, W;
    int vol_d[3], vol_h[3], vol_w[3];
    unsigned vol_off[3], img_off, view_bytes;      // byte offsets inside the channel-last buffer
    const char* cl;
    const float* conf;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* dirs;             // [n,3] view-direction feature per ray
    float near, far;
    float w2c_ref[12], K_ref[9];
    const float* w2cs;
    const float* Ks;
    unsigned div_m, div_sh;
    const char* wstream;
    float* raw;
    int slots, const_off_bytes;
};

template <int NSRC>
__global__ void __launch_bounds__(64 * BW, 2) mlp_fwd_bf16_w16_kernel(Gather fg, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [ring][constants][pe stash][bc operands][u][view table]
    char* ring = smem;
    float* cst = reinterpret_cast<float*>(smem + NBUF * SLOT_BYTES);
    Frag* stash_all = reinterpret_cast<Frag*>(smem + NBUF * SLOT_BYTES + ((CONST_FLOATS * 4 + 15) & ~15));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    constexpr int KD = (24 + 4 * NSRC + 31) / 32, KC = (8 * NSRC + 31) / 32;
    constexpr int NP = (NSRC + 1) / 2;                                 // view pairs
    const char* __restrict__ ws = fg.wstream;

    {
        const float* csrc = reinterpret_cast<const float*>(ws + fg.const_off_bytes);
        for (int i = threadIdx.x; i < CONST_FLOATS; i += 64 * BW) cst[i] = csrc[i];
    }
    const float* hb = cst + N_SEC * 128;
    const float* ha = hb + 516;
    Frag* stash = stash_all + (size_t)wave * (2 * 64) + lane;                          // point-encoding fragment of k32-step q at stash[q * 64]
    float* bc_all = reinterpret_cast<float*>(stash_all + (size_t)BW * 2 * 64);          // confidence-net operands: [wave][q][g'][i][8]
    float* bc_w = bc_all + (size_t)wave * (2 * 64 * 8);
    float* ustash_w = bc_all + (size_t)BW * (2 * 64 * 8) + wave * 16;                   // u of sample i
    float* vtab = bc_all + (size_t)BW * (2 * 64 * 8) + BW * 16;
    for (int i = threadIdx.x; i < fg.V * VIEW_TAB; i += 64 * BW) {
        const int vi = i / VIEW_TAB, e = i % VIEW_TAB;
        vtab[i] = e < 12 ? fg.w2cs[12 * vi + e] : e < 21 ? fg.Ks[9 * vi + e - 12] : 0.f;
    }
    for (int i = threadIdx.x; i < BW * 2 * 64 * 8; i += 64 * BW) bc_all[i] = 0.f;       // slots of views past NSRC are never written: they meet zero weights as zeros
    __syncthreads();

    Pipe P;
    P.gsrc = ws + wave * (DMA_PER_SLOT * 1024) + lane * 16;
    P.ring = ring; P.ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring; P.buf = ring;
    P.wave = wave; P.gpos = 0; P.next_src = 0; P.slots = fg.slots;
#pragma unroll
    for (int i = 0; i < NBUF; ++i) issue_dma(P, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * DMA_PER_SLOT) : "memory");
    __builtin_amdgcn_s_barrier();
    AF cur = read_half(P.buf, lane, 0);

    // tiles dealt wave-major inside a round, logical block index XCD-aware (mlp_bf16.hip)
    const int tiles_per_round = gridDim.x * BW;
    const int n_rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;
    const int nb_ = (int)gridDim.x, bx_ = (int)blockIdx.x;
    const int lblock = (nb_ & 7) == 0 ? (bx_ & 7) * (nb_ >> 3) + (bx_ >> 3) : bx_;
    const int tile0 = wave * nb_ + lblock;

    auto sample_of = [&](int tile) { const int s_raw = tile * TILE + (opaque(lane) & 15); return s_raw < fg.m ? s_raw : fg.m - 1; };

    float ro[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ro[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fg.rays_o[i])));

    // ---- the gather, staged as in mlp_bf16.hip: g_pre (depth and ray of the NEXT tile's sample, loaded in the last fill of the tile), then at the top
    //      of the tile loop footprints -> loads of the volume / confidence and the first view -> point encoding -> consume in issue order
    struct VolFp { unsigned o[4], dx; float w[4], wx0, wx1; };
    struct ImgFp { unsigned p00, p10, dx; float w00, w01, w10, w11, mask; };
    float gz = 0.f, grd[3] = {0.f, 0.f, 0.f};
    float npx[3], gqz = 0.f;
    float nfs[2][8];
    auto g_pre = [&](int tile) {
        const unsigned s = (unsigned)sample_of(tile);
        const unsigned r = fg.S == 1 ? s : (__umulhi(s, fg.div_m) >> fg.div_sh);
        gz = fg.z[s];
        const float* rd = fg.rays_d + 3 * (size_t)r;
        grd[0] = rd[0]; grd[1] = rd[1]; grd[2] = rd[2];
    };
    auto sop = [](int v) { asm volatile("" : "+s"(v)); return v; };     // (wave-uniform sizes: their float forms are not hoisted into loop-long vector registers)
    auto gather_tile = [&](auto encode) {
        const int gl = opaque(g);
        const int gW = sop(fg.W), gH = sop(fg.H);
        const float x = ro[0] + gz * grd[0], y = ro[1] + gz * grd[1], w = ro[2] + gz * grd[2];
        {   // reference projection: the point the encoding takes and the cascade coordinates
            float qx, qy, qz;
            project_cl(fg.w2c_ref, fg.K_ref, x, y, w, &qx, &qy, &qz);
            npx[0] = (qx / qz + 0.0f) / (float)(gW - 1); npx[1] = (qy / qz + 0.0f) / (float)(gH - 1);
            npx[2] = (qz - fg.near) / (fg.far - fg.near);
            gqz = qz;
        }
        // this lane group's volume (g < 3), all eight channels
        VolFp fa;
        {
            const int unit = gl < 3 ? gl : 2;
            const float zn = (gqz - fg.near) / (fg.far - fg.near);
            const int D = unit == 0 ? fg.vol_d[0] : unit == 1 ? fg.vol_d[1] : fg.vol_d[2];
            const int hh = unit == 0 ? fg.vol_h[0] : unit == 1 ? fg.vol_h[1] : fg.vol_h[2];
            const int ww = unit == 0 ? fg.vol_w[0] : unit == 1 ? fg.vol_w[1] : fg.vol_w[2];
            const unsigned vb = unit == 0 ? fg.vol_off[0] : unit == 1 ? fg.vol_off[1] : fg.vol_off[2];
            const LerpCl ax = axis_cl(npx[0] * 2.f - 1.0f, ww, false), ay = axis_cl(npx[1] * 2.f - 1.0f, hh, false), az = axis_cl(zn * 2.f - 1.0f, D, false);
            fa.o[0] = vb + (unsigned)((az.i0 * hh + ay.i0) * ww + ax.i0) * 32u; fa.o[1] = vb + (unsigned)((az.i0 * hh + ay.i1) * ww + ax.i0) * 32u;
            fa.o[2] = vb + (unsigned)((az.i1 * hh + ay.i0) * ww + ax.i0) * 32u; fa.o[3] = vb + (unsigned)((az.i1 * hh + ay.i1) * ww + ax.i0) * 32u;
            fa.dx = (unsigned)(ax.i1 - ax.i0) * 32u;
            fa.w[0] = az.w0 * ay.w0; fa.w[1] = az.w0 * ay.w1; fa.w[2] = az.w1 * ay.w0; fa.w[3] = az.w1 * ay.w1;
            fa.wx0 = ax.w0; fa.wx1 = ax.w1;
        }
        // confidence: the stage-3 grid (lane group 3 consumes it)
        unsigned co[4];
        float cw[4];
        {
            const LerpCl ax = axis_cl(npx[0] * 2.f - 1.0f, gW, false), ay = axis_cl(npx[1] * 2.f - 1.0f, gH, false);
            co[0] = (unsigned)(ay.i0 * gW + ax.i0); co[1] = (unsigned)(ay.i0 * gW + ax.i1);
            co[2] = (unsigned)(ay.i1 * gW + ax.i0); co[3] = (unsigned)(ay.i1 * gW + ax.i1);
            cw[0] = ay.w0 * ax.w0; cw[1] = ay.w0 * ax.w1; cw[2] = ay.w1 * ax.w0; cw[3] = ay.w1 * ax.w1;
        }
        // this lane group's two source views: 2 pr, 2 pr + 1 of pair pr = (g + 1) & 3
        const int pr = (gl + 1) & 3;
        ImgFp fi[2];
        bool vok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int view = 2 * pr + k;
            vok[k] = view < NSRC;
            const int vi = view < NSRC ? view : NSRC - 1;         // (a group without a view repeats the last one: finite values, discarded)
            const float* mt = vtab + vi * VIEW_TAB;
            float qx, qy, qv;
            project_cl(mt, mt + 12, x, y, w, &qx, &qy, &qv);
            const float gx = (qx / qv + 0.0f) / (float)(gW - 1) * 2.0f - 1.0f, gy = (qy / qv + 0.0f) / (float)(gH - 1) * 2.0f - 1.0f;
            const LerpCl ax = axis_cl(gx, gW, true), ay = axis_cl(gy, gH, true);
            const unsigned ib = fg.img_off + (unsigned)vi * fg.view_bytes;
            fi[k].p00 = ib + (unsigned)(ay.i0 * gW + ax.i0) * 48u; fi[k].p10 = ib + (unsigned)(ay.i1 * gW + ax.i0) * 48u;
            fi[k].dx = (unsigned)(ax.i1 - ax.i0) * 48u;
            fi[k].w00 = ay.w0 * ax.w0; fi[k].w01 = ay.w0 * ax.w1; fi[k].w10 = ay.w1 * ax.w0; fi[k].w11 = ay.w1 * ax.w1;
            fi[k].mask = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.f : 0.f;
        }
        // ---- loads: volume corners, confidence, the first view
        float4 va[16], vi0[12], vi1[12];
        float cv[4];
        WSB0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {                         // corner c = (z, y, x): the accumulation order of gather_cl.hip
            const unsigned o = fa.o[c >> 1] + ((c & 1) ? fa.dx : 0u);
            va[2 * c] = ld16(fg.cl, o); va[2 * c + 1] = ld16(fg.cl, o + 16u);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cv[c] = fg.conf[co[c]];
        auto img_loads = [&](const ImgFp& f, float4 (&v)[12]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {                     // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
                const unsigned o = ((c & 2) ? f.p10 : f.p00) + ((c & 1) ? f.dx : 0u);
                v[3 * c] = ld16(fg.cl, o); v[3 * c + 1] = ld16(fg.cl, o + 16u); v[3 * c + 2] = ld16(fg.cl, o + 32u);
            }
        };
        img_loads(fi[0], vi0);
        WSB0;
        encode();                                             // the point encoding while the loads are in flight
        WSB0;
        // ---- consume in issue order
        float vol8[8];
        {
            gf2 lo[2] = {{0, 0}, {0, 0}}, hi[2] = {{0, 0}, {0, 0}};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w__ = fa.w[c >> 1] * ((c & 1) ? fa.wx1 : fa.wx0);
                FMA4(lo, va[2 * c], w__) FMA4(hi, va[2 * c + 1], w__)
            }
            vol8[0] = lo[0].x; vol8[1] = lo[0].y; vol8[2] = lo[1].x; vol8[3] = lo[1].y;
            vol8[4] = hi[0].x; vol8[5] = hi[0].y; vol8[6] = hi[1].x; vol8[7] = hi[1].y;
        }
        {
            float acc_ = cv[0] * cw[0];
            acc_ += cv[1] * cw[1];
            acc_ += cv[2] * cw[2];
            acc_ += cv[3] * cw[3];
            if (gl == 3) ustash_w[opaque(lane) & 15] = 1.f - acc_;          // u: read back by lane group 0 at the very end of the tile
        }
        WSB0;
        img_loads(fi[1], vi1);
        WSB0;
        float rgbm[2][4];
        auto view_consume = [&](int k, const ImgFp& f, const float4 (&v)[12]) {
            gf2 c0[2] = {{0, 0}, {0, 0}}, c1[2] = {{0, 0}, {0, 0}}, c2[2] = {{0, 0}, {0, 0}};      // (r g b f0) (f1..f4) (f5 f6 f7 -)
            const float wt[4] = {f.w00, f.w01, f.w10, f.w11};
#pragma unroll
            for (int c = 0; c < 4; ++c) { FMA4(c0, v[3 * c], wt[c]) FMA4(c1, v[3 * c + 1], wt[c]) FMA4(c2, v[3 * c + 2], wt[c]) }
            rgbm[k][0] = c0[0].x; rgbm[k][1] = c0[0].y; rgbm[k][2] = c0[1].x; rgbm[k][3] = f.mask;
            if (vok[k]) {                                    // the view's eight image features -> the lane group whose confidence-net operand they are
                const int view = 2 * pr + k;
                f32x4* dst = reinterpret_cast<f32x4*>(bc_w + ((size_t)((view >> 2) * 4 + (view & 3)) * 16 + (opaque(lane) & 15)) * 8);
                dst[0] = (f32x4){c0[1].y, c1[0].x, c1[0].y, c1[1].x};
                dst[1] = (f32x4){c1[1].y, c2[0].x, c2[0].y, c2[1].x};
            }
        };
        view_consume(0, fi[0], vi0);
        view_consume(1, fi[1], vi1);
        // operands of the depth-bias net: step 0 = this group's volume (g < 3) or views 0, 1 (g = 3); step 1 = this group's views (g < 3)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float vw = (e < 4 ? (vok[0] ? rgbm[0][e] : 0.f) : (vok[1] ? rgbm[1][e - 4] : 0.f));
            nfs[0][e] = gl < 3 ? vol8[e] : vw;
            nfs[1][e] = gl < 3 ? vw : 0.f;
        }
    };

    g_pre(tile0);

    for (int round = 0; round < n_rounds; ++round) {                          // block-uniform trip count: every wave joins every barrier
        const int tile = round * tiles_per_round + tile0;
        if (tile >= n_tiles) {                               // (wave-uniform) a wave without a tile only keeps the weight ring turning
            for (int i = 0; i < fg.slots; ++i) advance(P);
            cur = read_half(P.buf, lane, 0);
            continue;
        }
        // ---- this tile's gather; its point encoding -> fragments in LDS: this lane group's arguments a = 8g + j, sin in step 0, cos in step 1
        gather_tile([&]() {
            const Rev2 t0 = to_revolutions(npx[0]), t1 = to_revolutions(npx[1]), t2 = to_revolutions(npx[2]);
            const int gl = opaque(g);
            const int a0 = 8 * gl;
            float sn[8], cs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) encode_arg(t0.hi, t0.lo, t1.hi, t1.lo, t2.hi, t2.lo, a0 + j, &sn[j], &cs[j]);
            if (gl == 3) { sn[6] = npx[0]; sn[7] = npx[1]; cs[6] = npx[2]; cs[7] = 0.f; }      // arguments 30, 31: raw coordinates
            stash[0] = split8(sn);
            stash[64] = split8(cs);
        });
        cur = read_half(P.buf, lane, 0);               // the fragments the last half-step left in `cur`, read again (registers for the gather)

        f32x4 bd[8], acc[8];                              // [row tile]
        Frag X[4], Y[4];                                  // [k32-step]
        float fsec[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) fsec[q][e] = nfs[q][e];

        // ---- depth-bias net (step-major over the four pairs)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) init_bias_pair(cst, SEC_BD, g, pp, bd);
        {
            Frag f0 = split8(fsec[0]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (q < KD) {
                    Frag n0;
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        WHS(pp & 1, f0, pp, bd,
                            if (pp == 0 && q + 1 < KD) { n0 = split8(fsec[q + 1]); pin(n0.hi); pin(n0.lo); }
                            if (q + 1 >= KD && pp >= 2) init_bias_pair(cst, SEC_L0, g, pp - 2, acc));
                    if (q + 1 < KD) f0 = n0;
                }
        }

        // ---- layer 0 (pair-split) on the point encoding
        {
#pragma unroll
            for (int q = 0; q < 2; ++q) {                                     // phase A -> pairs 0, 1
                const Frag b0 = stash[q * 64];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    WHS(pp, b0, pp, acc, if (q == 1) init_bias_pair(cst, SEC_L0, g, 2 + pp, acc));
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {                                     // phase B -> pairs 2, 3; epilogue of pairs 0, 1 underneath
                const Frag b0 = stash[q * 64];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    WHS(pp, b0, 2 + pp, acc,
                        if (pp == 0) X[q] = frag_of<2>(acc[2 * q], acc[2 * q + 1], bd[2 * q], bd[2 * q + 1]);
                        if (q == 1 && pp == 1) { init_bias_pair(cst, SEC_L0 + 1, g, 0, acc); init_bias_pair(cst, SEC_L0 + 1, g, 1, acc); });
            }
        }

        // ---- layers 1..4: in[0..1] + pairs 2, 3 of the previous layer in acc (their epilogue -> in[2..3] runs under phase A)
        auto layer128 = [&](Frag (&in)[4], Frag (&out)[4], int sec) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    WHS(pp, in[q], pp, acc,
                        if (q < 2 && pp == 0) in[2 + q] = frag_of<2>(acc[4 + 2 * q], acc[5 + 2 * q], bd[4 + 2 * q], bd[5 + 2 * q]);
                        if (q == 2) init_bias_pair(cst, sec, g, 2 + pp, acc));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    WHS(pp, in[q], 2 + pp, acc,
                        if (!(q & 1) && pp == 0) out[q >> 1] = frag_of<2>(acc[2 * (q >> 1)], acc[2 * (q >> 1) + 1], bd[2 * (q >> 1)], bd[2 * (q >> 1) + 1]);
                        if (q == 3) init_bias_pair(cst, sec + 1, g, pp, acc));
        };
#pragma unroll 1
        for (int l = 1; l < 5; l += 2) {
            layer128(X, Y, SEC_L0 + l);
            layer128(Y, X, SEC_L0 + l + 1);
        }

        // ---- layer 5 on [h | pe] in k order h0, h1, pe0, pe1, h2, h3; h5 = relu(. * bd) stays fp32 in acc
        float dv[3];
#pragma unroll
        for (int s = 0; s < 6; ++s) {                                         // phase A
            const bool pe = s == 2 || s == 3;
            const int q = s < 2 ? s : s - 2;
            const Frag b0 = pe ? stash[(s - 2) * 64] : X[q];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                WHS(pp, b0, pp, acc,
                    if (s < 2 && pp == 0) X[2 + s] = frag_of<2>(acc[4 + 2 * s], acc[5 + 2 * s], bd[4 + 2 * s], bd[5 + 2 * s]);
                    if (s == 2) init_bias_pair(cst, SEC_L0 + 5, g, 2 + pp, acc));
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {                                         // phase B; h5 of pairs 0, 1 underneath
            const bool pe = s == 2 || s == 3;
            const int q = s < 2 ? s : s - 2;
            const Frag b0 = pe ? stash[(s - 2) * 64] : X[q];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                WHS(pp, b0, 2 + pp, acc,
                    if (s == 0) {
                        _Pragma("unroll")
                        for (int t = 0; t < 2; ++t) {
                            _Pragma("unroll")
                            for (int r = 0; r < 4; ++r) acc[2 * pp + t][r] = fmaxf(acc[2 * pp + t][r] * bd[2 * pp + t][r], 0.f);
                            pin(acc[2 * pp + t]);
                        }
                    }
                    if (s == 4 && pp == 0) {                                   // operands of the confidence-bias net (from LDS) + view direction: one batch of loads
                        const int gl = opaque(g);
                        _Pragma("unroll")
                        for (int qq = 0; qq < 2; ++qq) {
                            const f32x4* src_ = reinterpret_cast<const f32x4*>(bc_w + ((size_t)(qq * 4 + gl) * 16 + (opaque(lane) & 15)) * 8);
                            const f32x4 a_ = src_[0], b_ = src_[1];
                            fsec[qq][0] = a_.x; fsec[qq][1] = a_.y; fsec[qq][2] = a_.z; fsec[qq][3] = a_.w;
                            fsec[qq][4] = b_.x; fsec[qq][5] = b_.y; fsec[qq][6] = b_.z; fsec[qq][7] = b_.w;
                        }
                        const unsigned sd = (unsigned)sample_of(tile);
                        const unsigned ray = fg.S == 1 ? sd : (__umulhi(sd, fg.div_m) >> fg.div_sh);
                        const float* drow = fg.dirs + (size_t)ray * 3;
                        dv[0] = drow[0]; dv[1] = drow[1]; dv[2] = drow[2];
                    });
        }
#pragma unroll
        for (int rt = 4; rt < 8; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][r] = fmaxf(acc[rt][r] * bd[rt][r], 0.f);

        // ---- confidence-bias net (step-major) -> bd; base heads of row tiles 0..3 underneath
        HeadAcc hbase = {{0.f, 0.f}, {0.f, 0.f}};
        auto ident = [](float v) { return v; };
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) init_bias_pair(cst, SEC_BC, g, pp, bd);
        {
            Frag f0 = split8(fsec[0]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (q < KC) {
                    Frag n0;
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
                        WHS(pp & 1, f0, pp, bd,
                            if (pp == 0 && q + 1 < KC) { n0 = split8(fsec[q + 1]); pin(n0.hi); pin(n0.lo); }
                            if (q == 0) head_part(hbase, hb, g, pp, acc[pp], ident));
                    if (q + 1 < KC) f0 = n0;
                }
        }
        // g = h5 * b_c: fragments of pairs 0, 1 now, of pairs 2, 3 under feature_linear's phase A
#pragma unroll
        for (int q = 0; q < 2; ++q) X[q] = frag_of<1>(acc[2 * q], acc[2 * q + 1], bd[2 * q], bd[2 * q + 1]);
        // (row tiles 0..3 of h5 are consumed: their base-head sums were taken above; tiles 4..7 stay in acc[4..7] until feature_linear's phase A has used them)
        init_bias_pair(cst, SEC_FT, g, 0, acc);
        init_bias_pair(cst, SEC_FT, g, 1, acc);

        // ---- feature_linear (pair-split); base heads of row tiles 4..7 underneath
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                WHS(pp, X[q], pp, acc,
                    if (q < 2 && pp == 0) X[2 + q] = frag_of<1>(acc[4 + 2 * q], acc[5 + 2 * q], bd[4 + 2 * q], bd[5 + 2 * q]);
                    if (q >= 2) head_part(hbase, hb, g, 4 + 2 * (q - 2) + pp, acc[4 + 2 * (q - 2) + pp], ident);
                    if (q == 3 && pp == 1) { init_bias_pair(cst, SEC_FT, g, 2, acc); init_bias_pair(cst, SEC_FT, g, 3, acc); });
        const f32x4 base = head_finish(hbase, hb);
        Frag D;                                                                // direction-encoding fragment
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                WHS(pp, X[q], 2 + pp, acc,
                    if (!(q & 1) && pp == 0) Y[q >> 1] = frag_of<0>(acc[2 * (q >> 1)], acc[2 * (q >> 1) + 1], acc[0], acc[0]);
                    if ((q & 1) && pp == 0) {
                        if (q == 1) {                                          // direction encoding: slot n = 8g + j
                            const Rev2 t0 = to_revolutions(dv[0]); const Rev2 t1 = to_revolutions(dv[1]); const Rev2 t2 = to_revolutions(dv[2]);
                            const int n0 = 8 * opaque(g);
                            float e[8];
                            _Pragma("unroll")
                            for (int j = 0; j < 8; ++j) {
                                const int n = n0 + j;
                                float sn; float cs;
                                encode_arg(t0.hi, t0.lo, t1.hi, t1.lo, t2.hi, t2.lo, n < 12 ? n : (n < 24 ? n - 12 : 0), &sn, &cs);
                                const float rawv = n == 24 ? dv[0] : (n == 25 ? dv[1] : (n == 26 ? dv[2] : 0.f));
                                e[j] = n < 12 ? sn : (n < 24 ? cs : rawv);
                            }
                            D = split8(e); pin(D.hi); pin(D.lo);
                        } else { init_bias_pair(cst, SEC_VC, g, 0, acc); init_bias_pair(cst, SEC_VC, g, 1, acc); }
                    });

        // ---- views_linears | view_confi_linears on [feature | dir encoding] (pair-split), relu
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const Frag b0 = s < 4 ? Y[s] : D;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                WHS(pp, b0, pp, acc,
                    if (s < 2 && pp == 0) Y[2 + s] = frag_of<0>(acc[4 + 2 * s], acc[5 + 2 * s], acc[0], acc[0]);
                    if (s == 2) init_bias_pair(cst, SEC_VC, g, 2 + pp, acc));
        }
        HeadAcc hadapt = {{0.f, 0.f}, {0.f, 0.f}};
        auto relu = [](float v) { return fmaxf(v, 0.f); };
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const Frag b0 = s < 4 ? Y[s] : D;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                WHS(pp, b0, 2 + pp, acc,
                    if (s < 2) head_part(hadapt, ha, g, 2 * s + pp, acc[2 * s + pp], relu);
                    if (s == 4 && pp == 1) g_pre(tile + tiles_per_round));      // depth and ray of the NEXT tile's sample (after the tile's last advance(); clamped past the end: harmless)
        }
        // ---- adapt heads of row tiles 4..7, uncertainty blend
#pragma unroll
        for (int rt = 4; rt < 8; ++rt) head_part(hadapt, ha, g, rt, acc[rt], relu);
        const f32x4 adapt = head_finish(hadapt, ha);
        {
            const float u = ustash_w[opaque(lane) & 15], omu = 1.f - u;
            const int s_raw = tile * TILE + (opaque(lane) & 15);
            f32x4 out;
            out.x = 1.f / (1.f + expf(-(base.x * omu + adapt.x * u)));
            out.y = 1.f / (1.f + expf(-(base.y * omu + adapt.y * u)));
            out.z = 1.f / (1.f + expf(-(base.z * omu + adapt.z * u)));
            out.w = fmaxf(adapt.w * omu + base.w * u, 0.f);
            if (g == 0 && s_raw < fg.m) reinterpret_cast<f32x4*>(fg.raw)[s_raw] = out;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
}

constexpr size_t smem_bytes() {
    return (size_t)NBUF * SLOT_BYTES + ((CONST_FLOATS * 4 + 15) & ~15) + (size_t)BW * 2 * 64 * sizeof(Frag) + (size_t)BW * 2 * 64 * 8 * sizeof(float)
           + (size_t)BW * 16 * sizeof(float) + 8 * VIEW_TAB * sizeof(float);
}
static_assert(smem_bytes() <= 160 * 1024, "the 16-sample kernel's LDS image must fit the CU");

}  // namespace w16

// called by render.hip (cfg.precision 4): gather + encoding + MLP of one pass in ONE launch on 16-sample tiles, from the channel-last sources and (ray, depth)
int launch_mlp_fwd_bf16x3_gather_w16(const ucnerf_render_params* rp, const float* repacked, const float* dirs, float* raw, hipStream_t st) {
    using namespace w16;
    const long long M = (long long)rp->n * rp->S;
    UCNERF_REQUIRE(M > 0 && M < (1ll << 31), "render (16-sample tiles): %lld samples in one pass", M);
    UCNERF_REQUIRE(dirs && rp->wstream && raw && repacked && rp->conf && rp->rays_o && rp->rays_d && rp->z, "render (16-sample tiles): null pointer");
    UCNERF_REQUIRE(!rp->pts_in && !rp->near_far && !rp->sources_cl_bf16 && !rp->gen_rays, "render (16-sample tiles): derived coordinates, the scene's depth range, fp32 source copies");
    UCNERF_REQUIRE(((uintptr_t)rp->wstream & 15) == 0 && ((uintptr_t)raw & 15) == 0, "render (16-sample tiles): wstream / raw must be 16-byte aligned");
    Layout B;
    UCNERF_REQUIRE(layout(rp->cfg.n_src, &B), "render (16-sample tiles): n_src %d outside 1..8", rp->cfg.n_src);
    Gather f;
    memset(&f, 0, sizeof(f));
    f.m = (int)M; f.S = rp->S; f.V = B.v; f.H = rp->H; f.W = rp->W;
    unsigned long long off = 0;
    for (int k = 0; k < 3; ++k) {
        f.vol_d[k] = rp->vol_d[k]; f.vol_h[k] = rp->vol_h[k]; f.vol_w[k] = rp->vol_w[k];
        f.vol_off[k] = (unsigned)off;
        off += 32ull * rp->vol_d[k] * rp->vol_h[k] * rp->vol_w[k];
    }
    f.img_off = (unsigned)off;
    f.view_bytes = (unsigned)(48ull * rp->H * rp->W);
    off += (unsigned long long)f.view_bytes * f.V;
    UCNERF_REQUIRE(off < (1ull << 32), "render (16-sample tiles): %llu bytes of channel-last sources (limit 4 GB)", off);
    f.cl = reinterpret_cast<const char*>(repacked);
    f.conf = rp->conf; f.rays_o = rp->rays_o; f.rays_d = rp->rays_d; f.z = rp->z; f.dirs = dirs;
    f.near = rp->near; f.far = rp->far;
    memcpy(f.w2c_ref, rp->w2c_ref, sizeof(f.w2c_ref));
    memcpy(f.K_ref, rp->K_ref, sizeof(f.K_ref));
    f.w2cs = rp->w2cs; f.Ks = rp->intrinsics;
    {   // magic for idx / S (gather_cl.hip)
        unsigned l = 1;
        while ((1u << l) < (unsigned)rp->S) ++l;
        f.div_m = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)rp->S - 1) / (unsigned)rp->S);
        f.div_sh = l - 1;
    }
    f.wstream = reinterpret_cast<const char*>(rp->wstream);
    f.raw = raw;
    f.slots = B.halfs / 2; f.const_off_bytes = (int)B.const_off_bytes;
    const int n_tiles = cdiv(M, TILE);
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "render (16-sample tiles): no device");
    int blocks = cdiv(n_tiles, BW);
    const int cap = rp->max_blocks > 0 ? rp->max_blocks : cus;                 // one 8-wave block per CU
    if (blocks < cap) { const int spread = cdiv(n_tiles, 4); blocks = spread < cap ? spread : cap; }      // fewer tiles than wave slots: every CU, one wave per SIMD first
    if (blocks > cap) blocks = cap;
    const size_t smem = smem_bytes();
    dim3 grid(blocks), block(64 * BW);
#define X(N)                                                                                                         \
    if (B.v == N) {                                                                                                  \
        const void* fn = (const void*)mlp_fwd_bf16_w16_kernel<N>;                                                    \
        if (int rc = ensure_dynamic_lds(fn, (int)smem, "mlp_fwd (bf16x3, gather fused, 16-sample tiles)")) return rc; \
        hipLaunchKernelGGL((mlp_fwd_bf16_w16_kernel<N>), grid, block, smem, st, f, n_tiles);                         \
    }
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
    return check_launch("mlp_fwd (bf16x3, gather fused, 16-sample tiles)");
}

}  // namespace ucnerf
