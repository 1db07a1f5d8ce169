// K5+K6, the TRAINING forward with fp32-grade accuracy on the bf16 matrix cores (cfg.precision 5, "bf16x6"): every operand travels as THREE
// bf16 pieces (x = hi + mid + lo exactly: 3 x 8 significant bits), every product a * w is evaluated as the six piece products
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// and the three dropped ones (mid*lo, lo*mid, lo*lo) are <= 2^-24 of the product each -- the size of ONE fp32 rounding.  The exact-fp32
// kernel (mlp.hip) runs its v_mfma_f32_32x32x2_f32 on the vector ALU at 1/16 of the bf16 rate; six bf16 MFMAs per product are 2.7x faster
// than that and leave the vector ALU to the element-wise work.  The two-piece kernel (mlp_bf16.hip, bf16x3: 2^-16) is another 2x faster but
// moves ~1e-5 of the pre-activations across their relu, which the training step's gradient parity does not accept.
//
// Structure: the gradient chain's (mlp_bwd_chain.hip), turned forwards.  One wave per SIMD with 512 registers (the three-piece fragments of a
// layer's input are 96 registers, b_d stays resident in 64), four waves per block, one block per CU, persistent over 32-sample tiles;
// lane = sample, activations live in accumulator registers (register r of row tile kt on lane half h = feature 32 kt + (r & 3) + 8 (r >> 2) + 4 h),
// so a layer's output is the next layer's B operand after the element-wise epilogue and a re-split.  The weight stream (148 STEPS per tile:
// one k16-step for one pair of 32-row tiles = 12 MFMAs on 6 KB of pieces [tile 0, 1][hi, mid, lo][lane][8 bf16]) reaches the four waves through
// ONE eight-slot ring in LDS, filled by global_load_lds eight steps ahead (a wave copies 1 536 bytes of every step: one dwordx4 and two dword
// instructions -- a dwordx3 copy pads every lane's 12 bytes to 16 in LDS, scripts/micro/lds_dma_x3.hip), with counted waits -- see x_advance.  The ten kept activation sets leave in the 24-bit format of p24.h through a per-wave LDS transpose (runs
// of 192 contiguous bytes per row, non-temporal: mlp.hip's save_rows), UNCONDITIONALLY: rows past the last sample land in the slack every
// set has behind its rows (a set is carved as m x 512 bytes and filled with m x 384), so that every wave issues the same vector-memory
// operations and the ring's waits can count them.
#include "common.h"
#include "mlp_layout.h"
#include "sincos_cw.h"
#include "p24.h"

// the LDS-DMA asm below names m0 as a clobber on purpose (it loads the LDS base into it)
#pragma clang diagnostic ignored "-Winline-asm"

namespace ucnerf {
namespace x6 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define XSB __builtin_amdgcn_sched_barrier(0)
#define XMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int XW = 4;                      // waves per block
constexpr int STEP_BYTES = 6144;           // [t 0..1][hi, mid, lo][64 lanes][8 bf16]
constexpr int STEP_FRAGS = STEP_BYTES / 16;
constexpr int XNB = 8;                     // ring slots
constexpr int DMA_PER_STEP = 3;            // global_load_lds per wave and step: 1 024 + 256 + 256 bytes
constexpr int KS_BIAS = 4;                 // k16-steps of a bias net (<= 64 inputs, whatever the view count: its weights past the last input are zero)
// sections, steps = 2 row-tile pairs x k16-steps:   BD 8 | L0 8 | L1..L4 16 each | L5 24 ([pe 4 | h 8]) | BC 8 | FT 16 | VC 20 ([f 8 | dir 2])
constexpr int ST_BD = 0, ST_L0 = 8, ST_L1 = 16, ST_L5 = 80, ST_BC = 104, ST_FT = 112, ST_VC = 128, X_STEPS = 148;
// constants behind the steps: ten bias vectors in feature order (BD, L0..L5, BC, FT, VC), the head table [8][128] (base rgb x3, base sigma, adapt
// rgb x3 on features 0..63, adapt sigma on 64..127) and its eight biases
constexpr int XC_BIAS = 0, XC_HEAD = 10 * 128, XC_HEADB = XC_HEAD + 8 * 128, X_CONST_FLOATS = XC_HEADB + 8;
constexpr int64_t X_STREAM_FLOATS = (int64_t)X_STEPS * STEP_BYTES / 4 + X_CONST_FLOATS;

__host__ __device__ inline int hid_feature16(int kt, int s, int j, int hh) { return 32 * kt + (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * hh; }

// ------------------------------------------------------------------------------------------------ packing (straight from the flat parameters)
enum { XK_FEAT = 0, XK_PE = 1, XK_HID = 2 };
// a section: `ks` k16-steps per pair, the first ks_a of kind_a, the rest of kind_b; row n of the layer = W[base + n * ld ...] (rows >= 64 from base_hi
// when given); columns: FEAT k < K at col0 + k; PE slot -> reference column (pe_column) at col0 + column; HID feature at col0 + feature
struct XSec { int step0, ks, ks_a, kind_a, kind_b, col0_a, col0_b, nf_a, nf_b, ld, K; long long base, base_hi; };
constexpr int X_N_SEC = 11;
struct XPackArgs {
    XSec sec[X_N_SEC];
    const float* flat;
    unsigned short* out;
    float* consts;
    int pe_layout;
    long long p_bias[10], p_vcb, p_crw, p_a1w, p_rw, p_aw, p_crb, p_a1b, p_rb, p_ab;
};

__global__ void __launch_bounds__(256) pack_x6_kernel(XPackArgs a) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < X_STEPS * STEP_FRAGS) {
        const int step = f / STEP_FRAGS, rem = f % STEP_FRAGS, t = rem / 192, plane = (rem / 64) % 3, lane = rem & 63, hh = lane >> 5;
        int si = 0;
#pragma unroll
        for (int i = 1; i < X_N_SEC; ++i) si = step >= a.sec[i].step0 ? i : si;
        const XSec sc = a.sec[si];
        const int ls = step - sc.step0, pair = ls / sc.ks, q = ls % sc.ks;
        const bool pa = q < sc.ks_a;
        const int kind = pa ? sc.kind_a : sc.kind_b, qq = pa ? q : q - sc.ks_a, col0 = pa ? sc.col0_a : sc.col0_b, nf = pa ? sc.nf_a : sc.nf_b;
        const int row = 32 * (2 * pair + t) + (lane & 31);
        const long long rbase = (sc.base_hi >= 0 && row >= 64) ? sc.base_hi + (long long)(row - 64) * sc.ld : sc.base + (long long)row * sc.ld;
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int col = -1;
            if (kind == XK_FEAT) { const int k = 16 * qq + 8 * hh + e; col = k < sc.K ? k : -1; }
            else if (kind == XK_HID) col = hid_feature16(qq >> 1, qq & 1, e, hh);
            else {
                int pk, pa_;
                pe_slot(8 * qq + e, hh, nf, &pk, &pa_);
                col = pe_column(pk, pa_, nf, a.pe_layout);
            }
            const float w = col >= 0 ? a.flat[rbase + col0 + col] : 0.f;
            const __bf16 hi = (__bf16)w;
            const float r1 = w - (float)hi;                       // (every residual is exact in fp32)
            const __bf16 mid = (__bf16)r1;
            const __bf16 lo = (__bf16)(r1 - (float)mid);
            v[e] = __builtin_bit_cast(unsigned short, plane == 0 ? hi : plane == 1 ? mid : lo);
        }
        u32x4 o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16)};
        reinterpret_cast<u32x4*>(a.out)[f] = o;
    }
    if (f < X_CONST_FLOATS) {
        float w = 0.f;
        if (f < XC_HEAD) {
            const int sec = f >> 7, k = f & 127;
            w = sec == 9 && k >= 64 ? a.flat[a.p_vcb + k - 64] : a.flat[a.p_bias[sec] + k];
        } else if (f < XC_HEADB) {
            const int o = (f - XC_HEAD) >> 7, k = f & 127;
            if (o < 3) w = a.flat[a.p_crw + o * 128 + k];
            else if (o == 3) w = a.flat[a.p_a1w + k];
            else if (o < 7) w = k < 64 ? a.flat[a.p_rw + (o - 4) * 64 + k] : 0.f;       // rgb_linear reads the views half of vc
            else w = k >= 64 ? a.flat[a.p_aw + k - 64] : 0.f;                           // alpha_linear the view_confi half
        } else {
            const int o = f - XC_HEADB;
            w = o < 3 ? a.flat[a.p_crb + o] : o == 3 ? a.flat[a.p_a1b] : o < 7 ? a.flat[a.p_rb + o - 4] : a.flat[a.p_ab];
        }
        a.consts[f] = w;
    }
}

// ------------------------------------------------------------------------------------------------ device helpers
struct Frag3 { bf16x8 hi, mid, lo; };
// three pieces of eight values: hi and mid truncated (their residuals are exact), lo rounded: 8 + 8 + 8 significant bits
__device__ __forceinline__ Frag3 split8_3(const float (&x)[8]) {
    u32x4 hi, mid;
    Frag3 f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[j]), b1 = __builtin_bit_cast(unsigned, x[j + 1]);
        hi[j >> 1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const f32x2 r = (f32x2){x[j], x[j + 1]} - (f32x2){__builtin_bit_cast(float, b0 & 0xffff0000u), __builtin_bit_cast(float, b1 & 0xffff0000u)};
        const unsigned c0 = __builtin_bit_cast(unsigned, r.x), c1 = __builtin_bit_cast(unsigned, r.y);
        mid[j >> 1] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
        const f32x2 r2 = r - (f32x2){__builtin_bit_cast(float, c0 & 0xffff0000u), __builtin_bit_cast(float, c1 & 0xffff0000u)};
        f.lo[j] = (__bf16)r2.x;
        f.lo[j + 1] = (__bf16)r2.y;
    }
    f.hi = __builtin_bit_cast(bf16x8, hi);
    f.mid = __builtin_bit_cast(bf16x8, mid);
    return f;
}
// the two k16-steps an accumulator tile supplies
__device__ __forceinline__ void split_tile(const f32x16& x, Frag3& f0, Frag3& f1) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = x[e];
    f0 = split8_3(t);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = x[8 + e];
    f1 = split8_3(t);
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// Positional encoding of a 3-vector in the slot order of mlp_layout.h's pe_slot (this lane half's 3 NF + 2 slots; mlp.hip's encode)
template <int NF, int KS>
__device__ __forceinline__ void encode(const float (&x)[3], int h, float (&pe)[KS]) {
    constexpr int half = 3 * NF / 2;
    const Rev2 t[3] = {to_revolutions(x[0]), to_revolutions(x[1]), to_revolutions(x[2])};
#pragma unroll
    for (int q = 0; q < half; ++q) {                       // argument a = 2q + h: frequency a / 3, coordinate a % 3 (selected by h)
        const int a0 = 2 * q, a1 = 2 * q + 1, c0 = a0 % 3, c1 = a1 % 3;
        const float h0 = c0 == 0 ? t[0].hi : c0 == 1 ? t[1].hi : t[2].hi, l0 = c0 == 0 ? t[0].lo : c0 == 1 ? t[1].lo : t[2].lo;
        const float h1 = c1 == 0 ? t[0].hi : c1 == 1 ? t[1].hi : t[2].hi, l1 = c1 == 0 ? t[0].lo : c1 == 1 ? t[1].lo : t[2].lo;
        const float hi = h ? h1 : h0, lo = h ? l1 : l0;
        const float scale = h ? (float)(1 << (a1 / 3)) : (float)(1 << (a0 / 3));
        sincos_rev(hi, lo, scale, &pe[q], &pe[half + q]);
    }
    pe[2 * half] = h ? x[2] : x[0];
    pe[2 * half + 1] = h ? 0.f : x[1];
#pragma unroll
    for (int q = 2 * half + 2; q < KS; ++q) pe[q] = 0.f;
}

// ---- the weight ring
struct XPipe {
    const char* gsrc;        // this wave's quarter of step 0, this lane's 16 bytes of its first kilobyte: stream + wave * 1536 + lane * 16
    unsigned dst;            // LDS byte address of this wave's quarter of slot 0
    const char* rd;          // this lane's 16 bytes of slot 0, plane 0 (LDS)
};
struct XAF { bf16x8 h0, m0, l0, h1, m1, l1; };
// (issued from inline asm: the compiler treats a global_load_lds as an access to both memories and degrades every counted wait around it)
__device__ __forceinline__ void x_dma(const XPipe& P, int step_src, int slot) {
    const char* src = P.gsrc + (size_t)step_src * STEP_BYTES;
    const unsigned dst = P.dst + slot * STEP_BYTES;
    const char* src4 = src + 1024 - (threadIdx.x & 63) * 12;      // (this lane's 4 bytes of the quarter's last 512: + lane * 4 instead of + lane * 16)
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(src4), "s"(dst + 1024) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(src4 + 256), "s"(dst + 1280) : "memory", "m0");
}
__device__ __forceinline__ XAF x_ldaf(const XPipe& P, int slot) {
    const bf16x8* a = reinterpret_cast<const bf16x8*>(P.rd + slot * STEP_BYTES);
    XAF f;
    f.h0 = a[0]; f.m0 = a[64]; f.l0 = a[128]; f.h1 = a[192]; f.m1 = a[256]; f.l1 = a[320];
    return f;
}
// Start of step `pos` of the running ring (its fragments are in `cur`): the next step's slot has landed for the whole block and everybody is
// done reading this step's slot, which is refilled with the step XNB ahead (`src_next`: its index in the stream); returns the next step's fragments.
// vmcnt: vector-memory operations retire in issue order.  Behind the awaited copy this wave has issued the copies of the XNB - 2 steps after
// it and -- in the first XNB - 1 steps of a section -- the loads and stores of the epilogue in front of the section: YOUNGER = a lower bound
// of their number, known at compile time because every wave issues every one of them unconditionally (tests/test_abi_host.py replays the
// compiled kernel's vector-memory stream against these counts).
template <int YOUNGER>
__device__ __forceinline__ XAF x_advance(const XPipe& P, int src_next, int& pos) {
    static_assert(YOUNGER >= DMA_PER_STEP * (XNB - 2) && YOUNGER <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(YOUNGER) : "memory");
    __builtin_amdgcn_s_barrier();
    x_dma(P, src_next, pos & (XNB - 1));
    ++pos;
    return x_ldaf(P, pos & (XNB - 1));
}
constexpr int x_younger(int epi) { return DMA_PER_STEP * (XNB - 2) + epi > 63 ? 63 : DMA_PER_STEP * (XNB - 2) + epi; }

// a layer: two row-tile pairs x (KSA + KSB) k16-steps, the first KSA on the fragments Ba, the rest on Bb; starts at step STEP0 of the stream;
// EPI = vector-memory operations every wave has issued between the previous section and this one.  Small products first.
template <int STEP0, int KSA, int KSB, int EPI>
__device__ __forceinline__ void x_section(const XPipe& P, XAF& cur, int& pos, const Frag3* Ba, const Frag3* Bb, f32x16 (&acc)[4]) {
    constexpr int KS = KSA + KSB;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            XSB;              // (one step at a time)
            const int i = KS * p + q;
            const XAF nxt = i < XNB - 1 ? x_advance<x_younger(EPI)>(P, (STEP0 + i + XNB) % X_STEPS, pos) : x_advance<DMA_PER_STEP * (XNB - 2)>(P, (STEP0 + i + XNB) % X_STEPS, pos);
            const Frag3& b = q < KSA ? Ba[q] : Bb[q - KSA];
            f32x16 c0 = acc[2 * p], c1 = acc[2 * p + 1];
            c0 = XMFMA(cur.m0, b.mid, c0); c1 = XMFMA(cur.m1, b.mid, c1);
            c0 = XMFMA(cur.h0, b.lo, c0);  c1 = XMFMA(cur.h1, b.lo, c1);
            c0 = XMFMA(cur.l0, b.hi, c0);  c1 = XMFMA(cur.l1, b.hi, c1);
            c0 = XMFMA(cur.m0, b.hi, c0);  c1 = XMFMA(cur.m1, b.hi, c1);
            c0 = XMFMA(cur.h0, b.mid, c0); c1 = XMFMA(cur.h1, b.mid, c1);
            c0 = XMFMA(cur.h0, b.hi, c0);  c1 = XMFMA(cur.h1, b.hi, c1);
            acc[2 * p] = c0; acc[2 * p + 1] = c1;
            cur = nxt;
        }
}

constexpr int XPOSE_ROW = 192 + 16;                   // LDS row stride of the transpose buffer (the pad spreads the rows over the banks)
constexpr int XPOSE_BYTES = 32 * XPOSE_ROW;
constexpr int X_LDS_CONST = X_CONST_FLOATS * 4;
constexpr int X_LDS_RING = (X_LDS_CONST + 15) & ~15;
constexpr int X_LDS_XPOSE = X_LDS_RING + XNB * STEP_BYTES;
constexpr int X_LDS_BYTES = X_LDS_XPOSE + XW * XPOSE_BYTES;       // 9 + 48 + 26 KB
constexpr int SAVE_STORES = 12;                       // vector-memory instructions of one kept set (x_save)

// accumulator-layout activation set -> rows of the 24-bit format (p24.h): the wave transposes half a set at a time (64 columns = 192 bytes of each
// of its 32 rows) through its LDS buffer and writes it out as six 1-KB instructions that walk the rows' bytes in order (mlp.hip's save_rows).
// No predicate: rows past the last sample land in the set's slack (see the header).
__device__ __forceinline__ void x_save(float* set, const f32x16 (&x)[4], char* xbuf, int lane_, int tile) {
    const int lane = opaque(lane_);                   // (the chunk addresses do not depend on the tile: unlaundered they are hoisted out of the tile loop)
    const int j = lane & 31, h = lane >> 5;
    char* const gtile = reinterpret_cast<char*>(set) + (size_t)tile * 32 * P24_ROW_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nt = 2 * half + t;
                const P24Piece pc = p24_pack4(x[nt][4 * q], x[nt][4 * q + 1], x[nt][4 * q + 2], x[nt][4 * q + 3]);
                *reinterpret_cast<p24_u32x3_a4*>(xbuf + j * XPOSE_ROW + 12 * (8 * t + 2 * q + h)) = (p24_u32x3){pc.d[0], pc.d[1], pc.d[2]};
            }
        // (LDS operations of a wave execute in order: the reads below see the writes above, and the next half's writes come after these reads)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            XSB;                                                  // (one chunk at a time: four registers, not twenty-four)
            const int c = k * 64 + lane;                          // 16-byte chunk of the half set: row c / 12, chunk c % 12 of its 192 bytes
            const int row = (c * 2731) >> 15, within = c - 12 * row;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xbuf + row * XPOSE_ROW + 16 * within);
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(gtile + row * P24_ROW_BYTES + 192 * half + 16 * within));
        }
    }
}

struct XArgs {
    int m, n_tiles, F, ldf, feats_tiled, n_mvs, S, dirs_per_sample;
    const float* pts; const float* dirs; const float* feats;
    float* raw;
    const char* wstream;           // pack_x6_kernel's steps ...
    const float* consts;           // ... and constants
    MlpSaved sv;
};

__global__ void __launch_bounds__(64 * XW, 1) mlp_fwd_x6_kernel(XArgs a) {
    extern __shared__ __attribute__((aligned(16))) float x_lds[];
    float* const cst = x_lds;
    for (int i = threadIdx.x; i < X_CONST_FLOATS; i += 64 * XW) cst[i] = a.consts[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    char* const lds = reinterpret_cast<char*>(x_lds);
    char* const xbuf = lds + X_LDS_XPOSE + wave * XPOSE_BYTES;

    XPipe P;
    P.gsrc = a.wstream + wave * (STEP_BYTES / XW) + lane * 16;
    P.dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + X_LDS_RING) + wave * (STEP_BYTES / XW);
    P.rd = lds + X_LDS_RING + lane * 16;
    // (the stream does not depend on the tile: unless its address is laundered per section, its fragment addresses are hoisted out of the tile loop and spilled)
    auto launder = [&]() { unsigned o_ = (unsigned)(wave * (STEP_BYTES / XW) + lane * 16); asm volatile("" : "+v"(o_)); P.gsrc = a.wstream + o_; };
    int pos = 0;
#pragma unroll
    for (int i = 0; i < XNB; ++i) x_dma(P, i, i);                   // steps 0 .. XNB - 1 of the first tile
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STEP * (XNB - 1)) : "memory");      // slot 0 has landed ...
    __builtin_amdgcn_s_barrier();                                                         // ... for every wave
    XAF cur = x_ldaf(P, 0);

    // accumulators <- bias vector `sec` (accumulator layout: four consecutive features per read)
    auto init_bias = [&](int sec, f32x16 (&acc)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(&cst[opaque(XC_BIAS + sec * 128) + 32 * nt + 8 * q + 4 * h]);
                acc[nt][4 * q] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
            }
    };
    // four head outputs of an activation set: this lane's 64 features, then both lane halves (rows o0 .. o0 + 3 of the head table)
    auto heads = [&](int o0, const f32x16 (&x)[4]) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 w[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) w[o] = *reinterpret_cast<const f32x4*>(&cst[opaque(XC_HEAD + (o0 + o) * 128) + 32 * nt + 8 * q + 4 * h]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int o = 0; o < 4; ++o) s[o] = __builtin_fmaf(x[nt][4 * q + c], w[o][c], s[o]);
                XSB;                                   // (a fence per group: the scheduler otherwise reads every head weight up front and spills them)
            }
        f32x4 r;
#pragma unroll
        for (int o = 0; o < 4; ++o) { s[o] += __shfl_xor(s[o], 32); s[o] += cst[opaque(XC_HEADB + o0 + o)]; }
        r.x = s[0]; r.y = s[1]; r.z = s[2]; r.w = s[3];
        return r;
    };

    // Every wave of the block walks the same number of tiles (the ring is turned by all four together): a wave whose tile lies past the end
    // computes the last tile once more and stores the same bytes.
    const int rounds = (a.n_tiles + gridDim.x * XW - 1) / (gridDim.x * XW);
    for (int rnd = 0; rnd < rounds; ++rnd) {
        const int tile_ = (rnd * gridDim.x + blockIdx.x) * XW + wave;
        const int tile = tile_ < a.n_tiles ? tile_ : a.n_tiles - 1;
        const int s_raw = tile * 32 + j;
        const bool valid = s_raw < a.m;
        const int s = valid ? s_raw : a.m - 1;

        // the B operand of a bias net: k16-step q of this lane = columns col0 + 16 q + 8 h + 0..7 of its sample's gathered features (four steps
        // whatever the view count: every wave issues the same 32 loads; columns past the row are clamped -- their weights are zero in the stream)
        auto ld_feats = [&](int col0, float (&f)[4][8]) {
            const int s_ = opaque(s), h_ = opaque(h);         // (laundered: the column offsets do not depend on the tile -- hoisted out of the tile loop they are 64 registers)
            const size_t f_base = a.feats_tiled ? (size_t)(s_ >> 5) * a.F * 32 + (s_ & 31) : (size_t)s_ * a.ldf;
            const int f_cs = a.feats_tiled ? 32 : 1;
            const float* const fb = a.feats + f_base;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int col_ = col0 + 16 * q + 8 * h_ + e, col = col_ < a.F ? col_ : a.F - 1;
                    f[q][e] = fb[col * f_cs];
                }
        };

        // ---- (1) everything the tile reads from memory except the image features: 3 + 3 + 1 + 32 loads
        float px[3], dv[3];
        {
            const float* prow = a.pts + (size_t)s * 3;
            px[0] = prow[0]; px[1] = prow[1]; px[2] = prow[2];
            const float* drow = a.dirs + (a.dirs_per_sample ? (size_t)s : (size_t)(s / a.S)) * 3;
            dv[0] = drow[0]; dv[1] = drow[1]; dv[2] = drow[2];
        }
        const float conf = a.feats_tiled ? a.feats[((size_t)(s >> 5) * a.F + a.F - 1) * 32 + (s & 31)] : a.feats[(size_t)s * a.ldf + a.F - 1];
        float fm[4][8];
        ld_feats(0, fm);

        // ---- (2) point encoding, as three-piece fragments for layers 0 and 5
        Frag3 PE[4];
        {
            float pe[KS_PE_PTS];
            encode<10, KS_PE_PTS>(px, h, pe);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = pe[8 * q + e];
                PE[q] = split8_3(t);
            }
        }
        XSB;

        f32x16 bd[4], acc[4], hin[4];
        Frag3 X[8];

        // ---- depth-bias net: bd = W_d [volume feats | colours + masks] + b      (models.py:150)
        {
            Frag3 Bm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) Bm[q] = split8_3(fm[q]);
            init_bias(0, bd);
            launder(); x_section<ST_BD, KS_BIAS, 0, 3 + 32>(P, cur, pos, Bm, Bm, bd)      /* the tile's loads: point, direction, confidence (one instruction each at least), 32 features; the previous tile's last set is not counted -- the first tile has none */;
        }
        x_save(a.sv.bd, bd, xbuf, lane, tile);

        // ---- layer 0 on the point encoding
        auto trunk_epi = [&](float* set) {                   // h = relu(y * b_d), kept, and re-split as the next layer's operand
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 m_ = (f32x2){acc[nt][r], acc[nt][r + 1]} * (f32x2){bd[nt][r], bd[nt][r + 1]};
                    hin[nt][r] = fmaxf(m_.x, 0.f); hin[nt][r + 1] = fmaxf(m_.y, 0.f);
                }
            x_save(set, hin, xbuf, lane, tile);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) split_tile(hin[nt], X[2 * nt], X[2 * nt + 1]);
            XSB;
        };
        init_bias(1, acc);
        launder(); x_section<ST_L0, 4, 0, SAVE_STORES>(P, cur, pos, PE, PE, acc);
        trunk_epi(a.sv.h[0]);

        // ---- layers 1..4                                                        (models.py:153-155)
#define UCNERF_X6_LAYER(L)                                                                              \
        init_bias(1 + (L), acc);                                                                        \
        launder(); x_section<ST_L1 + 16 * ((L) - 1), 8, 0, SAVE_STORES>(P, cur, pos, X, X, acc);        \
        trunk_epi(a.sv.h[(L)]);
        UCNERF_X6_LAYER(1) UCNERF_X6_LAYER(2) UCNERF_X6_LAYER(3)
        // (layer 4's epilogue is followed by the image-feature loads: they land under layer 5)
        UCNERF_X6_LAYER(4)
#undef UCNERF_X6_LAYER
        float fi[4][8];
        ld_feats(a.n_mvs, fi);

        // ---- layer 5 on [pe | h4]                                               (models.py:156-157)
        init_bias(6, acc);
        launder(); x_section<ST_L5, 4, 8, SAVE_STORES + 32>(P, cur, pos, PE, X, acc);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 m_ = (f32x2){acc[nt][r], acc[nt][r + 1]} * (f32x2){bd[nt][r], bd[nt][r + 1]};
                hin[nt][r] = fmaxf(m_.x, 0.f); hin[nt][r + 1] = fmaxf(m_.y, 0.f);
            }
        x_save(a.sv.h[5], hin, xbuf, lane, tile);

        // ---- base heads: confi_rgb_linear, alpha_linear_1                       (models.py:161-162)
        const f32x4 base = heads(0, hin);

        // ---- confidence-bias net, g = h5 * b_c                                  (models.py:151,164)
        {
            Frag3 Bi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) Bi[q] = split8_3(fi[q]);
            init_bias(7, acc);
            launder(); x_section<ST_BC, KS_BIAS, 0, SAVE_STORES>(P, cur, pos, Bi, Bi, acc);
        }
        x_save(a.sv.bc, acc, xbuf, lane, tile);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 m_ = (f32x2){hin[nt][r], hin[nt][r + 1]} * (f32x2){acc[nt][r], acc[nt][r + 1]};
                hin[nt][r] = m_.x; hin[nt][r + 1] = m_.y;
            }
            split_tile(hin[nt], X[2 * nt], X[2 * nt + 1]);
        }
        XSB;

        // ---- feature_linear                                                     (models.py:164)
        init_bias(8, acc);
        launder(); x_section<ST_FT, 8, 0, SAVE_STORES>(P, cur, pos, X, X, acc);
        x_save(a.sv.ft, acc, xbuf, lane, tile);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) split_tile(acc[nt], X[2 * nt], X[2 * nt + 1]);
        Frag3 PD[2];
        {
            float pd[KS_PE_DIR];
            encode<4, KS_PE_DIR>(dv, h, pd);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = pd[8 * q + e];
                PD[q] = split8_3(t);
            }
        }
        XSB;

        // ---- views_linears | view_confi_linears on [feature | dir encoding], relu   (models.py:166-173)
        init_bias(9, hin);
        launder(); x_section<ST_VC, 8, 2, SAVE_STORES>(P, cur, pos, X, PD, hin);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) hin[nt][r] = fmaxf(hin[nt][r], 0.f);
        x_save(a.sv.vc, hin, xbuf, lane, tile);

        // ---- adapt heads (rgb_linear on features 0..63, alpha_linear on 64..127), uncertainty blend   (models.py:149,175-178)
        const f32x4 adapt = heads(4, hin);
        const float u = 1.f - conf, omu = 1.f - u;
        f32x4 out;
        out.x = 1.f / (1.f + expf(-(base.x * omu + adapt.x * u)));
        out.y = 1.f / (1.f + expf(-(base.y * omu + adapt.y * u)));
        out.z = 1.f / (1.f + expf(-(base.z * omu + adapt.z * u)));
        out.w = fmaxf(adapt.w * omu + base.w * u, 0.f);
        if (h == 0 && valid) reinterpret_cast<f32x4*>(a.raw)[s_raw] = out;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the copies still in flight land before the block gives its LDS back
}

}  // namespace x6

// ------------------------------------------------------------------------------------------------ host side
int64_t x6_stream_floats() { return x6::X_STREAM_FLOATS; }

int launch_pack_x6(const ucnerf_mlp_config* cfg, const float* flat, float* out, hipStream_t st) {
    using namespace x6;
    MlpLayout L;
    UCNERF_REQUIRE(mlp_layout(cfg->n_src, &L), "mlp_pack: n_src %d outside 1..8", cfg->n_src);
    const int v = L.v, KV = MLP_W + MLP_PE_DIR, n_mvs = 24 + 4 * v, n_img = 8 * v;
    XPackArgs a;
    memset(&a, 0, sizeof(a));
    int step = 0, i = 0;
    auto sec = [&](int ks_a, int kind_a, int col0_a, int nf_a, int ks_b, int kind_b, int col0_b, int nf_b, long long base, int ld, int K, long long base_hi) {
        XSec& s = a.sec[i];
        s.step0 = step; s.ks = ks_a + ks_b; s.ks_a = ks_a; s.kind_a = kind_a; s.kind_b = kind_b; s.col0_a = col0_a; s.col0_b = col0_b; s.nf_a = nf_a; s.nf_b = nf_b;
        s.ld = ld; s.K = K; s.base = base; s.base_hi = base_hi;
        step += 2 * s.ks; ++i;
    };
    sec(KS_BIAS, XK_FEAT, 0, 0, 0, XK_FEAT, 0, 0, L.p_bdw, n_mvs, n_mvs, -1);                  // depth-bias net
    sec(4, XK_PE, 0, 10, 0, XK_PE, 0, 10, L.p_lw[0], MLP_PE_PTS, 0, -1);                        // layer 0 on the point encoding
    for (int l = 1; l <= 4; ++l) sec(8, XK_HID, 0, 0, 0, XK_HID, 0, 0, L.p_lw[l], 128, 0, -1);
    sec(4, XK_PE, 0, 10, 8, XK_HID, MLP_PE_PTS, 0, L.p_lw[5], 128 + MLP_PE_PTS, 0, -1);         // layer 5 on [pe | h4]
    sec(KS_BIAS, XK_FEAT, 0, 0, 0, XK_FEAT, 0, 0, L.p_bcw, n_img, n_img, -1);                  // confidence-bias net
    sec(8, XK_HID, 0, 0, 0, XK_HID, 0, 0, L.p_fw, 128, 0, -1);                                  // feature_linear
    sec(8, XK_HID, 0, 0, 2, XK_PE, 128, 4, L.p_vw, KV, 0, L.p_vcw);                             // [views | view_confi] on [feature | dir encoding]
    if (step != X_STEPS || i != X_N_SEC - 1) return fail(UCNERF_EINVAL, "mlp_pack (bf16x6): stream schedule mismatch");
    a.sec[i] = a.sec[i - 1]; a.sec[i].step0 = 1 << 30;          // (sentinel: the kernel's search runs over X_N_SEC entries)
    a.flat = flat;
    a.out = reinterpret_cast<unsigned short*>(out);
    a.consts = out + (size_t)X_STEPS * STEP_BYTES / 4;
    a.pe_layout = cfg->pe_layout;
    a.p_bias[0] = L.p_bdb;
    for (int l = 0; l < 6; ++l) a.p_bias[1 + l] = L.p_lb[l];
    a.p_bias[7] = L.p_bcb; a.p_bias[8] = L.p_fb; a.p_bias[9] = L.p_vb; a.p_vcb = L.p_vcb;
    a.p_crw = L.p_crw; a.p_a1w = L.p_a1w; a.p_rw = L.p_rw; a.p_aw = L.p_aw; a.p_crb = L.p_crb; a.p_a1b = L.p_a1b; a.p_rb = L.p_rb; a.p_ab = L.p_ab;
    hipLaunchKernelGGL(pack_x6_kernel, dim3(cdiv(X_STEPS * STEP_FRAGS, 256)), dim3(256), 0, st, a);
    return check_launch("mlp_pack (bf16x6)");
}

// the training forward of the gradient chain (24-bit sets): `save` as for launch_mlp_fwd
int launch_mlp_fwd_x6(const ucnerf_mlp_params* p, const MlpSaved* save, hipStream_t st) {
    using namespace x6;
    UCNERF_REQUIRE(p && save, "mlp_fwd (bf16x6): null params");
    if (p->m == 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->pts && p->dirs && p->feats && p->wstream && p->raw, "mlp_fwd (bf16x6): null pointer");
    MlpLayout L;
    UCNERF_REQUIRE(mlp_layout(p->cfg.n_src, &L), "mlp_fwd (bf16x6): n_src %d outside 1..8", p->cfg.n_src);
    UCNERF_REQUIRE(save->p24, "mlp_fwd (bf16x6): the three-piece training forward keeps its activation sets as 24-bit floats (bwd_mode 0)");
    UCNERF_REQUIRE(!p->encoded && (p->pts_stride == 0 || p->pts_stride == 3) && (p->dirs_stride == 0 || p->dirs_stride == 3),
                   "mlp_fwd (bf16x6): raw, densely stored points and directions only");
    UCNERF_REQUIRE(p->m >= 128, "mlp_fwd (bf16x6): %d samples -- the kept sets' slack holds the last tile's overhang from 128 samples on; use precision 0 below", p->m);
    UCNERF_REQUIRE(p->dirs_per_sample || p->S > 0, "mlp_fwd (bf16x6): S must be > 0 when dirs are per ray");
    UCNERF_REQUIRE(((uintptr_t)p->wstream & 15) == 0 && ((uintptr_t)p->raw & 15) == 0, "mlp_fwd (bf16x6): wstream/raw must be 16-byte aligned");
    XArgs a;
    memset(&a, 0, sizeof(a));
    a.m = p->m; a.n_tiles = cdiv(p->m, 32); a.F = L.F; a.ldf = p->feat_stride ? p->feat_stride : L.F; a.feats_tiled = p->feats_tiled;
    a.n_mvs = 24 + 4 * L.v; a.S = p->S > 0 ? p->S : 1; a.dirs_per_sample = p->dirs_per_sample;
    a.pts = p->pts; a.dirs = p->dirs; a.feats = p->feats; a.raw = p->raw;
    a.wstream = reinterpret_cast<const char*>(p->wstream);
    a.consts = p->wstream + (size_t)X_STEPS * STEP_BYTES / 4;
    a.sv = *save;
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_fwd (bf16x6): no device");
    int blocks = cdiv(a.n_tiles, XW);
    if (p->max_blocks > 0 && blocks > p->max_blocks) blocks = p->max_blocks;
    if (blocks > cus) blocks = cus;                        // one 4-wave block per CU (512 registers per wave), persistent
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_fwd_x6_kernel), X_LDS_BYTES, "mlp_fwd (bf16x6)")) return rc;
    hipLaunchKernelGGL(mlp_fwd_x6_kernel, dim3(blocks), dim3(64 * XW), X_LDS_BYTES, st, a);
    return check_launch("mlp_fwd (bf16x6)");
}

}  // namespace ucnerf
