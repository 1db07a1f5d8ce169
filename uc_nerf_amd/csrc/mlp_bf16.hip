// K5+K6 on the bf16 matrix cores with fp32-grade accuracy: every product a*w is evaluated as
//     a_hi*w_hi + a_hi*w_lo + a_lo*w_hi      (x_hi = bf16(x), x_lo = bf16(x - x_hi); fp32 accumulate)
// with v_mfma_f32_32x32x16_bf16.  The dropped terms are O(2^-16) relative: rendered colours / depths stay within
// 1e-5 of the fp64 result (tests pin 1e-4), while a plain bf16 evaluation is off by 3e-3.
//
// Why: on gfx950 the fp32 MFMA executes on the vector ALU (VALU fillers cost their full issue time, a partner
// wave's VALU starves while fp32 MFMAs stream -- scripts/micro/mfma_fillers.hip), so the fp32 kernel's ceiling is
// the 157 TFLOP/s vector rate minus all element-wise work.  bf16 MFMAs run on the separate matrix cores at 16x
// the rate: three of them per product are still > 5x faster, and the splitting / epilogue VALU overlaps them.
//
// Structure (same transposed formulation as mlp.hip: lane = sample, activations live in accumulator registers):
//   * a k16-step = 16 input features: lane-half hh supplies 8 of them as one bf16x8 fragment.  For hidden layers
//     the fragment of step (kt, s) is accumulator registers 8s..8s+7 of row-tile kt -- the accumulator layout is
//     again directly the next layer's operand, only re-split into (hi, lo) after the epilogue;
//   * weights are consumed 3x faster than in the fp32 kernel, too fast to stream per wave from L2, so the waves
//     of a block share them: the packed stream ([step][row-tile][hi|lo][lane][8 bf16], 8 KB per step) is copied
//     into a 4-slot LDS ring by global_load_lds (two 1-KB pieces per wave per step, issued 3.5 steps ahead,
//     counted vmcnt + one raw s_barrier per step), and every wave reads its A fragments with ds_read_b128, one
//     row-tile pair (6 MFMAs) ahead of the MFMAs that consume them;
//   * a block is 4 waves = one per SIMD, and TWO blocks share a CU: the two waves of a SIMD belong to different
//     blocks, are never coupled by a barrier and drift apart, so one wave's epilogue (bias-net product, relu,
//     hi/lo split: ~45 % of its MFMA time, on the VALU) runs under the other wave's MFMAs.
#include "common.h"
#include "mlp_layout.h"
#include "sincos_cw.h"

#include <cstdlib>
#include <vector>

namespace ucnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int BW = 4;                 // waves per block (one per SIMD; two blocks per CU)
constexpr int STEP_BYTES = 8192;      // [4 row-tiles][hi, lo][64 lanes][16 B]
constexpr int NBUF = 4;               // LDS ring slots (k16-steps); the slot of step g is refilled with step g + NBUF
constexpr int DMA_PER_STEP = STEP_BYTES / 1024 / BW;      // 1-KB global_load_lds pieces per wave per step
constexpr int KS16_PE_PTS = 4, KS16_PE_DIR = 2, KS16_HID = 8;

struct Bf16Layout {
    int v, F, kd16, kc16, steps;
    int sec_step[N_SEC];              // first step of each section (execution order = section order of mlp_layout.h)
    int64_t const_off_bytes, total_bytes;
};

bool bf16_layout(int v, Bf16Layout* B) {
    if (v < 1 || v > 8) return false;
    B->v = v; B->F = 24 + 12 * v + 1;
    B->kd16 = (24 + 4 * v + 15) / 16; B->kc16 = (8 * v + 15) / 16;
    const int ks[N_SEC] = {B->kd16, KS16_PE_PTS, KS16_HID, KS16_HID, KS16_HID, KS16_HID, KS16_PE_PTS + KS16_HID, B->kc16, KS16_HID,
                           KS16_HID + KS16_PE_DIR};
    int s = 0;
    for (int i = 0; i < N_SEC; ++i) { B->sec_step[i] = s; s += ks[i]; }
    B->steps = s;
    B->const_off_bytes = (int64_t)s * STEP_BYTES;
    B->total_bytes = B->const_off_bytes + (int64_t)CONST_FLOATS * 4;
    return true;
}

// feature held by element j of lane-half hh in hidden k16-step (kt, s): accumulator register 8s + j of row-tile kt
__host__ __device__ inline int hid_feature16(int kt, int s, int j, int hh) { return 32 * kt + (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * hh; }

// ------------------------------------------------------------------------------------------------ host: pack index
// idx16[e] for every bf16 element e of the stream: flat parameter index | (part << 30) (part 0 = hi, 1 = lo), -1 = zero.
int build_pack_index_bf16(const ucnerf_mlp_config* cfg, int32_t* idx) {
    Bf16Layout B;
    MlpLayout L;
    if (!bf16_layout(cfg->n_src, &B) || !mlp_layout(cfg->n_src, &L)) return -1;
    const int v = B.v, W = MLP_W;
    const int64_t n16 = (int64_t)B.steps * (STEP_BYTES / 2);
    for (int64_t i = 0; i < n16 + CONST_FLOATS; ++i) idx[i] = -1;
    auto put_step = [&](int step, const std::vector<int64_t>& row_base, const int (&col)[2][8]) {
        for (int nt = 0; nt < 4; ++nt)
            for (int part = 0; part < 2; ++part)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int c = col[lane >> 5][j];
                        const int64_t e = ((((int64_t)step * 4 + nt) * 2 + part) * 64 + lane) * 8 + j;
                        idx[e] = c < 0 ? -1 : (int32_t)((row_base[32 * nt + (lane & 31)] + c) | ((int64_t)part << 30));
                    }
    };
    auto rows = [&](int64_t base, int K) { std::vector<int64_t> rb(128); for (int n = 0; n < 128; ++n) rb[n] = base + (int64_t)n * K; return rb; };
    auto nat = [&](int q, int K, int (&col)[2][8]) { for (int hh = 0; hh < 2; ++hh) for (int j = 0; j < 8; ++j) { int f = 16 * q + 8 * hh + j; col[hh][j] = f < K ? f : -1; } };
    auto hid = [&](int q, int base, int (&col)[2][8]) { for (int hh = 0; hh < 2; ++hh) for (int j = 0; j < 8; ++j) col[hh][j] = base + hid_feature16(q >> 1, q & 1, j, hh); };
    auto pe = [&](int q, int nf, int base, int (&col)[2][8]) {
        for (int hh = 0; hh < 2; ++hh)
            for (int j = 0; j < 8; ++j) {
                int kind, a;
                pe_slot(8 * q + j, hh, nf, &kind, &a);
                const int c = pe_column(kind, a, nf, cfg->pe_layout);
                col[hh][j] = c < 0 ? -1 : base + c;
            }
    };
    int col[2][8];
    for (int q = 0; q < B.kd16; ++q) { nat(q, 24 + 4 * v, col); put_step(B.sec_step[SEC_BD] + q, rows(L.p_bdw, 24 + 4 * v), col); }
    for (int q = 0; q < B.kc16; ++q) { nat(q, 8 * v, col); put_step(B.sec_step[SEC_BC] + q, rows(L.p_bcw, 8 * v), col); }
    for (int q = 0; q < KS16_PE_PTS; ++q) { pe(q, 10, 0, col); put_step(B.sec_step[SEC_L0] + q, rows(L.p_lw[0], MLP_PE_PTS), col); }
    for (int l = 1; l < 5; ++l)
        for (int q = 0; q < KS16_HID; ++q) { hid(q, 0, col); put_step(B.sec_step[SEC_L0 + l] + q, rows(L.p_lw[l], W), col); }
    for (int q = 0; q < KS16_PE_PTS; ++q) { pe(q, 10, 0, col); put_step(B.sec_step[SEC_L0 + 5] + q, rows(L.p_lw[5], W + MLP_PE_PTS), col); }
    for (int q = 0; q < KS16_HID; ++q) { hid(q, MLP_PE_PTS, col); put_step(B.sec_step[SEC_L0 + 5] + KS16_PE_PTS + q, rows(L.p_lw[5], W + MLP_PE_PTS), col); }
    for (int q = 0; q < KS16_HID; ++q) { hid(q, 0, col); put_step(B.sec_step[SEC_FT] + q, rows(L.p_fw, W), col); }
    {
        std::vector<int64_t> rb(128);
        for (int n = 0; n < 64; ++n) { rb[n] = L.p_vw + (int64_t)n * (W + MLP_PE_DIR); rb[64 + n] = L.p_vcw + (int64_t)n * (W + MLP_PE_DIR); }
        for (int q = 0; q < KS16_HID; ++q) { hid(q, 0, col); put_step(B.sec_step[SEC_VC] + q, rb, col); }
        for (int q = 0; q < KS16_PE_DIR; ++q) { pe(q, 4, W, col); put_step(B.sec_step[SEC_VC] + KS16_HID + q, rb, col); }
    }
    return 0;
}

__global__ void pack_bf16_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx, unsigned short* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t k = idx[i];
    unsigned short r = 0;
    if (k >= 0) {
        const float w = flat[k & 0x3fffffff];
        const __bf16 hi = (__bf16)w;
        const __bf16 val = (k >> 30) ? (__bf16)(w - (float)hi) : hi;
        r = __builtin_bit_cast(unsigned short, val);
    }
    out[i] = r;
}

// ------------------------------------------------------------------------------------------------ device helpers
struct Frag { bf16x8 hi, lo; };

__device__ __forceinline__ Frag split8(const float (&x)[8]) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        f.hi[j] = h;
        f.lo[j] = (__bf16)(x[j] - (float)h);
    }
    return f;
}

struct BGeom {
    int F, kd16, kc16, f_img, steps, feat_stride;
    unsigned stream_bytes, feat_bytes;
    int const_off_bytes;
    unsigned long long* diag;          // diagnostic builds only (UCNERF_MLP_DIAG): per-wave phase clocks of one tile
};

// A fragments of one row-tile pair of a step
struct AF { bf16x8 h0, l0, h1, l1; };

// block-wide weight pipeline state (all values wave-uniform)
struct Pipe {
    const char* __restrict__ gsrc;    // this lane's source byte within step 0: stream + wave*2048 + lane*16
    char* ring;                        // LDS ring base
    unsigned ring_lds;                 // ... as an LDS byte address
    const char* buf;                   // slot of the step being multiplied
    int wave;
    int gstep;                         // running step counter (never reset: ring slot = gstep & (NBUF-1))
    int next_src;                      // step (mod steps) of the next DMA to issue
    int steps;
};

// The copy is issued from inline asm on purpose: the compiler models a global_load_lds as a FLAT access that may
// touch both memories and from then on degrades every counted wait of the kernel to vmcnt(0) / lgkmcnt(0), which
// serialises the fragment prefetch below.  All hazards of the ring are handled explicitly in advance().
__device__ __forceinline__ void issue_dma(Pipe& P, int slot_step) {
    const char* src = P.gsrc + (size_t)P.next_src * STEP_BYTES;
    const unsigned dst = P.ring_lds + (slot_step & (NBUF - 1)) * STEP_BYTES + P.wave * (DMA_PER_STEP * 1024);
#pragma unroll
    for (int i = 0; i < DMA_PER_STEP; ++i)
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(src), "s"(dst), "n"(i * 1024) : "memory", "m0");   // the offset moves both addresses
    P.next_src = P.next_src + 1 == P.steps ? 0 : P.next_src + 1;
}

__device__ __forceinline__ AF read_pair(const char* buf, int lane, int pair) {
    const bf16x8* a = reinterpret_cast<const bf16x8*>(buf) + lane;
    AF f;
    f.h0 = a[(pair * 4 + 0) * 64]; f.l0 = a[(pair * 4 + 1) * 64];
    f.h1 = a[(pair * 4 + 2) * 64]; f.l1 = a[(pair * 4 + 3) * 64];
    return f;
}

// acc += A_hi*B_hi + A_hi*B_lo + A_lo*B_hi for two row-tiles, interleaved so that dependent MFMAs are 2 apart
__device__ __forceinline__ void mfma6(const AF& a, const Frag& b, f32x16& c0, f32x16& c1) {
    c0 = MFMA16(a.h0, b.hi, c0); c1 = MFMA16(a.h1, b.hi, c1);
    c0 = MFMA16(a.h0, b.lo, c0); c1 = MFMA16(a.h1, b.lo, c1);
    c0 = MFMA16(a.l0, b.hi, c0); c1 = MFMA16(a.l1, b.hi, c1);
}

// Waits until the next step has landed for the whole block, refills the slot just read and moves on to it.
// On entry every ds_read of the current slot has returned (lgkmcnt(0)), so after the barrier no wave still reads it.
// vmcnt: the DMAs younger than the awaited step are those of the two steps after it.
#ifndef UCNERF_BF16_EXP
#define UCNERF_BF16_EXP 0      // timing experiments only (results are wrong): 1 no DMA wait, 2 no barrier, 4 no DMA
#endif
__device__ __forceinline__ void advance(Pipe& P) {
#if !(UCNERF_BF16_EXP & 1)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * DMA_PER_STEP) : "memory");
#endif
#if !(UCNERF_BF16_EXP & 2)
    __builtin_amdgcn_s_barrier();
#endif
#if !(UCNERF_BF16_EXP & 4)
    issue_dma(P, P.gstep);                                  // step gstep + NBUF into the slot of step gstep
#endif
    ++P.gstep;
    P.buf = P.ring + (P.gstep & (NBUF - 1)) * STEP_BYTES;
}

// One k16-step for the four row-tiles.  `cur` = fragments of (this step, pair 0) on entry, of (next step, pair 0) on exit.
__device__ __forceinline__ void step16(Pipe& P, AF& cur, int lane, const Frag& b, f32x16 (&acc)[4]) {
    const AF n1 = read_pair(P.buf, lane, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(cur, b, acc[0], acc[1]);
    __builtin_amdgcn_sched_barrier(0);
    advance(P);
    cur = read_pair(P.buf, lane, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(n1, b, acc[2], acc[3]);
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void init_bias16(const float* cst, int sec, int h, f32x16 (&acc)[4]) {
    const f32x4* b = reinterpret_cast<const f32x4*>(cst + sec * 128 + h * 64);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b[nt * 4 + q];
            acc[nt][4 * q] = v.x; acc[nt][4 * q + 1] = v.y; acc[nt][4 * q + 2] = v.z; acc[nt][4 * q + 3] = v.w;
        }
}

// Four head outputs from a 128-wide activation set (accumulator layout): VALU, two outputs per v_pk_fma_f32.  The
// weight rows come from LDS in batches of 8 reads so that their latency is paid once per batch, not once per read.
__device__ __forceinline__ f32x4 head4_16(const float* hd, int h, const f32x16 (&x)[4]) {
    const f32x4* w = reinterpret_cast<const f32x4*>(hd) + h * 64;
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        f32x4 wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = w[b * 8 + i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xv = x[b >> 1][(b & 1) * 8 + i];
            const f32x2 xx = {xv, xv};
            s01 = __builtin_elementwise_fma(xx, (f32x2){wv[i].x, wv[i].y}, s01);
            s23 = __builtin_elementwise_fma(xx, (f32x2){wv[i].z, wv[i].w}, s23);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 s = {s01.x, s01.y, s23.x, s23.y};
    s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32);
    const f32x4 b = *reinterpret_cast<const f32x4*>(hd + 512);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    return s;
}

template <int NF, int KS>
__device__ __forceinline__ void encode16(const float (&x)[3], int h, float (&pe)[KS]) {
    constexpr int half = 3 * NF / 2;
#pragma unroll
    for (int q = 0; q < half; ++q) {
        const int a = 2 * q + h;
        const int fr = a / 3, c = a - 3 * fr;
        const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
        float s, co;
        sincos_pe(xc * (float)(1 << fr), &s, &co);
        pe[q] = s;
        pe[half + q] = co;
    }
    pe[2 * half] = h ? x[2] : x[0];
    pe[2 * half + 1] = h ? 0.f : x[1];
#pragma unroll
    for (int q = 2 * half + 2; q < KS; ++q) pe[q] = 0.f;
}

// hidden layer: 8 k16-steps on the fragments of a 128-wide activation set
__device__ __forceinline__ void gemm16_hidden(Pipe& P, AF& cur, int lane, const Frag (&x)[8], f32x16 (&acc)[4]) {
#pragma unroll
    for (int q = 0; q < KS16_HID; ++q) step16(P, cur, lane, x[q], acc);
}

// fragments of an accumulator-layout fp32 activation set (optionally with an element-wise map applied first)
template <class Map>
__device__ __forceinline__ void to_frags(const f32x16 (&v)[4], Frag (&x)[8], Map map) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = map(kt, 8 * s + j, v[kt][8 * s + j]);
            x[2 * kt + s] = split8(t);
        }
}

template <bool TILED, int NSRC>
__global__ void __launch_bounds__(64 * BW, 2) mlp_fwd_bf16_kernel(ucnerf_mlp_params p, BGeom g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // ONE shared object: [ring][constants][pe stash]
    char* ring = smem;
    float* cst = reinterpret_cast<float*>(smem + NBUF * STEP_BYTES);
    Frag* stash_all = reinterpret_cast<Frag*>(smem + NBUF * STEP_BYTES + ((CONST_FLOATS * 4 + 15) & ~15));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    constexpr int KD_S = (24 + 4 * NSRC + 15) / 16, KC_S = (8 * NSRC + 15) / 16;
    const int kd16 = NSRC ? KD_S : g.kd16, kc16 = NSRC ? KC_S : g.kc16;
    const int F = NSRC ? 24 + 12 * NSRC + 1 : g.F, f_img = NSRC ? 24 + 4 * NSRC : g.f_img;
    const char* __restrict__ ws = reinterpret_cast<const char*>(p.wstream);

    {   // constants (fp32 biases + head blocks) -> LDS, plain loads: they are waited for before any DMA is issued
        const float* csrc = reinterpret_cast<const float*>(ws + g.const_off_bytes);
        for (int i = threadIdx.x; i < CONST_FLOATS; i += 64 * BW) cst[i] = csrc[i];
    }
    __syncthreads();
    const float* hb = cst + N_SEC * 128;
    const float* ha = hb + 516;
    Frag* stash = stash_all + (size_t)wave * (KS16_PE_PTS * 64) + lane;        // step q at stash[q * 64]

    Pipe P;
    P.gsrc = ws + wave * (DMA_PER_STEP * 1024) + lane * 16;
    P.ring = ring; P.ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring; P.buf = ring; P.wave = wave; P.gstep = 0; P.next_src = 0; P.steps = g.steps;
#pragma unroll
    for (int i = 0; i < NBUF; ++i) issue_dma(P, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * DMA_PER_STEP) : "memory");     // step 0 has landed ...
    __builtin_amdgcn_s_barrier();                                                        // ... for every wave
    AF cur = read_pair(P.buf, lane, 0);

    const int tiles_per_round = gridDim.x * BW;
    const int n_rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;
#ifdef UCNERF_MLP_DIAG
#define DIAG_STAMP(K) { __builtin_amdgcn_sched_barrier(0); if (g.diag && lane == 0 && round == 5) g.diag[(size_t)(blockIdx.x * BW + wave) * 16 + (K)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define DIAG_STAMP(K)
#endif
    // Inputs of a tile are fetched one tile ahead (under the previous tile's head / blend arithmetic, when few registers are live), so their latency never
    // shows: operands of the depth-bias net (element e of step q = feature 16q + 8h + e; columns past a section's
    // width meet zero weights, so their index is only clamped into the row -- no branch), confidence, point, direction.
    float nfs[4][8], nconf, npx[3];
    auto sample_of = [&](int tile) { const int s_raw = tile * 32 + j; return s_raw < p.m ? s_raw : p.m - 1; };
    auto feat_base = [&](int s) { return TILED ? p.feats + ((size_t)(s >> 5) * F * 32 + (s & 31)) : p.feats + (size_t)s * g.feat_stride; };
    constexpr int fstride = TILED ? 32 : 1;
    auto fetch = [&](int tile) {
        const int s = sample_of(tile);
        const float* fb = feat_base(s);
        const float* fh = fb + 8 * h * fstride;            // one per-lane base, constant offsets from it
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 16 * q + e;                  // feature c + 8h
                nfs[q][e] = q >= kd16 ? 0.f : (NSRC && c + 8 < F) ? fh[c * fstride] : fb[(size_t)min(c + 8 * h, F - 1) * fstride];
            }
        nconf = fb[(size_t)(F - 1) * fstride];
        const float* prow = p.pts + (size_t)s * 3;
        npx[0] = prow[0]; npx[1] = prow[1]; npx[2] = prow[2];
    };
    fetch(blockIdx.x * BW + wave);

    for (int round = 0; round < n_rounds; ++round) {                          // block-uniform trip count: every wave joins every barrier
        const int tile = round * tiles_per_round + blockIdx.x * BW + wave;
        // (few scalars are carried through the trunk -- every VGPR there is spoken for: sample index, feature base
        //  and view direction are re-derived / loaded where they are needed)
        DIAG_STAMP(0)
        f32x16 bd[4], acc[4];
        Frag xin[8];
        float fsec[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) fsec[q][e] = nfs[q][e];
        const float px[3] = {npx[0], npx[1], npx[2]};

        // ---- point encoding -> fragments, kept in LDS for the skip connection
        Frag pef[KS16_PE_PTS];
        {
            float pe[KS_PE_PTS];
            encode16<10, KS_PE_PTS>(px, h, pe);
#pragma unroll
            for (int q = 0; q < KS16_PE_PTS; ++q) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = pe[8 * q + e];
                pef[q] = split8(t);
                stash[q * 64] = pef[q];
            }
        }

        DIAG_STAMP(1)
        // ---- depth-bias net
        init_bias16(cst, SEC_BD, h, bd);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < kd16) step16(P, cur, lane, split8(fsec[q]), bd);
        const float u = 1.f - nconf;
        DIAG_STAMP(2)

        // ---- layer 0
        init_bias16(cst, SEC_L0, h, acc);
#pragma unroll
        for (int q = 0; q < KS16_PE_PTS; ++q) step16(P, cur, lane, pef[q], acc);
        DIAG_STAMP(3)
        to_frags(acc, xin, [&](int kt, int r, float y) { return fmaxf(y * bd[kt][r], 0.f); });
        DIAG_STAMP(4)

        // ---- layers 1..4
#pragma unroll 1
        for (int l = 1; l < 5; ++l) {
            init_bias16(cst, SEC_L0 + l, h, acc);
            gemm16_hidden(P, cur, lane, xin, acc);
            to_frags(acc, xin, [&](int kt, int r, float y) { return fmaxf(y * bd[kt][r], 0.f); });
        }

        DIAG_STAMP(5)
        // ---- layer 5 on [pe | h]
        init_bias16(cst, SEC_L0 + 5, h, acc);
#pragma unroll
        for (int q = 0; q < KS16_PE_PTS; ++q) { const Frag f = stash[q * 64]; step16(P, cur, lane, f, acc); }
        gemm16_hidden(P, cur, lane, xin, acc);
        DIAG_STAMP(6)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = fmaxf(acc[nt][r] * bd[nt][r], 0.f);      // h5 (fp32) stays in acc

        // ---- operands of the confidence-bias net; base heads meanwhile
        const float* fb = feat_base(sample_of(tile));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = f_img + 16 * q + e;
                fsec[q][e] = q >= kc16 ? 0.f : (NSRC && c + 8 < F) ? fb[(8 * h + c) * fstride] : fb[(size_t)min(c + 8 * h, F - 1) * fstride];
            }
        const f32x4 base = head4_16(hb, h, acc);
        DIAG_STAMP(7)

        // ---- confidence-bias net -> bd; g = h5 * b_c -> fragments
        init_bias16(cst, SEC_BC, h, bd);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < kc16) step16(P, cur, lane, split8(fsec[q]), bd);
        DIAG_STAMP(8)
        to_frags(acc, xin, [&](int kt, int r, float y) { return y * bd[kt][r]; });
        DIAG_STAMP(9)

        // ---- feature_linear
        init_bias16(cst, SEC_FT, h, acc);
        gemm16_hidden(P, cur, lane, xin, acc);
        to_frags(acc, xin, [&](int, int, float y) { return y; });
        DIAG_STAMP(10)

        // ---- views_linears | view_confi_linears on [feature | dir encoding], relu
        init_bias16(cst, SEC_VC, h, acc);
        const int sd = sample_of(tile);
        const float* drow = p.dirs + (p.dirs_per_sample ? (size_t)sd : (size_t)(sd / p.S)) * 3;
        const float dv[3] = {drow[0], drow[1], drow[2]};   // arrives under the 8 hidden steps
        gemm16_hidden(P, cur, lane, xin, acc);
        {
            float pd[KS_PE_DIR];
            encode16<4, KS_PE_DIR>(dv, h, pd);
#pragma unroll
            for (int q = 0; q < KS16_PE_DIR; ++q) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = pd[8 * q + e];
                step16(P, cur, lane, split8(t), acc);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = fmaxf(acc[nt][r], 0.f);

        DIAG_STAMP(11)
        fetch(tile + tiles_per_round);                     // next tile's inputs (clamped past the end: harmless)
        // ---- adapt heads, uncertainty blend
        const f32x4 adapt = head4_16(ha, h, acc);
        DIAG_STAMP(12)
        const float omu = 1.f - u;
        const int s_raw = tile * 32 + j;
        f32x4 out;
        out.x = 1.f / (1.f + expf(-(base.x * omu + adapt.x * u)));
        out.y = 1.f / (1.f + expf(-(base.y * omu + adapt.y * u)));
        out.z = 1.f / (1.f + expf(-(base.z * omu + adapt.z * u)));
        out.w = fmaxf(adapt.w * omu + base.w * u, 0.f);
        if (h == 0 && s_raw < p.m) reinterpret_cast<f32x4*>(p.raw)[s_raw] = out;
        DIAG_STAMP(13)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
}

constexpr size_t bf16_smem_bytes() {
    return (size_t)NBUF * STEP_BYTES + ((CONST_FLOATS * 4 + 15) & ~15) + (size_t)BW * KS16_PE_PTS * 64 * sizeof(Frag);
}

int launch_mlp_fwd_bf16(const ucnerf_mlp_params* p, hipStream_t st) {
    UCNERF_REQUIRE(p, "mlp_fwd: null params");
    if (p->m == 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->pts && p->dirs && p->feats && p->wstream && p->raw, "mlp_fwd: null pointer");
    UCNERF_REQUIRE(!p->encoded && !p->pts_stride && !p->dirs_stride, "mlp_fwd (bf16x3): encoded / strided inputs are only available in f32 precision");
    UCNERF_REQUIRE(p->dirs_per_sample || p->S > 0, "mlp_fwd: S must be > 0 when dirs are per ray");
    UCNERF_REQUIRE(((uintptr_t)p->wstream & 15) == 0 && ((uintptr_t)p->raw & 15) == 0, "mlp_fwd: wstream/raw must be 16-byte aligned");
    Bf16Layout B;
    UCNERF_REQUIRE(bf16_layout(p->cfg.n_src, &B), "mlp_fwd: n_src %d outside 1..8", p->cfg.n_src);
    const int n_tiles = cdiv(p->m, 32);
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_fwd: no device");
    int blocks = cdiv(n_tiles, BW);
    const int cap = p->max_blocks > 0 ? p->max_blocks : 2 * cus;
    if (blocks > cap) blocks = cap;
    BGeom g;
    g.F = B.F; g.kd16 = B.kd16; g.kc16 = B.kc16; g.f_img = 24 + 4 * B.v; g.steps = B.steps;
    g.feat_stride = p->feat_stride ? p->feat_stride : B.F;
    g.stream_bytes = (unsigned)B.total_bytes; g.feat_bytes = 0; g.const_off_bytes = (int)B.const_off_bytes;
    g.diag = nullptr;
#ifdef UCNERF_MLP_DIAG
    { const char* e = getenv("UCNERF_MLP_DIAG_PTR"); g.diag = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    static bool attr_set = false;
    const size_t smem = bf16_smem_bytes();
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<true, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    dim3 grid(blocks), block(64 * BW);
    if (B.v == 6) {
        if (p->feats_tiled) hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, 6>), grid, block, smem, st, *p, g, n_tiles);
        else hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, 6>), grid, block, smem, st, *p, g, n_tiles);
    } else {
        if (p->feats_tiled) hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, 0>), grid, block, smem, st, *p, g, n_tiles);
        else hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, 0>), grid, block, smem, st, *p, g, n_tiles);
    }
    return check_launch("mlp_fwd_bf16");
}

int64_t bf16_index_count(const ucnerf_mlp_config* cfg) {
    Bf16Layout B;
    if (!bf16_layout(cfg->n_src, &B)) return -1;
    return (int64_t)B.steps * (STEP_BYTES / 2) + CONST_FLOATS;
}

int64_t bf16_stream_floats(const ucnerf_mlp_config* cfg) {
    Bf16Layout B;
    if (!bf16_layout(cfg->n_src, &B)) return -1;
    return B.total_bytes / 4;
}

__global__ void pack_f32_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int32_t k = idx[i]; out[i] = k >= 0 ? flat[k] : 0.f; }
}

int launch_pack_bf16(const ucnerf_mlp_config* cfg, const float* flat, const int32_t* idx, float* out, hipStream_t st) {
    Bf16Layout B;
    UCNERF_REQUIRE(bf16_layout(cfg->n_src, &B), "mlp_pack: n_src %d outside 1..8", cfg->n_src);
    const int64_t n16 = (int64_t)B.steps * (STEP_BYTES / 2);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(cdiv(n16, 256)), dim3(256), 0, st, flat, idx, reinterpret_cast<unsigned short*>(out), n16);
    hipLaunchKernelGGL(pack_f32_kernel, dim3(cdiv(CONST_FLOATS, 256)), dim3(256), 0, st, flat, idx + n16,
                       reinterpret_cast<float*>(reinterpret_cast<char*>(out) + B.const_off_bytes), CONST_FLOATS);
    return check_launch("mlp_pack (bf16x3)");
}

}  // namespace ucnerf
