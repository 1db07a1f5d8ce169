// K5+K6 on the bf16 matrix cores with fp32-grade accuracy: every product a*w is evaluated as
//     a_hi*w_hi + a_hi*w_lo + a_lo*w_hi      (x_hi = bf16(x), x_lo = bf16(x - x_hi); fp32 accumulate)
// with v_mfma_f32_32x32x16_bf16.  The dropped terms are O(2^-16) relative: rendered colours / depths stay within
// 1e-5 of the fp64 result (tests pin 1e-4), while a plain bf16 evaluation is off by 3e-3.
//
// Why: on gfx950 the fp32 MFMA executes on the vector ALU (VALU fillers cost their full issue time, a partner
// wave's VALU starves while fp32 MFMAs stream -- scripts/micro/mfma_fillers.hip), so the fp32 kernel's ceiling is
// the 157 TFLOP/s vector rate minus all element-wise work.  bf16 MFMAs run on the separate matrix cores at 16x
// the rate: three of them per product are still > 5x faster, and the element-wise work can run underneath them.
//
// Structure (same transposed formulation as mlp.hip: lane = sample, activations live in accumulator registers):
//   * a k16-step = 16 input features: lane-half hh supplies 8 of them as one bf16x8 fragment.  For hidden layers
//     the fragment of step (kt, s) is accumulator registers 8s..8s+7 of row-tile kt -- the accumulator layout is
//     again directly the next layer's operand, only re-split into (hi, lo) by the epilogue;
//   * the unit of work is a HALF-STEP: one k16-step for one PAIR of 32-row output tiles = 6 MFMAs on a 4-KB block
//     of weights [hi0 | lo0 | hi1 | lo1][lane][8 bf16].  The 128-wide layers run pair-split: all k-steps for row
//     tiles 0,1 (phase A), then all k-steps for row tiles 2,3 (phase B).  Row tiles 0,1 are therefore final while
//     phase B still multiplies, and their epilogue (bias-net product, relu, hi/lo split: ~45 % of a layer's MFMA
//     time on the VALU) is issued BETWEEN phase B's MFMAs; the epilogue of tiles 2,3 runs under the next layer's
//     phase A, whose first four k-steps only need the fragments of tiles 0,1.  Bias loads, head dot products and
//     direction encodings are slotted into other phases the same way (sched_group_barrier pins the interleave);
//   * weights are consumed too fast to stream per wave from L2, so the waves of a block share them: the packed
//     stream (half-steps in consumption order) is copied into a 4-slot LDS ring (8 KB = two half-steps per slot) by
//     global_load_lds (two 1-KB pieces per wave per slot, issued 3.5 slots ahead, counted vmcnt + one raw s_barrier
//     per slot), and every wave reads its A fragments with ds_read_b128, one half-step ahead of the MFMAs;
//   * a block is 8 waves = two per SIMD, one block per CU: the whole CU shares one ring, so the stream crosses L2 -> LDS once per
//     eight tiles.  (Round 1 ran two 4-wave blocks per CU so that the two waves of a SIMD never met at a barrier; measured in
//     round 2: the chip is power-limited under this kernel -- one wave per SIMD renders 93 % of what two do -- so what pays is
//     less data movement per tile, not more overlap.)
//   * FUSED instantiation: the tile loop opens with the tile's feature gather -- the operands of the
//     two bias nets come straight from the channel-last sources instead of a feature buffer written by feat_gather_cl_kernel.
#include "common.h"
#include "mlp_layout.h"
#include "sincos_cw.h"
#include "gather_cl_device.h"
#include "raygen_device.h"
#include "p24.h"
#include "composite_device.h"
#include "sample_pdf_device.h"

// the LDS-DMA asm below names m0 as a clobber on purpose (it loads the LDS base into it)
#pragma clang diagnostic ignored "-Winline-asm"

#include <cstdlib>
#include <vector>

// This file is compiled twice (uc_nerf_amd/build.py): UCNERF_BF16_BUILD_TERMS = 3 gives the split-bf16 launcher
// (launch_mlp_fwd_bf16x3) plus the host-side packing shared by both precisions, = 1 the plain-bf16 launcher only.
#ifndef UCNERF_BF16_BUILD_TERMS
#define UCNERF_BF16_BUILD_TERMS 3
#endif
// A third build (-DUCNERF_BF16_BUILD_TAIL=1, TERMS = 3) holds nothing but the TAIL instantiations of the gather-fused kernel and their launcher
// (launch_fused_tail): the launch that also composites and re-samples the rays of a small pass -- compiled beside the main object, not after it.
#ifndef UCNERF_BF16_BUILD_TAIL
#define UCNERF_BF16_BUILD_TAIL 0
#endif

namespace ucnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// ucnerf_mlp_config.operand (ABI v6): every build of this file exists twice -- with bf16 terms (UCNERF_OPERAND_FP16 = 0: the entry points below) and with
// fp16 terms (UCNERF_OPERAND_FP16 = 1: 11-bit hi and lo terms, v_mfma_f32_32x32x16_f16; the same entry points under the suffix _h16, which the
// bf16 build's entry points forward to when cfg.operand == 1).  fp16 has no range guard: values beyond 65 504 overflow.
#ifndef UCNERF_OPERAND_FP16
#define UCNERF_OPERAND_FP16 0
#endif
#if UCNERF_OPERAND_FP16
#define build_pack_index_bf16 build_pack_index_bf16_h16
#define launch_fused_tail launch_fused_tail_h16
#define build_flags_mlp_bf16x3 build_flags_mlp_bf16x3_h16
#define build_flags_mlp_bf16_plain build_flags_mlp_bf16_plain_h16
#define launch_mlp_fwd_bf16x3 launch_mlp_fwd_bf16x3_h16
#define launch_mlp_fwd_bf16x3_save launch_mlp_fwd_bf16x3_save_h16
#define launch_mlp_fwd_bf16x3_gather launch_mlp_fwd_bf16x3_gather_h16
#define launch_mlp_fwd_bf16_plain launch_mlp_fwd_bf16_plain_h16
#define bf16_index_count bf16_index_count_h16
#define bf16_stream_floats bf16_stream_floats_h16
#define launch_pack_bf16 launch_pack_bf16_h16
#define launch_pack_bf16_tab launch_pack_bf16_tab_h16
#define mlp_fwd_bf16_kernel mlp_fwd_h16_kernel                      // (the kernels too: a template's instantiations are weak symbols -- the linker would keep ONE of two
#define pack_all_flat_kernel pack_all_flat_h16_kernel                //  equally named ones and both operand kinds would run the same code)
#define pack_all_tab_kernel pack_all_tab_h16_kernel
#define UCNERF_OPERAND_DISPATCH(cfg_operand, call)
#else
// (the fp16 build's entry points; `call` names one of them with this entry point's own arguments)
#define UCNERF_OPERAND_DISPATCH(cfg_operand, call) do { if ((cfg_operand) == 1) return call; } while (0)
#endif
#if UCNERF_OPERAND_FP16
typedef _Float16 op16;
#else
typedef __bf16 op16;
#endif
typedef op16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef UCNERF_BF16_HINT_V
#define UCNERF_BF16_HINT_V 5   // VALU instructions the scheduler may place after each MFMA of a half-step (7 would fill an MFMA's 32 cycles; measured
                               // over 1..9: 3..6 read 0.5 % faster than 7, 9 is 1.5 % slower -- profiles/r02_logs/r02_hint_v_sweep.log)
#endif
#ifndef UCNERF_BF16_PRIO_VALU
#define UCNERF_BF16_PRIO_VALU 0     // wave priority during the phases without MFMAs (point encoding, tail)
#endif
#ifndef UCNERF_BF16_PRIO_GEMM
#define UCNERF_BF16_PRIO_GEMM 0     // ... and during the GEMM phases
#endif
// scalar fp32 ops the SLP vectoriser cannot re-pack (one empty asm per result keeps every op its own instruction)
__device__ __forceinline__ float sc_mul(float a, float b) { float r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float sc_sub(float a, float b) { float r = a - b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float sc_fma(float a, float b, float c) { float r = __builtin_fmaf(a, b, c); asm volatile("" : "+v"(r)); return r; }
#if UCNERF_OPERAND_FP16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif
#define SB0 __builtin_amdgcn_sched_barrier(0)

#ifndef UCNERF_BF16_IDLE_SKIP
#define UCNERF_BF16_IDLE_SKIP 1          // a wave whose tile lies past the end only turns the weight ring (0: it computes a clamped tile and discards it)
#endif
#ifndef UCNERF_BF16_BW
#define UCNERF_BF16_BW 8       // 8: one 512-thread block per CU, all eight waves share ONE weight ring -- half the LDS-DMA traffic and DMA issues
                               //    per tile of two 4-wave blocks; measured -3.3 % on the launch (profiles/r02_mlp_bf16_experiments.md)
#endif
constexpr int BW = UCNERF_BF16_BW;    // waves per block sharing one weight ring: 4 (one per SIMD, two blocks per CU) or 8 (two per SIMD, one block per CU)
constexpr int SLOT_BYTES = 8192;      // two half-steps: [2][hi0, lo0, hi1, lo1][64 lanes][16 B]
constexpr int HALF_BYTES = 4096;
#ifndef UCNERF_BF16_NBUF
#define UCNERF_BF16_NBUF 4
#endif
constexpr int NBUF = UCNERF_BF16_NBUF;   // LDS ring slots (power of two); the slot of ring position g is refilled with position g + NBUF
constexpr int fused_ring_slots(bool fused, int n_src) { return fused && n_src > 6 && NBUF > 2 ? 2 : NBUF; }      // (see the kernel)
constexpr int DMA_PER_SLOT = SLOT_BYTES / 1024 / BW;      // 1-KB global_load_lds pieces per wave per slot
constexpr int KS16_PE_PTS = 4, KS16_PE_DIR = 2, KS16_HID = 8;

struct Bf16Layout {
    int v, F, kd16, kc16, slots;      // slots = k16-steps per tile (two half-steps each)
    int64_t const_off_bytes, total_bytes;
};

static bool bf16_layout(int v, Bf16Layout* B) {
    if (v < 1 || v > 8) return false;
    B->v = v; B->F = 24 + 12 * v + 1;
    B->kd16 = (24 + 4 * v + 15) / 16; B->kc16 = (8 * v + 15) / 16;
    B->slots = B->kd16 + KS16_PE_PTS + 4 * KS16_HID + (KS16_PE_PTS + KS16_HID) + B->kc16 + KS16_HID + (KS16_HID + KS16_PE_DIR);
    B->const_off_bytes = (int64_t)B->slots * SLOT_BYTES;
    B->total_bytes = B->const_off_bytes + (int64_t)CONST_FLOATS * 4;
    return true;
}

// feature held by element j of lane-half hh in hidden k16-step q = (kt, s): accumulator register 8s + j of row-tile kt
__host__ __device__ inline int hid_feature16(int kt, int s, int j, int hh) { return 32 * kt + (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * hh; }

#if UCNERF_BF16_BUILD_TERMS == 3 && !UCNERF_BF16_BUILD_TAIL
// ------------------------------------------------------------------------------------------------ host: pack index
// idx16[e] for every bf16 element e of the stream: flat parameter index | (part << 30) (part 0 = hi, 1 = lo), -1 = zero.
// Half-steps appear in the order the kernel consumes them (see the schedule in mlp_fwd_bf16_kernel):
//   bd: step-major (q: pair 0, pair 1) | L0: pair-split | L1..L4: pair-split | L5: pair-split over [h 0..3 | pe 0..3 | h 4..7]
//   bc: step-major | ft: pair-split | vc: pair-split over [h 0..7 | dir 0..1]
int build_pack_index_bf16(const ucnerf_mlp_config* cfg, int32_t* idx) {
    Bf16Layout B;
    MlpLayout L;
    if (!bf16_layout(cfg->n_src, &B) || !mlp_layout(cfg->n_src, &L)) return -1;
    const int v = B.v, W = MLP_W;
    const int64_t n16 = (int64_t)B.slots * (SLOT_BYTES / 2);
    for (int64_t i = 0; i < n16 + CONST_FLOATS; ++i) idx[i] = -1;
    int64_t hidx = 0;
    auto put_half = [&](const std::vector<int64_t>& row_base, const int (&col)[2][8], int pair) {
        for (int t = 0; t < 2; ++t)
            for (int part = 0; part < 2; ++part)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int c = col[lane >> 5][j];
                        const int64_t e = (((hidx * 4 + t * 2 + part) * 64) + lane) * 8 + j;
                        idx[e] = c < 0 ? -1 : (int32_t)((row_base[32 * (2 * pair + t) + (lane & 31)] + c) | ((int64_t)part << 30));
                    }
        ++hidx;
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
    // precision 3 (the gather runs inside the kernel, fused_operands below): the two bias nets take their operands in the order the
    // lane halves produce them -- bd: [stage 1 | stage 2], [stage 3 ch 0..3, view 0 | ch 4..7, view 1], then two views per half and
    // step (even views in half 0, odd views in half 1); bc: the image features of view 2q + hh in step q
    const bool fused = cfg->precision == 3;
    auto bd_fused = [&](int q, int (&col)[2][8]) {
        for (int hh = 0; hh < 2; ++hh)
            for (int j = 0; j < 8; ++j) {
                int c = -1;
                if (q == 0) c = 8 * hh + j;
                else if (q == 1 && j < 4) c = 16 + 4 * hh + j;
                else {
                    const int pair = q == 1 ? 0 : 1 + 2 * (q - 2) + (j >> 2), view = 2 * pair + hh;
                    if (view < v) c = 24 + 4 * view + (j & 3);
                }
                col[hh][j] = c;
            }
    };
    auto bc_fused = [&](int q, int (&col)[2][8]) {
        for (int hh = 0; hh < 2; ++hh)
            for (int j = 0; j < 8; ++j) col[hh][j] = 2 * q + hh < v ? 8 * (2 * q + hh) + j : -1;
    };
    int col[2][8];
    {   // bd, step-major
        const auto rb = rows(L.p_bdw, 24 + 4 * v);
        for (int q = 0; q < B.kd16; ++q) { if (fused) bd_fused(q, col); else nat(q, 24 + 4 * v, col); put_half(rb, col, 0); put_half(rb, col, 1); }
    }
    {   // L0
        const auto rb = rows(L.p_lw[0], MLP_PE_PTS);
        for (int p = 0; p < 2; ++p) for (int q = 0; q < KS16_PE_PTS; ++q) { pe(q, 10, 0, col); put_half(rb, col, p); }
    }
    for (int l = 1; l < 5; ++l) {
        const auto rb = rows(L.p_lw[l], W);
        for (int p = 0; p < 2; ++p) for (int q = 0; q < KS16_HID; ++q) { hid(q, 0, col); put_half(rb, col, p); }
    }
    {   // L5 on [pe | h]: k order h 0..3, pe 0..3, h 4..7
        const auto rb = rows(L.p_lw[5], W + MLP_PE_PTS);
        for (int p = 0; p < 2; ++p) {
            for (int q = 0; q < 4; ++q) { hid(q, MLP_PE_PTS, col); put_half(rb, col, p); }
            for (int q = 0; q < KS16_PE_PTS; ++q) { pe(q, 10, 0, col); put_half(rb, col, p); }
            for (int q = 4; q < 8; ++q) { hid(q, MLP_PE_PTS, col); put_half(rb, col, p); }
        }
    }
    {   // bc, step-major
        const auto rb = rows(L.p_bcw, 8 * v);
        for (int q = 0; q < B.kc16; ++q) { if (fused) bc_fused(q, col); else nat(q, 8 * v, col); put_half(rb, col, 0); put_half(rb, col, 1); }
    }
    {   // feature_linear
        const auto rb = rows(L.p_fw, W);
        for (int p = 0; p < 2; ++p) for (int q = 0; q < KS16_HID; ++q) { hid(q, 0, col); put_half(rb, col, p); }
    }
    {   // views_linears | view_confi_linears on [feature | dir encoding]: k order h 0..7, dir 0..1
        std::vector<int64_t> rb(128);
        for (int n = 0; n < 64; ++n) { rb[n] = L.p_vw + (int64_t)n * (W + MLP_PE_DIR); rb[64 + n] = L.p_vcw + (int64_t)n * (W + MLP_PE_DIR); }
        for (int p = 0; p < 2; ++p) {
            for (int q = 0; q < KS16_HID; ++q) { hid(q, 0, col); put_half(rb, col, p); }
            for (int q = 0; q < KS16_PE_DIR; ++q) { pe(q, 4, W, col); put_half(rb, col, p); }
        }
    }
    return hidx == 2 * (int64_t)B.slots ? 0 : -1;
}


#endif   // UCNERF_BF16_BUILD_TERMS == 3 (host-side packing)

// ------------------------------------------------------------------------------------------------ device helpers
struct Frag { bf16x8 hi, lo; };

// (hi, lo) split of eight activations: hi = bf16_rne(x), lo = bf16_rne(x - hi) (the difference is exact before its rounding): one v_cvt_pk_bf16_f32
// per pair for each, a shift and a mask to read the pair's hi back as floats, one packed subtract -- 5 VALU per pair.  Worst case |lo| <= 2^-8 |x|,
// the part of x neither term holds <= 2^-16 |x|, the dropped lo*lo term of a product 2^-16 relative (the weights are split the same way on packing).
// (Rounds 1-4 took hi by TRUNCATION -- v_perm + two v_and, also 5 VALU: |lo| < 2^-7 |x|, the activations' share of all that one bit worse.  Round 5's differential
//  fuzzer, tests/fuzz_render.py, put the renders of that split at up to 2.5e-4 from the oracle on networks with an unscaled density head --
//  4 of 43 such cases above the 1e-4 bar; with the rounded hi the median error halves (2.9e-5 -> 1.9e-5, 90th percentile 9.3e-5 -> 4.7e-5)
//  for +0.2 % of the headline kernel's time, profiles/r05_experiments.md.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef op16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Frag split8(const float (&x)[8]) {
    u32x4 hi;
    Frag f;
#if UCNERF_OPERAND_FP16 && !UCNERF_BF16_BUILD_TAIL
    float minus_one = -1.0f;
    asm volatile("" : "+s"(minus_one));
#endif
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
#if UCNERF_OPERAND_FP16
        // fp16 terms: BOTH conversions toward zero (v_cvt_pkrtz_f16_f32) -- a value beyond fp16's range then becomes 65 504, never an infinity: an
        // activation past 131 008 is clamped (wrong, but finite: float32 itself resolves no more than 1e-2 of it), nothing ever turns into a NaN.
        // hi + lo holds 20 .. 21 bits (a rounded pair would hold 22).
        const auto hp = __builtin_amdgcn_cvt_pkrtz(x[j], x[j + 1]);
        const unsigned packed = __builtin_bit_cast(unsigned, hp);                 // [hi(x[j+1]) | hi(x[j])]
        hi[j >> 1] = packed;
        // x - (float)hi as ONE mixed-precision fma per value: v_fma_mix_f32 reads the fp16 half in place (no conversion back, no packed subtract: four
        // vector instructions per pair).  The compiler selects it only for a multiplier it cannot fold -- hence the opaque -1 in a scalar register; it
        // must be the COMPILER's instruction, not inline asm: the operands come out of MFMA accumulators and only the compiler's hazard recognizer
        // inserts the wait states such a read needs (an asm version was wrong on a few rays in 10^4, profiles/r05_experiments.md section 13).
        // (The tail-route build forms the same value -- x - hi is exact either way -- by converting the half back: its kernels have no scalar register
        //  to spare for the -1 and spilled three vector registers with it.)
#if UCNERF_BF16_BUILD_TAIL
        const float l0 = x[j] - (float)hp[0], l1 = x[j + 1] - (float)hp[1];
#else
        const float l0 = __builtin_fmaf((float)hp[0], minus_one, x[j]), l1 = __builtin_fmaf((float)hp[1], minus_one, x[j + 1]);
#endif
        const auto lp = __builtin_amdgcn_cvt_pkrtz(l0, l1);
        f.lo[j] = (op16)lp[0];
        f.lo[j + 1] = (op16)lp[1];
#else
        const bf16x2 hp = __builtin_convertvector((f32x2){x[j], x[j + 1]}, bf16x2);      // ONE v_cvt_pk_bf16_f32 (two scalar conversions were not merged)
        const unsigned packed = __builtin_bit_cast(unsigned, hp);                 // [hi(x[j+1]) | hi(x[j])]
        hi[j >> 1] = packed;
        const f32x2 h = {__builtin_bit_cast(float, packed << 16), __builtin_bit_cast(float, packed & 0xffff0000u)};
        const f32x2 l = (f32x2){x[j], x[j + 1]} - h;
        f.lo[j] = (op16)l.x;
        f.lo[j + 1] = (op16)l.y;
#endif
    }
    f.hi = __builtin_bit_cast(bf16x8, hi);
    return f;
}

// (The constants never change, so the compiler would hoist these LDS loads out of the tile loop and then spill what it
//  hoisted: the lane offset is laundered through an empty asm to keep every load where it is written.)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// the same for a wave-uniform value (kept in a scalar register); `on` = 0: identity (inside a GEMM phase's fill the asm form does not select:
// "illegal VGPR to SGPR copy")
__device__ __forceinline__ int sopaque(int v, bool on = true) { if (on) asm volatile("" : "+s"(v)); return v; }
// Results that are only needed much later would be sunk towards their use -- out of the MFMA shadow they were written
// for, with their operands kept live meanwhile.  An empty volatile asm on the result pins the arithmetic in place.
template <class T> __device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }

// Training forward (SAVE): the activation sets the backward reads (MlpSaved, mlp_layout.h), row-major [m,128].  A lane owns 64 of its
// sample's 128 values per set: register group (nt, q = reg / 4) of lane-half h sits at floats 32 nt + 8 q + 4 h of the row.
// P24: the sets in the 24-bit tiled format of p24.h instead (`row` then = this lane's piece of column group h)
template <bool P24>
__device__ __forceinline__ void save8(float* row, int nt, int s, const float (&t)[8]) {       // fragment s of row-tile nt
    // (plain stores: the eight 16-byte pieces of a 128-byte line leave this wave hundreds of cycles apart -- between MFMA groups --
    //  and have to meet in L2; as streaming stores each piece went to memory on its own)
    if (P24) {
        char* r24 = reinterpret_cast<char*>(row) + P24_GROUP_BYTES * (8 * nt + 4 * s);      // groups 8 nt + 2 q + h, q = 2 s and 2 s + 1 (`row` carries h)
        p24_store(r24, p24_pack4(t[0], t[1], t[2], t[3]));
        p24_store(r24 + 2 * P24_GROUP_BYTES, p24_pack4(t[4], t[5], t[6], t[7]));
    } else {
        *reinterpret_cast<f32x4*>(row + 32 * nt + 16 * s) = (f32x4){t[0], t[1], t[2], t[3]};
        *reinterpret_cast<f32x4*>(row + 32 * nt + 16 * s + 8) = (f32x4){t[4], t[5], t[6], t[7]};
    }
}
template <bool RELU, bool P24>
__device__ __forceinline__ void save_tile(float* row, int nt, const f32x16& x) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (P24) p24_store(reinterpret_cast<char*>(row) + P24_GROUP_BYTES * (8 * nt + 2 * q), p24_pack4(v.x, v.y, v.z, v.w));
        else *reinterpret_cast<f32x4*>(row + 32 * nt + 8 * q) = v;
    }
}

// fragment s (0/1) of an accumulator tile: MODE 0 plain, 1 times m, 2 relu(times m)   (two values per v_pk_mul_f32);
// SV: the eight fp32 values also go to `srow` (this lane's row of an activation set, NULL past the last sample) as row-tile nt
template <int MODE, int SV = 0>       // SV: the kernel's SAVE (0 none, 1 fp32 sets, 2 24-bit sets)
__device__ __forceinline__ Frag frag_of(const f32x16& a, const f32x16& m, int s, float* srow = nullptr, int nt = 0) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        f32x2 v = {a[8 * s + j], a[8 * s + j + 1]};
        if (MODE >= 1) v = v * (f32x2){m[8 * s + j], m[8 * s + j + 1]};
        t[j] = MODE == 2 ? fmaxf(v.x, 0.f) : v.x;
        t[j + 1] = MODE == 2 ? fmaxf(v.y, 0.f) : v.y;
    }
    if (SV && srow) save8<SV == 2>(srow, nt, s, t);
    Frag f = split8(t);
    pin(f.hi); pin(f.lo);
    return f;
}

struct BGeom {
    int F, kd16, kc16, f_img, slots, feat_stride;
    int const_off_bytes;
};

// A fragments of one half-step (row-tile pair)
struct AF { bf16x8 h0, l0, h1, l1; };

// block-wide weight pipeline state (all values wave-uniform)
struct Pipe {
    const char* __restrict__ gsrc;    // this lane's source byte within slot 0: stream + wave*2048 + lane*16
    char* ring;                        // LDS ring base
    unsigned ring_lds;                 // ... as an LDS byte address
    const char* buf;                   // slot being multiplied
    int wave;
    int gpos;                          // running ring position (never reset: ring slot = gpos & (NBUF-1))
    int next_src;                      // stream slot (mod slots) of the next DMA to issue
    int slots;
};

// The copy is issued from inline asm on purpose: the compiler models a global_load_lds as a FLAT access that may
// touch both memories and from then on degrades every counted wait of the kernel to vmcnt(0) / lgkmcnt(0), which
// serialises the fragment prefetch below.  All hazards of the ring are handled explicitly in advance().
template <int NB = NBUF>
__device__ __forceinline__ void issue_dma(Pipe& P, int pos) {
    const char* src = P.gsrc + (size_t)P.next_src * SLOT_BYTES;
    const unsigned dst = P.ring_lds + (pos & (NB - 1)) * SLOT_BYTES + P.wave * (DMA_PER_SLOT * 1024);
#pragma unroll
    for (int i = 0; i < DMA_PER_SLOT; ++i)
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(src), "s"(dst), "n"(i * 1024) : "memory", "m0");   // the offset moves both addresses
    P.next_src = P.next_src + 1 == P.slots ? 0 : P.next_src + 1;
}

__device__ __forceinline__ AF read_half(const char* buf, int lane, int half) {
    const bf16x8* a = reinterpret_cast<const bf16x8*>(buf + half * HALF_BYTES) + lane;
    AF f;
    f.h0 = a[0]; f.l0 = a[64]; f.h1 = a[128]; f.l1 = a[192];
    return f;
}

// Waits until the next slot has landed for the whole block, refills the slot just read and moves on to it.
// The lgkmcnt(0) retires every ds_read of the current slot, so after the barrier no wave still reads it.
// vmcnt: the DMAs younger than the awaited slot are those of the NBUF - 2 slots after it.
template <int NB = NBUF>
__device__ __forceinline__ void advance(Pipe& P) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NB - 2) * DMA_PER_SLOT) : "memory");
    __builtin_amdgcn_s_barrier();
    issue_dma<NB>(P, P.gpos);                               // position gpos + NB into the slot of position gpos
    ++P.gpos;
    P.buf = P.ring + (P.gpos & (NB - 1)) * SLOT_BYTES;
}

// scheduling hint for a half-step region: one MFMA, then up to `V` VALU, six times (LDS reads stay with their users)
template <int V>
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
    }
}

// One half-step: (c0, c1) += A_hi*B_hi + A_hi*B_lo + A_lo*B_hi for one row-tile pair, with `fill()` -- element-wise work
// that does not depend on these MFMAs -- issued between them.  ODD = second half-step of its ring slot (the next
// fragments then come from the next slot).  `cur` holds this half-step's A fragments on entry, the next one's on exit.
template <int TERMS, int NB = NBUF, class F>
__device__ __forceinline__ void half_step(const int ODD, Pipe& P, AF& cur, int lane, const Frag& b, f32x16& c0, f32x16& c1, F&& fill) {
    if (ODD) advance<NB>(P);                                   // (a constant once the caller's loop is unrolled)
    AF nxt;
    if (TERMS == 3) nxt = read_half(P.buf, lane, ODD ? 0 : 1);
    else {                                                 // plain bf16: the lo halves are never read
        const bf16x8* a = reinterpret_cast<const bf16x8*>(P.buf + (ODD ? 0 : 1) * HALF_BYTES) + lane;
        nxt.h0 = a[0]; nxt.h1 = a[128]; nxt.l0 = nxt.h0; nxt.l1 = nxt.h1;
    }
    SB0;
    c0 = MFMA16(cur.h0, b.hi, c0);
    if (TERMS == 3) { c0 = MFMA16(cur.h0, b.lo, c0); c0 = MFMA16(cur.l0, b.hi, c0); }   // (one accumulation chain runs at full rate)
    c1 = MFMA16(cur.h1, b.hi, c1);
    if (TERMS == 3) { c1 = MFMA16(cur.h1, b.lo, c1); c1 = MFMA16(cur.l1, b.hi, c1); }
    fill();
    if (TERMS == 3) interleave_hint<UCNERF_BF16_HINT_V>();
    SB0;
    cur = nxt;
}
#define HS(ODD, B, C0, C1, ...) half_step<TERMS, NB>((ODD), P, cur, lane, (B), (C0), (C1), [&]() { __VA_ARGS__; })      // (NB: the kernel's ring depth)

// bias block of section `sec` -> accumulators of one row-tile pair
__device__ __forceinline__ void init_bias_pair(const float* cst, int sec, int h, int pair, f32x16 (&acc)[4]) {
    const f32x4* b = reinterpret_cast<const f32x4*>(cst + sec * 128 + opaque(h * 64)) + pair * 8;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b[t * 4 + q];
            acc[2 * pair + t][4 * q] = v.x; acc[2 * pair + t][4 * q + 1] = v.y; acc[2 * pair + t][4 * q + 2] = v.z; acc[2 * pair + t][4 * q + 3] = v.w;
        }
}

// head partial sums over registers [r0, r0 + n) of row-tile nt (two outputs per v_pk_fma_f32); weights [h][nt][r][4] in LDS
struct HeadAcc { f32x2 s01, s23; };
template <class Map>
__device__ __forceinline__ void head_part(HeadAcc& a, const float* hd, int h, const f32x16& x, int nt, int r0, int n, Map map) {
    const f32x4* w = reinterpret_cast<const f32x4*>(hd) + opaque(h * 64) + nt * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i >= r0 && i < r0 + n) {
            const f32x4 wv = w[i];
            const float xv = map(x[i]);
            const f32x2 xx = {xv, xv};
            a.s01 = __builtin_elementwise_fma(xx, (f32x2){wv.x, wv.y}, a.s01);
            a.s23 = __builtin_elementwise_fma(xx, (f32x2){wv.z, wv.w}, a.s23);
        }
    pin(a.s01); pin(a.s23);
}
__device__ __forceinline__ f32x4 head_finish(const HeadAcc& a, const float* hd) {
    // x + x[lane ^ 32] with v_permlane32_swap: swapping the upper half of one copy with the lower half of another
    // leaves both addends of every lane in the two results (no lane id, no LDS)
    // (inline asm: the builtin's second result is dropped by the compiler when both operands are one value; the
    //  s_nop covers the VALU-write -> permlane-read hazard the compiler would otherwise pad itself)
    auto fold = [](float v) {
        float a = v, b = v;
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        return a + b;
    };
    f32x4 s = {fold(a.s01.x), fold(a.s01.y), fold(a.s23.x), fold(a.s23.y)};
    const f32x4 b = *reinterpret_cast<const f32x4*>(hd + 512);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    return s;
}

// sin / cos of encoding argument q of this lane-half: a = 2q + h, frequency a / 3, coordinate a % 3.  Both candidates
// have compile-time (frequency, coordinate) and are selected by h (written with a / 3 and a % 3 of the lane-dependent a,
// the compiler keeps ~40 per-lane multipliers and selectors alive across the whole tile loop).  t = coordinates in
// revolutions (to_revolutions), so one pair costs 3 selects + mul, fract, fma, sin, cos.
__device__ __forceinline__ void encode_sincos(const Rev2 (&t)[3], int h, int q, float* s, float* c) {
    const int a0 = 2 * q, a1 = 2 * q + 1;
    const int c0 = a0 % 3, c1 = a1 % 3;
    // (values first, select second: a select between array ELEMENTS becomes an indexed load from a stack copy)
    const float h0 = c0 == 0 ? t[0].hi : c0 == 1 ? t[1].hi : t[2].hi, l0 = c0 == 0 ? t[0].lo : c0 == 1 ? t[1].lo : t[2].lo;
    const float h1 = c1 == 0 ? t[0].hi : c1 == 1 ? t[1].hi : t[2].hi, l1 = c1 == 0 ? t[0].lo : c1 == 1 ? t[1].lo : t[2].lo;
    const float hi = h ? h1 : h0, lo = h ? l1 : l0;
    const float scale = h ? (float)(1 << (a1 / 3)) : (float)(1 << (a0 / 3));
    sincos_rev(hi, lo, scale, s, c);
}

template <int NF, int KS>
__device__ __forceinline__ void encode16(const float (&x)[3], int h, float (&pe)[KS]) {
    constexpr int half = 3 * NF / 2;
    const Rev2 t[3] = {to_revolutions(x[0]), to_revolutions(x[1]), to_revolutions(x[2])};
#pragma unroll
    for (int q = 0; q < half; ++q) encode_sincos(t, h, q, &pe[q], &pe[half + q]);
    pe[2 * half] = h ? x[2] : x[0];
    pe[2 * half + 1] = h ? 0.f : x[1];
#pragma unroll
    for (int q = 2 * half + 2; q < KS; ++q) pe[q] = 0.f;
}

// Row f1 (FUSED): the feature gather runs inside this kernel.  A lane (sample j, half hh) works out its own operands of the two bias
// nets straight from the channel-last sources -- nothing per sample is read but z, nothing is written but raw:
//   stage-1 volume (hh = 0) / stage-2 volume (hh = 1): all eight channels            -> bd step 0
//   stage-3 volume: channels 4hh .. 4hh+3 of all eight corners                       -> bd step 1, elements 0..3
//   source view 2p + hh of pair p: colour + mask -> bd (elements 4..7 of step 1, then two pairs per step), image features -> bc step p
//   reference projection, confidence: both halves (same values)
// Arithmetic and accumulation order per feature are gather_cl.hip's (bit-identical features); the weight stream is packed in this
// operand order (precision 3, build_pack_index_bf16).
struct FusedGather {
    int S, V, H, W;
    int vol_d[3], vol_h[3], vol_w[3];
    // the channel-last sources, each its own array (ABI v5, ucnerf_cl_sources: read in place or repacked); offsets inside one are 32-bit
    const char* vol[3];      // [D,h,w,8]
    const char* feat;        // [V,H,W,8] image features
    const char* col;         // [V,H,W,col_px / 4] colours
    unsigned col_px;         // bytes per colour pixel: 12 or 16 (bf16: 8)
    const float* conf;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* near_far;
    float near, far;
    float w2c_ref[12], K_ref[9];
    const float* w2cs;
    const float* Ks;
    unsigned div_m, div_sh;
    // COORDS instantiation: coordinates GIVEN by the caller (what rendering() of the reference receives from build_rays / build_rays_test,
    // network/renderer.py:215-255) instead of derived from (ray, depth): world points, the three stage copies, the encoded copy -- [M,3] each
    const float* pts_in;
    const float* ndc_in[3];
    const float* ndc_enc;
    int s16;                 // the channel-last arrays hold bf16 (ucnerf_cl_sources.bf16): 16-byte voxels / feature pixels, 8-byte colours
    // RAYGEN instantiation (ABI v4 gen_rays / gen_depths: ucnerf_ray_gen_sample folded into this launch): pixels and jitter draws in, and the rays,
    // depths and view-direction features the launch generates are WRITTEN for the launches behind it (compositing, re-sampling, the fine pass)
    const float* gen_xs;     // [n] pixel columns / rows
    const float* gen_ys;
    const float* gen_noise;  // [n,S] or NULL (perturb == 0)
    float gen_K[4];          // K00, K02, K11, K12 of the target camera
    float gen_R[12];         // its c2w, row-major 3x4
    float gen_Q[12];         // rotation of the view-direction feature (w2c_dir)
    float gen_perturb;
    int gen_lindisp;
    float* gen_rays_d;       // [n,3] out
    float* gen_z;            // [n,S] out
    float* gen_angle;        // [n,3] out
    // TAIL instantiation (passes of at most three rounds of tiles): tiles are dealt in whole rays to blocks (tail_rpb rays = tail_rpb * tail_tpr
    // consecutive tiles per block) and, when its last tile is done, a block composites its rays itself (K7, composite_device.h) and -- coarse
    // pass -- draws the fine depths from them (K8 + K9, sample_pdf_device.h): one launch for K3 .. K9 of the pass
    int tail_rpb, tail_tpr, tail_resample;
    int tail_spb;            // samples per block = tail_rpb * S: the block's tiles start at ITS first sample (round 5: S need not be a multiple of 32,
                             // e.g. the 90 cascade samples of rendering()), so a block's last tile may be partly filled
    // view-direction features made in the block's prologue from the rays' directions (round 5: rendering() hands over rays_d and a rotation that
    // lives on the device -- no ucnerf_dir_feature launch): angle = (d / |d|) @ Q^T written to tail_dir_out [n,3], which the tiles then read
    const float* tail_dir_Q;     // [>=3,4] DEVICE, or NULL: the features are given (ucnerf_mlp_params.dirs)
    float* tail_dir_out;
    ucnerf_composite_params tail_c;
    ucnerf_sample_pdf_params tail_s;
};
[[maybe_unused]] constexpr int FUSED_MAX_V = 8;    // (seven and eight views: with a two-slot weight ring, fused_ring_slots)
constexpr int VIEW_TAB = 24;      // floats per source view in the LDS table: w2c (12), K (9), pad

template <bool TILED, int NSRC, int TERMS, int SAVE, bool FUSED = false, bool COORDS = false, bool S16 = false, bool RAYGEN = false, bool TAIL = false>       // TAIL (FUSED only): see FusedGather; RAYGEN (FUSED only): rays and stratified depths generated in the tile prologue; TERMS 3: split-bf16 (fp32-grade), 1: plain bf16 (the hi*hi term only); SAVE: training forward keeping the activation sets (1: fp32, 2: the 24-bit format of p24.h); COORDS (FUSED only): sample coordinates given; S16 (FUSED only): bf16 channel-last sources
#ifndef UCNERF_BF16_WPS
#define UCNERF_BF16_WPS 2      // waves per SIMD: 2 -> 256 VGPRs per wave, 1 -> 512
#endif
__global__ void __launch_bounds__(64 * BW, UCNERF_BF16_WPS) mlp_fwd_bf16_kernel(ucnerf_mlp_params p, BGeom g, int n_tiles, MlpSaved sv, FusedGather fg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // ONE shared object: [ring][constants][pe stash]
    // ring depth: the operand stash of seven and eight source views (FUSED) takes the room of two ring slots; a two-slot ring was measured
    // equal at full grids and 1.3 % slower on half-empty ones (profiles/r02_logs/r02_ring_two_slots.log)
    constexpr int NB = fused_ring_slots(FUSED, NSRC);
    char* ring = smem;
    float* cst = reinterpret_cast<float*>(smem + NB * SLOT_BYTES);
    Frag* stash_all = reinterpret_cast<Frag*>(smem + NB * SLOT_BYTES + ((CONST_FLOATS * 4 + 15) & ~15));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
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
    float* ustash_w = reinterpret_cast<float*>(stash_all + (size_t)BW * KS16_PE_PTS * 64) + wave * 64;          // per-sample scalar parked across the tile
    // FUSED: [view table][operands of the confidence-bias net, gathered one tile ahead: step q of this lane at bcst_of(q)]
    float* vtab = reinterpret_cast<float*>(stash_all + (size_t)BW * KS16_PE_PTS * 64) + BW * 64;
    auto bcst_of = [&](int q) { return vtab + 8 * VIEW_TAB + ((size_t)(wave * KC_S + q) * 64 + opaque(lane)) * 8; };      // (re-derived at each use: not a loop-long register)
    if (FUSED) {
        for (int i = threadIdx.x; i < fg.V * VIEW_TAB; i += 64 * BW) {
            const int vi = i / VIEW_TAB, e = i % VIEW_TAB;
            vtab[i] = e < 12 ? fg.w2cs[12 * vi + e] : e < 21 ? fg.Ks[9 * vi + e - 12] : 0.f;
        }
        if (TAIL && fg.gen_xs) {
            // ABI v4 gen_rays / gen_depths on the tail route: a block owns whole rays, so it makes THEM first -- ray, view-direction feature and the
            // stratified depths (raygen_device.h: ucnerf_ray_gen_sample's arithmetic and bits), a handful of values per thread, written where the
            // tiles below (and the launches behind this one) read them.  The stores are acknowledged before the barrier that follows.
            const int bx0 = (int)blockIdx.x, nbk = (int)gridDim.x;
            const int lb = (nbk & 7) == 0 ? (bx0 & 7) * (nbk >> 3) + (bx0 >> 3) : bx0;
            const int ray0 = lb * fg.tail_rpb, nr = fg.tail_c.n - ray0 < fg.tail_rpb ? fg.tail_c.n - ray0 : fg.tail_rpb;
            for (int r = threadIdx.x; r < nr; r += 64 * BW) {
                const int ray = ray0 + r;
                float wx, wy, wz, ax, ay, az;
                pinhole_ray(fg.gen_xs[ray], fg.gen_ys[ray], fg.gen_K[0], fg.gen_K[1], fg.gen_K[2], fg.gen_K[3], fg.gen_R, &wx, &wy, &wz);
                view_dir_feature(wx, wy, wz, fg.gen_Q, &ax, &ay, &az);
                fg.gen_rays_d[3 * (size_t)ray] = wx; fg.gen_rays_d[3 * (size_t)ray + 1] = wy; fg.gen_rays_d[3 * (size_t)ray + 2] = wz;
                fg.gen_angle[3 * (size_t)ray] = ax; fg.gen_angle[3 * (size_t)ray + 1] = ay; fg.gen_angle[3 * (size_t)ray + 2] = az;
            }
            for (int i = threadIdx.x; i < nr * fg.S; i += 64 * BW) {
                const size_t k = (size_t)ray0 * fg.S + i;
                fg.gen_z[k] = stratified_depth(fg.near, fg.far, i % fg.S, fg.S, fg.gen_lindisp, fg.gen_perturb, fg.gen_noise ? fg.gen_noise[k] : 0.f);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        if (TAIL && fg.tail_dir_Q) {
            const int bx0 = (int)blockIdx.x, nbk = (int)gridDim.x;
            const int lb = (nbk & 7) == 0 ? (bx0 & 7) * (nbk >> 3) + (bx0 >> 3) : bx0;
            const int ray0 = lb * fg.tail_rpb, nr = fg.tail_c.n - ray0 < fg.tail_rpb ? fg.tail_c.n - ray0 : fg.tail_rpb;
            for (int r = threadIdx.x; r < nr; r += 64 * BW) {
                const size_t o = 3 * (size_t)(ray0 + r);
                float ax, ay, az;
                view_dir_feature(fg.rays_d[o], fg.rays_d[o + 1], fg.rays_d[o + 2], fg.tail_dir_Q, &ax, &ay, &az);      // (dir_feature_kernel's arithmetic)
                fg.tail_dir_out[o] = ax; fg.tail_dir_out[o + 1] = ay; fg.tail_dir_out[o + 2] = az;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        __syncthreads();
        if (TAIL && (fg.gen_xs || fg.tail_dir_Q)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    Pipe P;
    P.gsrc = ws + wave * (DMA_PER_SLOT * 1024) + lane * 16;
    P.ring = ring; P.ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring; P.buf = ring;
    P.wave = wave; P.gpos = 0; P.next_src = 0; P.slots = g.slots;
#pragma unroll
    for (int i = 0; i < NB; ++i) issue_dma<NB>(P, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * DMA_PER_SLOT) : "memory");       // slot 0 has landed ...
    __builtin_amdgcn_s_barrier();                                                        // ... for every wave
    AF cur = read_half(P.buf, lane, 0);                     // (TERMS 1 reads the unused lo halves once here)

    // Tile of (round, block, wave), WAVE-MAJOR inside a round: a last round that is only partly filled then keeps the same number of waves
    // busy in every block -- with at most half of it filled, one wave per SIMD, which renders a tile in 0.70 of the time two waves sharing the
    // SIMD take (58 k against 83 k cycles) -- instead of running some blocks full and leaving others empty.  A wave without a tile only
    // keeps the weight ring turning (idle_tile below): the 512-ray shard of a strongly-scaled batch is 0.5 + 1.5 such rounds.
    // (TAIL: block b owns the tiles of its rays, [b * t_blk, (b + 1) * t_blk), eight per round; its last rounds may be partly filled -- waves 0..3,
    //  one per SIMD, first)
    const int nb_ = (int)gridDim.x, bx_ = (int)blockIdx.x;
    const int lblock = (nb_ & 7) == 0 ? (bx_ & 7) * (nb_ >> 3) + (bx_ >> 3) : bx_;
    const int t_blk = TAIL ? (fg.tail_spb + 31) / 32 : 0;
    // TAIL: the block's own samples [s_base, s_lim) = its whole rays; tile j of the block covers samples s_base + 32 j ...: with the block's
    // tiles numbered lblock * t_blk + j, sample = 32 tile + s_off + lane (two block constants, s_off and s_lim, in scalar registers)
    const int s_off = TAIL ? lblock * (fg.tail_spb - 32 * t_blk) : 0;
    const int s_lim = TAIL ? ((lblock + 1) * fg.tail_spb < p.m ? (lblock + 1) * fg.tail_spb : p.m) : p.m;
    const int tiles_per_round = TAIL ? BW : gridDim.x * BW;
    const int n_rounds = TAIL ? (t_blk + BW - 1) / BW : (n_tiles + tiles_per_round - 1) / tiles_per_round;
#ifndef UCNERF_BF16_WAVE_MAJOR
#define UCNERF_BF16_WAVE_MAJOR 1
#endif
    // (workgroups go to the eight XCDs round-robin: block b counts as logical block (b % 8) * (blocks / 8) + b / 8, so that consecutive tiles --
    //  the 32-sample pieces of one ray, which gather from the same corner of the sources -- stay behind one XCD's L2)
    const int tile0 = TAIL ? lblock * t_blk + wave : UCNERF_BF16_WAVE_MAJOR ? wave * nb_ + lblock : bx_ * BW + wave;

    // Inputs of a tile are fetched one tile ahead (under the previous tile's head / blend arithmetic, when few registers
    // are live): operands of the depth-bias net (element e of step q = feature 16q + 8h + e; columns past a section's
    // width meet zero weights, so where the upper half's column would leave the row BOTH halves read the lower half's --
    // a constant offset again: the per-lane clamp it replaces cost the odd view counts 64-80 bytes of scratch), confidence, point.
    float nfs[4][8], nconf, npx[3];
    // (j and the parked-scalar address are re-derived from `lane` at each use: as loop-long values they get spilled)
    auto s_raw_of = [&](int tile) { return tile * 32 + s_off + (opaque(lane) & 31); };
    auto sample_of = [&](int tile) { const int s_raw = s_raw_of(tile); return s_raw < s_lim ? s_raw : s_lim - 1; };
    auto feat_base = [&](int s) { return TILED ? p.feats + ((size_t)(s >> 5) * F * 32 + (s & 31)) : p.feats + (size_t)s * g.feat_stride; };
    constexpr int fstride = TILED ? 32 : 1;
    auto fetch = [&](int tile) {
        const int s = sample_of(tile);
        const float* fb = feat_base(s);
        const float* fh = fb + 8 * opaque(h) * fstride;    // one per-lane base, constant offsets from it (re-derived: not a loop-long value)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 16 * q + e;                  // feature c + 8h
                // (tiled layout: streaming loads -- every 128-byte line is consumed by one load; kept out of the L2 they leave the
                //  gather's sources there.  Row-major rows are read four bytes at a time and need the cache.)
                const float* src_ = (NSRC && c + 8 < F) ? &fh[c * fstride] : &fb[(size_t)min(c, F - 1) * fstride];
                nfs[q][e] = q >= kd16 ? 0.f : TILED ? __builtin_nontemporal_load(src_) : *src_;
            }
        nconf = fb[(size_t)(F - 1) * fstride];
        const float* prow = p.pts + (size_t)s * 3;
        npx[0] = prow[0]; npx[1] = prow[1]; npx[2] = prow[2];
    };
    // FUSED: the same values, gathered from the sources (see FusedGather)
    // The in-kernel gather runs with eight waves per CU (the stand-alone kernel has 28) and all of them reach it together, so it is
    // staged: g_pre (depth, ray, depth range of the NEXT tile's sample: plain loads in the last fill of the last GEMM phase, after the
    // tile's last advance()); then, opening the next iteration of the tile loop: g_foot (footprints of every unit), g_issue_first (corner
    // loads of the volumes + confidence, back to back), the tile's point encoding while those are in flight, and g_finish: consume in
    // issue order, the next view pair's loads going out as soon as a volume's registers are free again.
    struct VolFp { unsigned o[4], dx; float w[4], wx0, wx1; };
    struct ImgFp { unsigned p00, p10, dx; float w00, w01, w10, w11, mask; };
    constexpr int NP = NSRC > 0 ? (NSRC + 1) / 2 : 1;     // view pairs: this lane's view of pair pr is 2 pr + hl
    float gz = 0.f, grd[3] = {0.f, 0.f, 0.f}, gnf[4] = {0.f, 0.f, 0.f, 0.f}, gqz = 0.f;
    float ro[3] = {0.f, 0.f, 0.f};                           // ray origin, read ONCE into scalar registers (a load inside a GEMM phase would bring a compiler-counted wait with it)
    if (FUSED) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ro[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fg.rays_o[i])));
    }
    VolFp fa, fb3;
    unsigned co[4];
    float cw[4];
    ImgFp fi[NP];
    float4 va[16], vb3[8], vi_[NP][12];
    float cv[4];
    float gcs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};           // COORDS: (u, v, z) of this lane's first volume and of stage 3
    auto g_pre = [&](int tile) {
        const unsigned s = (unsigned)sample_of(tile);
        if (COORDS) {                                        // world point -> grd, encoded copy -> npx (dead since this tile's encoding), stage copies -> gcs
            const int hl = opaque(h);
            const float* pw_ = fg.pts_in + 3 * (size_t)s;
            grd[0] = pw_[0]; grd[1] = pw_[1]; grd[2] = pw_[2];
            const float* pe_ = fg.ndc_enc + 3 * (size_t)s;
            npx[0] = pe_[0]; npx[1] = pe_[1]; npx[2] = pe_[2];
            const float* pa_ = (hl ? fg.ndc_in[1] : fg.ndc_in[0]) + 3 * (size_t)s;
            gcs[0] = pa_[0]; gcs[1] = pa_[1]; gcs[2] = pa_[2];
            const float* p3_ = fg.ndc_in[2] + 3 * (size_t)s;
            gcs[3] = p3_[0]; gcs[4] = p3_[1]; gcs[5] = p3_[2];
            return;
        }
        const unsigned r = fg.S == 1 ? s : (__umulhi(s, fg.div_m) >> fg.div_sh);
        if (RAYGEN) {                                        // pixel and jitter draw of the sample: g_gen() turns them into ray and depth
            grd[0] = fg.gen_xs[r]; grd[1] = fg.gen_ys[r];
            gz = fg.gen_noise ? fg.gen_noise[s] : 0.f;
            gnf[0] = gnf[2] = fg.near; gnf[1] = gnf[3] = fg.far;
            return;
        }
        gz = fg.z[s];
        const float* rd = fg.rays_d + 3 * (size_t)r;
        grd[0] = rd[0]; grd[1] = rd[1]; grd[2] = rd[2];
        const int hl = opaque(h);
        gnf[0] = gnf[2] = fg.near; gnf[1] = gnf[3] = fg.far;
        if (fg.near_far) {                                    // this lane's first volume (stage 1 + hl), then stage 3
            const float* nf = fg.near_far + 6 * (size_t)r;
            gnf[0] = nf[2 * hl]; gnf[1] = nf[2 * hl + 1]; gnf[2] = nf[4]; gnf[3] = nf[5];
        }
    };
    constexpr unsigned VOXB = S16 ? 16u : 32u;              // bytes per voxel / feature pixel of the channel-last arrays
    // a corner of a source view in slots 3c, 3c + 1 (features) and 3c + 2 (r, g, b); S16: the raw 16-bit values stay in the registers as loaded --
    // features in slot 3c, colours in 3c + 1 -- and become floats where they are consumed.  fi[].p00 / p10 / dx count PIXELS (view included)
    auto img_loads = [&](int pr) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {                         // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
            const unsigned ix = ((c & 2) ? fi[pr].p10 : fi[pr].p00) + ((c & 1) ? fi[pr].dx : 0u);
            if (S16) { vi_[pr][3 * c] = ld16(fg.feat, ix * VOXB); const float2 t_ = ld8(fg.col, ix * 8u); vi_[pr][3 * c + 1] = make_float4(t_.x, t_.y, 0.f, 0.f); }
            else {
                vi_[pr][3 * c] = ld16(fg.feat, ix * VOXB); vi_[pr][3 * c + 1] = ld16(fg.feat, ix * VOXB + 16u);
                const gf3u k_ = ld12(fg.col, __umul24(ix, fg.col_px)); vi_[pr][3 * c + 2] = make_float4(k_.x, k_.y, k_.z, 0.f);      // (.w is never read)
            }
        }
    };
    // footprints of the gather, in parts (so that they can be spread over several fills): 0 reference projection (+ the point the
    // encoding takes), 1 this lane's first volume, 2 stage-3 volume + confidence, 3 + pr view pair pr
    auto g_part = [&](int part) {
        constexpr bool so_ = true;
        const int gW = sopaque(fg.W, so_), gH = sopaque(fg.H, so_);
        const float x = COORDS ? grd[0] : ro[0] + gz * grd[0], y = COORDS ? grd[1] : ro[1] + gz * grd[1], w = COORDS ? grd[2] : ro[2] + gz * grd[2];
        const int hl = opaque(h);
        if (part == 0) {
            if (COORDS) return;                              // (the encoded copy came in as it is)
            float qx, qy, qz;
            project_cl(fg.w2c_ref, fg.K_ref, x, y, w, &qx, &qy, &qz);
            npx[0] = (qx / qz + 0.0f) / (float)(gW - 1); npx[1] = (qy / qz + 0.0f) / (float)(gH - 1);
            npx[2] = (qz - fg.near) / (fg.far - fg.near);
            gqz = qz;
            return;
        }
        const float qz = gqz;
        // volumes: `unit` is this lane's (hl in the first sweep, 2 in the second), `c0` the first byte of its channels in a voxel
        auto vol_fp = [&](int unit, unsigned c0, float nk, float fk) {
            const float u = COORDS ? gcs[unit == 2 ? 3 : 0] : npx[0], v = COORDS ? gcs[unit == 2 ? 4 : 1] : npx[1];
            const float zn = COORDS ? gcs[unit == 2 ? 5 : 2] : (qz - nk) / (fk - nk);
            // (wave-uniform sizes pass through an empty asm: their float forms are otherwise hoisted out of the tile loop into vector registers and spilled)
            const int D = unit == 2 ? sopaque(fg.vol_d[2], so_) : fg.vol_d[unit], hh = unit == 2 ? sopaque(fg.vol_h[2], so_) : fg.vol_h[unit], ww = unit == 2 ? sopaque(fg.vol_w[2], so_) : fg.vol_w[unit];
            const LerpCl ax = axis_cl(u * 2.f - 1.0f, ww, false), ay = axis_cl(v * 2.f - 1.0f, hh, false), az = axis_cl(zn * 2.f - 1.0f, D, false);
            const unsigned vb = c0;                          // (offsets inside the lane's own volume: fg.vol[hl] in the first sweep, fg.vol[2] in the second)
            VolFp f;
            f.o[0] = vb + (unsigned)((az.i0 * hh + ay.i0) * ww + ax.i0) * VOXB; f.o[1] = vb + (unsigned)((az.i0 * hh + ay.i1) * ww + ax.i0) * VOXB;
            f.o[2] = vb + (unsigned)((az.i1 * hh + ay.i0) * ww + ax.i0) * VOXB; f.o[3] = vb + (unsigned)((az.i1 * hh + ay.i1) * ww + ax.i0) * VOXB;
            f.dx = (unsigned)(ax.i1 - ax.i0) * VOXB;
            f.w[0] = az.w0 * ay.w0; f.w[1] = az.w0 * ay.w1; f.w[2] = az.w1 * ay.w0; f.w[3] = az.w1 * ay.w1;
            f.wx0 = ax.w0; f.wx1 = ax.w1;
            return f;
        };
        if (part == 1) { fa = vol_fp(hl, 0u, gnf[0], gnf[1]); return; }
        if (part == 2) {
            fb3 = vol_fp(2, (VOXB / 2) * hl, gnf[2], gnf[3]);
            const float u = COORDS ? gcs[3] : npx[0], v = COORDS ? gcs[4] : npx[1];                           // confidence: the stage-3 grid
            const LerpCl ax = axis_cl(u * 2.f - 1.0f, gW, false), ay = axis_cl(v * 2.f - 1.0f, gH, false);
            co[0] = (unsigned)(ay.i0 * gW + ax.i0); co[1] = (unsigned)(ay.i0 * gW + ax.i1);
            co[2] = (unsigned)(ay.i1 * gW + ax.i0); co[3] = (unsigned)(ay.i1 * gW + ax.i1);
            cw[0] = ay.w0 * ax.w0; cw[1] = ay.w0 * ax.w1; cw[2] = ay.w1 * ax.w0; cw[3] = ay.w1 * ax.w1;
            return;
        }
        const int pr = part - 3;
        if (pr < NP) {
            const int vi = min(2 * pr + hl, NSRC - 1);       // (a half without a view repeats the last one: finite values onto zero weights)
            const float* mt = vtab + vi * VIEW_TAB;
            float qx, qy, qv;
            project_cl(mt, mt + 12, x, y, w, &qx, &qy, &qv);
            const float gx = (qx / qv + 0.0f) / (float)(gW - 1) * 2.0f - 1.0f, gy = (qy / qv + 0.0f) / (float)(gH - 1) * 2.0f - 1.0f;
            const LerpCl ax = axis_cl(gx, gW, true), ay = axis_cl(gy, gH, true);
            fi[pr].p00 = (unsigned)((vi * gH + ay.i0) * gW + ax.i0); fi[pr].p10 = (unsigned)((vi * gH + ay.i1) * gW + ax.i0);
            fi[pr].dx = (unsigned)(ax.i1 - ax.i0);
            fi[pr].w00 = ay.w0 * ax.w0; fi[pr].w01 = ay.w0 * ax.w1; fi[pr].w10 = ay.w1 * ax.w0; fi[pr].w11 = ay.w1 * ax.w1;
            fi[pr].mask = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.f : 0.f;
        }
    };
    // RAYGEN: ray, depth and view-direction feature of this lane's sample from its pixel and jitter draw (raygen_device.h: the arithmetic of
    // ray_gen_sample_kernel, same bits), stored for the launches behind this one; the feature also goes to the sample's own slot of a scratch
    // row -- the lane pair reads it back in layer 5 instead of carrying three registers through the trunk
    // RAYGEN: ray and depth of this lane's sample from its pixel and jitter draw (raygen_device.h: the arithmetic of ray_gen_sample_kernel, same
    // bits).  Nothing is stored here: vector-memory operations retire in issue order, so a store in FRONT of the gather's corner loads puts its
    // write acknowledgement (~4 us at 512 rays) in front of every one of them, and values kept across the gather spill (208 bytes per lane for
    // five views and more).  The launch's side outputs -- z, rays_d, the view-direction feature -- are produced once more in layer 5, where the
    // non-generating kernel loads its direction row (see RAYGEN there).
    auto g_gen = [&](int tile) {
        const unsigned s = (unsigned)sample_of(tile);
        const unsigned r = fg.S == 1 ? s : (__umulhi(s, fg.div_m) >> fg.div_sh);
        float wx, wy, wz;
        pinhole_ray(grd[0], grd[1], fg.gen_K[0], fg.gen_K[1], fg.gen_K[2], fg.gen_K[3], fg.gen_R, &wx, &wy, &wz);
        gz = stratified_depth(fg.near, fg.far, (int)(s - r * (unsigned)fg.S), fg.S, fg.gen_lindisp, fg.gen_perturb, gz);
        grd[0] = wx; grd[1] = wy; grd[2] = wz;
    };
    auto g_foot = [&](int tile) {
        if (RAYGEN) g_gen(tile);
#pragma unroll
        for (int part = 0; part < 3 + NP; ++part) g_part(part);
    };
    auto g_issue_first = [&]() {
        SB0;
        // (the first sweep's volume differs between the lane halves and the two are separate allocations: a per-lane 64-bit base, re-derived here)
        const char* const va_base = opaque(h) ? fg.vol[1] : fg.vol[0];
#pragma unroll
        for (int c = 0; c < 8; ++c) {                         // corner c = (z, y, x): the accumulation order of gather_cl.hip
            const char* const o = va_base + (fa.o[c >> 1] + ((c & 1) ? fa.dx : 0u));
            if (S16) va[2 * c] = *(const float4*)o;
            else { va[2 * c] = *(const float4*)o; va[2 * c + 1] = *(const float4*)(o + 16); }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned o = fb3.o[c >> 1] + ((c & 1) ? fb3.dx : 0u);
            if (S16) { const float2 t_ = ld8(fg.vol[2], o); vb3[c] = make_float4(t_.x, t_.y, 0.f, 0.f); }
            else vb3[c] = ld16(fg.vol[2], o);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cv[c] = fg.conf[co[c]];
        SB0;
    };
    auto g_finish = [&]() {
        SB0;
        img_loads(0);
        SB0;
#pragma unroll
        for (int q = 2; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) nfs[q][e] = 0.f;          // (slots no view fills meet zero weights)
        {
            gf2 lo[2] = {{0, 0}, {0, 0}}, hi[2] = {{0, 0}, {0, 0}};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w__ = fa.w[c >> 1] * ((c & 1) ? fa.wx1 : fa.wx0);
                if (S16) { FMA4(lo, bf16x4_lo(va[2 * c]), w__) FMA4(hi, bf16x4_hi(va[2 * c]), w__) }
                else { FMA4(lo, va[2 * c], w__) FMA4(hi, va[2 * c + 1], w__) }
            }
            nfs[0][0] = lo[0].x; nfs[0][1] = lo[0].y; nfs[0][2] = lo[1].x; nfs[0][3] = lo[1].y;
            nfs[0][4] = hi[0].x; nfs[0][5] = hi[0].y; nfs[0][6] = hi[1].x; nfs[0][7] = hi[1].y;
        }
        SB0;
        if (NP > 1) img_loads(1);                             // (into the registers the first volume has just left)
        SB0;
        {
            gf2 lo[2] = {{0, 0}, {0, 0}};
#pragma unroll
            for (int c = 0; c < 8; ++c) { const float w__ = fb3.w[c >> 1] * ((c & 1) ? fb3.wx1 : fb3.wx0); FMA4(lo, S16 ? bf16x4_lo(vb3[c]) : vb3[c], w__) }
            nfs[1][0] = lo[0].x; nfs[1][1] = lo[0].y; nfs[1][2] = lo[1].x; nfs[1][3] = lo[1].y;
        }
        {
            float acc = cv[0] * cw[0];
            acc += cv[1] * cw[1];
            acc += cv[2] * cw[2];
            acc += cv[3] * cw[3];
            nconf = acc;
        }
        SB0;
        if (NP > 2) img_loads(2);
        SB0;
        auto view_consume = [&](int pr) {
            gf2 rg = {0, 0}, c1[2] = {{0, 0}, {0, 0}}, c2[2] = {{0, 0}, {0, 0}};      // (r g) b | (f0..f3) (f4..f7): one fused multiply-add per channel and corner
            float bl = 0.f;
            const float wt[4] = {fi[pr].w00, fi[pr].w01, fi[pr].w10, fi[pr].w11};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 k_ = S16 ? bf16x4_lo(vi_[pr][3 * c + 1]) : vi_[pr][3 * c + 2];
                if (S16) { FMA4(c1, bf16x4_lo(vi_[pr][3 * c]), wt[c]) FMA4(c2, bf16x4_hi(vi_[pr][3 * c]), wt[c]) }
                else { FMA4(c1, vi_[pr][3 * c], wt[c]) FMA4(c2, vi_[pr][3 * c + 1], wt[c]) }
                rg = __builtin_elementwise_fma((gf2){k_.x, k_.y}, (gf2){wt[c], wt[c]}, rg); bl = __builtin_fmaf(k_.z, wt[c], bl);
            }
            float* col = pr == 0 ? &nfs[1][4] : &nfs[2 + (pr - 1) / 2][4 * ((pr - 1) & 1)];
            col[0] = rg.x; col[1] = rg.y; col[2] = bl; col[3] = fi[pr].mask;
            f32x4* dst = reinterpret_cast<f32x4*>(bcst_of(pr));
            dst[0] = (f32x4){c1[0].x, c1[0].y, c1[1].x, c1[1].y};
            dst[1] = (f32x4){c2[0].x, c2[0].y, c2[1].x, c2[1].y};
        };
        view_consume(0);
        SB0;
        if (NP > 3) img_loads(3);                             // (seven and eight views: the fourth pair into the registers the first has just left)
        SB0;
#pragma unroll
        for (int pr = 1; pr < NP; ++pr) view_consume(pr);
    };
    // point encoding -> fragments in LDS (layer 0 and the skip connection read them from there)
    auto encode_point = [&]() {
        const float px[3] = {npx[0], npx[1], npx[2]};
        float pe[KS_PE_PTS];
        encode16<10, KS_PE_PTS>(px, opaque(h), pe);
#pragma unroll
        for (int q = 0; q < KS16_PE_PTS; ++q) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = pe[8 * q + e];
            stash[q * 64] = split8(t);
        }
    };
    if (FUSED) {                                            // (the gather itself opens every iteration of the tile loop)
        g_pre(tile0);
    } else fetch(tile0);

    for (int round = 0; round < n_rounds; ++round) {                          // block-uniform trip count: every wave joins every barrier
        const int tile = round * tiles_per_round + tile0;
#if UCNERF_BF16_IDLE_SKIP
        if (TAIL ? (round * BW + wave >= t_blk || tile * 32 + s_off >= s_lim) : tile >= n_tiles) {      // (wave-uniform; every later tile of this wave is past the end too)
            for (int i = 0; i < g.slots; ++i) advance<NB>(P);      // this wave's DMA pieces and barriers of one tile, nothing else
            cur = read_half(P.buf, lane, 0);
            continue;
        }
#endif
        if (FUSED) {
            // this tile's gather (its depth / ray / depth range came in under the previous tile's last GEMM phase); the point encoding
            // runs while the first loads are in flight.  ONE copy of this code: as a prologue before the loop it spilled 208 bytes per lane
            g_foot(tile);
            g_issue_first();
            encode_point(); g_finish();
            cur = read_half(P.buf, lane, 0);               // the fragments the last half-step left in `cur`, read again: sixteen registers the gather can use
        }
        // SAVE: this lane's row of an activation set (re-derived at every use: nothing tile-long is kept in a register)
        auto srow = [&](float* base) -> float* {
            const int l_ = opaque(lane), s_ = tile * 32 + (l_ & 31);
            if (SAVE == 2)       // 24-bit sets (p24.h, tiled): this lane's piece of group h; the piece of group 8 nt + 2 q + h is (8 nt + 2 q) * 384 bytes on
                return s_ < p.m ? base + (p24_offset((size_t)s_, l_ >> 5, 32) >> 2) : nullptr;
            return s_ < p.m ? base + (size_t)s_ * 128 + 4 * (l_ >> 5) : nullptr;
        };
        const size_t hstride = SAVE ? (size_t)(sv.h[1] - sv.h[0]) : 0;      // the six trunk sets are carved back to back
        // (few scalars are carried through the trunk -- every VGPR there is spoken for: sample index, feature base
        //  and view direction are re-derived / loaded where they are needed)
        __builtin_amdgcn_s_setprio(UCNERF_BF16_PRIO_VALU);
        f32x16 bd[4], acc[4];
        Frag X[8], Y[8];
        float fsec[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) fsec[q][e] = nfs[q][e];
        ustash_w[opaque(lane)] = 1.f - nconf;                             // u: only needed again at the very end of the tile

        // ---- point encoding -> fragments in LDS (layer 0 and the skip connection read them from there); FUSED: done at the previous tail
        if (!FUSED) {
            const float px[3] = {npx[0], npx[1], npx[2]};
            float pe[KS_PE_PTS];
            encode16<10, KS_PE_PTS>(px, h, pe);
#pragma unroll
            for (int q = 0; q < KS16_PE_PTS; ++q) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = pe[8 * q + e];
                stash[q * 64] = split8(t);
            }
        }
        __builtin_amdgcn_s_setprio(UCNERF_BF16_PRIO_GEMM);

        // ---- depth-bias net (step-major): bd = W_bd feats + b
        init_bias_pair(cst, SEC_BD, h, 0, bd);
        init_bias_pair(cst, SEC_BD, h, 1, bd);
        {
            Frag fc = split8(fsec[0]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < kd16) {
                    Frag fn;
                    HS(0, fc, bd[0], bd[1], if (q + 1 < 4 && q + 1 < kd16) { fn = split8(fsec[q + 1]); pin(fn.hi); pin(fn.lo); });
                    HS(1, fc, bd[2], bd[3], if (q + 1 >= kd16) init_bias_pair(cst, SEC_L0, h, 0, acc));
                    if (q + 1 < 4 && q + 1 < kd16) fc = fn;
                }
        }
        if (SAVE) {
            float* r_ = srow(sv.bd);
            if (r_) { save_tile<false, SAVE == 2>(r_, 0, bd[0]); save_tile<false, SAVE == 2>(r_, 1, bd[1]); save_tile<false, SAVE == 2>(r_, 2, bd[2]); save_tile<false, SAVE == 2>(r_, 3, bd[3]); }
        }

        // ---- layer 0 (pair-split) on the point encoding
        {
            Frag pf = stash[0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                     // phase A -> acc[0], acc[1]
                const Frag b = pf;
                HS(q & 1, b, acc[0], acc[1], pf = stash[((q + 1) & 3) * 64]; if (q == 2) init_bias_pair(cst, SEC_L0, h, 1, acc));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                     // phase B -> acc[2], acc[3]; epilogue of tiles 0,1 underneath
                const Frag b = pf;
                HS(q & 1, b, acc[2], acc[3],
                   if (q < 3) pf = stash[(q + 1) * 64];
                   X[q] = frag_of<2, SAVE>(acc[q >> 1], bd[q >> 1], q & 1, SAVE ? srow(sv.h[0]) : nullptr, q >> 1);
                   if (q == 3) init_bias_pair(cst, SEC_L0 + 1, h, 0, acc));
            }
        }

        // ---- layers 1..4 (pair-split, fragments ping-pong between X and Y).  In: in[0..3] + acc[2], acc[3] of the
        // previous layer (their epilogue -> in[4..7] runs under phase A); out: out[0..3] + acc[2], acc[3].
        auto layer128 = [&](Frag (&in)[8], Frag (&out)[8], int sec) {
            float* const h_prev = SAVE ? sv.h[0] + (size_t)(sec - SEC_L0 - 1) * hstride : nullptr;     // h of layer sec - SEC_L0 - 1 (row tiles 2, 3)
            float* const h_this = SAVE ? h_prev + hstride : nullptr;                                   // h of this layer (row tiles 0, 1)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                HS(q & 1, in[q], acc[0], acc[1],
                   if (q < 4) in[4 + q] = frag_of<2, SAVE>(acc[2 + (q >> 1)], bd[2 + (q >> 1)], q & 1, SAVE ? srow(h_prev) : nullptr, 2 + (q >> 1));
                   if (q == 4) init_bias_pair(cst, sec, h, 1, acc));
#pragma unroll
            for (int q = 0; q < 8; ++q)
                HS(q & 1, in[q], acc[2], acc[3],
                   if (!(q & 1)) out[q >> 1] = frag_of<2, SAVE>(acc[q >> 2], bd[q >> 2], (q >> 1) & 1, SAVE ? srow(h_this) : nullptr, q >> 2);
                   if (q == 7) init_bias_pair(cst, sec + 1, h, 0, acc));
        };
#pragma unroll 1
        for (int l = 1; l < 5; l += 2) {
            layer128(X, Y, SEC_L0 + l);
            layer128(Y, X, SEC_L0 + l + 1);
        }

        // ---- layer 5 on [h | pe] in k order h 0..3, pe 0..3, h 4..7; its output h5 = relu(. * bd) stays fp32 in acc
        // One batch of plain loads per tile (confidence-net operands + view direction, issued under phase B below):
        // a compiler-counted wait cannot see the weight DMAs and so drains them -- once here instead of per load site.
        int s_here = sample_of(tile);
        asm volatile("" : "+v"(s_here));                    // (re-derive the row pointer instead of carrying 64 bits through the trunk)
        const float* fb = feat_base(s_here);
        const float* fhb = fb + 8 * opaque(h) * fstride;
        float dv[3];
        {
            Frag pf;
#pragma unroll
            for (int q = 0; q < 12; ++q) {                                    // phase A
                const Frag b = q < 4 ? X[q] : (q < 8 ? pf : X[q - 4]);
                HS(q & 1, b, acc[0], acc[1],
                   if (q < 4) X[4 + q] = frag_of<2, SAVE>(acc[2 + (q >> 1)], bd[2 + (q >> 1)], q & 1, SAVE ? srow(sv.h[4]) : nullptr, 2 + (q >> 1));
                   if (q >= 3 && q < 7) pf = stash[(q - 3) * 64];
                   if (q == 4) init_bias_pair(cst, SEC_L0 + 5, h, 1, acc));
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) {                                    // phase B; h5 of tiles 0,1 underneath
                const Frag b = q < 4 ? X[q] : (q < 8 ? pf : X[q - 4]);
                HS(q & 1, b, acc[2], acc[3],
                   if (q >= 3 && q < 7) pf = stash[(q - 3) * 64];
                   if (q < 2) {
                       _Pragma("unroll")
                       for (int r = 0; r < 16; ++r) acc[q][r] = fmaxf(acc[q][r] * bd[q][r], 0.f);
                       pin(acc[q]);
                   }
                   if (q == 8) {                                              // operands of the confidence-bias net
                       if (FUSED) {
                           _Pragma("unroll")
                           for (int qq = 0; qq < 4; ++qq) {
                               const f32x4* src_ = reinterpret_cast<const f32x4*>(bcst_of(qq < kc16 ? qq : 0));
                               const f32x4 a_ = src_[0], b_ = src_[1];
                               fsec[qq][0] = a_.x; fsec[qq][1] = a_.y; fsec[qq][2] = a_.z; fsec[qq][3] = a_.w;
                               fsec[qq][4] = b_.x; fsec[qq][5] = b_.y; fsec[qq][6] = b_.z; fsec[qq][7] = b_.w;
                           }
                       } else {
                       _Pragma("unroll")
                       for (int qq = 0; qq < 4; ++qq)
                           _Pragma("unroll")
                           for (int e = 0; e < 8; ++e) {
                               const int c = f_img + 16 * qq + e;
                               const float* src_ = (NSRC && c + 8 < F) ? &fhb[c * fstride] : &fb[(size_t)min(c, F - 1) * fstride];
                               fsec[qq][e] = qq >= kc16 ? 0.f : TILED ? __builtin_nontemporal_load(src_) : *src_;
                           }
                       }
                       if (RAYGEN) {                           // pixel and jitter draw again (the ray's row of xs / ys, the sample's draw): three loads where the
                           const unsigned r_ = fg.S == 1 ? (unsigned)s_here : (__umulhi((unsigned)s_here, fg.div_m) >> fg.div_sh);     // direction row would be loaded
                           dv[0] = fg.gen_xs[r_]; dv[1] = fg.gen_ys[r_]; dv[2] = fg.gen_noise ? fg.gen_noise[s_here] : 0.f;
                       } else {
                       int ray = s_here;
                       if (!p.dirs_per_sample) { int S = p.S; asm volatile("" : "+s"(S)); ray = s_here / S; }   // (opaque: no reciprocal hoisted into a loop-long VGPR)
                       const float* drow = p.dirs + (size_t)ray * 3;
                       dv[0] = drow[0]; dv[1] = drow[1]; dv[2] = drow[2];
                       }
                   }
                   if (RAYGEN && q == 10) {                // the launch's side outputs, beside the MFMAs: depth of every sample, ray and feature once per ray
                       const unsigned r_ = fg.S == 1 ? (unsigned)s_here : (__umulhi((unsigned)s_here, fg.div_m) >> fg.div_sh);
                       const bool first = (unsigned)s_here == r_ * (unsigned)fg.S;
                       float wx, wy, wz, ax, ay, az;
                       pinhole_ray(dv[0], dv[1], fg.gen_K[0], fg.gen_K[1], fg.gen_K[2], fg.gen_K[3], fg.gen_R, &wx, &wy, &wz);
                       const float zz = stratified_depth(fg.near, fg.far, (int)((unsigned)s_here - r_ * (unsigned)fg.S), fg.S, fg.gen_lindisp, fg.gen_perturb, dv[2]);
                       view_dir_feature(wx, wy, wz, fg.gen_Q, &ax, &ay, &az);
                       if (opaque(h) == 0 && tile * 32 + (opaque(lane) & 31) < p.m) {
                           fg.gen_z[s_here] = zz;
                           if (first) {
                               float* rd = fg.gen_rays_d + 3 * (size_t)r_;
                               rd[0] = wx; rd[1] = wy; rd[2] = wz;
                               float* ao = fg.gen_angle + 3 * (size_t)r_;
                               ao[0] = ax; ao[1] = ay; ao[2] = az;
                           }
                       }
                       dv[0] = ax; dv[1] = ay; dv[2] = az;
                       pin(dv[0]); pin(dv[1]); pin(dv[2]);
                   });
            }
        }
#pragma unroll
        for (int nt = 2; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = fmaxf(acc[nt][r] * bd[nt][r], 0.f);
        if (SAVE) {
            float* r_ = srow(sv.h[5]);
            if (r_) { save_tile<false, SAVE == 2>(r_, 0, acc[0]); save_tile<false, SAVE == 2>(r_, 1, acc[1]); save_tile<false, SAVE == 2>(r_, 2, acc[2]); save_tile<false, SAVE == 2>(r_, 3, acc[3]); }
        }

        // ---- confidence-bias net (step-major) -> bd; base heads of row tiles 0,1 underneath
        HeadAcc hbase = {{0.f, 0.f}, {0.f, 0.f}};
        auto ident = [](float v) { return v; };
        init_bias_pair(cst, SEC_BC, h, 0, bd);
        init_bias_pair(cst, SEC_BC, h, 1, bd);
        {
            Frag fc = split8(fsec[0]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < kc16) {
                    Frag fn;
                    HS(0, fc, bd[0], bd[1],
                       if (q + 1 < 4 && q + 1 < kc16) { fn = split8(fsec[q + 1]); pin(fn.hi); pin(fn.lo); }
                       if (q < 2) head_part(hbase, hb, h, acc[q], q, 0, 8, ident));
                    HS(1, fc, bd[2], bd[3],
                       if (q < 2) head_part(hbase, hb, h, acc[q], q, 8, 8, ident));
                    if (q + 1 < 4 && q + 1 < kc16) fc = fn;
                }
        }
        if (kc16 < 2) head_part(hbase, hb, h, acc[1], 1, 0, 16, ident);        // (one-step bias net: the rest is exposed)
        if (SAVE) {
            float* r_ = srow(sv.bc);
            if (r_) { save_tile<false, SAVE == 2>(r_, 0, bd[0]); save_tile<false, SAVE == 2>(r_, 1, bd[1]); save_tile<false, SAVE == 2>(r_, 2, bd[2]); save_tile<false, SAVE == 2>(r_, 3, bd[3]); }
        }
        // g = h5 * b_c: fragments of row tiles 0,1 now, of 2,3 under feature_linear's phase A
#pragma unroll
        for (int q = 0; q < 4; ++q) X[q] = frag_of<1>(acc[q >> 1], bd[q >> 1], q & 1);
        init_bias_pair(cst, SEC_FT, h, 0, acc);

        // ---- feature_linear (pair-split); base heads of row tiles 2,3 underneath
#pragma unroll
        for (int q = 0; q < 8; ++q)
            HS(q & 1, X[q], acc[0], acc[1],
               if (q < 4) X[4 + q] = frag_of<1>(acc[2 + (q >> 1)], bd[2 + (q >> 1)], q & 1);
               if (q >= 4) head_part(hbase, hb, h, acc[2 + ((q - 4) >> 1)], 2 + ((q - 4) >> 1), (q & 1) * 8, 8, ident);
               if (q == 7) init_bias_pair(cst, SEC_FT, h, 1, acc);
               );
        const f32x4 base = head_finish(hbase, hb);
        float pd[KS_PE_DIR];
        const Rev2 dt[3] = {to_revolutions(dv[0]), to_revolutions(dv[1]), to_revolutions(dv[2])};
#pragma unroll
        for (int q = 0; q < 8; ++q)
            HS(q & 1, X[q], acc[2], acc[3],
               if (!(q & 1)) Y[q >> 1] = frag_of<0, SAVE>(acc[q >> 2], acc[q >> 2], (q >> 1) & 1, SAVE ? srow(sv.ft) : nullptr, q >> 2);
               if (q & 1) {                                                    // direction encoding: 6 arguments, two per odd half-step
                   if (q < 7) {
                       const int hq_ = FUSED ? opaque(h) : h;           // (FUSED: keeps the per-lane frequency selects out of loop-long registers)
                       encode_sincos(dt, hq_, q - 1, &pd[q - 1], &pd[6 + q - 1]);
                       encode_sincos(dt, hq_, q, &pd[q], &pd[6 + q]);
                       pin(pd[q - 1]); pin(pd[q]); pin(pd[5 + q]); pin(pd[6 + q]);
                   } else init_bias_pair(cst, SEC_VC, h, 0, acc);
               });
        pd[12] = h ? dv[2] : dv[0];
        pd[13] = h ? 0.f : dv[1];
        pd[14] = 0.f; pd[15] = 0.f;

        // ---- views_linears | view_confi_linears on [feature | dir encoding] (pair-split), relu
        Frag D[2];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const Frag b = q < 8 ? Y[q] : D[q - 8];
            HS(q & 1, b, acc[0], acc[1],
               if (q < 4) Y[4 + q] = frag_of<0, SAVE>(acc[2 + (q >> 1)], acc[2 + (q >> 1)], q & 1, SAVE ? srow(sv.ft) : nullptr, 2 + (q >> 1));
               if (q == 4) init_bias_pair(cst, SEC_VC, h, 1, acc);
               if (q == 5 || q == 6) {
                   float t[8];
                   _Pragma("unroll")
                   for (int e = 0; e < 8; ++e) t[e] = pd[8 * (q - 5) + e];
                   D[q - 5] = split8(t); pin(D[q - 5].hi); pin(D[q - 5].lo);
               }
               );
        }
        HeadAcc hadapt = {{0.f, 0.f}, {0.f, 0.f}};
        auto relu = [](float v) { return fmaxf(v, 0.f); };
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const Frag b = q < 8 ? Y[q] : D[q - 8];
            HS(q & 1, b, acc[2], acc[3],
               if (q < 4) head_part(hadapt, ha, h, acc[q >> 1], q >> 1, (q & 1) * 8, 8, relu);
               if (FUSED && q == 9) g_pre(tile + tiles_per_round);              // (after the tile's last advance(): its counted wait would sit on these loads too; clamped past the end: harmless)
               if (SAVE && (q == 4 || q == 5)) { float* r_ = srow(sv.vc); if (r_) save_tile<true, SAVE == 2>(r_, q - 4, acc[q - 4]); });
        }
        __builtin_amdgcn_s_setprio(UCNERF_BF16_PRIO_VALU);
        if (!FUSED) fetch(tile + tiles_per_round);         // next tile's inputs (clamped past the end: harmless)
        if (SAVE) {
            float* r_ = srow(sv.vc);
            if (r_) { save_tile<true, SAVE == 2>(r_, 2, acc[2]); save_tile<true, SAVE == 2>(r_, 3, acc[3]); }
        }
        // ---- adapt heads of row tiles 2,3, uncertainty blend
        head_part(hadapt, ha, h, acc[2], 2, 0, 16, relu);
        head_part(hadapt, ha, h, acc[3], 3, 0, 16, relu);
        const f32x4 adapt = head_finish(hadapt, ha);
        const float u = ustash_w[opaque(lane)], omu = 1.f - u;
        const int s_raw = s_raw_of(tile);
        f32x4 out;
        out.x = 1.f / (1.f + expf(-(base.x * omu + adapt.x * u)));
        out.y = 1.f / (1.f + expf(-(base.y * omu + adapt.y * u)));
        out.z = 1.f / (1.f + expf(-(base.z * omu + adapt.z * u)));
        out.w = fmaxf(adapt.w * omu + base.w * u, 0.f);
        if (h == 0 && s_raw < s_lim) reinterpret_cast<f32x4*>(p.raw)[s_raw] = out;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
    if (TAIL) {
        // ---- K7 (+ K8, K9) of this block's rays.  Every output of every tile of these rays was stored by a wave of THIS block: the stores have been
        // acknowledged (vmcnt(0) above), the barrier orders them before the loads below, which miss the CU's vector cache (nobody has read these
        // lines in this launch) and find them in the XCD's L2.  One wave per ray, the stand-alone kernels' device code: bit-identical outputs.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        typedef PdfShared<128, 512> Sh;                      // (the ring, the constants and the stashes are dead: the rays' LDS arrays take their place)
        Sh& sh = reinterpret_cast<Sh*>(smem)[wave];
        float* const wl = reinterpret_cast<float*>(smem + BW * sizeof(Sh)) + wave * (128 + 2);
        const int S_ = fg.tail_c.S, E_ = composite_lane_samples(S_);
        for (int r = wave; r < fg.tail_rpb; r += BW) {
            const int ray = lblock * fg.tail_rpb + r;        // (wave-uniform)
            if (ray >= fg.tail_c.n) break;
            float* const keep = fg.tail_resample ? wl : nullptr;
            if (E_ == 1) composite_ray<1, 0>(fg.tail_c, ray, lane, keep);
            else if (E_ == 2) composite_ray<2, 0>(fg.tail_c, ray, lane, keep);
            else if (E_ == 3) composite_ray<3, 0>(fg.tail_c, ray, lane, keep);
            else composite_ray<4, 0>(fg.tail_c, ray, lane, keep);
            if (fg.tail_resample) {
                pdf_sync<true>();
                sample_pdf_ray<128, 512, true>(fg.tail_s, ray, lane, sh, wl, fg.tail_c.z + (size_t)ray * S_);
                pdf_sync<true>();                            // (the next ray of this wave re-uses the arrays)
            }
        }
    }
}

constexpr size_t bf16_smem_bytes(int ring_slots = NBUF) {
    return (size_t)ring_slots * SLOT_BYTES + ((CONST_FLOATS * 4 + 15) & ~15) + (size_t)BW * KS16_PE_PTS * 64 * sizeof(Frag) + (size_t)BW * 64 * sizeof(float);
}

constexpr size_t bf16_smem_bytes_fused(int v) {
    return bf16_smem_bytes(fused_ring_slots(true, v)) + 8 * VIEW_TAB * sizeof(float) + (size_t)BW * ((8 * v + 15) / 16) * 64 * 8 * sizeof(float);
}
static_assert(bf16_smem_bytes_fused(6) <= 160 * 1024 && bf16_smem_bytes_fused(8) <= 160 * 1024, "the fused kernel's LDS image must fit the CU");

// `save` (TERMS = 3 only): the training forward -- the activation sets of MlpSaved are written for ucnerf_mlp_bwd (saved_valid = 1)
#if UCNERF_BF16_BUILD_TAIL
// the TAIL instantiations (this build's only code objects): `blocks` = ceil(rays / fg->tail_rpb)
int launch_fused_tail(const ucnerf_mlp_params* p, const BGeom* g, int n_tiles, const FusedGather* fg, int blocks, hipStream_t st) {
    MlpSaved sv;
    memset(&sv, 0, sizeof(sv));
    const int v = p->cfg.n_src;
    const size_t smem_f = bf16_smem_bytes_fused(v);
    static_assert(BW * (sizeof(PdfShared<128, 512>) + (128 + 2) * sizeof(float)) <= bf16_smem_bytes_fused(1), "the rays' LDS arrays re-use the kernel's image");
    dim3 grid(blocks), block(64 * BW);
#define X(N)                                                                                                                   \
    if (v == N && !fg->pts_in) {                                                                                               \
        const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, false, true, false, false, false, true>;                 \
        if (int rc = ensure_dynamic_lds(fn, (int)smem_f, "mlp_fwd (bf16x3, gather fused, compositing in the tail)")) return rc; \
        hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, false, true, false, false, false, true>), grid, block, smem_f, st, *p, *g, n_tiles, sv, *fg); \
    }                                                                                                                          \
    if (v == N && fg->pts_in) {      /* coordinates given: what rendering() hands over (network/renderer.py:215-255) */       \
        const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, false, true, true, false, false, true>;                  \
        if (int rc = ensure_dynamic_lds(fn, (int)smem_f, "mlp_fwd (bf16x3, gather fused, given coordinates, compositing in the tail)")) return rc; \
        hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, false, true, true, false, false, true>), grid, block, smem_f, st, *p, *g, n_tiles, sv, *fg); \
    }
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
    return check_launch("mlp_fwd (bf16x3, gather fused, compositing in the tail)");
}
#else
int launch_fused_tail(const ucnerf_mlp_params* p, const BGeom* g, int n_tiles, const FusedGather* fg, int blocks, hipStream_t st);      // the TAIL build of this file

static int launch_bf16(const ucnerf_mlp_params* p, const MlpSaved* save, hipStream_t st, const FusedGather* fuse = nullptr) {
    UCNERF_REQUIRE(p, "mlp_fwd: null params");
    if (p->m == 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->dirs && p->wstream && p->raw && (fuse || (p->pts && p->feats)), "mlp_fwd: null pointer");
    UCNERF_REQUIRE((p->cfg.precision == 3) == (fuse != nullptr), "mlp_fwd: a weight stream packed with precision 3 serves the render pass with the gather "
                   "fused into the MLP kernel and nothing else (ucnerf_render_fused_fwd)");
    UCNERF_REQUIRE(!p->encoded && !p->pts_stride && !p->dirs_stride, "mlp_fwd (bf16x3): encoded / strided inputs are only available in f32 precision");
    UCNERF_REQUIRE(p->dirs_per_sample || p->S > 0, "mlp_fwd: S must be > 0 when dirs are per ray");
    UCNERF_REQUIRE(((uintptr_t)p->wstream & 15) == 0 && ((uintptr_t)p->raw & 15) == 0, "mlp_fwd: wstream/raw must be 16-byte aligned");
    Bf16Layout B;
    UCNERF_REQUIRE(bf16_layout(p->cfg.n_src, &B), "mlp_fwd: n_src %d outside 1..8", p->cfg.n_src);
    const int n_tiles = cdiv(p->m, 32);
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_fwd: no device");
    int blocks = cdiv(n_tiles, BW);
    const int cap = p->max_blocks > 0 ? p->max_blocks : UCNERF_BF16_WPS * 4 / BW * cus;      // blocks per CU = waves per SIMD * 4 / BW
    // fewer tiles than wave slots: rather every CU with one wave per SIMD than half the CUs with two (tiles are dealt wave-major, see the kernel)
    if (BW == 8 && UCNERF_BF16_IDLE_SKIP && blocks < cap) { const int spread = cdiv(n_tiles, 4); blocks = spread < cap ? spread : cap; }
    if (blocks > cap) blocks = cap;
    BGeom g;
    g.F = B.F; g.kd16 = B.kd16; g.kc16 = B.kc16; g.f_img = 24 + 4 * B.v; g.slots = B.slots;
    g.feat_stride = p->feat_stride ? p->feat_stride : B.F;
    g.const_off_bytes = (int)B.const_off_bytes;
    // One instantiation per source-view count (1..8: SCARED scripts 6, Hamlyn 3, the reference's opt.py default 4, ...):
    // with the section lengths known at compile time no instantiation carries the spills of a runtime-length version.
    // This translation unit is compiled twice (uc_nerf_amd/build.py): -DUCNERF_BF16_BUILD_TERMS=3 (bf16x3) and =1 (bf16).
    const size_t smem = bf16_smem_bytes();
    dim3 grid(blocks), block(64 * BW);
    const bool tiled = p->feats_tiled != 0;
    constexpr int K = UCNERF_BF16_BUILD_TERMS;
    MlpSaved sv;
    memset(&sv, 0, sizeof(sv));
    FusedGather fg;
    memset(&fg, 0, sizeof(fg));
#define UCNERF_BF16_FOR_ALL(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#if UCNERF_BF16_BUILD_TERMS == 3
    if (fuse) {
        UCNERF_REQUIRE(!save && B.v <= FUSED_MAX_V, "mlp_fwd (gather fused): inference forward, n_src <= %d", FUSED_MAX_V);
        fg = *fuse;
        UCNERF_REQUIRE(!(fg.s16 && fg.pts_in), "mlp_fwd (gather fused): bf16 channel-last sources are served on derived coordinates only (given coordinates: fp32 copies, "
                       "or the two-kernel pass)");
        if (fg.tail_rpb > 0) {      // the launch composites its rays itself (small passes, render.hip): whole rays per block, which it also generates when asked to
            UCNERF_REQUIRE(!fg.s16 && !(fg.gen_xs && (fg.near_far || fg.pts_in)) && p->max_blocks <= 0, "mlp_fwd (gather fused): compositing in the tail goes with fp32 sources (and generated rays with derived coordinates)");
            return launch_fused_tail(p, &g, n_tiles, &fg, cdiv(fg.tail_c.n, fg.tail_rpb), st);
        }
        UCNERF_REQUIRE(!fg.gen_xs || (!fg.s16 && !fg.pts_in && !fg.near_far && B.v <= 6), "mlp_fwd (gather fused): generated rays go with fp32 source copies, derived "
                       "coordinates, the scene's depth range and at most six source views");
        const size_t smem_f = bf16_smem_bytes_fused(B.v);
#define X(N)                                                                                                                   \
        if (B.v == N && !fg.pts_in && fg.s16) {                                                                                \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, false, true, false, true>;                           \
            if (int rc = ensure_dynamic_lds(fn, (int)smem_f, "mlp_fwd (bf16x3, gather fused, bf16 sources)")) return rc;      \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, false, true, false, true>), grid, block, smem_f, st, *p, g, n_tiles, sv, fg); \
        }                                                                                                                      \
        if (B.v == N && !fg.pts_in && !fg.s16 && !fg.gen_xs) {                                                                 \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, false, true>;                                        \
            if (int rc = ensure_dynamic_lds(fn, (int)smem_f, "mlp_fwd (bf16x3, gather fused)")) return rc;                    \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, false, true>), grid, block, smem_f, st, *p, g, n_tiles, sv, fg); \
        }                                                                                                                      \
        if (B.v == N && fg.pts_in) {                                                                                           \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, false, true, true>;                                  \
            if (int rc = ensure_dynamic_lds(fn, (int)smem_f, "mlp_fwd (bf16x3, gather fused, given coordinates)")) return rc; \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, false, true, true>), grid, block, smem_f, st, *p, g, n_tiles, sv, fg); \
        }
        UCNERF_BF16_FOR_ALL(X)
#undef X
        // rays generated inside the launch (ABI v4 gen_rays / gen_depths): one to six source views (seven and eight spill 24 bytes per lane)
#define X(N)                                                                                                                   \
        if (B.v == N && fg.gen_xs) {                                                                                           \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, false, true, false, false, true>;                    \
            if (int rc = ensure_dynamic_lds(fn, (int)smem_f, "mlp_fwd (bf16x3, gather fused, rays generated)")) return rc;    \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, false, true, false, false, true>), grid, block, smem_f, st, *p, g, n_tiles, sv, fg); \
        }
        X(1) X(2) X(3) X(4) X(5) X(6)
#undef X
        return check_launch("mlp_fwd (bf16x3, gather fused)");
    }
    if (save) {
        sv = *save;
        UCNERF_REQUIRE(sv.p24 || !tiled, "mlp_fwd_train (bf16x3): fp32 activation sets serve the layer-by-layer backward, which reads row-major features");
#define X(N)                                                                                                                   \
        if (B.v == N && !tiled && !sv.p24) {                                                                                   \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<false, N, 3, 1>;                                                 \
            if (int rc = ensure_dynamic_lds(fn, (int)smem, "mlp_fwd_train (bf16x3)")) return rc;                              \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, N, 3, 1>), grid, block, smem, st, *p, g, n_tiles, sv, fg);              \
        }                                                                                                                      \
        if (B.v == N && !tiled && sv.p24) {                                                                                    \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<false, N, 3, 2>;                                                 \
            if (int rc = ensure_dynamic_lds(fn, (int)smem, "mlp_fwd_train (bf16x3, 24-bit sets)")) return rc;                 \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, N, 3, 2>), grid, block, smem, st, *p, g, n_tiles, sv, fg);              \
        }                                                                                                                      \
        if (B.v == N && tiled) {                                                                                               \
            const void* fn = (const void*)mlp_fwd_bf16_kernel<true, N, 3, 2>;                                                  \
            if (int rc = ensure_dynamic_lds(fn, (int)smem, "mlp_fwd_train (bf16x3, tiled features, 24-bit sets)")) return rc; \
            hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, 3, 2>), grid, block, smem, st, *p, g, n_tiles, sv, fg);               \
        }
        UCNERF_BF16_FOR_ALL(X)
#undef X
        return check_launch("mlp_fwd_train (bf16x3)");
    }
#else
    UCNERF_REQUIRE(!save, "mlp_fwd_train: plain bf16 has no training forward");
#endif
#define X(N)                                                                                                                   \
    if (B.v == N) {                                                                                                            \
        const void* fn = tiled ? (const void*)mlp_fwd_bf16_kernel<true, N, K, false> : (const void*)mlp_fwd_bf16_kernel<false, N, K, false>; \
        if (int rc = ensure_dynamic_lds(fn, (int)smem, "mlp_fwd (bf16)")) return rc;                                          \
        if (tiled) hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, N, K, false>), grid, block, smem, st, *p, g, n_tiles, sv, fg);    \
        else hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, N, K, false>), grid, block, smem, st, *p, g, n_tiles, sv, fg);         \
    }
    UCNERF_BF16_FOR_ALL(X)
#undef X
#undef UCNERF_BF16_FOR_ALL
    return check_launch("mlp_fwd_bf16");
}

// (the timing-experiment switches of rounds 1-3 -- UCNERF_BF16_EXP, NO_PK, SPLIT_DOT, FUSED_FOOT_UNDER_GEMM, the DIAG stamps -- were taken out of
//  this file in round 4, their numbers are in profiles/r0N_experiments.md and DESIGN.md; what is left are structural parameters)
#define UCNERF_BF16_FLAGS UCNERF_FLAG(UCNERF_BF16_BW) UCNERF_FLAG(UCNERF_BF16_NBUF) UCNERF_FLAG(UCNERF_BF16_WPS) UCNERF_FLAG(UCNERF_BF16_HINT_V) \
    UCNERF_FLAG(UCNERF_BF16_PRIO_VALU) UCNERF_FLAG(UCNERF_BF16_PRIO_GEMM) UCNERF_FLAG(UCNERF_BF16_IDLE_SKIP) UCNERF_FLAG(UCNERF_BF16_WAVE_MAJOR)
#if UCNERF_BF16_BUILD_TERMS == 3
const char* build_flags_mlp_bf16x3() { return "mlp_bf16x3: " UCNERF_BF16_FLAGS; }
#else
const char* build_flags_mlp_bf16_plain() { return "mlp_bf16_plain: " UCNERF_BF16_FLAGS; }
#endif

#if UCNERF_BF16_BUILD_TERMS == 3
#if !UCNERF_OPERAND_FP16
int launch_mlp_fwd_bf16x3_h16(const ucnerf_mlp_params* p, hipStream_t st);                   // the fp16-term builds of this file (cfg.operand == 1)
int launch_mlp_fwd_bf16x3_gather_h16(const ucnerf_render_params* rp, const float* dirs, float* raw, hipStream_t st, const ucnerf_composite_params* tail_c,
                                     const ucnerf_sample_pdf_params* tail_s, float* tail_dir_out);
int launch_pack_bf16_h16(const ucnerf_mlp_config* cfg, const float* flat, const int32_t* idx, float* out, hipStream_t st);
int launch_pack_bf16_tab_h16(const ucnerf_mlp_config* cfg, const ParamTable& t, const int32_t* idx, float* out, hipStream_t st);
#endif
int launch_mlp_fwd_bf16x3(const ucnerf_mlp_params* p, hipStream_t st) {
    UCNERF_REQUIRE(p, "mlp_fwd: null params");
    UCNERF_REQUIRE(p->cfg.operand == 0 || p->cfg.operand == 1, "mlp_fwd: cfg.operand %d (0 = bf16 terms, 1 = fp16 terms)", p->cfg.operand);
    UCNERF_OPERAND_DISPATCH(p->cfg.operand, launch_mlp_fwd_bf16x3_h16(p, st));
    return launch_bf16(p, nullptr, st);
}
int launch_mlp_fwd_bf16x3_save(const ucnerf_mlp_params* p, const MlpSaved* save, hipStream_t st) {
    UCNERF_REQUIRE(p && p->cfg.operand == 0, "mlp_fwd_train: the training forward keeps its activations for a backward that splits them into bf16 terms "
                   "(cfg.operand 0); fp16 terms serve the inference forward");
    return launch_bf16(p, save, st);
}

// called by render.hip: gather + PE + MLP of one pass in ONE launch (row f1), from the channel-last sources and (ray, depth)
// `tail_c` (optional): the launch also composites the pass's rays (and, with `tail_s`, re-samples from them) -- see FusedGather
int check_cl_sources(const ucnerf_render_params* p, const char* who);      // gather_cl.hip
int launch_mlp_fwd_bf16x3_gather(const ucnerf_render_params* rp, const float* dirs, float* raw, hipStream_t st,
                                 const ucnerf_composite_params* tail_c, const ucnerf_sample_pdf_params* tail_s, float* tail_dir_out) {
    UCNERF_REQUIRE(rp->cfg.operand == 0 || rp->cfg.operand == 1, "render (gather fused): cfg.operand %d (0 = bf16 terms, 1 = fp16 terms)", rp->cfg.operand);
    UCNERF_OPERAND_DISPATCH(rp->cfg.operand, launch_mlp_fwd_bf16x3_gather_h16(rp, dirs, raw, st, tail_c, tail_s, tail_dir_out));
    const long long M = (long long)rp->n * rp->S;
    UCNERF_REQUIRE(M < (1ll << 31), "render (gather fused): %lld samples in one pass (limit 2^31 - 1)", M);
    if (int rc = check_cl_sources(rp, "render (gather fused)")) return rc;
    ucnerf_mlp_params m;
    memset(&m, 0, sizeof(m));
    m.cfg = rp->cfg; m.m = (int)M; m.S = rp->S; m.max_blocks = rp->max_blocks; m.dirs = dirs; m.wstream = rp->wstream; m.raw = raw;
    FusedGather f;
    memset(&f, 0, sizeof(f));
    f.S = rp->S; f.V = rp->cfg.n_src; f.H = rp->H; f.W = rp->W;
    for (int k = 0; k < 3; ++k) {
        f.vol_d[k] = rp->vol_d[k]; f.vol_h[k] = rp->vol_h[k]; f.vol_w[k] = rp->vol_w[k];
        f.vol[k] = reinterpret_cast<const char*>(rp->cl.vol[k]);
    }
    f.s16 = rp->cl.bf16 ? 1 : 0;
    f.feat = reinterpret_cast<const char*>(rp->cl.img_feat); f.col = reinterpret_cast<const char*>(rp->cl.imgs);
    f.col_px = rp->cl.bf16 ? 8u : 4u * (unsigned)rp->cl.rgb_stride;
    f.conf = rp->conf; f.rays_o = rp->rays_o; f.rays_d = rp->rays_d; f.z = rp->z; f.near_far = rp->near_far;
    f.near = rp->near; f.far = rp->far;
    memcpy(f.w2c_ref, rp->w2c_ref, sizeof(f.w2c_ref));
    memcpy(f.K_ref, rp->K_ref, sizeof(f.K_ref));
    f.w2cs = rp->w2cs; f.Ks = rp->intrinsics;
    f.pts_in = rp->pts_in; f.ndc_in[0] = rp->ndc1_in; f.ndc_in[1] = rp->ndc2_in; f.ndc_in[2] = rp->ndc3_in; f.ndc_enc = rp->ndc_in;
    if (rp->gen_rays) {          // ABI v4: the launch generates rays and stratified depths itself (validated by ucnerf_render_fused_fwd)
        const ucnerf_ray_gen_params* gr = rp->gen_rays;
        const ucnerf_sample_stratified_params* gs = rp->gen_depths;
        f.gen_xs = gr->xs; f.gen_ys = gr->ys; f.gen_noise = gs->perturb > 0.f ? gs->noise : nullptr;
        f.gen_K[0] = gr->K[0]; f.gen_K[1] = gr->K[2]; f.gen_K[2] = gr->K[4]; f.gen_K[3] = gr->K[5];
        memcpy(f.gen_R, gr->c2w, sizeof(f.gen_R));
        memcpy(f.gen_Q, gr->w2c_dir, sizeof(f.gen_Q));
        f.gen_perturb = gs->perturb; f.gen_lindisp = gs->lindisp;
        f.gen_rays_d = gr->rays_d; f.gen_z = gs->z; f.gen_angle = gr->angle;      // (`dirs` is not read: every lane derives its ray's feature itself)
    }
    {   // magic for idx / S (gather_cl.hip)
        unsigned l = 1;
        while ((1u << l) < (unsigned)rp->S) ++l;
        f.div_m = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)rp->S - 1) / (unsigned)rp->S);
        f.div_sh = l - 1;
    }
    if (tail_c) {
        const int cus = device_cus();
        if (cus <= 0) return fail(UCNERF_EHIP, "mlp_fwd: no device");
        f.tail_rpb = cdiv(rp->n, cus); f.tail_tpr = cdiv(rp->S, 32); f.tail_resample = tail_s ? 1 : 0;
        f.tail_spb = f.tail_rpb * rp->S;
        if (tail_dir_out) { f.tail_dir_Q = rp->w2c_dir_dev; f.tail_dir_out = tail_dir_out; }      // (render.hip: the features are made in the blocks' prologues)
        f.tail_c = *tail_c;
        if (tail_s) f.tail_s = *tail_s;
    }
    return launch_bf16(&m, nullptr, st, &f);
}
#else
#if !UCNERF_OPERAND_FP16
int launch_mlp_fwd_bf16_plain_h16(const ucnerf_mlp_params* p, hipStream_t st);
#endif
int launch_mlp_fwd_bf16_plain(const ucnerf_mlp_params* p, hipStream_t st) {
    UCNERF_REQUIRE(p, "mlp_fwd: null params");
    UCNERF_REQUIRE(p->cfg.operand == 0 || p->cfg.operand == 1, "mlp_fwd: cfg.operand %d (0 = bf16 terms, 1 = fp16 terms)", p->cfg.operand);
    UCNERF_OPERAND_DISPATCH(p->cfg.operand, launch_mlp_fwd_bf16_plain_h16(p, st));
    return launch_bf16(p, nullptr, st);
}
#endif

#if UCNERF_BF16_BUILD_TERMS == 3
int64_t bf16_index_count(const ucnerf_mlp_config* cfg) {
    Bf16Layout B;
    if (!bf16_layout(cfg->n_src, &B)) return -1;
    return (int64_t)B.slots * (SLOT_BYTES / 2) + CONST_FLOATS;
}

int64_t bf16_stream_floats(const ucnerf_mlp_config* cfg) {
    Bf16Layout B;
    if (!bf16_layout(cfg->n_src, &B)) return -1;
    return B.total_bytes / 4;
}

// ONE launch for the whole stream from the flat parameter vector (round 5: the evaluation loop re-packs in every rendering() call -- two launches were
// 5 us of GPU time and two launches' host time per 1024-pixel chunk): blocks [0, nb16) convert the bf16 half-steps, the rest copy the fp32 constants
__global__ void pack_all_flat_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx, unsigned short* __restrict__ out16, int64_t n16,
                                     float* __restrict__ outc, int nc, int nb16) {
    if ((int)blockIdx.x < nb16) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n16) return;
        const int32_t k = idx[i];
        unsigned short r = 0;
        if (k >= 0) {
            const float w = flat[k & 0x3fffffff];
            const op16 hi = (op16)w;
            const op16 val = (k >> 30) ? (op16)(w - (float)hi) : hi;
            r = __builtin_bit_cast(unsigned short, val);
        }
        out16[i] = r;
    } else {
        const int i = ((int)blockIdx.x - nb16) * blockDim.x + threadIdx.x;
        if (i < nc) { const int32_t k = idx[n16 + i]; outc[i] = k >= 0 ? flat[k] : 0.f; }
    }
}

int launch_pack_bf16(const ucnerf_mlp_config* cfg, const float* flat, const int32_t* idx, float* out, hipStream_t st) {
    UCNERF_REQUIRE(cfg->operand == 0 || cfg->operand == 1, "mlp_pack: cfg.operand %d (0 = bf16 terms, 1 = fp16 terms)", cfg->operand);
    UCNERF_OPERAND_DISPATCH(cfg->operand, launch_pack_bf16_h16(cfg, flat, idx, out, st));
    Bf16Layout B;
    UCNERF_REQUIRE(bf16_layout(cfg->n_src, &B), "mlp_pack: n_src %d outside 1..8", cfg->n_src);
    const int64_t n16 = (int64_t)B.slots * (SLOT_BYTES / 2);
    const int nb16 = cdiv(n16, 256), nbc = cdiv(CONST_FLOATS, 256);
    hipLaunchKernelGGL(pack_all_flat_kernel, dim3(nb16 + nbc), dim3(256), 0, st, flat, idx, reinterpret_cast<unsigned short*>(out), n16,
                       reinterpret_cast<float*>(reinterpret_cast<char*>(out) + B.const_off_bytes), CONST_FLOATS, nb16);
    return check_launch("mlp_pack (bf16x3)");
}

// ONE launch for the whole stream (the drop-in re-packs in every no_grad call): blocks [0, nb16) convert the bf16 half-steps, the rest copy the fp32 constants
__global__ void pack_all_tab_kernel(ParamTable t, const int32_t* __restrict__ idx, unsigned short* __restrict__ out16, int64_t n16, float* __restrict__ outc, int nc, int nb16) {
    __shared__ ParamTableLds l;
    param_table_to_lds(t, &l);
    if ((int)blockIdx.x < nb16) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n16) return;
        const int32_t k = idx[i];
        unsigned short r = 0;
        if (k >= 0) {
            const float w = param_table_load(&l, k & 0x3fffffff);
            const op16 hi = (op16)w;
            const op16 val = (k >> 30) ? (op16)(w - (float)hi) : hi;
            r = __builtin_bit_cast(unsigned short, val);
        }
        out16[i] = r;
    } else {
        const int i = ((int)blockIdx.x - nb16) * blockDim.x + threadIdx.x;
        if (i < nc) { const int32_t k = idx[n16 + i]; outc[i] = k >= 0 ? param_table_load(&l, k) : 0.f; }
    }
}

int launch_pack_bf16_tab(const ucnerf_mlp_config* cfg, const ParamTable& t, const int32_t* idx, float* out, hipStream_t st) {
    UCNERF_REQUIRE(cfg->operand == 0 || cfg->operand == 1, "mlp_pack_tensors: cfg.operand %d (0 = bf16 terms, 1 = fp16 terms)", cfg->operand);
    UCNERF_OPERAND_DISPATCH(cfg->operand, launch_pack_bf16_tab_h16(cfg, t, idx, out, st));
    Bf16Layout B;
    UCNERF_REQUIRE(bf16_layout(cfg->n_src, &B), "mlp_pack: n_src %d outside 1..8", cfg->n_src);
    const int64_t n16 = (int64_t)B.slots * (SLOT_BYTES / 2);
    const int nb16 = cdiv(n16, 256), nbc = cdiv(CONST_FLOATS, 256);
    hipLaunchKernelGGL(pack_all_tab_kernel, dim3(nb16 + nbc), dim3(256), 0, st, t, idx, reinterpret_cast<unsigned short*>(out), n16,
                       reinterpret_cast<float*>(reinterpret_cast<char*>(out) + B.const_off_bytes), CONST_FLOATS, nb16);
    return check_launch("mlp_pack_tensors (bf16x3)");
}

#endif   // UCNERF_BF16_BUILD_TERMS == 3
#endif   // !UCNERF_BF16_BUILD_TAIL

}  // namespace ucnerf
