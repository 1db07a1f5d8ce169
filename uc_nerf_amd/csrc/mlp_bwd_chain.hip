// K6 backward, data-gradient half: ONE persistent kernel walks the whole network backwards for a 32-sample tile with the
// gradient kept in accumulator registers from the output stage down to the two bias nets (autograd of network/models.py:138-184).
//
// It replaces head_bwd + 7 gemm_nn launches + trunk_top_bwd of the layer-by-layer backward (mlp_bwd.hip): those streamed g, h, b_d
// and g_bd through HBM once per layer (read-modify-write), 2 GB per 131 k samples.  Here a tile's gradient lives where the
// forward's activations live (mlp_bf16.hip): transposed, lane = sample, accumulator register r of row tile kt on lane half h =
// feature 32 kt + (r & 3) + 8 (r >> 2) + 4 h, so
//     G_in^T[k][sample] = W^T[k][n] * G_out^T[n][sample]
// takes W^T as the A operand (packed stream of bf16 (hi, lo) fragments, L2-resident) and the previous layer's accumulators,
// re-split into (hi, lo), as the B operand: three v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate (the forward's bf16x3
// scheme, 2^-16 relative).  Per layer the kernel reads the kept activation it needs for the relu mask and the b_d product (h_{l-1},
// once) and writes g_y (once) for the weight-gradient GEMMs: nothing else touches HBM.
//
// One wave per SIMD with 512 registers (the resident state is the accumulators, the B fragments and two activation sets; b_d and the
// running g_bd sum are parked in LDS).  The 512-KB stream of W^T fragments reaches the four waves of a block through ONE ring in LDS,
// filled by global_load_lds (each wave copies a quarter of every half-step, eight half-steps ahead) -- round 3's first version had every
// wave read the stream for itself from L2, three half-steps ahead: 2 GB of L2 traffic per launch and an L2 round trip (~1 000 cycles
// under that load) in front of most half-steps: 183 k cycles per tile against 25 k of MFMA work (SQ_WAIT_INST_ANY 0.61).
#include "common.h"
#include "mlp_layout.h"
#include "mlp_bwd_parts.h"

namespace ucnerf {

typedef float c_f32x16 __attribute__((ext_vector_type(16)));
typedef float c_f32x4 __attribute__((ext_vector_type(4)));
typedef float c_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 c_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned c_u32x4 __attribute__((ext_vector_type(4)));
#define CSB __builtin_amdgcn_sched_barrier(0)
#define CMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int BWD_HALF_STEPS = 128;            // VC 16 | FT 16 | BC 8 | L5 16 | L4 16 | L3 16 | L2 16 | L1 16 | BD 8
constexpr int BWD_HALF_BYTES = 4096;           // [t 0..1][hi, lo][64 lanes][8 bf16]
constexpr int BWD_HEAD_FLOATS = 8 * 128 + 8;   // head table [o][feature] (base rgb x3, base sigma, adapt rgb x3, adapt sigma) + 8 biases

// one transposed layer of the stream: rows = input features of the layer (padded to 32 * 2 * pairs), contraction over its outputs
struct BwdSec { int hs0, pairs, ld, col0, rows; long long base, base_hi; };
struct BwdPackArgs {
    BwdSec sec[9];
    const float* flat;
    unsigned short* out;           // [128][2048] bf16
    float* head;                   // [BWD_HEAD_FLOATS]
    long long p_crw, p_a1w, p_rw, p_aw, p_crb, p_a1b, p_rb, p_ab;
};

// feature held by element j of lane-half hh in k16-step q = (kt, s): accumulator register 8 s + j of row tile kt (as mlp_bf16.hip)
__host__ __device__ inline int c_hid_feature16(int kt, int s, int j, int hh) { return 32 * kt + (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * hh; }

// Thread = one 16-byte fragment of the stream.  A[row][n] = W[n][col0 + row]: lane l supplies row 32 (2 pair + t) + (l & 31),
// its eight elements the outputs n the B operand holds in that k16-step.
__global__ void __launch_bounds__(256) pack_bwd_kernel(BwdPackArgs a) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < BWD_HALF_STEPS * 256) {
        const int hs = f >> 8, t = (f >> 7) & 1, part = (f >> 6) & 1, lane = f & 63;
        int si = 0;
#pragma unroll
        for (int i = 1; i < 9; ++i) si = hs >= a.sec[i].hs0 ? i : si;
        const BwdSec sc = a.sec[si];
        const int lhs = hs - sc.hs0, pair = lhs >> 3, q = lhs & 7;
        const int row = 32 * (2 * pair + t) + (lane & 31);
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = c_hid_feature16(q >> 1, q & 1, e, lane >> 5);
            float w = 0.f;
            if (row < sc.rows) w = (sc.base_hi >= 0 && n >= 64) ? a.flat[sc.base_hi + (long long)(n - 64) * sc.ld + sc.col0 + row] : a.flat[sc.base + (long long)n * sc.ld + sc.col0 + row];
            const __bf16 hi = (__bf16)w;
            const __bf16 val = part ? (__bf16)(w - (float)hi) : hi;
            v[e] = __builtin_bit_cast(unsigned short, val);
        }
        c_u32x4 o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16)};
        reinterpret_cast<c_u32x4*>(a.out)[f] = o;
    }
    if (f < BWD_HEAD_FLOATS) {
        float w = 0.f;
        if (f < 1024) {
            const int o = f >> 7, k = f & 127;
            if (o < 3) w = a.flat[a.p_crw + o * 128 + k];
            else if (o == 3) w = a.flat[a.p_a1w + k];
            else if (o < 7) w = k < 64 ? a.flat[a.p_rw + (o - 4) * 64 + k] : 0.f;       // rgb_linear reads the views half of vc
            else w = k >= 64 ? a.flat[a.p_aw + k - 64] : 0.f;                           // alpha_linear the view_confi half
        } else {
            const int o = f - 1024;
            w = o < 3 ? a.flat[a.p_crb + o] : o == 3 ? a.flat[a.p_a1b] : o < 7 ? a.flat[a.p_rb + o - 4] : a.flat[a.p_ab];
        }
        a.head[f] = w;
    }
}

struct ChainArgs {
    int m, n_tiles, F, ldf, ldgf, n_mvs, n_img, feats_tiled;
    const float* raw;              // [m,4] forward output
    const float* g_raw;            // [m,4]
    const float* feats;            // conf = feats[s * ldf + F - 1], or feature F - 1 of the tile layout [m / 32][F][32] (feats_tiled)
    MlpSaved sv;
    const char* wstream;           // pack_bwd_kernel's stream
    const float* head;             // ... and head table
    float* G_vc; float* G_f; float* G_bc; float* gx; float* G_bd; float* G_y[6];      // [m,128] each, 24-bit (p24.h): operands of the weight-gradient GEMMs
    float* g_feats;                // [m, ldgf]
    float* g_base; float* g_adapt; float* g_sigma; // [m,4] each, 24-bit: (base rgb, base sigma), (adapt rgb, 0), (adapt sigma, 0, 0, 0)
};

__device__ __forceinline__ int c_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// a result that is only needed at the end of the tile would be SUNK there, its operands kept alive (spilled) meanwhile: pin it where it is written
__device__ __forceinline__ void c_pin(float& v) { asm volatile("" : "+v"(v)); }

struct CFrag { c_bf16x8 hi, lo; };
struct CAF { c_bf16x8 h0, l0, h1, l1; };

// (hi, lo) split of eight values: hi = truncated bf16, lo = bf16_rne(x - hi) 
// (the forward's split8, mlp_bf16.hip, rounds hi to nearest since round 5 -- the renders answer to an absolute 1e-4 bar; the gradients' bar is relative
//  and they keep the truncated hi, which measured 0.2 % faster in the forward kernel)
__device__ __forceinline__ CFrag c_split8(const float (&x)[8]) {
    c_u32x4 hi;
    CFrag f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[j]), b1 = __builtin_bit_cast(unsigned, x[j + 1]);
        hi[j >> 1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const c_f32x2 l = (c_f32x2){x[j], x[j + 1]} - (c_f32x2){__builtin_bit_cast(float, b0 & 0xffff0000u), __builtin_bit_cast(float, b1 & 0xffff0000u)};
        f.lo[j] = (__bf16)l.x;
        f.lo[j + 1] = (__bf16)l.y;
    }
    f.hi = __builtin_bit_cast(c_bf16x8, hi);
    return f;
}
__device__ __forceinline__ void c_split_tile(const c_f32x16& x, CFrag& f0, CFrag& f1) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = x[e];
    f0 = c_split8(t);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = x[8 + e];
    f1 = c_split8(t);
}

// ---- the weight ring: CHAIN_NB slots of one half-step (4 KB) each in LDS, shared by the block's four waves.  128 half-steps per tile
// = 0 mod CHAIN_NB, so slot and source of every copy are compile-time constants of the unrolled tile body and the ring runs on across tiles.
#ifndef UCNERF_CHAIN_WAVES
#define UCNERF_CHAIN_WAVES 4      // waves per block = per CU: 4 (one per SIMD, 512 registers) or 8 (two per SIMD, 256 registers: b_d is then re-read per layer
                                  //   instead of parked and the g_bd sum parked as 24-bit pieces -- experiment, profiles/r03_experiments.md)
#endif
constexpr int CHAIN_WAVES = UCNERF_CHAIN_WAVES;
constexpr int CHAIN_NB = 8;
struct CPipe {
    const char* gsrc;        // this lane's byte of half-step 0: stream + wave * 1024 + lane * 16 (laundered per section, see launder())
    unsigned dst;            // LDS byte address of this wave's quarter of slot 0
    const char* rd;          // this lane's 16 bytes of slot 0 (LDS)
};
// (issued from inline asm: the compiler treats a global_load_lds as an access to both memories and degrades every counted wait around it)
__device__ __forceinline__ void c_dma(const CPipe& P, int hs_src, int slot) {
    const char* src = P.gsrc + (size_t)hs_src * BWD_HALF_BYTES;
    const unsigned dst = P.dst + slot * BWD_HALF_BYTES;
    if (CHAIN_WAVES == 8) {                // eight waves: 512 bytes each -- the lower 32 lanes copy
        if ((threadIdx.x & 32) == 0) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
    } else
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ CAF c_ldaf(const CPipe& P, int slot) {
    const c_bf16x8* a = reinterpret_cast<const c_bf16x8*>(P.rd + slot * BWD_HALF_BYTES);
    CAF f;
    f.h0 = a[0]; f.l0 = a[64]; f.h1 = a[128]; f.l1 = a[192];
    return f;
}
// Start of half-step hs (its fragments are in `cur`): the next half-step's slot has landed for the whole block and everybody is done reading
// this half-step's slot, which is refilled with the half-step CHAIN_NB ahead; returns the next half-step's fragments.
// vmcnt: vector-memory operations retire in issue order.  Behind the awaited copy this wave has issued CHAIN_NB - 2 younger copies and -- in the
// first CHAIN_NB - 1 half-steps of a section -- the loads and stores of the epilogue in front of the section: YOUNGER = a lower bound of their
// number, known at compile time because every one of them is issued unconditionally by every wave (c_st_tile).  Waiting for all but the youngest
// CHAIN_NB - 2 operations regardless (the first build) made every epilogue's stores and the next layer's activation loads retire before the
// section behind them could start: 180 of the launch's 350 us (profiles/r03_experiments.md).
// (hs: a constant once the caller's loops are unrolled)
template <int YOUNGER>
__device__ __forceinline__ CAF c_advance(const CPipe& P, int hs) {
    static_assert(YOUNGER >= CHAIN_NB - 2 && YOUNGER <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(YOUNGER) : "memory");
    __builtin_amdgcn_s_barrier();
    c_dma(P, (hs + CHAIN_NB) % BWD_HALF_STEPS, hs % CHAIN_NB);
    return c_ldaf(P, (hs + 1) % CHAIN_NB);
}

// Every [m,128] set this kernel reads or writes is in the 24-bit tiled format of p24.h.
// row tile nt of this lane (rb = byte offset of its piece of column group h): registers 4 q + c <-> columns 32 nt + 8 q + 4 h + c = the piece of group 8 nt + 2 q + h
// (loaded as the pieces they are and unpacked where they are used: unpacked at the load, the byte permutes sit right behind it and with them
//  the wait for the data -- a full memory latency in front of the section the load was meant to fly under; 12 registers per row tile, not 16)
struct CRaw { unsigned d[12]; };
__device__ __forceinline__ CRaw c_ld_tile(const float* set, size_t rb, int nt) {
    const char* row = reinterpret_cast<const char*>(set) + rb;
    CRaw x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const P24Piece v = p24_load(row + P24_GROUP_BYTES * (8 * nt + 2 * q));
        x.d[3 * q] = v.d[0]; x.d[3 * q + 1] = v.d[1]; x.d[3 * q + 2] = v.d[2];
    }
    return x;
}
__device__ __forceinline__ c_f32x16 c_unpack(const CRaw& r) {
    c_f32x16 x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const P24Piece v = {{r.d[3 * q], r.d[3 * q + 1], r.d[3 * q + 2]}};
        float t0, t1, t2, t3;
        p24_unpack4(v, t0, t1, t2, t3);
        x[4 * q] = t0; x[4 * q + 1] = t1; x[4 * q + 2] = t2; x[4 * q + 3] = t3;
    }
    return x;
}
// (no predicate: the stores of a lane without a sample go to the spare row behind the set's last one -- `rb` then points there -- so that every
//  wave issues the same number of vector-memory operations, which c_advance's counted waits rely on)
__device__ __forceinline__ void c_st_tile(float* set, size_t rb, int nt, const c_f32x16& x) {
    char* row = reinterpret_cast<char*>(set) + rb;
#pragma unroll
    for (int q = 0; q < 4; ++q) p24_store(row + P24_GROUP_BYTES * (8 * nt + 2 * q), p24_pack4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]));
}

// PAIRS row-tile pairs x 8 k16-steps of one transposed layer, starting at half-step HS0 of the stream; EPI = vector-memory operations every wave
// has issued between the previous section and this one (see c_advance)
constexpr int c_younger(int epi) { return CHAIN_NB - 2 + epi > 63 ? 63 : CHAIN_NB - 2 + epi; }
template <int HS0, int PAIRS, int EPI>
__device__ __forceinline__ void c_section(const CPipe& P, CAF& cur, const CFrag (&B)[8], c_f32x16 (&acc)[4]) {
#pragma unroll
    for (int p = 0; p < PAIRS; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            CSB;              // (one half-step at a time)
            const CAF nxt = 8 * p + q < CHAIN_NB - 1 ? c_advance<c_younger(EPI)>(P, HS0 + 8 * p + q) : c_advance<CHAIN_NB - 2>(P, HS0 + 8 * p + q);
            acc[2 * p] = CMFMA(cur.h0, B[q].hi, acc[2 * p]);
            acc[2 * p] = CMFMA(cur.h0, B[q].lo, acc[2 * p]);
            acc[2 * p] = CMFMA(cur.l0, B[q].hi, acc[2 * p]);
            acc[2 * p + 1] = CMFMA(cur.h1, B[q].hi, acc[2 * p + 1]);
            acc[2 * p + 1] = CMFMA(cur.h1, B[q].lo, acc[2 * p + 1]);
            acc[2 * p + 1] = CMFMA(cur.l1, B[q].hi, acc[2 * p + 1]);
            cur = nxt;
        }
}

__device__ __forceinline__ void c_zero(c_f32x16 (&acc)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
}


constexpr int CHAIN_HEAD_PAD = (BWD_HEAD_FLOATS + 3) & ~3;
// LDS: [head table 4 KB][weight ring CHAIN_NB x 4 KB][per wave: b_d of its tile as 24-bit pieces (12 KB) | the running g_bd sum in fp32 (16 KB)]
constexpr int CHAIN_RING_OFF = CHAIN_HEAD_PAD * 4;
constexpr int CHAIN_PARK_OFF = CHAIN_RING_OFF + CHAIN_NB * BWD_HALF_BYTES;
constexpr int CHAIN_PARK_BD = CHAIN_WAVES == 4 ? 16 * 64 * 12 : 0, CHAIN_PARK_GBD = CHAIN_WAVES == 4 ? 16 * 64 * 16 : 16 * 64 * 12;
constexpr int CHAIN_LDS_BYTES = CHAIN_PARK_OFF + CHAIN_WAVES * (CHAIN_PARK_BD + CHAIN_PARK_GBD);      // 4 + 32 + 112 KB

__global__ void __launch_bounds__(64 * CHAIN_WAVES, CHAIN_WAVES / 4) mlp_bwd_chain_kernel(ChainArgs a) {
    // (b_d and the g_bd sum are parked in LDS rather than in 128 registers: with them resident the register allocator spilled ~200 values
    //  around every layer's MFMA section; b_d as the 24-bit pieces it was loaded as, which is what leaves room for the weight ring)
    extern __shared__ __attribute__((aligned(16))) float chain_lds[];
    float* hw = chain_lds;
    for (int i = threadIdx.x; i < BWD_HEAD_FLOATS; i += 64 * CHAIN_WAVES) hw[i] = a.head[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const lds = reinterpret_cast<char*>(chain_lds);
    char* const park_bd = lds + CHAIN_PARK_OFF + wave * (CHAIN_PARK_BD + CHAIN_PARK_GBD) + lane * 12;          // piece (nt, q) of this lane at + (4 nt + q) * 64 * 12
    c_f32x4* const park_gbd = reinterpret_cast<c_f32x4*>(lds + CHAIN_PARK_OFF + wave * (CHAIN_PARK_BD + CHAIN_PARK_GBD) + CHAIN_PARK_BD) + lane;
    auto ld_bd = [&](int nt, size_t ro_ = 0) {
        if (CHAIN_WAVES != 4) return c_unpack(c_ld_tile(a.sv.bd, ro_, nt));       // (not parked: re-read, L2 resident)
        c_f32x16 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const p24_u32x3 v = *reinterpret_cast<const p24_u32x3_a4*>(park_bd + (4 * nt + q) * 64 * 12);
            const P24Piece pc = {{v.x, v.y, v.z}};
            float t0, t1, t2, t3;
            p24_unpack4(pc, t0, t1, t2, t3);
            x[4 * q] = t0; x[4 * q + 1] = t1; x[4 * q + 2] = t2; x[4 * q + 3] = t3;
        }
        return x;
    };
    auto st_bd = [&](int nt, const CRaw& x) {          // (the pieces as they were loaded)
        if (CHAIN_WAVES != 4) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<p24_u32x3_a4*>(park_bd + (4 * nt + q) * 64 * 12) = (p24_u32x3){x.d[3 * q], x.d[3 * q + 1], x.d[3 * q + 2]};
    };
    char* const park_g24 = lds + CHAIN_PARK_OFF + wave * (CHAIN_PARK_BD + CHAIN_PARK_GBD) + CHAIN_PARK_BD + lane * 12;      // (8 waves: g_bd as 24-bit pieces)
    auto ld_gbd = [&](int nt) {
        c_f32x16 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (CHAIN_WAVES == 4) {
                const c_f32x4 v = park_gbd[(nt * 4 + q) * 64];
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            } else {
                const p24_u32x3 v = *reinterpret_cast<const p24_u32x3_a4*>(park_g24 + (4 * nt + q) * 64 * 12);
                const P24Piece pc = {{v.x, v.y, v.z}};
                float t0, t1, t2, t3;
                p24_unpack4(pc, t0, t1, t2, t3);
                x[4 * q] = t0; x[4 * q + 1] = t1; x[4 * q + 2] = t2; x[4 * q + 3] = t3;
            }
        }
        return x;
    };
    auto st_gbd = [&](int nt, const c_f32x16& x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (CHAIN_WAVES == 4) park_gbd[(nt * 4 + q) * 64] = (c_f32x4){x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
            else {
                const P24Piece pc = p24_pack4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
                *reinterpret_cast<p24_u32x3_a4*>(park_g24 + (4 * nt + q) * 64 * 12) = (p24_u32x3){pc.d[0], pc.d[1], pc.d[2]};
            }
        }
    };
    const int j = lane & 31, h = lane >> 5;
    // (the stream does not depend on the tile: unless its address is laundered per section, every one of its 2048 fragment registers is
    //  hoisted out of the tile loop and spilled; the OFFSET is laundered -- a pointer passed through an integer asm operand comes back
    //  as a flat pointer)
    CPipe P;
    constexpr int SHARE = BWD_HALF_BYTES / CHAIN_WAVES;          // bytes of a half-step this wave copies
    P.gsrc = a.wstream + wave * SHARE + (lane & (SHARE / 16 - 1)) * 16;
    P.dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + CHAIN_RING_OFF) + wave * SHARE;
    P.rd = lds + CHAIN_RING_OFF + lane * 16;
    auto launder = [&]() { unsigned o_ = (unsigned)(wave * SHARE + (lane & (SHARE / 16 - 1)) * 16); asm volatile("" : "+v"(o_)); P.gsrc = a.wstream + o_; };
#pragma unroll
    for (int i = 0; i < CHAIN_NB; ++i) c_dma(P, i, i);        // half-steps 0 .. CHAIN_NB - 1 of the first tile
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CHAIN_NB - 1) : "memory");      // slot 0 has landed ...
    __builtin_amdgcn_s_barrier();                                             // ... for every wave
    CAF cur = c_ldaf(P, 0);

    // Every wave of the block walks the same number of tiles (the ring is turned by all four together): a wave whose tile lies past the end
    // computes on the last sample's rows and stores nothing.
    const int rounds = (a.n_tiles + gridDim.x * CHAIN_WAVES - 1) / (gridDim.x * CHAIN_WAVES);
    for (int rnd = 0; rnd < rounds; ++rnd) {
        const int tile = (rnd * gridDim.x + blockIdx.x) * CHAIN_WAVES + wave;
        const int s_raw = tile * 32 + j;
        const bool valid = s_raw < a.m;
        const int s = valid ? s_raw : a.m - 1;
        const size_t ro = p24_offset((size_t)s, h, 32);        // this lane's piece of column group h of a [m,128] set, in bytes
        const size_t ro_st = p24_offset((size_t)(valid ? s_raw : a.m), h, 32);      // ... for stores: sample m is the spare row of every set written here

        // ---- output stage (models.py:177-178 backwards)
        const c_f32x4 raw = reinterpret_cast<const c_f32x4*>(a.raw)[s];
        c_f32x4 gr = reinterpret_cast<const c_f32x4*>(a.g_raw)[s];
        if (!valid) gr = (c_f32x4){0.f, 0.f, 0.f, 0.f};
        const float conf = a.feats_tiled ? a.feats[((size_t)(s >> 5) * a.F + a.F - 1) * 32 + (s & 31)] : a.feats[(size_t)s * a.ldf + a.F - 1];
        const float u = 1.f - conf, omu = 1.f - u;
        const float gp[4] = {gr.x * raw.x * (1.f - raw.x), gr.y * raw.y * (1.f - raw.y), gr.z * raw.z * (1.f - raw.z), raw.w > 0.f ? gr.w : 0.f};
        const float gb4[4] = {gp[0] * omu, gp[1] * omu, gp[2] * omu, gp[3] * u};
        const float ga4[4] = {gp[0] * u, gp[1] * u, gp[2] * u, gp[3] * omu};
        if (h == 0) {               // the G operands of the four head layers' weight gradients: three [m,4] sets (12-byte rows; row m = spare)
            const size_t r12 = (size_t)(valid ? s_raw : a.m) * 12;
            p24_store(reinterpret_cast<char*>(a.g_base) + r12, p24_pack4(gb4[0], gb4[1], gb4[2], gb4[3]));      // base rgb | base sigma
            p24_store(reinterpret_cast<char*>(a.g_adapt) + r12, p24_pack4(ga4[0], ga4[1], ga4[2], 0.f));       // adapt rgb
            p24_store(reinterpret_cast<char*>(a.g_sigma) + r12, p24_pack4(ga4[3], 0.f, 0.f, 0.f));             // adapt sigma
        }

        c_f32x16 acc[4];
        CRaw hn[4], hm[4], hx[4], bcr[4], bdr[4];
        CFrag X[8];
        float ad[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};      // this lane's share of the adapt / base head outputs

        // ---- adapt heads backwards + relu of [views | view_confi]: g_vc
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) hn[nt] = c_ld_tile(a.sv.vc, ro, nt);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const c_f32x16 hv_ = c_unpack(hn[nt]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * nt + 8 * q + 4 * c_opaque(h);       // (laundered: the table never changes, so its reads would be hoisted out of the tile loop and spilled)
                c_f32x4 w[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) w[o] = *reinterpret_cast<const c_f32x4*>(&hw[(4 + o) * 128 + f0]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = hv_[4 * q + c];
                    const float g = ga4[0] * w[0][c] + ga4[1] * w[1][c] + ga4[2] * w[2][c] + ga4[3] * w[3][c];
                    acc[nt][4 * q + c] = v > 0.f ? g : 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) ad[o] += v * w[o][c];
                }
                c_pin(ad[0]); c_pin(ad[1]); c_pin(ad[2]); c_pin(ad[3]);
                CSB;                                         // (a fence per group: the scheduler otherwise reads every head weight of the tile up front and spills them)
            }
            c_st_tile(a.G_vc, ro_st, nt, acc[nt]);
            c_split_tile(acc[nt], X[2 * nt], X[2 * nt + 1]);
            CSB;
        }

        // ---- [views | view_confi]^T: g_f (f = feature_linear's output, no activation)
        c_zero(acc);
        launder(); c_section<0, 2, 16 + 19>(P, cur, X, acc)          /* vc loads | three head-G stores + G_vc */;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            c_st_tile(a.G_f, ro_st, nt, acc[nt]);
            c_split_tile(acc[nt], X[2 * nt], X[2 * nt + 1]);
            CSB;
        }

        // ---- feature_linear^T: g_gx
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) hn[nt] = c_ld_tile(a.sv.h[5], ro, nt);       // (h5 lands under the section's MFMAs)
        c_zero(acc);
        launder(); c_section<16, 2, 32>(P, cur, X, acc)              /* G_f stores, h5 loads */;

        // ---- confidence-bias net^T first (its operand g_bc = g_gx * h5 dies before the trunk's state is born): gradient of the image
        //      features, columns n_mvs .. n_mvs + n_img of g_feats
        auto small_out = [&](const c_f32x16 (&r2)[4], int col0, int rows) {
            if (valid) {
                float* dst = a.g_feats + (size_t)s * a.ldgf + col0;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 32 * t + 8 * q + 4 * h;
                        if (f0 < rows) {                      // (rows is a multiple of 4: a group is in or out as a whole)
#pragma unroll
                            for (int c = 0; c < 4; ++c) dst[f0 + c] = r2[t][4 * q + c];
                        }
                    }
            }
        };
        {
            CFrag Z[8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                c_f32x16 gbc;
                const c_f32x16 h5_ = c_unpack(hn[nt]);
#pragma unroll
                for (int r = 0; r < 16; ++r) gbc[r] = acc[nt][r] * h5_[r];
                c_st_tile(a.G_bc, ro_st, nt, gbc);
                c_split_tile(gbc, Z[2 * nt], Z[2 * nt + 1]);
                CSB;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) hm[nt] = c_ld_tile(a.sv.h[4], ro, nt);      // for the first trunk epilogue, two sections away
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { bcr[nt] = c_ld_tile(a.sv.bc, ro, nt); bdr[nt] = c_ld_tile(a.sv.bd, ro, nt); }      // for the epilogue behind this section
            c_f32x16 a2[4];
            c_zero(a2);
            launder(); c_section<32, 1, 64>(P, cur, Z, a2)               /* G_bc stores, h4, b_c and b_d loads */;
            small_out(a2, a.n_mvs, a.n_img);
        }
        CSB;

        // ---- gx = h5 * bc (feature_linear's input), g_h5 = g_gx * bc + base heads^T g_base, relu / b_d backward of layer 5
        {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const c_f32x16 bct = c_unpack(bcr[nt]), bdt = c_unpack(bdr[nt]), h5_ = c_unpack(hn[nt]);
                c_f32x16 gxv, gy, gb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * nt + 8 * q + 4 * c_opaque(h);       // (laundered: the table never changes, so its reads would be hoisted out of the tile loop and spilled)
                    c_f32x4 w[4];
#pragma unroll
                    for (int o = 0; o < 4; ++o) w[o] = *reinterpret_cast<const c_f32x4*>(&hw[o * 128 + f0]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * q + c;
                        const float hv = h5_[r], gg = acc[nt][r];
                        gxv[r] = hv * bct[r];
#pragma unroll
                        for (int o = 0; o < 4; ++o) bs[o] += hv * w[o][c];
                        const float gh = gg * bct[r] + gb4[0] * w[0][c] + gb4[1] * w[1][c] + gb4[2] * w[2][c] + gb4[3] * w[3][c];
                        const float gpre = hv > 0.f ? gh : 0.f;
                        gb[r] = gpre * hv;
                        gy[r] = gpre * bdt[r];
                    }
                    c_pin(bs[0]); c_pin(bs[1]); c_pin(bs[2]); c_pin(bs[3]);
                    CSB;
                }
                st_bd(nt, bdr[nt]);
                st_gbd(nt, gb);
                c_st_tile(a.gx, ro_st, nt, gxv);
                c_st_tile(a.G_y[5], ro_st, nt, gy);
                c_split_tile(gy, X[2 * nt], X[2 * nt + 1]);
                CSB;
            }
        }

        // ---- trunk: layer 5 (hidden part) down to layer 1; the element-wise backward of layer l - 1 is the epilogue of layer l
        auto trunk_epi = [&](float* G_out, const CRaw (&hraw)[4]) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const c_f32x16 hcur_ = c_unpack(hraw[nt]);
                const c_f32x16 bdt = ld_bd(nt, ro);
                c_f32x16 gb = ld_gbd(nt), gy;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hv = hcur_[r];
                    const float gpre = hv > 0.f ? acc[nt][r] : 0.f;
                    gb[r] += gpre * hv;
                    gy[r] = gpre * bdt[r];
                }
                st_gbd(nt, gb);
                c_st_tile(G_out, ro_st, nt, gy);
                c_split_tile(gy, X[2 * nt], X[2 * nt + 1]);
                CSB;
            }
        };
        // The activation set a layer's epilogue needs is requested TWO layers early, into one of three register sets (hx takes the registers b_c and
        // b_d have just left), and BEHIND a section: vector-memory operations return in issue order, so a request in front of a section would hold
        // back the weight copies issued behind it.  One layer early (the first build) a set had one section (~4 us) to arrive: the launch spent
        // ~90 of its 330 us waiting for DRAM (scripts/prof_chain_variants.sh, UCNERF_CHAIN_EXP 64).
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) hx[nt] = c_ld_tile(a.sv.h[3], ro, nt);
#define UCNERF_CHAIN_LAYER(HS0, L, CUR, NXT, EPI)                                                          \
        {                                                                                                 \
            c_zero(acc);                                                                                  \
            launder(); c_section<HS0, 2, EPI>(P, cur, X, acc);                                               \
            if ((L) >= 3) {                                                                               \
                _Pragma("unroll")                                                                         \
                for (int nt = 0; nt < 4; ++nt) NXT[nt] = c_ld_tile(a.sv.h[(L) - 3], ro, nt);             \
            }                                                                                             \
            CSB;                                                                                          \
            trunk_epi(a.G_y[(L) - 1], CUR);                                                               \
        }
        UCNERF_CHAIN_LAYER(40, 5, hm, hn, 48)   // uses h4 (requested before the confidence-bias net's section), requests h2; in front: gx, G_y[5] stores, h3 loads
        UCNERF_CHAIN_LAYER(56, 4, hx, hm, 32)   // uses h3, requests h1; in front of each: a set's loads (while there are any), G_y stores
        UCNERF_CHAIN_LAYER(72, 3, hn, hx, 32)   // uses h2, requests h0
        UCNERF_CHAIN_LAYER(88, 2, hm, hn, 32)   // uses h1
        UCNERF_CHAIN_LAYER(104, 1, hx, hn, 16)  // uses h0
#undef UCNERF_CHAIN_LAYER

        // ---- depth-bias net: g_bd = sum_l g_pre_l * y_l with y_l = h_l / b_d on the active units; then its transposed layer
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const c_f32x16 bdt = ld_bd(nt, ro), gb = ld_gbd(nt);
            c_f32x16 g;
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = gb[r] != 0.f ? gb[r] / bdt[r] : 0.f;
            c_st_tile(a.G_bd, ro_st, nt, g);
            c_split_tile(g, X[2 * nt], X[2 * nt + 1]);
            CSB;
        }
        {
            c_f32x16 a2[4];
            c_zero(a2);
            launder(); c_section<120, 1, 32>(P, cur, X, a2)              /* G_y[0] and G_bd stores */;
            small_out(a2, 0, a.n_mvs);
        }

        // ---- d/d(confidence): u = 1 - confidence blends the two head pairs
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            ad[o] += __shfl_xor(ad[o], 32);
            bs[o] += __shfl_xor(bs[o], 32);
            ad[o] += hw[c_opaque(1024 + 4 + o)];
            bs[o] += hw[c_opaque(1024 + o)];
        }
        const float gu = gp[0] * (ad[0] - bs[0]) + gp[1] * (ad[1] - bs[1]) + gp[2] * (ad[2] - bs[2]) + gp[3] * (bs[3] - ad[3]);
        if (h == 0 && valid) a.g_feats[(size_t)s * a.ldgf + a.F - 1] = -gu;

    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the copies still in flight land before the block gives its LDS back
}

// ---- host side
size_t bwd_chain_stream_floats() { return (size_t)BWD_HALF_STEPS * BWD_HALF_BYTES / 4 + ((BWD_HEAD_FLOATS + 3) & ~3); }

int launch_pack_bwd(int n_src, const float* flat, float* stream_out, hipStream_t st) {
    MlpLayout L;
    UCNERF_REQUIRE(mlp_layout(n_src, &L), "mlp_bwd: n_src %d outside 1..8", n_src);
    const int v = L.v, KV = MLP_W + MLP_PE_DIR;
    BwdPackArgs a;
    memset(&a, 0, sizeof(a));
    int hs = 0, i = 0;
    auto sec = [&](int pairs, long long base, int ld, int col0, int rows, long long base_hi) {
        a.sec[i].hs0 = hs; a.sec[i].pairs = pairs; a.sec[i].ld = ld; a.sec[i].col0 = col0; a.sec[i].rows = rows; a.sec[i].base = base; a.sec[i].base_hi = base_hi;
        hs += 8 * pairs; ++i;
    };
    sec(2, L.p_vw, KV, 0, 128, L.p_vcw);                    // [views | view_confi]^T, feature part of their input
    sec(2, L.p_fw, 128, 0, 128, -1);                        // feature_linear^T
    sec(1, L.p_bcw, 8 * v, 0, 8 * v, -1);                   // confidence-bias net^T
    sec(2, L.p_lw[5], 128 + MLP_PE_PTS, MLP_PE_PTS, 128, -1);   // layer 5, hidden part of [pe | h4]
    for (int l = 4; l >= 1; --l) sec(2, L.p_lw[l], 128, 0, 128, -1);
    sec(1, L.p_bdw, 24 + 4 * v, 0, 24 + 4 * v, -1);         // depth-bias net^T
    if (hs != BWD_HALF_STEPS || i != 9) return fail(UCNERF_EINVAL, "mlp_bwd: stream schedule mismatch");
    a.flat = flat;
    a.out = reinterpret_cast<unsigned short*>(stream_out);
    a.head = stream_out + (size_t)BWD_HALF_STEPS * BWD_HALF_BYTES / 4;
    a.p_crw = L.p_crw; a.p_a1w = L.p_a1w; a.p_rw = L.p_rw; a.p_aw = L.p_aw; a.p_crb = L.p_crb; a.p_a1b = L.p_a1b; a.p_rb = L.p_rb; a.p_ab = L.p_ab;
    hipLaunchKernelGGL(pack_bwd_kernel, dim3(BWD_HALF_STEPS), dim3(256), 0, st, a);
    return check_launch("mlp_bwd pack");
}

// g_* / G_* operands: see ChainArgs.  `stream` = launch_pack_bwd's output.
int launch_mlp_bwd_chain(int n_src, int m, const float* raw, const float* g_raw, const float* feats, int ldf, int feats_tiled, const MlpSaved* sv, const float* stream,
                         float* G_vc, float* G_f, float* G_bc, float* gx, float* G_bd, float* const* G_y, float* g_feats, int ldgf, float* g_base,
                         float* g_adapt, float* g_sigma, hipStream_t st) {
    MlpLayout L;
    UCNERF_REQUIRE(mlp_layout(n_src, &L), "mlp_bwd: n_src %d outside 1..8", n_src);
    UCNERF_REQUIRE(m > 0 && m <= (1 << 24), "mlp_bwd: %d samples in one pass (limit 2^24)", m);
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.m = m; a.n_tiles = cdiv(m, 32); a.F = L.F; a.ldf = ldf; a.ldgf = ldgf; a.n_mvs = 24 + 4 * L.v; a.n_img = 8 * L.v;
    a.raw = raw; a.g_raw = g_raw; a.feats = feats; a.feats_tiled = feats_tiled; a.sv = *sv;
    a.wstream = reinterpret_cast<const char*>(stream);
    a.head = stream + (size_t)BWD_HALF_STEPS * BWD_HALF_BYTES / 4;
    a.G_vc = G_vc; a.G_f = G_f; a.G_bc = G_bc; a.gx = gx; a.G_bd = G_bd;
    for (int l = 0; l < 6; ++l) a.G_y[l] = G_y[l];
    a.g_feats = g_feats; a.g_base = g_base; a.g_adapt = g_adapt; a.g_sigma = g_sigma;
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_bwd: no device");
    int blocks = cdiv(a.n_tiles, CHAIN_WAVES);
    if (blocks > cus) blocks = cus;                        // one 4-wave block per CU (512 registers per wave), persistent
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_bwd_chain_kernel), CHAIN_LDS_BYTES, "mlp_bwd chain")) return rc;
    hipLaunchKernelGGL(mlp_bwd_chain_kernel, dim3(blocks), dim3(64 * CHAIN_WAVES), CHAIN_LDS_BYTES, st, a);
    return check_launch("mlp_bwd chain");
}

const char* build_flags_mlp_bwd_chain() { return "mlp_bwd_chain: " UCNERF_FLAG(UCNERF_CHAIN_WAVES); }

}  // namespace ucnerf
