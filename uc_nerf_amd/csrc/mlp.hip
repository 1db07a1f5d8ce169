// K5+K6: positional encoding + the uncertainty-conditioned MLP, forward (network/models.py:138-184,
// network/renderer.py:78-106 of the reference), on v_mfma_f32_32x32x2_f32.
//
// One wave owns a tile of 32 samples for the whole network.  Activations never leave the wave's
// registers (layout: mlp_layout.h); weights stream from L2 as one float4 per lane per k-step.
// A 256-thread block = 4 independent waves; the grid is persistent (waves stride over tiles).
#include "common.h"
#include "mlp_layout.h"
#include "p24.h"
#include "sincos_cw.h"

#include <cstdlib>
#include <vector>

namespace ucnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef UCNERF_MLP_WAVES
#define UCNERF_MLP_WAVES 8
#endif
// cache policy of the feature loads: nt (streaming) for the tiled layout, where every 128-byte line is consumed by one load
// (-0.4 % per f32 step); the row-major layout (training) reads its lines four bytes at a time and needs them cached
// (nt there: 391 -> 481 us)
constexpr int FEAT_AUX_TILED = 2, FEAT_AUX_ROWS = 0;
constexpr int MLP_WAVES = UCNERF_MLP_WAVES;   // waves per block: 8 = two per SIMD (waves w and w+4 share one)

// ------------------------------------------------------------------------------------------------
// host: pack index
// ------------------------------------------------------------------------------------------------
// One GEMM section: k-steps [t][lane][nt] at `off`, bias block [h][nt][r] at `off_bias`.
// row_base[n] = flat offset of weight row n (rows >= 64 may belong to a second matrix), col_h{0,1}[t] = column
// consumed by lane-half h at k-step t (-1 = zero pad).
static void fill_section(int32_t* idx, int64_t off, int64_t off_bias, int64_t p_bias_lo, int64_t p_bias_hi,
                         const std::vector<int64_t>& row_base, const std::vector<int>& col_h0,
                         const std::vector<int>& col_h1) {
    for (int h = 0; h < 2; ++h)
        for (int nt = 0; nt < 4; ++nt)
            for (int r = 0; r < 16; ++r) {
                int n = acc_feature(nt, r, h);
                idx[off_bias + (h * 4 + nt) * 16 + r] = (int32_t)(p_bias_hi >= 0 && n >= 64 ? p_bias_hi + n - 64 : p_bias_lo + n);
            }
    for (size_t t = 0; t < col_h0.size(); ++t)
        for (int lane = 0; lane < 64; ++lane)
            for (int nt = 0; nt < 4; ++nt) {
                int n = 32 * nt + (lane & 31);
                int col = (lane >> 5) ? col_h1[t] : col_h0[t];
                idx[off + (int64_t)t * KSTEP_FLOATS + lane * 4 + nt] = col < 0 ? -1 : (int32_t)(row_base[n] + (int64_t)col);
            }
}

static int build_pack_index(const ucnerf_mlp_config* cfg, int32_t* idx) {
    MlpLayout L;
    if (!mlp_layout(cfg->n_src, &L)) return -1;
    const int v = L.v, W = MLP_W;
    for (int64_t i = 0; i < L.total; ++i) idx[i] = -1;
    auto rows = [&](int64_t base, int K) {
        std::vector<int64_t> rb(128);
        for (int n = 0; n < 128; ++n) rb[n] = base + (int64_t)n * K;
        return rb;
    };
    auto nat_cols = [&](int ks, int K, int h) {
        std::vector<int> c(ks);
        for (int t = 0; t < ks; ++t) c[t] = 2 * t + h < K ? 2 * t + h : -1;
        return c;
    };
    auto acc_cols = [&](int h, int base) {
        std::vector<int> c(KS_HID);
        for (int t = 0; t < KS_HID; ++t) c[t] = base + acc_feature(t >> 4, t & 15, h);
        return c;
    };
    auto pe_cols = [&](int ks, int nf, int h, int base) {
        std::vector<int> c(ks);
        for (int t = 0; t < ks; ++t) {
            int kind, a;
            pe_slot(t, h, nf, &kind, &a);
            int col = pe_column(kind, a, nf, cfg->pe_layout);
            c[t] = col < 0 ? -1 : base + col;
        }
        return c;
    };
    auto cat = [](std::vector<int> a, const std::vector<int>& b) { a.insert(a.end(), b.begin(), b.end()); return a; };
    auto bias_off = [&](int sec) { return L.off_const + (int64_t)sec * 128; };
    // bias nets
    fill_section(idx, L.off_sec[SEC_BD], bias_off(SEC_BD), L.p_bdb, -1, rows(L.p_bdw, 24 + 4 * v),
                 nat_cols(L.kd, 24 + 4 * v, 0), nat_cols(L.kd, 24 + 4 * v, 1));
    fill_section(idx, L.off_sec[SEC_BC], bias_off(SEC_BC), L.p_bcb, -1, rows(L.p_bcw, 8 * v), nat_cols(L.kc, 8 * v, 0),
                 nat_cols(L.kc, 8 * v, 1));
    // trunk
    fill_section(idx, L.off_sec[SEC_L0], bias_off(SEC_L0), L.p_lb[0], -1, rows(L.p_lw[0], MLP_PE_PTS),
                 pe_cols(KS_PE_PTS, 10, 0, 0), pe_cols(KS_PE_PTS, 10, 1, 0));
    for (int i = 1; i < 5; ++i)
        fill_section(idx, L.off_sec[SEC_L0 + i], bias_off(SEC_L0 + i), L.p_lb[i], -1, rows(L.p_lw[i], W), acc_cols(0, 0),
                     acc_cols(1, 0));
    fill_section(idx, L.off_sec[SEC_L0 + 5], bias_off(SEC_L0 + 5), L.p_lb[5], -1, rows(L.p_lw[5], W + MLP_PE_PTS),
                 cat(pe_cols(KS_PE_PTS, 10, 0, 0), acc_cols(0, MLP_PE_PTS)),
                 cat(pe_cols(KS_PE_PTS, 10, 1, 0), acc_cols(1, MLP_PE_PTS)));
    fill_section(idx, L.off_sec[SEC_FT], bias_off(SEC_FT), L.p_fb, -1, rows(L.p_fw, W), acc_cols(0, 0), acc_cols(1, 0));
    {   // views_linears (rows 0..63) stacked on view_confi_linears (rows 64..127); input [feature 128 | dir PE 27]
        std::vector<int64_t> rb(128);
        for (int n = 0; n < 64; ++n) {
            rb[n] = L.p_vw + (int64_t)n * (W + MLP_PE_DIR);
            rb[64 + n] = L.p_vcw + (int64_t)n * (W + MLP_PE_DIR);
        }
        fill_section(idx, L.off_sec[SEC_VC], bias_off(SEC_VC), L.p_vb, L.p_vcb, rb,
                     cat(acc_cols(0, 0), pe_cols(KS_PE_DIR, 4, 0, W)), cat(acc_cols(1, 0), pe_cols(KS_PE_DIR, 4, 1, W)));
    }
    // wrap: the ring prefetch of a tile's last k-steps lands on the first k-steps of the next tile
    for (int64_t i = 0; i < (int64_t)RING * KSTEP_FLOATS; ++i) idx[L.off_wrap + i] = idx[i];
    // heads: [h][nt][r][4] + 4 biases
    const int64_t off_hb = L.off_const + N_SEC * 128, off_ha = off_hb + 516;
    for (int h = 0; h < 2; ++h)
        for (int nt = 0; nt < 4; ++nt)
            for (int r = 0; r < 16; ++r) {
                int f = acc_feature(nt, r, h);
                int64_t o = ((h * 4 + nt) * 16 + r) * 4;
                for (int c = 0; c < 3; ++c) idx[off_hb + o + c] = (int32_t)(L.p_crw + c * W + f);   // confi_rgb_linear
                idx[off_hb + o + 3] = (int32_t)(L.p_a1w + f);                                         // alpha_linear_1
                if (f < 64) for (int c = 0; c < 3; ++c) idx[off_ha + o + c] = (int32_t)(L.p_rw + c * 64 + f);  // rgb_linear
                else idx[off_ha + o + 3] = (int32_t)(L.p_aw + f - 64);                                 // alpha_linear
            }
    for (int c = 0; c < 3; ++c) { idx[off_hb + 512 + c] = (int32_t)(L.p_crb + c); idx[off_ha + 512 + c] = (int32_t)(L.p_rb + c); }
    idx[off_hb + 512 + 3] = (int32_t)L.p_a1b;
    idx[off_ha + 512 + 3] = (int32_t)L.p_ab;
    return 0;
}

__global__ void pack_kernel(const float* __restrict__ flat, const int32_t* __restrict__ idx, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int32_t k = idx[i]; out[i] = k >= 0 ? flat[k] : 0.f; }
}

__global__ void pack_tab_kernel(ParamTable t, const int32_t* __restrict__ idx, float* __restrict__ out, int64_t n) {
    __shared__ ParamTableLds l;
    param_table_to_lds(t, &l);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int32_t k = idx[i]; out[i] = k >= 0 ? param_table_load(&l, k) : 0.f; }
}

__global__ void unpack_grad_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx, float* __restrict__ gflat, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int32_t k = idx[i]; if (k >= 0) atomicAdd(gflat + k, g[i]); }
}

// ------------------------------------------------------------------------------------------------
// device: forward
// ------------------------------------------------------------------------------------------------
struct MlpGeom {      // MlpLayout subset the kernel needs (32-bit is plenty: the stream is < 1 MB)
    int F, kd, kc, f_img, off_const;
    int pts_stride, dirs_stride, feat_stride, pe_layout;
    int stream_bytes;     // size of the packed weight stream
    unsigned feat_bytes;  // size of the feature buffer
    int stagger;          // start-up delay of waves 4..7 in units of s_sleep(127) (= 8128 cycles)
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The weight stream is walked strictly in order by every wave.  It is addressed as a buffer: the descriptor and
// the running k-step offset `soff` live in SGPRs, the per-lane part (lane*16 bytes) in one VGPR, so a load costs
// no VALU address arithmetic.  `ring` holds the next RING k-steps already in flight; sched_barrier(0) after each
// k-step keeps hipcc from hoisting a whole layer's loads (which spills) -- the ring IS the prefetch.
struct Stream {
    __amdgpu_buffer_rsrc_t rs;
    int voff;             // lane * 16
    int soff;             // byte offset of the current k-step
    f32x4 ring[RING];
};

__device__ __forceinline__ f32x4 load_kstep(const Stream& S, int rel) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(S.rs, S.voff, S.soff + rel * (KSTEP_FLOATS * 4), 0));
}

#define KSTEP(S, T, BVAL, ACC)                                   \
    {                                                            \
        const f32x4 a_ = (S).ring[(T) & (RING - 1)];             \
        (S).ring[(T) & (RING - 1)] = load_kstep((S), (T) + RING);\
        const float b_ = (BVAL);                                 \
        (ACC)[0] = MFMA(a_.x, b_, (ACC)[0]);                     \
        (ACC)[1] = MFMA(a_.y, b_, (ACC)[1]);                     \
        (ACC)[2] = MFMA(a_.z, b_, (ACC)[2]);                     \
        (ACC)[3] = MFMA(a_.w, b_, (ACC)[3]);                     \
        __builtin_amdgcn_sched_barrier(0);                       \
    }

// gathered features of this lane's sample, also through a buffer descriptor: element k at voff + k * step bytes
struct FeatSrc {
    __amdgpu_buffer_rsrc_t rs;
    int voff;
};

template <bool TILED>
__device__ __forceinline__ float load_feat(const FeatSrc& F, int soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(F.rs, F.voff, soff_bytes, TILED ? FEAT_AUX_TILED : FEAT_AUX_ROWS));
}

// accumulators <- bias block of section `sec` (LDS copy of the constants)
__device__ __forceinline__ void init_bias(const float* cst, int sec, int h, f32x16 (&acc)[4]) {
    const f32x4* b = reinterpret_cast<const f32x4*>(cst + sec * 128 + h * 64);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b[nt * 4 + q];
            acc[nt][4 * q] = v.x; acc[nt][4 * q + 1] = v.y; acc[nt][4 * q + 2] = v.z; acc[nt][4 * q + 3] = v.w;
        }
}

// 64 k-steps whose B operands are the registers of an accumulator-layout activation set
__device__ __forceinline__ void gemm_hidden(Stream& S, const f32x16 (&x)[4], f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < KS_HID; ++t) KSTEP(S, t, x[t >> 4][t & 15], acc)
    S.soff += KS_HID * KSTEP_FLOATS * 4;
}

// KS k-steps whose B operands sit in a per-lane register array (encodings)
template <int KS>
__device__ __forceinline__ void gemm_regs(Stream& S, const float (&x)[KS], f32x16 (&acc)[4]) {
    static_assert(KS % RING == 0, "sections are multiples of the ring depth");
#pragma unroll
    for (int t = 0; t < KS; ++t) KSTEP(S, t, x[t], acc)
    S.soff += KS * KSTEP_FLOATS * 4;
}

// KS k-steps whose B operands come from this wave's LDS stash (slot t at stash[t*64 + lane])
template <int KS>
__device__ __forceinline__ void gemm_stash(Stream& S, const float* stash, int lane, f32x16 (&acc)[4]) {
    float b[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) b[i] = stash[i * 64 + lane];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        const float bv = b[t & (RING - 1)];
        if (t + RING < KS) b[t & (RING - 1)] = stash[(t + RING) * 64 + lane];
        KSTEP(S, t, bv, acc)
    }
    S.soff += KS * KSTEP_FLOATS * 4;
}

// The gathered features of a section are fetched ALL AT ONCE (one HBM latency per section instead of one per ring
// refill: they are streamed, never cached) and early, so the latency hides under VALU work issued in between.
constexpr int KS_FEAT_MAX = 32;       // k-steps of a bias net at most: (24 + 4*8)/2 = 28, 8*8/2 = 32

template <bool TILED>
__device__ __forceinline__ void load_section_feats(const FeatSrc& F, int first, int step, int ks, float (&b)[KS_FEAT_MAX]) {
#pragma unroll
    for (int t = 0; t < KS_FEAT_MAX; ++t) b[t] = t < ks ? load_feat<TILED>(F, first + t * step) : 0.f;
}

// ks (runtime, multiple of RING, <= KS_FEAT_MAX) k-steps on prefetched feature operands
__device__ __forceinline__ void gemm_feats(Stream& S, const float (&b)[KS_FEAT_MAX], int ks, f32x16 (&acc)[4]) {
#pragma unroll
    for (int t0 = 0; t0 < KS_FEAT_MAX; t0 += RING) {
        if (t0 < ks) {
#pragma unroll
            for (int i = 0; i < RING; ++i) KSTEP(S, i, b[t0 + i], acc)
            S.soff += RING * KSTEP_FLOATS * 4;
        }
    }
}

// 4-wide heads on the VALU: out4 = sum_f x[f] * Wh[f][0..3] over this lane's 64 features (two packed FMAs per
// feature), then both lane halves
__device__ __forceinline__ f32x4 head4(const float* hd, int h, const f32x16 (&x)[4]) {
    const f32x4* w = reinterpret_cast<const f32x4*>(hd) + h * 64;
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 wv = w[nt * 16 + r];
            const f32x2 xv = {x[nt][r], x[nt][r]};
            s01 = __builtin_elementwise_fma(xv, f32x2{wv.x, wv.y}, s01);
            s23 = __builtin_elementwise_fma(xv, f32x2{wv.z, wv.w}, s23);
        }
    f32x4 s = {s01.x, s01.y, s23.x, s23.y};
    s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32);
    const f32x4 b = *reinterpret_cast<const f32x4*>(hd + 512);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    return s;
}

// Positional encoding of a 3-vector in the k-step order of mlp_layout.h (this lane-half's slots).
template <int NF, int KS>
__device__ __forceinline__ void encode(const float (&x)[3], int h, float (&pe)[KS]) {   // x: raw 3-vector
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

// Encoded-input mode: the encoding already sits in memory (reference column order); pick this lane-half's slots.
template <int NF, int KS>
__device__ __forceinline__ void load_encoded(const float* __restrict__ row, int h, int layout, float (&pe)[KS]) {
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        int k0, a0, k1, a1;
        pe_slot(t, 0, NF, &k0, &a0);
        pe_slot(t, 1, NF, &k1, &a1);
        const int c0 = layout == 0 ? pe_column(k0, a0, NF, 0) : pe_column(k0, a0, NF, 1);
        const int c1 = layout == 0 ? pe_column(k1, a1, NF, 0) : pe_column(k1, a1, NF, 1);
        const int c = h ? c1 : c0;
        pe[t] = c >= 0 ? row[c] : 0.f;
    }
}

// Wave priority: the short VALU phases (encodings, epilogues, heads) run at high priority so that the partner
// wave's back-to-back MFMAs (which otherwise hold the SIMD's vector issue almost continuously: measured ~100
// cycles per VALU instruction, 35-65k cycles per phase) cannot stretch them; GEMM sections run at priority 0.
#ifndef UCNERF_MLP_SAVE_NT
#define UCNERF_MLP_SAVE_NT 1     // the kept 24-bit sets leave with non-temporal stores (whole 64-byte sectors per row here): the gradient chain reads them
                                 // back 15-20 us faster per 131 k samples than sets left dirty in the caches (this launch +7 us); 0 = plain stores (A/B)
#endif
#ifndef UCNERF_MLP_PRIO
#define UCNERF_MLP_PRIO 3
#endif
#define PRIO_VALU() __builtin_amdgcn_s_setprio(UCNERF_MLP_PRIO)
#define PRIO_GEMM() __builtin_amdgcn_s_setprio(0)

// On gfx950 the fp32 MFMA runs on the vector ALU itself (measured: VALU fillers cost their full issue time, a
// partner wave's VALU starves while MFMAs stream), so every VALU instruction is paid in matrix throughput.
// Packed fp32 math halves the instruction count of the element-wise stages.
#define EPILOGUE_RELU_MOD(DST, ACC, MOD)                                               \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                   \
    _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                \
        const f32x2 m_ = f32x2{(ACC)[nt][r], (ACC)[nt][r + 1]} * f32x2{(MOD)[nt][r], (MOD)[nt][r + 1]}; \
        (DST)[nt][r] = fmaxf(m_.x, 0.f); (DST)[nt][r + 1] = fmaxf(m_.y, 0.f);          \
    }

// accumulator-layout activation set -> row-major [m,128] (lane = sample, 4 consecutive features per piece).
// fp32 (P24 = false): 16-byte pieces straight from the registers.
// P24: the 24-bit format of p24.h (12-byte pieces, 384-byte rows) the gradient chain and the weight-gradient launch read.  Stored straight from
// the accumulator layout a store instruction writes 24 bytes into each of 32 rows; instead the wave TRANSPOSES half a set at a time (64
// columns = 192 bytes of each of its 32 rows) through a 6.5-KB LDS buffer of its own and writes it out as six 1-KB instructions that walk the
// rows' bytes in order (runs of 192 contiguous bytes) -- the stores of a tiled set layout gained the forward 54 us per 131 k samples, but that
// layout is hostile to the weight-gradient launch (profiles/r03_experiments.md); this gets the store side of it with rows kept.
constexpr int XPOSE_ROW = 192 + 16;                   // LDS row stride of the transpose buffer (the pad spreads the rows over the banks)
constexpr int XPOSE_BYTES = 32 * XPOSE_ROW;
template <bool P24>
__device__ __forceinline__ void save_rows(float* buf, int s, int h, bool valid, const f32x16 (&x)[4], char* xbuf = nullptr, int lane = 0, int tile = 0, int m = 0) {
    if (P24) {
        asm volatile("" : "+v"(lane));                // (the chunk addresses below do not depend on the tile: unlaundered they are hoisted out of the tile loop -- 24 registers for the whole kernel)
        const int j = lane & 31;
        char* const gtile = reinterpret_cast<char*>(buf) + (size_t)tile * 32 * P24_ROW_BYTES;
        const int rows = m - tile * 32;               // valid rows of this tile (>= 32: all)
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
                __builtin_amdgcn_sched_barrier(0);                    // (one chunk at a time: four registers, not twenty-four)
                const int c = k * 64 + lane;                          // 16-byte chunk of the half set: row c / 12, chunk c % 12 of its 192 bytes
                const int row = (c * 2731) >> 15, within = c - 12 * row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(xbuf + row * XPOSE_ROW + 16 * within);
#if UCNERF_MLP_SAVE_NT
                if (row < rows) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(gtile + row * P24_ROW_BYTES + 192 * half + 16 * within));
#else
                if (row < rows) *reinterpret_cast<f32x4*>(gtile + row * P24_ROW_BYTES + 192 * half + 16 * within) = v;
#endif
            }
        }
        return;
    }
    if (!valid) return;
    f32x4* row = reinterpret_cast<f32x4*>(buf + (size_t)s * 128);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {x[nt][4 * q], x[nt][4 * q + 1], x[nt][4 * q + 2], x[nt][4 * q + 3]};
            __builtin_nontemporal_store(v, &row[8 * nt + 2 * q + h]);      // (streaming hint: 670 MB per 131 k samples, read back a pass later; 411 -> 391 us)
        }
}

// NSRC > 0 fixes the number of source views at compile time: the two bias-net sections then have constant trip
// counts and the whole tile body is straight-line code.  (With runtime counts hipcc wraps the k-steps in uniform
// branches and drains the prefetch ring -- vmcnt(3),(2),(1),(0) -- at every join: ~12 full L2 latencies per tile.)
template <bool TILED, int SAVE, bool ENC, int NSRC>       // SAVE: 0 inference, 1 training forward keeping fp32 activation sets, 2 keeping 24-bit sets
__global__ void __launch_bounds__(64 * MLP_WAVES, MLP_WAVES / 4) mlp_fwd_kernel(ucnerf_mlp_params p, MlpGeom g, int n_tiles, MlpSaved sv) {
    constexpr int KD_STATIC = ((24 + 4 * NSRC) / 2 + RING - 1) / RING * RING, KC_STATIC = (4 * NSRC + RING - 1) / RING * RING;
    const int kd = NSRC ? KD_STATIC : g.kd, kc = NSRC ? KC_STATIC : g.kc;
    __shared__ __attribute__((aligned(16))) float cst[CONST_FLOATS];
    __shared__ __attribute__((aligned(16))) float pe_stash[MLP_WAVES][KS_PE_PTS * 64];
    __shared__ __attribute__((aligned(16))) char xpose[SAVE == 2 ? MLP_WAVES : 1][SAVE == 2 ? XPOSE_BYTES : 16];      // per-wave transpose buffer of save_rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* __restrict__ ws = p.wstream;
    const int n_waves = gridDim.x * MLP_WAVES;
    // workgroups go to the eight XCDs round-robin: block b counts as logical block (b % 8) * (blocks / 8) + b / 8, so that consecutive tiles
    // (pieces of one ray) stay behind one XCD's L2
    const int lblock = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    float* stash = pe_stash[wave];

    for (int i = threadIdx.x; i < CONST_FLOATS; i += 64 * MLP_WAVES) cst[i] = ws[g.off_const + i];
    __syncthreads();
    const float* hb = cst + N_SEC * 128;
    const float* ha = hb + 516;

    // Waves w and w+4 share a SIMD and run the same program; started together they reach their VALU phases
    // (encodings, epilogues, heads) together and the matrix pipe idles.  Delay the second half once, by about
    // half a tile, so one wave's VALU work falls under its partner's MFMAs.
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);

    Stream S;
    S.rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, g.stream_bytes, 0x00020000);
    S.voff = lane * 16;
    S.soff = 0;
#pragma unroll
    for (int i = 0; i < RING; ++i) S.ring[i] = load_kstep(S, i);   // k-steps 0..RING-1
    FeatSrc FS;
    FS.rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.feats, 0, g.feat_bytes, 0x00020000);

    // raw 3-vector of this lane's sample in the first tile; later tiles are prefetched one tile ahead
    float px[3] = {0.f, 0.f, 0.f};
    {
        const int t0 = wave * (int)gridDim.x + lblock;               // tiles are dealt wave-major (see the tile loop)
        if (!ENC && t0 < n_tiles) {
            const int s0 = t0 * 32 + j < p.m ? t0 * 32 + j : p.m - 1;
            const float* prow = p.pts + (size_t)s0 * g.pts_stride;
            px[0] = prow[0]; px[1] = prow[1]; px[2] = prow[2];
        }
    }

    // Wave-major inside a round of n_waves tiles: a partly filled last round keeps the same number of waves busy in every block (one per
    // SIMD while it is at most half full) instead of running some blocks with two waves per SIMD and leaving others empty.
    for (int tile = wave * (int)gridDim.x + lblock; tile < n_tiles; tile += n_waves) {
        PRIO_VALU();
        S.soff = 0;
        const int s_raw = tile * 32 + j;
        const int s = s_raw < p.m ? s_raw : p.m - 1;
        const bool valid = s_raw < p.m;
        // feature f of sample s lives fstride floats after feature f-1, starting at FS.voff bytes
        int fstride;
        if (TILED) { FS.voff = (int)((((size_t)(s >> 5) * g.F * 32) + (s & 31)) * 4); fstride = 32; }
        else { FS.voff = (int)((size_t)s * g.feat_stride * 4); fstride = 1; }

        f32x16 bd[4], hin[4], acc[4];
        float fsec[KS_FEAT_MAX];
        float pn[3] = {0.f, 0.f, 0.f};

        // ---- (1) all operands of the depth-bias net + the confidence, in flight before any arithmetic
        load_section_feats<TILED>(FS, h * fstride * 4, 2 * fstride * 4, kd, fsec);
        const float conf = load_feat<TILED>(FS, (g.F - 1) * fstride * 4);
        const size_t ray = p.dirs_per_sample ? (size_t)s : (size_t)(s / p.S);

        // ---- (2) point encoding (from registers), stashed in LDS for the skip connection
        float pe[KS_PE_PTS];
        if (ENC) {
            load_encoded<10, KS_PE_PTS>(p.pts + (size_t)s * g.pts_stride, h, g.pe_layout, pe);
        } else {
            encode<10, KS_PE_PTS>(px, h, pe);
            // (3) next tile's point (clamped, so the load is unconditional): consumed only at the end of this tile
            int sn = (tile + n_waves) * 32 + j;
            sn = sn < p.m ? sn : p.m - 1;
            const float* prow = p.pts + (size_t)sn * g.pts_stride;
            pn[0] = prow[0]; pn[1] = prow[1]; pn[2] = prow[2];
        }
#pragma unroll
        for (int t = 0; t < KS_PE_PTS; ++t) stash[t * 64 + lane] = pe[t];

        // ---- depth-bias net: bd = W_d [volume feats | colours+masks] + b      (models.py:150)
        init_bias(cst, SEC_BD, h, bd);
        PRIO_GEMM(); gemm_feats(S, fsec, kd, bd); PRIO_VALU();
        if (SAVE) save_rows<SAVE == 2>(sv.bd, s, h, valid, bd, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);
        const float u = 1.f - conf, omu = 1.f - u;          // models.py:149,177-178 (consumed at the very end)

        // ---- layer 0
        init_bias(cst, SEC_L0, h, acc);
        PRIO_GEMM(); gemm_regs<KS_PE_PTS>(S, pe, acc); PRIO_VALU();
        EPILOGUE_RELU_MOD(hin, acc, bd)
        if (SAVE) save_rows<SAVE == 2>(sv.h[0], s, h, valid, hin, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);

        // ---- layers 1..4                                                        (models.py:153-155)
#pragma unroll 1
        for (int l = 1; l < 5; ++l) {
            init_bias(cst, SEC_L0 + l, h, acc);
            PRIO_GEMM(); gemm_hidden(S, hin, acc); PRIO_VALU();
            EPILOGUE_RELU_MOD(hin, acc, bd)
            if (SAVE) save_rows<SAVE == 2>(sv.h[l], s, h, valid, hin, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);
        }

        // ---- layer 5 on [pe | h]                                                (models.py:156-157)
        init_bias(cst, SEC_L0 + 5, h, acc);
        PRIO_GEMM(); gemm_stash<KS_PE_PTS>(S, stash, lane, acc); PRIO_VALU();
        PRIO_GEMM(); gemm_hidden(S, hin, acc); PRIO_VALU();
        EPILOGUE_RELU_MOD(hin, acc, bd)
        // operands of the confidence-bias net: issued now (b_d's registers are free), they land during the base heads
        load_section_feats<TILED>(FS, (g.f_img + h) * fstride * 4, 2 * fstride * 4, kc, fsec);
        if (SAVE) save_rows<SAVE == 2>(sv.h[5], s, h, valid, hin, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);

        // ---- base heads: confi_rgb_linear, alpha_linear_1                       (models.py:161-162)
        const f32x4 base = head4(hb, h, hin);

        // ---- confidence-bias net, feature_linear(h * b_c)                       (models.py:151,164)
        init_bias(cst, SEC_BC, h, bd);
        PRIO_GEMM(); gemm_feats(S, fsec, kc, bd); PRIO_VALU();
        if (SAVE) save_rows<SAVE == 2>(sv.bc, s, h, valid, bd, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 m_ = f32x2{hin[nt][r], hin[nt][r + 1]} * f32x2{bd[nt][r], bd[nt][r + 1]};
                hin[nt][r] = m_.x; hin[nt][r + 1] = m_.y;
            }
        // view direction of this sample's ray: needed after feature_linear's 64 k-steps
        const float* drow = p.dirs + ray * g.dirs_stride;
        float dv[3] = {0.f, 0.f, 0.f};
        if (!ENC) { dv[0] = drow[0]; dv[1] = drow[1]; dv[2] = drow[2]; }
        init_bias(cst, SEC_FT, h, acc);
        PRIO_GEMM(); gemm_hidden(S, hin, acc); PRIO_VALU();
        if (SAVE) save_rows<SAVE == 2>(sv.ft, s, h, valid, acc, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);

        // ---- views_linears | view_confi_linears on [feature | dir encoding], relu   (models.py:166-173)
        init_bias(cst, SEC_VC, h, hin);
        PRIO_GEMM(); gemm_hidden(S, acc, hin); PRIO_VALU();
        {
            float pd[KS_PE_DIR];
            if (ENC) load_encoded<4, KS_PE_DIR>(drow, h, g.pe_layout, pd);
            else encode<4, KS_PE_DIR>(dv, h, pd);
            PRIO_GEMM(); gemm_regs<KS_PE_DIR>(S, pd, hin); PRIO_VALU();
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) hin[nt][r] = fmaxf(hin[nt][r], 0.f);
        if (SAVE) save_rows<SAVE == 2>(sv.vc, s, h, valid, hin, xpose[SAVE == 2 ? wave : 0], lane, tile, p.m);

        // ---- adapt heads (rgb_linear on rows 0..63, alpha_linear on rows 64..127), uncertainty blend
        const f32x4 adapt = head4(ha, h, hin);
        f32x4 out;
        out.x = 1.f / (1.f + expf(-(base.x * omu + adapt.x * u)));
        out.y = 1.f / (1.f + expf(-(base.y * omu + adapt.y * u)));
        out.z = 1.f / (1.f + expf(-(base.z * omu + adapt.z * u)));
        out.w = fmaxf(adapt.w * omu + base.w * u, 0.f);
        if (h == 0 && valid) reinterpret_cast<f32x4*>(p.raw)[s_raw] = out;
        px[0] = pn[0]; px[1] = pn[1]; px[2] = pn[2];
    }
}

static MlpGeom geom_of(const MlpLayout& L) {
    MlpGeom g;
    g.F = L.F; g.kd = L.kd; g.kc = L.kc; g.f_img = 24 + 4 * L.v; g.off_const = (int)L.off_const;
    g.pts_stride = g.dirs_stride = 3; g.feat_stride = L.F; g.pe_layout = 0; g.stream_bytes = 0; g.feat_bytes = 0; g.stagger = 0;
    return g;
}

// Launches the forward; `save` (optional) receives the activations the backward needs.
int launch_mlp_fwd(const ucnerf_mlp_params* p, const MlpSaved* save, hipStream_t st) {
    UCNERF_REQUIRE(p, "mlp_fwd: null params");
    if (p->m == 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->pts && p->dirs && p->feats && p->wstream && p->raw, "mlp_fwd: null pointer");
    MlpLayout L;
    UCNERF_REQUIRE(mlp_layout(p->cfg.n_src, &L), "mlp_fwd: n_src %d outside 1..8", p->cfg.n_src);
    UCNERF_REQUIRE(p->m >= 0, "mlp_fwd: m < 0");
    UCNERF_REQUIRE(p->dirs_per_sample || p->S > 0, "mlp_fwd: S must be > 0 when dirs are per ray");
    UCNERF_REQUIRE(((uintptr_t)p->wstream & 15) == 0 && ((uintptr_t)p->raw & 15) == 0, "mlp_fwd: wstream/raw must be 16-byte aligned");
    const int n_tiles = cdiv(p->m, 32);
    int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_fwd: no device");
    int blocks = cdiv(n_tiles, MLP_WAVES);
    int cap = p->max_blocks > 0 ? p->max_blocks : cus;          // one 8-wave block per CU (two waves per SIMD)
    if (MLP_WAVES == 8 && blocks < cap) { const int spread = cdiv(n_tiles, 4); blocks = spread < cap ? spread : cap; }      // fewer tiles than wave slots: every CU, one wave per SIMD
    if (blocks > cap) blocks = cap;
    MlpGeom g = geom_of(L);
    g.stream_bytes = (int)(L.total * 4);
    {
        const size_t fstride = p->feat_stride ? (size_t)p->feat_stride : (size_t)L.F;
        const size_t fbytes = p->feats_tiled ? (size_t)n_tiles * 32 * L.F * 4 : ((size_t)(p->m - 1) * fstride + L.F) * 4;
        UCNERF_REQUIRE(fbytes < (1ull << 31), "mlp_fwd: feature buffer of %zu bytes exceeds the 2 GiB buffer-addressing range; split the batch", fbytes);
        g.feat_bytes = (unsigned)fbytes;
    }
    {   // half a tile by default: 2304 MFMAs * 64 cycles shared by two waves ~ 295k cycles per tile
        static const int stagger = [] { const char* e = getenv("UCNERF_MLP_STAGGER"); return e ? atoi(e) : 0; }();   // (read once, thread-safe; measured on MI355X: 0 is best)
        g.stagger = n_tiles > blocks * 4 ? stagger : 0;          // nothing to hide when waves 4..7 have no partner work
    }
    UCNERF_REQUIRE(p->cfg.pe_layout == 0 || p->cfg.pe_layout == 1, "mlp_fwd: pe_layout %d", p->cfg.pe_layout);
    UCNERF_REQUIRE(!p->encoded || (p->dirs_per_sample && !p->feats_tiled), "mlp_fwd: encoded inputs need per-sample dirs and row-major feats");
    UCNERF_REQUIRE(p->pts_stride >= 0 && p->dirs_stride >= 0 && p->feat_stride >= 0, "mlp_fwd: negative stride");
    g.pe_layout = p->cfg.pe_layout;
    g.pts_stride = p->pts_stride ? p->pts_stride : (p->encoded ? MLP_PE_PTS : 3);
    g.dirs_stride = p->dirs_stride ? p->dirs_stride : (p->encoded ? MLP_PE_DIR : 3);
    g.feat_stride = p->feat_stride ? p->feat_stride : L.F;
    MlpSaved sv;
    memset(&sv, 0, sizeof(sv));
    if (save) sv = *save;
    dim3 grid(blocks), block(64 * MLP_WAVES);
#define LAUNCH(T, SV, E, N) hipLaunchKernelGGL((mlp_fwd_kernel<T, SV, E, N>), grid, block, 0, st, *p, g, n_tiles, sv)
#define LAUNCH_V(T, SV) { if (L.v == 6) LAUNCH(T, SV, false, 6); else if (L.v == 3) LAUNCH(T, SV, false, 3); else LAUNCH(T, SV, false, 0); }
    if (p->encoded) { if (save && sv.p24) LAUNCH(false, 2, true, 0); else if (save) LAUNCH(false, 1, true, 0); else LAUNCH(false, 0, true, 0); }
    else if (save && sv.p24) { if (p->feats_tiled) LAUNCH_V(true, 2) else LAUNCH_V(false, 2) }
    else if (save) {
        UCNERF_REQUIRE(!p->feats_tiled, "mlp_fwd_train: fp32 activation sets serve the layer-by-layer backward, which reads row-major features");
        LAUNCH_V(false, 1)
    }
    else { if (p->feats_tiled) LAUNCH_V(true, 0) else LAUNCH_V(false, 0) }
#undef LAUNCH_V
#undef LAUNCH
    return check_launch("mlp_fwd");
}

// mlp_bf16.hip
int build_pack_index_bf16(const ucnerf_mlp_config* cfg, int32_t* idx);
int64_t bf16_index_count(const ucnerf_mlp_config* cfg);
int64_t bf16_stream_floats(const ucnerf_mlp_config* cfg);
int launch_pack_bf16(const ucnerf_mlp_config* cfg, const float* flat, const int32_t* idx, float* out, hipStream_t st);
int launch_pack_bf16_tab(const ucnerf_mlp_config* cfg, const ParamTable& t, const int32_t* idx, float* out, hipStream_t st);
int launch_mlp_fwd_bf16x3(const ucnerf_mlp_params* p, hipStream_t st);      // mlp_bf16.hip built with TERMS = 3
int launch_mlp_fwd_bf16_plain(const ucnerf_mlp_params* p, hipStream_t st);   // ... and with TERMS = 1

const char* build_flags_mlp_f32() { return "mlp_f32: " UCNERF_FLAG(UCNERF_MLP_WAVES) UCNERF_FLAG(UCNERF_MLP_PRIO) UCNERF_FLAG(UCNERF_MLP_RING) UCNERF_FLAG(UCNERF_MLP_SAVE_NT); }

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int64_t ucnerf_mlp_param_count(const ucnerf_mlp_config* cfg) {
    MlpLayout L;
    if (!cfg || !mlp_layout(cfg->n_src, &L)) return fail(UCNERF_EINVAL, "mlp: n_src must be in 1..8");
    return L.n_params;
}

int64_t ucnerf_mlp_stream_count(const ucnerf_mlp_config* cfg) {
    MlpLayout L;
    if (!cfg || !mlp_layout(cfg->n_src, &L)) return fail(UCNERF_EINVAL, "mlp: n_src must be in 1..8");
    if (cfg->precision >= 1 && cfg->precision <= 3) return bf16_stream_floats(cfg);
    if (cfg->precision != 0) return fail(UCNERF_EINVAL, "mlp: precision %d (0 = f32, 1 = bf16x3, 2 = bf16, 3 = bf16x3 with the gather fused)", cfg->precision);
    return L.total;
}

int64_t ucnerf_mlp_index_count(const ucnerf_mlp_config* cfg) {
    const int64_t n = ucnerf_mlp_stream_count(cfg);
    if (n < 0) return n;
    return cfg->precision != 0 ? bf16_index_count(cfg) : n;
}

int ucnerf_mlp_pack_index(const ucnerf_mlp_config* cfg, int32_t* idx_host) {
    UCNERF_REQUIRE(cfg && idx_host, "mlp_pack_index: null pointer");
    UCNERF_REQUIRE(cfg->pe_layout == 0 || cfg->pe_layout == 1, "mlp_pack_index: pe_layout %d", cfg->pe_layout);
    UCNERF_REQUIRE(cfg->precision >= 0 && cfg->precision <= 3, "mlp_pack_index: precision %d", cfg->precision);
    if (cfg->precision == 0) {
        UCNERF_REQUIRE(build_pack_index(cfg, idx_host) == 0, "mlp_pack_index: n_src %d outside 1..8", cfg->n_src);
        return UCNERF_OK;
    }
    UCNERF_REQUIRE(build_pack_index_bf16(cfg, idx_host) == 0, "mlp_pack_index: n_src %d outside 1..8", cfg->n_src);
    // the fp32 constants block (biases, head weights) is shared with the f32 layout
    MlpLayout L;
    mlp_layout(cfg->n_src, &L);
    std::vector<int32_t> tmp(L.total);
    build_pack_index(cfg, tmp.data());
    const int64_t n16 = bf16_index_count(cfg) - CONST_FLOATS;
    for (int i = 0; i < CONST_FLOATS; ++i) idx_host[n16 + i] = tmp[L.off_const + i];
    return UCNERF_OK;
}

int ucnerf_mlp_pack(const ucnerf_mlp_config* cfg, const float* flat, const int32_t* idx, float* out, void* stream) {
    UCNERF_REQUIRE(cfg && flat && idx && out, "mlp_pack: null pointer");
    if (cfg->precision != 0) return launch_pack_bf16(cfg, flat, idx, out, (hipStream_t)stream);
    const int64_t n = ucnerf_mlp_stream_count(cfg);
    UCNERF_REQUIRE(n > 0, "mlp_pack: bad config");
    hipLaunchKernelGGL(pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, flat, idx, out, n);
    return check_launch("mlp_pack");
}

int ucnerf_mlp_pack_tensors(const ucnerf_mlp_config* cfg, int32_t n_tensors, const void* const* tensor_ptrs_host, const int64_t* tensor_numel_host,
                            const int32_t* idx, float* out, void* stream) {
    UCNERF_REQUIRE(cfg && tensor_ptrs_host && tensor_numel_host && idx && out, "mlp_pack_tensors: null pointer");
    UCNERF_REQUIRE(n_tensors >= 1 && n_tensors <= MAX_PACK_TENSORS, "mlp_pack_tensors: %d tensors (1..%d)", n_tensors, MAX_PACK_TENSORS);
    const int64_t n_params = ucnerf_mlp_param_count(cfg);
    UCNERF_REQUIRE(n_params > 0, "mlp_pack_tensors: bad config");
    ParamTable t;
    memset(&t, 0, sizeof(t));
    t.n = n_tensors;
    int64_t off = 0;
    for (int j = 0; j < n_tensors; ++j) {
        UCNERF_REQUIRE(tensor_ptrs_host[j] && tensor_numel_host[j] > 0, "mlp_pack_tensors: tensor %d is empty", j);
        t.ptr[j] = (const float*)tensor_ptrs_host[j]; t.start[j] = (int)off;
        off += tensor_numel_host[j];
    }
    t.start[n_tensors] = (int)off;
    UCNERF_REQUIRE(off == n_params, "mlp_pack_tensors: the tensors hold %lld floats, the network has %lld parameters", (long long)off, (long long)n_params);
    if (cfg->precision != 0) return launch_pack_bf16_tab(cfg, t, idx, out, (hipStream_t)stream);
    const int64_t n = ucnerf_mlp_stream_count(cfg);
    UCNERF_REQUIRE(n > 0, "mlp_pack_tensors: bad config");
    hipLaunchKernelGGL(pack_tab_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, t, idx, out, n);
    return check_launch("mlp_pack_tensors");
}

int ucnerf_mlp_unpack_grad(const float* g, const int32_t* idx, float* gflat, int64_t n, void* stream) {
    UCNERF_REQUIRE(g && idx && gflat && n > 0, "mlp_unpack_grad: bad arguments");
    hipLaunchKernelGGL(unpack_grad_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, idx, gflat, n);
    return check_launch("mlp_unpack_grad");
}

int ucnerf_mlp_fwd(const ucnerf_mlp_params* p, void* stream) {
    if (p && (p->cfg.precision == 1 || p->cfg.precision == 3)) return launch_mlp_fwd_bf16x3(p, (hipStream_t)stream);      // (3: refused there with the reason)
    if (p && p->cfg.precision == 2) return launch_mlp_fwd_bf16_plain(p, (hipStream_t)stream);
    return launch_mlp_fwd(p, nullptr, (hipStream_t)stream);
}

}  // extern "C"
