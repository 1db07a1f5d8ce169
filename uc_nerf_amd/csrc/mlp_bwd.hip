// K6 backward: gradients of the UC-NeRF MLP w.r.t. its parameters and its gathered features
// (autograd of network/models.py:138-184; positions / view directions get no gradient).
//
// Layer-by-layer from activations saved by the forward kernel (mlp.hip, SAVE variant), all row-major
// [m,128].  Two hand-written fp32-MFMA GEMM kernels do the heavy lifting:
//   gemm_nn:  G_in[m x N]  = G_out[m x K] * W[K x N]           data gradient  (lane = sample, as the forward)
//   gemm_tn:  gW[No x Ki] += G_out[m x No]^T * X[m x Ki]        weight gradient (contraction over samples;
//             row-major activations put FEATURES on lanes, so both MFMA operands are coalesced loads and the
//             accumulator tile maps to 128-byte row segments of the weight matrix -> well-shaped float atomics)
// plus small fused element-wise kernels between them.  Weight gradients are accumulated straight into the
// flat parameter-gradient vector (same layout as the flat parameter vector), so no unpacking is needed.
#include "common.h"
#include "mlp_layout.h"
#include "mlp_bwd_parts.h"

namespace ucnerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

int launch_mlp_fwd(const ucnerf_mlp_params* p, const MlpSaved* save, hipStream_t st);
int launch_mlp_fwd_bf16x3_save(const ucnerf_mlp_params* p, const MlpSaved* save, hipStream_t st);      // mlp_bf16.hip
int launch_embed_strided(int m, int n_freqs, int layout, const float* x, float* out, int out_stride, hipStream_t st);      // rays.hip

// 4 consecutive parameters (parameter tensors are only 4-byte aligned inside the flat vector)
__device__ __forceinline__ f32x4 ld4(const float* p) { return f32x4{p[0], p[1], p[2], p[3]}; }

// ------------------------------------------------------------------------------------------------
// gemm_nn: C[s][n] (=|+=) sum_k A[s][k] W[k][n],  K = 64*KH, n < N <= 32*NT, optionally followed by the trunk
// layer's element-wise backward (EPI 1):  g_pre = C * [h > 0];  out = g_pre * bd;  g_bd (+)= g_pre * (h / bd).
//
// D rows = samples, D columns = output features, so that everything the epilogue touches (h, bd, g_bd, the output)
// is a coalesced 128-byte row segment per accumulator register.  The activations reach the A operand (lane = sample)
// through a wave-private LDS tile filled with coalesced 16-byte loads, one 64-wide half of K at a time with the next
// half's loads in flight under the MFMAs; the weight matrix sits in LDS once per (persistent) block, padded so that
// the two k-halves of a wave read disjoint banks.
// ------------------------------------------------------------------------------------------------
struct NnArgs {
    const float* A; int lda;
    const float* W; int ldw;
    float* C; int ldc;
    int m, N, accumulate;
    const float* hh; const float* bd; float* gbd;                // EPI 1 only (all [m,128]; g_bd accumulates)
    const float* W_hi;                                            // optional: rows k >= 64 come from this matrix (row k - 64)
};

constexpr int NN_WAVES = 8;
constexpr int NN_TP = 68;                                          // tile pitch (floats): 64 + 4 -> conflict-free b128 rows
struct NnEpi { float hv[8], b[8], g[8]; };            // operands of half a column tile (8 accumulator registers)
template <int KH, int NT> constexpr int nn_lds_bytes() { return (64 * KH * (32 * NT + 8) + NN_WAVES * 32 * NN_TP) * 4; }

template <int KH, int NT, int EPI>
__global__ void __launch_bounds__(64 * NN_WAVES) gemm_nn_kernel(NnArgs a, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float nn_lds[];
    constexpr int NP = 32 * NT + 8, K = 64 * KH;
    float* Wl = nn_lds;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* tile = nn_lds + K * NP + wave * (32 * NN_TP);
    {                                                                        // weights -> LDS (columns >= N as zeros)
        constexpr int KS = 64 * NN_WAVES / (32 * NT);                        // rows covered per pass: thread = (row, column)
        const int n = threadIdx.x % (32 * NT), k0 = threadIdx.x / (32 * NT);
        const int nc = n < a.N ? n : a.N - 1;
        float wv[K / KS];
#pragma unroll
        for (int i = 0; i < K / KS; ++i) {                                    // all loads in flight at once
            const int k = k0 + i * KS;
            wv[i] = a.W_hi && k >= 64 ? a.W_hi[(size_t)(k - 64) * a.ldw + nc] : a.W[(size_t)k * a.ldw + nc];
        }
#pragma unroll
        for (int i = 0; i < K / KS; ++i) Wl[(k0 + i * KS) * NP + n] = n < a.N ? wv[i] : 0.f;
    }
    __syncthreads();

    // staging: 32 rows x 16 float4 per half = 8 float4 per lane; 16 consecutive lanes cover one row
    const int srow = lane >> 4, sc4 = lane & 15;
    auto fetch = [&](int t, int kh, f32x4 (&v)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int s = t * 32 + 4 * u + srow;
            s = s < a.m ? s : a.m - 1;
            v[u] = *reinterpret_cast<const f32x4*>(a.A + (size_t)s * a.lda + 64 * kh + 4 * sc4);
        }
    };
    auto stage = [&](const f32x4 (&v)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(tile + (4 * u + srow) * NN_TP + 4 * sc4) = v[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");              // (wave-private tile: no barrier needed,
        __builtin_amdgcn_wave_barrier();                                     //  LDS operations of one wave stay in order)
    };

    NnEpi e0, e1;
    // (32-bit element offsets from uniform bases: one address register per access instead of a 64-bit pair)
    auto epi_off = [&](int t, int nt, int r) {
        int s = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        s = s < a.m ? s : a.m - 1;
        return (unsigned)s * 128u + (unsigned)(32 * nt + j);
    };
    auto epi_load = [&](int t, int unit, NnEpi& e) {                          // unit = 2 nt + register half (EPI 1: N = ldc = 128)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned o = epi_off(t, unit >> 1, 8 * (unit & 1) + r);
            e.hv[r] = a.hh[o]; e.b[r] = a.bd[o]; e.g[r] = a.gbd[o];
        }
    };
    const int stride = gridDim.x * NN_WAVES;
    int t = blockIdx.x * NN_WAVES + wave;
    f32x4 v[8];
    if (t < n_tiles) fetch(t, 0, v);
    for (; t < n_tiles; t += stride) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            stage(v);
            if (kh + 1 < KH) fetch(t, kh + 1, v);                             // lands under this half's MFMAs
            else if (t + stride < n_tiles) fetch(t + stride, 0, v);
            if (EPI == 1 && kh == KH - 1) epi_load(t, 0, e0);                 // first column tile's h / bd / g_bd land under the MFMAs
            const float* wk = Wl + (64 * kh + 4 * h) * NP + j;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(tile + j * NN_TP + 8 * q + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = MFMA(av[c], wk[(8 * q + c) * NP + 32 * nt], acc[nt]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the tile is rewritten next
            __builtin_amdgcn_wave_barrier();
        }
        // D[row][col]: row = (r&3) + 8(r>>2) + 4h (sample), col = lane&31 (feature)
        if (EPI == 1) {
#pragma unroll
            for (int unit = 0; unit < 2 * NT; ++unit) {
                NnEpi& cur = (unit & 1) ? e1 : e0;
                asm volatile("" ::: "memory");                                   // (keeps later units' loads from being hoisted -> spills)
                if (unit + 1 < 2 * NT) epi_load(t, unit + 1, (unit & 1) ? e0 : e1);   // next unit's operands under this one's arithmetic
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int nt = unit >> 1, rr = 8 * (unit & 1) + r;
                    const int s = t * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                    const bool on = cur.hv[r] > 0.f;
                    const float gp = on ? acc[nt][rr] : 0.f;
                    const float add = on ? gp * (cur.hv[r] / cur.b[r]) : 0.f;
                    if (s < a.m) {
                        const unsigned o = epi_off(t, nt, rr);
                        a.gbd[o] = cur.g[r] + add;
                        a.C[o] = gp * cur.b[r];
                    }
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 32 * nt + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int s = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (s < a.m && col < a.N) {
                        float* c = a.C + (size_t)s * a.ldc + col;
                        *c = a.accumulate ? *c + acc[nt][r] : acc[nt][r];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: gW[n][k] += sum_s G[s][n] X[s / xdiv][k];  gb[n] += sum_s G[s][n]
// block = 16 waves = sample groups x 4 row tiles x KSPLIT: a wave contracts its group's share of the block's sample
// chunk into rows 32nt..32nt+31 of its KT column tiles (up to 64 columns: four groups; wider: two waves split the
// columns of a row tile, two groups), the groups are summed through LDS and group 0 merges the block's tile into the
// gradient with float atomics on 128-byte row segments (one flush per `chunk` samples).
// The loop is branch-free: out-of-range rows / columns read a clamped (valid) column and are simply never
// stored -- a D element only depends on its own A row and B column -- with loads a few stages (8 samples each) ahead of
// the MFMAs in a register ring, four waves per SIMD.  Measured on 131 072 samples x 128 x 128: 58.7 us, of which the
// MFMAs need ~35 at the sustained clock and the merge (LDS sums + 4.2 M float atomics) ~15.
// ------------------------------------------------------------------------------------------------
struct TnArgs {
    const float* G; int ldg;
    const float* X; int ldx; int xdiv;
    float* gW; int ldw;
    float* gb;
    int m, Nout, Kin;
    int chunk;            // samples per block, a multiple of 32
    float* gW_hi; float* gb_hi;   // optional: rows n >= 64 belong to a second layer (row n - 64 of these)
};

#ifndef UCNERF_TN_BF16X3
#define UCNERF_TN_BF16X3 1        // weight-gradient contraction on the bf16 matrix cores with split operands (0: exact fp32 MFMA)
#endif
#ifndef UCNERF_TN_DEPTH2
#define UCNERF_TN_DEPTH2 2      // stages (8 samples each) of loads in flight (measured: 1, 2 and 3 time the same -- the loop is not latency-bound)
#endif

template <int KT, int NE>
struct TnStage { float g[NE], x[NE][KT]; };

// split-bf16 operands for the bf16 matrix cores (same scheme as the forward's bf16x3 kernel, csrc/mlp_bf16.hip): x = hi + lo
// with hi the truncated top 16 bits and lo = bf16_rne(x - hi); a product is hi*hi + hi*lo + lo*hi, the dropped term 2^-16.
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tn_u32x4 __attribute__((ext_vector_type(4)));
typedef float tn_f32x2 __attribute__((ext_vector_type(2)));
struct TnFrag { tn_bf16x8 hi, lo; };
__device__ __forceinline__ TnFrag tn_split8(const float (&x)[8]) {
    tn_u32x4 hi;
    TnFrag f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[j]), b1 = __builtin_bit_cast(unsigned, x[j + 1]);
        hi[j >> 1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const tn_f32x2 l = (tn_f32x2){x[j], x[j + 1]} - (tn_f32x2){__builtin_bit_cast(float, b0 & 0xffff0000u), __builtin_bit_cast(float, b1 & 0xffff0000u)};
        f.lo[j] = (__bf16)l.x;
        f.lo[j + 1] = (__bf16)l.y;
    }
    f.hi = __builtin_bit_cast(tn_bf16x8, hi);
    return f;
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// KT column tiles per wave, KSPLIT waves share a row tile (each with its own KT column tiles), DEPTH stages of loads in
// flight ahead of the MFMAs.  16 waves = (4 / KSPLIT sample groups) x 4 row tiles x KSPLIT.
// BF: the contraction runs on the bf16 matrix cores with split operands (three 32x32x16 MFMAs per 16 samples and tile
// instead of eight fp32 32x32x2 ones): fp32-grade weight gradients (relative error ~1e-5) at 2.4x the loop rate.
template <int KT, int KSPLIT, int DEPTH, bool DIV, bool BF>
__global__ void __launch_bounds__(1024) gemm_tn_kernel(TnArgs a) {
    constexpr int GROUPS = 4 / KSPLIT, NB = DEPTH + 1;
    constexpr int NE = BF ? 8 : 4;                 // operand elements per lane and stage
    constexpr int SPS = BF ? 16 : 8;               // samples per stage
    __shared__ f32x4 red[(GROUPS - 1) * 4 * KSPLIT * 4 * 64];     // [writer wave][q][lane], <= 48 KB
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: uniform branches)
    const int nt = wave & 3, ks = (wave >> 2) % KSPLIT, g = wave / (4 * KSPLIT);
    const int per = a.chunk / GROUPS;                             // multiple of SPS (the host picks the chunk)
    const int ws0 = blockIdx.x * a.chunk + g * per;
    const int ws1 = ws0 + per < a.m ? ws0 + per : a.m;
    const bool rows = 32 * nt < a.Nout && 32 * ks * KT < a.Kin;   // (wave-uniform) this wave's tiles exist
    const int gcol = 32 * nt + i < a.Nout ? 32 * nt + i : a.Nout - 1;
    int xcol[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) xcol[kt] = 32 * (ks * KT + kt) + i < a.Kin ? 32 * (ks * KT + kt) + i : a.Kin - 1;
    f32x16 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
    float colsum = 0.f;

    if (rows && ws0 < ws1) {
        // addresses = wave-uniform base + 32-bit byte offset (lane constant + uniform row term): one v_add per load
        // element e of a stage is sample  (fp32) 2e + h  /  (BF) 8h + e  of the stage: the lane-half term rides in the lane
        // constant, the element term in NE uniform base pointers
        constexpr int HS = BF ? 8 : 1, ES = BF ? 1 : 2;            // row strides of the lane half and of the element index
        const char* Gw = reinterpret_cast<const char*>(a.G + (size_t)ws0 * a.ldg);
        const char* Xw = reinterpret_cast<const char*>(a.X + (DIV ? (size_t)0 : (size_t)ws0 * a.ldx));
        const unsigned g_row = 4u * (unsigned)a.ldg, x_row = 4u * (unsigned)a.ldx;
        const unsigned g_lane = 4u * (unsigned)(HS * h * a.ldg + gcol);
        unsigned x_lane[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) x_lane[kt] = 4u * (unsigned)((DIV ? 0 : HS * h * a.ldx) + xcol[kt]);
        auto ldf = [](const char* base, unsigned off) { return *reinterpret_cast<const float*>(base + off); };
        const char *Gu[NE], *Xu[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) { Gu[e] = Gw + (size_t)(ES * e) * g_row; Xu[e] = Xw + (DIV ? (size_t)0 : (size_t)(ES * e) * x_row); }
        auto load = [&](int it, TnStage<KT, NE>& st) {
            const unsigned go = g_lane + (unsigned)(SPS * it) * g_row;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                st.g[e] = ldf(Gu[e], go);
                const unsigned xo = DIV ? (unsigned)((ws0 + SPS * it + ES * e + HS * h) / a.xdiv) * x_row : (unsigned)(SPS * it) * x_row;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    st.x[e][kt] = ldf(Xu[e], x_lane[kt] + xo);
            }
        };
        auto load_tail = [&](int it, TnStage<KT, NE>& st) {      // ragged last stage: clamped row, zeroed operands
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int sl = SPS * it + ES * e + HS * h;
                const bool ok = ws0 + sl < ws1;
                const int sc = ok ? sl : ws1 - 1 - ws0;
                const float gv = ldf(Gw, 4u * (unsigned)gcol + (unsigned)sc * g_row);
                st.g[e] = ok ? gv : 0.f;
                const unsigned xo = (DIV ? (unsigned)((ws0 + sc) / a.xdiv) : (unsigned)sc) * x_row;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) { const float xv = ldf(Xw, 4u * (unsigned)xcol[kt] + xo); st.x[e][kt] = ok ? xv : 0.f; }
            }
        };
        auto mma = [&](const TnStage<KT, NE>& st) {
            if constexpr (BF) {
                float gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { gv[e] = st.g[e]; colsum += st.g[e]; }
                const TnFrag gf = tn_split8(gv);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    float xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[e] = st.x[e][kt];
                    const TnFrag xf = tn_split8(xv);
                    acc[kt] = MFMA16(gf.hi, xf.hi, acc[kt]);
                    acc[kt] = MFMA16(gf.hi, xf.lo, acc[kt]);
                    acc[kt] = MFMA16(gf.lo, xf.hi, acc[kt]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    colsum += st.g[u];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[kt] = MFMA(st.g[u], st.x[u][kt], acc[kt]);
                }
            }
        };
        const int n_full = (ws1 - ws0) / SPS;
        TnStage<KT, NE> sb[NB];                                  // register ring (indices are compile-time after unrolling)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < n_full) load(d, sb[d]);
        for (int it = 0; it < n_full; it += NB) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (it + b < n_full) {
                    if (it + b + DEPTH < n_full) load(it + b + DEPTH, sb[(b + DEPTH) % NB]);
                    mma(sb[b]);
                }
        }
        if ((ws1 - ws0) % SPS) { load_tail(n_full, sb[0]); mma(sb[0]); }
    }

    // sum the sample groups: one column tile at a time through LDS, group 0 keeps the total
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        if (g > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[(((g - 1) * 4 * KSPLIT + (wave & (4 * KSPLIT - 1))) * 4 + q) * 64 + lane] =
                    f32x4{acc[kt][4 * q], acc[kt][4 * q + 1], acc[kt][4 * q + 2], acc[kt][4 * q + 3]};
        }
        __syncthreads();
        if (g == 0 && rows) {
#pragma unroll
            for (int gg = 0; gg < GROUPS - 1; ++gg)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = red[((gg * 4 * KSPLIT + wave) * 4 + q) * 64 + lane];
                    acc[kt][4 * q] += v.x; acc[kt][4 * q + 1] += v.y; acc[kt][4 * q + 2] += v.z; acc[kt][4 * q + 3] += v.w;
                }
            // D[row][col]: row = (r&3) + 8(r>>2) + 4h (output feature), col = lane&31 (input feature)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h, k = 32 * (ks * KT + kt) + i;
                float* drow = a.gW_hi && n >= 64 ? a.gW_hi + (size_t)(n - 64) * a.ldw : a.gW + (size_t)n * a.ldw;
                if (n < a.Nout && k < a.Kin) atomicAdd(drow + k, acc[kt][r]);
            }
        }
        __syncthreads();
    }
    if (a.gb && rows && ks == 0) {
        colsum += __shfl_xor(colsum, 32);
        const int n = 32 * nt + i;
        if (h == 0 && n < a.Nout) atomicAdd(a.gb_hi && n >= 64 ? a.gb_hi + n - 64 : a.gb + n, colsum);
    }
}

// ------------------------------------------------------------------------------------------------
// element-wise stages
// ------------------------------------------------------------------------------------------------
// Output stage (models.py:177-178 backwards) + both 4-wide heads, data AND weight gradients.  32 lanes per sample
// (4 features each), sixteen samples per block pass, persistent blocks (one per CU): every lane keeps the 4 x 4 products of its
// features with the head gradients in registers across its samples, the sample slots of a block are summed
// through LDS at the end and go to the flat gradient with one atomic per weight and block.  (As separate
// weight-gradient GEMMs these four 1-to-3-row products cost a full pass over the activations each.)
struct HeadArgs {
    int m, F;
    const float* raw;     // [m,4] forward output
    const float* g_raw;   // [m,4]
    const float* feats; int ldf;    // conf = feats[s][F-1]
    const float* h5;      // [m,128]
    const float* vc;      // [m,128] (post-relu)
    const float* w_crgb;  // confi_rgb_linear.weight [3,128]
    const float* w_a1;    // alpha_linear_1.weight [1,128]
    const float* w_rgb;   // rgb_linear.weight [3,64]
    const float* w_a;     // alpha_linear.weight [1,64]
    const float* b_crgb; const float* b_a1; const float* b_rgb; const float* b_a;
    float* g_base;        // [m,4]  d/d(base rgb, base sigma)
    float* g_adapt;       // [m,4]
    float* g_vc;          // [m,128] gradient at the pre-relu output of views/view_confi linears
    float* g_feats; int ldgf;       // writes column F-1 (confidence)
    float *gw_crgb, *gw_a1, *gw_rgb, *gw_a, *gb_crgb, *gb_a1, *gb_rgb, *gb_a;     // parameter gradients (accumulated)
};

constexpr int HEAD_SLOTS = 16;                           // samples per block pass (32 lanes each)
__global__ void __launch_bounds__(32 * HEAD_SLOTS) head_bwd_kernel(HeadArgs a) {
    __shared__ float red[HEAD_SLOTS][32][33];            // [sample slot][lane][value] (+1: conflict-free column sums)
    __shared__ float redb[HEAD_SLOTS][8];
    const int slot = threadIdx.x >> 5;
    const int c = threadIdx.x & 31;                      // features 4c..4c+3
    // weights of this lane's features (constant over samples)
    f32x4 wc[3], w1, wr[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, wa = {0, 0, 0, 0};
#pragma unroll
    for (int o = 0; o < 3; ++o) wc[o] = ld4(a.w_crgb + o * 128 + 4 * c);
    w1 = ld4(a.w_a1 + 4 * c);
    if (c < 16) {
#pragma unroll
        for (int o = 0; o < 3; ++o) wr[o] = ld4(a.w_rgb + o * 64 + 4 * c);
    } else {
        wa = ld4(a.w_a + 4 * (c - 16));
    }
    float accb[4][4], acca[4][4], bsum[8];               // [output][feature]: base heads x h5, adapt heads x vc; bias sums
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int k = 0; k < 4; ++k) accb[o][k] = acca[o][k] = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) bsum[o] = 0.f;

    // (operands of the next sample are fetched before this one's arithmetic: the loop is latency-bound otherwise)
    struct In { f32x4 hv, vv, raw, gr; float conf; };
    auto fetch = [&](int s) {
        In x;
        const int sc = s < a.m ? s : a.m - 1;
        x.hv = reinterpret_cast<const f32x4*>(a.h5 + (size_t)sc * 128)[c];
        x.vv = reinterpret_cast<const f32x4*>(a.vc + (size_t)sc * 128)[c];
        x.raw = reinterpret_cast<const f32x4*>(a.raw)[sc];
        x.gr = reinterpret_cast<const f32x4*>(a.g_raw)[sc];
        x.conf = a.feats[(size_t)sc * a.ldf + a.F - 1];
        return x;
    };
    const int stride = gridDim.x * HEAD_SLOTS;
    In nxt = fetch(blockIdx.x * HEAD_SLOTS + slot);
    for (int s = blockIdx.x * HEAD_SLOTS + slot; s < a.m; s += stride) {
        const In cur = nxt;
        nxt = fetch(s + stride);
        const f32x4 hv = cur.hv, vv = cur.vv;
        // this lane's share of the eight head pre-activations (4 of the 128 / 64 features each)
        float base[4], adapt[4];
#pragma unroll
        for (int o = 0; o < 3; ++o) base[o] = hv.x * wc[o].x + hv.y * wc[o].y + hv.z * wc[o].z + hv.w * wc[o].w;
        base[3] = hv.x * w1.x + hv.y * w1.y + hv.z * w1.z + hv.w * w1.w;
#pragma unroll
        for (int o = 0; o < 3; ++o) adapt[o] = vv.x * wr[o].x + vv.y * wr[o].y + vv.z * wr[o].z + vv.w * wr[o].w;
        adapt[3] = vv.x * wa.x + vv.y * wa.y + vv.z * wa.z + vv.w * wa.w;

        const f32x4 raw = cur.raw, gr = cur.gr;
        const float conf = cur.conf;
        const float u = 1.f - conf, omu = 1.f - u;
        const float gp[4] = {gr.x * raw.x * (1.f - raw.x), gr.y * raw.y * (1.f - raw.y), gr.z * raw.z * (1.f - raw.z),
                             raw.w > 0.f ? gr.w : 0.f};
        const float gb4[4] = {gp[0] * omu, gp[1] * omu, gp[2] * omu, gp[3] * u};
        const float ga4[4] = {gp[0] * u, gp[1] * u, gp[2] * u, gp[3] * omu};
        // d/du needs the head outputs only through one linear combination: fold it per lane, then ONE value crosses the
        // 32 lanes of the sample (instead of eight)
        float gu = gp[0] * (adapt[0] - base[0]) + gp[1] * (adapt[1] - base[1]) + gp[2] * (adapt[2] - base[2]) + gp[3] * (base[3] - adapt[3]);
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) gu += __shfl_xor(gu, d);
        if (c == 0) {
            gu += gp[0] * (a.b_rgb[0] - a.b_crgb[0]) + gp[1] * (a.b_rgb[1] - a.b_crgb[1]) + gp[2] * (a.b_rgb[2] - a.b_crgb[2]) +
                  gp[3] * (a.b_a1[0] - a.b_a[0]);
            reinterpret_cast<f32x4*>(a.g_base)[s] = f32x4{gb4[0], gb4[1], gb4[2], gb4[3]};
            reinterpret_cast<f32x4*>(a.g_adapt)[s] = f32x4{ga4[0], ga4[1], ga4[2], ga4[3]};
            a.g_feats[(size_t)s * a.ldgf + a.F - 1] = -gu;          // u = 1 - confidence
#pragma unroll
            for (int o = 0; o < 4; ++o) { bsum[o] += gb4[o]; bsum[4 + o] += ga4[o]; }
        }
        // head weight gradients: (head gradient) x (this lane's features)
        const float hvv[4] = {hv.x, hv.y, hv.z, hv.w}, vvv[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int k = 0; k < 4; ++k) { accb[o][k] += gb4[o] * hvv[k]; acca[o][k] += ga4[o] * vvv[k]; }
        // g_vc = relu'(vc) * (W_head^T g_adapt)
        f32x4 g;
        g.x = ga4[0] * wr[0].x + ga4[1] * wr[1].x + ga4[2] * wr[2].x + ga4[3] * wa.x;
        g.y = ga4[0] * wr[0].y + ga4[1] * wr[1].y + ga4[2] * wr[2].y + ga4[3] * wa.y;
        g.z = ga4[0] * wr[0].z + ga4[1] * wr[1].z + ga4[2] * wr[2].z + ga4[3] * wa.z;
        g.w = ga4[0] * wr[0].w + ga4[1] * wr[1].w + ga4[2] * wr[2].w + ga4[3] * wa.w;
        g.x = vv.x > 0.f ? g.x : 0.f; g.y = vv.y > 0.f ? g.y : 0.f; g.z = vv.z > 0.f ? g.z : 0.f; g.w = vv.w > 0.f ? g.w : 0.f;
        reinterpret_cast<f32x4*>(a.g_vc + (size_t)s * 128)[c] = g;
    }

    // block sums over the eight sample slots, then one atomic per weight
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[slot][c][4 * o + k] = accb[o][k]; red[slot][c][16 + 4 * o + k] = acca[o][k]; }
    if (c == 0) {
#pragma unroll
        for (int o = 0; o < 8; ++o) redb[slot][o] = bsum[o];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 1024 / (32 * HEAD_SLOTS); ++q) {
        const int e = threadIdx.x + 32 * HEAD_SLOTS * q;            // 1024 = 32 lanes x 32 values
        const int cc = e >> 5, i = e & 31, o = (i & 15) >> 2, k = i & 3;
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < HEAD_SLOTS; ++sl) v += red[sl][cc][i];
        float* dst = nullptr;
        if (i < 16) dst = o < 3 ? a.gw_crgb + o * 128 + 4 * cc + k : a.gw_a1 + 4 * cc + k;
        else if (cc < 16) { if (o < 3) dst = a.gw_rgb + o * 64 + 4 * cc + k; }
        else if (o == 3) dst = a.gw_a + 4 * (cc - 16) + k;
        if (dst) atomicAdd(dst, v);
    }
    if (threadIdx.x < 8) {
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < HEAD_SLOTS; ++sl) v += redb[sl][threadIdx.x];
        const int o = threadIdx.x & 3;
        float* dst = threadIdx.x < 4 ? (o < 3 ? a.gb_crgb + o : a.gb_a1) : (o < 3 ? a.gb_rgb + o : a.gb_a);
        atomicAdd(dst, v);
    }
}

// after feature_linear, down to the trunk's top layer in one pass over the activations (models.py:158-165 and the layer-5
// relu / modulation backwards):  g_bc = g_g * h5;  g_h5 = g_g * bc + W_basehead^T g_base;  g_pre = g_h5 * [h5 > 0];
// g_y = g_pre * bd (written over g_g in place);  g_bd = g_pre * (h5 / bd);  also gx = h5 * bc for feature_linear's weights
struct TopArgs {
    size_t n4;
    f32x4* g;                // in: g_g, out: g_y of layer 5
    const f32x4* bc; const f32x4* h5; const f32x4* bd;
    const float* g_base;     // [m,4]
    const float* w_crgb; const float* w_a1;
    f32x4* g_bc; f32x4* g_bd;
    f32x4* gx;               // out: h5 * bc, the input of feature_linear (operand of its weight gradient)
};

__global__ void __launch_bounds__(256) trunk_top_bwd_kernel(TopArgs a) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * 256) {
        const size_t s = i >> 5;
        const int c = (int)(i & 31);
        const f32x4 g = a.g[i], b = a.bc[i], h = a.h5[i], d = a.bd[i];
        const f32x4 gb = reinterpret_cast<const f32x4*>(a.g_base)[s];
        const f32x4 w0 = ld4(a.w_crgb + 4 * c), w1 = ld4(a.w_crgb + 128 + 4 * c), w2 = ld4(a.w_crgb + 256 + 4 * c),
                    w3 = ld4(a.w_a1 + 4 * c);
        a.g_bc[i] = f32x4{g.x * h.x, g.y * h.y, g.z * h.z, g.w * h.w};
        a.gx[i] = f32x4{h.x * b.x, h.y * b.y, h.z * b.z, h.w * b.w};
        f32x4 y, acc;
#define ONE(C)                                                                                              \
        {                                                                                                   \
            const float gh = g.C * b.C + gb.x * w0.C + gb.y * w1.C + gb.z * w2.C + gb.w * w3.C;            \
            const bool on = h.C > 0.f;                                                                      \
            const float gp = on ? gh : 0.f;                                                                 \
            y.C = gp * d.C;                                                                                 \
            acc.C = on ? gp * (h.C / d.C) : 0.f;                                                            \
        }
        ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
        a.g[i] = y;
        a.g_bd[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
struct BwdWork {
    MlpSaved sv;
    float *pep, *ped, *g1, *g2, *g3, *gbd, *gx, *g_base, *g_adapt, *g_sigma, *raw;
    float* gy[6];           // bwd_mode 0: g_y of the six trunk layers (operands of their weight-gradient GEMMs)
    float* wstream_bwd;     // bwd_mode 0: transposed weights as split-bf16 fragments + head table (mlp_bwd_chain.hip)
    unsigned* wg_counters;  // bwd_mode 0: the weight-gradient launch's chunk counters (mlp_wgrad.hip)
};

static size_t carve_bwd(float* base, int m, int n_dirs, BwdWork* w) {
    size_t o = 0;
    auto take = [&](size_t k) { float* r = base ? base + o : nullptr; o += (k + 3) & ~(size_t)3; return r; };
    const size_t M = (size_t)m;
    w->sv.p24 = 0;
    w->sv.bd = take(M * 128);
    for (int l = 0; l < 6; ++l) w->sv.h[l] = take(M * 128);
    w->sv.bc = take(M * 128); w->sv.ft = take(M * 128); w->sv.vc = take(M * 128);
    w->pep = take(M * 64); w->ped = take((size_t)n_dirs * 32);      // encodings as matrices, rows padded to whole 16-byte pieces (64 / 32 floats)
    // (the sets the gradient chain writes have one spare row behind row m - 1 -- lanes without a sample store there, mlp_bwd_chain.hip --: as 24-bit
    //  rows they fill three quarters of M * 128 floats, which leaves it for M >= 3; the explicit tail covers smaller M)
    w->g1 = take(M * 128 + 96); w->g2 = take(M * 128 + 96); w->g3 = take(M * 128 + 96); w->gbd = take(M * 128 + 96); w->gx = take(M * 128 + 96);
    w->g_base = take(M * 4 + 4); w->g_adapt = take(M * 4 + 4); w->g_sigma = take(M * 4 + 4); w->raw = take(M * 4);
    for (int l = 0; l < 6; ++l) w->gy[l] = take(M * 128 + 96);
    w->wstream_bwd = take(bwd_chain_stream_floats());
    w->wg_counters = reinterpret_cast<unsigned*>(take(WG_MAX_PAIRS));
    return o;
}

template <int KH, int NT, int EPI>
static int launch_nn(hipStream_t st, const NnArgs& a) {
    // > 64 KB of dynamic LDS needs the opt-in, per device
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nn_kernel<KH, NT, EPI>), nn_lds_bytes<KH, NT>(), "mlp_bwd gemm_nn")) return rc;
    const int n_tiles = cdiv(a.m, 32);
    int blocks = cdiv(n_tiles, NN_WAVES);
    if (blocks > device_cus()) blocks = device_cus();                     // persistent: the weights are staged once per block
    const int lds = nn_lds_bytes<KH, NT>();
    hipLaunchKernelGGL((gemm_nn_kernel<KH, NT, EPI>), dim3(blocks), dim3(64 * NN_WAVES), lds, st, a, n_tiles);
    return check_launch("mlp_bwd gemm_nn");
}

// {h, bd, g_bd}: fuse the trunk layer's element-wise backward into the epilogue (N = K = 128 only)
static int run_nn(hipStream_t st, int m, const float* A, int lda, int K, const float* W, int ldw, int N, float* C, int ldc,
                  bool accumulate, const float* hh = nullptr, const float* bd = nullptr, float* gbd = nullptr, const float* W_hi = nullptr) {
    NnArgs a{A, lda, W, ldw, C, ldc, m, N, accumulate ? 1 : 0, hh, bd, gbd, W_hi};
    if (((uintptr_t)A & 15) || (lda & 3)) return fail(UCNERF_EINVAL, "mlp_bwd: gemm_nn operand must be 16-byte aligned rows");
    const bool wide = N > 64;
    if (hh) {
        if (K != 128 || N != 128 || ldc != 128 || accumulate || m > (1 << 24)) return fail(UCNERF_EINVAL, "mlp_bwd: fused trunk epilogue is 128 x 128 only");
        return launch_nn<2, 4, 1>(st, a);
    }
    if (K == 128) return wide ? launch_nn<2, 4, 0>(st, a) : launch_nn<2, 2, 0>(st, a);
    if (K == 64) return wide ? launch_nn<1, 4, 0>(st, a) : launch_nn<1, 2, 0>(st, a);
    return fail(UCNERF_EINVAL, "mlp_bwd: gemm_nn K = %d", K);
}

static int run_tn(hipStream_t st, int m, const float* G, int ldg, int Nout, const float* X, int ldx, int xdiv, int Kin, float* gW,
                  int ldw, float* gb, float* gW_hi = nullptr, float* gb_hi = nullptr) {
    // samples per block: about one block per CU for a training batch (~1e5 samples), so that the float atomics of the
    // merge stay a small part of the launch; larger batches get proportionally larger chunks
    int chunk = 64 * cdiv(cdiv(m, device_cus()), 64);           // (a multiple of 64: every sample group is whole 16-sample stages)
    if (chunk < 512) chunk = 512;
    if (chunk > 8192) chunk = 8192;
    for (int k0 = 0; k0 < Kin; k0 += 128) {               // at most 4 accumulator tiles per launch
        const int kin = Kin - k0 < 128 ? Kin - k0 : 128;
        TnArgs a{G, ldg, X + k0, ldx, xdiv, gW + k0, ldw, k0 == 0 ? gb : nullptr, m, Nout, kin, chunk,
                 gW_hi ? gW_hi + k0 : nullptr, k0 == 0 ? gb_hi : nullptr};
        dim3 grid(cdiv(m, chunk)), block(1024);
        const int kt = cdiv(kin, 32);
        constexpr bool BF = UCNERF_TN_BF16X3 != 0;
        if (xdiv != 1) {
            if (kt != 1) return fail(UCNERF_EINVAL, "mlp_bwd: gemm_tn with shared rows is built for <= 32 columns");
            hipLaunchKernelGGL((gemm_tn_kernel<1, 1, 2, true, false>), grid, block, 0, st, a);
        } else if (kt == 1) hipLaunchKernelGGL((gemm_tn_kernel<1, 1, BF ? 1 : UCNERF_TN_DEPTH2, false, BF>), grid, block, 0, st, a);
        else if (kt == 2) hipLaunchKernelGGL((gemm_tn_kernel<2, 1, BF ? 1 : UCNERF_TN_DEPTH2, false, BF>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, BF ? 1 : UCNERF_TN_DEPTH2, false, BF>), grid, block, 0, st, a);     // 3 or 4 column tiles: two waves per row tile
        int rc = check_launch("mlp_bwd gemm_tn");
        if (rc) return rc;
    }
    return UCNERF_OK;
}

#define RUN(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

const char* build_flags_mlp_bwd() { return "mlp_bwd: " UCNERF_FLAG(UCNERF_TN_BF16X3) UCNERF_FLAG(UCNERF_TN_DEPTH2); }

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int64_t ucnerf_mlp_bwd_workspace_floats(const ucnerf_mlp_config* cfg, int32_t m) {
    MlpLayout L;
    if (!cfg || !mlp_layout(cfg->n_src, &L) || m < 0) return fail(UCNERF_EINVAL, "mlp_bwd_workspace: bad arguments");
    BwdWork w;
    return (int64_t)carve_bwd(nullptr, m, m, &w);      // upper bound (direction rows <= m)
}

int ucnerf_mlp_fwd_train(const ucnerf_mlp_params* p, float* bwd_workspace, int32_t bwd_mode, void* stream) {
    UCNERF_REQUIRE(p && bwd_workspace, "mlp_fwd_train: null pointer");
    UCNERF_REQUIRE(bwd_mode == 0 || bwd_mode == 1, "mlp_fwd_train: bwd_mode %d", bwd_mode);
    UCNERF_COUNT(p->m);
    UCNERF_REQUIRE(p->cfg.precision == 0 || p->cfg.precision == 1, "mlp_fwd_train: the training forward runs in f32 or bf16x3 precision");
    UCNERF_REQUIRE(((uintptr_t)bwd_workspace & 15) == 0, "mlp_fwd_train: workspace must be 16-byte aligned");
    BwdWork w;
    carve_bwd(bwd_workspace, p->m, p->dirs_per_sample ? p->m : p->m / (p->S > 0 ? p->S : 1), &w);
    hipStream_t st = (hipStream_t)stream;
    w.sv.p24 = bwd_mode == 0;                       // the gradient chain reads the sets as 24-bit floats, the layer-by-layer backward as fp32
    if (p->cfg.precision == 1) { RUN(launch_mlp_fwd_bf16x3_save(p, &w.sv, st)); }      // split-bf16 matrix cores
    else RUN(launch_mlp_fwd(p, &w.sv, st));
    // the backward reads the forward's output from its own slot
    if (hipMemcpyAsync(w.raw, p->raw, (size_t)p->m * 4 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(UCNERF_EHIP, "mlp_fwd_train: copy failed");
    return UCNERF_OK;
}

int ucnerf_mlp_bwd(const ucnerf_mlp_bwd_params* bp, void* stream) {
    UCNERF_REQUIRE(bp, "mlp_bwd: null params");
    const ucnerf_mlp_params& f = bp->fwd;
    UCNERF_COUNT(f.m);
    UCNERF_REQUIRE(f.pts && f.dirs && f.feats && f.wstream && bp->g_raw && bp->flat_params && bp->g_flat && bp->g_feats &&
                       bp->workspace, "mlp_bwd: null pointer");
    UCNERF_REQUIRE(!f.feats_tiled || (bp->bwd_mode == 0 && !f.encoded), "mlp_bwd: features in the tile layout are read by the gradient chain (bwd_mode 0) only");
    UCNERF_REQUIRE(f.cfg.pe_layout == 0 || f.cfg.pe_layout == 1, "mlp_bwd: pe_layout %d", f.cfg.pe_layout);
    UCNERF_REQUIRE(((uintptr_t)bp->workspace & 15) == 0 && ((uintptr_t)bp->g_raw & 15) == 0, "mlp_bwd: workspace/g_raw must be 16-byte aligned");
    MlpLayout L;
    UCNERF_REQUIRE(mlp_layout(f.cfg.n_src, &L), "mlp_bwd: n_src %d outside 1..8", f.cfg.n_src);
    hipStream_t st = (hipStream_t)stream;
    const int m = f.m, v = L.v, F = L.F, n_mvs = 24 + 4 * v, n_img = 8 * v;
    const int n_dirs = f.dirs_per_sample ? m : m / f.S;
    const int xdiv_dir = f.dirs_per_sample ? 1 : f.S;
    const int ldf = f.feat_stride ? f.feat_stride : F;            // row stride of feats
    const int ldgf = bp->g_feat_stride ? bp->g_feat_stride : F;   // row stride of g_feats
    BwdWork w;
    carve_bwd(bp->workspace, m, n_dirs, &w);
    const float* P = bp->flat_params;
    float* G = bp->g_flat;
    const size_t n4 = (size_t)m * 32;
    const int ew_blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);

    // 0. forward with the activations kept; encodings as explicit matrices for the weight-gradient GEMMs
    UCNERF_REQUIRE(f.cfg.precision == 0 || (f.cfg.precision == 1 && bp->saved_valid),
                   "mlp_bwd: without the activations of ucnerf_mlp_fwd_train (saved_valid) the backward re-runs the forward itself, in f32 "
                   "precision only (pack the weights with precision 0)");
    if (!bp->saved_valid) {
        ucnerf_mlp_params fw = f;
        fw.raw = w.raw;
        w.sv.p24 = bp->bwd_mode == 0;
        RUN(launch_mlp_fwd(&fw, &w.sv, st));
    }
    const float *pep = w.pep, *ped = w.ped;      // encodings as matrices: [m,63] and [n_dirs,27]
    int ld_pep = 64, ld_ped = 32;
    if (f.encoded) {                              // already in memory (UCNeRF.forward(x) of the reference)
        pep = f.pts; ld_pep = f.pts_stride ? f.pts_stride : 63;
        ped = f.dirs; ld_ped = f.dirs_stride ? f.dirs_stride : 27;
    } else {
        UCNERF_REQUIRE((f.pts_stride == 0 || f.pts_stride == 3) && (f.dirs_stride == 0 || f.dirs_stride == 3),
                       "mlp_bwd: strided raw pts/dirs are not supported");
        RUN(launch_embed_strided(m, 10, f.cfg.pe_layout, f.pts, w.pep, ld_pep, st));
        RUN(launch_embed_strided(n_dirs, 4, f.cfg.pe_layout, f.dirs, w.ped, ld_ped, st));
    }

    const int KV = MLP_W + MLP_PE_DIR;
    UCNERF_REQUIRE(bp->bwd_mode == 0 || bp->bwd_mode == 1, "mlp_bwd: bwd_mode %d", bp->bwd_mode);
    if (bp->bwd_mode == 0) {
        // A. every data gradient in ONE launch (mlp_bwd_chain.hip): g stays in registers from the output stage to the two bias nets;
        //    written: the G operands of the weight-gradient GEMMs below, g_feats, g_base / g_adapt
        RUN(launch_pack_bwd(v, P, w.wstream_bwd, st));
        RUN(launch_mlp_bwd_chain(v, m, w.raw, bp->g_raw, f.feats, ldf, f.feats_tiled, &w.sv, w.wstream_bwd, w.g1, w.g2, w.g3, w.gx, w.gbd, w.gy, bp->g_feats, ldgf,
                                 w.g_base, w.g_adapt, w.g_sigma, st));
        // B. every parameter gradient in ONE launch (mlp_wgrad.hip): one (g, layer input) pair per product.  Every G and every kept
        //    activation set is in the 24-bit format (rows of 384 bytes; the head G's: 12); encodings and gathered features are fp32
        constexpr int R = P24_ROW_BYTES;
        const char* vc24 = reinterpret_cast<const char*>(w.sv.vc);
        WgArgs wg;
        wgrad_begin(&wg, m, w.wg_counters);
        RUN(wgrad_add(&wg, w.g1, R, 128, w.sv.ft, 1, R, 1, 128, G + L.p_vw, KV, G + L.p_vb, G + L.p_vcw, G + L.p_vcb, 64));       // [views | view_confi] x f
        RUN(wgrad_add(&wg, w.g1, R, 128, ped, 0, ld_ped, xdiv_dir, 27, G + L.p_vw + 128, KV, nullptr, G + L.p_vcw + 128, nullptr, 64));   // ... x dir encoding
        RUN(wgrad_add(&wg, w.g2, R, 128, w.gx, 1, R, 1, 128, G + L.p_fw, 128, G + L.p_fb, nullptr, nullptr, 0));                    // feature_linear
        const int xt = f.feats_tiled ? F : 0;                 // (tile layout: column c of the features = row c of every tile)
        RUN(wgrad_add(&wg, w.g3, R, 128, f.feats + (xt ? 32 * n_mvs : n_mvs), 0, ldf, 1, n_img, G + L.p_bcw, n_img, G + L.p_bcb, nullptr, nullptr, 0, xt));   // confidence-bias net
        RUN(wgrad_add(&wg, w.gy[5], R, 128, pep, 0, ld_pep, 1, 63, G + L.p_lw[5], 191, G + L.p_lb[5], nullptr, nullptr, 0));          // layer 5 on [pe | h4]
        RUN(wgrad_add(&wg, w.gy[5], R, 128, w.sv.h[4], 1, R, 1, 128, G + L.p_lw[5] + 63, 191, nullptr, nullptr, nullptr, 0));
        for (int l = 4; l >= 1; --l) RUN(wgrad_add(&wg, w.gy[l], R, 128, w.sv.h[l - 1], 1, R, 1, 128, G + L.p_lw[l], 128, G + L.p_lb[l], nullptr, nullptr, 0));
        RUN(wgrad_add(&wg, w.gy[0], R, 128, pep, 0, ld_pep, 1, 63, G + L.p_lw[0], 63, G + L.p_lb[0], nullptr, nullptr, 0));
        RUN(wgrad_add(&wg, w.gbd, R, 128, f.feats, 0, ldf, 1, n_mvs, G + L.p_bdw, n_mvs, G + L.p_bdb, nullptr, nullptr, 0, xt));      // depth-bias net
        // the four head layers: base rgb (3 rows) + base sigma (row 3) on h5; adapt rgb on the views half of vc, adapt sigma on the view_confi half
        RUN(wgrad_add(&wg, w.g_base, 12, 4, w.sv.h[5], 1, R, 1, 128, G + L.p_crw, 128, G + L.p_crb, G + L.p_a1w, G + L.p_a1b, 3));
        RUN(wgrad_add(&wg, w.g_adapt, 12, 3, vc24, 1, R, 1, 64, G + L.p_rw, 64, G + L.p_rb, nullptr, nullptr, 0));
        RUN(wgrad_add(&wg, w.g_sigma, 12, 1, vc24 + 3 * 64, 1, R, 1, 64, G + L.p_aw, 64, G + L.p_ab, nullptr, nullptr, 0));
        RUN(wgrad_launch(&wg, st));
        return UCNERF_OK;
    }

    // bwd_mode 1: layer by layer, exact-fp32 data-gradient GEMMs
    // 1. output stage + heads: g_base, g_adapt, d/d(confidence), g_vc -> g1, and the four head layers' parameter gradients
    HeadArgs ha;
    ha.m = m; ha.F = F; ha.raw = w.raw; ha.g_raw = bp->g_raw; ha.feats = f.feats; ha.ldf = ldf; ha.h5 = w.sv.h[5]; ha.vc = w.sv.vc;
    ha.w_crgb = P + L.p_crw; ha.w_a1 = P + L.p_a1w; ha.w_rgb = P + L.p_rw; ha.w_a = P + L.p_aw;
    ha.b_crgb = P + L.p_crb; ha.b_a1 = P + L.p_a1b; ha.b_rgb = P + L.p_rb; ha.b_a = P + L.p_ab;
    ha.g_base = w.g_base; ha.g_adapt = w.g_adapt; ha.g_vc = w.g1; ha.g_feats = bp->g_feats; ha.ldgf = ldgf;
    ha.gw_crgb = G + L.p_crw; ha.gw_a1 = G + L.p_a1w; ha.gw_rgb = G + L.p_rw; ha.gw_a = G + L.p_aw;
    ha.gb_crgb = G + L.p_crb; ha.gb_a1 = G + L.p_a1b; ha.gb_rgb = G + L.p_rb; ha.gb_a = G + L.p_ab;
    {
        int blocks = cdiv(m, HEAD_SLOTS);                 // one block per CU: the merge is one atomic per weight and block
        if (blocks > device_cus()) blocks = device_cus();
        hipLaunchKernelGGL(head_bwd_kernel, dim3(blocks), dim3(32 * HEAD_SLOTS), 0, st, ha);
    }
    RUN(check_launch("mlp_bwd head"));

    // 2. views_linears / view_confi_linears: weights [64,155] on [f | dir encoding]; g_f -> g2
    //    (g1 = [g_views | g_view_confi]: one 128-row product per operand, rows 64.. go to the second layer)
    RUN(run_tn(st, m, w.g1, 128, 128, w.sv.ft, 128, 1, 128, G + L.p_vw, KV, G + L.p_vb, G + L.p_vcw, G + L.p_vcb));
    RUN(run_tn(st, m, w.g1, 128, 128, ped, ld_ped, xdiv_dir, 27, G + L.p_vw + 128, KV, nullptr, G + L.p_vcw + 128, nullptr));
    RUN(run_nn(st, m, w.g1, 128, 128, P + L.p_vw, KV, 128, w.g2, 128, false, nullptr, nullptr, nullptr, P + L.p_vcw));

    // 3. feature_linear (input gx = h5 * bc): g_g = g_f * W_f -> g1; its weight gradient waits for gx, which step 4 writes
    RUN(run_nn(st, m, w.g2, 128, 128, P + L.p_fw, 128, 128, w.g1, 128, false));

    // 4. one pass: g_bc -> g3, and g_g (g1) becomes g_y of trunk layer 5 in place, g_bd initialised; confidence-bias net
    TopArgs ta;
    ta.n4 = n4; ta.g = (f32x4*)w.g1; ta.bc = (const f32x4*)w.sv.bc; ta.h5 = (const f32x4*)w.sv.h[5]; ta.bd = (const f32x4*)w.sv.bd;
    ta.g_base = w.g_base; ta.w_crgb = P + L.p_crw; ta.w_a1 = P + L.p_a1w; ta.g_bc = (f32x4*)w.g3; ta.g_bd = (f32x4*)w.gbd;
    ta.gx = (f32x4*)w.gx;
    hipLaunchKernelGGL(trunk_top_bwd_kernel, dim3(ew_blocks), dim3(256), 0, st, ta);
    RUN(check_launch("mlp_bwd trunk_top"));
    RUN(run_tn(st, m, w.g2, 128, 128, w.gx, 128, 1, 128, G + L.p_fw, 128, G + L.p_fb));      // (g2 = g_f is intact until the trunk loop)
    RUN(run_tn(st, m, w.g3, 128, 128, f.feats + n_mvs, ldf, 1, n_img, G + L.p_bcw, n_img, G + L.p_bcb));
    RUN(run_nn(st, m, w.g3, 128, 128, P + L.p_bcw, n_img, n_img, bp->g_feats + n_mvs, ldgf, false));

    // 5. trunk, layers 5..0.  g_y of layer 5 is in g1 (step 4); below that the data-gradient GEMM of layer l applies
    //    layer l-1's element-wise backward in its epilogue (g_y ping-pongs between g1 and g2, g_bd accumulates in place)
    float *gy = w.g1, *gnext = w.g2;
    for (int l = 5; l >= 0; --l) {
        if (l == 0) {
            RUN(run_tn(st, m, gy, 128, 128, pep, ld_pep, 1, 63, G + L.p_lw[0], 63, G + L.p_lb[0]));
            break;
        }
        const float* Wl = l == 5 ? P + L.p_lw[5] + 63 : P + L.p_lw[l];
        const int ldw = l == 5 ? 191 : 128;
        if (l == 5) RUN(run_tn(st, m, gy, 128, 128, pep, ld_pep, 1, 63, G + L.p_lw[5], 191, G + L.p_lb[5]));
        RUN(run_tn(st, m, gy, 128, 128, w.sv.h[l - 1], 128, 1, 128, G + (Wl - P), ldw, l == 5 ? nullptr : G + L.p_lb[l]));
        RUN(run_nn(st, m, gy, 128, 128, Wl, ldw, 128, gnext, 128, false, w.sv.h[l - 1], w.sv.bd, w.gbd));
        float* t = gy; gy = gnext; gnext = t;
    }

    // 6. depth-bias net
    RUN(run_tn(st, m, w.gbd, 128, 128, f.feats, ldf, 1, n_mvs, G + L.p_bdw, n_mvs, G + L.p_bdb));
    RUN(run_nn(st, m, w.gbd, 128, 128, P + L.p_bdw, n_mvs, n_mvs, bp->g_feats, ldgf, false));
    return UCNERF_OK;
}

}  // extern "C"
