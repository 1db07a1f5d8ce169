// K6 backward, parameter-gradient half: EVERY weight-gradient product of the network in one persistent launch
//     gW[n][k] += sum_s G[s][n] X[s][k],   gb[n] += sum_s G[s][n]        (contraction over samples)
// for the (G, X) pairs the gradient chain (mlp_bwd_chain.hip) leaves behind: g_y of a layer with the layer's input.
//
// Why not one gemm_tn launch per pair (mlp_bwd.hip, rounds 1-2): there every wave fetched its own operands from global
// memory as dwords (features on lanes), 24 vector-memory instructions per nine MFMAs and every byte of G / X through the
// CU's texture path four to eight times -- 57 us per 128 x 128 pair against 27 us for its bytes at HBM speed.  Here
//   * a block stages 64 samples of G and X through LDS ONCE: thread (column, sample group) fetches eight consecutive samples of its
//     column with coalesced dword loads (the next stage's, in registers, while this one is multiplied) -- exactly one MFMA fragment -- splits it into (hi, lo) bf16
//     ONCE and writes both planes in operand layout; the eight waves then read their fragments with one ds_read_b128 per plane;
//   * products are hi*hi + hi*lo + lo*hi on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate: the forward's bf16x3
//     scheme, 2^-16 relative), no vector arithmetic in the multiply loop;
//   * the launch is persistent over the CONCATENATION of all pairs: the total cost (bytes per 64-sample stage, summed over
//     pairs) is cut into equal ranges, one per block, two blocks per CU; a block flushes its 128 x 128 partial sum with float
//     atomics on 128-byte row segments when its range leaves a pair -- ~530 flushes per step instead of ~3 300, no tail.
#include "common.h"
#include "mlp_layout.h"
#include "mlp_bwd_parts.h"

namespace ucnerf {

typedef float w_f32x16 __attribute__((ext_vector_type(16)));
typedef float w_f32x4 __attribute__((ext_vector_type(4)));
typedef float w_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w_u32x4 __attribute__((ext_vector_type(4)));
#define WMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int WG_STAGE = 64;              // samples per stage (four k16-steps)
constexpr int WG_KS = WG_STAGE / 16;
constexpr int WG_THREADS = 512;

struct WFrag { w_bf16x8 hi, lo; };
__device__ __forceinline__ WFrag w_split8(const float (&x)[8]) {
    w_u32x4 hi;
    WFrag f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[j]), b1 = __builtin_bit_cast(unsigned, x[j + 1]);
        hi[j >> 1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const w_f32x2 l = (w_f32x2){x[j], x[j + 1]} - (w_f32x2){__builtin_bit_cast(float, b0 & 0xffff0000u), __builtin_bit_cast(float, b1 & 0xffff0000u)};
        f.lo[j] = (__bf16)l.x;
        f.lo[j + 1] = (__bf16)l.y;
    }
    f.hi = __builtin_bit_cast(w_bf16x8, hi);
    return f;
}

// LDS image of one stage and operand: [k16-step][hi, lo][row tile 0..3][64 lanes][8 bf16] = 32 KB: the fragment of (k-step, tile) of
// lane (feature i, half hh) = samples 16 ks + 8 hh + 0..7 of feature 32 tile + i, ready for one ds_read_b128 per plane
constexpr int WG_OP_BYTES = WG_KS * 2 * 4 * 64 * 16;

__global__ void __launch_bounds__(WG_THREADS, 4) mlp_wgrad_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wg_lds[];
    char* const Gl = wg_lds;
    char* const Xl = wg_lds + WG_OP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, kg = wave >> 2;                  // this wave multiplies: row tile nt, column tiles kg and kg + 2
    // ... and fetches: column `col` of both operands, samples 8 rg .. 8 rg + 7 of the stage = exactly one MFMA fragment each
    const int col = tid & 127, rg = tid >> 7;
    // (a stage has eight sample groups: this thread takes groups rg and rg + 4, i.e. k-steps rg >> 1 and 2 + (rg >> 1), same lane half)
    const int frag_off = (((rg >> 1) * 2 * 4 + (col >> 5)) * 64 + (rg & 1) * 32 + (col & 31)) * 16;      // hi plane of the first; lo plane 4 * 64 * 16 bytes on, second fragment 2 k-steps on

    // this block's share of the concatenated pairs, in cost units -> (pair, stage) at both ends
    const long long total = a.prefix[a.n_pairs];
    const long long c0 = total * blockIdx.x / gridDim.x, c1 = total * (blockIdx.x + 1) / gridDim.x;
    auto locate = [&](long long c, int* pair, int* stage) {   // first stage whose start is >= c (so consecutive blocks tile exactly)
        int p = 0;
        for (int q = 1; q < a.n_pairs; ++q) p = c >= a.prefix[q] ? q : p;
        if (c >= total) { *pair = a.n_pairs; *stage = 0; return; }
        const long long off = c - a.prefix[p];
        int st = (int)((off + a.p[p].cost - 1) / a.p[p].cost);
        if (st >= a.stages) { ++p; st = 0; }
        *pair = p; *stage = st;
    };
    int p, st, p_end, st_end;
    locate(c0, &p, &st);
    locate(c1, &p_end, &st_end);
    if (p == p_end && st == st_end) return;

    struct Regs { float g[16], x[16]; };
    // global -> registers: sixteen coalesced dword loads per operand, 256 contiguous bytes per wave instruction, and nothing else: rows /
    // columns outside the matrix read a clamped (valid) element that stage_lds replaces by zero -- with the selection (or an exec mask, or a
    // branch on xdiv) next to each load the compiler put every load in a basic block of its own, with waits in between.
    auto fetch = [&](Regs& r, int pr, int stg) {
        const WgPair& q = a.p[pr];
        const unsigned gc = (unsigned)(col < q.nout ? col : q.nout - 1), xc = (unsigned)(col < q.w ? col : q.w - 1);
        const unsigned one = q.xdiv == 1 ? 0xffffffffu : 0u;          // (uniform) s / 1 without a branch
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int s0 = stg * WG_STAGE + 32 * f + 8 * rg;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int s = s0 + e;
                const unsigned sc = (unsigned)(s < a.m ? s : a.m - 1);
                const unsigned xrow = (sc & one) | ((__umulhi(sc, q.div_m) >> q.div_sh) & ~one);
                r.g[8 * f + e] = q.G[(size_t)(sc * (unsigned)q.ldg + gc)];
                // (X in the MLP tile layout [m / 32][F][32]: column c of sample s at ((s / 32) F + c) 32 + s % 32)
                const unsigned xo = q.xtile_f > 0 ? ((sc >> 5) * (unsigned)q.xtile_f + xc) * 32u + (sc & 31u) : xrow * (unsigned)q.ldx + xc;
                r.x[8 * f + e] = q.X[(size_t)xo];
            }
        }
    };
    float colsum = 0.f;                                       // bias gradient of column `col`: this thread's samples
    // registers -> (hi, lo) fragments -> LDS: every value is split exactly once
    auto stage_lds = [&](const Regs& r, int pr, int stg) {
        const WgPair& q = a.p[pr];
        const bool gin = col < q.nout, xin = col < q.w;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int s0 = stg * WG_STAGE + 32 * f + 8 * rg;
            float g[8], x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = s0 + e < a.m;
                g[e] = in && gin ? r.g[8 * f + e] : 0.f;
                x[e] = in && xin ? r.x[8 * f + e] : 0.f;
                colsum += g[e];
            }
            const WFrag gf = w_split8(g), xf = w_split8(x);
            const int o = frag_off + f * (2 * 2 * 4 * 64 * 16);
            *reinterpret_cast<w_bf16x8*>(Gl + o) = gf.hi;
            *reinterpret_cast<w_bf16x8*>(Gl + o + 4 * 64 * 16) = gf.lo;
            *reinterpret_cast<w_bf16x8*>(Xl + o) = xf.hi;
            *reinterpret_cast<w_bf16x8*>(Xl + o + 4 * 64 * 16) = xf.lo;
        }
    };

    w_f32x16 acc[2];
    auto zero = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        colsum = 0.f;
    };
    auto multiply = [&](const WgPair& q) {
        if (32 * nt < q.nout) {                              // (wave-uniform) this wave's row tile exists
#pragma unroll
            for (int ks = 0; ks < WG_KS; ++ks) {
                const w_bf16x8* ga = reinterpret_cast<const w_bf16x8*>(Gl) + (ks * 2 * 4 + nt) * 64 + lane;
                const w_bf16x8 ahi = ga[0], alo = ga[4 * 64];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int kt = kg + 2 * t;
                    if (32 * kt < q.w) {
                        const w_bf16x8* xb = reinterpret_cast<const w_bf16x8*>(Xl) + (ks * 2 * 4 + kt) * 64 + lane;
                        const w_bf16x8 bhi = xb[0], blo = xb[4 * 64];
                        acc[t] = WMFMA(ahi, bhi, acc[t]);
                        acc[t] = WMFMA(ahi, blo, acc[t]);
                        acc[t] = WMFMA(alo, bhi, acc[t]);
                    }
                }
            }
        }
    };
    // leaving a pair: this block's partial sums go to the gradient (float atomics on 128-byte row segments)
    auto flush = [&](const WgPair& q) {
        const int i = lane & 31, hh = lane >> 5;
        // (descriptor fields into scalars first: a select between two FIELDS became a per-lane load of the chosen one, with a vmcnt(0) per atomic)
        float* const gW = q.gW;
        float* const gWh = q.gW_hi ? q.gW_hi : q.gW;
        float* const gb = q.gb;
        float* const gbh = q.gb_hi ? q.gb_hi : q.gb;
        const int split = q.split, ldw = q.ldw, nout = q.nout, w = q.w;
        const bool two = q.gW_hi != nullptr, twob = q.gb_hi != nullptr;
        if (32 * nt < nout) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = kg + 2 * t, k = 32 * kt + i;
                if (32 * kt < w) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (n < nout && k < w) {
                            const bool hi = two && n >= split;
                            float* base = hi ? gWh : gW;
                            atomicAdd(base + (size_t)(hi ? n - split : n) * ldw + k, acc[t][r]);
                        }
                    }
                }
            }
        }
        if (gb && col < nout && colsum != 0.f) {
            const bool hi = twob && col >= split;
            atomicAdd((hi ? gbh : gb) + (hi ? col - split : col), colsum);
        }
        zero();
    };

    zero();
    // One segment (the block's stages of ONE pair) at a time.  A stage's loads (64 KB per block, two blocks per CU) are in flight while the
    // previous stage is multiplied.  (Two stages ahead in two register sets was built: the compiler's wait-count pass waits for vmcnt(0)
    // at a loop header whatever the loop looks like -- conditional fetches, exits and the flush's atomics were all removed from it in
    // turn -- so the second set never stayed in flight.  Bytes in flight come from the stage size instead.)
    while (!(p == p_end && st == st_end)) {
        const int seg_end = p == p_end ? st_end : a.stages;   // stages [st, seg_end) of pair p
        Regs r;
        fetch(r, p, st);
        for (int t = st; t < seg_end; ++t) {
            __syncthreads();                                 // every wave has read the previous stage
            stage_lds(r, p, t);
            __syncthreads();
            fetch(r, p, t + 1 < seg_end ? t + 1 : t);        // (unconditional: the last stage is fetched again and ignored)
            multiply(a.p[p]);
        }
        flush(a.p[p]);
        if (seg_end == a.stages) { ++p; st = 0; } else st = seg_end;
    }
}

// ---- host side: the pair list is built by ucnerf_mlp_bwd (mlp_bwd.hip)
void wgrad_begin(WgArgs* a, int m) { memset(a, 0, sizeof(*a)); a->m = m; a->stages = cdiv(m, WG_STAGE); }

int wgrad_add(WgArgs* a, const float* G, int ldg, int nout, const float* X, int ldx, int xdiv, int w, float* gW, int ldw, float* gb, float* gW_hi,
              float* gb_hi, int split, int xtile_f) {
    if (a->n_pairs >= WG_MAX_PAIRS) return fail(UCNERF_EINVAL, "mlp_bwd: more than %d weight-gradient pairs", WG_MAX_PAIRS);
    if ((long long)a->m * (ldg > ldx ? ldg : ldx) >= (1ll << 32)) return fail(UCNERF_EINVAL, "mlp_bwd: %d samples x row stride %d overflow the kernel's 32-bit element offsets", a->m, ldg > ldx ? ldg : ldx);
    if (nout < 1 || nout > 128 || w < 1 || w > 128 || xdiv < 1) return fail(UCNERF_EINVAL, "mlp_bwd: weight-gradient pair %d x %d (xdiv %d) outside the kernel's tile", nout, w, xdiv);
    WgPair& q = a->p[a->n_pairs];
    q.G = G; q.ldg = ldg; q.nout = nout; q.X = X; q.ldx = ldx; q.xdiv = xdiv; q.w = w; q.gW = gW; q.ldw = ldw; q.gb = gb; q.gW_hi = gW_hi; q.gb_hi = gb_hi;
    q.split = gW_hi || gb_hi ? split : 1 << 30;
    q.xtile_f = xtile_f;
    if (xtile_f > 0 && xdiv != 1) return fail(UCNERF_EINVAL, "mlp_bwd: a tiled weight-gradient operand has one row per sample");
    {   // magic numbers for s / xdiv (s < 2^31): __umulhi(s, div_m) >> div_sh
        unsigned l = 1;
        while ((1u << l) < (unsigned)xdiv) ++l;
        q.div_m = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)xdiv - 1) / (unsigned)xdiv);
        q.div_sh = l - 1;
    }
    q.cost = ((nout + 31) & ~31) + ((w + 31) & ~31) + 256;      // ~ bytes moved per stage + a fixed part (a stage is mostly latency: measured, weighting by bytes alone
                                                                 //   left the blocks of the narrow pairs running 1.5x longer than the rest)
    a->prefix[a->n_pairs + 1] = a->prefix[a->n_pairs] + (long long)a->stages * q.cost;
    ++a->n_pairs;
    return UCNERF_OK;
}

int wgrad_launch(const WgArgs* a, hipStream_t st) {
    if (a->n_pairs == 0 || a->m <= 0) return UCNERF_OK;
    const int cus = device_cus();
    if (cus <= 0) return fail(UCNERF_EHIP, "mlp_bwd: no device");
    long long units = (long long)a->n_pairs * a->stages;
    int blocks = 2 * cus;
    if (blocks > units) blocks = (int)units;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_wgrad_kernel), 2 * WG_OP_BYTES, "mlp_bwd wgrad")) return rc;
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(blocks), dim3(WG_THREADS), 2 * WG_OP_BYTES, st, *a);
    return check_launch("mlp_bwd wgrad");
}

}  // namespace ucnerf
