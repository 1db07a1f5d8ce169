// K6 backward, parameter-gradient half: EVERY weight-gradient product of the network in one persistent launch
//     gW[n][k] += sum_s G[s][n] X[s][k],   gb[n] += sum_s G[s][n]        (contraction over samples)
// for the (G, X) pairs the gradient chain (mlp_bwd_chain.hip) leaves behind: g_y of a layer with the layer's input.
//
// Why not one gemm_tn launch per pair (mlp_bwd.hip, rounds 1-2): there every wave fetched its own operands from global
// memory as dwords (features on lanes), 24 vector-memory instructions per nine MFMAs and every byte of G / X through the
// CU's texture path four to eight times -- 57 us per 128 x 128 pair against 27 us for its bytes at HBM speed.  Here
//   * a block stages 64 samples of G and X through LDS ONCE: waves 0-3 fetch G, waves 4-7 fetch X (the next stage's, in registers, while this
//     one is multiplied).  Operands in the 24-bit format (p24.h: every G, and every X that is a kept activation set) arrive as 12-byte pieces
//     -- thread (4-column group, sample octet) reads eight rows' pieces = four MFMA fragments; fp32 operands (encodings, gathered features)
//     as coalesced dwords -- thread (column, sample octets hf, hf + 2, ...).  Either way every value is split into (hi, lo) bf16 ONCE and
//     written in operand layout; the eight waves then read their fragments with one ds_read_b128 per plane;
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
typedef w_u32x4 w_u32x4_a4 __attribute__((aligned(4)));
#define WMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#ifndef UCNERF_WGRAD_EXP
#define UCNERF_WGRAD_EXP 0      // timing experiments (wrong results), bit mask: 1 no unpack / split arithmetic (raw dwords into LDS), 2 no MFMAs,
                                //   4 every operand load inside the first 1024 samples (cache resident), 8 no LDS fragment writes, 16 no fragment reads
#endif
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

// The two sides' fetches have ONE shape whatever the operand's format: eight 16-byte loads at eight computed byte offsets.  (With a different
// load sequence per format behind wave-uniform branches the compiler merged the branches' tails, kept the merged values in a stack slot and
// waited for the loads right behind the fetch: nothing of the next stage was in flight during the multiply.)
//   24-bit rows (every G; kept activation sets as X): thread (grp = 4-column group, oct = sample octet): the 12-byte pieces of eight rows
//                (the load's fourth dword belongs to the next piece and is ignored) -> four fragments
//   fp32 rows   (encodings, row-major features; one row per xdiv samples): the same with 16-byte pieces = four fp32 columns
//   fp32 tiles  (features in the MLP tile layout [m / 32][F][32]): thread (column, tile of the stage): the 32 samples of its column = four fragments
enum { WG_X24 = 0, WG_XROWS = 1, WG_XTILES = 2 };
__global__ void __launch_bounds__(WG_THREADS, 4) mlp_wgrad_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wg_lds[];
    char* const Gl = wg_lds;
    char* const Xl = wg_lds + WG_OP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, kg = wave >> 2;                  // this wave multiplies: row tile nt, column tiles kg and kg + 2
    // ... and fetches: waves 0-3 the G operand, waves 4-7 the X operand (`xside`, wave-uniform), 256 threads per operand and stage
    const int xside = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int t8 = tid & 255;
    const int grp = t8 & 31, oct = t8 >> 5;                   // rows:  columns 4 grp .. 4 grp + 3, samples 8 oct .. 8 oct + 7 of the stage
    const int tcol = t8 & 127, ttile = t8 >> 7;               // tiles: column tcol, samples 32 ttile .. 32 ttile + 31 of the stage
    char* const Ol = xside ? Xl : Gl;                         // the operand image this thread fills
    // fragment (column f, octet o) of a stage: k-step o >> 1, lane half o & 1, row tile f >> 5; hi plane here, lo plane 4 * 64 * 16 bytes on
    // (the slot of column i inside its 32-slot half is i ^ key, key = (tile + half + 2 (i >> 4)) & 3: a thread writes the fragments of columns 4 grp + c one c
    //  at a time -- 64-byte strides, four of the sixteen bank groups; with the key the 64 lanes of a write cover all sixteen.  Readers apply the same key.)
    auto swz = [](int tile, int hh, int i) { return i ^ ((tile + hh + 2 * (i >> 4)) & 3); };
    auto frag_off = [&](int f, int o) { return ((((o >> 1) * 2 * 4) + (f >> 5)) * 64 + (o & 1) * 32 + swz(f >> 5, o & 1, f & 31)) * 16; };

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

    struct Regs { w_u32x4 d[8]; };
    // global -> registers, and nothing else: rows / columns outside the matrix read a clamped (valid) piece that stage_lds replaces by zero
    auto fetch = [&](Regs& r, int pr, int stg) {
        const WgPair& q = a.p[pr];
        const int mode = xside ? q.xmode : WG_X24;
        const char* base = reinterpret_cast<const char*>(xside ? q.X : q.G);
        unsigned off[8];                                      // byte offsets from the uniform base (wgrad_add checks that they fit 32 bits)
        if (mode == WG_XTILES) {
            const int s0 = stg * WG_STAGE + 32 * ttile;
            const unsigned tl = (unsigned)((s0 < a.m ? s0 : a.m - 1) >> 5), c = (unsigned)(tcol < q.w ? tcol : q.w - 1);
            const unsigned o0 = (tl * (unsigned)q.xtile_f + c) * 128u;
#pragma unroll
            for (int e = 0; e < 8; ++e) off[e] = o0 + 16u * e;
        } else {
            const unsigned ld = (unsigned)(xside ? q.ldx : q.ldg);
            const int width = xside ? q.w : q.nout;
            const unsigned piece = mode == WG_X24 ? 12u : 16u;
            const unsigned gc = piece * (unsigned)(4 * grp < width ? grp : (width - 1) >> 2);
            const unsigned one = !xside || q.xdiv == 1 ? 0xffffffffu : 0u;          // (uniform) s / 1 without a branch
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int s = stg * WG_STAGE + 8 * oct + e;
                const unsigned sc = (unsigned)(s < a.m ? s : a.m - 1);
                off[e] = ((sc & one) | ((__umulhi(sc, q.div_m) >> q.div_sh) & ~one)) * ld + gc;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#if UCNERF_WGRAD_EXP & 4
            off[e] &= 0x3ffffu;
#endif
            r.d[e] = *reinterpret_cast<const w_u32x4_a4*>(base + off[e]);
        }
    };
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};                   // bias gradients of columns 4 grp .. 4 grp + 3: this thread's samples (G side only)
    // registers -> (hi, lo) fragments -> LDS: every value is split exactly once
    auto put_frag = [&](const float (&x)[8], int f, int o) {
#if UCNERF_WGRAD_EXP & 1
        WFrag fr;
        fr.hi = __builtin_bit_cast(w_bf16x8, (w_f32x4){x[0], x[1], x[2], x[3]});
        fr.lo = __builtin_bit_cast(w_bf16x8, (w_f32x4){x[4], x[5], x[6], x[7]});
#else
        const WFrag fr = w_split8(x);
#endif
#if UCNERF_WGRAD_EXP & 8
        if (x[0] != 12345.678f) return;
#endif
        const int off = frag_off(f, o);
        *reinterpret_cast<w_bf16x8*>(Ol + off) = fr.hi;
        *reinterpret_cast<w_bf16x8*>(Ol + off + 4 * 64 * 16) = fr.lo;
    };
    auto stage_lds = [&](const Regs& r, int pr, int stg) {
        const WgPair& q = a.p[pr];
        const int mode = xside ? q.xmode : WG_X24;
        const int width = xside ? q.w : q.nout;
        if (mode == WG_XTILES) {                              // fragment k = samples 8 k .. 8 k + 7 of this thread's tile: loads 2 k, 2 k + 1
            const bool fin = tcol < width;
            const int s0 = stg * WG_STAGE + 32 * ttile;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (s0 + 8 * k + e < a.m) && fin ? __uint_as_float(r.d[2 * k + (e >> 2)][e & 3]) : 0.f;
                put_frag(x, tcol, 4 * ttile + k);
            }
        } else {                                              // fragment c = column 4 grp + c of the eight rows
            const int s0 = stg * WG_STAGE + 8 * oct;
            const bool is24 = mode == WG_X24;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = 4 * grp + c;
                const bool fin = f < width;
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v24 = p24_unpack1(r.d[e][0], r.d[e][1], r.d[e][2], c);     // (one byte permute per value, straight from the piece)
                    const float v = is24 ? v24 : __uint_as_float(r.d[e][c]);
                    x[e] = (s0 + e < a.m) && fin ? v : 0.f;
                    colsum[c] += x[e];                        // (the X side's sums are never read)
                }
                put_frag(x, f, oct);
            }
        }
    };

    w_f32x16 acc[2];
    auto zero = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) colsum[c] = 0.f;
    };
    auto multiply = [&](const WgPair& q) {
        if (32 * nt < q.nout) {                              // (wave-uniform) this wave's row tile exists
#pragma unroll
            for (int ks = 0; ks < WG_KS; ++ks) {
                __builtin_amdgcn_sched_barrier(0);           // (one k-step's fragments at a time: read up front for all four they crowd the prefetched stage out of the registers)
                const w_bf16x8* ga = reinterpret_cast<const w_bf16x8*>(Gl) + (ks * 2 * 4 + nt) * 64 + (lane & 32) + swz(nt, lane >> 5, lane & 31);
                const w_bf16x8 ahi = ga[0], alo = ga[4 * 64];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int kt = kg + 2 * t;
                    if (32 * kt < q.w) {
                        const w_bf16x8* xb = reinterpret_cast<const w_bf16x8*>(Xl) + (ks * 2 * 4 + kt) * 64 + (lane & 32) + swz(kt, lane >> 5, lane & 31);
                        const w_bf16x8 bhi = xb[0], blo = xb[4 * 64];
#if UCNERF_WGRAD_EXP & 2
                        acc[t][0] += __builtin_bit_cast(w_f32x4, ahi)[0] + __builtin_bit_cast(w_f32x4, bhi)[0] + __builtin_bit_cast(w_f32x4, alo)[1] + __builtin_bit_cast(w_f32x4, blo)[1];
#else
                        acc[t] = WMFMA(ahi, bhi, acc[t]);
                        acc[t] = WMFMA(ahi, blo, acc[t]);
                        acc[t] = WMFMA(alo, bhi, acc[t]);
#endif
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
        if (gb && !xside) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = 4 * grp + c;
                if (col < nout && colsum[c] != 0.f) {
                    const bool hi = twob && col >= split;
                    atomicAdd((hi ? gbh : gb) + (hi ? col - split : col), colsum[c]);
                }
            }
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

int wgrad_add(WgArgs* a, const void* G, int ldg_bytes, int nout, const void* X, int x24, int ldx, int xdiv, int wd, float* gW, int ldw, float* gb, float* gW_hi,
              float* gb_hi, int split, int xtile_f) {
    if (a->n_pairs >= WG_MAX_PAIRS) return fail(UCNERF_EINVAL, "mlp_bwd: more than %d weight-gradient pairs", WG_MAX_PAIRS);
    if ((long long)a->m * ldx * (x24 ? 1 : 4) >= (1ll << 32) || (long long)a->m * ldg_bytes >= (1ll << 32) || (long long)a->m * xtile_f * 4 >= (1ll << 32))
        return fail(UCNERF_EINVAL, "mlp_bwd: %d samples x row stride %d / %d overflow the kernel's 32-bit offsets; split the batch", a->m, ldg_bytes, ldx);
    if (nout < 1 || nout > 128 || wd < 1 || wd > 128 || xdiv < 1) return fail(UCNERF_EINVAL, "mlp_bwd: weight-gradient pair %d x %d (xdiv %d) outside the kernel's tile", nout, wd, xdiv);
    if ((ldg_bytes & 3) || ((uintptr_t)G & 3) || ((uintptr_t)X & 3) || (x24 && ((ldx & 3) || (wd & 3) || xdiv != 1)))
        return fail(UCNERF_EINVAL, "mlp_bwd: a 24-bit weight-gradient operand has 4-byte aligned rows, one per sample, of whole 4-column groups");
    WgPair& q = a->p[a->n_pairs];
    q.G = G; q.ldg = ldg_bytes; q.nout = nout; q.X = X; q.ldx = x24 ? ldx : 4 * ldx; q.xdiv = xdiv; q.w = wd; q.gW = gW; q.ldw = ldw; q.gb = gb; q.gW_hi = gW_hi; q.gb_hi = gb_hi;
    q.xmode = x24 ? WG_X24 : xtile_f > 0 ? WG_XTILES : WG_XROWS;
    q.split = gW_hi || gb_hi ? split : 1 << 30;
    q.xtile_f = xtile_f;
    if (xtile_f > 0 && (xdiv != 1 || x24)) return fail(UCNERF_EINVAL, "mlp_bwd: a tiled weight-gradient operand is fp32 with one row per sample");
    {   // magic numbers for s / xdiv (s < 2^31): __umulhi(s, div_m) >> div_sh
        unsigned l = 1;
        while ((1u << l) < (unsigned)xdiv) ++l;
        q.div_m = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)xdiv - 1) / (unsigned)xdiv);
        q.div_sh = l - 1;
    }
    // ~ bytes moved per stage (a 24-bit column weighs 3/4 of an fp32 one) + a fixed part (a stage is mostly latency: measured, weighting by bytes
    //   alone left the blocks of the narrow pairs running 1.5x longer than the rest)
    q.cost = ((nout + 31) & ~31) * 3 / 4 + ((wd + 31) & ~31) * (x24 ? 3 : 4) / 4 + 224;
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

const char* build_flags_mlp_wgrad() { return "mlp_wgrad: " UCNERF_FLAG(UCNERF_WGRAD_EXP); }

}  // namespace ucnerf
