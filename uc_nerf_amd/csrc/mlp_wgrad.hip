// K6 backward, parameter-gradient half: EVERY weight-gradient product of the network in one persistent launch
//     gW[n][k] += sum_s G[s][n] X[s][k],   gb[n] += sum_s G[s][n]        (contraction over samples)
// for the (G, X) pairs the gradient chain (mlp_bwd_chain.hip) leaves behind: g_y of a layer with the layer's input.
//
// Why not one gemm_tn launch per pair (mlp_bwd.hip, rounds 1-2): there every wave fetched its own operands from global
// memory as dwords (features on lanes), 24 vector-memory instructions per nine MFMAs and every byte of G / X through the
// CU's texture path four to eight times -- 57 us per 128 x 128 pair against 27 us for its bytes at HBM speed.  Here
//   * one block per CU, twelve waves in three ROLES, three waves per SIMD.  Waves 0-3 produce the X operand, waves 4-7 the G operand: each
//     fetches 64 samples of its operand per stage into registers, two stages ahead in two register sets, splits every value into (hi, lo)
//     bf16 ONCE and writes it in MFMA operand layout into one of TWO stage images in LDS.  Waves 8-11 are consumers: they read fragments (one
//     ds_read_b128 per plane) from the other image and multiply -- nothing else.  One barrier per stage hands an image over in each direction,
//     so a producer's vector arithmetic, its LDS writes and the wait for its loads run beside the consumers' MFMAs on the same SIMD, and
//     the next-but-one stage's loads are issued the moment a register set has been converted.
//     (Before: two blocks per CU, every wave fetching, converting and multiplying in turn behind two barriers per stage, the next stage's loads
//      issued only after the LDS writes: a stage cost its load round trip PLUS its arithmetic.)
//   * operands in the 24-bit format (p24.h: every G, and every X that is a kept activation set) arrive as 12-byte pieces -- thread
//     (4-column group, sample octet) reads eight rows' pieces = four MFMA fragments; fp32 operands (encodings, gathered features) as
//     16-byte pieces of rows or of the MLP tile layout;
//   * products are hi*hi + hi*lo + lo*hi on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate: the forward's bf16x3
//     scheme, 2^-16 relative);
//   * the launch is persistent over all pairs and hands its work out as it runs: chunks of 2-16 stages through one counter per pair
//     (see "the block's work" in the kernel); a block stays on one pair as long as that pair has chunks, and the consumers flush their
//     128 x 128 partial sum with float atomics on 128-byte row segments only when the block moves to another pair.
// What bounds it (in-kernel stamps of the diagnostic build, scripts/micro/load_shapes.hip, profiles/r03_experiments.md): the loads are never
// waited for (17 cycles per stage) but their ISSUE is -- the memory system takes this traffic at 3.8-4.2 TB/s, because the sets were written
// by the two launches before this one: the same streams read at 6.1 TB/s from data at rest and at 4.7 TB/s right after every byte was rewritten.
#include <type_traits>
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

constexpr int WG_STAGE = 64;              // samples per stage (four k16-steps)
constexpr int WG_KS = WG_STAGE / 16;
constexpr int WG_THREADS = 768;            // waves 0-3 produce the X operand, 4-7 the G operand, 8-11 multiply

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
constexpr int WG_IMG_BYTES = 2 * WG_OP_BYTES;                 // a stage: [G operand | X operand]
constexpr int WG_LDS_BYTES = 2 * WG_IMG_BYTES;                // two stages: one being written, one being multiplied
constexpr int WG_UNROLL = 2;                                  // stages per trip of the stage loops (register sets A, B / images 0, 1)

// A side's fetch has ONE shape whatever the operand's format: eight 16-byte loads at eight computed byte offsets.
//   24-bit rows (every G; kept activation sets as X): thread (grp = 4-column group, oct = sample octet): the 12-byte pieces of eight rows
//                (the load's fourth dword belongs to the next piece and is ignored) -> four fragments
//   fp32 rows   (encodings, row-major features; one row per xdiv samples): the same with 16-byte pieces = four fp32 columns
//   fp32 tiles  (features in the MLP tile layout [m / 32][F][32]): thread (column, tile of the stage): the 32 samples of its column = four fragments
enum { WG_X24 = 0, WG_XROWS = 1, WG_XTILES = 2 };
struct WgCur { int j, p, st, left; };                         // a stage of the block's work: chunk number, pair, stage of the pair, stages left in the chunk (0 = END)
struct WgChunk { int p, st, n; };                             // ring entry: pair, first stage, stages (0 = END)
constexpr int WG_CHUNK_MAX = 16;                              // stages per chunk: chosen per launch (wgrad_launch), 2 .. 16: 16 at the bench's sizes (4: +8 %, 8: +2 %, 32: +1.5 %)
constexpr int WG_RING = 8, WG_LEAD = 4;                       // ring entries; chunks asked for ahead of the one being multiplied (>= 4: chunks may be one stage long)
// float atomic add on a pointer SAID to be global memory (the descriptors travel through LDS: a generic pointer would make it a FLAT atomic, and
// every later wait of the wave vmcnt(0) lgkmcnt(0))
__device__ __forceinline__ void wg_atomic_add(float* p, float v) {
    typedef float __attribute__((address_space(1))) * gp_t;
    __hip_atomic_fetch_add((gp_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }     // LDS traffic only: loads in flight stay in flight

#define WG_T(k)
#define WG_STAMP_ARG
__global__ void __launch_bounds__(WG_THREADS, 3) mlp_wgrad_kernel(WgArgs a WG_STAMP_ARG) {
    extern __shared__ __attribute__((aligned(16))) char wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 8;                           // (wave-uniform)
    const int t8 = tid & 255;
    int m_ = a.m, stages_ = a.stages;                        // (made opaque: the compiler otherwise re-reads kernel arguments wherever it is short of a register)
    asm volatile("" : "+s"(m_), "+s"(stages_));
    const int wg_m = m_, wg_stages = stages_;
    const int grp = t8 & 31, oct = t8 >> 5;                   // rows:  columns 4 grp .. 4 grp + 3, samples 8 oct .. 8 oct + 7 of the stage
    const int tcol = t8 & 127, ttile = t8 >> 7;               // tiles: column tcol, samples 32 ttile .. 32 ttile + 31 of the stage
    // fragment (column f, octet o) of a stage: k-step o >> 1, lane half o & 1, row tile f >> 5; hi plane here, lo plane 4 * 64 * 16 bytes on
    // (the slot of column i inside its 32-slot half is i ^ key, key = (tile + half + 2 (i >> 4)) & 3: a thread writes the fragments of columns 4 grp + c one c
    //  at a time -- 64-byte strides, four of the sixteen bank groups; with the key the 64 lanes of a write cover all sixteen.  Readers apply the same key.)
    auto swz = [](int tile, int hh, int i) { return i ^ ((tile + hh + 2 * (i >> 4)) & 3); };
    auto frag_off = [&](int f, int o) { return ((((o >> 1) * 2 * 4) + (f >> 5)) * 64 + (o & 1) * 32 + swz(f >> 5, o & 1, f & 31)) * 16; };

    // The pair descriptors live in the kernel-argument segment: a.p[dynamic index] inside the stage loop is a handful of scalar loads from that
    // buffer per stage and wave.  One copy into LDS instead, and the current pair of every cursor in scalar registers (reloaded from LDS when
    // a cursor changes pair): the stage loops have no scalar memory access at all.
    __shared__ WgPair lds_pairs[WG_MAX_PAIRS];
    {
        constexpr int words = (int)(sizeof(WgPair) / 4);
        const unsigned* src = reinterpret_cast<const unsigned*>(&a.p[0]);
        unsigned* dst = reinterpret_cast<unsigned*>(&lds_pairs[0]);
        for (int i = tid; i < a.n_pairs * words; i += WG_THREADS) dst[i] = src[i];
        __syncthreads();
    }
    auto load_pair = [&](WgPair& q, int pr) {
        constexpr int words = (int)(sizeof(WgPair) / 4);
        unsigned w[words];
        const unsigned* src = reinterpret_cast<const unsigned*>(&lds_pairs[pr]);
#pragma unroll
        for (int i = 0; i < words; ++i) w[i] = __builtin_amdgcn_readfirstlane(src[i]);
        __builtin_memcpy(&q, w, sizeof(WgPair));
    };
    // ---- the block's work: a sequence of CHUNKS (a.chunk consecutive stages of one pair), decided as the launch runs.
    // With a static split (equal modelled cost per block: rounds 1-3) the blocks finished between 0.6 and 1.3 of the mean: under this launch
    // a stage took 2.1 us on four of the eight XCDs and 3.3 us on the other four (which four changed from box to box; in-kernel stamps,
    // profiles/r03_experiments.md), and the split leaned on a cost model fitted to one kernel version.  Handing the work out as it is done
    // leaves the launch's time where the memory system puts it (all eight XCDs then run at 2.9 us per stage) but takes the cost model and the
    // dies' differences out of it.  Every pair's chunks are handed out through a counter in global memory.  A block starts on its
    // HOME pair (blocks are spread over the pairs in proportion to their cost), whose first chunks it owns without asking (ns rows of the pair's
    // chunks, one chunk per home block and row), then takes the pair's remaining chunks from the counter, then helps the pair with the most
    // chunks left.  Consumer wave 8 does the asking, WG_LEAD chunks ahead of the stage being multiplied, and writes the answers into a ring in
    // LDS that every cursor of every wave follows; the ring's END entry carries the block's total number of stages (lds_total).  Partial sums
    // are flushed when a cursor's next chunk belongs to another pair.
    __shared__ WgChunk ring[WG_RING];
    __shared__ int lds_total;                                 // stages of this block, once its END is known (a huge number before)
    __shared__ int sch_fb[WG_MAX_PAIRS + 1], sch_base[WG_MAX_PAIRS], sch_dyn[WG_MAX_PAIRS];      // first home block of a pair; chunks owned statically; chunks behind the counter
    const int n_pairs = __builtin_amdgcn_readfirstlane(a.n_pairs);
    const int wg_chunk = __builtin_amdgcn_readfirstlane(a.chunk);
    const int chunks = (wg_stages + wg_chunk - 1) / wg_chunk;
    if (tid <= n_pairs) {
        const long long total = a.prefix[n_pairs];
        sch_fb[tid] = tid == n_pairs ? (int)gridDim.x : (int)((a.prefix[tid] * (long long)gridDim.x + total - 1) / total);      // blocks b with prefix[p] <= total b / grid
    }
    if (tid == 0) lds_total = 0x3fffffff;                   // ("not known yet"; rounded up to even below)
    __syncthreads();
    if (tid < n_pairs) {
        const int nb = sch_fb[tid + 1] - sch_fb[tid];
        const int ns = nb > 0 ? (chunks / nb < WG_LEAD ? chunks / nb : WG_LEAD) : 0;
        sch_base[tid] = ns * nb;
        sch_dyn[tid] = chunks - ns * nb;
    }
    __syncthreads();
    // producer of ring entries: wave 8 only (all 64 lanes run it: shuffles inside)
    int g_next = 0, g_cum = 0, g_pref = 0;                    // entries written, their stages, the pair asked first
    bool g_end = false;
    auto ring_put = [&](int pr, int c) {
        const int st = c * wg_chunk, n = wg_stages - st < wg_chunk ? wg_stages - st : wg_chunk;
        if (lane == 0) ring[g_next & (WG_RING - 1)] = WgChunk{pr, st, n};
        g_cum += n;
        ++g_next;
    };
    auto ring_end = [&]() {
        if (lane == 0) { ring[g_next & (WG_RING - 1)] = WgChunk{0, 0, 0}; lds_total = g_cum; }
        ++g_next;
        g_end = true;
    };
    auto grab = [&]() {                                       // the next chunk behind a counter -> ring; END when every counter has run out
        // (every lost race is a chunk somebody else took: the bound cannot be reached while chunks are left; it makes the loop finite whatever memory says)
        for (int tries = 0; tries < 2 * n_pairs * chunks + 64 && !g_end; ++tries) {
            const int dyn = __builtin_amdgcn_readfirstlane(sch_dyn[g_pref]);
            if (dyn > 0) {
                unsigned idx = 0;
                if (lane == 0) idx = atomicAdd(a.counters + g_pref, 1u);
                idx = __builtin_amdgcn_readfirstlane(idx);
                if (idx < (unsigned)dyn) { ring_put(g_pref, __builtin_amdgcn_readfirstlane(sch_base[g_pref]) + (int)idx); return; }
            }
            // that pair is done: look at all counters, go where most is left
            int key = 0;
            if (lane < n_pairs) {
                const unsigned cnt = __hip_atomic_load(a.counters + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int left = sch_dyn[lane] - (int)(cnt < (unsigned)sch_dyn[lane] ? cnt : (unsigned)sch_dyn[lane]);
                key = left > 0 ? (left << 8) | lane : 0;
            }
#pragma unroll
            for (int o = 32; o; o >>= 1) { const int other = __shfl_xor(key, o); key = other > key ? other : key; }
            key = __builtin_amdgcn_readfirstlane(key);
            if (key == 0) { ring_end(); return; }
            g_pref = key & 255;
        }
        if (!g_end) ring_end();                              // (not reached)
    };
    if (wave == 8) {
        int hp = 0;
        for (int q = 0; q < n_pairs; ++q) hp = sch_fb[q] <= (int)blockIdx.x && (int)blockIdx.x < sch_fb[q + 1] ? q : hp;
        hp = __builtin_amdgcn_readfirstlane(hp);
        const int nb = __builtin_amdgcn_readfirstlane(sch_fb[hp + 1] - sch_fb[hp]), rank = (int)blockIdx.x - __builtin_amdgcn_readfirstlane(sch_fb[hp]);
        const int ns = nb > 0 ? __builtin_amdgcn_readfirstlane(sch_base[hp]) / nb : 0;
        g_pref = hp;
        for (int i = 0; i < ns; ++i) ring_put(hp, rank + i * nb);                   // (ns <= WG_LEAD)
        while (g_next < WG_LEAD && !g_end) grab();
    }
    __syncthreads();
    // a cursor: chunk number j, and inside it the pair, the stage, the stages left (0 = END; pair and stage then stay on the last real stage)
    auto cur_init = [&](WgCur& c, WgPair& q) {
        const WgChunk e = ring[0];
        c.j = 0;
        c.p = __builtin_amdgcn_readfirstlane(e.p); c.st = __builtin_amdgcn_readfirstlane(e.st); c.left = __builtin_amdgcn_readfirstlane(e.n);
        load_pair(q, c.p);
    };
    auto advance = [&](WgCur& c, WgPair& q) {
        if (c.left == 0) return;
        if (--c.left > 0) { ++c.st; return; }
        ++c.j;
        const WgChunk e = ring[c.j & (WG_RING - 1)];
        const int n = __builtin_amdgcn_readfirstlane(e.n);
        if (n > 0) {
            const int pr = __builtin_amdgcn_readfirstlane(e.p);
            if (pr != c.p) load_pair(q, pr);
            c.p = pr; c.st = __builtin_amdgcn_readfirstlane(e.st);
        }
        c.left = n;
    };
    // is this the cursor's last stage before another pair (or the end)?
    auto leaving = [&](const WgCur& c) {
        if (c.left != 1) return false;
        const WgChunk e = ring[(c.j + 1) & (WG_RING - 1)];
        return __builtin_amdgcn_readfirstlane(e.n) == 0 || __builtin_amdgcn_readfirstlane(e.p) != c.p;
    };
    // both roles run the stage indices 0 .. (total rounded up to even) - 1, one barrier each; the bound makes every loop finite whatever the ring says
    const int k_max = n_pairs * wg_stages + 2;
    auto total_even = [&]() { return (__builtin_amdgcn_readfirstlane(lds_total) + 1) & ~1; };

    if (producer) {
        struct Regs { w_u32x4 d[8]; };
        // global -> registers, and nothing else: rows / columns outside the matrix read a clamped (valid) piece that `convert` replaces by zero
        auto fetch_side = [&](auto XS, w_u32x4 (&d)[8], const WgPair& q, int stg) {
            constexpr bool xs = decltype(XS)::value;
            const int mode = xs ? q.xmode : WG_X24;
            // (the descriptor came through LDS: the compiler no longer knows that its pointers are global memory, and a FLAT load counts as an LDS
            //  access too -- every wait behind one becomes vmcnt(0) lgkmcnt(0).  Said explicitly.)
            typedef const char __attribute__((address_space(1))) * gptr_t;
            const gptr_t base = (gptr_t)(xs ? q.X : q.G);
            unsigned off[8];                                  // byte offsets from the uniform base (wgrad_add checks that they fit 32 bits)
            if (xs && mode == WG_XTILES) {
                const int s0 = stg * WG_STAGE + 32 * ttile;
                const unsigned tl = (unsigned)((s0 < wg_m ? s0 : wg_m - 1) >> 5), c = (unsigned)(tcol < q.w ? tcol : q.w - 1);
                const unsigned o0 = (tl * (unsigned)q.xtile_f + c) * 128u;
#pragma unroll
                for (int e = 0; e < 8; ++e) off[e] = o0 + 16u * e;
            } else {
                const unsigned ld = (unsigned)(xs ? q.ldx : q.ldg);
                const int width = xs ? q.w : q.nout;
                const unsigned piece = mode == WG_X24 ? 12u : 16u;
                const unsigned gc = piece * (unsigned)(4 * grp < width ? grp : (width - 1) >> 2);
                if ((stg + 1) * WG_STAGE <= wg_m && (!xs || q.xdiv == 1)) {             // (uniform) the usual stage: whole, one row per sample
                    const unsigned o0 = (unsigned)(stg * WG_STAGE + 8 * oct) * ld + gc;
#pragma unroll
                    for (int e = 0; e < 8; ++e) off[e] = o0 + (unsigned)e * ld;
                } else {
                    const unsigned one = !xs || q.xdiv == 1 ? 0xffffffffu : 0u;         // (uniform) s / 1 without a branch
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int s = stg * WG_STAGE + 8 * oct + e;
                        const unsigned sc = (unsigned)(s < wg_m ? s : wg_m - 1);
                        off[e] = ((sc & one) | ((__umulhi(sc, q.div_m) >> q.div_sh) & ~one)) * ld + gc;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                d[e] = *(const w_u32x4_a4 __attribute__((address_space(1)))*)(base + off[e]);
            }
        };
        w_f32x2 colsum[2] = {{0.f, 0.f}, {0.f, 0.f}};        // bias gradients of columns 4 grp .. 4 grp + 3: this thread's samples
        // registers -> (hi, lo) fragments -> LDS: every value is split exactly once
        auto put_frag = [&](char* Ol, const float (&x)[8], int f, int o) {
            const WFrag fr = w_split8(x);
            const int off = frag_off(f, o);
            *reinterpret_cast<w_bf16x8*>(Ol + off) = fr.hi;
            *reinterpret_cast<w_bf16x8*>(Ol + off + 4 * 64 * 16) = fr.lo;
        };
        // rows (24-bit or fp32 pieces): fragment c = column 4 grp + c of the thread's eight rows.  CLEAN (uniform: a whole stage of a 128-wide operand)
        // = nothing to zero; otherwise clamped rows / columns are zeroed with one AND per value (a mask per column times a mask per row)
        auto convert_rows = [&](auto XS, auto IS24, auto CLEAN, const w_u32x4 (&d)[8], int width, int stg, char* Ol, bool live) {
            constexpr bool xs = decltype(XS)::value, is24 = decltype(IS24)::value, clean = decltype(CLEAN)::value;
            const int s0 = stg * WG_STAGE + 8 * oct;
            unsigned cm[4], rm[8];
            if (!clean) {
#pragma unroll
                for (int c = 0; c < 4; ++c) cm[c] = (4 * grp + c < width) && live ? 0xffffffffu : 0u;
#pragma unroll
                for (int e = 0; e < 8; ++e) rm[e] = s0 + e < wg_m ? 0xffffffffu : 0u;
            }
#pragma unroll
            for (int cp = 0; cp < 4; cp += 2) {                // two columns at a time: their bias sums advance as one packed add per row
                float x[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        unsigned b = is24 ? __float_as_uint(p24_unpack1(d[e][0], d[e][1], d[e][2], cp + h)) : d[e][cp + h];      // (one byte permute per value, straight from the piece)
                        if (!clean) b &= cm[cp + h] & rm[e];
                        x[h][e] = __uint_as_float(b);
                    }
                if (!xs) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) colsum[cp >> 1] += (w_f32x2){x[0][e], x[1][e]};
                }
                put_frag(Ol, x[0], 4 * grp + cp, oct);
                put_frag(Ol, x[1], 4 * grp + cp + 1, oct);
            }
            // (the fourth dword of a 24-bit piece's load is never read: left dead, the register allocator hands it to some temporary while
            //  the load is still in flight, and the write to it waits for the load -- a stage's prefetch drained by a loop counter)
            if (is24) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(d[e][3]));
            }
        };
        auto convert_side = [&](auto XS, const w_u32x4 (&d)[8], const WgPair& q, int stg, char* Ol, bool live) {
            constexpr bool xs = decltype(XS)::value;
            const int mode = xs ? q.xmode : WG_X24;
            const int width = xs ? q.w : q.nout;
            const bool whole = (stg + 1) * WG_STAGE <= wg_m && width == 128 && live;     // (uniform)
            constexpr std::true_type yes{};
            constexpr std::false_type no{};
            if (xs && mode == WG_XTILES) {                    // fragment k = samples 8 k .. 8 k + 7 of this thread's tile: loads 2 k, 2 k + 1
                const unsigned cmask = tcol < width ? 0xffffffffu : 0u;
                const int s0 = stg * WG_STAGE + 32 * ttile;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = __uint_as_float(d[2 * k + (e >> 2)][e & 3] & (s0 + 8 * k + e < wg_m ? cmask : 0u));
                    put_frag(Ol, x, tcol, 4 * ttile + k);
                }
            } else if (xs && mode == WG_XROWS) {
                if (whole) convert_rows(XS, no, yes, d, width, stg, Ol, live);
                else convert_rows(XS, no, no, d, width, stg, Ol, live);
            } else {
                if (whole) convert_rows(XS, yes, yes, d, width, stg, Ol, live);
                else convert_rows(XS, yes, no, d, width, stg, Ol, live);
            }
        };
        auto flush_bias = [&](const WgPair& q) {
            float* const gb = q.gb;
            float* const gbh = q.gb_hi ? q.gb_hi : q.gb;
            const int split = q.split, nout = q.nout;
            const bool twob = q.gb_hi != nullptr;
            if (gb) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = 4 * grp + c;
                    const float v = colsum[c >> 1][c & 1];
                    if (col < nout && v != 0.f) {
                        const bool hi = twob && col >= split;
                        wg_atomic_add((hi ? gbh : gb) + (hi ? col - split : col), v);
                    }
                }
            }
            colsum[0] = colsum[1] = (w_f32x2){0.f, 0.f};
        };
        WgCur fc, cv;                                         // the stage fetched next, the stage converted next
        WgPair fq, cq;                                        // ... and their pairs
        cur_init(fc, fq);
        cur_init(cv, cq);
        // Two register sets, A and B, each a stage of this wave's operand: stage k + 2 is requested the moment stage k has left its set.  The loop's
        // trip is two stages and starts BETWEEN a conversion and the fetch that refills its set, where only the other set is in flight; every
        // wait is counted (the eight younger loads of the other set stay in flight).
        auto run = [&](auto XS) {
            constexpr bool xs = decltype(XS)::value;
            auto fetch = [&](Regs& r) {
                fetch_side(XS, r.d, fq, fc.st);
                advance(fc, fq);
            };
            auto convert = [&](const Regs& r, int img) {
                const WgPair& q = cq;
                const bool live = cv.left > 0;                // (past the end: a re-read of the last stage, converted to zeros, never multiplied)
                convert_side(XS, r.d, q, cv.st, wg_lds + img * WG_IMG_BYTES + (xs ? WG_OP_BYTES : 0), live);
                if (!xs && leaving(cv)) flush_bias(q);
                advance(cv, cq);
            };
            Regs A, B;
            fetch(A);
            fetch(B);
            convert(A, 0);
            WG_T(0)
            for (int k = 0; k < k_max && k < total_even(); k += WG_UNROLL) {
                fetch(A); wg_barrier(); convert(B, 1);
                fetch(B); wg_barrier(); convert(A, 0);
            }
        };
        // (the X producers are the block's first four waves: their stage is the longer one -- fetch + conversion 4.5 k cycles against the G side's
        //  2.9 k -- and the older waves' loads go first: 304 -> 294 us per 131 k samples against the other order)
        if (wave < 4) run(std::true_type{});
        else run(std::false_type{});
        return;
    }

    // ---- consumers: wave 8 + nt owns output rows 32 nt .. 32 nt + 31, all four column tiles
    const int nt = wave - 8;
    w_f32x16 acc[4];
    auto zero = [&]() {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    };
    // FULL: all four column tiles exist (w > 96): no branch between the fragment reads and the MFMAs of a k-step
    auto multiply_as = [&](auto FULL, const WgPair& q, int img) {
        constexpr bool full = decltype(FULL)::value;
        const char* const Gl = wg_lds + img * WG_IMG_BYTES;
        const char* const Xl = Gl + WG_OP_BYTES;
#pragma unroll
        for (int ks = 0; ks < WG_KS; ++ks) {
            const w_bf16x8* ga = reinterpret_cast<const w_bf16x8*>(Gl) + (ks * 2 * 4 + nt) * 64 + (lane & 32) + swz(nt, lane >> 5, lane & 31);
            const w_bf16x8 ahi = ga[0], alo = ga[4 * 64];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (full || 32 * kt < q.w) {
                    const w_bf16x8* xb = reinterpret_cast<const w_bf16x8*>(Xl) + (ks * 2 * 4 + kt) * 64 + (lane & 32) + swz(kt, lane >> 5, lane & 31);
                    const w_bf16x8 bhi = xb[0], blo = xb[4 * 64];
                    acc[kt] = WMFMA(ahi, bhi, acc[kt]);
                    acc[kt] = WMFMA(ahi, blo, acc[kt]);
                    acc[kt] = WMFMA(alo, bhi, acc[kt]);
                }
            }
        }
    };
    auto multiply = [&](const WgPair& q, int img) {
        if (32 * nt >= q.nout) return;                       // (wave-uniform) this wave's row tile does not exist
        if (q.w > 96) multiply_as(std::true_type{}, q, img);
        else multiply_as(std::false_type{}, q, img);
    };
    // leaving a pair: this block's partial sums go to the gradient (float atomics on 128-byte row segments)
    auto flush = [&](const WgPair& q) {
        const int i = lane & 31, hh = lane >> 5;
        // (descriptor fields into scalars first: a select between two FIELDS became a per-lane load of the chosen one, with a vmcnt(0) per atomic)
        float* const gW = q.gW;
        float* const gWh = q.gW_hi ? q.gW_hi : q.gW;
        const int split = q.split, ldw = q.ldw, nout = q.nout, w = q.w;
        const bool two = q.gW_hi != nullptr;
        if (nout == 128 && w == 128 && (!two || (split & 31) == 0)) {
            // a whole 128 x 128 product (two thirds of all flushes): 64 atomics back to back, no per-element test (each test was a branch
            // around its atomic); a wave's 32 rows lie on one side of a split that is a multiple of 32
            float* const row0 = (two && 32 * nt >= split ? gWh + (size_t)(32 * nt - split) * ldw : gW + (size_t)(32 * nt) * ldw) + (size_t)(4 * hh) * ldw + i;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) wg_atomic_add(row0 + (size_t)((r & 3) + 8 * (r >> 2)) * ldw + 32 * kt, acc[kt][r]);
        } else if (32 * nt < nout) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int k = 32 * kt + i;
                if (32 * kt < w) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (n < nout && k < w) {
                            const bool hi = two && n >= split;
                            float* base = hi ? gWh : gW;
                            wg_atomic_add(base + (size_t)(hi ? n - split : n) * ldw + k, acc[kt][r]);
                        }
                    }
                }
            }
        }
        zero();
    };
    zero();
    WgCur mu;                                                 // the stage multiplied next
    WgPair mq;
    cur_init(mu, mq);
    for (int k = 0; k < k_max && k < total_even(); k += WG_UNROLL) {
#pragma unroll
        for (int h = 0; h < WG_UNROLL; ++h) {
            if (wave == 8) while (!g_end && g_next < mu.j + 1 + WG_LEAD) grab();      // (before the barrier: visible to every wave from this stage index on)
            WG_T(0)
            wg_barrier();                                    // image h holds stage k + h; everyone is done with the other one
            WG_T(1)
            if (mu.left > 0) {
                multiply(mq, h);
                WG_T(2)
                if (leaving(mu)) flush(mq);
                WG_T(3)
            }
            advance(mu, mq);
        }
    }
}

// ---- host side: the pair list is built by ucnerf_mlp_bwd (mlp_bwd.hip)
void wgrad_begin(WgArgs* a, int m, unsigned* counters) { memset(a, 0, sizeof(*a)); a->m = m; a->stages = cdiv(m, WG_STAGE); a->counters = counters; }

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
    if (!a->counters) return fail(UCNERF_EINVAL, "mlp_bwd: the weight-gradient launch needs its chunk counters");
    // chunk length: 2 .. WG_CHUNK_MAX stages (the kernel reads it from the arguments)
    WgArgs args = *a;
    {
        // about FIVE chunks per block (round 5; eight before): a block owns its first four chunks without asking and every further one costs the
        // consumers' leading wave a round trip to a counter in global memory (~3 us) -- at 22 500 samples (a 250-ray shard: 20 stages per block)
        // eight chunks per block meant 2-stage chunks and six round trips: 106 us against 89 us with 4-stage chunks (in-kernel stamps,
        // profiles/r05_experiments.md; 3 chunks: 89, 4: 92, 2: 98).  Sizes that reach the 16-stage cap either way (>= 87 k samples) are unchanged.
        long long c = (long long)a->n_pairs * a->stages / (5ll * cus);
        args.chunk = (int)(c < 2 ? 2 : c > WG_CHUNK_MAX ? WG_CHUNK_MAX : c);
    }
    a = &args;
    long long units = (long long)a->n_pairs * cdiv(a->stages, a->chunk);
    if (hipMemsetAsync(a->counters, 0, WG_MAX_PAIRS * sizeof(unsigned), st) != hipSuccess) return fail(UCNERF_EHIP, "mlp_bwd: clearing the chunk counters failed");
    int blocks = cus;                                        // one block per CU (128 KB of LDS)
    if (blocks > units) blocks = (int)units;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_wgrad_kernel), WG_LDS_BYTES, "mlp_bwd wgrad")) return rc;
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(blocks), dim3(WG_THREADS), WG_LDS_BYTES, st, *a);
    return check_launch("mlp_bwd wgrad");
}

const char* build_flags_mlp_wgrad() { return "mlp_wgrad: (no compile-time switches) "; }


}  // namespace ucnerf
