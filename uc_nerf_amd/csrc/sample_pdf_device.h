// K8 + K9 device code (data/ray_utils.py:98-141,219 of the reference): inverse-CDF sampling of one ray by one 64-lane wave with bit-exact
// searchsorted indices, and the sorted merge of the new depths with the coarse ones.  Shared by sample_pdf.hip (one wave = one block per ray)
// and by the gather-fused forward launch when it composites and re-samples a small pass's rays in its own tail (mlp_bf16.hip, TAIL).
// Every translation unit that includes this must be compiled with -ffp-contract=off (uc_nerf_amd/build.py: all are).
#pragma once
#include "common.h"

namespace ucnerf {

// between the phases of a ray: a block barrier where the wave IS the block; inside a larger block (WAVE_ONLY) the wave waits for its own LDS
// traffic -- the LDS operations of one wave execute in order, and no other wave touches its arrays
template <bool WAVE_ONLY>
__device__ __forceinline__ void pdf_sync() {
    if (WAVE_ONLY) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

constexpr int PDF_MAX_BINS = 1024;
constexpr int PDF_MAX_SORT = 2048;

// ATen multi_row_sum + row_sum over `n` items item(i), i < n, for ONE vector lane.
template <class Item>
__device__ float aten_row_sum(int n, Item item) {
    const int n_groups = n / 4;
    int ceil_log2 = 0;
    while ((1 << ceil_log2) < n_groups) ++ceil_log2;
    const int power = ceil_log2 / 4 > 4 ? ceil_log2 / 4 : 4;
    const int step = 1 << power, mask = step - 1;
    float acc[4][4];
    for (int a = 0; a < 4; ++a) for (int k = 0; k < 4; ++k) acc[a][k] = 0.f;
    int i = 0;
    while (i + step <= n_groups) {
        for (int s = 0; s < step; ++s, ++i)
            for (int k = 0; k < 4; ++k) acc[0][k] += item(4 * i + k);
        for (int lv = 1; lv < 4; ++lv) {
            for (int k = 0; k < 4; ++k) { acc[lv][k] += acc[lv - 1][k]; acc[lv - 1][k] = 0.f; }
            if (i & (mask << (lv * power))) break;
        }
    }
    for (; i < n_groups; ++i)
        for (int k = 0; k < 4; ++k) acc[0][k] += item(4 * i + k);
    for (int lv = 1; lv < 4; ++lv)
        for (int k = 0; k < 4; ++k) acc[0][k] += acc[lv][k];
    for (int r = n_groups * 4; r < n; ++r) acc[0][0] += item(r);
    acc[0][0] += acc[0][1];
    acc[0][0] += acc[0][2];
    acc[0][0] += acc[0][3];
    return acc[0][0];
}

// MAXB / MAXS size the per-ray LDS arrays: the common shapes (<= 128 bins, <= 512 merged depths) take 3.5 KB per ray, so
// that a CU holds 32 rays at once; with the full-size arrays (20 KB) it holds 8 and 4096 rays need two rounds.
template <int MAXB, int MAXS>
struct PdfShared {
    float w[MAXB];               // weights + 1e-5, then pdf
    float cdf[MAXB];
    float srt[MAXS];
    float bins[MAXB];
    float lane_part[8];
    float total;
};

// One ray by one 64-lane wave (= one block).  from_coarse: `w_coarse` / `z_coarse` are the ray's S coarse weights and depths -- rows of
// p.weights / p.z_merge, or (the launch fused with the coarse pass's compositing) the weights the wave has just left in LDS.
template <int MAXB, int MAXS, bool WAVE_ONLY = false>
__device__ __forceinline__ void sample_pdf_ray(const ucnerf_sample_pdf_params& p, int ray, int lane, PdfShared<MAXB, MAXS>& sh,
                                               const float* w_coarse, const float* z_coarse) {
    float* const w = sh.w; float* const cdf = sh.cdf; float* const srt = sh.srt; float* const bins = sh.bins; float* const lane_part = sh.lane_part;
    float& total = sh.total;
    const int L = p.n_bins, n = L - 1, M = p.n_samples;
    if (p.from_coarse) {        // bins = mid-points of the coarse depths, weights = w[1:-1]   (data/ray_utils.py:216-217)
        const float* zc = z_coarse;
        const float* wr = w_coarse + 1;
        for (int i = lane; i < L; i += 64) bins[i] = .5f * (zc[i] + zc[i + 1]);
        for (int i = lane; i < n; i += 64) w[i] = wr[i] + 1e-5f;
    } else {
        const float* wr = p.weights + (size_t)ray * n;
        const float* br = p.bins + (size_t)ray * L;
        for (int i = lane; i < L; i += 64) bins[i] = br[i];
        for (int i = lane; i < n; i += 64) w[i] = wr[i] + 1e-5f;
    }
    pdf_sync<WAVE_ONLY>();

    // ---- torch.sum(weights, -1)
    if (n < 8) {
        if (lane == 0) total = aten_row_sum(n, [&](int i) { return w[i]; });
    } else {
        const int nv = n / 8;
        if (lane < 8) lane_part[lane] = aten_row_sum(nv, [&](int i) { return w[8 * i + lane]; });
        pdf_sync<WAVE_ONLY>();
        if (lane == 0) {
            float acc = 0.f;
            for (int k = nv * 8; k < n; ++k) acc += w[k];
            for (int k = 0; k < 8; ++k) acc += lane_part[k];
            total = acc;
        }
    }
    pdf_sync<WAVE_ONLY>();
    const float tot = total;
    for (int i = lane; i < n; i += 64) w[i] = w[i] / tot;
    pdf_sync<WAVE_ONLY>();

    // ---- cdf = [0, cumsum(pdf)] with a float64 running sum.
    // The float64 sums are EXACT whenever every pdf value is 0 or in [2^-28, 1] (each is then a multiple of 2^-51 and
    // the partial sums stay below 2: 52 significant bits), so a wave-parallel scan gives the very doubles of torch's
    // sequential loop; compositing weights (+1e-5, <= 1023 bins) always satisfy this.  Anything else (negative or
    // huge weights, NaN) takes the sequential loop.
    {
        bool ok = true;
        for (int i = lane; i < n; i += 64) { const float v = w[i]; ok = ok && (v == 0.f || (v >= 3.7252902984619140625e-9f && v <= 1.f)); }
        if (__all(ok)) {
            double carry = 0.0;
            if (lane == 0) cdf[0] = 0.f;
            for (int c0 = 0; c0 < n; c0 += 64) {
                const int i = c0 + lane;
                double v = i < n ? (double)w[i] : 0.0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const double t = __shfl_up(v, d);
                    if (lane >= d) v += t;
                }
                if (i < n) cdf[i + 1] = (float)(carry + v);
                carry += __shfl(v, 63);
            }
        } else if (lane == 0) {
            double run = 0.0;
            cdf[0] = 0.f;
            for (int i = 0; i < n; ++i) { run += (double)w[i]; cdf[i + 1] = (float)run; }
        }
    }
    pdf_sync<WAVE_ONLY>();
    if (p.cdf) for (int i = lane; i < L; i += 64) p.cdf[(size_t)ray * L + i] = cdf[i];

    // ---- invert
    const float* ur = p.u + (size_t)ray * p.u_stride;
    for (int m = lane; m < M; m += 64) {
        const float u = ur[m];
        int lo = 0, hi = L;                          // first index with cdf[idx] > u  (right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < L - 1 ? lo : L - 1;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = bins[below], b1 = bins[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u - c0) / denom;
        const float smp = b0 + t * (b1 - b0);
        if (p.samples) p.samples[(size_t)ray * M + m] = smp;
        if (p.inds) p.inds[(size_t)ray * M + m] = (int64_t)lo;
        if (p.z_sorted) srt[m] = smp;
    }

    // ---- sort(cat(samples, z_merge)).  Only values are returned, so any tie order is right.
    // Both lists are normally sorted already (monotone u, increasing coarse depths): an element's rank is then its own
    // index plus a binary search in the other list.  Otherwise (random u) rank by counting:
    // rank_i = #{x_j < x_i} + #{x_j == x_i, j < i} -- a permutation even with ties.
    if (p.z_sorted) {
        const int tot_n = M + p.n_merge;
        const float* zm = p.from_coarse ? z_coarse : p.z_merge + (size_t)ray * p.n_merge;
        for (int i = lane; i < p.n_merge; i += 64) srt[M + i] = zm[i];
        pdf_sync<WAVE_ONLY>();
        float* dst = p.z_sorted + (size_t)ray * tot_n;
        bool sorted = true;
        for (int i = lane; i < tot_n - 1; i += 64) sorted = sorted && (i == M - 1 || srt[i] <= srt[i + 1]);
        if (__all(sorted)) {
            for (int i = lane; i < tot_n; i += 64) {
                const float x = srt[i];
                int lo, hi, r;
                if (i < M) {                     // samples come first in the concatenation: count coarse depths < x
                    lo = M; hi = tot_n;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (srt[mid] < x) lo = mid + 1; else hi = mid; }
                    r = i + (lo - M);
                } else {                         // count samples <= x
                    lo = 0; hi = M;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (srt[mid] <= x) lo = mid + 1; else hi = mid; }
                    r = (i - M) + lo;
                }
                dst[r] = x;
                if (p.merge_rank) p.merge_rank[(size_t)ray * tot_n + i] = r;
            }
        } else {
            for (int i0 = 0; i0 < tot_n; i0 += 256) {            // 4 elements per lane per sweep
                float x[4];
                int rank[4], id[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    id[e] = i0 + e * 64 + lane;
                    x[e] = id[e] < tot_n ? srt[id[e]] : 0.f;
                    rank[e] = 0;
                }
                for (int jj = 0; jj < tot_n; ++jj) {
                    const float v = srt[jj];
#pragma unroll
                    for (int e = 0; e < 4; ++e) rank[e] += (v < x[e]) || (v == x[e] && jj < id[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (id[e] < tot_n) {
                        dst[rank[e]] = x[e];
                        if (p.merge_rank) p.merge_rank[(size_t)ray * tot_n + id[e]] = rank[e];
                    }
            }
        }
    }
}

}  // namespace ucnerf
