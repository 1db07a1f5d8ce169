// K8 + K9: inverse-CDF sampling with bit-exact searchsorted indices, and the sorted merge of the fine
// depths with the coarse ones (data/ray_utils.py:98-141,219 of the reference).
//
// One 64-lane wave (= one block) per ray; everything lives in LDS.  To make
// inds == torch.searchsorted(cdf, u, right=True) bit for bit the cdf must be bit-identical to torch-CPU's:
//   * sum(weights + 1e-5) follows ATen's cascade_sum for a contiguous float32 row (8-lane vectors, four
//     interleaved partial accumulators with cascade levels, tail first, lanes folded left to right);
//   * pdf = w / sum with IEEE division;
//   * cumsum accumulates sequentially in float64 and rounds every output to float32.
// This file must be compiled with -ffp-contract=off.
#include "common.h"
#include "composite_device.h"

namespace ucnerf {

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
template <int MAXB, int MAXS>
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
    __syncthreads();

    // ---- torch.sum(weights, -1)
    if (n < 8) {
        if (lane == 0) total = aten_row_sum(n, [&](int i) { return w[i]; });
    } else {
        const int nv = n / 8;
        if (lane < 8) lane_part[lane] = aten_row_sum(nv, [&](int i) { return w[8 * i + lane]; });
        __syncthreads();
        if (lane == 0) {
            float acc = 0.f;
            for (int k = nv * 8; k < n; ++k) acc += w[k];
            for (int k = 0; k < 8; ++k) acc += lane_part[k];
            total = acc;
        }
    }
    __syncthreads();
    const float tot = total;
    for (int i = lane; i < n; i += 64) w[i] = w[i] / tot;
    __syncthreads();

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
    __syncthreads();
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
        __syncthreads();
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

template <int MAXB, int MAXS>
__global__ void __launch_bounds__(64) sample_pdf_kernel(ucnerf_sample_pdf_params p) {
    __shared__ PdfShared<MAXB, MAXS> sh;
    const int ray = blockIdx.x, lane = threadIdx.x;
    sample_pdf_ray<MAXB, MAXS>(p, ray, lane, sh, p.from_coarse ? p.weights + (size_t)ray * p.n_merge : nullptr,
                               p.from_coarse ? p.z_merge + (size_t)ray * p.n_merge : nullptr);
}

// K7 of the coarse pass + K8 + K9 in ONE launch (both were one wave per ray already): the wave composites its ray (composite_device.h, the
// arithmetic and lane split of composite_fwd_kernel: same weights bit for bit), leaves the S weights in LDS and re-samples from them.
template <int MAXB, int MAXS, int E>
__global__ void __launch_bounds__(64) composite_sample_pdf_kernel(ucnerf_composite_params c, ucnerf_sample_pdf_params p) {
    __shared__ PdfShared<MAXB, MAXS> sh;
    __shared__ float wl[MAXB + 2];
    const int ray = blockIdx.x, lane = threadIdx.x;
    composite_ray<E, 0>(c, ray, lane, wl);
    __syncthreads();
    sample_pdf_ray<MAXB, MAXS>(p, ray, lane, sh, wl, c.z + (size_t)ray * c.S);
}

__global__ void merge_rows_kernel(ucnerf_merge_rows_params p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int tot = p.na + p.nb;
    if (i >= (long long)p.n * tot) return;
    const int ray = (int)(i / tot), e = (int)(i - (long long)ray * tot);
    const float* src = e < p.na ? p.a + ((size_t)ray * p.na + e) * p.width : p.b + ((size_t)ray * p.nb + (e - p.na)) * p.width;
    float* dst = p.out + ((size_t)ray * tot + p.rank[i]) * p.width;
    if (p.width == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    else for (int k = 0; k < p.width; ++k) dst[k] = src[k];
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" int ucnerf_sample_pdf(const ucnerf_sample_pdf_params* p, void* stream) {
    UCNERF_REQUIRE(p, "sample_pdf: null params");
    if (p->n <= 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->weights && p->u, "sample_pdf: null pointer");
    if (p->from_coarse)
        UCNERF_REQUIRE(p->z_merge && p->n_merge == p->n_bins + 1, "sample_pdf: from_coarse needs z_merge with n_merge == n_bins + 1");
    else
        UCNERF_REQUIRE(p->bins, "sample_pdf: null bins");
    UCNERF_REQUIRE(p->n_bins >= 2 && p->n_bins <= PDF_MAX_BINS, "sample_pdf: n_bins = %d outside 2..%d", p->n_bins, PDF_MAX_BINS);
    UCNERF_REQUIRE(p->n_samples >= 1 && p->n_samples <= 1024, "sample_pdf: n_samples = %d outside 1..1024", p->n_samples);
    UCNERF_REQUIRE(p->u_stride == 0 || p->u_stride == p->n_samples, "sample_pdf: u_stride must be 0 or n_samples");
    UCNERF_REQUIRE(p->n_merge >= 0 && p->n_merge + p->n_samples <= PDF_MAX_SORT, "sample_pdf: n_merge + n_samples > %d", PDF_MAX_SORT);
    UCNERF_REQUIRE(!p->z_sorted || p->n_merge == 0 || p->z_merge, "sample_pdf: z_sorted with n_merge > 0 needs z_merge");
    UCNERF_REQUIRE(p->samples || p->inds || p->cdf || p->z_sorted, "sample_pdf: no outputs requested");
    UCNERF_REQUIRE(!p->merge_rank || p->z_sorted, "sample_pdf: merge_rank needs z_sorted");
    if (p->n <= 0) return UCNERF_OK;
    if (p->n_bins <= 128 && p->n_merge + p->n_samples <= 512)
        hipLaunchKernelGGL((sample_pdf_kernel<128, 512>), dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    else
        hipLaunchKernelGGL((sample_pdf_kernel<PDF_MAX_BINS, PDF_MAX_SORT>), dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    return check_launch("sample_pdf");
}

extern "C" int ucnerf_composite_sample_pdf(const ucnerf_composite_params* c, const ucnerf_sample_pdf_params* s, void* stream) {
    UCNERF_REQUIRE(c && s, "composite_sample_pdf: null params");
    if (c->n <= 0) return UCNERF_OK;
    UCNERF_REQUIRE(c->raw && c->z && c->rgb_map && c->depth_map, "composite_sample_pdf: null compositing pointer");
    UCNERF_REQUIRE(c->variant == 0 && !c->u && !c->wu, "composite_sample_pdf: the live compositing variant without uncertainty inputs");
    UCNERF_REQUIRE(!c->var || c->S >= 2, "composite_sample_pdf: var needs S >= 2");
    UCNERF_REQUIRE(((uintptr_t)c->raw & 15) == 0, "composite_sample_pdf: raw must be 16-byte aligned");
    UCNERF_REQUIRE(s->from_coarse && s->n == c->n && s->n_merge == c->S && s->n_bins == c->S - 1 && c->S >= 3 && c->S <= PDF_MAX_BINS + 1,
                   "composite_sample_pdf: the re-sampling must be the from_coarse form over the composited pass (n %d / %d, n_merge %d, n_bins %d, S %d)",
                   s->n, c->n, s->n_merge, s->n_bins, c->S);
    UCNERF_REQUIRE(!s->z_merge || s->z_merge == c->z, "composite_sample_pdf: z_merge, when given, must be the composited pass's depths");
    UCNERF_REQUIRE(s->u, "composite_sample_pdf: null draws");
    UCNERF_REQUIRE(s->n_samples >= 1 && s->n_samples <= 1024, "composite_sample_pdf: n_samples = %d outside 1..1024", s->n_samples);
    UCNERF_REQUIRE(s->u_stride == 0 || s->u_stride == s->n_samples, "composite_sample_pdf: u_stride must be 0 or n_samples");
    UCNERF_REQUIRE(s->n_merge + s->n_samples <= PDF_MAX_SORT, "composite_sample_pdf: n_merge + n_samples > %d", PDF_MAX_SORT);
    UCNERF_REQUIRE(s->samples || s->inds || s->cdf || s->z_sorted, "composite_sample_pdf: no re-sampling outputs requested");
    UCNERF_REQUIRE(!s->merge_rank || s->z_sorted, "composite_sample_pdf: merge_rank needs z_sorted");
    hipStream_t st = (hipStream_t)stream;
    const int E = composite_lane_samples(c->S);
    const bool small = s->n_bins <= 128 && s->n_merge + s->n_samples <= 512;
#define UCNERF_CSP(MB, MS, EE) hipLaunchKernelGGL((composite_sample_pdf_kernel<MB, MS, EE>), dim3(c->n), dim3(64), 0, st, *c, *s)
    if (small) {
        if (E == 1) UCNERF_CSP(128, 512, 1); else if (E == 2) UCNERF_CSP(128, 512, 2); else UCNERF_CSP(128, 512, 3);
    } else {
        if (E == 1) UCNERF_CSP(PDF_MAX_BINS, PDF_MAX_SORT, 1); else if (E == 2) UCNERF_CSP(PDF_MAX_BINS, PDF_MAX_SORT, 2);
        else if (E == 3) UCNERF_CSP(PDF_MAX_BINS, PDF_MAX_SORT, 3); else if (E == 4) UCNERF_CSP(PDF_MAX_BINS, PDF_MAX_SORT, 4);
        else if (E == 8) UCNERF_CSP(PDF_MAX_BINS, PDF_MAX_SORT, 8); else UCNERF_CSP(PDF_MAX_BINS, PDF_MAX_SORT, 16);
    }
#undef UCNERF_CSP
    return check_launch("composite_sample_pdf");
}

extern "C" int ucnerf_merge_rows(const ucnerf_merge_rows_params* p, void* stream) {
    UCNERF_REQUIRE(p, "merge_rows: null params");
    if (p->n <= 0 || p->na + p->nb <= 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->rank && p->out && (p->a || p->na == 0) && (p->b || p->nb == 0), "merge_rows: null pointer");
    UCNERF_REQUIRE(p->width >= 1 && p->width <= 8, "merge_rows: width = %d outside 1..8", p->width);
    UCNERF_REQUIRE(p->width != 4 || ((((uintptr_t)p->a | (uintptr_t)p->b | (uintptr_t)p->out) & 15) == 0), "merge_rows: 4-float rows must be 16-byte aligned");
    const long long tot = (long long)p->n * (p->na + p->nb);
    hipLaunchKernelGGL(merge_rows_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("merge_rows");
}
