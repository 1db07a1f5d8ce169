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
#include "sample_pdf_device.h"

namespace ucnerf {

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
    UCNERF_COUNT(p->n);
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
    UCNERF_COUNT(p->n);
    if (p->n_bins <= 128 && p->n_merge + p->n_samples <= 512)
        hipLaunchKernelGGL((sample_pdf_kernel<128, 512>), dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    else
        hipLaunchKernelGGL((sample_pdf_kernel<PDF_MAX_BINS, PDF_MAX_SORT>), dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    return check_launch("sample_pdf");
}

extern "C" int ucnerf_composite_sample_pdf(const ucnerf_composite_params* c, const ucnerf_sample_pdf_params* s, void* stream) {
    UCNERF_REQUIRE(c && s, "composite_sample_pdf: null params");
    UCNERF_COUNT(c->n);
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
    UCNERF_REQUIRE(p->na >= 0 && p->nb >= 0, "merge_rows: negative row length");
    UCNERF_COUNT(p->n);
    UCNERF_COUNT(p->na + p->nb);
    UCNERF_REQUIRE(p->rank && p->out && (p->a || p->na == 0) && (p->b || p->nb == 0), "merge_rows: null pointer");
    UCNERF_REQUIRE(p->width >= 1 && p->width <= 8, "merge_rows: width = %d outside 1..8", p->width);
    UCNERF_REQUIRE(p->width != 4 || ((((uintptr_t)p->a | (uintptr_t)p->b | (uintptr_t)p->out) & 15) == 0), "merge_rows: 4-float rows must be 16-byte aligned");
    const long long tot = (long long)p->n * (p->na + p->nb);
    hipLaunchKernelGGL(merge_rows_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("merge_rows");
}
