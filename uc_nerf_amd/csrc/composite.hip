// K7: alpha compositing along rays (network/renderer.py:25-36,109-140 and the nerf-pytorch variant
// utils/run_nerf_helpers.py:343-390 of the reference), forward and backward.
//
// One 64-lane wave per ray.  Lane l owns E consecutive samples [l*E, (l+1)*E): it multiplies its own
// transmittance factors sequentially, the wave combines the 64 lane products with a shuffle scan
// (inclusive product scan, shifted by one lane), and the ray sums are shuffle reductions.  HBM-bound:
// 20 B in + (4..8) B out per sample.
#include "common.h"
#include "composite_device.h"

namespace ucnerf {

template <int E, int VARIANT>
__global__ void __launch_bounds__(256) composite_fwd_kernel(ucnerf_composite_params p) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= p.n) return;
    composite_ray<E, VARIANT>(p, ray, lane, nullptr);
}

// Backward of the live variant.  With gw_i = dL/dw_i:
//   dL/dalpha_i = gw_i T_i - (sum_{k>i} gw_k w_k) / f_i,   dL/dsigma_i = dL/dalpha_i * exp(-sigma_i),
//   dL/drgb_i = w_i * g_rgb.
template <int E>
__global__ void __launch_bounds__(256) composite_bwd_kernel(ucnerf_composite_bwd_params bp) {
    const ucnerf_composite_params& p = bp.fwd;
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= p.n) return;
    float gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f, ga = 0.f;
    if (bp.g_rgb) { gr = bp.g_rgb[3 * (size_t)ray]; gg = bp.g_rgb[3 * (size_t)ray + 1]; gb = bp.g_rgb[3 * (size_t)ray + 2]; }
    if (bp.g_depth) gd = bp.g_depth[ray];
    if (bp.g_acc) ga = bp.g_acc[ray];
    if (p.white_bkgd) ga -= gr + gg + gb;       // rgb_map += 1 - acc
    Sample sm[E];
    float T[E], em[E];
    float prod = 1.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        sm[e] = load_sample<0>(p, ray, i, 0.f);
        em[e] = sm[e].ex;                       // exp(-sigma) = d(alpha)/d(sigma)
        T[e] = prod;
        prod *= sm[e].f;
    }
    const float pre = wave_excl_prod(prod, lane);
    float gw[E], gww[E];
    float local = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        T[e] *= pre;
        gw[e] = 0.f; gww[e] = 0.f;
        if (i < p.S) {
            gw[e] = gr * sm[e].r + gg * sm[e].g + gb * sm[e].b + gd * p.z[(size_t)ray * p.S + i] + ga;
            if (bp.g_weights) gw[e] += bp.g_weights[(size_t)ray * p.S + i];
            gww[e] = gw[e] * sm[e].alpha * T[e];
            local += gww[e];
        }
    }
    float suffix = wave_excl_suffix_sum(local, lane);   // contributions of higher lanes
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const int i = lane * E + e;
        if (i < p.S) {
            const float w = sm[e].alpha * T[e];
            const float galpha = gw[e] * T[e] - suffix / sm[e].f;
            float4 o;
            o.x = w * gr; o.y = w * gg; o.z = w * gb;
            o.w = galpha * em[e];
            reinterpret_cast<float4*>(bp.g_raw)[(size_t)ray * p.S + i] = o;
        }
        suffix += gww[e];
    }
}

template <int E>
static void launch_fwd(const ucnerf_composite_params& p, hipStream_t st) {
    dim3 grid(cdiv(p.n, 4)), block(256);
    if (p.variant == 0) hipLaunchKernelGGL((composite_fwd_kernel<E, 0>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((composite_fwd_kernel<E, 1>), grid, block, 0, st, p);
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int ucnerf_composite_fwd(const ucnerf_composite_params* p, void* stream) {
    UCNERF_REQUIRE(p, "composite_fwd: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->raw && p->z && p->rgb_map && p->depth_map, "composite_fwd: null pointer");
    UCNERF_REQUIRE(p->S >= 1 && p->S <= 1024, "composite_fwd: S = %d outside 1..1024", p->S);
    UCNERF_REQUIRE(p->variant == 0 || (p->variant == 1 && p->rays_d), "composite_fwd: variant %d (variant 1 needs rays_d)", p->variant);
    UCNERF_REQUIRE(!p->var || (p->variant == 0 && p->S >= 2), "composite_fwd: var needs the live variant and S >= 2");
    UCNERF_REQUIRE(((uintptr_t)p->raw & 15) == 0, "composite_fwd: raw must be 16-byte aligned");
    UCNERF_REQUIRE(!p->wu || p->u, "composite_fwd: wu (sum of w*u) needs the per-sample uncertainty u");
    UCNERF_COUNT(p->n);
    hipStream_t st = (hipStream_t)stream;
    const int E = composite_lane_samples(p->S);      // (shared with the launch fused with the re-sampling: same lane split, same weights)
    if (E <= 1) launch_fwd<1>(*p, st);
    else if (E <= 2) launch_fwd<2>(*p, st);
    else if (E <= 3) launch_fwd<3>(*p, st);
    else if (E <= 4) launch_fwd<4>(*p, st);
    else if (E <= 8) launch_fwd<8>(*p, st);
    else launch_fwd<16>(*p, st);
    return check_launch("composite_fwd");
}

int ucnerf_composite_bwd(const ucnerf_composite_bwd_params* bp, void* stream) {
    UCNERF_REQUIRE(bp, "composite_bwd: null params");
    UCNERF_COUNT(bp->fwd.n);
    UCNERF_REQUIRE(bp->fwd.raw && bp->fwd.z && bp->g_raw, "composite_bwd: null pointer");
    const ucnerf_composite_params& p = bp->fwd;
    UCNERF_REQUIRE(p.variant == 0, "composite_bwd: only the live variant (network/renderer.py) has a backward");
    UCNERF_REQUIRE(p.S >= 1 && p.S <= 1024, "composite_bwd: S = %d outside 1..1024", p.S);
    UCNERF_REQUIRE(((uintptr_t)p.raw & 15) == 0 && ((uintptr_t)bp->g_raw & 15) == 0, "composite_bwd: raw/g_raw must be 16-byte aligned");
    UCNERF_COUNT(p.n);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(p.n, 4)), block(256);
    const int E = cdiv(p.S, 64);
    if (E <= 1) hipLaunchKernelGGL(composite_bwd_kernel<1>, grid, block, 0, st, *bp);
    else if (E <= 2) hipLaunchKernelGGL(composite_bwd_kernel<2>, grid, block, 0, st, *bp);
    else if (E <= 3) hipLaunchKernelGGL(composite_bwd_kernel<3>, grid, block, 0, st, *bp);
    else if (E <= 4) hipLaunchKernelGGL(composite_bwd_kernel<4>, grid, block, 0, st, *bp);
    else if (E <= 8) hipLaunchKernelGGL(composite_bwd_kernel<8>, grid, block, 0, st, *bp);
    else hipLaunchKernelGGL(composite_bwd_kernel<16>, grid, block, 0, st, *bp);
    return check_launch("composite_bwd");
}

}  // extern "C"
