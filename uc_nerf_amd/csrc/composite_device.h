// K7 device code shared by composite.hip (the stand-alone compositing kernels) and sample_pdf.hip (compositing of the coarse pass fused
// with the hierarchical re-sampling: one launch instead of two).  network/renderer.py:25-36,109-140 and utils/run_nerf_helpers.py:343-390.
#pragma once
#include "common.h"

namespace ucnerf {

__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    // inclusive scan over 64 lanes, then shift: lane l gets prod_{k<l} v_k
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d);
        if (lane >= d) v *= o;
    }
    float e = __shfl_up(v, 1);
    return lane == 0 ? 1.f : e;
}

__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    // lane l gets sum_{k>l} v_k
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_down(v, d);
        if (lane + d < 64) v += o;
    }
    float e = __shfl_down(v, 1);
    return lane == 63 ? 0.f : e;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

struct Sample {
    float r, g, b, alpha, f, ex;   // activated colour, alpha, transmittance factor (1 - alpha + 1e-10), exp(-sigma)
};

// loads sample i of ray `ray` and applies the variant's activations; i >= S gives a neutral sample
template <int VARIANT>
__device__ __forceinline__ Sample load_sample(const ucnerf_composite_params& p, int ray, int i, float dnorm) {
    Sample s;
    if (i >= p.S) { s.r = s.g = s.b = s.alpha = s.ex = 0.f; s.f = 1.f; return s; }
    const float4 raw = reinterpret_cast<const float4*>(p.raw)[(size_t)ray * p.S + i];
    if (VARIANT == 0) {                       // renderer.py:29: alpha = 1 - exp(-sigma)
        s.r = raw.x; s.g = raw.y; s.b = raw.z;
        s.ex = expf(-raw.w);
        s.alpha = 1.f - s.ex;
    } else {                                  // run_nerf_helpers.py:356-375
        const float* z = p.z + (size_t)ray * p.S;
        float dist = (i + 1 < p.S ? z[i + 1] - z[i] : 1e10f) * dnorm;
        float sg = raw.w + (p.noise ? p.noise[(size_t)ray * p.S + i] : 0.f);
        s.r = 1.f / (1.f + expf(-raw.x)); s.g = 1.f / (1.f + expf(-raw.y)); s.b = 1.f / (1.f + expf(-raw.z));
        s.ex = expf(-fmaxf(sg, 0.f) * dist);
        s.alpha = 1.f - s.ex;
    }
    s.f = 1.f - s.alpha + 1e-10f;
    return s;
}

// Forward of one ray by one 64-lane wave: lane l owns samples [l*E, (l+1)*E).  Writes the ray's outputs named in `p`; `w_keep` (optional,
// any address space through a generic pointer) receives the S weights as well -- the fused re-sampling reads them from LDS.
template <int E, int VARIANT>
__device__ __forceinline__ void composite_ray(const ucnerf_composite_params& p, int ray, int lane, float* w_keep) {
    float dnorm = 0.f;
    if (VARIANT == 1) {
        const float* d = p.rays_d + 3 * (size_t)ray;
        dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    }
    Sample sm[E];
    float T[E];
    float prod = 1.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        sm[e] = load_sample<VARIANT>(p, ray, lane * E + e, dnorm);
        T[e] = prod;
        prod *= sm[e].f;
    }
    const float pre = wave_excl_prod(prod, lane);
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f, su = 0.f;
    float w[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        w[e] = sm[e].alpha * (pre * T[e]);
        if (i < p.S) {
            sr += w[e] * sm[e].r; sg += w[e] * sm[e].g; sb += w[e] * sm[e].b;
            sd += w[e] * p.z[(size_t)ray * p.S + i];
            sa += w[e];
            if (p.wu) su += w[e] * p.u[(size_t)ray * p.S + i];      // composited uncertainty sum_i w_i u_i (u: network/models.py:149)
            if (p.weights) p.weights[(size_t)ray * p.S + i] = w[e];
            if (w_keep) w_keep[i] = w[e];
        }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
    if (p.wu) su = wave_sum(su);
    float var = 0.f;
    if (p.var) {                                // torch.var_mean(weights, dim=1): unbiased
        const float mean = sa / (float)p.S;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (lane * E + e < p.S) { float d = w[e] - mean; q += d * d; }
        var = wave_sum(q) / (float)(p.S - 1);
    }
    if (lane == 0) {
        if (p.white_bkgd) { sr += 1.f - sa; sg += 1.f - sa; sb += 1.f - sa; }
        p.rgb_map[3 * (size_t)ray] = sr; p.rgb_map[3 * (size_t)ray + 1] = sg; p.rgb_map[3 * (size_t)ray + 2] = sb;
        p.depth_map[ray] = sd;
        if (p.acc_map) p.acc_map[ray] = sa;
        if (p.disp_map) {                       // 1 / max(1e-10, depth/acc); NaN (0/0) propagates as in torch.max
            float q = sd / sa;
            p.disp_map[ray] = q != q ? q : 1.f / fmaxf(1e-10f, q);
        }
        if (p.var) p.var[ray] = var;
        if (p.wu) p.wu[ray] = su;
    }
}

// E (samples per lane) the launchers instantiate for S samples per ray: the association order of the transmittance products depends on it,
// so every kernel that composites a ray of S samples must pick the same one
__host__ __device__ inline int composite_lane_samples(int S) {
    const int e = (S + 63) / 64;
    return e <= 4 ? (e < 1 ? 1 : e) : e <= 8 ? 8 : 16;
}

}  // namespace ucnerf
