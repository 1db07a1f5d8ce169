// sin/cos of r for the positional encodings (r = x * 2^k, an exact float product).
//
// |r| <= 2^16: three-step Cody-Waite reduction by pi/2 with FMAs (the split constants below sum to pi/2 to
// ~75 bits, so the reduced argument is good to ~1e-7 absolute for the quadrant counts that occur) followed by the
// classic degree-7/8 minimax polynomials on [-pi/4, pi/4]: ~25 VALU ops instead of OCML's ~150 (which inlines the
// Payne-Hanek path branch-free).  Larger arguments (points far outside the frustum) take OCML's sincosf.
// Measured against torch-CPU's libm on arguments to +-3e5 rad: within 2e-7 absolute (tests: G5).
#pragma once
#include <hip/hip_runtime.h>

namespace ucnerf {

constexpr float SINCOS_FAST_MAX = 65536.0f;

// branch-free part: valid for |r| <= SINCOS_FAST_MAX
__device__ __forceinline__ void sincos_pe_fast(float r, float* s_out, float* c_out) {
    const float qf = rintf(r * 0.63661977236758134f);            // r * 2/pi
    float y = fmaf(qf, -1.57079637050628662109375f, r);          // pi/2 split in three floats
    y = fmaf(qf, 4.37113900018624283e-8f, y);
    y = fmaf(qf, 1.71512451e-15f, y);
    const int q = (int)qf;
    const float z = y * y;
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, z, -1.6666654611e-1f);
    const float sn = fmaf(sp * z, y, y);                          // sin(y)
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, z, 4.166664568298827e-2f);
    const float cs = fmaf(cp * z, z, fmaf(z, -0.5f, 1.0f));       // cos(y)
    const bool swap = q & 1;
    float s = swap ? cs : sn, c = swap ? sn : cs;
    s = (q & 2) ? -s : s;
    c = ((q + 1) & 2) ? -c : c;
    *s_out = s;
    *c_out = c;
}

__device__ __forceinline__ void sincos_pe(float r, float* s_out, float* c_out) {
    if (fabsf(r) > SINCOS_FAST_MAX) {          // rare: wave-divergent slow path
        sincosf(r, s_out, c_out);
        return;
    }
    sincos_pe_fast(r, s_out, c_out);
}

// ---- hardware path used by the MLP kernels: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, so the range
// reduction is one v_fract_f32 -- exact -- provided x / (2 pi) is carried as an unevaluated two-float sum (the residual of
// the product and the low part of 1 / (2 pi) ride in `lo`).  2^k * hi is exact, fract() of it is exact, and the low part is
// added after the reduction: measured max error against float64 over |x| <= 1.3, k = 0..9: 4.2e-7 (scripts/micro/
// hw_sincos.hip) -- the accuracy of the 25-instruction Cody-Waite path above for 5 instructions per (sin, cos) pair, with
// no argument-size limit to branch on.
struct Rev2 { float hi, lo; };

__device__ __forceinline__ Rev2 to_revolutions(float x) {
    const float chi = 0.15915494309189535f;                                   // fl(1 / (2 pi))
    const float clo = 6.4206383e-09f;                                         // 1 / (2 pi) - chi
    Rev2 t;
    t.hi = x * chi;
    t.lo = fmaf(x, clo, fmaf(x, chi, -t.hi));
    return t;
}

__device__ __forceinline__ void sincos_rev(float hi, float lo, float scale, float* s_out, float* c_out) {
    const float f = fmaf(lo, scale, __builtin_amdgcn_fractf(hi * scale));
    *s_out = __builtin_amdgcn_sinf(f);
    *c_out = __builtin_amdgcn_cosf(f);
}

}  // namespace ucnerf
