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

__device__ __forceinline__ void sincos_pe(float r, float* s_out, float* c_out) {
    if (fabsf(r) > 65536.0f) {          // rare: wave-divergent slow path
        sincosf(r, s_out, c_out);
        return;
    }
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

}  // namespace ucnerf
