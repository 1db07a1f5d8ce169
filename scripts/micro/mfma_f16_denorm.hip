// Would an fp16 three-term split work on gfx950's matrix cores?  (1) does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the lo term of a small
// activation is one) or flush them to zero; (2) which instructions convert a pair of floats to fp16 (round to nearest / toward zero) and how precise is
// hi + lo; (3) the sustained rate of the f16 MFMA next to the bf16 one (same launch shape as mfma_shapes.hip: 1024 blocks x 256 threads, 4 accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void denorm(float* out, float a0, float b0) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a0; b[0] = (_Float16)b0;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}

__global__ void split(const float* x, float* err_rn, float* err_rtz, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    {   // round-to-nearest hi and lo
        const f16x2 hp = __builtin_convertvector((f32x2){v, v}, f16x2);
        const float h = (float)hp[0];
        const f16x2 lp = __builtin_convertvector((f32x2){v - h, v - h}, f16x2);
        err_rn[i] = fabsf((float)((double)v - (double)h - (double)(float)lp[0])) / fabsf(v);
    }
    {   // toward-zero hi (v_cvt_pkrtz_f16_f32), round-to-nearest lo
        const auto hp = __builtin_amdgcn_cvt_pkrtz(v, v);
        const float h = (float)hp[0];
        const f16x2 lp = __builtin_convertvector((f32x2){v - h, v - h}, f16x2);
        err_rtz[i] = fabsf((float)((double)v - (double)h - (double)(float)lp[0])) / fabsf(v);
    }
}

template <bool F16>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    f16x8 a16, b16; bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)(0.001f * (threadIdx.x + i)); b16[i] = (_Float16)(0.002f * (threadIdx.x - i)); ab[i] = (__bf16)(float)a16[i]; bb[i] = (__bf16)(float)b16[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (F16) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, c[j], 0, 0, 0);
            else c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    if (s == 12345.f) out[0] = s;
}

int main() {
    float* d; hipMalloc(&d, 64);
    struct { const char* name; float a, b; } cases[] = {
        {"normal x normal        (1e-3 * 1.0)", 1e-3f, 1.f}, {"subnormal x normal     (3e-5 * 1.0)", 3e-5f, 1.f}, {"subnormal x normal     (1e-6 * 1.0)", 1e-6f, 1.f},
        {"smallest subnormal     (6e-8 * 1.0)", 6e-8f, 1.f}, {"subnormal x 1024       (3e-5 * 1024)", 3e-5f, 1024.f}, {"subnormal x subnormal  (3e-5 * 3e-5)", 3e-5f, 3e-5f}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, d, cs.a, cs.b);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%-40s a as fp16 = %.6g, mfma -> %.6g (two equal products: expect %.6g)\n", cs.name, h[1], h[0], 2.0 * (double)h[1] * (double)(float)(_Float16)cs.b);
    }
    const int n = 1 << 20;
    float *x, *e1, *e2; hipMalloc(&x, 4 * n); hipMalloc(&e1, 4 * n); hipMalloc(&e2, 4 * n);
    float* hx = new float[n]; float* he = new float[n];
    for (int scale = 0; scale < 3; ++scale) {
        const float s = scale == 0 ? 1.f : scale == 1 ? 1e-2f : 100.f;
        for (int i = 0; i < n; ++i) hx[i] = s * (0.05f + 4.f * (float)rand() / RAND_MAX);
        hipMemcpy(x, hx, 4 * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(split, dim3(n / 256), dim3(256), 0, 0, x, e1, e2, n);
        double m1 = 0, m2 = 0;
        hipMemcpy(he, e1, 4 * n, hipMemcpyDeviceToHost); for (int i = 0; i < n; ++i) m1 = he[i] > m1 ? he[i] : m1;
        hipMemcpy(he, e2, 4 * n, hipMemcpyDeviceToHost); for (int i = 0; i < n; ++i) m2 = he[i] > m2 ? he[i] : m2;
        printf("split of values ~%g: max |x - hi - lo| / |x|: hi rounded %.3g (2^%.1f), hi toward zero %.3g (2^%.1f)\n", s, m1, log2(m1), m2, log2(m2));
    }
    hipEvent_t e0, e1v; hipEventCreate(&e0); hipEventCreate(&e1v);
    for (int f16 = 0; f16 < 2; ++f16) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) hipLaunchKernelGGL(rate<true>, dim3(1024), dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL(rate<false>, dim3(1024), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1v); hipEventSynchronize(e1v);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1v);
        const double flop = 1024.0 * 4 /*waves*/ * iters * 4 /*mfma*/ * 2.0 * 32 * 32 * 16;
        printf("%s MFMA 32x32x16: %.1f TFLOP/s executed (%.2f ms)\n", f16 ? "f16 " : "bf16", flop / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
