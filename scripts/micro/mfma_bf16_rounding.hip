// How does v_mfma_f32_32x32x16_bf16 round its fp32 accumulation?  C = 1.0 (or -1.0), one product of p ulps of 1.0 in the k-sum:
// round-to-nearest gives 1 + ulp for p = 0.75 and 1.0 for p = 0.25; truncation gives 1.0 for both (and -1.0 for the negative case).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float c0, float a0, float b0) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.f; b[i] = (__bf16)0.f; }
    a[0] = (__bf16)a0; b[0] = (__bf16)b0;              // (every lane: row l & 31 of A, column l & 31 of B, k = 8 (l >> 5): two equal products per output)
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float ulp = 1.1920929e-07f;                   // 2^-23
    struct { const char* name; float c, a, b; } cases[] = {
        {"+1 + 2 * 0.375 ulp (= 0.75 ulp)", 1.f, 0.375f, ulp}, {"+1 + 2 * 0.125 ulp (= 0.25 ulp)", 1.f, 0.125f, ulp},
        {"-1 - 0.75 ulp", -1.f, -0.375f, ulp}, {"+1 - 0.25 ulp (0.5 ulp of the binade below)", 1.f, -0.125f, ulp}, {"+1 - 0.75 ulp", 1.f, -0.375f, ulp},
        {"+1 + 1.5 ulp", 1.f, 0.75f, ulp}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cs.c, cs.a, cs.b);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%-48s -> %.9g  (= c %+0.2f ulp)\n", cs.name, h, (h - cs.c) / ulp);
    }
    return 0;
}
