// v_dot2c_f32_bf16 as the residual of a truncated-bf16 split: lo = x - hi computed from the PACKED hi pair.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dot2_split scripts/micro/dot2_split.hip && /tmp/dot2_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* lo_dot, float* lo_ref, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const unsigned b0 = __builtin_bit_cast(unsigned, x[i]), b1 = __builtin_bit_cast(unsigned, x[i + 1]);
    const unsigned packed = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const bf16x2 hp = __builtin_bit_cast(bf16x2, packed);
    unsigned m0 = 0x0000bf80u, m1 = 0xbf800000u;
    asm volatile("" : "+v"(m0), "+v"(m1));                       // keep the selectors in registers (no constant folding)
    lo_dot[i] = __builtin_amdgcn_fdot2_f32_bf16(hp, __builtin_bit_cast(bf16x2, m0), x[i], false);
    lo_dot[i + 1] = __builtin_amdgcn_fdot2_f32_bf16(hp, __builtin_bit_cast(bf16x2, m1), x[i + 1], false);
    lo_ref[i] = x[i] - __builtin_bit_cast(float, b0 & 0xffff0000u);
    lo_ref[i + 1] = x[i + 1] - __builtin_bit_cast(float, b1 & 0xffff0000u);
}
int main() {
    const int n = 1 << 20;
    float* h = new float[n];
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float m = (rand() / (float)RAND_MAX) * 2.f - 1.f;
        h[i] = ldexpf(m, (rand() % 60) - 40);                   // wide exponent range
    }
    h[0] = 0.f; h[1] = -0.f; h[2] = 1e-39f; h[3] = -3e-41f; h[4] = 65504.f; h[5] = -1e30f;
    float *x, *a, *b;
    hipMalloc(&x, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(x, a, b, n);
    float* ha = new float[n]; float* hb = new float[n];
    hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (memcmp(&ha[i], &hb[i], 4) && !(ha[i] == 0.f && hb[i] == 0.f)) { if (bad < 8) printf("x=%g dot=%g ref=%g\n", h[i], ha[i], hb[i]); ++bad; }
    printf("%d of %d residuals differ\n", bad, n);
    return 0;
}
