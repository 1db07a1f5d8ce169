// Accuracy probe: positional-encoding sin/cos through v_sin_f32 / v_cos_f32 (argument in revolutions) with a two-float
// reduction of x / (2 pi), against double precision, for the arguments the point encoding produces (|x| <= 1.3, 2^0..2^9).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* x, float* s, float* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float chi = 0.15915494309189535f, clo = (float)(0.15915494309189533576888 - (double)0.15915494309189535f);
    const float xv = x[i];
    const float thi = xv * chi;
    const float tlo = fmaf(xv, clo, fmaf(xv, chi, -thi));
    for (int kf = 0; kf < 10; ++kf) {
        const float sc = (float)(1 << kf);
        const float f = fmaf(tlo, sc, __builtin_amdgcn_fractf(thi * sc));
        s[i * 10 + kf] = __builtin_amdgcn_sinf(f);
        c[i * 10 + kf] = __builtin_amdgcn_cosf(f);
    }
}

int main() {
    const int n = 1 << 16;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = -1.3f + 2.6f * (float)i / n + 1e-4f * (float)((i * 7919) % 97);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 40); hipMalloc(&dc, n * 40);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, ds, dc, n);
    std::vector<float> s(n * 10), c(n * 10);
    hipMemcpy(s.data(), ds, n * 40, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 40, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; ++i)
        for (int kf = 0; kf < 10; ++kf) {
            const double a = (double)x[i] * (double)(1 << kf);
            es = fmax(es, fabs(s[i * 10 + kf] - sin(a)));
            ec = fmax(ec, fabs(c[i * 10 + kf] - cos(a)));
        }
    printf("max |sin err| %.3e  max |cos err| %.3e\n", es, ec);
    return 0;
}
