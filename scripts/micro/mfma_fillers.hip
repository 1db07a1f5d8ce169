// Micro-benchmark: how many independent VALU instructions hide in the gap of back-to-back v_mfma_f32_32x32x2_f32,
// (a) issued by the SAME wave between its MFMAs, (b) by a PARTNER wave on the same SIMD.  One block per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int FILL, int NACC>
__global__ void __launch_bounds__(256, 1) same_wave(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u % NACC] = MFMA(a, b, acc[u % NACC]);
#pragma unroll
            for (int f = 0; f < FILL; ++f) v[(u * FILL + f) & 7] = fmaf(v[(u * FILL + f) & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0-3 stream MFMAs, waves 4-7 run a fixed number of VALU instructions then exit: measures block time
template <int VALU_PER_ITER>
__global__ void __launch_bounds__(512, 2) partner(float* out, int iters, int valu_iters, unsigned long long* t_valu) {
    const bool mf = __builtin_amdgcn_readfirstlane(threadIdx.x) < 256;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float s = 0;
    if (mf) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = MFMA(a, b, acc[u & 3]);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int f = 0; f < VALU_PER_ITER; ++f) v[f & 7] = fmaf(v[f & 7], b, a);
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) s += v[i];
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) t_valu[threadIdx.x >> 6] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* tv; hipMalloc(&tv, 64);
    const int iters = 20000;       // 320k MFMAs per wave
    const double mf = 256.0 * 4 * iters * 16;    // MFMAs chip-wide (one wave per SIMD)
#define RUN(FILL, NACC) { float ms = time_ms([&] { hipLaunchKernelGGL((same_wave<FILL, NACC>), dim3(256), dim3(256), 0, 0, out, iters); }); \
        printf("same wave: %d acc, %2d fillers/MFMA: %.1f cycles/MFMA (at 2.4 GHz nominal), %.1f TFLOP/s\n", NACC, FILL, ms * 1e-3 * 2.4e9 / (iters * 16.0), mf * 4096 / (ms * 1e-3) / 1e12); }
    RUN(0, 4) RUN(1, 4) RUN(2, 4) RUN(4, 4) RUN(8, 4) RUN(12, 4) RUN(16, 4)
    RUN(0, 1) RUN(2, 1) RUN(4, 1) RUN(8, 1)
    // partner: 8 VALU waves-worth: count cycles the VALU waves need for 4096 FMAs next to streaming MFMA waves
    for (int mfi : {0, 2000}) {
        hipLaunchKernelGGL((partner<64>), dim3(256), dim3(512), 0, 0, out, mfi, 64, tv);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, tv, 64, hipMemcpyDeviceToHost);
        printf("partner waves: 4096 dependent-free FMAs take %llu cycles when the MFMA waves run %d iterations\n", h[4], mfi);
    }
    return 0;
}
