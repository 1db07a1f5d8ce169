// Device helpers of the channel-last gather (grid_sample semantics restated; reference lines cited in gather.hip), shared by the
// stand-alone gather kernels (gather_cl.hip) and the gather fused into the bf16x3 MLP kernel (mlp_bf16.hip, row f1).
#pragma once
#include <hip/hip_runtime.h>


namespace ucnerf {

__device__ __forceinline__ float unnorm_cl(float g, int size, bool align) {
    float i = align ? (g + 1.f) / 2.f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) / 2.f;
    return fminf(fmaxf(i, 0.f), (float)(size - 1));
}

struct LerpCl { int i0, i1; float w0, w1; };

__device__ __forceinline__ LerpCl axis_cl(float g, int size, bool align) {
    const float x = unnorm_cl(g, size, align);
    const float f = floorf(x);
    LerpCl a;
    a.i0 = (int)f;
    a.w1 = x - f;
    a.w0 = 1.f - a.w1;
    a.i1 = a.i0 + 1;
    if (a.i1 > size - 1) { a.i1 = size - 1; a.w1 = 0.f; }
    return a;
}

__device__ __forceinline__ void project_cl(const float* M, const float* K, float x, float y, float z, float* qx, float* qy, float* qz) {
    const float cx = x * M[0] + y * M[1] + z * M[2] + M[3];
    const float cy = x * M[4] + y * M[5] + z * M[6] + M[7];
    float cz = x * M[8] + y * M[9] + z * M[10] + M[11];
    if (fabsf(cz) < 1e-4f) cz = 1e-4f;
    *qx = cx * K[0] + cy * K[1] + cz * K[2];
    *qy = cx * K[3] + cy * K[4] + cz * K[5];
    *qz = cx * K[6] + cy * K[7] + cz * K[8];
}

typedef float gf2 __attribute__((ext_vector_type(2)));

// 16 bytes at a 32-bit byte offset from a block-uniform base: the address is one scalar pair + one VGPR (no 64-bit
// vector arithmetic per corner); every repacked source is far below 4 GB.
__device__ __forceinline__ float4 ld16(const char* base, unsigned off) {
    return *(const float4*)(base + off);
}

__device__ __forceinline__ float2 ld8(const char* base, unsigned off) {
    return *(const float2*)(base + off);
}
// 12 bytes at 4-byte alignment: the (r, g, b) of a pixel of a [V,H,W,3] image stack (torch's channels_last of [V,3,H,W]) -- one global_load_dwordx3
typedef float gf3 __attribute__((ext_vector_type(3)));
typedef gf3 gf3u __attribute__((aligned(4)));
__device__ __forceinline__ gf3u ld12(const char* base, unsigned off) {
    return *(const gf3u*)(base + off);
}
// bf16 channel-last copies: eight bf16 in 16 bytes -> the first / second four as floats (a bf16 is the upper half of its float)
__device__ __forceinline__ float4 bf16x4_lo(const float4& r) {
    const unsigned a = __float_as_uint(r.x), b = __float_as_uint(r.y);
    return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u));
}
__device__ __forceinline__ float4 bf16x4_hi(const float4& r) {
    const unsigned a = __float_as_uint(r.z), b = __float_as_uint(r.w);
    return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u));
}

// o += v * w on four packed pairs (v_pk_fma_f32: the products are not rounded separately -- the reference's own CUDA
// grid_sample contracts the same way; the parity bar on the features is 2e-5).
#define FMA4(O, A, WT)                                                                           \
    { const float4 a_ = (A); const gf2 w_ = {(WT), (WT)};                                        \
      O[0] = __builtin_elementwise_fma((gf2){a_.x, a_.y}, w_, O[0]);                             \
      O[1] = __builtin_elementwise_fma((gf2){a_.z, a_.w}, w_, O[1]); }

}  // namespace ucnerf
