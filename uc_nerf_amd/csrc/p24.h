// "p24": activations and gradients kept between the kernels of the training step as 24-bit floats.
//
// The gradient chain and the weight-gradient launch consume every kept value as a split-bf16 operand pair (hi = the top 16 bits,
// lo = bf16(x - hi)): 16 mantissa bits.  A value stored as the top 24 bits of its fp32 form (sign, exponent, 15 explicit mantissa
// bits; the dropped byte rounded half up) carries exactly those, so the fp32 sets' fourth byte was traffic nobody read:
//   a set [m, C] is m rows of 3 C bytes; column c of sample s = bytes 3 c .. 3 c + 2 of row s (little endian);
//   four consecutive columns (a column GROUP) = 12 bytes = one dwordx3 access of the lane that owns them in the MFMA accumulator layout.
// (A TILED variant -- [tile of 32 samples][group][sample][12 bytes], 768 contiguous bytes per accumulator-layout access -- was built and measured:
//  the training forward's stores gained 54 us per 131 k samples, the gradient chain nothing (its memory time is DRAM latency, not line requests),
//  and the weight-gradient launch, whose lanes run over column groups, lost 90-290 us: profiles/r03_experiments.md.  Rows it is.)
// Relative error of a stored value <= 2^-16 (half an ulp of a 16-bit significand: what the split-bf16 products themselves are good to).
#pragma once
#include <hip/hip_runtime.h>

namespace ucnerf {

typedef unsigned p24_u32x3 __attribute__((ext_vector_type(3)));
typedef p24_u32x3 p24_u32x3_a4 __attribute__((aligned(4)));                // ... at a 4-byte aligned address
struct P24Piece { unsigned d[3]; };     // 12 bytes

constexpr int P24_ROW_BYTES = 384;       // a [m,128] set
constexpr int P24_GROUP_BYTES = 12;      // from one column group of a sample to the next
// byte offset of the 12-byte piece of (sample s, column group g) in a set with G groups per row
__host__ __device__ inline size_t p24_offset(size_t s, int g, int G) { return (s * G + g) * 12; }

// four fp32 -> 12 bytes
__device__ __forceinline__ P24Piece p24_pack4(float v0, float v1, float v2, float v3) {
    const unsigned a = __float_as_uint(v0) + 0x80u, b = __float_as_uint(v1) + 0x80u, c = __float_as_uint(v2) + 0x80u, d = __float_as_uint(v3) + 0x80u;
    P24Piece p;
    p.d[0] = __builtin_amdgcn_perm(b, a, 0x05030201u);      // a.b1 a.b2 a.b3 b.b1
    p.d[1] = __builtin_amdgcn_perm(c, b, 0x06050302u);      // b.b2 b.b3 c.b1 c.b2
    p.d[2] = __builtin_amdgcn_perm(d, c, 0x07060503u);      // c.b3 d.b1 d.b2 d.b3
    return p;
}

// 12 bytes -> four fp32 (low byte zero)
__device__ __forceinline__ void p24_unpack4(const P24Piece& p, float& v0, float& v1, float& v2, float& v3) {
    v0 = __uint_as_float(__builtin_amdgcn_perm(p.d[0], p.d[0], 0x0201000cu));
    v1 = __uint_as_float(__builtin_amdgcn_perm(p.d[1], p.d[0], 0x0504030cu));
    v2 = __uint_as_float(__builtin_amdgcn_perm(p.d[2], p.d[1], 0x0403020cu));
    v3 = __uint_as_float(__builtin_amdgcn_perm(p.d[2], p.d[2], 0x0302010cu));
}

// column c (0..3, a compile-time constant after unrolling) of a piece
__device__ __forceinline__ float p24_unpack1(unsigned d0, unsigned d1, unsigned d2, int c) {
    return __uint_as_float(c == 0 ? __builtin_amdgcn_perm(d0, d0, 0x0201000cu) : c == 1 ? __builtin_amdgcn_perm(d1, d0, 0x0504030cu)
                         : c == 2 ? __builtin_amdgcn_perm(d2, d1, 0x0403020cu) : __builtin_amdgcn_perm(d2, d2, 0x0302010cu));
}

// (through a vector type, not the struct: a struct-typed access can leave a 12-byte stack object behind in divergent control flow)
__device__ __forceinline__ P24Piece p24_load(const char* p) {
    const p24_u32x3 v = *reinterpret_cast<const p24_u32x3_a4*>(p);
    P24Piece r;
    r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z;
    return r;
}
__device__ __forceinline__ void p24_store(char* p, const P24Piece& v) { *reinterpret_cast<p24_u32x3_a4*>(p) = (p24_u32x3){v.d[0], v.d[1], v.d[2]}; }

}  // namespace ucnerf
