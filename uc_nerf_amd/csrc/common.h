// Shared host-side helpers for libucnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/ucnerf_hip.h"

namespace ucnerf {

char* last_error_buf();   // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(UCNERF_EHIP, "%s: %s", what, hipGetErrorString(e));
    return UCNERF_OK;
}

#define UCNERF_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::ucnerf::fail(UCNERF_EINVAL, __VA_ARGS__); } while (0)

// An empty batch is a no-op; a NEGATIVE count is the caller's error (round 5: every entry point used to take it for an empty batch).
#define UCNERF_COUNT(n) \
    do { if ((n) < 0) return ::ucnerf::fail(UCNERF_EINVAL, "%s: negative count %lld", __func__, (long long)(n)); if ((n) == 0) return UCNERF_OK; } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

struct Mat34 { float m[12]; };
struct Mat33 { float m[9]; };

int device_cus();   // of the CURRENT device, cached per device (thread-safe)

// ucnerf_build_flags(): every translation unit with compile-time switches reports them as "NAME=value " (stringified after expansion)
#define UCNERF_STR2(x) #x
#define UCNERF_STR(x) UCNERF_STR2(x)
#define UCNERF_FLAG(NAME) #NAME "=" UCNERF_STR(NAME) " "
const char* build_flags_mlp_bf16x3();
const char* build_flags_mlp_bf16_plain();
const char* build_flags_mlp_bwd();
const char* build_flags_mlp_bwd_chain();
const char* build_flags_mlp_wgrad();
const char* build_flags_gather_cl();
const char* build_flags_mlp_f32();

// Opt-in for more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute belongs to the
// (device, function) pair, so it is set once per pair -- keyed on hipGetDevice(), under a lock -- and a failure is
// reported through fail().  Returns UCNERF_OK or UCNERF_EHIP.
int ensure_dynamic_lds(const void* kernel, int bytes, const char* what);


// Parameters living in SEPARATE tensors (a torch module's) addressed as if they were the flat parameter vector: tensor j holds flat
// indices [start[j], start[j + 1]).  Passed by value to the pack kernels of ucnerf_mlp_pack_tensors, searched from an LDS copy.
constexpr int MAX_PACK_TENSORS = 48;
struct ParamTable { const float* ptr[MAX_PACK_TENSORS]; int start[MAX_PACK_TENSORS + 1]; int n; };
#ifdef __HIPCC__
struct ParamTableLds { const float* ptr[MAX_PACK_TENSORS]; int start[MAX_PACK_TENSORS + 1]; };
__device__ __forceinline__ void param_table_to_lds(const ParamTable& t, ParamTableLds* l) {      // all threads of the block; ends with a barrier
    for (int i = threadIdx.x; i < MAX_PACK_TENSORS; i += blockDim.x) { l->ptr[i] = i < t.n ? t.ptr[i] : nullptr; l->start[i] = i <= t.n ? t.start[i] : 0x7fffffff; }
    if (threadIdx.x == 0) l->start[MAX_PACK_TENSORS] = t.n == MAX_PACK_TENSORS ? t.start[MAX_PACK_TENSORS] : 0x7fffffff;
    __syncthreads();
}
__device__ __forceinline__ float param_table_load(const ParamTableLds* l, int f) {
    int lo = 0, hi = MAX_PACK_TENSORS - 1;                    // largest j with start[j] <= f
#pragma unroll
    for (int it = 0; it < 6; ++it) { const int mid = (lo + hi + 1) >> 1; if (l->start[mid] <= f) lo = mid; else hi = mid - 1; }
    return l->ptr[lo][f - l->start[lo]];
}
#endif

// Segmented pre-combination of float atomics inside a wave (HIP device code only).  Lanes STRIDE apart hold consecutive
// positions of a run dimension (samples of a ray, depth hypotheses of a pixel; pos = lane / STRIDE); runs of equal `key`
// along it are summed with a suffix scan in log2(64 / STRIDE) shuffle steps and only the first lane of a run issues the
// atomic.  key < 0 = nothing to add.  Must be called by all 64 lanes.
#ifdef __HIPCC__
template <int STRIDE>                                                  // lanes STRIDE apart are consecutive samples; pos = lane / STRIDE
__device__ __forceinline__ void run_atomic_add(float* base, int key, float v, int pos) {
    constexpr int N = 64 / STRIDE;
    const int kp = __shfl_up(key, STRIDE), kn = __shfl_down(key, STRIDE);   // (unconditionally: a shuffle inside `a || b` would run
    const bool head = pos == 0 || kp != key;                                 //  with the short-circuited lanes masked off and read 0 from them)
    int end = pos == N - 1 || kn != key;
    float s = v;
#pragma unroll
    for (int d = 1; d < N; d <<= 1) {
        const float sn = __shfl_down(s, STRIDE * d);
        const int en = __shfl_down(end, STRIDE * d);
        if (!end) { s += sn; end = en; }          // (no run end within the covered span -> lane pos + d exists)
    }
    if (head && key >= 0 && s != 0.f) atomicAdd(base + key, s);
}

// The same over TWO batches of positions held by the same lanes (batch A = positions 0 .. N-1, batch B = N .. 2N-1 of one run dimension: a wave covers
// 2N consecutive samples): a run that crosses from A's last position into B's first is issued ONCE, by its head in A.  Twice the window of
// run_atomic_add for the same lanes -- views' runs are ~16 samples long, the finest volume's ~11 (round 5).
template <int STRIDE>
__device__ __forceinline__ void run_atomic_add2(float* base, int key_a, float v_a, int key_b, float v_b, int pos) {
    constexpr int N = 64 / STRIDE;
    const int lane = threadIdx.x & 63, col = lane % STRIDE;
    // ---- batch B on its own
    const int kpb = __shfl_up(key_b, STRIDE), knb = __shfl_down(key_b, STRIDE);
    int end_b = pos == N - 1 || knb != key_b;
    float s_b = v_b;
#pragma unroll
    for (int d = 1; d < N; d <<= 1) {
        const float sn = __shfl_down(s_b, STRIDE * d);
        const int en = __shfl_down(end_b, STRIDE * d);
        if (!end_b) { s_b += sn; end_b = en; }
    }
    // ---- the run at B's first position, seen from A's last
    const int key_b0 = __shfl(key_b, col);
    const float s_b0 = __shfl(s_b, col);
    const int key_a7 = __shfl(key_a, (N - 1) * STRIDE + col);
    const bool absorbed = key_a7 == key_b0 && key_b0 >= 0;              // (uniform over the lanes of a column)
    if (pos == N - 1 && absorbed) v_a += s_b0;
    // ---- batch A
    const int kpa = __shfl_up(key_a, STRIDE), kna = __shfl_down(key_a, STRIDE);
    int end_a = pos == N - 1 || kna != key_a;
    float s_a = v_a;
#pragma unroll
    for (int d = 1; d < N; d <<= 1) {
        const float sn = __shfl_down(s_a, STRIDE * d);
        const int en = __shfl_down(end_a, STRIDE * d);
        if (!end_a) { s_a += sn; end_a = en; }
    }
    const bool head_a = pos == 0 || kpa != key_a;
    const bool head_b = (pos == 0 ? !absorbed : kpb != key_b);
    if (head_a && key_a >= 0 && s_a != 0.f) atomicAdd(base + key_a, s_a);
    if (head_b && key_b >= 0 && s_b != 0.f) atomicAdd(base + key_b, s_b);
}
#endif

}  // namespace ucnerf
