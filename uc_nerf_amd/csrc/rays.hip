// K1/K2/K3 + a2/a5 small per-ray / per-sample kernels (HBM-bound, one thread per element).
#include "common.h"
#include "sincos_cw.h"
#include "raygen_device.h"

#include <atomic>
#include <mutex>
#include <set>
#include <utility>

namespace ucnerf {

thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }

// Per-device caches (a process may drive several GPUs, from several threads).
constexpr int MAX_DEVICES = 64;

int device_cus() {
    static std::atomic<int> cus[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) return -1;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

int ensure_dynamic_lds(const void* kernel, int bytes, const char* what) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(UCNERF_EHIP, "%s: no current device", what);
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, kernel);
    if (done.count(key)) return UCNERF_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(UCNERF_EHIP, "%s: cannot reserve %d bytes of dynamic LDS on device %d: %s", what, bytes, dev, hipGetErrorString(e));
    done.insert(key);
    return UCNERF_OK;
}

// ------------------------------------------------------------------------------------------- a1
__device__ __forceinline__ void ray_gen_one(const ucnerf_ray_gen_params& p, int i) {
    float x, y;
    if (p.xs) {
        x = p.xs[i];
        y = p.ys[i];
    } else {                                   // row-major pixel grid (meshgrid 'ij', flattened)
        int idx = p.grid_start + i;
        y = (float)(idx / p.W);
        x = (float)(idx % p.W);
    }
    const float* R = p.c2w;
    float wx, wy, wz;
    if (p.opengl) {                            // utils/run_nerf_helpers.py:252
        float f = p.K[0];
        const float dx = (x - p.W * .5f) / f, dy = -(y - p.H * .5f) / f, dz = -1.0f;
        wx = dx * R[0] + dy * R[1] + dz * R[2]; wy = dx * R[4] + dy * R[5] + dz * R[6]; wz = dx * R[8] + dy * R[9] + dz * R[10];
    } else {                                   // data/ray_utils.py:27, utils/utils.py:259-261
        pinhole_ray(x, y, p.K[0], p.K[2], p.K[4], p.K[5], R, &wx, &wy, &wz);
    }
    p.rays_d[3 * i + 0] = wx;
    p.rays_d[3 * i + 1] = wy;
    p.rays_d[3 * i + 2] = wz;
    if (p.angle)                               // the arithmetic of dir_feature_kernel on the values just stored
        view_dir_feature(wx, wy, wz, p.w2c_dir, &p.angle[3 * i + 0], &p.angle[3 * i + 1], &p.angle[3 * i + 2]);
    if (p.rays_o) {
        p.rays_o[3 * i + 0] = R[3];
        p.rays_o[3 * i + 1] = R[7];
        p.rays_o[3 * i + 2] = R[11];
    }
    if (p.pix) {
        p.pix[i] = y;
        p.pix[p.n + i] = x;
    }
}

__global__ void ray_gen_kernel(ucnerf_ray_gen_params p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    ray_gen_one(p, i);
}

// ------------------------------------------------------------------------------------------- a2
__global__ void ndc_rays_kernel(ucnerf_ndc_rays_params p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    float ox = p.rays_o[3 * i], oy = p.rays_o[3 * i + 1], oz = p.rays_o[3 * i + 2];
    float dx = p.rays_d[3 * i], dy = p.rays_d[3 * i + 1], dz = p.rays_d[3 * i + 2];
    float t = -(p.near + oz) / dz;
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
    float sx = -1.f / (p.W / (2.f * p.focal_x)), sy = -1.f / (p.H / (2.f * p.focal_y));
    float ox_oz = ox / oz, oy_oz = oy / oz;
    float o2 = 1.f + 2.f * p.near / oz;
    p.out_o[3 * i] = sx * ox_oz;
    p.out_o[3 * i + 1] = sy * oy_oz;
    p.out_o[3 * i + 2] = o2;
    p.out_d[3 * i] = sx * (dx / dz - ox_oz);
    p.out_d[3 * i + 1] = sy * (dy / dz - oy_oz);
    p.out_d[3 * i + 2] = p.variant == 0 ? 1.f - o2 : -2.f * p.near / oz;
}

// ------------------------------------------------------------------------------- view-dir feature
__global__ void dir_feature_kernel(ucnerf_dir_feature_params p) {
    const int rep = p.repeat > 1 ? p.repeat : 1;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.n * rep) return;
    const int i = (int)(idx / rep);
    float dx = p.rays_d[3 * i], dy = p.rays_d[3 * i + 1], dz = p.rays_d[3 * i + 2];
    float c = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= c; dy /= c; dz /= c;
    if (p.cos_angle && idx % rep == 0) p.cos_angle[i] = c;
    float* o = p.angle + 3 * idx;
    if (p.has_ref || p.w2c_ref_dev) {
        const float* R = p.w2c_ref_dev ? p.w2c_ref_dev : p.w2c_ref;
        o[0] = dx * R[0] + dy * R[1] + dz * R[2];
        o[1] = dx * R[4] + dy * R[5] + dz * R[6];
        o[2] = dx * R[8] + dy * R[9] + dz * R[10];
    } else {
        o[0] = dx; o[1] = dy; o[2] = dz;
    }
}

// ------------------------------------------------------------------------------------------- a3
__device__ __forceinline__ void sample_stratified_one(const ucnerf_sample_stratified_params& p, long long idx) {
    int r = (int)(idx / p.S), s = (int)(idx % p.S);
    const float* ray = p.rays ? p.rays + 8 * (size_t)r : nullptr;
    const float near = ray ? ray[6] : p.near, far = ray ? ray[7] : p.far;
    const float z = stratified_depth(near, far, s, p.S, p.lindisp, p.perturb, p.perturb > 0.f ? p.noise[idx] : 0.f);
    p.z[idx] = z;
    if (p.pts) {
        p.pts[3 * idx] = ray[0] + ray[3] * z;
        p.pts[3 * idx + 1] = ray[1] + ray[4] * z;
        p.pts[3 * idx + 2] = ray[2] + ray[5] * z;
    }
}

__global__ void sample_stratified_kernel(ucnerf_sample_stratified_params p) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.n * p.S) return;
    sample_stratified_one(p, idx);
}

// rays and their depths from ONE launch (the head of the hierarchical step: two ~5 us launches were mostly launch latency):
// thread (ray, sample) writes its depth; the thread of sample 0 also generates the ray
__global__ void ray_gen_sample_kernel(ucnerf_ray_gen_params rg, ucnerf_sample_stratified_params ss) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)ss.n * ss.S) return;
    sample_stratified_one(ss, idx);
    if (idx % ss.S == 0) ray_gen_one(rg, (int)(idx / ss.S));
}

// Live cascade sampler: one 64-thread block per ray; the S values are sorted in LDS (bitonic, padded to a
// power of two), then stratified.  S <= 768.
__global__ void __launch_bounds__(64) sample_cascade_kernel(ucnerf_sample_cascade_params p) {
    __shared__ float zs[1024];
    int r = blockIdx.x;
    int S = p.S, n3 = S / 3;
    int P = 1;
    while (P < S) P <<= 1;
    const float* nf = p.near_far + 6 * (size_t)r;
    for (int i = threadIdx.x; i < P; i += 64) {
        float v = __builtin_inff();
        if (i < S) {
            int k = i / n3, j = i % n3;
            float t = linspace01(j, n3);
            v = nf[2 * k] * (1.0f - t) + nf[2 * k + 1] * t;       // utils/utils.py:396
        }
        zs[i] = v;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 64) {
                int l = i ^ j;
                if (l > i) {
                    float a = zs[i], b = zs[l];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { zs[i] = b; zs[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int s = threadIdx.x; s < S; s += 64) {
        float z = zs[s];
        if (p.t_rand) {
            float lower = s > 0 ? .5f * (z + zs[s - 1]) : z;      // mids = .5*(z[1:] + z[:-1])
            float upper = s + 1 < S ? .5f * (zs[s + 1] + z) : z;
            z = lower + (upper - lower) * p.t_rand[(size_t)r * S + s];
        }
        size_t o = (size_t)r * S + s;
        p.z[o] = z;
        if (p.pts) {
            p.pts[3 * o] = p.rays_o[0] + z * p.rays_d[3 * r];
            p.pts[3 * o + 1] = p.rays_o[1] + z * p.rays_d[3 * r + 1];
            p.pts[3 * o + 2] = p.rays_o[2] + z * p.rays_d[3 * r + 2];
        }
    }
}

// ------------------------------------------------------------------------------------------- a1 + a3 + a4 of the evaluation loop, one launch
// utils/utils.py:600-739 (build_rays_test): one 64-thread block per ray, the arithmetic of ray_gen_one (grid mode), sample_cascade_kernel and
// ndc_project_kernel in that order on the same values -- bit-identical to the three launches.  Matrices come from device memory (uniform loads).
__global__ void __launch_bounds__(64) build_rays_test_kernel(ucnerf_build_rays_test_params p) {
    __shared__ float zs[1024];
    const int r = blockIdx.x;
    const int S = p.S, n3 = S / 3;
    const int idx = p.grid_start + r;
    const int row = idx / p.W, col = idx % p.W;
    float wx, wy, wz;
    pinhole_ray((float)col, (float)row, p.K[0], p.K[2], p.K[4], p.K[5], p.c2w, &wx, &wy, &wz);
    const float ox = p.c2w[3], oy = p.c2w[7], oz = p.c2w[11];
    float nf[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {                           // utils/utils.py:659-683: first / last hypothesis plane at the pixel of the stage's resolution
        const int div = 4 >> k;
        const size_t px = (size_t)(row / div) * p.dv_w[k] + (col / div), plane = (size_t)p.dv_h[k] * p.dv_w[k];
        nf[2 * k] = p.depth_values[k][px];
        nf[2 * k + 1] = p.depth_values[k][(size_t)(p.dv_d[k] - 1) * plane + px];
    }
    if (threadIdx.x == 0) {
        p.rays_d[3 * (size_t)r] = wx; p.rays_d[3 * (size_t)r + 1] = wy; p.rays_d[3 * (size_t)r + 2] = wz;
        if (p.near_far)
            for (int k = 0; k < 6; ++k) p.near_far[6 * (size_t)r + k] = nf[k];
        if (p.rays_o && r == 0) { p.rays_o[0] = ox; p.rays_o[1] = oy; p.rays_o[2] = oz; }
    }
    int P = 1;
    while (P < S) P <<= 1;
    for (int i = threadIdx.x; i < P; i += 64) {
        float v = __builtin_inff();
        if (i < S) {
            const int k = i / n3, j = i % n3;
            const float t = linspace01(j, n3);
            v = nf[2 * k] * (1.0f - t) + nf[2 * k + 1] * t;       // utils/utils.py:396
        }
        zs[i] = v;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = zs[i], b = zs[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { zs[i] = b; zs[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    const float* M = p.w2c_ref;
    const float* K = p.K_ref;
    const float near = p.near_far_ref[0], far = p.near_far_ref[1];
    const float inv_w = (float)(p.W - 1), inv_h = (float)(p.H - 1);
    for (int s = threadIdx.x; s < S; s += 64) {
        float z = zs[s];
        if (p.t_rand) {
            const float lower = s > 0 ? .5f * (z + zs[s - 1]) : z;      // mids = .5*(z[1:] + z[:-1])
            const float upper = s + 1 < S ? .5f * (zs[s + 1] + z) : z;
            z = lower + (upper - lower) * p.t_rand[(size_t)r * S + s];
        }
        const size_t o1 = (size_t)r * S + s, o = 3 * o1;
        p.z[o1] = z;
        const float x = ox + z * wx, y = oy + z * wy, w = oz + z * wz;
        p.pts[o] = x; p.pts[o + 1] = y; p.pts[o + 2] = w;
        const float cx = x * M[0] + y * M[1] + w * M[2] + M[3];      // utils/utils.py:333-367
        const float cy = x * M[4] + y * M[5] + w * M[6] + M[7];
        float cz = x * M[8] + y * M[9] + w * M[10] + M[11];
        if (fabsf(cz) < 1e-4f) cz = 1e-4f;
        const float qx = cx * K[0] + cy * K[1] + cz * K[2];
        const float qy = cx * K[3] + cy * K[4] + cz * K[5];
        const float qz = cx * K[6] + cy * K[7] + cz * K[8];
        const float u = (qx / qz + 0.0f) / inv_w, v = (qy / qz + 0.0f) / inv_h;
        p.ndc1[o] = u; p.ndc1[o + 1] = v; p.ndc1[o + 2] = (qz - nf[0]) / (nf[1] - nf[0]);
        p.ndc2[o] = u; p.ndc2[o + 1] = v; p.ndc2[o + 2] = (qz - nf[2]) / (nf[3] - nf[2]);
        p.ndc3[o] = u; p.ndc3[o + 1] = v; p.ndc3[o + 2] = (qz - nf[4]) / (nf[5] - nf[4]);
        p.ndc[o] = u; p.ndc[o + 1] = v; p.ndc[o + 2] = (qz - near) / (far - near);
    }
}

// ------------------------------------------------------------------------------------------- a4
__global__ void ndc_project_kernel(ucnerf_ndc_project_params p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.m) return;
    float x = p.pts[3 * (size_t)i], y = p.pts[3 * (size_t)i + 1], z = p.pts[3 * (size_t)i + 2];
    if (p.has_w2c) {
        const float* M = p.w2c;
        float cx = x * M[0] + y * M[1] + z * M[2] + M[3];
        float cy = x * M[4] + y * M[5] + z * M[6] + M[7];
        float cz = x * M[8] + y * M[9] + z * M[10] + M[11];
        if (fabsf(cz) < 1e-4f) cz = 1e-4f;
        x = cx; y = cy; z = cz;
    }
    const float* K = p.K;
    float qx = x * K[0] + y * K[1] + z * K[2];
    float qy = x * K[3] + y * K[4] + z * K[5];
    float qz = x * K[6] + y * K[7] + z * K[8];
    float u = (qx / qz + 0.0f) / p.inv_scale[0];
    float v = (qy / qz + 0.0f) / p.inv_scale[1];
    size_t o = 3 * (size_t)i;
    if (p.sample_2d) {
        p.out_ndc[o] = u; p.out_ndc[o + 1] = v; p.out_ndc[o + 2] = qz;
        return;
    }
    size_t k = (size_t)i * p.nf_stride;
    float* outs[3] = {p.out_stage1, p.out_stage2, p.out_stage3};
    const float* nears[3] = {p.near_1, p.near_2, p.near_3};
    const float* fars[3] = {p.far_1, p.far_2, p.far_3};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (!outs[s]) continue;
        float n = nears[s][k], f = fars[s][k];
        outs[s][o] = u; outs[s][o + 1] = v; outs[s][o + 2] = (qz - n) / (f - n);
    }
    if (p.out_ndc) {
        p.out_ndc[o] = u; p.out_ndc[o + 1] = v; p.out_ndc[o + 2] = (qz - p.near) / (p.far - p.near);
    }
}

// ------------------------------------------------------------------------------------------- a5
// out_stride: floats between consecutive output rows (the public entry point: dense, 3 + 6 n_freqs)
__global__ void embed_kernel(ucnerf_embed_params p, int out_stride) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (vector, freq slot)
    int L = p.n_freqs;
    if (idx >= (long long)p.m * (L + 1)) return;
    int v = (int)(idx / (L + 1)), k = (int)(idx % (L + 1));
    const float* x = p.x + 3 * (size_t)v;
    float* o = p.out + (size_t)v * out_stride;
    if (k == L) {
        o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
        return;
    }
    float f = (float)(1 << k);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s, co;
        sincos_pe(x[c] * f, &s, &co);
        if (p.layout == 0) {                 // [x | sin(f0..) | cos(f0..)], 3 per frequency
            o[3 + 3 * k + c] = s;
            o[3 + 3 * L + 3 * k + c] = co;
        } else {                             // [x | sin f0 | cos f0 | sin f1 | ...]
            o[3 + 6 * k + c] = s;
            o[3 + 6 * k + 3 + c] = co;
        }
    }
}

// rows `out_stride` floats apart (mlp_bwd.hip: the weight-gradient launch reads its fp32 operands in 16-byte pieces -- dense rows of 63 / 27
// floats put every piece on an unaligned address)
int launch_embed_strided(int m, int n_freqs, int layout, const float* x, float* out, int out_stride, hipStream_t st) {
    UCNERF_COUNT(m);
    ucnerf_embed_params p;
    p.m = m; p.n_freqs = n_freqs; p.layout = layout; p.x = x; p.out = out;
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv((long long)m * (n_freqs + 1), 256)), dim3(256), 0, st, p, out_stride);
    return check_launch("embed");
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

const char* ucnerf_last_error(void) { return last_error_buf(); }
int ucnerf_abi_version(void) { return UCNERF_ABI_VERSION; }
int ucnerf_device_cus(void) { return device_cus(); }
const char* ucnerf_build_flags(void) {
    static char buf[2048];
    static bool done = false;          // (idempotent content: a race writes the same bytes)
    if (!done) {
        snprintf(buf, sizeof(buf), "%s| %s| %s| %s| %s| %s| %s", build_flags_mlp_bf16x3(), build_flags_mlp_bf16_plain(), build_flags_mlp_f32(), build_flags_mlp_bwd(),
                 build_flags_mlp_bwd_chain(), build_flags_mlp_wgrad(), build_flags_gather_cl());
        done = true;
    }
    return buf;
}

void* ucnerf_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { fail(UCNERF_EHIP, "event_create failed"); return nullptr; }
    return (void*)e;
}
int ucnerf_event_record(void* event, void* stream) {
    UCNERF_REQUIRE(event, "event_record: null event");
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    return e == hipSuccess ? UCNERF_OK : fail(UCNERF_EHIP, "event_record: %s", hipGetErrorString(e));
}
int ucnerf_event_elapsed_ms(void* start, void* stop, float* ms_host) {
    UCNERF_REQUIRE(start && stop && ms_host, "event_elapsed_ms: null argument");
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e == hipSuccess) e = hipEventElapsedTime(ms_host, (hipEvent_t)start, (hipEvent_t)stop);
    return e == hipSuccess ? UCNERF_OK : fail(UCNERF_EHIP, "event_elapsed_ms: %s", hipGetErrorString(e));
}
int ucnerf_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
    return UCNERF_OK;
}

int ucnerf_sizeof(const char* name) {
#define SZ(T) if (!strcmp(name, #T)) return (int)sizeof(T)
    SZ(ucnerf_ray_gen_params); SZ(ucnerf_ndc_rays_params); SZ(ucnerf_dir_feature_params);
    SZ(ucnerf_sample_stratified_params); SZ(ucnerf_sample_cascade_params); SZ(ucnerf_ndc_project_params);
    SZ(ucnerf_embed_params); SZ(ucnerf_feat_gather_params); SZ(ucnerf_feat_gather_bwd_params);
    SZ(ucnerf_mlp_config); SZ(ucnerf_mlp_params); SZ(ucnerf_mlp_bwd_params); SZ(ucnerf_composite_params);
    SZ(ucnerf_composite_bwd_params); SZ(ucnerf_sample_pdf_params); SZ(ucnerf_merge_rows_params); SZ(ucnerf_cost_volume_params); SZ(ucnerf_depth_regress_params); SZ(ucnerf_cost_volume_bwd_params); SZ(ucnerf_depth_regress_bwd_params); SZ(ucnerf_render_params);
    SZ(ucnerf_render_bwd_params); SZ(ucnerf_cl_sources); SZ(ucnerf_cl_grads); SZ(ucnerf_build_rays_test_params);
#undef SZ
    return -1;
}

int ucnerf_ray_gen(const ucnerf_ray_gen_params* p, void* stream) {
    UCNERF_REQUIRE(p, "ray_gen: null params");
    UCNERF_REQUIRE(p->n >= 0, "ray_gen: n < 0");
    if (p->n == 0) return UCNERF_OK;
    UCNERF_REQUIRE(p->rays_d, "ray_gen: null rays_d");
    UCNERF_REQUIRE((p->xs == nullptr) == (p->ys == nullptr), "ray_gen: xs and ys must both be given or both NULL");
    UCNERF_REQUIRE(p->xs || (p->W > 0 && p->H > 0 && p->grid_start >= 0 &&
                             (long long)p->grid_start + p->n <= (long long)p->H * p->W),
                   "ray_gen: grid range [%d, %d) outside %dx%d", p->grid_start, p->grid_start + p->n, p->H, p->W);
    if (p->n == 0) return UCNERF_OK;
    hipLaunchKernelGGL(ray_gen_kernel, dim3(cdiv(p->n, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("ray_gen");
}

int ucnerf_ray_gen_sample(const ucnerf_ray_gen_params* rg, const ucnerf_sample_stratified_params* ss, void* stream) {
    UCNERF_REQUIRE(rg && ss, "ray_gen_sample: null params");
    UCNERF_REQUIRE(rg->n == ss->n && rg->n >= 0, "ray_gen_sample: %d rays but depths for %d", rg->n, ss->n);
    if (rg->n == 0) return UCNERF_OK;
    UCNERF_REQUIRE(rg->rays_d && ss->z && ss->S >= 1, "ray_gen_sample: null rays_d / z or S < 1");
    UCNERF_REQUIRE((rg->xs == nullptr) == (rg->ys == nullptr), "ray_gen_sample: xs and ys must both be given or both NULL");
    UCNERF_REQUIRE(rg->xs || (rg->W > 0 && rg->H > 0 && rg->grid_start >= 0 && (long long)rg->grid_start + rg->n <= (long long)rg->H * rg->W),
                   "ray_gen_sample: grid range outside the image");
    UCNERF_REQUIRE(!ss->rays && !ss->pts, "ray_gen_sample: depths use the scalar near / far form (rays = pts = NULL)");
    UCNERF_REQUIRE(!(ss->perturb > 0.f) || ss->noise, "ray_gen_sample: perturb > 0 needs noise draws");
    hipLaunchKernelGGL(ray_gen_sample_kernel, dim3(cdiv((long long)ss->n * ss->S, 256)), dim3(256), 0, (hipStream_t)stream, *rg, *ss);
    return check_launch("ray_gen_sample");
}

int ucnerf_ndc_rays(const ucnerf_ndc_rays_params* p, void* stream) {
    UCNERF_REQUIRE(p, "ndc_rays: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->rays_o && p->rays_d && p->out_o && p->out_d, "ndc_rays: null pointer");
    UCNERF_REQUIRE(p->variant == 0 || p->variant == 1, "ndc_rays: variant %d", p->variant);
    UCNERF_COUNT(p->n);
    hipLaunchKernelGGL(ndc_rays_kernel, dim3(cdiv(p->n, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("ndc_rays");
}

int ucnerf_dir_feature(const ucnerf_dir_feature_params* p, void* stream) {
    UCNERF_REQUIRE(p, "dir_feature: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->rays_d && p->angle, "dir_feature: null pointer");
    UCNERF_COUNT(p->n);
    hipLaunchKernelGGL(dir_feature_kernel, dim3(cdiv((long long)p->n * (p->repeat > 1 ? p->repeat : 1), 256)), dim3(256), 0,
                       (hipStream_t)stream, *p);
    return check_launch("dir_feature");
}

int ucnerf_sample_stratified(const ucnerf_sample_stratified_params* p, void* stream) {
    UCNERF_REQUIRE(p, "sample_stratified: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->z, "sample_stratified: null z");
    UCNERF_REQUIRE(p->rays || !p->pts, "sample_stratified: pts output needs the rays array");
    UCNERF_REQUIRE(p->S >= 1, "sample_stratified: S = %d", p->S);
    UCNERF_REQUIRE(!(p->perturb > 0.f) || p->noise, "sample_stratified: perturb > 0 needs noise draws");
    UCNERF_COUNT(p->n);
    hipLaunchKernelGGL(sample_stratified_kernel, dim3(cdiv((long long)p->n * p->S, 256)), dim3(256), 0,
                       (hipStream_t)stream, *p);
    return check_launch("sample_stratified");
}

int ucnerf_sample_cascade(const ucnerf_sample_cascade_params* p, void* stream) {
    UCNERF_REQUIRE(p, "sample_cascade: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->near_far && p->z, "sample_cascade: null pointer");
    UCNERF_REQUIRE(p->S >= 3 && p->S % 3 == 0 && p->S <= 768, "sample_cascade: S = %d (multiple of 3, <= 768)", p->S);
    UCNERF_REQUIRE(!p->pts || (p->rays_o && p->rays_d), "sample_cascade: pts needs rays_o and rays_d");
    UCNERF_COUNT(p->n);
    hipLaunchKernelGGL(sample_cascade_kernel, dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    return check_launch("sample_cascade");
}

int ucnerf_build_rays_test(const ucnerf_build_rays_test_params* p, void* stream) {
    UCNERF_REQUIRE(p, "build_rays_test: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->K && p->c2w && p->w2c_ref && p->K_ref && p->near_far_ref && p->depth_values[0] && p->depth_values[1] && p->depth_values[2],
                   "build_rays_test: null input");
    UCNERF_REQUIRE(p->rays_d && p->z && p->pts && p->ndc1 && p->ndc2 && p->ndc3 && p->ndc, "build_rays_test: null output");
    UCNERF_REQUIRE(p->S >= 3 && p->S % 3 == 0 && p->S <= 768, "build_rays_test: S = %d (multiple of 3, <= 768)", p->S);
    UCNERF_REQUIRE(p->H >= 2 && p->W >= 2 && p->grid_start >= 0 && (long long)p->grid_start + p->n <= (long long)p->H * p->W,
                   "build_rays_test: pixels %d .. %d of a %d x %d image", p->grid_start, p->grid_start + p->n - 1, p->H, p->W);
    for (int k = 0; k < 3; ++k) {
        const int div = 4 >> k;
        UCNERF_REQUIRE(p->dv_d[k] >= 1 && p->dv_h[k] >= (p->H - 1) / div + 1 && p->dv_w[k] >= (p->W - 1) / div + 1,
                       "build_rays_test: depth_values[%d] is %d x %d x %d, the image needs at least %d x %d", k, p->dv_d[k], p->dv_h[k], p->dv_w[k],
                       (p->H - 1) / div + 1, (p->W - 1) / div + 1);
    }
    hipLaunchKernelGGL(build_rays_test_kernel, dim3(p->n), dim3(64), 0, (hipStream_t)stream, *p);
    return check_launch("build_rays_test");
}

int ucnerf_ndc_project(const ucnerf_ndc_project_params* p, void* stream) {
    UCNERF_REQUIRE(p, "ndc_project: null params");
    UCNERF_COUNT(p->m);
    UCNERF_REQUIRE(p->pts, "ndc_project: null pts");
    UCNERF_REQUIRE(p->nf_stride == 0 || p->nf_stride == 1, "ndc_project: nf_stride %d", p->nf_stride);
    if (p->sample_2d) {
        UCNERF_REQUIRE(p->out_ndc, "ndc_project: sample_2d needs out_ndc");
    } else {
        UCNERF_REQUIRE(p->out_stage1 || p->out_stage2 || p->out_stage3 || p->out_ndc, "ndc_project: no outputs");
        UCNERF_REQUIRE((!p->out_stage1 || (p->near_1 && p->far_1)) && (!p->out_stage2 || (p->near_2 && p->far_2)) &&
                           (!p->out_stage3 || (p->near_3 && p->far_3)),
                       "ndc_project: stage output without its near/far arrays");
    }
    UCNERF_COUNT(p->m);
    hipLaunchKernelGGL(ndc_project_kernel, dim3(cdiv(p->m, 256)), dim3(256), 0, (hipStream_t)stream, *p);
    return check_launch("ndc_project");
}

int ucnerf_embed(const ucnerf_embed_params* p, void* stream) {
    UCNERF_REQUIRE(p, "embed: null params");
    UCNERF_COUNT(p->m);
    UCNERF_REQUIRE(p->x && p->out, "embed: null pointer");
    UCNERF_REQUIRE(p->n_freqs >= 0 && p->n_freqs <= 30, "embed: n_freqs %d", p->n_freqs);
    UCNERF_REQUIRE(p->layout == 0 || p->layout == 1, "embed: layout %d", p->layout);
    UCNERF_COUNT(p->m);
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv((long long)p->m * (p->n_freqs + 1), 256)), dim3(256), 0,
                       (hipStream_t)stream, *p, 3 + 6 * p->n_freqs);
    return check_launch("embed");
}

}  // extern "C"
