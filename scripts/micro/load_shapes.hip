// Micro-benchmark: what does the CU's vector-memory path make of the weight-gradient launch's fetch shapes?
// A block of 512 threads runs the skeleton of a mlp_wgrad_kernel stage -- eight loads per thread, wait, two barriers -- over
// a range of 64-row stages of 384-byte rows, with the per-lane shape a template parameter:
//   0  16-byte loads at a 12-byte stride (4-byte aligned; the kernel's fetch of 24-bit pieces: lanes 0-31 one row, 32-63 the row 8 below)
//   1  12-byte loads (dwordx3) at the same addresses
//   2  16-byte aligned loads at a 16-byte stride (512-byte rows: the fp32-rows operand)
//   3  12-byte loads, the wave's 64 lanes on 768 contiguous bytes (two adjacent rows)
//   4  16-byte aligned loads, the wave's 64 lanes on 1024 contiguous bytes (a plain copy of the stage's bytes)
// over a region that stays in L2 (4 MB) or streams from HBM (768 MB).  Output: microseconds, bytes per clock and CU, TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/load_shapes scripts/micro/load_shapes.hip && /tmp/load_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));

template <int SHAPE>
__global__ void __launch_bounds__(512, 4) stage_kernel(const char* __restrict__ base, unsigned region_mask, int stages, unsigned* out, unsigned long long* dur) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int t8 = tid & 255, side = tid >> 8;
    const int grp = t8 & 31, oct = t8 >> 5;
    unsigned acc = 0;
    constexpr unsigned ROW = SHAPE == 2 ? 512u : 384u;
    constexpr unsigned STAGE = 64u * ROW;
    for (int st = 0; st < stages; ++st) {
        // the two sides (G, X) of a stage are two different sets: 2 * STAGE bytes per block and stage
        const unsigned s0 = ((unsigned)(blockIdx.x * stages + st) * 2u + side) * STAGE;
        unsigned off[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (SHAPE <= 2) off[e] = s0 + (8u * oct + e) * ROW + grp * (SHAPE == 2 ? 16u : 12u);
            else if (SHAPE == 3) off[e] = s0 + (unsigned)e * 3072u + t8 * 12u;                 // 256 threads x 12 B = 3 KB per e
            else off[e] = s0 + (unsigned)e * 4096u + t8 * 16u;                                  // 256 threads x 16 B = 4 KB per e (6 of 8 carry the stage)
        }
        u32x4 d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const char* p = base + (off[e] & region_mask);
            if (SHAPE == 1 || SHAPE == 3) {
                const u32x3 v = *reinterpret_cast<const u32x3_a4*>(p);
                d[e] = (u32x4){v[0], v[1], v[2], 0u};
            } else if (SHAPE == 0) d[e] = *reinterpret_cast<const u32x4_a4*>(p);
            else d[e] = *reinterpret_cast<const u32x4*>(p);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) acc ^= d[e][0] ^ d[e][1] ^ d[e][2] ^ d[e][3];
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (tid == 0) dur[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;      // 100 MHz ticks
}

template <int SHAPE>
static void run(const char* name, const char* buf, size_t bytes, unsigned* out, int cus, double ghz) {
    static unsigned long long* dur = nullptr;
    if (!dur) hipMalloc(&dur, 4096 * 8);
    const int blocks = 2 * cus;
    const unsigned row = SHAPE == 2 ? 512u : 384u;
    for (int resident = 1; resident >= 0; --resident) {
        const int stages = 60;
        const size_t need = (size_t)blocks * stages * 2 * 64 * row;
        if (!resident && need > bytes) { printf("buffer too small\n"); exit(1); }
        const unsigned mask = resident ? 0x3ffffcu : 0xfffffffcu;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(stage_kernel<SHAPE>, dim3(blocks), dim3(512), 0, 0, buf, mask, stages, out, dur);
        hipEventRecord(a, 0);
        const int reps = 10;
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(stage_kernel<SHAPE>, dim3(blocks), dim3(512), 0, 0, buf, mask, stages, out, dur);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / reps;
        const double useful = (double)blocks * stages * 2 * 64 * (SHAPE == 4 ? 512.0 : (double)row);   // shape 4: 8 x 4 KB per side and stage
        printf("%-46s %-8s %8.1f us  %6.2f us/stage/block  %6.1f B/clk/CU (at %.1f GHz)  %5.2f TB/s\n", name, resident ? "L2" : "HBM", us, us / stages,
               useful / (us * 1e-6) / cus / (ghz * 1e9), ghz, useful / (us * 1e-6) / 1e12);
        // per-block durations by blockIdx % 8 (the XCD a block is dispatched to): do the eight dies stream at the same rate?
        static unsigned long long h[4096];
        hipMemcpy(h, dur, blocks * 8, hipMemcpyDeviceToHost);
        printf("      block duration by blockIdx %% 8 (us):");
        for (int x = 0; x < 8; ++x) {
            double sum = 0; int n = 0;
            for (int b = x; b < blocks; b += 8) { sum += h[b] * 0.01; ++n; }
            printf(" %.0f", sum / n);
        }
        printf("\n");
    }
}


// The weight-gradient launch's geometry: `pairs` pairs of two sets of `m` rows of 384 bytes, `spacing` bytes from set to set; the concatenated
// stages cut into equal ranges, one block per range, each block streaming its range of its pair's two sets.  Per-XCD durations as above.
__global__ void __launch_bounds__(512, 2) geometry_kernel(const char* __restrict__ base, int pairs, int stages_per_pair, size_t spacing, unsigned* out, unsigned long long* dur) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, t8 = tid & 255, side = tid >> 8, grp = t8 & 31, oct = t8 >> 5;
    const long long total = (long long)pairs * stages_per_pair;
    long long g0 = total * blockIdx.x / gridDim.x, g1 = total * (blockIdx.x + 1) / gridDim.x;
    unsigned acc = 0;
    for (long long g = g0; g < g1; ++g) {
        const int p = (int)(g / stages_per_pair), st = (int)(g % stages_per_pair);
        const char* set = base + (size_t)(2 * p + side) * spacing;
        u32x4 d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = *reinterpret_cast<const u32x4_a4*>(set + ((size_t)st * 64 + 8 * oct + e) * 384 + grp * 12);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) acc ^= d[e][0] ^ d[e][1] ^ d[e][2] ^ d[e][3];
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (tid == 0) dur[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

static void run_geometry(const char* buf, size_t bytes, unsigned* out, int cus, size_t spacing, int blocks_per_cu) {
    static unsigned long long* dur = nullptr;
    if (!dur) hipMalloc(&dur, 4096 * 8);
    const int pairs = 15, m = 131072, stages = m / 64, blocks = cus * blocks_per_cu;
    if ((size_t)2 * pairs * spacing + 4096 > bytes) { printf("buffer too small for spacing %zu\n", spacing); return; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(geometry_kernel, dim3(blocks), dim3(512), 0, 0, buf, pairs, stages, spacing, out, dur);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(geometry_kernel, dim3(blocks), dim3(512), 0, 0, buf, pairs, stages, spacing, out, dur);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, useful = (double)pairs * 2 * m * 384;
    printf("geometry: 15 pairs x 2 sets of 131072 x 384 B, %zu B from set to set, %d block(s) per CU: %8.1f us  %5.2f TB/s\n", spacing, blocks_per_cu, us, useful / (us * 1e-6) / 1e12);
    static unsigned long long h[4096];
    hipMemcpy(h, dur, blocks * 8, hipMemcpyDeviceToHost);
    printf("      block duration by blockIdx %% 8 (us):");
    for (int x = 0; x < 8; ++x) {
        double sum = 0; int n = 0;
        for (int bb = x; bb < blocks; bb += 8) { sum += h[bb] * 0.01; ++n; }
        printf(" %.0f", sum / n);
    }
    printf("\n");
}

// every byte of the buffer rewritten (what the forward and the gradient chain do to the sets before the weight-gradient launch reads them)
template <int NT>
__global__ void __launch_bounds__(256) rewrite_kernel(u32x4* buf, size_t n16, unsigned v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 x = (u32x4){v, v + 1, v + 2, v + 3};
        if (NT) __builtin_nontemporal_store(x, buf + i);
        else buf[i] = x;
    }
}
// a launch that keeps the CUs busy for a while without touching memory (does dirty data drain to HBM behind it?)
__global__ void __launch_bounds__(256) spin_kernel(unsigned* out, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    if (v == 12345.f) out[0] = 1;
}
template <int NT, int SPIN>
static void run_geometry_after_rewrite(char* buf, size_t bytes, unsigned* out, int cus) {
    static unsigned long long* dur = nullptr;
    if (!dur) hipMalloc(&dur, 4096 * 8);
    const int pairs = 15, m = 131072, stages = m / 64;
    const size_t spacing = (size_t)m * 384 + 384, used = (size_t)2 * pairs * spacing;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float tot = 0, wtot = 0;
    const int reps = 5;
    for (int w = 0; w < reps + 1; ++w) {
        hipEvent_t c, d;
        hipEventCreate(&c); hipEventCreate(&d);
        hipEventRecord(c, 0);
        hipLaunchKernelGGL(rewrite_kernel<NT>, dim3(cus * 8), dim3(256), 0, 0, (u32x4*)buf, used / 16, (unsigned)w);
        hipEventRecord(d, 0);
        if (SPIN) hipLaunchKernelGGL(spin_kernel, dim3(cus * 8), dim3(256), 0, 0, out, SPIN);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(geometry_kernel, dim3(cus), dim3(512), 0, 0, buf, pairs, stages, spacing, out, dur);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (w) tot += ms;
        hipEventElapsedTime(&ms, c, d);
        if (w) wtot += ms;
    }
    const double us = tot * 1e3 / reps, useful = (double)pairs * 2 * m * 384;
    printf("geometry right after every byte was rewritten by another kernel (%s stores: %.1f us, %.2f TB/s), 1 block per CU: %8.1f us  %5.2f TB/s\n", NT ? (SPIN ? "non-temporal, then a memory-free launch of ~150 us," : "non-temporal") : (SPIN ? "plain, then a memory-free launch of ~150 us," : "plain"),
           wtot * 1e3 / reps, (double)used / (wtot * 1e-3 / reps) / 1e12, us, useful / (us * 1e-6) / 1e12);
}

// The gradient chain's memory side alone: one wave per SIMD (4-wave blocks, one per CU), a wave walks 32-sample tiles; per tile it reads NR sets
// and writes NW sets of [m,128] 24-bit rows the way the chain does -- lane (sample j, half h) moves the 12-byte piece of column group 2 q + h of
// its own row, sixteen instructions per set -- with nothing else to do.  Reads and writes go to different sets.
template <int NR, int NW, int TILED>
__global__ void __launch_bounds__(256, 1) chain_traffic_kernel(char* __restrict__ base, int n_tiles, size_t spacing, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    unsigned acc = 0;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        // rows: the lane's own row, piece 2 q + h;  TILED: [tile][group 2 q + h][sample j][12 bytes] -- 768 contiguous bytes per instruction
        const size_t row = TILED ? (size_t)tile * 32 * 384 + (size_t)h * 384 + 12 * j : ((size_t)tile * 32 + j) * 384 + 12 * h;
        constexpr int QS = TILED ? 768 : 24;
#pragma unroll 1
        for (int s = 0; s < NR; ++s) {
            const char* set = base + (size_t)s * spacing + row;
            u32x3 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const u32x3_a4*>(set + QS * q);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc ^= v[q][0] ^ v[q][1] ^ v[q][2];
        }
#pragma unroll 1
        for (int s = 0; s < NW; ++s) {
            char* set = base + (size_t)(NR + s) * spacing + row;
#pragma unroll
            for (int q = 0; q < 16; ++q) *reinterpret_cast<u32x3_a4*>(set + QS * q) = (u32x3){acc + q, acc ^ (unsigned)s, acc};
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int NR, int NW, int TILED>
static void run_chain_traffic(char* buf, size_t bytes, unsigned* out, int cus) {
    const int m = 131072, n_tiles = m / 32;
    const size_t spacing = (size_t)m * 384 + 384;
    if ((size_t)(NR + NW) * spacing > bytes) { printf("buffer too small\n"); return; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((chain_traffic_kernel<NR, NW, TILED>), dim3(cus), dim3(256), 0, 0, buf, n_tiles, spacing, out);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((chain_traffic_kernel<NR, NW, TILED>), dim3(cus), dim3(256), 0, 0, buf, n_tiles, spacing, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, moved = (double)(NR + NW) * m * 384;
    printf("gradient chain's traffic alone (one wave per SIMD, %s): %d sets read, %d written: %8.1f us  %5.2f TB/s\n", TILED ? "tiled sets: 768 contiguous bytes per instruction" : "12-byte pieces of the lane's own row", NR, NW, us, moved / (us * 1e-6) / 1e12);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const size_t bytes = (size_t)2 * cus * 60 * 2 * 64 * 512 + 4096;
    char* buf;
    unsigned* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    hipMemset(out, 0, 64);
    printf("%d CUs, %.2f GHz nominal, %zu MB buffer\n", cus, ghz, bytes >> 20);
    run<0>("0: 16-B loads, 12-B stride, 4-B aligned", buf, bytes, out, cus, ghz);
    run<1>("1: 12-B loads, 12-B stride", buf, bytes, out, cus, ghz);
    run<2>("2: 16-B aligned loads, 16-B stride", buf, bytes, out, cus, ghz);
    run<3>("3: 12-B loads, wave on 768 contiguous bytes", buf, bytes, out, cus, ghz);
    run<4>("4: 16-B aligned loads, wave on 1 KB contiguous", buf, bytes, out, cus, ghz);
    const size_t set = (size_t)131072 * 384;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run_geometry(buf, bytes, out, cus, set, bpc);                 // back to back
        run_geometry(buf, bytes, out, cus, set + 96 * 4, bpc);        // the workspace's spacing (96 floats of padding)
        run_geometry(buf, bytes, out, cus, set + 4096 + 384, bpc);
        run_geometry(buf, bytes, out, cus, set + (1 << 20) + 12288, bpc);
    }
    run_geometry_after_rewrite<0, 0>(buf, bytes, out, cus);
    run_geometry_after_rewrite<1, 0>(buf, bytes, out, cus);
    run_geometry_after_rewrite<0, 60000>(buf, bytes, out, cus);
    run_geometry_after_rewrite<0, 0>(buf, bytes, out, cus);
    run_chain_traffic<9, 0, 0>(buf, bytes, out, cus);
    run_chain_traffic<0, 12, 0>(buf, bytes, out, cus);
    run_chain_traffic<9, 12, 0>(buf, bytes, out, cus);
    run_chain_traffic<9, 0, 1>(buf, bytes, out, cus);
    run_chain_traffic<0, 12, 1>(buf, bytes, out, cus);
    run_chain_traffic<9, 12, 1>(buf, bytes, out, cus);
    return 0;
}
