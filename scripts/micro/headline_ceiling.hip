// Micro-benchmark behind `roofline.frac_of_measured_ceiling`: what the split-bf16 (three MFMAs per product) formulation of the
// headline kernel can reach on THIS board, built up in four steps with the kernel's own instruction ratios, each run long enough for a hwmon
// power / clock reading.  The kernel's ratios per 32-sample tile (profiles/r04_mlp_bf16_fused_hbm_traffic.json): 864 MFMA 32x32x16 bf16, 4262 other
// vector instructions, 938 LDS instructions, 616 scalar ones -- per half-step of 6 MFMAs: 4 ds_read_b128 of weight fragments (+ 2.5 other LDS
// accesses), ~30 vector instructions, ~4 scalar ones; one s_barrier per 8-KB ring slot (two half-steps).  Geometry as the kernel's: 512-thread
// blocks (eight waves, two per SIMD), one block per CU.
//   (i)   bare:     6 MFMAs per half-step, fragments held in registers
//   (ii)  +lds:     the four fragments read from LDS one half-step ahead (ds_read_b128) + 2 more LDS reads and a write every other half-step
//   (iii) +valu:    (ii) + 30 vector instructions per half-step (fma / integer mix, 5 between consecutive MFMAs), 4 scalar
//   (iv)  +barrier: (iii) + one s_barrier per two half-steps (the ring's lock-step across the eight waves)
//   (v)   +1wps:    (iii) at ONE wave per SIMD (256-thread blocks) for reference
// Output: one JSON object per line (config, ms, executed TFLOP/s, mean / max power, mean clock).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/headline_ceiling scripts/micro/headline_ceiling.hip && /tmp/headline_ceiling [seconds per config]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <climits>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SBAR __builtin_amdgcn_sched_barrier(0);

// LDS: 1 = fragments from LDS; VALU: vector instructions between consecutive MFMAs; BAR: s_barrier every two half-steps
template <int LDS, int VALU, int BAR, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? 2 : 1) loop_kernel(const bf16x8* __restrict__ w, float* out, int iters) {
    __shared__ bf16x8 lds[4 * 64 * 8 + 64 * 8];          // 8 half-steps of A fragments (32 KB) + a stash the extra accesses use
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 64 * 8; i += THREADS) lds[i] = w[i];
    __syncthreads();
    const bf16x8 bh0 = w[lane], bl0 = w[64 + lane];
    bf16x8 rh0 = w[lane + 1], rl0 = w[65 + lane], rh1 = w[129 + lane], rl1 = w[193 + lane];
    f32x16 c0 = {}, c1 = {};
    // (per-lane values: uniform ones would be moved to the scalar unit)
    float v[4] = {1.f + 0.001f * lane, 2.f + 0.002f * lane, 3.f - 0.001f * lane, 4.f + 0.003f * lane};
    unsigned u[4] = {0x3f801234u ^ (lane * 2654435761u), 0x40012345u + lane * 40503u, 0x3fc45678u ^ (lane << 7), 0x3e89abcdu + lane};
    const float vm = out[0] * 1e-30f + 1.0001f, va = out[1] * 1e-30f + 0.5f;
    const unsigned um = (unsigned)(out[2] * 1e-30f) | 0xffff0000u;
    bf16x8* stash = lds + 4 * 64 * 8 + wave * 64 + lane;
    int sacc = 0;
    // five vector instructions: 2 fma + 3 integer (the split's and / sub-like ops); chains of four independent values
#define FILL(K)                                                                                                           \
    if (VALU) { SBAR                                                                                                      \
        v[(K) & 3] = __builtin_fmaf(v[(K) & 3], vm, va); v[((K) + 1) & 3] = __builtin_fmaf(v[((K) + 1) & 3], vm, va);   \
        u[(K) & 3] = u[(K) & 3] & um; u[((K) + 2) & 3] ^= u[(K) & 3];                                                   \
        if (VALU > 4) u[((K) + 3) & 3] = __builtin_amdgcn_perm(u[((K) + 3) & 3], u[(K) & 3], 0x07060302u);               \
        SBAR }
    // fragments one half-step ahead, as in the kernel: the reads of half-step hs + 1 are issued before the MFMAs of half-step hs
    bf16x8 n0 = rh0, n1 = rl0, n2 = rh1, n3 = rl1;
    auto frag_reads = [&](int hs) {
        int off = 0;
        asm volatile("" : "+v"(off));                  // (opaque: the reads stay inside the loop)
        const bf16x8* a = lds + (hs & 7) * 256 + lane + off;
        n0 = a[0]; n1 = a[64]; n2 = a[128]; n3 = a[192];
        if (hs & 1) { const bf16x8 t = stash[0]; stash[0] = n0; rh0 = t; }      // the activation stash traffic: ~2.5 accesses per half-step
        else { rl0 = stash[0]; rl1 = lds[(hs * 64 + lane) & 1023]; }
    };
    if (LDS) frag_reads(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int hs = 0; hs < 8; ++hs) {
            const bf16x8 h0 = n0, l0 = n1, h1 = n2, l1 = n3;
            if (LDS) { SBAR frag_reads(hs + 1); SBAR }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bh0, c0, 0, 0, 0); FILL(0)
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bl0, c0, 0, 0, 0); FILL(1)
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, bh0, c0, 0, 0, 0); FILL(2)
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bh0, c1, 0, 0, 0); FILL(3)
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bl0, c1, 0, 0, 0); FILL(4)
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, bh0, c1, 0, 0, 0); FILL(5)
            if (VALU) { sacc += hs; sacc ^= it; sacc = sacc * 3 + 1; sacc &= 0xffff; }
            if (BAR && (hs & 1)) __builtin_amdgcn_s_barrier();
        }
    }
    float s = (float)sacc;
    for (int r = 0; r < 4; ++r) s += v[r] + __uint_as_float(u[r] & 0x3fffffffu);
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    s += (float)rh0[0] + (float)rl0[0] + (float)rh1[0] + (float)rl1[0];
    out[8 + blockIdx.x * THREADS + threadIdx.x] = s;
}

// hwmon directory of THE card this process computes on (matched by PCI bus id: the host's other cards run other people's jobs)
struct Hwmon {
    std::vector<std::string> dirs;
    Hwmon() {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess) bus[0] = 0;
        for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
        for (int c = 0; c < 128; ++c) {
            const std::string dev = "/sys/class/drm/card" + std::to_string(c) + "/device";
            char real[512];
            if (!realpath(dev.c_str(), real)) continue;
            std::string r(real);
            for (auto& ch : r) ch = (char)tolower(ch);
            if (bus[0] && r.find(bus) == std::string::npos) continue;
            const std::string base = dev + "/hwmon";
            DIR* d = opendir(base.c_str());
            if (!d) continue;
            while (dirent* e = readdir(d))
                if (std::string(e->d_name).rfind("hwmon", 0) == 0) dirs.push_back(base + "/" + e->d_name);
            closedir(d);
        }
        fprintf(stderr, "device 0 = PCI %s, %zu hwmon director%s\n", bus, dirs.size(), dirs.size() == 1 ? "y" : "ies");
    }
    static long rd(const std::string& p) {
        FILE* f = fopen(p.c_str(), "r");
        if (!f) return -1;
        long v = -1;
        if (fscanf(f, "%ld", &v) != 1) v = -1;
        fclose(f);
        return v;
    }
};

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    const int n = 4 * 64 * 8;
    std::vector<unsigned short> h((n + 256) * 8);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));    // random mantissas, |x| ~ 0.01..0.03
    bf16x8* w; float* out;
    hipMalloc(&w, (n + 256) * 16); hipMalloc(&out, (8 + 1024 * 512) * 4);
    hipMemcpy(w, h.data(), (n + 256) * 16, hipMemcpyHostToDevice);
    hipMemset(out, 0, 64);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Hwmon hw;
    struct Cfg { const char* name; int id; int threads; };
    const Cfg cfgs[] = {{"i_bare_3mfma", 0, 512}, {"ii_plus_lds_fragments", 1, 512}, {"iii_plus_valu_5_per_mfma", 2, 512}, {"iv_plus_barrier_per_slot", 3, 512},
                        {"v_iii_one_wave_per_simd", 4, 256}, {"vi_iv_one_wave_per_simd", 5, 256}};
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep)
    for (const Cfg& c : cfgs) {
        auto launch = [&]() {
            switch (c.id) {
                case 0: loop_kernel<0, 0, 0, 512><<<cus, 512>>>(w, out, iters); break;
                case 1: loop_kernel<1, 0, 0, 512><<<cus, 512>>>(w, out, iters); break;
                case 2: loop_kernel<1, 5, 0, 512><<<cus, 512>>>(w, out, iters); break;
                case 3: loop_kernel<1, 5, 1, 512><<<cus, 512>>>(w, out, iters); break;
                case 4: loop_kernel<1, 5, 0, 256><<<cus, 256>>>(w, out, iters); break;
                case 5: loop_kernel<1, 5, 1, 256><<<cus, 256>>>(w, out, iters); break;
            }
        };
        launch(); hipDeviceSynchronize();
        std::vector<long> idle_p;
        for (auto& d : hw.dirs) idle_p.push_back(Hwmon::rd(d + "/power1_input"));
        std::atomic<bool> stop(false);
        std::vector<std::vector<long>> ps(hw.dirs.size()), fs(hw.dirs.size());
        std::thread sampler([&]() {
            while (!stop.load()) {
                for (size_t i = 0; i < hw.dirs.size(); ++i) { ps[i].push_back(Hwmon::rd(hw.dirs[i] + "/power1_input")); fs[i].push_back(Hwmon::rd(hw.dirs[i] + "/freq1_input")); }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
        const auto t0 = std::chrono::steady_clock::now();
        double ms_sum = 0; int launches = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            hipEventRecord(a);
            for (int k = 0; k < 4; ++k) launch();
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.6) { ms_sum += ms; launches += 4; }     // skip the ramp
        }
        stop.store(true); sampler.join();
        // the card whose power rose
        int card = -1; double best = -1e30;
        std::vector<double> mean_p(hw.dirs.size(), 0.0);
        for (size_t i = 0; i < hw.dirs.size(); ++i) {
            const size_t skip = ps[i].size() / 4;
            double s = 0; int cnt = 0;
            for (size_t k = skip; k < ps[i].size(); ++k) if (ps[i][k] >= 0) { s += ps[i][k]; ++cnt; }
            mean_p[i] = cnt ? s / cnt : 0;
            if (mean_p[i] - idle_p[i] > best) { best = mean_p[i] - idle_p[i]; card = (int)i; }
        }
        double pmax = 0, fmean = 0; int fc = 0;
        if (card >= 0) {
            const size_t skip = ps[card].size() / 4;
            for (size_t k = skip; k < ps[card].size(); ++k) { if (ps[card][k] > pmax) pmax = ps[card][k]; if (fs[card][k] > 0) { fmean += fs[card][k]; ++fc; } }
        }
        const double ms = launches ? ms_sum / launches : 0;
        const double waves = (double)cus * (c.threads / 64);
        const double flop = waves * iters * 8 * 6 * 32768.0;        // executed MFMA flops (32x32x16: 32768 per instruction and wave)
        printf("{\"config\": \"%s\", \"rep\": %d, \"waves_per_simd\": %d, \"ms_per_launch\": %.4f, \"executed_tflops\": %.1f, \"algorithmic_tflops\": %.1f, "
               "\"frac_of_2500\": %.4f, \"power_w_mean\": %.1f, \"power_w_max\": %.1f, \"power_cap_w\": %.1f, \"sclk_mhz_mean\": %.0f, \"launches\": %d}\n",
               c.name, rep, c.threads / 256, ms, flop / ms / 1e9, flop / 3 / ms / 1e9, flop / ms / 1e9 / 2500.0, card >= 0 ? mean_p[card] / 1e6 : 0.0, pmax / 1e6,
               card >= 0 ? Hwmon::rd(hw.dirs[card] + "/power1_cap") / 1e6 : 0.0, fc ? fmean / fc / 1e6 : 0.0, launches);
        fflush(stdout);
    }
    return 0;
}
