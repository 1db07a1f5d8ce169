// Where does global_load_lds_dwordx3 put its bytes?  One wave copies 768 bytes of a counting pattern; the LDS is dumped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds;
    const char* s = (const char*)src + threadIdx.x * 12;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx3 %0, off" ::"v"(s), "s"(dst) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(512);
    for (int i = 0; i < 512; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 2048); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, d, o);
    hipMemcpy(h.data(), o, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; ++i) printf("%x%c", h[i], (i & 15) == 15 ? '\n' : ' ');
    return 0;
}
