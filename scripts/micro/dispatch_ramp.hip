// Developer micro-benchmark: how fast does the chip start the workgroups of a launch?  Every workgroup stores the 100 MHz clock at its
// start and at its end (after spinning `spin_us`), for several (threads, LDS bytes, VGPR pressure) shapes.
// build: hipcc -O3 --offload-arch=gfx950 dispatch_ramp.hip -o dispatch_ramp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int REGS>
__global__ void probe(unsigned long long* out, int spin_ticks) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float keep[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) keep[i] = threadIdx.x * 0.5f + i;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += keep[i];
    if (s == 12345.678f) lds[threadIdx.x] = s;   // keep the registers live
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t0;
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int REGS>
void run(const char* name, int blocks, int threads, size_t lds, int spin_us) {
    unsigned long long* d;
    hipMalloc(&d, blocks * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0, blocks * 16);
        hipLaunchKernelGGL(probe<REGS>, dim3(blocks), dim3(threads), lds, 0, d, spin_us * 100);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> st(blocks);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); }
    for (int i = 0; i < blocks; ++i) st[i] = (h[2 * i] - t0) * 0.01;
    std::sort(st.begin(), st.end());
    printf("%-34s blocks %5d threads %4d lds %6zu spin %3d us: start p10 %6.2f p25 %6.2f p50 %6.2f p75 %6.2f p90 %6.2f max %6.2f us, span %7.2f us\n", name, blocks, threads,
           lds, spin_us, st[blocks / 10], st[blocks / 4], st[blocks / 2], st[3 * blocks / 4], st[9 * blocks / 10], st[blocks - 1], (t1 - t0) * 0.01);
    hipFree(d);
}

int main() {
    run<8>("512 thr, 66 KB, few regs", 512, 512, 66560, 50);
    run<8>("512 thr, 66 KB, few regs, 2 rounds", 1024, 512, 66560, 50);
    run<8>("512 thr, 0 KB, few regs", 512, 512, 0, 50);
    run<96>("512 thr, 66 KB, ~100 regs", 512, 512, 66560, 50);
    run<96>("512 thr, 71 KB, ~100 regs (gemm)", 512, 512, 73000, 25);
    run<96>("512 thr, 71 KB, gemm-like 5852", 5852, 512, 73000, 25);
    run<8>("256 thr, 33 KB, few regs", 1024, 256, 33280, 50);
    run<8>("64 thr, 8 KB, few regs", 4096, 64, 8192, 50);
    return 0;
}
