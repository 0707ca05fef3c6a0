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


// Second question: how long does a freed slot stay empty?  Many rounds of workgroups with spread-out lives (spin 20-50 us by a hash of
// the block index, or a constant); gap = a workgroup's start minus the matched end on its XCD (block b runs on XCD b % 8; starts
// after the first `slots` are matched, in order, with the sorted ends).
template <int REGS>
__global__ void probe_lives(unsigned long long* out, int spin_lo, int spin_span) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned h = blockIdx.x * 2654435761u;
    h ^= h >> 15;
    const int spin_ticks = spin_lo + (spin_span ? (int)(h % (unsigned)spin_span) : 0);
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
    if (s == 12345.678f) lds[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t0;
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int REGS>
void run_lives(const char* name, int blocks, int threads, size_t lds, int lo_us, int span_us, int slots_per_xcd) {
    unsigned long long* d;
    hipMalloc(&d, blocks * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_lives<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0, blocks * 16);
        hipLaunchKernelGGL(probe_lives<REGS>, dim3(blocks), dim3(threads), lds, 0, d, lo_us * 100, span_us * 100);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> gaps;
    double life = 0.0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); life += (h[2 * i + 1] - h[2 * i]) * 0.01; }
    for (int x = 0; x < 8; ++x) {
        std::vector<unsigned long long> b, e;
        for (int i = x; i < blocks; i += 8) { b.push_back(h[2 * i]); e.push_back(h[2 * i + 1]); }
        std::sort(b.begin(), b.end());
        std::sort(e.begin(), e.end());
        for (size_t i = slots_per_xcd; i < b.size(); ++i) gaps.push_back(((double)b[i] - (double)e[i - slots_per_xcd]) * 0.01);
    }
    std::sort(gaps.begin(), gaps.end());
    double mean = 0.0;
    for (double g : gaps) mean += g;
    mean /= gaps.size();
    printf("%-40s blocks %5d lds %6zu lives %2d + %2d us: gap p10 %5.2f p50 %5.2f p90 %5.2f mean %5.2f us; span %7.1f us, sum of lives / slots %7.1f us\n", name, blocks, lds,
           lo_us, span_us, gaps[gaps.size() / 10], gaps[gaps.size() / 2], gaps[9 * gaps.size() / 10], mean, (t1 - t0) * 0.01, life / (8.0 * slots_per_xcd));
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
    run_lives<96>("512 thr, 73 KB, ~100 regs, spread lives", 4096, 512, 74624, 20, 30, 64);
    run_lives<96>("512 thr, 73 KB, ~100 regs, equal lives", 4096, 512, 74624, 35, 0, 64);
    run_lives<96>("512 thr, 73 KB, ~100 regs, 33 + 3", 4096, 512, 74624, 33, 3, 64);
    run_lives<96>("512 thr, 73 KB, ~100 regs, 32 + 6", 4096, 512, 74624, 32, 6, 64);
    run_lives<96>("512 thr, 73 KB, ~100 regs, 28 + 14", 4096, 512, 74624, 28, 14, 64);
    run_lives<8>("512 thr, 73 KB, few regs, spread lives", 4096, 512, 74624, 20, 30, 64);
    run_lives<8>("512 thr, 1 KB, few regs, spread (8 WG/CU)", 16384, 512, 1024, 20, 30, 128);
    run_lives<8>("256 thr, 37 KB, few regs, spread (4 WG/CU)", 8192, 256, 37000, 20, 30, 128);
    return 0;
}
