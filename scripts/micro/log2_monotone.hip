// Developer check (GPU box): is the dB chain of the kernel-product finish — 3.0103 * v_log_f32(max(p, 1e-12)) - ref — monotone non-decreasing
// over EVERY non-negative finite float?  blockdft_banddots4c_db<FAST> takes a frame's extreme dB values as to_db of its extreme powers
// (band_finish_fast); that equals the maximum / minimum over the bins' own to_db values iff this holds.
// build + run:  hipcc --offload-arch=gfx950 -O3 scripts/micro/log2_monotone.hip -o /tmp/log2_monotone && /tmp/log2_monotone
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float to_db(float p) {
    const float ref_db = 10.0f * log10f(0.3f * 0.3f);
    return 3.01029995663981f * __log2f(fmaxf(p, 1e-6f * 1e-6f)) - ref_db;
}
__global__ void check(unsigned long long* bad, unsigned* first_bad) {
    const unsigned long long n = 0x7F800000ull;   // bit patterns 0 .. +INF (exclusive): every non-negative finite float
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i + 1 < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __builtin_bit_cast(float, (unsigned)i), b = __builtin_bit_cast(float, (unsigned)(i + 1));
        if (!(to_db(a) <= to_db(b))) {
            atomicAdd(bad, 1ull);
            atomicMin(first_bad, (unsigned)i);
        }
    }
}
int main() {
    unsigned long long* d_bad; unsigned* d_first;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4);
    hipMemset(d_bad, 0, 8); hipMemset(d_first, 0xFF, 4);
    check<<<4096, 256>>>(d_bad, d_first);
    unsigned long long bad = 0; unsigned first = 0;
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("to_db over all %llu non-negative finite floats: %llu adjacent pairs out of order%s\n", 0x7F800000ull, bad, bad ? "" : " (monotone)");
    if (bad) printf("first at bit pattern 0x%08x\n", first);
    return bad ? 1 : 0;
}
