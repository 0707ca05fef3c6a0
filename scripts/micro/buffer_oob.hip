// What a 16-byte raw buffer load returns when it lies partly (or wholly) outside the buffer on gfx950 — below offset 0 (the offset
// wraps as a 32-bit unsigned) or beyond num_records.  OOB buffer loads never fault: they return 0.  Question: per dword, or the whole load?
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/buffer_oob.hip -o scripts/micro/buffer_oob ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 raw_load_f32x4(i32x4 srsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

__global__ void probe(const float* buf, int n_bytes, const int* offs, int n_offs, float* out, int use_soffset) {
    const unsigned long long a = (unsigned long long)buf;
    const i32x4 rsrc = {(int)(unsigned)a, (int)(unsigned)(a >> 32), n_bytes, 0x00020000};
    const int i = threadIdx.x;
    if (i < n_offs) {
        // use_soffset: the negative part rides in the scalar offset, a non-negative remainder in the vector offset (how the K loop addresses)
        const f32x4 v = use_soffset ? raw_load_f32x4(rsrc, 64, __builtin_amdgcn_readfirstlane(offs[0]) - 64 + (offs[i] - offs[0]) * 0, 0) : raw_load_f32x4(rsrc, offs[i], 0, 0);
        out[4 * i + 0] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
    }
}
int main() {
    const int N = 64;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = 100.0f + i;
    float* d; hipMalloc(&d, 3 * N * sizeof(float));
    std::vector<float> pad(3 * N, -7.0f);   // the buffer proper sits in the middle third: what lies outside it is -7, not 0
    for (int i = 0; i < N; ++i) pad[N + i] = h[i];
    hipMemcpy(d, pad.data(), pad.size() * 4, hipMemcpyHostToDevice);
    const int offs_h[] = {-16, -12, -8, -4, 0, 4, N * 4 - 16, N * 4 - 12, N * 4 - 8, N * 4 - 4, N * 4, N * 4 + 4, -32, -20};
    const int n = sizeof(offs_h) / sizeof(int);
    int* d_offs; hipMalloc(&d_offs, sizeof(offs_h)); hipMemcpy(d_offs, offs_h, sizeof(offs_h), hipMemcpyHostToDevice);
    float* d_out; hipMalloc(&d_out, n * 16);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d + N, N * 4, d_offs, n, d_out, 0);
    std::vector<float> o(4 * n);
    hipMemcpy(o.data(), d_out, n * 16, hipMemcpyDeviceToHost);
    printf("buffer: %d floats 100 .. %d, num_records %d bytes; memory around it holds -7\n", N, 100 + N - 1, N * 4);
    for (int i = 0; i < n; ++i) printf("byte offset %5d: %7.1f %7.1f %7.1f %7.1f\n", offs_h[i], o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    return 0;
}
