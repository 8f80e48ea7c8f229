// membw3.hip -- what makes a 2.12 GB write fast?  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// every thread writes K consecutive float4 (a wave covers K KiB contiguous), blocks tile the buffer
template <int K>
__global__ void write_wave_contig(f4 *p, size_t n4, float val)
{
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    f4 v = {val, val, val, val};
    f4 *q = p + wave * (64 * K) + lane;
    if ((wave + 1) * 64 * K <= n4) {
#pragma unroll
        for (int k = 0; k < K; k++) q[k * 64] = v;
    }
}
// pass-A shape with value pattern from registers
__global__ void write_rows_f1(float *p, int W, int Dp, float val)
{
    float *t = p + (size_t)blockIdx.x * W * Dp + threadIdx.x;
    float acc = val;
    for (int x = 0; x < W; x++) { acc += val; t[(size_t)x * Dp] = acc; }
}
// pass-A shape but one block handles R consecutive image rows one after the other
__global__ void write_rows_seq(float *p, int W, int Dp, int R, float val)
{
    float acc = val;
    for (int r = 0; r < R; r++) {
        float *t = p + ((size_t)blockIdx.x * R + r) * W * Dp + threadIdx.x;
        for (int x = 0; x < W; x++) { acc += val; t[(size_t)x * Dp] = acc; }
    }
}
template <class F> void timeit(const char *name, double bytes, F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
}
int main()
{
    const int W = 1920, H = 1080, Dp = 256;
    size_t n = (size_t)W * H * Dp;
    float *a; CK(hipMalloc(&a, n * 4));
    double B = (double)n * 4;
    timeit("memset 0", B, [&] { CK(hipMemsetAsync(a, 0, n * 4)); });
    timeit("memsetD32 1.0f", B, [&] { CK(hipMemsetD32Async((hipDeviceptr_t)a, 0x3f800000, n)); });
    size_t n4 = n / 4;
#define WC(K) { size_t waves = n4 / (64 * K); timeit("wave-contig K=" #K " zeros", B, [&] { write_wave_contig<K><<<(unsigned)((waves + 3) / 4), 256>>>((f4 *)a, n4, 0.f); }); \
                timeit("wave-contig K=" #K " ones", B, [&] { write_wave_contig<K><<<(unsigned)((waves + 3) / 4), 256>>>((f4 *)a, n4, 1.f); }); }
    WC(1) WC(4) WC(16)
    timeit("rows f1 val=0", B, [&] { write_rows_f1<<<H, Dp>>>(a, W, Dp, 0.f); });
    timeit("rows f1 val=1.37", B, [&] { write_rows_f1<<<H, Dp>>>(a, W, Dp, 1.37f); });
    timeit("rows seq R=2 (540 blocks)", B, [&] { write_rows_seq<<<H / 2, Dp>>>(a, W, Dp, 2, 1.37f); });
    timeit("rows seq R=4 (270 blocks)", B, [&] { write_rows_seq<<<H / 4, Dp>>>(a, W, Dp, 4, 1.37f); });
    return 0;
}
