// membw5.hip -- pass-A write order (x serial, all rows concurrent, banded slab TY = 32) with
// different store widths per wave: 256 B (one float per lane, today's pass A), 1 KiB (float4 per
// lane, the four waves of a row's workgroup taking every fourth x), and one wave per row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void write_A1(float *p, int W, int Dp, int TY, float val)
{
    int y = blockIdx.x;
    float *t = p + ((size_t)(y / TY) * W) * TY * Dp + (size_t)(y % TY) * Dp + threadIdx.x;
    size_t xs = (size_t)TY * Dp;
    float acc = val;
    for (int x = 0; x < W; x++) { acc += val; t[xs * x] = acc; }
}
// 4 waves per row, wave w stores x = 4i + w, 256 levels as float4 per lane
__global__ void write_A4(float *p, int W, int Dp, int TY, float val)
{
    int y = blockIdx.x;
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *t = p + ((size_t)(y / TY) * W) * TY * Dp + (size_t)(y % TY) * Dp + lane * 4;
    size_t xs = (size_t)TY * Dp;
    f4 acc = {val, val, val, val};
    for (int x = w; x < W; x += 4) { acc += val; *(f4 *)(t + xs * x) = acc; }
}
// same, with a workgroup barrier every step (what an LDS transpose would add)
__global__ void write_A4s(float *p, int W, int Dp, int TY, float val)
{
    __shared__ float ex[4][256];
    int y = blockIdx.x;
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *t = p + ((size_t)(y / TY) * W) * TY * Dp + (size_t)(y % TY) * Dp + lane * 4;
    size_t xs = (size_t)TY * Dp;
    float acc = val;
    for (int xb = 0; xb < W; xb += 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) { acc += val; ex[j][threadIdx.x] = acc; }
        __syncthreads();
        f4 v = *(f4 *)&ex[w][lane * 4];
        *(f4 *)(t + xs * (xb + w)) = v;
        __syncthreads();
    }
}
// one wave per row
__global__ void write_A1w(float *p, int W, int Dp, int TY, float val)
{
    int y = blockIdx.x;
    int lane = threadIdx.x;
    float *t = p + ((size_t)(y / TY) * W) * TY * Dp + (size_t)(y % TY) * Dp + lane * 4;
    size_t xs = (size_t)TY * Dp;
    f4 acc = {val, val, val, val};
    for (int x = 0; x < W; x++) { acc += val; *(f4 *)(t + xs * x) = acc; }
}
template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    const int R = 20;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / R;
}
int main()
{
    const int W = 1920, H = 1080, Dp = 256;
    size_t n = (size_t)W * H * Dp;
    float *a; CK(hipMalloc(&a, n * 4 + (64 << 20)));
    CK(hipMemset(a, 0, n * 4));
    double B = (double)n * 4;
    for (int TY : {1, 8, 32}) {
        float t1 = timeit([&] { write_A1<<<H, Dp>>>(a, W, Dp, TY, 1.37f); });
        float t4 = timeit([&] { write_A4<<<H, 256>>>(a, W, Dp, TY, 1.37f); });
        float t4s = timeit([&] { write_A4s<<<H, 256>>>(a, W, Dp, TY, 1.37f); });
        float t1w = timeit([&] { write_A1w<<<H, 64>>>(a, W, Dp, TY, 1.37f); });
        printf("TY %3d  256B/wave %.3f ms %5.0f GB/s | 1KiB/wave %.3f ms %5.0f | 1KiB + LDS/barrier %.3f ms %5.0f | wave/row %.3f ms %5.0f\n",
               TY, t1, B / t1 / 1e6, t4, B / t4 / 1e6, t4s, B / t4s / 1e6, t1w, B / t1w / 1e6);
    }
    float tm = timeit([&] { CK(hipMemsetAsync(a, 0, n * 4)); });
    printf("memset %.3f ms %5.0f GB/s\n", tm, B / tm / 1e6);
    return 0;
}
