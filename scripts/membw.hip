// membw.hip -- achievable HBM read / write rates for the access shapes the fused passes use.
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/membw scripts/membw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void write_f4(float4 *p, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += s) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void write_f1(float *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = 1.f;
}
// pass-A shape: block = one row (W steps of 256 floats), each thread stores one float per step
__global__ void write_rows(float *p, int W, int Dp)
{
    float *t = p + (size_t)blockIdx.x * W * Dp + threadIdx.x;
    float acc = threadIdx.x;
    for (int x = 0; x < W; x++) { acc += 1.0f; t[(size_t)x * Dp] = acc; }
}
__global__ void read_f4(const float4 *p, size_t n4, float *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i < n4; i += s) { float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.456f) out[0] = a;
}
__global__ void read_f1(const float *p, size_t n, float *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i < n; i += s) a += p[i];
    if (a == 123.456f) out[0] = a;
}
// pass-B shape: block = one pixel column (H steps of 256 floats at stride W*Dp)
__global__ void read_cols(const float *p, int W, int H, int Dp, float *out)
{
    const float *t = p + (size_t)blockIdx.x * Dp + threadIdx.x;
    size_t stride = (size_t)W * Dp;
    float a = 0;
#pragma unroll 8
    for (int y = 0; y < H; y++) a += t[stride * y];
    if (a == 123.456f) out[0] = a;
}
__global__ void copy_f4(const float4 *a, float4 *b, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += s) b[i] = a[i];
}

template <class F> void timeit(const char *name, double bytes, F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
}

int main()
{
    const int W = 1920, H = 1080, Dp = 256;
    size_t n = (size_t)W * H * Dp;
    float *a, *b, *o;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    double B = (double)n * 4;
    timeit("memset", B, [&] { CK(hipMemsetAsync(a, 0, n * 4)); });
    timeit("write float4 grid-stride", B, [&] { write_f4<<<256 * 8, 256>>>((float4 *)a, n / 4); });
    timeit("write float  grid-stride", B, [&] { write_f1<<<256 * 8, 256>>>(a, n); });
    timeit("write rows (pass-A shape)", B, [&] { write_rows<<<H, Dp>>>(a, W, Dp); });
    timeit("read float4 grid-stride", B, [&] { read_f4<<<256 * 8, 256>>>((float4 *)a, n / 4, o); });
    timeit("read float  grid-stride", B, [&] { read_f1<<<256 * 8, 256>>>(a, n, o); });
    timeit("read cols (pass-B shape)", B, [&] { read_cols<<<W, Dp>>>(a, W, H, Dp, o); });
    timeit("copy float4 (r+w bytes)", 2 * B, [&] { copy_f4<<<256 * 8, 256>>>((float4 *)a, (float4 *)b, n / 4); });
    return 0;
}
