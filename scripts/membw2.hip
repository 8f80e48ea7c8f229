// membw2.hip -- sweep of write / read shapes for the slab T (2.12 GB).  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// each block writes one contiguous chunk with float4
__global__ void write_chunk_f4(f4 *p, size_t n4)
{
    size_t per = n4 / gridDim.x;
    f4 *q = p + per * blockIdx.x;
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) q[i] = v;
}
template <bool NT>
__global__ void write_rows_f1(float *p, int W, int Dp)
{
    float *t = p + (size_t)blockIdx.x * W * Dp + threadIdx.x;
    float acc = threadIdx.x;
    for (int x = 0; x < W; x++) {
        acc += 1.0f;
        if (NT) __builtin_nontemporal_store(acc, &t[(size_t)x * Dp]); else t[(size_t)x * Dp] = acc;
    }
}
// rows, 4 levels per lane: block of Dp/4 threads, float4 store per step (1 KB per wave)
template <bool NT>
__global__ void write_rows_d4(f4 *p, int W, int Dp4)
{
    f4 *t = p + (size_t)blockIdx.x * W * Dp4 + threadIdx.x;
    f4 acc = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int x = 0; x < W; x++) {
        acc += 1.0f;
        if (NT) __builtin_nontemporal_store(acc, &t[(size_t)x * Dp4]); else t[(size_t)x * Dp4] = acc;
    }
}
// rows, x blocked by 4: layout [y][x/4][d][4]; block = Dp threads, one float4 per 4 steps
template <bool NT>
__global__ void write_rows_x4(f4 *p, int W, int Dp)
{
    f4 *t = p + (size_t)blockIdx.x * (W / 4) * Dp + threadIdx.x;
    f4 acc = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int x = 0; x < W / 4; x++) {
        acc += 1.0f;
        if (NT) __builtin_nontemporal_store(acc, &t[(size_t)x * Dp]); else t[(size_t)x * Dp] = acc;
    }
}
template <int U, bool NT>
__global__ void read_cols_f1(const float *p, int W, int H, int Dp, float *out)
{
    const float *t = p + (size_t)blockIdx.x * Dp + threadIdx.x;
    size_t stride = (size_t)W * Dp;
    float a = 0;
    for (int y = 0; y + U <= H; y += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&t[stride * (y + u)]) : t[stride * (y + u)];
#pragma unroll
        for (int u = 0; u < U; u++) a += v[u];
    }
    if (a == 123.456f) out[0] = a;
}
// columns with float4 per lane: block = 64 threads covering 256 floats of one pixel (or x4 layout)
template <int U, bool NT>
__global__ void read_cols_f4(const f4 *p, int W, int H, int Dp4, int tpb_cols, float *out)
{
    const f4 *t = p + (size_t)blockIdx.x * tpb_cols + threadIdx.x;
    size_t stride = (size_t)W * Dp4;
    f4 a = {0, 0, 0, 0};
    for (int y = 0; y + U <= H; y += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&t[stride * (y + u)]) : t[stride * (y + u)];
#pragma unroll
        for (int u = 0; u < U; u++) a += v[u];
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

template <class F> void timeit(const char *name, double bytes, F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
}

int main()
{
    const int W = 1920, H = 1080, Dp = 256;
    size_t n = (size_t)W * H * Dp;
    float *a, *o;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, n * 4));
    double B = (double)n * 4;
    timeit("memset", B, [&] { CK(hipMemsetAsync(a, 0, n * 4)); });
    for (int g : {256, 1024, 2048, 4096, 16384})
        { char nm[64]; snprintf(nm, 64, "write chunk f4, %d blocks", g);
          timeit(nm, B, [&] { write_chunk_f4<<<g, 256>>>((f4 *)a, n / 4); }); }
    timeit("write rows f1 (pass A now)", B, [&] { write_rows_f1<false><<<H, Dp>>>(a, W, Dp); });
    timeit("write rows f1 nontemporal", B, [&] { write_rows_f1<true><<<H, Dp>>>(a, W, Dp); });
    timeit("write rows d4 (64 thr, f4/step)", B, [&] { write_rows_d4<false><<<H, Dp / 4>>>((f4 *)a, W, Dp / 4); });
    timeit("write rows d4 nontemporal", B, [&] { write_rows_d4<true><<<H, Dp / 4>>>((f4 *)a, W, Dp / 4); });
    timeit("write rows x4 (256 thr, f4/4 steps)", B, [&] { write_rows_x4<false><<<H, Dp>>>((f4 *)a, W, Dp); });
    timeit("write rows x4 nontemporal", B, [&] { write_rows_x4<true><<<H, Dp>>>((f4 *)a, W, Dp); });
    timeit("read cols f1 U8 (pass B now)", B, [&] { read_cols_f1<8, false><<<W, Dp>>>(a, W, H, Dp, o); });
    timeit("read cols f1 U8 nontemporal", B, [&] { read_cols_f1<8, true><<<W, Dp>>>(a, W, H, Dp, o); });
    timeit("read cols f1 U18", B, [&] { read_cols_f1<18, false><<<W, Dp>>>(a, W, H, Dp, o); });
    timeit("read cols f4 U8, 64 thr (4 d / lane)", B, [&] { read_cols_f4<8, false><<<W, 64>>>((const f4 *)a, W, H, Dp / 4, 64, o); });
    timeit("read cols f4 U8, 64 thr nontemporal", B, [&] { read_cols_f4<8, true><<<W, 64>>>((const f4 *)a, W, H, Dp / 4, 64, o); });
    timeit("read cols f4 U8, 256 thr (x4 layout)", B, [&] { read_cols_f4<8, false><<<W / 4, 256>>>((const f4 *)a, W / 4, H, Dp, 256, o); });
    timeit("read cols f4 U4, 256 thr (x4 layout)", B, [&] { read_cols_f4<4, false><<<W / 4, 256>>>((const f4 *)a, W / 4, H, Dp, 256, o); });
    return 0;
}
