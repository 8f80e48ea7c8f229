// membw8.hip -- write-only / read-only ceilings in the access order of pass A / pass B for a slab
// of Dp levels (argv: W H Dp): is the 128-level slab slower because its shape is?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
// pass A order: workgroup = image row x Dp levels (one float per lane), x serial
__global__ void write_A(float *p, int W, int Dp, int TY, float val)
{
    int y = blockIdx.x;
    float *t = p + ((size_t)(y / TY) * W) * TY * Dp + (size_t)(y % TY) * Dp + threadIdx.x;
    size_t xs = (size_t)TY * Dp;
    float acc = val;
    for (int x = 0; x < W; x++) { acc += val; t[xs * x] = acc; }
}
// pass B order: wave = pixel column, Dp / 64 floats per lane, y serial, 4 columns per block
template <class V>
__global__ void read_B(const float *p, int W, int H, int Dp, int TY, float *out)
{
    int x = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (x >= W) return;
    V a = {};
    for (int y = 0; y < H; y++) {
        const float *t = p + ((size_t)(y / TY) * W + x) * TY * Dp + (size_t)(y % TY) * Dp;
        a += __builtin_nontemporal_load((const V *)t + lane);
    }
    if (a[0] == 12345.678f) out[0] = 1;
}
template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) f();
    CK(hipEventRecord(e0));
    const int R = 6;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / R;
}
int main(int argc, char **argv)
{
    int W = argc > 1 ? atoi(argv[1]) : 7680, H = argc > 2 ? atoi(argv[2]) : 4320, Dp = argc > 3 ? atoi(argv[3]) : 128;
    const int TY = 32;
    size_t n = (size_t)W * ((H + TY - 1) / TY * TY) * Dp;
    float *a, *o; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, n * 4));
    double B = (double)W * H * Dp * 4;
    float tw = timeit([&] { write_A<<<H, Dp>>>(a, W, Dp, TY, 1.37f); });
    // the same with pass A's LDS footprint (4 bytes per pixel of the row): occupancy as in the product
    float twl = timeit([&] { write_A<<<H, Dp, (size_t)W * 4>>>(a, W, Dp, TY, 1.37f); });
    printf("write-only with %d KB of LDS per workgroup: %.3f ms\n", W * 4 / 1024, twl);
    float tr = Dp == 256 ? timeit([&] { read_B<f4><<<(W + 3) / 4, 256>>>(a, W, H, Dp, TY, o); })
                         : timeit([&] { read_B<f2><<<(W + 3) / 4, 256>>>(a, W, H, Dp, TY, o); });
    printf("%dx%d x %d levels: write-only (pass A order) %.3f ms %5.0f GB/s (%.3f of 8 TB/s) | read-only (pass B order) %.3f ms %5.0f GB/s (%.3f)\n",
           W, H, Dp, tw, B / tw / 1e6, B / tw / 8e9, tr, B / tr / 1e6, B / tr / 8e9);
    if (Dp == 128) {
        // two consecutive rows of a 128-level column are adjacent in the slab: one 1-KiB load per row PAIR
        // is the address stream of a 256-level slab with half the rows and 16-row bands
        float tp = timeit([&] { read_B<f4><<<(W + 3) / 4, 256>>>(a, W, H / 2, 256, TY / 2, o); });
        printf("read-only, row pairs (1 KiB per wave load): %.3f ms %5.0f GB/s (%.3f)\n", tp, B / tp / 1e6, B / tp / 8e9);
    }
    return 0;
}
