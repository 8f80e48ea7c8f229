// membw4.hip -- banded layouts of T: element (y, x, d) at ((y/TY)*W + x)*TY*Dp + (y%TY)*Dp + d.
// Times the pass-A write order (x serial, all y concurrent) and the pass-B read order
// (y serial, all x concurrent) for several TY.  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void write_A(float *p, int W, int Dp, int TY, float val)
{
    int y = blockIdx.x;
    float *t = p + ((size_t)(y / TY) * W) * TY * Dp + (size_t)(y % TY) * Dp + threadIdx.x;
    size_t xs = (size_t)TY * Dp;
    float acc = val;
    for (int x = 0; x < W; x++) { acc += val; t[xs * x] = acc; }
}
__global__ void read_B(const float *p, int W, int H, int Dp, int TY, float *out)
{
    int x = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    const f4 *t = (const f4 *)(p + (size_t)x * TY * Dp) + lane;
    size_t band = (size_t)W * TY * Dp / 4, row = Dp / 4;
    f4 a = {0, 0, 0, 0};
    for (int yb = 0; yb < H; yb += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { int y = yb + u; v[u] = __builtin_nontemporal_load(&t[(size_t)(y / TY) * band + (size_t)(y % TY) * row]); }
#pragma unroll
        for (int u = 0; u < 8; u++) a += v[u];
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}
template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / R;
}
int main()
{
    const int W = 1920, H = 1080, Dp = 256;
    size_t n = (size_t)W * H * Dp;
    float *a, *o; CK(hipMalloc(&a, n * 4 + (1 << 20))); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, n * 4));
    double B = (double)n * 4;
    for (int TY : {1, 2, 4, 8, 12, 24, 40, 120, 360, 1080}) {
        float tw = timeit([&] { write_A<<<H, Dp>>>(a, W, Dp, TY, 1.37f); });
        float tr = timeit([&] { read_B<<<W / 4, 256>>>(a, W, H, Dp, TY, o); });
        float tboth = timeit([&] { write_A<<<H, Dp>>>(a, W, Dp, TY, 1.37f); read_B<<<W / 4, 256>>>(a, W, H, Dp, TY, o); });
        printf("TY %4d  write %.3f ms %6.0f GB/s   read %.3f ms %6.0f GB/s   back-to-back %.3f ms\n", TY, tw,
               B / tw / 1e6, tr, B / tr / 1e6, tboth);
    }
    return 0;
}
