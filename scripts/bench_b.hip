// bench_b.hip -- ablation timing of pass B (includes the product kernel header).  GPU box only.
#include "../stereomatch_amd/csrc/smhip_fused_b.h"
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F> void timeit(const char *name, double bytes, F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.3f ms  %7.1f GB/s\n", name, ms / R, bytes / (ms / R * 1e-3) / 1e9);
}

int main()
{
    const int W = 1920, H = 1080, Dp = 256, D = 256;
    size_t trow = (size_t)W * Dp; const int tyshift = 5;
    float *T; long long *keys;
    CK(hipMalloc(&T, trow * (H + 32) * 4)); CK(hipMalloc(&keys, (size_t)W * H * 8));
    CK(hipMemset(T, 0, trow * (H + 32) * 4));
    double B = (double)W * H * Dp * 4;
    int xpx = (W + 7) / 8;
    float scale = (float)(1.0 / 9);
#define RUNK(ABL, WPB, KB, NAME) timeit(NAME, B, [&] { dim3 g(8 * ((xpx + WPB - 1) / WPB), 1); \
    smk::fused_boxv_wta_vpl_kernel<9, 4, ABL><<<g, 64 * WPB, (size_t)WPB * KB * 8>>>(T, tyshift, W, H, Dp, D, 9, scale, 0, -1, 1000.f, 0, xpx, keys, (size_t)W * H, nullptr, 0, KB); })
#define RUN(ABL, WPB, NAME) RUNK(ABL, WPB, 0, NAME)
    for (int rep = 0; rep < 3; rep++) {
        RUNK(0, 4, 0, "full, unbuffered");
        RUNK(0, 4, 63 * 18, "full, keys buffered 1134 rows");
        RUNK(0, 4, 63 * 8, "full, keys buffered 504 rows");
        RUNK(0, 4, 63 * 2, "full, keys buffered 126 rows");
        RUNK(1, 4, 0, "no wave reduce, unbuffered");
        RUNK(1, 4, 63 * 18, "no wave reduce, 1134 rows");
        RUNK(2, 4, 0, "no key stores");
        RUNK(3, 4, 0, "no reduce, no key stores");
    }
    return 0;
}
