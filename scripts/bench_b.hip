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
        RUNK(4, 4, 63 * 18, "non-temporal key stores, 1134 rows");
    }
    // 128-level slabs on a narrow frame (1920 waves of 512-byte rows): what bounds the walk there?
    {
        const int Dp2 = 128;
        double B2 = (double)W * H * Dp2 * 4;
#define RUN2(ABL, AH, KB, NAME) timeit(NAME, B2, [&] { dim3 g(8 * ((xpx + 3) / 4), 1); \
    smk::fused_boxv_wta_vpl_kernel<9, 2, ABL, false, false, AH><<<g, 256, (size_t)4 * KB * 8>>>(T, tyshift, W, H, Dp2, Dp2, 9, scale, 0, -1, 1000.f, 0, xpx, keys, (size_t)W * H, nullptr, 0, KB); })
        for (int rep = 0; rep < 3; rep++) {
            RUN2(0, 4, 63 * 18, "128 levels: full, ahead 4");
            RUN2(1, 4, 63 * 18, "128 levels: no wave reduce, ahead 4");
            RUN2(2, 4, 63 * 18, "128 levels: no key stores, ahead 4");
            RUN2(3, 4, 63 * 18, "128 levels: no reduce, no key stores");
            RUN2(0, 2, 63 * 18, "128 levels: full, ahead 2");
            RUN2(1, 2, 63 * 18, "128 levels: no wave reduce, ahead 2");
        }
    }
    // paired columns (128-level slabs from 3584 columns on; config 5's share per GPU at N = 8 is 7680 x 4320 x 128)
    {
        CK(hipFree(T)); CK(hipFree(keys));
        const int W3 = 7680, H3 = 4320, Dp3 = 128;
        CK(hipMalloc(&T, (size_t)W3 * Dp3 * (H3 + 32) * 4)); CK(hipMalloc(&keys, (size_t)W3 * H3 * 8));
        CK(hipMemset(T, 0, (size_t)W3 * Dp3 * (H3 + 32) * 4));
        double B3 = (double)W3 * H3 * Dp3 * 4;
        const int xp = ((W3 + 1) / 2 + 7) / 8, prow = 27 * 42;
#define RUN3(ABL, NAME) timeit(NAME, B3, [&] { dim3 g(8 * ((xp + 3) / 4), 1); \
    smk::fused_boxv_wta_vpl_kernel<9, 4, ABL, false, true><<<g, 256, (size_t)4 * 2 * prow * 8>>>(T, tyshift, W3, H3, Dp3, Dp3, 9, scale, 0, -1, 1000.f, 0, xp, keys, (size_t)W3 * H3, nullptr, 0, prow); })
        for (int rep = 0; rep < 2; rep++) {
            RUN3(0, "8K x 128 paired: full");
            RUN3(1, "8K x 128 paired: no wave reduce");
            RUN3(2, "8K x 128 paired: no key stores");
            RUN3(3, "8K x 128 paired: no reduce, no key stores");
            RUN3(4, "8K x 128 paired: non-temporal key stores");
        }
    }
    // one column per wave, 256 levels, large frames: plain against non-temporal key stores
    {
        CK(hipFree(T)); CK(hipFree(keys));
        for (int big = 0; big < 2; big++) {
            const int W4 = big ? 7680 : 3840, H4 = big ? 4320 : 2160, Dp4 = 256;
            CK(hipMalloc(&T, (size_t)W4 * Dp4 * (H4 + 32) * 4)); CK(hipMalloc(&keys, (size_t)W4 * H4 * 8));
            CK(hipMemset(T, 0, (size_t)W4 * Dp4 * (H4 + 32) * 4));
            double B4 = (double)W4 * H4 * Dp4 * 4;
            const int xp4 = (W4 + 7) / 8, kb = 63 * 36;
#define RUN4(ABL, NAME) timeit(NAME, B4, [&] { dim3 g(8 * ((xp4 + 3) / 4), 1); \
    smk::fused_boxv_wta_vpl_kernel<9, 4, ABL><<<g, 256, 81920>>>(T, tyshift, W4, H4, Dp4, Dp4, 9, scale, 0, -1, 1000.f, 0, xp4, keys, (size_t)W4 * H4, nullptr, 0, kb); })
            for (int rep = 0; rep < 2; rep++) {
                RUN4(0, big ? "8K x 256: plain key stores" : "4K x 256: plain key stores");
                RUN4(4, big ? "8K x 256: non-temporal key stores" : "4K x 256: non-temporal key stores");
                RUN4(2, big ? "8K x 256: no key stores" : "4K x 256: no key stores");
            }
            CK(hipFree(T)); CK(hipFree(keys));
        }
        T = nullptr; keys = nullptr;
    }
    return 0;
}
