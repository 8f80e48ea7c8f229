// bench_c.hip -- the recomputing form of the fused path (smhip_fused_c.h: boundary states + block kernel) against the
// two-pass form (pass A slab + pass B): keys compared bit for bit, both timed.  GPU box only.
//   bench_c [W H D]
#include "smhip_fused_c.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F> float timeit(F f, int R = 20)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < R; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / R;
}
static uint64_t sm64(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int DS, bool CUT>
static int run(int W, int H, int D, int d_first, int cutoff_in)
{
    constexpr int WIN = 9;
    const int Dp = D <= 128 ? 128 : (D + 255) / 256 * 256;
    const int CH = Dp < 256 ? Dp : 256, tyshift = 5;
    size_t bands = ((size_t)H + 31) >> 5;
    size_t n = (bands << 5) * (size_t)W * Dp;
    std::vector<uint8_t> hr((size_t)W * H), hm((size_t)W * H);
    uint64_t s = 0x5EED0003;
    for (auto &v : hr) v = (uint8_t)(sm64(s) & 0xFF);
    for (size_t i = 0; i < hm.size(); i++) hm[i] = (uint8_t)((hr[(i + 7) % hr.size()] + (sm64(s) & 7)) & 0xFF);
    uint8_t *dr, *dm; float *T, *S; long long *k0, *k1;
    CK(hipMalloc(&dr, hr.size())); CK(hipMalloc(&dm, hm.size()));
    CK(hipMemcpy(dr, hr.data(), hr.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hm.data(), hm.size(), hipMemcpyHostToDevice));
    const int nst = (W + WIN - 1) / WIN;
    CK(hipMalloc(&T, n * 4)); CK(hipMalloc(&S, (size_t)H * nst * Dp * 4));
    const size_t npix = (size_t)W * H;
    CK(hipMalloc(&k0, npix * 8 * (Dp / CH))); CK(hipMalloc(&k1, npix * 8 * (Dp / 128)));
    CK(hipMemset(k0, 0x11, npix * 8 * (Dp / CH))); CK(hipMemset(k1, 0x22, npix * 8 * (Dp / 128)));
    smk::FusedA fa;
    fa.rn = fa.rx = dr; fa.mn = fa.mx = dm; fa.pitch = W; fa.W = W; fa.H = H; fa.D = D;
    fa.dstep = DS; fa.disp0 = DS * d_first; fa.cutoff = cutoff_in; fa.outside = 255 < cutoff_in ? 255 : cutoff_in; fa.bt_inline = 1;
    const float scale = (float)(1.0 / WIN), outside = (float)fa.outside;
    printf("W %d H %d D %d (Dp %d) d_first %d dstep %d cutoff %d\n", W, H, D, Dp, d_first, DS, cutoff_in);
    const int nw = Dp % 256 == 0 ? 2 : 1, cw = 128 * nw;
    int cbs = 6 * 64 * nw - WIN - cw; if (cbs > 480) cbs = 480;
    const int cb = cbs / WIN, nref = cb * WIN + WIN, nmatch = nref + cw;
    const size_t ldsa = (size_t)nmatch * 8 + (size_t)nref * 4;
    auto passA = [&] {
        dim3 g((unsigned)H * (Dp / cw));
        if (nw == 2) smk::fused_rawbox_h_f_kernel<WIN, false, CUT, DS, 2, 2, 0, 16><<<g, 128, ldsa>>>(fa, T, tyshift, Dp, scale, 0, cb, nref, nmatch, CH / 64);
        else smk::fused_rawbox_h_f_kernel<WIN, false, CUT, DS, 2, 1, 0, 16><<<g, 64, ldsa>>>(fa, T, tyshift, Dp, scale, 0, cb, nref, nmatch, CH / 64);
    };
    const int xpx = (W + 7) / 8;
    auto passB = [&] {
        dim3 g(8 * ((xpx + 3) / 4), Dp / CH);
        if (CH == 256) smk::fused_boxv_wta_vpl_kernel<WIN, 4><<<g, 256>>>(T, tyshift, W, H, Dp, D, WIN, scale, fa.disp0, DS, outside, d_first, xpx, k0, npix);
        else smk::fused_boxv_wta_vpl_kernel<WIN, 2><<<g, 256>>>(T, tyshift, W, H, Dp, D, WIN, scale, fa.disp0, DS, outside, d_first, xpx, k0, npix);
    };
    auto phase1 = [&] {
        dim3 g((unsigned)H * (Dp / cw));
        if (nw == 2) smk::fused_rawbox_h_f_kernel<WIN, false, CUT, DS, 2, 2, 0, 0, false, false, true><<<g, 128, ldsa>>>(fa, S, nst, Dp, scale, 1, cb, nref, nmatch, 2);
        else smk::fused_rawbox_h_f_kernel<WIN, false, CUT, DS, 2, 1, 0, 0, false, false, true><<<g, 64, ldsa>>>(fa, S, nst, Dp, scale, 1, cb, nref, nmatch, 2);
    };
    auto phase2 = [&] {
        dim3 g((unsigned)nst * (Dp / 128));
        smk::fused_block_vwta_kernel<WIN, false, CUT, DS><<<g, 64 * WIN, smk::fused_c_lds(WIN)>>>(fa, S, nst, Dp, scale, outside, d_first, k1, npix, 0);
    };
    passA(); passB(); phase1(); phase2();
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    std::vector<long long> a(npix * (Dp / CH)), b(npix * (Dp / 128));
    CK(hipMemcpy(a.data(), k0, a.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), k1, b.size() * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t i = 0; i < npix; i++) {
        long long va = a[i], vb = b[i];
        for (int c = 1; c < Dp / CH; c++) va = std::min(va, a[i + c * npix]);
        for (int c = 1; c < Dp / 128; c++) vb = std::min(vb, b[i + c * npix]);
        if (va != vb) { if (bad < 8) printf("  key differs at x %zu y %zu: %016llx vs %016llx\n", i / H, i % H, va, vb); bad++; }
    }
    printf("  %zu keys compared, %ld differ\n", npix, bad);
    float ta = timeit(passA), tb = timeit(passB), t1 = timeit(phase1), t2 = timeit(phase2);
    float tp = timeit([&] { passA(); passB(); }), tc = timeit([&] { phase1(); phase2(); });
    {
        dim3 g((unsigned)nst * (Dp / 128));
        float a1 = timeit([&] { smk::fused_block_vwta_kernel<WIN, false, CUT, DS, 1><<<g, 64 * WIN, smk::fused_c_lds(WIN)>>>(fa, S, nst, Dp, scale, outside, d_first, k1, npix, 0); });
        float a2 = timeit([&] { smk::fused_block_vwta_kernel<WIN, false, CUT, DS, 2><<<g, 64 * WIN, smk::fused_c_lds(WIN)>>>(fa, S, nst, Dp, scale, outside, d_first, k1, npix, 0); });
        float a3 = timeit([&] { smk::fused_block_vwta_kernel<WIN, false, CUT, DS, 3><<<g, 64 * WIN, smk::fused_c_lds(WIN)>>>(fa, S, nst, Dp, scale, outside, d_first, k1, npix, 0); });
        printf("  blocks ablations: no lane reduction %.3f, no row staging %.3f, neither %.3f ms\n", a1, a2, a3);
    }
    double N8 = 8.0 * W * H * D;
    printf("  two-pass: A %.3f B %.3f pair %.3f ms (%.2f of 8N/8TB/s) | recompute: states %.3f blocks %.3f pair %.3f ms (%.2f)\n",
           ta, tb, tp, N8 / tp / 8e9, t1, t2, tc, N8 / tc / 8e9);
    CK(hipFree(dr)); CK(hipFree(dm)); CK(hipFree(T)); CK(hipFree(S)); CK(hipFree(k0)); CK(hipFree(k1));
    return bad != 0;
}

int main(int argc, char **argv)
{
    int bad = 0;
    if (argc >= 4) {
        bad += run<-1, false>(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), 0, 1000);
        return bad;
    }
    bad += run<-1, false>(200, 50, 128, 0, 1000);
    bad += run<-1, false>(203, 64, 256, 0, 1000);
    bad += run<1, false>(330, 70, 100, 3, 1000);
    bad += run<-1, true>(330, 70, 200, 0, 20);
    bad += run<1, true>(64, 9, 128, 0, 5);
    bad += run<-1, false>(1920, 1080, 256, 0, 1000);
    bad += run<-1, true>(1920, 1080, 256, 0, 20);
    bad += run<-1, false>(1920, 1080, 128, 0, 1000);
    bad += run<-1, false>(3840, 2160, 256, 0, 1000);
    printf(bad ? "FAILED\n" : "all equal\n");
    return bad;
}
