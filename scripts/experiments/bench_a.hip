// bench_a.hip -- pass A variants on config 3's shape (and the 128-level slab shape): timing and a
// bit-for-bit comparison of the slab T with the one-level-per-lane kernel.  GPU box only.
#include "smhip_fused_ps.h"
#include "../../stereomatch_amd/csrc/smhip_fused_f.h"
#include "smhip_fused_bf.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pass A alone, back to back
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

// the product's step: pass A then pass B, events around each (as smhip_profile_begin does)
struct PairTime { float a_avg, a_min, b_avg, b_min, wall; };
template <class FA, class FB> PairTime time_pair(FA fa, FB fb, int R = 30)
{
    std::vector<hipEvent_t> ev(3 * R);
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    for (int i = 0; i < 5; i++) { fa(); fb(); }
    CK(hipDeviceSynchronize());
    for (int i = 0; i < R; i++) {
        CK(hipEventRecord(ev[3 * i])); fa();
        CK(hipEventRecord(ev[3 * i + 1])); fb();
        CK(hipEventRecord(ev[3 * i + 2]));
    }
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    PairTime p = {0, 1e9f, 0, 1e9f, 0};
    for (int i = 0; i < R; i++) {
        float ta, tb;
        CK(hipEventElapsedTime(&ta, ev[3 * i], ev[3 * i + 1]));
        CK(hipEventElapsedTime(&tb, ev[3 * i + 1], ev[3 * i + 2]));
        p.a_avg += ta / R; p.b_avg += tb / R;
        if (ta < p.a_min) p.a_min = ta;
        if (tb < p.b_min) p.b_min = tb;
    }
    CK(hipEventElapsedTime(&p.wall, ev[0], ev[3 * R - 1]));
    p.wall /= R;
    for (auto &evt : ev) CK(hipEventDestroy(evt));
    return p;
}

static uint64_t sm64(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// compare: every element the baseline wrote (not the sentinel) must be equal in the candidate
static int g_W, g_Dp;
static long compare(const float *dbase, const float *dcand, size_t n)
{
    std::vector<uint32_t> a(n), b(n);
    CK(hipMemcpy(a.data(), dbase, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dcand, n * 4, hipMemcpyDeviceToHost));
    long bad = 0, written = 0;
    for (size_t i = 0; i < n; i++)
        if (a[i] != 0xFFFFFFFFu) { written++; if (a[i] != b[i]) { if (bad < 5) printf("  diff at %zu: %08x vs %08x\n", i, a[i], b[i]); bad++; } }
    printf("  compared %ld written elements, %ld differ\n", written, bad);
    if (bad) {   // where: slab layout [band][x][ty 32][Dp]
        std::vector<long> byx(g_W, 0), byrow(4096, 0), byk(g_Dp, 0);
        for (size_t i = 0; i < n; i++)
            if (a[i] != 0xFFFFFFFFu && a[i] != b[i]) {
                size_t per_band = (size_t)g_W * 32 * g_Dp;
                size_t band = i / per_band, r = i % per_band;
                int x = (int)(r / (32 * g_Dp)), ty = (int)(r % (32 * g_Dp)) / g_Dp, k = (int)(r % g_Dp);
                byx[x]++; byrow[band * 32 + ty]++; byk[k]++;
            }
        printf("  by x:"); for (int x = 0; x < g_W; x++) if (byx[x]) printf(" %d:%ld", x, byx[x]); printf("\n");
        printf("  by row:"); for (int y = 0; y < 4096; y++) if (byrow[y]) printf(" %d:%ld", y, byrow[y]); printf("\n");
        printf("  by level:"); for (int k = 0; k < g_Dp; k++) if (byk[k]) printf(" %d:%ld", k, byk[k]); printf("\n");
    }
    return bad;
}

// store-only kernels in pass A's order: the ceiling of each store shape
template <int LPL, int K = 0>     // K dependent VALU instructions per level and step ("think time")
__global__ void store_only_kernel(float *T, int W, int H, int Dp, int tyshift, float val)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int y = blockIdx.x * (blockDim.x >> 6) + wib;
    if (y >= H) return;
    float *t = T + smk::slab_row_offset(y, W, Dp, tyshift) + blockIdx.y * 64 * LPL + lane * LPL;
    const size_t xs = (size_t)Dp << tyshift;
    typename smk::vecf<LPL>::type acc;
    if constexpr (LPL == 1) acc = val; else for (int i = 0; i < LPL; i++) acc[i] = val;
    for (int x = 0; x < W; x++) {
        acc += val;
#pragma unroll
        for (int j = 0; j < K; j++) acc = acc * 1.0001f;
        *(typename smk::vecf<LPL>::type *)(t + xs * x) = acc;
    }
}

template <int DS>
static int run(int W, int H, int D, int d_first, int cutoff_in, bool check, int tyshift = 5)
{
    const int Dp = D <= 64 ? 64 : D <= 128 ? 128 : (D + 255) / 256 * 256;
    const int CH = Dp < 256 ? Dp : 256;
    g_W = W; g_Dp = Dp;
    size_t bands = ((size_t)H + (1 << tyshift) - 1) >> tyshift;
    size_t n = (bands << tyshift) * (size_t)W * Dp;
    std::vector<uint8_t> hr((size_t)W * H), hm((size_t)W * H);
    uint64_t s = 0x5EED0002;
    for (auto &v : hr) v = (uint8_t)sm64(s);
    for (auto &v : hm) v = (uint8_t)sm64(s);
    uint8_t *dr, *dm; float *T0, *T1; long long *keys;
    CK(hipMalloc(&dr, hr.size())); CK(hipMalloc(&dm, hm.size()));
    CK(hipMemcpy(dr, hr.data(), hr.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hm.data(), hm.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&T0, n * 4)); CK(hipMalloc(&T1, n * 4));
    CK(hipMalloc(&keys, (size_t)W * H * 8 * (Dp / CH)));
    smk::FusedA fa;
    fa.rn = fa.rx = dr; fa.mn = fa.mx = dm; fa.pitch = W; fa.W = W; fa.H = H; fa.D = D;
    fa.dstep = DS; fa.disp0 = DS * d_first; fa.cutoff = cutoff_in; fa.outside = 255 < cutoff_in ? 255 : cutoff_in; fa.bt_inline = 1;
    const float scale = (float)(1.0 / 9);
    double B = (double)W * H * D * 4;
    printf("W %d H %d D %d (Dp %d) d_first %d dstep %d cutoff %d tyshift %d\n", W, H, D, Dp, d_first, DS, cutoff_in, tyshift);
    const int xpx = (W + 7) / 8;
    auto passB = [&](const float *T) {
        dim3 g(8 * ((xpx + 3) / 4), Dp / CH);
        if (CH == 256) smk::fused_boxv_wta_vpl_kernel<9, 4><<<g, 256>>>(T, tyshift, W, H, Dp, D, 9, scale, fa.disp0, DS, (float)fa.outside, d_first, xpx, keys, (size_t)W * H);
        else if (CH == 128) smk::fused_boxv_wta_vpl_kernel<9, 2><<<g, 256>>>(T, tyshift, W, H, Dp, D, 9, scale, fa.disp0, DS, (float)fa.outside, d_first, xpx, keys, (size_t)W * H);
        else smk::fused_boxv_wta_vpl_kernel<9, 1><<<g, 256>>>(T, tyshift, W, H, Dp, D, 9, scale, fa.disp0, DS, (float)fa.outside, d_first, xpx, keys, (size_t)W * H);
    };
    long long *keys2; CK(hipMalloc(&keys2, (size_t)W * H * 8 * (Dp / CH)));
    const int kbr = 63 * std::min((H + 62) / 63, 18);
    auto passBf = [&](const float *T) {
        dim3 g(8 * ((xpx + 3) / 4), Dp / CH);
        size_t lds = (size_t)4 * kbr * 8;
        if (CH == 256) smk::fused_boxv_wta_f_kernel<9, 4><<<g, 256, lds>>>(T, tyshift, W, H, Dp, D, scale, fa.disp0, DS, (float)fa.outside, d_first, xpx, keys2, (size_t)W * H, kbr);
        else if (CH == 128) smk::fused_boxv_wta_f_kernel<9, 2><<<g, 256, lds>>>(T, tyshift, W, H, Dp, D, scale, fa.disp0, DS, (float)fa.outside, d_first, xpx, keys2, (size_t)W * H, kbr);
        else smk::fused_boxv_wta_f_kernel<9, 1><<<g, 256, lds>>>(T, tyshift, W, H, Dp, D, scale, fa.disp0, DS, (float)fa.outside, d_first, xpx, keys2, (size_t)W * H, kbr);
    };
    auto report = [&](const char *name, PairTime p, float alone) {
        printf("  %-30s A %.3f (min %.3f, alone %.3f) ms %6.0f GB/s | B %.3f (min %.3f) | pair wall %.3f ms\n", name, p.a_avg,
               p.a_min, alone, B / p.a_avg / 1e6, p.b_avg, p.b_min, p.wall);
    };
    int tpa = CH;
    auto base = [&] { dim3 g(H, Dp / tpa); smk::fused_rawbox_h_lds_kernel<9, false, 1><<<g, tpa, (size_t)W * 4>>>(fa, T0, tyshift, Dp, 9, scale, 0); };
    CK(hipMemset(T0, 0xFF, n * 4));
    { float al = timeit(base); report("1 level/lane (product r01)", time_pair(base, [&] { passB(T0); }), al); }
    int bad = 0;
#define CAND(LPL, RPW, NAME) CANDA(LPL, RPW, 0, NAME)
#define CANDA(LPL, RPW, ABL, NAME) do { \
        auto f = [&] { dim3 g((H + RPW - 1) / RPW, Dp / (64 * LPL)); \
            smk::fused_rawbox_h_a4_kernel<9, false, 1, DS, LPL, ABL><<<g, 64 * RPW, (size_t)RPW * W * 4>>>(fa, T1, tyshift, Dp, scale, 0); }; \
        CK(hipMemset(T1, 0xFF, n * 4)); \
        float al = timeit(f); report(NAME, time_pair(f, [&] { passB(T1); }), al); \
        if (check && !ABL) bad += compare(T0, T1, n) != 0; \
    } while (0)
    CAND(1, 1, "1 level/lane, 1 wave/WG (a4)");
    CAND(1, 4, "1 level/lane, 4 rows/WG (a4)");
    CANDA(1, 1, 1, "1 level/lane, no stores");
    CANDA(1, 4, 1, "1 level/lane 4 rows/WG, no stores");
    if (Dp % 128 == 0) CANDA(2, 1, 1, "2 levels/lane, no stores");
    if (Dp % 256 == 0) CANDA(4, 1, 1, "4 levels/lane, no stores");
#define CAND_F(LPL, NW, ABL, NAME) CAND_FK(LPL, NW, ABL, 0, NAME)
#define CAND_FK(LPL, NW, ABL, KEEP, NAME) do { \
        constexpr int CW = 64 * LPL * NW; \
        if (Dp % CW == 0 && (Dp == CW || CW == 256)) { \
        const int cbs = std::min(480, smk::FUSED_F_NI * 64 * NW - 9 - CW); const int cb = cbs / 9, nref = cb * 9 + 9, nmatch = nref + CW; \
        const bool cut = cutoff_in < 255; \
        auto f = [&] { dim3 g(H, Dp / CW); \
            if (cut) smk::fused_rawbox_h_f_kernel<9, false, true, DS, LPL, NW, ABL><<<g, 64 * NW, (size_t)nmatch * 8 + nref * 4>>>(fa, T1, tyshift, Dp, scale, KEEP, cb, nref, nmatch, CH / 64); \
            else smk::fused_rawbox_h_f_kernel<9, false, false, DS, LPL, NW, ABL><<<g, 64 * NW, (size_t)nmatch * 8 + nref * 4>>>(fa, T1, tyshift, Dp, scale, KEEP, cb, nref, nmatch, CH / 64); }; \
        CK(hipMemset(T1, 0xFF, n * 4)); \
        float al = timeit(f); report(NAME, time_pair(f, [&] { passB(T1); }), al); \
        if (check && !(ABL) && !(KEEP)) bad += compare(T0, T1, n) != 0; } \
    } while (0)
    CAND_F(2, 2, 0, "float lanes 2 levels x 2 waves");
    if (D % (CH / 64) == 0 && Dp % 256 == 0) {    // lean pass B on the slab the float kernel has just written: same keys as pass B
        size_t nk = (size_t)W * H * (Dp / CH);
        CK(hipMemset(keys, 0x11, nk * 8)); CK(hipMemset(keys2, 0x22, nk * 8));
        passB(T1); passBf(T1); CK(hipDeviceSynchronize());
        std::vector<long long> ka(nk), kb(nk);
        CK(hipMemcpy(ka.data(), keys, nk * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(kb.data(), keys2, nk * 8, hipMemcpyDeviceToHost));
        long kbad = 0; for (size_t i = 0; i < nk; i++) if (ka[i] != kb[i]) { if (kbad < 5) printf("  key diff at %zu: %llx vs %llx\n", i, ka[i], kb[i]); kbad++; }
        printf("  lean pass B: %ld of %zu keys differ\n", kbad, nk); bad += kbad != 0;
        auto f = [&] { dim3 g(H, Dp / 256); const int cbs = std::min(480, smk::FUSED_F_NI * 128 - 9 - 256); const int cb = cbs / 9, nref = cb * 9 + 9, nmatch = nref + 256;
            smk::fused_rawbox_h_f_kernel<9, false, false, DS, 2, 2, 0><<<g, 128, (size_t)nmatch * 8 + nref * 4>>>(fa, T1, tyshift, Dp, scale, 0, cb, nref, nmatch, CH / 64); };
        if (cutoff_in >= 255) report("float lanes 2x2 + lean pass B", time_pair(f, [&] { passBf(T1); }), 0.f);
    }
    CAND_F(2, 2, 16, "float lanes 2x2, stores sc0");
    CAND_F(2, 2, 32, "float lanes 2x2, stores nt");
    CAND_F(2, 2, 48, "float lanes 2x2, stores sc0 nt");
    CAND_F(2, 2, 256, "float lanes 2x2, stores sc1");
    CAND_F(2, 2, 288, "float lanes 2x2, stores sc1 nt");
    CAND_F(2, 2, 304, "float lanes 2x2, stores sc1 sc0 nt");
    CAND_FK(2, 2, 0, 1, "float lanes 2x2, keep all");
    CAND_F(2, 2, 1, "float lanes 2x2, no stores");
    CAND_F(2, 2, 2, "float lanes 2x2, no loads");
    CAND_F(2, 2, 3, "float lanes 2x2, no loads, no stores");
    CAND_F(1, 4, 0, "float lanes 1 level x 4 waves");
    CAND_F(1, 4, 1, "float lanes 1x4, no stores");
    CAND_F(2, 1, 0, "float lanes 2 levels x 1 wave");
    CAND_F(2, 1, 1, "float lanes 2x1, no stores");
    CAND_F(1, 2, 0, "float lanes 1 level x 2 waves");
    CAND_F(1, 1, 0, "float lanes 1 level x 1 wave");
#define CAND_PS(LPL, NP, DSV, NAME) CAND_PSA(LPL, NP, DSV, 0, NAME)
#define CAND_PSA(LPL, NP, DSV, ABL, NAME) do { \
        constexpr int CW = 64 * LPL * NP; \
        if (Dp % CW == 0 && (Dp == CW || CW == 256)) { \
        auto f = [&] { dim3 g(H, Dp / CW); \
            smk::fused_rawbox_h_ps_kernel<9, false, 1, LPL, NP, DSV, ABL><<<g, 64 * (NP + 1), (size_t)2 * 9 * CW * 4 + (size_t)W * 4>>>(fa, T1, tyshift, Dp, scale, 0); }; \
        CK(hipMemset(T1, 0xFF, n * 4)); \
        float al = timeit(f); report(NAME, time_pair(f, [&] { passB(T1); }), al); \
        if (check && !ABL) bad += compare(T0, T1, n) != 0; } \
    } while (0)
    CAND_PS(1, 4, DS, "prod/storer 4x1 levels");
    CAND_PSA(1, 4, DS, 1, "prod/storer 4x1, no stores");
    CAND_PSA(2, 2, DS, 1, "prod/storer 2x2, no stores");
    CAND_PSA(4, 1, DS, 1, "prod/storer 1x4, no stores");
    CAND_PS(1, 4, 0, "prod/storer 4x1 levels, any step");
    CAND_PS(2, 2, DS, "prod/storer 2x2 levels");
    CAND_PS(4, 1, DS, "prod/storer 1x4 levels");
    CAND_PS(1, 2, DS, "prod/storer 2x1 levels");
    CAND_PS(2, 1, DS, "prod/storer 1x2 levels");
    CAND_PS(1, 1, DS, "prod/storer 1x1 levels");
#define STORE(LPL, RPW, NAME) do { \
        auto f = [&] { dim3 g((H + RPW - 1) / RPW, Dp / (64 * LPL)); \
            store_only_kernel<LPL><<<g, 64 * RPW>>>(T1, W, H, Dp, tyshift, 1.37f); }; \
        float al = timeit(f); report(NAME, time_pair(f, [&] { passB(T1); }), al); \
    } while (0)
    if (Dp % 256 == 0) {
        CAND(4, 1, "4 levels/lane, 1 row/WG");
        CAND(4, 2, "4 levels/lane, 2 rows/WG");
    }
    if (Dp % 128 == 0) {
        CAND(2, 1, "2 levels/lane, 1 row/WG");
        CAND(2, 2, "2 levels/lane, 2 rows/WG");
    }
    if (check) {
        if (Dp % 256 == 0) { STORE(4, 1, "store only, x4, 1 row/WG"); STORE(4, 4, "store only, x4, 4 rows/WG"); }
        if (Dp % 128 == 0) { STORE(2, 1, "store only, x2, 1 row/WG"); STORE(2, 2, "store only, x2, 2 rows/WG"); STORE(2, 4, "store only, x2, 4 rows/WG"); }
        STORE(1, 1, "store only, x1, 1 row/WG"); STORE(1, 2, "store only, x1, 2 rows/WG"); STORE(1, 4, "store only, x1, 4 rows/WG");
#define THINK(LPL, RPW, K, NAME) do { \
        auto f = [&] { dim3 g((H + RPW - 1) / RPW, Dp / (64 * LPL)); \
            store_only_kernel<LPL, K><<<g, 64 * RPW>>>(T1, W, H, Dp, tyshift, 1.37f); }; \
        float al = timeit(f); report(NAME, time_pair(f, [&] { passB(T1); }), al); \
    } while (0)
        THINK(1, 1, 4, "store x1 + 4 VALU/level"); THINK(1, 1, 8, "store x1 + 8 VALU/level"); THINK(1, 1, 12, "store x1 + 12 VALU/level");
        THINK(1, 1, 24, "store x1 + 24 VALU/level");
        if (Dp % 128 == 0) { THINK(2, 1, 4, "store x2 + 4 VALU/level"); THINK(2, 1, 8, "store x2 + 8 VALU/level"); THINK(2, 1, 12, "store x2 + 12 VALU/level"); }
        if (Dp % 256 == 0) { THINK(4, 1, 4, "store x4 + 4 VALU/level"); THINK(4, 1, 8, "store x4 + 8 VALU/level"); THINK(4, 1, 12, "store x4 + 12 VALU/level"); }
        float tm = timeit([&] { CK(hipMemsetAsync(T1, 0, (size_t)W * H * D * 4)); });
        printf("  memset of the same bytes %.3f ms %6.0f GB/s\n", tm, B / tm / 1e6);
    }
    CK(hipFree(dr)); CK(hipFree(dm)); CK(hipFree(T0)); CK(hipFree(T1)); CK(hipFree(keys));
    return bad;
}

int main(int argc, char **argv)
{
    int bad = 0;
    bool quick = argc > 1 && !strcmp(argv[1], "quick");
    if (argc > 1 && !strcmp(argv[1], "ty")) {
        for (int ty : {0, 1, 2, 3, 4, 5, 6}) run<-1>(1920, 1080, 256, 0, 1000, false, ty);
        return 0;
    }
    bad += run<-1>(1920, 1080, 256, 0, 1000, true);
    if (!quick) {
        bad += run<1>(1920, 1080, 256, 0, 1000, true);
        bad += run<-1>(1920, 1080, 250, 3, 2, true);       // D % 4 != 0, truncation
        bad += run<-1>(1000, 70, 384, 0, 1000, true);      // two chunks, odd sizes
        bad += run<-1>(1920, 1080, 128, 0, 1000, true);
        bad += run<-1>(7680, 1080, 128, 896, 1000, true);  // config 5's last slab (a quarter of the rows)
    }
    printf(bad ? "FAILED\n" : "all equal\n");
    return bad != 0;
}
