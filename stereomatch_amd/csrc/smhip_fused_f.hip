// smhip_fused_f.hip -- instantiations and dispatch of the float-lane pass A (smhip_fused_f.h).
// Built with -fno-slp-vectorize (see the header of smhip_fused_f.h) and linked into libsmhip.so.
// The file is compiled six times (__graft_entry__.build_hip): once per window with -DSMK_F_WIN=3 / 5 / 7 / 9 / 11,
// each giving launch_fused_f_window<WIN>, and once without for the dispatcher -- the instantiations of one
// window take about a minute of compile time, and the build runs them side by side.
#include "smhip_fused_f.h"
#include "smhip_fused_f_launch.h"

namespace smk {

template <int WIN>
void launch_fused_f_window(const FusedA &a, float *T, int tyshift, int Dp, float scale, int keep_all, int glev, bool sd,
                           bool cut, hipStream_t stream, int policy, bool paired, long lds_min, int one_wave);

#ifdef SMK_F_WIN

namespace {

template <int WIN, bool SD, bool CUT, int DS, int POL>
void launch_shape_pol(const FusedA &a, float *T, int tyshift, int Dp, float scale, int keep_all, int glev,
                      hipStream_t stream, bool paired, long lds_min, int one_wave)
{
    // 256-level chunks: two waves of two levels per lane; a 128-level slab: one wave
    // one_wave (option a_nw): 1 = one-wave workgroups of 128 levels on 256-level chunks too; 4 = one wave of FOUR
    // levels per lane per 256-level chunk (1-KiB wave stores; window 9 only)
    const bool lpl4 = one_wave == 4 && WIN == 9 && Dp % 256 == 0 && !paired && glev == 4;
    // One-wave workgroups on a single 256-level chunk of frames of 1025..1280 rows (config 3: 2160 workgroups, 8.4
    // per CU, instead of 1080 two-wave ones, 4.2 per CU): pass A 0.352 -> 0.347 ms, windows 3..9; slower on every
    // other shape measured (512 / 1024 levels 2-5 %, window 11 4 %, 2560x1440 1 %; profiles/r03_pass_a_one_wave.log)
    if (one_wave < 0)
        one_wave = (Dp == 256 && WIN <= 9 && a.W < 2560 && a.H > 1024 && a.H <= 1280) ? 1 : 0;
    const int nw = (Dp % 256 == 0 && !one_wave) ? 2 : 1;
    const int cw = lpl4 ? 256 : 128 * nw;
    int cbs = fused_f_ni(lpl4 ? 4 : 2) * 64 * nw - WIN - cw;   // staged pixels per thread bounded by fused_f_ni()
    if (cbs > 480) cbs = 480;
    const int cb = cbs / WIN, nref = cb * WIN + WIN, nmatch = nref + cw;
    // Workgroups resident per CU, bounded through the LDS a workgroup asks for (160 KB per CU).  All rows take
    // the same time, so the grid runs in rounds; fewer, fuller rounds of few waves write the slab faster than
    // every workgroup at once (scripts/exp_occupancy*.sh, profiles/r02_occupancy.log): one-wave workgroups
    // (128-level slabs) 6 per CU -- 8K 3.11 -> 2.88 ms, 4K 0.848 -> 0.795, 1080p unchanged; two-wave workgroups
    // 2 per CU from 1280 rows on -- 4K 1.44 -> 1.38 ms, 8K 5.63 -> 5.30 (0.80 of 8 TB/s, the write ceiling),
    // but 1080 rows (two and a bit rounds of 512) 0.380 -> 0.419, hence the row bound; and only from 2560 columns on
    // (below).  lds_min < 0: this policy.
    size_t lds = (size_t)nmatch * 8 + (size_t)nref * 4;
    if (lds_min < 0)
    {
        // one-wave workgroups: five or six per CU (31 / 26 KB), whichever fills its rounds better -- 4K x 128 (2160
        // rows): 0.794 ms at six (two rounds of 1536, 0.70 full), 0.746 at five (two of 1280, 0.84); 5120x2880: 1.293
        // at six (0.94) against 1.334 at five (0.75); 8K: 2.897 / 2.893 (0.94 / 0.84)
        auto fill = [&](long per_cu) {
            const long slots = 256 * per_cu, wgs = (long)a.H * (Dp / cw);
            return (double)wgs / (double)((wgs + slots - 1) / slots * slots);
        };
        const long one_wave_lds = fill(5) > fill(6) + 0.05 ? 31 * 1024 : 26 * 1024;
        lds_min = a.W < 2560 ? 0 : nw == 1 ? one_wave_lds : (a.H >= 1280 && Dp <= 512 ? 54 * 1024 : 0);
    }
    // (four chunks, 8K x 1024: 22.4 against 22.1 ms without.  Rows shorter than 2560 pixels lose under a bound:
    // 2048x1536 3 %, 1280x1400 11 %, 800x1300x512 15 %, 640x4000x128 7 % -- profiles/r02_occupancy.log, last part.)
    if (lds < (size_t)lds_min) lds = (size_t)lds_min;
    dim3 grid((unsigned)a.H * (unsigned)(Dp / cw));        // row-major over (row, chunk)
    if constexpr (WIN == 9) {
        if (lpl4) {
            fused_rawbox_h_f_kernel<WIN, SD, CUT, DS, 4, 1, 0, POL><<<grid, 64, lds, stream>>>(a, T, tyshift, Dp, scale,
                                                                                      keep_all, cb, nref, nmatch, glev);
            return;
        }
    }
    if (nw == 2 && WIN > 9 && a.H < 1280) {                // all rows resident: three waves per SIMD (see the kernel)
        if constexpr (WIN > 9)
            fused_rawbox_h_f_kernel<WIN, SD, CUT, DS, 2, 2, 0, POL, false, true><<<grid, 128, lds, stream>>>(
                a, T, tyshift, Dp, scale, keep_all, cb, nref, nmatch, glev);
    } else if (nw == 2)
        fused_rawbox_h_f_kernel<WIN, SD, CUT, DS, 2, 2, 0, POL><<<grid, 128, lds, stream>>>(a, T, tyshift, Dp, scale, keep_all,
                                                                                   cb, nref, nmatch, glev);
    else if (paired)                                       // 128-level slab, columns in pairs (pass B PAIR)
        fused_rawbox_h_f_kernel<WIN, SD, CUT, DS, 2, 1, 0, POL, true><<<grid, 64, lds, stream>>>(
            a, T, tyshift, Dp, scale, keep_all, cb, nref, nmatch, glev);
    else
        fused_rawbox_h_f_kernel<WIN, SD, CUT, DS, 2, 1, 0, POL><<<grid, 64, lds, stream>>>(a, T, tyshift, Dp, scale, keep_all,
                                                                                  cb, nref, nmatch, glev);
}

// Cache policy of the slab stores (aux bits of buffer_store on gfx950: 2 = nt, 16 = sc1).  Measured in the
// pipeline (scripts/exp_store_policy.sh, pass A / pass B ms): config 3 plain 0.387 / 0.331, sc1 0.380 / 0.331,
// nt 0.410 / 0.315, sc1 + nt 0.396 / 0.316 (step 0.728 / 0.721 / 0.735 / 0.722); 3840x2160x256 step 3.12 plain,
// 3.08 sc1, 2.97 sc1 + nt; 7680x4320x256 11.99 / 11.84 / 11.59; 7680x4320x128 6.81 / 6.69 / 6.52.  nt stores
// leave no dirty lines in the L2 for pass B's first reads to wait behind, sc1 makes pass A itself faster; on
// slabs of a few GB the two effects cancel, on larger ones sc1 + nt wins.  policy < 0: the library's choice
// (sc1 below 4 GiB of slab, sc1 + nt from there on); 0 / 16 / 18 through SMHIP_OPT_STORE_POLICY.  (The integer
// kernels keep plain stores: non-temporal ones made the 4-band pass A 6 % slower for a 3 % faster pass B.)
template <int WIN, bool SD, bool CUT, int DS>
void launch_shape(const FusedA &a, float *T, int tyshift, int Dp, float scale, int keep_all, int glev,
                  hipStream_t stream, int policy, bool paired, long lds_min, int one_wave)
{
    if (policy < 0)
        policy = (size_t)a.W * a.H * Dp * sizeof(float) < ((size_t)4 << 30) ? 16 : 18;
    if (policy == 0)
        launch_shape_pol<WIN, SD, CUT, DS, 0>(a, T, tyshift, Dp, scale, keep_all, glev, stream, paired, lds_min, one_wave);
    else if (policy == 16)
        launch_shape_pol<WIN, SD, CUT, DS, 16>(a, T, tyshift, Dp, scale, keep_all, glev, stream, paired, lds_min, one_wave);
    else
        launch_shape_pol<WIN, SD, CUT, DS, 18>(a, T, tyshift, Dp, scale, keep_all, glev, stream, paired, lds_min, one_wave);
}

}  // namespace

template <>
void launch_fused_f_window<SMK_F_WIN>(const FusedA &a, float *T, int tyshift, int Dp, float scale, int keep_all, int glev,
                                      bool sd, bool cut, hipStream_t stream, int policy, bool paired, long lds_min, int one_wave)
{
    constexpr int WIN = SMK_F_WIN;
#define SMK_F(SD, CUT)                                                                             \
    do {                                                                                           \
        if (a.dstep < 0) launch_shape<WIN, SD, CUT, -1>(a, T, tyshift, Dp, scale, keep_all, glev, stream, policy, paired, lds_min, one_wave); \
        else launch_shape<WIN, SD, CUT, 1>(a, T, tyshift, Dp, scale, keep_all, glev, stream, policy, paired, lds_min, one_wave); \
    } while (0)
    if (sd) { if (cut) SMK_F(true, true); else SMK_F(true, false); }
    else    { if (cut) SMK_F(false, true); else SMK_F(false, false); }
#undef SMK_F
}

#else   // the dispatcher

bool launch_fused_rawbox_h_f(const FusedA &a, int n_bands, float *T, int tyshift, int Dp, int window, float scale,
                             int keep_all, int glev, bool sd, int max_cost, hipStream_t stream, int policy, bool paired, long lds_min, int one_wave)
{
    if (n_bands != 1 || (a.dstep != 1 && a.dstep != -1) || a.rn != a.rx || Dp % 128 != 0)
        return false;
    if (glev != 2 && glev != 4)
        return false;
    if (paired && (Dp != 128 || keep_all || glev != 4))
        return false;
    // buffer-store offsets are 32-bit: the row's extent in the slab and twice that must fit
    const unsigned long long extent = (unsigned long long)a.W * ((unsigned long long)Dp << tyshift) * 4ull;
    if (extent >= (1ull << 31) || a.cutoff < 0 || a.cutoff >= (1 << 24))
        return false;
    const bool cut = a.cutoff < max_cost;                  // truncation can bite
    switch (window) {
    case 3: launch_fused_f_window<3>(a, T, tyshift, Dp, scale, keep_all, glev, sd, cut, stream, policy, paired, lds_min, one_wave); break;
    case 5: launch_fused_f_window<5>(a, T, tyshift, Dp, scale, keep_all, glev, sd, cut, stream, policy, paired, lds_min, one_wave); break;
    case 7: launch_fused_f_window<7>(a, T, tyshift, Dp, scale, keep_all, glev, sd, cut, stream, policy, paired, lds_min, one_wave); break;
    case 9: launch_fused_f_window<9>(a, T, tyshift, Dp, scale, keep_all, glev, sd, cut, stream, policy, paired, lds_min, one_wave); break;
    case 11: launch_fused_f_window<11>(a, T, tyshift, Dp, scale, keep_all, glev, sd, cut, stream, policy, paired, lds_min, one_wave); break;
    default: return false;
    }
    return true;
}

#endif

}  // namespace smk
