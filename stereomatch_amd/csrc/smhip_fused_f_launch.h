// smhip_fused_f_launch.h -- host-side entry of the float-lane pass A (smhip_fused_f.h), which lives in a
// translation unit of its own (smhip_fused_f.hip) because it is compiled with -fno-slp-vectorize.
#pragma once

#include <hip/hip_runtime.h>

#include "smhip_fused_defs.h"

namespace smk {

// Launches pass A in its float-lane form on `stream` when the configuration is one it covers: one band,
// |dstep| == 1, reference interval a point (a.rn == a.rx), window 3 / 5 / 7 / 9 / 11, Dp a multiple of 128,
// slab row extent below 2 GiB.  paired: the 128-level slab with its columns in pairs (smhip_fused_f.h, PAIRED;
// needs Dp == 128, glev == 4, keys out).  glev = levels per lane of the pass B that follows.  Returns false
// (nothing launched) otherwise; the caller then takes fused_rawbox_h_lds_kernel.
bool launch_fused_rawbox_h_f(const FusedA &a, int n_bands, float *T, int tyshift, int Dp, int window, float scale,
                             int keep_all, int glev, bool sd, int max_cost, hipStream_t stream, int policy, bool paired, long lds_min, int one_wave);

}  // namespace smk
