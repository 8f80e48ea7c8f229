// smhip_fused_defs.h -- what every translation unit of the fused path shares: the argument block of
// pass A and the layout of the private slab T.  No kernels here (the float-lane pass A is a translation
// unit of its own, smhip_fused_f.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smk {

__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// Slab layout shared by pass A and pass B: rows are grouped in bands of TY = 1 << tyshift
// rows; element (y, x, k) lives at
//     (((y >> tyshift) * W + x) << tyshift) * Dp + (y & (TY-1)) * Dp + k.
// tyshift = 0 is the plain [y][x][k] layout.  Bands make the H concurrent row streams of
// pass A land in H/TY compact windows (measured: 4..10 % faster slab writes, reads unchanged).
__device__ __forceinline__ size_t slab_row_offset(int y, int W, int Dp, int tyshift)
{
    return (((size_t)(y >> tyshift) * W) << tyshift) * Dp + (size_t)(y & ((1 << tyshift) - 1)) * Dp;
}

struct FusedA {
    const uint8_t *rn, *rx;     // reference row values / upper interval end (== rn unless match_interpolated)
    const uint8_t *mn, *mx;     // matching row lower / upper interval end (== values when no BT)
    size_t pitch;
    int W, H, D;                // D = levels in the slab
    int disp0, dstep;           // disp(k) = disp0 + k*dstep
    int cutoff, outside;
    int bt_inline;              // 1: mn is the matching image itself, its Birchfield-Tomasi interval is
                                // formed while the row is staged (one band, whole-row kernel)
};

}  // namespace smk
