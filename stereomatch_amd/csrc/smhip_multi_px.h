// smhip_multi_px.h -- the "pixels on lanes" layout of the iterated stencils (K iterations of the separable binomial
// 14641, Convolve.cpp:86-254, or of the diffusion, in one pass): its constants.  The kernels are
// in smhip_iter_px.h.
//
// The kernels of smhip_multi.h keep the DSI's own layout in registers (a thread = 4 levels of one pixel) and fetch
// the horizontal neighbours of every level and row through LDS: 1 write + 4 reads of 16 bytes per thread, level
// and row, which is what bounds them once the tap arithmetic is down to 6 instructions (LDS 160 clocks against
// 96 of VALU per wave, level and row).  Here a thread owns 4 CONSECUTIVE PIXELS of ONE level: the horizontal
// neighbours x-2, x-1, x+4, x+5 are two registers of the lane below and two of the lane above -- four DPP moves
// (wave_shr:1 / wave_shl:1), no LDS and no barrier inside the level pipeline.  LDS is only the transpose between
// the DSI's layout (levels contiguous: coalesced 64- or 128-byte pieces per pixel) and this one, once per row on
// the way in and once on the way out, both behind ONE barrier per row (buffers alternate with the row's parity).
//
// Workgroup = 128 pixels (halo of 2 K on either side included) x LV levels, LV * 32 threads; a wave holds two
// levels of the tile (lanes 0..31 / 32..63).  Lanes 0 and 32 receive a neighbour that belongs to another level's
// row, lanes 31 and 63 likewise: they are the outermost pixels of the tile, whose values only feed the halo (an
// error entering at the tile's edge moves 2 pixels inwards per iteration, 2 K in all = the halo).  Replicate
// border (TrimIndex clamp, Convolve.cpp:42-43): the thread that holds pixel 0 takes its own first value for x-1
// and x-2, the one that holds W-1 its last (W % 4 == 0 is required, so both are thread boundaries).
// Arithmetic, guard and ring handling as binomial_multi_kernel (smhip_multi.h): bit-identical results.
#pragma once

#include "smhip_multi.h"
#include "smhip_dpp.h"

namespace smk {

constexpr int PX_TXT = 128;         // pixels per tile, halo included
constexpr int PX_STRIDE = 132;      // floats per level row of the transpose buffers (16-byte rows, banks skewed by 4)
// halo on either side of a tile: 2 K pixels, rounded up to the four pixels of a thread (pixel 0 starts a thread)
constexpr int px_halo(int K) { return (2 * K + 3) / 4 * 4; }
constexpr int px_txi(int K) { return PX_TXT - 2 * px_halo(K); }

// (The DSI -> DSI kernel that introduced this layout, binomial_px_kernel, is gone: iter_px_kernel<IPX_BINOMIAL, K, 32,
// IPX_DSI, IPX_DSI> of smhip_iter_px.h is the same walk with the pixel order (0, 2, 1, 3) inside a thread and serves
// smhip_convolve_14641_run.)

}  // namespace smk
