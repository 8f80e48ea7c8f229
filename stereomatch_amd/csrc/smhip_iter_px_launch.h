// smhip_iter_px_launch.h -- host-side entry of the translation unit smhip_iter_px.hip (the fused iterated-binomial
// kernels of smhip_iter_px.h are compiled on their own, side by side with the rest of the library).
#pragma once

#include "smhip_fused_defs.h"
#include "smhip_stencil.h"

namespace smk {

struct IterPxArgs;

// Launches iter_px_kernel<K, LV, src, sink, sd, nb>; LV = 16 for RAW -> KEYS, 32 when a DSI is read or written.
// nb: 2 = one band with a point reference interval, 4 = four bands.  Returns false when no instance covers
// the request (the caller then takes the staged route).
bool launch_iter_px(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, unsigned grid_x, unsigned grid_y,
                    unsigned grid_z, hipStream_t stream);
int iter_px_levels(int src, int sink);                  // levels per tile (LV) of that combination
int iter_px_tile_pixels(int K);                         // output pixels per tile
size_t iter_px_dynamic_lds(int src, int sink, int nb, int dstep);

}  // namespace smk
