// smhip_iter_px_launch.h -- host-side entry of the translation units built from smhip_iter_px.hip (the fused
// iterated-aggregation kernels of smhip_iter_px.h are compiled on their own, once per operator -- -DIPX_OP=0
// binomial, -DIPX_OP=1 diffusion -- side by side with the rest of the library).
#pragma once

#include "smhip_fused_defs.h"
#include "smhip_stencil.h"

namespace smk {

struct IterPxArgs;

// Launches iter_px_kernel<op, K, LV, src, sink, sd, nb>; LV = 16 for RAW -> KEYS, 32 when a DSI is read or written.
// op: 0 binomial (K <= 4), 1 diffusion (K <= 6).  nb: 2 = one band with a point reference interval, 4 = four
// bands.  Returns false when no instance covers the request (the caller then takes the staged route).
bool launch_iter_px(int op, int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, unsigned grid_x,
                    unsigned grid_y, unsigned grid_z, hipStream_t stream);
bool launch_iter_px_binomial(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds,
                             hipStream_t stream);
bool launch_iter_px_diffusion(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds,
                              hipStream_t stream);
int iter_px_levels(int src, int sink);                  // levels per tile (LV) of that combination
int iter_px_tile_pixels(int op, int K);                 // output pixels per tile
int iter_px_max_k(int op);                              // iterations one kernel holds
int iter_px_halo_rows(int op, int K);
size_t iter_px_dynamic_lds(int src, int sink, int nb, int dstep);

}  // namespace smk
