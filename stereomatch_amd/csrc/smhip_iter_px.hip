// smhip_iter_px.hip -- instances and dispatcher of the fused iterated-aggregation kernels (smhip_iter_px.h).
// Compiled once per operator (__graft_entry__.build_hip): -DIPX_OP=0 the binomial instances and the shared
// helpers, -DIPX_OP=1 the diffusion instances.
#define SMK_GLOBAL static __global__ __attribute__((unused))
#include "smhip_iter_px.h"
#include "smhip_iter_px_launch.h"

#ifndef IPX_OP
#error "compile with -DIPX_OP=0 (binomial) or -DIPX_OP=1 (diffusion)"
#endif

namespace smk {

namespace {

template <int OP, int K, int LV, int SRC, int SINK>
bool launch_k(int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds, hipStream_t s)
{
    if (SRC != IPX_RAW) {                       // the image format does not reach the kernel
        iter_px_kernel<OP, K, LV, SRC, SINK, false, 2><<<grid, LV * 32, lds, s>>>(g);
        return true;
    }
    if (nb == 2) {
        if (sd) iter_px_kernel<OP, K, LV, SRC, SINK, true, 2><<<grid, LV * 32, lds, s>>>(g);
        else    iter_px_kernel<OP, K, LV, SRC, SINK, false, 2><<<grid, LV * 32, lds, s>>>(g);
        return true;
    }
    if (nb == 4) {
        if (sd) iter_px_kernel<OP, K, LV, SRC, SINK, true, 4><<<grid, LV * 32, lds, s>>>(g);
        else    iter_px_kernel<OP, K, LV, SRC, SINK, false, 4><<<grid, LV * 32, lds, s>>>(g);
        return true;
    }
    return false;
}

template <int OP, int LV, int SRC, int SINK>
bool launch_s(int K, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds, hipStream_t s)
{
    switch (K) {
    case 1: return launch_k<OP, 1, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    case 2: return launch_k<OP, 2, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    case 3: return launch_k<OP, 3, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    case 4: return launch_k<OP, 4, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    default: break;
    }
    if constexpr (OP == IPX_DIFFUSION) {
        switch (K) {
        case 5: return launch_k<OP, 5, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
        case 6: return launch_k<OP, 6, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
        default: break;                         // (7 and 8 levels of three rows spill under 128 VGPRs)
        }
    }
    return false;
}

template <int OP>
bool launch_op(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds, hipStream_t stream)
{
    if (src == IPX_RAW && sink == IPX_KEYS)
        return launch_s<OP, 16, IPX_RAW, IPX_KEYS>(K, nb, sd, g, grid, lds, stream);
    if (src == IPX_RAW && sink == IPX_DSI)
        return launch_s<OP, 32, IPX_RAW, IPX_DSI>(K, nb, sd, g, grid, lds, stream);
    if (src == IPX_DSI && sink == IPX_KEYS)
        return launch_s<OP, 32, IPX_DSI, IPX_KEYS>(K, nb, sd, g, grid, lds, stream);
    if (src == IPX_DSI && sink == IPX_DSI)
        return launch_s<OP, 32, IPX_DSI, IPX_DSI>(K, nb, sd, g, grid, lds, stream);
    return false;
}

}  // namespace

#if IPX_OP == 0

int iter_px_levels(int src, int sink) { return (src == IPX_RAW && sink == IPX_KEYS) ? 16 : 32; }
int iter_px_tile_pixels(int op, int K) { return ipx_txi(op, K); }
int iter_px_max_k(int op) { return op == IPX_BINOMIAL ? 4 : 6; }
int iter_px_halo_rows(int op, int K) { return ipx_rows(op, K); }

size_t iter_px_dynamic_lds(int src, int sink, int nb, int dstep)
{
    if (src != IPX_RAW)
        return 0;
    const int lv = iter_px_levels(src, sink), a = dstep < 0 ? -dstep : dstep;
    return (size_t)2 * (2 * PX_TXT + (lv - 1) * a) * (nb == 4 ? 8 : 4);
}

bool launch_iter_px_binomial(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds,
                             hipStream_t stream)
{
    return launch_op<IPX_BINOMIAL>(K, src, sink, nb, sd, g, grid, lds, stream);
}

bool launch_iter_px(int op, int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, unsigned grid_x,
                    unsigned grid_y, unsigned grid_z, hipStream_t stream)
{
    const dim3 grid(grid_x, grid_y, grid_z);
    const size_t lds = iter_px_dynamic_lds(src, sink, nb, g.a.dstep);
    if (op == IPX_BINOMIAL)
        return launch_iter_px_binomial(K, src, sink, nb, sd, g, grid, lds, stream);
    if (op == IPX_DIFFUSION)
        return launch_iter_px_diffusion(K, src, sink, nb, sd, g, grid, lds, stream);
    return false;
}

#else

bool launch_iter_px_diffusion(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds,
                              hipStream_t stream)
{
    return launch_op<IPX_DIFFUSION>(K, src, sink, nb, sd, g, grid, lds, stream);
}

#endif

}  // namespace smk
