// smhip_iter_px.hip -- instances and dispatcher of the fused iterated-binomial kernels (smhip_iter_px.h).
#define SMK_GLOBAL static __global__ __attribute__((unused))
#include "smhip_iter_px.h"
#include "smhip_iter_px_launch.h"

namespace smk {

int iter_px_levels(int src, int sink) { return (src == IPX_RAW && sink == IPX_KEYS) ? 16 : 32; }
int iter_px_tile_pixels(int K) { return px_txi(K); }

size_t iter_px_dynamic_lds(int src, int sink, int nb, int dstep)
{
    if (src != IPX_RAW)
        return 0;
    const int lv = iter_px_levels(src, sink), a = dstep < 0 ? -dstep : dstep;
    return (size_t)2 * (2 * PX_TXT + (lv - 1) * a) * (nb == 4 ? 8 : 4);
}

template <int K, int LV, int SRC, int SINK>
static bool launch_k(int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds, hipStream_t s)
{
    if (SRC != IPX_RAW) {                       // the image format does not reach the kernel
        iter_px_kernel<K, LV, SRC, SINK, false, 2><<<grid, LV * 32, lds, s>>>(g);
        return true;
    }
    if (nb == 2) {
        if (sd) iter_px_kernel<K, LV, SRC, SINK, true, 2><<<grid, LV * 32, lds, s>>>(g);
        else    iter_px_kernel<K, LV, SRC, SINK, false, 2><<<grid, LV * 32, lds, s>>>(g);
        return true;
    }
    if (nb == 4) {
        if (sd) iter_px_kernel<K, LV, SRC, SINK, true, 4><<<grid, LV * 32, lds, s>>>(g);
        else    iter_px_kernel<K, LV, SRC, SINK, false, 4><<<grid, LV * 32, lds, s>>>(g);
        return true;
    }
    return false;
}

template <int LV, int SRC, int SINK>
static bool launch_s(int K, int nb, bool sd, const IterPxArgs &g, dim3 grid, size_t lds, hipStream_t s)
{
    switch (K) {
    case 1: return launch_k<1, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    case 2: return launch_k<2, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    case 3: return launch_k<3, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    case 4: return launch_k<4, LV, SRC, SINK>(nb, sd, g, grid, lds, s);
    default: return false;
    }
}

bool launch_iter_px(int K, int src, int sink, int nb, bool sd, const IterPxArgs &g, unsigned grid_x, unsigned grid_y,
                    unsigned grid_z, hipStream_t stream)
{
    const dim3 grid(grid_x, grid_y, grid_z);
    const size_t lds = iter_px_dynamic_lds(src, sink, nb, g.a.dstep);
    if (src == IPX_RAW && sink == IPX_KEYS)
        return launch_s<16, IPX_RAW, IPX_KEYS>(K, nb, sd, g, grid, lds, stream);
    if (src == IPX_RAW && sink == IPX_DSI)
        return launch_s<32, IPX_RAW, IPX_DSI>(K, nb, sd, g, grid, lds, stream);
    if (src == IPX_DSI && sink == IPX_KEYS)
        return launch_s<32, IPX_DSI, IPX_KEYS>(K, nb, sd, g, grid, lds, stream);
    return false;
}

}  // namespace smk
