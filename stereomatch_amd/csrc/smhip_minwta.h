// smhip_minwta.h -- separable min filter + PadCosts + OptWTA -> keys on a resident DSI without
// writing the filtered volume and without global atomics (smhip_min_pad_wta_keys; reference sequence
// MinFilter.cpp:122-202, StcRawCosts.cpp:437-457, StcOptimize.cpp:43-52).
//
// Workgroup = 64 staged pixel columns x 32 levels (thread = one pixel column x 4 levels, 8 lanes per
// pixel) walking a segment of at most MW_SEG rows; the level tiles of the slab are the OUTER loop, so
// the best key of every pixel of the segment stays in LDS across the level tiles and is stored once at
// the end -- the first version merged the level tiles with one 64-bit atomic min per pixel, row and tile,
// and device-scope atomics are executed by the memory side on this part (16.6 M read-modify-writes per
// 1920x1080x256 volume: TCC_EA0_WRREQ = TCC_ATOMIC in profiles/r02_exp2_atomics_kernel.json; 0.76 ms then,
// 0.66 ms now, counters of this kernel in profiles/r02_exp2_counters.json).
//
// Order of the two filter directions: vertical first, in registers.  Each thread loads the centre
// element of its own column once per row (no halo in registers, MW_PF rows in flight) and keeps
// the van Herk / Gil-Werman state of the vertical window: rows come in groups of WIN, the window of an
// output row is a suffix of the previous group plus a prefix of the current one, i.e. (3 WIN - 3) / WIN
// minima per row instead of WIN - 1.  The vertically filtered row goes through LDS (double buffered, one
// barrier per row); the 64 - (WIN - 1) output pixels of the tile take their WIN horizontal neighbours from
// there.  min is exact, so the order of the directions does not matter for the result.
#pragma once

#include "smhip_stencil.h"

namespace smk {

constexpr int MW_SEG = 64;      // rows per segment (bounds the key array in LDS)
constexpr int MW_PF = 4;        // rows a thread keeps in flight
constexpr int MW_KSTRIDE = 65;  // keys per pixel in LDS: odd, so the 8 pixels a wave updates hit 8 banks

// Minimum over each aligned group of 8 lanes, delivered to all 8 (xor 1, xor 2, mirror of the half row).
// One DPP instruction per step; the s_nop covers the two wait states a DPP read needs after the VALU
// write of its source, which the compiler cannot see inside the asm.
__device__ __forceinline__ float seg8_allmin_f32(float x)
{
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
}
__device__ __forceinline__ int seg8_allmin_u32(int x)
{
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
}

// TXS = staged pixel columns per tile (64 or 128; threads = 8 TXS).  The level tiles of the slab are dealt to
// gridDim.z groups (a group walks tiles_per_group of them and writes a key image of its own at
// keys + blockIdx.z * part_stride; the host merges the groups with keys_min_kernel): long row segments AND enough
// workgroups, where round 2's kernel had to shorten the segments of a 1080-row frame to 32 rows -- a third of its
// rows were halo.  The best key of a pixel is kept as ordered cost (int32) + level inside the slab (uint16): a
// 120-pixel x 64-row segment fits beside the staging buffers.  Tiles come in ascending level order, so a later
// tile replaces the stored key only with a strictly lower cost (OptWTA's strict '<').
template <int WIN, int TXS>
__global__ __launch_bounds__(TXS * 8) void min_wta_tile_kernel(const float *__restrict__ src, int W, int D, int H,
                                                               size_t row_elems, int seglen, PadWtaArgs pw,
                                                               int tiles_per_group, size_t part_stride)
{
    typedef typename fvec<4>::type vec;
    constexpr int PR = WIN / 2, PL = PR - WIN;
    constexpr int TXO = TXS - (WIN - 1);                // output pixels per tile
    constexpr int NT = TXS * 8;
    __shared__ vec vbuf[2][(TXS + WIN - 1) * 8];        // the tail is only read by threads without an output
    __shared__ int kcost[TXO * MW_KSTRIDE];             // [pixel][row of the segment]: best ordered cost ...
    __shared__ unsigned short klev[TXO * MW_KSTRIDE];   // ... and its level inside the slab
    const int t = threadIdx.x, hp = t >> 3, hl = t & 7;
    const int x0 = blockIdx.x * TXO, y0 = blockIdx.y * seglen, y1 = min(H, y0 + seglen);
    for (int i = t; i < TXO * MW_KSTRIDE; i += NT)
        kcost[i] = 0x7FFFFFFF;                          // (the row loop's barriers order this before its use)
    const int xs = clampi(x0 + PL + 1 + hp, W);         // staged pixel of this thread (replicate border)
    const bool has_out = hp < TXO && x0 + hp < W;       // output pixel x0 + hp, window = staged hp .. hp + WIN - 1
    const int ylast = y1 - 1 + PR;
    int buf = 0;
    const int kfirst = (int)blockIdx.z * tiles_per_group * 32, kend = min(D, kfirst + tiles_per_group * 32);
    for (int k0 = kfirst; k0 < kend; k0 += 32) {
        const bool lact = k0 + hl * 4 < D;
        const float *col = src + (size_t)xs * D + min(k0 + hl * 4, D - 4);
        auto gload = [&](int yy) -> vec { return vload<4>(col + (size_t)clampi(yy, H) * row_elems); };
        bool padded[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int xm = x0 + hp + pw.disp0 + (k0 + hl * 4 + e) * pw.dstep;
            padded[e] = (unsigned)xm >= (unsigned)W;
        }
        const bool live = has_out && lact;
        vec ring[WIN], suf[WIN], pre = 0.0f;
#pragma unroll
        for (int j = 0; j < WIN; j++)
            ring[j] = suf[j] = 0.0f;
        vec g[MW_PF];                                   // centre elements of the next MW_PF rows, in flight
#pragma unroll
        for (int j = 0; j < MW_PF; j++)
            g[j] = gload(y0 + PL + 1 + j);
        for (int yb = y0 + PL + 1; yb <= ylast; yb += WIN) {
#pragma unroll
            for (int rr = 0; rr < WIN; rr++) {
                const int yy = yb + rr;
                if (yy <= ylast) {                      // (uniform)
                    ring[rr] = g[0];
#pragma unroll
                    for (int j = 0; j + 1 < MW_PF; j++)
                        g[j] = g[j + 1];
                    if (yy + MW_PF <= ylast)
                        g[MW_PF - 1] = gload(yy + MW_PF);
                    pre = rr == 0 ? ring[0] : vmin<4>(pre, ring[rr]);
                    if (yy - PR >= y0) {                // (uniform) output row yy - PR
                        vbuf[buf][t] = rr == WIN - 1 ? pre : vmin<4>(suf[rr == WIN - 1 ? 0 : rr + 1], pre);
                        __syncthreads();
                        vec m = vbuf[buf][hp * 8 + hl];
#pragma unroll
                        for (int j = 1; j < WIN; j++)
                            m = vmin<4>(m, vbuf[buf][(hp + j) * 8 + hl]);
                        // PadCosts, then the lowest cost of the pixel's 32 levels and the lowest level holding
                        // it: float minimum of the lane's four values, butterfly over the pixel's 8 lanes
                        // (every lane then knows the minimum), each lane names its lowest level equal to it,
                        // integer minimum of those names
                        float c[4];
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            c[e] = padded[e] ? pw.outside : m[e];
                        float bc = fmin_hw(fmin_hw(c[0], c[1]), fmin_hw(c[2], c[3]));
                        bc = live ? bc : __builtin_inff();
                        const float mm = seg8_allmin_f32(bc);
                        const int lev = k0 + hl * 4;
                        int cand = 0x7FFFFFFF;
#pragma unroll
                        for (int e = 3; e >= 0; e--)
                            cand = c[e] == mm ? lev + e : cand;
                        cand = live ? cand : 0x7FFFFFFF;
                        cand = seg8_allmin_u32(cand);
                        if (hl == 0 && has_out) {
                            const int idx = hp * MW_KSTRIDE + (yy - PR - y0), oc = ordered_bits(mm);
                            if (oc < kcost[idx]) {
                                kcost[idx] = oc;
                                klev[idx] = (unsigned short)cand;
                            }
                        }
                        buf ^= 1;
                    }
                }
            }
            suf[WIN - 1] = ring[WIN - 1];               // suffix minima of the group just completed
#pragma unroll
            for (int j = WIN - 2; j >= 1; j--)
                suf[j] = vmin<4>(ring[j], suf[j + 1]);
        }
    }
    __syncthreads();
    const int rows = y1 - y0;
    long long *keys = pw.keys + (size_t)blockIdx.z * part_stride;
    for (int i = t; i < TXO * MW_SEG; i += NT) {        // column-major keys: a pixel's rows are contiguous
        int px = i / MW_SEG, r = i % MW_SEG;
        if (x0 + px < W && r < rows)
            keys[(size_t)(x0 + px) * H + y0 + r] = make_key(kcost[px * MW_KSTRIDE + r],
                                                            pw.d_first + (int)klev[px * MW_KSTRIDE + r]);
    }
}

}  // namespace smk
