// smhip_fused_chunk.h -- pass A of the fused path for rows too long to stage whole.
//
// fused_rawbox_h_lds_kernel (smhip_fused.h) keeps the full image row in LDS: 4 bytes per pixel and
// band.  A 4-band 7680-pixel row is 122 KB -- one workgroup per CU, and pass A 3.4 times slower
// than on a 1-band frame.  This variant stages the row in chunks of chunk_steps positions (a
// multiple of the static window), so the LDS footprint and the number of resident workgroups do
// not depend on the width.  The walk itself is the same, value for value.  For rows that fit, the
// whole-row kernel is 3-9 % faster (no chunk barriers, no index offsets), so the host picks this
// one only above 40 KB of row.
#pragma once

#include "smhip_fused.h"

namespace smk {

template <int WIN, bool SD, int NB>
__global__ __launch_bounds__(256) void fused_rawbox_h_chunk_kernel(FusedA a, float *__restrict__ T,
                                                                   int tyshift, int Dp, int win,
                                                                   float scale, int keep_all,
                                                                   int chunk_steps, int nref_cap, int nmatch_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_c[];
    const int W = a.W;
    RowLds<NB> L;
    L.init2(lds_c, nref_cap, nmatch_cap);
    const int y = blockIdx.x;
    const int kf0 = blockIdx.y * blockDim.x;            // first level of the workgroup
    if (kf0 >= a.D)
        return;                                         // whole workgroup, before any barrier
    const int k = kf0 + threadIdx.x;
    const bool active = k < a.D;
    const size_t ro = (size_t)y * a.pitch;
    const int disp = a.disp0 + (active ? k : a.D - 1) * a.dstep;
    const int cutoff = a.cutoff, outside = a.outside;
    float *t = T + slab_row_offset(y, W, Dp, tyshift) + k;
    const size_t xs = (size_t)Dp << tyshift;           // floats between consecutive x
    const int w = WIN ? WIN : win;
    const int pr = w / 2, pl = pr - w;
    // disparity range of the workgroup's levels: which matching pixels a chunk of steps can touch
    const int klw = min(kf0 + (int)blockDim.x - 1, a.D - 1);
    const int dwa = a.disp0 + kf0 * a.dstep, dwb = a.disp0 + klw * a.dstep;
    const int dlo = min(dwa, dwb), dhi = max(dwa, dwb);
    int rlo = 0, mlo = 0;                               // first staged reference / matching pixel

    // MatchLine (StcRawCosts.cpp:152-198) for reference pixel x at this lane's level
    auto cost = [&](int x) -> int {
        int xm = x + disp;
        bool out = (unsigned)xm >= (unsigned)W;
        int xc = clampi(xm, W);                         // (the value is not used when out)
        int e = min(L.template gap<SD>(x - rlo, xc - mlo), cutoff);
        return out ? outside : e;
    };

    // positions where no lane of the wave needs a bounds test (as in the whole-row kernel)
    const int kf = __builtin_amdgcn_readfirstlane(k);
    const int kl = min(kf + 63, a.D - 1);
    const int df = a.disp0 + kf * a.dstep, dl = a.disp0 + kl * a.dstep;
    const int qa = max(0, -min(df, dl)), qb = min(W, W - max(df, dl));

    float acc = 0;
    int ring[WIN ? WIN : 1];
    // chunk [xa, xe): reference pixels [xa + pl, xe - 1 + pr] and the matching pixels they can be
    // compared with at this workgroup's levels
    for (int xa = 0, xe; xa < W; xa = xe) {
        xe = min(W, (xa == 0 ? 1 : xa) + chunk_steps);
        const int rhi = min(W - 1, xe - 1 + pr);
        __syncthreads();                                // the previous chunk has been read
        rlo = max(0, xa + pl);
        mlo = clampi(rlo + dlo, W);
        const int mhi = clampi(rhi + dhi, W);
        for (int i = threadIdx.x; i <= rhi - rlo; i += blockDim.x)
            L.stage_r(a, ro, rlo + i, i);
        for (int i = threadIdx.x; i <= mhi - mlo; i += blockDim.x)
            L.stage_m(a, ro, mlo + i, i);
        __syncthreads();
        if (!active)
            continue;
        int xb = xa;
        if (xa == 0) {
            if (WIN) {
#pragma unroll
                for (int j = 0; j < WIN; j++) {
                    int c = cost(clampi(pl + 1 + j, W));
                    ring[j] = c;
                    acc += scale * (float)c;
                }
            } else {
                for (int i = pl + 1; i <= pr; i++)
                    acc += scale * (float)cost(clampi(i, W));
            }
            t[0] = acc;
            xb = 1;
        }
        if (WIN) {
            for (; xb + WIN <= xe; xb += WIN) {
                if (xb >= qa && xb + WIN - 1 + pr < qb) {
#pragma unroll
                    for (int s = 0; s < WIN; s++) {
                        int x = xb + s;
                        int c = min(L.template gap<SD>(x + pr - rlo, x + pr + disp - mlo), cutoff);
                        acc = acc + scale * (float)(c - ring[s]);
                        ring[s] = c;
                        t[xs * x] = acc;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < WIN; s++) {
                        int x = xb + s;
                        int c = cost(min(x + pr, W - 1));
                        acc = acc + scale * (float)(c - ring[s]);
                        ring[s] = c;
                        if (keep_all || (unsigned)(x + disp) < (unsigned)W)   // PadCosts overwrites the rest
                            t[xs * x] = acc;
                    }
                }
            }
            if (xe == W) {                              // the last, partial block of the row
#pragma unroll
                for (int s = 0; s < WIN; s++) {
                    int x = xb + s;
                    if (x < W) {
                        int c = cost(min(x + pr, W - 1));
                        acc = acc + scale * (float)(c - ring[s]);
                        ring[s] = c;
                        if (keep_all || (unsigned)(x + disp) < (unsigned)W)
                            t[xs * x] = acc;
                    }
                }
            }
        } else {
            for (int x = xb; x < xe; x++) {
                int cl = cost(max(x + pl, 0));
                int cr = cost(min(x + pr, W - 1));
                acc = acc + scale * (float)(cr - cl);
                if (keep_all || (unsigned)(x + disp) < (unsigned)W)
                    t[xs * x] = acc;
            }
        }
    }
}

}  // namespace smk
