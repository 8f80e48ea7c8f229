// smhip_fused_ps.h -- pass A of the fused path, producer / storer form.
//
// Measured on the one-wave-does-everything kernels (smhip_fused.h, smhip_fused_a4.h): a wave that has
// to compute between two stores keeps the memory pipe less busy than a wave that only stores -- a
// store-only kernel in pass A's order writes the slab in 0.34-0.35 ms (6.1-6.3 TB/s) at config 3,
// the same stores behind 8-12 VALU instructions per level take 0.36-0.41 ms (scripts/bench_a.hip).
// So the two jobs are split between the waves of a workgroup (one image row x CW levels):
//   * NP producer waves evaluate the raw costs (StcRawCosts.cpp:152-198) and the horizontal running
//     sum (BoxFilter.cpp:92-168) exactly as fused_rawbox_h_lds_kernel does, LPL levels per lane, and
//     put WIN steps at a time into an LDS block buffer;
//   * one storer wave drains the previous block to the slab T with one wide store per step
//     (CW = 256: 1 KiB per wave-instruction) while the producers fill the other buffer.
// One workgroup barrier per WIN steps; the storer is (almost) always waiting in a store, as in the
// store-only kernel.  Arithmetic and operation order per chain are unchanged (bit-identical T).
#pragma once

#include "smhip_fused_a4.h"

namespace smk {

// blockDim.x = 64 * (NP + 1); grid (H, Dp / CW) with CW = 64 * LPL * NP levels per workgroup.
// DS = +1 / -1: unit disparity step (a lane with LPL > 1 levels reads one new matching pixel per
// step and shifts the others through registers); DS = 0: any step, every level reads its own.
// Dynamic LDS: 2 * WIN * CW * 4 bytes of block buffers, then the staged row (RowLds<NB>::bytes(W)).
template <int WIN, bool SD, int NB, int LPL, int NP, int DS, int ABL = 0>   // ABL: ablations of scripts/bench_a.hip (0 = product)
__global__ __launch_bounds__(64 * (NP + 1)) void fused_rawbox_h_ps_kernel(FusedA a, float *__restrict__ T,
                                                                          int tyshift, int Dp, float scale,
                                                                          int keep_all)
{
    static_assert(WIN >= 3 && (WIN & 1), "static odd window");
    static_assert(LPL == 1 || LPL == 2 || LPL == 4, "levels per producer lane");
    constexpr int CW = 64 * LPL * NP;                  // levels per workgroup
    constexpr int SV = CW / 64;                        // consecutive levels per storer lane
    static_assert(SV == 1 || SV == 2 || SV == 4, "storer vector width");
    constexpr int pr = WIN / 2, pl = pr - WIN;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_ps[];
    float *buf = (float *)lds_ps;                      // [2][WIN][CW]
    const int W = a.W;
    RowLds<NB> L;
    L.init(lds_ps + (size_t)2 * WIN * CW * sizeof(float), W);
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const bool storer = wib == NP;
    const int y = blockIdx.x;
    const int kc = blockIdx.y * CW;                    // first level of this workgroup
    {
        const size_t ro = (size_t)y * a.pitch;
        for (int i = threadIdx.x; i < W; i += 64 * (NP + 1))
            L.stage(a, ro, i);
    }
    __syncthreads();

    const int dstep = DS ? DS : a.dstep;
    const int D = a.D;
    const int cutoff = a.cutoff, outside = a.outside;
    const size_t xs = (size_t)Dp << tyshift;           // floats between consecutive x
    float *trow = T + slab_row_offset(y, W, Dp, tyshift) + kc;
    const int nblocks = (W - 1 + WIN - 1) / WIN;       // block b = steps x = 1 + WIN*b .. WIN + WIN*b
    // Positions q with 0 <= q + disp < W for every live level of the workgroup: no bounds tests there.
    const int kl = min(kc + CW, D) - 1;
    const int df = a.disp0 + kc * dstep, dl = a.disp0 + kl * dstep;
    const int qa = max(0, -min(df, dl)), qb = min(W, W - max(df, dl));
    auto interior = [&](int xb) { return xb >= qa && xb + WIN - 1 + pr < qb; };

    // ---- producer state (levels k0 .. k0+LPL-1; levels beyond D shadow level D-1, never stored) ----
    const int kq = (wib * 64 + lane) * LPL;            // level inside the workgroup's chunk
    int disp[LPL];
#pragma unroll
    for (int i = 0; i < LPL; i++)
        disp[i] = a.disp0 + min(kc + kq + i, D - 1) * dstep;
    auto cost = [&](int x, int i) -> int {             // MatchLine for reference pixel x, level i of the lane
        int xm = x + disp[i];
        bool out = (unsigned)xm >= (unsigned)W;
        int xc = (int)min((unsigned)xm, (unsigned)(W - 1));
        int e = min(gapv<SD>(ld_ref(L, x), ld_mtc(L, xc)), cutoff);
        return out ? outside : e;
    };
    float acc[LPL];
    int ring[LPL][WIN];
    PixV<NB> mv[LPL];
    bool primed = false;
    typedef typename vecf<LPL>::type VP;
    typedef typename vecf<SV>::type VS;
    if (!storer) {
#pragma unroll
        for (int i = 0; i < LPL; i++)
            acc[i] = 0;
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const int xr = clampi(pl + 1 + j, W);
#pragma unroll
            for (int i = 0; i < LPL; i++) {
                int c = cost(xr, i);
                ring[i][j] = c;
                acc[i] += scale * (float)c;
            }
        }
        if (kc + kq < D) {                             // x = 0 goes straight to the slab
            VP v;
            if constexpr (LPL == 1) v = acc[0];
            else {
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    v[i] = acc[i];
            }
            *(VP *)(trow + kq) = v;
        }
    }
    auto emit = [&](float *slot) {                     // the lane's LPL sums of this step -> block buffer
        VP v;
        if constexpr (LPL == 1) v = acc[0];
        else {
#pragma unroll
            for (int i = 0; i < LPL; i++)
                v[i] = acc[i];
        }
        *(VP *)(slot + kq) = v;
    };
    auto produce = [&](int b) {
        const int xb = 1 + WIN * b;
        float *blk = buf + (size_t)(b & 1) * WIN * CW;
        if (interior(xb)) {
            if (DS != 0 && LPL > 1 && !primed) {       // matching pixels of the previous step
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    mv[i] = ld_mtc(L, clampi(xb - 1 + pr + disp[i], W));
                primed = true;
            }
#pragma unroll
            for (int s = 0; s < WIN; s++) {
                const int x = xb + s;
                const PixV<NB> r = ld_ref(L, x + pr);
                if (DS != 0 && LPL > 1) {
                    if (DS < 0) {                       // level i now sees what level i-1 saw
#pragma unroll
                        for (int i = LPL - 1; i > 0; i--)
                            mv[i] = mv[i - 1];
                        mv[0] = ld_mtc(L, x + pr + disp[0]);
                    } else {
#pragma unroll
                        for (int i = 0; i < LPL - 1; i++)
                            mv[i] = mv[i + 1];
                        mv[LPL - 1] = ld_mtc(L, x + pr + disp[LPL - 1]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < LPL; i++)
                        mv[i] = ld_mtc(L, x + pr + disp[i]);
                }
#pragma unroll
                for (int i = 0; i < LPL; i++) {
                    int c = min(gapv<SD>(r, mv[i]), cutoff);
                    acc[i] = acc[i] + scale * (float)(c - ring[i][s]);
                    ring[i][s] = c;
                }
                emit(blk + s * CW);
            }
        } else {
            primed = false;
#pragma unroll
            for (int s = 0; s < WIN; s++) {
                const int x = xb + s;
                if (x < W) {
                    const int xr = min(x + pr, W - 1);
#pragma unroll
                    for (int i = 0; i < LPL; i++) {
                        int c = cost(xr, i);
                        acc[i] = acc[i] + scale * (float)(c - ring[i][s]);
                        ring[i][s] = c;
                    }
                    emit(blk + s * CW);
                }
            }
        }
    };

    // ---- storer: lane l owns levels kc + SV*l .. of every step ----
    const int ks = lane * SV;
    auto drain = [&](int b) {
        const int xb = 1 + WIN * b;
        const float *blk = buf + (size_t)(b & 1) * WIN * CW;
        VS v[WIN];
#pragma unroll
        for (int s = 0; s < WIN; s++)
            v[s] = *(const VS *)(blk + s * CW + ks);
        if (kc + ks >= D)
            return;
        if ((ABL & 1) && xb + WIN < W)                 // ablation: compute and LDS traffic, (almost) no stores
            return;
        float *t = trow + ks + xs * xb;
        if (keep_all || interior(xb)) {
#pragma unroll
            for (int s = 0; s < WIN; s++) {
                if (xb + s < W)
                    *(VS *)(t + xs * s) = v[s];
            }
        } else {
            // a lane stores its levels when one of them survives PadCosts (pass B loads such a lane whole)
#pragma unroll
            for (int s = 0; s < WIN; s++) {
                const int x = xb + s;
                bool live = false;
#pragma unroll
                for (int i = 0; i < SV; i++)
                    live = live || (kc + ks + i < D && (unsigned)(x + a.disp0 + (kc + ks + i) * dstep) < (unsigned)W);
                if (x < W && live)
                    *(VS *)(t + xs * s) = v[s];
            }
        }
    };

    for (int b = 0; b <= nblocks; b++) {
        if (!storer) {
            if (b < nblocks)
                produce(b);
        } else if (b > 0) {
            drain(b - 1);
        }
        __syncthreads();
    }
}

}  // namespace smk
