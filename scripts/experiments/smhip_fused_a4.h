// smhip_fused_a4.h -- pass A of the fused path, "four levels per lane" form.
//
// One WAVE owns one image row x 256 consecutive levels (lane l: levels 4l .. 4l+3 of the chunk), so
// every x-step ends in ONE global_store_dwordx4 per lane = 1 KiB contiguous per wave-instruction
// (the one-level-per-lane kernel of smhip_fused.h issues four 256-B stores for the same bytes).
// Work per step shrinks with it: the reference pixel is shared by the four levels, and for
// |dstep| == 1 the matching pixel of level k at step x is the matching pixel of level k -/+ 1 at
// step x-1, so a lane reads ONE new matching pixel per step from LDS and shifts the other three
// through registers (StcRawCosts.cpp:152-190 evaluated for four neighbouring disparities).
// Rows are wave-private: each wave stages its own row in LDS, no exchange between waves.
// Arithmetic and operation order per chain are those of fused_rawbox_h_lds_kernel (bit-identical T).
#pragma once

#include "../../stereomatch_amd/csrc/smhip_fused_b.h"   // vecf

namespace smk {

template <int NB> struct PixV;
template <> struct PixV<1> { int lo, hi; };
template <> struct PixV<4> { uint32_t lo, hi; };

__device__ __forceinline__ PixV<1> ld_ref(const RowLds<1> &L, int x) { int a = L.r[x]; return {a & 0xFF, a >> 8}; }
__device__ __forceinline__ PixV<1> ld_mtc(const RowLds<1> &L, int x) { int b = L.m[x]; return {b & 0xFF, b >> 8}; }
__device__ __forceinline__ PixV<4> ld_ref(const RowLds<4> &L, int x) { return {L.rn[x], L.rx[x]}; }
__device__ __forceinline__ PixV<4> ld_mtc(const RowLds<4> &L, int x) { return {L.mn[x], L.mx[x]}; }

// distance between the reference interval a and the matching interval b, summed over bands
// (StcRawCosts.cpp:168-183)
template <bool SD> __device__ __forceinline__ int gapv(PixV<1> a, PixV<1> b)
{
    int e = max(max(b.lo - a.hi, a.lo - b.hi), 0);
    return SD ? e * e : e;
}
template <bool SD> __device__ __forceinline__ int gapv(PixV<4> a, PixV<4> b)
{
    int sum = 0;
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        int e = max(max((int)((b.lo >> s) & 0xFF) - (int)((a.hi >> s) & 0xFF),
                        (int)((a.lo >> s) & 0xFF) - (int)((b.hi >> s) & 0xFF)), 0);
        sum += SD ? e * e : e;
    }
    return sum;
}

// LPL levels per lane (a wave covers 64 * LPL levels; LPL = 2 for 128-level slabs).
// blockDim.x = 64 * (rows per workgroup); grid (ceil(H / rows per workgroup), Dp / (64 * LPL)).
// DS = dstep (+1 or -1).  Dynamic LDS: rows per workgroup * RowLds<NB>::bytes(W).
template <int WIN, bool SD, int NB, int DS, int LPL, int ABL = 0>   // ABL: ablations of scripts/bench_a.hip (0 = product)
__global__ __launch_bounds__(256) void fused_rawbox_h_a4_kernel(FusedA a, float *__restrict__ T, int tyshift,
                                                                int Dp, float scale, int keep_all)
{
    static_assert(WIN >= 3 && (WIN & 1), "static odd window");
    static_assert(DS == 1 || DS == -1, "unit disparity step");
    static_assert(LPL == 1 || LPL == 2 || LPL == 4, "one, two or four levels per lane");
    constexpr int CW = 64 * LPL;                       // levels per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_a4[];
    const int W = a.W;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int y = blockIdx.x * (blockDim.x >> 6) + wib;
    RowLds<NB> L;
    L.init(lds_a4 + (size_t)wib * RowLds<NB>::bytes(W), W);
    if (y < a.H) {
        const size_t ro = (size_t)y * a.pitch;
        for (int i = lane; i < W; i += 64)
            L.stage(a, ro, i);
    }
    __syncthreads();
    const int kc = blockIdx.y * CW;                    // first level of this wave's chunk
    const int k0 = kc + lane * LPL;
    if (y >= a.H || k0 >= a.D)
        return;
    const int cutoff = a.cutoff, outside = a.outside;
    int disp[LPL];
#pragma unroll
    for (int i = 0; i < LPL; i++)
        disp[i] = a.disp0 + (k0 + i) * DS;
    float *t = T + slab_row_offset(y, W, Dp, tyshift) + k0;
    const size_t xs = (size_t)Dp << tyshift;           // floats between consecutive x
    constexpr int pr = WIN / 2, pl = pr - WIN;

    // MatchLine (StcRawCosts.cpp:152-198) for reference pixel x at level i of this lane
    auto cost = [&](int x, int i) -> int {
        int xm = x + disp[i];
        bool out = (unsigned)xm >= (unsigned)W;
        int xc = (int)min((unsigned)xm, (unsigned)(W - 1));
        int e = min(gapv<SD>(ld_ref(L, x), ld_mtc(L, xc)), cutoff);
        return out ? outside : e;
    };
    // a lane stores its four levels when one of them survives PadCosts (pass B loads such a lane whole)
    auto any_live = [&](int x) -> bool {
        bool v = false;
#pragma unroll
        for (int i = 0; i < LPL; i++)
            v = v || ((unsigned)(x + disp[i]) < (unsigned)W && k0 + i < a.D);
        return v;
    };

    float acc[LPL];
    int ring[LPL][WIN];
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
    auto put = [&](int x) {
        if ((ABL & 1) && x + 2 < W)                    // ablation: compute only
            return;
        typename vecf<LPL>::type v;
        if constexpr (LPL == 1) v = acc[0];
        else
#pragma unroll
        for (int i = 0; i < LPL; i++)
            v[i] = acc[i];
        *(typename vecf<LPL>::type *)(t + xs * x) = v;
    };
    put(0);

    auto general_step = [&](int x, int s) {
        const int xr = min(x + pr, W - 1);
#pragma unroll
        for (int i = 0; i < LPL; i++) {
            int c = cost(xr, i);
            acc[i] = acc[i] + scale * (float)(c - ring[i][s]);
            ring[i][s] = c;
        }
        if (keep_all || any_live(x))
            put(x);
    };

    // Positions q with 0 <= q + disp < W for every level of the wave's chunk: there neither the raw
    // cost nor the store needs a bounds test.
    // (levels beyond D in a partly live lane count: the lane reads their matching pixels too)
    const int kl = min(kc + CW - 1, ((a.D + LPL - 1) / LPL) * LPL - 1);
    const int df = a.disp0 + kc * DS, dl = a.disp0 + kl * DS;
    const int qa = max(0, -min(df, dl)), qb = min(W, W - max(df, dl));
    auto interior = [&](int xb) { return xb >= qa && xb + WIN - 1 + pr < qb; };

    int xb = 1;
    for (; xb + WIN <= W && !interior(xb); xb += WIN) {
#pragma unroll
        for (int s = 0; s < WIN; s++)
            general_step(xb + s, s);
    }
    if (xb + WIN <= W) {
        // mv[i] = matching pixel of level i at the previous step (the slot that is shifted out is not used)
        PixV<NB> mv[LPL];
#pragma unroll
        for (int i = 0; i < LPL; i++)
            mv[i] = ld_mtc(L, clampi(xb - 1 + pr + disp[i], W));
        for (; xb + WIN <= W && interior(xb); xb += WIN) {
#pragma unroll
            for (int s = 0; s < WIN; s++) {
                const int x = xb + s;
                const PixV<NB> r = ld_ref(L, x + pr);
                if (DS < 0) {                           // level i now sees what level i-1 saw
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
#pragma unroll
                for (int i = 0; i < LPL; i++) {
                    int c = min(gapv<SD>(r, mv[i]), cutoff);
                    acc[i] = acc[i] + scale * (float)(c - ring[i][s]);
                    ring[i][s] = c;
                }
                put(x);
            }
        }
    }
    for (; xb + WIN <= W; xb += WIN) {
#pragma unroll
        for (int s = 0; s < WIN; s++)
            general_step(xb + s, s);
    }
#pragma unroll
    for (int s = 0; s < WIN; s++) {
        if (xb + s < W)
            general_step(xb + s, s);
    }
}

}  // namespace smk
