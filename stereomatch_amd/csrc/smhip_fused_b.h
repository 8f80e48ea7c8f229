// smhip_fused_b.h -- pass B of the fused path, "VPL levels per lane" form.
//
// One WAVE owns one pixel column x and a chunk of 64*VPL consecutive disparity levels:
// lane l holds levels chunk*64*VPL + l*VPL .. + VPL-1, so one row of the chunk is ONE
// vector load of VPL floats per lane (VPL = 4: global_load_dwordx4, 1 KiB per wave,
// non-temporal: T is read exactly once).  The lane walks y serially (the vertical running
// sum of BoxFilter.cpp:225-232 is a serial recurrence per (x,d) chain), applies PadCosts,
// takes the first minimum of its VPL levels, and the 64 lanes min-reduce the ordered cost
// with DPP; the lowest lane holding the minimum wins (= lowest d, OptWTA's strict '<').
// No LDS, no barrier: a 256-level pixel is finished inside one wave.
#pragma once

#include "smhip_fused.h"

namespace smk {

template <int VPL> struct vecf { typedef float type __attribute__((ext_vector_type(VPL))); };
template <> struct vecf<1> { typedef float type; };

template <int VPL>
__device__ __forceinline__ void load_row(const float *p, float (&v)[VPL])
{
    typedef typename vecf<VPL>::type V;
    V t = __builtin_nontemporal_load((const V *)p);
    if constexpr (VPL == 1) {
        v[0] = t;
    } else {
#pragma unroll
        for (int i = 0; i < VPL; i++)
            v[i] = t[i];
    }
}

// WIN > 0: static window, blocks of WIN rows, the row leaving the window is the row that
// entered one block earlier (registers).  WIN == 0: generic window, blocks of 8 rows, the
// leaving row is re-read.  blockDim.x = 64 * PPB (PPB pixel columns per block, one per wave).
template <int WIN, int VPL>
__global__ __launch_bounds__(256) void fused_boxv_wta_vpl_kernel(
    const float *__restrict__ T, size_t trow, int W, int H, int Dp, int D, int win, float scale,
    int disp0, int dstep, float outside, int d_first, int x_per_xcd,
    long long *__restrict__ keys, size_t keys_chunk_stride)
{
    constexpr int BLK = WIN ? WIN : 8;
    constexpr int CH = 64 * VPL;
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    // XCD-aware column order: workgroups are dealt round-robin to the 8 XCDs, so give each
    // XCD a contiguous range of columns (its L2 then merges the 8-byte key stores of
    // neighbouring columns into whole lines).  Correctness does not depend on placement.
    const int wpb = blockDim.x >> 6;
    const int blk = wave / wpb, wib = wave - blk * wpb;
    const int x = (blk & 7) * x_per_xcd + (blk >> 3) * wpb + wib;
    const int chunk = blockIdx.y;
    if (x >= W || (blk >> 3) * wpb + wib >= x_per_xcd)
        return;                                        // whole wave exits together
    const int k0 = chunk * CH + lane * VPL;            // first level of this lane
    const size_t j = (size_t)x * Dp + k0;
    const int w = WIN ? WIN : win;
    const int pr = w / 2, pl = pr - w;
    long long *kout = keys + (size_t)chunk * keys_chunk_stride;

    bool padded[VPL], live[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        int k = k0 + i;
        live[i] = k < D;
        int xm = x + disp0 + (k < D ? k : D - 1) * dstep;
        padded[i] = (xm < 0) || (xm >= W);             // PadLine, StcRawCosts.cpp:451-456
    }
    // lanes whose levels are all beyond D read the last valid vector instead (never used)
    const size_t jl = (k0 + VPL <= Dp) ? j : (size_t)x * Dp + (Dp - VPL);

    auto ld = [&](int row, float(&v)[VPL]) { load_row<VPL>(T + (size_t)clampi(row, H) * trow + jl, v); };

    float acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i++)
        acc[i] = 0;
    for (int r = pl + 1; r <= pr; r++) {
        float v[VPL];
        ld(r, v);
#pragma unroll
        for (int i = 0; i < VPL; i++)
            acc[i] += scale * v[i];
    }

    auto step = [&](const float(&od)[VPL], const float(&nw)[VPL], int y) {
        float best = 0;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            acc[i] = acc[i] + scale * (-od[i] + nw[i]);
            float v = padded[i] ? outside : acc[i];
            v = live[i] ? v : __builtin_inff();
            if (i == 0) {
                best = v;
            } else if (v < best) {                      // strict: lowest level wins inside the lane
                best = v;
                bi = i;
            }
        }
        int s = ordered_bits(best);
        int m = wave_min_i32(s);
        unsigned long long eq = __builtin_amdgcn_ballot_w64(s == m);
        int wl = (int)__builtin_ctzll(eq);              // lowest lane with the minimum
        int d = d_first + chunk * CH + wl * VPL + __builtin_amdgcn_readlane(bi, wl);
        if (lane == 0 && y < H)
            kout[(size_t)y * W + x] = make_key(m, d);
    };

    if (WIN) {
        float b0[BLK][VPL], b1[BLK][VPL], b2[BLK][VPL];
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            ld(r + pl, b0[r]);
            ld(r + pr, b1[r]);
        }
#pragma unroll
        for (int i = 0; i < VPL; i++)
            b0[0][i] = b1[0][i];   // y = 0 adds scale*(-a + a) = +0: the first value stays the window sum
        auto block = [&](float(&prv)[BLK][VPL], float(&cur)[BLK][VPL], float(&nxt)[BLK][VPL], int yb) {
#pragma unroll
            for (int r = 0; r < BLK; r++)
                ld(yb + BLK + r + pr, nxt[r]);
#pragma unroll
            for (int r = 0; r < BLK; r++)
                step(prv[r], cur[r], yb + r);
        };
        for (int yb = 0; yb < H; yb += 3 * BLK) {
            block(b0, b1, b2, yb);
            if (yb + BLK < H)
                block(b1, b2, b0, yb + BLK);
            if (yb + 2 * BLK < H)
                block(b2, b0, b1, yb + 2 * BLK);
        }
    } else {
        float nw[2][BLK][VPL], od[2][BLK][VPL];
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            ld(r + pr, nw[0][r]);
            ld(r + pl, od[0][r]);
        }
#pragma unroll
        for (int i = 0; i < VPL; i++)
            od[0][0][i] = nw[0][0][i];
        for (int y0 = 0; y0 < H; y0 += 2 * BLK) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int yb = y0 + half * BLK;
                if (yb < H) {
#pragma unroll
                    for (int r = 0; r < BLK; r++) {
                        ld(yb + BLK + r + pr, nw[half ^ 1][r]);
                        ld(yb + BLK + r + pl, od[half ^ 1][r]);
                    }
#pragma unroll
                    for (int r = 0; r < BLK; r++)
                        step(od[half][r], nw[half][r], yb + r);
                }
            }
        }
    }
}

}  // namespace smk
