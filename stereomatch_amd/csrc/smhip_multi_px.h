// smhip_multi_px.h -- K iterations of the separable binomial 14641 (Convolve.cpp:86-254, ConvolveSeparable called
// once per aggr_iter at StcAggregate.cpp:280-289) in one pass, "pixels on lanes".
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

__device__ __forceinline__ float tap5_pow2(float s0, float s1, float s2, float s3, float s4)
{
    float acc = 0.0625f * s0;
    acc = __builtin_fmaf(0.25f, s1, acc);
    acc += 0.375f * s2;
    acc = __builtin_fmaf(0.25f, s3, acc);
    acc = __builtin_fmaf(0.0625f, s4, acc);
    return acc;
}

template <int K, int LV>
__global__ __launch_bounds__(LV * 32) void binomial_px_kernel(const float *__restrict__ src,
                                                              float *__restrict__ dst, int W, int D, int H,
                                                              size_t row_elems, int seglen, int force_slow)
{
    typedef typename fvec<4>::type vec;
    constexpr int HL = px_halo(K), TXI = px_txi(K), QP = LV / 4;
    __shared__ __attribute__((aligned(16))) float tin[2][LV][PX_STRIDE];
    __shared__ __attribute__((aligned(16))) float tout[2][LV][PX_STRIDE];
    __shared__ int guard;
    const int t = threadIdx.x;
    const int X0 = (int)blockIdx.x * TXI - HL, k0 = (int)blockIdx.y * LV;
    const int y0 = (int)blockIdx.z * seglen, y1 = min(H, y0 + seglen);
    // compute layout: level l of the tile, pixels X0 + 4 g .. + 3
    const int l = t >> 5, g = t & 31;
    const int xb = X0 + 4 * g;
    const bool left_edge = xb == 0, right_edge = xb + 3 == W - 1;
    // DSI layout: pixel lp of the tile, levels k0 + 4 lq .. + 3
    const int lp = t / QP, lq = t % QP;
    const bool lact = k0 + 4 * lq < D;
    const size_t lcol = (size_t)clampi(X0 + lp, W) * D + (size_t)min(k0 + 4 * lq, D - 4);
    const bool writer = lact && lp >= HL && lp < PX_TXT - HL && X0 + lp < W;
    if (t == 0)
        guard = force_slow;
    const vec zero = {0.0f, 0.0f, 0.0f, 0.0f};
    vec ring[K][5];                             // level j: horizontally filtered rows n-4 .. n, slot = row mod 5
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int i = 0; i < 5; i++)
            ring[j][i] = zero;
    // step R takes input row R and finishes output row R - 2 K, which leaves through LDS at step R + 1
    const int Rlo = y0 - 2 * K, Rin = y1 - 1 + 2 * K, Rhi = Rin + 1;
    vec nxt = zero, res = zero;
    if (lact && Rlo >= 0 && Rlo < H)
        nxt = vload<4>(src + (size_t)Rlo * row_elems + lcol);
    bool slow = false;
    __syncthreads();
    auto tap = [&](vec s0, vec s1, vec s2, vec s3, vec s4) -> vec {
        if (slow)
            return vtap5<4>(s0, s1, s2, s3, s4);
        return vtap5_pow2<4>(s0, s1, s2, s3, s4);
    };
    // horizontal tap sums of the thread's four pixels; FAST: no run-time choice of the arithmetic
    auto htap = [&](vec v, auto fast) -> vec {
        float l2 = lane_from_below(v[2], v[0]), l3 = lane_from_below(v[3], v[0]);
        float r0 = lane_from_above(v[0], v[3]), r1 = lane_from_above(v[1], v[3]);
        if (left_edge)
            l2 = l3 = v[0];
        if (right_edge)
            r0 = r1 = v[3];
        // scalar tap sums: the four windows overlap by one register each, which packed instructions (aligned
        // register pairs) could only take after moving the operands
        const float w[8] = {l2, l3, v[0], v[1], v[2], v[3], r0, r1};
        vec h;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (decltype(fast)::value || !slow)
                h[e] = tap5_pow2(w[e], w[e + 1], w[e + 2], w[e + 3], w[e + 4]);
            else
                h[e] = tap5(w[e], w[e + 1], w[e + 2], w[e + 3], w[e + 4]);
        }
        return h;
    };
    auto step = [&](auto rr, const int R) {
        constexpr int RR = decltype(rr)::value;
        const int p = (RR & 1) ^ (((R - Rlo) / 5) & 1);     // parity of R - Rlo
        if (!slow && tap_guard_bad(nxt))
            guard = 1;
#pragma unroll
        for (int e = 0; e < 4; e++)
            tin[p][4 * lq + e][lp] = nxt[e];
        if (lact && R + 1 <= Rin && R + 1 >= 0 && R + 1 < H)
            nxt = vload<4>(src + (size_t)(R + 1) * row_elems + lcol);
        *(vec *)&tout[p][l][4 * g] = res;       // the row finished at the step before
        __syncthreads();
        if (!slow)                              // the guard of this row's inputs was set before the barrier
            slow = __builtin_amdgcn_readfirstlane(guard) != 0;
        const int qs = R - 1 - 2 * K;           // row in tout
        if (writer && qs >= y0 && qs < y1) {
            vec o;
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = tout[p][4 * lq + e][lp];
            vstore<4>(dst + (size_t)qs * row_elems + lcol, o);
        }
        vec val = *(const vec *)&tin[p][l][4 * g];          // level 0, row R
        if (!slow && R - 2 * (K - 1) >= 1 && R <= H - 1) {
            // every level's row lies inside the image and the exact-product taps apply: one basic block
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int s = (RR + 10 * K - 2 * j) % 5;
                ring[j][s] = htap(val, icst<1>{});
                val = vtap5_pow2<4>(ring[j][(s + 1) % 5], ring[j][(s + 2) % 5], ring[j][(s + 3) % 5],
                                    ring[j][(s + 4) % 5], ring[j][s]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int n = R - 2 * j;        // row of level j arriving now
                const int s = (RR + 10 * K - 2 * j) % 5;    // its ring slot; rows n-4 .. n-1 sit in s+1 .. s+4 (mod 5)
                if (n <= H - 1) {               // rows below the image replicate the last row
                    ring[j][s] = htap(val, icst<0>{});
                    if (n == 0)                 // rows above the image replicate row 0
                        ring[j][(s + 3) % 5] = ring[j][(s + 4) % 5] = ring[j][s];
                } else {
                    ring[j][s] = ring[j][(s + 4) % 5];
                }
                val = tap(ring[j][(s + 1) % 5], ring[j][(s + 2) % 5], ring[j][(s + 3) % 5], ring[j][(s + 4) % 5],
                          ring[j][s]);          // level j+1, row n-2
            }
        }
        res = val;                              // output row R - 2 K
    };
    for (int Rb = Rlo; Rb <= Rhi; Rb += 5) {
        step(icst<0>{}, Rb);
        if (Rb + 1 <= Rhi) step(icst<1>{}, Rb + 1);
        if (Rb + 2 <= Rhi) step(icst<2>{}, Rb + 2);
        if (Rb + 3 <= Rhi) step(icst<3>{}, Rb + 3);
        if (Rb + 4 <= Rhi) step(icst<4>{}, Rb + 4);
    }
}

}  // namespace smk
