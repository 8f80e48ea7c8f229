// smhip_opt.h -- scan-line optimisation (OptSO, StcOptSO.cpp:30-154) on the resident DSI.
//
// Per scan-line: sumcost[x][d] = min_d1 (sumcost[x-1][d1] + (d1 != d) * smooth[x-1]) + cost[x][d],
// then a backtrack from the cheapest level of the last column.  The reference scans all d1 for
// every d (O(D^2) per pixel) and keeps the FIRST strict minimum.  With a Potts term the minimum
// over d1 != d is the same for every d except the overall first minimum, so one step needs only
// the first minimum (m1, i1) of t[d1] = sumcost[x-1][d1] + smooth and the first minimum (m2, i2)
// of the others: O(D) per pixel, and exactly the reference's choice including ties (all sums
// are the same float additions).  One wave per scan-line, NV consecutive levels per lane (so a
// lower lane holds lower levels), DPP wave minima on order-preserving integer images of the
// floats; the transitions go to an int16 volume in the workspace; lane 0 backtracks.
#pragma once

#include "smhip_fused.h"

namespace smk {

// horizontal smoothness weight between pixel x and x+1 (ComputeNCost / ComputeSmoothnessCosts,
// StcOptimize.cpp:61-82, :114): band 1 of m_smooth, 0 in the last column
__global__ void smooth_h_kernel(const uint8_t *ref, size_t pitch, int W, int H, int nb, float smoothness,
                                float grad_thresh, float grad_penalty, float *out)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= W)
        return;
    float s = 0.0f;
    if (x < W - 1) {
        const uint8_t *p = ref + (size_t)y * pitch + (size_t)x * nb;
        float d2 = 0.0f;
        for (int b = 0; b < nb; b++) {
            float d = (float)((int)p[b] - (int)p[b + nb]);
            d2 += d * d;
        }
        d2 /= (float)(nb - (nb > 1));
        float da = sqrtf(d2);
        s = smoothness;
        if (da < grad_thresh)
            s *= grad_penalty;
    }
    out[(size_t)y * W + x] = s;
}

#define SMK_COST_MAX 1000000.0f   // StereoMatcher.h:57

// first minimum (ordered value, level) over the wave of per-lane candidates
__device__ __forceinline__ void wave_first_min(int bo, int bi, int &mo, int &mi)
{
    mo = wave_min_i32(bo);
    mi = wave_min_i32(bo == mo ? bi : 0x7FFFFFFF);
}

template <int NV>
__global__ __launch_bounds__(64) void opt_so_kernel(const float *__restrict__ cost, int W, int H, int D,
                                                    size_t row_elems, const float *__restrict__ smooth,
                                                    int16_t *__restrict__ trans, int32_t *__restrict__ disp,
                                                    float *__restrict__ best)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const float *c = cost + (size_t)y * row_elems;
    int16_t *tr = trans + (size_t)y * W * D;
    const int d0 = lane * NV;
    const float inf = __builtin_inff();
    auto load = [&](int x, float(&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; i++)
            v[i] = (d0 + i < D) ? c[(size_t)x * D + d0 + i] : inf;
    };
    float sum[NV], nxt[NV];
    load(0, sum);                                             // :56-60
#pragma unroll
    for (int i = 0; i < NV; i++)
        if (d0 + i < D)
            tr[d0 + i] = -1;
    if (W > 1)
        load(1, nxt);
    for (int x = 1; x < W; x++) {
        float cur[NV];
#pragma unroll
        for (int i = 0; i < NV; i++)
            cur[i] = nxt[i];
        if (x + 1 < W)
            load(x + 1, nxt);
        const float s = smooth[(size_t)y * W + x - 1];
        // t[d1] = sumcost[x-1][d1] + smooth: overall first minimum, then first minimum of the rest
        int to[NV];
        int bo = 0x7FFFFFFF, bi = 0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            to[i] = (d0 + i < D) ? ordered_bits(sum[i] + s) : 0x7FFFFFFF;
            if (to[i] < bo) {
                bo = to[i];
                bi = d0 + i;
            }
        }
        int m1, i1, m2, i2;
        wave_first_min(bo, bi, m1, i1);
        bo = 0x7FFFFFFF;
        bi = 0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < NV; i++)
            if (d0 + i != i1 && to[i] < bo) {
                bo = to[i];
                bi = d0 + i;
            }
        wave_first_min(bo, bi, m2, i2);
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int d = d0 + i;
            if (d < D) {
                const int oo = (d != i1) ? m1 : m2, oi = (d != i1) ? i1 : i2;
                const int so = ordered_bits(sum[i]);          // staying at d costs no smoothness term
                const bool stay = so < oo || (so == oo && d < oi);
                float bc = stay ? sum[i] : unordered_bits(oo);
                int bd = stay ? d : oi;
                if (!(bc < SMK_COST_MAX)) {                   // nothing beat COST_MAX (:77-78, :95)
                    bc = SMK_COST_MAX;
                    bd = -1;
                }
                sum[i] = bc + cur[i];
                tr[(size_t)x * D + d] = (int16_t)bd;
            }
        }
        // (sum[] of a lane is only replaced after every lane has taken part in both minima)
    }
    // cheapest level of the last column (:125-133), first strict minimum below COST_MAX
    int bo = 0x7FFFFFFF, bi = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        int o = (d0 + i < D) ? ordered_bits(sum[i]) : 0x7FFFFFFF;
        if (o < bo) {
            bo = o;
            bi = d0 + i;
        }
    }
    int mo, mi;
    wave_first_min(bo, bi, mo, mi);
    if (!(unordered_bits(mo) < SMK_COST_MAX))
        mi = 0;
    // backtrack (:144-151); the transitions of this row were written by this wave
    __threadfence();
    if (lane == 0) {
        int d = mi;
        for (int x = W - 1; x >= 0; x--) {
            disp[(size_t)y * W + x] = d;
            if (best)
                best[(size_t)y * W + x] = c[(size_t)x * D + d];
            d = tr[(size_t)x * D + d];
            if (d < 0)
                d = 0;      // x == 0, or a path through COST_MAX (the reference asserts there)
        }
    }
}

}  // namespace smk
