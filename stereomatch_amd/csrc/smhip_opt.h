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

// the same when lane order is level order (a lower lane holds lower levels, bi = the lane's own
// first minimum): the owner is the first lane holding the minimum -- a ballot instead of a second
// DPP reduction
__device__ __forceinline__ void wave_first_min_ordered(int bo, int bi, int &mo, int &mi)
{
    mo = wave_min_i32(bo);
    const unsigned long long owners = __ballot(bo == mo);
    mi = __builtin_amdgcn_readlane(bi, __ffsll((long long)owners) - 1);
}

// m_cost at the chosen level (the data term of final_energy reads it): one thread per pixel
__global__ void gather_best_kernel(const float *__restrict__ cost, int W, int H, int D, size_t row_elems,
                                   const int32_t *__restrict__ disp, float *__restrict__ best)
{
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)W * H)
        return;
    int y = (int)(p / W), x = (int)(p - (long)y * W);
    int d = min(max(disp[p], 0), D - 1);
    best[p] = cost[(size_t)y * row_elems + (size_t)x * D + d];
}

// One wave per scan-line.  Both sweeps are serial in x, so memory latency is taken out of the
// chain: DSI columns (forward) and transition columns (backtrack) are fetched a block of B
// columns ahead -- neither address depends on the running state -- the smoothness weights are
// held 64 at a time in a lane register, and the backtrack picks trans[x][d] out of the
// prefetched column with a readlane instead of a dependent load.
template <int NV>
__global__ __launch_bounds__(64) void opt_so_kernel(const float *__restrict__ cost, int W, int H, int D,
                                                    size_t row_elems, const float *__restrict__ smooth,
                                                    int16_t *__restrict__ trans, int32_t *__restrict__ disp)
{
    constexpr int B = (32 / NV) < 8 ? (32 / NV) : 8;          // columns per prefetch block
    const int y = blockIdx.x, lane = threadIdx.x;
    const float *c = cost + (size_t)y * row_elems;
    int16_t *tr = trans + (size_t)y * W * D;
    const float *sm = smooth + (size_t)y * W;
    const int d0 = lane * NV;
    const float inf = __builtin_inff();
    auto load = [&](int x, float(&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; i++)
            v[i] = (x < W && d0 + i < D) ? c[(size_t)x * D + d0 + i] : inf;
    };
    float sum[NV];
    load(0, sum);                                             // :56-60
#pragma unroll
    for (int i = 0; i < NV; i++)
        if (d0 + i < D)
            tr[d0 + i] = -1;
    // a lane's NV transitions are one aligned vector when D is a multiple of NV
    typedef short tvec __attribute__((ext_vector_type(NV)));
    const bool packed = NV > 1 && D % NV == 0;
    auto tstore = [&](int x, const short(&v)[NV]) {
        if (packed) {
            if (d0 < D) {
                tvec o;
#pragma unroll
                for (int i = 0; i < NV; i++)
                    o[i] = v[i];
                *(tvec *)(tr + (size_t)x * D + d0) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++)
                if (d0 + i < D)
                    tr[(size_t)x * D + d0 + i] = v[i];
        }
    };
    float cb[2][B][NV];
#pragma unroll
    for (int j = 0; j < B; j++)
        load(1 + j, cb[0][j]);
    float sv = 0.0f;                                          // smooth[sbase + lane]
    int sbase = -64;
    auto step = [&](int x, const float(&cur)[NV]) {
        if (((x - 1) & ~63) != sbase) {
            sbase = (x - 1) & ~63;
            sv = (sbase + lane < W) ? sm[sbase + lane] : 0.0f;
        }
        const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), (x - 1) & 63));
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
        short tv[NV];
        wave_first_min_ordered(bo, bi, m1, i1);
        bo = 0x7FFFFFFF;
        bi = 0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < NV; i++)
            if (d0 + i != i1 && to[i] < bo) {
                bo = to[i];
                bi = d0 + i;
            }
        wave_first_min_ordered(bo, bi, m2, i2);
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
                tv[i] = (short)bd;
            } else {
                tv[i] = 0;
            }
        }
        tstore(x, tv);
    };
    for (int xb = 1; xb < W; xb += 2 * B) {
#pragma unroll
        for (int j = 0; j < B; j++)
            load(xb + B + j, cb[1][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb + j < W)
                step(xb + j, cb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            load(xb + 2 * B + j, cb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb + B + j < W)
                step(xb + B + j, cb[1][j]);
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
    wave_first_min_ordered(bo, bi, mo, mi);
    if (!(unordered_bits(mo) < SMK_COST_MAX))
        mi = 0;
    // backtrack (:144-151): the transitions of this row were written by this wave
    __threadfence();
    auto tload = [&](int x, int(&v)[NV]) {
        if (packed) {
            tvec o = {};
            if (x >= 0 && d0 < D)
                o = *(const tvec *)(tr + (size_t)x * D + d0);
#pragma unroll
            for (int i = 0; i < NV; i++)
                v[i] = o[i];
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++)
                v[i] = (x >= 0 && d0 + i < D) ? (int)tr[(size_t)x * D + d0 + i] : 0;
        }
    };
    int tb[2][B][NV];
#pragma unroll
    for (int j = 0; j < B; j++)
        tload(W - 1 - j, tb[0][j]);
    int d = mi;                                               // wave-uniform
    int dl = 0;                                               // lane (x & 63) keeps the level of pixel x
    auto back = [&](int x, const int(&t)[NV]) {
        dl = (lane == (x & 63)) ? d : dl;
        if ((x & 63) == 0) {
            if (x + lane < W)
                disp[(size_t)y * W + x + lane] = dl;
        }
        const int src = d / NV, sub = d - src * NV;
        int nd = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            int v = __builtin_amdgcn_readlane(t[i], src);
            nd = (i == sub) ? v : nd;
        }
        d = nd < 0 ? 0 : nd;   // x == 0, or a path through COST_MAX (the reference asserts there)
    };
    for (int xb = W - 1; xb >= 0; xb -= 2 * B) {
#pragma unroll
        for (int j = 0; j < B; j++)
            tload(xb - B - j, tb[1][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb - j >= 0)
                back(xb - j, tb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            tload(xb - 2 * B - j, tb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb - B - j >= 0)
                back(xb - B - j, tb[1][j]);
    }
}

}  // namespace smk
