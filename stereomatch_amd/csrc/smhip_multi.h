// smhip_multi.h -- K aggregation iterations in ONE pass over the DSI (temporal blocking), for
// the iterated stencils of exp4_aggr.txt: diffusion / membrane (StcDiffusion.cpp:28-137) and the
// separable binomial 14641 (Convolve.cpp:86-254).
//
// Workgroup = 64 pixels x 32 levels (8 lanes of dwordx4 per pixel), walking a segment of rows.
// Every thread owns one pixel column of 4 levels: it loads its own input rows, keeps the
// vertical neighbourhood of every iteration level in registers, and gets the horizontal
// neighbours of each level from the adjacent threads through LDS.  Level j+1 of a row is
// computed as soon as level j of the rows it needs exists, so one pass reads the volume once,
// writes it once and applies K iterations.  The outermost HALO*K pixels of a tile only feed
// their neighbours (tiles overlap by 2*HALO*K pixels) and a segment re-computes HALO*K rows
// above and below.  Arithmetic per element and per iteration is exactly that of the
// one-iteration kernels, so results are bit-identical to K separate passes.
// HBM traffic per pass: (64 / (64 - 2*HALO*K)) reads + 1 write per voxel (+1 read of cost0
// for the membrane model), i.e. about (8..9.3)/K bytes per voxel and iteration.
#pragma once

#include "smhip_stencil.h"

namespace smk {

constexpr int MULTI_TXT = 64;   // pixels (threads / 8) per tile, halo included

template <int K, bool MEMBRANE>
__global__ __launch_bounds__(MULTI_TXT * 8) void diffusion_multi_kernel(
    const float *__restrict__ src, float *__restrict__ dst, const float *__restrict__ cost0, int W,
    int D, int H, size_t row_elems, int seglen, float lambda, float a, float lb)
{
    typedef typename fvec<4>::type vec;
    constexpr int TXT = MULTI_TXT, TXI = TXT - 2 * K;
    constexpr int NBUF = K == 1 ? 2 : K;       // one exchange buffer per level (two when K == 1)
    __shared__ vec ex[NBUF][(TXT + 2) * 8];
    const int t = threadIdx.x, px = t >> 3, lq = t & 7;
    const int x = (int)blockIdx.x * TXI - K + px, k = (int)blockIdx.y * 32 + lq * 4;
    const int y0 = (int)blockIdx.z * seglen, y1 = min(H, y0 + seglen);
    const bool alive = x >= 0 && x < W && k < D;
    const bool has_l = x > 0, has_r = x < W - 1;
    const bool writer = alive && px >= K && px < TXT - K;
    const size_t col = (size_t)min(max(x, 0), W - 1) * D + (size_t)min(k, D - 4);
    const vec zero = {0.0f, 0.0f, 0.0f, 0.0f};
    vec s[K][3];                                // level j: rows (q-1, q, q+1) around its current row
    vec c0d[MEMBRANE ? K + 1 : 1];              // cost0 rows R, R-1, ..., R-K
#pragma unroll
    for (int j = 0; j < K; j++)
        s[j][0] = s[j][1] = s[j][2] = zero;
#pragma unroll
    for (int j = 0; j < (MEMBRANE ? K + 1 : 1); j++)
        c0d[j] = zero;
    const int Rlo = y0 - K, Rhi = y1 - 1 + K;
    vec nxt = zero, nxt0 = zero;
    if (alive && Rlo >= 0) {
        nxt = vload<4>(src + (size_t)Rlo * row_elems + col);
        if (MEMBRANE)
            nxt0 = vload<4>(cost0 + (size_t)Rlo * row_elems + col);
    }
    int flip = 0;
    for (int R = Rlo; R <= Rhi; R++) {
        s[0][0] = s[0][1];
        s[0][1] = s[0][2];
        s[0][2] = nxt;
        if (MEMBRANE) {
#pragma unroll
            for (int j = K; j > 0; j--)
                c0d[j] = c0d[j - 1];
            c0d[0] = nxt0;
        }
        if (alive && R + 1 <= Rhi && R + 1 >= 0 && R + 1 < H) {
            nxt = vload<4>(src + (size_t)(R + 1) * row_elems + col);
            if (MEMBRANE)
                nxt0 = vload<4>(cost0 + (size_t)(R + 1) * row_elems + col);
        }
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int q = R - 1 - j;            // current row of level j = row of level j+1 made now
            const vec cur = s[j][1];
            vec *e = ex[K == 1 ? flip : j];
            e[(px + 1) * 8 + lq] = cur;
            __syncthreads();
            const vec l = has_l ? e[px * 8 + lq] : cur;
            const vec r = has_r ? e[(px + 2) * 8 + lq] : cur;
            const vec up = (q > 0) ? s[j][0] : cur;
            const vec dn = (q < H - 1) ? s[j][2] : cur;
            vec v = a * cur + lambda * (((l + r) + up) + dn);
            if (MEMBRANE)
                v += lb * c0d[1 + j];
            const int jn = (j + 1 < K) ? j + 1 : 0;   // (index kept in range for the last level)
            if (j + 1 < K) {
                s[jn][0] = s[jn][1];
                s[jn][1] = s[jn][2];
                s[jn][2] = v;
            } else if (writer && q >= y0 && q < y1) {
                vstore<4>(dst + (size_t)q * row_elems + col, v);
            }
        }
        flip ^= 1;
    }
}

// K iterations of the separable binomial: per level, horizontal taps through LDS, vertical taps
// over a register ring of the five horizontally filtered rows; replicate border per iteration
// (TrimIndex clamp, Convolve.cpp:42-43).
//
// Arithmetic.  The reference's tap sum is ((((0 + k0 s0) + k1 s1) + k2 s2) + k3 s3) + k4 s4 with every product
// rounded before it is added (Convolve.cpp:100-107).  k0 = k4 = 2^-4 and k1 = k3 = 2^-2: their products are exact
// unless they fall into the subnormal range, so fma(k, s, acc) rounds the very same real number as
// (k * s) + acc does -- one instruction instead of two, same bits (the fma is written out; -ffp-contract=off still
// holds for everything else), and 0 + k0 s0 is the bare product when s0 carries no sign bit.  10 -> 6 VALU
// instructions per tap sum; 0.375 keeps its multiply and add.  The two conditions are checked, not assumed: every
// input a workgroup loads is tested (tap_guard_bad: sign bit set, or 0 < s < 2^-80), and from the first row that
// holds such a value on the workgroup computes the plain ten-operation sum (`slow`, sticky).  Why that suffices:
// a tap sum of non-negative terms is never below its smallest non-zero product (rounding is monotone), so one
// application shrinks the smallest non-zero value by at most 2^-4 and the 2 K <= 8 applications of a pass by 2^-32:
// with inputs of zero or >= 2^-80 every product stays >= 2^-116, far above the subnormal range (2^-126).  Integer
// raw costs reach 2^-80 only after ten iterations on an isolated non-zero cost in a field of exact zeros.
constexpr unsigned TAP_GUARD_THR = (127u - 80u) << 23;      // bits of 2^-80f

__device__ __forceinline__ bool tap_guard_bad(typename fvec<4>::type v)
{
    const unsigned b0 = __float_as_uint(v[0]), b1 = __float_as_uint(v[1]), b2 = __float_as_uint(v[2]),
                   b3 = __float_as_uint(v[3]);
    const unsigned lo = min(min(b0 - 1u, b1 - 1u), min(b2 - 1u, b3 - 1u));    // 0 -> 0xFFFFFFFF: zeros pass
    return lo < TAP_GUARD_THR - 1u || (int)(b0 | b1 | b2 | b3) < 0;
}

template <int V>
__device__ __forceinline__ typename fvec<V>::type vtap5_pow2(typename fvec<V>::type s0, typename fvec<V>::type s1,
                                                             typename fvec<V>::type s2, typename fvec<V>::type s3,
                                                             typename fvec<V>::type s4)
{
    typedef typename fvec<V>::type vec;
    const vec k0 = 0.0625f, k1 = 0.25f;
    vec acc = k0 * s0;
    acc = __builtin_elementwise_fma(k1, s1, acc);
    acc += 0.375f * s2;
    acc = __builtin_elementwise_fma(k1, s3, acc);
    acc = __builtin_elementwise_fma(k0, s4, acc);
    return acc;
}

template <int N> struct icst { static constexpr int value = N; };

// TXT = pixels per tile (threads / 8), halo included.  The rows of a segment are taken in groups of five so
// that the ring slot of every (level, row) is static: no register moves to shift the rings.
template <int K, int TXT>
__global__ __launch_bounds__(TXT * 8) void binomial_multi_kernel(const float *__restrict__ src,
                                                                 float *__restrict__ dst, int W, int D,
                                                                 int H, size_t row_elems, int seglen, int force_slow)
{
    typedef typename fvec<4>::type vec;
    constexpr int TXI = TXT - 4 * K;
    __shared__ vec ex[K][TXT * 8];
    __shared__ int guard;
    const int t = threadIdx.x, px = t >> 3, lq = t & 7;
    const int x = (int)blockIdx.x * TXI - 2 * K + px, k = (int)blockIdx.y * 32 + lq * 4;
    const int y0 = (int)blockIdx.z * seglen, y1 = min(H, y0 + seglen);
    const bool alive = x >= 0 && x < W && k < D;
    const bool writer = alive && px >= 2 * K && px < TXT - 2 * K;
    const size_t col = (size_t)min(max(x, 0), W - 1) * D + (size_t)min(k, D - 4);
    // LDS slots of the horizontal neighbours x-2, x-1, x+1, x+2 (clamped to the image, then to the tile)
    int nb[4];
    {
        const int xc = min(max(x, 0), W - 1);
        const int o[4] = {-2, -1, 1, 2};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int p = px + (clampi(xc + o[i], W) - xc);
            nb[i] = min(max(p, 0), TXT - 1) * 8 + lq;
        }
    }
    if (t == 0)
        guard = force_slow;
    const vec zero = {0.0f, 0.0f, 0.0f, 0.0f};
    vec ring[K][5];                             // level j: horizontally filtered rows n-4 .. n, slot = row mod 5
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int i = 0; i < 5; i++)
            ring[j][i] = zero;
    const int Rlo = y0 - 2 * K, Rhi = y1 - 1 + 2 * K;
    vec nxt = zero;
    if (alive && Rlo >= 0 && Rlo < H)
        nxt = vload<4>(src + (size_t)Rlo * row_elems + col);
    bool slow = false;
    __syncthreads();
    auto tap = [&](vec s0, vec s1, vec s2, vec s3, vec s4) -> vec {
        if (slow)
            return vtap5<4>(s0, s1, s2, s3, s4);
        return vtap5_pow2<4>(s0, s1, s2, s3, s4);
    };
    // one row R = Rlo + 5 g + RR
    auto step = [&](auto rr, const int R) {
        constexpr int RR = decltype(rr)::value;
        vec val = nxt;                          // level 0, row R
        if (!slow && tap_guard_bad(val))
            guard = 1;
        if (alive && R + 1 <= Rhi && R + 1 >= 0 && R + 1 < H)
            nxt = vload<4>(src + (size_t)(R + 1) * row_elems + col);
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int n = R - 2 * j;            // row of level j arriving now
            constexpr int JJ = 2 * K;           // (keeps the slot arithmetic non-negative)
            const int s = (RR + 5 * JJ - 2 * j) % 5;    // its ring slot; rows n-4 .. n-1 sit in s+1 .. s+4 (mod 5)
            if (n <= H - 1) {                   // rows below the image replicate the last row
                vec *e = ex[j];
                e[px * 8 + lq] = val;
                __syncthreads();
                if (j == 0 && !slow)            // the guard of this row's inputs was set before the barrier
                    slow = __builtin_amdgcn_readfirstlane(guard) != 0;
                ring[j][s] = tap(e[nb[0]], e[nb[1]], val, e[nb[2]], e[nb[3]]);
                if (n == 0)                     // rows above the image replicate row 0
                    ring[j][(s + 3) % 5] = ring[j][(s + 4) % 5] = ring[j][s];
            } else {
                ring[j][s] = ring[j][(s + 4) % 5];
            }
            val = tap(ring[j][(s + 1) % 5], ring[j][(s + 2) % 5], ring[j][(s + 3) % 5], ring[j][(s + 4) % 5],
                      ring[j][s]);              // level j+1, row n-2
        }
        const int q = R - 2 * K;
        if (writer && q >= y0 && q < y1)
            vstore<4>(dst + (size_t)q * row_elems + col, val);
        __syncthreads();                        // every level's buffer is free again
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
