// smhip_stencil.h -- "column walk" kernels for the per-plane stencils on a resident DSI:
// diffusion / membrane (StcDiffusion.cpp:28-137), separable binomial 14641
// (Convolve.cpp:86-254) and the separable min filter (MinFilter.cpp:122-202).
//
// Common shape: a thread owns V consecutive floats (V = 4: one dwordx4) of a row -- i.e. V
// levels of one pixel, levels on lanes -- and walks a segment of rows serially, keeping the
// vertical neighbourhood in registers.  Every element is therefore read once from HBM as a
// "centre"; the horizontal neighbours (x +- 1, x +- 2, ... at the same level, D floats
// away) are the centres of nearby waves and come from L1/L2.  One kernel per application,
// source and destination distinct: HBM traffic = read once + write once = 8 bytes per voxel.
// All arithmetic is per element and keeps the reference's operation order.
#pragma once

#include "smhip_kernels.h"

namespace smk {

template <int V> struct fvec { typedef float type __attribute__((ext_vector_type(V))); };
template <> struct fvec<1> { typedef float type; };

template <int V>
__device__ __forceinline__ typename fvec<V>::type vload(const float *p)
{
    return *(const typename fvec<V>::type *)p;
}
template <int V>
__device__ __forceinline__ void vstore(float *p, typename fvec<V>::type v)
{
    *(typename fvec<V>::type *)p = v;
}

// ---- diffusion / membrane --------------------------------------------------------------
// o = a*c + lambda*(((l + r) + u) + d), a missing neighbour replaced by c in its slot
// (StcDiffusion.cpp:84-99, :112-118); membrane adds (lambda*beta)*cost0 afterwards (:122-136).
template <int V, bool MEMBRANE>
__global__ __launch_bounds__(256) void diffusion_walk_kernel(
    const float *__restrict__ src, float *__restrict__ dst, const float *__restrict__ cost0,
    int WD, int D, int H, size_t row_elems, int seglen, float lambda, float a, float lb)
{
    typedef typename fvec<V>::type vec;
    const long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (j >= WD)
        return;
    const int y0 = blockIdx.y * seglen, y1 = min(H, y0 + seglen);
    const long lf = (j - D >= 0) ? -(long)D : 0;
    const long rt = (j + D < WD) ? (long)D : 0;
    const float *s = src + (size_t)y0 * row_elems + j;
    float *o = dst + (size_t)y0 * row_elems + j;
    const float *c0 = MEMBRANE ? cost0 + (size_t)y0 * row_elems + j : nullptr;
    vec cur = vload<V>(s);
    vec up = (y0 > 0) ? vload<V>(s - row_elems) : cur;
    for (int y = y0; y < y1; y++) {
        vec dn = (y < H - 1) ? vload<V>(s + row_elems) : cur;
        vec l = vload<V>(s + lf), r = vload<V>(s + rt);
        vec v = a * cur + lambda * (((l + r) + up) + dn);
        if (MEMBRANE) {
            v += lb * vload<V>(c0);
            c0 += row_elems;
        }
        vstore<V>(o, v);
        up = cur;
        cur = dn;
        s += row_elems;
        o += row_elems;
    }
}

// ---- separable binomial 14641 ------------------------------------------------------------
// horizontal: t = ((((0 + k0*s[x-2]) + k1*s[x-1]) + k2*s[x]) + k3*s[x+1]) + k4*s[x+2], replicate
// border; vertical: the same over t, top row first (Convolve.cpp:96-110, :224-225).
template <int V>
__device__ __forceinline__ typename fvec<V>::type vtap5(typename fvec<V>::type s0, typename fvec<V>::type s1,
                                                        typename fvec<V>::type s2, typename fvec<V>::type s3,
                                                        typename fvec<V>::type s4)
{
    typename fvec<V>::type acc = 0.0f;
    acc += 0.0625f * s0;
    acc += 0.25f * s1;
    acc += 0.375f * s2;
    acc += 0.25f * s3;
    acc += 0.0625f * s4;
    return acc;
}

template <int V>
__global__ __launch_bounds__(256) void binomial_walk_kernel(const float *__restrict__ src,
                                                            float *__restrict__ dst, int W, int D, int H,
                                                            size_t row_elems, int seglen)
{
    typedef typename fvec<V>::type vec;
    const long WD = (long)W * D;
    const long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (j >= WD)
        return;
    const int x = (int)(j / D);
    const long o0 = (long)(clampi(x - 2, W) - x) * D, o1 = (long)(clampi(x - 1, W) - x) * D;
    const long o3 = (long)(clampi(x + 1, W) - x) * D, o4 = (long)(clampi(x + 2, W) - x) * D;
    auto hrow = [&](int yy) -> vec {
        const float *r = src + (size_t)clampi(yy, H) * row_elems + j;
        return vtap5<V>(vload<V>(r + o0), vload<V>(r + o1), vload<V>(r), vload<V>(r + o3), vload<V>(r + o4));
    };
    const int y0 = blockIdx.y * seglen, y1 = min(H, y0 + seglen);
    vec h0 = hrow(y0 - 2), h1 = hrow(y0 - 1), h2 = hrow(y0), h3 = hrow(y0 + 1);
    float *o = dst + (size_t)y0 * row_elems + j;
    for (int y = y0; y < y1; y++) {
        vec h4 = hrow(y + 2);
        vstore<V>(o, vtap5<V>(h0, h1, h2, h3, h4));
        h0 = h1; h1 = h2; h2 = h3; h3 = h4;
        o += row_elems;
    }
}

// ---- separable min filter -----------------------------------------------------------------
// o[y][x] = min over rows y+pl+1..y+pr of (min over x+pl+1..x+pr), replicate border
// (MinFilter.cpp:139-140, :49-52); min is exact in any order.  WIN is the (square) window.
template <int V>
__device__ __forceinline__ typename fvec<V>::type vmin(typename fvec<V>::type a, typename fvec<V>::type b)
{
    if constexpr (V == 1) {
        return b < a ? b : a;
    } else {
        typename fvec<V>::type r;
#pragma unroll
        for (int i = 0; i < V; i++)
            r[i] = b[i] < a[i] ? b[i] : a[i];
        return r;
    }
}

template <int V, int WIN>
__global__ __launch_bounds__(256) void minfilter_walk_kernel(const float *__restrict__ src,
                                                             float *__restrict__ dst, int W, int D, int H,
                                                             size_t row_elems, int seglen)
{
    typedef typename fvec<V>::type vec;
    constexpr int PR = WIN / 2, PL = PR - WIN;
    const long WD = (long)W * D;
    const long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (j >= WD)
        return;
    const int x = (int)(j / D);
    long off[WIN];
#pragma unroll
    for (int k = 0; k < WIN; k++)
        off[k] = (long)(clampi(x + PL + 1 + k, W) - x) * D;
    auto hrow = [&](int yy) -> vec {
        const float *r = src + (size_t)clampi(yy, H) * row_elems + j;
        vec m = vload<V>(r + off[0]);
#pragma unroll
        for (int k = 1; k < WIN; k++)
            m = vmin<V>(m, vload<V>(r + off[k]));
        return m;
    };
    const int y0 = blockIdx.y * seglen, y1 = min(H, y0 + seglen);
    vec ring[WIN];                      // ring[k] = horizontal min of row y + PL + 1 + k
#pragma unroll
    for (int k = 0; k < WIN - 1; k++)
        ring[k + 1] = hrow(y0 + PL + 1 + k);
    float *o = dst + (size_t)y0 * row_elems + j;
    for (int y = y0; y < y1; y++) {
#pragma unroll
        for (int k = 0; k < WIN - 1; k++)
            ring[k] = ring[k + 1];
        ring[WIN - 1] = hrow(y + PR);
        vec m = ring[0];
#pragma unroll
        for (int k = 1; k < WIN; k++)
            m = vmin<V>(m, ring[k]);
        vstore<V>(o, m);
        o += row_elems;
    }
}

// ---- LDS-tiled separable stencils ----------------------------------------------------------
// Workgroup = TX pixels x 32 levels (8 lanes of dwordx4 per pixel), walking a segment of rows.
// Each row of the tile plus its horizontal halo goes through LDS once (double buffered, one
// barrier per row): a thread takes its WIN horizontal neighbours from LDS instead of WIN global
// loads, and keeps the WIN horizontally-filtered rows of the vertical window in registers.
// HBM traffic: (1 + (WIN-1)/TX) reads + 1 write per voxel.
template <int WIN_> struct MinOp {
    static constexpr int WIN = WIN_;
    static constexpr bool SYMMETRIC = true;     // v() does not depend on the order of the rows
    typedef typename fvec<4>::type vec;
    static __device__ __forceinline__ vec h(const vec *w)
    {
        vec m = w[0];
#pragma unroll
        for (int k = 1; k < WIN; k++)
            m = vmin<4>(m, w[k]);
        return m;
    }
    static __device__ __forceinline__ vec v(const vec *r) { return h(r); }
};
struct BinomOp {
    static constexpr int WIN = 5;
    static constexpr bool SYMMETRIC = false;
    typedef typename fvec<4>::type vec;
    static __device__ __forceinline__ vec h(const vec *w) { return vtap5<4>(w[0], w[1], w[2], w[3], w[4]); }
    static __device__ __forceinline__ vec v(const vec *r) { return vtap5<4>(r[0], r[1], r[2], r[3], r[4]); }
};

// WTA = true: the filtered value is not stored; PadCosts (StcRawCosts.cpp:437-457) is applied in
// registers, the 8 lanes of a pixel min-reduce (ordered cost, level) and one lane merges the
// result of this 32-level tile into the pixel's key with a 64-bit atomic min (keys start at
// INT64_MAX; column-major like the fused path).  The filtered DSI is never written.
// Lexicographic (s, d) minimum of each aligned group of 8 lanes, delivered to the group's first
// lane (DPP row_shl 1, 2, 4; the levels of a group ascend with the lane, so on equal s the lower
// lane keeps its own pair).
__device__ __forceinline__ void seg8_min_to_first(int &s, int &d)
{
#define SMK_SEG8_STEP(CTRL)                                                              \
    {                                                                                    \
        int os = __builtin_amdgcn_update_dpp(0x7FFFFFFF, s, CTRL, 0xF, 0xF, false);      \
        int od = __builtin_amdgcn_update_dpp(0x7FFFFFFF, d, CTRL, 0xF, 0xF, false);      \
        bool take = os < s;                                                              \
        s = take ? os : s;                                                               \
        d = take ? od : d;                                                               \
    }
    SMK_SEG8_STEP(0x101)   // row_shl:1
    SMK_SEG8_STEP(0x102)   // row_shl:2
    SMK_SEG8_STEP(0x104)   // row_shl:4
#undef SMK_SEG8_STEP
}

struct PadWtaArgs {
    int disp0, dstep;        // disp(k) = disp0 + k*dstep for level k of this DSI
    int d_first;             // global index of level 0
    float outside;
    long long *keys;
};

template <class OP, int TX, bool WTA = false>
__global__ __launch_bounds__(TX * 8) void tile_walk_kernel(const float *__restrict__ src,
                                                           float *__restrict__ dst, int W, int D, int H,
                                                           size_t row_elems, int seglen,
                                                           PadWtaArgs pw = PadWtaArgs())
{
    typedef typename fvec<4>::type vec;
    constexpr int WIN = OP::WIN, PR = WIN / 2, PL = PR - WIN;
    constexpr int HX = TX + WIN - 1;               // pixels per staged row
    constexpr int NT = TX * 8, NL = (HX * 8 + NT - 1) / NT;
    __shared__ vec lds[2][HX * 8];
    const int t = threadIdx.x, px = t >> 3, lq = t & 7;
    const int x0 = blockIdx.x * TX, k0 = blockIdx.y * 32;
    const int y0 = blockIdx.z * seglen, y1 = min(H, y0 + seglen);
    size_t goff[NL];
    bool gact[NL];
#pragma unroll
    for (int n = 0; n < NL; n++) {
        int i = t + n * NT, hp = i >> 3, hl = i & 7;
        gact[n] = (i < HX * 8) && (k0 + hl * 4 < D);
        goff[n] = (size_t)clampi(x0 + PL + 1 + hp, W) * D + k0 + hl * 4;
    }
    vec g[NL];
#pragma unroll
    for (int n = 0; n < NL; n++)
        g[n] = 0.0f;
    auto gload = [&](int yy) {
        const float *r = src + (size_t)clampi(yy, H) * row_elems;
#pragma unroll
        for (int n = 0; n < NL; n++)
            if (gact[n])
                g[n] = vload<4>(r + goff[n]);
    };
    const bool live = (x0 + px < W) && (k0 + lq * 4 < D);
    float *o = dst + (size_t)y0 * row_elems + (size_t)min(x0 + px, W - 1) * D + min(k0 + lq * 4, D - 4);
    vec ring[WIN];
#pragma unroll
    for (int j = 0; j < WIN; j++)
        ring[j] = 0.0f;
    // PadCosts flags of this thread's four levels (they do not depend on the row)
    bool padded[4] = {false, false, false, false};
    if constexpr (WTA) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int xm = x0 + px + pw.disp0 + (k0 + lq * 4 + e) * pw.dstep;
            padded[e] = (unsigned)xm >= (unsigned)W;
        }
    }
    int buf = 0;
    const int ylast = y1 - 1 + PR;
    gload(y0 + PL + 1);
    // Rows are taken in groups of WIN so that a symmetric operator can overwrite the oldest ring
    // slot in place (static slot index) instead of shifting the ring.
    for (int yb = y0 + PL + 1; yb <= ylast; yb += WIN) {
#pragma unroll
        for (int rr = 0; rr < WIN; rr++) {
            const int yy = yb + rr;
            if (yy <= ylast) {                  // (uniform; no break, so the group unrolls)
    #pragma unroll
                for (int n = 0; n < NL; n++)
                    if (t + n * NT < HX * 8)
                        lds[buf][t + n * NT] = g[n];
                if (yy < ylast)
                    gload(yy + 1);
                __syncthreads();
                vec w[WIN];
    #pragma unroll
                for (int j = 0; j < WIN; j++)
                    w[j] = lds[buf][(px + j) * 8 + lq];
                if constexpr (OP::SYMMETRIC) {
                    ring[rr] = OP::h(w);
                } else {
    #pragma unroll
                    for (int j = 0; j < WIN - 1; j++)
                        ring[j] = ring[j + 1];
                    ring[WIN - 1] = OP::h(w);
                }
                if (yy - PR >= y0) {
                    if constexpr (!WTA) {
                        if (live)
                            vstore<4>(o, OP::v(ring));
                        o += row_elems;
                    } else {
                        const vec v = OP::v(ring);
                        int sb = 0x7FFFFFFF, sd = 0x7FFFFFFF;
                        if (live) {
                            float bc = padded[0] ? pw.outside : v[0];
                            int be = 0;
    #pragma unroll
                            for (int e = 1; e < 4; e++) {
                                float c = padded[e] ? pw.outside : v[e];
                                if (c < bc) {               // strict: the lowest level wins inside the lane
                                    bc = c;
                                    be = e;
                                }
                            }
                            sb = ordered_bits(bc);
                            sd = pw.d_first + k0 + lq * 4 + be;
                        }
                        seg8_min_to_first(sb, sd);
                        if (lq == 0 && x0 + px < W)
                            atomicMin(pw.keys + (size_t)(x0 + px) * H + (yy - PR), make_key(sb, sd));
                    }
                }
                buf ^= 1;
            }
        }
    }
}

__global__ void fill_keys_kernel(long long *keys, size_t n, long long value)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        keys[i] = value;
}

}  // namespace smk
