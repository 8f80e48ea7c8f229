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
    // Non-temporal: a DSI written here is read by another kernel, later, and stores that do not stay in the L2
    // as dirty lines cost the read stream of the same kernel less (3840x2160x512: binomial _to 7.22 -> 7.06 ms,
    // diffusion 7.71 -> 7.53, membrane 8.18 -> 7.87, binomial x2 7.2 -> 6.9).
    __builtin_nontemporal_store(v, (typename fvec<V>::type *)p);
}
// plain store, for kernels that only write (raw costs: 3.25 ms plain, 3.51 ms non-temporal at the same size)
template <int V>
__device__ __forceinline__ void vstore_plain(float *p, typename fvec<V>::type v)
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
// One v_min_f32 per element.  `b < a ? b : a` compiles to v_cmp + s_nop + v_cndmask (three issue slots:
// the select has to wait for VCC), and fminf() adds a canonicalising v_max per loaded operand; the
// instruction itself is what is wanted.  It differs from the compare-and-select form only where the
// reference's own two variants differ (MinFilter.cpp:29-82 against :85-120): a window holding both +0
// and -0 yields -0 whatever the order (costs are sums of non-negative terms: no -0 arises on the path),
// and a NaN loses against a number.
__device__ __forceinline__ float fmin_hw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int V>
__device__ __forceinline__ typename fvec<V>::type vmin(typename fvec<V>::type a, typename fvec<V>::type b)
{
    if constexpr (V == 1) {
        return fmin_hw(a, b);
    } else {
        typename fvec<V>::type r;
#pragma unroll
        for (int i = 0; i < V; i++)
            r[i] = fmin_hw(a[i], b[i]);
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
    // The vertical window minimum costs O(1) per row (van Herk / Gil-Werman): rows come in groups of WIN;
    // the window of an output row is a suffix of the previous group plus a prefix of the current one, so
    // one running prefix minimum, one suffix pass at the end of each group and one combining minimum per
    // row replace the WIN - 1 minima over the ring: (3 WIN - 3) / WIN instead of WIN - 1 per row.
    static constexpr bool VANHERK = true;
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
    static constexpr bool VANHERK = false;
    typedef typename fvec<4>::type vec;
    static __device__ __forceinline__ vec h(const vec *w) { return vtap5<4>(w[0], w[1], w[2], w[3], w[4]); }
    static __device__ __forceinline__ vec v(const vec *r) { return vtap5<4>(r[0], r[1], r[2], r[3], r[4]); }
};

// arguments of the PadCosts + WTA stage fused behind a filter (smhip_minwta.h)
struct PadWtaArgs {
    int disp0, dstep;        // disp(k) = disp0 + k*dstep for level k of this DSI
    int d_first;             // global index of level 0
    float outside;
    long long *keys;
};

template <class OP, int TX>
__global__ __launch_bounds__(TX * 8) void tile_walk_kernel(const float *__restrict__ src,
                                                           float *__restrict__ dst, int W, int D, int H,
                                                           size_t row_elems, int seglen)
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
    vec ring[WIN], suf[WIN], pre = 0.0f;        // suf, pre: van Herk state (unused otherwise)
#pragma unroll
    for (int j = 0; j < WIN; j++)
        ring[j] = suf[j] = 0.0f;
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
                if constexpr (OP::VANHERK) {
                    ring[rr] = OP::h(w);
                    pre = rr == 0 ? ring[0] : vmin<4>(pre, ring[rr]);
                } else if constexpr (OP::SYMMETRIC) {
                    ring[rr] = OP::h(w);
                } else {
    #pragma unroll
                    for (int j = 0; j < WIN - 1; j++)
                        ring[j] = ring[j + 1];
                    ring[WIN - 1] = OP::h(w);
                }
                if (yy - PR >= y0) {
                    // window of output row yy - PR = rows yy - WIN + 1 .. yy: suffix rr + 1 .. WIN - 1 of the
                    // previous group and prefix 0 .. rr of this one
                    vec v;
                    if constexpr (OP::VANHERK)
                        v = rr == WIN - 1 ? pre : vmin<4>(suf[rr == WIN - 1 ? 0 : rr + 1], pre);
                    else
                        v = OP::v(ring);
                    if (live)
                        vstore<4>(o, v);
                    o += row_elems;
                }
                buf ^= 1;
            }
        }
        if constexpr (OP::VANHERK) {            // suffix minima of the group just completed
            suf[WIN - 1] = ring[WIN - 1];
#pragma unroll
            for (int j = WIN - 2; j >= 1; j--)
                suf[j] = vmin<4>(ring[j], suf[j + 1]);
        }
    }
}

}  // namespace smk
