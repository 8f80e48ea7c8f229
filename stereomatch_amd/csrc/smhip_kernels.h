// smhip_kernels.h -- device code of the gfx950 cost-volume front end.
//
// Every kernel here is HBM-bound streaming / stencil / scan work over the
// disparity-space image C[y][x][d] (d innermost).  The mapping used throughout is
// "d on lanes": consecutive lanes hold consecutive disparities of one pixel, so that
// every wave-wide load/store of the DSI is one contiguous 256-B segment.  The serial
// recurrences of the reference's running-sum box filter (BoxFilter.cpp:117-168) are
// kept serial per chain (that is what makes the result bit-identical) and the
// parallelism comes from the H*D (horizontal) or W*D (vertical) independent chains.
//
// Compiled with -ffp-contract=off: the reference CPU path never fuses a*b+c.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smhip_fused_defs.h"

// The non-template kernels below are defined in a header: a second translation unit that includes it (for the
// device helpers) defines SMK_GLOBAL as `static __global__ __attribute__((unused))` so that the library holds one
// external definition of each.
#ifndef SMK_GLOBAL
#define SMK_GLOBAL __global__
#endif

namespace smk {

// Parameters of the raw-cost evaluation, shared by the stand-alone and fused kernels.
struct RawArgs {
    const uint8_t *rmn, *rmx;   // reference line source: values, and upper interval end (BT) or null
    const uint8_t *mmn, *mmx;   // matching line source: values / interval lower end, upper end or null
    size_t pitch;               // bytes per image row (all four arrays)
    int W, H, nb;               // image shape
    int D;                      // number of disparity levels handled (slab size)
    int disp0, dstep;           // disp(k) = disp0 + k*dstep  (= -sign*(disp_min + (d_first+k)*num))
    int sd;                     // 1: squared differences, 0: absolute
    int cutoff;                 // truncation (StcRawCosts.cpp:144)
    int outside;                // m_match_outside as an integer (StcRawCosts.cpp:365-367)
};


// MatchLine, StcRawCosts.cpp:152-198 (den = 1): cost of reference pixel x at level k.
__device__ __forceinline__ int raw_cost(const RawArgs &a, const uint8_t *rmn, const uint8_t *rmx,
                                        const uint8_t *mmn, const uint8_t *mmx, int x, int k)
{
    int xm = x + a.disp0 + k * a.dstep;
    if (xm < 0 || xm >= a.W)
        return a.outside;
    int sum = 0;
    for (int b = 0; b < a.nb; b++) {
        int rn = rmn[x * a.nb + b];
        int mn = mmn[xm * a.nb + b];
        int d1 = mn - rn;
        if (rmx) {
            int rx = rmx[x * a.nb + b];
            int mx = mmx[xm * a.nb + b];
            int lo = max(rn, mn), hi = min(rx, mx);
            d1 = (lo <= hi) ? 0 : ((mn > rx) ? mn - rx : rn - mx);
        }
        sum += a.sd ? d1 * d1 : abs(d1);
    }
    return min(sum, a.cutoff);
}

// BirchfieldTomasiMinMax, StcRawCosts.cpp:107-128: one thread per (y, x, band).
SMK_GLOBAL void bt_minmax_kernel(const uint8_t *img, uint8_t *mn, uint8_t *mx, size_t pitch,
                                 int W, int H, int nb)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // x*nb + band
    int y = blockIdx.y;
    if (i >= W * nb)
        return;
    int x = i / nb;
    const uint8_t *row = img + (size_t)y * pitch;
    int cur = row[i];
    int prv = row[x > 0 ? i - nb : i];
    int nxt = row[x < W - 1 ? i + nb : i];
    int hl = (x > 0) ? (prv + cur + 1) / 2 : cur;    // h[-1] = v[0]
    int hr = (cur + nxt + 1) / 2;
    mn[(size_t)y * pitch + i] = (uint8_t)min(hl, min(cur, hr));
    mx[(size_t)y * pitch + i] = (uint8_t)max(hl, max(cur, hr));
}

// RawCostsCPU, StcRawCosts.cpp:370-417 (den = 1): one thread per DSI element.
SMK_GLOBAL void raw_costs_kernel(RawArgs a, float *cost, size_t row_elems)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // x*D + k
    int y = blockIdx.y;
    if (i >= a.W * a.D)
        return;
    int x = i / a.D, k = i - x * a.D;
    size_t ro = (size_t)y * a.pitch;
    int c = raw_cost(a, a.rmn + ro, a.rmx ? a.rmx + ro : nullptr, a.mmn + ro,
                     a.mmx ? a.mmx + ro : nullptr, x, k);
    cost[(size_t)y * row_elems + i] = (float)c;
}

// boxFilterLines, BoxFilter.cpp:70-170, replicate border.  One thread per running sum:
// chain c -> (o = c / n_sums, k = c % n_sums) starts at o*outer_stride + k and steps by
// `stride` elements.  Horizontal pass: n_sums = D, stride = D, outer = rows.  Vertical
// pass: n_sums = W*D, stride = row_elems, one outer.
// borderIndex, BoxFilter.cpp:29-65: 1 replicate, 2 reflect, 3 cyclic (no double wrapping: |overhang| < len)
__device__ __forceinline__ int border_index(int i, int len, int mode)
{
    if (i < 0)
        return mode == 2 ? -i : (mode == 3 ? i + len : 0);
    if (i >= len)
        return mode == 2 ? len + len - i - 2 : (mode == 3 ? i - len : len - 1);
    return i;
}

// the same running sums with the reference's other border modes (smhip_box_filter_border); not a hot path
SMK_GLOBAL void box_lines_border_kernel(const float *src, float *dst, long n_chains, int n_sums,
                                        size_t outer_stride, size_t stride, int len, int w, float scale, int mode)
{
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains)
        return;
    long o = c / n_sums;
    size_t base = (size_t)o * outer_stride + (size_t)(c - o * n_sums);
    const float *s = src + base;
    float *t = dst + base;
    int pr = w / 2, pl = pr - w;
    float acc = 0;
    for (int i = pl + 1; i <= pr; i++)
        acc += scale * s[stride * border_index(i, len, mode)];
    t[0] = acc;
    for (int x = 1; x < len; x++) {
        float lft = s[stride * border_index(x + pl, len, mode)];
        float rgt = s[stride * border_index(x + pr, len, mode)];
        acc = acc + scale * (-lft + rgt);
        t[stride * x] = acc;
    }
}

SMK_GLOBAL void box_lines_kernel(const float *src, float *dst, long n_chains, int n_sums,
                                 size_t outer_stride, size_t stride, int len, int w, float scale)
{
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains)
        return;
    long o = c / n_sums;
    size_t base = (size_t)o * outer_stride + (size_t)(c - o * n_sums);
    const float *s = src + base;
    float *t = dst + base;
    int pr = w / 2, pl = pr - w;
    float acc = 0;
    for (int i = pl + 1; i <= pr; i++)
        acc += scale * s[stride * clampi(i, len)];
    t[0] = acc;
#pragma unroll 4
    for (int x = 1; x < len; x++) {
        float lft = s[stride * clampi(x + pl, len)];
        float rgt = s[stride * clampi(x + pr, len)];
        acc = acc + scale * (-lft + rgt);
        t[stride * x] = acc;
    }
}

// MinMaxFilter, MinFilter.cpp:122-202 (min, replicate): o = min over [x+pl+1, x+pr].
// One thread per output element; `axis_stride` is the element spacing along the filtered
// axis, `pos`/`len` the coordinate along it.
SMK_GLOBAL void min_h_kernel(const float *src, float *dst, int W, int D, size_t row_elems, int w)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * D)
        return;
    int x = i / D, d = i - x * D;
    int pr = w / 2, pl = pr - w;
    const float *row = src + (size_t)y * row_elems + d;
    float m = row[(size_t)clampi(x + pl + 1, W) * D];
    for (int k = x + pl + 2; k <= x + pr; k++) {
        float v = row[(size_t)clampi(k, W) * D];
        m = v < m ? v : m;
    }
    dst[(size_t)y * row_elems + i] = m;
}

SMK_GLOBAL void min_v_kernel(const float *src, float *dst, int WD, int H, size_t row_elems, int w)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= WD)
        return;
    int pr = w / 2, pl = pr - w;
    float m = src[(size_t)clampi(y + pl + 1, H) * row_elems + i];
    for (int k = y + pl + 2; k <= y + pr; k++) {
        float v = src[(size_t)clampi(k, H) * row_elems + i];
        m = v < m ? v : m;
    }
    dst[(size_t)y * row_elems + i] = m;
}

// ConvolveRow2D with the 1x5 / 5x1 binomial, Convolve.cpp:96-110: sum = 0; sum += k[l]*s[l].
__device__ __forceinline__ float tap5(float s0, float s1, float s2, float s3, float s4)
{
    float acc = 0.0f;
    acc += 0.0625f * s0;
    acc += 0.25f * s1;
    acc += 0.375f * s2;
    acc += 0.25f * s3;
    acc += 0.0625f * s4;
    return acc;
}

SMK_GLOBAL void binom_h_kernel(const float *src, float *dst, int W, int D, size_t row_elems)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * D)
        return;
    int x = i / D, d = i - x * D;
    const float *row = src + (size_t)y * row_elems + d;
    dst[(size_t)y * row_elems + i] =
        tap5(row[(size_t)clampi(x - 2, W) * D], row[(size_t)clampi(x - 1, W) * D],
             row[(size_t)x * D], row[(size_t)clampi(x + 1, W) * D],
             row[(size_t)clampi(x + 2, W) * D]);
}

SMK_GLOBAL void binom_v_kernel(const float *src, float *dst, int WD, int H, size_t row_elems)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= WD)
        return;
    dst[(size_t)y * row_elems + i] =
        tap5(src[(size_t)clampi(y - 2, H) * row_elems + i], src[(size_t)clampi(y - 1, H) * row_elems + i],
             src[(size_t)y * row_elems + i], src[(size_t)clampi(y + 1, H) * row_elems + i],
             src[(size_t)clampi(y + 2, H) * row_elems + i]);
}

// AggrDiffusion, StcDiffusion.cpp:75-136.
SMK_GLOBAL void diffusion_kernel(const float *src, float *dst, const float *cost0, int W, int D,
                                 int H, size_t row_elems, float lambda, float a, float lb,
                                 int membrane)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * D)
        return;
    const float *s = src + (size_t)y * row_elems + i;
    long lf = (i - D >= 0) ? -(long)D : 0;
    long rt = (i + D < W * D) ? (long)D : 0;
    long up = (y > 0) ? -(long)row_elems : 0;
    long dn = (y < H - 1) ? (long)row_elems : 0;
    float v = a * s[0] + lambda * (s[lf] + s[rt] + s[up] + s[dn]);
    if (membrane)
        v += lb * cost0[(size_t)y * row_elems + i];
    dst[(size_t)y * row_elems + i] = v;
}

// PadCosts, StcRawCosts.cpp:437-476 (den = 1): C[y][i][k] = outside where i + disp(k) is
// off the line.  One thread per (y, k, j): candidate pixels j from the left end and
// W-1-j from the right end.
SMK_GLOBAL void pad_costs_kernel(float *cost, size_t row_elems, int W, int D, int disp0, int dstep,
                                 int maxpad, float outside)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (t >= D * maxpad)
        return;
    int j = t / D, k = t - j * D;
    int disp = disp0 + k * dstep;
    float *row = cost + (size_t)y * row_elems + k;
    int il = j, ir = W - 1 - j;
    if (il < W && (il + disp < 0 || il + disp >= W))
        row[(size_t)il * D] = outside;
    if (ir >= 0 && ir != il && (ir + disp < 0 || ir + disp >= W))
        row[(size_t)ir * D] = outside;
}

// ---- winner-take-all ----------------------------------------------------------------

// float order -> signed int order (so that integer min == float min; -0 is folded to +0)
__device__ __forceinline__ int ordered_bits(float c)
{
    int b = __float_as_int(c + 0.0f);
    return b ^ ((b >> 31) & 0x7FFFFFFF);
}
__device__ __forceinline__ float unordered_bits(int s)
{
    return __int_as_float(s ^ ((s >> 31) & 0x7FFFFFFF));
}
__device__ __forceinline__ long long make_key(int s, int d)
{
    return (long long)(((unsigned long long)(unsigned)s << 32) | (unsigned)d);
}

// min over a segment of SEG consecutive lanes of (s, d) in lexicographic order; every lane of
// the segment receives the result.
template <int SEG>
__device__ __forceinline__ void seg_min(int &s, int &d)
{
#pragma unroll
    for (int off = 1; off < SEG; off <<= 1) {
        int os = __shfl_xor(s, off);
        int od = __shfl_xor(d, off);
        bool take = (os < s) || (os == s && od < d);
        s = take ? os : s;
        d = take ? od : d;
    }
}

// OptWTA, StcOptimize.cpp:26-56.  One wave per pixel (lanes stride over d), wave-level
// lexicographic (cost, d) min => first minimum wins exactly as the serial strict '<' scan.
SMK_GLOBAL void wta_wave_kernel(const float *cost, int W, int H, int D, size_t row_elems,
                                int32_t *disp, float *best)
{
    int lane = threadIdx.x & 63;
    long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    long n_waves = ((long)gridDim.x * blockDim.x) >> 6;
    long n_pix = (long)W * H;
    for (long p = wave; p < n_pix; p += n_waves) {
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        const float *c = cost + (size_t)y * row_elems + (size_t)x * D;
        int s = 0x7FFFFFFF, bd = 0x7FFFFFFF;
        for (int d = lane; d < D; d += 64) {
            int v = ordered_bits(c[d]);
            if (v < s) { s = v; bd = d; }
        }
        seg_min<64>(s, bd);
        if (lane == 0) {
            if (disp) disp[p] = bd;
            if (best) best[p] = c[bd];
        }
    }
}

// Small D: one thread per pixel, serial scan.
SMK_GLOBAL void wta_thread_kernel(const float *cost, int W, int H, int D, size_t row_elems,
                                  int32_t *disp, float *best)
{
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)W * H)
        return;
    int y = (int)(p / W), x = (int)(p - (long)y * W);
    const float *c = cost + (size_t)y * row_elems + (size_t)x * D;
    int bd = 0;
    float bc = c[0];
    for (int d = 1; d < D; d++) {
        float v = c[d];
        if (v < bc) { bc = v; bd = d; }
    }
    if (disp) disp[p] = bd;
    if (best) best[p] = bc;
}

SMK_GLOBAL void keys_min_kernel(const long long *part, int n_part, size_t n, long long *out, int merge = 0)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    long long m = part[i];
    if (merge) {                        // out[] holds the keys of other levels
        long long o = out[i];
        m = o < m ? o : m;
    }
    for (int c = 1; c < n_part; c++) {
        long long v = part[(size_t)c * n + i];
        m = v < m ? v : m;
    }
    out[i] = m;
}

// ---- fused path ------------------------------------------------------------------------
//
// Pass A: raw cost + horizontal running sum -> slab T[y][x][Dp] (Dp = padded slab width).
// One thread per (y, k) chain, k on lanes.  The raw costs entering and leaving the window
// are recomputed from the image lines (integers, exact in float), so the raw DSI is never
// written: HBM traffic = one write of T.
SMK_GLOBAL void fused_rawbox_h_kernel(RawArgs a, float *T, size_t trow, int Dp, int w, float scale)
{
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (long)a.H * Dp)
        return;
    int y = (int)(c / Dp), k = (int)(c - (long)y * Dp);
    if (k >= a.D)
        return;
    size_t ro = (size_t)y * a.pitch;
    const uint8_t *rmn = a.rmn + ro, *rmx = a.rmx ? a.rmx + ro : nullptr;
    const uint8_t *mmn = a.mmn + ro, *mmx = a.mmx ? a.mmx + ro : nullptr;
    int W = a.W;
    int pr = w / 2, pl = pr - w;
    float *t = T + (size_t)y * trow + k;
    float acc = 0;
    for (int i = pl + 1; i <= pr; i++)
        acc += scale * (float)raw_cost(a, rmn, rmx, mmn, mmx, clampi(i, W), k);
    t[0] = acc;
    for (int x = 1; x < W; x++) {
        float lft = (float)raw_cost(a, rmn, rmx, mmn, mmx, clampi(x + pl, W), k);
        float rgt = (float)raw_cost(a, rmn, rmx, mmn, mmx, clampi(x + pr, W), k);
        acc = acc + scale * (-lft + rgt);
        t[(size_t)x * Dp] = acc;
    }
}

// Pass B for narrow slabs (Dp a power of two below 64): vertical running sum over T +
// PadCosts + WTA.  Thread = (pixel column, level), the SEG = Dp lanes of a pixel are one
// aligned lane segment of a wave and min-reduce (ordered cost, d) with shuffles.  Keys are
// written column-major (pixel (x,y) at x*H + y), like the wide-slab kernel of smhip_fused_b.h.
template <int SEG>
SMK_GLOBAL void fused_boxv_wta_kernel(const float *T, size_t trow, int W, int H, int D, int w,
                                      float scale, int disp0, int dstep, float outside,
                                      int d_first, long long *keys)
{
    int tid = threadIdx.x;
    int pix_per_blk = blockDim.x / SEG;
    int pl_local = tid / SEG, k = tid - pl_local * SEG;
    int x = blockIdx.x * pix_per_blk + pl_local;
    bool live = (x < W) && (k < D);
    int xs = x < W ? x : W - 1;
    int ks = k < D ? k : D - 1;
    const float *t = T + (size_t)xs * SEG + ks;
    size_t stride = trow;
    int pr = w / 2, pl = pr - w;
    int disp = disp0 + ks * dstep;
    bool padded = (xs + disp < 0) || (xs + disp >= W);   // PadLine, StcRawCosts.cpp:451-456

    float acc = 0;
    for (int i = pl + 1; i <= pr; i++)
        acc += scale * t[stride * clampi(i, H)];
    for (int y = 0; y < H; y++) {
        if (y > 0) {
            float lft = t[stride * clampi(y + pl, H)];
            float rgt = t[stride * clampi(y + pr, H)];
            acc = acc + scale * (-lft + rgt);
        }
        float v = padded ? outside : acc;
        int s = live ? ordered_bits(v) : 0x7FFFFFFF;
        int d = live ? d_first + k : 0x7FFFFFFF;
        seg_min<SEG>(s, d);
        if (k == 0 && x < W)
            keys[(size_t)x * H + y] = make_key(s, d);
    }
}

}  // namespace smk
