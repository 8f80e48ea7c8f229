// smhip_subpix.h -- raw costs for fractional disparity steps (disp_step = 1/den, den > 1):
// InterpolateLine (StcRawCosts.cpp:67-105), BirchfieldTomasiMinMax on the up-sampled lines
// (:107-128), MatchLine with interpolated-cost averaging (:130-218) and PadLine (:437-457).
// Line buffers are int16 [H][n*nb], n = den*(W-1)+1 samples per row (cubic interpolation can
// leave [0,255], hence 16 bits).  General, untuned kernels: this path is not one of the
// BASELINE configurations.
#pragma once

#include "smhip_kernels.h"

namespace smk {

// Warp1D.cpp:130-142
__device__ __forceinline__ float cubic_interp(float x0, float v0, float v1, float v2, float v3)
{
    float x1 = 1.0f - x0;
    float s0 = v2 - v0;
    float s1 = v1 - v3;
    float d1 = v2 - v1;
    float phi0 = d1 * (x0 * x0) * (2.0f * x1 + 1.0f);
    float phi1a = s0 * x0 * (x1 * x1);
    float phi1b = s1 * x1 * (x0 * x0);
    return v1 + phi0 + phi1a + phi1b;
}

// One thread per (y, x, band): copies pixel x to sample x*s and fills the s-1 samples up to
// pixel x+1 (StcRawCosts.cpp:72-104).
__global__ void interp_lines_kernel(const uint8_t *img, size_t pitch, int16_t *buf, int W, int H,
                                    int nb, int s, int cubic)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // x*nb + band
    int y = blockIdx.y;
    if (i >= W * nb)
        return;
    int x = i / nb, b = i - x * nb;
    const uint8_t *row = img + (size_t)y * pitch;
    size_t n = (size_t)s * (W - 1) + 1;
    int16_t *v = buf + ((size_t)y * n + (size_t)x * s) * nb + b;
    v[0] = row[i];
    if (x >= W - 1)
        return;
    float si = 1.0f / s;
    float i0 = (float)row[i];
    float i1 = (float)row[i + nb];
    if (cubic) {
        float im = (x > 0) ? (float)row[i - nb] : (i0 - (i1 - i0));
        float ip = (x + 1 < W - 1) ? (float)row[i + 2 * nb] : (i1 + (i1 - i0));
        float sf = si;
        for (int is = 1; is < s; is++, sf += si)
            v[(size_t)is * nb] = (int16_t)(int)cubic_interp(sf, im, i0, i1, ip);
    } else {
        float d = (i1 - i0) / (float)s;
        for (int is = 1; is < s; is++) {
            i0 += d;
            v[(size_t)is * nb] = (int16_t)(int)i0;
        }
    }
}

// BirchfieldTomasiMinMax on an int16 line buffer: one thread per (y, sample, band).
__global__ void bt_minmax16_kernel(const int16_t *buf, int16_t *mn, int16_t *mx, int n, int H, int nb)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // q*nb + band
    int y = blockIdx.y;
    if (i >= n * nb)
        return;
    int q = i / nb;
    const int16_t *row = buf + (size_t)y * n * nb;
    int cur = row[i];
    int prv = row[q > 0 ? i - nb : i];
    int nxt = row[q < n - 1 ? i + nb : i];
    int hl = (q > 0) ? (prv + cur + 1) / 2 : cur;
    int hr = (cur + nxt + 1) / 2;
    mn[(size_t)y * n * nb + i] = (int16_t)min(hl, min(cur, hr));
    mx[(size_t)y * n * nb + i] = (int16_t)max(hl, max(cur, hr));
}

struct SubpixArgs {
    const int16_t *rmn, *rmx;   // reference samples / upper interval ends (null: plain difference)
    const int16_t *mmn, *mmx;   // matching samples or lower ends / upper ends (null)
    int W, H, nb, n, den;       // n = den*(W-1)+1
    int D, disp0, dstep;        // disp(k) = disp0 + k*dstep, in samples
    int sd, cutoff, interpolated;
    float outside;
};

// cost1[q] of MatchLine (StcRawCosts.cpp:152-198) as a float
__device__ __forceinline__ float subpix_cost1(const SubpixArgs &a, const int16_t *rmn, const int16_t *rmx,
                                              const int16_t *mmn, const int16_t *mmx, int q, int disp)
{
    int qm = q + disp;
    if (qm < 0 || qm >= a.n)
        return a.outside;
    int sum = 0;
    for (int b = 0; b < a.nb; b++) {
        int rn = rmn[q * a.nb + b], mn = mmn[qm * a.nb + b];
        int d1 = mn - rn;
        if (rmx) {
            int rx = rmx[q * a.nb + b], mx = mmx[qm * a.nb + b];
            int lo = max(rn, mn), hi = min(rx, mx);
            d1 = (lo <= hi) ? 0 : ((mn > rx) ? mn - rx : rn - mx);
        }
        sum += a.sd ? d1 * d1 : abs(d1);
    }
    return (float)min(sum, a.cutoff);
}

// one thread per DSI element (y, pixel i, level k)
__global__ void raw_costs_subpix_kernel(SubpixArgs a, float *cost, size_t row_elems)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;   // i*D + k
    int y = blockIdx.y;
    if (e >= a.W * a.D)
        return;
    int i = e / a.D, k = e - i * a.D;
    size_t ro = (size_t)y * a.n * a.nb;
    const int16_t *rmn = a.rmn + ro, *rmx = a.rmx ? a.rmx + ro : nullptr;
    const int16_t *mmn = a.mmn + ro, *mmx = a.mmx ? a.mmx + ro : nullptr;
    int disp = a.disp0 + k * a.dstep;
    int x = i * a.den;
    float c;
    if (a.interpolated) {                            // StcRawCosts.cpp:200-214
        int dh = a.den / 2;
        float box_scale = (float)(1.0 / (2 * dh + 1));
        float acc = 0;
        for (int t = -dh; t <= dh; t++)
            acc += subpix_cost1(a, rmn, rmx, mmn, mmx, clampi(x + t, a.n), disp);
        c = (float)(int)((double)(box_scale * acc) + 0.5);
    } else {
        c = subpix_cost1(a, rmn, rmx, mmn, mmx, x, disp);
    }
    cost[(size_t)y * row_elems + e] = c;
}

// PadLine (StcRawCosts.cpp:437-457) for any den: one thread per DSI element
__global__ void pad_costs_subpix_kernel(float *cost, size_t row_elems, int W, int D, int n, int den,
                                        int disp0, int dstep, float outside)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (e >= W * D)
        return;
    int i = e / D, k = e - i * D;
    int x = disp0 + k * dstep - (den - 1) + i * den;
    if (x < 0 || x >= n)
        cost[(size_t)y * row_elems + e] = outside;
}

// PreProcess (StcPreProcess.cpp:49-55): ConvolveSeparable<uchar>(img, img, 121, 14641) -- a
// horizontal 3-tap pass and a vertical 5-tap pass, each rounded down to uchar (Convolve.cpp:
// 224-225, Convert.cpp:41-45).  One thread per byte; images are a few MB.
__device__ __forceinline__ uint8_t to_uchar_clip(float v)
{
    v = v > 0.0f ? v : 0.0f;
    v = v < 255.0f ? v : 255.0f;
    return (uint8_t)v;
}

__global__ void blur121_h_kernel(const uint8_t *src, uint8_t *dst, size_t pitch, int W, int H, int nb)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // x*nb + band
    int y = blockIdx.y;
    if (i >= W * nb)
        return;
    int x = i / nb, b = i - x * nb;
    const uint8_t *r = src + (size_t)y * pitch;
    float sum = 0.0f;
    sum += 0.25f * (float)r[clampi(x - 1, W) * nb + b];
    sum += 0.5f * (float)r[i];
    sum += 0.25f * (float)r[clampi(x + 1, W) * nb + b];
    dst[(size_t)y * pitch + i] = to_uchar_clip(sum);
}

__global__ void blur14641_v_kernel(const uint8_t *src, uint8_t *dst, size_t pitch, int W, int H, int nb)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * nb)
        return;
    float sum = 0.0f;
    sum += 0.0625f * (float)src[(size_t)clampi(y - 2, H) * pitch + i];
    sum += 0.25f * (float)src[(size_t)clampi(y - 1, H) * pitch + i];
    sum += 0.375f * (float)src[(size_t)y * pitch + i];
    sum += 0.25f * (float)src[(size_t)clampi(y + 1, H) * pitch + i];
    sum += 0.0625f * (float)src[(size_t)clampi(y + 2, H) * pitch + i];
    dst[(size_t)y * pitch + i] = to_uchar_clip(sum);
}

}  // namespace smk
