// smhip_refine.h -- the along-d stages behind the aggregated DSI: AggrSubPixelFit
// (StcAggregate.cpp:66-132), AggrCollapse (:134-230) and Refine (StcRefine.cpp:28-100).
// Each output depends on at most df + 2 neighbouring levels of its own pixel, so these are
// one-thread-per-output kernels.  The reference evaluates the parabola in double precision
// and rounds to float at each assignment; the kernels do the same.
#pragma once

#include "smhip_kernels.h"

namespace smk {

// parabola through (-1, c0), (0, c1), (1, c2): curvature a and slope b (StcAggregate.cpp:107-108)
__device__ __forceinline__ void parabola(float c0, float c1, float c2, float &a, float &b)
{
    a = (float)(0.5 * (((double)c0 - 2.0 * (double)c1) + (double)c2));
    b = (float)(0.5 * (double)(c2 - c0));
}
__device__ __forceinline__ bool parabola_ok(float a, float b)   // :109-110, StcRefine.cpp:90
{
    return !((double)a <= 0.0 || (double)a < 0.5 * fabs((double)b));
}

// AggrSubPixelFit, src -> dst (+ mind, cert); c0, c1, c2 are the values before the fit
__global__ void subpixel_fit_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                    float *__restrict__ mind, float *__restrict__ cert, int W, int D,
                                    size_t row_elems, float outside)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * D)
        return;
    int x = i / D, d = i - x * D;
    const float *c = src + (size_t)y * row_elems + (size_t)x * D;
    float c0 = c[max(d - 1, 0)], c1 = c[d], c2 = c[min(d + 1, D - 1)];
    float oc = c1, om = 0.0f, oe = 0.0f;
    if (!(c0 == outside || c1 == outside || c2 == outside)) {
        if (c1 <= c0 && c1 <= c2) {
            float a, b;
            parabola(c0, c1, c2, a, b);
            if (parabola_ok(a, b)) {
                float dn = (float)(-0.5 * (double)(b / a));
                float cn = (float)((double)c1 + (0.5 * (double)b) * (double)dn);
                if (!((double)cn < 0.0)) {
                    oc = (0.0f > cn) ? 0.0f : cn;
                    om = dn;
                    oe = a;
                }
            }
        } else {
            oc = 0.5f * (c1 + (c0 < c2 ? c0 : c2));
            om = (c0 < c2) ? -0.5f : 0.5f;
        }
    }
    size_t o = (size_t)y * row_elems + i;
    dst[o] = oc;
    mind[o] = om;
    cert[o] = oe;
}

// AggrCollapse: integer level d2 takes the first minimum of fractional levels
// [d2*df - df2, d2*df + df - df2) clipped to [0, D1)
__global__ void collapse_kernel(const float *__restrict__ cost1, const float *__restrict__ mind1, int W,
                                int D1, size_t row_elems1, int df, int D2, size_t row_elems2,
                                float disp_step, float *__restrict__ cost2, float *__restrict__ mind2)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * D2)
        return;
    int x = i / D2, d2 = i - x * D2;
    int df2 = df / 2;
    int lo = d2 == 0 ? 0 : d2 * df - df2;
    int hi = min(D1, d2 * df + df - df2);
    const float *c1 = cost1 + (size_t)y * row_elems1 + (size_t)x * D1;
    int bst = lo;
    float bc = c1[lo];
    for (int d1 = lo + 1; d1 < hi; d1++) {
        float v = c1[d1];
        if (v < bc) {
            bc = v;
            bst = d1;
        }
    }
    size_t o = (size_t)y * row_elems2 + i;
    cost2[o] = bc;
    if (mind1)
        mind2[o] = ((mind1[(size_t)y * row_elems1 + (size_t)x * D1 + bst] + (float)bst) - (float)(d2 * df)) * disp_step;
    else
        mind2[o] = (float)(bst - d2 * df) * disp_step;
}

// Refine: mode 0 ScaleAndOffset only, 1 offsets from mind, 2 parabola at the winner
__global__ void refine_kernel(const float *__restrict__ cost, const float *__restrict__ mind,
                              const int32_t *__restrict__ disp, int W, int H, int D, size_t row_elems,
                              float step, float d_offset, int mode, float *__restrict__ fdisp)
{
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)W * H)
        return;
    int y = (int)(p / W), x = (int)(p - (long)y * W);
    int dw = min(max(disp[p], 0), D - 1);            // winners come from OptWTA: in range
    float f = (step != 1.0f || d_offset != 0.0f) ? (float)dw * step + d_offset : (float)dw;
    size_t l = (size_t)y * row_elems + (size_t)x * D;
    if (mode == 1 && D >= 3) {
        float x0 = mind[l + dw];
        f = step * ((float)dw + x0) + d_offset;
    } else if (mode == 2 && D >= 3) {
        int dm = dw + (dw == 0) - (dw == D - 1);
        float c0 = cost[l + dm - 1], c1 = cost[l + dm], c2 = cost[l + dm + 1];
        float a, b;
        parabola(c0, c1, c2, a, b);
        if (parabola_ok(a, b)) {
            float x0 = (float)((-0.5 * (double)b) / (double)a);
            f = step * ((float)dm + x0) + d_offset;
        }
    }
    fdisp[p] = f;
}

// ComputeMatchQuality (StcEvaluate.cpp:760-833): per pixel the aggregated cost at the (sub-pixel) winner
// -- the parabola through the winner's neighbours evaluated at the fractional offset of the float
// disparity -- and its curvature as a certainty; the cost is divided by the number of colour bands and,
// for squared differences, square-rooted (:829-831).  use_fit (aggr_subpixel): both come from the fitted
// volumes (:795-800; cert is indexed with the pixel stride D and its own row stride, as the reference does
// after a collapse).  The level is clamped into the DSI (the reference would read outside it).
__global__ void match_quality_kernel(const float *__restrict__ cost, const float *__restrict__ cert,
                                     const float *__restrict__ fdisp, int W, int H, int D, size_t row_elems,
                                     size_t cert_row_elems, float d_offset, float step_inv, int no_interp,
                                     int use_fit, int norm_bands, int take_sqrt, float *__restrict__ fcost,
                                     float *__restrict__ fcert)
{
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)W * H)
        return;
    int y = (int)(p / W), x = (int)(p - (long)y * W);
    const float *c = cost + (size_t)y * row_elems + (size_t)x * D;
    float d_sub = (fdisp[p] - d_offset) * step_inv;                  // :788
    int d = (int)((double)d_sub + 0.5);                              // :789
    float x0 = no_interp ? 0.0f : d_sub - (float)d;                  // :790-792
    d = min(max(d, 0), D - 1);
    float fc, ce;
    if (use_fit) {
        fc = c[d];
        ce = cert[(size_t)y * cert_row_elems + (size_t)x * D + d];
    } else {
        int dm = d + (d == 0) - (d == D - 1);
        float c0 = c[dm - 1], c1 = c[dm], c2 = c[dm + 1];
        float a, b;
        parabola(c0, c1, c2, a, b);
        if (parabola_ok(a, b)) {
            fc = a * x0 * x0 + b * x0 + c1;                          // :822, float arithmetic, no contraction
            ce = a;
        } else {
            fc = c1;
            ce = 0.0f;
        }
    }
    float favg = fc / (float)norm_bands;
    fcost[p] = take_sqrt ? (float)sqrt((double)favg) : favg;
    fcert[p] = ce;
}

}  // namespace smk
