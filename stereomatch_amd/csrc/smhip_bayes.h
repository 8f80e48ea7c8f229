// smhip_bayes.h -- one iteration of Bayesian diffusion (AggrBayesian, StcDiffusion.cpp:139-308).
//
//   1. P(d)  = exp(-E(d)) / sum            per pixel           (float exp, double accumulation)
//   2. Ps(d) = sum_d2 w[d][d2] * P(d2)     D x D weights       (double)
//   3. Es(d) = -log(max(1e-16, Ps(d)))                          (double log, stored as float)
//   4. E     = E0 + mu * (Es + left + right + up + down)        (float, 5-point, replicate border)
//
// TOLERANCE MODE: steps 1 and 3 use the device's expf and a short series for log, which differ
// from the host libm the reference uses in the last place (and the host's own result depends on which libm variant
// the CPU selects), so this operator is not bit-exact -- every other operator of the library is.
//
// Step 2 is a D x D product per pixel, but the matrix is a band plus a constant: beyond
// |d - d'| = B the Gaussian term of rhoP underflows and w(d, d') is exactly epsP / s_d =: e_d.
// With sum_d' P(d') = 1:   Ps(d) = sum_{|k| <= B} (w(d, d+k) - e_d) * P(d+k) + e_d,
// 2B+1 terms instead of D (B = 5 for the reference's sigmaP 0.4; B = D-1, i.e. the full product,
// when there is no such cut-off).  The host computes w as the reference does (:209-222), finds B
// and uploads the band; the device sums it with FMAs in double.  One wave per pixel, lane l holds
// levels l, l + 64, ... (D <= 512; the reference stops at 99, MAXDISP :200-202); the normalised
// probabilities go through a wave-private, zero-padded LDS line so lane d reads P(d+k).
#pragma once

#include "smhip_kernels.h"

namespace smk {

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// 1 / d to about an ulp: hardware seed + two Newton steps (tolerance mode: the reference divides)
__device__ __forceinline__ double fast_rcp_f64(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

// -log(x) for a normal positive double: x = m * 2^e with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(z), z = (m - 1) / (m + 1), |z| <= 0.172, odd series up to z^15 (next term
// 3e-14 relative) -- a quarter of the library routine's instructions, far inside the float
// rounding of the stored energy.
__device__ __forceinline__ double neg_log_f64(double x)
{
    const long long b = __double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    double m = __longlong_as_double((b & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double z = (m - 1.0) * fast_rcp_f64(m + 1.0);
    const double z2 = z * z;
    double p = 1.0 / 15.0;
    p = fma(p, z2, 1.0 / 13.0);
    p = fma(p, z2, 1.0 / 11.0);
    p = fma(p, z2, 1.0 / 9.0);
    p = fma(p, z2, 1.0 / 7.0);
    p = fma(p, z2, 1.0 / 5.0);
    p = fma(p, z2, 1.0 / 3.0);
    p = fma(p, z2, 1.0);
    return -fma((double)e, 0.6931471805599453, 2.0 * z * p);
}

constexpr int BAYES_MAXD = 512;          // == SMHIP_BAYES_MAX_LEVELS

// steps 1-3: E -> Es.  wb = band [(2B+1)][64*SLOTS] of (w - e_d), zero outside the matrix;
// ed = e_d per level.  Lane l holds levels l, l + 64, ..., l + 64 (SLOTS - 1) (SLOTS = 1, 2, 4, 8: up to
// 512 levels; the reference stops at 99).  E0 != nullptr (first iteration): E *= scale first, and
// E0 = E (:182-191).
// HB > 0: bands up to half-width HB are held in registers for the whole kernel (the weights
// depend on the level only; used for SLOTS <= 2); HB = 0: any band, weights read in the loop.
// FUSED: E is not read but formed on the fly from the previous iteration's smoothed energies
// EsIn, E = E0 + mu * (Es + left + right + up + down) -- step 4 of that iteration, the same float
// expression as bayes_update_kernel -- so a run of iterations moves 12 instead of 20 bytes per voxel
// and iteration (smhip_bayesian_run).
template <int SLOTS, int HB, bool FUSED = false>
__global__ __launch_bounds__(256) void bayes_smooth_kernel(float *__restrict__ E, float *__restrict__ Es,
                                                           float *__restrict__ E0, float scale,
                                                           const double *__restrict__ wb,
                                                           const double *__restrict__ ed, int B, int W, int H,
                                                           int D, size_t row_elems,
                                                           const float *__restrict__ EsIn = nullptr, float mu = 0.0f)
{
    constexpr int DP = 64 * SLOTS;
    constexpr int BASE = DP, LINE = 3 * DP;             // zero margins of DP on both sides (B < D <= DP)
    extern __shared__ __attribute__((aligned(16))) double bayes_lines[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *pl = bayes_lines + (size_t)wave * LINE;
    for (int i = lane; i < LINE; i += 64)
        pl[i] = 0.0;
    bool in[SLOTS];
    double edl[SLOTS];
    double wr[SLOTS][2 * HB + 1];
#pragma unroll
    for (int q = 0; q < SLOTS; q++) {
        const int d = lane + 64 * q;
        in[q] = d < D;
        edl[q] = in[q] ? ed[d] : 0.0;
#pragma unroll
        for (int j = 0; j < 2 * HB + 1; j++) {
            const int k = j - HB;
            wr[q][j] = (HB > 0 && k >= -B && k <= B) ? wb[(size_t)(k + B) * DP + d] : 0.0;
        }
    }
    // wave-uniform walk over the pixels, (x, y) stepped without a division per pixel
    const int n_waves = (int)gridDim.x * 4, first = (int)blockIdx.x * 4 + wave;
    const int step_y = n_waves / W, step_x = n_waves - step_y * W;
    int y = first / W, x = first - y * W;
    for (; y < H; x += step_x, y += step_y) {
        if (x >= W) {
            x -= W;
            y++;
            if (y >= H)
                break;
        }
        const size_t o = (size_t)y * row_elems + (size_t)x * D + lane;
        float e[SLOTS];
        if constexpr (FUSED) {
            const long lf = x > 0 ? -(long)D : 0, rt = x < W - 1 ? (long)D : 0;
            const long up = y > 0 ? -(long)row_elems : 0, dn = y < H - 1 ? (long)row_elems : 0;
#pragma unroll
            for (int q = 0; q < SLOTS; q++) {
                e[q] = 0.0f;
                if (in[q]) {
                    const float *s = EsIn + o + 64 * q;
                    e[q] = E0[o + 64 * q] + mu * (s[0] + s[lf] + s[rt] + s[up] + s[dn]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < SLOTS; q++)
                e[q] = in[q] ? E[o + 64 * q] : 0.0f;
        }
        if (!FUSED && E0) {
#pragma unroll
            for (int q = 0; q < SLOTS; q++) {
                e[q] = e[q] * scale + 0.0f;
                if (in[q])
                    E[o + 64 * q] = E0[o + 64 * q] = e[q];
            }
        }
        double pr[SLOTS], part = 0.0;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            pr[q] = in[q] ? (double)expf(-e[q]) : 0.0;       // :236
            part += pr[q];
        }
        const double s = wave_sum_f64(part);
        const double rs = s == 0 ? 0.0 : fast_rcp_f64(s);
#pragma unroll
        for (int q = 0; q < SLOTS; q++)
            pr[q] = s == 0 ? (in[q] ? 1.0 / D : 0.0) : pr[q] * rs;   // :238-241
        __builtin_amdgcn_wave_barrier();                    // the previous pixel's reads are done
#pragma unroll
        for (int q = 0; q < SLOTS; q++)
            if (in[q])
                pl[BASE + lane + 64 * q] = pr[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double ps[SLOTS];                                   // :247-250, band + constant
#pragma unroll
        for (int q = 0; q < SLOTS; q++)
            ps[q] = edl[q];
        if (HB > 0) {
#pragma unroll
            for (int j = 0; j < 2 * HB + 1; j++)
#pragma unroll
                for (int q = 0; q < SLOTS; q++)
                    ps[q] = fma(wr[q][j], pl[BASE + lane + 64 * q + j - HB], ps[q]);
        } else {
            const double *wk = wb + lane;
            for (int k = -B; k <= B; k++, wk += DP)
#pragma unroll
                for (int q = 0; q < SLOTS; q++)
                    ps[q] = fma(wk[64 * q], pl[BASE + lane + 64 * q + k], ps[q]);
        }
#pragma unroll
        for (int q = 0; q < SLOTS; q++)
            if (in[q])
                Es[o + 64 * q] = (float)neg_log_f64(1e-16 > ps[q] ? 1e-16 : ps[q]);   // :253
    }
}

// step 4 (:261-307): E = E0 + mu * ((((Es + left) + right) + up) + down), missing neighbour = centre
__global__ void bayes_update_kernel(float *__restrict__ E, const float *__restrict__ Es,
                                    const float *__restrict__ E0, int W, int D, int H, size_t row_elems, float mu)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (i >= W * D)
        return;
    const float *s = Es + (size_t)y * row_elems + i;
    long lf = (i - D >= 0) ? -(long)D : 0;
    long rt = (i + D < W * D) ? (long)D : 0;
    long up = (y > 0) ? -(long)row_elems : 0;
    long dn = (y < H - 1) ? (long)row_elems : 0;
    size_t o = (size_t)y * row_elems + i;
    E[o] = E0[o] + mu * (s[0] + s[lf] + s[rt] + s[up] + s[dn]);
}

}  // namespace smk
