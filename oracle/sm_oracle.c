/*
 * sm_oracle.c -- CPU restatement of the StereoMatch cost-volume front end.
 *
 * TEST INFRASTRUCTURE ONLY (see sm_oracle.h).  Plain C99; build with
 *   gcc -O2 -ffp-contract=off -fPIC -shared
 * (no -march=native, no -ffast-math: binary32 arithmetic in source order, the
 * semantics of the reference's `g++ -O2` x86-64 build, Makefile:1).
 *
 * Written from the reference's documented behaviour; citations give the
 * reference lines whose arithmetic each block follows.
 */
#include "sm_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

/* ------------------------------------------------------------------------- */
/* raw costs                                                                  */

static int cutoff_of(const smo_params *p)
{
    /* StcRawCosts.cpp:144, :366 */
    return p->match_fn == SMO_SD ? p->match_max * p->match_max : abs(p->match_max);
}

float smo_match_outside(const smo_params *p)
{
    /* StcRawCosts.cpp:365-367: worst per-pixel cost over all bands, trimmed to the cutoff */
    int worst = p->n_bands * (p->match_fn == SMO_SD ? 255 * 255 : 255);
    return (float)imin(worst, cutoff_of(p));
}

/* Warp1D.cpp:130-142 */
static float cubic_interp(float x0, float v0, float v1, float v2, float v3)
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

/* StcRawCosts.cpp:67-105: fill the s-1 samples between original pixels (spacing s) */
static void interpolate_line(int *buf, int s, int w, int nb, int interp)
{
    float si = 1.0f / s;
    for (int x = 0; x < w - 1; x++) {
        for (int b = 0; b < nb; b++) {
            int *v = &buf[s * x * nb + b];
            float i0 = (float)v[0];
            float i1 = (float)v[s * nb];
            if (interp == SMO_CUBIC) {
                float im = (x > 0) ? (float)v[-s * nb] : (i0 - (i1 - i0));
                float ip = (x + 1 < w - 1) ? (float)v[2 * s * nb] : (i1 + (i1 - i0));
                float sf = si;
                for (int is = 1; is < s; is++, sf += si) {
                    v += nb;
                    v[0] = (int)cubic_interp(sf, im, i0, i1, ip);
                }
            } else {
                float d = (i1 - i0) / (float)s;
                for (int is = 1; is < s; is++) {
                    v += nb;
                    i0 += d;
                    v[0] = (int)i0;
                }
            }
        }
    }
}

/* StcRawCosts.cpp:107-128: Birchfield-Tomasi half-sample min / max per band */
static void bt_min_max(const int *line, int *mn, int *mx, int n, int nb)
{
    for (int k = 0; k < nb; k++) {
        int half_r = line[k];   /* h[-1] = v[0] */
        int next = line[k];
        for (int x = 0, l = k; x < n; x++, l += nb) {
            int half_l = half_r;
            int cur = next;
            if (x < n - 1)
                next = line[l + nb];
            half_r = (cur + next + 1) / 2;
            mn[l] = imin(half_l, imin(cur, half_r));
            mx[l] = imax(half_l, imax(cur, half_r));
        }
    }
}

/* StcRawCosts.cpp:130-218.  rmx/mmx NULL => plain difference. */
static void match_line(int w, int nb, int interpolated, const int *rmn, const int *rmx,
                       const int *mmn, const int *mmx, float *cost, float *cost1,
                       int disp_n, int disp, int den, int match_fn, int match_max,
                       float match_outside)
{
    int n = (w - 1) * den + 1;
    int s = interpolated ? 1 : den;
    int cutoff = match_fn == SMO_SD ? match_max * match_max : abs(match_max);
    const float bad = -1.0f;

    for (int x = 0; x < n; x += s) {
        cost1[x] = bad;
        int xm = x + disp;
        if (xm < 0 || xm >= n)
            continue;
        const int *rn = &rmn[x * nb];
        const int *mn = &mmn[xm * nb];
        int sum = 0;
        for (int ib = 0; ib < nb; ib++) {
            int d1 = mn[ib] - rn[ib];
            if (rmx && mmx) {
                /* :168-180: distance between the intervals [rn,rx] and [mn,mx] */
                int rxv = rmx[x * nb + ib], mxv = mmx[xm * nb + ib];
                int lo = imax(rn[ib], mn[ib]);
                int hi = imin(rxv, mxv);
                if (lo <= hi)
                    d1 = 0;
                else
                    d1 = (mn[ib] > rxv) ? mn[ib] - rxv : rn[ib] - mxv;
            }
            sum += (match_fn == SMO_SD) ? d1 * d1 : abs(d1);
        }
        cost1[x] = (float)imin(sum, cutoff);
    }

    /* :193-198 with UNDEFINED_COST = true (CudaRawCosts.h:23) */
    for (int x = 0; x < n && cost1[x] == bad; x += s)
        cost1[x] = match_outside;
    for (int x = n - 1; x >= 0 && cost1[x] == bad; x -= s)
        cost1[x] = match_outside;

    /* :200-217 */
    int dh = den / 2;
    float box_scale = (float)(1.0 / (2 * dh + 1));
    for (int x = 0, i = 0; i < w; x += den, i++) {
        if (interpolated && den > 1) {
            float acc = 0;
            for (int k = -dh; k <= dh; k++)
                acc += cost1[clampi(x + k, n)];
            cost[(long)i * disp_n] = (float)(int)((double)(box_scale * acc) + 0.5);
        } else {
            cost[(long)i * disp_n] = cost1[x];
        }
    }
}

void smo_raw_costs(const uint8_t *ref, const uint8_t *mtc, long img_row_bytes,
                   const smo_params *p, float *cost, long row_elems)
{
    int w = p->width, h = p->height, nb = p->n_bands, den = p->disp_den;
    int n = den * (w - 1) + 1;                       /* :355 */
    int interval = p->match_interval ? 1 : 0;        /* :354 */
    float outside = smo_match_outside(p);
    size_t len = (size_t)n * nb;
    /* Rows are independent (the reference walks them one after the other with one set of line
     * buffers); with -fopenmp every thread gets its own buffers -- same arithmetic per row. */
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
    int *buf0 = calloc(len * 6, sizeof(int));
    int *buf1 = buf0 + len, *min0 = buf1 + len, *max0 = min0 + len;
    int *min1 = max0 + len, *max1 = min1 + len;
    float *cost1 = calloc((size_t)n, sizeof(float));

#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int y = 0; y < h; y++) {
        const uint8_t *r = ref + (size_t)y * img_row_bytes;
        const uint8_t *m = mtc + (size_t)y * img_row_bytes;
        for (int x = 0; x < w; x++)                  /* :383-390 */
            for (int k = 0; k < nb; k++) {
                buf0[(size_t)x * den * nb + k] = r[x * nb + k];
                buf1[(size_t)x * den * nb + k] = m[x * nb + k];
            }
        if (den > 1) {                               /* :393-397 */
            interpolate_line(buf1, den, w, nb, p->match_interp);
            interpolate_line(buf0, den, w, nb, p->match_interp);
        }
        if (interval) {                              /* :398-402 */
            bt_min_max(buf1, min1, max1, n, nb);
            if (p->match_interpolated)
                bt_min_max(buf0, min0, max0, n, nb);
        }
        for (int k = 0; k < p->disp_n; k++) {        /* :405-416 */
            int disp = -p->frame_diff_sign * (den * p->disp_min + k * p->disp_num);
            const int *rmn = interval ? (p->match_interpolated ? min0 : buf0) : buf0;
            const int *rmx = interval ? (p->match_interpolated ? max0 : buf0) : NULL;
            const int *mmn = interval ? min1 : buf1;
            const int *mmx = interval ? max1 : NULL;
            match_line(w, nb, p->match_interpolated, rmn, rmx, mmn, mmx,
                       cost + (size_t)y * row_elems + k, cost1, p->disp_n, disp, den,
                       p->match_fn, p->match_max, outside);
        }
    }
    free(cost1);
    free(buf0);
    }
}

/* StcRawCosts.cpp:437-476 */
void smo_pad_costs(float *cost, long row_elems, const smo_params *p)
{
    int w = p->width, den = p->disp_den;
    int n = (w - 1) * den + 1;
    float outside = smo_match_outside(p);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < p->height; y++)
        for (int k = 0; k < p->disp_n; k++) {
            int disp = -p->frame_diff_sign * (den * p->disp_min + k * p->disp_num);
            disp -= den - 1;                         /* :448 */
            float *c = cost + (size_t)y * row_elems + k;
            for (int x = disp, i = 0; x < n + disp; x += den, i++)
                if (x < 0 || x >= n)
                    c[(long)i * p->disp_n] = outside;
        }
}

/* ------------------------------------------------------------------------- */
/* box filter                                                                 */

/* BoxFilter.cpp:70-170: n_sums running sums along lines of length len, element
 * spacing `stride`; replicate border (borderIndex :29-65, eBorderReplicate). */
static void box_lines(const float *src, float *dst, long n_sums, int len, long stride,
                      int w, int average)
{
    int pr = w / 2, pl = pr - w;                     /* :209-210 */
    float scale = (float)(average ? 1.0 / w : 1);    /* :89 */

    for (long k = 0; k < n_sums; k++) {              /* :92-107 */
        float acc = 0;
        for (int i = pl + 1; i <= pr; i++)
            acc += scale * src[k + stride * clampi(i, len)];
        dst[k] = acc;
    }
    for (int x = 1; x < len; x++) {                  /* :117-168 */
        const float *lft = src + stride * clampi(x + pl, len);
        const float *rgt = src + stride * clampi(x + pr, len);
        const float *prev = dst + stride * (x - 1);
        float *cur = dst + stride * x;
        for (long k = 0; k < n_sums; k++)
            cur[k] = prev[k] + scale * (-lft[k] + rgt[k]);
    }
}

/* borderIndex, BoxFilter.cpp:29-65 (1 replicate, 2 reflect, 3 cyclic) */
static int border_index(int i, int len, int mode)
{
    if (i < 0)
        return mode == 2 ? -i : (mode == 3 ? i + len : 0);
    if (i >= len)
        return mode == 2 ? len + len - i - 2 : (mode == 3 ? i - len : len - 1);
    return i;
}

/* boxFilterLines with a border mode (BoxFilter.cpp:70-168): first sum :92-107, left border :117-130,
 * centre :133-147, right border :150-166 -- one formula with borderIndex applied on both sides */
static void box_lines_border(const float *src, float *dst, long n_sums, int len, long stride, int w,
                             int average, int mode)
{
    int pr = w / 2, pl = pr - w;
    float scale = (float)(average ? 1.0 / w : 1);
    for (long k = 0; k < n_sums; k++) {
        float acc = 0;
        for (int i = pl + 1; i <= pr; i++)
            acc += scale * src[k + stride * border_index(i, len, mode)];
        dst[k] = acc;
    }
    for (int x = 1; x < len; x++) {
        const float *lft = src + stride * border_index(x + pl, len, mode);
        const float *rgt = src + stride * border_index(x + pr, len, mode);
        const float *prev = dst + stride * (x - 1);
        float *cur = dst + stride * x;
        for (long k = 0; k < n_sums; k++)
            cur[k] = prev[k] + scale * (-lft[k] + rgt[k]);
    }
}

/* BoxFilter (BoxFilter.cpp:175-239) with src.borderMode for the rows (:219) and dst.borderMode for the
 * columns (:231) */
void smo_box_filter_border(float *cost, float *tmp, int w, int h, int nb, long row_elems, int window,
                           int average, int border_x, int border_y)
{
    for (int y = 0; y < h; y++)
        box_lines_border(cost + (size_t)y * row_elems, tmp + (size_t)y * row_elems, nb, w, nb, window,
                         average, border_x);
    box_lines_border(tmp, cost, (long)nb * w, h, row_elems, window, average, border_y);
}

void smo_box_filter(float *cost, float *tmp, int w, int h, int nb, long row_elems,
                    int window, int average)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < h; y++)                      /* :214-220 */
        box_lines(cost + (size_t)y * row_elems, tmp + (size_t)y * row_elems, nb, w, nb,
                  window, average);
#ifdef _OPENMP
    /* the column sums are independent per (x, d): hand each thread a block of them */
    long n_col = (long)nb * w, blk = 4096;
#pragma omp parallel for schedule(static)
    for (long k0 = 0; k0 < n_col; k0 += blk)
        box_lines(tmp + k0, cost + k0, (n_col - k0 < blk) ? n_col - k0 : blk, h, row_elems, window,
                  average);
#else
    box_lines(tmp, cost, (long)nb * w, h, row_elems, window, average);   /* :225-232 */
#endif
}

/* ------------------------------------------------------------------------- */
/* min filter                                                                 */

/* MinFilter.cpp:29-82 computes, for every x, min over buf[x+pl+1 .. x+pr] of the
 * replicate-padded line (its incremental update is an optimisation of exactly that,
 * cf. the slow form :85-120).  min is order independent, so the direct form is exact. */
static void min_lines(const float *src, float *dst, long n_lines, int len, long stride,
                      int w)
{
    int pr = w / 2, pl = pr - w;                     /* :139-140 */
    for (int x = 0; x < len; x++) {
        float *o = dst + stride * x;
        const float *first = src + stride * clampi(x + pl + 1, len);
        for (long k = 0; k < n_lines; k++)
            o[k] = first[k];
        for (int i = x + pl + 2; i <= x + pr; i++) {
            const float *s = src + stride * clampi(i, len);
            for (long k = 0; k < n_lines; k++)
                if (s[k] < o[k])
                    o[k] = s[k];
        }
    }
}

void smo_min_filter(float *cost, float *tmp, int w, int h, int nb, long row_elems,
                    int window)
{
    for (int y = 0; y < h; y++)                      /* rows, :156-170 */
        min_lines(cost + (size_t)y * row_elems, tmp + (size_t)y * row_elems, nb, w, nb,
                  window);
    min_lines(tmp, cost, (long)nb * w, h, row_elems, window);   /* columns, :183-201 */
}

/* ------------------------------------------------------------------------- */
/* separable binomial                                                         */

static const float k14641[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};  /* Convolve.cpp:290 */

void smo_convolve_14641(float *cost, float *tmp, int w, int h, int nb, long row_elems)
{
    /* horizontal, Convolve.cpp:224 via :86-111 with TrimIndex replicate (:42-43) */
    for (int y = 0; y < h; y++) {
        const float *s = cost + (size_t)y * row_elems;
        float *t = tmp + (size_t)y * row_elems;
        for (int x = 0; x < w; x++) {
            const float *t0 = s + (long)nb * clampi(x - 2, w);
            const float *t1 = s + (long)nb * clampi(x - 1, w);
            const float *t2 = s + (long)nb * x;
            const float *t3 = s + (long)nb * clampi(x + 1, w);
            const float *t4 = s + (long)nb * clampi(x + 2, w);
            for (int d = 0; d < nb; d++) {
                float acc = 0.0f;
                acc += k14641[0] * t0[d];
                acc += k14641[1] * t1[d];
                acc += k14641[2] * t2[d];
                acc += k14641[3] * t3[d];
                acc += k14641[4] * t4[d];
                t[(long)nb * x + d] = acc;
            }
        }
    }
    /* vertical, Convolve.cpp:225, top row first; result back into cost (:237-249) */
    for (int y = 0; y < h; y++) {
        const float *r0 = tmp + (size_t)clampi(y - 2, h) * row_elems;
        const float *r1 = tmp + (size_t)clampi(y - 1, h) * row_elems;
        const float *r2 = tmp + (size_t)y * row_elems;
        const float *r3 = tmp + (size_t)clampi(y + 1, h) * row_elems;
        const float *r4 = tmp + (size_t)clampi(y + 2, h) * row_elems;
        float *o = cost + (size_t)y * row_elems;
        for (long i = 0; i < (long)nb * w; i++) {
            float acc = 0.0f;
            acc += k14641[0] * r0[i];
            acc += k14641[1] * r1[i];
            acc += k14641[2] * r2[i];
            acc += k14641[3] * r3[i];
            acc += k14641[4] * r4[i];
            o[i] = acc;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* diffusion / membrane                                                       */

void smo_diffusion_iter(const float *src, float *dst, const float *cost0,
                        int w, int h, int nb, long row_elems, float lambda, float beta)
{
    /* StcDiffusion.cpp:84-99, :112-118: a missing neighbour is replaced by the centre
     * value in that neighbour's slot; sum order ((l + r) + u) + d. */
    float a = 1 - lambda * (beta + 4);
    long dx = nb;
    for (int y = 0; y < h; y++) {
        const float *s = src + (size_t)y * row_elems;
        float *o = dst + (size_t)y * row_elems;
        long up = (y > 0) ? -row_elems : 0;
        long dn = (y < h - 1) ? row_elems : 0;
        for (long i = 0; i < dx * w; i++) {
            long lf = (i - dx >= 0) ? -dx : 0;
            long rt = (i + dx < dx * w) ? dx : 0;
            o[i] = a * s[i] + lambda * (s[i + lf] + s[i + rt] + s[i + up] + s[i + dn]);
        }
    }
    if (beta > 0.0f) {                               /* :122-136 */
        float lb = lambda * beta;
        for (int y = 0; y < h; y++) {
            const float *c0 = cost0 + (size_t)y * row_elems;
            float *o = dst + (size_t)y * row_elems;
            for (long i = 0; i < dx * w; i++)
                o[i] += lb * c0[i];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* WTA, aggregation dispatch, energy, depth image                             */

void smo_wta(const float *cost, int w, int h, int nb, long row_elems,
             int32_t *disp, float *best_cost)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < h; y++) {
        const float *c = cost + (size_t)y * row_elems;
        for (int x = 0; x < w; x++, c += nb) {
            int best = 0;
            float bc = c[0];
            for (int d = 1; d < nb; d++)
                if (c[d] < bc) {                     /* strict: lowest d wins ties, :47 */
                    bc = c[d];
                    best = d;
                }
            if (disp)
                disp[(size_t)y * w + x] = best;
            if (best_cost)
                best_cost[(size_t)y * w + x] = bc;
        }
    }
}

void smo_aggregate(float *cost, long row_elems, const smo_params *p)
{
    int w = p->width, h = p->height, nb = p->disp_n;
    size_t bytes = (size_t)row_elems * h * sizeof(float);
    float *tmp = malloc(bytes);
    float *cost0 = NULL;
    float beta = p->diff_beta;

    if (p->aggr_fn == SMO_MEMBRANE && beta > 0.0f && p->aggr_iter > 0) {
        cost0 = malloc(bytes);                       /* StcAggregate.cpp:262, StcDiffusion.cpp:52-56 */
        memcpy(cost0, cost, bytes);
    }
    for (int iter = 0; iter < p->aggr_iter; iter++) {
        switch (p->aggr_fn) {
        case SMO_BOX:                                /* StcAggregate.cpp:54-58 */
            smo_box_filter(cost, tmp, w, h, nb, row_elems, p->aggr_window_size, 1);
            break;
        case SMO_BINOMIAL:                           /* :287-288 */
            smo_convolve_14641(cost, tmp, w, h, nb, row_elems);
            break;
        case SMO_DIFFUSION:                          /* :305-306: beta forced to 0 */
            memcpy(tmp, cost, bytes);
            smo_diffusion_iter(tmp, cost, NULL, w, h, nb, row_elems, p->diff_lambda, 0.0f);
            break;
        case SMO_MEMBRANE:                           /* :317 */
            memcpy(tmp, cost, bytes);
            smo_diffusion_iter(tmp, cost, cost0, w, h, nb, row_elems, p->diff_lambda, beta);
            break;
        case SMO_BAYESIAN:                           /* :320-331 */
            if (!cost0)
                cost0 = malloc(bytes);
            smo_bayesian_iter(cost, tmp, cost0, w, h, nb, row_elems, p->diff_scale_cost, p->diff_mu,
                              p->diff_sigmaP, p->diff_epsP, iter);
            break;
        default:
            break;
        }
    }
    if (p->aggr_minfilter > 1)                       /* :351-360 */
        smo_min_filter(cost, tmp, w, h, nb, row_elems, p->aggr_minfilter);
    smo_pad_costs(cost, row_elems, p);               /* :365 */
    free(cost0);
    free(tmp);
}

/* StcOptimize.cpp:61-82 */
static float ncost(const uint8_t *a, const uint8_t *b, int nb, const smo_params *p)
{
    float d2 = 0.0f;
    for (int k = 0; k < nb; k++) {
        float d = (float)((int)a[k] - (int)b[k]);
        d2 += d * d;
    }
    d2 /= (float)(nb - (nb > 1));
    float da = (float)sqrt((double)d2);
    float s = p->opt_smoothness;
    if (da < p->opt_grad_thresh)
        s *= p->opt_grad_penalty;
    return s;
}

float smo_final_energy(const float *cost, long row_elems, const int32_t *disp,
                       const uint8_t *ref, long img_row_bytes, const smo_params *p)
{
    /* StcGraphCut.cpp:46-70 with the smoothness weights of StcOptimize.cpp:103-116 */
    int w = p->width, h = p->height, nb = p->n_bands, nd = p->disp_n;
    float dsum = 0.0f, nsum = 0.0f;
    for (int y = 0; y < h; y++) {
        const int32_t *cur = disp + (size_t)y * w;
        const int32_t *nxt = disp + (size_t)(y + (y < h - 1)) * w;
        const uint8_t *i0 = ref + (size_t)y * img_row_bytes;
        const uint8_t *i1 = ref + (size_t)(y + (y < h - 1)) * img_row_bytes;
        const float *dc = cost + (size_t)y * row_elems;
        for (int x = 0; x < w; x++, dc += nd) {
            int d = cur[x];
            dsum += dc[d];
            if (y < h - 1 && d != nxt[x])
                nsum += ncost(i0 + x * nb, i1 + x * nb, nb, p);
            if (x < w - 1 && d != cur[x + 1])
                nsum += ncost(i0 + x * nb, i0 + (x + 1) * nb, nb, p);
        }
    }
    return dsum + nsum;
}

void smo_depth_image(const int32_t *disp, int n, const smo_params *p, uint8_t *depth)
{
    float d_offset = (float)p->disp_min;
    float offset = (float)(-p->disp_min * p->disp_scale + 0.5);   /* StereoMatcher.cpp:177 */
    for (int i = 0; i < n; i++) {
        /* StcRefine.cpp:37-38 via Convert.cpp:31-40 (int -> float, no clip) */
        float fd;
        if (p->disp_step != 1.0f || d_offset != 0.0f)
            fd = (float)disp[i] * p->disp_step + d_offset;
        else
            fd = (float)disp[i];
        /* float -> uchar: scale/offset, clip to [0,255], truncate (Convert.cpp:33-40) */
        float v;
        if (p->disp_scale != 1.0f || offset != 0.0f)
            v = fd * p->disp_scale + offset;
        else
            v = fd;
        v = v < 0.0f ? 0.0f : v;
        v = v > 255.0f ? 255.0f : v;
        depth[i] = (uint8_t)v;
    }
}

/* StcAggregate.cpp:66-132: every level of every pixel gets the interpolated minimum of the
 * parabola through (d-1, d, d+1) or the lower half-value; c0/c1/c2 are the values before the
 * fit (the loop reads cost[d+1] before it overwrites cost[d]). */
void smo_subpixel_fit(float *cost, float *mind, float *cert, int w, int h, int nb,
                      long row_elems, float outside)
{
    for (int y = 0; y < h; y++) {
        float *c = cost + (size_t)y * row_elems;
        float *mn = mind + (size_t)y * row_elems;
        float *ce = cert + (size_t)y * row_elems;
        for (int x = 0, l = 0; x < w; x++, l += nb) {
            float c0 = c[l], c1 = c[l], c2 = c[l];
            for (int d = 0; d < nb; d++) {
                c0 = c1, c1 = c2, c2 = c[l + imin(nb - 1, d + 1)];
                mn[l + d] = 0.0f;
                ce[l + d] = 0.0f;
                if (c0 == outside || c1 == outside || c2 == outside)
                    continue;
                if (c1 <= c0 && c1 <= c2) {
                    float a = 0.5 * (c0 - 2.0 * c1 + c2);          /* :107, double arithmetic */
                    float b = 0.5 * (c2 - c0);
                    if (a <= 0.0 || a < 0.5 * fabs(b))
                        continue;
                    float dn = -0.5 * (b / a);
                    float cn = c1 + 0.5 * b * dn;
                    if (cn < 0.0)
                        continue;
                    c[l + d] = (0.0f > cn) ? 0.0f : cn;               /* __max(0.0f, cn), Image.h:50 */
                    mn[l + d] = dn;
                    ce[l + d] = a;
                } else {
                    c[l + d] = 0.5f * (c1 + (c0 < c2 ? c0 : c2));
                    mn[l + d] = (c0 < c2) ? -0.5f : 0.5f;
                    ce[l + d] = 0.0f;
                }
            }
        }
    }
}

/* StcAggregate.cpp:134-230: df fractional levels -> one integer level; mind1 NULL when
 * aggr_subpixel is off.  nb1 = levels before, nb2 = (disp_max - disp_min) + 1. */
void smo_collapse(const float *cost1, const float *mind1, int w, int h, int nb1, long row_elems1,
                  int df, int nb2, long row_elems2, float disp_step, float *cost2, float *mind2)
{
    int df2 = df / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *c1 = cost1 + (size_t)y * row_elems1 + (size_t)x * nb1;
            const float *m1 = mind1 ? mind1 + (size_t)y * row_elems1 + (size_t)x * nb1 : NULL;
            float *c2 = cost2 + (size_t)y * row_elems2 + (size_t)x * nb2;
            float *m2 = mind2 + (size_t)y * row_elems2 + (size_t)x * nb2;
            for (int d1 = 0, d2 = 0; d2 < nb2; d2++) {
                int d1_end = imin(nb1, d2 * df + df - df2);
                int d1_bst = d1;
                for (d1 = d1 + 1; d1 < d1_end; d1++)
                    if (c1[d1] < c1[d1_bst])
                        d1_bst = d1;
                c2[d2] = c1[d1_bst];
                if (m1)
                    m2[d2] = (m1[d1_bst] + d1_bst - d2 * df) * disp_step;   /* :186 */
                else
                    m2[d2] = (d1_bst - d2 * df) * disp_step;                /* :191 */
            }
        }
}

/* StcRefine.cpp:28-100.  mode 0: ScaleAndOffset only; 1: offsets from mind (:50-68);
 * 2: parabola through the winner's neighbours (:75-98). */
void smo_refine(const float *cost, const float *mind, const int32_t *disp, int w, int h, int nb,
                long row_elems, float step, int disp_min, int mode, float *fdisp)
{
    float d_offset = (float)disp_min;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            int dw = disp[i];
            /* ScaleAndOffset int -> float, Convert.cpp:31-40 */
            fdisp[i] = (step != 1.0f || d_offset != 0.0f) ? (float)dw * step + d_offset : (float)dw;
            if (mode == 0 || nb < 3)
                continue;
            size_t l = (size_t)y * row_elems + (size_t)x * nb;
            if (mode == 1) {
                float x0 = mind[l + dw];
                fdisp[i] = step * (dw + x0) + d_offset;
                continue;
            }
            int d_min = dw + (dw == 0) - (dw == nb - 1);
            float c0 = cost[l + d_min - 1], c1 = cost[l + d_min], c2 = cost[l + d_min + 1];
            float a = 0.5 * (c0 - 2.0 * c1 + c2);
            float b = 0.5 * (c2 - c0);
            if (a <= 0.0 || a < 0.5 * fabs(b))
                continue;
            float x0 = -0.5 * b / a;
            fdisp[i] = step * (d_min + x0) + d_offset;
        }
}

/* ComputeMatchQuality, StcEvaluate.cpp:760-833: cost at the (sub-pixel) winner and a certainty
 * (curvature of the parabola through the winner's neighbours), per pixel.  quality_mode is
 * eval_match_quality (2: no sub-pixel interpolation, :791-792); use_fit = aggr_subpixel (:795-800:
 * cost and certainty come from the fitted volumes; cert keeps its own row stride, the pixel stride is
 * nb as in :784); norm_bands = 1 or 3 (:774), take_sqrt = match_fn == eSD (:830).  fdisp out of range
 * would index outside the DSI in the reference; levels are clamped here (tests stay inside). */
void smo_match_quality(const float *cost, const float *cert, const float *fdisp, int w, int h, int nb,
                       long row_elems, long cert_row_elems, int disp_min, float step_inv,
                       int quality_mode, int use_fit, int norm_bands, int take_sqrt,
                       float *fcost, float *fcert)
{
    float d_offset = (float)disp_min;                                  /* :773 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            const float *c = cost + (size_t)y * row_elems + (size_t)x * nb;
            float d_new = fdisp[i];
            float d_sub = (d_new - d_offset) * step_inv;               /* :788 */
            int d = (int)(d_sub + 0.5);                                /* :789 */
            float x0 = d_sub - d;                                      /* :790 */
            if (quality_mode == 2)
                x0 = 0.0;
            d = d < 0 ? 0 : (d > nb - 1 ? nb - 1 : d);
            if (use_fit) {
                fcost[i] = c[d];                                       /* :797-799 */
                fcert[i] = cert[(size_t)y * cert_row_elems + (size_t)x * nb + d];
            } else {
                int d_min = d + (d == 0) - (d == nb - 1);              /* :805 */
                float c0 = c[d_min - 1], c1 = c[d_min], c2 = c[d_min + 1];
                float a = 0.5 * (c0 - 2.0 * c1 + c2);                  /* :809-810 */
                float b = 0.5 * (c2 - c0);
                if (a <= 0.0 || a < 0.5 * fabs(b)) {                   /* :813-817 */
                    fcost[i] = c1;
                    fcert[i] = 0.0f;
                } else {
                    float ffit = a * x0 * x0 + b * x0 + c1;            /* :822 */
                    fcost[i] = ffit;
                    fcert[i] = a;
                }
            }
            float favg = fcost[i] / norm_bands;                        /* :829-831 */
            float fnew = take_sqrt ? (float)sqrt(favg) : favg;
            fcost[i] = fnew;
        }
}

/* StereoMatcher.cpp:177 via Convert.cpp:25-53: float -> uchar, clip, truncate */
void smo_depth_from_float(const float *fdisp, int n, const smo_params *p, uint8_t *depth)
{
    float offset = (float)(-p->disp_min * p->disp_scale + 0.5);
    for (int i = 0; i < n; i++) {
        float v = fdisp[i];
        if (p->disp_scale != 1.0f || offset != 0.0f)
            v = v * p->disp_scale + offset;
        v = v < 0.0f ? 0.0f : v;
        v = v > 255.0f ? 255.0f : v;
        depth[i] = (uint8_t)v;
    }
}

/* StcPreProcess.cpp:49-55: ConvolveSeparable<uchar>(img, img, 121, 14641) iterated.  Horizontal
 * 3-tap pass to a uchar image, then vertical 5-tap pass to a uchar image (Convolve.cpp:224-225);
 * sums start at 0 and add taps in order (:100-107); replicate border (TrimIndex :42-43); float ->
 * uchar clips to [0,255] and truncates (Convert.cpp:41-45). */
void smo_preprocess_blur(uint8_t *img, long row_bytes, int w, int h, int nb, int iters)
{
    static const float k121[3] = {0.25f, 0.5f, 0.25f};   /* Convolve.cpp:289 */
    uint8_t *tmp = malloc((size_t)row_bytes * h);
    for (int it = 0; it < iters; it++) {
        for (int y = 0; y < h; y++) {
            const uint8_t *s = img + (size_t)y * row_bytes;
            uint8_t *d = tmp + (size_t)y * row_bytes;
            for (int x = 0; x < w; x++)
                for (int b = 0; b < nb; b++) {
                    float sum = 0.0f;
                    for (int l = 0; l < 3; l++)
                        sum += k121[l] * (float)s[clampi(x + l - 1, w) * nb + b];
                    float v = sum > 0.0f ? sum : 0.0f;           /* __max(src, minVal) */
                    v = v < 255.0f ? v : 255.0f;                 /* __min(.., maxVal) */
                    d[x * nb + b] = (uint8_t)v;
                }
        }
        for (int y = 0; y < h; y++) {
            uint8_t *d = img + (size_t)y * row_bytes;
            for (int i = 0; i < w * nb; i++) {
                float sum = 0.0f;
                for (int k = 0; k < 5; k++)
                    sum += k14641[k] * (float)tmp[(size_t)clampi(y + k - 2, h) * row_bytes + i];
                float v = sum > 0.0f ? sum : 0.0f;
                v = v < 255.0f ? v : 255.0f;
                d[i] = (uint8_t)v;
            }
        }
    }
    free(tmp);
}

float smo_compute_correspondence(const uint8_t *ref, const uint8_t *mtc, long img_row_bytes,
                                 const smo_params *p, int32_t *disp, float *best_cost,
                                 uint8_t *depth)
{
    return smo_compute_correspondence_f(ref, mtc, img_row_bytes, p, disp, best_cost, depth, NULL);
}

float smo_compute_correspondence_f(const uint8_t *ref, const uint8_t *mtc, long img_row_bytes,
                                   const smo_params *p, int32_t *disp, float *best_cost,
                                   uint8_t *depth, float *fdisp)
{
    int w = p->width, h = p->height, nd = p->disp_n;
    long row_elems = (long)w * nd;
    size_t n_pix = (size_t)w * h;
    float *cost = malloc((size_t)row_elems * h * sizeof(float));
    float *mind = NULL;
    int32_t *d = disp ? disp : malloc(n_pix * sizeof(int32_t));
    float *fd = fdisp ? fdisp : malloc(n_pix * sizeof(float));
    float step = p->disp_step;
    uint8_t *pre = NULL;
    if (p->preproc_blur_iter > 0) {                      /* StereoMatcher.cpp:165, on copies (:103-106) */
        size_t img = (size_t)img_row_bytes * h;
        pre = malloc(2 * img);
        memcpy(pre, ref, img);
        memcpy(pre + img, mtc, img);
        smo_preprocess_blur(pre, img_row_bytes, w, h, p->n_bands, p->preproc_blur_iter);
        smo_preprocess_blur(pre + img, img_row_bytes, w, h, p->n_bands, p->preproc_blur_iter);
        ref = pre;                                       /* the energy's smoothness term sees it too */
        mtc = pre + img;
    }

    smo_raw_costs(ref, mtc, img_row_bytes, p, cost, row_elems);
    smo_aggregate(cost, row_elems, p);
    if (p->aggr_subpixel) {                              /* StcAggregate.cpp:368-385 */
        mind = malloc((size_t)row_elems * h * sizeof(float));
        float *cert = malloc((size_t)row_elems * h * sizeof(float));
        smo_subpixel_fit(cost, mind, cert, w, h, nd, row_elems, smo_match_outside(p));
        free(cert);
    }
    if (p->aggr_collapse && step < 1.0f) {               /* :388-400, :137-146 */
        int df = p->disp_den;
        int nd2 = (nd - 1) / df + 1;                     /* (disp_max - disp_min) + 1 */
        long re2 = (long)w * nd2;
        float *cost2 = malloc((size_t)re2 * h * sizeof(float));
        float *mind2 = malloc((size_t)re2 * h * sizeof(float));
        smo_collapse(cost, mind, w, h, nd, row_elems, df, nd2, re2, p->disp_step, cost2, mind2);
        free(cost);
        free(mind);
        cost = cost2; mind = mind2; nd = nd2; row_elems = re2; step = 1.0f;
    }
    if (p->opt_fn == 3 || p->opt_fn == 2) {              /* eScanlineOpt / eDynamicProg, StcOptimize.cpp:217-238 */
        if (p->opt_fn == 3)
            smo_opt_so(cost, w, h, nd, row_elems, ref, img_row_bytes, p, d);
        else
            smo_opt_dp(cost, w, h, nd, row_elems, ref, img_row_bytes, p, d);
        if (best_cost)
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++)
                    best_cost[(size_t)y * w + x] =
                        cost[(size_t)y * row_elems + (size_t)x * nd + d[(size_t)y * w + x]];
    } else {
        smo_wta(cost, w, h, nd, row_elems, d, best_cost);
    }
    smo_params q = *p;
    q.disp_n = nd;
    float energy = smo_final_energy(cost, row_elems, d, ref, img_row_bytes, &q);
    int mode = 0;                                        /* StcRefine.cpp:42-68 */
    if (p->refine_subpix && nd >= 3)
        mode = (p->aggr_subpixel || (p->aggr_collapse && p->disp_step < 1.0f)) ? 1 : 2;
    smo_refine(cost, mind, d, w, h, nd, row_elems, step, p->disp_min, mode, fd);
    if (depth)
        smo_depth_from_float(fd, w * h, p, depth);
    if (!disp)
        free(d);
    if (!fdisp)
        free(fd);
    free(mind);
    free(cost);
    free(pre);
    return energy;
}

/* number of threads the row / column loops use: 1 for the plain build (like the reference),
 * omp_get_max_threads() for the -fopenmp build that bench.py times next to it */
int smo_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* evaluator: StcEvaluate.cpp:42-587 + ForwardWarp (Warp1D.cpp:25-128)        */

#define SMO_ROUND(x) ((int)((x) >= 0 ? (x) + .5 : (x) - .5))    /* Warp1D.cpp:23 */

/* Warp1D.cpp:25-56, one band; clip = uchar destination (0..255, round_offset 0.5) */
static void draw_line_f(const float *s1, const float *s2, float *dst, float x1, float x2, int w)
{
    if (x2 < x1)
        return;
    int i0 = imax(0, SMO_ROUND(x1));
    int i1 = imin(w - 1, SMO_ROUND(x2));
    float iden = 1.0 / (x2 - x1 + (x2 == x1));
    for (int i = i0; i <= i1; i++) {
        float f = (i - x1) * iden;
        float v1 = s1[0], v2 = s2[0];
        float v = v1 + f * (v2 - v1);
        dst[i] = (float)(v + 0.0f);
    }
}

static void draw_line_u8(const uint8_t *s1, const uint8_t *s2, uint8_t *dst, float x1, float x2, int w)
{
    if (x2 < x1)
        return;
    int i0 = imax(0, SMO_ROUND(x1));
    int i1 = imin(w - 1, SMO_ROUND(x2));
    float iden = 1.0 / (x2 - x1 + (x2 == x1));
    for (int i = i0; i <= i1; i++) {
        float f = (i - x1) * iden;
        float v1 = s1[0], v2 = s2[0];
        float v = v1 + f * (v2 - v1);
        float r = v + 0.5f;
        r = r > 0 ? r : 0;                           /* __max(v + round_offset, minVal) */
        r = r < 255 ? r : 255;                       /* __min(.., maxVal) */
        dst[i] = (uint8_t)r;
    }
}

/* Warp1D.cpp:58-128 for one row; `is_u8` selects the uchar instantiation */
static void forward_warp_row(const void *src, void *dst, const float *dp, int w, float d_scale,
                             float disp_gap, int is_u8)
{
    int xstart = d_scale > 0 ? 0 : w - 1, xend = d_scale > 0 ? w : -1, xincr = d_scale > 0 ? 1 : -1;
    const float *psf = src;
    float *pdf = dst;
    const uint8_t *psu = src;
    uint8_t *pdu = dst;
    for (int x = xstart; x != xend; x += xincr) {
        int x2 = x + xincr;
        int draw = 0;
        if (x2 != xend) {                            /* dp[x2] is only read inside the row */
            float d_diff = fabsf(dp[x] - dp[x2]);
            draw = d_diff < disp_gap;
        }
        float d = d_scale * dp[x];
        if (draw) {
            float d2 = d_scale * dp[x2];
            if (xincr > 0) {
                if (is_u8) draw_line_u8(&psu[x], &psu[x2], pdu, x - d, x2 - d2, w);
                else       draw_line_f(&psf[x], &psf[x2], pdf, x - d, x2 - d2, w);
            } else {                                 /* (x2 - d, x - d2) as written at :115-117 */
                if (is_u8) draw_line_u8(&psu[x2], &psu[x], pdu, x2 - d, x - d2, w);
                else       draw_line_f(&psf[x2], &psf[x], pdf, x2 - d, x - d2, w);
            }
            continue;
        }
        int xx = x - SMO_ROUND(d);
        if (xx >= 0 && xx < w) {
            if (is_u8) pdu[xx] = psu[x];
            else       pdf[xx] = psf[x];
        }
    }
}

static void evaluate_impl(const float *fdisp, const uint8_t *truth, const uint8_t *ref, long img_row_bytes,
                          const smo_params *p, const smo_eval *e, int verbose_summary, smo_eval_out *out,
                          uint8_t *occ_out, uint8_t *texl_out)
{
    int w = p->width, h = p->height, nB = p->n_bands;
    size_t n = (size_t)w * h;
    float *trud = malloc(n * sizeof(float));
    float *fwd = malloc(n * sizeof(float));
    uint8_t *white = malloc(n), *occ = calloc(n, 1), *texl = malloc(n), *disc = malloc(n);
    float *sq = calloc(n, sizeof(float)), *tmp = malloc(n * sizeof(float));
    float *dd = calloc(n, sizeof(float));

    /* StereoMatcher.cpp:125: m_true_disparity = truth / disp_scale + disp_min (Convert.cpp:31-40) */
    {
        float scale = 1.0f / p->disp_scale, offset = (float)p->disp_min;
        for (size_t i = 0; i < n; i++)
            trud[i] = (scale != 1.0f || offset != 0.0f) ? truth[i] * scale + offset : (float)truth[i];
    }

    /* ComputeOcclusion(frame_match), StcEvaluate.cpp:42-79 */
    float shift = (float)p->frame_diff_sign;         /* (frame_match - frame_ref) / |m_frame_diff| */
    const float invalid_depth = -9999.0f;
    for (size_t i = 0; i < n; i++)
        fwd[i] = invalid_depth;
    for (int y = 0; y < h; y++)
        forward_warp_row(trud + (size_t)y * w, fwd + (size_t)y * w, trud + (size_t)y * w, w, shift,
                         e->disp_gap, 0);
    for (size_t i = 0; i < n; i++)
        white[i] = (fwd[i] == invalid_depth) ? 0 : 255;
    for (int y = 0; y < h; y++)
        forward_warp_row(white + (size_t)y * w, occ + (size_t)y * w, fwd + (size_t)y * w, w, -shift,
                         e->disp_gap, 1);
    /* ComputeOcclusions, :86-107: black borders, then occlusion = 255 - visibility */
    int b = e->ignore_border;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t *o = &occ[(size_t)y * w + x];
            if (x < b || x >= w - b || y < b || y >= h - b)
                *o = 0;
            *o = (uint8_t)(*o * -1.0f + 255.0f);
        }

    /* ComputeTextureless, :110-157 */
    {
        int nC = nB - (nB > 1);
        for (int y = 0; y < h; y++) {
            const uint8_t *q = ref + (size_t)y * img_row_bytes;
            float *s = sq + (size_t)y * w;
            for (int x = 0; x < w - 1; x++, q += nB) {
                float sum2 = 0;
                for (int c = 0; c < nC; c++) {
                    float diff = (int)q[c] - (int)q[c + nB];
                    sum2 += diff * diff;
                }
                sum2 /= (float)nC;
                s[x + 1] = sum2;
                if (x == 0)
                    s[x] = sum2;
                s[x] = sum2 > s[x] ? sum2 : s[x];
            }
        }
        if (e->textureless_width > 0)
            smo_box_filter(sq, tmp, w, h, 1, w, e->textureless_width, 1);
        float thr = e->textureless_thresh * e->textureless_thresh;
        for (size_t i = 0; i < n; i++)
            texl[i] = (sq[i] < thr) ? 255 : 0;
    }

    /* ComputeDisparityDiscont, :190-244 (int image; sums of 255s are exact in float) */
    {
        int bor = e->ignore_border + 1;
        for (int y = 0; y < h - 1; y++) {
            if (y < bor || y >= h - bor)
                continue;
            const float *p0 = trud + (size_t)y * w, *p1 = p0 + w;
            float *d0 = dd + (size_t)y * w, *d1 = d0 + w;
            for (int x = bor; x < w - bor - 1; x++) {
                if (fabsf(p0[x] - p0[x + 1]) >= e->disp_gap)
                    d0[x] = d0[x + 1] = 255;
                if (fabsf(p0[x] - p1[x]) >= e->disp_gap)
                    d0[x] = d1[x] = 255;
            }
        }
        if (e->discont_width > 0)
            smo_box_filter(dd, tmp, w, h, 1, w, e->discont_width, 0);
        for (size_t i = 0; i < n; i++)
            disc[i] = (dd[i] != 0) ? 255 : 0;
    }

    /* ComputeDisparityErrors, :373-560 (m_status is empty after WTA: all pixels count) */
    float ssd[6] = {0, 0, 0, 0, 0, 0};               /* all, nonocc, occ, textured, textureless, discont */
    int bad[6] = {0, 0, 0, 0, 0, 0}, cnt[6] = {0, 0, 0, 0, 0, 0};
    int total_all = 0;
    for (int y = b; y < h - b; y++)
        for (int x = b; x < w - b; x++) {
            size_t i = (size_t)y * w + x;
            float diff = fdisp[i] - trud[i];
            total_all++;
            int is_bad = fabs(diff) > e->bad_thresh;
#define SMO_ACC(k) do { ssd[k] += diff * diff; bad[k] += is_bad; cnt[k]++; } while (0)
            SMO_ACC(0);
            if (occ[i] > 1) {
                SMO_ACC(2);
            } else {
                SMO_ACC(1);
                if (texl[i] != 0)
                    SMO_ACC(4);
                else
                    SMO_ACC(3);
                if (disc[i] != 0)
                    SMO_ACC(5);
            }
#undef SMO_ACC
        }
    cnt[0] += (cnt[0] == 0);
    total_all += (total_all == 0);
    out->fraction_matched = (float)cnt[0] / (float)total_all;
    for (int k = 0; k < 6; k++) {
        cnt[k] += (cnt[k] == 0);
        out->rms[k] = sqrtf(ssd[k] / (float)cnt[k]);
        out->bad[k] = (float)bad[k] / (float)cnt[k];
    }
    if (verbose_summary)
        out->fraction_matched *= 0.1f;               /* :583 */
    if (occ_out) memcpy(occ_out, occ, n);
    if (texl_out) memcpy(texl_out, texl, n);
    free(trud); free(fwd); free(white); free(occ); free(texl); free(disc); free(sq); free(tmp); free(dd);
}

void smo_evaluate(const float *fdisp, const uint8_t *truth, const uint8_t *ref, long img_row_bytes,
                  const smo_params *p, const smo_eval *e, int verbose_summary, smo_eval_out *out)
{
    evaluate_impl(fdisp, truth, ref, img_row_bytes, p, e, verbose_summary, out, NULL, NULL);
}

/* the same evaluation, also handing out m_occlusion (255 = occluded or border) and m_textureless */
void smo_evaluate_masks(const float *fdisp, const uint8_t *truth, const uint8_t *ref, long img_row_bytes,
                        const smo_params *p, const smo_eval *e, uint8_t *occlusion, uint8_t *textureless)
{
    smo_eval_out out;
    evaluate_impl(fdisp, truth, ref, img_row_bytes, p, e, 0, &out, occlusion, textureless);
}

/* Histogram1D drawn as a figure (Histogram1D.cpp:32-139) for the arguments ComputeMatchQuality gives it
 * (StcEvaluate.cpp:848-861): one-band float image, fixed value range, n_bins > 0, height > 0.  mask may be NULL;
 * the first zero of the mask ends the scan of its row (:86-87 `break`).  *v_scale <= 0: found from the tallest
 * bin and handed back (:118-119, :855).  figure: n_bins x height bytes. */
void smo_cost_histogram(const float *img, const uint8_t *mask, int w, int h, float min_val, float max_val,
                        int n_bins, int height, float *v_scale, uint8_t *figure)
{
    float step_size = (max_val - min_val) / (float)n_bins;
    float inv_step_size = 1.0f / step_size;
    int *counts = calloc((size_t)n_bins, sizeof(int));
    for (int y = 0; y < h; y++) {
        const float *q = img + (size_t)y * w;
        const uint8_t *m = mask ? mask + (size_t)y * w : NULL;
        for (int x = 0; x < w; x++) {
            if (m && m[x] == 0)
                break;
            int bin = (int)((q[x] - min_val) * inv_step_size);
            bin = imin(imax(0, bin), n_bins - 1);
            counts[bin] += 1;
        }
    }
    int max_count = 0;
    for (int i = 0; i < n_bins; i++)
        max_count = imax(max_count, counts[i]);
    if (*v_scale <= 0)
        *v_scale = height / (float)max_count;
    for (int i = 0; i < n_bins; i++)
        counts[i] = imin(height - 1, (int)(counts[i] * *v_scale + 0.5));
    for (int y = 0; y < height; y++) {
        int y2 = height - 1 - y;
        for (int x = 0; x < n_bins; x++)
            figure[(size_t)y * n_bins + x] = (y2 >= counts[x]) ? 255 : 0;
    }
    free(counts);
}

/* ------------------------------------------------------------------------- */
/* prediction (re-projection) error of one frame: StcEvaluate.cpp:588-758,      */
/* ForwardWarp / InverseWarp of Warp1D.cpp:25-257, Min/MaxFilter<uchar> 3x3     */
/* (MinFilter.cpp:122-202).  No output of the reference pins these numbers:     */
/* "parity unpinned" (tests check known answers and the host driver against it) */

/* draw_intensity_line (Warp1D.cpp:25-56), uchar, all bands of a pixel */
static void draw_line_u8n(const uint8_t *s1, const uint8_t *s2, uint8_t *dst, float x1, float x2,
                          int w, int nb)
{
    if (x2 < x1)
        return;
    int i0 = imax(0, SMO_ROUND(x1));
    int i1 = imin(w - 1, SMO_ROUND(x2));
    float iden = 1.0 / (x2 - x1 + (x2 == x1));
    for (int i = i0; i <= i1; i++) {
        float f = (i - x1) * iden;
        for (int b = 0; b < nb; b++) {
            float v1 = s1[b], v2 = s2[b];
            float v = v1 + f * (v2 - v1);
            float r = v + 0.5f;
            r = r > 0 ? r : 0;
            r = r < 255 ? r : 255;
            dst[(size_t)i * nb + b] = (uint8_t)r;
        }
    }
}

/* ForwardWarp<uchar> (Warp1D.cpp:58-128), one row of an nb-band image */
static void forward_warp_row_u8n(const uint8_t *ps, uint8_t *pd, const float *dp, int w, int nb,
                                 float d_scale, int line_interpolate, float disp_gap)
{
    int xstart = d_scale > 0 ? 0 : w - 1, xend = d_scale > 0 ? w : -1, xincr = d_scale > 0 ? 1 : -1;
    for (int x = xstart; x != xend; x += xincr) {
        int x2 = x + xincr;
        int draw = 0;
        if (line_interpolate && x2 != xend)
            draw = fabsf(dp[x] - dp[x2]) < disp_gap;
        float d = d_scale * dp[x];
        if (draw) {
            float d2 = d_scale * dp[x2];
            if (xincr > 0)
                draw_line_u8n(&ps[(size_t)x * nb], &ps[(size_t)x2 * nb], pd, x - d, x2 - d2, w, nb);
            else                                     /* (x2 - d, x - d2) as written at :115-117 */
                draw_line_u8n(&ps[(size_t)x2 * nb], &ps[(size_t)x * nb], pd, x2 - d, x - d2, w, nb);
            continue;
        }
        int xx = x - SMO_ROUND(d);
        if (xx >= 0 && xx < w)
            memcpy(&pd[(size_t)xx * nb], &ps[(size_t)x * nb], nb);
    }
}

/* InverseWarpLine (Warp1D.cpp:144-194) */
static void inverse_warp_line(const float *src, float *dst, const float *disp, int w, int nb, int order,
                              const float *fwd_disp, float disp_gap)
{
    for (int x = 0; x < w; x++, dst += nb) {
        float d = disp[x];
        float y = x - d;
        if (y < 0.0 || y > w - 1)
            continue;
        int xx = (int)y;
        if (fwd_disp && disp_gap && fabs(d - fwd_disp[xx]) >= disp_gap)
            continue;
        const float *ps0 = &src[(size_t)xx * nb];
        if (order == 0 || xx == y) {
            memcpy(dst, ps0, nb * sizeof(float));
        } else if (order == 1 || xx - 1 < 0 || xx + 2 > w - 1) {
            float f = y - xx;
            const float *ps1 = &ps0[nb];
            for (int b = 0; b < nb; b++)
                dst[b] = ps0[b] + f * (ps1[b] - ps0[b]);
        } else {                                     /* order 3 */
            float f = y - xx;
            const float *psp = &ps0[-nb], *ps1 = &ps0[nb], *ps2 = &ps0[2 * nb];
            for (int b = 0; b < nb; b++)
                dst[b] = cubic_interp(f, psp[b], ps0[b], ps1[b], ps2[b]);
        }
    }
}

/* 3x3 minimum (do_max = 0) or maximum of every band, replicate border: what MinMaxFilter
 * (MinFilter.cpp:122-202) computes for xWidth = yWidth = 3 (pl = -2, pr = 1: cells x-1 .. x+1) */
static void minmax3_u8(const uint8_t *src, uint8_t *dst, int w, int h, int nb, int do_max)
{
    size_t row = (size_t)w * nb;
    uint8_t *tmp = malloc(row * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int b = 0; b < nb; b++) {
                const uint8_t *r = src + y * row;
                uint8_t a = r[(size_t)clampi(x - 1, w) * nb + b], c = r[(size_t)x * nb + b],
                        e = r[(size_t)clampi(x + 1, w) * nb + b];
                uint8_t m = do_max ? (a > c ? a : c) : (a < c ? a : c);
                tmp[y * row + (size_t)x * nb + b] = do_max ? (m > e ? m : e) : (m < e ? m : e);
            }
    for (int y = 0; y < h; y++)
        for (size_t i = 0; i < row; i++) {
            uint8_t a = tmp[clampi(y - 1, h) * row + i], c = tmp[y * row + i], e = tmp[clampi(y + 1, h) * row + i];
            uint8_t m = do_max ? (a > c ? a : c) : (a < c ? a : c);
            dst[y * row + i] = do_max ? (m > e ? m : e) : (m < e ? m : e);
        }
    free(tmp);
}

/* PartialShuffle (StcEvaluate.cpp:588-614) */
static void partial_shuffle(const uint8_t *img, uint8_t *mn, uint8_t *mx, int w, int h, int nb, float amt)
{
    minmax3_u8(img, mn, w, h, nb, 0);
    minmax3_u8(img, mx, w, h, nb, 1);
    size_t n = (size_t)w * h * nb;
    for (size_t i = 0; i < n; i++) {
        mn[i] = (int)(img[i] + amt * (mn[i] - (int)img[i]));              /* round down */
        mx[i] = (int)(img[i] + amt * (mx[i] - (int)img[i]) + 0.99f);      /* round up */
    }
}

/* ComputePredictionError (StcEvaluate.cpp:615-690); `predicted` is rewritten when predict_diff != 0 */
static float prediction_error(uint8_t *predicted, const uint8_t *original, int w, int h, int nB,
                              const smo_predict *q, float *fraction_visible)
{
    int nC = nB - (nB > 1);
    float sum2 = 0.0;
    int visible = 0, total = w * h;
    size_t n = (size_t)w * h * nB;
    int shuffle = q->partial_shuffle > 0.0f;
    uint8_t *pn = NULL, *px = NULL, *on = NULL, *ox = NULL;
    if (shuffle) {
        pn = malloc(n); px = malloc(n); on = malloc(n); ox = malloc(n);
        partial_shuffle(predicted, pn, px, w, h, nB, q->partial_shuffle);
        partial_shuffle(original, on, ox, w, h, nB, q->partial_shuffle);
    }
    for (int y = 0; y < h; y++) {
        uint8_t *p = predicted + (size_t)y * w * nB;
        const uint8_t *o = original + (size_t)y * w * nB;
        for (int x = 0; x < w; x++, p += nB, o += nB) {
            if (nB > 1 && p[nC] != 255)              /* not visible (alpha != 255) */
                continue;
            visible++;
            for (int b = 0; b < nC; b++) {
                float diff = (int)p[b] - (int)o[b];
                if (shuffle) {
                    size_t i = ((size_t)y * w + x) * nB + b;
                    int xn = imax(pn[i], on[i]), nx = imin(px[i], ox[i]);
                    if (xn <= nx)
                        diff = 0;
                    else
                        diff = (pn[i] > ox[i]) ? pn[i] - ox[i] : on[i] - px[i];
                }
                sum2 += diff * diff;
                if (q->predict_diff && (nB == 1 || b < nB - 1))
                    p[b] = imax(0, imin(255, 128 + (int)(diff * q->predict_diff)));
            }
        }
    }
    free(pn); free(px); free(on); free(ox);
    *fraction_visible = visible / (float)total;
    return sqrtf(sum2 / nC / (visible + (visible == 0)));
}

/* One frame of ComputePredictionErrors (StcEvaluate.cpp:692-758) */
float smo_prediction_error(const uint8_t *ref, const uint8_t *frame, int w, int h, int nb,
                           const float *fdisp, float shift, const smo_predict *q,
                           uint8_t *resampled, float *fraction_visible)
{
    size_t row = (size_t)w * nb;
    /* fill with the empty colour: the low byte for gray, the int's bytes for 4 bands (:704-717) */
    for (size_t i = 0; i < (size_t)w * h; i++) {
        if (nb == 1) {
            resampled[i] = (uint8_t)q->empty_color;
        } else {
            int32_t c = q->empty_color;
            memcpy(&resampled[i * 4], &c, 4);
        }
    }
    if (q->predict_type == 1) {                      /* ePredictForward: the reference image moved */
        for (int y = 0; y < h; y++)
            forward_warp_row_u8n(ref + y * row, resampled + y * row, fdisp + (size_t)y * w, w, nb, shift,
                                 q->lin_interp != 0, q->disp_gap);
        return prediction_error(resampled, frame, w, h, nb, q, fraction_visible);
    }
    /* ePredictInverse: InverseWarp (Warp1D.cpp:196-257), cubic, with the visibility test against
     * the forward-warped depth map */
    float *fwd = NULL, *fwd_tmp = NULL;
    if (q->disp_gap > 0.0f) {
        fwd = malloc((size_t)w * h * sizeof(float));
        fwd_tmp = malloc((size_t)w * sizeof(float));
        for (size_t i = 0; i < (size_t)w * h; i++)
            fwd[i] = -9999.0f;
        for (int y = 0; y < h; y++) {
            const float *dp = fdisp + (size_t)y * w;
            for (int x = 0; x < w; x++)              /* ScaleAndOffset(disp, fwd_tmp, d_scale, 0) */
                fwd_tmp[x] = (shift != 1.0f) ? dp[x] * shift + 0.0f : dp[x];
            forward_warp_row(fwd_tmp, fwd + (size_t)y * w, dp, w, shift, q->disp_gap, 0);
        }
    }
    float *src_buf = malloc(row * sizeof(float)), *dst_buf = malloc(row * sizeof(float));
    float *dsp_buf = malloc((size_t)w * sizeof(float));
    for (int y = 0; y < h; y++) {
        const uint8_t *ps = frame + y * row;
        uint8_t *pd = resampled + y * row;
        const float *dp = fdisp + (size_t)y * w;
        for (size_t i = 0; i < row; i++) {
            src_buf[i] = ps[i];
            dst_buf[i] = pd[i];
        }
        for (int x = 0; x < w; x++)
            dsp_buf[x] = (shift != 1.0f) ? dp[x] * shift + 0.0f : dp[x];
        inverse_warp_line(src_buf, dst_buf, dsp_buf, w, nb, 3, fwd ? fwd + (size_t)y * w : NULL, q->disp_gap);
        for (size_t i = 0; i < row; i++) {           /* back to uchar: + 0.5, clipped, truncated */
            float val = dst_buf[i] * 1.0f + 0.5f;
            val = val > 0 ? val : 0;
            val = val < 255 ? val : 255;
            pd[i] = (uint8_t)val;
        }
    }
    free(src_buf); free(dst_buf); free(dsp_buf); free(fwd); free(fwd_tmp);
    return prediction_error(resampled, ref, w, h, nb, q, fraction_visible);
}

/* ------------------------------------------------------------------------- */
/* scan-line optimisation: StcOptSO.cpp:30-154 with the horizontal smoothness */
/* weights of ComputeSmoothnessCosts (StcOptimize.cpp:84-116, band 1)          */

#define SMO_COST_MAX 1000000   /* StereoMatcher.h:57 */

void smo_opt_so(const float *cost, int w, int h, int nd, long row_elems, const uint8_t *ref,
                long img_row_bytes, const smo_params *p, int32_t *disp)
{
    float *sumcost = malloc((size_t)w * nd * sizeof(float));
    int *trans = malloc((size_t)w * nd * sizeof(int));
    int nb = p->n_bands;
    for (int y = 0; y < h; y++) {
        const float *c = cost + (size_t)y * row_elems;
        const uint8_t *I = ref + (size_t)y * img_row_bytes;
        for (int d = 0; d < nd; d++) {               /* :56-60 */
            trans[d] = -1;
            sumcost[d] = c[d];
        }
        for (int x = 1; x < w; x++) {                /* :71-111 */
            /* m_smooth(x-1, y, 1): weight between pixels x-1 and x (x-1 < W-1 always holds here) */
            float smooth = ncost(I + (x - 1) * nb, I + x * nb, nb, p);
            const float *prev = sumcost + (size_t)(x - 1) * nd;
            for (int d = 0; d < nd; d++) {
                float best_cost = SMO_COST_MAX;
                int best_d1 = -1;
                for (int d1 = 0; d1 < nd; d1++) {
                    float cc = prev[d1];
                    if (d1 != d)
                        cc += smooth;
                    if (cc < best_cost) {
                        best_cost = cc;
                        best_d1 = d1;
                    }
                }
                sumcost[(size_t)x * nd + d] = best_cost + c[(size_t)x * nd + d];
                trans[(size_t)x * nd + d] = best_d1;
            }
        }
        int x = w - 1;                               /* :121-133 */
        float best_cost = SMO_COST_MAX;
        int best_d = 0;
        for (int d = 0; d < nd; d++)
            if (sumcost[(size_t)x * nd + d] < best_cost) {
                best_cost = sumcost[(size_t)x * nd + d];
                best_d = d;
            }
        int d = best_d;
        while (x >= 0) {                             /* :144-151 */
            disp[(size_t)y * w + x] = d;
            d = trans[(size_t)x * nd + d];
            if (d < 0)
                d = 0;   /* only at x == 0 unless a path runs through COST_MAX (the reference asserts) */
            x--;
        }
    }
    free(trans);
    free(sumcost);
}

/* ------------------------------------------------------------------------- */
/* dynamic programming: StcOptDP.cpp:110-332.  Per scan-line, every cell (x, d) */
/* has the states M (matched), L, R (occluded) with cumulative costs; seven     */
/* transitions (:137-149), candidates tried in the order t = 0..6 with a strict */
/* '<' against the state's current value, which starts at COST_MAX (:188-229).  */
/* Predecessors: t 0,1,6 = (x-1, d), t 2,3 = (x-1, d-1), t 4,5 = (x, d+1) --    */
/* the column is filled from d = D-1 down, so the 'up' cell is already final.   */

void smo_opt_dp(const float *cost, int w, int h, int nd, long row_elems, const uint8_t *ref,
                long img_row_bytes, const smo_params *p, int32_t *disp)
{
    static const int cstate[7] = {0, 1, 1, 0, 2, 2, 0};     /* :137-138 */
    static const int pstate[7] = {0, 0, 1, 1, 0, 2, 2};
    static const int dx[7] = {-1, -1, -1, -1, 0, 0, -1};    /* :141-145 as (x, d) steps */
    static const int dd[7] = {0, 0, -1, -1, 1, 1, 0};
    const int occ = -9999;                                  /* :119 */
    const float ocL = (float)p->opt_occlusion_cost, ocR = ocL;   /* :116-117 */
    int nb = p->n_bands;
    float *sum = malloc((size_t)w * nd * 3 * sizeof(float));
    signed char *tr = malloc((size_t)w * nd * 3);
    for (int y = 0; y < h; y++) {
        const float *c = cost + (size_t)y * row_elems;
        const uint8_t *I = ref + (size_t)y * img_row_bytes;
        int32_t *out = disp + (size_t)y * w;
        for (int d = 0; d < nd; d++) {                      /* :162-170 */
            float *s = sum + (size_t)d * 3;
            signed char *t = tr + (size_t)d * 3;
            t[0] = 0; t[1] = -1; t[2] = -1;
            s[0] = c[d]; s[1] = SMO_COST_MAX; s[2] = SMO_COST_MAX;
        }
        for (int x = 1; x < w; x++) {                       /* :179-236 */
            float smooth = ncost(I + (x - 1) * nb, I + x * nb, nb, p);   /* m_smooth(x-1, y, 1), :174 */
            const float *cx = c + (size_t)x * nd;
            for (int d = nd - 1; d >= 0; d--) {
                float *s = sum + ((size_t)x * nd + d) * 3;
                signed char *t = tr + ((size_t)x * nd + d) * 3;
                s[0] = s[1] = s[2] = SMO_COST_MAX;
                t[0] = t[1] = t[2] = -1;
                for (int k = 0; k < 7; k++) {
                    if ((d == 0 && dd[k] < 0) || (d == nd - 1 && dd[k] > 0))   /* :148-149, :200-203 */
                        continue;
                    int cs = cstate[k];
                    float cinc = cs == 1 ? ocL : (cs == 2 ? ocR : cx[d]);
                    if (k == 3 || k == 6)
                        cinc += smooth;
                    float cand = sum[((size_t)(x + dx[k]) * nd + (d + dd[k])) * 3 + pstate[k]] + cinc;
                    if (cand < s[cs]) {
                        s[cs] = cand;
                        t[cs] = (signed char)k;
                    }
                }
            }
        }
        for (int x = 0; x < w; x++)
            out[x] = occ;                                   /* (only seen if the walk below stops early) */
        int x = w - 1, d = 0, st = 0;                       /* :256-274: matched cells of the last column */
        float best = SMO_COST_MAX;
        for (int k = 0; k < nd; k++)
            if (sum[((size_t)x * nd + k) * 3] < best) {
                best = sum[((size_t)x * nd + k) * 3];
                d = k;
            }
        while (x >= 0) {                                    /* :292-322 */
            int k = tr[((size_t)x * nd + d) * 3 + st];
            if (k < 0)
                break;   /* a path through COST_MAX: the reference indexes cstate[-1] here */
            out[x] = cstate[k] == 0 ? d : occ;
            st = pstate[k];
            x += dx[k];
            d += dd[k];
        }
        /* fill_occluded_pixels (:78-107), left to right: holes take the last matched value to
         * their left, leading holes the first matched value of the line */
        for (x = 0; x < w; x++)
            if (out[x] != occ)
                break;
        int oldd = x < w ? out[x] : 0;                      /* (the last pixel is always matched) */
        for (x = 0; x < w; x++) {
            if (out[x] == occ)
                out[x] = oldd;
            else
                oldd = out[x];
        }
    }
    free(tr);
    free(sum);
}

/* ------------------------------------------------------------------------- */
/* Bayesian diffusion: StcDiffusion.cpp:139-308 (one iteration)               */
/* g++ resolves exp(float) to the float overload, so the float-argument calls */
/* (:214, :236) are expf here; log (:253) takes a double.  TOLERANCE MODE:    */
/* the device's expf / log differ from libm's in the last place.              */

void smo_bayesian_iter(float *cost, float *cost2, float *cost0, int w, int h, int nd, long row_elems,
                       float scale_cost, float mu, float sigmaP, float epsP, int iter)
{
    if (iter == 0) {                                 /* :182-191 */
        for (int y = 0; y < h; y++)
            for (long i = 0; i < (long)w * nd; i++) {
                float *c = cost + (size_t)y * row_elems + i;
                *c = *c * scale_cost + 0.0f;
                cost0[(size_t)y * row_elems + i] = *c;
            }
    }
    double *p = malloc((size_t)nd * sizeof(double));
    double *wt = malloc((size_t)nd * nd * sizeof(double));
    for (int d = 0; d < nd; d++) {                   /* :209-222 */
        float s = 0.0;
        for (int d2 = 0; d2 < nd; d2++) {
            float diff = (d - d2);
            s += (wt[(size_t)d * nd + d2] = ((1 - epsP) * expf(-diff * diff / (2 * sigmaP * sigmaP)) + epsP));
        }
        for (int d2 = 0; d2 < nd; d2++)
            wt[(size_t)d * nd + d2] /= s;
    }
    for (int y = 0; y < h; y++)                      /* :226-256 */
        for (int x = 0; x < w; x++) {
            const float *pE = cost + (size_t)y * row_elems + (size_t)x * nd;
            float *pEs = cost2 + (size_t)y * row_elems + (size_t)x * nd;
            double s = 0.;
            for (int d = 0; d < nd; d++)
                s += (p[d] = expf(-pE[d]));
            if (s == 0) {
                for (int d = 0; d < nd; d++)
                    p[d] = 1.0 / nd;
            } else {
                for (int d = 0; d < nd; d++)
                    p[d] /= s;
            }
            for (int d = 0; d < nd; d++) {
                double ps = 0;
                for (int d2 = 0; d2 < nd; d2++)
                    ps += (wt[(size_t)d * nd + d2] * p[d2]);
                pEs[d] = -log(1e-16 > ps ? 1e-16 : ps);
            }
        }
    long dx = nd, dy = row_elems;                    /* :261-307 */
    for (int y = 0; y < h; y++) {
        float *pE = cost + (size_t)y * row_elems;
        const float *pEs = cost2 + (size_t)y * row_elems;
        const float *pE0 = cost0 + (size_t)y * row_elems;
        for (long x = 0; x < dx * w; x++) {
            long left = (x - dx >= 0 ? x - dx : x);
            long right = (x + dx < dx * w ? x + dx : x);
            long up = (y > 0 ? x - dy : x);
            long down = (y < h - 1 ? x + dy : x);
            pE[x] = pE0[x] + mu * (pEs[x] + pEs[left] + pEs[right] + pEs[up] + pEs[down]);
        }
    }
    free(wt);
    free(p);
}
