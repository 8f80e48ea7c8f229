// StcEvaluate.cpp -- the evaluator behind `depth_map`: occlusion, textureless and
// depth-discontinuity masks from the ground truth, then RMS / bad-pixel statistics of
// m_float_disparity (reference: StcEvaluate.cpp:42-587, ForwardWarp Warp1D.cpp:25-128).
// These are the rms_error_* / bad_pixels_* lines of results/*.txt, the reference's only quality
// read-out.  O(W*H) host work; the two box filters (BoxFilter<float> 3x3 average over the
// squared gradient, BoxFilter<int> 9x9 sum over the discontinuity marks) run through
// smhip_box_filter on one-band images -- the same operator that aggregates the DSI.
// Prediction (re-projection) errors (:588-758, ForwardWarp / InverseWarp Warp1D.cpp:58-257) fill
// predict_err_* and the frames' resampled images.  Not restated: match quality (:760-863) and the
// status statistics (:865-954; WTA leaves no status map).
#include "StereoMatcher.h"
#include <algorithm>
#include <vector>

#include "ImageIO.h"
#include "Error.h"
#include <cmath>
#include <vector>

#define ROUND(x) ((int)((x) >= 0 ? (x) + .5 : (x) - .5))     // Warp1D.cpp:23

namespace {

// draw_intensity_line (Warp1D.cpp:25-56), all bands of a pixel.  T = float: no clipping, no
// rounding offset; T = uchar: + 0.5, clipped to [0, 255], truncated.
template <class T>
void DrawLine(const T *src1, const T *src2, T *dst, float x1, float x2, int w, int nb)
{
    if (x2 < x1)
        return;                                               // backward facing line
    const bool is_float = sizeof(T) == sizeof(float);
    int i0 = std::max(0, ROUND(x1));
    int i1 = std::min(w - 1, ROUND(x2));
    float iden = 1.0 / (x2 - x1 + (x2 == x1));
    for (int i = i0; i <= i1; i++) {
        float f = (i - x1) * iden;
        T *dp = dst + (size_t)i * nb;
        for (int b = 0; b < nb; b++) {
            float v1 = src1[b], v2 = src2[b];
            float v = v1 + f * (v2 - v1);
            if (is_float) {
                dp[b] = (T)(v + 0.0f);
            } else {
                float r = v + 0.5f;
                r = r > 0 ? r : 0;
                r = r < 255 ? r : 255;
                dp[b] = (T)r;
            }
        }
    }
}

// ForwardWarp (Warp1D.cpp:58-128): pixels of src move by -d_scale * disp; pixels of dst nothing
// lands on keep their value
template <class T>
void ForwardWarp(const CImageOf<T> &src, CImageOf<T> &dst, const CFloatImage &disp, float d_scale,
                 bool line_interpolate, float disp_gap)
{
    CShape sh = src.Shape();
    int w = sh.width, h = sh.height, nb = sh.nBands;
    for (int y = 0; y < h; y++) {
        int xstart = d_scale > 0 ? 0 : w - 1, xend = d_scale > 0 ? w : -1, xincr = d_scale > 0 ? 1 : -1;
        const float *dp = disp.Row(y);
        const T *ps = src.Row(y);
        T *pd = dst.Row(y);
        for (int x = xstart; x != xend; x += xincr) {
            int x2 = x + xincr;
            bool draw_line = line_interpolate && (x2 != xend) && (std::fabs(dp[x] - dp[x2]) < disp_gap);
            float d = d_scale * dp[x];
            if (draw_line) {
                float d2 = d_scale * dp[x2];
                if (xincr > 0)
                    DrawLine(&ps[(size_t)x * nb], &ps[(size_t)x2 * nb], pd, x - d, x2 - d2, w, nb);
                else                                          // as written at Warp1D.cpp:115-117
                    DrawLine(&ps[(size_t)x2 * nb], &ps[(size_t)x * nb], pd, x2 - d, x - d2, w, nb);
                continue;
            }
            int xx = x - ROUND(d);
            if (xx >= 0 && xx < w)
                memcpy(&pd[(size_t)xx * nb], &ps[(size_t)x * nb], nb * sizeof(T));
        }
    }
}

// CubicInterpolate (Warp1D.cpp:130-142)
inline float CubicInterpolate(float x0, float v0, float v1, float v2, float v3)
{
    float x1 = 1.0f - x0;
    float s0 = v2 - v0, s1 = v1 - v3, d1 = v2 - v1;
    float phi0 = d1 * (x0 * x0) * (2.0f * x1 + 1.0f);
    float phi1a = s0 * x0 * (x1 * x1);
    float phi1b = s1 * x1 * (x0 * x0);
    return v1 + phi0 + phi1a + phi1b;
}

// InverseWarp<uchar> (Warp1D.cpp:144-257): dst(x) = src(x - d_scale * disp(x)), cubic between
// pixels (linear next to the borders), skipped where the forward-warped depth map disagrees by
// disp_gap or more (occluded); untouched pixels of dst keep their value
void InverseWarp(const CByteImage &src, CByteImage &dst, const CFloatImage &disp, float d_scale,
                 float disp_gap, int order)
{
    CShape sh = src.Shape();
    int w = sh.width, h = sh.height, nb = sh.nBands, n = w * nb;
    CFloatImage fwd;
    if (disp_gap > 0.0f) {
        CFloatImage scaled(disp.Shape());
        fwd.ReAllocate(disp.Shape());
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float v = disp.Row(y)[x];
                scaled.Row(y)[x] = (d_scale != 1.0f) ? v * d_scale + 0.0f : v;     // Convert.cpp:31-47
                fwd.Row(y)[x] = -9999.0f;
            }
        ForwardWarp(scaled, fwd, disp, d_scale, true, disp_gap);
    }
    std::vector<float> from(n), to(n), shift(w);
    for (int y = 0; y < h; y++) {
        const uchar *ps = src.Row(y);
        uchar *pd = dst.Row(y);
        for (int i = 0; i < n; i++) {
            from[i] = ps[i];
            to[i] = pd[i];
        }
        for (int x = 0; x < w; x++) {
            float v = disp.Row(y)[x];
            shift[x] = (d_scale != 1.0f) ? v * d_scale + 0.0f : v;
        }
        const float *vis = disp_gap > 0.0f ? fwd.Row(y) : nullptr;
        for (int x = 0; x < w; x++) {                         // InverseWarpLine, :144-194
            float d = shift[x];
            float at = x - d;
            if (at < 0.0 || at > w - 1)
                continue;
            int xx = int(at);
            if (vis && disp_gap && std::fabs(d - vis[xx]) >= disp_gap)
                continue;
            const float *p0 = &from[(size_t)xx * nb];
            float *out = &to[(size_t)x * nb];
            if (order == 0 || xx == at) {
                for (int b = 0; b < nb; b++) out[b] = p0[b];
            } else if (order == 1 || xx - 1 < 0 || xx + 2 > w - 1) {
                float f = at - xx;
                for (int b = 0; b < nb; b++) out[b] = p0[b] + f * (p0[nb + b] - p0[b]);
            } else if (order == 3) {
                float f = at - xx;
                for (int b = 0; b < nb; b++)
                    out[b] = CubicInterpolate(f, p0[b - nb], p0[b], p0[nb + b], p0[2 * nb + b]);
            } else {
                throw CError("InverseWarp: order = %d not implemented", order);
            }
        }
        for (int i = 0; i < n; i++) {                         // back to uchar: + 0.5, clipped, truncated
            float v = to[i] * 1.0f + 0.5f;
            v = v > 0 ? v : 0;
            v = v < 255 ? v : 255;
            pd[i] = (uchar)v;
        }
    }
}

// Min/MaxFilter<uchar> 3x3 of every band, replicate border (what MinFilter.cpp:122-202 computes for
// xWidth = yWidth = 3), then the partially shuffled interval (PartialShuffle, StcEvaluate.cpp:588-614)
void PartialShuffle(const CByteImage &img, CByteImage &lo, CByteImage &hi, float amt)
{
    CShape sh = img.Shape();
    int w = sh.width, h = sh.height, nb = sh.nBands;
    lo.ReAllocate(sh);
    hi.ReAllocate(sh);
    auto at = [&](int x, int y, int b) {
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        return (int)img.Pixel(x, y, b);
    };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int b = 0; b < nb; b++) {
                int mn = 255, mx = 0;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        int v = at(x + dx, y + dy, b);
                        mn = std::min(mn, v);
                        mx = std::max(mx, v);
                    }
                int c = img.Pixel(x, y, b);
                lo.Pixel(x, y, b) = int(c + amt * (mn - c));                  // round down
                hi.Pixel(x, y, b) = int(c + amt * (mx - c) + 0.99f);          // round up
            }
}

}  // namespace

// BoxFilter of a one-band float image on the device (rows of the host image are 8-byte padded,
// the device copy is tight)
void CStereoMatcher::BoxFilterOnDevice(CFloatImage &img, int window, bool average)
{
    CShape sh = img.Shape();
    size_t wbytes = (size_t)sh.width * sizeof(float);
    void *d = DeviceBuffer(B_FDISP, wbytes * sh.height);
    Check(smhip_memcpy2d_h2d(m_ctx, d, wbytes, img.Data(), img.RowBytes(), wbytes, sh.height));
    Check(smhip_box_filter(m_ctx, (float *)d, sh.width, sh.height, 1, sh.width, window, window, average ? 1 : 0));
    Check(smhip_memcpy2d_d2h(m_ctx, img.Data(), img.RowBytes(), d, wbytes, wbytes, sh.height));
}

void CStereoMatcher::ComputeOcclusions()
{
    // forward-warp the true disparities into the matching frame, mark what nothing maps to, and
    // warp that visibility image back (StcEvaluate.cpp:42-79)
    CShape sh = m_true_disparity.Shape();
    int w = sh.width, h = sh.height;
    float fractional_shift = (frame_match - frame_ref) / std::fabs((float)m_frame_diff);
    const float invalid_depth = -9999.0f;
    CFloatImage fwd_depth(sh);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            fwd_depth.Row(y)[x] = invalid_depth;
    ForwardWarp(m_true_disparity, fwd_depth, m_true_disparity, fractional_shift, true, eval_disp_gap);
    CByteImage white(sh);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            white.Row(y)[x] = (fwd_depth.Row(y)[x] == invalid_depth) ? 0 : 255;
    m_occlusion.ReAllocate(sh);                               // pixels nothing maps to stay 0
    ForwardWarp(white, m_occlusion, fwd_depth, -fractional_shift, true, eval_disp_gap);
    // black borders (no statistics there), then visibility -> occlusion (:86-107)
    int b = eval_ignore_border;
    for (int y = 0; y < h; y++) {
        uchar *occ = m_occlusion.Row(y);
        for (int x = 0; x < w; x++) {
            if (x < b || x >= w - b || y < b || y >= h - b)
                occ[x] = 0;
            occ[x] = (uchar)(occ[x] * -1.0f + 255.0f);
        }
    }
}

void CStereoMatcher::ComputeTextureless()
{
    // squared horizontal gradient, box filter, threshold (StcEvaluate.cpp:110-157)
    CShape sh = m_reference.Shape();
    int w = sh.width, h = sh.height, nB = sh.nBands, nC = nB - (nB > 1);
    CShape s1(w, h, 1);
    m_textureless.ReAllocate(s1);
    CFloatImage sq_grad(s1);
    for (int y = 0; y < h; y++) {
        const uchar *p = m_reference.Row(y);
        float *s = sq_grad.Row(y);
        for (int x = 0; x < w - 1; x++, p += nB) {
            float sum2 = 0;
            for (int c = 0; c < nC; c++) {
                float diff = int(p[c]) - int(p[c + nB]);     // not centred ...
                sum2 += diff * diff;
            }
            sum2 /= (float)nC;
            s[x + 1] = sum2;
            if (x == 0)
                s[x] = sum2;
            s[x] = sum2 > s[x] ? sum2 : s[x];                 // ... this makes it centred
        }
    }
    if (eval_textureless_width > 0)
        BoxFilterOnDevice(sq_grad, eval_textureless_width, true);
    float squared_thresh = eval_textureless_thresh * eval_textureless_thresh;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            m_textureless.Row(y)[x] = (sq_grad.Row(y)[x] < squared_thresh) ? 255 : 0;
}

void CStereoMatcher::ComputeDisparityDiscont()
{
    // mark horizontal / vertical jumps >= eval_disp_gap of the true disparities, spread them with
    // a box filter, threshold above 0 (StcEvaluate.cpp:190-244).  The reference sums ints; the
    // marks are 0 / 255 and the sums stay far below 2^24, so the float operator is exact.
    CShape sh = m_true_disparity.Shape();
    int w = sh.width, h = sh.height;
    m_depth_discont.ReAllocate(sh);
    CFloatImage d_disc(sh);
    int bor = eval_ignore_border + 1;
    for (int y = 0; y < h - 1; y++) {
        if (y < bor || y >= h - bor)
            continue;
        const float *p0 = m_true_disparity.Row(y), *p1 = m_true_disparity.Row(y + 1);
        float *d0 = d_disc.Row(y), *d1 = d_disc.Row(y + 1);
        for (int x = bor; x < w - bor - 1; x++) {
            if (std::fabs(p0[x] - p0[x + 1]) >= eval_disp_gap)
                d0[x] = d0[x + 1] = 255;
            if (std::fabs(p0[x] - p1[x]) >= eval_disp_gap)
                d0[x] = d1[x] = 255;
        }
    }
    if (eval_discont_width > 0)
        BoxFilterOnDevice(d_disc, eval_discont_width, false);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            m_depth_discont.Row(y)[x] = (d_disc.Row(y)[x] != 0) ? 255 : 0;
}

void CStereoMatcher::ComputeDisparityErrors()
{
    // StcEvaluate.cpp:373-587; float sums in raster order, like the reference
    CShape sh = m_float_disparity.Shape();
    if (m_true_disparity.Shape() != sh)
        throw CError("Evaluate: invalid ground truth\n");
    int w = sh.width, h = sh.height;
    enum { ALL, NOCC, OCC, TEXD, TEXLESS, DISC, NSETS };
    float ssd[NSETS] = {0, 0, 0, 0, 0, 0};
    int bad[NSETS] = {0, 0, 0, 0, 0, 0}, cnt[NSETS] = {0, 0, 0, 0, 0, 0};
    int total_cnt_all = 0;
    eval_certain_matches_only = 0;                            // no status image after WTA (:441-443)
    auto add = [&](int k, float diff, int is_bad) { ssd[k] += diff * diff; bad[k] += is_bad; cnt[k]++; };
    for (int y = eval_ignore_border; y < h - eval_ignore_border; y++) {
        const float *disp = m_float_disparity.Row(y), *trud = m_true_disparity.Row(y);
        const uchar *ocnt = m_occlusion.Row(y), *texl = m_textureless.Row(y), *disc = m_depth_discont.Row(y);
        for (int x = eval_ignore_border; x < w - eval_ignore_border; x++) {
            float diff = disp[x] - trud[x];
            total_cnt_all++;
            int is_bad = (std::fabs(diff) > eval_bad_thresh);
            add(ALL, diff, is_bad);
            if (ocnt[x] > 1) {
                add(OCC, diff, is_bad);
            } else {
                add(NOCC, diff, is_bad);
                if (texl[x] != 0)
                    add(TEXLESS, diff, is_bad);
                else
                    add(TEXD, diff, is_bad);
                if (disc[x] != 0)
                    add(DISC, diff, is_bad);
            }
        }
    }
    cnt[ALL] += (cnt[ALL] == 0);
    total_cnt_all += (total_cnt_all == 0);
    fraction_matched = (float)cnt[ALL] / (float)total_cnt_all;
    float rms[NSETS], frac[NSETS];
    for (int k = 0; k < NSETS; k++) {
        cnt[k] += (cnt[k] == 0);
        rms[k] = std::sqrt(ssd[k] / (float)cnt[k]);
        frac[k] = (float)bad[k] / (float)cnt[k];
    }
    rms_error_all = rms[ALL];                 bad_pixels_all = frac[ALL];
    rms_error_nonocc = rms[NOCC];             bad_pixels_nonocc = frac[NOCC];
    rms_error_occ = rms[OCC];                 bad_pixels_occ = frac[OCC];
    rms_error_textured = rms[TEXD];           bad_pixels_textured = frac[TEXD];
    rms_error_textureless = rms[TEXLESS];     bad_pixels_textureless = frac[TEXLESS];
    rms_error_discont = rms[DISC];            bad_pixels_discont = frac[DISC];

    if (verbose >= eVerboseSummary) {
        if (verbose >= eVerboseProgress) {
            fprintf(stderr, "Results");
            if (eval_ignore_border > 0)
                fprintf(stderr, " (ignoring borders of %d pixels)", eval_ignore_border);
        }
        fprintf(stderr, ":\n  ALL   NON OCCL   OCCL   TEXTRD TEXTRLS D_DISCNT\n");
        fprintf(stderr, "%7.2f %7.2f %7.2f %7.2f %7.2f %7.2f  RMS disparity error\n", rms_error_all,
                rms_error_nonocc, rms_error_occ, rms_error_textured, rms_error_textureless, rms_error_discont);
        fprintf(stderr, "%7.2f%%%7.2f%%%7.2f%%%7.2f%%%7.2f%%%7.2f%% bad pixels (disp error > %g)\n",
                100.0f * bad_pixels_all, 100.0f * bad_pixels_nonocc, 100.0f * bad_pixels_occ,
                100.0f * bad_pixels_textured, 100.0f * bad_pixels_textureless, 100.0f * bad_pixels_discont,
                eval_bad_thresh);
        if (total_time >= 0.0f)
            fprintf(stderr, "  time = %gs", total_time);
        if (final_energy >= 0.0f)
            fprintf(stderr, "  energy = %.0f", final_energy);
        fraction_matched *= 0.1f;                             // plotted on the bad-pixel scale (:583)
    }
}

void CStereoMatcher::ComputePredictionError(CByteImage &predicted, const CByteImage &original,
                                            float &rms_error, float &fraction_visible)
{
    // StcEvaluate.cpp:615-690: squared colour differences over the visible (alpha == 255) pixels
    CShape sh = predicted.Shape();
    int w = sh.width, h = sh.height, nB = sh.nBands;
    int nC = nB - (nB > 1);
    float sum2 = 0.0;
    int visible = 0, total = w * h;
    CByteImage pred_min, pred_max, orig_min, orig_max;
    bool shuffle = eval_partial_shuffle > 0.0f;
    if (shuffle) {
        PartialShuffle(predicted, pred_min, pred_max, eval_partial_shuffle);
        PartialShuffle(original, orig_min, orig_max, eval_partial_shuffle);
    }
    for (int y = 0; y < h; y++) {
        uchar *p = predicted.Row(y);
        const uchar *o = original.Row(y);
        for (int x = 0; x < w; x++, p += nB, o += nB) {
            if (nB > 1 && p[nC] != 255)
                continue;
            visible++;
            for (int b = 0; b < nC; b++) {
                float diff = int(p[b]) - int(o[b]);
                if (shuffle) {
                    int pn = pred_min.Pixel(x, y, b), px = pred_max.Pixel(x, y, b);
                    int on = orig_min.Pixel(x, y, b), ox = orig_max.Pixel(x, y, b);
                    if (std::max(pn, on) <= std::min(px, ox))
                        diff = 0;                             // the intervals overlap
                    else
                        diff = (pn > ox) ? pn - ox : on - px; // the gap between them
                }
                sum2 += diff * diff;
                if (eval_predict_diff && (nB == 1 || b < nB - 1))
                    p[b] = std::max(0, std::min(255, 128 + int(diff * eval_predict_diff)));
            }
        }
    }
    rms_error = std::sqrt(sum2 / nC / (visible + (visible == 0)));
    fraction_visible = visible / (float)total;
}

void CStereoMatcher::ComputePredictionErrors()
{
    // StcEvaluate.cpp:692-758: every frame is predicted from the reference image and the computed
    // disparities (forward warp) or pulled back onto the reference (inverse warp, cubic)
    const int inverse_warp_order = 3;
    for (int f = 0; f < (int)m_frame.size(); f++) {
        CStereoFrame &frame = m_frame[f];
        const CByteImage &original = frame.input_image;
        CByteImage &resampled = frame.resampled_image;
        CShape sh = original.Shape();
        if (sh != m_reference.Shape())
            throw CError("ComputePredictionErrors: frame %d has another shape than the reference frame", f);
        resampled.ReAllocate(sh);
        for (int y = 0; y < sh.height; y++) {                 // the empty colour (ARGB; alpha must be 0)
            uchar *r = resampled.Row(y);
            for (int x = 0; x < sh.width; x++) {
                if (sh.nBands == 1)
                    r[x] = (uchar)eval_empty_color;
                else
                    memcpy(&r[4 * x], &eval_empty_color, 4);
            }
        }
        float fractional_shift = (f - frame_ref) / std::fabs((float)m_frame_diff);
        if (eval_predict_type == 1) {
            ForwardWarp(m_reference, resampled, m_float_disparity, fractional_shift, eval_lin_interp != 0,
                        eval_disp_gap);
            ComputePredictionError(resampled, original, frame.predict_err, frame.predict_visible);
        } else {
            InverseWarp(original, resampled, m_float_disparity, fractional_shift, eval_disp_gap,
                        inverse_warp_order);
            ComputePredictionError(resampled, m_reference, frame.predict_err, frame.predict_visible);
        }
        if (verbose >= eVerbosePredictionError)
            fprintf(stderr, " prediction error for frame %d: RMS error = %.2f, visible = %.2f%%\n", f,
                    frame.predict_err, frame.predict_visible * 100);
        // view positions (0 = ref, 1 = match): near -0.5 or -1, middle 0.5, match 1, far 1.5 or 2
        if (2 * f == 4 * frame_ref - 2 * frame_match || 2 * f == 3 * frame_ref - 1 * frame_match)
            predict_err_near = frame.predict_err;
        if (2 * f == 1 * frame_ref + 1 * frame_match)
            predict_err_middle = frame.predict_err;
        if (2 * f == 0 * frame_ref + 2 * frame_match)
            predict_err_match = frame.predict_err;
        if (2 * f == -1 * frame_ref + 3 * frame_match || 2 * f == -2 * frame_ref + 4 * frame_match)
            predict_err_far = frame.predict_err;
    }
}

// ComputeMatchQuality (StcEvaluate.cpp:760-863): cost and certainty at the winner, computed by
// smhip_match_quality on the DSI that is still resident on the device; with verbose >= eVerboseDumpFiles
// the two images go to reprojected/final_cost.pgm (x16) and reprojected/certainty.pgm (x0.5) as in
// :836-846, and the three cost histograms of :848-861 follow (all pixels, occluded, textureless).

// Histogram of a one-band float image drawn as a figure of n_bins x height pixels (the reference's Histogram1D,
// Histogram1D.cpp:32-139, for the arguments ComputeMatchQuality gives it: fixed value range, n_bins > 0,
// height > 0).  A mask of the image's size restricts the count -- and, as there, the first zero of the mask ends
// the scan of its ROW (Histogram1D.cpp:86-87 leaves the loop over x), it does not skip one pixel.  v_scale <= 0:
// the tallest bin fills the height, and the scale found is handed back for the next call (:855).
static void DrawCostHistogram(const CFloatImage &img, const CByteImage &mask, CByteImage &figure, float min_val,
                              float max_val, int n_bins, int height, float &v_scale)
{
    const CShape sh = img.Shape(), msh = mask.Shape();
    const bool valid_mask = sh.width == msh.width && sh.height == msh.height;
    const float step_size = (max_val - min_val) / (float)n_bins;           // Histogram1D.cpp:69-70
    const float inv_step_size = 1.0f / step_size;
    std::vector<int> counts((size_t)n_bins, 0);
    for (int y = 0; y < sh.height; y++) {
        const float *p = img.Row(y);
        const uchar *m = valid_mask ? mask.Row(y) : nullptr;
        for (int x = 0; x < sh.width; x++) {
            if (m && m[x] == 0)
                break;
            int bin = (int)((p[x] - min_val) * inv_step_size);
            bin = std::min(std::max(0, bin), n_bins - 1);
            counts[(size_t)bin] += 1;
        }
    }
    const int max_count = *std::max_element(counts.begin(), counts.end());
    if (v_scale <= 0)
        v_scale = height / (float)max_count;                                // Histogram1D.cpp:118-119
    figure.ReAllocate(CShape(n_bins, height, 1));
    for (int y = 0; y < height; y++) {
        uchar *row = figure.Row(y);
        const int y2 = height - 1 - y;
        for (int x = 0; x < n_bins; x++) {
            const int c = std::min(height - 1, (int)(counts[(size_t)x] * v_scale + 0.5));
            row[x] = y2 >= c ? 255 : 0;
        }
    }
}

void CStereoMatcher::ComputeMatchQuality()
{
    if (m_used_fused || m_keys_done || !m_dev[B_COST])
        throw CError("ComputeMatchQuality: the aggregated DSI is not available");
    const int w = m_mp.width, h = m_mp.height;
    const size_t wbytes = (size_t)w * sizeof(float);
    DeviceBuffer(B_FDISP, wbytes * h);
    DeviceBuffer(B_FCOST, wbytes * h);
    DeviceBuffer(B_FCERT, wbytes * h);
    Check(smhip_memcpy2d_h2d(m_ctx, m_dev[B_FDISP], wbytes, m_float_disparity.Data(),
                             m_float_disparity.RowBytes(), wbytes, h));
    const int norm_bands = m_reference.Shape().nBands == 1 ? 1 : 3;              // :774
    Check(smhip_match_quality(m_ctx, (const float *)m_dev[B_COST],
                              aggr_subpixel ? (const float *)m_dev[B_CERT] : nullptr, (const float *)m_dev[B_FDISP],
                              w, h, m_disp_n, m_row_elems, m_cert_row_elems, (float)disp_min, m_disp_step_inv,
                              eval_match_quality, aggr_subpixel != 0, norm_bands, match_fn == eSD,
                              (float *)m_dev[B_FCOST], (float *)m_dev[B_FCERT]));
    m_final_cost.ReAllocate(CShape(w, h, 1));
    m_certainty.ReAllocate(CShape(w, h, 1));
    Check(smhip_memcpy2d_d2h(m_ctx, m_final_cost.Data(), m_final_cost.RowBytes(), m_dev[B_FCOST], wbytes, wbytes, h));
    Check(smhip_memcpy2d_d2h(m_ctx, m_certainty.Data(), m_certainty.RowBytes(), m_dev[B_FCERT], wbytes, wbytes, h));
    if (verbose < eVerboseDumpFiles)
        return;
    fprintf(stderr, "Writing final_cost.pgm and certainty.pgm\n");
    CByteImage fc(w, h, 1);
    const struct { const CFloatImage *img; float scale; const char *name; } dumps[] = {
        {&m_final_cost, 16.0f, "reprojected/final_cost.pgm"}, {&m_certainty, 0.5f, "reprojected/certainty.pgm"}};
    for (const auto &d : dumps) {
        for (int y = 0; y < h; y++) {
            const float *src = d.img->Row(y);
            uchar *dst = fc.Row(y);
            for (int x = 0; x < w; x++) {                     // ScaleAndOffset float -> uchar, Convert.cpp:33-45
                float v = src[x] * d.scale + 0.0f;
                v = v < 0.0f ? 0.0f : v;
                v = v > 255.0f ? 255.0f : v;
                dst[x] = (uchar)v;
            }
        }
        WriteImage(fc, d.name);
    }
    // :848-861 -- cost range [0, 32) in 256 bins, 256 rows; the first figure's vertical scale serves all three
    float vscale = 0.0f;
    const struct { const CByteImage *mask; const char *name; } hists[] = {
        {nullptr, "reprojected/final_cost_hist_all.pgm"},
        {&m_occlusion, "reprojected/final_cost_hist_occluded.pgm"},
        {&m_textureless, "reprojected/final_cost_hist_textureless.pgm"}};
    const CByteImage no_mask;
    for (const auto &hst : hists) {
        DrawCostHistogram(m_final_cost, hst.mask ? *hst.mask : no_mask, fc, 0.0f, 32.0f, 256, 256, vscale);
        WriteImage(fc, hst.name);
    }
}

void CStereoMatcher::Evaluate()
{
    // StcEvaluate.cpp:957-984
    if (m_float_disparity.Shape().width == 0)
        throw CError("Evaluate: no disparities computed");
    ComputeOcclusions();
    ComputeTextureless();
    ComputeDisparityDiscont();
    ComputeDisparityErrors();
    if (eval_predict_type != 0)
        ComputePredictionErrors();
    if (eval_match_quality && !evaluate_only)                 // :977-979
        ComputeMatchQuality();
    // ComputeStatusErrors (:865-954) needs the status map that only OptSymmetric writes (out of scope,
    // SURVEY 8): on this path m_status is empty and the reference skips it as well (:982-983).
    if (verbose >= eVerboseSummary)
        fprintf(stderr, "\n");
}
