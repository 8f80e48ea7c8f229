// StcEvaluate.cpp -- the evaluator behind `depth_map`: occlusion, textureless and
// depth-discontinuity masks from the ground truth, then RMS / bad-pixel statistics of
// m_float_disparity (reference: StcEvaluate.cpp:42-587, ForwardWarp Warp1D.cpp:25-128).
// These are the rms_error_* / bad_pixels_* lines of results/*.txt, the reference's only quality
// read-out.  O(W*H) host work; the two box filters (BoxFilter<float> 3x3 average over the
// squared gradient, BoxFilter<int> 9x9 sum over the discontinuity marks) run through
// smhip_box_filter on one-band images -- the same operator that aggregates the DSI.
// Not restated: prediction errors by re-projection (eval_predict_type, :615-758), match quality
// (:760-863) and the status statistics (:865-954; WTA leaves no status map) -- their outputs
// keep the "not computed" value -1.
#include "StereoMatcher.h"
#include "Error.h"
#include <cmath>

#define ROUND(x) ((int)((x) >= 0 ? (x) + .5 : (x) - .5))     // Warp1D.cpp:23

namespace {

// draw_intensity_line (Warp1D.cpp:25-56) for one band.  T = float: no clipping, no rounding
// offset; T = uchar: + 0.5, clipped to [0, 255], truncated.
template <class T>
void DrawLine(const T *src1, const T *src2, T *dst, float x1, float x2, int w)
{
    if (x2 < x1)
        return;                                               // backward facing line
    const bool is_float = sizeof(T) == sizeof(float);
    int i0 = std::max(0, ROUND(x1));
    int i1 = std::min(w - 1, ROUND(x2));
    float iden = 1.0 / (x2 - x1 + (x2 == x1));
    for (int i = i0; i <= i1; i++) {
        float f = (i - x1) * iden;
        float v1 = src1[0], v2 = src2[0];
        float v = v1 + f * (v2 - v1);
        if (is_float) {
            dst[i] = (T)(v + 0.0f);
        } else {
            float r = v + 0.5f;
            r = r > 0 ? r : 0;
            r = r < 255 ? r : 255;
            dst[i] = (T)r;
        }
    }
}

// ForwardWarp (Warp1D.cpp:58-128) of a one-band image with line interpolation
template <class T>
void ForwardWarp1(const CImageOf<T> &src, CImageOf<T> &dst, const CFloatImage &disp, float d_scale,
                  float disp_gap)
{
    CShape sh = src.Shape();
    int w = sh.width, h = sh.height;
    for (int y = 0; y < h; y++) {
        int xstart = d_scale > 0 ? 0 : w - 1, xend = d_scale > 0 ? w : -1, xincr = d_scale > 0 ? 1 : -1;
        const float *dp = disp.Row(y);
        const T *ps = src.Row(y);
        T *pd = dst.Row(y);
        for (int x = xstart; x != xend; x += xincr) {
            int x2 = x + xincr;
            bool draw_line = (x2 != xend) && (std::fabs(dp[x] - dp[x2]) < disp_gap);
            float d = d_scale * dp[x];
            if (draw_line) {
                float d2 = d_scale * dp[x2];
                if (xincr > 0)
                    DrawLine(&ps[x], &ps[x2], pd, x - d, x2 - d2, w);
                else
                    DrawLine(&ps[x2], &ps[x], pd, x2 - d, x - d2, w);   // as written at Warp1D.cpp:115-117
                continue;
            }
            int xx = x - ROUND(d);
            if (xx >= 0 && xx < w)
                pd[xx] = ps[x];
        }
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
    ForwardWarp1(m_true_disparity, fwd_depth, m_true_disparity, fractional_shift, eval_disp_gap);
    CByteImage white(sh);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            white.Row(y)[x] = (fwd_depth.Row(y)[x] == invalid_depth) ? 0 : 255;
    m_occlusion.ReAllocate(sh);                               // pixels nothing maps to stay 0
    ForwardWarp1(white, m_occlusion, fwd_depth, -fractional_shift, eval_disp_gap);
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

void CStereoMatcher::Evaluate()
{
    // StcEvaluate.cpp:957-984
    if (m_float_disparity.Shape().width == 0)
        throw CError("Evaluate: no disparities computed");
    ComputeOcclusions();
    ComputeTextureless();
    ComputeDisparityDiscont();
    ComputeDisparityErrors();
    if (eval_predict_type != 0 && verbose >= eVerboseProgress)
        fprintf(stderr, "\n  (prediction errors by re-projection are not computed: predict_err_* stay -1)");
    if (verbose >= eVerboseSummary)
        fprintf(stderr, "\n");
}
