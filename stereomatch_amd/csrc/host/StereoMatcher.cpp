// StereoMatcher.cpp -- stage sequencing of the reference (StereoMatcher.cpp:38-181,
// StcAggregate.cpp:66-230, 255-400, StcOptimize.cpp:162-280, StcRefine.cpp:28-100) over the C-ABI.
#include "StereoMatcher.h"
#include "Error.h"
#include "ImageIO.h"
#include <cmath>
#include <cstring>
#include <ctime>

#define ROUND(x) ((int)((x) >= 0 ? (x) + .5 : (x) - .5))

CStereoMatcher::CStereoMatcher()
{
    for (int i = 0; i < B_COUNT; i++) { m_dev[i] = nullptr; m_dev_bytes[i] = 0; }
    if (smhip_init(0, &m_ctx) != SMHIP_OK)      // device 0, like prepareDevice (CudaUtilities.cu:124)
        throw CError("%s", smhip_last_error());
}

CStereoMatcher::~CStereoMatcher()
{
    if (!m_ctx)
        return;
    for (int i = 0; i < B_COUNT; i++)
        if (m_dev[i]) smhip_free(m_ctx, m_dev[i]);
    smhip_shutdown(m_ctx);
}

void CStereoMatcher::Check(int rc)
{
    if (rc != SMHIP_OK)
        throw CError("%s", smhip_last_error());
}

void *CStereoMatcher::DeviceBuffer(int slot, size_t bytes)
{
    if (m_dev_bytes[slot] < bytes) {
        if (m_dev[slot]) Check(smhip_free(m_ctx, m_dev[slot]));
        m_dev[slot] = nullptr; m_dev_bytes[slot] = 0;
        Check(smhip_malloc(m_ctx, bytes, &m_dev[slot]));
        m_dev_bytes[slot] = bytes;
    }
    return m_dev[slot];
}

bool CStereoMatcher::Fusable() const
{
    // one box iteration + WTA and nothing downstream needs the full DSI
    return allow_fused && aggr_fn == eBox && aggr_iter == 1 && aggr_minfilter <= 1 && opt_fn == eWTA &&
           !refine_subpix && !aggr_subpixel && !aggr_collapse && !eval_match_quality &&
           m_disp_den == 1 && verbose < eVerboseDumpFiles;
}

bool CStereoMatcher::IterFusable() const
{
    // binomial iterations + WTA and nothing downstream needs the DSI: smhip_fused_iter_wta_keys (aggr_iter <= 4 as
    // one kernel that moves no DSI, more between its raw-cost and its WTA end)
    return allow_fused && aggr_fn == eBinomial && aggr_iter >= 1 && aggr_minfilter <= 1 && opt_fn == eWTA &&
           !refine_subpix && !aggr_subpixel && !aggr_collapse && !eval_match_quality &&
           m_disp_den == 1 && m_disp_n <= 65535 && verbose < eVerboseDumpFiles;
}

bool CStereoMatcher::BoxDsiFusable() const
{
    // RawCosts + one AggrBox in two passes over a private slab (12 instead of 20 bytes per voxel)
    return allow_fused && aggr_fn == eBox && aggr_iter == 1 && m_disp_den == 1 && m_disp_n >= 33 &&
           m_disp_n % 4 == 0 && (m_mp.n_bands == 1 || m_reference.RowBytes() % 4 == 0) &&
           !eval_match_quality && verbose < eVerboseDumpFiles;
}

bool CStereoMatcher::MinWtaFusable() const
{
    // AggrMin + PadCosts + OptWTA without writing the filtered DSI
    return allow_fused && aggr_minfilter >= 3 && aggr_minfilter <= 11 && aggr_minfilter % 2 == 1 &&
           opt_fn == eWTA && m_disp_den == 1 && m_disp_n % 4 == 0 && !refine_subpix && !aggr_subpixel &&
           !aggr_collapse && !eval_match_quality && verbose < eVerboseDumpFiles;
}

void CStereoMatcher::ComputeCorrespondence()
{
    if (frame_ref < 0 || frame_ref >= (int)m_frame.size())
        throw CError("ComputeCorrespondence: invalid reference frame number %d", frame_ref);
    if (frame_match < 0 || frame_match >= (int)m_frame.size())
        throw CError("ComputeCorrespondence: invalid matching frame number %d", frame_match);
    CByteImage &ref = m_frame[frame_ref].input_image;
    CByteImage &mat = m_frame[frame_match].input_image;
    CShape sh0 = ref.Shape(), sh1 = mat.Shape();
    if (sh0.width * sh0.height == 0)
        throw CError("ComputeCorrespondence: invalid reference image");
    if (sh1.width * sh1.height == 0)
        throw CError("ComputeCorrespondence: invalid matching image");
    if (sh0 != sh1)
        throw CError("ComputeCorrespondence: reference and matching not the same size");
    m_frame_diff = frame_match - frame_ref;
    m_frame_diff_sign = (m_frame_diff > 0) ? 1 : -1;

    const float min_precision = 1e-3f;
    if (disp_step <= 0.0 ||
        (disp_step < 1.0 && fabs(1.0 / disp_step - ROUND(1.0 / disp_step)) > min_precision) ||
        (disp_step > 1.0 && fabs(disp_step - ROUND(disp_step)) > min_precision))
        throw CError("ComputeCorrespondence: disp_step must integer N or 1.0/N");
    m_disp_num = disp_step < 1.0f ? 1 : ROUND(disp_step);
    m_disp_den = disp_step < 1.0f ? ROUND(1.0 / disp_step) : 1;
    disp_step = m_disp_num / (float)m_disp_den;
    m_disp_step_inv = m_disp_den / (float)m_disp_num;
    disp_n = int(m_disp_step_inv * (disp_max - disp_min)) + 1;
    m_disp_step = disp_step;
    m_disp_n = disp_n;
    if (verbose >= eVerboseProgress) {
        if (evaluate_only)
            fprintf(stderr, "\nEvaluating existing depth map\n");
        else
            fprintf(stderr, "\nComputing correspondence between frames %d and %d\n"
                            "disp_n = %d, disp_min = %d, m_disp_num = %d, m_disp_den = %d, disp_scale = %g\n",
                    frame_ref, frame_match, disp_n, disp_min, m_disp_num, m_disp_den, disp_scale);
    }

    m_reference = ref;
    m_matching = mat;
    CShape shD(sh0.width, sh0.height, 1);
    CByteImage &depth_image = m_frame[frame_ref].depth_image;
    if (depth_image.Shape() != shD)
        depth_image.ReAllocate(shD);                          // (zeros; a depth map read for evaluate_only is kept)
    m_float_disparity.ReAllocate(shD);
    {   // m_float_disparity from the depth map on file (StereoMatcher.cpp:117): what evaluate_only evaluates
        float d_scale = 1.0f / disp_scale, d_offset = (float)disp_min;
        bool d_so = (d_scale != 1.0f) || (d_offset != 0.0f);
        for (int y = 0; y < shD.height; y++) {
            const uchar *src = depth_image.Row(y);
            float *dst = m_float_disparity.Row(y);
            for (int x = 0; x < shD.width; x++)
                dst[x] = d_so ? src[x] * d_scale + d_offset : (float)src[x];
        }
    }
    // ground truth in disparity units, zeros when there is none (StereoMatcher.cpp:121-130)
    m_true_disparity.ReAllocate(shD);
    const CByteImage &truth_image = m_frame[frame_ref].truth_image;
    if (truth_image.Shape() == shD) {
        float t_scale = 1.0f / disp_scale, t_offset = (float)disp_min;
        bool t_so = (t_scale != 1.0f) || (t_offset != 0.0f);
        for (int y = 0; y < shD.height; y++) {
            const uchar *src = truth_image.Row(y);
            float *dst = m_true_disparity.Row(y);
            for (int x = 0; x < shD.width; x++)
                dst[x] = t_so ? src[x] * t_scale + t_offset : (float)src[x];
        }
    }
    m_disparity.ReAllocate(shD);
    m_best_cost.ReAllocate(shD);
    if (evaluate_only) {
        // StereoMatcher.cpp:132-152: integer levels k = (d - disp_min) / disp_step from the float
        // disparities (Convert.cpp float -> int: truncation), and nothing of the matcher runs
        for (int y = 0; y < shD.height; y++) {
            const float *src = m_float_disparity.Row(y);
            int *dst = m_disparity.Row(y);
            for (int x = 0; x < shD.width; x++)
                dst[x] = (int)(src[x] * m_disp_step_inv + (-(float)disp_min * disp_step));
        }
        return;
    }
    if (m_disp_n < 2)
        throw CError("ComputeCorrespondence: too few disparity levels (%d)", m_disp_n);

    m_mp.width = sh0.width; m_mp.height = sh0.height; m_mp.n_bands = sh0.nBands;
    m_mp.disp_min = disp_min; m_mp.disp_num = m_disp_num; m_mp.disp_den = m_disp_den;
    m_mp.disp_n = m_disp_n; m_mp.frame_diff_sign = m_frame_diff_sign;
    m_mp.match_fn = match_fn; m_mp.match_interp = match_interp; m_mp.match_max = match_max;
    match_interval = match_interval ? 1 : 0;          // StcRawCosts.cpp:354
    m_mp.match_interval = match_interval; m_mp.match_interpolated = match_interpolated;
    m_match_outside = smhip_match_outside(&m_mp);

    clock_t time0 = clock();
    // images to the device (host rows are padded to 8 bytes; the pitch travels with them)
    Check(smhip_memcpy_h2d(m_ctx, DeviceBuffer(B_REF, m_reference.ImageSize()), m_reference.Data(),
                           m_reference.ImageSize()));
    Check(smhip_memcpy_h2d(m_ctx, DeviceBuffer(B_MTC, m_matching.ImageSize()), m_matching.Data(),
                           m_matching.ImageSize()));
    size_t n_pix = (size_t)sh0.width * sh0.height;
    DeviceBuffer(B_DISP, n_pix * sizeof(int32_t));
    DeviceBuffer(B_BEST, n_pix * sizeof(float));

    PreProcess();
    m_keys_done = m_raw_deferred = false;
    m_used_fused = Fusable();
    if (m_used_fused) {
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- fused device pass: costs, box filter (size %d), pad, WTA\n", aggr_window_size);
        Check(smhip_fused_box_wta(m_ctx, (const uint8_t *)m_dev[B_REF], (const uint8_t *)m_dev[B_MTC],
                                  m_reference.RowBytes(), &m_mp, aggr_window_size,
                                  (int32_t *)m_dev[B_DISP], (float *)m_dev[B_BEST]));
    } else if (IterFusable()) {
        m_used_fused = true;
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- fused device pass: costs, binomial filter iterated %d times, pad, WTA\n", aggr_iter);
        Check(smhip_fused_iter_wta_keys(m_ctx, (const uint8_t *)m_dev[B_REF], (const uint8_t *)m_dev[B_MTC],
                                        m_reference.RowBytes(), &m_mp, aggr_fn, aggr_iter, diff_lambda, diff_beta, 0,
                                        m_disp_n, (int64_t *)DeviceBuffer(B_KEYS, n_pix * sizeof(int64_t))));
        Check(smhip_keys_decode(m_ctx, (const int64_t *)m_dev[B_KEYS], m_mp.width, m_mp.height,
                                (int32_t *)m_dev[B_DISP], (float *)m_dev[B_BEST]));
    } else {
        RawCosts();
        Aggregate();
    }
    Optimize();
    Refine();
    total_time = (float)(clock() - time0) / (float)CLOCKS_PER_SEC;
    if (verbose >= eVerboseTiming)
        fprintf(stderr, "* total time: %gs\n", total_time);

    // depth map: uchar(clip(fdisp*disp_scale + (-disp_min*disp_scale + 0.5))), StereoMatcher.cpp:177
    float offset = (float)(-disp_min * disp_scale + 0.5);
    bool scale_offset = (disp_scale != 1.0f) || (offset != 0.0f);
    for (int y = 0; y < shD.height; y++) {
        const float *src = m_float_disparity.Row(y);
        uchar *dst = depth_image.Row(y);
        for (int x = 0; x < shD.width; x++) {
            float val = scale_offset ? src[x] * disp_scale + offset : src[x];
            val = val < 0.0f ? 0.0f : (val > 255.0f ? 255.0f : val);
            dst[x] = (uchar)val;
        }
    }
    m_reference = ref;                                        // undo PreProcess for the evaluator (:180)
}

void CStereoMatcher::PreProcess()
{
    if (preproc_addnoise_stddev > 0.0f)                       // StcPreProcess.cpp:33-37
        throw CError("PreProcess: additive noise not yet implemented");
    if (preproc_blur_iter <= 0)
        return;
    if (verbose >= eVerboseProgress)
        fprintf(stderr, "- pre-process: binomial 121 filter iterated %d times\n", preproc_blur_iter);
    // blur the device copies of both frames (:49-55); the blurred reference comes back to the host
    // because the smoothness term of the energy reads it (StcOptimize.cpp:103-116)
    Check(smhip_preprocess_blur(m_ctx, (uint8_t *)m_dev[B_REF], m_reference.RowBytes(), m_mp.width, m_mp.height,
                                m_mp.n_bands, preproc_blur_iter));
    Check(smhip_preprocess_blur(m_ctx, (uint8_t *)m_dev[B_MTC], m_matching.RowBytes(), m_mp.width, m_mp.height,
                                m_mp.n_bands, preproc_blur_iter));
    Check(smhip_memcpy_d2h(m_ctx, m_reference.Data(), m_dev[B_REF], m_reference.ImageSize()));
}

void CStereoMatcher::RawCosts()
{
    if (verbose >= eVerboseProgress)
        fprintf(stderr, "- computing costs: %s%s%s\n", match_fn == eAD ? "AD" : "SD",
                match_max < 1000 ? ", trunc" : "", match_interval ? ", interval" : "");
    m_row_elems = (size_t)m_mp.width * m_disp_n;
    DeviceBuffer(B_COST, m_row_elems * m_mp.height * sizeof(float));
    m_keys_done = false;
    m_raw_deferred = BoxDsiFusable();
    if (m_raw_deferred)
        return;                                               // made inside AggrBox
    Check(smhip_raw_costs(m_ctx, (const uint8_t *)m_dev[B_REF], (const uint8_t *)m_dev[B_MTC],
                          m_reference.RowBytes(), &m_mp, (float *)m_dev[B_COST], m_row_elems));
    if (verbose >= eVerboseDumpFiles)                         // StcRawCosts.cpp:421-422
        WriteCosts("reprojected/RAW_DSI_%03d.pgm");
}

// WriteCosts (StcAggregate.cpp:27-52): the resident DSI comes back to the host and goes out as
// one PGM per level, scaled by 2 and clipped (ScaleAndOffset float -> uchar, Convert.cpp:33-40);
// a name ending in .pmf writes the whole DSI losslessly instead (the CAN_VIEW_PMFS branch, :29-35).
void CStereoMatcher::WriteCosts(const char *filename)
{
    int w = m_mp.width, h = m_mp.height, nd = m_disp_n;
    CFloatImage cost(w, h, nd);
    size_t wbytes = (size_t)w * nd * sizeof(float);
    Check(smhip_memcpy2d_d2h(m_ctx, cost.Data(), cost.RowBytes(), m_dev[B_COST], m_row_elems * sizeof(float),
                             wbytes, h));
    size_t len = strlen(filename);
    if (len >= 4 && strcmp(filename + len - 4, ".pmf") == 0) {
        WriteImage(cost, filename);
        return;
    }
    const float scaleUp = 2;
    CByteImage plane(w, h, 1);
    for (int d = 0; d < nd; d++) {
        char name[1024];
        snprintf(name, sizeof(name), filename, d);
        for (int y = 0; y < h; y++) {
            const float *src = cost.Row(y);
            uchar *dst = plane.Row(y);
            for (int x = 0; x < w; x++) {
                float v = src[(size_t)x * nd + d] * scaleUp + 0.0f;
                v = v < 0.0f ? 0.0f : v;
                v = v > 255.0f ? 255.0f : v;
                dst[x] = (uchar)v;
            }
        }
        WriteImage(plane, name);
    }
}

void CStereoMatcher::AggrBox()
{
    if (m_raw_deferred) {
        Check(smhip_fused_box_dsi(m_ctx, (const uint8_t *)m_dev[B_REF], (const uint8_t *)m_dev[B_MTC],
                                  m_reference.RowBytes(), &m_mp, aggr_window_size, 0, m_disp_n,
                                  (float *)m_dev[B_COST], m_row_elems));
        return;
    }
    Check(smhip_box_filter(m_ctx, (float *)m_dev[B_COST], m_mp.width, m_mp.height, m_disp_n, m_row_elems,
                           aggr_window_size, aggr_window_size, 1));
}

void CStereoMatcher::SwapCost()
{
    std::swap(m_dev[B_COST], m_dev[B_COST2]);                // m_cost <-> m_cost2 (StcDiffusion.cpp:58-63)
    std::swap(m_dev_bytes[B_COST], m_dev_bytes[B_COST2]);
}

void CStereoMatcher::AggrMin()
{
    DeviceBuffer(B_COST2, m_row_elems * m_mp.height * sizeof(float));
    Check(smhip_min_filter_to(m_ctx, (const float *)m_dev[B_COST], (float *)m_dev[B_COST2], m_mp.width,
                              m_mp.height, m_disp_n, m_row_elems, aggr_minfilter, aggr_minfilter));
    SwapCost();
}

void CStereoMatcher::AggrDiffusion(int iter)
{
    size_t bytes = m_row_elems * m_mp.height * sizeof(float);
    if (diff_beta > 0.0f && iter == 0)                       // StcDiffusion.cpp:52-56
        Check(smhip_memcpy_d2d(m_ctx, DeviceBuffer(B_COST0, bytes), m_dev[B_COST], bytes));
    DeviceBuffer(B_COST2, bytes);
    Check(smhip_diffusion_iter(m_ctx, (const float *)m_dev[B_COST], (float *)m_dev[B_COST2],
                               diff_beta > 0.0f ? (const float *)m_dev[B_COST0] : nullptr, m_mp.width,
                               m_mp.height, m_disp_n, m_row_elems, diff_lambda, diff_beta));
    SwapCost();
}

void CStereoMatcher::PadCosts()
{
    Check(smhip_pad_costs(m_ctx, (float *)m_dev[B_COST], m_row_elems, &m_mp));
}

void CStereoMatcher::Aggregate()
{
    // The aggr_iter loop of StcAggregate.cpp:265-343.  The iterated stencils go to the device as
    // one call: it runs up to four iterations per pass over the DSI and says which of m_cost /
    // m_cost2 holds the result.
    size_t bytes = m_row_elems * m_mp.height * sizeof(float);
    int in_b = 0;
    switch (aggr_fn) {
    case eBox:
        for (int iter = 0; iter < aggr_iter; iter++) {
            if (verbose >= eVerboseProgress && iter < 3)
                fprintf(stderr, "- aggregating: box filter, size %d\n", aggr_window_size);
            AggrBox();
        }
        break;
    case eBinomial:
        if (verbose >= eVerboseProgress && aggr_iter > 0)
            fprintf(stderr, "- aggregating: binomial 14641 filter\n");
        if (aggr_iter > 0) {
            DeviceBuffer(B_COST2, bytes);
            Check(smhip_convolve_14641_run(m_ctx, (float *)m_dev[B_COST], (float *)m_dev[B_COST2], m_mp.width,
                                           m_mp.height, m_disp_n, m_row_elems, aggr_iter, &in_b));
        }
        break;
    case eDiffusion:
    case eMembrane:
        if (aggr_fn == eDiffusion) {
            if (verbose >= eVerboseProgress && aggr_iter > 0)
                fprintf(stderr, "- aggregating: diffusion, lambda = %g\n", diff_lambda);
            if (aggr_iter > 0)
                diff_beta = 0.0f;                             // StcAggregate.cpp:305
        } else if (verbose >= eVerboseProgress && aggr_iter > 0) {
            fprintf(stderr, "- aggregating: membrane model, lambda = %g, beta = %g\n", diff_lambda, diff_beta);
        }
        if (aggr_iter > 0) {
            if (diff_beta > 0.0f)                             // StcDiffusion.cpp:52-56
                Check(smhip_memcpy_d2d(m_ctx, DeviceBuffer(B_COST0, bytes), m_dev[B_COST], bytes));
            DeviceBuffer(B_COST2, bytes);
            Check(smhip_diffusion_run(m_ctx, (float *)m_dev[B_COST], (float *)m_dev[B_COST2],
                                      diff_beta > 0.0f ? (const float *)m_dev[B_COST0] : nullptr, m_mp.width,
                                      m_mp.height, m_disp_n, m_row_elems, diff_lambda, diff_beta, aggr_iter,
                                      &in_b));
        }
        break;
    case eBayesian:                                           // StcAggregate.cpp:320-330, tolerance mode
        if (verbose >= eVerboseProgress && aggr_iter > 0)
            fprintf(stderr, "- aggregating: Bayesian model, scale=%g, mu=%g, sigmaP=%g, epsP=%g\n",
                    diff_scale_cost, diff_mu, diff_sigmaP, diff_epsP);
        if (aggr_iter > 0) {
            DeviceBuffer(B_COST2, bytes);
            DeviceBuffer(B_COST0, bytes);
        }
        if (aggr_iter > 0)
            Check(smhip_bayesian_run(m_ctx, (float *)m_dev[B_COST], (float *)m_dev[B_COST2],
                                     (float *)m_dev[B_COST0], m_mp.width, m_mp.height, m_disp_n, m_row_elems,
                                     diff_scale_cost, diff_mu, diff_sigmaP, diff_epsP, aggr_iter, &in_b));
        break;
    default:
        throw CError("CStereoMatcher::Aggregate(): unknown aggregation function");
    }
    if (in_b)
        SwapCost();
    if (verbose >= eVerboseProgress && aggr_iter > 3)
        fprintf(stderr, "...  (doing %d iterations total)\n", aggr_iter);
    if (aggr_minfilter > 1) {
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- min filter (shiftable windows): filter size %d\n", aggr_minfilter);
        if (MinWtaFusable()) {                                // + PadCosts + OptWTA, keys out
            size_t n_pix = (size_t)m_mp.width * m_mp.height;
            DeviceBuffer(B_KEYS, n_pix * sizeof(int64_t));
            Check(smhip_min_pad_wta_keys(m_ctx, (const float *)m_dev[B_COST], m_row_elems, &m_mp, aggr_minfilter,
                                         0, m_disp_n, (int64_t *)m_dev[B_KEYS]));
            Check(smhip_keys_decode(m_ctx, (const int64_t *)m_dev[B_KEYS], m_mp.width, m_mp.height,
                                    (int32_t *)m_dev[B_DISP], (float *)m_dev[B_BEST]));
            m_keys_done = true;
            m_have_mind = false;
            return;
        }
        AggrMin();
    }
    PadCosts();
    m_have_mind = false;
    if (aggr_subpixel) {                                      // StcAggregate.cpp:368-385
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- sub-pixel fit\n");
        AggrSubPixelFit();
    }
    if (aggr_collapse) {                                      // :388-400
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- collapse to integral DSI\n");
        AggrCollapse();
    }
    if (verbose >= eVerboseDumpFiles)                         // :403-404
        WriteCosts("reprojected/DSIa_%03d.pgm");
}

void CStereoMatcher::AggrSubPixelFit()
{
    size_t bytes = m_row_elems * m_mp.height * sizeof(float);
    DeviceBuffer(B_COST2, bytes);
    DeviceBuffer(B_MIND, bytes);
    DeviceBuffer(B_CERT, bytes);
    Check(smhip_subpixel_fit(m_ctx, (const float *)m_dev[B_COST], (float *)m_dev[B_COST2],
                             (float *)m_dev[B_MIND], (float *)m_dev[B_CERT], m_mp.width, m_mp.height,
                             m_disp_n, m_row_elems, m_match_outside));
    SwapCost();
    m_have_mind = true;
    m_cert_row_elems = m_row_elems;
}

void CStereoMatcher::AggrCollapse()
{
    if (m_disp_step >= 1.0f)
        return;                                               // StcAggregate.cpp:137-138
    int df = int(m_disp_step_inv + 0.5);
    if (df != m_disp_step_inv)
        throw CError("AggrCollapse:  disparity step %g is not a pure fraction", m_disp_step);
    int n2 = (disp_max - disp_min) + 1;
    size_t re2 = (size_t)m_mp.width * n2;
    size_t bytes2 = re2 * m_mp.height * sizeof(float);
    // collapsed cost and offsets go to the spare slots, then take the places of m_cost / m_sub_pixel_min
    // (m_sub_pixel_cert stays as it is: the reference does not collapse it, StcEvaluate.cpp:782-799 reads it later)
    DeviceBuffer(B_COST2, bytes2);
    DeviceBuffer(B_MIND2, bytes2);
    Check(smhip_collapse(m_ctx, (const float *)m_dev[B_COST], m_have_mind ? (const float *)m_dev[B_MIND] : nullptr,
                         m_mp.width, m_mp.height, m_disp_n, m_row_elems, df, n2, re2, disp_step,
                         (float *)m_dev[B_COST2], (float *)m_dev[B_MIND2]));
    SwapCost();
    std::swap(m_dev[B_MIND], m_dev[B_MIND2]);
    std::swap(m_dev_bytes[B_MIND], m_dev_bytes[B_MIND2]);
    m_have_mind = true;
    m_disp_step = m_disp_step_inv = 1.0f;                     // :145-146
    m_disp_n = n2;
    m_row_elems = re2;
}

void CStereoMatcher::OptWTA()
{
    Check(smhip_wta(m_ctx, (const float *)m_dev[B_COST], m_mp.width, m_mp.height, m_disp_n, m_row_elems,
                    (int32_t *)m_dev[B_DISP], (float *)m_dev[B_BEST]));
}

void CStereoMatcher::OptSO()
{
    // the smoothness weights come from the (pre-processed) reference image on the device
    Check(smhip_opt_so(m_ctx, (const float *)m_dev[B_COST], m_mp.width, m_mp.height, m_disp_n, m_row_elems,
                       (const uint8_t *)m_dev[B_REF], m_reference.RowBytes(), m_mp.n_bands, opt_smoothness,
                       opt_grad_thresh, opt_grad_penalty, (int32_t *)m_dev[B_DISP], (float *)m_dev[B_BEST]));
}

void CStereoMatcher::OptDP()
{
    if (m_disp_step != 1.0f)                                  // StcOptDP.cpp:112-114
        fprintf(stderr, "WARNING: dynamic programming will not work properly for m_disp_step != 1\n");
    Check(smhip_opt_dp(m_ctx, (const float *)m_dev[B_COST], m_mp.width, m_mp.height, m_disp_n, m_row_elems,
                       (const uint8_t *)m_dev[B_REF], m_reference.RowBytes(), m_mp.n_bands, opt_smoothness,
                       opt_grad_thresh, opt_grad_penalty, opt_occlusion_cost, (int32_t *)m_dev[B_DISP],
                       (float *)m_dev[B_BEST]));
}

void CStereoMatcher::Optimize()
{
    if (opt_fn == eDynamicProg) {                             // StcOptimize.cpp:217-227, StcOptDP.cpp
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- computing smoothness costs (smoothness = %g, grad_thresh = %g, penalty=%g)\n"
                            "- optimizing: Dynamic Programming (occlusion_cost = %d)\n",
                    opt_smoothness, opt_grad_thresh, opt_grad_penalty, opt_occlusion_cost);
        OptDP();
    } else if (opt_fn == eScanlineOpt) {                             // StcOptimize.cpp:229-238, StcOptSO.cpp
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- computing smoothness costs (smoothness = %g, grad_thresh = %g, penalty=%g)\n"
                            "- optimizing: Scanline Optimization\n",
                    opt_smoothness, opt_grad_thresh, opt_grad_penalty);
        OptSO();
    } else if (opt_fn == eWTA) {
        if (verbose >= eVerboseProgress)
            fprintf(stderr, "- optimizing: WTA (local minimum)\n");
        if (!m_used_fused && !m_keys_done)
            OptWTA();
    } else {
        throw CError("CStereoMatcher::Optimize(): only opt_fn 1 (WTA), 2 (dynamic programming) and "
                     "3 (scan-line optimisation) are on the device path");
    }
    // results to the host (disparity rows are 4*W bytes: already a multiple of 8 or padded here)
    size_t wbytes = (size_t)m_mp.width * 4;
    Check(smhip_memcpy2d_d2h(m_ctx, m_disparity.Data(), m_disparity.RowBytes(), m_dev[B_DISP], wbytes, wbytes,
                             m_mp.height));
    Check(smhip_memcpy2d_d2h(m_ctx, m_best_cost.Data(), m_best_cost.RowBytes(), m_dev[B_BEST], wbytes, wbytes,
                             m_mp.height));
    if (final_energy < 0.0f) {                                // StcOptimize.cpp:270-278
        float ed, en;
        ComputeEnergy(ed, en);
        final_energy = ed + en;
    }
}

// Smoothness weight between two reference pixels (StcOptimize.cpp:61-82)
static float NCost(const uchar *a, const uchar *b, int nB, float smoothness, float thresh, float penalty)
{
    float d2 = 0.0f;
    for (int k = 0; k < nB; k++) {
        float d = (float)(int(a[k]) - int(b[k]));
        d2 += d * d;
    }
    d2 /= (nB - (nB > 1));
    float s = smoothness;
    if (sqrtf(d2) < thresh)
        s *= penalty;
    return s;
}

void CStereoMatcher::ComputeEnergy(float &denergy, float &nenergy)
{
    // data term: raster-order float sum of the cost at the winner; neighbour term: smoothness
    // weight wherever adjacent labels differ (StcGraphCut.cpp:46-70, StcOptimize.cpp:103-116)
    int W = m_mp.width, H = m_mp.height, nB = m_mp.n_bands;
    float dSum = 0.0f, nSum = 0.0f;
    for (int y = 0; y < H; y++) {
        const int *cur = m_disparity.Row(y);
        const int *nxt = m_disparity.Row(y + (y < H - 1));
        const float *bc = m_best_cost.Row(y);
        const uchar *i0 = m_reference.Row(y), *i1 = m_reference.Row(y + (y < H - 1));
        for (int x = 0; x < W; x++) {
            dSum += bc[x];
            if (y < H - 1 && cur[x] != nxt[x])
                nSum += NCost(i0 + x * nB, i1 + x * nB, nB, opt_smoothness, opt_grad_thresh, opt_grad_penalty);
            if (x < W - 1 && cur[x] != cur[x + 1])
                nSum += NCost(i0 + x * nB, i0 + (x + 1) * nB, nB, opt_smoothness, opt_grad_thresh,
                              opt_grad_penalty);
        }
    }
    denergy = dSum;
    nenergy = nSum;
}

void CStereoMatcher::Refine()
{
    float d_offset = (float)disp_min;
    if (m_used_fused || !refine_subpix || m_disp_n < 3) {
        // m_float_disparity = disp * step + disp_min (StcRefine.cpp:34-43 via Convert.cpp:31-52)
        bool scale_offset = (m_disp_step != 1.0f) || (d_offset != 0.0f);
        for (int y = 0; y < m_mp.height; y++) {
            const int *src = m_disparity.Row(y);
            float *dst = m_float_disparity.Row(y);
            for (int x = 0; x < m_mp.width; x++)
                dst[x] = scale_offset ? src[x] * m_disp_step + d_offset : (float)src[x];
        }
        return;
    }
    // sub-pixel fit on the resident DSI: offsets of the aggregation-stage fit (:50-68) or a
    // parabola through the winner's neighbours (:75-98)
    bool from_fit = aggr_subpixel || (aggr_collapse && disp_step < 1.0f);
    if (from_fit && !m_have_mind)
        throw CError("CStereoMatcher::Refine(): no sub-pixel minima available");
    if (!from_fit && verbose >= eVerboseProgress)
        fprintf(stderr, "- refining: subpixel parabola fit\n");
    size_t wbytes = (size_t)m_mp.width * 4;
    DeviceBuffer(B_FDISP, wbytes * m_mp.height);
    Check(smhip_refine(m_ctx, (const int32_t *)m_dev[B_DISP], (const float *)m_dev[B_COST],
                       from_fit ? (const float *)m_dev[B_MIND] : nullptr, m_mp.width, m_mp.height, m_disp_n,
                       m_row_elems, m_disp_step, d_offset, from_fit ? 1 : 2, (float *)m_dev[B_FDISP]));
    Check(smhip_memcpy2d_d2h(m_ctx, m_float_disparity.Data(), m_float_disparity.RowBytes(), m_dev[B_FDISP],
                             wbytes, wbytes, m_mp.height));
}
