/*
 * sm_oracle.h -- CPU restatement of the StereoMatch cost-volume front end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.  The product path
 * (stereomatch_amd/, libsmhip.so) never links or calls it.
 *
 * Parity pinning: the reference cannot be built in this image under this
 * round's rules (its CPU sources include CUDA SDK headers the image lacks, and
 * stand-in headers are not allowed), so oracle/_ref does not exist.  The
 * restatement is pinned by the outputs of the reference CPU path that the survey
 * stage recorded in BASELINE.md section 2 (five tsukuba disparity-map md5s and
 * the SSD09 final_energy); tests/test_oracle_pins.py checks all of them.  Every
 * configuration outside those anchors is "parity unpinned" (it shares the
 * operators the anchors exercise, but no reference output exists for it).
 *
 * Each function cites the reference file:line whose arithmetic it restates
 * (paths relative to /root/reference).  All float arithmetic is IEEE binary32,
 * left to right, no FMA contraction (compile with -ffp-contract=off).
 *
 * Layout everywhere: DSI C[y][x][d], d innermost, row stride `row_elems` floats
 * (Image.h:160-164; row_elems*4 = ((4*D*W+7)&~7) for reference-shaped hosts).
 */
#ifndef SM_ORACLE_H
#define SM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SMO_AD = 1, SMO_SD = 2 };             /* EStereoMatchFn, StereoParameters.h:33-40 */
enum { SMO_LINEAR = 1, SMO_CUBIC = 3 };      /* EStereoInterpFn, StereoParameters.h:42-48 */
enum { SMO_BOX = 1, SMO_BINOMIAL = 2, SMO_DIFFUSION = 3, SMO_MEMBRANE = 4, SMO_BAYESIAN = 5 }; /* :81-97 */

typedef struct smo_params {
    /* image geometry */
    int width, height, n_bands;
    /* derived in ComputeCorrespondence, StereoMatcher.cpp:60-82 */
    int disp_min, disp_num, disp_den, disp_n, frame_diff_sign;
    /* CStereoMatchParameters, StereoParameters.h:60-79 */
    int match_fn, match_interp, match_max, match_interval, match_interpolated;
    /* CStereoAggregateParameters, StereoParameters.h:99-123 */
    int aggr_fn, aggr_window_size, aggr_iter, aggr_minfilter;
    float diff_lambda, diff_beta;
    /* smoothness term of final_energy, StereoParameters.h:153-173 */
    float opt_smoothness, opt_grad_thresh, opt_grad_penalty;
    /* depth-map scaling, StereoMatcher.cpp:177 */
    float disp_step, disp_scale;
    /* sub-pixel stages, StereoParameters.h:118-120,179 */
    int aggr_subpixel, aggr_collapse, refine_subpix;
    /* StereoParameters.h:57-58 */
    int preproc_blur_iter;
    /* EStereoOptimizeFn, StereoParameters.h:125-136: 1 = WTA (and any other value), 3 = scan-line optimisation */
    int opt_fn;
    /* Bayesian diffusion, StereoParameters.h:112-115 */
    float diff_scale_cost, diff_mu, diff_sigmaP, diff_epsP;
    int opt_occlusion_cost;              /* StereoParameters.cpp:90 (20) */
} smo_params;

/* StcRawCosts.cpp:364-367 */
float smo_match_outside(const smo_params *p);

/* StcRawCosts.cpp:327-435 (RawCostsCPU) with :67-105, :107-128, :130-218 */
void smo_raw_costs(const uint8_t *ref, const uint8_t *mtc, long img_row_bytes,
                   const smo_params *p, float *cost, long row_elems);

/* BoxFilter.cpp:173-239 (CPU branch), :70-170; replicate border only */
void smo_box_filter(float *cost, float *tmp, int w, int h, int nb, long row_elems,
                    int window, int average);
/* the same with EBorderMode 1 replicate / 2 reflect / 3 cyclic per pass (BoxFilter.cpp:29-65) */
void smo_box_filter_border(float *cost, float *tmp, int w, int h, int nb, long row_elems, int window,
                           int average, int border_x, int border_y);

/* MinFilter.cpp:122-202 (doMax = false); replicate border */
void smo_min_filter(float *cost, float *tmp, int w, int h, int nb, long row_elems,
                    int window);

/* Convolve.cpp:166-254 with ConvolveKernel_14641 (:290,302-303), scale 1 offset 0 */
void smo_convolve_14641(float *cost, float *tmp, int w, int h, int nb, long row_elems);

/* StcDiffusion.cpp:28-137; one iteration src -> dst (cost0 used when beta > 0) */
void smo_diffusion_iter(const float *src, float *dst, const float *cost0,
                        int w, int h, int nb, long row_elems, float lambda, float beta);

/* StcRawCosts.cpp:437-476 */
void smo_pad_costs(float *cost, long row_elems, const smo_params *p);

/* StcOptimize.cpp:26-56 */
void smo_wta(const float *cost, int w, int h, int nb, long row_elems,
             int32_t *disp, float *best_cost);

/* StcAggregate.cpp:255-365 (sub-pixel fit / collapse: separate calls below) */
void smo_aggregate(float *cost, long row_elems, const smo_params *p);

/* StcOptimize.cpp:84-156 + StcGraphCut.cpp:36-75: data + smoothness energy */
float smo_final_energy(const float *cost, long row_elems, const int32_t *disp,
                       const uint8_t *ref, long img_row_bytes, const smo_params *p);

/* StcRefine.cpp:34-39 + StereoMatcher.cpp:177 (Convert.cpp:25-53) */
void smo_depth_image(const int32_t *disp, int n, const smo_params *p, uint8_t *depth);

/* StereoMatcher.cpp:155-177: RawCosts -> Aggregate -> OptWTA -> Refine.
 * disp/best_cost/depth: w*h each (any may be NULL); returns final_energy.
 * Single-threaded, like the reference. */
float smo_compute_correspondence(const uint8_t *ref, const uint8_t *mtc, long img_row_bytes,
                                 const smo_params *p, int32_t *disp, float *best_cost,
                                 uint8_t *depth);

/* StcPreProcess.cpp:49-55 (in place) */
void smo_preprocess_blur(uint8_t *img, long row_bytes, int w, int h, int nb, int iters);

/* StcAggregate.cpp:66-132 (in place; mind/cert: same shape as cost) */
void smo_subpixel_fit(float *cost, float *mind, float *cert, int w, int h, int nb,
                      long row_elems, float outside);

/* StcAggregate.cpp:134-230 (mind1 NULL when aggr_subpixel is off) */
void smo_collapse(const float *cost1, const float *mind1, int w, int h, int nb1, long row_elems1,
                  int df, int nb2, long row_elems2, float disp_step, float *cost2, float *mind2);

/* StcRefine.cpp:28-100; mode 0 plain, 1 offsets from mind, 2 parabola fit at the winner */
void smo_refine(const float *cost, const float *mind, const int32_t *disp, int w, int h, int nb,
                long row_elems, float step, int disp_min, int mode, float *fdisp);

/* StcEvaluate.cpp:760-833 (ComputeMatchQuality): final cost and certainty at the winner */
void smo_match_quality(const float *cost, const float *cert, const float *fdisp, int w, int h, int nb,
                       long row_elems, long cert_row_elems, int disp_min, float step_inv,
                       int quality_mode, int use_fit, int norm_bands, int take_sqrt,
                       float *fcost, float *fcert);

/* StereoMatcher.cpp:177 */
void smo_depth_from_float(const float *fdisp, int n, const smo_params *p, uint8_t *depth);

/* the same pipeline, also returning m_float_disparity (fdisp may be NULL) */
float smo_compute_correspondence_f(const uint8_t *ref, const uint8_t *mtc, long img_row_bytes,
                                   const smo_params *p, int32_t *disp, float *best_cost,
                                   uint8_t *depth, float *fdisp);

/* Evaluator (StcEvaluate.cpp:42-587, ForwardWarp Warp1D.cpp:25-128): occlusion / textureless /
 * discontinuity masks from the ground truth and the reference image, then RMS and bad-pixel
 * statistics of fdisp against the truth.  Order of the six entries: all, nonocc, occ, textured,
 * textureless, discont (the rms_error_* / bad_pixels_* output parameters,
 * StereoParameters.cpp:217-229).  truth: uchar [h][w] (tight rows); ref: the un-preprocessed
 * reference image. */
typedef struct smo_eval {
    int ignore_border;
    float bad_thresh, disp_gap;
    int textureless_width;
    float textureless_thresh;
    int discont_width;
} smo_eval;
typedef struct smo_eval_out {
    float rms[6], bad[6], fraction_matched;
} smo_eval_out;
void smo_evaluate(const float *fdisp, const uint8_t *truth, const uint8_t *ref, long img_row_bytes,
                  const smo_params *p, const smo_eval *e, int verbose_summary, smo_eval_out *out);
/* the same evaluation, handing out m_occlusion and m_textureless (w x h bytes each) */
void smo_evaluate_masks(const float *fdisp, const uint8_t *truth, const uint8_t *ref, long img_row_bytes,
                        const smo_params *p, const smo_eval *e, uint8_t *occlusion, uint8_t *textureless);
/* Histogram1D as ComputeMatchQuality draws it (Histogram1D.cpp:32-139, StcEvaluate.cpp:848-861): parity
 * unpinned (the reference holds no such figure) */
void smo_cost_histogram(const float *img, const uint8_t *mask, int w, int h, float min_val, float max_val,
                        int n_bins, int height, float *v_scale, uint8_t *figure);

/* Prediction (re-projection) error of one frame (ComputePredictionErrors, StcEvaluate.cpp:588-758;
 * ForwardWarp / InverseWarp, Warp1D.cpp:25-257).  ref = the un-preprocessed reference image, frame =
 * that frame's input image (tight rows, nb bands each), shift = (f - frame_ref) / |frame_match -
 * frame_ref|.  Writes the re-projected image (what resampled_file receives), returns the RMS error
 * over visible pixels and the visible fraction.  No reference output pins it: parity unpinned. */
typedef struct smo_predict {
    int predict_type;                   /* 1 = forward warp of ref, 2 = inverse warp of frame */
    int lin_interp;
    float disp_gap;
    int predict_diff;
    int empty_color;
    float partial_shuffle;
} smo_predict;
float smo_prediction_error(const uint8_t *ref, const uint8_t *frame, int w, int h, int nb,
                           const float *fdisp, float shift, const smo_predict *q,
                           uint8_t *resampled, float *fraction_visible);

/* OptSO (StcOptSO.cpp:30-154), the reference's O(D^2)-per-pixel scan, with the horizontal
 * smoothness weights of StcOptimize.cpp:61-116 computed from `ref` */
void smo_opt_dp(const float *cost, int w, int h, int nd, long row_elems, const uint8_t *ref,
                long img_row_bytes, const smo_params *p, int32_t *disp);
void smo_opt_so(const float *cost, int w, int h, int nd, long row_elems, const uint8_t *ref,
                long img_row_bytes, const smo_params *p, int32_t *disp);

/* AggrBayesian (StcDiffusion.cpp:139-308), one iteration in place; cost2 = scratch (Es), cost0 =
 * E0 (written when iter == 0).  Tolerance mode: see the note in sm_oracle.c. */
void smo_bayesian_iter(float *cost, float *cost2, float *cost0, int w, int h, int nd, long row_elems,
                       float scale_cost, float mu, float sigmaP, float epsP, int iter);

/* 1, or the OpenMP thread count of the -fopenmp build (_build/libsm_oracle_omp.so: the same
 * source, rows / column blocks spread over threads, identical results) */
int smo_threads(void);

#ifdef __cplusplus
}
#endif
#endif
