/*
 * smhip.h -- C-ABI of the MI355X (gfx950) cost-volume front end of StereoMatch.
 *
 * This is the drop-in boundary: the functions below are what the reference's device
 * layer (CudaRawCosts.h, CudaBoxFilter.h, CudaConvolve.h, CudaUtilities.h) would bind
 * to.  Plain C, pointers and sizes only.  All `d_*` pointers are DEVICE pointers
 * (from smhip_malloc, or any hipMalloc'd / torch-allocated buffer on the context's
 * device).  Every call is enqueued on the context's stream (default: the null stream);
 * calls return after enqueueing -- use smhip_sync() before reading results through
 * anything other than smhip_memcpy_d2h*() (which synchronise themselves).
 *
 * Return value: SMHIP_OK (0) or a negative error code; smhip_last_error() returns a
 * message for the last failing call of the calling thread.  Host misuse that the
 * reference reports with `throw CError(...)` (e.g. BoxFilter.cpp:179-180) comes back
 * as SMHIP_EINVAL; HIP runtime failures (reference: GPUERRORCHECK -> exit(1),
 * CudaUtilities.h:32-40) come back as SMHIP_EHIP and never terminate the process.
 *
 * Threading: like the reference (single host thread) a context is not
 * thread-safe; use one context per host thread.
 *
 * Data layout (Image.h:160-164): images are uint8 [H][W][n_bands] with a row pitch in
 * bytes; the disparity-space image (DSI) is float [H][W][D], D innermost, row stride
 * `row_elems` floats (>= W*D; the reference's CImage rounds rows up to 8 bytes,
 * Image.cpp:97-98).  All float arithmetic follows the reference CPU path's operation
 * order with no FMA contraction, so results are bit-identical to it.
 */
#ifndef SMHIP_H
#define SMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMHIP_OK        0
#define SMHIP_EINVAL   (-1)   /* bad argument / unsupported parameter combination */
#define SMHIP_EHIP     (-2)   /* HIP runtime error */
#define SMHIP_ENOMEM   (-3)   /* device allocation failed */

/* images and DSIs of more rows are refused (SMHIP_EINVAL) */
#define SMHIP_MAX_ROWS 65535

typedef struct smhip_ctx smhip_ctx;

/* Replaces LineProcessStruct (CudaRawCosts.h:97-114): the parameters RawCosts derives
 * in ComputeCorrespondence (StereoMatcher.cpp:60-82) and StcRawCosts.cpp:354-367. */
typedef struct smhip_match_params {
    int width, height, n_bands;         /* image shape; n_bands 1 or 4 (StereoIO.cpp:76-81) */
    int disp_min;                       /* smallest disparity */
    int disp_num, disp_den;             /* disparity step = num/den */
    int disp_n;                         /* number of disparity levels D */
    int frame_diff_sign;                /* +1 / -1 (StereoMatcher.cpp:60-61) */
    int match_fn;                       /* 1 = AD, 2 = SD (EStereoMatchFn) */
    int match_interp;                   /* 1 = linear, 3 = cubic (EStereoInterpFn) */
    int match_max;                      /* truncation */
    int match_interval;                 /* Birchfield-Tomasi */
    int match_interpolated;             /* intervals on the reference line too */
} smhip_match_params;

/* ---- context (replaces prepareDevice, CudaUtilities.h:90-91 / .cu:93-127) ---- */
int smhip_init(int device, smhip_ctx **ctx);
int smhip_shutdown(smhip_ctx *ctx);
const char *smhip_last_error(void);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = null stream.
 * The context's workspaces are shared by all its calls whatever stream they ran on, so changing the
 * stream first waits for the work queued on the old one. */
int smhip_set_stream(smhip_ctx *ctx, void *hip_stream);
/* Tuning / test knobs of a context (no counterpart in the reference; nothing in the library reads the
 * environment).  value < 0 restores the library's own policy.  Waits for the context's queued work. */
enum smhip_option {
    SMHIP_OPT_NARROW = 0,     /* 1: general-shape fused kernels for slabs of <= 32 levels */
    SMHIP_OPT_TYSHIFT,        /* log2 of the band height of the fused path's private slab, 0..8 */
    SMHIP_OPT_CHUNK,          /* 1 / 0: pass A stages rows in chunks / whole (integer kernels) */
    SMHIP_OPT_BT_PREPASS,     /* 1: Birchfield-Tomasi intervals from the pre-pass kernel, always */
    SMHIP_OPT_TSEG,           /* rows per workgroup of the LDS-tiled filters, 8..4096 (min + WTA keys: 8..64) */
    SMHIP_OPT_MAXK,           /* iterations per pass of the multi-iteration filters, 1..4 */
    SMHIP_OPT_FLOAT_A,        /* 0: pass A by the integer kernels also where the float-lane kernel applies */
    SMHIP_OPT_KEYBUF,         /* key groups (of ~64 rows) pass B collects in LDS before it stores them; 0: none */
    SMHIP_OPT_STORE_POLICY,   /* cache policy of the float-lane pass A's slab stores: 0 plain, 16 sc1, 18 sc1 + nt (18 also makes pass B's key stores non-temporal, the library's choice from 16 M pixels on) */
    SMHIP_OPT_PAIRED,         /* slabs of 65..128 levels: 0 one column per wave, 1 columns in pairs at any width */
    SMHIP_OPT_A_LDS_KB,       /* least LDS per workgroup of the float-lane pass A, KiB (<= 80): bounds the workgroups per CU; 0 none */
    SMHIP_OPT_B_WAVES,        /* pass B waves resident per CU, 6..16 (even; bounded through the LDS a workgroup asks for); 0 no bound */
    SMHIP_OPT_STAGE_LDS_KB,   /* LDS floor of the stage kernels, KiB, 0..80 (bounds the workgroups per CU) */
    SMHIP_OPT_B_AHEAD,        /* pass B of slabs up to 128 levels: 1 = one block of rows in flight, 2 = two; default two, four on up to 2048 columns */
    SMHIP_OPT_MULTI_TXT,      /* pixels per tile of the multi-iteration binomial pass, halo included: 64 or 128 (default: 128 from 1 GiB on) */
    SMHIP_OPT_TAP_SLOW,       /* 1: the binomial passes always take the plain ten-operation tap sum (tests) */
    SMHIP_OPT_MULTI_PX,       /* pixels-on-lanes kernels in the _run calls (width % 4 == 0): binomial default on, 0 = the LDS-tiled form; diffusion > 0 = always, default from 64 MiB on, 0 = never */
    SMHIP_OPT_ITER_FUSED,     /* smhip_fused_iter_wta_keys: 0 = always the stage calls on a private DSI (tests) */
    SMHIP_OPT_ITER_GROUPS,    /* smhip_fused_iter_wta_keys: level groups (workgroups along the levels) of the keys kernels, 1..16 */
    SMHIP_OPT_A_NW,           /* float-lane pass A on 256-level chunks: 0 = two-wave workgroups, 1 = one-wave workgroups of 128 levels, 4 = one wave of four levels per lane (window 9) */
    SMHIP_OPT_B_LRED,         /* pass B of slabs up to 128 levels: 1 = the wave minimum of a group of rows through LDS at any width, 0 = the per-row DPP chain; default: LDS on up to 2048 columns */
    SMHIP_OPT_ITER_DIRECT,    /* smhip_fused_iter_wta_keys: 1 = one level tile per workgroup with long row segments and a key image per tile (2..64 tiles), 0 = level tiles looped inside 64-row segments */
    SMHIP_OPT_COUNT
};
int smhip_set_option(smhip_ctx *ctx, int option, int value);
int smhip_sync(smhip_ctx *ctx);
/* bytes of device workspace the context currently holds (temporaries of the filters) */
size_t smhip_workspace_bytes(const smhip_ctx *ctx);

/* ---- memory (replaces Allocate/Copy/FreeGPUMemory, CudaUtilities.cu:49-77) ---- */
int smhip_malloc(smhip_ctx *ctx, size_t bytes, void **d_ptr);
int smhip_free(smhip_ctx *ctx, void *d_ptr);
int smhip_memcpy_h2d(smhip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int smhip_memcpy_d2h(smhip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* device-to-device on the context's stream (CopyPixels(m_cost, m_cost0), StcAggregate.cpp:262) */
int smhip_memcpy_d2d(smhip_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* pitched copies, for hosts whose rows are padded (Image.cpp:97-98) */
int smhip_memcpy2d_h2d(smhip_ctx *ctx, void *d_dst, size_t d_pitch, const void *h_src,
                       size_t h_pitch, size_t width_bytes, size_t rows);
int smhip_memcpy2d_d2h(smhip_ctx *ctx, void *h_dst, size_t h_pitch, const void *d_src,
                       size_t d_pitch, size_t width_bytes, size_t rows);

/* ---- stage operators ---- */

/* PreProcess (StcPreProcess.cpp:49-55): n_iter times the separable uchar blur (121 in x, 14641 in
 * y, replicate border, truncating conversion) on a device image, in place. */
int smhip_preprocess_blur(smhip_ctx *ctx, uint8_t *d_img, size_t img_pitch, int w, int h,
                          int n_bands, int n_iter);

/* m_match_outside (StcRawCosts.cpp:365-367) */
float smhip_match_outside(const smhip_match_params *p);

/* Replaces LineProcess (CudaRawCosts.h:145, called from RawCostsGPU StcRawCosts.cpp:305);
 * arithmetic of RawCostsCPU/MatchLine/BirchfieldTomasiMinMax/InterpolateLine
 * (StcRawCosts.cpp:67-218, 327-435).  disp_den > 1 (disp_step = 1/den) up-samples both
 * lines den times (linear or cubic) and, with match_interpolated, averages 2*(den/2)+1
 * sample costs per pixel, as the reference does. */
int smhip_raw_costs(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                    size_t img_pitch, const smhip_match_params *p,
                    float *d_cost, size_t row_elems);

/* Replaces CudaBoxFilterXY (CudaBoxFilter.h:31-32, called from BoxFilter.cpp:197) with the
 * semantics of the CPU BoxFilter<float> (BoxFilter.cpp:173-239): any window, running sum,
 * replicate border, in place.  x_width != y_width is rejected like BoxFilter.cpp:179-180.
 * Windows larger than the image width or height are refused (SMHIP_EINVAL): there the reference's
 * running sum reads beyond the line (BoxFilter.cpp:114-124) or trips its double-wrapping asserts. */
int smhip_box_filter(smhip_ctx *ctx, float *d_cost, int w, int h, int n_bands,
                     size_t row_elems, int x_width, int y_width, int average);

/* The same with the reference's other border modes (EBorderMode, Image.h:98-104; borderIndex,
 * BoxFilter.cpp:29-65): 1 replicate, 2 reflect, 3 cyclic, for the row pass (the source image's mode,
 * BoxFilter.cpp:219) and the column pass (the destination's, :231).  0 (zero padding) is refused: the
 * reference throws on it as soon as a window crosses the border.  The hot path only uses replicate. */
int smhip_box_filter_border(smhip_ctx *ctx, float *d_cost, int w, int h, int n_bands, size_t row_elems,
                            int window, int average, int border_x, int border_y);

/* MinFilter<float> (MinFilter.h:31, MinFilter.cpp:122-209), replicate border, in place */
int smhip_min_filter(smhip_ctx *ctx, float *d_cost, int w, int h, int n_bands,
                     size_t row_elems, int x_width, int y_width);
/* the same, d_src -> d_dst (distinct buffers, as MinFilter(src, dst, ...) allows, MinFilter.h:31):
 * one pass over the volume instead of filter + copy back; hosts swap m_cost / m_cost2 */
int smhip_min_filter_to(smhip_ctx *ctx, const float *d_src, float *d_dst, int w, int h,
                        int n_bands, size_t row_elems, int x_width, int y_width);

/* Replaces CudaConvolveXY(..., BINOMIAL14641) (CudaConvolve.h:57-58, Convolve.cpp:214-215)
 * with the semantics of the CPU ConvolveSeparable (Convolve.cpp:166-254): kernel 14641 in
 * x then y, replicate border, scale 1, offset 0, in place. */
int smhip_convolve_14641(smhip_ctx *ctx, float *d_cost, int w, int h, int n_bands,
                         size_t row_elems);
/* the same, d_src -> d_dst (distinct buffers, ConvolveSeparable(src, dst, ...), Convolve.h:55-60) */
int smhip_convolve_14641_to(smhip_ctx *ctx, const float *d_src, float *d_dst, int w, int h,
                            int n_bands, size_t row_elems);

/* One AggrDiffusion iteration (StcDiffusion.cpp:28-137) d_src -> d_dst (distinct buffers);
 * d_cost0 (the un-aggregated DSI) is read only when beta > 0 (membrane model). */
int smhip_diffusion_iter(smhip_ctx *ctx, const float *d_src, float *d_dst,
                         const float *d_cost0, int w, int h, int n_bands,
                         size_t row_elems, float lambda, float beta);

/* One AggrBayesian iteration (StcDiffusion.cpp:139-308): d_cost = E (in place), d_cost2 = Es
 * (scratch the size of the DSI), d_cost0 = E0.  iter == 0 first scales E by scale_cost and copies
 * it to E0 (:182-191).  TOLERANCE MODE -- the only operator here that is not bit-exact: exp and
 * log come from the device's math library (about 1 ulp from the host's), everything else follows
 * the reference's order of operations in the reference's precisions. */
#define SMHIP_BAYES_MAX_LEVELS 512
int smhip_bayesian_iter(smhip_ctx *ctx, float *d_cost, float *d_cost2, float *d_cost0, int w, int h,
                        int n_bands, size_t row_elems, float scale_cost, float mu, float sigmaP,
                        float epsP, int iter);

/* The aggr_iter loop of AggrBayesian as one call (StcAggregate.cpp:265-343 with :320-330): the same
 * values as n_iter calls of smhip_bayesian_iter, but between iterations only the smoothed energies
 * are stored (12 instead of 20 bytes per voxel and iteration).  *result_in_b = 1 when the energies
 * E end up in d_cost2 instead of d_cost (the host swaps m_cost / m_cost2, as after smhip_diffusion_run). */
int smhip_bayesian_run(smhip_ctx *ctx, float *d_cost, float *d_cost2, float *d_cost0, int w, int h,
                       int n_bands, size_t row_elems, float scale_cost, float mu, float sigmaP,
                       float epsP, int n_iter, int *result_in_b);

/* n_iter iterations of AggrDiffusion / of the binomial filter (the aggr_iter loop of
 * StcAggregate.cpp:265-343) with up to four iterations per pass over the DSI (temporal blocking,
 * csrc/smhip_multi.h); bit-identical to n_iter single calls.  d_a holds the input, d_a and d_b
 * (distinct, same shape) alternate as source and destination of the passes; *result_in_b says
 * which one holds the result (hosts swap m_cost / m_cost2 accordingly, StcDiffusion.cpp:58-63). */
int smhip_diffusion_run(smhip_ctx *ctx, float *d_a, float *d_b, const float *d_cost0, int w, int h,
                        int n_bands, size_t row_elems, float lambda, float beta, int n_iter,
                        int *result_in_b);
int smhip_convolve_14641_run(smhip_ctx *ctx, float *d_a, float *d_b, int w, int h, int n_bands,
                             size_t row_elems, int n_iter, int *result_in_b);

/* PadCosts (StcRawCosts.cpp:437-476), any disp_den */
int smhip_pad_costs(smhip_ctx *ctx, float *d_cost, size_t row_elems,
                    const smhip_match_params *p);

/* OptWTA (StcOptimize.cpp:26-56); d_best (may be NULL) receives m_cost at the winner, which
 * Optimize needs for final_energy (StcOptimize.cpp:270-278). */
int smhip_wta(smhip_ctx *ctx, const float *d_cost, int w, int h, int n_bands,
              size_t row_elems, int32_t *d_disp, float *d_best);

/* ---- along-d stages behind the aggregated DSI (disp_step < 1 / refine_subpix runs) ---- */

/* AggrSubPixelFit (StcAggregate.cpp:66-132): d_src -> d_dst (distinct), plus m_sub_pixel_min and
 * m_sub_pixel_cert, all [h][row_elems]; match_outside marks invalid levels (:98-101). */
int smhip_subpixel_fit(smhip_ctx *ctx, const float *d_src, float *d_dst, float *d_min,
                       float *d_cert, int w, int h, int n_bands, size_t row_elems,
                       float match_outside);

/* AggrCollapse (StcAggregate.cpp:134-230): n_bands1 fractional levels (step 1/df) -> n_bands2 =
 * (n_bands1 - 1)/df + 1 integer levels; d_min1 is m_sub_pixel_min or NULL when aggr_subpixel is
 * off; disp_step is the fractional step before the collapse (:186,:191). */
int smhip_collapse(smhip_ctx *ctx, const float *d_cost1, const float *d_min1, int w, int h,
                   int n_bands1, size_t row_elems1, int df, int n_bands2, size_t row_elems2,
                   float disp_step, float *d_cost2, float *d_min2);

/* Refine (StcRefine.cpp:28-100): d_fdisp = disp_step * d + disp_min, then mode 1: + the
 * offsets of m_sub_pixel_min (:50-68), mode 2: + the parabola fit through the winner's
 * neighbours in d_cost (:75-98); mode 0 or n_bands < 3: ScaleAndOffset only. */
int smhip_refine(smhip_ctx *ctx, const int32_t *d_disp, const float *d_cost, const float *d_min,
                 int w, int h, int n_bands, size_t row_elems, float disp_step, float disp_min,
                 int mode, float *d_fdisp);

/* ComputeMatchQuality (StcEvaluate.cpp:760-833): d_final_cost = aggregated cost at the winner, taken from
 * the parabola through its neighbours at the fractional offset of d_fdisp (quality_mode = eval_match_quality;
 * 2: no interpolation), divided by norm_bands (1, or 3 for colour: :774) and square-rooted when take_sqrt
 * (match_fn == eSD, :830); d_certainty = the parabola's curvature, 0 where it is degenerate.  use_fit
 * (aggr_subpixel): both are read from the fitted volumes d_cost / d_cert instead (:795-800). */
int smhip_match_quality(smhip_ctx *ctx, const float *d_cost, const float *d_cert, const float *d_fdisp,
                        int w, int h, int n_bands, size_t row_elems, size_t cert_row_elems, float disp_min,
                        float disp_step_inv, int quality_mode, int use_fit, int norm_bands, int take_sqrt,
                        float *d_final_cost, float *d_certainty);

/* OptSO (StcOptSO.cpp:30-154): per scan-line, the path through the DSI that minimises data cost +
 * Potts smoothness, with the horizontal weights of ComputeSmoothnessCosts (StcOptimize.cpp:61-116)
 * taken from the reference image.  Same winners as the reference's O(D^2) scan (first strict
 * minimum everywhere, COST_MAX 1e6 as in StereoMatcher.h:57) in O(D) per pixel.  d_best (may be
 * NULL) receives m_cost at the chosen level.  n_bands (levels) <= 1024. */
int smhip_opt_so(smhip_ctx *ctx, const float *d_cost, int w, int h, int n_bands, size_t row_elems,
                 const uint8_t *d_ref, size_t img_pitch, int img_bands, float opt_smoothness,
                 float opt_grad_thresh, float opt_grad_penalty, int32_t *d_disp, float *d_best);

/* OptDP (StcOptDP.cpp:110-332): per scan-line dynamic programming over matched / left-occluded /
 * right-occluded cells with occlusion cost opt_occlusion_cost and the horizontal smoothness weights
 * of ComputeSmoothnessCosts on the occluded -> matched transitions; occluded pixels are then filled
 * from the left (:78-107).  Same float operations in the same order as the reference (bit-exact).
 * d_best (may be NULL) receives m_cost at the chosen level.  n_bands (levels) <= 1024. */
int smhip_opt_dp(smhip_ctx *ctx, const float *d_cost, int w, int h, int n_bands, size_t row_elems,
                 const uint8_t *d_ref, size_t img_pitch, int img_bands, float opt_smoothness,
                 float opt_grad_thresh, float opt_grad_penalty, int opt_occlusion_cost,
                 int32_t *d_disp, float *d_best);

/* ---- fused path: RawCosts -> AggrBox (1 iteration) -> PadCosts -> OptWTA ----
 * Same results as the four stage calls, without materialising the raw or the aggregated
 * DSI.  Works on the disparity slab [d_first, d_first + d_count) of the D = p->disp_n
 * levels (independent per level: StcRawCosts.cpp:405-416, BoxFilter.cpp:119-125) and
 * emits one signed 64-bit key per pixel:
 *     key = (ordered_bits(best cost) << 32) | d        (d = global level index)
 * ordered_bits() maps float order to int32 order, so min(key) over slabs (e.g. an
 * int64 MIN all-reduce across GPUs) is OptWTA's winner with its lowest-d tie break.
 * d_keys holds width*height keys, COLUMN-MAJOR: the key of pixel (x, y) is d_keys[x*height + y]
 * (each wave owns a pixel column and stores runs of consecutive rows).
 * Integer-pel steps only (disp_den == 1); fractional steps go through the stage calls. */
int smhip_fused_box_wta_keys(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                             size_t img_pitch, const smhip_match_params *p, int window,
                             int d_first, int d_count, int64_t *d_keys);
/* smhip_fused_box_wta_keys for a caller that walks a level range in several calls (a rank's share of a sharded
 * volume in calls of bounded workspace): d_keys already holds the keys of other levels and receives the
 * element-wise minimum with this call's -- the merge OptWTA's scan makes between levels
 * (StcOptimize.cpp:43-52), folded into the store of pass B instead of a separate pass over two key images.
 * The first call of a frame is smhip_fused_box_wta_keys. */
int smhip_fused_box_wta_keys_merge(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                                   size_t img_pitch, const smhip_match_params *p, int window,
                                   int d_first, int d_count, int64_t *d_keys);

/* The first half of the above with the aggregated DSI as the result: RawCosts + one AggrBox of
 * levels [d_first, d_first + d_count) written to d_cost[h][row_elems] (d_count levels per pixel,
 * NOT yet padded: AggrMin / PadCosts / ... follow as in StcAggregate.cpp:351-400).  12 bytes per
 * voxel of HBM traffic instead of the 20 of smhip_raw_costs + smhip_box_filter.  Needs
 * d_count >= 33, d_count % 4 == 0, row_elems % 4 == 0, 16-byte aligned d_cost and disp_den == 1
 * (SMHIP_EINVAL otherwise: use the stage calls). */
int smhip_fused_box_dsi(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                        size_t img_pitch, const smhip_match_params *p, int window,
                        int d_first, int d_count, float *d_cost, size_t row_elems);
/* AggrMin (window x window, odd, 3..11) + PadCosts + OptWTA on a resident DSI of d_count levels
 * (global levels d_first ...) without writing the filtered DSI: one read of d_cost, keys out
 * (same format as smhip_fused_box_wta_keys).  The exp2 pipeline (Scripts/exp2_costSSDMF9.txt) is
 * smhip_fused_box_dsi + this call: 16 bytes per voxel.  Needs d_count % 4 == 0, row_elems % 4 == 0,
 * 16-byte aligned d_cost. */
int smhip_min_pad_wta_keys(smhip_ctx *ctx, const float *d_cost, size_t row_elems,
                           const smhip_match_params *p, int min_window, int d_first, int d_count,
                           int64_t *d_keys);
/* PadCosts + OptWTA (StcRawCosts.cpp:437-476, StcOptimize.cpp:26-56) on a resident slab of d_count
 * levels (global levels d_first ...), whatever aggregated it: pads d_cost in place like smhip_pad_costs,
 * keys out (same format as smhip_fused_box_wta_keys).  With smhip_raw_costs on the slab (disp_min +
 * d_first * disp_num, disp_n = d_count) and any per-level aggregator (box, binomial, diffusion, min
 * filter: all independent per level) this shards every WTA configuration by disparity slab; the one
 * exchange is the MIN all-reduce of the keys.  disp_den == 1. */
int smhip_pad_wta_keys(smhip_ctx *ctx, float *d_cost, size_t row_elems, const smhip_match_params *p,
                       int d_first, int d_count, int64_t *d_keys);
/* column-major keys -> row-major disparity and best cost images (either may be NULL) */
int smhip_keys_decode(smhip_ctx *ctx, const int64_t *d_keys, int width, int height,
                      int32_t *d_disp, float *d_best);
/* convenience: all D levels on this device, then decode */
int smhip_fused_box_wta(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                        size_t img_pitch, const smhip_match_params *p, int window,
                        int32_t *d_disp, float *d_best);

/* Fused path of the ITERATED aggregators: raw costs (StcRawCosts.cpp:229-435) -> n_iter iterations of aggr_fn
 * (2 = binomial 14641, StcAggregate.cpp:280-289; 3 = diffusion, 4 = membrane, :298-318, StcDiffusion.cpp:28-137)
 * -> PadCosts (:365) -> OptWTA (StcOptimize.cpp:26-56) on levels [d_first, d_first + d_count) -> one 64-bit key
 * per pixel, as smhip_fused_box_wta_keys.  Binomial with n_iter <= 4 (unit disparity step denominators, width % 4
 * == 0, one band with a point reference interval or four bands, |disparity step| <= 16) runs as ONE kernel that
 * moves no DSI at all: raw costs are formed in the tile, the iterations run in registers, the winners collect in
 * LDS.  More binomial iterations bracket the multi-iteration passes of smhip_convolve_14641_run between a
 * "raw costs + first iterations -> DSI" and a "last iterations + PadCosts + WTA -> keys" kernel.  Every other case
 * (diffusion / membrane, other shapes) runs the stage calls on a private DSI and gives the same keys.
 * lambda, beta: diff_lambda / diff_beta (membrane only; diffusion forces beta = 0, StcAggregate.cpp:305).
 * n_iter = 0 is allowed (raw costs -> PadCosts -> WTA). */
int smhip_fused_iter_wta_keys(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                              const smhip_match_params *p, int aggr_fn, int n_iter, float lambda, float beta,
                              int d_first, int d_count, int64_t *d_keys);

/* The same call for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): no counterpart in the reference.  The slab of
 * horizontal sums, the Birchfield-Tomasi images and the key image live in a static device array of
 * SMHIP_NOALLOC_SCRATCH_BYTES that every device the library runs on holds until the process ends; the
 * call itself makes no hipMalloc / hipFree.  Problems whose scratch does not fit are refused with
 * SMHIP_ENOMEM (the slab alone takes width * roundup(height, 32) * Dp * 4 bytes, Dp = levels rounded up
 * to 64 / 128 / a multiple of 256).  All calls on one device share that array: calls that overlap in
 * time on one device (two contexts, two streams) race; results are those of smhip_fused_box_wta. */
#define SMHIP_NOALLOC_SCRATCH_BYTES (64u << 20)
int smhip_fused_box_wta_noalloc(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                                const smhip_match_params *p, int window, int32_t *d_disp, float *d_best);

/* ---- kernel timing (replaces the Timer of CudaUtilities.h:47-68 / profilingTimer) ----
 * Between begin and end every smhip_fused_box_wta_keys() call brackets its two kernels with
 * HIP events on the context's stream (no synchronisation is added).  smhip_profile_end()
 * synchronises and returns the number of recorded calls and the summed milliseconds of pass A
 * (raw cost + horizontal running sum) and pass B (vertical running sum + pad + WTA). */
int smhip_profile_begin(smhip_ctx *ctx, int max_calls);
int smhip_profile_end(smhip_ctx *ctx, int *n_calls, float *ms_pass_a, float *ms_pass_b);
/* per-call times of the profiled calls [first, first + count) (waits for the stream); before _end */
int smhip_profile_get(smhip_ctx *ctx, int first, int count, float *ms_pass_a, float *ms_pass_b);
/* Name of the pass A kernel the context's most recent fused call launched ("" before the first call): the
 * dispatcher picks among four kernels by shape and options, and a report should name the one that ran
 * (the printf lines of CudaRawCosts.cu:381-420 name their one kernel). */
const char *smhip_last_pass_a_kernel(const smhip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
