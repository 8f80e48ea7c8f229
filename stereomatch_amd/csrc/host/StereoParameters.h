// StereoParameters.h -- the parameter bag of the matcher.  Field names, enum values and
// defaults are the reference's (StereoParameters.h:33-270, StereoParameters.cpp:33-144), so
// that the reference's scripts and parameter files drive this driver unchanged.  Parameters
// of stages outside the hot path (global optimisers, evaluator) are kept as registered names
// only: scripts may set them, the result file lists them, nothing here acts on them.
#pragma once
#include "ParameterIO.h"

enum EVerbosityLevel {                 // Verbose.h:18-31
    eVerboseQuiet = 0, eVerboseWarning = 1, eVerboseSummary = 2, eVerboseProgress = 3,
    eVerboseFileIO = 4, eVerboseTiming = 5, eVerbosePredictionError = 6, eVerboseScriptFile = 10,
    eVerboseInnerLoops = 15, eVerboseDumpFiles = 25, eVerboseAllMessages = 99
};
enum EStereoMatchFn { eAD = 1, eSD = 2 };
enum EStereoInterpFn { eLinear = 1, eCubic = 3 };
enum EStereoAggrFn { eBox = 1, eBinomial = 2, eDiffusion = 3, eMembrane = 4, eBayesian = 5 };
enum EStereoOptimizeFn { eNoOpt = 0, eWTA = 1, eDynamicProg = 2, eScanlineOpt = 3, eGraphCut = 4,
                         eSimulAnnl = 5, eSymmetric = 6 };

struct CStereoParameters {
    // pre-processing
    float preproc_addnoise_stddev; int preproc_blur_iter;
    // matching
    int frame_ref, frame_match, disp_min, disp_max; float disp_step; int disp_n; float disp_scale;
    int match_fn, match_interp, match_max, match_interval, match_interpolated;
    // aggregation
    int aggr_fn, aggr_window_size, aggr_iter, aggr_minfilter, aggr_subpixel, aggr_collapse;
    float diff_lambda, diff_beta, diff_scale_cost, diff_mu, diff_sigmaP, diff_epsP;
    // optimisation
    int opt_fn; float opt_smoothness, opt_grad_thresh, opt_grad_penalty;
    int opt_occlusion_cost, opt_max_iter, opt_random, opt_sa_var;
    float opt_sa_start_T, opt_sa_end_T; int opt_sa_schedule; float opt_min_margin; int opt_sym_passes;
    // refinement
    int refine_subpix;
    // evaluation (registered names only)
    int eval_ignore_border; float eval_bad_thresh, eval_error_scale; int eval_lin_interp;
    float eval_disp_gap; int eval_predict_type, eval_textureless_width; float eval_textureless_thresh;
    int eval_discont_width, eval_predict_diff, eval_empty_color; float eval_partial_shuffle;
    int eval_match_quality, eval_certain_matches_only;
    // outputs
    float rms_error_all, rms_error_nonocc, rms_error_occ, rms_error_textured, rms_error_textureless,
        rms_error_discont, bad_pixels_all, bad_pixels_nonocc, bad_pixels_occ, bad_pixels_textured,
        bad_pixels_textureless, bad_pixels_discont, fraction_matched, predict_err_near,
        predict_err_middle, predict_err_match, predict_err_far, final_energy, total_time;
    // misc
    int verbose, evaluate_only;

    CStereoParameters();
    void ReInitializeSeqParams();     // not touched by "reset"
    void ReInitializeAlgParams();     // "reset"
    void ResetOutputParams();
    void PIOInitialize(CParameterIO &o);
};
