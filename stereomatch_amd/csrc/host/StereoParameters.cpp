// StereoParameters.cpp -- defaults (reference StereoParameters.cpp:33-144) and the registration
// order of the result file (reference :146-241).
#include "StereoParameters.h"

CStereoParameters::CStereoParameters()
{
    ReInitializeSeqParams();
    ReInitializeAlgParams();
    ResetOutputParams();
    fraction_matched = 0.0f;
}

void CStereoParameters::ReInitializeSeqParams()
{
    frame_ref = 0; frame_match = 1;
    disp_min = 0; disp_max = 7; disp_step = 1.0f; disp_n = 0; disp_scale = 1.0f;
    eval_ignore_border = 0; eval_disp_gap = 2.0f;
    verbose = eVerboseProgress; evaluate_only = 0;
}

void CStereoParameters::ReInitializeAlgParams()
{
    preproc_addnoise_stddev = 0.0f; preproc_blur_iter = 0;
    match_fn = eAD; match_interp = eCubic; match_max = 1000; match_interval = 0; match_interpolated = 0;
    aggr_fn = eBox; aggr_window_size = 7; aggr_iter = 1; aggr_minfilter = 0; aggr_subpixel = 0;
    aggr_collapse = 0;
    diff_lambda = 0.15f; diff_beta = 0.5f; diff_scale_cost = 0.01f; diff_mu = 0.5f;
    diff_sigmaP = 0.4f; diff_epsP = 0.01f;
    opt_fn = eWTA; opt_smoothness = 1.0f; opt_grad_thresh = 5.0f; opt_grad_penalty = 1.0f;
    opt_occlusion_cost = 20; opt_max_iter = 100; opt_random = 1; opt_sa_var = 3;
    opt_sa_start_T = 10.0f; opt_sa_end_T = 0.01f; opt_sa_schedule = 1; opt_min_margin = 0.7f;
    opt_sym_passes = 1;
    refine_subpix = 0;
    eval_bad_thresh = 1.0f; eval_error_scale = 2.0f; eval_lin_interp = 1; eval_predict_type = 1;
    eval_textureless_width = 3; eval_textureless_thresh = 4.0f; eval_discont_width = 9;
    eval_predict_diff = 0; eval_empty_color = 0x00ffc0ff; eval_partial_shuffle = 0.f;
    eval_match_quality = 0; eval_certain_matches_only = 0;
}

void CStereoParameters::ResetOutputParams()
{
    rms_error_all = rms_error_nonocc = rms_error_occ = rms_error_textured = -1.0f;
    rms_error_textureless = rms_error_discont = -1.0f;
    bad_pixels_all = bad_pixels_nonocc = bad_pixels_occ = bad_pixels_textured = -1.0f;
    bad_pixels_textureless = bad_pixels_discont = -1.0f;
    predict_err_near = predict_err_middle = predict_err_match = predict_err_far = -1.0f;
    final_energy = -1.0f; total_time = -1.0f;
}

void CStereoParameters::PIOInitialize(CParameterIO &o)
{
    o.PPPF(preproc_addnoise_stddev); o.PPPD(preproc_blur_iter);
    o.PPPD(frame_ref); o.PPPD(frame_match); o.PPPD(disp_min); o.PPPD(disp_max); o.PPPF(disp_step);
    o.PPPD(disp_n); o.PPPF(disp_scale); o.PPPD(match_fn); o.PPPD(match_interp); o.PPPD(match_max);
    o.PPPD(match_interval); o.PPPD(match_interpolated);
    o.PPPD(aggr_fn); o.PPPD(aggr_window_size); o.PPPD(aggr_iter); o.PPPD(aggr_minfilter);
    o.PPPD(aggr_subpixel); o.PPPD(aggr_collapse); o.PPPF(diff_lambda); o.PPPF(diff_beta);
    o.PPPF(diff_scale_cost); o.PPPF(diff_mu); o.PPPF(diff_sigmaP); o.PPPF(diff_epsP);
    o.PPPD(opt_fn); o.PPPF(opt_smoothness); o.PPPF(opt_grad_thresh); o.PPPF(opt_grad_penalty);
    o.PPPD(opt_occlusion_cost); o.PPPD(opt_max_iter); o.PPPD(opt_random); o.PPPD(opt_sa_var);
    o.PPPF(opt_sa_start_T); o.PPPF(opt_sa_end_T); o.PPPD(opt_sa_schedule); o.PPPF(opt_min_margin);
    o.PPPD(opt_sym_passes);
    o.PPPD(refine_subpix);
    o.PPPD(eval_ignore_border); o.PPPF(eval_bad_thresh); o.PPPF(eval_error_scale);
    o.PPPD(eval_lin_interp); o.PPPF(eval_disp_gap); o.PPPD(eval_predict_type);
    o.PPPD(eval_textureless_width); o.PPPF(eval_textureless_thresh); o.PPPD(eval_discont_width);
    o.PPPD(eval_predict_diff); o.PPPH(eval_empty_color); o.PPPF(eval_partial_shuffle);
    o.PPPD(eval_match_quality); o.PPPD(eval_certain_matches_only);
    o.PPPF(rms_error_all); o.PPPF(rms_error_nonocc); o.PPPF(rms_error_occ); o.PPPF(rms_error_textured);
    o.PPPF(rms_error_textureless); o.PPPF(rms_error_discont); o.PPPF(bad_pixels_all);
    o.PPPF(bad_pixels_nonocc); o.PPPF(bad_pixels_occ); o.PPPF(bad_pixels_textured);
    o.PPPF(bad_pixels_textureless); o.PPPF(bad_pixels_discont); o.PPPF(fraction_matched);
    o.PPPF(predict_err_near); o.PPPF(predict_err_middle); o.PPPF(predict_err_match);
    o.PPPF(predict_err_far);
    o.PPPF(final_energy); o.PPPF(total_time);
    o.PPPD(verbose); o.PPPD(evaluate_only);
}
