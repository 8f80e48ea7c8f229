// StereoMatcher.h -- CStereoMatcher: the pipeline driver of the hot path on the device.
// Same entry points as the reference class (StereoMatcher.h:68-173): ComputeCorrespondence()
// runs PreProcess -> RawCosts -> Aggregate -> Optimize -> Refine; the stage bodies call the
// C-ABI of include/smhip.h instead of the CPU operators, and the disparity-space image stays
// on the device between stages.
#pragma once
#include "../../../include/smhip.h"
#include "Image.h"
#include "StereoParameters.h"
#include <vector>

struct CStereoFrame {
    CByteImage input_image, depth_image, truth_image, resampled_image;
    float predict_err = -1.0f, predict_visible = 0.0f;    // re-projection error of this frame (StereoMatcher.h:52-53)
};

class CStereoMatcher : public CStereoParameters {
public:
    CStereoMatcher();
    ~CStereoMatcher();
    void ComputeCorrespondence();
    void Evaluate();                    // StcEvaluate.cpp: masks + RMS / bad-pixel statistics vs the ground truth
    bool UsedFusedPath() const { return m_used_fused; }
    bool allow_fused = true;            // RawCosts+AggrBox+PadCosts+OptWTA in one device call when legal

protected:
    void PreProcess();
    void RawCosts();
    void Aggregate();
    void Optimize();
    void Refine();
    void AggrBox();
    void AggrDiffusion(int iter);
    void AggrMin();
    void AggrSubPixelFit();
    void AggrCollapse();
    void PadCosts();
    void OptWTA();
    void OptSO();
    void OptDP();
    void ComputeEnergy(float &denergy, float &nenergy);
    void ComputeOcclusions();
    void ComputeTextureless();
    void ComputeDisparityDiscont();
    void ComputeDisparityErrors();
    void ComputePredictionErrors();
    void ComputeMatchQuality();
    void ComputePredictionError(CByteImage &predicted, const CByteImage &original, float &rms_error,
                                float &fraction_visible);
    void BoxFilterOnDevice(CFloatImage &img, int window, bool average);

    void Check(int rc);                 // smhip status -> CError
    void *DeviceBuffer(int slot, size_t bytes);
    void SwapCost();
    void WriteCosts(const char *filename);
    bool Fusable() const;

    smhip_ctx *m_ctx = nullptr;
    smhip_match_params m_mp;
    float m_disp_step, m_disp_step_inv, m_match_outside;
    int m_disp_n, m_disp_num, m_disp_den, m_frame_diff, m_frame_diff_sign;
    bool m_used_fused = false;
    bool m_raw_deferred = false;        // RawCosts + AggrBox run as one smhip_fused_box_dsi call
    bool m_keys_done = false;           // AggrMin + PadCosts + OptWTA already ran as smhip_min_pad_wta_keys
    bool IterFusable() const;
    bool BoxDsiFusable() const;
    bool MinWtaFusable() const;
    bool m_have_mind = false;           // m_sub_pixel_min holds offsets (StcAggregate.cpp:77,228)
    CByteImage m_reference, m_matching;
    CIntImage m_disparity;              // winning levels [0 .. m_disp_n-1]
    CFloatImage m_best_cost;            // m_cost at the winner (data term of final_energy)
    CFloatImage m_float_disparity;
    CFloatImage m_true_disparity;       // ground truth / disp_scale + disp_min (StereoMatcher.cpp:121-130)
    CByteImage m_occlusion, m_textureless, m_depth_discont;
    CFloatImage m_final_cost, m_certainty;   // ComputeMatchQuality: cost / certainty at the winner
    size_t m_cert_row_elems = 0;        // row stride of m_sub_pixel_cert (it is not collapsed with the DSI)
    size_t m_row_elems = 0;
    enum { B_REF, B_MTC, B_COST, B_COST2, B_COST0, B_DISP, B_BEST, B_MIND, B_MIND2, B_CERT, B_FDISP, B_KEYS, B_FCOST, B_FCERT, B_COUNT };
    void *m_dev[B_COUNT];
    size_t m_dev_bytes[B_COUNT];
    std::vector<CStereoFrame> m_frame;
};
