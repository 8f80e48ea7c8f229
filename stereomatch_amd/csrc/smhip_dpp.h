// smhip_dpp.h -- whole-wave one-lane shifts by DPP (wave_shl:1 / wave_shr:1), shared by the scan-line optimiser
// (smhip_dp.h) and the pixels-on-lanes stencils (smhip_multi_px.h).
#pragma once

#include <hip/hip_runtime.h>

namespace smk {

constexpr int DPP_WAVE_SHL1 = 0x130;          // lane n <- lane n+1
constexpr int DPP_WAVE_SHR1 = 0x138;          // lane n <- lane n-1

// lane n <- v of lane n+1; lane 63 <- edge
__device__ __forceinline__ float lane_from_above(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), DPP_WAVE_SHL1,
                                                      0xF, 0xF, false));
}
// lane n <- v of lane n-1; lane 0 <- edge
__device__ __forceinline__ float lane_from_below(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), DPP_WAVE_SHR1,
                                                      0xF, 0xF, false));
}
// The same moves for callers whose lane 63 / lane 0 is a halo lane that may receive anything (here: 0, bound_ctrl):
// no fallback register has to be loaded in front of the move.
__device__ __forceinline__ float lane_from_above_any(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WAVE_SHL1, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_from_below_any(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_WAVE_SHR1, 0xF, 0xF, true));
}
}  // namespace smk
