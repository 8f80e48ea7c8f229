// smhip_fused.h -- pass A of the tuned fused path (raw cost + horizontal running sum) and the
// helpers it shares with pass B (smhip_fused_b.h): the slab layout and the DPP wave minimum.
//
// Pass A  (fused_rawbox_h_lds_kernel): one workgroup = one image row x up to 256 levels.
//   The row of the reference image and the (min,max) Birchfield-Tomasi row of the matching
//   image are staged once in LDS (2 x 2 bytes per pixel); each lane then walks x serially,
//   evaluating ONE new integer raw cost per step and taking the cost that leaves the window
//   from a register ring (window size is a template parameter, so ring slots are static).
//   Each step a wave stores 64 consecutive floats (256 B) of the slab T.
//   HBM traffic: W*H*Dp*4 bytes written, images read once.
#pragma once

#include "smhip_kernels.h"

namespace smk {

// Slab layout shared by pass A and pass B: rows are grouped in bands of TY = 1 << tyshift
// rows; element (y, x, k) lives at
//     (((y >> tyshift) * W + x) << tyshift) * Dp + (y & (TY-1)) * Dp + k.
// tyshift = 0 is the plain [y][x][k] layout.  Bands make the H concurrent row streams of
// pass A land in H/TY compact windows (measured: 8..10 % faster slab writes, reads unchanged).
__device__ __forceinline__ size_t slab_row_offset(int y, int W, int Dp, int tyshift)
{
    return (((size_t)(y >> tyshift) * W) << tyshift) * Dp + (size_t)(y & ((1 << tyshift) - 1)) * Dp;
}

struct FusedA {
    const uint8_t *rn, *rx;     // reference row values / upper interval end (== rn unless match_interpolated)
    const uint8_t *mn, *mx;     // matching row lower / upper interval end (== values when no BT)
    size_t pitch;
    int W, H, D;                // D = levels in the slab
    int disp0, dstep;           // disp(k) = disp0 + k*dstep
    int cutoff, outside;
};

// One raw cost (n_bands = 1).  Rl: packed (rn | rx << 8) per pixel, Ml: packed (mn | mx << 8).
template <bool SD>
__device__ __forceinline__ int cost1(const uint16_t *Rl, const uint16_t *Ml, int x, int disp, int W,
                                     int cutoff, int outside)
{
    int xm = x + disp;
    bool out = (unsigned)xm >= (unsigned)W;
    int xc = (int)min((unsigned)xm, (unsigned)(W - 1));
    int r = Rl[x], m = Ml[xc];
    int rn = r & 0xFF, rx = r >> 8, mn = m & 0xFF, mx = m >> 8;
    // gap between [rn,rx] and [mn,mx] (StcRawCosts.cpp:168-180); with point intervals this is |m - r|
    int e = max(max(mn - rx, rn - mx), 0);
    if (SD)
        e = e * e;
    e = min(e, cutoff);
    return out ? outside : e;
}

template <int WIN, bool SD>
__global__ __launch_bounds__(256) void fused_rawbox_h_lds_kernel(FusedA a, float *__restrict__ T,
                                                                 int tyshift, int Dp, int win,
                                                                 float scale)
{
    extern __shared__ uint16_t lds_a[];
    const int W = a.W;
    uint16_t *Rl = lds_a, *Ml = lds_a + W;
    const int y = blockIdx.x;
    const int k = blockIdx.y * blockDim.x + threadIdx.x;
    {
        const size_t ro = (size_t)y * a.pitch;
        for (int i = threadIdx.x; i < W; i += blockDim.x) {
            Rl[i] = (uint16_t)(a.rn[ro + i] | (a.rx[ro + i] << 8));
            Ml[i] = (uint16_t)(a.mn[ro + i] | (a.mx[ro + i] << 8));
        }
    }
    __syncthreads();
    if (k >= a.D)
        return;
    const int disp = a.disp0 + k * a.dstep;
    const int cutoff = a.cutoff, outside = a.outside;
    float *t = T + slab_row_offset(y, W, Dp, tyshift) + k;
    const size_t xs = (size_t)Dp << tyshift;           // floats between consecutive x
    const int w = WIN ? WIN : win;
    const int pr = w / 2, pl = pr - w;

    float acc = 0;
    if (WIN) {
        int ring[WIN ? WIN : 1];
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            int c = cost1<SD>(Rl, Ml, clampi(pl + 1 + j, W), disp, W, cutoff, outside);
            ring[j] = c;
            acc += scale * (float)c;
        }
        t[0] = acc;
        // Positions q with 0 <= q + disp < W for every lane of the wave: there neither the
        // raw cost nor the store needs a bounds test (disp is affine in the lane index).
        const int kf = __builtin_amdgcn_readfirstlane(k);
        const int kl = min(kf + 63, a.D - 1);
        const int df = a.disp0 + kf * a.dstep, dl = a.disp0 + kl * a.dstep;
        const int qa = max(0, -min(df, dl)), qb = min(W, W - max(df, dl));
        int xb = 1;
        for (; xb + WIN <= W; xb += WIN) {
            if (xb >= qa && xb + WIN - 1 + pr < qb) {
#pragma unroll
                for (int s = 0; s < WIN; s++) {
                    int x = xb + s;
                    int r = Rl[x + pr], m = Ml[x + pr + disp];
                    int e = max(max((m & 0xFF) - (r >> 8), (r & 0xFF) - (m >> 8)), 0);
                    if (SD)
                        e = e * e;
                    int c = min(e, cutoff);
                    acc = acc + scale * (float)(c - ring[s]);
                    ring[s] = c;
                    t[xs * x] = acc;
                }
            } else {
#pragma unroll
                for (int s = 0; s < WIN; s++) {
                    int x = xb + s;
                    int c = cost1<SD>(Rl, Ml, min(x + pr, W - 1), disp, W, cutoff, outside);
                    acc = acc + scale * (float)(c - ring[s]);
                    ring[s] = c;
                    if ((unsigned)(x + disp) < (unsigned)W)   // PadCosts overwrites the rest: never read
                        t[xs * x] = acc;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            int x = xb + s;
            if (x < W) {
                int c = cost1<SD>(Rl, Ml, min(x + pr, W - 1), disp, W, cutoff, outside);
                acc = acc + scale * (float)(c - ring[s]);
                ring[s] = c;
                if ((unsigned)(x + disp) < (unsigned)W)
                    t[xs * x] = acc;
            }
        }
    } else {
        for (int i = pl + 1; i <= pr; i++)
            acc += scale * (float)cost1<SD>(Rl, Ml, clampi(i, W), disp, W, cutoff, outside);
        t[0] = acc;
        for (int x = 1; x < W; x++) {
            int cl = cost1<SD>(Rl, Ml, max(x + pl, 0), disp, W, cutoff, outside);
            int cr = cost1<SD>(Rl, Ml, min(x + pr, W - 1), disp, W, cutoff, outside);
            acc = acc + scale * (float)(cr - cl);
            if ((unsigned)(x + disp) < (unsigned)W)
                t[xs * x] = acc;
        }
    }
}

// ---- DPP wave reduction -------------------------------------------------------------------

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min(int v)
{
    int t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, CTRL, ROW_MASK, 0xF, false);
    return min(v, t);
}

// min over the 64 lanes, returned as a wave-uniform value
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = dpp_min<0x111, 0xF>(v);   // row_shr:1
    v = dpp_min<0x112, 0xF>(v);   // row_shr:2
    v = dpp_min<0x114, 0xF>(v);   // row_shr:4
    v = dpp_min<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of each row holds the row minimum
    v = dpp_min<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_min<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the minimum
    return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace smk
