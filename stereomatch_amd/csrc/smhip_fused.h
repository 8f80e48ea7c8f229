// smhip_fused.h -- tuned kernels of the fused raw-cost -> box -> pad -> WTA path for slabs
// whose padded width Dp is a multiple of 64 (one disparity per lane, 64 levels per wave).
//
// Pass A  (fused_rawbox_h_lds_kernel): one workgroup = one image row x up to 256 levels.
//   The row of the reference image and the (min,max) Birchfield-Tomasi row of the matching
//   image are staged once in LDS (3 bytes per pixel); each lane then walks x serially,
//   evaluating ONE new integer raw cost per step and taking the cost that leaves the window
//   from a register ring (window size is a template parameter, so ring slots are static).
//   Each step a wave stores 64 consecutive floats (256 B) of T[y][x][.].
//   HBM traffic: W*H*Dp*4 bytes written, images read once.
//
// Pass B  (fused_boxv_wta_dpp_kernel): one workgroup = 1..4 pixels columns x up to 256 levels;
//   each lane walks y serially over T (loads of a block of rows are issued together, two
//   blocks in flight), applies PadCosts and the 64 lanes of a wave min-reduce the ordered
//   cost with DPP row shifts / broadcasts; the winner's lane index is the lowest set bit of
//   the equality ballot (lowest d wins, like the strict '<' scan of OptWTA).  Waves of one
//   pixel are combined through a double-buffered LDS ring once per block of rows.
//   HBM traffic: W*H*Dp*4 bytes read (the row leaving the window is taken from registers
//   or, for a generic window, re-read from L2), 8 bytes per pixel written.
#pragma once

#include "smhip_kernels.h"

namespace smk {

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
                                                                 size_t trow, int Dp, int win,
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
    float *t = T + (size_t)y * trow + k;   // trow = floats per image row of T (>= W*Dp)
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
        int xb = 1;
        for (; xb + WIN <= W; xb += WIN) {
#pragma unroll
            for (int s = 0; s < WIN; s++) {
                int x = xb + s;
                int c = cost1<SD>(Rl, Ml, min(x + pr, W - 1), disp, W, cutoff, outside);
                acc = acc + scale * (float)(c - ring[s]);
                ring[s] = c;
                t[(size_t)x * Dp] = acc;
            }
        }
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            int x = xb + s;
            if (x < W) {
                int c = cost1<SD>(Rl, Ml, min(x + pr, W - 1), disp, W, cutoff, outside);
                acc = acc + scale * (float)(c - ring[s]);
                ring[s] = c;
                t[(size_t)x * Dp] = acc;
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
            t[(size_t)x * Dp] = acc;
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

// One step of pass B: running-sum value -> PadCosts -> wave argmin -> key.
__device__ __forceinline__ long long wta_step(float acc, bool padded, float outside, bool live,
                                              int dbase)
{
    float v = padded ? outside : acc;
    int s = live ? ordered_bits(v) : 0x7FFFFFFF;
    int m = wave_min_i32(s);
    unsigned long long eq = __builtin_amdgcn_ballot_w64(s == m);
    return make_key(m, dbase + (int)__builtin_ctzll(eq));
}

// WIN > 0: the window is a template parameter, rows are processed in blocks of WIN so that
// the row leaving the window at step r of a block is the row that entered at step r of the
// previous block (kept in registers: every element of T is loaded exactly once).
// WIN == 0: generic window, blocks of 8 rows, the leaving row is re-read (L2 hit).
// WPP = waves per pixel (1, 2, 3 or 4); blockDim.x = 64*WPP*PPW.
template <int WIN, int WPP>
__global__ __launch_bounds__(256) void fused_boxv_wta_dpp_kernel(
    const float *__restrict__ T, size_t trow, int W, int H, int Dp, int D, int win, float scale,
    int disp0, int dstep, float outside, int d_first, long long *__restrict__ keys,
    size_t keys_chunk_stride)
{
    constexpr int BLK = WIN ? WIN : 8;
    constexpr int CH = 64 * WPP;
    __shared__ long long ring[2][BLK][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int ppw = blockDim.x / CH;                  // pixels per workgroup
    const int pix = tid / CH, kloc = tid - pix * CH;
    const int x = blockIdx.x * ppw + pix;
    const int chunk = blockIdx.y;
    const int k = chunk * CH + kloc;
    const bool live = (x < W) && (k < D);
    const int xs = x < W ? x : W - 1;
    const int ks = k < D ? k : D - 1;
    const unsigned j = (unsigned)xs * (unsigned)Dp + (unsigned)ks;   // lane offset inside a row of T
    const size_t stride = trow;
    const int w = WIN ? WIN : win;
    const int pr = w / 2, pl = pr - w;
    const int disp = disp0 + ks * dstep;
    const bool padded = (xs + disp < 0) || (xs + disp >= W);
    long long *kout = keys + (size_t)chunk * keys_chunk_stride;
    const int dbase = d_first + chunk * CH + (kloc & ~63);   // level of lane 0 of this wave

    // row index is wave-uniform: scalar base + 32-bit lane offset
    auto ld = [&](int row) -> float { return (T + (size_t)clampi(row, H) * stride)[j]; };

    float acc = 0;
    for (int i = pl + 1; i <= pr; i++)
        acc += scale * ld(i);

    auto emit = [&](long long key, int y, int r, int buf) {
        if (WPP == 1) {
            if (lane == 0 && y < H && x < W)
                kout[(size_t)y * W + x] = key;
        } else if (lane == 0) {
            ring[buf][r][wv] = key;
        }
    };
    auto combine = [&](int yb, int buf) {
        if (WPP > 1) {
            __syncthreads();
            if (tid < BLK * ppw) {
                int r = tid / ppw, p = tid - r * ppw;
                long long m = ring[buf][r][p * WPP];
#pragma unroll
                for (int q = 1; q < WPP; q++) {
                    long long o = ring[buf][r][p * WPP + q];
                    m = o < m ? o : m;
                }
                int xx = blockIdx.x * ppw + p;
                if (xx < W && yb + r < H)
                    kout[(size_t)(yb + r) * W + xx] = m;
            }
        }
    };

    if (WIN) {
        float b0[BLK], b1[BLK], b2[BLK];
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            b0[r] = ld(r + pl);
            b1[r] = ld(r + pr);
        }
        b0[0] = b1[0];   // step y = 0 adds scale*(-a + a) = +0: the first value is the plain window sum
        int buf = 0;
        auto block = [&](float(&prv)[BLK], float(&cur)[BLK], float(&nxt)[BLK], int yb) {
#pragma unroll
            for (int r = 0; r < BLK; r++)
                nxt[r] = ld(yb + BLK + r + pr);
#pragma unroll
            for (int r = 0; r < BLK; r++) {
                acc = acc + scale * (-prv[r] + cur[r]);
                emit(wta_step(acc, padded, outside, live, dbase), yb + r, r, buf);
            }
            combine(yb, buf);
            buf ^= 1;
        };
        for (int yb = 0; yb < H; yb += 3 * BLK) {
            block(b0, b1, b2, yb);
            if (yb + BLK < H)
                block(b1, b2, b0, yb + BLK);
            if (yb + 2 * BLK < H)
                block(b2, b0, b1, yb + 2 * BLK);
        }
    } else {
        float nw[2][BLK], od[2][BLK];
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            nw[0][r] = ld(r + pr);
            od[0][r] = ld(r + pl);
        }
        od[0][0] = nw[0][0];
        int buf = 0;
        for (int y0 = 0; y0 < H; y0 += 2 * BLK) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int yb = y0 + half * BLK;
                if (yb < H) {
#pragma unroll
                    for (int r = 0; r < BLK; r++) {
                        nw[half ^ 1][r] = ld(yb + BLK + r + pr);
                        od[half ^ 1][r] = ld(yb + BLK + r + pl);
                    }
#pragma unroll
                    for (int r = 0; r < BLK; r++) {
                        acc = acc + scale * (-od[half][r] + nw[half][r]);
                        emit(wta_step(acc, padded, outside, live, dbase), yb + r, r, buf);
                    }
                    combine(yb, buf);
                    buf ^= 1;
                }
            }
        }
    }
}

}  // namespace smk
