// smhip_fused.h -- pass A of the tuned fused path (raw cost + horizontal running sum) and the
// helpers it shares with pass B (smhip_fused_b.h): the slab layout and the DPP wave minimum.
//
// Pass A  (fused_rawbox_h_lds_kernel): one workgroup = one image row x up to 256 levels.
//   The row of the reference image and the (min,max) Birchfield-Tomasi row of the matching
//   image are staged once in LDS (4 bytes per pixel and band); each lane then walks x
//   serially, evaluating ONE new integer raw cost per step and taking the cost that leaves
//   the window from a register ring (window size is a template parameter, so ring slots are
//   static).  Each step a wave stores 64 consecutive floats (256 B) of the slab T.
//   HBM traffic: W*H*Dp*4 bytes written, images read once.
//   keep_all = 0: elements that PadCosts will overwrite right after the box filter are not
//   stored (pass B skips them too); keep_all = 1 (a min filter follows): everything is stored.
#pragma once

#include "smhip_kernels.h"

namespace smk {

// Staged image rows.  NB = 1: one uint16 per pixel, (lower | upper << 8).  NB = 4: one uint32
// of the four lower ends and one of the four upper ends per pixel.
template <int NB> struct RowLds;
template <> struct RowLds<1> {
    uint16_t *r, *m;
    __device__ static size_t bytes(int W) { return (size_t)W * 4; }
    __device__ void init(void *base, int W) { r = (uint16_t *)base; m = r + W; }
    __device__ void stage(const FusedA &a, size_t ro, int i) const
    {
        r[i] = (uint16_t)(a.rn[ro + i] | (a.rx[ro + i] << 8));
        if (a.bt_inline) {      // BirchfieldTomasiMinMax (StcRawCosts.cpp:107-128), as bt_minmax_kernel
            const uint8_t *row = a.mn + ro;
            const int cur = row[i], prv = row[i > 0 ? i - 1 : i], nxt = row[i < a.W - 1 ? i + 1 : i];
            const int hl = (i > 0) ? (prv + cur + 1) / 2 : cur;
            const int hr = (cur + nxt + 1) / 2;
            m[i] = (uint16_t)(min(hl, min(cur, hr)) | (max(hl, max(cur, hr)) << 8));
        } else {
            m[i] = (uint16_t)(a.mn[ro + i] | (a.mx[ro + i] << 8));
        }
    }
    // chunked staging (smhip_fused_chunk.h): nref reference and nmatch matching pixels,
    // slot = pixel - first pixel of the chunk
    __device__ void init2(void *base, int nref, int) { r = (uint16_t *)base; m = r + nref; }
    __device__ void stage_r(const FusedA &a, size_t ro, int src, int slot) const
    {
        r[slot] = (uint16_t)(a.rn[ro + src] | (a.rx[ro + src] << 8));
    }
    __device__ void stage_m(const FusedA &a, size_t ro, int src, int slot) const
    {
        m[slot] = (uint16_t)(a.mn[ro + src] | (a.mx[ro + src] << 8));
    }
    // distance between the intervals of reference pixel xr and matching pixel xm
    // (StcRawCosts.cpp:168-180; with point intervals it is |m - r|), summed over bands
    template <bool SD> __device__ __forceinline__ int gap(int xr, int xm) const
    {
        int a = r[xr], b = m[xm];
        int e = max(max((b & 0xFF) - (a >> 8), (a & 0xFF) - (b >> 8)), 0);
        return SD ? e * e : e;
    }
};
template <> struct RowLds<4> {
    uint32_t *rn, *rx, *mn, *mx;
    __device__ static size_t bytes(int W) { return (size_t)W * 16; }
    __device__ void init(void *base, int W) { rn = (uint32_t *)base; rx = rn + W; mn = rx + W; mx = mn + W; }
    __device__ void stage(const FusedA &a, size_t ro, int i) const
    {
        rn[i] = *(const uint32_t *)(a.rn + ro + 4 * (size_t)i);
        rx[i] = *(const uint32_t *)(a.rx + ro + 4 * (size_t)i);
        mn[i] = *(const uint32_t *)(a.mn + ro + 4 * (size_t)i);
        mx[i] = *(const uint32_t *)(a.mx + ro + 4 * (size_t)i);
    }
    __device__ void init2(void *base, int nref, int nmatch)
    {
        rn = (uint32_t *)base; rx = rn + nref; mn = rx + nref; mx = mn + nmatch;
    }
    __device__ void stage_r(const FusedA &a, size_t ro, int src, int slot) const
    {
        rn[slot] = *(const uint32_t *)(a.rn + ro + 4 * (size_t)src);
        rx[slot] = *(const uint32_t *)(a.rx + ro + 4 * (size_t)src);
    }
    __device__ void stage_m(const FusedA &a, size_t ro, int src, int slot) const
    {
        mn[slot] = *(const uint32_t *)(a.mn + ro + 4 * (size_t)src);
        mx[slot] = *(const uint32_t *)(a.mx + ro + 4 * (size_t)src);
    }
    template <bool SD> __device__ __forceinline__ int gap(int xr, int xm) const
    {
        uint32_t an = rn[xr], ax = rx[xr], bn = mn[xm], bx = mx[xm];
        int sum = 0;
#pragma unroll
        for (int s = 0; s < 32; s += 8) {
            int e = max(max((int)((bn >> s) & 0xFF) - (int)((ax >> s) & 0xFF),
                            (int)((an >> s) & 0xFF) - (int)((bx >> s) & 0xFF)), 0);
            sum += SD ? e * e : e;
        }
        return sum;
    }
};

// NB = 2: ONE band with the reference interval a point (every case but match_interpolated).  A pixel is a 32-bit
// word: reference r | r << 8 | 255 << 16, matching mn | mx << 8 | (mx - mn) << 16, so that one v_sad_u8 gives
// |r - mn| + |r - mx| + 255 - (mx - mn) = 2 dist(r, [mn, mx]) + 255: the raw cost is (sad(r, m, -255)) >> 1 -- two
// instructions instead of unpacking two 16-bit words and a max3 (windows 13 / 17 at 1920x1080x256: pass A 0.428 /
// 0.411 -> 0.387 ms; raw_costs_rows at 3840x2160x512 stays at 3.25 ms, its stores bound it).
template <> struct RowLds<2> {
    uint32_t *r, *m;
    __device__ static size_t bytes(int W) { return (size_t)W * 8; }
    __device__ static uint32_t ref_word(int v) { return (uint32_t)v * 0x101u | 0xFF0000u; }
    __device__ static uint32_t match_word(int lo, int hi) { return (uint32_t)lo | (uint32_t)hi << 8 | (uint32_t)(hi - lo) << 16; }
    __device__ void init(void *base, int W) { r = (uint32_t *)base; m = r + W; }
    __device__ void stage(const FusedA &a, size_t ro, int i) const
    {
        r[i] = ref_word(a.rn[ro + i]);
        if (a.bt_inline) {      // BirchfieldTomasiMinMax (StcRawCosts.cpp:107-128), as RowLds<1>
            const uint8_t *row = a.mn + ro;
            const int cur = row[i], prv = row[i > 0 ? i - 1 : i], nxt = row[i < a.W - 1 ? i + 1 : i];
            const int hl = (i > 0) ? (prv + cur + 1) / 2 : cur;
            const int hr = (cur + nxt + 1) / 2;
            m[i] = match_word(min(hl, min(cur, hr)), max(hl, max(cur, hr)));
        } else {
            m[i] = match_word(a.mn[ro + i], a.mx[ro + i]);
        }
    }
    __device__ void init2(void *base, int nref, int) { r = (uint32_t *)base; m = r + nref; }
    __device__ void stage_r(const FusedA &a, size_t ro, int src, int slot) const { r[slot] = ref_word(a.rn[ro + src]); }
    __device__ void stage_m(const FusedA &a, size_t ro, int src, int slot) const
    {
        m[slot] = match_word(a.mn[ro + src], a.mx[ro + src]);
    }
    template <bool SD> __device__ __forceinline__ int gap(int xr, int xm) const
    {
        const int e = (int)(__builtin_amdgcn_sad_u8(r[xr], m[xm], (uint32_t)-255) >> 1);
        return SD ? e * e : e;
    }
};

// NB = 5: four bands with POINT intervals on both sides (no Birchfield-Tomasi sampling): half the staged words,
// and the SAD of a pixel pair is one v_sad_u8 instead of four interval distances (1920x1080x256, b = 4: pass A
// 0.48 -> 0.43 ms; a run-time switch inside RowLds<4> cost the interval form 8 %, hence a type of its own)
template <> struct RowLds<5> {
    uint32_t *rn, *mn;
    __device__ static size_t bytes(int W) { return (size_t)W * 8; }
    __device__ void init(void *base, int W) { rn = (uint32_t *)base; mn = rn + W; }
    __device__ void stage(const FusedA &a, size_t ro, int i) const
    {
        rn[i] = *(const uint32_t *)(a.rn + ro + 4 * (size_t)i);
        mn[i] = *(const uint32_t *)(a.mn + ro + 4 * (size_t)i);
    }
    __device__ void init2(void *base, int nref, int) { rn = (uint32_t *)base; mn = rn + nref; }
    __device__ void stage_r(const FusedA &a, size_t ro, int src, int slot) const
    {
        rn[slot] = *(const uint32_t *)(a.rn + ro + 4 * (size_t)src);
    }
    __device__ void stage_m(const FusedA &a, size_t ro, int src, int slot) const
    {
        mn[slot] = *(const uint32_t *)(a.mn + ro + 4 * (size_t)src);
    }
    template <bool SD> __device__ __forceinline__ int gap(int xr, int xm) const
    {
        const uint32_t r = rn[xr], m = mn[xm];
        if (!SD)
            return (int)__builtin_amdgcn_sad_u8(r, m, 0u);
        int sum = 0;
#pragma unroll
        for (int s = 0; s < 32; s += 8) {
            const int e = (int)((r >> s) & 0xFF) - (int)((m >> s) & 0xFF);
            sum += e * e;
        }
        return sum;
    }
};

// NB = 6: four bands, the reference interval a point, the matching interval [mn, mx] (Birchfield-Tomasi without
// match_interpolated), absolute differences.  Per byte the distance of r to [mn, mx] is
// (|r - mn| + |r - mx| - (mx - mn)) / 2 -- 2 (mn - r) below the interval, 0 inside, 2 (r - mx) above -- so the sum
// over the four bands is (sad(r, mn) + sad(r, mx) - sad(mx, mn)) / 2: two v_sad_u8, a subtraction and a shift, with
// sad(mx, mn) staged once per matching pixel (1920x1080x256, b = 4: pass A 0.48 -> 0.39 ms).  gap<true> is not
// provided: squared differences need the bands one by one (RowLds<4>).
template <> struct RowLds<6> {
    uint32_t *rn, *mn, *mx, *mw;
    __device__ static size_t bytes(int W) { return (size_t)W * 16; }
    __device__ void init(void *base, int W) { rn = (uint32_t *)base; mn = rn + W; mx = mn + W; mw = mx + W; }
    __device__ void stage(const FusedA &a, size_t ro, int i) const
    {
        rn[i] = *(const uint32_t *)(a.rn + ro + 4 * (size_t)i);
        const uint32_t lo = *(const uint32_t *)(a.mn + ro + 4 * (size_t)i);
        const uint32_t hi = *(const uint32_t *)(a.mx + ro + 4 * (size_t)i);
        mn[i] = lo; mx[i] = hi; mw[i] = __builtin_amdgcn_sad_u8(hi, lo, 0u);
    }
    __device__ void init2(void *base, int nref, int nmatch)
    {
        rn = (uint32_t *)base; mn = rn + nref; mx = mn + nmatch; mw = mx + nmatch;
    }
    __device__ void stage_r(const FusedA &a, size_t ro, int src, int slot) const
    {
        rn[slot] = *(const uint32_t *)(a.rn + ro + 4 * (size_t)src);
    }
    __device__ void stage_m(const FusedA &a, size_t ro, int src, int slot) const
    {
        const uint32_t lo = *(const uint32_t *)(a.mn + ro + 4 * (size_t)src);
        const uint32_t hi = *(const uint32_t *)(a.mx + ro + 4 * (size_t)src);
        mn[slot] = lo; mx[slot] = hi; mw[slot] = __builtin_amdgcn_sad_u8(hi, lo, 0u);
    }
    template <bool SD> __device__ __forceinline__ int gap(int xr, int xm) const
    {
        static_assert(!SD, "absolute differences only");
        const uint32_t r = rn[xr];
        const uint32_t t = __builtin_amdgcn_sad_u8(r, mx[xm], __builtin_amdgcn_sad_u8(r, mn[xm], 0u));
        return (int)((t - mw[xm]) >> 1);
    }
};

template <int WIN, bool SD, int NB>
__global__ __launch_bounds__(256) void fused_rawbox_h_lds_kernel(FusedA a, float *__restrict__ T,
                                                                 int tyshift, int Dp, int win,
                                                                 float scale, int keep_all)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_a[];
    const int W = a.W;
    RowLds<NB> L;
    L.init(lds_a, W);
    const int y = blockIdx.x;
    const int k = blockIdx.y * blockDim.x + threadIdx.x;
    {
        const size_t ro = (size_t)y * a.pitch;
        for (int i = threadIdx.x; i < W; i += blockDim.x)
            L.stage(a, ro, i);
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

    // MatchLine (StcRawCosts.cpp:152-198) for reference pixel x at this lane's level
    auto cost = [&](int x) -> int {
        int xm = x + disp;
        bool out = (unsigned)xm >= (unsigned)W;
        int xc = (int)min((unsigned)xm, (unsigned)(W - 1));
        int e = min(L.template gap<SD>(x, xc), cutoff);
        return out ? outside : e;
    };

    float acc = 0;
    if (WIN) {
        int ring[WIN ? WIN : 1];
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            int c = cost(clampi(pl + 1 + j, W));
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
                    int c = min(L.template gap<SD>(x + pr, x + pr + disp), cutoff);
                    acc = acc + scale * (float)(c - ring[s]);
                    ring[s] = c;
                    t[xs * x] = acc;
                }
            } else {
#pragma unroll
                for (int s = 0; s < WIN; s++) {
                    int x = xb + s;
                    int c = cost(min(x + pr, W - 1));
                    acc = acc + scale * (float)(c - ring[s]);
                    ring[s] = c;
                    if (keep_all || (unsigned)(x + disp) < (unsigned)W)   // PadCosts overwrites the rest: never read
                        t[xs * x] = acc;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            int x = xb + s;
            if (x < W) {
                int c = cost(min(x + pr, W - 1));
                acc = acc + scale * (float)(c - ring[s]);
                ring[s] = c;
                if (keep_all || (unsigned)(x + disp) < (unsigned)W)
                    t[xs * x] = acc;
            }
        }
    } else {
        for (int i = pl + 1; i <= pr; i++)
            acc += scale * (float)cost(clampi(i, W));
        t[0] = acc;
        for (int x = 1; x < W; x++) {
            int cl = cost(max(x + pl, 0));
            int cr = cost(min(x + pr, W - 1));
            acc = acc + scale * (float)(cr - cl);
            if (keep_all || (unsigned)(x + disp) < (unsigned)W)
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
