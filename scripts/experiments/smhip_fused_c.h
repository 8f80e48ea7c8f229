// smhip_fused_c.h -- the fused path without the slab: "recompute" form of raw costs -> box filter -> PadCosts -> WTA
// for one-band frames (StcRawCosts.cpp:229-435, BoxFilter.cpp:86-232, StcAggregate.cpp:86-143, StcOptimize.cpp:26-56).
//
// The two-pass form (smhip_fused_f.h + smhip_fused_b.h) writes the horizontally summed slab T (4 N bytes) and reads
// it back (4 N): 0.69 ms at config 3, both passes within a few per cent of what the memory system gives their access
// patterns.  Its arithmetic alone is far cheaper than those 8 N bytes (pass A without its stores: 0.13-0.16 ms), so
// this form trades the slab for arithmetic:
//   phase 1  fused_rawbox_h_f_kernel<..., STATES = true>: the horizontal walk of pass A, storing only the running sum
//            in front of every block of WIN columns -- 4 N / WIN bytes;
//   phase 2  (this file) a workgroup owns a block of WIN columns x 128 levels over the whole height.  Rows are taken
//            in batches of WIN.  H role: wave s restarts the horizontal chain of row s of the batch from its state and
//            walks the block's WIN columns (2 WIN raw costs per level for WIN sums: the halo of the restart), bit for
//            bit the additions pass A makes, and puts the WIN x 128 sums into an LDS tile.  V role: wave c owns column c
//            of the block; its lanes hold the vertical running sums of the column's 128 levels (two per lane) and the
//            previous batch's rows (the rows leaving the window), take the batch's rows from the tile, and do exactly
//            what pass B does per row: sum update, PadCosts, first minimum of the lane's levels, DPP minimum over the
//            lanes, lowest lane holding it, key.  Two barriers per batch.
// The recurrences are the reference's (serial in x per row and level, serial in y per column and level, every
// rounding in its order); only where they are evaluated differs.  Keys leave column-major in runs of GRP rows.
//
// MEASURED AND NOT ADOPTED (scripts/experiments/bench_c.hip, profiles/r03_recompute.log): bit-identical keys on every
// shape tried, but 1920x1080x256 takes 0.13 ms (states) + 0.87 ms (blocks) against 0.68 ms for the two passes.  The
// block kernel is bound by its instruction count and its two barriers per batch, not by memory: ~700 wave instructions
// per wave and batch of 9 x 9 x 128 sums (a third of them the per-pixel DPP minimum, a tenth the row staging), 167
// VGPRs, one nine-wave workgroup per CU and 428 workgroups on 256 CUs; without the lane reduction 0.68 ms, without the
// row staging 0.73, with neither 0.49.  Even at twice the occupancy and with the reduction batched through LDS the
// estimate stays at 0.45-0.5 ms for the pair -- not enough margin over 0.68 to carry the second code path.
#pragma once

#include "../../stereomatch_amd/csrc/smhip_fused_b.h"
#include "../../stereomatch_amd/csrc/smhip_fused_f.h"

namespace smk {

constexpr int FUSED_C_CW = 128;                        // levels per workgroup (two per lane)
constexpr size_t fused_c_lds(int win)
{
    return (size_t)win * win * FUSED_C_CW * 4 + (size_t)win * (2 * win + FUSED_C_CW) * 8 + (size_t)win * 2 * win * 4;
}

// grid: ceil(W / WIN) * (Dp / 128) workgroups of 64 * WIN threads (level chunk fastest); dynamic LDS fused_c_lds(WIN).
// S, nst: the states phase 1 wrote.  keys: one column-major key image per level chunk (keys_chunk_stride apart).
template <int WIN, bool SD, bool CUT, int DS, int ABL = 0>   // ABL: ablations of scripts/experiments/bench_c.hip (0 = product)
__global__ __launch_bounds__(64 * WIN) void fused_block_vwta_kernel(
    FusedA a, const float *__restrict__ S, int nst, int Dp, float scale, float outside, int d_first,
    long long *__restrict__ keys, size_t keys_chunk_stride, int merge)
{
    static_assert(WIN >= 3 && (WIN & 1) && WIN <= 15, "static odd window, at most 15 waves");
    constexpr int LPL = 2, CW = FUSED_C_CW;
    constexpr int NP = 2 * WIN;                        // reference positions x0 + pl .. x0 + WIN - 1 + pr of a block
    constexpr int NPX = NP + LPL - 1;                  // matching positions a lane reads
    constexpr int NM = NP + CW;                        // matching positions a row stages (at most)
    constexpr int NI = (NM + 63) / 64;
    constexpr int pr = WIN / 2, pl = pr - WIN;
    constexpr int GRP = (64 / WIN) * WIN;              // rows per key run
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_c[];
    float *tile = (float *)lds_c;                      // [row of the batch][column of the block][level]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x2 *ml = (f32x2 *)(tile + WIN * WIN * CW) + wv * NM;     // this wave's row: matching intervals from M0
    float *rl = (float *)((f32x2 *)(tile + WIN * WIN * CW) + WIN * NM) + wv * NP;   // ... reference values from R0
    const int W = a.W, H = a.H, D = a.D;
    const int nck = Dp / CW;
    const int jb = (int)blockIdx.x / nck;
    const int kc = ((int)blockIdx.x - jb * nck) * CW;
    const int x0 = jb * WIN;

    // ---- H role: row wv of the batch, levels kbase, kbase + 1 (as pass A: lanes beyond D shadow the last live one)
    const int Dl = (D + LPL - 1) / LPL * LPL;
    const int kbase = min(kc + lane * LPL, Dl - LPL);
    int disp[LPL];
#pragma unroll
    for (int i = 0; i < LPL; i++)
        disp[i] = a.disp0 + (kbase + i) * DS;
    const int dA = a.disp0 + kc * DS, dB = a.disp0 + (min(kc + CW, Dl) - 1) * DS;
    const int dlo = min(dA, dB), dhi = max(dA, dB);
    const int R0 = clampi(x0 + pl, W), R1 = clampi(x0 + WIN - 1 + pr, W);
    const int M0 = clampi(R0 + dlo, W), M1 = clampi(R1 + dhi, W);
    // every position of the block and every matching position of every level inside the row: no clamps, no tests
    const bool interior = x0 + pl >= 0 && x0 + WIN - 1 + pr <= W - 1 && x0 + pl + dlo >= 0 &&
                          x0 + WIN - 1 + pr + dhi <= W - 1;
    const float cutf = (float)a.cutoff, outf = (float)a.outside;
    constexpr bool RAW = !SD && !CUT;
    auto mag = [](float v) -> float { return RAW ? __builtin_fabsf(v) : v; };
    auto dist = [&](float r, f32x2 m) -> float {       // StcRawCosts.cpp:168-185, as smhip_fused_f.h
        float q = r - __builtin_amdgcn_fmed3f(m.x, m.y, r);
        if (RAW) return q;
        float e = SD ? q * q : __builtin_fabsf(q);
        if (CUT) e = __builtin_fminf(e, cutf);
        return e;
    };
    uint8_t f_cur[NI], f_prv[NI], f_nxt[NI], f_ref = 0;
    auto fetch = [&](int y) {
        const uint8_t *mrow = a.mn + (size_t)y * a.pitch, *xrow = a.mx + (size_t)y * a.pitch;
        const uint8_t *rrow = a.rn + (size_t)y * a.pitch;
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int i = lane + j * 64;
            f_cur[j] = f_prv[j] = f_nxt[j] = 0;
            if (i <= M1 - M0) {
                const int p = M0 + i;
                if (a.bt_inline) {
                    f_cur[j] = mrow[p];
                    f_prv[j] = mrow[p > 0 ? p - 1 : p];
                    f_nxt[j] = mrow[p < W - 1 ? p + 1 : p];
                } else {
                    f_cur[j] = mrow[p];
                    f_nxt[j] = xrow[p];
                }
            }
        }
        if (lane <= R1 - R0)
            f_ref = rrow[R0 + lane];
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int i = lane + j * 64;
            if (i <= M1 - M0) {
                const int p = M0 + i;
                int lo, hi;
                if (a.bt_inline) {                     // BirchfieldTomasiMinMax (StcRawCosts.cpp:107-128)
                    const int cur = f_cur[j], prv = f_prv[j], nxt = f_nxt[j];
                    const int hl = (p > 0) ? (prv + cur + 1) / 2 : cur;
                    const int hr = (cur + nxt + 1) / 2;
                    lo = min(hl, min(cur, hr));
                    hi = max(hl, max(cur, hr));
                } else {
                    lo = f_cur[j];
                    hi = f_nxt[j];
                }
                f32x2 v = {(float)lo, (float)hi};
                ml[i] = v;
            }
        }
        if (lane <= R1 - R0)
            rl[lane] = (float)f_ref;
    };
    auto cost_at = [&](int q, int i) -> float {        // MatchLine at reference position clamp(q), level i of the lane
        const int xr = clampi(q, W);
        const int xm = xr + disp[i];
        const bool out = (unsigned)xm >= (unsigned)W;
        const int xc = (int)min((unsigned)max(xm, 0), (unsigned)(W - 1));
        const float e = dist(rl[xr - R0], ml[xc - M0]);
        return out ? outf : e;
    };
    auto row_of = [&](int m) { return clampi(WIN * m + wv + pr, H); };   // the row wave wv computes in batch m
    const float *sp = S + (size_t)jb * Dp + kbase;     // + row * nst * Dp
    auto state = [&](int y) -> f32x2 {
        if (jb == 0)
            return f32x2{0.0f, 0.0f};
        return *(const f32x2 *)(sp + (size_t)y * (size_t)nst * Dp);
    };
    // the horizontal sums of this wave's row over the block's columns -> tile
    auto h_role = [&](f32x2 st) {
        float cst[NP][LPL];
        if (interior) {
            const f32x2 *mp = ml + (x0 + pl + disp[0] - (DS < 0 ? LPL - 1 : 0) - M0);
            f32x2 pm[NPX];
#pragma unroll
            for (int j = 0; j < NPX; j++)
                pm[j] = mp[j];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const float r = rl[j];                 // (R0 = x0 + pl here)
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    cst[j][i] = dist(r, pm[j + (DS < 0 ? LPL - 1 - i : i)]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; j++)
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    cst[j][i] = cost_at(x0 + pl + j, i);
        }
        float acc[LPL] = {st.x, st.y};
        float *tp = tile + (size_t)wv * WIN * CW + lane * LPL;
#pragma unroll
        for (int c = 0; c < WIN; c++) {
            if (c == 0 && jb == 0) {                   // x = 0: the window sum itself (BoxFilter.cpp:92-107)
#pragma unroll
                for (int j = 1; j <= WIN; j++)
#pragma unroll
                    for (int i = 0; i < LPL; i++)
                        acc[i] += scale * mag(cst[j][i]);
            } else {
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    acc[i] = acc[i] + scale * (mag(cst[c + WIN][i]) - mag(cst[c][i]));
            }
            *(f32x2 *)(tp + c * CW) = f32x2{acc[0], acc[1]};
        }
    };

    // ---- V role: column wv of the block, levels kc + 2 lane, + 1 (as pass B with two levels per lane)
    const int x = x0 + wv;
    const bool col = x < W;
    const int k0 = kc + lane * LPL;
    bool repl[LPL];
    float replv[LPL];
#pragma unroll
    for (int i = 0; i < LPL; i++) {
        const int k = k0 + i;
        const bool live = k < D && col;
        const int xm = x + a.disp0 + (k < D ? k : D - 1) * DS;
        const bool padded = (xm < 0) || (xm >= W);     // PadLine, StcRawCosts.cpp:451-456
        repl[i] = padded || !live;
        replv[i] = live ? outside : __builtin_inff();
    }
    long long *kout = keys + (size_t)(kc / CW) * keys_chunk_stride + (size_t)min(x, W - 1) * H;
    float accv[LPL] = {0.0f, 0.0f}, prev[WIN][LPL];
    int key_lo = 0, key_hi = 0;
    const float *vp = tile + (size_t)wv * CW + lane * LPL;     // + row * WIN * CW

    const int nbat = (H + WIN - 1) / WIN;
    fetch(row_of(-1));
    f32x2 st_next = state(row_of(-1));
    for (int m = -1; m < nbat; m++) {
        if (!(ABL & 2) || m < 0)                       // ablation: the first batch's rows for every batch
        commit();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x2 st = st_next;
        if (m + 1 < nbat) {                            // the next batch's bytes and state travel meanwhile
            const int yn = row_of(m + 1);
            if (!(ABL & 2))
            fetch(yn);
            st_next = state(yn);
        }
        h_role(st);
        __syncthreads();
        float cur[WIN][LPL];
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            const f32x2 v = *(const f32x2 *)(vp + (size_t)s * WIN * CW);
            cur[s][0] = v.x;
            cur[s][1] = v.y;
        }
        __syncthreads();                               // the tile may be overwritten
        if (m < 0) {                                   // rows pl .. pr - 1 (clamped): what leaves the window first
#pragma unroll
            for (int s = 0; s < WIN; s++)
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    prev[s][i] = cur[s][i];
            continue;
        }
        if (m == 0) {                                  // window sum of y = 0: rows pl + 1 .. pr, in this order (BoxFilter.cpp:225-232)
#pragma unroll
            for (int s = 1; s < WIN; s++)
#pragma unroll
                for (int i = 0; i < LPL; i++)
                    accv[i] += scale * prev[s][i];
#pragma unroll
            for (int i = 0; i < LPL; i++) {
                accv[i] += scale * cur[0][i];
                prev[0][i] = cur[0][i];                // y = 0 adds scale * (-a + a) = +0
            }
        }
        const int slot0 = (m % (64 / WIN)) * WIN;
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            float best = 0;
            int bi = 0;
#pragma unroll
            for (int i = 0; i < LPL; i++) {
                accv[i] = accv[i] + scale * (-prev[s][i] + cur[s][i]);
                prev[s][i] = cur[s][i];
                const float v = repl[i] ? replv[i] : accv[i];
                if (i == 0) {
                    best = v;
                } else if (v < best) {                  // strict: the lowest level wins inside the lane
                    best = v;
                    bi = i;
                }
            }
            if (ABL & 1) {                             // ablation: no reduction over the lanes
                key_lo += bi;
                key_hi += __builtin_bit_cast(int, best);
                continue;
            }
            const int so = ordered_bits(best);
            const int mn = wave_min_i32(so);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(so == mn);
            const int wl = (int)__builtin_ctzll(eq);   // lowest lane with the minimum
            const int d = d_first + kc + wl * LPL + __builtin_amdgcn_readlane(bi, wl);
            key_lo = (lane != slot0 + s) ? key_lo : d;
            key_hi = (lane != slot0 + s) ? key_hi : mn;
        }
        if ((m + 1) % (64 / WIN) == 0 || m == nbat - 1) {
            const int g0 = (WIN * m / GRP) * GRP;      // first row of the group
            if (lane < GRP && g0 + lane < H && col) {
                long long v = make_key(key_hi, key_lo);
                if (merge) {
                    const long long o = kout[g0 + lane];
                    v = o < v ? o : v;
                }
                kout[g0 + lane] = v;
            }
        }
    }
}

}  // namespace smk
