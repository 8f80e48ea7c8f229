// smhip_fused_f.h -- pass A of the fused path for one-band frames, "float lanes" form.
//
// What the measurements of scripts/bench_a.hip and scripts/valu_rate.hip say about pass A on gfx950:
//   * the slab stores alone take 0.34 ms at config 3 (store-only kernel, 512-B wave stores), the
//     arithmetic of the integer kernel (smhip_fused.h) alone 0.23 ms, and the two overlap badly
//     (0.41 ms together): every nanosecond of issue time per step shows up as ~0.3 ns of kernel time;
//   * only v_add/sub/mul_f32 (modifiers included), v_sub_u32, v_and/or/mov issue at the full rate
//     (1.0 ns per wave-instruction and SIMD); min/max/max3/med3, every cvt, SDWA, packed and bit-field
//     instruction takes 1.75 ns; a wave alone on its SIMD issues one instruction per 2 ns whatever it is.
// So this kernel spends as few instructions per level and step as the arithmetic allows:
//   * the staged row lives in LDS as FLOATS (converted once per pixel, not once per level and step) and
//     MatchLine (StcRawCosts.cpp:152-190) is evaluated in float: every value is an integer below 2^24,
//     so the float results equal the reference's integers and scale * (float)(c_new - c_old) is the
//     same product.  With the reference interval a point (no match_interpolated) the distance to the
//     matching interval is |r - med3(lo, hi, r)|: per level and step med3, sub, sub|.|, mul, add;
//   * two levels per lane (512-B wave stores); level i of a lane at step s uses matching pixel
//     s + (LPL-1-i) (dstep -1) or s + i (dstep +1) of the WIN + LPL - 1 pixels the lane reads per block,
//     prefetched one block ahead; two blocks per loop trip, so the ring never moves between registers;
//   * buffer stores: wave-uniform descriptor and step offset, constant lane offset -- no address
//     arithmetic per store; lanes that must not store get an out-of-range offset instead of a branch;
//   * the row is staged in chunks (LDS and prologue independent of the width), and the bytes of the
//     next chunk are fetched into registers while the current one is being worked on.
// Compile with -fno-slp-vectorize (SMHIP build flags): packed f32 instructions cost two plain ones, need
// register shuffles and cannot take the |.| modifiers.
//
// NOTE (gfx950, ROCm 7.2): a buffer_store_dwordx4 whose data registers are rewritten by the VALU instruction right
// behind it stores the NEW value in the last lanes of every 16 -- the ">64-bit store data" hazard of the ISA
// manual.  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) exempts MUBUF stores whose soffset is
// an SGPR, the hardware does not: scripts/experiments/store_hazard.hip places the instructions by hand -- SGPR
// soffset, rewrite at once: lanes 12-15 of every 16 wrong; one wait state (s_nop 0) between: clean; soffset 0:
// lanes 8-15 wrong at once, lanes 12-15 still wrong after one wait state (the compiler puts two there itself).
// So the wait states are missing from the COMPILER's output for the form used here (SGPR step offset); a
// dwordx4 store in this kernel is followed by an explicit s_nop 1 (put(), LPL == 4).  dwordx2 / dword stores are
// not affected.  profiles/r03_store_hazard.log, tests/test_gpu_store_hazard.py.
#pragma once

#include "smhip_fused_defs.h"

namespace smk {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// blockDim.x = 64 * NW; grid (H, Dp / (64 * LPL * NW)).  One band, |dstep| == 1 (DS), reference
// interval a point (a.rn == a.rx).  CUT = false: cutoff >= the largest possible cost (no truncation).
// cb = blocks (of WIN steps) per staged chunk; nref = cb * WIN + WIN, nmatch = nref + 64 * LPL * NW, both
// <= FUSED_F_NI * 64 * NW.  Dynamic LDS: nmatch * 8 + nref * 4 bytes.  W * (Dp << tyshift) * 4 < 2^31.
// glev = levels per lane of the pass B that follows (1, 2 or 4): the granule of the store predicate.
constexpr int FUSED_F_NI = 6;                          // staged pixels per thread and chunk (at most)
constexpr int fused_f_ni(int lpl) { return lpl == 4 ? 12 : FUSED_F_NI; }   // (a 256-level wave alone stages its row)

// STATES = true (the recomputing form of the fused path, smhip_fused_c.h): the same walk, but instead of the slab the
// kernel writes only the running sum at the last column before every block of WIN columns -- state slot j of a row
// holds t[WIN * j - 1], j = 1 .. ceil(W / WIN) - 1, at S[(y * nst + j) * Dp + k] (T = S, the `tyshift` argument = nst,
// the slots per row): 4 N / WIN bytes instead of 4 N.  All levels are written (no dead elements).
template <int WIN, bool SD, bool CUT, int DS, int LPL, int NW, int ABL = 0, int POL = 0, bool PAIRED = false, bool W3 = false, bool STATES = false>   // ABL: ablations of scripts/bench_a.hip; POL: cache policy bits of the stores
// Waves per SIMD: the kernels of windows up to 9 need 157 VGPRs or fewer, so three waves fit a SIMD; with the
// bound of two that this kernel had, a CU held four two-wave workgroups, the chip 1024, and config 3's 1080 rows
// ran as one round of 1024 and a tail of 56: 0.379 ms, against 0.353 ms with all rows resident (1024 rows alone
// take 0.314 ms, 0.80 of 8 TB/s).  Window 11 takes 180 VGPRs: W3 = true forces it under 168 (44-60 bytes of scratch
// per lane), which costs about 3 % and saves the tail (config 3: 0.379 -> 0.363 ms) -- used when no LDS floor bounds
// the residency anyway (frames below 1280 rows).
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(W3 ? 3 : 1, ((WIN <= 9 && LPL <= 2) || W3) ? 3 : 2)))
void fused_rawbox_h_f_kernel(FusedA a, float *__restrict__ T, int tyshift,
                                                                   int Dp, float scale, int keep_all, int cb,
                                                                   int nref, int nmatch, int glev)
{
    static_assert(WIN >= 3 && (WIN & 1), "static odd window");
    static_assert(DS == 1 || DS == -1, "unit disparity step");
    static_assert(LPL == 1 || LPL == 2 || LPL == 4, "levels per lane");
    constexpr int NT = 64 * NW;                        // threads per workgroup
    constexpr int CW = 64 * LPL * NW;                  // levels per workgroup
    constexpr int NPX = WIN + LPL - 1;                 // matching pixels a lane reads per block
    constexpr int NI = fused_f_ni(LPL);
    constexpr int pr = WIN / 2, pl = pr - WIN;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_f[];
    f32x2 *ml = (f32x2 *)lds_f;                        // matching interval (lo, hi) of positions M0 ..
    float *rl = (float *)(ml + nmatch);                // reference value of positions R0 ..
    // Two levels per lane put neighbouring lanes two positions (16 bytes) apart: read as one array, a wave's
    // ds_read_b64 hits every bank four times (16.9 M conflict cycles per launch at config 3).  SPLIT keeps the
    // even and the odd positions in two arrays, so the lanes of one read are 8 bytes apart: conflict-free (0.5 M).
    // The parity of a lane's first position is the same in every lane of the wave.  Measured (-DSMK_F_SPLIT=1,
    // profiles/r03_pass_a_lds_split.log): the conflicts go, the time does not -- 5 % more VALU instructions and one
    // more LDS instruction per block (two runs of five positions instead of one of ten), and pass A is no faster
    // (traced 353.8 against 346.4 / 350.4 us); the LDS pipe is 10 % active and not what pass A waits for.  Off in the product.
#ifndef SMK_F_SPLIT
#define SMK_F_SPLIT 0
#endif
    constexpr bool SPLIT = SMK_F_SPLIT && LPL == 2;
    f32x2 *mle = ml, *mlo = ml + (nmatch + 1) / 2;     // SPLIT: positions M0 + 2 i / M0 + 2 i + 1
    auto mlat = [&](int i) -> f32x2 & { return SPLIT ? ((i & 1) ? mlo : mle)[i >> 1] : ml[i]; };
    const int W = a.W, D = a.D;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    // workgroup index = row * chunks + chunk: the chunks of a row are dispatched together, so the halves /
    // quarters of the row's slab lines are written at about the same time whatever the occupancy
    const int nck = Dp / CW;
    const int y = (int)blockIdx.x / nck;
    const int kc = ((int)blockIdx.x - y * nck) * CW;   // first level of this workgroup
    const int kq = (wib * 64 + lane) * LPL;            // first level of this lane inside the chunk
    // lanes beyond D shadow the last live lane, so disp[i] = disp[0] + i * DS in every lane
    const int Dl = (D + LPL - 1) / LPL * LPL;
    const int kbase = min(kc + kq, Dl - LPL);
    int disp[LPL];
#pragma unroll
    for (int i = 0; i < LPL; i++)
        disp[i] = a.disp0 + (kbase + i) * DS;
    const int dA = a.disp0 + kc * DS, dB = a.disp0 + (min(kc + CW, Dl) - 1) * DS;
    const int dlo = min(dA, dB), dhi = max(dA, dB);
    // Positions q with 0 <= q + disp < W for every level of the workgroup: no bounds tests there.
    const int qa = max(0, -dlo), qb = min(W, W - dhi);
    const float cutf = (float)a.cutoff, outf = (float)a.outside;
    const uint8_t *mrow = a.mn + (size_t)y * a.pitch, *xrow = a.mx + (size_t)y * a.pitch;
    const uint8_t *rrow = a.rn + (size_t)y * a.pitch;
    const int nblocks = (W - 1 + WIN - 1) / WIN;       // block b = steps x = 1 + WIN*b .. WIN + WIN*b
    // blocks [b_int0, b_int1) are interior; blocks below b_edge never clamp the reference position
    const int b_int0 = qa <= 1 ? 0 : (qa - 1 + WIN - 1) / WIN;
    const int b_int1 = (qb - WIN - pr - 1 >= 0) ? (qb - WIN - pr - 1) / WIN + 1 : 0;
    const int b_edge = (W - WIN - pr - 1 >= 0) ? (W - WIN - pr - 1) / WIN + 1 : 0;

    // ---- chunk geometry and staging ----
    int R0 = 0, M0 = 0;                                // first staged positions of the current chunk
    struct Geo { int R0, R1, M0, M1; };
    auto geometry = [&](int c0) -> Geo {
        const int c1 = min(c0 + cb, nblocks);
        const int xa = c0 == 0 ? 0 : 1 + WIN * c0, xe = min(W - 1, WIN * c1);
        Geo g;
        g.R0 = xa;
        g.R1 = min(xe + pr, W - 1);
        g.M0 = clampi(g.R0 + dlo, W);
        g.M1 = clampi(g.R1 + dhi, W);
        return g;
    };
    // raw bytes of a chunk, fetched one chunk ahead: matching pixel and its neighbours (or the two
    // interval images), reference pixel
    uint8_t f_cur[NI], f_prv[NI], f_nxt[NI], f_ref[NI];
    auto fetch = [&](const Geo &g) {
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int i = threadIdx.x + j * NT;
            if (ABL & 2) {                             // ablation: no global loads
                f_cur[j] = (uint8_t)(i * 7 + y); f_prv[j] = (uint8_t)(i * 5); f_nxt[j] = (uint8_t)(i * 3 + y); f_ref[j] = (uint8_t)(i * 11 + y);
                continue;
            }
            if (i <= g.M1 - g.M0) {
                const int p = g.M0 + i;
                if (a.bt_inline) {
                    f_cur[j] = mrow[p];
                    f_prv[j] = mrow[p > 0 ? p - 1 : p];
                    f_nxt[j] = mrow[p < W - 1 ? p + 1 : p];
                } else {
                    f_cur[j] = mrow[p];
                    f_nxt[j] = xrow[p];
                }
            }
            if (i <= g.R1 - g.R0)
                f_ref[j] = rrow[g.R0 + i];
        }
    };
    auto commit = [&](const Geo &g) {
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int i = threadIdx.x + j * NT;
            if (i <= g.M1 - g.M0) {
                const int p = g.M0 + i;
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
                mlat(i) = v;
            }
            if (i <= g.R1 - g.R0)
                rl[i] = (float)f_ref[j];
        }
    };

    // ---- arithmetic ----
    // The ring keeps, per level and window slot, a value whose magnitude is the raw cost: without
    // squaring and truncation the signed difference itself (|.| is a free source modifier), so the
    // new cost is born in the register that stays in the ring.
    constexpr bool RAW = !SD && !CUT;
    auto mag = [](float v) -> float { return RAW ? __builtin_fabsf(v) : v; };
    // distance between reference value r and the matching interval m (StcRawCosts.cpp:168-185)
    auto dist = [&](float r, f32x2 m) -> float {
        float q = r - __builtin_amdgcn_fmed3f(m.x, m.y, r);
        if (RAW) return q;
        float e = SD ? q * q : __builtin_fabsf(q);
        if (CUT) e = __builtin_fminf(e, cutf);
        return e;
    };
    auto cost = [&](int xr, int i) -> float {          // MatchLine for reference pixel xr, level i of the lane
        int xm = xr + disp[i];
        bool out = (unsigned)xm >= (unsigned)W;
        int xc = (int)min((unsigned)max(xm, 0), (unsigned)(W - 1));
        float e = dist(rl[xr - R0], mlat(xc - M0));
        return out ? outf : e;
    };
    // Slab addresses: buffer stores with a wave-uniform descriptor (this row at x = 0, first level of the
    // workgroup), a wave-uniform byte offset per step and a 32-bit lane offset.  Lanes beyond D write what
    // the lane they shadow writes, to the same place; an offset of `oob` (= the descriptor's size) makes
    // the hardware drop the store.
    // PAIRED (128-level slabs, one wave per row): the slab keeps the 128 levels of columns 2c and 2c + 1 side by
    // side -- T[band][c][ty][x & 1][128], i.e. a slab of (W + 1) / 2 "columns" of 256 floats -- so that pass B
    // reads a row of two columns with one 1-KiB wave load (smhip_fused_b.h, PAIR).  Step x then lies at
    // (x >> 1) * pair stride + (x & 1) * 512 bytes instead of x * xsb.
    static_assert(!PAIRED || (NW == 1 && LPL == 2), "paired layout: 128-level slabs");
    static_assert(!STATES || !PAIRED, "states: plain level order");
    const unsigned xsb = STATES ? (unsigned)Dp * (unsigned)sizeof(float)                           // bytes between consecutive state slots
                                : ((unsigned)(PAIRED ? 256 : Dp) << tyshift) * (unsigned)sizeof(float);   // bytes between consecutive x (PAIRED: pairs)
    const int oob = (int)((unsigned)(STATES ? tyshift : PAIRED ? (W + 1) / 2 : W) * xsb);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        T + (STATES ? (size_t)y * (size_t)tyshift * Dp
                    : PAIRED ? slab_row_offset(y, (W + 1) / 2, 256, tyshift) : slab_row_offset(y, W, Dp, tyshift)) + kc, 0, oob,
        0x00020000);
    const int lane_off = (kbase - kc) * 4;
    unsigned tb = 0;                                   // byte offset of the current block's first step
    int px = 0;                                        // PAIRED: parity of that step's x
    // byte offset of step s of the current block / moving on by one block of WIN (odd) steps
    auto soff = [&](int s) -> unsigned {
        if constexpr (!PAIRED)
            return tb + s * xsb;
        else
            return tb + (px ? (unsigned)((s + 1) >> 1) * xsb - (unsigned)(s & 1) * 512u
                            : (unsigned)(s >> 1) * xsb + (unsigned)(s & 1) * 512u);
    };
    auto next_block = [&]() {
        if constexpr (STATES) {
            tb += xsb;                                 // one slot per block
        } else if constexpr (!PAIRED) {
            tb += WIN * xsb;
        } else {
            static_assert(!PAIRED || (WIN & 1), "odd windows");
            tb += px ? (unsigned)((WIN + 1) / 2) * xsb - 512u : (unsigned)(WIN / 2) * xsb + 512u;
            px ^= 1;
        }
    };
    auto put = [&](int voff, unsigned off, const float (&acc)[LPL]) {
        if constexpr (LPL == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[0]), trs, voff, off, POL | (ABL >> 4));
        } else if constexpr (LPL == 2) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 v = {__builtin_bit_cast(unsigned, acc[0]), __builtin_bit_cast(unsigned, acc[1])};
            __builtin_amdgcn_raw_buffer_store_b64(v, trs, voff, off, POL | (ABL >> 4));
        } else {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v = {__builtin_bit_cast(unsigned, acc[0]), __builtin_bit_cast(unsigned, acc[1]),
                       __builtin_bit_cast(unsigned, acc[2]), __builtin_bit_cast(unsigned, acc[3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, trs, voff, off, POL | (ABL >> 4));
            // two wait states before anything may rewrite the data registers (see the note at the top): the
            // compiler's hazard recogniser does not cover a store with an SGPR soffset
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // does one level of the lane's group of glev levels (one lane of pass B, glev a multiple of LPL)
    // survive PadCosts at pixel x: pass B loads such a group whole and wants numbers in all of it
    const int kg = kbase / glev * glev;
    const int dg0 = a.disp0 + kg * DS, dg1 = a.disp0 + (kg + glev - 1) * DS;
    auto lane_live = [&](int x) -> bool {
        return (unsigned)(x + dg0) < (unsigned)W || (unsigned)(x + dg1) < (unsigned)W;
    };

    float acc[LPL], ring[LPL][WIN];
    // the lane's NPX matching pixels of block b start at the lowest position it needs
    auto pos0 = [&](int b) { return 1 + WIN * b + pr + disp[0] - (DS < 0 ? LPL - 1 : 0); };
    // SPLIT: where the lane's positions e, e + 2, .. (pa) and e + 1, e + 3, .. (pb) of the next block to load lie;
    // WIN is odd, so from one block to the next the two arrays change roles
    const f32x2 *pa = mle, *pb = mlo;
    auto aim = [&](int b) {
        if constexpr (SPLIT) {
            const int e = pos0(b) - M0;                // (its parity is wave-uniform)
            pa = ((e & 1) ? mlo : mle) + (e >> 1);
            pb = (e & 1) ? mle + ((e + 1) >> 1) : mlo + (e >> 1);
        }
    };
    auto load_block = [&](f32x2 (&pm)[NPX], float (&rv)[WIN], int b) {     // blocks in sequence after aim(first block)
        const float *rp = rl + (1 + WIN * b + pr - R0);
        if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < NPX; j++)
                pm[j] = (j & 1) ? pb[j >> 1] : pa[j >> 1];
            const f32x2 *t = pa;
            pa = pb + (WIN - 1) / 2;
            pb = t + (WIN + 1) / 2;
        } else {
            const f32x2 *mp = ml + (pos0(b) - M0);
#pragma unroll
            for (int j = 0; j < NPX; j++)
                pm[j] = mp[j];
        }
#pragma unroll
        for (int s = 0; s < WIN; s++)
            rv[s] = rp[s];
    };
    // one block of interior steps
    auto fast_block = [&](const f32x2 (&pm)[NPX], const float (&rv)[WIN]) {
#pragma unroll
        for (int s = 0; s < WIN; s++) {
#pragma unroll
            for (int i = 0; i < LPL; i++) {
                float c = dist(rv[s], pm[s + (DS < 0 ? LPL - 1 - i : i)]);
                acc[i] = acc[i] + scale * (mag(c) - mag(ring[i][s]));
                ring[i][s] = c;
            }
            if constexpr (STATES) {
                if (s == WIN - 2)                      // x = WIN * (b + 1) - 1: the state in front of the next block
                    put(lane_off, tb, acc);
            } else if (!(ABL & 1))
                put(lane_off, soff(s), acc);
        }
        next_block();
    };
    // one block whose matching positions may leave the row (its reference positions do not): the same
    // walk with clamped pixel fetches, `outside` for the invalid ones and stores switched off per lane
    auto edge_block = [&](int b) {
        const int xb = 1 + WIN * b, p0 = pos0(b);
        f32x2 pm[NPX];
        bool ok[NPX];
        float rv[WIN];
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            ok[j] = (unsigned)(p0 + j) < (unsigned)W;
            pm[j] = mlat(clampi(p0 + j, W) - M0);
        }
#pragma unroll
        for (int s = 0; s < WIN; s++)
            rv[s] = rl[xb + s + pr - R0];
#pragma unroll
        for (int s = 0; s < WIN; s++) {
#pragma unroll
            for (int i = 0; i < LPL; i++) {
                const int j = s + (DS < 0 ? LPL - 1 - i : i);
                float c = ok[j] ? dist(rv[s], pm[j]) : outf;
                acc[i] = acc[i] + scale * (mag(c) - mag(ring[i][s]));
                ring[i][s] = c;
            }
            if constexpr (STATES) {
                if (s == WIN - 2)
                    put(lane_off, tb, acc);
            } else if (!(ABL & 1))
                put(lane_off + (-(int)!(keep_all || lane_live(xb + s)) & oob), soff(s), acc);   // (a ?: here became a table in scratch)
        }
        next_block();
    };
    // the general walk (blocks at the right end of the row, where the reference position is clamped)
    auto general_block = [&](int b) {
        const int xb = 1 + WIN * b;
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            const int x = xb + s;
            if (x < W) {
                const int xr = min(x + pr, W - 1);
#pragma unroll
                for (int i = 0; i < LPL; i++) {
                    float c = cost(xr, i);
                    acc[i] = acc[i] + scale * (mag(c) - mag(ring[i][s]));
                    ring[i][s] = c;
                }
                if constexpr (STATES) {
                    if (s == WIN - 2)
                        put(lane_off, tb, acc);
                } else if (!((ABL & 1) && x + 2 < W))
                    put(lane_off + (-(int)!(keep_all || lane_live(x)) & oob), soff(s), acc);
            }
        }
        next_block();
    };

    Geo g = geometry(0);
    fetch(g);
    for (int c0 = 0; c0 < nblocks || c0 == 0; c0 += cb) {
        const int c1 = min(c0 + cb, nblocks);
        if (c0 > 0)
            __syncthreads();                           // everybody has left the previous chunk
        commit(g);
        R0 = g.R0;
        M0 = g.M0;
        __syncthreads();
        if (c0 + cb < nblocks) {                       // the next chunk's bytes travel while this one is worked on
            g = geometry(c0 + cb);
            fetch(g);
        }

        if (c0 == 0) {                                 // x = 0: the window sum itself (BoxFilter.cpp:92-107)
#pragma unroll
            for (int i = 0; i < LPL; i++)
                acc[i] = 0;
#pragma unroll
            for (int j = 0; j < WIN; j++) {
                const int xr = clampi(pl + 1 + j, W);
#pragma unroll
                for (int i = 0; i < LPL; i++) {
                    float c = cost(xr, i);
                    ring[i][j] = c;
                    acc[i] += scale * mag(c);
                }
            }
            if constexpr (!STATES)
                put(lane_off, 0, acc);
            tb = PAIRED ? 512u : xsb;                  // tb = offset of step x = 1 + WIN * b from here on (STATES: of slot b + 1)
            px = 1;
        }

        int b = c0;
        while (b < c1) {
            if (b >= b_int0 && b < b_int1) {
                const int be = min(c1, b_int1);        // interior blocks are one interval: [b, be) here
                f32x2 mA[NPX], mB[NPX];
                float rA[WIN], rB[WIN];
                aim(b);
                load_block(mA, rA, b);
                for (; b + 2 <= be; b += 2) {          // two blocks per trip: the ring and the LDS
                    load_block(mB, rB, b + 1);         // prefetch alternate between two register sets
                    fast_block(mA, rA);
                    if (b + 2 < be)
                        load_block(mA, rA, b + 2);
                    fast_block(mB, rB);
                }
                if (b < be) {
                    fast_block(mA, rA);
                    b++;
                }
            } else if (b < b_edge) {
                edge_block(b);
                b++;
            } else {
                general_block(b);
                b++;
            }
        }
    }
}

}  // namespace smk
