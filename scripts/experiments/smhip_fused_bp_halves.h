// EXPERIMENT (not built into the library): pass B for 128-level slabs with the two rows of a pair in the two
// half-waves (four levels per lane).  Bit-exact, 1-KiB loads, one half-wave reduction serves two rows -- but the
// leaving-row exchange, the delta broadcast and the row select cost what the shared reduction saves: 83 VALU
// instructions per row pair (PMC: VALU active 78 %), 3.39 ms at 7680x4320x128 against 3.35 ms for the general pass B.
//
// smhip_fused_bp.h -- pass B of the fused path for 128-level slabs (65..128 levels: what every GPU runs
// under config 5's sharding at 8 GPUs), "row pair" form.
//
// Two things hold the general pass B back on these slabs (7680x4320x128: 3.35 ms, 0.64 of 8 TB/s):
//  * it reads a 128-level row as 64 lanes x 2 floats, 512 bytes per wave load; a read-only kernel in that
//    order reaches 0.68, one that takes 1 KiB per load 0.86 (scripts/membw8.hip);
//  * the per-row work that does not depend on the number of levels (wave reduction, key bookkeeping) is
//    spread over half as many bytes: at 512 bytes per wave and row a SIMD has about 50 issue slots per row
//    before it, not HBM, is the limit.
// In the banded slab two consecutive rows of a pixel column are adjacent (2 x 128 floats = 1 KiB), so here
// a wave works on a ROW PAIR per step: lanes 0..31 hold four levels of the first row of the pair, lanes
// 32..63 the same four levels of the second row (one dwordx4 per lane, 1 KiB per wave load), and most
// instructions of a step serve two rows:
//  * the window update delta = scale * (-leaving + entering) is formed for both rows at once.  The entering
//    rows are a loaded pair as it is; the leaving rows straddle two loaded pairs, whose halves are exchanged
//    with one v_permlane32_swap per register (pairs start at the row parity that makes the ENTERING rows
//    aligned: even rows for windows 5 and 9, odd rows for 3, 7 and 11);
//  * the serial vertical running sum (BoxFilter.cpp:225-232) is kept by every lane for its four levels:
//    one more swap hands each half the other row's delta, then acc(y) = acc(y-1) + delta(y) and
//    acc(y+1) = acc(y) + delta(y+1) in the reference's order, and a lane keeps the value of its own row;
//  * PadCosts, the first minimum of the lane's four levels, then ONE DPP minimum over each half-wave gives
//    both rows' minima (lanes 31 and 63); the lowest lane of each half that holds it names the level.
// Loaded pairs live in a register ring of RP pairs (the loop is unrolled over a ring period, every slot a
// fixed register) from the request, LAP steps ahead, until their second row has left the window.  Keys:
// lane (y mod 4 RP) keeps the key of row y; complete groups go to the wave's LDS key buffer and from there
// to memory in long runs (as in smhip_fused_b.h).
#pragma once

#include "smhip_fused.h"

namespace smk {

template <int WIN, int RP>
__global__ __launch_bounds__(256) void fused_boxv_wta_pair_kernel(
    const float *__restrict__ T, int tyshift, int W, int H, int D, float scale, int disp0, int dstep,
    float outside, int d_first, int x_per_xcd, long long *__restrict__ keys, int kb_rows)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int Dp = 128, PR = WIN / 2, PL = PR - WIN;
    constexpr int Q = PR & 1;                   // pair p = rows 2p + Q, 2p + Q + 1
    constexpr int NWO = (PR - Q) / 2;           // step j: entering rows = pair j + NWO
    constexpr int ODB = (PR + Q) / 2;           //         leaving rows  = second row of pair j - ODB - 1, first of j - ODB
    constexpr int LAP = RP - PR - 2;            // pairs requested ahead: as many as the ring allows
    constexpr int GRP = 4 * RP;                 // rows per key group (two ring periods)
    static_assert(WIN % 2 == 1 && LAP >= 1 && GRP <= 64, "ring too small for this window");
    extern __shared__ __attribute__((aligned(16))) long long kbuf_all[];
    const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
    const bool upper = half != 0;
    const int wpb = blockDim.x >> 6, wib = threadIdx.x >> 6;
    const int xin = ((int)blockIdx.x >> 3) * wpb + wib;
    const int x = ((int)blockIdx.x & 7) * x_per_xcd + xin;          // XCD-aware column order (smhip_fused_b.h)
    if (x >= W || xin >= x_per_xcd)
        return;                                                     // whole wave exits together
    long long *kout = keys + (size_t)x * H;
    long long *kbuf = kbuf_all + (size_t)wib * kb_rows;
    int kb_first = 0;

    // the lane's levels 4 * l5 .. + 3: PadCosts / beyond-the-slab replacement values
    bool repl[4];
    float replv[4];
    bool need = false;                                              // pass A does not write padded elements:
#pragma unroll                                                      // lanes with nothing to use read one element
    for (int i = 0; i < 4; i++) {
        const int k = 4 * l5 + i;
        const int xm = x + disp0 + (k < D ? k : D - 1) * dstep;
        const bool padded = xm < 0 || xm >= W;                      // PadLine, StcRawCosts.cpp:451-456
        repl[i] = padded || k >= D;
        replv[i] = k < D ? outside : __builtin_inff();
        need = need || !repl[i];
    }
    const float *const tcol = T + ((size_t)x << tyshift) * Dp + 4 * l5;
    const int tymask = (1 << tyshift) - 1;
    const size_t band_elems = ((size_t)W << tyshift) * Dp;
    // element offset of row r (clamped: replicate border) in this column
    auto row_off = [&](int r) -> size_t {
        r = clampi(r, H);
        return (size_t)(r >> tyshift) * band_elems + (size_t)((r & tymask) * Dp);
    };
    // both halves row r (prologue) / pair p: each half its own row.  No branch around the load: lanes that
    // need nothing read their column's first element (a branch per request makes the allocator copy the ring)
    auto load_row = [&](int r) -> f4 {
        return __builtin_nontemporal_load((const f4 *)(tcol + (need ? row_off(r) : 0)));
    };
    auto load_pair = [&](int p) -> f4 {
        return __builtin_nontemporal_load((const f4 *)(tcol + (need ? row_off(2 * p + Q + half) : 0)));
    };
    auto swap32 = [&](float a, float b, float &a_out, float &b_out) {   // a.hi <-> b.lo
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a_out = __uint_as_float(r[0]);
        b_out = __uint_as_float(r[1]);
    };

    f4 ring[RP];
    // prologue: pairs the first steps use (entering pairs 0 + NWO .., leaving pairs -ODB - 1, -ODB ..)
#pragma unroll
    for (int p = -ODB - 1; p < NWO + LAP; p++)
        ring[(p + 4 * RP) % RP] = load_pair(p);
    // window sum of row 0 in the reference's order, except for its last term, row PR, which step 0 adds
    // (the leaving row of row 0 is taken as 0: -0 + v = v exactly); every lane sums its four levels
    f4 acc = 0.0f;
#pragma unroll
    for (int r = PL + 1; r <= PR - 1; r++)
        acc += scale * load_row(r);

    int key_lo = 0, key_hi = 0;
    // Every trip runs all RP steps: rows beyond the image read the clamped last row and their keys are never
    // stored, so the body needs no per-row condition.
    for (int j0 = 0; 2 * j0 < H; j0 += RP) {
        const int lane_rel = lane - ((2 * j0) % GRP);
#pragma unroll
        for (int s = 0; s < RP; s++) {
            ring[(s + NWO + LAP) % RP] = load_pair(j0 + s + NWO + LAP);
            // leaving rows: A = pair j - ODB - 1 (its second row, for the first row of this step),
            //               B = pair j - ODB     (its first row, for the second row of this step)
            f4 &A = ring[(s - ODB - 1 + 4 * RP) % RP], &B = ring[(s - ODB + 4 * RP) % RP];
            f4 od;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float ap, bp;
                swap32(A[i], B[i], ap, bp);                         // ap = [A.lo, B.lo], bp = [A.hi, B.hi]
                od[i] = upper ? ap : bp;
                B[i] = bp;                                          // next step takes B's second row from here
            }
            if (s == 0)
                od = (j0 == 0 && !upper) ? (f4)0.0f : od;           // row 0: the window sum's last term
            const f4 delta = scale * (-od + ring[(s + NWO) % RP]);
            f4 d0, d1;                                              // delta of the first / second row, in all lanes
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float lo, hi;
                swap32(delta[i], delta[i], lo, hi);                 // [d.lo, d.lo], [d.hi, d.hi]
                d0[i] = lo;
                d1[i] = hi;
            }
            const f4 acc0 = acc + d0;
            acc = acc0 + d1;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v[i] = upper ? acc[i] : acc0[i];
                v[i] = repl[i] ? replv[i] : v[i];
            }
            float best = v[0];
            int bi = 0;
#pragma unroll
            for (int i = 1; i < 4; i++) {
                const bool c = v[i] < best;                         // strict: the lowest level wins inside the lane
                best = c ? v[i] : best;
                bi = c ? i : bi;
            }
            const int sb = ordered_bits(best);
            int m = sb;
            m = dpp_min<0x111, 0xF>(m);                             // row_shr:1
            m = dpp_min<0x112, 0xF>(m);                             // row_shr:2
            m = dpp_min<0x114, 0xF>(m);                             // row_shr:4
            m = dpp_min<0x118, 0xF>(m);                             // row_shr:8
            m = dpp_min<0x142, 0xA>(m);                             // row_bcast:15 -> lanes 31, 63: the halves' minima
            const int m0 = __builtin_amdgcn_readlane(m, 31), m1 = __builtin_amdgcn_readlane(m, 63);
            const unsigned e0 = (unsigned)__builtin_amdgcn_ballot_w64(sb == m0);
            const unsigned e1 = (unsigned)(__builtin_amdgcn_ballot_w64(sb == m1) >> 32);
            const int w0 = __builtin_ctz(e0), w1 = __builtin_ctz(e1);      // (never zero: the minimum sits in its half)
            const int k0 = d_first + 4 * w0 + __builtin_amdgcn_readlane(bi, w0);
            const int k1 = d_first + 4 * w1 + __builtin_amdgcn_readlane(bi, 32 + w1);
            key_lo = (lane_rel == 2 * s) ? k0 : key_lo;
            key_hi = (lane_rel == 2 * s) ? m0 : key_hi;
            key_lo = (lane_rel == 2 * s + 1) ? k1 : key_lo;
            key_hi = (lane_rel == 2 * s + 1) ? m1 : key_hi;
        }
        // a key group is complete, or the column ends
        const int y_next = 2 * (j0 + RP);
        const bool last = y_next >= H;
        if (y_next % GRP == 0 || last) {
            const int g0 = (2 * j0) / GRP * GRP;
            if (kb_rows == 0) {
                if (lane < GRP && g0 + lane < H)
                    kout[g0 + lane] = make_key(key_hi, key_lo);
            } else {
                if (lane < GRP)
                    kbuf[g0 - kb_first + lane] = make_key(key_hi, key_lo);
                if (g0 + GRP - kb_first == kb_rows || last) {
                    const int n = min(kb_rows, H - kb_first);
                    for (int i = lane; i < n; i += 64)
                        kout[kb_first + i] = kbuf[i];
                    kb_first += kb_rows;
                }
            }
        }
    }
}

}  // namespace smk
