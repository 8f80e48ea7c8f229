// EXPERIMENT (not built into the library): pass B for 128-level slabs with 1-KiB row-pair loads regrouped by
// v_permlane32_swap so that every lane owns two levels of both rows.  Bit-exact (it passed
// tests/test_gpu_fused.py::test_slabs_of_65_to_128_levels while it was wired in), 29.5 VALU + 30.6 SALU instructions
// per row (PMC), VALU active 50 %: 3.49 ms at 7680x4320x128 against 3.35 ms for the general pass B, 0.303 against
// 0.269 ms at 1920x1080x128 -- the per-row dependent chain (reduction, ballot, scalar level pick) is what a wave
// waits for, not the 512-byte loads.
//
// smhip_fused_bp.h -- pass B of the fused path for 128-level slabs (65..128 levels: what every GPU runs
// under config 5's sharding at 8 GPUs), "row pair" form.
//
// The general pass B reads a 128-level row as 64 lanes x 2 floats, 512 bytes per wave load; a read-only
// kernel in that order reaches 0.68 of 8 TB/s at 7680x4320, one that takes 1 KiB per load 0.86
// (scripts/membw8.hip), and the general kernel sits at the former (3.35 ms).  In the banded slab a pixel
// column's rows y and y + 1 (y even) are adjacent, 2 x 128 floats = 1 KiB, so here a wave loads a ROW PAIR
// at once -- lanes 0..31 four levels of the even row, lanes 32..63 four levels of the odd row, one dwordx4
// each -- and two v_permlane32_swap per pair regroup the registers so that every lane owns TWO levels of
// BOTH rows (lanes 0..31: levels 4l, 4l+1; lanes 32..63: levels 4l+2, 4l+3).  From there on it is the walk
// of smhip_fused_b.h, row by row: serial vertical running sum per level (BoxFilter.cpp:225-232), PadCosts,
// first minimum inside the lane, DPP minimum over the wave, lowest LEVEL among the lanes that hold it (one
// candidate per half-wave, picked on the scalar unit from the ballots).
// At 512 bytes per wave and row a SIMD has about 50 issue slots per row before it, not HBM, is the limit,
// so the per-row instruction count matters as much as the load width: addresses are a wave-uniform base
// plus a 32-bit lane offset, nothing branches inside the loop, the lane's "second level won" flag is kept
// as a ballot instead of a register.  (A variant with the two rows of a pair in the two half-waves, one
// half-wave reduction for both rows, is in scripts/experiments/smhip_fused_bp_halves.h: the exchanges it
// needs cost what it saves.)
//
// Rows live in a register ring of 16 rows (slot = row & 15; the loop is unrolled over a ring period, so
// every slot is a fixed register): the row entering the window and the row leaving it are both there,
// pairs are requested PB_LA rows ahead and regrouped in place when they are first used.  Keys: lane
// (y & 63) keeps the key of row y; every 64 rows the group goes to the wave's LDS key buffer and from
// there to memory in long runs (as in smhip_fused_b.h).
#pragma once

#include "smhip_fused.h"

namespace smk {

constexpr int PB_RING = 16;     // rows in the register ring (a power of two)
constexpr int PB_LA = 8;        // a pair is requested this many rows before the step that issues it

template <int WIN>
__global__ __launch_bounds__(256) void fused_boxv_wta_pair_kernel(
    const float *__restrict__ T, int tyshift, int W, int H, int D, float scale, int disp0, int dstep,
    float outside, int d_first, int x_per_xcd, long long *__restrict__ keys, int kb_rows)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int Dp = 128, PR = WIN / 2, PL = PR - WIN;
    // a requested pair overwrites rows y + PB_LA - 16 and y + PB_LA - 15, which must have left the window
    // (< y + PL); and it must not be needed by the step that requests it (y + PR < y + PB_LA)
    static_assert(WIN % 2 == 1 && PB_LA % 2 == 0 && PB_LA + 1 - PB_RING < PL && PB_LA > PR, "ring too small");
    extern __shared__ __attribute__((aligned(16))) long long kbuf_all[];
    const int lane = threadIdx.x & 63, l5 = lane & 31;
    const bool upper = lane >= 32;
    const int wpb = blockDim.x >> 6, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xin = ((int)blockIdx.x >> 3) * wpb + wib;
    const int x = ((int)blockIdx.x & 7) * x_per_xcd + xin;          // XCD-aware column order (smhip_fused_b.h)
    if (x >= W || xin >= x_per_xcd)
        return;                                                     // whole wave exits together
    long long *kout = keys + (size_t)x * H;
    long long *kbuf = kbuf_all + (size_t)wib * kb_rows;
    int kb_first = 0;

    // levels after the regrouping: kown, kown + 1 (PadCosts / beyond-the-slab replacement values);
    // levels this lane LOADS: 4 * l5 .. 4 * l5 + 3
    const int kown = 4 * l5 + (upper ? 2 : 0);
    bool repl[2];
    float replv[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int k = kown + i;
        const int xm = x + disp0 + (k < D ? k : D - 1) * dstep;
        repl[i] = k >= D || xm < 0 || xm >= W;                      // PadLine, StcRawCosts.cpp:451-456
        replv[i] = k < D ? outside : __builtin_inff();
    }
    bool need = false;                                              // pass A does not write padded elements:
#pragma unroll                                                      // lanes with nothing to use read one element
    for (int i = 0; i < 4; i++) {
        const int k = 4 * l5 + i;
        const int xm = x + disp0 + (k < D ? k : D - 1) * dstep;
        need = need || (k < D && xm >= 0 && xm < W);
    }
    // address of a pair = wave-uniform row base + 32-bit lane offset in bytes: 16 bytes per lane of the
    // half-wave, the odd row's distance for the upper half (bits do not overlap: or instead of add)
    const char *const tx = (const char *)(T + ((size_t)x << tyshift) * Dp);
    const unsigned lane_bytes = need ? 16u * l5 : 0u;
    const unsigned upper_mask = (need && upper) ? ~0u : 0u;
    // rows a (even; may lie outside the image: replicate border) and a + 1, raw (lane = 4 levels of ITS row)
    auto request = [&](int a) -> f4 {
        const int r0 = clampi(a, H), r1 = clampi(a + 1, H);
        const size_t o0 = slab_row_offset(r0, W, Dp, tyshift), o1 = slab_row_offset(r1, W, Dp, tyshift);
        const unsigned dist = (unsigned)(o1 - o0) * 4u;             // 0 or 512 (a pair never straddles a band)
        return __builtin_nontemporal_load((const f4 *)(tx + o0 * 4 + (lane_bytes | (upper_mask & dist))));
    };
    // raw pair -> (row a: levels kown, kown + 1 in .x .y; row a + 1: in .z .w)
    auto regroup = [&](f4 &v) {
        auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.z), false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.y), __float_as_uint(v.w), false, false);
        v.x = __uint_as_float(s0[0]);
        v.z = __uint_as_float(s0[1]);
        v.y = __uint_as_float(s1[0]);
        v.w = __uint_as_float(s1[1]);
    };
    f4 ring[PB_RING / 2];                                           // ring[p] = rows 2p, 2p + 1 (mod 16)
    // value of level kown + i in the ring row with static slot s
#define PB_ROW(s, i) (((s) & 1) ? ((i) ? ring[((s) & (PB_RING - 1)) >> 1].w : ring[((s) & (PB_RING - 1)) >> 1].z) \
                                : ((i) ? ring[((s) & (PB_RING - 1)) >> 1].y : ring[((s) & (PB_RING - 1)) >> 1].x))

    // prologue: rows PL + 1 .. PB_LA - 1 requested; the window sum of row 0 in the reference's order, except
    // for its last term, row PR, which step 0 adds (its leaving row is taken as 0: -0 + v = v exactly)
    constexpr int A0 = (PL + 1) & ~1;
#pragma unroll
    for (int p = 0; p < PB_RING / 2; p++)
        ring[p] = 0.0f;
#pragma unroll
    for (int a = A0; a < PB_LA; a += 2)
        ring[(a & (PB_RING - 1)) >> 1] = request(a);
#pragma unroll
    for (int a = A0; a <= PR - 1; a += 2)
        regroup(ring[(a & (PB_RING - 1)) >> 1]);
    float acc[2] = {0.0f, 0.0f};
#pragma unroll
    for (int r = PL + 1; r <= PR - 1; r++) {
        acc[0] += scale * PB_ROW(r & (PB_RING - 1), 0);
        acc[1] += scale * PB_ROW(r & (PB_RING - 1), 1);
    }

    int key_lo = 0, key_hi = 0;
    // Every trip runs all 16 steps: rows beyond the image read the clamped last row and their keys are
    // never stored, so the body needs no per-row condition (conditions around ring elements make the
    // register allocator copy the ring).
    for (int y0 = 0; y0 < H; y0 += PB_RING) {
        const int lane_rel = lane - (y0 & 63);
#pragma unroll
        for (int s = 0; s < PB_RING; s++) {
            if ((s & 1) == 0)
                ring[((s + PB_LA) & (PB_RING - 1)) >> 1] = request(y0 + s + PB_LA);
            if (((s + PR) & 1) == 0)                                // the entering row opens a new pair
                regroup(ring[((s + PR) & (PB_RING - 1)) >> 1]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                float od = PB_ROW(s + PL + PB_RING, i);
                if (s == 0)
                    od = y0 == 0 ? 0.0f : od;                       // row 0: the window sum's last term
                acc[i] = acc[i] + scale * (-od + PB_ROW(s + PR, i));
            }
            const float v0 = repl[0] ? replv[0] : acc[0];
            const float v1 = repl[1] ? replv[1] : acc[1];
            const bool second = v1 < v0;                            // strict: the lower level wins inside the lane
            const unsigned long long sec = __builtin_amdgcn_ballot_w64(second);
            const int sb = ordered_bits(second ? v1 : v0);
            const int m = wave_min_i32(sb);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(sb == m);
            // lowest level among the lanes holding the minimum: one candidate per half-wave, scalar work
            const unsigned elo = (unsigned)eq, ehi = (unsigned)(eq >> 32);
            const int tlo = __builtin_ctz(elo | 0x80000000u), thi = __builtin_ctz(ehi | 0x80000000u);
            const int clo = elo ? 4 * tlo + (int)((sec >> tlo) & 1) : 0x7FFFFFFF;
            const int chi = ehi ? 4 * thi + 2 + (int)((sec >> (32 + thi)) & 1) : 0x7FFFFFFF;
            const int d = d_first + min(clo, chi);
            key_lo = (lane_rel == s) ? d : key_lo;
            key_hi = (lane_rel == s) ? m : key_hi;
        }
        // a group of 64 rows is complete, or the column ends
        const bool last = y0 + PB_RING >= H;
        if (((y0 + PB_RING) & 63) == 0 || last) {
            const int g0 = y0 & ~63;
            if (kb_rows == 0) {
                if (g0 + lane < H)
                    kout[g0 + lane] = make_key(key_hi, key_lo);
            } else {
                kbuf[g0 - kb_first + lane] = make_key(key_hi, key_lo);
                if (g0 + 64 - kb_first == kb_rows || last) {
                    const int n = min(kb_rows, H - kb_first);
                    for (int i = lane; i < n; i += 64)
                        kout[kb_first + i] = kbuf[i];
                    kb_first += kb_rows;
                }
            }
        }
    }
#undef PB_ROW
}

}  // namespace smk
