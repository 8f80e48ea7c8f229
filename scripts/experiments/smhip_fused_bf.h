// smhip_fused_bf.h -- pass B of the fused path, lean form (vertical running sum + PadCosts + WTA -> keys).
//
// Same mapping as fused_boxv_wta_vpl_kernel (smhip_fused_b.h): one wave = one pixel column x 64*VPL
// levels, lane l holds VPL consecutive levels, rows walked serially with the leaving row in registers.
// scripts/bench_b.hip says what is left above the load-only time of this shape (0.30 ms at config 3) is
// instruction issue: the wave reduction and the selects around it cost 0.02 ms, the key stores 0.015 ms.
// This form spends about 28 instead of 45 vector instructions per row:
//   * PadCosts and the levels beyond D need no select per row: such a level starts at `outside`
//     (+inf beyond D) and runs with scale 0, so its sum never moves (x + 0 * finite == x).  That needs
//     finite numbers in every slab element a loading lane touches: pass A (smhip_fused_f.h) stores a
//     lane whole when one of its levels is live, in groups of VPL levels; D % VPL == 0;
//   * the wave minimum runs on the floats themselves (v_min_f32 with DPP), so no lane builds ordered
//     integer keys; the winner's key is formed once per row from wave-uniform values on the scalar unit;
//   * a lane takes the minimum of its VPL levels with two instructions; which of them held it is read
//     back for the winning lane only, from three compare masks, on the scalar unit;
//   * the key of row y is dropped into lane (y mod G) with v_writelane (no compare, no select).
// OptWTA's strict '<' (StcOptimize.cpp:40-47): lowest level wins among equal costs -- the lowest lane
// holding the wave minimum, inside it the lowest of its levels; -0 == +0 in every compare used here,
// and the key folds -0 into +0 as ordered_bits does.
#pragma once

#include "../../stereomatch_amd/csrc/smhip_fused_b.h"

#pragma clang diagnostic ignored "-Winline-asm"   // the M0 clobber below

namespace smk {

// min over the 64 lanes (no NaNs), wave-uniform.  v_min_f32 with a DPP source: a lane whose source lane
// does not exist (or whose row is masked off) keeps its value.  Written as asm because the compiler
// wraps a float minimum of a DPP result into a canonicalising v_max_f32 and two moves; the s_nop 1 are
// the two wait states a DPP read of a just-written register needs (the compiler does not look into asm).
__device__ __forceinline__ float wave_min_f32(float v)
{
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"   // lane 15 of each row: row minimum
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" // into rows 1 and 3
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" // into rows 2 and 3: lane 63 holds the minimum
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// blockDim.x = 64 * (pixel columns per block); dynamic LDS = waves * kb_rows * 8 bytes (kb_rows a
// multiple of GRP = (64 / WIN) * WIN).  D % VPL == 0; every slab element of a lane with a live,
// unpadded level holds a finite number.
template <int WIN, int VPL>
__global__ __launch_bounds__(256) void fused_boxv_wta_f_kernel(
    const float *__restrict__ T, int tyshift, int W, int H, int Dp, int D, float scale, int disp0, int dstep,
    float outside, int d_first, int x_per_xcd, long long *__restrict__ keys, size_t keys_chunk_stride, int kb_rows)
{
    static_assert(WIN >= 3 && (WIN & 1) && WIN <= 32, "static odd window");
    constexpr int BLK = WIN;
    constexpr int GRP = (64 / BLK) * BLK;              // rows per key group (63 for WIN 9)
    constexpr int CH = 64 * VPL;
    constexpr int pr = WIN / 2, pl = pr - WIN;
    extern __shared__ __attribute__((aligned(16))) long long kbuf_f[];
    const int lane = threadIdx.x & 63;
    // XCD-aware column order: workgroups are dealt round-robin to the 8 XCDs, so give each
    // XCD a contiguous range of columns.  Correctness does not depend on placement.
    const int wpb = blockDim.x >> 6;
    const int blk = blockIdx.x, wib = threadIdx.x >> 6;
    const int xin = (blk >> 3) * wpb + wib;
    const int x = (blk & 7) * x_per_xcd + xin;
    const int chunk = blockIdx.y;
    if (x >= W || xin >= x_per_xcd)
        return;                                        // whole wave exits together (no barriers below)
    const int k0 = chunk * CH + lane * VPL;            // first level of this lane (k0 + VPL <= Dp)
    const size_t j = ((size_t)x << tyshift) * Dp + k0;
    long long *kout = keys + (size_t)chunk * keys_chunk_stride + (size_t)x * H;
    long long *kbuf = kbuf_f + (size_t)wib * kb_rows;
    int kb_first = 0;                                  // first row held in the LDS key buffer

    // per level: scale (0 when PadCosts overwrites the level or it lies beyond D) and the value it keeps
    float sc[VPL], fixed[VPL];
    bool need = false;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        const int k = k0 + i;
        const bool live = k < D;
        const int xm = x + disp0 + (live ? k : D - 1) * dstep;
        const bool padded = (xm < 0) || (xm >= W);     // PadLine, StcRawCosts.cpp:451-456
        sc[i] = (live && !padded) ? scale : 0.0f;
        fixed[i] = live ? outside : __builtin_inff();
        need = need || (live && !padded);
    }
    auto ld = [&](int row, float(&v)[VPL]) {
        if (need)
            load_row<VPL>(T + slab_row_offset(clampi(row, H), W, Dp, tyshift) + j, v);
        else {
#pragma unroll
            for (int i = 0; i < VPL; i++)
                v[i] = 0.0f;
        }
    };

    float acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; i++)
        acc[i] = 0;
    for (int r = pl + 1; r <= pr; r++) {
        float v[VPL];
        ld(r, v);
#pragma unroll
        for (int i = 0; i < VPL; i++)
            acc[i] += sc[i] * v[i];
    }
#pragma unroll
    for (int i = 0; i < VPL; i++)
        acc[i] = sc[i] != 0.0f ? acc[i] : fixed[i];

    int key_lo = 0, key_hi = 0;
    // slot = lane that keeps the key of this row
    auto step = [&](const float(&od)[VPL], const float(&nw)[VPL], int slot) {
#pragma unroll
        for (int i = 0; i < VPL; i++)
            acc[i] = acc[i] + sc[i] * (-od[i] + nw[i]);
        float best = acc[0];
        if constexpr (VPL == 2) best = __builtin_fminf(acc[0], acc[1]);
        if constexpr (VPL == 4) best = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), __builtin_fminf(acc[2], acc[3]));
        unsigned long long e[VPL > 1 ? VPL - 1 : 1];
#pragma unroll
        for (int i = 0; i + 1 < VPL; i++)
            e[i] = __builtin_amdgcn_ballot_w64(acc[i] == best);
        const float m = wave_min_f32(best);
        const unsigned long long eq = __builtin_amdgcn_ballot_w64(best == m);
        const int wl = (int)__builtin_ctzll(eq);       // lowest lane with the minimum
        int bi = VPL - 1;                              // lowest of its levels holding it
#pragma unroll
        for (int i = VPL - 2; i >= 0; i--)
            bi = ((e[i] >> wl) & 1) ? i : bi;
        const int d = d_first + chunk * CH + wl * VPL + bi;
        int b = __builtin_amdgcn_readfirstlane(__float_as_int(m));
        b = (b == (int)0x80000000) ? 0 : b;            // -0 -> +0, then ordered_bits
        const int s = b ^ ((b >> 31) & 0x7FFFFFFF);
        // lane select in M0 (gfx9 allows one SGPR per vector instruction; nothing else in this kernel
        // uses M0)
        asm("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
            : "+v"(key_lo), "+v"(key_hi)
            : "s"(d), "s"(s), "s"(slot)
            : "m0");
    };
    // after the block that starts at row yb: move the group's keys to LDS when it is complete, and the
    // LDS buffer to memory when it is full or the column ends
    auto flush = [&](int yb) {
        const int g0 = (yb / GRP) * GRP;               // first row of the group
        const bool last = yb + BLK >= H;
        if (yb + BLK - g0 == GRP || last) {
            if (lane < GRP)
                kbuf[g0 - kb_first + lane] = make_key(key_hi, key_lo);
            if (g0 + GRP - kb_first == kb_rows || last) {
                const int n = min(kb_rows, H - kb_first);
                for (int i = lane; i < n; i += 64)
                    kout[kb_first + i] = kbuf[i];
                kb_first += kb_rows;
            }
        }
    };

    float b0[BLK][VPL], b1[BLK][VPL], b2[BLK][VPL];
#pragma unroll
    for (int r = 0; r < BLK; r++) {
        ld(r + pl, b0[r]);
        ld(r + pr, b1[r]);
    }
#pragma unroll
    for (int i = 0; i < VPL; i++)
        b0[0][i] = b1[0][i];   // y = 0 adds scale*(-a + a) = +0: the first value stays the window sum
    auto block = [&](float(&prv)[BLK][VPL], float(&cur)[BLK][VPL], float(&nxt)[BLK][VPL], int yb) {
        const int slot0 = yb % GRP;
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            ld(yb + BLK + r + pr, nxt[r]);              // rolling prefetch, one block ahead
            step(prv[r], cur[r], slot0 + r);
        }
        flush(yb);
    };
    for (int yb = 0; yb < H; yb += 3 * BLK) {
        block(b0, b1, b2, yb);
        if (yb + BLK < H)
            block(b1, b2, b0, yb + BLK);
        if (yb + 2 * BLK < H)
            block(b2, b0, b1, yb + 2 * BLK);
    }
}

}  // namespace smk
