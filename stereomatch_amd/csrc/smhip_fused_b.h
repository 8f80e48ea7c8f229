// smhip_fused_b.h -- pass B of the fused path, "VPL levels per lane" form.
//
// One WAVE owns one pixel column x and a chunk of 64*VPL consecutive disparity levels:
// lane l holds levels chunk*64*VPL + l*VPL .. + VPL-1, so one row of the chunk is ONE
// vector load of VPL floats per lane (VPL = 4: global_load_dwordx4, 1 KiB per wave,
// non-temporal: T is read exactly once).  The lane walks y serially (the vertical running
// sum of BoxFilter.cpp:225-232 is a serial recurrence per (x,d) chain), applies PadCosts,
// takes the first minimum of its VPL levels, and the 64 lanes min-reduce the ordered cost
// with DPP; the lowest lane holding the minimum wins (= lowest d, OptWTA's strict '<').
// No LDS, no barrier: a 256-level pixel is finished inside one wave.
//
// Keys are written COLUMN-MAJOR (key of pixel (x,y) at x*H + y): lane (y mod G) keeps the key
// of row y and the wave puts G consecutive keys (G = 63 or 64 rows) into a wave-private LDS
// buffer of kb_rows rows, which goes to memory in one run when it is full or the column ends.
// Every store into the read stream costs an HBM bus turn-around: one scattered 8-byte store per row
// made pass B 0.36 ms, one 504-byte store per 63 rows 0.33 ms, one 9-KB run per column 0.31 ms
// (scripts/bench_b.hip: without any key store 0.31 ms).
#pragma once

#include "smhip_fused.h"

namespace smk {

// rows per block of pass B for a static window (0: the generic kernel); shared with the host's key-buffer sizing
constexpr int passb_blk(int win) { return win == 0 ? 8 : win == 3 ? 9 : win == 5 ? 10 : win; }

template <int VPL> struct vecf { typedef float type __attribute__((ext_vector_type(VPL))); };
template <> struct vecf<1> { typedef float type; };

template <int VPL>
__device__ __forceinline__ void load_row(const float *p, float (&v)[VPL])
{
    typedef typename vecf<VPL>::type V;
    V t = __builtin_nontemporal_load((const V *)p);
    if constexpr (VPL == 1) {
        v[0] = t;
    } else {
#pragma unroll
        for (int i = 0; i < VPL; i++)
            v[i] = t[i];
    }
}

// WIN > 0: static window, blocks of passb_blk(WIN) rows (WIN, or a multiple of it for the small windows, so
// that the rolling prefetch stays about nine rows deep); the row leaving the window is the row that entered
// WIN rows earlier (registers: this block or the one before).  WIN == 0: generic window, blocks of 8 rows, the
// leaving row is re-read.  Windows 13 .. 17 hold 156 .. 204 row registers: two waves per SIMD, which is what
// the occupancy policy wants anyway (smhip_api.hip).  blockDim.x = 64 * (pixel columns per block), one wave per column.
// STORE = true: the same walk, but instead of PadCosts + WTA every aggregated row is written to
// the DSI S[y][x][D] (reference layout, row stride s_row_elems; D % VPL == 0): RawCosts + one
// AggrBox with 12 bytes per voxel instead of 20.  All levels are kept (the min filter that may
// follow reads the aggregated costs of padded positions too, StcAggregate.cpp:351-365).
// PAIR = true (VPL 4, 128-level slabs in the paired layout T[band][x / 2][ty][x & 1][128] that
// the float-lane pass A writes, smhip_fused_f.h PAIRED): a wave owns TWO pixel
// columns, 2c in lanes 0..31 and 2c + 1 in lanes 32..63, four levels per lane.  In that layout the 128 levels
// of the two columns are adjacent, so a row of the pair is one 1-KiB wave load as for a 256-level column (the
// unpaired 128-level slab is read with 512-byte loads, whose ceiling is 0.68 instead of 0.86 of 8 TB/s,
// scripts/membw8.hip), the two columns are independent (no exchange between the half-waves), and one DPP
// reduction over each half-wave serves two pixels.  W, H stay the image's; Dp must be 128.
// LRED = true (VPL 1 / 2, keys out): the minimum over the 64 lanes is not taken row by row with a DPP chain
// (seven dependent steps, a ballot, two readlanes: about 300 clocks of latency per row, which nothing hides
// when a frame gives each SIMD fewer than two waves -- 1920x1080x128: pass B 0.227 ms, without the reduction
// 0.170, scripts/bench_b.hip) but for a whole group of rows at once: every row's 64 ordered costs go to a
// wave-private LDS tile red[row][lane] (row stride 65 words), and when the group is complete lane r (and lane
// r + 32 for the upper half of the levels) scans row r -- 32 conflict-free reads, compare + two selects each, the
// scan order and the strict '<' keep the lowest level -- and so ends up holding exactly the key it has to store.
// The level inside the winning lane (VPL 2: one bit per row, collected in a register) comes by ds_bpermute.
// Groups are (32 / BLK) * BLK rows; dynamic LDS: the key buffers, then 32 x 65 words per wave.
template <int WIN, int VPL, int ABL = 0, bool STORE = false, bool PAIR = false, int AHEAD = 1, bool LRED = false>   // ABL: ablation switches of scripts/bench_b.hip (0 = product); AHEAD: blocks of rolling prefetch
__global__ __launch_bounds__(256) void fused_boxv_wta_vpl_kernel(
    const float *__restrict__ T, int tyshift, int W, int H, int Dp, int D, int win, float scale,
    int disp0, int dstep, float outside, int d_first, int x_per_xcd,
    long long *__restrict__ keys, size_t keys_chunk_stride, float *__restrict__ S = nullptr,
    size_t s_row_elems = 0, int kb_rows = 0,          // kb_rows: a multiple of GRP; dynamic LDS = waves * kb_rows * 8 (PAIR: * 16)
    int merge = 0)                                     // bit 0: keys[] holds the keys of other levels, store the minimum; bit 1: non-temporal key stores
{
    static_assert(!PAIR || (VPL == 4 && !STORE), "paired columns: four levels per lane, keys out");
    static_assert(!LRED || (VPL <= 2 && !STORE && !PAIR), "row groups reduced through LDS: one or two levels per lane, keys out");
    extern __shared__ __attribute__((aligned(16))) long long kbuf_all[];
    constexpr int BLK = passb_blk(WIN);
    constexpr int OFF = WIN ? BLK - WIN : 0;           // the block before: its last WIN rows are still needed
    constexpr int KLANES = PAIR ? 32 : 64;             // lanes that keep the keys of one pixel column
    constexpr int GRP = ((LRED ? 32 : KLANES) / BLK) * BLK;   // rows per key store (63 for WIN 9; PAIR, LRED: 27)
    constexpr int RSTRIDE = 65;                        // LRED: words per row of the reduction tile
    constexpr int CH = 64 * VPL;
    const int lane = threadIdx.x & 63;
    const int klane = PAIR ? (lane & 31) : lane;
    // XCD-aware column order: workgroups are dealt round-robin to the 8 XCDs, so give each
    // XCD a contiguous range of columns (PAIR: of column pairs).  Correctness does not depend on placement.
    const int wpb = blockDim.x >> 6;
    const int blk = blockIdx.x, wib = threadIdx.x >> 6;
    const int xin = (blk >> 3) * wpb + wib;
    const int xc = (blk & 7) * x_per_xcd + xin;        // column, or column pair
    const int chunk = blockIdx.y;
    if (xc >= (PAIR ? (W + 1) / 2 : W) || xin >= x_per_xcd)
        return;                                        // whole wave exits together
    const int x = PAIR ? 2 * xc + (lane >> 5) : xc;
    const int DpL = PAIR ? 256 : Dp;                   // floats per slab row of this wave's column (pair)
    const int Wc = PAIR ? (W + 1) / 2 : W;
    const int k0 = PAIR ? (lane & 31) * 4 : chunk * CH + lane * VPL;   // first level of this lane (k0 + VPL <= Dp)
    const size_t j = PAIR ? ((size_t)xc << tyshift) * 256 + lane * 4 : ((size_t)x << tyshift) * Dp + k0;
    const int w = WIN ? WIN : win;
    const int pr = w / 2, pl = pr - w;
    long long *kout = keys + (size_t)chunk * keys_chunk_stride + (size_t)min(x, W - 1) * H;
    long long *kbuf = kbuf_all + (size_t)wib * kb_rows * (PAIR ? 2 : 1) + (PAIR ? (size_t)(lane >> 5) * kb_rows : 0);
    int kb_first = 0;                                  // first row held in the LDS key buffer
    int *red = LRED ? (int *)(kbuf_all + (size_t)wpb * kb_rows) + wib * (32 * RSTRIDE) : nullptr;
    unsigned bip = 0;                                  // LRED, VPL 2: bit `slot` = which of the lane's two levels won row `slot`

    bool padded[VPL], live[VPL], repl[VPL];
    float replv[VPL];                                  // what PadCosts / the end of the slab put in the level's place
#pragma unroll
    for (int i = 0; i < VPL; i++) {
        int k = k0 + i;
        live[i] = k < D && x < W;                      // (PAIR: the odd column of the last pair may not exist)
        int xm = x + disp0 + (k < D ? k : D - 1) * dstep;
        padded[i] = (xm < 0) || (xm >= W);             // PadLine, StcRawCosts.cpp:451-456
        repl[i] = padded[i] || !live[i];
        replv[i] = live[i] ? outside : __builtin_inff();
    }

    // PadCosts replaces the aggregated cost of a padded (x, k) by `outside`, so the chains of
    // padded levels are never used: lanes whose levels are all padded (or beyond D) load
    // nothing (pass A does not write those elements either).
    bool need = false;
#pragma unroll
    for (int i = 0; i < VPL; i++)
        need = need || (live[i] && (STORE || !padded[i]));
    float *sout = STORE ? S + (size_t)x * D + (k0 < D ? k0 : 0) : nullptr;
    // The kernel is bound by instruction issue as much as by HBM (VALU + SALU + s_nop slots per row against
    // the bytes of a row), so the per-row instruction count matters: the row offset is one 32 x 32 -> 64 bit
    // product, PadCosts and the end of the slab are one select per level, the key select takes its scalars
    // without a copy.  (The loads are non-temporal; with the sc0 / sc1 bits added -- buffer loads -- the time is
    // the same, without nt pass B takes 0.383 instead of 0.334 ms.)  The branch around the load stays: a form without it (idle lanes read one shared
    // element) is 4 % faster when pass B runs alone (scripts/bench_b.hip) and 7 % slower behind pass A.
    const unsigned band_elems = ((unsigned)Wc << tyshift) * (unsigned)DpL;   // (< 2^32: the launch sees to it)
    const int tymask = (1 << tyshift) - 1;
    auto ld = [&](int row, float(&v)[VPL]) {
        if (need) {
            const int r = clampi(row, H);
            const size_t off = (size_t)(unsigned)(r >> tyshift) * band_elems + (unsigned)((r & tymask) * DpL);
            load_row<VPL>(T + off + j, v);
        } else {
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
            acc[i] += scale * v[i];
    }

    int key_lo = 0, key_hi = 0;
    // one row: running sums, PadCosts, first minimum of the lane's levels -> ordered cost and its index
    auto row_value = [&](const float(&od)[VPL], const float(&nw)[VPL], int &s, int &bi) {
        float best = 0;
        bi = 0;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            acc[i] = acc[i] + scale * (-od[i] + nw[i]);
            const float v = repl[i] ? replv[i] : acc[i];
            if (i == 0) {
                best = v;
            } else if (v < best) {                      // strict: lowest level wins inside the lane
                best = v;
                bi = i;
            }
        }
        s = ordered_bits(best);
    };
    // the wave minimum m is known: lowest lane holding it, its level, key into the slot lane
    auto row_key = [&](int s, int bi, int m, int slot) {
        unsigned long long eq = __builtin_amdgcn_ballot_w64(s == m);
        int wl = (int)__builtin_ctzll(eq);              // lowest lane with the minimum
        int d = d_first + chunk * CH + wl * VPL + __builtin_amdgcn_readlane(bi, wl);
        key_lo = (lane != slot) ? key_lo : d;           // (this way round the scalars need no copy to a VGPR)
        key_hi = (lane != slot) ? key_hi : m;
    };
    // PAIR: one minimum per half-wave (lanes 31 and 63 after five DPP steps), one key per pixel column
    auto row_key_pair = [&](int s, int bi, int slot) {
        int v = s;
        v = dpp_min<0x111, 0xF>(v);                     // row_shr:1
        v = dpp_min<0x112, 0xF>(v);                     // row_shr:2
        v = dpp_min<0x114, 0xF>(v);                     // row_shr:4
        v = dpp_min<0x118, 0xF>(v);                     // row_shr:8
        v = dpp_min<0x142, 0xA>(v);                     // row_bcast:15
        const int m0 = __builtin_amdgcn_readlane(v, 31), m1 = __builtin_amdgcn_readlane(v, 63);
        const unsigned e0 = (unsigned)__builtin_amdgcn_ballot_w64(s == m0);
        const unsigned e1 = (unsigned)(__builtin_amdgcn_ballot_w64(s == m1) >> 32);
        const int w0 = __builtin_ctz(e0), w1 = __builtin_ctz(e1);      // (never zero: each minimum sits in its half)
        const int d0 = d_first + w0 * 4 + __builtin_amdgcn_readlane(bi, w0);
        const int d1 = d_first + w1 * 4 + __builtin_amdgcn_readlane(bi, 32 + w1);
        const bool upper = lane >= 32;
        const int d = upper ? d1 : d0, m = upper ? m1 : m0;
        key_lo = (klane != slot) ? key_lo : d;
        key_hi = (klane != slot) ? key_hi : m;
    };
    // slot = lane that keeps the key of this row
    auto step = [&](const float(&od)[VPL], const float(&nw)[VPL], int slot, int y) {
        if constexpr (STORE) {
            typename vecf<VPL>::type o;
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                acc[i] = acc[i] + scale * (-od[i] + nw[i]);
                if constexpr (VPL == 1) o = acc[0]; else o[i] = acc[i];
            }
            if (need && y < H)                           // (non-temporal: the DSI is read by another kernel, later)
                __builtin_nontemporal_store(o, (typename vecf<VPL>::type *)(sout + (size_t)y * s_row_elems));
            return;
        }
        int s, bi;
        row_value(od, nw, s, bi);
        if (ABL & 1) {                                  // ablation: no wave reduction
            key_lo += bi;
            key_hi += s;
            return;
        }
        if constexpr (LRED) {
            red[slot * RSTRIDE + lane] = s;
            if constexpr (VPL > 1)
                bip |= (unsigned)bi << slot;
        } else if constexpr (PAIR) {
            row_key_pair(s, bi, slot);
        } else {
            int m = wave_min_i32(s);
            row_key(s, bi, m, slot);
        }
    };
    // (Two rows at once with their two DPP chains interleaved, so that one chain's step fills the other's
    // wait states, was measured: 100 fewer s_nop per 27 rows, 136 instead of 128 VGPRs, pass B 0.344 instead
    // of 0.332 ms behind pass A.)
    // after the block that starts at row yb: store the group's keys when it is complete
    auto flush = [&](int yb) {
        if constexpr (STORE)
            return;
        int g0 = (yb / GRP) * GRP;                      // first row of the group
        bool last = yb + BLK >= H;
        if (yb + BLK - g0 == GRP || last) {
            if constexpr (LRED) {                       // the group's rows: lane r takes row r, the halves of the wave 32 levels (lanes) each
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int r = lane & 31, half = lane & 32;
                const int *row = red + r * RSTRIDE + half;
                int m = row[0], wl = 0;
#pragma unroll
                for (int i = 1; i < 32; i++) {
                    const int v = row[i];
                    const bool lt = v < m;              // strict: the lowest lane (level) keeps a tie
                    m = lt ? v : m;
                    wl = lt ? i : wl;
                }
                wl += half;
                const int m1 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, m);
                const int w1 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, wl);
                if (m1 < m) {                           // (lanes 0 .. 31: the upper half wins only when smaller; lanes 32 .. 63 hold no key)
                    m = m1;
                    wl = w1;
                }
                int bi = 0;
                if constexpr (VPL > 1)
                    bi = ((unsigned)__builtin_amdgcn_ds_bpermute(wl << 2, (int)bip) >> r) & 1;
                key_lo = d_first + chunk * CH + wl * VPL + bi;
                key_hi = m;
                bip = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (ABL & 2) {                              // ablation: (almost) no key stores
                if (last && lane == 0)
                    kout[0] = make_key(key_hi, key_lo);
                return;
            }
            const bool col = x < W;                     // (PAIR: kout / kbuf belong to the lane's own column)
            if (kb_rows == 0) {                         // unbuffered (no dynamic LDS given)
                if (klane < GRP && g0 + klane < H && col) {
                    long long v = make_key(key_hi, key_lo);
                    if (merge & 1) {
                        const long long o = kout[g0 + klane];
                        v = o < v ? o : v;
                    }
                    kout[g0 + klane] = v;
                }
                return;
            }
            if (klane < GRP)
                kbuf[g0 - kb_first + klane] = make_key(key_hi, key_lo);
            if (g0 + GRP - kb_first == kb_rows || last) {
                const int n = col ? min(kb_rows, H - kb_first) : 0;
                if (merge & 1) {                        // (the old keys are read in the same long runs the new ones are written in)
                    for (int i = klane; i < n; i += KLANES) {
                        const long long o = kout[kb_first + i], v = kbuf[i];
                        if (merge & 2)
                            __builtin_nontemporal_store(o < v ? o : v, kout + kb_first + i);
                        else
                            kout[kb_first + i] = o < v ? o : v;
                    }
                } else if (PAIR || (merge & 2) || (ABL & 4)) {   // non-temporal key stores: 7680x4320x128 (paired) 2.78 -> 2.68 ms, 7680x4320x256 5.40 -> 5.29; 3840x2160x256 and 1920x1080x256 the same either way, so one column per wave takes them from 16 M pixels on (merge bit 1)
                    for (int i = klane; i < n; i += KLANES)
                        __builtin_nontemporal_store(kbuf[i], kout + kb_first + i);
                } else {
                    for (int i = klane; i < n; i += KLANES)
                        kout[kb_first + i] = kbuf[i];
                }
                kb_first += kb_rows;
            }
        }
    };

    if (WIN && AHEAD >= 2) {                           // prefetch AHEAD blocks ahead (AHEAD + 2 register blocks): VPL 1 and 2
        constexpr int NBUF = AHEAD + 2;
        float bb[NBUF][BLK][VPL];                      // (every index below is a constant after unrolling)
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            if (r < WIN)
                ld(r + pl, bb[0][OFF + r]);
#pragma unroll
            for (int a = 0; a < AHEAD; a++)
                ld(a * BLK + r + pr, bb[1 + a][r]);
        }
#pragma unroll
        for (int i = 0; i < VPL; i++)
            bb[0][OFF][i] = bb[1][0][i];
        auto block = [&](float(&prv)[BLK][VPL], float(&cur)[BLK][VPL], float(&nxt)[BLK][VPL], int yb) {
            const int slot0 = yb % GRP;
#pragma unroll
            for (int r = 0; r < BLK; r++) {
                ld(yb + AHEAD * BLK + r + pr, nxt[r]);
                if (r >= WIN)
                    step(cur[r - (r >= WIN ? WIN : 0)], cur[r], slot0 + r, yb + r);
                else
                    step(prv[OFF + r], cur[r], slot0 + r, yb + r);
            }
            flush(yb);
        };
        for (int yb = 0; yb < H; yb += NBUF * BLK) {
#pragma unroll
            for (int k = 0; k < NBUF; k++)
                if (yb + k * BLK < H)                  // block k: rows in bb[k] | bb[k + 1], loads into bb[k + AHEAD + 1]
                    block(bb[k], bb[(k + 1) % NBUF], bb[(k + AHEAD + 1) % NBUF], yb + k * BLK);
        }
    } else if (WIN) {
        float b0[BLK][VPL], b1[BLK][VPL], b2[BLK][VPL];
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            if (r < WIN)
                ld(r + pl, b0[OFF + r]);
            ld(r + pr, b1[r]);
        }
#pragma unroll
        for (int i = 0; i < VPL; i++)
            b0[OFF][i] = b1[0][i];   // y = 0 adds scale*(-a + a) = +0: the first value stays the window sum
        auto block = [&](float(&prv)[BLK][VPL], float(&cur)[BLK][VPL], float(&nxt)[BLK][VPL], int yb) {
            const int slot0 = yb % GRP;
#pragma unroll
            for (int r = 0; r < BLK; r++) {
                ld(yb + BLK + r + pr, nxt[r]);          // rolling prefetch, one block ahead
                if (r >= WIN)                           // the row that entered WIN rows ago
                    step(cur[r - (r >= WIN ? WIN : 0)], cur[r], slot0 + r, yb + r);
                else
                    step(prv[OFF + r], cur[r], slot0 + r, yb + r);
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
    } else {
        float nw[2][BLK][VPL], od[2][BLK][VPL];
#pragma unroll
        for (int r = 0; r < BLK; r++) {
            ld(r + pr, nw[0][r]);
            ld(r + pl, od[0][r]);
        }
#pragma unroll
        for (int i = 0; i < VPL; i++)
            od[0][0][i] = nw[0][0][i];
        for (int y0 = 0; y0 < H; y0 += 2 * BLK) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int yb = y0 + half * BLK;
                if (yb < H) {
                    const int slot0 = yb % GRP;
#pragma unroll
                    for (int r = 0; r < BLK; r++) {
                        ld(yb + BLK + r + pr, nw[half ^ 1][r]);
                        ld(yb + BLK + r + pl, od[half ^ 1][r]);
                        step(od[half][r], nw[half][r], slot0 + r, yb + r);
                    }
                    flush(yb);
                }
            }
        }
    }
}

// Keys (column-major, W columns of H keys) -> row-major disparity / best cost: 32x32 tiles
// through LDS so that both the key reads and the image writes are coalesced.
__global__ void keys_decode_t_kernel(const long long *__restrict__ keys, int W, int H,
                                     int32_t *__restrict__ disp, float *__restrict__ best)
{
    __shared__ long long tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8 threads
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int xx = x0 + ty + 8 * i, yy = y0 + tx;
        if (xx < W && yy < H)
            tile[ty + 8 * i][tx] = keys[(size_t)xx * H + yy];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int yy = y0 + ty + 8 * i, xx = x0 + tx;
        if (xx < W && yy < H) {
            long long k = tile[tx][ty + 8 * i];
            if (disp) disp[(size_t)yy * W + xx] = (int32_t)(k & 0xFFFFFFFFll);
            if (best) best[(size_t)yy * W + xx] = unordered_bits((int)(k >> 32));
        }
    }
}

// Row-major winners of a slab (local level + best cost) -> column-major keys with the global
// level d_first + d: the inverse tiling of keys_decode_t_kernel.
__global__ void keys_encode_t_kernel(const int32_t *__restrict__ disp, const float *__restrict__ best,
                                     int W, int H, int d_first, long long *__restrict__ keys)
{
    __shared__ long long tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8 threads
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int yy = y0 + ty + 8 * i, xx = x0 + tx;
        if (xx < W && yy < H) {
            size_t j = (size_t)yy * W + xx;
            tile[tx][ty + 8 * i] = make_key(ordered_bits(best[j]), d_first + disp[j]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int xx = x0 + ty + 8 * i, yy = y0 + tx;
        if (xx < W && yy < H)
            keys[(size_t)xx * H + yy] = tile[ty + 8 * i][tx];
    }
}

}  // namespace smk
