// smhip_iter_px.h -- the iterated binomial aggregation of exp4_aggr.txt with its neighbours fused in, "pixels on
// lanes" (the layout of smhip_multi_px.h):
//
//     source  RAW : raw matching costs formed in the tile from the staged image rows (MatchLine,
//                   StcRawCosts.cpp:130-218) -- the raw DSI is never written or read
//             DSI : a resident DSI
//     K <= 4 iterations of ConvolveSeparable 14641 (StcAggregate.cpp:280-289, Convolve.cpp:86-254)
//     sink    DSI : the aggregated DSI
//             KEYS: PadCosts (StcRawCosts.cpp:437-476) + OptWTA (StcOptimize.cpp:26-56) -> one 64-bit key per
//                   pixel -- the aggregated DSI is never written or read
//
// RAW -> KEYS is aggr_iter <= 4 with no DSI traffic at all; for more iterations RAW -> DSI and DSI -> KEYS bracket
// the plain multi-iteration passes (smhip_convolve_14641_run).
//
// Workgroup = 128 pixels (halo included) x LV levels, LV * 32 threads; thread = 4 consecutive pixels of one level;
// horizontal neighbours by DPP from the adjacent lanes, vertical window in registers (static ring slots), one
// barrier per row -- see smhip_multi_px.h.  With the KEYS sink a workgroup walks a segment of at most IPX_SEG rows
// through SEVERAL level tiles (outer loop) and keeps the best (cost, level) of every pixel of the segment in LDS
// (int32 ordered cost + uint16 level; tiles come in ascending level order, so a later tile wins only with a
// strictly lower cost -- OptWTA's strict '<'); one store per pixel at the end.  A row leaves the level pipeline
// through the transpose buffer `tout` (write before the row's barrier, read after it, one row late): for the DSI
// sink the reader stores 4 levels of a pixel (16 bytes), for the KEYS sink it takes the minimum of the pixel's LV
// levels: v_min_f32 over its four, a DPP butterfly over the pixel's LV / 4 lanes, lowest level equal to the minimum
// by a second integer butterfly (as min_wta_tile_kernel).
// RAW source: per row the reference pixels of the tile and the matching pixels its LV levels reach
// (128 + (LV - 1) |dstep|) are staged in LDS as the packed words of smhip_fused.h (fetched one row ahead into
// registers); the raw costs are non-negative integers, so the exact-product taps need no guard here.
#pragma once

#include "smhip_fused.h"
#include "smhip_minwta.h"
#include "smhip_multi_px.h"

namespace smk {

constexpr int IPX_SEG = 64;         // rows per segment with the KEYS sink (bounds the key arrays in LDS)
constexpr int IPX_KSTRIDE = 65;     // keys per pixel in LDS: odd, the pixels a wave updates hit different banks
enum { IPX_DSI = 0, IPX_RAW = 1, IPX_KEYS = 1 };
enum { IPX_BINOMIAL = 0, IPX_DIFFUSION = 1 };      // the iterated operator
// halo of a tile on either side, in pixels (a multiple of the four pixels of a thread) and in rows, per K iterations:
// the binomial reaches 2 pixels / rows per iteration, the diffusion stencil 1
constexpr int ipx_halo(int op, int K) { return op == IPX_BINOMIAL ? px_halo(K) : (K + 3) / 4 * 4; }
constexpr int ipx_txi(int op, int K) { return PX_TXT - 2 * ipx_halo(op, K); }
constexpr int ipx_rows(int op, int K) { return op == IPX_BINOMIAL ? 2 * K : K; }

struct IterPxArgs {
    const float *src;               // DSI source
    float *dst;                     // DSI sink
    size_t row_elems;
    int W, H, D;                    // width, rows, levels of the volume (slab) handled
    int seglen, force_slow;
    float lambda, acoef;            // diffusion: out = acoef * c + lambda * (((l + r) + u) + d), acoef = 1 - 4 lambda
    FusedA a;                       // RAW source
    PadWtaArgs pw;                  // KEYS sink
    int tiles_per_group;            // KEYS sink: level tiles a workgroup walks
    int direct;                     // KEYS sink, one tile per workgroup: the LDS key arrays are a ring of two chunks of 32
                                    // rows that goes to the tile's key image chunk by chunk -- segments of any length
    size_t part_stride;             // KEYS sink: elements between the key images of the level groups (blockIdx.y)
};

// packed image words of the RAW source
template <int NB> struct PxStage;
template <> struct PxStage<2> {     // one band, point reference interval: the words of RowLds<2>
    typedef uint32_t word;
    static __device__ __forceinline__ word fetch_r(const FusedA &a, size_t ro, int x) { return RowLds<2>::ref_word(a.rn[ro + x]); }
    static __device__ __forceinline__ word fetch_m(const FusedA &a, size_t ro, int x)
    {
        return RowLds<2>::match_word(a.mn[ro + x], a.mx[ro + x]);
    }
    template <bool SD> static __device__ __forceinline__ int gap(word r, word m)
    {
        const int e = (int)(__builtin_amdgcn_sad_u8(r, m, (uint32_t)-255) >> 1);
        return SD ? e * e : e;
    }
};
template <> struct PxStage<4> {     // four bands, intervals on both sides: the words of RowLds<4>
    typedef uint2 word;             // (lower ends, upper ends)
    static __device__ __forceinline__ word fetch_r(const FusedA &a, size_t ro, int x)
    {
        return make_uint2(*(const uint32_t *)(a.rn + ro + 4 * (size_t)x), *(const uint32_t *)(a.rx + ro + 4 * (size_t)x));
    }
    static __device__ __forceinline__ word fetch_m(const FusedA &a, size_t ro, int x)
    {
        return make_uint2(*(const uint32_t *)(a.mn + ro + 4 * (size_t)x), *(const uint32_t *)(a.mx + ro + 4 * (size_t)x));
    }
    template <bool SD> static __device__ __forceinline__ int gap(word r, word m)
    {
        int sum = 0;
#pragma unroll
        for (int s = 0; s < 32; s += 8) {
            const int e = max(max((int)((m.x >> s) & 0xFF) - (int)((r.y >> s) & 0xFF),
                                  (int)((r.x >> s) & 0xFF) - (int)((m.y >> s) & 0xFF)), 0);
            sum += SD ? e * e : e;
        }
        return sum;
    }
};

// minimum over each aligned group of QP lanes (QP = 4 or 8), delivered to all of them
template <int QP> __device__ __forceinline__ float segq_allmin_f32(float x)
{
    if constexpr (QP == 8) {
        return seg8_allmin_f32(x);
    } else {
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
            : "+v"(x));
        return x;
    }
}
template <int QP> __device__ __forceinline__ int segq_allmin_u32(int x)
{
    if constexpr (QP == 8) {
        return seg8_allmin_u32(x);
    } else {
        asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
            : "+v"(x));
        return x;
    }
}

template <int OP, int K, int LV, int SRC, int SINK, bool SD, int NB>
__global__ __launch_bounds__(LV * 32) __attribute__((amdgpu_waves_per_eu(4))) void iter_px_kernel(const IterPxArgs g)
{
    typedef typename fvec<4>::type vec;
    typedef typename PxStage<NB>::word word;
    constexpr int HL = ipx_halo(OP, K), TXI = ipx_txi(OP, K), QP = LV / 4, NT = LV * 32;
    constexpr int RL = ipx_rows(OP, K);                // rows between a level-0 row and the output row it completes
    constexpr int PER = OP == IPX_BINOMIAL ? 5 : 3;    // rows per level kept in registers = period of the ring slots
    constexpr bool BINOM = OP == IPX_BINOMIAL;
    constexpr bool RAW = SRC == IPX_RAW, KEYS = SINK == IPX_KEYS;
    __shared__ __attribute__((aligned(16))) float tin[RAW ? 1 : 2][RAW ? 1 : LV][PX_STRIDE];
    __shared__ __attribute__((aligned(16))) float tout[2][LV][PX_STRIDE];
    __shared__ int kcost[KEYS ? TXI * IPX_KSTRIDE : 1];             // best ordered cost [pixel][row of the segment]
    __shared__ unsigned short klev[KEYS ? TXI * IPX_KSTRIDE : 1];   // its level within the slab
    __shared__ int guard;
    extern __shared__ __attribute__((aligned(16))) uint8_t stage_raw[];     // RAW: 2 x (128 + NM) words
    const int W = g.W, H = g.H, D = g.D;
    const int t = threadIdx.x;
    const int X0 = (int)blockIdx.x * TXI - HL;
    const int y0 = (int)blockIdx.z * g.seglen, y1 = min(H, y0 + g.seglen);
    // compute layout: level l of the tile, pixels X0 + 4 q .. + 3
    const int l = t >> 5, q4 = t & 31;
    const int xb = X0 + 4 * q4;
    const bool left_edge = xb == 0, right_edge = xb + 3 == W - 1;
    // DSI layout: pixel lp of the tile, levels 4 lq .. + 3 of the tile
    const int lp = t / QP, lq = t % QP;
    // A compute thread keeps its four pixels in the order (0, 2, 1, 3): with packed f32 instructions working on aligned
    // register pairs, the pairs (px0, px2) and (px1, px3) make every operand pair of the horizontal taps either one of
    // the two pairs themselves or one of them shifted by a whole pixel pair -- four register moves per level instead of
    // nine, and no tap left unpaired (204 -> 17x instructions per row step at four binomial iterations).  The LDS
    // transposes do the permutation: column lpc holds pixel lp.
    const int lpc = (lp & ~3) | ((lp & 1) << 1) | ((lp >> 1) & 1);
    const bool has_out = lp >= HL && lp < PX_TXT - HL && X0 + lp < W;
    // RAW: staged words
    const int adstep = g.a.dstep < 0 ? -g.a.dstep : g.a.dstep;
    const int NM = PX_TXT + (LV - 1) * adstep, n_items = PX_TXT + NM;
    word *sr[2], *sm[2];
    sr[0] = (word *)stage_raw; sm[0] = sr[0] + PX_TXT; sr[1] = sm[0] + NM; sm[1] = sr[1] + PX_TXT;

    if (KEYS)
        for (int i = t; i < TXI * IPX_KSTRIDE; i += NT)
            kcost[i] = 0x7FFFFFFF;
    if (t == 0)
        guard = g.force_slow;
    const vec zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const int Rlo = y0 - RL, Rin = y1 - 1 + RL, Rhi = Rin + 1;

    // ---- per level tile ------------------------------------------------------------------------
    int k0 = 0;                     // first level of the tile
    size_t lcol = 0;                // DSI source / sink: this thread's column
    bool lact = false;              // DSI layout: its four levels exist
    bool repl[4] = {false, false, false, false};       // KEYS: the level's cost is replaced (PadCosts, beyond the slab, no output pixel)
    float replv[4] = {0.0f, 0.0f, 0.0f, 0.0f};         // ... by this
    int dl = 0, mslot = 0, M0 = 0;  // RAW: this thread's disparity, its first matching slot, first staged matching pixel
    vec ring[K][PER];               // binomial: level j's horizontally filtered rows n-4 .. n, slot = row mod 5;
                                    // diffusion: level j's rows n-2 .. n, slot = row mod 3
    vec nxt = zero, res = zero;
    word pre[2];                    // RAW: the words this thread stages for the next row
    bool slow = false;

    auto tap = [&](vec s0, vec s1, vec s2, vec s3, vec s4) -> vec {
        if (!RAW && slow)
            return vtap5<4>(s0, s1, s2, s3, s4);
        return vtap5_pow2<4>(s0, s1, s2, s3, s4);
    };
    auto htap = [&](vec v, auto fast, auto edge) -> vec {
        // v = (px0, px2, px1, px3); pixels 2, 3 of the lane below and 0, 1 of the lane above
        // (lanes 0 and 63, like the first and last lane of every level, hold halo pixels: what they receive is never used)
        float l2 = lane_from_below_any(v[1]), l3 = lane_from_below_any(v[3]);
        float r0 = lane_from_above_any(v[0]), r1 = lane_from_above_any(v[2]);
        if constexpr (decltype(edge)::value) {   // (only tiles that hold pixel 0 or W-1 pay for the selects)
            if (left_edge)
                l2 = l3 = v[0];
            if (right_edge)
                r0 = r1 = v[3];
        }
        // outputs (h0, h2) take the operand pairs (w[j], w[j + 2]), outputs (h1, h3) the pairs (w[j + 1], w[j + 3]) of
        // w = l2 l3 px0 px1 px2 px3 r0 r1: five pairs each, two of them v's own register pairs
        typedef typename fvec<2>::type f2;
        const f2 pa = {l2, v[0]}, pb = {l3, v[2]}, p0 = {v[0], v[1]}, p1 = {v[2], v[3]}, pc = {v[1], r0}, pd = {v[3], r1};
        f2 h02, h13;
        if (RAW || decltype(fast)::value || !slow) {
            h02 = vtap5_pow2<2>(pa, pb, p0, p1, pc);
            h13 = vtap5_pow2<2>(pb, p0, p1, pc, pd);
        } else {
            h02 = vtap5<2>(pa, pb, p0, p1, pc);
            h13 = vtap5<2>(pb, p0, p1, pc, pd);
        }
        const vec h = {h02[0], h02[1], h13[0], h13[1]};
        return h;
    };
    auto fetch_row = [&](int R) {   // RAW: words of row R into registers
        const size_t ro = (size_t)clampi(R, H) * g.a.pitch;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int it = t + i * NT;
            if (it < PX_TXT)
                pre[i] = PxStage<NB>::fetch_r(g.a, ro, clampi(X0 + it, W));
            else if (it < n_items)
                pre[i] = PxStage<NB>::fetch_m(g.a, ro, clampi(M0 + it - PX_TXT, W));
        }
    };
    // direct mode: rows y0 + rel0 .. + n - 1 (one chunk of the ring) -> this tile's key image, column-major runs
    auto flush = [&](int rel0, int n) {
        long long *keys = g.pw.keys + (size_t)blockIdx.y * g.part_stride;
        for (int i = t; i < TXI * 32; i += NT) {
            const int px = i >> 5, r = i & 31;
            const int x = X0 + HL + px;
            if (x < W && r < n) {
                const int slot = px * IPX_KSTRIDE + ((rel0 + r) & 63);
                keys[(size_t)x * H + y0 + rel0 + r] = make_key(kcost[slot], g.pw.d_first + (int)klev[slot]);
            }
        }
    };
    auto step = [&](auto rr, auto edge, const int R) {
        constexpr int RR = decltype(rr)::value;
        const int p = PER == 5 ? (RR & 1) ^ (((R - Rlo) / 5) & 1)      // parity of R - Rlo = PER * g + RR
                               : (RR & 1) ^ (((R - Rlo) / 3) & 1);
        if constexpr (RAW) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int it = t + i * NT;
                if (it < PX_TXT)
                    sr[p][it] = pre[i];
                else if (it < n_items)
                    sm[p][it - PX_TXT] = pre[i];
            }
            if (R + 1 <= Rin)
                fetch_row(R + 1);
        } else {
            if (BINOM && !slow && tap_guard_bad(nxt))
                guard = 1;
#pragma unroll
            for (int e = 0; e < 4; e++)
                tin[p][4 * lq + e][lpc] = nxt[e];
            if (lact && R + 1 <= Rin && R + 1 >= 0 && R + 1 < H)
                nxt = vload<4>(g.src + (size_t)(R + 1) * g.row_elems + lcol);
        }
        *(vec *)&tout[p][l][4 * q4] = res;      // the row finished at the step before
        __syncthreads();
        if (BINOM && !RAW && !slow)             // the guard of this row's inputs was set before the barrier
            slow = __builtin_amdgcn_readfirstlane(guard) != 0;
        const int qs = R - 1 - RL;              // row in tout
        if (qs >= y0 && qs < y1) {
            if constexpr (KEYS) {
                // PadCosts, then the lowest cost of the pixel's LV levels and the lowest level holding it
                float c[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    c[e] = repl[e] ? replv[e] : tout[p][4 * lq + e][lpc];   // (threads without an output pixel: +inf, never stored)
                const float bc = fmin_hw(fmin_hw(c[0], c[1]), fmin_hw(c[2], c[3]));
                const float mm = segq_allmin_f32<QP>(bc);
                const int lev = k0 + 4 * lq;
                int cand = 0x7FFFFFFF;
#pragma unroll
                for (int e = 3; e >= 0; e--)
                    cand = c[e] == mm ? lev + e : cand;
                cand = segq_allmin_u32<QP>(cand);
                if (g.direct) {                     // (wave-uniform) the chunk before is complete: out it goes
                    const int rel = qs - y0;
                    if (rel >= 32 && (rel & 31) == 0)
                        flush(rel - 32, 32);
                }
                if (lq == 0 && has_out) {
                    const int oc = ordered_bits(mm);
                    if (g.direct) {
                        const int idx = (lp - HL) * IPX_KSTRIDE + ((qs - y0) & 63);
                        kcost[idx] = oc;
                        klev[idx] = (unsigned short)cand;
                    } else {
                        const int idx = (lp - HL) * IPX_KSTRIDE + (qs - y0);
                        if (oc < kcost[idx]) {
                            kcost[idx] = oc;
                            klev[idx] = (unsigned short)cand;
                        }
                    }
                }
            } else {
                if (has_out && lact) {
                    vec o;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        o[e] = tout[p][4 * lq + e][lpc];
                    vstore<4>(g.dst + (size_t)qs * g.row_elems + lcol, o);
                }
            }
        }
        vec val;                                // level 0, row R
        if constexpr (RAW) {
            // MatchLine (StcRawCosts.cpp:152-198) for this thread's four reference pixels at its level
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = min(PxStage<NB>::template gap<SD>(sr[p][4 * q4 + e], sm[p][mslot + e]), g.a.cutoff);
                if constexpr (decltype(edge)::value)    // (interior tiles: every matching position of the tile's levels is inside the row)
                    val[(e & 2) >> 1 | (e & 1) << 1] = (unsigned)(xb + e + dl) >= (unsigned)W ? (float)g.a.outside : (float)c;   // pixel e -> its slot
                else
                    val[(e & 2) >> 1 | (e & 1) << 1] = (float)c;
            }
        } else {
            val = *(const vec *)&tin[p][l][4 * q4];
        }
        if constexpr (BINOM) {
        if ((RAW || !slow) && R - 2 * (K - 1) >= 1 && R <= H - 1) {
            // every level's row lies inside the image and the exact-product taps apply: one basic block
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int s = (RR + 10 * K - 2 * j) % 5;
                ring[j][s] = htap(val, icst<1>{}, edge);
                val = vtap5_pow2<4>(ring[j][(s + 1) % 5], ring[j][(s + 2) % 5], ring[j][(s + 3) % 5],
                                    ring[j][(s + 4) % 5], ring[j][s]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int n = R - 2 * j;        // row of level j arriving now
                const int s = (RR + 10 * K - 2 * j) % 5;    // its ring slot; rows n-4 .. n-1 sit in s+1 .. s+4 (mod 5)
                if (n <= H - 1) {               // rows below the image replicate the last row
                    ring[j][s] = htap(val, icst<0>{}, edge);
                    if (n == 0)                 // rows above the image replicate row 0
                        ring[j][(s + 3) % 5] = ring[j][(s + 4) % 5] = ring[j][s];
                } else {
                    ring[j][s] = ring[j][(s + 4) % 5];
                }
                val = tap(ring[j][(s + 1) % 5], ring[j][(s + 2) % 5], ring[j][(s + 3) % 5], ring[j][(s + 4) % 5],
                          ring[j][s]);          // level j+1, row n-2
            }
        }
        } else {
            // AggrDiffusion (StcDiffusion.cpp:75-120): o = a c + lambda (((l + r) + u) + d), a missing neighbour
            // replaced by c in its slot.  Level j holds its rows n-2, n-1, n (n = R - j arrives now) and emits
            // row n-1 of level j+1; rows outside the image are never looked at (u / d fall back to c).
            const vec la = g.lambda, ac = g.acoef;
            auto dlevels = [&](auto interior) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const int n = R - j;
                    const int s = (RR + 3 * K - j) % 3;         // slot of row n; n-1 in s+2, n-2 in s+1 (mod 3)
                    ring[j][s] = val;
                    const vec c = ring[j][(s + 2) % 3];   // (px0, px2, px1, px3)
                    float lf = lane_from_below_any(c[3]), rt = lane_from_above_any(c[0]);
                    if constexpr (decltype(edge)::value) {
                        if (left_edge)
                            lf = c[0];
                        if (right_edge)
                            rt = c[3];
                    }
                    // left / right neighbours of (px0, px2, px1, px3): the second pair of lv and the first of rv are c's own
                    const vec lv = {lf, c[2], c[0], c[1]}, rv = {c[2], c[3], c[1], rt};
                    vec u = ring[j][(s + 1) % 3], d = val;
                    if constexpr (!decltype(interior)::value) {
                        u = (n - 1 > 0) ? u : c;
                        d = (n - 1 < H - 1) ? d : c;
                    }
                    val = ac * c + la * (((lv + rv) + u) + d);  // level j+1, row n-1
                }
            };
            if (R - K > 0 && R <= H - 1)        // every level's centre row has both vertical neighbours
                dlevels(icst<1>{});
            else
                dlevels(icst<0>{});
        }
        res = val;                              // output row R - RL
    };

    const int n_tiles = KEYS ? g.tiles_per_group : 1;
    for (int tile = 0; tile < n_tiles; tile++) {
        k0 = (KEYS ? (int)blockIdx.y * g.tiles_per_group + tile : (int)blockIdx.y) * LV;
        if (k0 >= D)
            break;
        lact = k0 + 4 * lq < D;
        lcol = (size_t)clampi(X0 + lp, W) * D + (size_t)min(k0 + 4 * lq, D - 4);
        if constexpr (KEYS) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool padded = (unsigned)(X0 + lp + g.pw.disp0 + (k0 + 4 * lq + e) * g.pw.dstep) >= (unsigned)W;
                const bool dead = k0 + 4 * lq + e >= D || !has_out;     // beyond the slab, or a halo pixel
                repl[e] = padded || dead;
                replv[e] = dead ? __builtin_inff() : g.pw.outside;
            }
        }
        if constexpr (RAW) {
            const int d_lo = g.a.disp0 + k0 * g.a.dstep + min(0, (LV - 1) * g.a.dstep);
            M0 = X0 + d_lo;
            dl = g.a.disp0 + (k0 + l) * g.a.dstep;
            mslot = 4 * q4 + dl - d_lo;
        }
#pragma unroll
        for (int j = 0; j < K; j++)
#pragma unroll
            for (int i = 0; i < PER; i++)
                ring[j][i] = zero;
        nxt = zero;
        res = zero;
        if constexpr (RAW) {
            fetch_row(Rlo);
        } else {
            if (lact && Rlo >= 0 && Rlo < H)
                nxt = vload<4>(g.src + (size_t)Rlo * g.row_elems + lcol);
        }
        __syncthreads();            // the buffers of the level tile before are free (and kcost / guard are set)
        auto rows = [&](auto edge) {
            for (int Rb = Rlo; Rb <= Rhi; Rb += PER) {
                step(icst<0>{}, edge, Rb);
                if (Rb + 1 <= Rhi) step(icst<1>{}, edge, Rb + 1);
                if (Rb + 2 <= Rhi) step(icst<2>{}, edge, Rb + 2);
                if constexpr (PER == 5) {
                    if (Rb + 3 <= Rhi) step(icst<3>{}, edge, Rb + 3);
                    if (Rb + 4 <= Rhi) step(icst<4>{}, edge, Rb + 4);
                }
            }
        };
        // the tile holds an end of the image row, or (raw costs) a level of the tile matches beyond one (wave-uniform)
        if (X0 <= 0 || X0 + PX_TXT >= W || (RAW && (M0 < 0 || M0 + NM > W)))
            rows(icst<1>{});
        else
            rows(icst<0>{});
    }
    if (KEYS && g.direct) {
        __syncthreads();
        const int n = y1 - y0, last = ((n - 1) >> 5) << 5;          // the last chunk, complete or not
        flush(last, n - last);
    } else if constexpr (KEYS) {
        __syncthreads();
        const int rows = y1 - y0;
        long long *keys = g.pw.keys + (size_t)blockIdx.y * g.part_stride;
        for (int i = t; i < TXI * IPX_SEG; i += NT) {   // column-major keys: a pixel's rows are contiguous
            const int px = i / IPX_SEG, r = i % IPX_SEG;
            const int x = X0 + HL + px;
            if (x < W && r < rows)
                keys[(size_t)x * H + y0 + r] = make_key(kcost[px * IPX_KSTRIDE + r],
                                                        g.pw.d_first + (int)klev[px * IPX_KSTRIDE + r]);
        }
    }
}

}  // namespace smk
