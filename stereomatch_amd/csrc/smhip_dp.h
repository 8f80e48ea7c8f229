// smhip_dp.h -- scan-line dynamic programming (OptDP, StcOptDP.cpp:110-332) on the resident DSI.
//
// Every cell (x, d) of a scan-line has three states -- M matched, L / R occluded (visible in one
// image only) -- with cumulative costs; seven transitions (:137-149) are tried in the order
// t = 0..6 with a strict '<' against the state's value, which starts at COST_MAX (:188-229):
//   M(x,d) <- t0: M(x-1,d) + c     t3: L(x-1,d-1) + (c + smooth)   t6: R(x-1,d) + (c + smooth)
//   L(x,d) <- t1: M(x-1,d) + oc    t2: L(x-1,d-1) + oc
//   R(x,d) <- t4: M(x,d+1) + oc    t5: R(x,d+1) + oc          (same column, filled from d = D-1 down)
// One wave per scan-line; lane l holds levels l, l+64, ... (register i = level l + 64 i).  M and L
// depend on the previous column only: all levels at once, the (d-1) neighbour by a one-lane DPP
// shift.  R is a serial chain down the column -- float additions of oc, so it cannot be
// re-associated -- run systolically: every lane repeats r = min(a, r[lane+1] + oc); after k steps
// the top k lanes hold their final value, after 64 the whole register does.  Transitions go to a
// byte volume (2 bits per state); lane-held columns + readlane walk the path back, then the
// occluded pixels are filled from the left (:78-107).  Bit-exact: the same float operations in
// the same order as the reference for every value that is kept.
#pragma once

#include "smhip_opt.h"
#include "smhip_dpp.h"

namespace smk {

constexpr int DP_OCC = -9999;                 // occLabel, :119
__device__ __forceinline__ float lane_value(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int NV>
__global__ __launch_bounds__(64) void opt_dp_kernel(const float *__restrict__ cost, int W, int H, int D,
                                                    size_t row_elems, const float *__restrict__ smooth, float oc,
                                                    uint8_t *__restrict__ trans, int32_t *__restrict__ disp)
{
    constexpr int B = (32 / NV) < 8 ? (32 / NV) : 8;          // columns per prefetch block
    const int y = blockIdx.x, lane = threadIdx.x;
    const float *c = cost + (size_t)y * row_elems;
    uint8_t *tr = trans + (size_t)y * W * D;
    const float *sm = smooth + (size_t)y * W;
    int32_t *out = disp + (size_t)y * W;
    const float inf = __builtin_inff(), cmax = SMK_COST_MAX;
    const float ocv = oc;                                     // in a VGPR for the DPP add
    auto load = [&](int x, float(&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; i++)
            v[i] = (x < W && lane + 64 * i < D) ? c[(size_t)x * D + lane + 64 * i] : inf;
    };
    float Mp[NV], Lp[NV], Rp[NV];                             // the previous column
    load(0, Mp);                                              // :162-170
#pragma unroll
    for (int i = 0; i < NV; i++) {
        Lp[i] = cmax;
        Rp[i] = cmax;
        if (lane + 64 * i < D)
            tr[lane + 64 * i] = 1;                            // M: t = 0; L, R: t = -1
    }
    float cb[2][B][NV];
#pragma unroll
    for (int j = 0; j < B; j++)
        load(1 + j, cb[0][j]);
    float sv = 0.0f;                                          // smooth[sbase + lane]
    int sbase = -64;
    auto step = [&](int x, const float(&cur)[NV]) {
        if (((x - 1) & ~63) != sbase) {
            sbase = (x - 1) & ~63;
            sv = (sbase + lane < W) ? sm[sbase + lane] : 0.0f;
        }
        const float s = lane_value(sv, (x - 1) & 63);         // m_smooth(x-1, y, 1), :174
        float Mn[NV], Ln[NV], a[NV];
        int code[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int d = lane + 64 * i;
            // L(x-1, d-1): lane 0 of register i takes lane 63 of register i-1
            const float edge = i > 0 ? lane_value(Lp[i > 0 ? i - 1 : 0], 63) : cmax;
            const float Ld = lane_from_below(Lp[i], edge);
            const float cs = cur[i] + s;                      // cinc of t = 3, 6 (:213-218)
            float m = cmax, l = cmax;
            int tm = 0, tl = 0;
            float c0 = Mp[i] + cur[i];
            if (c0 < m) { m = c0; tm = 1; }
            float c1 = Mp[i] + oc;
            if (c1 < l) { l = c1; tl = 1; }
            if (d > 0) {                                      // border0, :148
                float c2 = Ld + oc;
                if (c2 < l) { l = c2; tl = 2; }
                float c3 = Ld + cs;
                if (c3 < m) { m = c3; tm = 2; }
            }
            float c6 = Rp[i] + cs;
            if (c6 < m) { m = c6; tm = 3; }
            Mn[i] = m;
            Ln[i] = l;
            code[i] = tm | (tl << 2);
        }
        // a(d) = min(COST_MAX, M(x, d+1) + oc): what t = 4 offers level d; nothing above D-1
        float Rn[NV];
#pragma unroll
        for (int i = NV - 1; i >= 0; i--) {
            const float q = fminf(Mn[i] + oc, cmax);
            const float edge = i + 1 < NV ? lane_value(fminf(Mn[i + 1 < NV ? i + 1 : i] + oc, cmax), 0) : inf;
            const int d = lane + 64 * i;
            const float ab = lane_from_above(q, edge);
            a[i] = d < D - 1 ? ab : (d == D - 1 ? cmax : inf);
        }
        // the chain R(d) = min(a(d), R(d+1) + oc), top register first
        float carry = inf;                                    // R above the top level: never wins
#pragma unroll
        for (int i = NV - 1; i >= 0; i--) {
            if (64 * i < D) {                                 // wave-uniform
                float r = a[i];
                const int steps = min(64, D - 64 * i);        // lanes past D-1 hold +inf already
                // r = min(a, r[lane+1] + oc) as two instructions: the DPP add leaves lane 63 of t
                // alone (no source lane), where t keeps carry + oc; a VALU result needs two wait
                // states before a DPP read
                float t = carry + oc;
                for (int k = 0; k < steps; k += 4)            // extra steps past the fixed point change nothing
                    asm volatile("s_nop 1\n\t"
                                 "v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_min_f32 %1, %3, %0\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_min_f32 %1, %3, %0\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_min_f32 %1, %3, %0\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_min_f32 %1, %3, %0"
                                 : "+v"(t), "+v"(r)
                                 : "v"(ocv), "v"(a[i]));
                Rn[i] = r;
                carry = lane_value(r, 0);
            } else {
                Rn[i] = inf;
            }
        }
        // which transition produced R (:197-229): t = 4 if it beat COST_MAX, t = 5 if strictly below that
        carry = inf;
#pragma unroll
        for (int i = NV - 1; i >= 0; i--) {
            const int d = lane + 64 * i;
            const float c5 = lane_from_above(Rn[i], carry) + oc;
            int trn = a[i] < cmax ? 1 : 0;
            if (c5 < a[i])
                trn = 2;
            if (d >= D - 1)
                trn = 0;                                      // border1, :149
            code[i] |= trn << 4;
            carry = lane_value(Rn[i], 0);
            if (d < D)
                tr[(size_t)x * D + d] = (uint8_t)code[i];
            Mp[i] = Mn[i];
            Lp[i] = Ln[i];
            Rp[i] = d < D ? Rn[i] : cmax;
        }
    };
    for (int xb = 1; xb < W; xb += 2 * B) {
#pragma unroll
        for (int j = 0; j < B; j++)
            load(xb + B + j, cb[1][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb + j < W)
                step(xb + j, cb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            load(xb + 2 * B + j, cb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb + B + j < W)
                step(xb + B + j, cb[1][j]);
    }
    // cheapest matched cell of the last column (:256-274), first strict minimum below COST_MAX
    int bo = 0x7FFFFFFF, bi = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        int o = (lane + 64 * i < D) ? ordered_bits(Mp[i]) : 0x7FFFFFFF;
        if (o < bo) {
            bo = o;
            bi = lane + 64 * i;
        }
    }
    int mo, mi;
    wave_first_min(bo, bi, mo, mi);
    if (!(unordered_bits(mo) < cmax))
        mi = 0;
    // backtrack (:292-322); the transitions of this row were written by this wave
    __threadfence();
    auto tload = [&](int x, int(&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; i++)
            v[i] = (x >= 0 && lane + 64 * i < D) ? (int)tr[(size_t)x * D + lane + 64 * i] : 0;
    };
    int tb[2][B][NV];
#pragma unroll
    for (int j = 0; j < B; j++)
        tload(W - 1 - j, tb[0][j]);
    int d = mi, st = 0;                                       // wave-uniform
    bool dead = false;                                        // a path through COST_MAX ends the walk
    int dl = DP_OCC;                                          // lane (x & 63) keeps the label of pixel x
    auto back = [&](int x, const int(&t)[NV]) {
        if (lane == (x & 63))
            dl = DP_OCC;
        for (int guard = 0; guard <= D && !dead; guard++) {   // at most D-1 'up' moves in one column
            const int ln = d & 63, reg = d >> 6;
            int code = 0;
#pragma unroll
            for (int i = 0; i < NV; i++) {
                int v = __builtin_amdgcn_readlane(t[i], ln);
                code = (i == reg) ? v : code;
            }
            const int f = (code >> (2 * st)) & 3;             // 0: t = -1
            if (f == 0) {
                dead = true;
                break;
            }
            if (lane == (x & 63))
                dl = st == 0 ? d : DP_OCC;                    // :305-309
            if (st == 0) {                                    // t = 0, 3, 6
                if (f == 2) { d -= 1; st = 1; }
                else if (f == 3) st = 2;
                break;
            } else if (st == 1) {                             // t = 1, 2
                if (f == 2) d -= 1;
                else st = 0;
                break;
            } else {                                          // t = 4, 5: same column, one level up
                d += 1;
                if (f == 1) st = 0;
            }
        }
        if ((x & 63) == 0) {
            if (x + lane < W)
                out[x + lane] = dl;
            dl = DP_OCC;
        }
    };
    for (int xb = W - 1; xb >= 0; xb -= 2 * B) {
#pragma unroll
        for (int j = 0; j < B; j++)
            tload(xb - B - j, tb[1][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb - j >= 0)
                back(xb - j, tb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            tload(xb - 2 * B - j, tb[0][j]);
#pragma unroll
        for (int j = 0; j < B; j++)
            if (xb - B - j >= 0)
                back(xb - B - j, tb[1][j]);
    }
    // fill_occluded_pixels (:78-107), left to right: a hole takes the last matched value on its
    // left, the leading holes the first matched value of the line
    __threadfence();
    int first = 0;
    for (int b = 0; b < W; b += 64) {
        int v = b + lane < W ? out[b + lane] : DP_OCC;
        unsigned long long mask = __ballot(v != DP_OCC);
        if (mask) {
            first = __builtin_amdgcn_readlane(v, __ffsll((long long)mask) - 1);
            break;
        }
    }
    int last = first;
    for (int b = 0; b < W; b += 64) {
        int v = b + lane < W ? out[b + lane] : DP_OCC;
        int src = v != DP_OCC ? lane : -1;                    // nearest matched lane at or below this one
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            int o = __shfl_up(src, sft, 64);
            if (lane >= sft)
                src = max(src, o);
        }
        int f = __shfl(v, src < 0 ? 0 : src, 64);
        f = src < 0 ? last : f;
        if (b + lane < W)
            out[b + lane] = f;
        last = __builtin_amdgcn_readlane(f, 63);
    }
}

}  // namespace smk
