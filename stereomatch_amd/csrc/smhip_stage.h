// smhip_stage.h -- tuned kernels of the stage operators on a resident DSI whose rows are
// 16-byte aligned (D % 4 == 0, row_elems % 4 == 0): every DSI access is a dwordx4 per lane,
// 1 KB per wave instruction.  The generic one-float-per-lane kernels of smhip_kernels.h stay
// as the fall-back for other shapes; both produce identical bits.
//
//   raw_costs_rows_kernel   StcRawCosts.cpp:130-218, 370-417   write 4 B/voxel
//   box_h_walk_kernel       BoxFilter.cpp:70-170, rows          read 4 + write 4 B/voxel
//   box_v_walk_kernel       BoxFilter.cpp:70-170, columns       read 4 + write 4 B/voxel
//   wta_vec_kernel          StcOptimize.cpp:26-56               read 4 B/voxel
#pragma once

#include "smhip_fused.h"
#include "smhip_stencil.h"

namespace smk {

// ---- raw costs ------------------------------------------------------------------------------
// One workgroup = one image row x a range of pixels.  The reference row and the (lower, upper)
// row of the matching image are staged in LDS as in pass A of the fused path; the workgroup
// then sweeps its part of the DSI row linearly, four consecutive levels of a pixel per lane.
template <bool SD, int NB>
__global__ __launch_bounds__(256) void raw_costs_rows_kernel(FusedA a, float *__restrict__ cost,
                                                             size_t row_elems, int xseg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_r[];
    const int W = a.W;
    RowLds<NB> L;
    L.init(lds_r, W);
    const int y = blockIdx.y;
    {
        const size_t ro = (size_t)y * a.pitch;
        for (int i = threadIdx.x; i < W; i += blockDim.x)
            L.stage(a, ro, i);
    }
    __syncthreads();
    const int D4 = a.D >> 2;
    const int x1 = min(W, ((int)blockIdx.x + 1) * xseg);
    int x = blockIdx.x * xseg + (int)threadIdx.x / D4;
    int k4 = (int)threadIdx.x % D4;
    const int dx = 256 / D4, dk = 256 % D4;
    const int cutoff = a.cutoff;
    const float outside = (float)a.outside;
    float *row = cost + (size_t)y * row_elems;
    while (x < x1) {
        int xm = x + a.disp0 + 4 * k4 * a.dstep;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++, xm += a.dstep) {
            bool out = (unsigned)xm >= (unsigned)W;
            int xc = (int)min((unsigned)xm, (unsigned)(W - 1));
            int e = min(L.template gap<SD>(x, xc), cutoff);
            v[j] = out ? outside : (float)e;
        }
        typename fvec<4>::type o = {v[0], v[1], v[2], v[3]};
        vstore_plain<4>(row + ((size_t)x * D4 + k4) * 4, o);
        x += dx;
        k4 += dk;
        if (k4 >= D4) {
            k4 -= D4;
            x++;
        }
    }
}

// ---- box filter -----------------------------------------------------------------------------
// Running sums exactly as boxFilterLines (BoxFilter.cpp:117-168): t[0] = sum of scale*s over the
// clamped window in ascending order, then t[x] = t[x-1] + scale*((-s[x+pl]) + s[x+pr]).  A
// thread owns V chains (V consecutive levels) and keeps the window in a register ring (WIN is
// static), so each element is loaded once.
template <int V, int WIN>
__device__ __forceinline__ void box_walk(const float *__restrict__ s, float *__restrict__ t, size_t stride,
                                         int len, float scale)
{
    typedef typename fvec<V>::type vec;
    constexpr int PR = WIN / 2, PL = PR - WIN;
    vec ring[WIN];
    vec acc = 0.0f;
#pragma unroll
    for (int j = 0; j < WIN; j++) {
        ring[j] = vload<V>(s + stride * clampi(PL + 1 + j, len));
        acc += scale * ring[j];
    }
    vstore<V>(t, acc);
    int xb = 1;
    for (; xb + WIN <= len; xb += WIN) {
        vec in[WIN];
#pragma unroll
        for (int j = 0; j < WIN; j++)
            in[j] = vload<V>(s + stride * min(xb + j + PR, len - 1));
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            acc = acc + scale * ((-ring[j]) + in[j]);
            ring[j] = in[j];
            vstore<V>(t + stride * (xb + j), acc);
        }
    }
#pragma unroll
    for (int j = 0; j < WIN; j++) {
        int x = xb + j;
        if (x < len) {
            vec in = vload<V>(s + stride * min(x + PR, len - 1));
            acc = acc + scale * ((-ring[j]) + in);
            ring[j] = in;
            vstore<V>(t + stride * x, acc);
        }
    }
}

// horizontal pass: thread = (row y, V levels), chains run along x at stride D
template <int V, int WIN>
__global__ __launch_bounds__(256) void box_h_walk_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                         int W, int H, int D, size_t row_elems, float scale)
{
    const int DV = D / V;
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (long)H * DV)
        return;
    const int y = (int)(c / DV), k = (int)(c - (long)y * DV) * V;
    const size_t base = (size_t)y * row_elems + k;
    box_walk<V, WIN>(src + base, dst + base, (size_t)D, W, scale);
}

// vertical pass: thread = V consecutive floats of a row, chains run along y at stride row_elems
template <int V, int WIN>
__global__ __launch_bounds__(256) void box_v_walk_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                         long WD, int H, size_t row_elems, float scale)
{
    const long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (j >= WD)
        return;
    box_walk<V, WIN>(src + j, dst + j, row_elems, H, scale);
}

// ---- winner-take-all ------------------------------------------------------------------------
// One wave per run of up to 64 consecutive pixels; per pixel every lane loads NV dwordx4
// (levels 256*j + 4*lane ..+3), keeps its own first minimum, and two DPP wave minima pick the
// smallest cost and then the lowest level that attains it (the strict '<' scan of
// StcOptimize.cpp:40-50).  Lane i keeps the result of pixel i; one coalesced store per run.
template <int NV>
__global__ __launch_bounds__(256) void wta_vec_kernel(const float *__restrict__ cost, int W, int H, int D,
                                                      size_t row_elems, int32_t *__restrict__ disp,
                                                      float *__restrict__ best, int run)
{
    typedef typename fvec<4>::type vec;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long n_pix = (long)W * H;
    const long p0 = wave * run;
    if (p0 >= n_pix)
        return;
    const int n = (int)min((long)run, n_pix - p0);
    int y = (int)(p0 / W), x = (int)(p0 - (long)y * W);
    const vec inf = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    auto load = [&](vec *v, int yy, int xx) {
        const float *c = cost + (size_t)yy * row_elems + (size_t)xx * D + 4 * lane;
#pragma unroll
        for (int j = 0; j < NV; j++)
            v[j] = (256 * j + 4 * lane < D) ? __builtin_nontemporal_load((const vec *)(c + 256 * j)) : inf;
    };
    vec cur[NV], nxt[NV];
    load(cur, y, x);
    int res_d = 0;
    float res_c = 0;
    for (int i = 0; i < n; i++) {
        int xn = x + 1, yn = y;
        if (xn == W) {
            xn = 0;
            yn++;
        }
        if (i + 1 < n)
            load(nxt, yn, xn);
        int s = 0x7FFFFFFF, d = 0x7FFFFFFF;
        float bv = 0;
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int dd = 256 * j + 4 * lane + e;
                int v = ordered_bits(cur[j][e]);
                if (dd < D && v < s) {
                    s = v;
                    d = dd;
                    bv = cur[j][e];
                }
            }
        const int smin = wave_min_i32(s);
        const int dmin = wave_min_i32(s == smin ? d : 0x7FFFFFFF);
        const float cmin = __int_as_float(
            __builtin_amdgcn_readlane(__float_as_int(bv), ((unsigned)dmin >> 2) & 63));
        if (lane == i) {
            res_d = dmin;
            res_c = cmin;
        }
#pragma unroll
        for (int j = 0; j < NV; j++)
            cur[j] = nxt[j];
        x = xn;
        y = yn;
    }
    if (lane < n) {
        if (disp) disp[p0 + lane] = res_d;
        if (best) best[p0 + lane] = res_c;
    }
}

// narrow DSI (D < 128): one thread per pixel, its D levels are one contiguous run of dwordx4
__global__ __launch_bounds__(256) void wta_thread_vec_kernel(const float *__restrict__ cost, int W, int H, int D,
                                                             size_t row_elems, int32_t *__restrict__ disp,
                                                             float *__restrict__ best)
{
    typedef typename fvec<4>::type vec;
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)W * H)
        return;
    int y = (int)(p / W), x = (int)(p - (long)y * W);
    const float *c = cost + (size_t)y * row_elems + (size_t)x * D;
    vec v = vload<4>(c);
    int bd = 0;
    float bc = v[0];
#pragma unroll
    for (int e = 1; e < 4; e++)
        if (v[e] < bc) { bc = v[e]; bd = e; }
    for (int d = 4; d < D; d += 4) {
        v = vload<4>(c + d);
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (v[e] < bc) { bc = v[e]; bd = d + e; }
    }
    if (disp) disp[p] = bd;
    if (best) best[p] = bc;
}

// D <= 128 (D % 4 == 0): SEG = 2..32 lanes per pixel, four levels per lane, so a wave reads its
// piece of the DSI row in one coalesced sweep; the first minimum of the segment by three integer
// DPP butterflies (ordered value, then lowest level holding it, then that lane's float).
template <int CTRL>
__device__ __forceinline__ int dpp_xmin(int v)
{
    return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false));
}
template <int SEG>
__device__ __forceinline__ int seg_min_all(int v)
{
    if (SEG >= 2)  v = dpp_xmin<0xB1>(v);                 // quad_perm [1,0,3,2]
    if (SEG >= 4)  v = dpp_xmin<0x4E>(v);                 // quad_perm [2,3,0,1]
    if (SEG >= 8)  v = dpp_xmin<0x141>(v);                // row_half_mirror
    if (SEG >= 16) v = dpp_xmin<0x140>(v);                // row_mirror
    if (SEG >= 32) v = min(v, __shfl_xor(v, 16, 64));
    return v;
}

template <int SEG, int PPT>
__global__ __launch_bounds__(256) void wta_seg_kernel(const float *__restrict__ cost, int W, int H, int D,
                                                      size_t row_elems, int32_t *__restrict__ disp,
                                                      float *__restrict__ best)
{
    // a block covers PPT * 256 / SEG consecutive pixels of row y; every thread issues its PPT loads
    // (one pixel each, 256 / SEG pixels apart) before the first reduction
    typedef typename fvec<4>::type vec;
    constexpr int PPB = 256 / SEG;
    const int y = blockIdx.y;
    const int sub = threadIdx.x % SEG;
    const int x0 = blockIdx.x * (PPT * PPB) + threadIdx.x / SEG;
    const float *row = cost + (size_t)y * row_elems + 4 * sub;
    vec v[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int x = x0 + k * PPB;
        v[k] = vec{0, 0, 0, 0};
        if (x < W && 4 * sub < D)
            v[k] = __builtin_nontemporal_load((const vec *)(row + (size_t)x * D));
    }
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int x = x0 + k * PPB;
        int s = 0x7FFFFFFF, d = 0x7FFFFFFF;
        float bv = 0;
        if (x < W && 4 * sub < D) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int o = ordered_bits(v[k][e]);
                if (o < s) {
                    s = o;
                    d = 4 * sub + e;
                    bv = v[k][e];
                }
            }
        }
        const int smin = seg_min_all<SEG>(s);
        const int dmin = seg_min_all<SEG>(s == smin ? d : 0x7FFFFFFF);
        const int bits = seg_min_all<SEG>(d == dmin ? __float_as_int(bv) : 0x7FFFFFFF);
        if (sub == 0 && x < W) {
            if (disp) disp[(size_t)y * W + x] = dmin;
            if (best) best[(size_t)y * W + x] = __int_as_float(bits);
        }
    }
}

}  // namespace smk
