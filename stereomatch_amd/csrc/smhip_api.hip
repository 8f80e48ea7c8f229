// smhip_api.hip -- the C-ABI of include/smhip.h: argument checks, workspace, launches.
#include "../../include/smhip.h"
#include "smhip_kernels.h"
#include "smhip_fused.h"
#include "smhip_fused_b.h"
#include "smhip_fused_chunk.h"
#include "smhip_fused_f_launch.h"
#include "smhip_stencil.h"
#include "smhip_minwta.h"
#include "smhip_subpix.h"
#include "smhip_stage.h"
#include "smhip_refine.h"
#include "smhip_multi.h"
#include "smhip_multi_px.h"
#include "smhip_iter_px.h"
#include "smhip_iter_px_launch.h"
#include "smhip_opt.h"
#include "smhip_bayes.h"
#include "smhip_dp.h"

#include <math.h>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(e_ == hipErrorOutOfMemory ? SMHIP_ENOMEM : SMHIP_EHIP,          \
                        "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum { WS_TMP = 0, WS_BT = 1, WS_KEYS = 2, WS_COUNT = 3 };

}  // namespace

// Scratch of the no-allocation entry point (smhip_fused_box_wta_noalloc): every device the library runs
// on holds its own copy of this array until the process ends.  Host code never uses the array's name as
// an address: smhip_init asks for the device's address once (hipGetSymbolAddress) and keeps it.
static __device__ __align__(256) unsigned char smhip_static_scratch[SMHIP_NOALLOC_SCRATCH_BYTES];
enum { SMHIP_MAX_DEVICES = 64 };
static unsigned char *g_scratch_ptr[SMHIP_MAX_DEVICES];   // per device, filled by smhip_init under g_scratch_mu
static std::mutex g_scratch_mu;

struct smhip_ctx {
    int device;
    hipStream_t stream;
    void *ws[WS_COUNT];
    size_t ws_bytes[WS_COUNT];
    hipEvent_t *ev;      // 3 events per profiled fused call
    int ev_cap, ev_used; // in calls
    double *bayes_w, *bayes_wh;          // AggrBayesian weights: device copy, host copy
    int bayes_n, bayes_band;             // levels they were computed for (0: none), half band width
    float bayes_sigma, bayes_eps;
    int opt[SMHIP_OPT_COUNT];            // smhip_set_option; -1 = the library's own policy
    const char *last_pass_a;             // kernel of the most recent fused call's pass A (smhip_last_pass_a_kernel)
    // no-allocation mode (inside smhip_fused_box_wta_noalloc): the workspace slots are carved out of this
    // device's static scratch array instead of hipMalloc'ed
    unsigned char *scratch;              // this device's address of smhip_static_scratch
    bool arena_on;
    size_t arena_used;
    void *arena_ptr[WS_COUNT];
    size_t arena_bytes[WS_COUNT];
};

namespace {

// grow-only device workspace slot
// Dynamic LDS the stage kernels ask for beyond their static arrays: nothing uses it, it bounds the workgroups
// resident per CU (fewer resident waves stream faster on this part).  Measured on the
// 3840x2160x512 DSI (scripts/bench_ops.py, profiles/r02_stage_occupancy.log), floors of 0 / 27 / 41 / 54 / 80 KB:
// binomial _to 7.08 / 6.63 / 6.78 / 8.14 / 13.2 ms, one diffusion iteration 7.53 / 7.34 / 7.04 / 7.33 / 7.33, membrane
// 7.88 / 7.83 / 7.48 / 7.54 / 7.52 (the same 4-8 % at 1920x1080x256 and x64); every other stage kernel is fastest
// without a floor (policy_kb 0; the box filter's two walks differ more from process to process than between floors).  SMHIP_OPT_STAGE_LDS_KB >= 0 overrides the floor of all of them.
static inline size_t stage_lds(const smhip_ctx *ctx, int policy_kb = 0)
{
    const int kb = ctx->opt[SMHIP_OPT_STAGE_LDS_KB] >= 0 ? ctx->opt[SMHIP_OPT_STAGE_LDS_KB] : policy_kb;
    return (size_t)kb * 1024;
}

int ws_get(smhip_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (ctx->arena_on) {                 // carve from the static scratch: no hipMalloc, no hipFree
        if (ctx->arena_bytes[slot] < bytes) {
            size_t at = (ctx->arena_used + 255) & ~(size_t)255;
            if (bytes > (size_t)SMHIP_NOALLOC_SCRATCH_BYTES || at > (size_t)SMHIP_NOALLOC_SCRATCH_BYTES - bytes)
                return fail(SMHIP_ENOMEM, "the no-allocation entry point needs %zu more bytes of scratch, %zu of its "
                            "%zu are taken: problem too large", bytes, at, (size_t)SMHIP_NOALLOC_SCRATCH_BYTES);
            ctx->arena_ptr[slot] = ctx->scratch + at;
            ctx->arena_bytes[slot] = bytes;
            ctx->arena_used = at + bytes;
        }
        *out = ctx->arena_ptr[slot];
        return SMHIP_OK;
    }
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        HIP_TRY(hipMalloc(&ctx->ws[slot], bytes));
        ctx->ws_bytes[slot] = bytes;
    }
    *out = ctx->ws[slot];
    return SMHIP_OK;
}

int check_ctx(smhip_ctx *ctx)
{
    if (!ctx)
        return fail(SMHIP_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return SMHIP_OK;
}

int check_dsi(const void *d_cost, int w, int h, int nb, size_t row_elems)
{
    if (!d_cost)
        return fail(SMHIP_EINVAL, "null DSI pointer");
    if (w <= 0 || h <= 0 || nb <= 0)
        return fail(SMHIP_EINVAL, "bad DSI shape %dx%dx%d", w, h, nb);
    if (row_elems < (size_t)w * nb)
        return fail(SMHIP_EINVAL, "row_elems %zu < width*bands %zu", row_elems, (size_t)w * nb);
    if ((size_t)w * nb > 0x7FFFFFFFul)
        return fail(SMHIP_EINVAL, "width*bands exceeds 2^31");
    if (h > SMHIP_MAX_ROWS)                            // rows ride on gridDim.y in several kernels
        return fail(SMHIP_EINVAL, "more than %d rows (%d) are not supported", SMHIP_MAX_ROWS, h);
    return SMHIP_OK;
}

int check_params(const smhip_match_params *p, const void *d_ref, const void *d_mtc, size_t pitch)
{
    if (!p || !d_ref || !d_mtc)
        return fail(SMHIP_EINVAL, "null argument");
    if (p->width <= 0 || p->height <= 0)
        return fail(SMHIP_EINVAL, "bad image shape %dx%d", p->width, p->height);
    if (p->height > SMHIP_MAX_ROWS)
        return fail(SMHIP_EINVAL, "more than %d rows (%d) are not supported", SMHIP_MAX_ROWS, p->height);
    if (p->n_bands != 1 && p->n_bands != 4)   // StereoIO.cpp:76-81
        return fail(SMHIP_EINVAL, "number of bands (%d) not supported by stereo matcher", p->n_bands);
    if (pitch < (size_t)p->width * p->n_bands)
        return fail(SMHIP_EINVAL, "image pitch %zu < width*bands", pitch);
    if (p->disp_n < 2)                        // StereoMatcher.cpp:158-159
        return fail(SMHIP_EINVAL, "too few disparity levels (%d)", p->disp_n);
    if (p->match_fn != 1 && p->match_fn != 2)
        return fail(SMHIP_EINVAL, "unknown match_fn %d", p->match_fn);
    if (p->disp_num < 1 || p->disp_den < 1)
        return fail(SMHIP_EINVAL, "bad disparity step %d/%d", p->disp_num, p->disp_den);
    if (((size_t)p->disp_den * (p->width - 1) + 1) * p->n_bands > 0x7FFFFFFFul)
        return fail(SMHIP_EINVAL, "up-sampled line (den=%d) exceeds 2^31 samples", p->disp_den);
    if (p->frame_diff_sign != 1 && p->frame_diff_sign != -1)
        return fail(SMHIP_EINVAL, "frame_diff_sign must be +1 or -1");
    return SMHIP_OK;
}

int cutoff_of(const smhip_match_params *p)
{
    return p->match_fn == 2 ? p->match_max * p->match_max : abs(p->match_max);
}

int outside_of(const smhip_match_params *p)
{
    int worst = p->n_bands * (p->match_fn == 2 ? 255 * 255 : 255);
    int c = cutoff_of(p);
    return worst < c ? worst : c;
}

// Fills RawArgs for levels [d_first, d_first+d_count); runs the BT pre-pass when needed.
int prepare_raw(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t pitch,
                const smhip_match_params *p, int d_first, int d_count, smk::RawArgs *a,
                bool bt_in_kernel = false)
{
    int W = p->width, H = p->height, nb = p->n_bands;
    a->pitch = pitch;
    a->W = W; a->H = H; a->nb = nb; a->D = d_count;
    a->dstep = -p->frame_diff_sign * p->disp_num;
    a->disp0 = -p->frame_diff_sign * (p->disp_den * p->disp_min) + d_first * a->dstep;
    a->sd = p->match_fn == 2;
    a->cutoff = cutoff_of(p);
    a->outside = outside_of(p);
    a->rmn = d_ref; a->rmx = nullptr;
    a->mmn = d_mtc; a->mmx = nullptr;
    if (p->match_interval && !bt_in_kernel) {
        size_t img = pitch * H;
        void *bt = nullptr;
        int rc = ws_get(ctx, WS_BT, img * 4, &bt);
        if (rc) return rc;
        uint8_t *b = (uint8_t *)bt;
        dim3 grid((W * nb + 255) / 256, H);
        smk::bt_minmax_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_mtc, b, b + img, pitch, W, H, nb);
        a->mmn = b; a->mmx = b + img;
        if (p->match_interpolated) {
            smk::bt_minmax_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_ref, b + 2 * img, b + 3 * img,
                                                                 pitch, W, H, nb);
            a->rmn = b + 2 * img; a->rmx = b + 3 * img;
        } else {
            a->rmx = d_ref;     // point interval [R, R] (StcRawCosts.cpp:410-411)
        }
        HIP_TRY(hipGetLastError());
    }
    return SMHIP_OK;
}

// floats per thread of the column-walk kernels: 4 when every row, the level count and the
// base pointers keep 16-byte alignment, else 1
static int walk_vec(int nb, size_t row_elems, const void *p0, const void *p1, const void *p2)
{
    uintptr_t bits = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2;
    return (nb % 4 == 0 && row_elems % 4 == 0 && (bits & 15) == 0) ? 4 : 1;
}

// disp_step = 1/den: up-samples both images into int16 line buffers (and their
// Birchfield-Tomasi intervals) in the WS_BT slot and fills SubpixArgs for all levels.
int prepare_subpix(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t pitch,
                   const smhip_match_params *p, smk::SubpixArgs *s)
{
    int W = p->width, H = p->height, nb = p->n_bands, den = p->disp_den;
    int n = den * (W - 1) + 1;
    size_t len = (size_t)n * nb * H;             // samples per buffer
    void *ws = nullptr;
    int rc = ws_get(ctx, WS_BT, len * 6 * sizeof(int16_t), &ws);
    if (rc) return rc;
    int16_t *buf0 = (int16_t *)ws, *buf1 = buf0 + len;
    int16_t *min1 = buf1 + len, *max1 = min1 + len, *min0 = max1 + len, *max0 = min0 + len;
    int cubic = p->match_interp == 3;            // eCubic, StereoParameters.h:44
    dim3 gi((W * nb + 255) / 256, H), gs((n * nb + 255) / 256, H);
    smk::interp_lines_kernel<<<gi, 256, stage_lds(ctx), ctx->stream>>>(d_ref, pitch, buf0, W, H, nb, den, cubic);
    smk::interp_lines_kernel<<<gi, 256, stage_lds(ctx), ctx->stream>>>(d_mtc, pitch, buf1, W, H, nb, den, cubic);
    s->rmn = buf0; s->rmx = nullptr;
    s->mmn = buf1; s->mmx = nullptr;
    if (p->match_interval) {
        smk::bt_minmax16_kernel<<<gs, 256, stage_lds(ctx), ctx->stream>>>(buf1, min1, max1, n, H, nb);
        s->mmn = min1; s->mmx = max1;
        if (p->match_interpolated) {
            smk::bt_minmax16_kernel<<<gs, 256, stage_lds(ctx), ctx->stream>>>(buf0, min0, max0, n, H, nb);
            s->rmn = min0; s->rmx = max0;
        } else {
            s->rmx = buf0;
        }
    }
    HIP_TRY(hipGetLastError());
    s->W = W; s->H = H; s->nb = nb; s->n = n; s->den = den;
    s->D = p->disp_n;
    s->dstep = -p->frame_diff_sign * p->disp_num;
    s->disp0 = -p->frame_diff_sign * (den * p->disp_min);
    s->sd = p->match_fn == 2;
    s->cutoff = cutoff_of(p);
    s->interpolated = p->match_interpolated != 0;
    s->outside = (float)outside_of(p);
    return SMHIP_OK;
}

}  // namespace

extern "C" {

const char *smhip_last_error(void) { return g_err; }

int smhip_init(int device, smhip_ctx **out)
{
    if (!out)
        return fail(SMHIP_EINVAL, "null context pointer");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n)
        return fail(SMHIP_EINVAL, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    smhip_ctx *ctx = (smhip_ctx *)calloc(1, sizeof(smhip_ctx));
    if (!ctx)
        return fail(SMHIP_ENOMEM, "out of host memory");
    ctx->device = device;
    ctx->stream = nullptr;
    if (device < SMHIP_MAX_DEVICES) {    // this device's copy of the static scratch (device is current)
        std::lock_guard<std::mutex> lock(g_scratch_mu);      // contexts may be created from several host threads
        if (!g_scratch_ptr[device]) {
            void *sp = nullptr;
            hipError_t e = hipGetSymbolAddress(&sp, HIP_SYMBOL(smhip_static_scratch));
            if (e != hipSuccess || !sp) {
                free(ctx);
                return fail(SMHIP_EHIP, "hipGetSymbolAddress(smhip_static_scratch): %s", hipGetErrorString(e));
            }
            g_scratch_ptr[device] = (unsigned char *)sp;
        }
        ctx->scratch = g_scratch_ptr[device];
    }
    for (int i = 0; i < SMHIP_OPT_COUNT; i++)
        ctx->opt[i] = -1;
    *out = ctx;
    return SMHIP_OK;
}

int smhip_set_option(smhip_ctx *ctx, int option, int value)
{
    int rc = check_ctx(ctx);            // also makes the context's device current for the synchronize below
    if (rc) return rc;
    if (option < 0 || option >= SMHIP_OPT_COUNT)
        return fail(SMHIP_EINVAL, "unknown option %d", option);
    if (option == SMHIP_OPT_TYSHIFT && value > 8)
        return fail(SMHIP_EINVAL, "slab band shift %d outside 0..8", value);
    if (option == SMHIP_OPT_STAGE_LDS_KB && value > 80)
        return fail(SMHIP_EINVAL, "stage kernels' LDS floor %d KiB above 80", value);
    if (option == SMHIP_OPT_A_LDS_KB && value > 80)
        return fail(SMHIP_EINVAL, "pass A LDS floor %d KiB above 80 (two workgroups per CU)", value);
    if (option == SMHIP_OPT_B_WAVES && value > 0 && (value < 6 || value > 16))
        return fail(SMHIP_EINVAL, "pass B waves per CU %d outside 6..16 (0: no bound)", value);
    if (option == SMHIP_OPT_MULTI_TXT && value >= 0 && value != 64 && value != 128)
        return fail(SMHIP_EINVAL, "binomial tile width %d: 64 or 128", value);
    // the slab layout and the kernels of calls already queued must not change under them
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->opt[option] = value < 0 ? -1 : value;
    return SMHIP_OK;
}

int smhip_shutdown(smhip_ctx *ctx)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < WS_COUNT; i++)
        if (ctx->ws[i])
            HIP_TRY(hipFree(ctx->ws[i]));
    if (ctx->bayes_w)
        HIP_TRY(hipFree(ctx->bayes_w));
    free(ctx->bayes_wh);
    if (ctx->ev) {                                   // profiling left open
        for (int i = 0; i < 3 * ctx->ev_cap; i++)
            (void)hipEventDestroy(ctx->ev[i]);
        free(ctx->ev);
    }
    free(ctx);
    return SMHIP_OK;
}

int smhip_set_stream(smhip_ctx *ctx, void *hip_stream)
{
    int rc = check_ctx(ctx);            // the context's device is current from here on (null stream = its null stream)
    if (rc) return rc;
    // The workspaces (slab, interval images, partial keys) belong to the context, not to a stream: work
    // queued on the old stream may still be reading them, so the old stream is drained first.
    if ((hipStream_t)hip_stream != ctx->stream)
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    return SMHIP_OK;
}

const char *smhip_last_pass_a_kernel(const smhip_ctx *ctx)
{
    return ctx && ctx->last_pass_a ? ctx->last_pass_a : "";
}

int smhip_sync(smhip_ctx *ctx)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SMHIP_OK;
}

size_t smhip_workspace_bytes(const smhip_ctx *ctx)
{
    size_t s = 0;
    if (ctx)
        for (int i = 0; i < WS_COUNT; i++)
            s += ctx->ws_bytes[i];
    return s;
}

int smhip_malloc(smhip_ctx *ctx, size_t bytes, void **d_ptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_ptr || bytes == 0)
        return fail(SMHIP_EINVAL, "bad allocation request");
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return SMHIP_OK;
}

int smhip_free(smhip_ctx *ctx, void *d_ptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_ptr));
    return SMHIP_OK;
}

int smhip_memcpy_h2d(smhip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SMHIP_OK;
}

int smhip_memcpy_d2h(smhip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SMHIP_OK;
}

int smhip_memcpy_d2d(smhip_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SMHIP_OK;
}

int smhip_memcpy2d_h2d(smhip_ctx *ctx, void *d_dst, size_t d_pitch, const void *h_src,
                       size_t h_pitch, size_t width_bytes, size_t rows)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(d_dst, d_pitch, h_src, h_pitch, width_bytes, rows,
                             hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SMHIP_OK;
}

int smhip_memcpy2d_d2h(smhip_ctx *ctx, void *h_dst, size_t h_pitch, const void *d_src,
                       size_t d_pitch, size_t width_bytes, size_t rows)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(h_dst, h_pitch, d_src, d_pitch, width_bytes, rows,
                             hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SMHIP_OK;
}

int smhip_preprocess_blur(smhip_ctx *ctx, uint8_t *d_img, size_t img_pitch, int w, int h, int nb, int n_iter)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_img || w < 1 || h < 1 || nb < 1 || img_pitch < (size_t)w * nb || n_iter < 0)
        return fail(SMHIP_EINVAL, "PreProcess: bad image arguments");
    if (n_iter == 0)
        return SMHIP_OK;
    void *tmp = nullptr;
    rc = ws_get(ctx, WS_BT, img_pitch * (size_t)h, &tmp);
    if (rc) return rc;
    dim3 grid((w * nb + 255) / 256, h);
    for (int i = 0; i < n_iter; i++) {
        smk::blur121_h_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_img, (uint8_t *)tmp, img_pitch, w, h, nb);
        smk::blur14641_v_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>((const uint8_t *)tmp, d_img, img_pitch, w, h, nb);
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

float smhip_match_outside(const smhip_match_params *p)
{
    return p ? (float)outside_of(p) : 0.0f;
}

int smhip_raw_costs(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                    size_t img_pitch, const smhip_match_params *p, float *d_cost,
                    size_t row_elems)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_params(p, d_ref, d_mtc, img_pitch);
    if (rc) return rc;
    rc = check_dsi(d_cost, p->width, p->height, p->disp_n, row_elems);
    if (rc) return rc;
    if ((size_t)p->width * p->disp_n > 0x7FFFFFFFul)
        return fail(SMHIP_EINVAL, "width*levels exceeds 2^31");
    dim3 grid((p->width * p->disp_n + 255) / 256, p->height);
    if (p->disp_den > 1) {
        smk::SubpixArgs s;
        rc = prepare_subpix(ctx, d_ref, d_mtc, img_pitch, p, &s);
        if (rc) return rc;
        smk::raw_costs_subpix_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(s, d_cost, row_elems);
        HIP_TRY(hipGetLastError());
        return SMHIP_OK;
    }
    smk::RawArgs a;
    rc = prepare_raw(ctx, d_ref, d_mtc, img_pitch, p, 0, p->disp_n, &a);
    if (rc) return rc;
    const int nbands = p->n_bands, W = p->width, H = p->height;
    const bool img_ok = nbands == 1 || (img_pitch % 4 == 0 && ((uintptr_t)d_ref & 3) == 0 &&
                                        ((uintptr_t)d_mtc & 3) == 0);
    const bool ref_point = !(p->match_interval && p->match_interpolated);   // a.rmx is d_ref or null: RowLds<2>
    const size_t lds = (size_t)W * (nbands == 1 ? (ref_point ? 8 : 4) : 16);
    if (img_ok && lds <= 150 * 1024 && walk_vec(p->disp_n, row_elems, d_cost, nullptr, nullptr) == 4) {
        smk::FusedA fa;
        fa.rn = a.rmn; fa.rx = a.rmx ? a.rmx : a.rmn;
        fa.mn = a.mmn; fa.mx = a.mmx ? a.mmx : a.mmn;
        fa.pitch = img_pitch; fa.W = W; fa.H = H; fa.D = p->disp_n;
        fa.disp0 = a.disp0; fa.dstep = a.dstep; fa.cutoff = a.cutoff; fa.outside = a.outside;
        fa.bt_inline = 0;
        int nseg = (4096 + H - 1) / H;                 // >= 4096 workgroups when the row allows it
        int xseg = (W + nseg - 1) / nseg;
        if (xseg < 32) xseg = W < 32 ? W : 32;
        dim3 gr((W + xseg - 1) / xseg, H);
        if (nbands == 1) {
            if (ref_point && fa.rn == fa.rx) {                    // one band, point reference: RowLds<2>
                if (a.sd) smk::raw_costs_rows_kernel<true, 2><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
                else      smk::raw_costs_rows_kernel<false, 2><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
            } else {
                if (a.sd) smk::raw_costs_rows_kernel<true, 1><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
                else      smk::raw_costs_rows_kernel<false, 1><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
            }
        } else {
            const bool pts = fa.rn == fa.rx && fa.mn == fa.mx;     // point intervals: RowLds<5>
            if (pts) {
                if (a.sd) smk::raw_costs_rows_kernel<true, 5><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
                else      smk::raw_costs_rows_kernel<false, 5><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
            } else if (fa.rn == fa.rx && !a.sd) {                 // point reference, interval matching, SAD: RowLds<6>
                smk::raw_costs_rows_kernel<false, 6><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
            } else {
                if (a.sd) smk::raw_costs_rows_kernel<true, 4><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
                else      smk::raw_costs_rows_kernel<false, 4><<<gr, 256, lds, ctx->stream>>>(fa, d_cost, row_elems, xseg);
            }
        }
    } else {
        smk::raw_costs_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(a, d_cost, row_elems);
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_box_filter(smhip_ctx *ctx, float *d_cost, int w, int h, int nb, size_t row_elems,
                     int x_width, int y_width, int average)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (x_width != y_width)   // BoxFilter.cpp:179-180
        return fail(SMHIP_EINVAL, "BoxFilter: xWidth != yWidth not implemented yet");
    int win = x_width;
    if (win < 1)
        return fail(SMHIP_EINVAL, "BoxFilter: bad window %d", win);
    // the reference's running sum reads src[x + pr] unchecked for x < -pl (BoxFilter.cpp:114-124),
    // i.e. up to index w - 1, and asserts against double wrapping (:37, :51): lines shorter than
    // the window are outside its domain
    if (win > w || win > h)
        return fail(SMHIP_EINVAL, "BoxFilter: window %d too large for %dx%d (double wrapping)", win, w, h);
    void *tmp = nullptr;
    rc = ws_get(ctx, WS_TMP, row_elems * (size_t)h * sizeof(float), &tmp);
    if (rc) return rc;
    float scale = (float)(average ? 1.0 / win : 1);   // BoxFilter.cpp:89
    long nh = (long)h * nb, nv = (long)w * nb;
    const int vec = walk_vec(nb, row_elems, d_cost, tmp, nullptr);
    // one chain per thread: measured faster than four chains per thread (16-byte accesses) at every
    // size -- 12.9 against 14.3 ms at 3840x2160x512, 0.44 against 0.78 ms at 1920x1080x64 -- the
    // coalescing is the same and there are four times as many threads to hide the serial walk
    const int vh = 1, vv = 1;
    const long th = nh / vh, tv = nv / vv;
    (void)vec;
#define LAUNCH_BOXH(V, WIN) smk::box_h_walk_kernel<V, WIN><<<(unsigned)((th + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>( \
        d_cost, (float *)tmp, w, h, nb, row_elems, scale)
#define LAUNCH_BOXV(V, WIN) smk::box_v_walk_kernel<V, WIN><<<(unsigned)((tv + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>( \
        (const float *)tmp, d_cost, nv, h, row_elems, scale)
#define LAUNCH_BOX(WIN) do { LAUNCH_BOXH(1, WIN); LAUNCH_BOXV(1, WIN); } while (0)
    switch (win) {               // static windows: register ring, every element loaded once
    case 3: LAUNCH_BOX(3); break;
    case 5: LAUNCH_BOX(5); break;
    case 7: LAUNCH_BOX(7); break;
    case 9: LAUNCH_BOX(9); break;
    case 11: LAUNCH_BOX(11); break;
    default:
        smk::box_lines_kernel<<<(unsigned)((nh + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            d_cost, (float *)tmp, nh, nb, row_elems, (size_t)nb, w, win, scale);
        smk::box_lines_kernel<<<(unsigned)((nv + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            (const float *)tmp, d_cost, nv, (int)nv, 0, row_elems, h, win, scale);
    }
#undef LAUNCH_BOX
#undef LAUNCH_BOXH
#undef LAUNCH_BOXV
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_box_filter_border(smhip_ctx *ctx, float *d_cost, int w, int h, int nb, size_t row_elems, int window,
                            int average, int border_x, int border_y)
{
    if (border_x == 1 && border_y == 1)
        return smhip_box_filter(ctx, d_cost, w, h, nb, row_elems, window, window, average);
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    // eBorderZero: the reference only treats it in the first sum of a line and throws "bad borderMode" from
    // borderIndex as soon as a window reaches over the border (BoxFilter.cpp:29-65, :117-124)
    if (border_x < 1 || border_x > 3 || border_y < 1 || border_y > 3)
        return fail(SMHIP_EINVAL, "BoxFilter: border modes are 1 (replicate), 2 (reflect), 3 (cyclic)");
    if (window < 1 || window > w || window > h)
        return fail(SMHIP_EINVAL, "BoxFilter: window %d too large for %dx%d (double wrapping)", window, w, h);
    void *tmp = nullptr;
    rc = ws_get(ctx, WS_TMP, row_elems * (size_t)h * sizeof(float), &tmp);
    if (rc) return rc;
    float scale = (float)(average ? 1.0 / window : 1);
    long nh = (long)h * nb, nv = (long)w * nb;
    smk::box_lines_border_kernel<<<(unsigned)((nh + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
        d_cost, (float *)tmp, nh, nb, row_elems, (size_t)nb, w, window, scale, border_x);
    smk::box_lines_border_kernel<<<(unsigned)((nv + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
        (const float *)tmp, d_cost, nv, (int)nv, 0, row_elems, h, window, scale, border_y);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

// Rows per workgroup of the tile kernels: 128 for large volumes (6 % re-computed halo rows), down
// to 16 for small ones so that a launch still has on the order of a thousand workgroups.
static int tile_seglen(const smhip_ctx *ctx, int w, int h, int nb)
{
    if (ctx->opt[SMHIP_OPT_TSEG] >= 8 && ctx->opt[SMHIP_OPT_TSEG] <= 4096)   // forced segment length
        return ctx->opt[SMHIP_OPT_TSEG];
    long tiles = (long)((w + 31) / 32) * ((nb + 31) / 32);
    int seg = 128;
    while (seg > 16 && tiles * ((h + seg - 1) / seg) < 2048)
        seg >>= 1;
    return seg;
}

static dim3 walk_grid(int w, int nb, int h, int vec, int seglen)
{
    long threads = ((long)w * nb + vec - 1) / vec;
    return dim3((unsigned)((threads + 255) / 256), (unsigned)((h + seglen - 1) / seglen));
}

int smhip_min_filter_to(smhip_ctx *ctx, const float *d_src, float *d_dst, int w, int h, int nb,
                        size_t row_elems, int x_width, int y_width)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_src, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_dst || d_dst == d_src)
        return fail(SMHIP_EINVAL, "MinFilter: source and destination must differ");
    if (x_width < 1 || y_width < 1)
        return fail(SMHIP_EINVAL, "MinFilter: bad window %dx%d", x_width, y_width);
    const int seglen = 128;
    int vec = walk_vec(nb, row_elems, d_src, d_dst, nullptr);
    dim3 grid = walk_grid(w, nb, h, vec, seglen);
#define LAUNCH_MIN(WIN)                                                                              \
    do {                                                                                             \
        if (vec == 4)                                                                                \
            smk::minfilter_walk_kernel<4, WIN><<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_src, d_dst, w, nb, h, \
                                                                            row_elems, seglen);     \
        else                                                                                         \
            smk::minfilter_walk_kernel<1, WIN><<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_src, d_dst, w, nb, h, \
                                                                            row_elems, seglen);     \
    } while (0)
    const int tseg = tile_seglen(ctx, w, h, nb);
    dim3 tgrid((w + 31) / 32, (nb + 31) / 32, (h + tseg - 1) / tseg);
#define LAUNCH_TMIN(WIN) smk::tile_walk_kernel<smk::MinOp<WIN>, 32><<<tgrid, 256, stage_lds(ctx), ctx->stream>>>( \
        d_src, d_dst, w, nb, h, row_elems, tseg)
    if (vec == 4 && x_width == y_width && x_width % 2 == 1 && x_width >= 3 && x_width <= 11 &&
        tgrid.y <= 65535 && tgrid.z <= 65535) {
        switch (x_width) {
        case 3: LAUNCH_TMIN(3); break;
        case 5: LAUNCH_TMIN(5); break;
        case 7: LAUNCH_TMIN(7); break;
        case 9: LAUNCH_TMIN(9); break;
        default: LAUNCH_TMIN(11); break;
        }
    } else if (x_width == y_width && x_width == 9) {
        LAUNCH_MIN(9);
    } else if (x_width == y_width && x_width == 5) {
        LAUNCH_MIN(5);
    } else if (x_width == y_width && x_width == 3) {
        LAUNCH_MIN(3);
    } else {
        // general window: horizontal pass into the workspace, vertical pass into the destination
        void *tmp = nullptr;
        rc = ws_get(ctx, WS_TMP, row_elems * (size_t)h * sizeof(float), &tmp);
        if (rc) return rc;
        if (tmp == d_src || tmp == d_dst)
            return fail(SMHIP_EINVAL, "MinFilter: buffers alias the context workspace");
        dim3 g2((w * nb + 255) / 256, h);
        smk::min_h_kernel<<<g2, 256, stage_lds(ctx), ctx->stream>>>(d_src, (float *)tmp, w, nb, row_elems, x_width);
        smk::min_v_kernel<<<g2, 256, stage_lds(ctx), ctx->stream>>>((const float *)tmp, d_dst, w * nb, h, row_elems,
                                                      y_width);
    }
#undef LAUNCH_MIN
#undef LAUNCH_TMIN
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_min_filter(smhip_ctx *ctx, float *d_cost, int w, int h, int nb, size_t row_elems,
                     int x_width, int y_width)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (x_width < 1 || y_width < 1)
        return fail(SMHIP_EINVAL, "MinFilter: bad window %dx%d", x_width, y_width);
    void *tmp = nullptr;
    size_t bytes = row_elems * (size_t)h * sizeof(float);
    const bool tiled = walk_vec(nb, row_elems, d_cost, nullptr, nullptr) == 4 && x_width % 2 == 1 &&
                       x_width >= 3 && x_width <= 11;
    if (x_width == y_width && (tiled || x_width == 9 || x_width == 5 || x_width == 3)) {
        rc = ws_get(ctx, WS_TMP, bytes, &tmp);
        if (rc) return rc;
        rc = smhip_min_filter_to(ctx, d_cost, (float *)tmp, w, h, nb, row_elems, x_width, y_width);
        if (rc) return rc;
        HIP_TRY(hipMemcpy2DAsync(d_cost, row_elems * sizeof(float), tmp, row_elems * sizeof(float),
                                     (size_t)w * nb * sizeof(float), h, hipMemcpyDeviceToDevice, ctx->stream));
        return SMHIP_OK;
    }
    rc = ws_get(ctx, WS_TMP, bytes, &tmp);
    if (rc) return rc;
    dim3 grid((w * nb + 255) / 256, h);
    smk::min_h_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_cost, (float *)tmp, w, nb, row_elems, x_width);
    smk::min_v_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>((const float *)tmp, d_cost, w * nb, h, row_elems,
                                                    y_width);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_convolve_14641_to(smhip_ctx *ctx, const float *d_src, float *d_dst, int w, int h, int nb,
                            size_t row_elems)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_src, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_dst || d_dst == d_src)
        return fail(SMHIP_EINVAL, "ConvolveSeparable: source and destination must differ");
    const int seglen = 64;
    int vec = walk_vec(nb, row_elems, d_src, d_dst, nullptr);
    dim3 grid = walk_grid(w, nb, h, vec, seglen);
    const int tseg = tile_seglen(ctx, w, h, nb);
    dim3 tgrid((w + 31) / 32, (nb + 31) / 32, (h + tseg - 1) / tseg);
    if (vec == 4 && tgrid.y <= 65535 && tgrid.z <= 65535)
        smk::tile_walk_kernel<smk::BinomOp, 32><<<tgrid, 256, stage_lds(ctx, 27), ctx->stream>>>(d_src, d_dst, w, nb, h,
                                                                               row_elems, tseg);
    else if (vec == 4)
        smk::binomial_walk_kernel<4><<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_src, d_dst, w, nb, h, row_elems, seglen);
    else
        smk::binomial_walk_kernel<1><<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_src, d_dst, w, nb, h, row_elems, seglen);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_convolve_14641(smhip_ctx *ctx, float *d_cost, int w, int h, int nb, size_t row_elems)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    void *tmp = nullptr;
    size_t bytes = row_elems * (size_t)h * sizeof(float);
    rc = ws_get(ctx, WS_TMP, bytes, &tmp);
    if (rc) return rc;
    rc = smhip_convolve_14641_to(ctx, d_cost, (float *)tmp, w, h, nb, row_elems);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(d_cost, row_elems * sizeof(float), tmp, row_elems * sizeof(float),
                                 (size_t)w * nb * sizeof(float), h, hipMemcpyDeviceToDevice, ctx->stream));
    return SMHIP_OK;
}

int smhip_diffusion_iter(smhip_ctx *ctx, const float *d_src, float *d_dst, const float *d_cost0,
                         int w, int h, int nb, size_t row_elems, float lambda, float beta)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_src, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_dst || d_dst == d_src)
        return fail(SMHIP_EINVAL, "diffusion needs distinct source and destination");
    if (beta > 0.0f && !d_cost0)
        return fail(SMHIP_EINVAL, "membrane model (beta > 0) needs the initial DSI");
    float a = 1 - lambda * (beta + 4);   // StcDiffusion.cpp:84
    float lb = lambda * beta;            // :132
    const int seglen = 64;
    const bool mem = beta > 0.0f;
    int vec = walk_vec(nb, row_elems, d_src, d_dst, mem ? d_cost0 : nullptr);
    dim3 grid = walk_grid(w, nb, h, vec, seglen);
#define LAUNCH_DIFF(V, M) smk::diffusion_walk_kernel<V, M><<<grid, 256, stage_lds(ctx, 41), ctx->stream>>>( \
        d_src, d_dst, d_cost0, w * nb, nb, h, row_elems, seglen, lambda, a, lb)
    if (vec == 4) { if (mem) LAUNCH_DIFF(4, true); else LAUNCH_DIFF(4, false); }
    else          { if (mem) LAUNCH_DIFF(1, true); else LAUNCH_DIFF(1, false); }
#undef LAUNCH_DIFF
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

// iterations per pass for n_iter iterations: ceil(n/4) passes of nearly equal depth
static int multi_depth(int remaining, int passes_left)
{
    return (remaining + passes_left - 1) / passes_left;
}
// Most iterations one pass may hold.  Temporal blocking trades VALU work (halo recomputation, and
// half-empty 32-level tiles when nb <= 16) for passes over the DSI, and a pass is cheap while the
// volume sits in the 256 MiB Infinity Cache.  Measured at 1920x1080 (scripts/run_depth.py): binomial
// x4 with 16 levels 0.37 ms at depth 4, 0.25 ms one by one; 32..64 levels depth 2 beats depth 4 by
// 3..11 %, 3840x2160x64 (2.1 GB) is back at depth 4 (1.68 against 1.87 ms); diffusion x10 with 16
// levels 0.39 -> 0.32 ms at depth 2, but 384x288x16 0.084 (depth 4) against 0.088 ms; from 32 levels
// on, and for the config-4 volume (17 GB), depth 4 wins.  SMHIP_OPT_MAXK forces a depth (tests).
static int multi_maxk(const smhip_ctx *ctx, bool binomial, int nb, size_t volume_bytes)
{
    if (ctx->opt[SMHIP_OPT_MAXK] >= 1 && ctx->opt[SMHIP_OPT_MAXK] <= 4)
        return ctx->opt[SMHIP_OPT_MAXK];
    if (binomial)
        return nb <= 16 ? 1 : (volume_bytes < ((size_t)1 << 30) ? 2 : 4);
    return (nb <= 16 && volume_bytes > ((size_t)32 << 20)) ? 2 : 4;   // tsukuba-size volumes: launches count, depth 4
}

static int launch_iter_dsi(smhip_ctx *ctx, int op, int K, int src, int nbk, bool sd, smk::IterPxArgs &g);

int smhip_diffusion_run(smhip_ctx *ctx, float *d_a, float *d_b, const float *d_cost0, int w, int h,
                        int nb, size_t row_elems, float lambda, float beta, int n_iter, int *result_in_b)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_a, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_b || d_b == d_a || !result_in_b)
        return fail(SMHIP_EINVAL, "diffusion needs two distinct buffers and a result flag");
    if (n_iter < 0)
        return fail(SMHIP_EINVAL, "negative iteration count");
    if (beta > 0.0f && !d_cost0)
        return fail(SMHIP_EINVAL, "membrane model (beta > 0) needs the initial DSI");
    const bool mem = beta > 0.0f;
    float *cur = d_a, *oth = d_b;
    *result_in_b = 0;
    const int tseg = tile_seglen(ctx, w, h, nb);
    const unsigned gy = (nb + 31) / 32, gz = (h + tseg - 1) / tseg;
    if (walk_vec(nb, row_elems, d_a, d_b, mem ? d_cost0 : nullptr) != 4 || gy > 65535 || gz > 65535) {
        for (int i = 0; i < n_iter; i++) {          // unaligned shapes: one iteration per pass
            rc = smhip_diffusion_iter(ctx, cur, oth, d_cost0, w, h, nb, row_elems, lambda, beta);
            if (rc) return rc;
            float *t = cur; cur = oth; oth = t;
            *result_in_b ^= 1;
        }
        return SMHIP_OK;
    }
    float a = 1 - lambda * (beta + 4);   // StcDiffusion.cpp:84
    float lb = lambda * beta;            // :132
    // Diffusion (beta = 0) on frames whose width is a multiple of 4: the pixels-on-lanes kernel of smhip_iter_px.h,
    // up to six iterations per pass, horizontal neighbours by DPP instead of LDS (3840x2160x512: 2.2 -> 1.2 ms per
    // iteration; profiles/r03_iter_fused.log).  SMHIP_OPT_MULTI_PX = 0 keeps the LDS form.
    if (!mem && w % 4 == 0 && (ctx->opt[SMHIP_OPT_MULTI_PX] > 0 ||
                               (ctx->opt[SMHIP_OPT_MULTI_PX] < 0 && row_elems * (size_t)h * sizeof(float) >= ((size_t)64 << 20)))) {
        smk::IterPxArgs g;
        memset(&g, 0, sizeof(g));
        g.W = w; g.H = h; g.D = nb; g.row_elems = row_elems;
        g.lambda = lambda; g.acoef = a;
        const int pk = smk::iter_px_max_k(smk::IPX_DIFFUSION);
        int px_passes = (n_iter + pk - 1) / pk;
        for (int left = n_iter; left > 0; px_passes--) {
            const int K = multi_depth(left, px_passes);
            g.src = cur; g.dst = oth;
            rc = launch_iter_dsi(ctx, smk::IPX_DIFFUSION, K, smk::IPX_DSI, 2, false, g);
            if (rc) return rc;
            left -= K;
            float *t = cur; cur = oth; oth = t;
            *result_in_b ^= 1;
        }
        return SMHIP_OK;
    }
    const int maxk = multi_maxk(ctx, false, nb, row_elems * (size_t)h * sizeof(float));
    int passes = (n_iter + maxk - 1) / maxk;
    for (int left = n_iter; left > 0; passes--) {
        int K = multi_depth(left, passes);
        dim3 grid((w + (smk::MULTI_TXT - 2 * K) - 1) / (smk::MULTI_TXT - 2 * K), gy, gz);
#define LAUNCH_DM(KK)                                                                              \
        do {                                                                                       \
            if (mem) smk::diffusion_multi_kernel<KK, true><<<grid, smk::MULTI_TXT * 8, stage_lds(ctx), ctx->stream>>>(  \
                cur, oth, d_cost0, w, nb, h, row_elems, tseg, lambda, a, lb);                      \
            else smk::diffusion_multi_kernel<KK, false><<<grid, smk::MULTI_TXT * 8, stage_lds(ctx), ctx->stream>>>(     \
                cur, oth, d_cost0, w, nb, h, row_elems, tseg, lambda, a, lb);                      \
        } while (0)
        switch (K) {
        case 1: LAUNCH_DM(1); break;
        case 2: LAUNCH_DM(2); break;
        case 3: LAUNCH_DM(3); break;
        default: LAUNCH_DM(4); break;
        }
#undef LAUNCH_DM
        left -= K;
        float *t = cur; cur = oth; oth = t;
        *result_in_b ^= 1;
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_convolve_14641_run(smhip_ctx *ctx, float *d_a, float *d_b, int w, int h, int nb,
                             size_t row_elems, int n_iter, int *result_in_b)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_a, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_b || d_b == d_a || !result_in_b)
        return fail(SMHIP_EINVAL, "ConvolveSeparable: needs two distinct buffers and a result flag");
    if (n_iter < 0)
        return fail(SMHIP_EINVAL, "negative iteration count");
    float *cur = d_a, *oth = d_b;
    *result_in_b = 0;
    const int tseg0 = tile_seglen(ctx, w, h, nb);
    const unsigned gy = (nb + 31) / 32;
    if (walk_vec(nb, row_elems, d_a, d_b, nullptr) != 4 || gy > 65535 || (h + tseg0 - 1) / tseg0 > 65535) {
        for (int i = 0; i < n_iter; i++) {
            rc = smhip_convolve_14641_to(ctx, cur, oth, w, h, nb, row_elems);
            if (rc) return rc;
            float *t = cur; cur = oth; oth = t;
            *result_in_b ^= 1;
        }
        return SMHIP_OK;
    }
    const int maxk = multi_maxk(ctx, true, nb, row_elems * (size_t)h * sizeof(float));
    int passes = (n_iter + maxk - 1) / maxk;
    for (int left = n_iter; left > 0; passes--) {
        int K = multi_depth(left, passes);
        if (K == 1) {                               // the one-iteration tile kernel has the smaller halo
            rc = smhip_convolve_14641_to(ctx, cur, oth, w, h, nb, row_elems);
            if (rc) return rc;
        } else {
            // Tile shape (3840x2160x512, K = 4, scripts/bench_binom.py -> profiles/r03_binom_variants.log): 64 pixels x
            // 128 rows 10.7 ms, 128 pixels (1024 threads, one workgroup per CU) 9.6, with 256-row segments 9.2 --
            // the halo's share of the arithmetic falls from 1.50 to 1.21.  Small volumes keep the 64-pixel tile
            // (more, smaller workgroups).  Since the pixels-on-lanes kernel keeps its pixels in the order (0, 2, 1, 3)
            // (smhip_iter_px.h) it beats the LDS form at every size measured -- 3840x2160x512 x4 9.4 -> 8.2 ms, x8 19.3 ->
            // 16.9, 1920x1080x256 1.73 -> 1.41, x64 0.55 -> 0.40, 640x480x64 0.120 -> 0.109 -- and is the library's choice
            // for widths that are a multiple of 4; SMHIP_OPT_MULTI_PX = 0 keeps the LDS form.
            const size_t vol = row_elems * (size_t)h * sizeof(float);
            const bool big = vol >= ((size_t)1 << 30) && w >= 1024;
            const int txt = ctx->opt[SMHIP_OPT_MULTI_TXT] > 0 ? ctx->opt[SMHIP_OPT_MULTI_TXT] : (big ? 128 : 64);
            const int slow = ctx->opt[SMHIP_OPT_TAP_SLOW] == 1;                 // tests: the ten-operation tap sum
            const int pxo = ctx->opt[SMHIP_OPT_MULTI_PX];
            int tseg = tseg0;
            if (big && ctx->opt[SMHIP_OPT_TSEG] < 0 && h >= 512)
                tseg = 256;
            const unsigned gz = (h + tseg - 1) / tseg;
            if (w % 4 == 0 && pxo != 0) {       // pixels on lanes (smhip_iter_px.h, resident DSI in and out)
                smk::IterPxArgs g;
                memset(&g, 0, sizeof(g));
                g.W = w; g.H = h; g.D = nb; g.row_elems = row_elems;
                g.force_slow = slow;
                g.src = cur; g.dst = oth;
                rc = launch_iter_dsi(ctx, smk::IPX_BINOMIAL, K, smk::IPX_DSI, 2, false, g);
                if (rc) return rc;
                left -= K;
                float *t = cur; cur = oth; oth = t;
                *result_in_b ^= 1;
                continue;
            }
            dim3 grid((w + (txt - 4 * K) - 1) / (txt - 4 * K), gy, gz);
#define LAUNCH_BM(KK, TT) smk::binomial_multi_kernel<KK, TT><<<grid, TT * 8, stage_lds(ctx), ctx->stream>>>( \
                cur, oth, w, nb, h, row_elems, tseg, slow)
            if (txt == 128) {
                switch (K) {
                case 2: LAUNCH_BM(2, 128); break;
                case 3: LAUNCH_BM(3, 128); break;
                default: LAUNCH_BM(4, 128); break;
                }
            } else {
                switch (K) {
                case 2: LAUNCH_BM(2, 64); break;
                case 3: LAUNCH_BM(3, 64); break;
                default: LAUNCH_BM(4, 64); break;
                }
            }
#undef LAUNCH_BM
        }
        left -= K;
        float *t = cur; cur = oth; oth = t;
        *result_in_b ^= 1;
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_pad_costs(smhip_ctx *ctx, float *d_cost, size_t row_elems, const smhip_match_params *p)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!p)
        return fail(SMHIP_EINVAL, "null parameters");
    rc = check_dsi(d_cost, p->width, p->height, p->disp_n, row_elems);
    if (rc) return rc;
    if (p->disp_num < 1 || p->disp_den < 1)
        return fail(SMHIP_EINVAL, "bad disparity step %d/%d", p->disp_num, p->disp_den);
    if (p->disp_den > 1) {
        if ((size_t)p->width * p->disp_n > 0x7FFFFFFFul)
            return fail(SMHIP_EINVAL, "width*levels exceeds 2^31");
        int den = p->disp_den, n = den * (p->width - 1) + 1;
        dim3 grid((p->width * p->disp_n + 255) / 256, p->height);
        smk::pad_costs_subpix_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(
            d_cost, row_elems, p->width, p->disp_n, n, den, -p->frame_diff_sign * (den * p->disp_min),
            -p->frame_diff_sign * p->disp_num, smhip_match_outside(p));
        HIP_TRY(hipGetLastError());
        return SMHIP_OK;
    }
    int dstep = -p->frame_diff_sign * p->disp_num;
    int disp0 = -p->frame_diff_sign * p->disp_min;
    int dlast = disp0 + (p->disp_n - 1) * dstep;
    int maxpad = abs(disp0) > abs(dlast) ? abs(disp0) : abs(dlast);
    if (maxpad > p->width) maxpad = p->width;
    if (maxpad == 0)
        return SMHIP_OK;
    dim3 grid((p->disp_n * maxpad + 255) / 256, p->height);
    smk::pad_costs_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_cost, row_elems, p->width, p->disp_n, disp0,
                                                        dstep, maxpad, smhip_match_outside(p));
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_wta(smhip_ctx *ctx, const float *d_cost, int w, int h, int nb, size_t row_elems,
              int32_t *d_disp, float *d_best)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_disp && !d_best)
        return fail(SMHIP_EINVAL, "no output requested");
    long n_pix = (long)w * h;
    if (nb <= 1024 && walk_vec(nb, row_elems, d_cost, nullptr, nullptr) == 4) {
        if (nb > 128) {
            const int run = 16;                        // pixels per wave
            long waves = (n_pix + run - 1) / run;
            unsigned blocks = (unsigned)((waves + 3) / 4);
            int nv = (nb + 255) / 256;
            if (nv == 1)      smk::wta_vec_kernel<1><<<blocks, 256, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, d_disp, d_best, run);
            else if (nv == 2) smk::wta_vec_kernel<2><<<blocks, 256, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, d_disp, d_best, run);
            else if (nv == 3) smk::wta_vec_kernel<3><<<blocks, 256, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, d_disp, d_best, run);
            else              smk::wta_vec_kernel<4><<<blocks, 256, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, d_disp, d_best, run);
        } else if (nb >= 8) {
            int seg = 2;                               // lanes per pixel: the power of two >= nb / 4
            while (4 * seg < nb) seg *= 2;
            const int ppt = 4;                         // pixels per thread (8 measured 2.5x slower)
            const int per_block = ppt * 256 / seg;
            dim3 grid((unsigned)((w + per_block - 1) / per_block), h);
#define LAUNCH_WS(S) smk::wta_seg_kernel<S, ppt><<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, d_disp, d_best)
            switch (seg) {
            case 2: LAUNCH_WS(2); break;
            case 4: LAUNCH_WS(4); break;
            case 8: LAUNCH_WS(8); break;
            case 16: LAUNCH_WS(16); break;
            default: LAUNCH_WS(32); break;
            }
#undef LAUNCH_WS
        } else {
            smk::wta_thread_vec_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
                d_cost, w, h, nb, row_elems, d_disp, d_best);
        }
    } else if (nb >= 64) {
        long blocks = (n_pix + 3) / 4;
        if (blocks > 256 * 32) blocks = 256 * 32;
        smk::wta_wave_kernel<<<(unsigned)blocks, 256, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems,
                                                                        d_disp, d_best);
    } else {
        smk::wta_thread_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            d_cost, w, h, nb, row_elems, d_disp, d_best);
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_subpixel_fit(smhip_ctx *ctx, const float *d_src, float *d_dst, float *d_min, float *d_cert,
                       int w, int h, int nb, size_t row_elems, float match_outside)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_src, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_dst || !d_min || !d_cert || d_dst == d_src)
        return fail(SMHIP_EINVAL, "AggrSubPixelFit: needs distinct source, destination, min and cert buffers");
    dim3 grid((w * nb + 255) / 256, h);
    smk::subpixel_fit_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_src, d_dst, d_min, d_cert, w, nb, row_elems,
                                                           match_outside);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_collapse(smhip_ctx *ctx, const float *d_cost1, const float *d_min1, int w, int h, int nb1,
                   size_t row_elems1, int df, int nb2, size_t row_elems2, float disp_step,
                   float *d_cost2, float *d_min2)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost1, w, h, nb1, row_elems1);
    if (rc) return rc;
    rc = check_dsi(d_cost2, w, h, nb2, row_elems2);
    if (rc) return rc;
    if (!d_min2)
        return fail(SMHIP_EINVAL, "AggrCollapse: null output");
    if (df < 2 || (nb1 - 1) % df != 0 || nb2 != (nb1 - 1) / df + 1)   // StcAggregate.cpp:139-146
        return fail(SMHIP_EINVAL, "AggrCollapse: %d levels at step 1/%d do not collapse to %d", nb1, df, nb2);
    dim3 grid((w * nb2 + 255) / 256, h);
    smk::collapse_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(d_cost1, d_min1, w, nb1, row_elems1, df, nb2,
                                                       row_elems2, disp_step, d_cost2, d_min2);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_refine(smhip_ctx *ctx, const int32_t *d_disp, const float *d_cost, const float *d_min, int w,
                 int h, int nb, size_t row_elems, float disp_step, float disp_min, int mode,
                 float *d_fdisp)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_disp || !d_fdisp || w < 1 || h < 1 || nb < 1)
        return fail(SMHIP_EINVAL, "Refine: bad arguments");
    if (mode < 0 || mode > 2)
        return fail(SMHIP_EINVAL, "Refine: unknown mode %d", mode);
    if (mode == 2) {
        rc = check_dsi(d_cost, w, h, nb, row_elems);
        if (rc) return rc;
    }
    if (mode == 1) {
        rc = check_dsi(d_min, w, h, nb, row_elems);
        if (rc) return rc;
    }
    long n_pix = (long)w * h;
    smk::refine_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
        d_cost, d_min, d_disp, w, h, nb, row_elems, disp_step, disp_min, mode, d_fdisp);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_match_quality(smhip_ctx *ctx, const float *d_cost, const float *d_cert, const float *d_fdisp,
                        int w, int h, int nb, size_t row_elems, size_t cert_row_elems, float disp_min,
                        float disp_step_inv, int quality_mode, int use_fit, int norm_bands, int take_sqrt,
                        float *d_final_cost, float *d_certainty)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_fdisp || !d_final_cost || !d_certainty)
        return fail(SMHIP_EINVAL, "ComputeMatchQuality: null argument");
    if (nb < 3 && !use_fit)
        return fail(SMHIP_EINVAL, "ComputeMatchQuality: the parabola needs 3 disparity levels (%d given)", nb);
    if (use_fit && (!d_cert || cert_row_elems < (size_t)w * nb))
        return fail(SMHIP_EINVAL, "ComputeMatchQuality: aggr_subpixel needs the certainty volume");
    if (norm_bands != 1 && norm_bands != 3)
        return fail(SMHIP_EINVAL, "ComputeMatchQuality: costs are normalised by 1 or 3 bands (%d given)", norm_bands);
    long n_pix = (long)w * h;
    smk::match_quality_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
        d_cost, d_cert, d_fdisp, w, h, nb, row_elems, cert_row_elems, disp_min, disp_step_inv,
        quality_mode == 2, use_fit != 0, norm_bands, take_sqrt != 0, d_final_cost, d_certainty);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_opt_so(smhip_ctx *ctx, const float *d_cost, int w, int h, int nb, size_t row_elems,
                 const uint8_t *d_ref, size_t img_pitch, int img_bands, float opt_smoothness,
                 float opt_grad_thresh, float opt_grad_penalty, int32_t *d_disp, float *d_best)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_ref || !d_disp || img_bands < 1 || img_pitch < (size_t)w * img_bands)
        return fail(SMHIP_EINVAL, "OptSO: bad image / output arguments");
    if (nb > 1024)
        return fail(SMHIP_EINVAL, "OptSO: at most 1024 disparity levels (%d given)", nb);
    void *smooth = nullptr, *trans = nullptr;
    rc = ws_get(ctx, WS_BT, (size_t)w * h * sizeof(float), &smooth);
    if (rc) return rc;
    rc = ws_get(ctx, WS_TMP, (size_t)w * h * nb * sizeof(int16_t), &trans);
    if (rc) return rc;
    dim3 gs((w + 255) / 256, h);
    smk::smooth_h_kernel<<<gs, 256, stage_lds(ctx), ctx->stream>>>(d_ref, img_pitch, w, h, img_bands, opt_smoothness,
                                                     opt_grad_thresh, opt_grad_penalty, (float *)smooth);
#define LAUNCH_SO(NV) smk::opt_so_kernel<NV><<<h, 64, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, \
        (const float *)smooth, (int16_t *)trans, d_disp)
    if (nb <= 64)       LAUNCH_SO(1);
    else if (nb <= 128) LAUNCH_SO(2);
    else if (nb <= 256) LAUNCH_SO(4);
    else if (nb <= 512) LAUNCH_SO(8);
    else                LAUNCH_SO(16);
#undef LAUNCH_SO
    if (d_best) {
        long n_pix = (long)w * h;
        smk::gather_best_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            d_cost, w, h, nb, row_elems, d_disp, d_best);
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_opt_dp(smhip_ctx *ctx, const float *d_cost, int w, int h, int nb, size_t row_elems,
                 const uint8_t *d_ref, size_t img_pitch, int img_bands, float opt_smoothness,
                 float opt_grad_thresh, float opt_grad_penalty, int opt_occlusion_cost, int32_t *d_disp,
                 float *d_best)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_ref || !d_disp || img_bands < 1 || img_pitch < (size_t)w * img_bands)
        return fail(SMHIP_EINVAL, "OptDP: bad image / output arguments");
    if (nb > 1024)
        return fail(SMHIP_EINVAL, "OptDP: at most 1024 disparity levels (%d given)", nb);
    void *smooth = nullptr, *trans = nullptr;
    rc = ws_get(ctx, WS_BT, (size_t)w * h * sizeof(float), &smooth);
    if (rc) return rc;
    rc = ws_get(ctx, WS_TMP, (size_t)w * h * nb, &trans);
    if (rc) return rc;
    dim3 gs((w + 255) / 256, h);
    smk::smooth_h_kernel<<<gs, 256, stage_lds(ctx), ctx->stream>>>(d_ref, img_pitch, w, h, img_bands, opt_smoothness,
                                                     opt_grad_thresh, opt_grad_penalty, (float *)smooth);
    const float oc = (float)opt_occlusion_cost;      // ocL = ocR, StcOptDP.cpp:116-117
#define LAUNCH_DP(NV) smk::opt_dp_kernel<NV><<<h, 64, stage_lds(ctx), ctx->stream>>>(d_cost, w, h, nb, row_elems, \
        (const float *)smooth, oc, (uint8_t *)trans, d_disp)
    if (nb <= 64)       LAUNCH_DP(1);
    else if (nb <= 128) LAUNCH_DP(2);
    else if (nb <= 256) LAUNCH_DP(4);
    else if (nb <= 512) LAUNCH_DP(8);
    else                LAUNCH_DP(16);
#undef LAUNCH_DP
    if (d_best) {
        long n_pix = (long)w * h;
        smk::gather_best_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            d_cost, w, h, nb, row_elems, d_disp, d_best);
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

namespace {

// levels per lane of the smoothing kernel: 1, 2, 4 or 8 (64 ... 512 levels)
int bayes_slots(int nb) { return nb <= 64 ? 1 : nb <= 128 ? 2 : nb <= 256 ? 4 : 8; }

// w(d, d') = rhoP(d - d') / row sum, on the host as the reference computes it (:209-222), then the
// band (w - e_d) and e_d = epsP / s_d the kernel sums (see smhip_bayes.h); cached per context
int bayes_prepare(smhip_ctx *ctx, const float *d_cost, const float *d_cost2, const float *d_cost0, int w, int h,
                  int nb, size_t row_elems, float sigmaP, float epsP)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_dsi(d_cost, w, h, nb, row_elems);
    if (rc) return rc;
    if (!d_cost2 || !d_cost0 || d_cost2 == d_cost || d_cost0 == d_cost || d_cost0 == d_cost2)
        return fail(SMHIP_EINVAL, "Bayesian diffusion needs three distinct volumes (E, Es, E0)");
    if (nb > SMHIP_BAYES_MAX_LEVELS)                 // the reference asserts m_disp_n < 100 (:200-202)
        return fail(SMHIP_EINVAL, "Bayesian diffusion: at most %d disparity levels (%d given)",
                    SMHIP_BAYES_MAX_LEVELS, nb);
    static_assert(SMHIP_BAYES_MAX_LEVELS == smk::BAYES_MAXD, "header and kernel limits differ");
    const int DP = bayes_slots(nb) * 64;
    const size_t band_cap = (size_t)(2 * SMHIP_BAYES_MAX_LEVELS - 1) * SMHIP_BAYES_MAX_LEVELS;
    if (ctx->bayes_n == nb && ctx->bayes_sigma == sigmaP && ctx->bayes_eps == epsP)
        return SMHIP_OK;
    const size_t cap = (band_cap + SMHIP_BAYES_MAX_LEVELS) * sizeof(double);
    if (!ctx->bayes_wh && !(ctx->bayes_wh = (double *)malloc(cap)))
        return fail(SMHIP_ENOMEM, "out of host memory");
    if (!ctx->bayes_w)
        HIP_TRY(hipMalloc((void **)&ctx->bayes_w, cap));
    HIP_TRY(hipStreamSynchronize(ctx->stream));      // an earlier iteration may still read the old ones
    int B = 0;
    for (int d = 1; d < nb; d++) {
        float diff = d;
        float r = ((1 - epsP) * expf(-diff * diff / (2 * sigmaP * sigmaP)) + epsP);
        if (!(r == epsP))
            B = d;
    }
    double *band = ctx->bayes_wh, *ed = band + band_cap;
    std::vector<double> row(nb);
    memset(band, 0, cap);
    for (int d = 0; d < nb; d++) {
        float s = 0.0;
        for (int d2 = 0; d2 < nb; d2++) {
            float diff = (d - d2);
            s += (row[d2] = ((1 - epsP) * expf(-diff * diff / (2 * sigmaP * sigmaP)) + epsP));
        }
        for (int d2 = 0; d2 < nb; d2++)
            row[d2] /= s;
        double e = B < nb - 1 ? (double)epsP : 0.0;   // the full product has no constant part
        e /= s;
        ed[d] = e;
        for (int k = -B; k <= B; k++)
            if (d + k >= 0 && d + k < nb)
                band[(size_t)(k + B) * DP + d] = row[d + k] - e;
    }
    HIP_TRY(hipMemcpyAsync(ctx->bayes_w, band, cap, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->bayes_n = nb;
    ctx->bayes_sigma = sigmaP;
    ctx->bayes_eps = epsP;
    ctx->bayes_band = B;
    return SMHIP_OK;
}

// steps 1-3 of one iteration: E (or, fused, E0 + mu * 5-point(es_in)) -> es_out
void bayes_smooth(smhip_ctx *ctx, float *e, float *es_out, float *e0_init, float scale_cost, const float *es_in,
                  float *e0, float mu, int w, int h, int nb, size_t row_elems)
{
    const size_t band_cap = (size_t)(2 * SMHIP_BAYES_MAX_LEVELS - 1) * SMHIP_BAYES_MAX_LEVELS;
    long n_pix = (long)w * h;
    unsigned blocks = (unsigned)((n_pix + 3) / 4 < 4096 ? (n_pix + 3) / 4 : 4096);
    const int slots = bayes_slots(nb);
    // sigmaP up to about 0.5: the band lives in registers (one or two levels per lane only: 13 doubles each)
    const bool narrow = ctx->bayes_band <= 6 && slots <= 2;
    const size_t lds = (size_t)4 * 3 * 64 * slots * sizeof(double);
#define LAUNCH_BS(S, HB)                                                                                    \
    do {                                                                                                    \
        if (es_in) smk::bayes_smooth_kernel<S, HB, true><<<blocks, 256, lds, ctx->stream>>>(                \
            nullptr, es_out, e0, 0.0f, ctx->bayes_w, ctx->bayes_w + band_cap, ctx->bayes_band, w, h, nb,    \
            row_elems, es_in, mu);                                                                          \
        else smk::bayes_smooth_kernel<S, HB, false><<<blocks, 256, lds, ctx->stream>>>(                     \
            e, es_out, e0_init, scale_cost, ctx->bayes_w, ctx->bayes_w + band_cap, ctx->bayes_band, w, h,   \
            nb, row_elems);                                                                                 \
    } while (0)
    switch (slots) {
    case 1: if (narrow) LAUNCH_BS(1, 6); else LAUNCH_BS(1, 0); break;
    case 2: if (narrow) LAUNCH_BS(2, 6); else LAUNCH_BS(2, 0); break;
    case 4: LAUNCH_BS(4, 0); break;
    default: LAUNCH_BS(8, 0); break;
    }
#undef LAUNCH_BS
}

}  // namespace

int smhip_bayesian_iter(smhip_ctx *ctx, float *d_cost, float *d_cost2, float *d_cost0, int w, int h, int nb,
                        size_t row_elems, float scale_cost, float mu, float sigmaP, float epsP, int iter)
{
    if (iter < 0)
        return fail(SMHIP_EINVAL, "negative iteration index");
    int rc = bayes_prepare(ctx, d_cost, d_cost2, d_cost0, w, h, nb, row_elems, sigmaP, epsP);
    if (rc) return rc;
    bayes_smooth(ctx, d_cost, d_cost2, iter == 0 ? d_cost0 : nullptr, scale_cost, nullptr, nullptr, mu, w, h, nb,
                 row_elems);
    dim3 ge((w * nb + 255) / 256, h);
    smk::bayes_update_kernel<<<ge, 256, stage_lds(ctx), ctx->stream>>>(d_cost, d_cost2, d_cost0, w, nb, h, row_elems, mu);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_bayesian_run(smhip_ctx *ctx, float *d_cost, float *d_cost2, float *d_cost0, int w, int h, int nb,
                       size_t row_elems, float scale_cost, float mu, float sigmaP, float epsP, int n_iter,
                       int *result_in_b)
{
    if (n_iter < 0 || !result_in_b)
        return fail(SMHIP_EINVAL, "Bayesian diffusion: bad iteration count / result flag");
    int rc = bayes_prepare(ctx, d_cost, d_cost2, d_cost0, w, h, nb, row_elems, sigmaP, epsP);
    if (rc) return rc;
    *result_in_b = 0;
    if (n_iter == 0)
        return SMHIP_OK;
    // iteration 0: E = scale * cost (kept as E0), Es -> d_cost2.  Iteration k: E is formed inside the
    // smoothing kernel from the previous Es, which ping-pongs between d_cost2 and d_cost (E itself is not
    // needed between iterations).  The last step materialises E in the buffer that does not hold Es.
    bayes_smooth(ctx, d_cost, d_cost2, d_cost0, scale_cost, nullptr, nullptr, mu, w, h, nb, row_elems);
    float *es = d_cost2, *other = d_cost;
    for (int it = 1; it < n_iter; it++) {
        bayes_smooth(ctx, nullptr, other, nullptr, 0.0f, es, d_cost0, mu, w, h, nb, row_elems);
        float *t = es; es = other; other = t;
    }
    dim3 ge((w * nb + 255) / 256, h);
    smk::bayes_update_kernel<<<ge, 256, stage_lds(ctx), ctx->stream>>>(other, es, d_cost0, w, nb, h, row_elems, mu);
    *result_in_b = (other == d_cost2);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_keys_decode(smhip_ctx *ctx, const int64_t *d_keys, int width, int height,
                      int32_t *d_disp, float *d_best)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_keys || width < 1 || height < 1 || (!d_disp && !d_best))
        return fail(SMHIP_EINVAL, "bad decode request");
    dim3 grid((width + 31) / 32, (height + 31) / 32);
    smk::keys_decode_t_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>((const long long *)d_keys, width, height,
                                                            d_disp, d_best);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

// RawCosts + one AggrBox on the slab [d_first, d_first + d_count), then either PadCosts + WTA ->
// keys (d_keys) or the aggregated DSI itself (d_dsi, reference layout, no PadCosts yet).
static int fused_impl(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                      const smhip_match_params *p, int window, int d_first, int d_count,
                      int64_t *d_keys, float *d_dsi, size_t dsi_row_elems, bool merge = false)
{
    const int keep_all = d_dsi != nullptr;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_params(p, d_ref, d_mtc, img_pitch);
    if (rc) return rc;
    if (p->disp_den != 1)   // the hosts route disp_step < 1 through the staged entry points
        return fail(SMHIP_EINVAL, "fused path: disp_step < 1 (den=%d) not supported, use smhip_raw_costs",
                    p->disp_den);
    if (!d_keys && !d_dsi)
        return fail(SMHIP_EINVAL, "null output buffer");
    if (d_first < 0 || d_count < 1 || d_first + d_count > p->disp_n)
        return fail(SMHIP_EINVAL, "disparity slab [%d,%d) outside [0,%d)", d_first, d_first + d_count,
                    p->disp_n);
    int W = p->width, H = p->height;
    if (window < 1 || window > W || window > H)           // as smhip_box_filter
        return fail(SMHIP_EINVAL, "BoxFilter: window %d unusable for %dx%d", window, W, H);

    // padded slab width Dp: a power of two below 64 (narrow kernels), else 64 / 128 / 256 with
    // 1 / 2 / 4 levels per lane, else chunks of 256 levels
    int Dp, CH;
    const bool img_aligned = p->n_bands == 1 || (img_pitch % 4 == 0 && ((uintptr_t)d_ref & 3) == 0 &&
                                                 ((uintptr_t)d_mtc & 3) == 0);
    if (d_count <= 32 && !(img_aligned && ctx->opt[SMHIP_OPT_NARROW] != 1)) {
        Dp = 2;                      // general-shape kernels (4-band images with unaligned rows)
        while (Dp < d_count) Dp <<= 1;
        CH = Dp;
    } else if (d_count <= 64) {      // also narrow slabs: lanes beyond d_count idle, T stays sparse
        Dp = CH = 64;
    } else if (d_count <= 128) {
        Dp = CH = 128;
    } else if (d_count <= 256) {
        Dp = CH = 256;
    } else {
        Dp = (d_count + 255) / 256 * 256;
        CH = 256;
    }
    int n_chunks = Dp / CH;
    size_t n_pix = (size_t)W * H;
    const bool wide = (Dp % 64 == 0);
    const bool tuned_a = wide && (p->n_bands == 1 || (img_pitch % 4 == 0 && ((uintptr_t)d_ref & 3) == 0 &&
                                                      ((uintptr_t)d_mtc & 3) == 0));
    if (keep_all && !(wide && tuned_a && d_count % 4 == 0 && dsi_row_elems % 4 == 0 &&
                      ((uintptr_t)d_dsi & 15) == 0))
        return fail(SMHIP_EINVAL, "fused box -> DSI needs >= 33 levels, D %% 4 == 0 and 16-byte aligned rows");
    // banded slab layout (smhip_fused.h) only when both passes are the tuned kernels
    int tyshift = tuned_a ? 5 : 0;
    if (tuned_a && ctx->opt[SMHIP_OPT_TYSHIFT] >= 0)
        tyshift = ctx->opt[SMHIP_OPT_TYSHIFT];           // (smhip_set_option keeps it within 0..8)
    // 128-level slabs (every GPU's share of config 5 at N = 8): columns in pairs, so that pass B reads a row of
    // two columns with one 1-KiB wave load (smhip_fused_b.h, PAIR).  Only with the float-lane pass A, which
    // writes that layout, and from 3584 columns on: a wave per pair halves the waves, and below about 1800 of
    // them the walk is bound by the latency of a wave's own row loop (3840 wide: pass B 0.79 -> 0.69 ms, 7680:
    // 3.22 -> 2.84; 3072: 0.49 -> 0.51, 1920: 0.25 -> 0.27).  SMHIP_OPT_PAIRED = 0 switches it off, 1 forces it
    // at any width (tests).
    bool paired = tuned_a && !keep_all && Dp == 128 && p->n_bands == 1 && ctx->opt[SMHIP_OPT_PAIRED] != 0 &&
                  (W >= 3584 || ctx->opt[SMHIP_OPT_PAIRED] > 0) &&
                  ctx->opt[SMHIP_OPT_CHUNK] < 0 && ctx->opt[SMHIP_OPT_FLOAT_A] != 0 &&
                  (window == 3 || window == 5 || window == 7 || window == 9 || window == 11);
    const size_t Wc = paired ? ((size_t)W + 1) / 2 : (size_t)W, DpL = paired ? 256 : (size_t)Dp;
    while (tyshift > 0 && ((((unsigned long long)Wc << tyshift) * (unsigned long long)DpL) >> 32) != 0)
        tyshift--;                                      // pass B multiplies the band size as a 32-bit number
    size_t bands = ((size_t)H + (1u << tyshift) - 1) >> tyshift;
    size_t trow = Wc * DpL;                             // floats per image row of T
    size_t t_elems = (bands << tyshift) * trow;

    // pass A stages whole rows in LDS; rows above 40 KB (4 bands: wider than 2560 pixels) are staged
    // chunk by chunk (smhip_fused_chunk.h); SMHIP_OPT_CHUNK = 1 / 0 forces the choice in tests (and with
    // it the integer kernels)
    const bool ref_point = !(p->match_interval && p->match_interpolated);   // one band: RowLds<2>, 8 bytes per pixel
    bool chunked = (size_t)W * (p->n_bands == 1 ? (ref_point ? 8 : 4) : 16) > 40 * 1024;
    const bool chunked_forced = ctx->opt[SMHIP_OPT_CHUNK] >= 0;
    if (chunked_forced) chunked = ctx->opt[SMHIP_OPT_CHUNK] != 0;
    // one band, whole-row kernel: the matching row's Birchfield-Tomasi interval is formed while the row
    // is staged, so the pre-pass over the image (bt_minmax_kernel) and its workspace are not needed
    // (bit-identical, 0.8 % faster per step); SMHIP_OPT_BT_PREPASS = 1 keeps the pre-pass (tests)
    const bool bt_inline = tuned_a && !chunked && p->n_bands == 1 && p->match_interval &&
                           !p->match_interpolated && ctx->opt[SMHIP_OPT_BT_PREPASS] != 1;

    // every workspace request comes before the first launch (prepare_raw asks for the interval images and then
    // runs the pre-pass): a call that cannot get its memory -- the no-allocation entry point on a problem beyond
    // its static scratch -- is refused with nothing queued
    void *T = nullptr;
    rc = ws_get(ctx, WS_TMP, t_elems * sizeof(float), &T);
    if (rc) return rc;
    long long *kout = (long long *)d_keys;
    // merge (smhip_fused_box_wta_keys_merge): the wide pass B of a single chunk takes the minimum with the old key
    // where it stores; every other shape writes its keys to the workspace and keys_min_kernel folds d_keys in
    const bool merge_in_b = merge && n_chunks == 1 && wide && !keep_all;
    const bool merge_after = merge && !merge_in_b && !keep_all;
    if ((n_chunks > 1 || merge_after) && !keep_all) {
        void *part = nullptr;
        rc = ws_get(ctx, WS_KEYS, n_pix * n_chunks * sizeof(long long), &part);
        if (rc) return rc;
        kout = (long long *)part;
    }
    // (bit 1: non-temporal key stores in pass B, frames of 16 M pixels and more -- smhip_fused_b.h)
    //  SMHIP_OPT_STORE_POLICY = 18, the large-slab policy of pass A, forces them as well: tests)
    const int merge_b = (merge_in_b ? 1 : 0) |
                        ((n_pix >= ((size_t)16 << 20) || ctx->opt[SMHIP_OPT_STORE_POLICY] == 18) ? 2 : 0);
    smk::RawArgs a;
    rc = prepare_raw(ctx, d_ref, d_mtc, img_pitch, p, d_first, d_count, &a, bt_inline);
    if (rc) return rc;
    float scale = (float)(1.0 / window);
    float outside = smhip_match_outside(p);

    bool prof = ctx->ev && ctx->ev_used < ctx->ev_cap;
    hipEvent_t *ev = prof ? ctx->ev + 3 * ctx->ev_used : nullptr;
    if (prof) HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    bool float_a_ran = false;
    if (tuned_a) {
        smk::FusedA fa;
        fa.rn = a.rmn; fa.rx = a.rmx ? a.rmx : a.rmn;
        fa.mn = a.mmn; fa.mx = a.mmx ? a.mmx : a.mmn;
        fa.pitch = img_pitch; fa.W = W; fa.H = H; fa.D = d_count;
        fa.disp0 = a.disp0; fa.dstep = a.dstep; fa.cutoff = a.cutoff; fa.outside = a.outside;
        fa.bt_inline = bt_inline;
        int tpa = Dp < 256 ? Dp : 256;
        dim3 ga(H, Dp / tpa);
        const int nbands = p->n_bands;
        size_t lds = (size_t)W * (nbands == 1 ? (ref_point ? 8 : 4) : 16);
        const int adstep = a.dstep < 0 ? -a.dstep : a.dstep;
        const int static_win = (window >= 3 && window <= 17 && (window & 1)) ? window : 0;
        int chunk_steps = static_win ? (480 / static_win) * static_win : 480;
        if (chunk_steps > W) chunk_steps = static_win ? ((W + static_win - 1) / static_win) * static_win : W;
        long nref = (long)chunk_steps + window + 2, nmatch = nref + (long)(tpa - 1) * adstep;
        if (nref > W) nref = W;
        if (nmatch > W) nmatch = W;
        if (chunked)
            lds = nbands == 1 ? (size_t)(nref + nmatch) * (ref_point ? 4 : 2)   // (four bands: RowLds<4> 8 + 8 bytes, RowLds<6> 4 + 12)
                              : (size_t)nref * 8 + (size_t)nmatch * 12;
        if (lds > 150 * 1024)
            return fail(SMHIP_EINVAL, "row staging needs %zu bytes of LDS (width %d, window %d, step %d)", lds, W,
                        window, adstep);
        // one band, unit step, point reference interval, window 3..11: the float-lane kernel
        // (smhip_fused_f.h, a translation unit of its own); everything else: the integer kernels below
        bool float_a = false;
        if (!chunked_forced && ctx->opt[SMHIP_OPT_FLOAT_A] != 0) {
            const int max_cost = nbands * (a.sd ? 255 * 255 : 255), pol = ctx->opt[SMHIP_OPT_STORE_POLICY];
            const long a_lds_min = ctx->opt[SMHIP_OPT_A_LDS_KB] < 0 ? -1 : (long)ctx->opt[SMHIP_OPT_A_LDS_KB] * 1024;
            const int a_one_wave = ctx->opt[SMHIP_OPT_A_NW];     // -1: the launcher's policy; 0 / 1 / 4 forced
            if (paired)
                paired = smk::launch_fused_rawbox_h_f(fa, nbands, (float *)T, tyshift, Dp, window, scale, keep_all, 4,
                                                      a.sd != 0, max_cost, ctx->stream, pol, true, a_lds_min, a_one_wave);
            float_a = paired || smk::launch_fused_rawbox_h_f(fa, nbands, (float *)T, tyshift, Dp, window, scale,
                                                             keep_all, CH / 64, a.sd != 0, max_cost, ctx->stream,
                                                             pol, false, a_lds_min, a_one_wave);
        } else {
            paired = false;
        }
#define LAUNCH_A3(WIN, SD, NB)                                                                           \
        do {                                                                                             \
            if (chunked) smk::fused_rawbox_h_chunk_kernel<WIN, SD, NB><<<ga, tpa, lds, ctx->stream>>>(   \
                fa, (float *)T, tyshift, Dp, window, scale, keep_all, chunk_steps, (int)nref, (int)nmatch); \
            else smk::fused_rawbox_h_lds_kernel<WIN, SD, NB><<<ga, tpa, lds, ctx->stream>>>(             \
                fa, (float *)T, tyshift, Dp, window, scale, keep_all);                                   \
        } while (0)
#define LAUNCH_A(WIN)                                                             \
        do {                                                                      \
            if (nbands == 1 && ref_point && fa.rn == fa.rx) { if (a.sd) LAUNCH_A3(WIN, true, 2); else LAUNCH_A3(WIN, false, 2); } \
            else if (nbands == 1) { if (a.sd) LAUNCH_A3(WIN, true, 1); else LAUNCH_A3(WIN, false, 1); } \
            else if (fa.rn == fa.rx && fa.mn == fa.mx) { if (a.sd) LAUNCH_A3(WIN, true, 5); else LAUNCH_A3(WIN, false, 5); } \
            else if (fa.rn == fa.rx && !a.sd) LAUNCH_A3(WIN, false, 6); \
            else             { if (a.sd) LAUNCH_A3(WIN, true, 4); else LAUNCH_A3(WIN, false, 4); } \
        } while (0)
        float_a_ran = float_a;
        if (!float_a)
        switch (window) {            // static windows: the leaving cost comes from a register ring
        case 3: LAUNCH_A(3); break;
        case 5: LAUNCH_A(5); break;
        case 7: LAUNCH_A(7); break;
        case 9: LAUNCH_A(9); break;
        case 11: LAUNCH_A(11); break;
        case 13: LAUNCH_A(13); break;
        case 15: LAUNCH_A(15); break;
        case 17: LAUNCH_A(17); break;
        default: LAUNCH_A(0); break;
        }
#undef LAUNCH_A
#undef LAUNCH_A3
    } else {
        long chains = (long)H * Dp;
        smk::fused_rawbox_h_kernel<<<(unsigned)((chains + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            a, (float *)T, trow, Dp, window, scale);
    }
    ctx->last_pass_a = !tuned_a ? "fused_rawbox_h_kernel" : float_a_ran ? "fused_rawbox_h_f_kernel"
                       : chunked ? "fused_rawbox_h_chunk_kernel" : "fused_rawbox_h_lds_kernel";
    if (prof) HIP_TRY(hipEventRecord(ev[1], ctx->stream));

    if (wide) {
        // one wave per pixel column and chunk, 4 columns per block, columns dealt per XCD
        // Resident pass B waves per CU, bounded through the LDS a workgroup asks for (160 KB per CU).  Every wave
        // walks a whole column, so the grid runs in rounds; with b workgroups per XCD (32 CUs), two workgroups
        // per CU (rounds of 64) beat the four that the key buffer alone allows whenever the last round of 64 is
        // at least 0.6 full -- b = 104 .. 120: 14-19 % faster (3840x2160x256 1.50 -> 1.29 ms, 0.82 of 8 TB/s),
        // 176 .. 240: 2-5 % -- and lose otherwise (b = 80: 20 % slower, 96: 9 %, 136: 11 %, 160: 4 %): measured,
        // scripts/exp_occupancy*.sh, profiles/r02_occupancy.log.  SMHIP_OPT_B_WAVES: 0 no bound, n waves per CU.
        const int wpb_policy = 4;
        int x_per_xcd = (W + 7) / 8;
        const int b_chunks = keep_all ? (Dp + 255) / 256 : n_chunks;
        int b_waves = ctx->opt[SMHIP_OPT_B_WAVES] > 0 ? (ctx->opt[SMHIP_OPT_B_WAVES] + 1) & ~1 : 0;
        const bool b_policy = ctx->opt[SMHIP_OPT_B_WAVES] < 0;
        auto two_per_cu = [](long b) { return b > 64 && (b % 64 == 0 || b % 64 >= 38); };
        const long b_per_xcd = (long)((x_per_xcd + wpb_policy - 1) / wpb_policy) * b_chunks;
        // (with more than two chunks only up to two rounds: 1920x1080x1024, b = 240 in four chunks, is 3 % slower under
        // the bound and 960x540x2048, eight chunks, 14 %, while 1000x1500x1024 and 1280x720x768, b = 128 / 120, gain 9 %)
        if (b_policy && !keep_all && CH == 256 && two_per_cu(b_per_xcd) && (b_chunks <= 2 || b_per_xcd <= 128))
            b_waves = 8;
        const int wpb = (b_waves % 4) ? 2 : 4;
        dim3 gb(8 * ((x_per_xcd + wpb - 1) / wpb), b_chunks);
        // keys collect in a wave-private LDS buffer of kb_rows rows (a multiple of the kernel's key group)
        // and go to memory in long runs: stores into the read stream cost HBM bus turn-arounds
        const bool b_static = window >= 3 && window <= 17 && (window & 1);
        const int kblk = smk::passb_blk(b_static ? window : 0);
        // slabs of up to 128 levels: four blocks of rows in flight where few waves walk (up to 2048 columns; windows
        // up to 11): 1920x1080x128 0.252 -> 0.244 (two) -> 0.239 ms (four), x64 0.213 -> 0.196, 1280x720x128 0.170 ->
        // 0.160; 3072 columns: 0.470 / 0.471 / 0.478 -- and there the minimum over the lanes is taken per group of rows
        // through LDS instead of per row with DPP (smhip_fused_b.h, LRED); SMHIP_OPT_B_LRED = 1 / 0 forces / forbids it.
        // 128-level waves beyond 2048 columns: 2240 3 % faster, 2560 equal, 3072 7 % slower; 64-level waves gain up to
        // 4096 columns (2560: 0.297 -> 0.214 ms, 3072: 0.360 -> 0.271, 3840: 0.510 -> 0.446, 4096: 0.560 -> 0.519)
        const bool ahead2 = ctx->opt[SMHIP_OPT_B_AHEAD] != 1;
        const bool few_cols = (long)W * n_chunks <= 2048;
        const bool ahead4 = ahead2 && few_cols && ctx->opt[SMHIP_OPT_B_AHEAD] != 2;
        const bool lred = !keep_all && !paired && b_static && window <= 11 && CH <= 128 &&
                          (ctx->opt[SMHIP_OPT_B_LRED] < 0 ? (ahead4 || (CH == 64 && (long)W * n_chunks <= 4096))
                                                          : ctx->opt[SMHIP_OPT_B_LRED] != 0);
        const int kgrp = ((lred ? 32 : 64) / kblk) * kblk;
        // (two workgroups per CU leave each 80 KB: 36 groups, 8K x 256 5.54 -> 5.42 ms)
        int kb_groups = ctx->opt[SMHIP_OPT_KEYBUF] >= 0 ? ctx->opt[SMHIP_OPT_KEYBUF] : (b_policy && b_waves == 8 ? 36 : 18);   // 0: unbuffered
        if (lred && ctx->opt[SMHIP_OPT_KEYBUF] < 0) {
            kb_groups = (1134 + kgrp - 1) / kgrp;           // (the same ~9-KB runs out of shorter groups)
            // with that buffer and the reduction tiles (8320 bytes per wave) two workgroups share a CU: rounds of 64 per
            // XCD; a smaller buffer where three or four per CU fill their rounds better (as for the paired walk below)
            const long wg_xcd = (long)((x_per_xcd + wpb - 1) / wpb) * b_chunks;
            auto fill = [&](long per_cu) {
                const long slots = 32 * per_cu;
                return (double)wg_xcd / (double)((wg_xcd + slots - 1) / slots * slots);
            };
            int per_cu = 2;
            if (b_policy && wg_xcd <= 128) {                // all resident at once where four per CU can hold them
                per_cu = wg_xcd <= 64 ? 2 : wg_xcd <= 96 ? 3 : 4;
            } else if (b_policy && fill(2) < 0.8) {
                per_cu = fill(3) >= fill(4) - 0.05 ? 3 : 4;
                if (fill(per_cu) <= fill(2)) per_cu = 2;
            }
            // (a workgroup that asks for exactly a third of the 160 KB gets a CU with one other, not two: 3 KB of margin)
            const long room = 163840 / per_cu - 3072 - (long)wpb * 32 * 65 * (long)sizeof(int);
            const int fit = (int)(room / ((long)wpb * kgrp * (long)sizeof(long long)));
            if (per_cu > 2 && kb_groups > fit) kb_groups = fit > 1 ? fit : 1;
        }
        if (kb_groups > (H + kgrp - 1) / kgrp) kb_groups = (H + kgrp - 1) / kgrp;
        if (kb_groups > 64) kb_groups = 64;                 // 4 waves x 64 groups x 64 rows x 8 B = 128 KB
        const int kb_rows = kgrp * kb_groups;
        size_t kb_lds = (size_t)wpb * kb_rows * sizeof(long long) + (lred ? (size_t)wpb * 32 * 65 * sizeof(int) : 0);
        // 160 KB of LDS per CU, the exact share.  (Asking for 4 KB less -- 76 instead of 80 KB, still two workgroups
        // per CU -- measured 1.44 instead of 1.29 ms at 3840x2160x256: unexplained, so the share stays exact; a third
        // of 160 KB, 53.25 KB, gives two per CU, not three.)
        const size_t b_lds_min = b_waves ? (size_t)(163840 / (b_waves / wpb)) & ~(size_t)255 : 0;
        if (kb_lds < b_lds_min) kb_lds = b_lds_min;
#define LAUNCH_BT(WIN, VPL) smk::fused_boxv_wta_vpl_kernel<WIN, VPL><<<gb, 64 * wpb, kb_lds, ctx->stream>>>( \
        (const float *)T, tyshift, W, H, Dp, d_count, window, scale, a.disp0, a.dstep, outside,        \
        d_first, x_per_xcd, kout, n_pix, nullptr, 0, kb_rows, merge_b)
#define LAUNCH_BT2(WIN, VPL) LAUNCH_BTA(WIN, VPL, 2)
#define LAUNCH_BT4(WIN, VPL) LAUNCH_BTA(WIN, VPL, (WIN <= 11 ? 4 : 2))
#define LAUNCH_BTL(WIN, VPL) smk::fused_boxv_wta_vpl_kernel<(WIN >= 3 && WIN <= 11 ? WIN : 9), VPL, 0, false, false, 4, true><<<gb, 64 * wpb, kb_lds, ctx->stream>>>( \
        (const float *)T, tyshift, W, H, Dp, d_count, window, scale, a.disp0, a.dstep, outside,        \
        d_first, x_per_xcd, kout, n_pix, nullptr, 0, kb_rows, merge_b)
#define LAUNCH_BTA(WIN, VPL, AH) smk::fused_boxv_wta_vpl_kernel<WIN, VPL, 0, false, false, AH><<<gb, 64 * wpb, kb_lds, ctx->stream>>>( \
        (const float *)T, tyshift, W, H, Dp, d_count, window, scale, a.disp0, a.dstep, outside,        \
        d_first, x_per_xcd, kout, n_pix, nullptr, 0, kb_rows, merge_b)
        // slabs of up to 128 levels (256- / 512-byte rows per wave): two blocks of rows in flight instead of one,
        // 3-4 % faster from 1280x720 to 3840x2160 (one block of nine such rows per wave is less than the memory's
        // latency x bandwidth); 256-level waves gain nothing from it (0.335 against 0.331 ms).  SMHIP_OPT_B_AHEAD = 1: off
#define LAUNCH_BW(WIN)                                 \
        switch (CH / 64) {                             \
        case 1: if (lred && WIN) LAUNCH_BTL(WIN, 1); else if (ahead4 && WIN) LAUNCH_BT4(WIN, 1); else if (ahead2 && WIN) LAUNCH_BT2(WIN, 1); else LAUNCH_BT(WIN, 1); break; \
        case 2: if (lred && WIN) LAUNCH_BTL(WIN, 2); else if (ahead4 && WIN) LAUNCH_BT4(WIN, 2); else if (ahead2 && WIN) LAUNCH_BT2(WIN, 2); else LAUNCH_BT(WIN, 2); break; \
        default: LAUNCH_BT(WIN, 4); break;             \
        }
        if (paired) {                // (the float-lane pass A ran: window 3 .. 11, Dp 128, keys out)
            const int xp = (((W + 1) / 2) + 7) / 8;     // column pairs per XCD
            dim3 gp(8 * ((xp + wpb - 1) / wpb), 1);
            const int pgrp = (32 / smk::passb_blk(window)) * smk::passb_blk(window);   // rows per key group of a half-wave
            int pg = ctx->opt[SMHIP_OPT_KEYBUF] >= 0 ? ctx->opt[SMHIP_OPT_KEYBUF] : (1134 + pgrp - 1) / pgrp;   // runs of ~9 KB per column, as above
            if (ctx->opt[SMHIP_OPT_KEYBUF] < 0 && b_policy) {
                // That buffer (72 KB per workgroup) lets two workgroups share a CU, and the grid runs in rounds of 64 per
                // XCD: when the last round would be less than 0.8 full, a smaller buffer -- three or four workgroups per
                // CU -- that fills its rounds better (5120x2880x128, 80 workgroups per XCD: 1.54 ms in rounds of 64)
                const long wg_xcd = (xp + wpb - 1) / wpb;
                auto fill = [&](long per_cu) {
                    const long slots = 32 * per_cu;
                    return (double)wg_xcd / (double)((wg_xcd + slots - 1) / slots * slots);
                };
                int per_cu = 2;
                if (fill(2) < 0.8) {
                    per_cu = fill(3) >= fill(4) - 0.05 ? 3 : 4;
                    if (fill(per_cu) <= fill(2)) per_cu = 2;
                }
                const int fit = (int)((163840 / per_cu - 3072) / ((size_t)wpb * 2 * pgrp * sizeof(long long)));   // (3 KB of margin: an exact third gets two per CU)
                if (per_cu > 2 && pg > fit) pg = fit;
            }
            if (pg > (H + pgrp - 1) / pgrp) pg = (H + pgrp - 1) / pgrp;
            if (pg > 64) pg = 64;                       // 4 waves x 2 columns x 64 groups x 32 rows x 8 B = 128 KB
            const int prow = pgrp * pg;
            size_t plds = (size_t)wpb * 2 * prow * sizeof(long long);
            if (plds < b_lds_min) plds = b_lds_min;
#define LAUNCH_BP(WIN) smk::fused_boxv_wta_vpl_kernel<WIN, 4, 0, false, true><<<gp, 64 * wpb, plds, ctx->stream>>>( \
        (const float *)T, tyshift, W, H, Dp, d_count, window, scale, a.disp0, a.dstep, outside,                      \
        d_first, xp, kout, n_pix, nullptr, 0, prow, merge_b)
            switch (window) {
            case 3: LAUNCH_BP(3); break;
            case 5: LAUNCH_BP(5); break;
            case 7: LAUNCH_BP(7); break;
            case 9: LAUNCH_BP(9); break;
            default: LAUNCH_BP(11); break;
            }
#undef LAUNCH_BP
        } else if (!keep_all) {
            switch (window) {        // static windows: the leaving row comes from registers
            case 3: LAUNCH_BW(3) break;
            case 5: LAUNCH_BW(5) break;
            case 7: LAUNCH_BW(7) break;
            case 9: LAUNCH_BW(9) break;
            case 11: LAUNCH_BW(11) break;
            case 13: LAUNCH_BW(13) break;
            case 15: LAUNCH_BW(15) break;
            case 17: LAUNCH_BW(17) break;
            default: LAUNCH_BW(0) break;
            }
        } else {                     // aggregated DSI out: 4 levels per lane, d_count % 4 == 0
#define LAUNCH_BS(WIN) smk::fused_boxv_wta_vpl_kernel<WIN, 4, 0, true><<<gb, 64 * wpb, stage_lds(ctx), ctx->stream>>>( \
        (const float *)T, tyshift, W, H, Dp, d_count, window, scale, a.disp0, a.dstep, outside,           \
        d_first, x_per_xcd, nullptr, 0, d_dsi, dsi_row_elems)
            switch (window) {
            case 3: LAUNCH_BS(3); break;
            case 5: LAUNCH_BS(5); break;
            case 7: LAUNCH_BS(7); break;
            case 9: LAUNCH_BS(9); break;
            case 11: LAUNCH_BS(11); break;
            case 13: LAUNCH_BS(13); break;
            case 15: LAUNCH_BS(15); break;
            case 17: LAUNCH_BS(17); break;
            default: LAUNCH_BS(0); break;
            }
#undef LAUNCH_BS
        }
#undef LAUNCH_BW
#undef LAUNCH_BT4
#undef LAUNCH_BTL
#undef LAUNCH_BT2
#undef LAUNCH_BTA
#undef LAUNCH_BT
    } else {
        int pix_per_blk = 256 / CH;
        unsigned gx = (unsigned)((W + pix_per_blk - 1) / pix_per_blk);
#define LAUNCH_B(SEG)                                                                        \
    smk::fused_boxv_wta_kernel<SEG><<<gx, 256, stage_lds(ctx), ctx->stream>>>(                            \
        (const float *)T, trow, W, H, d_count, window, scale, a.disp0, a.dstep, outside, d_first, kout)
        switch (CH) {
        case 2: LAUNCH_B(2); break;
        case 4: LAUNCH_B(4); break;
        case 8: LAUNCH_B(8); break;
        case 16: LAUNCH_B(16); break;
        default: LAUNCH_B(32); break;
        }
#undef LAUNCH_B
    }
    if ((n_chunks > 1 || merge_after) && !keep_all)
        smk::keys_min_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            kout, n_chunks, n_pix, (long long *)d_keys, merge ? 1 : 0);
    if (prof) {
        HIP_TRY(hipEventRecord(ev[2], ctx->stream));
        ctx->ev_used++;
    }
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_fused_box_wta_keys(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                             size_t img_pitch, const smhip_match_params *p, int window,
                             int d_first, int d_count, int64_t *d_keys)
{
    if (!d_keys)
        return fail(SMHIP_EINVAL, "null key buffer");
    return fused_impl(ctx, d_ref, d_mtc, img_pitch, p, window, d_first, d_count, d_keys, nullptr, 0);
}

int smhip_fused_box_wta_keys_merge(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc,
                                   size_t img_pitch, const smhip_match_params *p, int window,
                                   int d_first, int d_count, int64_t *d_keys)
{
    if (!d_keys)
        return fail(SMHIP_EINVAL, "null key buffer");
    return fused_impl(ctx, d_ref, d_mtc, img_pitch, p, window, d_first, d_count, d_keys, nullptr, 0, true);
}

int smhip_fused_box_dsi(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                        const smhip_match_params *p, int window, int d_first, int d_count,
                        float *d_cost, size_t row_elems)
{
    if (!p)
        return fail(SMHIP_EINVAL, "null argument");
    int rc = check_params(p, d_ref, d_mtc, img_pitch);
    if (rc) return rc;
    if (d_count < 1)
        return fail(SMHIP_EINVAL, "disparity slab needs at least one level (d_count=%d)", d_count);
    rc = check_dsi(d_cost, p->width, p->height, d_count, row_elems);
    if (rc) return rc;
    return fused_impl(ctx, d_ref, d_mtc, img_pitch, p, window, d_first, d_count, nullptr, d_cost, row_elems);
}

int smhip_min_pad_wta_keys(smhip_ctx *ctx, const float *d_cost, size_t row_elems,
                           const smhip_match_params *p, int min_window, int d_first, int d_count,
                           int64_t *d_keys)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!p || !d_keys)
        return fail(SMHIP_EINVAL, "null argument");
    if (d_first < 0 || d_count < 1 || d_first + d_count > p->disp_n)
        return fail(SMHIP_EINVAL, "disparity slab [%d,%d) outside [0,%d)", d_first, d_first + d_count, p->disp_n);
    if (p->disp_den != 1)
        return fail(SMHIP_EINVAL, "min filter + WTA keys: disp_step < 1 not supported, use the stage calls");
    int w = p->width, h = p->height;
    rc = check_dsi(d_cost, w, h, d_count, row_elems);
    if (rc) return rc;
    if (walk_vec(d_count, row_elems, d_cost, nullptr, nullptr) != 4 || min_window % 2 != 1 || min_window < 3 ||
        min_window > 11)
        return fail(SMHIP_EINVAL, "min filter + WTA keys needs an odd window 3..11, D %% 4 == 0 and 16-byte aligned rows");
    smk::PadWtaArgs pw;
    pw.dstep = -p->frame_diff_sign * p->disp_num;
    pw.disp0 = -p->frame_diff_sign * p->disp_min + d_first * pw.dstep;
    pw.d_first = d_first;
    pw.outside = smhip_match_outside(p);
    pw.keys = (long long *)d_keys;
    if (d_count > 65535)
        return fail(SMHIP_EINVAL, "min filter + WTA keys: more than 65535 levels in one call");
    // Tile and grid: 64-row segments (the keys of a segment live in LDS) of 64 staged columns; the level tiles are
    // dealt to G groups of workgroups so that a few rounds of the chip are filled (each group writes a key image,
    // keys_min_kernel merges them).  Round 2 had to shorten the segments of a 1080-row frame to 32 rows to get enough
    // workgroups (1.46x the volume read); with the groups 64 rows stay (1.29x): 1920x1080x256 0.66 -> 0.61 ms.
    // 128 staged columns (1024 threads, one workgroup per CU, 1.20x) read less and take longer: 0.68 ms -- the kernel
    // is bound by its barriers and LDS reads, not by the bytes (profiles/r03_exp2.log).  SMHIP_OPT_TSEG forces a segment
    // length, SMHIP_OPT_ITER_GROUPS the number of groups, SMHIP_OPT_MULTI_TXT = 128 the wide tile.
    const int txs = ctx->opt[SMHIP_OPT_MULTI_TXT] == 128 ? 128 : 64;
    const int txo = txs - (min_window - 1);
    int seg = smk::MW_SEG;
    if (ctx->opt[SMHIP_OPT_TSEG] >= 8 && ctx->opt[SMHIP_OPT_TSEG] <= smk::MW_SEG)
        seg = ctx->opt[SMHIP_OPT_TSEG];
    const int n_tiles = (d_count + 31) / 32;
    const long wgs = (long)((w + txo - 1) / txo) * ((h + seg - 1) / seg);
    int G = 1;
    while (G < n_tiles && G < 16 && wgs * G < (txs == 128 ? 1024 : 2048))
        G++;
    if (ctx->opt[SMHIP_OPT_ITER_GROUPS] >= 1)
        G = ctx->opt[SMHIP_OPT_ITER_GROUPS] < n_tiles ? ctx->opt[SMHIP_OPT_ITER_GROUPS] : n_tiles;
    if (G > 16) G = 16;
    const int tpg = (n_tiles + G - 1) / G;
    G = (n_tiles + tpg - 1) / tpg;
    const size_t n_pix = (size_t)w * h;
    if (G > 1) {
        void *part = nullptr;
        rc = ws_get(ctx, WS_KEYS, n_pix * G * sizeof(long long), &part);
        if (rc) return rc;
        pw.keys = (long long *)part;
    }
    dim3 grid((w + txo - 1) / txo, (h + seg - 1) / seg, G);
    if (grid.y > 65535)
        return fail(SMHIP_EINVAL, "min filter + WTA keys: more than %d rows", 65535 * seg);
#define LAUNCH_MW(WIN)                                                                                              \
    do {                                                                                                            \
        if (txs == 128) smk::min_wta_tile_kernel<WIN, 128><<<grid, 1024, stage_lds(ctx), ctx->stream>>>(            \
            d_cost, w, d_count, h, row_elems, seg, pw, tpg, n_pix);                                                 \
        else smk::min_wta_tile_kernel<WIN, 64><<<grid, 512, stage_lds(ctx), ctx->stream>>>(                         \
            d_cost, w, d_count, h, row_elems, seg, pw, tpg, n_pix);                                                 \
    } while (0)
    switch (min_window) {
    case 3: LAUNCH_MW(3); break;
    case 5: LAUNCH_MW(5); break;
    case 7: LAUNCH_MW(7); break;
    case 9: LAUNCH_MW(9); break;
    default: LAUNCH_MW(11); break;
    }
    if (G > 1)
        smk::keys_min_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            pw.keys, G, n_pix, (long long *)d_keys);
#undef LAUNCH_MW
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_pad_wta_keys(smhip_ctx *ctx, float *d_cost, size_t row_elems, const smhip_match_params *p,
                       int d_first, int d_count, int64_t *d_keys)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!p || !d_keys)
        return fail(SMHIP_EINVAL, "null argument");
    if (d_first < 0 || d_count < 1 || d_first + d_count > p->disp_n)
        return fail(SMHIP_EINVAL, "disparity slab [%d,%d) outside [0,%d)", d_first, d_first + d_count, p->disp_n);
    if (p->disp_den != 1)
        return fail(SMHIP_EINVAL, "PadCosts + WTA keys: disp_step < 1 not supported, use the stage calls");
    int w = p->width, h = p->height;
    rc = check_dsi(d_cost, w, h, d_count, row_elems);
    if (rc) return rc;
    // the slab is a DSI of its own whose first level has disparity disp_min + d_first * disp_num
    smhip_match_params ps = *p;
    ps.disp_min = p->disp_min + d_first * p->disp_num;
    ps.disp_n = d_count;
    rc = smhip_pad_costs(ctx, d_cost, row_elems, &ps);
    if (rc) return rc;
    size_t n_pix = (size_t)w * h;
    void *win = nullptr;
    rc = ws_get(ctx, WS_KEYS, n_pix * 8, &win);
    if (rc) return rc;
    int32_t *disp = (int32_t *)win;
    float *best = (float *)win + n_pix;
    rc = smhip_wta(ctx, d_cost, w, h, d_count, row_elems, disp, best);
    if (rc) return rc;
    dim3 grid((w + 31) / 32, (h + 31) / 32);
    smk::keys_encode_t_kernel<<<grid, 256, stage_lds(ctx), ctx->stream>>>(disp, best, w, h, d_first, (long long *)d_keys);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

// Rows per workgroup of the keys kernels of smhip_iter_px.h: IPX_SEG at most (the keys of a segment live in LDS),
// shorter where the frame would give too few workgroups.
static int iter_keys_seglen(const smhip_ctx *ctx, int w, int h, int op, int K)
{
    const int txi = smk::iter_px_tile_pixels(op, K);
    int seg = smk::IPX_SEG;
    while (seg > 16 && (long)((w + txi - 1) / txi) * ((h + seg - 1) / seg) < 512)
        seg >>= 1;
    if (ctx->opt[SMHIP_OPT_TSEG] >= 8 && ctx->opt[SMHIP_OPT_TSEG] <= smk::IPX_SEG)
        seg = ctx->opt[SMHIP_OPT_TSEG];
    return seg;
}

// Launch of a keys-sink kernel of smhip_iter_px.h: the level tiles are dealt to G workgroup groups (blockIdx.y);
// with G > 1 every group writes a key image of its own and keys_min_kernel merges them.
static int launch_iter_keys(smhip_ctx *ctx, int op, int K, int src, int nbk, bool sd, smk::IterPxArgs &g, int64_t *d_keys)
{
    const int w = g.W, h = g.H, lv = smk::iter_px_levels(src, smk::IPX_KEYS), txi = smk::iter_px_tile_pixels(op, K);
    const int n_tiles = (g.D + lv - 1) / lv;
    const int seg = iter_keys_seglen(ctx, w, h, op, K);
    const unsigned gx = (w + txi - 1) / txi, gz = (h + seg - 1) / seg;
    // enough workgroups for a few full rounds of the chip (two 512-thread or one 1024-thread workgroup per CU)
    int G = 1;
    const long per_round = lv == 16 ? 512 : 256;
    while (G < n_tiles && G < 16 && (long)gx * gz * G < 4 * per_round)
        G++;
    if (ctx->opt[SMHIP_OPT_ITER_GROUPS] >= 1)
        G = ctx->opt[SMHIP_OPT_ITER_GROUPS] < n_tiles ? ctx->opt[SMHIP_OPT_ITER_GROUPS] : n_tiles;
    if (G > 16) G = 16;
    g.tiles_per_group = (n_tiles + G - 1) / G;
    G = (n_tiles + g.tiles_per_group - 1) / g.tiles_per_group;     // no empty group
    g.seglen = seg;
    g.direct = 0;
    unsigned gz_direct = 0;
    // Direct mode: one level tile per workgroup and a key image per tile (merged by keys_min_kernel), the LDS key
    // arrays a ring that is flushed every 32 rows -- so a segment can be hundreds of rows long and the pipeline's
    // fill (2 RL + 1 row steps per segment: 81 steps for 64 rows of four binomial iterations) is paid once per
    // segment of 128 .. 512 rows.  Costs 16 bytes per pixel and level tile of key traffic.  Measured, 3840x2160: all
    // 512 levels binomial x4 11.6 -> 9.7 ms, x8 20.2 -> 17.9, diffusion x6 10.0 -> 8.7, x10 14.8 -> 13.8; levels
    // 256..319 binomial x4 1.64 -> 1.50; 1920x1080x256 binomial x4 1.87 -> 1.48 (profiles/r03_iter_direct.log).
    // The library's choice from 128 rows on; SMHIP_OPT_ITER_DIRECT 1 / 0 forces / forbids it.
    const bool direct_ok = n_tiles >= 2 && n_tiles <= 64;
    const bool direct = direct_ok && (ctx->opt[SMHIP_OPT_ITER_DIRECT] < 0 ? h >= 128 : ctx->opt[SMHIP_OPT_ITER_DIRECT] != 0);
    if (direct) {
        // segments: as long as possible while the grid still fills at least eight rounds of the chip evenly
        int dseg = ctx->opt[SMHIP_OPT_TSEG] >= 64 ? ctx->opt[SMHIP_OPT_TSEG] : 0;
        if (!dseg) {
            long S = 1;
            while ((long)gx * n_tiles * S < 8 * per_round && (h + S) / (S + 1) >= 128)
                S++;
            dseg = (int)((h + S - 1) / S);
            if (dseg > 512) dseg = 512;                // (longer segments measured slower: 768 rows 4 % behind 512)
            dseg = (dseg + 31) / 32 * 32;
        }
        g.direct = 1;
        g.tiles_per_group = 1;
        g.seglen = dseg;
        G = n_tiles;
        gz_direct = (unsigned)((h + dseg - 1) / dseg);
    }
    const size_t n_pix = (size_t)w * h;
    g.part_stride = n_pix;
    g.pw.keys = (long long *)d_keys;
    if (G > 1) {
        void *part = nullptr;
        int rc = ws_get(ctx, WS_KEYS, n_pix * G * sizeof(long long), &part);
        if (rc) return rc;
        g.pw.keys = (long long *)part;
    }
    if (gz > 65535)
        return fail(SMHIP_EINVAL, "iterated fused path: more than %d rows", 65535 * seg);
    if (!smk::launch_iter_px(op, K, src, smk::IPX_KEYS, nbk, sd, g, gx, (unsigned)G, direct ? gz_direct : gz, ctx->stream))
        return fail(SMHIP_EINVAL, "iterated fused path: no kernel for %d iterations", K);
    if (G > 1)
        smk::keys_min_kernel<<<(unsigned)((n_pix + 255) / 256), 256, stage_lds(ctx), ctx->stream>>>(
            g.pw.keys, G, n_pix, (long long *)d_keys);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

// Launch of a DSI-sink kernel of smhip_iter_px.h (source: raw costs or a resident DSI)
static int launch_iter_dsi(smhip_ctx *ctx, int op, int K, int src, int nbk, bool sd, smk::IterPxArgs &g)
{
    const int lv = smk::iter_px_levels(src, smk::IPX_DSI), txi = smk::iter_px_tile_pixels(op, K);
    int tseg = tile_seglen(ctx, g.W, g.H, g.D);
    if ((size_t)g.row_elems * g.H * sizeof(float) >= ((size_t)1 << 30) && ctx->opt[SMHIP_OPT_TSEG] < 0 && g.H >= 512)
        tseg = 256;
    g.seglen = tseg;
    const unsigned gz = (g.H + tseg - 1) / tseg, gy = (g.D + lv - 1) / lv;
    if (gz > 65535 || gy > 65535)
        return fail(SMHIP_EINVAL, "iterated fused path: grid too large");
    if (!smk::launch_iter_px(op, K, src, smk::IPX_DSI, nbk, sd, g, (g.W + txi - 1) / txi, gy, gz, ctx->stream))
        return fail(SMHIP_EINVAL, "iterated fused path: no kernel for %d iterations", K);
    HIP_TRY(hipGetLastError());
    return SMHIP_OK;
}

int smhip_fused_iter_wta_keys(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                              const smhip_match_params *p, int aggr_fn, int n_iter, float lambda, float beta,
                              int d_first, int d_count, int64_t *d_keys)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = check_params(p, d_ref, d_mtc, img_pitch);
    if (rc) return rc;
    if (!d_keys)
        return fail(SMHIP_EINVAL, "null key buffer");
    if (d_first < 0 || d_count < 1 || d_first + d_count > p->disp_n)
        return fail(SMHIP_EINVAL, "disparity slab [%d,%d) outside [0,%d)", d_first, d_first + d_count, p->disp_n);
    if (aggr_fn < 2 || aggr_fn > 4)
        return fail(SMHIP_EINVAL, "iterated fused path: aggr_fn %d is not binomial (2), diffusion (3) or membrane (4)", aggr_fn);
    if (n_iter < 0)
        return fail(SMHIP_EINVAL, "negative iteration count");
    if (p->disp_den != 1)
        return fail(SMHIP_EINVAL, "iterated fused path: disp_step < 1 not supported, use the stage calls");
    if (d_count > 65535)
        return fail(SMHIP_EINVAL, "iterated fused path: more than 65535 levels in one call");
    const int W = p->width, H = p->height, nb = p->n_bands;
    const int dstep = -p->frame_diff_sign * p->disp_num;
    const bool ref_point = !(p->match_interval && p->match_interpolated);
    const bool img4 = nb == 4 && img_pitch % 4 == 0 && ((uintptr_t)d_ref & 3) == 0 && ((uintptr_t)d_mtc & 3) == 0;
    const int nbk = (nb == 1 && ref_point) ? 2 : (img4 ? 4 : 0);
    // the diffusion model with beta forced to 0 (StcAggregate.cpp:305); the membrane model keeps the stage calls
    const int op = aggr_fn == 2 ? smk::IPX_BINOMIAL : smk::IPX_DIFFUSION;
    const bool px_ok = (aggr_fn == 2 || aggr_fn == 3) && n_iter >= 1 && W % 4 == 0 && nbk != 0 && abs(dstep) <= 16 &&
                       ctx->opt[SMHIP_OPT_ITER_FUSED] != 0;
    const int maxk = smk::iter_px_max_k(op);
    const size_t re = (size_t)W * d_count;                          // row of the private slab DSI
    const size_t vol = re * (size_t)H;
    const bool several = px_ok && n_iter > maxk && d_count % 4 == 0;

    smk::IterPxArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.H = H; g.D = d_count;
    g.force_slow = ctx->opt[SMHIP_OPT_TAP_SLOW] == 1;
    g.lambda = lambda;
    g.acoef = 1 - lambda * (0.0f + 4);                              // StcDiffusion.cpp:84 with beta = 0
    g.pw.dstep = dstep;
    g.pw.disp0 = -p->frame_diff_sign * p->disp_min + d_first * dstep;
    g.pw.d_first = d_first;
    g.pw.outside = smhip_match_outside(p);
    auto fill_raw = [&](const smk::RawArgs &a) {
        g.a.rn = a.rmn; g.a.rx = a.rmx ? a.rmx : a.rmn;
        g.a.mn = a.mmn; g.a.mx = a.mmx ? a.mmx : a.mmn;
        g.a.pitch = img_pitch; g.a.W = W; g.a.H = H; g.a.D = d_count;
        g.a.disp0 = a.disp0; g.a.dstep = a.dstep; g.a.cutoff = a.cutoff; g.a.outside = a.outside;
        g.a.bt_inline = 0;
    };

    if (px_ok && n_iter <= maxk) {                                  // one kernel, no DSI
        smk::RawArgs a;
        rc = prepare_raw(ctx, d_ref, d_mtc, img_pitch, p, d_first, d_count, &a);
        if (rc) return rc;
        fill_raw(a);
        return launch_iter_keys(ctx, op, n_iter, smk::IPX_RAW, nbk, a.sd != 0, g, d_keys);
    }

    // a private DSI of the slab (two buffers: the iterated passes ping-pong; membrane keeps the raw costs too)
    const int n_buf = aggr_fn == 4 && n_iter > 0 ? 3 : 2;
    void *ws = nullptr;
    rc = ws_get(ctx, WS_TMP, vol * sizeof(float) * n_buf, &ws);
    if (rc) return rc;
    float *A = (float *)ws, *B = A + vol, *C0 = n_buf == 3 ? B + vol : nullptr;

    if (several) {
        // the first iterations behind the raw costs, the last in front of the WTA, plain passes in between
        smk::RawArgs a;
        rc = prepare_raw(ctx, d_ref, d_mtc, img_pitch, p, d_first, d_count, &a);
        if (rc) return rc;
        fill_raw(a);
        g.row_elems = re;
        float *cur = A, *oth = B;
        int left = n_iter;
        if (op == smk::IPX_BINOMIAL) {
            int K1 = 4, K2 = 4;
            if (n_iter < 8) { K1 = (n_iter + 1) / 2; K2 = n_iter - K1; }
            g.dst = A;
            rc = launch_iter_dsi(ctx, op, K1, smk::IPX_RAW, nbk, a.sd != 0, g);
            if (rc) return rc;
            const int mid = n_iter - K1 - K2;
            if (mid > 0) {
                int in_b = 0;
                rc = smhip_convolve_14641_run(ctx, A, B, W, H, d_count, re, mid, &in_b);
                if (rc) return rc;
                cur = in_b ? B : A;
            }
            left = K2;
        } else {
            // diffusion: passes of nearly equal depth, all in the pixels-on-lanes form (up to six iterations each)
            int passes = (n_iter + maxk - 1) / maxk;
            for (int i = 0; passes > 1; passes--, i++) {
                const int K = multi_depth(left, passes);
                g.src = cur; g.dst = i == 0 ? cur : oth;
                if (i == 0) {
                    rc = launch_iter_dsi(ctx, op, K, smk::IPX_RAW, nbk, a.sd != 0, g);
                } else {
                    rc = launch_iter_dsi(ctx, op, K, smk::IPX_DSI, nbk, false, g);
                    float *t = cur; cur = oth; oth = t;
                }
                if (rc) return rc;
                left -= K;
            }
        }
        g.src = cur; g.dst = nullptr;
        return launch_iter_keys(ctx, op, left, smk::IPX_DSI, nbk, false, g, d_keys);
    }

    // the stage calls on the private DSI: raw costs of the slab, the aggregator, PadCosts + WTA -> keys
    smhip_match_params ps = *p;
    ps.disp_min = p->disp_min + d_first * p->disp_num;
    ps.disp_n = d_count;
    rc = smhip_raw_costs(ctx, d_ref, d_mtc, img_pitch, &ps, A, re);
    if (rc) return rc;
    float *cur = A;
    if (n_iter > 0) {
        int in_b = 0;
        if (aggr_fn == 2) {
            rc = smhip_convolve_14641_run(ctx, A, B, W, H, d_count, re, n_iter, &in_b);
        } else {
            const float bt = aggr_fn == 4 ? beta : 0.0f;            // eDiffusion forces beta = 0 (StcAggregate.cpp:305)
            if (bt > 0.0f)
                HIP_TRY(hipMemcpyAsync(C0, A, vol * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            rc = smhip_diffusion_run(ctx, A, B, bt > 0.0f ? C0 : nullptr, W, H, d_count, re, lambda, bt, n_iter, &in_b);
        }
        if (rc) return rc;
        cur = in_b ? B : A;
    }
    return smhip_pad_wta_keys(ctx, cur, re, p, d_first, d_count, d_keys);
}

int smhip_profile_begin(smhip_ctx *ctx, int max_calls)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (max_calls < 1 || max_calls > 4096)
        return fail(SMHIP_EINVAL, "max_calls must be in [1, 4096]");
    if (ctx->ev)
        return fail(SMHIP_EINVAL, "profiling already active");
    ctx->ev = (hipEvent_t *)calloc((size_t)3 * max_calls, sizeof(hipEvent_t));
    if (!ctx->ev)
        return fail(SMHIP_ENOMEM, "out of host memory");
    for (int i = 0; i < 3 * max_calls; i++)
        HIP_TRY(hipEventCreate(&ctx->ev[i]));
    ctx->ev_cap = max_calls;
    ctx->ev_used = 0;
    return SMHIP_OK;
}

int smhip_profile_get(smhip_ctx *ctx, int first, int count, float *ms_pass_a, float *ms_pass_b)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->ev)
        return fail(SMHIP_EINVAL, "profiling not active");
    if (first < 0 || count < 0 || first + count > ctx->ev_used || !ms_pass_a || !ms_pass_b)
        return fail(SMHIP_EINVAL, "profiled calls [%d,%d) outside [0,%d)", first, first + count, ctx->ev_used);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < count; i++) {
        HIP_TRY(hipEventElapsedTime(&ms_pass_a[i], ctx->ev[3 * (first + i)], ctx->ev[3 * (first + i) + 1]));
        HIP_TRY(hipEventElapsedTime(&ms_pass_b[i], ctx->ev[3 * (first + i) + 1], ctx->ev[3 * (first + i) + 2]));
    }
    return SMHIP_OK;
}

int smhip_profile_end(smhip_ctx *ctx, int *n_calls, float *ms_pass_a, float *ms_pass_b)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->ev)
        return fail(SMHIP_EINVAL, "profiling not active");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float a = 0, b = 0;
    for (int i = 0; i < ctx->ev_used; i++) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, ctx->ev[3 * i], ctx->ev[3 * i + 1]));
        a += t;
        HIP_TRY(hipEventElapsedTime(&t, ctx->ev[3 * i + 1], ctx->ev[3 * i + 2]));
        b += t;
    }
    if (n_calls) *n_calls = ctx->ev_used;
    if (ms_pass_a) *ms_pass_a = a;
    if (ms_pass_b) *ms_pass_b = b;
    for (int i = 0; i < 3 * ctx->ev_cap; i++)
        (void)hipEventDestroy(ctx->ev[i]);
    free(ctx->ev);
    ctx->ev = nullptr;
    ctx->ev_cap = ctx->ev_used = 0;
    return SMHIP_OK;
}

int smhip_fused_box_wta(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                        const smhip_match_params *p, int window, int32_t *d_disp, float *d_best)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!p)
        return fail(SMHIP_EINVAL, "null parameters");
    if (!d_disp && !d_best)
        return fail(SMHIP_EINVAL, "no output requested");
    rc = check_params(p, d_ref, d_mtc, img_pitch);      // before any size is derived from p
    if (rc) return rc;
    size_t n_pix = (size_t)p->width * p->height;
    // keys live behind the per-chunk partials in the key workspace
    int d_count = p->disp_n;
    int n_chunks = d_count > 256 ? (d_count + 255) / 256 : 1;
    void *kws = nullptr;
    rc = ws_get(ctx, WS_KEYS, n_pix * (size_t)(n_chunks + 1) * sizeof(long long), &kws);
    if (rc) return rc;
    int64_t *keys = (int64_t *)kws + n_pix * (size_t)n_chunks;
    rc = smhip_fused_box_wta_keys(ctx, d_ref, d_mtc, img_pitch, p, window, 0, d_count, keys);
    if (rc) return rc;
    return smhip_keys_decode(ctx, keys, p->width, p->height, d_disp, d_best);
}

int smhip_fused_box_wta_noalloc(smhip_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_mtc, size_t img_pitch,
                                const smhip_match_params *p, int window, int32_t *d_disp, float *d_best)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->scratch)
        return fail(SMHIP_EINVAL, "no static scratch on device %d (device index above %d)", ctx->device,
                    SMHIP_MAX_DEVICES - 1);
    // the slab, the interval images and the keys of this call live in the device's static scratch; the
    // kernels and their order are those of smhip_fused_box_wta
    ctx->arena_on = true;
    ctx->arena_used = 0;
    for (int i = 0; i < WS_COUNT; i++) {
        ctx->arena_ptr[i] = nullptr;
        ctx->arena_bytes[i] = 0;
    }
    rc = smhip_fused_box_wta(ctx, d_ref, d_mtc, img_pitch, p, window, d_disp, d_best);
    ctx->arena_on = false;
    return rc;
}

}  // extern "C"
