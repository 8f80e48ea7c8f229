/* c_abi_demo.c -- a plain C99 caller of libsmhip.so: what a C host (or any FFI) links against.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lstereomatch_amd/lib -lsmhip \
 *       -Wl,-rpath,$PWD/stereomatch_amd/lib -o c_abi_demo && ./c_abi_demo W H D
 *
 * Builds a deterministic gray image pair on the host, runs the fused raw-cost -> 9x9 box -> PadCosts
 * -> WTA path and the same thing through the stage calls, checks that the two disparity maps are
 * identical and prints a checksum of the map (tests/test_gpu_c_abi.py compares it with the oracle). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "smhip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_) {                                                               \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, smhip_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 24; }

int main(int argc, char **argv)
{
    int W = argc > 1 ? atoi(argv[1]) : 160, H = argc > 2 ? atoi(argv[2]) : 48, D = argc > 3 ? atoi(argv[3]) : 32;
    size_t n_pix = (size_t)W * H;
    uint8_t *ref = malloc(n_pix), *mtc = malloc(n_pix);
    int32_t *disp = malloc(n_pix * 4), *disp2 = malloc(n_pix * 4);
    uint32_t seed = 12345u;
    for (size_t i = 0; i < n_pix; i++) ref[i] = (uint8_t)lcg(&seed);
    for (int y = 0; y < H; y++)                     /* the matching frame: the reference shifted by 5 pixels */
        for (int x = 0; x < W; x++)
            mtc[(size_t)y * W + x] = ref[(size_t)y * W + (x + 5 < W ? x + 5 : W - 1)];

    smhip_ctx *ctx = NULL;
    CHECK(smhip_init(0, &ctx));
    void *d_ref, *d_mtc, *d_disp, *d_best, *d_cost;
    size_t row_elems = (size_t)W * D;
    CHECK(smhip_malloc(ctx, n_pix, &d_ref));
    CHECK(smhip_malloc(ctx, n_pix, &d_mtc));
    CHECK(smhip_malloc(ctx, n_pix * 4, &d_disp));
    CHECK(smhip_malloc(ctx, n_pix * 4, &d_best));
    CHECK(smhip_malloc(ctx, row_elems * H * 4, &d_cost));
    CHECK(smhip_memcpy_h2d(ctx, d_ref, ref, n_pix));
    CHECK(smhip_memcpy_h2d(ctx, d_mtc, mtc, n_pix));

    smhip_match_params p;
    memset(&p, 0, sizeof(p));
    p.width = W; p.height = H; p.n_bands = 1;
    p.disp_min = 0; p.disp_num = 1; p.disp_den = 1; p.disp_n = D; p.frame_diff_sign = 1;
    p.match_fn = 1; p.match_interp = 3; p.match_max = 1000; p.match_interval = 1; p.match_interpolated = 0;

    CHECK(smhip_fused_box_wta(ctx, d_ref, d_mtc, (size_t)W, &p, 9, d_disp, d_best));
    CHECK(smhip_memcpy_d2h(ctx, disp, d_disp, n_pix * 4));

    CHECK(smhip_raw_costs(ctx, d_ref, d_mtc, (size_t)W, &p, d_cost, row_elems));
    CHECK(smhip_box_filter(ctx, d_cost, W, H, D, row_elems, 9, 9, 1));
    CHECK(smhip_pad_costs(ctx, d_cost, row_elems, &p));
    CHECK(smhip_wta(ctx, d_cost, W, H, D, row_elems, d_disp, d_best));
    CHECK(smhip_memcpy_d2h(ctx, disp2, d_disp, n_pix * 4));

    if (memcmp(disp, disp2, n_pix * 4) != 0) {
        fprintf(stderr, "fused and staged disparity maps differ\n");
        return 2;
    }
    uint64_t sum = 0, hash = 1469598103934665603ull;   /* FNV-1a over the int32 map */
    for (size_t i = 0; i < n_pix; i++) {
        sum += (uint64_t)disp[i];
        for (int b = 0; b < 4; b++) {
            hash ^= (uint64_t)((disp[i] >> (8 * b)) & 0xFF);
            hash *= 1099511628211ull;
        }
    }
    printf("%dx%d D=%d sum=%llu fnv=%016llx\n", W, H, D, (unsigned long long)sum, (unsigned long long)hash);

    /* misuse comes back as a status + message, never an exit (reference: CudaUtilities.h:32-40 exits) */
    if (smhip_box_filter(ctx, d_cost, W, H, D, row_elems, 5, 7, 1) == 0) {
        fprintf(stderr, "xWidth != yWidth was accepted\n");
        return 3;
    }
    printf("error path: %s\n", smhip_last_error());

    CHECK(smhip_free(ctx, d_ref)); CHECK(smhip_free(ctx, d_mtc)); CHECK(smhip_free(ctx, d_disp));
    CHECK(smhip_free(ctx, d_best)); CHECK(smhip_free(ctx, d_cost));
    CHECK(smhip_shutdown(ctx));
    free(ref); free(mtc); free(disp); free(disp2);
    return 0;
}
