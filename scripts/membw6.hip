// membw6.hip -- does a write-then-read slab that fits the 256 MiB Infinity Cache beat HBM?
// Streams a 2 GiB volume as (write band, read band) pairs, the bands either spread over the
// whole volume (today's fused path: every byte goes to HBM and comes back) or all mapped onto
// one re-used band buffer of S MiB (an x-banded fused path).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void wr(f4 *p, size_t n4, float v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    f4 x = {v, v, v, v};
    for (; i < n4; i += st) p[i] = x;
}
__global__ void rd(const f4 *p, size_t n4, float *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    f4 a = {0, 0, 0, 0};
    for (; i < n4; i += st) a += __builtin_nontemporal_load(p + i);
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[0] = 1;
}
__global__ void rd_plain(const f4 *p, size_t n4, float *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    f4 a = {0, 0, 0, 0};
    for (; i < n4; i += st) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[0] = 1;
}
int main()
{
    const size_t total = (size_t)2 << 30;
    float *a, *o; CK(hipMalloc(&a, total)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int plain = 0; plain < 2; plain++)
    for (size_t S : {(size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)192 << 20, (size_t)256 << 20, (size_t)512 << 20}) {
        for (int reuse = 0; reuse < 2; reuse++) {
            size_t bands = total / S, n4 = S / 16;
            unsigned grid = 256 * 8;
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                for (size_t b = 0; b < bands; b++) {
                    f4 *p = (f4 *)((char *)a + (reuse ? 0 : b * S));
                    wr<<<grid, 256>>>(p, n4, 1.0f + b);
                    if (plain) rd_plain<<<grid, 256>>>(p, n4, o); else rd<<<grid, 256>>>(p, n4, o);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%s loads, band %4zu MiB, %s: %.3f ms for 2 GiB written + 2 GiB read  (%.0f GB/s)\n",
                   plain ? "plain" : "nt   ", S >> 20, reuse ? "one re-used band " : "bands spread out ", best,
                   2.0 * total / best / 1e6);
        }
    }
    return 0;
}
