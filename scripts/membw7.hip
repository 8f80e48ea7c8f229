// membw7.hip -- is a 2 GiB write concurrent with a 2 GiB read (two streams, two buffers) faster
// than the two one after the other?  (Would overlapping pass A of frame i+1 with pass B of frame i pay?)
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
int main()
{
    const size_t S = (size_t)2 << 30, n4 = S / 16;
    float *a, *b, *o; CK(hipMalloc(&a, S)); CK(hipMalloc(&b, S)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, S)); CK(hipMemset(b, 0, S));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, j; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&j));
    for (unsigned grid : {256u * 4, 256u * 8, 256u * 16}) {
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s1));
                const int R = 8;
                for (int r = 0; r < R; r++) {
                    if (mode == 0) {
                        wr<<<grid, 256, 0, s1>>>((f4 *)a, n4, 1.0f);
                        rd<<<grid, 256, 0, s1>>>((const f4 *)b, n4, o);
                    } else {
                        CK(hipEventRecord(j, s1)); CK(hipStreamWaitEvent(s2, j, 0));
                        wr<<<grid, 256, 0, s1>>>((f4 *)a, n4, 1.0f);
                        rd<<<grid / 2, 256, 0, s2>>>((const f4 *)b, n4, o);
                        CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s1, j, 0));
                    }
                }
                CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / R < best) best = ms / R;
            }
            printf("grid %5u %s: %.3f ms per (2 GiB written + 2 GiB read)  %.0f GB/s\n", grid,
                   mode ? "concurrent (two streams)" : "one after the other     ", best, 2.0 * S / best / 1e6);
        }
    }
    return 0;
}
