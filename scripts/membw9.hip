// membw9.hip -- does the cache policy of loads / stores change what a streaming read + write kernel gets?
// Copy of N bytes, one dwordx4 per lane and step, through raw buffer loads / stores whose aux bits are template
// parameters (gfx950: 1 sc0, 2 nt, 16 sc1).  argv: GiB per buffer (default 8)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// each workgroup copies a contiguous chunk of `chunk` bytes (< 2 GiB) through its own buffer resource
template <int LP, int SP>
__global__ __launch_bounds__(256) void copy_k(const char *src, char *dst, size_t chunk)
{
    const char *s = src + (size_t)blockIdx.x * chunk;
    char *d = dst + (size_t)blockIdx.x * chunk;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)s, 0, (int)chunk, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)d, 0, (int)chunk, 0x00020000);
    for (unsigned off = threadIdx.x * 16; off < chunk; off += 256 * 16 * 4) {
        u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, LP);
        u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 256 * 16, 0, LP);
        u4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 2 * 256 * 16, 0, LP);
        u4 e = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 3 * 256 * 16, 0, LP);
        __builtin_amdgcn_raw_buffer_store_b128(a, rd, off, 0, SP);
        __builtin_amdgcn_raw_buffer_store_b128(b, rd, off + 256 * 16, 0, SP);
        __builtin_amdgcn_raw_buffer_store_b128(c, rd, off + 2 * 256 * 16, 0, SP);
        __builtin_amdgcn_raw_buffer_store_b128(e, rd, off + 3 * 256 * 16, 0, SP);
    }
}
template <int LP, int SP> void run(const char *name, const char *a, char *b, size_t n)
{
    const size_t chunk = 1 << 20;                   // 1 MiB per workgroup
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) copy_k<LP, SP><<<(unsigned)(n / chunk), 256>>>(a, b, chunk);
    CK(hipEventRecord(e0));
    const int R = 5;
    for (int i = 0; i < R; i++) copy_k<LP, SP><<<(unsigned)(n / chunk), 256>>>(a, b, chunk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    printf("%-28s %8.3f ms  %6.0f GB/s read+write\n", name, ms, 2.0 * n / ms / 1e6);
}
int main(int argc, char **argv)
{
    size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 8) << 30;
    char *a, *b; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n)); CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, n));
    for (int rep = 0; rep < 2; rep++) {
        run<0, 0>("load plain, store plain", a, b, n);
        run<2, 0>("load nt,    store plain", a, b, n);
        run<0, 2>("load plain, store nt", a, b, n);
        run<2, 2>("load nt,    store nt", a, b, n);
        run<2, 16>("load nt,    store sc1", a, b, n);
        run<2, 18>("load nt,    store sc1 nt", a, b, n);
        run<18, 18>("load sc1 nt, store sc1 nt", a, b, n);
        run<16, 16>("load sc1,   store sc1", a, b, n);
        run<3, 3>("load sc0 nt, store sc0 nt", a, b, n);
        run<17, 17>("load sc0 sc1, store sc0 sc1", a, b, n);
        run<19, 19>("load all,   store all", a, b, n);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-28s %8.3f ms  %6.0f GB/s read+write\n", "hipMemcpy d2d", ms, 2.0 * n / ms / 1e6);
    }
    return 0;
}
