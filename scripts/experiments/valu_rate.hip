// valu_rate.hip -- issue rate of the VALU instructions pass A is made of (gfx950): cycles per
// wave-instruction at 1, 2, 4 waves per SIMD, independent and dependent streams.  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
// 8 independent chains a0..a7, one instruction each, repeated 8 times per loop trip = 64 instructions
#define KERNEL(NAME, INS)                                                                     \
__global__ void k_##NAME(unsigned *out, unsigned long long *cyc, int iters, unsigned seed)   \
{                                                                                             \
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3,      \
             a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 17 + 7;           \
    unsigned b = seed | 0x01010101u, c = seed ^ 0x3f800000u;                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int i = 0; i < iters; i++) {                                                         \
        REP8(asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n"    \
                          INS(%5) "\n" INS(%6) "\n" INS(%7)                                   \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                          : [b] "v"(b), [c] "v"(c));)                                                 \
    }                                                                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
}                                                                                             \
/* dependent: one chain */                                                                    \
__global__ void d_##NAME(unsigned *out, unsigned long long *cyc, int iters, unsigned seed)   \
{                                                                                             \
    unsigned a0 = threadIdx.x + seed;                                                         \
    unsigned b = seed | 0x01010101u, c = seed ^ 0x3f800000u;                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int i = 0; i < iters; i++) {                                                         \
        REP8(asm volatile(INS(%0) "\n" INS(%0) "\n" INS(%0) "\n" INS(%0) "\n" INS(%0) "\n"    \
                          INS(%0) "\n" INS(%0) "\n" INS(%0)                                   \
                          : "+v"(a0) : [b] "v"(b), [c] "v"(c));)                                      \
    }                                                                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;                                          \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
}

#define I_ADDF(r) "v_add_f32 " #r ", " #r ", %[b]"
#define I_MULF(r) "v_mul_f32 " #r ", " #r ", %[c]"
#define I_SUBF(r) "v_sub_f32 " #r ", " #r ", %[b]"
#define I_MAX3F(r) "v_max3_f32 " #r ", " #r ", %[b], %[c]"
#define I_MED3F(r) "v_med3_f32 " #r ", " #r ", %[b], %[c]"
#define I_MINF(r) "v_min_f32 " #r ", |" #r "|, %[c]"
#define I_SUBI(r) "v_sub_u32 " #r ", " #r ", %[b]"
#define I_SUBSDWA(r) "v_sub_u32_sdwa " #r ", " #r ", %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"
#define I_MAX3I(r) "v_max3_i32 " #r ", " #r ", %[b], 0"
#define I_MINI(r) "v_min_i32 " #r ", " #r ", %[c]"
#define I_CVTFI(r) "v_cvt_f32_i32 " #r ", " #r
#define I_CVTUB(r) "v_cvt_f32_ubyte1 " #r ", " #r
#define I_MOV(r) "v_mov_b32 " #r ", %[b]"
#define I_PKSUB16(r) "v_pk_sub_i16 " #r ", " #r ", %[b]"
#define I_PKMAX16(r) "v_pk_max_i16 " #r ", " #r ", %[b]"
#define I_SADU8(r) "v_sad_u8 " #r ", " #r ", %[b], %[c]"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %[b]"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 8, 8"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 1, %[b]"
#define I_MINFP(r) "v_min_f32 " #r ", " #r ", %[c]"
#define I_MAXFP(r) "v_max_f32 " #r ", " #r ", %[c]"
#define I_MAXI(r) "v_max_i32 " #r ", " #r ", %[c]"
#define I_ADD3(r) "v_add3_u32 " #r ", " #r ", %[b], %[c]"
#define I_SUBABS(r) "v_sub_f32_e64 " #r ", |" #r "|, %[c]"
#define I_ADDE64(r) "v_add_f32_e64 " #r ", " #r ", %[c]"
#define I_ADDNEG(r) "v_add_f32_e64 " #r ", -" #r ", |%[c]|"
#define I_OR(r) "v_or_b32 " #r ", " #r ", %[b]"
#define I_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %[b], vcc"
#define I_MULE64(r) "v_mul_f32_e64 " #r ", |" #r "|, %[c]"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %[b], %[c]"

KERNEL(add_f32, I_ADDF) KERNEL(mul_f32, I_MULF) KERNEL(sub_f32, I_SUBF) KERNEL(max3_f32, I_MAX3F)
KERNEL(med3_f32, I_MED3F) KERNEL(min_f32_abs, I_MINF) KERNEL(sub_u32, I_SUBI) KERNEL(sub_u32_sdwa, I_SUBSDWA)
KERNEL(max3_i32, I_MAX3I) KERNEL(min_i32, I_MINI) KERNEL(cvt_f32_i32, I_CVTFI) KERNEL(cvt_f32_ubyte1, I_CVTUB)
KERNEL(mov_b32, I_MOV) KERNEL(pk_sub_i16, I_PKSUB16) KERNEL(pk_max_i16, I_PKMAX16) KERNEL(sad_u8, I_SADU8)
KERNEL(min_f32, I_MINFP) KERNEL(max_f32, I_MAXFP) KERNEL(max_i32, I_MAXI) KERNEL(add3_u32, I_ADD3)
KERNEL(sub_f32_abs, I_SUBABS) KERNEL(add_f32_e64, I_ADDE64) KERNEL(add_f32_negabs, I_ADDNEG) KERNEL(or_b32, I_OR) KERNEL(cndmask, I_CNDMASK) KERNEL(mul_f32_abs, I_MULE64)
KERNEL(and_b32, I_AND) KERNEL(bfe_u32, I_BFE) KERNEL(lshl_add_u32, I_LSHLADD) KERNEL(fma_f32, I_FMA)

// packed f32 needs register pairs: written separately
#define KERNEL_PK(NAME, INS)                                                                  \
__global__ void k_##NAME(unsigned *out, unsigned long long *cyc, int iters, unsigned seed)   \
{                                                                                             \
    typedef float f2 __attribute__((ext_vector_type(2)));                                     \
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 * 9.f, a5 = a0 * 11.f, a6 = a0 * 13.f, a7 = a0 * 17.f; \
    f2 b = {1.0001f, 0.9999f};                                                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int i = 0; i < iters; i++) {                                                         \
        REP8(asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n"    \
                          INS(%5) "\n" INS(%6) "\n" INS(%7)                                   \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                          : [b] "v"(b));)                                                         \
    }                                                                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s.x + s.y);                  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
}                                                                                             \
__global__ void d_##NAME(unsigned *out, unsigned long long *cyc, int iters, unsigned seed)   \
{                                                                                             \
    typedef float f2 __attribute__((ext_vector_type(2)));                                     \
    f2 a0 = {(float)threadIdx.x, 1.f};                                                        \
    f2 b = {1.0001f, 0.9999f};                                                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int i = 0; i < iters; i++) {                                                         \
        REP8(asm volatile(INS(%0) "\n" INS(%0) "\n" INS(%0) "\n" INS(%0) "\n" INS(%0) "\n"    \
                          INS(%0) "\n" INS(%0) "\n" INS(%0)                                   \
                          : "+v"(a0) : [b] "v"(b));)                                              \
    }                                                                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(a0.x + a0.y);                \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
}
#define I_PKADD(r) "v_pk_add_f32 " #r ", " #r ", %[b]"
#define I_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %[b]"
KERNEL_PK(pk_add_f32, I_PKADD) KERNEL_PK(pk_mul_f32, I_PKMUL)

typedef void (*kfn)(unsigned *, unsigned long long *, int, unsigned);
struct Ent { const char *name; kfn indep, dep; };
#define E(NAME) { #NAME, k_##NAME, d_##NAME }
int main()
{
    Ent tab[] = { E(add_f32), E(mul_f32), E(sub_f32), E(fma_f32), E(max3_f32), E(med3_f32), E(sub_f32_abs), E(add_f32_e64), E(add_f32_negabs), E(mul_f32_abs), E(or_b32), E(cndmask), E(min_f32_abs), E(min_f32), E(max_f32), E(max_i32), E(add3_u32), E(pk_add_f32), E(pk_mul_f32),
                  E(sub_u32), E(sub_u32_sdwa), E(max3_i32), E(min_i32), E(cvt_f32_i32), E(cvt_f32_ubyte1), E(mov_b32),
                  E(pk_sub_i16), E(pk_max_i16), E(sad_u8), E(and_b32), E(bfe_u32), E(lshl_add_u32) };
    const int iters = 2000, ncu = 256;
    unsigned *out; unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4)); CK(hipMalloc(&cyc, (size_t)ncu * 8 * 8));
    unsigned long long *h = (unsigned long long *)malloc((size_t)ncu * 8 * 8);
    printf("cycles per wave-instruction seen by ONE wave (median over workgroups); SIMD rate = that / waves per SIMD\n");
    printf("ticks seen by one wave / wall ns per wave-instruction per SIMD\n%-16s %14s %14s %14s %14s | %14s\n", "instruction", "1 w/SIMD", "2", "4", "8", "dep 1w");
    for (auto &e : tab) {
        printf("%-16s", e.name);
        for (int wps : {1, 2, 4, 8, -1}) {
            kfn f = wps > 0 ? e.indep : e.dep;
            int nb = ncu * (wps > 0 ? wps : 1);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            f<<<nb, 256>>>(out, cyc, iters, 12345u);   // warm
            CK(hipEventRecord(e0));
            f<<<nb, 256>>>(out, cyc, iters, 12345u);   // 256 threads = one wave per SIMD; nb/ncu blocks per CU
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double ns_per = (double)ms * 1e6 / ((double)(wps > 0 ? wps : 1) * iters * 64);
            CK(hipMemcpy(h, cyc, (size_t)nb * 8, hipMemcpyDeviceToHost));
            // median
            for (int i = 1; i < nb; i++) { unsigned long long v = h[i]; int j = i - 1; while (j >= 0 && h[j] > v) { h[j + 1] = h[j]; j--; } h[j + 1] = v; }
            double per = (double)h[nb / 2] / ((double)iters * 64);
            if (wps == -1) printf(" | %6.2f/%5.2fns", per, ns_per); else printf(" %6.2f/%5.2fns", per, ns_per);
        }
        printf("\n");
    }
    return 0;
}
