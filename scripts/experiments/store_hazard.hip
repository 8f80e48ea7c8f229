// store_hazard.hip -- reproducer: buffer_store_dwordx4 whose first data VGPR is rewritten by the VALU instruction
// right behind it (gfx950).  Hand-placed instructions (inline asm, fixed registers), so nothing depends on the
// scheduler:   variant 0: SGPR soffset, rewrite immediately     1: SGPR soffset, s_nop 0 (one wait state) between
//              2: soffset 0 (offset in the VGPR), immediately    3: soffset 0, s_nop 0 between
// and two compiled loops (C0: SGPR soffset, C1: soffset 0) whose ISA (--save-temps) shows whether the compiler's
// hazard recogniser puts the wait state there itself.  Prints, per variant, the dwords that hold the value written
// AFTER the store was issued, and the lanes they sit in.
//   hipcc -O3 --offload-arch=gfx950 --save-temps -o store_hazard store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 64, WAVES = 4096;

template <int VAR> __global__ __launch_bounds__(64) void k(unsigned *out)
{
    const unsigned lane = threadIdx.x, wave = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)wave * STEPS * 256, 0, STEPS * 1024, 0x00020000);
    if (VAR >= 4) {                                 // compiled loops: v is updated in place right behind the store
        u32x4 v = {lane, lane + 0x100u, lane + 0x200u, lane + 0x300u};
#pragma unroll 1
        for (unsigned i = 0, soff = 0; i < STEPS; i++, soff += 1024) {
            if (VAR == 5) __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16 + soff, 0, 0);
            else          __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, __builtin_amdgcn_readfirstlane(soff), 0);
            v += 0x1000u;
        }
        return;
    }
    unsigned voff = lane * 16;
    for (unsigned i = 0, soff = 0; i < STEPS; i++, soff += 1024) {
        const unsigned s = __builtin_amdgcn_readfirstlane(soff), vo = VAR >= 2 ? voff + soff : voff;
        asm volatile("v_add_u32 v20, %0, %1\n\tv_add_u32 v21, 0x100, v20\n\tv_add_u32 v22, 0x200, v20\n\t"
                     "v_add_u32 v23, 0x300, v20\n\ts_nop 4\n\t" ::"v"(lane), "v"(i * 0x1000u) : "v20", "v21", "v22", "v23");
        if (VAR == 0) asm volatile("buffer_store_dwordx4 v[20:23], %0, %1, %2 offen\n\tv_add_u32 v20, 0x1000, v20" ::"v"(vo), "s"(rs), "s"(s) : "v20", "memory");
        if (VAR == 1) asm volatile("buffer_store_dwordx4 v[20:23], %0, %1, %2 offen\n\ts_nop 0\n\tv_add_u32 v20, 0x1000, v20" ::"v"(vo), "s"(rs), "s"(s) : "v20", "memory");
        if (VAR == 2) asm volatile("buffer_store_dwordx4 v[20:23], %0, %1, 0 offen\n\tv_add_u32 v20, 0x1000, v20" ::"v"(vo), "s"(rs) : "v20", "memory");
        if (VAR == 3) asm volatile("buffer_store_dwordx4 v[20:23], %0, %1, 0 offen\n\ts_nop 0\n\tv_add_u32 v20, 0x1000, v20" ::"v"(vo), "s"(rs) : "v20", "memory");
    }
}

template <int VAR> static void run(unsigned *d, std::vector<unsigned> &h)
{
    (void)hipMemset(d, 0xFF, h.size() * 4);
    k<VAR><<<WAVES, 64>>>(d);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, late = 0;
    int hist[64] = {0};
    for (size_t n = 0; n < h.size(); n++) {
        const unsigned c = n & 3, lane = (n >> 2) & 63, i = (n >> 8) % STEPS;
        const unsigned want = lane + 0x100u * c + 0x1000u * i;
        if (h[n] != want) { bad++; late += h[n] == want + 0x1000u; hist[lane]++; }
    }
    printf("variant %d: %ld of %zu dwords wrong, %ld of them hold the value written after the store; lane:count", VAR, bad, h.size(), late);
    for (int l = 0; l < 64; l++) if (hist[l]) printf(" %d:%d", l, hist[l]);
    printf("\n");
}

int main()
{
    std::vector<unsigned> h((size_t)WAVES * STEPS * 256);
    unsigned *d; (void)hipMalloc(&d, h.size() * 4);
    for (int rep = 0; rep < 2; rep++) { run<0>(d, h); run<1>(d, h); run<2>(d, h); run<3>(d, h); run<4>(d, h); run<5>(d, h); }
    return 0;
}
