// Micro-benchmark with inline asm: exact per-instruction wave64 issue rate on gfx950 (8 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define BODY(ASM) \
    for (int it = 0; it < ITER; ++it) { \
        asm volatile(ASM : "+v"(a0) : "v"(a1), "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(a2), "v"(c)); \
        asm volatile(ASM : "+v"(a2) : "v"(a3), "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(a4), "v"(c)); \
        asm volatile(ASM : "+v"(a4) : "v"(a5), "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(a6), "v"(c)); \
        asm volatile(ASM : "+v"(a6) : "v"(a7), "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(a0), "v"(c)); }
#define KERNEL(NAME, ASM) __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) { \
    unsigned a0 = seed * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, c = seed | 0x01020304; \
    BODY(ASM) out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_min, "v_min_u32 %0, %0, %1")
KERNEL(k_max_i, "v_max_i32 %0, %0, %1")
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_align, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 8")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_sad, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_max3, "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
KERNEL(k_minsdwa, "v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
KERNEL(k_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_pkmax_i, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_pksub_i, "v_pk_sub_i16 %0, %0, %1")
KERNEL(k_cvt, "v_cvt_f32_ubyte1 %0, %1")
template <class F> void run(const char* name, F kern) {
    unsigned* d; hipMalloc(&d, 4 * 256 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<2048, 256>>>(d, 3); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<2048, 256>>>(d, 5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 2048.0 * 256 * ITER * 8;
    double rate = ops / ms / 1e9;
    printf("%-22s %7.3f ms %7.2f T lane-instr/s  (~%.1f cyc/wave-instr)\n", name, ms, rate, 256.0 * 4 * 64 * 2.4e9 / (rate * 1e12));
    hipFree(d);
}
int main() {
#define R(k) run(#k, k)
    R(k_add); R(k_sub); R(k_min); R(k_max_i); R(k_min3); R(k_max3); R(k_med3); R(k_pkmin); R(k_pkmax_i); R(k_pkadd); R(k_pksub_i);
    R(k_perm); R(k_align); R(k_alignbit); R(k_bfe); R(k_lshlor); R(k_andor); R(k_mad24); R(k_mullo); R(k_bcnt); R(k_sad); R(k_dot4);
    return 0;
}
