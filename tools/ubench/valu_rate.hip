// Micro-benchmark: sustained wave64 VALU issue rate on gfx950 for the integer ops the ORB kernels lean on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int OP> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 77;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
            if (OP == 1) a[i] = a[i] + (a[(i + 1) & 7] | 1);
            if (OP == 2) a[i] = min(a[i], a[(i + 1) & 7] + it);
            if (OP == 3) a[i] = min(a[i], min(a[(i + 1) & 7], a[(i + 2) & 7] ^ it));
            if (OP == 4) a[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, a[(i + 1) & 7] + it)));
            if (OP == 5) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x0c010c00u + it);
            if (OP == 6) a[i] = __builtin_amdgcn_alignbyte(a[i], a[(i + 1) & 7], it);
            if (OP == 7) a[i] = __builtin_amdgcn_udot4(a[i], a[(i + 1) & 7], it, false);
            if (OP == 8) a[i] = __builtin_amdgcn_sad_u8(a[i], a[(i + 1) & 7], it);
            if (OP == 9) a[i] = __popc(a[i] ^ a[(i + 1) & 7]) + it;
            if (OP == 10) a[i] = (a[i] & 0xff00ff) + (a[(i+1)&7] >> 8);
            if (OP == 11) a[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(us2, a[i]) - __builtin_bit_cast(us2, a[(i + 1) & 7] + it));
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (unsigned)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char* name, int perIter) {
    unsigned* d; hipMalloc(&d, 4 * 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<2048, 256>>>(d, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<2048, 256>>>(d, 5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 2048.0 * 256 * ITER * 8 * perIter;
    printf("%-28s %8.3f ms  %7.2f T lane-instr/s\n", name, ms, ops / ms / 1e9);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 1); run<1>("v_add_u32 (+or)", 2); run<2>("v_min_u32 (+add)", 2); run<3>("v_min3_u32 (+xor)", 2);
    run<4>("v_pk_min_u16 (+add)", 2); run<5>("v_perm_b32 (+add)", 2); run<6>("v_alignbyte_b32", 1); run<7>("v_dot4_u32_u8", 1);
    run<8>("v_sad_u8", 1); run<9>("v_bcnt (+xor)", 2); run<10>("and+lshr+add", 3); run<11>("v_pk_sub_u16 (+add)", 2);
    return 0;
}
