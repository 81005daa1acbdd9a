// Semantics probes for gfx950 instructions the kernels use through inline asm (hipcc --offload-arch=gfx950 -O2 -o isa_probe isa_probe.hip):
//   v_ashr_pk_u8_i32 d, a, b, s   -- expected: d[7:0] = sat_u8(a >> s), d[15:8] = sat_u8(b >> s), upper half 0
//   v_pk_minimum3_f16 / v_pk_maximum3_f16 on halves 0..1023 taken as binary16 bit patterns -- expected: the integer minimum / maximum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); return 2; } } while (0)
__global__ void k_ashr(unsigned* o, const int* a, const int* b, unsigned sh, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsigned r; asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(sh));
    o[i] = r;
}
__global__ void k_m3(unsigned* o, const unsigned* a, const unsigned* b, const unsigned* c, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsigned r, s;
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(s) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    o[2 * i] = r; o[2 * i + 1] = s;
}
int main() {
    const int n = 1 << 20;
    std::vector<int> a(n), b(n); std::vector<unsigned> ua(n), ub(n), uc(n), o(2 * n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        a[i] = (rand() % 3000) - 500; b[i] = (rand() % 3000) - 500;
        if (i < 4096) { a[i] = i - 1024; b[i] = 2047 - i; }
        ua[i] = (rand() % 1024) | ((rand() % 1024) << 16); ub[i] = (rand() % 1024) | ((rand() % 1024) << 16); uc[i] = (rand() % 1024) | ((rand() % 1024) << 16);
        if (i < 65536) { ua[i] = (i & 255) | (((i >> 8) & 255) << 16); }
    }
    int *da, *db; unsigned *dua, *dub, *duc, *dout;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dua, n * 4)); CK(hipMalloc(&dub, n * 4)); CK(hipMalloc(&duc, n * 4)); CK(hipMalloc(&dout, 2 * n * 4));
    CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dua, ua.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dub, ub.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(duc, uc.data(), n * 4, hipMemcpyHostToDevice));
    int bad = 0;
    for (unsigned sh : {0u, 2u, 5u}) {
        hipLaunchKernelGGL(k_ashr, dim3(n / 256), dim3(256), 0, 0, dout, da, db, sh, n);
        CK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            auto sat = [](int v) { return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); };
            const unsigned want = sat(a[i] >> sh) | (sat(b[i] >> sh) << 8);
            if (o[i] != want && bad++ < 5) printf("ashr_pk sh=%u a=%d b=%d got %08x want %08x\n", sh, a[i], b[i], o[i], want);
        }
    }
    printf("v_ashr_pk_u8_i32: %s\n", bad ? "DIFFERS" : "as expected (low byte = first source, saturating, upper half zero)");
    int bad2 = 0;
    hipLaunchKernelGGL(k_m3, dim3(n / 256), dim3(256), 0, 0, dout, dua, dub, duc, n);
    CK(hipMemcpy(o.data(), dout, 2 * n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        unsigned mn = 0, mx = 0;
        for (int h = 0; h < 2; ++h) {
            const unsigned x = (ua[i] >> (16 * h)) & 0xFFFF, y = (ub[i] >> (16 * h)) & 0xFFFF, z = (uc[i] >> (16 * h)) & 0xFFFF;
            mn |= std::min(x, std::min(y, z)) << (16 * h); mx |= std::max(x, std::max(y, z)) << (16 * h);
        }
        if ((o[2 * i] != mn || o[2 * i + 1] != mx) && bad2++ < 5) printf("m3 %08x %08x %08x got %08x %08x want %08x %08x\n", ua[i], ub[i], uc[i], o[2 * i], o[2 * i + 1], mn, mx);
    }
    printf("v_pk_minimum3_f16 / v_pk_maximum3_f16 on halves 0..1023: %s\n", bad2 ? "DIFFERS" : "the integer minimum / maximum");
    return (bad || bad2) ? 1 : 0;
}
