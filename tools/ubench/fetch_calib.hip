// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access shapes of the pyramid+FAST+blur pass, when the bytes
// really fetched are known?  MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide (16 B/lane, 1 KiB per wave) stream on gfx950 and
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below reads
// each of its bytes exactly once from a 1 GiB buffer (cold: far beyond L2 and the 256 MiB Infinity Cache) and sums them into
// one word so that nothing is optimised away.  Printed: the bytes each kernel requests and the distinct 64-B / 128-B granules
// it touches; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare (tools/calib.sh).
//   k_wide      wave = 1 KiB contiguous, 16 B per lane                         (the guide's reference shape)
//   k_rows160   10 lanes x 16 B = one 160-B row run, rows 768 B apart          (k_blur3's source window: 128 px + 16-px halo pieces)
//   k_rows512   32 lanes x 16 B = one 512-B row run, rows 768 B apart          (k_fast3's strip tile rows)
//   k_u8x8      8 B per lane at odd byte offsets, 6 bytes apart                (k_resize2's tap loads: overlapping unaligned dwordx2)
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__global__ __launch_bounds__(256) void k_wide(const uint4* __restrict__ p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// rows of `run` bytes (a multiple of 16) every `pitch` bytes; lane -> 16-B piece of a row, `run / 16` lanes per row
__global__ __launch_bounds__(256) void k_rows(const unsigned char* __restrict__ p, size_t nrows, int run, int pitch, int xoff, unsigned* out) {
    const int ppr = run / 16;
    unsigned acc = 0;
    const size_t npieces = nrows * ppr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npieces; i += (size_t)gridDim.x * 256) {
        const size_t row = i / ppr; const int pc = (int)(i - row * ppr);
        const uint4 v = *(const uint4*)(p + row * pitch + xoff + 16 * pc);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_u8x8(const unsigned char* __restrict__ p, size_t nloads, unsigned* out) {
    typedef u64 __attribute__((aligned(1))) u64a1;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nloads; i += (size_t)gridDim.x * 256) {
        const u64 v = *(const u64a1*)(p + 1 + 6 * i);
        acc += (unsigned)v ^ (unsigned)(v >> 32);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
static size_t granules(size_t nrows, int run, int pitch, int xoff, int g) {   // distinct g-byte granules touched by the row runs
    size_t total = 0, prevLast = (size_t)-1;
    for (size_t r = 0; r < nrows; ++r) {
        const size_t a = (r * pitch + xoff) / g, b = (r * pitch + xoff + run - 1) / g;
        total += b - a + 1 - (a == prevLast ? 1 : 0);
        prevLast = b;
    }
    return total * g;
}
int main() {
    const size_t N = (size_t)1 << 30;
    unsigned char* d; unsigned* o;
    if (hipMalloc(&d, N + 4096) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 1, N + 4096);
    (void)hipDeviceSynchronize();
    const int grid = 2048;
    {
        const size_t n16 = N / 16;
        hipLaunchKernelGGL(k_wide, dim3(grid), dim3(256), 0, 0, (const uint4*)d, n16, o);
        printf("k_wide    requested %zu B, distinct 64-B %zu B, 128-B %zu B\n", N, N, N);
    }
    {
        const int pitch = 768, run = 160, xoff = 112;                       // window starts 16 B left of a 128-px group: x = 128 k - 16
        const size_t nrows = N / pitch;
        hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, d, nrows, run, pitch, xoff, o);
        printf("k_rows160 requested %zu B, distinct 64-B %zu B, 128-B %zu B\n", nrows * run, granules(nrows, run, pitch, xoff, 64), granules(nrows, run, pitch, xoff, 128));
    }
    {
        const int pitch = 768, run = 512, xoff = 16;
        const size_t nrows = N / pitch;
        hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, d, nrows, run, pitch, xoff, o);
        printf("k_rows512 requested %zu B, distinct 64-B %zu B, 128-B %zu B\n", nrows * run, granules(nrows, run, pitch, xoff, 64), granules(nrows, run, pitch, xoff, 128));
    }
    {
        const size_t nloads = (N - 16) / 6;
        hipLaunchKernelGGL(k_u8x8, dim3(grid), dim3(256), 0, 0, d, nloads, o);
        printf("k_u8x8    requested %zu B, distinct 64-B %zu B, 128-B %zu B\n", nloads * 8, N, N);
    }
    (void)hipDeviceSynchronize();
    return 0;
}
