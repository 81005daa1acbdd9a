// Micro-benchmark: the blur kernel's memory pattern without its arithmetic.  One wavefront walks R+6 source rows of a
// 256-byte-wide column strip (one dword per lane and row, 7 rows of loads in flight) and stores R rows.  Compared with the
// same bytes moved by a wide copy it tells whether k_blur2 is bound by its 4-byte-per-lane request stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R 32
__global__ __launch_bounds__(256) void k_rows(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int pitchDw, int h, int stripsPerRowBlock) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = task % stripsPerRowBlock, rb = task / stripsPerRowBlock;
    const int y0 = rb * R;
    if (y0 >= h) return;
    const size_t frameOff = (size_t)blockIdx.y * pitchDw * h;
    const unsigned* s = src + frameOff + strip * 64 + lane;
    unsigned* d = dst + frameOff + strip * 64 + lane;
    unsigned acc = 0, win[7];
    const int nrows = min(R, h - y0) + 6;
    unsigned nxt[7];
    for (int k = 0; k < 7; ++k) { int y = min(max(y0 - 3 + k, 0), h - 1); nxt[k] = s[(size_t)y * pitchDw]; }
    for (int r0 = 0; r0 < nrows; r0 += 7) {
        for (int k = 0; k < 7; ++k) win[k] = nxt[k];
        for (int k = 0; k < 7; ++k) { int y = min(max(y0 - 3 + r0 + 7 + k, 0), h - 1); nxt[k] = s[(size_t)y * pitchDw]; }
        for (int k = 0; k < 7; ++k) {
            const int r = r0 + k;
            if (r < nrows) {
                acc = acc * 3u + win[k];
                if (r >= 6) d[(size_t)(y0 + r - 6) * pitchDw] = acc;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_wide(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main() {
    const int w = 752, h = 480, frames = 512, pitchDw = 768 / 4;
    const size_t bytes = (size_t)pitchDw * 4 * h * frames;
    unsigned *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int strips = (w / 4 + 63) / 64, rblocks = (h + R - 1) / R, tasks = strips * rblocks;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rows, dim3((tasks + 3) / 4, frames), dim3(256), 0, 0, a, b, pitchDw, h, strips);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double moved = (double)frames * strips * 256.0 * ((double)rblocks * (R + 6) + (double)h);   // bytes read (with halo) + written
        printf("row stream (4 B/lane): %.3f ms, %.2f TB/s (level 0 of %d frames)\n", ms, moved / ms / 1e9, frames);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_wide, dim3(4096), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("wide copy (16 B/lane):  %.3f ms, %.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
