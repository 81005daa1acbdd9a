// orbm_kernels.hip.h -- hand-written gfx950 kernels of the ORB matcher primitives.
// Integer/bitwise bound: 256-bit XOR + popcount (v_bcnt) per pair, operands broadcast from LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbmk {

typedef unsigned long long u64;

__device__ __forceinline__ int ham256(const u64 a[4], u64 b0, u64 b1, u64 b2, u64 b3) {
    return __popcll(a[0] ^ b0) + __popcll(a[1] ^ b1) + __popcll(a[2] ^ b2) + __popcll(a[3] ^ b3);
}

// (dist, idx) lexicographic insert into a sorted top-2
__device__ __forceinline__ void top2_insert(int d, int j, int& d0, int& j0, int& d1, int& j1) {
    if (d < d0 || (d == d0 && j < j0)) { d1 = d0; j1 = j0; d0 = d; j0 = j; }
    else if (d < d1 || (d == d1 && j < j1)) { d1 = d; j1 = j; }
}

// ------------------------------------------------------------------------------------------------
// k_knn2: dense brute-force 2-NN in Hamming space (Frame.cc:1440-1480, cv::BFMatcher knnMatch k=2).
// grid (ceil(q_stride/64), npairs), 256 threads.  The block stages the pair's train descriptors in LDS
// in chunks; lane = query, the 4 waves scan interleaved quarters of each chunk (all lanes of a wave read
// the same LDS address -> broadcast, conflict free); per-wave top-2 are merged through LDS at the end.
// ------------------------------------------------------------------------------------------------
#define KNN_CHUNK 1024
__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, int q_stride, const int* __restrict__ nq,
                                              const uint8_t* __restrict__ t, int t_stride, const int* __restrict__ nt,
                                              int* __restrict__ idx2, int* __restrict__ dist2) {
    __shared__ __attribute__((aligned(16))) u64 tr[KNN_CHUNK * 4];
    __shared__ int mrg[4][64][4];
    const int pair = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nQ = nq[pair], nT = nt[pair];
    const int qi = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= nQ) return;                      // whole block idle (uniform)
    u64 a[4] = {0, 0, 0, 0};
    if (qi < nQ) {
        const uint4* qp = (const uint4*)(q + ((size_t)pair * q_stride + qi) * 32);
        const uint4 lo = qp[0], hi = qp[1];
        a[0] = (u64)lo.x | ((u64)lo.y << 32); a[1] = (u64)lo.z | ((u64)lo.w << 32);
        a[2] = (u64)hi.x | ((u64)hi.y << 32); a[3] = (u64)hi.z | ((u64)hi.w << 32);
    }
    int d0 = 1 << 20, d1 = 1 << 20, j0 = -1, j1 = -1;
    const uint8_t* tb = t + (size_t)pair * t_stride * 32;
    for (int c0 = 0; c0 < nT; c0 += KNN_CHUNK) {
        const int cn = min(KNN_CHUNK, nT - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < cn * 2; i += 256)       // 16 B per thread per step, coalesced
            ((uint4*)tr)[i] = ((const uint4*)(tb + (size_t)c0 * 32))[i];
        __syncthreads();
        for (int j = wv; j < cn; j += 4) {
            const int d = ham256(a, tr[4 * j], tr[4 * j + 1], tr[4 * j + 2], tr[4 * j + 3]);
            const int jj = c0 + j;
            if (d < d0) { d1 = d0; j1 = j0; d0 = d; j0 = jj; }    // per wave jj ascends: strict < keeps the lower index
            else if (d < d1) { d1 = d; j1 = jj; }
        }
    }
    mrg[wv][lane][0] = d0; mrg[wv][lane][1] = j0; mrg[wv][lane][2] = d1; mrg[wv][lane][3] = j1;
    __syncthreads();
    if (wv == 0 && qi < nQ) {
        for (int w = 1; w < 4; ++w) {
            if (mrg[w][lane][1] >= 0) top2_insert(mrg[w][lane][0], mrg[w][lane][1], d0, j0, d1, j1);
            if (mrg[w][lane][3] >= 0) top2_insert(mrg[w][lane][2], mrg[w][lane][3], d0, j0, d1, j1);
        }
        const size_t o = ((size_t)pair * q_stride + qi) * 2;
        idx2[o] = j0; idx2[o + 1] = j1;
        dist2[o] = j0 < 0 ? -1 : d0; dist2[o + 1] = j1 < 0 ? -1 : d1;
    }
}

}  // namespace orbmk
