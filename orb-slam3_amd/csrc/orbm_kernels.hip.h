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

__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {   // popcount(x) + acc in one instruction
#if __HIP_DEVICE_COMPILE__
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
#else
    return (unsigned)__builtin_popcount(x) + acc;
#endif
}

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
#if __HIP_DEVICE_COMPILE__
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return max(min(a, b), min(max(a, b), c));
#endif
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
                                              int* __restrict__ idx2, int* __restrict__ dist2, double ratio, uint8_t* __restrict__ good) {
    __shared__ __attribute__((aligned(16))) u64 tr[KNN_CHUNK * 4];
    __shared__ int mrg[4][64][4];
    const int pair = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    // top-2 as packed keys (distance << 22 | train index): unsigned order == (distance, index) lexicographic, which is the
    // tie rule (lower train index first); one v_min + one v_med3 per candidate.  t_stride < 2^22 is checked by the host.
    unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    const unsigned* a32 = (const unsigned*)a;
    const uint8_t* tb = t + (size_t)pair * t_stride * 32;
    for (int c0 = 0; c0 < nT; c0 += KNN_CHUNK) {
        const int cn = min(KNN_CHUNK, nT - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < cn * 2; i += 256)       // 16 B per thread per step, coalesced
            ((uint4*)tr)[i] = ((const uint4*)(tb + (size_t)c0 * 32))[i];
        __syncthreads();
        const uint4* tv = (const uint4*)tr;
        auto one = [&](int j) {
            const uint4 lo = tv[2 * j], hi = tv[2 * j + 1];     // same address in every lane: LDS broadcast
            unsigned d = bcnt_acc(a32[0] ^ lo.x, 0u);
            d = bcnt_acc(a32[1] ^ lo.y, d); d = bcnt_acc(a32[2] ^ lo.z, d); d = bcnt_acc(a32[3] ^ lo.w, d);
            d = bcnt_acc(a32[4] ^ hi.x, d); d = bcnt_acc(a32[5] ^ hi.y, d); d = bcnt_acc(a32[6] ^ hi.z, d);
            d = bcnt_acc(a32[7] ^ hi.w, d);                                           // 8 xor + 8 accumulating v_bcnt
            const unsigned key = (d << 22) | (unsigned)(c0 + j);
            k1 = umed3(k0, k1, key);           // second smallest of {k0 <= k1, key}
            k0 = min(k0, key);
        };
        int j = wv;
        for (; j + 12 < cn; j += 16) { one(j); one(j + 4); one(j + 8); one(j + 12); }
        for (; j < cn; j += 4) one(j);
    }
    __syncthreads();
    unsigned* mk = (unsigned*)mrg;                            // [4][64][2] keys
    mk[(wv * 64 + lane) * 2] = k0; mk[(wv * 64 + lane) * 2 + 1] = k1;
    __syncthreads();
    if (wv == 0 && qi < nQ) {
        for (int w = 1; w < 4; ++w)
            for (int e = 0; e < 2; ++e) {
                const unsigned key = mk[(w * 64 + lane) * 2 + e];
                k1 = umed3(k0, k1, key);
                k0 = min(k0, key);
            }
        const size_t o = ((size_t)pair * q_stride + qi) * 2;
        const bool h0 = k0 != 0xFFFFFFFFu, h1 = k1 != 0xFFFFFFFFu;
        idx2[o] = h0 ? (int)(k0 & 0x3FFFFFu) : -1; idx2[o + 1] = h1 ? (int)(k1 & 0x3FFFFFu) : -1;
        dist2[o] = h0 ? (int)(k0 >> 22) : -1; dist2[o + 1] = h1 ? (int)(k1 >> 22) : -1;
        if (good) good[o >> 1] = (h0 && h1 && (double)(float)(int)(k0 >> 22) < (double)(float)(int)(k1 >> 22) * ratio) ? 1 : 0;   // Frame.cc:1465
    }
}


// ------------------------------------------------------------------------------------------------
// k_knn2_mfma: the same dense 2-NN on the matrix cores.  A 1000 x 1000 x 256-bit Hamming table is O(n^2) integer work
// on 64 KB of operands -- compute bound, the one GEMM-shaped piece of the path -- and it is exact in int8:
//   ham(q, t) = popc(q) + sum_k t_k * (1 - 2 q_k)            (t_k, q_k the descriptor bits)
// so with train bits as 0/1 bytes in the A operand and query bits as +-1 bytes in the B operand, v_mfma_i32_32x32x32_i8
// accumulates ham - popc(q) for 32 train rows x 32 query columns; popc(q) is constant per column and is added at the end.
// Bit -> byte expansion is one v_and per operand dword: a word is rotated once so that bits 4h + e + 8i sit at 3 + e + 8i, and
// dword e of the fragment = rotated & (0x08080808 << e), i.e. bytes worth 0 or 2^(3+e); the query side carries the matching
// +-2^(6-e), so every product is +-512 and the accumulator is 512 * (ham - popc(q)).  Which k each (lane half h, byte) lands on
// does not matter: A and B use the same map.
// Selection without leaving the accumulator layout (column = lane & 31 = query, 16 train rows per lane): the accumulator
// starts at C = 2^17 + row, so the result IS a packed key (ham' + 256) << 9 | row whose unsigned order is (distance, train
// index) -- the reference's tie rule -- and one v_min + one v_med3 per distance keep the two smallest: no per-distance op
// besides those two.  Rows are tile-relative: tiles are walked from the last to the first and the kept keys move up by 32 per
// tile; the 9-bit row field holds 16 tiles, after which the two keys are widened to (distance + 256) << 19 | absolute index.
// A wave owns 64 queries (two B fragments sets) and streams the pair's train descriptors from L2; no LDS, no barrier.
// Limits (host-checked): t_stride <= KM_MAX_NT.
// ------------------------------------------------------------------------------------------------
#define KM_MAX_NT ((1 << 19) - 64)
typedef int km_i32x4 __attribute__((ext_vector_type(4)));
typedef int km_i32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_knn2_mfma(const uint8_t* __restrict__ q, int q_stride, const int* __restrict__ nq,
                                                   const uint8_t* __restrict__ t, int t_stride, const int* __restrict__ nt,
                                                   int* __restrict__ idx2, int* __restrict__ dist2, double ratio, uint8_t* __restrict__ good) {
#if __HIP_DEVICE_COMPILE__
    const int pair = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nQ = nq[pair], nT = nt[pair];
    const int q0 = blockIdx.x * 256 + wv * 64;
    if (q0 >= nQ) return;                                     // wave-uniform; the kernel has no barrier
    const int r = lane & 31, h = lane >> 5;
    // ---- query side: +-2^(6-e) bytes, built once
    km_i32x4 bf[2][8];
    int pq[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = q0 + n * 32 + r;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nQ) {
            const uint4* qp = (const uint4*)(q + ((size_t)pair * q_stride + qi) * 32);
            lo = qp[0]; hi = qp[1];
        }
        const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            pc += __popc(w[s]);
            const unsigned wp = w[s] >> (4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned y = (wp >> e) & 0x01010101u;              // the bit, per byte
                const unsigned mag = 64u >> e;
                bf[n][s][e] = (int)(mag * 0x01010101u + y * (256u - 2u * mag));   // bit ? -mag : +mag as int8 (no carries: <= 255 per byte)
            }
        }
        pq[n] = pc;
    }
    km_i32x16 cin;
#pragma unroll
    for (int i = 0; i < 16; ++i) cin[i] = (1 << 17) + (i & 3) + 8 * (i >> 2) + 4 * h;
    const unsigned SENT = 0x7FFFFE00u;                        // window key "no neighbour" (survives 16 x += 32)
    unsigned k0[2] = {SENT, SENT}, k1[2] = {SENT, SENT};      // two smallest window keys per query set
    unsigned G0[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, G1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};   // two smallest (distance+256) << 19 | train index
    const int rot = h ? 1 : 29;                               // rotate right: bits 4h + e + 8i of a word -> positions 3 + e + 8i
    const uint8_t* tb = t + (size_t)pair * t_stride * 32;
    const int ntile = (nT + 31) >> 5;
    uint4 nlo = make_uint4(0, 0, 0, 0), nhi = nlo;
    if (ntile > 0 && (ntile - 1) * 32 + r < nT) {
        const uint4* tp = (const uint4*)(tb + (size_t)((ntile - 1) * 32 + r) * 32);
        nlo = tp[0]; nhi = tp[1];
    }
    for (int tile = ntile - 1; tile >= 0; --tile) {
        const uint4 lo = nlo, hi = nhi;
        if (tile > 0) {                                       // the next tile is always a full one
            const uint4* tp = (const uint4*)(tb + (size_t)((tile - 1) * 32 + r) * 32);
            nlo = tp[0]; nhi = tp[1];
        }
        const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        km_i32x16 acc0 = cin, acc1 = cin;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned wp = __builtin_amdgcn_alignbit(w[s], w[s], rot);
            km_i32x4 a;
            a[0] = (int)(wp & 0x08080808u); a[1] = (int)(wp & 0x10101010u); a[2] = (int)(wp & 0x20202020u); a[3] = (int)(wp & 0x40404040u);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[1][s], acc1, 0, 0, 0);
        }
        k0[0] += 32u; k1[0] += 32u; k0[1] += 32u; k1[1] += 32u;          // kept keys become relative to this tile's first row
        if (tile * 32 + 32 <= nT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                k1[0] = umed3(k0[0], k1[0], (unsigned)acc0[i]); k0[0] = min(k0[0], (unsigned)acc0[i]);
                k1[1] = umed3(k0[1], k1[1], (unsigned)acc1[i]); k0[1] = min(k0[1], (unsigned)acc1[i]);
            }
        } else {                                              // the ragged last tile: rows past nT are not candidates
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ok = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h < nT;
                const unsigned ka = ok ? (unsigned)acc0[i] : SENT, kb = ok ? (unsigned)acc1[i] : SENT;
                k1[0] = umed3(k0[0], k1[0], ka); k0[0] = min(k0[0], ka);
                k1[1] = umed3(k0[1], k1[1], kb); k0[1] = min(k0[1], kb);
            }
        }
        if ((tile & 15) == 0) {                               // window done: widen its two keys to absolute train indices
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const unsigned base = (unsigned)tile * 32u;
                const unsigned g0 = k0[n] < 0x40000000u ? ((k0[n] >> 9) << 19) | (base + (k0[n] & 511u)) : 0xFFFFFFFFu;
                const unsigned g1 = k1[n] < 0x40000000u ? ((k1[n] >> 9) << 19) | (base + (k1[n] & 511u)) : 0xFFFFFFFFu;
                G1[n] = umed3(G0[n], G1[n], g0); G0[n] = min(G0[n], g0);
                G1[n] = umed3(G0[n], G1[n], g1); G0[n] = min(G0[n], g1);
                k0[n] = SENT; k1[n] = SENT;
            }
        }
    }
    // ---- the two lane halves saw disjoint train rows of the same query: merge, then half h stores query set h
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const unsigned o0 = (unsigned)__shfl_xor((int)G0[n], 32), o1 = (unsigned)__shfl_xor((int)G1[n], 32);
        G1[n] = umed3(G0[n], G1[n], o0); G0[n] = min(G0[n], o0);
        G1[n] = umed3(G0[n], G1[n], o1); G0[n] = min(G0[n], o1);
    }
    const unsigned f0 = h ? G0[1] : G0[0], f1 = h ? G1[1] : G1[0];
    const int pqs = h ? pq[1] : pq[0];
    const int qi = q0 + h * 32 + r;
    if (qi < nQ) {
        const size_t o = ((size_t)pair * q_stride + qi) * 2;
        const bool h0 = f0 != 0xFFFFFFFFu, h1 = f1 != 0xFFFFFFFFu;
        idx2[o] = h0 ? (int)(f0 & 0x7FFFFu) : -1; idx2[o + 1] = h1 ? (int)(f1 & 0x7FFFFu) : -1;
        const int d0 = h0 ? (int)(f0 >> 19) - 256 + pqs : -1, d1 = h1 ? (int)(f1 >> 19) - 256 + pqs : -1;
        dist2[o] = d0;
        dist2[o + 1] = d1;
        // Lowe's ratio as Frame.cc:1465 writes it: two neighbours and (float)d0 < (float)d1 * 0.7 evaluated in double
        if (good) good[(size_t)pair * q_stride + qi] = (h0 && h1 && (double)(float)d0 < (double)(float)d1 * ratio) ? 1 : 0;
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// k_grid_build: Frame::AssignFeaturesToGrid (Frame.cc:446-480).  One workgroup per frame.
// Stable order inside a cell = ascending keypoint index: sort keys (cell<<16 | index) with a bitonic
// sort in LDS, then every cell finds its start with a binary search.
// ------------------------------------------------------------------------------------------------
struct KpIn { float x, y, size, angle, response; int octave, class_id; };

// Counting sort form of the grid build (n <= GRID_CS_MAX): cell histogram in LDS -> exclusive scan over the 3072 cells ->
// scatter into a per-cell cursor -> each cell's short list sorted by keypoint index (insertion order of the reference,
// Frame.cc:446-480: i ascending) -> written out.  One pass over the keypoints instead of a 66-stage bitonic sort of n keys.
#define GRID_CS_MAX 8192
#define GRID_CELLS (64 * 48)
__device__ __forceinline__ int grid_build_counting(const KpIn* __restrict__ kp, int n, float min_x, float min_y, float inv_w, float inv_h,
                                                   int* __restrict__ gs, int* __restrict__ gi, unsigned char* smem) {
    int* hist = (int*)smem;                                    // [GRID_CELLS + 1] counts -> starts
    int* cur = hist + GRID_CELLS + 1;                          // [GRID_CELLS] fill cursors
    unsigned short* lgi = (unsigned short*)(cur + GRID_CELLS); // [n] indices, cell by cell
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = tid; c <= GRID_CELLS; c += 256) hist[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int px = (int)roundf((kp[i].x - min_x) * inv_w);              // PosInGrid: round, not floor (Frame.cc:888-889)
        const int py = (int)roundf((kp[i].y - min_y) * inv_h);
        if (px >= 0 && px < 64 && py >= 0 && py < 48) atomicAdd(&hist[px * 48 + py], 1);
    }
    __syncthreads();
    // exclusive scan: thread t owns cells [12 t, 12 t + 12) (3072 = 256 * 12)
    int loc[12], sum = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) { loc[k] = hist[12 * tid + k]; sum += loc[k]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < wv; ++w) base += s_wave[w];
    const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 12; ++k) { hist[12 * tid + k] = base; cur[12 * tid + k] = base; base += loc[k]; }
    if (tid == 0) hist[GRID_CELLS] = total;
    __syncthreads();
    for (int c = tid; c <= GRID_CELLS; c += 256) gs[c] = hist[c];
    for (int i = tid; i < n; i += 256) {
        const int px = (int)roundf((kp[i].x - min_x) * inv_w);
        const int py = (int)roundf((kp[i].y - min_y) * inv_h);
        if (px >= 0 && px < 64 && py >= 0 && py < 48) lgi[atomicAdd(&cur[px * 48 + py], 1)] = (unsigned short)i;
    }
    __syncthreads();
    for (int c = tid; c < GRID_CELLS; c += 256) {                            // restore insertion order inside every cell (lists are short)
        const int a = hist[c], b = hist[c + 1];
        for (int i = a + 1; i < b; ++i) {
            const unsigned short v = lgi[i];
            int j = i - 1;
            while (j >= a && lgi[j] > v) { lgi[j + 1] = lgi[j]; --j; }
            lgi[j + 1] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += 256) gi[i] = (int)lgi[i];
    return total;
}
__global__ __launch_bounds__(256) void k_grid_build_cs(const KpIn* __restrict__ kps, int n, float min_x, float min_y, float inv_w, float inv_h,
                                                       int* __restrict__ grid_start, int* __restrict__ grid_idx, int* __restrict__ placed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int total = grid_build_counting(kps, n, min_x, min_y, inv_w, inv_h, grid_start, grid_idx, gsm);
    if (threadIdx.x == 0) *placed = total;
}
__global__ __launch_bounds__(256) void k_grid_build_batch_cs(const KpIn* __restrict__ kps, const int* __restrict__ counts, int cap,
                                                             float min_x, float min_y, float inv_w, float inv_h,
                                                             int* __restrict__ grid_start, int* __restrict__ grid_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int frame = blockIdx.x;
    (void)grid_build_counting(kps + (size_t)frame * cap, min(counts[frame], cap), min_x, min_y, inv_w, inv_h,
                              grid_start + (size_t)frame * (GRID_CELLS + 1), grid_idx + (size_t)frame * cap, gsm);
}

__global__ __launch_bounds__(256) void k_grid_build(const KpIn* __restrict__ kps, int n, int n2, float min_x, float min_y,
                                                    float inv_w, float inv_h, int* __restrict__ grid_start,
                                                    int* __restrict__ grid_idx, int* __restrict__ placed) {
    extern __shared__ unsigned int keys[];
    const int tid = threadIdx.x;
    __shared__ int s_placed;
    if (tid == 0) s_placed = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += 256) {
        unsigned int key = 0xFFFFFFFFu;
        if (i < n) {
            const int px = (int)roundf((kps[i].x - min_x) * inv_w);          // PosInGrid: round, not floor (Frame.cc:888-889)
            const int py = (int)roundf((kps[i].y - min_y) * inv_h);
            if (px >= 0 && px < 64 && py >= 0 && py < 48) { key = ((unsigned)(px * 48 + py) << 16) | (unsigned)i; atomicAdd(&s_placed, 1); }
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned int a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const int np = s_placed;
    for (int i = tid; i < np; i += 256) grid_idx[i] = (int)(keys[i] & 0xFFFFu);
    for (int c = tid; c <= 64 * 48; c += 256) {
        const unsigned int target = (unsigned)c << 16;
        int lo = 0, hi = np;                                                  // first key >= target
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < target) lo = mid + 1; else hi = mid; }
        grid_start[c] = lo;
    }
    if (tid == 0) *placed = np;
}

// ------------------------------------------------------------------------------------------------
// k_window: Frame::GetFeaturesInArea (Frame.cc:784-871) + DescriptorDistance for a batch of windows.
// One wavefront per window.  Cells of one grid column are contiguous in the CSR (cell = ix*48+iy), so for every
// ix the lanes sweep ONE index range in the reference's visiting order; survivors are compacted with a ballot
// (order preserved) and their 256-bit Hamming distance to the window's descriptor is taken on the spot.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                const float* __restrict__ uright, const int* __restrict__ gs,
                                                const int* __restrict__ gi, float min_x, float min_y, float inv_w, float inv_h,
                                                int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                const float* __restrict__ qr, const int* __restrict__ qminl,
                                                const int* __restrict__ qmaxl, const float* __restrict__ qur,
                                                const float* __restrict__ qer, const uint8_t* __restrict__ qdesc, int cap,
                                                int* __restrict__ out_cnt, int* __restrict__ out_idx, int* __restrict__ out_dist,
                                                int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minLevel = qminl[q], maxLevel = qmaxl[q];
    int cnt = 0;
    const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
    const int nMaxCellX = min(63, (int)ceilf((x - min_x + r) * inv_w));
    const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
    const int nMaxCellY = min(47, (int)ceilf((y - min_y + r) * inv_h));
    if (r >= 0 && nMinCellX < 64 && nMaxCellX >= 0 && nMinCellY < 48 && nMaxCellY >= 0) {
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        const uint4* qp = (const uint4*)(qdesc + (size_t)q * 32);
        const uint4 qlo = qp[0], qhi = qp[1];
        const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                          (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
        const float er = qer ? qer[q] : -1.f, ur = qur ? qur[q] : 0.f;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            const int j0 = gs[ix * 48 + nMinCellY], j1 = gs[ix * 48 + nMaxCellY + 1];
            for (int jb = j0; jb < j1; jb += 64) {
                const int j = jb + lane;
                bool ok = false;
                int k = 0;
                if (j < j1) {
                    k = gi[j];
                    const KpIn kp = kps[k];
                    ok = true;
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) ok = false;
                        if (maxLevel >= 0 && kp.octave > maxLevel) ok = false;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
                    if (ok && er >= 0.f && uright) {
                        const float urk = uright[k];
                        if (urk > 0 && fabsf(ur - urk) > er) ok = false;
                    }
                }
                const unsigned long long bal = __ballot(ok);
                if (ok) {
                    const int pos = cnt + __popcll(bal & lt);
                    if (pos < cap) {
                        const uint4* tp = (const uint4*)(desc + (size_t)k * 32);
                        const uint4 lo = tp[0], hi = tp[1];
                        const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                             (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                        out_idx[(size_t)q * cap + pos] = k;
                        out_dist[(size_t)q * cap + pos] = d;
                    } else *overflow = 1;
                }
                cnt += __popcll(bal);
            }
        }
    }
    if (lane == 0) out_cnt[q] = min(cnt, cap);
}

// ------------------------------------------------------------------------------------------------
// k_window_topk: the same windows, but only what the sequential claim replay of the projection searches can ever look at
// comes back: per window the WT_K candidates with the smallest (distance, position in the reference's visiting order) and
// the candidate count.  The replay takes the first (M4, M5) or the first two (M3: best and second) candidates that are not
// blocked in that order -- exactly the order in which `dist < bestDist` / `else if dist < bestDist2` would have met them --
// so unless every returned candidate of a longer list is blocked, the full [windows][capacity] lists never leave the GPU.
// key = dist << 40 | position << 20 | keypoint index; the running top-K lives in wave-uniform registers.
// ------------------------------------------------------------------------------------------------
#define WT_K 8
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o);
        const u64 w = (u64)lo | ((u64)hi << 32);
        v = w < v ? w : v;
    }
    return v;
}
__global__ __launch_bounds__(256) void k_window_topk(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                     const float* __restrict__ uright, const int* __restrict__ gs,
                                                     const int* __restrict__ gi, float min_x, float min_y, float inv_w, float inv_h,
                                                     int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                     const float* __restrict__ qr, const int* __restrict__ qminl,
                                                     const int* __restrict__ qmaxl, const float* __restrict__ qur,
                                                     const float* __restrict__ qer, const uint8_t* __restrict__ qdesc,
                                                     int* __restrict__ out_cnt, unsigned int* __restrict__ out_keys) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int minLevel = qminl[q], maxLevel = qmaxl[q];
    const u64 INV = ~0ull;
    u64 top[WT_K];
#pragma unroll
    for (int i = 0; i < WT_K; ++i) top[i] = INV;
    int cnt = 0;
    const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
    const int nMaxCellX = min(63, (int)ceilf((x - min_x + r) * inv_w));
    const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
    const int nMaxCellY = min(47, (int)ceilf((y - min_y + r) * inv_h));
    if (r >= 0 && nMinCellX < 64 && nMaxCellX >= 0 && nMinCellY < 48 && nMaxCellY >= 0) {
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        const uint4* qp = (const uint4*)(qdesc + (size_t)q * 32);
        const uint4 qlo = qp[0], qhi = qp[1];
        const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                          (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
        const float er = qer ? qer[q] : -1.f, ur = qur ? qur[q] : 0.f;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            const int j0 = gs[ix * 48 + nMinCellY], j1 = gs[ix * 48 + nMaxCellY + 1];
            for (int jb = j0; jb < j1; jb += 64) {
                const int j = jb + lane;
                bool ok = false;
                int k = 0;
                if (j < j1) {
                    k = gi[j];
                    const KpIn kp = kps[k];
                    ok = true;
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) ok = false;
                        if (maxLevel >= 0 && kp.octave > maxLevel) ok = false;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
                    if (ok && er >= 0.f && uright) {
                        const float urk = uright[k];
                        if (urk > 0 && fabsf(ur - urk) > er) ok = false;
                    }
                }
                const unsigned long long bal = __ballot(ok);
                u64 key = INV;
                if (ok) {
                    const int pos = cnt + __popcll(bal & lt);
                    const uint4* tp = (const uint4*)(desc + (size_t)k * 32);
                    const uint4 lo = tp[0], hi = tp[1];
                    const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                         (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                    key = ((u64)d << 40) | ((u64)pos << 20) | (u64)k;
                }
                cnt += __popcll(bal);
                if (bal == 0) continue;
                // merge this chunk into the running top-K: pull its minima one by one until one no longer beats the K-th
                for (int rnd = 0; rnd < WT_K; ++rnd) {
                    const u64 m = wave_min_u64(key);
                    if (m >= top[WT_K - 1]) break;               // INV included
                    if (key == m) key = INV;                     // keys are unique (position)
                    u64 c = m;
#pragma unroll
                    for (int i = 0; i < WT_K; ++i) { const u64 t = top[i]; const bool sw = c < t; top[i] = sw ? c : t; c = sw ? t : c; }
                }
            }
        }
    }
    if (lane == 0) {
        out_cnt[q] = cnt;
#pragma unroll
        for (int i = 0; i < WT_K; ++i)                            // the array position carries the (distance, visiting order) rank: 32 bits suffice
            out_keys[(size_t)q * WT_K + i] = top[i] == INV ? 0xFFFFFFFFu : (unsigned)((top[i] >> 40) << 20) | (unsigned)(top[i] & 0xFFFFFu);
    }
}

// ------------------------------------------------------------------------------------------------
// k_pairdist: Hamming distances for a job list (bucket joins of SearchByBoW / SearchForTriangulation_).
// job j: query row q1[j] of set 1 against rows idx2[l2[j] .. l2[j]+len[j]) of set 2, outputs at off[j]...
// One thread per output element (the job is found by binary search in the output offsets).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pairdist(const uint8_t* __restrict__ d1, const uint8_t* __restrict__ d2,
                                                  const int* __restrict__ idx2, int njobs, const int* __restrict__ jq,
                                                  const int* __restrict__ jl2, const int* __restrict__ joff, int total,
                                                  int* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int lo = 0, hi = njobs;                                                   // last job with joff <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (joff[mid] <= e) lo = mid; else hi = mid; }
    const int k2 = idx2[jl2[lo] + (e - joff[lo])];
    const uint4* p1 = (const uint4*)(d1 + (size_t)jq[lo] * 32);
    const uint4* p2 = (const uint4*)(d2 + (size_t)k2 * 32);
    const uint4 a0 = p1[0], a1 = p1[1], b0 = p2[0], b1 = p2[1];
    out[e] = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
             __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// ------------------------------------------------------------------------------------------------
// k_stereo: Frame::ComputeStereoMatches (Frame.cc:1027-1256) up to (not including) the median cut.
// One wavefront per left keypoint.  Candidates = right keypoints whose row band [floor(y-r), ceil(y+r)],
// r = 2*scale[octave], contains (int)vL (the reference's vRowIndices table, visited in iR order), octave within
// +-1 and u in [uL-maxD, uL]; best = min (distance, iR).  Then the 11x11 SAD slide over +-5 px on the two
// un-blurred pyramids, parabola refinement and the disparity gates, all in the reference's float arithmetic.
// ------------------------------------------------------------------------------------------------
struct StereoLevels { const uint8_t* L[12]; const uint8_t* R[12]; int pitchL[12], pitchR[12], wR[12]; float sf[12], isf[12]; };

// LV supplies the pyramids: sf(o), isf(o), L(level), R(level), pitchL(level), pitchR(level), wR(level)
// rowStart / rowIdx (optional): the reference's vRowIndices table (Frame.cc:1064-1083) as CSR over image rows -- the right
// keypoints whose band [floor(y - r), ceil(y + r)] covers a row; with it a left keypoint only looks at the ~20 keypoints of its
// row instead of all of them.  The winner is min (distance, iR) either way, so the order inside a row list does not matter.
// GS = lanes per left keypoint (64: one per wave; 16: four per wave -- the kernel is bound by its chain of dependent memory round trips
// at full occupancy, so the batched form quarters the number of waves that wait: 0.33 -> 0.21 ms per 256 pairs; 8 lanes: no further gain)
template <int GS, class LV>
__device__ __forceinline__ void stereo_body(const KpIn* __restrict__ kl, const uint8_t* __restrict__ dl, int nl,
                                            const KpIn* __restrict__ kr, const uint8_t* __restrict__ dr, int nr,
                                            const LV& lv, float mb, float mbf, float* __restrict__ uright,
                                            float* __restrict__ depth, int* __restrict__ bestSad, int iL,
                                            const int* __restrict__ rowStart = nullptr, const unsigned short* __restrict__ rowIdx = nullptr, int nrows = 0) {
    const int lane = threadIdx.x & (GS - 1), grp = (threadIdx.x & 63) / GS;
    const KpIn kpL = kl[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int rowL = (int)vL;
    const float minD = 0, maxD = mbf / mb;
    const float minU = uL - maxD, maxU = uL - minD;
    float ur_out = -1.0f, depth_out = -1.0f;
    int sad_out = -1;
    if (!(maxU < 0)) {
        const uint4* qp = (const uint4*)(dl + (size_t)iL * 32);
        const uint4 qlo = qp[0], qhi = qp[1];
        const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                          (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
        unsigned int best = 0xFFFFFFFFu;                                     // (dist << 16) | iR, strict < keeps the first
        float bestx = 0.f;                                                   // x of this lane's best candidate
        int c0 = 0, c1 = nr;
        if (rowStart) { if (rowL >= 0 && rowL < nrows) { c0 = rowStart[rowL]; c1 = rowStart[rowL + 1]; } else c1 = 0; }
        for (int b0 = c0; b0 < c1; b0 += GS) {
            const int ci = b0 + lane;
            const int iR = rowStart ? (ci < c1 ? (int)rowIdx[ci] : nr) : ci;
            if (iR < nr) {
                // the candidate's descriptor is requested together with its keypoint, not after the gates (the kernel is bound by its
                // chain of dependent memory round trips; a pair's right descriptors are 38 KB that stay in L2)
                const float rx = kr[iR].x, ry = kr[iR].y;
                const int ro = kr[iR].octave;
                const uint4* tp = (const uint4*)(dr + (size_t)iR * 32);
                const uint4 lo = tp[0], hi = tp[1];
                const float r = 2.0f * lv.sf(ro);
                const int maxr = (int)ceilf(ry + r), minr = (int)floorf(ry - r);
                if (rowL >= minr && rowL <= maxr && ro >= levelL - 1 && ro <= levelL + 1 && rx >= minU && rx <= maxU) {
                    const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                         (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                    const unsigned key = ((unsigned)d << 16) | (unsigned)iR;
                    if (key < best) { best = key; bestx = rx; }
                }
            }
        }
        const unsigned mine = best;
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
        const int bestDist = best == 0xFFFFFFFFu ? 100 : (int)(best >> 16);
        if (best != 0xFFFFFFFFu && bestDist < 100 && bestDist < 75) {        // < TH_HIGH to replace the init, < thOrbDist to go on
            // the winner's x comes from the lane that holds it (keys are unique: they carry the index), not from memory again
            unsigned long long who = __ballot(mine == best);
            if (GS < 64) who = (who >> (GS * grp)) & ((1ull << GS) - 1ull);   // this keypoint's lanes
            const float uR0 = __shfl(bestx, (int)__builtin_ctzll(who) + GS * grp);
            const float scaleFactor = lv.isf(levelL);
            const float scaleduL = roundf(kpL.x * scaleFactor), scaledvL = roundf(kpL.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, Lh = 5;
            const float iniu = scaleduR0 + Lh - w, endu = scaleduR0 + Lh + w + 1;
            if (!(iniu < 0 || endu >= (float)lv.wR(levelL))) {
                const uint8_t* IL = lv.L(levelL);
                const uint8_t* IR = lv.R(levelL);
                const int pl = lv.pitchL(levelL), pr = lv.pitchR(levelL);
                const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr = (int)scaleduR0;
                int sad[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) sad[k] = 0;
                for (int p = lane; p < 121; p += GS) {
                    const int dy = p / 11 - w, dx = p % 11 - w;
                    const unsigned vl = IL[(size_t)(cy + dy) * pl + cxl + dx];
                    const uint8_t* rr = IR + (size_t)(cy + dy) * pr + cxr + dx;
                    // the 11 right pixels rr[-5..5] as three dword loads (global memory takes any alignment) instead of 11 byte loads;
                    // v_sad_u8 on single-byte operands is |a - b| + accumulator in one instruction
                    typedef unsigned int u32a __attribute__((aligned(1)));
                    const unsigned w0 = *(const u32a*)(rr - 5), w1 = *(const u32a*)(rr - 1), w2 = *(const u32a*)(rr + 3);
#pragma unroll
                    for (int k = 0; k < 11; ++k) {
                        const unsigned wk = k < 4 ? w0 : k < 8 ? w1 : w2;
                        sad[k] = (int)__builtin_amdgcn_sad_u8(vl, (wk >> (8 * (k & 3))) & 0xFFu, (unsigned)sad[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 11; ++k)
#pragma unroll
                    for (int o = GS / 2; o > 0; o >>= 1) sad[k] += __shfl_xor(sad[k], o);
                int bestD = 0x7FFFFFFF, bestinc = 0;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float dist = (float)sad[k];
                    if (dist < (float)bestD) { bestD = (int)dist; bestinc = k - Lh; }
                }
                if (!(bestinc == -Lh || bestinc == Lh)) {
                    float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                    for (int k = 1; k < 10; ++k) if (k - Lh == bestinc) { d1 = (float)sad[k - 1]; d2 = (float)sad[k]; d3 = (float)sad[k + 1]; }
                    const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = lv.sf(levelL) * ((float)scaleduR0 + (float)bestinc + deltaR);
                        float disparity = (uL - bestuR);
                        if (disparity >= minD && disparity < maxD) {
                            if (disparity <= 0) { disparity = 0.01; bestuR = (float)((double)uL - 0.01); }
                            depth_out = mbf / disparity;
                            ur_out = bestuR;
                            sad_out = bestD;
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) { uright[iL] = ur_out; depth[iL] = depth_out; bestSad[iL] = sad_out; }
}

struct StereoLevelsView {                                  // accessor form of StereoLevels (one pair: orbm_stereo_matches)
    const StereoLevels& s;
    __device__ float sf(int o) const { return s.sf[o]; }
    __device__ float isf(int o) const { return s.isf[o]; }
    __device__ const uint8_t* L(int l) const { return s.L[l]; }
    __device__ const uint8_t* R(int l) const { return s.R[l]; }
    __device__ int pitchL(int l) const { return s.pitchL[l]; }
    __device__ int pitchR(int l) const { return s.pitchR[l]; }
    __device__ int wR(int l) const { return s.wR[l]; }
};
__global__ __launch_bounds__(256) void k_stereo(const KpIn* __restrict__ kl, const uint8_t* __restrict__ dl, int nl,
                                                const KpIn* __restrict__ kr, const uint8_t* __restrict__ dr, int nr,
                                                StereoLevels lv, float mb, float mbf, float* __restrict__ uright,
                                                float* __restrict__ depth, int* __restrict__ bestSad) {
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= nl) return;
    const StereoLevelsView v{lv};
    stereo_body<64>(kl, dl, nl, kr, dr, nr, v, mb, mbf, uright, depth, bestSad, iL);
}

// ------------------------------------------------------------------------------------------------
// Batched, device-resident stereo (config C3): pair p = frames (first_l + p, first_r + p) of ONE extractor batch whose
// result block (kps / desc / counts, [frames][cap]) and pyramids sit in HBM.  k_stereo_batch = k_stereo per pair;
// k_stereo_cut = the median cut of Frame.cc:1261-1275 on the device: sort the SAD of the pair's stereo points, take the
// element at size/2, drop every point with SAD >= 1.5f*1.4f*median.
// ------------------------------------------------------------------------------------------------
struct StereoBatchLayout {                                  // where an extractor keeps a batch's pyramids (orbx_internal_batch_layout)
    const uint8_t* const* l0; int l0pitch; const uint8_t* pyr; size_t frameBytes;
    int off[12], pitch[12], w[12]; float sf[12], isf[12];
};
struct StereoBatchView {
    const StereoBatchLayout& b; int fl, fr;
    __device__ float sf(int o) const { return b.sf[o]; }
    __device__ float isf(int o) const { return b.isf[o]; }
    __device__ const uint8_t* L(int l) const { return l == 0 ? b.l0[fl] : b.pyr + (size_t)fl * b.frameBytes + b.off[l]; }
    __device__ const uint8_t* R(int l) const { return l == 0 ? b.l0[fr] : b.pyr + (size_t)fr * b.frameBytes + b.off[l]; }
    __device__ int pitchL(int l) const { return l == 0 ? b.l0pitch : b.pitch[l]; }
    __device__ int pitchR(int l) const { return l == 0 ? b.l0pitch : b.pitch[l]; }
    __device__ int wR(int l) const { return b.w[l]; }
};
// k_stereo_rows: vRowIndices of every pair's right image as CSR (Frame.cc:1064-1083): rowStart [npairs][nrows + 1],
// rowIdx [npairs][rowCap] (right keypoint indices; a keypoint enters the rows of its band).  One workgroup per pair: band
// histogram in LDS -> exclusive scan -> scatter.  Entries beyond rowCap are dropped and flagged (never with rowCap = 16 * cap).
__global__ __launch_bounds__(256) void k_stereo_rows(const KpIn* __restrict__ kps, const int* __restrict__ counts, int cap, int first_r,
                                                     StereoBatchLayout B, int nrows, int rowCap, int* __restrict__ rowStart,
                                                     unsigned short* __restrict__ rowIdx, int* __restrict__ err) {
    extern __shared__ int srow[];                                        // [nrows + 1] counts -> starts, then fill cursors [nrows]
    const int pair = blockIdx.x, tid = threadIdx.x, fr = first_r + pair;
    const int nr = min(counts[fr], cap);
    const KpIn* kr = kps + (size_t)fr * cap;
    int* cur = srow + nrows + 1;
    for (int i = tid; i <= nrows; i += 256) srow[i] = 0;
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        const float r = 2.0f * B.sf[kr[i].octave];
        const int maxr = min((int)ceilf(kr[i].y + r), nrows - 1), minr = max((int)floorf(kr[i].y - r), 0);
        for (int y = minr; y <= maxr; ++y) atomicAdd(&srow[y], 1);
    }
    __syncthreads();
    if (tid < 64) {                                                      // exclusive scan of the row counts by one wave
        int carry = 0;
        for (int b0 = 0; b0 <= nrows; b0 += 64) {
            const int i = b0 + tid;
            const int v = i < nrows ? srow[i] : 0;
            int sc_ = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc_, o); if (tid >= o) sc_ += t; }
            if (i <= nrows) srow[i] = carry + sc_ - v;
            carry += __shfl(sc_, 63);
        }
    }
    __syncthreads();
    int* rs = rowStart + (size_t)pair * (nrows + 1);
    for (int i = tid; i <= nrows; i += 256) { rs[i] = min(srow[i], rowCap); if (i < nrows) cur[i] = srow[i]; }
    __syncthreads();
    unsigned short* ri = rowIdx + (size_t)pair * rowCap;
    for (int i = tid; i < nr; i += 256) {
        const float r = 2.0f * B.sf[kr[i].octave];
        const int maxr = min((int)ceilf(kr[i].y + r), nrows - 1), minr = max((int)floorf(kr[i].y - r), 0);
        for (int y = minr; y <= maxr; ++y) {
            const int pos = atomicAdd(&cur[y], 1);
            if (pos < rowCap) ri[pos] = (unsigned short)i; else *err = 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_stereo_batch(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc, const int* __restrict__ counts,
                                                      int cap, int first_l, int first_r, StereoBatchLayout B, float mb, float mbf,
                                                      float* __restrict__ uright, float* __restrict__ depth, int* __restrict__ bestSad,
                                                      const int* __restrict__ rowStart, const unsigned short* __restrict__ rowIdx, int nrows, int rowCap) {
    const int pair = blockIdx.y, fl = first_l + pair, fr = first_r + pair;
    const int nl = min(counts[fl], cap), nr = min(counts[fr], cap);
#ifndef ST_GS
#define ST_GS 16
#endif
    const int iL = blockIdx.x * (256 / ST_GS) + (threadIdx.x / ST_GS);   // 64 / ST_GS left keypoints per wave (ST_GS lanes each)
    if (iL >= nl) return;
    const StereoBatchView v{B, fl, fr};
    const size_t o = (size_t)pair * cap;
    if (nr == 0) { if ((threadIdx.x & (ST_GS - 1)) == 0) { uright[o + iL] = -1.0f; depth[o + iL] = -1.0f; bestSad[o + iL] = -1; } return; }
    stereo_body<ST_GS>(kps + (size_t)fl * cap, desc + (size_t)fl * cap * 32, nl, kps + (size_t)fr * cap, desc + (size_t)fr * cap * 32, nr, v, mb, mbf,
                uright + o, depth + o, bestSad + o, iL, rowStart ? rowStart + (size_t)pair * (nrows + 1) : nullptr,
                rowIdx ? rowIdx + (size_t)pair * rowCap : nullptr, nrows);
}

// the element of rank m/2 of the pair's SAD values (< 2^15: 121 pixels x 255) by a two-level histogram -- 256 bins of sad >> 7,
// then 128 bins of the low 7 bits inside the bin that holds the rank -- instead of sorting them
__global__ __launch_bounds__(256) void k_stereo_cut(const int* __restrict__ counts, int cap, int first_l, int n2, const int* __restrict__ bestSad,
                                                    float* __restrict__ uright, float* __restrict__ depth, int* __restrict__ kept) {
    __shared__ int hist[256];
    __shared__ int s_m, s_bin, s_before, s_med, s_kept;
    (void)n2;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nl = min(counts[first_l + pair], cap);
    const size_t o = (size_t)pair * cap;
    hist[tid] = 0;
    if (tid == 0) { s_m = 0; s_kept = 0; }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += 256) { const int v = bestSad[o + i]; if (v >= 0) { atomicAdd(&hist[min(v >> 7, 255)], 1); ++mine; } }
    atomicAdd(&s_m, mine);
    __syncthreads();
    const int m = s_m;
    if (m == 0) { if (tid == 0) kept[pair] = 0; return; }
    const int rank = m / 2;                                                  // vDistIdx[size / 2] of the sorted list (Frame.cc:1263)
    auto locate = [&](int r, int nb) {                                       // bin whose cumulative count first exceeds r; s_before = count below it
        if (tid < 64) {
            int carry = 0;
            for (int b0 = 0; b0 < nb; b0 += 64) {
                const int v = hist[b0 + tid];
                int inc = v;
#pragma unroll
                for (int s_ = 1; s_ < 64; s_ <<= 1) { const int t = __shfl_up(inc, s_); if (tid >= s_) inc += t; }
                const unsigned long long hit = __ballot(carry + inc > r);
                if (hit) {
                    const int l = __builtin_ctzll(hit);
                    if (tid == l) { s_bin = b0 + l; s_before = carry + inc - v; }
                    break;
                }
                carry += __shfl(inc, 63);
            }
        }
    };
    locate(rank, 256);
    __syncthreads();
    const int hb = s_bin, before = s_before;
    __syncthreads();
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 256) { const int v = bestSad[o + i]; if (v >= 0 && min(v >> 7, 255) == hb) atomicAdd(&hist[hb == 255 ? min(v - (255 << 7), 127) : (v & 127)], 1); }
    __syncthreads();
    locate(rank - before, 128);
    __syncthreads();
    if (tid == 0) s_med = (hb << 7) + s_bin;
    __syncthreads();
    const float median = (float)s_med;
    const float thDist = 1.5f * 1.4f * median;
    mine = 0;
    for (int i = tid; i < nl; i += 256) {
        const int sdv = bestSad[o + i];
        if (sdv < 0) continue;
        if ((float)sdv < thDist) ++mine;
        else { uright[o + i] = -1.0f; depth[o + i] = -1.0f; }
    }
    atomicAdd(&s_kept, mine);
    __syncthreads();
    if (tid == 0) kept[pair] = s_kept;
}

// ------------------------------------------------------------------------------------------------
// k_triangulate_batch: ORBmatcher::SearchForTriangulation_ (ORBmatcher.cc:1388-1629, pinhole, orientation check off as
// LocalMapping calls it) for a batch of KeyFrame pairs held in extractor result blocks.  One wavefront per feature of
// KeyFrame 1.  Its bucket = the features of KeyFrame 2 with the same vocabulary node (FeatureVector order = ascending
// index).  The reference walks the bucket keeping `dist <= TH_LOW && dist <= bestDist` candidates that pass the epipole /
// epipolar gates, so the survivor is the smallest distance among the gate-passing candidates and, on ties, the LAST one:
// a min-reduction over keys (dist << 16 | 0xFFFF - idx2).  vbMatched2 is not kept by this overload (:1567), so the
// features of KeyFrame 1 are independent.
// ------------------------------------------------------------------------------------------------
struct TriParams { float F12[9]; float epx, epy; float sf2[12], sigma2[12]; int onlyStereo, coarse; };
// k_tri_buckets: the features of every KeyFrame 2 grouped by (vocabulary node & 255): bStart [npairs][257], bIdx [npairs][cap].
// A FeatureVector bucket is then one short list (plus the few features of other nodes that share the low byte, filtered by
// the exact node id in the search); the order inside a list does not matter to k_triangulate_batch's min-key reduction.
__global__ __launch_bounds__(256) void k_tri_buckets(const int* __restrict__ counts2, const int* __restrict__ node2, int cap,
                                                     int* __restrict__ bStart, unsigned short* __restrict__ bIdx) {
    __shared__ int hist[257], cur[256];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n2 = min(counts2[pair], cap);
    const int* nd = node2 + (size_t)pair * cap;
    hist[tid] = 0; if (tid == 0) hist[256] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += 256) atomicAdd(&hist[nd[i] & 255], 1);
    __syncthreads();
    if (tid < 64) {
        int carry = 0;
        for (int b0 = 0; b0 < 256; b0 += 64) {
            const int v = hist[b0 + tid];
            int sc_ = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc_, o); if (tid >= o) sc_ += t; }
            hist[b0 + tid] = carry + sc_ - v;
            carry += __shfl(sc_, 63);
        }
        if (tid == 0) hist[256] = carry;
    }
    __syncthreads();
    bStart[(size_t)pair * 257 + tid] = hist[tid]; if (tid == 0) bStart[(size_t)pair * 257 + 256] = hist[256];
    cur[tid] = hist[tid];
    __syncthreads();
    for (int i = tid; i < n2; i += 256) bIdx[(size_t)pair * cap + atomicAdd(&cur[nd[i] & 255], 1)] = (unsigned short)i;
}

__global__ __launch_bounds__(256) void k_triangulate_batch(const KpIn* __restrict__ kps1, const uint8_t* __restrict__ desc1, const int* __restrict__ counts1,
                                                           const int* __restrict__ node1, const float* __restrict__ ur1,
                                                           const KpIn* __restrict__ kps2, const uint8_t* __restrict__ desc2, const int* __restrict__ counts2,
                                                           const int* __restrict__ node2, const float* __restrict__ ur2,
                                                           int cap, TriParams P, int* __restrict__ matches12, int* __restrict__ nmatches,
                                                           const int* __restrict__ bStart, const unsigned short* __restrict__ bIdx) {
    // one THREAD per feature of KeyFrame 1 (a bucket holds a dozen candidates: a wave per feature spent its time on the chain of
    // dependent loads, not on the distances); the bucket is walked serially, the survivor is the min of (dist << 16 | 0xFFFF - idx2)
    const int pair = blockIdx.y;
    const int n1 = min(counts1[pair], cap);
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    const size_t o = (size_t)pair * cap;
    int res = -1;
    if (i1 < n1) {
        const KpIn kp1 = kps1[o + i1];
        const int nd = node1[o + i1];
        const bool bStereo1 = ur1 && ur1[o + i1] >= 0;
        unsigned int best = 0xFFFFFFFFu;
        if (!(P.onlyStereo && !bStereo1)) {
            const uint4* qp = (const uint4*)(desc1 + (o + i1) * 32);
            const uint4 qlo = qp[0], qhi = qp[1];
            const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                              (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
            // the epipolar line of kp1 in image 2 (Pinhole::epipolarConstrain_, Pinhole.cpp:281-287)
            const float la = kp1.x * P.F12[0] + kp1.y * P.F12[3] + P.F12[6];
            const float lb = kp1.x * P.F12[1] + kp1.y * P.F12[4] + P.F12[7];
            const float lc = kp1.x * P.F12[2] + kp1.y * P.F12[5] + P.F12[8];
            const float den = la * la + lb * lb;
            const int c0 = bStart[(size_t)pair * 257 + (nd & 255)], c1 = bStart[(size_t)pair * 257 + (nd & 255) + 1];
            for (int ci = c0; ci < c1; ++ci) {
                const int i2 = (int)bIdx[o + ci];
                if (node2[o + i2] != nd) continue;
                const bool bStereo2 = ur2 && ur2[o + i2] >= 0;
                if (P.onlyStereo && !bStereo2) continue;
                const uint4* tp = (const uint4*)(desc2 + (o + i2) * 32);
                const uint4 lo = tp[0], hi = tp[1];
                const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                     (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                if (d > 50) continue;                                                // TH_LOW
                const KpIn kp2 = kps2[o + i2];
                if (!bStereo1 && !bStereo2) {
                    const float distex = P.epx - kp2.x, distey = P.epy - kp2.y;
                    if (distex * distex + distey * distey < 100 * P.sf2[kp2.octave]) continue;
                }
                bool epi = false;
                if (den != 0) {
                    const float num = la * kp2.x + lb * kp2.y + lc;
                    const float dsqr = num * num / den;
                    epi = (double)dsqr < 3.84 * (double)P.sigma2[kp2.octave];         // float compared with the double product, as written (Pinhole.cpp:295)
                }
                if (epi || P.coarse) best = min(best, ((unsigned)d << 16) | (0xFFFFu - (unsigned)i2));
            }
        }
        res = best == 0xFFFFFFFFu ? -1 : (int)(0xFFFFu - (best & 0xFFFFu));
        matches12[o + i1] = res;
    }
    const unsigned long long found = __ballot(res >= 0);
    if ((threadIdx.x & 63) == 0 && found) atomicAdd(&nmatches[pair], __popcll(found));
}

// ------------------------------------------------------------------------------------------------
// Batched, device-resident forms (frame-to-frame tracking with no host round trip).
// k_grid_build_batch: one workgroup per frame of an extractor result block [nframes][cap].
// k_track_window: wave per keypoint of the query frame; window = (x+dx, y+dy) +- th*scale[octave], levels
// [octave-1, octave+1] in the train frame's grid (the mono SearchByProjection window, ORBmatcher.cc:2543-2549);
// writes the first-minimum best and the runner-up (strict <, candidate order) -- the claim-free part of the search.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_build_batch(const KpIn* __restrict__ kps, const int* __restrict__ counts, int cap, int n2,
                                                          float min_x, float min_y, float inv_w, float inv_h,
                                                          int* __restrict__ grid_start, int* __restrict__ grid_idx) {
    extern __shared__ unsigned int keys[];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts[frame], cap);
    const KpIn* kp = kps + (size_t)frame * cap;
    __shared__ int s_placed;
    if (tid == 0) s_placed = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += 256) {
        unsigned int key = 0xFFFFFFFFu;
        if (i < n) {
            const int px = (int)roundf((kp[i].x - min_x) * inv_w);
            const int py = (int)roundf((kp[i].y - min_y) * inv_h);
            if (px >= 0 && px < 64 && py >= 0 && py < 48) { key = ((unsigned)(px * 48 + py) << 16) | (unsigned)i; atomicAdd(&s_placed, 1); }
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned int a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const int np = s_placed;
    int* gi = grid_idx + (size_t)frame * cap;
    int* gs = grid_start + (size_t)frame * (64 * 48 + 1);
    for (int i = tid; i < np; i += 256) gi[i] = (int)(keys[i] & 0xFFFFu);
    for (int c = tid; c <= 64 * 48; c += 256) {
        const unsigned int target = (unsigned)c << 16;
        int lo = 0, hi = np;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < target) lo = mid + 1; else hi = mid; }
        gs[c] = lo;
    }
}

struct ScaleTab { float sf[12]; };

__global__ __launch_bounds__(256) void k_track_window(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                      const int* __restrict__ counts, int cap, const int* __restrict__ grid_start,
                                                      const int* __restrict__ grid_idx, float min_x, float min_y, float inv_w, float inv_h,
                                                      int q_first, int t_first, float th, ScaleTab st, float dx, float dy,
                                                      int* __restrict__ best_idx, int* __restrict__ best_dist,
                                                      int* __restrict__ second_dist) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.y;
    const int qf = q_first + pair, tf = t_first + pair;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= min(counts[qf], cap)) return;
    const KpIn kq = kps[(size_t)qf * cap + q];
    const KpIn* kt = kps + (size_t)tf * cap;
    const uint8_t* dt = desc + (size_t)tf * cap * 32;
    const int* gs = grid_start + (size_t)tf * (64 * 48 + 1);
    const int* gi = grid_idx + (size_t)tf * cap;
    const float x = kq.x + dx, y = kq.y + dy, r = th * st.sf[kq.octave];
    const int minLevel = kq.octave - 1, maxLevel = kq.octave + 1;
    unsigned int best = 0xFFFFFFFFu;                                        // (dist << 16 | order): first minimum in candidate order
    int second = 256, bestk = -1;
    const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
    const int nMaxCellX = min(63, (int)ceilf((x - min_x + r) * inv_w));
    const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
    const int nMaxCellY = min(47, (int)ceilf((y - min_y + r) * inv_h));
    int ord0 = 0;
    int bd = 256, bk = -1, sd = 256;                                        // per-lane partials
    unsigned int bo = 0xFFFFFFFFu;
    if (nMinCellX < 64 && nMaxCellX >= 0 && nMinCellY < 48 && nMaxCellY >= 0) {
        const uint4* qp = (const uint4*)(desc + ((size_t)qf * cap + q) * 32);
        const uint4 qlo = qp[0], qhi = qp[1];
        const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                          (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            const int j0 = gs[ix * 48 + nMinCellY], j1 = gs[ix * 48 + nMaxCellY + 1];
            for (int jb = j0; jb < j1; jb += 64) {
                const int j = jb + lane;
                if (j < j1) {
                    const int k = gi[j];
                    const KpIn kp = kt[k];
                    bool ok = !(kp.octave < minLevel) && !(kp.octave > maxLevel);   // bCheckLevels is true here (maxLevel >= 0)
                    if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) ok = false;
                    if (ok) {
                        const uint4* tp = (const uint4*)(dt + (size_t)k * 32);
                        const uint4 lo = tp[0], hi = tp[1];
                        const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                             (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                        const unsigned int key = ((unsigned)d << 20) | (unsigned)(ord0 + (j - j0));   // order within the whole sweep
                        if (key < bo) { sd = bd; bo = key; bd = d; bk = k; }
                        else if (d < sd) sd = d;
                    }
                }
            }
            ord0 += j1 - j0;
        }
    }
    // wave merge of (best key, runner-up distance): runner-up = min over all candidates except the winner
    unsigned int wbest = bo;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wbest = min(wbest, (unsigned)__shfl_xor((int)wbest, o));
    int cand2 = (bo == wbest) ? sd : bd;                                     // lanes that lost contribute their own best
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand2 = min(cand2, __shfl_xor(cand2, o));
    best = wbest; second = cand2;
    const unsigned long long who = __ballot(bo == wbest && wbest != 0xFFFFFFFFu);
    if (who) bestk = __shfl(bk, __ffsll((long long)who) - 1);
    if (lane == 0) {
        const size_t o = (size_t)pair * cap + q;
        best_idx[o] = bestk;
        best_dist[o] = bestk < 0 ? 256 : (int)(best >> 20);
        second_dist[o] = second;
    }
}

// ------------------------------------------------------------------------------------------------
// Batched SearchByProjection(Frame, Frame), final matches on the device (ORBmatcher.cc:2469-2711, mono branch).
// k_track_topk: wave per query keypoint (same window as k_track_window).  What the sequential claim replay can ever look at:
//   the window's candidate count and its TK_K best candidates in (distance, visiting order) rank -- the order in which the
//   reference's `dist < bestDist` scan would prefer them -- each as ONE word: dist << 21 | rotation bin << 16 | keypoint index
//   (bin = round((angle_q - angle_t [+360]) * 30/360), 30 -> 0, ORBmatcher.cc:2596-2603; the replay needs no second gather).
// k_track_claim: ONE wave per frame pair replays the claims in query order (:2555-2593): a query takes its first candidate that is
//   not blocked (a MapPoint with observations already sits there: cur_blocked, or an earlier query with observations claimed it,
//   :2565-2567), bestDist <= TH_HIGH assigns the slot (:2589-2592), the rotation histogram keeps (slot, bin) of every assignment
//   and the three-maxima cull clears the others (:2690-2708).  Eight queries' lists (8 x 8 keys) are fetched per round trip and the
//   next eight are requested before the current ones are resolved; the blocked set is a bit array in LDS, claims inside a group
//   travel by register compare.  A query whose TK_K listed candidates are all blocked although its window holds more is rescanned
//   in place, blocked set applied (rare; the single-frame host replay falls back to full lists in the same case).
// ------------------------------------------------------------------------------------------------
#define TK_K 8
#define TK_NOBIN 31
// wave-wide minimum of a 32-bit key by DPP (six v_min_u32 with data movement folded in; the result of a full reduction sits in lane 63)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum over the 16 lanes of a DPP row, in every lane of the row (four rotations)
__device__ __forceinline__ unsigned row16_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}
__device__ __forceinline__ int wave_excl_scan(int v, int* total) {           // exclusive prefix sum over the 64 lanes
    int s = v;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);
    *total = __builtin_amdgcn_readlane(s, 63);
    return s - v;
}
#ifdef ORBX_AB   /* A/B reference (a wave per query), not in the product library */
// The window's grid columns are flattened into ONE candidate list (column starts / lengths loaded by one lane each, prefix sum across
// lanes), so that a window of up to 64 candidates costs four dependent round trips (cell ranges, indices, keypoints, descriptors)
// whatever its shape -- the per-column loop of k_track_window pays them per column -- and the eight best come out of eight DPP
// minimum reductions of a 32-bit key (distance << 16 | position), already in rank order.
__global__ __launch_bounds__(256) void k_track_topk(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                    const int* __restrict__ counts, int cap, const int* __restrict__ grid_start,
                                                    const int* __restrict__ grid_idx, float min_x, float min_y, float inv_w, float inv_h,
                                                    int q_first, int t_first, float th, ScaleTab st, float dx, float dy, float factor,
                                                    int* __restrict__ out_cnt, unsigned int* __restrict__ out_keys) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.y;
    const int qf = q_first + pair, tf = t_first + pair;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= min(counts[qf], cap)) return;
    const KpIn kq = kps[(size_t)qf * cap + q];
    const KpIn* kt = kps + (size_t)tf * cap;
    const uint8_t* dt = desc + (size_t)tf * cap * 32;
    const int* gs = grid_start + (size_t)tf * (64 * 48 + 1);
    const int* gi = grid_idx + (size_t)tf * cap;
    const float x = kq.x + dx, y = kq.y + dy, r = th * st.sf[kq.octave];
    const int minLevel = kq.octave - 1, maxLevel = kq.octave + 1;
    const unsigned INV = 0xFFFFFFFFu;
    unsigned topKey[TK_K], topPay[TK_K];                                   // (distance << 16 | position) ascending; payload = bin << 16 | keypoint
#pragma unroll
    for (int i = 0; i < TK_K; ++i) { topKey[i] = INV; topPay[i] = INV; }
    int cnt = 0;
    const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
    const int nMaxCellX = min(63, (int)ceilf((x - min_x + r) * inv_w));
    const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
    const int nMaxCellY = min(47, (int)ceilf((y - min_y + r) * inv_h));
    if (nMinCellX < 64 && nMaxCellX >= 0 && nMinCellY < 48 && nMaxCellY >= 0) {
        const uint4* qp = (const uint4*)(desc + ((size_t)qf * cap + q) * 32);
        const uint4 qlo = qp[0], qhi = qp[1];
        const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                          (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int ncols = nMaxCellX - nMinCellX + 1;                       // <= 64 (the grid has 64 columns)
        int cj0 = 0, clen = 0;
        if (lane < ncols) {
            const int ix = nMinCellX + lane;
            cj0 = gs[ix * 48 + nMinCellY];
            clen = gs[ix * 48 + nMaxCellY + 1] - cj0;
        }
        int total;
        const int coff = wave_excl_scan(clen, &total);
        bool first = true;
        for (int base = 0; base < total; base += 64) {
            const int t = base + lane;
            // the lane's column: the last one whose offset is <= t (columns in ascending order, empty ones share an offset)
            int myj0 = 0, myoff = 0;
            for (int c = 0; c < ncols; ++c) {
                const int oc = __builtin_amdgcn_readlane(coff, c), jc = __builtin_amdgcn_readlane(cj0, c);
                if (t >= oc) { myoff = oc; myj0 = jc; }
            }
            bool ok = false;
            int k = 0;
            float ang = 0.f;
            if (t < total) {
                k = gi[myj0 + (t - myoff)];
                const KpIn kp = kt[k];
                ok = !(kp.octave < minLevel) && !(kp.octave > maxLevel);   // bCheckLevels is true here (maxLevel >= 0)
                if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) ok = false;
                ang = kp.angle;
            }
            const unsigned long long bal = __ballot(ok);
            if (bal == 0) continue;
            unsigned key = INV, pay = INV;
            if (ok) {
                const int pos = cnt + __popcll(bal & lt);
                const uint4* tp = (const uint4*)(dt + (size_t)k * 32);
                const uint4 lo = tp[0], hi = tp[1];
                const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                     (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                float rot = kq.angle - ang;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == 30) bin = 0;
                if (bin < 0 || bin >= 30) bin = TK_NOBIN;
                key = ((unsigned)d << 16) | (unsigned)pos;                   // positions < 65536 (cap)
                pay = ((unsigned)bin << 16) | (unsigned)k;
            }
            cnt += __popcll(bal);
            if (first) {
                // first chunk: the minima come out in rank order
#pragma unroll
                for (int i = 0; i < TK_K; ++i) {
                    const unsigned m = wave_min_u32(key);
                    if (m == INV) break;
                    const int src = __ffsll((long long)__ballot(key == m)) - 1;     // keys are unique (position)
                    topKey[i] = m; topPay[i] = (unsigned)__builtin_amdgcn_readlane((int)pay, src);
                    if (lane == src) key = INV;
                }
                first = false;
            } else {
                for (int rnd = 0; rnd < TK_K; ++rnd) {                     // later chunks (windows of more than 64 grid entries): insertion
                    const unsigned m = wave_min_u32(key);
                    if (m >= topKey[TK_K - 1]) break;                      // INV included
                    const int src = __ffsll((long long)__ballot(key == m)) - 1;
                    unsigned ck = m, cp = (unsigned)__builtin_amdgcn_readlane((int)pay, src);
                    if (lane == src) key = INV;
#pragma unroll
                    for (int i = 0; i < TK_K; ++i) {
                        const bool sw = ck < topKey[i];
                        const unsigned tk = topKey[i], tp2 = topPay[i];
                        topKey[i] = sw ? ck : tk; topPay[i] = sw ? cp : tp2;
                        ck = sw ? tk : ck; cp = sw ? tp2 : cp;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        const size_t o = (size_t)pair * cap + q;
        out_cnt[o] = cnt;
#pragma unroll
        for (int i = 0; i < TK_K; ++i)
            out_keys[o * TK_K + i] = topKey[i] == INV ? 0xFFFFFFFFu : ((topKey[i] >> 16) << 21) | (topPay[i] & 0x1FFFFFu);
    }
}

#endif  /* ORBX_AB */

// k_track_topk16: the same lists with SIXTEEN lanes per query (four queries per wave).  A mono window (th = 15) holds 6 grid entries at
// level 0 and ~40 at level 7 (14 on average over a 1000-feature frame), so a wave per query keeps most lanes idle and the kernel is bound
// by the number of waves it can keep in flight across five dependent round trips.  A lane owns entries t = base + 16 j + l (j < NJ <= 4:
// up to 64 entries per query and pass); column ranges sit in LDS (row-local prefix sums by DPP row_shr); the eight best come from
// row-wide minimum reductions (DPP row_ror) whose winner writes its own word to the LDS list -- or, with one entry per lane, from
// each lane counting the smaller keys of its row -- and later passes (windows of more than 64 grid entries) feed the list back in as
// one more key per lane.  The pass body is BRANCH-FREE per NJ (inactive entries read row 0 instead of being predicated off), so that
// the NJ loads of every step are in flight together; consecutive queries sit on the same pyramid level (similar windows), so NJ is
// chosen per wave.
struct Tk16 {
    const uint4* ent; const uint8_t* dt; int cap;
    float x, y, r, qangle, factor; int minLevel, maxLevel;
    int total, ncols, qr, l16, wr;
};
template <int NJ, bool FB>
__device__ __forceinline__ void tk16_pass(const Tk16& c, const u64 (&a)[4], int base, int& cnt, const int2 (*sCol)[64], const int (*sAdj)[64], uint2 (*sK)[16], uint2 (*sTop)[TK_K]) {
    const unsigned INV = 0xFFFFFFFFu;
    int t[NJ], eidx[NJ];
    const int cnt0 = cnt;
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = base + 16 * j + c.l16;
    if (!FB) {
        // first 64 entries of a window: the prologue left (grid position - list position) of every entry in LDS
#pragma unroll
        for (int j = 0; j < NJ; ++j) eidx[j] = t[j] < c.total ? t[j] + sAdj[c.qr][t[j]] : 0;
    } else {
        // later passes: each entry's column is the last one whose offset is <= t (columns ascending, empty ones share an offset)
        int off[NJ], j0[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { off[j] = 0; j0[j] = 0; }
        for (int col = 0; __any(col < c.ncols); ++col) {
            const int2 e = sCol[c.qr][col];
            const bool in = col < c.ncols;
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (in && t[j] >= e.x) { off[j] = e.x; j0[j] = e.y; }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) eidx[j] = t[j] < c.total ? j0[j] + (t[j] - off[j]) : 0;
    }
    // one 16-byte record per grid entry, in the grid's own order (k_track_pack): (x, y, angle, octave << 16 | keypoint) -- the entry and
    // its keypoint in ONE load from consecutive addresses instead of an index and three dependent scattered ones
    int k[NJ];
    float kx[NJ], ky[NJ], ang[NJ];
    int oct[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint4 e = c.ent[eidx[j]];
        kx[j] = __uint_as_float(e.x); ky[j] = __uint_as_float(e.y); ang[j] = __uint_as_float(e.z);
        oct[j] = (int)(e.w >> 16); k[j] = (int)(e.w & 0xFFFFu);
    }
    bool ok[NJ];
    uint4 lo[NJ], hi[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        ok[j] = t[j] < c.total && !(oct[j] < c.minLevel) && !(oct[j] > c.maxLevel) && (fabsf(kx[j] - c.x) < c.r && fabsf(ky[j] - c.y) < c.r);   // bCheckLevels is true here
        const uint4* tp = (const uint4*)(c.dt + (size_t)(ok[j] ? k[j] : 0) * 32);
        lo[j] = tp[0]; hi[j] = tp[1];
    }
    const unsigned below = (1u << c.l16) - 1u;
    unsigned key[NJ + 1], word[NJ + 1];
    int posj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned rowbits = (unsigned)(__ballot(ok[j]) >> (16 * c.wr)) & 0xFFFFu;
        const int pos = cnt + __popc(rowbits & below);
        posj[j] = pos;
        const int d = ham256(a, (u64)lo[j].x | ((u64)lo[j].y << 32), (u64)lo[j].z | ((u64)lo[j].w << 32),
                             (u64)hi[j].x | ((u64)hi[j].y << 32), (u64)hi[j].z | ((u64)hi[j].w << 32));
        float rot = c.qangle - ang[j];
        if (rot < 0.0f) rot += 360.0f;
        int bin = (int)roundf(rot * c.factor);
        if (bin == 30) bin = 0;
        if (bin < 0 || bin >= 30) bin = TK_NOBIN;
        key[j] = ok[j] ? ((unsigned)d << 16) | (unsigned)pos : INV;          // positions < 65536 (cap)
        word[j] = ((unsigned)d << 21) | ((unsigned)bin << 16) | (unsigned)k[j];
        cnt += __popc(rowbits);
    }
    key[NJ] = INV; word[NJ] = INV;
    if (FB && c.l16 < TK_K) { const uint2 pv = sTop[c.qr][c.l16]; key[NJ] = pv.x; word[NJ] = pv.y; }   // the list so far competes again
    // !FB: at most 16 candidates of a row passed the tests (the rule: 14 grid entries per window, a quarter of them on the right levels
    // and inside the window): they are packed to one per lane -- in visiting order -- and each lane's rank is the number of smaller
    // keys in its row (15 rotations; keys are unique by position).  Otherwise eight row-wide minimum reductions.
    const int nv = cnt - cnt0;
    if (!FB && (NJ == 1 || !__any(nv > 16))) {
        unsigned k1 = key[0], w1 = word[0];
        if (NJ > 1) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (key[j] != INV) sK[c.qr][posj[j] - cnt0] = make_uint2(key[j], word[j]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint2 e = sK[c.qr][c.l16];
            k1 = c.l16 < nv ? e.x : INV; w1 = e.y;
        }
        int rank = 0;
        unsigned rk = k1;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            rk = (unsigned)__builtin_amdgcn_update_dpp((int)rk, (int)rk, 0x121, 0xf, 0xf, false);   // row_ror:1
            rank += rk < k1 ? 1 : 0;
        }
        if (k1 != INV && rank < TK_K) sTop[c.qr][rank] = make_uint2(k1, w1);
    } else {
#pragma unroll
        for (int i = 0; i < TK_K; ++i) {
            unsigned mine = key[0];
#pragma unroll
            for (int j = 1; j <= (FB ? NJ : NJ - 1); ++j) mine = min(mine, key[j]);
            const unsigned m = row16_min_u32(mine);
            if (!__any(m != INV)) break;
            if (mine == m && m != INV) {                                    // keys are unique (position): exactly one lane of the row
                unsigned w = word[0];
#pragma unroll
                for (int j = 1; j <= (FB ? NJ : NJ - 1); ++j) if (key[j] == m) w = word[j];
                sTop[c.qr][i] = make_uint2(m, w);
#pragma unroll
                for (int j = 0; j <= (FB ? NJ : NJ - 1); ++j) if (key[j] == m) key[j] = INV;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// k_track_pack: the searched frames' grid entries as 16-byte records in grid order (see tk16_pass)
__global__ __launch_bounds__(256) void k_track_pack(const KpIn* __restrict__ kps, int cap, const int* __restrict__ grid_start,
                                                    const int* __restrict__ grid_idx, int t_first, uint4* __restrict__ ent) {
    const int pair = blockIdx.y, tf = t_first + pair;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    const int n = min(grid_start[(size_t)tf * (64 * 48 + 1) + 64 * 48], cap);
    if (pos >= n) return;
    const int k = grid_idx[(size_t)tf * cap + pos];
    const KpIn kp = kps[(size_t)tf * cap + k];
    ent[(size_t)pair * cap + pos] = make_uint4(__float_as_uint(kp.x), __float_as_uint(kp.y), __float_as_uint(kp.angle), ((unsigned)kp.octave << 16) | (unsigned)k);
}

__global__ __launch_bounds__(256) void k_track_topk16(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                      const int* __restrict__ counts, int cap, const int* __restrict__ grid_start,
                                                      const uint4* __restrict__ ent, float min_x, float min_y, float inv_w, float inv_h,
                                                      int q_first, int t_first, float th, ScaleTab st, float dx, float dy, float factor,
                                                      int* __restrict__ out_cnt, unsigned int* __restrict__ out_keys) {
    __shared__ int2 sCol[16][64];                                           // per query: (offset in the flattened list, first grid entry) of each window column
    __shared__ uint2 sTop[16][TK_K];                                        // per query: (distance << 16 | position, output word), ascending
    __shared__ int sAdj[16][64];                                            // per query: grid position - list position of the first 64 window entries
    __shared__ uint2 sK[16][16];                                            // per query: the candidates that passed, one per lane
    const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15, wr = lane >> 4, qr = threadIdx.x >> 4;
    const int pair = blockIdx.y;
    const int qf = q_first + pair, tf = t_first + pair;
    const int nq = min(counts[qf], cap);
    if ((int)blockIdx.x * 16 >= nq) return;
    const int q = blockIdx.x * 16 + qr;
    const bool live = q < nq;
    const KpIn kq = kps[(size_t)qf * cap + (live ? q : 0)];
    const int* gs = grid_start + (size_t)tf * (64 * 48 + 1);
    const float x = kq.x + dx, y = kq.y + dy, r = th * st.sf[kq.octave];
    const unsigned INV = 0xFFFFFFFFu;
    const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
    const int nMaxCellX = min(63, (int)ceilf((x - min_x + r) * inv_w));
    const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
    const int nMaxCellY = min(47, (int)ceilf((y - min_y + r) * inv_h));
    const bool hit = live && nMinCellX < 64 && nMaxCellX >= 0 && nMinCellY < 48 && nMaxCellY >= 0;
    const int ncols = hit ? nMaxCellX - nMinCellX + 1 : 0;                  // <= 64
    const uint4* qp = (const uint4*)(desc + ((size_t)qf * cap + (live ? q : 0)) * 32);
    const uint4 qlo = qp[0], qhi = qp[1];
    const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                      (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
    if (l16 < TK_K) sTop[qr][l16] = make_uint2(INV, INV);
    int total = 0;
    for (int cb = 0; __any(cb < ncols); cb += 16) {
        const int c = cb + l16;
        int cj0 = 0, clen = 0;
        if (c < ncols) {
            const int ix = nMinCellX + c;
            cj0 = gs[ix * 48 + nMinCellY];
            clen = gs[ix * 48 + nMaxCellY + 1] - cj0;
        }
        int sc = clen;                                                      // inclusive prefix sum inside the 16-lane row
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xf, 0xf, true);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xf, 0xf, true);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xf, 0xf, true);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x118, 0xf, 0xf, true);
        const int excl = total + sc - clen;
        if (c < ncols) sCol[qr][c] = make_int2(excl, cj0);
        for (int e = 0; __any(e < clen && excl + e < 64); ++e)              // (a cell column of a window holds one or two entries as a rule)
            if (e < clen && excl + e < 64) sAdj[qr][excl + e] = cj0 - excl;
        const int s0 = __builtin_amdgcn_readlane(sc, 15), s1 = __builtin_amdgcn_readlane(sc, 31),
                  s2 = __builtin_amdgcn_readlane(sc, 47), s3 = __builtin_amdgcn_readlane(sc, 63);
        total += wr == 0 ? s0 : wr == 1 ? s1 : wr == 2 ? s2 : s3;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                  // a row lives inside one wave: LDS traffic of the same wave is ordered
    __builtin_amdgcn_wave_barrier();
    Tk16 c;
    c.ent = ent + (size_t)pair * cap; c.dt = desc + (size_t)tf * cap * 32; c.cap = cap;
    c.x = x; c.y = y; c.r = r; c.qangle = kq.angle; c.factor = factor; c.minLevel = kq.octave - 1; c.maxLevel = kq.octave + 1;
    c.total = total; c.ncols = ncols; c.qr = qr; c.l16 = l16; c.wr = wr;
    const int maxTotal = max(max(__builtin_amdgcn_readlane(total, 0), __builtin_amdgcn_readlane(total, 16)),
                             max(__builtin_amdgcn_readlane(total, 32), __builtin_amdgcn_readlane(total, 48)));
    int cnt = 0;
    for (int base = 0; base < maxTotal; base += 64) {
        const int nj = min(4, (maxTotal - base + 15) >> 4);                 // wave-uniform
        if (base == 0) {
            if (nj == 1) tk16_pass<1, false>(c, a, base, cnt, sCol, sAdj, sK, sTop);
            else if (nj == 2) tk16_pass<2, false>(c, a, base, cnt, sCol, sAdj, sK, sTop);
            else if (nj == 3) tk16_pass<3, false>(c, a, base, cnt, sCol, sAdj, sK, sTop);
            else tk16_pass<4, false>(c, a, base, cnt, sCol, sAdj, sK, sTop);
        } else {
            if (nj <= 2) tk16_pass<2, true>(c, a, base, cnt, sCol, sAdj, sK, sTop);
            else tk16_pass<4, true>(c, a, base, cnt, sCol, sAdj, sK, sTop);
        }
    }
    if (live) {
        const size_t o = (size_t)pair * cap + q;
        if (l16 == 0) out_cnt[o] = cnt;
        if (l16 < TK_K) { const uint2 e = sTop[qr][l16]; out_keys[o * TK_K + l16] = e.x == INV ? INV : e.y; }
    }
}

// every listed candidate of query qi is blocked but its window holds more: the window again, blocked set applied (all 64 lanes; qi wave-uniform).
// Returns the best candidate as distance << 21 | rotation bin << 16 | keypoint, or 0xFFFFFFFF.
__device__ __forceinline__ unsigned int tk_rescan(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc, int cap, const int* __restrict__ grid_start,
                                                  const int* __restrict__ grid_idx, float min_x, float min_y, float inv_w, float inv_h, int qf, int tf, int qi,
                                                  float th, const ScaleTab& st, float dx, float dy, float factor, const unsigned int* blk, int lane) {
// every listed candidate is blocked but the window holds more: scan it again with the blocked set applied
    const KpIn kq = kps[(size_t)qf * cap + qi];
    const KpIn* kt = kps + (size_t)tf * cap;
    const uint8_t* dt = desc + (size_t)tf * cap * 32;
    const int* gs = grid_start + (size_t)tf * (64 * 48 + 1);
    const int* gi = grid_idx + (size_t)tf * cap;
    const float x = kq.x + dx, y = kq.y + dy, r = th * st.sf[kq.octave];
    const int minLevel = kq.octave - 1, maxLevel = kq.octave + 1;
    const int nMinCellX = max(0, (int)floorf((x - min_x - r) * inv_w));
    const int nMaxCellX = min(63, (int)ceilf((x - min_x + r) * inv_w));
    const int nMinCellY = max(0, (int)floorf((y - min_y - r) * inv_h));
    const int nMaxCellY = min(47, (int)ceilf((y - min_y + r) * inv_h));
    const uint4* qp = (const uint4*)(desc + ((size_t)qf * cap + qi) * 32);
    const uint4 qlo = qp[0], qhi = qp[1];
    const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                      (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
    u64 bk = ~0ull;
    int ord0 = 0;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
        const int j0 = gs[ix * 48 + nMinCellY], j1 = gs[ix * 48 + nMaxCellY + 1];
        for (int jb = j0; jb < j1; jb += 64) {
            const int j = jb + lane;
            if (j < j1) {
                const int k = gi[j];
                const KpIn kp = kt[k];
                bool ok = !(kp.octave < minLevel) && !(kp.octave > maxLevel);
                if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) ok = false;
                if (ok && !((blk[k >> 5] >> (k & 31)) & 1u)) {
                    const uint4* tp = (const uint4*)(dt + (size_t)k * 32);
                    const uint4 lo = tp[0], hi = tp[1];
                    const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                         (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                    float rot = kq.angle - kp.angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == 30) bin = 0;
                    if (bin < 0 || bin >= 30) bin = TK_NOBIN;
                    // (order among grid entries, not among window members: monotone in it, which is all the first-minimum rule needs)
                    const u64 kk = ((u64)d << 44) | ((u64)(ord0 + (j - j0)) << 24) | ((u64)bin << 16) | (u64)k;
                    bk = kk < bk ? kk : bk;
                }
            }
        }
        ord0 += j1 - j0;
    }
    bk = wave_min_u64(bk);
    return bk != ~0ull ? ((unsigned)(bk >> 44) << 21) | (unsigned)(bk & 0x1FFFFFu) : 0xFFFFFFFFu;
}

#ifdef ORBX_AB   /* A/B reference (eight queries per step), not in the product library */
__global__ __launch_bounds__(64) void k_track_claim(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                    const int* __restrict__ counts, int cap, const int* __restrict__ grid_start,
                                                    const int* __restrict__ grid_idx, float min_x, float min_y, float inv_w, float inv_h,
                                                    int q_first, int t_first, float th, ScaleTab st, float dx, float dy, float factor,
                                                    const int* __restrict__ topCnt, const unsigned int* __restrict__ topKeys,
                                                    const uint8_t* __restrict__ t_blocked, const uint8_t* __restrict__ q_obs, int check_ori,
                                                    unsigned int* __restrict__ accepted, int* __restrict__ match, int* __restrict__ nmatches) {
    extern __shared__ unsigned int tk_lds[];                                // blocked bit array [ceil(cap / 32)], hist[32], "query has observations" bits [ceil(cap / 32)]
    const int lane = threadIdx.x, pair = blockIdx.x;
    const int qf = q_first + pair, tf = t_first + pair;
    const int nq = min(counts[qf], cap), nt = min(counts[tf], cap);
    const int nwords = (cap + 31) >> 5;
    unsigned int* blk = tk_lds;
    unsigned int* hist = tk_lds + nwords;
    unsigned int* obsb = hist + 32;
    int* mrow = match + (size_t)pair * cap;
    unsigned int* acc = accepted + (size_t)pair * cap;
    for (int w = lane; w < nwords; w += 64) {
        unsigned int bits = 0;
        if (t_blocked) {
            const uint8_t* tb = t_blocked + (size_t)tf * cap;
            const int last = max(nt - 1, 0);
            unsigned v[32];                                                 // 32 byte loads in flight (clamped, not predicated), then the bits
#pragma unroll
            for (int b = 0; b < 32; ++b) v[b] = tb[min(w * 32 + b, last)];
#pragma unroll
            for (int b = 0; b < 32; ++b) if (w * 32 + b < nt && v[b]) bits |= 1u << b;
        }
        blk[w] = bits;
        unsigned int ob = 0xFFFFFFFFu;                                      // q_obs absent: every query counts as observed
        if (q_obs) {
            const uint8_t* qo = q_obs + (size_t)qf * cap;
            const int last = max(nq - 1, 0);
            unsigned v[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) v[b] = qo[min(w * 32 + b, last)];
            ob = 0;
#pragma unroll
            for (int b = 0; b < 32; ++b) if (v[b]) ob |= 1u << b;
        }
        obsb[w] = ob;
    }
    if (lane < 32) hist[lane] = 0;
    for (int k = lane; k < cap; k += 64) mrow[k] = -1;                        // ORBM_NO_MATCH
    __syncthreads();
    const size_t rowBase = (size_t)pair * cap;
    const int sub = lane >> 3;                                              // the lane's query inside a group of eight
    int nm = 0, nacc = 0;
    // Two register sets of four groups (32 queries) each: set A is resolved while set B's lists are in flight (a list comes from another
    // XCD's writes, i.e. from memory: ~1.6 us per round trip, against ~0.3 us to resolve a group).  The loaded values are touched only
    // when their set becomes current -- a move or a select behind the load would make the wave wait for it there -- and the loads are
    // UNCONDITIONAL (clamped index): a load under a branch turns every later s_waitcnt into a full drain.
    const int qLast = max(nq - 1, 0);
    unsigned int ak0, ak1, ak2, ak3;
    int ac0, ac1, ac2, ac3;
    auto fetch = [&](int qi, unsigned int& k_, int& c_) {
        const int qc = min(qi, qLast);
        k_ = topKeys[(rowBase + qc) * TK_K + (lane & 7)];
        c_ = topCnt[rowBase + qc];
    };
    fetch(sub, ak0, ac0); fetch(8 + sub, ak1, ac1); fetch(16 + sub, ak2, ac2); fetch(24 + sub, ak3, ac3);
    for (int G = 0; G < nq; G += 32) {
      unsigned int bk0, bk1, bk2, bk3;
      int bc0, bc1, bc2, bc3;
      fetch(G + 32 + sub, bk0, bc0); fetch(G + 40 + sub, bk1, bc1); fetch(G + 48 + sub, bk2, bc2); fetch(G + 56 + sub, bk3, bc3);
#pragma clang loop unroll(disable)
      for (int s4 = 0; s4 < 4; ++s4) {
        const int g0 = G + 8 * s4;
        if (g0 >= nq) break;
        const bool ql = g0 + sub < nq;
        const unsigned int kraw = s4 == 0 ? ak0 : s4 == 1 ? ak1 : s4 == 2 ? ak2 : ak3;
        const int craw = s4 == 0 ? ac0 : s4 == 1 ? ac1 : s4 == 2 ? ac2 : ac3;
        const unsigned int key = ql ? kraw : 0xFFFFFFFFu;
        const int cnt = ql ? craw : 0;
        const int qme = min(g0 + sub, qLast);
        const unsigned ob = (obsb[qme >> 5] >> (qme & 31)) & 1u;               // != 0: the lane's query has observations (:2565: only those block a slot)
        const unsigned long long obsMask = __ballot(ob != 0);
        const bool valid = key != 0xFFFFFFFFu;
        const unsigned int myk = key & 0xFFFFu;
        bool blocked = valid && ((blk[myk >> 5] >> (myk & 31)) & 1u);       // as of the start of the group; claims inside it: below
        const int gend = min(8, nq - g0);
        {   // all eight queries at once: each takes the first of its listed candidates that was free when the group started.  That IS the
            // sequential outcome unless two accepted queries of the group want the same slot (the later one must then see the claim, or
            // overwrite it) or a query has run out of listed candidates with more in its window -- such a group is replayed one by one.
            const unsigned long long fr = __ballot(valid && !blocked);
            const unsigned m8 = (unsigned)(fr >> (8 * sub)) & 0xFFu;
            const unsigned pick = (unsigned)__shfl((int)key, (sub << 3) + (m8 ? __ffs((int)m8) - 1 : 0));
            const bool qlive = g0 + sub < nq;
            const bool take = qlive && m8 != 0 && (pick >> 21) <= 100u;    // TH_HIGH (:2589)
            const unsigned pk = pick & 0xFFFFu;
            bool clash = qlive && m8 == 0 && cnt > TK_K;
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const unsigned ke = (unsigned)__builtin_amdgcn_readlane((int)pk, 8 * e);
                const int te = __builtin_amdgcn_readlane((int)take, 8 * e);
                if (te && take && sub > e && ke == pk) clash = true;
            }
            if (!__any(clash)) {
                const bool lead = (lane & 7) == 0 && take;
                const unsigned bin = (pick >> 16) & 31u;
                const bool withBin = lead && check_ori && bin != TK_NOBIN;
                const unsigned long long tb = __ballot(lead), bb = __ballot(withBin);
                if (lead) {
                    mrow[pk] = g0 + sub;
                    if (ob != 0) atomicOr(&blk[pk >> 5], 1u << (pk & 31));
                }
                if (withBin) {
                    acc[nacc + __popcll(bb & ((1ull << lane) - 1ull))] = pk | (bin << 16);
                    atomicAdd(&hist[bin], 1u);
                }
                nm += __popcll(tb);
                nacc += __popcll(bb);
                continue;
            }
        }
        for (int s = 0; s < gend; ++s) {
            const int qi = g0 + s;
            const unsigned long long sel = 0xFFull << (8 * s);
            const unsigned long long bal = __ballot(valid && !blocked) & sel;
            unsigned int best = 0xFFFFFFFFu;
            if (bal) best = (unsigned int)__builtin_amdgcn_readlane((int)key, __ffsll((long long)bal) - 1);   // (wave-uniform lane: no LDS crossbar round trip)
            else {
                const int c = __builtin_amdgcn_readlane(cnt, 8 * s);
                if (c > TK_K) {
                    best = tk_rescan(kps, desc, cap, grid_start, grid_idx, min_x, min_y, inv_w, inv_h, qf, tf, qi, th, st, dx, dy, factor, blk, lane);
                }
            }
            if (best == 0xFFFFFFFFu) continue;
            const int d = (int)(best >> 21);
            if (d > 100) continue;                                          // TH_HIGH (:2589)
            const unsigned int k = best & 0xFFFFu, bin = (best >> 16) & 31u;
            const bool obs = (obsMask >> (8 * s)) & 1ull;
            if (obs && valid && myk == k) blocked = true;                  // later queries of this group see the claim
            if (lane == 0) {
                // LDS updates as returnless atomics: nothing in the chain of the next query waits for them
                mrow[k] = qi;
                if (obs) atomicOr(&blk[k >> 5], 1u << (k & 31));
                if (check_ori && bin != TK_NOBIN) { acc[nacc] = k | (bin << 16); atomicAdd(&hist[bin], 1u); }
            }
            ++nm;
            if (check_ori && bin != TK_NOBIN) ++nacc;
        }
      }
      ak0 = bk0; ak1 = bk1; ak2 = bk2; ak3 = bk3; ac0 = bc0; ac1 = bc1; ac2 = bc2; ac3 = bc3;
    }
    __syncthreads();
    if (check_ori) {
        // ComputeThreeMaxima (ORBmatcher.cc:2870-2909) on the bin counts, then every assignment of the other bins is cleared (:2696-2707)
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < 30; ++i) {
            const int sz = (int)hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
            else if (sz > max3) { max3 = sz; i3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) i3 = -1;
        int pruned = 0;
        for (int e = lane; e < nacc; e += 64) {
            const unsigned int v = acc[e];
            const int bin = (int)(v >> 16), k = (int)(v & 0xFFFFu);
            if (bin != i1 && bin != i2 && bin != i3) { mrow[k] = -2; ++pruned; }      // ORBM_MATCH_PRUNED
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pruned += __shfl_xor(pruned, o);
        nm -= pruned;
    }
    if (lane == 0) nmatches[pair] = nm;
}

#endif  /* ORBX_AB */

__global__ __launch_bounds__(64) void k_track_claim64(const KpIn* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                    const int* __restrict__ counts, int cap, const int* __restrict__ grid_start,
                                                    const int* __restrict__ grid_idx, float min_x, float min_y, float inv_w, float inv_h,
                                                    int q_first, int t_first, float th, ScaleTab st, float dx, float dy, float factor,
                                                    const int* __restrict__ topCnt, const unsigned int* __restrict__ topKeys,
                                                    const uint8_t* __restrict__ t_blocked, const uint8_t* __restrict__ q_obs, int check_ori,
                                                    unsigned int* __restrict__ accepted, int* __restrict__ match, int* __restrict__ nmatches) {
    extern __shared__ unsigned int tk_lds[];                                // blocked bit array [ceil(cap / 32)], hist[32], "query has observations" bits [ceil(cap / 32)]
    const int lane = threadIdx.x, pair = blockIdx.x;
    const int qf = q_first + pair, tf = t_first + pair;
    const int nq = min(counts[qf], cap), nt = min(counts[tf], cap);
    const int nwords = (cap + 31) >> 5;
    unsigned int* blk = tk_lds;
    unsigned int* hist = tk_lds + nwords;
    unsigned int* obsb = hist + 32;
    unsigned int* tag = obsb + nwords;                                      // [cap]: per slot the latest proposal (round << 6 | 63 - lane), see below
    int* mrow = match + (size_t)pair * cap;
    unsigned int* acc = accepted + (size_t)pair * cap;
    for (int w = lane; w < nwords; w += 64) {
        unsigned int bits = 0;
        if (t_blocked) {
            const uint8_t* tb = t_blocked + (size_t)tf * cap;
            const int last = max(nt - 1, 0);
            unsigned v[32];                                                 // 32 byte loads in flight (clamped, not predicated), then the bits
#pragma unroll
            for (int b = 0; b < 32; ++b) v[b] = tb[min(w * 32 + b, last)];
#pragma unroll
            for (int b = 0; b < 32; ++b) if (w * 32 + b < nt && v[b]) bits |= 1u << b;
        }
        blk[w] = bits;
        unsigned int ob = 0xFFFFFFFFu;                                      // q_obs absent: every query counts as observed
        if (q_obs) {
            const uint8_t* qo = q_obs + (size_t)qf * cap;
            const int last = max(nq - 1, 0);
            unsigned v[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) v[b] = qo[min(w * 32 + b, last)];
            ob = 0;
#pragma unroll
            for (int b = 0; b < 32; ++b) if (v[b]) ob |= 1u << b;
        }
        obsb[w] = ob;
    }
    if (lane < 32) hist[lane] = 0;
    for (int k = lane; k < cap; k += 64) { mrow[k] = -1; tag[k] = 0; }       // ORBM_NO_MATCH
    __syncthreads();
    const size_t rowBase = (size_t)pair * cap;
    int nm = 0, nacc = 0;
    // SIXTY-FOUR queries per step, one per lane with its eight listed candidates in registers.  Every lane proposes its first candidate
    // that is free right now; per slot the lowest lane wins (LDS atomicMax of round << 6 | 63 - lane, then a read-back).  The queries
    // below the first loser f take their proposals at once -- none of them wanted a slot a lower query took, and everything listed before
    // a proposal was blocked already, so this IS what the one-by-one replay (:2555-2593) does for them -- query f is resolved alone with
    // those claims applied (its proposal is gone, or its list ran dry with more in the window: tk_rescan), and the rest of the 64 propose
    // again.  Lists come from another XCD's writes, i.e. from memory: the next 64 queries' loads are in flight while these are resolved;
    // the loaded values are touched only when their window becomes current, and the loads are unconditional (clamped index).
    const int qLast = max(nq - 1, 0);
    uint4 alo, ahi, blo, bhi;
    int ac, bc;
    auto fetch = [&](int qi, uint4& lo_, uint4& hi_, int& c_) {
        const int qc = min(qi, qLast);
        const uint4* kp = (const uint4*)(topKeys + (rowBase + qc) * TK_K);
        lo_ = kp[0]; hi_ = kp[1];
        c_ = topCnt[rowBase + qc];
    };
    fetch(lane, alo, ahi, ac);
    unsigned int round = 1;
    for (int W0 = 0; W0 < nq; W0 += 64) {
        fetch(W0 + 64 + lane, blo, bhi, bc);
        const int q = W0 + lane;
        const bool ql = q < nq;
        unsigned int key[TK_K] = {alo.x, alo.y, alo.z, alo.w, ahi.x, ahi.y, ahi.z, ahi.w};
#pragma unroll
        for (int i = 0; i < TK_K; ++i) if (!ql) key[i] = 0xFFFFFFFFu;
        const int cnt = ql ? ac : 0;
        const int qme = min(q, qLast);
        const bool ob = (obsb[qme >> 5] >> (qme & 31)) & 1u;                // the lane's query has observations (:2565: only those block a slot)
        int start = 0;
        while (start < 64 && W0 + start < nq) {
            unsigned int pick = 0xFFFFFFFFu;
#pragma unroll
            for (int i = TK_K - 1; i >= 0; --i) {
                const unsigned int k = key[i] & 0xFFFFu;
                const bool fr = key[i] != 0xFFFFFFFFu && !((blk[k >> 5] >> (k & 31)) & 1u);
                pick = fr ? key[i] : pick;
            }
            const bool active = ql && lane >= start;
            const bool take = active && pick != 0xFFFFFFFFu && (pick >> 21) <= 100u;   // TH_HIGH (:2589)
            const bool resc = active && pick == 0xFFFFFFFFu && cnt > TK_K;
            const unsigned int pk = pick & 0xFFFFu;
            const unsigned int tv = (round << 6) | (unsigned)(63 - lane);
            if (take) atomicMax(&tag[pk], tv);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool lose = take && tag[pk] != tv;
            const unsigned long long bad = __ballot(lose || resc);
            const int f = bad ? __ffsll((long long)bad) - 1 : 64;          // >= start
            {
                const bool com = take && lane < f;
                const unsigned bin = (pick >> 16) & 31u;
                const bool withBin = com && check_ori && bin != TK_NOBIN;
                const unsigned long long tb = __ballot(com), bb = __ballot(withBin);
                if (com) {
                    mrow[pk] = q;
                    if (ob) atomicOr(&blk[pk >> 5], 1u << (pk & 31));
                }
                if (withBin) {
                    acc[nacc + __popcll(bb & ((1ull << lane) - 1ull))] = pk | (bin << 16);
                    atomicAdd(&hist[bin], 1u);
                }
                nm += __popcll(tb);
                nacc += __popcll(bb);
            }
            if (f < 64) {
                // query W0 + f alone (wave-uniform), the claims above applied
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int qi = W0 + f;
                unsigned int best = 0xFFFFFFFFu;
#pragma unroll
                for (int i = TK_K - 1; i >= 0; --i) {
                    const unsigned int ki = (unsigned int)__builtin_amdgcn_readlane((int)key[i], f);
                    const unsigned int k = ki & 0xFFFFu;
                    const bool fr = ki != 0xFFFFFFFFu && !((blk[k >> 5] >> (k & 31)) & 1u);
                    best = fr ? ki : best;
                }
                if (best == 0xFFFFFFFFu && __builtin_amdgcn_readlane(cnt, f) > TK_K)
                    best = tk_rescan(kps, desc, cap, grid_start, grid_idx, min_x, min_y, inv_w, inv_h, qf, tf, qi, th, st, dx, dy, factor, blk, lane);
                if (best != 0xFFFFFFFFu && (best >> 21) <= 100u) {
                    const unsigned int k = best & 0xFFFFu, bin = (best >> 16) & 31u;
                    const bool obs = (obsb[qi >> 5] >> (qi & 31)) & 1u;
                    if (lane == 0) {
                        mrow[k] = qi;
                        if (obs) atomicOr(&blk[k >> 5], 1u << (k & 31));
                        if (check_ori && bin != TK_NOBIN) { acc[nacc] = k | (bin << 16); atomicAdd(&hist[bin], 1u); }
                    }
                    ++nm;
                    if (check_ori && bin != TK_NOBIN) ++nacc;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            start = f + 1;
            ++round;
        }
        alo = blo; ahi = bhi; ac = bc;
    }
    __syncthreads();
    if (check_ori) {
        // ComputeThreeMaxima (ORBmatcher.cc:2870-2909) on the bin counts, then every assignment of the other bins is cleared (:2696-2707)
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < 30; ++i) {
            const int sz = (int)hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
            else if (sz > max3) { max3 = sz; i3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) i3 = -1;
        int pruned = 0;
        for (int e = lane; e < nacc; e += 64) {
            const unsigned int v = acc[e];
            const int bin = (int)(v >> 16), k = (int)(v & 0xFFFFu);
            if (bin != i1 && bin != i2 && bin != i3) { mrow[k] = -2; ++pruned; }      // ORBM_MATCH_PRUNED
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pruned += __shfl_xor(pruned, o);
        nm -= pruned;
    }
    if (lane == 0) nmatches[pair] = nm;
}

// ------------------------------------------------------------------------------------------------
// k_bow_transform2: DBoW2 TemplatedVocabulary::transform (TemplatedVocabulary.h:1196-1262) for a batch of descriptors: at every level the
// child with the smallest Hamming distance (first minimum, strict <) is taken; the node reached at level L - levelsup is recorded.
// The tree is in the level-major layout the host builds (orbm_vocab_create): nodes renumbered breadth-first so
// that the children of a node are CONTIGUOUS rows in its own child order; info[n] = first child << 5 | child count; orig[n] = the
// vocabulary's node id.  16 lanes per descriptor (4 descriptors per wave): lane c takes child c, the 16-lane minimum of
// (distance << 5 | c) by four DPP row rotations picks the first-minimum child (strict `d < best_d` in child order,
// TemplatedVocabulary.h:1239-1250).  A wave instruction then touches the 3 lines of each of its 4 child blocks instead of 64 scattered
// rows -- at ORBvoc's size (35.6 MB of node descriptors) the L1 line rate, not the arithmetic, bounds the thread-per-descriptor kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_transform2(const uint8_t* __restrict__ desc, int n, const unsigned* __restrict__ info,
                                                        const uint8_t* __restrict__ ndesc, const int* __restrict__ orig,
                                                        const int* __restrict__ nword, const double* __restrict__ nweight,
                                                        int L, int levelsup, int* __restrict__ word_id, int* __restrict__ node_id,
                                                        double* __restrict__ weight) {
    const int c = threadIdx.x & 15;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = i < n;
    const uint4* qp = (const uint4*)(desc + (size_t)(live ? i : 0) * 32);
    const uint4 qlo = qp[0], qhi = qp[1];
    const u64 a[4] = {(u64)qlo.x | ((u64)qlo.y << 32), (u64)qlo.z | ((u64)qlo.w << 32),
                      (u64)qhi.x | ((u64)qhi.y << 32), (u64)qhi.z | ((u64)qhi.w << 32)};
    const int nid_level = L - levelsup;
    int cur = 0, nid = 0, level = 0;
    for (;;) {
        const unsigned inf = info[cur];
        const int cnt = (int)(inf & 31u), first = (int)(inf >> 5);
        // (uniform inside a 16-lane group; groups of a wave may leave at different levels: the DPP rotations stay inside a row)
        if (cnt == 0) break;
        ++level;
        unsigned best = 0xFFFFFFFFu;
        for (int c0 = 0; c0 < cnt; c0 += 16) {                             // k <= 16 in one step (DBoW2 allows k up to 20)
            unsigned key = 0xFFFFFFFFu;
            if (c0 + c < cnt) {
                const uint4* tp = (const uint4*)(ndesc + (size_t)(first + c0 + c) * 32);
                const uint4 lo = tp[0], hi = tp[1];
                const int d = ham256(a, (u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                                     (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32));
                key = ((unsigned)d << 5) | (unsigned)(c0 + c);
            }
            best = min(best, row16_min_u32(key));
        }
        cur = first + (int)(best & 31u);
        if (level == nid_level) nid = cur;
    }
    if (live && c == 0) {
        if (word_id) word_id[i] = nword[cur];
        if (weight) weight[i] = nweight[cur];
        node_id[i] = nid ? orig[nid] : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// k_undistort: Frame::UndistortKeyPoints (Frame.cc:924-970) = cv::undistortPoints(K, D, R = I, P = newK), one thread per
// keypoint, double arithmetic in OpenCV's operation order (5 fixed-point iterations; compiled without contraction).
// ------------------------------------------------------------------------------------------------
struct UndistParams { double k[14]; double fx, fy, cx, cy, nfx, nfy, ncx, ncy; };

__device__ __forceinline__ void undistort_point(const UndistParams& P, double u, double v, float* ox, float* oy) {
    const double ifx = 1.0 / P.fx, ify = 1.0 / P.fy;
    double x = (u - P.cx) * ifx, y = (v - P.cy) * ify;
    const double x0 = x, y0 = y;
    const double* k = P.k;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - P.cx) * ifx; y = (v - P.cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = P.nfx * x + 0.0 * y + P.ncx, yy = 0.0 * x + P.nfy * y + P.ncy, ww = 1.0 / (0.0 * x + 0.0 * y + 1.0);
    *ox = (float)(xx * ww); *oy = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort(const KpIn* __restrict__ in, int n, UndistParams P, int passthrough,
                                                   KpIn* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    KpIn kp = in[i];
    if (!passthrough) undistort_point(P, (double)kp.x, (double)kp.y, &kp.x, &kp.y);
    out[i] = kp;
}

// ------------------------------------------------------------------------------------------------
// k_frustum: Frame::isInFrustum (Nleft == -1, Frame.cc:603-671) + MapPoint::PredictScale (MapPoint.cc:725-740), one
// thread per map point.  Float expressions in the reference's order (Matx products accumulate from 0 in float, cv::norm
// accumulates in double); log(ratio) is evaluated in double and rounded to float (the host libm's logf is within 1 ulp of
// that; the predicted level can differ only when log(ratio)/logScaleFactor is within an ulp of an integer).
// ------------------------------------------------------------------------------------------------
struct FrustumParams { float rcw[9], tcw[3], ow[3], k[4], bounds[4], bf, cosLimit, logSF; int nLevels; };

__global__ __launch_bounds__(256) void k_frustum(int n, const float* __restrict__ pw, const float* __restrict__ normal,
                                                 const float* __restrict__ minDist, const float* __restrict__ maxDist, FrustumParams F,
                                                 uint8_t* inView, float* projX, float* projY, float* projXR, float* depth, int* level,
                                                 float* viewCos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    inView[i] = 0; projX[i] = -1.f; projY[i] = -1.f;
    const float P0 = pw[3 * i], P1 = pw[3 * i + 1], P2 = pw[3 * i + 2];
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = 0.f;
        s += F.rcw[3 * r] * P0; s += F.rcw[3 * r + 1] * P1; s += F.rcw[3 * r + 2] * P2;
        Pc[r] = s + F.tcw[r];
    }
    double n2 = 0.0;
    n2 += (double)Pc[0] * (double)Pc[0]; n2 += (double)Pc[1] * (double)Pc[1]; n2 += (double)Pc[2] * (double)Pc[2];
    const float PcDist = (float)sqrt(n2);
    const float PcZ = Pc[2];
    const float invz = 1.0f / PcZ;
    if (PcZ < 0.0f) return;
    const float u = F.k[0] * Pc[0] / Pc[2] + F.k[2], v = F.k[1] * Pc[1] / Pc[2] + F.k[3];
    if (u < F.bounds[0] || u > F.bounds[1]) return;
    if (v < F.bounds[2] || v > F.bounds[3]) return;
    projX[i] = u; projY[i] = v;
    const float maxD = 1.2f * maxDist[i], minD = 0.8f * minDist[i];
    const float O0 = P0 - F.ow[0], O1 = P1 - F.ow[1], O2 = P2 - F.ow[2];
    double d2 = 0.0;
    d2 += (double)O0 * (double)O0; d2 += (double)O1 * (double)O1; d2 += (double)O2 * (double)O2;
    const float dist = (float)sqrt(d2);
    if (dist < minD || dist > maxD) return;
    float dot = 0.f;
    dot += O0 * normal[3 * i]; dot += O1 * normal[3 * i + 1]; dot += O2 * normal[3 * i + 2];
    const float vc = dot / dist;
    if (vc < F.cosLimit) return;
    const float ratio = maxDist[i] / dist;
    const float lg = (float)log((double)ratio);
    int ns = (int)ceilf(lg / F.logSF);
    if (ns < 0) ns = 0; else if (ns >= F.nLevels) ns = F.nLevels - 1;
    inView[i] = 1; projXR[i] = u - F.bf * invz; depth[i] = PcDist; level[i] = ns; viewCos[i] = vc;
}

// k_gather_rows: packs the used prefix of every row of the two [nq][cap] candidate arrays into [nq][maxc] (one contiguous
// device-to-host copy instead of nq*cap entries or a strided copy).
__global__ __launch_bounds__(256) void k_gather_rows(const int* __restrict__ idx, const int* __restrict__ dist, int nq, int cap, int maxc,
                                                     int* __restrict__ oidx, int* __restrict__ odist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * maxc) return;
    const int q = i / maxc, k = i - q * maxc;
    oidx[i] = idx[(size_t)q * cap + k];
    odist[i] = dist[(size_t)q * cap + k];
}

}  // namespace orbmk
