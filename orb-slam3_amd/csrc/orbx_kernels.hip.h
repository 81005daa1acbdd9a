// orbx_kernels.hip.h -- hand-written gfx950 kernels of the ORB extractor.
//
// Pipeline per batch (all kernels are launched over (work-item, frame) so a batch fills the chip):
//   k_resize2     x(L-1)  level l-1 -> level l, OpenCV INTER_LINEAR fixed point  (ORBextractor.cc:1664-1717)   [k_resize: general fallback]
//   k_fast3<P>    x3      per 35-px cell: FAST-9/16 score, 3x3 NMS, ini/min threshold   (:1038-1143)          [+ k_fast_fix; k_fast: A/B]
//   k_quadtree2   x1      DistributeOctTree per (frame, level)                            (:688-1034)          [k_quadtree: A/B]
//   k_slots       x1      octave/size fix-up, level-0 scaling, lapping partition          (:1161-1176,1633-1655)
//   k_blur3       x1      7x7 Gaussian, sigma 2, reflect-101, bit-exact fixed point, as two int8 Toeplitz products on the
//                         matrix cores (v_mfma_i32_32x32x32_i8)                          (:1606-1614)         [k_blur2: VALU A/B]
//   k_orient_desc2 x1     IC_Angle + 256-bit steered BRIEF, four keypoints per wavefront  (:91-203)            [k_orient_desc: A/B]
// Ingest: k_gray (cvtColor), k_remap (stereo rectification), k_clahe_lut / k_clahe_apply.
// Plumbing: k_stamp (span boundaries of a graph replay), k_copy_out (results to pinned host memory, A/B path).
// Integer / byte work throughout (the pass is priced against HBM, the kernels are integer-issue bound); the
// matrix cores only where the work is an exact int8 product (the blur here, the dense Hamming 2-NN in orbm_kernels.hip.h).
// Compiled with -ffp-contract=off so the few float expressions evaluate exactly as the reference writes them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbxk {

typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
struct RzTab { int s; short a0, a1; };                     // source index + two Q11 taps (8 bytes)

struct LevelDesc {                                         // one pyramid level of the current geometry
    int w, h, pitch;
    int off;                                               // byte offset inside a frame's pyramid slab (levels >= 1)
    int boff, btpr;                                        // blurred copy: byte offset inside a frame's blurred slab; tiles per tile row (tiled layout)
    int rzx, rzy;                                          // offsets into the resize tap tables
    int cellBase, nCells;                                  // active FAST cells of this level in the cell table
    int slotBase, slotCap;                                 // candidate slots: slotBase + cell*slotCap + k
    int N, nIni, qtW, qtH;                                 // quadtree target, roots, extent (maxX-minX, maxY-minY)
    float hX;
    int selBase, selCap;                                   // quadtree output slots inside a frame
    float sf;                                              // mvScaleFactor[level]
    float patch;                                           // (float)(int)(31*sf)
};

struct CellInfo {                                          // one FAST cell (sub-image handed to cv::FAST)
    short level, x0, y0, cw, ch, addx, addy, pad;
    int slot;                                              // index of slot 0 of this cell within the frame
    int cnt;                                               // index of the count word within the frame
};

struct TileInfo { short level, x0, y0, pad; };            // blur tiles

struct Geom {                                              // kernel argument block (passed by value)
    int nlevels, w0, h0;
    int iniTh, minTh, lowTh;
    int totalCells, totalSlots, totalSel, kpCap;
    int nodeCap, sortCap;
    size_t pyrFrameBytes;                                  // per-frame slab holding levels 1..L-1 (and a copy slot for 0)
    size_t blrFrameBytes;                                  // per-frame slab of the blurred levels
    int blurTiled;                                         // blurred levels are stored as 16 x 8-px tiles (one 128-byte line each), see k_blur3
    LevelDesc lv[12];
};

__device__ __forceinline__ us2 as_us2(u32 v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ u32 as_u32(us2 v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ us2 pkmin(us2 a, us2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ us2 pkmax(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }
// three-input forms for halves in 0..255 (see fast_score16x2_tl)
__device__ __forceinline__ us2 pkmin3(us2 a, us2 b, us2 c) {
    u32 d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(__builtin_bit_cast(u32, a)), "v"(__builtin_bit_cast(u32, b)), "v"(__builtin_bit_cast(u32, c)));
    return __builtin_bit_cast(us2, d);
}
__device__ __forceinline__ us2 pkmax3(us2 a, us2 b, us2 c) {
    u32 d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(__builtin_bit_cast(u32, a)), "v"(__builtin_bit_cast(u32, b)), "v"(__builtin_bit_cast(u32, c)));
    return __builtin_bit_cast(us2, d);
}

__device__ __forceinline__ u32 ashr_pk_u8(u32 a, u32 b, u32 sh) {  // sat_u8(a >> sh) | sat_u8(b >> sh) << 8 (arithmetic shifts; upper half 0)
    u32 d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(sh));
    return d;
}
__device__ __forceinline__ u32 mad24(u32 a, u32 b, u32 c) {      // a*b + c on 24-bit operands, one VALU instruction
    u32 d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ u32 mulhi24(u32 a_uniform, u32 b) {   // bits 47..32 of the product of two 24-bit operands (a: wave-uniform), one VALU instruction
    u32 d;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "s"(a_uniform), "v"(b));
    return d;
}

// Loads through an explicit global address space: pointers fetched from memory (the per-frame level-0 table)
// are generic to the compiler, which would emit flat_load + conservative vmcnt(0)/lgkmcnt(0) waits.
#if defined(__HIP_DEVICE_COMPILE__)
// wave-uniform base + 32-bit per-lane byte offset: lets the compiler use the SGPR-base form of global_load (no 64-bit
// per-lane address arithmetic)
__device__ __forceinline__ u32 gload32u(const void* base, u32 off) {
    return *(const __attribute__((address_space(1))) u32*)((const __attribute__((address_space(1))) u8*)base + off);
}
__device__ __forceinline__ void gstore32u(void* base, u32 off, u32 v) {
    *(__attribute__((address_space(1))) u32*)((__attribute__((address_space(1))) u8*)base + off) = v;
}
__device__ __forceinline__ uint2 gload64u_unaligned(const void* base, u32 off) {   // 8 bytes at ANY byte offset (global memory: unaligned mode)
    typedef unsigned long long __attribute__((aligned(1))) u64a1;
    const unsigned long long v = *(const __attribute__((address_space(1))) u64a1*)((const __attribute__((address_space(1))) u8*)base + off);
    return make_uint2((u32)v, (u32)(v >> 32));
}
__device__ __forceinline__ uint4 gload128u(const void* base, u32 off) {
    return *(const __attribute__((address_space(1))) uint4*)((const __attribute__((address_space(1))) u8*)base + off);
}
__device__ __forceinline__ u32 gload32(const void* p) { return *(const __attribute__((address_space(1))) u32*)p; }
__device__ __forceinline__ uint4 gload128(const void* p) { return *(const __attribute__((address_space(1))) uint4*)p; }
__device__ __forceinline__ u8 gload8(const void* p) { return *(const __attribute__((address_space(1))) u8*)p; }
#else
__device__ __forceinline__ u32 gload32u(const void* base, u32 off) { return *(const u32*)((const u8*)base + off); }
__device__ __forceinline__ uint4 gload128u(const void* base, u32 off) { return *(const uint4*)((const u8*)base + off); }
__device__ __forceinline__ uint2 gload64u_unaligned(const void* base, u32 off) { uint2 r; __builtin_memcpy(&r, (const u8*)base + off, 8); return r; }
__device__ __forceinline__ void gstore32u(void* base, u32 off, u32 v) { *(u32*)((u8*)base + off) = v; }
__device__ __forceinline__ u32 gload32(const void* p) { return *(const u32*)p; }
__device__ __forceinline__ uint4 gload128(const void* p) { return *(const uint4*)p; }
__device__ __forceinline__ u8 gload8(const void* p) { return *(const u8*)p; }
#endif

// XCD-aware work order.  Workgroup ids are dealt round-robin over the 8 XCDs (ids b and b + 8 share an XCD and its L2), so a
// task list in which NEIGHBOURING tasks touch the same memory lines (adjacent column groups / strips of one image share their
// halo lines) must be walked so that neighbours get ids that are 8 apart: id b takes the (b / 8)-th task of the (b % 8)-th
// eighth of the list.  Placement is never relied on for correctness, only for L2 hits.
__device__ __forceinline__ unsigned xcd_task(unsigned b, unsigned n) {
    const unsigned q = n >> 3, rem = n & 7u, x = b & 7u;
    return x * q + (x < rem ? x : rem) + (b >> 3);
}

__device__ __forceinline__ const u8* level_ptr(const Geom& g, const u8* const* l0, int l0pitch,
                                               const u8* pyr, int frame, int level, int* pitch) {
    if (level == 0) {
        *pitch = l0pitch;
        return l0[frame];
    }
    *pitch = g.lv[level].pitch;
    return pyr + (size_t)frame * g.pyrFrameBytes + g.lv[level].off;
}

// ------------------------------------------------------------------------------------------------
// k_resize: dst level from src level.  cv::resize INTER_LINEAR 8UC1:
//   H[r][dx] = src[r][sx]*a0 + src[r][sx+1]*a1 ; dst = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2
// block (64,4): each thread makes 4 consecutive dst pixels of one row and stores them as one dword.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(Geom g, const u8* const* l0, int l0pitch, u8* pyr, int level,
                                                const RzTab* __restrict__ xt, const RzTab* __restrict__ yt) {
    const LevelDesc& D = g.lv[level];
    const LevelDesc& S = g.lv[level - 1];
    const int frame = blockIdx.z;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= D.h || dx0 >= D.w) return;
    int sp;
    const u8* src = level_ptr(g, l0, l0pitch, pyr, frame, level - 1, &sp);
    u8* dst = pyr + (size_t)frame * g.pyrFrameBytes + D.off;
    const RzTab ty = yt[D.rzy + dy];
    const int sy0 = min(max(ty.s, 0), S.h - 1), sy1 = min(max(ty.s + 1, 0), S.h - 1);
    const u8* r0 = src + (size_t)sy0 * sp;
    const u8* r1 = src + (size_t)sy1 * sp;
    u32 packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = dx0 + i;
        if (dx < D.w) {
            const RzTab tx = xt[D.rzx + dx];
            const int sx1 = min(tx.s + 1, S.w - 1);
            const int h0 = r0[tx.s] * tx.a0 + r0[sx1] * tx.a1;
            const int h1 = r1[tx.s] * tx.a0 + r1[sx1] * tx.a1;
            int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            packed |= (u32)v << (8 * i);
        }
    }
    u8* o = dst + (size_t)dy * D.pitch + dx0;
    if (dx0 + 3 < D.pitch) *(u32*)o = packed;              // pitch is a multiple of 64: the padding absorbs the tail
    else for (int i = 0; i < 4 && dx0 + i < D.w; ++i) o[i] = (u8)(packed >> (8 * i));
}

// ------------------------------------------------------------------------------------------------
// k_resize2: streaming form of the same arithmetic.  One wavefront owns 256 destination pixels of a row
// (4 px per lane, stored as one dword) and walks RZ_R destination rows downwards.  Per SOURCE row a lane
// loads the 8 bytes that start at its first tap (one unaligned global_load_dwordx2 from the wave-uniform row base:
// the four tap pairs of a lane span < 8 bytes for scale factors up to 1.5); the pair of pixel i is picked and widened
// by ONE v_perm with a per-lane selector and reduced with one v_dot2_u32_u16 against (a0,a1).  The
// horizontal result of source row sy+1 is reused as row sy of the next destination row (the reference's
// cv::resize keeps the same two-row cache).  Row taps come from the scalar unit (wave-uniform).
// ------------------------------------------------------------------------------------------------
#define RZ_R 8
#define RZ_SRC 12                                          // source rows one task may touch (host checks)
struct RzX4 { int bs; u8 o[4]; u32 a[4]; };                // per destination dword: byte offset of the first tap, tap offsets from it (<= 6), (2*a0 | 2*a1<<16): DOUBLED Q11 taps
struct RzTask { short level, g0, y0, pad; };

__global__ __launch_bounds__(256) void k_resize2(Geom g, const u8* const* l0, int l0pitch, u8* pyr,
                                                 const RzTask* __restrict__ tasks, int ntasks,
                                                 const RzX4* __restrict__ x4, const RzTab* __restrict__ yt) {
    const int lane = threadIdx.x & 63;
    const int ti = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ti >= ntasks) return;
    RzTask t = tasks[ti];
    const int level = __builtin_amdgcn_readfirstlane(t.level);
    const int g0 = __builtin_amdgcn_readfirstlane(t.g0), y0 = __builtin_amdgcn_readfirstlane(t.y0);
    const int frame = blockIdx.y;
    const LevelDesc& D = g.lv[level];
    int sp;
    const u8* src = level_ptr(g, l0, l0pitch, pyr, frame, level - 1, &sp);
    u8* dst = pyr + (size_t)frame * g.pyrFrameBytes + D.off;
    const int gcol = g0 + lane;
    const int ndw = (D.w + 3) >> 2;
    const bool act = gcol < ndw;
    const RzX4 X = x4[D.rzx / 4 + (act ? gcol : ndw - 1)];   // rzx is the level's offset in pixels; the x4 table is per dword
    // the 8-byte window never leaves the row's pitch: near the right edge it slides left and the selectors follow
    const int bsc = min(X.bs, sp - 8);
    const u32 delta = (u32)(X.bs - bsc);                      // 0..6; every tap index + delta stays <= 7 (taps lie inside the row)
    u32 sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sel[i] = 0x0c000c00u | ((u32)X.o[i] + delta) | (((u32)X.o[i] + delta + 1u) << 16);
    const u32 colo = (u32)bsc;
    const int yend = min(y0 + RZ_R, D.h);
    const int sFirst = yt[D.rzy + y0].s;
    const int nsrc = yt[D.rzy + yend - 1].s + 2 - sFirst;               // host guarantees 0 <= s, s+1 < S.h, nsrc <= RZ_SRC
    // all source rows of the task in flight before any arithmetic
    uint2 d[RZ_SRC];
#pragma unroll
    for (int j = 0; j < RZ_SRC; ++j) {
        d[j] = make_uint2(0, 0);
        if (j < nsrc) d[j] = gload64u_unaligned(src, (u32)((sFirst + j) * sp) + colo);     // level base + 32-bit offset (row part is wave-uniform)
    }
    int Hp[4] = {0, 0, 0, 0};
    int dy = y0;
#pragma unroll
    for (int j = 0; j < RZ_SRC; ++j) {
        if (j < nsrc) {
            int Hc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const us2 s2 = as_us2(__builtin_amdgcn_perm(d[j].y, d[j].x, sel[i]));      // (byte o, byte o+1) as a u16 pair
                // only (H >> 4) is ever used; it is kept as (H >> 4) << 5 -- the table's taps are doubled, so that is one AND -- for the
                // 24-bit high multiply below
                Hc[i] = (int)(__builtin_amdgcn_udot2(s2, as_us2(X.a[i]), 0u, false) & ~31u);
            }
            if (dy < yend) {
                const RzTab ty = yt[D.rzy + dy];
                if (ty.s + 1 == sFirst + j) {                       // rows (sy, sy+1) = (j-1, j) are both here: emit dy
                    // (b * (H >> 4)) >> 16 as ONE v_mul_hi_u32_u24: (b << 11) * ((H >> 4) << 5) = b * (H >> 4) * 2^16, whose bits 47..32 are
                    // the wanted quotient (b <= 2048 -> 22 bits, (H >> 4) << 5 < 2^20: both operands fit 24 bits, nothing is rounded)
                    const u32 A0 = (u32)ty.a0 << 11, A1 = (u32)ty.a1 << 11;
                    u32 sum[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum[i] = mulhi24(A0, (u32)Hp[i]) + mulhi24(A1, (u32)Hc[i]) + 2u;
                    // each term is <= 1020, so (sum + 2) >> 2 <= 255: cv::resize's saturate_cast is the identity here.  gfx950's
                    // v_ashr_pk_u8_i32 shifts two sums and packs them as two bytes in one instruction (tools/ubench/isa_probe.hip)
                    const u32 packed = ashr_pk_u8(sum[0], sum[1], 2u) | (ashr_pk_u8(sum[2], sum[3], 2u) << 16);
                    if (act) gstore32u(dst, (u32)(dy * D.pitch) + (u32)gcol * 4u, packed);
                    ++dy;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) Hp[i] = Hc[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 score  S = max(A,B) - 1  (cv::cornerScore<16>), branch-free:
//   B' = max_k min(ring[k..k+8]) ,  A' = min_k max(ring[k..k+8]) ;  A = v - A' , B = B' - v
// The 16 nine-windows are taken in pairs that share eight pixels W_j = ring[2j..2j+7]:
//   max(min(W_j, r[2j+8]), min(r[2j-1], W_j)) = min(W_j, max(r[2j+8], r[2j-1]))
// so B' = max_j min3(m4[j], m4[j+2], max(r[2j-1], r[2j+8])) with m4[j] = min(ring[2j..2j+3]) -- 36 min/max per side.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int fast_score16(const u8* t, int p) {
    // row bases at column -3 so that every LDS offset is a non-negative immediate
    const u8* b0 = t - 3 * p - 3;
    const u8 *b1 = b0 + p, *b2 = b1 + p, *b3 = b2 + p, *b4 = b3 + p, *b5 = b4 + p, *b6 = b5 + p;
    int r[16];
    r[0] = b6[3];   r[1] = b6[4];   r[2] = b5[5];   r[3] = b4[6];
    r[4] = b3[6];   r[5] = b2[6];   r[6] = b1[5];   r[7] = b0[4];
    r[8] = b0[3];   r[9] = b0[2];   r[10] = b1[1];  r[11] = b2[0];
    r[12] = b3[0];  r[13] = b4[0];  r[14] = b5[1];  r[15] = b6[2];
    const int v = b3[3];
    int m2[8], M2[8], m4[8], M4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m2[j] = min(r[2 * j], r[2 * j + 1]); M2[j] = max(r[2 * j], r[2 * j + 1]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { m4[j] = min(m2[j], m2[(j + 1) & 7]); M4[j] = max(M2[j], M2[(j + 1) & 7]); }
    int bmax = 0, amin = 255;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int a = r[(2 * j + 15) & 15], c = r[(2 * j + 8) & 15];
        const int u = min(min(m4[j], m4[(j + 2) & 7]), max(a, c));
        const int U = max(max(M4[j], M4[(j + 2) & 7]), min(a, c));
        bmax = max(bmax, u);
        amin = min(amin, U);
    }
    return max(v - amin, bmax - v) - 1;
}

#ifdef ORBX_AB   /* A/B reference, not in the product library */
// Two pixels per lane: the same network on packed 16-bit halves (v_pk_min/max_u16 have no 3-input form, so 47 packed ops
// per side serve two pixels: 47 per pixel against 72).  The ring bytes of pixel A land in the low halves, those of
// pixel B in the high halves (ds_read_u8_d16 / _d16_hi).  Returns the two scores as signed 16-bit halves.
__device__ __forceinline__ ss2 fast_score16x2(const u8* ta, const u8* tb, int p) {
    // (an 8-byte-per-row variant with unaligned ds_read_b64 -- 7 LDS reads per pixel instead of 17 -- is bit-exact but 1.7x
    // slower on gfx950: unaligned LDS reads are split.  The scattered byte reads bound this phase, not the min/max network.)
    const u8* a0 = ta - 3 * p - 3; const u8* b0 = tb - 3 * p - 3;
    const u8 *a1 = a0 + p, *a2 = a1 + p, *a3 = a2 + p, *a4 = a3 + p, *a5 = a4 + p, *a6 = a5 + p;
    const u8 *b1 = b0 + p, *b2 = b1 + p, *b3 = b2 + p, *b4 = b3 + p, *b5 = b4 + p, *b6 = b5 + p;
#define F2(ra, rb, o) us2{(u16)(ra)[o], (u16)(rb)[o]}
    us2 r[16];
    r[0] = F2(a6, b6, 3);   r[1] = F2(a6, b6, 4);   r[2] = F2(a5, b5, 5);   r[3] = F2(a4, b4, 6);
    r[4] = F2(a3, b3, 6);   r[5] = F2(a2, b2, 6);   r[6] = F2(a1, b1, 5);   r[7] = F2(a0, b0, 4);
    r[8] = F2(a0, b0, 3);   r[9] = F2(a0, b0, 2);   r[10] = F2(a1, b1, 1);  r[11] = F2(a2, b2, 0);
    r[12] = F2(a3, b3, 0);  r[13] = F2(a4, b4, 0);  r[14] = F2(a5, b5, 1);  r[15] = F2(a6, b6, 2);
    const us2 v = F2(a3, b3, 3);
#undef F2
    us2 m2[8], M2[8], m4[8], M4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m2[j] = pkmin(r[2 * j], r[2 * j + 1]); M2[j] = pkmax(r[2 * j], r[2 * j + 1]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { m4[j] = pkmin(m2[j], m2[(j + 1) & 7]); M4[j] = pkmax(M2[j], M2[(j + 1) & 7]); }
    // three-input packed minimum / maximum (gfx950: v_pk_minimum3_f16 / v_pk_maximum3_f16).  The halves hold 0..255: as binary16 bit
    // patterns those are non-negative subnormals, which order exactly like the integers (the kernels run with f16 denormals preserved,
    // amdhsa_float_denorm_mode_16_64 = 3; checked against the two-input integer network on the GPU by the parity tests and the fuzz)
    us2 u[8], U[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const us2 a = r[(2 * j + 15) & 15], c = r[(2 * j + 8) & 15];
        u[j] = pkmin3(m4[j], m4[(j + 2) & 7], pkmax(a, c));
        U[j] = pkmax3(M4[j], M4[(j + 2) & 7], pkmin(a, c));
    }
    const us2 bmax = pkmax3(pkmax3(u[0], u[1], u[2]), pkmax3(u[3], u[4], u[5]), pkmax(u[6], u[7]));
    const us2 amin = pkmin3(pkmin3(U[0], U[1], U[2]), pkmin3(U[3], U[4], U[5]), pkmin(U[6], U[7]));
    const ss2 d1 = __builtin_bit_cast(ss2, v - amin), d2 = __builtin_bit_cast(ss2, bmax - v);   // |.| <= 255: exact as signed 16-bit
    return __builtin_elementwise_max(d1, d2) - ss2{1, 1};
}
#endif  /* ORBX_AB */

// ------------------------------------------------------------------------------------------------
// k_fast: one workgroup per FAST cell (the sub-image the reference hands to cv::FAST).
// LDS: the (cw x ch) byte tile and the score tile.  Non-max suppression never looks across the cell
// (cv::FAST zero-fills outside the sub-image's 3-px frame), and the ini->min threshold fallback is
// decided AFTER suppression (ORBextractor.cc:1118).  Because corner@t <=> S >= t, the set of 3x3 strict
// local maxima is threshold independent: keypoints@ini = maxima with S >= ini, else maxima with S >= min.
// Output: packed (x | y<<12 | S<<24) in row-major order into the cell's slot array + its count.
// ------------------------------------------------------------------------------------------------
#define ORBX_FAST_TILE 6400
__device__ __forceinline__ void fast_cell_block(const Geom& g, const u8* const* l0, int l0pitch, const u8* pyr,
                                                const CellInfo* __restrict__ cells, u32* candCnt, u32* candEnt,
                                                int* err, int cellIdx, int frame) {
    __shared__ u8 tile[ORBX_FAST_TILE];
    __shared__ u8 sc[ORBX_FAST_TILE];
    __shared__ int s_cntIni;
    __shared__ int s_wave[4];
    const CellInfo c = cells[cellIdx];
    const int tid = threadIdx.x;
    const int cw = c.cw, ch = c.ch;
    int sp;
    const u8* src = level_ptr(g, l0, l0pitch, pyr, frame, c.level, &sp) + (size_t)c.y0 * sp + c.x0;
    const int npx = cw * ch;
    for (int i = tid; i < npx; i += 256) {
        const int y = i / cw, x = i - y * cw;
        tile[i] = src[(size_t)y * sp + x];
        sc[i] = 0;
    }
    if (tid == 0) s_cntIni = 0;
    __syncthreads();
    const int vw = cw - 6, vh = ch - 6;                    // valid detection window (3-px frame excluded)
    const int nv = vw > 0 && vh > 0 ? vw * vh : 0;
    for (int i = tid; i < nv; i += 256) {
        const int y = i / vw + 3, x = i - (y - 3) * vw + 3;
        const int s = fast_score16(tile + y * cw + x, cw);
        sc[y * cw + x] = (u8)(s >= g.lowTh ? s : 0);
    }
    __syncthreads();
    // strict 3x3 maxima (frame of the tile holds zeros); flags kept in `tile` (no longer needed)
    for (int i = tid; i < nv; i += 256) {
        const int y = i / vw + 3, x = i - (y - 3) * vw + 3;
        const u8* q = sc + y * cw + x;
        const int s = q[0];
        const bool keep = s > 0 && s > q[-1] && s > q[1] && s > q[-cw - 1] && s > q[-cw] && s > q[-cw + 1] &&
                          s > q[cw - 1] && s > q[cw] && s > q[cw + 1];
        tile[y * cw + x] = keep ? 1 : 0;
        if (keep && s >= g.iniTh) atomicAdd(&s_cntIni, 1);
    }
    __syncthreads();
    const int thr = s_cntIni > 0 ? g.iniTh : g.minTh;
    const LevelDesc& L = g.lv[c.level];
    u32* out = candEnt + (size_t)frame * g.totalSlots + c.slot;
    const int lane = tid & 63, wv = tid >> 6;
    int base = 0;
    for (int i0 = 0; i0 < nv; i0 += 256) {                 // ordered (row-major) compaction, 256 px per step
        const int i = i0 + tid;
        bool f = false;
        int x = 0, y = 0, s = 0;
        if (i < nv) {
            y = i / vw + 3; x = i - (y - 3) * vw + 3;
            s = sc[y * cw + x];
            f = tile[y * cw + x] && s >= thr;
        }
        const unsigned long long m = __ballot(f);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wv; ++k) off += s_wave[k];
        const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (f) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < L.slotCap) out[pos] = (u32)(x + c.addx) | ((u32)(y + c.addy) << 12) | ((u32)s << 24);
            else atomicExch(err, 1);
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) candCnt[(size_t)frame * g.totalCells + c.cnt] = (u32)base;
}

#ifdef ORBX_AB   /* k_fast: A/B reference, not in the product library */
__global__ __launch_bounds__(256) void k_fast(Geom g, const u8* const* l0, int l0pitch, const u8* pyr,
                                              const CellInfo* __restrict__ cells, u32* candCnt, u32* candEnt,
                                              int* err) {
    fast_cell_block(g, l0, l0pitch, pyr, cells, candCnt, candEnt, err, blockIdx.x, blockIdx.y);
}

#endif  /* ORBX_AB */

// k_fast_fix: cells whose quick-reject survivors did not fit k_fast3's bounded LDS queue (rare: the queue holds 500+
// of a cell's ~1400 pixels; a dense-corner texture is needed) are redone here by the per-cell kernel body, which has no
// queue.  ovf[0] = number of entries, ovf[1] = workgroups finished (the last one re-arms both for the next batch).
__global__ __launch_bounds__(256) void k_fast_fix(Geom g, const u8* const* l0, int l0pitch, const u8* pyr,
                                                  const CellInfo* __restrict__ cells, u32* candCnt, u32* candEnt,
                                                  int* err, u32* ovf, const u32* __restrict__ ovfList) {
    const u32 n = __builtin_amdgcn_readfirstlane(ovf[0]);
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u32 e = ovfList[i];
        const int frame = (int)(e / (u32)g.totalCells);
        fast_cell_block(g, l0, l0pitch, pyr, cells, candCnt, candEnt, err, (int)(e - (u32)frame * g.totalCells), frame);
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&ovf[1], 1u) == gridDim.x - 1) { ovf[0] = 0; ovf[1] = 0; }
}

struct StripInfo { short level, ncell, x0, y0, w, h, xal, lp; int cell0; };   // lp = LDS tile pitch in bytes

#define F3_NT 256
#ifndef F3_XCD
#define F3_XCD 1                                            // neighbouring strips (they share 6 columns and whole lines) on one XCD
#endif
#ifdef ORBX_AB   /* k_fast3: A/B reference (round 2's FAST), not in the product library */
// ------------------------------------------------------------------------------------------------
// k_fast3: strip tile shared by the workgroup, but every FAST cell is processed by ONE wavefront with no
// workgroup barrier after the tile load (k_fast2 spent its time parked at 7 barriers with 2 WGs/CU).
// Per cell the wave replays the reference literally: cv::FAST at iniTh, and only if that leaves the cell
// empty AFTER non-max suppression, cv::FAST at minTh (ORBextractor.cc:1112-1125):
//   quick reject (packed 16-bit, 8 px per lane, sign bits gathered by v_perm) -> DPP prefix scan -> the wave's LDS queue
//   -> exact score on dense lanes, in-place compaction of pixels with S >= t -> strict 3x3 maxima inside the cell
//   window, compacted again -> packed store (the queue stays row-major throughout, which is cv::FAST's order).
// No atomics, no bitmap: all counts live in wave-uniform registers.  The queue is bounded (occupancy); a cell whose
// survivors do not fit is handed to k_fast_fix through a global list.
// ------------------------------------------------------------------------------------------------
#ifndef F3_ASM_SCAN
#define F3_ASM_SCAN 1
#endif
// PITCH > 0: every strip of the launch uses this LDS tile pitch (bytes), so that row offsets become instruction immediates
// (18 address adds per scored pixel pair, and the +-3-row reads of the quick test); PITCH == 0: per-strip pitch st.lp (wide cells).
template <int PITCH>
__global__ __launch_bounds__(F3_NT) void k_fast3(Geom g, const u8* const* l0, int l0pitch, const u8* pyr,
                                                 const CellInfo* __restrict__ cells, const StripInfo* __restrict__ strips,
                                                 u32* candCnt, u32* candEnt, int* err, int tileBytes, int qcap,
                                                 u32* ovf, u32* ovfList) {
    extern __shared__ __attribute__((aligned(16))) unsigned char f3smem[];
    u8* img = f3smem;
    u8* sc = f3smem + tileBytes;
    StripInfo st = strips[gridDim.x - 1 - (F3_XCD ? xcd_task(blockIdx.x, gridDim.x) : blockIdx.x)];   // coarser (denser, slower) strips of the group first: a lighter tail
    st.level = (short)__builtin_amdgcn_readfirstlane(st.level);
    const int frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave-uniform: per-cell metadata comes through the scalar cache
    u16* q = (u16*)(f3smem + 2 * tileBytes) + wv * qcap;
    const int Pb = PITCH > 0 ? PITCH : (int)st.lp, H = st.h;              // tile pitch in bytes
    // this wave's first cell: fetched now so that its latency hides behind the tile load
    CellInfo cell0 = cells[st.cell0 + min(wv, st.ncell - 1)];
    int sp;
    const u8* src = level_ptr(g, l0, l0pitch, pyr, frame, st.level, &sp);
    const LevelDesc& L = g.lv[st.level];
    {   // tile load: 32 lanes x 16 B per row; the loads of 6 row-passes (48 rows at 256 threads) are issued back to back
        // before the first LDS write (one memory round trip instead of one per pass)
        const int cpr = Pb >> 4;
        const int rowLimit = min((L.w + 15) & ~15, sp);
        const int ck = tid & 31;
        const int gx = st.xal + ck * 16;
        const bool colOk = ck < cpr;
        const bool ldOk = colOk && gx < rowLimit;
        const int RP = F3_NT / 32;
        uint4 v[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int r = (tid >> 5) + p * RP;
            v[p] = make_uint4(0, 0, 0, 0);
            if (ldOk && r < H) v[p] = gload128u(src, (u32)(__mul24(st.y0 + r, sp) + gx));
        }
#ifdef F3_ABL_LATENCY2X
        {   // experiment: a second, data-dependent round trip (same bytes) -- how exposed is the tile-load latency?
            const u32 z = (u32)qcap >> 30;                      // runtime zero
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int r = (tid >> 5) + p * RP;
                if (ldOk && r < H) v[p] = gload128(src + (size_t)(st.y0 + r) * sp + gx + (v[p].x & z));
            }
        }
#endif
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int r = (tid >> 5) + p * RP;
            if (colOk && r < H) {
                *(uint4*)(img + r * Pb + ck * 16) = v[p];
                *(uint4*)(sc + r * Pb + ck * 16) = make_uint4(0, 0, 0, 0);
            }
        }
        for (int r = (tid >> 5) + 6 * RP; r < H; r += RP) {               // taller cells (tiny images only)
            if (colOk) {
                uint4 t = make_uint4(0, 0, 0, 0);
                if (gx < rowLimit) t = gload128u(src, (u32)(__mul24(st.y0 + r, sp) + gx));
                *(uint4*)(img + r * Pb + ck * 16) = t;
                *(uint4*)(sc + r * Pb + ck * 16) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int c = wv; c < st.ncell; c += F3_NT / 64) {
        const CellInfo cell = c == wv ? cell0 : cells[st.cell0 + c];
        const int cx0 = cell.x0 + 3 - st.xal, cx1 = cell.x0 + cell.cw - 3 - st.xal;   // valid columns (tile coords)
        const int vy0 = 3, vy1 = H - 3;
        int n3 = 0;
#ifdef F3_ABL_NOAPPEND
        u32 ablAcc = 0;
#endif
        bool ovfl = false;                                                // survivors of the quick reject exceed the queue
        if (cx1 > cx0 && vy1 > vy0) {
            // lane -> (row offset lr, 8-px column lc): ncol columns cover [cx0, cx1), rpi rows per wave-iteration
            const int c80 = cx0 >> 3, ncol = ((cx1 + 7) >> 3) - c80, nrow = vy1 - vy0;
            const int rpi = (int)(64.0f / (float)ncol);                  // floor(64 / ncol), ncol in 1..64 (exact: IEEE divide)
            // lane / ncol by a wave-uniform reciprocal: magic = ceil(2^16 / ncol) is exact for lane < 64, ncol <= 64
            const int magic = (int)ceilf(65536.0f / (float)ncol);        // == ceil(65536 / ncol) for ncol <= 64 (checked exhaustively)
            const int lr = (int)(((u32)lane * (u32)magic) >> 16);
            const int lc = lane - lr * ncol;
            const bool lact = lr < rpi;
            const int tx8 = (c80 + lc) << 3;
            // survivor flags live in a sparse layout: pixel k of the lane's 8 at bit 4k+3 (that is where two v_perm drop
            // the sign bits; popcount and ctz do not care about the spacing, and ascending bits are ascending x).
            // colmask = the lane's pixels inside [cx0, cx1): nibbles [lo, hi) of 0x88888888 (empty for idle lanes)
            const int lo = min(max(cx0 - tx8, 0), 8), hi = lact ? min(max(cx1 - tx8, 0), 8) : 0;
            const u32 below_hi = hi >= 8 ? 0xFFFFFFFFu : (1u << (4 * hi)) - 1u;
            const u32 below_lo = lo >= 8 ? 0xFFFFFFFFu : (1u << (4 * lo)) - 1u;
            const u32 colmask = 0x88888888u & below_hi & ~below_lo;
            for (int pass = 0; pass < 2; ++pass) {
                const int t = pass == 0 ? g.iniTh : g.minTh;
                if (pass == 1 && g.minTh >= g.iniTh) break;               // a higher retry threshold cannot add corners
                const us2 t2 = as_us2((u32)t * 0x00010001u);
                // ---- quick reject + compaction: 8 px per lane, lane -> fixed 8-px column (no per-iteration index math)
                int n1 = 0;
#ifdef F3_ABL_NOQUICK
                for (int r0 = 0; r0 < 0; r0 += rpi) {
#else
                for (int r0 = 0; r0 < nrow; r0 += rpi) {
#endif
                    // no divergent branch around the test: lanes past the last row redo row nrow-1 and are masked out
                    const bool rowok = r0 + lr < nrow;
                    const int r = min(r0 + lr, nrow - 1);
                    u32 m = 0;
                    const int ty = vy0 + r;
                    {
                        const u8* rowc = img + ty * Pb + tx8;
                        const uint2 Bq = *(const uint2*)rowc;
                        const u32 A = *(const u32*)(rowc - 4), Cw = *(const u32*)(rowc + 8);
                        const uint2 Uq = *(const uint2*)(rowc - 3 * Pb), Dq = *(const uint2*)(rowc + 3 * Pb);
                        // even / odd bytes of every word as 16-bit pairs: e* = pixels (0,2) of the word, o* = pixels (1,3)
                        const u32 SE = 0x0c020c00u, SO = 0x0c030c01u;
#define F3_E(w) as_us2(__builtin_amdgcn_perm(0, (w), SE))
#define F3_O(w) as_us2(__builtin_amdgcn_perm(0, (w), SO))
#define F3_AL(hi, lo) as_us2(__builtin_amdgcn_alignbyte(as_u32(hi), as_u32(lo), 2))   /* (lo.hi16, hi.lo16) */
                        const us2 eA = F3_E(A), oA = F3_O(A), eB0 = F3_E(Bq.x), oB0 = F3_O(Bq.x), eB1 = F3_E(Bq.y),
                                  oB1 = F3_O(Bq.y), eC = F3_E(Cw), oC = F3_O(Cw);
                        // pixel groups of the lane's 8: G0 = (0,2) G1 = (1,3) G2 = (4,6) G3 = (5,7); left = x-3, right = x+3
                        const us2 vv[4] = {eB0, oB0, eB1, oB1};
                        const us2 uu[4] = {F3_E(Uq.x), F3_O(Uq.x), F3_E(Uq.y), F3_O(Uq.y)};
                        const us2 dd[4] = {F3_E(Dq.x), F3_O(Dq.x), F3_E(Dq.y), F3_O(Dq.y)};
                        const us2 ll[4] = {oA, F3_AL(eB0, eA), oB0, F3_AL(eB1, eB0)};
                        const us2 rr4[4] = {F3_AL(oB1, oB0), eB1, F3_AL(oC, oB1), eC};
#undef F3_E
#undef F3_O
#undef F3_AL
                        u32 sg[4];
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const us2 X = pkmax(pkmin(uu[gq], dd[gq]), pkmin(ll[gq], rr4[gq]));
                            const us2 Y = pkmin(pkmax(uu[gq], dd[gq]), pkmax(ll[gq], rr4[gq]));
                            // survivor <=> v - X > t or Y - v > t <=> t - max(v - X, Y - v) < 0 (signed halves, |.| <= 255): sign bit of a half
                            sg[gq] = as_u32(__builtin_bit_cast(us2, __builtin_bit_cast(ss2, t2) - __builtin_elementwise_max(__builtin_bit_cast(ss2, vv[gq] - X), __builtin_bit_cast(ss2, Y - vv[gq]))));
                        }
                        const u32 Me = __builtin_amdgcn_perm(sg[2], sg[0], 0x07050301u);   // high bytes of px 0,2,4,6
                        const u32 Mo = __builtin_amdgcn_perm(sg[3], sg[1], 0x07050301u);   // px 1,3,5,7
                        m = (((Me >> 4) & 0x08080808u) | (Mo & 0x80808080u)) & (rowok ? colmask : 0u);
                    }
#ifdef F3_ABL_NOAPPEND
                    ablAcc |= m; continue;
#endif
                    // wave-inclusive prefix of popcount(m) (0..8) by a DPP scan, then each lane appends its own survivors
                    const int c = __popc(m);
                    int sc_;
#if F3_ASM_SCAN
                    // seven fused DPP adds (the builtin form compiles to a v_mov_dpp + v_mov 0 + v_add per step: 17 VALU); the s_nops
                    // are the two wait states a DPP read needs behind the VALU write of the same register
                    asm volatile("s_nop 1\n\t"
                                 "v_add_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                                 "s_nop 0"
                                 : "=&v"(sc_) : "v"(c));
#else
                    sc_ = c;
                    sc_ += __builtin_amdgcn_update_dpp(0, c, 0x111, 0xf, 0xf, true);        // row_shr:1
                    sc_ += __builtin_amdgcn_update_dpp(0, c, 0x112, 0xf, 0xf, true);        // row_shr:2
                    sc_ += __builtin_amdgcn_update_dpp(0, c, 0x113, 0xf, 0xf, true);        // row_shr:3
                    sc_ += __builtin_amdgcn_update_dpp(0, sc_, 0x114, 0xf, 0xe, true);      // row_shr:4
                    sc_ += __builtin_amdgcn_update_dpp(0, sc_, 0x118, 0xf, 0xc, true);      // row_shr:8
                    sc_ += __builtin_amdgcn_update_dpp(0, sc_, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1,3
                    sc_ += __builtin_amdgcn_update_dpp(0, sc_, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2,3
#endif
                    const int tot = __builtin_amdgcn_readlane(sc_, 63);
                    if (n1 + tot > qcap) { ovfl = true; break; }             // wave-uniform; the cell goes to k_fast_fix
                    int pos = n1 + sc_ - c;
                    u32 mm = m;
                    while (mm) {
                        const int bp = __builtin_ctz(mm);
                        q[pos++] = (u16)((ty << 9) | (tx8 + (bp >> 2)));
                        mm &= mm - 1;
                    }
                    n1 += tot;
                }
                if (ovfl) break;
#ifdef F3_ABL_NOSCORE
                n1 = 0;
#endif
                // ---- exact score, two survivors per lane (entries 2*lane and 2*lane+1 keep the queue order); keep S >= t in place
                int n2 = 0;
                for (int e0 = 0; e0 < n1; e0 += 128) {
                    const int eA = e0 + 2 * lane, eB = eA + 1;
                    const u32 both = *(const u32*)(q + min(eA, (n1 - 1) & ~1));      // entries eA, eB in one aligned 32-bit read
                    const int pqA = (int)(both & 0xFFFFu), pqB = (int)(both >> 16);
                    const int pxA = (pqA >> 9) * Pb + (pqA & 511);
                    const int pxB = eB < n1 ? (pqB >> 9) * Pb + (pqB & 511) : pxA;    // odd tail: score A twice, B is masked out
                    const ss2 sv = fast_score16x2(img + pxA, img + pxB, Pb);
                    const bool fA = eA < n1 && sv.x >= t, fB = eB < n1 && sv.y >= t;
                    if (fA) sc[pxA] = (u8)sv.x;
                    if (fB) sc[pxB] = (u8)sv.y;
                    const unsigned long long balA = __ballot(fA), balB = __ballot(fB);
                    const int posA = n2 + __popcll(balA & lt) + __popcll(balB & lt);
                    if (fA) q[posA] = (u16)pqA;
                    if (fB) q[posA + (fA ? 1 : 0)] = (u16)pqB;
                    n2 += __popcll(balA) + __popcll(balB);
                }
                // ---- strict 3x3 maxima inside the cell window (in place)
                n3 = 0;
                for (int e0 = 0; e0 < n2; e0 += 64) {
                    const int e = e0 + lane;
                    bool keep = false;
                    int pq = 0;
                    if (e < n2) {
                        pq = q[e];
                        const int tx = pq & 511;
                        const u8* p = sc + (pq >> 9) * Pb + tx;
                        // branch-free: all eight neighbours are read (the tile has a halo; a neighbour column outside the cell's
                        // window may hold another cell's score, possibly mid-write -- it is masked to 0, never used)
                        const int s = p[0];
                        const int ml = max(max((int)p[-1], (int)p[-Pb - 1]), (int)p[Pb - 1]);
                        const int mr = max(max((int)p[1], (int)p[-Pb + 1]), (int)p[Pb + 1]);
                        const int mv = max((int)p[-Pb], (int)p[Pb]);
                        keep = s > max(mv, max(tx > cx0 ? ml : 0, tx < cx1 - 1 ? mr : 0));
                    }
                    const unsigned long long bal = __ballot(keep);
                    if (keep) q[n3 + __popcll(bal & lt)] = (u16)pq;
                    n3 += __popcll(bal);
                }
                if (n3 > 0) break;
            }
        }
        if (ovfl) {
            if (lane == 0) ovfList[atomicAdd(&ovf[0], 1u)] = (u32)frame * (u32)g.totalCells + (u32)(st.cell0 + c);
            continue;
        }
        // ---- packed store.  The queue is row-major by construction: the quick reject appends iteration by iteration
        // (ascending rows), lanes in (row, column) order through the wave prefix, pixels of a lane in ascending x; the two
        // in-place compactions are stable.  cv::FAST emits in the same order, so entry e is candidate e of the cell.
        u32* out = candEnt + (size_t)frame * g.totalSlots + cell.slot;
        for (int e0 = 0; e0 < n3; e0 += 64) {
            const int e = e0 + lane;
            if (e < n3) {
                const int my = q[e];
                const int tx = my & 511, ty = my >> 9;
                const int s = sc[ty * Pb + tx];
                if (e < L.slotCap)
                    out[e] = (u32)(st.xal + tx - 16) | ((u32)(st.y0 + ty - 16) << 12) | ((u32)s << 24);
                else atomicExch(err, 1);
            }
        }
#ifdef F3_ABL_NOAPPEND
        if (ablAcc == 0x12345u) sc[0] = 1;
#endif
        if (lane == 0) candCnt[(size_t)frame * g.totalCells + cell.cnt] = (u32)n3;
    }
}

#endif  /* ORBX_AB */

// ------------------------------------------------------------------------------------------------
// k_fast4: k_fast3 on an instruction diet (round 3).  Same algorithm, same order, same outputs; what changed:
//   * the quick reject walks a cell as ONE linear list of (row, 8-px column) items, 64 per wave-iteration, with the columns
//     starting at a 4-px boundary: a 36 x 38-px window is 5 columns x 38 rows = 190 items = 3 iterations at 99 % lane use
//     (k_fast3: lane -> fixed 8-aligned column, 6 columns x 10 rows per iteration: 4 iterations, 66 % of the lanes on valid
//     pixels);
//   * everything that depends only on the cell's SHAPE -- which tile bytes a lane reads in iteration i, which of its 8 pixels
//     lie inside the window, the (row, column) part of its queue entries -- comes from a host-built table (F4Item, one 8-byte
//     load per lane and iteration, shared by all cells of that shape and therefore L1-resident); what depends on the cell
//     comes through the scalar cache (CellAux).  The per-cell set-up of k_fast3 (two IEEE divides, magic reciprocals,
//     column masks: ~80 VALU) is gone;
//   * LDS addresses are 32-bit offsets from window corners, so that every ring / neighbour read is base + immediate (k_fast3:
//     18 address adds per scored pixel pair, a 64-bit multiply-add per quick iteration);
//   * the tile load reads clamped addresses instead of predicating (columns right of the image only reach masked pixels).
// Queue entry (u16): row << 9 | xs << 2 | 3, xs = column relative to the cell's first 4-aligned column (so xs = 8 * col + k and
// the low five bits are the sparse mask's bit index 4k + 3: the append loop ORs the bit position in, no shift).
// ------------------------------------------------------------------------------------------------
struct CellAux {                                            // 32 bytes, read with one scalar load
    u32 tab;                                                // first F4Item of the cell's shape table: [nit][64]
    int slot, cnt;                                          // as CellInfo
    u16 base;                                               // tile byte offset of (row 0, xs 0): 3 * pitch + s4
    u16 nit;                                                // quick-pass iterations (0: empty window)
    short outx, outy;                                       // packed output coordinates of (row 0, xs 0)
    u16 xlo, xhi;                                           // xs of the window's first / last column
    u32 pad[2];
};
struct F4Item { u32 mask; u32 offq; };                      // mask: pixel k of the lane's 8 inside the window -> bit 4k+3; offq: lo16 = row * pitch + 8 * col, hi16 = row << 9 | col << 5

// LDS accesses of k_fast4 by ABSOLUTE 32-bit LDS address (base of the dynamic allocation folded into wave-uniform offsets once):
// `shared_array + offset` makes the compiler add the array's (zero) address per access and widen pointer arithmetic to 64 bits.
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef __attribute__((address_space(3))) unsigned short lds_u16_t;
typedef __attribute__((address_space(3))) u32 lds_u32_t;
typedef u32 f4_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4_v4u lds_u128_t;
__device__ __forceinline__ u32 lds_r32(u32 a) { return *(const lds_u32_t*)(uintptr_t)a; }
__device__ __forceinline__ u32 lds_r16(u32 a) { return (u32)*(const lds_u16_t*)(uintptr_t)a; }
__device__ __forceinline__ u32 lds_r8(u32 a) { return (u32)*(const lds_u8_t*)(uintptr_t)a; }
__device__ __forceinline__ void lds_w8(u32 a, u32 v) { *(lds_u8_t*)(uintptr_t)a = (unsigned char)v; }
__device__ __forceinline__ void lds_w16(u32 a, u32 v) { *(lds_u16_t*)(uintptr_t)a = (unsigned short)v; }
__device__ __forceinline__ void lds_w128(u32 a, uint4 v) { *(lds_u128_t*)(uintptr_t)a = f4_v4u{v.x, v.y, v.z, v.w}; }

// exact score of two pixels; oa / ob = LDS addresses of the top-left corners (x-3, y-3) of their 7x7 windows
template <int P>
__device__ __forceinline__ ss2 fast_score16x2_tl(u32 oa, u32 ob, int prt) {
    const int p = P > 0 ? P : prt;
#define F2(row, o) us2{(u16)lds_r8(oa + (row) * p + (o)), (u16)lds_r8(ob + (row) * p + (o))}
    us2 r[16];
    r[0] = F2(6, 3);   r[1] = F2(6, 4);   r[2] = F2(5, 5);   r[3] = F2(4, 6);
    r[4] = F2(3, 6);   r[5] = F2(2, 6);   r[6] = F2(1, 5);   r[7] = F2(0, 4);
    r[8] = F2(0, 3);   r[9] = F2(0, 2);   r[10] = F2(1, 1);  r[11] = F2(2, 0);
    r[12] = F2(3, 0);  r[13] = F2(4, 0);  r[14] = F2(5, 1);  r[15] = F2(6, 2);
    const us2 v = F2(3, 3);
#undef F2
    us2 m2[8], M2[8], m4[8], M4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m2[j] = pkmin(r[2 * j], r[2 * j + 1]); M2[j] = pkmax(r[2 * j], r[2 * j + 1]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { m4[j] = pkmin(m2[j], m2[(j + 1) & 7]); M4[j] = pkmax(M2[j], M2[(j + 1) & 7]); }
    // three-input packed minimum / maximum (gfx950: v_pk_minimum3_f16 / v_pk_maximum3_f16).  The halves hold 0..255: as binary16 bit
    // patterns those are non-negative subnormals, which order exactly like the integers (the kernels run with f16 denormals preserved,
    // amdhsa_float_denorm_mode_16_64 = 3; checked against the two-input integer network on the GPU by the parity tests and the fuzz)
    us2 u[8], U[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const us2 a = r[(2 * j + 15) & 15], c = r[(2 * j + 8) & 15];
        u[j] = pkmin3(m4[j], m4[(j + 2) & 7], pkmax(a, c));
        U[j] = pkmax3(M4[j], M4[(j + 2) & 7], pkmin(a, c));
    }
    const us2 bmax = pkmax3(pkmax3(u[0], u[1], u[2]), pkmax3(u[3], u[4], u[5]), pkmax(u[6], u[7]));
    const us2 amin = pkmin3(pkmin3(U[0], U[1], U[2]), pkmin3(U[3], U[4], U[5]), pkmin(U[6], U[7]));
    const ss2 d1 = __builtin_bit_cast(ss2, v - amin), d2 = __builtin_bit_cast(ss2, bmax - v);   // |.| <= 255: exact as signed 16-bit
    return __builtin_elementwise_max(d1, d2) - ss2{1, 1};
}

// number of set bits of a wave mask below this lane (v_mbcnt_lo + v_mbcnt_hi on the scalar mask: no per-lane "lanes below" constant)
__device__ __forceinline__ int f4_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// One FAST cell by one wave, in three phases.
struct F4Ctx {                                              // wave-uniform context of a cell
    u32 imgA, scD, qA, cbase;                               // LDS addresses: image tile, score tile distance, the wave's queue, (row 0, xs 0)
    int Pb, qcap, lane;
};

// quick reject at threshold t: survivors appended to the wave's queue in row-major order; returns their number, -1 if the queue overflowed
template <int PITCH>
__device__ __forceinline__ int f4_quick(const F4Ctx& cx, const CellAux& ax, const F4Item* __restrict__ items, const int t) {
    const int Pb = PITCH > 0 ? PITCH : cx.Pb, qcap = cx.qcap;
    const u32 qA = cx.qA;
    const u32 laneOff = (u32)cx.lane * 8u;
    const int nit = ax.nit;
    const F4Item* tab = items + ax.tab;
    // window corner of the lane's reads: 3 rows up, 4 bytes left of its 8 pixels (all offsets below are >= 0)
    const u32 qbase = cx.cbase - (u32)(3 * Pb + 4);
    const us2 t2 = as_us2((u32)t * 0x00010001u);
    int n1 = 0;
    // Item loads are inline asm.  A compiler-visible load of the NEXT item gets sunk below the overflow test (it then
    // completes right behind the short append loop, exposed), and one of the first item makes the compiler wait for
    // vmcnt(0) inside the loop (the counter is in order), i.e. for the prefetch.  The asm load writes nx / ny while the
    // iteration runs; the statement at the bottom waits and only then reads them (its inputs ARE the load's registers:
    // no tied operand, so no copy can be placed in front of the wait -- checked in the ISA).
    uint2 cur;                                                 // .x = mask, .y = offq of the current item
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(cur) : "v"(laneOff), "s"(tab) : "memory");
    for (int it = 0; it < nit; ++it) {
        uint2 nxt;
        {
            const F4Item* np = tab + (size_t)min(it + 1, nit - 1) * 64;
            asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=&v"(nxt) : "v"(laneOff), "s"(np) : "memory");
        }
        u32 m;
        {
            const u32 o = qbase + (cur.y & 0xFFFFu);
            const u32 U0 = lds_r32(o + 4), U1 = lds_r32(o + 8);
            const u32 A = lds_r32(o + 3 * Pb), B0 = lds_r32(o + 3 * Pb + 4), B1 = lds_r32(o + 3 * Pb + 8), Cw = lds_r32(o + 3 * Pb + 12);
            const u32 D0 = lds_r32(o + 6 * Pb + 4), D1 = lds_r32(o + 6 * Pb + 8);
            // even / odd bytes of every word as 16-bit pairs: e* = pixels (0,2) of the word, o* = pixels (1,3)
            const u32 SE = 0x0c020c00u, SO = 0x0c030c01u;
#define F3_E(w) as_us2(__builtin_amdgcn_perm(0, (w), SE))
#define F3_O(w) as_us2(__builtin_amdgcn_perm(0, (w), SO))
#define F3_AL(hi, lo) as_us2(__builtin_amdgcn_alignbyte(as_u32(hi), as_u32(lo), 2))   /* (lo.hi16, hi.lo16) */
            // pixel i of the lane sits at byte i of (B0, B1); left = x-3 starts at byte 1 of A, right = x+3 at byte 3 of B0.  Pairs that
            // straddle two words come from ONE v_perm of both (selector bytes 0-3 = low word, 4-7 = high word)
#define F3_P2(hi, lo, sel) as_us2(__builtin_amdgcn_perm((hi), (lo), (sel)))
            const us2 oA = F3_O(A), eB0 = F3_E(B0), oB0 = F3_O(B0), eB1 = F3_E(B1), oB1 = F3_O(B1), eC = F3_E(Cw);
            // pixel groups of the lane's 8: G0 = (0,2) G1 = (1,3) G2 = (4,6) G3 = (5,7)
            const us2 vv[4] = {eB0, oB0, eB1, oB1};
            const us2 uu[4] = {F3_E(U0), F3_O(U0), F3_E(U1), F3_O(U1)};
            const us2 dd[4] = {F3_E(D0), F3_O(D0), F3_E(D1), F3_O(D1)};
            const us2 ll[4] = {oA, F3_P2(B0, A, 0x0c040c02u), oB0, F3_P2(B1, B0, 0x0c040c02u)};          // (A.1,A.3) (A.2,B0.0) (B0.1,B0.3) (B0.2,B1.0)
            const us2 rr4[4] = {F3_P2(B1, B0, 0x0c050c03u), eB1, F3_P2(Cw, B1, 0x0c050c03u), eC};        // (B0.3,B1.1) (B1.0,B1.2) (B1.3,C.1) (C.0,C.2)
#undef F3_P2
#undef F3_E
#undef F3_O
#undef F3_AL
            u32 sg[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const us2 X = pkmax(pkmin(uu[gq], dd[gq]), pkmin(ll[gq], rr4[gq]));
                const us2 Y = pkmin(pkmax(uu[gq], dd[gq]), pkmax(ll[gq], rr4[gq]));
                // survivor <=> v - X > t or Y - v > t <=> t - max(v - X, Y - v) < 0 (signed halves, |.| <= 255): sign bit of a half
                sg[gq] = as_u32(__builtin_bit_cast(us2, __builtin_bit_cast(ss2, t2) - __builtin_elementwise_max(__builtin_bit_cast(ss2, vv[gq] - X), __builtin_bit_cast(ss2, Y - vv[gq]))));
            }
            const u32 Me = __builtin_amdgcn_perm(sg[2], sg[0], 0x07050301u);   // high bytes of px 0,2,4,6
            const u32 Mo = __builtin_amdgcn_perm(sg[3], sg[1], 0x07050301u);   // px 1,3,5,7
            // even pixels' sign bits to bit 3 of each byte, odd pixels' stay at bit 7 (v_lshrrev, v_bfi, v_and: the item mask has bits 3 and 7
            // of a byte only and clears the rest)
            m = (((Me >> 4) & 0x08080808u) | (Mo & ~0x08080808u)) & cur.x;
        }
        // wave-inclusive prefix of popcount(m) (0..8) by a DPP scan, then each lane appends its own survivors
        const int cn = __popc(m);
        int sc_;
        asm volatile("s_nop 1\n\t"
                     "v_add_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 0"
                     : "=&v"(sc_) : "v"(cn));
#ifdef F4_ABL_NOAPPEND
        const int tot = (m == 0x12345u) ? 1 : 0;
#else
        const int tot = __builtin_amdgcn_readlane(sc_, 63);
#endif
        if (n1 + tot > qcap) {                                   // wave-uniform; the cell goes to k_fast_fix
            // the prefetched item must LAND before its registers die: the compiler is free to reuse them at once, and a load that
            // arrives later would overwrite whatever they hold by then
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt.x), "+v"(nxt.y) : : "memory");
            return -1;
        }
        u32 qa = qA + (u32)(n1 + sc_ - cn) * 2u;
        const u32 eb = cur.y >> 16;
        u32 mm = m;
        while (mm) {
            lds_w16(qa, eb | (u32)__builtin_ctz(mm));
            qa += 2;
            mm &= mm - 1;
        }
        n1 += tot;
        asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(cur.x), "=&v"(cur.y) : "v"(nxt.x), "v"(nxt.y) : "memory");
    }
    return n1;
}

// exact score of the n1 queued survivors at threshold t, then strict 3x3 maxima inside the cell window; both compact the queue in place.
// Returns the number of keypoints left in the queue.
template <int PITCH>
__device__ __forceinline__ int f4_score_nms(const F4Ctx& cx, const CellAux& ax, const int t, const int n1) {
    const int Pb = PITCH > 0 ? PITCH : cx.Pb, lane = cx.lane;
    const u32 qA = cx.qA, scD = cx.scD, cbase = cx.cbase;
#if defined(F4_ABL_NOSCORE) || defined(F4_ABL_NOAPPEND)
    if (n1 < 60000) return 0;
#endif
    // ---- exact score, two survivors per lane (entries 2*lane and 2*lane+1 keep the queue order); keep S >= t in place
    int n2 = 0;
    const u32 wbase = cbase - (u32)(3 * Pb + 3);                 // (row 0, xs 0) -> top-left corner of its 7x7 window
    for (int e0 = 0; e0 < n1; e0 += 128) {
        // lane l scores entries e0 + l and e0 + 64 + l: the 32 lanes of one LDS access group then read 32 CONSECUTIVE survivors (about
        // five tile rows) instead of every other one of 64 (ten rows) -- fewer rows, fewer bank conflicts among the scattered ring bytes
        const int eA = e0 + lane, eB = eA + 64;
        const u32 ea = lds_r16(qA + 2u * (u32)min(eA, n1 - 1)), eb = lds_r16(qA + 2u * (u32)min(eB, n1 - 1));
        const u32 xa = __builtin_amdgcn_ubfe(ea, 2, 7), ya = ea >> 9;
        const u32 xb = __builtin_amdgcn_ubfe(eb, 2, 7), yb = eb >> 9;
        const u32 oa = ya * (u32)Pb + xa + wbase;
        const u32 ob = yb * (u32)Pb + xb + wbase;                   // (entries past the end re-score the last one and are masked out)
        const ss2 sv = fast_score16x2_tl<PITCH>(oa, ob, Pb);
        const bool fA = eA < n1 && sv.x >= t, fB = eB < n1 && sv.y >= t;
        if (fA) lds_w8(oa + scD + 3 * Pb + 3, (u32)sv.x);
        if (fB) lds_w8(ob + scD + 3 * Pb + 3, (u32)sv.y);
        const unsigned long long balA = __ballot(fA), balB = __ballot(fB);
        const int nA = __popcll(balA);
        if (fA) lds_w16(qA + 2u * (u32)(n2 + f4_below(balA)), ea);            // stable: the A half (entries e0 .. e0+63) before the B half
        if (fB) lds_w16(qA + 2u * (u32)(n2 + nA + f4_below(balB)), eb);
        n2 += nA + __popcll(balB);
    }
#ifdef F4_ABL_NONMS
    if (n2 < 60000) return 0;
#endif
    // ---- strict 3x3 maxima inside the cell window (in place)
    int n3 = 0;
    const u32 sbase = cbase + scD - (u32)(Pb + 1);               // (row 0, xs 0) -> its upper-left neighbour in the score tile
    for (int e0 = 0; e0 < n2; e0 += 64) {
        const int e = e0 + lane;
        bool keep = false;
        u32 pq = 0;
        if (e < n2) {
            pq = lds_r16(qA + 2u * (u32)e);
            const u32 xs = __builtin_amdgcn_ubfe(pq, 2, 7);
            const u32 o = (pq >> 9) * (u32)Pb + xs + sbase;
            // branch-free: all eight neighbours are read (the tile has a halo; a neighbour column outside the cell's
            // window may hold another cell's score, possibly mid-write -- it is masked to 0, never used)
            const int s = (int)lds_r8(o + Pb + 1);
            const int ml = max(max((int)lds_r8(o + Pb), (int)lds_r8(o)), (int)lds_r8(o + 2 * Pb));
            const int mr = max(max((int)lds_r8(o + Pb + 2), (int)lds_r8(o + 2)), (int)lds_r8(o + 2 * Pb + 2));
            const int mv = max((int)lds_r8(o + 1), (int)lds_r8(o + 2 * Pb + 1));
            keep = s > max(mv, max(xs > (u32)ax.xlo ? ml : 0, xs < (u32)ax.xhi ? mr : 0));
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) lds_w16(qA + 2u * (u32)(n3 + f4_below(bal)), pq);
        n3 += __popcll(bal);
    }
    return n3;
}

// packed store of the n3 keypoints in the queue + the cell's count
template <int PITCH>
__device__ __forceinline__ void f4_store(const Geom& g, const F4Ctx& cx, const CellAux& ax, const int n3, const int slotCap, const int frame,
                                         u32* candCnt, u32* candEnt, int* err) {
    const int Pb = PITCH > 0 ? PITCH : cx.Pb, lane = cx.lane;
    const u32 qA = cx.qA, scD = cx.scD, cbase = cx.cbase;
    // ---- packed store.  The queue is row-major by construction: the quick reject appends iteration by iteration
    // (items in (row, column) order), lanes in item order through the wave prefix, pixels of a lane in ascending x; the two
    // in-place compactions are stable.  cv::FAST emits in the same order, so entry e is candidate e of the cell.
    u32* out = candEnt + (size_t)frame * g.totalSlots + ax.slot;
    for (int e0 = 0; e0 < n3; e0 += 64) {
        const int e = e0 + lane;
        if (e < n3) {
            const u32 my = lds_r16(qA + 2u * (u32)e);
            const u32 xs = __builtin_amdgcn_ubfe(my, 2, 7), row = my >> 9;
            const u32 s = lds_r8(cbase + scD + row * (u32)Pb + xs);
            if (e < slotCap)
                out[e] = (u32)((int)ax.outx + (int)xs) | ((u32)((int)ax.outy + (int)row) << 12) | (s << 24);
            else atomicExch(err, 1);
        }
    }
    if (lane == 0) candCnt[(size_t)frame * g.totalCells + ax.cnt] = (u32)n3;
}

// cv::FAST at iniTh and, only if that leaves the cell empty after non-max suppression, at minTh (ORBextractor.cc:1112-1125);
// `n1` = result of the first quick pass (already run by the caller)
template <int PITCH>
__device__ __forceinline__ void f4_finish(const Geom& g, const F4Ctx& cx, const CellAux& ax, const F4Item* __restrict__ items, int n1, const int slotCap,
                                          const int frame, const int cellIdx, u32* candCnt, u32* candEnt, int* err, u32* ovf, u32* ovfList) {
    int n3 = 0;
    if (n1 > 0) n3 = f4_score_nms<PITCH>(cx, ax, g.iniTh, n1);
#if !defined(F4_ABL_NOSCORE) && !defined(F4_ABL_NOAPPEND) && !defined(F4_ABL_NONMS)
    if (n1 >= 0 && n3 == 0 && ax.nit > 0 && g.minTh < g.iniTh) {       // (a higher retry threshold cannot add corners)
        n1 = f4_quick<PITCH>(cx, ax, items, g.minTh);
        if (n1 > 0) n3 = f4_score_nms<PITCH>(cx, ax, g.minTh, n1);
    }
#endif
    if (n1 < 0) {
        if (cx.lane == 0) ovfList[atomicAdd(&ovf[0], 1u)] = (u32)frame * (u32)g.totalCells + (u32)cellIdx;
        return;
    }
    f4_store<PITCH>(g, cx, ax, n3, slotCap, frame, candCnt, candEnt, err);
}

template <int PITCH>
__global__ __launch_bounds__(F3_NT) void k_fast4(Geom g, const u8* const* l0, int l0pitch, const u8* pyr,
                                                 const CellAux* __restrict__ aux, const F4Item* __restrict__ items,
                                                 const StripInfo* __restrict__ strips,
                                                 u32* candCnt, u32* candEnt, int* err, int tileBytes, int qcap,
                                                 u32* ovf, u32* ovfList) {
    extern __shared__ __attribute__((aligned(16))) unsigned char f3smem[];
    const u32 imgA = (u32)(uintptr_t)(lds_u8_t*)f3smem;                  // LDS address of the image tile (wave-uniform)
    const u32 scD = (u32)tileBytes;                                       // score tile behind the image tile, same indexing
    StripInfo st = strips[gridDim.x - 1 - (F3_XCD ? xcd_task(blockIdx.x, gridDim.x) : blockIdx.x)];   // coarser (denser, slower) strips of the group first: a lighter tail
    st.level = (short)__builtin_amdgcn_readfirstlane(st.level);
    const int frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave-uniform: per-cell metadata comes through the scalar cache
    const u32 qA = imgA + 2u * (u32)tileBytes + (u32)(wv * qcap) * 2u;    // this wave's queue (u16 entries)
    const int Pb = PITCH > 0 ? PITCH : (int)st.lp, H = st.h;              // tile pitch in bytes
    int sp;
    const u8* src = level_ptr(g, l0, l0pitch, pyr, frame, st.level, &sp);
    const LevelDesc& L = g.lv[st.level];
    {   // tile load: 32 lanes x 16 B per row, 8 rows per pass; six passes in flight (one memory round trip).  Rows and columns are
        // CLAMPED into the level instead of predicated: a column at or right of the image's 16-byte-rounded width is only ever
        // read by lanes whose pixels are masked out (valid pixels end 13 columns left of the image edge), a row >= H is not stored.
        const int cpr = Pb >> 4;
        const int rowLimit = min((L.w + 15) & ~15, sp);
        const int ck = tid & 31;
        const int gx = min(st.xal + ck * 16, rowLimit - 16);
        const bool colOk = ck < cpr;
        const int RP = F3_NT / 32;
        const int r0 = tid >> 5;
        uint4 v[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int r = min(r0 + p * RP, H - 1);
#ifdef F4_ABL_NOLOAD
            v[p] = make_uint4(r, gx, sp, 0);
#else
            v[p] = gload128u(src, mad24((u32)(st.y0 + r), (u32)sp, (u32)gx));   // (written as __mul24 + add the compiler picks the 64-bit multiply-add)
#endif
        }
        const u32 lo = imgA + (u32)(r0 * Pb + ck * 16);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int r = r0 + p * RP;
            if (colOk && r < H) {
                lds_w128(lo + (u32)(p * RP) * (u32)Pb, v[p]);
                lds_w128(lo + scD + (u32)(p * RP) * (u32)Pb, make_uint4(0, 0, 0, 0));
            }
        }
        for (int r = r0 + 6 * RP; r < H; r += RP) {                       // taller cells (coarse levels, tiny images)
            if (colOk) {
                const uint4 t = gload128u(src, (u32)(__mul24(st.y0 + r, sp) + gx));
                lds_w128(imgA + (u32)(r * Pb + ck * 16), t);
                lds_w128(imgA + scD + (u32)(r * Pb + ck * 16), make_uint4(0, 0, 0, 0));
            }
        }
    }
    __syncthreads();
    for (int c = wv; c < st.ncell; c += F3_NT / 64) {
        const CellAux ax = aux[st.cell0 + c];
        F4Ctx cx; cx.imgA = imgA; cx.scD = scD; cx.qA = qA; cx.cbase = imgA + ax.base; cx.Pb = Pb; cx.qcap = qcap; cx.lane = lane;
#ifdef F4_ABL_NOQUICK
        const int n1 = 0;
#else
        const int n1 = ax.nit > 0 ? f4_quick<PITCH>(cx, ax, items, g.iniTh) : 0;
#endif
        f4_finish<PITCH>(g, cx, ax, items, n1, L.slotCap, frame, st.cell0 + c, candCnt, candEnt, err, ovf, ovfList);
    }
}

// ------------------------------------------------------------------------------------------------
// k_quadtree: ORBextractor::DistributeOctTree for one (level, frame) per workgroup.
// Keypoint-side work (quadrant histograms, re-labelling, per-node arg-max) is parallel over candidates;
// the order-defining list surgery (push_front / erase / early break at N) is replayed by lane 0 on a
// linked list held in LDS, which keeps the reference's output ORDER bit-exact.
// Tie-break among equal-count nodes in the final phase: creation sequence (normative, see oracle).
// ------------------------------------------------------------------------------------------------
struct QtShared {                                         // carved from dynamic LDS, all arrays [nodeCap]
    short4* rect; u32* cnt; u32* qc; u16* child; u16* nxt; u16* prv; u32* seq; u32* best; u8* split;
    u16* freeStk; u16* cand; unsigned long long* sortKey; u16* order;
};

#define QT_NIL 0xFFFFu

__device__ __forceinline__ int qt_quadrant(short4 r, int x, int y) {
    const int hx = (r.z - r.x + 1) >> 1, hy = (r.w - r.y + 1) >> 1;      // ceil(d/2), d >= 0
    const int mx = r.x + hx, my = r.y + hy;
    return x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
}

#ifdef ORBX_AB   /* A/B reference, not in the product library */
__global__ __launch_bounds__(256) void k_quadtree(Geom g, const CellInfo* __restrict__ cells,
                                                  const u32* __restrict__ candCnt, const u32* __restrict__ candEnt,
                                                  u16* kpNode, u32* selOut, u32* selCnt, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // grid = (frames, levels): consecutive workgroup ids go round-robin over the 8 XCDs, so the fastest-varying index must
    // be the frame -- with the level there and 8 levels, every level-0 (heaviest) workgroup would land on the same XCD
    const int level = blockIdx.y, frame = blockIdx.x;
    const LevelDesc& L = g.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cap = g.nodeCap;
    QtShared S;
    {
        unsigned char* p = smem;
        S.sortKey = (unsigned long long*)p; p += (size_t)g.sortCap * 8;
        S.rect = (short4*)p; p += (size_t)cap * 8;
        S.cnt = (u32*)p; p += (size_t)cap * 4;
        S.qc = (u32*)p; p += (size_t)cap * 16;
        S.seq = (u32*)p; p += (size_t)cap * 4;
        S.best = (u32*)p; p += (size_t)cap * 4;
        S.child = (u16*)p; p += (size_t)cap * 8;
        S.nxt = (u16*)p; p += (size_t)cap * 2;
        S.prv = (u16*)p; p += (size_t)cap * 2;
        S.freeStk = (u16*)p; p += (size_t)cap * 2;
        S.cand = (u16*)p; p += (size_t)cap * 2;
        S.order = (u16*)p; p += (size_t)cap * 2;
        S.split = (u8*)p; p += (size_t)cap;
    }
    __shared__ int s_head, s_size, s_state, s_nCand, s_nFree, s_seq, s_nPend;
    const u32* cnts = candCnt + (size_t)frame * g.totalCells + L.cellBase;
    const u32* ents = candEnt + (size_t)frame * g.totalSlots;
    u16* kn = kpNode + (size_t)frame * g.totalSlots;

    // ---- roots (ORBextractor.cc:695-763)
    for (int i = tid; i < cap; i += 256) { S.cnt[i] = 0; S.split[i] = 0; }
    __syncthreads();
    for (int c = wv; c < L.nCells; c += 4) {
        const int n = (int)min(cnts[c], (u32)L.slotCap);
        const int sb = cells[L.cellBase + c].slot;
        for (int k = lane; k < n; k += 64) {
            const u32 e = ents[sb + k];
            const int root = (int)((float)(e & 0xFFF) / L.hX);
            kn[sb + k] = (u16)root;
            atomicAdd(&S.cnt[root], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int head = QT_NIL, tail = QT_NIL, size = 0;
        for (int i = 0; i < L.nIni; ++i) {
            S.rect[i] = make_short4((short)(int)(L.hX * (float)i), 0, (short)(int)(L.hX * (float)(i + 1)), (short)L.qtH);
            S.seq[i] = i;
            if (S.cnt[i] == 0) continue;                       // empty roots are erased
            S.nxt[i] = QT_NIL; S.prv[i] = (u16)tail;
            if (tail != QT_NIL) S.nxt[tail] = (u16)i; else head = i;
            tail = i; ++size;
        }
        int nf = 0;
        for (int i = cap - 1; i >= L.nIni; --i) S.freeStk[nf++] = (u16)i;
        s_head = head; s_size = size; s_nFree = nf; s_seq = L.nIni; s_state = 0; s_nCand = 0; s_nPend = 0;
    }
    __syncthreads();

    // ---- refinement passes
    for (int iter = 0;; ++iter) {
        const int state = s_state;                          // uniform: lane 0 only rewrites it after the next barrier
        if (state == 2) break;
        if (iter > 8192) { if (tid == 0) atomicExch(err, 4); break; }
        // (1) quadrant histograms of every splittable node
        for (int i = tid; i < cap * 4; i += 256) S.qc[i] = 0;
        __syncthreads();
        for (int c = wv; c < L.nCells; c += 4) {
            const int n = (int)min(cnts[c], (u32)L.slotCap);
            const int sb = cells[L.cellBase + c].slot;
            for (int k = lane; k < n; k += 64) {
                const int nd = kn[sb + k];
                if (S.cnt[nd] > 1) {
                    const u32 e = ents[sb + k];
                    atomicAdd(&S.qc[nd * 4 + qt_quadrant(S.rect[nd], e & 0xFFF, (e >> 12) & 0xFFF)], 1u);
                }
            }
        }
        __syncthreads();
        // (1b) final phase: order the candidates by (count, creation seq) ascending -- bitonic sort in LDS
        if (state == 1) {
            const int nc = s_nCand;
            int n2 = 1;
            while (n2 < nc) n2 <<= 1;
            for (int i = tid; i < n2; i += 256) {
                unsigned long long key = ~0ull;
                if (i < nc) {
                    const int id = S.cand[i];
                    key = ((unsigned long long)S.cnt[id] << 40) | ((unsigned long long)S.seq[id] << 16) | (unsigned)id;
                }
                S.sortKey[i] = key;
            }
            __syncthreads();
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < n2; i += 256) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = S.sortKey[i], b = S.sortKey[ixj];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { S.sortKey[i] = b; S.sortKey[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
        }
        // (2) list surgery by lane 0
        if (tid == 0) {
            int head = s_head, size = s_size, nf = s_nFree, seq = s_seq, npend = 0;
            const int prevSize = size;
            const int N = L.N;
            int nNew = 0, nToExpand = 0;
            bool overflow = false;
            auto split_node = [&](int it) {
                const short4 r = S.rect[it];
                const int hx = (r.z - r.x + 1) >> 1, hy = (r.w - r.y + 1) >> 1;
                const int mx = r.x + hx, my = r.y + hy;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 qn = S.qc[it * 4 + q];
                    int id = QT_NIL;
                    if (qn > 0) {
                        if (nf == 0) { overflow = true; }
                        else {
                            id = S.freeStk[--nf];
                            S.rect[id] = make_short4((short)((q & 1) ? mx : r.x), (short)((q & 2) ? my : r.y),
                                                     (short)((q & 1) ? r.z : mx), (short)((q & 2) ? r.w : my));
                            S.cnt[id] = qn; S.seq[id] = seq++; S.split[id] = 0;
                            S.prv[id] = QT_NIL; S.nxt[id] = (u16)head;          // push_front
                            if (head != QT_NIL) S.prv[head] = (u16)id;
                            head = id; ++size;
                            if (qn > 1) { ++nToExpand; S.order[nNew++] = (u16)id; }
                        }
                    }
                    S.child[it * 4 + q] = (u16)id;
                }
                const int p = S.prv[it], n = S.nxt[it];                         // erase(it)
                if (p != QT_NIL) S.nxt[p] = (u16)n; else head = n;
                if (n != QT_NIL) S.prv[n] = (u16)p;
                --size;
                S.split[it] = 1;
                S.cand[cap - 1 - npend] = (u16)it; ++npend;                     // pending free (tail of cand[])
            };
            bool finish = false;
            if (state == 0) {                                                    // ORBextractor.cc:779-895
                int it = head;
                // nodes pushed to the front are not revisited in this pass
                while (it != QT_NIL) {
                    const int nx = S.nxt[it];
                    if (S.cnt[it] > 1) split_node(it);
                    it = nx;
                }
                if (size >= N || size == prevSize) finish = true;
                else if (size + nToExpand * 3 > N) s_state = 1;
            } else {                                                             // ORBextractor.cc:912-992
                const int nc = s_nCand;
                for (int j = nc - 1; j >= 0; --j) {
                    split_node((int)(S.sortKey[j] & 0xFFFF));
                    if (size >= N) break;
                }
                if (size >= N || size == prevSize) finish = true;
            }
            if (overflow) { atomicExch(err, 2); finish = true; }
            // candidates of the next round = the children created now, in creation order
            for (int i = 0; i < nNew; ++i) S.cand[i] = S.order[i];
            s_nCand = nNew; s_nPend = npend;
            s_head = head; s_size = size; s_nFree = nf; s_seq = seq;
            if (finish) s_state = 2;
        }
        __syncthreads();
        // (3) move the keypoints of split nodes to their children
        for (int c = wv; c < L.nCells; c += 4) {
            const int n = (int)min(cnts[c], (u32)L.slotCap);
            const int sb = cells[L.cellBase + c].slot;
            for (int k = lane; k < n; k += 64) {
                const int nd = kn[sb + k];
                if (S.split[nd]) {
                    const u32 e = ents[sb + k];
                    kn[sb + k] = S.child[nd * 4 + qt_quadrant(S.rect[nd], e & 0xFFF, (e >> 12) & 0xFFF)];
                }
            }
        }
        __syncthreads();
        // (4) recycle the erased parents
        if (tid == 0) {
            int nf = s_nFree;
            for (int i = 0; i < s_nPend; ++i) {
                const int it = S.cand[cap - 1 - i];
                S.split[it] = 0; S.cnt[it] = 0;
                S.freeStk[nf++] = (u16)it;
            }
            s_nFree = nf; s_nPend = 0;
        }
        __syncthreads();
    }
    // ---- one keypoint per node: first maximum of `response` in candidate order (ORBextractor.cc:1005-1030)
    for (int i = tid; i < cap; i += 256) S.best[i] = 0;
    __syncthreads();
    for (int c = wv; c < L.nCells; c += 4) {
        const int n = (int)min(cnts[c], (u32)L.slotCap);
        const int sb = cells[L.cellBase + c].slot;
        for (int k = lane; k < n; k += 64) {
            const u32 e = ents[sb + k];
            const u32 rel = (u32)(sb + k - L.slotBase);                          // canonical order inside the level
            atomicMax(&S.best[kn[sb + k]], ((e >> 24) << 24) | (0xFFFFFFu - rel));
        }
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int it = s_head; it != QT_NIL; it = S.nxt[it]) S.order[pos++] = (u16)it;
        s_size = pos;
    }
    __syncthreads();
    const int nsel = s_size;
    u32* so = selOut + (size_t)frame * g.totalSel + L.selBase;
    for (int i = tid; i < nsel; i += 256) {
        if (i < L.selCap) {
            const u32 key = S.best[S.order[i]];
            so[i] = ents[L.slotBase + (0xFFFFFFu - (key & 0xFFFFFFu))];
        } else atomicExch(err, 3);
    }
    if (tid == 0) selCnt[frame * g.nlevels + level] = (u32)min(nsel, L.selCap);
}

#endif  /* ORBX_AB */

// ------------------------------------------------------------------------------------------------
// k_quadtree2: the same DistributeOctTree, with the list surgery itself data-parallel.
// The std::list is an ARRAY in list order; a node's id is its position.  One refinement pass =
//   quadrant histograms (parallel over candidates) -> per-node child counts -> block scans that give every
//   child / surviving node its position in the NEXT list (children of later parents go further to the front,
//   inside a parent n4,n3,n2,n1: exactly what push_front in the order n1..n4 produces) -> keypoints relabelled.
// Final phase (ORBextractor.cc:912-992): candidates sorted by (count, creation seq) with a bitonic sort, the
// early `break` at N becomes a prefix-scan cut.  Node tables ping-pong between two LDS buffers.
// Output order and content are identical to k_quadtree (kept for A/B), at ~1/20 of its latency.
// ------------------------------------------------------------------------------------------------
struct QNode { short4 r; u32 cnt; u32 seq; };

template <int NT>
__device__ __forceinline__ u32 block_scan_excl(u32* a, int n, u32* wsum, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const int per = (n + NT - 1) / NT;
    const int b = min(tid * per, n), e = min(b + per, n);
    u32 s = 0;
    for (int i = b; i < e; ++i) s += a[i];
    u32 inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();                                       // wsum may still be read from a previous call
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 base = 0;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    u32 total = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) total += wsum[k];
    u32 run = base + inc - s;
    for (int i = b; i < e; ++i) { const u32 v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

// two exclusive scans behind the same three barriers (the quadtree's iterations are barrier-latency bound)
template <int NT>
__device__ __forceinline__ void block_scan2_excl(u32* a1, int n1, u32* a2, int n2, u32* wsum, int tid, u32* tot1, u32* tot2) {
    const int lane = tid & 63, wv = tid >> 6;
    const int per1 = (n1 + NT - 1) / NT, per2 = (n2 + NT - 1) / NT;
    const int b1 = min(tid * per1, n1), e1 = min(b1 + per1, n1), b2 = min(tid * per2, n2), e2 = min(b2 + per2, n2);
    u32 s1 = 0, s2 = 0;
    for (int i = b1; i < e1; ++i) s1 += a1[i];
    for (int i = b2; i < e2; ++i) s2 += a2[i];
    u32 i1 = s1, i2 = s2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t1 = __shfl_up(i1, o), t2 = __shfl_up(i2, o);
        if (lane >= o) { i1 += t1; i2 += t2; }
    }
    __syncthreads();                                       // wsum may still be read from a previous call
    if (lane == 63) { wsum[wv] = i1; wsum[NT / 64 + wv] = i2; }
    __syncthreads();
    u32 base1 = 0, base2 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) {
        const u32 w1 = wsum[k], w2 = wsum[NT / 64 + k];
        if (k < wv) { base1 += w1; base2 += w2; }
        t1 += w1; t2 += w2;
    }
    u32 run1 = base1 + i1 - s1, run2 = base2 + i2 - s2;
    for (int i = b1; i < e1; ++i) { const u32 v = a1[i]; a1[i] = run1; run1 += v; }
    for (int i = b2; i < e2; ++i) { const u32 v = a2[i]; a2[i] = run2; run2 += v; }
    __syncthreads();
    *tot1 = t1; *tot2 = t2;
}

__device__ __forceinline__ short4 qt_child_rect(short4 r, int q) {
    const int hx = (r.z - r.x + 1) >> 1, hy = (r.w - r.y + 1) >> 1;
    const int mx = r.x + hx, my = r.y + hy;
    return make_short4((short)((q & 1) ? mx : r.x), (short)((q & 2) ? my : r.y),
                       (short)((q & 1) ? r.z : mx), (short)((q & 2) ? r.w : my));
}

// NT = workgroup size: NT for the usual few thousand candidates per level, 1024 when a level has tens of thousands (1080p
// with 4000 features: the single workgroup per (frame, level) is the whole parallelism of this latency-bound kernel).
template <int NT>
__global__ __launch_bounds__(NT) void k_quadtree2(Geom g, const u32* __restrict__ candCnt, const u32* __restrict__ candEnt,
                                                   u32* dense, u16* kpNode, u32* selOut, u32* selCnt, int* err, int maxCells, int maxIni, int fuseD) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // grid = (frames, levels): consecutive workgroup ids go round-robin over the 8 XCDs, so the fastest-varying index must
    // be the frame -- with the level there and 8 levels, every level-0 (heaviest) workgroup would land on the same XCD
    const int level = blockIdx.y, frame = blockIdx.x;
    const LevelDesc& L = g.lv[level];
    const int tid = threadIdx.x;
    const int cap = g.nodeCap, n2cap = g.sortCap;
    unsigned long long* sortKey = (unsigned long long*)smem;
    QNode* tabA = (QNode*)(sortKey + n2cap);
    QNode* tabB = tabA + cap;
    u32* qc = (u32*)(tabB + cap);                           // [cap][4]; reused as best[] at the end
    u32* s1 = qc + cap * 4;
    u32* s2 = s1 + cap;
    u32* cellOff = s2 + cap;                                // [maxCells + 1]
    u16* childPos = (u16*)(cellOff + maxCells + 1);         // [cap][4]
    u16* newPos = childPos + cap * 4;
    u8* nch = (u8*)(newPos + cap);
    u8* proc = nch + cap;
    // fused first iterations (below): quadrant histograms of the first fuseD (3 or 4) subdivision depths, indexed by the path
    // ((root * 4 + q1) * 4 + q2) ...; the depth-d histogram [maxIni * 4^d] starts at word maxIni * (4^d - 4) / 3
    u32* Hb = (u32*)(smem + (((size_t)(proc + cap - smem) + 3) & ~(size_t)3));
#define QT_H(d) (Hb + maxIni * (((1 << (2 * (d))) - 4) / 3))
    u32* HD = QT_H(fuseD);
    u16* TD = (u16*)(HD + (maxIni << (2 * fuseD)));         // [maxIni * 4^fuseD] deepest path -> node position
    u16* pathA = TD + (maxIni << (2 * fuseD));              // [cap] per node: path | depth << 12
    u16* pathB = pathA + cap;
    __shared__ u32 wsum[2 * (NT / 64)];
    __shared__ int s_size, s_state, s_seqBase, s_cnt, s_ncand;
    const u32* cnts = candCnt + (size_t)frame * g.totalCells + L.cellBase;
    const u32* ents = candEnt + (size_t)frame * g.totalSlots + L.slotBase;
    u32* de = dense + (size_t)frame * g.totalSlots + L.slotBase;     // candidates of this level, densely packed in
    u16* kn = kpNode + (size_t)frame * g.totalSlots + L.slotBase;    // vToDistributeKeys order; their current node
    const int N = L.N;

    // ---- gather the per-cell slot arrays into one dense list (cells are stored in reference order)
    const int nCells = L.nCells;
    for (int c = tid; c < nCells; c += NT) cellOff[c] = min(cnts[c], (u32)L.slotCap);   // (never beyond a cell's slots, whatever the count word holds)
    for (int i = tid; i < cap; i += NT) s1[i] = 0;
    for (int i = tid; i < (maxIni << (2 * fuseD)); i += NT) HD[i] = 0;
    __syncthreads();
    const int nk = (int)block_scan_excl<NT>(cellOff, nCells, wsum, tid);
    if (tid == 0) cellOff[nCells] = (u32)nk;
    __syncthreads();
    {   // four candidates per thread at a time: their (uniform-length, branch-free) searches for the last cell with cellOff <= i
        // interleave, and so do the four loads of the entries
        int top = 1;
        while (top * 2 < nCells) top *= 2;                   // largest power of two < nCells (nCells >= 1)
        for (int i0 = tid; i0 < nk; i0 += 4 * NT) {
            int lo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) lo[u] = 0;
            for (int sft = top; sft > 0; sft >>= 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(i0 + u * NT, nk - 1), c = lo[u] + sft;
                    if (c < nCells && (int)cellOff[c] <= i) lo[u] = c;
                }
            }
            u32 e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = min(i0 + u * NT, nk - 1); e4[u] = ents[lo[u] * L.slotCap + (i - (int)cellOff[lo[u]])]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NT;
                if (i < nk) {
                    const u32 e = e4[u];
                    de[i] = e;
                    const int x = (int)(e & 0xFFF), y = (int)((e >> 12) & 0xFFF);
                    const int root = (int)((float)x / L.hX);            // ORBextractor.cc:740
                    atomicAdd(&s1[root], 1u);
                    // the candidate's path through the first fuseD subdivisions, by the same rectangle arithmetic the iterations use
                    short4 r = make_short4((short)(int)(L.hX * (float)root), 0, (short)(int)(L.hX * (float)(root + 1)), (short)L.qtH);
                    int path = root;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (d < fuseD) { const int q = qt_quadrant(r, x, y); path = path * 4 + q; r = qt_child_rect(r, q); }
                    kn[i] = (u16)path;
                    atomicAdd(&HD[path], 1u);
                }
            }
        }
    }
    __syncthreads();

    // a pass over the level's candidates: the (entry, node) pairs of QT_U candidates per thread are requested together before any
    // is processed -- a rolled loop pays one L2 round trip per candidate (the LDS atomics in the bodies keep the compiler from hoisting
    // the next iteration's loads), and these passes are most of the kernel's time
#ifndef QT_U
#define QT_U 4
#endif
#define QT_FOR_KP(BODY) for (int k0_ = tid; k0_ < nk; k0_ += QT_U * NT) {                                         \
        u32 e4_[QT_U]; int n4_[QT_U];                                                                                \
        _Pragma("unroll") for (int u_ = 0; u_ < QT_U; ++u_) { const int ki = k0_ + u_ * NT; const bool in_ = ki < nk;  \
            e4_[u_] = in_ ? de[ki] : 0u; n4_[u_] = in_ ? (int)kn[ki] : 0; }                                          \
        _Pragma("unroll") for (int u_ = 0; u_ < QT_U; ++u_) { const int ki = k0_ + u_ * NT;                          \
            if (ki < nk) { const u32 e = e4_[u_]; const int nd = n4_[u_]; BODY } } }

    // ---- roots (ORBextractor.cc:695-763)
    if (tid == 0) {
        int size = 0;
        for (int i = 0; i < L.nIni; ++i) {
            if (s1[i] == 0) continue;                            // empty roots are erased
            QNode nd;
            nd.r = make_short4((short)(int)(L.hX * (float)i), 0, (short)(int)(L.hX * (float)(i + 1)), (short)L.qtH);
            nd.cnt = s1[i]; nd.seq = i;
            tabA[size] = nd; pathA[size] = (u16)i; ++size;       // path = root index, depth 0
        }
        s_size = size; s_state = 0; s_seqBase = L.nIni; s_cnt = 0; s_ncand = 0;
    }
    for (int d = fuseD - 1; d >= 1; --d) {                    // coarser histograms by summation
        u32* Hc = QT_H(d); const u32* Hf = QT_H(d + 1);
        for (int i = tid; i < (L.nIni << (2 * d)); i += NT) Hc[i] = Hf[4 * i] + Hf[4 * i + 1] + Hf[4 * i + 2] + Hf[4 * i + 3];
        __syncthreads();
    }

    QNode* A = tabA; QNode* B = tabB;
    u16* PA = pathA; u16* PB = pathB;
    // ---- the first (up to fuseD = three or four) iterations without touching the candidates.  While the list is far from N the loop at :779-895
    // splits EVERY node that holds more than one key, so after d such iterations the splittable nodes are exactly the depth-d
    // prefixes of the candidates' subdivision paths, and a node's four quadrant counts are the depth-(d+1) histogram entries under
    // its path.  The paths and the deepest histogram were computed once in the gather pass; each of these iterations is then
    // node-level work only (same list surgery, same stop / phase-change tests as the general iteration below), and ONE relabel
    // pass at the end gives every candidate its node.  Two passes over the candidates per iteration is what the kernel's time was.
    for (int d = 0; d < fuseD; ++d) {
        const int state = s_state, size = s_size;
        if (state != 0) break;
        const u32* Hn = QT_H(d + 1);
        for (int i = tid; i < size; i += NT) {
            int c = 0;
            if (A[i].cnt > 1) {
                const int p4 = (int)(PA[i] & 0xFFF) * 4;
                const u32 q0 = Hn[p4], q1 = Hn[p4 + 1], q2 = Hn[p4 + 2], q3 = Hn[p4 + 3];
                qc[i * 4] = q0; qc[i * 4 + 1] = q1; qc[i * 4 + 2] = q2; qc[i * 4 + 3] = q3;
                c = (q0 > 0) + (q1 > 0) + (q2 > 0) + (q3 > 0);
            }
            nch[i] = (u8)c;
            s1[i] = c; s2[i] = c > 0 ? 0 : 1;
        }
        __syncthreads();
        u32 totCh_, totKeep_;
        block_scan2_excl<NT>(s1, size, s2, size, wsum, tid, &totCh_, &totKeep_);
        const int totalCh = (int)totCh_, newSize = totalCh + (int)totKeep_;
        for (int i = tid; i < size; i += NT) {
            const int c = nch[i];
            if (c > 0) {
                const int blockStart = totalCh - (int)s1[i] - c;
                const int p4 = (int)(PA[i] & 0xFFF) * 4;
                int fwd = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 qn = qc[i * 4 + q];
                    if (qn > 0) {
                        const int pos = blockStart + (c - 1 - fwd);
                        QNode nd; nd.r = qt_child_rect(A[i].r, q); nd.cnt = qn; nd.seq = (u32)(s_seqBase + (int)s1[i] + fwd);
                        B[pos] = nd; PB[pos] = (u16)((p4 + q) | ((d + 1) << 12));
                        if (qn > 1) atomicAdd(&s_cnt, 1);
                        ++fwd;
                    }
                }
            } else {
                const int pos = totalCh + (int)s2[i];
                B[pos] = A[i]; PB[pos] = PA[i];
            }
        }
        __syncthreads();
        if (tid == 0) {
            const int nToExpand = s_cnt;
            s_cnt = 0;
            s_seqBase += totalCh;
            s_size = newSize;
            if (newSize > cap) { atomicExch(err, 2); s_state = 2; }
            else if (newSize >= N || newSize == size) s_state = 2;
            else if (newSize + nToExpand * 3 > N) s_state = 1;
        }
        QNode* t_ = A; A = B; B = t_;
        u16* tp_ = PA; PA = PB; PB = tp_;
        __syncthreads();
    }
    {   // every candidate's node: deepest path -> position of the node whose path is a prefix of it
        const int size = min(s_size, cap);
        for (int i = tid; i < size; i += NT) {
            const int v = PA[i], sh = 2 * (fuseD - (v >> 12)), p = v & 0xFFF;
            for (int j = p << sh; j < (p + 1) << sh; ++j) TD[j] = (u16)i;
        }
        __syncthreads();
        QT_FOR_KP({ (void)e; kn[ki] = TD[nd]; })
        for (int i = tid; i < size * 4; i += NT) qc[i] = 0;      // quadrant histograms of the next iteration (later ones: zeroed at the end of the previous)
        __syncthreads();
    }

    for (int iter = 0;; ++iter) {
        const int state = s_state, size = s_size;
        if (state == 2) break;
        if (iter > 4096) { if (tid == 0) atomicExch(err, 4); break; }
        // (1) quadrant histograms of every splittable node (qc, s_cnt, s_ncand were reset at the end of the previous iteration)
        QT_FOR_KP({ if (A[nd].cnt > 1) atomicAdd(&qc[nd * 4 + qt_quadrant(A[nd].r, e & 0xFFF, (e >> 12) & 0xFFF)], 1u); })
        __syncthreads();
        int totalCh = 0, newSize = 0;
        if (state == 0) {                                                    // ORBextractor.cc:779-895
            for (int i = tid; i < size; i += NT) {
                int c = 0;
                if (A[i].cnt > 1) c = (qc[i * 4] > 0) + (qc[i * 4 + 1] > 0) + (qc[i * 4 + 2] > 0) + (qc[i * 4 + 3] > 0);
                nch[i] = (u8)c; proc[i] = c > 0;
                s1[i] = c; s2[i] = c > 0 ? 0 : 1;
            }
            __syncthreads();
            u32 totCh_, totKeep_;
            block_scan2_excl<NT>(s1, size, s2, size, wsum, tid, &totCh_, &totKeep_);
            totalCh = (int)totCh_;
            newSize = totalCh + (int)totKeep_;
            for (int i = tid; i < size; i += NT) {
                const int c = nch[i];
                if (c > 0) {
                    const int blockStart = totalCh - (int)s1[i] - c;
                    int fwd = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32 qn = qc[i * 4 + q];
                        if (qn > 0) {
                            const int pos = blockStart + (c - 1 - fwd);
                            QNode nd; nd.r = qt_child_rect(A[i].r, q); nd.cnt = qn; nd.seq = (u32)(s_seqBase + (int)s1[i] + fwd);
                            B[pos] = nd; childPos[i * 4 + q] = (u16)pos;
                            if (qn > 1) atomicAdd(&s_cnt, 1);
                            ++fwd;
                        }
                    }
                } else {
                    const int pos = totalCh + (int)s2[i];
                    B[pos] = A[i]; newPos[i] = (u16)pos;
                }
            }
        } else {                                                             // ORBextractor.cc:912-992
            int n2 = 1;
            while (n2 < size) n2 <<= 1;
            for (int i = tid; i < n2; i += NT) {
                unsigned long long key = 0;
                if (i < size && A[i].cnt > 1) {
                    key = ((unsigned long long)A[i].cnt << 40) | ((unsigned long long)A[i].seq << 16) | (unsigned)i;
                    atomicAdd(&s_ncand, 1);
                }
                sortKey[i] = key;
                if (i < size) {
                    int c = 0;
                    if (A[i].cnt > 1) c = (qc[i * 4] > 0) + (qc[i * 4 + 1] > 0) + (qc[i * 4 + 2] > 0) + (qc[i * 4 + 3] > 0);
                    nch[i] = (u8)c; proc[i] = 0;
                }
            }
            __syncthreads();
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < n2; i += NT) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = sortKey[i], b = sortKey[ixj];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { sortKey[i] = b; sortKey[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            const int ncand = s_ncand;
            // processing order k = 0.. : largest (count, seq) first = sortKey[n2-1-k]
            for (int k = tid; k < ncand; k += NT) s1[k] = (u32)nch[(int)(sortKey[n2 - 1 - k] & 0xFFFF)] - 1u;
            __syncthreads();
            block_scan_excl<NT>(s1, ncand, wsum, tid);
            // the reference stops right after the split that makes size >= N: count the splits that leave size < N
            for (int k = tid; k < ncand; k += NT) {
                const int gain = (int)nch[(int)(sortKey[n2 - 1 - k] & 0xFFFF)] - 1;
                if (size + (int)s1[k] + gain < N) atomicAdd(&s_cnt, 1);
            }
            __syncthreads();
            const int P = min(ncand, s_cnt + 1);
            __syncthreads();
            if (tid == 0) s_cnt = 0;
            for (int k = tid; k < ncand; k += NT) {
                const int id = (int)(sortKey[n2 - 1 - k] & 0xFFFF);
                s1[k] = k < P ? (u32)nch[id] : 0u;
                if (k < P) proc[id] = 1;
            }
            __syncthreads();
            for (int i = tid; i < size; i += NT) s2[i] = proc[i] ? 0 : 1;
            __syncthreads();
            u32 totCh_, totKeep_;
            block_scan2_excl<NT>(s1, ncand, s2, size, wsum, tid, &totCh_, &totKeep_);
            totalCh = (int)totCh_;
            newSize = totalCh + (int)totKeep_;
            for (int k = tid; k < P; k += NT) {
                const int i = (int)(sortKey[n2 - 1 - k] & 0xFFFF);
                const int c = nch[i];
                const int blockStart = totalCh - (int)s1[k] - c;
                int fwd = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 qn = qc[i * 4 + q];
                    if (qn > 0) {
                        const int pos = blockStart + (c - 1 - fwd);
                        QNode nd; nd.r = qt_child_rect(A[i].r, q); nd.cnt = qn; nd.seq = (u32)(s_seqBase + (int)s1[k] + fwd);
                        B[pos] = nd; childPos[i * 4 + q] = (u16)pos;
                        ++fwd;
                    }
                }
            }
            for (int i = tid; i < size; i += NT)
                if (!proc[i]) { const int pos = totalCh + (int)s2[i]; B[pos] = A[i]; newPos[i] = (u16)pos; }
        }
        __syncthreads();
        // (3) relabel the keypoints; the next iteration's histograms start from zero
        for (int i = tid; i < min(newSize, cap) * 4; i += NT) qc[i] = 0;
        QT_FOR_KP({ kn[ki] = proc[nd] ? childPos[nd * 4 + qt_quadrant(A[nd].r, e & 0xFFF, (e >> 12) & 0xFFF)] : newPos[nd]; })
        __syncthreads();
        if (tid == 0) {
            const int nToExpand = s_cnt;
            s_cnt = 0; s_ncand = 0;
            s_seqBase += totalCh;
            s_size = newSize;
            if (newSize > cap) { atomicExch(err, 2); s_state = 2; }
            else if (newSize >= N || newSize == size) s_state = 2;
            else if (state == 0 && newSize + nToExpand * 3 > N) s_state = 1;
        }
        QNode* t_ = A; A = B; B = t_;
        __syncthreads();
    }
    // ---- one keypoint per node: first maximum of `response` in candidate order (ORBextractor.cc:1005-1030)
    const int nsel = s_size;
    u32* best = qc;
    for (int i = tid; i < nsel; i += NT) best[i] = 0;
    __syncthreads();
    QT_FOR_KP({ atomicMax(&best[nd], ((e >> 24) << 24) | (0xFFFFFFu - (u32)ki)); })
    __syncthreads();
    u32* so = selOut + (size_t)frame * g.totalSel + L.selBase;
    for (int i = tid; i < nsel; i += NT) {
        if (i < L.selCap) so[i] = de[0xFFFFFFu - (best[i] & 0xFFFFFFu)];
        else atomicExch(err, 3);
    }
    if (tid == 0) selCnt[frame * g.nlevels + level] = (u32)min(nsel, L.selCap);
#undef QT_FOR_KP
#undef QT_H
}

// ------------------------------------------------------------------------------------------------
// k_slots: one workgroup per frame.  Walks the levels in order, fixes up coordinates/octave/size
// (ORBextractor.cc:1161-1176), scales to level-0 coordinates and assigns the output row:
// lapping keypoints fill the output from the back, the rest from the front (:1633-1655).
// ------------------------------------------------------------------------------------------------
struct KpWork { short level, x, y, pad; int slot; };       // 12 bytes

struct KpOut { float x, y, size, angle, response; int octave, class_id; };

__global__ __launch_bounds__(256) void k_slots(Geom g, const u32* __restrict__ selOut, const u32* __restrict__ selCnt,
                                               const int* __restrict__ lap01, KpOut* kps, KpWork* work,
                                               int* nOut, int* monoOut) {
    __shared__ int s_wave[4];
    __shared__ int s_lvOff[13];
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
        int o = 0;
        for (int l = 0; l < g.nlevels; ++l) { s_lvOff[l] = o; o += (int)selCnt[frame * g.nlevels + l]; }
        s_lvOff[g.nlevels] = o;
    }
    __syncthreads();
    const int n = s_lvOff[g.nlevels];
    const int lap0 = lap01 ? lap01[2 * frame] : 0, lap1 = lap01 ? lap01[2 * frame + 1] : 0;
    const u32* so = selOut + (size_t)frame * g.totalSel;
    KpOut* ko = kps + (size_t)frame * g.kpCap;
    KpWork* wo = work + (size_t)frame * g.kpCap;
    int lapBefore = 0;                                      // lapping keypoints before this chunk
    for (int p0 = 0; p0 < n; p0 += 256) {
        const int p = p0 + tid;
        bool isLap = false;
        int level = 0, x = 0, y = 0, r = 0;
        float fx = 0, fy = 0;
        if (p < n) {
            while (p >= s_lvOff[level + 1]) ++level;
            const u32 e = so[g.lv[level].selBase + (p - s_lvOff[level])];
            x = (int)(e & 0xFFF) + 16; y = (int)((e >> 12) & 0xFFF) + 16; r = (int)(e >> 24);
            fx = (float)x; fy = (float)y;
            if (level != 0) { fx *= g.lv[level].sf; fy *= g.lv[level].sf; }
            isLap = fx >= (float)lap0 && fx <= (float)lap1;
        }
        const unsigned long long m = __ballot(isLap);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int before = lapBefore;
        for (int k = 0; k < wv; ++k) before += s_wave[k];
        before += __popcll(m & ((1ull << lane) - 1ull));
        const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (p < n) {
            const int slot = isLap ? (n - 1 - before) : (p - before);
            KpOut k;
            k.x = fx; k.y = fy; k.size = g.lv[level].patch; k.angle = -1.f; k.response = (float)r;
            k.octave = level; k.class_id = -1;
            ko[slot] = k;
            KpWork w; w.level = (short)level; w.x = (short)x; w.y = (short)y; w.pad = 0; w.slot = slot;
            wo[p] = w;
        }
        lapBefore += tot;
        __syncthreads();
    }
    if (tid == 0) { nOut[frame] = n; monoOut[frame] = n - lapBefore; }
}

// ------------------------------------------------------------------------------------------------
// k_blur: cv::GaussianBlur 7x7 sigma 2 reflect-101, taps [18,34,48,56,48,34,18]/256:
// horizontal pass exact in Q8.8, vertical pass to Q16.16, one rounding (+32768)>>16.
// Tile 64x32 outputs per workgroup, source window (70x38) staged in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {   // valid for -n < p < 2n-1 (callers stay within 3 px)
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}

// Streaming form: one wavefront owns a 248-pixel-wide column strip (4 px per lane, one dword; lanes 0 and 63 are halo) and walks
// BL_R output rows downwards.  Per source row: ONE coalesced dword load per lane, neighbours' dwords through
// DPP wave_shr/wave_shl (one VALU op, no LDS), horizontal 7-tap as two v_dot4_u32_u8 per pixel, then the vertical 7-tap
// as four v_dot2_u32_u16 over a rotating 7-row register window of (previous row, row) sums.  Reflect-101 at the left/right image edge is done with v_perm selectors chosen on
// the host from (width & 3); rows reflect through the row index.
// ------------------------------------------------------------------------------------------------
// k_blur3: the same blur on the matrix cores.  Both passes are small integer matrix products with banded (Toeplitz)
// weight matrices, and the arithmetic is exact in int8 x int8 -> int32:
//   pass 1   H[32 rows][32 cols] = X[32 rows][64 px] . Th[64 px][32 cols]      (2 x v_mfma_i32_32x32x32_i8)
//            X as (pixel - 128); Th[p][c] = sum of the taps whose reflect-101 source pixel for output column c is p (7 taps,
//            sum 256), so the accumulator holds the reference's Q8.8 row sum (<= 65280) minus 32768 exactly, borders
//            included: a signed 16-bit number.
//   pass 2   V = Tv[32 out rows][32 H rows] . H   with H split into its low and high bytes (2 MFMAs): the accumulator tile
//            of pass 1 (column on the lane, 16 rows in the registers) IS the B operand layout of a product that sums over
//            its rows -- 4 v_perm gather the bytes of 4 registers, no lane movement, no LDS; Tv carries the k order of the
//            accumulator registers.  out = ((accHi << 8) + accLo) >> 16 with the -128 offsets and the +32768 rounding folded
//            into accLo's start value.
// A tile yields 26 output rows from 32 source rows (3-row aprons recomputed: 23 % more pass-1 work, no state between tiles).
// A workgroup owns a 128-px column group (one 32-px block per wave) and walks up to B3_CHUNK tiles down it; the 32 x 160 B
// source window of a tile is staged through LDS with full-line loads, double-buffered, one barrier per tile (the A-operand
// layout wants 16 bytes of a different row in every lane: read straight from memory that is 32-byte pieces of 128-byte
// lines and the kernel ran at the L2's line rate, 0.72 ms).  The weight fragments depend only on the
// geometry -- Th on (level, strip), Tv on (level, tile row) -- and are tabulated by the host in fragment order when the
// geometry is built (built in the kernel they cost more than the tiles themselves); border folds are just table entries.
// ------------------------------------------------------------------------------------------------
#define B3_ROWS 26
#ifndef B3_XCD
#define B3_XCD 1                                            // adjacent column groups on one XCD: their shared halo lines hit in L2
#endif
#ifndef B3_CHUNK
#define B3_CHUNK 8
#endif
struct Blur3Task { short level, x0, t0, nt; int th, tv; };   // th / tv: entries of the weight tables (tv: of tile t0)
typedef int b3_i32x4 __attribute__((ext_vector_type(4)));
typedef int b3_i32x16 __attribute__((ext_vector_type(16)));

// weight of source position p in output position `out` of an n-long line: sum_i [reflect101(out + i - 3) == p] * K[i]
// (host side: the weight fragments are tabulated per column strip and per tile row when the geometry is built)
static inline int b3_toep(int p, int out, int n) {
    const int K[7] = {18, 34, 48, 56, 48, 34, 18};
    int wsum = 0;
    for (int i = 0; i < 7; ++i) {
        int q = out + i - 3;
        q = q < 0 ? -q : q;
        q = q >= n ? 2 * n - 2 - q : q;
        wsum += q == p ? K[i] : 0;
    }
    return wsum;
}
// pass-1 weights (B operand) of the strip at x0: [k-step s][lane] uint4; lane (r, hh) holds column x0 + r, element j of
// k-step s <-> pixel x0 - 16 + 32 s + 16 hh + j
static inline void b3_build_th(int x0, int w, u32* out /* [2][64][4] */) {
    for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
                const int r = lane & 31, hh = lane >> 5;
                u32 v = 0;
                for (int b = 0; b < 4; ++b) v |= (u32)b3_toep(x0 - 16 + 32 * s + 16 * hh + 4 * e + b, x0 + r, w) << (8 * b);
                out[(s * 64 + lane) * 4 + e] = v;
            }
}
// pass-2 weights (A operand) of the tile at y0: [lane] uint4; lane (r, hh) holds output row y0 + r (26 valid), element j <->
// H row (j&3) + 8 (j>>2) + 4 hh of the tile = image row y0 - 3 + that (the register order of the pass-1 accumulator)
static inline void b3_build_tv(int y0, int h, u32* out /* [64][4] */) {
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int r = lane & 31, hh = lane >> 5;
            u32 v = 0;
            for (int b = 0; b < 4; ++b) {
                const int j = 4 * e + b, krow = (j & 3) + 8 * (j >> 2) + 4 * hh;
                v |= (r < B3_ROWS ? (u32)b3_toep(y0 - 3 + krow, y0 + r, h) : 0u) << (8 * b);
            }
            out[lane * 4 + e] = v;
        }
}

#define B3_OROW 144                                        // output staging row stride (128 B + 16 B pad)
#define B3_LROW 176                                        // LDS row stride: 160 B of pixels + 16 B pad (b128 reads conflict-free)
__global__ __launch_bounds__(256) void k_blur3(Geom g, const u8* const* l0, int l0pitch, const u8* pyr, u8* blr,
                                               const Blur3Task* __restrict__ tasks,
                                               const uint4* __restrict__ thTab, const uint4* __restrict__ tvTab) {
#if __HIP_DEVICE_COMPILE__
    __shared__ __attribute__((aligned(16))) u8 xs[2][32 * B3_LROW];
    __shared__ __attribute__((aligned(16))) u8 ob[2][32 * B3_OROW];   // blurred tile of the workgroup, 32 rows x 128 px
    const int tid = threadIdx.x, lane = tid & 63;
    const int cb = __builtin_amdgcn_readfirstlane(tid >> 6);            // this wave's 32-px column block of the 128-px group
    Blur3Task t = tasks[B3_XCD ? xcd_task(blockIdx.x, gridDim.x) : blockIdx.x];   // one task per workgroup: (level, 128-px group, tile chunk)
    t.level = (short)__builtin_amdgcn_readfirstlane(t.level);
    t.x0 = (short)__builtin_amdgcn_readfirstlane(t.x0);
    t.t0 = (short)__builtin_amdgcn_readfirstlane(t.t0);
    t.nt = (short)__builtin_amdgcn_readfirstlane(t.nt);
    t.th = __builtin_amdgcn_readfirstlane(t.th);
    t.tv = __builtin_amdgcn_readfirstlane(t.tv);
    const int frame = blockIdx.y;
    const LevelDesc& L = g.lv[t.level];
    int sp;
    const u8* im = level_ptr(g, l0, l0pitch, pyr, frame, t.level, &sp);
    u8* dst = blr + (size_t)frame * g.blrFrameBytes + L.boff;
    const int w = L.w, h = L.h, dp = L.pitch, xg = t.x0, x0 = xg + 32 * cb;
    const bool live = x0 < w;                                           // wave-uniform: a block past the right edge only helps loading
    const int r = lane & 31, hh = lane >> 5;
    b3_i32x4 th[2];
    {
        const int e = live ? t.th + cb : t.th;
        const uint4 u0 = thTab[(size_t)e * 128 + lane], u1 = thTab[(size_t)e * 128 + 64 + lane];
        th[0][0] = (int)u0.x; th[0][1] = (int)u0.y; th[0][2] = (int)u0.z; th[0][3] = (int)u0.w;
        th[1][0] = (int)u1.x; th[1][1] = (int)u1.y; th[1][2] = (int)u1.z; th[1][3] = (int)u1.w;
    }
    // staging: 32 source rows x 160 B (pixels xg - 16 .. xg + 143) per tile = 320 16-byte pieces; thread -> piece tid and
    // piece min(256 + tid, 319) (only the first 64 threads park the second one).  Every load is unconditional on a clamped
    // row / column -- a piece outside the image only ever meets zero weights, so any real pixels will do -- which keeps
    // branches (and the compiler's conservative vmcnt waits) out of the loop.
    const int it1 = min(256 + tid, 319);
    const int pr0 = tid / 10, ps0 = tid - pr0 * 10, pr1 = it1 / 10, ps1 = it1 - pr1 * 10;
    const int px0 = xg - 16 + 16 * ps0, px1 = xg - 16 + 16 * ps1;
    const int pxc0 = (px0 >= 0 && px0 < sp) ? px0 : 0, pxc1 = (px1 >= 0 && px1 < sp) ? px1 : 0;
    const int ldsW0 = pr0 * B3_LROW + 16 * ps0, ldsW1 = pr1 * B3_LROW + 16 * ps1;
    const int ldsR = r * B3_LROW + 32 * cb + 16 * hh;
    int yr0 = t.t0 * B3_ROWS - 3 + pr0, yr1 = t.t0 * B3_ROWS - 3 + pr1;          // source rows of this thread's pieces
    u32 tvo = (u32)(t.tv * 1024 + lane * 16);
    uint4 n0, n1;
    auto fetch = [&]() {
        n0 = gload128u(im, (u32)(__mul24(min(max(yr0, 0), h - 1), sp) + pxc0));
        n1 = gload128u(im, (u32)(__mul24(min(max(yr1, 0), h - 1), sp) + pxc1));
        yr0 += B3_ROWS; yr1 += B3_ROWS;
    };
    auto park = [&](int buf) {
        const u32 m = 0x80808080u;
        *(uint4*)(xs[buf] + ldsW0) = make_uint4(n0.x ^ m, n0.y ^ m, n0.z ^ m, n0.w ^ m);
        if (tid < 64) *(uint4*)(xs[buf] + ldsW1) = make_uint4(n1.x ^ m, n1.y ^ m, n1.z ^ m, n1.w ^ m);
    };
    // memory operations retire in order and the compiler cannot count this kernel's predicated stores: per tile the order is
    // [next Tv, next pixels] ... park (waits for both, nothing younger outstanding) ... stores, so the only wait of the loop
    // is the one before parking and it never includes a store of the same tile
    uint4 tvq = gload128u(tvTab, tvo), tvn = tvq;
    fetch();
    park(0);
    __syncthreads();
    b3_i32x16 c2;                                                                  // pass-2 start value: offsets of both passes + rounding
#pragma unroll
    for (int i = 0; i < 16; ++i) c2[i] = 256 * 32768 + 32768 + 32768;
    // write-out: the workgroup's 26 x 128 px tile leaves as 16-byte pieces, 8 per row: full 128-byte lines per row
    // (tiled output: 8 consecutive lanes take 8 consecutive rows of one 16-byte column = one tile's line when the rows are aligned)
    const bool tiledOut = g.blurTiled != 0;
    const int orow = tiledOut ? (tid & 7) + 8 * (tid >> 6) : tid >> 3, oseg = tiledOut ? (tid >> 3) & 7 : tid & 7;
    const bool ocol = xg + 16 * oseg < w && orow < B3_ROWS;
    const int obR = orow * B3_OROW + 16 * oseg;
    int hrem = h - t.t0 * B3_ROWS - orow;                                          // > 0: this thread's output row is inside the image
    // Tiled output (g.blurTiled): the blurred level is only ever read back as 37-row x 40-byte patches around keypoints
    // (k_orient_desc2), and that gather is bound by the number of 128-byte lines it touches: as 16 x 8-px tiles of one line each
    // a patch covers ~21 lines instead of ~48.  Piece (row y, 16-byte column c) lives at ((y >> 3) * btpr + c) * 128 + (y & 7) * 16;
    // the 8 rows x 8 pieces a wave stores are still whole or half lines (L2 merges the halves of a tile written by two steps).
    const bool tiled = tiledOut;
    int yo = t.t0 * B3_ROWS + orow;
    const u32 tcol = (u32)(((xg >> 4) + oseg) << 7);
    u32 so = (u32)(yo * dp + xg + 16 * oseg);
    const u32 sstep = (u32)(B3_ROWS * dp);
    for (int k = 0; k < t.nt; ++k) {
        const int buf = k & 1;
        const bool more = k + 1 < t.nt;
        if (more) {                                                     // in flight while this tile is computed
            tvo += 1024u;
            tvn = gload128u(tvTab, tvo);
            fetch();
        }
        b3_i32x16 aH, aL;
        if (live) {
            const uint4 a0 = *(const uint4*)(xs[buf] + ldsR);
            const uint4 a1 = *(const uint4*)(xs[buf] + ldsR + 32);
            b3_i32x4 A0, A1;
            A0[0] = (int)a0.x; A0[1] = (int)a0.y; A0[2] = (int)a0.z; A0[3] = (int)a0.w;
            A1[0] = (int)a1.x; A1[1] = (int)a1.y; A1[2] = (int)a1.z; A1[3] = (int)a1.w;
            b3_i32x16 acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, th[0], (b3_i32x16)(0), 0, 0, 0);   // H - 32768: signed 16-bit
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, th[1], acc, 0, 0, 0);
            // ---- H - 32768 (signed 16-bit, one column per lane, 16 rows in the registers) -> low / high byte fragments: the
            // high byte is a signed int8 as it stands, the low byte becomes one by - 128 (the 128s are part of c2)
            b3_i32x4 blo, bhi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 p01 = __builtin_amdgcn_perm((u32)acc[4 * e + 1], (u32)acc[4 * e], 0x05010400u);       // r0.b0 r1.b0 r0.b1 r1.b1
                const u32 p23 = __builtin_amdgcn_perm((u32)acc[4 * e + 3], (u32)acc[4 * e + 2], 0x05010400u);
                blo[e] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
                bhi[e] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
            }
            b3_i32x4 tv;
            tv[0] = (int)tvq.x; tv[1] = (int)tvq.y; tv[2] = (int)tvq.z; tv[3] = (int)tvq.w;
            aH = __builtin_amdgcn_mfma_i32_32x32x32_i8(tv, bhi, (b3_i32x16)(0), 0, 0, 0);
            aL = __builtin_amdgcn_mfma_i32_32x32x32_i8(tv, blo, c2, 0, 0, 0);
        }
        if (live) {
            // ---- lane = column x0 + r, register i = output row (i&3) + 8 (i>>2) + 4 hh of the tile, value in bits 16..23: one LDS
            // byte write each into the workgroup's output tile (double-buffered; the barrier below is the loop's only one), which
            // then leaves as full 128-byte rows.  (Global byte stores crawl at a lane per clock, dword stores straight from this
            // layout write 32-byte runs: 0.55 ms; a 4x4 byte transpose across lane quads -- 3 v_perm + 2 DPP moves + 2 v_perm per
            // four rows -- ahead of dword LDS writes costs 28 VALU per tile more than these 16 LDS writes and the pass is
            // VALU-bound.)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const u32 R = (u32)((aH[i] << 8) + aL[i]);
                ob[buf][((i & 3) + 8 * (i >> 2) + 4 * hh) * B3_OROW + 32 * cb + r] = (u8)(R >> 16);   // ds_write_b8_d16_hi
            }
        }
        // the next tile's pixels go to LDS before this tile's stores are issued: the wait for them then never includes a store
        if (more) { park(buf ^ 1); tvq = tvn; }                         // (Tv was requested before the pixels: it has landed too)
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the stores and the prefetch
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ocol && hrem > 0) {
            const uint4 v = *(const uint4*)(ob[buf] + obR);
            const u32 sa = tiled ? (u32)(__mul24(yo >> 3, L.btpr) << 7) + tcol + (u32)((yo & 7) << 4) : so;
            *(__attribute__((address_space(1))) uint4*)((__attribute__((address_space(1))) u8*)dst + sa) = v;
        }
        hrem -= B3_ROWS; so += sstep; yo += B3_ROWS;
    }
#endif
}

#define BL_R 32
struct BlurTask { short level, g0, y0, pad; };            // g0 = first dword column of the strip, y0 = first output row

struct BlurSel { u32 selB[12], selC[12]; };               // per level: right-edge fix-up selectors

#ifdef ORBX_AB   /* A/B reference, not in the product library */
__global__ __launch_bounds__(256) void k_blur2(Geom g, const u8* const* l0, int l0pitch, const u8* pyr, u8* blr,
                                               const BlurTask* __restrict__ tasks, int ntasks, BlurSel bs) {
    const int lane = threadIdx.x & 63;
    const int ti = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ti >= ntasks) return;
    BlurTask t = tasks[ti];
    t.level = (short)__builtin_amdgcn_readfirstlane(t.level);   // wave-uniform: keeps Geom indexing on the scalar unit
    t.g0 = (short)__builtin_amdgcn_readfirstlane(t.g0);
    t.y0 = (short)__builtin_amdgcn_readfirstlane(t.y0);
    const int frame = blockIdx.y;
    const LevelDesc& L = g.lv[t.level];
    int sp;
    const u8* im = level_ptr(g, l0, l0pitch, pyr, frame, t.level, &sp);
    u8* dst = blr + (size_t)frame * g.blrFrameBytes + L.boff;
    const int w = L.w, h = L.h;
    const int gl = (w - 1) >> 2;                           // last dword column holding image pixels
    // lanes 0 and 63 are halo lanes (they load and filter but do not store) -- except at the image's own left / right edge,
    // where the neighbour is the reflection and the lane can store too (752 px = 63 + 62 + 63 dwords: exactly 3 strips)
    const int lead = t.g0 > 0 ? 1 : 0;
    const int gc = t.g0 - lead + lane;
    const int gcl = min(max(gc, 0), gl);                   // loads are unconditional on a clamped column (no exec juggling)
    const bool isGl = gc == gl, isG0 = gc == 0;
    const u32 selB = bs.selB[t.level], selC = bs.selC[t.level];
    const u32 K1 = 18u | (34u << 8) | (48u << 16) | (56u << 24), K2 = 48u | (34u << 8) | (18u << 16);
    const int nrows = min(BL_R, h - t.y0) + 6;
    // vertical window: Q[k][i] = (row sum of the previous source row, row sum of the row in slot k) as a u16 pair
    // (row sums <= 256*255), so that the vertical 7-tap is four v_dot2_u32_u16 per pixel
    u32 Q[7][4], hprev[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 7; ++k) { Q[k][0] = Q[k][1] = Q[k][2] = Q[k][3] = 0; }
    const us2 KA = as_us2(18u | (34u << 16)), KB = as_us2(48u | (56u << 16)), KC = as_us2(48u | (34u << 16)), KD = as_us2(18u << 16);
    // software pipeline: the loads of row group n+1 are issued BEFORE the arithmetic and stores of group n, so
    // waiting for them never waits for younger stores (vmcnt retires in issue order on gfx9)
    u32 Bn[7];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            // rows past the block are clamped, not skipped: no zero-initialised destination, no branch (their sums are never used)
            const int r = min(r0 + k, nrows - 1);
            const int ys = reflect101(t.y0 - 3 + r, h);
            Bn[k] = gload32u(im, (u32)(ys * sp) + (u32)gcl * 4u);            // 32-bit offset from the level base (one s_mul, one v_add)
        }
    };
    fetch(0);
    const bool doStore = lane >= lead && gc <= gl && (lane < 63 || gc == gl);
    for (int r0 = 0; r0 < nrows; r0 += 7) {
        u32 Bq[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) Bq[k] = Bn[k];
        fetch(r0 + 7);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int r = r0 + k;
            if (r < nrows) {
                const u32 B = Bq[k];
                u32 A = (u32)__builtin_amdgcn_mov_dpp((int)B, 0x138, 0xf, 0xf, true);             // wave_shr:1 = left neighbour's dword (no `old` operand to materialise)
                const u32 Bf = isGl ? __builtin_amdgcn_perm(A, B, selB) : B;                     // reflect the bytes beyond the last pixel
                u32 C = (u32)__builtin_amdgcn_mov_dpp((int)Bf, 0x130, 0xf, 0xf, true);            // wave_shl:1 = right neighbour's dword
                C = isGl ? __builtin_amdgcn_perm(A, B, selC) : C;                                 // the dword beyond the image
                A = isG0 ? __builtin_amdgcn_perm(C, Bf, 0x01020304u) : A;                         // pixels -4..-1 <- 4,3,2,1
                const u32 w1[4] = {__builtin_amdgcn_alignbyte(Bf, A, 1), __builtin_amdgcn_alignbyte(Bf, A, 2),
                                   __builtin_amdgcn_alignbyte(Bf, A, 3), Bf};
                const u32 w2[4] = {__builtin_amdgcn_alignbyte(C, Bf, 1), __builtin_amdgcn_alignbyte(C, Bf, 2),
                                   __builtin_amdgcn_alignbyte(C, Bf, 3), C};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 hs = __builtin_amdgcn_udot4(w2[i], K2, __builtin_amdgcn_udot4(w1[i], K1, 0u, false), false);
                    Q[k][i] = hprev[i] | (hs << 16);
                    hprev[i] = hs;
                }
                if (r >= 6) {
                    // rows r-6..r: (r-6,r-5).(18,34) + (r-4,r-3).(48,56) + (r-2,r-1).(48,34) + (r-1,r).(0,18), + 0.5 ulp; the sum is
                    // < 2^24, so the rounded result is byte 2 of the accumulator
                    u32 acc[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        u32 a = __builtin_amdgcn_udot2(as_us2(Q[(k + 2) % 7][i]), KA, 32768u, false);
                        a = __builtin_amdgcn_udot2(as_us2(Q[(k + 4) % 7][i]), KB, a, false);
                        a = __builtin_amdgcn_udot2(as_us2(Q[(k + 6) % 7][i]), KC, a, false);
                        acc[i] = __builtin_amdgcn_udot2(as_us2(Q[k][i]), KD, a, false);
                    }
                    const u32 packed = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
                    if (doStore) gstore32u(dst, (u32)((t.y0 + r - 6) * L.pitch) + (u32)gc * 4u, packed);
                }
            }
        }
    }
}
#endif  /* ORBX_AB */

// ------------------------------------------------------------------------------------------------
// k_orient_desc: one wavefront per keypoint.
//  * IC_Angle (ORBextractor.cc:91-138): integer moments over the 749-pixel disc on the UN-blurred level,
//    wave-reduced with __shfl_xor, then cv::fastAtan2 in float32 without contraction.
//  * computeOrbDescriptor (:150-203): lane l evaluates pairs 4l..4l+3 on the BLURRED level,
//    nibbles are merged across 8-lane groups with shuffles; lanes 0,8,..,56 store one dword each.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    // branch-free: both branches of cv::fastAtan2 divide the smaller by (the larger + eps) and run the same polynomial; one IEEE
    // division per call instead of two when the keypoints of a wave disagree about the branch
    const bool xbig = ax >= ay;
    const float c = (xbig ? ay : ax) / ((xbig ? ax : ay) + eps), c2 = c * c;
    const float pl = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = xbig ? pl : 90.f - pl;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

#define OD_FN __device__ __forceinline__
#include "od_sincos.h"

struct Umax { int v[16]; };

#ifdef ORBX_AB   /* A/B reference, not in the product library */
__global__ __launch_bounds__(256) void k_orient_desc(Geom g, const u8* const* l0, int l0pitch, const u8* pyr,
                                                     const u8* blr, const KpWork* __restrict__ work,
                                                     const int* __restrict__ nOut, KpOut* kps, u8* desc,
                                                     const int8_t* __restrict__ pattern, Umax um) {
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= nOut[frame]) return;
    const KpWork w = work[(size_t)frame * g.kpCap + p];
    int sp;
    const u8* im = level_ptr(g, l0, l0pitch, pyr, frame, w.level, &sp);
    const u8* c = im + (size_t)w.y * sp + w.x;
    int m10 = 0, m01 = 0;
    for (int i = lane; i < 31 * 32; i += 64) {
        const int v = (i >> 5) - 15, u = (i & 31) - 15;
        const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
        if (au <= um.v[av]) {
            const int val = c[v * sp + u];
            m10 += u * val; m01 += v * val;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // descriptor
    const LevelDesc& L = g.lv[w.level];
    const u8* bc = blr + (size_t)frame * g.blrFrameBytes + L.boff + (size_t)w.y * L.pitch + w.x;
    const int bp = L.pitch;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = angle * factorPI;
    const float a = (float)cos((double)ar), b = (float)sin((double)ar);
    const int4 raw = *(const int4*)(pattern + 16 * lane);
    const int wds[4] = {raw.x, raw.y, raw.z, raw.w};
    u32 nib = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = (float)(int8_t)(wds[k] & 0xFF), y0 = (float)(int8_t)((wds[k] >> 8) & 0xFF);
        const float x1 = (float)(int8_t)((wds[k] >> 16) & 0xFF), y1 = (float)(int8_t)((wds[k] >> 24) & 0xFF);
        const int t0 = bc[__float2int_rn(x0 * b + y0 * a) * bp + __float2int_rn(x0 * a - y0 * b)];
        const int t1 = bc[__float2int_rn(x1 * b + y1 * a) * bp + __float2int_rn(x1 * a - y1 * b)];
        nib |= (u32)(t0 < t1) << k;
    }
    u32 word = nib << (4 * (lane & 7));
    word |= __shfl_xor(word, 1);
    word |= __shfl_xor(word, 2);
    word |= __shfl_xor(word, 4);
    const size_t row = (size_t)frame * g.kpCap + w.slot;
    if ((lane & 7) == 0) ((u32*)(desc + row * 32))[lane >> 3] = word;
    if (lane == 0) kps[row].angle = angle;
}
#endif  /* ORBX_AB */

// ------------------------------------------------------------------------------------------------
// k_orient_desc2: FOUR keypoints per wavefront (16 lanes each) -- same arithmetic as k_orient_desc, a quarter of
// the waves and 4x the bytes in flight per wave (the first version was pure memory latency: 3 dependent trips
// per keypoint at 642 VALU instructions each).
//  * IC_Angle: the 31 patch rows are fetched as 8 dwords per row starting exactly at cx-15 (unaligned global loads: 248
//    dword loads per keypoint, 4 per lane) and reduced with v_dot4 against per-row byte weights held in LDS; moments
//    are wave-reduced once per keypoint.
//  * fastAtan2 / sin / cos run ONCE per wave for 4 different angles (lane group = keypoint).
//  * rBRIEF: lane s of a group evaluates pairs {16q + s}; one __ballot per q yields, for all 4 keypoints at once,
//    descriptor bytes 2q and 2q+1 already in LSB-first bit order (bit s of the group's 16-bit field).
//    The 512 pattern points sit in LDS as floats (staged once per workgroup).
// ------------------------------------------------------------------------------------------------
#ifndef OD_XCD
#define OD_XCD 1
#endif
#define OD_PPITCH 40
#define OD_PPITCH_T 48                                     // tiled blurred level: 6 aligned 8-byte pieces per patch row
#define OD_WTAB (17 * 8)                                   // IC_Angle weight table: [|v| (16 = zero row)][dword of the 32-byte patch row]
#define OD_PATCH (37 * OD_PPITCH)
#define OD_PATCH_T (37 * OD_PPITCH_T)
__global__ __launch_bounds__(256) void k_orient_desc2(Geom g, const u8* const* l0, int l0pitch, const u8* pyr,
                                                      const u8* blr, const KpWork* __restrict__ work,
                                                      const int* __restrict__ nOut, KpOut* kps, u8* desc,
                                                      const int8_t* __restrict__ pattern, const u32* __restrict__ odw) {
    __shared__ u32 spat[256];                                  // (x0, y0, x1, y1) per pair, int8 each (converted on use: 1 KB instead of 4 keeps 5 workgroups per CU)
    __shared__ __attribute__((aligned(16))) u8 bpatch[16 * OD_PATCH_T];   // blurred 37-row patch of each of the 16 keypoints (40- or 48-byte rows)
    __shared__ __attribute__((aligned(8))) u32 sW[OD_WTAB];   // IC_Angle byte weights (see orbx_create)
    __shared__ unsigned long long sBal[4][16];                 // per wave: the 16 comparison ballots on their way to the lanes that store them as descriptor words
    const int tid = threadIdx.x;
    {
        for (int i = tid; i < OD_WTAB; i += 256) sW[i] = odw[i];
        spat[tid] = ((const u32*)pattern)[tid];
    }
    __syncthreads();
#if OD_XCD
    // all workgroups of a frame on ONE XCD: the patches of a frame's keypoints overlap heavily (together they cover its two
    // pyramids about 2.4 times) and both pyramids of a frame fit that XCD's L2.  Workgroup ids are dealt round-robin over the 8 XCDs:
    // the j-th id that lands on XCD x works on frame 8 (j / chunks) + x.  (Placement only matters for L2 hits, never for results.)
    int frame, chunk;
    {
        const unsigned gx = gridDim.x, b = blockIdx.x + gx * blockIdx.y, full = (gridDim.y >> 3) << 3;
        if (b < gx * full) { const unsigned j = b >> 3; frame = (int)((j / gx) * 8 + (b & 7u)); chunk = (int)(j % gx); }
        else { frame = (int)blockIdx.y; chunk = (int)blockIdx.x; }
    }
#else
    const int frame = blockIdx.y, chunk = blockIdx.x;
#endif
    const int lane = tid & 63, sub = lane >> 4, sl = lane & 15;
    const int n = nOut[frame];
    const int base = (chunk * 4 + (tid >> 6)) * 4;
    if (base >= n) return;
    const int p = base + sub;
    const bool valid = p < n;
    const KpWork w = work[(size_t)frame * g.kpCap + (valid ? p : base)];
    // ---- orientation: all 64 lanes work on one keypoint's 31 x 8 dwords at a time; the 4 x 4 loads of the whole
    // wave are issued before any arithmetic (one memory round trip instead of four)
    int m10s[4] = {0, 0, 0, 0}, m01s[4] = {0, 0, 0, 0};
    // 8-byte pieces (unaligned dwordx2 loads): 31 x 4 pieces of the unblurred rows + 37 x 5 of the blurred ones = 5 load
    // instructions per keypoint instead of 10 (same bytes; worth 1.5 %: the kernel is bound by the lines it misses in L2)
    const bool tiled = g.blurTiled != 0;                        // wave-uniform
    uint2 dq[4][2], bq[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int it = 0; it < 2; ++it) dq[k][it] = make_uint2(0, 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) bq[k][it] = make_uint2(0, 0);
        if (base + k < n) {                                     // wave-uniform
            const int level = __builtin_amdgcn_readlane((int)w.level, 16 * k);
            const int cx = __builtin_amdgcn_readlane((int)w.x, 16 * k);
            const int cy = __builtin_amdgcn_readlane((int)w.y, 16 * k);
            int sp;
            const u8* im = level_ptr(g, l0, l0pitch, pyr, frame, level, &sp);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = it * 64 + lane;
                const int r = idx >> 2, j = idx & 3;
                if (r < 31) dq[k][it] = gload64u_unaligned(im, (u32)(__mul24(cy + r - 15, sp) + cx - 15 + 8 * j));
            }
            const LevelDesc& Lk = g.lv[level];
            const u8* bl = blr + (size_t)frame * g.blrFrameBytes + Lk.boff;
            if (tiled) {
                // rows cy-18..cy+18, six 8-byte pieces from (cx-18) & ~7 (each inside one 16-byte tile row): 37 x 6 = 222 pieces
                const int xal8 = (cx - 18) & ~7;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = it * 64 + lane;
                    const int r = (idx * 171) >> 10;            // idx / 6 for idx < 256
                    const int yy = cy + r - 18, xx = xal8 + 8 * (idx - r * 6);
                    // (yy >= 1: keypoints keep 19 px from the border; written as __mul24 the compiler picks a quarter-rate v_mul_lo_u32)
                    if (r < 37) bq[k][it] = gload64u_unaligned(bl, (mad24((u32)(yy >> 3), (u32)Lk.btpr, (u32)(xx >> 4)) << 7) + (u32)(((yy & 7) << 4) + (xx & 15)));
                }
            } else {
                const int xalb = (cx - 18) & ~3;
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int idx = it * 64 + lane;
                    const int r = (idx * 205) >> 10;            // idx / 5 for idx < 192
                    const int j = idx - r * 5;
                    if (r < 37) bq[k][it] = gload64u_unaligned(bl, (u32)(__mul24(cy + r - 18, Lk.pitch) + xalb + 8 * j));
                }
            }
        }
    }
    const int ppitch = tiled ? OD_PPITCH_T : OD_PPITCH;
    {   // park the blurred patches in LDS (each wave owns 4 patches; same-wave LDS traffic needs no barrier): piece idx sits at byte 8 idx
        u8* mine = bpatch + (threadIdx.x >> 6) * 4 * OD_PATCH_T;
        const int npc = tiled ? 222 : 185;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane;
                if (idx < npc) *(uint2*)(mine + k * OD_PATCH_T + idx * 8) = bq[k][it];
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int bxoff = w.x - ((w.x - 18) & (tiled ? ~7 : ~3));   // column of the keypoint inside its patch
    int tbl[2], vrow[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 64 + lane;
        const int v = (idx >> 2) - 15;
        vrow[it] = v;
        tbl[it] = min(v < 0 ? -v : v, 16) * 8 + 2 * (idx & 3);  // the two dwords (2j, 2j+1) of the 32-byte patch row
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) {
            u32 a1 = 0, a2 = 0;
            int m01 = 0;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const uint2 w1 = *(const uint2*)&sW[tbl[it]];
                const u32 w2x = ((w1.x + 0x7F7F7F7Fu) & 0x80808080u) >> 7, w2y = ((w1.y + 0x7F7F7F7Fu) & 0x80808080u) >> 7;
                a1 = __builtin_amdgcn_udot4(dq[k][it].x, w1.x, a1, false);
                a1 = __builtin_amdgcn_udot4(dq[k][it].y, w1.y, a1, false);
                u32 s2 = __builtin_amdgcn_udot4(dq[k][it].x, w2x, 0u, false);
                s2 = __builtin_amdgcn_udot4(dq[k][it].y, w2y, s2, false);
                a2 += s2;
                m01 += __mul24(vrow[it], (int)s2);
            }
            m10s[k] = (int)a1 - 16 * (int)a2; m01s[k] = m01;      // this lane's share of keypoint k's moments
        }
    }
    // the four keypoints' moments summed over the wave so that 16-lane group g ends with keypoint g's totals: the two halves of the wave
    // swap the two keypoints they do not keep, the quarters of a half one more, then a sum inside the 16-lane row -- 3 cross-lane
    // moves and 4 DPP adds per moment instead of 4 x 6 butterflies (integer sums: any order is exact)
    int m10, m01;
    {
        const bool hi = lane >= 32, q1 = (lane & 16) != 0;
        auto fold = [&](const int (&v)[4]) {
            const int keep0 = hi ? v[2] : v[0], keep1 = hi ? v[3] : v[1];
            const int a0 = keep0 + __shfl_xor(hi ? v[0] : v[2], 32), a1_ = keep1 + __shfl_xor(hi ? v[1] : v[3], 32);
            int x = (q1 ? a1_ : a0) + __shfl_xor(q1 ? a0 : a1_, 16);
            x += __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);      // row_ror:8
            x += __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false);      // row_ror:4
            x += __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, false);      // row_ror:2
            x += __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, false);      // row_ror:1
            return x;
        };
        m10 = fold(m10s); m01 = fold(m01s);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = angle * factorPI;
    double sd, cd;
    od_sincos(ar, &sd, &cd);                                    // ar in [0, 2 pi] (fastAtan2 returns [0, 360]); checked against libm on every float
    const float a = (float)cd, b = (float)sd;
    // ---- descriptor: 16 pairs per lane, sampled from the LDS copy of the keypoint's 37-row blurred patch (the patch
    // rows were requested together with the orientation rows: one global round trip per wave instead of two)
    // Rotation and rounding on PACKED f32 (v_pk_mul_f32 / v_pk_add_f32, two floats per instruction at full rate) -- same products, same
    // sums, same rounding as the reference's scalar expressions (ORBextractor.cc:163-170):
    //   (x*b, x*a) + (y*a, y*(-b)) = (x*b + y*a, x*a - y*b)        [(-b)*y == -(y*b) and p + (-q) == p - q exactly]
    //   cvRound = rint: adding 1.5 * 2^23 leaves rint(v) in the mantissa for |v| < 2^22 (round-to-nearest-even, like v_rndne);
    // the biased integers go straight into the address: mul_i24 sees 0x400000 + row (its low 24 bits), the column carries 0x4B400000,
    // and the constant part is folded into the patch base.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 ba = {b, a}, anb = {a, -b}, magic = {12582912.0f, 12582912.0f};
    const u32 pcA = (u32)(uintptr_t)(lds_u8_t*)(bpatch + ((threadIdx.x >> 6) * 4 + sub) * OD_PATCH_T + 18 * ppitch + bxoff)
                    - (0x400000u * (u32)ppitch + 0x4B400000u);
    u8 t0[16], t1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 raw = spat[q * 16 + sl];
        const float x0 = (float)(int)(int8_t)(raw & 0xFF), y0 = (float)(int)(int8_t)((raw >> 8) & 0xFF);
        const float x1 = (float)(int)(int8_t)((raw >> 16) & 0xFF), y1 = (float)((int)raw >> 24);
        const f2 r0 = (f2{x0, x0} * ba + f2{y0, y0} * anb) + magic;        // (row, column) of the first sample, biased
        const f2 r1 = (f2{x1, x1} * ba + f2{y1, y1} * anb) + magic;
        const float r0r = r0.x, r0c = r0.y, r1r = r1.x, r1c = r1.y;      // (plain floats first: bit-casting a vector element reads element 0)
        const u32 o0 = (u32)__mul24((int)__float_as_uint(r0r), ppitch) + __float_as_uint(r0c) + pcA;   // 24-bit multiply: full rate
        const u32 o1 = (u32)__mul24((int)__float_as_uint(r1r), ppitch) + __float_as_uint(r1c) + pcA;
        t0[q] = (u8)lds_r8(o0);                                      // (groups without a keypoint sample their own, unwritten patch slot: in bounds, never stored)
        t1[q] = (u8)lds_r8(o1);
    }
    // bytes 2q, 2q+1 of a keypoint's descriptor = its group's 16 bits of comparison q's ballot: lane 0 parks the 16 ballots (scalars) in
    // LDS as they come, lane (group, sl) then picks 16-bit word `group` of ballot sl -- no per-lane shifting, masking or selecting
    unsigned long long* sb = sBal[threadIdx.x >> 6];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const unsigned long long bal = __ballot(t0[q] < t1[q]);
        if (lane == 0) sb[q] = bal;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 myword = ((const u16*)(sb + sl))[sub];
    if (valid) {
        const size_t row = (size_t)frame * g.kpCap + w.slot;
        ((u16*)(desc + row * 32))[sl] = (u16)myword;
        if (sl == 0) kps[row].angle = angle;
    }
}

// ------------------------------------------------------------------------------------------------
// k_gray: cv::cvtColor(..., COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) for 8-bit images (Tracking.cc:1264-1290): OpenCV's fixed-point
// RGB2Gray<uchar>, gray = (c0*C0 + c1*GY + c2*C2 + half) >> bits.  One thread makes 4 gray pixels (one dword store) from 12 /
// 16 source bytes fetched as dwords when the row allows it.  Pure streaming: 4 or 5 bytes of HBM traffic per pixel.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// time stamp (graph replays): HIP events recorded during stream capture stamp nothing on replay (and the runtime PyTorch
// ships rejects hipEventRecordExternal), so a captured sequence marks its span boundaries with this one-lane kernel instead:
// the constant-rate wall clock (hipDeviceAttributeWallClockRate, 100 MHz on gfx950) at the point of the stream it sits on
// ------------------------------------------------------------------------------------------------------------------
// results -> pinned host memory, written by the GPU itself (posted PCIe writes, 16 bytes per lane): a plain kernel launch on the
// copy stream.  hipMemcpyAsync does the same job through the runtime's copy path, whose calls stall the enqueueing thread for
// several milliseconds every ~10 copies on the runtime bench.py runs on; a launch does not.  A few workgroups saturate the link.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#ifdef ORBX_AB   /* A/B reference, not in the product library */
__global__ __launch_bounds__(256) void k_copy_out(v4u_t* __restrict__ dst, const v4u_t* __restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
#endif  /* ORBX_AB */

// k_fetch_one: the single-frame call's results in ONE piece -- header (n, mono, error flag, pad) + n keypoints + n descriptors, packed,
// written straight into pinned host memory (five synchronous device-to-host copies of a few bytes to a few KB each cost the caller
// ~0.1 ms; this costs one stream synchronisation).
__global__ __launch_bounds__(256) void k_fetch_one(const KpOut* __restrict__ kps, const u8* __restrict__ desc, const int* __restrict__ nOut,
                                                   const int* __restrict__ monoOut, const int* __restrict__ err, int frame, int kpCap, u32* out, int descWord) {
    const int n = nOut[frame];
    const int t = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
    if (t == 0) { out[0] = (u32)n; out[1] = (u32)monoOut[frame]; out[2] = (u32)*err; out[3] = 0; }
    const u32* ks = (const u32*)(kps + (size_t)frame * kpCap);
    const uint4* ds = (const uint4*)(desc + (size_t)frame * kpCap * 32);
    u32* ko = out + 4;
    uint4* dout = (uint4*)(out + descWord);                      // fixed, 16-byte aligned offset behind room for kpCap keypoints
    for (int i = t; i < n * 7; i += nt) ko[i] = ks[i];
    for (int i = t; i < n * 2; i += nt) dout[i] = ds[i];
}

__global__ void k_stamp(unsigned long long* p) { *p = (unsigned long long)wall_clock64(); }

__global__ __launch_bounds__(256) void k_gray(const u8* const* srcs, int w, int h, int sstride, int ch,
                                              int c0, int gy, int c2, int bits, u8* const* dsts, int dstride) {
    const int frame = blockIdx.z;
    const int y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w || y >= h) return;
    const u8* s = srcs[frame] + (size_t)y * sstride + (size_t)x0 * ch;
    u8* d = dsts[frame] + (size_t)y * dstride + x0;
    const int half = 1 << (bits - 1);
    const int n = min(4, w - x0);
    u32 px[4] = {0, 0, 0, 0};                                // low 24 bits: the three colour bytes of pixel i
    if (n == 4 && (((size_t)s) & 3) == 0) {                  // aligned fast path: whole dwords
        if (ch == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) px[i] = gload32(s + 4 * i);
        } else {
            const u32 a = gload32(s), b = gload32(s + 4), c = gload32(s + 8);
            px[0] = a; px[1] = __builtin_amdgcn_alignbyte(b, a, 3); px[2] = __builtin_amdgcn_alignbyte(c, b, 2); px[3] = c >> 8;
        }
    } else {
        for (int i = 0; i < n; ++i) px[i] = (u32)gload8(s + i * ch) | ((u32)gload8(s + i * ch + 1) << 8) | ((u32)gload8(s + i * ch + 2) << 16);
    }
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 v = (u32)(mad24(px[i] & 0xFFu, (u32)c0, mad24((px[i] >> 8) & 0xFFu, (u32)gy, mad24((px[i] >> 16) & 0xFFu, (u32)c2, (u32)half)))) >> bits;
        out |= v << (8 * i);
    }
    if (n == 4 && (((size_t)d) & 3) == 0) *(u32*)d = out;
    else for (int i = 0; i < n; ++i) d[i] = (u8)(out >> (8 * i));
}

// ------------------------------------------------------------------------------------------------
// k_remap: cv::remap(..., INTER_LINEAR, BORDER_CONSTANT 0) for 8-bit images with float maps (stereo rectification,
// Examples/Stereo/stereo_euroc.cc:168-169): OpenCV's fixed point -- cvRound(map*32), 5-bit fractions, weights
// (32-fx)(32-fy)*32 .., (sum + 2^14) >> 15.  One thread makes 4 destination pixels (one dword store); the taps are
// gathers (the map is smooth, so neighbouring lanes hit neighbouring lines).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_remap(const u8* const* srcs, int sw, int sh, int sstride, const float* __restrict__ mapx,
                                               const float* __restrict__ mapy, int dw, int dh, u8* const* dsts, int dstride) {
    const int frame = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= dw || y >= dh) return;
    const u8* s = srcs[frame];
    u8* d = dsts[frame] + (size_t)y * dstride + x0;
    const int n = min(4, dw - x0);
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < n) {
            const size_t mi = (size_t)y * dw + x0 + i;
            const int sx = __float2int_rn(mapx[mi] * 32.f), sy = __float2int_rn(mapy[mi] * 32.f);
            const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
            const bool x0ok = ix >= 0 && ix < sw, x1ok = ix + 1 >= 0 && ix + 1 < sw, y0ok = iy >= 0 && iy < sh, y1ok = iy + 1 >= 0 && iy + 1 < sh;
            const u8* p = s + (size_t)iy * sstride + ix;
            const int v00 = x0ok && y0ok ? gload8(p) : 0, v01 = x1ok && y0ok ? gload8(p + 1) : 0;
            const int v10 = x0ok && y1ok ? gload8(p + sstride) : 0, v11 = x1ok && y1ok ? gload8(p + sstride + 1) : 0;
            const int v = v00 * ((32 - fx) * (32 - fy) * 32) + v01 * (fx * (32 - fy) * 32) + v10 * ((32 - fx) * fy * 32) + v11 * (fx * fy * 32);
            out |= (u32)((v + (1 << 14)) >> 15) << (8 * i);
        }
    }
    if (n == 4 && (((size_t)d) & 3) == 0) *(u32*)d = out;
    else for (int i = 0; i < n; ++i) d[i] = (u8)(out >> (8 * i));
}

// ------------------------------------------------------------------------------------------------
// CLAHE (cv::createCLAHE(clip, Size(tx, ty))->apply, Examples/Monocular/mono_tum_vi.cc:101-109; OpenCV 4.x clahe.cpp).
// k_clahe_lut: one workgroup per (tile, frame): LDS histogram, clip + redistribution, cumulative sum -> 256-byte LUT.
// k_clahe_apply: per pixel the float blend of the four neighbouring tile LUTs, in OpenCV's operation order.
// ------------------------------------------------------------------------------------------------
struct ClaheGeom { int w, h, tilesX, tilesY, tw, th, clip; float lutScale, invTw, invTh; };

__device__ __forceinline__ int clahe_reflect(int p, int n) {      // BORDER_REFLECT_101 for the right / bottom extension (p < 2n-1)
    return p < n ? p : 2 * n - 2 - p;
}

__global__ __launch_bounds__(256) void k_clahe_lut(const u8* const* srcs, int sstride, ClaheGeom G, u8* luts) {
    __shared__ u32 hist[256];
    __shared__ u32 wsum[4];
    __shared__ u32 s_clipped;
    const int tile = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
    const int tx = tile % G.tilesX, ty = tile / G.tilesX;
    const u8* s = srcs[frame];
    hist[tid] = 0;
    if (tid == 0) s_clipped = 0;
    __syncthreads();
    const int area = G.tw * G.th;
    for (int i = tid; i < area; i += 256) {
        const int yy = i / G.tw, xx = i - yy * G.tw;
        const int y = clahe_reflect(ty * G.th + yy, G.h), x = clahe_reflect(tx * G.tw + xx, G.w);
        atomicAdd(&hist[gload8(s + (size_t)y * sstride + x)], 1u);
    }
    __syncthreads();
    int hv = (int)hist[tid];
    if (G.clip > 0) {
        if (hv > G.clip) { atomicAdd(&s_clipped, (u32)(hv - G.clip)); hv = G.clip; }
        __syncthreads();
        const int clipped = (int)s_clipped;
        const int batch = clipped / 256;
        const int residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);               // bins 0, step, 2 step, ... get one more until the residual is spent
            if (tid % step == 0 && tid / step < residual) ++hv;
        }
    }
    // inclusive prefix sum over the 256 bins
    u32 inc = (u32)hv;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 base = 0;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    const int sum = (int)(base + inc);
    const int v = __float2int_rn((float)sum * G.lutScale);
    luts[((size_t)frame * G.tilesX * G.tilesY + tile) * 256 + tid] = (u8)min(255, max(0, v));
}

__global__ __launch_bounds__(256) void k_clahe_apply(const u8* const* srcs, int sstride, ClaheGeom G, const u8* __restrict__ luts,
                                                     u8* const* dsts, int dstride) {
    const int frame = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= G.w || y >= G.h) return;
    const u8* s = srcs[frame] + (size_t)y * sstride + x0;
    u8* d = dsts[frame] + (size_t)y * dstride + x0;
    const u8* L = luts + (size_t)frame * G.tilesX * G.tilesY * 256;
    const float tyf = (float)y * G.invTh - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0); ty2 = min(ty2, G.tilesY - 1);
    const int n = min(4, G.w - x0);
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < n) {
            const float txf = (float)(x0 + i) * G.invTw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            tx1 = max(tx1, 0); tx2 = min(tx2, G.tilesX - 1);
            const int v = gload8(s + i);
            const float l11 = (float)L[(ty1 * G.tilesX + tx1) * 256 + v], l12 = (float)L[(ty1 * G.tilesX + tx2) * 256 + v];
            const float l21 = (float)L[(ty2 * G.tilesX + tx1) * 256 + v], l22 = (float)L[(ty2 * G.tilesX + tx2) * 256 + v];
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            out |= (u32)min(255, max(0, __float2int_rn(res))) << (8 * i);
        }
    }
    if (n == 4 && (((size_t)d) & 3) == 0) *(u32*)d = out;
    else for (int i = 0; i < n; ++i) d[i] = (u8)(out >> (8 * i));
}

}  // namespace orbxk
