/* orbref.h -- C interface of the CPU ORACLE ("orbref").
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free CPU restatement of the
 * reference's ORB front-end (src/ORBextractor.cc, src/ORBmatcher.cc, parts of
 * src/Frame.cc) plus the OpenCV primitives it calls (FAST, resize,
 * GaussianBlur, fastAtan2 -- OpenCV is NOT vendored in the reference and is
 * absent from this image; its published 4.x algorithms are restated, see
 * DESIGN.md "Oracle").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (orb-slam3_amd/) never does.
 *
 * PARITY UNPINNED: the reference ships no golden vectors / unit tests for this
 * path and cannot be compiled here (needs OpenCV, Eigen, Boost).  The oracle is
 * pinned only by hand-derivable known-answer tests (tests/test_oracle_kat.py).
 */
#ifndef ORBREF_H_
#define ORBREF_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbref orbref_t;
/* == cv::KeyPoint memory layout (28 bytes) */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orbref_kp_t;

/* ORBextractor::ORBextractor  (src/ORBextractor.cc:468-571) */
orbref_t* orbref_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void orbref_destroy(orbref_t*);
/* ORBextractor::operator()  (src/ORBextractor.cc:1534-1659).
 * returns n >= 0 keypoints written, -1 on empty image, -2 capacity too small, -3 image too small */
int orbref_extract(orbref_t*, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                   orbref_kp_t* kps, uint8_t* desc, int cap, int* mono_index);

/* tables: sf/inv_sf/sig2/inv_sig2 [nlevels] float, nfeat [nlevels] int, umax[16] int */
void orbref_tables(const orbref_t*, float* sf, float* inv_sf, float* sig2, float* inv_sig2,
                   int* nfeat_per_level, int* umax);
/* stage introspection (state of the last orbref_extract call) */
int orbref_level_size(const orbref_t*, int level, int* w, int* h);
int orbref_level_image(const orbref_t*, int level, uint8_t* dst, int dst_stride);
int orbref_level_blurred(const orbref_t*, int level, uint8_t* dst, int dst_stride);
/* candidates handed to DistributeOctTree, in order: (x,y,response) ints, coords relative to (16,16) */
int orbref_level_candidates(const orbref_t*, int level, int32_t* xyr, int cap);
/* per-level keypoints after distribution + orientation, level coordinates (x,y,response as int; angle) */
int orbref_level_keypoints(const orbref_t*, int level, int32_t* xyr, float* angle, int cap);
/* accumulated stage time [ms]: pyramid, fast, quadtree, angle, blur, descriptor */
void orbref_stage_ms(const orbref_t*, double* out6);
void orbref_stage_reset(orbref_t*);

/* stand-alone primitives (KATs) */
float orbref_fast_atan2(float y, float x);
/* cv::FAST(img, thr, nms=true): writes (x,y,score) triples, returns count */
int orbref_fast(const uint8_t* img, int w, int h, int stride, int threshold, int32_t* xys, int cap);
int orbref_fast_score(const uint8_t* img, int stride, int x, int y); /* max(A,B)-1 at pixel */
void orbref_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void orbref_gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* DistributeOctTree on integer candidates; returns n selected, writes candidate indices in output order */
int orbref_distribute(const int32_t* xyr, int n, int minX, int maxX, int minY, int maxY, int N, int32_t* out_idx, int cap);
const int8_t* orbref_pattern(void);

/* ---- matcher side (src/ORBmatcher.cc, src/Frame.cc) ---- */
int orbref_hamming(const uint8_t* a, const uint8_t* b);                 /* ORBmatcher.cc:2911-2931 */
void orbref_three_maxima(const int* hist_counts, int L, int* ind3);      /* ORBmatcher.cc:2863-2905 */
/* brute-force 2-NN (Frame.cc:1440-1480 / cv::BFMatcher knnMatch k=2); ties -> lower train index */
void orbref_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2);

#ifdef __cplusplus
}
#endif
#endif
