/* orbm.h -- C ABI of the MI355X-native ORB matcher primitives (drop-in path for ORB_SLAM3::ORBmatcher
 * and the Hamming association loops of Frame).
 *
 * The reference has no FFI layer; its boundary is the C++ class ORBmatcher (include/ORBmatcher.h:35-111)
 * whose search methods walk Frame / KeyFrame / MapPoint object graphs.  The C++ facade
 * (orb-slam3_amd/facade/ORBmatcher.h) flattens those objects into the plain arrays below, runs the
 * data-parallel distance phase on the GPU and replays the order-dependent "claim" bookkeeping on the host
 * (SURVEY 8(a) M-rows, 8(b)).
 *
 * Conventions as in orbx.h: negative return = ORBM_E_*, caller-owned buffers, no CPU fallback.
 * Descriptors are rows of 32 bytes (256 bit), exactly cv::Mat(n,32,CV_8U) rows.
 */
#ifndef ORBM_H_
#define ORBM_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORBM_OK = 0, ORBM_E_INVALID = -2, ORBM_E_CAPACITY = -3, ORBM_E_HIP = -5 };
enum { ORBM_HOST = 0, ORBM_DEVICE = 1 };
enum { ORBM_TH_HIGH = 100, ORBM_TH_LOW = 50, ORBM_HISTO_LENGTH = 30 };   /* ORBmatcher.cc:36-38 */

typedef struct orbm orbm_t;       /* owns a stream + scratch on one device */
int orbm_create(orbm_t** out, int device_id);
void orbm_destroy(orbm_t*);
const char* orbm_last_error(void);
int orbm_sync(orbm_t*);
void* orbm_stream(const orbm_t*);

/* M0  ORBmatcher::DescriptorDistance (ORBmatcher.cc:2911-2931): host-side scalar, 4 x popcount64 */
int orbm_hamming(const uint8_t* a, const uint8_t* b);

/* M1  ORBmatcher::ComputeThreeMaxima on bin sizes (ORBmatcher.cc:2863-2905) */
void orbm_three_maxima(const int* bin_sizes, int L, int* ind3);

/* M16 dense brute-force 2-NN (cv::BFMatcher(NORM_HAMMING).knnMatch k=2 in Frame::ComputeStereoFishEyeMatches,
 * Frame.cc:1440-1480).  Batched over `npairs` independent (query set, train set) pairs.
 *   q, t     : [npairs][q_stride rows][32] / [npairs][t_stride rows][32] descriptors (host or device: `space`)
 *   nq, nt   : per-pair row counts, int32[npairs] in the same space
 *   idx2/dist2 : [npairs][q_stride][2] int32, same space; idx -1 / dist -1 when fewer than k train rows.
 * Ties: lower train index first (the oracle's normative order). */
int orbm_knn2_batch(orbm_t*, int space, const uint8_t* q, int q_stride, const int32_t* nq,
                    const uint8_t* t, int t_stride, const int32_t* nt, int npairs,
                    int32_t* idx2, int32_t* dist2);
/* async form (device pointers only, no sync) -- the timed body of bench.py */
int orbm_knn2_batch_async(orbm_t*, const uint8_t* q, int q_stride, const int32_t* nq,
                          const uint8_t* t, int t_stride, const int32_t* nt, int npairs, int max_nt,
                          int32_t* idx2, int32_t* dist2);
int orbm_last_timing(orbm_t*, float* ms);

#ifdef __cplusplus
}
#endif
#endif
