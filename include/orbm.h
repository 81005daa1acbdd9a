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
/* match[] values of the searches that write into an EXISTING Frame::mvpMapPoints (M4, M5 and the fisheye twin of M4):
 * >= 0 the query matched here; ORBM_NO_MATCH the slot was not touched; ORBM_MATCH_PRUNED the slot was assigned and then culled
 * by the rotation-consistency check -- the reference leaves NULL there (ORBmatcher.cc:2700-2708, 2843-2847), whatever the slot
 * held before, so a wrapper must clear it. */
enum { ORBM_NO_MATCH = -1, ORBM_MATCH_PRUNED = -2 };

typedef struct orbm orbm_t;       /* owns a stream + scratch on one device */
int orbm_create(orbm_t** out, int device_id);
void orbm_destroy(orbm_t*);
const char* orbm_last_error(void);
int orbm_sync(orbm_t*);
void* orbm_stream(const orbm_t*);
/* run the matcher's kernels on the caller's hipStream_t (e.g. orbx_stream() of the extractor whose results they read: one
 * stream, no cross-stream event waits); NULL returns to the handle's own stream */
int orbm_set_stream(orbm_t*, void* stream);

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
/* M16 with the reference's acceptance test in the kernel's epilogue: good[npairs][q_stride] (device) = 1 where the query has two
 * neighbours and `(*it)[0].distance < (*it)[1].distance * ratio` holds as Frame.cc:1465 evaluates it (float distances, double
 * product; the reference's ratio is 0.7), else 0.  idx2 / dist2 as orbm_knn2_batch_async.  Device pointers, enqueue only. */
int orbm_knn2_ratio_batch_async(orbm_t*, const uint8_t* q, int q_stride, const int32_t* nq,
                                const uint8_t* t, int t_stride, const int32_t* nt, int npairs, double ratio,
                                int32_t* idx2, int32_t* dist2, uint8_t* good);
/* async form (device pointers only, no sync) -- the timed body of bench.py */
int orbm_knn2_batch_async(orbm_t*, const uint8_t* q, int q_stride, const int32_t* nq,
                          const uint8_t* t, int t_stride, const int32_t* nt, int npairs, int max_nt,
                          int32_t* idx2, int32_t* dist2);
int orbm_last_timing(orbm_t*, float* ms);

/* ------------------------------------------------------------------------------------------------------------
 * Flattened Frame / KeyFrame views.  The reference's searches walk Frame/KeyFrame/MapPoint object graphs; the C++
 * facade copies the few fields each search reads into these plain arrays.  All pointers below are
 * HOST pointers; every search uploads them, runs the data-parallel phase (grid gather + 256-bit Hamming) on the
 * GPU and replays the reference's order-dependent bookkeeping on the host with the device-computed distances.
 * ------------------------------------------------------------------------------------------------------------ */
#define ORBM_GRID_COLS 64     /* Frame.h:37 */
#define ORBM_GRID_ROWS 48     /* Frame.h:38 */

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orbm_kp_t;   /* == cv::KeyPoint */

typedef struct {
    int32_t n;                 /* N keypoints */
    const orbm_kp_t* kps;      /* mvKeysUn (Frame.h) */
    const uint8_t* desc;       /* mDescriptors rows */
    const float* uright;       /* mvuRight or NULL */
    float min_x, min_y, inv_w, inv_h;   /* mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv */
    const int32_t* grid_start; /* [64*48+1] CSR, cell = ix*48+iy == mGrid[ix][iy] */
    const int32_t* grid_idx;   /* keypoint indices in insertion order */
} orbm_frame_t;

/* M14  Frame::AssignFeaturesToGrid + PosInGrid (Frame.cc:446-480, 883-899) on the GPU.
 * grid_start[3073], grid_idx[n] are host outputs; returns the number of keypoints placed. */
int orbm_grid_build(orbm_t*, const orbm_kp_t* kps, int n, float min_x, float min_y, float inv_w, float inv_h,
                    int32_t* grid_start, int32_t* grid_idx);

/* M14  Frame::GetFeaturesInArea (Frame.cc:784-871) for a batch of windows, with the Hamming distance of every
 * returned keypoint to the window's query descriptor.  Candidates come out in the reference's order
 * (ix, iy, insertion).  q_* arrays have nq entries; er_max < 0 disables the stereo gate
 * (`uright[i] > 0 && fabs(ur - uright[i]) > er_max` -> skipped, ORBmatcher.cc:107-117, 2569-2576).
 * out_idx/out_dist: [nq][cap]; out_cnt[nq].  Returns 0, or ORBM_E_CAPACITY if any window overflowed `cap`. */
int orbm_window_candidates(orbm_t*, const orbm_frame_t* f, int nq, const float* qx, const float* qy, const float* qr,
                           const int32_t* min_level, const int32_t* max_level, const float* q_ur, const float* q_er_max,
                           const uint8_t* qdesc, int cap, int32_t* out_cnt, int32_t* out_idx, int32_t* out_dist);

/* M4  ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (ORBmatcher.cc:2469-2711),
 * mono / rectified-stereo path.  Per last-frame feature i: valid[i] (MapPoint present, not an outlier, invzc >= 0,
 * projection inside the image: decided by the caller's camera model), projection (u,v), invzc, octave, angle,
 * MapPoint descriptor, mp_obs[i] = pMP->Observations() > 0.  cur_blocked[i2] = Cur.mvpMapPoints[i2] already holds a
 * MapPoint with observations.  match[i2] = index of the last-frame feature whose MapPoint lands on i2, or -1. */
int orbm_search_by_projection_frame(orbm_t*, const orbm_frame_t* cur, const uint8_t* cur_blocked, const float* scale_factors,
                                    int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                    const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                    float th, int forward, int backward, float mbf, int check_ori, int32_t* match);

/* M3  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) (ORBmatcher.cc:45-239), left camera */
int orbm_search_by_projection_points(orbm_t*, const orbm_frame_t* f, const uint8_t* blocked, const float* scale_factors,
                                     int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                     const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                     float th, float nnratio, int32_t* match);

/* ---- frames resident in HBM.  Tracking runs 2-4 searches on the same Frame (TrackWithMotionModel / TrackReferenceKeyFrame,
 * then SearchLocalPoints; Tracking.cc:3002-3211, 3867-3891): orbm_frame_create uploads keypoints, descriptors and mvuRight
 * ONCE (space = ORBM_HOST), or adopts device arrays without copying them (space = ORBM_DEVICE: an extractor's result block,
 * orbx_result_device -- they must stay valid while the frame lives), and builds the 64 x 48 grid (M14) on the device.  The
 * *_resident searches then send only their queries and get back, per query, the candidate count and the 8 best (distance,
 * visiting-order) candidates -- all the claim replay can look at unless every one of them is blocked, in which case the call
 * falls back to the full candidate lists by itself.  Same arguments and results as M4 / M3 above. */
typedef struct orbm_dframe orbm_dframe_t;
int orbm_frame_create(orbm_t*, int space, int n, const orbm_kp_t* kps, const uint8_t* desc, const float* uright,
                      float min_x, float min_y, float inv_w, float inv_h, orbm_dframe_t** out);
void orbm_frame_destroy(orbm_dframe_t*);
int orbm_frame_size(const orbm_dframe_t*);
int orbm_search_by_projection_frame_resident(orbm_t*, const orbm_dframe_t* cur, const uint8_t* cur_blocked, const float* scale_factors,
                                             int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                             const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                             float th, int forward, int backward, float mbf, int check_ori, int32_t* match);
int orbm_search_by_projection_points_resident(orbm_t*, const orbm_dframe_t* f, const uint8_t* blocked, const float* scale_factors,
                                              int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                              const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                              float th, float nnratio, int32_t* match);

/* M5  ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:2723-2852), relocalisation.
 * valid[i] folds the caller-side gates (MapPoint present / not bad / not already found / projection in bounds /
 * distance invariance); level[i] = PredictScale; blocked[i2] = CurrentFrame.mvpMapPoints[i2] != NULL. */
int orbm_search_by_projection_kf(orbm_t*, const orbm_frame_t* cur, const uint8_t* blocked, const float* scale_factors,
                                 int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                 const float* angle, const uint8_t* qdesc, float th, int orb_dist, int check_ori, int32_t* match);

/* M6  ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) (ORBmatcher.cc:549-679; the
 * +vpPointsKFs overload :681-797 matches identically).  valid[i] folds the caller-side Sim3 projection gates;
 * matched_in[idx] = vpMatched[idx] != NULL; match[idx] = iMP or -1.  `kf` is the KeyFrame's view (KeyFrame.h:243-250,319). */
int orbm_search_by_projection_sim3(orbm_t*, const orbm_frame_t* kf, const uint8_t* matched_in, const float* scale_factors,
                                   int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                   const uint8_t* qdesc, int th, float ratio_hamming, int32_t* match);

/* M13 search core of ORBmatcher::Fuse (ORBmatcher.cc:1823-2049: chi2_gate = 1; Sim3 variant :2051-2199: chi2_gate = 0).
 * best_idx[i] = KeyFrame feature MapPoint i fuses into, or -1; AddObservation / Replace stay with the caller. */
int orbm_fuse(orbm_t*, const orbm_frame_t* kf, const float* scale_factors, const float* inv_sigma2,
              int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const int32_t* level,
              const uint8_t* qdesc, float th, int chi2_gate, int32_t* best_idx);

/* M13 ORBmatcher::SearchBySim3 (ORBmatcher.cc:2201-2467): two independent guided searches (KeyFrame 1's MapPoints in
 * KeyFrame 2 and vice versa; projections + gates done by the caller) and the mutual-consistency check.
 * matches12[i1] = idx2 or -1; returns nFound. */
int orbm_search_by_sim3(orbm_t*, const orbm_frame_t* kf1, const orbm_frame_t* kf2, const float* sf1, const float* sf2,
                        const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* qdesc1,
                        const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* qdesc2,
                        float th, int32_t* matches12);

/* ---- fisheye stereo (Nleft != -1): left / right keypoints in separate arrays and grids (mvKeys + mGrid, mvKeysRight +
 * mGridRight); MapPoint slots [0,Nleft) and [Nleft, Nleft+Nright) are reported as match_l / match_r ---- */
/* M4 with the right-camera block (ORBmatcher.cc:2615-2680); (ur, vr) = projection into the right camera */
int orbm_search_by_projection_frame_fisheye(orbm_t*, const orbm_frame_t* cur_l, const orbm_frame_t* cur_r,
                                            const uint8_t* blocked_l, const uint8_t* blocked_r, const float* scale_factors,
                                            int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const float* vr,
                                            const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                            float th, int forward, int backward, int check_ori, int32_t* match_l, int32_t* match_r);
/* M3 with the right-camera block (ORBmatcher.cc:170-236) and the mvLeftToRightMatch / mvRightToLeftMatch cross
 * assignments (:152-157, :222-226); l2r[nL], r2l[nR] hold -1 or the partner index */
int orbm_search_by_projection_points_fisheye(orbm_t*, const orbm_frame_t* f_l, const orbm_frame_t* f_r,
                                             const uint8_t* blocked_l, const uint8_t* blocked_r,
                                             const int32_t* l2r, const int32_t* r2l, const float* scale_factors,
                                             int nq, const uint8_t* in_view, const float* px, const float* py, const float* view_cos, const int32_t* level,
                                             const uint8_t* in_view_r, const float* pxr, const float* pyr, const float* view_cos_r, const int32_t* level_r,
                                             const uint8_t* qdesc, const uint8_t* mp_obs, float th, float nnratio, int32_t* match_l, int32_t* match_r);
/* M7 with F.Nleft != -1 (ORBmatcher.cc:405-426, 471-500): frame features [0,nleft) are left, the rest right */
int orbm_search_by_bow_fisheye(orbm_t*, int nkf, const orbm_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                               int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                               int nf, int nleft, const orbm_kp_t* kps_f, const uint8_t* desc_f,
                               int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                               float nnratio, int check_ori, int32_t* f_match);

/* M9  ORBmatcher::SearchForInitialization (ORBmatcher.cc:799-943); prev_matched_xy is updated in place */
int orbm_search_for_initialization(orbm_t*, const orbm_frame_t* f1, const orbm_frame_t* f2, float* prev_matched_xy,
                                   int window, float nnratio, int check_ori, int32_t* matches12);

/* M10 ORBmatcher::SearchForTriangulation_ (ORBmatcher.cc:1388-1629), pinhole cameras.  FeatureVectors are CSR:
 * `nodes` ascending, start[nn+1], idx[].  F12: row-major 3x3 (what Pinhole::epipolarConstrain_ builds,
 * Pinhole.cpp:273-280); (epx,epy) the epipole in image 2 (ORBmatcher.cc:1399-1400). */
int orbm_search_for_triangulation(orbm_t*, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                  int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                  int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                  int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                  const float* F12, float epx, float epy, const float* scale_factors2, const float* level_sigma2_2,
                                  int only_stereo, int coarse, int check_ori, int32_t* matches12);

/* M7  ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:314-547), Nleft == -1.  f_match[iF] = KF index or -1 */
int orbm_search_by_bow(orbm_t*, int nkf, const orbm_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                       int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                       int nf, const orbm_kp_t* kps_f, const uint8_t* desc_f,
                       int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                       float nnratio, int check_ori, int32_t* f_match);

/* M8  ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:955-1105); matches12[idx1] = idx2 or -1 */
int orbm_search_by_bow_kf(orbm_t*, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* good1,
                          int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                          int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* good2,
                          int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                          float nnratio, int check_ori, int32_t* matches12);

/* M11 ORBmatcher::SearchForTriangulation, cv::Mat F12 overload (ORBmatcher.cc:1107-1386): as M10 but vbMatched2 is kept
 * (set :1319, cleared by the orientation cull :1366) and the histogram factor is 30/360 (:1166). */
int orbm_search_for_triangulation_legacy(orbm_t*, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                  int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                  int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                  int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                  const float* F12, float epx, float epy, const float* scale_factors2, const float* level_sigma2_2,
                                  int only_stereo, int coarse, int check_ori, int32_t* matches12);

/* M10 with a second camera (pKF1->mpCamera2, ORBmatcher.cc:1413-1426, 1526-1557) and M12
 * (ORBmatcher::SearchForTriangulation(+vMatchedPoints), ORBmatcher.cc:1632-1821).  Both are the M10 bucket search with the
 * geometric gate supplied by the camera model (GeometricCamera::epipolarConstrain_ :1552, ::matchAndtriangulate :1729 —
 * KannalaBrandt8 unprojects, triangulates with cv::SVD and tests the reprojection, KannalaBrandt8.cpp:356-361, 363-472,
 * 559-700; Pinhole::matchAndtriangulate is `return false`, Pinhole.h:88-91), no epipole
 * gate (:1517 needs !mpCamera2; M12 has none) and no stereo flags (bStereo is false with mpCamera2, :1462; M12 ignores
 * bOnlyStereo).  The gate stays with the caller: `gate(user, idx1, idx2)` is called exactly where the reference calls the
 * camera model — after the MapPoint and `dist <= TH_LOW && dist <= bestDist` tests, in bucket order — and a nonzero
 * return accepts the pair, so a callback that records its last accepted x3D per idx1 reproduces M12's vMatchedPoints.
 * kps1/kps2 are the N-long keypoint arrays the reference indexes (mvKeysUn, or mvKeys followed by mvKeysRight when
 * NLeft != -1, :1467-1469); histogram factor 1/30 (:1441, :1672).  Descriptor distances are computed on the GPU. */
typedef int (*orbm_pair_gate_fn)(void* user, int idx1, int idx2);
int orbm_search_for_triangulation_gated(orbm_t*, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1,
                                        int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                        int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2,
                                        int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                        orbm_pair_gate_fn gate, void* user, int check_ori, int32_t* matches12);

/* ---- batched, DEVICE-resident forms: frame-to-frame tracking chained behind orbx_extract_batch_async with no host round
 * trip (kps/desc/counts are the extractor's result block, orbx_result_device; all pointers are device pointers).
 * orbm_grid_build_batch_async: M14 for every frame of the block; grid_start [nframes][3073], grid_idx [nframes][cap].
 * orbm_track_window_batch_async: for pair p the keypoints of frame q_first+p search frame t_first+p inside the mono
 * SearchByProjection window (centre (x+dx, y+dy), radius th*scale[octave], levels octave-1..octave+1,
 * ORBmatcher.cc:2543-2549): first-minimum best index/distance and runner-up distance per keypoint, [npairs][cap].
 * This is the claim-free, data-parallel part; the final matches of the batch: orbm_search_by_projection_batch_async. */
/* orbm_search_by_projection_batch_async: M4 SearchByProjection(CurrentFrame, LastFrame, th, bMono = true) END TO END on the device
 * for `npairs` frame pairs of one result block (ORBmatcher.cc:2469-2711, the monocular branch: no mvuRight gate, window levels
 * octave-1..octave+1).  Pair p: the keypoints of frame q_first+p play LastFrame's MapPoints in index order -- every one valid,
 * projected to (x+dx, y+dy), carrying its own descriptor, octave and angle -- and search frame t_first+p through its grid.
 * t_blocked [frames][cap] (device, indexed by FRAME id; NULL = none): slots of the searched frame that already hold a MapPoint
 * with observations (:2565-2567).  q_obs [frames][cap] (NULL = all 1): pMP->Observations() > 0 of the query's MapPoint, i.e.
 * whether its assignment blocks later queries.  The claim sequence, `bestDist <= TH_HIGH`, the rotation histogram and the
 * ComputeThreeMaxima cull (:2595-2605, 2690-2708) run in query order on the device.  Outputs (device): match [npairs][cap] =
 * query index assigned to that slot of the searched frame, ORBM_NO_MATCH or ORBM_MATCH_PRUNED -- the same row
 * orbm_search_by_projection_frame returns for the pair --, nmatches [npairs] = its return value.  Enqueue-only (capturable;
 * the first call with a larger batch grows the handle's scratch and must run outside a capture).  cap <= ~16 000 keypoint slots per
 * frame (the claim replay keeps 4 B per slot in 64 KB of LDS); ORBM_E_INVALID above. */
int orbm_search_by_projection_batch_async(orbm_t*, const orbm_kp_t* kps, const uint8_t* desc, const int32_t* counts, int cap,
                                          const int32_t* grid_start, const int32_t* grid_idx,
                                          float min_x, float min_y, float inv_w, float inv_h,
                                          int q_first, int t_first, int npairs, float th, const float* scale_factors_host, int nlevels,
                                          float dx, float dy, const uint8_t* t_blocked, const uint8_t* q_obs, int check_orientation,
                                          int32_t* match, int32_t* nmatches);
int orbm_grid_build_batch_async(orbm_t*, const orbm_kp_t* kps, const int32_t* counts, int nframes, int cap,
                                float min_x, float min_y, float inv_w, float inv_h, int32_t* grid_start, int32_t* grid_idx);
int orbm_track_window_batch_async(orbm_t*, const orbm_kp_t* kps, const uint8_t* desc, const int32_t* counts, int cap,
                                  const int32_t* grid_start, const int32_t* grid_idx,
                                  float min_x, float min_y, float inv_w, float inv_h,
                                  int q_first, int t_first, int npairs, float th, const float* scale_factors_host, int nlevels,
                                  float dx, float dy, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);

/* ---- batched, DEVICE-resident stereo step (config C3: EuRoC stereo).  All pointers are device pointers; enqueue only.
 * orbm_stereo_batch_async: M15 Frame::ComputeStereoMatches (Frame.cc:1027-1276) for `npairs` stereo pairs of ONE extractor
 * batch: pair p = frames (first_l + p, first_r + p); kps / desc / counts = that extractor's result block (orbx_result_device),
 * cap = orbx_max_keypoints.  The row-band candidates, Hamming, SAD slide and parabola run per left keypoint on the
 * device-resident pyramids; the median cut (:1261-1275) is a second kernel.  Outputs [npairs][cap]: uright (mvuRight), depth
 * (mvDepth), sad (scratch: best SAD or -1); kept[npairs] = stereo points that survive the cut.
 * orbm_bow_nodes_batch_async: Frame::ComputeBoW's FeatureVector bucket of every descriptor row (8(f).1 tree descent, node at
 * `levelsup` levels above the leaves) for nrows rows of a result block: node_id[nrows].
 * orbm_triangulation_batch_async: M10 SearchForTriangulation_ (ORBmatcher.cc:1388-1629; pinhole, no MapPoints attached, no
 * orientation check: how LocalMapping calls it, LocalMapping.cc:514-516,592) for `npairs` KeyFrame pairs: KeyFrame 1 of pair p
 * = row p of the *1 arrays ([npairs][cap] slices of a result block + its node ids and mvuRight), KeyFrame 2 likewise.  A
 * bucket = equal node id, walked in ascending index order as the FeatureVector is.  matches12 [npairs][cap] = idx2 or -1,
 * nmatches[npairs]. */
int orbm_stereo_batch_async(orbm_t*, void* extractor, int first_l, int first_r, int npairs, const orbm_kp_t* kps, const uint8_t* desc,
                            const int32_t* counts, int cap, float mb, float mbf, float* uright, float* depth, int32_t* sad, int32_t* kept);
int orbm_bow_nodes_batch_async(orbm_t*, const struct orbm_vocab* vocab, const uint8_t* desc, int nrows, int levelsup, int32_t* node_id);
int orbm_triangulation_batch_async(orbm_t*, int npairs, int cap,
                                   const orbm_kp_t* kps1, const uint8_t* desc1, const int32_t* counts1, const int32_t* node1, const float* uright1,
                                   const orbm_kp_t* kps2, const uint8_t* desc2, const int32_t* counts2, const int32_t* node2, const float* uright2,
                                   const float* F12, float epx, float epy, const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                   int only_stereo, int coarse, int32_t* matches12, int32_t* nmatches);

/* ---- DBoW2 vocabulary transform (Frame::ComputeBoW, Frame.cc:905-918;
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1125-1262, FORB::distance FORB.cpp:81-101) ----
 * The tree lives in HBM; orbm_bow_transform descends it for n descriptors (host pointers) and returns per feature the
 * word id, the node id `levelsup` levels above the leaves (the SearchByBoW bucket) and the word weight (0 = stopped).
 * orbm_bow_vectors assembles BowVector (TF-IDF, L1-normalised: the ORBvoc configuration) and FeatureVector (CSR) on the
 * host exactly as the std::map based classes do. */
typedef struct orbm_vocab orbm_vocab_t;
int orbm_vocab_load_text(orbm_t*, orbm_vocab_t** out, const char* path);     /* loadFromTextFile, :1338-1440 */
int orbm_vocab_create(orbm_t*, orbm_vocab_t** out, int k, int L, int nnodes, const int32_t* parent, const uint8_t* is_leaf,
                      const uint8_t* desc, const double* weight);            /* node 0 = root; ids in file order */
void orbm_vocab_destroy(orbm_vocab_t*);
int orbm_vocab_info(const orbm_vocab_t*, int* k, int* L, int* nnodes, int* nwords);
int orbm_bow_transform(orbm_t*, const orbm_vocab_t*, const uint8_t* desc, int n, int levelsup,
                       int32_t* word_id, int32_t* node_id, double* weight);
int orbm_bow_vectors(int n, const int32_t* word_id, const int32_t* node_id, const double* weight,
                     int32_t* bow_ids, double* bow_vals, int* nbow,
                     int32_t* fv_nodes, int32_t* fv_start, int32_t* fv_idx, int* nfv);

/* M15 Frame::ComputeStereoMatches (Frame.cc:1027-1276).  left/right are orbx_t* extractor handles (include/orbx.h)
 * on the same device whose LAST call produced the two keypoint sets: their device-resident pyramids supply the
 * 11x11 SAD windows (mvImagePyramid, include/ORBextractor.h:83).  frame_l/frame_r select the batch slot.
 * uright/depth: host outputs [nl] (mvuRight, mvDepth).  Returns the number of stereo points kept. */
int orbm_stereo_matches(orbm_t*, void* left_extractor, int frame_l, void* right_extractor, int frame_r,
                        int nl, const orbm_kp_t* kl, const uint8_t* dl, int nr, const orbm_kp_t* kr, const uint8_t* dr,
                        float mb, float mbf, float* uright, float* depth);

/* SURVEY 8(f).2  Frame::UndistortKeyPoints (Frame.cc:924-970): cv::undistortPoints(points, K, D, R = I, P = newK) on the
 * coordinates of n keypoints (other fields copied).  k / newk = (fx, fy, cx, cy); dist = (k1, k2, p1, p2[, k3 ...]),
 * ndist <= 14.  dist[0] == 0 copies the input (Frame.cc:928-932).  space = ORBM_HOST | ORBM_DEVICE for kps and out. */
int orbm_undistort_keypoints(orbm_t*, int space, const orbm_kp_t* kps, int n, const float* k, const float* dist, int ndist,
                             const float* newk, orbm_kp_t* out);
/* Frame::ComputeImageBounds (Frame.cc:977-1021): bounds[4] = (mnMinX, mnMaxX, mnMinY, mnMaxY) (host output). */
int orbm_image_bounds(orbm_t*, int cols, int rows, const float* k, const float* dist, int ndist, const float* newk, float* bounds);

/* SURVEY 8(f).3  Frame::isInFrustum (Nleft == -1; Frame.cc:603-671) + MapPoint::PredictScale (MapPoint.cc:725-740) +
 * Pinhole::project for n map points: the producer of the arrays orbm_search_by_projection_points consumes.
 * pw / normal: [n][3] world position and mean viewing direction; min_dist / max_dist: mfMinDistance / mfMaxDistance;
 * rcw[9] row-major, tcw[3], ow[3]; k = (fx, fy, cx, cy); bounds = (minX, maxX, minY, maxY); bf = mbf.
 * Outputs mirror the MapPoint members: in_view (mbTrackInView), proj_x / proj_y (mTrackProjX/Y, -1 unless the point
 * passed the image-bounds test), proj_xr, depth (mTrackDepth), level (mnTrackScaleLevel), view_cos -- the last four are
 * written only where in_view.  All arrays in `space`.  Returns the number of points in view (host space) or 0. */
int orbm_is_in_frustum(orbm_t*, int space, int n, const float* pw, const float* normal, const float* min_dist, const float* max_dist,
                       const float* rcw, const float* tcw, const float* ow, const float* k, const float* bounds,
                       float bf, float viewing_cos_limit, float log_scale_factor, int n_scale_levels,
                       uint8_t* in_view, float* proj_x, float* proj_y, float* proj_xr, float* depth, int32_t* level, float* view_cos);

#ifdef __cplusplus
}
#endif
#endif
