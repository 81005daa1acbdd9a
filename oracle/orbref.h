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


/* ---- flattened Frame view used by the matcher restatements (Frame.h:37-38 grid 64x48, Frame.cc:446-480) ---- */
#define ORBREF_GRID_COLS 64
#define ORBREF_GRID_ROWS 48
typedef struct {
    int32_t n;                 /* N keypoints */
    const orbref_kp_t* kps;    /* mvKeysUn (== mvKeys when undistortion is the identity) */
    const uint8_t* desc;       /* mDescriptors, n x 32 */
    const float* uright;       /* mvuRight or NULL */
    float min_x, min_y, inv_w, inv_h;   /* mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv */
    const int32_t* grid_start; /* [64*48+1] CSR over cells, cell = ix*48+iy (mGrid[ix][iy]) */
    const int32_t* grid_idx;   /* keypoint indices, insertion order inside a cell */
} orbref_frame_t;

/* Frame::AssignFeaturesToGrid + PosInGrid (Frame.cc:446-480, 883-899); returns #keypoints placed */
int orbref_grid_build(const orbref_kp_t* kps, int n, float min_x, float min_y, float inv_w, float inv_h,
                      int32_t* grid_start, int32_t* grid_idx);
/* Frame::GetFeaturesInArea (Frame.cc:784-871) */
int orbref_features_in_area(const orbref_frame_t* f, float x, float y, float r, int min_level, int max_level,
                            int32_t* out, int cap);

/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (ORBmatcher.cc:2469-2711), left/mono path.
 * Per last-frame feature i (nq of them): valid[i] = has MapPoint && !outlier && invzc>=0 && projection inside
 * the image bounds (all decided by the caller's camera model), (u,v) the projection, invzc, octave, angle, the
 * MapPoint descriptor and mp_obs[i] = (pMP->Observations()>0).  cur_blocked[i2] = CurrentFrame.mvpMapPoints[i2]
 * already holds a MapPoint with observations.  Output match[i2] = i or -1. */
int orbref_search_by_projection_frame(const orbref_frame_t* cur, const uint8_t* cur_blocked, const float* scale_factors,
                                      int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                      const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                      float th, int forward, int backward, float mbf, int check_ori, int32_t* match);
/* ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) (ORBmatcher.cc:45-239), left path.
 * Per MapPoint: in_view, proj x/y/xr, view cos, predicted level, descriptor, obs flag. */
int orbref_search_by_projection_points(const orbref_frame_t* f, const uint8_t* blocked, const float* scale_factors,
                                       int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                       const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                       float th, float nnratio, int32_t* match);
/* ORBmatcher::SearchForInitialization (ORBmatcher.cc:799-943) */
int orbref_search_for_initialization(const orbref_frame_t* f1, const orbref_frame_t* f2, float* prev_matched_xy,
                                     int window, float nnratio, int check_ori, int32_t* matches12);
/* ORBmatcher::SearchForTriangulation_ (ORBmatcher.cc:1388-1629), pinhole / no second camera.
 * FeatureVectors flattened as CSR: nodes sorted ascending, per node a list of feature indices.
 * F12 = the 3x3 fundamental matrix epipolarConstrain_ builds (Pinhole.cpp:273-296), row-major. */
int orbref_search_for_triangulation(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                    int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                    int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                    int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                    const float* F12, float epx, float epy, const float* scale_factors2, const float* level_sigma2_2,
                                    int only_stereo, int coarse, int check_ori, int32_t* matches12);
/* SearchForTriangulation_ with pKF1->mpCamera2 (ORBmatcher.cc:1413-1426, 1526-1557) and SearchForTriangulation(+vMatchedPoints)
 * (ORBmatcher.cc:1632-1821): the bucket search with the camera model's gate (epipolarConstrain_ :1552 / matchAndtriangulate
 * :1729) as a callback, called lazily at the reference's call site; no epipole gate, no stereo flags; factor 1/30. */
typedef int (*orbref_pair_gate_fn)(void* user, int idx1, int idx2);
int orbref_search_for_triangulation_gated(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1,
                                          int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                          int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2,
                                          int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                          orbref_pair_gate_fn gate, void* user, int check_ori, int32_t* matches12);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:314-547), Nleft == -1 path.
 * kf_good[i] = KF feature i has a MapPoint that is not bad.  Output f_match[iF] = KF feature index or -1. */
int orbref_search_by_bow(int nkf, const orbref_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                         int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                         int nf, const orbref_kp_t* kps_f, const uint8_t* desc_f,
                         int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                         float nnratio, int check_ori, int32_t* f_match);
/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:2723-2852).
 * valid[i] folds: MapPoint present, not bad, not already found, projection in bounds, distance gates; level[i] is
 * PredictScale's result.  blocked[i2] = CurrentFrame.mvpMapPoints[i2] != NULL. */
int orbref_search_by_projection_kf(const orbref_frame_t* cur, const uint8_t* blocked, const float* scale_factors,
                                   int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                   const float* angle, const uint8_t* qdesc, float th, int orb_dist, int check_ori, int32_t* match);
/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:955-1105); matches12[idx1] = idx2 or -1 */
int orbref_search_by_bow_kf(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* good1,
                            int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                            int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* good2,
                            int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                            float nnratio, int check_ori, int32_t* matches12);
/* ORBmatcher::SearchForTriangulation (cv::Mat F12 overload, ORBmatcher.cc:1107-1386): as the _ variant but
 * vbMatched2 is maintained (set :1319, cleared on the orientation cull :1366) and factor = 30/360. */
int orbref_search_for_triangulation_legacy(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                    int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                    int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                    int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                    const float* F12, float epx, float epy, const float* scale_factors2, const float* level_sigma2_2,
                                    int only_stereo, int coarse, int check_ori, int32_t* matches12);
/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) (ORBmatcher.cc:549-679; the
 * +vpPointsKFs overload :681-797 matches identically).  valid[i] folds the caller-side gates (bad / already found /
 * depth / IsInImage / distance / normal); matched_in[idx] = vpMatched[idx] != NULL; match[idx] = iMP or -1. */
int orbref_search_by_projection_sim3(const orbref_frame_t* kf, const uint8_t* matched_in, const float* scale_factors,
                                     int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                     const uint8_t* qdesc, int th, float ratio_hamming, int32_t* match);
/* search core of ORBmatcher::Fuse (ORBmatcher.cc:1823-2049 with chi2_gate=1, Sim3 variant :2051-2199 with 0):
 * best_idx[i] = KeyFrame feature the MapPoint i fuses into, or -1.  The map mutation stays with the caller. */
int orbref_fuse(const orbref_frame_t* kf, const float* scale_factors, const float* inv_sigma2,
                int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const int32_t* level,
                const uint8_t* qdesc, float th, int chi2_gate, int32_t* best_idx);
/* ORBmatcher::SearchBySim3 (ORBmatcher.cc:2201-2467): valid1/u1/v1/level1/qdesc1 describe KeyFrame 1's MapPoints
 * projected into KeyFrame 2 (caller-side Sim3 + gates), valid2/... the reverse; matches12[i1] = idx2 for mutually
 * consistent pairs, else -1.  Returns nFound. */
int orbref_search_by_sim3(const orbref_frame_t* kf1, const orbref_frame_t* kf2, const float* sf1, const float* sf2,
                          const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* qdesc1,
                          const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* qdesc2,
                          float th, int32_t* matches12);
/* ---- DBoW2 (vendored: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h, FORB.cpp) ----
 * loadFromTextFile (:1338-1440) + transform (:1125-1262): per feature the word id, the node id `levelsup` levels above
 * the leaves and the word weight (0 = stopped).  Returns 0 or -1 (bad file). */
typedef struct orbref_vocab orbref_vocab_t;
orbref_vocab_t* orbref_vocab_load_text(const char* path);
orbref_vocab_t* orbref_vocab_create(int k, int L, int nnodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight);
void orbref_vocab_destroy(orbref_vocab_t*);
int orbref_vocab_info(const orbref_vocab_t*, int* k, int* L, int* nnodes, int* nwords);
int orbref_bow_transform(const orbref_vocab_t*, const uint8_t* desc, int n, int levelsup,
                         int32_t* word_id, int32_t* node_id, double* weight);
/* BowVector (TF-IDF weighting, L1 normalisation: the ORBvoc configuration) + FeatureVector from the per-feature
 * results, as Frame::ComputeBoW builds them (Frame.cc:905-918).  bow_ids/bow_vals: capacity n; fv_*: CSR. */
int orbref_bow_vectors(int n, const int32_t* word_id, const int32_t* node_id, const double* weight,
                       int32_t* bow_ids, double* bow_vals, int* nbow,
                       int32_t* fv_nodes, int32_t* fv_start, int32_t* fv_idx, int* nfv);
/* ---- fisheye stereo (Nleft != -1) twins: left and right keypoints live in separate arrays / grids
 * (mvKeys + mGrid, mvKeysRight + mGridRight), MapPoint slots are [0,Nleft) and [Nleft, Nleft+Nright) ---- */
/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) with the right-camera block (ORBmatcher.cc:2615-2680).
 * (uR, vR) = projection into the right camera.  match_l[nL] / match_r[nR] = last-frame feature index or -1. */
int orbref_search_by_projection_frame_fisheye(const orbref_frame_t* cur_l, const orbref_frame_t* cur_r,
                                              const uint8_t* blocked_l, const uint8_t* blocked_r, const float* scale_factors,
                                              int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const float* vr,
                                              const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                              float th, int forward, int backward, int check_ori, int32_t* match_l, int32_t* match_r);
/* ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, ...) with the right-camera block (ORBmatcher.cc:170-236) and
 * the mvLeftToRightMatch / mvRightToLeftMatch cross assignments (:152-157, :222-226). */
int orbref_search_by_projection_points_fisheye(const orbref_frame_t* f_l, const orbref_frame_t* f_r,
                                               const uint8_t* blocked_l, const uint8_t* blocked_r,
                                               const int32_t* l2r, const int32_t* r2l, const float* scale_factors,
                                               int nq, const uint8_t* in_view, const float* px, const float* py, const float* view_cos, const int32_t* level,
                                               const uint8_t* in_view_r, const float* pxr, const float* pyr, const float* view_cos_r, const int32_t* level_r,
                                               const uint8_t* qdesc, const uint8_t* mp_obs, float th, float nnratio, int32_t* match_l, int32_t* match_r);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&) with F.Nleft != -1 (ORBmatcher.cc:405-426, 471-500): frame features
 * [0,nleft) are left, the rest right; the right twin has its ratio test disabled (`|| true`). */
int orbref_search_by_bow_fisheye(int nkf, const orbref_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                         int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                         int nf, int nleft, const orbref_kp_t* kps_f, const uint8_t* desc_f,
                         int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                         float nnratio, int check_ori, int32_t* f_match);
/* Frame::ComputeStereoMatches (Frame.cc:1027-1276).  Pyramids are those of the two extractors' last call. */
int orbref_stereo_matches(const orbref_t* left, const orbref_t* right,
                          int nl, const orbref_kp_t* kl, const uint8_t* dl, int nr, const orbref_kp_t* kr, const uint8_t* dr,
                          float mb, float mbf, float* uright, float* depth);

/* Frame::UndistortKeyPoints (Frame.cc:924-970): cv::undistortPoints(pts, K, D, R = I, P = newK) on keypoint
 * coordinates.  OpenCV's routine is restated (cvUndistortPointsInternal: double arithmetic, exactly 5 fixed-point
 * iterations -- TermCriteria(MAX_ITER, 5, 0.01) -- with the `icdist < 0` bail-out; rational/thin-prism terms are zero for
 * the 4/5-coefficient models ORB-SLAM3 configures).  k = (fx, fy, cx, cy); dist = (k1, k2, p1, p2[, k3]).  dist[0] == 0
 * copies the input (Frame.cc:928-932).  Other keypoint fields are copied.  OpenCV version unpinned -> parity unpinned. */
int orbref_undistort_keypoints(const orbref_kp_t* kps, int n, const float* k, const float* dist, int ndist, const float* newk,
                               orbref_kp_t* out);
/* Frame::ComputeImageBounds (Frame.cc:977-1021): bounds = (minX, maxX, minY, maxY) from the four undistorted corners. */
int orbref_image_bounds(int cols, int rows, const float* k, const float* dist, int ndist, const float* newk, float* bounds);
/* Frame::isInFrustum, Nleft == -1 branch (Frame.cc:603-671) + MapPoint::PredictScale (MapPoint.cc:725-740) +
 * Pinhole::project (Pinhole.cpp:44-62) for n map points.  rcw[9] row-major, tcw[3], ow[3] = camera centre,
 * k = (fx, fy, cx, cy), bounds = (minX, maxX, minY, maxY); min_dist / max_dist are mfMinDistance / mfMaxDistance (the
 * 0.8 / 1.2 invariance factors are applied here, MapPoint.cc:668-681).  Outputs as the MapPoint members:
 * in_view (mbTrackInView), proj_x/proj_y (-1 unless the point passed the image-bounds test, Frame.cc:607-608,637-638),
 * proj_xr, depth (mTrackDepth = |Pc|), level (mnTrackScaleLevel), view_cos; the last four only where in_view. Returns
 * the number of points in view. */
int orbref_is_in_frustum(int n, const float* pw, const float* normal, const float* min_dist, const float* max_dist,
                         const float* rcw, const float* tcw, const float* ow, const float* k, const float* bounds,
                         float bf, float viewing_cos_limit, float log_scale_factor, int n_scale_levels,
                         uint8_t* in_view, float* proj_x, float* proj_y, float* proj_xr, float* depth, int32_t* level, float* view_cos);

/* SURVEY 8(f).4 image ingest: cv::cvtColor(src, gray, COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) on 8-bit images
 * (Tracking.cc:1264-1290, 1339-1348, 1393-1402).  OpenCV's fixed-point RGB2Gray<uchar> restated:
 *   gray = (R*RY + G*GY + B*BY + (1 << (bits-1))) >> bits
 * bits = 14: (RY, GY, BY) = (4899, 9617, 1868) -- OpenCV 3.x `yuv_shift`; bits = 15: (9798, 19235, 3735) -- OpenCV 4.x.
 * Version (and IPP use) unpinned -> parity unpinned.  channels = 3 | 4, blue_first = 1 for BGR / BGRA. */
int orbref_gray_from_color(const uint8_t* src, int w, int h, int src_stride, int channels, int blue_first, int coef_bits,
                           uint8_t* dst, int dst_stride);

/* SURVEY 8(f).4 stereo rectification: cv::remap(src, dst, map1, map2, INTER_LINEAR) for CV_8UC1 with CV_32FC1 maps
 * (Examples/Stereo/stereo_euroc.cc:168-169), default BORDER_CONSTANT 0.  OpenCV's fixed-point path restated: coordinates
 * cvRound(map * 32) -> integer part >> 5, 5-bit fractions (INTER_BITS); weights (32-fx)(32-fy)*32 ... fx*fy*32 (they sum to
 * 1 << 15 exactly, so the table's sum correction never triggers for INTER_LINEAR); dst = (sum + (1 << 14)) >> 15; taps
 * outside the source read 0.  OpenCV version unpinned -> parity unpinned. */
int orbref_remap_linear(const uint8_t* src, int sw, int sh, int src_stride, const float* mapx, const float* mapy,
                        int dw, int dh, uint8_t* dst, int dst_stride);

/* SURVEY 8(f).4  cv::createCLAHE(clip, Size(tx, ty))->apply(src, dst) for CV_8UC1 (Examples/Monocular/mono_tum_vi.cc:101-109),
 * OpenCV 4.x imgproc/src/clahe.cpp restated: if the size is not a multiple of the tile grid the image is extended to the
 * right / bottom with BORDER_REFLECT_101 by (tiles - size % tiles); per tile: 256-bin histogram, clip at
 * max(int(clip * tileArea / 256), 1), excess spread as excess/256 per bin plus the residual in steps of max(256/residual, 1)
 * from bin 0; LUT = cvRound(cumsum * (255.f / tileArea)); per pixel bilinear blend of the four neighbouring tile LUTs in
 * float ((l11*xa1 + l12*xa)*ya1 + (l21*xa1 + l22*xa)*ya, no contraction) and cvRound.  Version unpinned (3.x spreads the
 * residual over the first bins instead) -> parity unpinned. */
int orbref_clahe(const uint8_t* src, int w, int h, int src_stride, double clip_limit, int tiles_x, int tiles_y,
                 uint8_t* dst, int dst_stride);

#ifdef __cplusplus
}
#endif
#endif
