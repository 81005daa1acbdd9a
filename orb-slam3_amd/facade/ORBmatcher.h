// ORBmatcher.h -- drop-in C++ facade for ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:35-111) over include/orbm.h.
//
// Every public search of the reference class is here with the reference's signature.  They are templates on the Frame /
// KeyFrame / MapPoint types, so this header compiles without the rest of ORB-SLAM3; instantiated with the reference's own
// classes they read exactly the members the original code reads (names below are the reference's: include/Frame.h,
// include/KeyFrame.h, include/MapPoint.h).  Each wrapper does the caller-side part of the original function in the original's
// own expressions (pose algebra, projection through the frame's camera model, the validity gates in front of
// GetFeaturesInArea), flattens the result into the plain arrays of the orbm_* entry point, runs the GPU distance phase and
// the bookkeeping replay there, and writes the outcome back into the objects (mvpMapPoints, vpMatched, AddObservation ...).
// The same bodies compile against real OpenCV (-DORBX_WITH_OPENCV) and against cvcompat.h.
//
// A facade ORBmatcher is as cheap to construct as the reference's value object: the reference builds one on the stack at
// every call site, several times per frame (Tracking.cc:3002,3176,3867,4175,4232), so the orbm_t handle (stream, scratch
// arena) behind it comes from a per-thread pool and goes back there in the destructor.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>
#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cvcompat.h"
#endif
#include "../../include/orbm.h"

namespace ORB_SLAM3 {

namespace facade_detail {
// per-thread pool of matcher handles, one free list per device; the handles live until the thread ends
struct HandlePool {
    static const int kMaxDev = 16;
    std::vector<orbm_t*> free_[kMaxDev];
    ~HandlePool() { for (auto& v : free_) for (orbm_t* h : v) orbm_destroy(h); }
    static HandlePool& tls() { static thread_local HandlePool p; return p; }
    static orbm_t* acquire(int dev) {
        auto& v = tls().free_[dev];
        if (!v.empty()) { orbm_t* h = v.back(); v.pop_back(); return h; }
        orbm_t* h = nullptr;
        if (orbm_create(&h, dev) != ORBM_OK) throw std::runtime_error(std::string("orbm_create: ") + orbm_last_error());   // no CPU fallback
        return h;
    }
    static void release(int dev, orbm_t* h) { if (h) tls().free_[dev].push_back(h); }
};
inline int default_device() {
    static const int d = [] { const char* e = std::getenv("ORBX_DEVICE"); const int v = e ? std::atoi(e) : 0; return v < 0 || v >= HandlePool::kMaxDev ? 0 : v; }();
    return d;
}
[[noreturn]] inline void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + orbm_last_error()); }

// Frames kept in HBM across searches.  Tracking runs two to four searches on the SAME current Frame per image (TrackWithMotionModel,
// then SearchLocalPoints: Tracking.cc:3002-3211, 3867-3891); with the reference's Frame -- whose mnId identifies its immutable
// keypoints / descriptors / mvuRight -- the first search uploads them once (orbm_frame_create builds the grid on the device) and the
// later ones send only their queries (orbm_search_by_projection_*_resident: 0.05 instead of 0.13 ms).  Per thread, the two most
// recently used frames; a Frame type without mnId (or a fisheye rig) takes the host-frame entry points.
template <class F, class = void> struct has_mnId : std::false_type {};
template <class F> struct has_mnId<F, std::void_t<decltype(std::declval<const F&>().mnId)>> : std::true_type {};
struct ResidentCache {
    struct E { unsigned long long id = 0; int n = -1, dev = -1; const void* desc = nullptr; orbm_dframe_t* df = nullptr; unsigned long long used = 0; };
    E e[2];
    unsigned long long clock = 0;
    ~ResidentCache() { for (auto& x : e) if (x.df) orbm_frame_destroy(x.df); }
    static ResidentCache& tls() { static thread_local ResidentCache c; return c; }
    template <class FrameT> orbm_dframe_t* get(orbm_t* h, int dev, const FrameT& F) {
        const unsigned long long id = (unsigned long long)F.mnId;
        for (auto& x : e) if (x.df && x.id == id && x.n == F.N && x.dev == dev && x.desc == (const void*)F.mDescriptors.data) { x.used = ++clock; return x.df; }
        E& v = e[0].used <= e[1].used ? e[0] : e[1];
        if (v.df) { orbm_frame_destroy(v.df); v.df = nullptr; }
        if (orbm_frame_create(h, ORBM_HOST, F.N, (const orbm_kp_t*)F.mvKeysUn.data(), F.mDescriptors.data, F.mvuRight.empty() ? nullptr : F.mvuRight.data(),
                              F.mnMinX, F.mnMinY, F.mfGridElementWidthInv, F.mfGridElementHeightInv, &v.df) != ORBM_OK) fail("orbm_frame_create");
        v.id = id; v.n = F.N; v.dev = dev; v.desc = (const void*)F.mDescriptors.data; v.used = ++clock;
        return v.df;
    }
};

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) -> CSR
template <class FeatVec> struct FlatFeatVec {
    std::vector<int32_t> nodes, start, idx;
    explicit FlatFeatVec(const FeatVec& fv) {
        for (auto it = fv.begin(); it != fv.end(); ++it) {
            nodes.push_back((int32_t)it->first); start.push_back((int32_t)idx.size());
            for (unsigned k : it->second) idx.push_back((int32_t)k);
        }
        start.push_back((int32_t)idx.size());
    }
};
}  // namespace facade_detail

class ORBmatcher {
public:
    static const int TH_LOW = ORBM_TH_LOW, TH_HIGH = ORBM_TH_HIGH, HISTO_LENGTH = ORBM_HISTO_LENGTH;   // ORBmatcher.cc:36-38

    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = -1)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), dev(device < 0 ? facade_detail::default_device() : device) {
        h = facade_detail::HandlePool::acquire(dev);
    }
    ~ORBmatcher() { facade_detail::HandlePool::release(dev, h); }
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    // ORBmatcher.cc:2911-2931
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbm_hamming(a.data, b.data); }

    // ------------------------------------------------------------------------------------------------------------------
    // flattened views
    // ------------------------------------------------------------------------------------------------------------------
    // A Frame (Frame.h: N, mvKeysUn, mDescriptors, mvuRight, mGrid, mnMinX ...; FRAME_GRID_COLS/ROWS are 64/48, Frame.h:37-38).
    // right = the second camera of a fisheye rig: mvKeysRight, mGridRight, descriptor rows [Nleft, N).
    template <class FrameT> struct View {
        std::vector<int32_t> gs, gi;
        orbm_frame_t f;
        explicit View(const FrameT& F, bool right = false) {
            gs.assign(ORBM_GRID_COLS * ORBM_GRID_ROWS + 1, 0);
            const auto& G = right ? F.mGridRight : F.mGrid;
            for (int ix = 0; ix < ORBM_GRID_COLS; ++ix)
                for (int iy = 0; iy < ORBM_GRID_ROWS; ++iy) {
                    gs[ix * ORBM_GRID_ROWS + iy] = (int32_t)gi.size();
                    for (size_t j = 0; j < G[ix][iy].size(); ++j) gi.push_back((int32_t)G[ix][iy][j]);
                }
            gs[ORBM_GRID_COLS * ORBM_GRID_ROWS] = (int32_t)gi.size();
            if (F.Nleft == -1) { f.n = F.N; f.kps = (const orbm_kp_t*)F.mvKeysUn.data(); f.desc = F.mDescriptors.data; }
            else if (!right) { f.n = F.Nleft; f.kps = (const orbm_kp_t*)F.mvKeys.data(); f.desc = F.mDescriptors.data; }
            else { f.n = F.N - F.Nleft; f.kps = (const orbm_kp_t*)F.mvKeysRight.data(); f.desc = F.mDescriptors.ptr(F.Nleft); }
            f.uright = (F.Nleft != -1 || F.mvuRight.empty()) ? nullptr : F.mvuRight.data();
            f.min_x = F.mnMinX; f.min_y = F.mnMinY; f.inv_w = F.mfGridElementWidthInv; f.inv_h = F.mfGridElementHeightInv;
            f.grid_start = gs.data(); f.grid_idx = gi.data();
        }
    };
    // A KeyFrame: its mGrid is protected (KeyFrame.h:319), but it is by construction the Frame's grid (KeyFrame.cc copies
    // F.mGrid / F.mGridRight), i.e. what M14 builds from the camera's keypoints in index order -- rebuilt on the GPU here.
    // cam: 0 = mvKeysUn (NLeft == -1), 1 = the left camera of a fisheye rig (mvKeys[0, NLeft)), 2 = its right camera
    // (mvKeysRight, descriptor rows [NLeft, N)).  stereo_gate: hand mvuRight to the search (only Fuse looks at it,
    // ORBmatcher.cc:1952, with the camera-local index).
    template <class KeyFrameT> struct KFView {
        std::vector<int32_t> gs, gi;
        orbm_frame_t f;
        int first = 0;                                                          // descriptor / MapPoint row of the view's keypoint 0
        KFView(orbm_t* h, KeyFrameT* pKF, bool stereo_gate = false, int cam = 0) {
            const std::vector<cv::KeyPoint>& K = cam == 0 ? pKF->mvKeysUn : cam == 1 ? pKF->mvKeys : pKF->mvKeysRight;
            const int n = cam == 1 ? pKF->NLeft : (int)K.size();
            first = cam == 2 ? pKF->NLeft : 0;
            gs.assign(ORBM_GRID_COLS * ORBM_GRID_ROWS + 1, 0); gi.assign(n > 0 ? n : 1, 0);
            f.n = n; f.kps = (const orbm_kp_t*)K.data(); f.desc = pKF->mDescriptors.ptr(first);
            f.uright = stereo_gate && !pKF->mvuRight.empty() ? pKF->mvuRight.data() : nullptr;
            f.min_x = (float)pKF->mnMinX; f.min_y = (float)pKF->mnMinY; f.inv_w = pKF->mfGridElementWidthInv; f.inv_h = pKF->mfGridElementHeightInv;
            if (n > 0 && orbm_grid_build(h, f.kps, n, f.min_x, f.min_y, f.inv_w, f.inv_h, gs.data(), gi.data()) < 0) facade_detail::fail("orbm_grid_build");
            f.grid_start = gs.data(); f.grid_idx = gi.data();
        }
    };

    // ------------------------------------------------------------------------------------------------------------------
    // M3  SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)   (ORBmatcher.cc:45-239)
    // ------------------------------------------------------------------------------------------------------------------
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                           const float thFarPoints = 50.0f) {
        const int nq = (int)vpMapPoints.size();
        std::vector<uint8_t> inL(nq, 0), inR(nq, 0), obs(nq, 0), qdesc((size_t)nq * 32, 0);
        std::vector<float> px(nq, 0), py(nq, 0), pxr(nq, 0), pyr(nq, 0), vc(nq, 0), vcr(nq, 0);
        std::vector<int32_t> lvl(nq, 0), lvlr(nq, -1);
        for (int i = 0; i < nq; ++i) {
            MapPointT* pMP = vpMapPoints[i];
            if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;          // :56-57
            if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;         // :59-60
            if (pMP->isBad()) continue;                                         // :62-63
            inL[i] = pMP->mbTrackInView; px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; vc[i] = pMP->mTrackViewCos; lvl[i] = pMP->mnTrackScaleLevel;
            if (F.Nleft == -1) pxr[i] = pMP->mTrackProjXR;                       // rectified stereo: the u_R gate (:107-117)
            else { inR[i] = pMP->mbTrackInViewR; pxr[i] = pMP->mTrackProjXR; pyr[i] = pMP->mTrackProjYR; vcr[i] = pMP->mTrackViewCosR; lvlr[i] = pMP->mnTrackScaleLevelR; }
            obs[i] = pMP->Observations() > 0;
            std::memcpy(&qdesc[(size_t)i * 32], pMP->GetDescriptor().data, 32);
        }
        auto blockedOf = [&](int first, int count) {
            std::vector<uint8_t> b(count, 0);
            for (int k = 0; k < count; ++k) { auto* p = F.mvpMapPoints[first + k]; b[k] = p && p->Observations() > 0; }   // :96-98
            return b;
        };
        if (F.Nleft == -1) {
            std::vector<uint8_t> blocked = blockedOf(0, F.N);
            std::vector<int32_t> match(F.N > 0 ? F.N : 1, -1);
            int n;
            if constexpr (facade_detail::has_mnId<FrameT>::value) {
                if (F.N > 0) {
                    n = orbm_search_by_projection_points_resident(h, facade_detail::ResidentCache::tls().get(h, dev, F), blocked.data(), F.mvScaleFactors.data(), nq,
                                                                  inL.data(), px.data(), py.data(), pxr.data(), vc.data(), lvl.data(), qdesc.data(), obs.data(),
                                                                  th, mfNNratio, match.data());
                } else n = 0;
            } else {
                View<FrameT> v(F);
                n = orbm_search_by_projection_points(h, &v.f, blocked.data(), F.mvScaleFactors.data(), nq, inL.data(), px.data(), py.data(), pxr.data(),
                                                     vc.data(), lvl.data(), qdesc.data(), obs.data(), th, mfNNratio, match.data());
            }
            if (n < 0) facade_detail::fail("orbm_search_by_projection_points");
            for (int k = 0; k < F.N; ++k) if (match[k] >= 0) F.mvpMapPoints[k] = vpMapPoints[match[k]];
            return n;
        }
        View<FrameT> vl(F, false), vr(F, true);                                  // fisheye rig: left block :65-163, right block :165-236
        const int nL = F.Nleft, nR = F.N - F.Nleft;
        std::vector<uint8_t> bl = blockedOf(0, nL), br = blockedOf(nL, nR);
        std::vector<int32_t> l2r(F.mvLeftToRightMatch.begin(), F.mvLeftToRightMatch.end()), r2l(F.mvRightToLeftMatch.begin(), F.mvRightToLeftMatch.end());
        std::vector<int32_t> ml(nL > 0 ? nL : 1, -1), mr(nR > 0 ? nR : 1, -1);
        const int n = orbm_search_by_projection_points_fisheye(h, &vl.f, &vr.f, bl.data(), br.data(), l2r.data(), r2l.data(), F.mvScaleFactors.data(), nq,
                                                               inL.data(), px.data(), py.data(), vc.data(), lvl.data(), inR.data(), pxr.data(), pyr.data(), vcr.data(),
                                                               lvlr.data(), qdesc.data(), obs.data(), th, mfNNratio, ml.data(), mr.data());
        if (n < 0) facade_detail::fail("orbm_search_by_projection_points_fisheye");
        for (int k = 0; k < nL; ++k) if (ml[k] >= 0) F.mvpMapPoints[k] = vpMapPoints[ml[k]];
        for (int k = 0; k < nR; ++k) if (mr[k] >= 0) F.mvpMapPoints[nL + k] = vpMapPoints[mr[k]];
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M4  SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)   (ORBmatcher.cc:2469-2711)
    // ------------------------------------------------------------------------------------------------------------------
    template <class FrameT>
    int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
        const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3), tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat twc = -Rcw.t() * tcw;
        const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3), tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat tlc = Rlw * twc + tlw;
        const bool bForward = tlc.template at<float>(2) > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc.template at<float>(2) > CurrentFrame.mb && !bMono;
        const bool fisheye = CurrentFrame.Nleft != -1;
        const int nq = LastFrame.N;
        std::vector<uint8_t> valid(nq, 0), obs(nq, 0), qdesc((size_t)nq * 32, 0);
        std::vector<float> u(nq, 0), v(nq, 0), ur(nq, 0), vr(nq, 0), invz(nq, 0), ang(nq, 0);
        std::vector<int32_t> oct(nq, 0);
        for (int i = 0; i < nq; i++) {
            auto* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            cv::Mat x3Dw = pMP->GetWorldPos();
            cv::Mat x3Dc = Rcw * x3Dw + tcw;
            const float invzc = 1.0 / x3Dc.template at<float>(2);
            if (invzc < 0) continue;
            cv::Point2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
            if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
            valid[i] = 1; u[i] = uv.x; v[i] = uv.y; invz[i] = invzc;
            const bool lastLeft = LastFrame.Nleft == -1 || i < LastFrame.Nleft;                        // :2530-2531, :2603-2606
            oct[i] = lastLeft ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
            ang[i] = LastFrame.Nleft == -1 ? LastFrame.mvKeysUn[i].angle : lastLeft ? LastFrame.mvKeys[i].angle : LastFrame.mvKeysRight[i - LastFrame.Nleft].angle;
            obs[i] = pMP->Observations() > 0;
            std::memcpy(&qdesc[(size_t)i * 32], pMP->GetDescriptor().data, 32);
            if (fisheye) {                                                                          // :2616-2617
                cv::Mat x3Dr = CurrentFrame.mTrl.colRange(0, 3).rowRange(0, 3) * x3Dc + CurrentFrame.mTrl.col(3);
                cv::Point2f uvr = CurrentFrame.mpCamera->project(x3Dr);
                ur[i] = uvr.x; vr[i] = uvr.y;
            }
        }
        auto blockedOf = [&](int first, int count) {
            std::vector<uint8_t> b(count, 0);
            for (int k = 0; k < count; ++k) { auto* p = CurrentFrame.mvpMapPoints[first + k]; b[k] = p && p->Observations() > 0; }   // :2565-2567
            return b;
        };
        // match >= 0: the slot takes the last frame's MapPoint; ORBM_MATCH_PRUNED: assigned, then culled by the rotation check: NULL (:2700-2708)
        auto writeBack = [&](const std::vector<int32_t>& match, int first, int count) {
            for (int k = 0; k < count; ++k) {
                if (match[k] >= 0) CurrentFrame.mvpMapPoints[first + k] = LastFrame.mvpMapPoints[match[k]];
                else if (match[k] == ORBM_MATCH_PRUNED) CurrentFrame.mvpMapPoints[first + k] = nullptr;
            }
        };
        if (!fisheye) {
            std::vector<uint8_t> blocked = blockedOf(0, CurrentFrame.N);
            std::vector<int32_t> match(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
            int n;
            if constexpr (facade_detail::has_mnId<FrameT>::value) {
                if (CurrentFrame.N > 0) {
                    n = orbm_search_by_projection_frame_resident(h, facade_detail::ResidentCache::tls().get(h, dev, CurrentFrame), blocked.data(),
                                                                 CurrentFrame.mvScaleFactors.data(), nq, valid.data(), u.data(), v.data(), invz.data(), oct.data(),
                                                                 ang.data(), qdesc.data(), obs.data(), th, bForward, bBackward, CurrentFrame.mbf,
                                                                 mbCheckOrientation, match.data());
                } else n = 0;
            } else {
                View<FrameT> cur(CurrentFrame);
                n = orbm_search_by_projection_frame(h, &cur.f, blocked.data(), CurrentFrame.mvScaleFactors.data(), nq, valid.data(), u.data(), v.data(),
                                                    invz.data(), oct.data(), ang.data(), qdesc.data(), obs.data(), th, bForward, bBackward,
                                                    CurrentFrame.mbf, mbCheckOrientation, match.data());
            }
            if (n < 0) facade_detail::fail("orbm_search_by_projection_frame");
            writeBack(match, 0, CurrentFrame.N);
            return n;
        }
        View<FrameT> vl(CurrentFrame, false), vrt(CurrentFrame, true);
        const int nL = CurrentFrame.Nleft, nR = CurrentFrame.N - nL;
        std::vector<uint8_t> bl = blockedOf(0, nL), br = blockedOf(nL, nR);
        std::vector<int32_t> ml(nL > 0 ? nL : 1, -1), mr(nR > 0 ? nR : 1, -1);
        const int n = orbm_search_by_projection_frame_fisheye(h, &vl.f, &vrt.f, bl.data(), br.data(), CurrentFrame.mvScaleFactors.data(), nq, valid.data(), u.data(),
                                                              v.data(), ur.data(), vr.data(), oct.data(), ang.data(), qdesc.data(), obs.data(), th, bForward, bBackward,
                                                              mbCheckOrientation, ml.data(), mr.data());
        if (n < 0) facade_detail::fail("orbm_search_by_projection_frame_fisheye");
        writeBack(ml, 0, nL); writeBack(mr, nL, nR);
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M5  SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist)   (ORBmatcher.cc:2723-2852)
    // ------------------------------------------------------------------------------------------------------------------
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist) {
        const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
        const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat Ow = -Rcw.t() * tcw;
        const std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();
        const int nq = (int)vpMPs.size();
        std::vector<uint8_t> valid(nq, 0), qdesc((size_t)nq * 32, 0);
        std::vector<float> u(nq, 0), v(nq, 0), ang(nq, 0);
        std::vector<int32_t> lvl(nq, 0);
        for (int i = 0; i < nq; ++i) {
            MapPointT* pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            cv::Mat x3Dw = pMP->GetWorldPos();
            cv::Mat x3Dc = Rcw * x3Dw + tcw;
            const cv::Point2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
            if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
            cv::Mat PO = x3Dw - Ow;
            float dist3D = cv::norm(PO);
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            valid[i] = 1; u[i] = uv.x; v[i] = uv.y;
            lvl[i] = pMP->PredictScale(dist3D, &CurrentFrame);
            ang[i] = pKF->mvKeysUn[i].angle;                                    // :2819
            std::memcpy(&qdesc[(size_t)i * 32], pMP->GetDescriptor().data, 32);
        }
        View<FrameT> cur(CurrentFrame);
        std::vector<uint8_t> blocked(CurrentFrame.N, 0);
        for (int k = 0; k < CurrentFrame.N; ++k) blocked[k] = CurrentFrame.mvpMapPoints[k] != nullptr;   // :2793-2794
        std::vector<int32_t> match(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
        const int n = orbm_search_by_projection_kf(h, &cur.f, blocked.data(), CurrentFrame.mvScaleFactors.data(), nq, valid.data(), u.data(), v.data(), lvl.data(),
                                                   ang.data(), qdesc.data(), th, ORBdist, mbCheckOrientation, match.data());
        if (n < 0) facade_detail::fail("orbm_search_by_projection_kf");
        for (int k = 0; k < CurrentFrame.N; ++k) {
            if (match[k] >= 0) CurrentFrame.mvpMapPoints[k] = vpMPs[match[k]];
            else if (match[k] == ORBM_MATCH_PRUNED) CurrentFrame.mvpMapPoints[k] = nullptr;                // :2843-2847
        }
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M6  SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming)   (ORBmatcher.cc:549-679) and the
    //     +vpPointsKFs / vpMatchedKF overload (:681-797)
    // ------------------------------------------------------------------------------------------------------------------
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th,
                           float ratioHamming = 1.0) {
        std::vector<int32_t> match;
        const int n = sim3_projection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, true, match);
        for (size_t idx = 0; idx < match.size() && idx < vpMatched.size(); ++idx) if (match[idx] >= 0) vpMatched[idx] = vpPoints[match[idx]];
        return n;
    }
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, const std::vector<KeyFrameT*>& vpPointsKFs,
                           std::vector<MapPointT*>& vpMatched, std::vector<KeyFrameT*>& vpMatchedKF, int th, float ratioHamming = 1.0) {
        std::vector<int32_t> match;
        const int n = sim3_projection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, false, match);
        for (size_t idx = 0; idx < match.size() && idx < vpMatched.size(); ++idx)
            if (match[idx] >= 0) { vpMatched[idx] = vpPoints[match[idx]]; vpMatchedKF[idx] = vpPointsKFs[match[idx]]; }
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M7  SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)   (ORBmatcher.cc:314-547)
    // ------------------------------------------------------------------------------------------------------------------
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches) {
        const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(nullptr));
        const int nkf = (int)vpMapPointsKF.size();
        std::vector<uint8_t> good(nkf > 0 ? nkf : 1, 0);
        for (int i = 0; i < nkf; ++i) good[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();            // :355-361
        facade_detail::FlatFeatVec<decltype(pKF->mFeatVec)> fk(pKF->mFeatVec);
        facade_detail::FlatFeatVec<decltype(F.mFeatVec)> ff(F.mFeatVec);
        std::vector<int32_t> fm(F.N > 0 ? F.N : 1, -1);
        int n;
        if (F.Nleft == -1) {
            n = orbm_search_by_bow(h, nkf, (const orbm_kp_t*)pKF->mvKeysUn.data(), pKF->mDescriptors.data, good.data(), (int)fk.nodes.size(), fk.nodes.data(),
                                   fk.start.data(), fk.idx.data(), F.N, (const orbm_kp_t*)F.mvKeys.data(), F.mDescriptors.data, (int)ff.nodes.size(), ff.nodes.data(),
                                   ff.start.data(), ff.idx.data(), mfNNratio, mbCheckOrientation, fm.data());     // the angle comes from F.mvKeys (:441-445)
        } else {
            std::vector<cv::KeyPoint> kf(F.mvKeys.begin(), F.mvKeys.begin() + F.Nleft);                      // frame features: left keys, then right keys (:441-445, :478-482)
            kf.insert(kf.end(), F.mvKeysRight.begin(), F.mvKeysRight.end());
            std::vector<cv::KeyPoint> kk;                                                                    // KeyFrame side (:443, :480)
            const cv::KeyPoint* kkp = pKF->mvKeysUn.data();
            if (pKF->mpCamera2) { kk.assign(pKF->mvKeys.begin(), pKF->mvKeys.begin() + pKF->NLeft); kk.insert(kk.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end()); kkp = kk.data(); }
            n = orbm_search_by_bow_fisheye(h, nkf, (const orbm_kp_t*)kkp, pKF->mDescriptors.data, good.data(), (int)fk.nodes.size(), fk.nodes.data(), fk.start.data(),
                                           fk.idx.data(), F.N, F.Nleft, (const orbm_kp_t*)kf.data(), F.mDescriptors.data, (int)ff.nodes.size(), ff.nodes.data(),
                                           ff.start.data(), ff.idx.data(), mfNNratio, mbCheckOrientation, fm.data());
        }
        if (n < 0) facade_detail::fail("orbm_search_by_bow");
        for (int iF = 0; iF < F.N; ++iF) if (fm[iF] >= 0) vpMapPointMatches[iF] = vpMapPointsKF[fm[iF]];
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M8  SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)   (ORBmatcher.cc:955-1105)
    // ------------------------------------------------------------------------------------------------------------------
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12) {
        const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
        vpMatches12 = std::vector<MapPointT*>(n1, static_cast<MapPointT*>(nullptr));
        auto goodOf = [](KeyFrameT* kf, const std::vector<MapPointT*>& mp) {
            std::vector<uint8_t> g(mp.empty() ? 1 : mp.size(), 0);
            for (size_t i = 0; i < mp.size(); ++i)                                                           // :993-1000, :1012-1022
                g[i] = !(kf->NLeft != -1 && i >= kf->mvKeysUn.size()) && mp[i] && !mp[i]->isBad();
            return g;
        };
        const std::vector<uint8_t> g1 = goodOf(pKF1, vpMapPoints1), g2 = goodOf(pKF2, vpMapPoints2);
        facade_detail::FlatFeatVec<decltype(pKF1->mFeatVec)> f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
        std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
        const int n = orbm_search_by_bow_kf(h, n1, (const orbm_kp_t*)pKF1->mvKeysUn.data(), pKF1->mDescriptors.data, g1.data(), (int)f1.nodes.size(), f1.nodes.data(),
                                            f1.start.data(), f1.idx.data(), n2, (const orbm_kp_t*)pKF2->mvKeysUn.data(), pKF2->mDescriptors.data, g2.data(),
                                            (int)f2.nodes.size(), f2.nodes.data(), f2.start.data(), f2.idx.data(), mfNNratio, mbCheckOrientation, m12.data());
        if (n < 0) facade_detail::fail("orbm_search_by_bow_kf");
        for (int i = 0; i < n1; ++i) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M9  SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)   (ORBmatcher.cc:799-943)
    // ------------------------------------------------------------------------------------------------------------------
    template <class FrameT>
    int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {
        View<FrameT> v1(F1), v2(F2);
        const int n1 = (int)F1.mvKeysUn.size();
        std::vector<float> prev((size_t)2 * (n1 > 0 ? n1 : 1), 0.f);
        for (int i = 0; i < n1; ++i) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
        const int n = orbm_search_for_initialization(h, &v1.f, &v2.f, prev.data(), windowSize, mfNNratio, mbCheckOrientation, m12.data());
        if (n < 0) facade_detail::fail("orbm_search_for_initialization");
        vnMatches12 = std::vector<int>(m12.begin(), m12.begin() + n1);
        for (int i = 0; i < n1; ++i) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }   // :938-940 (updated only where matched)
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M10 SearchForTriangulation_(pKF1, pKF2, cv::Matx33f F12, vMatchedPairs, bOnlyStereo, bCoarse)   (ORBmatcher.cc:1388-1629)
    //     The reference never reads F12: its geometric gate is the VIRTUAL pCamera1->epipolarConstrain_(pCamera2, kp1, kp2,
    //     R12, t12, ...) (:1555).  Dispatch on the camera model as that call does:
    //       * both KeyFrames single-camera Pinhole: Pinhole::epipolarConstrain_ (Pinhole.cpp:273-299) rebuilds
    //         K1^-T [t12]x R12 K2^-1 from the same R12 / t12 -- the very expression LocalMapping::ComputeF12_
    //         (LocalMapping.cc:1102-1119) hands in as F12 -- and tests the epipolar-line distance: device fast path;
    //       * a single camera of any other model (monocular KannalaBrandt8: TriangulateMatches_ > 0.0001f,
    //         KannalaBrandt8.cpp:356-360): the bucket search with the camera object's own gate, R12 / t12 as :1413-1416;
    //       * a second camera (mpCamera2): the four-pose gate of :1417-1426, 1526-1557.
    // ------------------------------------------------------------------------------------------------------------------
    template <class KeyFrameT>
    int SearchForTriangulation_(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Matx33f F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                const bool bOnlyStereo, const bool bCoarse = false) {
        if (pKF1->mpCamera2 || pKF2->mpCamera2) return SearchForTriangulationTwoCameras_(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse);
        // epipole in the second image (:1393-1400)
        auto Cw = pKF1->GetCameraCenter_();
        auto R2w = pKF2->GetRotation_();
        auto t2w = pKF2->GetTranslation_();
        auto C2 = R2w * Cw + t2w;
        cv::Point2f ep = pKF2->mpCamera->project(C2);
        const bool pinhole = pKF1->mpCamera->GetType() == pKF1->mpCamera->CAM_PINHOLE && pKF2->mpCamera->GetType() == pKF2->mpCamera->CAM_PINHOLE;
        if (!pinhole) return SearchForTriangulationOneCamera_(pKF1, pKF2, ep, vMatchedPairs, bOnlyStereo, bCoarse);
        float F[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[3 * i + j] = F12(i, j);
        return triangulation(pKF1, pKF2, F, ep, vMatchedPairs, bOnlyStereo, bCoarse, false);
    }

    // SearchForTriangulation_ for single-camera KeyFrames whose camera model is not Pinhole.  Loop conditions in the
    // reference's order: MapPoint / bOnlyStereo skips (:1458-1466, 1488-1497, folded into the skip flags of the bucket
    // search: both are plain `continue`s on the feature), TH_LOW / bestDist (:1503-1504, on the device), the epipole gate
    // (:1512-1520, !bStereo1 && !bStereo2 && !mpCamera2), then pCamera1->epipolarConstrain_(pCamera2, ...) || bCoarse (:1555).
    template <class KeyFrameT>
    int SearchForTriangulationOneCamera_(KeyFrameT* pKF1, KeyFrameT* pKF2, const cv::Point2f& ep, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                         const bool bOnlyStereo, const bool bCoarse) {
        const cv::Matx33f R1w = pKF1->GetRotation_(), R2w = pKF2->GetRotation_();
        const cv::Matx31f t1w = pKF1->GetTranslation_(), t2w = pKF2->GetTranslation_();
        const cv::Matx33f R12 = R1w * R2w.t();                                              // :1414
        const cv::Matx31f t12 = -R1w * R2w.t() * t2w + t1w;                                 // :1415
        auto* pCamera1 = pKF1->mpCamera; auto* pCamera2 = pKF2->mpCamera;
        auto key = [](KeyFrameT* kf, int i) -> const cv::KeyPoint& {
            return kf->NLeft == -1 ? kf->mvKeysUn[i] : i < kf->NLeft ? kf->mvKeys[i] : kf->mvKeysRight[i - kf->NLeft];
        };
        auto gate = [&](int idx1, int idx2) -> bool {
            const cv::KeyPoint& kp1 = key(pKF1, idx1); const cv::KeyPoint& kp2 = key(pKF2, idx2);
            const bool bStereo1 = pKF1->mvuRight[idx1] >= 0, bStereo2 = pKF2->mvuRight[idx2] >= 0;
            if (!bStereo1 && !bStereo2) {
                const float distex = ep.x - kp2.pt.x;
                const float distey = ep.y - kp2.pt.y;
                if (distex * distex + distey * distey < 100 * pKF2->mvScaleFactors[kp2.octave]) return false;
            }
            return pCamera1->epipolarConstrain_(pCamera2, kp1, kp2, R12, t12, pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave]) || bCoarse;
        };
        auto skip = [bOnlyStereo](KeyFrameT* kf, int i) -> bool { return bOnlyStereo && !(kf->mvuRight[i] >= 0); };
        return SearchForTriangulationGated(pKF1, pKF2, gate, vMatchedPairs, skip);
    }

    // M11 SearchForTriangulation(pKF1, pKF2, cv::Mat F12, vMatchedPairs, bOnlyStereo, bCoarse)   (ORBmatcher.cc:1107-1386)
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                               const bool bOnlyStereo, const bool bCoarse = false) {
        // epipole in the second image (:1114-1125)
        cv::Mat Cw = pKF1->GetCameraCenter();
        cv::Mat R2w = pKF2->GetRotation();
        cv::Mat t2w = pKF2->GetTranslation();
        cv::Mat C2 = R2w * Cw + t2w;
        cv::Point2f ep = pKF2->mpCamera->project(C2);
        float F[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[3 * i + j] = F12.template at<float>(i, j);
        return triangulation(pKF1, pKF2, F, ep, vMatchedPairs, bOnlyStereo, bCoarse, true);
    }

    // The bucket search of SearchForTriangulation_ / SearchForTriangulation(+vMatchedPoints) with the geometric gate left to
    // the caller's camera model (include/orbm.h, orbm_search_for_triangulation_gated).  `gate(idx1, idx2)` is called where
    // the reference calls epipolarConstrain_ (ORBmatcher.cc:1552) / matchAndtriangulate (:1729).  Keypoints are indexed
    // as the reference does (:1467-1469): mvKeysUn when NLeft == -1, else mvKeys followed by mvKeysRight.
    template <class KeyFrameT, class Gate>
    int SearchForTriangulationGated(KeyFrameT* pKF1, KeyFrameT* pKF2, Gate&& gate, std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
        return SearchForTriangulationGated(pKF1, pKF2, gate, vMatchedPairs, [](KeyFrameT*, int) { return false; });
    }
    // `skip(kf, idx)`: features the reference passes over with a `continue` next to the MapPoint test (bOnlyStereo, :1464-1466, 1494-1497)
    template <class KeyFrameT, class Gate, class Skip>
    int SearchForTriangulationGated(KeyFrameT* pKF1, KeyFrameT* pKF2, Gate&& gate, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, Skip&& skip) {
        auto keys = [](KeyFrameT* kf, std::vector<cv::KeyPoint>& tmp) -> const cv::KeyPoint* {
            if (kf->NLeft == -1) return kf->mvKeysUn.data();
            tmp.assign(kf->mvKeys.begin(), kf->mvKeys.begin() + kf->NLeft);
            tmp.insert(tmp.end(), kf->mvKeysRight.begin(), kf->mvKeysRight.end());
            return tmp.data();
        };
        facade_detail::FlatFeatVec<decltype(pKF1->mFeatVec)> f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
        std::vector<uint8_t> mp1(pKF1->N > 0 ? pKF1->N : 1), mp2(pKF2->N > 0 ? pKF2->N : 1);
        for (int i = 0; i < pKF1->N; ++i) mp1[i] = pKF1->GetMapPoint(i) != nullptr || skip(pKF1, i);
        for (int i = 0; i < pKF2->N; ++i) mp2[i] = pKF2->GetMapPoint(i) != nullptr || skip(pKF2, i);
        std::vector<cv::KeyPoint> t1, t2;
        const cv::KeyPoint* k1 = keys(pKF1, t1); const cv::KeyPoint* k2 = keys(pKF2, t2);
        std::vector<int32_t> m12(pKF1->N > 0 ? pKF1->N : 1, -1);
        auto tramp = [](void* user, int a, int b) -> int { return (*static_cast<typename std::remove_reference<Gate>::type*>(user))(a, b) ? 1 : 0; };
        const int n = orbm_search_for_triangulation_gated(h, pKF1->N, (const orbm_kp_t*)k1, pKF1->mDescriptors.data, mp1.data(),
                                                          (int)f1.nodes.size(), f1.nodes.data(), f1.start.data(), f1.idx.data(),
                                                          pKF2->N, (const orbm_kp_t*)k2, pKF2->mDescriptors.data, mp2.data(),
                                                          (int)f2.nodes.size(), f2.nodes.data(), f2.start.data(), f2.idx.data(),
                                                          tramp, (void*)&gate, mbCheckOrientation, m12.data());
        if (n < 0) facade_detail::fail("orbm_search_for_triangulation_gated");
        vMatchedPairs.clear();
        for (int i = 0; i < pKF1->N; ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return n;
    }

    // SearchForTriangulation_ when the KeyFrames carry a second camera (ORBmatcher.cc:1413-1426, 1526-1557):
    // relative poses for the four left/right combinations, gate = pCamera1->epipolarConstrain_(...) || bCoarse.
    template <class KeyFrameT>
    int SearchForTriangulationTwoCameras_(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                          const bool bOnlyStereo, const bool bCoarse) {
        if (bOnlyStereo) { vMatchedPairs.clear(); return 0; }              // bStereo1 is false with mpCamera2 (:1462), so every feature is skipped (:1464-1466)
        const cv::Matx33f Rl1 = pKF1->GetRotation_(), Rr1 = pKF1->GetRightRotation_(), Rl2 = pKF2->GetRotation_(), Rr2 = pKF2->GetRightRotation_();
        const cv::Matx31f tl1 = pKF1->GetTranslation_(), tr1 = pKF1->GetRightTranslation_(), tl2 = pKF2->GetTranslation_(), tr2 = pKF2->GetRightTranslation_();
        const cv::Matx33f R[4] = {Rl1 * Rl2.t(), Rl1 * Rr2.t(), Rr1 * Rl2.t(), Rr1 * Rr2.t()};                              // ll, lr, rl, rr (:1418-1421)
        const cv::Matx31f t[4] = {Rl1 * (-Rl2.t() * tl2) + tl1, Rl1 * (-Rr2.t() * tr2) + tl1, Rr1 * (-Rl2.t() * tl2) + tr1, Rr1 * (-Rr2.t() * tr2) + tr1};
        auto key = [](KeyFrameT* kf, int i) -> const cv::KeyPoint& { return i < kf->NLeft ? kf->mvKeys[i] : kf->mvKeysRight[i - kf->NLeft]; };
        auto gate = [&](int idx1, int idx2) -> bool {
            const bool r1 = idx1 >= pKF1->NLeft, r2 = idx2 >= pKF2->NLeft;
            const int c = (r1 ? 2 : 0) + (r2 ? 1 : 0);
            auto* cam1 = r1 ? pKF1->mpCamera2 : pKF1->mpCamera;
            auto* cam2 = r2 ? pKF2->mpCamera2 : pKF2->mpCamera;
            const cv::KeyPoint& kp1 = key(pKF1, idx1); const cv::KeyPoint& kp2 = key(pKF2, idx2);
            return cam1->epipolarConstrain_(cam2, kp1, kp2, R[c], t[c], pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave]) || bCoarse;
        };
        return SearchForTriangulationGated(pKF1, pKF2, gate, vMatchedPairs);
    }

    // M12 SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, vMatchedPoints) (ORBmatcher.cc:1632-1821):
    // gate = matchAndtriangulate; the x3D of the last accepted candidate of each idx1 is the reference's bestPoint.
    // F12 and bOnlyStereo are unused by the reference too.
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat /*F12*/, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                               const bool /*bOnlyStereo*/, std::vector<cv::Mat>& vMatchedPoints) {
        std::vector<cv::Mat> best(pKF1->N > 0 ? pKF1->N : 1);
        auto key = [](KeyFrameT* kf, int i) -> const cv::KeyPoint& {
            return kf->NLeft == -1 ? kf->mvKeysUn[i] : i < kf->NLeft ? kf->mvKeys[i] : kf->mvKeysRight[i - kf->NLeft];
        };
        auto gate = [&](int idx1, int idx2) -> bool {
            const bool r1 = pKF1->NLeft != -1 && idx1 >= pKF1->NLeft, r2 = pKF2->NLeft != -1 && idx2 >= pKF2->NLeft;
            cv::Mat Tcw1 = r1 ? pKF1->GetRightPose() : pKF1->GetPose(), Tcw2 = r2 ? pKF2->GetRightPose() : pKF2->GetPose();
            auto* cam1 = r1 ? pKF1->mpCamera2 : pKF1->mpCamera;
            auto* cam2 = r2 ? pKF2->mpCamera2 : pKF2->mpCamera;
            const cv::KeyPoint& kp1 = key(pKF1, idx1); const cv::KeyPoint& kp2 = key(pKF2, idx2);
            cv::Mat x3D;
            if (!cam1->matchAndtriangulate(kp1, kp2, cam2, Tcw1, Tcw2, pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave], x3D)) return false;
            best[idx1] = x3D;
            return true;
        };
        const int n = SearchForTriangulationGated(pKF1, pKF2, gate, vMatchedPairs);
        for (const auto& pr : vMatchedPairs) vMatchedPoints.push_back(best[pr.first]);       // appended, not cleared (:1815)
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M13 SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)   (ORBmatcher.cc:2201-2467)
    // ------------------------------------------------------------------------------------------------------------------
    template <class KeyFrameT, class MapPointT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12,
                     const float th) {
        const float& fx = pKF1->fx; const float& fy = pKF1->fy; const float& cx = pKF1->cx; const float& cy = pKF1->cy;
        cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
        cv::Mat sR12 = s12 * R12;
        cv::Mat sR21 = (1.0 / s12) * R12.t();
        cv::Mat t21 = -sR21 * t12;
        const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPointT* pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        // one direction: the MapPoints of `src` (in its camera frame through Rsw, tsw) moved by (sR, tt) into `dst` and projected (:2246-2292 / :2339-2384)
        auto project = [&](const std::vector<MapPointT*>& mps, const std::vector<bool>& already, const cv::Mat& Rsw, const cv::Mat& tsw, const cv::Mat& sR, const cv::Mat& tt,
                           KeyFrameT* dst, std::vector<uint8_t>& valid, std::vector<float>& u, std::vector<float>& v, std::vector<int32_t>& lvl, std::vector<uint8_t>& qd) {
            const int N = (int)mps.size();
            valid.assign(N > 0 ? N : 1, 0); u.assign(N > 0 ? N : 1, 0.f); v.assign(N > 0 ? N : 1, 0.f); lvl.assign(N > 0 ? N : 1, 0); qd.assign((size_t)(N > 0 ? N : 1) * 32, 0);
            for (int i = 0; i < N; ++i) {
                MapPointT* pMP = mps[i];
                if (!pMP || already[i]) continue;
                if (pMP->isBad()) continue;
                cv::Mat p3Dw = pMP->GetWorldPos();
                cv::Mat p3Dcs = Rsw * p3Dw + tsw;
                cv::Mat p3Dcd = sR * p3Dcs + tt;
                if (p3Dcd.template at<float>(2) < 0.0) continue;
                const float invz = 1.0 / p3Dcd.template at<float>(2);
                const float x = p3Dcd.template at<float>(0) * invz;
                const float y = p3Dcd.template at<float>(1) * invz;
                const float uu = fx * x + cx;
                const float vv = fy * y + cy;
                if (!dst->IsInImage(uu, vv)) continue;
                const float maxDistance = pMP->GetMaxDistanceInvariance();
                const float minDistance = pMP->GetMinDistanceInvariance();
                const float dist3D = cv::norm(p3Dcd);
                if (dist3D < minDistance || dist3D > maxDistance) continue;
                valid[i] = 1; u[i] = uu; v[i] = vv;
                lvl[i] = pMP->PredictScale(dist3D, dst);
                std::memcpy(&qd[(size_t)i * 32], pMP->GetDescriptor().data, 32);
            }
        };
        std::vector<uint8_t> valid1, valid2, qd1, qd2;
        std::vector<float> u1, v1, u2, v2;
        std::vector<int32_t> l1, l2;
        project(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, valid1, u1, v1, l1, qd1);
        project(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1, valid2, u2, v2, l2, qd2);
        KFView<KeyFrameT> k1(h, pKF1), k2(h, pKF2);
        std::vector<int32_t> m12(N1 > 0 ? N1 : 1, -1);
        const int n = orbm_search_by_sim3(h, &k1.f, &k2.f, pKF1->mvScaleFactors.data(), pKF2->mvScaleFactors.data(), valid1.data(), u1.data(), v1.data(), l1.data(),
                                          qd1.data(), valid2.data(), u2.data(), v2.data(), l2.data(), qd2.data(), th, m12.data());
        if (n < 0) facade_detail::fail("orbm_search_by_sim3");
        for (int i1 = 0; i1 < N1; ++i1) if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];         // :2455-2458
        return n;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // M13 Fuse(pKF, vpMapPoints, th, bRight)   (ORBmatcher.cc:1823-2049)
    // ------------------------------------------------------------------------------------------------------------------
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0, const bool bRight = false) {
        cv::Mat Rcw, tcw, Ow;
        auto* pCamera = pKF->mpCamera;
        if (bRight) { Rcw = pKF->GetRightRotation(); tcw = pKF->GetRightTranslation(); Ow = pKF->GetRightCameraCenter(); pCamera = pKF->mpCamera2; }
        else { Rcw = pKF->GetRotation(); tcw = pKF->GetTranslation(); Ow = pKF->GetCameraCenter(); }
        const float& bf = pKF->mbf;
        const int nMPs = (int)vpMapPoints.size();
        std::vector<uint8_t> valid(nMPs > 0 ? nMPs : 1, 0), qd((size_t)(nMPs > 0 ? nMPs : 1) * 32, 0);
        std::vector<float> u(nMPs > 0 ? nMPs : 1, 0.f), v(nMPs > 0 ? nMPs : 1, 0.f), ur(nMPs > 0 ? nMPs : 1, 0.f);
        std::vector<int32_t> lvl(nMPs > 0 ? nMPs : 1, 0);
        for (int i = 0; i < nMPs; i++) {
            MapPointT* pMP = vpMapPoints[i];
            if (!pMP) continue;
            if (pMP->isBad()) continue;
            else if (pMP->IsInKeyFrame(pKF)) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            cv::Mat p3Dc = Rcw * p3Dw + tcw;
            if (p3Dc.template at<float>(2) < 0.0f) continue;
            const float invz = 1 / p3Dc.template at<float>(2);
            const float x = p3Dc.template at<float>(0), y = p3Dc.template at<float>(1), z = p3Dc.template at<float>(2);
            const cv::Point2f uv = pCamera->project(cv::Point3f(x, y, z));
            if (!pKF->IsInImage(uv.x, uv.y)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            cv::Mat PO = p3Dw - Ow;
            const float dist3D = cv::norm(PO);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            cv::Mat Pn = pMP->GetNormal();
            if (PO.dot(Pn) < 0.5 * dist3D) continue;
            valid[i] = 1; u[i] = uv.x; v[i] = uv.y; ur[i] = uv.x - bf * invz;
            lvl[i] = pMP->PredictScale(dist3D, pKF);
            std::memcpy(&qd[(size_t)i * 32], pMP->GetDescriptor().data, 32);
        }
        std::vector<int32_t> best(nMPs > 0 ? nMPs : 1, -1);
        // the KeyFrame's view: mvKeysUn, or the left / right camera's keys, grid and descriptor rows (:1936-1940, :1994)
        KFView<KeyFrameT> kfv(h, pKF, true, pKF->NLeft == -1 ? 0 : bRight ? 2 : 1);
        const int first = kfv.first;
        if (orbm_fuse(h, &kfv.f, pKF->mvScaleFactors.data(), pKF->mvInvLevelSigma2.data(), nMPs, valid.data(), u.data(), v.data(), ur.data(), lvl.data(), qd.data(), th, 1,
                      best.data()) < 0) facade_detail::fail("orbm_fuse");
        // map surgery in the reference's order (:2005-2041); a point an earlier iteration made bad / put into the KeyFrame is skipped as there
        int nFused = 0;
        for (int i = 0; i < nMPs; ++i) {
            if (best[i] < 0) continue;
            MapPointT* pMP = vpMapPoints[i];
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            const int bestIdx = best[i] + first;
            MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

    // M13 Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)   (ORBmatcher.cc:2051-2199)
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint) {
        cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
        cv::Mat Rcw = sRcw / scw;
        cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
        cv::Mat Ow = -Rcw.t() * tcw;
        const std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();
        const int nPoints = (int)vpPoints.size();
        std::vector<uint8_t> valid(nPoints > 0 ? nPoints : 1, 0), qd((size_t)(nPoints > 0 ? nPoints : 1) * 32, 0);
        std::vector<float> u(nPoints > 0 ? nPoints : 1, 0.f), v(nPoints > 0 ? nPoints : 1, 0.f), ur(nPoints > 0 ? nPoints : 1, 0.f);
        std::vector<int32_t> lvl(nPoints > 0 ? nPoints : 1, 0);
        for (int iMP = 0; iMP < nPoints; iMP++) {
            MapPointT* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            float dist3D;
            cv::Point2f uv;
            if (!sim3_gates(pKF, pMP, Rcw, tcw, Ow, true, uv, dist3D)) continue;
            valid[iMP] = 1; u[iMP] = uv.x; v[iMP] = uv.y;
            lvl[iMP] = pMP->PredictScale(dist3D, pKF);
            std::memcpy(&qd[(size_t)iMP * 32], pMP->GetDescriptor().data, 32);
        }
        KFView<KeyFrameT> kf(h, pKF);
        std::vector<int32_t> best(nPoints > 0 ? nPoints : 1, -1);
        const int rc = orbm_fuse(h, &kf.f, pKF->mvScaleFactors.data(), pKF->mvInvLevelSigma2.data(), nPoints, valid.data(), u.data(), v.data(), ur.data(), lvl.data(),
                                 qd.data(), th, 0, best.data());
        if (rc < 0) facade_detail::fail("orbm_fuse");
        int nFused = 0;
        for (int iMP = 0; iMP < nPoints; ++iMP) {                                   // :2175-2193
            if (best[iMP] < 0) continue;
            MapPointT* pMP = vpPoints[iMP];
            MapPointT* pMPinKF = pKF->GetMapPoint(best[iMP]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, best[iMP]);
                pKF->AddMapPoint(pMP, best[iMP]);
            }
            nFused++;
        }
        return nFused;
    }

    orbm_t* handle() { return h; }

protected:
    // caller-side gates shared by the Sim3 searches (ORBmatcher.cc:590-640, 704-742, 2084-2124): camera-frame point, positive depth,
    // projection inside the image, distance invariance, viewing angle below 60 degrees.  useCamera: project through
    // pKF->mpCamera (:611, :2097) or with the pinhole expression fx*x/z+cx (:718-722).
    template <class KeyFrameT, class MapPointT>
    bool sim3_gates(KeyFrameT* pKF, MapPointT* pMP, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow, bool useCamera, cv::Point2f& uv, float& dist) {
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.template at<float>(2) < 0.0) return false;
        if (useCamera) {
            const float x = p3Dc.template at<float>(0), y = p3Dc.template at<float>(1), z = p3Dc.template at<float>(2);
            uv = pKF->mpCamera->project(cv::Point3f(x, y, z));
        } else {
            const float invz = 1 / p3Dc.template at<float>(2);
            const float x = p3Dc.template at<float>(0) * invz, y = p3Dc.template at<float>(1) * invz;
            uv = cv::Point2f(pKF->fx * x + pKF->cx, pKF->fy * y + pKF->cy);
        }
        if (!pKF->IsInImage(uv.x, uv.y)) return false;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        dist = cv::norm(PO);
        if (dist < minDistance || dist > maxDistance) return false;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist) return false;
        return true;
    }

    template <class KeyFrameT, class MapPointT>
    int sim3_projection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, const std::vector<MapPointT*>& vpMatched, int th, float ratioHamming,
                        bool useCamera, std::vector<int32_t>& match) {
        cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
        cv::Mat Rcw = sRcw / scw;
        cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
        cv::Mat Ow = -Rcw.t() * tcw;
        std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPointT*>(nullptr));
        const int nq = (int)vpPoints.size();
        std::vector<uint8_t> valid(nq > 0 ? nq : 1, 0), qd((size_t)(nq > 0 ? nq : 1) * 32, 0);
        std::vector<float> u(nq > 0 ? nq : 1, 0.f), v(nq > 0 ? nq : 1, 0.f);
        std::vector<int32_t> lvl(nq > 0 ? nq : 1, 0);
        for (int iMP = 0; iMP < nq; ++iMP) {
            MapPointT* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            float dist;
            cv::Point2f uv;
            if (!sim3_gates(pKF, pMP, Rcw, tcw, Ow, useCamera, uv, dist)) continue;
            valid[iMP] = 1; u[iMP] = uv.x; v[iMP] = uv.y;
            lvl[iMP] = pMP->PredictScale(dist, pKF);
            std::memcpy(&qd[(size_t)iMP * 32], pMP->GetDescriptor().data, 32);
        }
        KFView<KeyFrameT> kf(h, pKF);
        std::vector<uint8_t> matchedIn(kf.f.n > 0 ? kf.f.n : 1, 0);
        for (int idx = 0; idx < kf.f.n && idx < (int)vpMatched.size(); ++idx) matchedIn[idx] = vpMatched[idx] != nullptr;      // :653-654
        match.assign(kf.f.n > 0 ? kf.f.n : 1, -1);
        const int n = orbm_search_by_projection_sim3(h, &kf.f, matchedIn.data(), pKF->mvScaleFactors.data(), nq, valid.data(), u.data(), v.data(), lvl.data(), qd.data(),
                                                     th, ratioHamming, match.data());
        if (n < 0) facade_detail::fail("orbm_search_by_projection_sim3");
        match.resize(kf.f.n);
        return n;
    }

    // M10 / M11 body: pinhole cameras, F12 row-major (what Pinhole::epipolarConstrain_ builds, Pinhole.cpp:273-280)
    template <class KeyFrameT>
    int triangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, const float* F12, const cv::Point2f& ep, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                      const bool bOnlyStereo, const bool bCoarse, const bool legacy) {
        facade_detail::FlatFeatVec<decltype(pKF1->mFeatVec)> f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
        std::vector<uint8_t> mp1(pKF1->N > 0 ? pKF1->N : 1), mp2(pKF2->N > 0 ? pKF2->N : 1);
        for (int i = 0; i < pKF1->N; ++i) mp1[i] = pKF1->GetMapPoint(i) != nullptr;
        for (int i = 0; i < pKF2->N; ++i) mp2[i] = pKF2->GetMapPoint(i) != nullptr;
        std::vector<int32_t> m12(pKF1->N > 0 ? pKF1->N : 1, -1);
        auto fn = legacy ? orbm_search_for_triangulation_legacy : orbm_search_for_triangulation;
        const int n = fn(h, pKF1->N, (const orbm_kp_t*)pKF1->mvKeysUn.data(), pKF1->mDescriptors.data, mp1.data(), pKF1->mvuRight.data(),
                         (int)f1.nodes.size(), f1.nodes.data(), f1.start.data(), f1.idx.data(),
                         pKF2->N, (const orbm_kp_t*)pKF2->mvKeysUn.data(), pKF2->mDescriptors.data, mp2.data(), pKF2->mvuRight.data(),
                         (int)f2.nodes.size(), f2.nodes.data(), f2.start.data(), f2.idx.data(),
                         F12, ep.x, ep.y, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(), bOnlyStereo, bCoarse, mbCheckOrientation, m12.data());
        if (n < 0) facade_detail::fail("orbm_search_for_triangulation");
        vMatchedPairs.clear();
        for (int i = 0; i < pKF1->N; ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return n;
    }

    float mfNNratio;
    bool mbCheckOrientation;
    int dev = 0;
    orbm_t* h = nullptr;
};

}  // namespace ORB_SLAM3
