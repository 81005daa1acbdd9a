// ORBmatcher.h -- drop-in C++ facade for ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:35-111) over include/orbm.h.
// The searches are templates on the Frame / KeyFrame / MapPoint types so this header compiles without the rest of
// ORB-SLAM3; instantiated with the reference's own classes they read exactly the members the original code reads,
// flatten them into the orbm_* plain-array views, run the GPU distance phase and write the results back
// (mvpMapPoints / vMatchedPairs) -- the member names below are the reference's (include/Frame.h, include/KeyFrame.h).
#pragma once
#include <stdexcept>
#include <string>
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

class ORBmatcher {
public:
    static const int TH_LOW = ORBM_TH_LOW, TH_HIGH = ORBM_TH_HIGH, HISTO_LENGTH = ORBM_HISTO_LENGTH;   // ORBmatcher.cc:36-38

    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
        if (orbm_create(&h, device) != ORBM_OK) throw std::runtime_error(std::string("orbm_create: ") + orbm_last_error());
    }
    ~ORBmatcher() { if (h) orbm_destroy(h); }
    ORBmatcher(const ORBmatcher&) = delete;

    // ORBmatcher.cc:2911-2931
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbm_hamming(a.data, b.data); }

    // Flattened view of a Frame (Frame.h: N, mvKeysUn, mDescriptors, mvuRight, mGrid, mnMinX...).  FrameT is the
    // reference's Frame; the static members FRAME_GRID_COLS/ROWS are 64/48 (Frame.h:37-38).
    template <class FrameT> struct View {
        std::vector<int32_t> gs, gi;
        orbm_frame_t f;
        explicit View(const FrameT& F) {
            gs.assign(ORBM_GRID_COLS * ORBM_GRID_ROWS + 1, 0);
            for (int ix = 0; ix < ORBM_GRID_COLS; ++ix)
                for (int iy = 0; iy < ORBM_GRID_ROWS; ++iy) {
                    gs[ix * ORBM_GRID_ROWS + iy] = (int32_t)gi.size();
                    for (size_t j = 0; j < F.mGrid[ix][iy].size(); ++j) gi.push_back((int32_t)F.mGrid[ix][iy][j]);
                }
            gs[ORBM_GRID_COLS * ORBM_GRID_ROWS] = (int32_t)gi.size();
            f.n = F.N; f.kps = (const orbm_kp_t*)F.mvKeysUn.data(); f.desc = F.mDescriptors.data;
            f.uright = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
            f.min_x = F.mnMinX; f.min_y = F.mnMinY; f.inv_w = F.mfGridElementWidthInv; f.inv_h = F.mfGridElementHeightInv;
            f.grid_start = gs.data(); f.grid_idx = gi.data();
        }
    };

#ifdef ORBX_WITH_OPENCV   // needs cv::Mat algebra (pose products), exactly as the reference writes it
    // ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  (ORBmatcher.cc:2469-2711), mono / rectified stereo.
    // The projection itself (camera model, pose) stays with the caller's types, exactly as written in the reference.
    template <class FrameT>
    int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
        const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3), tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat twc = -Rcw.t() * tcw;
        const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3), tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat tlc = Rlw * twc + tlw;
        const bool bForward = tlc.template at<float>(2) > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc.template at<float>(2) > CurrentFrame.mb && !bMono;
        const int nq = LastFrame.N;
        std::vector<uint8_t> valid(nq, 0), obs(nq, 0), qdesc((size_t)nq * 32, 0);
        std::vector<float> u(nq, 0), v(nq, 0), invz(nq, 0), ang(nq, 0);
        std::vector<int32_t> oct(nq, 0);
        for (int i = 0; i < nq; i++) {
            auto* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            cv::Mat x3Dc = Rcw * pMP->GetWorldPos() + tcw;
            const float invzc = 1.0 / x3Dc.template at<float>(2);
            if (invzc < 0) continue;
            cv::Point2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX || uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
            valid[i] = 1; u[i] = uv.x; v[i] = uv.y; invz[i] = invzc;
            oct[i] = LastFrame.mvKeys[i].octave; ang[i] = LastFrame.mvKeysUn[i].angle;
            obs[i] = pMP->Observations() > 0;
            std::memcpy(&qdesc[(size_t)i * 32], pMP->GetDescriptor().data, 32);
        }
        std::vector<uint8_t> blocked(CurrentFrame.N, 0);
        for (int i = 0; i < CurrentFrame.N; ++i)
            blocked[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;
        View<FrameT> cur(CurrentFrame);
        std::vector<int32_t> match(CurrentFrame.N, -1);
        const int n = orbm_search_by_projection_frame(h, &cur.f, blocked.data(), CurrentFrame.mvScaleFactors.data(), nq, valid.data(),
                                                      u.data(), v.data(), invz.data(), oct.data(), ang.data(), qdesc.data(), obs.data(),
                                                      th, bForward, bBackward, CurrentFrame.mbf, mbCheckOrientation, match.data());
        if (n < 0) throw std::runtime_error(std::string("orbm_search_by_projection_frame: ") + orbm_last_error());
        for (int i2 = 0; i2 < CurrentFrame.N; ++i2)
            if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[match[i2]];
        return n;
    }

#endif  // ORBX_WITH_OPENCV

    // ORBmatcher::SearchForTriangulation_ (ORBmatcher.cc:1388-1629), pinhole cameras.  F12 is what
    // Pinhole::epipolarConstrain_ builds from (R12,t12,K1,K2) (Pinhole.cpp:273-280); the caller computes it once.
    template <class KeyFrameT>
    int SearchForTriangulation_(KeyFrameT* pKF1, KeyFrameT* pKF2, const float* F12_rowmajor, const cv::Point2f& ep,
                                std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse) {
        auto flatten = [](const decltype(pKF1->mFeatVec)& fv, std::vector<int32_t>& nodes, std::vector<int32_t>& start, std::vector<int32_t>& idx) {
            for (auto it = fv.begin(); it != fv.end(); ++it) {
                nodes.push_back((int32_t)it->first); start.push_back((int32_t)idx.size());
                for (unsigned k : it->second) idx.push_back((int32_t)k);
            }
            start.push_back((int32_t)idx.size());
        };
        std::vector<int32_t> n1, s1, i1, n2, s2, i2;
        flatten(pKF1->mFeatVec, n1, s1, i1); flatten(pKF2->mFeatVec, n2, s2, i2);
        std::vector<uint8_t> mp1(pKF1->N), mp2(pKF2->N);
        for (int i = 0; i < pKF1->N; ++i) mp1[i] = pKF1->GetMapPoint(i) != nullptr;
        for (int i = 0; i < pKF2->N; ++i) mp2[i] = pKF2->GetMapPoint(i) != nullptr;
        std::vector<int32_t> m12(pKF1->N, -1);
        const int n = orbm_search_for_triangulation(h, pKF1->N, (const orbm_kp_t*)pKF1->mvKeysUn.data(), pKF1->mDescriptors.data, mp1.data(),
                                                    pKF1->mvuRight.data(), (int)n1.size(), n1.data(), s1.data(), i1.data(),
                                                    pKF2->N, (const orbm_kp_t*)pKF2->mvKeysUn.data(), pKF2->mDescriptors.data, mp2.data(),
                                                    pKF2->mvuRight.data(), (int)n2.size(), n2.data(), s2.data(), i2.data(),
                                                    F12_rowmajor, ep.x, ep.y, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                                    bOnlyStereo, bCoarse, mbCheckOrientation, m12.data());
        if (n < 0) throw std::runtime_error(std::string("orbm_search_for_triangulation: ") + orbm_last_error());
        vMatchedPairs.clear();
        for (size_t i = 0; i < m12.size(); ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return n;
    }

    // The bucket search of SearchForTriangulation_ / SearchForTriangulation(+vMatchedPoints) with the geometric gate left to
    // the caller's camera model (include/orbm.h, orbm_search_for_triangulation_gated).  `gate(idx1, idx2)` is called where
    // the reference calls epipolarConstrain_ (ORBmatcher.cc:1552) / matchAndtriangulate (:1729).  Keypoints are indexed
    // as the reference does (:1467-1469): mvKeysUn when NLeft == -1, else mvKeys followed by mvKeysRight.
    template <class KeyFrameT, class Gate>
    int SearchForTriangulationGated(KeyFrameT* pKF1, KeyFrameT* pKF2, Gate&& gate, std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
        auto flatten = [](const decltype(pKF1->mFeatVec)& fv, std::vector<int32_t>& nodes, std::vector<int32_t>& start, std::vector<int32_t>& idx) {
            for (auto it = fv.begin(); it != fv.end(); ++it) {
                nodes.push_back((int32_t)it->first); start.push_back((int32_t)idx.size());
                for (unsigned k : it->second) idx.push_back((int32_t)k);
            }
            start.push_back((int32_t)idx.size());
        };
        auto keys = [](KeyFrameT* kf, std::vector<cv::KeyPoint>& tmp) -> const cv::KeyPoint* {
            if (kf->NLeft == -1) return kf->mvKeysUn.data();
            tmp.assign(kf->mvKeys.begin(), kf->mvKeys.begin() + kf->NLeft);
            tmp.insert(tmp.end(), kf->mvKeysRight.begin(), kf->mvKeysRight.end());
            return tmp.data();
        };
        std::vector<int32_t> n1, s1, i1, n2, s2, i2;
        flatten(pKF1->mFeatVec, n1, s1, i1); flatten(pKF2->mFeatVec, n2, s2, i2);
        std::vector<uint8_t> mp1(pKF1->N), mp2(pKF2->N);
        for (int i = 0; i < pKF1->N; ++i) mp1[i] = pKF1->GetMapPoint(i) != nullptr;
        for (int i = 0; i < pKF2->N; ++i) mp2[i] = pKF2->GetMapPoint(i) != nullptr;
        std::vector<cv::KeyPoint> t1, t2;
        const cv::KeyPoint* k1 = keys(pKF1, t1); const cv::KeyPoint* k2 = keys(pKF2, t2);
        std::vector<int32_t> m12(pKF1->N, -1);
        auto tramp = [](void* user, int a, int b) -> int { return (*static_cast<typename std::remove_reference<Gate>::type*>(user))(a, b) ? 1 : 0; };
        const int n = orbm_search_for_triangulation_gated(h, pKF1->N, (const orbm_kp_t*)k1, pKF1->mDescriptors.data, mp1.data(),
                                                          (int)n1.size(), n1.data(), s1.data(), i1.data(),
                                                          pKF2->N, (const orbm_kp_t*)k2, pKF2->mDescriptors.data, mp2.data(),
                                                          (int)n2.size(), n2.data(), s2.data(), i2.data(),
                                                          tramp, (void*)&gate, mbCheckOrientation, m12.data());
        if (n < 0) throw std::runtime_error(std::string("orbm_search_for_triangulation_gated: ") + orbm_last_error());
        vMatchedPairs.clear();
        for (size_t i = 0; i < m12.size(); ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return n;
    }

#ifdef ORBX_WITH_OPENCV
    // ORBmatcher::SearchForTriangulation_ when the KeyFrames carry a second camera (ORBmatcher.cc:1413-1426, 1526-1557):
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

    // ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, vMatchedPoints) (ORBmatcher.cc:1632-1821):
    // gate = matchAndtriangulate; the x3D of the last accepted candidate of each idx1 is the reference's bestPoint.
    // F12 and bOnlyStereo are unused by the reference too.
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat /*F12*/, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                               const bool /*bOnlyStereo*/, std::vector<cv::Mat>& vMatchedPoints) {
        std::vector<cv::Mat> best(pKF1->N);
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
#endif  // ORBX_WITH_OPENCV

    orbm_t* handle() { return h; }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    orbm_t* h = nullptr;
};

}  // namespace ORB_SLAM3
