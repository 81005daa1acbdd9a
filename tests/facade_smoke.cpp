// Compiles the C++ facade against the minimal cv-compat layer and (on a GPU box) drives EVERY public search of
// ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:44-91) through its reference-signature wrapper, on mock Frame / KeyFrame /
// MapPoint / camera types that carry the members the reference's own classes have.  Each wrapper's outcome is compared
// with the flattened C entry point fed by hand-built arrays of the same scene, or with what the scene makes certain.
#include <chrono>
#include <cstdio>
#include <map>
#include <set>
#include <tuple>
#include <vector>
#include "../orb-slam3_amd/facade/ORBextractor.h"
#include "../orb-slam3_amd/facade/ORBmatcher.h"
#include "../orb-slam3_amd/facade/FrameGeometry.h"

#define CHECK(cond, code) do { if (!(cond)) { std::printf("facade_smoke: check failed at line %d: %s\n", __LINE__, #cond); return code; } } while (0)

struct MockKeyFrame;
struct MockFrame;

struct MockCamera {                                             // Pinhole (CameraModels/Pinhole.cpp:44-60)
    float fx = 435.2f, fy = 435.2f, cx = 367.4f, cy = 252.2f;
    cv::Point2f project(const cv::Point3f& p) const { return cv::Point2f(fx * p.x / p.z + cx, fy * p.y / p.z + cy); }
    cv::Point2f project(const cv::Mat& m) const { return project(cv::Point3f(m.at<float>(0), m.at<float>(1), m.at<float>(2))); }
    cv::Point2f project(const cv::Matx31f& m) const { return project(cv::Point3f(m(0), m(1), m(2))); }
    // GeometricCamera.h:82-85: GetType() and the two type constants are (non-static) members of the camera object
    unsigned int mnType = 0;
    unsigned int GetType() { return mnType; }
    const unsigned int CAM_PINHOLE = 0;
    const unsigned int CAM_FISHEYE = 1;
    // a non-pinhole mock (mnType = CAM_FISHEYE) answers with a test that is NOT an epipolar-line distance, so that the
    // outcome shows which gate the matcher consulted (KannalaBrandt8::epipolarConstrain_ triangulates, KannalaBrandt8.cpp:356-360)
    int gateCalls = 0;
    cv::Matx33f lastR12; cv::Matx31f lastT12;
    bool epipolarConstrain_(MockCamera*, const cv::KeyPoint& kp1, const cv::KeyPoint&, const cv::Matx33f& R12, const cv::Matx31f& t12, float, float) {
        ++gateCalls; lastR12 = R12; lastT12 = t12;
        return mnType == 0 ? true : kp1.pt.y < 240.f;
    }
    bool matchAndtriangulate(const cv::KeyPoint&, const cv::KeyPoint&, MockCamera*, cv::Mat&, cv::Mat&, float, float, cv::Mat& x3D) { x3D = cv::Mat(3, 1, CV_32F); return true; }
};

struct MockMapPoint {                                           // include/MapPoint.h
    cv::Mat pos, normal, desc;
    float minD = 0.1f, maxD = 100.f;
    bool bad = false;
    int nobs = 1;
    std::map<MockKeyFrame*, int> obs;
    bool mbTrackInView = false, mbTrackInViewR = false;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackProjYR = 0, mTrackDepth = 0, mTrackViewCos = 1, mTrackViewCosR = 1;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
    int level = 0;                                              // what PredictScale returns in this scene
    cv::Mat GetWorldPos() { return pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    cv::Mat GetDescriptor() { return desc.clone(); }
    bool isBad() { return bad; }
    int Observations() { return nobs; }
    float GetMinDistanceInvariance() { return minD; }
    float GetMaxDistanceInvariance() { return maxD; }
    int PredictScale(const float&, MockKeyFrame*) { return level; }
    int PredictScale(const float&, MockFrame*) { return level; }
    bool IsInKeyFrame(MockKeyFrame* kf) { return obs.count(kf) > 0; }
    std::tuple<int, int> GetIndexInKeyFrame(MockKeyFrame* kf) { auto it = obs.find(kf); return std::make_tuple(it == obs.end() ? -1 : it->second, -1); }
    void AddObservation(MockKeyFrame* kf, int idx) { obs[kf] = idx; ++nobs; }
    void Replace(MockMapPoint* other) { bad = true; replacedBy = other; }
    MockMapPoint* replacedBy = nullptr;
};

static cv::Mat eye4() { return cv::Mat::eye(4, 4, CV_32F); }

struct MockFrame {                                              // include/Frame.h
    static inline unsigned long nNextId = 0;                    // Frame.h: static long unsigned int nNextId; long unsigned int mnId
    unsigned long mnId = nNextId++;                             // (identifies the frame's immutable feature data: the facade keeps such frames in HBM)
    int N = 0, Nleft = -1;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    cv::Mat mDescriptors;
    std::vector<float> mvuRight, mvScaleFactors;
    std::vector<MockMapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<std::size_t> mGrid[ORBM_GRID_COLS][ORBM_GRID_ROWS], mGridRight[ORBM_GRID_COLS][ORBM_GRID_ROWS];
    float mnMinX = 0, mnMaxX = 752, mnMinY = 0, mnMaxY = 480, mfGridElementWidthInv = 64.f / 752.f, mfGridElementHeightInv = 48.f / 480.f;
    cv::Mat mTcw = eye4(), mTrl = eye4();
    MockCamera* mpCamera = nullptr;
    float mb = 0.11f, mbf = 47.9f;
    std::map<unsigned, std::vector<unsigned>> mFeatVec;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
};

struct MockKeyFrame {                                           // include/KeyFrame.h
    int N = 0, NLeft = -1;
    std::vector<cv::KeyPoint> mvKeysUn, mvKeys, mvKeysRight;
    cv::Mat mDescriptors;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    std::map<unsigned, std::vector<unsigned>> mFeatVec;
    float fx = 435.2f, fy = 435.2f, cx = 367.4f, cy = 252.2f, mbf = 47.9f;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 752, mnMaxY = 480;
    float mfGridElementWidthInv = 64.f / 752.f, mfGridElementHeightInv = 48.f / 480.f;
    MockCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    std::vector<MockMapPoint*> mps;
    cv::Mat Tcw = eye4();
    std::vector<MockMapPoint*> GetMapPointMatches() { return mps; }
    MockMapPoint* GetMapPoint(int i) { return mps[i]; }
    std::set<MockMapPoint*> GetMapPoints() { std::set<MockMapPoint*> s; for (auto* p : mps) if (p) s.insert(p); return s; }
    void AddMapPoint(MockMapPoint* p, int i) { mps[i] = p; }
    cv::Mat GetPose() { return Tcw.clone(); }
    cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
    cv::Mat GetCameraCenter() { return -GetRotation().t() * GetTranslation(); }
    cv::Mat GetRightPose() { return GetPose(); }
    cv::Mat GetRightRotation() { return GetRotation(); }
    cv::Mat GetRightTranslation() { return GetTranslation(); }
    cv::Mat GetRightCameraCenter() { return GetCameraCenter(); }
    cv::Matx33f GetRotation_() { cv::Matx33f R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = Tcw.at<float>(i, j); return R; }
    cv::Matx31f GetTranslation_() { cv::Matx31f t; for (int i = 0; i < 3; ++i) t(i) = Tcw.at<float>(i, 3); return t; }
    cv::Matx31f GetCameraCenter_() { return -GetRotation_().t() * GetTranslation_(); }
    cv::Matx33f GetRightRotation_() { return GetRotation_(); }
    cv::Matx31f GetRightTranslation_() { return GetTranslation_(); }
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }   // KeyFrame.cc: IsInImage
};

typedef ORB_SLAM3::ORBmatcher Matcher;

int main(int argc, char** argv) {
    const int w = 752, h = 480;
    std::vector<uint8_t> buf((size_t)w * h);
    unsigned s = 12345;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) { s = s * 1664525u + 1013904223u; buf[(size_t)y * w + x] = (uint8_t)(((x / 16 + y / 16) & 1) * 120 + 60 + (s >> 28)); }
    cv::Mat img(h, w, CV_8U, buf.data());
    if (orbx_device_count() < 1) { std::printf("facade compiled; no GPU here\n"); return argc > 1 ? 1 : 0; }
    ORB_SLAM3::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps; cv::Mat desc; std::vector<int> lap = {0, 1000};
    const int mono = ex(img, cv::Mat(), kps, desc, lap);
    std::printf("facade: %zu keypoints, mono=%d, desc %dx%d, levels=%d sf1=%.3f\n", kps.size(), mono, desc.rows, desc.cols,
                ex.GetLevels(), ex.GetScaleFactors()[1]);
    CHECK(!kps.empty() && desc.rows == (int)kps.size(), 2);
    CHECK(ex.mvImagePyramid[1].cols == 627 && ex.mvImagePyramid[0].rows == 480, 3);   // filled by operator() itself (Frame.cc:1168 slices it)
    CHECK(ORB_SLAM3::ORBmatcher::DescriptorDistance(desc.row(0), desc.row(0)) == 0, 4);
    {   // the reference builds an ORBmatcher on the stack at every call site: constructing one must cost next to nothing
        { Matcher warm(0.7f); }
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 1000; ++i) { Matcher m1(0.9f, true); (void)m1.handle(); }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("facade: 1000 ORBmatcher constructions + destructions in %.3f ms\n", ms);
        CHECK(ms < 1.0, 20);
    }
    Matcher m(0.7f);
    {   // Frame-level helpers: undistorted keypoints stay put at the principal point, bounds grow for k1 < 0
        const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
        const std::vector<float> D = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
        std::vector<cv::KeyPoint> un;
        ORB_SLAM3::UndistortKeyPoints(m.handle(), kps, K, D, K, un);
        float x0, x1, y0, y1;
        ORB_SLAM3::ComputeImageBounds(m.handle(), w, h, K, D, K, x0, x1, y0, y1);
        CHECK(un.size() == kps.size() && x0 < 0.f && x1 > (float)w && y0 < 0.f && y1 > (float)h, 5);
        ORB_SLAM3::FrustumOut fo;
        const float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, O[3] = {0, 0, 0}, b[4] = {x0, x1, y0, y1};
        const int nin = ORB_SLAM3::IsInFrustumBatch(m.handle(), {0.f, 0.f, 4.f}, {0.f, 0.f, 1.f}, {0.5f}, {6.f}, R, t, O, K, b, 47.9f, 0.5f, 0.18232156f, 8, fo);
        CHECK(nin == 1 && fo.level[0] >= 0, 6);
    }

    // ---------------------------------------------------------------------------------------------------------------
    // the scene: every keypoint of the frame carries a MapPoint 2..6 m in front of an identity-pose pinhole camera
    // ---------------------------------------------------------------------------------------------------------------
    const int n = (int)kps.size();
    MockCamera cam;
    std::vector<float> sf(8), sig2(8), isig2(8);
    sf[0] = 1.f; for (int i = 1; i < 8; ++i) sf[i] = sf[i - 1] * 1.2f;
    for (int i = 0; i < 8; ++i) { sig2[i] = sf[i] * sf[i]; isig2[i] = 1.f / sig2[i]; }
    std::vector<MockMapPoint> pts(n);
    for (int i = 0; i < n; ++i) {
        const float z = 2.f + 4.f * (float)((i * 37) % 101) / 101.f;
        MockMapPoint& p = pts[i];
        p.pos = cv::Mat(3, 1, CV_32F);
        p.pos.at<float>(0) = (kps[i].pt.x - cam.cx) * z / cam.fx; p.pos.at<float>(1) = (kps[i].pt.y - cam.cy) * z / cam.fy; p.pos.at<float>(2) = z;
        p.normal = cv::Mat(3, 1, CV_32F);
        const float nn = (float)cv::norm(p.pos);
        for (int k = 0; k < 3; ++k) p.normal.at<float>(k) = p.pos.at<float>(k) / nn;
        p.desc = desc.row(i).clone();
        p.level = kps[i].octave;
    }
    auto fillGrid = [&](std::vector<std::size_t> (*G)[ORBM_GRID_ROWS], const std::vector<cv::KeyPoint>& K) {
        std::vector<int32_t> gs(ORBM_GRID_COLS * ORBM_GRID_ROWS + 1), gi(K.size() + 1);
        if (orbm_grid_build(m.handle(), (const orbm_kp_t*)K.data(), (int)K.size(), 0.f, 0.f, 64.f / 752.f, 48.f / 480.f, gs.data(), gi.data()) < 0) return false;
        for (int ix = 0; ix < ORBM_GRID_COLS; ++ix) for (int iy = 0; iy < ORBM_GRID_ROWS; ++iy) {
            G[ix][iy].clear();
            for (int k = gs[ix * ORBM_GRID_ROWS + iy]; k < gs[ix * ORBM_GRID_ROWS + iy + 1]; ++k) G[ix][iy].push_back((std::size_t)gi[k]);
        }
        return true;
    };
    auto makeFrame = [&](MockFrame& F) {
        F.N = n; F.mvKeys = kps; F.mvKeysUn = kps; F.mDescriptors = desc; F.mvuRight.assign(n, -1.f); F.mvScaleFactors = sf;
        F.mvpMapPoints.assign(n, nullptr); F.mvbOutlier.assign(n, false); F.mpCamera = &cam;
        for (int i = 0; i < n; ++i) F.mFeatVec[desc.ptr(i)[0] & 31].push_back(i);
        return fillGrid(F.mGrid, F.mvKeysUn);
    };
    auto makeKF = [&](MockKeyFrame& K, bool withPoints) {
        K.N = n; K.mvKeysUn = kps; K.mvKeys = kps; K.mDescriptors = desc; K.mvuRight.assign(n, -1.f); K.mvScaleFactors = sf; K.mvLevelSigma2 = sig2; K.mvInvLevelSigma2 = isig2;
        K.mpCamera = &cam; K.mps.assign(n, nullptr);
        for (int i = 0; i < n; ++i) { K.mFeatVec[desc.ptr(i)[0] & 31].push_back(i); if (withPoints) K.mps[i] = &pts[i]; }
    };
    auto resetPoints = [&] { for (auto& p : pts) { p.bad = false; p.nobs = 1; p.obs.clear(); p.replacedBy = nullptr; } };

    {   // ---- M4 SearchByProjection(Frame, Frame): identity motion -> every last-frame MapPoint lands on its own keypoint
        MockFrame last, cur;
        CHECK(makeFrame(last) && makeFrame(cur), 30);
        for (int i = 0; i < n; ++i) last.mvpMapPoints[i] = &pts[i];
        last.mvbOutlier[3] = true;
        MockMapPoint stale; stale.nobs = 0;                    // a leftover with no observations in a slot the rotation check may cull
        cur.mvpMapPoints[5] = &stale;
        Matcher mm(0.9f, true);
        const int got = mm.SearchByProjection(cur, last, 7.f, true);
        int assigned = 0, own = 0;
        for (int i = 0; i < n; ++i) if (cur.mvpMapPoints[i] && cur.mvpMapPoints[i] != &stale) { ++assigned; own += cur.mvpMapPoints[i]->desc.data[0] == desc.ptr(i)[0]; }
        std::printf("facade: M4 SearchByProjection(Frame,Frame) %d matches, %d slots assigned\n", got, assigned);
        CHECK(got > n / 2 && assigned == got && own >= got * 9 / 10, 31);
        CHECK(cur.mvpMapPoints[3] != &pts[3] || kps[3].pt.x == kps[2].pt.x, 32);
    }
    {   // ---- M3 SearchByProjection(Frame, vpMapPoints): the local-map search, compared with the flattened C call
        resetPoints();
        MockFrame F;
        CHECK(makeFrame(F), 33);
        std::vector<MockMapPoint*> vp;
        for (int i = 0; i < n; ++i) {
            MockMapPoint& p = pts[i];
            p.mbTrackInView = (i % 7) != 0; p.mTrackProjX = kps[i].pt.x; p.mTrackProjY = kps[i].pt.y; p.mTrackProjXR = kps[i].pt.x - 3.f; p.mTrackDepth = p.pos.at<float>(2);
            p.mTrackViewCos = (i % 3) ? 0.9995f : 0.9f; p.mnTrackScaleLevel = kps[i].octave;
            vp.push_back(&p);
        }
        pts[11].bad = true;
        Matcher mm(0.8f);
        const int got = mm.SearchByProjection(F, vp, 3.f, true, 5.5f);
        // the same search through the flattened entry point
        std::vector<uint8_t> inv(n), obs(n, 1), blocked(n, 0), qd((size_t)n * 32);
        std::vector<float> px(n), py(n), pxr(n), vc(n); std::vector<int32_t> lv(n), match(n);
        for (int i = 0; i < n; ++i) {
            inv[i] = pts[i].mbTrackInView && !(pts[i].mTrackDepth > 5.5f) && !pts[i].bad;
            px[i] = pts[i].mTrackProjX; py[i] = pts[i].mTrackProjY; pxr[i] = pts[i].mTrackProjXR; vc[i] = pts[i].mTrackViewCos; lv[i] = pts[i].mnTrackScaleLevel;
            std::memcpy(&qd[(size_t)i * 32], desc.ptr(i), 32);
        }
        MockFrame F2; CHECK(makeFrame(F2), 34);
        Matcher::View<MockFrame> view(F2);
        const int want = orbm_search_by_projection_points(m.handle(), &view.f, blocked.data(), sf.data(), n, inv.data(), px.data(), py.data(), pxr.data(), vc.data(), lv.data(),
                                                          qd.data(), obs.data(), 3.f, 0.8f, match.data());
        std::printf("facade: M3 SearchByProjection(Frame,MapPoints) %d matches (flattened call: %d)\n", got, want);
        CHECK(got == want && got > n / 3, 35);
        for (int k = 0; k < n; ++k) CHECK((match[k] >= 0 ? &pts[match[k]] : nullptr) == F.mvpMapPoints[k], 36);
        // the same Frame again (now resident in HBM: only the queries travel) and a copy of it (same mnId, as mLastFrame = Frame(mCurrentFrame))
        for (auto& q : F.mvpMapPoints) q = nullptr;
        CHECK(mm.SearchByProjection(F, vp, 3.f, true, 5.5f) == want, 37);
        for (int k = 0; k < n; ++k) CHECK((match[k] >= 0 ? &pts[match[k]] : nullptr) == F.mvpMapPoints[k], 38);
        MockFrame F3 = F;
        for (auto& q : F3.mvpMapPoints) q = nullptr;
        CHECK(mm.SearchByProjection(F3, vp, 3.f, true, 5.5f) == want, 39);
        for (auto& p : pts) p.mbTrackInView = false;
    }
    {   // ---- M5 SearchByProjection(Frame, KeyFrame, sAlreadyFound, th, ORBdist): relocalisation
        resetPoints();
        MockFrame F; MockKeyFrame K;
        CHECK(makeFrame(F), 37); makeKF(K, true);
        std::set<MockMapPoint*> found = {&pts[0], &pts[1]};
        Matcher mm(0.9f, true);
        const int got = mm.SearchByProjection(F, &K, found, 10.f, 100);
        int assigned = 0; for (auto* p : F.mvpMapPoints) assigned += p != nullptr;
        std::printf("facade: M5 SearchByProjection(Frame,KeyFrame) %d matches\n", got);
        CHECK(got > n / 2 && assigned == got, 38);
        for (auto* p : F.mvpMapPoints) CHECK(p != &pts[0] && p != &pts[1], 39);
    }
    {   // ---- M6 both Sim3 overloads: Scw = identity Sim3
        resetPoints();
        MockKeyFrame K; makeKF(K, false);
        std::vector<MockMapPoint*> vp, matched(n, nullptr), matched2(n, nullptr);
        std::vector<MockKeyFrame*> vpKFs, matchedKF(n, nullptr);
        MockKeyFrame other;
        for (int i = 0; i < n; ++i) { vp.push_back(&pts[i]); vpKFs.push_back(&other); }
        matched[4] = &pts[4]; matched2[4] = &pts[4];
        Matcher mm(0.75f, true);
        const int a = mm.SearchByProjection(&K, eye4(), vp, matched, 3, 1.5f);
        const int b = mm.SearchByProjection(&K, eye4(), vp, vpKFs, matched2, matchedKF, 3, 1.5f);
        int na = 0, nb = 0;
        for (int i = 0; i < n; ++i) { na += matched[i] != nullptr; nb += matched2[i] != nullptr; CHECK((matchedKF[i] == &other) == (matched2[i] != nullptr && i != 4), 40); }
        std::printf("facade: M6 SearchByProjection(KeyFrame,Scw) %d / %d new matches\n", a, b);
        CHECK(a > n / 2 && na == a + 1 && b == a && nb == na, 41);
    }
    {   // ---- M7 / M8 SearchByBoW
        resetPoints();
        MockFrame F; MockKeyFrame K1, K2;
        CHECK(makeFrame(F), 42); makeKF(K1, true); makeKF(K2, true);
        std::vector<MockMapPoint*> fm, m12;
        Matcher mm(0.75f, true);
        const int a = mm.SearchByBoW(&K1, F, fm);
        const int b = mm.SearchByBoW(&K1, &K2, m12);
        int na = 0, nb = 0; for (auto* p : fm) na += p != nullptr; for (auto* p : m12) nb += p != nullptr;
        std::printf("facade: M7 SearchByBoW(KeyFrame,Frame) %d, M8 SearchByBoW(KeyFrame,KeyFrame) %d\n", a, b);
        CHECK((int)fm.size() == n && a == na && a > n / 2 && (int)m12.size() == n && b == nb && b > n / 2, 43);
    }
    {   // ---- M9 SearchForInitialization
        MockFrame F1, F2;
        CHECK(makeFrame(F1) && makeFrame(F2), 44);
        std::vector<cv::Point2f> prev(n);
        for (int i = 0; i < n; ++i) prev[i] = F1.mvKeysUn[i].pt;
        std::vector<int> m12;
        Matcher mm(0.9f, true);
        const int got = mm.SearchForInitialization(F1, F2, prev, m12, 100);
        int cnt = 0; for (int i = 0; i < n; ++i) if (m12[i] >= 0) { ++cnt; CHECK(kps[m12[i]].octave == 0 && kps[i].octave == 0, 45); }
        std::printf("facade: M9 SearchForInitialization %d matches\n", got);
        CHECK((int)m12.size() == n && cnt == got && got > 50, 46);
    }
    {   // ---- M10 SearchForTriangulation_(cv::Matx33f F12) and M11 (cv::Mat F12): compared with the flattened call
        resetPoints();
        MockKeyFrame K1, K2; makeKF(K1, false); makeKF(K2, false);
        K2.Tcw.at<float>(0, 3) = -0.2f;                        // baseline along x: F12 = [t]x for identical intrinsics and R = I
        cv::Matx33f F12; F12(1, 2) = 0.2f; F12(2, 1) = -0.2f;
        cv::Mat F12m(3, 3, CV_32F); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F12m.at<float>(i, j) = F12(i, j);
        std::vector<std::pair<size_t, size_t>> pairs, pairsL;
        Matcher mm(0.6f, false);
        const int a = mm.SearchForTriangulation_(&K1, &K2, F12, pairs, false, true);
        const int b = mm.SearchForTriangulation(&K1, &K2, F12m, pairsL, false, true);
        // flattened: the epipole is C2 = R2w * Cw1 + t2w projected by camera 2 (ORBmatcher.cc:1393-1400)
        const cv::Point2f ep = cam.project(cv::Point3f(-0.2f, 0.f, 0.f));
        ORB_SLAM3::facade_detail::FlatFeatVec<decltype(K1.mFeatVec)> f1(K1.mFeatVec), f2(K2.mFeatVec);
        std::vector<uint8_t> mp(n, 0); std::vector<int32_t> want(n, -1);
        float Fr[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fr[3 * i + j] = F12(i, j);
        const int c = orbm_search_for_triangulation(m.handle(), n, (const orbm_kp_t*)kps.data(), desc.data, mp.data(), K1.mvuRight.data(), (int)f1.nodes.size(), f1.nodes.data(),
                                                    f1.start.data(), f1.idx.data(), n, (const orbm_kp_t*)kps.data(), desc.data, mp.data(), K2.mvuRight.data(), (int)f2.nodes.size(),
                                                    f2.nodes.data(), f2.start.data(), f2.idx.data(), Fr, ep.x, ep.y, sf.data(), sig2.data(), 0, 1, 0, want.data());
        std::printf("facade: M10 SearchForTriangulation_ %d (flattened %d), M11 legacy overload %d\n", a, c, b);
        CHECK(a == c && a > n / 2 && (int)pairs.size() == a && b > n / 2 && (int)pairsL.size() == b, 47);
        for (auto& pr : pairs) CHECK(want[pr.first] == (int)pr.second, 48);
    }
    {   // ---- M10 on single-camera KeyFrames whose camera is NOT a pinhole (monocular fisheye, Examples/Monocular/TUM_512.yaml):
        //      the reference's gate is the virtual pCamera1->epipolarConstrain_ (ORBmatcher.cc:1555), not the F12 line test
        resetPoints();
        MockCamera fish; fish.mnType = fish.CAM_FISHEYE;
        MockKeyFrame K1, K2; makeKF(K1, false); makeKF(K2, false);
        K1.mpCamera = &fish; K2.mpCamera = &fish;
        K2.Tcw.at<float>(0, 3) = -0.2f;
        cv::Matx33f F12; F12(1, 2) = 0.2f; F12(2, 1) = -0.2f;    // what the line test would use: it accepts every pair of this scene (previous block)
        std::vector<std::pair<size_t, size_t>> pairs, pairsC, pairsS;
        Matcher mm(0.6f, false);
        const int got = mm.SearchForTriangulation_(&K1, &K2, F12, pairs, false, false);
        CHECK(fish.gateCalls > 0, 60);                           // the camera object was consulted ...
        CHECK(got > 0 && (int)pairs.size() == got, 61);
        int upper = 0; for (int i = 0; i < n; ++i) upper += kps[i].pt.y < 240.f;
        for (auto& pr : pairs) CHECK(K1.mvKeysUn[pr.first].pt.y < 240.f, 62);   // ... and its answer decided (the line test passes the lower half too)
        CHECK(got <= upper && got < n, 63);
        CHECK(fish.lastR12(0, 0) == 1.f && fish.lastR12(1, 1) == 1.f && fish.lastT12(0) == 0.2f && fish.lastT12(1) == 0.f, 64);   // R12 = R1w R2w^T, t12 = -R1w R2w^T t2w + t1w (:1414-1415)
        // the same search through the flattened gated entry point with the epipole + camera gate written out
        const cv::Point2f ep = fish.project(cv::Point3f(-0.2f, 0.f, 0.f));
        const int want = mm.SearchForTriangulationGated(&K1, &K2, [&](int i1, int i2) {
            const float dx = ep.x - kps[i2].pt.x, dy = ep.y - kps[i2].pt.y;
            if (dx * dx + dy * dy < 100 * sf[kps[i2].octave]) return false;
            return kps[i1].pt.y < 240.f; }, pairsS);
        CHECK(want == got && pairsS == pairs, 65);
        const int calls0 = fish.gateCalls;
        const int coarse = mm.SearchForTriangulation_(&K1, &K2, F12, pairsC, false, true);   // || bCoarse (:1555): the camera is still asked first
        CHECK(fish.gateCalls > calls0 && coarse >= got && coarse > n / 2, 66);
        const int only = mm.SearchForTriangulation_(&K1, &K2, F12, pairsC, true, true);      // bOnlyStereo on monocular KeyFrames: every feature skipped (:1464-1466)
        CHECK(only == 0 && pairsC.empty(), 67);
        std::printf("facade: M10 non-pinhole single camera: %d matches through the camera's own gate (%d gate calls), %d coarse\n", got, fish.gateCalls, coarse);
    }
    {   // ---- M10 with two cameras / M12: the gated bucket search
        MockKeyFrame a, b;
        const int nl = n / 2;
        for (MockKeyFrame* kf : {&a, &b}) {
            makeKF(*kf, false);
            kf->NLeft = nl; kf->mpCamera2 = &cam;
            kf->mvKeys.assign(kps.begin(), kps.begin() + nl); kf->mvKeysRight.assign(kps.begin() + nl, kps.end());
        }
        std::vector<std::pair<size_t, size_t>> pairs;
        int calls = 0;
        Matcher mt(0.6f, false);
        const int all = mt.SearchForTriangulationGated(&a, &b, [&](int, int) { ++calls; return true; }, pairs);
        CHECK(all == n && (int)pairs.size() == n, 7);          // distance 0 to itself: every feature keeps a match
        for (auto& pr : pairs) CHECK(orbm_hamming(desc.ptr((int)pr.first), desc.ptr((int)pr.second)) == 0, 8);
        const int off = mt.SearchForTriangulationGated(&a, &b, [&](int i1, int i2) { return i1 != i2; }, pairs);
        for (auto& pr : pairs) CHECK(pr.first != pr.second, 9);
        cv::Matx33f F12;
        const int two = mt.SearchForTriangulation_(&a, &b, F12, pairs, false, false);       // routed to the two-camera gate (mpCamera2 set)
        std::vector<cv::Mat> x3D;
        const int tri = mt.SearchForTriangulation(&a, &b, cv::Mat(), pairs, false, x3D);
        std::printf("facade: gated triangulation search %d / %d matches, %d gate calls; two-camera %d, +vMatchedPoints %d\n", all, off, calls, two, tri);
        CHECK(two == n && tri == n && (int)x3D.size() == n, 10);
    }
    {   // ---- M13 SearchBySim3: identity Sim3 between two copies of the KeyFrame
        resetPoints();
        MockKeyFrame K1, K2; makeKF(K1, true); makeKF(K2, true);
        std::vector<MockMapPoint> pts2(pts);                    // KeyFrame 2 observes its own MapPoint objects
        for (int i = 0; i < n; ++i) K2.mps[i] = &pts2[i];
        std::vector<MockMapPoint*> m12(n, nullptr);
        m12[2] = &pts2[2]; pts2[2].obs[&K2] = 2;
        cv::Mat R12 = cv::Mat::eye(3, 3, CV_32F), t12 = cv::Mat::zeros(3, 1, CV_32F);
        Matcher mm(0.75f, true);
        const float s12 = 1.f;
        const int got = mm.SearchBySim3(&K1, &K2, m12, s12, R12, t12, 7.5f);
        int cnt = 0; for (int i = 0; i < n; ++i) if (m12[i]) { ++cnt; CHECK(m12[i] >= &pts2[0] && m12[i] <= &pts2[n - 1], 49); }
        std::printf("facade: M13 SearchBySim3 %d new matches\n", got);
        CHECK(got > n / 2 && cnt == got + 1, 50);
    }
    {   // ---- M13 Fuse (pose) and Fuse (Sim3)
        resetPoints();
        MockKeyFrame K; makeKF(K, false);
        std::vector<MockMapPoint> mine(pts);                     // the KeyFrame already holds its own MapPoints on every third feature
        for (int i = 0; i < n; i += 3) { K.mps[i] = &mine[i]; mine[i].obs[&K] = i; mine[i].nobs = 5; }
        std::vector<MockMapPoint*> vp;
        for (int i = 0; i < n; ++i) vp.push_back(&pts[i]);
        vp.push_back(nullptr);
        Matcher mm(0.6f, true);
        const int fused = mm.Fuse(&K, vp, 3.0f, false);
        int added = 0, replaced = 0;
        for (int i = 0; i < n; ++i) { added += pts[i].obs.count(&K) > 0; replaced += pts[i].bad; }
        std::printf("facade: M13 Fuse %d fused (%d added to the KeyFrame, %d replaced by its own points)\n", fused, added, replaced);
        CHECK(fused > n / 2 && added + replaced == fused && replaced > 0 && added > 0, 51);
        resetPoints();
        MockKeyFrame K2; makeKF(K2, false);
        for (int i = 0; i < n; i += 3) K2.mps[i] = &mine[i];
        std::vector<MockMapPoint*> vp2, repl(n, nullptr);
        for (int i = 0; i < n; ++i) vp2.push_back(&pts[i]);
        const int fused2 = mm.Fuse(&K2, eye4(), vp2, 4.0f, repl);
        int nrep = 0, nadd = 0; for (int i = 0; i < n; ++i) { nrep += repl[i] != nullptr; nadd += pts[i].obs.count(&K2) > 0; }
        std::printf("facade: M13 Fuse(Scw) %d fused (%d to replace, %d added)\n", fused2, nrep, nadd);
        CHECK(fused2 > n / 2 && nrep + nadd == fused2 && nrep > 0, 52);
    }
    return 0;
}
