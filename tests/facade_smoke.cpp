// Compiles the C++ facade against the minimal cv-compat layer and (on a GPU box) runs one extraction through it.
#include <cstdio>
#include <map>
#include <vector>
#include "../orb-slam3_amd/facade/ORBextractor.h"
#include "../orb-slam3_amd/facade/ORBmatcher.h"
#include "../orb-slam3_amd/facade/FrameGeometry.h"

// the members of ORB_SLAM3::KeyFrame the triangulation searches read (include/KeyFrame.h)
struct MockKeyFrame {
    int N = 0, NLeft = -1;
    std::vector<cv::KeyPoint> mvKeysUn, mvKeys, mvKeysRight;
    cv::Mat mDescriptors;
    std::map<unsigned, std::vector<unsigned>> mFeatVec;
    std::vector<void*> mps;
    void* GetMapPoint(int i) { return mps[i]; }
};

int main(int argc, char** argv) {
    const int w = 752, h = 480;
    std::vector<uint8_t> buf((size_t)w * h);
    unsigned s = 12345;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) { s = s * 1664525u + 1013904223u; buf[(size_t)y * w + x] = (uint8_t)(((x / 16 + y / 16) & 1) * 120 + 60 + (s >> 28)); }
    cv::Mat img(h, w, cv::CV_8U, buf.data());
    if (orbx_device_count() < 1) { std::printf("facade compiled; no GPU here\n"); return argc > 1 ? 1 : 0; }
    ORB_SLAM3::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps; cv::Mat desc; std::vector<int> lap = {0, 1000};
    const int mono = ex(img, cv::Mat(), kps, desc, lap);
    std::printf("facade: %zu keypoints, mono=%d, desc %dx%d, levels=%d sf1=%.3f\n", kps.size(), mono, desc.rows, desc.cols,
                ex.GetLevels(), ex.GetScaleFactors()[1]);
    if (kps.empty() || desc.rows != (int)kps.size()) return 2;
    ex.FetchImagePyramid();
    if (ex.mvImagePyramid[1].cols != 627) return 3;
    ORB_SLAM3::ORBmatcher m(0.7f);
    if (ORB_SLAM3::ORBmatcher::DescriptorDistance(desc.row(0), desc.row(0)) != 0) return 4;
    {   // Frame-level helpers: undistorted keypoints stay put at the principal point, bounds grow for k1 < 0
        const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
        const std::vector<float> D = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
        std::vector<cv::KeyPoint> un;
        ORB_SLAM3::UndistortKeyPoints(m.handle(), kps, K, D, K, un);
        float x0, x1, y0, y1;
        ORB_SLAM3::ComputeImageBounds(m.handle(), w, h, K, D, K, x0, x1, y0, y1);
        if (un.size() != kps.size() || !(x0 < 0.f && x1 > (float)w && y0 < 0.f && y1 > (float)h)) return 5;
        ORB_SLAM3::FrustumOut fo;
        const float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, O[3] = {0, 0, 0}, b[4] = {x0, x1, y0, y1};
        const int nin = ORB_SLAM3::IsInFrustumBatch(m.handle(), {0.f, 0.f, 4.f}, {0.f, 0.f, 1.f}, {0.5f}, {6.f}, R, t, O, K, b, 47.9f, 0.5f, 0.18232156f, 8, fo);
        if (nin != 1 || fo.level[0] < 0) return 6;
    }
    {   // M10-with-two-cameras / M12 bucket search: a frame matched against itself under a gate that forbids the identity pair
        MockKeyFrame a, b;
        const int n = (int)kps.size();
        const int nl = n / 2;
        for (MockKeyFrame* kf : {&a, &b}) {
            kf->N = n; kf->NLeft = nl; kf->mDescriptors = desc; kf->mps.assign(n, nullptr);
            kf->mvKeys.assign(kps.begin(), kps.begin() + nl); kf->mvKeysRight.assign(kps.begin() + nl, kps.end());
            for (int i = 0; i < n; ++i) kf->mFeatVec[desc.ptr(i)[0] & 15].push_back(i);
        }
        std::vector<std::pair<size_t, size_t>> pairs;
        int calls = 0;
        ORB_SLAM3::ORBmatcher mt(0.6f, false);
        const int all = mt.SearchForTriangulationGated(&a, &b, [&](int, int) { ++calls; return true; }, pairs);
        if (all != n || (int)pairs.size() != n) return 7;             // distance 0 to itself: every feature keeps a match
        for (auto& pr : pairs) if (orbm_hamming(desc.ptr((int)pr.first), desc.ptr((int)pr.second)) != 0) return 8;
        const int off = mt.SearchForTriangulationGated(&a, &b, [&](int i1, int i2) { return i1 != i2; }, pairs);
        for (auto& pr : pairs) if (pr.first == pr.second) return 9;
        std::printf("facade: gated triangulation search %d / %d matches, %d gate calls\n", all, off, calls);
    }
    return 0;
}
