// Latency of the drop-in C++ facade (ORB_SLAM3::ORBextractor::operator()) on one 752x480 frame, median of 300 calls:
//   default (mvImagePyramid brought back to the host after every call) and KeepPyramidOnDevice(true).
// build: g++ -std=c++17 -O2 -o tools/facade_latency_bin tools/facade_latency.cpp -Lorb-slam3_amd -lorbslam3_amd -Wl,-rpath,$PWD/orb-slam3_amd
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <random>
#include "../orb-slam3_amd/facade/ORBextractor.h"

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int W = 752, H = 480;
    cv::Mat img(H, W, CV_8U);
    std::mt19937 rng(7);
    // blocky texture with noise: plenty of corners
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const int b = ((x / 23) * 37 + (y / 19) * 91) % 200;
        img.data[(size_t)y * img.step + x] = (uint8_t)std::min(255, b + (int)(rng() % 24));
    }
    ORB_SLAM3::ORBextractor ex(1200, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps; cv::Mat desc; std::vector<int> lap = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {
        ex.KeepPyramidOnDevice(mode == 1);
        for (int i = 0; i < 10; ++i) ex(img, cv::Mat(), kps, desc, lap);
        std::vector<double> t;
        for (int i = 0; i < 300; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            ex(img, cv::Mat(), kps, desc, lap);
            t.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%s: %.3f ms per frame (%zu keypoints)\n", mode ? "operator(), pyramid kept on the device" : "operator(), mvImagePyramid on the host (default)", med(t), kps.size());
    }
    return 0;
}
