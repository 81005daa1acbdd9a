// ORBextractor.h -- drop-in C++ facade: the reference's class API (include/ORBextractor.h:49-83) over the C ABI of
// liborbslam3_amd.so (include/orbx.h).  Replace `#include "ORBextractor.h"` of the reference with this header,
// delete src/ORBextractor.cc from the build and link liborbslam3_amd.so: Frame.cc / Tracking.cc compile unchanged
// (same constructor, operator(), getters and public mvImagePyramid).  See INTEGRATION.md.
#pragma once
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cvcompat.h"
#endif
#include "../../include/orbx.h"

namespace ORB_SLAM3 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // ORBextractor.cc:468-571.  The device comes from ORBX_DEVICE (default 0); the device-resident pyramid is sized
    // lazily from the first image (the reference allocates per call, ORBextractor.cc:1669-1672).
    ORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_)
        : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_) {
        const char* d = std::getenv("ORBX_DEVICE");
        device = d ? std::atoi(d) : 0;
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { if (h) orbx_destroy(h); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // ORBextractor.cc:1534-1659: returns monoIndex, -1 for an empty image.  `_mask` is ignored, as in the reference.
    int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                   cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
        const cv::Mat image = _image.getMat();
        if (image.empty()) return -1;
        ensure(image.cols, image.rows);
        const int cap = orbx_max_keypoints(h);
        kpbuf.resize(cap); descbuf.resize((size_t)cap * 32);
        int mono = 0;
        const int n = orbx_extract(h, image.data, image.cols, image.rows, (int)image.step, vLappingArea[0], vLappingArea[1],
                                   kpbuf.data(), descbuf.data(), cap, &mono);
        if (n < 0) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error());
        if (n == 0) _descriptors.release(); else _descriptors.create(n, 32, CV_8U);
        _keypoints.resize(n);
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_kp_t), "KeyPoint layout");
        if (n) {
            std::memcpy((void*)_keypoints.data(), kpbuf.data(), sizeof(orbx_kp_t) * n);
            cv::Mat d = _descriptors.getMat();
            for (int i = 0; i < n; ++i) std::memcpy(d.ptr(i), descbuf.data() + (size_t)i * 32, 32);
        }
        pyramidStale = true;
        if (hostPyramid) FetchImagePyramid();
        return mono;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return table(0); }
    std::vector<float> inline GetInverseScaleFactors() { return table(1); }
    std::vector<float> inline GetScaleSigmaSquares() { return table(2); }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return table(3); }

    // include/ORBextractor.h:83 -- Frame::ComputeStereoMatches slices these on the host (Frame.cc:1168,1194), so by default
    // every operator() brings the levels back (one copy of the pyramid slab out of HBM into pinned memory, orbx_pyramid_map; the Mats are headers over it).  An integration whose stereo
    // association runs on the device (orbm_stereo_matches reads the levels in HBM) switches that off with
    // KeepPyramidOnDevice(true) and calls FetchImagePyramid() only if it ever needs the pixels.
    std::vector<cv::Mat> mvImagePyramid;
    void KeepPyramidOnDevice(bool on) { hostPyramid = !on; }
    void FetchImagePyramid() {
        if (!h || !pyramidStale) return;
        // zero-copy: the levels are cv::Mat headers over the handle's pinned memory (one copy out of HBM for all of them), valid until
        // the next operator() -- the reference's extractor overwrites mvImagePyramid on every call too
        std::vector<const uint8_t*> ptr(nlevels);
        std::vector<int> pitch(nlevels);
        if (orbx_pyramid_map(h, 0, ptr.data(), pitch.data()) < 0)
            throw std::runtime_error(std::string("orbx_pyramid_map: ") + orbx_last_error());
        for (int l = 0; l < nlevels; ++l) {
            int w = 0, hh = 0;
            orbx_level_size(h, l, &w, &hh);
            if (ptr[l]) mvImagePyramid[l] = cv::Mat(hh, w, CV_8U, (void*)ptr[l], (size_t)pitch[l]);
            else {                                               // (device-resident level 0: copy)
                mvImagePyramid[l].create(hh, w, CV_8U);
                if (orbx_level_image(h, 0, l, 0, mvImagePyramid[l].data, (int)mvImagePyramid[l].step) < 0)
                    throw std::runtime_error(std::string("orbx_level_image: ") + orbx_last_error());
            }
        }
        pyramidStale = false;
    }
    orbx_t* handle() { return h; }          // for orbm_stereo_matches

protected:
    void ensure(int w, int hgt) {
        if (h && w <= maxW && hgt <= maxH) return;
        if (h) orbx_destroy(h);
        h = nullptr;
        maxW = w; maxH = hgt;
        if (orbx_create(&h, nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, device, w, hgt, 1) != ORBX_OK)
            throw std::runtime_error(std::string("orbx_create: ") + orbx_last_error());   // no CPU fallback
        // per-stage GPU timings are a profiling aid: without them orbx_extract replays the whole frame as one graph from the third
        // call on (one launch instead of ~17 kernels + their events: 0.21 -> 0.17 ms per 752x480 frame, host image in, results out)
        (void)orbx_set_stage_timing(h, 0);
    }
    std::vector<float> table(int which) {
        // scale tables depend only on (scaleFactor, nlevels): a 1x1-image-free handle is not needed, replay on the host
        std::vector<float> sf(nlevels), isf(nlevels), s2(nlevels), is2(nlevels);
        sf[0] = 1.0f; s2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) { sf[i] = (float)(sf[i - 1] * scaleFactor); s2[i] = sf[i] * sf[i]; }
        for (int i = 0; i < nlevels; i++) { isf[i] = 1.0f / sf[i]; is2[i] = 1.0f / s2[i]; }
        return which == 0 ? sf : which == 1 ? isf : which == 2 ? s2 : is2;
    }

    int nfeatures;
    double scaleFactor;                      // include/ORBextractor.h:96: double member initialised from the float argument
    int nlevels, iniThFAST, minThFAST;
    int device = 0, maxW = 0, maxH = 0;
    orbx_t* h = nullptr;
    bool pyramidStale = false, hostPyramid = true;
    std::vector<orbx_kp_t> kpbuf;
    std::vector<uint8_t> descbuf;
};

}  // namespace ORB_SLAM3
