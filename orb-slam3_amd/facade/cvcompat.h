// cvcompat.h -- the handful of OpenCV types the ORBextractor / ORBmatcher class API mentions, for builds WITHOUT
// OpenCV (this image, the GPU box).  With real OpenCV define ORBX_WITH_OPENCV before including the facade headers
// and this file is skipped: the facade then speaks cv::InputArray / cv::OutputArray / cv::Mat / cv::KeyPoint.
#pragma once
#ifndef ORBX_WITH_OPENCV
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_8UC1 = 0 };

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };

// memory layout == orbx_kp_t == real cv::KeyPoint (28 bytes)
struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// minimal owning/non-owning 8-bit matrix
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void* ext, size_t step_ = 0) : rows(r), cols(c), step(step_ ? step_ : (size_t)c), data((uint8_t*)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && own_) return;
        own_.reset(new uint8_t[(size_t)r * c]); data = own_.get(); rows = r; cols = c; step = (size_t)c;
    }
    void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
    Mat row(int r) const { return Mat(1, cols, CV_8U, (void*)ptr(r), step); }
    Mat getMat() const { return *this; }
private:
    std::shared_ptr<uint8_t[]> own_;
};

typedef const Mat& InputArray;
typedef Mat& OutputArray;

}  // namespace cv
#endif
