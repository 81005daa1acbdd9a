// cvcompat.h -- the handful of OpenCV types and operations the ORBextractor / ORBmatcher class API and the facade's
// wrappers mention, for builds WITHOUT OpenCV (this image, the GPU box).  With real OpenCV define ORBX_WITH_OPENCV before
// including the facade headers and this file is skipped: the SAME wrapper bodies then compile against cv::Mat / cv::Matx /
// cv::KeyPoint proper.  Only what the wrappers write is here: 8-bit images / descriptor matrices, and small float matrices
// with the algebra of the reference's pose expressions (Rcw*p3Dw+tcw, -Rcw.t()*tcw, sRcw/scw, PO.dot(Pn), cv::norm(PO)).
// Numerics of the float algebra follow OpenCV's generic paths: products and dot accumulate in double and round once to
// float (gemm / dotProd_32f), norm is the square root of a double sum; Mat / s multiplies by (float)(1.0 / s).
#pragma once
#ifndef ORBX_WITH_OPENCV
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

// OpenCV spells these as macros (types_c.h / interface.h): plain CV_8U, never cv::CV_8U
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3f { float x = 0, y = 0, z = 0; Point3f() {} Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };

// memory layout == orbx_kp_t == real cv::KeyPoint (28 bytes)
struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// small fixed matrices (cv::Matx33f, Matx31f, Matx44f): row-major val[]
template <class T, int M, int N> struct Matx {
    T val[M * N];
    Matx() { for (auto& v : val) v = 0; }
    T& operator()(int i, int j) { return val[i * N + j]; }
    const T& operator()(int i, int j) const { return val[i * N + j]; }
    T& operator()(int i) { return val[i]; }
    const T& operator()(int i) const { return val[i]; }
    Matx<T, N, M> t() const { Matx<T, N, M> r; for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) r(j, i) = (*this)(i, j); return r; }
};
template <class T, int M, int K, int N> Matx<T, M, N> operator*(const Matx<T, M, K>& a, const Matx<T, K, N>& b) {
    Matx<T, M, N> r;
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { T s = 0; for (int k = 0; k < K; ++k) s += a(i, k) * b(k, j); r(i, j) = s; }   // Matx: element type accumulation
    return r;
}
template <class T, int M, int N> Matx<T, M, N> operator+(const Matx<T, M, N>& a, const Matx<T, M, N>& b) { Matx<T, M, N> r; for (int i = 0; i < M * N; ++i) r.val[i] = a.val[i] + b.val[i]; return r; }
template <class T, int M, int N> Matx<T, M, N> operator-(const Matx<T, M, N>& a) { Matx<T, M, N> r; for (int i = 0; i < M * N; ++i) r.val[i] = -a.val[i]; return r; }
typedef Matx<float, 3, 3> Matx33f;
typedef Matx<float, 3, 1> Matx31f;
typedef Matx<float, 4, 4> Matx44f;

// minimal matrix: 8-bit (images, descriptors) or 32-bit float (poses, points); owning or a view into another one
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;                       // bytes per row
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), step(step_ ? step_ : (size_t)c * esz(type)), data((uint8_t*)ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && own_) return;
        type_ = type;
        own_.reset(new uint8_t[(size_t)r * c * esz(type) + 1]()); data = own_.get(); rows = r; cols = c; step = (size_t)c * esz(type);
    }
    void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
    Mat row(int r) const { return view(r, r + 1, 0, cols); }
    Mat col(int c) const { return view(0, rows, c, c + 1); }
    Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }
    Mat getMat() const { return *this; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * esz(type_));
        return m;
    }
    template <class T> T& at(int i, int j) { return *(T*)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    template <class T> const T& at(int i, int j) const { return *(const T*)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    template <class T> T& at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }              // vectors: 3x1 or 1x3
    template <class T> const T& at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    Mat t() const { Mat m(cols, rows, type_); for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) m.at<float>(j, i) = at<float>(i, j); return m; }
    double dot(const Mat& o) const {
        double s = 0;
        for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) s += (double)at<float>(i, j) * (double)o.at<float>(i, j);
        return s;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type) { Mat m(r, c, type); for (int i = 0; i < r && i < c; ++i) m.at<float>(i, i) = 1.f; return m; }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    Mat view(int r0, int r1, int c0, int c1) const {
        Mat m; m.rows = r1 - r0; m.cols = c1 - c0; m.step = step; m.type_ = type_; m.own_ = own_;
        m.data = data + (size_t)r0 * step + (size_t)c0 * esz(type_);
        return m;
    }
    int type_ = CV_8U;
    std::shared_ptr<uint8_t[]> own_;
};

inline Mat operator*(const Mat& a, const Mat& b) {
    Mat r(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < b.cols; ++j) {
        double s = 0;
        for (int k = 0; k < a.cols; ++k) s += (double)a.at<float>(i, k) * (double)b.at<float>(k, j);
        r.at<float>(i, j) = (float)s;
    }
    return r;
}
inline Mat mat_scaled(const Mat& a, float f) { Mat r(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = a.at<float>(i, j) * f; return r; }
inline Mat operator*(double s, const Mat& a) { return mat_scaled(a, (float)s); }
inline Mat operator*(const Mat& a, double s) { return mat_scaled(a, (float)s); }
inline Mat operator/(const Mat& a, double s) { return mat_scaled(a, (float)(1.0 / s)); }
inline Mat operator+(const Mat& a, const Mat& b) { Mat r(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j); return r; }
inline Mat operator-(const Mat& a, const Mat& b) { Mat r(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j); return r; }
inline Mat operator-(const Mat& a) { return mat_scaled(a, -1.f); }
inline double norm(const Mat& a) { return std::sqrt(a.dot(a)); }

typedef const Mat& InputArray;
typedef Mat& OutputArray;

}  // namespace cv
#endif
