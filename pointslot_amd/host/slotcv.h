// Minimal cv::-compatible POD layer so that ORB_SLAM2-shaped host code compiles without OpenCV (which is not
// available in the build image).  Only what the hot-path classes touch: Point2f, KeyPoint (byte-compatible with
// cv::KeyPoint), and a reference-counted-free Mat that either owns its pixels or views caller memory.
// Define POINTSLOT_WITH_OPENCV to use the real OpenCV types instead (a maintainer's build of the reference).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#ifdef POINTSLOT_WITH_OPENCV
#include <opencv2/core/core.hpp>
namespace pscv = cv;
#else
namespace slotcv {
enum { CV_8U = 0, CV_8UC1 = 0, CV_32F = 5 };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct KeyPoint {
  Point2f pt; float size = 0; float angle = -1; float response = 0; int octave = 0; int class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");
class Mat {
 public:
  int rows = 0, cols = 0, flags = CV_8U;
  uint8_t* data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), flags(type), data((uint8_t*)ext),
      step(step_ ? step_ : (size_t)c * elemSize(type)) {}
  void create(int r, int c, int type) {
    rows = r; cols = c; flags = type; step = (size_t)c * elemSize(type);
    own_ = std::shared_ptr<std::vector<uint8_t>>(new std::vector<uint8_t>((size_t)r * step));
    data = own_->data();
  }
  void release() { own_.reset(); data = nullptr; rows = cols = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return flags; }
  template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
  template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
  template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  // element i of a row or column vector (cv::Mat::at<T>(int))
  template <typename T> T& at(int i) { return cols == 1 ? ptr<T>(i)[0] : ptr<T>(0)[i]; }
  template <typename T> const T& at(int i) const { return cols == 1 ? ptr<T>(i)[0] : ptr<T>(0)[i]; }
  Mat clone() const {
    Mat m(rows, cols, flags);
    for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize(flags));
    return m;
  }
  bool isContinuous() const { return step == (size_t)cols * elemSize(flags); }
  Mat row(int r) const { Mat m(1, cols, flags, data + (size_t)r * step, step); m.own_ = own_; return m; }
  static size_t elemSize(int type) { return type == CV_32F ? 4 : 1; }
 private:
  std::shared_ptr<std::vector<uint8_t>> own_;
};
}  // namespace slotcv
namespace pscv = slotcv;
#endif
