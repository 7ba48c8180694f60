// ORACLE / TEST INFRASTRUCTURE ONLY — a minimal stand-in for <opencv2/opencv.hpp>.
//
// The reference's hot path uses only cv::Mat::{data, size(), type(), at<T>()}, the
// (rows, cols, type) and (Size, type, Scalar) constructors, cv::imread(path, 0) for the
// 11x11 templates and cv::imwrite for SavePatch.  This header
// provides exactly that so the reference's translation units compile UNMODIFIED
// (`make -C oracle ref`).  cv::Mat copies share their pixels (reference counting), like
// OpenCV's: MonoSLAM::copy_into_patch writes through a by-value Mat (monoslam.cpp:1235).
//   imread: binary PGM (P5) files, or an image registered in memory with
//           cv::shim_register_image(name, mat) (so tests need no temporary files);
//   imwrite: writes a binary PGM whatever the extension (no PNG encoder here) and keeps a
//           copy retrievable with cv::shim_last_written().
#ifndef SL2_REF_SHIM_OPENCV
#define SL2_REF_SHIM_OPENCV

#include <math.h>
#include <stdlib.h>
#include <cstdio>
#include <iostream>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace cv {

typedef unsigned char uchar;

enum { CV_8U = 0, CV_64F = 6 };
#define CV_8UC1 0
#define CV_64FC1 6

struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
};

struct Scalar {
  double v;
  Scalar(double a = 0.0) : v(a) {}
};

class Mat {
 public:
  uchar* data;
  int rows, cols;

  Mat() : data(nullptr), rows(0), cols(0), type_(CV_8UC1) {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type) { allocate(); }
  Mat(Size s, int type) : rows(s.height), cols(s.width), type_(type) { allocate(); }
  Mat(Size s, int type, const Scalar& v) : rows(s.height), cols(s.width), type_(type) {
    allocate();
    fill(v.v);
  }
  Mat(int r, int c, int type, const Scalar& v) : rows(r), cols(c), type_(type) {
    allocate();
    fill(v.v);
  }
  // Non-owning header over caller memory (step == width), like cv::Mat(rows, cols, type, ptr).
  Mat(int r, int c, int type, void* ext) : data((uchar*)ext), rows(r), cols(c), type_(type) {}

  Size size() const { return Size(cols, rows); }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows * cols == 0; }
  size_t elemSize() const { return type_ == CV_64FC1 ? sizeof(double) : 1; }

  template <class T>
  T& at(int r, int c) { return ((T*)data)[(size_t)r * (size_t)cols + (size_t)c]; }
  template <class T>
  const T& at(int r, int c) const { return ((const T*)data)[(size_t)r * (size_t)cols + (size_t)c]; }

  Mat clone() const {
    Mat m(rows, cols, type_);
    if (data) std::memcpy(m.data, data, (size_t)rows * cols * elemSize());
    return m;
  }

 private:
  void allocate() {
    buf_ = std::make_shared<std::vector<uchar> >((size_t)rows * cols * elemSize());
    data = buf_->data();
  }
  void fill(double v) {
    if (type_ == CV_64FC1) {
      double* p = (double*)data;
      for (size_t i = 0; i < (size_t)rows * cols; ++i) p[i] = v;
    } else {
      std::memset(data, (int)v, (size_t)rows * cols);
    }
  }
  int type_;
  std::shared_ptr<std::vector<uchar> > buf_;
};

inline std::map<std::string, Mat>& shim_registry() {
  static std::map<std::string, Mat> r;
  return r;
}
inline void shim_register_image(const std::string& name, const Mat& m) { shim_registry()[name] = m.clone(); }
inline Mat& shim_last_written() {
  static Mat m;
  return m;
}

inline Mat imread(const std::string& path, int /*flags*/ = 0) {
  std::map<std::string, Mat>::iterator it = shim_registry().find(path);
  if (it != shim_registry().end()) return it->second.clone();
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return Mat();
  std::vector<uchar> bytes;
  uchar tmp[4096];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) bytes.insert(bytes.end(), tmp, tmp + n);
  std::fclose(f);
  // P5 header: magic, width, height, maxval separated by whitespace, '#' comments
  size_t pos = 0;
  std::string tok[4];
  for (int t = 0; t < 4; ++t) {
    for (;;) {
      while (pos < bytes.size() && (bytes[pos] == ' ' || bytes[pos] == '\n' || bytes[pos] == '\r' || bytes[pos] == '\t')) ++pos;
      if (pos < bytes.size() && bytes[pos] == '#') {
        while (pos < bytes.size() && bytes[pos] != '\n') ++pos;
        continue;
      }
      break;
    }
    while (pos < bytes.size() && !(bytes[pos] == ' ' || bytes[pos] == '\n' || bytes[pos] == '\r' || bytes[pos] == '\t'))
      tok[t].push_back((char)bytes[pos++]);
  }
  ++pos;
  if (tok[0] != "P5") return Mat();
  const int w = std::atoi(tok[1].c_str()), h = std::atoi(tok[2].c_str());
  if (w <= 0 || h <= 0 || std::atoi(tok[3].c_str()) > 255 || pos + (size_t)w * h > bytes.size()) return Mat();
  Mat m(h, w, CV_8UC1);
  std::memcpy(m.data, bytes.data() + pos, (size_t)w * h);
  return m;
}

inline bool imwrite(const std::string& path, const Mat& m) {
  shim_last_written() = m.clone();
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  std::fprintf(f, "P5\n%d %d\n255\n", m.cols, m.rows);
  std::fwrite(m.data, 1, (size_t)m.rows * m.cols, f);
  std::fclose(f);
  return true;
}

}  // namespace cv

#endif  // SL2_REF_SHIM_OPENCV
