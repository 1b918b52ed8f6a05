// TEST INFRASTRUCTURE ONLY: declarations of the handful of OpenCV 2.x names the reference's demo programs use (rect.cpp, poly.cpp,
// vidrect.cpp, vidpoly.cpp), so that those programs can be pushed through a compiler and the linker UNCHANGED against include/*.h and
// librectdetect_hip.so in an image without OpenCV (tests/test_cpu_abi.py).  Not an image library: the bodies do nothing.
#ifndef RD_TEST_OPENCV_STUB_HPP
#define RD_TEST_OPENCV_STUB_HPP
#include <stddef.h>
#include <string.h>     // (the real opencv2/core pulls these in; the demo programs rely on that)
#include <assert.h>
#include <math.h>
#include <string>

struct CvPoint { int x, y; };
struct CvSize { int width, height; };
inline CvPoint cvPoint(int x, int y) { CvPoint p = { x, y }; return p; }
inline CvSize cvSize(int w, int h) { CvSize s = { w, h }; return s; }
enum { CV_LOAD_IMAGE_COLOR = 1, CV_CAP_PROP_FRAME_WIDTH = 3, CV_CAP_PROP_FRAME_HEIGHT = 4 };

namespace cv {
enum { WINDOW_AUTOSIZE = 1 };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; } };
struct MatStep { size_t v; MatStep() : v(0) {} operator size_t() const { return v; } };
class Mat {
public:
  unsigned char *data; int cols, rows; MatStep step;
  Mat() : data(0), cols(0), rows(0) {}
  int channels() const { return 3; }
  Mat clone() const { return *this; }
  void copyTo(Mat &m) const { m = *this; }
};
inline Mat imread(const std::string &, int = 1) { return Mat(); }
inline bool imwrite(const std::string &, const Mat &) { return true; }
inline void line(Mat &, CvPoint, CvPoint, const Scalar &, int = 1, int = 8, int = 0) {}
class VideoCapture {
public:
  VideoCapture(int) {}
  VideoCapture(const std::string &) {}
  bool isOpened() const { return false; }
  bool set(int, double) { return false; }
  double get(int) { return 0; }
  bool grab() { return false; }
  bool retrieve(Mat &, int = 0) { return false; }
};
class VideoWriter {
public:
  VideoWriter(const std::string &, int, double, CvSize, bool = true) {}
  bool isOpened() const { return false; }
  void write(const Mat &) {}
};
inline void namedWindow(const std::string &, int = 1) {}
inline void imshow(const std::string &, const Mat &) {}
inline int waitKey(int = 0) { return -1; }
inline void destroyAllWindows() {}
}  // namespace cv
#endif
