// tests/cpp/other_mat_layout/opencv2/core.hpp -- TEST FIXTURE, not OpenCV.
// A cv::Mat whose object layout differs from the stand-in of include/cvp/cvmat_min.hpp (leading flags / dims words,
// the data pointer before the step, a step object instead of a size_t -- the member order of OpenCV's own class),
// so that tests/cpp/test_cvpipeline built with -Itests/cpp/other_mat_layout proves that no cv::Mat crosses the
// libcvProcessing_hip.so boundary: the library was compiled against the other layout and must still read the frame.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv
{
typedef unsigned char uchar;
struct MatStep
{
  std::size_t *p;
  std::size_t buf[2];
  MatStep() : p(buf) { buf[0] = buf[1] = 0; }
  MatStep(const MatStep &o) : p(buf) { buf[0] = o.buf[0]; buf[1] = o.buf[1]; }
  MatStep &operator=(const MatStep &o) { buf[0] = o.buf[0]; buf[1] = o.buf[1]; return *this; }
  operator std::size_t() const { return buf[0]; }
};

class Mat
{
public:
  Mat() = default;
  Mat(int r, int c, int type_) : flags(0x42FF0000 | type_), dims(2), rows(r), cols(c)
  {
    step.buf[0] = static_cast<std::size_t>(c) * elemSize();
    step.buf[1] = elemSize();
    m_owned.reset(new uchar[step.buf[0] * static_cast<std::size_t>(r)](), std::default_delete<uchar[]>());
    data = m_owned.get();
  }
  Mat(int r, int c, int type_, void *d, std::size_t s = 0) : flags(0x42FF0000 | type_), dims(2), rows(r), cols(c), data(static_cast<uchar *>(d))
  {
    step.buf[0] = s ? s : static_cast<std::size_t>(c) * elemSize();
    step.buf[1] = elemSize();
  }
  int type() const { return flags & 0xFFF; }
  int depth() const { return flags & 7; }
  int channels() const { return ((flags & 0xFF8) >> 3) + 1; }
  std::size_t elemSize() const { return static_cast<std::size_t>(channels()) * (depth() == CV_32F ? 4 : 1); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  uchar *ptr(int r = 0) { return data + static_cast<std::size_t>(step) * static_cast<std::size_t>(r); }
  const uchar *ptr(int r = 0) const { return data + static_cast<std::size_t>(step) * static_cast<std::size_t>(r); }

  int flags = 0x42FF0000;
  int dims = 0;
  int rows = 0, cols = 0;
  uchar *data = nullptr;
  const uchar *datastart = nullptr, *dataend = nullptr, *datalimit = nullptr;
  void *allocator = nullptr, *u = nullptr;
  MatStep step;

private:
  std::shared_ptr<uchar> m_owned;
};
}// namespace cv
