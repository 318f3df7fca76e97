// include/cvp/cvmat_min.hpp -- cv::Mat for builds without OpenCV.
// The reference operator takes `cv::Mat` by value (src/cvp/cannyEdgeH.hpp:23, cvPipeline.hpp:26) and
// touches only rows, cols, step, channels(), type(), empty(), ptr().  When <opencv2/core.hpp> is
// available the real type is used and this header adds nothing.
#pragma once

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <memory>

#ifndef CV_8UC1
#define CV_8U 0
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(5, 1)
#endif

namespace cv
{
// Reference-counted header over caller or owned pixel memory (copying a Mat copies the header only,
// like cv::Mat).
class Mat
{
public:
  Mat() = default;
  Mat(int rows_, int cols_, int type_) : rows(rows_), cols(cols_), m_type(type_)
  {
    step = static_cast<std::size_t>(cols_) * elemSize();
    m_owned.reset(new std::uint8_t[step * static_cast<std::size_t>(rows_)](), std::default_delete<std::uint8_t[]>());
    data = m_owned.get();
  }
  Mat(int rows_, int cols_, int type_, void *data_, std::size_t step_ = 0)
    : rows(rows_), cols(cols_), data(static_cast<std::uint8_t *>(data_)), m_type(type_)
  {
    step = step_ ? step_ : static_cast<std::size_t>(cols_) * elemSize();
  }
  int type() const { return m_type; }
  int channels() const { return (m_type >> 3) + 1; }
  int depth() const { return m_type & 7; }
  std::size_t elemSize() const { return static_cast<std::size_t>(channels()) * (depth() == 5 ? 4 : 1); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  std::uint8_t *ptr(int r = 0) { return data + step * static_cast<std::size_t>(r); }
  const std::uint8_t *ptr(int r = 0) const { return data + step * static_cast<std::size_t>(r); }

  int rows = 0, cols = 0;
  std::size_t step = 0;
  std::uint8_t *data = nullptr;

private:
  int m_type = CV_8UC1;
  std::shared_ptr<std::uint8_t> m_owned;
};
}// namespace cv
#endif
