// include/cvp/frameView.hpp -- what crosses the libcvProcessing_hip.so boundary instead of cv::Mat.
//
// The reference operator takes `cv::Mat` by value (src/cvp/cannyEdgeH.hpp:23, src/cvp/cvPipeline.hpp:26).  A Mat's
// layout belongs to whichever OpenCV (or the stand-in of cvmat_min.hpp) the including translation unit sees, so it
// must not be part of a compiled interface: the host application and this library may be built against different
// ones.  The Mat-taking members of cvp::cuda::CannyEdge, cvp::cvPipeline and cvp::io are therefore inline wrappers,
// compiled in the CALLER's translation unit, that reduce the Mat to this plain view; only the view crosses.
#pragma once

#include <cstddef>
#include <cstdint>

#include "cvmat_min.hpp"

namespace cvp
{
struct FrameView
{
  const std::uint8_t *data = nullptr;  // first pixel of row 0 (null: blank frame)
  std::size_t step = 0;                // bytes between rows (cv::Mat::step)
  int rows = 0, cols = 0;
  int type = 0;                        // cv::Mat::type(): CV_8UC1 / CV_8UC3 are the supported ones
  int channels = 0;
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
};

inline FrameView viewOf(const cv::Mat &m)
{
  FrameView v;
  if (m.empty()) return v;
  v.data = m.ptr(0);
  v.step = static_cast<std::size_t>(m.step);
  v.rows = m.rows;
  v.cols = m.cols;
  v.type = m.type();
  v.channels = m.channels();
  return v;
}
}// namespace cvp
