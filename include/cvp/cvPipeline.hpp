// include/cvp/cvPipeline.hpp -- cvp::cvPipeline, the "pure cpp proxy" the UI talks to.
// Signatures as in the reference (src/cvp/cvPipeline.hpp:20-39); behaviour as in
// src/cvp/cvPipeline.cpp:19-96 (false for blank frames and for types other than CV_8UC1 / CV_8UC3).
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "cvmat_min.hpp"
#include "define.hpp"

namespace cvp
{
namespace cuda
{
  class CannyEdge;
}

class cvPipeline
{
public:
  cvPipeline(const unsigned int pbo, const unsigned int inputImageCols, const unsigned int inputImageRows, const int inputImageNbChannels);
  ~cvPipeline();

  bool process(cv::Mat inputImage, CannyStage finalStage);

  void setLowThreshold(unsigned char low);
  unsigned char getLowThreshold() const;

  void setHighThreshold(unsigned char high);
  unsigned char getHighThreshold() const;

  void enableCudaProfiling(bool profiling);
  bool isCudaProfilingEnabled() const;

  // display-less addition: the image the reference would have left in the GL PBO
  const std::vector<std::uint8_t> &output() const;

private:
  std::unique_ptr<cuda::CannyEdge> m_cudaCannyEdge;
};
}// namespace cvp
