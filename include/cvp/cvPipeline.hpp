// include/cvp/cvPipeline.hpp -- cvp::cvPipeline: the host-only proxy CudaCam's UI owns (src/imgui/imguiApp.cpp:102)
// and calls once per camera frame (imguiApp.cpp:515).  Every public member keeps the reference's name, argument
// list and meaning (src/cvp/cvPipeline.hpp:20-39), so the UI sources compile against this header unchanged; the body
// (cudacam_amd/csrc/cvp_host.cpp) forwards to the MI355X detector instead of the CUDA one.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "cvmat_min.hpp"  // cv::Mat: OpenCV's when available, a minimal stand-in otherwise
#include "define.hpp"     // cvp::CannyStage
#include "frameView.hpp"  // what actually crosses the library boundary

namespace cvp
{
namespace cuda
{
  class CannyEdge;  // defined in cannyEdgeH.hpp; kept out of this header as in the reference
}

class cvPipeline
{
public:
  // pbo: GL pixel-buffer id of the reference's display path; must be 0 here (no GL on a headless MI355X, see output()).
  // cols / rows / channels: geometry of every frame that will be processed (CV_8UC1 or CV_8UC3).
  cvPipeline(unsigned int pbo, unsigned int inputImageCols, unsigned int inputImageRows, int inputImageNbChannels);
  ~cvPipeline();

  // One frame through the detector up to `finalStage`.  Returns false -- and logs, as src/cvp/cvPipeline.cpp:27-36
  // does -- for an empty frame or a type other than CV_8UC1 / CV_8UC3; true otherwise.
  // Inline on purpose: cv::Mat never crosses the library boundary (frameView.hpp); the compiled entry point is processView.
  bool process(cv::Mat inputImage, CannyStage finalStage) { return processView(viewOf(inputImage), finalStage); }
  bool processView(const FrameView &inputImage, CannyStage finalStage);

  // Double-threshold limits, clamped against each other (src/cvp/cannyEdgeH.hpp:25-29).
  unsigned char getLowThreshold() const;
  unsigned char getHighThreshold() const;
  void setLowThreshold(unsigned char low);
  void setHighThreshold(unsigned char high);

  // Per-stage timers (timerManager entries named by cvp::CANNY_STAGES); on by default like the reference.
  bool isCudaProfilingEnabled() const;
  void enableCudaProfiling(bool profiling);

  // Not in the reference: the tight width x height u8 image of the last processed frame -- what the reference leaves
  // in the GL pixel buffer (src/cvp/cannyEdgeH.cu:154-212).
  const std::vector<std::uint8_t> &output() const;

private:
  std::unique_ptr<cuda::CannyEdge> m_detector;
};
}// namespace cvp
