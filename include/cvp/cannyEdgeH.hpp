// include/cvp/cannyEdgeH.hpp -- cvp::cuda::CannyEdge for MI355X.
// Same namespace, class name, constructor and public members as the reference operator
// (src/cvp/cannyEdgeH.hpp:17-32) so cvp::cvPipeline and the ImGui host compile against it
// unchanged; the body is a thin layer over the C ABI of libhipcanny.so (include/hipcanny.h).
// Differences, all forced by the platform: the `pbo` GL buffer id must be 0 (no GL on a headless
// MI355X) and the result is read with output()/download(); 1-channel input skips stage 0 instead of
// running rgb2mono on an unwritten buffer (reference bug, cannyEdgeH.cu:60-110 vs :140-146).
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "cvmat_min.hpp"
#include "define.hpp"
#include "frameView.hpp"

struct hc_ctx;

namespace cvp
{
namespace cuda
{
  class CannyEdge
  {
  public:
    CannyEdge(unsigned int pbo, unsigned int imageWidth, unsigned int imageHeight, int imageNbChannels);
    ~CannyEdge();
    CannyEdge(const CannyEdge &) = delete;
    CannyEdge &operator=(const CannyEdge &) = delete;

    // Reference signature (src/cvp/cannyEdgeH.hpp:23).  Inline on purpose: cv::Mat never crosses the library boundary
    // (frameView.hpp); the compiled entry point is runView.
    void run(cv::Mat input, cvp::CannyStage finalStage) { runView(viewOf(input), finalStage); }
    void runView(const FrameView &input, cvp::CannyStage finalStage);

    void setLowThreshold(unsigned char low);
    unsigned char getLowThreshold() const { return m_lowThresh; }

    void setHighThreshold(unsigned char high);
    unsigned char getHighThreshold() const { return m_highThresh; }

    void enableKernelProfiling(bool profiling);
    bool isKernelProfilingEnabled() const { return m_isKernelProfilingEnabled; }

    // -- additions for a display-less device (the reference writes into the GL PBO instead) --
    // tight W x H u8 image of the last run's final stage (what _sendOutputToOpenGL put in the PBO)
    const std::vector<std::uint8_t> &output() const { return m_output; }
    hc_ctx *handle() const { return m_ctx; }

  private:
    hc_ctx *m_ctx = nullptr;
    int m_inputW, m_inputH, m_inputNbChannels;
    unsigned char m_lowThresh, m_highThresh;
    bool m_isKernelProfilingEnabled;
    std::vector<std::uint8_t> m_output;
  };
}// namespace cuda
}// namespace cvp
