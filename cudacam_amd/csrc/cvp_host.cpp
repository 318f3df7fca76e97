// cvp_host.cpp -- cvp::cuda::CannyEdge and cvp::cvPipeline over the C ABI (libhipcanny.so).
// Plain C++17, no HIP headers: this is what replaces the reference's cvProcessing library
// (src/cvp/cannyEdgeH.cu + cvPipeline.cpp) in CudaCam's build.
#include "../../include/cvp/cannyEdgeH.hpp"
#include "../../include/cvp/cvPipeline.hpp"
#include "../../include/cvp/logging.hpp"
#include "../../include/cvp/timer.hpp"
#include "../../include/hipcanny.h"

#include <cstdlib>

// ---- timerManager (include/cvp/timer.hpp) -------------------------------------------------------------------
float timer::averageTime() const { return nbCount ? static_cast<float>(totalTime / static_cast<double>(nbCount)) : 0.0f; }

timerManager &timerManager::Get()
{
  static timerManager table;
  return table;
}

const timer *timerManager::lookup(const std::string &name) const
{
  const auto hit = m_timers.find(name);
  if (hit != m_timers.end()) return &hit->second;
  LOG_ERROR("Timer {} unknown", name);
  return nullptr;
}

void timerManager::createTimer(std::string name) { m_timers.emplace(std::move(name), timer{}); }

void timerManager::addTime(std::string name, double time)
{
  if (lookup(name) == nullptr) return;
  timer &slot = m_timers.find(name)->second;
  slot.totalTime += time;
  ++slot.nbCount;
}

double timerManager::getAverageTime(std::string name) const
{
  const timer *t = lookup(name);
  if (t && t->nbCount == 0) LOG_ERROR("Timer {} unknown", name);  // the reference reports an empty timer the same way
  return (t && t->nbCount) ? t->totalTime / static_cast<double>(t->nbCount) : 0.0;
}

timerManager::TimerMap::const_iterator timerManager::beginTimerList() const { return m_timers.cbegin(); }
timerManager::TimerMap::const_iterator timerManager::endTimerList() const { return m_timers.cend(); }

namespace cvp
{
namespace cuda
{
  namespace
  {
    // reference convention for device errors: log, then stop the application (src/cvp/helper.hpp:4-17)
    void checkHip(int rc, const char *what)
    {
      if (rc == HC_OK) return;
      LOG_ERROR("HIP NON-KERNEL ERROR errorcode={} {} : {}", rc, what, hc_last_error());
      LOG_ERROR("Stopping Application");
      std::exit(EXIT_FAILURE);
    }
  }// namespace

  CannyEdge::CannyEdge(unsigned int pbo, unsigned int imageWidth, unsigned int imageHeight, int imageNbChannels)
    : m_inputW(static_cast<int>(imageWidth)), m_inputH(static_cast<int>(imageHeight)), m_inputNbChannels(imageNbChannels),
      m_lowThresh(10), m_highThresh(40), m_isKernelProfilingEnabled(true)// defaults: cannyEdgeH.cu:22-24
  {
    if (pbo != 0) LOG_ERROR("GL interop is not available on this device: pbo {} ignored, read output() instead", pbo);
    m_ctx = hc_create(0, m_inputW, m_inputH, m_inputNbChannels, 1, HC_MODE_R);
    if (!m_ctx) checkHip(HC_E_HIP, "hc_create");
    checkHip(hc_set_thresholds(m_ctx, m_lowThresh, m_highThresh), "hc_set_thresholds");
    checkHip(hc_enable_profiling(m_ctx, 1), "hc_enable_profiling");
    auto &timers = timerManager::Get();
    for (const auto &stage : cvp::CANNY_STAGES) timers.createTimer(stage.second);// cannyEdgeH.cu:35-37
  }

  CannyEdge::~CannyEdge() { hc_destroy(m_ctx); }

  void CannyEdge::setLowThreshold(unsigned char low)
  {
    m_lowThresh = std::min(low, m_highThresh);// cannyEdgeH.hpp:25
    checkHip(hc_set_thresholds(m_ctx, m_lowThresh, m_highThresh), "hc_set_thresholds");
  }

  void CannyEdge::setHighThreshold(unsigned char high)
  {
    m_highThresh = std::max(high, m_lowThresh);// cannyEdgeH.hpp:28
    checkHip(hc_set_thresholds(m_ctx, m_lowThresh, m_highThresh), "hc_set_thresholds");
  }

  void CannyEdge::enableKernelProfiling(bool profiling)
  {
    m_isKernelProfilingEnabled = profiling;
    checkHip(hc_enable_profiling(m_ctx, profiling ? 1 : 0), "hc_enable_profiling");
  }

  void CannyEdge::runView(const FrameView &input, cvp::CannyStage finalStage)
  {
    LOG_DEBUG("Start Canny Edge Filter on HIP device");
    if (static_cast<int>(finalStage) < MONO || static_cast<int>(finalStage) > HYSTER) {
      LOG_ERROR("Canny Stage Not Recognized");// cannyEdgeH.cu:111-114
      return;
    }
    // cannyEdgeH.cu:124-130: a frame whose geometry differs from construction is logged and skipped
    if (input.rows != m_inputH || input.cols != m_inputW || input.channels != m_inputNbChannels) {
      LOG_ERROR("Cannot load image to GPU, specs different since initialization");
      return;
    }
    if (input.type != CV_8UC3 && input.type != CV_8UC1) {
      LOG_ERROR("Only supporting CV_8UC3 and CV_8UC1 input types for now");
      return;
    }
    checkHip(hc_upload(m_ctx, input.data, input.step, input.step * static_cast<std::size_t>(input.rows), 1), "hc_upload");
    checkHip(hc_run(m_ctx, static_cast<int>(finalStage), 1), "hc_run");
    m_output.resize(static_cast<std::size_t>(m_inputW) * m_inputH);
    checkHip(hc_download(m_ctx, m_output.data(), static_cast<std::size_t>(m_inputW), m_output.size(), 1), "hc_download");
    if (m_isKernelProfilingEnabled) {
      // one sample per stage that ran, as _endCudaTimer(stage) books them (cannyEdgeH.cu:415-430); how a fused kernel's
      // time is shared among the stages it covers: hipcanny.h, hc_stage_time_ms
      auto &timers = timerManager::Get();
      for (int s = static_cast<int>(MONO); s <= static_cast<int>(HYSTER); ++s) {
        float ms = 0.0f;
        checkHip(hc_stage_time_ms(m_ctx, s, &ms), "hc_stage_time_ms");
        if (ms >= 0.0f) timers.addTime(CANNY_STAGES.at(static_cast<CannyStage>(s)), ms);
      }
    }
    LOG_DEBUG("End Canny Edge Filter on HIP device");
  }
}// namespace cuda

cvPipeline::cvPipeline(const unsigned int pbo, const unsigned int inputImageCols, const unsigned int inputImageRows, const int inputImageNbChannels)
  : m_detector(std::make_unique<cuda::CannyEdge>(pbo, inputImageCols, inputImageRows, inputImageNbChannels))
{
}

// the reference calls unique_ptr::release() here and leaks the operator (cvPipeline.cpp:14-17)
cvPipeline::~cvPipeline() = default;

bool cvPipeline::processView(const FrameView &inputImage, CannyStage finalStage)
{
  if (!m_detector) {
    LOG_ERROR("Cannot process the webcam stream, device is not ready.");
    return false;
  }
  if (inputImage.empty()) {
    LOG_ERROR("Blank frame grabbed");
    return false;
  }
  if (inputImage.type != CV_8UC3 && inputImage.type != CV_8UC1) {
    LOG_ERROR("Only supporting CV_8UC3 and CV_8UC1 input types for now");
    return false;
  }
  m_detector->runView(inputImage, finalStage);
  return true;
}

void cvPipeline::setLowThreshold(unsigned char low)
{
  if (m_detector) m_detector->setLowThreshold(low);
  else LOG_ERROR("Cannot modify low threshold, device is not ready.");
}

unsigned char cvPipeline::getLowThreshold() const { return m_detector ? m_detector->getLowThreshold() : 0; }

void cvPipeline::setHighThreshold(unsigned char high)
{
  if (m_detector) m_detector->setHighThreshold(high);
  else LOG_ERROR("Cannot modify high threshold, device is not ready.");
}

unsigned char cvPipeline::getHighThreshold() const { return m_detector ? m_detector->getHighThreshold() : 255; }

void cvPipeline::enableCudaProfiling(bool profiling)
{
  if (m_detector) m_detector->enableKernelProfiling(profiling);
  else LOG_ERROR("Cannot modify profiling, device is not ready.");
}

bool cvPipeline::isCudaProfilingEnabled() const { return m_detector ? m_detector->isKernelProfilingEnabled() : false; }

const std::vector<std::uint8_t> &cvPipeline::output() const { return m_detector->output(); }
}// namespace cvp
