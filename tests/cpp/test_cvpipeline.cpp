// Host-side check of the C++ drop-in surface: drives cvp::cvPipeline exactly as CudaCam's UI does
// (src/imgui/imguiApp.cpp:102, 328-348, 515) on a PGM-less synthetic frame read from stdin-free args:
//   test_cvpipeline <w> <h> <channels> <in.raw> <stage> <out.raw>
// pytest generates the input, runs this binary on the GPU box and compares <out.raw> with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/cvp/cvPipeline.hpp"
#include "../../include/cvp/timer.hpp"

int main(int argc, char **argv)
{
  if (argc != 7) return 2;
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), ch = std::atoi(argv[3]), stage = std::atoi(argv[5]);
  std::vector<unsigned char> in(static_cast<size_t>(w) * h * ch);
  FILE *f = std::fopen(argv[4], "rb");
  if (!f || std::fread(in.data(), 1, in.size(), f) != in.size()) return 3;
  std::fclose(f);

  cvp::cvPipeline pipeline(0, static_cast<unsigned>(w), static_cast<unsigned>(h), ch);
  if (pipeline.getLowThreshold() != 10 || pipeline.getHighThreshold() != 40 || !pipeline.isCudaProfilingEnabled()) return 4;
  pipeline.setLowThreshold(90);// -> min(90, 40)
  if (pipeline.getLowThreshold() != 40) return 5;
  pipeline.setLowThreshold(10);
  if (pipeline.process(cv::Mat(), static_cast<cvp::CannyStage>(stage))) return 6;                  // blank frame -> false
  if (pipeline.process(cv::Mat(h, w, CV_32FC1), static_cast<cvp::CannyStage>(stage))) return 7;   // wrong type -> false
  cv::Mat frame(h, w, ch == 3 ? CV_8UC3 : CV_8UC1, in.data());
  if (!pipeline.process(frame, static_cast<cvp::CannyStage>(stage))) return 8;
  const auto &out = pipeline.output();
  f = std::fopen(argv[6], "wb");
  if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) return 9;
  std::fclose(f);
  // Per-stage timers (reference: _endCudaTimer(stage) books every stage that ran, src/cvp/cannyEdgeH.cu:415-430, and the
  // UI sums the rows up to the selected stage, src/imgui/imguiApp.cpp:364-376): after one frame every stage up to
  // `stage` holds exactly one non-zero sample -- stage 0 only for 3-channel input (1-channel input skips it, unless
  // MONO itself was asked for) -- and no later stage holds any.
  if (!pipeline.process(frame, static_cast<cvp::CannyStage>(stage))) return 8;// a second frame: two samples each
  const auto &timers = timerManager::Get();
  float sum = 0.0f;
  bool past = false;
  for (auto it = timers.beginTimerList(); it != timers.endTimerList(); ++it) {
    int st = -1;
    for (const auto &kv : cvp::CANNY_STAGES)
      if (kv.second == it->first) st = static_cast<int>(kv.first);
    if (st < 0) return 10;
    const bool ran = st <= stage && (st != cvp::MONO || ch == 3 || stage == cvp::MONO);
    std::printf("timer '%s' samples %zu avg %.4f ms%s\n", it->first.c_str(), it->second.nbCount, it->second.averageTime(), ran ? "" : " (not run)");
    if (ran && (it->second.nbCount != 2 || !(it->second.totalTime > 0.0))) return 11;
    if (!ran && it->second.nbCount != 0) return 12;
    if (!past) sum += it->second.averageTime();
    if (cvp::CANNY_STAGES.at(static_cast<cvp::CannyStage>(stage)) == it->first) past = true;// the UI's partial sum stops here
  }
  std::printf("ok %dx%dx%d stage %d, total %.4f ms\n", w, h, ch, stage, sum);
  return 0;
}
