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
  const double ms = timerManager::Get().getAverageTime(cvp::CANNY_STAGES.at(cvp::HYSTER));
  std::printf("ok %dx%dx%d stage %d, avg '%s' %.3f ms\n", w, h, ch, stage, cvp::CANNY_STAGES.at(cvp::HYSTER).c_str(), ms);
  return 0;
}
