// include/cvp/define.hpp -- stage identifiers of the Canny operator, header-compatible with the
// reference's src/cvp/define.hpp:9-34 (same enumerators, same display strings: the strings double
// as timerManager keys, src/cvp/cannyEdgeH.cu:35-37 and src/imgui/imguiApp.cpp:357-376).
#pragma once

#include <array>
#include <map>
#include <string>
#include <utility>

namespace cvp
{
enum CannyStage : int { MONO = 0, GAUSSIAN = 1, GRADIENT = 2, NMS = 3, THRESH = 4, HYSTER = 5 };

struct CompareCannyStage
{
  bool operator()(const CannyStage &lhs, const CannyStage &rhs) const { return static_cast<int>(lhs) < static_cast<int>(rhs); }
};

namespace detail
{
  inline const std::array<std::pair<CannyStage, const char *>, 6> &stageTable()
  {
    static const std::array<std::pair<CannyStage, const char *>, 6> table{ {
      { MONO, "1/6 Mono Conversion" },
      { GAUSSIAN, "2/6 Gaussian Noise Removal" },
      { GRADIENT, "3/6 Gradient Computation" },
      { NMS, "4/6 Non Maximum Suppression" },
      { THRESH, "5/6 Double Threshold" },
      { HYSTER, "6/6 Hysteresis" },
    } };
    return table;
  }
  inline std::map<CannyStage, std::string, CompareCannyStage> makeStageMap()
  {
    std::map<CannyStage, std::string, CompareCannyStage> m;
    for (const auto &e : stageTable()) m.emplace(e.first, e.second);
    return m;
  }
}// namespace detail

// same name and type as the reference's table, so `CANNY_STAGES.find(stage)` / iteration keep working
static const std::map<CannyStage, std::string, CompareCannyStage> CANNY_STAGES = detail::makeStageMap();
}// namespace cvp
