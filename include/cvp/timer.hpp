// include/cvp/timer.hpp -- the timerManager singleton the UI reads (reference: src/utils/timer.hpp:13-67;
// used at src/imgui/imguiApp.cpp:357-376) with the same public members: createTimer, addTime,
// getAverageTime, beginTimerList / endTimerList over std::map<std::string, timer>.
#pragma once

#include <cstddef>
#include <map>
#include <string>

#include "logging.hpp"

struct timer
{
  double totalTime = 0.0;
  std::size_t nbCount = 0;
  float averageTime() const { return nbCount ? static_cast<float>(totalTime / static_cast<double>(nbCount)) : 0.0f; }
};

class timerManager
{
public:
  using TimerMap = std::map<std::string, timer>;

  static timerManager &Get()
  {
    static timerManager instance;
    return instance;
  }
  timerManager(const timerManager &) = delete;
  timerManager &operator=(const timerManager &) = delete;

  void createTimer(std::string name) { m_timers.emplace(std::move(name), timer{}); }

  void addTime(std::string name, double time)
  {
    const auto hit = m_timers.find(name);
    if (hit == m_timers.end()) {
      LOG_ERROR("Timer {} unknown", name);
      return;
    }
    hit->second.totalTime += time;
    ++hit->second.nbCount;
  }

  double getAverageTime(std::string name) const
  {
    const auto hit = m_timers.find(name);
    if (hit == m_timers.end() || hit->second.nbCount == 0) {
      LOG_ERROR("Timer {} unknown", name);
      return 0.0;
    }
    return hit->second.totalTime / static_cast<double>(hit->second.nbCount);
  }

  TimerMap::const_iterator beginTimerList() const { return m_timers.cbegin(); }
  TimerMap::const_iterator endTimerList() const { return m_timers.cend(); }

private:
  timerManager() = default;
  TimerMap m_timers;
};
