// include/cvp/timer.hpp -- per-stage timing table shared by the detector (which feeds it) and the UI (which lists
// it: src/imgui/imguiApp.cpp:357-376).  Same public names and meanings as the reference's src/utils/timer.hpp:13-67;
// the bodies live in cudacam_amd/csrc/cvp_host.cpp (libcvProcessing_hip.so).
#pragma once

#include <cstddef>
#include <map>
#include <string>

// Accumulated milliseconds of one named stage and how many runs contributed.
struct timer
{
  double totalTime = 0.0;
  std::size_t nbCount = 0;
  // mean of the contributions, 0 while there are none
  float averageTime() const;
};

// Process-wide table of named timers.  The detector creates one per cvp::CANNY_STAGES entry and adds the hipEvent
// time of every profiled run; readers walk it with beginTimerList() / endTimerList().
class timerManager
{
public:
  using TimerMap = std::map<std::string, timer>;

  static timerManager &Get();  // the one instance

  void createTimer(std::string name);                 // no effect if the name exists
  void addTime(std::string name, double time);        // unknown name: logged, ignored
  double getAverageTime(std::string name) const;      // unknown or empty timer: logged, 0.0

  TimerMap::const_iterator beginTimerList() const;
  TimerMap::const_iterator endTimerList() const;

  timerManager(const timerManager &) = delete;
  timerManager &operator=(const timerManager &) = delete;

private:
  timerManager() = default;
  const timer *lookup(const std::string &name) const;  // null + log when the name is unknown
  TimerMap m_timers;
};
