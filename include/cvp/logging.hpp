// include/cvp/logging.hpp -- LOG_DEBUG / LOG_INFO / LOG_ERROR with the reference's macro names
// (src/utils/logging.hpp:12-14) on a tiny stderr logger: spdlog/fmt are Conan dependencies of the
// reference that a headless MI355X node does not need.  "{}" placeholders are filled in order.
#pragma once

#include <cstdio>
#include <sstream>
#include <string>

namespace Utils
{
namespace detail
{
  inline void fmtInto(std::ostringstream &os, const char *f) { os << f; }
  template<typename T, typename... Rest>
  void fmtInto(std::ostringstream &os, const char *f, const T &v, const Rest &...rest)
  {
    for (; *f; ++f) {
      if (f[0] == '{' && f[1] == '}') {
        os << v;
        fmtInto(os, f + 2, rest...);
        return;
      }
      os << *f;
    }
  }
  template<typename... Args>
  void logLine(const char *level, const char *func, const char *f, const Args &...args)
  {
    std::ostringstream os;
    fmtInto(os, f, args...);
    std::fprintf(stderr, "[%s] [%s] %s\n", level, func, os.str().c_str());
  }
}// namespace detail
inline void InitializeLogger() {}
}// namespace Utils

#if defined(DEBUG_BUILD)
#define LOG_DEBUG(...) ::Utils::detail::logLine("debug", __func__, __VA_ARGS__)
#else
#define LOG_DEBUG(...) (void)0
#endif
#define LOG_INFO(...) ::Utils::detail::logLine("info", __func__, __VA_ARGS__)
#define LOG_ERROR(...) ::Utils::detail::logLine("error", __func__, __VA_ARGS__)
