// Sanitizer driver for the image readers (tests/test_sanitizers.py): compiled together with cudacam_amd/csrc/frame_io.cpp
// under -fsanitize=address,undefined.  Every path on the command line goes through cvp::io::readImageRaw; a well-formed
// file prints its size and a checksum, a malformed one prints "reject" -- the process must never be stopped by a
// sanitizer report, an uncaught exception or an allocation the file's own size does not justify.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cvp/frameIO.hpp"

int main(int argc, char **argv)
{
  for (int i = 1; i < argc; ++i) {
    std::vector<std::uint8_t> px;
    int w = 0, h = 0, ch = 0;
    if (!cvp::io::readImageRaw(argv[i], px, w, h, ch)) {
      std::printf("reject\n");
      continue;
    }
    if (px.size() != static_cast<std::size_t>(w) * h * ch) return 3;
    std::uint64_t hash = 1469598103934665603ull;
    for (std::uint8_t b : px) { hash ^= b; hash *= 1099511628211ull; }
    std::printf("ok %d %d %d %016llx\n", w, h, ch, static_cast<unsigned long long>(hash));
  }
  return 0;
}
