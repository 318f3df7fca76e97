// canny_files -- headless front end of the detector: PNG or binary PGM / PPM files in, <name>.edges.pgm out.
//   canny_files [-o outdir] [--mode R|O] [--low L] [--high H] [--batch N] [--repeat R] [--threads T] [--no-write] file...
// --mode O: cv::Canny(img, low, high, 3, false) semantics instead of the reference's pipeline (defaults 50 / 150).
// All files must have the same size and channel count.  Frames are streamed through cvp::io::FrameStreamer
// (page-locked staging, upload / compute / download overlapped); prints the end-to-end rate, disk excluded
// when --repeat re-streams the already loaded frames.
#include "../include/cvp/frameIO.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

int main(int argc, char **argv)
{
  std::string outdir = ".";
  int low = -1, high = -1, batch = 16, repeat = 1, threads = 4, mode = 0;
  bool nowrite = false;  // rate measurements: stream, but leave the disk out
  std::vector<std::string> files;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&](int &v) { if (i + 1 < argc) v = std::atoi(argv[++i]); };
    if (a == "-o" && i + 1 < argc) outdir = argv[++i];
    else if (a == "--mode" && i + 1 < argc) mode = (argv[++i][0] == 'O' || argv[i][0] == 'o') ? 1 : 0;
    else if (a == "--low") next(low);
    else if (a == "--high") next(high);
    else if (a == "--batch") next(batch);
    else if (a == "--repeat") next(repeat);
    else if (a == "--threads") next(threads);
    else if (a == "--no-write") nowrite = true;
    else files.push_back(a);
  }
  if (files.empty()) {
    std::fprintf(stderr, "usage: canny_files [-o outdir] [--mode R|O] [--low L] [--high H] [--batch N] [--repeat R] [--threads T] [--no-write] file.png|file.pgm|file.ppm ...\n");
    return 2;
  }
  std::vector<cv::Mat> frames(files.size());
  for (std::size_t i = 0; i < files.size(); ++i)
    if (!cvp::io::readImage(files[i], frames[i])) {
      std::fprintf(stderr, "cannot read %s (binary P5/P6 with maxval 255 expected)\n", files[i].c_str());
      return 1;
    }
  const int w = frames[0].cols, h = frames[0].rows, ch = frames[0].channels();
  for (const cv::Mat &m : frames)
    if (m.cols != w || m.rows != h || m.channels() != ch) {
      std::fprintf(stderr, "all frames must share one size and channel count\n");
      return 1;
    }
  if (batch < 1) batch = 1;
  if (low < 0) low = mode ? 50 : 10;
  if (high < 0) high = mode ? 150 : 40;
  cvp::io::FrameStreamer streamer(w, h, ch, batch, 3, 0, mode);
  streamer.setThresholds(low, high);
  const std::size_t frameBytes = static_cast<std::size_t>(w) * ch * h;
  long written = 0;
  int pass = 0;
  auto sink = [&](const std::uint8_t *edges, int n, long first) {
    if (pass != repeat - 1 || nowrite) return;// only the last pass is written out
    for (int k = 0; k < n; ++k) {
      const std::size_t idx = static_cast<std::size_t>((first + k) % static_cast<long>(files.size()));
      std::string base = files[idx];
      const std::size_t slash = base.find_last_of('/');
      if (slash != std::string::npos) base = base.substr(slash + 1);
      const std::size_t dot = base.find_last_of('.');
      if (dot != std::string::npos) base = base.substr(0, dot);
      cv::Mat img(h, w, CV_8UC1, const_cast<std::uint8_t *>(edges) + static_cast<std::size_t>(k) * w * h);
      if (cvp::io::writePGM(outdir + "/" + base + ".edges.pgm", img)) ++written;
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  for (pass = 0; pass < repeat; ++pass) {
    std::size_t i = 0;
    while (i < frames.size()) {
      std::uint8_t *dst = streamer.stage();
      const int n = static_cast<int>(std::min<std::size_t>(static_cast<std::size_t>(batch), frames.size() - i));
      // the staging copy is the host-side bottleneck (one core moves ~9 GB/s): spread the frames of a batch over threads
      auto fill = [&](int k0, int k1) {
        for (int k = k0; k < k1; ++k)
          for (int r = 0; r < h; ++r)
            std::memcpy(dst + k * frameBytes + static_cast<std::size_t>(r) * w * ch, frames[i + static_cast<std::size_t>(k)].ptr(r), static_cast<std::size_t>(w) * ch);
      };
      const int nt = std::max(1, std::min(threads, n));
      std::vector<std::thread> pool;
      for (int t = 1; t < nt; ++t) pool.emplace_back(fill, n * t / nt, n * (t + 1) / nt);
      fill(0, n / nt);
      for (std::thread &t : pool) t.join();
      i += static_cast<std::size_t>(n);
      streamer.commit(n, sink);
    }
    if (pass == repeat - 1) streamer.flush(sink);
    else streamer.flush(cvp::io::FrameStreamer::Sink());
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("{\"frames\": %ld, \"width\": %d, \"height\": %d, \"channels\": %d, \"batch\": %d, \"seconds\": %.6f, \"frames_per_s\": %.1f, \"written\": %ld}\n",
              streamer.framesIn(), w, h, ch, batch, dt, streamer.framesIn() / dt, written);
  return 0;
}
