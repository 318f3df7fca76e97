// frame_io.cpp -- cvp::io (include/cvp/frameIO.hpp): PNM files and the overlapped host <-> device frame ring.
// Plain C++17 over the C ABI, part of libcvProcessing_hip.so.
#include "../../include/cvp/frameIO.hpp"
#include "../../include/cvp/logging.hpp"
#include "../../include/hipcanny.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cvp
{
namespace io
{
  namespace
  {
    // next header token of a PNM file; '#' starts a comment that runs to the end of the line
    bool pnmToken(std::FILE *f, int &value)
    {
      int ch = std::fgetc(f);
      for (;;) {
        while (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') ch = std::fgetc(f);
        if (ch != '#') break;
        while (ch != '\n' && ch != EOF) ch = std::fgetc(f);
      }
      if (ch < '0' || ch > '9') return false;
      long v = 0;
      while (ch >= '0' && ch <= '9') {
        v = v * 10 + (ch - '0');
        if (v > 1000000) return false;
        ch = std::fgetc(f);
      }
      value = static_cast<int>(v);
      return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n';// exactly one whitespace byte precedes the raster
    }

    [[noreturn]] void die(const char *what)
    {
      LOG_ERROR("FrameStreamer: {} : {}", what, hc_last_error());
      LOG_ERROR("Stopping Application");// reference convention for device errors (src/cvp/helper.hpp:4-17)
      std::exit(EXIT_FAILURE);
    }
  }// namespace

  bool readPNMRaw(const std::string &path, std::vector<std::uint8_t> &pixels, int &width, int &height, int &channels)
  {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    const int m0 = std::fgetc(f), m1 = std::fgetc(f);
    int w = 0, h = 0, maxval = 0;
    if (m0 == 'P' && (m1 == '5' || m1 == '6') && pnmToken(f, w) && pnmToken(f, h) && pnmToken(f, maxval) && w > 0 && h > 0 && maxval == 255) {
      const int ch = m1 == '5' ? 1 : 3;
      const std::size_t total = static_cast<std::size_t>(w) * ch * static_cast<std::size_t>(h);
      std::vector<std::uint8_t> img(total);
      ok = std::fread(img.data(), 1, total, f) == total;
      if (ok && ch == 3)
        for (std::size_t i = 0; i + 2 < total; i += 3) std::swap(img[i], img[i + 2]);// RGB on disk -> BGR in memory, as cv::imread
      if (ok) {
        pixels.swap(img);
        width = w; height = h; channels = ch;
      }
    }
    std::fclose(f);
    return ok;
  }

  bool writePGMRaw(const std::string &path, const std::uint8_t *data, std::size_t step, int width, int height)
  {
    if (!data || width <= 0 || height <= 0 || step < static_cast<std::size_t>(width)) return false;
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fprintf(f, "P5\n%d %d\n255\n", width, height) > 0;
    for (int r = 0; r < height && ok; ++r) ok = std::fwrite(data + step * static_cast<std::size_t>(r), 1, static_cast<std::size_t>(width), f) == static_cast<std::size_t>(width);
    return std::fclose(f) == 0 && ok;
  }

  FrameStreamer::FrameStreamer(int width, int height, int channels, int batch, int depth, int device)
    : m_w(width), m_h(height), m_c(channels), m_batch(batch), m_slots(static_cast<std::size_t>(depth < 2 ? 2 : depth))
  {
    const std::size_t inBytes = static_cast<std::size_t>(m_w) * m_c * m_h * m_batch, outBytes = static_cast<std::size_t>(m_w) * m_h * m_batch;
    for (Slot &s : m_slots) {
      s.ctx = hc_create(device, m_w, m_h, m_c, m_batch, HC_MODE_R);
      if (!s.ctx) die("hc_create");
      s.hostIn = static_cast<std::uint8_t *>(hc_host_alloc(inBytes));
      s.hostOut = static_cast<std::uint8_t *>(hc_host_alloc(outBytes));
      if (!s.hostIn || !s.hostOut) die("hc_host_alloc");
    }
  }

  FrameStreamer::~FrameStreamer()
  {
    for (Slot &s : m_slots) {
      if (s.ctx) {
        (void)hc_sync(s.ctx);
        hc_destroy(s.ctx);
      }
      hc_host_free(s.hostIn);
      hc_host_free(s.hostOut);
    }
  }

  void FrameStreamer::setThresholds(int low, int high)
  {
    for (Slot &s : m_slots)
      if (hc_set_thresholds(s.ctx, low, high) != HC_OK) die("hc_set_thresholds");
  }

  std::uint8_t *FrameStreamer::stage() { return m_slots[static_cast<std::size_t>(m_head)].hostIn; }

  void FrameStreamer::complete(Slot &s, const Sink &sink)
  {
    if (!s.busy) return;
    // waits for this slot's stream only; the other slots keep uploading / computing meanwhile
    if (hc_download(s.ctx, s.hostOut, static_cast<std::size_t>(m_w), static_cast<std::size_t>(m_w) * m_h, s.n) != HC_OK) die("hc_download");
    s.busy = false;
    if (sink) sink(s.hostOut, s.n, s.first);
  }

  void FrameStreamer::commit(int n, const Sink &sink)
  {
    if (n <= 0 || n > m_batch) die("commit: frame count out of range");
    Slot &s = m_slots[static_cast<std::size_t>(m_head)];
    // stage() handed out this slot's buffer, so its previous batch was completed when the ring wrapped (below)
    s.n = n;
    s.first = m_in;
    const std::size_t row = static_cast<std::size_t>(m_w) * m_c;
    if (hc_upload(s.ctx, s.hostIn, row, row * m_h, n) != HC_OK) die("hc_upload");                  // asynchronous: page-locked source
    if (hc_run(s.ctx, HC_STAGE_HYSTER, n) != HC_OK) die("hc_run");                                   // asynchronous
    s.busy = true;
    m_in += n;
    m_head = (m_head + 1) % static_cast<int>(m_slots.size());
    complete(m_slots[static_cast<std::size_t>(m_head)], sink);// the slot stage() returns next must be free: finish the oldest batch
  }

  void FrameStreamer::flush(const Sink &sink)
  {
    for (std::size_t k = 0; k < m_slots.size(); ++k) complete(m_slots[(static_cast<std::size_t>(m_head) + k) % m_slots.size()], sink);
  }
}// namespace io
}// namespace cvp
