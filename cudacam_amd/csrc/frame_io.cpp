// frame_io.cpp -- cvp::io (include/cvp/frameIO.hpp): PNM files and the overlapped host <-> device frame ring.
// Plain C++17 over the C ABI, part of libcvProcessing_hip.so.
#include "../../include/cvp/frameIO.hpp"
#include "../../include/cvp/logging.hpp"
#include "../../include/hipcanny.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <zlib.h>

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
      // a header may claim any size: nothing is allocated before the file is known to hold that many raster bytes
      const long at = std::ftell(f);
      long end = -1;
      if (at >= 0 && std::fseek(f, 0, SEEK_END) == 0) end = std::ftell(f);
      if (at < 0 || end < at || static_cast<std::size_t>(end - at) < total || std::fseek(f, at, SEEK_SET) != 0) {
        std::fclose(f);
        return false;
      }
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

  // ---- PNG (ISO/IEC 15948): signature, IHDR / PLTE / IDAT / IEND chunks, zlib stream, per-scanline filters ----------------
  namespace
  {
    std::uint32_t be32(const std::uint8_t *p) { return (std::uint32_t(p[0]) << 24) | (std::uint32_t(p[1]) << 16) | (std::uint32_t(p[2]) << 8) | p[3]; }
    void put32(std::vector<std::uint8_t> &v, std::uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back(static_cast<std::uint8_t>(x >> s)); }
    const std::uint8_t PNG_SIG[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    int paeth(int a, int b, int c)
    {
      const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
      return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
    }
  }// namespace

  bool readPNGRaw(const std::string &path, std::vector<std::uint8_t> &pixels, int &width, int &height, int &channels)
  {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<std::uint8_t> file;
    std::uint8_t buf[65536];
    for (std::size_t n; (n = std::fread(buf, 1, sizeof buf, f)) > 0;) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    if (file.size() < 8 + 25 || std::memcmp(file.data(), PNG_SIG, 8) != 0) return false;
    std::uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<std::uint8_t> idat, plte;
    for (std::size_t pos = 8; pos + 12 <= file.size();) {
      const std::uint32_t len = be32(&file[pos]);
      if (len > file.size() - pos - 12) return false;
      const std::uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
      if (be32(data + len) != static_cast<std::uint32_t>(crc32(crc32(0L, Z_NULL, 0), type, len + 4))) return false;// chunk CRC covers type + data
      if (!std::memcmp(type, "IHDR", 4) && len == 13) {
        w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
      else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
      else if (!std::memcmp(type, "IEND", 4)) break;
      pos += 12 + len;
    }
    if (w == 0 || h == 0 || w > 65535 || h > 65535 || depth != 8 || interlace != 0) return false;
    const int spp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;// samples per pixel in the file
    if (spp == 0 || (ctype == 3 && plte.size() < 3)) return false;
    const std::size_t stride = static_cast<std::size_t>(w) * spp;
    // deflate expands by at most 1032 : 1, so a header that claims more pixels than the IDAT bytes can hold is rejected
    // before anything of that size is allocated (a 60-byte file may claim 65535 x 65535 RGBA)
    if ((stride + 1) * h > idat.size() * 1032 + 1024) return false;
    std::vector<std::uint8_t> raw((stride + 1) * h);
    uLongf rawLen = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &rawLen, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || rawLen != raw.size()) return false;
    // undo the scanline filters in place (None, Sub, Up, Average, Paeth)
    std::vector<std::uint8_t> prev(stride, 0);
    for (std::uint32_t r = 0; r < h; ++r) {
      std::uint8_t *line = &raw[(stride + 1) * r];
      const int ft = line[0];
      std::uint8_t *x = line + 1;
      if (ft > 4) return false;
      for (std::size_t i = 0; i < stride; ++i) {
        const int a = i >= static_cast<std::size_t>(spp) ? x[i - spp] : 0, b = prev[i], c = i >= static_cast<std::size_t>(spp) ? prev[i - spp] : 0;
        const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : paeth(a, b, c);
        x[i] = static_cast<std::uint8_t>(x[i] + pred);
      }
      std::memcpy(prev.data(), x, stride);
    }
    const int ch = (ctype == 0 || ctype == 4) ? 1 : 3;
    std::vector<std::uint8_t> out(static_cast<std::size_t>(w) * h * ch);
    for (std::uint32_t r = 0; r < h; ++r) {
      const std::uint8_t *x = &raw[(stride + 1) * r + 1];
      std::uint8_t *o = &out[static_cast<std::size_t>(w) * ch * r];
      for (std::uint32_t c = 0; c < w; ++c) {
        if (ctype == 0) o[c] = x[c];
        else if (ctype == 4) o[c] = x[2 * c];
        else if (ctype == 3) {
          const std::size_t e = static_cast<std::size_t>(x[c]) * 3;
          if (e + 3 > plte.size()) return false;
          o[3 * c] = plte[e + 2]; o[3 * c + 1] = plte[e + 1]; o[3 * c + 2] = plte[e];// RGB palette entry -> B,G,R
        } else {
          const std::uint8_t *q = x + static_cast<std::size_t>(c) * spp;
          o[3 * c] = q[2]; o[3 * c + 1] = q[1]; o[3 * c + 2] = q[0];// RGB(A) on disk -> B,G,R in memory, as cv::imread
        }
      }
    }
    pixels.swap(out);
    width = static_cast<int>(w); height = static_cast<int>(h); channels = ch;
    return true;
  }

  bool writePNGRaw(const std::string &path, const std::uint8_t *data, std::size_t step, int width, int height, int channels)
  {
    if (!data || width <= 0 || height <= 0 || (channels != 1 && channels != 3) || step < static_cast<std::size_t>(width) * channels) return false;
    const std::size_t stride = static_cast<std::size_t>(width) * channels;
    std::vector<std::uint8_t> raw((stride + 1) * height);
    for (int r = 0; r < height; ++r) {
      std::uint8_t *line = &raw[(stride + 1) * r];
      line[0] = 0;// filter None
      const std::uint8_t *src = data + step * static_cast<std::size_t>(r);
      if (channels == 1) std::memcpy(line + 1, src, stride);
      else
        for (int c = 0; c < width; ++c) { line[1 + 3 * c] = src[3 * c + 2]; line[2 + 3 * c] = src[3 * c + 1]; line[3 + 3 * c] = src[3 * c]; }// B,G,R -> RGB
    }
    uLongf zlen = compressBound(static_cast<uLong>(raw.size()));
    std::vector<std::uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) return false;
    std::vector<std::uint8_t> out(PNG_SIG, PNG_SIG + 8);
    auto chunk = [&](const char *type, const std::uint8_t *d, std::size_t n) {
      put32(out, static_cast<std::uint32_t>(n));
      const std::size_t at = out.size();
      out.insert(out.end(), type, type + 4);
      out.insert(out.end(), d, d + n);
      put32(out, static_cast<std::uint32_t>(crc32(crc32(0L, Z_NULL, 0), &out[at], static_cast<uInt>(n + 4))));
    };
    std::vector<std::uint8_t> ihdr;
    put32(ihdr, static_cast<std::uint32_t>(width)); put32(ihdr, static_cast<std::uint32_t>(height));
    ihdr.push_back(8); ihdr.push_back(channels == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr.data(), ihdr.size());
    chunk("IDAT", z.data(), zlen);
    chunk("IEND", nullptr, 0);
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return std::fclose(f) == 0 && ok;
  }

  bool readImageRaw(const std::string &path, std::vector<std::uint8_t> &pixels, int &width, int &height, int &channels)
  {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::uint8_t sig[8] = { 0 };
    const std::size_t n = std::fread(sig, 1, 8, f);
    std::fclose(f);
    if (n == 8 && std::memcmp(sig, PNG_SIG, 8) == 0) return readPNGRaw(path, pixels, width, height, channels);
    return readPNMRaw(path, pixels, width, height, channels);
  }

  FrameStreamer::FrameStreamer(int width, int height, int channels, int batch, int depth, int device, int mode)
    : m_w(width), m_h(height), m_c(channels), m_batch(batch), m_slots(static_cast<std::size_t>(depth < 2 ? 2 : depth))
  {
    const std::size_t inBytes = static_cast<std::size_t>(m_w) * m_c * m_h * m_batch, outBytes = static_cast<std::size_t>(m_w) * m_h * m_batch;
    for (Slot &s : m_slots) {
      s.ctx = hc_create(device, m_w, m_h, m_c, m_batch, mode == 1 ? HC_MODE_O : HC_MODE_R);
      if (!s.ctx) die("hc_create");
      // uploads and downloads of all slots on the device's two copy streams: both directions of the link stay busy
      if (hc_set_option(s.ctx, HC_OPT_COPY_STREAMS, 1) != HC_OK) die("hc_set_option(HC_OPT_COPY_STREAMS)");
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
    // waits for this slot's stream only -- its download was queued behind its run by commit() -- while the other slots
    // keep uploading / computing
    if (hc_download_end(s.ctx) != HC_OK) die("hc_download_end");
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
    // ... and so is the download: queued now, it moves over PCIe while the NEXT batches upload (both directions at once)
    if (hc_download_begin(s.ctx, s.hostOut, static_cast<std::size_t>(m_w), static_cast<std::size_t>(m_w) * m_h, n) != HC_OK) die("hc_download_begin");
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
