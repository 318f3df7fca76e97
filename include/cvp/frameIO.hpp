// include/cvp/frameIO.hpp -- the step either side of the hot path: frames from files into
// the detector and edge maps back out, with the transfers overlapped.
//
// The reference gets its frames from cv::VideoCapture and shows the result through a GL pixel buffer
// (src/imgui/imguiApp.cpp:427-431, 496-522; src/cvp/cannyEdgeH.cu:122-212); neither exists on a headless
// MI355X.  This header is the headless counterpart:
//   * cvp::io::readPNM / writePGM   binary PGM (P5, grey) and PPM (P6, RGB -> the BGR byte order cv::imread
//                                   would deliver) -- formats that need no codec library;
//   * cvp::io::readPNG / writePNG   8-bit PNG through zlib (the only codec dependency; libz ships with the image);
//   * cvp::io::FrameStreamer        a ring of `depth` slots, each with its own device context (own stream and
//                                   device buffers) and page-locked staging: while slot k computes, slot k+1
//                                   uploads and slot k-1 downloads.  Frames go in in order and come out in order.
#pragma once
#include "cvmat_min.hpp"
#include "frameView.hpp"

#include <algorithm>

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

struct hc_ctx;

namespace cvp
{
namespace io
{
  // Binary PNM (P5: 8-bit grey -> CV_8UC1, P6: 8-bit RGB -> CV_8UC3 in B,G,R order).  maxval must be 255.
  // The compiled entry points take plain buffers; the cv::Mat forms below are inline (frameView.hpp: a Mat never
  // crosses the library boundary).  pixels: tight rows of width * channels bytes.
  bool readPNMRaw(const std::string &path, std::vector<std::uint8_t> &pixels, int &width, int &height, int &channels);
  bool writePGMRaw(const std::string &path, const std::uint8_t *data, std::size_t step, int width, int height);
  // PNG (BASELINE configs[0] is "a single grayscale PNG"): 8-bit, non-interlaced; grey and grey+alpha -> 1 channel,
  // RGB, RGBA and palette -> 3 channels in B,G,R order (alpha dropped), as cv::imread(IMREAD_UNCHANGED minus alpha)
  // would deliver them.  Decoded with zlib's inflate and the five PNG scanline filters; 16-bit and interlaced files are refused.
  bool readPNGRaw(const std::string &path, std::vector<std::uint8_t> &pixels, int &width, int &height, int &channels);
  // 8-bit greyscale (channels 1) or B,G,R (channels 3) image as PNG (filter 0 on every scanline).
  bool writePNGRaw(const std::string &path, const std::uint8_t *data, std::size_t step, int width, int height, int channels);
  // PNG or binary PNM, told apart by the file's first bytes
  bool readImageRaw(const std::string &path, std::vector<std::uint8_t> &pixels, int &width, int &height, int &channels);
  inline bool readPNM(const std::string &path, cv::Mat &out)
  {
    std::vector<std::uint8_t> px;
    int w = 0, h = 0, ch = 0;
    if (!readPNMRaw(path, px, w, h, ch)) return false;
    cv::Mat img(h, w, ch == 1 ? CV_8UC1 : CV_8UC3);
    const std::size_t row = static_cast<std::size_t>(w) * static_cast<std::size_t>(ch);
    for (int r = 0; r < h; ++r) std::copy(px.begin() + static_cast<std::ptrdiff_t>(row * r), px.begin() + static_cast<std::ptrdiff_t>(row * (r + 1)), img.ptr(r));
    out = img;
    return true;
  }
  // PNG or PNM file -> cv::Mat (CV_8UC1 or CV_8UC3 in B,G,R order)
  inline bool readImage(const std::string &path, cv::Mat &out)
  {
    std::vector<std::uint8_t> px;
    int w = 0, h = 0, ch = 0;
    if (!readImageRaw(path, px, w, h, ch)) return false;
    cv::Mat img(h, w, ch == 1 ? CV_8UC1 : CV_8UC3);
    const std::size_t row = static_cast<std::size_t>(w) * static_cast<std::size_t>(ch);
    for (int r = 0; r < h; ++r) std::copy(px.begin() + static_cast<std::ptrdiff_t>(row * r), px.begin() + static_cast<std::ptrdiff_t>(row * (r + 1)), img.ptr(r));
    out = img;
    return true;
  }
  inline bool writePNG(const std::string &path, const cv::Mat &img)
  {
    const FrameView v = viewOf(img);
    if (v.empty() || (v.type != CV_8UC1 && v.type != CV_8UC3)) return false;
    return writePNGRaw(path, v.data, v.step, v.cols, v.rows, v.channels);
  }
  // 8-bit single-channel image as binary PGM.
  inline bool writePGM(const std::string &path, const cv::Mat &img)
  {
    const FrameView v = viewOf(img);
    if (v.empty() || v.channels != 1) return false;
    return writePGMRaw(path, v.data, v.step, v.cols, v.rows);
  }

  class FrameStreamer
  {
  public:
    // edges: `n` tight width x height u8 maps (0 / 255); firstIndex: running index of the first frame of the batch
    using Sink = std::function<void(const std::uint8_t *edges, int n, long firstIndex)>;

    // mode: 0 = the reference's pipeline (HC_MODE_R), 1 = cv::Canny semantics (HC_MODE_O)
    FrameStreamer(int width, int height, int channels, int batch, int depth = 3, int device = 0, int mode = 0);
    ~FrameStreamer();
    FrameStreamer(const FrameStreamer &) = delete;
    FrameStreamer &operator=(const FrameStreamer &) = delete;

    void setThresholds(int low, int high);
    // page-locked staging of the next batch: `batch` frames of width*channels bytes per row, tight.  Fill it, then commit.
    std::uint8_t *stage();
    // queues upload + detector for the `n` frames written into stage(); if the ring is full, the oldest batch is
    // completed first and handed to `sink`
    void commit(int n, const Sink &sink);
    // completes everything still in flight, oldest first
    void flush(const Sink &sink);
    long framesIn() const { return m_in; }

  private:
    struct Slot
    {
      hc_ctx *ctx = nullptr;
      std::uint8_t *hostIn = nullptr, *hostOut = nullptr;
      int n = 0;
      long first = 0;
      bool busy = false;
    };
    void complete(Slot &s, const Sink &sink);
    int m_w, m_h, m_c, m_batch;
    std::vector<Slot> m_slots;
    int m_head = 0;// slot the next batch goes to (also the oldest one when the ring is full)
    long m_in = 0;
  };
}// namespace io
}// namespace cvp
