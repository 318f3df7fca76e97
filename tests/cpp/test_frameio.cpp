// test_frameio -- cvp::io PNM reader / writer, no GPU involved (the FrameStreamer is exercised by tests/test_frame_io.py).
#include "../../include/cvp/frameIO.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

int main(int argc, char **argv)
{
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  // grey round trip, pitched source
  {
    std::uint8_t buf[5 * 8];
    for (int i = 0; i < 40; ++i) buf[i] = static_cast<std::uint8_t>(i * 7);
    cv::Mat src(5, 6, CV_8UC1, buf, 8);
    CHECK(cvp::io::writePGM(dir + "/t_grey.pgm", src));
    cv::Mat back;
    CHECK(cvp::io::readPNM(dir + "/t_grey.pgm", back));
    CHECK(back.rows == 5 && back.cols == 6 && back.channels() == 1);
    for (int r = 0; r < 5; ++r) CHECK(std::memcmp(back.ptr(r), src.ptr(r), 6) == 0);
  }
  // P6 with header comments: RGB on disk -> BGR in memory
  {
    std::FILE *f = std::fopen((dir + "/t_rgb.ppm").c_str(), "wb");
    CHECK(f != nullptr);
    std::fputs("P6\n# a comment\n2 1\n# another\n255\n", f);
    const unsigned char px[6] = { 10, 20, 30, 40, 50, 60 };
    std::fwrite(px, 1, 6, f);
    std::fclose(f);
    cv::Mat img;
    CHECK(cvp::io::readPNM(dir + "/t_rgb.ppm", img));
    CHECK(img.rows == 1 && img.cols == 2 && img.channels() == 3);
    const std::uint8_t *p = img.ptr(0);
    CHECK(p[0] == 30 && p[1] == 20 && p[2] == 10 && p[3] == 60 && p[4] == 50 && p[5] == 40);
  }
  // rejected inputs: ASCII PGM, 16-bit maxval, truncated raster, 3-channel write
  {
    std::FILE *f = std::fopen((dir + "/t_bad1.pgm").c_str(), "wb");
    std::fputs("P2\n2 2\n255\n1 2 3 4\n", f);
    std::fclose(f);
    cv::Mat m;
    CHECK(!cvp::io::readPNM(dir + "/t_bad1.pgm", m));
    f = std::fopen((dir + "/t_bad2.pgm").c_str(), "wb");
    std::fputs("P5\n2 2\n65535\n", f);
    std::fwrite("12345678", 1, 8, f);
    std::fclose(f);
    CHECK(!cvp::io::readPNM(dir + "/t_bad2.pgm", m));
    f = std::fopen((dir + "/t_bad3.pgm").c_str(), "wb");
    std::fputs("P5\n4 4\n255\n", f);
    std::fwrite("123", 1, 3, f);
    std::fclose(f);
    CHECK(!cvp::io::readPNM(dir + "/t_bad3.pgm", m));
    CHECK(!cvp::io::readPNM(dir + "/does_not_exist.pgm", m));
    cv::Mat c3(2, 2, CV_8UC3);
    CHECK(!cvp::io::writePGM(dir + "/t_c3.pgm", c3));
  }
  // PNG: grey and B,G,R round trips through the writer (zlib deflate, filter None), pitched source
  {
    std::uint8_t buf[7 * 12];
    for (int i = 0; i < 84; ++i) buf[i] = static_cast<std::uint8_t>(i * 11 + 3);
    cv::Mat g(7, 9, CV_8UC1, buf, 12);
    CHECK(cvp::io::writePNG(dir + "/t_grey.png", g));
    cv::Mat back;
    CHECK(cvp::io::readImage(dir + "/t_grey.png", back));
    CHECK(back.rows == 7 && back.cols == 9 && back.channels() == 1);
    for (int r = 0; r < 7; ++r) CHECK(std::memcmp(back.ptr(r), g.ptr(r), 9) == 0);
    cv::Mat c(2, 4, CV_8UC3, buf, 12);
    CHECK(cvp::io::writePNG(dir + "/t_bgr.png", c));
    cv::Mat cb;
    CHECK(cvp::io::readImage(dir + "/t_bgr.png", cb));
    CHECK(cb.rows == 2 && cb.cols == 4 && cb.channels() == 3);
    for (int r = 0; r < 2; ++r) CHECK(std::memcmp(cb.ptr(r), c.ptr(r), 12) == 0);
    // readImage tells PNM from PNG by content, not by name
    CHECK(cvp::io::readImage(dir + "/t_grey.pgm", back) && back.cols == 6);
    // a corrupted chunk (CRC) and a truncated file are refused
    std::FILE *f = std::fopen((dir + "/t_grey.png").c_str(), "rb");
    std::vector<unsigned char> bytes(4096);
    const std::size_t n = std::fread(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    bytes.resize(n);
    std::vector<unsigned char> bad = bytes;
    bad[n / 2] ^= 0x40;
    f = std::fopen((dir + "/t_badcrc.png").c_str(), "wb");
    std::fwrite(bad.data(), 1, bad.size(), f);
    std::fclose(f);
    CHECK(!cvp::io::readImage(dir + "/t_badcrc.png", back));
    f = std::fopen((dir + "/t_trunc.png").c_str(), "wb");
    std::fwrite(bytes.data(), 1, n - 20, f);
    std::fclose(f);
    CHECK(!cvp::io::readImage(dir + "/t_trunc.png", back));
  }
  std::printf(fails ? "test_frameio: %d failure(s)\n" : "test_frameio: ok\n", fails);
  return fails ? 1 : 0;
}
