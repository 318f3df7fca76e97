/* Sanitizer driver for the CPU oracle (test infrastructure): compiled by tests/test_sanitizers.py together with
 * oracle/canny_oracle.c under -fsanitize=address,undefined, it runs every oracle entry point the parity tests use on
 * small and ragged frames and prints one checksum per output.  The test compares the checksums with the regular build
 * of the oracle -- so the run proves both "no out-of-bounds access, no undefined arithmetic" and "same results". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/canny_oracle.h"

static uint32_t lcg_state;
static uint32_t lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state >> 24; }

static uint64_t fnv(const void *p, size_t n)
{
  const uint8_t *b = (const uint8_t *)p;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

/* content: kind 0 noise, 1 smooth ramp + a bright box (edges and flat areas), 2 flat 255 */
static void fill(uint8_t *img, int w, int h, int ch, int kind, uint32_t seed)
{
  lcg_state = seed;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < ch; ++c) {
        uint8_t v;
        if (kind == 0) v = (uint8_t)lcg();
        else if (kind == 1) v = (uint8_t)(((x * 3 + y * 2 + c * 5) & 127) + ((x > w / 3 && x < 2 * w / 3 && y > h / 4 && y < 3 * h / 4) ? 120 : 0) + (lcg() & 3));
        else v = 255;
        img[((size_t)y * w + x) * ch + c] = v;
      }
}

int main(void)
{
  static const int sizes[][2] = { { 1, 1 }, { 2, 1 }, { 1, 7 }, { 3, 3 }, { 5, 4 }, { 8, 8 }, { 17, 9 }, { 31, 33 }, { 64, 48 }, { 67, 45 } };
  for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; ++si) {
    const int w = sizes[si][0], h = sizes[si][1];
    const size_t n = (size_t)w * h;
    for (int ch = 1; ch <= 3; ch += 2)
      for (int kind = 0; kind < 3; ++kind) {
        /* exact-size heap blocks: any access one past the end is an ASan report */
        uint8_t *img = (uint8_t *)malloc(n * ch);
        fill(img, w, h, ch, kind, 1234u + si * 7u + (unsigned)kind);
        orc_outputs o;
        o.mono = (uint8_t *)malloc(n); o.blur = (uint8_t *)malloc(n); o.grad_disp = (uint8_t *)malloc(n);
        o.nms = (uint8_t *)malloc(n); o.thresh = (uint8_t *)malloc(n); o.edges = (uint8_t *)malloc(n);
        o.sumx = (int16_t *)malloc(n * 2); o.sumy = (int16_t *)malloc(n * 2);
        for (int sat = 0; sat < 2; ++sat) {
          if (orc_canny_r(img, (size_t)w * ch, w, h, ch, 10, 40, sat, &o) != 0) { printf("orc_canny_r failed\n"); return 2; }
          printf("R %dx%dx%d k%d s%d %016llx %016llx %016llx %016llx %016llx %016llx %016llx\n", w, h, ch, kind, sat,
                 (unsigned long long)fnv(o.blur, n), (unsigned long long)fnv(o.sumx, n * 2), (unsigned long long)fnv(o.sumy, n * 2),
                 (unsigned long long)fnv(o.grad_disp, n), (unsigned long long)fnv(o.nms, n), (unsigned long long)fnv(o.thresh, n),
                 (unsigned long long)fnv(o.edges, n));
        }
        {
          int launches = 0;
          uint8_t *t = (uint8_t *)malloc(n);
          orc_hysteresis_tiled(o.thresh, (size_t)w, w, h, t, (size_t)w, 30, 100, &launches);
          printf("T %dx%dx%d k%d %016llx %d\n", w, h, ch, kind, (unsigned long long)fnv(t, n), launches);
          free(t);
        }
        for (int l2 = 0; l2 < 2; ++l2) {
          uint8_t *e = (uint8_t *)malloc(n), *pm = (uint8_t *)malloc(n);
          if (orc_canny_o_ex(img, (size_t)w * ch, w, h, ch, 50.0, 150.0, l2, e, pm) != 0) { printf("orc_canny_o_ex failed\n"); return 2; }
          printf("O %dx%dx%d k%d l%d %016llx %016llx\n", w, h, ch, kind, l2, (unsigned long long)fnv(e, n), (unsigned long long)fnv(pm, n));
          free(e); free(pm);
        }
        if (ch == 1) { /* the batch helpers (OpenMP) */
          uint8_t *two = (uint8_t *)malloc(2 * n), *e2 = (uint8_t *)malloc(2 * n);
          memcpy(two, img, n); memcpy(two + n, img, n);
          orc_canny_r_batch(two, w, h, 2, 10, 40, e2, 2);
          printf("B %dx%d k%d %016llx\n", w, h, kind, (unsigned long long)fnv(e2, 2 * n));
          orc_canny_o_batch(two, w, h, 2, 50.0, 150.0, 0, e2, 2);
          printf("P %dx%d k%d %016llx\n", w, h, kind, (unsigned long long)fnv(e2, 2 * n));
          free(two); free(e2);
        }
        free(img); free(o.mono); free(o.blur); free(o.grad_disp); free(o.nms); free(o.thresh); free(o.edges); free(o.sumx); free(o.sumy);
      }
  }
  printf("done\n");
  return 0;
}
