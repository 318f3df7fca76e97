/*
 * canny_oracle.c -- CPU restatement of CudaCam's Canny pipeline.  TEST INFRASTRUCTURE ONLY
 * (see canny_oracle.h for the rules and the parity-pinning status: "parity unpinned" by the
 * reference's own tests; pinned by SURVEY App. C.4 KATs, exhaustive checks and oracle/_ref).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference/).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (the float chain must not be re-associated or
 * contracted by the compiler; the fused variant calls fmaf() explicitly).
 */
#include "canny_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/cvp/cannyEdgeH.cu:372 */
static const int K5[25] = { 2, 4, 5, 4, 2, 4, 9, 12, 9, 4, 5, 12, 15, 12, 5, 4, 9, 12, 9, 4, 2, 4, 5, 4, 2 };

/* ---- stage 0: src/cvp/cannyEdgeD.cu:14-19, 53-69 -------------------------------------------- */
void orc_gray_bgr(const uint8_t *bgr, size_t stride, int w, int h, uint8_t *mono, size_t mstride)
{
  /* B_WT = (int)(64*0.114f+0.5f) = 7, G_WT = 38, R_WT = 19 */
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const uint8_t *p = bgr + (size_t)r * stride + 3 * (size_t)c;
      int v = (p[0] * 7 + p[1] * 38 + p[2] * 19) >> 6;
      mono[(size_t)r * mstride + c] = (uint8_t)(v < 255 ? v : 255);
    }
}

/* ---- stage 1 coefficients: src/cvp/cannyEdgeH.cu:372-380 ------------------------------------ */
void orc_gauss_coeffs(float gk[25])
{
  /* `GK_CPU[i][j] *= 1 / 159.0f;` : the int 1 is converted to float, the quotient is rounded to
   * binary32, then each integer-valued float is multiplied by it (one more rounding). */
  volatile float one = 1.0f, d = 159.0f;
  volatile float r = one / d;
  for (int i = 0; i < 25; ++i) {
    volatile float k = (float)K5[i];
    volatile float p = k * r;
    gk[i] = p;
  }
}

static inline uint8_t gauss_chain_at(const uint8_t *in, size_t istride, int w, int h, int row, int col,
                                     const float *gk, int fused)
{
  /* src/cvp/cannyEdgeD.cu:102-115; out-of-image taps are 0 (:91-98) */
  float fsum = 0.0f;
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 5; ++c) {
      int rr = row - 2 + r, cc = col - 2 + c;
      float px = 0.0f;
      if (rr >= 0 && rr < h && cc >= 0 && cc < w) px = (float)in[(size_t)rr * istride + cc];
      if (fused) {
        fsum = fmaf(gk[r * 5 + c], px, fsum);
      } else {
        volatile float prod = gk[r * 5 + c] * px;
        volatile float s = fsum + prod;
        fsum = s;
      }
    }
  return (uint8_t)(int)fsum; /* truncation toward zero; 0 <= fsum < 256 */
}

void orc_gaussian(const uint8_t *in, size_t istride, int w, int h, uint8_t *out, size_t ostride, int fused)
{
  float gk[25];
  orc_gauss_coeffs(gk);
  for (int row = 0; row < h; ++row)
    for (int col = 0; col < w; ++col)
      out[(size_t)row * ostride + col] = gauss_chain_at(in, istride, w, h, row, col, gk, fused);
}

void orc_gaussian_shortcut(const uint8_t *in, size_t istride, int w, int h, uint8_t *out, size_t ostride)
{
  /* The float chain differs from S/159 by < 4.2e-4 absolute (25 roundings of partial sums < 256
   * plus coefficient error), far less than 1/159, so trunc(chain) == floor(S/159) whenever
   * S % 159 != 0.  Only exact multiples need the literal chain. */
  float gk[25];
  orc_gauss_coeffs(gk);
  for (int row = 0; row < h; ++row)
    for (int col = 0; col < w; ++col) {
      unsigned S = 0;
      for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 5; ++c) {
          int rr = row - 2 + r, cc = col - 2 + c;
          if (rr >= 0 && rr < h && cc >= 0 && cc < w) S += (unsigned)K5[r * 5 + c] * in[(size_t)rr * istride + cc];
        }
      unsigned q = (S * 52759u) >> 23; /* == S / 159 for S <= 40545, checked exhaustively in tests */
      uint8_t v = (uint8_t)q;
      if (q * 159u == S) v = gauss_chain_at(in, istride, w, h, row, col, gk, 1);
      out[(size_t)row * ostride + col] = v;
    }
}

/* ---- stage 2a: src/cvp/cannyEdgeD.cu:121-172 ------------------------------------------------- */
static inline int px0(const uint8_t *img, size_t stride, int w, int h, int r, int c)
{
  return (r >= 0 && r < h && c >= 0 && c < w) ? img[(size_t)r * stride + c] : 0;
}

void orc_sobel(const uint8_t *blur, size_t bstride, int w, int h, int16_t *sumx, int16_t *sumy, size_t sstride)
{
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
#define B(dr, dc) px0(blur, bstride, w, h, r + (dr), c + (dc))
      int sx = -B(-1, -1) + B(-1, 1) - 2 * B(0, -1) + 2 * B(0, 1) - B(1, -1) + B(1, 1);     /* :158-161 */
      int sy = (B(-1, -1) + 2 * B(-1, 0) + B(-1, 1)) - (B(1, -1) + 2 * B(1, 0) + B(1, 1)); /* :165-167 */
#undef B
      sumx[(size_t)r * sstride + c] = (int16_t)sx;
      sumy[(size_t)r * sstride + c] = (int16_t)sy;
    }
}

/* ---- stage 2b: src/cvp/cannyEdgeD.cu:175-198 ------------------------------------------------- */
float orc_grad_float(int sumx, int sumy)
{
  volatile float sX = (float)sumx / 8.0f, sY = (float)sumy / 8.0f; /* :163,169 */
  volatile float xx = sX * sX, yy = sY * sY;
  volatile float s = xx + yy; /* unfused; sX*sX and sY*sY are exact (<= 22 significant bits), so an
                                 fma here gives the same value */
  volatile float g = 4.0f * sqrtf(s); /* GRAD_COEFF * sqrtf(...) :195 */
  return g;
}

int orc_grad_trunc_float(int sumx, int sumy) { return (int)orc_grad_float(sumx, sumy); }

static unsigned isqrt_u32(unsigned x)
{
  unsigned r = (unsigned)sqrt((double)x);
  while ((unsigned long long)r * r > x) --r;
  while ((unsigned long long)(r + 1) * (r + 1) <= x) ++r;
  return r;
}

int orc_grad_trunc_int(int sumx, int sumy)
{
  unsigned S = (unsigned)(sumx * sumx + sumy * sumy);
  return (int)isqrt_u32(S >> 2); /* floor(sqrt(S)/2) */
}

int orc_dir_bin(int sumx, int sumy)
{
  /* SURVEY App. A.5 exact integer rule for cannyEdgeD.cu:196 + :239-264 */
  if (sumx == 0) return 0;
  if (sumx < 0) { sumx = -sumx; sumy = -sumy; }
  long a = sumx, b = sumy < 0 ? -sumy : sumy;
  long t = (a + b) * (a + b);
  int lt22 = t < 2 * b * b;
  int gt67 = t < 2 * a * a;
  if (sumy >= 0) return lt22 ? 0 : (!gt67 ? 1 : 2);
  return gt67 ? 2 : (!lt22 ? 3 : 0);
}

int orc_dir_bin_kernel(int sumx, int sumy)
{
  /* HIP-kernel form: a=|sumx|, b=|sumy|, P=2ab, D=a^2-b^2;  P<|D| -> axis bin by sign of D,
   * else a diagonal whose orientation is the sign of sumx*sumy.  Equal to orc_dir_bin() for every
   * pair except (0,0), where the gradient is 0 and the bin cannot matter. */
  int a = sumx < 0 ? -sumx : sumx, b = sumy < 0 ? -sumy : sumy;
  int P = 2 * a * b, D = a * a - b * b;
  int aD = D < 0 ? -D : D;
  if (P < aD) return D > 0 ? 2 : 0;
  return ((sumx ^ sumy) < 0) ? 3 : 1;
}

int orc_dir_bin_float(int sumx, int sumy)
{
  volatile float sX = (float)sumx / 8.0f, sY = (float)sumy / 8.0f;
  volatile float slope = atan2f(sX, sY);          /* :196 (argument order as in the source) */
  volatile float a180 = slope * 180.0f;           /* :239 */
  volatile float angle = a180 / 3.141592654f;     /* CUDART_PI_F */
  if (angle < 0.0f) angle = angle + 180.0f;       /* :240 */
  if (angle < 22.5f || angle > 157.5f) return 0;  /* :245 */
  if (22.5f <= angle && angle <= 67.5f) return 1; /* :250 */
  if (67.5f < angle && angle <= 112.5f) return 2; /* :255 */
  if (112.5f < angle && angle <= 157.5f) return 3;/* :260 */
  return -1;
}

/* ---- stage 3: src/cvp/cannyEdgeD.cu:201-270 -------------------------------------------------- */
void orc_nms(const int16_t *sumx, const int16_t *sumy, size_t sstride, int w, int h, uint8_t *nms, size_t nstride, int saturate)
{
  /* Comparisons of the float grad are comparisons of the integer S = sumx^2+sumy^2 (grad is a
   * strictly increasing function of S; checked exhaustively in tests). */
#define SQ(r, c) (((r) >= 0 && (r) < h && (c) >= 0 && (c) < w) ? \
  ((long)sumx[(size_t)(r) * sstride + (c)] * sumx[(size_t)(r) * sstride + (c)] + \
   (long)sumy[(size_t)(r) * sstride + (c)] * sumy[(size_t)(r) * sstride + (c)]) : 0L)
  static const int dq[4][2] = { { 1, 0 }, { 1, -1 }, { 0, 1 }, { -1, -1 } }; /* q: :247,252,257,262 */
  static const int dr[4][2] = { { -1, 0 }, { -1, 1 }, { 0, -1 }, { 1, 1 } }; /* r: :248,253,258,263 */
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      int sx = sumx[(size_t)r * sstride + c], sy = sumy[(size_t)r * sstride + c];
      long g = SQ(r, c);
      int bin = orc_dir_bin(sx, sy);
      long q = SQ(r + dq[bin][0], c + dq[bin][1]);
      long rr = SQ(r + dr[bin][0], c + dr[bin][1]);
      int keep = (q <= g) && (rr <= g);
      /* :267 `min((unsigned char)gradVal, 255)`: the cast comes first and is out of range (UB) for
       * gradients 256..721.  Canonical Mode R (saturate == 0): wrap mod 256 -- integer min(int,int)
       * folds away and the low byte is stored (nvcc by analysis, x86 host emulation observed, SURVEY
       * App. A.5/C.4).  saturate != 0: min(g, 255) -- what hipcc makes of the same line on gfx950
       * (HIP resolves min(unsigned char,int) to the double overload and skips the byte truncation;
       * observed with oracle/_ref). */
      int gt = orc_grad_trunc_int(sx, sy);
      nms[(size_t)r * nstride + c] = keep ? (uint8_t)(saturate ? (gt > 255 ? 255 : gt) : (gt & 0xFF)) : 0;
    }
#undef SQ
}

/* ---- stage 4: src/cvp/cannyEdgeD.cu:273-293 -------------------------------------------------- */
void orc_threshold(const uint8_t *nms, size_t nstride, int w, int h, int low, int high, uint8_t *thr, size_t tstride)
{
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      int v = nms[(size_t)r * nstride + c];
      thr[(size_t)r * tstride + c] = v > high ? ORC_FINAL_EDGE : v > low ? ORC_CANDIDATE_EDGE : ORC_NO_EDGE;
    }
}

/* ---- stage 5: src/cvp/cannyEdgeD.cu:295-395, src/cvp/cannyEdgeH.cu:297-338 ------------------- */
long orc_hysteresis(const uint8_t *thr, size_t tstride, int w, int h, uint8_t *out, size_t ostride)
{
  /* Unique fixpoint of "a 128 with any 255 among its 8 neighbours becomes 255" (monotone), then
   * removeCandidates.  Flood fill from every strong pixel. */
  long promoted = 0;
  size_t n = (size_t)w * h;
  int *stack = (int *)malloc(sizeof(int) * (n ? n : 1));
  size_t sp = 0;
  for (int r = 0; r < h; ++r) {
    memcpy(out + (size_t)r * ostride, thr + (size_t)r * tstride, (size_t)w);
  }
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c)
      if (out[(size_t)r * ostride + c] == ORC_FINAL_EDGE) stack[sp++] = r * w + c;
  while (sp) {
    int idx = stack[--sp];
    int r = idx / w, c = idx % w;
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        if (rr < 0 || rr >= h || cc < 0 || cc >= w) continue;
        uint8_t *p = out + (size_t)rr * ostride + cc;
        if (*p == ORC_CANDIDATE_EDGE) {
          *p = ORC_FINAL_EDGE;
          stack[sp++] = rr * w + cc;
          ++promoted;
        }
      }
  }
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      uint8_t *p = out + (size_t)r * ostride + c;
      if (*p == ORC_CANDIDATE_EDGE) *p = 0; /* removeCandidates :379-395 */
    }
  free(stack);
  return promoted;
}

static int hyst_launch(const uint8_t *in, uint8_t *out, int w, int h, int tile)
{
  /* one `hysteresis<<<grid,blocks>>>` launch (cannyEdgeD.cu:295-377) with every block reaching its
   * local fixpoint; returns the number of blocks that modified something (isImageModified). */
  int modified_blocks = 0;
  int tw = tile + 2;
  uint8_t *s = (uint8_t *)malloc((size_t)tw * tw);
  for (int by = 0; by * tile < h; ++by)
    for (int bx = 0; bx * tile < w; ++bx) {
      for (int ty = 0; ty < tw; ++ty)
        for (int tx = 0; tx < tw; ++tx) {
          int r = by * tile + ty - 1, c = bx * tile + tx - 1;
          s[ty * tw + tx] = (r >= 0 && r < h && c >= 0 && c < w) ? in[(size_t)r * w + c] : 0;
        }
      int any = 0, changed = 1;
      while (changed) {
        changed = 0;
        for (int ty = 1; ty <= tile; ++ty)
          for (int tx = 1; tx <= tile; ++tx) {
            if (s[ty * tw + tx] != ORC_CANDIDATE_EDGE) continue;
            int f = 0;
            for (int dy = -1; dy <= 1 && !f; ++dy)
              for (int dx = -1; dx <= 1; ++dx)
                if (s[(ty + dy) * tw + tx + dx] == ORC_FINAL_EDGE) { f = 1; break; }
            if (f) { s[ty * tw + tx] = ORC_FINAL_EDGE; changed = 1; any = 1; }
          }
      }
      modified_blocks += any;
      for (int ty = 1; ty <= tile; ++ty)
        for (int tx = 1; tx <= tile; ++tx) {
          int r = by * tile + ty - 1, c = bx * tile + tx - 1;
          if (r < h && c < w) out[(size_t)r * w + c] = s[ty * tw + tx];
        }
    }
  free(s);
  return modified_blocks;
}

void orc_hysteresis_tiled(const uint8_t *thr, size_t tstride, int w, int h, uint8_t *out, size_t ostride,
                          int tile, int max_extra, int *launches)
{
  size_t n = (size_t)w * h;
  uint8_t *a = (uint8_t *)malloc(n ? n : 1), *b = (uint8_t *)malloc(n ? n : 1), *t0 = (uint8_t *)malloc(n ? n : 1);
  for (int r = 0; r < h; ++r) memcpy(t0 + (size_t)r * w, thr + (size_t)r * tstride, (size_t)w);
  uint8_t *hyster = a, *hysterTemp = b, *tmp;
  int nl = 1;
  int modified = hyst_launch(t0, hyster, w, h, tile); /* cannyEdgeH.cu:309 */
  int iters = 0;
  while (iters < max_extra && modified) {             /* :314 */
    tmp = hyster; hyster = hysterTemp; hysterTemp = tmp; /* :316 */
    modified = hyst_launch(hysterTemp, hyster, w, h, tile); /* :321 */
    ++iters; ++nl;
  }
  tmp = hyster; hyster = hysterTemp; hysterTemp = tmp;   /* :328 */
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {                        /* removeCandidates :331-333 */
      uint8_t v = hysterTemp[(size_t)r * w + c];
      out[(size_t)r * ostride + c] = v == ORC_CANDIDATE_EDGE ? 0 : v;
    }
  if (launches) *launches = nl;
  free(a); free(b); free(t0);
}

void orc_grad_display(const int16_t *sumx, const int16_t *sumy, size_t sstride, int w, int h, uint8_t *out, size_t ostride)
{
  /* float2uchar cannyEdgeD.cu:48: (unsigned char)min(abs(in), 255.0f) -- saturates (no wrap here) */
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      float g = orc_grad_float(sumx[(size_t)r * sstride + c], sumy[(size_t)r * sstride + c]);
      g = fabsf(g);
      if (g > 255.0f) g = 255.0f;
      out[(size_t)r * ostride + c] = (uint8_t)(int)g;
    }
}

int orc_canny_r(const uint8_t *in, size_t stride, int w, int h, int channels, int low, int high, int saturate, orc_outputs *o)
{
  if (w <= 0 || h <= 0 || (channels != 1 && channels != 3)) return -1;
  size_t n = (size_t)w * h;
  uint8_t *mono = (uint8_t *)malloc(n), *blur = (uint8_t *)malloc(n), *nms = (uint8_t *)malloc(n), *thr = (uint8_t *)malloc(n);
  uint8_t *edges = (uint8_t *)malloc(n);
  int16_t *sx = (int16_t *)malloc(n * 2), *sy = (int16_t *)malloc(n * 2);
  if (channels == 3) orc_gray_bgr(in, stride, w, h, mono, (size_t)w);
  else for (int r = 0; r < h; ++r) memcpy(mono + (size_t)r * w, in + (size_t)r * stride, (size_t)w); /* stage 0 skipped */
  orc_gaussian_shortcut(mono, (size_t)w, w, h, blur, (size_t)w); /* == orc_gaussian(fused=1), tested */
  orc_sobel(blur, (size_t)w, w, h, sx, sy, (size_t)w);
  orc_nms(sx, sy, (size_t)w, w, h, nms, (size_t)w, saturate);
  orc_threshold(nms, (size_t)w, w, h, low, high, thr, (size_t)w);
  orc_hysteresis(thr, (size_t)w, w, h, edges, (size_t)w);
  if (o) {
    if (o->mono) memcpy(o->mono, mono, n);
    if (o->blur) memcpy(o->blur, blur, n);
    if (o->sumx) memcpy(o->sumx, sx, n * 2);
    if (o->sumy) memcpy(o->sumy, sy, n * 2);
    if (o->grad_disp) orc_grad_display(sx, sy, (size_t)w, w, h, o->grad_disp, (size_t)w);
    if (o->nms) memcpy(o->nms, nms, n);
    if (o->thresh) memcpy(o->thresh, thr, n);
    if (o->edges) memcpy(o->edges, edges, n);
  }
  free(mono); free(blur); free(nms); free(thr); free(edges); free(sx); free(sy);
  return 0;
}

int orc_canny_r_batch(const uint8_t *in, int w, int h, int nframes, int low, int high, uint8_t *edges, int threads)
{
  size_t n = (size_t)w * h;
  int rc = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int f = 0; f < nframes; ++f) {
    orc_outputs o;
    memset(&o, 0, sizeof o);
    o.edges = edges + n * (size_t)f;
    if (orc_canny_r(in + n * (size_t)f, (size_t)w, w, h, 1, low, high, 0, &o)) rc = -1;
  }
  return rc;
}

/* ============================ Mode O: cv::Canny restatement ================================== */
/* OpenCV 4.x modules/imgproc/src/canny.cpp (parallelCanny, non-SIMD path), restated from the
 * published algorithm; opencv is a Conan dependency of the reference (conanfile.py:22,
 * "opencv/[>=4.5.3]"), is never called by the reference for Canny and is not installed here:
 * PARITY UNPINNED for this mode. */
static inline int pxr(const uint8_t *img, size_t stride, int w, int h, int cn, int r, int c, int k)
{
  if (r < 0) r = 0; if (r >= h) r = h - 1; /* BORDER_REPLICATE */
  if (c < 0) c = 0; if (c >= w) c = w - 1;
  return img[(size_t)r * stride + (size_t)c * cn + k];
}

int orc_canny_o(const uint8_t *in, size_t stride, int w, int h, int channels, double low_thresh, double high_thresh,
                int l2gradient, uint8_t *edges)
{
  return orc_canny_o_ex(in, stride, w, h, channels, low_thresh, high_thresh, l2gradient, edges, NULL);
}

/* premap (optional): the map as it stands after non-maximum suppression and the two thresholds, before the flood
 * (canny.cpp's `map` with 2 -> 255 seed, 0 -> 128 candidate, 1 -> 0): what the HIP front kernel's bit planes hold. */
int orc_canny_o_ex(const uint8_t *in, size_t stride, int w, int h, int channels, double low_thresh, double high_thresh,
                   int l2gradient, uint8_t *edges, uint8_t *premap)
{
  if (w <= 0 || h <= 0 || (channels != 1 && channels != 3)) return -1;
  if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
  if (l2gradient) {
    if (low_thresh > 32767.0) low_thresh = 32767.0;
    if (high_thresh > 32767.0) high_thresh = 32767.0;
    if (low_thresh > 0) low_thresh *= low_thresh;
    if (high_thresh > 0) high_thresh *= high_thresh;
  }
  int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
  size_t n = (size_t)w * h;
  int *mag = (int *)malloc(n * sizeof(int));
  int16_t *dx = (int16_t *)malloc(n * 2), *dy = (int16_t *)malloc(n * 2);
  uint8_t *map = (uint8_t *)malloc(n); /* 0 candidate, 1 non-edge, 2 edge */
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      int best = -1, bx = 0, by = 0;
      for (int k = 0; k < channels; ++k) {
#define P(dr, dc) pxr(in, stride, w, h, channels, r + (dr), c + (dc), k)
        int gx = (P(-1, 1) + 2 * P(0, 1) + P(1, 1)) - (P(-1, -1) + 2 * P(0, -1) + P(1, -1));
        int gy = (P(1, -1) + 2 * P(1, 0) + P(1, 1)) - (P(-1, -1) + 2 * P(-1, 0) + P(-1, 1));
#undef P
        int m = l2gradient ? gx * gx + gy * gy : abs(gx) + abs(gy);
        if (m > best) { best = m; bx = gx; by = gy; } /* first channel with the largest magnitude */
      }
      mag[(size_t)r * w + c] = best; dx[(size_t)r * w + c] = (int16_t)bx; dy[(size_t)r * w + c] = (int16_t)by;
    }
#define M(r, c) (((r) >= 0 && (r) < h && (c) >= 0 && (c) < w) ? mag[(size_t)(r) * w + (c)] : 0)
  int *stack = (int *)malloc(sizeof(int) * n);
  size_t sp = 0;
  const int TG22 = 13573; /* (int)(0.4142135623730950488016887242097*(1<<15) + 0.5) */
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      int m = mag[(size_t)r * w + c];
      int keep = 0;
      if (m > low) {
        int xs = dx[(size_t)r * w + c], ys = dy[(size_t)r * w + c];
        int x = abs(xs), y = abs(ys) << 15;
        int tg22x = x * TG22;
        if (y < tg22x) {
          keep = (m > M(r, c - 1) && m >= M(r, c + 1));
        } else {
          int tg67x = tg22x + (x << 16);
          if (y > tg67x) keep = (m > M(r - 1, c) && m >= M(r + 1, c));
          else {
            int s = (xs ^ ys) < 0 ? -1 : 1;
            keep = (m > M(r - 1, c - s) && m > M(r + 1, c + s));
          }
        }
      }
      if (!keep) map[(size_t)r * w + c] = 1;
      else if (m > high) { map[(size_t)r * w + c] = 2; stack[sp++] = r * w + c; }
      else map[(size_t)r * w + c] = 0;
    }
#undef M
  if (premap)
    for (size_t i = 0; i < n; ++i) premap[i] = map[i] == 2 ? 255 : map[i] == 0 ? 128 : 0;
  while (sp) {
    int idx = stack[--sp];
    int r = idx / w, c = idx % w;
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        if (rr < 0 || rr >= h || cc < 0 || cc >= w) continue;
        if (map[(size_t)rr * w + cc] == 0) { map[(size_t)rr * w + cc] = 2; stack[sp++] = rr * w + cc; }
      }
  }
  for (size_t i = 0; i < n; ++i) edges[i] = (uint8_t)(-(map[i] >> 1)); /* 2 -> 255, else 0 */
  free(mag); free(dx); free(dy); free(map); free(stack);
  return 0;
}

int orc_canny_o_batch(const uint8_t *in, int w, int h, int nframes, double low, double high, int l2gradient,
                      uint8_t *edges, int threads)
{
  size_t n = (size_t)w * h;
  int rc = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int f = 0; f < nframes; ++f)
    if (orc_canny_o(in + n * (size_t)f, (size_t)w, w, h, 1, low, high, l2gradient, edges + n * (size_t)f)) rc = -1;
  return rc;
}

/* ============================ exhaustive self-checks (used by tests/) ======================== */
/* Every reachable Sobel pair: |sumx|,|sumy| <= 1020.  Counts pairs where the kernel form of the
 * direction bin differs from the integer rule (excluding (0,0)), and lists the pairs where this
 * libm's literal float formula differs from the integer rule (SURVEY App. C.2 found two). */
int orc_check_dir_all(int *mism_kernel, int *mism_float, int16_t *float_pairs, int max_pairs)
{
  int mk = 0, mf = 0;
  for (int sx = -1020; sx <= 1020; ++sx)
    for (int sy = -1020; sy <= 1020; ++sy) {
      int b = orc_dir_bin(sx, sy);
      if ((sx || sy) && orc_dir_bin_kernel(sx, sy) != b) ++mk;
      if (orc_dir_bin_float(sx, sy) != b) {
        if (mf < max_pairs) { float_pairs[2 * mf] = (int16_t)sx; float_pairs[2 * mf + 1] = (int16_t)sy; }
        ++mf;
      }
    }
  *mism_kernel = mk;
  *mism_float = mf;
  return 0;
}

/* For every reachable pair: trunc of the literal float gradient == isqrt(S>>2); and the float
 * gradient is a strictly increasing function of S (so float comparisons == integer comparisons).
 * Returns the number of violations. */
int orc_check_grad_all(void)
{
  const unsigned SMAX = 2u * 1020u * 1020u;
  float *gof = (float *)calloc(SMAX + 1, sizeof(float));
  unsigned char *seen = (unsigned char *)calloc(SMAX + 1, 1);
  int bad = 0;
  for (int sx = 0; sx <= 1020; ++sx)
    for (int sy = 0; sy <= 1020; ++sy) {
      unsigned S = (unsigned)(sx * sx + sy * sy);
      float g = orc_grad_float(sx, sy);
      if ((int)g != orc_grad_trunc_int(sx, sy)) ++bad;
      if (seen[S] && gof[S] != g) ++bad; /* same S must give the same float */
      seen[S] = 1; gof[S] = g;
    }
  float prev = -1.0f;
  for (unsigned S = 0; S <= SMAX; ++S)
    if (seen[S]) { if (!(gof[S] > prev)) ++bad; prev = gof[S]; }
  free(gof); free(seen);
  return bad;
}

/* Random 5x5 patches: trunc(fused chain) vs floor(S/159).  Returns the number of patches where
 * they differ although S % 159 != 0 (must be 0), and reports how many differ in total. */
long orc_check_gauss_random(unsigned long long seed, long npatches, long *ndiff_total, long *nmultiples)
{
  float gk[25];
  orc_gauss_coeffs(gk);
  long bad = 0, diff = 0, mult = 0;
  unsigned long long s = seed;
  uint8_t p[25];
  for (long i = 0; i < npatches; ++i) {
    for (int k = 0; k < 25; k += 8) {
      s += 0x9E3779B97F4A7C15ull;
      unsigned long long z = s;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      for (int j = 0; j < 8 && k + j < 25; ++j) p[k + j] = (uint8_t)(z >> (8 * j));
    }
    /* bias some patches towards saturated / low-entropy content */
    if ((i & 7) == 1) for (int k = 0; k < 25; ++k) p[k] = p[k] > 127 ? 255 : 0;
    if ((i & 7) == 2) for (int k = 0; k < 25; ++k) p[k] = (uint8_t)(p[0] + (p[k] & 3));
    unsigned S = 0;
    for (int k = 0; k < 25; ++k) S += (unsigned)K5[k] * p[k];
    int v = gauss_chain_at(p, 5, 5, 5, 2, 2, gk, 1);
    int q = (int)(S / 159u);
    if (S % 159u == 0) ++mult;
    if (v != q) { ++diff; if (S % 159u != 0) ++bad; else if (v != q - 1) ++bad; }
  }
  if (ndiff_total) *ndiff_total = diff;
  if (nmultiples) *nmultiples = mult;
  return bad;
}
