/*
 * canny_oracle.h -- CPU restatement of CudaCam's Canny pipeline (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle.  It is NOT part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libhipcanny.so) never links or
 * calls anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" by the reference's own tests -- the reference ships no golden
 * vectors for this path (its test/ directory only tests a Factorial placeholder).  The oracle is
 * pinned instead by (i) the known-answer values recorded in SURVEY.md App. C.4 (tests/golden/
 * survey_kat.json), (ii) exhaustive domain checks of every integer shortcut against the literal
 * float formulas of the reference (tests/test_oracle_exhaustive.py), (iii) on the GPU box, the
 * reference's own device kernels compiled in place by oracle/build_ref.sh (oracle/_ref/) and (iv) the
 * committed outputs of those kernels for 17 cases, every stage (tests/golden/ref_kernels_*.npz, checked on
 * CPU by tests/test_oracle_golden_ref.py).  Those kernels are the reference's source compiled by hipcc, not
 * nvcc, and launched by a restatement of the reference's host code: see DESIGN.md section 5 for what that
 * does and does not pin.
 *
 * "Mode R" = reference-exact (src/cvp/cannyEdgeD.cu).  "Mode O" = OpenCV cv::Canny restatement.
 * All images are tightly described by (pointer, row stride in elements, width, height).
 */
#ifndef CANNY_ORACLE_H
#define CANNY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Edge codes, reference cannyEdgeD.cu:31-33 */
#define ORC_FINAL_EDGE 255
#define ORC_CANDIDATE_EDGE 128
#define ORC_NO_EDGE 0

/* ---- Mode R stages ------------------------------------------------------------------------ */

/* cannyEdgeD.cu:14-19,53-69: mono = min(255,(p0*7 + p1*38 + p2*19) >> 6), p0 = first byte (B). */
void orc_gray_bgr(const uint8_t *bgr, size_t stride, int w, int h, uint8_t *mono, size_t mstride);

/* cannyEdgeH.cu:372-380: GK[i][j] = (float)K[i][j] * (1 / 159.0f), all in binary32. */
void orc_gauss_coeffs(float gk[25]);

/* cannyEdgeD.cu:72-118: zero padded 5x5, 25-term chain starting at 0.0f in r-major/c-minor order,
 * truncated to u8.  fused != 0: each step is fmaf() (nvcc default -fmad=true, canonical Mode R);
 * fused == 0: separately rounded multiply and add (investigation variant). */
void orc_gaussian(const uint8_t *in, size_t istride, int w, int h, uint8_t *out, size_t ostride, int fused);

/* Same result as orc_gaussian(fused=1) but by the integer shortcut the HIP kernel uses:
 * floor(S/159) with S = sum K*x, falling back to the float chain only when S % 159 == 0. */
void orc_gaussian_shortcut(const uint8_t *in, size_t istride, int w, int h, uint8_t *out, size_t ostride);

/* cannyEdgeD.cu:121-172: integer 3x3 sums before the /8.0f (zero padded). */
void orc_sobel(const uint8_t *blur, size_t bstride, int w, int h, int16_t *sumx, int16_t *sumy, size_t sstride);

/* Direction bin 0..3 of cannyEdgeD.cu:239-264 by the exact integer rule. */
int orc_dir_bin(int sumx, int sumy);
/* The same bin by the literal float formulas (atan2f of this libm); for cross-checking only. */
int orc_dir_bin_float(int sumx, int sumy);
/* The third form: what the HIP kernel evaluates (P = 2ab against D = a^2-b^2). */
int orc_dir_bin_kernel(int sumx, int sumy);

/* trunc(GRAD_COEFF * sqrtf(sX*sX + sY*sY)) by literal float ops (cannyEdgeD.cu:195), and by
 * integer isqrt((sumx^2+sumy^2) >> 2); both return the untruncated integer part 0..721. */
int orc_grad_trunc_float(int sumx, int sumy);
int orc_grad_trunc_int(int sumx, int sumy);
/* The float gradient itself (for the GRADIENT display stage, float2uchar cannyEdgeD.cu:35-50). */
float orc_grad_float(int sumx, int sumy);

/* cannyEdgeD.cu:201-270: keep iff q <= g && r <= g along the bin; out = (u8)(int)g: wraps mod 256
 * (canonical) or, with saturate != 0, min(g,255) (the hipcc lowering of the same line, see .c). */
void orc_nms(const int16_t *sumx, const int16_t *sumy, size_t sstride, int w, int h, uint8_t *nms, size_t nstride, int saturate);

/* cannyEdgeD.cu:273-293 */
void orc_threshold(const uint8_t *nms, size_t nstride, int w, int h, int low, int high, uint8_t *thr, size_t tstride);

/* cannyEdgeD.cu:295-395 + cannyEdgeH.cu:297-338: full fixpoint (queue flood fill), then 128 -> 0.
 * Returns the number of candidate pixels promoted. */
long orc_hysteresis(const uint8_t *thr, size_t tstride, int w, int h, uint8_t *out, size_t ostride);

/* Literal emulation of the reference's launch loop: 30x30 output tiles each iterated to a local
 * fixpoint per launch, at most 1 + max_extra launches (reference: 100), then 128 -> 0.
 * *launches receives the number of hysteresis launches performed. */
void orc_hysteresis_tiled(const uint8_t *thr, size_t tstride, int w, int h, uint8_t *out, size_t ostride,
                          int tile, int max_extra, int *launches);

/* GRADIENT display plane: (u8)min(|grad|, 255.0f) (float2uchar, cannyEdgeD.cu:35-50). */
void orc_grad_display(const int16_t *sumx, const int16_t *sumy, size_t sstride, int w, int h, uint8_t *out, size_t ostride);

/* Whole pipeline.  channels = 1 (stage 0 skipped) or 3 (BGR interleaved).  Any of the stage
 * outputs may be NULL.  All outputs are tight (stride = w).  final map in `edges`. */
typedef struct {
  uint8_t *mono, *blur, *grad_disp, *nms, *thresh, *edges;
  int16_t *sumx, *sumy;
} orc_outputs;
int orc_canny_r(const uint8_t *in, size_t stride, int w, int h, int channels, int low, int high, int saturate, orc_outputs *o);

/* Batch helper for the CPU baseline: nframes tight mono frames -> edge maps, `threads` OpenMP threads. */
int orc_canny_r_batch(const uint8_t *in, int w, int h, int nframes, int low, int high, uint8_t *edges, int threads);

/* ---- Mode O: cv::Canny(src 8UC1, low, high, apertureSize=3, L2gradient) restatement ---------- */
/* OpenCV 4.x modules/imgproc/src/canny.cpp semantics (NOT under /root/reference; unpinned). */
int orc_canny_o(const uint8_t *in, size_t stride, int w, int h, int channels, double low, double high,
                int l2gradient, uint8_t *edges);
/* The same, also returning the tri-state map before the flood (NULL: not wanted): 255 seed, 128 candidate, 0 none. */
int orc_canny_o_ex(const uint8_t *in, size_t stride, int w, int h, int channels, double low, double high,
                   int l2gradient, uint8_t *edges, uint8_t *premap);
int orc_canny_o_batch(const uint8_t *in, int w, int h, int nframes, double low, double high, int l2gradient,
                      uint8_t *edges, int threads);

/* ---- exhaustive self-checks used by tests/ (see canny_oracle.c) ---------------------------- */
int orc_check_dir_all(int *mism_kernel, int *mism_float, int16_t *float_pairs, int max_pairs);
int orc_check_grad_all(void);
long orc_check_gauss_random(unsigned long long seed, long npatches, long *ndiff_total, long *nmultiples);

#ifdef __cplusplus
}
#endif
#endif
