// front8.hip -- k_front8: the WHOLE front path of the detector as one kernel, 8 pixels per lane (Mode R); and k_front8o, the
// cv::Canny-semantics (Mode O) kernel on the same skeleton (second half of the file).
//
// Replaces, in one launch and with no intermediate plane in HBM, the reference's rgb2mono, gaussianFilter5x5, sobelXY,
// gradSlope, nonMaxSuppr and doubleThreshold kernels (src/cvp/cannyEdgeD.cu:53-293; launch sites
// src/cvp/cannyEdgeH.cu:214-295): input frames in, the STRONG / CANDIDATE bit planes (and, in pipelined mode, the
// provisional 0/255 edge map) out.  HBM traffic per pixel: 1 B read, 1/4 B of bit planes written, 1 B of provisional
// map written -- against 25 B/px of intermediates in the reference and 4.4 B/px more in the k_blur + k_nms pair
// (canny_kernels.hip), whose arithmetic this kernel shares.
//
// Work decomposition.  A wave owns a vertical STRIP of 496 output columns and a RUN of rows it marches down.  Lane l
// holds the 8 adjacent pixels at columns strip*496 - 8 + 8l .. +7 of the current row (two dwords); lanes 0 and 63 are
// halo lanes (of their 8 columns only the inner 2 are ever consumed).  Horizontal neighbours inside a lane come from
// its own registers, across lanes by two DPP wave shifts per stage -- half the exchanges, address arithmetic, ballots
// and scalar bookkeeping per pixel of the 4-px kernels.  A lane's 8 pixels are exactly one byte of each bit plane, so a
// wave-row stores 62 contiguous bytes per plane and 496 contiguous bytes of provisional map.
//
// The run proceeds in WINDOWS of 6 rows; within a window every phase is straight-line code (no per-row branches), so
// that the compiler can overlap the rows' dependency chains -- the kernel is bound by instruction issue and latency, not
// by HBM:
//   phase 1   6 input rows -> vertical accumulators -> exact integer quotient floor(S/159) -> 6 blur rows into a
//             wave-private LDS ring of the last 10 blur rows; pixels whose float result cannot be decided by integers
//             (S % 159 == 0, 0.6 % of them) are queued
//   fix-up    the queued pixels get the literal 25-fmaf chain of the reference, read from a wave-private LDS ring of
//             the last 10 masked input rows, and overwrite their byte of the blur ring
//   phase 2   6 blur rows -> Sobel -> sumX^2 + sumY^2, summed over pixel pairs -> low-threshold test (a necessary
//             condition: the batches decide exactly).  Every byte of the output row is stored as zero at once; the
//             few half-lanes (4 px) that may hold a candidate only queue their identity (row, lane, half: one dword)
//   NMS       dense batches of 64 queued half-lanes, one per lane: each re-derives the 3 x 6 S2 values around its
//             4 pixels from the blur ring (5 rows x 12 bytes), applies direction, non-maximum suppression and the two
//             thresholds, and overwrites its nibble of the planes (and its 4 bytes of the provisional map).  About
//             6 % of the half-lanes hold a candidate, so redoing their Sobel costs less than carrying three rows of
//             S2 for everybody in registers -- or parking 96 bytes per candidate in LDS, which is what made the first
//             form of this kernel LDS-write-bound.
// Only the vertical blur accumulators (16 VGPRs) and the Sobel terms of the two previous blur rows (16 VGPRs) are
// carried from row to row.  Input rows are requested one window ahead.
#include "canny_device.h"
#include <type_traits>

#ifndef F8_ABL
#define F8_ABL 0  // timing measurements with parts left out (tools/build_variant.sh, WRONG results): 1 no NMS batches, 2 no fix-up, 4 no phase 2 (Sobel, test, queue) and no batches, 8 no zero stores of the provisional map
#endif

namespace hc {

constexpr int F8_STRIP_W = 62 * 8;                 // 496 output columns per wave
constexpr int F8_HSTRIP_W = 30 * 8;                // HALF form: 240 output columns per half-wave (lanes 0 / 31 and 32 / 63 are its halo lanes)
constexpr int F8_HALO = 8;                         // one lane each side
constexpr int F8_SUB = 6;                          // rows per window = lcm(2, 3) rows: the d / s register ring has period 2
constexpr int F8_RING = F8_SUB + 4;                // rows kept in each LDS ring (masked input rows; blur rows)
constexpr int F8_FQ = 128;                         // flagged-pixel queue entries per window (expected fill ~20); [F8_FQ] is a dump slot
constexpr int F8_NQ = 512;                         // NMS queue (ids), circular; [F8_NQ] is a dump slot
constexpr int F8_ROW_BYTES = 64 * 8;
constexpr int F8_WPB = 4;                          // waves per workgroup of k_front8 (mono / BGR; one-wave form: 1; per-channel mode: 3, one per channel).  The waves are independent.
constexpr int F8_WAVE_BYTES = 2 * F8_RING * F8_ROW_BYTES + (F8_FQ + 4) * 4 + (F8_NQ + 4) * 4;  // 12,832 B: 3 workgroups of 4 waves per CU

int front8_run_rows(int windows) { return F8_SUB * windows - 4; }
int front8_strips(int W) { return (W + F8_STRIP_W - 1) / F8_STRIP_W; }
int front8_half_strips(int W) { return (W + F8_HSTRIP_W - 1) / F8_HSTRIP_W; }
size_t front8_lds_bytes() { return (size_t)4 * F8_WAVE_BYTES; }

typedef __attribute__((address_space(3))) u32 lds_u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// global memory: caller buffers are only promised to be 4-byte aligned (pointer, pitch, frame stride)
typedef u32x2 __attribute__((aligned(4))) g_u32x2;
typedef u32x4 __attribute__((aligned(4))) g_u32x4;

// the literal reference chain (cannyEdgeD.cu:102-115) on the LDS ring of masked input rows: rows and columns outside the
// image are stored as 0 there, and a 0 tap leaves the running sum unchanged exactly as the reference's skipped taps do.
// slot0: ring slot of input row (blur row - 2), the first of the five
static __device__ __forceinline__ u32 f8_chain(const unsigned char *ring, u32 slot0, u32 colbyte)
{
  u32 px[25];  // all 25 taps are requested before the first is used: one LDS wait instead of 25 dependent ones
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    u32 sl = slot0 + (u32)r;
    sl = min(sl, sl - (u32)F8_RING);  // wrap (unsigned): slot0 < F8_RING
    const unsigned char *q = ring + sl * (u32)F8_ROW_BYTES + colbyte - 2u;
#pragma unroll
    for (int c = 0; c < 5; ++c) px[r * 5 + c] = q[c];
  }
  asm volatile("" ::: "memory");  // keeps the loads above the chain
  float f = 0.0f;
#pragma unroll
  for (int i = 0; i < 25; ++i) f = __builtin_fmaf(GKC.v[i], (float)px[i], f);
  return (u32)(int)f;
}

// IN: 0 mono plane, 1 interleaved BGR -> grey (stage 0 fused into the load), 2 one channel of interleaved 3-channel data
// PROV: also write the provisional 0/255 map (strong pixels) that the hysteresis then only patches (pipelined mode)
// HALF: narrow frames.  A wave of the plain form spans 496 columns whether they exist or not -- 640 columns cost two waves, 62.5 %
// of their lanes.  In the HALF form a wave is TWO independent half-waves of 30 own lanes + 2 halo lanes each (240 columns):
// the units (frame, half-strip) of one run of rows are dealt to the half-waves in pairs -- 640 columns = 3 units = 1.5 waves --
// so the two halves of a wave may belong to different strips and to different (adjacent) frames.  Rows, and with them all
// control flow, stay wave-uniform; what depends on the strip or the frame becomes a per-lane offset.  The DPP wave shifts
// need no fence between lanes 31 and 32: both are halo lanes, whose far neighbour is never consumed.
// ONE: one-wave workgroups (mono / BGR).  A retiring wave's slot is then refilled at once -- a four-wave workgroup needs a free
// slot on each of the CU's four SIMDs at the same moment -- which gains the front kernel 2-3 % beside the hysteresis and
// costs the hysteresis 40 % more stream time: the host picks it while that stream has the slack (hipcanny.hip, watch_chain).
template <int IN, bool PROV, bool HALF, bool ONE = false>
__global__ __launch_bounds__(256) void k_front8(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  unsigned char *ring = smem + wib * F8_WAVE_BYTES;           // masked input rows (fix-up chain)
  unsigned char *bring = ring + F8_RING * F8_ROW_BYTES;       // blur rows (Sobel stage, NMS batches)
  u32 *fq = reinterpret_cast<u32 *>(bring + F8_RING * F8_ROW_BYTES);
  lds_u32 *nq = (lds_u32 *)(fq + F8_FQ + 4);

  // Mono / BGR: 4 independent waves per workgroup.  Per-channel mode: a workgroup is the THREE channels of one (frame,
  // strip, run), a wave each, kept within one window of each other by a barrier per window -- so the 24 interleaved bytes
  // per lane and row are fetched from HBM once and served to the other two waves from the CU's L1 / the XCD's L2.
  // (Without the barrier the three drifted apart and the input was fetched about twice: 5.5 GB per 16 8K frames.)
  constexpr int WPB = IN == 2 ? 3 : ONE ? 1 : F8_WPB;
  // the flag words of this run's hysteresis (a few hundred to 140 k dwords), zeroed by the first workgroups on their way in
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_count; i += gridDim.x * blockDim.x) p.zero_words[i] = 0u;
  int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPB + wib);
  if (item >= p.total_items) return;  // (per-channel: total_items is a multiple of 3, a workgroup leaves as a whole)
  int ch = 0;
  if (IN == 2) { ch = item % 3; item /= 3; }
  const int chunk = item % p.nchunks;
  const int W = p.W, H = p.H;
  // plain form: (strip, frame) of the wave.  HALF form: those of half-wave A (lanes 0..31); half-wave B (lanes 32..63)
  // takes the next unit in (frame, half-strip) order -- sB / dframe -- or, beyond the last unit, a strip right of the image
  int strip, in_frame, sB = 0, dframe = 0;
  if constexpr (HALF) {
    const int uA = 2 * (item / p.nchunks), uB = uA + 1;
    in_frame = uA / p.nhalf; strip = uA % p.nhalf;
    if (uB < p.nhalf * (IN == 2 ? p.nframes / 3 : p.nframes)) { sB = uB % p.nhalf; dframe = uB / p.nhalf - in_frame; }
    else { sB = p.nhalf; dframe = 0; }
  } else {
    strip = (item / p.nchunks) % p.nstrips;
    in_frame = item / (p.nchunks * p.nstrips);
  }
  const int frame = IN == 2 ? in_frame * 3 + ch : in_frame;  // output frame = bit-plane index (of half-wave A)
  const int r0 = chunk * p.run_rows;  // output rows [r0, rend)
  const int rend = min(r0 + p.run_rows, H);
  const bool hB = HALF && lane >= 32;        // this lane belongs to half-wave B
  const int ll = HALF ? (lane & 31) : lane;  // lane within its (half-)wave
  const int c0 = HALF ? (hB ? sB : strip) * F8_HSTRIP_W - F8_HALO + ll * 8 : strip * F8_STRIP_W - F8_HALO + lane * 8;
  // per-lane offsets of half-wave B's frame (0 in the plain form): launch_front8 checks that they fit 32 bits
  const u32 d_in = hB ? (u32)dframe * (u32)p.in_frame_stride : 0u;
  const u32 d_plane = hB ? (u32)(IN == 2 ? 3 * dframe : dframe) * (u32)H * (u32)p.RD * 4u : 0u;
  const u32 d_prov = (PROV && hB) ? (u32)(IN == 2 ? 3 * dframe : dframe) * (u32)p.prov_fs : 0u;

  // per-lane column validity: byte masks of the two packed u8 dwords
  u32 cmask[2] = { 0, 0 };
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    cmask[k >> 2] |= in ? (0xFFu << (8 * (k & 3))) : 0u;
  }
  constexpr int LAST = HALF ? 31 : 63;  // the right halo lane of a (half-)wave
  const bool own_lane = ll >= 1 && ll <= LAST - 1;
  // "undecidable" flag positions this lane answers for: its own 8 pixels; in the halo lanes the two next to the strip
  const u32 hmask0 = cmask[0] & (own_lane ? 0x80808080u : ll == LAST ? 0x00008080u : 0u);
  const u32 hmask1 = cmask[1] & (own_lane ? 0x80808080u : ll == 0 ? 0x80800000u : 0u);
  const bool col_any = (cmask[0] | cmask[1]) != 0;
  const uint8_t *frame_base = p.in + (size_t)in_frame * p.in_frame_stride;
  const u32 in_pitch32 = (u32)p.in_pitch;  // launch_front8 checks H * pitch < 2^32 and pitch >= (IN ? 3 : 1) * round_up(W, 8)
  // lanes without an image column read the row's first bytes (masked).  (HALF: rows outside the image are read at the same
  // offset from the page of zeros, which launch_front8 requires to be a frame stride longer than in the plain form.)
  const u32 ld_safe = (col_any ? (u32)((IN ? 3 : 1) * c0) : 0u) + d_in;
  const u32 selA = ch == 0 ? 0x0c060300u : ch == 1 ? 0x0c070401u : 0x0c0c0502u;
  const u32 selB = ch == 0 ? 0x05020100u : ch == 1 ? 0x06020100u : 0x07040100u;

  // ---- input rows ----------------------------------------------------------------------------------------------------
  constexpr int ND = IN == 0 ? 2 : 6;      // dwords per lane and row
  constexpr int G = IN == 0 ? F8_SUB : 2;  // rows requested ahead (3-channel rows are 6 dwords each: fewer in flight)
  struct Raw { u32 d[ND]; };
  const int rlast = min(H - 1, rend + 3);  // last input row this run needs
  auto load_raw = [&](int row) -> Raw {  // unconditional: rows clamped, masked when used (a conditional load would make every wait a full drain)
    u32 lo = ld_safe;
    asm volatile("" : "+v"(lo));  // keeps the lane offset out of a hoisted 64-bit VGPR pointer
    // a row above / below the image is zero padding (cannyEdgeD.cu:91-98): read from a page of zeros instead (wave-uniform select)
    const uint8_t *q = ((u32)row < (u32)H ? frame_base + (u32)min(row, rlast) * in_pitch32 : p.zeros) + lo;
    Raw r;
    if constexpr (IN == 0) {
      const u32x2 t = *reinterpret_cast<const g_u32x2 *>(q);
      r.d[0] = t.x; r.d[1] = t.y;
    } else {
      const u32x4 t = *reinterpret_cast<const g_u32x4 *>(q);
      const u32x2 v = *reinterpret_cast<const g_u32x2 *>(q + 16);
      r.d[0] = t.x; r.d[1] = t.y; r.d[2] = t.z; r.d[3] = t.w; r.d[4] = v.x; r.d[5] = v.y;
    }
    return r;
  };
  // 12 bytes of interleaved 3-channel data -> 4 pixels: one channel (IN == 2) or the grey value (IN == 1, cannyEdgeD.cu:53-69)
  auto from3 = [&](u32 d0, u32 d1, u32 d2) -> u32 {
    if (IN == 2) return __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, selA), selB);
    const u32 wts = 0x00132607u;  // (7, 38, 19, 0): sum 64, so the reference's min(255, .) never triggers
    const u32 m0 = __builtin_amdgcn_udot4(d0, wts, 0u, false) >> 6;
    const u32 m1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), wts, 0u, false) >> 6;
    const u32 m2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), wts, 0u, false) >> 6;
    const u32 m3 = __builtin_amdgcn_udot4(d2 >> 8, wts, 0u, false) >> 6;
    return m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
  };
  // the row's 8 pixels, zero outside the image (zero padding of the Gaussian, cannyEdgeD.cu:91-98)
  auto use_raw = [&](int row, const Raw &r, u32 x[2]) {
    if constexpr (IN == 0) { x[0] = r.d[0]; x[1] = r.d[1]; }
    else { x[0] = from3(r.d[0], r.d[1], r.d[2]); x[1] = from3(r.d[ND > 3 ? 3 : 0], r.d[ND > 4 ? 4 : 0], r.d[ND > 5 ? 5 : 0]); }
    x[0] &= cmask[0]; x[1] &= cmask[1];  // (rows outside the image are read from a page of zeros, see load_raw)
  };

  // ---- phase 1: the exact Gaussian (see k_blur in canny_kernels.hip for the derivation) ------------------------------
  // per input row and pixel pair: p = x[-2]+x[+2], q = x[-1]+x[+1], c = x[0];  h0 = 2p+4q+5c, h1 = 4p+9q+12c, h2 = 5p+12q+15c;
  // S(row i) = h0[i-2] + h1[i-1] + h2[i] + h1[i+1] + h0[i+2] by four running accumulators.  Every packed u16 sum stays
  // below 2^16 per half (S <= 40545), so plain 32-bit adds act on both halves at once -- and on gfx950 v_add / v_sub / v_and /
  // v_lshrrev issue at twice the rate of the v_pk_* and VOP3 forms (profiles/r01/valu_rate*.txt): h0..h2 are built from
  // additions only.
  u32 a1[4], a2[4], a3[4], a4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a1[j] = a2[j] = a3[j] = a4[j] = 0;
  auto accumulate = [&](const u32 x[2], u32 Sp[4]) {
    const u32 A0 = unpack_lo(x[0]), B0 = unpack_hi(x[0]), A1 = unpack_lo(x[1]), B1 = unpack_hi(x[1]);
    const u32 Bl = from_lane_below(B1), Ar = from_lane_above(A0);  // (x-2, x-1) and (x8, x9)
    const u32 m1 = pair_shift(A0, Bl), p1 = pair_shift(B0, A0), p3 = pair_shift(A1, B0), p5 = pair_shift(B1, A1), p7 = pair_shift(Ar, B1);
    const u32 Cc[4] = { A0, B0, A1, B1 };
    const u32 P[4] = { Bl + B0, A0 + A1, B0 + B1, A1 + Ar };
    const u32 Q[4] = { m1 + p1, p1 + p3, p3 + p5, p5 + p7 };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 c2 = Cc[j] + Cc[j];
      const u32 a = Q[j] + c2;         // q + 2c
      const u32 b = P[j] + a;          // p + q + 2c
      const u32 t = b + Q[j];          // p + 2q + 2c
      const u32 h0 = (t + t) + Cc[j];  // 2p + 4q + 5c
      const u32 h1 = (h0 + h0) + a;    // 4p + 9q + 12c
      const u32 h2 = (h0 + h1) - b;    // 5p + 12q + 15c
      Sp[j] = a4[j] + h0;
      a4[j] = a3[j] + h1;
      a3[j] = a2[j] + h2;
      a2[j] = a1[j] + h1;
      a1[j] = h0;
    }
  };
  u32 magic159 = 105518u;  // ceil(2^24 / 159)
  asm volatile("" : "+s"(magic159));  // (an SGPR operand of the SDWA multiplies, not a literal)
  int qn = 0;                        // flagged-pixel queue fill of the current window (wave-uniform)
  const u32 lane_id2 = (u32)lane << 2;
  // input row jr (its masked pixels in x) goes to ring slot `islot` and completes blur row jr - 2, which goes to blur-ring
  // slot `bslot` (warm: a warm-up row, it only feeds the accumulators)
  auto blur_row = [&](auto warm, int jr, int islot, int bslot, const u32 x[2]) {
    *reinterpret_cast<u32x2 *>(ring + islot * F8_ROW_BYTES + lane * 8) = u32x2{ x[0], x[1] };
    u32 Sp[4];
    accumulate(x, Sp);
    if constexpr (decltype(warm)::value) return;
    const int rb = jr - 2;
    const u32 rowm = (u32)rb < (u32)H ? 0xFFFFFFFFu : 0u;  // blur rows outside the image are zero padding for the Sobel stage
    // n = floor(S / 159) = (S * 105518) >> 24, exact for S <= 40545 (the product stays below 2^32), and S % 159 == 0 <=> bits
    // 16..23 of the product are all zero: byte 3 of the product IS the blur value, byte 2 the "fraction byte" whose zero
    // test flags the pixel -- one 24-bit multiply per pixel (SDWA picks the half of the packed sum) and byte permutes,
    // no shifts (round 2: two v_dot2 per pair by 52759, >> 15, an SDWA shift to merge the halves)
    u32 P[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(P[2 * j]) : "v"(Sp[j]), "s"(magic159));
      asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(P[2 * j + 1]) : "v"(Sp[j]), "s"(magic159));
    }
    // (q, q, f, f) of two pixels, then the four quotients / fractions of a dword of pixels
    const u32 X01 = __builtin_amdgcn_perm(P[1], P[0], 0x06020703u), X23 = __builtin_amdgcn_perm(P[3], P[2], 0x06020703u);
    const u32 X45 = __builtin_amdgcn_perm(P[5], P[4], 0x06020703u), X67 = __builtin_amdgcn_perm(P[7], P[6], 0x06020703u);
    const u32 qsel = (u32)rb < (u32)H ? 0x05040100u : 0x0c0c0c0cu;  // wave-uniform selector: a blur row outside the image is all zero
    const u32 bl0 = __builtin_amdgcn_perm(X23, X01, qsel) & cmask[0];
    const u32 bl1 = __builtin_amdgcn_perm(X67, X45, qsel) & cmask[1];
    const u32 fz0 = __builtin_amdgcn_perm(X23, X01, 0x07060302u), fz1 = __builtin_amdgcn_perm(X67, X45, 0x07060302u);
    // zero-byte detector (a byte equal to 1 above a zero byte may be flagged too: harmless)
    const u32 hz0 = (fz0 - 0x01010101u) & ~fz0 & hmask0, hz1 = (fz1 - 0x01010101u) & ~fz1 & hmask1;
    const u32 hz = (hz0 >> 7) | (hz1 >> 6);
    // one queue entry per flagged lane, no branch: the other lanes (and an overflowing queue) write a dump slot
    const u64 any = __ballot(hz != 0) & (u64)(int64_t)(int32_t)rowm;
    const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(any >> 32), __builtin_amdgcn_mbcnt_lo((u32)any, (u32)qn));
    // entry: bits 0/8/16/24 = pixels 0..3, bits 1/9/17/25 = pixels 4..7, bits 2..7 lane, bits 10..13 blur-ring slot
    fq[min(lane_sel(any, rank, (u32)F8_FQ), (u32)F8_FQ)] = hz | lane_id2 | ((u32)bslot << 10);
    qn += __popcll(any);
    *reinterpret_cast<u32x2 *>(bring + bslot * F8_ROW_BYTES + lane * 8) = u32x2{ bl0, bl1 };
  };

  // ---- phase 2: blur -> Sobel -> S2 -> which half-lanes hold a candidate (see k_nms for the arithmetic) ---------------
  u32 dr[2][4], sr[2][4];  // d = b[+1]-b[-1] and s = b[-1]+2b[0]+b[+1] of the two previous blur rows, [ring][pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) dr[a][b] = sr[a][b] = 0;

  const size_t plane_off = (size_t)frame * H * p.RD * 4;
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const u32 plane_pitch = (u32)p.RD * 4u;
  // the lane's 8 columns are byte c0 / 8 of a plane row.  Lanes that own no byte (halo lanes, lanes right of the image)
  // store their zero onto lane 1's byte instead: no exec masking around the stores
  const bool st_lane = own_lane && col_any;
  // (byte c0 / 8 of the plane row; the others store onto the byte of the wave's lane 1, which always owns columns)
  const u32 st_off = HALF ? (st_lane ? (u32)(c0 >> 3) + d_plane : (u32)(strip * 30)) : st_lane ? (u32)(strip * 62 + lane - 1) : (u32)(strip * 62);
  const u32 prov_voff = HALF ? (st_lane ? (u32)c0 + d_prov : (u32)(strip * F8_HSTRIP_W)) : st_lane ? (u32)c0 : (u32)(strip * F8_STRIP_W);
  const u32 a_lo0 = p.a_lo[0], a_hi0 = p.a_hi[0], wrap_limit = p.wrap_limit;
  // own lanes whose half h has a column inside the image (lanes 0 and 63 only carry halo columns)
  const u64 lanes0 = uniform64(__ballot(own_lane && cmask[0] != 0)), lanes1 = uniform64(__ballot(own_lane && cmask[1] != 0));
  uint8_t *prov_frame = PROV ? p.prov_out + (size_t)frame * p.prov_fs : nullptr;

  int qhead = 0, qcount = 0;  // NMS queue (circular, F8_NQ ids): wave-uniform
  int wq = 0;                 // half-lanes that passed the low-threshold test in the window being processed (wave-uniform)
  // blur row k (its 8 bytes per lane in b0, b1) arrives -> Sobel row k-1 -> the row's candidates are queued.  No branch.
  auto step = [&](auto uc, int k, u32 b0, u32 b1) {
    constexpr int u = decltype(uc)::value;
    constexpr int rn = u % 2, rp = (u + 1) % 2;
    const u32 A0 = unpack_lo(b0), B0 = unpack_hi(b0), A1 = unpack_lo(b1), B1 = unpack_hi(b1);
    const u32 Bl = from_lane_below(B1), Ar = from_lane_above(A0);
    const u32 m1 = pair_shift(A0, Bl), p1 = pair_shift(B0, A0), p3 = pair_shift(A1, B0), p5 = pair_shift(B1, A1), p7 = pair_shift(Ar, B1);
    const u32 Cc[4] = { A0, B0, A1, B1 };
    const u32 Lf[4] = { m1, p1, p3, p5 }, Rt[4] = { p1, p3, p5, p7 };
    u32 T2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 dk = R(I(Rt[j]) - I(Lf[j]));        // signed halves: packed op
      const u32 sk = pk_mad2(Cc[j], Lf[j] + Rt[j]);  // non-negative halves < 2^16: plain add
      const u32 X = pk_mad2(dr[rp][j], R(I(dr[rn][j]) + I(dk)));  // sumX = d[i-1] + 2 d[i] + d[i+1] (cannyEdgeD.cu:158-162)
      const u32 Y = R(I(sr[rn][j]) - I(sk));                       // sumY = s[i-1] - s[i+1]        (:163-167)
      // S2 = sumX^2 + sumY^2: the reference's float gradient 4*sqrtf((sumX/8)^2 + (sumY/8)^2) (:195) is a strictly
      // increasing function of it, so every comparison of gradients is a comparison of S2.  Here only a NECESSARY
      // condition is needed (the batch decides exactly): the S2 of the pair's two pixels summed -- one v_dot2 each for
      // sumX and sumY instead of two multiply-adds per pixel -- is at least either of them.  (Summing all four pixels
      // of the half queues enough half-lanes for nothing to cost more than the two maxima it saves.)
      T2[j] = (u32)__builtin_amdgcn_sdot2(I(Y), I(Y), sdot2_0(X, X), false);
      dr[rn][j] = dk;
      sr[rn][j] = sk;
    }
    const int c = k - 1;  // the Sobel / output row
    const bool valid = (u32)(c - r0) < (u32)(rend - r0);  // wave-uniform; rows of the neighbouring runs and beyond the image are not ours
    // half-lanes that may hold a pixel passing the low threshold (itself a necessary condition in every wrap band)
    const u32 g0 = max(T2[0], T2[1]), g1 = max(T2[2], T2[3]);
    // (pixels right of the image are not masked here: a half-lane queued for nothing costs a batch slot, the batch
    // applies the zero padding exactly; half-lanes entirely outside the image are excluded by the lane masks)
    const u64 mh0 = __ballot(g0 >= a_lo0) & (valid ? lanes0 : 0ull), mh1 = __ballot(g1 >= a_lo0) & (valid ? lanes1 : 0ull);
    // every byte of the row is zero unless an NMS batch says otherwise: stored at once by all lanes, the few others are
    // overwritten later by this same wave (stores of one wave to one address keep their order).  Rows that are not ours
    // go to a dump area.
    {
      // (typed as global pointers: from the opaque select the compiler would otherwise build 64-bit flat addresses per lane)
      typedef __attribute__((address_space(1))) uint8_t gmem_u8;
      typedef __attribute__((address_space(1))) g_u32x2 gmem_u32x2;
      gmem_u8 *sp = (gmem_u8 *)uniform_sel(valid, splane + (u32)c * plane_pitch, p.dump), *cp = (gmem_u8 *)uniform_sel(valid, cplane + (u32)c * plane_pitch, p.dump_c);
      u32 so = st_off;
      asm volatile("" : "+v"(so));
      sp[so] = 0;
      cp[so] = 0;
      if constexpr (PROV) {
        gmem_u8 *pp = (gmem_u8 *)uniform_sel(valid, prov_frame + (u32)c * p.prov_pitch, p.dump_p);
        u32 o = prov_voff;
        asm volatile("" : "+v"(o));
        if (!(F8_ABL & 8)) *reinterpret_cast<gmem_u32x2 *>(pp + o) = u32x2{ 0u, 0u };
      }
    }
    // ids of the candidate half-lanes: a lane's two halves share an output byte -- if both are queued they sit in
    // adjacent entries (half 0 first, flagged 0x80) and the batch joins them; a lone half knows the other nibble is zero
    const u32 below = mbcnt64(mh0) + mbcnt64(mh1);  // entries of lower lanes
    const u32 idbase = ((u32)c << 8) | (u32)lane;
    const u32 pos0 = ((u32)(qhead + qcount) + below) & (u32)(F8_NQ - 1);
    const u32 pos1 = (pos0 + lane_sel(mh0, 1u, 0u)) & (u32)(F8_NQ - 1);
    nq[lane_sel(mh0, pos0, (u32)F8_NQ)] = idbase | lane_sel(mh1, 0x80u, 0u);
    nq[lane_sel(mh1, pos1, (u32)F8_NQ)] = idbase | 0x40u;
    u32 n0, n1;  // as instructions: the builtin's result is widened and the additions land on the VALU
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n0) : "s"(mh0) : "scc");
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n1) : "s"(mh1) : "scc");
    qcount += (int)(n0 + n1);
    // (rows 0, 1, H-2, H-1 are not counted: the zero padding makes them candidates across the whole width of every frame)
    wq += (c >= 2 && c < H - 2) ? (int)(n0 + n1) : 0;
  };

  // One dense NMS pass: up to 64 queued half-lanes, an entry per lane.  sbase: blur-ring slot of blur row bw0 - 4.
  auto nms_batch = [&](int nwant, int bw0, u32 sbase) {
    wave_lds_sync();
    int nent = nwant;
    {  // never split a lane's pair of entries over two batches
      const u32 idl = nq[(u32)(qhead + nent - 1) & (u32)(F8_NQ - 1)];
      if (__builtin_amdgcn_readfirstlane((int)idl) & 0x80) nent -= 1;
    }
    const bool live = lane < nent;  // the other lanes compute on stale ids and store nothing
    const u32 id = nq[(u32)(qhead + lane) & (u32)(F8_NQ - 1)];  // bits 0..5 lane, bit 6 half, bit 7 "the next entry is my lane's other half", bits 8.. row
    const u32 sl = live ? (id & 63u) : 1u, half = (id >> 6) & 1u;
    const int row = live ? (int)(id >> 8) : bw0;
    const bool eB = HALF && sl >= 32u;  // HALF form: the entry belongs to half-wave B (its strip, its frame)
    const int col0 = HALF ? (eB ? sB : strip) * F8_HSTRIP_W - F8_HALO + 8 * (int)(sl & 31u) + 4 * (int)half
                          : strip * F8_STRIP_W - F8_HALO + 8 * (int)sl + 4 * (int)half;  // column of the half's pixel 0
    // blur rows row-2 .. row+2, columns col0-4 .. col0+7 (three aligned dwords; -2 .. +5 are used)
    const u32 rel = (u32)(row - (bw0 - 2));  // 0..5: the window's output rows are bw0-2 .. bw0+3
    const u32 lo = sl * 8u + half * 4u - 4u;
    u32 d[5][3], s[5][3];  // per blur row: d = b[+1]-b[-1], s = b[-1]+2b[0]+b[+1] of the pixel pairs (-1,0), (1,2), (3,4)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      u32 slot = sbase + rel + (u32)r;
      slot = min(slot, slot - (u32)F8_RING);  // wrap (unsigned): sbase + rel + r < 2 * F8_RING
      const u32 *q = reinterpret_cast<const u32 *>(bring + slot * (u32)F8_ROW_BYTES + lo);
      const u32 D0 = q[0], D1 = q[1], D2 = q[2];
      const u32 Pm = unpack_hi(D0), A = unpack_lo(D1), B = unpack_hi(D1), Cq = unpack_lo(D2);  // (b-2,b-1) (b0,b1) (b2,b3) (b4,b5)
      const u32 c0p = pair_shift(A, Pm), c1p = pair_shift(B, A), c2p = pair_shift(Cq, B);       // (b-1,b0) (b1,b2) (b3,b4)
      d[r][0] = R(I(A) - I(Pm)); d[r][1] = R(I(B) - I(A)); d[r][2] = R(I(Cq) - I(B));
      s[r][0] = pk_mad2(c0p, Pm + A); s[r][1] = pk_mad2(c1p, A + B); s[r][2] = pk_mad2(c2p, B + Cq);
    }
    // zero padding of the Sobel / gradient stages (cannyEdgeD.cu:142-149, 222-229): the sums of columns and rows outside
    // the image are zero.  Only half-lanes at the frame's border see any -- one batch in many has such an entry at all,
    // so the masks are built and applied in a variant of their own (wave-uniform choice)
    u32 SU[6], SC[6], SN[6], Xc[3], Yc[3];
    auto sums = [&](auto masked) {
      constexpr bool MASKED = decltype(masked)::value;
      u32 pm[3] = { ~0u, ~0u, ~0u }, mU = ~0u, mN = ~0u;
      if constexpr (MASKED) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const u32 vlo = (u32)(col0 + 2 * j - 1) < (u32)W ? 0x0000FFFFu : 0u, vhi = (u32)(col0 + 2 * j) < (u32)W ? 0xFFFF0000u : 0u;
          pm[j] = vlo | vhi;
        }
        mU = row > 0 ? 0xFFFFFFFFu : 0u;
        mN = row + 1 < H ? 0xFFFFFFFFu : 0u;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        u32 XU = pk_mad2(d[1][j], R(I(d[0][j]) + I(d[2][j]))), YU = R(I(s[0][j]) - I(s[2][j]));
        u32 XC = pk_mad2(d[2][j], R(I(d[1][j]) + I(d[3][j]))), YC = R(I(s[1][j]) - I(s[3][j]));
        u32 XN = pk_mad2(d[3][j], R(I(d[2][j]) + I(d[4][j]))), YN = R(I(s[2][j]) - I(s[4][j]));
        if constexpr (MASKED) {
          XU &= pm[j] & mU; YU &= pm[j] & mU;
          XC &= pm[j]; YC &= pm[j];
          XN &= pm[j] & mN; YN &= pm[j] & mN;
        }
        SU[2 * j] = (u32)mad16<0, 0>(XU, XU, mul16<0, 0>(YU, YU)); SU[2 * j + 1] = (u32)mad16<1, 1>(XU, XU, mul16<1, 1>(YU, YU));
        SC[2 * j] = (u32)mad16<0, 0>(XC, XC, mul16<0, 0>(YC, YC)); SC[2 * j + 1] = (u32)mad16<1, 1>(XC, XC, mul16<1, 1>(YC, YC));
        SN[2 * j] = (u32)mad16<0, 0>(XN, XN, mul16<0, 0>(YN, YN)); SN[2 * j + 1] = (u32)mad16<1, 1>(XN, XN, mul16<1, 1>(YN, YN));
        Xc[j] = XC; Yc[j] = YC;
      }
    };
    // (an entry's pixels -1 .. 4 are columns col0-1 .. col0+4; its rows row-1 .. row+1)
    const bool at_border = live && (col0 == 0 || col0 + 4 >= W || row == 0 || row + 1 >= H);
    if (__ballot(at_border) != 0) sums(std::true_type{});
    else sums(std::false_type{});
    // S*[0] / [5]: the neighbouring pixels -1 / 4; S*[1 + q]: the half's own pixel q
    const u32 gmax = max(max(SC[1], SC[2]), max(SC[3], SC[4]));
    const bool wraps = __ballot(live && gmax >= wrap_limit) != 0;
    u32 nibS = 0, nibC = 0;
    // pixel q lives in pair (q + 1) / 2, half (q + 1) % 2 of the pairs (-1,0), (1,2), (3,4)
    auto px = [&](auto qc, u32 A2, u32 Um, u32 Vp) {
      constexpr int q = decltype(qc)::value, e = (q + 1) % 2;
      const u32 g = SC[1 + q];
      bool cand = g >= a_lo0, strong = g >= a_hi0;
      if (wraps) {  // u8 wrap of gradients >= 256 (cannyEdgeD.cu:267): the bands of S2 whose low byte passes the thresholds
        const bool w0 = g >= 262144u, w1 = g >= 1048576u;
        cand = (cand && !w0) || (g >= p.a_lo[1] && !w1) || g >= p.a_lo[2];
        strong = (strong && !w0) || (g >= p.a_hi[1] && !w1) || g >= p.a_hi[2];
      }
      // direction bins (cannyEdgeD.cu:239-264) without atan2: E1 = 2x(x-y) - S2 > 0, E2 = 2x(x+y) - S2 > 0
      const bool p1 = mul16<e, e>(A2, Um) > (int)g, p2 = mul16<e, e>(A2, Vp) > (int)g;
      // neighbours (:245-264): bin0 down/up, bin1 down-left/up-right, bin2 right/left, bin3 up-left/down-right
      const u32 m0 = max(SN[1 + q], SU[1 + q]), m1 = max(SN[q], SU[2 + q]);
      const u32 m2 = max(SC[2 + q], SC[q]), m3 = max(SU[q], SN[2 + q]);
      const u32 mb = p1 ? (p2 ? m2 : m3) : (p2 ? m1 : m0);
      const bool keep = mb <= g;  // non-strict on both sides, as the reference
      nibS |= (strong && keep) ? (1u << q) : 0u;
      nibC |= (cand && keep) ? (1u << q) : 0u;
    };
    u32 A2[3], Um[3], Vp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { A2[j] = R(U(Xc[j]) + U(Xc[j])); Um[j] = R(I(Xc[j]) - I(Yc[j])); Vp[j] = R(U(Xc[j]) + U(Yc[j])); }
    px(std::integral_constant<int, 0>{}, A2[0], Um[0], Vp[0]);
    px(std::integral_constant<int, 1>{}, A2[1], Um[1], Vp[1]);
    px(std::integral_constant<int, 2>{}, A2[1], Um[1], Vp[1]);
    px(std::integral_constant<int, 3>{}, A2[2], Um[2], Vp[2]);
    const u32 nib = nibS | (nibC << 8);  // out-of-image pixels have S2 = 0: never candidates (a_lo >= 4)
    const u32 nxt = from_lane_above(nib);
    const bool first = (id & 0x80u) != 0;
    const u32 prev_id = from_lane_below(id);
    const bool second = lane > 0 && (prev_id & 0x80u) != 0;  // a half that follows its lane's half 0 is stored by that entry
    const u32 w = first ? (nib | (nxt << 4)) : (nib << (4 * half));
    if (live && !second) {
      const u32 o = HALF ? (u32)row * plane_pitch + (u32)((col0 - 4 * (int)half) >> 3) + (eB ? (u32)(IN == 2 ? 3 * dframe : dframe) * (u32)H * plane_pitch : 0u)
                         : (u32)row * plane_pitch + (u32)(strip * 62) + sl - 1u;
      splane[o] = (uint8_t)w;
      cplane[o] = (uint8_t)(w >> 8);
    }
    if (PROV && live)
      *reinterpret_cast<u32 *>(prov_frame + (u32)row * p.prov_pitch + (u32)col0 + ((HALF && eB) ? (u32)(IN == 2 ? 3 * dframe : dframe) * (u32)p.prov_fs : 0u)) = nibble_to_bytes(nib & 0xFu);
    qhead = (qhead + nent) & (F8_NQ - 1);
    qcount -= nent;
  };


  // ---- dense windows: wave-wide non-maximum suppression in registers ---------------------------------------------------
  // The queue + batch scheme above wins while few half-lanes pass the low threshold (6-7 % on camera-like frames: 27 + 41
  // instructions per row for the test and the queue, ~230 per batch of 64).  On dense content -- noise: every half-lane
  // passes, two batches per row -- it costs 2.6 x a natural frame (round 2: 132 k frames/s against 225 k with round 1's
  // wave-wide k_nms).  A window that follows one with more than p.dense_enter queued half-lanes is therefore processed
  // whole, by every lane for its own 8 pixels: the 10 blur rows of the ring -> d / s -> the Sobel sums and the exact S2
  // of 8 rows (two more than it outputs: no state is carried from the window before) -> thresholds, direction bins and
  // the non-strict NMS of the 6 output rows against the rows above / below in registers and the neighbours' columns by
  // DPP -> one byte of each plane and 8 bytes of provisional map per lane and row, stored once (no zero-stores, no queue,
  // no LDS beyond the 10 row reads).  ~260 instructions per row whatever the content.  The arithmetic is the batch's.
  auto dense_window = [&](int bw0, u32 sbase) {
    // pixels outside the image have zero gradients (cannyEdgeD.cu:142-149, 222-229): half-word masks of the lane's aligned pairs
    const u32 pm[4] = { __builtin_amdgcn_perm(0u, cmask[0], 0x01010000u), __builtin_amdgcn_perm(0u, cmask[0], 0x03030202u),
                        __builtin_amdgcn_perm(0u, cmask[1], 0x01010000u), __builtin_amdgcn_perm(0u, cmask[1], 0x03030202u) };
    u32 dP[2][4], sP[2][4];              // d / s of the two previous blur rows
    u32 SU[10], SC[10], SN[10];          // S2 of three Sobel rows: [0] = pixel -1 (the lane below), [1 + q] = pixel q, [9] = pixel 8
    u32 Xc[4], Yc[4], Xn[4], Yn[4];      // the Sobel sums of the centre row / the newest row (packed pairs)
#pragma unroll
    for (int k = 0; k < 10; ++k) SU[k] = SC[k] = SN[k] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) dP[0][j] = dP[1][j] = sP[0][j] = sP[1][j] = Xc[j] = Yc[j] = Xn[j] = Yn[j] = 0;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {  // blur row bw0 - 4 + i
      u32 slot = sbase + (u32)i;
      slot = slot >= (u32)F8_RING ? slot - (u32)F8_RING : slot;
      const u32x2 b = *reinterpret_cast<const u32x2 *>(bring + slot * (u32)F8_ROW_BYTES + lane * 8);
      const u32 A0 = unpack_lo(b.x), B0 = unpack_hi(b.x), A1 = unpack_lo(b.y), B1 = unpack_hi(b.y);
      const u32 Bl = from_lane_below(B1), Ar = from_lane_above(A0);
      const u32 m1 = pair_shift(A0, Bl), p1 = pair_shift(B0, A0), p3 = pair_shift(A1, B0), p5 = pair_shift(B1, A1), p7 = pair_shift(Ar, B1);
      const u32 Cc[4] = { A0, B0, A1, B1 };
      const u32 Lf[4] = { m1, p1, p3, p5 }, Rt[4] = { p1, p3, p5, p7 };
      u32 dk[4], sk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dk[j] = R(I(Rt[j]) - I(Lf[j]));
        sk[j] = pk_mad2(Cc[j], Lf[j] + Rt[j]);
      }
      if (i >= 2) {  // Sobel row bw0 - 5 + i from blur rows i - 2, i - 1, i
        int c = bw0 - 5 + i;
        asm volatile("" : "+s"(c));
        const u32 rm = (u32)c < (u32)H ? 0xFFFFFFFFu : 0u;  // (wave-uniform) rows outside the image: zero gradients
#pragma unroll
        for (int k = 0; k < 10; ++k) { SU[k] = SC[k]; SC[k] = SN[k]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Xc[j] = Xn[j]; Yc[j] = Yn[j];
          const u32 msk = pm[j] & rm;
          const u32 X = pk_mad2(dP[(i - 1) & 1][j], R(I(dP[i & 1][j]) + I(dk[j]))) & msk;  // cannyEdgeD.cu:158-162
          const u32 Y = R(I(sP[i & 1][j]) - I(sk[j])) & msk;                                // :163-167
          Xn[j] = X; Yn[j] = Y;
          SN[1 + 2 * j] = (u32)mad16<0, 0>(X, X, mul16<0, 0>(Y, Y));
          SN[2 + 2 * j] = (u32)mad16<1, 1>(X, X, mul16<1, 1>(Y, Y));
        }
        SN[0] = from_lane_below(SN[8]);
        SN[9] = from_lane_above(SN[1]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { dP[i & 1][j] = dk[j]; sP[i & 1][j] = sk[j]; }
      if (i == 7 || i == 8) {  // what the next window's first steps expect: d / s of blur rows bw0 + 3 and bw0 + 4
#pragma unroll
        for (int j = 0; j < 4; ++j) { dr[i - 7][j] = dk[j]; sr[i - 7][j] = sk[j]; }
      }
      if (i >= 4) {  // NMS of output row bw0 - 6 + i: centre SC, above SU, below SN
        int c = bw0 - 6 + i;
        asm volatile("" : "+s"(c));  // (keeps the six rows' addresses from being computed -- and held in SGPRs -- ahead of the loop)
        const bool valid = (u32)(c - r0) < (u32)(rend - r0);  // wave-uniform
        const u32 g0 = max(max(SC[1], SC[2]), max(SC[3], SC[4])), g1 = max(max(SC[5], SC[6]), max(SC[7], SC[8]));
        // the same measure as the queue path's: half-lanes with a pixel above the low threshold
        const u64 mh0 = __ballot(g0 >= a_lo0) & (valid ? lanes0 : 0ull), mh1 = __ballot(g1 >= a_lo0) & (valid ? lanes1 : 0ull);
        cnt += (c >= 2 && c < H - 2) ? __popcll(mh0) + __popcll(mh1) : 0;
        if (valid) {
          const bool wraps = __ballot(max(g0, g1) >= wrap_limit) != 0;
          u32 bitsS = 0, bitsC = 0;
          u32 A2[4], Um[4], Vp[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { A2[j] = R(U(Xc[j]) + U(Xc[j])); Um[j] = R(I(Xc[j]) - I(Yc[j])); Vp[j] = R(U(Xc[j]) + U(Yc[j])); }
          auto px = [&](auto qc) {
            constexpr int q = decltype(qc)::value, j = q / 2, e = q % 2;
            const u32 g = SC[1 + q];
            bool cand = g >= a_lo0, strong = g >= a_hi0;
            if (wraps) {  // u8 wrap of gradients >= 256 (cannyEdgeD.cu:267)
              const bool w0 = g >= 262144u, w1 = g >= 1048576u;
              cand = (cand && !w0) || (g >= p.a_lo[1] && !w1) || g >= p.a_lo[2];
              strong = (strong && !w0) || (g >= p.a_hi[1] && !w1) || g >= p.a_hi[2];
            }
            // direction bins (cannyEdgeD.cu:239-264): E1 = 2x(x-y) - S2 > 0, E2 = 2x(x+y) - S2 > 0
            const bool p1 = mul16<e, e>(A2[j], Um[j]) > (int)g, p2 = mul16<e, e>(A2[j], Vp[j]) > (int)g;
            const u32 n0 = max(SN[1 + q], SU[1 + q]), n1 = max(SN[q], SU[2 + q]);
            const u32 n2 = max(SC[2 + q], SC[q]), n3 = max(SU[q], SN[2 + q]);
            const u32 mb = p1 ? (p2 ? n2 : n3) : (p2 ? n1 : n0);
            const bool keep = mb <= g;  // non-strict on both sides, as the reference
            bitsS |= (strong && keep) ? (1u << q) : 0u;
            bitsC |= (cand && keep) ? (1u << q) : 0u;
          };
          px(std::integral_constant<int, 0>{}); px(std::integral_constant<int, 1>{}); px(std::integral_constant<int, 2>{}); px(std::integral_constant<int, 3>{});
          px(std::integral_constant<int, 4>{}); px(std::integral_constant<int, 5>{}); px(std::integral_constant<int, 6>{}); px(std::integral_constant<int, 7>{});
          if (st_lane) {  // (out-of-image pixels have S2 = 0: never candidates, a_lo >= 4)
            splane[(u32)c * plane_pitch + st_off] = (uint8_t)bitsS;
            cplane[(u32)c * plane_pitch + st_off] = (uint8_t)bitsC;
            if constexpr (PROV)
              *reinterpret_cast<g_u32x2 *>(prov_frame + (u32)c * p.prov_pitch + prov_voff) = u32x2{ nibble_to_bytes(bitsS & 0xFu), nibble_to_bytes(bitsS >> 4) };
          }
        }
      }
    }
    wq = cnt;
  };

  // ---- the run -------------------------------------------------------------------------------------------------------
  // input rows r0-4 .. rend+3 -> blur rows r0-2 .. rend+1 -> Sobel / output rows r0 .. rend-1
  int islot = 0;  // ring slot of the next input row
  {  // warm-up: input rows r0-4 .. r0-1 only feed the accumulators (and the fix-up ring)
    Raw xw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xw[j] = load_raw(r0 - 4 + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 x[2];
      use_raw(r0 - 4 + j, xw[j], x);
      blur_row(std::true_type{}, r0 - 4 + j, islot, 0, x);
      islot = islot + 1 == F8_RING ? 0 : islot + 1;
    }
  }
  Raw xn[G];
#pragma unroll
  for (int j = 0; j < G; ++j) xn[j] = load_raw(r0 + j);
  const int nwin = (rend + 2 - (r0 - 2) + F8_SUB - 1) / F8_SUB;
  int bslot0 = 0;  // blur-ring slot of the window's first blur row (blur row r0 - 2 sits in slot 0)
  // one window; DENSE: its phase 2 is the dense path.  (A lambda instantiated twice, for two separate loops below.)
  auto window = [&](auto dense_c, int w) {
    constexpr bool DENSE = decltype(dense_c)::value;
    const int bw0 = r0 - 2 + w * F8_SUB;  // the window's blur rows are bw0 .. bw0+5 (its last input row is bw0 + 7)
    qn = 0;
#pragma unroll
    for (int j = 0; j < F8_SUB; ++j) {
      u32 x[2];
      use_raw(bw0 + 2 + j, xn[j % G], x);
      xn[j % G] = load_raw(bw0 + 2 + j + G);
      const int bs = bslot0 + j;
      blur_row(std::false_type{}, bw0 + 2 + j, islot, bs >= F8_RING ? bs - F8_RING : bs, x);
      islot = islot + 1 == F8_RING ? 0 : islot + 1;
    }
    // fix-up: the queued pixels get the literal chain, written over their byte of the blur ring.  Queue overflow (flat
    // areas: every pixel of a constant region has S = 159 v): every pixel of the window is recomputed.
    wave_lds_sync();
    // the ring now holds input rows bw0-2 .. bw0+7 and `islot` is where bw0+8 will go, i.e. where the oldest, bw0-2, sits:
    // the chain of blur row bw0 + e starts at input row bw0 + e - 2, ring slot (islot + e) mod F8_RING
    if (F8_ABL & 2) qn = 0;
    if (qn <= F8_FQ) {
#pragma nounroll
      for (int base = 0; base < qn; base += 64) {
        const int e = base + lane;
        const u32 ent = e < qn ? fq[e] : 0u;
        u32 fl = ent & 0x03030303u;
        const u32 el = (ent >> 2) & 63u, es = (ent >> 10) & 15u;  // lane, blur-ring slot
        u32 rel = es + (u32)F8_RING - (u32)bslot0;                // row of the window, 0..5
        rel = min(rel, rel - (u32)F8_RING);
        u32 s0 = (u32)islot + rel;
        s0 = min(s0, s0 - (u32)F8_RING);
        while (fl) {
          const u32 b = (u32)__builtin_ctz(fl);
          fl &= fl - 1;
          const u32 k = (b >> 3) + 4u * (b & 1u);  // pixel 0..7 of the lane
          bring[es * (u32)F8_ROW_BYTES + el * 8u + k] = (unsigned char)f8_chain(ring, s0, el * 8u + k);
        }
      }
    } else {
#pragma nounroll
      for (int e = lane; e < F8_SUB * F8_ROW_BYTES; e += 64) {
        const u32 rel = (u32)e / (u32)F8_ROW_BYTES;
        const int row = bw0 + (int)rel;
        const u32 cb = (u32)e % (u32)F8_ROW_BYTES;
        const int col = HALF ? (cb >= 256u ? sB : strip) * F8_HSTRIP_W - F8_HALO + (int)(cb & 255u) : strip * F8_STRIP_W - F8_HALO + (int)cb;
        u32 es = (u32)bslot0 + rel;
        es = min(es, es - (u32)F8_RING);
        u32 s0 = (u32)islot + rel;
        s0 = min(s0, s0 - (u32)F8_RING);
        if (cb >= 2u && cb < (u32)F8_ROW_BYTES - 2u && row >= 0 && row < H && col >= 0 && col < W)
          bring[es * (u32)F8_ROW_BYTES + cb] = (unsigned char)f8_chain(ring, s0, cb);
      }
    }
    wave_lds_sync();
    // phase 2: blur rows bw0-1 .. bw0+4 (the first one is the previous window's last) -> output rows bw0-2 .. bw0+3,
    // whose NMS needs blur rows up to bw0+5: all present
    u32x2 bq[F8_SUB];
#pragma unroll
    for (int j = 0; j < F8_SUB; ++j) {
      int bs = bslot0 + j - 1;
      bs = bs < 0 ? bs + F8_RING : bs >= F8_RING ? bs - F8_RING : bs;
      bq[j] = *reinterpret_cast<const u32x2 *>(bring + bs * F8_ROW_BYTES + lane * 8);
    }
    if (p.dbg_blur && own_lane && c0 < W) {  // diagnostics (HC_OPT_DEBUG_TAPS): the fixed-up blur rows of this run
#pragma unroll
      for (int j = 0; j < F8_SUB; ++j) {
        const int k = bw0 - 1 + j;
        if (k >= r0 && k < rend) *reinterpret_cast<g_u32x2 *>(p.dbg_blur + (size_t)(frame + (hB ? (IN == 2 ? 3 * dframe : dframe) : 0)) * p.dbg_fs + (size_t)k * p.dbg_pitch + (u32)c0) = bq[j];
      }
    }
    int sb = bslot0 - 4;  // blur-ring slot of blur row bw0 - 4
    if (sb < 0) sb += F8_RING;
    if constexpr (DENSE) {
      dense_window(bw0, (u32)sb);
    } else {
      wq = 0;
      if (!(F8_ABL & 4)) {
        step(std::integral_constant<int, 0>{}, bw0 - 1, bq[0].x, bq[0].y);
        step(std::integral_constant<int, 1>{}, bw0 + 0, bq[1].x, bq[1].y);
        step(std::integral_constant<int, 2>{}, bw0 + 1, bq[2].x, bq[2].y);
        if (F8_ABL & 1) { qhead = (qhead + qcount) & (F8_NQ - 1); qcount = 0; }
        while (qcount >= 64) nms_batch(64, bw0, (u32)sb);
        step(std::integral_constant<int, 3>{}, bw0 + 2, bq[3].x, bq[3].y);
        step(std::integral_constant<int, 4>{}, bw0 + 3, bq[4].x, bq[4].y);
        step(std::integral_constant<int, 5>{}, bw0 + 4, bq[5].x, bq[5].y);
        if (F8_ABL & 1) { qhead = (qhead + qcount) & (F8_NQ - 1); qcount = 0; }
        while (qcount > 0) nms_batch(min(qcount, 64), bw0, (u32)sb);
      }
    }
    wave_lds_sync();  // the next window's phase 1 overwrites the oldest ring rows
    if (IN == 2) __syncthreads();  // the three channels of this run stay within a window of each other (see above)
    bslot0 = bslot0 + F8_SUB >= F8_RING ? bslot0 + F8_SUB - F8_RING : bslot0 + F8_SUB;
  };
  // Two loops, not one loop with a branch in it: the dense path needs more SGPRs than the kernel has (its compare masks),
  // and with the branch inside the window loop the spills it caused were paid by every window of every frame (+2 % on
  // frames that never take the path).  A window that queued more than p.dense_enter half-lanes hands over to the dense
  // loop, a dense window that counted fewer than p.dense_leave hands back.
  int w = 0;
  bool dense = p.dense_enter < 0;  // HC_OPT_FRONT_DENSE = 1: every window (tests)
  while (w < nwin) {
#pragma nounroll
    for (; w < nwin && !dense; ++w) {
      window(std::false_type{}, w);
      dense = wq > p.dense_enter;
    }
#pragma nounroll
    for (; w < nwin && dense; ++w) {
      window(std::true_type{}, w);
      dense = wq > p.dense_leave;
    }
  }
}


// =====================================================================================================================
// k_front8o -- Mode O (cv::Canny(src, low, high, 3, L2gradient) semantics, OpenCV 4.x modules/imgproc/src/canny.cpp) on
// the skeleton of k_front8: the same strips, runs, windows, id-only NMS queue and dense batches; no blur, so no phase 1
// and no fix-up -- the source rows themselves go into the wave's LDS ring.
//   Sobel 3x3 with BORDER_REPLICATE: rows are replicated by clamping the row of the load; columns outside the image are
//   stored as zero in the ring, the half-lanes that hold column 0 or W-1 are always queued, and the batch re-derives
//   their gradients from the replicated bytes (a v_perm per dword, in the border variant of the batch only).
//   magnitude m = |dx| + |dy| (or dx^2 + dy^2), zero outside the image; m <= low dropped; direction by the integer
//   tangent test (TG22 = 13573, shift 15); asymmetric non-maximum suppression (m > first neighbour, m >= second on the
//   axes; strict on both diagonal neighbours); m > high seeds.  (k_front_o in canny_kernels.hip is the 4-px form of the
//   same arithmetic and still serves 3-channel sources.)
// Phase 2 queues the half-lanes with a pixel above low: exactly for the L1 magnitude (packed |dx| + |dy|, five
// instructions per pixel pair; the cheaper pair-summed dx^2 + dy^2 >= ceil((low + 1)^2 / 2) queued three times as many
// half-lanes as pass, and the batches cost more than the test saves); with L2gradient by the necessary condition
// dx^2 + dy^2 summed over the pair (two v_dot2) >= low + 1.
constexpr int F8O_WAVE_BYTES = F8_RING * F8_ROW_BYTES + (F8_NQ + 4) * 4;  // 7,184 B per wave

template <bool L2, bool PROV>
__global__ __launch_bounds__(256) void k_front8o(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  unsigned char *bring = smem + wib * F8O_WAVE_BYTES;  // source rows (columns outside the image zero)
  lds_u32 *nq = (lds_u32 *)(bring + F8_RING * F8_ROW_BYTES);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_count; i += gridDim.x * blockDim.x) p.zero_words[i] = 0u;  // (as k_front8)
  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items) return;
  const int chunk = item % p.nchunks;
  const int strip = (item / p.nchunks) % p.nstrips;
  const int frame = item / (p.nchunks * p.nstrips);
  const int W = p.W, H = p.H;
  const int r0 = chunk * p.run_rows;  // output rows [r0, rend)
  const int rend = min(r0 + p.run_rows, H);
  const int c0 = strip * F8_STRIP_W - F8_HALO + lane * 8;
  u32 cmask[2] = { 0, 0 };
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    cmask[k >> 2] |= in ? (0xFFu << (8 * (k & 3))) : 0u;
  }
  const bool own_lane = lane >= 1 && lane <= 62;
  const bool col_any = (cmask[0] | cmask[1]) != 0;
  const uint8_t *frame_base = p.in + (size_t)frame * p.in_frame_stride;
  const u32 in_pitch32 = (u32)p.in_pitch;  // launch_front8o checks H * pitch < 2^32 and pitch >= round_up(W, 8)
  const u32 ld_safe = col_any ? (u32)c0 : 0u;  // lanes without an image column read the row's first bytes (masked)
  const int rlast = min(H - 1, rend + 1);  // last source row this run needs
  // unconditional loads; BORDER_REPLICATE down the columns = the row index clamped into the image
  auto load_raw = [&](int row) -> u32x2 {
    u32 lo = ld_safe;
    asm volatile("" : "+v"(lo));
    const uint8_t *q = frame_base + (u32)min(max(row, 0), rlast) * in_pitch32 + lo;
    return *reinterpret_cast<const g_u32x2 *>(q);
  };

  u32 dr[2][4], sr[2][4];  // d = x[+1]-x[-1] and s = x[-1]+2x[0]+x[+1] of the two previous rows, [ring][pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) dr[a][b] = sr[a][b] = 0;
  const size_t plane_off = (size_t)frame * H * p.RD * 4;
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const u32 plane_pitch = (u32)p.RD * 4u;
  const bool st_lane = own_lane && col_any;
  const u32 st_off = st_lane ? (u32)(strip * 62 + lane - 1) : (u32)(strip * 62);
  const u32 prov_voff = st_lane ? (u32)c0 : (u32)(strip * F8_STRIP_W);
  const u32 low = p.a_lo[0], high = p.a_hi[0];  // plain thresholds on m (squared by the host for L2gradient)
  // phase 2's test.  L2gradient: pair-summed dx^2 + dy^2 >= low + 1 (necessary).  L1: the exact magnitudes, packed; `nec`
  // is the per-half bias that carries "m > low" into bit 15 (thresholds are capped at 32767 by hc_set_thresholds)
  const u32 nec = L2 ? low + 1u : (0x7FFFu - min(low, 0x7FFFu)) * 0x10001u;
  const u64 lanes0 = uniform64(__ballot(own_lane && cmask[0] != 0)), lanes1 = uniform64(__ballot(own_lane && cmask[1] != 0));
  // half-lanes that hold the image's first or last column: their gradients depend on the replicated border, which only
  // the batch applies -- always queued
  const u64 bord0 = uniform64(__ballot(own_lane && (c0 == 0 || (c0 <= W - 1 && W - 1 <= c0 + 3))));
  const u64 bord1 = uniform64(__ballot(own_lane && (c0 + 4 <= W - 1 && W - 1 <= c0 + 7)));
  uint8_t *prov_frame = PROV ? p.prov_out + (size_t)frame * p.prov_fs : nullptr;
  const u32 k_tg22 = 13573u, k_m32768 = 0x8000u;  // 16-bit multiplier operands (low halves): TG22 and -2^15

  int qhead = 0, qcount = 0;  // NMS queue (circular, F8_NQ ids): wave-uniform
  // source row k (its 8 bytes per lane in b0, b1) arrives -> gradient row k-1 -> the row's possible candidates are queued
  auto step = [&](auto uc, int k, u32 b0, u32 b1) {
    constexpr int u = decltype(uc)::value;
    constexpr int rn = u % 2, rp = (u + 1) % 2;
    const u32 A0 = unpack_lo(b0), B0 = unpack_hi(b0), A1 = unpack_lo(b1), B1 = unpack_hi(b1);
    const u32 Bl = from_lane_below(B1), Ar = from_lane_above(A0);
    const u32 m1 = pair_shift(A0, Bl), p1 = pair_shift(B0, A0), p3 = pair_shift(A1, B0), p5 = pair_shift(B1, A1), p7 = pair_shift(Ar, B1);
    const u32 Cc[4] = { A0, B0, A1, B1 };
    const u32 Lf[4] = { m1, p1, p3, p5 }, Rt[4] = { p1, p3, p5, p7 };
    u32 T2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 dk = R(I(Rt[j]) - I(Lf[j]));
      const u32 sk = pk_mad2(Cc[j], Lf[j] + Rt[j]);
      const u32 X = pk_mad2(dr[rp][j], R(I(dr[rn][j]) + I(dk)));  // dx = right - left, smoothed 1-2-1 down the rows
      const u32 Y = R(I(sk) - I(sr[rn][j]));                      // dy = bottom - top
      if constexpr (L2) T2[j] = (u32)__builtin_amdgcn_sdot2(I(Y), I(Y), sdot2_0(X, X), false);  // pair sum: a necessary condition
      else T2[j] = R(__builtin_elementwise_max(I(X), -I(X))) + R(__builtin_elementwise_max(I(Y), -I(Y)));  // |dx| + |dy| of both pixels, exact (<= 2040 per half)
      dr[rn][j] = dk;
      sr[rn][j] = sk;
    }
    const int c = k - 1;  // the gradient / output row
    const bool valid = (u32)(c - r0) < (u32)(rend - r0);
    bool q0, q1;  // the half-lane may hold (L2) / holds (L1) a pixel with m > low
    if constexpr (L2) {
      q0 = __builtin_elementwise_max(T2[0], T2[1]) >= nec; q1 = __builtin_elementwise_max(T2[2], T2[3]) >= nec;
    } else {
      // any of the four 16-bit magnitudes > low  <=>  bit 15 of (m + 0x7FFF - low) in one of the halves (low <= 32767)
      const u32 g0 = R(__builtin_elementwise_max(U(T2[0]), U(T2[1]))), g1 = R(__builtin_elementwise_max(U(T2[2]), U(T2[3])));
      q0 = ((g0 + nec) & 0x80008000u) != 0; q1 = ((g1 + nec) & 0x80008000u) != 0;
    }
    const u64 mh0 = (__ballot(q0) | bord0) & (valid ? lanes0 : 0ull), mh1 = (__ballot(q1) | bord1) & (valid ? lanes1 : 0ull);
    {
      typedef __attribute__((address_space(1))) uint8_t gmem_u8;
      typedef __attribute__((address_space(1))) g_u32x2 gmem_u32x2;
      gmem_u8 *sp = (gmem_u8 *)uniform_sel(valid, splane + (u32)c * plane_pitch, p.dump), *cp = (gmem_u8 *)uniform_sel(valid, cplane + (u32)c * plane_pitch, p.dump_c);
      u32 so = st_off;
      asm volatile("" : "+v"(so));
      sp[so] = 0;
      cp[so] = 0;
      if constexpr (PROV) {
        gmem_u8 *pp = (gmem_u8 *)uniform_sel(valid, prov_frame + (u32)c * p.prov_pitch, p.dump_p);
        u32 o = prov_voff;
        asm volatile("" : "+v"(o));
        if (!(F8_ABL & 8)) *reinterpret_cast<gmem_u32x2 *>(pp + o) = u32x2{ 0u, 0u };
      }
    }
    const u32 below = mbcnt64(mh0) + mbcnt64(mh1);
    const u32 idbase = ((u32)c << 8) | (u32)lane;
    const u32 pos0 = ((u32)(qhead + qcount) + below) & (u32)(F8_NQ - 1);
    const u32 pos1 = (pos0 + lane_sel(mh0, 1u, 0u)) & (u32)(F8_NQ - 1);
    nq[lane_sel(mh0, pos0, (u32)F8_NQ)] = idbase | lane_sel(mh1, 0x80u, 0u);
    nq[lane_sel(mh1, pos1, (u32)F8_NQ)] = idbase | 0x40u;
    u32 n0, n1;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n0) : "s"(mh0) : "scc");
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n1) : "s"(mh1) : "scc");
    qcount += (int)(n0 + n1);
  };

  // One dense NMS pass: up to 64 queued half-lanes, an entry per lane.  sbase: ring slot of source row bw0 - 4.
  auto nms_batch = [&](int nwant, int bw0, u32 sbase) {
    wave_lds_sync();
    int nent = nwant;
    {  // never split a lane's pair of entries over two batches
      const u32 idl = nq[(u32)(qhead + nent - 1) & (u32)(F8_NQ - 1)];
      if (__builtin_amdgcn_readfirstlane((int)idl) & 0x80) nent -= 1;
    }
    const bool live = lane < nent;
    const u32 id = nq[(u32)(qhead + lane) & (u32)(F8_NQ - 1)];
    const u32 sl = live ? (id & 63u) : 1u, half = (id >> 6) & 1u;
    const int row = live ? (int)(id >> 8) : bw0;
    const int col0 = strip * F8_STRIP_W - F8_HALO + 8 * (int)sl + 4 * (int)half;  // column of the half's pixel 0
    const u32 rel = (u32)(row - (bw0 - 2));
    const u32 lo = sl * 8u + half * 4u - 4u;
    // pixels -2 .. 5 of the entry lie outside the image's columns somewhere, or its rows -1 / +1 outside the image
    const bool at_border = live && (col0 == 0 || col0 + 5 >= W || row == 0 || row + 1 >= H);
    u32 XU[3], YU[3], XC[3], YC[3], XN[3], YN[3];
    auto sobel = [&](auto masked) {
      constexpr bool MASKED = decltype(masked)::value;
      u32 selA = 0, selB = 0, pm[3] = { ~0u, ~0u, ~0u }, mU = ~0u, mN = ~0u;
      if constexpr (MASKED) {
        // BORDER_REPLICATE along the row: byte t of the 8-byte window (columns col0-2 .. col0+5) comes from the column
        // clamped into the image, which lies in the same window
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const u32 idx = (u32)(min(max(col0 - 2 + t, 0), W - 1) - (col0 - 2)) & 7u;
          if (t < 4) selA |= idx << (8 * t);
          else selB |= idx << (8 * (t - 4));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // magnitudes outside the image are zero
          const u32 vlo = (u32)(col0 + 2 * j - 1) < (u32)W ? 0x0000FFFFu : 0u, vhi = (u32)(col0 + 2 * j) < (u32)W ? 0xFFFF0000u : 0u;
          pm[j] = vlo | vhi;
        }
        mU = row > 0 ? 0xFFFFFFFFu : 0u;
        mN = row + 1 < H ? 0xFFFFFFFFu : 0u;
      }
      u32 d[5][3], s[5][3];
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        u32 slot = sbase + rel + (u32)r;
        slot = min(slot, slot - (u32)F8_RING);
        const u32 *q = reinterpret_cast<const u32 *>(bring + slot * (u32)F8_ROW_BYTES + lo);
        const u32 D0 = q[0], D1 = q[1], D2 = q[2];
        u32 Pm, A, B, Cq;  // (b-2,b-1) (b0,b1) (b2,b3) (b4,b5)
        if constexpr (MASKED) {
          const u32 E0 = __builtin_amdgcn_alignbyte(D1, D0, 2), E1 = __builtin_amdgcn_alignbyte(D2, D1, 2);  // bytes -2..1, 2..5
          const u32 F0 = __builtin_amdgcn_perm(E1, E0, selA), F1 = __builtin_amdgcn_perm(E1, E0, selB);
          Pm = unpack_lo(F0); A = unpack_hi(F0); B = unpack_lo(F1); Cq = unpack_hi(F1);
        } else {
          Pm = unpack_hi(D0); A = unpack_lo(D1); B = unpack_hi(D1); Cq = unpack_lo(D2);
        }
        const u32 c0p = pair_shift(A, Pm), c1p = pair_shift(B, A), c2p = pair_shift(Cq, B);  // (b-1,b0) (b1,b2) (b3,b4)
        d[r][0] = R(I(A) - I(Pm)); d[r][1] = R(I(B) - I(A)); d[r][2] = R(I(Cq) - I(B));
        s[r][0] = pk_mad2(c0p, Pm + A); s[r][1] = pk_mad2(c1p, A + B); s[r][2] = pk_mad2(c2p, B + Cq);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        XU[j] = pk_mad2(d[1][j], R(I(d[0][j]) + I(d[2][j]))); YU[j] = R(I(s[2][j]) - I(s[0][j]));
        XC[j] = pk_mad2(d[2][j], R(I(d[1][j]) + I(d[3][j]))); YC[j] = R(I(s[3][j]) - I(s[1][j]));
        XN[j] = pk_mad2(d[3][j], R(I(d[2][j]) + I(d[4][j]))); YN[j] = R(I(s[4][j]) - I(s[2][j]));
        if constexpr (MASKED) {
          XU[j] &= pm[j] & mU; YU[j] &= pm[j] & mU;
          XC[j] &= pm[j]; YC[j] &= pm[j];
          XN[j] &= pm[j] & mN; YN[j] &= pm[j] & mN;
        }
      }
    };
    if (__ballot(at_border) != 0) sobel(std::true_type{});
    else sobel(std::false_type{});
    // magnitudes of pixels -1 .. 4 in the rows above / of / below the entry: index t + 1
    u32 MU[6], MC[6], MN[6], aXc[3], aYc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      aXc[j] = R(__builtin_elementwise_max(I(XC[j]), -I(XC[j])));
      aYc[j] = R(__builtin_elementwise_max(I(YC[j]), -I(YC[j])));
      if constexpr (L2) {
        MU[2 * j] = (u32)mad16<0, 0>(XU[j], XU[j], mul16<0, 0>(YU[j], YU[j])); MU[2 * j + 1] = (u32)mad16<1, 1>(XU[j], XU[j], mul16<1, 1>(YU[j], YU[j]));
        MC[2 * j] = (u32)mad16<0, 0>(XC[j], XC[j], mul16<0, 0>(YC[j], YC[j])); MC[2 * j + 1] = (u32)mad16<1, 1>(XC[j], XC[j], mul16<1, 1>(YC[j], YC[j]));
        MN[2 * j] = (u32)mad16<0, 0>(XN[j], XN[j], mul16<0, 0>(YN[j], YN[j])); MN[2 * j + 1] = (u32)mad16<1, 1>(XN[j], XN[j], mul16<1, 1>(YN[j], YN[j]));
      } else {  // |dx| + |dy| <= 2040 per half: one add on both halves
        const u32 mu = R(__builtin_elementwise_max(I(XU[j]), -I(XU[j]))) + R(__builtin_elementwise_max(I(YU[j]), -I(YU[j])));
        const u32 mc = aXc[j] + aYc[j];
        const u32 mn = R(__builtin_elementwise_max(I(XN[j]), -I(XN[j]))) + R(__builtin_elementwise_max(I(YN[j]), -I(YN[j])));
        MU[2 * j] = mu & 0xFFFFu; MU[2 * j + 1] = mu >> 16;
        MC[2 * j] = mc & 0xFFFFu; MC[2 * j + 1] = mc >> 16;
        MN[2 * j] = mn & 0xFFFFu; MN[2 * j + 1] = mn >> 16;
      }
    }
    u32 nibS = 0, nibC = 0;
    // pixel q lives in pair (q + 1) / 2, half (q + 1) % 2 of the pairs (-1,0), (1,2), (3,4)
    auto px = [&](auto qc) {
      constexpr int q = decltype(qc)::value, j = (q + 1) / 2, e = (q + 1) % 2;
      const u32 m = MC[1 + q];
      // tangent test on x = |dx|, y = |dy| (canny.cpp): horizontal if y*2^15 < x*TG22, vertical if y*2^15 > x*(TG22 + 2^16)
      const int E = mad16<e, 0>(aYc[j], k_m32768, mul16<e, 0>(aXc[j], k_tg22));  // x*TG22 - y*2^15
      const int x16 = (int)(e ? (aXc[j] & 0xFFFF0000u) : (aXc[j] << 16));       // x * 2^16
      const bool hz = E > 0, vt = E + x16 < 0;
      const bool dneg = mul16<e, e>(XC[j], YC[j]) < 0;  // sign(dx) != sign(dy): the diagonal that runs up-right / down-left
      // first neighbour (strict), second neighbour (>= on the axes, strict on the diagonals)
      const u32 n1 = hz ? MC[q] : vt ? MU[1 + q] : dneg ? MU[2 + q] : MU[q];
      const u32 n2 = hz ? MC[2 + q] : vt ? MN[1 + q] : dneg ? MN[q] : MN[2 + q];
      const bool keep = m > n1 && m + ((hz || vt) ? 1u : 0u) > n2;
      nibS |= (m > high && keep) ? (1u << q) : 0u;
      nibC |= (m > low && keep) ? (1u << q) : 0u;
    };
    px(std::integral_constant<int, 0>{});
    px(std::integral_constant<int, 1>{});
    px(std::integral_constant<int, 2>{});
    px(std::integral_constant<int, 3>{});
    const u32 nib = nibS | (nibC << 8);  // pixels outside the image have m = 0: never above low
    const u32 nxt = from_lane_above(nib);
    const bool first = (id & 0x80u) != 0;
    const u32 prev_id = from_lane_below(id);
    const bool second = lane > 0 && (prev_id & 0x80u) != 0;
    const u32 w = first ? (nib | (nxt << 4)) : (nib << (4 * half));
    if (live && !second) {
      const u32 o = (u32)row * plane_pitch + (u32)(strip * 62) + sl - 1u;
      splane[o] = (uint8_t)w;
      cplane[o] = (uint8_t)(w >> 8);
    }
    if (PROV && live)
      *reinterpret_cast<u32 *>(prov_frame + (u32)row * p.prov_pitch + (u32)col0) = nibble_to_bytes(nib & 0xFu);
    qhead = (qhead + nent) & (F8_NQ - 1);
    qcount -= nent;
  };

  // ---- the run: source rows r0-2 .. rend+1 -> gradient / output rows r0 .. rend-1 ---------------------------------------
  u32x2 xn[F8_SUB];
#pragma unroll
  for (int j = 0; j < F8_SUB; ++j) xn[j] = load_raw(r0 - 2 + j);
  const int nwin = (rend + 2 - (r0 - 2) + F8_SUB - 1) / F8_SUB;
  int bslot0 = 0;  // ring slot of the window's first source row (row r0 - 2 sits in slot 0)
#pragma nounroll
  for (int w = 0; w < nwin; ++w) {
    const int bw0 = r0 - 2 + w * F8_SUB;  // the window brings source rows bw0 .. bw0+5 and produces output rows bw0-2 .. bw0+3
#pragma unroll
    for (int j = 0; j < F8_SUB; ++j) {
      const u32x2 x = u32x2{ xn[j].x & cmask[0], xn[j].y & cmask[1] };
      xn[j] = load_raw(bw0 + j + F8_SUB);
      const int bs = bslot0 + j;
      *reinterpret_cast<u32x2 *>(bring + (bs >= F8_RING ? bs - F8_RING : bs) * F8_ROW_BYTES + lane * 8) = x;
    }
    wave_lds_sync();
    int sb = bslot0 - 4;  // ring slot of source row bw0 - 4
    if (sb < 0) sb += F8_RING;
    u32x2 bq[F8_SUB];  // rows bw0-1 .. bw0+4 (the first one is the previous window's last)
#pragma unroll
    for (int j = 0; j < F8_SUB; ++j) {
      int bs = bslot0 + j - 1;
      bs = bs < 0 ? bs + F8_RING : bs >= F8_RING ? bs - F8_RING : bs;
      bq[j] = *reinterpret_cast<const u32x2 *>(bring + bs * F8_ROW_BYTES + lane * 8);
    }
    if (!(F8_ABL & 4)) {
      step(std::integral_constant<int, 0>{}, bw0 - 1, bq[0].x, bq[0].y);
      step(std::integral_constant<int, 1>{}, bw0 + 0, bq[1].x, bq[1].y);
      step(std::integral_constant<int, 2>{}, bw0 + 1, bq[2].x, bq[2].y);
      if (F8_ABL & 1) { qhead = (qhead + qcount) & (F8_NQ - 1); qcount = 0; }
      while (qcount >= 64) nms_batch(64, bw0, (u32)sb);
      step(std::integral_constant<int, 3>{}, bw0 + 2, bq[3].x, bq[3].y);
      step(std::integral_constant<int, 4>{}, bw0 + 3, bq[4].x, bq[4].y);
      step(std::integral_constant<int, 5>{}, bw0 + 4, bq[5].x, bq[5].y);
      if (F8_ABL & 1) { qhead = (qhead + qcount) & (F8_NQ - 1); qcount = 0; }
      while (qcount > 0) nms_batch(min(qcount, 64), bw0, (u32)sb);
    }
    wave_lds_sync();  // the next window overwrites the oldest ring rows
    bslot0 = bslot0 + F8_SUB >= F8_RING ? bslot0 + F8_SUB - F8_RING : bslot0 + F8_SUB;
  }
}

template <int IN>
static hipError_t launch_front8_t(const FrontParams &p, hipStream_t s)
{
  constexpr int WPB = IN == 2 ? 3 : F8_WPB;
  if (IN == 2 && p.total_items % 3 != 0) return hipErrorInvalidValue;
  if constexpr (IN != 2) {
    if (p.one_wave) {  // (only beside a hysteresis, i.e. with the provisional map)
      if (!p.prov_out) return hipErrorInvalidValue;
      const dim3 grid1((unsigned)p.total_items), block1(64);
      if (p.half) hipLaunchKernelGGL((k_front8<IN, true, true, true>), grid1, block1, (size_t)F8_WAVE_BYTES, s, p);
      else hipLaunchKernelGGL((k_front8<IN, true, false, true>), grid1, block1, (size_t)F8_WAVE_BYTES, s, p);
      return hipGetLastError();
    }
  }
  const dim3 grid((unsigned)((p.total_items + WPB - 1) / WPB)), block(64 * WPB);
  const size_t lds = (size_t)WPB * F8_WAVE_BYTES;
  if (p.half) {
    if (p.prov_out) hipLaunchKernelGGL((k_front8<IN, true, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_front8<IN, false, true>), grid, block, lds, s, p);
  } else {
    if (p.prov_out) hipLaunchKernelGGL((k_front8<IN, true, false>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_front8<IN, false, false>), grid, block, lds, s, p);
  }
  return hipGetLastError();
}

hipError_t launch_front8(const FrontParams &p, hipStream_t s)
{
  const int windows = (p.run_rows + 4) / F8_SUB;
  if (windows < 1 || p.run_rows != front8_run_rows(windows) || p.nchunks * p.run_rows < p.H) return hipErrorInvalidValue;
  const size_t w8 = ((size_t)p.W + 7) / 8 * 8;
  if ((unsigned long long)p.H * p.in_pitch >= (1ull << 32) || p.in_pitch < (p.bgr ? 3 : 1) * w8) return hipErrorInvalidValue;
  if (p.prov_out && (p.W % 8 != 0)) return hipErrorInvalidValue;
  if (p.dbg_blur && p.dbg_pitch < w8) return hipErrorInvalidValue;
  if (!p.dump || !p.dump_c || !p.dump_p || !p.zeros) return hipErrorInvalidValue;
  const int in_frames = p.bgr == 2 ? p.nframes / 3 : p.nframes, per = p.bgr == 2 ? 3 : 1;
  if (p.half) {
    // two half-strips per wave; half-wave B's frame is reached by 32-bit lane offsets (the caller sized dump / zeros for them)
    const long units = (long)in_frames * p.nhalf;
    if (p.nhalf != front8_half_strips(p.W) || (long)p.total_items != (units + 1) / 2 * p.nchunks * per) return hipErrorInvalidValue;
    if ((unsigned long long)p.in_frame_stride + (unsigned long long)p.H * p.in_pitch >= (1ull << 32)) return hipErrorInvalidValue;
    if (p.prov_out && (unsigned long long)per * p.prov_fs + (unsigned long long)p.H * p.prov_pitch >= (1ull << 32)) return hipErrorInvalidValue;
  } else if (p.nstrips != front8_strips(p.W) || (long)p.total_items != (long)p.nframes * p.nstrips * p.nchunks) return hipErrorInvalidValue;
  return p.bgr == 2 ? launch_front8_t<2>(p, s) : p.bgr == 1 ? launch_front8_t<1>(p, s) : launch_front8_t<0>(p, s);
}

// Mode O, one-channel source: same strips and run lengths as launch_front8 (p.bgr must be 0; p.l2gradient selects the magnitude)
hipError_t launch_front8o(const FrontParams &p, hipStream_t s)
{
  const int windows = (p.run_rows + 4) / F8_SUB;
  if (windows < 1 || p.run_rows != front8_run_rows(windows) || p.nchunks * p.run_rows < p.H || p.nstrips != front8_strips(p.W) || p.bgr) return hipErrorInvalidValue;
  const size_t w8 = ((size_t)p.W + 7) / 8 * 8;
  if ((unsigned long long)p.H * p.in_pitch >= (1ull << 32) || p.in_pitch < w8) return hipErrorInvalidValue;
  if (p.prov_out && (p.W % 8 != 0)) return hipErrorInvalidValue;
  if (!p.dump) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((p.total_items + 3) / 4)), block(256);
  const size_t lds = (size_t)4 * F8O_WAVE_BYTES;
  if (p.l2gradient) {
    if (p.prov_out) hipLaunchKernelGGL((k_front8o<true, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_front8o<true, false>), grid, block, lds, s, p);
  } else {
    if (p.prov_out) hipLaunchKernelGGL((k_front8o<false, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_front8o<false, false>), grid, block, lds, s, p);
  }
  return hipGetLastError();
}

}  // namespace hc
