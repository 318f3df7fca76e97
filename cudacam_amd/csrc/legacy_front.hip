// legacy_front.hip -- the round-1 front kernels of Mode R: the 4-px fused k_front and the pair k_blur + k_nms (a u8 blur
// plane in HBM between them).  NOT part of the product library since round 3: k_front8 (front8.hip) serves every width and
// every row layout.  They are compiled into cudacam_amd/libhipcanny_legacy.so (python -m cudacam_amd.build, -DHC_LEGACY_FRONT)
// together with the product's sources, where HC_OPT_FRONT_SPLIT 1 / 0 selects them: two independent implementations of
// the same arithmetic that the parity tests run every case through (tests/test_gpu_taps.py, tests/test_gpu_parity.py,
// tests/fuzz_parity.py).
#include "canny_device.h"
#include <algorithm>

namespace hc {

// =================================================================================================
// k_front
// =================================================================================================
// Work item = (frame, strip, run of RUN = FSUB*m - 4 output rows), one per wave, 4 independent waves
// per workgroup (no workgroup barrier anywhere).  A wave marches down its run in sub-chunks of FSUB
// blur rows; every intermediate stays in registers (horizontal neighbours come from the adjacent lane
// by DPP) except the blur rows of the current sub-chunk, which pass through a wave-private LDS slab:
//   phase 1  FSUB input rows -> FSUB blur rows (u8) into the slab
//   fix-up   the few pixels whose exact float result cannot be decided by integers (see below)
//   phase 2  FSUB blur rows -> Sobel -> S = sumX^2+sumY^2 -> direction -> NMS -> thresholds -> bit planes
// The vertical accumulators of phase 1 and the row rings of phase 2 are carried across sub-chunks, so
// the only redundant work per run is the 4-row blur warm-up and the 4 extra blur rows (RUN+8 input
// rows and RUN+4 blur rows per RUN output rows), while LDS stays at 6.5 KiB per wave.
constexpr int FSUB = 24;    // blur rows per sub-chunk: multiple of the prefetch group (4) and of the ring period (6)
constexpr int QCAP = 128;   // fix-up queue entries per wave and sub-chunk (one lane-dword each; expected fill ~37)
constexpr int FRONT_WAVE_BYTES = FSUB * 256 + QCAP * 4;

size_t front_lds_bytes() { return (size_t)4 * FRONT_WAVE_BYTES; }  // 26,624 B: 6 workgroups per CU
int front_run_rows(int subchunks) { return FSUB * subchunks - 4; }

#ifndef HC_FRONT_WAVES
#define HC_FRONT_WAVES 4
#endif
template <int IN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HC_FRONT_WAVES, 8))) void k_front(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  unsigned char *blur_s = smem + wib * FRONT_WAVE_BYTES;
  u32 *queue = reinterpret_cast<u32 *>(blur_s + FSUB * 256);

  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items) return;
  const int chunk = item % p.nchunks;
  const int strip = (item / p.nchunks) % p.nstrips;
  // per-channel mode: an output "frame" is one channel of an input frame (3 edge maps per input frame)
  const int frame = item / (p.nchunks * p.nstrips);           // output frame = bit-plane index
  const int in_frame = IN == 2 ? frame / 3 : frame, ch = IN == 2 ? frame % 3 : 0;
  const int W = p.W, H = p.H;
  const int r0 = chunk * p.run_rows;              // output rows [r0, rend)
  const int rend = min(r0 + p.run_rows, H);
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;

  // per-lane column validity: byte mask for packed u8 rows, dword masks for the 4 S values
  u32 cmask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    cmask |= in ? (0xFFu << (8 * k)) : 0u;
  }
  const u32 hmask = cmask & 0x80808080u;  // "undecidable" flag positions of the pixels inside the image
  // packed-i16 masks (0xFFFF per in-image pixel) for the Sobel pairs, kept as plain VGPR values
  const u32 pm0 = __builtin_amdgcn_perm(0u, cmask, 0x01010000u), pm1 = __builtin_amdgcn_perm(0u, cmask, 0x03030202u);
  // nibble of pixel slots this lane may set in the bit planes (lanes 1..62, columns inside the image)
  const u32 oknib1 = (lane >= 1 && lane <= 62) ? ((cmask & 1u) | ((cmask >> 7) & 2u) | ((cmask >> 14) & 4u) | ((cmask >> 21) & 8u)) : 0u;
  const u32 oknib = oknib1 | (oknib1 << 8);
  const bool col_any = cmask != 0;
  const uint8_t *frame_base = p.in + (size_t)in_frame * p.in_frame_stride;
  const u32 plane_pitch = (u32)p.RD * 4u, in_pitch32 = (u32)p.in_pitch;  // launch_front checks H * pitch < 2^32
  const u32 ld_off = (u32)((IN ? 3 : 1) * c0);          // used only where col_any (then c0 >= 0): uniform row base + 32-bit lane offset
  // per-channel mode: byte selectors that pull channel ch of 4 pixels out of 12 interleaved bytes
  const u32 selA = ch == 0 ? 0x0c060300u : ch == 1 ? 0x0c070401u : 0x0c0c0502u;  // from {d1,d0}: bytes ch, ch+3, (ch+6 if < 8)
  const u32 selB = ch == 0 ? 0x05020100u : ch == 1 ? 0x06020100u : 0x07040100u;  // from {d2,t}: t.b0, t.b1, (t.b2 | d2 byte), d2 byte

  auto load_row = [&](int row) -> u32 {
    u32 v = 0;
    if (row >= 0 && row < H && col_any) {
      const uint8_t *rowp = frame_base + (u32)row * in_pitch32;     // wave-uniform: scalar row base + 32-bit lane offset
      u32 lo = ld_off;
      asm volatile("" : "+v"(lo));                                  // keeps the lane offset out of a hoisted 64-bit VGPR pointer
      if (IN == 2) {
        const u32 *q = reinterpret_cast<const u32 *>(rowp + lo);
        const u32 t = __builtin_amdgcn_perm(q[1], q[0], selA);
        v = __builtin_amdgcn_perm(q[2], t, selB);
      } else if (IN == 1) {
        // 4 interleaved BGR pixels = 12 bytes = 3 dwords; stage 0 (cannyEdgeD.cu:53-69) fused into the
        // load: each pixel's 3 bytes are aligned into one dword and reduced by one v_dot4 with the
        // weights (7, 38, 19, 0); sum of weights = 64, so the reference's min(255, .) never triggers
        const u32 *q = reinterpret_cast<const u32 *>(rowp + lo);
        const u32 d0 = q[0], d1 = q[1], d2 = q[2];
        const u32 wts = 0x00132607u;
        const u32 m0 = __builtin_amdgcn_udot4(d0, wts, 0u, false) >> 6;
        const u32 m1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), wts, 0u, false) >> 6;
        const u32 m2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), wts, 0u, false) >> 6;
        const u32 m3 = __builtin_amdgcn_udot4(d2 >> 8, wts, 0u, false) >> 6;
        v = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
      } else v = *reinterpret_cast<const u32 *>(rowp + lo);
    }
    return v;
  };

  // ------------------------------------------------------------------ phase 1: blur rows -> LDS
  // Packed u16 arithmetic, two pixels per VALU op.  With K the 5x5 integer kernel (sum 159) and
  // S = sum K*x (<= 40545 < 2^16):  rows of K are [2 4 5 4 2], [4 9 12 9 4], [5 12 15 12 5], so per
  // input row p = x[-2]+x[+2], q = x[-1]+x[+1], c = x[0] give
  //   h0 = 2p+4q+5c, h1 = 4p+9q+12c = 2*h0 + (q+2c), h2 = 5p+12q+15c = h0 + h1 - (p + q + 2c)
  // and S(row i) = h0[i-2] + h1[i-1] + h2[i] + h1[i+1] + h0[i+2] (running accumulators a1..a4).
  // The reference's float chain differs from S/159 by < 4.2e-4 (25 roundings of partial sums < 256
  // plus coefficient error) << 1/159, so trunc(chain) == floor(S/159) unless S % 159 == 0; those
  // pixels (0.6 % of random data) are queued and recomputed with the literal fmaf chain.
  u32 a1[2] = { 0, 0 }, a2[2] = { 0, 0 }, a3[2] = { 0, 0 }, a4[2] = { 0, 0 };
  int qn = 0;  // queue fill of the current sub-chunk (wave-uniform, lives in an SGPR; may run past QCAP: overflow)

  // one input row into the vertical accumulators; returns the two packed pairs of S for blur row (row - 2)
  auto accumulate = [&](u32 xraw, u32 Sp[2]) {
    const u32 x = xraw & cmask;
    const u32 A = unpack_lo(x), B = unpack_hi(x);
    const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
    const u32 m1 = pair_shift(A, Bl);  // (x-1, x0)
    const u32 p1 = pair_shift(B, A);   // (x1, x2)
    const u32 p3 = pair_shift(Ar, B);  // (x3, x4)
    // NB: every packed u16 sum below stays < 2^16 per half (S <= 40545), so plain 32-bit adds and
    // subtractions act on both halves at once without carry/borrow between them -- and v_add_u32 /
    // v_sub_u32 issue at twice the rate of the v_pk_* forms on gfx950 (tools/experiments/valu_rate2.hip).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32 P = h == 0 ? Bl + B : A + Ar;
      const u32 Q = h == 0 ? m1 + p1 : p1 + p3;
      const u32 Cc = h == 0 ? A : B;
      const u32 e = pk_mad2(Q, P);             // 2q + p
      const u32 h0 = pk_mad2(e, pk_mul5(Cc));  // 2p + 4q + 5c
      const u32 w = pk_mad2(Cc, Q);            // q + 2c
      const u32 h1 = pk_mad2(h0, w);           // 4p + 9q + 12c
      const u32 h2 = (h0 + h1) - (P + w);      // 5p + 12q + 15c
      Sp[h] = a4[h] + h0;
      a4[h] = a3[h] + h1;
      a3[h] = a2[h] + h2;
      a2[h] = a1[h] + h1;
      a1[h] = h0;
    }
  };

  // input row jr -> blur row rb = jr - 2 into slab slot `slot`
  auto phase1_row = [&](int rb, int slot, u32 xraw) {
    u32 Sp[2];
    accumulate(xraw, Sp);
    u32 bl = 0;
    if (rb >= 0 && rb < H) {  // wave-uniform
      // n = floor(S/159) = (S*52759) >> 23, exact for S <= 40545 (tests/test_oracle_exhaustive.py).
      // S % 159 == 0  <=>  bits 15..22 of the product are all zero (the fraction is 73*n/2^23 < 2^-8 then
      // and >= 52759/2^23 > 2^-8 otherwise): after >> 15 the low byte is that "fraction byte", the next is n.
      const u16x2 mlo = { 52759, 0 }, mhi = { 0, 52759 };
      const u32 t0 = __builtin_amdgcn_udot2(U(Sp[0]), mlo, 0u, false) >> 15;
      const u32 t1 = __builtin_amdgcn_udot2(U(Sp[0]), mhi, 0u, false) >> 15;
      const u32 t2 = __builtin_amdgcn_udot2(U(Sp[1]), mlo, 0u, false) >> 15;
      const u32 t3 = __builtin_amdgcn_udot2(U(Sp[1]), mhi, 0u, false) >> 15;
      const u32 nf01 = __builtin_amdgcn_perm(t1, t0, 0x04000501u);  // (n0, n1, f0, f1)
      const u32 nf23 = __builtin_amdgcn_perm(t3, t2, 0x04000501u);
      bl = __builtin_amdgcn_perm(nf23, nf01, 0x05040100u) & cmask;
      const u32 fz = __builtin_amdgcn_perm(nf23, nf01, 0x07060302u);
      // zero-byte detector: bit 7 of every byte that is 0 (a byte equal to 1 above a zero byte may be
      // flagged too: harmless, the exact chain is then evaluated for a pixel that did not need it)
      const u32 hz = (fz - 0x01010101u) & ~fz & hmask;
      const u64 any = __ballot(hz != 0);
      if (any != 0) {  // most rows have a pixel or two: one queue entry per flagged lane
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(any >> 32), __builtin_amdgcn_mbcnt_lo((u32)any, (u32)qn));
        if (hz != 0 && rank < (u32)QCAP) queue[rank] = hz | (u32)lane | ((u32)slot << 8);
        qn += __popcll(any);
      }
    }
    reinterpret_cast<u32 *>(blur_s)[slot * 64 + lane] = bl;
  };

  // ------------------------------------------------------------------ phase 2 state: blur -> bit planes
  // Sobel is separable: per blur row d = b[+1]-b[-1], s = b[-1]+2b[0]+b[+1] (packed i16 pairs), then
  // sumX(i) = d[i-1]+2d[i]+d[i+1], sumY(i) = s[i-1]-s[i+1] (cannyEdgeD.cu:158-167).
  // S = sumX^2+sumY^2 by one v_dot2 per pixel; comparisons of the reference's float gradient are
  // comparisons of S (strictly monotone, tests).  Direction bins (cannyEdgeD.cu:239-264) exactly:
  // with D = sumX^2-sumY^2 and Q = sumX*sumY (two more dot products on the packed pair):
  //   E1 = D-2Q, E2 = D+2Q;  both > 0: bin 2 (horizontal), both <= 0: bin 0 (vertical),
  //   E1 > 0 >= E2: bin 3, E2 > 0 >= E1: bin 1   (|2Q| < |D| decides axis vs diagonal; no atan2).
  // All rings below are indexed by compile-time constants (the row loop is unrolled by 6 = lcm(2,3)).
  u32 dr[2][2], sr[2][2];  // d and s of the two previous blur rows, [ring][pair]
  u32 Sr[3][6];            // S rows: [ring][0]=left neighbour, [1..4]=own 4 px, [5]=right neighbour
  u32 Xr[2][2], Yr[2][2];  // packed sumX / sumY pairs of the two newest Sobel rows, [ring][pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) dr[a][b] = sr[a][b] = Xr[a][b] = Yr[a][b] = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) Sr[a][b] = 0;

  // this strip's 31 bytes of each bit-plane row: lane pair (2b+1, 2b+2) -> byte b
  const size_t plane_off = (size_t)frame * H * p.RD * 4;   // wave-uniform
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const bool store_lane = (lane & 1) && lane < 63;
  const u32 st_off = (u32)(strip * 31 + (lane >> 1));      // odd lanes only: (lane - 1) / 2
  const u32 a_lo0 = p.a_lo[0], a_hi0 = p.a_hi[0], wrap_limit = p.wrap_limit;

  // ------------------------------------------------------------------ the run
#ifndef HC_FRONT_G
#define HC_FRONT_G 2
#endif
  constexpr int G = HC_FRONT_G;  // rows per prefetch group (FSUB is a multiple of it)
  u32 xn[G];
  {  // warm-up: input rows r0-4 .. r0-1 only feed the accumulators (first blur row of the run is r0-2)
    u32 xw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xw[j] = load_row(r0 - 4 + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) { u32 Sp[2]; accumulate(xw[j], Sp); }
  }

#pragma nounroll
  for (int sub = 0; sub < p.subchunks; ++sub) {
    const int b0 = r0 - 2 + sub * FSUB;  // blur row of slab slot 0 (input row b0 + 2)
    if (b0 > rend + 1) break;            // NMS row c needs blur rows up to c + 2 <= rend + 1
    qn = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) xn[j] = load_row(b0 + 2 + j);
#pragma nounroll
    for (int g = 0; g < FSUB / G; ++g) {
      u32 xc[G];
#pragma unroll
      for (int j = 0; j < G; ++j) xc[j] = xn[j];
      if (g + 1 < FSUB / G)  // next group in flight while this one is processed (nothing is held across phase 2)
#pragma unroll
        for (int j = 0; j < G; ++j) xn[j] = load_row(b0 + 2 + (g + 1) * G + j);
#pragma unroll
      for (int j = 0; j < G; ++j) phase1_row(b0 + g * G + j, g * G + j, xc[j]);
    }
    // fix-up: the queued pixels get the literal chain.  If the queue overflowed (large flat regions:
    // every pixel of a constant area has S = 159*v), every pixel of the slab is recomputed instead.
    wave_lds_sync();
    if (qn <= QCAP) {
#pragma nounroll
      for (int base = 0; base < qn; base += 64) {
        const int e = base + lane;
        const u32 ent = e < qn ? queue[e] : 0u;
        u32 fl = ent & 0x80808080u;
        const u32 el = ent & 63u, es = (ent >> 8) & 31u;
        while (fl) {
          const u32 k = (u32)__builtin_ctz(fl) >> 3;
          fl &= fl - 1;
          const int row = b0 + (int)es;
          const int col = strip * STRIP_W - STRIP_HALO + (int)(el * 4u + k);
          blur_s[es * 256u + el * 4u + k] = (unsigned char)gauss_chain_px<IN>(frame_base, p.in_pitch, W, H, row, col, ch);
        }
      }
    } else {
#pragma nounroll
      for (int e = lane; e < FSUB * 256; e += 64) {  // e = slot*256 + lane'*4 + k
        const int row = b0 + (e >> 8);
        const int col = strip * STRIP_W - STRIP_HALO + (e & 255);
        if (row >= 0 && row < H && col >= 0 && col < W)
          blur_s[e] = (unsigned char)gauss_chain_px<IN>(frame_base, p.in_pitch, W, H, row, col, ch);
      }
    }
    wave_lds_sync();

    // the S-row halos of the two carried rows are re-fetched here so that they are dead during phase 1
    Sr[1][0] = from_lane_below(Sr[1][4]); Sr[1][5] = from_lane_above(Sr[1][1]);
    Sr[2][0] = from_lane_below(Sr[2][4]); Sr[2][5] = from_lane_above(Sr[2][1]);
#pragma nounroll
    for (int t0 = 0; t0 < FSUB; t0 += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int t = t0 + u;
        const int k = b0 + t;  // blur row arriving
        const int rn = u % 2, rp = (u + 1) % 2;            // d/s ring: new row -> [rn] (holds row k-2), previous row k-1 in [rp]
        const int sN = u % 3, sC = (u + 2) % 3, sU = (u + 1) % 3;  // S ring: new / centre / up
        const u32 b = reinterpret_cast<const u32 *>(blur_s)[t * 64 + lane];
        if (p.dbg_blur && k >= r0 && k < rend && lane >= 1 && lane <= 62 && c0 < W)  // diagnostics: the fixed-up blur row (pitch >= round_up(W, 4))
          *reinterpret_cast<u32 *>(p.dbg_blur + (size_t)frame * p.dbg_fs + (size_t)k * p.dbg_pitch + (u32)c0) = b;
        const u32 A = unpack_lo(b), B = unpack_hi(b);
        const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
        const u32 m1 = pair_shift(A, Bl), p1 = pair_shift(B, A), p3 = pair_shift(Ar, B);
        u32 dk[2], sk[2];
        dk[0] = R(I(p1) - I(m1));      // signed halves: packed op
        sk[0] = pk_mad2(A, m1 + p1);   // non-negative halves < 2^16: plain add
        dk[1] = R(I(p3) - I(p1));
        sk[1] = pk_mad2(B, p1 + p3);
        // Sobel row i = k-1 from blur rows k-2 (ring rn), k-1 (ring rp), k (new); rows outside the image
        // give 0 (zero padding of every stage): the column masks are cleared for them
        const int i = k - 1;
        const u32 rowm = (i >= 0 && i < H) ? 0xFFFFFFFFu : 0u;  // wave-uniform
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32 pm = (h == 0 ? pm0 : pm1) & rowm;
          const u32 X = pk_mad2(dr[rp][h], R(I(dr[rn][h]) + I(dk[h]))) & pm;  // two's complement: the u16 mad is exact for i16
          const u32 Y = R(I(sr[rn][h]) - I(sk[h])) & pm;
          Xr[rn][h] = X;
          Yr[rn][h] = Y;
          // S = sumX^2 + sumY^2 (< 2^22): two 16x16 multiply-adds per pixel on the packed halves
          Sr[sN][1 + 2 * h] = (u32)mad16<0, 0>(X, X, mul16<0, 0>(Y, Y));
          Sr[sN][2 + 2 * h] = (u32)mad16<1, 1>(X, X, mul16<1, 1>(Y, Y));
        }
        Sr[sN][0] = from_lane_below(Sr[sN][4]);
        Sr[sN][5] = from_lane_above(Sr[sN][1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) { dr[rn][h] = dk[h]; sr[rn][h] = sk[h]; }

        // NMS + thresholds for row c = k-2: centre ring sC (its sumX/sumY are in ring rp), up sU, down sN
        const int c = k - 2;
        if (c >= r0 && c < rend) {  // wave-uniform
          u32 nib = 0;
          // candidate masks of the 4 pixel slots (wave-wide, in SGPR pairs); rows without a single candidate skip the rest
          u64 cl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) cl[q] = __ballot(Sr[sC][1 + q] >= a_lo0);
          if ((cl[0] | cl[1] | cl[2] | cl[3]) != 0) {
            u64 st[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) st[q] = __ballot(Sr[sC][1 + q] >= a_hi0);
            // a gradient >= 256 (S >= 2^18 >= the strong threshold) makes the u8 wrap bands of cannyEdgeD.cu:267 apply (rare)
            if ((st[0] | st[1] | st[2] | st[3]) != 0 && __ballot(max(max(Sr[sC][1], Sr[sC][2]), max(Sr[sC][3], Sr[sC][4])) >= wrap_limit) != 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const u32 g = Sr[sC][1 + q];
                const u64 w0 = __ballot(g >= 262144u), w1 = __ballot(g >= 1048576u);
                cl[q] = (cl[q] & ~w0) | (__ballot(g >= p.a_lo[1]) & ~w1) | __ballot(g >= p.a_lo[2]);
                st[q] = (st[q] & ~w0) | (__ballot(g >= p.a_hi[1]) & ~w1) | __ballot(g >= p.a_hi[2]);
              }
            }
            u32 nibS = 0, nibC = 0;
            // direction bins (cannyEdgeD.cu:239-264) without atan2: with x = sumX, y = sumY
            //   E1 = x^2 - 2xy - y^2 = 2x(x - y) - S,  E2 = x^2 + 2xy - y^2 = 2x(x + y) - S;
            //   E1 > 0 && E2 > 0: bin 2, neither: bin 0, only E1: bin 3, only E2: bin 1.
            // 2x, x - y, x + y are formed once per packed pair, the two products are 16x16 multiplies.
            auto slot = [&](auto hc, auto ec, u32 A2, u32 Um, u32 Vp) {
              constexpr int h = decltype(hc)::value, e = decltype(ec)::value, q = 2 * h + e;
              u64 mS = 0, mC = 0;
              if (cl[q] != 0) {  // some lane has a candidate in this pixel slot
                const u32 g = Sr[sC][1 + q];
                const u64 p1m = __ballot(mul16<e, e>(A2, Um) > (int)g), p2m = __ballot(mul16<e, e>(A2, Vp) > (int)g);
                // neighbours (cannyEdgeD.cu:245-264): bin0 down/up, bin1 down-left/up-right, bin2 right/left, bin3 up-left/down-right.
                // The four "both neighbours <= g" masks are combined with the direction masks by scalar logic
                // (SALU issues beside the VALU; the kernel is VALU-issue-bound).
                const u64 k0 = __ballot(max(Sr[sN][1 + q], Sr[sU][1 + q]) <= g);
                const u64 k1 = __ballot(max(Sr[sN][q], Sr[sU][2 + q]) <= g);
                const u64 k2 = __ballot(max(Sr[sC][2 + q], Sr[sC][q]) <= g);
                const u64 k3 = __ballot(max(Sr[sU][q], Sr[sN][2 + q]) <= g);
                const u64 keep = (~p1m & ~p2m & k0) | (~p1m & p2m & k1) | (p1m & p2m & k2) | (p1m & ~p2m & k3);
                mS = st[q] & keep;
                mC = cl[q] & keep;
              }
              // per-lane nibbles (bit q = pixel slot q): one carry-in add per mask
              nibS = shift_in(nibS, mS);
              nibC = shift_in(nibC, mC);
            };
            auto pair = [&](auto hc) {
              constexpr int h = decltype(hc)::value;
              const u32 X = Xr[rp][h], Y = Yr[rp][h];
              const u32 A2 = R(U(X) + U(X));   // packed 2x (|x| <= 1020)
              const u32 Um = R(I(X) - I(Y));
              const u32 Vp = R(U(X) + U(Y));   // wrap-around add = signed add
              slot(hc, std::integral_constant<int, 1>{}, A2, Um, Vp);
              slot(hc, std::integral_constant<int, 0>{}, A2, Um, Vp);
            };
            pair(std::integral_constant<int, 1>{});  // slots 3, 2
            pair(std::integral_constant<int, 0>{});  // slots 1, 0
            nib = (nibS | (nibC << 8)) & oknib;  // strong in bits 0..3, candidate in bits 8..11
          }
          const u32 w = nib | (from_lane_above(nib) << 4);  // bits 0..7 strong byte, 8..15 candidate byte
          if (store_lane) {
            // wave-uniform row base (a plane is < 4 GiB) + 32-bit lane offset.  The multiply is pinned to the
            // SALU: as plain C it became a VGPR induction variable and 9 VALU ops per row to rebuild the pointers
            u32 roff;
            asm("s_mul_i32 %0, %1, %2" : "=s"(roff) : "s"(c), "s"(plane_pitch));
            u32 so = st_off;
            asm volatile("" : "+v"(so));                // keeps the lane offset out of a hoisted 64-bit VGPR pointer
            (splane + roff)[so] = (uint8_t)w;
            (cplane + roff)[so] = (uint8_t)(w >> 8);
          }
        }
      }
    }
    wave_lds_sync();  // the next sub-chunk's phase 1 overwrites the slab
  }
}

template <int IN>
static hipError_t launch_front_t(const FrontParams &p, hipStream_t s)
{
  const int nblocks = (p.total_items + 3) / 4;
  hipLaunchKernelGGL((k_front<IN>), dim3(nblocks), dim3(256), front_lds_bytes(), s, p);
  return hipGetLastError();
}

hipError_t launch_front(const FrontParams &p, hipStream_t s)
{
  if (p.subchunks < 1 || p.run_rows != front_run_rows(p.subchunks) || p.nchunks * p.run_rows < p.H) return hipErrorInvalidValue;
  if ((unsigned long long)p.H * p.in_pitch >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit row offsets inside a frame
  return p.bgr == 2 ? launch_front_t<2>(p, s) : p.bgr == 1 ? launch_front_t<1>(p, s) : launch_front_t<0>(p, s);
}

// =================================================================================================
// k_blur + k_nms: the front path as two kernels ("split" mode, the default)
// =================================================================================================
// The fused k_front above carries the vertical blur accumulators through its Sobel/NMS phase and the
// Sobel/NMS rings through its blur phase: ~90 VGPRs, 4-5 waves per SIMD, and a VALU pipe that is only
// ~75 % busy.  Split in two, each half needs < 64 VGPRs (8 waves per SIMD), no LDS slab, and runs can be
// long (a 4-row warm-up per 135 rows instead of 8 per 68).  The price is one u8 blur plane through HBM
// (1 B/px written, 1 B/px read back): 4 MB per 1080p frame against kernels that are VALU-bound.
//   k_blur   input rows -> exact Gaussian blur (u8), written to the blur plane
//   k_nms    blur plane -> Sobel -> S -> direction -> NMS -> thresholds -> the two bit planes
// Blur plane layout: [frame][strip][H][256 B] -- every wave-row of k_blur is one aligned 256-byte store of
// all 64 lanes (two full 128 B lines; a [H][W] plane would take 248-byte pieces at unaligned offsets, and
// the L2 fetches every partially written line first: measured 2.6x read amplification).  Bytes 4..251 of a
// segment row are the strip's own 248 columns; the two halo dwords are junk and never read: k_nms takes its
// halo columns from the neighbouring strips' segments.
#ifndef HC_BSUB
#define HC_BSUB 24
#define HC_BRING 32
#define HC_BG 8
#endif
constexpr int BSUB = HC_BSUB;   // blur rows between two fix-up passes of k_blur
constexpr int BRING = HC_BRING;  // input rows (masked, grey) kept in a wave-private LDS ring for the fix-up: >= BSUB + 4, power of 2
constexpr int BLUR_WAVE_BYTES = BRING * 256 + QCAP * 4;

// the literal reference chain (cannyEdgeD.cu:102-115) on the LDS ring: rows and columns outside the image
// are stored as 0 there, and a 0 tap leaves the running sum unchanged (c * 0 = 0, f + 0 = f), exactly as
// the reference's skipped taps do -- so no bounds checks and no scattered global loads.
static __device__ __forceinline__ u32 gauss_chain_lds(const unsigned char *ring, int row, u32 colbyte)
{
  float f = 0.0f;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const unsigned char *q = ring + (u32)((row - 2 + r) & (BRING - 1)) * 256u + colbyte - 2u;
#pragma unroll
    for (int c = 0; c < 5; ++c) f = __builtin_fmaf(GKC.v[r * 5 + c], (float)q[c], f);
  }
  return (u32)(int)f;
}

template <int IN>
__global__ __launch_bounds__(256) void k_blur(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  unsigned char *ring = smem + wib * BLUR_WAVE_BYTES;
  u32 *queue = reinterpret_cast<u32 *>(ring + BRING * 256);

  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items) return;
  const int chunk = item % p.nchunks;
  const int strip = (item / p.nchunks) % p.nstrips;
  const int frame = item / (p.nchunks * p.nstrips);  // output frame (per-channel mode: 3 per input frame)
  const int in_frame = IN == 2 ? frame / 3 : frame, ch = IN == 2 ? frame % 3 : 0;
  const int W = p.W, H = p.H;
  const int r0 = chunk * p.run_rows;  // blur rows [r0, rend)
  const int rend = min(r0 + p.run_rows, H);
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;

  u32 cmask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    cmask |= in ? (0xFFu << (8 * k)) : 0u;
  }
  const bool own_lane = lane >= 1 && lane <= 62;                // lanes 0 and 63 only feed their neighbours' taps
  // "undecidable" flag positions this lane is responsible for: its own 4 pixels, and in the halo lanes the two pixels
  // next to the strip (k_nms needs 2 valid blur columns beyond each side; all their taps lie inside this wave)
  const u32 hmask = cmask & (own_lane ? 0x80808080u : lane == 0 ? 0x80800000u : 0x00008080u);
  const bool col_any = cmask != 0;
  const uint8_t *frame_base = p.in + (size_t)in_frame * p.in_frame_stride;
  const u32 in_pitch32 = (u32)p.in_pitch;                       // launch_blur checks H * pitch < 2^32
  const u32 ld_off = (u32)((IN ? 3 : 1) * c0);
  const u32 selA = ch == 0 ? 0x0c060300u : ch == 1 ? 0x0c070401u : 0x0c0c0502u;
  const u32 selB = ch == 0 ? 0x05020100u : ch == 1 ? 0x06020100u : 0x07040100u;
  uint8_t *bseg = p.blur + (size_t)frame * p.blur_frame_stride + (size_t)strip * H * 256;  // this strip's segment (wave-uniform)
  const u32 bo = (u32)(4 * lane);

  // 12 bytes of interleaved 3-channel data -> the lane's 4 pixels: one channel (IN == 2) or the grey value (IN == 1)
  auto from3 = [&](u32 d0, u32 d1, u32 d2) -> u32 {
    if (IN == 2) return __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, selA), selB);
    const u32 wts = 0x00132607u;
    const u32 m0 = __builtin_amdgcn_udot4(d0, wts, 0u, false) >> 6;
    const u32 m1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), wts, 0u, false) >> 6;
    const u32 m2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), wts, 0u, false) >> 6;
    const u32 m3 = __builtin_amdgcn_udot4(d2 >> 8, wts, 0u, false) >> 6;
    return m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
  };
  auto load_row = [&](int row) -> u32 {  // same input forms as k_front (only the four warm-up rows of a run come this way)
    u32 v = 0;
    if (row >= 0 && row < H && col_any) {
      const uint8_t *rowp = frame_base + (u32)row * in_pitch32;
      u32 lo = ld_off;
      asm volatile("" : "+v"(lo));
      if (IN != 0) {
        const u32 *q = reinterpret_cast<const u32 *>(rowp + lo);
        v = from3(q[0], q[1], q[2]);
      } else v = *reinterpret_cast<const u32 *>(rowp + lo);
    }
    return v;
  };

  // vertical accumulators, as in k_front (see the derivation there)
  u32 a1[2] = { 0, 0 }, a2[2] = { 0, 0 }, a3[2] = { 0, 0 }, a4[2] = { 0, 0 };
  auto accumulate = [&](u32 xraw, u32 Sp[2]) {
    const u32 x = xraw & cmask;
    const u32 A = unpack_lo(x), B = unpack_hi(x);
    const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
    const u32 m1 = pair_shift(A, Bl), p1 = pair_shift(B, A), p3 = pair_shift(Ar, B);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32 P = h == 0 ? Bl + B : A + Ar;
      const u32 Q = h == 0 ? m1 + p1 : p1 + p3;
      const u32 Cc = h == 0 ? A : B;
      const u32 e = pk_mad2(Q, P);
      const u32 h0 = pk_mad2(e, pk_mul5(Cc));
      const u32 w = pk_mad2(Cc, Q);
      const u32 h1 = pk_mad2(h0, w);
      const u32 h2 = (h0 + h1) - (P + w);
      Sp[h] = a4[h] + h0;
      a4[h] = a3[h] + h1;
      a3[h] = a2[h] + h2;
      a2[h] = a1[h] + h1;
      a1[h] = h0;
    }
  };

  u32 fifteen = 15u;
  asm volatile("" : "+v"(fifteen));  // the SDWA shift takes its count from a VGPR
  int qn = 0;
  // input row rb + 2 completes blur row rb; `slot` = rb - b0 inside the current fix-up window
  auto blur_row = [&](int rb, int slot, u32 xraw) {
    u32 Sp[2];
    reinterpret_cast<u32 *>(ring)[((rb + 2) & (BRING - 1)) * 64 + lane] = xraw & cmask;
    accumulate(xraw, Sp);
    if (rb < rend) {  // wave-uniform (rb >= r0 >= 0 by construction)
      const u16x2 mlo = { 52759, 0 }, mhi = { 0, 52759 };
      // (S * 52759) >> 15 is 16 bits: quotient byte above fraction byte.  The second pixel of a pair is shifted
      // straight into the upper half of the first one's register (SDWA), so two v_perm collect the four quotient
      // and the four fraction bytes.
      u32 t01 = __builtin_amdgcn_udot2(U(Sp[0]), mlo, 0u, false) >> 15;
      u32 t23 = __builtin_amdgcn_udot2(U(Sp[1]), mlo, 0u, false) >> 15;
      const u32 p1 = __builtin_amdgcn_udot2(U(Sp[0]), mhi, 0u, false), p3 = __builtin_amdgcn_udot2(U(Sp[1]), mhi, 0u, false);
      asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(t01) : "v"(fifteen), "v"(p1));
      asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(t23) : "v"(fifteen), "v"(p3));
      const u32 bl = __builtin_amdgcn_perm(t23, t01, 0x07050301u) & cmask;  // bytes f0 q0 f1 q1 | f2 q2 f3 q3 -> q0..q3
      const u32 fz = __builtin_amdgcn_perm(t23, t01, 0x06040200u);         // -> f0..f3
      const u32 hz = (fz - 0x01010101u) & ~fz & hmask;
      const u64 any = __ballot(hz != 0);
      if (any != 0) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(any >> 32), __builtin_amdgcn_mbcnt_lo((u32)any, (u32)qn));
        if (hz != 0 && rank < (u32)QCAP) queue[rank] = hz | (u32)lane | ((u32)slot << 8);
        qn += __popcll(any);
      }
      {
        u32 o = bo;
        asm volatile("" : "+v"(o));
        *reinterpret_cast<u32 *>(bseg + (u32)rb * 256u + o) = bl;  // all 64 lanes: one aligned 256-byte row
      }
    }
  };

  // The kernel is bound by memory latency, not arithmetic (a row is ~60 VALU ops): the G rows of the next
  // group are requested before the current group is processed.  (The compiler drains the memory counter
  // once per loop trip, vmcnt(0), so loads issued inside the group would be waited for almost at once.)
  constexpr int G = HC_BG;
  static_assert(BSUB % G == 0 && BRING >= BSUB + 4 && (BRING & (BRING - 1)) == 0, "fix-up windows are whole groups; the ring holds a window and its 4 halo rows");
  {  // warm-up: input rows r0-2 .. r0+1 only feed the accumulators
    u32 xw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xw[j] = load_row(r0 - 2 + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 Sp[2];
      reinterpret_cast<u32 *>(ring)[((r0 - 2 + j) & (BRING - 1)) * 64 + lane] = xw[j] & cmask;
      accumulate(xw[j], Sp);
    }
  }
  // The loads of the main loop are unconditional (rows clamped to what the run needs, lanes without an image column
  // read the row's first bytes; both are masked when the row is used), kept as loaded (1 or 3 dwords per row) and
  // converted when the row is consumed, and each row's registers are refilled as soon as it has been consumed: a row
  // waits for the oldest of G requests in flight.  With conditional loads the compiler can only wait for everything,
  // the wave's stores included (see k_nms); converting at the load made the wave wait for each request at once.
  constexpr int ND = IN == 0 ? 1 : 3;  // dwords per lane and row
  const int rlast = min(H - 1, rend + 1);  // last input row this run needs
  const u32 ld_safe = col_any ? ld_off : 0u;
  struct Raw { u32 d[ND]; };
  auto load_raw = [&](int row) -> Raw {
    u32 lo = ld_safe;
    asm volatile("" : "+v"(lo));
    const u32 *q = reinterpret_cast<const u32 *>(frame_base + (u32)min(max(row, 0), rlast) * in_pitch32 + lo);
    Raw r;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = q[i];
    return r;
  };
  auto use_raw = [&](int row, const Raw &r) -> u32 {
    u32 x;
    if constexpr (IN == 0) x = r.d[0];
    else x = from3(r.d[0], r.d[ND > 1 ? 1 : 0], r.d[ND > 2 ? 2 : 0]);
    if ((u32)row >= (u32)H) x = 0;  // wave-uniform: zero padding below the image (above it: the warm-up rows)
    return x;
  };
  Raw xn[G];
#pragma unroll
  for (int j = 0; j < G; ++j) xn[j] = load_raw(r0 + 2 + j);
  int wb0 = r0;  // first blur row of the current fix-up window
#pragma nounroll
  for (int rb0 = r0; rb0 < rend; rb0 += G) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const u32 x = use_raw(rb0 + 2 + j, xn[j]);
      xn[j] = load_raw(rb0 + 2 + G + j);
      blur_row(rb0 + j, rb0 - wb0 + j, x);
    }
    if (rb0 + G - wb0 < BSUB && rb0 + G < rend) continue;
    // fix-up of the window [wb0, rb0 + G): the queued pixels get the literal chain, written over the plane
    // bytes (same wave, program order).  Queue overflow (flat areas): every pixel of the window is recomputed.
    wave_lds_sync();
    if (qn <= QCAP) {
#pragma nounroll
      for (int base = 0; base < qn; base += 64) {
        const int e = base + lane;
        const u32 ent = e < qn ? queue[e] : 0u;
        u32 fl = ent & 0x80808080u;
        const u32 el = ent & 63u, es = (ent >> 8) & 31u;
        while (fl) {
          const u32 k = (u32)__builtin_ctz(fl) >> 3;
          fl &= fl - 1;
          const int row = wb0 + (int)es;
          bseg[(u32)row * 256u + el * 4u + k] = (unsigned char)gauss_chain_lds(ring, row, el * 4u + k);
        }
      }
    } else {
      const int nrows = min(rb0 + G, rend) - wb0;
      constexpr int FIXW = STRIP_W + 4;  // the strip's columns and two on each side (bytes 2 .. 253 of a segment row)
#pragma nounroll
      for (int e = lane; e < nrows * FIXW; e += 64) {
        const int row = wb0 + e / FIXW;
        const u32 b = 2u + (u32)(e % FIXW);
        const int col = strip * STRIP_W - STRIP_HALO + (int)b;
        if (col >= 0 && col < W) bseg[(u32)row * 256u + b] = (unsigned char)gauss_chain_lds(ring, row, b);
      }
    }
    wave_lds_sync();
    wb0 = rb0 + G;
    qn = 0;
  }
}

// The NMS of the rare candidate pixels is queued (see "NMS queue" in the kernel): entries per wave and their size
#ifndef HC_NQ_INLINE
#define HC_NQ_INLINE 16  // a row with more queued lanes than this runs the NMS wave-wide instead
#endif
constexpr int NQ_INLINE = HC_NQ_INLINE;
constexpr int NQ_CAP = 64 + NQ_INLINE;  // a batch is taken as soon as 64 entries wait, so at most 63 + NQ_INLINE are ever queued
constexpr int NQ_DW = 24;               // dwords per entry: 3 x 6 S values, 2 X pairs, 2 Y pairs, (row, lane), pad
constexpr int NMS_WAVE_BYTES = NQ_CAP * NQ_DW * 4;
static_assert(NQ_CAP % 2 == 0 && NQ_INLINE % 2 == 0, "entries come in lane pairs");

__global__ __launch_bounds__(256) void k_nms(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  typedef __attribute__((address_space(3))) u32 lds_u32;
  lds_u32 *nq = (lds_u32 *)(smem + wib * NMS_WAVE_BYTES);  // wave-private
  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items_b) return;
  const int chunk = item % p.nchunks_b;
  const int strip = (item / p.nchunks_b) % p.nstrips;
  const int frame = item / (p.nchunks_b * p.nstrips);
  const int W = p.W, H = p.H;
  const int r0 = chunk * p.run_rows_b;  // output rows [r0, rend)
  const int rend = min(r0 + p.run_rows_b, H);
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;

  u32 cmask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    cmask |= in ? (0xFFu << (8 * k)) : 0u;
  }
  const u32 pm0 = __builtin_amdgcn_perm(0u, cmask, 0x01010000u), pm1 = __builtin_amdgcn_perm(0u, cmask, 0x03030202u);
  const u32 oknib1 = (lane >= 1 && lane <= 62) ? ((cmask & 1u) | ((cmask >> 7) & 2u) | ((cmask >> 14) & 4u) | ((cmask >> 21) & 8u)) : 0u;
  const u32 oknib = oknib1 | (oknib1 << 8);
  const u32 bm0 = pm0, bm1 = pm1;  // packed-i16 column masks of the two pixel pairs
  const uint8_t *bframe = p.blur + (size_t)frame * p.blur_frame_stride;  // wave-uniform
  const bool col_any = cmask != 0;  // false for lane 0 of strip 0 and for lanes right of the image (incl. a missing strip+1)
  // every lane reads its own dword of this strip's segment: one aligned 256-byte row per wave.  The halo lanes' two
  // pixels next to the strip are exact there (k_blur fixes them up like the strip's own), the outer two are never used.
  const u32 bo = (u32)strip * (u32)H * 256u + (u32)(4 * lane);
  const u32 lane_keep = col_any ? ~0u : 0u;
  const int klast = min(H - 1, rend + 1);  // last blur row this run needs
  // The loads are unconditional -- every lane reads a valid dword of the plane (rows clamped, lanes without an image
  // column read their own strip's) and what must read as zero padding (cannyEdgeD.cu:150-156) is masked when the
  // row is used.  An s_waitcnt for a conditional load could not count the loads issued after it and would drain
  // the wave's stores as well.
  auto load_b = [&](int k) -> u32 {
    u32 o = bo;
    asm volatile("" : "+v"(o));
    return *reinterpret_cast<const u32 *>(bframe + (u32)min(max(k, 0), klast) * 256u + o);
  };
  auto use_b = [&](int k, u32 v) -> u32 {
    u32 r = v & lane_keep;
    if ((u32)k >= (u32)H) r = 0;  // wave-uniform
    return r;
  };

  u32 dr[2][2], sr[2][2];  // d and s of the two previous blur rows, [ring][pair]
  u32 Sr[3][6];            // S rows: [ring][0]=left neighbour, [1..4]=own 4 px, [5]=right neighbour
  u32 Xr[2][2], Yr[2][2];  // packed sumX / sumY pairs of the two newest Sobel rows
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) dr[a][b] = sr[a][b] = Xr[a][b] = Yr[a][b] = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) Sr[a][b] = 0;

  const size_t plane_off = (size_t)frame * H * p.RD * 4;
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const bool store_lane = (lane & 1) && lane < 63;
  const u32 st_off = (u32)(strip * 31 + (lane >> 1));
  const u32 a_lo0 = p.a_lo[0], a_hi0 = p.a_hi[0], wrap_limit = p.wrap_limit;
  const u32 plane_pitch = (u32)p.RD * 4u;

  // ---- NMS queue ------------------------------------------------------------------------------------------------
  // About 5 % of the pixels pass the low threshold and 6 % of the lanes hold one, yet three wave-rows in four contain
  // some: a wave-wide NMS spends nearly all of its lanes on pixels that are dropped anyway.  Instead a lane with a
  // candidate -- and its partner in the output byte, lanes 2j+1 / 2j+2 -- parks what the NMS needs (the 3 x 6 S
  // values around its 4 pixels, the Sobel sums of the centre row, its row and lane) in a wave-private LDS queue;
  // once 64 entries wait, one dense pass does them, an entry per lane, and stores their bytes.  All other bytes
  // of a row are written as zeros straight away.  A row with many candidates (a horizontal edge) runs wave-wide.
  int qhead = 0, qcount = 0;  // wave-uniform; both even
  // lanes that store a dword of the provisional map / a byte of the planes, and where row r0 starts (both advance by
  // a pitch per output row: scalar adds instead of a multiply per row)
  const u64 m_prov = uniform64(__ballot(lane >= 1 && lane <= 62 && c0 < W));
  const u64 m_st = uniform64(__ballot(store_lane));
  const u32 prov_voff = (u32)(strip * STRIP_W + 4 * (lane - 1));
  u32 plane_roff = (u32)r0 * plane_pitch;
  uint8_t *prov_row = p.prov_out ? p.prov_out + (size_t)frame * p.prov_fs + (size_t)r0 * p.prov_pitch : nullptr;
  auto nms_batch = [&](int nent) {
    wave_lds_sync();
    int idx = qhead + lane;
    if (idx >= NQ_CAP) idx -= NQ_CAP;
    const bool live = lane < nent;  // the other lanes compute on stale entries and store nothing
    u32 v[NQ_DW];
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const lds_u32x4 *ent = (const lds_u32x4 *)(nq + idx * NQ_DW);
#pragma unroll
    for (int j = 0; j < NQ_DW / 4; ++j) {
      const u32x4 t = ent[j];
      v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
    const u32 *SU = v, *SC = v + 6, *SN = v + 12;  // S rows above / at / below, [0] and [5] the neighbouring lanes' pixels
    const u32 gmax = max(max(SC[1], SC[2]), max(SC[3], SC[4]));
    const bool wraps = __ballot(live && gmax >= wrap_limit) != 0;
    u32 nibS = 0, nibC = 0;
    auto px = [&](auto hc, auto ec, u32 A2, u32 Um, u32 Vp) {
      constexpr int e = decltype(ec)::value, q = 2 * decltype(hc)::value + e;
      const u32 g = SC[1 + q];
      bool cand = g >= a_lo0, strong = g >= a_hi0;
      if (wraps) {  // u8 wrap of gradients >= 256: the bands of S whose low byte passes the thresholds
        const bool w0 = g >= 262144u, w1 = g >= 1048576u;
        cand = (cand && !w0) || (g >= p.a_lo[1] && !w1) || g >= p.a_lo[2];
        strong = (strong && !w0) || (g >= p.a_hi[1] && !w1) || g >= p.a_hi[2];
      }
      const bool p1 = mul16<e, e>(A2, Um) > (int)g, p2 = mul16<e, e>(A2, Vp) > (int)g;
      const u32 m0 = max(SN[1 + q], SU[1 + q]), m1 = max(SN[q], SU[2 + q]);
      const u32 m2 = max(SC[2 + q], SC[q]), m3 = max(SU[q], SN[2 + q]);
      const u32 mb = p1 ? (p2 ? m2 : m3) : (p2 ? m1 : m0);
      const bool keep = mb <= g;
      nibS |= (strong && keep) ? (1u << q) : 0u;
      nibC |= (cand && keep) ? (1u << q) : 0u;
    };
    auto pr = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      const u32 X = v[18 + h], Y = v[20 + h];
      const u32 A2 = R(U(X) + U(X)), Um = R(I(X) - I(Y)), Vp = R(U(X) + U(Y));
      px(hc, std::integral_constant<int, 0>{}, A2, Um, Vp);
      px(hc, std::integral_constant<int, 1>{}, A2, Um, Vp);
    };
    pr(std::integral_constant<int, 0>{});
    pr(std::integral_constant<int, 1>{});
    const u32 nib = nibS | (nibC << 8);
    const u32 w = nib | (from_lane_above(nib) << 4);  // the partner's entry sits in the next lane
    const u32 sl = v[22] & 63u, row = v[22] >> 8;
    if (live && (sl & 1u)) {
      const u32 o = row * plane_pitch + (u32)(strip * 31) + (sl >> 1);
      splane[o] = (uint8_t)w;
      cplane[o] = (uint8_t)(w >> 8);
    }
    if (p.prov_out && live && strip * STRIP_W - STRIP_HALO + 4 * (int)sl < W)
      *reinterpret_cast<u32 *>(p.prov_out + (size_t)frame * p.prov_fs + (size_t)row * p.prov_pitch + (u32)(strip * STRIP_W) + 4u * (sl - 1u)) = nibble_to_bytes(nibS);
    qhead += nent;
    if (qhead >= NQ_CAP) qhead -= NQ_CAP;
    qcount -= nent;
  };

  // one step: blur row k arrives -> Sobel row k-1 -> NMS/threshold row k-2 (see k_front's phase 2)
  auto step = [&](auto uc, int k, u32 b) {
    constexpr int u = decltype(uc)::value;
    constexpr int rn = u % 2, rp = (u + 1) % 2;
    constexpr int sN = u % 3, sC = (u + 2) % 3, sU = (u + 1) % 3;
    const u32 A = unpack_lo(b), B = unpack_hi(b);
    const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
    const u32 m1 = pair_shift(A, Bl), p1 = pair_shift(B, A), p3 = pair_shift(Ar, B);
    u32 dk[2], sk[2];
    dk[0] = R(I(p1) - I(m1));
    sk[0] = pk_mad2(A, m1 + p1);
    dk[1] = R(I(p3) - I(p1));
    sk[1] = pk_mad2(B, p1 + p3);
    const int i = k - 1;
    const bool rowbad = (u32)i >= (u32)H;  // the Sobel rows just above / below the image are 0 (zero padding of every stage)
    u32 Xv[2], Yv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // column masks: only the first and the last strips have columns outside the image (bmask is all ones elsewhere)
      Xv[h] = pk_mad2(dr[rp][h], R(I(dr[rn][h]) + I(dk[h]))) & (h == 0 ? bm0 : bm1);
      Yv[h] = R(I(sr[rn][h]) - I(sk[h])) & (h == 0 ? bm0 : bm1);
    }
    if (rowbad) {  // wave-uniform, two rows per frame
      asm volatile("" ::: "memory");  // keeps this one branch: as selects it would cost every row extra VALU ops
      Xv[0] = Xv[1] = Yv[0] = Yv[1] = 0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32 X = Xv[h], Y = Yv[h];
      Xr[rn][h] = X;
      Yr[rn][h] = Y;
      Sr[sN][1 + 2 * h] = (u32)mad16<0, 0>(X, X, mul16<0, 0>(Y, Y));
      Sr[sN][2 + 2 * h] = (u32)mad16<1, 1>(X, X, mul16<1, 1>(Y, Y));
    }
    Sr[sN][0] = from_lane_below(Sr[sN][4]);
    Sr[sN][5] = from_lane_above(Sr[sN][1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) { dr[rn][h] = dk[h]; sr[rn][h] = sk[h]; }

    const int c = k - 2;
    if ((u32)(c - r0) < (u32)(rend - r0)) {
      u64 cl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cl[q] = __ballot(Sr[sC][1 + q] >= a_lo0);
      const u64 any = (cl[0] | cl[1] | cl[2] | cl[3]) & 0x7FFFFFFFFFFFFFFEull;  // lanes 0 and 63 only carry halo columns
      const u64 pm = any | ((any & 0x2AAAAAAAAAAAAAAAull) << 1) | ((any & 0x5555555555555554ull) >> 1);  // + partners
      u32 nadd;  // as an instruction: the builtin's result is widened and the comparison below lands on the VALU
      asm("s_bcnt1_i32_b64 %0, %1" : "=s"(nadd) : "s"(pm) : "scc");
      if (nadd <= (u32)NQ_INLINE) {
        // the common case: few (or no) candidate lanes.  They go to the queue, every other byte of the row is zero.
        if (any != 0) {
          if (__builtin_amdgcn_inverse_ballot_w64(pm)) {
            int idx = qhead + qcount + (int)mbcnt64(pm);
            if (idx >= NQ_CAP) idx -= NQ_CAP;
            lds_u32 *ent = nq + idx * NQ_DW;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
              ent[j] = Sr[sU][j];
              ent[6 + j] = Sr[sC][j];
              ent[12 + j] = Sr[sN][j];
            }
            ent[18] = Xr[rp][0];
            ent[19] = Xr[rp][1];
            ent[20] = Yr[rp][0];
            ent[21] = Yr[rp][1];
            ent[22] = ((u32)c << 8) | (u32)lane;
          }
          qcount += (int)nadd;
        }
        if (p.prov_out && __builtin_amdgcn_inverse_ballot_w64(m_prov & ~pm)) {
          u32 o = prov_voff;
          asm volatile("" : "+v"(o));
          *reinterpret_cast<u32 *>(prov_row + o) = 0u;
        }
        if (__builtin_amdgcn_inverse_ballot_w64(m_st & ~pm)) {  // a queued pair's byte is stored by its batch
          u32 so = st_off;
          asm volatile("" : "+v"(so));
          (splane + plane_roff)[so] = 0;
          (cplane + plane_roff)[so] = 0;
        }
        if (qcount >= 64) nms_batch(64);
      } else {
        // many candidates (a horizontal edge): the NMS on the whole wave
        u64 st[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) st[q] = __ballot(Sr[sC][1 + q] >= a_hi0);
        if ((st[0] | st[1] | st[2] | st[3]) != 0 && __ballot(max(max(Sr[sC][1], Sr[sC][2]), max(Sr[sC][3], Sr[sC][4])) >= wrap_limit) != 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32 g = Sr[sC][1 + q];
            const u64 w0 = __ballot(g >= 262144u), w1 = __ballot(g >= 1048576u);
            cl[q] = (cl[q] & ~w0) | (__ballot(g >= p.a_lo[1]) & ~w1) | __ballot(g >= p.a_lo[2]);
            st[q] = (st[q] & ~w0) | (__ballot(g >= p.a_hi[1]) & ~w1) | __ballot(g >= p.a_hi[2]);
          }
        }
        u32 nibS = 0, nibC = 0;
        auto slot = [&](auto hc, auto ec, u32 A2, u32 Um, u32 Vp) {
          constexpr int h = decltype(hc)::value, e = decltype(ec)::value, q = 2 * h + e;
          u64 mS = 0, mC = 0;
          if (cl[q] != 0) {
            const u32 g = Sr[sC][1 + q];
            const bool p1 = mul16<e, e>(A2, Um) > (int)g, p2 = mul16<e, e>(A2, Vp) > (int)g;
            // the larger neighbour of each of the 4 directions (cannyEdgeD.cu:245-264: bin0 down/up, bin1
            // down-left/up-right, bin2 right/left, bin3 up-left/down-right), then the one of this pixel's bin.
            // All per lane on the VALU: the scalar unit, shared by the CU's 4 SIMDs, is as loaded as the VALU here.
            const u32 m0 = max(Sr[sN][1 + q], Sr[sU][1 + q]), m1 = max(Sr[sN][q], Sr[sU][2 + q]);
            const u32 m2 = max(Sr[sC][2 + q], Sr[sC][q]), m3 = max(Sr[sU][q], Sr[sN][2 + q]);
            const u32 mb = p1 ? (p2 ? m2 : m3) : (p2 ? m1 : m0);
            const u64 keep = __ballot(mb <= g);  // non-strict on both sides
            mS = st[q] & keep;
            mC = cl[q] & keep;
          }
          nibS = shift_in(nibS, mS);
          nibC = shift_in(nibC, mC);
        };
        auto pair = [&](auto hc) {
          constexpr int h = decltype(hc)::value;
          const u32 X = Xr[rp][h], Y = Yr[rp][h];
          const u32 A2 = R(U(X) + U(X));
          const u32 Um = R(I(X) - I(Y));
          const u32 Vp = R(U(X) + U(Y));
          slot(hc, std::integral_constant<int, 1>{}, A2, Um, Vp);
          slot(hc, std::integral_constant<int, 0>{}, A2, Um, Vp);
        };
        pair(std::integral_constant<int, 1>{});
        pair(std::integral_constant<int, 0>{});
        const u32 nib = (nibS | (nibC << 8)) & oknib;
        if (p.prov_out && __builtin_amdgcn_inverse_ballot_w64(m_prov)) {  // provisional 0/255 map (strong bits); W % 4 == 0 here
          u32 o = prov_voff;
          asm volatile("" : "+v"(o));
          *reinterpret_cast<u32 *>(prov_row + o) = nibble_to_bytes(nib & 0xFu);
        }
        const u32 w = nib | (from_lane_above(nib) << 4);
        if (store_lane) {
          u32 so = st_off;
          asm volatile("" : "+v"(so));
          (splane + plane_roff)[so] = (uint8_t)w;
          (cplane + plane_roff)[so] = (uint8_t)(w >> 8);
        }
      }
      plane_roff += plane_pitch;
      prov_row += p.prov_pitch;
    }
  };

  // blur rows r0-2 .. rend+1, six per loop trip (the ring period); the next trip's six rows are requested
  // before this trip's are processed
  const int k0 = r0 - 2, kend = rend + 2;
  // blur rows r0-2 .. rend+1, six steps per loop trip (the ring period).  Each row was requested six steps before it
  // is used; its register is refilled at once, so a step only ever waits for the oldest of six loads in flight.
  u32 bn[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) bn[j] = load_b(k0 + j);
  auto advance = [&](auto uc, int k) {
    constexpr int j = decltype(uc)::value;
    const u32 b = use_b(k, bn[j]);
    bn[j] = load_b(k + 6);
    step(uc, k, b);
  };
#pragma nounroll
  for (int k = k0; k < kend; k += 6) {
    advance(std::integral_constant<int, 0>{}, k + 0);
    advance(std::integral_constant<int, 1>{}, k + 1);
    advance(std::integral_constant<int, 2>{}, k + 2);
    advance(std::integral_constant<int, 3>{}, k + 3);
    advance(std::integral_constant<int, 4>{}, k + 4);
    advance(std::integral_constant<int, 5>{}, k + 5);
  }
  if (qcount > 0) nms_batch(qcount);
}

template <int IN>
static hipError_t launch_blur_t(const FrontParams &p, hipStream_t s)
{
  hipLaunchKernelGGL((k_blur<IN>), dim3((p.total_items + 3) / 4), dim3(256), (size_t)4 * BLUR_WAVE_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_blur(const FrontParams &p, hipStream_t s)
{
  if (p.run_rows < 1 || p.nchunks * p.run_rows < p.H || !p.blur || p.blur_frame_stride < (size_t)p.nstrips * p.H * 256) return hipErrorInvalidValue;
  if ((unsigned long long)p.H * p.in_pitch >= (1ull << 32) || (unsigned long long)(p.nstrips + 1) * p.H * 256 >= (1ull << 32)) return hipErrorInvalidValue;
  return p.bgr == 2 ? launch_blur_t<2>(p, s) : p.bgr == 1 ? launch_blur_t<1>(p, s) : launch_blur_t<0>(p, s);
}

hipError_t launch_nms(const FrontParams &p, hipStream_t s)
{
  if (p.run_rows_b < 1 || p.nchunks_b * p.run_rows_b < p.H || !p.blur) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_nms, dim3((p.total_items_b + 3) / 4), dim3(256), (size_t)4 * NMS_WAVE_BYTES, s, p);
  return hipGetLastError();
}


}  // namespace hc
