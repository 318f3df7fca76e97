// canny_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hipcanny hot path.
//
// What the reference does in 9+k launches over 25 B/px of intermediates (src/cvp/cannyEdgeD.cu,
// launch sites src/cvp/cannyEdgeH.cu:214-338) is done by k_front8 (front8.hip: the whole front path, one kernel) and, here,
//   k_hyst    edge hysteresis on the bit planes (64 px per 64-bit op), device-side convergence flag,
//             0/255 u8 edge map written (or, after the front kernel's provisional map, patched) by the same kernel
//   k_front_o the 4-px cv::Canny ("Mode O") front kernel (3-channel sources; one-channel sources: k_front8o, front8.hip)
// (the round-1 front kernels of Mode R -- k_front, k_blur + k_nms -- live in legacy_front.hip, outside the product library)
// plus the plain per-stage kernels behind the finalStage taps and k_pack for hc_hysteresis_device.
// MFMA is deliberately not used: there is no dense contraction (an f32 MFMA would reproduce the
// Gaussian's fmaf chain bit for bit, but as a banded 36x32 Toeplitz product it wastes 31/36 of its
// multiplies and runs at the f32 vector rate -- 4-7x slower than the packed integer form below).
//
// Numerical contract ("Mode R"): identical to the reference kernels, including the
// float Gaussian chain (via the exact integer shortcut explained in front8.hip), the u8 wrap of gradients
// >= 256 and the non-strict NMS.
#include "canny_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace hc {

// The reference uploads its coefficient table to __constant__ GK[5][5] (cannyEdgeD.cu:11, cannyEdgeH.cu:372-380); here the
// 25 values are literals of the instruction stream (as __constant__ loads they pinned 25 SGPRs across the hot loops), so
// the host's table -- computed the reference's way at hc_create -- is only CHECKED against them.
hipError_t check_gauss_coeffs(const float gk[25])
{
  return memcmp(gk, GKC.v, sizeof(GKC.v)) == 0 ? hipSuccess : hipErrorInvalidValue;
}

// =================================================================================================
// Self-test of the primitives above (hc_selftest): catches a wrong DPP direction or perm selector.
// =================================================================================================
__global__ void k_selftest(u32 *res)
{
  const u32 lane = threadIdx.x & 63;
  u32 bad = 0;
  bad |= (from_lane_below(lane + 100) != (lane ? lane + 99 : 0)) ? 1u : 0u;
  bad |= (from_lane_above(lane + 100) != (lane < 63 ? lane + 101 : 0)) ? 2u : 0u;
  const u32 x = 0x44332211u + lane;
  bad |= (unpack_lo(x) != ((x & 0xFF) | (((x >> 8) & 0xFF) << 16))) ? 4u : 0u;
  bad |= (unpack_hi(x) != (((x >> 16) & 0xFF) | ((x >> 24) << 16))) ? 8u : 0u;
  bad |= (pair_shift(0xAAAA1111u + lane, 0x2222BBBBu) != (((0xAAAA1111u + lane) << 16) | 0x2222u)) ? 16u : 0u;
  const u32 p0 = 0x00050003u, p1 = 0x00090007u;
  bad |= (__builtin_amdgcn_perm(p1, p0, 0x06040200u) != 0x09070503u) ? 32u : 0u;
  bad |= (__builtin_amdgcn_perm(0x00BB00AAu, 0x00220011u, 0x05040100u) != 0x00AA0011u) ? 64u : 0u;
  bad |= (__builtin_amdgcn_perm(0x00BB00AAu, 0x00220011u, 0x07060302u) != 0x00BB0022u) ? 128u : 0u;
  bad |= (__builtin_amdgcn_udot2(U(0x9E61u | (40545u << 16)), U(0x0000CE17u), 0u, false) != 0x9E61u * 52759u) ? 256u : 0u;
  bad |= (__builtin_amdgcn_sdot2(I(0xFC04u | (1020u << 16)), I(0xFC04u | (1020u << 16)), 0, false) != 2 * 1020 * 1020) ? 512u : 0u;
  const u64 m = __ballot(lane & 1);
  bad |= (m != 0xAAAAAAAAAAAAAAAAull) ? 1024u : 0u;
  bad |= (mbcnt64(m) != lane / 2) ? 2048u : 0u;
  bad |= (shift_in(lane, m) != 2 * lane + (lane & 1)) ? 4096u : 0u;
  {
    const int xl = (int)lane - 32, xh = 1000 - 3 * (int)lane;            // packed i16 pair (xl, xh)
    const u32 pk = ((u32)xl & 0xFFFFu) | ((u32)xh << 16);
    bad |= (mul16<0, 0>(pk, pk) != xl * xl) ? 8192u : 0u;
    bad |= (mul16<1, 1>(pk, pk) != xh * xh) ? 8192u : 0u;
    bad |= (mad16<0, 1>(pk, pk, 7) != xl * xh + 7) ? 8192u : 0u;
    bad |= (mad16<1, 0>(pk, pk, -5) != xh * xl - 5) ? 8192u : 0u;
  }
  if (bad) atomicOr(res, bad);
}

hipError_t launch_selftest(u32 *d_result, hipStream_t s)
{
  hipLaunchKernelGGL(k_selftest, dim3(2), dim3(128), 0, s, d_result);
  return hipGetLastError();
}

// =================================================================================================
// k_front_o: "Mode O" -- cv::Canny(src 8UC1, low, high, apertureSize 3, L2gradient) semantics
// =================================================================================================
// OpenCV 4.x modules/imgproc/src/canny.cpp (the parity tests check it against a CPU restatement): no blur,
// Sobel 3x3 on the source with BORDER_REPLICATE, magnitude m = |dx|+|dy| (or dx^2+dy^2 with L2gradient; 0
// outside the image), pixels with m <= low are dropped, direction by the integer tangent test
// (TG22 = 13573, shift 15), asymmetric non-maximum suppression (m > first neighbour, m >= second on the
// axes; strict on both diagonal neighbours), m > high seeds.  Same strip / lane / DPP layout and the same
// bit-plane output as k_nms, so k_hyst finishes the job.  One pass, registers only (no LDS): a work item is
// (frame, strip, chunk of p.chunk_rows rows) with a 4-row warm-up.
// NC: channels of the (interleaved) source.  cv::Canny on a 3-channel image computes the Sobel derivatives of every
// channel and keeps, per pixel, those of the channel with the largest magnitude -- the first one on ties (canny.cpp).
template <bool L2, int NC>
__global__ __launch_bounds__(256) void k_front_o(const FrontParams p)
{
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items) return;
  const int chunk = item % p.nchunks;
  const int strip = (item / p.nchunks) % p.nstrips;
  const int frame = item / (p.nchunks * p.nstrips);
  const int W = p.W, H = p.H, CH = p.chunk_rows;
  const int r0 = chunk * CH, rend = min(r0 + CH, H);
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;

  // BORDER_REPLICATE along the row: the 4 columns of this lane, clamped into the image, always lie in
  // one aligned dword; a per-lane byte selector arranges (and repeats) them
  u32 cmask = 0, rsel = 0;
  const int cl0 = min(max(c0, 0), W - 1);
  const int ld_col = cl0 & ~3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    cmask |= in ? (0xFFu << (8 * k)) : 0u;
    const int cc = min(max(c0 + k, 0), W - 1);
    rsel |= (u32)(cc - ld_col) << (8 * k);  // 0..3: byte of the loaded dword (cc - ld_col < 4 by construction)
  }
  const u32 pm0 = __builtin_amdgcn_perm(0u, cmask, 0x01010000u), pm1 = __builtin_amdgcn_perm(0u, cmask, 0x03030202u);
  const u32 oknib1 = (lane >= 1 && lane <= 62) ? ((cmask & 1u) | ((cmask >> 7) & 2u) | ((cmask >> 14) & 4u) | ((cmask >> 21) & 8u)) : 0u;
  const u32 oknib = oknib1 | (oknib1 << 8);
  const uint8_t *fbase = p.in + (size_t)frame * p.in_frame_stride;  // wave-uniform
  const u32 in_pitch32 = (u32)p.in_pitch;                            // launch_front_o checks H * pitch < 2^32
  const u32 ld_off = (u32)(NC * ld_col);
  const int rlast = min(H - 1, rend + 1);  // last source row this run needs
  struct RawO { u32 d[NC]; };  // the lane's 4 pixels as loaded: 4 bytes, or 12 interleaved ones
  auto load_row = [&](int row) -> RawO {  // BORDER_REPLICATE along the column: clamp the row; the raw dwords (see use_row)
    const int rr = min(max(row, 0), rlast);
    u32 roff;
    asm("s_mul_i32 %0, %1, %2" : "=s"(roff) : "s"(rr), "s"(in_pitch32));
    u32 o = ld_off;
    asm volatile("" : "+v"(o));  // scalar row base + 32-bit lane offset
    const u32 *q = reinterpret_cast<const u32 *>(fbase + roff + o);
    RawO r;
#pragma unroll
    for (int i = 0; i < NC; ++i) r.d[i] = q[i];
    return r;
  };
  // BORDER_REPLICATE along the row: the byte selector, applied when the row is consumed -- applied at the load it
  // made the wave wait for each request at once.  3-channel data: channel ch of the 4 pixels is picked out of the
  // 12 interleaved bytes first (bytes ch, ch+3, ch+6, ch+9).
  auto use_row = [&](const RawO &raw, int ch) -> u32 {
    u32 v = raw.d[0];
    if constexpr (NC == 3) {
      const u32 selA = ch == 0 ? 0x0c060300u : ch == 1 ? 0x0c070401u : 0x0c0c0502u;
      const u32 selB = ch == 0 ? 0x05020100u : ch == 1 ? 0x06020100u : 0x07040100u;
      v = __builtin_amdgcn_perm(raw.d[NC > 2 ? 2 : 0], __builtin_amdgcn_perm(raw.d[NC > 1 ? 1 : 0], raw.d[0], selA), selB);
    }
    return __builtin_amdgcn_perm(0u, v, rsel);
  };

  u32 dr[NC][2][2], sr[NC][2][2];  // per channel: d = x[+1]-x[-1] and s = x[-1]+2x[0]+x[+1] of the two previous rows, [ring][pair]
  u32 Mr[3][6];            // magnitude rows: [ring][0]=left neighbour, [1..4]=own 4 px, [5]=right neighbour
  u32 Xr[2][2], Yr[2][2];  // packed dx / dy pairs of the two newest gradient rows
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      Xr[a][b] = Yr[a][b] = 0;
#pragma unroll
      for (int ch = 0; ch < NC; ++ch) dr[ch][a][b] = sr[ch][a][b] = 0;
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) Mr[a][b] = 0;
  const size_t plane_off = (size_t)frame * H * p.RD * 4;
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const bool store_lane = (lane & 1) && lane < 63;
  const u32 st_off = (u32)(strip * 31 + (lane >> 1));
  const u32 plane_pitch = (u32)p.RD * 4u;
  const u32 low = p.a_lo[0], high = p.a_hi[0];  // Mode O: plain thresholds on m
  const u32 k_tg22 = 13573u, k_m32768 = 0x8000u;  // 16-bit multiplier operands (low halves): TG22 and -2^15

  // one step: source row k arrives -> gradient row k-1 -> NMS / threshold row k-2
  auto step = [&](auto uc, int k, const RawO &raw) {
    constexpr int u = decltype(uc)::value;
    constexpr int rn = u % 2, rp = (u + 1) % 2;
    constexpr int sN = u % 3, sC = (u + 2) % 3, sU = (u + 1) % 3;
    const int i = k - 1;  // gradient row from source rows k-2 (ring rn), k-1 (ring rp), k (new)
    const bool rowbad = i < 0 || i >= H;  // magnitude outside the image is 0
    u32 Xv[2] = { 0, 0 }, Yv[2] = { 0, 0 };   // dx / dy of the channel kept so far, [pair]
    u32 Mp[2] = { 0, 0 };                     // L1: its packed magnitudes
    u32 M4[4] = { 0, 0, 0, 0 };               // L2: its magnitudes, one per pixel
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
      const u32 b = use_row(raw, ch);
      const u32 A = unpack_lo(b), B = unpack_hi(b);
      const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
      const u32 m1 = pair_shift(A, Bl), p1 = pair_shift(B, A), p3 = pair_shift(Ar, B);
      u32 dk[2], sk[2];
      dk[0] = R(I(p1) - I(m1));
      sk[0] = pk_mad2(A, m1 + p1);
      dk[1] = R(I(p3) - I(p1));
      sk[1] = pk_mad2(B, p1 + p3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32 X = pk_mad2(dr[ch][rp][h], R(I(dr[ch][rn][h]) + I(dk[h]))) & (h == 0 ? pm0 : pm1);  // dx = right - left, smoothed 1-2-1 down the rows
        const u32 Y = R(I(sk[h]) - I(sr[ch][rn][h])) & (h == 0 ? pm0 : pm1);                          // dy = bottom - top
        dr[ch][rn][h] = dk[h];
        sr[ch][rn][h] = sk[h];
        if (L2) {
          const u32 ma = (u32)mad16<0, 0>(X, X, mul16<0, 0>(Y, Y)), mb = (u32)mad16<1, 1>(X, X, mul16<1, 1>(Y, Y));
          if (ch == 0) { Xv[h] = X; Yv[h] = Y; M4[2 * h] = ma; M4[2 * h + 1] = mb; }
          else {  // strictly larger: ties keep the earlier channel
            const bool ta = ma > M4[2 * h], tb = mb > M4[2 * h + 1];
            const u32 msk = (ta ? 0x0000FFFFu : 0u) | (tb ? 0xFFFF0000u : 0u);
            Xv[h] = (X & msk) | (Xv[h] & ~msk);
            Yv[h] = (Y & msk) | (Yv[h] & ~msk);
            M4[2 * h] = ta ? ma : M4[2 * h];
            M4[2 * h + 1] = tb ? mb : M4[2 * h + 1];
          }
        } else {  // |dx| + |dy| <= 2040 per half: packed
          const u32 mp = R(__builtin_elementwise_max(I(X), -I(X))) + R(__builtin_elementwise_max(I(Y), -I(Y)));
          if (ch == 0) { Xv[h] = X; Yv[h] = Y; Mp[h] = mp; }
          else {
            // per half: 0xFFFF where this channel's magnitude is strictly larger (saturating difference, then 0 - min(d, 1))
            const u16x2 dif = __builtin_elementwise_sub_sat(U(mp), U(Mp[h]));
            const u16x2 one = { 1, 1 }, zero = { 0, 0 };
            const u32 msk = R((u16x2)(zero - __builtin_elementwise_min(dif, one)));
            Xv[h] = (X & msk) | (Xv[h] & ~msk);
            Yv[h] = (Y & msk) | (Yv[h] & ~msk);
            Mp[h] = (mp & msk) | (Mp[h] & ~msk);
          }
        }
      }
    }
    if (rowbad) {  // wave-uniform, two rows per frame
      asm volatile("" ::: "memory");  // keeps this one branch instead of per-row selects
      Xv[0] = Xv[1] = Yv[0] = Yv[1] = 0;
      Mp[0] = Mp[1] = 0;
      M4[0] = M4[1] = M4[2] = M4[3] = 0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      Xr[rn][h] = Xv[h];
      Yr[rn][h] = Yv[h];
      if (L2) {
        Mr[sN][1 + 2 * h] = M4[2 * h];
        Mr[sN][2 + 2 * h] = M4[2 * h + 1];
      } else {
        Mr[sN][1 + 2 * h] = Mp[h] & 0xFFFFu;
        Mr[sN][2 + 2 * h] = Mp[h] >> 16;
      }
    }
    Mr[sN][0] = from_lane_below(Mr[sN][4]);
    Mr[sN][5] = from_lane_above(Mr[sN][1]);

    const int c = k - 2;
    if (c >= r0 && c < rend) {
      u32 nib = 0;
      u64 cl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cl[q] = __ballot(Mr[sC][1 + q] > low);
      if ((cl[0] | cl[1] | cl[2] | cl[3]) != 0) {  // rows without a candidate skip direction + NMS
        u32 nibS = 0, nibC = 0;
        auto slot = [&](auto hc, auto ec, u32 aX, u32 aY, u32 X, u32 Y) {
          constexpr int h = decltype(hc)::value, e = decltype(ec)::value, q = 2 * h + e;
          u64 mS = 0, mC = 0;
          if (cl[q] != 0) {
            const u32 m = Mr[sC][1 + q];
            // tangent test on x = |dx|, y = |dy|: horizontal if y*2^15 < x*TG22, vertical if y*2^15 > x*(TG22 + 2^16)
            const int E = mad16<e, 0>(aY, k_m32768, mul16<e, 0>(aX, k_tg22));  // x*TG22 - y*2^15
            const int x16 = (int)(e ? (aX & 0xFFFF0000u) : (aX << 16));          // x * 2^16
            const u64 hz = __ballot(E > 0), vt = __ballot(E + x16 < 0);
            const u64 dneg = __ballot(mul16<e, e>(X, Y) < 0);  // sign(dx) != sign(dy) (both non-zero on a diagonal)
            const u64 kh = __ballot(m > Mr[sC][q]) & __ballot(m >= Mr[sC][2 + q]);       // left, right
            const u64 kv = __ballot(m > Mr[sU][1 + q]) & __ballot(m >= Mr[sN][1 + q]);   // up, down
            const u64 kp = __ballot(m > Mr[sU][q]) & __ballot(m > Mr[sN][2 + q]);        // s = +1: up-left, down-right
            const u64 kn = __ballot(m > Mr[sU][2 + q]) & __ballot(m > Mr[sN][q]);        // s = -1: up-right, down-left
            const u64 dg = ~hz & ~vt;
            const u64 keep = (hz & kh) | (~hz & vt & kv) | (dg & ~dneg & kp) | (dg & dneg & kn);
            mS = __ballot(m > high) & keep;
            mC = cl[q] & keep;
          }
          nibS = shift_in(nibS, mS);
          nibC = shift_in(nibC, mC);
        };
        auto pair = [&](auto hc) {
          constexpr int h = decltype(hc)::value;
          const u32 X = Xr[rp][h], Y = Yr[rp][h];
          const u32 aX = R(__builtin_elementwise_max(I(X), -I(X))), aY = R(__builtin_elementwise_max(I(Y), -I(Y)));
          slot(hc, std::integral_constant<int, 1>{}, aX, aY, X, Y);
          slot(hc, std::integral_constant<int, 0>{}, aX, aY, X, Y);
        };
        pair(std::integral_constant<int, 1>{});
        pair(std::integral_constant<int, 0>{});
        nib = (nibS | (nibC << 8)) & oknib;
      }
      if (p.prov_out && lane >= 1 && lane <= 62 && c0 < W) {  // provisional 0/255 map (strong bits); W % 4 == 0 here
        u32 po;
        asm("s_mul_i32 %0, %1, %2" : "=s"(po) : "s"(c), "s"(p.prov_pitch));
        u32 o = (u32)(strip * STRIP_W + 4 * (lane - 1));
        asm volatile("" : "+v"(o));
        *reinterpret_cast<u32 *>(p.prov_out + (size_t)frame * p.prov_fs + po + o) = nibble_to_bytes(nib & 0xFu);
      }
      const u32 w = nib | (from_lane_above(nib) << 4);
      if (store_lane) {
        u32 roff;
        asm("s_mul_i32 %0, %1, %2" : "=s"(roff) : "s"(c), "s"(plane_pitch));
        u32 so = st_off;
        asm volatile("" : "+v"(so));
        (splane + roff)[so] = (uint8_t)w;
        (cplane + roff)[so] = (uint8_t)(w >> 8);
      }
    }
  };

  // source rows r0-2 .. rend+1, six steps per loop trip (the ring period); a row is requested six steps before it is
  // used and its register refilled at once (unconditional loads: a step waits for the oldest of six, see k_nms)
  const int k0 = r0 - 2, kend = rend + 2;
  RawO bn[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) bn[j] = load_row(k0 + j);
  auto advance = [&](auto uc, int k) {
    constexpr int j = decltype(uc)::value;
    const RawO b = bn[j];
    bn[j] = load_row(k + 6);
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
}

// p.bgr != 0: interleaved 3-channel source (rows hold whole 12-byte groups of 4 pixels: pitch >= 3 * round_up(W, 4))
hipError_t launch_front_o(const FrontParams &p, hipStream_t s)
{
  if (p.chunk_rows < 1 || (unsigned long long)p.H * p.in_pitch >= (1ull << 32)) return hipErrorInvalidValue;
  if (p.in_pitch < (size_t)(p.bgr ? 3 : 1) * (((size_t)p.W + 3) / 4 * 4)) return hipErrorInvalidValue;
  const dim3 grid((p.total_items + 3) / 4), block(256);
  if (p.bgr) {
    if (p.l2gradient) hipLaunchKernelGGL((k_front_o<true, 3>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_front_o<false, 3>), grid, block, 0, s, p);
  } else {
    if (p.l2gradient) hipLaunchKernelGGL((k_front_o<true, 1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_front_o<false, 1>), grid, block, 0, s, p);
  }
  return hipGetLastError();
}

// =================================================================================================
// k_pack: tri-state u8 map (0 / 128 / 255) -> bit planes (entry of hc_hysteresis_device)
// =================================================================================================
__global__ __launch_bounds__(256) void k_pack(const PackParams p)
{
  const int lane = threadIdx.x & 63;
  const int segs = (p.W + 255) / 256;  // 256 px (64 lanes x 4) per wave
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)p.nframes * p.H * segs;
  if (wave >= total) return;
  const int seg = (int)(wave % segs);
  const int row = (int)((wave / segs) % p.H);
  const int frame = (int)(wave / ((long long)segs * p.H));
  const int c0 = seg * 256 + lane * 4;
  const uint8_t *rowp = p.in + (size_t)frame * p.in_frame_stride + (size_t)row * p.in_pitch;
  u32 nib = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 v = (c0 + k < p.W) ? rowp[c0 + k] : 0u;
    nib |= (v == 255u) ? (1u << k) : 0u;
    nib |= (v >= 128u) ? (0x100u << k) : 0u;
  }
  const u32 w = nib | (from_lane_above(nib) << 4);
  const size_t off = ((size_t)frame * p.H + row) * p.RD * 4 + (size_t)seg * 32 + (size_t)(lane >> 1);
  if (!(lane & 1) && (size_t)seg * 32 + (size_t)(lane >> 1) < (size_t)p.RD * 4) {
    reinterpret_cast<uint8_t *>(p.sbits)[off] = (uint8_t)w;
    reinterpret_cast<uint8_t *>(p.cbits)[off] = (uint8_t)(w >> 8);
  }
}

// =================================================================================================
// k_copy_rows: pitched device-to-device copy of n frames of `rows` rows of `row_bytes` bytes -- how caller buffers that the
// kernels cannot use in place (pointer / pitch / frame stride not a multiple of 4, rows that do not hold whole pixel
// groups) reach the context's internal pitched buffers and back.  16 bytes per lane, any alignment on either side (the
// hardware takes unaligned global dwordx4 accesses); the row-by-row DMA of hipMemcpy2DAsync moved 512 frames of
// 1918 x 1079 in 1.5 - 2.5 ms each way.
// =================================================================================================
typedef u32 u32x4_any __attribute__((ext_vector_type(4), aligned(1)));
__global__ __launch_bounds__(256) void k_copy_rows(uint8_t *dst, size_t dpitch, size_t dfs, const uint8_t *src, size_t spitch, size_t sfs, unsigned row_bytes, int rows)
{
  const unsigned x = (blockIdx.x * 256u + threadIdx.x) * 16u;
  const int row = blockIdx.y, f = blockIdx.z;
  if (x >= row_bytes || row >= rows) return;
  const uint8_t *q = src + (size_t)f * sfs + (size_t)row * spitch + x;
  uint8_t *d = dst + (size_t)f * dfs + (size_t)row * dpitch + x;
  if (x + 16u <= row_bytes) *reinterpret_cast<u32x4_any *>(d) = *reinterpret_cast<const u32x4_any *>(q);
  else
    for (unsigned k = 0; x + k < row_bytes; ++k) d[k] = q[k];
}

hipError_t launch_copy_rows(void *dst, size_t dpitch, size_t dfs, const void *src, size_t spitch, size_t sfs, size_t row_bytes, int rows, int n, hipStream_t s)
{
  if (row_bytes == 0 || rows <= 0 || n <= 0) return hipSuccess;
  if (row_bytes > 0x7FFFFFFFull || rows > 65535 || n > 65535) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((row_bytes + 4095) / 4096), (unsigned)rows, (unsigned)n), block(256);
  hipLaunchKernelGGL(k_copy_rows, grid, block, 0, s, (uint8_t *)dst, dpitch, dfs, (const uint8_t *)src, spitch, sfs, (unsigned)row_bytes, rows);
  return hipGetLastError();
}

hipError_t launch_pack(const PackParams &p, hipStream_t s)
{
  const long long total = (long long)p.nframes * p.H * ((p.W + 255) / 256);
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, p);
  return hipGetLastError();
}

// =================================================================================================
// k_hyst: edge hysteresis on the bit planes, by row sweeps with carry look-ahead
// =================================================================================================
// A candidate with a strong 8-neighbour becomes strong, to the fixpoint (cannyEdgeD.cu:342-352,
// launch loop cannyEdgeH.cu:307-324).  Work item = (frame, tile of tile_rows rows), one per wave;
// a whole bit-plane row lives in the wave (lane l holds dwords l*NW .. l*NW+NW-1).  The wave sweeps
// its rows downwards, then upwards: row r takes the strong bits of the previous row dilated by one
// column each way, ANDs with its candidates, and then FILLS every candidate run touched by a strong
// bit along the whole row at once: adding the seeds to the candidate word ripples a carry through
// each run ((c + s) ^ c marks the bits above the seed), lanes are chained by a carry look-ahead over
// the per-lane generate/propagate ballots (one 64-bit scalar add), and the bit-reversed pass fills
// the other direction.  One down+up pair settles every path that is monotone in the row index, so
// a tile converges in a few sweeps however long its chains are -- unlike pixel-per-iteration
// propagation (the reference moves one 30x30 tile per launch).  Tiles exchange boundary rows across
// launches; per-tile change flags let later launches touch only the tiles next to a change, and the
// flag word of the last queued launch tells the host whether the fixpoint was reached.
template <int NW>
struct RowBits {
  u32 w[NW];
};

template <int NW>
static __device__ __forceinline__ RowBits<NW> row_load(const u32 *plane_row, int lane, int RD)
{
  RowBits<NW> r;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int d = lane * NW + i;
    r.w[i] = d < RD ? plane_row[d] : 0u;
  }
  return r;
}

// fill every run of `c` that contains a bit of `s` (s subset of c), over the whole row
template <int NW>
static __device__ __forceinline__ RowBits<NW> row_fill(const RowBits<NW> &c, const RowBits<NW> &s)
{
  RowBits<NW> out;
  // towards higher columns
  {
    u32 t[NW];
    u32 carry = 0;
    bool allones = true;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const u64 x = (u64)c.w[i] + s.w[i] + carry;
      t[i] = (u32)x;
      carry = (u32)(x >> 32);
      allones = allones && (t[i] == 0xFFFFFFFFu);
    }
    const u64 G = __ballot(carry != 0), P = __ballot(allones);
    const u64 A = G | P;
    const u64 cin = (A + G) ^ A ^ G;  // carry into each lane (look-ahead by one scalar add)
    carry = __builtin_amdgcn_inverse_ballot_w64(cin) ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const u64 x = (u64)c.w[i] + s.w[i] + carry;
      carry = (u32)(x >> 32);
      out.w[i] = (((u32)x ^ c.w[i]) & c.w[i]) | s.w[i];
    }
  }
  // towards lower columns: the same on the bit-reversed row (lane order reversed in the look-ahead)
  {
    u32 rc[NW], rs[NW], t[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      rc[i] = __builtin_bitreverse32(c.w[NW - 1 - i]);
      rs[i] = __builtin_bitreverse32(s.w[NW - 1 - i]);
    }
    u32 carry = 0;
    bool allones = true;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const u64 x = (u64)rc[i] + rs[i] + carry;
      t[i] = (u32)x;
      carry = (u32)(x >> 32);
      allones = allones && (t[i] == 0xFFFFFFFFu);
    }
    const u64 G = __builtin_bitreverse64(__ballot(carry != 0)), P = __builtin_bitreverse64(__ballot(allones));
    const u64 A = G | P;
    const u64 cin = __builtin_bitreverse64((A + G) ^ A ^ G);
    carry = __builtin_amdgcn_inverse_ballot_w64(cin) ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const u64 x = (u64)rc[i] + rs[i] + carry;
      carry = (u32)(x >> 32);
      const u32 f = (((u32)x ^ rc[i]) & rc[i]);
      out.w[NW - 1 - i] |= __builtin_bitreverse32(f);
    }
  }
  return out;
}

// strong bits of the neighbouring row, dilated by one column each way
template <int NW>
static __device__ __forceinline__ RowBits<NW> row_dilate(const RowBits<NW> &p)
{
  RowBits<NW> d;
  const u32 below = from_lane_below(p.w[NW - 1]);  // previous lane's last dword
  const u32 above = from_lane_above(p.w[0]);       // next lane's first dword
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const u32 lo = i == 0 ? below : p.w[i - 1];
    const u32 hi = i == NW - 1 ? above : p.w[i + 1];
    d.w[i] = p.w[i] | __builtin_amdgcn_alignbit(p.w[i], lo, 31) | __builtin_amdgcn_alignbit(hi, p.w[i], 1);
  }
  return d;
}


// Workgroup tile = WAVES waves x TR rows.  Every wave keeps its TR rows of both planes in REGISTERS
// (lane l holds dwords l*NW.. of each row; rows are picked with a wave-uniform index, which the
// compiler turns into VGPR-indexed moves); LDS only carries the rows neighbours look at.
struct HystGeom { int nw, tr, waves; };
// 8 waves x 32 rows (256-row tiles) when the hysteresis has the chip to itself: fewer tile boundaries, fewer
// launches.  4 waves x 32 rows (one wave per SIMD) when it runs beside the next run's front kernels (pipelined mode):
// a 4-wave workgroup finds a place as soon as one wave slot per SIMD frees up, an 8-wave one has to wait for two --
// measured 1.7 ms against 4.2 ms for the hysteresis of 1024 frames under overlap.
// (beside k_front8, whose three workgroups fill a CU's LDS and registers, a hysteresis workgroup only finds room when a
// front workgroup retires: 2-wave workgroups fit the freed wave slots best -- 376 k frames/s against 368 k with 4 waves,
// 350 k with 8; one-wave workgroups need more launches than are queued for a 1080-row frame)
static inline HystGeom hyst_geom(bool beside_front) { return beside_front ? HystGeom{ 1, 32, 2 } : HystGeom{ 1, 32, 8 }; }
// frames_x_rows: frames x rows of the run.  geom: 0 = by the rules here; otherwise a shape picked by the caller for tuning experiments (encoded rows * 100 + waves:
// 3208, 3204, 3202, 1608, 3216 -- hc_create reads HC_HYST_GEOM once)
void hyst_tile_geometry(int geom, bool beside_front, long frames_x_rows, int H, int *tile_rows, int *waves)
{
  HystGeom g = hyst_geom(beside_front);
  // (Taller frames had taller tiles here -- 4 waves above 1200 rows, 8 above 2400 -- from the time when 16 launches were
  // queued per run.  With up to 48 launches queued and the tile height following the content (queue_hyst_expand), the small
  // 2-wave workgroups are as good at 4K (101 k frames/s either way) and better at 8K x 3 channels: 7.5 k against 6.6 k
  // frames/s -- an 8-wave workgroup needs two free wave slots on every SIMD of a CU at once, and launch 0 ran starved
  // beside the front kernel for as long as that took.)
  // a few frames only (the reference's one-frame-per-call pattern): the chip is nearly empty and the launches are pure
  // latency -- 8 waves x 16 rows per workgroup halve the rows a wave walks one after the other (measured on one 1080p
  // frame: hysteresis 0.122 ms against 0.139 ms with 8 x 32 and 0.130 ms with 4 x 32)
  if (frames_x_rows < 128 * 1024) g = HystGeom{ 1, 16, 8 };
  if (geom == 3208 || geom == 3204 || geom == 3202 || geom == 1608 || geom == 3216 || geom == 3201 || geom == 1604 || geom == 1602) g = HystGeom{ 1, geom / 100, geom % 100 };
  *tile_rows = g.tr;
  *waves = g.waves;
}

// PANELS: the frame is wider than one 2048-column panel (tiles then also have left / right neighbours); the common
// narrower case is compiled without that code.
// One workgroup tile, gtile = frame * tiles per frame + tile.  How launches >= 1 find the tiles with work (k_hyst below):
// MODE 0: every launch starts a workgroup per tile, and a tile looks at the reason word its neighbours left it in the
//         previous launch (p.wl_reason); launch index at run time (p.iter).
// MODE 1: launch 0 of the worklist scheme -- every tile, every row open; tiles that change a boundary append the
//         neighbours that look at it to the next launch's list.
// MODE 2: a later launch of the worklist scheme; the tile is on the list because a neighbour above / below / beside
//         changed the row or column it looks at (top / bot / side).
// MODE 3: as MODE 0, but the neighbours also go on the next launch's list: the launch between per-tile launches and list
//         launches of a run that starts with the former and ends with the latter.
template <int NW, int TR, int WAVES, bool PANELS, int MODE>
static __device__ __forceinline__ void hyst_tile(const HystParams &p, int gtile, bool top, bool bot, bool side)
{
  constexpr bool WORDS_IN = MODE == 0 || MODE == 3;  // the tile finds its reason in its word (a workgroup per tile); MODE 3 also writes lists: the launch before the first list launch
  const bool LATE = WORDS_IN ? p.iter > 0 : MODE == 2;
  static_assert(NW == 1, "frames wider than one panel are tiled in column panels; a lane holds one dword per row");
  static_assert((TR & (TR - 1)) == 0, "row indices are wrapped with TR - 1");
  constexpr int ROWW = 64 * NW;  // dwords per row
  constexpr int BR = WAVES * TR;
  constexpr int XQ_CAP = 256;  // a row adds up to 128 groups to a queue holding fewer than 64
  __shared__ u32 edge[(2 * WAVES + 2) * ROWW];  // per wave: first and last row of S; then the two halo rows
  __shared__ u32 bchg[24];
  __shared__ u32 xqueue[WAVES * XQ_CAP];
  const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // tiles are 2-D: row tile bt x column panel pn (a panel = ROWW dwords = 2048 columns; frames up to 2048
  // columns have one panel).  A wave always holds one dword per lane and row, whatever the frame width.
  const int NP = PANELS ? p.npanels : 1, ntile = p.nrtiles * NP;
  const int tile = gtile % ntile, frame = gtile / ntile;
  const int bt = tile / NP, pn = tile % NP;
  const int H = p.H, RD = p.RD;
  const int pcol = pn * ROWW;                          // first dword of this panel in a plane row
  const int b0 = bt * BR, nb = min(H, b0 + BR) - b0;  // rows of this workgroup tile
  if (WORDS_IN && LATE) {
    // work only if a neighbouring tile changed the row / column / corner this tile looks at: it left its reason in this
    // tile's word of the launch's parity (one load; cleared for the launch after next)
    u32 *reason = p.wl_reason + (size_t)(p.iter & 1) * p.wl_stride;
    const u32 why = (u32)__builtin_amdgcn_readfirstlane((int)reason[gtile]);
    top = (why & 1u) != 0; bot = (why & 2u) != 0; side = (why & 4u) != 0;
    if (why != 0) {
      __syncthreads();  // (uniform branch) everyone has the reason before it is cleared
      if (threadIdx.x == 0) {
        reason[gtile] = 0;
        if (PANELS) atomicAdd(&p.wl_count[p.iter], 1u);  // wide frames: how many tiles this launch visits (the host picks worklists or this form by it)
      }
    }
  }
  if (LATE) {
    if (!top && !bot && !side) return;  // uniform for the workgroup
    // A neighbour's boundary row changed somewhere -- but does a new bit reach a candidate of this tile?  Only
    // then can anything change here (the tile is at its own fixpoint).  Checked on the two boundary rows alone
    // (4 row loads) before the 2 x TR rows per wave are fetched: most tiles leave here in launches >= 1.
    if (!PANELS && !side) {
      u32 *S0 = p.sbits + (size_t)frame * H * RD;
      const u32 *C0 = p.cbits + (size_t)frame * H * RD;
      bool hit = false;
      auto reaches = [&](int halo_row, int my_row) {
        const RowBits<NW> hv = row_load<NW>(S0 + (size_t)halo_row * RD, lane, RD);
        const RowBits<NW> sv = row_load<NW>(S0 + (size_t)my_row * RD, lane, RD), cv = row_load<NW>(C0 + (size_t)my_row * RD, lane, RD);
        const RowBits<NW> d = row_dilate<NW>(hv);
        bool h = false;
#pragma unroll
        for (int j = 0; j < NW; ++j) h = h || ((cv.w[j] & d.w[j] & ~sv.w[j]) != 0);
        return __ballot(h) != 0;
      };
      if (top && wib == 0) hit = reaches(b0 - 1, b0);
      if (bot && wib == WAVES - 1) hit = reaches(b0 + nb, b0 + nb - 1) || hit;
      if (threadIdx.x == 0) bchg[20] = 0;
      __syncthreads();
      if (hit && lane == 0) atomicOr(&bchg[20], 1u);
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(bchg[20]) == 0) return;
    }
  }
  // this wave's rows inside the workgroup tile
  const int w0 = min(wib * TR, nb), n = min((wib + 1) * TR, nb) - w0;
  const bool owns_last = n > 0 && w0 + n == nb;
  const u64 all_rows = n >= 64 ? ~0ull : ((1ull << n) - 1);
  u64 dirty;  // bit r = row b0 + w0 + r needs (re)evaluation
  if (LATE) dirty = side ? all_rows : (((top && w0 == 0 && n > 0) ? 1ull : 0ull) | ((bot && owns_last) ? (1ull << (n - 1)) : 0ull));
  else dirty = all_rows;
  dirty = uniform64(dirty);
  u64 unfilled = p.first_pass ? all_rows : 0ull;  // rows not yet closed under the in-row fill

  u32 *S = p.sbits + (size_t)frame * H * RD;
  const u32 *C = p.cbits + (size_t)frame * H * RD;
  // the wave's rows: straight from HBM into registers, all loads in flight at once
  // (ext_vector types: the compiler keeps them in VGPRs and indexes them with s_set_gpr_idx;
  //  plain arrays with dynamic stores would be demoted to scratch)
  typedef u32 RowVec __attribute__((ext_vector_type(TR)));
  RowVec cr[NW], sr[NW];
  // (a panel is always 64 whole dwords wide -- launch_hyst: RD % 64 == 0 -- so only the row count limits the loads)
  if (n == TR) {  // all but the last wave tile of a frame: no per-row test
#pragma unroll
    for (int i = 0; i < TR; ++i) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const size_t off = (size_t)(b0 + w0 + i) * RD + pcol + lane * NW + j;
        cr[j][i] = C[off];
        sr[j][i] = S[off];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TR; ++i) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const bool ok = i < n;  // wave-uniform
        const size_t off = (size_t)(b0 + w0 + i) * RD + pcol + lane * NW + j;
        cr[j][i] = ok ? C[off] : 0u;
        sr[j][i] = ok ? S[off] : 0u;
      }
    }
  }
  u32 *my_first = edge + (2 * wib) * ROWW, *my_last = edge + (2 * wib + 1) * ROWW;
  u32 *halo_top = edge + (2 * WAVES) * ROWW, *halo_bot = edge + (2 * WAVES + 1) * ROWW;
  auto publish = [&](u32 *dst, int r) {
#pragma unroll
    for (int j = 0; j < NW; ++j) dst[lane * NW + j] = sr[j][r];
  };
  if (n > 0) {
    publish(my_first, 0);
    publish(my_last, n - 1);
  }
  if (wib == 0 || owns_last) {  // rows just outside the tile (owned by the neighbouring workgroups, or outside the frame)
    const int gr = wib == 0 ? b0 - 1 : b0 + nb;
    RowBits<NW> v;
#pragma unroll
    for (int j = 0; j < NW; ++j) v.w[j] = 0;
    if (wib == 0) {
      if (gr >= 0) v = row_load<NW>(S + (size_t)gr * RD + pcol, lane, RD - pcol);
#pragma unroll
      for (int j = 0; j < NW; ++j) halo_top[lane * NW + j] = v.w[j];
    }
    if (owns_last) {
      const int gb = b0 + nb;
      RowBits<NW> vb;
#pragma unroll
      for (int j = 0; j < NW; ++j) vb.w[j] = 0;
      if (gb < H) vb = row_load<NW>(S + (size_t)gb * RD + pcol, lane, RD - pcol);
#pragma unroll
      for (int j = 0; j < NW; ++j) halo_bot[lane * NW + j] = vb.w[j];
    }
  }
  // Column halos (frames wider than one panel): the strong bits just left / right of the panel, for this wave's
  // rows and the row above / below them -- bit k of the mask = row w0 - 1 + k.  Like the row halos they belong to
  // other workgroups and are as of the start of this launch.
  u64 lmask = 0, rmask = 0;
  if (PANELS && n > 0) {
    const int hr = b0 + w0 - 1 + lane;  // lanes 0 .. n+1 fetch one row each
    const bool rok = lane < n + 2 && hr >= 0 && hr < H;
    const u32 lv = (rok && pn > 0) ? S[(size_t)hr * RD + pcol - 1] : 0u;
    const u32 rv = (rok && pn + 1 < NP) ? S[(size_t)hr * RD + pcol + ROWW] : 0u;
    lmask = uniform64(__ballot((lv >> 31) != 0));
    rmask = uniform64(__ballot((rv & 1u) != 0));
  }
  if (threadIdx.x < 24) bchg[threadIdx.x] = 0;
  __syncthreads();
  const u32 *up_src = wib == 0 ? halo_top : edge + (2 * (wib - 1) + 1) * ROWW;  // row above my first row
  const u32 *dn_src = owns_last ? halo_bot : edge + (2 * (wib + 1)) * ROWW;      // row below my last row

  // Row worklist per wave, lowest dirty row first: a downward sweep that steps back up whenever a
  // row's new strong bits reach candidates of the row above.  Work is proportional to the rows that
  // change.  Waves exchange their boundary rows through LDS between rounds.
  u64 changed = 0;
  u32 colchg = 0;  // bit 0: first column of the panel changed, bit 1: last column
  // First launch, full wave tile: instead of visiting every row once to find out that most have nothing to do (a
  // worklist step costs ~55 instructions even then, more than half of them scalar), the rows that can change at all
  // are found first, all at once: a row is active iff one of its open candidates has a strong 8-neighbour -- in the
  // rows above / below as they are now, or in the row itself.  32 independent tests with compile-time row registers
  // (no s_set_gpr_idx, no dependency between them); every other row is only visited if a neighbour changes later.
  // (Two unrolled sequential sweeps with compile-time registers were tried instead: 63 inlined row updates are 76 KB of
  // code, the instruction cache misses made the hysteresis 1.9x slower.)
  if (NW == 1 && !LATE && n == TR) {
    u64 act = 0;
    auto test_row = [&](auto self, auto rc) {
      constexpr int r = decltype(rc)::value;
      RowBits<NW> nbr;
      const u32 upv = r == 0 ? up_src[lane] : sr[0][r > 0 ? r - 1 : 0];
      const u32 dnv = r == TR - 1 ? dn_src[lane] : sr[0][r < TR - 1 ? r + 1 : 0];
      const u32 sv = sr[0][r], cv = cr[0][r];
      nbr.w[0] = upv | dnv | sv;
      RowBits<NW> d = row_dilate<NW>(nbr);
      if (PANELS) {
        if (((lmask >> r) & 7ull) != 0 && lane == 0) d.w[0] |= 1u;
        if (((rmask >> r) & 7ull) != 0 && lane == 63) d.w[NW - 1] |= 0x80000000u;
      }
      if (__ballot((cv & ~sv & d.w[0]) != 0) != 0) act |= 1ull << r;
      if constexpr (r + 1 < TR) self(self, std::integral_constant<int, r + 1>{});
    };
    test_row(test_row, std::integral_constant<int, 0>{});
    dirty = uniform64(act);
    unfilled = dirty;  // (only) the active rows may still need their first in-row fill: in every other row no open candidate touches a strong bit of the row
  }
  for (int round = 0; round < 4096; ++round) {
    u64 round_changed = 0;
    while (dirty) {
      const int r = __builtin_ctzll(dirty);
      dirty &= dirty - 1;
      RowBits<NW> up, dn, s, c;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        // (index wrapped instead of clamped -- TR is a power of two -- the wrapped row is never the one used)
        up.w[j] = r == 0 ? up_src[lane * NW + j] : sr[j][(r - 1) & (TR - 1)];
        dn.w[j] = r == n - 1 ? dn_src[lane * NW + j] : sr[j][(r + 1) & (TR - 1)];
        s.w[j] = sr[j][r];
        c.w[j] = cr[j][r];
      }
      RowBits<NW> nbr;
#pragma unroll
      for (int j = 0; j < NW; ++j) nbr.w[j] = up.w[j] | dn.w[j];
      RowBits<NW> d = row_dilate<NW>(nbr);
      if (PANELS) {  // a strong pixel in the column next to the panel, rows r-1 .. r+1, touches my first / last column
        if (((lmask >> r) & 7ull) != 0 && lane == 0) d.w[0] |= 1u;
        if (((rmask >> r) & 7ull) != 0 && lane == 63) d.w[NW - 1] |= 0x80000000u;
      }
      RowBits<NW> seed;
      bool grew = false, hs = false, hc = false;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        seed.w[j] = s.w[j] | (c.w[j] & d.w[j]);
        grew = grew || (seed.w[j] != s.w[j]);
        hs = hs || s.w[j] != 0;
        hc = hc || (c.w[j] & ~s.w[j]) != 0;
      }
      bool todo = __ballot(grew) != 0;
      if ((unfilled >> r) & 1) {
        unfilled &= ~(1ull << r);
        todo = todo || (__ballot(hs) != 0 && __ballot(hc) != 0);
      }
      if (!todo) continue;
      const RowBits<NW> f = row_fill<NW>(c, seed);
      bool ch = false;
#pragma unroll
      for (int j = 0; j < NW; ++j) ch = ch || (f.w[j] != s.w[j]);
      if (__ballot(ch) == 0) continue;
      if (PANELS) {  // did the panel's first / last column change? (lane 0 bit 0, lane 63 bit 31)
        const u32 x0 = f.w[0] ^ s.w[0], x1 = f.w[NW - 1] ^ s.w[NW - 1];
        colchg |= (u32)(__builtin_amdgcn_readlane((int)x0, 0) & 1) | (((u32)__builtin_amdgcn_readlane((int)x1, 63) >> 31) << 1);
      }
#pragma unroll
      for (int j = 0; j < NW; ++j) sr[j][r] = f.w[j];
      if (r == 0) publish(my_first, 0);
      if (r == n - 1) publish(my_last, n - 1);
      round_changed |= 1ull << r;
      // The row below is looked at again in any case.  The row above only if a new bit reaches one of its open
      // candidates: it is at its own fixpoint for everything but this change (in a downward sweep nearly every step
      // used to be followed by a second look at the row above that found nothing).
      dirty |= ((1ull << r) << 1) & all_rows;
      if (r > 0) {
        RowBits<NW> nb;
#pragma unroll
        for (int j = 0; j < NW; ++j) nb.w[j] = f.w[j] & ~s.w[j];
        const RowBits<NW> reach = row_dilate<NW>(nb);
        bool hit = false;
#pragma unroll
        for (int j = 0; j < NW; ++j) hit = hit || (reach.w[j] & cr[j][r - 1] & ~sr[j][r - 1]) != 0;
        if (__ballot(hit) != 0) dirty |= (1ull << r) >> 1;
      }
    }
    changed |= round_changed;
    if (lane == 0) bchg[wib] = (n > 0 && (round_changed & 1ull) ? 1u : 0u) | (n > 0 && ((round_changed >> (n - 1)) & 1ull) ? 2u : 0u);
    __syncthreads();
    if (n > 0) {
      if (wib > 0 && (bchg[wib - 1] & 2u)) dirty |= 1ull;
      if (!owns_last && wib + 1 < WAVES && (bchg[wib + 1] & 1u)) dirty |= 1ull << (n - 1);
    }
    dirty = uniform64(dirty);
    // workgroup-wide "any wave has work": OR through an LDS word per round parity
    if (lane == 0 && dirty != 0) atomicOr(&bchg[18 + (round & 1)], 1u);
    __syncthreads();
    const bool more = __builtin_amdgcn_readfirstlane(bchg[18 + (round & 1)]) != 0;
    if (threadIdx.x == 0) bchg[18 + ((round + 1) & 1)] = 0;
    if (!more) break;
  }

  // Rows that changed go back to the plane, and the 0/255 map is written (removeCandidates + output copy,
  // cannyEdgeD.cu:379-395: strong bits -> 255, rest 0; 16 px per lane per store).  When the output already shows
  // the planes as they were in memory -- k_nms wrote the strong pixels (p.prov), or an earlier launch left it so --
  // only the 16-pixel groups whose bits changed are rewritten: the old dword of a row is read back (one row ahead)
  // and compared with the new one.  A changed row typically has two or three such groups out of 120, so they are not
  // expanded row by row (a few active lanes per instruction) but collected in a wave-private LDS queue -- one dword
  // per group: its 16 bits, row and position -- and expanded 64 at a time, a group per lane.
  {
    const bool patch = p.out && (LATE || p.prov);
    const bool a16 = (((uintptr_t)p.out | p.out_pitch | p.out_frame_stride) & 15u) == 0;
    uint8_t *obase = p.out ? p.out + (size_t)frame * p.out_frame_stride : nullptr;
    u32 *Sw = S + (size_t)(b0 + w0) * RD;  // the wave's first row (uniform); this lane's dword is at pcol + lane
    const u32 s_lane = (u32)(pcol + lane);
    // 16 pixels whose bits are `b`, starting at column c0 of row `row` of this frame
    auto put16 = [&](u32 b, int row, int c0) {
      if (c0 >= p.W) return;
      u32 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = nibble_to_bytes((b >> (4 * k)) & 0xFu);
      uint8_t *dst = obase + (size_t)row * p.out_pitch + c0;
      if (c0 + 15 < p.W) {
        if (a16) {
          // (the pointer goes through an empty asm: seeing two branches that store the same bytes, the optimizer otherwise
          //  merges them into the four dword stores -- twice the store instructions of the full-map launch)
          uint8_t *d16 = dst;
          asm volatile("" : "+v"(d16));
          *reinterpret_cast<uint4 *>(d16) = make_uint4(v[0], v[1], v[2], v[3]);
        } else
#pragma unroll
          for (int k = 0; k < 4; ++k) reinterpret_cast<u32 *>(dst)[k] = v[k];
      } else {
#pragma nounroll  // (unrolled, the 16 exec masks of this ragged last group cost the kernel an SGPR spill, i.e. a VGPR: 81 instead of 80)
        for (int k = 0; k < 16 && c0 + k < p.W; ++k) dst[k] = ((b >> k) & 1u) ? (uint8_t)255 : (uint8_t)0;
      }
    };
    u32 *xq = xqueue + wib * XQ_CAP;  // ring of changed groups: bits | row (6 bits) << 16 | group-of-the-panel-row << 22
    int xhead = 0, xcount = 0;
    auto xflush = [&](int nent) {
      wave_lds_sync();
      const u32 ent = xq[(xhead + lane) & (XQ_CAP - 1)];
      if (lane < nent) put16(ent & 0xFFFFu, b0 + w0 + (int)((ent >> 16) & 63u), pcol * 32 + (int)(ent >> 22) * 16);
      xhead = (xhead + nent) & (XQ_CAP - 1);
      xcount -= nent;
    };
    if (p.out && !patch) {
      // first launch on planes the output does not show yet: every row of the tile, whole rows, two passes of 64 groups
      for (int r = 0; r < n; ++r) {
        const u32 rowv = sr[0][r];
        if ((changed >> r) & 1ull) (Sw + (size_t)r * RD)[s_lane] = rowv;
        uint8_t *orow = obase + (size_t)(b0 + w0 + r) * p.out_pitch;
        for (int pass = 0; pass * 1024 < ROWW * 32 && pcol * 32 + pass * 1024 < p.W; ++pass) {
          // this lane writes px [32*pcol + 1024*pass + 16*lane, +16): half-word 64*pass + lane of the panel row
          const u32 x = __shfl(rowv, 32 * pass + (lane >> 1));  // before any lane drops out: the permute only sees active lanes
          const u32 b = (x >> (16 * (lane & 1))) & 0xFFFFu;
          const int c0 = pcol * 32 + pass * 1024 + lane * 16;
          if (c0 + 15 >= p.W) continue;  // whole 16-pixel groups here; the ragged last group of a row below
          u32 v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = nibble_to_bytes((b >> (4 * k)) & 0xFu);
          uint8_t *dst = orow + c0;
          if (a16) {
            uint8_t *d16 = dst;
            asm volatile("" : "+v"(d16));  // (keeps the 16-byte store: see put16)
            *reinterpret_cast<uint4 *>(d16) = make_uint4(v[0], v[1], v[2], v[3]);
          } else
#pragma unroll
            for (int k = 0; k < 4; ++k) reinterpret_cast<u32 *>(dst)[k] = v[k];
        }
      }
      // Widths that are not a multiple of 16: the last W % 16 pixels of every row, a byte per lane, in a loop of their own
      // (inside the loop above, the byte-wise tail -- unrolled 16 times under 16 exec masks -- cost the kernel an SGPR
      // spill, i.e. an 81st VGPR, and as a plain loop it made the compiler split the hot loop: 0.94 instead of 0.76 ms
      // for launch 0 of 1024 frames in plain mode)
      const int ragged = p.W & 15, cl = p.W - ragged;  // first column of the ragged group
      if (ragged != 0 && cl >= pcol * 32 && cl < (pcol + ROWW) * 32) {
        const int hw = (cl - pcol * 32) >> 4;  // its half-word in the panel row: dword hw / 2 is held by lane hw / 2
        for (int r = 0; r < n; ++r) {
          const u32 x = __shfl(sr[0][r], hw >> 1);
          const u32 b = (x >> (16 * (hw & 1))) & 0xFFFFu;
          if (lane < ragged) (obase + (size_t)(b0 + w0 + r) * p.out_pitch + cl)[lane] = ((b >> lane) & 1u) ? (uint8_t)255 : (uint8_t)0;
        }
      }
    } else {
      u64 m = changed;
      typedef u32 Old16 __attribute__((ext_vector_type(16)));
      for (;;) {
        if (xcount >= 64 || (m == 0 && xcount > 0)) {  // the one place where groups are expanded
          xflush(min(xcount, 64));
          continue;
        }
        if (m == 0) break;
        // What the plane (and with it the output) showed before this launch, for the next up to 16 changed rows: all
        // requests at once, one wait.  (Read back one row ahead inside the loop, each changed row waited for its own
        // load and -- the load being conditional -- for the stores before it as well.)
        Old16 oldv;
        if (patch) {
          u64 mb = m;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            u32 v = 0;
            if (mb) {
              v = (Sw + (size_t)__builtin_ctzll(mb) * RD)[s_lane];
              mb &= mb - 1;
            }
            oldv[j] = v;
          }
        }
        for (int j = 0; j < 16 && m != 0 && xcount < 64; ++j) {
          const int r = __builtin_ctzll(m);
          m &= m - 1;
          const u32 rowv = sr[0][r];
          (Sw + (size_t)r * RD)[s_lane] = rowv;
          if (!patch) continue;  // no output at all (hc_hysteresis_device on planes only)
          const u32 dv = oldv[j] ^ rowv;
          const bool clo = (dv & 0xFFFFu) != 0, chi = (dv >> 16) != 0;
          const u64 mlo = __ballot(clo), mhi = __ballot(chi);
          if ((mlo | mhi) == 0) continue;
          u32 nlo, nhi;
          asm("s_bcnt1_i32_b64 %0, %1" : "=s"(nlo) : "s"(mlo) : "scc");
          asm("s_bcnt1_i32_b64 %0, %1" : "=s"(nhi) : "s"(mhi) : "scc");
          const u32 base = (u32)(xhead + xcount), tag = (u32)r << 16;
          if (clo) xq[(base + mbcnt64(mlo)) & (XQ_CAP - 1)] = (rowv & 0xFFFFu) | tag | ((u32)(2 * lane) << 22);
          if (chi) xq[(base + nlo + mbcnt64(mhi)) & (XQ_CAP - 1)] = (rowv >> 16) | tag | ((u32)(2 * lane + 1) << 22);
          xcount += (int)(nlo + nhi);
        }
      }
    }
  }
  const bool first_changed = n > 0 && w0 == 0 && (changed & 1ull);
  const bool last_changed = owns_last && ((changed >> (n - 1)) & 1ull);
  if (lane == 0 && (first_changed || last_changed || colchg)) atomicOr(&bchg[16], (first_changed ? 1u : 0u) | (last_changed ? 2u : 0u) | (colchg << 2));
  __syncthreads();
  if (MODE == 0) {
    // the tiles that look at what changed get their reason (no list: every launch starts a workgroup per tile, and a tile
    // without a reason leaves after one load).  One lane per neighbour, as below.
    if (wib == 0) {
      const u32 vis = bchg[16];
      if (vis != 0) {
        if (lane == 0) atomicOr(&p.flags[p.iter], 1u);
        u32 *reason = p.wl_reason + (size_t)((p.iter + 1) & 1) * p.wl_stride;
        const int k = lane;
        const int t = k < 3 ? bt + 1 : k < 6 ? bt - 1 : bt;
        const int q = k < 6 ? pn + (k % 3) - 1 : (k == 6 ? pn - 1 : pn + 1);
        const u32 need = k < 3 ? 2u : k < 6 ? 1u : k == 6 ? 4u : 8u;
        const u32 why = k < 3 ? 1u : k < 6 ? 2u : 4u;
        if (k < 8 && (vis & need) != 0 && t >= 0 && t < p.nrtiles && q >= 0 && q < NP) atomicOr(&reason[frame * ntile + t * NP + q], why);
      }
    }
  } else if (wib == 0) {
    // The neighbours that look at what changed go on the next launch's worklist -- once each: the first reason to arrive
    // appends the tile, later ones only add their bit.  One lane per neighbour, so that the atomics' round trips overlap.
    const u32 vis = bchg[16];
    if (vis != 0) {
      if (lane == 0) atomicOr(&p.flags[p.iter], 1u);
      const int nxt = (p.iter + 1) & 1;
      u32 *reason = p.wl_reason + (size_t)nxt * p.wl_stride, *list = p.wl_list + (size_t)nxt * p.wl_stride;
      // lanes 0-2: the tiles below (my last row changed: their `top`), 3-5: above (my first row: their `bot`), 6 / 7: the
      // panel left / right (my first / last column: their `side`)
      const int k = lane;
      const int t = k < 3 ? bt + 1 : k < 6 ? bt - 1 : bt;
      const int q = k < 6 ? pn + (k % 3) - 1 : (k == 6 ? pn - 1 : pn + 1);
      const u32 need = k < 3 ? 2u : k < 6 ? 1u : k == 6 ? 4u : 8u;
      const u32 why = k < 3 ? 1u : k < 6 ? 2u : 4u;
      if (k < 8 && (vis & need) != 0 && t >= 0 && t < p.nrtiles && q >= 0 && q < NP) {
        const u32 g = (u32)(frame * ntile + t * NP + q);
        if (atomicOr(&reason[g], why) == 0) list[atomicAdd(&p.wl_count[p.iter + 1], 1u)] = g;
      }
    }
  }
  if (lane == 0 && p.stats && n > 0) {  // diagnostics (opt-in): changed rows summed / max over wave tiles, active wave tiles
    const u32 nch = (u32)__builtin_popcountll(changed);
    atomicAdd(&p.stats[0], nch);
    atomicMax(&p.stats[1], nch);
    atomicAdd(&p.stats[2], 1u);
  }
}

// MODE 0 / 3 (frames of one column panel, up to 2048 columns; dense wide frames -- their launches 0 to 2): a workgroup per
// tile; a tile whose neighbours left it no reason exits after one load.
// MODE 1 / 2 (wider frames): launch 0 as above; launch k > 0 takes its tiles from the worklist its predecessor wrote --
// the tiles whose neighbours changed a boundary row / column -- with a grid that is a fraction of the tile count
// (launch_hyst), one list entry per workgroup.  With panels a tile has eight neighbours, the flag test of MODE 0 is nine
// dependent byte loads, and the late launches that follow the few long edges of a frame through its tiles each started
// 17 k workgroups to find 1-2 % of them with work: on 4K and 8K streams most of the hysteresis chain's time, which is
// what their step follows (4K 100 -> 108 k frames/s, 8K x 3 channels 7.4 -> 8.3 k, 8K grey 15.8 -> 19.7 k).  At 1080p
// the flags are better: 388 k against 378 k frames/s -- there the step follows the front kernel, and a hysteresis that
// is spread thinly over it costs it less than the same work done in two thirds of the time (1.64 against 2.27 ms).
// No loop over list entries: around the tile code it costs registers (85-91 VGPRs; 148 and a stack as a real function),
// and the tile code must stay at 80 -- two hysteresis waves per 160-register hole a retiring front wave leaves.  Entries
// beyond the grid (dense or adversarial content: more than the grid's share of the tiles still active) are handed on to
// the next launch's list instead.
template <int NW, int TR, int WAVES, bool PANELS, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_hyst(const HystParams p)
{
  if ((MODE == 0 || MODE == 3) && p.iter > 0 && p.flags[p.iter - 1] == 0) return;  // previous launch changed no tile boundary: fixpoint reached
  // latency-bound kernel (a few waves walking dependent row steps): when it shares a SIMD with the next
  // run's front waves (pipelined mode) it should win the instruction arbitration
  __builtin_amdgcn_s_setprio(3);
  if constexpr (MODE != 2) {
    hyst_tile<NW, TR, WAVES, PANELS, MODE>(p, (int)blockIdx.x, false, false, false);
  } else {
    const u32 cnt = p.wl_count[p.iter];  // 0: the previous launch changed no tile boundary, the fixpoint is reached
    if (blockIdx.x >= cnt) return;
    const u32 *list = p.wl_list + (size_t)(p.iter & 1) * p.wl_stride;
    u32 *reason = p.wl_reason + (size_t)(p.iter & 1) * p.wl_stride;
    if (cnt > gridDim.x && threadIdx.x == 0 && blockIdx.x + gridDim.x < cnt) {
      const int nxt = (p.iter + 1) & 1;
      u32 *reason_n = p.wl_reason + (size_t)nxt * p.wl_stride, *list_n = p.wl_list + (size_t)nxt * p.wl_stride;
      for (u32 j = blockIdx.x + gridDim.x; j < cnt; j += gridDim.x) {
        const u32 g2 = list[j], w2 = reason[g2];
        reason[g2] = 0;
        if (atomicOr(&reason_n[g2], w2) == 0) list_n[atomicAdd(&p.wl_count[p.iter + 1], 1u)] = g2;
      }
      atomicOr(&p.flags[p.iter], 1u);  // work is left for another launch
    }
    const u32 g = (u32)__builtin_amdgcn_readfirstlane((int)list[blockIdx.x]);
    const u32 why = (u32)__builtin_amdgcn_readfirstlane((int)reason[g]);
    __syncthreads();  // everyone has the reason before it is cleared for launch k + 2
    if (threadIdx.x == 0) reason[g] = 0;
    hyst_tile<NW, TR, WAVES, PANELS, 2>(p, (int)g, (why & 1u) != 0, (why & 2u) != 0, (why & 4u) != 0);
  }
}

// ---- one launch for the whole hysteresis of a small run -------------------------------------------------------------
// A run of a few frames (the reference's one-frame-per-call pattern and the small pipelined batches) has at most a few
// dozen workgroup tiles, all resident at once, and its K dependent launches are K host calls and K trips through the
// command processor for kernels that mostly find nothing to do.  k_hyst_loop runs the same rounds -- hyst_tile in its
// workgroup-per-tile form, round `it` exactly what launch `it` would have been -- inside ONE launch, separated by a
// device-wide barrier: every workgroup releases its stores (the XCDs' L2s are not coherent with each other: agent-scope
// release = write-back, acquire = invalidate), arrives at a counter in device memory, waits for the others, acquires.
// The flag word of a round tells all workgroups alike whether another round is needed.
// Every wait is bounded: a workgroup that does not see the others arrive within ~4 ms (they are not resident -- another
// process fills the device) raises the abort word, marks the run as not converged and leaves; so do the others when
// they see it.  The host then continues the run with ordinary launches (finish_slot), as after any run whose queued
// launches were too few.  The grid never waits for a workgroup that cannot come.
static __device__ __forceinline__ bool grid_barrier(u32 *bar, u32 target)
{
  __shared__ u32 ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 good = 1;
    for (u32 spin = 0;; ++spin) {
      if (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      if (spin > 4000u || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    ok = good;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok != 0;
}

template <int TR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_hyst_loop(const HystParams p0, int rounds, u32 *bar)
{
  __builtin_amdgcn_s_setprio(3);
  HystParams p = p0;
  for (int it = 0; it < rounds; ++it) {
    p.iter = it;
    hyst_tile<1, TR, WAVES, false, 0>(p, (int)blockIdx.x, false, false, false);
    if (it + 1 == rounds) return;  // (the host reads this round's flag: set = not converged, finish_slot continues)
    if (!grid_barrier(bar, gridDim.x * (u32)(it + 1))) {
      if (threadIdx.x == 0) atomicOr(&p.flags[rounds - 1], 1u);
      return;
    }
    if (__hip_atomic_load(&p.flags[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // no tile boundary changed: the fixpoint (the same answer in every workgroup)
  }
}

// rounds <= MAX launches of the workgroup-per-tile form in one launch; the caller guarantees one column panel, at most
// HYST_LOOP_MAX_TILES tiles (resident together, with room for the loops of the other runs in flight) and bar[0..1] == 0
hipError_t launch_hyst_loop(const HystParams &p, int rounds, u32 *bar, hipStream_t s)
{
  const size_t tiles = (size_t)p.nframes * p.nrtiles;
  if (p.npanels != 1 || p.RD != 64 || tiles == 0 || tiles > (size_t)HYST_LOOP_MAX_TILES || rounds < 1 || !bar || !p.wl_reason || p.wl_stride < tiles) return hipErrorInvalidValue;
  if (p.tile_rows == 16 && p.waves == 8) hipLaunchKernelGGL((k_hyst_loop<16, 8>), dim3((unsigned)tiles), dim3(512), 0, s, p, rounds, bar);

  else if (p.tile_rows == 32 && p.waves == 2) hipLaunchKernelGGL((k_hyst_loop<32, 2>), dim3((unsigned)tiles), dim3(128), 0, s, p, rounds, bar);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_hyst(const HystParams &p, hipStream_t s)
{
  const HystGeom g = { 1, p.tile_rows, p.waves };
  if (p.RD > 256) return hipErrorInvalidValue;
  if (p.npanels != (p.RD + 63) / 64 || p.RD % 64) return hipErrorInvalidValue;
  const size_t tiles = (size_t)p.nframes * p.nrtiles * p.npanels;
  const bool wide = p.npanels > 1, late = p.iter > 0, lists = p.lists == 1;
  if (tiles > 0x7FFFFFFFull || !p.wl_reason || p.wl_stride < tiles) return hipErrorInvalidValue;
  if ((wide || p.lists != 0) && (!p.wl_count || !p.wl_list)) return hipErrorInvalidValue;
  // a workgroup per tile -- except the later launches of the worklist scheme: a workgroup per list entry.  Grid: the
  // caller's (p.late_grid, from the last run's list lengths), or a schedule that shrinks to an eighth of the tiles (at
  // least 2048 workgroups): on camera-like frames a third of the tiles are listed for launch 1, 1-2 % from launch 5 on;
  // entries beyond the grid wait for the next launch (k_hyst).
  size_t wgs = tiles;
  if (lists && late) {
    wgs = std::min(tiles, std::max<size_t>(2048, tiles >> std::min(std::max(p.iter - 2, 0), 3)));
    if (p.late_grid > 0) wgs = std::min(tiles, (size_t)p.late_grid);
  }
  const dim3 grid((unsigned)wgs), block(64 * g.waves);
#define HC_HYST_LAUNCH_P(TR_, WAVES_, PANELS_)                                                             \
  {                                                                                                         \
    if (p.lists == 2) hipLaunchKernelGGL((k_hyst<1, TR_, WAVES_, PANELS_, 3>), dim3((unsigned)tiles), block, 0, s, p); \
    else if (!lists) hipLaunchKernelGGL((k_hyst<1, TR_, WAVES_, PANELS_, 0>), grid, block, 0, s, p);        \
    else if (late) hipLaunchKernelGGL((k_hyst<1, TR_, WAVES_, PANELS_, 2>), grid, block, 0, s, p);          \
    else hipLaunchKernelGGL((k_hyst<1, TR_, WAVES_, PANELS_, 1>), grid, block, 0, s, p);                    \
  }
#define HC_HYST_LAUNCH(TR_, WAVES_)                 \
  {                                                  \
    if (wide) HC_HYST_LAUNCH_P(TR_, WAVES_, true)    \
    else HC_HYST_LAUNCH_P(TR_, WAVES_, false)        \
  }
  if (g.nw == 1 && g.tr == 32 && g.waves == 8) HC_HYST_LAUNCH(32, 8)
  else if (g.nw == 1 && g.tr == 32 && g.waves == 4) HC_HYST_LAUNCH(32, 4)
  else if (g.nw == 1 && g.tr == 32 && g.waves == 2) HC_HYST_LAUNCH(32, 2)
  else if (g.nw == 1 && g.tr == 16 && g.waves == 8) HC_HYST_LAUNCH(16, 8)
  else if (g.nw == 1 && g.tr == 32 && g.waves == 16) HC_HYST_LAUNCH(32, 16)
  else if (g.nw == 1 && g.tr == 32 && g.waves == 1) HC_HYST_LAUNCH(32, 1)
  else if (g.nw == 1 && g.tr == 16 && g.waves == 4) HC_HYST_LAUNCH(16, 4)
  else if (g.nw == 1 && g.tr == 16 && g.waves == 2) HC_HYST_LAUNCH(16, 2)

#undef HC_HYST_LAUNCH
#undef HC_HYST_LAUNCH_P
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// =================================================================================================
// Plain per-stage kernels: the MONO..THRESH taps of CannyEdge::run(finalStage) (cannyEdgeH.cu:58-117).
// One thread per pixel, literal arithmetic; not the fast path.
// =================================================================================================
#define PIX_PROLOG                                              \
  const int col = blockIdx.x * blockDim.x + threadIdx.x;        \
  const int row = blockIdx.y * blockDim.y + threadIdx.y;        \
  const int f = blockIdx.z;                                     \
  if (col >= W || row >= H) return;

__global__ void k_gray(const uint8_t *bgr, size_t bpitch, size_t bfs, uint8_t *mono, size_t mpitch, size_t mfs, int W, int H)
{
  PIX_PROLOG
  const uint8_t *q = bgr + f * bfs + row * bpitch + 3 * col;
  const int v = (q[0] * 7 + q[1] * 38 + q[2] * 19) >> 6;  // cannyEdgeD.cu:17-19,67
  mono[f * mfs + row * mpitch + col] = (uint8_t)min(255, v);
}

__global__ void k_gauss(const uint8_t *mono, size_t mpitch, size_t mfs, uint8_t *blur, size_t bpitch, size_t bfs, int W, int H)
{
  PIX_PROLOG
  blur[f * bfs + row * bpitch + col] = (uint8_t)gauss_chain_px(mono + f * mfs, mpitch, W, H, row, col);
}

__global__ void k_sobel(const uint8_t *blur, size_t bpitch, size_t bfs, int16_t *sx, int16_t *sy, size_t sp, size_t sfs, int W, int H)
{
  PIX_PROLOG
  const uint8_t *b = blur + f * bfs;
  auto at = [&](int r, int c) -> int { return (r >= 0 && r < H && c >= 0 && c < W) ? b[(size_t)r * bpitch + c] : 0; };
  const int x = -at(row - 1, col - 1) + at(row - 1, col + 1) - 2 * at(row, col - 1) + 2 * at(row, col + 1) - at(row + 1, col - 1) + at(row + 1, col + 1);
  const int y = (at(row - 1, col - 1) + 2 * at(row - 1, col) + at(row - 1, col + 1)) - (at(row + 1, col - 1) + 2 * at(row + 1, col) + at(row + 1, col + 1));
  sx[f * sfs + row * sp + col] = (int16_t)x;
  sy[f * sfs + row * sp + col] = (int16_t)y;
}

static __device__ __forceinline__ u32 isqrt_u32(u32 x)
{
  u32 r = (u32)__builtin_sqrtf((float)x);
  while ((u64)r * r > x) --r;
  while ((u64)(r + 1) * (r + 1) <= x) ++r;
  return r;
}

__global__ void k_graddisp(const int16_t *sx, const int16_t *sy, size_t sp, size_t sfs, uint8_t *out, size_t op, size_t ofs, int W, int H)
{
  PIX_PROLOG
  const int x = sx[f * sfs + row * sp + col], y = sy[f * sfs + row * sp + col];
  // float2uchar(min(|grad|,255)) with grad = 4*sqrtf((x/8)^2+(y/8)^2): trunc(grad) = isqrt((x^2+y^2)>>2)
  const u32 g = isqrt_u32((u32)(x * x + y * y) >> 2);
  out[f * ofs + row * op + col] = (uint8_t)min(g, 255u);
}

__global__ void k_nms_tap(const int16_t *sx, const int16_t *sy, size_t sp, size_t sfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int saturate)
{
  PIX_PROLOG
  const int16_t *X = sx + f * sfs, *Y = sy + f * sfs;
  auto S = [&](int r, int c) -> int {
    if (r < 0 || r >= H || c < 0 || c >= W) return 0;
    const int x = X[(size_t)r * sp + c], y = Y[(size_t)r * sp + c];
    return x * x + y * y;
  };
  const int x = X[(size_t)row * sp + col], y = Y[(size_t)row * sp + col];
  const int g = x * x + y * y;
  const int a = abs(x), b = abs(y);
  const int P = 2 * a * b, D = a * a - b * b;
  int bin;
  if (P < abs(D)) bin = D > 0 ? 2 : 0;
  else bin = ((x ^ y) < 0) ? 3 : 1;
  int q, r;
  if (bin == 0) { q = S(row + 1, col); r = S(row - 1, col); }
  else if (bin == 1) { q = S(row + 1, col - 1); r = S(row - 1, col + 1); }
  else if (bin == 2) { q = S(row, col + 1); r = S(row, col - 1); }
  else { q = S(row - 1, col - 1); r = S(row + 1, col + 1); }
  const bool keep = q <= g && r <= g;
  const u32 gt = isqrt_u32((u32)g >> 2);
  out[f * ofs + row * op + col] = keep ? (uint8_t)(saturate ? min(gt, 255u) : (gt & 0xFFu)) : 0;
}

__global__ void k_thresh(const uint8_t *nms, size_t np, size_t nfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int low, int high)
{
  PIX_PROLOG
  const int v = nms[f * nfs + row * np + col];
  out[f * ofs + row * op + col] = v > high ? 255 : v > low ? 128 : 0;
}

#define PIX_GRID dim3 blk(64, 4, 1), grd((W + 63) / 64, (H + 3) / 4, n)
hipError_t launch_gray(const uint8_t *bgr, size_t bpitch, size_t bfs, uint8_t *mono, size_t mpitch, size_t mfs, int W, int H, int n, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_gray, grd, blk, 0, s, bgr, bpitch, bfs, mono, mpitch, mfs, W, H); return hipGetLastError(); }
hipError_t launch_gauss(const uint8_t *mono, size_t mpitch, size_t mfs, uint8_t *blur, size_t bpitch, size_t bfs, int W, int H, int n, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_gauss, grd, blk, 0, s, mono, mpitch, mfs, blur, bpitch, bfs, W, H); return hipGetLastError(); }
hipError_t launch_sobel(const uint8_t *blur, size_t bpitch, size_t bfs, int16_t *sx, int16_t *sy, size_t sp, size_t sfs, int W, int H, int n, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_sobel, grd, blk, 0, s, blur, bpitch, bfs, sx, sy, sp, sfs, W, H); return hipGetLastError(); }
hipError_t launch_graddisp(const int16_t *sx, const int16_t *sy, size_t sp, size_t sfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int n, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_graddisp, grd, blk, 0, s, sx, sy, sp, sfs, out, op, ofs, W, H); return hipGetLastError(); }
hipError_t launch_nms(const int16_t *sx, const int16_t *sy, size_t sp, size_t sfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int n, int saturate, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_nms_tap, grd, blk, 0, s, sx, sy, sp, sfs, out, op, ofs, W, H, saturate); return hipGetLastError(); }
hipError_t launch_thresh(const uint8_t *nms, size_t np, size_t nfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int n, int low, int high, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_thresh, grd, blk, 0, s, nms, np, nfs, out, op, ofs, W, H, low, high); return hipGetLastError(); }

}  // namespace hc
