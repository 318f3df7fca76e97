// canny_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hipcanny hot path.
//
// What the reference does in 9+k launches over 25 B/px of intermediates (src/cvp/cannyEdgeD.cu,
// launch sites src/cvp/cannyEdgeH.cu:214-338) is done here in
//   k_front   blur + Sobel + magnitude + direction + NMS + double threshold, one pass, output = 2 bit planes
//   k_hyst    edge hysteresis on the bit planes (64 px per 64-bit op), device-side convergence flag
//   k_expand  bit plane -> 0/255 u8 edge map
// MFMA is deliberately not used: there is no dense contraction (an f32 MFMA would reproduce the
// Gaussian's fmaf chain bit for bit, but as a banded 36x32 Toeplitz product it wastes 31/36 of its
// multiplies and runs at the f32 vector rate -- 4-7x slower than the packed integer form below).
//
// Numerical contract ("Mode R"): identical to the reference kernels, including the
// float Gaussian chain (via the exact shortcut explained at gauss_row), the u8 wrap of gradients
// >= 256 and the non-strict NMS.
#include "canny_common.h"

namespace hc {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__constant__ float c_gk[25];  // reference: __constant__ float GK[5][5], cannyEdgeD.cu:11

hipError_t upload_gauss_coeffs(const float gk[25]) { return hipMemcpyToSymbol(HIP_SYMBOL(c_gk), gk, 25 * sizeof(float)); }

// ---- cross-lane and packed helpers -------------------------------------------------------------
// value held by lane-1 (0 in lane 0) / lane+1 (0 in lane 63): DPP wave shifts, no LDS involved
static __device__ __forceinline__ u32 from_lane_below(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); }
static __device__ __forceinline__ u32 from_lane_above(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }
// bytes of x -> two u16 pairs
static __device__ __forceinline__ u32 unpack_lo(u32 x) { return __builtin_amdgcn_perm(0u, x, 0x0c010c00u); }  // (b0, b1)
static __device__ __forceinline__ u32 unpack_hi(u32 x) { return __builtin_amdgcn_perm(0u, x, 0x0c030c02u); }  // (b2, b3)
// (hi16 of lo_src, lo16 of hi_src): the pair shifted by one pixel
static __device__ __forceinline__ u32 pair_shift(u32 hi_src, u32 lo_src) { return __builtin_amdgcn_alignbyte(hi_src, lo_src, 2); }
static __device__ __forceinline__ u16x2 U(u32 v) { return __builtin_bit_cast(u16x2, v); }
static __device__ __forceinline__ i16x2 I(u32 v) { return __builtin_bit_cast(i16x2, v); }
static __device__ __forceinline__ u32 R(u16x2 v) { return __builtin_bit_cast(u32, v); }
static __device__ __forceinline__ u32 R(i16x2 v) { return __builtin_bit_cast(u32, v); }
static __device__ __forceinline__ void wave_lds_sync()
{
  // wave-private LDS hand-off between lanes of ONE wave: DS ops of a wave execute in order, this
  // only stops the compiler from moving accesses across the point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
static __device__ __forceinline__ u32 mbcnt64(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }

// XCD-aware work-item order: blocks are dealt round-robin over the 8 XCDs, so block b and b+8 share
// an L2.  Give each XCD a contiguous range of work items (neighbouring chunks share halo rows).
static __device__ __forceinline__ int xcd_remap(int bid, int nblocks)
{
  const int per = nblocks >> 3, rem = nblocks & 7;
  const int x = bid & 7, k = bid >> 3;
  // XCD x owns per + (x < rem) blocks; its range starts after those of XCDs 0..x-1
  return x * per + (x < rem ? x : rem) + k;
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
  if (bad) atomicOr(res, bad);
}

hipError_t launch_selftest(u32 *d_result, hipStream_t s)
{
  hipLaunchKernelGGL(k_selftest, dim3(2), dim3(128), 0, s, d_result);
  return hipGetLastError();
}

// =================================================================================================
// k_front
// =================================================================================================
// Work item = (frame, strip, chunk of CHUNK output rows), one per wave, 4 independent waves per
// workgroup (no workgroup barrier anywhere).  Two phases per item, both marching down the rows with
// every intermediate in registers and horizontal neighbours fetched from the adjacent lane by DPP:
//   phase 1  input rows -> blur rows (u8) into a wave-private LDS slab of CHUNK+4 rows
//   fix-up   the few pixels whose exact float result cannot be decided by integers (see gauss_row)
//   phase 2  blur rows -> Sobel -> S = sumX^2+sumY^2 -> direction -> NMS -> thresholds -> bit planes
constexpr int QCAP = 1024;  // fix-up queue entries (u16) per wave

template <int CHUNK>
struct FrontLds {
  static constexpr int BROWS = CHUNK + 4;
  static constexpr int WAVE_BYTES = BROWS * 256 + QCAP * 2;
};

size_t front_lds_bytes(int chunk_rows) { return (size_t)4 * ((chunk_rows + 4) * 256 + QCAP * 2); }

// literal reference chain for one pixel (cannyEdgeD.cu:102-115): 25 fused multiply-adds from 0.0f in
// r-major / c-minor order, truncation.  Only used for the rare undecidable pixels.
static __device__ __forceinline__ u32 gauss_chain_px(const uint8_t *frame, size_t pitch, int W, int H, int row, int col)
{
  float f = 0.0f;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int rr = row - 2 + r;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int cc = col - 2 + c;
      float px = 0.0f;
      if (rr >= 0 && rr < H && cc >= 0 && cc < W) px = (float)frame[(size_t)rr * pitch + cc];
      f = __builtin_fmaf(c_gk[r * 5 + c], px, f);
    }
  }
  return (u32)(int)f;
}

template <int CHUNK>
__global__ __launch_bounds__(256) void k_front(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  unsigned char *blur_s = smem + wib * FrontLds<CHUNK>::WAVE_BYTES;
  unsigned short *queue = reinterpret_cast<unsigned short *>(blur_s + FrontLds<CHUNK>::BROWS * 256);

  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items) return;
  const int chunk = item % p.nchunks;
  const int strip = (item / p.nchunks) % p.nstrips;
  const int frame = item / (p.nchunks * p.nstrips);
  const int W = p.W, H = p.H;
  const int r0 = chunk * CHUNK;
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;

  // per-lane column validity: byte mask for packed u8 rows, dword masks for the 4 S values
  u32 cmask = 0;
  u32 vm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (c0 + k >= 0) && (c0 + k < W);
    vm[k] = in ? 0xFFFFFFFFu : 0u;
    cmask |= in ? (0xFFu << (8 * k)) : 0u;
  }
  const bool col_any = cmask != 0;
  const uint8_t *frame_base = p.in + (size_t)frame * p.in_frame_stride;
  const uint8_t *src = frame_base + c0;  // dereferenced only where col_any

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
  int qn = 0;  // wave-uniform queue fill

  auto load_row = [&](int row) -> u32 {
    u32 v = 0;
    if (row >= 0 && row < H && col_any) v = *reinterpret_cast<const u32 *>(src + (size_t)row * p.in_pitch);
    return v & cmask;
  };

  auto fixup = [&]() {
    wave_lds_sync();
    for (int base = 0; base < qn; base += 64) {
      const int e = base + lane;
      if (e < qn) {
        const u32 a = queue[e];
        const int row = r0 - 2 + (int)(a >> 8);
        const int col = strip * STRIP_W - STRIP_HALO + (int)(a & 255u);
        blur_s[a] = (unsigned char)gauss_chain_px(frame_base, p.in_pitch, W, H, row, col);
      }
    }
    wave_lds_sync();
    qn = 0;
  };

  u32 xn1 = load_row(r0 - 4), xn2 = load_row(r0 - 3);
  for (int jr = r0 - 4; jr < r0 + CHUNK + 4; ++jr) {
    const u32 x = xn1;
    xn1 = xn2;
    xn2 = load_row(jr + 2);
    const u32 A = unpack_lo(x), B = unpack_hi(x);
    const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
    const u32 m1 = pair_shift(A, Bl);  // (x-1, x0)
    const u32 p1 = pair_shift(B, A);   // (x1, x2)
    const u32 p3 = pair_shift(Ar, B);  // (x3, x4)
    u32 Sp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u16x2 P = h == 0 ? U(Bl) + U(B) : U(A) + U(Ar);
      const u16x2 Q = h == 0 ? U(m1) + U(p1) : U(p1) + U(p3);
      const u16x2 Cc = h == 0 ? U(A) : U(B);
      const u16x2 two = { 2, 2 }, five = { 5, 5 };
      const u16x2 e = Q * two + P;
      const u16x2 h0 = e * two + Cc * five;
      const u16x2 w = Cc * two + Q;
      const u16x2 h1 = h0 * two + w;
      const u16x2 h2 = (h0 + h1) - (P + w);
      Sp[h] = R(U(a4[h]) + h0);
      a4[h] = R(U(a3[h]) + h1);
      a3[h] = R(U(a2[h]) + h2);
      a2[h] = R(U(a1[h]) + h1);
      a1[h] = R(h0);
    }
    const int rb = jr - 2;  // blur row completed by this step
    if (rb >= r0 - 2) {     // wave-uniform
      const int slot = rb - (r0 - 2);
      u32 bl = 0;
      if (rb >= 0 && rb < H) {
        // n = floor(S/159) = (S*52759) >> 23, exact for S <= 40545 (tests/test_oracle_exhaustive.py)
        const u16x2 mlo = { 52759, 0 }, mhi = { 0, 52759 };
        const u32 n0 = __builtin_amdgcn_udot2(U(Sp[0]), mlo, 0u, false) >> 23;
        const u32 n1 = __builtin_amdgcn_udot2(U(Sp[0]), mhi, 0u, false) >> 23;
        const u32 n2 = __builtin_amdgcn_udot2(U(Sp[1]), mlo, 0u, false) >> 23;
        const u32 n3 = __builtin_amdgcn_udot2(U(Sp[1]), mhi, 0u, false) >> 23;
        const u32 B0 = n0 | (n1 << 16), B1 = n2 | (n3 << 16);
        const u16x2 c159 = { 159, 159 };
        const u16x2 rem0 = U(Sp[0]) - U(B0) * c159, rem1 = U(Sp[1]) - U(B1) * c159;
        bl = __builtin_amdgcn_perm(B1, B0, 0x06040200u) & cmask;
        // queue the undecidable pixels (remainder 0, inside the image)
        const bool f0 = rem0.x == 0 && vm[0], f1 = rem0.y == 0 && vm[1], f2 = rem1.x == 0 && vm[2], f3 = rem1.y == 0 && vm[3];
        const u64 m0 = __ballot(f0), mm1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
        if (m0 | mm1 | m2 | m3) {
          if (qn + 256 > QCAP) {
            // flush: earlier rows of the slab are final, the current row is not written yet
            fixup();
          }
          const u32 abase = (u32)slot * 256u + (u32)lane * 4u;
          if (m0) { if (f0) queue[qn + mbcnt64(m0)] = (unsigned short)(abase + 0); qn += __builtin_popcountll(m0); }
          if (mm1) { if (f1) queue[qn + mbcnt64(mm1)] = (unsigned short)(abase + 1); qn += __builtin_popcountll(mm1); }
          if (m2) { if (f2) queue[qn + mbcnt64(m2)] = (unsigned short)(abase + 2); qn += __builtin_popcountll(m2); }
          if (m3) { if (f3) queue[qn + mbcnt64(m3)] = (unsigned short)(abase + 3); qn += __builtin_popcountll(m3); }
        }
      }
      reinterpret_cast<u32 *>(blur_s)[slot * 64 + lane] = bl;
    }
  }
  fixup();

  // ------------------------------------------------------------------ phase 2: blur -> bit planes
  // Sobel is separable: per blur row d = b[+1]-b[-1], s = b[-1]+2b[0]+b[+1] (packed i16 pairs), then
  // sumX(i) = d[i-1]+2d[i]+d[i+1], sumY(i) = s[i-1]-s[i+1] (cannyEdgeD.cu:158-167).
  // S = sumX^2+sumY^2 by one v_dot2 per pixel; comparisons of the reference's float gradient are
  // comparisons of S (strictly monotone, tests).  Direction bins (cannyEdgeD.cu:239-264) exactly:
  // with X2=sumX^2, Q=sumX*sumY, D=X2-(S-X2):  |2Q| < |D| -> axis bin (D>0: 2 horizontal, D<0: 0
  // vertical) else diagonal (Q<0: bin 3, else bin 1).
  u32 dA[2] = { 0, 0 }, dB[2] = { 0, 0 }, sA[2] = { 0, 0 }, sB[2] = { 0, 0 };  // (k-2), (k-1) rows of d and s
  u32 Su[6], Sc[6], Sd[6];   // S rows: [0]=left neighbour, [1..4]=own 4 px, [5]=right neighbour
  u32 Vc[4], Vd[4];          // packed (sumX,sumY) of the centre / newest row
#pragma unroll
  for (int k = 0; k < 6; ++k) Su[k] = Sc[k] = Sd[k] = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) Vc[k] = Vd[k] = 0;

  // lanes / pixel slots that may be written: valid lane, column inside the image
  u64 okm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) okm[k] = __ballot(vm[k] != 0) & BM_VALID;
  u64 *bm_strip = p.bm + ((size_t)(frame * p.nstrips + strip) * H) * BM_WORDS;

  for (int k = r0 - 2; k < r0 + CHUNK + 2; ++k) {  // k = blur row arriving
    const u32 b = reinterpret_cast<const u32 *>(blur_s)[(k - (r0 - 2)) * 64 + lane];
    const u32 A = unpack_lo(b), B = unpack_hi(b);
    const u32 Bl = from_lane_below(B), Ar = from_lane_above(A);
    const u32 m1 = pair_shift(A, Bl), p1 = pair_shift(B, A), p3 = pair_shift(Ar, B);
    const i16x2 two = { 2, 2 };
    u32 dk[2], sk[2];
    dk[0] = R(I(p1) - I(m1));
    sk[0] = R(I(A) * two + (I(m1) + I(p1)));
    dk[1] = R(I(p3) - I(p1));
    sk[1] = R(I(B) * two + (I(p1) + I(p3)));
    // Sobel row i = k-1
    const int i = k - 1;
#pragma unroll
    for (int q = 0; q < 6; ++q) { Su[q] = Sc[q]; Sc[q] = Sd[q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) Vc[q] = Vd[q];
    if (i >= 0 && i < H) {  // wave-uniform
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32 X = R(I(dA[h]) + I(dk[h]) + I(dB[h]) * two);
        const u32 Y = R(I(sA[h]) - I(sk[h]));
        Vd[2 * h + 0] = __builtin_amdgcn_perm(Y, X, 0x05040100u);  // (sumX, sumY) of pixel 2h
        Vd[2 * h + 1] = __builtin_amdgcn_perm(Y, X, 0x07060302u);  // pixel 2h+1
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) Sd[1 + q] = (u32)__builtin_amdgcn_sdot2(I(Vd[q]), I(Vd[q]), 0, false) & vm[q];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { Sd[1 + q] = 0; Vd[q] = 0; }
    }
    Sd[0] = from_lane_below(Sd[4]);
    Sd[5] = from_lane_above(Sd[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) { dA[h] = dB[h]; dB[h] = dk[h]; sA[h] = sB[h]; sB[h] = sk[h]; }

    // NMS + thresholds for row c = k-2
    const int c = k - 2;
    if (c >= r0 && c < H) {  // wave-uniform (c < r0 + CHUNK by the loop bound)
      u64 strong[4], cand[4];
      const u32 mx = max(max(Sc[1], Sc[2]), max(Sc[3], Sc[4]));
      const bool wrap = __ballot(mx >= p.wrap_limit) != 0;  // some gradient >= 256: u8 wrap bands needed
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32 g = Sc[1 + q];
        const int sx = (int)(short)(Vc[q] & 0xFFFFu), sy = (int)(short)(Vc[q] >> 16);
        const int X2 = sx * sx, Q2 = 2 * sx * sy;
        const int D = 2 * X2 - (int)g;
        const u64 e1p = __ballot(D - Q2 > 0), e2p = __ballot(D + Q2 > 0);
        const u64 e1n = __ballot(D - Q2 < 0), e2n = __ballot(D + Q2 < 0);
        const u64 b2 = e1p & e2p, b0 = e1n & e2n;
        const u64 dg = ~(b0 | b2);
        const u64 qneg = __ballot(Q2 < 0);
        const u64 b3 = dg & qneg, b1 = dg & ~qneg;
        // neighbours (cannyEdgeD.cu:245-264): bin0 down/up, bin1 down-left/up-right, bin2 right/left, bin3 up-left/down-right
        const u64 k0 = __ballot(max(Sd[1 + q], Su[1 + q]) <= g);
        const u64 k1 = __ballot(max(Sd[q], Su[2 + q]) <= g);
        const u64 k2 = __ballot(max(Sc[2 + q], Sc[q]) <= g);
        const u64 k3 = __ballot(max(Su[q], Sd[2 + q]) <= g);
        const u64 keep = (b0 & k0) | (b1 & k1) | (b2 & k2) | (b3 & k3);
        u64 cl, st;
        if (!wrap) {
          cl = __ballot(g >= p.a_lo[0]);
          st = __ballot(g >= p.a_hi[0]);
        } else {
          const u64 w0 = __ballot(g >= 262144u), w1 = __ballot(g >= 1048576u);
          cl = (__ballot(g >= p.a_lo[0]) & ~w0) | (__ballot(g >= p.a_lo[1]) & ~w1) | __ballot(g >= p.a_lo[2]);
          st = (__ballot(g >= p.a_hi[0]) & ~w0) | (__ballot(g >= p.a_hi[1]) & ~w1) | __ballot(g >= p.a_hi[2]);
        }
        cand[q] = cl & keep & okm[q];
        strong[q] = st & keep & okm[q];
      }
      u64 wv = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        wv = lane == q ? strong[q] : wv;
        wv = lane == 4 + q ? cand[q] : wv;
      }
      if (lane < BM_WORDS) bm_strip[(size_t)c * BM_WORDS + lane] = wv;
    }
  }
}

hipError_t launch_front(const FrontParams &p, int chunk_rows, hipStream_t s)
{
  const int nblocks = (p.total_items + 3) / 4;
  const size_t lds = front_lds_bytes(chunk_rows);
  static bool attr_done = false;
  if (!attr_done) {  // chunk 64 needs more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_front<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_lds_bytes(64));
    attr_done = true;
  }
  switch (chunk_rows) {
    case 16: hipLaunchKernelGGL(k_front<16>, dim3(nblocks), dim3(256), lds, s, p); break;
    case 32: hipLaunchKernelGGL(k_front<32>, dim3(nblocks), dim3(256), lds, s, p); break;
    case 64: hipLaunchKernelGGL(k_front<64>, dim3(nblocks), dim3(256), lds, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// =================================================================================================
// k_pack: tri-state u8 map (0 / 128 / 255) -> bit planes (entry of hc_hysteresis_device)
// =================================================================================================
__global__ __launch_bounds__(256) void k_pack(const PackParams p)
{
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)p.nframes * p.nstrips * p.H;
  if (wave >= total) return;
  const int row = (int)(wave % p.H);
  const int strip = (int)((wave / p.H) % p.nstrips);
  const int frame = (int)(wave / ((long long)p.H * p.nstrips));
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;
  const uint8_t *rowp = p.in + (size_t)frame * p.in_frame_stride + (size_t)row * p.in_pitch;
  u64 wv = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = c0 + k;
    const bool in = col >= 0 && col < p.W && lane >= 1 && lane <= 62;
    const u32 v = in ? rowp[col] : 0u;
    const u64 st = __ballot(v == 255u), cd = __ballot(v >= 128u);
    wv = lane == k ? st : wv;
    wv = lane == 4 + k ? cd : wv;
  }
  if (lane < BM_WORDS) p.bm[(((size_t)frame * p.nstrips + strip) * p.H + row) * BM_WORDS + lane] = wv;
}

hipError_t launch_pack(const PackParams &p, hipStream_t s)
{
  const long long total = (long long)p.nframes * p.nstrips * p.H;
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, p);
  return hipGetLastError();
}

// =================================================================================================
// k_hyst: hysteresis on the bit planes
// =================================================================================================
// Workgroup = (frame, strip, row tile); lane = one row, holding its 4 strong + 4 candidate words in
// registers.  A candidate with a strong 8-neighbour becomes strong (cannyEdgeD.cu:342-352); here
// 64 columns per 64-bit op.  Column neighbours in the 4-way interleaved plane layout: slot j-1 / j+1,
// across the lane boundary a 1-bit shift of slot 3 / slot 0.  Row neighbours: the adjacent lane
// (wave edges through LDS).  The tile iterates to its local fixpoint; cross-tile propagation happens
// over successive launches, gated by a device-side flag (no host round trip, unlike
// cannyEdgeH.cu:307-324).  The fixpoint is unique (monotone updates), so tiling cannot change it.
static __device__ __forceinline__ u64 load_strong(const u64 *bm, int H, int nstrips, int frame, int strip, int row, int j)
{
  // own word plus the two halo bits taken from the neighbouring strips (bit 0 <- their bit 62, bit 63 <- their bit 1)
  if (row < 0 || row >= H) return 0;
  const size_t base = (((size_t)frame * nstrips + strip) * H + row) * BM_WORDS + j;
  u64 v = bm[base] & BM_VALID;
  if (strip > 0) v |= (bm[base - (size_t)H * BM_WORDS] >> 62) & 1ull;
  if (strip + 1 < nstrips) v |= ((bm[base + (size_t)H * BM_WORDS] >> 1) & 1ull) << 63;
  return v;
}

__global__ __launch_bounds__(1024) void k_hyst(const HystParams p)
{
  if (p.iter > 0 && p.flags[p.iter - 1] == 0) return;  // previous launch changed nothing visible: fixpoint reached
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *edge = reinterpret_cast<u64 *>(smem);  // [wave][2][4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int rt = blockIdx.x % p.nrtiles;
  const int strip = (blockIdx.x / p.nrtiles) % p.nstrips;
  const int frame = blockIdx.x / (p.nrtiles * p.nstrips);
  const int H = p.H;
  const int row = rt * p.tile_rows + (int)threadIdx.x;
  const bool rv = row < H && (int)threadIdx.x < p.tile_rows;

  u64 S[4], C[4], S0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    S[j] = rv ? load_strong(p.bm, H, p.nstrips, frame, strip, row, j) : 0;
    C[j] = rv ? (p.bm[(((size_t)frame * p.nstrips + strip) * H + row) * BM_WORDS + 4 + j] & BM_VALID) : 0;
    S0[j] = S[j];
  }
  // rows just outside the tile (owned by other tiles; constant during this launch)
  u64 halo_top[4] = { 0, 0, 0, 0 }, halo_bot[4] = { 0, 0, 0, 0 };
  if (threadIdx.x == 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) halo_top[j] = load_strong(p.bm, H, p.nstrips, frame, strip, rt * p.tile_rows - 1, j);
  if ((int)threadIdx.x == p.tile_rows - 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) halo_bot[j] = load_strong(p.bm, H, p.nstrips, frame, strip, rt * p.tile_rows + p.tile_rows, j);

  for (int it = 0; it < 100000; ++it) {
    if (lane == 0)
#pragma unroll
      for (int j = 0; j < 4; ++j) edge[(wave * 2 + 0) * 4 + j] = S[j];
    if (lane == 63)
#pragma unroll
      for (int j = 0; j < 4; ++j) edge[(wave * 2 + 1) * 4 + j] = S[j];
    __syncthreads();
    u64 N[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u64 up = __shfl_up(S[j], 1), dn = __shfl_down(S[j], 1);
      if (lane == 0) up = wave > 0 ? edge[((wave - 1) * 2 + 1) * 4 + j] : halo_top[j];
      if (lane == 63) dn = wave + 1 < nw ? edge[((wave + 1) * 2 + 0) * 4 + j] : halo_bot[j];
      if ((int)threadIdx.x == p.tile_rows - 1) dn = halo_bot[j];
      N[j] = S[j] | up | dn;
    }
    u64 T[4];
    T[0] = S[0] | (C[0] & (N[0] | (N[3] << 1) | N[1]));
    T[1] = S[1] | (C[1] & (N[1] | N[0] | N[2]));
    T[2] = S[2] | (C[2] & (N[2] | N[1] | N[3]));
    T[3] = S[3] | (C[3] & (N[3] | N[2] | (N[0] >> 1)));
    // in-row propagation, 16 px each way per iteration
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T[1] |= C[1] & T[0]; T[2] |= C[2] & T[1]; T[3] |= C[3] & T[2]; T[0] |= C[0] & (T[3] << 1);
      T[2] |= C[2] & T[3]; T[1] |= C[1] & T[2]; T[0] |= C[0] & T[1]; T[3] |= C[3] & (T[0] >> 1);
    }
    const bool ch = (T[0] != S[0]) | (T[1] != S[1]) | (T[2] != S[2]) | (T[3] != S[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) S[j] = T[j];
    if (!__syncthreads_or(ch)) break;
  }

  // write back; tell the next launch whether anything another tile reads has changed:
  // columns 1 and 62 of any row (the neighbouring strips' halo bits) or the first / last tile row
  bool vis = false;
  if (rv) {
    const u64 edge_cols = (1ull << 1) | (1ull << 62);
    const bool edge_row = threadIdx.x == 0 || (int)threadIdx.x == p.tile_rows - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u64 diff = (S[j] ^ S0[j]) & BM_VALID;
      if (diff) p.bm[(((size_t)frame * p.nstrips + strip) * H + row) * BM_WORDS + j] = S[j] & BM_VALID;
      vis |= (diff & (edge_row ? BM_VALID : edge_cols)) != 0;
    }
  }
  if (__syncthreads_or(vis) && threadIdx.x == 0) atomicOr(&p.flags[p.iter], 1u);
}

hipError_t launch_hyst(const HystParams &p, hipStream_t s)
{
  const int nw = p.tile_rows / 64;
  hipLaunchKernelGGL(k_hyst, dim3((unsigned)(p.nframes * p.nstrips * p.nrtiles)), dim3(p.tile_rows), (size_t)nw * 2 * 4 * sizeof(u64), s, p);
  return hipGetLastError();
}

// =================================================================================================
// k_expand: strong plane -> u8 edge map (255 / 0); candidates left over are dropped here
// (removeCandidates, cannyEdgeD.cu:379-395).
// =================================================================================================
__global__ __launch_bounds__(256) void k_expand(const ExpandParams p)
{
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)p.nframes * p.nstrips * p.H;
  if (wave >= total) return;
  // consecutive waves walk the strips of one row, then the next row: contiguous output
  const int strip = (int)(wave % p.nstrips);
  const int row = (int)((wave / p.nstrips) % p.H);
  const int frame = (int)(wave / ((long long)p.nstrips * p.H));
  const u64 *rec = p.bm + (((size_t)frame * p.nstrips + strip) * p.H + row) * BM_WORDS;
  u32 v = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u64 w = rec[j];  // wave-uniform address
    const u32 half = lane < 32 ? (u32)w : (u32)(w >> 32);
    v |= ((half >> (lane & 31)) & 1u) ? (0xFFu << (8 * j)) : 0u;
  }
  if (lane < 1 || lane > 62) return;
  const int c0 = strip * STRIP_W - STRIP_HALO + lane * PX_PER_LANE;
  uint8_t *dst = p.out + (size_t)frame * p.out_frame_stride + (size_t)row * p.out_pitch + c0;
  if (c0 + 3 < p.W) *reinterpret_cast<u32 *>(dst) = v;
  else
    for (int k = 0; k < 4; ++k)
      if (c0 + k < p.W) dst[k] = (uint8_t)(v >> (8 * k));
}

hipError_t launch_expand(const ExpandParams &p, hipStream_t s)
{
  const long long total = (long long)p.nframes * p.nstrips * p.H;
  hipLaunchKernelGGL(k_expand, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, p);
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

__global__ void k_nms(const int16_t *sx, const int16_t *sy, size_t sp, size_t sfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int saturate)
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
{ PIX_GRID; hipLaunchKernelGGL(k_nms, grd, blk, 0, s, sx, sy, sp, sfs, out, op, ofs, W, H, saturate); return hipGetLastError(); }
hipError_t launch_thresh(const uint8_t *nms, size_t np, size_t nfs, uint8_t *out, size_t op, size_t ofs, int W, int H, int n, int low, int high, hipStream_t s)
{ PIX_GRID; hipLaunchKernelGGL(k_thresh, grd, blk, 0, s, nms, np, nfs, out, op, ofs, W, H, low, high); return hipGetLastError(); }

}  // namespace hc
