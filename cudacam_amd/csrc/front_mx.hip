// front_mx.hip -- k_front_mx: the front path of Mode R with its two integer contractions on the MATRIX pipe.
//
// Replaces the same reference kernels as k_front8 (gaussianFilter5x5, sobelXY, gradSlope, nonMaxSuppr, doubleThreshold:
// src/cvp/cannyEdgeD.cu:72-293; launch sites src/cvp/cannyEdgeH.cu:228-295) for big batches of one-channel frames, and
// hands the same things to the hysteresis: the STRONG / CANDIDATE bit planes and, in pipelined mode, the provisional map.
//
// Why.  k_front8 is bound by vector-instruction issue (35 lane-instructions per pixel, the vector pipe 95 % busy,
// profiles/r03) while the matrix pipe idles.  Two thirds of its phase 1 and phase 2 are two exact integer contractions:
//   S(r, c)    = sum K[i][j] x[r+i-2][c+j-2]                      (K = the 5x5 integer Gaussian, sum 159)
//   sumX, sumY = the 3x3 Sobel sums of the blurred bytes          (cannyEdgeD.cu:158-167)
// v_mfma_i32_32x32x32_i8 computes both exactly: D[m][n] = sum_k A[m][k] B[k][n] with i8 operands and i32 sums.  Per vertical
// tap i, A is a constant banded Toeplitz matrix (output column m <- input columns m .. m+4 of a 32-column window) and B is
// the window itself: lane (n, kh) holds 16 CONTIGUOUS row bytes of the row of its index n -- a plain ds_read_b128 from a
// row-major LDS ring, shifted by one ring row per tap.  Five MFMAs give 28 columns x 32 lanes of exact S, five more the
// Sobel sums of 28 columns.  Bytes enter as x ^ 0x80 (signed), which biases S by -128 * 159 (folded into the quotient's
// multiply-add) and leaves the zero-sum Sobel masks unchanged; zero padding is the byte 0x80 everywhere.
// Measured (tools/mfma_rate.hip, profiles/r04/mfma_rate.txt): an MFMA takes 32.5 cycles of the matrix pipe and 7-8 cycles
// of the wave's vector issue, and overlaps with the wave's own vector instructions otherwise.
//
// Geometry.  A wave owns a STRIP of 216 output columns and a RUN of rows, walked in BLOCKS of 16 rows.  The MFMA's 32
// lanes-of-N are 16 rows x 2 adjacent column TILES of 28 columns (lane = q + 16 par + 32 kh: row q, tile parity par, K-half
// / output half kh), so a block is 4 MFMA groups per stage (8 tiles).  ds_read_b128 off 16-byte alignment runs at 1/6 of
// the aligned rate (profiles/r04/lds_b128.txt), and tiles 28 columns apart are not aligned: both LDS rings therefore store
// a row as 8 SEGMENTS of 32 bytes -- segment t = the 32-column window of tile t, its last 4 bytes a copy of the next
// segment's first 4 -- 272 bytes apart (17 x 16: conflict-free).
//   input ring  20 rows: x = column - (s0 - 4); tile t's window = x in [28 t, 28 t + 32)  -> blur columns y = 28 t + m
//   blur ring   20 rows: y = column - (s0 - 2); tile u's window = y in [28 u, 28 u + 32)  -> outputs at window pos 2 .. 29,
//               i.e. image columns s0 + 28 u + m: groups of 4 aligned with the image's dwords (plane nibbles, map dwords)
// Accumulator register 4 g + j of lane (q, par, kh) is output column m = 8 g + 4 kh + j of tile 2 p + par, row q.
//
// A block (output rows R0 .. R0+15):
//   input    16 new rows (requested a block ahead) -> ^ 0x80 -> input ring (two ds_write_b32 per row: the copy)
//   blur     4 x (5 ds_read_b128, 5 MFMA) -> per pixel one v_mad_i32_i24: P = S * 105518 (+ bias), byte 3 = floor(S / 159) ^ 0x80,
//            byte 2 = 0 <=> S % 159 == 0 <=> the float chain cannot be decided by integers (see k_front8); byte permutes pack
//            4 pixels -> blur ring; flagged lanes queue (lane, group, 16 flag bits)
//   fix-up   the literal 25-fmaf chain for the flagged pixels (0.6 %), from the input ring, over their byte of the blur ring
//   Sobel    4 x (3 ds_read_b128, 5 MFMA) -> sumX, sumY per pixel -> sumX^2 + sumY^2 summed over an aligned group of 4 pixels
//            (a NECESSARY condition for "one of them passes the low threshold") -> the groups that pass queue their identity
//   NMS      k_front8's batches: 64 queued groups, one per lane, re-derive the 3 x 6 S2 values around their 4 pixels from
//            the blur ring, decide direction / non-maximum suppression / thresholds exactly, OR their nibble into a 16-row
//            plane tile in LDS and store their 4 bytes of provisional map
//   output   the plane tiles are stored once per block; the map rows were stored as zeros before the batches
// Everything a neighbouring pixel needs is re-read from the LDS rings: nothing is carried between blocks but the rings.
#include "canny_device.h"
#include <type_traits>

namespace hc {

constexpr int MX_STRIP_W = 216;                      // output columns per strip: 7 tiles of 28 + 20 columns of the eighth
constexpr int MX_ROWS = 16;                          // rows per block
constexpr int MX_RING = 20;                          // rows per LDS ring
constexpr int MX_PITCH = 272;                        // bytes per ring row: 8 segments of 32 + 16 (row 0's spare bytes are the dump slot)
constexpr int MX_RING_BYTES = MX_RING * MX_PITCH;    // 5440
constexpr int MX_DUMP = 256;                         // offset of the dump dword in the input ring
constexpr int MX_NQ = 512;                           // NMS queue entries (u16), circular
constexpr int MX_AUX = 2048;                         // fix-up queue (256 dwords) | NMS queue (1024 B) + two plane tiles (2 x 16 x 32 B)
constexpr int MX_WAVE_BYTES = 2 * MX_RING_BYTES + MX_AUX;  // 12928: 12 waves per CU
constexpr u32 MX_PAD = 0x80808080u;                  // four zero pixels, biased
constexpr u32 MX_MAGIC = 105518u;                    // ceil(2^24 / 159)
constexpr u32 MX_C0 = (u32)(20352ull * 105518ull + 0x80000000ull);  // (S - 128 * 159) * M + C0 = S * M + 2^31 (mod 2^32)

int front_mx_strips(int W) { return (W + MX_STRIP_W - 1) / MX_STRIP_W; }
int front_mx_run_rows(int blocks) { return MX_ROWS * blocks; }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef u32 gu32 __attribute__((aligned(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((aligned(4))) gu32x2;

// The MFMA A operands: lane l holds row m = l % 32, k = 16 (l / 32) .. + 15 (four dwords of four i8).  Output column m of a
// tile takes the window's columns m .. m + 4; rows m >= 28 are zero (28 outputs per 32-column window).
//   0..2  Gaussian kernel rows 0 / 4, 1 / 3, 2      3, 4  Sobel X, blur rows -1 / +1 and blur row 0      5, 6  Sobel Y, row -1 / row +1
struct alignas(16) MxATable {
  u32 v[7][64][4];
  constexpr MxATable() : v{}
  {
    constexpr int KR[3][5] = { { 2, 4, 5, 4, 2 }, { 4, 9, 12, 9, 4 }, { 5, 12, 15, 12, 5 } };
    for (int l = 0; l < 64; ++l) {
      const int m = l % 32, kh = l / 32;
      for (int dd = 0; dd < 4; ++dd)
        for (int jj = 0; jj < 4; ++jj) {
          const int t = 16 * kh + 4 * dd + jj - m;  // tap: window column - output index
          for (int a = 0; a < 7; ++a) {
            int c = 0;
            if (m < 28) {
              if (a < 3) c = (t >= 0 && t <= 4) ? KR[a][t] : 0;
              else if (a == 3) c = t == 1 ? -1 : t == 3 ? 1 : 0;                  // - left + right   (cannyEdgeD.cu:158-162)
              else if (a == 4) c = t == 1 ? -2 : t == 3 ? 2 : 0;
              else if (a == 5) c = (t == 1 || t == 3) ? 1 : t == 2 ? 2 : 0;       // row above, +      (:163-167)
              else c = (t == 1 || t == 3) ? -1 : t == 2 ? -2 : 0;                 // row below, -
            }
            v[a][l][dd] |= (u32)(c & 0xFF) << (8 * jj);
          }
        }
    }
  }
};
__device__ const MxATable MX_A{};

// 24-bit signed multiply(-add), low 32 bits: the accumulators fit 24 bits (|S - 20352| < 2^15, |sumX|, |sumY| <= 1020).
// Compiler intrinsics, NOT inline assembly: these are the first readers of the MFMA results, and the wait states an MFMA's
// destination needs before a vector instruction may read it are only inserted for instructions the compiler can see into
// (the inline-asm form read accumulator registers 0..3 too early: group 0 of every tile came out wrong).
static __device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }
static __device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
static __device__ __forceinline__ v16i mfma8(v4i a, v4i b, v16i c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

template <bool PROV>
__global__ __launch_bounds__(256, 3) void k_front_mx(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  // the run's hysteresis flag words, zeroed by the first workgroups on their way in (as k_front8)
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_count; i += gridDim.x * blockDim.x) p.zero_words[i] = 0u;
  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wib);
  if (item >= p.total_items) return;
  const int chunk = item % p.nchunks;
  const int strip = (item / p.nchunks) % p.nstrips;
  const int frame = item / (p.nchunks * p.nstrips);
  const int W = p.W, H = p.H;
  const int r0 = chunk * p.run_rows;  // output rows [r0, rend)
  const int rend = min(r0 + p.run_rows, H);
  const int s0 = strip * MX_STRIP_W;
  const int q = lane & 15, par = (lane >> 4) & 1, kh = lane >> 5;

  const u32 wbase = (u32)wib * (u32)MX_WAVE_BYTES;     // input ring
  const u32 bbase = wbase + (u32)MX_RING_BYTES;        // blur ring
  const u32 xbase = bbase + (u32)MX_RING_BYTES;        // fix-up queue | NMS queue + plane tiles
  const u32 tbase = xbase + 1024u;                     // plane tiles: [STRONG, CANDIDATE][16 rows][32 B]
  auto lds32 = [&](u32 off) -> u32 & { return *reinterpret_cast<u32 *>(smem + off); };
  auto lds16 = [&](u32 off) -> unsigned short & { return *reinterpret_cast<unsigned short *>(smem + off); };
  auto lds128 = [&](u32 off) -> v4i { return *reinterpret_cast<const v4i *>(smem + off); };

  v4i A[7];
#pragma unroll
  for (int a = 0; a < 7; ++a) A[a] = *reinterpret_cast<const v4i *>(MX_A.v[a][lane]);

  // ---- input rows ----------------------------------------------------------------------------------------------------
  // lane d holds dword d of the strip's input row: columns s0 - 4 + 4 d .. + 3 (57 dwords: x = 0 .. 227)
  const int icol = s0 - 4 + 4 * lane;
  u32 cmask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) cmask |= (lane <= 56 && icol + k >= 0 && icol + k < W) ? (0xFFu << (8 * k)) : 0u;
  const u32 ld_off = cmask ? (u32)icol : 0u;  // lanes without an image column read the row's first bytes (masked)
  const int iseg = lane / 7, ipos = lane % 7;
  // where the dword goes: its segment, and -- the first dword of a segment -- bytes 28..31 of the segment before
  const u32 wr1 = wbase + (lane <= 55 ? (u32)(32 * iseg + 4 * ipos) : (u32)MX_DUMP);
  const u32 wr2 = wbase + ((ipos == 0 && iseg >= 1 && lane <= 56) ? (u32)(32 * (iseg - 1) + 28) : (u32)MX_DUMP);
  const uint8_t *frame_base = p.in + (size_t)frame * p.in_frame_stride;
  const u32 in_pitch32 = (u32)p.in_pitch;  // launch_front_mx checks H * pitch < 2^32
  auto load_row = [&](int row) -> u32 {  // unconditional: the row clamped into the image, masked when used
    u32 lo = ld_off;
    asm volatile("" : "+v"(lo));  // keeps the lane offset out of a hoisted 64-bit VGPR pointer
    return *reinterpret_cast<const gu32 *>(frame_base + (u32)min(max(row, 0), H - 1) * in_pitch32 + lo);
  };

  // ---- per-lane constants of the epilogues ------------------------------------------------------------------------------
  const u32 lcB = (u32)(32 * par + 16 * kh);                       // B operand: the tile's segment, the lane's K half
  const u32 lcW = bbase + (u32)(32 * par + 4 * kh);                // blur write: the tile's segment, the lane's 4-column group
  const u32 fm3 = kh ? 0u : 0x80808080u;                           // group g = 3: columns 24..27 exist (kh = 0), 28..31 do not
  const u32 m00 = (strip == 0 && par == 0 && kh == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;  // blur columns -2, -1 of the frame: zero padding
  const bool edge_cols = s0 + 224 > W;                             // the strip's input window reaches past the image
  const u32 a_lo0 = p.a_lo[0], a_hi0 = p.a_hi[0], wrap_limit = p.wrap_limit;
  const int magic = (int)MX_MAGIC, c0 = (int)MX_C0;
  // Sobel tile 7 only has the five groups that lie left of column s0 + 216
  const u32 thr32 = (par && kh) ? 0xFFFFFFFFu : a_lo0, thr33 = par ? 0xFFFFFFFFu : a_lo0;

  const size_t plane_off = (size_t)frame * H * p.RD * 4;
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const u32 plane_pitch = (u32)p.RD * 4u;
  uint8_t *prov_frame = PROV ? p.prov_out + (size_t)frame * p.prov_fs : nullptr;

  int sb = 0;                  // ring slot of input row R0 and of blur row R0 - 2 (the rings advance together)
  u32 rowoff[5];               // byte offset of ring row (sb + q + k) mod 20, k = 0..4
  int qhead = 0, qcount = 0;   // NMS queue (wave-uniform)
  auto slot_off = [&](u32 k) -> u32 {  // k <= 19 + 19
    u32 t = (u32)sb + k;
    t = min(t, t - (u32)MX_RING);
    return t * (u32)MX_PITCH;
  };

  // ---- blur: input ring -> exact quotient -> blur ring; the undecidable pixels are queued ----------------------------------
  // EDGE: the block touches the frame's right / top / bottom border (or is the run's warm-up): bytes outside the image are
  // zero padding for the Sobel stage (cannyEdgeD.cu:142-149) and are never flagged
  int fqn = 0;
  auto blur_phase = [&](auto edge_c, int R0, bool warm) {
    constexpr bool EDGE = decltype(edge_c)::value;
    const int br = R0 + 2 + q;  // the lane's blur row
    const bool row_img = (u32)br < (u32)H;
    const bool row_flag = row_img && (!warm || q >= 12);  // (the warm-up block only serves blur rows r0-2 .. r0+1)
    const u32 wb = rowoff[4] + lcW;
    const v16i zero16 = {};
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      v4i B[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) B[i] = lds128(rowoff[i] + wbase + lcB + (u32)(64 * pp));
      v16i acc = mfma8(A[0], B[0], zero16);
      acc = mfma8(A[1], B[1], acc);
      acc = mfma8(A[2], B[2], acc);
      acc = mfma8(A[1], B[3], acc);
      acc = mfma8(A[0], B[4], acc);
      u32 w = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32 P[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) P[j] = (u32)mad24(acc[4 * g + j], magic, c0);
        // (q, q, f, f) of two pixels, then the four quotients / fractions of the group
        const u32 X01 = __builtin_amdgcn_perm(P[1], P[0], 0x06020703u), X23 = __builtin_amdgcn_perm(P[3], P[2], 0x06020703u);
        u32 qd = __builtin_amdgcn_perm(X23, X01, 0x05040100u);
        const u32 fd = __builtin_amdgcn_perm(X23, X01, 0x07060302u);
        u32 fm = g == 3 ? fm3 : 0x80808080u;
        if (pp == 0 && g == 0) { qd = (qd & m00) | (MX_PAD & ~m00); fm &= m00; }
        if constexpr (EDGE) {
          const int col0 = s0 - 2 + 28 * (2 * pp + par) + 8 * g + 4 * kh;  // the group's first column
          const int nv = min(max(W - col0, 0), 4);                       // its columns inside the image (col0 >= -2: handled above)
          const u32 bm = row_img ? (nv >= 4 ? 0xFFFFFFFFu : (1u << (8 * nv)) - 1u) : 0u;
          qd = (qd & bm) | (MX_PAD & ~bm);
          fm &= row_flag ? bm : 0u;
        }
        // zero-byte detector on the fraction bytes (a byte equal to 1 above a zero byte may be flagged too: harmless)
        w |= ((fd - 0x01010101u) & ~fd & fm) >> g;
        if (g < 3) lds32(wb + (u32)(64 * pp + 8 * g)) = qd;
        else if (kh == 0) lds32(wb + (u32)(64 * pp + 24)) = qd;
        // the group that opens a segment is also bytes 28..31 of the segment before
        if (g == 0 && kh == 0 && (pp > 0 || par == 1)) lds32(wb + (u32)(64 * pp) - 4u) = qd;
      }
      // entry: bits 4..7 of byte j = groups 3..0 of column j; low nibble of byte 0 = q, of byte 1 = par, kh, pp
      const u64 any = __ballot(w != 0);
      if (any) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(any >> 32), __builtin_amdgcn_mbcnt_lo((u32)any, (u32)fqn));
        if (w) lds32(xbase + 4u * rank) = w | (u32)q | ((u32)(lane >> 4) << 8) | ((u32)pp << 10);
        fqn += __popcll(any);
      }
    }
  };

  // the literal reference chain (cannyEdgeD.cu:102-115) from the input ring: bytes outside the image are stored as zero
  // pixels there, and a zero tap leaves the running sum unchanged exactly as the reference's skipped taps do
  auto fixup = [&]() {
#pragma nounroll
    for (int base = 0; base < fqn; base += 64) {
      const int e = base + lane;
      const u32 ent = e < fqn ? lds32(xbase + 4u * (u32)e) : 0u;
      u32 fl = ent & 0xF0F0F0F0u;
      const u32 eq = ent & 15u, epar = (ent >> 8) & 1u, ekh = (ent >> 9) & 1u, t = 2u * ((ent >> 10) & 3u) + epar;
      u32 ro[5];
#pragma unroll
      for (int r = 0; r < 5; ++r) ro[r] = slot_off(eq + (u32)r);
      while (fl) {
        const u32 b = (u32)__builtin_ctz(fl);
        fl &= fl - 1;
        const u32 m = 8u * (7u - (b & 7u)) + 4u * ekh + (b >> 3);  // the pixel's column of the tile
        u32 px[25];  // all 25 taps are requested before the first is used
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
          for (int c = 0; c < 5; ++c) px[r * 5 + c] = smem[wbase + ro[r] + 32u * t + m + (u32)c];
        asm volatile("" ::: "memory");
        float f = 0.0f;
#pragma unroll
        for (int i = 0; i < 25; ++i) f = __builtin_fmaf(GKC.v[i], (float)(px[i] ^ 0x80u), f);
        const unsigned char v = (unsigned char)((u32)(int)f ^ 0x80u);
        smem[bbase + ro[4] + 32u * t + m] = v;
        if (m < 4u && t >= 1u) smem[bbase + ro[4] + 32u * t - 4u + m] = v;
      }
    }
  };

  // ---- one dense NMS pass: up to 64 queued groups, an entry per lane (the arithmetic of k_front8's nms_batch) ---------------
  auto nms_batch = [&](int nent, int R0) {
    wave_lds_sync();
    const bool live = lane < nent;  // the other lanes compute on stale ids and store nothing
    const u32 id = lds16(xbase + 2u * ((u32)(qhead + lane) & (u32)(MX_NQ - 1)));  // bits 0..3 row, 4 tile parity, 5 half, 6..7 g, 8..9 pp
    const u32 eq = id & 15u, eh = (id >> 5) & 1u, eg = (id >> 6) & 3u, u = 2u * ((id >> 8) & 3u) + ((id >> 4) & 1u);
    const int row = R0 + (int)eq;
    const u32 c4 = 7u * u + 2u * eg + eh;           // the group's index in the strip
    const int col0 = s0 + 4 * (int)c4;              // column of the group's pixel 0
    // blur rows row-2 .. row+2; window bytes 8 g + 4 h .. + 7 of segment u = pixels -2 .. 5
    const u32 lo = bbase + 32u * u + 8u * eg + 4u * eh;
    u32 d[5][3], s[5][3];  // per blur row: d = b[+1]-b[-1], s = b[-1]+2b[0]+b[+1] of the pixel pairs (-1,0), (1,2), (3,4)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const u32 a = lo + slot_off(eq + (u32)r);
      const u32 E0 = lds32(a) ^ MX_PAD, E1 = lds32(a + 4u) ^ MX_PAD;  // (the ring holds x ^ 0x80: a uniform bias only for SIGNED readers)
      const u32 Pm = unpack_lo(E0), Aq = unpack_hi(E0), Bq = unpack_lo(E1), Cq = unpack_hi(E1);  // (b-2,b-1) (b0,b1) (b2,b3) (b4,b5)
      const u32 c0p = pair_shift(Aq, Pm), c1p = pair_shift(Bq, Aq), c2p = pair_shift(Cq, Bq);   // (b-1,b0) (b1,b2) (b3,b4)
      d[r][0] = R(I(Aq) - I(Pm)); d[r][1] = R(I(Bq) - I(Aq)); d[r][2] = R(I(Cq) - I(Bq));
      s[r][0] = pk_mad2(c0p, Pm + Aq); s[r][1] = pk_mad2(c1p, Aq + Bq); s[r][2] = pk_mad2(c2p, Bq + Cq);
    }
    // zero padding of the Sobel / gradient stages (cannyEdgeD.cu:142-149, 222-229): the sums of columns and rows outside
    // the image are zero.  Only groups at the frame's border see any: the masks are a variant of their own (wave-uniform)
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
    const bool at_border = live && (col0 == 0 || col0 + 4 >= W || row == 0 || row + 1 >= H);
    if (__ballot(at_border) != 0) sums(std::true_type{});
    else sums(std::false_type{});
    // S*[0] / [5]: the neighbouring pixels -1 / 4; S*[1 + k]: the group's own pixel k
    const u32 gmax = max(max(SC[1], SC[2]), max(SC[3], SC[4]));
    const bool wraps = __ballot(live && gmax >= wrap_limit) != 0;
    u32 nibS = 0, nibC = 0;
    auto px = [&](auto kc, u32 A2, u32 Um, u32 Vp) {
      constexpr int k = decltype(kc)::value, e = (k + 1) % 2;
      const u32 g = SC[1 + k];
      bool cand = g >= a_lo0, strong = g >= a_hi0;
      if (wraps) {  // u8 wrap of gradients >= 256 (cannyEdgeD.cu:267): the bands of S2 whose low byte passes the thresholds
        const bool w0 = g >= 262144u, w1 = g >= 1048576u;
        cand = (cand && !w0) || (g >= p.a_lo[1] && !w1) || g >= p.a_lo[2];
        strong = (strong && !w0) || (g >= p.a_hi[1] && !w1) || g >= p.a_hi[2];
      }
      // direction bins (cannyEdgeD.cu:239-264) without atan2: E1 = 2x(x-y) - S2 > 0, E2 = 2x(x+y) - S2 > 0
      const bool p1 = mul16<e, e>(A2, Um) > (int)g, p2 = mul16<e, e>(A2, Vp) > (int)g;
      // neighbours (:245-264): bin0 down/up, bin1 down-left/up-right, bin2 right/left, bin3 up-left/down-right
      const u32 n0 = max(SN[1 + k], SU[1 + k]), n1 = max(SN[k], SU[2 + k]);
      const u32 n2 = max(SC[2 + k], SC[k]), n3 = max(SU[k], SN[2 + k]);
      const u32 mb = p1 ? (p2 ? n2 : n3) : (p2 ? n1 : n0);
      const bool keep = mb <= g;  // non-strict on both sides, as the reference
      nibS |= (strong && keep) ? (1u << k) : 0u;
      nibC |= (cand && keep) ? (1u << k) : 0u;
    };
    u32 A2[3], Um[3], Vp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { A2[j] = R(U(Xc[j]) + U(Xc[j])); Um[j] = R(I(Xc[j]) - I(Yc[j])); Vp[j] = R(U(Xc[j]) + U(Yc[j])); }
    px(std::integral_constant<int, 0>{}, A2[0], Um[0], Vp[0]);
    px(std::integral_constant<int, 1>{}, A2[1], Um[1], Vp[1]);
    px(std::integral_constant<int, 2>{}, A2[1], Um[1], Vp[1]);
    px(std::integral_constant<int, 3>{}, A2[2], Um[2], Vp[2]);
    // (out-of-image pixels have S2 = 0: never candidates, a_lo >= 4)
    if (live) {
      const u32 byte = c4 >> 1, sh = ((byte & 3u) << 3) + ((c4 & 1u) << 2);
      const u32 ta = tbase + eq * 32u + (byte & ~3u);
      if (nibS) atomicOr(&lds32(ta), nibS << sh);
      if (nibC) atomicOr(&lds32(ta + 512u), nibC << sh);
      if constexpr (PROV) *reinterpret_cast<gu32 *>(prov_frame + (u32)row * p.prov_pitch + (u32)col0) = nibble_to_bytes(nibS);
    }
    qhead = (qhead + nent) & (MX_NQ - 1);
    qcount -= nent;
  };

  // ---- Sobel: blur ring -> sumX, sumY -> which groups of 4 pixels may hold a candidate -> NMS batches ------------------------
  auto sobel_phase = [&](auto edge_c, int R0) {
    constexpr bool EDGE = decltype(edge_c)::value;
    const bool row_ok = R0 + q < rend;
    const v16i zero16 = {};
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      v4i B[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) B[i] = lds128(rowoff[1 + i] + bbase + lcB + (u32)(64 * pp));
      v16i aX = mfma8(A[3], B[0], zero16);
      v16i aY = mfma8(A[5], B[0], zero16);
      aX = mfma8(A[4], B[1], aX);
      aX = mfma8(A[3], B[2], aX);
      aY = mfma8(A[6], B[2], aY);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // S2 = sumX^2 + sumY^2 (the reference's float gradient is a strictly increasing function of it, k_front8); summed over
        // the group: at least any of the four -- a necessary condition, the batch decides exactly
        int T = mul24(aX[4 * g], aX[4 * g]);
        T = mad24(aY[4 * g], aY[4 * g], T);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          T = mad24(aX[4 * g + j], aX[4 * g + j], T);
          T = mad24(aY[4 * g + j], aY[4 * g + j], T);
        }
        u32 thr = (pp == 3 && g == 2) ? thr32 : (pp == 3 && g == 3) ? thr33 : a_lo0;
        if constexpr (EDGE) {
          const int col0 = s0 + 28 * (2 * pp + par) + 8 * g + 4 * kh;
          thr = (row_ok && col0 < W) ? thr : 0xFFFFFFFFu;
        }
        const bool pass = (u32)T >= thr;
        const u64 mk = __ballot(pass);
        if (mk) {
          const u32 pos = ((u32)(qhead + qcount) + mbcnt64(mk)) & (u32)(MX_NQ - 1);
          if (pass) lds16(xbase + 2u * pos) = (unsigned short)((u32)lane | ((u32)g << 6) | ((u32)pp << 8));
          qcount += __popcll(mk);
        }
      }
      while (qcount >= 64) nms_batch(64, R0);
    }
    while (qcount > 0) nms_batch(min(qcount, 64), R0);
  };

  // ---- the run: a warm-up block (blur rows r0-2 .. r0+1), then the blocks of output rows r0 + 16 b .. + 15 --------------------
  u32 xr[MX_ROWS];
#pragma unroll
  for (int k = 0; k < MX_ROWS; ++k) xr[k] = load_row(r0 - 12 + k);
  const int nblocks = (rend - r0 + MX_ROWS - 1) / MX_ROWS;
  const u32 prow = (u32)lane >> 2, ppart = (u32)lane & 3u;  // plane-tile / map-row stores: 4 lanes per row
#pragma nounroll
  for (int b = -1; b < nblocks; ++b) {
    const int R0 = r0 + MX_ROWS * b;
    const bool warm = b < 0;
    // new input rows R0+4 .. R0+19
#pragma unroll
    for (int k = 0; k < MX_ROWS; ++k) {
      const int row = R0 + 4 + k;
      const u32 m = (u32)row < (u32)H ? cmask : 0u;  // rows above / below the image are zero padding (cannyEdgeD.cu:91-98)
      const u32 x = (xr[k] & m) ^ MX_PAD;
      u32 sl = (u32)sb + 4u + (u32)k;
      sl = sl >= (u32)MX_RING ? sl - (u32)MX_RING : sl;
      lds32(wr1 + sl * (u32)MX_PITCH) = x;
      lds32(wr2 + sl * (u32)MX_PITCH) = x;
    }
    if (b + 1 < nblocks) {
#pragma unroll
      for (int k = 0; k < MX_ROWS; ++k) xr[k] = load_row(R0 + 20 + k);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) rowoff[k] = slot_off((u32)(q + k));
    wave_lds_sync();
    fqn = 0;
    const bool edge_rows = warm || R0 + 2 < 0 || R0 + 17 >= H;
    if (edge_rows || edge_cols) blur_phase(std::true_type{}, R0, warm);
    else blur_phase(std::false_type{}, R0, warm);
    wave_lds_sync();
    fixup();
    wave_lds_sync();
    if (p.dbg_blur) {  // diagnostics (HC_OPT_DEBUG_TAPS): the fixed-up blur rows of this run
      for (int k = 0; k < MX_ROWS; ++k) {
        const int br = R0 + 2 + k;
        if (br < r0 || br >= rend) continue;
        for (int c = lane; c < MX_STRIP_W && s0 + c < W; c += 64) {
          const u32 y = (u32)c + 2u;
          p.dbg_blur[(size_t)frame * p.dbg_fs + (size_t)br * p.dbg_pitch + (u32)(s0 + c)] = smem[bbase + slot_off(4u + (u32)k) + 32u * (y / 28u) + y % 28u] ^ 0x80u;
        }
      }
    }
    if (!warm) {
      // zero the plane tiles (the fix-up queue is done with the space); the map rows of the block are stored as zeros now,
      // the few groups with strong pixels are overwritten by the batches (stores of one wave to one address keep their order)
      *reinterpret_cast<v4i *>(smem + tbase + 16u * (u32)lane) = v4i{ 0, 0, 0, 0 };
      if constexpr (PROV) {
        if (R0 + (int)prow < rend) {
          uint8_t *pr = prov_frame + (u32)(R0 + (int)prow) * p.prov_pitch + (u32)s0;
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            const u32 i8 = ppart + 4u * (u32)k;  // 27 groups of 8 columns
            if (i8 < 27u && s0 + 8 * (int)i8 < W) *reinterpret_cast<gu32x2 *>(pr + 8u * i8) = u32x2{ 0u, 0u };
          }
        }
      }
      wave_lds_sync();
      const bool edge_out = R0 + MX_ROWS > rend || s0 + MX_STRIP_W > W;
      if (edge_out) sobel_phase(std::true_type{}, R0);
      else sobel_phase(std::false_type{}, R0);
      wave_lds_sync();
      // the block's plane bytes: 16 rows x 27 bytes per plane, 7 bytes per lane
      if (R0 + (int)prow < rend) {
        const u32 go = (u32)(R0 + (int)prow) * plane_pitch + (u32)(27 * strip);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const u32 bb = ppart * 7u + (u32)k;
          if (bb < 27u && (u32)(27 * strip) + bb < plane_pitch) {
            splane[go + bb] = smem[tbase + prow * 32u + bb];
            cplane[go + bb] = smem[tbase + 512u + prow * 32u + bb];
          }
        }
      }
      wave_lds_sync();  // (the next block's fix-up queue overwrites the tiles)
    }
    sb = sb + MX_ROWS >= MX_RING ? sb + MX_ROWS - MX_RING : sb + MX_ROWS;
  }
}

size_t front_mx_lds_bytes() { return (size_t)4 * MX_WAVE_BYTES; }

// Mode R, one-channel frames: strips of 216 columns, runs of 16 * blocks rows
hipError_t launch_front_mx(const FrontParams &p, hipStream_t s)
{
  if (p.bgr || p.run_rows < MX_ROWS || p.run_rows % MX_ROWS != 0 || p.nchunks * p.run_rows < p.H) return hipErrorInvalidValue;
  if (p.nstrips != front_mx_strips(p.W) || (long)p.total_items != (long)p.nframes * p.nstrips * p.nchunks) return hipErrorInvalidValue;
  const size_t w4 = ((size_t)p.W + 3) / 4 * 4;
  if ((unsigned long long)p.H * p.in_pitch >= (1ull << 32) || p.in_pitch < w4) return hipErrorInvalidValue;
  if (p.prov_out && (p.W % 8 != 0)) return hipErrorInvalidValue;
  if (p.dbg_blur && p.dbg_pitch < (u32)p.W) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((p.total_items + 3) / 4)), block(256);
  const size_t lds = front_mx_lds_bytes();
  if (p.prov_out) hipLaunchKernelGGL((k_front_mx<true>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((k_front_mx<false>), grid, block, lds, s, p);
  return hipGetLastError();
}

}  // namespace hc
