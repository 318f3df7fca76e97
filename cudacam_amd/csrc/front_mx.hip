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
// tap i, A is a constant banded Toeplitz matrix (an output column <- input columns +0 .. +4 of a 32-column window) and B is
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
// a row as 16 bytes of padding + 8 SEGMENTS of 32 bytes -- segment t = the 32-column window of tile t, its last 4 bytes a
// copy of the next segment's first 4 -- 272 bytes apart (17 x 16: rows spread over the banks).
//   input ring  20 rows: x = column - (s0 - 4); tile t's window = x in [28 t, 28 t + 32)  -> blur columns y = 28 t + o
//   blur ring   20 rows: y = column - (s0 - 2); tile u's window = y in [28 u, 28 u + 32)  -> outputs at window pos 2 .. 29,
//               i.e. image columns s0 + 28 u + o: groups of 4 aligned with the image's dwords (plane nibbles, map dwords)
// The rows of the A matrices are permuted so that accumulator register v of lane (q, par, kh) is output column
// o = 16 kh + v of tile 2 pp + par, row q: a lane's 16 results are 16 ADJACENT columns -- the blur goes back to LDS as one
// ds_write_b128, four aligned groups of 4 pixels per lane.
//
// A block b of a run (output rows r0 .. rend): R0 = r0 - 4 + 16 b
//   input    rows R0+4 .. R0+19 (requested a block ahead; the first block also R0 .. R0+3) -> ^ 0x80 -> input ring, one
//            ds_write_b32 per row: lanes 0..55 hold the window's dwords, lanes 56..63 load the first dword of a segment once
//            more and store it as the tail of the segment before
//   blur     rows R0+2 .. R0+17: 4 x (5 ds_read_b128, 5 MFMA) -> per pixel one v_mad_i32_i24: P = S * 105518 (+ bias), byte 3 =
//            floor(S / 159) ^ 0x80, byte 2 = 0 <=> S % 159 == 0 <=> the float chain cannot be decided by integers (see
//            k_front8); byte permutes pack 4 pixels -> blur ring; flagged lanes queue (lane, group, 16 flag bits)
//   fix-up   the literal 25-fmaf chain for the flagged pixels (0.6 %), from the input ring, over their byte of the blur ring
//   Sobel    rows R0 .. R0+15: 4 x (3 ds_read_b128, 5 MFMA) -> sumX, sumY per pixel -> sumX^2 + sumY^2 summed over an aligned
//            group of 4 pixels (a NECESSARY condition for "one of them passes the low threshold") -> the groups that pass
//            queue their identity
//   NMS      k_front8's batches: 64 queued groups, one per lane, re-derive the 3 x 6 S2 values around their 4 pixels from
//            the blur ring, decide direction / non-maximum suppression / thresholds exactly, OR their nibble into a 16-row
//            plane tile in LDS and store their 4 bytes of provisional map
//   output   the plane tiles are stored once per block; the map rows were stored as zeros before the batches
// The Sobel stage lags the blur stage by 4 rows, so a run needs no warm-up block: its first block discards 4 of its 16
// output rows (they belong to the run above), a run of n blocks covers 16 n - 4 rows.
// Everything a neighbouring pixel needs is re-read from the LDS rings: nothing is carried between blocks but the rings.
#include "canny_device.h"
#include <type_traits>

#ifndef MX_ABL
#define MX_ABL 0  // timing experiments (tools/build_variant.sh, WRONG results): 1 no NMS batches, 2 no fix-up, 4 no stores, 8 no Sobel stage, 16 no blur stage, 32 no input loads, 64 no zero stores of the provisional map
#endif

namespace hc {

constexpr int MX_STRIP_W = 216;                      // output columns per strip: 7 tiles of 28 + 20 columns of the eighth
constexpr int MX_ROWS = 16;                          // rows per block
constexpr int MX_LAG = 4;                            // the Sobel stage's rows trail the blur stage's by 4
constexpr int MX_RING = 20;                          // rows per LDS ring
constexpr int MX_SEG0 = 16;                          // a ring row: 16 bytes of padding, then 8 segments of 32 bytes
constexpr int MX_PITCH = 272;
constexpr int MX_RING_BYTES = MX_RING * MX_PITCH;    // 5440
constexpr int MX_NQ = 512;                           // NMS queue entries (u16), circular
constexpr int MX_AUX = 2048;                         // fix-up queue (256 dwords) | NMS queue (1024 B); two plane tiles (2 x 16 x 32 B)
constexpr int MX_WAVE_BYTES = 2 * MX_RING_BYTES + MX_AUX;  // 12928: 12 waves per CU
constexpr u32 MX_PAD = 0x80808080u;                  // four zero pixels, biased
constexpr u32 MX_MAGIC = 105518u;                    // ceil(2^24 / 159)
constexpr u32 MX_C0 = (u32)(20352ull * 105518ull + 0x80000000ull);  // (S - 128 * 159) * M + C0 = S * M + 2^31 (mod 2^32)

int front_mx_strips(int W) { return (W + MX_STRIP_W - 1) / MX_STRIP_W; }
int front_mx_run_rows(int blocks) { return MX_ROWS * blocks - MX_LAG; }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef u32 gu32 __attribute__((aligned(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((aligned(4))) gu32x2;

// The MFMA A operands: lane l holds row m = l % 32, k = 16 (l / 32) .. + 15 (four dwords of four i8).  Row m produces
// output column o = 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3) of the tile (so that register v of a lane is column 16 kh + v)
// from the window's columns o .. o + 4; rows with o >= 28 are zero (28 outputs per 32-column window).
//   0..2  Gaussian kernel rows 0 / 4, 1 / 3, 2      3, 4  Sobel X, blur rows -1 / +1 and blur row 0      5, 6  Sobel Y, row -1 / row +1
struct alignas(16) MxATable {
  u32 v[7][64][4];
  constexpr MxATable() : v{}
  {
    constexpr int KR[3][5] = { { 2, 4, 5, 4, 2 }, { 4, 9, 12, 9, 4 }, { 5, 12, 15, 12, 5 } };
    for (int l = 0; l < 64; ++l) {
      const int m = l % 32, kh = l / 32;
      const int o = 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
      for (int dd = 0; dd < 4; ++dd)
        for (int jj = 0; jj < 4; ++jj) {
          const int t = 16 * kh + 4 * dd + jj - o;  // tap: window column - output column
          for (int a = 0; a < 7; ++a) {
            int c = 0;
            if (o < 28) {
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

// WPB: waves per workgroup (independent waves; 1: a retiring wave's slot and LDS are free for the next one at once)
template <bool PROV, int WPB>
__global__ __launch_bounds__(64 * WPB, 3) void k_front_mx(const FrontParams p)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  // the run's hysteresis flag words, zeroed by the first workgroups on their way in (as k_front8)
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_count; i += gridDim.x * blockDim.x) p.zero_words[i] = 0u;
  const int item = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPB + wib);
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
  const u32 xbase = bbase + (u32)MX_RING_BYTES;        // fix-up queue | NMS queue
  const u32 tbase = xbase + 1024u;                     // plane tiles: [STRONG, CANDIDATE][16 rows][32 B]
  // where a lane stores what nobody reads: the padding of the input ring's rows, a dword of its own per lane
  const u32 dump = wbase + (u32)(lane >> 2) * (u32)MX_PITCH + 4u * (u32)(lane & 3);
  auto lds32 = [&](u32 off) -> u32 & { return *reinterpret_cast<u32 *>(smem + off); };
  auto lds16 = [&](u32 off) -> unsigned short & { return *reinterpret_cast<unsigned short *>(smem + off); };
  auto lds128 = [&](u32 off) -> v4i & { return *reinterpret_cast<v4i *>(smem + off); };

  v4i A[7];
#pragma unroll
  for (int a = 0; a < 7; ++a) A[a] = *reinterpret_cast<const v4i *>(MX_A.v[a][lane]);

  // ---- input rows ----------------------------------------------------------------------------------------------------
  // lanes 0..56 hold dword d = lane of the strip's input row: window x = 4 d .. + 3, columns s0 - 4 + x (57 dwords: x = 0 ..
  // 227); lanes 57..63 hold the first dword of segments 1..7 (x = 28, 56, ..) a second time
  const int xw = lane <= 56 ? 4 * lane : 28 * (lane - 56);
  const int icol = s0 - 4 + xw;
  u32 cmask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) cmask |= (icol + k >= 0 && icol + k < W) ? (0xFFu << (8 * k)) : 0u;
  const u32 ld_off = cmask ? (u32)icol : 0u;  // lanes without an image column read the row's first bytes (masked)
  // where the dword goes: lanes 0..55 its place in its segment; lane 56 (x = 224) and lanes 57..63 bytes 28..31 of the segment before
  const u32 wr1 = wbase + (u32)MX_SEG0 + (lane <= 55 ? (u32)(32 * (lane / 7) + 4 * (lane % 7)) : lane == 56 ? (u32)(32 * 7 + 28) : (u32)(32 * (lane - 57) + 28));
  const uint8_t *frame_base = p.in + (size_t)frame * p.in_frame_stride;
  const u32 in_pitch32 = (u32)p.in_pitch;  // launch_front_mx checks H * pitch < 2^32
  auto load_row = [&](int row) -> u32 {  // unconditional: the row clamped into the image, masked when used
    if (MX_ABL & 32) return (u32)row;
    u32 lo = ld_off;
    asm volatile("" : "+v"(lo));  // keeps the lane offset out of a hoisted 64-bit VGPR pointer
    return *reinterpret_cast<const gu32 *>(frame_base + (u32)min(max(row, 0), H - 1) * in_pitch32 + lo);
  };

  // ---- per-lane constants of the epilogues ------------------------------------------------------------------------------
  const u32 lcS = (u32)(MX_SEG0 + 32 * par + 16 * kh);             // the tile's segment, the lane's half: B operands, blur writes
  // the first group of a tile is also the tail of the segment before (tile 0: the row's padding); lanes kh = 1 have no such
  // group and store into their row's padding instead: 8 bytes a row, a dword per lane
  const u32 lcC = kh ? 4u * (u32)par : (u32)(MX_SEG0 + 32 * par - 4), lcCstep = kh ? 0u : 64u;
  const u32 fm3 = kh ? 0u : 0x80808080u;                           // dword 3 of the lane: columns 12..15 exist, 28..31 do not
  const u32 m00 = (strip == 0 && par == 0 && kh == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;  // blur columns -2, -1 of the frame: zero padding
  const bool edge_cols = s0 + 224 > W;                             // the strip's input window reaches past the image
  const u32 a_lo0 = p.a_lo[0], a_hi0 = p.a_hi[0], wrap_limit = p.wrap_limit;
  const int magic = (int)MX_MAGIC, c0 = (int)MX_C0;
  // Sobel groups that do not exist: dword 3 of the lanes kh = 1 (columns 28..31 of a tile), and in tile 7 everything right
  // of column s0 + 216 (its lanes kh = 1 only have dword 0)
  const u32 thr3 = kh ? 0xFFFFFFFFu : a_lo0, thr7 = (par && kh) ? 0xFFFFFFFFu : a_lo0;

  const size_t plane_off = (size_t)frame * H * p.RD * 4;
  uint8_t *splane = reinterpret_cast<uint8_t *>(p.sbits) + plane_off;
  uint8_t *cplane = reinterpret_cast<uint8_t *>(p.cbits) + plane_off;
  const u32 plane_pitch = (u32)p.RD * 4u;
  // A strip's 27 plane bytes of a row start at byte 27 * strip: byte stores are slow (14 of them per block cost 0.4 ms of a
  // 2.3 ms launch, profiles/r04/mx_ablation.txt).  The plane tiles in LDS are therefore kept in the global dwords'
  // alignment -- the strip's byte k is tile byte psh + k, psh = (27 * strip) % 4 -- and a row goes out as its 6 whole
  // dwords (96 a block: lanes 0..63, then 0..31) and its 3 leading / trailing bytes (48 a block: lanes 0..47)
  const u32 psh = (u32)(27 * strip) & 3u, pd0 = psh ? 1u : 0u, pnh = psh ? 4u - psh : 0u;
  const u32 pg0 = (u32)(27 * strip) - psh;  // the row's byte offset of tile byte 0
  const u32 pdA = (u32)lane / 6u, pdAo = 4u * (pd0 + (u32)lane % 6u);                  // dword store 1: row, tile byte
  const u32 pdB = (64u + (u32)lane) / 6u, pdBo = 4u * (pd0 + (64u + (u32)lane) % 6u);  // dword store 2 (lanes 0..31)
  const u32 pbR = (u32)lane / 3u, pbK = (u32)lane % 3u;                                // byte store (lanes 0..47)
  const u32 pbO = pbK < pnh ? psh + pbK : 4u * (pd0 + 6u) + (pbK - pnh);
  uint8_t *prov_frame = PROV ? p.prov_out + (size_t)frame * p.prov_fs : nullptr;

  int sb = 0;                  // ring slot of input row R0 and of blur row R0 - 2 (the rings advance together)
  u32 rowoff[5];               // byte offset of ring row (sb + q + k) mod 20, k = 0..4
  int qhead = 0, qcount = 0;   // NMS queue (wave-uniform)
  auto slot_off = [&](u32 k) -> u32 {  // k <= 19 + 19
    u32 t = (u32)sb + k;
    t = min(t, t - (u32)MX_RING);
    return __umul24(t, (u32)MX_PITCH);
  };

  // ---- blur: input ring -> exact quotient -> blur ring; the undecidable pixels are queued ----------------------------------
  // EDGE: the block touches the frame's right / top / bottom border: bytes outside the image are zero padding for the Sobel
  // stage (cannyEdgeD.cu:142-149) and are never flagged
  int fqn = 0;
  auto blur_phase = [&](auto edge_c, int R0) {
    constexpr bool EDGE = decltype(edge_c)::value;
    const int br = R0 + 2 + q;  // the lane's blur row
    const bool row_img = (u32)br < (u32)H;
    const u32 wb = rowoff[4] + bbase + lcS;
    u32 wc = rowoff[4] + bbase + lcC;
    const v16i zero16 = {};
    // The MFMAs of group pp + 1 are issued one by one between the four parts of the vector code that finishes group pp (a
    // dependent MFMA chain issues one every 32 cycles, 8 of them the wave's own; alone in a row the five of a group stall the
    // wave for 160 cycles and the first reader of the sums for 64 more).  sched_barrier pins that order.
    v4i B[5];
    auto reads = [&](int pp) {
#pragma unroll
      for (int i = 0; i < 5; ++i) B[i] = lds128(rowoff[i] + wbase + lcS + (u32)(64 * pp));
    };
    reads(0);
    v16i acc = mfma8(A[0], B[0], zero16), nxt = zero16;
    acc = mfma8(A[1], B[1], acc);
    acc = mfma8(A[2], B[2], acc);
    acc = mfma8(A[1], B[3], acc);
    acc = mfma8(A[0], B[4], acc);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      if (pp < 3) reads(pp + 1);
      u32 w = 0;
      v4i out;
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // the lane's dword g: columns 16 kh + 4 g .. + 3 of the tile
        __builtin_amdgcn_sched_barrier(0);
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
          const int col0 = s0 - 2 + 28 * (2 * pp + par) + 16 * kh + 4 * g;  // the group's first column
          const int nv = min(max(W - col0, 0), 4);                         // its columns inside the image (col0 >= -2: handled above)
          const u32 bm = row_img ? (nv >= 4 ? 0xFFFFFFFFu : (1u << (8 * nv)) - 1u) : 0u;
          qd = (qd & bm) | (MX_PAD & ~bm);
          fm &= bm;
        }
        // zero-byte detector on the fraction bytes (a byte equal to 1 above a zero byte may be flagged too: harmless)
        w |= ((fd - 0x01010101u) & ~fd & fm) >> g;
        out[g] = (int)qd;
        __builtin_amdgcn_sched_barrier(0);
        if (pp < 3) {  // (the reads were issued a part ago)
          if (g == 0) nxt = mfma8(A[0], B[0], zero16);
          else if (g == 1) nxt = mfma8(A[1], B[1], nxt);
          else if (g == 2) nxt = mfma8(A[2], B[2], nxt);
          else nxt = mfma8(A[1], B[3], nxt);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      lds128(wb + (u32)(64 * pp)) = out;      // (dword 3 of the lanes kh = 1 is not a column of the tile: overwritten below)
      lds32(wc) = (u32)out[0];                 // after it, in program order: the tail of the segment before
      wc += lcCstep;
      // entry: bits 4..7 of byte j = dwords 3..0 of column j; low nibble of byte 0 = q, of byte 1 = par, kh, pp
      const u64 any = __ballot(w != 0);
      const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(any >> 32), __builtin_amdgcn_mbcnt_lo((u32)any, (u32)fqn));
      lds32(lane_sel(any, xbase + 4u * rank, dump)) = w | (u32)q | ((u32)(lane >> 4) << 8) | ((u32)pp << 10);
      fqn += __popcll(any);
      __builtin_amdgcn_sched_barrier(0);
      if (pp < 3) acc = mfma8(A[0], B[4], nxt);
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
      const u32 segb = (u32)MX_SEG0 + 32u * t;
      while (fl) {
        const u32 b = (u32)__builtin_ctz(fl);
        fl &= fl - 1;
        const u32 o = 16u * ekh + 4u * (7u - (b & 7u)) + (b >> 3);  // the pixel's column of the tile: its taps are bytes o .. o + 4 of the segment
        const u32 a0 = wbase + segb + (o & ~3u), sh = o & 3u, sh8 = 8u * sh;
        u32 lo[5], hi[5];  // all 10 dwords are requested before the first is used
#pragma unroll
        for (int r = 0; r < 5; ++r) { lo[r] = lds32(a0 + ro[r]); hi[r] = lds32(a0 + ro[r] + 4u); }
        asm volatile("" ::: "memory");
        float f = 0.0f;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const u32 w0 = __builtin_amdgcn_alignbyte(hi[r], lo[r], sh) ^ MX_PAD;   // taps 0..3
          const u32 w1 = (hi[r] >> sh8) ^ 0x80u;                                  // tap 4 (byte 0)
          f = __builtin_fmaf(GKC.v[5 * r + 0], (float)(w0 & 0xFFu), f);
          f = __builtin_fmaf(GKC.v[5 * r + 1], (float)((w0 >> 8) & 0xFFu), f);
          f = __builtin_fmaf(GKC.v[5 * r + 2], (float)((w0 >> 16) & 0xFFu), f);
          f = __builtin_fmaf(GKC.v[5 * r + 3], (float)(w0 >> 24), f);
          f = __builtin_fmaf(GKC.v[5 * r + 4], (float)(w1 & 0xFFu), f);
        }
        const unsigned char v = (unsigned char)((u32)(int)f ^ 0x80u);
        const u32 wa = bbase + ro[4] + segb + o;
        smem[wa] = v;
        if (o < 4u) smem[wa - 4u] = v;  // (tile 0: into the row's padding)
      }
    }
  };

  // ---- one dense NMS pass: up to 64 queued groups, an entry per lane (the arithmetic of k_front8's nms_batch) ---------------
  auto nms_batch = [&](int nent, int R0) {
    wave_lds_sync();
    const bool live = lane < nent;  // the other lanes compute on stale ids and store nothing
    const u32 id = lds16(xbase + 2u * ((u32)(qhead + lane) & (u32)(MX_NQ - 1)));  // bits 0..3 row, 4 tile parity, 5 half, 6..7 dword, 8..9 pp
    const u32 eq = id & 15u, eh = (id >> 5) & 1u, eg = (id >> 6) & 3u, u = 2u * ((id >> 8) & 3u) + ((id >> 4) & 1u);
    const int row = R0 + (int)eq;
    const u32 c4 = 7u * u + 4u * eh + eg;           // the group's index in the strip
    const int col0 = s0 + 4 * (int)c4;              // column of the group's pixel 0
    // blur rows row-2 .. row+2; window bytes 16 h + 4 g .. + 7 of segment u = pixels -2 .. 5
    const u32 lo = bbase + (u32)MX_SEG0 + 32u * u + 16u * eh + 4u * eg;
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
      const u32 byte = psh + (c4 >> 1), sh = ((byte & 3u) << 3) + ((c4 & 1u) << 2);
      const u32 ta = tbase + eq * 32u + (byte & ~3u);
      if (nibS) atomicOr(&lds32(ta), nibS << sh);
      if (nibC) atomicOr(&lds32(ta + 512u), nibC << sh);
      if constexpr (PROV && !(MX_ABL & 4)) *reinterpret_cast<gu32 *>(prov_frame + (u32)row * p.prov_pitch + (u32)col0) = nibble_to_bytes(nibS);
    }
    qhead = (qhead + nent) & (MX_NQ - 1);
    qcount -= nent;
  };

  // ---- Sobel: blur ring -> sumX, sumY -> which groups of 4 pixels may hold a candidate -> NMS batches ------------------------
  // EDGE: some of the block's rows or columns are not this run's (the first block's rows above r0, rows from rend on, columns
  // from W on)
  auto sobel_phase = [&](auto edge_c, int R0) {
    constexpr bool EDGE = decltype(edge_c)::value;
    const bool row_ok = R0 + q >= r0 && R0 + q < rend;
    const v16i zero16 = {};
    const u32 idl = (u32)lane;
    // (the blur stage's order -- the next group's MFMAs between the parts of this group's vector code -- needs a second pair of
    //  sums alive across the batches here: 17 to 42 registers spilled, 2.8 ms instead of 1.8; with the A operands fetched again
    //  per stage to make room: no spills, 2.06 ms.  profiles/r04/mx_experiments.md)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      v4i B[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) B[i] = lds128(rowoff[1 + i] + bbase + lcS + (u32)(64 * pp));
      v16i aX = mfma8(A[3], B[0], zero16);
      v16i aY = mfma8(A[5], B[0], zero16);
      aX = mfma8(A[4], B[1], aX);
      aY = mfma8(A[6], B[2], aY);
      aX = mfma8(A[3], B[2], aX);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // S2 = sumX^2 + sumY^2 (the reference's float gradient is a strictly increasing function of it, k_front8): the largest
        // of the group's four against the low threshold -- the batch decides the rest (their sum, 3 instructions cheaper,
        // queued enough groups for nothing to cost 2 % more in the batches)
        int T = mad24(aY[4 * g], aY[4 * g], mul24(aX[4 * g], aX[4 * g]));
#pragma unroll
        for (int j = 1; j < 4; ++j) T = max(T, mad24(aY[4 * g + j], aY[4 * g + j], mul24(aX[4 * g + j], aX[4 * g + j])));
        u32 thr = g == 3 ? thr3 : (pp == 3 && g >= 1) ? thr7 : a_lo0;
        if constexpr (EDGE) {
          const int col0 = s0 + 28 * (2 * pp + par) + 16 * kh + 4 * g;
          thr = (row_ok && col0 < W) ? thr : 0xFFFFFFFFu;
        }
        const u64 mk = __ballot((u32)T >= thr);
        const u32 pos = ((u32)(qhead + qcount) + mbcnt64(mk)) & (u32)(MX_NQ - 1);
        lds16(lane_sel(mk, xbase + 2u * pos, dump)) = (unsigned short)(idl | ((u32)g << 6) | ((u32)pp << 8));
        qcount += __popcll(mk);
      }
      if (MX_ABL & 1) { qhead = (qhead + qcount) & (MX_NQ - 1); qcount = 0; }
      while (qcount >= 64) nms_batch(64, R0);
    }
    while (qcount > 0) nms_batch(min(qcount, 64), R0);
  };

  // ---- the run: blocks of output rows R0 .. R0+15, R0 = r0 - 4 + 16 b ------------------------------------------------------
  const int nblocks = (rend - r0 + MX_LAG + MX_ROWS - 1) / MX_ROWS;
  // zero stores of the map rows: the lane's byte offset from (row 2 k, column s0) and whether its group of 8 columns exists
  const u32 zoff = (u32)(lane >> 5) * (PROV ? p.prov_pitch : 0u) + 8u * (u32)(lane & 31);
  const bool zok = (lane & 31) < 27 && s0 + 8 * (lane & 31) < W;
  u32 xr[MX_ROWS];
  {  // input rows R0 .. R0+3 of the first block
    u32 x0[MX_LAG];
#pragma unroll
    for (int k = 0; k < MX_LAG; ++k) x0[k] = load_row(r0 - MX_LAG + k);
#pragma unroll
    for (int k = 0; k < MX_ROWS; ++k) xr[k] = load_row(r0 + k);
#pragma unroll
    for (int k = 0; k < MX_LAG; ++k) {
      const int row = r0 - MX_LAG + k;
      const u32 m = (u32)row < (u32)H ? cmask : 0u;
      lds32(wr1 + (u32)(k * MX_PITCH)) = (x0[k] & m) ^ MX_PAD;
    }
  }
#pragma nounroll
  for (int b = 0; b < nblocks; ++b) {
    const int R0 = r0 - MX_LAG + MX_ROWS * b;
    // new input rows R0+4 .. R0+19
#pragma unroll
    for (int k = 0; k < MX_ROWS; ++k) {
      const int row = R0 + MX_LAG + k;
      const u32 m = (u32)row < (u32)H ? cmask : 0u;  // rows above / below the image are zero padding (cannyEdgeD.cu:91-98)
      u32 sl = (u32)sb + (u32)(MX_LAG + k);
      sl = sl >= (u32)MX_RING ? sl - (u32)MX_RING : sl;
      lds32(wr1 + sl * (u32)MX_PITCH) = (xr[k] & m) ^ MX_PAD;
    }
    if (b + 1 < nblocks) {
#pragma unroll
      for (int k = 0; k < MX_ROWS; ++k) xr[k] = load_row(R0 + MX_LAG + MX_ROWS + k);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) rowoff[k] = slot_off((u32)(q + k));
    wave_lds_sync();
    fqn = 0;
    const bool edge_rows = R0 + 2 < 0 || R0 + 17 >= H;
    if (!(MX_ABL & 16)) {
      if (edge_rows || edge_cols) blur_phase(std::true_type{}, R0);
      else blur_phase(std::false_type{}, R0);
    }
    wave_lds_sync();
    if (!(MX_ABL & 2)) fixup();
    wave_lds_sync();
    if (p.dbg_blur) {  // diagnostics (HC_OPT_DEBUG_TAPS): the fixed-up blur rows of this run
      for (int k = 0; k < MX_ROWS; ++k) {
        const int br = R0 + 2 + k;
        if (br < r0 || br >= rend) continue;
        for (int c = lane; c < MX_STRIP_W && s0 + c < W; c += 64) {
          const u32 y = (u32)c + 2u;
          p.dbg_blur[(size_t)frame * p.dbg_fs + (size_t)br * p.dbg_pitch + (u32)(s0 + c)] = smem[bbase + slot_off(4u + (u32)k) + (u32)MX_SEG0 + 32u * (y / 28u) + y % 28u] ^ 0x80u;
        }
      }
    }
    // zero the plane tiles; the map rows of the block are stored as zeros now, the few groups with strong pixels are
    // overwritten by the batches (stores of one wave to one address keep their order)
    lds128(tbase + 16u * (u32)lane) = v4i{ 0, 0, 0, 0 };
    if constexpr (PROV && !(MX_ABL & 4) && !(MX_ABL & 64)) {
      // two rows per instruction: lanes 0..26 / 32..58 hold the 27 groups of 8 columns of row 2 k / 2 k + 1 -- 216 contiguous
      // bytes per row.  (4 lanes per row, every instruction touching all 16 rows with 32 bytes each, cost 0.26 ms of the
      // pipelined 2.4: profiles/r04/mx_ablation.txt)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int row = R0 + 2 * k + (lane >> 5);
        if (zok && row >= r0 && row < rend)
          *reinterpret_cast<gu32x2 *>(prov_frame + (u32)(R0 + 2 * k) * p.prov_pitch + (u32)s0 + zoff) = u32x2{ 0u, 0u };
      }
    }
    wave_lds_sync();
    const bool edge_out = R0 < r0 || R0 + MX_ROWS > rend || s0 + MX_STRIP_W > W;
    if (!(MX_ABL & 8)) {
      if (edge_out) sobel_phase(std::true_type{}, R0);
      else sobel_phase(std::false_type{}, R0);
    }
    wave_lds_sync();
    // the block's plane bytes: 16 rows x 27 bytes per plane
    if (!(MX_ABL & 4)) {
      auto put32 = [&](u32 r, u32 o) {
        const int row = R0 + (int)r;
        if (row >= r0 && row < rend && pg0 + o < plane_pitch) {
          const u32 go = (u32)row * plane_pitch + pg0 + o;
          *reinterpret_cast<u32 *>(splane + go) = lds32(tbase + r * 32u + o);
          *reinterpret_cast<u32 *>(cplane + go) = lds32(tbase + 512u + r * 32u + o);
        }
      };
      put32(pdA, pdAo);
      if (lane < 32) put32(pdB, pdBo);
      const int row = R0 + (int)pbR;
      if (lane < 48 && row >= r0 && row < rend && pg0 + pbO < plane_pitch) {
        const u32 go = (u32)row * plane_pitch + pg0 + pbO;
        splane[go] = smem[tbase + pbR * 32u + pbO];
        cplane[go] = smem[tbase + 512u + pbR * 32u + pbO];
      }
    }
    wave_lds_sync();  // (the next block's fix-up queue and input rows overwrite the queue and the dump dwords)
    sb = sb + MX_ROWS >= MX_RING ? sb + MX_ROWS - MX_RING : sb + MX_ROWS;
  }
}

size_t front_mx_lds_bytes(int wpb) { return (size_t)wpb * MX_WAVE_BYTES; }

// Mode R, one-channel frames: strips of 216 columns, runs of p.run_rows rows (16 n - 4 rows cost n blocks)
hipError_t launch_front_mx(const FrontParams &p, hipStream_t s)
{
  if (p.bgr || p.run_rows < 1 || (long)p.nchunks * p.run_rows < p.H) return hipErrorInvalidValue;
  if (p.nstrips != front_mx_strips(p.W) || (long)p.total_items != (long)p.nframes * p.nstrips * p.nchunks) return hipErrorInvalidValue;
  const size_t w4 = ((size_t)p.W + 3) / 4 * 4;
  if ((unsigned long long)p.H * p.in_pitch >= (1ull << 32) || p.in_pitch < w4) return hipErrorInvalidValue;
  if (p.prov_out && (p.W % 8 != 0)) return hipErrorInvalidValue;
  if (p.dbg_blur && p.dbg_pitch < (u32)p.W) return hipErrorInvalidValue;
  const int wpb = p.one_wave ? 1 : 4;
  const dim3 grid((unsigned)((p.total_items + wpb - 1) / wpb)), block(64 * wpb);
  const size_t lds = front_mx_lds_bytes(wpb);
  if (p.prov_out) {
    if (wpb == 1) hipLaunchKernelGGL((k_front_mx<true, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_front_mx<true, 4>), grid, block, lds, s, p);
  } else {
    if (wpb == 1) hipLaunchKernelGGL((k_front_mx<false, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_front_mx<false, 4>), grid, block, lds, s, p);
  }
  return hipGetLastError();
}

}  // namespace hc
