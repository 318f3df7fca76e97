// canny_device.h -- device-side helpers shared by the kernel files of libhipcanny.so (gfx950, wave64): packed-u16 math,
// DPP lane shifts, ballot/mbcnt compaction, the Gaussian coefficients as literals, the XCD-aware work-item order.
#pragma once
#include "canny_common.h"

namespace hc {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// The same 25 values as compile-time constants: K * (1 / 159.0f), both roundings in binary32 (constant
// folding is IEEE round-to-nearest).  check_gauss_coeffs() refuses a host table that differs.
struct GaussLiterals {
  float v[25];
  constexpr GaussLiterals() : v{}
  {
    constexpr int K[25] = { 2, 4, 5, 4, 2, 4, 9, 12, 9, 4, 5, 12, 15, 12, 5, 4, 9, 12, 9, 4, 2, 4, 5, 4, 2 };
    constexpr float r = 1.0f / 159.0f;
    for (int i = 0; i < 25; ++i) v[i] = (float)K[i] * r;
  }
};
constexpr GaussLiterals GKC{};

// literal reference chain for one pixel (cannyEdgeD.cu:102-115): 25 fused multiply-adds from 0.0f in
// r-major / c-minor order, truncation.  Only used for the rare undecidable pixels.
// IN: how a pixel is read -- 0 mono plane, 1 BGR -> grey (stage 0 fused), 2 channel `ch` of interleaved 3-channel data
template <int IN = 0>
static __device__ __forceinline__ u32 gauss_chain_px(const uint8_t *frame, size_t pitch, int W, int H, int row, int col, int ch = 0)
{
  // the coefficients are literals of the instruction stream (GKC): as __constant__ loads they were
  // hoisted to the kernel entry and pinned 25 SGPRs across the hot loops
  float f = 0.0f;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int rr = row - 2 + r;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int cc = col - 2 + c;
      float px = 0.0f;
      if (rr >= 0 && rr < H && cc >= 0 && cc < W) {
        if (IN == 1) {  // stage 0 on the fly: (b*7 + g*38 + r*19) >> 6 (cannyEdgeD.cu:17-19,67)
          const uint8_t *q = frame + (size_t)rr * pitch + 3 * (size_t)cc;
          px = (float)((q[0] * 7 + q[1] * 38 + q[2] * 19) >> 6);
        } else if (IN == 2) px = (float)frame[(size_t)rr * pitch + 3 * (size_t)cc + ch];
        else px = (float)frame[(size_t)rr * pitch + cc];
      }
      f = __builtin_fmaf(GKC.v[r * 5 + c], px, f);
    }
  }
  return (u32)(int)f;
}

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
// single packed-math instructions the compiler would otherwise expand into shift+add pairs, or emit in
// the accumulate form (v_dot2c) that needs an extra v_mov 0
static __device__ __forceinline__ u32 pk_mad2(u32 a, u32 c) { u32 d; asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c)); return d; }
static __device__ __forceinline__ u32 pk_mul5(u32 a) { u32 d; asm("v_pk_mul_lo_u16 %0, %1, 5 op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }
// 16 x 16 -> 32 bit signed multiply-add on a chosen half (HA / HB: 0 = low, 1 = high) of each packed operand
template <int HA, int HB>
static __device__ __forceinline__ int mad16(u32 a, u32 b, int c)
{
  int d;
  if (HA == 0 && HB == 0) asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  else if (HA == 1 && HB == 1) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  else if (HA == 1) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  else asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int HA, int HB>
static __device__ __forceinline__ int mul16(u32 a, u32 b)
{
  int d;
  if (HA == 0 && HB == 0) asm("v_mad_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  else if (HA == 1 && HB == 1) asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b));
  else if (HA == 1) asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b));
  else asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[0,1,0,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
static __device__ __forceinline__ int sdot2_0(u32 a, u32 b) { int d; asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b)); return d; }
// v*2 + (bit of `mask` for this lane): shifts one ballot mask into per-lane words (carry-in form of v_addc)
static __device__ __forceinline__ u32 shift_in(u32 v, u64 mask)
{
  u32 d; u64 co;
  asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(d), "=s"(co) : "v"(v), "s"(mask));
  return d;
}
static __device__ __forceinline__ void wave_lds_sync()
{
  // wave-private LDS hand-off between lanes of ONE wave: DS ops of a wave execute in order, this
  // only stops the compiler from moving accesses across the point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// lane in `mask` ? a : b as ONE v_cndmask (written as a select, the compiler may turn it into an exec-masked region,
// which ends the scheduling region of otherwise straight-line code)
static __device__ __forceinline__ u32 lane_sel(u64 mask, u32 a, u32 b)
{
  u32 d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(mask));
  return d;
}
// wave-uniform cond ? a : b on pointers as two s_cselect (the compiler tends to branch around the address arithmetic of `a`)
static __device__ __forceinline__ uint8_t *uniform_sel(bool cond, uint8_t *a, uint8_t *b)
{
  u64 d;
  asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b64 %0, %2, %3" : "=s"(d) : "s"(__builtin_amdgcn_readfirstlane((int)cond)), "s"((u64)a), "s"((u64)b) : "scc");
  return (uint8_t *)d;
}
// tell the compiler a value is wave-uniform (keeps masks and row indices in SGPRs, control flow scalar)
static __device__ __forceinline__ u64 uniform64(u64 v)
{
  return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
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


// bit k of nib -> byte k = 0xFF: (nib * 0x00204081) & 0x01010101 spreads the 4 bits to byte positions
static __device__ __forceinline__ u32 nibble_to_bytes(u32 nib)
{
  const u32 x = (nib * 0x00204081u) & 0x01010101u;
  return (x << 8) - x;
}

}  // namespace hc
