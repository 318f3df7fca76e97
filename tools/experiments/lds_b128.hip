// lds_b128.hip -- can ds_read_b128 take an address that is only 4- (or 2-) byte aligned on gfx950, and at what price?
// The matrix-pipe front kernel reads its MFMA B operands (16 contiguous row bytes per lane) from an LDS ring at tile
// offsets 28 t: 0, 12, 8, 4 (mod 16).  Rows are 240 bytes apart (15 x 16: conflict-free for aligned b128).
// Prints correctness of the misaligned forms and ns per read instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_check(u32x4 *out, int off)
{
  __shared__ __attribute__((aligned(16))) unsigned char s[36 * 240 + 64];
  for (int i = threadIdx.x; i < 36 * 240 + 64; i += 64) s[i] = (unsigned char)(i * 7 + (i >> 8));
  __syncthreads();
  const u32 addr = (u32)(size_t)(s) + (threadIdx.x & 31) * 240 + (threadIdx.x >> 5) * 16 + off;
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x] = v;
}

// MODE 0: ds_read_b128 at offset `off`; 1: two ds_read2_b32; 2: ds_read_b128 aligned but pitch 256 (bank conflicts, for scale)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(u32 *out, int iters, int off, int pitch)
{
  __shared__ __attribute__((aligned(16))) unsigned char s[40960];
  for (int i = threadIdx.x; i < 40960; i += 256) s[i] = (unsigned char)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 addr = (u32)(size_t)(s) + w * 10240 + (lane & 31) * pitch + (lane >> 5) * 16 + off;
  u32x4 acc = { 0, 0, 0, 0 };
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      u32x4 v;
      if (MODE == 1) {
        unsigned long long lo, hi;
        asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3" : "=v"(lo), "=v"(hi) : "v"(addr) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v = u32x4{ (u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32) };
      } else {
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += v;
      addr ^= (u32)(u & 1) * 32u;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345) out[0] = 1;
}

template <int MODE>
static void rate(const char *name, int off, int pitch, int cus)
{
  u32 *d;
  CK(hipMalloc(&d, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int wpc = 4; wpc <= 8; wpc += 4) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_rate<MODE>, dim3(cus * wpc / 4), dim3(256), 0, 0, d, iters, off, pitch);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / (8.0 * iters * wpc);
    printf("%-44s off %2d pitch %3d  %d waves/CU  %6.2f ns = %5.1f cycles per 1 KiB read per CU (%.0f B/clk/CU)\n", name, off, pitch, wpc, ns, ns * 2.4, 1024.0 / (ns * 2.4));
  }
}

int main()
{
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  u32x4 *d;
  CK(hipMalloc(&d, 64 * 16));
  for (int off : { 0, 4, 8, 12, 2, 6, 1 }) {
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d, off);
    std::vector<unsigned char> h(1024);
    CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 16; ++j) {
        const int i = (l & 31) * 240 + (l >> 5) * 16 + off + j;
        bad += h[l * 16 + j] != (unsigned char)(i * 7 + (i >> 8));
      }
    printf("ds_read_b128 at address = 16-aligned + %2d: %s (%ld wrong bytes)\n", off, bad ? "WRONG" : "correct", bad);
  }
  rate<0>("ds_read_b128", 0, 240, cus);
  rate<0>("ds_read_b128", 4, 240, cus);
  rate<0>("ds_read_b128", 8, 240, cus);
  rate<0>("ds_read_b128", 12, 240, cus);
  rate<0>("ds_read_b128", 2, 240, cus);
  rate<1>("2 x ds_read2_b32", 4, 240, cus);
  rate<1>("2 x ds_read2_b32", 0, 240, cus);
  rate<0>("ds_read_b128 (pitch 256: conflicts)", 0, 256, cus);
  rate<0>("ds_read_b128 (pitch 272)", 0, 272, cus);
  rate<0>("ds_read_b128 (pitch 272)", 12, 272, cus);
  return 0;
}
