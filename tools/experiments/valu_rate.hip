// Micro-benchmark: sustained issue rate of the VALU instruction classes k_front uses, per SIMD,
// at 1/2/4/8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(unsigned *out, int iters)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = blockIdx.x | 1;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP8(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 1) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 2) { REP8(asm volatile("v_dot2_i32_i16 %0, %0, %8, 0\n v_dot2_i32_i16 %1, %1, %8, 0\n v_dot2_i32_i16 %2, %2, %8, 0\n v_dot2_i32_i16 %3, %3, %8, 0\n v_dot2_i32_i16 %4, %4, %8, 0\n v_dot2_i32_i16 %5, %5, %8, 0\n v_dot2_i32_i16 %6, %6, %8, 0\n v_dot2_i32_i16 %7, %7, %8, 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 3) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 4) { REP8(asm volatile("v_pk_mad_u16 %0, %0, %8, %8\n v_pk_mad_u16 %1, %1, %8, %8\n v_pk_mad_u16 %2, %2, %8, %8\n v_pk_mad_u16 %3, %3, %8, %8\n v_pk_mad_u16 %4, %4, %8, %8\n v_pk_mad_u16 %5, %5, %8, %8\n v_pk_mad_u16 %6, %6, %8, %8\n v_pk_mad_u16 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 5) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (KIND == 6) { REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 7) { REP8(asm volatile("v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int KIND>
void run(const char *name, unsigned *d)
{
  const int iters = 2000;
  for (int wps : { 1, 2, 4, 8 }) {
    const int blocks = 256 * wps;  // 256-thread blocks: 4 waves = one per SIMD; wps blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * wps;
    printf("%-14s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
}
int main()
{
  unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_pk_add_u16", d); run<1>("v_and_b32", d); run<2>("v_dot2_i32_i16", d); run<3>("v_fma_f32", d);
  run<4>("v_pk_mad_u16", d); run<5>("v_cmp_lt_u32", d); run<6>("v_mov_dpp", d); run<7>("v_max_u32", d);
  return 0;
}
