// Issue rate of a few more VALU ops on gfx950 (4 waves per SIMD, 8 independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(S) asm volatile(S(0) "\n" S(1) "\n" S(2) "\n" S(3) "\n" S(4) "\n" S(5) "\n" S(6) "\n" S(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m) : "vcc", "s20", "s21");
#define REP8(x) x x x x x x x x
#define S_MAD16(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %8"
#define S_MAD16H(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %8 op_sel:[1,1,0,0]"
#define S_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %8"
#define S_DOT2(i) "v_dot2_i32_i16 %" #i ", %" #i ", %8, 0"
#define S_CMPS(i) "v_cmp_le_u32_e64 s[20:21], %" #i ", %8"
#define S_CMPV(i) "v_cmp_le_u32_e32 vcc, %" #i ", %8"
#define S_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9"
#define S_MAX(i) "v_max_u32 %" #i ", %" #i ", %8"
#define S_ADDC(i) "v_addc_co_u32_e64 %" #i ", s[20:21], %" #i ", %" #i ", %9"
#define S_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", 2, %8 op_sel_hi:[1,0,1]"
#define S_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define S_SUBREV(i) "v_subrev_u32 %" #i ", %" #i ", %8"
#define S_LSHLADD64(i) "v_lshlrev_b32 %" #i ", 1, %" #i
template <int KIND>
__global__ void k(unsigned *out, int iters, unsigned long long m)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = blockIdx.x | 1;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP8(OP8(S_MAD16)) }
    if (KIND == 1) { REP8(OP8(S_MAD16H)) }
    if (KIND == 2) { REP8(OP8(S_MAD24)) }
    if (KIND == 3) { REP8(OP8(S_DOT2)) }
    if (KIND == 4) { REP8(OP8(S_CMPS)) }
    if (KIND == 5) { REP8(OP8(S_CMPV)) }
    if (KIND == 6) { REP8(OP8(S_CNDS)) }
    if (KIND == 7) { REP8(OP8(S_MAX)) }
    if (KIND == 8) { REP8(OP8(S_ADDC)) }
    if (KIND == 9) { REP8(OP8(S_PKMAD)) }
    if (KIND == 10) { REP8(OP8(S_DPP)) }
    if (KIND == 11) { REP8(OP8(S_SUBREV)) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int KIND>
void run(const char *name, unsigned *d)
{
  const int iters = 2000, wps = 4;
  const int blocks = 256 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 0x5555555555555555ull);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x5555555555555555ull);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 64 * wps;
  printf("%-22s %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main()
{
  unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_mad_i32_i16", d);
  run<1>("v_mad_i32_i16 op_sel", d);
  run<2>("v_mad_i32_i24", d);
  run<3>("v_dot2_i32_i16", d);
  run<4>("v_cmp e64 -> sgpr", d);
  run<5>("v_cmp e32 -> vcc", d);
  run<6>("v_cndmask e64 sgpr", d);
  run<7>("v_max_u32", d);
  run<8>("v_addc_co e64", d);
  run<9>("v_pk_mad_u16", d);
  run<10>("v_mov_b32_dpp", d);
  run<11>("v_subrev_u32", d);
  return 0;
}
