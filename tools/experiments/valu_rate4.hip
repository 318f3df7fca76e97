// Issue rate of the byte-SAD family, the 64-bit forms, SDWA / DPP variants of the fast ops, f32 VOP2 ops, and of MIXED
// streams (does a 2.5-cycle op hide behind a 4.4-cycle one?) on gfx950: 4 waves per SIMD, 8 independent chains per wave.
// Round 3 question (VERDICT lead a): is v_qsad_pk_u16_u8 / v_mqsad_pk_u16_u8 cheap enough to replace phase 1's plumbing?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define REP8(x) x x x x x x x x
// 32-bit chains a0..a7, operand b (VGPR), m (SGPR pair)
#define OP8(S) asm volatile(S(0) "\n" S(1) "\n" S(2) "\n" S(3) "\n" S(4) "\n" S(5) "\n" S(6) "\n" S(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m) : "vcc", "s20", "s21");
// 64-bit chains q0..q7 (VGPR pairs), 64-bit operand w, 32-bit operand b
#define OQ8(S) asm volatile(S(0) "\n" S(1) "\n" S(2) "\n" S(3) "\n" S(4) "\n" S(5) "\n" S(6) "\n" S(7) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(w), "v"(b) : "vcc");
// two interleaved streams: chains a0..a3 take op A, a4..a7 op B
#define OM8(SA, SB) asm volatile(SA(0) "\n" SB(4) "\n" SA(1) "\n" SB(5) "\n" SA(2) "\n" SB(6) "\n" SA(3) "\n" SB(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m) : "vcc", "s20", "s21");

#define S_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %8"
#define S_SADHI(i) "v_sad_hi_u8 %" #i ", %" #i ", %8, %8"
#define S_SAD16(i) "v_sad_u16 %" #i ", %" #i ", %8, %8"
#define S_MSAD(i) "v_msad_u8 %" #i ", %" #i ", %8, %8"
#define S_LERP(i) "v_lerp_u8 %" #i ", %" #i ", %8, %8"
#define S_ADDF(i) "v_add_f32_e32 %" #i ", %" #i ", %8"
#define S_MULF(i) "v_mul_f32_e32 %" #i ", %" #i ", %8"
#define S_FMAC(i) "v_fmac_f32_e32 %" #i ", %8, %8"
#define S_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_1"
#define S_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define S_ADDDPPROW(i) "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define S_ADDLIT(i) "v_add_u32_e32 %" #i ", 0x12345, %" #i
#define S_ANDLIT(i) "v_and_b32_e32 %" #i ", 0x00ff00ff, %" #i
#define S_SUB16(i) "v_sub_u16_e32 %" #i ", %" #i ", %8"
#define S_MULLO16(i) "v_mul_lo_u16_e32 %" #i ", %" #i ", %8"
#define S_LSHR16(i) "v_lshrrev_b16_e32 %" #i ", 1, %" #i
#define S_MAX16(i) "v_max_u16_e32 %" #i ", %" #i ", %8"
#define S_ASHR(i) "v_ashrrev_i32_e32 %" #i ", 1, %" #i
#define S_NOT(i) "v_not_b32_e32 %" #i ", %" #i
#define S_XNOR(i) "v_xnor_b32_e32 %" #i ", %" #i ", %8"
#define S_ADDCO(i) "v_add_co_u32_e32 %" #i ", vcc, %" #i ", %8"
#define S_MUL24SDWA(i) "v_mul_u32_u24_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
#define S_MUL24(i) "v_mul_u32_u24_e32 %" #i ", %" #i ", %8"
#define S_MADU32U16(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %8"
#define S_MADU16(i) "v_mad_u16 %" #i ", %" #i ", %8, %8"
#define S_SATPK(i) "v_sat_pk_u8_i16_e32 %" #i ", %" #i
#define S_CVTPKU8(i) "v_cvt_pk_u8_f32 %" #i ", %" #i ", %8, %8"
#define S_CVTPKU16(i) "v_cvt_pk_u16_u32 %" #i ", %" #i ", %8"
#define S_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %8"
#define S_ADD(i) "v_add_u32_e32 %" #i ", %" #i ", %8"
#define S_AND(i) "v_and_b32_e32 %" #i ", %" #i ", %8"
#define S_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", 2, %8 op_sel_hi:[1,0,1]"
#define S_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %8"
#define S_DOT2(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, 0"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8"
#define S_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8"
#define S_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define S_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %8"
#define S_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8"
#define S_PKSUBI(i) "v_pk_sub_i16 %" #i ", %" #i ", %8"
#define S_PKMAXI(i) "v_pk_max_i16 %" #i ", %" #i ", %8"
#define S_SUBSDWA(i) "v_sub_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
#define S_LSHRSDWA(i) "v_lshrrev_b32_sdwa %" #i ", %8, %" #i " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
#define S_SNOP(i) "s_nop 0"
#define S_SADD(i) "s_add_u32 s20, s20, 1"

#define Q_QSAD(i) "v_qsad_pk_u16_u8 %" #i ", %" #i ", %9, %8"
#define Q_MQSAD(i) "v_mqsad_pk_u16_u8 %" #i ", %" #i ", %9, %8"
#define Q_LSHL64(i) "v_lshlrev_b64 %" #i ", 1, %" #i
#define Q_LSHR64(i) "v_lshrrev_b64 %" #i ", 1, %" #i
#define Q_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %9, %9, %" #i
#define Q_PKMOV(i) "v_pk_mov_b32 %" #i ", %" #i ", %8 op_sel:[1,0]"
#define Q_MOV64(i) "v_mov_b64 %" #i ", %8"
#define Q_PKADDF(i) "v_pk_add_f32 %" #i ", %" #i ", %8"
#define Q_PKMULF(i) "v_pk_mul_f32 %" #i ", %" #i ", %8"
#define Q_PKFMAF(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8"

template <int KIND>
__global__ void k(unsigned *out, int iters, u64 m)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = blockIdx.x | 1;
  u64 q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7, w = ((u64)b << 32) | a0;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP8(OP8(S_SAD)) }
    if (KIND == 1) { REP8(OP8(S_SADHI)) }
    if (KIND == 2) { REP8(OP8(S_SAD16)) }
    if (KIND == 3) { REP8(OP8(S_MSAD)) }
    if (KIND == 4) { REP8(OP8(S_LERP)) }
    if (KIND == 5) { REP8(OP8(S_ADDF)) }
    if (KIND == 6) { REP8(OP8(S_MULF)) }
    if (KIND == 7) { REP8(OP8(S_FMAC)) }
    if (KIND == 8) { REP8(OP8(S_ADDSDWA)) }
    if (KIND == 9) { REP8(OP8(S_ADDDPP)) }
    if (KIND == 10) { REP8(OP8(S_ADDDPPROW)) }
    if (KIND == 11) { REP8(OP8(S_ADDLIT)) }
    if (KIND == 12) { REP8(OP8(S_ANDLIT)) }
    if (KIND == 13) { REP8(OP8(S_SUB16)) }
    if (KIND == 14) { REP8(OP8(S_MULLO16)) }
    if (KIND == 15) { REP8(OP8(S_LSHR16)) }
    if (KIND == 16) { REP8(OP8(S_MAX16)) }
    if (KIND == 17) { REP8(OP8(S_ASHR)) }
    if (KIND == 18) { REP8(OP8(S_NOT)) }
    if (KIND == 19) { REP8(OP8(S_XNOR)) }
    if (KIND == 20) { REP8(OP8(S_ADDCO)) }
    if (KIND == 21) { REP8(OP8(S_MUL24SDWA)) }
    if (KIND == 22) { REP8(OP8(S_MUL24)) }
    if (KIND == 23) { REP8(OP8(S_MADU32U16)) }
    if (KIND == 24) { REP8(OP8(S_MADU16)) }
    if (KIND == 25) { REP8(OP8(S_SATPK)) }
    if (KIND == 26) { REP8(OP8(S_CVTPKU8)) }
    if (KIND == 27) { REP8(OP8(S_CVTPKU16)) }
    if (KIND == 28) { REP8(OP8(S_BFI)) }
    if (KIND == 29) { REP8(OP8(S_SUBSDWA)) }
    if (KIND == 30) { REP8(OP8(S_LSHRSDWA)) }
    if (KIND == 31) { REP8(OP8(S_PKMAXI)) }
    // 64-bit destination forms
    if (KIND == 40) { REP8(OQ8(Q_QSAD)) }
    if (KIND == 41) { REP8(OQ8(Q_MQSAD)) }
    if (KIND == 42) { REP8(OQ8(Q_LSHL64)) }
    if (KIND == 43) { REP8(OQ8(Q_LSHR64)) }
    if (KIND == 44) { REP8(OQ8(Q_MAD64)) }
    if (KIND == 45) { REP8(OQ8(Q_PKMOV)) }
    if (KIND == 46) { REP8(OQ8(Q_MOV64)) }
    if (KIND == 47) { REP8(OQ8(Q_PKADDF)) }
    if (KIND == 48) { REP8(OQ8(Q_PKMULF)) }
    if (KIND == 49) { REP8(OQ8(Q_PKFMAF)) }
    // mixed streams, 4 + 4 per group of 8: time per instruction vs the mean of the two pure rates
    if (KIND == 60) { REP8(OM8(S_ADD, S_PKMAD)) }
    if (KIND == 61) { REP8(OM8(S_ADD, S_PERM)) }
    if (KIND == 62) { REP8(OM8(S_ADD, S_DOT2)) }
    if (KIND == 63) { REP8(OM8(S_AND, S_ADD)) }
    if (KIND == 64) { REP8(OM8(S_PKMAD, S_PERM)) }
    if (KIND == 65) { REP8(OM8(S_ADD, S_SNOP)) }
    if (KIND == 66) { REP8(OM8(S_PKMAD, S_SADD)) }
    if (KIND == 67) { REP8(OM8(S_ADD, S_SADD)) }
    if (KIND == 68) { REP8(OM8(S_ADD, S_LSHLADD)) }
    if (KIND == 69) { REP8(OM8(S_ADD, S_MOVDPP)) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (unsigned)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32);
}
template <int KIND>
void run(const char *name, unsigned *d, int wps = 4)
{
  const int iters = 2000;
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
  printf("%-34s wps %d  %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms * 1e6 / n, ms * 1e6 / n * 2.4);
  hipEventDestroy(e0); hipEventDestroy(e1);
}
int main()
{
  unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_sad_u8", d);
  run<1>("v_sad_hi_u8", d);
  run<2>("v_sad_u16", d);
  run<3>("v_msad_u8", d);
  run<4>("v_lerp_u8", d);
  run<5>("v_add_f32_e32", d);
  run<6>("v_mul_f32_e32", d);
  run<7>("v_fmac_f32_e32", d);
  run<8>("v_add_u32_sdwa", d);
  run<9>("v_add_u32_dpp wave_shr", d);
  run<10>("v_add_u32_dpp row_shr", d);
  run<11>("v_add_u32 literal", d);
  run<12>("v_and_b32 literal", d);
  run<13>("v_sub_u16", d);
  run<14>("v_mul_lo_u16", d);
  run<15>("v_lshrrev_b16", d);
  run<16>("v_max_u16", d);
  run<17>("v_ashrrev_i32", d);
  run<18>("v_not_b32", d);
  run<19>("v_xnor_b32", d);
  run<20>("v_add_co_u32", d);
  run<21>("v_mul_u32_u24_sdwa", d);
  run<22>("v_mul_u32_u24", d);
  run<23>("v_mad_u32_u16", d);
  run<24>("v_mad_u16", d);
  run<25>("v_sat_pk_u8_i16", d);
  run<26>("v_cvt_pk_u8_f32", d);
  run<27>("v_cvt_pk_u16_u32", d);
  run<28>("v_bfi_b32", d);
  run<29>("v_sub_u32_sdwa", d);
  run<30>("v_lshrrev_b32_sdwa", d);
  run<31>("v_pk_max_i16", d);
  run<40>("v_qsad_pk_u16_u8", d);
  run<41>("v_mqsad_pk_u16_u8", d);
  run<42>("v_lshlrev_b64", d);
  run<43>("v_lshrrev_b64", d);
  run<44>("v_mad_u64_u32", d);
  run<45>("v_pk_mov_b32", d);
  run<46>("v_mov_b64", d);
  run<47>("v_pk_add_f32", d);
  run<48>("v_pk_mul_f32", d);
  run<49>("v_pk_fma_f32", d);
  run<60>("mix v_add_u32 / v_pk_mad_u16", d);
  run<61>("mix v_add_u32 / v_perm_b32", d);
  run<62>("mix v_add_u32 / v_dot2_u32_u16", d);
  run<63>("mix v_and_b32 / v_add_u32", d);
  run<64>("mix v_pk_mad_u16 / v_perm_b32", d);
  run<65>("mix v_add_u32 / s_nop", d);
  run<66>("mix v_pk_mad_u16 / s_add_u32", d);
  run<67>("mix v_add_u32 / s_add_u32", d);
  run<68>("mix v_add_u32 / v_lshl_add_u32", d);
  run<69>("mix v_add_u32 / v_mov_dpp", d);
  // the same mixes with 2 and 3 waves per SIMD (k_front8 runs three)
  run<60>("mix v_add_u32 / v_pk_mad_u16", d, 2);
  run<60>("mix v_add_u32 / v_pk_mad_u16", d, 3);
  run<40>("v_qsad_pk_u16_u8", d, 2);
  run<41>("v_mqsad_pk_u16_u8", d, 3);
  return 0;
}
