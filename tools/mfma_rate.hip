// mfma_rate.hip -- round 4, step (a) of the matrix-pipe plan: what the i8 MFMA forms of gfx950 cost and what they leave of
// the vector pipe.  Prints
//   1. the operand layout of v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 (which (row, k) a lane's 16 bytes are), found
//      by comparing the instruction with a CPU product under two hypotheses;
//   2. cycles per MFMA per SIMD, back to back, independent and dependent accumulators, 1 / 2 / 3 waves per SIMD;
//   3. one wave's stream of 1 MFMA + n vector instructions (slow class: v_perm_b32; fast class: v_add_u32) per iteration:
//      cycles per iteration per SIMD against the same n instructions without the MFMA -- the model the front kernel needs
//      (does the MFMA hide behind the wave's own vector work, and what does it take from the issue port);
//   4. MFMA-only waves and VALU-only waves as partners on one SIMD (512-thread workgroups, waves 0-3 matrix, 4-7 vector).
// Timing: wall clock (hipEvents) over a launch that fills the chip, as tools/experiments/valu_rate*.hip do; cycles at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- 1. layout ---------------------------------------------------------------------------------------------------------
__global__ void k_layout32(const v4i *a, const v4i *b, v16i *d)
{
  v16i c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  d[threadIdx.x] = c;
}
__global__ void k_layout16(const v4i *a, const v4i *b, v4i *d)
{
  v4i c = {};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  d[threadIdx.x] = c;
}

// hypothesis h for the k index of byte j (0..15) held by lane-group g (lane / MN):
//   h = 0: k = KPL * g + j                      (16 contiguous k per lane)
//   h = 1: k = 8 * g + (j & 7) + (K / 2) * (j >> 3)   (two halves of 8, as two issues of the half-K form)
static int kidx(int h, int K, int groups, int g, int j)
{
  const int kpl = K / groups;  // 16
  if (h == 0) return kpl * g + j;
  return 8 * g + (j & 7) + (K / 2) * (j >> 3);
}

static void layout_probe()
{
  for (int form = 0; form < 2; ++form) {
    const int MN = form == 0 ? 32 : 16, K = form == 0 ? 32 : 64, groups = 64 / MN;
    std::vector<int8_t> A(MN * K), B(K * MN);
    srand(7 + form);
    for (auto &x : A) x = (int8_t)(rand() % 255 - 127);
    for (auto &x : B) x = (int8_t)(rand() % 255 - 127);
    std::vector<int> D(MN * MN);
    for (int m = 0; m < MN; ++m)
      for (int n = 0; n < MN; ++n) {
        int s = 0;
        for (int k = 0; k < K; ++k) s += (int)A[m * K + k] * (int)B[k * MN + n];
        D[m * MN + n] = s;
      }
    for (int h = 0; h < 2; ++h) {
      std::vector<int8_t> ra(64 * 16), rb(64 * 16);
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 16; ++j) {
          const int g = l / MN, i = l % MN, k = kidx(h, K, groups, g, j);
          ra[l * 16 + j] = A[i * K + k];
          rb[l * 16 + j] = B[k * MN + i];
        }
      void *da, *db, *dd;
      const size_t dbytes = form == 0 ? 64 * 64 : 64 * 16;
      CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, dbytes));
      CK(hipMemcpy(da, ra.data(), 1024, hipMemcpyHostToDevice));
      CK(hipMemcpy(db, rb.data(), 1024, hipMemcpyHostToDevice));
      if (form == 0) hipLaunchKernelGGL(k_layout32, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, (v16i *)dd);
      else hipLaunchKernelGGL(k_layout16, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, (v4i *)dd);
      std::vector<int> out(dbytes / 4);
      CK(hipMemcpy(out.data(), dd, dbytes, hipMemcpyDeviceToHost));
      // D layout: lane l, register v: n = l % MN; 32x32: m = 8 * (v / 4) + 4 * (l / 32) + v % 4; 16x16: m = 4 * (l / 16) + v
      long bad = 0;
      const int nv = form == 0 ? 16 : 4;
      for (int l = 0; l < 64; ++l)
        for (int v = 0; v < nv; ++v) {
          const int n = l % MN, m = form == 0 ? 8 * (v / 4) + 4 * (l / 32) + v % 4 : 4 * (l / 16) + v;
          bad += out[l * nv + v] != D[m * MN + n];
        }
      printf("layout %s  k-hypothesis %d (%s): %s (%ld of %d outputs differ)\n", form == 0 ? "v_mfma_i32_32x32x32_i8" : "v_mfma_i32_16x16x64_i8", h,
             h == 0 ? "byte j of lane-group g = k 16g+j" : "k = 8g + j%8 + (K/2)(j/8)", bad == 0 ? "MATCH" : "no", bad, MN * MN);
      CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dd));
    }
  }
  printf("  (D: lane l, register v holds n = l %% MN and, 32x32: m = 8 (v / 4) + 4 (l / 32) + v %% 4; 16x16: m = 4 (l / 16) + v)\n");
}

// ---- 2..4 rates --------------------------------------------------------------------------------------------------------
#define MF32(C) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
#define MF16(C) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
#define PERM(X) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(X) : "v"(s));
#define ADDU(X) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(X) : "v"(s));
#define MUL24(X) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(X) : "v"(s));
#define MAD24(X) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(X) : "v"(s));
#define CMPS(X) asm volatile("v_cmp_eq_u32_sdwa s[20:21], %0, %1 src0_sel:BYTE_2 src1_sel:DWORD" : : "v"(X), "v"(s) : "s20", "s21");
#define CMPG(X) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" : : "v"(X), "v"(s) : "s20", "s21");
#define SWAP(X, Y) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(X), "+v"(Y));
#define LSHR(X) asm volatile("v_lshrrev_b32_e32 %0, 24, %0" : "+v"(X));

// KIND: 0 = 32x32x32 two independent accumulators, 1 = 32x32x32 one accumulator (dependent), 2 = 16x16x64 four
// independent, 3 = 16x16x64 dependent
template <int KIND>
__global__ void k_mfma(int *out, int iters)
{
  v4i a = { (int)threadIdx.x, 1, 2, 3 }, b = { 4, 5, (int)blockIdx.x, 7 };
  v16i c0 = {}, c1 = {};
  v4i e0 = {}, e1 = {}, e2 = {}, e3 = {};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { MF32(c0) MF32(c1) MF32(c0) MF32(c1) MF32(c0) MF32(c1) MF32(c0) MF32(c1) }
    if (KIND == 1) { MF32(c0) MF32(c0) MF32(c0) MF32(c0) MF32(c0) MF32(c0) MF32(c0) MF32(c0) }
    if (KIND == 2) { MF16(e0) MF16(e1) MF16(e2) MF16(e3) MF16(e0) MF16(e1) MF16(e2) MF16(e3) }
    if (KIND == 3) { MF16(e0) MF16(e0) MF16(e0) MF16(e0) MF16(e0) MF16(e0) MF16(e0) MF16(e0) }
  }
  int r = 0;
  for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
  for (int j = 0; j < 4; ++j) r += e0[j] + e1[j] + e2[j] + e3[j];
  if (r == 0x12345678) out[0] = r;
}

// one stream: per iteration NM MFMAs (32x32x32, accumulators alternate) and NV vector instructions of class CLS
// (0 = v_perm_b32, 1 = v_add_u32, 2 = v_mul_u32_u24, 3 = v_cmp_eq_u32_sdwa, 4 = v_mad_i32_i24, 5 = v_cmp_gt_u32 e64,
//  6 = v_permlane32_swap, 7 = v_lshrrev_b32) on 8 independent chains
template <int NM, int NV, int CLS>
__global__ void k_mix(int *out, int iters)
{
  v4i a = { (int)threadIdx.x, 1, 2, 3 }, b = { 4, 5, (int)blockIdx.x, 7 };
  v16i c0 = {}, c1 = {};
  unsigned x[8], s = blockIdx.x | 1;
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // four MFMA groups per loop trip
      if (NM >= 1) { if (u & 1) { MF32(c1) } else { MF32(c0) } }
      if (NM >= 2) { if (u & 1) { MF32(c0) } else { MF32(c1) } }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (CLS == 0) { PERM(x[v & 7]) }
        if (CLS == 1) { ADDU(x[v & 7]) }
        if (CLS == 2) { MUL24(x[v & 7]) }
        if (CLS == 3) { CMPS(x[v & 7]) }
        if (CLS == 4) { MAD24(x[v & 7]) }
        if (CLS == 5) { CMPG(x[v & 7]) }
        if (CLS == 6) { SWAP(x[v & 7], x[(v + 4) & 7]) }
        if (CLS == 7) { LSHR(x[v & 7]) }
      }
    }
  }
  int r = 0;
  for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
  for (int j = 0; j < 8; ++j) r += (int)x[j];
  if (r == 0x12345678) out[0] = r;
}

// partners: 512-thread workgroups; waves 0..3 (one per SIMD) issue MFMAs back to back, waves 4..7 issue NV v_perm per trip.
// ROLE mask: bit 0 = the matrix waves work, bit 1 = the vector waves work (the others leave at once)
template <int ROLE>
__global__ __launch_bounds__(512) void k_pair(int *out, int iters)
{
  v4i a = { (int)threadIdx.x, 1, 2, 3 }, b = { 4, 5, (int)blockIdx.x, 7 };
  v16i c0 = {}, c1 = {};
  unsigned x[8], s = blockIdx.x | 1;
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
  const bool matrix = threadIdx.x < 256;
  if (matrix) {
    if (ROLE & 1)
      for (int i = 0; i < iters; ++i) { MF32(c0) MF32(c1) MF32(c0) MF32(c1) }
  } else {
    if (ROLE & 2)
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int v = 0; v < 32; ++v) { PERM(x[v & 7]) }
      }
  }
  int r = 0;
  for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
  for (int j = 0; j < 8; ++j) r += (int)x[j];
  if (r == 0x12345678) out[0] = r;
}

static int *g_out;
static int g_cus = 256;

template <typename F>
static double time_ms(F launch)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return best;
}

// per SIMD: ns (and cycles at 2.4 GHz) per unit, `units_per_wave` of them issued by each of `wps` waves of a SIMD
static void report(const char *name, int wps, double ms, double units_per_wave)
{
  const double ns = ms * 1e6 / (units_per_wave * wps);
  printf("%-64s wps %d  %7.2f ns = %6.1f cycles per unit per SIMD\n", name, wps, ns, ns * 2.4);
}

template <int KIND>
static void run_mfma(const char *name)
{
  const int iters = 20000;
  for (int wps = 1; wps <= 3; ++wps) {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<KIND>, dim3(g_cus * wps), dim3(256), 0, 0, g_out, iters); });
    report(name, wps, ms, 8.0 * iters);
  }
}

template <int NM, int NV, int CLS>
static void run_mix(const char *name)
{
  const int iters = 8000;
  for (int wps = 1; wps <= 3; ++wps) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_mix<NM, NV, CLS>), dim3(g_cus * wps), dim3(256), 0, 0, g_out, iters); });
    char buf[128];
    snprintf(buf, sizeof buf, "%s: %d MFMA + %2d vector per group", name, NM, NV);
    report(buf, wps, ms, 4.0 * iters);  // unit = one group
  }
}

int main()
{
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  g_cus = pr.multiProcessorCount;
  printf("# tools/mfma_rate.hip on %s (%s), %d CUs, clock %d MHz; cycles quoted at 2.4 GHz\n", pr.name, pr.gcnArchName, g_cus, pr.clockRate / 1000);
  CK(hipMalloc(&g_out, 64));
  layout_probe();

  printf("## back-to-back MFMAs (unit = one MFMA)\n");
  run_mfma<0>("v_mfma_i32_32x32x32_i8, two accumulators alternating");
  run_mfma<1>("v_mfma_i32_32x32x32_i8, one accumulator (dependent chain)");
  run_mfma<2>("v_mfma_i32_16x16x64_i8, four accumulators");
  run_mfma<3>("v_mfma_i32_16x16x64_i8, one accumulator (dependent chain)");

  printf("## one wave's stream: groups of MFMAs + vector instructions (unit = one group)\n");
  run_mix<0, 8, 0>("v_perm only");
  run_mix<0, 16, 0>("v_perm only");
  run_mix<0, 32, 0>("v_perm only");
  run_mix<1, 0, 0>("v_perm");
  run_mix<1, 2, 0>("v_perm");
  run_mix<1, 4, 0>("v_perm");
  run_mix<1, 6, 0>("v_perm");
  run_mix<1, 8, 0>("v_perm");
  run_mix<1, 12, 0>("v_perm");
  run_mix<1, 16, 0>("v_perm");
  run_mix<1, 24, 0>("v_perm");
  run_mix<1, 32, 0>("v_perm");
  run_mix<2, 32, 0>("v_perm");
  run_mix<0, 16, 1>("v_add_u32 only");
  run_mix<1, 8, 1>("v_add_u32");
  run_mix<1, 16, 1>("v_add_u32");
  run_mix<1, 32, 1>("v_add_u32");
  printf("## single vector instructions the C-layout epilogue would use (unit = group of 16)\n");
  run_mix<0, 16, 2>("v_mul_u32_u24 only");
  run_mix<0, 16, 3>("v_cmp_eq_u32_sdwa BYTE_2 only");
  run_mix<0, 16, 4>("v_mad_i32_i24 only");
  run_mix<0, 16, 5>("v_cmp_gt_u32 (VOP3, sgpr pair) only");
  run_mix<0, 16, 6>("v_permlane32_swap only");
  run_mix<0, 16, 7>("v_lshrrev_b32 only");
  run_mix<1, 16, 2>("v_mul_u32_u24");
  run_mix<1, 16, 4>("v_mad_i32_i24");

  printf("## partners on one SIMD: 512-thread workgroups, waves 0-3 MFMA back to back, waves 4-7 v_perm (one workgroup per CU)\n");
  {
    const int iters = 20000;
    const double m1 = time_ms([&] { hipLaunchKernelGGL(k_pair<1>, dim3(g_cus), dim3(512), 0, 0, g_out, iters); });
    const double m2 = time_ms([&] { hipLaunchKernelGGL(k_pair<2>, dim3(g_cus), dim3(512), 0, 0, g_out, iters); });
    const double m3 = time_ms([&] { hipLaunchKernelGGL(k_pair<3>, dim3(g_cus), dim3(512), 0, 0, g_out, iters); });
    printf("matrix waves alone: %.3f ms (%.1f cycles per MFMA)\n", m1, m1 * 1e6 * 2.4 / (4.0 * iters));
    printf("vector waves alone: %.3f ms (%.2f cycles per v_perm)\n", m2, m2 * 1e6 * 2.4 / (32.0 * iters));
    printf("both together:      %.3f ms (sum of the two alone %.3f, max %.3f) -> %s\n", m3, m1 + m2, m1 > m2 ? m1 : m2,
           m3 < 0.6 * (m1 + m2) + 0.4 * (m1 > m2 ? m1 : m2) ? "the pipes overlap" : "the streams serialise");
  }
  return 0;
}
