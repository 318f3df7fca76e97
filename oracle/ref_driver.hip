// ref_driver.hip -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).
//
// Runs the REFERENCE's own device kernels (src/cvp/cannyEdgeD.cu, compiled in place from
// /root/reference by oracle/build_ref.sh with hipcc for gfx950 -- nothing is copied into this
// repository) so that the CPU restatement in canny_oracle.c can be validated against the
// reference itself on the GPU box.  The reference's host file (src/cvp/cannyEdgeH.cu) cannot be
// built here (needs OpenCV, CUDA-GL interop, spdlog), so the launch sequence below restates it:
// grid/block shapes and the hysteresis host loop follow cannyEdgeH.cu:214-338 line by line.
// The product (libhipcanny.so) never links this file.
//
// The kernels come from the single translation unit below; the include path is given by the
// build script (-I/root/reference/src/cvp), <math_constants.h> is the genuine CUDA header that
// ships with this image's triton wheel, and hip/hip_runtime.h is force-included by the script.
#include "cannyEdgeD.cu"

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>

using namespace cvp::cuda;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "ref_driver: %s -> %s\n", #x, hipGetErrorString(e_));     \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)

namespace {
struct Bufs {
  unsigned char *rgb = nullptr, *mono = nullptr, *blurr = nullptr, *nms = nullptr, *thresh = nullptr, *hyster = nullptr,
                *hysterTemp = nullptr, *disp = nullptr;
  float *sobelX = nullptr, *sobelY = nullptr, *grad = nullptr, *slope = nullptr;
  int *flag = nullptr;
  size_t rgbP = 0, monoP = 0, blurrP = 0, nmsP = 0, threshP = 0, hysterP = 0, hysterTempP = 0, sxP = 0, syP = 0, gradP = 0,
         slopeP = 0;
};

int alloc(Bufs &b, int W, int H, int C)
{
  // cannyEdgeH.cu:346-369
  CK(hipMallocPitch((void **)&b.rgb, &b.rgbP, (size_t)W * C, H));
  CK(hipMallocPitch((void **)&b.mono, &b.monoP, W, H));
  CK(hipMallocPitch((void **)&b.blurr, &b.blurrP, W, H));
  CK(hipMallocPitch((void **)&b.sobelX, &b.sxP, W * sizeof(float), H));
  CK(hipMallocPitch((void **)&b.sobelY, &b.syP, W * sizeof(float), H));
  CK(hipMallocPitch((void **)&b.grad, &b.gradP, W * sizeof(float), H));
  CK(hipMallocPitch((void **)&b.slope, &b.slopeP, W * sizeof(float), H));
  CK(hipMallocPitch((void **)&b.nms, &b.nmsP, W, H));
  CK(hipMallocPitch((void **)&b.thresh, &b.threshP, W, H));
  CK(hipMallocPitch((void **)&b.hyster, &b.hysterP, W, H));
  CK(hipMallocPitch((void **)&b.hysterTemp, &b.hysterTempP, W, H));
  CK(hipMalloc((void **)&b.disp, (size_t)W * H));
  CK(hipMalloc((void **)&b.flag, sizeof(int)));
  // cannyEdgeH.cu:372-380
  std::array<std::array<float, 5>, 5> GK_CPU = { { { 2, 4, 5, 4, 2 }, { 4, 9, 12, 9, 4 }, { 5, 12, 15, 12, 5 }, { 4, 9, 12, 9, 4 }, { 2, 4, 5, 4, 2 } } };
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) GK_CPU[i][j] *= 1 / 159.0f;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(GK), GK_CPU.data(), 25 * sizeof(float)));
  return 0;
}

void release(Bufs &b)
{
  hipFree(b.rgb); hipFree(b.mono); hipFree(b.blurr); hipFree(b.sobelX); hipFree(b.sobelY); hipFree(b.grad);
  hipFree(b.slope); hipFree(b.nms); hipFree(b.thresh); hipFree(b.hyster); hipFree(b.hysterTemp); hipFree(b.disp);
  hipFree(b.flag);
}

// One frame through the reference launch sequence.  Returns hysteresis launch count (>=1).
int run_frame(Bufs &b, int W, int H, int C, unsigned char low, unsigned char high, bool with_host_sync)
{
  const int BS = MAX_2D_BLOCK_SIDE;
  dim3 blocks(BS, BS, 1);
  dim3 gridP((W + BS - 1) / BS, (H + BS - 1) / BS, 1);
  dim3 gridG((W + BS - 4 - 1) / (BS - 4), (H + BS - 4 - 1) / (BS - 4), 1);
  dim3 gridT((W + BS - 2 - 1) / (BS - 2), (H + BS - 2 - 1) / (BS - 2), 1);
  if (C == 3)  // cannyEdgeH.cu:220-222 (for 1-channel input the reference's call is a bug, SURVEY §3 ii: skipped)
    hipLaunchKernelGGL(rgb2mono, gridP, blocks, 0, 0, b.rgb, b.mono, W, H, (int)b.rgbP, (int)b.monoP);
  hipLaunchKernelGGL(gaussianFilter5x5, gridG, blocks, 0, 0, b.mono, b.blurr, W, H, (int)b.monoP, (int)b.blurrP);  // :235-238
  hipLaunchKernelGGL(sobelXY, gridT, blocks, 0, 0, b.blurr, b.sobelX, b.sobelY, W, H, (int)b.blurrP,
                     (int)(b.sxP / sizeof(float)), (int)(b.syP / sizeof(float)));  // :253-255
  hipLaunchKernelGGL(gradSlope, gridP, blocks, 0, 0, b.sobelX, b.sobelY, b.grad, b.slope, W, H, (int)(b.sxP / sizeof(float)),
                     (int)(b.syP / sizeof(float)), (int)(b.gradP / sizeof(float)), (int)(b.slopeP / sizeof(float)));  // :258-259
  hipLaunchKernelGGL(nonMaxSuppr, gridT, blocks, 0, 0, b.grad, b.slope, b.nms, W, H, (int)(b.gradP / sizeof(float)),
                     (int)(b.slopeP / sizeof(float)), (int)b.nmsP);  // :272-275
  hipLaunchKernelGGL(doubleThreshold, gridP, blocks, 0, 0, b.nms, b.thresh, W, H, (int)b.nmsP, (int)b.threshP, low, high);  // :288-290
  // :307-324
  int isImageModified = 0;
  hipMemcpy(b.flag, &isImageModified, sizeof(int), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(hysteresis, gridT, blocks, 0, 0, b.thresh, b.hyster, b.flag, W, H, (int)b.threshP, (int)b.hysterP);
  hipMemcpy(&isImageModified, b.flag, sizeof(int), hipMemcpyDeviceToHost);
  int launches = 1, nbIters = 0;
  const int maxNbIters = 100;
  while (nbIters < maxNbIters && isImageModified) {
    std::swap(b.hyster, b.hysterTemp);
    std::swap(b.hysterP, b.hysterTempP);
    isImageModified = 0;
    hipMemcpy(b.flag, &isImageModified, sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(hysteresis, gridT, blocks, 0, 0, b.hysterTemp, b.hyster, b.flag, W, H, (int)b.hysterTempP, (int)b.hysterP);
    hipMemcpy(&isImageModified, b.flag, sizeof(int), hipMemcpyDeviceToHost);
    nbIters++;
    launches++;
  }
  std::swap(b.hyster, b.hysterTemp);  // :328-329
  std::swap(b.hysterP, b.hysterTempP);
  hipLaunchKernelGGL(removeCandidates, gridP, blocks, 0, 0, b.hysterTemp, b.hyster, W, H, (int)b.hysterTempP, (int)b.hysterP);  // :331-333
  (void)with_host_sync;
  return launches;
}
}  // namespace

extern "C" {

// Runs one host frame (tight rows, `channels` = 1 or 3) through the reference kernels and copies
// every stage buffer back (tight W*H; sobel as float planes).  Any output pointer may be null.
int ref_run(const uint8_t *host_in, int W, int H, int C, int low, int high, uint8_t *mono, uint8_t *blurr, float *sobelX,
            float *sobelY, float *grad, float *slope, uint8_t *grad_disp, uint8_t *nms, uint8_t *thresh, uint8_t *hyster,
            int *hyst_launches)
{
  Bufs b;
  if (alloc(b, W, H, C)) return -1;
  if (C == 3) CK(hipMemcpy2D(b.rgb, b.rgbP, host_in, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyHostToDevice));  // :136
  else CK(hipMemcpy2D(b.mono, b.monoP, host_in, W, W, H, hipMemcpyHostToDevice));                               // :144
  int n = run_frame(b, W, H, C, (unsigned char)low, (unsigned char)high, true);
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  if (hyst_launches) *hyst_launches = n;
  if (mono) CK(hipMemcpy2D(mono, W, b.mono, b.monoP, W, H, hipMemcpyDeviceToHost));
  if (blurr) CK(hipMemcpy2D(blurr, W, b.blurr, b.blurrP, W, H, hipMemcpyDeviceToHost));
  if (sobelX) CK(hipMemcpy2D(sobelX, W * 4, b.sobelX, b.sxP, W * 4, H, hipMemcpyDeviceToHost));
  if (sobelY) CK(hipMemcpy2D(sobelY, W * 4, b.sobelY, b.syP, W * 4, H, hipMemcpyDeviceToHost));
  if (grad) CK(hipMemcpy2D(grad, W * 4, b.grad, b.gradP, W * 4, H, hipMemcpyDeviceToHost));
  if (slope) CK(hipMemcpy2D(slope, W * 4, b.slope, b.slopeP, W * 4, H, hipMemcpyDeviceToHost));
  if (grad_disp) {  // cannyEdgeH.cu:183-185
    dim3 blocks(MAX_2D_BLOCK_SIDE, MAX_2D_BLOCK_SIDE, 1);
    dim3 grid((W + MAX_2D_BLOCK_SIDE - 1) / MAX_2D_BLOCK_SIDE, (H + MAX_2D_BLOCK_SIDE - 1) / MAX_2D_BLOCK_SIDE, 1);
    hipLaunchKernelGGL(float2uchar, grid, blocks, 0, 0, b.grad, b.disp, W, H, (int)(b.gradP / sizeof(float)), W);
    CK(hipMemcpy(grad_disp, b.disp, (size_t)W * H, hipMemcpyDeviceToHost));
  }
  if (nms) CK(hipMemcpy2D(nms, W, b.nms, b.nmsP, W, H, hipMemcpyDeviceToHost));
  if (thresh) CK(hipMemcpy2D(thresh, W, b.thresh, b.threshP, W, H, hipMemcpyDeviceToHost));
  if (hyster) CK(hipMemcpy2D(hyster, W, b.hyster, b.hysterP, W, H, hipMemcpyDeviceToHost));
  release(b);
  return 0;
}

// Reference hysteresis alone on a host tri-state map (0/128/255).
int ref_hysteresis(const uint8_t *host_thresh, int W, int H, uint8_t *hyster, int *hyst_launches)
{
  Bufs b;
  if (alloc(b, W, H, 1)) return -1;
  CK(hipMemcpy2D(b.thresh, b.threshP, host_thresh, W, W, H, hipMemcpyHostToDevice));
  const int BS = MAX_2D_BLOCK_SIDE;
  dim3 blocks(BS, BS, 1);
  dim3 gridP((W + BS - 1) / BS, (H + BS - 1) / BS, 1);
  dim3 gridT((W + BS - 2 - 1) / (BS - 2), (H + BS - 2 - 1) / (BS - 2), 1);
  int isImageModified = 0;
  hipMemcpy(b.flag, &isImageModified, sizeof(int), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(hysteresis, gridT, blocks, 0, 0, b.thresh, b.hyster, b.flag, W, H, (int)b.threshP, (int)b.hysterP);
  hipMemcpy(&isImageModified, b.flag, sizeof(int), hipMemcpyDeviceToHost);
  int launches = 1, nbIters = 0;
  while (nbIters < 100 && isImageModified) {
    std::swap(b.hyster, b.hysterTemp);
    std::swap(b.hysterP, b.hysterTempP);
    isImageModified = 0;
    hipMemcpy(b.flag, &isImageModified, sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(hysteresis, gridT, blocks, 0, 0, b.hysterTemp, b.hyster, b.flag, W, H, (int)b.hysterTempP, (int)b.hysterP);
    hipMemcpy(&isImageModified, b.flag, sizeof(int), hipMemcpyDeviceToHost);
    nbIters++;
    launches++;
  }
  std::swap(b.hyster, b.hysterTemp);
  std::swap(b.hysterP, b.hysterTempP);
  hipLaunchKernelGGL(removeCandidates, gridP, blocks, 0, 0, b.hysterTemp, b.hyster, W, H, (int)b.hysterTempP, (int)b.hysterP);
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  if (hyst_launches) *hyst_launches = launches;
  CK(hipMemcpy2D(hyster, W, b.hyster, b.hysterP, W, H, hipMemcpyDeviceToHost));
  release(b);
  return 0;
}

// Wall time (ms per frame, hipEvent-timed, device-resident input) of the reference launch
// sequence repeated `iters` times on one frame: what CudaCam's own design costs on this GPU.
int ref_time_ms(const uint8_t *host_in, int W, int H, int C, int low, int high, int iters, float *ms_per_frame, int *hyst_launches)
{
  Bufs b;
  if (alloc(b, W, H, C)) return -1;
  if (C == 3) CK(hipMemcpy2D(b.rgb, b.rgbP, host_in, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyHostToDevice));
  else CK(hipMemcpy2D(b.mono, b.monoP, host_in, W, W, H, hipMemcpyHostToDevice));
  int n = 0;
  for (int i = 0; i < 3; ++i) n = run_frame(b, W, H, C, (unsigned char)low, (unsigned char)high, true);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) n = run_frame(b, W, H, C, (unsigned char)low, (unsigned char)high, true);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  if (ms_per_frame) *ms_per_frame = ms / iters;
  if (hyst_launches) *hyst_launches = n;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  release(b);
  return 0;
}
}
