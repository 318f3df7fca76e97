// How ROCm moves page-locked batches when copies share a stream with kernels (the FrameStreamer pattern) and when they
// have streams of their own.  A: three streams, each [H2D, kernel, D2H] per batch, host waits for the oldest batch.
// B: one H2D stream, one D2H stream, three compute streams, events between them.  32 MiB batches, GB/s each way.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void touch(unsigned *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] ^= 1u; }
int run(size_t bytes, int D_, int reps);
static int g_extra = 0;  // 1: a memset before and a 1 KiB D2H copy after the kernel on the compute stream (the detector's flag words)
static void *g_hflag = nullptr, *g_dflag = nullptr;
int main(int argc, char **argv)
{
  (void)hipHostMalloc(&g_hflag, 4096, hipHostMallocDefault); (void)hipMalloc(&g_dflag, 4096);
  for (g_extra = 0; g_extra < 2; ++g_extra) { std::printf("-- %s\n", g_extra ? "with a memset and a 1 KiB D2H copy on the compute stream" : "plain"); for (size_t mb : { 32, 64 }) for (int d : { 3 }) run(mb << 20, d, (int)(4096 / mb)); }
  return 0;
}
int run(size_t bytes, int D_, int reps)
{
  constexpr int D = 8; const int DD = D_;
  void *hin[D], *hout[D], *din[D], *dout[D];
  hipStream_t st[D], sin_, sout_;
  hipEvent_t up[D], done[D], down[D];
  for (int k = 0; k < DD; ++k) {
    (void)hipHostMalloc(&hin[k], bytes, hipHostMallocDefault); (void)hipHostMalloc(&hout[k], bytes, hipHostMallocDefault);
    (void)hipMalloc(&din[k], bytes); (void)hipMalloc(&dout[k], bytes);
    std::memset(hin[k], k + 1, bytes);
    (void)hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&up[k], hipEventDisableTiming); (void)hipEventCreateWithFlags(&done[k], hipEventDisableTiming); (void)hipEventCreateWithFlags(&down[k], hipEventDisableTiming);
  }
  (void)hipStreamCreateWithFlags(&sin_, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sout_, hipStreamNonBlocking);
  for (int mode = 0; mode < 2; ++mode) {
    double best = 0;
    for (int w = 0; w < 2; ++w) {
      bool busy[D] = { false };
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) {
        const int k = r % DD;
        if (busy[k]) { if (mode == 0) (void)hipStreamSynchronize(st[k]); else (void)hipEventSynchronize(down[k]); }
        if (mode == 0) {
          (void)hipMemcpyAsync(din[k], hin[k], bytes, hipMemcpyHostToDevice, st[k]);
          hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st[k], (unsigned *)din[k], bytes / 4);
          (void)hipMemcpyAsync(hout[k], dout[k], bytes, hipMemcpyDeviceToHost, st[k]);
        } else {
          (void)hipMemcpyAsync(din[k], hin[k], bytes, hipMemcpyHostToDevice, sin_);
          (void)hipEventRecord(up[k], sin_);
          (void)hipStreamWaitEvent(st[k], up[k], 0);
          if (g_extra) (void)hipMemsetAsync(g_dflag, 0, 4096, st[k]);
          hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st[k], (unsigned *)din[k], bytes / 4);
          if (g_extra) (void)hipMemcpyAsync(g_hflag, g_dflag, 1024, hipMemcpyDeviceToHost, st[k]);
          (void)hipEventRecord(done[k], st[k]);
          (void)hipStreamWaitEvent(sout_, done[k], 0);
          (void)hipMemcpyAsync(hout[k], dout[k], bytes, hipMemcpyDeviceToHost, sout_);
          (void)hipEventRecord(down[k], sout_);
        }
        busy[k] = true;
      }
      (void)hipDeviceSynchronize();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      best = (double)bytes * reps / dt / 1e9;
    }
    std::printf("%3zu MiB batches, %d slots  %s  %.1f GB/s each way\n", bytes >> 20, DD, mode == 0 ? "A: copies and kernels share a stream per batch slot " : "B: dedicated H2D and D2H streams, events between  ", best);
  }
  for (int k = 0; k < DD; ++k) { (void)hipHostFree(hin[k]); (void)hipHostFree(hout[k]); (void)hipFree(din[k]); (void)hipFree(dout[k]); (void)hipStreamDestroy(st[k]); }
  (void)hipStreamDestroy(sin_); (void)hipStreamDestroy(sout_);
  return 0;
}
