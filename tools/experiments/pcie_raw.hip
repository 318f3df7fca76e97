// Raw PCIe rates of the GPU box, page-locked host memory: H2D alone, D2H alone, both at once on two streams -- the
// ceiling for bench.py's host_fed leg and tools/stream_bench.  hipcc tools/experiments/pcie_raw.hip -o tools/bin/pcie_raw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
  const size_t bytes = 32u << 20;  // a batch of 16 1080p frames
  const int reps = 40;
  void *h0, *h1, *d0, *d1;
  CK(hipHostMalloc(&h0, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&h1, bytes, hipHostMallocDefault));
  CK(hipMalloc(&d0, bytes)); CK(hipMalloc(&d1, bytes));
  std::memset(h0, 1, bytes); std::memset(h1, 2, bytes);
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  auto run = [&](int mode) -> double {  // 0: H2D, 1: D2H, 2: both
    for (int w = 0; w < 2; ++w) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) {
        if (mode != 1) (void)hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0);
        if (mode != 0) (void)hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s1);
      }
      (void)hipStreamSynchronize(s0); (void)hipStreamSynchronize(s1);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (w == 1) return (double)bytes * reps / dt / 1e9;
    }
    return 0;
  };
  std::printf("H2D alone        %.1f GB/s\n", run(0));
  std::printf("D2H alone        %.1f GB/s\n", run(1));
  std::printf("H2D + D2H at once %.1f GB/s each way\n", run(2));
  return 0;
}
