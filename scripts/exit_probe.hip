// exit_probe.hip -- how long does a process that used the GPU take to go away after _exit, and what makes it longer?
// usage: exit_probe VRAM_MB PINNED_MB THREADS PAGEABLE_MB QUICK(0|1) [WHEN]   (diagnostic, not part of the product)
// WHEN: 0 = the busy threads run during the HIP calls and are joined before leaving (default), 1 = they start only after the HIP
// calls, run 50 ms and are joined, 2 = as 1 but still running at _exit, 3 = during the HIP calls, but sleeping instead of busy
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <unistd.h>
#include <vector>
__global__ void k(int *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (int)i; }
int main(int argc, char **argv)
{
  const long vram = argc > 1 ? atol(argv[1]) : 0, pinned = argc > 2 ? atol(argv[2]) : 0, threads = argc > 3 ? atol(argv[3]) : 0, pageable = argc > 4 ? atol(argv[4]) : 0;
  const int quick = argc > 5 ? atoi(argv[5]) : 1, when = argc > 6 ? atoi(argv[6]) : 0;
  std::vector<std::thread> th; volatile bool stop = false;
  auto busy = [&stop, when] { std::vector<char> buf(8 << 20); while (!stop) { if (when == 3) usleep(1000); else memset(buf.data(), 1, buf.size()); } };
  if (when == 0 || when == 3) for (long t = 0; t < threads; t++) th.emplace_back(busy);
  int *d = nullptr; void *h = nullptr;
  if (hipSetDevice(0) != hipSuccess) return 2;
  if (vram) { if (hipMalloc(&d, (size_t)vram << 20) != hipSuccess) return 3; k<<<(unsigned)(((size_t)vram << 18) / 256), 256>>>(d, (size_t)vram << 18); }
  if (pinned) { if (hipHostMalloc(&h, (size_t)pinned << 20) != hipSuccess) return 4; memset(h, 1, (size_t)pinned << 20); }
  if (pageable) { std::vector<char> src((size_t)pageable << 20, 1); int *d2; if (hipMalloc(&d2, (size_t)pageable << 20) != hipSuccess) return 5; if (hipMemcpy(d2, src.data(), src.size(), hipMemcpyHostToDevice) != hipSuccess) return 6; }
  (void)hipDeviceSynchronize();
  if (when == 1 || when == 2) { for (long t = 0; t < threads; t++) th.emplace_back(busy); usleep(50000); }
  if (when != 2) { stop = true; for (auto &t : th) t.join(); }
  fprintf(stderr, "leaving at epoch ms %lld\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count());
  if (quick) _exit(0);
  return 0;
}
