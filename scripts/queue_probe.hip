// which of a process's streams share a hardware queue: two spinning kernels on streams i and j overlap (different queues) or not
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  std::vector<hipStream_t> s(n);
  hipStream_t pre; hipStreamCreateWithFlags(&pre, hipStreamNonBlocking);           // (like a context's copy stream)
  hipStream_t hp; hipStreamCreateWithPriority(&hp, hipStreamNonBlocking, hi);       // (like a member's exchange stream)
  for (int i = 0; i < n; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  const long long ticks = 20000;    // 100 MHz wall clock: 200 us
  spin<<<1, 64, 0, s[0]>>>(100); hipDeviceSynchronize();
  auto pair = [&](hipStream_t a, hipStream_t b) { hipDeviceSynchronize(); double t = now(); spin<<<1, 64, 0, a>>>(ticks); spin<<<1, 64, 0, b>>>(ticks); hipDeviceSynchronize(); return now() - t; };
  printf("     "); for (int j = 0; j < n; j++) printf(" s%-3d", j); printf("  null  pre   hp\n");
  for (int i = 0; i < n; i++) {
    printf("s%-3d ", i);
    for (int j = 0; j < n; j++) printf(" %c   ", i == j ? '.' : pair(s[i], s[j]) > 330 ? 'X' : '-');
    printf("  %c     %c    %c\n", pair(s[i], 0) > 330 ? 'X' : '-', pair(s[i], pre) > 330 ? 'X' : '-', pair(s[i], hp) > 330 ? 'X' : '-');
  }
  printf("X = the two kernels ran one after the other (same hardware queue), - = side by side\n");
  return 0;
}
