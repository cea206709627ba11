// host-side cost of the runtime calls a group member's device call is made of (us per call, enqueue only)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[200]; };
__global__ void k_small(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b, int *p) { if (p && threadIdx.x == 9999) *p = b.v[3]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  hipStream_t s0, s1, s2; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
  hipEvent_t e0, e1; hipEventCreateWithFlags(&e0, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  Big b{}; const int N = 2000;
  k_small<<<1, 64, 0, s0>>>(nullptr); k_big<<<1, 64, 0, s1>>>(b, nullptr); hipEventRecord(e1, s0); hipDeviceSynchronize();
  auto run = [&](const char *name, auto fn) { hipDeviceSynchronize(); double t = now(); for (int i = 0; i < N; i++) fn(i); double dt = now() - t; hipDeviceSynchronize(); double tot = now() - t; printf("%-46s %6.2f us enqueue   %6.2f us with drain\n", name, dt / N, tot / N); };
  run("launch small kernel, one stream", [&](int) { k_small<<<1, 64, 0, s0>>>(nullptr); });
  run("launch 800-byte-arg kernel, one stream", [&](int) { k_big<<<1, 64, 0, s0>>>(b, nullptr); });
  run("launch small kernel, two streams in turn", [&](int i) { k_small<<<1, 64, 0, (i & 1) ? s1 : s0>>>(nullptr); });
  run("hipEventRecord", [&](int) { hipEventRecord(e0, s0); });
  run("hipEventRecord + hipStreamWaitEvent(other)", [&](int) { hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0); });
  run("launch; record; wait(other); launch(other)", [&](int) { k_small<<<1, 64, 0, s0>>>(nullptr); hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0); k_small<<<1, 64, 0, s1>>>(nullptr); });
  run("same with the other stream high priority", [&](int) { k_small<<<1, 64, 0, s0>>>(nullptr); hipEventRecord(e0, s0); hipStreamWaitEvent(s2, e0, 0); k_small<<<1, 64, 0, s2>>>(nullptr); });
  run("hipStreamQuery (idle stream)", [&](int) { (void)hipStreamQuery(s1); });
  run("hipEventQuery (complete)", [&](int) { (void)hipEventQuery(e0); });
  run("hipSetDevice", [&](int) { hipSetDevice(0); });
  run("hipGetLastError", [&](int) { (void)hipGetLastError(); });
  run("hipStreamWaitEvent on a completed event", [&](int) { hipStreamWaitEvent(s1, e1, 0); });
  { hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    run("record + wait(other), events with timing", [&](int) { hipEventRecord(t0, s0); hipStreamWaitEvent(s1, t0, 0); }); }
  { unsigned *flag = nullptr; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
    run("hipStreamWriteValue32", [&](int i) { hipStreamWriteValue32(s0, flag, (unsigned)i + 1, 0); });
    hipMemset(flag, 0, 64);
    run("WriteValue32 + WaitValue32(other, GEQ)", [&](int i) { hipStreamWriteValue32(s0, flag + 1, (unsigned)i + 1, 0); hipStreamWaitValue32(s1, flag + 1, (unsigned)i + 1, hipStreamWaitValueGte, 0xffffffffu); });
    hipMemset(flag, 0, 64);
    run("launch; WriteValue; WaitValue(other); launch(other)", [&](int i) { k_small<<<1, 64, 0, s0>>>(nullptr); hipStreamWriteValue32(s0, flag + 2, (unsigned)i + 1, 0); hipStreamWaitValue32(s1, flag + 2, (unsigned)i + 1, hipStreamWaitValueGte, 0xffffffffu); k_small<<<1, 64, 0, s1>>>(nullptr); }); }
  // a 3-kernel chain as a graph
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal); k_big<<<1, 64, 0, s0>>>(b, nullptr); k_small<<<1, 64, 0, s0>>>(nullptr); k_small<<<1, 64, 0, s0>>>(nullptr); hipStreamEndCapture(s0, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  run("graph of 3 kernels, one stream", [&](int) { hipGraphLaunch(ge, s0); });
  run("3 kernels launched one by one", [&](int) { k_big<<<1, 64, 0, s0>>>(b, nullptr); k_small<<<1, 64, 0, s0>>>(nullptr); k_small<<<1, 64, 0, s0>>>(nullptr); });
  return 0;
}
