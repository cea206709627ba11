// membench.hip -- what HBM read rate does the triple-stream access pattern allow on this chip?
// Calibration only (not part of the product): prints GB/s for a few load shapes over a 1.2 GB buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct __attribute__((packed, aligned(4))) Tri { int c, s, e; };

// contiguous span per wave, one dwordx3 per lane per chunk, DEPTH chunks in flight
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void k_x3(const Tri *p, long n, int chunksPerWave, int *out)
{
  const int lane = threadIdx.x & 63;
  long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  long nChunks = n >> 6;
  long c0 = wave * chunksPerWave, c1 = c0 + chunksPerWave; if (c1 > nChunks) c1 = nChunks;
  int acc = 0;
  for (long c = c0; c < c1; c += DEPTH) {
    Tri t[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      long i = (c + d) * 64 + lane;
      if (c + d < c1) { if (NT) { t[d].c = __builtin_nontemporal_load(&p[i].c); t[d].s = __builtin_nontemporal_load(&p[i].s); t[d].e = __builtin_nontemporal_load(&p[i].e); } else t[d] = p[i]; }
      else { t[d].c = 0; t[d].s = 0; t[d].e = 0; }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc += t[d].c ^ t[d].s ^ t[d].e;
  }
  if (acc == 0x12345678) out[0] = acc;
}

// flat dwordx4 per lane, contiguous span per wave
template <int DEPTH>
__global__ __launch_bounds__(256) void k_x4(const int4 *p, long n16, int vecsPerWave, int *out)
{
  const int lane = threadIdx.x & 63;
  long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  long v0 = wave * vecsPerWave, v1 = v0 + vecsPerWave; if (v1 > n16) v1 = n16;
  int acc = 0;
  for (long v = v0; v < v1; v += 64 * DEPTH) {
    int4 t[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { long i = v + d * 64 + lane; t[d] = i < v1 ? p[i] : make_int4(0, 0, 0, 0); }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc += t[d].x ^ t[d].y ^ t[d].z ^ t[d].w;
  }
  if (acc == 0x12345678) out[0] = acc;
}

// persistent waves claiming spans of chunksPerSpan chunks from an atomic ticket counter (4 x 64 reads in flight, non-temporal);
// taper > 0: the last `taper` spans' worth of chunks is dealt out in quarter spans (a short tail)
__global__ __launch_bounds__(256) void k_queue(const Tri *p, long n, int chunksPerSpan, long taperFrom, unsigned long long *ticket, int *out)
{
  const int lane = threadIdx.x & 63;
  const long nChunks = n >> 6;
  int acc = 0;
  for (;;) {
    unsigned long long tk = 0;
    if (lane == 0) tk = atomicAdd(ticket, 1ull);
    tk = __shfl(tk, 0);
    long c0 = (long)tk * chunksPerSpan, c1 = c0 + chunksPerSpan;
    if (taperFrom >= 0 && c0 >= taperFrom) { const int q = chunksPerSpan / 4; c0 = taperFrom + ((long)tk - taperFrom / chunksPerSpan) * q; c1 = c0 + q; }
    if (c0 >= nChunks) break;
    if (c1 > nChunks) c1 = nChunks;
    for (long c = c0; c < c1; c += 4) {
      Tri t[4];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        long i = (c + d) * 64 + lane;
        if (c + d < c1) { t[d].c = __builtin_nontemporal_load(&p[i].c); t[d].s = __builtin_nontemporal_load(&p[i].s); t[d].e = __builtin_nontemporal_load(&p[i].e); }
        else { t[d].c = 0; t[d].s = 0; t[d].e = 0; }
      }
#pragma unroll
      for (int d = 0; d < 4; d++) acc += t[d].c ^ t[d].s ^ t[d].e;
    }
  }
  if (acc == 0x12345678) out[0] = acc;
}

template <class F> static float timeit(F f, int reps = 10)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 100000000; const size_t bytes = (size_t)n * 12;
  void *buf; int *out; unsigned long long *ticket; CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&ticket, 8 * 4096)); CK(hipMemset(ticket, 0, 8 * 4096)); int tk = 0;   // a fresh ticket counter per launch
  CK(hipMemset(buf, 1, bytes));
  printf("n = %ld reads (%.2f GB)\n", n, bytes / 1e9);
  // the streaming kernel's own pattern (4 x dwordx3 per lane in flight, non-temporal) against the span per wave, in us
  for (int cpw : {16, 24, 32, 48, 56, 64, 96, 128, 192, 256}) {
    long waves = ((n >> 6) + cpw - 1) / cpw; unsigned grid = (unsigned)((waves + 3) / 4);
    float c = timeit([&] { k_x3<4, true><<<grid, 256>>>((const Tri *)buf, n, cpw, out); });
    printf("x3 depth4-nt cpw=%3d  waves %7ld  %.1f us  %.0f GB/s\n", cpw, waves, c * 1e3, bytes / c / 1e6);
  }
  // persistent waves + ticket queue: spans in stream order, dynamic balance, no partial last round
  for (int wavesPerSimd : {8, 6, 4})
    for (int cps : {8, 16, 32, 56}) {
      const unsigned grid = 256u * wavesPerSimd;
      float c = timeit([&] { k_queue<<<grid, 256>>>((const Tri *)buf, n, cps, -1, ticket + tk++, out); });
      const long nCh = n >> 6, taperFrom = (nCh - (long)grid * 4 * cps / 2) / cps * cps;     // the last half round in quarter spans
      float d = timeit([&] { k_queue<<<grid, 256>>>((const Tri *)buf, n, cps, taperFrom > 0 ? taperFrom : -1, ticket + tk++, out); });
      printf("queue waves/simd=%d span=%2d chunks  %.1f us  %.0f GB/s   tapered tail %.1f us  %.0f GB/s\n", wavesPerSimd, cps, c * 1e3, bytes / c / 1e6, d * 1e3, bytes / d / 1e6);
    }
  if (n > 200000000) return 0;
  for (int cpw : {8, 16, 32, 64, 128}) {
    long waves = ((n >> 6) + cpw - 1) / cpw; unsigned grid = (unsigned)((waves + 3) / 4);
    float a = timeit([&] { k_x3<1, false><<<grid, 256>>>((const Tri *)buf, n, cpw, out); });
    float b = timeit([&] { k_x3<2, false><<<grid, 256>>>((const Tri *)buf, n, cpw, out); });
    float c = timeit([&] { k_x3<4, false><<<grid, 256>>>((const Tri *)buf, n, cpw, out); });
    float d = timeit([&] { k_x3<2, true><<<grid, 256>>>((const Tri *)buf, n, cpw, out); });
    printf("x3 cpw=%3d  depth1 %.0f  depth2 %.0f  depth4 %.0f  depth2-nt %.0f GB/s\n", cpw, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6, bytes / d / 1e6);
  }
  long n16 = bytes / 16;
  for (int vpw : {1024, 4096, 16384}) {
    long waves = (n16 + vpw - 1) / vpw; unsigned grid = (unsigned)((waves + 3) / 4);
    float a = timeit([&] { k_x4<1><<<grid, 256>>>((const int4 *)buf, n16, vpw, out); });
    float b = timeit([&] { k_x4<2><<<grid, 256>>>((const int4 *)buf, n16, vpw, out); });
    float c = timeit([&] { k_x4<4><<<grid, 256>>>((const int4 *)buf, n16, vpw, out); });
    printf("x4 vecs/wave=%5d  depth1 %.0f  depth2 %.0f  depth4 %.0f GB/s\n", vpw, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6);
  }
  return 0;
}
