// gtx_scanown.hip -- genomic_scans counts for reads that ARE in (class, start) order: owner-computes.
//
// The general scan (gtx_kernels.hip: scan_hist_kernel + scan_window_kernel) lets every wave add its reads to a global
// micro-window histogram with atomics and sums windows in a second kernel.  With a fine window step (-w 500 -d 25: 124 M
// micro-windows for hg38) that is a 0.5 GB memset, 0.5 GB of memory-side atomics, and 0.5 GB read back: 1.0 ms per 100 M reads
// of which the reads themselves are 0.2.  When the reads are sorted the work can be turned around: a block OWNS a range of
// `tile` consecutive windows of one class, finds the reads that fall into them with two binary searches (scan_bounds_kernel,
// one thread per block), counts them into LDS counters, prefix-sums the counters and writes its windows
//     out[k] = sum_{j < W/D} v[k + j]                (UnsortedGenomicRegionSetScanner, gtools/genomic_intervals.cpp:5058-5075;
//                                                      the same numbers SortedGenomicRegionSetScanner::Next yields, :4928-4957)
// once, with plain coalesced stores.  No micro-window array in HBM, no atomics, no memset: the traffic is the reads
// (neighbouring blocks share the reads of W/D - 1 micro-windows) and the 8-byte window sums.
//
// Exactness does not rest on the caller's word.  The blocks' own read ranges tile the whole stream, and every block checks
// that its range is in non-decreasing (class, start) order (one comparison per read it loads anyway, plus the seam to the
// read before): if every block passes, the stream is sorted, the searches were exact and so are the sums.  If any block
// fails, `flag` is set and the caller's conditional launches of the general kernels recompute everything.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "gtx_kernels.h"

namespace gtx {

typedef unsigned long long u64;
typedef long long i64;
struct __attribute__((packed, aligned(4))) Tri5 { int c, s, e; };

static constexpr int kOwnThreads = 256;
static constexpr int kOwnMaxTile = 8192;

// first read whose (class, start) is not below (kc, kp)
__device__ __forceinline__ i64 lower_bound_reads(const Tri5 *__restrict__ reads, i64 n, int kc, int kp)
{
  i64 lo = 0, hi = n;
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    const int c = reads[mid].c, s = reads[mid].s;
    if (c < kc || (c == kc && s < kp)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int class_of_block(const ScanOwn &o, int nClasses, i64 b)
{
  int lo = 0, hi = nClasses - 1;                                  // last class whose first block is <= b (classes without windows have none)
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (o.blkOff[mid] <= b) lo = mid; else hi = mid - 1; }
  return lo;
}

// bounds[b] = first read of block b's own range (the ranges of consecutive blocks tile [0, n)), bounds[total + 1 + b] = end of
// the reads block b counts (its own range + the reads of the next W/D - 1 micro-windows, inside its class)
__global__ __launch_bounds__(256) void scan_bounds_kernel(const Tri5 *__restrict__ reads, i64 n, ScanArgs a, ScanOwn o)
{
  const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > o.totalBlocks) return;
  if (b == o.totalBlocks) { o.bounds[b] = n; return; }
  const int c = class_of_block(o, a.nClasses, b);
  const i64 k = b - o.blkOff[c], nb = o.blkOff[c + 1] - o.blkOff[c];
  const i64 D = a.winStep;
  // start key of block k: the first position of its first micro-window; the first block of the first class starts at read 0
  i64 first;
  if (b == 0) first = 0;
  else first = lower_bound_reads(reads, n, c, k == 0 ? INT_MIN : (int)(k * o.tile * D + 1));
  o.bounds[b] = first;
  i64 endPos = ((k + 1) * (i64)o.tile + a.comb - 1) * D + 1;       // first position behind the micro-windows the block sums
  i64 end;
  if (k == nb - 1 || endPos > (i64)INT_MAX) end = lower_bound_reads(reads, n, c + 1, INT_MIN);
  else end = lower_bound_reads(reads, n, c, (int)endPos);
  o.bounds[o.totalBlocks + 1 + b] = end;
}

template <bool WEIGHTED>
__global__ __launch_bounds__(kOwnThreads) void scan_own_kernel(const Tri5 *__restrict__ reads, const int *__restrict__ weights, i64 n, ScanArgs a, ScanOwn o,
                                                               u64 *__restrict__ out)
{
  typedef typename std::conditional<WEIGHTED, u64, unsigned>::type ct;
  extern __shared__ unsigned char ldsRaw[];
  ct *v = (ct *)ldsRaw;                                            // micro-window counters of the block, then their exclusive prefix sums
  __shared__ ct wsum[kOwnThreads / 64];
  const i64 b = blockIdx.x;
  const int c = class_of_block(o, a.nClasses, b);
  const i64 k = b - o.blkOff[c];
  const i64 nWin = a.winOff[c + 1] - a.winOff[c], nMicro = a.nMicro[c];
  const i64 w0 = k * o.tile;                                       // first window = first micro-window of the block
  const int cntWin = (int)(nWin - w0 < o.tile ? nWin - w0 : o.tile);
  const int L = cntWin + a.comb - 1;                               // micro-windows the block needs (all inside the class: w0 + L <= nMicro)
  for (int i = threadIdx.x; i <= L; i += kOwnThreads) v[i] = 0;
  __syncthreads();
  const i64 own0 = o.bounds[b], own1 = o.bounds[b + 1], cnt1 = o.bounds[o.totalBlocks + 1 + b];
  const i64 hi = own1 > cnt1 ? own1 : cnt1;
  bool bad = own0 > own1 || own0 > cnt1;                           // searches on unsorted data need not be monotone
  // four reads in flight per thread; the read before a lane's comes from the lane below (one extra load per wave)
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  for (i64 base = own0; base < hi; base += U * kOwnThreads) {
    Tri5 t[U]; int wt[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const i64 i = base + u * kOwnThreads + threadIdx.x;
      t[u].c = INT_MAX; t[u].s = INT_MAX; t[u].e = 0; wt[u] = 1;
      if (i < hi) { t[u] = reads[i]; if (WEIGHTED) wt[u] = weights[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const i64 i = base + u * kOwnThreads + threadIdx.x;
      int pc = __shfl_up(t[u].c, 1), ps = __shfl_up(t[u].s, 1);
      if (lane == 0 && i > 0 && i < own1) { pc = reads[i - 1].c; ps = reads[i - 1].s; }
      if (i < own1 && i > 0) bad |= t[u].c < pc || (t[u].c == pc && t[u].s < ps);   // order inside the own range and across the seam before it
      if (i < cnt1 && t[u].c == c && (a.sortedRule || (t[u].s <= t[u].e && t[u].e > 0))) {
        i64 pos = t[u].s;
        if (a.sortedRule && pos < 1) pos = 1;                      // the sorted scanner takes START <= stop of the first window
        if (pos >= 1) {
          const unsigned x = (unsigned)(pos - 1), d = (unsigned)a.winStep;
          unsigned q = d == 1 ? x : __umulhi(x, a.winStepInv);
          unsigned r = x - q * d;
          if (r >= d) { q++; r -= d; }
          if (r >= d) q++;
          const i64 m = (i64)q - w0;                               // micro-window inside the block
          if ((i64)q < nMicro && m >= 0 && m < L) atomicAdd(&v[m], WEIGHTED ? (ct)(i64)wt[u] : (ct)1);
        }
      }
    }
  }
  if (__syncthreads_or(bad)) { if (threadIdx.x == 0) *o.flag = 1; return; }    // the general kernels will redo everything
  // exclusive prefix sums of v[0..L] in place (thread = a run of consecutive entries)
  const int per = (L + 1 + kOwnThreads - 1) / kOwnThreads, i0 = threadIdx.x * per;
  ct s = 0;
  for (int j = 0; j < per; j++) if (i0 + j <= L) s += v[i0 + j];
  const int wv = threadIdx.x >> 6;
  ct inc = s;
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) { const ct up = __shfl_up(inc, sh); if (lane >= sh) inc += up; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  ct run = inc - s;
  for (int j = 0; j < wv; j++) run += wsum[j];
  for (int j = 0; j < per; j++) if (i0 + j <= L) { const ct x = v[i0 + j]; v[i0 + j] = run; run += x; }
  __syncthreads();
  u64 *__restrict__ dst = out + a.outOff[c] + w0;
  for (int i = threadIdx.x; i < cntWin; i += kOwnThreads) dst[i] = (u64)(ct)(v[i + a.comb] - v[i]);
}

// Windows per block, 0 = the owner pass does not apply.  It pays when the micro-windows are many for the reads -- the general
// kernels then spend their time on the micro-window array (100 M reads, hg38: -d 25 general 0.96 ms, owner 0.49 ms; -d 1000
// general 0.22 ms, owner 0.38 ms: there the array is 12 MB and the atomics are few) -- so: at most 8 reads per micro-window.
// Tile: 2048 windows (100 M reads, -w 500 -d 25: 8192 -> 0.56 ms, 4096 -> 0.49, 2048 -> 0.49, 1024 -> 0.55; weighted 0.84 / 0.63 /
// 0.56 / 0.60), at least 2 x (W/D) so that the micro-windows shared with the next block stay a fraction.
int scan_own_tile(i64 nReads, i64 totalMicro, int comb)
{
  if (comb > kOwnMaxTile / 4 || totalMicro <= 0) return 0;
  const char *force = getenv("GTX_SCAN_OWN_ALWAYS");                 // (tests: the pass on any geometry)
  if (!(force && atoi(force)) && (double)nReads > 8.0 * (double)totalMicro) return 0;
  const i64 cap = getenv("GTX_SCAN_OWN_TILE") ? atoll(getenv("GTX_SCAN_OWN_TILE")) : 2048;
  i64 t = cap;
  if (t < 2 * comb) t = 2 * comb;
  if (t < 64) t = 64;
  if (t > kOwnMaxTile) t = kOwnMaxTile;
  int p = 64; while (p < t) p *= 2;                                 // a power of two: few distinct layouts
  return p > kOwnMaxTile ? kOwnMaxTile : p;
}

hipError_t launch_scan_own(const void *reads, const void *weights, i64 n, const ScanArgs &a, const ScanOwn &o, u64 *out, hipStream_t st)
{
  if (o.totalBlocks <= 0) return hipSuccess;
  scan_bounds_kernel<<<(unsigned)((o.totalBlocks + 1 + 255) / 256), 256, 0, st>>>((const Tri5 *)reads, n, a, o);
  const size_t lds = (size_t)(o.tile + a.comb + 1) * (weights ? 8 : 4);
  static PerDevice attr;
  {
    hipError_t e = attr.once([] {
      hipError_t e2 = hipFuncSetAttribute((const void *)scan_own_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void *)scan_own_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      return e2;
    });
    if (e != hipSuccess) return e;
  }
  if (weights) scan_own_kernel<true><<<(unsigned)o.totalBlocks, kOwnThreads, lds, st>>>((const Tri5 *)reads, (const int *)weights, n, a, o, out);
  else scan_own_kernel<false><<<(unsigned)o.totalBlocks, kOwnThreads, lds, st>>>((const Tri5 *)reads, (const int *)weights, n, a, o, out);
  return hipGetLastError();
}

}  // namespace gtx
