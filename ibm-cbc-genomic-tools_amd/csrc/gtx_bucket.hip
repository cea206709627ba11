// gtx_bucket.hip -- counting reads that arrive in NO particular order (keep unsorted input
// off the per-read global atomics).
//
// The order-agnostic kernel of gtx_kernels.hip does two searches and two memory-side atomics per read; at
// ~2.1e10 scattered atomics/s that is 9.4 ms for 100 M reads.  Here the reads are first PARTITIONED by where
// their start falls among the reference ends -- buckets of <= 2048 consecutive boundaries of one class,
// i.e. a 2049-slot piece of histogram A -- and then every bucket is counted by blocks that keep the
// bucket's piece of both boundary arrays and of both histograms in LDS: searches and atomics stay on the
// CU, the global histograms receive one contiguous flush per block.  A full sort is not needed (counting does
// not depend on the order inside a bucket), one partition level is enough.
//
//   bucket_hist_kernel     bucket id of every read (binary search of the start in the bucket table, LDS),
//                          ids written out (2 B/read), per-bucket totals via LDS counters; also counts the
//                          reads of unknown class / start > end for gtx_count_info
//   bucket_scan_kernel     exclusive prefix of the totals (one block)
//   bucket_scatter_kernel  blocks of 4096 reads: LDS histogram of the ids, ONE global reservation per
//                          (block, bucket), reads copied to their bucket's range
//   bucket_count_kernel    grid = buckets x splits: slices + histograms in LDS (72 KB), two LDS binary
//                          searches + two LDS atomics per read, contiguous atomic flush.  A read whose end lies
//                          beyond the bucket's slice of the starts array (longer than the bucket is wide)
//                          falls back to a global search + atomic for histogram B.
// Traffic per read: 12 B read twice, 8 B (start, end) written and read once, 2 x 2 B ids: ~4.4 GB for 100 M reads.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <type_traits>
#include "gtx_kernels.h"

namespace gtx {

typedef unsigned long long u64;
typedef long long i64;
struct __attribute__((packed, aligned(4))) Tri3 { int c, s, e; };

static constexpr unsigned short kNoBucket = 0xFFFF;

template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_hist_kernel(const Tri3 *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a, BucketTable t, BucketWork w)
{
  extern __shared__ int lds[];
  int *posHi = lds; unsigned *cnt = (unsigned *)(lds + t.nB);
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) { posHi[i] = t.posHi[i]; cnt[i] = 0; }
  __syncthreads();
  i64 nNoClass = 0, nDegen = 0, firstDegen = INT64_MAX;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const Tri3 r = reads[i];
    unsigned short id = kNoBucket;
    if ((unsigned)r.c >= (unsigned)a.nClasses) nNoClass++;
    else if (r.s > r.e + a.zeroLenOk) {
      nDegen++; if (i < firstDegen) firstDegen = i;
      if (a.side) { const unsigned k = atomicAdd(a.sideCount, 1u); if (k < (unsigned)a.sideCap) a.side[k] = make_int4(r.c, r.s, r.e, WEIGHTED ? weights[i] : 1); }   // see CountArgs::side
    }
    else {
      int lo = t.clsStart[r.c], hi = t.clsStart[r.c + 1];
      if (lo < hi) {
        hi--;                                                     // the class's last bucket takes everything above
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (r.s <= posHi[mid]) hi = mid; else lo = mid + 1; }
        id = (unsigned short)lo;
        atomicAdd(&cnt[lo], 1u);
      }
    }
    w.ids[i] = id;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) if (cnt[i]) atomicAdd(&w.count[i], cnt[i]);
  if (nNoClass) atomicAdd((u64 *)&a.info->n_no_class, (u64)nNoClass);
  if (nDegen) { atomicAdd((u64 *)&a.info->n_degenerate, (u64)nDegen); atomicMin((i64 *)&a.info->first_degenerate, firstDegen + a.indexBase); }
}

// offset[b] = reads in buckets < b; cursor = copy for the scatter's reservations; the totals are zeroed for the next call
__global__ __launch_bounds__(1024) void bucket_scan_kernel(BucketTable t, BucketWork w)
{
  __shared__ unsigned part[1024];
  const int per = (t.nB + 1023) / 1024, b0 = threadIdx.x * per;
  unsigned s = 0;
  for (int k = 0; k < per; k++) if (b0 + k < t.nB) s += w.count[b0 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned run = 0; for (int i = 0; i < 1024; i++) { unsigned v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  unsigned run = part[threadIdx.x];
  for (int k = 0; k < per; k++) if (b0 + k < t.nB) {
    const unsigned v = w.count[b0 + k];
    w.offset[b0 + k] = run; w.cursor[b0 + k] = run; w.count[b0 + k] = 0;
    run += v;
  }
  if (threadIdx.x == 1023) w.offset[t.nB] = run;
}

template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_scatter_kernel(const Tri3 *__restrict__ reads, const int *__restrict__ weights, i64 n, BucketTable t, BucketWork w)
{
  extern __shared__ int lds[];
  unsigned *cnt = (unsigned *)lds, *base = cnt + t.nB;
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  const i64 first = (i64)blockIdx.x * 4096;
  unsigned short id[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const i64 i = first + k * 1024 + threadIdx.x;
    id[k] = i < n ? w.ids[i] : kNoBucket;
    if (id[k] != kNoBucket) atomicAdd(&cnt[id[k]], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) {
    const unsigned c = cnt[i];
    if (c) base[i] = atomicAdd(&w.cursor[i], c);               // one reservation per (block, bucket)
    cnt[i] = 0;
  }
  __syncthreads();
  int2 *__restrict__ out = (int2 *)w.tmpReads;                  // (start, end): the class is the bucket's
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (id[k] == kNoBucket) continue;
    const i64 i = first + k * 1024 + threadIdx.x;
    const unsigned pos = base[id[k]] + atomicAdd(&cnt[id[k]], 1u);
    const Tri3 r = reads[i];
    out[pos] = make_int2(r.s, r.e);
    if (WEIGHTED) w.tmpWeights[pos] = weights[i];
  }
}

// all lanes of the wave call this; slot < 0 = nothing to add.  One LDS atomic per run of equal slots.
template <class H>
__device__ __forceinline__ void lds_run_add(H *hist, int slot, int lane)
{
  const int prev = __builtin_amdgcn_update_dpp(slot, slot, 0x138, 0xf, 0xf, false);     // lane below (wave_shr:1)
  const bool head = lane == 0 || slot != prev;
  const u64 heads = __ballot(head);
  if (head && slot >= 0) {
    const u64 rest = lane == 63 ? 0 : heads >> (lane + 1);
    atomicAdd(&hist[slot], (H)(rest ? __builtin_ctzll(rest) + 1 : 64 - lane));
  }
}

static constexpr int kBktE = 2048, kBktS = 4096;                 // boundaries per bucket (ends array) / starts-array slice in LDS

template <bool WEIGHTED>
__global__ __launch_bounds__(512) void bucket_count_kernel(CountArgs a, BucketTable t, BucketWork w, int splits)
{
  __shared__ int sE[kBktE], sS[kBktS];
  typedef typename std::conditional<WEIGHTED, u64, unsigned>::type hist_t;   // a block sees < 2^32 reads
  __shared__ hist_t hA[kBktE + 1], hB[kBktS + 1];
  const int b = blockIdx.x / splits, k = blockIdx.x % splits;
  const unsigned off = w.offset[b], cntB = w.offset[b + 1] - off;
  const unsigned r0 = off + (unsigned)((u64)cntB * k / splits), r1 = off + (unsigned)((u64)cntB * (k + 1) / splits);
  if (r0 == r1) return;
  const int eLo = t.eLo[b], nE = t.eHi[b] - eLo, sLo = t.sLo[b], sHi = t.sHi[b], nS = sHi - sLo, cls = t.cls[b];
  const int segEnd = a.segStart[cls + 1];
  // slices padded with +inf to their full power-of-two size: the searches below are branch-free with a fixed trip count
  for (int i = threadIdx.x; i < kBktE; i += blockDim.x) sE[i] = i < nE ? a.sortedE[eLo + i] : INT_MAX;
  for (int i = threadIdx.x; i < kBktS; i += blockDim.x) sS[i] = i < nS ? a.sortedS[sLo + i] : INT_MAX;
  for (int i = threadIdx.x; i <= nE; i += blockDim.x) hA[i] = 0;
  for (int i = threadIdx.x; i <= nS; i += blockDim.x) hB[i] = 0;
  __syncthreads();
  const int2 *__restrict__ reads = (const int2 *)w.tmpReads;
  const int lane = threadIdx.x & 63;
  const unsigned cnt = r1 - r0;
  for (unsigned at = 0; at < cnt; at += blockDim.x) {            // wave-uniform trip count: the run compression below needs all lanes
    const unsigned i = r0 + at + threadIdx.x;
    const bool live = at + threadIdx.x < cnt;
    int slotA = -1, slotB = -1;
    u64 wt = 1;
    if (live) {
      const int2 se = reads[i];
      struct { int s, e; } r = {se.x, se.y};
      if (WEIGHTED) wt = (u64)(i64)w.tmpWeights[i];
      int lo = 0;                                                 // #{E < s} inside the slice
#pragma unroll
      for (int half = kBktE / 2; half >= 1; half >>= 1) lo += sE[lo + half - 1] < r.s ? half : 0;
      lo += sE[lo] < r.s ? 1 : 0;
      slotA = lo;
      lo = 0;                                                     // #{S <= e} inside the slice (coordinates are < INT_MAX)
#pragma unroll
      for (int half = kBktS / 2; half >= 1; half >>= 1) lo += sS[lo + half - 1] <= r.e ? half : 0;
      lo += sS[lo] <= r.e ? 1 : 0;
      if (lo < nS || sHi == segEnd) slotB = lo;
      else {                                                      // the read ends beyond the slice: global search above it
        int glo = sHi, ghi = segEnd;
        while (glo < ghi) { const int mid = (int)(((i64)glo + ghi) >> 1); if (a.sortedS[mid] <= r.e) glo = mid + 1; else ghi = mid; }
        atomicAdd(&a.histB[(i64)glo + cls], wt);
      }
    }
    if (WEIGHTED) {
      if (slotA >= 0) atomicAdd(&hA[slotA], (hist_t)wt);
      if (slotB >= 0) atomicAdd(&hB[slotB], (hist_t)wt);
    } else {
      // neighbouring lanes in the same slot (input that was in order before the partition) share one LDS atomic
      lds_run_add(hA, slotA, lane);
      lds_run_add(hB, slotB, lane);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= nE; i += blockDim.x) { const u64 v = hA[i]; if (v) atomicAdd(&a.histA[(i64)eLo + i + cls], v); }
  for (int i = threadIdx.x; i <= nS; i += blockDim.x) { const u64 v = hB[i]; if (v) atomicAdd(&a.histB[(i64)sLo + i + cls], v); }
}

int bucket_e_size() { return kBktE; }
int bucket_s_size() { return kBktS; }

hipError_t launch_count_bucketed(const void *reads, const void *weights, i64 n, const CountArgs &a, const BucketTable &t, const BucketWork &w,
                                 hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const size_t ldsHist = sizeof(int) * 2 * (size_t)t.nB;
  i64 blocks = (n + 1023) / 1024; if (blocks > 2048) blocks = 2048;
  if (weights) bucket_hist_kernel<true><<<(unsigned)blocks, 1024, ldsHist, st>>>((const Tri3 *)reads, (const int *)weights, n, a, t, w);
  else bucket_hist_kernel<false><<<(unsigned)blocks, 1024, ldsHist, st>>>((const Tri3 *)reads, (const int *)weights, n, a, t, w);
  bucket_scan_kernel<<<1, 1024, 0, st>>>(t, w);
  const unsigned sblocks = (unsigned)((n + 4095) / 4096);
  if (weights) bucket_scatter_kernel<true><<<sblocks, 1024, ldsHist, st>>>((const Tri3 *)reads, (const int *)weights, n, t, w);
  else bucket_scatter_kernel<false><<<sblocks, 1024, ldsHist, st>>>((const Tri3 *)reads, (const int *)weights, n, t, w);
  // blocks of ~32k reads on average, at least one per bucket
  i64 splits = (n + (i64)t.nB * 32768 - 1) / ((i64)t.nB * 32768);
  if (splits < 1) splits = 1;
  if (splits > 512) splits = 512;
  if (weights) bucket_count_kernel<true><<<(unsigned)(t.nB * splits), 512, 0, st>>>(a, t, w, (int)splits);
  else bucket_count_kernel<false><<<(unsigned)(t.nB * splits), 512, 0, st>>>(a, t, w, (int)splits);
  return hipGetLastError();
}

}  // namespace gtx
