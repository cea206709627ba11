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
//   bucket_hist_kernel     one streaming read of the triples: bucket of every read (binary search of the start in an LDS
//                          copy of the bucket table), per-bucket totals through LDS counters; nothing per read is
//                          written.  Also counts the reads of unknown class / start > end for gtx_count_info.
//   bucket_scan_kernel     exclusive prefix of the totals (one block)
//   bucket_split_kernel    second read of the triples, a tile of 4096 reads per block: bucket again (cheaper than carrying
//                          2 bytes per read through HBM), rank inside (block, bucket) from the LDS counter, ONE global
//                          reservation per (block, bucket), the tile regrouped by bucket in LDS and copied out so that
//                          consecutive lanes write consecutive (start, end) pairs: bursts of tile/buckets pairs
//                          instead of 8-byte scattered stores (round 1: 1.2-1.35 ms of its 3.0 ms went there)
//   bucket_count_kernel    grid = buckets x splits: the bucket's slices of both boundary arrays in LDS with a
//                          direct-address table over their value range (cell -> first boundary in the cell), so a rank is two
//                          table reads + a search among the few boundaries of one cell instead of a 12-probe binary
//                          search whose probes of different lanes fall on one LDS bank; two LDS atomics per read,
//                          contiguous atomic flush.  A read whose end lies beyond the bucket's slice of the starts array
//                          (longer than the bucket is wide) falls back to a global search + atomic for histogram B.
// Traffic per read: 12 B read twice, 8 B (start, end) written and read once: ~3.2 GB for 100 M reads.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "gtx_kernels.h"

namespace gtx {

typedef unsigned long long u64;
typedef long long i64;
struct __attribute__((packed, aligned(4))) Tri3 { int c, s, e; };

static constexpr int kNoBucket = -1;

__device__ __forceinline__ Tri3 load_tri3(const Tri3 *p)
{
  const int *q = (const int *)p;
  Tri3 t;
  t.c = __builtin_nontemporal_load(q); t.s = __builtin_nontemporal_load(q + 1); t.e = __builtin_nontemporal_load(q + 2);
  return t;
}

// bucket of a countable read (class known, not degenerate): the class's buckets are posHi-sorted, the last takes everything above
__device__ __forceinline__ int bucket_of(const int *__restrict__ posHi, const int *__restrict__ clsStart, int c, int s)
{
  int lo = clsStart[c], hi = clsStart[c + 1];
  if (lo >= hi) return kNoBucket;                                  // a class without reference regions
  hi--;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (s <= posHi[mid]) hi = mid; else lo = mid + 1; }
  return lo;
}

template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_hist_kernel(const Tri3 *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a, BucketTable t, BucketWork w)
{
  extern __shared__ int lds[];
  int *posHi = lds; unsigned *cnt = (unsigned *)(lds + t.nB); int *clsStart = lds + 2 * t.nB;
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) { posHi[i] = t.posHi[i]; cnt[i] = 0; }
  for (int i = threadIdx.x; i <= a.nClasses; i += blockDim.x) clsStart[i] = t.clsStart[i];
  __syncthreads();
  i64 nNoClass = 0, nDegen = 0, firstDegen = INT64_MAX;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const Tri3 r = load_tri3(reads + i);
    if ((unsigned)r.c >= (unsigned)a.nClasses) nNoClass++;
    else if (r.s > r.e + a.zeroLenOk) {
      nDegen++; if (i < firstDegen) firstDegen = i;
      if (a.side) { const unsigned k = atomicAdd(a.sideCount, 1u); if (k < (unsigned)a.sideCap) a.side[k] = make_int4(r.c, r.s, r.e, WEIGHTED ? weights[i] : 1); }   // see CountArgs::side
    } else {
      const int b = bucket_of(posHi, clsStart, r.c, r.s);
      if (b >= 0) atomicAdd(&cnt[b], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) if (cnt[i]) atomicAdd(&w.count[i], cnt[i]);
  if (nNoClass) atomicAdd((u64 *)&a.info->n_no_class, (u64)nNoClass);
  if (nDegen) { atomicAdd((u64 *)&a.info->n_degenerate, (u64)nDegen); atomicMin((i64 *)&a.info->first_degenerate, firstDegen + a.indexBase); }
}

// offset[b] = reads in buckets < b; cursor = copy for the split's reservations; the totals are zeroed for the next call
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

// One tile of TB reads per block.  LDS: the bucket table (posHi, clsStart), per bucket {count, first place in the tile,
// reserved place in the output}, and the tile regrouped by bucket: (start, end), bucket [, weight] per read.
template <bool WEIGHTED, int TB>
__global__ __launch_bounds__(1024) void bucket_split_kernel(const Tri3 *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a, BucketTable t, BucketWork w)
{
  constexpr int PER = TB / 1024;
  extern __shared__ int lds[];
  const int nB = t.nB;
  int *posHi = lds; unsigned *cnt = (unsigned *)(lds + nB), *lstart = cnt + nB, *gbase = lstart + nB; int *clsStart = (int *)(gbase + nB);
  int2 *stage = (int2 *)(clsStart + ((a.nClasses + 2) & ~1));
  int *sw = (int *)(stage + TB);                                   // weights of the regrouped tile (WEIGHTED)
  unsigned short *sid = (unsigned short *)(sw + (WEIGHTED ? TB : 0));
  __shared__ unsigned wsum[16];
  for (int i = threadIdx.x; i < nB; i += 1024) { posHi[i] = t.posHi[i]; cnt[i] = 0; }
  for (int i = threadIdx.x; i <= a.nClasses; i += 1024) clsStart[i] = t.clsStart[i];
  __syncthreads();
  const i64 first = (i64)blockIdx.x * TB;
  int rs[PER], re[PER], rw[PER], id[PER]; unsigned rank[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const i64 i = first + k * 1024 + threadIdx.x;
    id[k] = kNoBucket; rs[k] = re[k] = 0; rw[k] = 1; rank[k] = 0;
    if (i < n) {
      const Tri3 r = load_tri3(reads + i);
      rs[k] = r.s; re[k] = r.e;
      if (WEIGHTED) rw[k] = weights[i];
      if ((unsigned)r.c < (unsigned)a.nClasses && !(r.s > r.e + a.zeroLenOk)) id[k] = bucket_of(posHi, clsStart, r.c, r.s);
      if (id[k] >= 0) rank[k] = atomicAdd(&cnt[id[k]], 1u);       // its rank among the tile's reads of that bucket
    }
  }
  __syncthreads();
  // exclusive scan of the counts over the buckets (thread = a run of buckets) + one reservation per bucket present
  {
    const int per = (nB + 1023) / 1024, b0 = threadIdx.x * per;
    unsigned s = 0;
    for (int k = 0; k < per; k++) if (b0 + k < nB) s += cnt[b0 + k];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned run = inc - s;
    for (int k = 0; k < wv; k++) run += wsum[k];
    for (int k = 0; k < per; k++) if (b0 + k < nB) {
      const unsigned c = cnt[b0 + k];
      lstart[b0 + k] = run;
      if (c) gbase[b0 + k] = atomicAdd(&w.cursor[b0 + k], c);
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) if (id[k] >= 0) {
    const unsigned p = lstart[id[k]] + rank[k];
    stage[p] = make_int2(rs[k], re[k]); sid[p] = (unsigned short)id[k];
    if (WEIGHTED) sw[p] = rw[k];
  }
  __syncthreads();
  const unsigned total = lstart[nB - 1] + cnt[nB - 1];
  int2 *__restrict__ out = (int2 *)w.tmpReads;                    // (start, end): the class is the bucket's
  for (unsigned j = threadIdx.x; j < total; j += 1024) {
    const unsigned b = sid[j], dst = gbase[b] + (j - lstart[b]);  // neighbours of one bucket are neighbours in the output
    out[dst] = stage[j];
    if (WEIGHTED) w.tmpWeights[dst] = sw[j];
  }
}

static constexpr int kBktE = 2048, kBktS = 4096;                 // boundaries per bucket (ends array) / starts-array slice in LDS
static constexpr int kCellsE = 2048, kCellsS = 4096;             // cells of the direct-address tables

// rank of `key` among the sorted boundaries v[0..n): #{v < key} (LE = false) or #{v <= key} (true), through the table
// tab[c] = first index whose value lies in cell >= c (cell(x) = (x - lo) >> sh, clamped to [0, cells)): two table reads bound
// the search to the boundaries of one cell
template <bool LE>
__device__ __forceinline__ int table_rank(const int *__restrict__ v, const unsigned short *__restrict__ tab, int cells, int lo, int sh, int key)
{
  const i64 d = (i64)key - lo;
  const int c = d < 0 ? 0 : (int)((d >> sh) < cells - 1 ? (d >> sh) : cells - 1);
  int a = tab[c], b = tab[c + 1];
  while (a < b) { const int mid = (a + b) >> 1; if (LE ? v[mid] <= key : v[mid] < key) a = mid + 1; else b = mid; }
  return a;
}

// tab[0..cells]: tab[c] = #{v_i : cell(v_i) < c}; tab[cells] = n
__device__ __forceinline__ void build_table(const int *__restrict__ v, int n, unsigned short *__restrict__ tab, int cells, int lo, int sh)
{
  for (int c = threadIdx.x; c <= cells; c += blockDim.x) {
    int a = 0, b = n;
    if (c == cells) a = n;
    else while (a < b) {
      const int mid = (a + b) >> 1;
      const i64 d = (i64)v[mid] - lo;
      const i64 cm = (d >> sh) < cells - 1 ? (d >> sh) : cells - 1;    // cell of v[mid] (d >= 0: lo is the smallest value)
      if (cm < c) a = mid + 1; else b = mid;
    }
    tab[c] = (unsigned short)a;
  }
}

__device__ __forceinline__ int shift_for(int lo, int hi, int cells)
{
  const u64 span = (u64)((i64)hi - lo);
  int sh = 0;
  while ((span >> sh) >= (u64)cells) sh++;
  return sh;
}

// (1024 threads, 4 reads in flight per thread: the loop is a chain global load -> table -> search -> atomic, and what bounds
// the kernel is how many of those chains a CU has open)
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_count_kernel(CountArgs a, BucketTable t, BucketWork w, int splits)
{
  __shared__ int sE[kBktE], sS[kBktS];
  __shared__ unsigned short tE[kCellsE + 2], tS[kCellsS + 2];
  typedef typename std::conditional<WEIGHTED, u64, unsigned>::type hist_t;   // a block sees < 2^32 reads
  __shared__ hist_t hA[kBktE + 1], hB[kBktS + 1];
  const int b = blockIdx.x / splits, k = blockIdx.x % splits;
  const unsigned off = w.offset[b], cntB = w.offset[b + 1] - off;
  const unsigned r0 = off + (unsigned)((u64)cntB * k / splits), r1 = off + (unsigned)((u64)cntB * (k + 1) / splits);
  if (r0 == r1) return;
  const int eLo = t.eLo[b], nE = t.eHi[b] - eLo, sLo = t.sLo[b], sHi = t.sHi[b], nS = sHi - sLo, cls = t.cls[b];
  const int segEnd = a.segStart[cls + 1];
  for (int i = threadIdx.x; i < nE; i += blockDim.x) sE[i] = a.sortedE[eLo + i];
  for (int i = threadIdx.x; i < nS; i += blockDim.x) sS[i] = a.sortedS[sLo + i];
  for (int i = threadIdx.x; i <= nE; i += blockDim.x) hA[i] = 0;
  for (int i = threadIdx.x; i <= nS; i += blockDim.x) hB[i] = 0;
  __syncthreads();
  const int loE = nE ? sE[0] : 0, shE = nE ? shift_for(loE, sE[nE - 1], kCellsE) : 0;
  const int loS = nS ? sS[0] : 0, shS = nS ? shift_for(loS, sS[nS - 1], kCellsS) : 0;
  build_table(sE, nE, tE, kCellsE, loE, shE);
  build_table(sS, nS, tS, kCellsS, loS, shS);
  __syncthreads();
  const int2 *__restrict__ reads = (const int2 *)w.tmpReads;
  const unsigned cnt = r1 - r0;
  constexpr int U = 4;
  for (unsigned at = threadIdx.x; at < cnt; at += U * blockDim.x) {
    int2 se[U]; int wt4[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned j = at + u * blockDim.x;
      se[u] = make_int2(0, -1); wt4[u] = 1;
      if (j < cnt) { se[u] = reads[r0 + j]; if (WEIGHTED) wt4[u] = w.tmpWeights[r0 + j]; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (at + u * blockDim.x >= cnt) break;
      const u64 wt = WEIGHTED ? (u64)(i64)wt4[u] : 1;
      const int slotA = table_rank<false>(sE, tE, kCellsE, loE, shE, se[u].x);           // #{E < s} inside the slice
      atomicAdd(&hA[slotA], (hist_t)wt);
      const int lo = table_rank<true>(sS, tS, kCellsS, loS, shS, se[u].y);               // #{S <= e} inside the slice
      if (lo < nS || sHi == segEnd) atomicAdd(&hB[lo], (hist_t)wt);
      else {                                                        // the read ends beyond the slice: global search above it
        int glo = sHi, ghi = segEnd;
        while (glo < ghi) { const int mid = (int)(((i64)glo + ghi) >> 1); if (a.sortedS[mid] <= se[u].y) glo = mid + 1; else ghi = mid; }
        atomicAdd(&a.histB[(i64)glo + cls], wt);
      }
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
  const size_t ldsHist = sizeof(int) * (2 * (size_t)t.nB + (size_t)a.nClasses + 2);
  i64 blocks = (n + 1023) / 1024; if (blocks > 2048) blocks = 2048;
  if (weights) bucket_hist_kernel<true><<<(unsigned)blocks, 1024, ldsHist, st>>>((const Tri3 *)reads, (const int *)weights, n, a, t, w);
  else bucket_hist_kernel<false><<<(unsigned)blocks, 1024, ldsHist, st>>>((const Tri3 *)reads, (const int *)weights, n, a, t, w);
  bucket_scan_kernel<<<1, 1024, 0, st>>>(t, w);
  // tile of the split: as large as the LDS allows next to the per-bucket tables (longer bursts per bucket)
  const size_t tables = sizeof(int) * (4 * (size_t)t.nB + (size_t)a.nClasses + 4);
  const size_t perRead = 8 + 2 + (weights ? 4 : 0);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipSuccess;
    const void *fn[] = {(const void *)bucket_split_kernel<false, 8192>, (const void *)bucket_split_kernel<true, 8192>, (const void *)bucket_split_kernel<false, 4096>,
                        (const void *)bucket_split_kernel<true, 4096>, (const void *)bucket_split_kernel<false, 2048>, (const void *)bucket_split_kernel<true, 2048>};
    for (const void *f : fn) if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const size_t budget = 150 * 1024;
  // (100 M shuffled reads, 540 buckets: tile 8192 = 1 block per CU 0.78 ms, 4096 = 3 blocks per CU 0.73 ms, 2048 = 1.37 ms: the bursts
  // get too short; what the kernel waits for is the chain load -> LDS rank -> barrier -> reservation -> barrier -> copy of ONE tile per
  // block, so blocks per CU count as much as burst length)
  static const int tbMax = getenv("GTX_SPLIT_TILE") ? atoi(getenv("GTX_SPLIT_TILE")) : 4096;
  int tb = tbMax >= 8192 ? 8192 : tbMax >= 4096 ? 4096 : 2048;
  while (tb > 2048 && tables + (size_t)tb * perRead > budget) tb >>= 1;
  const size_t ldsSplit = tables + (size_t)tb * perRead + 16;
  const unsigned sblocks = (unsigned)((n + tb - 1) / tb);
#define GTX_SPLIT(W, TBV) bucket_split_kernel<W, TBV><<<sblocks, 1024, ldsSplit, st>>>((const Tri3 *)reads, (const int *)weights, n, a, t, w)
  if (weights) { if (tb == 8192) GTX_SPLIT(true, 8192); else if (tb == 4096) GTX_SPLIT(true, 4096); else GTX_SPLIT(true, 2048); }
  else { if (tb == 8192) GTX_SPLIT(false, 8192); else if (tb == 4096) GTX_SPLIT(false, 4096); else GTX_SPLIT(false, 2048); }
#undef GTX_SPLIT
  // blocks of ~64k reads on average, at least one per bucket
  static const i64 perBlock = getenv("GTX_COUNT_BLOCK_READS") ? atoll(getenv("GTX_COUNT_BLOCK_READS")) : 65536;   // 16 k: 0.48 ms, 32 k: 0.37, 64 k: 0.34 (table build per block)
  i64 splits = (n + (i64)t.nB * perBlock - 1) / ((i64)t.nB * perBlock);
  if (splits < 1) splits = 1;
  if (splits > 512) splits = 512;
  if (weights) bucket_count_kernel<true><<<(unsigned)(t.nB * splits), 1024, 0, st>>>(a, t, w, (int)splits);
  else bucket_count_kernel<false><<<(unsigned)(t.nB * splits), 1024, 0, st>>>(a, t, w, (int)splits);
  return hipGetLastError();
}

}  // namespace gtx
