// Pair kernels for multi-interval regions in `count` without -gaps (see gtx_pairs.h).  One lane per query: these pairs are a side
// channel of the counting path (the reads with one interval against single-interval regions stay with the streaming kernel), so
// the kernels are written for exactness first: the candidate walk and the interval test follow the reference's predicates
// (genomic_intervals.cpp:5752 envelope test, :1167-1172 any pair of intervals, GenomicInterval::OverlapsWith on each pair).
#include "gtx_pairs.h"

namespace gtx {
namespace {

typedef unsigned long long u64;

// does [s, e] overlap an interval of the list iv[0..n)?  starts and stops of the list are non-decreasing, so the intervals with
// stop >= s are a suffix and those with start <= e a prefix: they meet iff the first of the suffix is inside the prefix
__device__ __forceinline__ bool overlaps_list(const int2 *__restrict__ iv, int n, int s, int e)
{
  int a = 0, b = n;
  while (a < b) { const int m = (a + b) >> 1; if (iv[m].y >= s) b = m; else a = m + 1; }
  return a < n && iv[a].x <= e;
}

template <bool MULTI_Q>
__global__ __launch_bounds__(256) void pair_kernel(const int *__restrict__ reads, const int *__restrict__ weights, const int4 *__restrict__ q,
                                                   const int2 *__restrict__ qBlk, const int2 *__restrict__ qIv, long long n, PairIndex ix,
                                                   RegionBlocks rb, u64 *__restrict__ acc)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  int cls, qs, qe, w; int2 qb = make_int2(0, 0);
  if (MULTI_Q) { const int4 v = q[t]; cls = v.x; qs = v.y; qe = v.z; w = v.w; qb = qBlk[t]; }
  else { cls = reads[3 * t]; qs = reads[3 * t + 1]; qe = reads[3 * t + 2]; w = weights ? weights[t] : 1; }
  if (cls < 0 || cls >= ix.nClasses) return;
  const int lo = ix.seg[cls], hi = ix.seg[cls + 1];
  int a = lo, b = hi;
  while (a < b) { const int m = (a + b) >> 1; if (ix.start[m] <= qe) a = m + 1; else b = m; }   // entries [lo, a) start at or before qe
  for (int i = a - 1; i >= lo;) {
    if (ix.pmax[i] < qs) break;                                                                  // nothing further down reaches the query
    if ((i & 63) == 63 && i - 63 >= lo && ix.bmax[i >> 6] < qs) { i -= 64; continue; }
    if (ix.end[i] >= qs) {
      const int r = ix.id[i];
      const int2 blk = rb.blkOf ? rb.blkOf[r] : make_int2(0, 0);
      bool hit = false;
      if (MULTI_Q) {
        for (int k = 0; k < qb.y && !hit; k++) {
          const int2 qi = qIv[qb.x + k];
          hit = blk.y ? overlaps_list(rb.iv + blk.x, blk.y, qi.x, qi.y) : (ix.start[i] <= qi.y && ix.end[i] >= qi.x);
        }
        if (hit) atomicAdd(acc + r, (u64)(long long)w);
      } else {
        hit = blk.y ? overlaps_list(rb.iv + blk.x, blk.y, qs, qe) : true;
        if (!hit) atomicAdd(acc + r, (u64)(long long)w);
      }
    }
    i--;
  }
}

__global__ void pair_apply_kernel(u64 *__restrict__ out, u64 *__restrict__ add, u64 *__restrict__ sub, long long m)
{
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const u64 a = add[k], s = sub[k];
  if (a | s) { out[k] = out[k] + a - s; add[k] = 0; sub[k] = 0; }
}

}  // namespace

hipError_t launch_pair_miss(const void *reads, const void *weights, long long n, const PairIndex &ix, const RegionBlocks &rb, u64 *sub, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const long long blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pair_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, (const int *)reads, (const int *)weights, (const int4 *)nullptr,
                     (const int2 *)nullptr, (const int2 *)nullptr, n, ix, rb, sub);
  return hipGetLastError();
}

hipError_t launch_pair_hit(const int4 *q, const int2 *qBlk, const int2 *qIv, long long nq, const PairIndex &ix, const RegionBlocks &rb, u64 *add, hipStream_t st)
{
  if (nq <= 0) return hipSuccess;
  const long long blocks = (nq + 255) / 256;
  if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pair_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, (const int *)nullptr, (const int *)nullptr, q, qBlk, qIv, nq, ix, rb, add);
  return hipGetLastError();
}

hipError_t launch_pair_apply(u64 *out, u64 *add, u64 *sub, long long m, hipStream_t st)
{
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(pair_apply_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, out, add, sub, m);
  return hipGetLastError();
}

}  // namespace gtx
