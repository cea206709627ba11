// gtx_special.hip -- the pairs the rank difference of gtx_kernels.hip does not cover.
//
// SortedGenomicRegionSetOverlaps (gtools/genomic_intervals.cpp:5807-5937) never validates an interval: a query q and an
// index region r are a match exactly when the two comparisons of CalcDirection (:1225-1236) both fail,
//     NOT (r.stop < q.start)  and  NOT (q.stop < r.start)        i.e.   q.start <= r.stop  and  q.stop >= r.start,
// whatever the order of start and stop inside either interval (checked against the restated merge on random sorted
// inputs with inverted and zero-length intervals, in the CPU test suite).  For intervals with start <= stop + 1
// that count is a difference of two ranks (gtx_kernels.hip); for INVERTED ones (start > stop + 1) it is a genuine
// two-sided condition.  Such intervals are input errors in practice and rare, so they are matched pair by pair:
//
//   special_refs_kernel   every read of a batch (inverted ones included) x the K inverted reference regions
//   side_reads_kernel     every other reference region x the inverted reads the counting kernels set aside (CountArgs::side)
//   special_scatter_kernel  results of the first into their places of the output vector
//
// Work is N x K + M x K': nothing when there are no such intervals (the second kernel reads the side counter and leaves).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gtx_kernels.h"

namespace gtx {

typedef unsigned long long u64;
typedef long long i64;
struct __attribute__((packed, aligned(4))) Tri4 { int c, s, e; };

// what a matching (read, region) pair adds
template <int MODE>
__device__ __forceinline__ i64 pair_value(int qs, int qe, int rs, int re, int w)
{
  if (MODE == 0) return (i64)w;
  const i64 hi = qe < re ? qe : re, lo = qs > rs ? qs : rs;
  return (hi - lo + 1) * (i64)w;                                 // -gaps formula: not clamped (genomic_intervals.cpp:5278)
}

template <int MODE>
__global__ __launch_bounds__(256) void special_refs_kernel(const Tri4 *__restrict__ reads, const int *__restrict__ weights, i64 n,
                                                           const int4 *__restrict__ refs, int nSpecial, u64 *__restrict__ out)
{
  __shared__ int4 tile[256];
  __shared__ u64 acc[256];
  for (int j0 = 0; j0 < nSpecial; j0 += 256) {
    const int nj = nSpecial - j0 < 256 ? nSpecial - j0 : 256;
    __syncthreads();
    if ((int)threadIdx.x < nj) tile[threadIdx.x] = refs[j0 + threadIdx.x];
    acc[threadIdx.x] = 0;
    __syncthreads();
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
      const Tri4 q = reads[i];
      const int w = weights ? weights[i] : 1;
      for (int j = 0; j < nj; j++) {
        const int4 r = tile[j];
        if (q.c == r.x && q.s <= r.z && q.e >= r.y) atomicAdd(&acc[j], (u64)pair_value<MODE>(q.s, q.e, r.y, r.z, w));
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < nj && acc[threadIdx.x]) atomicAdd(&out[j0 + threadIdx.x], acc[threadIdx.x]);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void side_reads_kernel(const int *__restrict__ refC, const int *__restrict__ refS, const int *__restrict__ refE, i64 m,
                                                         const int4 *__restrict__ side, const unsigned *__restrict__ sideCount, int sideCap,
                                                         u64 *__restrict__ hits, DevInfo *info)
{
  const unsigned total = *sideCount;
  if (total == 0) return;
  const unsigned nSide = total < (unsigned)sideCap ? total : (unsigned)sideCap;
  if (total > nSide && blockIdx.x == 0 && threadIdx.x == 0) info->n_unplaced = (i64)(total - nSide);
  __shared__ int4 tile[256];
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  int c = -1, rs = 0, re = 0;
  if (k < m) { c = refC[k]; rs = refS[k]; re = refE[k]; }
  const bool inverted = (i64)rs > (i64)re + 1;                  // those are the first kernel's
  u64 sum = 0;
  for (unsigned j0 = 0; j0 < nSide; j0 += 256) {
    __syncthreads();
    if (j0 + threadIdx.x < nSide) tile[threadIdx.x] = side[j0 + threadIdx.x];
    __syncthreads();
    const unsigned nj = nSide - j0 < 256 ? nSide - j0 : 256;
    if (c >= 0 && !inverted)
      for (unsigned j = 0; j < nj; j++) {
        const int4 q = tile[j];
        if (q.x == c && q.y <= re && q.z >= rs) sum += (u64)pair_value<MODE>(q.y, q.z, rs, re, q.w);
      }
  }
  if (sum) hits[k] += sum;
}

__global__ __launch_bounds__(256) void special_scatter_kernel(const int *__restrict__ idx, u64 *__restrict__ out, int nSpecial, u64 *__restrict__ hits,
                                                              unsigned *sideCount)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0 && sideCount) *sideCount = 0;
  if (j < nSpecial) { hits[idx[j]] = out[j]; out[j] = 0; }
}

hipError_t launch_special_refs(const void *reads, const void *weights, i64 n, const int4 *refs, int nSpecial, int mode, u64 *out, hipStream_t st)
{
  if (n <= 0 || nSpecial <= 0) return hipSuccess;
  i64 blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  if (mode == 0) special_refs_kernel<0><<<(unsigned)blocks, 256, 0, st>>>((const Tri4 *)reads, (const int *)weights, n, refs, nSpecial, out);
  else special_refs_kernel<2><<<(unsigned)blocks, 256, 0, st>>>((const Tri4 *)reads, (const int *)weights, n, refs, nSpecial, out);
  return hipGetLastError();
}

hipError_t launch_side_reads(const int *refC, const int *refS, const int *refE, i64 m, const int4 *side, const unsigned *sideCount, int sideCap,
                             int mode, u64 *hits, DevInfo *info, hipStream_t st)
{
  if (m <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)((m + 255) / 256);
  if (mode == 0) side_reads_kernel<0><<<blocks, 256, 0, st>>>(refC, refS, refE, m, side, sideCount, sideCap, hits, info);
  else side_reads_kernel<2><<<blocks, 256, 0, st>>>(refC, refS, refE, m, side, sideCount, sideCap, hits, info);
  return hipGetLastError();
}

hipError_t launch_special_scatter(const int *specialIdx, u64 *specialOut, int nSpecial, u64 *hits, unsigned *sideCount, hipStream_t st)
{
  const unsigned blocks = (unsigned)((nSpecial > 0 ? nSpecial : 1) + 255) / 256;
  special_scatter_kernel<<<blocks, 256, 0, st>>>(specialIdx, specialOut, nSpecial, hits, sideCount);
  return hipGetLastError();
}

}  // namespace gtx
