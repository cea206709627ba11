// Multi-interval (BED12) regions in `count` without -gaps: a query counts ONCE for an index region when their envelopes overlap
// (what GetMatch / NextMatch deliver, genomic_intervals.cpp:5717-5760, :5903-5930) and some interval of the one overlaps some
// interval of the other (GenomicRegion::OverlapsWith, :1167-1172).  "Some pair" is not a sum over intervals, so the rank closed
// form of the streaming kernel cannot express it; what it CAN do is count every pair on the envelopes.  The two kernels here
// settle the difference pair by pair, and only for pairs with a multi-interval side:
//   * reads with one interval against the multi-interval index regions: the pairs whose envelopes overlap while the read lies
//     in a gap of the region are taken off again (pair_miss);
//   * reads with several intervals never enter the streaming kernel: their pairs with ALL index regions are found here and
//     added (pair_hit).
// Candidates come from the envelopes of the index regions in the order of their starts, per class: a binary search for the last
// start <= the query's end, then a walk down that ends where the running maximum of the ends falls below the query's start and
// skips 64 entries at a time where none of them reaches it.
#pragma once
#include <hip/hip_runtime.h>

namespace gtx {

struct PairIndex {
  const int *seg;      // [nClasses + 1]: class c's entries are [seg[c], seg[c+1])
  const int *start;    // envelope starts, ascending within a class
  const int *end;      // envelope ends, same order
  const int *pmax;     // max of end over the class's entries up to and including this one
  const int *bmax;     // max of end over entries [64 k, 64 k + 64) (all classes: only ever used to skip)
  const int *id;       // region ordinal (position in the caller's set)
  int nClasses;
};

// intervals of the multi-interval regions: blkOf[ordinal] = {first, count} into iv[] ({start, stop} pairs, starts and stops both
// non-decreasing); count == 0: a single-interval region, its interval is its envelope.  blkOf == nullptr: every region is single.
struct RegionBlocks { const int2 *blkOf; const int2 *iv; };

// reads: (class, start, stop) triples, weights may be null (1).  sub[ordinal] += w for every pair (read, region of ix) whose
// envelopes overlap while no interval of the region overlaps the read.
hipError_t launch_pair_miss(const void *reads, const void *weights, long long n, const PairIndex &ix, const RegionBlocks &rb,
                            unsigned long long *sub, hipStream_t st);

// q[i] = {class, envelope start, envelope stop, weight}, qBlk[i] = {first, count} into qIv.  add[ordinal] += w for every pair
// (query, region of ix) with overlapping envelopes and an overlapping pair of intervals.
hipError_t launch_pair_hit(const int4 *q, const int2 *qBlk, const int2 *qIv, long long nq, const PairIndex &ix, const RegionBlocks &rb,
                           unsigned long long *add, hipStream_t st);

// out[k] += add[k] - sub[k]; add and sub are left zeroed
hipError_t launch_pair_apply(unsigned long long *out, unsigned long long *add, unsigned long long *sub, long long m, hipStream_t st);

}  // namespace gtx
