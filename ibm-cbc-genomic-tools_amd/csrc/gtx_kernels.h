// gtx_kernels.h -- launch interface between the C ABI (gtx_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

namespace gtx {

// Function attributes (dynamic LDS limits) and device properties are per DEVICE: a gtx_group drives several devices from one
// process.  once(fn) runs fn the first time it is reached with a given current device, under a lock.
struct PerDevice {
  std::mutex m; unsigned long long done = 0;
  template <class F> hipError_t once(F fn)
  {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(m);
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1)) return hipSuccess;
    e = fn();
    if (e == hipSuccess && dev >= 0 && dev < 64) done |= 1ull << dev;
    return e;
  }
};

// device-side mirror of gtx_count_info (include/gtx.h)
struct DevInfo {
  long long first_unsorted;
  long long n_no_class;
  long long n_degenerate;
  long long first_degenerate;
  long long n_unplaced;          // inverted reads that did not fit the side buffer (counts incomplete: the caller must fail)
  long long fault;               // a kernel gave up a bounded wait (finalize_scan_chained_kernel): the call's result is void
};

// Direct placement of a wave's two windows at the start of its span: the positions of a class are cut into cells of
// 2^shift, rank[2 * (first cell of the class + cell)] / [.. + 1] = how many boundaries of the ends / starts array lie before
// the cell's first position (as global indices into the sorted arrays).  Read with scalar loads.
struct PlaceTable {
  const int4 *cls;               // [nClasses] {segment start, segment end, first cell, cells}
  const int *rank;               // [2 * cells]
  int shift;
};

// How the read stream is dealt to the waves of a launch: segment k = the waves from wave0[k] up to the next segment take cpw[k]
// chunks each, beginning at chunk chunk0[k].  Workgroups are dispatched in grid order, so the segments are phases in time: a
// head of spans of growing length (the first round's waves all start together; equal spans would also end together, and again
// a round later), the main segment, and a tail of spans that shrink in step with the time the launch has left.
struct SpanSchedule {
  static constexpr int kMax = 32;
  int wave0[kMax];               // INT32_MAX for the segments not in use
  int chunk0[kMax];
  int cpw[kMax];
  int nSeg;
  long long nWaves;              // all segments together (the waves a grid has beyond them find their span behind the stream's end)
};

struct CountArgs {
  const int *sortedE;            // reference ends, sorted by (class, value)
  const int *sortedS;            // reference starts, sorted by (class, value)
  const int *segStart;           // [nClasses+1] class segments in both arrays
  unsigned long long *histA;     // [nValid + nClasses] rank histogram over sortedE
  unsigned long long *histB;     // same over sortedS
  unsigned long long *partA;     // per-tile sums of histA / histB (1024 slots per tile)
  unsigned long long *partB;
  DevInfo *info;
  int nClasses;
  const unsigned char *owned;    // [nClasses] (may be null): a group member's classes -- a read of any other class is counted like a read of no class
                                 //   (n_no_class) and nowhere else: its tiles are not in the member's share of the finalize step
  int chunksPerWave;             // 64-read chunks one wave streams
  int checkSorted;               // verify (class >> sortClassShift, start) order
  int sortClassShift;
  int prefetch;                  // reads per lane per step of the streaming kernel (1..4)
  int zeroLenOk;                 // start == end+1 is a countable read (sorted-merge semantics)
  const int *sampE;              // every (1 << sampShift)-th element of sortedE / sortedS: the
  const int *sampS;              //   search kernel keeps them in LDS as the top level of its searches
  int sampShift, nSamp;
  const int *topE;               // every 256th element of sortedE / sortedS (index i << 8): first hop of the streaming
  const int *topS;               //   kernel's start-of-span search (rank_pair)
  PlaceTable place;
  SpanSchedule sched;
  int flip;                      // streaming kernel: meet all boundaries of a window at once (dense references)
  int hist32;                    // streaming kernel (unweighted, R = 4): histA / histB are read and written as unsigned[] -- the caller's finalize step too
                                 //   (launch_finalize(..., hist32)).  For calls of fewer than 2^32 reads in ONE launch: no slot or prefix can pass that
  long long indexBase;           // position of reads[0] in the caller's stream: added to the indices reported in `info`
  // sorted-merge semantics (zeroLenOk): inverted reads (start > end + 1) are not degenerate there -- the merge matches them
  // by its two comparisons like any other read (genomic_intervals.cpp:1225-1236).  The rank difference does not hold for
  // them, so the kernels set them aside here (class, start, end, weight) for the pair kernels of gtx_special.hip.
  int4 *side; unsigned *sideCount; int sideCap;
  int keyCenter;                 // the partition pass of the scan path with preprocess 'c': a read is placed by its centre start + (end - start) / 2
  int coverRule;                 // the partition pass of the coverage path: a zero-length read (start == end + 1) is dropped silently, only
                                 // inverted ones (start > end + 1) are reported and set aside -- as coverage_walk_kernel does
#ifdef GTX_WAVE_TRACE
  unsigned long long *trace;     // diagnostic build only (make trace): 4 words per wave -- start, windows placed, end (100 MHz ticks), XCC id
#endif
};

// coverage: ONE boundary array per class -- the thresholds E_k and S_k - 1 of all its regions, merged and sorted (sortedT) --
// walked by two windows, keyed by the read starts and by the read ends; 4 histograms / tile-sum arrays over its slots:
//   0 (key s, w)   1 (key s, w*s)   2 (key e, w)   3 (key e, w*e)
struct CoverArgs {
  const int *sortedT, *segStartT;  // thresholds sorted by (class, value); [nClasses+1] class segments
  const int *topT;                 // every 256th threshold (as CountArgs::topE)
  PlaceTable place;                // over sortedT (both entries of a cell are the same rank: the two windows walk one array)
  SpanSchedule sched;
  unsigned long long *hist[4], *part[4];
  DevInfo *info;
  int nClasses, chunksPerWave;
  int wfast;                       // weighted reads: steps of 4 x 64 with prefix sums in LDS (0: general per-chunk code only)
  long long indexBase;             // as CountArgs
  int4 *side; unsigned *sideCount; int sideCap;   // as CountArgs (only the -gaps coverage formula needs them)
};

struct CoverGather {
  unsigned long long *pref[4], *part[4];
  const int *posTE, *posTS, *classBaseT;   // slot of E_k / of S_k - 1 / just below the class's first slot, per region in FILE order
  const int *refS, *refE;        // region coordinates in FILE order
};

struct ScanArgs {
  unsigned long long *micro;     // micro-window histogram, all classes: uint64 when weighted, uint32 (same buffer) otherwise
  const long long *microOff;     // [nClasses] offset of class c in micro
  const long long *nMicro;       // [nClasses] micro-windows of class c (len / step)
  const long long *winOff;       // [nClasses+1] prefix of window counts (launch order)
  const long long *outOff;       // [nClasses] caller's class_offsets
  const long long *tileOff;      // [nClasses+1] prefix of window tiles (scan_window_tile() windows each)
  int nClasses;
  int winStep;
  unsigned winStepInv;           // floor(2^32 / winStep) (unused for winStep == 1)
  int comb;                      // win_size / win_step
  int center;                    // preprocess 'c'
  int sortedRule;                // sorted scanner: no validity test, pos < 1 lands in the first micro-window
};

// ---- unsorted reads: partition into buckets, count per bucket in LDS (gtx_bucket.hip) ----
struct BucketTable {               // built by gtx_set_refs; a bucket = <= bucket_e_size() consecutive boundaries of ONE class
  const int *posHi;                // [nB] largest read start the bucket takes (INT_MAX for the last bucket of its class)
  const int *eLo, *eHi;            // [nB] its range of the ends array; histogram A slots eLo+cls .. eHi+cls
  const int *sLo, *sHi;            // [nB] the slice of the starts array kept in LDS; histogram B slots sLo+cls .. sHi+cls
  const int *cls;                  // [nB]
  const int *clsStart;             // [nClasses+1] buckets of each class
  int nB;
  // direct-address lookup of a read's bucket: per class {first cell, lowest cut, cells, first bucket}; a cell is 2^cellShift
  // positions wide; cellTab[cell] = first bucket of the class whose posHi is not below the cell's first position
  const int4 *clsCell;             // [nClasses]
  const unsigned short *cellTab;   // [nCells]
  int nCells, cellShift;
};
struct BucketWork {                // scratch of one call (sizes: bucket_plan)
  void *tmpReads; int *tmpWeights; // [pairs] (start, end) [, weight] of the reads, in chunks of 64 of one bucket; an arena per block
  unsigned arenaPairs;             // pairs per arena
  unsigned *dir;                   // [chunks] chunk -> bucket | fill << 16
  unsigned *list;                  // [chunks] the chunks bucket by bucket: chunk << 6 | (fill - 1)
  unsigned *chunkCount;            // [nB * blocks] chunks of bucket b in arena k, then their exclusive prefix inside the bucket's row
  unsigned *rowOff;                // [nB + 1] first list entry of every bucket
  unsigned *arenaUsed;             // [blocks] chunks dealt out
};
struct BucketPlan { int per, line; unsigned blocks; size_t arenaPairs, pairs, chunks, matrix; };   // line: pairs per whole line of the scatter pass (0: as they come)
BucketPlan bucket_plan(long long n, int nClasses, int nB, int nCells, bool weighted);
bool bucket_tables_fit(int nClasses, int nB, int nCells);
int bucket_e_size();
int bucket_s_size();
int bucket_t_size();                 // coverage: entries of the threshold array a bucket keeps in LDS
hipError_t launch_count_bucketed(const void *reads, const void *weights, long long n, const CountArgs &a, const BucketTable &t,
                                 const BucketWork &w, const BucketPlan &p, hipStream_t st);
// coverage of reads in no particular order: the same partition over a bucket table of the THRESHOLD array (cuts and slices both
// in sortedT), then per bucket the four histograms of CoverArgs in LDS.  The tile sums are NOT kept: launch_tile_sums before the
// finalize step.
hipError_t launch_cover_bucketed(const void *reads, const void *weights, long long n, const CountArgs &a, const CoverArgs &cv, const BucketTable &t,
                                 const BucketWork &w, const BucketPlan &p, hipStream_t st);
// genomic_scans counts of reads in no particular order (unsorted rule, preprocess '1'): the same partition over a bucket table of
// POSITION cuts (a bucket = a run of consecutive micro-windows of one class: eLo/eHi hold its first / end micro-window), then every
// part -- (bucket, first micro-window, micro-windows) -- counts the bucket's reads that fall into it in LDS and adds its counters to
// the micro-window histogram of ScanArgs (which the caller has zeroed, as for launch_scan_hist).
struct ScanPart { int bucket, first, count, pad; };
int scan_part_bins(bool weighted);     // micro-windows one part keeps in LDS
hipError_t launch_scan_bucketed(const void *reads, const void *weights, long long n, const CountArgs &a, const ScanArgs &sc, const BucketTable &t,
                                const BucketWork &w, const BucketPlan &p, const ScanPart *parts, int nParts, hipStream_t st,
                                unsigned long long *out = nullptr);
// out (may be null): the caller's window vector -- the parts then write their windows themselves (sliding sums from the LDS histogram,
// the windows across a part's edge by atomics on cleared places): no micro-window array, no window pass.  comb <= scan_fused_max_comb().
int scan_fused_max_comb();
hipError_t launch_tile_sums(unsigned long long *histA, unsigned long long *histB, long long histLen, unsigned long long *tileA, unsigned long long *tileB, hipStream_t st);

// ---- intervals the rank difference does not cover (gtx_special.hip): plain pair tests, reference semantics of the sorted merge
// value of a matching pair: mode 0 = w (count), mode 2 = w x (min(ends) - max(starts) + 1), the unclamped -gaps formula of
// CalcIndexCoverage (genomic_intervals.cpp:5278)
hipError_t launch_special_refs(const void *reads, const void *weights, long long n, const int4 *refs, int nSpecial, int mode,
                               unsigned long long *out, hipStream_t st);                 // out[j] += sum over reads matching special ref j
hipError_t launch_side_reads(const int *refC, const int *refS, const int *refE, long long m, const int4 *side, const unsigned *sideCount,
                             int sideCap, int mode, unsigned long long *hits, DevInfo *info, hipStream_t st);   // hits[k] += sum over side reads matching ref k
hipError_t launch_special_scatter(const int *specialIdx, unsigned long long *specialOut, int nSpecial, unsigned long long *hits,
                                  unsigned *sideCount, hipStream_t st);                  // hits[idx[j]] = out[j]; out and the side counter cleared

SpanSchedule span_schedule(long long nChunks, int cpw, int r, long long slots);
int search_sample_shift(long long nValid);   // stride of the sample arrays such that both fit the LDS budget
int scan_tiles(long long len);

hipError_t launch_count(const void *reads, const void *weights, long long n, const CountArgs &a, bool sortedHint, hipStream_t st);
// tileSumsValid: the streaming kernel kept tileA/tileB up to date (the search kernel does not).
// Leaves histA/histB and the tile sums zeroed for the next call.
// share (may be null): the finalize step of a group member -- only the histogram tiles that cover the classes it owns (a class
// has slots seg+cls-1 .. segEnd+cls; other tiles hold no counts) and only its regions, written in the group's compact order
struct FinalizeShare { const int *tileList; int nTiles; const int *regionList; long long nRegions; bool scatter; };   // scatter: hits[region] instead of hits[place in the list]
hipError_t launch_finalize(unsigned long long *histA, unsigned long long *histB, long long histLen,
                           unsigned long long *tileA, unsigned long long *tileB, bool tileSumsValid,
                           unsigned long long *prefA, unsigned long long *prefB,
                           const int *posE, const int *posS, const int *classBase, long long m,
                           unsigned long long *hits, DevInfo *nextInfo, hipStream_t st, const FinalizeShare *share = nullptr,
                           unsigned *chainFlags = nullptr, unsigned epoch = 0, DevInfo *info = nullptr, unsigned long long *chainDraws = nullptr,
                           bool hist32 = false);
// chainFlags (may be null): 8 x (tiles + 2) words (two 64-bit words per tile and histogram; the word behind each histogram's is the kernel's ticket counter), zero when
// made, never written by the caller; epoch: a value no earlier call on these flags used (and not 0); info: the call's block (DevInfo::fault); chainDraws: the host's count of
// the tickets drawn from these flags so far (0 when they are made; the launcher advances it).  With them a call whose tile sums are not valid and whose tiles are few
// enough takes one launch for tile sums + scan (finalize_scan_chained_kernel).
hipError_t launch_coverage(const void *reads, const void *weights, long long n, const CoverArgs &a, hipStream_t st);
hipError_t launch_coverage_finalize(const CoverArgs &a, long long histLen, const CoverGather &g, long long m,
                                    unsigned long long *cov, DevInfo *nextInfo, hipStream_t st);
// runIf (may be NULL): device flag; the kernels leave at once when it is 0 -- the general scan as the fallback of the owner-computes pass
hipError_t launch_scan_zero(void *micro, long long bytes, const int *runIf, hipStream_t st);
hipError_t launch_scan_hist(const void *reads, const void *weights, long long n, const ScanArgs &a, hipStream_t st, const int *runIf = nullptr);
int scan_window_tile();
hipError_t launch_scan_windows(const void *micro, bool micro64, const ScanArgs &a, long long totalTiles,
                               unsigned long long *out, hipStream_t st, const int *runIf = nullptr);

// ---- genomic_scans counts on reads sorted by (class, start): every block OWNS a range of windows (gtx_scanown.hip) ----
struct ScanOwn {
  const long long *blkOff;       // [nClasses+1] prefix of blocks per class (tile windows each)
  int tile;                      // windows per block
  long long totalBlocks;
  long long *bounds;             // [2 * totalBlocks + 2] scratch: first read of each block's own range | end of its reads
  int *flag;                     // set to 1 when the reads turn out not to be in order (the caller falls back to the general kernels)
};
int scan_own_tile(long long nReads, long long totalMicro, int comb);   // windows per block, 0 = the owner pass does not apply
hipError_t launch_scan_own(const void *reads, const void *weights, long long n, const ScanArgs &a, const ScanOwn &o,
                           unsigned long long *out, hipStream_t st);

} // namespace gtx
