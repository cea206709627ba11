// gtx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the interval-overlap engine.
//
// What is computed (reference: GenomicRegionSetOverlaps::CountIndexOverlaps,
// gtools/genomic_intervals.cpp:5304-5317 over GetOverlap :5224-5248 and
// GenomicInterval::OverlapsWith :624-630), for single-interval regions inside one class
// (chromosome, or chromosome+strand):
//
//     hits[k] = sum_q w_q * [s_q <= E_k  and  e_q >= S_k]
//             = sum_q w_q [s_q <= E_k]  -  sum_q w_q [e_q < S_k]          (s_q <= e_q, S_k <= E_k)
//
// so one streaming pass over the reads fills two histograms over *ranks*:
//     HA[#{E_j <  s_q}] += w_q       (E sorted ascending inside the class)
//     HB[#{S_j <= e_q}] += w_q       (S sorted ascending inside the class)
// and  hits[k] = prefix(HA)[posE(k)] - prefix(HB)[posS(k)].  Integer adds commute, so any
// execution order is bit-exact.
//
// The streaming kernel (count_walk_kernel) is wave-autonomous -- no barriers, LDS only in the variants named below: a
// wave64 owns a contiguous span of reads (<= 56 x 64) and takes them 4 x 64 at a time (register r of lane l = read
// 64r+l of the step: four coalesced non-temporal 768-byte requests).  The sorted boundaries of the class it is in live
// in a 63-slot window in ONE VGPR across the wave (next window prefetched).  For a boundary W (wave-uniform,
// v_readlane) the number of keys of the step at or below it is popcount(ballot(key <= W)): per boundary crossed a
// handful of compares and scalar popcounts (a hand-scheduled 24-instruction loop), nothing per read; a step that
// crosses no boundary costs two compares per array.  A span starts with ONE paired search for both windows through the
// every-256th-boundary arrays (rank_pair).  80 SGPRs / 58 VGPRs: 8 waves per SIMD.  The kernel is bound by the HBM
// read of the triples (DESIGN.md section 7 lists what else was in the way).  Any input order is handled exactly
// (backward walk, re-seek by wave-cooperative 64-ary search, lanes that add themselves by binary search); only the
// speed depends on the order.  Variants: count_walk_kernel_flip (dense references: all boundaries of a window at once,
// the keys of a step in wave-private LDS), count_walk_kernel_weighted (label weights: the same with the weights'
// prefix sums next to the keys).
// No MFMA anywhere: this is integer indexing, not a contraction.
//
// Kernels in this file: count_walk_kernel (dominant) and its two variants, count_search_kernel (order-agnostic, small
// batches; large unsorted batches take the bucket path of gtx_bucket.hip), coverage_walk_kernel (CalcIndexCoverage, one
// launch per boundary array), tile_sums/finalize_scan/gather_hits/gather_coverage (prefix + gather), scan_hist_kernel +
// scan_window_kernel (genomic_scans counts).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <type_traits>
#include <vector>
#include <algorithm>
#include "gtx_kernels.h"

namespace gtx {

typedef unsigned long long u64;
typedef long long i64;

static constexpr int kHi = 0x7fffffff;          // +inf sentinel (coordinates are < 2^31-1)
static constexpr int kLo = (int)0x80000000;     // -inf sentinel
static constexpr int kSlots = 63;               // slots per register window (lane 0 holds the boundary below slot 0)
static constexpr int kTileShift = 10;           // finalize scan tile = 1024 histogram slots
static constexpr int kTile = 1 << kTileShift;
static constexpr int kSearchLdsBytes = 128 * 1024;  // LDS top level of the search kernel (both arrays)

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// value of the lane below (lane 0 keeps its own); DPP wave_shr:1 -- call with all lanes active
__device__ __forceinline__ int lane_prev(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }

// inclusive prefix sum over the 64 lanes, DPP only (no LDS): Hillis-Steele inside each row of 16 lanes
// (row_shr 1, 2, 4, 8 with zero fill), then row_bcast15 hands a row's total to the next odd row and
// row_bcast31 the total of lanes 0..31 to the upper half.  Call with all lanes active.
__device__ __forceinline__ int wave_scan_add(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// the same for 64-bit values: both halves travel by DPP, the add carries
__device__ __forceinline__ u64 wave_scan_add64(u64 v)
{
#define GTX_DPP64(ctrl, rmask, bc)                                                                                  \
  v += ((u64)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(v >> 32), ctrl, rmask, 0xf, bc) << 32) |                \
       (u64)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, ctrl, rmask, 0xf, bc)
  GTX_DPP64(0x111, 0xf, true); GTX_DPP64(0x112, 0xf, true); GTX_DPP64(0x114, 0xf, true); GTX_DPP64(0x118, 0xf, true);
  GTX_DPP64(0x142, 0xa, false); GTX_DPP64(0x143, 0xc, false);
#undef GTX_DPP64
  return v;
}

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
  return rfl(v);
}

__device__ __forceinline__ i64 wave_sum(i64 v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return __shfl(v, 0);
}

// ---------------------------------------------------------------------------------------------
// One boundary array being walked by one wave.  Uniform members live in SGPRs; W, Wn, acc are
// per-lane.  Lane L of W holds arr[base-1+L] (kLo below the class segment, kHi above it); slot j
// (global rank base+j) is bounded below by lane j (prevW) and above by lane j+1 (curW).
//
// A key x belongs to the slots at or below boundary W iff  below(x, W):
//   STRICT = false (ends array E, keys = read starts):  x <= W      rank = #{E_j <  x}
//   STRICT = true  (starts array S, keys = read ends):  x <  W      rank = #{S_j <= x}
// ---------------------------------------------------------------------------------------------
struct Seg { int start, end, cls; };            // class segment [start,end) of both boundary arrays

// H32: the histogram's slots are 32-bit (an unweighted count of fewer than 2^32 reads: no slot, and no prefix over the slots, can pass
// the number of reads -- half the bytes for the flushes here and for the finalize step to scan, clear and gather from)
template <bool WEIGHTED, bool STRICT, bool SUMKEY = false, bool H32 = false>
struct Win {
  static_assert(!H32 || (!WEIGHTED && !SUMKEY), "32-bit slots are for unweighted counts");
  typedef typename std::conditional<H32, unsigned, u64>::type hist_t;
  // SUMKEY (coverage): besides the weight of the keys that fall into a slot, the sum of weight x key
  // is kept in a second histogram (hist2 / part2 / acc2 / pend2)
  typedef typename std::conditional<WEIGHTED || SUMKEY, i64, unsigned>::type acc_t;
  const int *arr;       // sorted boundaries of all classes
  hist_t *hist;         // rank histogram, index = rank + class id
  u64 *part;            // per-tile (kTile slots) sums of hist, kept up to date for the finalize scan (nullptr: not kept)
  u64 *hist2, *part2;   // SUMKEY only
  i64 acc2, pend2;      // SUMKEY only
  int base;             // global rank of slot 0
  int j;                // current slot 0..62
  int prevW, curW;      // arr[base+j-1], arr[base+j]
  acc_t pend;           // weight pending for slot j
  int W, Wn;            // window and the prefetched next window
  acc_t acc;            // per-lane accumulator (lane L <-> slot L-1)

  static constexpr bool kStrict = STRICT;
  static __device__ __forceinline__ bool below(int x, int w) { return STRICT ? x < w : x <= w; }
  // boundary v sorts before key x  (v counts towards x's rank)
  static __device__ __forceinline__ bool before(int v, int x) { return STRICT ? v <= x : v < x; }

  __device__ __forceinline__ int load_window(const Seg &sg, int b, int lane) const
  {
    int idx = b - 1 + lane;
    int v = idx < sg.start ? kLo : kHi;
    if (idx >= sg.start && idx < sg.end) v = arr[idx];
    return v;
  }

  __device__ __forceinline__ void deposit(int lane)
  {
    if (pend != 0) { if (lane == j + 1) acc += pend; pend = 0; }
    if (SUMKEY) { if (pend2 != 0) { if (lane == j + 1) acc2 += pend2; pend2 = 0; } }
  }

  // publish per-lane accumulators `a` of the window whose slot 0 has rank b: one contiguous 64-lane atomic add into
  // the histogram, and the same amounts into the (at most two) tile sums the window touches
  __device__ __forceinline__ void flush_at(const Seg &sg, int lane, int b, acc_t a)
  {
    const i64 idx0 = (i64)b - 1 + sg.cls;                    // wave-uniform: histogram index of lane 0
    const i64 idx = idx0 + lane;
    if (a != 0) atomicAdd(&hist[idx], (hist_t)(i64)a);
    if (!part) return;                                       // few tiles: the finalize step rebuilds their sums (see count_args)
    if constexpr (sizeof(acc_t) == 4) {
      // a wave streams at most 128 x 64 reads, so the lane sums fit 32 bits: one DPP scan, and the tile border
      // (at most one inside 64 consecutive slots) splits the total by a v_readlane
      const int p = wave_scan_add((int)a);
      const int total = rdlane(p, 63);
      const int t0 = (int)((idx0 + 1) >> kTileShift);        // tile of lane 1 (lane 0 never holds a count)
      const i64 border = ((i64)(t0 + 1) << kTileShift) - idx0;   // first lane of the next tile
      const int s0 = border < 64 ? rdlane(p, (int)border - 1) : total;
      const int s1 = total - s0;
      if (lane == 0) { if (s0 != 0) atomicAdd(&part[t0], (u64)(unsigned)s0); if (s1 != 0) atomicAdd(&part[t0 + 1], (u64)(unsigned)s1); }
    } else {
      const int tile = (int)(idx >> kTileShift);
      const int t0 = rdlane(tile, 1);                        // lane 0 never holds a count
      const i64 s0 = wave_sum(tile == t0 ? (i64)a : 0), s1 = wave_sum(tile != t0 ? (i64)a : 0);
      if (lane == 0) { if (s0 != 0) atomicAdd(&part[t0], (u64)s0); if (s1 != 0) atomicAdd(&part[t0 + 1], (u64)s1); }
    }
  }

  __device__ __forceinline__ void flush_acc(const Seg &sg, int lane)
  {
    flush_at(sg, lane, base, acc);
    acc = 0;
    if (SUMKEY) {
      const i64 idx = (i64)base - 1 + lane + sg.cls;
      const int tile = (int)(idx >> kTileShift);
      const int t0 = rdlane(tile, 1);
      if (acc2 != 0) atomicAdd(&hist2[idx], (u64)acc2);
      const i64 q0 = wave_sum(tile == t0 ? acc2 : 0), q1 = wave_sum(tile != t0 ? acc2 : 0);
      if (lane == 0) { if (q0 != 0) atomicAdd(&part2[t0], (u64)q0); if (q1 != 0) atomicAdd(&part2[t0 + 1], (u64)q1); }
      acc2 = 0;
    }
  }

  __device__ __forceinline__ void flush(const Seg &sg, int lane)
  {
    deposit(lane);
    if (__ballot(acc != 0 || (SUMKEY && acc2 != 0))) flush_acc(sg, lane);
  }

  // position slot 0 at rank p (the caller has flushed)
  __device__ __forceinline__ void place(const Seg &sg, int p, int lane)
  {
    base = p; j = 0; pend = 0; acc = 0; acc2 = 0; pend2 = 0;
    W = load_window(sg, base, lane);
    Wn = load_window(sg, base + kSlots, lane);
    prevW = rdlane(W, 0); curW = rdlane(W, 1);
  }

  // the windows W / Wn already hold base p and p + kSlots: stand in slot j of it
  __device__ __forceinline__ void place_loaded(int p, int jj)
  {
    base = p; j = jj; pend = 0; acc = 0; acc2 = 0; pend2 = 0;
    prevW = rdlane(W, jj); curW = rdlane(W, jj + 1);
  }

  // wave-cooperative 64-ary search: sg.start + #{v in arr[sg.start..sg.end) : before(v, key)}
  __device__ __forceinline__ int rank_of(const Seg &sg, int key, int lane) const
  {
    int lo = sg.start, hi = sg.end;
    while (hi - lo > 64) {
      int step = (hi - lo + 63) >> 6;
      i64 idx = (i64)lo + (i64)lane * step;
      int v = idx < hi ? arr[idx] : kHi;
      int nless = __popcll(__ballot(idx < hi && before(v, key)));
      if (nless == 0) { hi = lo; break; }
      int nlo = lo + (nless - 1) * step + 1;
      i64 nhi = (i64)lo + (i64)nless * step;
      lo = nlo; if (nhi < hi) hi = (int)nhi;
    }
    int idx = lo + lane;
    int v = idx < hi ? arr[idx] : kHi;
    return lo + __popcll(__ballot(idx < hi && before(v, key)));
  }

  __device__ __forceinline__ void seek(const Seg &sg, int key, u64 m, int lane)
  {
    int k = wave_min(((m >> lane) & 1) ? key : kHi);
    place(sg, rank_of(sg, k, lane), lane);
  }

  // returns true when the register window was exchanged
  __device__ __forceinline__ bool fwd(const Seg &sg, int lane)
  {
    deposit(lane);
    bool sw = false;
    if (++j == kSlots) {
      flush(sg, lane);
      base += kSlots; j = 0; W = Wn; Wn = load_window(sg, base + kSlots, lane); sw = true;
    }
    prevW = curW; curW = rdlane(W, j + 1);
    return sw;
  }

  __device__ __forceinline__ void back(const Seg &sg, int lane)
  {
    deposit(lane);
    if (--j < 0) {
      flush(sg, lane);
      base -= kSlots; j = kSlots - 1; Wn = W; W = load_window(sg, base, lane);
    }
    curW = prevW; prevW = rdlane(W, j);
  }

  // Lanes on their own (keys scattered over many windows, reads of a class the wave is not following):
  // every lane of `m` finds its slot with a binary search in segment `sg` and adds itself to the
  // histogram; the tile sums get ONE atomic per distinct tile of the wave (per-lane atomics on the few
  // hundred tile counters would serialise on each other).  Called in wave-uniform control flow; sg may
  // differ per lane.
  __device__ __forceinline__ void lanes_add(const Seg &sg, int key, i64 w, u64 m, int lane) const
  {
    const bool mine = (m >> lane) & 1;
    int tile = -1;
    if (mine) {
      int lo = sg.start, hi = sg.end;
      while (lo < hi) { int mid = (int)(((i64)lo + hi) >> 1); if (before(arr[mid], key)) lo = mid + 1; else hi = mid; }
      const i64 idx = (i64)lo + sg.cls;
      atomicAdd(&hist[idx], (hist_t)w);
      if (SUMKEY) atomicAdd(&hist2[idx], (u64)(w * key));
      tile = (int)(idx >> kTileShift);
    }
    u64 rem = part ? m : 0;                                  // part == nullptr: the finalize step rebuilds the tile sums
    while (rem) {
      const int t0 = rdlane(tile, __ffsll((unsigned long long)rem) - 1);
      const u64 g = __ballot(tile == t0) & rem;
      rem &= ~g;
      const bool in = (g >> lane) & 1;
      const i64 s = wave_sum(in ? w : 0);
      if (lane == 0 && s != 0) atomicAdd(&part[t0], (u64)s);
      if (SUMKEY) { const i64 q = wave_sum(in ? w * key : 0); if (lane == 0 && q != 0) atomicAdd(&part2[t0], (u64)q); }
    }
  }

  // Add the lanes of m (keys `key`, weights `w`) to the histogram; returns the lanes it left
  // for lanes_add() because their keys are spread over too many windows (then `valid` is dropped).
  __device__ __forceinline__ u64 walk(const Seg &sg, int key, int w, u64 m, int lane, bool &valid)
  {
    if (!valid) { seek(sg, key, m, lane); valid = true; }
    // backward: some key at or below the boundary under the current slot
    int nback = 0;
    while (__ballot(below(key, prevW)) & m) {
      if (++nback > 6) { flush(sg, lane); seek(sg, key, m, lane); break; }
      back(sg, lane);
    }
    // forward: lanes at or below curW belong to slots <= j; the rest is still ahead
    u64 done = 0; int adv = 0;
    for (;;) {
      u64 le = __ballot(below(key, curW)) & m;
      u64 fresh = le & ~done;
      if (fresh) {
        if (WEIGHTED) pend += (acc_t)wave_sum(((fresh >> lane) & 1) ? (i64)w : 0);
        else pend += (acc_t)__popcll(fresh);
        if (SUMKEY) pend2 += wave_sum(((fresh >> lane) & 1) ? (i64)w * key : 0);
      }
      done = le;
      if (le == m) return 0;
      if (fwd(sg, lane) && ++adv > 2) { flush(sg, lane); valid = false; return m & ~done; }
    }
  }
};

struct __attribute__((packed, aligned(4))) Tri { int c, s, e; };

// a class the call counts: known, and -- for a group member -- one of its own (CountArgs::owned)
__device__ __forceinline__ bool class_counts(const CountArgs &a, int c) { return (unsigned)c < (unsigned)a.nClasses && (!a.owned || a.owned[c]); }

// sets the reads of the lanes in `m` aside for the pair kernels (see CountArgs::side); any control flow
template <class ARGS>
__device__ __forceinline__ void side_append(const ARGS &a, int c, int s, int e, int w, bool mine)
{
  if (a.side && mine) {
    const unsigned i = atomicAdd(a.sideCount, 1u);
    if (i < (unsigned)a.sideCap) a.side[i] = make_int4(c, s, e, w);
  }
}

// the read stream is touched exactly once: non-temporal loads keep it from displacing the boundary
// arrays and histograms in L2 / Infinity Cache (+10 % on the bare load pattern, scripts/membench.hip)
__device__ __forceinline__ Tri load_tri(const char *p)
{
  const int *q = (const int *)p;
  Tri t;
  t.c = __builtin_nontemporal_load(q); t.s = __builtin_nontemporal_load(q + 1); t.e = __builtin_nontemporal_load(q + 2);
  return t;
}

// per-wave running state of the streaming kernel (all wave-uniform except the windows inside A, B)
template <bool WEIGHTED, bool H32 = false>
struct WaveState {
  Win<WEIGHTED, false, false, H32> A;           // ends array, keyed by read start
  Win<WEIGHTED, true, false, H32> B;            // starts array, keyed by read end
  bool validA, validB;
  Seg sg;
  int nNoClass, nDegen;
  i64 firstDegen, firstUnsorted;
  int pc, ps;                                   // order check: class/start of the previous read
};

// General path: one chunk of 64 reads (lane-resident in t, w); `active` masks a partial chunk.
// Lanes whose read is of the wave's current class go through the two register windows; any other
// lane (class change inside a chunk, interleaved classes, keys scattered over many windows) adds
// itself with two binary searches.  Reads of an unknown class or with start > end are only counted
// into gtx_count_info.
template <bool WEIGHTED, bool H32>
__device__ __forceinline__ void walk_chunk(WaveState<WEIGHTED, H32> &st, const CountArgs &a, const Tri &t, int w, u64 active, i64 firstIndex, int lane)
{
  if (a.checkSorted) {
    int cc = t.c >> a.sortClassShift;
    int upc = __shfl_up(cc, 1), ups = __shfl_up(t.s, 1);
    if (lane == 0) { upc = st.pc; ups = st.ps; }
    u64 bad = __ballot(cc < upc || (cc == upc && t.s < ups)) & active;
    if (bad) { i64 at = firstIndex + (__ffsll((unsigned long long)bad) - 1); if (at < st.firstUnsorted) st.firstUnsorted = at; }
    int last = 63 - __clzll(active);
    st.pc = rdlane(cc, last); st.ps = rdlane(t.s, last);
  }

  const u64 degen = __ballot(t.s > t.e + a.zeroLenOk) & active;
  const u64 noclass = __ballot(!class_counts(a, t.c)) & active;
  if (degen | noclass) {
    st.nNoClass += __popcll(noclass);
    u64 dg = degen & ~noclass;
    if (dg) {
      st.nDegen += __popcll(dg); i64 at = firstIndex + (__ffsll((unsigned long long)dg) - 1); if (at < st.firstDegen) st.firstDegen = at;
      side_append(a, t.c, t.s, t.e, w, (dg >> lane) & 1);
    }
  }
  // The chunk class by class: the wave stays with its class while the chunk still has reads of it, then follows the class of
  // the first read left -- sorted reads: the one chunk of a span in which the chromosome changes takes two turns, each through
  // the register windows (the windows of the new class placed through the cell table, as at the start of a span).  Reads of
  // further classes (interleaved input) add themselves with two binary searches per lane, as do keys scattered over many windows.
  u64 todo = active & ~degen & ~noclass;                       // valid reads of known classes
#pragma unroll 1
  for (int turn = 0; todo && turn < 2; ++turn) {
    int c0 = rdlane(t.c, __ffsll((unsigned long long)todo) - 1);
    if (st.sg.cls >= 0 && (__ballot(t.c == st.sg.cls) & todo)) c0 = st.sg.cls;
    if (c0 != st.sg.cls) {
      if (st.validA) st.A.flush(st.sg, lane);
      if (st.validB) st.B.flush(st.sg, lane);
      st.validA = st.validB = false;
      const int4 pc = a.place.cls[c0];                         // {segment start, end, first cell, cells}
      st.sg.start = rfl(pc.x); st.sg.end = rfl(pc.y); st.sg.cls = c0;
      if (st.sg.start != st.sg.end) {
        const u64 m = __ballot(t.c == c0) & todo;
        const int kA = wave_min(((m >> lane) & 1) ? t.s : kHi), kB = wave_min(((m >> lane) & 1) ? t.e : kHi);
        int cell = (kA > 0 ? kA : 0) >> a.place.shift; cell = cell < rfl(pc.w) - 1 ? cell : rfl(pc.w) - 1;
        const int *rk = a.place.rank + 2 * ((i64)rfl(pc.z) + cell);
        const int pA = rfl(rk[0]), pB = rfl(rk[1]);
        st.A.W = st.A.load_window(st.sg, pA, lane); st.A.Wn = st.A.load_window(st.sg, pA + kSlots, lane);
        st.B.W = st.B.load_window(st.sg, pB, lane); st.B.Wn = st.B.load_window(st.sg, pB + kSlots, lane);
        const u64 mA = __ballot(st.A.before(st.A.W, kA)), mB = __ballot(st.B.before(st.B.W, kB));
        if ((mA & 1) && (i64)mA >= 0) { st.A.place_loaded(pA, __popcll(mA) - 1); st.validA = true; }
        if ((mB & 1) && (i64)mB >= 0) { st.B.place_loaded(pB, __popcll(mB) - 1); st.validB = true; }
      }
    }
    const u64 mine = __ballot(t.c == st.sg.cls) & todo;
    todo &= ~mine;
    if (mine && st.sg.start != st.sg.end) {
      u64 ra = st.A.walk(st.sg, t.s, w, mine, lane, st.validA);
      u64 rb = st.B.walk(st.sg, t.e, w, mine, lane, st.validB);
      if (ra) st.A.lanes_add(st.sg, t.s, w, ra, lane);
      if (rb) st.B.lanes_add(st.sg, t.e, w, rb, lane);
    }
  }
  if (todo) {
    Seg so; so.start = 0; so.end = 0; so.cls = 0;
    if ((todo >> lane) & 1) { so.start = a.segStart[t.c]; so.end = a.segStart[t.c + 1]; so.cls = t.c; }
    const u64 has = __ballot(so.start != so.end) & todo;           // classes without reference regions: nothing to add
    if (has) { st.A.lanes_add(so, t.s, w, has, lane); st.B.lanes_add(so, t.e, w, has, lane); }
  }
}

// Fast path for R x 64 reads of the current class, all valid, unweighted, window placed and no key
// behind the current slot (the caller has checked all that).  k[r] are the keys.
// The boundary-crossing loop inside one register window, hand-scheduled (gfx950 ISA): hipcc turns
// the C++ form of this multi-exit loop into a state machine of ~50 instructions per boundary; this
// is 24.  The loop is entered knowing that some key lies above the boundary W.  Per boundary: 4 compares
// against the wave-uniform W + 4 scalar popcounts give c = the number of keys at or below it; the slot
// below W is then complete and receives q + c (q = what was pending for it minus the keys counted by
// earlier boundaries of this step) in the lane that owns it (a one-lane exec mask around a v_add), q = -c;
// the next boundary is fetched with v_readlane and one compare of the per-lane maximum tells whether
// any key is still above it.  Leaves with status 0 when no key is above W (the open slot has q + 64 R
// pending), 1 when the window is exhausted (j == 63).  The wave issues one instruction per turn and the
// scalar pipe is the busiest one (PMC), so the instruction count of this loop is what the kernel's
// distance to the load-only ceiling is made of.
// CMP is "v_cmp_ge_i32" (key <= W) for the ends array, "v_cmp_gt_i32" (key < W) for the starts array.
#define GTX_CROSS_LOOP4(CMP)                                                                        \
  asm volatile(                                                                                     \
      "1:\n\t"                                                                                      \
      CMP " vcc, %[cw], %[k0]\n\t"                                                                  \
      "s_bcnt1_i32_b64 %[c], vcc\n\t"                                                               \
      CMP " vcc, %[cw], %[k1]\n\t"                                                                  \
      "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                               \
      "s_add_i32 %[c], %[c], %[t]\n\t"                                                              \
      CMP " vcc, %[cw], %[k2]\n\t"                                                                  \
      "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                               \
      "s_add_i32 %[c], %[c], %[t]\n\t"                                                              \
      CMP " vcc, %[cw], %[k3]\n\t"                                                                  \
      "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                               \
      "s_add_i32 %[j], %[j], 1\n\t"                /* lane j+1 owns the slot that is now complete */ \
      "s_add_i32 %[c], %[c], %[t]\n\t"                                                              \
      "s_add_i32 %[t1], %[j], 1\n\t"               /* lane of the next upper boundary             */ \
      "s_lshl_b64 exec, 1, %[j]\n\t"                                                                \
      "s_add_i32 %[t], %[q], %[c]\n\t"                                                              \
      "s_sub_i32 %[q], 0, %[c]\n\t"                                                                 \
      "v_add_u32 %[acc], %[t], %[acc]\n\t"                                                          \
      "s_mov_b64 exec, -1\n\t"                                                                      \
      "s_cmpk_eq_i32 %[j], 63\n\t"                                                                  \
      "s_cbranch_scc1 2f\n\t"                                                                       \
      "v_readlane_b32 %[cw], %[w], %[t1]\n\t"                                                       \
      "s_nop 0\n\t"                                                                                 \
      CMP " vcc, %[cw], %[km]\n\t"                                                                  \
      "s_cmp_eq_u64 vcc, -1\n\t"                                                                    \
      "s_cbranch_scc0 1b\n\t"                                                                       \
      "s_mov_b32 %[st], 0\n\t"                                                                      \
      "2:\n\t"                                                                                      \
      : [c] "=&s"(c), [t] "=&s"(t), [t1] "=&s"(t1), [st] "+s"(status), [cw] "+s"(curW), [j] "+s"(j), \
        [q] "+s"(q), [acc] "+v"(X.acc)                                                              \
      : [k0] "v"(k[0]), [k1] "v"(k[1]), [k2] "v"(k[2]), [k3] "v"(k[3]), [km] "v"(kmax), [w] "v"(X.W) \
      : "vcc", "scc")

// Fast path for R x 64 reads of the current class, all valid, unweighted, window placed and no key
// behind the current slot (the caller has checked all that).  k[r] are the keys, kmax their per-lane maximum.
template <int R, class WIN>
__device__ __forceinline__ void walk_fast(WIN &X, const Seg &sg, const int (&k)[R], int kmax, int lane, bool &valid)
{
  if (__ballot(WIN::below(kmax, X.curW)) == ~0ull) { X.pend += 64u * R; return; }   // no boundary crossed
  // (readfirstlane: these are wave-uniform, and the asm below needs them in SGPRs)
  int j = rfl(X.j), curW = rfl(X.curW);
  unsigned pend = (unsigned)rfl((int)X.pend), cprev = 0;
  int adv = 0;
  for (;;) {
    int status;
    if constexpr (R == 4 && sizeof(X.acc) == 4) {
      unsigned c, t, t1, q = pend - cprev;
      status = 1;
      if (WIN::kStrict) GTX_CROSS_LOOP4("v_cmp_gt_i32"); else GTX_CROSS_LOOP4("v_cmp_ge_i32");
      // asm results count as divergent for the compiler; they are SGPRs: tell it so
      status = rfl(status); j = rfl(j); curW = rfl(curW); q = (unsigned)rfl((int)q);
      if (status == 0) { pend = q + 64u * R; }
      else { pend = 0; cprev = 0u - q; }                     // window exhausted right after a deposit: q = -c
    } else {
      status = 1;
      for (;;) {                                             // inside one register window
        unsigned c = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) c += (unsigned)__popcll(__ballot(WIN::below(k[r], curW)));
        pend += c - cprev; cprev = c;
        if (c == 64u * R) { status = 0; break; }
        ++j;
        X.acc += (lane == j) ? pend : 0u; pend = 0;         // the slot is complete: hand its count to the lane that owns it
        if (j == kSlots) break;
        curW = rdlane(X.W, j + 1);
      }
    }
    if (status == 0) { X.j = j; X.prevW = rdlane(X.W, j); X.curW = curW; X.pend = pend; return; }
    // window exhausted (once per 63 boundaries): publish it and slide
    if (__ballot(X.acc != 0)) X.flush_acc(sg, lane);
    X.base += kSlots; j = 0;
    X.W = X.Wn; X.Wn = X.load_window(sg, X.base + kSlots, lane);
    curW = rdlane(X.W, 1);
    if (++adv > 2) {
      // keys spread over many windows: every key above the boundary just passed adds itself
      const int passed = rdlane(X.W, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) { const u64 ahead = __ballot(!WIN::below(k[r], passed)); if (ahead) X.lanes_add(sg, k[r], 1, ahead, lane); }
      X.j = 0; X.prevW = passed; X.curW = curW; X.pend = 0;
      valid = false;                                         // acc is 0 and nothing is pending: nothing to flush
      return;
    }
    if constexpr (R == 4 && sizeof(X.acc) == 4) {
      // the asm loop is entered only with some key above curW
      if (__ballot(WIN::below(kmax, curW)) == ~0ull) { X.j = 0; X.prevW = rdlane(X.W, 0); X.curW = curW; X.pend = 64u * R - cprev; return; }
    }
  }
}

__device__ __forceinline__ int max_of4(const int (&k)[4]) { int m = k[0]; m = k[1] > m ? k[1] : m; m = k[2] > m ? k[2] : m; return k[3] > m ? k[3] : m; }

// All boundaries of the window at once (see cov_step4_run; chosen by the host when the references are dense): for ordered keys, lane L counts the keys at or below its
// boundary by binary search in a wave-private LDS copy of the step's 256 keys; slots receive the lane differences.
// Returns false (nothing touched) when the keys are not in order.
template <class WIN>
__device__ __forceinline__ bool walk_flip4(WIN &X, const Seg &sg, const int (&k)[4], int lane, bool &valid, int *ldsK)
{
  if (__ballot(WIN::below(max_of4(k), X.curW)) == ~0ull) { X.pend += 256u; return true; }   // no boundary crossed
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int prev = lane_prev(k[r]);
    if (r > 0) { const int last = rdlane(k[r - 1], 63); prev = lane == 0 ? last : prev; }
    bad |= k[r] < prev;
  }
  if (__ballot(bad)) return false;
#pragma unroll
  for (int r = 0; r < 4; ++r) ldsK[64 * r + lane] = k[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (X.pend != 0) { X.acc += (lane == X.j + 1) ? X.pend : 0u; X.pend = 0; }
  int adv = 0;
  for (;;) {
    const int bnd = X.W;
    int cnt = 0;
#pragma unroll
    for (int half = 128; half >= 1; half >>= 1) cnt += WIN::below(ldsK[cnt + half - 1], bnd) ? half : 0;
    cnt += WIN::below(ldsK[cnt], bnd) ? 1 : 0;
    X.acc += (unsigned)(cnt - lane_prev(cnt));
    const int top = rdlane(cnt, 63);
    if (top == 256) {
      const int first = __ffsll((unsigned long long)__ballot(cnt == 256)) - 1;
      X.j = first - 1; X.prevW = rdlane(X.W, first - 1); X.curW = rdlane(X.W, first);
      return true;
    }
    if (__ballot(X.acc != 0)) X.flush_acc(sg, lane);
    X.base += kSlots; X.j = 0;
    X.W = X.Wn; X.Wn = X.load_window(sg, X.base + kSlots, lane);
    X.prevW = rdlane(X.W, 0); X.curW = rdlane(X.W, 1);
    if (++adv > 2) {
      valid = false;
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {
        const int kr = r == 0 ? k[0] : r == 1 ? k[1] : r == 2 ? k[2] : k[3];
        const u64 m = __ballot(64 * r + lane >= top);
        if (m) X.lanes_add(sg, kr, 1, m, lane);
      }
      return true;
    }
  }
}

// Weighted reads (label values): the all-boundaries-at-once step with the weights' exclusive prefix sums next to the keys
// (as cov_step4_run does for the key sums).  The caller has checked that the keys are non-decreasing over the 256 reads and
// that |weight| < 2^22 (so the prefix sums of a step fit 32 bits), and has stored the prefix sums in ldsP[0..256].
__device__ __forceinline__ bool keys_ordered4(const int (&k)[4], int lane)
{
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int prev = lane_prev(k[r]);
    if (r > 0) { const int last = rdlane(k[r - 1], 63); prev = lane == 0 ? last : prev; }
    bad |= k[r] < prev;
  }
  return __ballot(bad) == 0;
}

template <class WIN>
__device__ __forceinline__ void walk_flipw4(WIN &X, const Seg &sg, const int (&k)[4], int total, int lane, bool &valid,
                                            int *ldsK, const int *ldsP)
{
  if (__ballot(WIN::below(max_of4(k), X.curW)) == ~0ull) { X.pend += total; return; }   // no boundary crossed
#pragma unroll
  for (int r = 0; r < 4; ++r) ldsK[64 * r + lane] = k[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  X.deposit(lane);                                               // what is pending belongs to the slot we are about to leave
  int adv = 0;
  for (;;) {
    const int bnd = X.W;
    int cnt = 0;
#pragma unroll
    for (int half = 128; half >= 1; half >>= 1) cnt += WIN::below(ldsK[cnt + half - 1], bnd) ? half : 0;
    cnt += WIN::below(ldsK[cnt], bnd) ? 1 : 0;                   // 0..256 keys at or below this lane's boundary
    const int sum = ldsP[cnt];                                   // their weight
    X.acc += sum - lane_prev(sum);                               // lane 0 differs from itself: 0
    const int top = rdlane(cnt, 63);
    if (top == 256) {
      const int first = __ffsll((unsigned long long)__ballot(cnt == 256)) - 1;
      X.j = first - 1; X.prevW = rdlane(X.W, first - 1); X.curW = rdlane(X.W, first);
      return;
    }
    // keys beyond the window: publish it and slide (lane 0 of the new window holds the old lane 63, so the differences keep working)
    if (__ballot(X.acc != 0)) X.flush_acc(sg, lane);
    X.base += kSlots; X.j = 0;
    X.W = X.Wn; X.Wn = X.load_window(sg, X.base + kSlots, lane);
    X.prevW = rdlane(X.W, 0); X.curW = rdlane(X.W, 1);
    if (++adv > 2) {
      valid = false;                                             // acc is empty and nothing is pending
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {                              // the keys not yet placed add themselves (key and weight back from LDS)
        const int e = 64 * r + lane, kr = ldsK[e], wr = ldsP[e + 1] - ldsP[e];
        const u64 m = __ballot(e >= top);
        if (m) X.lanes_add(sg, kr, wr, m, lane);
      }
      return;
    }
  }
}

template <int R>
__device__ __forceinline__ int min_of(const int (&k)[R]) { int m = k[0];
#pragma unroll
  for (int r = 1; r < R; ++r) m = k[r] < m ? k[r] : m;
  return m; }
template <int R>
__device__ __forceinline__ int max_of(const int (&k)[R]) { int m = k[0];
#pragma unroll
  for (int r = 1; r < R; ++r) m = k[r] > m ? k[r] : m;
  return m; }

// Start of a wave's span: the ranks of its first keys in both boundary arrays, found together.  What a wave pays before
// it streams is a chain of dependent memory round trips (~2 us each under load) and the cache lines its probes touch:
// one 64-ary search after the other (Win::seek: three levels of 64 scattered lines each for 40 k boundaries per class)
// plus the two window placements made that chain nine round trips and ~400 lines long -- ~30 us of the 215 us a
// 100 M-read launch takes, three waves per slot.  Here both searches go in lockstep through two CONTIGUOUS hops: the
// class's piece of the every-256th-boundary arrays (topE / topS, <= 256 entries up to 65 k boundaries per class; larger
// classes narrow it first with 256 strided probes per level), then the <= 255 boundaries between two samples.
// Reads, samples, boundaries, windows = four round trips, ~70 lines.
//   rank = sg.start + #{v in arr[sg.start..sg.end) : before(v, key)}   (as Win::rank_of)
template <class WA, class WB>
__device__ __forceinline__ void rank_pair(const Seg &sg, const WA &A, const int *__restrict__ topA, int keyA, const WB &B,
                                          const int *__restrict__ topB, int keyB, int lane, int &pA, int &pB)
{
  // samples of the class: indices t0 .. t1-1 of the top arrays (boundary t << 8 lies in [sg.start, sg.end))
  const int t0 = (sg.start + 255) >> 8, t1 = ((sg.end - 1) >> 8) + 1;
  int loA = t0, hiA = t1 > t0 ? t1 : t0, loB = loA, hiB = hiA;
  while (hiA - loA > 256 || hiB - loB > 256) {
    const int stA = (hiA - loA + 255) >> 8, stB = (hiB - loB + 255) >> 8;
    int vA[4], vB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const i64 ia = (i64)loA + (i64)(64 * i + lane) * stA, ib = (i64)loB + (i64)(64 * i + lane) * stB;
      vA[i] = ia < hiA ? topA[ia] : kHi;
      vB[i] = ib < hiB ? topB[ib] : kHi;
    }
    int nA = 0, nB = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const i64 ia = (i64)loA + (i64)(64 * i + lane) * stA, ib = (i64)loB + (i64)(64 * i + lane) * stB;
      nA += __popcll(__ballot(ia < hiA && WA::before(vA[i], keyA)));
      nB += __popcll(__ballot(ib < hiB && WB::before(vB[i], keyB)));
    }
    // the probes are in array order, so nX of them before the key means: the rank lies behind probe nX-1 and at or before probe nX
    if (nA == 0) hiA = loA; else { const i64 nh = (i64)loA + (i64)nA * stA; loA = loA + (nA - 1) * stA + 1; if (nh < hiA) hiA = (int)nh; }
    if (nB == 0) hiB = loB; else { const i64 nh = (i64)loB + (i64)nB * stB; loB = loB + (nB - 1) * stB + 1; if (nh < hiB) hiB = (int)nh; }
  }
  int vA[4], vB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ia = loA + 64 * i + lane, ib = loB + 64 * i + lane;
    vA[i] = ia < hiA ? topA[ia] : kHi;
    vB[i] = ib < hiB ? topB[ib] : kHi;
  }
  int sA = loA, sB = loB;                                    // first sample that is not before the key
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ia = loA + 64 * i + lane, ib = loB + 64 * i + lane;
    sA += __popcll(__ballot(ia < hiA && WA::before(vA[i], keyA)));
    sB += __popcll(__ballot(ib < hiB && WB::before(vB[i], keyB)));
  }
  // the boundaries behind the last sample before the key, up to the first sample that is not: < 256 of them
  const int eLoA = sA > t0 ? ((sA - 1) << 8) + 1 : sg.start, eHiA = sA < t1 ? (sA << 8) : sg.end;
  const int eLoB = sB > t0 ? ((sB - 1) << 8) + 1 : sg.start, eHiB = sB < t1 ? (sB << 8) : sg.end;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ia = eLoA + 64 * i + lane, ib = eLoB + 64 * i + lane;
    vA[i] = ia < eHiA ? A.arr[ia] : kHi;
    vB[i] = ib < eHiB ? B.arr[ib] : kHi;
  }
  pA = eLoA; pB = eLoB;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ia = eLoA + 64 * i + lane, ib = eLoB + 64 * i + lane;
    pA += __popcll(__ballot(ia < eHiA && WA::before(vA[i], keyA)));
    pB += __popcll(__ballot(ib < eHiB && WB::before(vB[i], keyB)));
  }
}

// One wave owns chunksPerWave*64 consecutive reads and takes them in steps of R x 64 (register r of
// lane l holds read 64 r + l of the step: R coalesced 768-byte requests).  The scalar work per step
// (loop control, class / validity test, two "did anything cross a boundary" tests) is amortised over
// R x 64 reads.  The kernel is two nested loops: a tight inner loop that only knows the fast path
// and leaves -- before touching any state -- as soon as a step needs anything else (another class,
// a degenerate read, a key behind a window, an unplaced window, the partial last step), and the
// outer loop that gives exactly that step to the general per-chunk code and re-enters.
// PF (the plain kernel): the NEXT step of the fast loop travels by LDS-DMA (global_load_lds_dwordx3, no register destination)
// into 4 KB of LDS of the wave's own while the current step is being worked on in registers.  Without it a wave has bytes in
// flight only between issuing a step's loads and their arrival -- about 60 % of the time; a wave's rate is bytes in flight /
// latency, and the launch is short of resident waves to cover for that (8 per SIMD is the hardware's limit, a second register
// set would cost three of them).
template <bool WEIGHTED, int R, bool FLIP, bool PF = false, bool H32 = false>
__device__ __forceinline__ void count_walk_body(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, const CountArgs &a)
{
  static_assert(!PF || (R == 4 && !WEIGHTED), "the LDS prefetch is built for steps of 4 x 64 unweighted reads");
  // per wave: one step of triples as global_load_lds_dwordx3 lays them down -- lane l's 12 bytes at 16 l (a 16-byte pitch, the
  // fourth dword untouched; probed on the chip, scripts/dma_probe.hip): 1 KB per 64-read chunk
  __shared__ __attribute__((aligned(16))) int ldsT[PF ? 4 : 1][PF ? 256 * R : 4];
  __shared__ int ldsK[FLIP ? 8 : 1][FLIP ? 256 : 1];           // per wave: the keys of a step (walk_flip4)
  __shared__ int ldsP[(FLIP && WEIGHTED) ? 8 : 1][(FLIP && WEIGHTED) ? 264 : 1];   // and the prefix sums of their weights (walk_flipw4)
  const int wid = rfl(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + rfl(threadIdx.x >> 6);
  // this wave's span from the launch's schedule (SpanSchedule): long spans first, the short ones of the tail last
  int sw0 = a.sched.wave0[0], sc0 = a.sched.chunk0[0], scp = a.sched.cpw[0];
#pragma unroll
  for (int i = 1; i < SpanSchedule::kMax; ++i)
    if (wave >= a.sched.wave0[i]) { sw0 = a.sched.wave0[i]; sc0 = a.sched.chunk0[i]; scp = a.sched.cpw[i]; }
  const i64 first = ((i64)sc0 + (wave - sw0) * scp) * 64;      // first read of this wave's span
  if (first >= n) return;
  i64 cnt = n - first; if (cnt > (i64)scp * 64) cnt = (i64)scp * 64;
  const int nMine = (int)cnt;                                  // reads in this span
#ifdef GTX_WAVE_TRACE
  const u64 trT0 = __builtin_amdgcn_s_memrealtime(); u64 trT1 = 0;
#endif
  const int nSteps = (nMine + 64 * R - 1) / (64 * R);
  const int nFull = nMine / (64 * R);                          // steps with all R x 64 reads present

  WaveState<WEIGHTED, H32> st;
  typedef typename Win<WEIGHTED, false, false, H32>::hist_t hist_t;
  st.A.arr = a.sortedE; st.A.hist = (hist_t *)a.histA; st.A.part = a.partA; st.B.arr = a.sortedS; st.B.hist = (hist_t *)a.histB; st.B.part = a.partB;
  st.A.acc = st.B.acc = 0; st.A.pend = st.B.pend = 0; st.A.j = st.B.j = 0; st.A.base = st.B.base = 0;
  st.A.W = st.A.Wn = st.B.W = st.B.Wn = kHi; st.A.prevW = st.B.prevW = kLo; st.A.curW = st.B.curW = kHi;
  st.validA = st.validB = false;
  st.sg.start = 0; st.sg.end = 0; st.sg.cls = -1;
  st.nNoClass = 0; st.nDegen = 0; st.firstDegen = INT64_MAX; st.firstUnsorted = INT64_MAX;
  st.pc = kLo; st.ps = kLo;
  if (a.checkSorted && first > 0) { Tri p = reads[first - 1]; st.pc = rfl(p.c) >> a.sortClassShift; st.ps = rfl(p.s); }

  const char *base = (const char *)(reads + first);            // wave-uniform
  const int *wbase = WEIGHTED ? weights + first : nullptr;
  const unsigned loff = (unsigned)lane * 12u;
  const bool fastOk = (!a.checkSorted || R == 4) && (!WEIGHTED || (R == 4 && FLIP));
  const int zl = a.zeroLenOk;

  int s = 0;
  Tri t[R];
  int tw[WEIGHTED ? R : 1];                                    // the weights of step s (weighted fast path)
  bool have = false;                                           // t (and tw) hold step s
  int dmaStep = -1;                                            // PF: the step whose triples are in, or on their way to, ldsT[wid]
  auto dma_step = [&](int step) {
    if constexpr (PF) {
      const char *p = base + (size_t)step * (768 * R) + loff;
#pragma unroll
      for (int r = 0; r < R; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 768 * r),
                                         (__attribute__((address_space(3))) void *)&ldsT[wid][256 * r], 12, 0, 2);   // aux 2 = nt
    }
  };
  if (fastOk && nFull > 0) {
    // the common start: the first step is all of one class with reference regions -- place both windows at once.
    // The scalar load of the span's first read goes out first: when a launch begins every wave asks for its 3 KB at once, and
    // what is asked for behind them waits for 24 MB.
    const int *fr = (const int *)(reads + first);
    const int fc0 = fr[0], fs0 = fr[1];
    __builtin_amdgcn_sched_barrier(0);                           // (left alone hipcc issues the scalar load behind the vector loads)
#pragma unroll
    for (int r = 0; r < R; ++r) { t[r] = load_tri(base + 768 * r + loff); if constexpr (WEIGHTED) tw[r] = wbase[64 * r + lane]; }
    have = true;
    const int fc = rfl(fc0), fs = rfl(fs0);
    if constexpr (PF) { if (nFull > 1) { dma_step(1); dmaStep = 1; } }
    // While those loads are on their way: the span's FIRST read by scalar loads (their path does not queue behind the CU's
    // streaming loads), its class record and the two ranks of the cell its start lies in (PlaceTable) -- lower bounds of the
    // ranks of every key that is not below the cell -- and both windows loaded there.  Three short scalar round trips and one
    // vector one, under the step's own load, instead of the four vector round trips of rank_pair behind it (12 us median
    // under load, a quarter of a wave's life).  Nothing here is trusted: the step's real keys decide below.
    int preA = 0, preB = 0, preCls = -1; Seg preSeg; preSeg.start = 0; preSeg.end = 0; preSeg.cls = -1;
    {
      if (class_counts(a, fc)) {
        const int4 pc = a.place.cls[fc];
        preSeg.start = rfl(pc.x); preSeg.end = rfl(pc.y); preSeg.cls = fc;
        if (preSeg.start != preSeg.end) {
          int cell = (fs > 0 ? fs : 0) >> a.place.shift; cell = cell < rfl(pc.w) - 1 ? cell : rfl(pc.w) - 1;
          const int *rk = a.place.rank + 2 * ((i64)rfl(pc.z) + cell);
          preA = rfl(rk[0]); preB = rfl(rk[1]); preCls = fc;
          st.A.W = st.A.load_window(preSeg, preA, lane); st.A.Wn = st.A.load_window(preSeg, preA + kSlots, lane);
          st.B.W = st.B.load_window(preSeg, preB, lane); st.B.Wn = st.B.load_window(preSeg, preB + kSlots, lane);
        }
      }
    }
    const int c0 = rdlane(t[0].c, 0);
    int odd = 0, dg = 0, ks[R], ke[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      odd |= t[r].c ^ c0; ks[r] = t[r].s; ke[r] = t[r].e;
      dg |= __builtin_elementwise_sub_sat(__builtin_elementwise_add_sat(t[r].e, zl), t[r].s);
    }
    if (class_counts(a, c0) && !__ballot((odd != 0) | (dg < 0))) {
      if (c0 == preCls) st.sg = preSeg;
      else { st.sg.start = rfl(a.segStart[c0]); st.sg.end = rfl(a.segStart[c0 + 1]); st.sg.cls = c0; }
      if (st.sg.start != st.sg.end) {
        const int kA = wave_min(min_of<R>(ks)), kB = wave_min(min_of<R>(ke));
        // the boundaries of a window that sort before the key are a prefix of its lanes: lane 0 among them = the rank is not
        // below the window, lane 63 not among them = it is inside; then the slot is their number - 1
        const u64 mA = __ballot(st.A.before(st.A.W, kA)), mB = __ballot(st.B.before(st.B.W, kB));
        if (c0 == preCls && (mA & 1) && (mB & 1) && (i64)mA >= 0 && (i64)mB >= 0) {
          st.A.place_loaded(preA, __popcll(mA) - 1); st.B.place_loaded(preB, __popcll(mB) - 1);
        } else {
          int pA, pB;
          rank_pair(st.sg, st.A, a.topE, kA, st.B, a.topS, kB, lane, pA, pB);
          st.A.place(st.sg, pA, lane); st.B.place(st.sg, pB, lane);
        }
        st.validA = st.validB = true;
      }
    }
  }
#ifdef GTX_WAVE_TRACE
  trT1 = __builtin_amdgcn_s_memrealtime();
#endif
  while (s < nSteps) {
    // ---------------- fast loop ----------------
    // No software prefetch: a second register set (two-step ping-pong) takes the kernel from 58 to 91 VGPRs
    // (8 -> 5 waves per SIMD) and measured 0.243 against 0.222 ms; the 8 resident waves per SIMD overlap each
    // other's load latency instead.
    if (fastOk && st.validA && st.validB) {
      // one step of the fast path on registers tt; false (nothing touched) when the step needs the general path
      auto fast_step = [&](const Tri (&tt)[R]) -> bool {
        int odd = 0, dg = 0, ks[R], ke[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          odd |= tt[r].c ^ st.sg.cls; ks[r] = tt[r].s; ke[r] = tt[r].e;
          dg |= __builtin_elementwise_sub_sat(__builtin_elementwise_add_sat(tt[r].e, zl), tt[r].s);   // saturating: the sign is exact for any int32
        }
        // dg < 0 in some lane <=> some read has start > end (+zl)
        const int kmin = min_of<R>(ks), emin = min_of<R>(ke);
        if (__ballot((odd != 0) | (dg < 0) | (kmin <= st.A.prevW) | (emin < st.B.prevW))) return false;
        if constexpr (R == 4) {
          if (a.checkSorted) {
            // GTX_CHECK_SORTED: a step in (class, start) order behind the previous read is consumed here; the first
            // violation is found -- and reported with its index -- by the general code
            const int cc = st.sg.cls >> a.sortClassShift, k0 = rdlane(ks[0], 0);
            if (cc < st.pc || (cc == st.pc && k0 < st.ps) || !keys_ordered4(ks, lane)) return false;
          }
        }
        if constexpr (WEIGHTED && R == 4 && FLIP) {
          // weighted step: ordered keys and small weights, else the general path takes it
          bool big = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) big |= (unsigned)(tw[r] + (1 << 22)) >= (1u << 23);
          if (__ballot(big) || !keys_ordered4(ks, lane) || !keys_ordered4(ke, lane)) return false;
          int run = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int p = wave_scan_add(tw[r]); ldsP[wid][64 * r + lane] = run + p - tw[r]; run += rdlane(p, 63); }
          if (lane == 0) ldsP[wid][256] = run;
          walk_flipw4(st.A, st.sg, ks, run, lane, st.validA, ldsK[wid], ldsP[wid]);
          walk_flipw4(st.B, st.sg, ke, run, lane, st.validB, ldsK[wid], ldsP[wid]);
        } else if constexpr (!WEIGHTED && R == 4 && FLIP) {
          if (!walk_flip4(st.A, st.sg, ks, lane, st.validA, ldsK[wid])) walk_fast<R>(st.A, st.sg, ks, max_of<R>(ks), lane, st.validA);
          if (!walk_flip4(st.B, st.sg, ke, lane, st.validB, ldsK[wid])) walk_fast<R>(st.B, st.sg, ke, max_of<R>(ke), lane, st.validB);
        } else if constexpr (!WEIGHTED) {
          walk_fast<R>(st.A, st.sg, ks, max_of<R>(ks), lane, st.validA);
          walk_fast<R>(st.B, st.sg, ke, max_of<R>(ke), lane, st.validB);
        }
        if constexpr (R == 4) {
          if (a.checkSorted) { st.pc = st.sg.cls >> a.sortClassShift; st.ps = rdlane(ks[3], 63); }   // the step is consumed: it is the previous read now
        }
        return true;
      };
      auto load_step = [&](Tri (&tt)[R], int step) {
        const char *p = base + (size_t)step * (768 * R) + loff;
#pragma unroll
        for (int r = 0; r < R; ++r) { tt[r] = load_tri(p + 768 * r); if constexpr (WEIGHTED) tw[r] = wbase[(size_t)step * (64 * R) + 64 * r + lane]; }
      };
      while (s < nFull) {
        if constexpr (PF) {
          if (!have) {
            if (dmaStep != s) dma_step(s);                            // (only behind a step that went through the general code twice over)
            // nothing orders a ds_read behind a pending LDS-DMA except the issuing wave's vmcnt (hipcc does not insert it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            typedef int v3i __attribute__((ext_vector_type(3)));     // 16 bytes in memory, 3 registers: ds_read_b96
            const v3i *L = (const v3i *)&ldsT[wid][4 * lane];
#pragma unroll
            for (int r = 0; r < R; ++r) { const v3i v = L[64 * r]; t[r].c = v.x; t[r].s = v.y; t[r].e = v.z; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the buffer is about to be overwritten
            if (s + 1 < nFull) { dma_step(s + 1); dmaStep = s + 1; }
          }
        } else {
          if (!have) load_step(t, s);
        }
        have = true;
        if (!fast_step(t)) break;
        ++s; have = false;
        if (!(st.validA && st.validB)) break;
      }
      if (s >= nSteps) break;
    }
    // ---------------- general path for step s ----------------
    const int at = s * 64 * R;
    int w[R];
    if (!have || WEIGHTED) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        t[r].c = -1; t[r].s = 0; t[r].e = 0; w[r] = 1;
        if (at + 64 * r + lane < nMine) { t[r] = load_tri(base + (size_t)(at + 64 * r) * 12 + loff); if (WEIGHTED) w[r] = wbase[at + 64 * r + lane]; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) w[r] = 1;
    }
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      const int left = nMine - (at + 64 * r);
      if (left <= 0) break;
      Tri tt = t[0]; int ww = w[0];
#pragma unroll
      for (int q = 1; q < R; ++q) if (r == q) { tt = t[q]; ww = w[q]; }
      const u64 active = left >= 64 ? ~0ull : ((1ull << left) - 1);
      walk_chunk<WEIGHTED, H32>(st, a, tt, ww, active, first + at + 64 * r, lane);
    }
    ++s; have = false;
  }
  if (st.validA) st.A.flush(st.sg, lane);
  if (st.validB) st.B.flush(st.sg, lane);
  if (lane == 0) {
    if (st.nNoClass) atomicAdd((u64 *)&a.info->n_no_class, (u64)st.nNoClass);
    if (st.nDegen) { atomicAdd((u64 *)&a.info->n_degenerate, (u64)st.nDegen); atomicMin((i64 *)&a.info->first_degenerate, st.firstDegen + a.indexBase); }
    if (st.firstUnsorted != INT64_MAX) atomicMin((i64 *)&a.info->first_unsorted, st.firstUnsorted + a.indexBase);
  }
#ifdef GTX_WAVE_TRACE
  if (a.trace && lane == 0) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    u64 *t = a.trace + wave * 4;
    t[0] = trT0; t[1] = trT1; t[2] = __builtin_amdgcn_s_memrealtime(); t[3] = ((u64)hw << 32) | xcc;
  }
#endif
}

// Residency is set by the scalar registers: 256-thread blocks are admitted per CU by floor(800 / (ceil(sgpr/16)*16 + 16)) --
// 8 blocks (8 waves per SIMD) up to 80 SGPRs, 7 up to 96, 6 above (MI355X_MICROARCH.md, "Residency").  Left alone the
// compiler takes 96-106 for this kernel; capped, it parks a dozen rarely used scalars in the lanes of one VGPR.
// 100 M x 1 M: 6 -> 8 waves per SIMD, 0.240 -> 0.222 ms.
template <bool WEIGHTED, int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void count_walk_kernel(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a)
{
  count_walk_body<WEIGHTED, R, false>(reads, weights, n, a);
}
// the same with 32-bit histogram slots (CountArgs::hist32: an unweighted call of fewer than 2^32 reads on resident reads)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void count_walk_kernel_h32(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a)
{
  count_walk_body<false, 4, false, false, true>(reads, weights, n, a);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void count_walk_kernel_flip_h32(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a)
{
  count_walk_body<false, 4, true, false, true>(reads, weights, n, a);
}
// experiment (GTX_PF=1): the next step prefetched through LDS
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void count_walk_kernel_pf(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a)
{
  count_walk_body<false, 4, false, true>(reads, weights, n, a);
}
// weighted reads: steps of 4 x 64 with the weights' prefix sums in LDS (walk_flipw4); the general code takes what does not qualify.
// (93 VGPRs = 5 waves per SIMD left alone; 80 = 6 without a spill when told to: 0.272 -> 0.268 ms on one box; 7 waves spill: 0.39)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(6, 6))) void count_walk_kernel_weighted(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a)
{
  count_walk_body<true, 4, true>(reads, weights, n, a);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void count_walk_kernel_flip(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a)
{
  count_walk_body<false, 4, true>(reads, weights, n, a);
}

// ---------------------------------------------------------------------------------------------
// Coverage (GenomicRegionSetOverlaps::CalcIndexCoverage, genomic_intervals.cpp:5269-5285):
//   cov[k] = sum over reads q overlapping k of w_q * (min(e_q,E_k) - max(s_q,S_k) + 1)
// With Ws(x) = sum w[s<=x], We(x) = sum w[e<=x], Fs(x) = sum w*s[s<=x], Fe(x) = sum w*e[e<=x] (valid reads,
// s <= e) the overlapping reads split by where their ends lie, and
//   cov[k] = Fe(E) - Fe(S-1) + E*(Ws(E) - We(E))  -  Fs(E) + Fs(S-1) - S*(Ws(S-1) - We(S-1))  +  Ws(E) - We(S-1)
// (all in wrap-around 64-bit arithmetic, like the reference's unsigned long).  All eight sums are "weight [x key] of the
// reads whose key lies at or below a threshold", the thresholds being E_k and S_k - 1: so the thresholds of a class are
// merged into ONE sorted array (sortedT, built by the host) and ONE streaming pass walks two windows over it -- keyed by
// the read starts and by the read ends -- each with a weight histogram and a weight x key histogram.  (Round 1 walked
// four windows, over the ends array and over the starts array, in two launches that each read the triples.)
// Zero-length reads and regions always contribute 0 and are left out.
// ---------------------------------------------------------------------------------------------
// Fast form of Win::walk for one full chunk of the coverage pass (unweighted, all 64 lanes of the wave's class,
// window placed).  When the keys are non-decreasing across the lanes -- sorted reads: always for the starts,
// for the ends whenever the read length does not shrink faster than the starts grow -- the lanes at or
// below a boundary are a PREFIX of the wave, so the key sum of a slot is a difference of two entries of the
// lane-wise prefix sum of the keys: one v_readlane per boundary instead of a 64-lane reduction.  Keys are
// taken relative to lane 0's (spread < 2^24, so 64 of them fit 32 bits).  Anything else goes to walk().
template <class WIN>
__device__ __forceinline__ u64 walk_cov_chunk(WIN &X, const Seg &sg, int key, int lane, bool &valid)
{
  const int kbase = rdlane(key, 0), ktop = rdlane(key, 63);
  const int prev = lane_prev(key);
  const bool ok = valid && (unsigned)(ktop - kbase) < (1u << 24) && __ballot(key < prev) == 0 && !WIN::below(kbase, X.prevW);
  if (!ok) return X.walk(sg, key, 1, ~0ull, lane, valid);
  const int rel = key - kbase;
  const int p = wave_scan_add(rel);                            // inclusive prefix sum of the relative keys
  const int q = p - rel;                                       // exclusive: sum over the lanes below
  const int total = rdlane(p, 63);
  int cprev = 0, sprev = 0, adv = 0;
  for (;;) {
    const int c = __popcll(__ballot(WIN::below(key, X.curW)));  // = number of leading lanes at or below the boundary
    const int srel = c == 64 ? total : rdlane(q, c);
    const int dc = c - cprev;
    X.pend += dc;
    X.pend2 += (i64)kbase * dc + (srel - sprev);
    cprev = c; sprev = srel;
    if (c == 64) return 0;
    if (X.fwd(sg, lane) && ++adv > 2) { X.flush(sg, lane); valid = false; return ~0ull << c; }   // the lanes not yet placed add themselves
  }
}

// The same for a whole step of 4 x 64 reads (register r of lane l = read 64 r + l): keys non-decreasing over all
// 256, prefix sums per register, and per boundary one count over the four registers + ONE v_readlane in the
// register the boundary falls into.  The per-step bookkeeping (class / validity tests, window checks, scans) is
// paid once per 256 reads instead of once per 64.  cov_step4_ok() says whether a window can take the step.
template <class WIN>
__device__ __forceinline__ bool cov_step4_ok(const WIN &X, const int (&k)[4], int lane, bool valid)
{
  if (!valid) return false;
  const int kbase = rdlane(k[0], 0), ktop = rdlane(k[3], 63);
  if ((unsigned)(ktop - kbase) >= (1u << 22) || WIN::below(kbase, X.prevW)) return false;
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int prev = lane_prev(k[r]);
    if (r > 0) { const int last = rdlane(k[r - 1], 63); prev = lane == 0 ? last : prev; }
    bad |= k[r] < prev;
  }
  return __ballot(bad) == 0;
}

// One step of 4 x 64 ordered keys against one window, all boundaries of the window AT ONCE: the 256 keys and the
// exclusive prefix sums of their (relative) values go to a wave-private piece of LDS; lane L, which holds boundary
// base-1+L, finds by a fixed-depth binary search how many keys lie at or below it (cnt) and reads the prefix sum at
// that position (sum); slot L-1 then receives cnt[L] - cnt[L-1] keys and sum[L] - sum[L-1] of key value (DPP lane
// differences).  The cost does not depend on how many boundaries the step crosses -- the per-boundary loop this
// replaces cost ~60 instructions per crossing.  A step that runs past the window publishes it, slides and repeats.
__device__ __forceinline__ i64 rd64(i64 v, int l) { return (i64)(((u64)(unsigned)rdlane((int)((u64)v >> 32), l) << 32) | (unsigned)rdlane((int)(unsigned)(u64)v, l)); }
__device__ __forceinline__ i64 prev64(i64 v) { return (i64)(((u64)(unsigned)lane_prev((int)((u64)v >> 32)) << 32) | (unsigned)lane_prev((int)(unsigned)(u64)v)); }

// The same step for weighted reads: what a slot receives is the weight and the weight x key of its keys, so the LDS holds the
// exclusive prefix sums of w (ldsW, stored once per step by the caller: both windows of a launch share it) and of
// w x (key - kbase) (ldsWK, per window), in 64 bits -- no bound on the label values.
// SHIFTED (as cov_step4_run): this window's keys are the keys the other window has just staged plus the wave-uniform `shift`, so ldsK and
// ldsWK are that window's and `run` its total (passed in); otherwise `run` goes out for such a second window.
template <bool SHIFTED, class WIN>
__device__ __forceinline__ void cov_step4_run_w(WIN &X, const Seg &sg, const int (&k)[4], const int (&w)[4], i64 totalW, int lane, bool &valid,
                                                int *ldsK, const i64 *ldsW, i64 *ldsWK, i64 &run, int shift = 0)
{
  const int kbase = rdlane(k[0], 0);
  if constexpr (!SHIFTED) {
    run = 0;                                                     // sum of w x relative key of the registers below
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const i64 v = (i64)w[r] * (k[r] - kbase), p = (i64)wave_scan_add64((u64)v);
      ldsK[64 * r + lane] = k[r];
      ldsWK[64 * r + lane] = run + p - v;                        // exclusive prefix over all reads before this one
      run += rd64(p, 63);
    }
  }
  if (WIN::below(rdlane(k[3], 63), X.curW)) {                    // the whole step stays in the current slot
    X.pend += totalW;
    X.pend2 += (i64)kbase * totalW + run;
    return;
  }
  if (lane == 0) ldsWK[256] = run;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  X.deposit(lane);                                               // what is pending belongs to the slot we are about to leave
  int adv = 0;
  for (;;) {
    const int bnd = SHIFTED ? __builtin_elementwise_sub_sat(X.W, shift) : X.W;
    int cnt = 0;
#pragma unroll
    for (int half = 128; half >= 1; half >>= 1) cnt += WIN::below(ldsK[cnt + half - 1], bnd) ? half : 0;
    cnt += WIN::below(ldsK[cnt], bnd) ? 1 : 0;                   // 0..256 keys at or below this lane's boundary
    const i64 sw = ldsW[cnt], swk = ldsWK[cnt];
    const i64 dw = sw - prev64(sw), dwk = swk - prev64(swk);     // lane 0 differs from itself: 0
    X.acc += dw;
    X.acc2 += (i64)kbase * dw + dwk;
    const int top = rdlane(cnt, 63);
    if (top == 256) {
      const int first = __ffsll((unsigned long long)__ballot(cnt == 256)) - 1;
      X.j = first - 1; X.prevW = rdlane(X.W, first - 1); X.curW = rdlane(X.W, first);
      return;
    }
    if (__ballot(X.acc != 0 || X.acc2 != 0)) X.flush_acc(sg, lane);
    X.base += kSlots; X.j = 0;
    X.W = X.Wn; X.Wn = X.load_window(sg, X.base + kSlots, lane);
    X.prevW = rdlane(X.W, 0); X.curW = rdlane(X.W, 1);
    if (++adv > 2) {
      valid = false;                                             // acc is empty and nothing is pending
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {                              // the keys not yet placed add themselves (key and weight back from LDS)
        const int e = 64 * r + lane, kr = ldsK[e] + (SHIFTED ? shift : 0), wr = (int)(ldsW[e + 1] - ldsW[e]);
        const u64 m = __ballot(e >= top);
        if (m) X.lanes_add(sg, kr, wr, m, lane);
      }
      return;
    }
  }
}

// SHIFTED: the keys of this window are the keys ANOTHER window has just staged plus a wave-uniform `shift` (read ends of equal-length
// reads behind their starts): their relative values, hence the prefix sums in ldsP, are the staged ones, and key <= W is staged key <=
// W - shift (saturating: a sentinel stays one) -- nothing is staged again.  Returns whether ldsK / ldsP hold this step's keys.
template <bool SHIFTED, class WIN>
__device__ __forceinline__ bool cov_step4_run(WIN &X, const Seg &sg, const int (&k)[4], int lane, bool &valid, int *ldsK, int *ldsP, int shift = 0)
{
  const int kbase = rdlane(k[0], 0);
  if (WIN::below(rdlane(k[3], 63), X.curW)) {                    // the whole step stays in the current slot
    X.pend += 256;
    if constexpr (SHIFTED) X.pend2 += (i64)kbase * 256 + rdlane(ldsP[256], 0);
    else {
      const int t = (k[0] - kbase) + (k[1] - kbase) + (k[2] - kbase) + (k[3] - kbase);
      X.pend2 += (i64)kbase * 256 + rdlane(wave_scan_add(t), 63);
    }
    return SHIFTED;
  }
  if constexpr (!SHIFTED) {
    int run = 0;                                                 // sum of the relative keys of the registers below
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rel = k[r] - kbase, p = wave_scan_add(rel);
      ldsK[64 * r + lane] = k[r];
      ldsP[64 * r + lane] = run + p - rel;                       // exclusive prefix over all keys before this one
      run += rdlane(p, 63);
    }
    if (lane == 0) ldsP[256] = run;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  X.deposit(lane);                                               // what is pending belongs to the slot we are about to leave
  int adv = 0;
  for (;;) {
    const int bnd = SHIFTED ? __builtin_elementwise_sub_sat(X.W, shift) : X.W;
    int cnt = 0;
#pragma unroll
    for (int half = 128; half >= 1; half >>= 1) cnt += WIN::below(ldsK[cnt + half - 1], bnd) ? half : 0;
    cnt += WIN::below(ldsK[cnt], bnd) ? 1 : 0;                   // 0..256 keys at or below this lane's boundary
    const int sum = ldsP[cnt];
    const int dc = cnt - lane_prev(cnt), ds = sum - lane_prev(sum);     // lane 0 differs from itself: 0
    X.acc += dc;
    X.acc2 += (i64)kbase * dc + ds;
    const int top = rdlane(cnt, 63);
    if (top == 256) {
      // every key is placed; the slot of the last key is the one below the first boundary that has them all
      const int first = __ffsll((unsigned long long)__ballot(cnt == 256)) - 1;
      X.j = first - 1; X.prevW = rdlane(X.W, first - 1); X.curW = rdlane(X.W, first);
      return true;
    }
    // keys beyond the window: publish it and slide (lane 0 of the new window holds the old lane 63, so its count
    // is `top` and the differences above keep working)
    if (__ballot(X.acc != 0 || X.acc2 != 0)) X.flush_acc(sg, lane);
    X.base += kSlots; X.j = 0;
    X.W = X.Wn; X.Wn = X.load_window(sg, X.base + kSlots, lane);
    X.prevW = rdlane(X.W, 0); X.curW = rdlane(X.W, 1);
    if (++adv > 2) {
      valid = false;                                             // acc is empty and nothing is pending
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {                              // the keys not yet placed add themselves
        const int kr = r == 0 ? k[0] : r == 1 ? k[1] : r == 2 ? k[2] : k[3];
        const u64 m = __ballot(64 * r + lane >= top);
        if (m) X.lanes_add(sg, kr, 1, m, lane);
      }
      return true;
    }
  }
}

// per-wave state of the coverage pass: two windows over the merged threshold array, keyed by the read starts (Ws) and by
// the read ends (We); a key x belongs to the slots at or below threshold T iff x <= T
template <bool WEIGHTED>
struct CovState {
  Win<WEIGHTED, false, true> Ws, We;
  bool vs, ve;
  Seg sg;
  int nNoClass, nDegen; i64 firstDegen;
};

// one chunk of 64 reads, any mix of classes / invalid reads (`active` masks a partial chunk)
template <bool WEIGHTED>
__device__ __forceinline__ void cov_chunk(CovState<WEIGHTED> &st, const CoverArgs &a, const Tri &t, int w, u64 active, i64 firstIndex, int lane)
{
  Seg &sg = st.sg;
  const u64 noclass = __ballot((unsigned)t.c >= (unsigned)a.nClasses) & active;
  const u64 degen = __ballot(t.s > t.e) & active & ~noclass;        // zero-length or inverted: contributes nothing
  if (degen | noclass) {
    st.nNoClass += __popcll(noclass);
    const u64 inv = __ballot(t.s > t.e + 1) & degen;                // only these are reported (the packer's business)
    if (inv) {
      st.nDegen += __popcll(inv); i64 p = firstIndex + (__ffsll((unsigned long long)inv) - 1); if (p < st.firstDegen) st.firstDegen = p;
      side_append(a, t.c, t.s, t.e, w, (inv >> lane) & 1);
    }
  }
  // class by class, as walk_chunk: the chunk in which the chromosome changes takes two turns through the register windows, the
  // new class's windows placed through the cell table
  u64 todo = active & ~degen & ~noclass;
#pragma unroll 1
  for (int turn = 0; todo && turn < 2; ++turn) {
    int c0 = rdlane(t.c, __ffsll((unsigned long long)todo) - 1);
    if (sg.cls >= 0 && (__ballot(t.c == sg.cls) & todo)) c0 = sg.cls;
    if (c0 != sg.cls) {
      if (st.vs) st.Ws.flush(sg, lane);
      if (st.ve) st.We.flush(sg, lane);
      st.vs = st.ve = false;
      const int4 pc = a.place.cls[c0];
      sg.start = rfl(pc.x); sg.end = rfl(pc.y); sg.cls = c0;
      if (sg.start != sg.end) {
        const u64 m = __ballot(t.c == c0) & todo;
        const int ks = wave_min(((m >> lane) & 1) ? t.s : kHi), ke = wave_min(((m >> lane) & 1) ? t.e : kHi);
        int cell = (ks > 0 ? ks : 0) >> a.place.shift; cell = cell < rfl(pc.w) - 1 ? cell : rfl(pc.w) - 1;
        const int p0 = rfl(a.place.rank[2 * ((i64)rfl(pc.z) + cell)]);
        st.Ws.W = st.We.W = st.Ws.load_window(sg, p0, lane); st.Ws.Wn = st.We.Wn = st.Ws.load_window(sg, p0 + kSlots, lane);
        const u64 ms = __ballot(st.Ws.before(st.Ws.W, ks)), me = __ballot(st.We.before(st.We.W, ke));
        if ((ms & 1) && (i64)ms >= 0) { st.Ws.place_loaded(p0, __popcll(ms) - 1); st.vs = true; }
        if ((me & 1) && (i64)me >= 0) { st.We.place_loaded(p0, __popcll(me) - 1); st.ve = true; }
      }
    }
    const u64 mine = __ballot(t.c == sg.cls) & todo;
    todo &= ~mine;
    if (mine && sg.start != sg.end) {
      u64 r0, r1;
      if (!WEIGHTED && mine == ~0ull) {
        r0 = walk_cov_chunk(st.Ws, sg, t.s, lane, st.vs);
        r1 = walk_cov_chunk(st.We, sg, t.e, lane, st.ve);
      } else {
        r0 = st.Ws.walk(sg, t.s, w, mine, lane, st.vs);
        r1 = st.We.walk(sg, t.e, w, mine, lane, st.ve);
      }
      if (r0) st.Ws.lanes_add(sg, t.s, w, r0, lane);
      if (r1) st.We.lanes_add(sg, t.e, w, r1, lane);
    }
  }
  if (todo) {
    Seg so; so.start = 0; so.end = 0; so.cls = 0;
    if ((todo >> lane) & 1) { so.start = a.segStartT[t.c]; so.end = a.segStartT[t.c + 1]; so.cls = t.c; }
    const u64 has = __ballot(so.start != so.end) & todo;
    if (has) { st.Ws.lanes_add(so, t.s, w, has, lane); st.We.lanes_add(so, t.e, w, has, lane); }
  }
}

template <bool WEIGHTED>
__device__ __forceinline__ void coverage_walk_body(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, const CoverArgs &a)
{
  const int lane = threadIdx.x & 63;
  const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + rfl(threadIdx.x >> 6);
  int sw0 = a.sched.wave0[0], sc0 = a.sched.chunk0[0], scp = a.sched.cpw[0];      // this wave's span (SpanSchedule)
#pragma unroll
  for (int i = 1; i < SpanSchedule::kMax; ++i)
    if (wave >= a.sched.wave0[i]) { sw0 = a.sched.wave0[i]; sc0 = a.sched.chunk0[i]; scp = a.sched.cpw[i]; }
  const i64 first = ((i64)sc0 + (wave - sw0) * scp) * 64;
  if (first >= n) return;
  i64 cnt = n - first; if (cnt > (i64)scp * 64) cnt = (i64)scp * 64;
  const int nMine = (int)cnt;

  __shared__ int ldsK[4][256], ldsP[WEIGHTED ? 1 : 4][WEIGHTED ? 1 : 264];   // per wave: the keys of a step and their prefix sums
  __shared__ i64 ldsW[WEIGHTED ? 4 : 1][WEIGHTED ? 264 : 1], ldsWK[WEIGHTED ? 4 : 1][WEIGHTED ? 264 : 1];   // weighted: prefix sums of w and of w x key
  const int wid = rfl(threadIdx.x >> 6);
  CovState<WEIGHTED> st;
  st.Ws.arr = st.We.arr = a.sortedT;
  st.Ws.hist = a.hist[0]; st.Ws.hist2 = a.hist[1]; st.Ws.part = a.part[0]; st.Ws.part2 = a.part[1];
  st.We.hist = a.hist[2]; st.We.hist2 = a.hist[3]; st.We.part = a.part[2]; st.We.part2 = a.part[3];
  st.Ws.acc = st.We.acc = 0; st.Ws.acc2 = st.We.acc2 = 0; st.Ws.pend = st.We.pend = 0; st.Ws.pend2 = st.We.pend2 = 0;
  st.Ws.j = st.We.j = 0; st.Ws.base = st.We.base = 0;
  st.Ws.W = st.Ws.Wn = st.We.W = st.We.Wn = kHi;
  st.Ws.prevW = st.We.prevW = kLo; st.Ws.curW = st.We.curW = kHi;
  st.vs = st.ve = false;
  st.sg.start = 0; st.sg.end = 0; st.sg.cls = -1;
  st.nNoClass = 0; st.nDegen = 0; st.firstDegen = INT64_MAX;

  // Steps of 4 x 64 reads wherever both windows can take them (one class, valid reads, ordered keys); otherwise ONE
  // chunk goes through the general code and the next step is tried right after it -- steps need no alignment.
  // The general code exists once (it is large; two copies of it would not share the instruction cache well).
  const char *base = (const char *)(reads + first) + (size_t)lane * 12;
  int at = 0;
  if ((!WEIGHTED || a.wfast) && nMine >= 256) {
    // the common start (as in count_walk_body): the first step is all of one class with reference regions -- both
    // windows placed by one paired search instead of two 64-ary searches in a row through the general code
    // (as count_walk_body: the span's first read by a scalar load ahead of the step's vector loads, its cell's rank, the windows
    // loaded there while the step is on its way; the step's real keys decide)
    const int *fr = (const int *)(reads + first);
    const int fc0 = fr[0], fs0 = fr[1];
    __builtin_amdgcn_sched_barrier(0);
    Tri t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = load_tri(base + 768 * r);
    const int fc = rfl(fc0), fs = rfl(fs0);
    int pre = 0, preCls = -1; Seg preSeg; preSeg.start = 0; preSeg.end = 0; preSeg.cls = -1;
    if ((unsigned)fc < (unsigned)a.nClasses) {
      const int4 pc = a.place.cls[fc];
      preSeg.start = rfl(pc.x); preSeg.end = rfl(pc.y); preSeg.cls = fc;
      if (preSeg.start != preSeg.end) {
        int cell = (fs > 0 ? fs : 0) >> a.place.shift; cell = cell < rfl(pc.w) - 1 ? cell : rfl(pc.w) - 1;
        pre = rfl(a.place.rank[2 * ((i64)rfl(pc.z) + cell)]); preCls = fc;
        st.Ws.W = st.We.W = st.Ws.load_window(preSeg, pre, lane); st.Ws.Wn = st.We.Wn = st.Ws.load_window(preSeg, pre + kSlots, lane);
      }
    }
    const int c0 = rdlane(t[0].c, 0);
    bool odd = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) odd |= t[r].c != c0 || t[r].s > t[r].e;
    if ((unsigned)c0 < (unsigned)a.nClasses && __ballot(odd) == 0) {
      if (c0 == preCls) st.sg = preSeg;
      else { st.sg.start = rfl(a.segStartT[c0]); st.sg.end = rfl(a.segStartT[c0 + 1]); st.sg.cls = c0; }
      if (st.sg.start != st.sg.end) {
        const int ks[4] = {t[0].s, t[1].s, t[2].s, t[3].s}, ke[4] = {t[0].e, t[1].e, t[2].e, t[3].e};
        const int kS = wave_min(min_of<4>(ks)), kE = wave_min(min_of<4>(ke));
        const u64 ms = __ballot(st.Ws.before(st.Ws.W, kS)), me = __ballot(st.We.before(st.We.W, kE));
        if (c0 == preCls && (ms & 1) && (me & 1) && (i64)ms >= 0 && (i64)me >= 0) {
          st.Ws.place_loaded(pre, __popcll(ms) - 1); st.We.place_loaded(pre, __popcll(me) - 1);
        } else {
          const int *top = a.topT;
          int ps, pe;
          rank_pair(st.sg, st.Ws, top, kS, st.We, top, kE, lane, ps, pe);
          st.Ws.place(st.sg, ps, lane); st.We.place(st.sg, pe, lane);
        }
        st.vs = st.ve = true;
      }
    }
  }
  while (at < nMine) {
    if ((!WEIGHTED || a.wfast) && at + 256 <= nMine && st.sg.cls >= 0) {
      Tri t[4]; int w4[4];
      const char *p = base + (size_t)at * 12;
#pragma unroll
      for (int r = 0; r < 4; ++r) { t[r] = load_tri(p + 768 * r); w4[r] = WEIGHTED ? weights[first + at + 64 * r + lane] : 1; }
      bool odd = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) odd |= t[r].c != st.sg.cls || t[r].s > t[r].e;
      if (__ballot(odd) == 0) {
        if (st.sg.start == st.sg.end) { at += 256; continue; }    // a class without reference regions
        const int ks[4] = {t[0].s, t[1].s, t[2].s, t[3].s}, ke[4] = {t[0].e, t[1].e, t[2].e, t[3].e};
        if (cov_step4_ok(st.Ws, ks, lane, st.vs) && cov_step4_ok(st.We, ke, lane, st.ve)) {
          if constexpr (WEIGHTED) {
            i64 runW = 0;                                          // prefix sums of the weights: once per step, for both windows
#pragma unroll
            for (int r = 0; r < 4; ++r) { const i64 v = w4[r], pw = (i64)wave_scan_add64((u64)v); ldsW[wid][64 * r + lane] = runW + pw - v; runW += rd64(pw, 63); }
            if (lane == 0) ldsW[wid][256] = runW;
            const int len0 = rdlane(ke[0] - ks[0], 0);             // reads of one length: the second window searches the keys the first one staged
            bool uneven = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) uneven |= ke[r] - ks[r] != len0;
            i64 runWK = 0;
            cov_step4_run_w<false>(st.Ws, st.sg, ks, w4, runW, lane, st.vs, ldsK[wid], ldsW[wid], ldsWK[wid], runWK);
            if (len0 >= 0 && __ballot(uneven) == 0) cov_step4_run_w<true>(st.We, st.sg, ke, w4, runW, lane, st.ve, ldsK[wid], ldsW[wid], ldsWK[wid], runWK, len0);
            else cov_step4_run_w<false>(st.We, st.sg, ke, w4, runW, lane, st.ve, ldsK[wid], ldsW[wid], ldsWK[wid], runWK);
          } else {
            // reads of one length (the usual case): the ends are the starts plus a constant -- the second window searches the keys the
            // first one staged
            const int len0 = rdlane(ke[0] - ks[0], 0);
            bool uneven = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) uneven |= ke[r] - ks[r] != len0;
            const bool staged = cov_step4_run<false>(st.Ws, st.sg, ks, lane, st.vs, ldsK[wid], ldsP[wid]);
            if (staged && len0 >= 0 && __ballot(uneven) == 0) cov_step4_run<true>(st.We, st.sg, ke, lane, st.ve, ldsK[wid], ldsP[wid], len0);
            else cov_step4_run<false>(st.We, st.sg, ke, lane, st.ve, ldsK[wid], ldsP[wid]);
          }
          at += 256;
          continue;
        }
      }
    }
    const int left = nMine - at;
    const u64 active = left >= 64 ? ~0ull : ((1ull << left) - 1);
    Tri t; t.c = -1; t.s = 0; t.e = 0; int w = 1;
    if (lane < left) { t = load_tri(base + (size_t)at * 12); if (WEIGHTED) w = weights[first + at + lane]; }
    cov_chunk(st, a, t, w, active, first + at, lane);
    at += 64;
  }
  if (st.vs) st.Ws.flush(st.sg, lane);
  if (st.ve) st.We.flush(st.sg, lane);
  if (lane == 0) {
    if (st.nNoClass) atomicAdd((u64 *)&a.info->n_no_class, (u64)st.nNoClass);
    if (st.nDegen) { atomicAdd((u64 *)&a.info->n_degenerate, (u64)st.nDegen); atomicMin((i64 *)&a.info->first_degenerate, st.firstDegen + a.indexBase); }
  }
}

// Residency (same-box A/B, 100 M x 1 M, round 4): the unweighted kernel takes 92 VGPRs = 5 waves per SIMD left alone; forced to 6 it
// spills four registers inside the loop and is 10 % SLOWER (0.363 against 0.327 ms), at 7 more so.  The weighted one takes 104 = 4
// waves left alone and fits 80 without a spill when told to: 0.577 -> 0.504 ms.
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void coverage_walk_kernel(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CoverArgs a)
{
  coverage_walk_body<WEIGHTED>(reads, weights, n, a);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(6, 6))) void coverage_walk_kernel_weighted(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, CoverArgs a)
{
  coverage_walk_body<true>(reads, weights, n, a);
}

// ---------------------------------------------------------------------------------------------
// Order-agnostic count kernel ("search"): every read does two binary searches in the (L2 /
// Infinity-Cache resident) boundary arrays.  Used when the caller does not claim sorted reads.
// ---------------------------------------------------------------------------------------------
// all lanes of the wave call this; slot < 0 = nothing to add.  One atomic per run of equal slots.
__device__ __forceinline__ void run_add(u64 *__restrict__ hist, int slot, int lane)
{
  const int prev = lane_prev(slot);
  const bool head = lane == 0 || slot != prev;
  const u64 heads = __ballot(head);
  if (head && slot >= 0) {
    const u64 rest = lane == 63 ? 0 : heads >> (lane + 1);
    const int len = rest ? __builtin_ctzll(rest) + 1 : 64 - lane;
    atomicAdd(&hist[slot], (u64)len);
  }
}

template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void count_search_kernel(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n,
                                                            CountArgs a)
{
  // top level of both searches in LDS: every (1 << sampShift)-th boundary of each array
  extern __shared__ int smp[];
  int *__restrict__ se = smp, *__restrict__ ss = smp + a.nSamp;
  for (int i = threadIdx.x; i < a.nSamp; i += blockDim.x) { se[i] = a.sampE[i]; ss[i] = a.sampS[i]; }
  __syncthreads();
  const int sh = a.sampShift, rnd = (1 << sh) - 1;
  i64 nNoClass = 0, nDegen = 0, firstDegen = INT64_MAX;
  const int lane = threadIdx.x & 63;
  for (i64 base = (i64)blockIdx.x * blockDim.x; base < n; base += (i64)gridDim.x * blockDim.x) {
    const i64 i = base + threadIdx.x;
    int slotA = -1, slotB = -1;
    i64 w = 1;
    if (i < n) {
      const Tri t = reads[i];
      if (WEIGHTED) w = (i64)weights[i];
      const bool known = class_counts(a, t.c);
      const int s0 = known ? a.segStart[t.c] : 0, s1 = known ? a.segStart[t.c + 1] : 0;
      if (!known) nNoClass++;
      else if (t.s > t.e + a.zeroLenOk) { nDegen++; if (i < firstDegen) firstDegen = i; side_append(a, t.c, t.s, t.e, (int)w, true); }
      else if (s0 != s1) {
        // samples i0..i1-1 are the ones that lie inside the class segment [s0, s1)
        const int i0 = (int)(((i64)s0 + rnd) >> sh), i1 = (int)(((i64)s1 + rnd) >> sh);
        int lo = i0, hi = i1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (se[mid] < t.s) lo = mid + 1; else hi = mid; }
        int glo = lo == i0 ? s0 : ((lo - 1) << sh) + 1, ghi = lo == i1 ? s1 : lo << sh;
        while (glo < ghi) { int mid = (int)(((i64)glo + ghi) >> 1); if (a.sortedE[mid] < t.s) glo = mid + 1; else ghi = mid; }
        slotA = glo + t.c;
        lo = i0; hi = i1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ss[mid] <= t.e) lo = mid + 1; else hi = mid; }
        glo = lo == i0 ? s0 : ((lo - 1) << sh) + 1; ghi = lo == i1 ? s1 : lo << sh;
        while (glo < ghi) { int mid = (int)(((i64)glo + ghi) >> 1); if (a.sortedS[mid] <= t.e) glo = mid + 1; else ghi = mid; }
        slotB = glo + t.c;
      }
    }
    if (WEIGHTED) {
      if (slotA >= 0) { atomicAdd(&a.histA[slotA], (u64)w); atomicAdd(&a.histB[slotB], (u64)w); }
    } else {
      // neighbouring lanes that landed in the same slot (reads that happen to be in order) share one atomic
      run_add(a.histA, slotA, lane);
      run_add(a.histB, slotB, lane);
    }
  }
  if (nNoClass) atomicAdd((u64 *)&a.info->n_no_class, (u64)nNoClass);
  if (nDegen) { atomicAdd((u64 *)&a.info->n_degenerate, (u64)nDegen); atomicMin((i64 *)&a.info->first_degenerate, firstDegen + a.indexBase); }
}

// ---------------------------------------------------------------------------------------------
// Finalize (2 launches):
//   finalize_scan_kernel   one block per tile of kTile histogram slots: tile offset = sum of the
//                          tile sums below it (kept up to date by the streaming kernel), local scan,
//                          writes the inclusive prefixes PA/PB and ZEROES the histograms for the next call
//   gather_hits_kernel     hits[k] = (PA[posE[k]] - PA[classBase[k]]) - (PB[posS[k]] - PB[classBase[k]])
//                          and zeroes the tile sums
// (tile_sums_kernel rebuilds the tile sums for the search kernel, which does not maintain them.)
// ---------------------------------------------------------------------------------------------
// four consecutive histogram slots of a thread (32-byte aligned for 64-bit slots, 16-byte for 32-bit ones: i0 is a multiple of 4)
template <class T> __device__ __forceinline__ void load4(const T *p, u64 (&v)[4]);
template <> __device__ __forceinline__ void load4<u64>(const u64 *p, u64 (&v)[4])
{ const ulonglong2 x = *(const ulonglong2 *)p, y = *(const ulonglong2 *)(p + 2); v[0] = x.x; v[1] = x.y; v[2] = y.x; v[3] = y.y; }
template <> __device__ __forceinline__ void load4<unsigned>(const unsigned *p, u64 (&v)[4])
{ const uint4 x = *(const uint4 *)p; v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
template <class T> __device__ __forceinline__ void store4(T *p, u64 a, u64 b, u64 c, u64 d);
template <> __device__ __forceinline__ void store4<u64>(u64 *p, u64 a, u64 b, u64 c, u64 d) { *(ulonglong2 *)p = make_ulonglong2(a, b); *(ulonglong2 *)(p + 2) = make_ulonglong2(c, d); }
template <> __device__ __forceinline__ void store4<unsigned>(unsigned *p, u64 a, u64 b, u64 c, u64 d) { *(uint4 *)p = make_uint4((unsigned)a, (unsigned)b, (unsigned)c, (unsigned)d); }

__device__ __forceinline__ u64 block_sum(u64 v, u64 *lds)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if (lane == 0) lds[wv] = v;
  __syncthreads();
  u64 s = 0;
  for (int k = 0; k < (int)(blockDim.x >> 6); k++) s += lds[k];
  return s;
}

// tileList (may be null): the tiles to work on -- a member of a group only has counts in the tiles of the classes it owns
// (T: the histograms' slot type; the tile sums are 64-bit either way)
template <class T>
__global__ __launch_bounds__(256) void tile_sums_kernel(const T *__restrict__ ha, const T *__restrict__ hb, i64 len,
                                                        u64 *__restrict__ pa, u64 *__restrict__ pb, const int *__restrict__ tileList)
{
  // grid (tiles, 2): blockIdx.y picks the histogram; thread t owns 4 consecutive slots
  __shared__ u64 lds[4];
  const T *__restrict__ h = blockIdx.y ? hb : ha;
  const int tileIdx = tileList ? tileList[blockIdx.x] : (int)blockIdx.x;
  const i64 i0 = (i64)tileIdx * kTile + (i64)threadIdx.x * 4;
  u64 s = 0;
  if (i0 + 4 <= len) { u64 v[4]; load4<T>(h + i0, v); s = v[0] + v[1] + v[2] + v[3]; }
  else for (int k = 0; k < 4; k++) if (i0 + k < len) s += h[i0 + k];
  s = block_sum(s, lds);
  if (threadIdx.x == 0) (blockIdx.y ? pb : pa)[tileIdx] = s;
}

// grid (tiles, 2): blockIdx.y picks the histogram (A or B) -- twice the blocks, half the work per block
template <class T>
__global__ __launch_bounds__(256) void finalize_scan_kernel(T *__restrict__ ha, T *__restrict__ hb, i64 len,
                                                            const u64 *__restrict__ ta, const u64 *__restrict__ tb,
                                                            T *__restrict__ pa, T *__restrict__ pb, const int *__restrict__ tileList)
{
  __shared__ u64 lds[4];
  __shared__ u64 wsum[4];
  T *__restrict__ h = blockIdx.y ? hb : ha;
  const u64 *__restrict__ ts = blockIdx.y ? tb : ta;
  T *__restrict__ p = blockIdx.y ? pb : pa;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tile = tileList ? tileList[blockIdx.x] : (int)blockIdx.x;   // (tiles not in the list hold no counts: their sums are 0)
  // thread t owns 4 consecutive slots; issued first, used last
  const i64 i0 = (i64)tile * kTile + (i64)threadIdx.x * 4;
  const bool full = i0 + 4 <= len;                   // (the histograms come from hipMalloc, i0 is a multiple of 4)
  u64 v[4];
  if (full) load4<T>(h + i0, v);
  else {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = i0 + k < len ? (u64)h[i0 + k] : 0;
  }
  // offset of this tile: sum of the tile sums below it
  u64 o = 0;
  for (int t = threadIdx.x; t < tile; t += 256) o += ts[t];
  o = block_sum(o, lds);
  v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
  const u64 x = wave_scan_add64(v[3]);             // inclusive scan of the thread totals across the wave (DPP)
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  o += x - v[3];
  for (int k = 0; k < wv; k++) o += wsum[k];
  if (full) { store4<T>(p + i0, v[0] + o, v[1] + o, v[2] + o, v[3] + o); store4<T>(h + i0, 0, 0, 0, 0); }
  else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i0 + k < len) { p[i0 + k] = (T)(v[k] + o); h[i0 + k] = 0; }
  }
}

// Tile sums and prefix scan in ONE launch (the streaming kernel of a large call leaves the tile sums to the finalize step): a tile's
// block publishes its sum and then adds up the sums of the tiles below it as they turn up.  A block's place in the chain is a TICKET
// drawn when it starts (one counter per histogram, never reset: `base` is what the calls before have drawn from it, kept by the host;
// position = the tile number, or the place in a member's tile list: the tiles outside the list hold no counts), so a block only ever
// waits for blocks that started before it -- resident or done, whatever order the dispatcher starts workgroups in and whatever else
// runs on the device (a second chained scan on another stream, a streaming kernel that fills the chip): the lowest unfinished
// ticket never waits, so every wait ends.  Same-address returning atomics serialise at ~12 ns (scripts/membench.hip), which is why the
// launcher takes this kernel for at most kChainMaxTiles tiles.  The wait is bounded all the same: a block that has polled
// kChainSpinMax times gives up, raises DevInfo::fault (the call's result is then refused by gtx_last_info) and finishes with what it has.
static constexpr unsigned kChainSpinMax = 1u << 22;           // polls of one word (~1 us each under load): seconds
static constexpr int kChainMaxTiles = 512;
template <class T>
__global__ __launch_bounds__(256) void finalize_scan_chained_kernel(T *__restrict__ ha, T *__restrict__ hb, i64 len,
                                                                    u64 *fa, u64 *fb, u64 *ctlA, u64 *ctlB, u64 base, unsigned epoch, int nRun,
                                                                    T *__restrict__ pa, T *__restrict__ pb, const int *__restrict__ tileList,
                                                                    DevInfo *info)
{
  __shared__ u64 lds[4];
  __shared__ u64 wsum[4];
  __shared__ u64 tk;
  const bool second = blockIdx.y != 0;
  if (threadIdx.x == 0) tk = atomicAdd(second ? ctlB : ctlA, 1ull) - base;
  __syncthreads();
  if (tk >= (u64)nRun) { if (threadIdx.x == 0) atomicAdd((u64 *)&info->fault, 1ull); return; }   // (the host's count of draws is off: touch nothing)
  T *__restrict__ h = second ? hb : ha;
  u64 *fl = second ? fb : fa;
  T *__restrict__ p = second ? pb : pa;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int pos = (int)tk, tile = tileList ? tileList[pos] : pos;
  const i64 i0 = (i64)tile * kTile + (i64)threadIdx.x * 4;
  const bool full = i0 + 4 <= len;
  u64 v[4];
  if (full) load4<T>(h + i0, v);
  else {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = i0 + k < len ? (u64)h[i0 + k] : 0;
  }
  v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
  const u64 x = wave_scan_add64(v[3]);             // inclusive scan of the thread totals across the wave (DPP)
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  // (no fences: a sum travels as two words, each half of it next to the epoch -- relaxed device-scope atomics, complete when both carry the epoch)
  if (threadIdx.x < 2) {
    const u64 s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __hip_atomic_store(fl + 2 * pos + threadIdx.x, ((u64)epoch << 32) | (threadIdx.x ? s >> 32 : s & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  u64 o = 0;
  bool gaveUp = false;
  for (int t = threadIdx.x; t < pos; t += 256) {
    u64 lo, hi; unsigned spins = 0;
    while (((lo = __hip_atomic_load(fl + 2 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch && ++spins < kChainSpinMax) __builtin_amdgcn_s_sleep(1);
    while (((hi = __hip_atomic_load(fl + 2 * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch && ++spins < kChainSpinMax) __builtin_amdgcn_s_sleep(1);
    if (spins >= kChainSpinMax) { gaveUp = true; break; }
    o += (lo & 0xffffffffull) | (hi << 32);
  }
  if (gaveUp) atomicAdd((u64 *)&info->fault, 1ull);
  o = block_sum(o, lds);
  o += x - v[3];
  for (int k = 0; k < wv; k++) o += wsum[k];
  if (full) { store4<T>(p + i0, v[0] + o, v[1] + o, v[2] + o, v[3] + o); store4<T>(h + i0, 0, 0, 0, 0); }
  else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i0 + k < len) { p[i0 + k] = (T)(v[k] + o); h[i0 + k] = 0; }
  }
}

// (T = unsigned: the differences are taken modulo 2^32 -- every prefix and every count is below the number of reads, which is)
template <class T>
__global__ __launch_bounds__(256) void gather_hits_kernel(const T *__restrict__ pa, const T *__restrict__ pb,
                                                          const int *__restrict__ posE, const int *__restrict__ posS,
                                                          const int *__restrict__ classBase, i64 m, u64 *__restrict__ hits,
                                                          u64 *__restrict__ ta, u64 *__restrict__ tb, int nTiles, DevInfo *nextInfo,
                                                          const int *__restrict__ regionList, int scatter)
{
  // regionList (may be null): m entries, hits[j] = the count of region regionList[j] -- a group member's own regions, compact
  // (scatter: hits[regionList[j]] -- the same regions at their places in the file's order, the other entries untouched)
  i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nTiles) { ta[k] = 0; tb[k] = 0; }         // the tile sums have been consumed: clean for the next call
  if (k == 0) { nextInfo->first_unsorted = INT64_MAX; nextInfo->n_no_class = 0; nextInfo->n_degenerate = 0; nextInfo->first_degenerate = INT64_MAX; nextInfo->n_unplaced = 0; nextInfo->fault = 0; }
  if (k >= m) return;
  i64 outIdx = k;
  if (regionList) { k = regionList[k]; if (scatter) outIdx = k; }
  int pe = posE[k];
  u64 h = 0;
  if (pe >= 0) {
    int cb = classBase[k];                       // slot just below the class's first slot, -1 if none
    T ba = cb >= 0 ? pa[cb] : 0, bb = cb >= 0 ? pb[cb] : 0;
    h = (u64)(T)((T)(pa[pe] - ba) - (T)(pb[posS[k]] - bb));
  }
  hits[outIdx] = h;
}

__global__ __launch_bounds__(256) void gather_coverage_kernel(CoverGather g, i64 m, u64 *__restrict__ cov, int nTiles, DevInfo *nextInfo)
{
  i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nTiles) {
#pragma unroll
    for (int q = 0; q < 4; q++) g.part[q][k] = 0;
  }
  if (k == 0) { nextInfo->first_unsorted = INT64_MAX; nextInfo->n_no_class = 0; nextInfo->n_degenerate = 0; nextInfo->first_degenerate = INT64_MAX; nextInfo->n_unplaced = 0; nextInfo->fault = 0; }
  if (k >= m) return;
  const int pe = g.posTE[k];
  u64 c = 0;
  if (pe >= 0 && g.refE[k] >= g.refS[k]) {          // zero-length regions: coverage 0
    const int ps = g.posTS[k], cb = g.classBaseT[k];
    u64 v[8];
#pragma unroll
    for (int q = 0; q < 4; q++) { const u64 *p = g.pref[q]; const u64 b = cb >= 0 ? p[cb] : 0; v[q] = p[pe] - b; v[4 + q] = p[ps] - b; }
    // v: 0 Ws(E) 1 Fs(E) 2 We(E) 3 Fe(E) 4 Ws(S-1) 5 Fs(S-1) 6 We(S-1) 7 Fe(S-1)
    const u64 E = (u64)(i64)g.refE[k], S = (u64)(i64)g.refS[k];
    c = v[3] - v[7] + E * (v[0] - v[2]) - v[1] + v[5] - S * (v[4] - v[6]) + v[0] - v[6];
  }
  cov[k] = c;
}

// ---------------------------------------------------------------------------------------------
// genomic_scans counts: micro-window histogram (UnsortedGenomicRegionSetScanner ctor,
// genomic_intervals.cpp:5036-5055) and sliding sums (:5058-5075).
// Sorted reads put runs of equal micro-window index in neighbouring lanes: one atomic per run.
// ---------------------------------------------------------------------------------------------
// previous lane's value (lane 0 keeps its own): DPP wave_shr:1, no LDS traffic
// micro-window (class, index) of one read, or (-1,-1) when it does not count
// clsU >= 0: the caller knows that every read of the step has class clsU (valid) and passes its micro-window count nmU --
// a scalar instead of a per-lane table lookup, i.e. no memory round trip between the arrival of the reads and their use
__device__ __forceinline__ void scan_slot(const Tri &t, bool in_range, const ScanArgs &a, int &cls, int &mw, int clsU = -1, i64 nmU = 0)
{
  cls = -1; mw = -1;
  if (in_range && (clsU >= 0 || (unsigned)t.c < (unsigned)a.nClasses) && (a.sortedRule || (t.s <= t.e && t.e > 0))) {
    i64 pos = a.center ? (i64)t.s + ((i64)t.e - t.s) / 2 : (i64)t.s;
    if (a.sortedRule && pos < 1) pos = 1;           // the sorted scanner takes START <= stop of the first window
    if (pos >= 1) {
      // (pos-1) / winStep without a 64-bit divide: 32-bit reciprocal estimate (never too high, at most 1 low) + fix-up
      const unsigned x = (unsigned)(pos - 1), d = (unsigned)a.winStep;
      unsigned q = d == 1 ? x : __umulhi(x, a.winStepInv);
      unsigned r = x - q * d;
      if (r >= d) { q++; r -= d; }
      if (r >= d) q++;
      if ((i64)q < (clsU >= 0 ? nmU : a.nMicro[t.c])) { cls = t.c; mw = (int)q; }
    }
  }
}

// Per-wave tile of micro-window counters in LDS.  Sorted reads stay inside a window of consecutive
// micro-windows for a long time, so their increments are LDS atomics (pre-aggregated per run of equal
// lanes) and reach HBM only when the tile is flushed: 64 contiguous 8-byte atomics per instruction,
// i.e. whole 64-byte requests at the memory side instead of one request per read.
static constexpr int kScanTile = 1024;

// (unweighted scans count in 32 bits -- at most n_reads < 2^32 per micro-window -- which halves the
//  atomic, memset and window-sum traffic; weighted scans keep the reference's 64-bit counters, in the tile too)
template <bool WEIGHTED>
struct ScanTile { typedef typename std::conditional<WEIGHTED, u64, unsigned>::type ct; ct *lds; int cls, base; bool used; };

template <bool WEIGHTED>
__device__ __forceinline__ void scan_tile_flush(ScanTile<WEIGHTED> &T, const ScanArgs &a, int lane)
{
  typedef typename ScanTile<WEIGHTED>::ct ct;
  if (!T.used) return;
  ct *dst = (ct *)a.micro + a.microOff[T.cls] + T.base;
  for (int k = 0; k < kScanTile; k += 64) {
    ct v = T.lds[k + lane];
    if (__ballot(v != 0) == 0) continue;
    if (v != 0) { atomicAdd(&dst[k + lane], v); T.lds[k + lane] = 0; }
  }
  T.used = false;
}

template <bool WEIGHTED>
__device__ __forceinline__ void scan_add64(const Tri &t, int w, bool in_range, const ScanArgs &a, ScanTile<WEIGHTED> &T, int lane, int clsU = -1, i64 nmU = 0)
{
  int cls, mw;
  scan_slot(t, in_range, a, cls, mw, clsU, nmU);
  const u64 valid = __ballot(mw >= 0);
  if (valid == 0) return;
  // keep the tile where the reads are: if the first counting read of this register is outside, move the tile there
  const int f = __ffsll((unsigned long long)valid) - 1;
  const int fc = rdlane(cls, f), fm = rdlane(mw, f);
  if (!T.used || fc != T.cls || (unsigned)(fm - T.base) >= (unsigned)kScanTile) {
    scan_tile_flush(T, a, lane);
    T.cls = fc; T.base = fm; T.used = true;
  }
  const bool inTile = mw >= 0 && cls == T.cls && (unsigned)(mw - T.base) < (unsigned)kScanTile;
  if constexpr (WEIGHTED) {
    // weighted: every read adds its weight -- to the tile (64-bit LDS atomic) or, outside it, straight to HBM
    if (inTile) atomicAdd(&T.lds[mw - T.base], (u64)(i64)w);
    else if (mw >= 0) atomicAdd(&a.micro[a.microOff[cls] + mw], (u64)(i64)w);
    return;
  }
  // runs of equal slots in neighbouring lanes: the first lane of a run adds the run length
  const int key = inTile ? mw : -1 - lane;         // lanes outside the tile never join a run
  const int pk = lane_prev(key);                   // (all lanes active here: DPP must not run under a partial exec mask)
  const bool head = lane == 0 || pk != key;
  const u64 heads = __ballot(head);
  if (inTile && head) {
    const u64 later = lane == 63 ? 0 : (heads >> (lane + 1));
    const int run = later ? __ffsll((unsigned long long)later) : 64 - lane;
    atomicAdd(&T.lds[mw - T.base], (unsigned)run);
  }
  if (mw >= 0 && !inTile) atomicAdd((unsigned *)a.micro + a.microOff[cls] + mw, 1u);   // scattered read: straight to HBM
}

// genomic_scans counts, histogram pass: like the count kernel a wave takes 4 x 64 reads per step
// (four coalesced non-temporal 768-byte requests in flight), spans are dealt to waves contiguously.
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void scan_hist_kernel(const Tri *__restrict__ reads, const int *__restrict__ weights, i64 n, ScanArgs a, const int *__restrict__ runIf)
{
  if (runIf && *runIf == 0) return;                  // fallback of the owner-computes pass (gtx_scanown.hip): only when it gave up
  constexpr int R = 4;
  __shared__ typename ScanTile<WEIGHTED>::ct tiles[4][kScanTile];
  const int lane = threadIdx.x & 63;
  const int wv = rfl(threadIdx.x >> 6);
  const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + wv;
  const i64 nWaves = (i64)gridDim.x * (blockDim.x >> 6);
  const i64 nSteps = (n + 64 * R - 1) / (64 * R);
  const i64 per = (nSteps + nWaves - 1) / nWaves;
  i64 s0 = wave * per, s1 = s0 + per; if (s1 > nSteps) s1 = nSteps;
  ScanTile<WEIGHTED> T; T.lds = tiles[wv]; T.cls = -1; T.base = 0; T.used = false;
  for (int k = lane; k < kScanTile; k += 64) T.lds[k] = 0;
  int cachedCls = -1; i64 cachedNm = 0;
  for (i64 s = s0; s < s1; ++s) {
    const i64 at = s * 64 * R;
    Tri t[R]; int w[R];
    if (at + 64 * R <= n) {
      const char *p = (const char *)(reads + at) + (unsigned)lane * 12u;
#pragma unroll
      for (int r = 0; r < R; ++r) { t[r] = load_tri(p + 768 * r); w[r] = WEIGHTED ? weights[at + 64 * r + lane] : 1; }
      // one class for the whole step (the rule on sorted reads): its micro-window count comes from a scalar load
      const int c0 = rdlane(t[0].c, 0);
      int odd = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) odd |= t[r].c ^ c0;
      int clsU = -1; i64 nmU = 0;
      if ((unsigned)c0 < (unsigned)a.nClasses && __ballot(odd != 0) == 0) {
        if (c0 != cachedCls) { cachedCls = c0; cachedNm = a.nMicro[c0]; }
        clsU = c0; nmU = cachedNm;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) scan_add64<WEIGHTED>(t[r], w[r], true, a, T, lane, clsU, nmU);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const i64 i = at + 64 * r + lane;
        t[r].c = -1; t[r].s = 0; t[r].e = 0; w[r] = 1;
        if (i < n) { t[r] = reads[i]; if (WEIGHTED) w[r] = weights[i]; }
        scan_add64<WEIGHTED>(t[r], w[r], i < n, a, T, lane);
      }
    }
  }
  scan_tile_flush(T, a, lane);
}

// Window sums: out[k] = sum_{j<comb} micro[k+j].  A block owns a tile of kWinTile consecutive windows
// of one class: the tile's kWinTile+comb-1 micro-windows go through LDS once (coalesced), every thread
// then slides over kWinPer consecutive windows (first sum, then +new -old).
static constexpr int kWinPer = 8;
static constexpr int kWinTile = 256 * kWinPer;
static constexpr int kWinMaxComb = 2048;           // LDS: (kWinTile + kWinMaxComb) * 9/8 * 8 B = 36 KB

template <class MT>
__global__ __launch_bounds__(256) void scan_window_kernel(const MT *__restrict__ micro, ScanArgs a, u64 *__restrict__ out, const int *__restrict__ runIf)
{
  if (runIf && *runIf == 0) return;
  __shared__ u64 lds[(kWinTile + kWinMaxComb) / 8 * 9 + 8];
  // class of this tile (tileOff is a prefix over classes; few dozen entries)
  int c = 0;
  while (c + 1 < a.nClasses && (i64)blockIdx.x >= a.tileOff[c + 1]) c++;
  const i64 nWin = a.winOff[c + 1] - a.winOff[c];
  const i64 k0 = ((i64)blockIdx.x - a.tileOff[c]) * kWinTile;
  if (k0 >= nWin) return;
  const i64 cntWin = nWin - k0 < kWinTile ? nWin - k0 : kWinTile;
  const MT *src = micro + a.microOff[c] + k0;
  u64 *dst = out + a.outOff[c] + k0;
  if (a.comb == 1) {                               // windows == micro-windows: plain copy
    for (i64 i = threadIdx.x; i < cntWin; i += 256) dst[i] = src[i];
    return;
  }
  if (a.comb > kWinMaxComb) {                      // very long windows: direct sums
    for (i64 i = threadIdx.x; i < cntWin; i += 256) { u64 s = 0; for (int j = 0; j < a.comb; j++) s += src[i + j]; dst[i] = s; }
    return;
  }
  // LDS index with one spare slot per 8: a thread slides over 8 consecutive windows, so neighbouring lanes are 8 elements
  // apart -- 9 after padding: conflict-free for the 32-bit micro-windows of unweighted scans (the stride of 8 put all lanes
  // on two banks).  The sums go back through the same buffer so that the stores to `out` are contiguous across the wave.
  MT *in = (MT *)lds;
  const i64 need = cntWin + a.comb - 1;
  for (i64 i = threadIdx.x; i < need; i += 256) in[i + (i >> 3)] = src[i];
  __syncthreads();
  const i64 w0 = (i64)threadIdx.x * kWinPer;
  u64 sum[kWinPer];
  if (w0 < cntWin) {
    u64 s = 0;
    for (int j = 0; j < a.comb; j++) { const i64 i = w0 + j; s += in[i + (i >> 3)]; }
    sum[0] = s;
#pragma unroll
    for (int q = 1; q < kWinPer; q++) {
      if (w0 + q < cntWin) { const i64 ia = w0 + q + a.comb - 1, ib = w0 + q - 1; s += (u64)in[ia + (ia >> 3)] - (u64)in[ib + (ib >> 3)]; }
      sum[q] = s;
    }
  }
  __syncthreads();                                 // all reads of `in` are done: the buffer now takes the sums
  if (w0 < cntWin) {
#pragma unroll
    for (int q = 0; q < kWinPer; q++) if (w0 + q < cntWin) { const i64 i = w0 + q; lds[i + (i >> 3)] = sum[q]; }
  }
  __syncthreads();
  for (i64 i = threadIdx.x; i < cntWin; i += 256) dst[i] = lds[i + (i >> 3)];
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
int search_sample_shift(i64 nValid)
{
  int sh = 6;
  while ((((nValid + (1ll << sh) - 1) >> sh) * 2 * (i64)sizeof(int)) > kSearchLdsBytes) sh++;
  return sh;
}

int scan_tiles(i64 len) { return (int)((len + kTile - 1) / kTile); }

// Schedule of a launch over nChunks 64-read chunks, `cpw` chunks per wave in the main segment (all spans multiples of `r`),
// `slots` = resident waves of the chip.  A wave lives p + c * tau (placement + c chunks at the chip's rate shared by all
// slots; p / tau ~ 8 chunks measured), and a slot frees once per life.
//  * tail (default for launches of three rounds or more): a wave dispatched when the launch has time T left should take T's
//    worth -- levels cpw-r, cpw-2r, ... down to 8 chunks, each dealt to about as many waves as free up while the level is
//    current, slots * r / (cpw + 8), times 5/4 (measured: 100 M reads, 512 / 640 / 800 waves per level within 1 % of each
//    other, all 3-4 % ahead of equal spans; same-box A/B, scripts/ab_count.py).
//  * head (off by default): levels 12, 12+r, ... below cpw spread the ends of the first round, whose waves all start together,
//    over a wave's life.  The dips in the wave time line (scripts/wave_trace.py) go, and so does the first round's burst at
//    full occupancy: no net gain measured.
// GTX_SCHED: "none" | "lin:<waves per head level>:<waves per tail level>[:<shortest span>]" | "h=<c>x<w>,...;t=<c>x<w>,..."
// (chunks x waves, in launch order) for experiments.
SpanSchedule span_schedule(i64 nChunks, int cpw, int r, i64 slots)
{
  SpanSchedule s;
  for (int i = 0; i < SpanSchedule::kMax; ++i) { s.wave0[i] = INT32_MAX; s.chunk0[i] = 0; s.cpw[i] = cpw; }
  struct SegSpec { int cpw; i64 waves; };
  std::vector<SegSpec> head, tail;
  auto up = [&](i64 c) { return (int)((c + r - 1) / r * r); };
  auto linear = [&](i64 perHead, i64 perTail, int minSpan = 8) {
    if (perHead > 0) for (int c = up(12); c < cpw; c += r) head.push_back({c, (perHead + 3) / 4 * 4});
    if (perTail > 0) for (int c = cpw - r; c >= minSpan; c -= r) tail.push_back({c, (perTail + 3) / 4 * 4});
  };
  auto parse = [&](const char *p, std::vector<SegSpec> &out) {
    while (*p && *p != ';') {
      char *q; const long c = strtol(p, &q, 10); if (q == p || *q != 'x') break;
      p = q + 1; const long long w = strtoll(p, &q, 10); if (q == p) break;
      if (c > 0 && w > 0) out.push_back({up(c), (i64)(w + 3) / 4 * 4});
      p = *q == ',' ? q + 1 : q;
    }
    return p;
  };
  const char *spec = getenv("GTX_SCHED");
  if (spec && !strncmp(spec, "lin:", 4)) { long long a = 0, b = 0; int ms = 8; sscanf(spec + 4, "%lld:%lld:%d", &a, &b, &ms); linear(a, b, ms > 0 ? ms : 8); }
  else if (spec && !strcmp(spec, "none")) {}
  else if (spec) {
    const char *p = spec;
    while (*p) {
      if (!strncmp(p, "h=", 2)) p = parse(p + 2, head); else if (!strncmp(p, "t=", 2)) p = parse(p + 2, tail); else break;
      if (*p == ';') ++p;
    }
  } else if (nChunks >= 3 * slots * (i64)cpw) {
    linear(0, slots * r * 5 / (4 * (cpw + 8)));
  }
  auto chunksOf = [](const std::vector<SegSpec> &v) { i64 t = 0; for (auto &g : v) t += g.cpw * g.waves; return t; };
  while ((int)(head.size() + tail.size()) > SpanSchedule::kMax - 1) { if (!head.empty()) head.erase(head.begin()); else tail.pop_back(); }
  if ((chunksOf(head) + chunksOf(tail)) * 3 > nChunks * 2) head.clear();
  if (chunksOf(tail) * 3 > nChunks * 2) tail.clear();
  int k = 0; i64 wave = 0, chunk = 0;
  for (auto &g : head) { s.wave0[k] = (int)wave; s.chunk0[k] = (int)chunk; s.cpw[k] = g.cpw; wave += g.waves; chunk += g.waves * g.cpw; ++k; }
  const i64 tailChunks = chunksOf(tail);
  const i64 mainWaves = tail.empty() ? (nChunks - chunk + cpw - 1) / cpw : (nChunks - chunk - tailChunks) / cpw;
  s.wave0[k] = (int)wave; s.chunk0[k] = (int)chunk; s.cpw[k] = cpw; wave += mainWaves; chunk += mainWaves * cpw; ++k;
  for (size_t i = 0; i < tail.size(); ++i) {
    i64 w = tail[i].waves;
    if (i + 1 == tail.size()) w = (nChunks - chunk + tail[i].cpw - 1) / tail[i].cpw;   // the last level takes what is left
    s.wave0[k] = (int)wave; s.chunk0[k] = (int)chunk; s.cpw[k] = tail[i].cpw; wave += w; chunk += w * tail[i].cpw; ++k;
  }
  s.nSeg = k; s.nWaves = wave;
  return s;
}

hipError_t launch_count(const void *reads, const void *weights, i64 n, const CountArgs &a, bool sortedHint, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  if (sortedHint) {
    const i64 waves = a.sched.nWaves;
    static const int wpb = getenv("GTX_WAVES_PER_BLOCK") ? std::min(4, std::max(1, atoi(getenv("GTX_WAVES_PER_BLOCK")))) : 4;   // (the kernels size their LDS for 4)
    const unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
    const unsigned bs = 64u * wpb;
    // a.prefetch = reads per lane per step (R)
    static const bool wfast = !(getenv("GTX_WEIGHTED_FAST") && atoi(getenv("GTX_WEIGHTED_FAST")) == 0);
    if (weights && wfast) count_walk_kernel_weighted<<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (weights) count_walk_kernel<true, 2><<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (a.prefetch <= 1) count_walk_kernel<false, 1><<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (a.prefetch == 2) count_walk_kernel<false, 2><<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (a.prefetch == 3) count_walk_kernel<false, 3><<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (a.flip && a.hist32) count_walk_kernel_flip_h32<<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (a.flip) count_walk_kernel_flip<<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (a.hist32) count_walk_kernel_h32<<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else if (getenv("GTX_PF") && atoi(getenv("GTX_PF"))) count_walk_kernel_pf<<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else count_walk_kernel<false, 4><<<grid, bs, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
  } else {
    // one 1024-thread block per CU (the LDS top level fills most of the CU's 160 KB)
    static PerDevice attr;
    {
      hipError_t e = attr.once([] {
        hipError_t e2 = hipFuncSetAttribute((const void *)count_search_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSearchLdsBytes);
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void *)count_search_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSearchLdsBytes);
        return e2;
      });
      if (e != hipSuccess) return e;
    }
    i64 blocks = (n + 1023) / 1024; if (blocks > 512) blocks = 512;
    const size_t lds = sizeof(int) * 2 * (size_t)a.nSamp;
    if (weights) count_search_kernel<true><<<(unsigned)blocks, 1024, lds, st>>>((const Tri *)reads, (const int *)weights, n, a);
    else count_search_kernel<false><<<(unsigned)blocks, 1024, lds, st>>>((const Tri *)reads, (const int *)weights, n, a);
  }
  return hipGetLastError();
}

hipError_t launch_tile_sums(u64 *histA, u64 *histB, i64 histLen, u64 *tileA, u64 *tileB, hipStream_t st)
{
  const int nb = scan_tiles(histLen);
  if (nb > 0) tile_sums_kernel<u64><<<dim3(nb, 2), 256, 0, st>>>(histA, histB, histLen, tileA, tileB, nullptr);
  return hipGetLastError();
}

template <class T>
static hipError_t launch_finalize_t(T *histA, T *histB, i64 histLen, u64 *tileA, u64 *tileB, bool tileSumsValid, T *prefA, T *prefB,
                                    const int *posE, const int *posS, const int *classBase, i64 m, u64 *hits, DevInfo *nextInfo, hipStream_t st,
                                    const FinalizeShare *share, unsigned *chainFlags, unsigned epoch, DevInfo *info, unsigned long long *chainDraws)
{
  const int nb = scan_tiles(histLen);
  const int nbRun = share ? share->nTiles : nb;                // a group member: the tiles of its classes, its regions (compact)
  const int *tl = share ? share->tileList : nullptr;
  static const bool chained = !(getenv("GTX_CHAINED_SCAN") && atoi(getenv("GTX_CHAINED_SCAN")) == 0);
  if (nbRun > 0) {
    static const int chainMax = getenv("GTX_CHAIN_MAX_TILES") ? atoi(getenv("GTX_CHAIN_MAX_TILES")) : kChainMaxTiles;
    if (!tileSumsValid && chainFlags && chained && info && chainDraws && nbRun <= chainMax) {
      finalize_scan_chained_kernel<T><<<dim3(nbRun, 2), 256, 0, st>>>(histA, histB, histLen, (u64 *)chainFlags, (u64 *)chainFlags + 2 * (nb + 2), (u64 *)chainFlags + 2 * nb,
                                                                      (u64 *)chainFlags + 2 * (nb + 2) + 2 * nb, *chainDraws, epoch, nbRun, prefA, prefB, tl, info);
      *chainDraws += (unsigned long long)nbRun;
    }
    else {
      if (!tileSumsValid) tile_sums_kernel<T><<<dim3(nbRun, 2), 256, 0, st>>>(histA, histB, histLen, tileA, tileB, tl);
      finalize_scan_kernel<T><<<dim3(nbRun, 2), 256, 0, st>>>(histA, histB, histLen, tileA, tileB, prefA, prefB, tl);
    }
  }
  const i64 mm = share ? share->nRegions : m;
  const i64 work = (mm > nb ? mm : nb) > 0 ? (mm > nb ? mm : nb) : 1;
  gather_hits_kernel<T><<<(unsigned)((work + 255) / 256), 256, 0, st>>>(prefA, prefB, posE, posS, classBase, mm, hits, tileA, tileB, nb, nextInfo,
                                                                        share ? share->regionList : nullptr, share && share->scatter ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_finalize(u64 *histA, u64 *histB, i64 histLen, u64 *tileA, u64 *tileB, bool tileSumsValid, u64 *prefA, u64 *prefB,
                           const int *posE, const int *posS, const int *classBase, i64 m, u64 *hits, DevInfo *nextInfo, hipStream_t st,
                           const FinalizeShare *share, unsigned *chainFlags, unsigned epoch, DevInfo *info, unsigned long long *chainDraws, bool hist32)
{
  // hist32: the call's streaming kernel counted into 32-bit slots (CountArgs::hist32): the same buffers, read as unsigned[]
  if (hist32) return launch_finalize_t<unsigned>((unsigned *)histA, (unsigned *)histB, histLen, tileA, tileB, tileSumsValid, (unsigned *)prefA, (unsigned *)prefB,
                                                 posE, posS, classBase, m, hits, nextInfo, st, share, chainFlags, epoch, info, chainDraws);
  return launch_finalize_t<u64>(histA, histB, histLen, tileA, tileB, tileSumsValid, prefA, prefB, posE, posS, classBase, m, hits, nextInfo, st, share, chainFlags, epoch,
                                info, chainDraws);
}

hipError_t launch_coverage(const void *reads, const void *weights, i64 n, const CoverArgs &a, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const i64 waves = a.sched.nWaves;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (weights) coverage_walk_kernel_weighted<<<grid, 256, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
  else coverage_walk_kernel<false><<<grid, 256, 0, st>>>((const Tri *)reads, (const int *)weights, n, a);
  return hipGetLastError();
}

hipError_t launch_coverage_finalize(const CoverArgs &a, i64 histLen, const CoverGather &g, i64 m, u64 *cov, DevInfo *nextInfo, hipStream_t st)
{
  const int nb = scan_tiles(histLen);
  for (int q = 0; q < 4 && nb > 0; q += 2)
    finalize_scan_kernel<u64><<<dim3(nb, 2), 256, 0, st>>>(a.hist[q], a.hist[q + 1], histLen, a.part[q], a.part[q + 1], g.pref[q], g.pref[q + 1], nullptr);
  const i64 work = (m > nb ? m : nb) > 0 ? (m > nb ? m : nb) : 1;
  gather_coverage_kernel<<<(unsigned)((work + 255) / 256), 256, 0, st>>>(g, m, cov, nb, nextInfo);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void scan_zero_kernel(uint4 *__restrict__ p, i64 n16, const int *__restrict__ runIf)
{
  if (runIf && *runIf == 0) return;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (i64)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}

hipError_t launch_scan_zero(void *micro, i64 bytes, const int *runIf, hipStream_t st)
{
  const i64 n16 = (bytes + 15) / 16;                       // (the buffer is allocated with room for the round-up)
  if (n16 <= 0) return hipSuccess;
  i64 blocks = (n16 + 255) / 256; if (blocks > 4096) blocks = 4096;
  scan_zero_kernel<<<(unsigned)blocks, 256, 0, st>>>((uint4 *)micro, n16, runIf);
  return hipGetLastError();
}

hipError_t launch_scan_hist(const void *reads, const void *weights, i64 n, const ScanArgs &a, hipStream_t st, const int *runIf)
{
  if (n <= 0) return hipSuccess;
  // >= 4 steps per wave; unweighted <= 32 blocks per CU (four rounds of resident waves: the launch's tail is a quarter of a round of
  // short spans; 100 M reads, same-box sweeps: 8 per CU 0.219 ms, 16: 0.205-0.211, 24-48: 0.200-0.207, 64: 0.204 -- 1-3 %), with
  // label weights <= 16 (64-bit counters, twice the tile flushes per span: 0.324-0.329 at 16 against 0.334-0.341 at 32)
  static const int perCuEnv = getenv("GTX_SCAN_BLOCKS_PER_CU") ? atoi(getenv("GTX_SCAN_BLOCKS_PER_CU")) : 0;
  const int perCu = perCuEnv > 0 ? perCuEnv : (weights ? 16 : 32);
  i64 blocks = (n + 4095) / 4096; if (blocks > 256 * (i64)perCu) blocks = 256 * (i64)perCu;
  if (weights) scan_hist_kernel<true><<<(unsigned)blocks, 256, 0, st>>>((const Tri *)reads, (const int *)weights, n, a, runIf);
  else scan_hist_kernel<false><<<(unsigned)blocks, 256, 0, st>>>((const Tri *)reads, (const int *)weights, n, a, runIf);
  return hipGetLastError();
}

int scan_window_tile() { return kWinTile; }

hipError_t launch_scan_windows(const void *micro, bool micro64, const ScanArgs &a, i64 totalTiles, u64 *out, hipStream_t st, const int *runIf)
{
  if (totalTiles <= 0) return hipSuccess;
  if (micro64) scan_window_kernel<u64><<<(unsigned)totalTiles, 256, 0, st>>>((const u64 *)micro, a, out, runIf);
  else scan_window_kernel<unsigned><<<(unsigned)totalTiles, 256, 0, st>>>((const unsigned *)micro, a, out, runIf);
  return hipGetLastError();
}

} // namespace gtx
