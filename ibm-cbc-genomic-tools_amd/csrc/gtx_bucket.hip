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
// The partition is ONE pass over the triples and takes no global atomic (round 1 and the first half of round 2 read the
// triples twice -- a histogram pass to size the buckets' regions -- and reserved room per (tile, bucket) with a returning
// atomic on a per-bucket counter: 25 k tiles adding to one address is a chain of ~50 ns steps, which bounded the pass):
//
//   bucket_scatter_kernel  blocks stay and take tiles of 4096 reads in turn (the next tile's loads in flight meanwhile).
//                          Every block owns an ARENA of the scratch array -- as many pairs as it will see reads, plus a
//                          chunk per bucket -- and deals it out in chunks of 64 (start, end) pairs, a current chunk per
//                          bucket.  Per tile: bucket of every read (direct-address table in LDS, no branches), rank inside
//                          (tile, bucket) from the return value of an LDS counter's atomic, one LDS-only step per bucket
//                          (room left in its chunk, new chunks from the arena when that does not do; wave prefix sums),
//                          the tile regrouped by bucket in LDS and copied out so that consecutive lanes store consecutive
//                          pairs.  A chunk is filled completely before the bucket gets another, so the arena cannot run
//                          out and nothing is estimated.  Also counts the reads of unknown class / start > end for
//                          gtx_count_info and sets inverted reads aside (CountArgs::side) -- on a branch of its own.
//   chunk_rows_kernel, chunk_offsets_kernel   exclusive prefixes over the (bucket, block) matrix of chunk counts: where each
//                          block's chunks of each bucket go in the bucket-major chunk list
//   chunk_place_kernel     one block per arena: its directory entries (chunk -> bucket, fill) into the list
//   bucket_count_kernel    grid = buckets x splits: the bucket's slices of both boundary arrays in LDS with a
//                          direct-address table over their value range (cell -> the boundaries of the cell), so a rank is one
//                          table read + a search among the few boundaries of one cell instead of a 12-probe binary
//                          search whose probes of different lanes fall on one LDS bank; a wave takes a chunk per step (four in
//                          flight, their ranks step by step together), two LDS atomics per read, contiguous atomic flush.
//                          A read whose end lies beyond the bucket's slice of the starts array (longer than the bucket is
//                          wide) falls back to a global search + atomic for histogram B.
// Traffic per read: 12 B read, 8 B (start, end) written and read once: 2.8 GB for 100 M reads (+ 8 B per chunk of directory and
// list).  What bounds the scatter pass is not that traffic but instruction issue and LDS round trips between four barriers per
// tile (cycle counters per phase: DESIGN.md 4.2); storing the pairs straight from the loading threads' registers instead of
// regrouping costs a 64-byte L2 request per pair and comes out the same.
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

static constexpr int kChunk = 64, kChunkShift = 6;                // pairs per chunk: a wave's load in the counting kernel, 512 B

__device__ __forceinline__ Tri3 load_tri3(const Tri3 *p)
{
  const int *q = (const int *)p;
  Tri3 t;
  t.c = __builtin_nontemporal_load(q); t.s = __builtin_nontemporal_load(q + 1); t.e = __builtin_nontemporal_load(q + 2);
  return t;
}

// Bucket of a read through the direct-address table (BucketTable::clsCell / cellTab, copies in LDS): the class's entry, the cell
// of s, the first bucket that can hold a position of that cell, then upwards while s lies above the bucket (the cuts are wider
// than a cell on average: ~1 comparison).  No branches on the way: a read that is not counted (unknown class, start > end, beyond
// the input) and a class without reference regions go through a NULL entry -- class index nClasses, cell nCells, bucket nB, which
// takes any position -- and N reads of a thread go step by step together: three LDS round trips for all of them.
// LDS copies: clsCell[c] = {first cell, lowest cut, cells - 1, first bucket}, [nClasses] = null; tab[nCells] = 0; posHi[nB] = INT_MAX.
template <int N>
__device__ __forceinline__ void bucket_lookup(const int4 *__restrict__ clsCell, const unsigned short *__restrict__ tab, const int *__restrict__ posHi, int sh,
                                              int nullClass, const bool (&want)[N], const int (&c)[N], const int (&s)[N], int (&id)[N])
{
  int4 cc[N]; int ph[N];
#pragma unroll
  for (int k = 0; k < N; k++) cc[k] = clsCell[want[k] ? c[k] : nullClass];
#pragma unroll
  for (int k = 0; k < N; k++) {
    unsigned cell = ((unsigned)s[k] - (unsigned)cc[k].y) >> sh;
    cell = cell < (unsigned)cc[k].z ? cell : (unsigned)cc[k].z;
    cell = s[k] > cc[k].y ? cell : 0u;
    id[k] = cc[k].w + tab[cc[k].x + cell];
  }
#pragma unroll
  for (int k = 0; k < N; k++) ph[k] = posHi[id[k]];
#pragma unroll
  for (int k = 0; k < N; k++) id[k] += s[k] > ph[k];                                      // the first step upwards for all of them together
#pragma unroll
  for (int k = 0; k < N; k++) ph[k] = posHi[id[k]];
#pragma unroll
  for (int k = 0; k < N; k++) while (s[k] > ph[k]) ph[k] = posHi[++id[k]];              // (two cuts inside one cell: rare.  The class's last bucket has posHi = INT_MAX)
}

// inclusive prefix sum over the 64 lanes by DPP (row shifts, then row broadcasts); call with all lanes active
__device__ __forceinline__ unsigned wave_prefix(unsigned x)
{
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return (unsigned)v;
}

// LDS: the lookup tables (clsCell, posHi, cellTab); per bucket {reads of this tile, next free place of its chunk, chunks so far}
// and where this tile's reads go {A = place of rank 0 (room = pairs left in that chunk = -A mod 64), B + rank = place of the
// ranks beyond, L = first place in the staged tile}; the tile regrouped by bucket: (start, end, place in the scratch array,
// weight) per read, an array each -- neighbours in a bucket are neighbours in LDS and in the scratch array, so the copy out of LDS stores runs of
// pairs (one 64-byte request per ~8 pairs; storing the pairs straight from the registers of the threads that loaded them is a
// request per pair: no faster when a tile's pairs of a bucket still leave together, 1.8x slower when they leave one by one).
// (launch bounds: weighted tiles of 4096 reads are one block per CU by LDS anyway)
// LINE > 0: pairs leave for the scratch array only as whole, aligned groups of LINE pairs (64 or 32 bytes).  A (tile, bucket) group
// is ~8 pairs at an arbitrary offset of its chunk: written as it comes, most groups are two partial lines, and a line completed
// only tiles later has long left the L2 -- HBM then takes two masked writes for it (1.63x the bytes by WRITE_SIZE, and far more
// than that in time: with the pairs written to an L2-resident window instead the pass takes 0.34 ms, with the reads from one
// 0.45, with both from HBM 0.66).  So every bucket keeps its last < LINE pairs in LDS (`carry`): per tile it sends
// floor((carried + new) / LINE) * LINE pairs -- the carried ones first, picked up into registers before the tile is regrouped --
// and keeps the rest; what is left at the end goes out with the chunk's fill count.
template <bool WEIGHTED, int PER, int LINE>
__global__ __launch_bounds__(1024, ((WEIGHTED && PER == 4) || LINE) ? 4 : 8) void bucket_scatter_kernel(const Tri3 *__restrict__ reads, const int *__restrict__ weights, i64 n, CountArgs a, BucketTable t, BucketWork w)
{
  constexpr int TB = PER * 1024;
  extern __shared__ int4 lds4[];
  const int nB = t.nB;
  int4 *clsCell = lds4; uint4 *gb = (uint4 *)(clsCell + a.nClasses + 1); int2 *stage = (int2 *)(gb + nB + 1); unsigned *sat = (unsigned *)(stage + TB);
  int *sw = (int *)(sat + TB); int *posHi = sw + (WEIGHTED ? TB : 0); unsigned *cnt = (unsigned *)(posHi + nB + 1), *next = cnt + nB + 1, *nch = next + nB;
  unsigned *rcar = nch + nB;                                       // LINE: pairs a bucket carries (< LINE), the pairs, their weights
  int2 *carry = (int2 *)(rcar + (LINE ? nB + (nB & 1) : 0)); int *carryW = (int *)(carry + (LINE ? (size_t)nB * LINE : 0));
  unsigned short *tab = (unsigned short *)(carryW + ((LINE && WEIGHTED) ? (size_t)nB * LINE : 0));
  __shared__ unsigned arenaCur, tileCur[2];
  const unsigned arena0 = blockIdx.x * w.arenaPairs;               // this block's part of the scratch array
  const int4 nullEntry = make_int4(t.nCells, INT_MAX, 0, nB);       // (bucket_lookup)
  for (int i = threadIdx.x; i < nB; i += 1024) { posHi[i] = t.posHi[i]; cnt[i] = 0; next[i] = 0; nch[i] = 0; if (LINE) rcar[i] = 0; }   // next = 0: no chunk yet, no room
  for (int i = threadIdx.x; i < a.nClasses; i += 1024) { const int4 e = t.clsCell[i]; const bool none = e.z <= 0; clsCell[i] = make_int4(none ? nullEntry.x : e.x, none ? nullEntry.y : e.y, none ? 0 : e.z - 1, none ? nullEntry.w : e.w); }
  for (int i = threadIdx.x; i < t.nCells; i += 1024) tab[i] = t.cellTab[i];
  if (threadIdx.x == 0) { arenaCur = arena0; tileCur[0] = tileCur[1] = 0; clsCell[a.nClasses] = nullEntry; tab[t.nCells] = 0; posHi[nB] = INT_MAX; cnt[nB] = 0; }
  // (this path takes n < 2^31: 32-bit indices)
  const unsigned un = (unsigned)n, tiles = (unsigned)((n + TB - 1) / TB);
  Tri3 nx[PER]; int nw[PER];
  auto fetch = [&](unsigned tile) {
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const unsigned i = tile * TB + k * 1024 + threadIdx.x;
      nx[k].c = -1; nx[k].s = nx[k].e = 0; nw[k] = 1;
      if (tile < tiles && i < un) { nx[k] = load_tri3(reads + i); if (WEIGHTED) nw[k] = __builtin_nontemporal_load(weights + i); }
    }
  };
  fetch(blockIdx.x);
  __syncthreads();
  unsigned nNoClass = 0, nDegen = 0, firstDegen = 0xffffffffu;
  int2 *__restrict__ out = (int2 *)w.tmpReads;                    // (start, end): the class is the bucket's
  const int sh = t.cellShift, lane = threadIdx.x & 63;
  unsigned parity = 0;
  for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x, parity ^= 1) {
    int rc[PER], rs[PER], re[PER], rw[PER], id[PER]; unsigned rank[PER]; bool want[PER];
    bool other = false;                                             // a read of this thread that is not counted: unknown class or start > end
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const unsigned i = tile * TB + k * 1024 + threadIdx.x;
      const Tri3 r = nx[k];
      rc[k] = r.c; rs[k] = r.s; re[k] = r.e; rw[k] = nw[k]; rank[k] = 0;
      bool cls = (unsigned)r.c < (unsigned)a.nClasses;
      if (a.owned) cls = cls && a.owned[cls ? r.c : 0];             // (a group member: reads of other members' classes count as reads of no class)
      const bool in = i < un, ok = cls && r.s <= r.e + a.zeroLenOk;
      want[k] = in && ok; other |= in && !ok;
    }
    if (__builtin_amdgcn_ballot_w64(other)) {                        // (rare: kept out of the way of the common path, which is bound by instruction issue)
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const unsigned i = tile * TB + k * 1024 + threadIdx.x;
        if (i < un && !want[k]) {
          if ((unsigned)rc[k] >= (unsigned)a.nClasses || (a.owned && !a.owned[rc[k]])) nNoClass++;
          else if (!a.coverRule || rs[k] > re[k] + 1) {
            nDegen++; if (i < firstDegen) firstDegen = i;
            if (a.side) { const unsigned q = atomicAdd(a.sideCount, 1u); if (q < (unsigned)a.sideCap) a.side[q] = make_int4(rc[k], rs[k], re[k], rw[k]); }   // see CountArgs::side
          }
        }
      }
    }
    if (a.keyCenter) {                                              // (genomic_scans -op c: the read's centre decides)
      int key[PER];
#pragma unroll
      for (int k = 0; k < PER; k++) key[k] = (int)((i64)rs[k] + ((i64)re[k] - rs[k]) / 2);
      bucket_lookup<PER>(clsCell, tab, posHi, sh, a.nClasses, want, rc, key, id);
    } else bucket_lookup<PER>(clsCell, tab, posHi, sh, a.nClasses, want, rc, rs, id);
    fetch(tile + gridDim.x);                                        // in flight until the top of the next round
#pragma unroll
    for (int k = 0; k < PER; k++) rank[k] = atomicAdd(&cnt[id[k]], 1u);   // its rank among the tile's reads of that bucket (bucket nB: the reads that are not counted)
    __syncthreads();
    // per bucket present: room in its chunk or new chunks from the arena, and its stretch of the staged tile
    if (threadIdx.x == 0) { tileCur[parity ^ 1] = 0; cnt[nB] = 0; }
    for (int b0 = 0; b0 < nB; b0 += 1024) {
      if (b0 + (int)(threadIdx.x & ~63u) >= nB) continue;          // (a whole wave without buckets: 7 of 16 at 540 buckets)
      const int b = b0 + threadIdx.x;
      const unsigned c = b < nB ? cnt[b] : 0u;
      const unsigned A = b < nB ? next[b] : 0u, room = (0u - A) & (kChunk - 1);
      // LINE: r pairs carried, F = what leaves now (whole lines: the carried pairs, then the first F - r new ones)
      const unsigned r = (LINE && b < nB) ? rcar[b] : 0u, F = LINE ? (c ? (r + c) & ~(unsigned)(LINE ? LINE - 1 : 0) : 0u) : c;
      // the chunk fills up: as many new ones as the rest needs, side by side.  Both running sums (places in the staged tile, pairs
      // of the arena) by wave prefix and one LDS atomic per wave
      const unsigned need = F > room ? F - room : 0u, fresh = (need + kChunk - 1) & ~(unsigned)(kChunk - 1);
      const unsigned incL = wave_prefix(c), incF = wave_prefix(fresh);
      const unsigned totL = (unsigned)__builtin_amdgcn_readlane((int)incL, 63), totF = (unsigned)__builtin_amdgcn_readlane((int)incF, 63);
      unsigned baseL = 0, baseF = 0;
      if (totL) {                                                     // (wave-uniform; the sums are scalars: one lane adds them)
        if (lane == 0) { baseL = atomicAdd(&tileCur[parity], totL); if (totF) baseF = atomicAdd(&arenaCur, totF); }
        baseL = (unsigned)__builtin_amdgcn_readfirstlane((int)baseL); baseF = (unsigned)__builtin_amdgcn_readfirstlane((int)baseF);
      }
      if (LINE && b < nB && !c) gb[b].w = 0u;                        // (nothing leaves: the carried pairs stay where they are)
      if (c) {
        const unsigned at = baseF + incF - fresh;
        cnt[b] = 0; next[b] = need ? at + need : A + F;
        if (LINE) rcar[b] = r + c - F;
        gb[b] = make_uint4(A, at - room, baseL + incL - c, LINE ? (r | (F << 4)) : 0u);
        if (need) {
          nch[b] += fresh >> kChunkShift;
          for (unsigned j = 0; j < fresh >> kChunkShift; j++) w.dir[(at >> kChunkShift) + j] = (unsigned)b | ((unsigned)kChunk << 16);   // full, unless it stays the bucket's last (below)
        }
      }
    }
    __syncthreads();
    constexpr int CARRY_ROUNDS = LINE ? 9 : 1;                      // (nB * LINE <= 9 * 1024: launch_partition)
    int2 cv[CARRY_ROUNDS]; unsigned cat[CARRY_ROUNDS]; int cw[CARRY_ROUNDS];
    {
      uint4 g[PER];
#pragma unroll
      for (int k = 0; k < PER; k++) g[k] = gb[id[k]];               // (entry nB: whatever; not used)
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const unsigned room = (0u - g[k].x) & (kChunk - 1), p = g[k].z + rank[k];
        unsigned at;
        if (LINE) {
          const unsigned q = (g[k].w & 15u) + rank[k], F = g[k].w >> 4;        // its place among the bucket's pairs of this round
          at = q < F ? (q < room ? g[k].x : g[k].y) + q : 0x80000000u | ((unsigned)id[k] * LINE + (q - F));
        } else at = (rank[k] < room ? g[k].x : g[k].y) + rank[k];
        if (id[k] != nB) {
          stage[p] = make_int2(rs[k], re[k]); sat[p] = at;
          if (WEIGHTED) sw[p] = rw[k];
        }
      }
      if (LINE) {
        // the carried pairs of the buckets that send lines now: into registers (their places in `carry` are taken over below)
#pragma unroll
        for (int i = 0; i < CARRY_ROUNDS; i++) {
          const unsigned idx = threadIdx.x + 1024u * i, b = idx / (LINE ? LINE : 1), slot = idx % (LINE ? LINE : 1);
          cat[i] = 0xffffffffu;
          if (b < (unsigned)nB) {
            const uint4 gg = gb[b];
            const unsigned room = (0u - gg.x) & (kChunk - 1);
            if ((gg.w >> 4) && slot < (gg.w & 15u)) {
              cat[i] = (slot < room ? gg.x : gg.y) + slot; cv[i] = carry[idx];
              if (WEIGHTED) cw[i] = carryW[idx];
            }
          }
        }
      }
    }
    __syncthreads();
    const unsigned total = tileCur[parity];
    for (unsigned j = threadIdx.x; j < total; j += 1024) {
      const unsigned at = sat[j];
      if (LINE && (at & 0x80000000u)) {
        carry[at & 0x7fffffffu] = stage[j];
        if (WEIGHTED) carryW[at & 0x7fffffffu] = sw[j];
        continue;
      }
      out[at] = stage[j];
      if (WEIGHTED) w.tmpWeights[at] = sw[j];
    }
    if (LINE) {
#pragma unroll
      for (int i = 0; i < CARRY_ROUNDS; i++)
        if (cat[i] != 0xffffffffu) { out[cat[i]] = cv[i]; if (WEIGHTED) w.tmpWeights[cat[i]] = cw[i]; }
    }
    // (the next round's writes to cnt and tileCur come behind barriers every wave reaches after this copy; those to stage and gb too)
  }

  __syncthreads();
  for (int b = threadIdx.x; b < nB; b += 1024) {
    unsigned e = next[b];
    if (LINE) {
      const unsigned r = rcar[b];                                   // what the bucket still carries: behind its last line
      if (r) {
        if (!(e & (kChunk - 1))) { e = atomicAdd(&arenaCur, (unsigned)kChunk); nch[b]++; }      // (the arena has a chunk per bucket beyond its reads)
        for (unsigned k = 0; k < r; k++) { out[e + k] = carry[(unsigned)b * LINE + k]; if (WEIGHTED) w.tmpWeights[e + k] = carryW[(unsigned)b * LINE + k]; }
        e += r;
      }
    }
    if (e & (kChunk - 1)) w.dir[e >> kChunkShift] = (unsigned)b | ((e & (kChunk - 1)) << 16);   // the bucket's last chunk is part full
    w.chunkCount[(size_t)b * gridDim.x + blockIdx.x] = nch[b];
  }
  if (LINE) __syncthreads();
  if (threadIdx.x == 0) w.arenaUsed[blockIdx.x] = (arenaCur - arena0) >> kChunkShift;
  if (nNoClass) atomicAdd((u64 *)&a.info->n_no_class, (u64)nNoClass);
  if (nDegen) { atomicAdd((u64 *)&a.info->n_degenerate, (u64)nDegen); atomicMin((i64 *)&a.info->first_degenerate, (i64)firstDegen + a.indexBase); }
}

// chunkCount[b * G + k] (chunks of bucket b in the arena of block k): exclusive prefix inside the row of every bucket (one block
// per bucket), then over the buckets' totals (one block): bucket b's chunks of arena k start at rowOff[b] + chunkCount[b * G + k]
__global__ __launch_bounds__(256) void chunk_rows_kernel(BucketWork w, unsigned nArenas)
{
  __shared__ unsigned wsum[4];
  unsigned *row = w.chunkCount + (size_t)blockIdx.x * nArenas;
  const unsigned per = (nArenas + 255) / 256, i0 = threadIdx.x * per;
  unsigned s = 0;
  for (unsigned k = 0; k < per; k++) if (i0 + k < nArenas) s += row[i0 + k];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  unsigned run = inc - s;
  for (int k = 0; k < wv; k++) run += wsum[k];
  for (unsigned k = 0; k < per; k++) if (i0 + k < nArenas) { const unsigned v = row[i0 + k]; row[i0 + k] = run; run += v; }
  if (threadIdx.x == 255) w.rowOff[blockIdx.x] = run;             // the bucket's total, turned into its offset by the next kernel
}

__global__ __launch_bounds__(1024) void chunk_offsets_kernel(BucketWork w, int nB)
{
  __shared__ unsigned wsum[16];
  const int per = (nB + 1023) / 1024, i0 = threadIdx.x * per;
  unsigned s = 0;
  for (int k = 0; k < per; k++) if (i0 + k < nB) s += w.rowOff[i0 + k];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  unsigned run = inc - s;
  for (int k = 0; k < wv; k++) run += wsum[k];
  for (int k = 0; k < per; k++) if (i0 + k < nB) { const unsigned v = w.rowOff[i0 + k]; w.rowOff[i0 + k] = run; run += v; }
  if (threadIdx.x == 1023) w.rowOff[nB] = run;
}

// block k: the chunks of its arena into the bucket-major list: entry = chunk << 6 | (fill - 1)
__global__ __launch_bounds__(1024) void chunk_place_kernel(BucketTable t, BucketWork w, unsigned nArenas)
{
  extern __shared__ unsigned seen[];
  for (int i = threadIdx.x; i < t.nB; i += blockDim.x) seen[i] = 0;
  __syncthreads();
  const unsigned k = blockIdx.x, first = k * (w.arenaPairs >> kChunkShift), used = w.arenaUsed[k];
  constexpr int U = 4;                                             // entries per thread and round: their loads together
  for (unsigned j0 = threadIdx.x; j0 < used; j0 += U * blockDim.x) {
    unsigned d[U], at[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const unsigned j = j0 + u * blockDim.x; d[u] = j < used ? w.dir[first + j] : 0u; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned j = j0 + u * blockDim.x, b = d[u] & 0xffffu;
      at[u] = 0;
      if (j < used) at[u] = w.rowOff[b] + w.chunkCount[(size_t)b * nArenas + k] + atomicAdd(&seen[b], 1u);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned j = j0 + u * blockDim.x;
      if (j < used) w.list[at[u]] = ((first + j) << kChunkShift) | ((d[u] >> 16) - 1);
    }
  }
}

static constexpr int kBktE = 2048, kBktS = 4096;                 // boundaries per bucket (ends array) / starts-array slice in LDS
static constexpr int kCellsE = 2048, kCellsS = 4096;             // cells of the direct-address tables
static constexpr int kBktT = 2560, kCellsT = 2560;               // coverage: slice of the threshold array in LDS (the bucket's 2048 + what a read's end can lie beyond them) / its cells

// ranks of U keys among the sorted boundaries v[0..n): #{v < key} (LE = false) or #{v <= key} (true), through the table
// tab[c] = i0 | i1 << 16: the boundaries whose value lies in cell c are v[i0..i1) (cell(x) = (x - lo) >> sh, clamped to [0, cells)):
// one table read bounds the search to the boundaries of one cell (~1).  The U keys of a thread go step by step together.
template <bool LE, int U>
__device__ __forceinline__ void table_ranks(const int *__restrict__ v, const unsigned *__restrict__ tab, int cells, int lo, int sh, const int (&key)[U], int (&rank)[U])
{
  unsigned t[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    unsigned c = ((unsigned)key[u] - (unsigned)lo) >> sh;
    c = c < (unsigned)cells - 1 ? c : (unsigned)cells - 1;
    t[u] = tab[key[u] > lo ? c : 0u];
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    int a = (int)(t[u] & 0xffffu), b = (int)(t[u] >> 16);
    while (a < b) { const int mid = (a + b) >> 1; if (LE ? v[mid] <= key[u] : v[mid] < key[u]) a = mid + 1; else b = mid; }
    rank[u] = a;
  }
}

// tab[c] for c in [0, cells): low half = #{v_i : cell(v_i) < c}, then the high half = the next cell's low half (n for the last)
__device__ __forceinline__ void build_table(const int *__restrict__ v, int n, unsigned *__restrict__ tab, int cells, int lo, int sh)
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
    tab[c] = (unsigned)a;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cells; c += blockDim.x) tab[c] |= (tab[c + 1] & 0xffffu) << 16;   // (a neighbour's update leaves the low half as it is)
}

__device__ __forceinline__ int shift_for(int lo, int hi, int cells)
{
  const u64 span = (u64)((i64)hi - lo);
  int sh = 0;
  while ((span >> sh) >= (u64)cells) sh++;
  return sh;
}

// (1024 threads, a chunk of <= 64 reads per wave and step, 4 in flight: the loop is a chain global load -> table -> search ->
// atomic, and what bounds the kernel is how many of those chains a CU has open)
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_count_kernel(CountArgs a, BucketTable t, BucketWork w, int splits)
{
  __shared__ int sE[kBktE], sS[kBktS];
  __shared__ unsigned tE[kCellsE + 1], tS[kCellsS + 1];
  typedef typename std::conditional<WEIGHTED, u64, unsigned>::type hist_t;   // a block sees < 2^32 reads
  __shared__ hist_t hA[kBktE + 1], hB[kBktS + 1];
  const int b = blockIdx.x / splits, k = blockIdx.x % splits;
  const unsigned c0 = w.rowOff[b], nCh = w.rowOff[b + 1] - c0;   // the bucket's stretch of the chunk list
  const unsigned r0 = c0 + (unsigned)((u64)nCh * k / splits), r1 = c0 + (unsigned)((u64)nCh * (k + 1) / splits);
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
  constexpr int U = 4;
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nW = blockDim.x >> 6;
  unsigned ent[U];                                                 // list entries of the round: loaded a round ahead
  auto entries = [&](unsigned at) {
#pragma unroll
    for (int u = 0; u < U; u++) { const unsigned ci = at + u * nW; ent[u] = ci < r1 ? w.list[ci] : 0u; }
  };
  entries(r0 + wv);
  for (unsigned at = r0 + wv; at < r1; at += U * nW) {
    int2 se[U]; int wt4[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = ent[u], place = (e >> kChunkShift << kChunkShift) + lane;
      se[u] = make_int2(0, -1); wt4[u] = 1;
      on[u] = at + u * nW < r1 && lane <= (e & (kChunk - 1));
      if (on[u]) { se[u] = reads[place]; if (WEIGHTED) wt4[u] = w.tmpWeights[place]; }
    }
    entries(at + U * nW);
    int ks[U], ke[U], slotA[U], slotB[U];
#pragma unroll
    for (int u = 0; u < U; u++) { ks[u] = se[u].x; ke[u] = se[u].y; }
    table_ranks<false, U>(sE, tE, kCellsE, loE, shE, ks, slotA);      // #{E < s} inside the slice
    table_ranks<true, U>(sS, tS, kCellsS, loS, shS, ke, slotB);       // #{S <= e} inside the slice
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 wt = WEIGHTED ? (u64)(i64)wt4[u] : 1;
      if (on[u]) atomicAdd(&hA[slotA[u]], (hist_t)wt);
      if (on[u] && (slotB[u] < nS || sHi == segEnd)) atomicAdd(&hB[slotB[u]], (hist_t)wt);
      else if (on[u]) {                                             // the read ends beyond the slice: global search above it
        int glo = sHi, ghi = segEnd;
        while (glo < ghi) { const int mid = (int)(((i64)glo + ghi) >> 1); if (a.sortedS[mid] <= ke[u]) glo = mid + 1; else ghi = mid; }
        atomicAdd(&a.histB[(i64)glo + cls], wt);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= nE; i += blockDim.x) { const u64 v = hA[i]; if (v) atomicAdd(&a.histA[(i64)eLo + i + cls], v); }
  for (int i = threadIdx.x; i <= nS; i += blockDim.x) { const u64 v = hB[i]; if (v) atomicAdd(&a.histB[(i64)sLo + i + cls], v); }
}

// Coverage (CoverArgs: one threshold array, four histograms; a key x belongs to the slots at or below threshold T iff x <= T): the
// bucket's cuts and its slice are both in sortedT, the slice beginning where the bucket does (a read ends at or behind its start).
// Per read four LDS atomics: (w, w x start) at the rank of the start, (w, w x end) at the rank of the end; an end beyond the slice
// goes to the global histograms.
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_cover_kernel(CoverArgs cv, BucketTable t, BucketWork w, int splits)
{
  // unweighted reads count in 32 bits (a block sees < 2^31 reads; the key sums need 64): 74 KB, two blocks per CU (weighted: 92 KB, one)
  typedef typename std::conditional<WEIGHTED, u64, unsigned>::type cnt_t;
  __shared__ int sT[kBktT];
  __shared__ unsigned tT[kCellsT + 1];
  __shared__ cnt_t hWs[kBktE + 1], hWe[kBktT + 1];
  __shared__ u64 hFs[kBktE + 1], hFe[kBktT + 1];
  const int b = blockIdx.x / splits, k = blockIdx.x % splits;
  const unsigned c0 = w.rowOff[b], nCh = w.rowOff[b + 1] - c0;   // the bucket's stretch of the chunk list
  const unsigned r0 = c0 + (unsigned)((u64)nCh * k / splits), r1 = c0 + (unsigned)((u64)nCh * (k + 1) / splits);
  if (r0 == r1) return;
  const int sLo = t.sLo[b], sHi = t.sHi[b], nS = sHi - sLo, nE = t.eHi[b] - t.eLo[b], cls = t.cls[b];   // (t.eLo[b] == sLo)
  const int segEnd = cv.segStartT[cls + 1];
  for (int i = threadIdx.x; i < nS; i += blockDim.x) sT[i] = cv.sortedT[sLo + i];
  for (int i = threadIdx.x; i <= nE; i += blockDim.x) { hWs[i] = 0; hFs[i] = 0; }
  for (int i = threadIdx.x; i <= nS; i += blockDim.x) { hWe[i] = 0; hFe[i] = 0; }
  __syncthreads();
  const int loT = nS ? sT[0] : 0, shT = nS ? shift_for(loT, sT[nS - 1], kCellsT) : 0;
  build_table(sT, nS, tT, kCellsT, loT, shT);
  __syncthreads();
  const int2 *__restrict__ reads = (const int2 *)w.tmpReads;
  constexpr int U = 4;
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nW = blockDim.x >> 6;
  unsigned ent[U];
  auto entries = [&](unsigned at) {
#pragma unroll
    for (int u = 0; u < U; u++) { const unsigned ci = at + u * nW; ent[u] = ci < r1 ? w.list[ci] : 0u; }
  };
  entries(r0 + wv);
  for (unsigned at = r0 + wv; at < r1; at += U * nW) {
    int2 se[U]; int wt4[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = ent[u], place = (e >> kChunkShift << kChunkShift) + lane;
      se[u] = make_int2(0, -1); wt4[u] = 1;
      on[u] = at + u * nW < r1 && lane <= (e & (kChunk - 1));
      if (on[u]) { se[u] = reads[place]; if (WEIGHTED) wt4[u] = w.tmpWeights[place]; }
    }
    entries(at + U * nW);
    int ks[U], ke[U], slotS[U], slotE[U];
#pragma unroll
    for (int u = 0; u < U; u++) { ks[u] = se[u].x; ke[u] = se[u].y; }
    table_ranks<false, U>(sT, tT, kCellsT, loT, shT, ks, slotS);      // #{T < s} inside the slice
    table_ranks<false, U>(sT, tT, kCellsT, loT, shT, ke, slotE);      // #{T < e}
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!on[u]) continue;
      const u64 wt = WEIGHTED ? (u64)(i64)wt4[u] : 1;
      atomicAdd(&hWs[slotS[u]], (cnt_t)wt); atomicAdd(&hFs[slotS[u]], wt * (u64)(i64)ks[u]);
      if (slotE[u] < nS || sHi == segEnd) { atomicAdd(&hWe[slotE[u]], (cnt_t)wt); atomicAdd(&hFe[slotE[u]], wt * (u64)(i64)ke[u]); }
      else {                                                          // the read ends beyond the slice: global search above it
        int glo = sHi, ghi = segEnd;
        while (glo < ghi) { const int mid = (int)(((i64)glo + ghi) >> 1); if (cv.sortedT[mid] < ke[u]) glo = mid + 1; else ghi = mid; }
        atomicAdd(&cv.hist[2][(i64)glo + cls], wt); atomicAdd(&cv.hist[3][(i64)glo + cls], wt * (u64)(i64)ke[u]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= nE; i += blockDim.x) {
    const u64 a0 = (u64)(WEIGHTED ? (i64)hWs[i] : (i64)(u64)hWs[i]), a1 = hFs[i];
    if (a0) atomicAdd(&cv.hist[0][(i64)sLo + i + cls], a0);
    if (a1) atomicAdd(&cv.hist[1][(i64)sLo + i + cls], a1);
  }
  for (int i = threadIdx.x; i <= nS; i += blockDim.x) {
    const u64 a0 = (u64)(WEIGHTED ? (i64)hWe[i] : (i64)(u64)hWe[i]), a1 = hFe[i];
    if (a0) atomicAdd(&cv.hist[2][(i64)sLo + i + cls], a0);
    if (a1) atomicAdd(&cv.hist[3][(i64)sLo + i + cls], a1);
  }
}

// genomic_scans counts (unsorted rule; start positions or centres): one part = some consecutive micro-windows of a bucket; the block reads ALL
// chunks of the bucket (they come from the L2 for the second and later parts of a bucket) and counts the reads of its own
// micro-windows in LDS -- no bucket is wider than a few parts.  (pos - 1) / step as in scan_slot (gtx_kernels.hip).
static constexpr int kScanBins32 = 30720, kScanBins64 = 15360;   // 120 KB of LDS counters

template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void bucket_scanhist_kernel(ScanArgs sc, BucketTable t, BucketWork w, const ScanPart *__restrict__ parts, u64 *__restrict__ out)
{
  typedef typename std::conditional<WEIGHTED, u64, unsigned>::type ct;
  extern __shared__ unsigned char ldsRaw[];
  ct *h = (ct *)ldsRaw;
  const ScanPart pt = parts[blockIdx.x];
  const int b = pt.bucket, cls = t.cls[b];
  const unsigned r0 = w.rowOff[b], r1 = w.rowOff[b + 1];
  if (r0 == r1 && !out) return;                                     // (out: the part's windows are written here, zeros included)
  for (int i = threadIdx.x; i < pt.count; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const i64 nm = sc.nMicro[cls];
  const unsigned d = (unsigned)sc.winStep;
  const int2 *__restrict__ reads = (const int2 *)w.tmpReads;
  constexpr int U = 4;
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nW = blockDim.x >> 6;
  unsigned ent[U];
  auto entries = [&](unsigned at) {
#pragma unroll
    for (int u = 0; u < U; u++) { const unsigned ci = at + u * nW; ent[u] = ci < r1 ? w.list[ci] : 0u; }
  };
  entries(r0 + wv);
  for (unsigned at = r0 + wv; at < r1; at += U * nW) {
    int2 se[U]; int wt4[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = ent[u], place = (e >> kChunkShift << kChunkShift) + lane;
      se[u] = make_int2(0, -1); wt4[u] = 1;
      on[u] = at + u * nW < r1 && lane <= (e & (kChunk - 1));
      if (on[u]) { se[u] = reads[place]; if (WEIGHTED) wt4[u] = w.tmpWeights[place]; }
    }
    entries(at + U * nW);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const i64 pos = sc.center ? (i64)se[u].x + ((i64)se[u].y - se[u].x) / 2 : (i64)se[u].x;
      if (!on[u] || pos < 1 || se[u].y <= 0) continue;               // the unsorted scanner's rule: start <= stop (the partition's), stop > 0, position >= 1
      const unsigned x = (unsigned)(pos - 1);
      unsigned q = d == 1 ? x : __umulhi(x, sc.winStepInv);
      unsigned r = x - q * d;
      if (r >= d) { q++; r -= d; }
      if (r >= d) q++;
      const i64 m = (i64)q - pt.first;
      if ((i64)q < nm && m >= 0 && m < pt.count) atomicAdd(&h[m], WEIGHTED ? (ct)(i64)wt4[u] : (ct)1);
    }
  }
  __syncthreads();
  if (out) {
    // The sliding sums straight from the part's histogram (no micro-window array in HBM, no window pass): window k = sum of the
    // micro-windows k .. k + comb - 1.  A window that lies inside the part is stored; one that reaches into the next part is a
    // partial sum here and gets the rest from there, both with atomics on a place scan_zero_edges_kernel has cleared -- the
    // windows that start less than comb - 1 before this part's first micro-window are the other half of that.
    const i64 nWin = sc.winOff[cls + 1] - sc.winOff[cls];
    u64 *__restrict__ o = out + sc.outOff[cls];
    const int comb = sc.comb;
    // the histogram becomes its own inclusive prefix sum (a thread takes `per` consecutive bins -- an odd number, so that the lanes of
    // a wave stay on different banks --, wave scan of the threads' totals, offsets added in a second turn): a window is then two reads
    __shared__ u64 wtot[16];
    int per = (pt.count + (int)blockDim.x - 1) / (int)blockDim.x; per |= 1;
    const int lo = min((int)threadIdx.x * per, pt.count), hi = min(lo + per, pt.count);
    ct run = 0;
    for (int x = lo; x < hi; x++) { run += h[x]; h[x] = run; }
    u64 inc = (u64)run;
#pragma unroll
    for (int d2 = 1; d2 < 64; d2 <<= 1) { const u64 up = __shfl_up(inc, d2); if ((int)lane >= d2) inc += up; }
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    u64 off = inc - (u64)run;
    for (unsigned k2 = 0; k2 < wv; k2++) off += wtot[k2];
    if (off) for (int x = lo; x < hi; x++) h[x] += (ct)off;
    __syncthreads();
    for (int i = threadIdx.x; i < pt.count; i += blockDim.x) {
      const i64 k = (i64)pt.first + i;
      if (k >= nWin) break;
      const int top = i + comb < pt.count ? i + comb : pt.count;
      const u64 s = (u64)h[top - 1] - (i ? (u64)h[i - 1] : 0ull);
      if (i + comb <= pt.count) o[k] = s;
      else if (s) atomicAdd(o + k, s);
    }
    for (int e = threadIdx.x; e < comb - 1; e += blockDim.x) {
      const i64 k = (i64)pt.first - (comb - 1) + e;                   // its micro-windows k .. k + comb - 1 end inside (or beyond) this part
      if (k < 0 || k >= nWin) continue;
      const i64 top = k + comb - pt.first < pt.count ? k + comb - pt.first : pt.count;
      const u64 s = top > 0 ? (u64)h[top - 1] : 0ull;
      if (s) atomicAdd(o + k, s);
    }
    return;
  }
  ct *dst = (ct *)sc.micro + sc.microOff[cls] + pt.first;
  for (int i = threadIdx.x; i < pt.count; i += blockDim.x) { const ct v = h[i]; if (v) atomicAdd(&dst[i], v); }
}

// the windows that take contributions from two parts (bucket_scanhist_kernel with `out`): cleared before the parts run
__global__ __launch_bounds__(64) void scan_zero_edges_kernel(ScanArgs sc, BucketTable t, const ScanPart *__restrict__ parts, u64 *__restrict__ out)
{
  const ScanPart pt = parts[blockIdx.x];
  if (pt.first == 0) return;
  const int cls = t.cls[pt.bucket];
  const i64 nWin = sc.winOff[cls + 1] - sc.winOff[cls];
  u64 *__restrict__ o = out + sc.outOff[cls];
  for (int e = threadIdx.x; e < sc.comb - 1; e += blockDim.x) {
    const i64 k = (i64)pt.first - (sc.comb - 1) + e;
    if (k >= 0 && k < nWin) o[k] = 0;
  }
}

int scan_part_bins(bool weighted) { return weighted ? kScanBins64 : kScanBins32; }
int scan_fused_max_comb() { return 64; }

int bucket_e_size() { return kBktE; }
int bucket_s_size() { return kBktS; }
int bucket_t_size() { return kBktT; }

static int device_cus()
{
  static int cus[64] = {0};                                       // per device (a race writes the same value twice)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus[dev] = n; }
  return cus[dev] > 0 ? cus[dev] : 256;
}

static size_t scatter_lds(int nClasses, int nB, int nCells, int per, bool weighted, int line = 0)
{
  return 16 * ((size_t)nClasses + 2) + 32 * (size_t)nB + 8 + (weighted ? 16 : 12) * 1024 * (size_t)per + 2 * ((size_t)nCells + 2) + 16 +
         (line ? (size_t)nB * (4 + (weighted ? 12 : 8) * (size_t)line) + 8 : 0);
}

// whether a reference set's tables fit the scatter kernel's LDS at its smallest tile (if not, the search kernel serves)
bool bucket_tables_fit(int nClasses, int nB, int nCells) { return nB <= 65535 && scatter_lds(nClasses, nB, nCells, 1, true) <= 150 * 1024; }

// The launch geometry of a call of n reads, and what it needs of scratch: blocks that stay (two per CU when the LDS tables
// allow), an arena per block (its tiles' reads + a chunk per bucket), a directory and a list entry per chunk, the (bucket, block)
// matrix of chunk counts.
BucketPlan bucket_plan(i64 n, int nClasses, int nB, int nCells, bool weighted)
{
  BucketPlan p;
  static const int perCu = getenv("GTX_SPLIT_BLOCKS_PER_CU") ? atoi(getenv("GTX_SPLIT_BLOCKS_PER_CU")) : 0;
  static const int tbMax = getenv("GTX_SPLIT_TILE") ? atoi(getenv("GTX_SPLIT_TILE")) : 4096;
  p.per = tbMax >= 4096 ? 4 : tbMax >= 2048 ? 2 : 1;
  while (p.per > 1 && scatter_lds(nClasses, nB, nCells, p.per, weighted) > 150 * 1024) p.per >>= 1;     // (gtx_set_refs keeps nB and the classes within what per = 1 takes)
  // whole lines only (LINE of bucket_scatter_kernel): 8 pairs per bucket carried in LDS where that fits next to the tile, else 4.
  // (100 M reads x 1 M regions, scatter pass alone: as they come 0.68 ms, WRITE_SIZE 1.28 GB for 0.80 GB of pairs; lines of 4: 0.61 ms,
  //  1.24 GB; of 8: 0.50-0.51 ms, 0.99 GB; of 16 -- GTX_SPLIT_LINE=16 -- 0.54 ms, 0.79 GB = 1.00x: nearly every pair then waits in LDS
  //  once, and the pass is bound by its instructions again)
  static const int lineEnv = getenv("GTX_SPLIT_LINE") ? atoi(getenv("GTX_SPLIT_LINE")) : -1;
  p.line = 0;
  for (int line : {16, 8, 4})
    if (!p.line && (lineEnv < 0 ? line != 16 : lineEnv == line) && (size_t)nB * line <= 9 * 1024 && scatter_lds(nClasses, nB, nCells, p.per, weighted, line) <= 150 * 1024) p.line = line;
  const i64 tb = (i64)p.per * 1024, tiles = (n + tb - 1) / tb;
  const size_t lds = scatter_lds(nClasses, nB, nCells, p.per, weighted, p.line);
  const int fit = perCu > 0 ? perCu : ((!p.line && 2 * (lds + 1024) <= 160 * 1024) ? 2 : 1);      // (the LINE kernels take more than 64 registers: one block of 16 waves per CU)
  const i64 most = (i64)fit * device_cus();
  p.blocks = (unsigned)(tiles < most ? (tiles > 0 ? tiles : 1) : most);
  const i64 tilesPerBlock = (tiles + p.blocks - 1) / p.blocks;
  p.arenaPairs = (size_t)tilesPerBlock * tb + (size_t)nB * kChunk;
  p.pairs = p.arenaPairs * p.blocks;
  if (p.pairs >= (1ull << 31)) p.line = 0;                       // (the kernel marks a pair that stays in LDS with the top bit of its place)
  p.chunks = p.pairs >> kChunkShift;
  p.matrix = (size_t)nB * p.blocks;
  return p;
}

// the partition: scatter pass, chunk list
static hipError_t launch_partition(const void *reads, const void *weights, i64 n, const CountArgs &a, const BucketTable &t, const BucketWork &w,
                                   const BucketPlan &p, hipStream_t st)
{
  static PerDevice attr;
  {
    hipError_t e = attr.once([] {
      hipError_t e = hipSuccess;
#define GTX_SCATTER_FNS(L) (const void *)bucket_scatter_kernel<false, 1, L>, (const void *)bucket_scatter_kernel<true, 1, L>, (const void *)bucket_scatter_kernel<false, 4, L>, \
                           (const void *)bucket_scatter_kernel<true, 4, L>, (const void *)bucket_scatter_kernel<false, 2, L>, (const void *)bucket_scatter_kernel<true, 2, L>
      const void *fn[] = {GTX_SCATTER_FNS(0), GTX_SCATTER_FNS(4), GTX_SCATTER_FNS(8), GTX_SCATTER_FNS(16)};
#undef GTX_SCATTER_FNS
      for (const void *f : fn) if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
      return e;
    });
    if (e != hipSuccess) return e;
  }
  const size_t lds = scatter_lds(a.nClasses, t.nB, t.nCells, p.per, weights != nullptr, p.line);
#define GTX_SCATTER(W, P, L) bucket_scatter_kernel<W, P, L><<<p.blocks, 1024, lds, st>>>((const Tri3 *)reads, (const int *)weights, n, a, t, w)
#define GTX_SCATTER_L(W, P) do { if (p.line == 16) GTX_SCATTER(W, P, 16); else if (p.line == 8) GTX_SCATTER(W, P, 8); else if (p.line == 4) GTX_SCATTER(W, P, 4); else GTX_SCATTER(W, P, 0); } while (0)
  if (weights) { if (p.per == 4) GTX_SCATTER_L(true, 4); else if (p.per == 2) GTX_SCATTER_L(true, 2); else GTX_SCATTER_L(true, 1); }
  else { if (p.per == 4) GTX_SCATTER_L(false, 4); else if (p.per == 2) GTX_SCATTER_L(false, 2); else GTX_SCATTER_L(false, 1); }
#undef GTX_SCATTER_L
#undef GTX_SCATTER
  chunk_rows_kernel<<<(unsigned)t.nB, 256, 0, st>>>(w, p.blocks);
  chunk_offsets_kernel<<<1, 1024, 0, st>>>(w, t.nB);     // (one more launch: the block of chunk_rows_kernel that finishes last doing it was slower, 540 blocks adding to one counter)
  chunk_place_kernel<<<p.blocks, 1024, sizeof(unsigned) * (size_t)t.nB, st>>>(t, w, p.blocks);
  return hipGetLastError();
}

static i64 count_splits(i64 n, int nB)
{
  // blocks of ~128k reads on average, at least one per bucket (100 M reads, whole path: 32 k 1.17 ms, 64 k 1.12, 128 k 1.085, 256 k 1.10:
  // the tables are built per block)
  static const i64 perBlock = getenv("GTX_COUNT_BLOCK_READS") ? atoll(getenv("GTX_COUNT_BLOCK_READS")) : 131072;
  i64 splits = (n + (i64)nB * perBlock - 1) / ((i64)nB * perBlock);
  if (splits < 1) splits = 1;
  if (splits > 512) splits = 512;
  return splits;
}

hipError_t launch_count_bucketed(const void *reads, const void *weights, i64 n, const CountArgs &a, const BucketTable &t, const BucketWork &w,
                                 const BucketPlan &p, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipError_t e = launch_partition(reads, weights, n, a, t, w, p, st);
  if (e != hipSuccess) return e;
  const i64 splits = count_splits(n, t.nB);
  if (weights) bucket_count_kernel<true><<<(unsigned)(t.nB * splits), 1024, 0, st>>>(a, t, w, (int)splits);
  else bucket_count_kernel<false><<<(unsigned)(t.nB * splits), 1024, 0, st>>>(a, t, w, (int)splits);
  return hipGetLastError();
}

hipError_t launch_scan_bucketed(const void *reads, const void *weights, i64 n, const CountArgs &a, const ScanArgs &sc, const BucketTable &t,
                                const BucketWork &w, const BucketPlan &p, const ScanPart *parts, int nParts, hipStream_t st, u64 *out)
{
  if (n <= 0 || nParts <= 0) return hipSuccess;
  if (out && sc.comb > 1) scan_zero_edges_kernel<<<(unsigned)nParts, 64, 0, st>>>(sc, t, parts, out);
  hipError_t e = launch_partition(reads, weights, n, a, t, w, p, st);
  if (e != hipSuccess) return e;
  static PerDevice attr;
  e = attr.once([] {
    hipError_t e2 = hipFuncSetAttribute((const void *)bucket_scanhist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void *)bucket_scanhist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return e2;
  });
  if (e != hipSuccess) return e;
  const size_t lds = (size_t)scan_part_bins(weights != nullptr) * (weights ? 8 : 4);
  if (weights) bucket_scanhist_kernel<true><<<(unsigned)nParts, 1024, lds, st>>>(sc, t, w, parts, out);
  else bucket_scanhist_kernel<false><<<(unsigned)nParts, 1024, lds, st>>>(sc, t, w, parts, out);
  return hipGetLastError();
}

hipError_t launch_cover_bucketed(const void *reads, const void *weights, i64 n, const CountArgs &a, const CoverArgs &cv, const BucketTable &t,
                                 const BucketWork &w, const BucketPlan &p, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipError_t e = launch_partition(reads, weights, n, a, t, w, p, st);
  if (e != hipSuccess) return e;
  const i64 splits = count_splits(n, t.nB);
  if (weights) bucket_cover_kernel<true><<<(unsigned)(t.nB * splits), 1024, 0, st>>>(cv, t, w, (int)splits);
  else bucket_cover_kernel<false><<<(unsigned)(t.nB * splits), 1024, 0, st>>>(cv, t, w, (int)splits);
  return hipGetLastError();
}

}  // namespace gtx
