// gtx_capi.hip -- implementation of the C ABI declared in include/gtx.h.
//
// Host side of the engine: builds the device-resident rank structure from the reference
// regions (replacing the bin-index construction of UnsortedGenomicRegionSetOverlaps,
// gtools/genomic_intervals.cpp:5593-5675), enqueues the streaming count / scan kernels of
// gtx_kernels.hip, and moves buffers.  There is deliberately no CPU implementation of the
// counting path in this library: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "gtx.h"
#include "gtx_kernels.h"
#include "gtx_pairs.h"
#include "gtx_text.h"
#include "gtx_internal.h"

typedef unsigned long long u64;

static thread_local std::string g_create_error;

struct gtx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // reference side
  int64_t nRefs = -1, nValid = 0;
  int nClasses = 0;
  int *d_sortedE = nullptr, *d_sortedS = nullptr, *d_segStart = nullptr;
  int *d_sampE = nullptr, *d_sampS = nullptr; int sampShift = 6, nSamp = 0;   // top level of the search kernel
  int *d_topE = nullptr, *d_topS = nullptr;   // every 256th boundary: first hop of the streaming kernel's start-of-span search
  // direct placement at the start of a span (gtx::PlaceTable): per class {segment start, end, first cell, cells}; per cell the rank
  // of the cell's first position in the ends array and in the starts array
  int4 *d_placeCls = nullptr; int *d_placeRank = nullptr; int placeShift = 0;
  int4 *d_placeClsT = nullptr; int *d_placeRankT = nullptr; int placeShiftT = 0;      // ... over the coverage thresholds (cover_prepare)
  std::vector<int32_t> h_seg;                        // [nClasses+1] class segments of the sorted boundary arrays (host copy)
  // a group member's share of the finalize step (gtxi_set_share): tiles of its classes, its regions in the group's compact order
  bool shareOn = false; int *d_shareTiles = nullptr; int nShareTiles = 0; int *d_shareRegions = nullptr; int64_t nShareRegions = 0, shareOffset = 0; unsigned char *d_shareOwned = nullptr;
  // unsorted reads, bucket path (gtx_bucket.hip): table built with the references, scratch sized by the largest call
  void *d_clsCell = nullptr, *d_cellTab = nullptr; int nCells = 0, cellShift = 0;   // direct-address bucket lookup (BucketTable)
  int *d_bktT = nullptr; void *d_clsCellT = nullptr, *d_cellTabT = nullptr; int nBT = 0, nCellsT = 0, cellShiftT = 0;   // the same tables over the coverage thresholds (cover_prepare)
  int *d_bktS = nullptr; void *d_clsCellS = nullptr, *d_cellTabS = nullptr; int nBS = 0, nCellsS = 0, cellShiftS = 0;          // ... and over the positions of a scan geometry (scan_bucket_tables)
  gtx::ScanPart *d_scanParts = nullptr; int nScanParts = 0; std::vector<long long> scanBktKey; gtx::DevInfo *d_scanInfo = nullptr;
  bool covTileSums = true;                           // false: a batch went through the partition path, the tile sums are rebuilt before the finalize step
  int *d_bkt = nullptr; int nB = 0;                  // posHi | eLo | eHi | sLo | sHi | cls (nB each) | clsStart (nClasses+1)
  unsigned *d_bktCnt = nullptr, *d_bktDir = nullptr; size_t capBktMatrix = 0;   // scratch of the bucket path (gtx::BucketWork)
  void *d_bktReads = nullptr; int *d_bktWeights = nullptr; size_t capBkt = 0;
  int64_t bucketMinReads = 1 << 18;                  // below this the per-read search kernel is used (GTX_BUCKET_MIN_READS)
  int *d_posE = nullptr, *d_posS = nullptr, *d_classBase = nullptr;
  u64 *d_histA = nullptr, *d_histB = nullptr, *d_partA = nullptr, *d_partB = nullptr, *d_prefA = nullptr, *d_prefB = nullptr;
  unsigned *d_chainFlags = nullptr; unsigned chainEpoch = 0; unsigned long long chainDraws = 0;    // finalize_scan_chained_kernel: a flag per tile and histogram, the call's epoch
  // The group's device calls (gtxi_count_device_share_async) finalize call k on the group's exchange stream UNDER the streaming kernel
  // of call k+1: two more sets of histograms, tile sums, prefix arrays and chain flags in turn (never set 0: the other entry points
  // stay as they are), three info blocks (call k counts into block k % 3, its finalize resets block (k + 2) % 3 -- the one call k+1
  // uses was reset by the finalize of call k-1, which the streaming kernel of call k+1 is made to wait for anyway).
  struct HistSet { u64 *histA = nullptr, *histB = nullptr, *partA = nullptr, *partB = nullptr, *prefA = nullptr, *prefB = nullptr; unsigned *flags = nullptr; unsigned epoch = 0; unsigned long long draws = 0; } alt[GTXI_SHARE_STREAMS];
  gtx::DevInfo *d_info3 = nullptr; unsigned shareSeq = 0; unsigned shareTurn[GTXI_SHARE_STREAMS] = {}; const gtx::DevInfo *lastShareInfo = nullptr;
  bool histDirty = false;              // a call was abandoned between begin and end
  // coverage (allocated on first use): 8 histograms, 8 tile-sum arrays, 8 prefix arrays, region coordinates
  // coverage (made on first use): the merged threshold array of the regions (E_k and S_k - 1, sorted per class) with its
  // 4 histograms, 4 tile-sum arrays, 4 prefix arrays; region coordinates in file order
  u64 *d_cov[12] = {}; int *d_refS = nullptr, *d_refE = nullptr; bool covReady = false, covDirty = false, covOpen = false;
  int *d_sortedT = nullptr, *d_segT = nullptr, *d_topT = nullptr, *d_posTE = nullptr, *d_posTS = nullptr, *d_classBaseT = nullptr;
  int64_t histLenT = 0;
  std::vector<int32_t> h_refS, h_refE, h_refC;
  // sorted-merge semantics, intervals with start > end + 1 (gtx_special.hip): the K inverted reference regions and their sums,
  // the inverted reads the kernels set aside, the region columns the second pair kernel reads
  bool mergeRefs = false;               // the reference set was given with GTX_REFS_KEEP_ZERO_LENGTH
  int nSpecial = 0; int4 *d_specialRefs = nullptr; int *d_specialIdx = nullptr; u64 *d_specialOut = nullptr;
  int4 *d_side = nullptr; unsigned *d_sideCount = nullptr; int sideCap = 1 << 20; int *d_refC = nullptr; bool sideUsed = false;
  int specialMode = 0;                  // value of a pair in the open call: 0 count, 2 the -gaps coverage formula
  bool tileSumsValid = true;           // every kernel since the last finalize maintained the tile sums
  int64_t histLen = 0;
  // count without -gaps over multi-interval regions (gtx_pairs.hip): envelope indexes over the multi-interval index regions
  // (reads with one interval are checked against them batch by batch) and over all regions (made on the first multi-interval
  // read), the regions' interval lists, and the two correction vectors add[nRefs] | sub[nRefs] (zero between calls)
  struct PairIdx { int *d_mem = nullptr; int n = 0; bool built = false; gtx::PairIndex ix = {}; };
  PairIdx pairMulti, pairAll;
  int2 *d_blkOf = nullptr, *d_blkIv = nullptr; bool refBlocks = false;
  u64 *d_pairAcc = nullptr; bool pairUsed = false;
  int4 *d_pairQ = nullptr; size_t capPairQ = 0, capPairIv = 0; int2 *d_pairQBlk = nullptr, *d_pairQIv = nullptr;

  gtx::DevInfo *d_info = nullptr;       // 2 blocks: the finalize of one call resets the block of the next
  int infoCur = 0;
  gtx::DevInfo *h_info = nullptr;       // pinned: [0] = readback, [1] = init pattern

  // staging for the host-buffer entry points: two device slots fed from two pinned host slots by a copy stream, so that
  // the host->device copy of batch i+1 runs under the kernels of batch i (Stage* functions below)
  hipStream_t copyStream = nullptr;
  void *d_stage[2] = {nullptr, nullptr}; int *d_stageW[2] = {nullptr, nullptr}; size_t capStage = 0, capStageW = 0;
  char *h_pin[2] = {nullptr, nullptr}; size_t capPin = 0;       // bytes per slot
  hipEvent_t evCopied[2] = {nullptr, nullptr}, evConsumed[2] = {nullptr, nullptr}; bool slotBusy[2] = {false, false};
  long long stageSeq = 0; bool directPending = false;   // a DMA may still be reading the page-locked buffer of the last call
  int copyThreads = 8;                  // host threads that move a pageable batch into the pinned slot (GTX_COPY_THREADS)
  u64 *d_out = nullptr; size_t capOut = 0;
  char *d_scratch = nullptr; size_t capScratch = 0;   // gtxi_scratch
  // region text tokenised on the device (gtx_count_add_text): two blocks in flight
  struct TextSlot {
    char *d_text = nullptr; size_t capText = 0; unsigned *d_seg = nullptr; size_t capSeg = 0;
    unsigned *d_nl = nullptr; int *d_tri = nullptr, *d_w = nullptr, *d_tri2 = nullptr, *d_w2 = nullptr; unsigned *d_blk = nullptr; size_t capLines = 0, capLines2 = 0;
    int *d_flag = nullptr, *h_flag = nullptr; char *h_pin = nullptr; size_t capPin = 0; char *h_seam = nullptr; unsigned long long *d_sum = nullptr;
    hipEvent_t evParsed = nullptr, evConsumed = nullptr, evCopied = nullptr; bool busy = false;
  } text[2];
  long long textSeq = 0;
  int *d_textTable = nullptr; char *d_textNames = nullptr; size_t capTextNames = 0; unsigned textMask = 0, textBlobLen = 0;
  std::string textBlob;                               // the chromosome names the device tables were built from

  // scan state
  u64 *d_micro = nullptr; size_t capMicro = 0;
  long long *d_scanTab = nullptr; size_t capScanTab = 0;
  std::vector<long long> scanKey;       // geometry the tables on the device were built for
  int64_t scanTotalWindows = 0, scanTotalMicro = 0, scanTotalTiles = 0;
  // owner-computes scan of sorted reads (gtx_scanown.hip): block table for the current tile, bounds scratch, give-up flag;
  // reads of a host-buffer call held resident for its single launch
  int64_t scanOwnBlocks = 0; long long *d_scanBounds = nullptr; size_t capScanBounds = 0; int *d_scanFlag = nullptr;
  void *d_resReads = nullptr; int *d_resWeights = nullptr; size_t capRes = 0, capResW = 0; hipEvent_t evRes[2] = {nullptr, nullptr};

  // measurement
  static constexpr int kProfSlots = 64;   // ring: the last 64 profiled calls can be read back
  bool prof = false; long long profCalls = 0;
  int profEvery = 1; long long profSeq = 0; bool profThis = false;   // gtx_profile_enable(N >= 2): kernel-only events on every N-th call
  hipEvent_t evRing[kProfSlots][4] = {};
  hipEvent_t *ev = evRing[0];

  // streaming count (begin/add/end)
  bool streamOpen = false; int64_t streamSeen = 0; int32_t streamLast[2] = {0, 0};
  // gtx_scan_begin .. gtx_scan_end: the open scan's geometry and what its batches have added so far
  struct ScanOpen { bool open = false, weighted = false; gtx::ScanArgs a; std::vector<int32_t> classLen; int64_t extent = 0; char prep = '1'; uint32_t flags = 0;
                    unsigned long long *d_labelSum = nullptr; } scan;
  int64_t seamUnsorted = INT64_MAX;    // first order violation found at a seam between batches (host-side check)

  int64_t batchReads = 8ll << 20;       // reads per device batch of the host-buffer entry points (96 MiB of triples: ~2 ms of PCIe)
  int chunksPerWave = 0;                // 0 = choose per call from the number of reads
  int64_t waveSlots = 8192;             // resident waves of the device (CUs x 32)
  int prefetch = 4;                     // reads per lane per step (R) of the streaming kernel (GTX_READS_PER_LANE)
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return GTX_E_HIP; } } while (0)

static int fail(gtx_ctx *c, int code, const char *msg) { c->err = msg; return code; }

template <class T> static void dfree(T *&p);
static void free_alt_sets(gtx_ctx *c);

template <class T> static void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
static void free_alt_sets(gtx_ctx *c)
{
  for (auto &h : c->alt) { dfree(h.histA); dfree(h.histB); dfree(h.partA); dfree(h.partB); dfree(h.prefA); dfree(h.prefB); dfree(h.flags); h.epoch = 0; h.draws = 0; }
  c->shareSeq = 0; c->lastShareInfo = nullptr; for (unsigned &t : c->shareTurn) t = 0;
  dfree(c->d_info3);                              // (the ring restarts with clean blocks: the last calls' blocks are cleared by calls that never came)
}

// direct placement (gtx::PlaceTable) over two boundary arrays with the class segments `seg` (the same array twice for the
// coverage thresholds): cells of 2^sh positions, sh the smallest shift that keeps the table at about one cell per eight
// boundaries; a cell's entries = how many boundaries of the class lie below the cell's first position in either array (cell 0: none)
static int make_place_table(gtx_ctx *c, const std::vector<int32_t> &seg, const std::vector<int32_t> &arrA, const std::vector<int32_t> &arrB,
                            int nClasses, int64_t nv, int4 **d_cls, int **d_rank, int *shift)
{
  const int64_t budget = std::max<int64_t>(1024, nv / 8) + 2 * (int64_t)nClasses;
  auto cellsOf = [&](int cl, int sh) -> int64_t {
    if (seg[cl] == seg[cl + 1]) return 0;
    const int64_t top = std::max<int64_t>(0, std::max(arrA[seg[cl + 1] - 1], arrB[seg[cl + 1] - 1]));
    return (top >> sh) + 2;
  };
  int sh = 0;
  for (;; sh++) { int64_t t = 0; for (int cl = 0; cl < nClasses; cl++) t += cellsOf(cl, sh); if (t <= budget || sh >= 31) break; }
  std::vector<int4> pc(std::max(nClasses, 1));
  std::vector<int32_t> rank;
  for (int cl = 0; cl < nClasses; cl++) {
    const int64_t nc = cellsOf(cl, sh);
    pc[cl] = make_int4(seg[cl], seg[cl + 1], (int)(rank.size() / 2), (int)nc);
    int32_t ia = seg[cl], ib = seg[cl];
    for (int64_t k = 0; k < nc; k++) {
      const int64_t first = k << sh;                           // cell 0 stands for everything below 2^sh, negative keys included
      if (k > 0) { while (ia < seg[cl + 1] && arrA[ia] < first) ia++; while (ib < seg[cl + 1] && arrB[ib] < first) ib++; }
      rank.push_back(ia); rank.push_back(ib);
    }
  }
  rank.push_back(0); rank.push_back(0);
  dfree(*d_cls); dfree(*d_rank);
  HIPCHK(c, hipMalloc(d_cls, sizeof(int4) * pc.size()));
  HIPCHK(c, hipMalloc(d_rank, sizeof(int32_t) * rank.size()));
  HIPCHK(c, hipMemcpy(*d_cls, pc.data(), sizeof(int4) * pc.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(*d_rank, rank.data(), sizeof(int32_t) * rank.size(), hipMemcpyHostToDevice));
  *shift = sh;
  return GTX_OK;
}

extern "C" {

int gtx_version(void) { return 100; }

gtx_ctx *gtx_create(int device_id)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("gtx_create: no usable HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                     "); this library has no CPU path";
    return nullptr;
  }
  if (device_id < 0 || device_id >= n) { g_create_error = "gtx_create: device id out of range"; return nullptr; }
  if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  gtx_ctx *c = new gtx_ctx();
  c->device = device_id;
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->waveSlots = 32ll * cus; }
  if (hipMalloc(&c->d_info, 2 * sizeof(gtx::DevInfo)) != hipSuccess || hipHostMalloc(&c->h_info, 2 * sizeof(gtx::DevInfo)) != hipSuccess) {
    g_create_error = "gtx_create: allocation failed"; delete c; return nullptr;
  }
  c->h_info[1].first_unsorted = INT64_MAX; c->h_info[1].n_no_class = 0; c->h_info[1].n_degenerate = 0; c->h_info[1].first_degenerate = INT64_MAX; c->h_info[1].n_unplaced = 0; c->h_info[1].fault = 0;
  c->h_info[0] = c->h_info[1];
  if (hipMemcpy(c->d_info, &c->h_info[1], sizeof(gtx::DevInfo), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_info + 1, &c->h_info[1], sizeof(gtx::DevInfo), hipMemcpyHostToDevice) != hipSuccess) {
    g_create_error = "gtx_create: hipMemcpy failed"; delete c; return nullptr;
  }
  for (auto &slot : c->evRing) for (auto &ev : slot) if (hipEventCreate(&ev) != hipSuccess) { g_create_error = "gtx_create: hipEventCreate failed"; delete c; return nullptr; }
  if (hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "gtx_create: hipStreamCreate failed"; delete c; return nullptr; }
  for (int k = 0; k < 2; k++)
    if (hipEventCreateWithFlags(&c->evCopied[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->evConsumed[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evRes[k], hipEventDisableTiming) != hipSuccess) {
      g_create_error = "gtx_create: hipEventCreate failed"; delete c; return nullptr;
    }
  { unsigned hc = std::thread::hardware_concurrency(); c->copyThreads = (int)std::max(1u, std::min(hc ? hc : 4u, 8u));
    if (const char *ct = getenv("GTX_COPY_THREADS")) if (atoi(ct) > 0) c->copyThreads = atoi(ct); }
  const char *cpw = getenv("GTX_CHUNKS_PER_WAVE");
  if (cpw && atoi(cpw) > 0) c->chunksPerWave = atoi(cpw);
  const char *br = getenv("GTX_BATCH_READS");
  if (br && atoll(br) > 0) c->batchReads = atoll(br);
  if (const char *bm = getenv("GTX_BUCKET_MIN_READS")) c->bucketMinReads = atoll(bm);   // unsorted reads: batches below this use the search kernel
  const char *pf = getenv("GTX_READS_PER_LANE");        // tuning knob (1..4), default 4
  if (pf && atoi(pf) > 0) c->prefetch = atoi(pf);
  return c;
}

void gtx_destroy(gtx_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  dfree(c->d_sortedE); dfree(c->d_sortedS); dfree(c->d_segStart); dfree(c->d_posE); dfree(c->d_posS); dfree(c->d_classBase);
  dfree(c->d_bktReads); dfree(c->d_bktWeights); dfree(c->d_bktDir); dfree(c->d_sampE); dfree(c->d_sampS); dfree(c->d_topE); dfree(c->d_topS); dfree(c->d_bkt); dfree(c->d_clsCell); dfree(c->d_cellTab); dfree(c->d_bktT); dfree(c->d_clsCellT); dfree(c->d_cellTabT); c->nBT = 0;
  dfree(c->d_placeCls); dfree(c->d_placeRank); dfree(c->d_placeClsT); dfree(c->d_placeRankT); dfree(c->d_shareTiles); dfree(c->d_shareRegions); dfree(c->d_shareOwned);
  dfree(c->d_bktCnt); dfree(c->d_bktS); dfree(c->d_clsCellS); dfree(c->d_cellTabS); dfree(c->d_scanParts); dfree(c->d_scanInfo); c->nBS = 0;
  dfree(c->d_histA); dfree(c->d_histB); dfree(c->d_partA); dfree(c->d_partB); dfree(c->d_prefA); dfree(c->d_prefB); dfree(c->d_info); dfree(c->d_chainFlags); free_alt_sets(c); dfree(c->d_info3); dfree(c->scan.d_labelSum);
  if (c->copyStream) (void)hipStreamSynchronize(c->copyStream);
  for (int k = 0; k < 2; k++) {
    dfree(c->d_stage[k]); dfree(c->d_stageW[k]);
    if (c->h_pin[k]) (void)hipHostFree(c->h_pin[k]);
    if (c->evCopied[k]) (void)hipEventDestroy(c->evCopied[k]);
    if (c->evConsumed[k]) (void)hipEventDestroy(c->evConsumed[k]);
    if (c->evRes[k]) (void)hipEventDestroy(c->evRes[k]);
  }
  if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
  for (auto &t : c->text) {
    dfree(t.d_text); dfree(t.d_seg); dfree(t.d_nl); dfree(t.d_tri); dfree(t.d_w); dfree(t.d_tri2); dfree(t.d_w2); dfree(t.d_blk); dfree(t.d_flag); dfree(t.d_sum);
    if (t.h_flag) (void)hipHostFree(t.h_flag);
    if (t.h_pin) (void)hipHostFree(t.h_pin);
    if (t.h_seam) (void)hipHostFree(t.h_seam);
    for (hipEvent_t e : {t.evParsed, t.evConsumed, t.evCopied}) if (e) (void)hipEventDestroy(e);
  }
  dfree(c->d_textTable); dfree(c->d_textNames);
  dfree(c->d_out); dfree(c->d_scratch); dfree(c->d_micro); dfree(c->d_scanTab); dfree(c->d_scanBounds); dfree(c->d_scanFlag); dfree(c->d_resReads); dfree(c->d_resWeights);
  for (auto &p : c->d_cov) dfree(p);
  dfree(c->d_sortedT); dfree(c->d_segT); dfree(c->d_topT); dfree(c->d_posTE); dfree(c->d_posTS); dfree(c->d_classBaseT);
  dfree(c->d_refS); dfree(c->d_refE); dfree(c->d_refC); dfree(c->d_specialRefs); dfree(c->d_specialIdx); dfree(c->d_specialOut); dfree(c->d_side); dfree(c->d_sideCount);
  dfree(c->pairMulti.d_mem); dfree(c->pairAll.d_mem); dfree(c->d_blkOf); dfree(c->d_blkIv); dfree(c->d_pairAcc); dfree(c->d_pairQ); dfree(c->d_pairQBlk); dfree(c->d_pairQIv);
  if (c->h_info) (void)hipHostFree(c->h_info);
  for (auto &slot : c->evRing) for (auto &ev : slot) if (ev) (void)hipEventDestroy(ev);
  delete c;
}

const char *gtx_last_error(const gtx_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int gtx_set_stream(gtx_ctx *c, void *s) { if (!c) return GTX_E_ARG; c->stream = (hipStream_t)s; return GTX_OK; }

int gtx_sync(gtx_ctx *c)
{
  if (!c) return GTX_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->copyStream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTX_OK;
}

void *gtx_host_alloc(gtx_ctx *c, size_t bytes)
{
  if (!c) return nullptr;
  void *p = nullptr;
  // (portable: any device of a group may read it)
  if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { c->err = "gtx_host_alloc: hipHostMalloc failed"; (void)hipGetLastError(); return nullptr; }
  return p;
}

void gtx_host_free(gtx_ctx *c, void *p) { (void)c; if (p) (void)hipHostFree(p); }

int64_t gtx_n_refs(const gtx_ctx *c) { return c ? c->nRefs : -1; }

// ---------------------------------------------------------------------------------------------
// reference side
// ---------------------------------------------------------------------------------------------
int gtx_set_refs(gtx_ctx *c, const int32_t *tri, int64_t m, int32_t nClasses) { return gtx_set_refs_ex(c, tri, m, nClasses, 0); }

int gtx_set_refs_ex(gtx_ctx *c, const int32_t *tri, int64_t m, int32_t nClasses, uint32_t flags)
{
  if (!c || m < 0 || (m > 0 && !tri)) return c ? fail(c, GTX_E_ARG, "gtx_set_refs: bad argument") : GTX_E_ARG;
  if (m >= (int64_t)INT32_MAX - 4096) return fail(c, GTX_E_ARG, "gtx_set_refs: too many reference regions");
  HIPCHK(c, hipSetDevice(c->device));
  int maxc = -1;
  for (int64_t k = 0; k < m; k++) {
    int32_t cl = tri[3 * k], s = tri[3 * k + 1], e = tri[3 * k + 2];
    if (cl < -1) return fail(c, GTX_E_RANGE, "gtx_set_refs: negative class id");      // -1: a placeholder that never matches
    if (s >= INT32_MAX - 1 || e >= INT32_MAX - 1) return fail(c, GTX_E_RANGE, "gtx_set_refs: coordinate >= 2^31-2");
    if (cl > maxc) maxc = cl;
  }
  if (nClasses <= 0) nClasses = maxc + 1;
  if (maxc >= nClasses) return fail(c, GTX_E_RANGE, "gtx_set_refs: class id >= n_classes");
  if (nClasses < 1) nClasses = 1;

  // valid regions only take part (genomic_intervals.cpp:5659: start>stop or stop<=0 is skipped)
  std::vector<u64> keyE, keyS;                   // (class, biased coordinate, ordinal) packed for one sort each
  std::vector<int32_t> ord;
  ord.reserve(m);
  const bool keepZero = (flags & GTX_REFS_KEEP_ZERO_LENGTH) != 0;
  for (int64_t k = 0; k < m; k++) {
    int32_t s = tri[3 * k + 1], e = tri[3 * k + 2];
    const bool take = tri[3 * k] >= 0 && (keepZero ? (int64_t)s <= (int64_t)e + 1 : !(s > e || e <= 0));
    if (take) ord.push_back((int32_t)k);
  }
  const int64_t nv = (int64_t)ord.size();
  struct Item { int32_t cls; int32_t val; int32_t k; };
  std::vector<Item> itE(nv), itS(nv);
  for (int64_t i = 0; i < nv; i++) {
    int32_t k = ord[i];
    itE[i] = {tri[3 * (int64_t)k], tri[3 * (int64_t)k + 2], k};
    itS[i] = {tri[3 * (int64_t)k], tri[3 * (int64_t)k + 1], k};
  }
  auto cmp = [](const Item &a, const Item &b) { return a.cls != b.cls ? a.cls < b.cls : (a.val != b.val ? a.val < b.val : a.k < b.k); };
  if (nv > (1 << 16)) {                                      // the two orders are independent: sort them side by side
    std::thread other([&] { std::sort(itS.begin(), itS.end(), cmp); });
    std::sort(itE.begin(), itE.end(), cmp);
    other.join();
  } else {
    std::sort(itE.begin(), itE.end(), cmp);
    std::sort(itS.begin(), itS.end(), cmp);
  }

  std::vector<int32_t> sortedE(nv + 1), sortedS(nv + 1), seg(nClasses + 1, 0), posE(m > 0 ? m : 1, -1), posS(m > 0 ? m : 1, -1), classBase(m > 0 ? m : 1, -1);
  for (int64_t i = 0; i < nv; i++) seg[itE[i].cls + 1]++;
  for (int cl = 0; cl < nClasses; cl++) seg[cl + 1] += seg[cl];
  for (int64_t i = 0; i < nv; i++) {
    sortedE[i] = itE[i].val; posE[itE[i].k] = (int32_t)(i + itE[i].cls);
    sortedS[i] = itS[i].val; posS[itS[i].k] = (int32_t)(i + itS[i].cls);
    classBase[itE[i].k] = seg[itE[i].cls] + itE[i].cls - 1;
  }

  dfree(c->d_sortedE); dfree(c->d_sortedS); dfree(c->d_segStart); dfree(c->d_posE); dfree(c->d_posS); dfree(c->d_classBase);
  dfree(c->d_sampE); dfree(c->d_sampS); dfree(c->d_topE); dfree(c->d_topS); dfree(c->d_bkt); dfree(c->d_clsCell); dfree(c->d_cellTab);
  dfree(c->d_placeCls); dfree(c->d_placeRank);
  dfree(c->d_bktT); dfree(c->d_clsCellT); dfree(c->d_cellTabT); c->nBT = 0;      // tables over the coverage thresholds: rebuilt by cover_prepare
  dfree(c->d_histA); dfree(c->d_histB); dfree(c->d_partA); dfree(c->d_partB); dfree(c->d_prefA); dfree(c->d_prefB); dfree(c->d_chainFlags); free_alt_sets(c);
  c->nRefs = -1;
  const int64_t histLen = nv + nClasses;
  const int nTiles = gtx::scan_tiles(histLen);
  HIPCHK(c, hipMalloc(&c->d_sortedE, sizeof(int32_t) * (nv + 1)));
  HIPCHK(c, hipMalloc(&c->d_sortedS, sizeof(int32_t) * (nv + 1)));
  HIPCHK(c, hipMalloc(&c->d_segStart, sizeof(int32_t) * (nClasses + 1)));
  HIPCHK(c, hipMalloc(&c->d_posE, sizeof(int32_t) * (m + 1)));
  HIPCHK(c, hipMalloc(&c->d_posS, sizeof(int32_t) * (m + 1)));
  HIPCHK(c, hipMalloc(&c->d_classBase, sizeof(int32_t) * (m + 1)));
  HIPCHK(c, hipMalloc(&c->d_histA, sizeof(u64) * histLen));
  HIPCHK(c, hipMalloc(&c->d_histB, sizeof(u64) * histLen));
  HIPCHK(c, hipMalloc(&c->d_partA, sizeof(u64) * (nTiles + 2)));
  HIPCHK(c, hipMalloc(&c->d_partB, sizeof(u64) * (nTiles + 2)));
  HIPCHK(c, hipMalloc(&c->d_prefA, sizeof(u64) * histLen));
  HIPCHK(c, hipMalloc(&c->d_prefB, sizeof(u64) * histLen));
  // invariant between calls: histograms and tile sums are all zero (the finalize kernels leave them so)
  HIPCHK(c, hipMemset(c->d_histA, 0, sizeof(u64) * histLen));
  HIPCHK(c, hipMemset(c->d_histB, 0, sizeof(u64) * histLen));
  HIPCHK(c, hipMemset(c->d_partA, 0, sizeof(u64) * (nTiles + 2)));
  HIPCHK(c, hipMemset(c->d_partB, 0, sizeof(u64) * (nTiles + 2)));
  HIPCHK(c, hipMalloc(&c->d_chainFlags, sizeof(unsigned) * 8 * (nTiles + 2)));
  HIPCHK(c, hipMemset(c->d_chainFlags, 0, sizeof(unsigned) * 8 * (nTiles + 2)));
  c->chainEpoch = 0; c->chainDraws = 0;
  c->histDirty = false; c->tileSumsValid = true;
  HIPCHK(c, hipMemcpy(c->d_sortedE, sortedE.data(), sizeof(int32_t) * nv, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_sortedS, sortedS.data(), sizeof(int32_t) * nv, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_segStart, seg.data(), sizeof(int32_t) * (nClasses + 1), hipMemcpyHostToDevice));
  {
    c->sampShift = gtx::search_sample_shift(nv);
    c->nSamp = (int)((nv + (1ll << c->sampShift) - 1) >> c->sampShift);
    std::vector<int32_t> sampE(c->nSamp + 1), sampS(c->nSamp + 1);
    for (int i = 0; i < c->nSamp; i++) { sampE[i] = sortedE[(int64_t)i << c->sampShift]; sampS[i] = sortedS[(int64_t)i << c->sampShift]; }
    HIPCHK(c, hipMalloc(&c->d_sampE, sizeof(int32_t) * (c->nSamp + 1)));
    HIPCHK(c, hipMalloc(&c->d_sampS, sizeof(int32_t) * (c->nSamp + 1)));
    HIPCHK(c, hipMemcpy(c->d_sampE, sampE.data(), sizeof(int32_t) * (c->nSamp + 1), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_sampS, sampS.data(), sizeof(int32_t) * (c->nSamp + 1), hipMemcpyHostToDevice));
    const int64_t nTop = (nv + 255) >> 8;
    std::vector<int32_t> topE(nTop + 1), topS(nTop + 1);
    for (int64_t i = 0; i < nTop; i++) { topE[i] = sortedE[i << 8]; topS[i] = sortedS[i << 8]; }
    HIPCHK(c, hipMalloc(&c->d_topE, sizeof(int32_t) * (nTop + 1)));
    HIPCHK(c, hipMalloc(&c->d_topS, sizeof(int32_t) * (nTop + 1)));
    HIPCHK(c, hipMemcpy(c->d_topE, topE.data(), sizeof(int32_t) * (nTop + 1), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_topS, topS.data(), sizeof(int32_t) * (nTop + 1), hipMemcpyHostToDevice));
  }
  { int rc = make_place_table(c, seg, sortedE, sortedS, nClasses, nv, &c->d_placeCls, &c->d_placeRank, &c->placeShift); if (rc) return rc; }
  {
    // bucket table of the unsorted path: cuts of the ends array every bucket_e_size() boundaries, never across classes
    const int kE = gtx::bucket_e_size(), kS = gtx::bucket_s_size();
    std::vector<int32_t> posHi, eLo, eHi, sLo, sHi, cls, clsStart(nClasses + 1, 0);
    for (int cl = 0; cl < nClasses; cl++) {
      clsStart[cl] = (int32_t)posHi.size();
      const int32_t s0 = seg[cl], s1 = seg[cl + 1];
      for (int32_t e0 = s0; e0 < s1; e0 += kE) {
        const int32_t e1 = std::min<int64_t>((int64_t)e0 + kE, s1);
        posHi.push_back(e1 == s1 ? INT32_MAX : sortedE[e1 - 1]);
        eLo.push_back(e0); eHi.push_back(e1); cls.push_back(cl);
        // a read of this bucket starts above E[e0-1], so it ends at or above it: ranks in the starts array begin here
        const int32_t lo = e0 == s0 ? s0 : (int32_t)(std::upper_bound(sortedS.begin() + s0, sortedS.begin() + s1, sortedE[e0 - 1]) - sortedS.begin());
        sLo.push_back(lo); sHi.push_back((int32_t)std::min<int64_t>((int64_t)lo + kS, s1));
      }
    }
    clsStart[nClasses] = (int32_t)posHi.size();
    c->nB = (int)posHi.size();
    if (c->nB > 8192 || nClasses > 2048) c->nB = 0;               // (certainly too many for the LDS tables of the scatter kernel; the exact test follows)
    std::vector<int32_t> clsCell(4 * (size_t)nClasses);
    std::vector<uint16_t> cellTab;
    if (c->nB > 0) {
      // cells over the span of each class's cuts: the shift that keeps all classes within kCells cells
      const int kCells = 4096;
      int sh = 0;
      auto cellsAt = [&](int shift) {
        int64_t total = 0;
        for (int cl = 0; cl < nClasses; cl++) {
          const int b0 = clsStart[cl], b1 = clsStart[cl + 1];
          total += b1 - b0 <= 1 ? b1 - b0 : ((((int64_t)posHi[b1 - 2] - posHi[b0]) >> shift) + 1);
        }
        return total;
      };
      while (sh < 40 && cellsAt(sh) > kCells) sh++;
      for (int cl = 0; cl < nClasses; cl++) {
        const int b0 = clsStart[cl], b1 = clsStart[cl + 1];
        const int32_t lo = b1 - b0 <= 1 ? 0 : posHi[b0];
        const int64_t nc = b1 == b0 ? 0 : b1 - b0 == 1 ? 1 : ((((int64_t)posHi[b1 - 2] - lo) >> sh) + 1);   // 0 cells: a class without reference regions
        clsCell[4 * cl] = (int32_t)cellTab.size(); clsCell[4 * cl + 1] = lo; clsCell[4 * cl + 2] = (int32_t)nc; clsCell[4 * cl + 3] = b0;
        int b = b0;
        for (int64_t k = 0; k < nc; k++) {
          const int64_t first = (int64_t)lo + (k << sh);
          while (b < b1 - 1 && (int64_t)posHi[b] < first) b++;
          cellTab.push_back((uint16_t)(b - b0));                   // relative to the class's first bucket
        }
      }
      c->nCells = (int)cellTab.size(); c->cellShift = sh;
      if (!gtx::bucket_tables_fit(nClasses, c->nB, c->nCells)) c->nB = 0;     // 32 B per bucket, 16 per class: the search kernel serves
    }
    if (c->nB > 0) {
      cellTab.push_back(0);
      HIPCHK(c, hipMalloc(&c->d_clsCell, sizeof(int32_t) * clsCell.size() + 16));
      HIPCHK(c, hipMemcpy(c->d_clsCell, clsCell.data(), sizeof(int32_t) * clsCell.size(), hipMemcpyHostToDevice));
      HIPCHK(c, hipMalloc(&c->d_cellTab, sizeof(uint16_t) * cellTab.size()));
      HIPCHK(c, hipMemcpy(c->d_cellTab, cellTab.data(), sizeof(uint16_t) * cellTab.size(), hipMemcpyHostToDevice));
      std::vector<int32_t> all;
      for (auto *v : {&posHi, &eLo, &eHi, &sLo, &sHi, &cls, &clsStart}) all.insert(all.end(), v->begin(), v->end());
      HIPCHK(c, hipMalloc(&c->d_bkt, sizeof(int32_t) * all.size()));
      HIPCHK(c, hipMemcpy(c->d_bkt, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice));
    }
  }
  if (m > 0) {
    HIPCHK(c, hipMemcpy(c->d_posE, posE.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_posS, posS.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_classBase, classBase.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
  }
  for (auto &p : c->d_cov) dfree(p);
  dfree(c->d_sortedT); dfree(c->d_segT); dfree(c->d_topT); dfree(c->d_posTE); dfree(c->d_posTS); dfree(c->d_classBaseT);
  dfree(c->d_refS); dfree(c->d_refE); c->covReady = false; c->covDirty = false;
  c->h_refS.resize(m > 0 ? m : 1); c->h_refE.resize(m > 0 ? m : 1); c->h_refC.resize(m > 0 ? m : 1);
  for (int64_t k = 0; k < m; k++) { c->h_refC[k] = tri[3 * k]; c->h_refS[k] = tri[3 * k + 1]; c->h_refE[k] = tri[3 * k + 2]; }
  dfree(c->d_refC); dfree(c->d_specialRefs); dfree(c->d_specialIdx); dfree(c->d_specialOut);
  c->mergeRefs = keepZero; c->nSpecial = 0;
  if (keepZero) {
    std::vector<int4> sp; std::vector<int32_t> spIdx;
    for (int64_t k = 0; k < m; k++)
      if (tri[3 * k] >= 0 && (int64_t)tri[3 * k + 1] > (int64_t)tri[3 * k + 2] + 1) { sp.push_back(make_int4(tri[3 * k], tri[3 * k + 1], tri[3 * k + 2], 0)); spIdx.push_back((int32_t)k); }
    c->nSpecial = (int)sp.size();
    if (c->nSpecial) {
      HIPCHK(c, hipMalloc(&c->d_specialRefs, sizeof(int4) * sp.size()));
      HIPCHK(c, hipMalloc(&c->d_specialIdx, sizeof(int32_t) * sp.size()));
      HIPCHK(c, hipMalloc(&c->d_specialOut, sizeof(u64) * sp.size()));
      HIPCHK(c, hipMemcpy(c->d_specialRefs, sp.data(), sizeof(int4) * sp.size(), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->d_specialIdx, spIdx.data(), sizeof(int32_t) * sp.size(), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemset(c->d_specialOut, 0, sizeof(u64) * sp.size()));
    }
  }
  dfree(c->pairMulti.d_mem); dfree(c->pairAll.d_mem); dfree(c->d_blkOf); dfree(c->d_blkIv); dfree(c->d_pairAcc);
  c->pairMulti = gtx_ctx::PairIdx(); c->pairAll = gtx_ctx::PairIdx(); c->refBlocks = false; c->pairUsed = false;
  c->nRefs = m; c->nValid = nv; c->nClasses = nClasses; c->histLen = histLen;
  c->h_seg = seg;
  c->shareOn = false; dfree(c->d_shareTiles); dfree(c->d_shareRegions); dfree(c->d_shareOwned); c->nShareTiles = 0; c->nShareRegions = 0; c->shareOffset = 0;
  return GTX_OK;
}

// ---------------------------------------------------------------------------------------------
// count
// ---------------------------------------------------------------------------------------------
#ifdef GTX_WAVE_TRACE
// diagnostic build (make trace -> libgtx_trace.so, scripts/wave_trace.py): per-wave time stamps of the streaming count kernel
static constexpr size_t kTraceWaves = 1u << 20;
static unsigned long long *g_trace = nullptr;
static unsigned long long *gtx_debug_trace_buffer()
{
  if (!g_trace && hipMalloc(&g_trace, kTraceWaves * 32) == hipSuccess) (void)hipMemset(g_trace, 0, kTraceWaves * 32);
  return g_trace;
}
extern "C" int gtx_debug_trace_read(unsigned long long *out, long long nWaves)
{
  if (!g_trace || nWaves > (long long)kTraceWaves) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out, g_trace, (size_t)nWaves * 32, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  return hipMemset(g_trace, 0, kTraceWaves * 32) == hipSuccess ? 0 : -4;      // (the next launch may have fewer waves)
}
#endif

// hist32: the call is ONE launch of the streaming kernel over unweighted reads (a *_device call): 32-bit histogram slots
// (CountArgs::hist32); the caller hands the same flag to launch_finalize
static gtx::CountArgs count_args(gtx_ctx *c, uint32_t flags, int64_t nReads, int64_t indexBase = 0, const gtx_ctx::HistSet *set = nullptr, bool share = false,
                                 bool hist32 = false)
{
  gtx::CountArgs a;
  a.indexBase = indexBase;
  { static const bool off = (getenv("GTX_HIST32") && atoi(getenv("GTX_HIST32")) == 0) || (getenv("GTX_PF") && atoi(getenv("GTX_PF")));
    a.hist32 = hist32 && !off && nReads < (1ll << 32) && c->prefetch >= 4; }
  a.owned = (share || set) && c->shareOn ? c->d_shareOwned : nullptr;
  a.sortedE = c->d_sortedE; a.sortedS = c->d_sortedS; a.segStart = c->d_segStart;
  a.histA = c->d_histA; a.histB = c->d_histB; a.partA = c->d_partA; a.partB = c->d_partB; a.info = c->d_info + c->infoCur;
  if (set) { a.histA = set->histA; a.histB = set->histB; a.partA = set->partA; a.partB = set->partB; }
  a.nClasses = c->nClasses;
  // Large batches: the streaming kernel leaves the per-tile sums alone and the finalize step rebuilds them from the
  // histograms (tile_sums_kernel, one pass over 16 B per region: 6 us at 1 M regions).  Keeping them up to date costs two
  // more atomics and a wave scan per window flush -- 100 M reads x 1 M regions: kernel 0.218 -> 0.207 ms, step 0.253 ->
  // 0.248 ms; with few regions every wave hits the same handful of counters and the same-address atomics serialise
  // (10 k regions: 0.28 -> 0.19 ms; a group member's 1/8 of 100 M reads over its 3 chromosomes: 0.041 -> 0.103 ms).  Small batches
  // keep the sums (no extra launch).  GTX_PART_MAX_HIST=0 restores them.
  { static const char *mx = getenv("GTX_PART_MAX_HIST"); const int64_t lim = mx ? atoll(mx) : INT64_MAX;
    if (c->histLen <= lim && nReads >= (1 << 20)) { a.partA = nullptr; a.partB = nullptr; c->tileSumsValid = false; } }
  // span of one wave: long enough to amortise the window placement at its start, short enough that the grid has >= 3
  // rounds of the 8192 wave slots of the chip (256 CUs x 32 waves) and that the waves resident at one time read a
  // compact piece of the stream (100 M reads: flat from 48 to 64 chunks, +3 % at 96, +12 % at 192 = one round;
  // 1 G reads: 1.74 ms at 48, 1.76 at 64-96, 1.82 at 128; the bare load pattern behaves the same, scripts/membench.hip)
  int cpw = c->chunksPerWave;
  // (launches of one to three rounds -- a group member's share of 100 M reads -- take 16 chunks per wave from a full round of
  // 16-chunk spans on: the start of a span is paid once per 16 chunks instead of 8, 0.039 -> 0.034 ms for 12.9 M reads; scripts/r04_share.sh)
  if (cpw <= 0) { int64_t nChunks = (nReads + 63) >> 6; cpw = (int)std::min<int64_t>(56, std::max<int64_t>(nChunks >= 16 * c->waveSlots ? 16 : 8, nChunks / 24576)); }
  const int r = std::max(1, std::min(4, c->prefetch));
  a.chunksPerWave = (cpw + r - 1) / r * r;
  a.sched = gtx::span_schedule((nReads + 63) >> 6, a.chunksPerWave, r, c->waveSlots);
  a.checkSorted = (flags & GTX_CHECK_SORTED) ? 1 : 0; a.sortClassShift = 0; a.prefetch = c->prefetch;
  a.zeroLenOk = (flags & GTX_ZERO_LENGTH_OK) ? 1 : 0;
  const bool merge = (flags & GTX_ZERO_LENGTH_OK) && c->mergeRefs && c->d_side;       // full sorted-merge semantics (see merge_prepare)
  a.side = merge ? c->d_side : nullptr; a.sideCount = merge ? c->d_sideCount : nullptr; a.sideCap = c->sideCap; a.coverRule = 0; a.keyCenter = 0;
  a.sampE = c->d_sampE; a.sampS = c->d_sampS; a.sampShift = c->sampShift; a.nSamp = c->nSamp;
  a.topE = c->d_topE; a.topS = c->d_topS;
  a.place.cls = c->d_placeCls; a.place.rank = c->d_placeRank; a.place.shift = c->placeShift;
  // dense references (>= 4 boundaries per 256 reads and array): all boundaries of a window at once instead of the
  // per-boundary loop (100 M reads x 4 M regions: 0.41 -> 0.28 ms; at 1 M regions the loop is 3 % faster).  GTX_FLIP=0|1 forces.
  // (a group member streams the reads of ITS classes only: their density is against its own regions, not the whole set's)
  { static const char *fl = getenv("GTX_FLIP"); const int64_t regions = set && c->shareOn ? std::min<int64_t>(c->nValid, c->nShareRegions) : c->nValid;
    a.flip = fl ? atoi(fl) : (regions * 256 >= 4 * std::max<int64_t>(nReads, 1)); }
#ifdef GTX_WAVE_TRACE
  a.trace = gtx_debug_trace_buffer();
#endif
  return a;
}

// scratch of the partition path for a call of plan p (grown, never shrunk), and the views the kernels take
static int bucket_scratch(gtx_ctx *c, const gtx::BucketPlan &p, int nB, gtx::BucketWork *w)
{
  if (p.pairs > c->capBkt) {
    dfree(c->d_bktReads); dfree(c->d_bktWeights); dfree(c->d_bktDir); c->capBkt = 0;
    HIPCHK(c, hipMalloc(&c->d_bktReads, 8 * p.pairs));
    HIPCHK(c, hipMalloc(&c->d_bktWeights, 4 * p.pairs));
    HIPCHK(c, hipMalloc(&c->d_bktDir, 4 * 2 * p.chunks));             // directory | list
    c->capBkt = p.pairs;
  }
  const size_t words = p.matrix + p.blocks + (size_t)nB + 1;         // chunkCount | arenaUsed | rowOff
  if (words > c->capBktMatrix) {
    dfree(c->d_bktCnt); c->capBktMatrix = 0;
    HIPCHK(c, hipMalloc(&c->d_bktCnt, 4 * words));
    c->capBktMatrix = words;
  }
  w->tmpReads = c->d_bktReads; w->tmpWeights = c->d_bktWeights; w->arenaPairs = (unsigned)p.arenaPairs;
  w->dir = c->d_bktDir; w->list = c->d_bktDir + c->capBkt / 64; w->chunkCount = c->d_bktCnt; w->arenaUsed = c->d_bktCnt + p.matrix; w->rowOff = w->arenaUsed + p.blocks;
  return GTX_OK;
}

static gtx::BucketTable bucket_table(const int *d_bkt, int nB, const void *clsCell, const void *cellTab, int nCells, int cellShift)
{
  gtx::BucketTable t;
  t.posHi = d_bkt; t.eLo = d_bkt + nB; t.eHi = d_bkt + 2 * nB; t.sLo = d_bkt + 3 * nB; t.sHi = d_bkt + 4 * nB;
  t.cls = d_bkt + 5 * nB; t.clsStart = d_bkt + 6 * nB; t.nB = nB;
  t.clsCell = (const int4 *)clsCell; t.cellTab = (const unsigned short *)cellTab; t.nCells = nCells; t.cellShift = cellShift;
  return t;
}

// reads in no particular order: bucket partition + LDS counting for large batches, per-read search kernel otherwise
static int launch_unsorted(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, const gtx::CountArgs &a)
{
  if (c->nB == 0 || n < c->bucketMinReads || n >= (1ll << 31)) {
    HIPCHK(c, gtx::launch_count(d_reads, d_weights, n, a, false, c->stream));
    return GTX_OK;
  }
  const gtx::BucketPlan p = gtx::bucket_plan(n, a.nClasses, c->nB, c->nCells, d_weights != nullptr);
  if (p.pairs >= (1ull << 32)) { HIPCHK(c, gtx::launch_count(d_reads, d_weights, n, a, false, c->stream)); return GTX_OK; }
  gtx::BucketWork w;
  { int rc = bucket_scratch(c, p, c->nB, &w); if (rc) return rc; }
  const gtx::BucketTable t = bucket_table(c->d_bkt, c->nB, c->d_clsCell, c->d_cellTab, c->nCells, c->cellShift);
  HIPCHK(c, gtx::launch_count_bucketed(d_reads, d_weights, n, a, t, w, p, c->stream));
  return GTX_OK;
}

// Full sorted-merge semantics (GTX_ZERO_LENGTH_OK on a reference set given with GTX_REFS_KEEP_ZERO_LENGTH): intervals with
// start > end + 1 take part by the merge's two comparisons (gtx_special.hip).  Buffers are made on first use.
static int ref_columns(gtx_ctx *c)
{
  if (!c->d_refS) {
    HIPCHK(c, hipMalloc(&c->d_refS, sizeof(int32_t) * (c->nRefs + 1)));
    HIPCHK(c, hipMalloc(&c->d_refE, sizeof(int32_t) * (c->nRefs + 1)));
    if (c->nRefs > 0) {
      HIPCHK(c, hipMemcpy(c->d_refS, c->h_refS.data(), sizeof(int32_t) * c->nRefs, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->d_refE, c->h_refE.data(), sizeof(int32_t) * c->nRefs, hipMemcpyHostToDevice));
    }
  }
  return GTX_OK;
}

static int merge_prepare(gtx_ctx *c, uint32_t flags, int mode)
{
  if (!(flags & GTX_ZERO_LENGTH_OK) || !c->mergeRefs) return GTX_OK;
  if (!c->d_side) {
    HIPCHK(c, hipMalloc(&c->d_side, sizeof(int4) * (size_t)c->sideCap));
    HIPCHK(c, hipMalloc(&c->d_sideCount, sizeof(unsigned)));
    HIPCHK(c, hipMemset(c->d_sideCount, 0, sizeof(unsigned)));
  }
  if (!c->d_refC) {
    int rc = ref_columns(c); if (rc) return rc;
    HIPCHK(c, hipMalloc(&c->d_refC, sizeof(int32_t) * (c->nRefs + 1)));
    if (c->nRefs > 0) HIPCHK(c, hipMemcpy(c->d_refC, c->h_refC.data(), sizeof(int32_t) * c->nRefs, hipMemcpyHostToDevice));
  }
  c->sideUsed = true; c->specialMode = mode;
  return GTX_OK;
}

// after the count kernels of one batch, multi-interval index regions (gtx_set_ref_blocks): a read with one interval that lies in
// a gap of such a region was counted on the region's envelope -- off again
static int pairs_batch(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n)
{
  if (c->pairMulti.n > 0) {
    HIPCHK(c, gtx::launch_pair_miss(d_reads, d_weights, n, c->pairMulti.ix, gtx::RegionBlocks{c->d_blkOf, c->d_blkIv}, c->d_pairAcc + c->nRefs, c->stream));
    c->pairUsed = true;
  }
  return GTX_OK;
}

// after the kernels of one batch: the batch against the inverted reference regions
static int merge_batch(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n)
{
  if (c->sideUsed && c->nSpecial) HIPCHK(c, gtx::launch_special_refs(d_reads, d_weights, n, c->d_specialRefs, c->nSpecial, c->specialMode, c->d_specialOut, c->stream));
  return GTX_OK;
}

// after the gather: the inverted reads against the other regions, then the inverted regions' sums into their places
static int merge_end(gtx_ctx *c, void *d_out)
{
  if (c->pairUsed) {
    c->pairUsed = false;
    HIPCHK(c, gtx::launch_pair_apply((u64 *)d_out, c->d_pairAcc, c->d_pairAcc + c->nRefs, c->nRefs, c->stream));
  }
  if (!c->sideUsed) return GTX_OK;
  c->sideUsed = false;
  HIPCHK(c, gtx::launch_side_reads(c->d_refC, c->d_refS, c->d_refE, c->nRefs, c->d_side, c->d_sideCount, c->sideCap, c->specialMode, (u64 *)d_out,
                                   c->d_info + c->infoCur, c->stream));
  HIPCHK(c, gtx::launch_special_scatter(c->d_specialIdx, c->d_specialOut, c->nSpecial, (u64 *)d_out, c->d_sideCount, c->stream));
  return GTX_OK;
}

// begin: zero histograms + info; accumulate: one kernel per resident batch; end: prefix + gather
static int count_begin(gtx_ctx *c)
{
  if (c->histDirty) {                             // only after an abandoned call
    const int nTiles = gtx::scan_tiles(c->histLen);
    HIPCHK(c, hipMemsetAsync(c->d_histA, 0, sizeof(u64) * c->histLen, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_histB, 0, sizeof(u64) * c->histLen, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_partA, 0, sizeof(u64) * (nTiles + 2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_partB, 0, sizeof(u64) * (nTiles + 2), c->stream));
  }
  // the info block is shared by count and coverage calls: an abandoned call of either kind leaves counts in it
  if (c->histDirty || c->covDirty) HIPCHK(c, hipMemcpyAsync(c->d_info + c->infoCur, &c->h_info[1], sizeof(gtx::DevInfo), hipMemcpyHostToDevice, c->stream));
  if ((c->histDirty || c->covDirty) && c->d_sideCount) {            // an abandoned call may have left inverted reads / sums behind
    HIPCHK(c, hipMemsetAsync(c->d_sideCount, 0, sizeof(unsigned), c->stream));
    if (c->nSpecial) HIPCHK(c, hipMemsetAsync(c->d_specialOut, 0, sizeof(u64) * c->nSpecial, c->stream));
  }
  if (c->histDirty && c->d_pairAcc) HIPCHK(c, hipMemsetAsync(c->d_pairAcc, 0, sizeof(u64) * 2 * (size_t)std::max<int64_t>(c->nRefs, 1), c->stream));
  c->sideUsed = false; c->pairUsed = false;
  c->histDirty = true; c->tileSumsValid = true;
  c->lastShareInfo = nullptr;
  return GTX_OK;
}

// share: the context is a group member -- finalize its classes only, d_hits receives its regions in the group's compact order
static int count_end(gtx_ctx *c, void *d_hits, bool share = false, bool scatter = false, bool hist32 = false)
{
  gtx::FinalizeShare fs = {c->d_shareTiles, c->nShareTiles, c->d_shareRegions, c->nShareRegions, scatter};
  if (++c->chainEpoch == 0) {                       // (after 2^32 calls: the flags start over)
    HIPCHK(c, hipMemsetAsync(c->d_chainFlags, 0, sizeof(unsigned) * 8 * (gtx::scan_tiles(c->histLen) + 2), c->stream));
    c->chainEpoch = 1; c->chainDraws = 0;
  }
  HIPCHK(c, gtx::launch_finalize(c->d_histA, c->d_histB, c->histLen, c->d_partA, c->d_partB, c->tileSumsValid, c->d_prefA, c->d_prefB,
                                 c->d_posE, c->d_posS, c->d_classBase, c->nRefs, (u64 *)d_hits, c->d_info + (c->infoCur ^ 1), c->stream,
                                 share ? &fs : nullptr, c->d_chainFlags, c->chainEpoch, c->d_info + c->infoCur, &c->chainDraws, hist32));
  { int rc = merge_end(c, d_hits); if (rc) return rc; }
  c->histDirty = false;
  c->infoCur ^= 1;                                // the block just used stays readable until the call after next
  return GTX_OK;
}

int gtx_count_device(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, uint32_t flags, void *d_hits)
{
  if (!c) return GTX_E_ARG;
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtx_count_device: gtx_set_refs has not been called");
  if (n < 0 || (n > 0 && !d_reads) || (c->nRefs > 0 && !d_hits)) return fail(c, GTX_E_ARG, "gtx_count_device: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = count_begin(c); if (rc) return rc;
  c->profThis = c->prof && (c->profEvery <= 1 || (c->profSeq++ % c->profEvery) == 0);
  if (c->profThis) { c->ev = c->evRing[c->profCalls % gtx_ctx::kProfSlots]; HIPCHK(c, hipEventRecord(c->ev[1], c->stream)); }
  rc = merge_prepare(c, flags, 0); if (rc) return rc;
  // GTX_CHECK_SORTED is answered by the streaming kernel (exact for any order; only it looks at the order of the reads)
  const bool streaming = (flags & (GTX_READS_SORTED | GTX_CHECK_SORTED)) != 0;
  if (!streaming) c->tileSumsValid = false;
  bool h32 = false;
  if (streaming) {
    const gtx::CountArgs a = count_args(c, flags, n, 0, nullptr, false, d_weights == nullptr);
    h32 = a.hist32 != 0 && n > 0;
    HIPCHK(c, gtx::launch_count(d_reads, d_weights, n, a, true, c->stream));
  }
  else { rc = launch_unsorted(c, d_reads, d_weights, n, count_args(c, flags, n)); if (rc) return rc; }
  rc = merge_batch(c, d_reads, d_weights, n); if (rc) return rc;
  rc = pairs_batch(c, d_reads, d_weights, n); if (rc) return rc;
  if (c->profThis) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  rc = count_end(c, d_hits, false, false, h32); if (rc) return rc;
  if (c->profThis) { if (c->profEvery <= 1) HIPCHK(c, hipEventRecord(c->ev[3], c->stream)); c->profCalls++; }
  return GTX_OK;
}

static void info_out(const gtx::DevInfo &d, gtx_count_info *o, int64_t base)
{
  o->first_unsorted = d.first_unsorted == INT64_MAX ? -1 : d.first_unsorted + base;
  o->n_no_class = d.n_no_class; o->n_degenerate = d.n_degenerate;
  o->first_degenerate = d.first_degenerate == INT64_MAX ? -1 : d.first_degenerate + base;
  o->n_unplaced = d.n_unplaced;
}

// a kernel of the call gave up a bounded wait (DevInfo::fault): its result vector is void
static int fault_check(gtx_ctx *c, const gtx::DevInfo &d)
{
  return d.fault ? fail(c, GTX_E_HIP, "the finalize step gave up waiting for a tile sum (finalize_scan_chained_kernel): the result of this call is void; GTX_CHAINED_SCAN=0 takes the two-launch finalize") : GTX_OK;
}

int gtx_last_info(gtx_ctx *c, gtx_count_info *info)
{
  if (!c || !info) return GTX_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(&c->h_info[0], c->d_info + (c->infoCur ^ 1), sizeof(gtx::DevInfo), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  info_out(c->h_info[0], info, 0);
  return fault_check(c, c->h_info[0]);
}

} // extern "C" (templates below)

// ---------------------------------------------------------------------------------------------
// host buffers -> device, double-buffered
// ---------------------------------------------------------------------------------------------
// A batch of the caller's reads travels: [pageable memory -> pinned slot (host threads)] -> device slot (DMA on the copy
// stream) -> kernels (the context's stream).  Two slots of each kind alternate, so the copy of batch i+1 -- host part and
// DMA -- runs under the kernels of batch i; the only host waits are for a slot to come free.  Memory the caller obtained
// from gtx_host_alloc (or any other page-locked memory HIP knows) skips the pinned slot: the DMA reads it directly.
static bool is_pinned(const void *p)
{
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

static void parallel_copy(char *dst, const char *src, size_t bytes, int threads)
{
  const int T = (int)std::min<size_t>((size_t)threads, bytes / (4u << 20) + 1);
  if (T <= 1) { memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back([=] { const size_t b0 = bytes * (size_t)t / T, b1 = bytes * (size_t)(t + 1) / T; memcpy(dst + b0, src + b0, b1 - b0); });
  memcpy(dst, src, bytes / T);
  for (auto &x : th) x.join();
}

static int stage_reserve(gtx_ctx *c, size_t nReads, bool weights, bool pinnedSlots)
{
  if (nReads > c->capStage || (weights && nReads > c->capStageW) || (pinnedSlots && nReads * 16 > c->capPin)) {
    // growing: nothing may still be in flight on the old buffers
    HIPCHK(c, hipStreamSynchronize(c->copyStream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->slotBusy[0] = c->slotBusy[1] = false;
  }
  if (nReads > c->capStage) {
    for (int k = 0; k < 2; k++) { dfree(c->d_stage[k]); HIPCHK(c, hipMalloc(&c->d_stage[k], nReads * 12)); }
    c->capStage = nReads;
  }
  if (weights && nReads > c->capStageW) {
    for (int k = 0; k < 2; k++) { dfree(c->d_stageW[k]); HIPCHK(c, hipMalloc(&c->d_stageW[k], nReads * 4)); }
    c->capStageW = nReads;
  }
  if (pinnedSlots && nReads * 16 > c->capPin) {
    for (int k = 0; k < 2; k++) { if (c->h_pin[k]) { (void)hipHostFree(c->h_pin[k]); c->h_pin[k] = nullptr; } HIPCHK(c, hipHostMalloc((void **)&c->h_pin[k], nReads * 16)); }
    c->capPin = nReads * 16;
  }
  return GTX_OK;
}

// Streams reads[0..n) (and weights) through the device in batches; launch(d_reads, d_weights, cnt, off) enqueues the
// kernels of one batch on the context's stream.  Returns with everything enqueued; the caller's buffers are free again
// when the function returns unless they are page-locked, in which case they must stay untouched until the context's next
// host-buffer call, gtx_*_end or gtx_sync has returned.
template <class Launch>
static int stage_batches(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, Launch launch)
{
  if (c->directPending) { HIPCHK(c, hipStreamSynchronize(c->copyStream)); c->directPending = false; }   // the previous call's page-locked source is free now
  if (n <= 0) return GTX_OK;
  const int64_t batch = c->batchReads;
  const bool direct = is_pinned(reads) && (!weights || is_pinned(weights));
  c->directPending = direct;
  int rc = stage_reserve(c, (size_t)std::min<int64_t>(n, batch), weights != nullptr, !direct); if (rc) return rc;
  for (int64_t off = 0; off < n; off += batch) {
    const int64_t cnt = std::min(batch, n - off);
    const int slot = (int)(c->stageSeq++ & 1);
    if (c->slotBusy[slot]) {
      // the pinned slot is free once its DMA is done; the device slot once the kernels that read it are
      if (!direct) HIPCHK(c, hipEventSynchronize(c->evCopied[slot]));
      HIPCHK(c, hipStreamWaitEvent(c->copyStream, c->evConsumed[slot], 0));
    }
    const char *srcR = (const char *)(reads + 3 * off), *srcW = (const char *)(weights ? weights + off : nullptr);
    if (!direct) {
      parallel_copy(c->h_pin[slot], srcR, (size_t)cnt * 12, c->copyThreads);
      if (weights) parallel_copy(c->h_pin[slot] + (size_t)cnt * 12, srcW, (size_t)cnt * 4, c->copyThreads);
      srcR = c->h_pin[slot]; srcW = c->h_pin[slot] + (size_t)cnt * 12;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_stage[slot], srcR, (size_t)cnt * 12, hipMemcpyHostToDevice, c->copyStream));
    if (weights) HIPCHK(c, hipMemcpyAsync(c->d_stageW[slot], srcW, (size_t)cnt * 4, hipMemcpyHostToDevice, c->copyStream));
    HIPCHK(c, hipEventRecord(c->evCopied[slot], c->copyStream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evCopied[slot], 0));
    rc = launch(c->d_stage[slot], weights ? c->d_stageW[slot] : nullptr, cnt, off); if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->evConsumed[slot], c->stream));
    c->slotBusy[slot] = true;
  }
  return GTX_OK;
}

extern "C" {

static int ensure_out(gtx_ctx *c, size_t n)
{
  if (n > c->capOut) { dfree(c->d_out); c->capOut = 0; HIPCHK(c, hipMalloc(&c->d_out, (n + 1) * sizeof(u64))); c->capOut = n; }
  return GTX_OK;
}

// Host buffers: the reads are streamed through the device in batches (the histograms simply keep
// accumulating across batches), so N is bounded by host memory only -- the analogue of the
// reference never holding the query set in memory (genomic_intervals.cpp:3855-3861).
int gtx_count_begin(gtx_ctx *c)
{
  if (!c) return GTX_E_ARG;
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtx_count_begin: gtx_set_refs has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = count_begin(c); if (rc) return rc;
  c->streamSeen = 0; c->streamOpen = true; c->streamLast[0] = c->streamLast[1] = INT32_MIN; c->seamUnsorted = INT64_MAX;
  return GTX_OK;
}

int gtx_count_add(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags)
{
  if (!c) return GTX_E_ARG;
  if (!c->streamOpen) return fail(c, GTX_E_STATE, "gtx_count_add: gtx_count_begin has not been called");
  if (n < 0 || (n > 0 && !reads)) return fail(c, GTX_E_ARG, "gtx_count_add: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const bool streaming = (flags & (GTX_READS_SORTED | GTX_CHECK_SORTED)) != 0;
  if ((flags & GTX_CHECK_SORTED) && n > 0) {
    // order across the seams between batches (inside a batch the kernel checks): host-side, two reads per seam
    const int64_t batch = c->batchReads;
    for (int64_t off = 0; off < n; off += batch) {
      const int32_t *q = reads + 3 * off;
      const int32_t pc = off ? q[-3] : c->streamLast[0], ps = off ? q[-2] : c->streamLast[1];
      if ((c->streamSeen + off) > 0 && (q[0] < pc || (q[0] == pc && q[1] < ps)) && c->seamUnsorted == INT64_MAX) c->seamUnsorted = c->streamSeen + off;
    }
  }
  if (n > 0) { c->streamLast[0] = reads[3 * (n - 1)]; c->streamLast[1] = reads[3 * (n - 1) + 1]; }
  const int64_t seen = c->streamSeen;
  int rc = merge_prepare(c, flags, 0); if (rc) return rc;
  rc = stage_batches(c, reads, weights, n, [&](const void *dR, const int *dW, int64_t cnt, int64_t off) -> int {
    // indices in the info block are positions in the whole stream (indexBase); the block accumulates over the batches
    if (!streaming) c->tileSumsValid = false;
    if (streaming) HIPCHK(c, gtx::launch_count(dR, dW, cnt, count_args(c, flags, cnt, seen + off), true, c->stream));
    else { int rcu = launch_unsorted(c, dR, dW, cnt, count_args(c, flags, cnt, seen + off)); if (rcu) return rcu; }
    { int rcp = pairs_batch(c, dR, dW, cnt); if (rcp) return rcp; }
    return merge_batch(c, dR, dW, cnt);
  });
  if (rc) return rc;
  c->streamSeen += n;
  return GTX_OK;
}

// closes the open stream call: finalize into the context's own output vector in HBM (enqueued, not waited for)
// share != 0 (a group member with gtxi_set_share): only its classes are finalized, *d_out = its piece of the group's compact vector
int gtxi_count_finish(gtx_ctx *c, void **d_out, int share)
{
  if (!c->streamOpen) return fail(c, GTX_E_STATE, "gtx_count_end: gtx_count_begin has not been called");
  if (share && !c->shareOn) return fail(c, GTX_E_STATE, "gtxi_count_finish: no share set");
  if (share && (c->refBlocks || c->pairUsed)) return fail(c, GTX_E_STATE, "gtxi_count_finish: multi-interval regions are finalized over the whole vector");
  HIPCHK(c, hipSetDevice(c->device));
  c->streamOpen = false;
  int rc = ensure_out(c, (size_t)c->nRefs); if (rc) return rc;
  u64 *dst = share ? c->d_out + c->shareOffset : c->d_out;
  rc = count_end(c, dst, share != 0); if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(&c->h_info[0], c->d_info + (c->infoCur ^ 1), sizeof(gtx::DevInfo), hipMemcpyDeviceToHost, c->stream));
  *d_out = dst;
  return GTX_OK;
}

// after gtxi_*_finish and a wait for the stream: what the call observed
void gtxi_fetch_info(gtx_ctx *c, gtx_count_info *info)
{
  info_out(c->h_info[0], info, 0);
  if (c->seamUnsorted != INT64_MAX && (info->first_unsorted < 0 || c->seamUnsorted < info->first_unsorted)) info->first_unsorted = c->seamUnsorted;
}

int gtx_count_end(gtx_ctx *c, uint64_t *hits, gtx_count_info *info)
{
  if (!c) return GTX_E_ARG;
  if (c->streamOpen && c->nRefs > 0 && !hits) return fail(c, GTX_E_ARG, "gtx_count_end: null output");
  void *d = nullptr;
  int rc = gtxi_count_finish(c, &d, 0); if (rc) return rc;
  if (c->nRefs > 0) HIPCHK(c, hipMemcpyAsync(hits, d, sizeof(u64) * c->nRefs, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->copyStream));
  if (info) gtxi_fetch_info(c, info);
  return fault_check(c, c->h_info[0]);
}

int gtx_count(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags, uint64_t *hits, gtx_count_info *info)
{
  if (!c) return GTX_E_ARG;
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtx_count: gtx_set_refs has not been called");
  if (n < 0 || (n > 0 && !reads) || (c->nRefs > 0 && !hits)) return fail(c, GTX_E_ARG, "gtx_count: bad argument");
  int rc = gtx_count_begin(c); if (rc) return rc;
  rc = gtx_count_add(c, reads, weights, n, flags); if (rc) { c->streamOpen = false; return rc; }
  return gtx_count_end(c, hits, info);
}

// ---------------------------------------------------------------------------------------------
// count without -gaps over multi-interval regions (gtx_pairs.hip)
// ---------------------------------------------------------------------------------------------
} // extern "C"

// the envelopes of the regions `pick` selects, per class in the order of their starts, with the running and the per-64 maxima
// of their ends.  Which regions take part at all follows gtx_set_refs (placeholders and, under the bin index's rules, regions
// with start > stop or stop <= 0 never match); under the sorted merge's rules every region with a class does, inverted or not:
// the envelope test below is the merge's own (CalcDirection == 0, genomic_intervals.cpp:1225-1236).
template <class Pick>
static int build_pair_index(gtx_ctx *c, gtx_ctx::PairIdx *out, Pick pick)
{
  dfree(out->d_mem); *out = gtx_ctx::PairIdx();
  struct Item { int32_t cls, s, e, k; };
  std::vector<Item> it;
  for (int64_t k = 0; k < c->nRefs; k++) {
    const int32_t cl = c->h_refC[k], s = c->h_refS[k], e = c->h_refE[k];
    if (cl < 0 || !(c->mergeRefs || !(s > e || e <= 0)) || !pick(k)) continue;
    it.push_back({cl, s, e, (int32_t)k});
  }
  std::sort(it.begin(), it.end(), [](const Item &a, const Item &b) { return a.cls != b.cls ? a.cls < b.cls : (a.s != b.s ? a.s < b.s : a.k < b.k); });
  const size_t n = it.size(), nb = (n + 63) / 64, nc = (size_t)c->nClasses;
  std::vector<int32_t> mem(nc + 1 + 4 * n + nb + 1, 0);
  int32_t *seg = mem.data(), *st = seg + nc + 1, *en = st + n, *pm = en + n, *bm = pm + n, *id = bm + nb;
  for (size_t i = 0; i < n; i++) seg[it[i].cls + 1]++;
  for (size_t cl = 0; cl < nc; cl++) seg[cl + 1] += seg[cl];
  for (size_t i = 0; i < nb; i++) bm[i] = INT32_MIN;
  for (size_t i = 0; i < n; i++) {
    st[i] = it[i].s; en[i] = it[i].e; id[i] = it[i].k;
    pm[i] = (i > 0 && it[i - 1].cls == it[i].cls) ? std::max(pm[i - 1], it[i].e) : it[i].e;
    bm[i >> 6] = std::max(bm[i >> 6], it[i].e);
  }
  HIPCHK(c, hipMalloc(&out->d_mem, sizeof(int32_t) * mem.size()));
  HIPCHK(c, hipMemcpy(out->d_mem, mem.data(), sizeof(int32_t) * mem.size(), hipMemcpyHostToDevice));
  int *d = out->d_mem;
  out->ix = gtx::PairIndex{d, d + nc + 1, d + nc + 1 + n, d + nc + 1 + 2 * n, d + nc + 1 + 3 * n, d + nc + 1 + 3 * n + nb, c->nClasses};
  out->n = (int)n; out->built = true;
  return GTX_OK;
}

static int pair_acc(gtx_ctx *c)
{
  if (c->d_pairAcc) return GTX_OK;
  const size_t bytes = sizeof(u64) * 2 * (size_t)std::max<int64_t>(c->nRefs, 1);
  HIPCHK(c, hipMalloc(&c->d_pairAcc, bytes));
  HIPCHK(c, hipMemset(c->d_pairAcc, 0, bytes));
  return GTX_OK;
}

// {start, stop} lists as the kernels need them: starts and stops both non-decreasing (sorted, disjoint intervals are)
static bool blocks_monotone(const int32_t *b, int64_t cnt)
{
  for (int64_t j = 1; j < cnt; j++) if (b[2 * j] < b[2 * j - 2] || b[2 * j + 1] < b[2 * j - 1]) return false;
  return true;
}

extern "C" {

int gtx_set_ref_blocks(gtx_ctx *c, const int64_t *first, const int32_t *blocks)
{
  if (!c) return GTX_E_ARG;
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtx_set_ref_blocks: gtx_set_refs has not been called");
  if (c->streamOpen || c->covOpen) return fail(c, GTX_E_STATE, "gtx_set_ref_blocks: a call is open");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  dfree(c->pairMulti.d_mem); dfree(c->pairAll.d_mem); dfree(c->d_blkOf); dfree(c->d_blkIv);
  c->pairMulti = gtx_ctx::PairIdx(); c->pairAll = gtx_ctx::PairIdx(); c->refBlocks = false;
  if (!first) return GTX_OK;                                   // back to single-interval regions
  const int64_t m = c->nRefs;
  if (first[0] != 0) return fail(c, GTX_E_ARG, "gtx_set_ref_blocks: first[0] must be 0");
  if (first[m] >= INT32_MAX || (first[m] > 0 && !blocks)) return fail(c, GTX_E_ARG, "gtx_set_ref_blocks: bad argument");
  std::vector<int2> blkOf((size_t)std::max<int64_t>(m, 1), make_int2(0, 0)), iv;
  for (int64_t k = 0; k < m; k++) {
    const int64_t cnt = first[k + 1] - first[k];
    if (cnt < 1) return fail(c, GTX_E_ARG, "gtx_set_ref_blocks: every region has at least one interval");
    const int32_t *b = blocks + 2 * first[k];
    if (c->h_refC[k] >= 0 && (b[0] != c->h_refS[k] || b[2 * cnt - 1] != c->h_refE[k]))
      return fail(c, GTX_E_ARG, "gtx_set_ref_blocks: a region's triple must be its envelope (first interval's start, last interval's stop)");
    if (cnt == 1 || c->h_refC[k] < 0) continue;
    if (!blocks_monotone(b, cnt)) return fail(c, GTX_E_RANGE, "gtx_set_ref_blocks: the intervals of a region must be sorted (starts and stops non-decreasing)");
    blkOf[k] = make_int2((int)iv.size(), (int)cnt);
    for (int64_t j = 0; j < cnt; j++) iv.push_back(make_int2(b[2 * j], b[2 * j + 1]));
  }
  if (iv.empty()) return GTX_OK;                               // no multi-interval region: nothing to correct
  HIPCHK(c, hipMalloc(&c->d_blkOf, sizeof(int2) * blkOf.size()));
  HIPCHK(c, hipMalloc(&c->d_blkIv, sizeof(int2) * iv.size()));
  HIPCHK(c, hipMemcpy(c->d_blkOf, blkOf.data(), sizeof(int2) * blkOf.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_blkIv, iv.data(), sizeof(int2) * iv.size(), hipMemcpyHostToDevice));
  int rc = build_pair_index(c, &c->pairMulti, [&](int64_t k) { return blkOf[k].y > 0; }); if (rc) return rc;
  rc = pair_acc(c); if (rc) return rc;
  c->refBlocks = true;
  return GTX_OK;
}

int gtx_count_add_regions(gtx_ctx *c, const int32_t *env, const int32_t *weights, const int64_t *first, const int32_t *blocks, int64_t n)
{
  if (!c) return GTX_E_ARG;
  if (!c->streamOpen) return fail(c, GTX_E_STATE, "gtx_count_add_regions: gtx_count_begin has not been called");
  if (n < 0 || (n > 0 && (!env || !first || !blocks))) return fail(c, GTX_E_ARG, "gtx_count_add_regions: bad argument");
  if (n == 0) return GTX_OK;
  if (first[0] != 0 || first[n] >= INT32_MAX) return fail(c, GTX_E_ARG, "gtx_count_add_regions: bad interval lists");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<int4> q((size_t)n); std::vector<int2> qb((size_t)n), iv((size_t)first[n]);
  for (int64_t i = 0; i < n; i++) {
    const int64_t cnt = first[i + 1] - first[i];
    if (cnt < 1) return fail(c, GTX_E_ARG, "gtx_count_add_regions: every region has at least one interval");
    const int32_t *b = blocks + 2 * first[i];
    if (b[0] != env[3 * i + 1] || b[2 * cnt - 1] != env[3 * i + 2]) return fail(c, GTX_E_ARG, "gtx_count_add_regions: a region's triple must be its envelope");
    if (!blocks_monotone(b, cnt)) return fail(c, GTX_E_RANGE, "gtx_count_add_regions: the intervals of a region must be sorted (starts and stops non-decreasing)");
    q[i] = make_int4(env[3 * i], env[3 * i + 1], env[3 * i + 2], weights ? weights[i] : 1);
    qb[i] = make_int2((int)first[i], (int)cnt);
    for (int64_t j = 0; j < cnt; j++) iv[first[i] + j] = make_int2(b[2 * j], b[2 * j + 1]);
  }
  if (!c->pairAll.built) { int rc = build_pair_index(c, &c->pairAll, [](int64_t) { return true; }); if (rc) return rc; }
  { int rc = pair_acc(c); if (rc) return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));                  // (the kernel of the previous call may still read the query buffers)
  if ((size_t)n > c->capPairQ) {
    dfree(c->d_pairQ); dfree(c->d_pairQBlk); c->capPairQ = 0;
    HIPCHK(c, hipMalloc(&c->d_pairQ, sizeof(int4) * (size_t)n)); HIPCHK(c, hipMalloc(&c->d_pairQBlk, sizeof(int2) * (size_t)n));
    c->capPairQ = (size_t)n;
  }
  if (iv.size() > c->capPairIv) { dfree(c->d_pairQIv); c->capPairIv = 0; HIPCHK(c, hipMalloc(&c->d_pairQIv, sizeof(int2) * iv.size())); c->capPairIv = iv.size(); }
  HIPCHK(c, hipMemcpy(c->d_pairQ, q.data(), sizeof(int4) * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_pairQBlk, qb.data(), sizeof(int2) * (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_pairQIv, iv.data(), sizeof(int2) * iv.size(), hipMemcpyHostToDevice));
  HIPCHK(c, gtx::launch_pair_hit(c->d_pairQ, c->d_pairQBlk, c->d_pairQIv, n, c->pairAll.ix, gtx::RegionBlocks{c->refBlocks ? c->d_blkOf : nullptr, c->d_blkIv},
                                 c->d_pairAcc, c->stream));
  c->pairUsed = true;
  return GTX_OK;
}

int gtxi_pairs_on(gtx_ctx *c) { return c && (c->refBlocks || c->pairUsed) ? 1 : 0; }

// ---------------------------------------------------------------------------------------------
// coverage
// ---------------------------------------------------------------------------------------------
// bucket tables over ONE sorted array (the coverage thresholds): cuts every bucket_e_size() entries, never across classes; the slice
// a bucket keeps in LDS begins where the bucket does (a read ends at or behind its start) -- see the same for the two boundary
// arrays of the count path in gtx_set_refs_ex
static int cover_bucket_tables(gtx_ctx *c, const std::vector<int32_t> &sortedT, const std::vector<int32_t> &seg)
{
  const int nClasses = c->nClasses;
  const int kE = gtx::bucket_e_size(), kS = gtx::bucket_t_size();
  std::vector<int32_t> posHi, eLo, eHi, sLo, sHi, cls, clsStart(nClasses + 1, 0);
  for (int cl = 0; cl < nClasses; cl++) {
    clsStart[cl] = (int32_t)posHi.size();
    const int32_t s0 = seg[cl], s1 = seg[cl + 1];
    for (int32_t e0 = s0; e0 < s1; e0 += kE) {
      const int32_t e1 = (int32_t)std::min<int64_t>((int64_t)e0 + kE, s1);
      posHi.push_back(e1 == s1 ? INT32_MAX : sortedT[e1 - 1]);
      eLo.push_back(e0); eHi.push_back(e1); cls.push_back(cl);
      sLo.push_back(e0); sHi.push_back((int32_t)std::min<int64_t>((int64_t)e0 + kS, s1));
    }
  }
  clsStart[nClasses] = (int32_t)posHi.size();
  c->nBT = (int)posHi.size();
  if (c->nBT > 8192 || nClasses > 2048) c->nBT = 0;
  if (c->nBT == 0) return GTX_OK;
  const int kCells = 4096;
  int sh = 0;
  auto cellsAt = [&](int shift) {
    int64_t total = 0;
    for (int cl = 0; cl < nClasses; cl++) {
      const int b0 = clsStart[cl], b1 = clsStart[cl + 1];
      total += b1 - b0 <= 1 ? b1 - b0 : ((((int64_t)posHi[b1 - 2] - posHi[b0]) >> shift) + 1);
    }
    return total;
  };
  while (sh < 40 && cellsAt(sh) > kCells) sh++;
  std::vector<int32_t> clsCell(4 * (size_t)nClasses);
  std::vector<uint16_t> cellTab;
  for (int cl = 0; cl < nClasses; cl++) {
    const int b0 = clsStart[cl], b1 = clsStart[cl + 1];
    const int32_t lo = b1 - b0 <= 1 ? 0 : posHi[b0];
    const int64_t nc = b1 == b0 ? 0 : b1 - b0 == 1 ? 1 : ((((int64_t)posHi[b1 - 2] - lo) >> sh) + 1);
    clsCell[4 * cl] = (int32_t)cellTab.size(); clsCell[4 * cl + 1] = lo; clsCell[4 * cl + 2] = (int32_t)nc; clsCell[4 * cl + 3] = b0;
    int b = b0;
    for (int64_t k = 0; k < nc; k++) {
      const int64_t first = (int64_t)lo + (k << sh);
      while (b < b1 - 1 && (int64_t)posHi[b] < first) b++;
      cellTab.push_back((uint16_t)(b - b0));
    }
  }
  c->nCellsT = (int)cellTab.size(); c->cellShiftT = sh;
  if (!gtx::bucket_tables_fit(nClasses, c->nBT, c->nCellsT)) { c->nBT = 0; return GTX_OK; }
  cellTab.push_back(0);
  std::vector<int32_t> all;
  for (auto *v : {&posHi, &eLo, &eHi, &sLo, &sHi, &cls, &clsStart}) all.insert(all.end(), v->begin(), v->end());
  HIPCHK(c, hipMalloc(&c->d_bktT, sizeof(int32_t) * all.size()));
  HIPCHK(c, hipMemcpy(c->d_bktT, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc(&c->d_clsCellT, sizeof(int32_t) * clsCell.size() + 16));
  HIPCHK(c, hipMemcpy(c->d_clsCellT, clsCell.data(), sizeof(int32_t) * clsCell.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc(&c->d_cellTabT, sizeof(uint16_t) * cellTab.size()));
  HIPCHK(c, hipMemcpy(c->d_cellTabT, cellTab.data(), sizeof(uint16_t) * cellTab.size(), hipMemcpyHostToDevice));
  return GTX_OK;
}

static int cover_prepare(gtx_ctx *c)
{
  if (c->covReady) return GTX_OK;
  // the thresholds of every region that takes part (the rule of gtx_set_refs_ex): E_k and S_k - 1, sorted by (class, value)
  const int64_t m = c->nRefs;
  struct Item { int32_t cls, val, k2; };                      // k2 = 2 * region + (0: the E threshold, 1: the S - 1 threshold)
  std::vector<Item> it;
  it.reserve((size_t)2 * c->nValid);
  for (int64_t k = 0; k < m; k++) {
    const int32_t cl = c->h_refC[k], s = c->h_refS[k], e = c->h_refE[k];
    const bool take = cl >= 0 && (c->mergeRefs ? (int64_t)s <= (int64_t)e + 1 : !(s > e || e <= 0));
    if (!take) continue;
    it.push_back({cl, e, (int32_t)(2 * k)}); it.push_back({cl, s - 1, (int32_t)(2 * k + 1)});
  }
  std::sort(it.begin(), it.end(), [](const Item &a, const Item &b) { return a.cls != b.cls ? a.cls < b.cls : (a.val != b.val ? a.val < b.val : a.k2 < b.k2); });
  const int64_t nt = (int64_t)it.size();
  std::vector<int32_t> sortedT(nt + 1), seg(c->nClasses + 1, 0), posTE(m > 0 ? m : 1, -1), posTS(m > 0 ? m : 1, -1), base(m > 0 ? m : 1, -1);
  for (int64_t i = 0; i < nt; i++) seg[it[i].cls + 1]++;
  for (int cl = 0; cl < c->nClasses; cl++) seg[cl + 1] += seg[cl];
  for (int64_t i = 0; i < nt; i++) {
    sortedT[i] = it[i].val;
    const int32_t k = it[i].k2 >> 1;
    ((it[i].k2 & 1) ? posTS : posTE)[k] = (int32_t)(i + it[i].cls);          // histogram slot = rank + class id, as in gtx_set_refs_ex
    base[k] = seg[it[i].cls] + it[i].cls - 1;
  }
  const int64_t nTop = (nt + 255) >> 8;
  std::vector<int32_t> topT(nTop + 1);
  for (int64_t i = 0; i < nTop; i++) topT[i] = sortedT[i << 8];
  c->histLenT = nt + c->nClasses;
  const int nTiles = gtx::scan_tiles(c->histLenT);
  HIPCHK(c, hipMalloc(&c->d_sortedT, sizeof(int32_t) * (nt + 1)));
  HIPCHK(c, hipMalloc(&c->d_segT, sizeof(int32_t) * (c->nClasses + 1)));
  HIPCHK(c, hipMalloc(&c->d_topT, sizeof(int32_t) * (nTop + 1)));
  HIPCHK(c, hipMalloc(&c->d_posTE, sizeof(int32_t) * (m + 1)));
  HIPCHK(c, hipMalloc(&c->d_posTS, sizeof(int32_t) * (m + 1)));
  HIPCHK(c, hipMalloc(&c->d_classBaseT, sizeof(int32_t) * (m + 1)));
  HIPCHK(c, hipMemcpy(c->d_sortedT, sortedT.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_segT, seg.data(), sizeof(int32_t) * (c->nClasses + 1), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_topT, topT.data(), sizeof(int32_t) * (nTop + 1), hipMemcpyHostToDevice));
  if (m > 0) {
    HIPCHK(c, hipMemcpy(c->d_posTE, posTE.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_posTS, posTS.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_classBaseT, base.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
  }
  for (int q = 0; q < 4; q++) {
    HIPCHK(c, hipMalloc(&c->d_cov[q], sizeof(u64) * c->histLenT));          // histograms
    HIPCHK(c, hipMalloc(&c->d_cov[4 + q], sizeof(u64) * (nTiles + 2)));     // tile sums
    HIPCHK(c, hipMalloc(&c->d_cov[8 + q], sizeof(u64) * c->histLenT));      // prefixes
    HIPCHK(c, hipMemset(c->d_cov[q], 0, sizeof(u64) * c->histLenT));
    HIPCHK(c, hipMemset(c->d_cov[4 + q], 0, sizeof(u64) * (nTiles + 2)));
  }
  { int rc = ref_columns(c); if (rc) return rc; }
  { int rc = cover_bucket_tables(c, sortedT, seg); if (rc) return rc; }
  { int rc = make_place_table(c, seg, sortedT, sortedT, c->nClasses, nt, &c->d_placeClsT, &c->d_placeRankT, &c->placeShiftT); if (rc) return rc; }
  c->covReady = true; c->covDirty = false;
  return GTX_OK;
}

static gtx::CoverArgs cover_args(gtx_ctx *c, int64_t nReads, int64_t indexBase = 0, uint32_t flags = 0)
{
  gtx::CoverArgs a;
  a.indexBase = indexBase;
  a.sortedT = c->d_sortedT; a.segStartT = c->d_segT; a.topT = c->d_topT;
  for (int q = 0; q < 4; q++) { a.hist[q] = c->d_cov[q]; a.part[q] = c->d_cov[4 + q]; }
  a.info = c->d_info + c->infoCur; a.nClasses = c->nClasses;
  const bool merge = (flags & GTX_ZERO_LENGTH_OK) && (flags & GTX_GAPS_FORMULA) && c->mergeRefs && c->d_side;
  a.side = merge ? c->d_side : nullptr; a.sideCount = merge ? c->d_sideCount : nullptr; a.sideCap = c->sideCap;
  { static const bool wf = !(getenv("GTX_WEIGHTED_FAST") && atoi(getenv("GTX_WEIGHTED_FAST")) == 0); a.wfast = wf ? 1 : 0; }
  int64_t nChunks = (nReads + 63) >> 6;
  // span per wave: as count_args, a little longer (the start of a span costs more here: 100 M reads: 0.55 ms at 16 chunks, 0.44 at 32,
  // 0.382 at 56, 0.374 at 64, 0.383 at 96)
  a.chunksPerWave = c->chunksPerWave > 0 ? c->chunksPerWave : (int)std::min<int64_t>(64, std::max<int64_t>(8, nChunks / 24576));
  a.chunksPerWave = (a.chunksPerWave + 3) / 4 * 4;
  a.sched = gtx::span_schedule(nChunks, a.chunksPerWave, 4, c->waveSlots);
  a.place.cls = c->d_placeClsT; a.place.rank = c->d_placeRankT; a.place.shift = c->placeShiftT;
  return a;
}

static int cover_begin(gtx_ctx *c)
{
  int rc = cover_prepare(c); if (rc) return rc;
  if (c->covDirty) {
    const int nTiles = gtx::scan_tiles(c->histLenT);
    for (int q = 0; q < 4; q++) {
      HIPCHK(c, hipMemsetAsync(c->d_cov[q], 0, sizeof(u64) * c->histLenT, c->stream));
      HIPCHK(c, hipMemsetAsync(c->d_cov[4 + q], 0, sizeof(u64) * (nTiles + 2), c->stream));
    }
  }
  if (c->covDirty || c->histDirty) {
    HIPCHK(c, hipMemcpyAsync(c->d_info + c->infoCur, &c->h_info[1], sizeof(gtx::DevInfo), hipMemcpyHostToDevice, c->stream));
    if (c->d_sideCount) {
      HIPCHK(c, hipMemsetAsync(c->d_sideCount, 0, sizeof(unsigned), c->stream));
      if (c->nSpecial) HIPCHK(c, hipMemsetAsync(c->d_specialOut, 0, sizeof(u64) * c->nSpecial, c->stream));
    }
  }
  c->sideUsed = false;
  c->covDirty = true; c->covTileSums = true;
  return GTX_OK;
}

// One batch of reads into the four coverage histograms: the streaming kernel (exact for any order, fast for sorted reads), or --
// for a batch the caller (GTX_READS_UNSORTED) or a sample of the host buffer says is in no particular order -- the partition path.
static int cover_launch(gtx_ctx *c, const void *dR, const int *dW, int64_t n, int64_t indexBase, uint32_t flags, bool unsorted)
{
  const gtx::CoverArgs cv = cover_args(c, n, indexBase, flags);
  if (unsorted && c->nBT > 0 && n >= c->bucketMinReads && n < (1ll << 31)) {
    const gtx::BucketPlan p = gtx::bucket_plan(n, c->nClasses, c->nBT, c->nCellsT, dW != nullptr);
    if (p.pairs < (1ull << 32)) {
      gtx::BucketWork w;
      { int rc = bucket_scratch(c, p, c->nBT, &w); if (rc) return rc; }
      const gtx::BucketTable t = bucket_table(c->d_bktT, c->nBT, c->d_clsCellT, c->d_cellTabT, c->nCellsT, c->cellShiftT);
      gtx::CountArgs a = {};                                        // what the partition pass reads of it
      a.nClasses = c->nClasses; a.zeroLenOk = 0; a.coverRule = 1; a.info = cv.info; a.indexBase = indexBase;
      a.side = cv.side; a.sideCount = cv.sideCount; a.sideCap = cv.sideCap;
      HIPCHK(c, gtx::launch_cover_bucketed(dR, dW, n, a, cv, t, w, p, c->stream));
      c->covTileSums = false;
      return GTX_OK;
    }
  }
  HIPCHK(c, gtx::launch_coverage(dR, dW, n, cv, c->stream));
  return GTX_OK;
}

// a host buffer's order, from 4096 adjacent pairs: a few descents are a few sorted runs (still the streaming kernel's business)
static bool host_reads_look_unsorted(const int32_t *tri, int64_t n)
{
  if (n < 2) return false;
  const int64_t stride = n > 4096 ? n / 4096 : 1;
  int descents = 0;
  for (int64_t i = 0; i + 1 < n; i += stride) {
    const int32_t *a = tri + 3 * i, *b = a + 3;
    if ((b[0] < a[0] || (b[0] == a[0] && b[1] < a[1])) && ++descents > 8) return true;
  }
  return false;
}

static int cover_end(gtx_ctx *c, void *d_cov_out)
{
  gtx::CoverGather g;
  for (int q = 0; q < 4; q++) { g.pref[q] = c->d_cov[8 + q]; g.part[q] = c->d_cov[4 + q]; }
  g.posTE = c->d_posTE; g.posTS = c->d_posTS; g.classBaseT = c->d_classBaseT; g.refS = c->d_refS; g.refE = c->d_refE;
  if (!c->covTileSums)                                               // (the partition path does not keep them)
    for (int q = 0; q < 4; q += 2) HIPCHK(c, gtx::launch_tile_sums(c->d_cov[q], c->d_cov[q + 1], c->histLenT, c->d_cov[4 + q], c->d_cov[5 + q], c->stream));
  HIPCHK(c, gtx::launch_coverage_finalize(cover_args(c, 0), c->histLenT, g, c->nRefs, (u64 *)d_cov_out, c->d_info + (c->infoCur ^ 1), c->stream));
  { int rc = merge_end(c, d_cov_out); if (rc) return rc; }
  c->covDirty = false;
  c->infoCur ^= 1;
  return GTX_OK;
}

int gtx_coverage_device(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, uint32_t flags, void *d_cov)
{
  if (!c) return GTX_E_ARG;
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtx_coverage_device: gtx_set_refs has not been called");
  if (n < 0 || (n > 0 && !d_reads) || (c->nRefs > 0 && !d_cov)) return fail(c, GTX_E_ARG, "gtx_coverage_device: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = cover_begin(c); if (rc) return rc;
  c->profThis = c->prof && (c->profEvery <= 1 || (c->profSeq++ % c->profEvery) == 0);
  if (c->profThis) { c->ev = c->evRing[c->profCalls % gtx_ctx::kProfSlots]; HIPCHK(c, hipEventRecord(c->ev[1], c->stream)); }
  if (flags & GTX_GAPS_FORMULA) { rc = merge_prepare(c, flags, 2); if (rc) return rc; }
  rc = cover_launch(c, d_reads, (const int *)d_weights, n, 0, flags, (flags & GTX_READS_UNSORTED) != 0); if (rc) return rc;
  rc = merge_batch(c, d_reads, d_weights, n); if (rc) return rc;
  if (c->profThis) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  rc = cover_end(c, d_cov); if (rc) return rc;
  if (c->profThis) { if (c->profEvery <= 1) HIPCHK(c, hipEventRecord(c->ev[3], c->stream)); c->profCalls++; }
  return GTX_OK;
}

int gtx_coverage_begin(gtx_ctx *c)
{
  if (!c) return GTX_E_ARG;
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtx_coverage_begin: gtx_set_refs has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = cover_begin(c); if (rc) return rc;
  c->streamSeen = 0; c->covOpen = true;
  return GTX_OK;
}

int gtx_coverage_add(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags)
{
  if (!c) return GTX_E_ARG;
  if (!c->covOpen) return fail(c, GTX_E_STATE, "gtx_coverage_add: gtx_coverage_begin has not been called");
  if (n < 0 || (n > 0 && !reads)) return fail(c, GTX_E_ARG, "gtx_coverage_add: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t seen = c->streamSeen;
  int rc = GTX_OK;
  if (flags & GTX_GAPS_FORMULA) { rc = merge_prepare(c, flags, 2); if (rc) return rc; }
  rc = stage_batches(c, reads, weights, n, [&](const void *dR, const int *dW, int64_t cnt, int64_t off) -> int {
    const bool unsorted = (flags & GTX_READS_UNSORTED) || (!(flags & GTX_READS_SORTED) && host_reads_look_unsorted(reads + 3 * off, cnt));
    { int rc2 = cover_launch(c, dR, dW, cnt, seen + off, flags, unsorted); if (rc2) return rc2; }
    return merge_batch(c, dR, dW, cnt);
  });
  if (rc) return rc;
  c->streamSeen += n;
  return GTX_OK;
}

int gtxi_coverage_finish(gtx_ctx *c, void **d_out)
{
  if (!c->covOpen) return fail(c, GTX_E_STATE, "gtx_coverage_end: gtx_coverage_begin has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  c->covOpen = false; c->seamUnsorted = INT64_MAX;
  int rc = ensure_out(c, (size_t)c->nRefs); if (rc) return rc;
  rc = cover_end(c, c->d_out); if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(&c->h_info[0], c->d_info + (c->infoCur ^ 1), sizeof(gtx::DevInfo), hipMemcpyDeviceToHost, c->stream));
  *d_out = c->d_out;
  return GTX_OK;
}

int gtx_coverage_end(gtx_ctx *c, uint64_t *cov, gtx_count_info *info)
{
  if (!c) return GTX_E_ARG;
  if (c->covOpen && c->nRefs > 0 && !cov) return fail(c, GTX_E_ARG, "gtx_coverage_end: null output");
  void *d = nullptr;
  int rc = gtxi_coverage_finish(c, &d); if (rc) return rc;
  if (c->nRefs > 0) HIPCHK(c, hipMemcpyAsync(cov, d, sizeof(u64) * c->nRefs, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->copyStream));
  if (info) gtxi_fetch_info(c, info);
  return GTX_OK;
}

int gtx_coverage(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags, uint64_t *cov, gtx_count_info *info)
{
  if (!c) return GTX_E_ARG;
  if (n < 0 || (n > 0 && !reads)) return fail(c, GTX_E_ARG, "gtx_coverage: bad argument");
  int rc = gtx_coverage_begin(c); if (rc) return rc;
  rc = gtx_coverage_add(c, reads, weights, n, flags); if (rc) { c->covOpen = false; return rc; }
  return gtx_coverage_end(c, cov, info);
}

// ---------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------
int64_t gtx_scan_n_windows(int64_t len, int64_t step, int64_t size)
{
  if (step <= 0 || size <= 0 || len < 0) return 0;
  int64_t n = len / step, comb = size / step;
  return n < comb ? 0 : n - comb + 1;
}

// ownTile > 0: also lay out the blocks of the owner-computes pass (ownTile windows each); *own receives its table
static int scan_prepare(gtx_ctx *c, const int32_t *classLen, int nClasses, int step, int size, const int64_t *classOff, gtx::ScanArgs *out,
                        int ownTile = 0, gtx::ScanOwn *own = nullptr)
{
  if (nClasses < 1 || !classLen || !classOff) return fail(c, GTX_E_ARG, "gtx_scan: bad class table");
  if (step <= 0 || size <= 0 || size % step) return fail(c, GTX_E_ARG, "gtx_scan: window size must be a positive multiple of window step");
  std::vector<long long> key;
  key.push_back(nClasses); key.push_back(step); key.push_back(size); key.push_back(ownTile);
  for (int i = 0; i < nClasses; i++) { key.push_back(classLen[i]); key.push_back(classOff[i]); }
  const size_t tabLen = (size_t)6 * nClasses + 3;
  if (key != c->scanKey) {
    std::vector<long long> tab(tabLen);
    long long *microOff = tab.data(), *nMicro = microOff + nClasses, *winOff = nMicro + nClasses, *outOff = winOff + nClasses + 1,
              *tileOff = outOff + nClasses, *blkOff = tileOff + nClasses + 1;
    long long mo = 0, wo = 0, to = 0, bo = 0;
    const long long tile = gtx::scan_window_tile();
    for (int i = 0; i < nClasses; i++) {
      long long nm = classLen[i] < 0 ? 0 : classLen[i] / step;
      long long nw = gtx_scan_n_windows(classLen[i] < 0 ? 0 : classLen[i], step, size);
      microOff[i] = mo; nMicro[i] = nm; winOff[i] = wo; outOff[i] = classOff[i]; tileOff[i] = to; blkOff[i] = bo;
      mo += nm; wo += nw; to += (nw + tile - 1) / tile;
      if (ownTile > 0) bo += (nw + ownTile - 1) / ownTile;
    }
    winOff[nClasses] = wo; tileOff[nClasses] = to; blkOff[nClasses] = bo;
    c->scanTotalTiles = to; c->scanOwnBlocks = bo;
    if (tabLen > c->capScanTab) { dfree(c->d_scanTab); c->capScanTab = 0; HIPCHK(c, hipMalloc(&c->d_scanTab, tabLen * sizeof(long long))); c->capScanTab = tabLen; }
    if ((size_t)mo + 1 > c->capMicro) { dfree(c->d_micro); c->capMicro = 0; HIPCHK(c, hipMalloc(&c->d_micro, ((size_t)mo + 3) * sizeof(u64))); c->capMicro = (size_t)mo + 1; }
    if ((size_t)(2 * bo + 2) > c->capScanBounds) { dfree(c->d_scanBounds); c->capScanBounds = 0; HIPCHK(c, hipMalloc(&c->d_scanBounds, (size_t)(2 * bo + 2) * sizeof(long long))); c->capScanBounds = (size_t)(2 * bo + 2); }
    if (!c->d_scanFlag) HIPCHK(c, hipMalloc(&c->d_scanFlag, sizeof(int)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->d_scanTab, tab.data(), tabLen * sizeof(long long), hipMemcpyHostToDevice));
    c->scanKey = key; c->scanTotalMicro = mo; c->scanTotalWindows = wo;
  }
  out->micro = c->d_micro; out->microOff = c->d_scanTab; out->nMicro = c->d_scanTab + nClasses;
  out->winOff = c->d_scanTab + 2 * nClasses; out->outOff = c->d_scanTab + 3 * nClasses + 1; out->tileOff = c->d_scanTab + 4 * nClasses + 1;
  out->nClasses = nClasses; out->winStep = step; out->comb = size / step;
  out->winStepInv = step > 1 ? (unsigned)((1ull << 32) / (unsigned)step) : 0;
  if (own) { own->blkOff = c->d_scanTab + 5 * nClasses + 2; own->tile = ownTile; own->totalBlocks = c->scanOwnBlocks; own->bounds = c->d_scanBounds; own->flag = c->d_scanFlag; }
  return GTX_OK;
}

// Bucket tables over the POSITIONS of a scan geometry, for reads in no particular order (gtx_bucket.hip: bucket_scanhist_kernel):
// a bucket is a run of consecutive micro-windows of one class -- at most ~2000 buckets in all, at least 16 k micro-windows each --
// and is counted in parts of scan_part_bins() micro-windows.  Built when the geometry (or weighted / not) changes.
static int scan_bucket_tables(gtx_ctx *c, const int32_t *classLen, int nClasses, int step, bool weighted)
{
  std::vector<long long> key = c->scanKey; key.push_back(weighted ? 1 : 0);
  if (key == c->scanBktKey) return GTX_OK;
  c->scanBktKey.clear(); c->nBS = 0; c->nScanParts = 0;
  dfree(c->d_bktS); dfree(c->d_clsCellS); dfree(c->d_cellTabS); dfree(c->d_scanParts);
  if (!c->d_scanInfo) HIPCHK(c, hipMalloc(&c->d_scanInfo, sizeof(gtx::DevInfo)));
  long long total = 0;
  for (int i = 0; i < nClasses; i++) total += classLen[i] < 0 ? 0 : classLen[i] / step;
  static const long long want = getenv("GTX_SCAN_BUCKETS") && atoll(getenv("GTX_SCAN_BUCKETS")) > 0 ? atoll(getenv("GTX_SCAN_BUCKETS")) : 2000;     // (100 M shuffled reads, -d 25: 500 -> 2.77 ms, 1000 -> 2.30, 2000 -> 2.15, 3000 -> 2.10)
  long long per = std::max<long long>(16384, (total + want - 1) / want);           // micro-windows per bucket
  std::vector<int32_t> posHi, eLo, eHi, sLo, sHi, cls, clsStart(nClasses + 1, 0);
  std::vector<gtx::ScanPart> parts;
  const int bins = gtx::scan_part_bins(weighted);
  if (per > bins) per = per / bins * bins;          // whole parts: every part of a bucket reads all of the bucket's chunks, a short last one as well
  for (int cl = 0; cl < nClasses; cl++) {
    clsStart[cl] = (int32_t)posHi.size();
    const long long nm = classLen[cl] < 0 ? 0 : classLen[cl] / step;
    for (long long f = 0; f < nm; f += per) {
      const long long g = std::min(f + per, nm);
      posHi.push_back(g == nm ? INT32_MAX : (int32_t)(g * step));                // positions <= g * step lie in micro-windows < g
      eLo.push_back((int32_t)f); eHi.push_back((int32_t)g); sLo.push_back(0); sHi.push_back(0); cls.push_back(cl);
      for (long long m = f; m < g; m += bins) parts.push_back({(int)posHi.size() - 1, (int)m, (int)std::min<long long>(bins, g - m), 0});
    }
  }
  clsStart[nClasses] = (int32_t)posHi.size();
  const int nB = (int)posHi.size();
  if (nB == 0 || nB > 4096 || nClasses > 2048 || total >= (1ll << 31)) { c->scanBktKey = key; return GTX_OK; }   // (nBS == 0: the general kernels serve)
  const int kCells = 4096;
  int sh = 0;
  auto cellsAt = [&](int shift) {
    int64_t tot = 0;
    for (int cl = 0; cl < nClasses; cl++) {
      const int b0 = clsStart[cl], b1 = clsStart[cl + 1];
      tot += b1 - b0 <= 1 ? b1 - b0 : ((((int64_t)posHi[b1 - 2] - posHi[b0]) >> shift) + 1);
    }
    return tot;
  };
  while (sh < 40 && cellsAt(sh) > kCells) sh++;
  std::vector<int32_t> clsCell(4 * (size_t)nClasses);
  std::vector<uint16_t> cellTab;
  for (int cl = 0; cl < nClasses; cl++) {
    const int b0 = clsStart[cl], b1 = clsStart[cl + 1];
    const int32_t lo = b1 - b0 <= 1 ? 0 : posHi[b0];
    const int64_t nc = b1 == b0 ? 0 : b1 - b0 == 1 ? 1 : ((((int64_t)posHi[b1 - 2] - lo) >> sh) + 1);
    clsCell[4 * cl] = (int32_t)cellTab.size(); clsCell[4 * cl + 1] = lo; clsCell[4 * cl + 2] = (int32_t)nc; clsCell[4 * cl + 3] = b0;
    int b = b0;
    for (int64_t k = 0; k < nc; k++) {
      const int64_t first = (int64_t)lo + (k << sh);
      while (b < b1 - 1 && (int64_t)posHi[b] < first) b++;
      cellTab.push_back((uint16_t)(b - b0));
    }
  }
  if (!gtx::bucket_tables_fit(nClasses, nB, (int)cellTab.size())) { c->scanBktKey = key; return GTX_OK; }
  c->nCellsS = (int)cellTab.size(); c->cellShiftS = sh;
  cellTab.push_back(0);
  std::vector<int32_t> all;
  for (auto *v : {&posHi, &eLo, &eHi, &sLo, &sHi, &cls, &clsStart}) all.insert(all.end(), v->begin(), v->end());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMalloc(&c->d_bktS, sizeof(int32_t) * all.size()));
  HIPCHK(c, hipMemcpy(c->d_bktS, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc(&c->d_clsCellS, sizeof(int32_t) * clsCell.size() + 16));
  HIPCHK(c, hipMemcpy(c->d_clsCellS, clsCell.data(), sizeof(int32_t) * clsCell.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc(&c->d_cellTabS, sizeof(uint16_t) * cellTab.size()));
  HIPCHK(c, hipMemcpy(c->d_cellTabS, cellTab.data(), sizeof(uint16_t) * cellTab.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc(&c->d_scanParts, sizeof(gtx::ScanPart) * parts.size()));
  HIPCHK(c, hipMemcpy(c->d_scanParts, parts.data(), sizeof(gtx::ScanPart) * parts.size(), hipMemcpyHostToDevice));
  c->nBS = nB; c->nScanParts = (int)parts.size(); c->scanBktKey = key;
  return GTX_OK;
}

// one batch of reads into the micro-window histogram (zeroed by the caller): the partition path for a batch in no particular order
// under the unsorted scanner's rule, the general kernel otherwise
// the partition path serves this call (reads in no order, the bin index's rules, a batch worth partitioning, tables that fit)
static int scan_takes_buckets(gtx_ctx *c, const int *dW, int64_t n, const gtx::ScanArgs &a, const int32_t *classLen, bool unsorted, bool *yes)
{
  *yes = false;
  if (!(unsorted && !a.sortedRule && n >= c->bucketMinReads && n < (1ll << 31))) return GTX_OK;
  int rc = scan_bucket_tables(c, classLen, a.nClasses, a.winStep, dW != nullptr); if (rc) return rc;
  *yes = c->nBS > 0 && gtx::bucket_plan(n, a.nClasses, c->nBS, c->nCellsS, dW != nullptr).pairs < (1ull << 32);
  return GTX_OK;
}

// d_windows (may be null): the partition path writes the windows itself (gtx::launch_scan_bucketed)
static int scan_hist_any(gtx_ctx *c, const void *dR, const int *dW, int64_t n, const gtx::ScanArgs &a, const int32_t *classLen, bool unsorted, u64 *d_windows = nullptr)
{
  if (unsorted && !a.sortedRule && n >= c->bucketMinReads && n < (1ll << 31)) {
    int rc = scan_bucket_tables(c, classLen, a.nClasses, a.winStep, dW != nullptr); if (rc) return rc;
    if (c->nBS > 0) {
      const gtx::BucketPlan p = gtx::bucket_plan(n, a.nClasses, c->nBS, c->nCellsS, dW != nullptr);
      if (p.pairs < (1ull << 32)) {
        gtx::BucketWork w;
        rc = bucket_scratch(c, p, c->nBS, &w); if (rc) return rc;
        const gtx::BucketTable t = bucket_table(c->d_bktS, c->nBS, c->d_clsCellS, c->d_cellTabS, c->nCellsS, c->cellShiftS);
        gtx::CountArgs ca = {};                                      // what the partition pass reads of it; its counts of dropped reads go nowhere
        ca.nClasses = a.nClasses; ca.zeroLenOk = 0; ca.coverRule = 1; ca.keyCenter = a.center; ca.info = c->d_scanInfo; ca.indexBase = 0;
        HIPCHK(c, gtx::launch_scan_bucketed(dR, dW, n, ca, a, t, w, p, c->d_scanParts, c->nScanParts, c->stream, d_windows));
        return GTX_OK;
      }
    }
  }
  if (d_windows) return fail(c, GTX_E_STATE, "gtx_scan: internal: the partition path was announced and not taken");
  HIPCHK(c, gtx::launch_scan_hist(dR, dW, n, a, c->stream));
  return GTX_OK;
}

// the scan of reads resident in HBM, enqueued: owner-computes when the caller says the reads are sorted (checked on the device; the
// general kernels follow as conditional launches and run only if the check fails), the general kernels otherwise
static int scan_launch(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, const int32_t *classLen, int32_t nClasses, int32_t step,
                       int32_t size, char prep, uint32_t flags, const int64_t *classOff, void *d_out, bool profile)
{
  static const bool ownOff = getenv("GTX_SCAN_OWN") && atoi(getenv("GTX_SCAN_OWN")) == 0;
  int64_t totalMicro = 0;
  for (int i = 0; i < nClasses; i++) totalMicro += classLen[i] < 0 ? 0 : classLen[i] / step;
  const int tile = ((flags & GTX_READS_SORTED) && prep == '1' && !ownOff && n > 0) ? gtx::scan_own_tile(n, totalMicro, size / step) : 0;
  gtx::ScanArgs a; gtx::ScanOwn own;
  int rc = scan_prepare(c, classLen, nClasses, step, size, classOff, &a, tile, &own); if (rc) return rc;
  a.center = prep == 'c'; a.sortedRule = (flags & GTX_ZERO_LENGTH_OK) ? 1 : 0;
  if (c->scanTotalWindows > 0 && !d_out) return fail(c, GTX_E_ARG, "gtx_scan: null output");
  const bool micro64 = d_weights != nullptr;
  const size_t microBytes = (size_t)c->scanTotalMicro * (micro64 ? 8 : 4);
  const int *runIf = nullptr;
  if (tile > 0 && own.totalBlocks > 0) {
    HIPCHK(c, hipMemsetAsync(c->d_scanFlag, 0, sizeof(int), c->stream));
    if (profile) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, gtx::launch_scan_own(d_reads, d_weights, n, a, own, (u64 *)d_out, c->stream));
    if (profile) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    runIf = c->d_scanFlag;
    if (microBytes) HIPCHK(c, gtx::launch_scan_zero(c->d_micro, (long long)microBytes, runIf, c->stream));
    HIPCHK(c, gtx::launch_scan_hist(d_reads, d_weights, n, a, c->stream, runIf));
  } else {
    // reads in no order through the partition path: its parts write the windows themselves (no micro-window array, no window pass)
    static const bool fuseOff = getenv("GTX_SCAN_FUSED") && atoi(getenv("GTX_SCAN_FUSED")) == 0;
    bool buckets = false;
    if (!fuseOff && a.comb <= gtx::scan_fused_max_comb() && c->scanTotalWindows > 0) {
      rc = scan_takes_buckets(c, (const int *)d_weights, n, a, classLen, (flags & GTX_READS_UNSORTED) != 0, &buckets); if (rc) return rc;
    }
    if (buckets) {
      if (profile) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
      rc = scan_hist_any(c, d_reads, (const int *)d_weights, n, a, classLen, true, (u64 *)d_out); if (rc) return rc;
      if (profile) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
      return GTX_OK;
    }
    if (microBytes) HIPCHK(c, hipMemsetAsync(c->d_micro, 0, microBytes, c->stream));
    if (profile) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    rc = scan_hist_any(c, d_reads, (const int *)d_weights, n, a, classLen, (flags & GTX_READS_UNSORTED) != 0); if (rc) return rc;
    if (profile) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  }
  HIPCHK(c, gtx::launch_scan_windows(c->d_micro, micro64, a, c->scanTotalTiles, (u64 *)d_out, c->stream, runIf));
  return GTX_OK;
}

// a host-buffer call's reads brought into ONE device buffer (pageable memory through the page-locked slots), so that the
// owner-computes scan can be a single launch over all of them
static int stage_resident(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, void **dR, int **dW)
{
  if (c->directPending) { HIPCHK(c, hipStreamSynchronize(c->copyStream)); c->directPending = false; }
  if ((size_t)n > c->capRes || (weights && (size_t)n > c->capResW)) {
    HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->copyStream));
    if ((size_t)n > c->capRes) { dfree(c->d_resReads); c->capRes = 0; HIPCHK(c, hipMalloc(&c->d_resReads, (size_t)n * 12)); c->capRes = (size_t)n; }
    if (weights && (size_t)n > c->capResW) { dfree(c->d_resWeights); c->capResW = 0; HIPCHK(c, hipMalloc(&c->d_resWeights, (size_t)n * 4)); c->capResW = (size_t)n; }
  }
  const bool direct = is_pinned(reads) && (!weights || is_pinned(weights));
  const int64_t batch = c->batchReads;
  const size_t slotBytes = (size_t)std::min<int64_t>(n, batch) * 16;
  if (!direct && slotBytes > c->capPin) {
    HIPCHK(c, hipStreamSynchronize(c->copyStream));
    for (int k = 0; k < 2; k++) { if (c->h_pin[k]) { (void)hipHostFree(c->h_pin[k]); c->h_pin[k] = nullptr; } HIPCHK(c, hipHostMalloc((void **)&c->h_pin[k], slotBytes)); }
    c->capPin = slotBytes; c->slotBusy[0] = c->slotBusy[1] = false;
  }
  // the kernels of the previous call (the context's stream) may still be reading the resident buffer
  HIPCHK(c, hipEventRecord(c->evRes[0], c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->copyStream, c->evRes[0], 0));
  for (int64_t off = 0; off < n; off += batch) {
    const int64_t cnt = std::min(batch, n - off);
    const char *srcR = (const char *)(reads + 3 * off), *srcW = (const char *)(weights ? weights + off : nullptr);
    int slot = -1;
    if (!direct) {
      slot = (int)(c->stageSeq++ & 1);
      if (c->slotBusy[slot]) HIPCHK(c, hipEventSynchronize(c->evCopied[slot]));      // the DMA out of this page-locked slot is done
      parallel_copy(c->h_pin[slot], srcR, (size_t)cnt * 12, c->copyThreads);
      if (weights) parallel_copy(c->h_pin[slot] + (size_t)cnt * 12, srcW, (size_t)cnt * 4, c->copyThreads);
      srcR = c->h_pin[slot]; srcW = c->h_pin[slot] + (size_t)cnt * 12;
    }
    HIPCHK(c, hipMemcpyAsync((char *)c->d_resReads + (size_t)off * 12, srcR, (size_t)cnt * 12, hipMemcpyHostToDevice, c->copyStream));
    if (weights) HIPCHK(c, hipMemcpyAsync(c->d_resWeights + off, srcW, (size_t)cnt * 4, hipMemcpyHostToDevice, c->copyStream));
    if (slot >= 0) { HIPCHK(c, hipEventRecord(c->evCopied[slot], c->copyStream)); c->slotBusy[slot] = true; }
  }
  c->directPending = direct;
  HIPCHK(c, hipEventRecord(c->evRes[1], c->copyStream));                              // everything has arrived before the scan starts
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->evRes[1], 0));
  *dR = c->d_resReads; *dW = weights ? c->d_resWeights : nullptr;
  return GTX_OK;
}

int gtx_scan_device(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, const int32_t *classLen, int32_t nClasses,
                    int32_t step, int32_t size, char prep, uint32_t flags, void *d_out, const int64_t *classOff)
{
  if (!c) return GTX_E_ARG;
  if (n < 0 || (n > 0 && !d_reads)) return fail(c, GTX_E_ARG, "gtx_scan_device: bad argument");
  if (prep != '1' && prep != 'c') return fail(c, GTX_E_ARG, "gtx_scan_device: preprocess operator must be '1' or 'c'");
  if (!d_weights && n >= (1ll << 32)) return fail(c, GTX_E_ARG, "gtx_scan_device: unweighted scans count in 32 bits per micro-window: at most 2^32-1 reads per call");
  HIPCHK(c, hipSetDevice(c->device));
  c->profThis = c->prof && (c->profEvery <= 1 || (c->profSeq++ % c->profEvery) == 0);
  if (c->profThis) c->ev = c->evRing[c->profCalls % gtx_ctx::kProfSlots];
  int rc = scan_launch(c, d_reads, d_weights, n, classLen, nClasses, step, size, prep, flags, classOff, d_out, c->profThis); if (rc) return rc;
  if (c->profThis) { if (c->profEvery <= 1) HIPCHK(c, hipEventRecord(c->ev[3], c->stream)); c->profCalls++; }
  return GTX_OK;
}

// the whole scan of host reads enqueued, result left in the context's own output vector in HBM
int gtxi_scan_enqueue(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *classLen, int32_t nClasses,
                      int32_t step, int32_t size, char prep, uint32_t flags, const int64_t *classOff, void **d_out, int64_t *extent_out)
{
  if (n < 0 || (n > 0 && !reads)) return fail(c, GTX_E_ARG, "gtx_scan: bad argument");
  if (prep != '1' && prep != 'c') return fail(c, GTX_E_ARG, "gtx_scan: preprocess operator must be '1' or 'c'");
  if (!weights && n >= (1ll << 32)) return fail(c, GTX_E_ARG, "gtx_scan: unweighted scans count in 32 bits per micro-window: at most 2^32-1 reads per call");
  if (nClasses < 1 || !classLen || !classOff) return fail(c, GTX_E_ARG, "gtx_scan: bad class table");
  if (step <= 0 || size <= 0 || size % step) return fail(c, GTX_E_ARG, "gtx_scan: window size must be a positive multiple of window step");
  HIPCHK(c, hipSetDevice(c->device));
  // output layout in the caller's buffer is given by class_offsets: find its extent
  int64_t extent = 0;
  for (int i = 0; i < nClasses; i++) extent = std::max<int64_t>(extent, classOff[i] + gtx_scan_n_windows(classLen[i] < 0 ? 0 : classLen[i], step, size));
  int rc = ensure_out(c, (size_t)extent); if (rc) return rc;
  if (extent > 0) HIPCHK(c, hipMemsetAsync(c->d_out, 0, (size_t)extent * sizeof(u64), c->stream));
  if ((flags & GTX_READS_SORTED) && prep == '1' && n > 0) {
    // sorted reads: all of them resident, one owner-computes launch (gtx_scanown.hip)
    void *dR = nullptr; int *dW = nullptr;
    rc = stage_resident(c, reads, weights, n, &dR, &dW); if (rc) return rc;
    rc = scan_launch(c, dR, dW, n, classLen, nClasses, step, size, prep, flags, classOff, c->d_out, false); if (rc) return rc;
  } else {
    gtx::ScanArgs a;
    rc = scan_prepare(c, classLen, nClasses, step, size, classOff, &a); if (rc) return rc;
    a.center = prep == 'c'; a.sortedRule = (flags & GTX_ZERO_LENGTH_OK) ? 1 : 0;
    const bool micro64 = weights != nullptr;
    if (c->scanTotalMicro > 0) HIPCHK(c, hipMemsetAsync(c->d_micro, 0, (size_t)c->scanTotalMicro * (micro64 ? 8 : 4), c->stream));
    rc = stage_batches(c, reads, weights, n, [&](const void *dR, const int *dW, int64_t cnt, int64_t off) -> int {
      const bool unsorted = (flags & GTX_READS_UNSORTED) || host_reads_look_unsorted(reads + 3 * off, cnt);
      return scan_hist_any(c, dR, dW, cnt, a, classLen, unsorted);
    });
    if (rc) return rc;
    HIPCHK(c, gtx::launch_scan_windows(c->d_micro, micro64, a, c->scanTotalTiles, c->d_out, c->stream));
  }
  *d_out = c->d_out; *extent_out = extent;
  return GTX_OK;
}

int gtx_scan(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *classLen, int32_t nClasses,
             int32_t step, int32_t size, char prep, uint32_t flags, uint64_t *out, const int64_t *classOff)
{
  if (!c) return GTX_E_ARG;
  void *d = nullptr; int64_t extent = 0;
  int rc = gtxi_scan_enqueue(c, reads, weights, n, classLen, nClasses, step, size, prep, flags, classOff, &d, &extent); if (rc) return rc;
  if (extent > 0 && !out) return fail(c, GTX_E_ARG, "gtx_scan: null output");
  if (extent > 0) HIPCHK(c, hipMemcpyAsync(out, d, (size_t)extent * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->copyStream));
  return GTX_OK;
}

// ---- the same scan fed as a stream (genomic_scans on an input of any size: host batches, or text tokenised on the device) ----
// The micro-window histogram accumulates over the batches (general kernels; batches in no order through the partition path), the
// sliding sums follow at the end.  The all-at-once calls above stay what a caller with every read in hand uses: they can take the
// owner-computes pass.
int gtx_scan_begin(gtx_ctx *c, const int32_t *classLen, int32_t nClasses, int32_t step, int32_t size, char prep, uint32_t flags, int weighted,
                   const int64_t *classOff)
{
  if (!c) return GTX_E_ARG;
  if (prep != '1' && prep != 'c') return fail(c, GTX_E_ARG, "gtx_scan_begin: preprocess operator must be '1' or 'c'");
  if (nClasses < 1 || !classLen || !classOff) return fail(c, GTX_E_ARG, "gtx_scan_begin: bad class table");
  if (step <= 0 || size <= 0 || size % step) return fail(c, GTX_E_ARG, "gtx_scan_begin: window size must be a positive multiple of window step");
  if (c->scan.open || c->streamOpen || c->covOpen) return fail(c, GTX_E_STATE, "gtx_scan_begin: a call is open");
  HIPCHK(c, hipSetDevice(c->device));
  int64_t extent = 0;
  for (int i = 0; i < nClasses; i++) extent = std::max<int64_t>(extent, classOff[i] + gtx_scan_n_windows(classLen[i] < 0 ? 0 : classLen[i], step, size));
  int rc = ensure_out(c, (size_t)extent); if (rc) return rc;
  if (extent > 0) HIPCHK(c, hipMemsetAsync(c->d_out, 0, (size_t)extent * sizeof(u64), c->stream));
  gtx_ctx::ScanOpen &s = c->scan;
  rc = scan_prepare(c, classLen, nClasses, step, size, classOff, &s.a); if (rc) return rc;
  s.a.center = prep == 'c'; s.a.sortedRule = (flags & GTX_ZERO_LENGTH_OK) ? 1 : 0;
  s.weighted = weighted != 0; s.classLen.assign(classLen, classLen + nClasses); s.extent = extent; s.prep = prep; s.flags = flags;
  if (c->scanTotalMicro > 0) HIPCHK(c, hipMemsetAsync(c->d_micro, 0, (size_t)c->scanTotalMicro * (s.weighted ? 8 : 4), c->stream));
  if (!s.d_labelSum) HIPCHK(c, hipMalloc(&s.d_labelSum, sizeof(unsigned long long)));
  HIPCHK(c, hipMemsetAsync(s.d_labelSum, 0, sizeof(unsigned long long), c->stream));
  s.open = true;
  return GTX_OK;
}

int gtx_scan_add(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags)
{
  if (!c) return GTX_E_ARG;
  if (!c->scan.open) return fail(c, GTX_E_STATE, "gtx_scan_add: gtx_scan_begin has not been called");
  if (n < 0 || (n > 0 && !reads) || (c->scan.weighted && n > 0 && !weights) || (!c->scan.weighted && weights)) return fail(c, GTX_E_ARG, "gtx_scan_add: bad argument (weights as announced to gtx_scan_begin)");
  HIPCHK(c, hipSetDevice(c->device));
  gtx_ctx::ScanOpen &s = c->scan;
  return stage_batches(c, reads, weights, n, [&](const void *dR, const int *dW, int64_t cnt, int64_t off) -> int {
    const bool unsorted = (flags & GTX_READS_UNSORTED) || host_reads_look_unsorted(reads + 3 * off, cnt);
    return scan_hist_any(c, dR, dW, cnt, s.a, s.classLen.data(), unsorted);
  });
}

int gtx_scan_end(gtx_ctx *c, uint64_t *out, int64_t *labelSum)
{
  if (!c) return GTX_E_ARG;
  if (!c->scan.open) return fail(c, GTX_E_STATE, "gtx_scan_end: gtx_scan_begin has not been called");
  gtx_ctx::ScanOpen &s = c->scan;
  s.open = false;
  if (s.extent > 0 && !out) return fail(c, GTX_E_ARG, "gtx_scan_end: null output");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gtx::launch_scan_windows(c->d_micro, s.weighted, s.a, c->scanTotalTiles, c->d_out, c->stream));
  if (s.extent > 0) HIPCHK(c, hipMemcpyAsync(out, c->d_out, (size_t)s.extent * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  unsigned long long sum = 0;
  HIPCHK(c, hipMemcpyAsync(&sum, s.d_labelSum, sizeof sum, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->copyStream));
  if (labelSum) *labelSum = (int64_t)sum;
  return GTX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// region text tokenised on the device (gtx_text.hip)
// ---------------------------------------------------------------------------------------------
enum TextMode { TEXT_COUNT, TEXT_COVER, TEXT_SCAN };
static int add_text(gtx_ctx *c, TextMode mode, const char *text, size_t bytes, int64_t nLines, const gtx_text_rules *r, uint32_t flags, int *ticket)
{
  const bool coverage = mode == TEXT_COVER;
  const char *who = mode == TEXT_SCAN ? "gtx_scan_add_text" : coverage ? "gtx_coverage_add_text" : "gtx_count_add_text";
  if (mode == TEXT_SCAN ? !c->scan.open : coverage ? !c->covOpen : !c->streamOpen) return fail(c, GTX_E_STATE, "gtx_*_add_text: no open count / coverage / scan call");
  if (mode == TEXT_SCAN && r && (r->max_label_value > 1) != c->scan.weighted) return fail(c, GTX_E_ARG, "gtx_scan_add_text: the rules' label weights do not match gtx_scan_begin's");
  if (!text || !r || !ticket || nLines < 0 || bytes >= (1ull << 32) - 4096 || nLines >= (1ll << 31) || r->n_chrom < 0 || (r->n_chrom > 0 && !r->chrom_names))
    { c->err = std::string(who) + ": bad argument"; return GTX_E_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const int slot = (int)(c->textSeq & 1);
  gtx_ctx::TextSlot &t = c->text[slot];
  *ticket = slot;
  if (!t.evParsed) {
    HIPCHK(c, hipEventCreateWithFlags(&t.evParsed, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&t.evConsumed, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&t.evCopied, hipEventDisableTiming));
    HIPCHK(c, hipMalloc(&t.d_flag, sizeof(int))); HIPCHK(c, hipHostMalloc((void **)&t.h_flag, sizeof(int))); HIPCHK(c, hipHostMalloc((void **)&t.h_seam, 4096));
    HIPCHK(c, hipMalloc(&t.d_sum, sizeof(unsigned long long))); HIPCHK(c, hipMemset(t.d_sum, 0, sizeof(unsigned long long)));
  }
  if (t.busy) { HIPCHK(c, hipEventSynchronize(t.evConsumed)); t.busy = false; }          // the block before last has been counted: its buffers are free
  c->textSeq++;
  if (nLines == 0 || bytes == 0) { *t.h_flag = 0; HIPCHK(c, hipEventRecord(t.evParsed, c->stream)); return GTX_OK; }
  const size_t nSeg = (bytes + 1023) / 1024;
  if (bytes + 64 > t.capText) { dfree(t.d_text); t.capText = 0; HIPCHK(c, hipMalloc(&t.d_text, bytes + (bytes >> 3) + 4096)); t.capText = bytes + (bytes >> 3) + 4096 - 64; }
  if (nSeg + 2 > t.capSeg) { dfree(t.d_seg); t.capSeg = 0; HIPCHK(c, hipMalloc(&t.d_seg, sizeof(unsigned) * (nSeg + (nSeg >> 3) + 16))); t.capSeg = nSeg + (nSeg >> 3) + 14; }
  if ((size_t)nLines > t.capLines) {
    dfree(t.d_nl); dfree(t.d_tri); dfree(t.d_w); t.capLines = 0;
    const size_t cap = (size_t)nLines + ((size_t)nLines >> 3) + 1024;
    HIPCHK(c, hipMalloc(&t.d_nl, sizeof(unsigned) * cap)); HIPCHK(c, hipMalloc(&t.d_tri, sizeof(int) * 3 * cap)); HIPCHK(c, hipMalloc(&t.d_w, sizeof(int) * cap));
    t.capLines = cap;
  }
  // the names' tables: per set of names (rebuilt when they change)
  {
    std::string blob; std::vector<int32_t> table; unsigned mask = 0;
    size_t total = 0; for (int i = 0; i < r->n_chrom; i++) total += strlen(r->chrom_names[i]) + 1;
    std::string key; key.reserve(total);
    for (int i = 0; i < r->n_chrom; i++) { key += r->chrom_names[i]; key += '\n'; }
    if (key != c->textBlob || !c->d_textTable) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      gtxtext::build_tables(*r, &table, &mask, &blob);
      dfree(c->d_textTable); dfree(c->d_textNames);
      HIPCHK(c, hipMalloc(&c->d_textTable, sizeof(int32_t) * table.size()));
      HIPCHK(c, hipMalloc(&c->d_textNames, blob.size() + 4096 * 2 + 16));
      HIPCHK(c, hipMemcpy(c->d_textTable, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice));
      if (!blob.empty()) HIPCHK(c, hipMemcpy(c->d_textNames, blob.data(), blob.size(), hipMemcpyHostToDevice));
      c->textMask = mask; c->textBlobLen = (unsigned)blob.size(); c->textBlob = key;
    }
  }
  gtxtext::TextTables tabs; tabs.table = c->d_textTable; tabs.tableMask = c->textMask; tabs.names = c->d_textNames; tabs.prevOff = 0; tabs.prevLen = 0;
  if (r->have_prev && r->prev_chrom) {
    const size_t len = strlen(r->prev_chrom);
    if (len > 0 && len < 4096) {                                       // the seam's name behind the names, one place per slot
      memcpy(t.h_seam, r->prev_chrom, len);
      tabs.prevOff = c->textBlobLen + (unsigned)slot * 4096; tabs.prevLen = (unsigned)len;
      HIPCHK(c, hipMemcpyAsync(c->d_textNames + tabs.prevOff, t.h_seam, len, hipMemcpyHostToDevice, c->stream));
    }
  }
  // the text: page-locked memory is read where it is, anything else goes through a page-locked slot of the context
  const char *src = text;
  if (!is_pinned(text)) {
    if (bytes > t.capPin) { if (t.h_pin) (void)hipHostFree(t.h_pin); t.h_pin = nullptr; t.capPin = 0; HIPCHK(c, hipHostMalloc((void **)&t.h_pin, bytes + (bytes >> 3))); t.capPin = bytes + (bytes >> 3); }
    parallel_copy(t.h_pin, text, bytes, c->copyThreads);
    src = t.h_pin;
  }
  HIPCHK(c, hipMemcpyAsync(t.d_text, src, bytes, hipMemcpyHostToDevice, c->copyStream));
  HIPCHK(c, hipEventRecord(t.evCopied, c->copyStream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, t.evCopied, 0));
  HIPCHK(c, hipMemsetAsync(t.d_flag, 0, sizeof(int), c->stream));
  if (r->strand_aware && (size_t)nLines > t.capLines2) {
    dfree(t.d_tri2); dfree(t.d_w2); dfree(t.d_blk); t.capLines2 = 0;
    HIPCHK(c, hipMalloc(&t.d_tri2, sizeof(int) * 3 * t.capLines)); HIPCHK(c, hipMalloc(&t.d_w2, sizeof(int) * t.capLines)); HIPCHK(c, hipMalloc(&t.d_blk, sizeof(unsigned) * (t.capLines / 128 + 4)));
    t.capLines2 = t.capLines;
  }
  gtxtext::TextDevice d; d.text = t.d_text; d.segCount = t.d_seg; d.nl = t.d_nl; d.tri = t.d_tri; d.w = t.d_w; d.flag = t.d_flag;
  d.tri2 = r->strand_aware ? t.d_tri2 : nullptr; d.w2 = r->strand_aware ? t.d_w2 : nullptr; d.blkMinus = r->strand_aware ? t.d_blk : nullptr;
  const int *triOut = r->strand_aware ? t.d_tri2 : t.d_tri;
  d.labelSum = mode == TEXT_SCAN ? c->scan.d_labelSum : nullptr; d.blockSum = t.d_sum;
  HIPCHK(c, gtxtext::launch_tokenize(d, tabs, *r, bytes, (unsigned)nLines, c->stream, mode == TEXT_SCAN ? (c->scan.a.sortedRule ? 2 : 1) : 0));
  HIPCHK(c, hipMemcpyAsync(t.h_flag, t.d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(t.evParsed, c->stream));
  // ... and counted where the triples are
  const int *dW = r->max_label_value > 1 ? (r->strand_aware ? t.d_w2 : t.d_w) : nullptr;
  const int64_t seen = c->streamSeen;
  int rc = GTX_OK;
  if (mode == TEXT_SCAN) {
    // (a block of a sorted stream is in order: the general kernels aggregate runs of equal micro-windows; any other block: the partition path)
    rc = scan_hist_any(c, triOut, dW, nLines, c->scan.a, c->scan.classLen.data(), (flags & GTX_READS_UNSORTED) != 0); if (rc) return rc;
    HIPCHK(c, hipEventRecord(t.evConsumed, c->stream));
    t.busy = true;
    return GTX_OK;
  }
  if (coverage) {
    if (flags & GTX_GAPS_FORMULA) { rc = merge_prepare(c, flags, 2); if (rc) return rc; }
    rc = cover_launch(c, triOut, dW, nLines, seen, flags, (flags & GTX_READS_UNSORTED) != 0 || !(flags & GTX_READS_SORTED)); if (rc) return rc;
  } else {
    rc = merge_prepare(c, flags, 0); if (rc) return rc;
    const bool streaming = (flags & (GTX_READS_SORTED | GTX_CHECK_SORTED)) != 0;
    if (!streaming) c->tileSumsValid = false;
    if (streaming) HIPCHK(c, gtx::launch_count(triOut, dW, nLines, count_args(c, flags & ~GTX_CHECK_SORTED, nLines, seen), true, c->stream));
    else { rc = launch_unsorted(c, triOut, dW, nLines, count_args(c, flags, nLines, seen)); if (rc) return rc; }
    rc = pairs_batch(c, triOut, dW, nLines); if (rc) return rc;
  }
  rc = merge_batch(c, triOut, dW, nLines); if (rc) return rc;
  HIPCHK(c, hipEventRecord(t.evConsumed, c->stream));
  t.busy = true;
  c->streamSeen += nLines;
  return GTX_OK;
}

extern "C" {
int gtx_count_add_text(gtx_ctx *c, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket)
{ return c ? add_text(c, TEXT_COUNT, text, bytes, n_lines, rules, flags, ticket) : GTX_E_ARG; }
int gtx_coverage_add_text(gtx_ctx *c, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket)
{ return c ? add_text(c, TEXT_COVER, text, bytes, n_lines, rules, flags, ticket) : GTX_E_ARG; }
int gtx_scan_add_text(gtx_ctx *c, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket)
{ return c ? add_text(c, TEXT_SCAN, text, bytes, n_lines, rules, flags, ticket) : GTX_E_ARG; }
int gtx_text_result(gtx_ctx *c, int ticket, int *needs_host)
{
  if (!c || !needs_host || (ticket & ~1)) return c ? fail(c, GTX_E_ARG, "gtx_text_result: bad argument") : GTX_E_ARG;
  gtx_ctx::TextSlot &t = c->text[ticket];
  if (!t.evParsed) return fail(c, GTX_E_STATE, "gtx_text_result: no such block");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(t.evParsed));
  *needs_host = *t.h_flag;
  return GTX_OK;
}
}

extern "C" {

// A group member's share: the classes `owned` (flags per class).  tiles = the 1024-slot histogram tiles that hold a slot of an
// owned class or the slot just below its first (class c has slots seg[c]+c-1 .. seg[c+1]+c: the gather reads the prefix at the
// slot below as the class's base); regions = `regions[0..nRegions)`, the file indices of the member's regions in the group's
// compact order, its piece beginning at compact position `offset`.
int gtxi_set_share(gtx_ctx *c, const uint8_t *owned, int32_t nClasses, const int32_t *regions, int64_t nRegions, int64_t offset)
{
  if (c->nRefs < 0) return fail(c, GTX_E_STATE, "gtxi_set_share: gtx_set_refs has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int nTilesAll = gtx::scan_tiles(c->histLen);
  std::vector<uint8_t> mark(nTilesAll > 0 ? nTilesAll : 1, 0);
  for (int cl = 0; cl < c->nClasses && cl < nClasses; cl++) {
    if (!owned[cl] || c->h_seg[cl] == c->h_seg[cl + 1]) continue;
    const int64_t lo = std::max<int64_t>(0, (int64_t)c->h_seg[cl] + cl - 1), hi = (int64_t)c->h_seg[cl + 1] + cl;   // slots lo..hi
    for (int64_t t = lo >> 10; t <= (hi >> 10) && t < nTilesAll; t++) mark[t] = 1;
  }
  std::vector<int32_t> tiles;
  for (int t = 0; t < nTilesAll; t++) if (mark[t]) tiles.push_back(t);
  dfree(c->d_shareTiles); dfree(c->d_shareRegions); dfree(c->d_shareOwned);
  {
    std::vector<uint8_t> own((size_t)std::max(c->nClasses, 1), 0);
    for (int cl = 0; cl < c->nClasses && cl < nClasses; cl++) own[cl] = owned[cl] ? 1 : 0;
    HIPCHK(c, hipMalloc(&c->d_shareOwned, own.size()));
    HIPCHK(c, hipMemcpy(c->d_shareOwned, own.data(), own.size(), hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMalloc(&c->d_shareTiles, sizeof(int32_t) * (tiles.size() + 1)));
  HIPCHK(c, hipMalloc(&c->d_shareRegions, sizeof(int32_t) * (size_t)(nRegions + 1)));
  if (!tiles.empty()) HIPCHK(c, hipMemcpy(c->d_shareTiles, tiles.data(), sizeof(int32_t) * tiles.size(), hipMemcpyHostToDevice));
  if (nRegions > 0) HIPCHK(c, hipMemcpy(c->d_shareRegions, regions, sizeof(int32_t) * (size_t)nRegions, hipMemcpyHostToDevice));
  c->nShareTiles = (int)tiles.size(); c->nShareRegions = nRegions; c->shareOffset = offset; c->shareOn = true;
  return ensure_out(c, GTXI_SHARE_SLOTS * (size_t)c->nRefs);     // the compact vectors gtxi_count_device_share[_async] takes in turn (slot)
}

// gtx_count_device for a group member: the reads (of the member's classes, resident on its device) are counted and the member's
// regions finalized into its piece of compact vector `slot` (0 | 1), c->d_out + slot * nRefs + shareOffset.  Enqueued on the
// context's stream.
int gtxi_count_device_share(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, uint32_t flags, int slot, void *direct_out, void **d_piece, int64_t *pieceLen)
{
  if (!c->shareOn) return fail(c, GTX_E_STATE, "gtxi_count_device_share: no share set");
  if (c->refBlocks) return fail(c, GTX_E_STATE, "gtx_group_count_device: multi-interval regions (gtx_set_ref_blocks) are outside the members' shares");
  if (n < 0 || (n > 0 && !d_reads)) return fail(c, GTX_E_ARG, "gtx_group_count_device: bad argument");
  if (flags & GTX_ZERO_LENGTH_OK) return fail(c, GTX_E_ARG, "gtx_group_count_device: GTX_ZERO_LENGTH_OK (sorted-merge semantics with their host-side corrections) is served by the host-buffer group calls only");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = count_begin(c); if (rc) return rc;
  c->profThis = c->prof && (c->profEvery <= 1 || (c->profSeq++ % c->profEvery) == 0);
  if (c->profThis) { c->ev = c->evRing[c->profCalls % gtx_ctx::kProfSlots]; HIPCHK(c, hipEventRecord(c->ev[1], c->stream)); }
  const bool streaming = (flags & (GTX_READS_SORTED | GTX_CHECK_SORTED)) != 0;
  if (!streaming) c->tileSumsValid = false;
  if (n > 0) {
    if (streaming) HIPCHK(c, gtx::launch_count(d_reads, d_weights, n, count_args(c, flags, n, 0, nullptr, true), true, c->stream));
    else { rc = launch_unsorted(c, d_reads, d_weights, n, count_args(c, flags, n, 0, nullptr, true)); if (rc) return rc; }
  }
  if (c->profThis) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  u64 *dst = direct_out ? (u64 *)direct_out : c->d_out + (size_t)(slot % GTXI_SHARE_SLOTS) * c->nRefs + c->shareOffset;
  rc = count_end(c, dst, true, direct_out != nullptr); if (rc) return rc;
  if (c->profThis) { if (c->profEvery <= 1) HIPCHK(c, hipEventRecord(c->ev[3], c->stream)); c->profCalls++; }
  *d_piece = dst; *pieceLen = c->nShareRegions;
  return GTX_OK;
}

// The same for reads in stream order, without a wait between successive calls: the streaming kernel and the finalize step of the
// member's share run on `run` -- the group alternates between two streams of its own, so that the kernel of call k+1 is not ordered
// behind the kernel of call k and takes the wave slots its tail frees -- with histogram set `set` (0 | 1: one per stream; the stream's
// order is what keeps call k+2's kernel out of what call k's finalize step is scanning and zeroing).  The finalize step writes the
// member's piece of compact vector `slot`; the caller has made `run` wait for whatever last read that piece.
static constexpr int kInfoRing = 2 * GTXI_SHARE_STREAMS;
int gtxi_count_device_share_async(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, uint32_t flags, int slot, int set, hipStream_t run,
                                  void *direct_out, void **d_piece, int64_t *pieceLen)
{
  if (!c->shareOn) return fail(c, GTX_E_STATE, "gtxi_count_device_share_async: no share set");
  if (c->refBlocks || (flags & GTX_ZERO_LENGTH_OK) || !(flags & GTX_READS_SORTED) || n < 0 || (n > 0 && !d_reads))
    return fail(c, GTX_E_ARG, "gtxi_count_device_share_async: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  gtx_ctx::HistSet &h = c->alt[set % GTXI_SHARE_STREAMS];
  const int nTiles = gtx::scan_tiles(c->histLen);
  if (!h.histA) {
    u64 **arr[] = {&h.histA, &h.histB, &h.prefA, &h.prefB};
    for (u64 **p : arr) HIPCHK(c, hipMalloc(p, sizeof(u64) * c->histLen));
    HIPCHK(c, hipMalloc(&h.partA, sizeof(u64) * (nTiles + 2))); HIPCHK(c, hipMalloc(&h.partB, sizeof(u64) * (nTiles + 2)));
    HIPCHK(c, hipMalloc(&h.flags, sizeof(unsigned) * 8 * (nTiles + 2)));
    HIPCHK(c, hipMemset(h.histA, 0, sizeof(u64) * c->histLen)); HIPCHK(c, hipMemset(h.histB, 0, sizeof(u64) * c->histLen));
    HIPCHK(c, hipMemset(h.partA, 0, sizeof(u64) * (nTiles + 2))); HIPCHK(c, hipMemset(h.partB, 0, sizeof(u64) * (nTiles + 2)));
    HIPCHK(c, hipMemset(h.flags, 0, sizeof(unsigned) * 8 * (nTiles + 2)));
    h.epoch = 0; h.draws = 0;
  }
  if (!c->d_info3) {
    HIPCHK(c, hipMalloc(&c->d_info3, kInfoRing * sizeof(gtx::DevInfo)));
    for (int k = 0; k < kInfoRing; k++) HIPCHK(c, hipMemcpy(c->d_info3 + k, &c->h_info[1], sizeof(gtx::DevInfo), hipMemcpyHostToDevice));
  }
  // info blocks: the calls on stream `set` take two blocks in turn; a call's finalize step clears the other one -- the block of the
  // call that ran on this stream before, for the call that runs on it next
  const int st = set % GTXI_SHARE_STREAMS;
  const unsigned turn = c->shareTurn[st]++;
  c->shareSeq++;
  gtx::DevInfo *info = c->d_info3 + 2 * st + (turn & 1), *infoNext = c->d_info3 + 2 * st + ((turn + 1) & 1);
  c->profThis = c->prof && (c->profEvery <= 1 || (c->profSeq++ % c->profEvery) == 0);
  if (c->profThis) { c->ev = c->evRing[c->profCalls % gtx_ctx::kProfSlots]; HIPCHK(c, hipEventRecord(c->ev[1], run)); }
  const bool keepDefault = c->tileSumsValid;
  c->tileSumsValid = true;
  gtx::CountArgs a = count_args(c, flags, n, 0, &h, false, d_weights == nullptr);   // (clears c->tileSumsValid when the kernel leaves the tile sums to the finalize step)
  a.info = info;
  const bool sumsValid = c->tileSumsValid;
  c->tileSumsValid = keepDefault;
  if (n > 0) HIPCHK(c, gtx::launch_count(d_reads, d_weights, n, a, true, run));
  if (c->profThis) HIPCHK(c, hipEventRecord(c->ev[2], run));
  // direct_out (may be null): the caller's result vector in file order (n_refs entries) -- the member's regions go to their places in it
  u64 *dst = direct_out ? (u64 *)direct_out : c->d_out + (size_t)(slot % GTXI_SHARE_SLOTS) * c->nRefs + c->shareOffset;
  gtx::FinalizeShare fs = {c->d_shareTiles, c->nShareTiles, c->d_shareRegions, c->nShareRegions, direct_out != nullptr};
  if (++h.epoch == 0) { HIPCHK(c, hipMemsetAsync(h.flags, 0, sizeof(unsigned) * 8 * (nTiles + 2), run)); h.epoch = 1; h.draws = 0; }
  HIPCHK(c, gtx::launch_finalize(h.histA, h.histB, c->histLen, h.partA, h.partB, sumsValid, h.prefA, h.prefB, c->d_posE, c->d_posS, c->d_classBase,
                                 c->nRefs, dst, infoNext, run, &fs, h.flags, h.epoch, info, &h.draws, a.hist32 != 0 && n > 0));
  if (c->profThis) { if (c->profEvery <= 1) HIPCHK(c, hipEventRecord(c->ev[3], run)); c->profCalls++; }
  c->lastShareInfo = info;
  *d_piece = dst; *pieceLen = c->nShareRegions;
  return GTX_OK;
}

// what the last gtxi_count_device_share_async call observed (the caller has waited for its streams)
int gtxi_last_share_info(gtx_ctx *c, gtx_count_info *info)
{
  if (!c->lastShareInfo) return gtx_last_info(c, info);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(&c->h_info[0], c->lastShareInfo, sizeof(gtx::DevInfo), hipMemcpyDeviceToHost));
  info_out(c->h_info[0], info, 0);
  return fault_check(c, c->h_info[0]);
}

void *gtxi_out_buffer(gtx_ctx *c) { return c->d_out; }
int gtxi_ensure_out(gtx_ctx *c, int64_t n)
{
  HIPCHK(c, hipSetDevice(c->device));
  if ((size_t)n > c->capOut) HIPCHK(c, hipStreamSynchronize(c->stream));      // (the old vector may still be read)
  return ensure_out(c, (size_t)std::max<int64_t>(n, 0));
}
// a second device buffer of the context (grown, never shrunk): where a group's member 0 assembles a result
int gtxi_scratch(gtx_ctx *c, size_t bytes, void **p)
{
  HIPCHK(c, hipSetDevice(c->device));
  if (bytes > c->capScratch) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfree(c->d_scratch); c->capScratch = 0;
    HIPCHK(c, hipMalloc(&c->d_scratch, bytes));
    c->capScratch = bytes;
  }
  *p = c->d_scratch;
  return GTX_OK;
}

// a DMA out of the caller's page-locked buffer may still be in flight (stage_batches returns with it enqueued): wait for it
int gtxi_wait_direct(gtx_ctx *c)
{
  if (c->directPending) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->copyStream)); c->directPending = false; }
  return GTX_OK;
}

hipStream_t gtxi_stream(gtx_ctx *c) { return c->stream; }
int gtxi_device(gtx_ctx *c) { return c->device; }
void gtxi_set_error(gtx_ctx *c, const char *msg) { c->err = msg; }

// ---------------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------------
int gtx_profile_enable(gtx_ctx *c, int on)
{
  if (!c || on < 0) return GTX_E_ARG;
  c->prof = on != 0; c->profEvery = on > 1 ? on : 1; c->profCalls = 0; c->profSeq = 0; c->profThis = false;
  return GTX_OK;
}

int gtx_profile_read(gtx_ctx *c, int back, float *msKernel, float *msTotal)
{
  if (!c) return GTX_E_ARG;
  if (back < 0 || back >= gtx_ctx::kProfSlots || back >= c->profCalls) return fail(c, GTX_E_STATE, "gtx_profile_read: no such profiled call");
  HIPCHK(c, hipSetDevice(c->device));
  hipEvent_t *ev = c->evRing[(c->profCalls - 1 - back) % gtx_ctx::kProfSlots];
  const bool kernelOnly = c->profEvery > 1;
  HIPCHK(c, hipEventSynchronize(ev[kernelOnly ? 2 : 3]));
  float a = 0, b = 0;
  HIPCHK(c, hipEventElapsedTime(&a, ev[1], ev[2]));
  if (kernelOnly) b = a; else HIPCHK(c, hipEventElapsedTime(&b, ev[1], ev[3]));
  if (msKernel) *msKernel = a;
  if (msTotal) *msTotal = b;
  return GTX_OK;
}

int gtx_profile_last(gtx_ctx *c, float *msKernel, float *msTotal) { return gtx_profile_read(c, 0, msKernel, msTotal); }

int gtx_profile_count(gtx_ctx *c) { return c ? (int)std::min<long long>(c->profCalls, gtx_ctx::kProfSlots) : 0; }

} // extern "C"
