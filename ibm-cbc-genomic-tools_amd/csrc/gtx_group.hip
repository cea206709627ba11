// gtx_group.hip -- the counting path on several MI355X of one node (gtx_group_* of include/gtx.h).
//
// Two regions can overlap only on one chromosome [and strand] (GenomicInterval::OverlapsWith,
// gtools/genomic_intervals.cpp:624-630), so the reads and regions of one class are an independent unit of work: classes
// are dealt to the members of a group (longest-processing-time packing of a per-class load), every member holds the whole
// reference set (20 MB) and counts the reads of ITS classes.  What a member does after its streaming kernel shrinks with its
// share: it finalizes only the histogram tiles of its classes and only its regions, into its piece of a COMPACT vector -- the
// regions ordered by (owner of their class, position in the file) -- and the pieces travel to member 0 over xGMI with one
// grouped RCCL send / receive per member (the reduce of north_star with its addends known to be disjoint: a sum of the members'
// zero-padded full vectors would move and add N times the bytes).  Member 0 puts the compact vector into file order.  The
// sliding windows of genomic_scans are per class as well: a member scans its classes into a packed vector of its own and the
// per-class pieces travel the same way.
//
// A group is either ONE process driving all devices (gtx_group_create: a context per device, asynchronous enqueues from the
// caller's thread -- the reference's tools are single processes) or one process per device (gtx_group_create_rank: the same
// object holding only the local member, the communicator made from an id the caller's launcher hands around -- the shape of
// `python -m torch.distributed.run bench.py`).  librccl is resolved at run time; a group of one needs none.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include <rccl/rccl.h>          // types and enums only: the functions are looked up with dlsym
#include "gtx.h"
#include "gtx_internal.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string *err)
  {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { *err = std::string("gtx_group: cannot load librccl: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
    Send = (decltype(Send))dlsym(lib, "ncclSend");
    Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommInitRank || !GetUniqueId || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !Send || !Recv || !GetErrorString) {
      *err = "gtx_group: librccl lacks an expected symbol"; return false;
    }
    return true;
  }
};

// RCCL announces its version with a printf on the process's stdout when the first communicator comes up, whatever NCCL_DEBUG
// says (checked: it does under NCCL_DEBUG=WARN); the tools' stdout is their result.  Descriptor 1 points at stderr while a
// communicator is being made.  The swap is process-wide: a caller that has other threads writing to stdout must keep them
// from flushing it until gtx_group_create / gtx_group_create_rank has returned (include/gtx.h; the tools hold their header
// lines back, csrc/genomic_intervals.cpp: StdoutIsOurs).
struct StdoutAside {
  int saved;
  StdoutAside() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
  ~StdoutAside() { fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};

thread_local std::string g_group_create_error;
static_assert(sizeof(ncclUniqueId) == GTX_GROUP_ID_BYTES, "include/gtx.h states the size of the communicator id");

// rehearsal on one device (GTX_GROUP_REHEARSE=1: all members may sit on the same GPU, which RCCL refuses): the members'
// full vectors of the legacy finish are summed by this kernel instead of ncclReduce
__global__ void rehearse_add_kernel(unsigned long long *__restrict__ root, const unsigned long long *__restrict__ other, long long n)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) root[i] += other[i];
}

// The runtime maps a process's streams onto a handful of hardware queues, and two streams on one queue run their kernels one after
// the other -- the streaming kernels of successive device calls are meant to run side by side (seen: of three streams made in a row
// two shared a queue, 0.045 ms per call of a member instead of 0.029; scripts/queue_probe.hip).  So the group makes more streams than
// it needs and keeps `want` of them that were SEEN to overlap pairwise: two kernels that idle for 100 us, one per stream, take 100 us
// together or 200.  A few milliseconds, once per member.
__global__ void queue_probe_kernel(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }

static hipError_t make_overlapping_streams(int want, hipStream_t *out)
{
  constexpr int kCand = 8;
  hipStream_t cand[kCand] = {};
  hipError_t e = hipSuccess;
  for (int i = 0; i < kCand && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking);
  bool clash[kCand][kCand] = {};
  if (e == hipSuccess && want > 1) {
    const long long ticks = 10000;                           // wall_clock64 runs at 100 MHz
    queue_probe_kernel<<<1, 64, 0, cand[0]>>>(100);
    e = hipDeviceSynchronize();
    auto usNow = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int i = 0; i < kCand && e == hipSuccess; i++)
      for (int j = i + 1; j < kCand && e == hipSuccess; j++) {
        const double t0 = usNow();
        queue_probe_kernel<<<1, 64, 0, cand[i]>>>(ticks); queue_probe_kernel<<<1, 64, 0, cand[j]>>>(ticks);
        e = hipStreamSynchronize(cand[i]); if (e == hipSuccess) e = hipStreamSynchronize(cand[j]);
        clash[i][j] = clash[j][i] = usNow() - t0 > 160.0;
      }
  }
  // greedy: the first streams that clash with none already taken (then, if the probe found fewer than `want`, any that are left)
  int taken[kCand], nt = 0; bool used[kCand] = {};
  for (int i = 0; i < kCand && nt < want; i++) {
    bool ok = true;
    for (int k = 0; k < nt; k++) ok = ok && !clash[i][taken[k]];
    if (ok) { taken[nt++] = i; used[i] = true; }
  }
  for (int i = 0; i < kCand && nt < want; i++) if (!used[i]) { taken[nt++] = i; used[i] = true; }
  for (int k = 0; k < nt; k++) out[k] = cand[taken[k]];
  for (int i = 0; i < kCand; i++) if (cand[i] && (!used[i] || e != hipSuccess)) (void)hipStreamDestroy(cand[i]);
  return e;
}

// member 0: the compact vector (regions ordered by owner, then file position) into file order
__global__ __launch_bounds__(256) void unpermute_kernel(const unsigned long long *__restrict__ compact, const int *__restrict__ perm, long long m,
                                                        unsigned long long *__restrict__ hits)
{
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) hits[perm[j]] = compact[j];
}

}  // namespace

static constexpr int kSlots = GTXI_SHARE_SLOTS;

struct gtx_group {
  int nm = 0;                              // members of the group (all processes together)
  int rank = -1;                           // >= 0: one process per member, this process holds member `rank` only
  std::vector<gtx_ctx *> ctx;              // the local members' contexts (all of them, or one)
  std::vector<int> dev;
  std::vector<ncclComm_t> comm;            // per local member; empty for a group of one (unless GTX_GROUP_FORCE_RCCL=1)
  Rccl rccl;
  std::string err;
  std::vector<int32_t> owner;              // class -> member
  std::vector<int64_t> memberReads;        // reads routed to each member in the open call
  bool countOpen = false, coverOpen = false;
  bool fullVectors = false;                // the open count call has counted text blocks: reads of any class on any member -- the members' full vectors are summed
  unsigned long long textTurn = 0;         // gtx_group_*_add_text: next member
  unsigned long long regionsTurn = 0;                 // gtx_group_count_add_regions: next member
  bool lastAsync = false;                             // the last device count call finalized on the exchange streams
  bool rehearse = false;                   // GTX_GROUP_REHEARSE=1
  bool selfExchange = false;               // GTX_GROUP_SELF_EXCHANGE=1 (test hook): member 0's own piece travels through RCCL to itself
  int64_t nRefs = 0; uint32_t refFlags = 0; int32_t nClasses = 0;
  std::vector<int32_t> refClass;           // class of every reference region (file order)
  // compact layout (gtx_group_plan): piece of member m = compact positions segOff[m] .. segOff[m+1]
  bool planValid = false; std::vector<int64_t> segOff; std::vector<int32_t> perm;
  // runs of a member's piece that are consecutive in the FILE as well (a reference file in class order: one run per class): with few
  // of them the pieces travel run by run straight to their places in the caller's vector -- no compact vector on member 0, no reordering
  struct Run { int64_t compact, file, len; };
  std::vector<std::vector<Run>> runs;      // per member
  bool direct = false;
  int *d_perm = nullptr; unsigned long long *d_selfTmp = nullptr; size_t capSelfTmp = 0;
  hipEvent_t evPiece = nullptr;            // rehearsal (scans): a member's piece is ready
  // gtx_group_count_device: the pieces travel on a stream of their own per local member, behind an event of the member's
  // finalize step, into one of two compact vectors in turn -- call k+1's kernels run under call k's exchange
  std::vector<hipStream_t> xs; std::vector<hipEvent_t> evFinal[kSlots], evXchg[kSlots]; bool xchgUsed[kSlots] = {}; long long seq = 0;
  // ... and the streaming kernels of successive calls (reads in stream order) alternate between two streams of the group's own per
  // local member: nothing orders the kernel of call k+1 behind the kernel of call k (they count into different histogram sets), so
  // its waves take the slots the tail of call k frees -- a member's launch at 1/8 of the reads is short enough for start and tail to
  // be a third of it.  evReady: the caller's stream at the moment of the call (the reads are resident behind it).
  std::vector<hipStream_t> ks[GTXI_SHARE_STREAMS]; int nks = 3; std::vector<hipEvent_t> evReady[kSlots], evJoin[GTXI_SHARE_STREAMS + 1];
  std::vector<hipStream_t> lastRan;        // where each local member's finalize step of the last device call was enqueued
  bool resultPending = false;
  // scratch of the router for interleaved input
  std::vector<std::vector<int32_t>> partTri, partW;

  int local(int m) const { return rank >= 0 ? (m == rank ? 0 : -1) : m; }     // index into ctx / comm, -1 = another process's
  int member(int li) const { return rank >= 0 ? rank : li; }
};

#define GCHK_HIP(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e_); return GTX_E_HIP; } } while (0)
#define GCHK_NCCL(g, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (g)->err = std::string(#call) + ": " + (g)->rccl.GetErrorString(r_); return GTX_E_HIP; } } while (0)
#define GCHK_CTX(g, i, call) do { int rc_ = (call); if (rc_ != GTX_OK) { (g)->err = std::string("member ") + std::to_string((g)->member(i)) + ": " + gtx_last_error((g)->ctx[i]); return rc_; } } while (0)

static int gfail(gtx_group *g, int code, const char *msg) { g->err = msg; return code; }

extern "C" {

gtx_group *gtx_group_create(int n, const int *device_ids)
{
  if (n < 1) { g_group_create_error = "gtx_group_create: need at least one device"; return nullptr; }
  gtx_group *g = new gtx_group();
  g->nm = n;
  { const char *rh = getenv("GTX_GROUP_REHEARSE"); g->rehearse = rh && atoi(rh); }
  { const char *se = getenv("GTX_GROUP_SELF_EXCHANGE"); g->selfExchange = se && atoi(se); }
  for (int i = 0; i < n; i++) {
    const int d = device_ids ? device_ids[i] : i;
    for (int j = 0; j < i && !g->rehearse; j++) if (g->dev[j] == d) { g_group_create_error = "gtx_group_create: a device is listed twice"; gtx_group_destroy(g); return nullptr; }
    gtx_ctx *c = gtx_create(d);
    if (!c) { g_group_create_error = gtx_last_error(nullptr); gtx_group_destroy(g); return nullptr; }
    g->ctx.push_back(c); g->dev.push_back(d);
  }
  const char *force = getenv("GTX_GROUP_FORCE_RCCL");
  if (!g->rehearse && (n > 1 || (force && atoi(force)) || g->selfExchange)) {
    if (!g->rccl.load(&g_group_create_error)) { gtx_group_destroy(g); return nullptr; }
    g->comm.resize(n);
    ncclResult_t r;
    { StdoutAside aside; r = g->rccl.CommInitAll(g->comm.data(), n, g->dev.data()); }
    if (r != ncclSuccess) { g_group_create_error = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r); g->comm.clear(); gtx_group_destroy(g); return nullptr; }
  }
  g->memberReads.assign(n, 0);
  g->partTri.resize(n); g->partW.resize(n);
  return g;
}

int gtx_group_unique_id(void *id_out)
{
  if (!id_out) return GTX_E_ARG;
  Rccl r;
  if (!r.load(&g_group_create_error)) return GTX_E_HIP;
  ncclUniqueId id;
  ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) { g_group_create_error = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return GTX_E_HIP; }
  memcpy(id_out, &id, sizeof id);
  return GTX_OK;
}

gtx_group *gtx_group_create_rank(int device_id, int rank, int world, const void *unique_id)
{
  if (world < 1 || rank < 0 || rank >= world) { g_group_create_error = "gtx_group_create_rank: bad rank / world size"; return nullptr; }
  // GTX_GROUP_NO_EXCHANGE=1 (measurement only): a member of a larger group without a communicator -- what it does locally
  // (streaming kernel, finalize of its share, member 0's reordering) can be timed on one GPU; nothing travels
  const char *nx = getenv("GTX_GROUP_NO_EXCHANGE");
  if (world > 1 && !unique_id && !(nx && atoi(nx))) { g_group_create_error = "gtx_group_create_rank: a group of more than one process needs the id of gtx_group_unique_id"; return nullptr; }
  gtx_group *g = new gtx_group();
  g->nm = world; g->rank = rank;
  { const char *se = getenv("GTX_GROUP_SELF_EXCHANGE"); g->selfExchange = se && atoi(se) && world == 1; }
  gtx_ctx *c = gtx_create(device_id);
  if (!c) { g_group_create_error = gtx_last_error(nullptr); delete g; return nullptr; }
  g->ctx.push_back(c); g->dev.push_back(device_id);
  if ((world > 1 && unique_id) || (unique_id && g->selfExchange)) {
    if (!g->rccl.load(&g_group_create_error)) { gtx_group_destroy(g); return nullptr; }
    ncclUniqueId id; memcpy(&id, unique_id, sizeof id);
    g->comm.resize(1);
    ncclResult_t r;
    { StdoutAside aside; r = g->rccl.CommInitRank(&g->comm[0], world, id, rank); }     // (on the device of the calling thread: gtx_create has set it)
    if (r != ncclSuccess) { g_group_create_error = std::string("ncclCommInitRank: ") + g->rccl.GetErrorString(r); g->comm.clear(); gtx_group_destroy(g); return nullptr; }
  }
  g->memberReads.assign(world, 0);
  return g;
}

void gtx_group_destroy(gtx_group *g)
{
  if (!g) return;
  for (ncclComm_t c : g->comm) if (c) g->rccl.CommDestroy(c);
  for (size_t li = 0; li < g->xs.size(); li++) {
    (void)hipSetDevice(g->dev[li]);
    for (int k = 0; k < GTXI_SHARE_STREAMS; k++) if (li < g->ks[k].size() && g->ks[k][li]) { (void)hipStreamSynchronize(g->ks[k][li]); (void)hipStreamDestroy(g->ks[k][li]); }
    if (g->xs[li]) { (void)hipStreamSynchronize(g->xs[li]); (void)hipStreamDestroy(g->xs[li]); }
    for (int k = 0; k <= GTXI_SHARE_STREAMS; k++) if (li < g->evJoin[k].size() && g->evJoin[k][li]) (void)hipEventDestroy(g->evJoin[k][li]);
    for (int k = 0; k < kSlots; k++) {
      if (li < g->evFinal[k].size() && g->evFinal[k][li]) (void)hipEventDestroy(g->evFinal[k][li]);
      if (li < g->evXchg[k].size() && g->evXchg[k][li]) (void)hipEventDestroy(g->evXchg[k][li]);
      if (li < g->evReady[k].size() && g->evReady[k][li]) (void)hipEventDestroy(g->evReady[k][li]);
    }
  }
  if (!g->ctx.empty()) {
    (void)hipSetDevice(g->dev[0]);
    if (g->d_perm) (void)hipFree(g->d_perm);
    if (g->d_selfTmp) (void)hipFree(g->d_selfTmp);
    if (g->evPiece) (void)hipEventDestroy(g->evPiece);
  }
  for (gtx_ctx *c : g->ctx) gtx_destroy(c);
  delete g;
}

int gtx_group_size(const gtx_group *g) { return g ? g->nm : 0; }
int gtx_group_rank(const gtx_group *g) { return g ? g->rank : -1; }
gtx_ctx *gtx_group_ctx(gtx_group *g, int member)
{
  if (!g || member < 0 || member >= g->nm) return nullptr;
  const int li = g->local(member);
  return li >= 0 ? g->ctx[li] : nullptr;
}
const char *gtx_group_last_error(const gtx_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

// longest-processing-time packing: classes by decreasing load, each to the member with the least load so far
// (ties: lowest class id first, lowest member first -- deterministic, so every process of a group computes the same)
void gtx_lpt_assign(const int64_t *load, int32_t n_classes, int n_members, int32_t *owner_out)
{
  std::vector<int32_t> order(n_classes);
  for (int32_t c = 0; c < n_classes; c++) order[c] = c;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return load[a] > load[b]; });
  std::vector<int64_t> sum(n_members, 0);
  for (int32_t c : order) {
    int best = 0;
    for (int m = 1; m < n_members; m++) if (sum[m] < sum[best]) best = m;
    owner_out[c] = best; sum[best] += load[c] > 0 ? load[c] : 0;
  }
}

// the compact order of a group's result: regions by (owner of their class, position in the file); regions of no class (a
// placeholder, or an id beyond the assignment) belong to member 0.  Pure host code.
int gtx_group_plan(const int32_t *ref_class, int64_t stride, int64_t n_refs, const int32_t *owner, int32_t n_classes, int n_members,
                   int64_t *seg_offset, int32_t *perm)
{
  if (n_refs < 0 || n_members < 1 || (n_refs > 0 && (!ref_class || !perm)) || !seg_offset || (n_classes > 0 && !owner) || stride < 1) return GTX_E_ARG;
  std::vector<int64_t> cnt(n_members + 1, 0);
  auto own = [&](int64_t k) { const int32_t c = ref_class[k * stride]; const int o = (uint32_t)c < (uint32_t)n_classes ? owner[c] : 0; return o >= 0 && o < n_members ? o : 0; };
  for (int64_t k = 0; k < n_refs; k++) cnt[own(k) + 1]++;
  for (int m = 0; m < n_members; m++) cnt[m + 1] += cnt[m];
  for (int m = 0; m <= n_members; m++) seg_offset[m] = cnt[m];
  std::vector<int64_t> at(cnt.begin(), cnt.end() - 1);
  for (int64_t k = 0; k < n_refs; k++) perm[at[own(k)]++] = (int32_t)k;
  return GTX_OK;
}

int gtx_group_assign(gtx_group *g, const int64_t *class_load, int32_t n_classes, int32_t *owner_out)
{
  if (!g || n_classes < 0 || (n_classes > 0 && !class_load)) return g ? gfail(g, GTX_E_ARG, "gtx_group_assign: bad argument") : GTX_E_ARG;
  g->owner.resize(n_classes);
  gtx_lpt_assign(class_load, n_classes, g->nm, g->owner.data());
  g->planValid = false;
  if (owner_out) memcpy(owner_out, g->owner.data(), sizeof(int32_t) * n_classes);
  return GTX_OK;
}

int gtx_group_set_refs(gtx_group *g, const int32_t *tri, int64_t m, int32_t n_classes, uint32_t flags)
{
  if (!g) return GTX_E_ARG;
  { int rcs = gtx_group_sync(g); if (rcs) return rcs; }        // (device calls in flight on the group's own streams read what is about to be replaced)
  const int n = (int)g->ctx.size();
  std::vector<int> rc(n, GTX_OK);
  std::vector<std::thread> th;                               // the host-side sorts of the members run side by side
  for (int i = 1; i < n; i++) th.emplace_back([&, i] { rc[i] = gtx_set_refs_ex(g->ctx[i], tri, m, n_classes, flags); });
  rc[0] = gtx_set_refs_ex(g->ctx[0], tri, m, n_classes, flags);
  for (auto &t : th) t.join();
  for (int i = 0; i < n; i++) GCHK_CTX(g, i, rc[i]);
  g->nRefs = m; g->refFlags = flags; g->planValid = false;
  int32_t nc = n_classes;
  if (nc <= 0) { nc = 1; for (int64_t k = 0; k < m; k++) nc = std::max(nc, tri[3 * k] + 1); }
  g->nClasses = nc;
  g->refClass.resize((size_t)m);
  for (int64_t k = 0; k < m; k++) g->refClass[k] = tri[3 * k];
  if ((int32_t)g->owner.size() != nc) {
    // no assignment given for these classes: load = the span of a class's reference regions, a stand-in for the chromosome length
    std::vector<int64_t> lo(nc, INT64_MAX), hi(nc, INT64_MIN), load(nc, 0);
    for (int64_t k = 0; k < m; k++) { const int32_t c = tri[3 * k]; if (c < 0 || c >= nc) continue; lo[c] = std::min<int64_t>(lo[c], tri[3 * k + 1]); hi[c] = std::max<int64_t>(hi[c], tri[3 * k + 2]); }
    for (int32_t c = 0; c < nc; c++) load[c] = hi[c] >= lo[c] ? hi[c] - lo[c] + 1 : 0;
    g->owner.resize(nc);
    gtx_lpt_assign(load.data(), nc, g->nm, g->owner.data());
  }
  return GTX_OK;
}

}  // extern "C"

// the compact layout and every local member's share of the finalize step, (re)made when the reference set or the assignment changed
static int ensure_plan(gtx_group *g)
{
  if (g->planValid) return GTX_OK;
  { int rcs = gtx_group_sync(g); if (rcs) return rcs; }        // (a finalize step in flight on an exchange stream reads the share lists about to be replaced)
  const int64_t m = g->nRefs;
  g->segOff.assign(g->nm + 1, 0); g->perm.assign((size_t)std::max<int64_t>(m, 1), 0);
  int rc = gtx_group_plan(g->refClass.data(), 1, m, g->owner.data(), (int32_t)g->owner.size(), g->nm, g->segOff.data(), g->perm.data());
  if (rc) return gfail(g, rc, "gtx_group: planning the compact layout failed");
  g->runs.assign(g->nm, {});
  size_t nRuns = 0;
  for (int mem = 0; mem < g->nm; mem++)
    for (int64_t j = g->segOff[mem]; j < g->segOff[mem + 1];) {
      int64_t e = j + 1;
      while (e < g->segOff[mem + 1] && g->perm[e] == g->perm[e - 1] + 1) e++;
      g->runs[mem].push_back({j, g->perm[j], e - j}); nRuns++;
      j = e;
    }
  // (GTX_GROUP_DIRECT=0 keeps the compact vector)
  const bool directOff = getenv("GTX_GROUP_DIRECT") && atoi(getenv("GTX_GROUP_DIRECT")) == 0;      // (read when a plan is made)
  g->direct = !directOff && nRuns <= 64 * (size_t)g->nm;
  std::vector<uint8_t> owned(std::max<size_t>(g->owner.size(), 1));
  for (size_t li = 0; li < g->ctx.size(); li++) {
    const int mem = g->member((int)li);
    for (size_t c = 0; c < g->owner.size(); c++) owned[c] = g->owner[c] == mem;
    GCHK_CTX(g, li, gtxi_set_share(g->ctx[li], owned.data(), (int32_t)g->owner.size(), g->perm.data() + g->segOff[mem], g->segOff[mem + 1] - g->segOff[mem], g->segOff[mem]));
  }
  if (g->local(0) >= 0) {
    GCHK_HIP(g, hipSetDevice(g->dev[g->local(0)]));
    if (g->d_perm) { (void)hipFree(g->d_perm); g->d_perm = nullptr; }
    GCHK_HIP(g, hipMalloc(&g->d_perm, sizeof(int32_t) * (size_t)std::max<int64_t>(m, 1)));
    if (m > 0) GCHK_HIP(g, hipMemcpy(g->d_perm, g->perm.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice));
    if (!g->evPiece) GCHK_HIP(g, hipEventCreateWithFlags(&g->evPiece, hipEventDisableTiming));
  }
  g->planValid = true;
  return GTX_OK;
}

// The pieces of the compact vector travel to member 0: piece[li] / len of every local member (already finalized on its stream);
// member 0 receives at root + segOff[m] (its own piece is in place).  One grouped RCCL call per process; asynchronous.
static int gather_pieces(gtx_group *g, const std::vector<void *> &piece, unsigned long long *root, const std::vector<hipStream_t> &st)
{
  const int l0 = g->local(0);
  if (l0 >= 0) GCHK_HIP(g, hipSetDevice(g->dev[l0]));
  if (g->rehearse) {                                       // one device, no RCCL: copies on member 0's stream behind the members' events
    for (size_t li = 0; li < g->ctx.size(); li++) {
      const int mem = g->member((int)li); const int64_t len = g->segOff[mem + 1] - g->segOff[mem];
      if (mem == 0 || len == 0) continue;
      if (st[li] != st[l0]) { GCHK_HIP(g, hipEventRecord(g->evPiece, st[li])); GCHK_HIP(g, hipStreamWaitEvent(st[l0], g->evPiece, 0)); }
      GCHK_HIP(g, hipMemcpyAsync(root + g->segOff[mem], piece[li], sizeof(uint64_t) * (size_t)len, hipMemcpyDeviceToDevice, st[l0]));
    }
    return GTX_OK;
  }
  if (g->comm.empty()) return GTX_OK;
  const int64_t len0 = g->segOff[1] - g->segOff[0];
  if (g->selfExchange && l0 >= 0 && len0 > 0) {            // test hook: member 0's piece leaves and comes back through RCCL
    GCHK_HIP(g, hipSetDevice(g->dev[l0]));
    if ((size_t)len0 > g->capSelfTmp) { if (g->d_selfTmp) (void)hipFree(g->d_selfTmp); g->d_selfTmp = nullptr; GCHK_HIP(g, hipMalloc(&g->d_selfTmp, sizeof(uint64_t) * (size_t)len0)); g->capSelfTmp = (size_t)len0; }
    GCHK_HIP(g, hipMemcpyAsync(g->d_selfTmp, root + g->segOff[0], sizeof(uint64_t) * (size_t)len0, hipMemcpyDeviceToDevice, st[l0]));
    GCHK_HIP(g, hipMemsetAsync(root + g->segOff[0], 0xff, sizeof(uint64_t) * (size_t)len0, st[l0]));
  }
  GCHK_NCCL(g, g->rccl.GroupStart());
  ncclResult_t r = ncclSuccess;
  for (size_t li = 0; li < g->ctx.size() && r == ncclSuccess; li++) {
    const int mem = g->member((int)li); const int64_t len = g->segOff[mem + 1] - g->segOff[mem];
    if (mem != 0 && len > 0) r = g->rccl.Send(piece[li], (size_t)len, ncclUint64, 0, g->comm[li], st[li]);
  }
  if (l0 >= 0) {
    for (int mem = 1; mem < g->nm && r == ncclSuccess; mem++) {
      const int64_t len = g->segOff[mem + 1] - g->segOff[mem];
      if (len > 0) r = g->rccl.Recv(root + g->segOff[mem], (size_t)len, ncclUint64, mem, g->comm[l0], st[l0]);
    }
    if (g->selfExchange && len0 > 0 && r == ncclSuccess) {
      r = g->rccl.Send(g->d_selfTmp, (size_t)len0, ncclUint64, 0, g->comm[l0], st[l0]);
      if (r == ncclSuccess) r = g->rccl.Recv(root + g->segOff[0], (size_t)len0, ncclUint64, 0, g->comm[l0], st[l0]);
    }
  }
  if (r != ncclSuccess) { g->rccl.GroupEnd(); g->err = std::string("ncclSend/ncclRecv: ") + g->rccl.GetErrorString(r); return GTX_E_HIP; }
  GCHK_NCCL(g, g->rccl.GroupEnd());
  return GTX_OK;
}

// The same run by run, straight into the caller's vector on member 0 (g->direct): a run of a member's piece is consecutive in the
// file as well, so member 0 receives it at d_hits + its file position; its own regions its finalize step has written there already.
static int gather_runs(gtx_group *g, const std::vector<void *> &piece, unsigned long long *d_hits, const std::vector<hipStream_t> &st)
{
  const int l0 = g->local(0);
  if (g->rehearse) {                                       // one device, no RCCL: copies on member 0's stream behind the members' events
    if (l0 < 0) return GTX_OK;
    GCHK_HIP(g, hipSetDevice(g->dev[l0]));
    for (size_t li = 0; li < g->ctx.size(); li++) {
      const int mem = g->member((int)li);
      if (mem == 0) continue;
      for (const gtx_group::Run &r : g->runs[mem])
        GCHK_HIP(g, hipMemcpyAsync(d_hits + r.file, (const unsigned long long *)piece[li] + (r.compact - g->segOff[mem]), sizeof(uint64_t) * (size_t)r.len, hipMemcpyDeviceToDevice, st[l0]));
    }
    return GTX_OK;
  }
  if (g->comm.empty()) return GTX_OK;
  const bool self = g->selfExchange && l0 >= 0 && g->segOff[1] > g->segOff[0];
  if (self) {                                              // test hook: member 0's own runs leave and come back through RCCL, like another member's
    const int64_t len0 = g->segOff[1] - g->segOff[0];
    GCHK_HIP(g, hipSetDevice(g->dev[l0]));
    if ((size_t)len0 > g->capSelfTmp) { if (g->d_selfTmp) (void)hipFree(g->d_selfTmp); g->d_selfTmp = nullptr; GCHK_HIP(g, hipMalloc(&g->d_selfTmp, sizeof(uint64_t) * (size_t)len0)); g->capSelfTmp = (size_t)len0; }
    for (const gtx_group::Run &run : g->runs[0]) {
      GCHK_HIP(g, hipMemcpyAsync(g->d_selfTmp + (run.compact - g->segOff[0]), d_hits + run.file, sizeof(uint64_t) * (size_t)run.len, hipMemcpyDeviceToDevice, st[l0]));
      GCHK_HIP(g, hipMemsetAsync(d_hits + run.file, 0xff, sizeof(uint64_t) * (size_t)run.len, st[l0]));
    }
  }
  GCHK_NCCL(g, g->rccl.GroupStart());
  ncclResult_t r = ncclSuccess;
  for (size_t li = 0; li < g->ctx.size() && r == ncclSuccess; li++) {
    const int mem = g->member((int)li);
    if (mem == 0) continue;
    for (const gtx_group::Run &run : g->runs[mem]) {
      if (r != ncclSuccess) break;
      r = g->rccl.Send((const unsigned long long *)piece[li] + (run.compact - g->segOff[mem]), (size_t)run.len, ncclUint64, 0, g->comm[li], st[li]);
    }
  }
  if (l0 >= 0) {
    for (int mem = 1; mem < g->nm && r == ncclSuccess; mem++)
      for (const gtx_group::Run &run : g->runs[mem]) {
        if (r != ncclSuccess) break;
        r = g->rccl.Recv(d_hits + run.file, (size_t)run.len, ncclUint64, mem, g->comm[l0], st[l0]);
      }
    if (self)
      for (const gtx_group::Run &run : g->runs[0]) {
        if (r != ncclSuccess) break;
        r = g->rccl.Send(g->d_selfTmp + (run.compact - g->segOff[0]), (size_t)run.len, ncclUint64, 0, g->comm[l0], st[l0]);
        if (r == ncclSuccess) r = g->rccl.Recv(d_hits + run.file, (size_t)run.len, ncclUint64, 0, g->comm[l0], st[l0]);
      }
  }
  if (r != ncclSuccess) { g->rccl.GroupEnd(); g->err = std::string("ncclSend/ncclRecv: ") + g->rccl.GetErrorString(r); return GTX_E_HIP; }
  GCHK_NCCL(g, g->rccl.GroupEnd());
  return GTX_OK;
}

// member 0: compact -> file order into d_hits (enqueued on its stream)
static int unpermute(gtx_group *g, const unsigned long long *root, void *d_hits, hipStream_t st)
{
  const int l0 = g->local(0);
  if (l0 < 0 || g->nRefs <= 0) return GTX_OK;
  GCHK_HIP(g, hipSetDevice(g->dev[l0]));
  unpermute_kernel<<<(unsigned)((g->nRefs + 255) / 256), 256, 0, st>>>(root, g->d_perm, g->nRefs, (unsigned long long *)d_hits);
  GCHK_HIP(g, hipGetLastError());
  return GTX_OK;
}

static bool pipeOffNow() { static const bool off = getenv("GTX_GROUP_PIPELINE") && atoi(getenv("GTX_GROUP_PIPELINE")) == 0; return off; }

extern "C" {

int gtx_group_count_device(gtx_group *g, const void *const *d_reads, const void *const *d_weights, const int64_t *n_reads, uint32_t flags, void *d_hits)
{
  if (!g) return GTX_E_ARG;
  if (!d_reads || !n_reads) return gfail(g, GTX_E_ARG, "gtx_group_count_device: bad argument");
  if (g->local(0) >= 0 && g->nRefs > 0 && !d_hits) return gfail(g, GTX_E_ARG, "gtx_group_count_device: member 0 needs the output vector");
  int rc = ensure_plan(g); if (rc) return rc;
  const size_t nl = g->ctx.size();
  if (g->xs.empty()) {
    g->xs.assign(nl, nullptr);
    for (int k = 0; k < kSlots; k++) { g->evFinal[k].assign(nl, nullptr); g->evXchg[k].assign(nl, nullptr); g->evReady[k].assign(nl, nullptr); }
    for (int k = 0; k < GTXI_SHARE_STREAMS; k++) g->ks[k].assign(nl, nullptr);
    for (int k = 0; k <= GTXI_SHARE_STREAMS; k++) g->evJoin[k].assign(nl, nullptr);
    if (const char *ns = getenv("GTX_GROUP_STREAMS")) g->nks = std::min(GTXI_SHARE_STREAMS, std::max(1, atoi(ns)));
    for (size_t li = 0; li < nl; li++) {
      GCHK_HIP(g, hipSetDevice(g->dev[li]));
      // A process that holds ONE member (a rank of a multi-process group: bench.py, a caller per GPU) makes the exchange stream with the
      // highest priority: the runtime gives such a stream a hardware queue of its own.  Streams of one priority may share a queue, and
      // then run in the order of their submissions -- the exchange of call k in FRONT of the kernels of call k+1 instead of under them
      // (seen with the caller's default stream: 0.283 ms per step of the single-rank self-test against 0.255 with the queue apart).
      // Several members on one process keep the default priority: with their kernels on one device (the rehearsals) a waiting
      // high-priority queue slows the dispatch of everybody else's kernels (4 members on one GPU: 0.46 -> 1.16 ms per step).
      int prLow = 0, prHigh = 0;
      GCHK_HIP(g, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
      static const int prEnv = getenv("GTX_GROUP_XS_PRIORITY") ? atoi(getenv("GTX_GROUP_XS_PRIORITY")) : -1;    // 0 / 1 override
      const bool high = prEnv >= 0 ? prEnv != 0 : nl == 1;
      GCHK_HIP(g, hipStreamCreateWithPriority(&g->xs[li], hipStreamNonBlocking, high ? prHigh : 0));
      for (int k = 0; k < kSlots; k++) {
        GCHK_HIP(g, hipEventCreateWithFlags(&g->evFinal[k][li], hipEventDisableTiming)); GCHK_HIP(g, hipEventCreateWithFlags(&g->evXchg[k][li], hipEventDisableTiming));
        GCHK_HIP(g, hipEventCreateWithFlags(&g->evReady[k][li], hipEventDisableTiming));
      }
      // a process with one member looks for streams that run side by side; several members on one device (rehearsals) take them as they come
      static const bool probeOff = getenv("GTX_GROUP_QUEUE_PROBE") && atoi(getenv("GTX_GROUP_QUEUE_PROBE")) == 0;
      if (nl == 1 && !probeOff && g->nks > 1) {
        hipStream_t got[GTXI_SHARE_STREAMS] = {};
        GCHK_HIP(g, make_overlapping_streams(g->nks, got));
        for (int k = 0; k < g->nks; k++) g->ks[k][li] = got[k];
      } else
        for (int k = 0; k < g->nks; k++) GCHK_HIP(g, hipStreamCreateWithFlags(&g->ks[k][li], hipStreamNonBlocking));
      for (int k = 0; k <= g->nks; k++) GCHK_HIP(g, hipEventCreateWithFlags(&g->evJoin[k][li], hipEventDisableTiming));
    }
  }
  const long long call = g->seq++;
  const int slot = (int)(call % kSlots);
  const int l0 = g->local(0);
  // g->direct: the pieces travel run by run to their places in d_hits, member 0's own regions are written there by its finalize step;
  // otherwise pieces of compact vector `slot`, which member 0 reorders into d_hits
  const bool direct = g->direct;
  std::vector<void *> piece(nl, nullptr);
  std::vector<char> onXs(nl, 0);                                // the member's part of the exchange goes through its exchange stream
  g->lastRan.assign(nl, nullptr);
  for (size_t li = 0; li < nl; li++) {
    int64_t len = 0;
    const int mem = g->member((int)li);
    g->memberReads[mem] = n_reads[li];
    GCHK_HIP(g, hipSetDevice(g->dev[li]));
    void *directOut = direct && mem == 0 ? d_hits : nullptr;
    // what of this member's call goes through its exchange stream: its piece on the way to member 0 (a communicator, or the copies of a
    // rehearsal), and on member 0 the reordering of the compact vector.  Otherwise (member 0 in direct mode; a rank without a
    // communicator: the measurement mode GTX_GROUP_NO_EXCHANGE) the exchange stream stays out of it.
    const bool sends = mem != 0 && (!g->comm.empty() || g->rehearse);
    onXs[li] = direct ? (sends || (mem == 0 && g->selfExchange && !g->comm.empty())) : (mem == 0 || sends);
    // Reads in stream order: kernel and finalize step go to one of the group's streams of this member in turn, histogram set with
    // the stream (GTX_GROUP_PIPELINE=0: always the first stream -- every call behind the one before, for A/B runs).  The stream
    // waits for the caller's stream as it stands now (the reads are resident behind it) and for the exchange that last read the
    // member's piece of compact vector `slot` (call k - GTXI_SHARE_SLOTS) -- device-side waits, both long over when the stream gets
    // there in a run of calls; the caller's stream is not held up, and nothing orders the kernel of this call behind the kernel of
    // the call before: its waves take the slots that one's tail frees.
    // A batch in no order (or GTX_GROUP_ASYNC_FINALIZE=0): everything on the member's own stream.
    static const bool asyncOff = getenv("GTX_GROUP_ASYNC_FINALIZE") && atoi(getenv("GTX_GROUP_ASYNC_FINALIZE")) == 0;
    hipStream_t run = gtxi_stream(g->ctx[li]);
    if (!asyncOff && (flags & GTX_READS_SORTED) && !(flags & GTX_ZERO_LENGTH_OK)) {
      const int q = pipeOffNow() ? 0 : (int)(call % g->nks);
      run = g->ks[q][li];
      // (a wait that is already over is not enqueued: an event pair between two streams costs the host 2.5-10 us and the command
      // processor ~5, a query 0.1 -- scripts/api_cost.hip -- and a member's whole call is to fit 30)
      if (hipStreamQuery(gtxi_stream(g->ctx[li])) != hipSuccess) {
        GCHK_HIP(g, hipEventRecord(g->evReady[slot][li], gtxi_stream(g->ctx[li])));
        GCHK_HIP(g, hipStreamWaitEvent(run, g->evReady[slot][li], 0));
      }
      if (g->xchgUsed[slot] && hipEventQuery(g->evXchg[slot][li]) != hipSuccess) GCHK_HIP(g, hipStreamWaitEvent(run, g->evXchg[slot][li], 0));
      GCHK_CTX(g, li, gtxi_count_device_share_async(g->ctx[li], d_reads[li], d_weights ? d_weights[li] : nullptr, n_reads[li], flags & ~GTX_CHECK_SORTED, slot, q,
                                                     run, directOut, &piece[li], &len));
      g->lastAsync = true;
    } else {
      if (g->xchgUsed[slot]) GCHK_HIP(g, hipStreamWaitEvent(run, g->evXchg[slot][li], 0));
      GCHK_CTX(g, li, gtxi_count_device_share(g->ctx[li], d_reads[li], d_weights ? d_weights[li] : nullptr, n_reads[li], flags & ~GTX_CHECK_SORTED, slot, directOut, &piece[li], &len));
      g->lastAsync = false;
    }
    g->lastRan[li] = run;
    if (onXs[li]) {
      GCHK_HIP(g, hipEventRecord(g->evFinal[slot][li], run));
      // (a rehearsal's copies all run on member 0's exchange stream)
      GCHK_HIP(g, hipStreamWaitEvent(g->rehearse && l0 >= 0 ? g->xs[l0] : g->xs[li], g->evFinal[slot][li], 0));
    }
  }
  if (direct) {
    rc = gather_runs(g, piece, (unsigned long long *)d_hits, g->xs); if (rc) return rc;
  } else {
    unsigned long long *root = l0 >= 0 ? (unsigned long long *)gtxi_out_buffer(g->ctx[l0]) + (size_t)slot * (size_t)g->nRefs : nullptr;
    rc = gather_pieces(g, piece, root, g->xs); if (rc) return rc;
    rc = unpermute(g, root, d_hits, l0 >= 0 ? g->xs[l0] : nullptr); if (rc) return rc;
  }
  // the event behind which the member's piece of compact vector `slot` may be overwritten: the end of its part of the exchange, or of
  // its finalize step when nothing of it travels
  for (size_t li = 0; li < nl; li++) {
    GCHK_HIP(g, hipSetDevice(g->dev[li]));
    const bool recvs = (int)li == l0 && (g->rehearse || !g->comm.empty());         // member 0's exchange stream receives (or copies)
    if (onXs[li] || recvs) GCHK_HIP(g, hipEventRecord(g->evXchg[slot][li], g->xs[li]));
    else if (!(direct && (int)li == l0)) GCHK_HIP(g, hipEventRecord(g->evXchg[slot][li], g->lastRan[li]));
    else continue;                                               // (member 0 in direct mode with nothing to receive: no piece of a compact vector in use)
  }
  g->xchgUsed[slot] = true;
  g->resultPending = true;
  return GTX_OK;
}

// everything the group's own streams of the local members hold, joined into the members' own streams (device-side waits)
static int join_streams(gtx_group *g)
{
  if (g->xs.empty()) return GTX_OK;
  for (size_t li = 0; li < g->ctx.size(); li++) {
    GCHK_HIP(g, hipSetDevice(g->dev[li]));
    hipStream_t own = gtxi_stream(g->ctx[li]);
    for (int k = 0; k <= g->nks; k++) {
      hipStream_t s = k < g->nks ? g->ks[k][li] : g->xs[li];
      if (hipStreamQuery(s) == hipSuccess) continue;
      GCHK_HIP(g, hipEventRecord(g->evJoin[k][li], s));
      GCHK_HIP(g, hipStreamWaitEvent(own, g->evJoin[k][li], 0));
    }
  }
  g->resultPending = false;
  return GTX_OK;
}

int gtx_group_wait_result(gtx_group *g)
{
  if (!g) return GTX_E_ARG;
  if (g->seq == 0 || !g->resultPending) return GTX_OK;
  return join_streams(g);
}

int gtx_group_sync(gtx_group *g)
{
  if (!g) return GTX_E_ARG;
  for (size_t li = 0; li < g->ctx.size(); li++) {
    GCHK_CTX(g, li, gtx_sync(g->ctx[li]));
    if (li < g->xs.size() && g->xs[li]) {
      GCHK_HIP(g, hipSetDevice(g->dev[li]));
      for (int k = 0; k < g->nks; k++) GCHK_HIP(g, hipStreamSynchronize(g->ks[k][li]));
      GCHK_HIP(g, hipStreamSynchronize(g->xs[li]));
    }
  }
  return GTX_OK;
}

int gtx_group_last_info(gtx_group *g, gtx_count_info *info)
{
  if (!g || !info) return GTX_E_ARG;
  gtx_count_info tot; tot.first_unsorted = -1; tot.first_degenerate = -1; tot.n_no_class = 0; tot.n_degenerate = 0; tot.n_unplaced = 0;
  for (size_t li = 0; li < g->ctx.size(); li++) {
    gtx_count_info one;
    if (g->lastAsync) {                                          // (the member's finalize step ran on one of the group's streams)
      if (li < g->lastRan.size() && g->lastRan[li]) { GCHK_HIP(g, hipSetDevice(g->dev[li])); GCHK_HIP(g, hipStreamSynchronize(g->lastRan[li])); }
      GCHK_CTX(g, li, gtxi_last_share_info(g->ctx[li], &one));
    } else
    GCHK_CTX(g, li, gtx_last_info(g->ctx[li], &one));
    tot.n_no_class += one.n_no_class; tot.n_degenerate += one.n_degenerate; tot.n_unplaced += one.n_unplaced;
  }
  *info = tot;
  return GTX_OK;
}

}  // extern "C"

// ---- routing -------------------------------------------------------------------------------------------------------------
// A batch is cut into maximal runs of reads with one owner (sorted input: a handful per batch) and every run goes to its
// member as it is -- a contiguous piece of the caller's buffer.  Input whose owners interleave (more than kMaxRuns runs)
// is partitioned on the host by a few threads, order kept inside every member's share.
static constexpr size_t kMaxRuns = 4096;

static inline int owner_of(const gtx_group *g, int32_t cls) { return (uint32_t)cls < g->owner.size() ? g->owner[cls] : 0; }

struct Run { int64_t a, b; int owner; };

// returns false when the owners interleave too finely (more than kMaxRuns runs): the caller partitions instead
static bool find_runs(const gtx_group *g, const int32_t *tri, int64_t n, std::vector<Run> *runs)
{
  std::atomic<bool> hopeless(false);
  // pieces scanned in parallel, seams merged
  const int T = (int)std::min<int64_t>(8, n / (1 << 20) + 1);
  std::vector<std::vector<Run>> part(T);
  auto scan = [&](int t) {
    const int64_t a = n * t / T, b = n * (t + 1) / T;
    std::vector<Run> &r = part[t];
    int64_t i = a;
    while (i < b) {
      const int32_t cls = tri[3 * i]; const int ow = owner_of(g, cls);
      int64_t j = i + 1;
      while (j < b && (tri[3 * j] == cls || owner_of(g, tri[3 * j]) == ow)) j++;
      if (!r.empty() && r.back().owner == ow) r.back().b = j; else r.push_back({i, j, ow});
      if (r.size() > kMaxRuns) { hopeless = true; return; }
      i = j;
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back(scan, t);
  scan(0);
  for (auto &x : th) x.join();
  runs->clear();
  for (int t = 0; t < T; t++)
    for (const Run &r : part[t]) {
      if (!runs->empty() && runs->back().owner == r.owner && runs->back().b == r.a) runs->back().b = r.b; else runs->push_back(r);
    }
  return !hopeless && runs->size() <= kMaxRuns;
}

// interleaved owners: every member's reads gathered in stream order -- two passes over T slices of the batch (count per
// (slice, member), then every slice copies into its own ranges of the members' buffers)
static void partition_by_owner(gtx_group *g, const int32_t *tri, const int32_t *w, int64_t n)
{
  const int nm = g->nm;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(8, n / (1 << 18)));
  std::vector<std::vector<int64_t>> cnt(T, std::vector<int64_t>(nm, 0));
  auto slice = [&](int t, bool place, const std::vector<std::vector<int64_t>> *at) {
    const int64_t a = n * t / T, b = n * (t + 1) / T;
    std::vector<int64_t> pos(nm, 0);
    if (place) pos = (*at)[t];
    for (int64_t i = a; i < b; i++) {
      const int m = owner_of(g, tri[3 * i]);
      if (!place) { cnt[t][m]++; continue; }
      int32_t *dst = g->partTri[m].data() + 3 * pos[m];
      dst[0] = tri[3 * i]; dst[1] = tri[3 * i + 1]; dst[2] = tri[3 * i + 2];
      if (w) g->partW[m][pos[m]] = w[i];
      pos[m]++;
    }
  };
  { std::vector<std::thread> th; for (int t = 1; t < T; t++) th.emplace_back(slice, t, false, nullptr); slice(0, false, nullptr); for (auto &x : th) x.join(); }
  std::vector<std::vector<int64_t>> at(T, std::vector<int64_t>(nm, 0));
  for (int m = 0; m < nm; m++) {
    int64_t run = 0;
    for (int t = 0; t < T; t++) { at[t][m] = run; run += cnt[t][m]; }
    g->partTri[m].resize((size_t)run * 3); g->partW[m].resize(w ? (size_t)run : 0);
  }
  { std::vector<std::thread> th; for (int t = 1; t < T; t++) th.emplace_back(slice, t, true, &at); slice(0, true, &at); for (auto &x : th) x.join(); }
}

// Every member's DMA out of the caller's previous page-locked batch must be done before the caller may refill it: the members
// that get a piece of THIS batch wait by themselves (stage_batches), the others would not -- include/gtx.h promises the
// caller its buffer back "when the next call has returned".
static int wait_direct_all(gtx_group *g)
{
  for (size_t li = 0; li < g->ctx.size(); li++) GCHK_CTX(g, li, gtxi_wait_direct(g->ctx[li]));
  return GTX_OK;
}

template <class Add>
static int route(gtx_group *g, const int32_t *tri, const int32_t *w, int64_t n, Add add)
{
  const int nm = g->nm;
  int rc = wait_direct_all(g); if (rc) return rc;
  if (n <= 0) return GTX_OK;
  if (nm == 1) { g->memberReads[0] += n; return add(0, tri, w, n); }
  std::vector<Run> runs;
  if (find_runs(g, tri, n, &runs)) {
    for (const Run &r : runs) { g->memberReads[r.owner] += r.b - r.a; rc = add(r.owner, tri + 3 * r.a, w ? w + r.a : nullptr, r.b - r.a); if (rc) return rc; }
    return GTX_OK;
  }
  partition_by_owner(g, tri, w, n);
  for (int m = 0; m < nm; m++) {
    const int64_t cnt = (int64_t)(g->partTri[m].size() / 3);
    if (!cnt) continue;
    g->memberReads[m] += cnt;
    rc = add(m, g->partTri[m].data(), w ? g->partW[m].data() : nullptr, cnt); if (rc) return rc;
  }
  return GTX_OK;
}

// legacy finish (coverage; count on a reference set with sorted-merge semantics, whose inverted intervals are matched into the
// full vector): RCCL reduce(sum) of the members' full uint64 vectors to member 0, each on its own stream
static int reduce_to_root(gtx_group *g, std::vector<void *> &d, int64_t count)
{
  const int n = (int)g->ctx.size();
  if (g->rehearse && count > 0) {
    for (int i = 1; i < n; i++) {
      GCHK_HIP(g, hipStreamSynchronize(gtxi_stream(g->ctx[i])));
      rehearse_add_kernel<<<(unsigned)((count + 255) / 256), 256, 0, gtxi_stream(g->ctx[0])>>>((unsigned long long *)d[0], (const unsigned long long *)d[i], count);
    }
    GCHK_HIP(g, hipGetLastError());
    return GTX_OK;
  }
  if (g->comm.empty() || count <= 0) return GTX_OK;
  GCHK_NCCL(g, g->rccl.GroupStart());
  for (int i = 0; i < n; i++) {
    ncclResult_t r = g->rccl.Reduce(d[i], d[0], (size_t)count, ncclUint64, ncclSum, 0, g->comm[i], gtxi_stream(g->ctx[i]));
    if (r != ncclSuccess) { g->rccl.GroupEnd(); g->err = std::string("ncclReduce: ") + g->rccl.GetErrorString(r); return GTX_E_HIP; }
  }
  GCHK_NCCL(g, g->rccl.GroupEnd());
  return GTX_OK;
}

// the host-buffer calls and the scans work on the members' own streams and on compact vector 0: behind any exchange of an
// earlier gtx_group_count_device that is still on its way (device-side waits)
static int join_streams(gtx_group *g);
static int wait_exchanges(gtx_group *g) { return join_streams(g); }

static int finish(gtx_group *g, bool coverage, uint64_t *out, gtx_count_info *info)
{
  const int n = (int)g->ctx.size();
  { int rc = wait_exchanges(g); if (rc) return rc; }
  std::vector<void *> d(n, nullptr);
  // count on a plain reference set: every member finalizes its own classes into its piece of the compact vector
  bool pairs = false;                                            // multi-interval regions: corrections go into a member's whole vector
  for (int i = 0; i < n; i++) pairs = pairs || gtxi_pairs_on(g->ctx[i]);
  const bool pieces = !coverage && !(g->refFlags & GTX_REFS_KEEP_ZERO_LENGTH) && !pairs && !g->fullVectors;
  if (pieces) { int rc = ensure_plan(g); if (rc) return rc; }
  for (int i = 0; i < n; i++) GCHK_CTX(g, i, coverage ? gtxi_coverage_finish(g->ctx[i], &d[i]) : gtxi_count_finish(g->ctx[i], &d[i], pieces ? 1 : 0));
  GCHK_HIP(g, hipSetDevice(g->dev[0]));
  if (pieces) {
    unsigned long long *root = (unsigned long long *)gtxi_out_buffer(g->ctx[0]);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; i++) st[i] = gtxi_stream(g->ctx[i]);
    int rc = gather_pieces(g, d, root, st); if (rc) return rc;
    // compact -> file order on the way out: the host puts the copy in order (no second device vector)
    if (g->nRefs > 0) {
      std::vector<uint64_t> compact((size_t)g->nRefs);
      GCHK_HIP(g, hipMemcpyAsync(compact.data(), root, sizeof(uint64_t) * g->nRefs, hipMemcpyDeviceToHost, gtxi_stream(g->ctx[0])));
      GCHK_HIP(g, hipStreamSynchronize(gtxi_stream(g->ctx[0])));
      for (int64_t j = 0; j < g->nRefs; j++) out[g->perm[j]] = compact[j];
    }
  } else {
    int rc = reduce_to_root(g, d, g->nRefs); if (rc) return rc;
    if (g->nRefs > 0) GCHK_HIP(g, hipMemcpyAsync(out, d[0], sizeof(uint64_t) * g->nRefs, hipMemcpyDeviceToHost, gtxi_stream(g->ctx[0])));
  }
  gtx_count_info tot; tot.first_unsorted = -1; tot.first_degenerate = -1; tot.n_no_class = 0; tot.n_degenerate = 0; tot.n_unplaced = 0;
  for (int i = 0; i < n; i++) {
    GCHK_CTX(g, i, gtx_sync(g->ctx[i]));
    gtx_count_info one; gtxi_fetch_info(g->ctx[i], &one);
    tot.n_no_class += one.n_no_class; tot.n_degenerate += one.n_degenerate; tot.n_unplaced += one.n_unplaced;
  }
  if (info) *info = tot;
  return GTX_OK;
}

#define NEED_ALL_LOCAL(g, what) do { if ((g)->rank >= 0 && (g)->nm > 1) return gfail(g, GTX_E_STATE, what ": the host-buffer calls need a group that holds all its members (gtx_group_create); a rank of a multi-process group has the *_device calls"); } while (0)

extern "C" {

int gtx_group_count_begin(gtx_group *g)
{
  if (!g) return GTX_E_ARG;
  NEED_ALL_LOCAL(g, "gtx_group_count_begin");
  { int rcw = wait_exchanges(g); if (rcw) return rcw; }            // (device calls still finalizing / exchanging: their output vector is about to be reused)
  g->lastAsync = false; g->fullVectors = false;
  for (size_t i = 0; i < g->ctx.size(); i++) GCHK_CTX(g, i, gtx_count_begin(g->ctx[i]));
  std::fill(g->memberReads.begin(), g->memberReads.end(), 0);
  g->countOpen = true;
  return GTX_OK;
}

int gtx_group_count_add(gtx_group *g, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags)
{
  if (!g) return GTX_E_ARG;
  if (!g->countOpen) return gfail(g, GTX_E_STATE, "gtx_group_count_add: gtx_group_count_begin has not been called");
  if (n < 0 || (n > 0 && !reads)) return gfail(g, GTX_E_ARG, "gtx_group_count_add: bad argument");
  flags &= ~GTX_CHECK_SORTED;                                   // positions in a member's share are not positions in the stream
  return route(g, reads, weights, n, [&](int m, const int32_t *r, const int32_t *w, int64_t cnt) -> int {
    GCHK_CTX(g, m, gtx_count_add(g->ctx[m], r, w, cnt, flags));
    return GTX_OK;
  });
}

// Region text tokenised on the device (gtx_count_add_text) in a group: a block of the stream goes to the members in turn, whatever
// the classes of its lines -- the read stream is split evenly, every member holds the whole reference set -- so the call ends with the
// sum of the members' full vectors (ncclReduce) instead of pieces.  ticket = 2 x local member + the member's slot.
int gtx_group_count_add_text(gtx_group *g, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket)
{
  if (!g || !ticket) return GTX_E_ARG;
  if (!g->countOpen) return gfail(g, GTX_E_STATE, "gtx_group_count_add_text: gtx_group_count_begin has not been called");
  const int m = (int)(g->textTurn++ % g->ctx.size());
  int t = -1;
  GCHK_CTX(g, m, gtx_count_add_text(g->ctx[m], text, bytes, n_lines, rules, flags & ~GTX_CHECK_SORTED, &t));
  if (g->ctx.size() > 1) g->fullVectors = true;
  g->memberReads[g->member(m)] += n_lines;
  *ticket = 2 * m + t;
  return GTX_OK;
}

int gtx_group_coverage_add_text(gtx_group *g, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket)
{
  if (!g || !ticket) return GTX_E_ARG;
  if (!g->coverOpen) return gfail(g, GTX_E_STATE, "gtx_group_coverage_add_text: gtx_group_coverage_begin has not been called");
  const int m = (int)(g->textTurn++ % g->ctx.size());
  int t = -1;
  GCHK_CTX(g, m, gtx_coverage_add_text(g->ctx[m], text, bytes, n_lines, rules, flags, &t));
  g->memberReads[g->member(m)] += n_lines;
  *ticket = 2 * m + t;
  return GTX_OK;
}

int gtx_group_text_result(gtx_group *g, int ticket, int *needs_host)
{
  if (!g || !needs_host || ticket < 0 || (size_t)(ticket >> 1) >= g->ctx.size()) return g ? gfail(g, GTX_E_ARG, "gtx_group_text_result: bad argument") : GTX_E_ARG;
  GCHK_CTX(g, ticket >> 1, gtx_text_result(g->ctx[ticket >> 1], ticket & 1, needs_host));
  return GTX_OK;
}

int gtx_group_set_ref_blocks(gtx_group *g, const int64_t *first, const int32_t *blocks)
{
  if (!g) return GTX_E_ARG;
  for (size_t i = 0; i < g->ctx.size(); i++) GCHK_CTX(g, i, gtx_set_ref_blocks(g->ctx[i], first, blocks));
  return GTX_OK;
}

// multi-interval queries are a side channel (one lane per query against the whole reference set, which every member holds):
// they go to the members in turn, whatever their class
int gtx_group_count_add_regions(gtx_group *g, const int32_t *env, const int32_t *weights, const int64_t *first, const int32_t *blocks, int64_t n)
{
  if (!g) return GTX_E_ARG;
  if (!g->countOpen) return gfail(g, GTX_E_STATE, "gtx_group_count_add_regions: gtx_group_count_begin has not been called");
  if (n <= 0) return n < 0 ? gfail(g, GTX_E_ARG, "gtx_group_count_add_regions: bad argument") : GTX_OK;
  const int m = (int)(g->regionsTurn++ % g->ctx.size());
  GCHK_CTX(g, m, gtx_count_add_regions(g->ctx[m], env, weights, first, blocks, n));
  return GTX_OK;
}

int gtx_group_count_end(gtx_group *g, uint64_t *hits, gtx_count_info *info)
{
  if (!g) return GTX_E_ARG;
  if (!g->countOpen) return gfail(g, GTX_E_STATE, "gtx_group_count_end: gtx_group_count_begin has not been called");
  if (g->nRefs > 0 && !hits) return gfail(g, GTX_E_ARG, "gtx_group_count_end: null output");
  g->countOpen = false;
  return finish(g, false, hits, info);
}

int gtx_group_coverage_begin(gtx_group *g)
{
  if (!g) return GTX_E_ARG;
  NEED_ALL_LOCAL(g, "gtx_group_coverage_begin");
  for (size_t i = 0; i < g->ctx.size(); i++) GCHK_CTX(g, i, gtx_coverage_begin(g->ctx[i]));
  std::fill(g->memberReads.begin(), g->memberReads.end(), 0);
  g->coverOpen = true;
  return GTX_OK;
}

int gtx_group_coverage_add(gtx_group *g, const int32_t *reads, const int32_t *weights, int64_t n, uint32_t flags)
{
  if (!g) return GTX_E_ARG;
  if (!g->coverOpen) return gfail(g, GTX_E_STATE, "gtx_group_coverage_add: gtx_group_coverage_begin has not been called");
  if (n < 0 || (n > 0 && !reads)) return gfail(g, GTX_E_ARG, "gtx_group_coverage_add: bad argument");
  return route(g, reads, weights, n, [&](int m, const int32_t *r, const int32_t *w, int64_t cnt) -> int {
    GCHK_CTX(g, m, gtx_coverage_add(g->ctx[m], r, w, cnt, flags));
    return GTX_OK;
  });
}

int gtx_group_coverage_end(gtx_group *g, uint64_t *cov, gtx_count_info *info)
{
  if (!g) return GTX_E_ARG;
  if (!g->coverOpen) return gfail(g, GTX_E_STATE, "gtx_group_coverage_end: gtx_group_coverage_begin has not been called");
  if (g->nRefs > 0 && !cov) return gfail(g, GTX_E_ARG, "gtx_group_coverage_end: null output");
  g->coverOpen = false;
  return finish(g, true, cov, info);
}

int gtx_group_member_reads(const gtx_group *g, int64_t *reads_out)
{
  if (!g || !reads_out) return GTX_E_ARG;
  for (int i = 0; i < g->nm; i++) reads_out[i] = g->memberReads[i];
  return GTX_OK;
}

}  // extern "C"

// ---- scans ---------------------------------------------------------------------------------------------------------------
// A member scans the classes it owns into a PACKED vector of its own (the other classes get length 0: no micro-windows, no
// windows, no memset of theirs), and the per-class pieces travel to member 0's vector in the caller's layout.
struct ScanShare { std::vector<int32_t> len; std::vector<int64_t> off; int64_t extent = 0; };

static void scan_assign(gtx_group *g, const int32_t *class_len, int32_t n_classes)
{
  if ((int32_t)g->owner.size() == n_classes) return;
  std::vector<int64_t> load(n_classes);                          // no assignment given for these classes: by class length
  for (int32_t c = 0; c < n_classes; c++) load[c] = class_len[c] > 0 ? class_len[c] : 0;
  g->owner.resize(n_classes);
  gtx_lpt_assign(load.data(), n_classes, g->nm, g->owner.data());
  g->planValid = false;
}

static ScanShare scan_share(const gtx_group *g, int mem, const int32_t *class_len, int32_t n_classes, int32_t step, int32_t size)
{
  ScanShare s; s.len.assign(n_classes, 0); s.off.assign(n_classes, 0);
  for (int32_t c = 0; c < n_classes; c++) {
    if (g->owner[c] != mem) continue;
    s.len[c] = class_len[c]; s.off[c] = s.extent;
    s.extent += gtx_scan_n_windows(class_len[c] < 0 ? 0 : class_len[c], step, size);
  }
  return s;
}

// the per-class pieces of the members' packed vectors (piece[li], layout share[li]) to root (layout class_offsets) on member 0
static int gather_windows(gtx_group *g, const std::vector<void *> &piece, const std::vector<ScanShare> &share, const int32_t *class_len, int32_t n_classes,
                          int32_t step, int32_t size, const int64_t *class_offsets, unsigned long long *root)
{
  const int l0 = g->local(0);
  auto nwin = [&](int32_t c) { return gtx_scan_n_windows(class_len[c] < 0 ? 0 : class_len[c], step, size); };
  if (l0 >= 0) GCHK_HIP(g, hipSetDevice(g->dev[l0]));
  // member 0's own classes: copies on its device
  if (l0 >= 0)
    for (int32_t c = 0; c < n_classes; c++) {
      if (g->owner[c] != 0 || nwin(c) == 0) continue;
      GCHK_HIP(g, hipMemcpyAsync(root + class_offsets[c], (const unsigned long long *)piece[l0] + share[l0].off[c], sizeof(uint64_t) * (size_t)nwin(c),
                                 hipMemcpyDeviceToDevice, gtxi_stream(g->ctx[l0])));
    }
  if (g->rehearse) {
    for (size_t li = 0; li < g->ctx.size(); li++) {
      const int mem = g->member((int)li);
      if (mem == 0) continue;
      GCHK_HIP(g, hipEventRecord(g->evPiece, gtxi_stream(g->ctx[li])));
      GCHK_HIP(g, hipStreamWaitEvent(gtxi_stream(g->ctx[l0]), g->evPiece, 0));
      for (int32_t c = 0; c < n_classes; c++)
        if (g->owner[c] == mem && nwin(c) > 0)
          GCHK_HIP(g, hipMemcpyAsync(root + class_offsets[c], (const unsigned long long *)piece[li] + share[li].off[c], sizeof(uint64_t) * (size_t)nwin(c),
                                     hipMemcpyDeviceToDevice, gtxi_stream(g->ctx[l0])));
    }
    return GTX_OK;
  }
  if (g->comm.empty()) return GTX_OK;
  GCHK_NCCL(g, g->rccl.GroupStart());
  ncclResult_t r = ncclSuccess;
  for (int32_t c = 0; c < n_classes && r == ncclSuccess; c++) {          // class order on both sides: sends and receives of a pair match up
    const int mem = g->owner[c];
    if (mem == 0 || nwin(c) == 0) continue;
    const int li = g->local(mem);
    if (li >= 0) r = g->rccl.Send((const unsigned long long *)piece[li] + share[li].off[c], (size_t)nwin(c), ncclUint64, 0, g->comm[li], gtxi_stream(g->ctx[li]));
    if (l0 >= 0 && r == ncclSuccess) r = g->rccl.Recv(root + class_offsets[c], (size_t)nwin(c), ncclUint64, mem, g->comm[l0], gtxi_stream(g->ctx[l0]));
  }
  if (r != ncclSuccess) { g->rccl.GroupEnd(); g->err = std::string("ncclSend/ncclRecv: ") + g->rccl.GetErrorString(r); return GTX_E_HIP; }
  GCHK_NCCL(g, g->rccl.GroupEnd());
  return GTX_OK;
}

extern "C" {

int gtx_group_scan_device(gtx_group *g, const void *const *d_reads, const void *const *d_weights, const int64_t *n_reads, const int32_t *class_len,
                          int32_t n_classes, int32_t win_step, int32_t win_size, char preprocess, uint32_t flags, void *d_windows, const int64_t *class_offsets)
{
  if (!g) return GTX_E_ARG;
  if (!d_reads || !n_reads || n_classes < 1 || !class_len || !class_offsets) return gfail(g, GTX_E_ARG, "gtx_group_scan_device: bad argument");
  if (win_step <= 0 || win_size <= 0 || win_size % win_step) return gfail(g, GTX_E_ARG, "gtx_group_scan_device: window size must be a positive multiple of window step");
  scan_assign(g, class_len, n_classes);
  { int rc = wait_exchanges(g); if (rc) return rc; }
  if (g->local(0) >= 0 && !g->evPiece) { GCHK_HIP(g, hipSetDevice(g->dev[g->local(0)])); GCHK_HIP(g, hipEventCreateWithFlags(&g->evPiece, hipEventDisableTiming)); }
  std::vector<ScanShare> share(g->ctx.size());
  std::vector<void *> piece(g->ctx.size(), nullptr);
  for (size_t li = 0; li < g->ctx.size(); li++) {
    const int mem = g->member((int)li);
    share[li] = scan_share(g, mem, class_len, n_classes, win_step, win_size);
    g->memberReads[mem] = n_reads[li];
    GCHK_CTX(g, li, gtxi_ensure_out(g->ctx[li], share[li].extent));
    piece[li] = gtxi_out_buffer(g->ctx[li]);
    GCHK_CTX(g, li, gtx_scan_device(g->ctx[li], d_reads[li], d_weights ? d_weights[li] : nullptr, n_reads[li], share[li].len.data(), n_classes, win_step, win_size,
                                    preprocess, flags, piece[li], share[li].off.data()));
  }
  if (g->local(0) >= 0 && !d_windows) {
    int64_t extent = 0;
    for (int32_t c = 0; c < n_classes; c++) extent += gtx_scan_n_windows(class_len[c] < 0 ? 0 : class_len[c], win_step, win_size);
    if (extent > 0) return gfail(g, GTX_E_ARG, "gtx_group_scan_device: member 0 needs the output vector");
  }
  return gather_windows(g, piece, share, class_len, n_classes, win_step, win_size, class_offsets, (unsigned long long *)d_windows);
}

int gtx_group_scan(gtx_group *g, const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *class_len, int32_t n_classes,
                   int32_t win_step, int32_t win_size, char preprocess, uint32_t flags, uint64_t *windows_out, const int64_t *class_offsets)
{
  if (!g) return GTX_E_ARG;
  NEED_ALL_LOCAL(g, "gtx_group_scan");
  if (n < 0 || (n > 0 && !reads) || n_classes < 1 || !class_len || !class_offsets) return gfail(g, GTX_E_ARG, "gtx_group_scan: bad argument");
  const int nm = g->nm;
  { int rc = wait_direct_all(g); if (rc) return rc; }
  if (nm == 1) {
    g->memberReads[0] = n;
    void *d = nullptr; int64_t extent = 0;
    GCHK_CTX(g, 0, gtxi_scan_enqueue(g->ctx[0], reads, weights, n, class_len, n_classes, win_step, win_size, preprocess, flags, class_offsets, &d, &extent));
    if (extent > 0 && !windows_out) return gfail(g, GTX_E_ARG, "gtx_group_scan: null output");
    GCHK_HIP(g, hipSetDevice(g->dev[0]));
    if (extent > 0) GCHK_HIP(g, hipMemcpyAsync(windows_out, d, sizeof(uint64_t) * extent, hipMemcpyDeviceToHost, gtxi_stream(g->ctx[0])));
    GCHK_CTX(g, 0, gtx_sync(g->ctx[0]));
    return GTX_OK;
  }
  scan_assign(g, class_len, n_classes);
  { int rc = wait_exchanges(g); if (rc) return rc; }
  if (!g->evPiece) { GCHK_HIP(g, hipSetDevice(g->dev[0])); GCHK_HIP(g, hipEventCreateWithFlags(&g->evPiece, hipEventDisableTiming)); }
  std::fill(g->memberReads.begin(), g->memberReads.end(), 0);
  // every member scans its share (a scan is one call per member: the shares are gathered first)
  std::vector<Run> runs;
  bool contiguous = true;
  if (n > 0) {
    contiguous = find_runs(g, reads, n, &runs);
    std::vector<int> seen(nm, 0);
    for (const Run &r : runs) { if (seen[r.owner]++) contiguous = false; }
  }
  if (!contiguous) partition_by_owner(g, reads, weights, n);
  std::vector<Run> mine(nm, Run{0, 0, 0});
  if (contiguous) for (const Run &r : runs) mine[r.owner] = r;
  std::vector<ScanShare> share(nm);
  std::vector<void *> piece(nm, nullptr);
  int64_t extent = 0, total = 0;
  for (int32_t c = 0; c < n_classes; c++) {
    const int64_t w = gtx_scan_n_windows(class_len[c] < 0 ? 0 : class_len[c], win_step, win_size);
    extent = std::max<int64_t>(extent, class_offsets[c] + w); total += w;
  }
  if (total > 0 && !windows_out) return gfail(g, GTX_E_ARG, "gtx_group_scan: null output");
  for (int m = 0; m < nm; m++) {
    share[m] = scan_share(g, m, class_len, n_classes, win_step, win_size);
    const int32_t *r = contiguous ? reads + 3 * mine[m].a : g->partTri[m].data();
    const int32_t *w = !weights ? nullptr : contiguous ? weights + mine[m].a : g->partW[m].data();
    const int64_t cnt = contiguous ? mine[m].b - mine[m].a : (int64_t)(g->partTri[m].size() / 3);
    g->memberReads[m] = cnt;
    int64_t ext = 0;
    GCHK_CTX(g, m, gtxi_scan_enqueue(g->ctx[m], r, w, cnt, share[m].len.data(), n_classes, win_step, win_size, preprocess, flags, share[m].off.data(), &piece[m], &ext));
  }
  // member 0 assembles the caller's layout in a second vector of its own, then the classes go out range by range (the layout may have gaps)
  GCHK_HIP(g, hipSetDevice(g->dev[0]));
  unsigned long long *root = nullptr;
  GCHK_CTX(g, 0, gtxi_scratch(g->ctx[0], sizeof(uint64_t) * (size_t)std::max<int64_t>(extent, 1), (void **)&root));
  int rc = gather_windows(g, piece, share, class_len, n_classes, win_step, win_size, class_offsets, root); if (rc) return rc;
  for (int32_t c = 0; c < n_classes; c++) {
    const int64_t w = gtx_scan_n_windows(class_len[c] < 0 ? 0 : class_len[c], win_step, win_size);
    if (w > 0) GCHK_HIP(g, hipMemcpyAsync(windows_out + class_offsets[c], root + class_offsets[c], sizeof(uint64_t) * (size_t)w, hipMemcpyDeviceToHost, gtxi_stream(g->ctx[0])));
  }
  for (int i = 0; i < nm; i++) GCHK_CTX(g, i, gtx_sync(g->ctx[i]));
  return GTX_OK;
}

}  // extern "C"
