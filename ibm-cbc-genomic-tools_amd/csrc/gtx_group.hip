// gtx_group.hip -- the counting path on several MI355X of one node (gtx_group_* of include/gtx.h).
//
// Two regions can overlap only on one chromosome [and strand] (GenomicInterval::OverlapsWith,
// gtools/genomic_intervals.cpp:624-630), so the reads and regions of one class are an independent unit of work: classes
// are dealt to the GPUs (longest-processing-time packing of a per-class load), every GPU holds the whole reference set and
// counts the reads of its classes into a full-length vector, and ONE RCCL reduce(sum) of that uint64 vector over xGMI
// yields the result -- the vectors are disjoint by class, so the sum is the single-GPU vector bit for bit.  The sliding
// windows of genomic_scans are per chromosome and strand as well: the same split, the same reduce of the window vector.
// One process drives all devices (the reference's tools are single processes): a context per device, asynchronous
// enqueues from the caller's thread, librccl resolved at run time (only a group of more than one device needs it).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
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
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string *err)
  {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { *err = std::string("gtx_group: cannot load librccl: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !GetErrorString) { *err = "gtx_group: librccl lacks an expected symbol"; return false; }
    return true;
  }
};

thread_local std::string g_group_create_error;

// rehearsal on one device (GTX_GROUP_REHEARSE=1: all members may sit on the same GPU, which RCCL refuses): the members'
// vectors are summed by this kernel instead of ncclReduce.  Exercises the routing and the finish on a one-GPU box.
__global__ void rehearse_add_kernel(unsigned long long *__restrict__ root, const unsigned long long *__restrict__ other, long long n)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) root[i] += other[i];
}

}  // namespace

struct gtx_group {
  std::vector<gtx_ctx *> ctx;
  std::vector<int> dev;
  std::vector<ncclComm_t> comm;            // empty for a group of one (unless GTX_GROUP_FORCE_RCCL=1)
  Rccl rccl;
  std::string err;
  std::vector<int32_t> owner;              // class -> member
  std::vector<int64_t> memberReads;        // reads routed to each member in the open call
  bool countOpen = false, coverOpen = false;
  bool rehearse = false;                   // GTX_GROUP_REHEARSE=1
  int64_t nRefs = 0;
  // scratch of the router for interleaved input
  std::vector<std::vector<int32_t>> partTri, partW;
};

#define GCHK_HIP(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e_); return GTX_E_HIP; } } while (0)
#define GCHK_NCCL(g, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (g)->err = std::string(#call) + ": " + (g)->rccl.GetErrorString(r_); return GTX_E_HIP; } } while (0)
#define GCHK_CTX(g, i, call) do { int rc_ = (call); if (rc_ != GTX_OK) { (g)->err = std::string("member ") + std::to_string(i) + ": " + gtx_last_error((g)->ctx[i]); return rc_; } } while (0)

static int gfail(gtx_group *g, int code, const char *msg) { g->err = msg; return code; }

extern "C" {

gtx_group *gtx_group_create(int n, const int *device_ids)
{
  if (n < 1) { g_group_create_error = "gtx_group_create: need at least one device"; return nullptr; }
  gtx_group *g = new gtx_group();
  { const char *rh = getenv("GTX_GROUP_REHEARSE"); g->rehearse = rh && atoi(rh); }
  for (int i = 0; i < n; i++) {
    const int d = device_ids ? device_ids[i] : i;
    for (int j = 0; j < i && !g->rehearse; j++) if (g->dev[j] == d) { g_group_create_error = "gtx_group_create: a device is listed twice"; gtx_group_destroy(g); return nullptr; }
    gtx_ctx *c = gtx_create(d);
    if (!c) { g_group_create_error = gtx_last_error(nullptr); gtx_group_destroy(g); return nullptr; }
    g->ctx.push_back(c); g->dev.push_back(d);
  }
  const char *force = getenv("GTX_GROUP_FORCE_RCCL");
  if (!g->rehearse && (n > 1 || (force && atoi(force)))) {
    if (!g->rccl.load(&g_group_create_error)) { gtx_group_destroy(g); return nullptr; }
    g->comm.resize(n);
    // RCCL announces its version on the process's stdout when the first communicator comes up; the tools' stdout is
    // their result.  The descriptor points at stderr for the duration of the call.
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) dup2(2, 1);
    ncclResult_t r = g->rccl.CommInitAll(g->comm.data(), n, g->dev.data());
    fflush(stdout);
    if (saved >= 0) { dup2(saved, 1); close(saved); }
    if (r != ncclSuccess) { g_group_create_error = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r); g->comm.clear(); gtx_group_destroy(g); return nullptr; }
  }
  g->memberReads.assign(n, 0);
  g->partTri.resize(n); g->partW.resize(n);
  return g;
}

void gtx_group_destroy(gtx_group *g)
{
  if (!g) return;
  for (ncclComm_t c : g->comm) if (c) g->rccl.CommDestroy(c);
  for (gtx_ctx *c : g->ctx) gtx_destroy(c);
  delete g;
}

int gtx_group_size(const gtx_group *g) { return g ? (int)g->ctx.size() : 0; }
gtx_ctx *gtx_group_ctx(gtx_group *g, int member) { return g && member >= 0 && member < (int)g->ctx.size() ? g->ctx[member] : nullptr; }
const char *gtx_group_last_error(const gtx_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

// longest-processing-time packing: classes by decreasing load, each to the member with the least load so far
// (ties: lowest class id first, lowest member first -- deterministic)
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

int gtx_group_assign(gtx_group *g, const int64_t *class_load, int32_t n_classes, int32_t *owner_out)
{
  if (!g || n_classes < 0 || (n_classes > 0 && !class_load)) return g ? gfail(g, GTX_E_ARG, "gtx_group_assign: bad argument") : GTX_E_ARG;
  g->owner.resize(n_classes);
  gtx_lpt_assign(class_load, n_classes, (int)g->ctx.size(), g->owner.data());
  if (owner_out) memcpy(owner_out, g->owner.data(), sizeof(int32_t) * n_classes);
  return GTX_OK;
}

int gtx_group_set_refs(gtx_group *g, const int32_t *tri, int64_t m, int32_t n_classes, uint32_t flags)
{
  if (!g) return GTX_E_ARG;
  const int n = (int)g->ctx.size();
  std::vector<int> rc(n, GTX_OK);
  std::vector<std::thread> th;                               // the host-side sorts of the members run side by side
  for (int i = 1; i < n; i++) th.emplace_back([&, i] { rc[i] = gtx_set_refs_ex(g->ctx[i], tri, m, n_classes, flags); });
  rc[0] = gtx_set_refs_ex(g->ctx[0], tri, m, n_classes, flags);
  for (auto &t : th) t.join();
  for (int i = 0; i < n; i++) GCHK_CTX(g, i, rc[i]);
  g->nRefs = m;
  if (g->owner.empty() && n_classes > 0) {
    // no assignment given: load = the span of a class's reference regions, a stand-in for the chromosome length
    int32_t nc = n_classes;
    if (nc <= 0) { for (int64_t k = 0; k < m; k++) nc = std::max(nc, tri[3 * k] + 1); }
    std::vector<int64_t> lo(nc, INT64_MAX), hi(nc, INT64_MIN), load(nc, 0);
    for (int64_t k = 0; k < m; k++) { const int32_t c = tri[3 * k]; if (c < 0 || c >= nc) continue; lo[c] = std::min<int64_t>(lo[c], tri[3 * k + 1]); hi[c] = std::max<int64_t>(hi[c], tri[3 * k + 2]); }
    for (int32_t c = 0; c < nc; c++) load[c] = hi[c] >= lo[c] ? hi[c] - lo[c] + 1 : 0;
    g->owner.resize(nc);
    gtx_lpt_assign(load.data(), nc, n, g->owner.data());
  }
  return GTX_OK;
}

}  // extern "C"

// ---- routing -------------------------------------------------------------------------------------------------------------
// A batch is cut into maximal runs of reads with one owner (sorted input: a handful per batch) and every run goes to its
// member as it is -- a contiguous piece of the caller's buffer.  Input whose owners interleave (more than kMaxRuns runs)
// is partitioned on the host, order kept inside every member's share.
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

template <class Add>
static int route(gtx_group *g, const int32_t *tri, const int32_t *w, int64_t n, Add add)
{
  const int nm = (int)g->ctx.size();
  if (n <= 0) return GTX_OK;
  if (nm == 1) { g->memberReads[0] += n; return add(0, tri, w, n); }
  std::vector<Run> runs;
  if (find_runs(g, tri, n, &runs)) {
    for (const Run &r : runs) { g->memberReads[r.owner] += r.b - r.a; int rc = add(r.owner, tri + 3 * r.a, w ? w + r.a : nullptr, r.b - r.a); if (rc) return rc; }
    return GTX_OK;
  }
  for (int m = 0; m < nm; m++) { g->partTri[m].clear(); g->partW[m].clear(); }
  for (int64_t i = 0; i < n; i++) {
    const int m = owner_of(g, tri[3 * i]);
    g->partTri[m].insert(g->partTri[m].end(), tri + 3 * i, tri + 3 * i + 3);
    if (w) g->partW[m].push_back(w[i]);
  }
  for (int m = 0; m < nm; m++) {
    const int64_t cnt = (int64_t)(g->partTri[m].size() / 3);
    if (!cnt) continue;
    g->memberReads[m] += cnt;
    int rc = add(m, g->partTri[m].data(), w ? g->partW[m].data() : nullptr, cnt); if (rc) return rc;
  }
  return GTX_OK;
}

// RCCL reduce(sum) of the members' uint64 vectors to member 0, each on its own stream
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

static int finish(gtx_group *g, bool coverage, uint64_t *out, gtx_count_info *info)
{
  const int n = (int)g->ctx.size();
  std::vector<void *> d(n, nullptr);
  for (int i = 0; i < n; i++) GCHK_CTX(g, i, coverage ? gtxi_coverage_finish(g->ctx[i], &d[i]) : gtxi_count_finish(g->ctx[i], &d[i]));
  int rc = reduce_to_root(g, d, g->nRefs); if (rc) return rc;
  GCHK_HIP(g, hipSetDevice(g->dev[0]));
  if (g->nRefs > 0) GCHK_HIP(g, hipMemcpyAsync(out, d[0], sizeof(uint64_t) * g->nRefs, hipMemcpyDeviceToHost, gtxi_stream(g->ctx[0])));
  gtx_count_info tot; tot.first_unsorted = -1; tot.first_degenerate = -1; tot.n_no_class = 0; tot.n_degenerate = 0; tot.n_unplaced = 0;
  for (int i = 0; i < n; i++) {
    GCHK_CTX(g, i, gtx_sync(g->ctx[i]));
    gtx_count_info one; gtxi_fetch_info(g->ctx[i], &one);
    tot.n_no_class += one.n_no_class; tot.n_degenerate += one.n_degenerate; tot.n_unplaced += one.n_unplaced;
  }
  if (info) *info = tot;
  return GTX_OK;
}

extern "C" {

int gtx_group_count_begin(gtx_group *g)
{
  if (!g) return GTX_E_ARG;
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
  for (size_t i = 0; i < g->ctx.size(); i++) reads_out[i] = g->memberReads[i];
  return GTX_OK;
}

int gtx_group_scan(gtx_group *g, const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *class_len, int32_t n_classes,
                   int32_t win_step, int32_t win_size, char preprocess, uint32_t flags, uint64_t *windows_out, const int64_t *class_offsets)
{
  if (!g) return GTX_E_ARG;
  if (n < 0 || (n > 0 && !reads) || n_classes < 1 || !class_len || !class_offsets) return gfail(g, GTX_E_ARG, "gtx_group_scan: bad argument");
  const int nm = (int)g->ctx.size();
  if ((int32_t)g->owner.size() != n_classes) {
    // no assignment given for these classes: by class length
    std::vector<int64_t> load(n_classes);
    for (int32_t c = 0; c < n_classes; c++) load[c] = class_len[c] > 0 ? class_len[c] : 0;
    g->owner.resize(n_classes);
    gtx_lpt_assign(load.data(), n_classes, nm, g->owner.data());
  }
  std::fill(g->memberReads.begin(), g->memberReads.end(), 0);
  // every member scans its share (a scan is one call per member: the shares are gathered first)
  std::vector<std::vector<Run>> share(nm);
  std::vector<Run> runs;
  bool contiguous = true;
  if (nm > 1 && n > 0) {
    contiguous = find_runs(g, reads, n, &runs);
    std::vector<int> seen(nm, 0);
    for (const Run &r : runs) { if (seen[r.owner]++) contiguous = false; }
  }
  std::vector<void *> d(nm, nullptr);
  int64_t extent = 0;
  if (nm == 1) {
    g->memberReads[0] = n;
    GCHK_CTX(g, 0, gtxi_scan_enqueue(g->ctx[0], reads, weights, n, class_len, n_classes, win_step, win_size, preprocess, flags, class_offsets, &d[0], &extent));
  } else if (contiguous) {
    std::vector<Run> mine(nm, Run{0, 0, 0});
    for (const Run &r : runs) mine[r.owner] = r;
    for (int m = 0; m < nm; m++) {
      const Run &r = mine[m];
      g->memberReads[m] = r.b - r.a;
      GCHK_CTX(g, m, gtxi_scan_enqueue(g->ctx[m], reads + 3 * r.a, weights ? weights + r.a : nullptr, r.b - r.a, class_len, n_classes, win_step, win_size,
                                        preprocess, flags, class_offsets, &d[m], &extent));
    }
  } else {
    for (int m = 0; m < nm; m++) { g->partTri[m].clear(); g->partW[m].clear(); }
    for (int64_t i = 0; i < n; i++) {
      const int m = owner_of(g, reads[3 * i]);
      g->partTri[m].insert(g->partTri[m].end(), reads + 3 * i, reads + 3 * i + 3);
      if (weights) g->partW[m].push_back(weights[i]);
    }
    for (int m = 0; m < nm; m++) {
      const int64_t cnt = (int64_t)(g->partTri[m].size() / 3);
      g->memberReads[m] = cnt;
      GCHK_CTX(g, m, gtxi_scan_enqueue(g->ctx[m], g->partTri[m].data(), weights ? g->partW[m].data() : nullptr, cnt, class_len, n_classes, win_step, win_size,
                                        preprocess, flags, class_offsets, &d[m], &extent));
    }
  }
  if (extent > 0 && !windows_out) return gfail(g, GTX_E_ARG, "gtx_group_scan: null output");
  int rc = reduce_to_root(g, d, extent); if (rc) return rc;
  GCHK_HIP(g, hipSetDevice(g->dev[0]));
  if (extent > 0) GCHK_HIP(g, hipMemcpyAsync(windows_out, d[0], sizeof(uint64_t) * extent, hipMemcpyDeviceToHost, gtxi_stream(g->ctx[0])));
  for (int i = 0; i < nm; i++) GCHK_CTX(g, i, gtx_sync(g->ctx[i]));
  return GTX_OK;
}

}  // extern "C"
