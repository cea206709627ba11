// genomic_intervals.cpp -- see genomic_intervals.h.  Host side only: ingest, option semantics,
// error messages; every reduction is a call into libgtx.so (include/gtx.h).
#include "genomic_intervals.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <functional>
#include <shared_mutex>
#include <sys/mman.h>
#include <stdint.h>
#include <condition_variable>
#include <mutex>
#include <atomic>
#include <future>
#include <thread>
#include <iostream>

#include "gtx.h"
#include "gtx_bed.h"

using gtxhost::BedPacker;
using gtxhost::ChromTable;
using gtxhost::LineSource;
using gtxhost::PackedBatch;
using gtxhost::PackError;
using gtxhost::PackOptions;

bool _MESSAGES_ = false;

// GTX_TIMING=1: wall-clock marks on stderr (where the end-to-end time of a CLI run goes)
#include <chrono>
void GtxMark(const char *what);
static void Mark(const char *what) { GtxMark(what); }

// End of a tool's main(): everything is written; what is left at a normal return is freeing a million region
// objects and the HIP runtime's own shutdown (~0.3 s of a 0.8 s run).  The process is about to disappear anyway:
// flush and leave.  A normal return is kept when a profiler is attached (it reports from exit handlers) or when
// GTX_FULL_EXIT is set.
void GtxFinish(int code)
{
  if (getenv("GTX_TIMING")) fprintf(stderr, "[gtx leaving at epoch ms %lld]\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count());
  fflush(stdout); fflush(stderr);
  const char *pre = getenv("LD_PRELOAD");
  if (getenv("GTX_FULL_EXIT") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("HSA_TOOLS_LIB") ||
      (pre && strstr(pre, "rocprof")))
    return;
  _exit(code);
}
void GtxMark(const char *what)
{
  static const bool on = getenv("GTX_TIMING") != NULL;
  static const auto t0 = std::chrono::steady_clock::now();
  static const bool first = on && fprintf(stderr, "[gtx first mark at epoch ms %lld]\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count()) > 0;
  (void)first;
  if (on) fprintf(stderr, "[gtx %8.3f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what);
}

// ---- memory of region objects built in parallel (see genomic_intervals.h) ----
namespace {
struct Block { char *cur = NULL, *end = NULL; std::vector<std::pair<void *, size_t> > owned; };
thread_local Block *tls_block = NULL;
std::shared_timed_mutex g_blocks_mu;
std::vector<std::pair<uintptr_t, uintptr_t> > g_blocks;            // [first, last) of every live block, sorted

void BlockGrow(Block *b, size_t at_least)
{
  const size_t want = std::max(at_least, (size_t)4 << 20), bytes = (want + 4095) & ~(size_t)4095;
  void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);   // (pre-faulted: one call instead of a fault per page)
  if (p == MAP_FAILED) { fprintf(stderr, "Error: out of memory!\n"); exit(1); }
  b->cur = (char *)p; b->end = b->cur + bytes; b->owned.push_back(std::make_pair(p, bytes));
  std::unique_lock<std::shared_timed_mutex> lk(g_blocks_mu);
  g_blocks.insert(std::upper_bound(g_blocks.begin(), g_blocks.end(), std::make_pair((uintptr_t)p, (uintptr_t)0)), std::make_pair((uintptr_t)p, (uintptr_t)p + bytes));
}

bool InBlocks(const void *p)
{
  std::shared_lock<std::shared_timed_mutex> lk(g_blocks_mu);
  if (g_blocks.empty()) return false;
  auto it = std::upper_bound(g_blocks.begin(), g_blocks.end(), std::make_pair((uintptr_t)p, ~(uintptr_t)0));
  if (it == g_blocks.begin()) return false;
  --it;
  return (uintptr_t)p >= it->first && (uintptr_t)p < it->second;
}

void BlocksRelease(std::vector<std::pair<void *, size_t> > &owned)
{
  if (owned.empty()) return;
  std::unique_lock<std::shared_timed_mutex> lk(g_blocks_mu);
  for (auto &o : owned) {
    auto it = std::lower_bound(g_blocks.begin(), g_blocks.end(), std::make_pair((uintptr_t)o.first, (uintptr_t)0));
    if (it != g_blocks.end() && it->first == (uintptr_t)o.first) g_blocks.erase(it);
    munmap(o.first, o.second);
  }
  owned.clear();
}
}  // namespace

void *GtxRegionAlloc(size_t bytes)
{
  Block *b = tls_block;
  if (!b) return ::operator new(bytes);
  bytes = (bytes + 15) & ~(size_t)15;
  if ((size_t)(b->end - b->cur) < bytes) BlockGrow(b, bytes);
  void *p = b->cur; b->cur += bytes;
  return p;
}

void GtxRegionFree(void *p) { if (p && !InBlocks(p)) ::operator delete(p); }

static char *CopyString(const char *s) { size_t n = strlen(s) + 1; char *p = (char *)GtxRegionAlloc(n); memcpy(p, s, n); return p; }

// The threads that build an in-memory set (GenomicRegionSet::Init) must not exit() on a malformed line -- another thread may hold
// an earlier one: they note the error here and unwind; Init raises the one with the smallest line number, as the reference's
// line-by-line reader would have.
struct LoadError { bool set = false; long int line = 0; std::string msg; bool with_prefix = true; };
struct LoadAbort {};
static thread_local LoadError *tls_load_error = NULL;

static void DieLine(long int n_line, const std::string &msg)
{
  if (tls_load_error) { tls_load_error->set = true; tls_load_error->line = n_line; tls_load_error->msg = msg; tls_load_error->with_prefix = true; throw LoadAbort(); }
  fflush(stdout);
  fprintf(stderr, "\n");
  fprintf(stderr, "Error: Line %ld: %s\n", n_line, msg.c_str());
  exit(1);
}

static void DiePack(const PackError &e)
{
  if (e.no_prefix) { fflush(stdout); fprintf(stderr, "%s\n", e.msg.c_str()); exit(1); }
  DieLine(e.line, e.msg);
}

// ---------------------------------------------------------------------------------------------------
// GenomicInterval / GenomicRegion
// ---------------------------------------------------------------------------------------------------
GenomicInterval::GenomicInterval(const char *chromosome, char strand, long int start, long int stop, long int n_line)
{
  CHROMOSOME = CopyString(chromosome); STRAND = strand; START = start; STOP = stop; this->n_line = n_line;
}

GenomicInterval::~GenomicInterval() { GtxRegionFree(CHROMOSOME); }

void GenomicInterval::PrintInterval() { printf("%s %c %ld %ld", CHROMOSOME, STRAND, START, STOP); }

int GenomicInterval::CalcDirection(GenomicInterval *i, bool sorted_by_strand)
{
  const int by_chrom = strcmp(CHROMOSOME, i->CHROMOSOME);
  if (by_chrom != 0) return by_chrom;
  if (sorted_by_strand && STRAND != i->STRAND) return STRAND - i->STRAND;
  if (i->STOP < START) return 1;
  if (STOP < i->START) return -1;
  return 0;
}
void GenomicInterval::PrintInterval(FILE *f) { fprintf(f, "%s %c %ld %ld", CHROMOSOME, STRAND, START, STOP); }

bool GenomicInterval::OverlapsWith(GenomicInterval *i, bool ignore_strand)
{
  if (strcmp(CHROMOSOME, i->CHROMOSOME) != 0) return false;
  if (!ignore_strand && STRAND != i->STRAND) return false;
  return !(START > i->STOP || STOP < i->START);
}

long int GenomicInterval::CalcOverlap(GenomicInterval *i, bool ignore_strand)
{
  if (strcmp(CHROMOSOME, i->CHROMOSOME) != 0) return 0;
  if (!ignore_strand && STRAND != i->STRAND) return 0;
  const long int y = std::min(STOP, i->STOP) - std::max(START, i->START) + 1;
  return y > 0 ? y : 0;
}

GenomicRegion::GenomicRegion() : n_line(0), LABEL(NULL) {}

GenomicRegion::~GenomicRegion()
{
  GtxRegionFree(LABEL);
  for (size_t k = 0; k < I.size(); k++) delete I[k];
}

void GenomicRegion::PrintError(std::string error_msg) { DieLine(n_line, error_msg); }

size_t GenomicRegion::GetSize(bool skip_gaps)
{
  if (!skip_gaps) return (size_t)(I.back()->STOP - I.front()->START + 1);
  size_t size = 0;
  for (size_t k = 0; k < I.size(); k++) size += I[k]->GetSize();
  return size;
}

long int GenomicRegion::GetLabelValue(long int max_label_value)
{
  if (max_label_value <= 1) return 1;
  return std::min(max_label_value, atol(LABEL));
}

bool GenomicRegion::IsBefore(GenomicRegion *r, bool sorted_by_strand)
{
  GenomicInterval *a = I.front(), *b = r->I.front();
  int d = strcmp(a->CHROMOSOME, b->CHROMOSOME);
  if (d != 0) return d < 0;
  if (sorted_by_strand && a->STRAND != b->STRAND) return a->STRAND < b->STRAND;
  return a->START < b->START;
}

bool GenomicRegion::IsCompatibleSortedAndNonoverlapping()
{
  for (size_t k = 1; k < I.size(); k++) {
    if (strcmp(I[0]->CHROMOSOME, I[k]->CHROMOSOME) != 0 || I[0]->STRAND != I[k]->STRAND) return false;
    if (I[k]->START < I[k - 1]->START) return false;
  }
  for (size_t k = 1; k < I.size(); k++) if (I[k]->START <= I[k - 1]->STOP) return false;
  return true;
}

bool GenomicRegion::OverlapsWith(GenomicRegion *r, bool ignore_strand)
{
  for (size_t a = 0; a < I.size(); a++)
    for (size_t b = 0; b < r->I.size(); b++) if (I[a]->OverlapsWith(r->I[b], ignore_strand)) return true;
  return false;
}

long int GenomicRegion::CalcOverlap(GenomicRegion *r, bool ignore_strand)
{
  long int y = 0;
  for (size_t a = 0; a < I.size(); a++)
    for (size_t b = 0; b < r->I.size(); b++) y += I[a]->CalcOverlap(r->I[b], ignore_strand);
  return y;
}

int GenomicRegion::CalcDirection(GenomicRegion *r, bool sorted_by_strand)
{
  const int by_chrom = strcmp(I.front()->CHROMOSOME, r->I.front()->CHROMOSOME);
  if (by_chrom != 0) return by_chrom;
  if (sorted_by_strand && I.front()->STRAND != r->I.front()->STRAND) return I.front()->STRAND - r->I.front()->STRAND;
  if (r->I.back()->STOP < I.front()->START) return 1;
  if (I.back()->STOP < r->I.front()->START) return -1;
  return 0;
}

GenomicRegionBED::GenomicRegionBED(char *inp, long int n_line)
{
  this->n_line = n_line;
  gtxhost::BedFields f; char *bad = NULL;
  gtxhost::BedStatus st = gtxhost::ParseBedLine(inp, &f, &bad);
  if (st == gtxhost::BED_TOO_FEW_TOKENS) PrintError("number of tokens should be at least 3 for BED format!");
  if (st == gtxhost::BED_BAD_STRAND) {
    if (tls_load_error) { tls_load_error->set = true; tls_load_error->line = n_line; tls_load_error->msg = std::string("Error: invalid strand '") + bad + "'!"; tls_load_error->with_prefix = false; throw LoadAbort(); }
    fflush(stdout); std::cerr << "Error: invalid strand '" << bad << "'!\n"; exit(1);
  }
  n_tokens = f.n_tokens;
  LABEL = CopyString(f.label ? f.label : "_");
  if (n_tokens != 12) { I.push_back(new GenomicInterval(f.chrom, f.strand, f.start, f.stop, n_line)); return; }
  std::vector<long> iv; gtxhost::BedBlocks(f, &iv);                                     // :2174-2181
  for (size_t k = 0; k + 1 < iv.size(); k += 2) I.push_back(new GenomicInterval(f.chrom, f.strand, iv[k], iv[k + 1], n_line));
  if (I.empty()) PrintError("BED12 line without blocks!");
}

// ---------------------------------------------------------------------------------------------------
// GenomicRegionSet
// ---------------------------------------------------------------------------------------------------
void StdoutIsOurs();   // (below, with the GPU start-up)

GenomicRegionSet::GenomicRegionSet(char *file, unsigned long int buffer_size, bool verbose, bool load_in_memory, bool hide_header)
{
  this->file = file == NULL ? NULL : CopyString(file); this->file_ptr = NULL;
  this->buffer_size = buffer_size;
  this->verbose = verbose;
  this->from_stdin = file == NULL;
  this->load_in_memory = from_stdin ? false : load_in_memory;
  this->hide_header = hide_header;
  this->src = NULL; this->packed = NULL; this->R = NULL; this->n_regions = 0; this->r_index = 0;
  Init();
}

GenomicRegionSet::GenomicRegionSet(FILE *file_ptr, unsigned long int buffer_size, bool verbose, bool load_in_memory, bool hide_header)
{
  this->file = NULL; this->file_ptr = file_ptr;
  this->buffer_size = buffer_size;
  this->verbose = verbose;
  this->from_stdin = file_ptr == stdin;
  this->load_in_memory = from_stdin ? false : load_in_memory;
  this->hide_header = hide_header;
  this->src = NULL; this->packed = NULL; this->R = NULL; this->n_regions = 0; this->r_index = 0;
  Init();
}

GenomicRegionSet::~GenomicRegionSet()
{
  GtxRegionFree(file);
  delete src;
  delete packed;
  if (R) {
    long int n = load_in_memory ? n_regions : 1;
    for (long int k = 0; k < n; k++) delete R[k];
    delete[] R;
  }
  BlocksRelease(blocks_);
  if (load_in_memory) Mark("GenomicRegionSet (in memory): released");
}

void GenomicRegionSet::PrintError(std::string error_msg)
{
  fflush(stdout);
  fprintf(stderr, "\n");
  fprintf(stderr, "Error: %s\n", error_msg.c_str());
  exit(1);
}

// genomic_intervals.cpp:3736-3759: by the number of TAB-separated tokens of the first data line
void GenomicRegionSet::DetectFormat(const char *line)
{
  if (line[0] == '>' || line[0] == '@' || (line[0] == '#' && line[1] == '#')) PrintError("unsupported input format!\n");
  int nt = gtxhost::CountTokensLike(line, '\t');
  bool bed = nt == 1 || (nt >= 3 && nt <= 6);
  if (!bed && nt >= 6) {
    std::string copy(line);
    const char *p = copy.c_str(); int tabs = 0;
    while (*p && tabs < 5) { if (*p == '\t') tabs++; p++; }
    std::string tok(p, strcspn(p, "\t"));
    bed = tok.find('+') != std::string::npos || tok.find('-') != std::string::npos;
  }
  if (!bed) PrintError("unsupported input format!\n");     // REG / SAM / GFF are outside the path
  format = "BED";
}

void GtxWarmUp();

void GenomicRegionSet::Init()
{
  if (getenv("GTX_NO_WARMUP") == NULL) GtxWarmUp();
  Mark(load_in_memory ? "GenomicRegionSet (in memory): open" : "GenomicRegionSet (stream): open");
  std::string err;
  if (file_ptr == NULL && gtxhost::GtxView::IsGtx(file)) {
    // a packed region file: no text to tokenise; regions are made from its records on demand
    packed = gtxhost::GtxView::Open(file, &err);
    if (!packed) { fprintf(stderr, "%s\n", err.c_str()); exit(1); }
    format = packed->n ? "GTX" : "EMPTY";
    r_index = 0;
    if (load_in_memory) {
      n_regions = (long int)packed->n;
      R = n_regions > 0 ? new GenomicRegion *[n_regions] : NULL;
      for (long int k = 0; k < n_regions; k++) R[k] = PackedRegion(k);
    } else if (packed->n) {
      n_regions = 1;
      R = new GenomicRegion *[1];
      R[0] = PackedRegion(0);
    }
    return;
  }
  src = file_ptr ? LineSource::FromFile(file_ptr) : LineSource::Open(file, &err);
  if (!src) { fprintf(stderr, "%s\n", err.c_str()); exit(1); }
  char *line = src->Next();
  while (line && (strncmp(line, "browser ", 8) == 0 || strncmp(line, "track ", 6) == 0)) {
    if (!hide_header) { StdoutIsOurs(); printf("%s\n", line); }
    line = src->Next();
  }
  if (!line) { format = "EMPTY"; n_regions = 0; }
  else DetectFormat(line);

  if (load_in_memory) {
    // The first line here, the rest in blocks of complete lines, each block cut at line ends into one piece per thread: the
    // region objects (five small allocations each) are what the load costs, and the allocator scales with the threads
    // (1 M regions: 0.23 s on one thread).  The set is the same objects in the same order; a malformed line is reported as the
    // line-by-line reader would have met it -- the first one in the file.
    std::vector<GenomicRegion *> regs;
    if (line) regs.push_back(new GenomicRegionBED(line, src->line_no()));
    // Four threads, not all of them: this runs while the start-up thread brings up the HIP runtime, and a process that had many
    // threads running then takes 0.15-0.2 s longer to go away at exit (measured on the MI355X boxes, 16 threads against 4: every
    // run against none; where the time goes inside the driver's teardown was not found).  Four are enough: the runtime is what the
    // set's caller waits for anyway.
    const int T = getenv("GTX_LOAD_THREADS") && atoi(getenv("GTX_LOAD_THREADS")) > 0 ? atoi(getenv("GTX_LOAD_THREADS")) : std::min(gtxhost::WorkerThreads(), 4);
    std::vector<char> block; char *view = NULL; long first_line = 0;
    size_t got;
    while (line && (got = src->NextBlockView(block, &view, (size_t)64 << 20, &first_line)) > 0) {
      struct Piece { char *b, *e; long lines = 0, first = 0; std::vector<GenomicRegion *> out; LoadError err; Block mem; };
      std::vector<Piece> pc((size_t)std::max(1, std::min<int>(T, (int)(got / (256u << 10)) + 1)));
      char *end = view + got;
      for (size_t t = 0; t < pc.size(); t++) {
        char *b = t == 0 ? view : pc[t - 1].e, *e = t + 1 == pc.size() ? end : view + got * (t + 1) / pc.size();
        if (e < b) e = b;
        while (e < end && e > view && e[-1] != '\n') e++;                   // up to the end of the line it falls into
        pc[t].b = b; pc[t].e = e;
      }
      auto count = [&](size_t t) { pc[t].lines = gtxhost::CountNewlines(pc[t].b, pc[t].e); };
      auto build = [&](size_t t) {
        Piece &p = pc[t];
        p.out.reserve((size_t)p.lines);
        const auto t0 = std::chrono::steady_clock::now();
        BlockGrow(&p.mem, (size_t)p.lines * 176 + (size_t)(p.e - p.b) / 8 + 4096);   // a region (64 B), an interval (48), the vector's slot (16), two strings (more: another block)
        const auto t1 = std::chrono::steady_clock::now();
        tls_block = &p.mem;
        tls_load_error = &p.err;
        long no = p.first;
        try {
          for (char *q = p.b; q < p.e; no++) {
            char *nl = (char *)memchr(q, '\n', (size_t)(p.e - q));
            if (!nl) break;                                                // (cannot happen: pieces end at line ends)
            *nl = 0;
            p.out.push_back(new GenomicRegionBED(q, no));
            q = nl + 1;
          }
        } catch (const LoadAbort &) {}
        tls_load_error = NULL; tls_block = NULL;
        if (getenv("GTX_PACK_TRACE")) fprintf(stderr, "[load] piece %zu: %ld lines, block %.1f ms, objects %.1f ms\n", t, p.lines, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
      };
      auto run = [&](const std::function<void(size_t)> &f) { gtxhost::ParallelFor((int)pc.size(), [&](int t) { f((size_t)t); }); };
      const auto tA = std::chrono::steady_clock::now();
      run(count);
      long total = 0;
      for (auto &p : pc) { p.first = first_line + total; total += p.lines; }
      src->AdvanceLines(total);
      const auto tB = std::chrono::steady_clock::now();
      run(build);
      const auto tC = std::chrono::steady_clock::now();
      if (getenv("GTX_PACK_TRACE")) fprintf(stderr, "[load] %zu pieces: count %.1f ms, build %.1f ms\n", pc.size(), std::chrono::duration<double, std::milli>(tB - tA).count(), std::chrono::duration<double, std::milli>(tC - tB).count());
      for (auto &p : pc) {
        blocks_.insert(blocks_.end(), p.mem.owned.begin(), p.mem.owned.end());
        regs.insert(regs.end(), p.out.begin(), p.out.end());
        if (p.err.set) {                                                    // the pieces are in file order: this is the first bad line
          if (p.err.with_prefix) DieLine(p.err.line, p.err.msg);
          fflush(stdout); fprintf(stderr, "%s\n", p.err.msg.c_str()); exit(1);
        }
      }
    }
    n_regions = (long int)regs.size();
    R = n_regions > 0 ? new GenomicRegion *[n_regions] : NULL;
    for (long int k = 0; k < n_regions; k++) R[k] = regs[k];
  } else if (line) {
    n_regions = 1;
    cur_raw = line;
    R = new GenomicRegion *[1];
    R[0] = new GenomicRegionBED(line, src->line_no());
  }
  if (verbose) {
    std::cerr << "Reading from '" << (file == NULL ? "<standard input>" : file) << "'; ";
    std::cerr << "read-from-stdin = " << (from_stdin ? "true" : "false") << "; ";
    std::cerr << "load-in-memory = " << (load_in_memory ? "true" : "false") << "; ";
    std::cerr << "number of regions = " << n_regions << "; ";
    std::cerr << "format = " << format << "\n";
  }
  r_index = 0;
}

void GenomicRegionSet::Reset()
{
  if (from_stdin) PrintError("stdin cannot be reset!\n");
  if (file_ptr && !load_in_memory) { if (fseek(file_ptr, 0, SEEK_SET) != 0) PrintError("the stream cannot be reset!\n"); }
  if (load_in_memory) { r_index = 0; return; }
  if (R) { delete R[0]; delete[] R; R = NULL; }
  delete src; src = NULL;
  delete packed; packed = NULL;
  Init();
}

GenomicRegion *GenomicRegionSet::Get()
{
  if (n_regions == 0) return NULL;
  return !load_in_memory ? R[0] : (r_index >= n_regions ? NULL : R[r_index]);
}

GenomicRegion *GenomicRegionSet::PackedRegion(long int k)
{
  const gtxhost::GtxView &g = *packed;
  char line[512];
  const char strand = (g.minus[k >> 3] >> (k & 7)) & 1 ? '-' : '+';
  if (g.label) snprintf(line, sizeof line, "%s\t%ld\t%ld\t%ld\t0\t%c", g.chrom[g.chrom_idx[k]].c_str(), (long)g.start[k] - 1, (long)g.stop[k], (long)g.label[k], strand);
  else snprintf(line, sizeof line, "%s\t%ld\t%ld\t_\t0\t%c", g.chrom[g.chrom_idx[k]].c_str(), (long)g.start[k] - 1, (long)g.stop[k], strand);
  return new GenomicRegionBED(line, k + 1);
}

GenomicRegion *GenomicRegionSet::Next(bool retain_current)
{
  if (load_in_memory) { ++r_index; return r_index >= n_regions ? NULL : R[r_index]; }
  if (packed) {
    if (n_regions == 0) return NULL;
    if (r_index + 1 >= (long int)packed->n) return NULL;
    if (R[0] && !retain_current) delete R[0];
    R[0] = PackedRegion(++r_index);
    return R[0];
  }
  if (n_regions == 0 || !src) return NULL;
  char *line = src->Next();
  if (!line) return NULL;
  if (R[0] && !retain_current) delete R[0];
  cur_raw = line;
  R[0] = new GenomicRegionBED(line, src->line_no());
  return R[0];
}

GenomicRegion *GenomicRegionSet::Next(bool sorted_by_strand, bool retain_current)
{
  GenomicRegion *r0 = Get();
  if (r0 == NULL) return NULL;
  GenomicRegion *r = Next(true);
  if (r != NULL && r->IsBefore(r0, sorted_by_strand))
    r->PrintError(std::string("input regions are not sorted (sorted-by-strand = ") + (sorted_by_strand ? "true" : "false") + ")!");
  if (!load_in_memory) {
    if (!retain_current) delete r0;
    if (r == NULL) { R[0] = NULL; n_regions = 0; }            // end of the stream: Get() answers NULL from now on
  }
  return r;
}

const gtxhost::GtxView *GenomicRegionSet::DetachPacked(long int *current_record)
{
  if (load_in_memory || !packed) PrintError("[DetachPacked] not a streamed packed region file!");
  *current_record = n_regions > 0 ? r_index : (long int)packed->n;
  if (n_regions > 0) { delete R[0]; R[0] = NULL; }
  n_regions = 0;
  return packed;
}

long int GenomicRegionSet::StreamBytesLeft() { return (load_in_memory || !src) ? -1 : src->regular_file_bytes(); }

LineSource *GenomicRegionSet::DetachStream(std::string *current_line, long int *current_line_no)
{
  if (load_in_memory) PrintError("[DetachStream] the set is loaded in memory!");
  *current_line = n_regions > 0 ? cur_raw : std::string();
  *current_line_no = n_regions > 0 ? R[0]->n_line : 0;
  if (n_regions > 0) { delete R[0]; R[0] = NULL; }
  n_regions = 0;
  return src;
}

// ---------------------------------------------------------------------------------------------------
// GPU context shared by the classes of this file
// ---------------------------------------------------------------------------------------------------
// HIP start-up takes ~0.2 s: it runs on its own thread from the first region set on, next to the
// parsing of the reference file (GtxWarmUp), and is joined when the context is first needed.
// The classes of this file drive a gtx_group: one context per GPU (--ngpu N / GTX_NGPU, default 1; devices GTX_DEVICE,
// GTX_DEVICE+1, ...), classes dealt to the GPUs, RCCL reduce of the result vector (include/gtx.h).  A group of one is a
// plain context.
static std::future<gtx_group *> g_group_future;
static std::string g_ctx_error;
static int g_ngpu = 0;                                   // 0 = not set: GTX_NGPU or 1

void GtxSetDevices(int n) { g_ngpu = n; }

// Page-locked batch buffers for the bulk packer (gtxhost::BatchArena): two of them, filled in turn by DrainSet -- the packer
// threads write the packed triples where the DMA engine reads them (gtx_host_alloc memory skips the library's staging copy, and
// the call returns with the copy in flight: the contract of include/gtx.h is "untouched until the NEXT call has returned", which
// two alternating buffers satisfy).  Made by the start-up thread behind the loading of the reference set.
static const size_t kBatchReads = getenv("GTX_HOST_BATCH_READS") && atol(getenv("GTX_HOST_BATCH_READS")) > 0 ? (size_t)atol(getenv("GTX_HOST_BATCH_READS")) : (8u << 20);   // reads per batch handed to the device
static const size_t kPoolBytes = (kBatchReads * 3 + (24u << 20)) * sizeof(int32_t);   // what BedPacker::NextBatch reserves for it
static struct { void *buf[2]; bool used[2]; gtx_ctx *owner; } g_pool = {{NULL, NULL}, {false, false}, NULL};
static std::future<void> g_pool_future;

static void *PoolTake(size_t bytes)
{
  if (bytes > kPoolBytes || bytes < kPoolBytes / 2) return NULL;               // the triples of a batch, nothing else
  for (int k = 0; k < 2; k++) if (g_pool.buf[k] && !g_pool.used[k]) { g_pool.used[k] = true; return g_pool.buf[k]; }
  return NULL;
}
static bool PoolGive(void *p)
{
  for (int k = 0; k < 2; k++) if (p && g_pool.buf[k] == p) { g_pool.used[k] = false; return true; }
  return false;
}

static gtx_group *CreateGroup()
{
  int n = g_ngpu;
  if (n <= 0) { const char *e = getenv("GTX_NGPU"); n = e ? atoi(e) : 1; }
  if (n < 1) n = 1;
  const char *d = getenv("GTX_DEVICE");
  const int first = d ? atoi(d) : 0;
  std::vector<int> ids(n);
  const char *rh = getenv("GTX_GROUP_REHEARSE");           // test mode of the library: all members on one device
  for (int i = 0; i < n; i++) ids[i] = first + ((rh && atoi(rh)) ? 0 : i);
  gtx_group *g = gtx_group_create(n, ids.data());
  Mark("HIP context(s) created (start-up thread)");
  if (!g) { g_ctx_error = gtx_group_last_error(NULL); return g; }   // thread-local in the library: copy it out on this thread
  if (getenv("GTX_NO_PINNED_BATCHES") == NULL) {
    // page-locking 2 x 190 MB takes ~70 ms: on a thread of its own, behind the packing of the index set and gtx_set_refs; DrainSet waits for it
    g_pool.owner = gtx_group_ctx(g, 0);
    g_pool_future = std::async(std::launch::async, [] {
      for (int k = 0; k < 2; k++) g_pool.buf[k] = gtx_host_alloc(g_pool.owner, kPoolBytes);    // (NULL: the heap serves)
      Mark("page-locked batch buffers ready (start-up thread)");
      gtxhost::BatchArena::take = PoolTake; gtxhost::BatchArena::give = PoolGive;
    });
  }
  return g;
}

// exit() on an input error may come while the start-up threads are still inside the HIP runtime: they are joined before
// the static destructors (the runtime's own among them) run
static void JoinStartUp()
{
  if (g_group_future.valid()) g_group_future.wait();
  if (g_pool_future.valid()) g_pool_future.wait();
}

// Descriptor 1 points at stderr while the library brings RCCL communicators up on the start-up thread (RCCL prints a banner):
// nothing may reach stdout before that is over.  Only groups of more than one GPU (or the single-member RCCL self-test) make
// communicators.
void StdoutIsOurs()
{
  static bool checked = false;
  if (checked) return;
  checked = true;
  int n = g_ngpu;
  if (n <= 0) { const char *e = getenv("GTX_NGPU"); n = e ? atoi(e) : 1; }
  const char *f = getenv("GTX_GROUP_FORCE_RCCL"), *x = getenv("GTX_GROUP_SELF_EXCHANGE");
  if ((n > 1 || (f && atoi(f)) || (x && atoi(x))) && g_group_future.valid()) g_group_future.wait();
}

void GtxWarmUp()
{
  if (!g_group_future.valid()) {
    atexit(JoinStartUp);
    // (GTX_SYNC_STARTUP=1, diagnostic: bring the device up on the calling thread instead of next to the parsing of the index set)
    g_group_future = std::async(getenv("GTX_SYNC_STARTUP") ? std::launch::deferred : std::launch::async, CreateGroup);
  }
}

static gtx_group *Devices()
{
  static gtx_group *grp = NULL;
  if (!grp) {
    GtxWarmUp();
    grp = g_group_future.get();
    if (!grp) { fflush(stdout); fprintf(stderr, "\nError: %s\n", g_ctx_error.c_str()); exit(1); }
  }
  return grp;
}

static void CheckGrp(gtx_group *g, int rc)
{
  if (rc != GTX_OK) { fflush(stdout); fprintf(stderr, "\nError: [gtx %d] %s\n", rc, gtx_group_last_error(g)); exit(1); }
}

// Packs the rest of a query/input set batch by batch and hands every batch to `sink`, in order; `prep` (wait for the HIP context
// that the start-up thread is making, gtx_set_refs, *_begin) runs first.  In-memory sets are walked region by region with the same
// rules.  Two batches in turn, in the two page-locked buffers: one is packed while the device still reads the other (the hand-over
// returns with the copy in flight).  Measured and dropped: packing AHEAD of the hand-over on a helper thread, started before the HIP
// runtime is up -- batches outside the page-locked buffers cost first-touch page faults and a staging copy in the library, more
// page-locked memory to release at exit, and 100 M reads from text came out 10 % slower (0.95 -> 1.06 s on one box).
static std::atomic<bool> g_drain_stop(false);                 // set by a sink that has seen enough (an error it will raise after DrainSet): no more batches

// What DrainSet needs to have a streamed BED file tokenised on the device (gtx_count_add_text, include/gtx.h) instead of parsing it
// here: add(text, bytes, lines, rules) -> ticket, and needs_host(ticket).  The device takes the plain case only; a block with
// anything else in it comes back and is packed here, with the reference's reading of it and the reference's errors.
struct TextSink {
  std::function<bool()> usable;                                   // (asked behind prep(): one GPU)
  std::function<int(const char *, size_t, int64_t, const gtx_text_rules &)> add;
  std::function<bool(int)> needs_host;
};

static bool TextOnDevice(GenomicRegionSet *set, const PackOptions &opt, const TextSink *ts)
{
  static const char *e = getenv("GTX_TEXT_ON_DEVICE");        // 0: never; 1: whenever the input qualifies (tests); default: files of 32 MB or more
  if (!ts || (e && atoi(e) == 0)) return false;
  if (set->load_in_memory || set->format != "BED") return false;
  if (opt.guard || opt.collect_zero_length) return false;                 // (explode_blocks: a 12-column line sends its block back to the packer, which does it)
  if (!g_pool.buf[0] || !g_pool.buf[1] || !ts->usable()) return false;
  // a regular uncompressed file: worth it from 32 MB on.  A stream (stdin / a pipe, a .gz file, a FILE* of the caller's: -1) has no
  // size to go by: it takes the device path, and pump_text hands a first block that turns out to be all there is to the host packer
  const long left = set->StreamBytesLeft();
  return left < 0 || left >= ((e && atoi(e) == 1) ? 1 : (32l << 20));
}

// on_error (may be empty): called with the packer's error instead of dying with it, after the batch in hand -- which, under
// PackOptions::keep_prefix_on_error, holds the regions of the lines in front of the offending one -- has gone to `sink`; nothing more is packed
template <class Prep, class Sink>
static void DrainSet(GenomicRegionSet *set, PackOptions opt, Prep prep, Sink sink, const TextSink *text_sink = NULL,
                     const std::function<void(const PackError &)> &on_error = std::function<void(const PackError &)>())
{
  g_drain_stop = false;
  const size_t batch_reads = kBatchReads;
  PackedBatch two[2]; PackError err;
  // (a scanner's batch without a single region for the windows still carries the label values of its lines)
  const bool scan_mode = opt.mode == gtxhost::PACK_SCAN_SORTED || opt.mode == gtxhost::PACK_SCAN_UNSORTED;
  auto worth = [&](const PackedBatch &b) { return !b.empty() || (scan_mode && b.label_sum != 0); };
  auto pump_text = [&](BedPacker &packer) {
    // blocks of complete lines straight into the two page-locked buffers, tokenised and counted on the device; what is not plain is
    // packed here.  Two blocks in flight: block i's verdict is collected before block i+2 is read over it.
    std::vector<const char *> names((size_t)opt.chroms->size());
    for (int i = 0; i < opt.chroms->size(); i++) names[i] = opt.chroms->name(i).c_str();
    gtx_text_rules rules;
    rules.chrom_names = names.empty() ? NULL : names.data(); rules.n_chrom = opt.chroms->size();
    rules.strand_aware = opt.strand_aware; rules.sorted_rules = opt.mode == gtxhost::PACK_OVERLAPS_SORTED || opt.mode == gtxhost::PACK_SCAN_SORTED; rules.sorted_by_strand = opt.sorted_by_strand;
    rules.max_label_value = opt.max_label_value;
    g_pool.used[0] = g_pool.used[1] = true;                      // (the buffers hold text now: a batch packed here takes heap memory)
    packer.UseTextBuffers((char *)g_pool.buf[0], (char *)g_pool.buf[1], kPoolBytes);
    PackedBatch batch;
    if (!packer.PackPrimedText(&batch, &err) && !on_error) DiePack(err);
    if (worth(batch)) sink(batch);
    if (err.set) { on_error(err); g_drain_stop = true; g_pool.used[0] = g_pool.used[1] = false; return; }
    BedPacker::TextBlock blk[2]; int ticket[2] = {-1, -1}; bool host_only = false, first_block = true;
    static const bool text_forced = getenv("GTX_TEXT_ON_DEVICE") && atoi(getenv("GTX_TEXT_ON_DEVICE")) == 1;
    long on_device = 0, redone = 0, host_blocks = 0;
    auto settle = [&](int k) {                                   // the verdict on the block in blk[k]
      if (ticket[k] < 0) return;
      const bool redo = text_sink->needs_host(ticket[k]);
      ticket[k] = -1;
      if (!redo) { on_device++; return; }
      redone++;
      batch.clear();
      const bool ok = packer.PackTextBlock(blk[k], &batch, &err);
      if (err.set && !on_error) DiePack(err);
      (void)ok;
      if (worth(batch)) sink(batch);
      if (err.set) { on_error(err); g_drain_stop = true; }
    };
    for (int cur = 0;; cur ^= 1) {
      settle(cur);                                               // (its buffer is about to be read over)
      if (g_drain_stop) break;
      if (!packer.NextTextBlock(&blk[cur])) break;
      BedPacker::TextBlock &b = blk[cur];
      if (!b.seam_ok) host_only = true;                          // a last line that could not be read: no seam key for the device
      if (first_block && !text_forced && b.bytes < (32u << 20) && packer.SourceAtEnd()) host_only = true;   // a short stream: the host packer is done before the device has the text
      first_block = false;
      if (host_only) {
        host_blocks++;
        batch.clear();
        packer.PackTextBlock(b, &batch, &err);
        if (err.set && !on_error) { settle(cur ^ 1); DiePack(err); }
        if (worth(batch)) sink(batch);
        if (err.set) { on_error(err); g_drain_stop = true; break; }
        continue;
      }
      rules.have_prev = b.have_prev; rules.prev_chrom = b.prev_chrom.c_str(); rules.prev_strand = b.prev_strand; rules.prev_start = b.prev_start;
      ticket[cur] = text_sink->add(b.text, b.bytes, b.n_lines, rules);
      if (on_error) settle(cur);                                 // (a caller that goes on after an error wants nothing behind the offending line counted: one block at a time)
    }
    settle(0); settle(1);
    if (getenv("GTX_TEXT_TRACE")) fprintf(stderr, "[gtx text] blocks tokenised on the device: %ld, sent back to the host packer: %ld, packed on the host from the start: %ld\n", on_device, redone, host_blocks);
    g_pool.used[0] = g_pool.used[1] = false;
  };
  auto pump = [&](BedPacker &packer, bool prepared = false) {
    if (!prepared) prep();
    if (g_pool_future.valid()) g_pool_future.get();             // the page-locked batch buffers are there
    for (int cur = 0;;) {
      PackedBatch &batch = two[cur];
      bool more = packer.NextBatch(&batch, batch_reads, &err);
      if (g_drain_stop) break;
      if (err.set && !on_error) DiePack(err);
      if (worth(batch)) {
        const auto t0 = std::chrono::steady_clock::now();
        sink(batch); cur ^= 1;
        if (getenv("GTX_PACK_TRACE")) fprintf(stderr, "[sink] %zu reads handed over in %.1f ms\n", batch.tri.size() / 3, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      if (err.set) { on_error(err); break; }
      if (!more) break;
    }
  };
  if (!set->load_in_memory && set->format == "GTX") {
    long int at = 0;
    const gtxhost::GtxView *view = set->DetachPacked(&at);
    BedPacker packer(view, opt);
    packer.SkipRecords((uint64_t)at);
    pump(packer);
    return;
  }
  if (!set->load_in_memory) {
    std::string first; long int first_no = 0;
    LineSource *src = NULL;
    bool on_device = false;
    if (text_sink) {                                             // the decision needs the page-locked buffers and the device: made behind prep()
      prep();
      if (g_pool_future.valid()) g_pool_future.get();
      on_device = TextOnDevice(set, opt, text_sink);
    }
    src = set->DetachStream(&first, &first_no);
    BedPacker packer(src, opt);
    if (first_no > 0) packer.Prime(first, first_no);
    if (on_device) pump_text(packer); else pump(packer, text_sink != NULL);
    return;
  }
  // in-memory set: re-emit its regions as lines through the same packer rules
  std::string text;
  for (GenomicRegion *r = set->Get(); r != NULL; r = set->Next()) {
    GenomicInterval *i = r->I.front();
    if (r->I.size() > 1) {                                  // a multi-interval region: its envelope under -gaps, its intervals one by one for coverage
      if (!opt.match_gaps && !opt.explode_blocks && !opt.collect_blocks) r->PrintError("multi-interval (BED12) regions are outside the MI355X counting path (except genomic_overlaps count, coverage and density)!");
      if (!r->IsCompatibleSortedAndNonoverlapping()) r->PrintError("query regions should be compatible, sorted and non-overlapping!");
    }
    char buf[64];
    if (opt.collect_blocks && r->I.size() > 1) {               // as a 12-column line: the packer lists its intervals
      text += i->CHROMOSOME; text += '\t';
      snprintf(buf, sizeof buf, "%ld\t%ld\t", i->START - 1, r->I.back()->STOP); text += buf;
      text += r->LABEL; text += "\t0\t"; text += i->STRAND; text += "\t0\t0\t0\t";
      snprintf(buf, sizeof buf, "%zu\t", r->I.size()); text += buf;
      for (GenomicIntervalSet::iterator b = r->I.begin(); b != r->I.end(); b++) { snprintf(buf, sizeof buf, "%ld,", (*b)->STOP - (*b)->START + 1); text += buf; }
      text += '\t';
      for (GenomicIntervalSet::iterator b = r->I.begin(); b != r->I.end(); b++) { snprintf(buf, sizeof buf, "%ld,", (*b)->START - i->START); text += buf; }
      text += '\n';
      continue;
    }
    if (opt.explode_blocks && r->I.size() > 1) {
      for (GenomicIntervalSet::iterator b = r->I.begin(); b != r->I.end(); b++) {
        text += i->CHROMOSOME; text += '\t';
        snprintf(buf, sizeof buf, "%ld\t%ld\t", (*b)->START - 1, (*b)->STOP); text += buf;
        text += r->LABEL; text += "\t0\t"; text += i->STRAND; text += '\n';
      }
      continue;
    }
    text += i->CHROMOSOME; text += '\t';
    snprintf(buf, sizeof buf, "%ld\t%ld\t", i->START - 1, r->I.back()->STOP); text += buf;
    text += r->LABEL; text += "\t0\t"; text += i->STRAND; text += '\n';
  }
  BedPacker packer((LineSource *)NULL, opt);
  packer.PrimeBlock(text, 1);
  pump(packer);
}

// quick order hint for the kernel choice (a wrong hint only costs speed): sample adjacent pairs
static bool LooksSorted(const gtxhost::RawVec &tri)
{
  const size_t n = tri.size() / 3;
  if (n < 2) return true;
  const size_t stride = n > 4096 ? n / 4096 : 1;
  int descents = 0;                                     // a batch is a few sorted runs at most (e.g. the '+' then the '-' reads)
  for (size_t i = 0; i + 1 < n; i += stride) {
    const int32_t *a = &tri[3 * i], *b = a + 3;
    if ((b[0] < a[0] || (b[0] == a[0] && b[1] < a[1])) && ++descents > 2) return false;
  }
  return true;
}

// the same hint from the text of a block: pairs of adjacent lines at ~4096 places, (chromosome token, column 2) compared
static bool TextLooksSorted(const char *text, size_t bytes)
{
  if (bytes < 64) return true;
  const size_t stride = bytes > (4096u * 64u) ? bytes / 4096 : 64;
  int descents = 0;
  auto key = [&](const char *l, const char *e, const char **tok, size_t *len, long *start) {
    const char *t = (const char *)memchr(l, '\t', (size_t)(e - l));
    if (!t) return false;
    *tok = l; *len = (size_t)(t - l); *start = atol(t + 1);
    return true;
  };
  for (size_t at = 0; at + 2 < bytes; at += stride) {
    const char *a = (const char *)memchr(text + at, '\n', bytes - at);
    if (!a || a + 1 >= text + bytes) break;
    a++;
    const char *ae = (const char *)memchr(a, '\n', (size_t)(text + bytes - a));
    if (!ae || ae + 1 >= text + bytes) break;
    const char *b = ae + 1, *be = (const char *)memchr(b, '\n', (size_t)(text + bytes - b));
    if (!be) break;
    const char *ta, *tb; size_t la, lb; long sa, sb;
    if (!key(a, ae, &ta, &la, &sa) || !key(b, be, &tb, &lb, &sb)) continue;
    const int d = memcmp(ta, tb, std::min(la, lb));
    const bool before = d ? d > 0 : (la != lb ? la > lb : sb < sa);
    if (before && ++descents > 2) return false;
  }
  return true;
}

// an error of a member context, reported through the group's channel (CheckGrp prints gtx_group_last_error)

static bool LooksSortedVec(const std::vector<int32_t> &tri)
{
  const size_t n = tri.size() / 3;
  if (n < 2) return true;
  const size_t stride = n > 4096 ? n / 4096 : 1;
  for (size_t i = 0; i + 1 < n; i += stride) {
    const int32_t *a = &tri[3 * i], *b = a + 3;
    if (b[0] < a[0] || (b[0] == a[0] && b[1] < a[1])) return false;
  }
  return true;                                            // (a hint: the device verifies it and falls back by itself)
}

// ---------------------------------------------------------------------------------------------------
// GenomicRegionSetOverlaps
// ---------------------------------------------------------------------------------------------------
GenomicRegionSetOverlaps::GenomicRegionSetOverlaps(GenomicRegionSet *QuerySet, GenomicRegionSet *IndexSet)
{
  this->QuerySet = QuerySet; this->IndexSet = IndexSet; current_qreg = NULL; current_ireg = NULL;
}

GenomicRegionSetOverlaps::~GenomicRegionSetOverlaps() {}

unsigned long int *GenomicRegionSetOverlaps::CountIndexOverlaps(bool match_gaps, bool ignore_strand, long int max_label_value)
{
  // single-interval regions have no gaps: both settings select the same pairs (genomic_intervals.cpp:5226)
  if (IndexSet->load_in_memory == false) {
    fprintf(stderr, "[GenomicRegionSetOverlaps::CountIndexOverlaps]: index set must be loaded in memory for this operation!\n");
    exit(1);
  }
  return Reduce(false, match_gaps, ignore_strand, max_label_value);
}

unsigned long int *GenomicRegionSetOverlaps::CalcIndexCoverage(bool match_gaps, bool ignore_strand, long int max_label_value)
{
  // single intervals: the envelope formula (:5278) and CalcOverlap (:1196-1202) coincide except that the former is not clamped
  // at 0 -- pairs with an inverted interval under the sorted merge (GTX_GAPS_FORMULA)
  if (IndexSet->load_in_memory == false) {
    fprintf(stderr, "[GenomicRegionSetOverlaps::CalcIndexCoverage]: index set must be loaded in memory for this operation!\n");
    exit(1);
  }
  return Reduce(true, match_gaps, ignore_strand, max_label_value);
}

unsigned long int *GenomicRegionSetOverlaps::Reduce(bool coverage, bool match_gaps, bool ignore_strand, long int max_label_value)
{
  const long int M = IndexSet->n_regions;
  const SortedGenomicRegionSetOverlaps *merge = dynamic_cast<const SortedGenomicRegionSetOverlaps *>(this);
  const bool sorted = merge != NULL, by_strand = merge && merge->sorted_by_strand;
  for (long int k = 0; k < M; k++) IndexSet->R[k]->n_line = k;                       // :5309
  Mark("CountIndexOverlaps: start");

  // ---- index side ----
  // Sorted merge: the reference notices an index set that is out of order only at the moment the merge pulls the offending
  // region (genomic_intervals.cpp:5868).  v = the first such region; the regions from v on are never matched (either the
  // queries end before the merge gets there, or the run ends with the error), the packer's IndexGuard decides which.
  long int v = M;
  if (sorted) for (long int k = 1; k < M; k++) if (IndexSet->R[k]->IsBefore(IndexSet->R[k - 1], by_strand)) { v = k; break; }
  gtxhost::IndexGuard guard;
  if (v < M) {
    guard.by_strand = by_strand;
    for (long int k = 0; k < v; k++) {
      GenomicInterval *i = IndexSet->R[k]->I.front();
      guard.chrom.push_back(i->CHROMOSOME); guard.strand.push_back(i->STRAND); guard.start.push_back(i->START); guard.stop.push_back(IndexSet->R[k]->I.back()->STOP);
    }
    char buf[160];
    snprintf(buf, sizeof buf, "\nError: Line %ld: index regions are not sorted (sorted-by-strand = %s)!", v, by_strand ? "true" : "false");
    guard.msg = buf;
  }
  ChromTable chroms;
  bool explode = false;                                                             // coverage without -gaps over multi-interval index regions
  bool blocks_mode = false;                                                         // count without -gaps over multi-interval index regions
  const char *last_name = NULL;                                                     // region files repeat a chromosome many times in a row
  for (long int k = 0; k < v; k++) {
    GenomicRegion *r = IndexSet->R[k];
    GenomicInterval *i = r->I.front();
    if (r->I.size() > 1) {
      // a multi-interval (BED12) region: with -gaps it is matched on its envelope (:5226, :5752, :5278) -- what the device computes.
      // Without: coverage is a sum over ALL interval pairs of the two regions (CalcOverlap :1196-1202, pairs that do not overlap
      // add 0), so the intervals go to the device one by one (`explode`) and a region's value is the sum of its intervals';
      // count needs "some interval pair overlaps" (:1167-1172), which no sum of independent pieces gives: the device counts on the
      // envelopes and settles the pairs with a multi-interval side one by one (`blocks_mode`, gtx_set_ref_blocks);
      if (!r->IsCompatibleSortedAndNonoverlapping()) r->PrintError("index regions should be compatible, sorted and non-overlapping!");   // :5607, :5853
      if (!match_gaps && coverage) explode = true;
      if (!match_gaps && !coverage) blocks_mode = true;
    }
    if (!sorted && (i->START > r->I.back()->STOP || r->I.back()->STOP <= 0)) continue;   // :5609, :5659
    if (last_name && strcmp(last_name, i->CHROMOSOME) == 0) continue;
    chroms.Add(i->CHROMOSOME); last_name = i->CHROMOSOME;
  }
  chroms.Freeze();
  const int n_chrom = chroms.size();
  const bool strand_aware = !ignore_strand;
  bool zero_length_refs = false;
  std::vector<int32_t> refs((size_t)3 * (M > 0 ? M : 1));
  last_name = NULL; int last_id = -1;
  for (long int k = 0; k < M; k++) {
    GenomicRegion *r = IndexSet->R[k];
    GenomicInterval *i = r->I.front();
    if (k >= v) { refs[3 * k] = -1; refs[3 * k + 1] = 1; refs[3 * k + 2] = 0; continue; }   // behind the out-of-order spot: a placeholder
    const long int STOP = r->I.back()->STOP;                                       // the envelope's end (single interval: its own)
    if (sorted && i->START == STOP + 1) zero_length_refs = true;
    if (i->START >= INT_MAX - 1 || STOP >= INT_MAX - 1 || i->START <= INT_MIN + 1 || STOP <= INT_MIN + 1)
      r->PrintError("coordinate does not fit the packed 32-bit representation of the MI355X path!");
    if (!last_name || strcmp(last_name, i->CHROMOSOME) != 0) { last_name = i->CHROMOSOME; last_id = chroms.Find(i->CHROMOSOME); }
    const int id = last_id;
    if (id < 0) { refs[3 * k] = -1; refs[3 * k + 1] = 1; refs[3 * k + 2] = 0; continue; }  // invalid region: never matches
    refs[3 * k] = id + ((strand_aware && i->STRAND == '-') ? n_chrom : 0);
    refs[3 * k + 1] = (int32_t)i->START; refs[3 * k + 2] = (int32_t)STOP;
  }
  // `explode`: one device region per interval; first_piece[k] .. first_piece[k+1] are region k's
  std::vector<long int> first_piece;
  long int MD = M;
  if (explode) {
    std::vector<int32_t> pieces;
    first_piece.assign((size_t)M + 1, 0);
    for (long int k = 0; k < M; k++) {
      first_piece[k] = (long int)(pieces.size() / 3);
      GenomicRegion *r = IndexSet->R[k];
      if (refs[3 * k] < 0 || r->I.size() == 1) { pieces.insert(pieces.end(), refs.begin() + 3 * k, refs.begin() + 3 * k + 3); continue; }
      for (GenomicIntervalSet::iterator b = r->I.begin(); b != r->I.end(); b++) {
        if ((*b)->START >= INT_MAX - 1 || (*b)->STOP >= INT_MAX - 1 || (*b)->START <= INT_MIN + 1 || (*b)->STOP <= INT_MIN + 1)
          r->PrintError("coordinate does not fit the packed 32-bit representation of the MI355X path!");
        pieces.push_back(refs[3 * k]); pieces.push_back((int32_t)(*b)->START); pieces.push_back((int32_t)(*b)->STOP);
      }
    }
    first_piece[M] = (long int)(pieces.size() / 3);
    MD = first_piece[M];
    refs.swap(pieces);
  }
  // `blocks_mode`: the intervals of every region, as gtx_set_ref_blocks takes them
  std::vector<int64_t> blk_first; std::vector<int32_t> blk_iv;
  if (blocks_mode) {
    blk_first.assign((size_t)M + 1, 0);
    for (long int k = 0; k < M; k++) {
      blk_first[k] = (int64_t)(blk_iv.size() / 2);
      GenomicRegion *r = IndexSet->R[k];
      if (refs[3 * k] < 0 || r->I.size() == 1) { blk_iv.push_back(refs[3 * k + 1]); blk_iv.push_back(refs[3 * k + 2]); continue; }
      long int prev_stop = LONG_MIN;
      for (GenomicIntervalSet::iterator b = r->I.begin(); b != r->I.end(); b++) {
        if ((*b)->START >= INT_MAX - 1 || (*b)->STOP >= INT_MAX - 1 || (*b)->START <= INT_MIN + 1 || (*b)->STOP <= INT_MIN + 1)
          r->PrintError("coordinate does not fit the packed 32-bit representation of the MI355X path!");
        if ((*b)->STOP < prev_stop) r->PrintError("multi-interval (BED12) region with an interval of negative size is outside the MI355X counting path!");
        prev_stop = (*b)->STOP;
        blk_iv.push_back((int32_t)(*b)->START); blk_iv.push_back((int32_t)(*b)->STOP);
      }
    }
    blk_first[M] = (int64_t)(blk_iv.size() / 2);
  }
  Mark("index packed");
  gtx_group *grp = NULL;
  const int n_classes = std::max(1, n_chrom * (strand_aware ? 2 : 1));
  auto device_side = [&](bool cover) {                              // (on DrainSet's hand-over thread, while the queries are already being packed)
    grp = Devices();
    Mark("device ready");
    CheckGrp(grp, gtx_group_set_refs(grp, refs.data(), MD, n_classes, sorted ? GTX_REFS_KEEP_ZERO_LENGTH : 0));
    if (blocks_mode) CheckGrp(grp, gtx_group_set_ref_blocks(grp, blk_first.data(), blk_iv.data()));
    Mark("gtx_set_refs done");
    CheckGrp(grp, cover ? gtx_group_coverage_begin(grp) : gtx_group_count_begin(grp));
  };

  // ---- query side: stream -> packed batches -> device ----
  PackOptions opt;
  opt.mode = sorted ? gtxhost::PACK_OVERLAPS_SORTED : gtxhost::PACK_OVERLAPS_UNSORTED;
  opt.chroms = &chroms; opt.strand_aware = strand_aware; opt.sorted_by_strand = by_strand;
  opt.max_label_value = max_label_value; opt.collect_zero_length = !coverage && sorted && zero_length_refs;
  opt.match_gaps = match_gaps;
  opt.explode_blocks = coverage && !match_gaps;                  // multi-interval queries: their intervals one by one (see the index side)
  opt.collect_blocks = !coverage && !match_gaps;                 // ... or on a list of their own, for the pair kernel
  if (v < M) opt.guard = &guard;
  std::vector<int32_t> zero_len;
  unsigned long int *hits = new unsigned long int[M > 0 ? M : 1];
  gtx_count_info info;
  // the query file's text tokenised on the device where that applies (one GPU, a plain BED file: TextOnDevice)
  TextSink text_sink;
  text_sink.usable = [&] { return gtx_group_size(grp) >= 1; };     // (several GPUs: the blocks go to the members in turn, gtx_group_count_add_text)
  text_sink.needs_host = [&](int ticket) { int redo = 0; CheckGrp(grp, gtx_group_text_result(grp, ticket, &redo)); return redo != 0; };
  if (coverage) {
    // zero-length reads (sorted rules let them through) and zero-length regions contribute 0: the device leaves them out
    const uint32_t cflags = sorted ? (GTX_ZERO_LENGTH_OK | (match_gaps ? GTX_GAPS_FORMULA : 0u)) : 0u;
    text_sink.add = [&](const char *text, size_t bytes, int64_t lines, const gtx_text_rules &rules) {
      int ticket = -1;
      CheckGrp(grp, gtx_group_coverage_add_text(grp, text, bytes, lines, &rules, cflags | (sorted ? GTX_READS_SORTED : 0u), &ticket));
      return ticket;
    };
    DrainSet(QuerySet, opt, [&] { device_side(true); }, [&](const PackedBatch &b) {
      CheckGrp(grp, gtx_group_coverage_add(grp, b.tri.data(), b.w.empty() ? NULL : b.w.data(), (int64_t)(b.tri.size() / 3), cflags));
    }, &text_sink);
    Mark("queries packed and enqueued");
    if (explode) {
      std::vector<uint64_t> part((size_t)std::max<long int>(MD, 1));
      CheckGrp(grp, gtx_group_coverage_end(grp, part.data(), &info));
      for (long int k = 0; k < M; k++) { unsigned long int sum = 0; for (long int j = first_piece[k]; j < first_piece[k + 1]; j++) sum += part[j]; hits[k] = sum; }
    } else
    CheckGrp(grp, gtx_group_coverage_end(grp, (uint64_t *)hits, &info));
    if (info.n_unplaced != 0) { fflush(stdout); fprintf(stderr, "\nError: %ld inverted query regions (start > stop) exceed what the MI355X path sets aside for pairwise matching!\n", (long)info.n_unplaced); exit(1); }
    Mark("coverage on the host");
    return hits;
  }
  const uint32_t mode_flags = sorted ? GTX_ZERO_LENGTH_OK : 0;
  text_sink.add = [&](const char *text, size_t bytes, int64_t lines, const gtx_text_rules &rules) {
    int ticket = -1;
    // (the sorted merge's input is in order, or the block comes back; the bin index takes any order: a look at the text decides the kernel)
    const uint32_t flags = mode_flags | ((sorted || TextLooksSorted(text, bytes)) ? GTX_READS_SORTED : 0);
    CheckGrp(grp, gtx_group_count_add_text(grp, text, bytes, lines, &rules, flags, &ticket));
    return ticket;
  };
  DrainSet(QuerySet, opt, [&] { device_side(false); }, [&](const PackedBatch &b) {
    uint32_t flags = mode_flags | (LooksSorted(b.tri) ? GTX_READS_SORTED : 0);
    CheckGrp(grp, gtx_group_count_add(grp, b.tri.data(), b.w.empty() ? NULL : b.w.data(), (int64_t)(b.tri.size() / 3), flags));
    if (!b.m_cnt.empty()) {                                       // the multi-interval queries of the batch
      std::vector<int64_t> first(b.m_cnt.size() + 1, 0);
      for (size_t i = 0; i < b.m_cnt.size(); i++) first[i + 1] = first[i] + b.m_cnt[i];
      CheckGrp(grp, gtx_group_count_add_regions(grp, b.m_tri.data(), b.m_w.data(), first.data(), b.m_blocks.data(), (int64_t)b.m_cnt.size()));
    }
    zero_len.insert(zero_len.end(), b.zero_len.begin(), b.zero_len.end());
  }, &text_sink);
  Mark("queries packed and enqueued");
  CheckGrp(grp, gtx_group_count_end(grp, (uint64_t *)hits, &info));
  if (getenv("GTX_TIMING") && gtx_group_size(grp) > 1) {
    std::vector<int64_t> mr((size_t)gtx_group_size(grp));
    gtx_group_member_reads(grp, mr.data());
    for (size_t i = 0; i < mr.size(); i++) fprintf(stderr, "[gtx] GPU %zu counted %ld reads\n", i, (long)mr[i]);
  }
  Mark("counts on the host");
  // (sorted merge: n_degenerate counts the inverted reads, which the library matched pair by pair)
  if (!sorted && info.n_degenerate != 0) { fflush(stdout); fprintf(stderr, "\nError: internal: the packer let %ld degenerate reads through\n", (long)info.n_degenerate); exit(1); }
  if (info.n_unplaced != 0) { fflush(stdout); fprintf(stderr, "\nError: %ld inverted query regions (start > stop) exceed what the MI355X path sets aside for pairwise matching!\n", (long)info.n_unplaced); exit(1); }

  // sorted merge only: a zero-length read never overlaps a zero-length region at the same spot
  // (rS <= qE and rE >= qS cannot both hold), while the rank difference counts it as -1: undo that.
  if (!zero_len.empty()) {
    for (long int k = 0; k < M; k++) {
      if (refs[3 * k + 1] != refs[3 * k + 2] + 1) continue;
      for (size_t z = 0; z + 2 < zero_len.size(); z += 3)
        if (zero_len[z] == refs[3 * k] && zero_len[z + 1] == refs[3 * k + 1]) hits[k] += (unsigned long int)(long int)zero_len[z + 2];
    }
  }
  return hits;
}

// ---- per-query iteration (host side, like the reference's: these calls hand out GenomicRegion pointers) -------------------
// What GetMatch/NextMatch deliver are candidates on the envelopes; the filter of genomic_intervals.cpp:5224-5248 decides which of them
// are overlaps: under match_gaps the envelope is all that counts, otherwise some interval pair must overlap; and unless strands are
// ignored both regions must lie on the same strand (that of their first interval).
namespace {
struct OverlapFilter {
  GenomicRegion *query; bool gaps, any_strand;
  bool operator()(GenomicRegion *cand) const
  {
    if (!any_strand && query->I.front()->STRAND != cand->I.front()->STRAND) return false;
    return gaps || query->OverlapsWith(cand, any_strand);
  }
};

// value(r) summed over the overlaps of the current query, in `unsigned long` like the reference's accumulators (:5254-5263, :5291-5296)
template <class Value>
unsigned long int sum_over_overlaps(GenomicRegionSetOverlaps *o, bool match_gaps, bool ignore_strand, Value value)
{
  unsigned long int total = 0;
  GenomicRegion *r = o->GetOverlap(match_gaps, ignore_strand);
  while (r != NULL) { total += (unsigned long int)value(r); r = o->NextOverlap(match_gaps, ignore_strand); }
  return total;
}
}  // namespace

GenomicRegion *GenomicRegionSetOverlaps::GetOverlap(bool match_gaps, bool ignore_strand)
{
  const OverlapFilter overlaps = {current_qreg, match_gaps, ignore_strand};
  GenomicRegion *cand = GetMatch();
  while (cand != NULL && !overlaps(cand)) cand = NextMatch();
  return cand;
}

GenomicRegion *GenomicRegionSetOverlaps::NextOverlap(bool match_gaps, bool ignore_strand)
{
  const OverlapFilter overlaps = {current_qreg, match_gaps, ignore_strand};
  GenomicRegion *cand = NextMatch();
  while (cand != NULL && !overlaps(cand)) cand = NextMatch();
  return cand;
}

unsigned long int GenomicRegionSetOverlaps::CalcQueryCoverage(bool match_gaps, bool ignore_strand, long int max_label_value)
{
  GenomicRegion *q = current_qreg;
  return sum_over_overlaps(this, match_gaps, ignore_strand, [=](GenomicRegion *r) -> long int {
    // envelope against envelope under match_gaps (not clamped: :5258), interval pairs otherwise; times the INDEX region's label value
    const long int len = match_gaps ? std::min(r->I.back()->STOP, q->I.back()->STOP) - std::max(r->I.front()->START, q->I.front()->START) + 1
                                    : q->CalcOverlap(r, ignore_strand);
    return len * r->GetLabelValue(max_label_value);
  });
}

unsigned long int GenomicRegionSetOverlaps::CountQueryOverlaps(bool match_gaps, bool ignore_strand, long int max_label_value)
{
  return sum_over_overlaps(this, match_gaps, ignore_strand, [=](GenomicRegion *r) -> long int { return r->GetLabelValue(max_label_value); });
}

// The bin index of UnsortedGenomicRegionSetOverlaps (genomic_intervals.cpp:5593-5675) for GetMatch/NextMatch: a region lives at the
// lowest level where its (clamped) start and its stop fall into one bin; a bin keeps its regions in the order they came.
struct UnsortedGenomicRegionSetOverlaps::MatchIndex {
  std::vector<int> bits;
  struct Chrom { std::vector<long int> n_bins; std::vector<std::vector<std::vector<long int> > > bins; };   // [level][bin] -> region ordinals
  std::map<std::string, Chrom> chrom;
  // cursor of the walk for the current query (:5729-5764)
  Chrom *cur; long int start, stop, b, b_stop; int l; long int at;          // `at` counts down inside the bin: last inserted first
};

UnsortedGenomicRegionSetOverlaps::UnsortedGenomicRegionSetOverlaps(GenomicRegionSet *QuerySet, GenomicRegionSet *IndexSet, const char *bin_bits)
    : GenomicRegionSetOverlaps(QuerySet, IndexSet)
{
  // -B tunes the reference's bin index; the rank structure on the device has no bins, the host-side iteration (GetMatch) does
  match = NULL; bin_bits_ = bin_bits ? bin_bits : "";
  if (IndexSet->load_in_memory == false) { fprintf(stderr, "Error: [UnsortedGenomicRegionSetOverlaps] index regions must be loaded in memory!\n"); exit(1); }
}
UnsortedGenomicRegionSetOverlaps::~UnsortedGenomicRegionSetOverlaps() { delete match; }
GenomicRegion *UnsortedGenomicRegionSetOverlaps::GetQuery()
{
  current_qreg = QuerySet->Get();
  if (current_qreg && !current_qreg->IsCompatibleSortedAndNonoverlapping()) current_qreg->PrintError("query regions should be compatible, sorted and non-overlapping!");
  return current_qreg;
}
GenomicRegion *UnsortedGenomicRegionSetOverlaps::NextQuery()
{
  current_qreg = QuerySet->Next();
  if (current_qreg && !current_qreg->IsCompatibleSortedAndNonoverlapping()) current_qreg->PrintError("query regions should be compatible, sorted and non-overlapping!");
  return current_qreg;
}

GenomicRegion *UnsortedGenomicRegionSetOverlaps::GetMatch()
{
  if (!match) {
    match = new MatchIndex;
    MatchIndex &mx = *match;
    // levels: the listed shift widths, the last one forced to 60 bits = one bin (:5619-5636)
    if (bin_bits_.empty()) mx.bits = {17, 20, 23, 26, 60};
    else {
      size_t p = 0;
      for (;;) { size_t q = bin_bits_.find(',', p); mx.bits.push_back(atoi(bin_bits_.substr(p, q == std::string::npos ? q : q - p).c_str())); if (q == std::string::npos) break; p = q + 1; }
      mx.bits.push_back(60);
    }
    const int L = (int)mx.bits.size();
    std::map<std::string, long int> chrom_size;
    for (long int k = 0; k < IndexSet->n_regions; k++) {
      GenomicRegion *r = IndexSet->R[k];
      if (!r->IsCompatibleSortedAndNonoverlapping()) r->PrintError("index regions should be compatible, sorted and non-overlapping!");
      const long int start = r->I.front()->START, stop = r->I.back()->STOP;
      if (start > stop || stop <= 0) continue;
      std::map<std::string, long int>::iterator it = chrom_size.find(r->I.front()->CHROMOSOME);
      if (it == chrom_size.end()) chrom_size[r->I.front()->CHROMOSOME] = stop; else it->second = std::max(it->second, stop);
    }
    for (std::map<std::string, long int>::iterator it = chrom_size.begin(); it != chrom_size.end(); it++) {
      MatchIndex::Chrom &c = mx.chrom[it->first];
      c.n_bins.resize(L); c.bins.resize(L);
      for (int l = 0; l < L; l++) { c.n_bins[l] = (it->second >> mx.bits[l]) + 1; c.bins[l].resize((size_t)c.n_bins[l]); }
    }
    for (long int k = 0; k < IndexSet->n_regions; k++) {
      GenomicRegion *r = IndexSet->R[k];
      long int start = r->I.front()->START; const long int stop = r->I.back()->STOP;
      if (start > stop || stop <= 0) continue;                                     // :5659
      if (start <= 0) start = 1;
      MatchIndex::Chrom &c = mx.chrom[r->I.front()->CHROMOSOME];
      for (int l = 0; l < L; l++) if ((start >> mx.bits[l]) == (stop >> mx.bits[l])) { c.bins[l][(size_t)(start >> mx.bits[l])].push_back(k); break; }
    }
  }
  MatchIndex &mx = *match;
  std::map<std::string, MatchIndex::Chrom>::iterator it = mx.chrom.find(current_qreg->I.front()->CHROMOSOME);
  mx.cur = it == mx.chrom.end() ? NULL : &it->second;
  if (mx.cur == NULL) return current_ireg = NULL;
  mx.l = 0;
  mx.start = current_qreg->I.front()->START; mx.stop = current_qreg->I.back()->STOP;
  if (mx.stop <= 0) current_qreg->PrintError("stop position must be positive!");
  if (mx.start > mx.stop) current_qreg->PrintError("start position cannot be greater than stop position!");
  if (mx.start <= 0) mx.start = 1;
  mx.b = mx.start >> mx.bits[0];
  mx.b_stop = std::min(mx.stop >> mx.bits[0], mx.cur->n_bins[0] - 1);
  if (mx.b >= mx.cur->n_bins[0]) { mx.cur = NULL; return current_ireg = NULL; }
  mx.at = (long int)mx.cur->bins[0][(size_t)mx.b].size();
  return NextMatch();
}

GenomicRegion *UnsortedGenomicRegionSetOverlaps::NextMatch()
{
  if (!match || match->cur == NULL) return current_ireg = NULL;
  MatchIndex &mx = *match;
  const int L = (int)mx.bits.size();
  for (;;) {
    if (mx.l < L && mx.b <= mx.b_stop && mx.b < mx.cur->n_bins[mx.l]) {
      const std::vector<long int> &bin = mx.cur->bins[mx.l][(size_t)mx.b];
      while (mx.at > 0) {
        GenomicRegion *r = IndexSet->R[bin[(size_t)--mx.at]];
        if (mx.start <= r->I.back()->STOP && mx.stop >= r->I.front()->START) return current_ireg = r;
      }
    }
    mx.b++;
    if (mx.l >= L || mx.b > mx.b_stop) {
      mx.l++;
      if (mx.l >= L) break;
      mx.b = mx.start >> mx.bits[mx.l];
      mx.b_stop = std::min(mx.stop >> mx.bits[mx.l], mx.cur->n_bins[mx.l] - 1);
    }
    mx.at = (mx.b <= mx.b_stop && mx.b < mx.cur->n_bins[mx.l]) ? (long int)mx.cur->bins[mx.l][(size_t)mx.b].size() : 0;
  }
  mx.cur = NULL;
  return current_ireg = NULL;
}
bool UnsortedGenomicRegionSetOverlaps::Done() { return current_qreg == NULL; }

SortedGenomicRegionSetOverlaps::SortedGenomicRegionSetOverlaps(GenomicRegionSet *QuerySet, GenomicRegionSet *IndexSet, bool sorted_by_strand)
    : GenomicRegionSetOverlaps(QuerySet, IndexSet)
{
  this->sorted_by_strand = sorted_by_strand;
  buffer_at_ = 0; index_at_ = 0; have_union_ = false; union_strand_ = '+'; union_start_ = union_stop_ = 0;
  current_qreg = QuerySet->Get();
  current_ireg = IndexSet->Get();
}
SortedGenomicRegionSetOverlaps::~SortedGenomicRegionSetOverlaps() {}

// LoadIndexBuffer (genomic_intervals.cpp:5844-5873): drop the buffer when the query has passed its union interval, then pull index
// regions while the query is not before them, keeping the ones it meets; the order of the index set is checked as it is pulled
void SortedGenomicRegionSetOverlaps::LoadIndexBuffer()
{
  if (current_qreg == NULL) return;
  if (!IndexSet->load_in_memory) { fprintf(stderr, "Error: [SortedGenomicRegionSetOverlaps] per-query iteration needs the index set loaded in memory in this build!\n"); exit(1); }
  if (!buffer_.empty() && have_union_) {
    GenomicInterval u(union_chrom_.c_str(), union_strand_, union_start_, union_stop_);
    GenomicInterval q(current_qreg->I.front()->CHROMOSOME, current_qreg->I.front()->STRAND, current_qreg->I.front()->START, current_qreg->I.back()->STOP);
    if (q.CalcDirection(&u, sorted_by_strand) > 0) { buffer_.clear(); have_union_ = false; }
  }
  while (index_at_ < IndexSet->n_regions) {
    GenomicRegion *r = IndexSet->R[index_at_];
    if (!r->IsCompatibleSortedAndNonoverlapping()) r->PrintError("index regions should be compatible, sorted and non-overlapping!");
    const int d = current_qreg->CalcDirection(r, sorted_by_strand);
    if (d < 0) break;
    if (d == 0) {
      if (buffer_.empty()) { union_chrom_ = r->I.front()->CHROMOSOME; union_strand_ = r->I.front()->STRAND; union_start_ = r->I.front()->START; union_stop_ = r->I.back()->STOP; have_union_ = true; }
      else { union_start_ = std::min(union_start_, r->I.front()->START); union_stop_ = std::max(union_stop_, r->I.back()->STOP); }
      buffer_.push_back(index_at_);
    }
    index_at_++;
    if (index_at_ < IndexSet->n_regions && IndexSet->R[index_at_]->IsBefore(r, sorted_by_strand))
      IndexSet->R[index_at_]->PrintError(std::string("index regions are not sorted (sorted-by-strand = ") + (sorted_by_strand ? "true" : "false") + ")!");
  }
  buffer_at_ = 0;
  current_ireg = buffer_.empty() ? NULL : IndexSet->R[buffer_[0]];
}

GenomicRegion *SortedGenomicRegionSetOverlaps::GetQuery()
{
  current_qreg = QuerySet->Get();
  if (current_qreg && !current_qreg->IsCompatibleSortedAndNonoverlapping()) current_qreg->PrintError("query regions should be compatible, sorted and non-overlapping!");
  LoadIndexBuffer();
  return current_qreg;
}
GenomicRegion *SortedGenomicRegionSetOverlaps::NextQuery()
{
  GenomicRegion *prev = QuerySet->Get();
  GenomicRegion *next = QuerySet->Next(true);
  if (next != NULL && !next->IsCompatibleSortedAndNonoverlapping()) next->PrintError("query regions should be compatible, sorted and non-overlapping!");
  if (next != NULL && prev != NULL && next->IsBefore(prev, sorted_by_strand))
    next->PrintError(std::string("query regions are not sorted (sorted-by-strand = ") + (sorted_by_strand ? "true" : "false") + ")!");
  if (!QuerySet->load_in_memory && next != NULL) delete prev;
  current_qreg = next;
  LoadIndexBuffer();
  return current_qreg;
}
// :5903-5918: buffered regions the query has passed are dropped, the first one ahead of it ends the walk
GenomicRegion *SortedGenomicRegionSetOverlaps::GetMatch()
{
  if (current_qreg == NULL || current_ireg == NULL) return NULL;
  for (;;) {
    const int d = current_qreg->CalcDirection(current_ireg, sorted_by_strand);
    if (d > 0) {
      buffer_.erase(buffer_.begin() + (long)buffer_at_);
      if (buffer_at_ >= buffer_.size()) return current_ireg = NULL;
      current_ireg = IndexSet->R[buffer_[buffer_at_]];
    } else if (d < 0) return NULL;
    else return current_ireg;
  }
}
GenomicRegion *SortedGenomicRegionSetOverlaps::NextMatch()
{
  buffer_at_++;
  current_ireg = buffer_at_ >= buffer_.size() ? NULL : IndexSet->R[buffer_[buffer_at_]];
  if (current_ireg == NULL) return NULL;
  return GetMatch();
}
bool SortedGenomicRegionSetOverlaps::Done() { return current_qreg == NULL || (index_at_ >= IndexSet->n_regions && buffer_.empty()); }

// ---------------------------------------------------------------------------------------------------
// scanners
// ---------------------------------------------------------------------------------------------------
GenomicRegionSetScanner::GenomicRegionSetScanner(GenomicRegionSet *R, StringLIntMap *bounds, long int win_step, long int win_size,
                                                 long int max_label_value, bool ignore_strand, char preprocess)
{
  if (R->format == "SEQ") { std::cerr << "Error: this operation does not accept SEQ format!\n"; exit(1); }
  if (bounds == NULL) { std::cerr << "Error: this operation requires genomic bounds!\n"; exit(1); }
  this->R = R; this->bounds = bounds; this->win_step = win_step; this->win_size = win_size;
  this->max_label_value = max_label_value; this->ignore_strand = ignore_strand; this->preprocess = preprocess;
  if (win_step <= 0 || win_size % win_step != 0) { std::cerr << "Error: window size must be a multiple of window step in 'GenomicRegionSetScanner'!\n"; exit(1); }
  n_win_combine = win_size / win_step;
  cur_block = 0; cur_win = 0; computed = false; total_label_value = 0;
  halt_set = false; halt_block = 0; halt_win = 0; halt_line = 0; halt_no_prefix = false;
}

GenomicRegionSetScanner::~GenomicRegionSetScanner() {}

void GenomicRegionSetScanner::Compute(bool sorted_rules)
{
  computed = true;
  if (win_step > INT_MAX || win_size > INT_MAX) { std::cerr << "Error: window geometry does not fit the MI355X path!\n"; exit(1); }
  ChromTable chroms;
  for (StringLIntMap::iterator p = bounds->begin(); p != bounds->end(); p++) chroms.Add(p->first.c_str());
  chroms.Freeze();
  const int n_chrom = chroms.size(), ns = ignore_strand ? 1 : 2;
  // iteration order of the reference: chromosomes in map order, '+' block then '-' block
  std::vector<int32_t> class_len((size_t)std::max(1, n_chrom * ns));
  std::vector<int64_t> class_off((size_t)std::max(1, n_chrom * ns));
  long long total = 0;
  chrom_names.clear(); n_windows.clear(); block_offset.clear();
  for (int r = 0; r < n_chrom; r++) {
    long int len = (*bounds)[chroms.name(r)];
    if (len >= INT_MAX - 1) { std::cerr << "Error: chromosome length does not fit the MI355X path!\n"; exit(1); }
    chrom_names.push_back(chroms.name(r));
    for (int s = 0; s < ns; s++) {
      long long nw = gtx_scan_n_windows(len < 0 ? 0 : len, win_step, win_size);
      class_len[(size_t)(s * n_chrom + r)] = (int32_t)(len < 0 ? 0 : len);
      class_off[(size_t)(s * n_chrom + r)] = total;
      n_windows.push_back((long int)nw); block_offset.push_back(total);
      total += nw;
    }
  }
  values.assign((size_t)std::max<long long>(total, 1), 0);
  if (n_chrom == 0) return;
  if (sorted_rules && preprocess == 'p') { ComputeMappable(class_len, class_off); return; }

  PackOptions opt;
  opt.mode = sorted_rules ? gtxhost::PACK_SCAN_SORTED : gtxhost::PACK_SCAN_UNSORTED;
  opt.chroms = &chroms; opt.strand_aware = !ignore_strand; opt.sorted_by_strand = !ignore_strand;
  opt.max_label_value = max_label_value;
  opt.keep_prefix_on_error = sorted_rules;
  PackError input_error;                                           // sorted rules: met where the reference's walk meets it (below), not here
  std::function<void(const PackError &)> on_error;
  if (sorted_rules) on_error = [&](const PackError &e) { input_error = e; };
  std::vector<int32_t> tri, w;
  bool bad_preprocess = false;                                     // raised at the first region that is processed, like the reference
  const bool preprocess_ok = sorted_rules ? preprocess == '1' : (preprocess == '1' || preprocess == 'c');
  const char prep = (preprocess == 'c' && !sorted_rules) ? 'c' : '1';
  const uint32_t rule_flags = sorted_rules ? GTX_ZERO_LENGTH_OK : 0u;
  gtx_group *grp = NULL;
  gtx_ctx *one = NULL;                                             // one GPU: the scan is fed as a stream (gtx_scan_begin .. gtx_scan_end)
  auto device_side = [&] {
    grp = Devices();
    if (gtx_group_size(grp) != 1) return;
    one = gtx_group_ctx(grp, 0);
    if (gtx_scan_begin(one, class_len.data(), n_chrom * ns, (int32_t)win_step, (int32_t)win_size, prep, rule_flags, max_label_value > 1, class_off.data()) != GTX_OK) {
      fflush(stdout); fprintf(stderr, "\nError: [gtx] %s\n", gtx_last_error(one)); exit(1);
    }
  };
  auto check_one = [&](int rc) { if (rc != GTX_OK) { fflush(stdout); fprintf(stderr, "\nError: [gtx %d] %s\n", rc, gtx_last_error(one)); exit(1); } };
  // the input's text tokenised on the device where that applies (one GPU, a streamed BED input: TextOnDevice)
  TextSink text_sink;
  text_sink.usable = [&] { return one != NULL; };
  text_sink.needs_host = [&](int ticket) { int redo = 0; check_one(gtx_text_result(one, ticket, &redo)); return redo != 0; };
  text_sink.add = [&](const char *text, size_t bytes, int64_t lines, const gtx_text_rules &rules) {
    if (!preprocess_ok) { bad_preprocess = true; g_drain_stop = true; return -1; }
    int ticket = -1;
    check_one(gtx_scan_add_text(one, text, bytes, lines, &rules, (sorted_rules || TextLooksSorted(text, bytes)) ? 0u : GTX_READS_UNSORTED, &ticket));
    return ticket;
  };
  DrainSet(R, opt, device_side, [&](const PackedBatch &b) {
    if (!preprocess_ok) { bad_preprocess = true; g_drain_stop = true; return; }
    total_label_value += (long int)b.label_sum;
    if (b.tri.empty()) return;
    if (one) { check_one(gtx_scan_add(one, b.tri.data(), b.w.empty() ? NULL : b.w.data(), (int64_t)(b.tri.size() / 3), LooksSorted(b.tri) ? 0u : GTX_READS_UNSORTED)); return; }
    tri.insert(tri.end(), b.tri.begin(), b.tri.end());
    w.insert(w.end(), b.w.begin(), b.w.end());
  }, &text_sink, on_error);
  if (input_error.set && !bad_preprocess) {
    // Where does the reference's walk fetch the offending line?  When it consumes the region in front of it (R->Next, :4944 / :4936):
    // inside that region's block at the micro-window its start falls into -- the windows whose last micro-window lies before that
    // one are out by then --, or, for a region no block takes (a chromosome without bounds, a start behind the last micro-window),
    // by the skip loop at the head of the first block behind it (:4934); a region behind every block is never consumed, and neither
    // is the line behind it met.  No region in front of the line: the constructor's first read meets it (:4882).
    const long int comb = win_size / win_step;
    halt_set = true; halt_block = 0; halt_win = 0;
    halt_line = input_error.line; halt_no_prefix = input_error.no_prefix; halt_msg = input_error.msg;
    if (input_error.have_last) {
      const size_t ci = (size_t)(std::lower_bound(chrom_names.begin(), chrom_names.end(), input_error.last_chrom) - chrom_names.begin());
      const bool known = ci < chrom_names.size() && chrom_names[ci] == input_error.last_chrom;
      size_t at_head_of = ci * (size_t)ns;                         // unknown chromosome: skipped at the head of the first block of a later one
      if (known) {
        const size_t B = ci * (size_t)ns + ((!ignore_strand && input_error.last_strand == '-') ? 1 : 0);
        const long int n_mw = std::max<long int>(0, (*bounds)[chrom_names[ci]]) / win_step;
        const long int k = input_error.last_start <= 0 ? 1 : (input_error.last_start + win_step - 1) / win_step;
        if (k <= n_mw) { halt_block = B; halt_win = std::max<long int>(0, std::min<long int>(k - comb, n_windows[B])); at_head_of = (size_t)-1; }
        else at_head_of = B + 1;
      }
      if (at_head_of != (size_t)-1) {
        if (at_head_of < n_windows.size()) { halt_block = at_head_of; halt_win = 0; }
        else halt_set = false;                                     // never consumed: the walk ends without meeting the line
      }
    }
  }
  if (bad_preprocess && sorted_rules) { fprintf(stderr, "Error: [SortedGenomicRegionSetScanner] preprocess operator '%c' not supported!\n", preprocess); exit(1); }
  if (bad_preprocess) { fprintf(stderr, "Error: [UnsortedGenomicRegionSetScanner] preprocess operator '%c' not supported!\n", preprocess); exit(1); }
  if (!grp) device_side();                                         // (an in-memory input: DrainSet has not run the hand-over's first step)
  if (one) {
    int64_t text_labels = 0;                                       // the lines the device took: their label values are summed there
    check_one(gtx_scan_end(one, (uint64_t *)values.data(), &text_labels));
    total_label_value += (long int)text_labels;
    return;
  }
  CheckGrp(grp, gtx_group_scan(grp, tri.data(), w.empty() ? NULL : w.data(), (int64_t)(tri.size() / 3), class_len.data(), n_chrom * ns,
                               (int32_t)win_step, (int32_t)win_size, prep, rule_flags | (LooksSortedVec(tri) ? GTX_READS_SORTED : 0u),
                               (uint64_t *)values.data(), class_off.data()));
}

// The sorted scanner's operator 'p' (genomic_intervals.cpp:4939-4942; what `genomic_scans peaks` scans its mappability track with): a
// micro-window receives the part of a region that lies at or below its stop -- and the walk behind it is what the reference wrote, not
// what one would expect: after a region that ends inside the micro-window it moves on twice (:4940, then :4945: the region behind it
// is never looked at), after one that reaches beyond the micro-window once, with the region's start set to stop + 1 by then (:4941):
// the rest of that region is dropped and the order check of the pull (:3879) sees the moved start.  Which region is looked at
// depends on every region before it, so this is a host-side walk over the set's own iterators (a mappability track is a secondary
// input, read once); what it yields is (block, micro-window, amount), and those go to the device as weighted point reads: the
// micro-window histogram and the window sums are the scan kernels' as for any other input.  An input error is met where the walk
// meets it (halt_*), like the sorted scanner's other errors.  One case the reference leaves undefined: the first of the two pulls
// meeting the end of the stream (its second pull deletes the last region again); the walk ends there here.
void GenomicRegionSetScanner::ComputeMappable(const std::vector<int32_t> &class_len, const std::vector<int64_t> &class_off)
{
  const int n_chrom = (int)chrom_names.size(), ns = ignore_strand ? 1 : 2;
  const bool by_strand = !ignore_strand;
  std::vector<int32_t> tri, w;
  auto emit = [&](int cls, long int mw, long int amount) {
    const int32_t pos = (int32_t)((mw - 1) * win_step + 1);          // a position of that micro-window
    while (amount != 0) {                                              // (a weight is 32 bits; an amount beyond that goes in pieces)
      const long int piece = std::max<long int>(INT_MIN + 1, std::min<long int>(INT_MAX, amount));
      tri.push_back(cls); tri.push_back(pos); tri.push_back(pos); w.push_back((int32_t)piece);
      amount -= piece;
    }
  };
  LoadError err;
  size_t at_block = 0; long int at_mw = 0;                             // where the walk is: what the halt position is read off
  tls_load_error = &err;
  try {
    GenomicRegion *r = R->Get();
    for (int c = 0; c < n_chrom && r; c++)
      for (int s = 0; s < ns && r; s++) {
        const char *chrom = chrom_names[(size_t)c].c_str();
        const char strand = s ? '-' : '+';
        at_block = (size_t)(c * ns + s); at_mw = 0;
        for (;;) {                                                     // the regions that sort before this block (:4934)
          if (!r) break;
          const int t = strcmp(chrom, r->I.front()->CHROMOSOME);
          if (!(t > 0 || (t == 0 && strand > r->I.front()->STRAND))) break;
          r = R->Next(by_strand, false);
        }
        const long int n_mw = (long int)class_len[(size_t)(s * n_chrom + c)] / win_step;
        while (r && strcmp(r->I.front()->CHROMOSOME, chrom) == 0 && (ignore_strand || r->I.front()->STRAND == strand)) {
          GenomicInterval *i = r->I.front();
          const long int first = i->START <= 0 ? 1 : (i->START + win_step - 1) / win_step;   // the first micro-window whose stop reaches the start
          if (std::max(first, at_mw) > n_mw) break;                    // (left for the skip loop of the next block)
          at_mw = std::max<long int>(std::max(first, at_mw), 1);
          if (r->I.size() != 1) r->PrintError("single-interval regions expected for this operation!\n");
          const long int stop = at_mw * win_step;
          if (i->STOP <= stop) {
            emit(s * n_chrom + c, at_mw, i->STOP - i->START + 1);
            r = R->Next(by_strand, false);
            if (!r) break;
          } else {
            emit(s * n_chrom + c, at_mw, stop - i->START + 1);
            i->START = stop + 1;
          }
          r = R->Next(by_strand, false);
        }
      }
  } catch (const LoadAbort &) {}
  tls_load_error = NULL;
  if (err.set) {
    const long int comb = win_size / win_step;
    halt_set = true; halt_line = err.line; halt_msg = err.msg; halt_no_prefix = !err.with_prefix;
    halt_block = at_block; halt_win = std::max<long int>(0, std::min<long int>(at_mw - comb, n_windows[at_block]));
  }
  gtx_group *grp = Devices();
  const int64_t n = (int64_t)w.size();
  const uint32_t order = LooksSortedVec(tri) ? 0u : GTX_READS_UNSORTED;
  if (gtx_group_size(grp) == 1) {
    gtx_ctx *one = gtx_group_ctx(grp, 0);
    auto check = [&](int rc) { if (rc != GTX_OK) { fflush(stdout); fprintf(stderr, "\nError: [gtx %d] %s\n", rc, gtx_last_error(one)); exit(1); } };
    check(gtx_scan_begin(one, class_len.data(), n_chrom * ns, (int32_t)win_step, (int32_t)win_size, '1', GTX_ZERO_LENGTH_OK, 1, class_off.data()));
    if (n) check(gtx_scan_add(one, tri.data(), w.data(), n, order));
    check(gtx_scan_end(one, (uint64_t *)values.data(), NULL));
    return;
  }
  CheckGrp(grp, gtx_group_scan(grp, tri.data(), w.data(), n, class_len.data(), n_chrom * ns, (int32_t)win_step, (int32_t)win_size, '1',
                               GTX_ZERO_LENGTH_OK | (order ? 0u : GTX_READS_SORTED), (uint64_t *)values.data(), class_off.data()));
}

long int GenomicRegionSetScanner::TotalLabelValue()
{
  if (!computed) Compute(false);
  return total_label_value;
}

// genomic_intervals.cpp:6206-6214: every region of the file, by the unsorted reader's rules (no order check; a line it cannot read
// ends the run), its label value capped.  The scanners deliver this sum with their windows (TotalLabelValue); this pass is for the
// caller that needs it although a sorted scanner stopped at an input error.
long int CountGenomicRegions(char *reg_file, long int max_label_value)
{
  GenomicRegionSet set(reg_file, 10000, false, false, true);
  ChromTable none; none.Freeze();
  PackOptions opt;
  opt.mode = gtxhost::PACK_SCAN_UNSORTED; opt.chroms = &none; opt.max_label_value = max_label_value;
  long int n = 0;
  DrainSet(&set, opt, [] {}, [&](const PackedBatch &b) { n += (long int)b.label_sum; });
  return n;
}

// genomic_intervals.cpp:6032-6040: the sizes of the regions' intervals, summed over the file by the plain reader
unsigned long int CalcRegSize(char *reg_file)
{
  GenomicRegionSet set(reg_file, 100000, false, false, true);
  unsigned long int n = 0;
  for (GenomicRegion *r = set.Get(); r != NULL; r = set.Next()) n += (unsigned long int)r->GetSize(true);
  return n;
}

unsigned long int CalcBoundSize(StringLIntMap *bounds)
{
  unsigned long int y = 0;
  for (StringLIntMap::iterator x = bounds->begin(); x != bounds->end(); x++) y += (unsigned long int)x->second;
  return y;
}

void GenomicRegionSetScanner::RaiseHalt()
{
  PackError e; e.set = true; e.line = halt_line; e.no_prefix = halt_no_prefix; e.msg = halt_msg;
  fflush(stdout);
  DiePack(e);
}

long int GenomicRegionSetScanner::Next()
{
  if (!computed) Compute(false);
  while (cur_block < n_windows.size()) {
    if (halt_set && (cur_block > halt_block || (cur_block == halt_block && cur_win >= halt_win))) RaiseHalt();
    if (cur_win < n_windows[cur_block]) { cur_win++; return (long int)values[(size_t)(block_offset[cur_block] + cur_win - 1)]; }
    cur_block++; cur_win = 0;
  }
  if (halt_set) RaiseHalt();
  return -1;
}

void GenomicRegionSetScanner::PrintRemaining(FILE *out_file, long int min_value)
{
  if (!computed) Compute(false);
  const int ns = ignore_strand ? 1 : 2;
  std::vector<char> buf; buf.reserve(8u << 20);
  auto put_num = [&](long int v) {
    char tmp[24]; int n = 0;
    unsigned long int u = v < 0 ? 0ul - (unsigned long int)v : (unsigned long int)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) buf.push_back('-');
    while (n) buf.push_back(tmp[--n]);
  };
  for (; cur_block < n_windows.size(); cur_block++, cur_win = 0) {
    const std::string &chrom = chrom_names[cur_block / ns];
    const char strand = (cur_block % ns) ? '-' : '+';
    for (; cur_win < n_windows[cur_block]; cur_win++) {
      if (halt_set && (cur_block > halt_block || (cur_block == halt_block && cur_win >= halt_win))) goto done;
      const long int v = (long int)values[(size_t)(block_offset[cur_block] + cur_win)];
      if (v == -1) { cur_win++; goto done; }                      // (a caller's loop takes Next() == -1 for the end, whatever made the value)
      if (v < min_value) continue;
      put_num(v); buf.push_back('\t');
      buf.insert(buf.end(), chrom.begin(), chrom.end()); buf.push_back(' '); buf.push_back(strand); buf.push_back(' ');
      put_num(win_step * cur_win + 1); buf.push_back(' '); put_num(win_step * cur_win + win_size); buf.push_back('\n');
      if (buf.size() > (7u << 20)) { fwrite(buf.data(), 1, buf.size(), out_file); buf.clear(); }
    }
  }
done:
  if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), out_file);
  if (halt_set && (cur_block >= n_windows.size() || cur_block > halt_block || (cur_block == halt_block && cur_win >= halt_win))) { fflush(out_file); RaiseHalt(); }
}

void GenomicRegionSetScanner::PrintInterval(FILE *out_file)
{
  const int ns = ignore_strand ? 1 : 2;
  fprintf(out_file, "%s %c %ld %ld", chrom_names[cur_block / ns].c_str(), (cur_block % ns) ? '-' : '+', win_step * (cur_win - 1) + 1,
          win_step * (cur_win - 1) + win_size);
}

GenomicInterval *GenomicRegionSetScanner::GetInterval()
{
  const int ns = ignore_strand ? 1 : 2;
  return new GenomicInterval(chrom_names[cur_block / ns].c_str(), (cur_block % ns) ? '-' : '+', win_step * (cur_win - 1) + 1,
                             win_step * (cur_win - 1) + win_size);
}

// reference filter of `genomic_scans counts -r`: windows are produced by the GPU scan as always; which of them
// are reported is a host-side question per window (at most one rank query or one merge step each)
long int GenomicRegionSetScanner::Next(GenomicRegionSet *Ref)
{
  if (Ref == NULL) return Next();
  GenomicRegion *q = Ref->Get();
  const bool sorted_by_strand = !ignore_strand;
  while (q != NULL) {
    const long int c = Next();
    if (c == -1) return -1;
    const int ns = ignore_strand ? 1 : 2;
    GenomicInterval w(chrom_names[cur_block / ns].c_str(), (cur_block % ns) ? '-' : '+', win_step * (cur_win - 1) + 1, win_step * (cur_win - 1) + win_size);
    while (q != NULL) {
      const int d = q->I.front()->CalcDirection(&w, sorted_by_strand);
      if (d < 0) q = Ref->Next(sorted_by_strand, false);
      else if (d == 0) return c;
      else break;
    }
  }
  return -1;
}

long int GenomicRegionSetScanner::Next(GenomicRegionSetIndex *index)
{
  if (index == NULL) return Next();
  while (true) {
    const long int c = Next();
    if (c == -1) return -1;
    const int ns = ignore_strand ? 1 : 2;
    GenomicInterval w(chrom_names[cur_block / ns].c_str(), (cur_block % ns) ? '-' : '+', win_step * (cur_win - 1) + 1, win_step * (cur_win - 1) + win_size);
    if (index->GetOverlap(&w, false, ignore_strand) != NULL) return c;
  }
}

// ---- GenomicRegionSetIndex ---------------------------------------------------------------------------
struct GenomicRegionSetIndex::Impl {
  struct Track { std::vector<long> start; std::vector<long> max_stop; std::vector<long> arg; };   // sorted by start; running max of stop
  std::map<std::string, Track> by_chrom[3];                       // 0: '+', 1: '-', 2: both strands
};

GenomicRegionSetIndex::GenomicRegionSetIndex(GenomicRegionSet *regSet, const char *)
{
  this->regSet = regSet;
  impl = new Impl;
  if (!regSet->load_in_memory) regSet->PrintError("[GenomicRegionSetIndex] the region set must be loaded in memory!");
  struct Item { long start, stop, k; };
  std::map<std::string, std::vector<Item>> items[3];
  for (long k = 0; k < regSet->n_regions; k++) {
    GenomicRegion *r = regSet->R[k];
    if (r->I.size() != 1) r->PrintError("multi-interval regions are outside the MI355X path!");
    GenomicInterval *i = r->I.front();
    if (i->STOP <= 0 || i->START > i->STOP) continue;              // never found (genomic_intervals.cpp:5659)
    items[i->STRAND == '-' ? 1 : 0][i->CHROMOSOME].push_back({i->START, i->STOP, k});
    items[2][i->CHROMOSOME].push_back({i->START, i->STOP, k});
  }
  for (int s = 0; s < 3; s++)
    for (auto &kv : items[s]) {
      std::stable_sort(kv.second.begin(), kv.second.end(), [](const Item &a, const Item &b) { return a.start < b.start; });
      Impl::Track &t = impl->by_chrom[s][kv.first];
      long best = LONG_MIN, who = -1;
      for (const Item &it : kv.second) {
        if (it.stop > best) { best = it.stop; who = it.k; }
        t.start.push_back(it.start); t.max_stop.push_back(best); t.arg.push_back(who);
      }
    }
}

GenomicRegionSetIndex::~GenomicRegionSetIndex() { delete impl; }

GenomicRegion *GenomicRegionSetIndex::GetOverlap(GenomicInterval *i, bool, bool ignore_strand)
{
  const auto &m = impl->by_chrom[ignore_strand ? 2 : (i->STRAND == '-' ? 1 : 0)];
  auto it = m.find(i->CHROMOSOME);
  if (it == m.end()) return NULL;
  const Impl::Track &t = it->second;
  // regions with start <= i->STOP; among them the largest stop must reach i->START
  const size_t n = (size_t)(std::upper_bound(t.start.begin(), t.start.end(), i->STOP) - t.start.begin());
  if (n == 0 || t.max_stop[n - 1] < i->START) return NULL;
  return regSet->R[t.arg[n - 1]];
}

SortedGenomicRegionSetScanner::SortedGenomicRegionSetScanner(GenomicRegionSet *R, StringLIntMap *bounds, long int win_step, long int win_size,
                                                             long int max_label_value, bool ignore_strand, char preprocess)
    : GenomicRegionSetScanner(R, bounds, win_step, win_size, max_label_value, ignore_strand, preprocess)
{
  Compute(true);
}

UnsortedGenomicRegionSetScanner::UnsortedGenomicRegionSetScanner(GenomicRegionSet *R, StringLIntMap *bounds, long int win_step, long int win_size,
                                                                 long int max_label_value, bool ignore_strand, char preprocess)
    : GenomicRegionSetScanner(R, bounds, win_step, win_size, max_label_value, ignore_strand, preprocess)
{
  Compute(false);
}

// the reference's scanners override the five virtuals (genomic_intervals.h:2291-2295, :2349-2353); here both run the shared body
#define GTX_SCANNER_OVERRIDES(CLS)                                                                                  \
  CLS::~CLS() {}                                                                                                    \
  void CLS::PrintInterval(FILE *out_file) { GenomicRegionSetScanner::PrintInterval(out_file); }                     \
  GenomicInterval *CLS::GetInterval() { return GenomicRegionSetScanner::GetInterval(); }                            \
  long int CLS::Next() { return GenomicRegionSetScanner::Next(); }                                                  \
  long int CLS::Next(GenomicRegionSet *Ref) { return GenomicRegionSetScanner::Next(Ref); }                          \
  long int CLS::Next(GenomicRegionSetIndex *index) { return GenomicRegionSetScanner::Next(index); }
GTX_SCANNER_OVERRIDES(SortedGenomicRegionSetScanner)
GTX_SCANNER_OVERRIDES(UnsortedGenomicRegionSetScanner)
#undef GTX_SCANNER_OVERRIDES

// ---------------------------------------------------------------------------------------------------
StringLIntMap *ReadBounds(char *genome_reg_file, bool verbose)
{
  if (genome_reg_file == NULL || strlen(genome_reg_file) == 0) { std::cerr << "Error: genome region file is necessary for this operation!\n"; exit(1); }
  StringLIntMap *bounds = new StringLIntMap();
  GenomicRegionSet RegSet(genome_reg_file, 10000, verbose, false, true);
  long int line = 1;
  for (GenomicRegion *r = RegSet.Get(); r != NULL; r = RegSet.Next(), line++) {
    std::string chr = r->I.front()->CHROMOSOME;
    if (bounds->find(chr) == bounds->end()) (*bounds)[chr] = r->I.front()->STOP;
    else if ((*bounds)[chr] != r->I.front()->STOP) {
      std::cerr << "Error: chromosome " << chr << " has multiple lengths in genome file '" << genome_reg_file << "' line " << line << "!\n";
      exit(1);
    }
  }
  return bounds;
}
