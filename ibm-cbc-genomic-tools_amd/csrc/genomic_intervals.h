// genomic_intervals.h -- the GenomicTools class API for the count / scan hot path, MI355X edition.
//
// Same class names, constructor signatures, public members and error behaviour as the reference's
// gtools/genomic_intervals.h for the classes on the path (file:line of each counterpart is given
// at the declaration), so that callers written like gtools/genomic_overlaps.cpp:408-431 or
// gtools/genomic_scans.cpp:399-436 recompile against this header unchanged.  Underneath nothing
// is shared with the reference: region sets keep their input as a line stream that is parsed in
// bulk by a thread pool into packed int32 triples (gtx_bed.h), and the reductions
// (CountIndexOverlaps, the scanners' window sums) are one call each into the C ABI of libgtx.so
// (include/gtx.h), i.e. HIP kernels on the MI355X.  There is no CPU implementation of those
// reductions here; without a GPU they fail with an error message and exit(1), like every other
// error of the reference (genomic_intervals.cpp:1001-1006).
//
// Scope: BED3..BED6 single-interval regions.  REG/SAM/GFF/SEQ input, BED12
// blocks, the ~45 Run*/Print* text transforms of GenomicRegionSet, GenomicRegionSetIndex beyond the
// "does anything overlap" query of the scanners' reference filter, and the per-pair enumeration
// (GetMatch/NextMatch) are outside the path.
#ifndef GTX_GENOMIC_INTERVALS_H
#define GTX_GENOMIC_INTERVALS_H

#include <stdio.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace gtxhost { class LineSource; struct GtxView; }

typedef std::map<std::string, long int> StringLIntMap;          // genomic_intervals.h:41

extern bool _MESSAGES_;                                          // verbose switch (core.h; set from -v)

// Region objects of an in-memory set are built by several threads at once (GenomicRegionSet::Init); the general heap does not
// scale there (every arena growth takes the address-space lock that page faults need), so those objects, their strings and their
// interval vectors come out of per-thread blocks owned by the set.  Objects made anywhere else use the ordinary heap; `delete`
// works on both.
void *GtxRegionAlloc(size_t bytes);                               // from the calling thread's block if it has one, else operator new
void GtxRegionFree(void *p);                                      // nothing for block memory, operator delete otherwise
template <class T> struct GtxRegionAllocator {
  typedef T value_type;
  GtxRegionAllocator() {}
  template <class U> GtxRegionAllocator(const GtxRegionAllocator<U> &) {}
  T *allocate(size_t n) { return (T *)GtxRegionAlloc(n * sizeof(T)); }
  void deallocate(T *p, size_t) { GtxRegionFree(p); }
  template <class U> bool operator==(const GtxRegionAllocator<U> &) const { return true; }
  template <class U> bool operator!=(const GtxRegionAllocator<U> &) const { return false; }
};

// ---- GenomicInterval (genomic_intervals.h:91) ---------------------------------------------------------
class GenomicInterval
{
 public:
  GenomicInterval(const char *chromosome, char strand, long int start, long int stop, long int n_line = 0);
  ~GenomicInterval();
  void PrintInterval();
  void PrintInterval(FILE *file_ptr);
  size_t GetSize() { return (size_t)(STOP - START + 1); }
  int CalcDirection(GenomicInterval *i, bool sorted_by_strand);   // <0 before i, 0 overlapping, >0 after (genomic_intervals.cpp:448-459)
  bool OverlapsWith(GenomicInterval *i, bool ignore_strand);      // :624-630
  long int CalcOverlap(GenomicInterval *i, bool ignore_strand);   // :427-432

  static void *operator new(size_t n) { return GtxRegionAlloc(n); }
  static void operator delete(void *p) { GtxRegionFree(p); }

  char *CHROMOSOME;
  char STRAND;
  long int START, STOP;          // 1-based, inclusive
  long int n_line;
};

typedef std::vector<GenomicInterval *, GtxRegionAllocator<GenomicInterval *> > GenomicIntervalSet;   // genomic_intervals.h:46 (a vector of pointers there too)

// ---- GenomicRegion (genomic_intervals.h:591) / GenomicRegionBED (:1112) ------------------------------
class GenomicRegion
{
 public:
  GenomicRegion();
  virtual ~GenomicRegion();
  void PrintError(std::string error_msg);                         // "\nError: Line N: msg\n", exit(1)
  char *GetChromosome() { return I.front()->CHROMOSOME; }
  size_t GetSize(bool skip_gaps);                                 // genomic_intervals.cpp:1049-1058
  long int GetLabelValue(long int max_label_value);               // :1081-1085
  bool IsBefore(GenomicRegion *r, bool sorted_by_strand);         // :1177-1180
  bool IsCompatibleSortedAndNonoverlapping();                     // :1153-1161 (intervals of one region: same chromosome and strand, start-sorted, disjoint)
  bool OverlapsWith(GenomicRegion *r, bool ignore_strand);        // :1167-1172 (any interval pair)
  long int CalcOverlap(GenomicRegion *r, bool ignore_strand);     // :1196-1202 (sum over interval pairs)
  int CalcDirection(GenomicRegion *r, bool sorted_by_strand);     // :1225-1236 (on the envelopes)

  static void *operator new(size_t n) { return GtxRegionAlloc(n); }
  static void operator delete(void *p) { GtxRegionFree(p); }

  long int n_line;
  char *LABEL;
  GenomicIntervalSet I;
};

class GenomicRegionBED : public GenomicRegion
{
 public:
  // parses one BED line (the line is modified); genomic_intervals.cpp:2157-2182.  A 12-column line becomes a region of several
  // intervals (its blocks, :2174-2181)
  GenomicRegionBED(char *inp, long int n_line);
  long int n_tokens;
};

// ---- GenomicRegionSet (genomic_intervals.h:1828) ------------------------------------------------------
class GenomicRegionSet
{
 public:
  GenomicRegionSet(char *file, unsigned long int buffer_size, bool verbose, bool load_in_memory, bool hide_header = true);
  // the same from an open stream (genomic_intervals.h:1836, .cpp:3685-3696); stdin is never loaded in memory
  GenomicRegionSet(FILE *file_ptr, unsigned long int buffer_size, bool verbose, bool load_in_memory, bool hide_header = true);
  ~GenomicRegionSet();

  GenomicRegion *Get();                                           // genomic_intervals.cpp:3845-3849
  GenomicRegion *Next(bool retain_current = false);               // :3855-3867
  GenomicRegion *Next(bool sorted_by_strand, bool retain_current); // same + order check (:3874-3882)
  void Reset();
  void PrintError(std::string error_msg);                         // "\nError: msg\n", exit(1)

  // MI355X path: hands the not yet consumed part of a streaming set (the current region's raw
  // line first) to the bulk packer.  After this call Get()/Next() report the end of the set.
  gtxhost::LineSource *DetachStream(std::string *current_line, long int *current_line_no);
  // the same for a packed region file (format "GTX"): the view and the index of the current record
  const gtxhost::GtxView *DetachPacked(long int *current_record);

  char *file;
  FILE *file_ptr;                                                  // the FILE* constructor's stream (NULL otherwise)
  unsigned long int buffer_size;
  bool verbose, load_in_memory, from_stdin, hide_header;
  long int StreamBytesLeft();                     // size of a streamed regular text file, -1 otherwise (MI355X build: the device-side tokenizer's test)
  long int n_regions;
  std::string format;                                              // "BED", "EMPTY" or "GTX" (a packed region file, gtx_bed.h)
  GenomicRegion **R;

 private:
  void Init();
  void DetectFormat(const char *first_line);
  gtxhost::LineSource *src;
  gtxhost::GtxView *packed;                                        // format "GTX"
  std::vector<std::pair<void *, size_t> > blocks_;                  // memory of the region objects built in parallel (GtxRegionAlloc)
  GenomicRegion *PackedRegion(long int record);                    // region object of one record of the packed file
  std::string cur_raw;                                             // unparsed copy of the current line (streaming mode)
  long int r_index;
};

// ---- GenomicRegionSetOverlaps (genomic_intervals.h:2387) ----------------------------------------------
class GenomicRegionSetOverlaps
{
 public:
  GenomicRegionSetOverlaps(GenomicRegionSet *QuerySet, GenomicRegionSet *IndexSet);
  virtual ~GenomicRegionSetOverlaps();

  virtual GenomicRegion *GetQuery() = 0;
  virtual GenomicRegion *NextQuery() = 0;
  virtual GenomicRegion *GetMatch() = 0;
  virtual GenomicRegion *NextMatch() = 0;
  virtual bool Done() = 0;

  // Per-query iteration, on the host like the reference's (it hands out GenomicRegion pointers): the index regions that overlap the
  // current query after the gap / strand filter (genomic_intervals.cpp:5224-5248), and the two per-query sums over them --
  // label values (:5291-5296) and overlap lengths x label values of the INDEX regions (:5254-5263).
  GenomicRegion *GetOverlap(bool match_gaps, bool ignore_strand);
  GenomicRegion *NextOverlap(bool match_gaps, bool ignore_strand);
  unsigned long int CalcQueryCoverage(bool match_gaps, bool ignore_strand, long int max_label_value);
  unsigned long int CountQueryOverlaps(bool match_gaps, bool ignore_strand, long int max_label_value);

  // hits[k] = sum over query regions q of w_q * [q overlaps index region k], index FILE order;
  // a new[] array the caller releases (genomic_intervals.cpp:5304-5317).  Runs on the GPU.
  unsigned long int *CountIndexOverlaps(bool match_gaps, bool ignore_strand, long int max_label_value);

  // coverage[k] = sum over query regions q overlapping index region k of w_q * overlap length
  // (genomic_intervals.cpp:5269-5285); same ownership and device path as CountIndexOverlaps.
  unsigned long int *CalcIndexCoverage(bool match_gaps, bool ignore_strand, long int max_label_value);

  GenomicRegionSet *QuerySet;
  GenomicRegionSet *IndexSet;
  GenomicRegion *current_qreg;
  GenomicRegion *current_ireg;

 protected:
  // the two reductions on the device.  Which reference algorithm's input rules apply is read off the object's type: a
  // SortedGenomicRegionSetOverlaps (or a class derived from it) has the merge's, anything else the bin index's -- the class adds no
  // virtual member to the reference's five, so a subclass written against gtools/genomic_intervals.h compiles and overrides unchanged
  unsigned long int *Reduce(bool coverage, bool match_gaps, bool ignore_strand, long int max_label_value);
};

// genomic_intervals.h:2607 -- query rules of the bin-index algorithm (any order; a query with
// stop <= 0 or start > stop on a known chromosome is an error, :5740-5741)
class UnsortedGenomicRegionSetOverlaps : public GenomicRegionSetOverlaps
{
 public:
  UnsortedGenomicRegionSetOverlaps(GenomicRegionSet *QuerySet, GenomicRegionSet *IndexSet, const char *bin_bits = NULL);
  ~UnsortedGenomicRegionSetOverlaps();
  GenomicRegion *GetQuery();
  GenomicRegion *NextQuery();
  // candidates whose envelope meets the current query's, in the reference's order: bin level by bin level, bins in ascending order,
  // inside a bin the region inserted last first (genomic_intervals.cpp:5717-5764).  A host-side bin index, built at the first call.
  GenomicRegion *GetMatch();
  GenomicRegion *NextMatch();
  bool Done();
 private:
  struct MatchIndex;
  MatchIndex *match;
  std::string bin_bits_;
};

// genomic_intervals.h:2733 -- rules of the sorted merge (both sets sorted by chromosome[, strand],
// start; violations are errors, :5868, :5894)
class SortedGenomicRegionSetOverlaps : public GenomicRegionSetOverlaps
{
 public:
  SortedGenomicRegionSetOverlaps(GenomicRegionSet *QuerySet, GenomicRegionSet *IndexSet, bool sorted_by_strand);
  ~SortedGenomicRegionSetOverlaps();
  GenomicRegion *GetQuery();
  GenomicRegion *NextQuery();
  // the merge's buffer of index regions around the current query (LoadIndexBuffer :5844-5873, GetMatch :5903-5918), on the host;
  // the index set must be loaded in memory
  GenomicRegion *GetMatch();
  GenomicRegion *NextMatch();
  bool Done();
  bool sorted_by_strand;
 private:
  void LoadIndexBuffer();
  std::vector<long int> buffer_;                                   // ordinals of the buffered index regions
  size_t buffer_at_;
  long int index_at_;                                              // next index region to pull
  bool have_union_; std::string union_chrom_; char union_strand_; long int union_start_, union_stop_;
};

// ---- GenomicRegionSetIndex (genomic_intervals.h:2505) -- only what the scanners' reference filter uses ---
// "does any region of the (in-memory) set overlap this interval": per chromosome [and strand] the regions
// sorted by start with a running maximum of their stops.  The bin levels of the reference (bin_bits) are an
// implementation detail of its search and have no observable effect; NextOverlap enumeration is outside the path.
class GenomicRegionSetIndex
{
 public:
  GenomicRegionSetIndex(GenomicRegionSet *regSet, const char *bin_bits = NULL);
  ~GenomicRegionSetIndex();
  GenomicRegion *GetOverlap(GenomicInterval *i, bool match_gaps, bool ignore_strand);   // an overlapping region or NULL (:5687-5711)
  GenomicRegionSet *regSet;
 private:
  struct Impl;
  Impl *impl;
};

// ---- scanners (genomic_intervals.h:2196, 2274, 2330) ---------------------------------------------------
class GenomicRegionSetScanner
{
 public:
  GenomicRegionSetScanner(GenomicRegionSet *R, StringLIntMap *bounds, long int win_step, long int win_size, long int max_label_value,
                          bool ignore_strand, char preprocess);
  virtual ~GenomicRegionSetScanner();

  // the reference's five pure virtuals (genomic_intervals.h:2213-2217): a scanner written against it derives from this class and
  // implements them.  The two scanners of this package share one implementation, which lives in the bodies of these members (a
  // pure virtual may have one) and which their overrides call.
  virtual void PrintInterval(FILE *out_file = stdout) = 0;        // "chr strand start stop" of the current window
  virtual GenomicInterval *GetInterval() = 0;                     // heap object owned by the caller
  virtual long int Next() = 0;                                    // next window's value, -1 at the end
  virtual long int Next(GenomicRegionSet *Ref) = 0;               // ... of the next window that overlaps a region of the sorted set (:4960-4977, :5144-5163)
  virtual long int Next(GenomicRegionSetIndex *index) = 0;        // ... of the indexed set (:4982-4991, :5168-5178)
  // MI355X path: sum of GetLabelValue(max_label_value) over every region of the input, collected by the same pass that
  // fills the windows -- what the reference gets from a separate read of the file (CountGenomicRegions, :6206-6214)
  long int TotalLabelValue();
  // an input error is waiting for the Next() call that meets it (sorted scanners): TotalLabelValue() then covers the lines in front of it only
  bool InputErrorPending() { if (!computed) Compute(false); return halt_set; }
  // MI355X path: every remaining window with a value >= min_value as "value\tchr strand start stop\n" -- what a caller's
  // Next() / PrintInterval() loop prints (genomic_scans.cpp:421-428), formatted in bulk (three stdio calls per window are half a
  // second for the three million windows of a genome at -w 1000)
  void PrintRemaining(FILE *out_file, long int min_value);

  GenomicRegionSet *R;
  StringLIntMap *bounds;
  long int win_step, win_size, max_label_value, n_win_combine;
  bool ignore_strand;
  char preprocess;

 protected:
  void Compute(bool sorted_rules);                                // runs the GPU scan over the whole input
  void ComputeMappable(const std::vector<int32_t> &class_len, const std::vector<int64_t> &class_off);   // ... for the sorted scanner's operator 'p'
  std::vector<std::string> chrom_names;                           // bounds in std::map (strcmp) order
  std::vector<long int> n_windows;                                // per (chromosome, strand) block, iteration order
  std::vector<long long> block_offset;
  std::vector<unsigned long long> values;
  size_t cur_block;
  long int cur_win;                                               // 1-based inside the block
  bool computed;
  long int total_label_value;
  // the sorted scanner streams: an error in its input is met when the region in front of the offending line is consumed, with the
  // windows before that point already handed out (genomic_intervals.cpp:4928-4957).  Compute() scans the regions in front of the
  // line and notes where the walk stops: all windows of the blocks before halt_block, halt_win windows of that block, then the error.
  bool halt_set; size_t halt_block; long int halt_win; long int halt_line; bool halt_no_prefix; std::string halt_msg;
  void RaiseHalt();
};

class SortedGenomicRegionSetScanner : public GenomicRegionSetScanner
{
 public:
  SortedGenomicRegionSetScanner(GenomicRegionSet *R, StringLIntMap *bounds, long int win_step, long int win_size, long int max_label_value,
                                bool ignore_strand, char preprocess);
  virtual ~SortedGenomicRegionSetScanner();
  virtual void PrintInterval(FILE *out_file = stdout);
  virtual GenomicInterval *GetInterval();
  virtual long int Next();
  virtual long int Next(GenomicRegionSet *Ref);
  virtual long int Next(GenomicRegionSetIndex *index);
};

class UnsortedGenomicRegionSetScanner : public GenomicRegionSetScanner
{
 public:
  UnsortedGenomicRegionSetScanner(GenomicRegionSet *R, StringLIntMap *bounds, long int win_step, long int win_size, long int max_label_value,
                                  bool ignore_strand, char preprocess);
  virtual ~UnsortedGenomicRegionSetScanner();
  virtual void PrintInterval(FILE *out_file = stdout);
  virtual GenomicInterval *GetInterval();
  virtual long int Next();
  virtual long int Next(GenomicRegionSet *Ref);
  virtual long int Next(GenomicRegionSetIndex *index);
};

void GtxSetDevices(int n_gpus);                                // MI355X path: GPUs the reductions are spread over (--ngpu; not in the reference)
void GtxMark(const char *what);                                  // GTX_TIMING=1: wall-clock mark on stderr (not in the reference)
void GtxFinish(int code);                                        // flush and leave without the teardown (see genomic_intervals.cpp)

long int CountGenomicRegions(char *reg_file, long int max_label_value);   // a host pass over the file by the unsorted reader's rules (genomic_intervals.cpp:6206-6214)
unsigned long int CalcRegSize(char *reg_file);                   // sum of the sizes of a file's regions, gaps left out (genomic_intervals.cpp:6032-6040)
unsigned long int CalcBoundSize(StringLIntMap *bounds);          // sum of the chromosome lengths (genomic_intervals.cpp:6021-6026)

// chromosome -> length from a genome region file (genomic_intervals.cpp:5997-6015)
StringLIntMap *ReadBounds(char *genome_reg_file, bool verbose = false);

#endif
