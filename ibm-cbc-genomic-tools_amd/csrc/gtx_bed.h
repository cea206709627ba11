// gtx_bed.h -- host-side BED ingest for the MI355X path: text / .gz / stdin -> packed int32 triples.
//
// Behaviour follows the reference's reader exactly where it is observable:
//   - line delivery: a line counts only if it ended in '\n' (gtools/core.cpp:241-259, 331-349);
//   - gzip sniffing by magic bytes (core.cpp:1757-1775); "browser "/"track " header lines are
//     skipped at the top of the file (genomic_intervals.cpp:3713-3720);
//   - tokenising: separator is TAB if the line has one, else SPACE; blanks before a token are
//     skipped (core.cpp:577-625, genomic_intervals.cpp:2159);
//   - BED3..BED6 fields: start = atol(col2)+1, stop = atol(col3), label = col4 or "_",
//     strand = col6 through ProcessStrand (genomic_intervals.cpp:2157-2172, 5956-5962).
// What is new is the shape of the work: blocks of complete lines are parsed by a pool of
// threads straight into (class, start, end[, weight]) arrays that go to the GPU.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <future>
#include <memory>
#include <functional>
#include <string>
#include <vector>
#include <zlib.h>

namespace gtxhost {

// ---- lines ------------------------------------------------------------------------------------
class LineSource {
 public:
  // path == NULL reads stdin.  On failure returns NULL and sets *err to the reference's message.
  static LineSource *Open(const char *path, std::string *err);
  static LineSource *FromFile(FILE *fp);        // an already open stream (the FILE* constructor of GenomicRegionSet, genomic_intervals.h:1836)
  ~LineSource();

  // Next complete line as a mutable NUL-terminated string (valid until the next call), or NULL at
  // the end.  line_no() is its 1-based number in the file.
  char *Next();
  long line_no() const { return line_no_; }

  // Bulk: the rest of the input as blocks of complete lines.  Returns the number of bytes put in
  // `block` (0 at the end); *first_line receives the file line number of the block's first line.
  // The caller counts the lines of the block and reports them with AdvanceLines().
  size_t NextBlock(std::vector<char> &block, size_t target_bytes, long *first_line);
  // Same, but a regular text file is read with several pread() calls in parallel straight into
  // `block` (no intermediate buffer); *view = block.data().  .gz and stdin go through NextBlock.
  size_t NextBlockView(std::vector<char> &block, char **view, size_t target_bytes, long *first_line);
  size_t ReadTextInto(char *dst, size_t cap, long *first_line);     // complete lines straight into the caller's memory (files by parallel preads, streams by straight reads)
  bool at_end() const { return fd_ >= 0 ? (bulk_started_ && file_pos_ >= file_len_) : eof_; }   // no complete line is left to read (a stream: known once a read has met the end)
  void AdvanceLines(long n) { line_no_ += n; }
  long regular_file_bytes() const { return fd_ >= 0 ? (long)file_len_ : -1; }   // -1: stdin, .gz, a FILE* of the caller's

 private:
  LineSource() {}
  size_t Fill();                     // read more raw bytes into buf_; 0 at EOF
  FILE *fp_ = nullptr; gzFile gz_ = nullptr; bool is_stdin_ = false;
  int raw_fd_ = -1;                  // the process's stdin: read by descriptor (no stdio buffer in between)
  size_t ReadStream(char *dst, size_t want);   // up to `want` bytes of a stream into dst (0: the end)
  int fd_ = -1; size_t file_len_ = 0, file_pos_ = 0;          // regular text file: bulk path uses pread on fd_
  bool bulk_started_ = false;
  std::vector<char> buf_;            // raw bytes [pos_, end_) not yet handed out
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
  long line_no_ = 0;
};

int WorkerThreads();                                  // GTX_PACK_THREADS, else the container's CPU quota (<= 64)
long CountNewlines(const char *b, const char *e);
// fn(t) for t in [0, n) on the library's worker threads (t = 0 on the caller); returns when all are done.  The workers live as long as
// the process.
void ParallelFor(int n, const std::function<void(int)> &fn);

// ---- one BED line ---------------------------------------------------------------------------------
struct BedFields {
  char *chrom; char *label;          // label == NULL means "_" (3-column line)
  long start, stop;                  // 1-based inclusive
  char strand;
  int n_tokens;
  // BED12 (n_tokens == 12, genomic_intervals.cpp:2174-2181): the blocks become the region's intervals
  char *block_sizes = nullptr, *block_starts = nullptr; long n_blocks = 0;
};
// the intervals of a 12-column line, as the reference computes them: interval k = [start + off_k, start + off_k + size_k - 1]
// (missing list entries read as 0, like atol of an empty token).  iv receives 2 * n_blocks longs.
void BedBlocks(const BedFields &f, std::vector<long> *iv);
enum BedStatus { BED_OK = 0, BED_TOO_FEW_TOKENS, BED_BAD_STRAND };
// Parses in place (the line is cut into tokens).  On BED_BAD_STRAND *bad points at the strand token.
BedStatus ParseBedLine(char *line, BedFields *out, char **bad);
// Number of `delim`-separated tokens as the reference counts them (core.cpp:577-593).
int CountTokensLike(const char *s, char delim);

// ---- chromosome table: names in strcmp order, id = rank ------------------------------------------
class ChromTable {
 public:
  void Add(const char *name);        // collect (duplicates ignored); call Freeze() before Find()
  void Freeze();
  int Find(const char *name) const;  // rank or -1
  int size() const { return (int)names_.size(); }
  const std::string &name(int id) const { return names_[id]; }
 private:
  std::vector<std::string> names_;
  bool frozen_ = false;
};

// ---- packed region files (.gtx): a BED file after tokenising, column by column ------------------------
// Layout (little endian): "GTXP" u32 version=1 | u32 n_chrom | u32 flags (1 = label values present) | u64 n |
// n_chrom x (u16 length, bytes) | padding to 8 | u16 chrom_idx[n] | pad | i32 start[n] (1-based) | pad | i32 stop[n] |
// pad | u8 minus[(n+7)/8] (strand bits) | pad | i32 label_value[n] (atol of column 4; present if flags & 1).
// Record i stands for line i+1 of the text it was made from (header lines not counted).  Labels themselves are not
// kept: a packed file can be the streamed TEST set of genomic_overlaps / the input of genomic_scans, not a
// reference set whose labels are printed.  Not a format of the reference: a cache that skips the text parse on re-runs.
struct GtxView {
  uint64_t n = 0; uint32_t flags = 0;
  std::vector<std::string> chrom;
  const uint16_t *chrom_idx = nullptr; const int32_t *start = nullptr, *stop = nullptr, *label = nullptr; const uint8_t *minus = nullptr;
  static bool IsGtx(const char *path);                              // regular file starting with the magic
  static GtxView *Open(const char *path, std::string *err);         // mmap; NULL + message on failure
  ~GtxView();
 private:
  void *map_ = nullptr; size_t map_len_ = 0;
};

// ---- bulk packing ---------------------------------------------------------------------------------
enum PackMode {
  PACK_OVERLAPS_UNSORTED,   // UnsortedGenomicRegionSetOverlaps query rules (genomic_intervals.cpp:5717-5764)
  PACK_OVERLAPS_SORTED,     // SortedGenomicRegionSetOverlaps query rules (:5889-5898)
  PACK_SCAN_UNSORTED,       // UnsortedGenomicRegionSetScanner (:5036-5055)
  PACK_SCAN_SORTED          // SortedGenomicRegionSetScanner (:4928-4957)
};

// Sorted merge, index side: SortedGenomicRegionSetOverlaps pulls index regions while the current query is not before them
// and checks their order only as it pulls (genomic_intervals.cpp:5851-5870), so an index set that is out of order at
// region v is an error only if some query gets the merge as far as region v-1.  When the index set has such a spot, the
// packer replays that pull loop over the sorted prefix [0, v) query by query (single-threaded: the loop is sequential
// by nature) and raises the reference's error at the query that reaches it.
struct IndexGuard {
  std::vector<const char *> chrom; std::vector<char> strand; std::vector<long> start, stop;   // regions 0 .. v-1
  bool by_strand = false;
  long p = 0;                        // regions pulled so far
  std::string msg;                   // the whole error text
};

struct PackOptions {
  PackMode mode = PACK_OVERLAPS_UNSORTED;
  const ChromTable *chroms = nullptr;
  bool strand_aware = false;         // class = strand * n_chrom + chrom rank
  bool sorted_by_strand = false;     // order check uses (chrom, strand, start) instead of (chrom, start)
  long max_label_value = 1;          // > 1: emit weights = min(max, atol(label))
  bool collect_zero_length = false;  // keep (class, start, weight) of zero-length reads (sorted mode correction)
  bool match_gaps = false;           // overlaps: multi-interval (BED12) regions are matched on their envelope -- what -gaps means
                                     // (genomic_intervals.cpp:5226, :5752); without it they are outside the path, unless ...
  bool explode_blocks = false;       // ... coverage without -gaps: CalcOverlap is a sum over ALL interval pairs (:1196-1202, :5278), so
                                     // every interval of a region goes out as a read of its own with the region's label value
  bool collect_blocks = false;       // ... count without -gaps: "some interval overlaps some interval" (:1167-1172) -- a region with several
                                     // intervals goes out on its own list (PackedBatch::m_*), after the same checks as any other region
  int threads = 0;                   // 0 = hardware concurrency
  IndexGuard *guard = nullptr;       // PACK_OVERLAPS_SORTED with an out-of-order index set (forces one thread)
  bool keep_prefix_on_error = false; // an error leaves the regions of the lines in front of it in the batch (the sorted scanner streams: what it
                                     // has counted by then is on its way out, genomic_intervals.cpp:4928-4957) and names the last of them in PackError
};

struct PackError {
  bool set = false;
  long line = 0;                     // file line number of the offending line
  bool no_prefix = false;            // message is printed as is (no "Error: Line N: " in front)
  std::string msg;                   // what the reference prints after "Error: Line N: "
  // keep_prefix_on_error: the last region in front of the offending line (order key), if there is one
  bool have_last = false; std::string last_chrom; char last_strand = '+'; long last_start = 0;
};

// Where the big batch arrays live.  By default ordinary heap memory; the GPU-side host code installs a pool of page-locked
// buffers here (genomic_intervals.cpp), so that the packer threads write the triples straight into memory the DMA engine reads --
// no staging copy, and no first-touch page faults in the middle of the parse (the pool is filled while the reference set loads).
// take(bytes) may return NULL (pool empty / request too large): the heap serves.  give(p) returns false for pointers it does not own.
struct BatchArena {
  static void *(*take)(size_t bytes);
  static bool (*give)(void *p);
};

// std::vector that does not zero-fill on resize(): the packer threads overwrite every element
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { typedef NoInitAlloc<U> other; };
  template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
  T *allocate(size_t n)
  {
    if (BatchArena::take && n * sizeof(T) >= (1u << 20)) { void *p = BatchArena::take(n * sizeof(T)); if (p) return (T *)p; }
    return (T *)::operator new(n * sizeof(T));
  }
  void deallocate(T *p, size_t) noexcept
  {
    if (BatchArena::give && BatchArena::give((void *)p)) return;
    ::operator delete((void *)p);
  }
};
typedef std::vector<int32_t, NoInitAlloc<int32_t>> RawVec;

struct PackedBatch {
  RawVec tri;                        // 3 per read
  RawVec w;                          // empty unless max_label_value > 1
  std::vector<int32_t> zero_len;     // (class, start, weight) triples, see collect_zero_length
  // collect_blocks: the multi-interval regions of the batch -- (class, envelope start, envelope stop) triples, label values,
  // interval counts, and the (start, stop) pairs of their intervals one region after the other
  std::vector<int32_t> m_tri, m_w, m_cnt, m_blocks;
  bool empty() const { return tri.empty() && m_cnt.empty(); }
  void clear() { tri.clear(); w.clear(); zero_len.clear(); m_tri.clear(); m_w.clear(); m_cnt.clear(); m_blocks.clear(); n_lines = 0; label_sum = 0; }
  int64_t n_lines = 0;               // lines consumed (regions seen), including dropped ones
  int64_t label_sum = 0;             // sum of GetLabelValue(max_label_value) over ALL regions seen (CountGenomicRegions, genomic_intervals.cpp:6206-6214)
};

// text -> packed file; lines are validated like GenomicRegionBED (token count, strand); returns false with *err set
bool WriteGtx(LineSource *src, const char *out_path, PackError *err);
// the same file from columns the caller holds (minus: one bit per record; lab may be NULL)
bool WriteGtxColumns(const char *out_path, const std::vector<std::string> &names, uint64_t n, const uint16_t *cidx, const int32_t *st, const int32_t *en,
                     const uint8_t *minus, const int32_t *lab, PackError *err);

// Packs blocks of lines from `src` until about `target_reads` reads are in `out` or the input ends.
// Returns false when the input is exhausted (out may still hold reads).  The first error in file
// order, if any, is left in *err and packing stops there.
class BedPacker {
 public:
  BedPacker(LineSource *src /* may be NULL: primed text only */, const PackOptions &opt);
  BedPacker(const GtxView *packed, const PackOptions &opt);           // the records of a packed file instead of text
  void SkipRecords(uint64_t n) { gtx_pos_ = n; }                     // packed file: start at record n
  // text to be packed before anything is read from the source: one line without its '\n' (a
  // region the caller had already pulled from the stream), or a block of '\n'-terminated lines
  void Prime(const std::string &line, long line_no);
  void PrimeBlock(const std::string &lines, long first_line);
  bool NextBatch(PackedBatch *out, size_t target_reads, PackError *err);
  // For a caller that has the text tokenised on the device (gtx_count_add_text) and keeps this packer for what is not plain:
  // the next block of complete lines as it is (false at the end of the input; its lines are counted and the source's line number
  // moves on).  have_prev / prev_*: the order key of the line before the block, for the sorted merge's order check at the seam;
  // seam_ok = false: the block's own last line could not be read as a plain BED line (the blocks after it are the host's).
  // The text stays valid until the call after next.  PackPrimedText: what Prime() / PrimeBlock() left, packed here.
  struct TextBlock { char *text = nullptr; size_t bytes = 0; long first_line = 0; int64_t n_lines = 0;
                     bool have_prev = false; std::string prev_chrom; char prev_strand = '+'; long prev_start = 0; bool seam_ok = true; };
  bool PackPrimedText(PackedBatch *out, PackError *err);
  void UseTextBuffers(char *b0, char *b1, size_t cap) { text_buf_[0] = b0; text_buf_[1] = b1; text_cap_ = cap; }   // NextTextBlock reads into these in turn
  bool NextTextBlock(TextBlock *b);
  bool SourceAtEnd() const { return !src_ || exhausted_ || src_->at_end(); }
  bool PackTextBlock(const TextBlock &b, PackedBatch *out, PackError *err);   // the host's reading of that block (appended to *out)
 private:
  char *text_buf_[2] = {nullptr, nullptr}; size_t text_cap_ = 0;
  bool seam_have_ = false, seam_ok_ = true; std::string seam_chrom_; char seam_strand_ = '+'; long seam_start_ = 0;   // NextTextBlock's view of the line before
  bool PackBlock(char *block, size_t got, long first_line, PackedBatch *out, PackError *err);
  bool PackPieces(void *pieces, long first_line, PackedBatch *out, PackError *err);
  const GtxView *gtx_ = nullptr; uint64_t gtx_pos_ = 0;
  std::vector<char> primed_; long primed_first_line_ = 0; bool primed_set_ = false;
  struct Ahead { int buf = 0; char *view = nullptr; size_t got = 0; };   // block read ahead of the parsers
  std::future<Ahead> ahead_; bool exhausted_ = false;
  std::vector<char> blocks_[2]; int next_buf_ = 0;   // two block buffers in turn: their pages are touched once, not once per block
  LineSource *src_; PackOptions opt_;
  // order check across blocks
  bool have_prev_ = false; std::string prev_chrom_; char prev_strand_ = '+'; long prev_start_ = 0;
};

}  // namespace gtxhost
