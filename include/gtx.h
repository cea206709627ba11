/*
 * gtx.h -- C ABI of the MI355X interval-overlap engine (libgtx.so).
 *
 * This is the drop-in boundary for the GenomicTools hot path
 *     genomic_overlaps count / rpkm      and      genomic_scans counts.
 * The reference has no FFI layer of its own: its boundary is the C++ class API of
 * gtools/genomic_intervals.h, called by in-tree main()s.  Each entry point below therefore
 * cites the reference method whose work it replaces; the C++ classes with the reference's
 * own names (GenomicRegionSetOverlaps, ...Scanner; see csrc/genomic_intervals.h of the
 * package) are thin shims over these calls, and INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; status-code returns: 0 = ok, <0 = error
 *     (gtx_last_error() has the text).  No exceptions cross the boundary.
 *   - regions are packed int32 triples (class_id, start, end), 1-based inclusive coordinates,
 *     exactly what GenomicRegionBED::Read produces from a BED3..BED6 line
 *     (genomic_intervals.cpp:2157-2172: start = atol(col2)+1, stop = atol(col3)).
 *     Coordinates must lie in (-2^31+2, 2^31-2) (the kernels keep +-inf sentinels beyond them).
 *   - class_id = rank of the chromosome name in strcmp order (genomic_intervals.cpp:1227), so
 *     id order == the sort order -S expects; for strand-aware runs the caller folds the strand
 *     into the id (any injective mapping works; (strand, chrom) major order keeps a
 *     position-sorted stream class-sorted).  Two regions can overlap only inside one class
 *     (genomic_intervals.cpp:624-630).  Reads whose class has no reference region -- or lies
 *     outside [0, n_classes) -- match nothing, as an unknown chromosome does in the reference
 *     (genomic_intervals.cpp:5719-5720).
 *   - one gtx_ctx per GPU and per caller thread (the reference is single-threaded and
 *     non-reentrant, genomic_intervals.cpp:5732); multi-GPU = one process per GPU, each with
 *     its own context, partial count vectors summed by the caller (RCCL all-reduce).
 *   - the library never falls back to a CPU implementation: without a usable HIP device
 *     gtx_create() fails.
 */
#ifndef GTX_H
#define GTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gtx_ctx gtx_ctx;

/* error codes */
#define GTX_OK              0
#define GTX_E_ARG          -1   /* bad argument                                         */
#define GTX_E_HIP          -2   /* HIP runtime error (text in gtx_last_error)           */
#define GTX_E_STATE        -3   /* call order (e.g. count before set_refs)              */
#define GTX_E_RANGE        -4   /* coordinate >= 2^31-1 or negative class id in refs    */

/* flags for gtx_count* / gtx_scan* */
#define GTX_READS_SORTED    1u  /* hint: reads are sorted by (class, start) -- the streaming
                                   wave-ballot kernel is used; it is exact for ANY order, only
                                   slower on unsorted input.  Without the hint the order-agnostic
                                   path runs (bucket partition + LDS counting for large batches,
                                   per-read binary search for small ones): ~13x slower than the
                                   streaming kernel on sorted reads, ~3x faster than it on
                                   shuffled ones.  Position-sorted reads whose classes interleave
                                   (strand-aware ids on a (chrom, start)-sorted stream) are best
                                   passed WITHOUT the hint, or grouped by class first.          */
#define GTX_ZERO_LENGTH_OK   4u  /* sorted-merge semantics for degenerate reads: a zero-length read
                                   (start == end+1, BED start == end) IS counted, as
                                   SortedGenomicRegionSetOverlaps does (genomic_intervals.cpp:5903-5918
                                   has no start<=stop check); only start > end+1 is reported as
                                   degenerate.  On a reference set given with GTX_REFS_KEEP_ZERO_LENGTH the
                                   call has the merge's semantics in full: an inverted read or region
                                   (start > end+1) matches by the two comparisons of CalcDirection
                                   (:1225-1236) -- q.start <= r.stop and q.stop >= r.start -- like any other;
                                   such reads are still counted in n_degenerate, for information.
                                   Scans: GTX_ZERO_LENGTH_OK selects the sorted scanner's
                                   rule (no validity test, genomic_intervals.cpp:4933-4947).       */
#define GTX_GAPS_FORMULA     8u  /* gtx_coverage*: the overlap of a matching pair is min(ends) - max(starts) + 1 without
                                   clamping at 0 -- what CalcIndexCoverage computes under match_gaps
                                   (genomic_intervals.cpp:5278); it differs from the clamped CalcOverlap (:427-432)
                                   only for pairs with an inverted interval, so it matters only together with
                                   GTX_ZERO_LENGTH_OK on a GTX_REFS_KEEP_ZERO_LENGTH reference set            */
#define GTX_READS_UNSORTED  16u  /* gtx_coverage*, gtx_scan*: hint that the reads are in no particular order -- the partition path
                                   (buckets in LDS) instead of the streaming kernel, which is exact for any order but
                                   slow for shuffled reads.  The host-buffer calls sample their input and decide by
                                   themselves; gtx_coverage_device / gtx_scan_device take the caller's word.  Results do not depend on it. */
#define GTX_CHECK_SORTED    2u  /* also verify the order the sorted merge requires
                                   (SortedGenomicRegionSetOverlaps::NextQuery,
                                   genomic_intervals.cpp:5889-5898) and report the first
                                   violation in gtx_count_info.first_unsorted.  Only the streaming
                                   kernel looks at the order of the reads, so this flag selects it
                                   whether or not GTX_READS_SORTED is given (counts stay exact for
                                   any order; unsorted input is only slower that way)          */

/* what one count/scan call observed; valid after the call's stream has been synchronised
 * (gtx_count/gtx_scan synchronise themselves; after a *_device call use gtx_sync). */
typedef struct {
  int64_t first_unsorted;    /* index of first read that sorts before its predecessor, -1 if none
                                (only with GTX_CHECK_SORTED)                                   */
  int64_t n_no_class;        /* reads whose class is outside [0,n_classes): ignored            */
  int64_t n_degenerate;      /* reads with start > end: NOT counted by the device path; the
                                caller applies the reference's rule for them (error exit in the
                                unsorted algorithm, genomic_intervals.cpp:5740-5741)           */
  int64_t first_degenerate;  /* index of the first such read, -1 if none                       */
  int64_t n_unplaced;        /* sorted-merge semantics only: inverted reads (start > end+1) beyond the
                                capacity of the side buffer (2^20) that could NOT be matched -- nonzero
                                means the result is incomplete and the caller must treat it as an error */
} gtx_count_info;

/* ---- context ---------------------------------------------------------------------------- */

/* Binds a context to one HIP device.  Fails (NULL) when no HIP device is usable. */
gtx_ctx    *gtx_create(int device_id);
void        gtx_destroy(gtx_ctx *ctx);
/* Text of the last error on this context (or of a failed gtx_create when ctx == NULL). */
const char *gtx_last_error(const gtx_ctx *ctx);
/* All later work is enqueued on this hipStream_t (NULL = the default stream). */
int         gtx_set_stream(gtx_ctx *ctx, void *hip_stream);
/* Waits for everything the context has enqueued (its stream and its host->device copy stream). */
int         gtx_sync(gtx_ctx *ctx);

/* Page-locked host memory for the host-buffer entry points (gtx_count[_add], gtx_coverage[_add], gtx_scan).
 * Those calls stream their input through the device in batches, the host->device copy of one batch under the
 * kernels of the one before.  Ordinary (pageable) buffers are first moved into page-locked staging slots by a few
 * host threads and are free again when the call returns.  Buffers from gtx_host_alloc are read by the DMA engine
 * directly -- no staging copy -- and the call returns with the copy still in flight: such a buffer must stay
 * untouched until the context's NEXT host-buffer call, gtx_*_end or gtx_sync has returned (a producer that fills
 * two of them in turn -- what the reference's streaming GenomicRegionSet::Next loop, genomic_intervals.cpp:3855-3861,
 * becomes here -- never waits). */
void       *gtx_host_alloc(gtx_ctx *ctx, size_t bytes);
void        gtx_host_free(gtx_ctx *ctx, void *p);

/* ---- index side ------------------------------------------------------------------------- */

/* Replaces the index construction of UnsortedGenomicRegionSetOverlaps
 * (genomic_intervals.cpp:5593-5675) / the IRegBuffer of SortedGenomicRegionSetOverlaps
 * (:5844-5873): takes the M single-interval reference regions in FILE order and builds the
 * device-resident rank structure (two boundary arrays sorted by (class, coordinate)).
 * Regions with start > end or end <= 0 stay in the numbering but never match, as at :5659.
 * n_classes <= 0 means "max class id + 1". */
int gtx_set_refs(gtx_ctx *ctx, const int32_t *ref_triples, int64_t n_refs, int32_t n_classes);
/* flags for gtx_set_refs_ex */
#define GTX_REFS_KEEP_ZERO_LENGTH 1u   /* sorted-merge semantics: the merge never validates index
                                          regions, so zero-length ones (start == end+1), ones with
                                          end <= 0 and inverted ones (start > end+1) all take part
                                          (the last only in calls made with GTX_ZERO_LENGTH_OK)      */
/* A region with class id -1 is a placeholder: it keeps its place in the numbering and never matches. */
int gtx_set_refs_ex(gtx_ctx *ctx, const int32_t *ref_triples, int64_t n_refs, int32_t n_classes, uint32_t flags);
int64_t gtx_n_refs(const gtx_ctx *ctx);

/* ---- genomic_overlaps count ------------------------------------------------------------- */

/* Replaces GenomicRegionSetOverlaps::CountIndexOverlaps (genomic_intervals.cpp:5304-5317,
 * decl genomic_intervals.h:2471) for single-interval regions with match_gaps = false:
 *     hits[k] = sum over reads q of w_q * [q overlaps reference k]
 * in reference FILE order, 64-bit unsigned wrap-around arithmetic like the reference's
 * `unsigned long`.  weights == NULL means w_q = 1 (--max-label-value <= 1,
 * genomic_intervals.cpp:1081-1085); otherwise weights[q] is the already clamped label value.
 * Host buffers in, host buffer out; copies + kernels + sync inside. */
int gtx_count(gtx_ctx *ctx, const int32_t *read_triples, const int32_t *weights, int64_t n_reads,
              uint32_t flags, uint64_t *hits_out /* n_refs */, gtx_count_info *info /* may be NULL */);

/* Streaming form of gtx_count for a query set that is never held in memory (the reference's
 * load_in_memory=false mode, genomic_intervals.cpp:3855-3861): begin, any number of add calls with
 * consecutive batches of the stream, end.  gtx_count(...) == begin + add + end.  Indices in `info`
 * are positions in the whole stream. */
int gtx_count_begin(gtx_ctx *ctx);
int gtx_count_add(gtx_ctx *ctx, const int32_t *read_triples, const int32_t *weights, int64_t n_reads, uint32_t flags);
int gtx_count_end(gtx_ctx *ctx, uint64_t *hits_out /* n_refs */, gtx_count_info *info /* may be NULL */);

/* Same with reads (and weights) already resident in this device's HBM and the count vector
 * left in HBM (d_hits_out: uint64[n_refs], overwritten).  Asynchronous on the context's stream.
 * This is the entry the benchmark times and the one a multi-GPU caller reduces from. */
int gtx_count_device(gtx_ctx *ctx, const void *d_read_triples, const void *d_weights, int64_t n_reads,
                     uint32_t flags, void *d_hits_out);
/* Result of the most recent *_device call (synchronises the stream). */
int gtx_last_info(gtx_ctx *ctx, gtx_count_info *info);

/* ---- count over multi-interval (BED12) regions, match_gaps = false ------------------------ */

/* CountIndexOverlaps counts a query once for an index region when GetMatch / NextMatch deliver the pair -- their envelopes
 * (first interval's start .. last interval's stop) overlap, genomic_intervals.cpp:5752 -- and GenomicRegion::OverlapsWith holds:
 * SOME interval of the one overlaps SOME interval of the other (:1167-1172, :5226-5232).  With match_gaps = true the envelope
 * alone decides, and a caller simply hands over envelopes as triples.  Without it:
 *
 * gtx_set_ref_blocks declares the intervals of the index regions given to gtx_set_refs[_ex] (whose triples must be the
 * envelopes): region k's intervals are blocks[2 * first[k]] .. blocks[2 * first[k+1] - 1] as (start, stop) pairs, sorted and
 * disjoint as IsCompatibleSortedAndNonoverlapping demands (:1153-1161; GTX_E_RANGE when starts or stops decrease).  From then
 * on every count call of the context (gtx_count*, gtx_count_device, gtx_count_add_text) takes the reads that lie in a gap of a
 * multi-interval region off that region's count again.  first == NULL: back to envelopes only.  gtx_set_refs clears it.
 *
 * gtx_count_add_regions adds multi-interval QUERIES to an open count stream (gtx_count_begin): env_triples are their
 * (class, envelope start, envelope stop), first / blocks their intervals as above; each is counted once for every index region
 * one of its intervals overlaps.  They never enter the streaming kernel (a side channel: one lane per query, candidates from
 * the index regions' envelopes in the order of their starts). */
int gtx_set_ref_blocks(gtx_ctx *ctx, const int64_t *first /* n_refs + 1, or NULL */, const int32_t *blocks);
int gtx_count_add_regions(gtx_ctx *ctx, const int32_t *env_triples, const int32_t *weights /* may be NULL */,
                          const int64_t *first /* n + 1 */, const int32_t *blocks, int64_t n);

/* ---- genomic_overlaps coverage / density ------------------------------------------------- */

/* Replaces GenomicRegionSetOverlaps::CalcIndexCoverage (genomic_intervals.cpp:5269-5285, decl
 * genomic_intervals.h:2455) for single-interval regions:
 *     cov[k] = sum over reads q overlapping reference k of w_q * (min(e_q,E_k) - max(s_q,S_k) + 1)
 * in FILE order, 64-bit wrap-around arithmetic.  Same calling pattern as the count entry points; the
 * reference set is the one given to gtx_set_refs[_ex].  Reads and regions of zero length contribute 0
 * (both CalcOverlap and the -gaps formula yield 0 for them). */
int gtx_coverage_begin(gtx_ctx *ctx);
int gtx_coverage_add(gtx_ctx *ctx, const int32_t *read_triples, const int32_t *weights, int64_t n_reads, uint32_t flags);
int gtx_coverage_end(gtx_ctx *ctx, uint64_t *cov_out /* n_refs */, gtx_count_info *info /* may be NULL */);
int gtx_coverage(gtx_ctx *ctx, const int32_t *read_triples, const int32_t *weights, int64_t n_reads,
                 uint32_t flags, uint64_t *cov_out /* n_refs */, gtx_count_info *info /* may be NULL */);
int gtx_coverage_device(gtx_ctx *ctx, const void *d_read_triples, const void *d_weights, int64_t n_reads,
                        uint32_t flags, void *d_cov_out);

/* ---- genomic_scans counts --------------------------------------------------------------- */

/* Number of sliding windows the scanners report for a chromosome of length `len`
 * (genomic_intervals.cpp:5025, :5061-5064): n = len/win_step micro-windows, c = win_size/win_step,
 * max(0, n - c + 1) windows; window k (0-based) is [win_step*k + 1, win_step*k + win_size] (:5111). */
int64_t gtx_scan_n_windows(int64_t len, int64_t win_step, int64_t win_size);

/* Replaces the UnsortedGenomicRegionSetScanner constructor (genomic_intervals.cpp:5019-5080)
 * -- equivalently the window sums SortedGenomicRegionSetScanner::Next (:4928-4957) yields:
 * per class c, micro-window histogram v[(pos-1)/win_step] += w_q for reads with start <= end,
 * end > 0, pos >= 1 inside the first class_len[c]/win_step micro-windows
 * (pos = start for preprocess '1', start + (end-start)/2 for 'c'), then sums of
 * win_size/win_step consecutive micro-windows.  windows_out is the concatenation over classes,
 * class c at class_offsets[c], gtx_scan_n_windows(class_len[c],..) entries each. */
int gtx_scan(gtx_ctx *ctx, const int32_t *read_triples, const int32_t *weights, int64_t n_reads,
             const int32_t *class_len, int32_t n_classes, int32_t win_step, int32_t win_size, char preprocess,
             uint32_t flags, uint64_t *windows_out, const int64_t *class_offsets);

/* Device-resident form: d_windows_out uint64[total windows] in HBM; asynchronous. */
int gtx_scan_device(gtx_ctx *ctx, const void *d_read_triples, const void *d_weights, int64_t n_reads,
                    const int32_t *class_len, int32_t n_classes, int32_t win_step, int32_t win_size, char preprocess,
                    uint32_t flags, void *d_windows_out, const int64_t *class_offsets);

/* ---- several GPUs of one node ------------------------------------------------------------ */

/* Two regions overlap only inside one class (genomic_intervals.cpp:624-630), so a class is an independent unit of work.
 * A group has one member (context) per device; classes are dealt to the members (longest-processing-time packing of a
 * per-class load), every member holds the whole reference set and counts the reads of ITS classes.  What follows a member's
 * streaming kernel shrinks with its share: it finalizes only the histogram tiles of its classes and only its regions, into
 * its piece of a COMPACT vector (regions ordered by owner of their class, then position in the file: gtx_group_plan), the
 * pieces travel to member 0 over xGMI -- one grouped RCCL send / receive per member: the reduce(sum) of the per-region
 * vector with its addends known to be disjoint by class -- and member 0 puts them into file order.  genomic_scans windows
 * are per class too: a member scans its classes into a packed vector of its own, the per-class pieces travel the same way.
 * A group is ONE process and one caller thread driving all members (gtx_group_create; calls enqueue and return), or one
 * process per member (gtx_group_create_rank).  What the reference's single loop over queries
 * (genomic_intervals.cpp:5304-5317) becomes on a node. */
typedef struct gtx_group gtx_group;

/* device_ids == NULL: devices 0 .. n_devices-1.  NULL on failure (gtx_group_last_error(NULL) has the text).  librccl is
 * loaded at run time, for groups of more than one device only.  RCCL prints a version banner on stdout when a communicator
 * comes up: descriptor 1 points at stderr for the duration of that step, so a caller with other threads writing to stdout
 * keeps them from flushing it until gtx_group_create / gtx_group_create_rank has returned. */
gtx_group  *gtx_group_create(int n_devices, const int *device_ids);
/* One process per member: rank 0 obtains an id (GTX_GROUP_ID_BYTES bytes, an ncclUniqueId) and hands it to the others by
 * whatever launched them; every process then creates its member.  The object holds the local member only; the calls that
 * take per-member arrays (gtx_group_count_device, gtx_group_scan_device) take arrays of ONE entry, results arrive on rank
 * 0, and every rank makes the same calls in the same order.  The host-buffer calls need a group that holds all its members. */
#define GTX_GROUP_ID_BYTES 128
int         gtx_group_unique_id(void *id_out /* GTX_GROUP_ID_BYTES */);
gtx_group  *gtx_group_create_rank(int device_id, int rank, int world_size, const void *unique_id /* may be NULL for a world of one */);
void        gtx_group_destroy(gtx_group *g);
int         gtx_group_size(const gtx_group *g);     /* members, all processes together */
int         gtx_group_rank(const gtx_group *g);     /* -1: the group holds all its members */
gtx_ctx    *gtx_group_ctx(gtx_group *g, int member);   /* NULL for another process's member */
const char *gtx_group_last_error(const gtx_group *g);

/* class -> member by LPT packing of class_load (e.g. reads per class, or chromosome lengths); owner_out (n_classes, may be
 * NULL) receives the assignment.  Without this call gtx_group_set_refs assigns by the span of each class's reference
 * regions and the scans by class_len.  gtx_lpt_assign is the packing itself (no group, no GPU; deterministic, so the ranks
 * of a multi-process group all compute the same). */
int  gtx_group_assign(gtx_group *g, const int64_t *class_load, int32_t n_classes, int32_t *owner_out);
void gtx_lpt_assign(const int64_t *class_load, int32_t n_classes, int n_members, int32_t *owner_out);
/* The compact order of a group's result (no group, no GPU): ref_class[k * stride] = class of region k (stride 3 reads the
 * class column of packed triples); perm[j] = file position of the region at compact position j, member m's piece =
 * positions seg_offset[m] .. seg_offset[m+1]; regions of no class (a placeholder, an id beyond the assignment) are member 0's. */
int  gtx_group_plan(const int32_t *ref_class, int64_t stride, int64_t n_refs, const int32_t *owner, int32_t n_classes, int n_members,
                    int64_t *seg_offset /* n_members + 1 */, int32_t *perm /* n_refs */);

/* gtx_set_refs_ex on every local member. */
int gtx_group_set_refs(gtx_group *g, const int32_t *ref_triples, int64_t n_refs, int32_t n_classes, uint32_t flags);

/* The reads of every member already resident in ITS device's HBM (member m: the reads of the classes it owns -- reads of
 * other classes would be counted into tiles nobody finalizes): d_reads / d_weights / n_reads are indexed by local member
 * (d_weights may be NULL).  Per member the streaming kernel and the finalize step of its share, the pieces to member 0, the
 * result in file order in d_hits (n_refs uint64 on member 0's device; ignored on other ranks).  Everything is enqueued on the
 * members' streams (gtx_set_stream of gtx_group_ctx), except that the pieces travel -- and member 0 writes d_hits -- on a
 * stream of the group's own behind each member's finalize step, into one of two compact vectors in turn: the kernels of the
 * next call run under the exchange of this one.  d_hits is complete after gtx_group_sync, or, for work enqueued afterwards on
 * the members' streams, behind gtx_group_wait_result (a device-side wait, the host does not block); until then the caller
 * leaves d_hits alone (alternate two vectors to keep calls in flight).  GTX_CHECK_SORTED is ignored, GTX_ZERO_LENGTH_OK
 * refused (the sorted merge's host-side corrections live in the host-buffer calls).
 * gtx_group_last_info: the sums over the local members (first_unsorted / first_degenerate are not tracked: -1). */
int gtx_group_count_device(gtx_group *g, const void *const *d_reads, const void *const *d_weights, const int64_t *n_reads,
                           uint32_t flags, void *d_hits);
int gtx_group_scan_device(gtx_group *g, const void *const *d_reads, const void *const *d_weights, const int64_t *n_reads,
                          const int32_t *class_len, int32_t n_classes, int32_t win_step, int32_t win_size, char preprocess,
                          uint32_t flags, void *d_windows /* member 0's device, layout class_offsets */, const int64_t *class_offsets);
int gtx_group_sync(gtx_group *g);
int gtx_group_wait_result(gtx_group *g);
int gtx_group_last_info(gtx_group *g, gtx_count_info *info);

/* The streaming count / coverage calls of a single context, on the group: every read goes to the owner of its class (reads
 * of no known class to member 0).  Sorted input is cut into a few contiguous runs per batch; interleaved input is
 * partitioned on the host.  Page-locked batches (gtx_host_alloc) follow the single-context rule: a batch must stay
 * untouched until the group's next host-buffer call has returned (the call waits for every member's copy of the previous
 * batch, also of members that get nothing of the new one).  count on a plain reference set ends with the pieces of the
 * compact vector as above; coverage, and count on a GTX_REFS_KEEP_ZERO_LENGTH set (whose inverted intervals are matched
 * into the full vector), with one ncclReduce(sum) of the members' full vectors.  info: the sums over the members;
 * first_unsorted / first_degenerate are not tracked (-1), and GTX_CHECK_SORTED is ignored. */
int gtx_group_count_begin(gtx_group *g);
int gtx_group_count_add(gtx_group *g, const int32_t *read_triples, const int32_t *weights, int64_t n_reads, uint32_t flags);
int gtx_group_count_end(gtx_group *g, uint64_t *hits_out /* n_refs */, gtx_count_info *info /* may be NULL */);
/* gtx_set_ref_blocks on every member / gtx_count_add_regions on the members in turn (every member holds the whole reference set) */
int gtx_group_set_ref_blocks(gtx_group *g, const int64_t *first /* n_refs + 1, or NULL */, const int32_t *blocks);
int gtx_group_count_add_regions(gtx_group *g, const int32_t *env_triples, const int32_t *weights /* may be NULL */,
                                const int64_t *first /* n + 1 */, const int32_t *blocks, int64_t n);
int gtx_group_coverage_begin(gtx_group *g);
int gtx_group_coverage_add(gtx_group *g, const int32_t *read_triples, const int32_t *weights, int64_t n_reads, uint32_t flags);
int gtx_group_coverage_end(gtx_group *g, uint64_t *cov_out /* n_refs */, gtx_count_info *info /* may be NULL */);
/* gtx_scan on the group (arguments as gtx_scan; with more than one member only the classes' own ranges of windows_out are
 * written, whatever class_offsets leaves between them is not touched). */
int gtx_group_scan(gtx_group *g, const int32_t *read_triples, const int32_t *weights, int64_t n_reads,
                   const int32_t *class_len, int32_t n_classes, int32_t win_step, int32_t win_size, char preprocess,
                   uint32_t flags, uint64_t *windows_out, const int64_t *class_offsets);
/* reads each member received in the last (or open) group call: the load balance actually achieved */
int gtx_group_member_reads(const gtx_group *g, int64_t *reads_out /* gtx_group_size */);

/* ---- region text tokenised on the device ---------------------------------------------------
 * Ingest (GenomicRegionBED::Read genomic_intervals.cpp:2157-2182, tokenizer core.cpp:577-625, FileBufferText::Next
 * core.cpp:241-259) for the query stream of the two reductions: a block of COMPLETE lines of a BED file ('\n' after every line,
 * n_lines of them) is copied to the device as text and cut into packed triples there, then counted like a gtx_count_add /
 * gtx_coverage_add batch of an open call.  The device recognises only the plain case -- tab-separated, decimal columns 2 and 3,
 * strand column one of + - . 1 -1, not 12 columns, in the order the sorted merge requires, reads the mode accepts; the reference's
 * reading of everything else (blanks as separators, signs, '\r', BED12, its error messages with their line numbers) stays with
 * the host-side packer: a block with ANY other line is not counted at all and gtx_text_result says so -- the caller packs that
 * block itself and adds it with gtx_count_add.  chrom_names: the chromosome of class c (classes beyond n_chrom are the '-' strand
 * when strand_aware); a line of another chromosome is dropped, as the reference's index lookup does (:5719-5720).
 * sorted_rules: the sorted merge's rules (order check against the line before -- prev_* describe the line before the block --, no
 * validation of the interval), else the bin index's (stop <= 0 or start > stop is the reference's error: not plain).
 * max_label_value > 1: weights = min(max, atol(column 4)) (genomic_intervals.cpp:1081-1085). */
typedef struct gtx_text_rules {
  const char *const *chrom_names; int32_t n_chrom;
  int32_t strand_aware, sorted_rules, sorted_by_strand;
  int64_t max_label_value;
  int32_t have_prev; const char *prev_chrom; int32_t prev_strand /* '+' | '-' */; int64_t prev_start;
} gtx_text_rules;
/* text: host memory (page-locked memory is read by the DMA engine directly and must stay untouched until gtx_text_result of the
 * ticket has returned).  flags as gtx_count_add / gtx_coverage_add.  Up to two blocks are in flight: the call waits for the block
 * before last. */
int gtx_count_add_text(gtx_ctx *ctx, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket);
int gtx_coverage_add_text(gtx_ctx *ctx, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket);
/* waits for the tokenizer of that block; *needs_host != 0: nothing of the block was counted, the caller packs and adds it */
int gtx_text_result(gtx_ctx *ctx, int ticket, int *needs_host);
/* The same in a group that holds all its members (between gtx_group_count_begin / gtx_group_coverage_begin and their _end): a block goes
 * to the members in turn, whatever the classes of its lines -- the read stream split evenly over members that each hold the whole
 * reference set (SURVEY 8(e)'s second partition) -- and a count call that took text blocks ends with the ncclReduce(sum) of the
 * members' full vectors instead of pieces.  Blocks that come back (needs_host) are packed by the caller and go through gtx_group_*_add. */
int gtx_group_count_add_text(gtx_group *g, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket);
int gtx_group_coverage_add_text(gtx_group *g, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket);
int gtx_group_text_result(gtx_group *g, int ticket, int *needs_host);

/* Regions into position order on the device: what the reference's bin/sortbed (`sort -k1,1 -k2,2n`, or
 * `-k1,1 -k6,6 -k2,2n`) and `genomic_regions gsort` (RunGlobalSort genomic_intervals.cpp:4547-4570 over BinGenomicRegions :6095-6150
 * and CompareBinnedGenomicRegions :6044-6048) are run for -- the input order the sorted merge (:5807-5937) and the sorted scanner
 * (:4928-4957) insist on.  order[i] = the input ordinal of the region that comes i-th under (class ascending, start ascending, stop
 * DESCENDING, input order): gsort's order when the caller folds the chromosome's strcmp rank -- and, sorting by strand, the strand
 * below it -- into the class.  sorted (may be NULL) receives the triples in that order.  n_reads < 2^32; a class id outside
 * [0, n_classes) is GTX_E_RANGE.  No reference set is needed.  gtx_sort_device: reads, order (uint32[n_reads]) and sorted in the
 * context's HBM; the call returns when the result is complete. */
int gtx_sort(gtx_ctx *ctx, const int32_t *read_triples, int64_t n_reads, int32_t n_classes, uint32_t *order_out, int32_t *sorted_out);
int gtx_sort_device(gtx_ctx *ctx, const void *d_reads, int64_t n_reads, int32_t n_classes, void *d_order, void *d_sorted);

/* genomic_scans counts fed as a stream (UnsortedGenomicRegionSetScanner ctor genomic_intervals.cpp:5019-5080, sorted scanner :4928-4957):
 * gtx_scan_begin fixes the geometry (arguments as gtx_scan; flags: GTX_ZERO_LENGTH_OK = the sorted scanner's rule; weighted != 0: every
 * batch brings label weights), gtx_scan_add adds packed reads from host memory (flags: GTX_READS_UNSORTED as a hint), gtx_scan_add_text a
 * block of BED text tokenised on the device (rules as for the overlap calls; lines the device does not take come back through
 * gtx_text_result and are packed by the caller), gtx_scan_end writes the windows (layout of gtx_scan) and, when label_sum is not NULL,
 * the sum of the label values of all lines of the text blocks that did NOT come back (CountGenomicRegions, :6206-6214: peaks' read
 * total).  The micro-window histogram accumulates over the batches; a caller with every read in hand uses gtx_scan / gtx_scan_device. */
int gtx_scan_begin(gtx_ctx *ctx, const int32_t *class_len, int32_t n_classes, int32_t win_step, int32_t win_size, char preprocess,
                   uint32_t flags, int weighted, const int64_t *class_offsets);
int gtx_scan_add(gtx_ctx *ctx, const int32_t *read_triples, const int32_t *weights, int64_t n_reads, uint32_t flags);
int gtx_scan_add_text(gtx_ctx *ctx, const char *text, size_t bytes, int64_t n_lines, const gtx_text_rules *rules, uint32_t flags, int *ticket);
int gtx_scan_end(gtx_ctx *ctx, uint64_t *windows_out, int64_t *label_sum);

/* ---- measurement ------------------------------------------------------------------------ */

/* on = 1: every *_device call brackets its dominant kernel and the whole call with HIP events on the
 * context's stream (three records, ~5 us of stream bubble each on this platform).
 * on = N >= 2: only every N-th call is profiled and only its dominant kernel is bracketed (two records;
 * ms_total then repeats ms_stream_kernel) -- for timing loops that should not be stretched by their own
 * instrumentation.  on = 0: off.  Resets the ring of profiled calls. */
int gtx_profile_enable(gtx_ctx *ctx, int on);
/* Elapsed ms of the last profiled call: the streaming kernel alone, and the whole enqueue
 * (memsets + stream kernel + finalize kernels).  Waits for that call to finish. */
int gtx_profile_last(gtx_ctx *ctx, float *ms_stream_kernel, float *ms_total);
/* Same for the call `back` calls before the last one (0 = last); the last 64 profiled calls
 * are kept, so a timed loop can be read back after it ends without synchronising inside it. */
int gtx_profile_read(gtx_ctx *ctx, int back, float *ms_stream_kernel, float *ms_total);
/* How many profiled calls gtx_profile_read can reach (<= 64) since the last gtx_profile_enable. */
int gtx_profile_count(gtx_ctx *ctx);

/* Library/ABI version, e.g. 100 = 1.0.0 */
int gtx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GTX_H */
