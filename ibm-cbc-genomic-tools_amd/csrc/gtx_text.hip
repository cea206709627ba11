// gtx_text.hip -- BED text to packed triples ON THE DEVICE (gtx_count_add_text / gtx_coverage_add_text of include/gtx.h).
//
// What the reference does per query line (GenomicRegionBED::Read, gtools/genomic_intervals.cpp:2157-2182; tokenizer
// core.cpp:577-625; FileBufferText::Next core.cpp:241-259): cut the line at tabs, atol() two columns, look the chromosome up,
// validate.  The host packer of this build (gtx_bed.cpp) does that at ~5 GB/s on the 16 CPUs a GPU box grants -- 0.43 s of a
// 0.9 s run over 100 M reads -- while the text itself crosses PCIe in 45 ms.  Here a block of complete lines is copied to the
// device as it is and tokenised there:
//
//   nl_count_kernel / nl_scan_kernel / nl_write_kernel   positions of the newlines (a count per 1 KB segment, an exclusive scan,
//                       the positions written in order): line j is text[nl[j-1]+1 .. nl[j])
//   text_parse_kernel   128 lines per block: their bytes staged in LDS (coalesced 16-byte loads), one thread per line --
//                       chromosome token through a hash table of the known names, two decimal columns, label value (atol rules),
//                       strand column, token count; the order check of the sorted merge against the line before; the mode's
//                       validity rules.  Output: (class, start, stop) [, weight] at index j; a line the rules drop (unknown
//                       chromosome) leaves class -1.
//   text_void_kernel    a block with ANY line outside the fast grammar -- no tab, an empty token, a sign or blank in a number, more
//                       than 10 digits, '\r', 12 columns (BED12), an order violation, a read the mode rejects -- is not counted at
//                       all (every class set to -1) and reported: the host packer redoes that block and produces the reference's
//                       result or error message with its line number.  The device path never decides an error, it only recognises
//                       the plain case.
// The counting kernels then run on the triples where they are.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "gtx.h"
#include "gtx_internal.h"
#include "gtx_text.h"

namespace gtxtext {

typedef unsigned long long u64;

static constexpr int kSeg = 1024;                  // bytes per newline-count segment (one wave: 64 lanes x 16 bytes)
static constexpr int kLines = 128;                 // lines per parse block
static constexpr int kLdsText = 40 * 1024;         // bytes of text a parse block stages (lines of 320 bytes on average fit)

__device__ __forceinline__ unsigned nl_mask16(const uint4 v)
{
  // bit k set iff byte k of the 16 is '\n'
  unsigned m = 0;
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int b = 0; b < 4; b++) m |= (((w[q] >> (8 * b)) & 0xffu) == 0x0au ? 1u : 0u) << (4 * q + b);
  return m;
}

// segment s = bytes [s * kSeg, ...): one wave, lane l owns 16 bytes
__global__ __launch_bounds__(64) void nl_count_kernel(const char *__restrict__ text, size_t bytes, unsigned *__restrict__ segCount)
{
  const size_t at = (size_t)blockIdx.x * kSeg + threadIdx.x * 16;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (at + 16 <= bytes) v = *(const uint4 *)(text + at);
  else if (at < bytes) { unsigned char t[16] = {0}; for (size_t k = 0; at + k < bytes; k++) t[k] = (unsigned char)text[at + k]; v = *(const uint4 *)t; }
  int c = __popc(nl_mask16(v));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (threadIdx.x == 0) segCount[blockIdx.x] = (unsigned)c;
}

// exclusive prefix of segCount (one block; the total behind the last entry)
__global__ __launch_bounds__(1024) void nl_scan_kernel(unsigned *__restrict__ segCount, unsigned nSeg)
{
  __shared__ unsigned wsum[16];
  const unsigned per = (nSeg + 1023) / 1024, i0 = threadIdx.x * per;
  unsigned s = 0;
  for (unsigned k = 0; k < per; k++) if (i0 + k < nSeg) s += segCount[i0 + k];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(inc, o); if (lane >= o) inc += up; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  unsigned run = inc - s;
  for (int k = 0; k < wv; k++) run += wsum[k];
  for (unsigned k = 0; k < per; k++) if (i0 + k < nSeg) { const unsigned v = segCount[i0 + k]; segCount[i0 + k] = run; run += v; }
  if (threadIdx.x == 1023) segCount[nSeg] = run;
}

__global__ __launch_bounds__(64) void nl_write_kernel(const char *__restrict__ text, size_t bytes, const unsigned *__restrict__ segOff, unsigned *__restrict__ nl, unsigned nLines)
{
  const size_t at = (size_t)blockIdx.x * kSeg + threadIdx.x * 16;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (at + 16 <= bytes) v = *(const uint4 *)(text + at);
  else if (at < bytes) { unsigned char t[16] = {0}; for (size_t k = 0; at + k < bytes; k++) t[k] = (unsigned char)text[at + k]; v = *(const uint4 *)t; }
  unsigned m = nl_mask16(v);
  const int c = __popc(m);
  int inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += up; }
  unsigned k = segOff[blockIdx.x] + (unsigned)(inc - c);
  while (m) { const int b = __ffs(m) - 1; m &= m - 1; if (k < nLines) nl[k] = (unsigned)(at + b); k++; }
}

struct ChromEntry { unsigned hash; int id; unsigned off, len; };

struct ParseArgs {
  const char *text; size_t bytes; const unsigned *nl; unsigned nLines;
  const ChromEntry *table; unsigned tableMask; const char *names; const unsigned *segTotal;
  int nChrom, strandAware, sortedRules, byStrand, weighted;
  long long maxLabel;
  int havePrev; unsigned prevNameOff, prevNameLen; int prevStrand; long long prevStart;   // the line before the block (its chromosome name in `names`' blob)
  int scanRules; unsigned long long *blockSum;   // see launch_tokenize
  int prevLost;                // there is a line before the block, but its key did not travel (a chromosome name of 4096 bytes or more)
  int *tri; int *w; int *flag;
  unsigned *blkMinus;             // strand-aware runs: '-' lines per parse block (for the grouping pass), else null
};

__device__ __forceinline__ unsigned fnv1a(const unsigned char *p, unsigned n)
{
  unsigned h = 2166136261u;
  for (unsigned i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}

// what a line says, as far as the order check of the NEXT line needs it
struct LineKey { unsigned tokOff, tokLen; int strand; long long start; bool ok; };

// parse the line [b, e) of `s` (bytes in LDS or global).  Returns false when the line is outside the fast grammar.
template <class PTR>
__device__ __forceinline__ bool parse_line(PTR s, unsigned b, unsigned e, bool wantLabel, long long &v2, long long &v3, int &nTok, unsigned &tokLen,
                                           int &strand, long long &label)
{
  unsigned p = b;
  while (p < e && s[p] != '\t') p++;
  tokLen = p - b;
  if (p >= e || tokLen == 0) return false;                        // no tab in the line (space-separated, or one token), or an empty name
  auto number = [&](long long &out, bool last) -> bool {
    unsigned q = ++p;                                             // behind the tab
    u64 v = 0;
    while (p < e && (unsigned)(s[p] - '0') < 10u) { v = v * 10 + (unsigned)(s[p] - '0'); p++; }
    const unsigned nd = p - q;
    if (nd == 0 || nd > 10) return false;
    if (p < e && s[p] != '\t') return false;                      // a sign, a blank, '\r', a letter ...
    if (!last && p >= e) return false;
    out = (long long)v;
    return true;
  };
  if (!number(v2, false)) return false;
  if (!number(v3, true)) return false;
  nTok = 3; strand = '+'; label = 0;
  while (p < e) {                                                 // s[p] == '\t': one more token
    const unsigned q = ++p;
    while (p < e && s[p] != '\t') p++;
    if (p == q) return false;                                     // an empty token
    nTok++;
    if (nTok == 4 && wantLabel) {
      // atol: optional sign, digits, the rest ignored; no digits = 0.  (A leading blank would be skipped by atol: not the plain case.)
      unsigned r = q; bool neg = false;
      if ((unsigned char)s[r] <= ' ') return false;
      if (s[r] == '-') { neg = true; r++; } else if (s[r] == '+') r++;
      u64 v = 0; unsigned nd = 0;
      while (r < p && (unsigned)(s[r] - '0') < 10u) { v = v * 10 + (unsigned)(s[r] - '0'); r++; if (++nd > 18) return false; }
      label = neg ? -(long long)v : (long long)v;
    }
    if (nTok == 6) {                                              // ProcessStrand (:5956-5962)
      const unsigned n = p - q;
      if (n == 1 && (s[q] == '+' || s[q] == '1' || s[q] == '.')) strand = '+';
      else if ((n == 1 && s[q] == '-') || (n == 2 && s[q] == '-' && s[q + 1] == '1')) strand = '-';
      else return false;
    }
  }
  if (nTok == 12) return false;                                   // BED12: the host's business
  return true;
}

__global__ __launch_bounds__(kLines) void text_parse_kernel(ParseArgs a)
{
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsText];
  __shared__ int tooLong;
  __shared__ unsigned long long sumLabels;
  if (threadIdx.x == 0) sumLabels = 0;
  const unsigned j0 = blockIdx.x * kLines;
  const unsigned j1 = min(j0 + kLines, a.nLines);
  // the caller's line count is not the block's: nl[] holds nothing to go by beyond the real count -- every parse block leaves before it
  // reads a position (the flag voids the block, the host packer redoes it)
  if (a.segTotal[0] != a.nLines) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.flag, 4); return; }
  // text of the block's lines: [t0, t1) (the newline of the last line included)
  const size_t t0 = j0 ? (size_t)a.nl[j0 - 1] + 1 : 0, t1 = (size_t)a.nl[j1 - 1] + 1;
  if (threadIdx.x == 0) tooLong = (t1 - t0 > (size_t)kLdsText - 32) ? 1 : 0;
  __syncthreads();
  if (tooLong) { if (threadIdx.x == 0) atomicOr(a.flag, 2); return; }
  const size_t a0 = t0 & ~(size_t)15;                             // staged from the 16-byte line below t0: lds[k] = text[a0 + k]
  for (size_t k = (size_t)threadIdx.x * 16; a0 + k < t1; k += (size_t)kLines * 16) {
    if (a0 + k + 16 <= a.bytes) *(uint4 *)(lds + k) = *(const uint4 *)(a.text + a0 + k);
    else for (size_t q = 0; a0 + k + q < a.bytes; q++) lds[k + q] = (unsigned char)a.text[a0 + k + q];
  }
  __syncthreads();
  const unsigned j = j0 + threadIdx.x;
  if (j >= j1) { if (a.blockSum) __syncthreads(); return; }       // (the barrier of the label sum below)
  const unsigned b = (unsigned)((j ? (size_t)a.nl[j - 1] + 1 : 0) - a0), e = (unsigned)((size_t)a.nl[j] - a0);
  long long v2, v3, label; int nTok, strand; unsigned tokLen;
  bool plain = parse_line((const unsigned char *)lds, b, e, a.weighted != 0, v2, v3, nTok, tokLen, strand, label);
  int cls = -1, start = 0, stop = 0, wv = 1;
  long long lineLabel = 0;                                        // GetLabelValue of the line (genomic_intervals.cpp:1081-1085), whatever becomes of it
  if (plain) {
    lineLabel = 1;
    if (a.weighted) { const long long lv = nTok >= 4 ? label : 0; lineLabel = lv < a.maxLabel ? lv : a.maxLabel; }
    const long long S = v2 + 1, E = v3;                           // BED: start = atol(col2) + 1, stop = atol(col3)
    if (S >= INT_MAX - 1 || E >= INT_MAX - 1) plain = false;
    // the order check of the sorted merge (NextQuery :5889-5898 via IsBefore :396-401): key (chromosome, [strand,] start) against the line before
    if (plain && a.sortedRules) {
      bool before = false, havePrev = true;
      const unsigned char *pn; unsigned pl; int ps; long long pstart;
      long long q2 = 0, q3 = 0, ql = 0; int qn = 0, qs = '+'; unsigned qlen = 0;
      if (j > j0) {
        const unsigned pb = (unsigned)((j - 1 ? (size_t)a.nl[j - 2] + 1 : 0) - a0), pe = (unsigned)((size_t)a.nl[j - 1] - a0);
        if (!parse_line((const unsigned char *)lds, pb, pe, false, q2, q3, qn, qlen, qs, ql)) havePrev = false;   // (that line voids the block anyway)
        pn = lds + pb; pl = qlen; ps = qs; pstart = q2 + 1;
      } else if (j > 0) {
        const size_t pb = j - 1 ? (size_t)a.nl[j - 2] + 1 : 0, pe = a.nl[j - 1];
        const unsigned char *g = (const unsigned char *)a.text + pb;
        // (a line the parser refuses voids the block from its own thread; one that is merely too long to look at here leaves the
        // order of this line undecided -- not a plain case: the host packer takes the block)
        if (pe - pb > 4096) { havePrev = false; plain = false; }
        else if (!parse_line(g, 0u, (unsigned)(pe - pb), false, q2, q3, qn, qlen, qs, ql)) havePrev = false;
        pn = g; pl = qlen; ps = qs; pstart = q2 + 1;
      } else {
        havePrev = a.havePrev != 0;
        if (a.prevLost) plain = false;                            // the seam's key is not here to compare with: the host packer's block
        pn = (const unsigned char *)a.names + a.prevNameOff; pl = a.prevNameLen; ps = a.prevStrand; pstart = a.prevStart;
      }
      if (havePrev) {
        int d = 0;
        const unsigned n = tokLen < pl ? tokLen : pl;
        for (unsigned k = 0; k < n && d == 0; k++) d = (int)lds[b + k] - (int)pn[k];     // strcmp
        if (d == 0) d = (int)tokLen - (int)pl;
        if (d != 0) before = d < 0;
        else if (a.byStrand && strand != ps) before = strand < ps;
        else before = S < pstart;
      }
      if (before) plain = false;                                  // "query regions are not sorted": the host says so, with the line number
    }
    if (plain) {
      // chromosome -> class through the hash table of the index set's names (unknown: the line is dropped, :5719-5720)
      const unsigned h = fnv1a(lds + b, tokLen);
      int id = -1;
      for (unsigned q = h & a.tableMask;; q = (q + 1) & a.tableMask) {
        const ChromEntry en = a.table[q];
        if (en.id < 0) break;
        if (en.hash == h && en.len == tokLen) {
          bool same = true;
          for (unsigned k = 0; k < tokLen && same; k++) same = (unsigned char)a.names[en.off + k] == lds[b + k];
          if (same) { id = en.id; break; }
        }
      }
      if (a.scanRules == 1 && (S > E || E <= 0)) id = -1;         // the unsorted scanner skips such an interval before it looks at the chromosome (:5039)
      if (id >= 0) {
        if (!a.scanRules && !a.sortedRules && (E <= 0 || S > E)) plain = false;   // the unsorted algorithm's errors (:5740-5741): the host reports them
        else {
          cls = id + ((a.strandAware && strand == '-') ? a.nChrom : 0);
          start = (int)S; stop = (int)E;
          if (a.weighted) { const long long lv = nTok >= 4 ? label : 0; wv = (int)(lv < a.maxLabel ? lv : a.maxLabel); }
        }
      }
    }
  }
  if (!plain) atomicOr(a.flag, 1);
  if (a.blockSum) {
    if (plain && lineLabel != 0) atomicAdd(&sumLabels, (unsigned long long)lineLabel);
    __syncthreads();
    if (threadIdx.x == 0 && sumLabels != 0) atomicAdd(a.blockSum, sumLabels);
  }
  a.tri[3 * (size_t)j] = cls; a.tri[3 * (size_t)j + 1] = start; a.tri[3 * (size_t)j + 2] = stop;
  if (a.weighted) a.w[j] = wv;
  if (a.blkMinus && cls >= a.nChrom) atomicAdd(&a.blkMinus[blockIdx.x], 1u);
}

// strand-aware runs: the '+' lines first, then the '-' lines, each group in file order -- a position-sorted stream of both strands
// becomes two class-sorted runs for the streaming kernel (what the host packer does per batch).  blkOff = exclusive prefix of the
// '-' counts per parse block, the total behind the last.
__global__ __launch_bounds__(kLines) void strand_group_kernel(const int *__restrict__ tri, const int *__restrict__ w, unsigned nLines, int nChrom,
                                                              const unsigned *__restrict__ blkOff, unsigned nBlocks, int *__restrict__ tri2, int *__restrict__ w2,
                                                              const int *__restrict__ flag)
{
  __shared__ unsigned waveMinus[kLines / 64];
  const unsigned j = blockIdx.x * kLines + threadIdx.x;
  const bool in = j < nLines;
  if (*flag) { if (in) { tri2[3 * (size_t)j] = -1; tri2[3 * (size_t)j + 1] = 0; tri2[3 * (size_t)j + 2] = 0; if (w2) w2[j] = 1; } return; }   // a block that goes back to the host: nothing to count
  int c = -1, s0 = 0, e0 = 0, wv = 1;
  if (in) { c = tri[3 * (size_t)j]; s0 = tri[3 * (size_t)j + 1]; e0 = tri[3 * (size_t)j + 2]; if (w) wv = w[j]; }
  const bool minus = in && c >= nChrom;
  const unsigned long long m = __ballot(minus);
  const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
  if (lane == 0) waveMinus[wv_id] = (unsigned)__popcll(m);
  __syncthreads();
  unsigned before = (unsigned)__popcll(m & ((1ull << lane) - 1));
  for (int k = 0; k < wv_id; k++) before += waveMinus[k];
  if (!in) return;
  const unsigned minusBefore = blkOff[blockIdx.x] + before, totalMinus = blkOff[nBlocks], nPlus = nLines - totalMinus;
  const size_t at = minus ? (size_t)nPlus + minusBefore : (size_t)j - minusBefore;
  tri2[3 * at] = c; tri2[3 * at + 1] = s0; tri2[3 * at + 2] = e0;
  if (w2) w2[at] = wv;
}

// a block that is not plain is not counted at all: the host packer redoes it
__global__ __launch_bounds__(256) void text_void_kernel(int *__restrict__ tri, unsigned nLines, const int *__restrict__ flag,
                                                        unsigned long long *blockSum, unsigned long long *labelSum)
{
  // (scans: the label values of a plain block's lines join the call's total; a block that goes back to the host brings its own)
  if (blockSum && blockIdx.x == 0 && threadIdx.x == 0) { if (*flag == 0) atomicAdd(labelSum, *blockSum); *blockSum = 0; }
  if (*flag == 0) return;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < nLines; j += (size_t)gridDim.x * 256) tri[3 * j] = -1;
}

hipError_t launch_tokenize(const TextDevice &d, const TextTables &t, const gtx_text_rules &r, size_t bytes, unsigned nLines, hipStream_t st, int scanRules)
{
  const unsigned nSeg = (unsigned)((bytes + kSeg - 1) / kSeg);
  nl_count_kernel<<<nSeg, 64, 0, st>>>(d.text, bytes, d.segCount);
  nl_scan_kernel<<<1, 1024, 0, st>>>(d.segCount, nSeg);
  nl_write_kernel<<<nSeg, 64, 0, st>>>(d.text, bytes, d.segCount, d.nl, nLines);
  ParseArgs a;
  a.text = d.text; a.bytes = bytes; a.nl = d.nl; a.nLines = nLines;
  a.segTotal = d.segCount + nSeg; a.table = (const ChromEntry *)t.table; a.tableMask = t.tableMask; a.names = t.names;
  a.nChrom = r.n_chrom; a.strandAware = r.strand_aware; a.sortedRules = r.sorted_rules; a.byStrand = r.sorted_by_strand;
  a.weighted = r.max_label_value > 1; a.maxLabel = r.max_label_value;
  a.havePrev = r.have_prev && t.prevLen > 0; a.prevLost = r.have_prev && t.prevLen == 0; a.prevNameOff = t.prevOff; a.prevNameLen = t.prevLen; a.prevStrand = r.prev_strand; a.prevStart = r.prev_start;
  a.tri = d.tri; a.w = d.w; a.flag = d.flag;
  a.scanRules = scanRules; a.blockSum = d.labelSum ? d.blockSum : nullptr;
  const unsigned nBlocks = (nLines + kLines - 1) / kLines;
  a.blkMinus = nullptr;
  if (r.strand_aware && d.tri2) {
    a.blkMinus = d.blkMinus;
    hipError_t e = hipMemsetAsync(d.blkMinus, 0, sizeof(unsigned) * ((size_t)nBlocks + 1), st);
    if (e != hipSuccess) return e;
  }
  text_parse_kernel<<<nBlocks, kLines, 0, st>>>(a);
  text_void_kernel<<<256, 256, 0, st>>>(d.tri, nLines, d.flag, d.labelSum ? d.blockSum : nullptr, d.labelSum);
  if (a.blkMinus) {
    nl_scan_kernel<<<1, 1024, 0, st>>>(d.blkMinus, nBlocks);
    strand_group_kernel<<<nBlocks, kLines, 0, st>>>(d.tri, a.weighted ? d.w : nullptr, nLines, r.n_chrom, d.blkMinus, nBlocks, d.tri2, a.weighted ? d.w2 : nullptr, d.flag);
  }
  return hipGetLastError();
}

// the hash table of the chromosome names and the name blob (+ room for the seam's name behind it), host side
void build_tables(const gtx_text_rules &r, std::vector<int32_t> *table, unsigned *mask, std::string *blob)
{
  unsigned size = 64;
  while (size < 4u * (unsigned)std::max(r.n_chrom, 1)) size <<= 1;
  *mask = size - 1;
  table->assign((size_t)size * 4, 0);
  for (unsigned q = 0; q < size; q++) (*table)[4 * q + 1] = -1;
  blob->clear();
  for (int c = 0; c < r.n_chrom; c++) {
    const char *nm = r.chrom_names[c];
    const unsigned len = (unsigned)strlen(nm), off = (unsigned)blob->size();
    blob->append(nm, len);
    unsigned h = 2166136261u;
    for (unsigned i = 0; i < len; i++) { h ^= (unsigned char)nm[i]; h *= 16777619u; }
    unsigned q = h & *mask;
    while ((*table)[4 * q + 1] >= 0) q = (q + 1) & *mask;
    (*table)[4 * q] = (int32_t)h; (*table)[4 * q + 1] = c; (*table)[4 * q + 2] = (int32_t)off; (*table)[4 * q + 3] = (int32_t)len;
  }
}

}  // namespace gtxtext
