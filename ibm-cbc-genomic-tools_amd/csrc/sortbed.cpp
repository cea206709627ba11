// sortbed -- MI355X edition of GenomicTools' bin/sortbed: BED lines into the order the sorted algorithms of genomic_overlaps /
// genomic_scans insist on (-S: SortedGenomicRegionSetOverlaps genomic_intervals.cpp:5807-5937, SortedGenomicRegionSetScanner :4928-4957).
//
// The reference's script is `sort -k1,1 -k6,6 -k2,2n` (chromosome, strand, start), with -i `sort -k1,1 -k2,2n` (chromosome, start).
// Here the lines are read and keyed by host threads, the order is found on the GPU (gtx_sort, include/gtx.h) and the lines leave in
// that order -- byte for byte what `LC_ALL=C sort` with those keys prints: keys as sort(1) cuts them (a field is a run of blanks
// followed by a run of non-blanks; -k1,1 and -k6,6 compare the field's bytes, -k2,2n its leading number), lines that agree in every
// key by their bytes (sort's last resort).  The C locale is what the tools themselves compare chromosomes with (strcmp).
//
//   sortbed [-i] [-o OUT.gtx] [FILE]       FILE: BED text, default stdin; sorted lines on stdout
//     -o OUT.gtx   write the sorted regions as a packed region file instead (csrc/gtx_bed.h: what genomic_overlaps / genomic_scans
//                  take in place of text; the lines must be BED lines then)
//
// Column 2 must be a plain integer of 32 bits (what a BED start is).  Without a GPU the tool stops with an error.
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "gtx.h"
#include "gtx_bed.h"

using namespace gtxhost;

static inline bool IsBlank(char c) { return c == ' ' || c == '\t'; }

struct Line { const char *p; uint32_t len; };

// field k (1-based) of a line as sort(1) cuts it without -t: [b, e) with the field's leading blanks
static void Field(const Line &l, int k, const char **b, const char **e)
{
  const char *q = l.p, *end = l.p + l.len, *fb = q;
  for (int f = 1; ; f++) {
    fb = q;
    while (q < end && IsBlank(*q)) q++;
    while (q < end && !IsBlank(*q)) q++;
    if (f == k || q == end) { if (f != k) fb = q; break; }
  }
  *b = fb; *e = q;
}

static bool LessBytes(const char *a, size_t na, const char *b, size_t nb)
{
  const int c = memcmp(a, b, std::min(na, nb));
  return c < 0 || (c == 0 && na < nb);
}

int main(int argc, char **argv)
{
  bool ignore_strand = false; const char *out_gtx = NULL, *file = NULL;
  for (int a = 1; a < argc; a++) {
    if (!strcmp(argv[a], "-i")) ignore_strand = true;
    else if (!strcmp(argv[a], "-o") && a + 1 < argc) out_gtx = argv[++a];
    else if (argv[a][0] == '-' && argv[a][1]) { fprintf(stderr, "##\n## USAGE: sortbed [-i] [-o OUT.gtx] <BED-FILE>\n##\n"); return 1; }
    else file = argv[a];
  }
  gtx_ctx *ctx = gtx_create(getenv("GTX_DEVICE") ? atoi(getenv("GTX_DEVICE")) : 0);
  if (!ctx) { fprintf(stderr, "Error: [gtx] %s\n", gtx_last_error(NULL)); return 1; }

  // the whole input, its lines (a last line without '\n' is a line to sort(1), and leaves with one)
  int fd = 0;
  if (file && strcmp(file, "-") != 0) { fd = open(file, O_RDONLY); if (fd < 0) { fprintf(stderr, "sortbed: cannot read: %s: %s\n", file, strerror(errno)); return 2; } }
  std::vector<char> text;
  {
    struct stat sb;
    size_t cap = (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) ? (size_t)sb.st_size + 1 : ((size_t)64 << 20);
    text.resize(cap);
    size_t have = 0;
    for (;;) {
      if (have == text.size()) text.resize(text.size() * 2);
      const ssize_t got = read(fd, text.data() + have, std::min(text.size() - have, (size_t)1 << 30));
      if (got < 0) { if (errno == EINTR) continue; fprintf(stderr, "sortbed: read failed: %s\n", strerror(errno)); return 2; }
      if (got == 0) break;
      have += (size_t)got;
    }
    text.resize(have);
  }
  std::vector<Line> lines;
  {
    // line starts found by the threads, piece by piece
    const int TT = std::max(1, WorkerThreads());
    std::vector<std::vector<Line>> part((size_t)TT);
    const char *base = text.data(), *end = base + text.size();
    ParallelFor(TT, [&](int t) {
      const char *lo = base + text.size() * (size_t)t / TT, *hi = base + text.size() * (size_t)(t + 1) / TT;
      if (lo > base) { while (lo < end && lo[-1] != '\n') lo++; }       // the first line that starts in this piece
      const char *p = lo;
      while (p < hi && p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl : end;
        part[(size_t)t].push_back({p, (uint32_t)(e - p)});
        p = e + 1;
      }
    });
    size_t total = 0; for (auto &v : part) total += v.size();
    lines.reserve(total);
    for (auto &v : part) lines.insert(lines.end(), v.begin(), v.end());
  }
  const size_t n = lines.size();
  if (n >= ((size_t)1 << 32)) { fprintf(stderr, "Error: more than 2^32 - 1 lines!\n"); return 1; }

  // keys: the bytes of field 1 (and 6), the number in field 2
  const int T = std::max(1, WorkerThreads());
  std::vector<std::map<std::string, int>> chromOf((size_t)T), strandOf((size_t)T);
  std::vector<int32_t> start(n);
  std::vector<uint32_t> cLocal(n), sLocal(ignore_strand ? 0 : n);
  std::atomic<long> badLine(-1);
  ParallelFor(T, [&](int t) {
    std::string key; int lastC = -1; std::string lastKey;
    for (size_t i = n * (size_t)t / T; i < n * (size_t)(t + 1) / T; i++) {
      const char *b, *e;
      Field(lines[i], 1, &b, &e);
      if (lastC < 0 || lastKey.size() != (size_t)(e - b) || memcmp(lastKey.data(), b, (size_t)(e - b)) != 0) {
        lastKey.assign(b, e);
        auto it = chromOf[(size_t)t].find(lastKey);
        if (it == chromOf[(size_t)t].end()) it = chromOf[(size_t)t].emplace(lastKey, (int)chromOf[(size_t)t].size()).first;
        lastC = it->second;
      }
      cLocal[i] = (uint32_t)lastC;
      Field(lines[i], 2, &b, &e);
      while (b < e && IsBlank(*b)) b++;
      bool neg = false; long long v = 0; const char *q = b;
      if (q < e && *q == '-') { neg = true; q++; }
      int digits = 0;
      while (q < e && *q >= '0' && *q <= '9' && digits < 11) { v = v * 10 + (*q - '0'); q++; digits++; }
      if (neg) v = -v;
      if (digits > 10 || v > INT_MAX - 2 || v < INT_MIN + 2 || (q < e && (*q == '.' || (*q >= '0' && *q <= '9')))) { long want = -1; badLine.compare_exchange_strong(want, (long)i); start[i] = 0; }
      else start[i] = (int32_t)v;
      if (!ignore_strand) {
        Field(lines[i], 6, &b, &e);
        key.assign(b, e);
        auto it = strandOf[(size_t)t].find(key);
        if (it == strandOf[(size_t)t].end()) it = strandOf[(size_t)t].emplace(key, (int)strandOf[(size_t)t].size()).first;
        sLocal[i] = (uint32_t)it->second;
      }
    }
  });
  if (badLine.load() >= 0) { fprintf(stderr, "\nError: Line %ld: column 2 is not an integer of 32 bits!\n", badLine.load() + 1); return 1; }
  // ranks of the distinct keys in byte order
  auto ranks = [&](std::vector<std::map<std::string, int>> &per, std::vector<std::vector<int>> *toGlobal) {
    std::vector<std::string> all;
    for (auto &m : per) for (auto &kv : m) all.push_back(kv.first);
    std::sort(all.begin(), all.end(), [](const std::string &a, const std::string &b) { return LessBytes(a.data(), a.size(), b.data(), b.size()); });
    all.erase(std::unique(all.begin(), all.end()), all.end());
    toGlobal->assign(per.size(), {});
    for (size_t t = 0; t < per.size(); t++) {
      (*toGlobal)[t].assign(per[t].size(), 0);
      for (auto &kv : per[t])
        (*toGlobal)[t][(size_t)kv.second] = (int)(std::lower_bound(all.begin(), all.end(), kv.first, [](const std::string &a, const std::string &b) { return LessBytes(a.data(), a.size(), b.data(), b.size()); }) - all.begin());
    }
    return (int)all.size();
  };
  std::vector<std::vector<int>> cMap, sMap;
  const int nChromKeys = ranks(chromOf, &cMap);
  const int nStrandKeys = ignore_strand ? 1 : std::max(1, ranks(strandOf, &sMap));
  if ((long long)std::max(1, nChromKeys) * nStrandKeys > INT_MAX) { fprintf(stderr, "Error: too many distinct keys!\n"); return 1; }
  std::vector<int32_t> tri(3 * n);
  ParallelFor(T, [&](int t) {
    for (size_t i = n * (size_t)t / T; i < n * (size_t)(t + 1) / T; i++) {
      const int c = cMap[(size_t)t][cLocal[i]], s = ignore_strand ? 0 : sMap[(size_t)t][sLocal[i]];
      tri[3 * i] = c * nStrandKeys + s; tri[3 * i + 1] = start[i]; tri[3 * i + 2] = start[i];
    }
  });

  // the order, on the device
  std::vector<uint32_t> order(n);
  if (n) {
    const int rc = gtx_sort(ctx, tri.data(), (int64_t)n, std::max(1, nChromKeys) * nStrandKeys, order.data(), NULL);
    if (rc != GTX_OK) { fprintf(stderr, "Error: [gtx %d] %s\n", rc, gtx_last_error(ctx)); return 1; }
  }
  // lines that agree in every key: by their bytes.  A group of such lines belongs to the piece it starts in: the pieces' first group
  // starts are found first (nothing moves meanwhile), then every piece sorts its groups.
  {
    auto same = [&](size_t a, size_t b) { return tri[3 * (size_t)order[a]] == tri[3 * (size_t)order[b]] && tri[3 * (size_t)order[a] + 1] == tri[3 * (size_t)order[b] + 1]; };
    std::vector<size_t> first((size_t)T + 1, n);
    ParallelFor(T, [&](int t) {
      size_t lo = n * (size_t)t / T;
      while (lo > 0 && lo < n && same(lo - 1, lo)) lo++;
      first[(size_t)t] = lo;
    });
    ParallelFor(T, [&](int t) {
      for (size_t i = first[(size_t)t]; i < first[(size_t)t + 1]; ) {
        size_t j = i + 1;
        while (j < n && same(i, j)) j++;
        if (j - i > 1) std::sort(order.begin() + (long)i, order.begin() + (long)j, [&](uint32_t a, uint32_t b) { return LessBytes(lines[a].p, lines[a].len, lines[b].p, lines[b].len); });
        i = j;
      }
    });
  }

  if (out_gtx) {
    // the sorted regions as a packed file: BED fields of every line, in the new order (parsed by the threads, pieces of a multiple of
    // eight records each: a strand bit per record)
    std::vector<uint16_t> cidx(n); std::vector<int32_t> st(n), en(n), lab(n); std::vector<uint8_t> minus((n + 7) / 8, 0);
    std::vector<std::vector<std::string>> localNames((size_t)T);
    std::vector<uint32_t> localIdx(n);
    struct Bad { long line = -1; std::string msg; };
    std::vector<Bad> bad((size_t)T);
    std::vector<char> anyLabel((size_t)T, 0);
    auto piece = [&](int t) { return std::min(n, ((n * (size_t)t / T) + 7) & ~(size_t)7); };
    ParallelFor(T, [&](int t) {
      std::map<std::string, uint32_t> nameOf;
      std::string copy;
      const size_t lo = piece(t), hi = t + 1 == T ? n : piece(t + 1);
      for (size_t i = lo; i < hi; i++) {
        const Line &l = lines[order[i]];
        copy.assign(l.p, l.len);
        BedFields f; char *badTok = nullptr;
        const BedStatus s = ParseBedLine(&copy[0], &f, &badTok);
        const long line_no = (long)order[i] + 1;
        auto fail = [&](const std::string &m) { if (bad[(size_t)t].line < 0) { bad[(size_t)t].line = line_no; bad[(size_t)t].msg = m; } };
        if (s == BED_TOO_FEW_TOKENS) { fail("number of tokens should be at least 3 for BED format!"); return; }
        if (s == BED_BAD_STRAND) { fail(std::string("invalid strand '") + badTok + "'!"); return; }
        if (f.n_tokens == 12) { fail("multi-interval (BED12) regions do not fit a packed region file!"); return; }
        const long v = f.label ? atol(f.label) : 0;
        if (f.start >= INT_MAX - 1 || f.stop >= INT_MAX - 1 || f.start <= INT_MIN + 1 || f.stop <= INT_MIN + 1 || v > INT_MAX || v < INT_MIN) { fail("coordinate or label value does not fit the packed 32-bit representation of the MI355X path!"); return; }
        auto it = nameOf.find(f.chrom);
        if (it == nameOf.end()) { it = nameOf.emplace(f.chrom, (uint32_t)localNames[(size_t)t].size()).first; localNames[(size_t)t].push_back(f.chrom); }
        localIdx[i] = it->second; st[i] = (int32_t)f.start; en[i] = (int32_t)f.stop; lab[i] = (int32_t)v;
        if (f.strand == '-') minus[i >> 3] |= (uint8_t)(1u << (i & 7));
        if (f.label) anyLabel[(size_t)t] = 1;
      }
    });
    for (const Bad &b : bad) if (b.line >= 0) { fprintf(stderr, "\nError: Line %ld: %s\n", b.line, b.msg.c_str()); return 1; }
    std::vector<std::string> names; std::map<std::string, int> nameOf;
    std::vector<std::vector<uint16_t>> toGlobal((size_t)T);
    for (int t = 0; t < T; t++)
      for (const std::string &nm : localNames[(size_t)t]) {
        auto it = nameOf.find(nm);
        if (it == nameOf.end()) {
          if (names.size() >= 65535) { fprintf(stderr, "Error: too many chromosomes for a packed region file!\n"); return 1; }
          it = nameOf.emplace(nm, (int)names.size()).first; names.push_back(nm);
        }
        toGlobal[(size_t)t].push_back((uint16_t)it->second);
      }
    ParallelFor(T, [&](int t) { const size_t lo = piece(t), hi = t + 1 == T ? n : piece(t + 1); for (size_t i = lo; i < hi; i++) cidx[i] = toGlobal[(size_t)t][localIdx[i]]; });
    bool any_label = false; for (char c : anyLabel) any_label |= c != 0;
    PackError e;
    if (!WriteGtxColumns(out_gtx, names, (uint64_t)n, cidx.data(), st.data(), en.data(), minus.data(), any_label ? lab.data() : nullptr, &e)) { fprintf(stderr, "%s\n", e.msg.c_str()); return 1; }
  } else {
    // the lines, in pieces put together by the threads and written in turn
    const size_t per = (size_t)1 << 20;
    for (size_t base = 0; base < n; base += per * (size_t)T) {
      std::vector<std::string> out((size_t)T);
      ParallelFor(T, [&](int t) {
        const size_t lo = std::min(n, base + per * (size_t)t), hi = std::min(n, lo + per);
        size_t bytes = 0;
        for (size_t i = lo; i < hi; i++) bytes += lines[order[i]].len + 1;
        std::string &o = out[(size_t)t];
        o.resize(bytes);
        char *w = &o[0];
        for (size_t i = lo; i < hi; i++) { const Line &l = lines[order[i]]; memcpy(w, l.p, l.len); w += l.len; *w++ = '\n'; }
      });
      for (auto &o : out) if (!o.empty() && fwrite(o.data(), 1, o.size(), stdout) != o.size()) { fprintf(stderr, "Error: cannot write the output!\n"); return 1; }
    }
  }
  fflush(stdout);
  _exit(0);                                                          // everything is written: skip the teardown of the HIP runtime
}
