// gtx_packtool -- host-only view of the BED ingest (gtx_bed.*): packs a BED stream with the rules of one
// of the four consumers and prints the packed triples as text.  No GPU, no libgtx: this is how the
// CPU test suite checks the packer (line handling, tokenising, order checks, error text and line numbers,
// thread-count independence) on machines without an MI355X.
//
//   gtx_packtool MODE [-t THREADS] [-s] [-a] [-l MAXLABEL] [-c CHROM,CHROM,...] [FILE]
//     MODE   ou = overlaps/unsorted   os = overlaps/sorted   su = scan/unsorted   ss = scan/sorted
//     -s sorted by strand   -a strand-aware classes   -c known chromosomes (default: all seen in FILE order? no:
//        the list is required for reproducible class ids)
//   output: one line per packed read "class start end [weight]", then "# lines=N"; errors like the CLIs.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "gtx_bed.h"
#include "gtx_stats.h"

using namespace gtxhost;

// ---- synthetic BED files (the workload of SURVEY 8(d): hg38 chromosome lengths) -------------------------------------------
#include <algorithm>
#include <thread>
#include <vector>
static int SynthBed(bool refs, long long n, unsigned long long seed, const char *out_path, long read_len)
{
  static const struct { const char *name; long len; } chr[24] = {   // strcmp order
    {"chr1", 248956422}, {"chr10", 133797422}, {"chr11", 135086622}, {"chr12", 133275309}, {"chr13", 114364328}, {"chr14", 107043718},
    {"chr15", 101991189}, {"chr16", 90338345}, {"chr17", 83257441}, {"chr18", 80373285}, {"chr19", 58617616}, {"chr2", 242193529},
    {"chr20", 64444167}, {"chr21", 46709983}, {"chr22", 50818468}, {"chr3", 198295559}, {"chr4", 190214555}, {"chr5", 181538259},
    {"chr6", 170805979}, {"chr7", 159345973}, {"chr8", 145138636}, {"chr9", 138394717}, {"chrX", 156040895}, {"chrY", 57227415}};
  double total = 0; for (auto &c : chr) total += (double)c.len;
  std::vector<long long> per(24); long long given = 0;
  for (int c = 0; c < 24; c++) { per[c] = (long long)((double)n * chr[c].len / total); given += per[c]; }
  for (int c = 0; given < n; c = (c + 1) % 24) { per[c]++; given++; }
  std::vector<long long> first(25, 0);
  for (int c = 0; c < 24; c++) first[c + 1] = first[c] + per[c];
  std::vector<std::string> text(24);
  auto make = [&](int c) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)(c + 1) * 0xBF58476D1CE4E5B9ull;
    auto next = [&x]() { x += 0x9E3779B97F4A7C15ull; unsigned long long z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    const long long k = per[c];
    std::vector<unsigned> s((size_t)k), ln(refs ? (size_t)k : 0);
    const unsigned long long span = (unsigned long long)(chr[c].len - (refs ? 2001 : read_len + 1));
    for (long long i = 0; i < k; i++) s[(size_t)i] = (unsigned)(next() % span);
    std::sort(s.begin(), s.end());
    if (refs) for (long long i = 0; i < k; i++) ln[(size_t)i] = 50 + (unsigned)(next() % 1950);
    std::string &t = text[c];
    t.reserve((size_t)k * (refs ? 40 : 30));
    char buf[96];
    const size_t nl = strlen(chr[c].name);
    for (long long i = 0; i < k; i++) {
      char *p = buf; memcpy(p, chr[c].name, nl); p += nl; *p++ = '\t';
      auto put = [&p](unsigned long long v) { char tmp[24]; int d = 0; do { tmp[d++] = (char)('0' + v % 10); v /= 10; } while (v); while (d) *p++ = tmp[--d]; };
      put(s[(size_t)i]); *p++ = '\t'; put((unsigned long long)s[(size_t)i] + (refs ? ln[(size_t)i] : (unsigned long long)read_len));
      if (refs) { *p++ = '\t'; *p++ = 'g'; put((unsigned long long)(first[c] + i)); }
      *p++ = '\n';
      t.append(buf, (size_t)(p - buf));
    }
  };
  {
    std::vector<std::thread> th;
    for (int c = 0; c < 24; c++) th.emplace_back(make, c);
    for (auto &x : th) x.join();
  }
  FILE *o = fopen(out_path, "wb");
  if (!o) { fprintf(stderr, "Error: cannot create file '%s'!\n", out_path); return 1; }
  for (int c = 0; c < 24; c++) if (fwrite(text[c].data(), 1, text[c].size(), o) != text[c].size()) { fprintf(stderr, "Error: cannot write file '%s'!\n", out_path); return 1; }
  return fclose(o) == 0 ? 0 : 1;
}

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "usage: gtx_packtool ou|os|su|ss [-t N] [-s] [-a] [-l MAX] -c chr1,chr2,... [FILE]\n"); return 2; }
  PackOptions opt;
  std::string m = argv[1];
  if (m == "pack") {
    // gtx_packtool pack IN.bed[.gz] OUT.gtx : tokenise once, keep the columns (gtx_bed.h, "packed region files")
    if (argc != 4) { fprintf(stderr, "usage: gtx_packtool pack IN.bed OUT.gtx\n"); return 2; }
    std::string err;
    LineSource *src = LineSource::Open(argv[2], &err);
    if (!src) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    PackError e;
    if (!WriteGtx(src, argv[3], &e)) {
      if (e.no_prefix) fprintf(stderr, "%s\n", e.msg.c_str()); else fprintf(stderr, "\nError: Line %ld: %s\n", e.line, e.msg.c_str());
      return 1;
    }
    delete src;
    return 0;
  }
  if (m == "synth" || m == "synthrefs") {
    // gtx_packtool synth N SEED OUT.bed [LEN]      N reads of LEN bp (default 50) on the 24 hg38 chromosomes, in proportion to their
    //                                              length, sorted by (chromosome in strcmp order, start), BED3
    // gtx_packtool synthrefs M SEED OUT.bed        M regions of 50..2000 bp, same order, BED4 with labels g0, g1, ...
    // Workload files for the end-to-end timings (bench.py text_to_stdout): written by a thread per chromosome, ~1 GB/s.
    if (argc < 5) { fprintf(stderr, "usage: gtx_packtool synth|synthrefs N SEED OUT.bed [LEN]\n"); return 2; }
    return SynthBed(m == "synthrefs", atoll(argv[2]), strtoull(argv[3], NULL, 10), argv[4], argc > 5 ? atol(argv[5]) : 50);
  }
  if (m == "stats") {
    // the host-side tail probabilities of `genomic_scans peaks` (gtx_stats.h), one "b K P N" / "p K MU" / "g X" query per stdin line
    char kind; double x, y, z;
    char line[256];
    while (fgets(line, sizeof line, stdin)) {
      int n = sscanf(line, " %c %lf %lf %lf", &kind, &x, &y, &z);
      if (n >= 4 && kind == 'b') printf("%.17g\n", gtxstats::BinomialQ((long)x, y, (long)z));
      else if (n >= 3 && kind == 'p') printf("%.17g\n", gtxstats::PoissonQ((long)x, y));
      else if (n >= 2 && kind == 'g') printf("%.17g\n", gtxstats::GaussianQ(x));
      else { fprintf(stderr, "bad query: %s", line); return 2; }
    }
    return 0;
  }
  if (m == "ou") opt.mode = PACK_OVERLAPS_UNSORTED; else if (m == "os") opt.mode = PACK_OVERLAPS_SORTED;
  else if (m == "su") opt.mode = PACK_SCAN_UNSORTED; else if (m == "ss") opt.mode = PACK_SCAN_SORTED;
  else { fprintf(stderr, "unknown mode '%s'\n", argv[1]); return 2; }
  ChromTable chroms;
  const char *file = NULL; size_t batch = 1u << 20;
  bool quiet = false; long long checksum = 0, n_reads = 0;                    // -q: timing runs, print only a checksum
  for (int a = 2; a < argc; a++) {
    if (!strcmp(argv[a], "-t") && a + 1 < argc) opt.threads = atoi(argv[++a]);
    else if (!strcmp(argv[a], "-s")) opt.sorted_by_strand = true;
    else if (!strcmp(argv[a], "-a")) opt.strand_aware = true;
    else if (!strcmp(argv[a], "-z")) opt.collect_zero_length = true;
    else if (!strcmp(argv[a], "-q")) quiet = true;
    else if (!strcmp(argv[a], "-l") && a + 1 < argc) opt.max_label_value = atol(argv[++a]);
    else if (!strcmp(argv[a], "-b") && a + 1 < argc) batch = (size_t)atol(argv[++a]);
    else if (!strcmp(argv[a], "-c") && a + 1 < argc) {
      std::string list = argv[++a];
      size_t p = 0;
      while (p <= list.size()) { size_t q = list.find(',', p); if (q == std::string::npos) q = list.size(); if (q > p) chroms.Add(list.substr(p, q - p).c_str()); p = q + 1; }
    } else file = argv[a];
  }
  chroms.Freeze();
  opt.chroms = &chroms;
  std::string err;
  LineSource *src = nullptr; GtxView *packed = nullptr;
  if (GtxView::IsGtx(file)) { packed = GtxView::Open(file, &err); if (!packed) { fprintf(stderr, "%s\n", err.c_str()); return 1; } }
  else { src = LineSource::Open(file, &err); if (!src) { fprintf(stderr, "%s\n", err.c_str()); return 1; } }
  BedPacker packer_text(src, opt), packer_packed(packed, opt);
  BedPacker &packer = packed ? packer_packed : packer_text;
  PackedBatch b; PackError e; int64_t lines = 0;
  for (;;) {
    bool more = packer.NextBatch(&b, batch, &e);
    if (e.set) {
      fflush(stdout);
      if (e.no_prefix) fprintf(stderr, "%s\n", e.msg.c_str()); else fprintf(stderr, "\nError: Line %ld: %s\n", e.line, e.msg.c_str());
      return 1;
    }
    if (quiet) { long long cs = 0; for (size_t i = 0; i < b.tri.size(); i++) cs += b.tri[i]; checksum += cs; n_reads += (long long)(b.tri.size() / 3); b.tri.clear(); b.w.clear(); b.zero_len.clear(); lines += b.n_lines; b.n_lines = 0; if (!more) break; continue; }
    for (size_t i = 0; i + 2 < b.tri.size(); i += 3) {
      if (b.w.empty()) printf("%d %d %d\n", b.tri[i], b.tri[i + 1], b.tri[i + 2]);
      else printf("%d %d %d %d\n", b.tri[i], b.tri[i + 1], b.tri[i + 2], b.w[i / 3]);
    }
    for (size_t i = 0; i + 2 < b.zero_len.size(); i += 3) printf("# zero %d %d %d\n", b.zero_len[i], b.zero_len[i + 1], b.zero_len[i + 2]);
    lines += b.n_lines;
    if (!more) break;
  }
  if (quiet) printf("# reads=%lld checksum=%lld\n", n_reads, checksum);
  printf("# lines=%ld\n", (long)lines);
  delete src; delete packed;
  return 0;
}
