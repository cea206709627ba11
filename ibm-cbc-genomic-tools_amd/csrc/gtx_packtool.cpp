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
