// api_caller -- a caller written the way the reference's own tools use the class API of gtools/genomic_intervals.h (the iteration
// loops of genomic_overlaps.cpp: e.g. `overlap` :630-660, `subset` :797-811), compiled against this package's
// csrc/genomic_intervals.h.  It exercises the members SURVEY 8(b) lists beyond the two reductions: the FILE* constructor,
// GetQuery/NextQuery, GetOverlap/NextOverlap, CountQueryOverlaps, CalcQueryCoverage, Done.
//   api_caller pairs|qcount|qcover|icount|subclass [-S] [-s] [-i] [-gaps] [-B bits] [--max-label-value N] REF QUERY
// subclass: the pairs of every other query, walked through a caller-defined subclass of GenomicRegionSetOverlaps; then a caller-defined scanner
// pairs:  "<query line>\t<index label>" per overlap, in iteration order;  qcount / qcover: "<query line>\t<value>" per query
// icount: CountIndexOverlaps with both sets in memory, "<index label>\t<count>" per index region
// (one walk per query: a second GetOverlap walk of the same query finds the merge's buffer already consumed, in the reference too)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "genomic_intervals.h"

// `subclass` mode: classes a reference-side caller could have written against gtools/genomic_intervals.h -- derived straight from
// the two abstract bases, implementing exactly the reference's pure virtuals (genomic_intervals.h:2403-2419, :2213-2217) and nothing
// else.  They must compile against this package's header and be used through base-class pointers.
class EveryOtherQuery : public GenomicRegionSetOverlaps
{
 public:
  EveryOtherQuery(GenomicRegionSet *Q, GenomicRegionSet *I, const char *bits) : GenomicRegionSetOverlaps(Q, I), inner(Q, I, bits), calls(0) {}
  GenomicRegion *GetQuery() { calls++; current_qreg = inner.GetQuery(); return current_qreg; }
  GenomicRegion *NextQuery()                                           // skips every second query: a walk only this class defines
  {
    calls++;
    if (inner.NextQuery() == NULL) { current_qreg = NULL; return NULL; }
    current_qreg = inner.NextQuery();
    return current_qreg;
  }
  GenomicRegion *GetMatch() { calls++; return inner.GetMatch(); }
  GenomicRegion *NextMatch() { calls++; return inner.NextMatch(); }
  bool Done() { return inner.Done(); }
  UnsortedGenomicRegionSetOverlaps inner;
  long calls;
};

class ThreeWindows : public GenomicRegionSetScanner
{
 public:
  ThreeWindows(GenomicRegionSet *R, StringLIntMap *bounds) : GenomicRegionSetScanner(R, bounds, 10, 20, 1, true, '1'), at(0) {}
  void PrintInterval(FILE *out_file = stdout) { fprintf(out_file, "window %ld", at); }
  GenomicInterval *GetInterval() { return NULL; }
  long int Next() { return at < 3 ? 100 + at++ : -1; }
  long int Next(GenomicRegionSet *) { return Next(); }
  long int Next(GenomicRegionSetIndex *) { return Next(); }
  long at;
};

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: api_caller pairs|qcount|qcover [OPTIONS] REF QUERY\n"); return 2; }
  const bool pairs = !strcmp(argv[1], "pairs"), cover = !strcmp(argv[1], "qcover");
  bool sorted = false, by_strand = false, ignore_strand = false, gaps = false; const char *bits = "17,20,23,26"; long mlv = 1;
  int a = 2;
  for (; a < argc && argv[a][0] == '-'; a++) {
    if (!strcmp(argv[a], "-S")) sorted = true; else if (!strcmp(argv[a], "-s")) by_strand = true; else if (!strcmp(argv[a], "-i")) ignore_strand = true;
    else if (!strcmp(argv[a], "-gaps")) gaps = true; else if (!strcmp(argv[a], "-B")) bits = argv[++a]; else if (!strcmp(argv[a], "--max-label-value")) mlv = atol(argv[++a]);
    else { fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
  }
  GenomicRegionSet RefRegSet(argv[a], 10000, false, true, true);
  if (!strcmp(argv[1], "subclass")) {
    FILE *qf = fopen(argv[a + 1], "r");
    if (!qf) { fprintf(stderr, "cannot open %s\n", argv[a + 1]); return 2; }
    GenomicRegionSet Queries(qf, 10000, false, false, true);
    EveryOtherQuery own(&Queries, &RefRegSet, bits);
    GenomicRegionSetOverlaps *ov = &own;                                // everything below goes through the base class
    for (GenomicRegion *q = ov->GetQuery(); q != NULL; q = ov->NextQuery())
      for (GenomicRegion *r = ov->GetOverlap(gaps, ignore_strand); r != NULL; r = ov->NextOverlap(gaps, ignore_strand))
        printf("%ld\t%s\n", q->n_line, r->LABEL);
    printf("virtual calls %s\n", own.calls > 0 ? "seen" : "missing");
    StringLIntMap bounds; bounds["chr1"] = 1000;
    ThreeWindows win(&Queries, &bounds);
    GenomicRegionSetScanner *sc = &win;
    for (long int v = sc->Next(); v != -1; v = sc->Next()) { printf("%ld\t", v); sc->PrintInterval(stdout); printf("\n"); }
    fclose(qf);
    return 0;
  }
  if (!strcmp(argv[1], "icount")) {                                     // CountIndexOverlaps with the QUERY set held in memory as well
    GenomicRegionSet Queries(argv[a + 1], 10000, false, true, true);
    GenomicRegionSetOverlaps *ov;
    if (sorted) ov = new SortedGenomicRegionSetOverlaps(&Queries, &RefRegSet, by_strand);
    else ov = new UnsortedGenomicRegionSetOverlaps(&Queries, &RefRegSet, bits);
    unsigned long int *hits = ov->CountIndexOverlaps(gaps, ignore_strand, mlv);
    for (long int k = 0; k < RefRegSet.n_regions; k++) printf("%s\t%lu\n", RefRegSet.R[k]->LABEL, hits[k]);
    delete[] hits; delete ov;
    return 0;
  }
  FILE *qf = fopen(argv[a + 1], "r");                                   // the FILE* constructor (genomic_intervals.h:1836)
  if (!qf) { fprintf(stderr, "cannot open %s\n", argv[a + 1]); return 2; }
  GenomicRegionSet TestRegSet(qf, 10000, false, false, true);
  GenomicRegionSetOverlaps *overlaps;
  if (sorted) overlaps = new SortedGenomicRegionSetOverlaps(&TestRegSet, &RefRegSet, by_strand);
  else overlaps = new UnsortedGenomicRegionSetOverlaps(&TestRegSet, &RefRegSet, bits);
  for (GenomicRegion *qreg = overlaps->GetQuery(); qreg != NULL; qreg = overlaps->NextQuery()) {
    if (pairs) {
      for (GenomicRegion *ireg = overlaps->GetOverlap(gaps, ignore_strand); ireg != NULL; ireg = overlaps->NextOverlap(gaps, ignore_strand))
        printf("%ld\t%s\n", qreg->n_line, ireg->LABEL);
    } else {
      const unsigned long v = cover ? overlaps->CalcQueryCoverage(gaps, ignore_strand, mlv) : overlaps->CountQueryOverlaps(gaps, ignore_strand, mlv);
      printf("%ld\t%lu\n", qreg->n_line, v);
    }
  }
  delete overlaps;
  fclose(qf);
  return 0;
}
