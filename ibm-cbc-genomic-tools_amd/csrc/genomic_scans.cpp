// genomic_scans -- MI355X edition of the `counts` operation of GenomicTools' genomic_scans
// (reference driver: gtools/genomic_scans.cpp:73-150 options, :399-436 RunCounts, :449-462).
// Sliding-window read counts over the chromosomes of a genome file; the histogram + window sums
// run on the GPU behind the reference's GenomicRegionSetScanner classes.  `peaks` (GSL tail
// probabilities) is outside this path; the -r reference filter is a host-side test per reported window.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "genomic_intervals.h"
#include "gtx_cmdline.h"

static const char *PROGRAM = "genomic_scans";
static const long int BUFFER_SIZE = 10000;

int main(int argc, char *argv[])
{
  if (argc < 2) {
    fprintf(stderr, "\nUSAGE: \n  %s OPERATION [OPTIONS] INPUT-FILES\n\nOPERATIONS (MI355X path): \n"
                    "  counts     Determines input read counts in sliding windows of reference regions.\n\n", PROGRAM);
    return 1;
  }
  std::string op = argv[1];
  if (op[0] == '-') op = op.substr(1);
  if (op == "peaks") { fprintf(stderr, "Operation 'peaks' is outside the MI355X path of this build (counts)!\n"); return 1; }
  if (op != "counts") { fprintf(stderr, "Unknown operation '%s'!\n", op.c_str()); return 1; }

  bool HELP, HELP2, VERBOSE, SORTED, REF_SORTED, IGNORE_STRAND;
  const char *GENOME_REG_FILE, *REF_REG_FILE; char PREPROCESS; long MAX_LABEL_VALUE, WIN_SIZE, WIN_DIST, MIN_READS;
  gtxhost::Options opts;
  opts.Flag("--help", &HELP, "help");
  opts.Flag("-h", &HELP2, "help");
  opts.Flag("-v", &VERBOSE, "verbose mode");
  opts.Flag("-S", &SORTED, "input regions are sorted");
  opts.Flag("-Sref", &REF_SORTED, "reference regions (option -r) are sorted");
  opts.Str("-g", &GENOME_REG_FILE, "", "genome region file");
  opts.Str("-r", &REF_REG_FILE, "", "reference region file");
  opts.Flag("-i", &IGNORE_STRAND, "ignore strand information");
  opts.Char("-op", &PREPROCESS, '1', "preprocess operator (1=start, c=center)");
  opts.Long("--max-label-value", &MAX_LABEL_VALUE, 1, "maximum region label value to be used");
  opts.Long("-w", &WIN_SIZE, 500, "window size (must be a multiple of window distance)");
  opts.Long("-d", &WIN_DIST, 25, "window distance");
  opts.Long("-min", &MIN_READS, 10, "minimum reads in window");
  int next_arg = opts.Parse(argc, argv, 2);
  if (HELP || HELP2) { opts.Usage(PROGRAM, "counts", "[OPTIONS] <REG-FILE>"); return 1; }
  _MESSAGES_ = VERBOSE;

  char *INPUT_REG_FILE = next_arg == argc ? NULL : argv[next_arg];
  StringLIntMap *bounds = ReadBounds((char *)GENOME_REG_FILE, false);
  GenomicRegionSet InputRegSet(INPUT_REG_FILE, BUFFER_SIZE, VERBOSE, false, true);
  GenomicRegionSetScanner *scanner;
  if (SORTED) scanner = new SortedGenomicRegionSetScanner(&InputRegSet, bounds, WIN_DIST, WIN_SIZE, MAX_LABEL_VALUE, IGNORE_STRAND, PREPROCESS);
  else scanner = new UnsortedGenomicRegionSetScanner(&InputRegSet, bounds, WIN_DIST, WIN_SIZE, MAX_LABEL_VALUE, IGNORE_STRAND, PREPROCESS);
  // reference regions (-r): report only the windows that overlap one of them (genomic_scans.cpp:411-420)
  GenomicRegionSet *RefRegSet = NULL;
  GenomicRegionSetIndex *RefIndex = NULL;
  if (strlen(REF_REG_FILE) > 0) {
    if (REF_SORTED) RefRegSet = new GenomicRegionSet((char *)REF_REG_FILE, BUFFER_SIZE, VERBOSE, false, true);
    else {
      RefRegSet = new GenomicRegionSet((char *)REF_REG_FILE, BUFFER_SIZE, VERBOSE, true, true);
      RefIndex = new GenomicRegionSetIndex(RefRegSet, "17,20,23,26");
    }
  }
  for (long int v = REF_SORTED ? scanner->Next(RefRegSet) : scanner->Next(RefIndex); v != -1; v = REF_SORTED ? scanner->Next(RefRegSet) : scanner->Next(RefIndex)) {
    if (v >= MIN_READS) {
      printf("%ld\t", v);
      scanner->PrintInterval();
      printf("\n");
    }
  }
  delete scanner;
  delete bounds;
  delete RefIndex;
  delete RefRegSet;
  return 0;
}
