// genomic_overlaps -- MI355X edition of the `count`, `rpkm`, `coverage` and `density` operations of
// GenomicTools' genomic_overlaps (reference driver: gtools/genomic_overlaps.cpp:73-261 options,
// :298-305, :408-431 count, :438-459 coverage, :466-490 density, :746-775 rpkm).  Same command line,
// same output, same errors; the reductions are GenomicRegionSetOverlaps::CountIndexOverlaps /
// CalcIndexCoverage of this package, i.e. HIP kernels through libgtx.so.  The other seven operations
// of the reference tool emit per-pair text and are outside this path.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "genomic_intervals.h"
#include "gtx_cmdline.h"

static const char *PROGRAM = "genomic_overlaps";
static const long int BUFFER_SIZE = 10000;

int main(int argc, char *argv[])
{
  if (argc < 2) {
    fprintf(stderr, "\nUSAGE: \n  %s OPERATION [OPTIONS] REFERENCE-REGION-FILE <TEST-REGION-FILE>\n\nOPERATIONS (MI355X path): \n"
                    "  count      Counts the number of overlapping test regions per reference region.\n"
                    "  coverage   Calculates the depth coverage (total number of overlapping nucleotides) per reference region.\n"
                    "  density    Computes the density (coverage divided by the size of the reference region) per reference region.\n"
                    "  rpkm       Computing reference region RPKM values.\n\n", PROGRAM);
    return 1;
  }
  std::string op = argv[1];
  if (op[0] == '-') op = op.substr(1);                        // compatibility with the old "-count" spelling
  static const char *others[] = {"annotate", "bin", "dist", "intersect", "offset", "overlap", "subset"};
  for (const char *o : others) if (op == o) { fprintf(stderr, "Operation '%s' is outside the MI355X counting path of this build (count, coverage, density, rpkm)!\n", o); return 1; }
  if (op != "count" && op != "rpkm" && op != "coverage" && op != "density") { fprintf(stderr, "Unknown operation '%s'!\n", op.c_str()); return 1; }

  bool HELP, HELP2, VERBOSE, IS_SORTED, SORTED_BY_STRAND, IGNORE_STRAND, MATCH_GAPS;
  const char *BIN_BITS; long MAX_LABEL_VALUE; unsigned long MIN_COUNT = 0; double MIN_RPKM, MIN_DENSITY = 0.0;
  gtxhost::Options opts;
  opts.Flag("--help", &HELP, "help");
  opts.Flag("-h", &HELP2, "help");
  opts.Flag("-v", &VERBOSE, "verbose mode");
  opts.Str("-B", &BIN_BITS, "17,20,23,26", "number of shift-bits for each bin level (accepted, unused: no bin index on the device)");
  opts.Flag("-S", &IS_SORTED, "test and reference regions are sorted by chromosome and start position");
  opts.Flag("-s", &SORTED_BY_STRAND, "test and reference regions are also sorted by strand (-S must be set)");
  opts.Flag("-i", &IGNORE_STRAND, "ignore strand while finding overlaps");
  opts.Flag("-gaps", &MATCH_GAPS, "matching gaps between intervals are considered overlaps");
  opts.Long("--max-label-value", &MAX_LABEL_VALUE, 1, "maximum region label value to be used");
  if (op == "count") opts.ULong("-min", &MIN_COUNT, 0, "minimum count");
  else if (op == "coverage") opts.ULong("-min", &MIN_COUNT, 0, "minimum coverage");
  else if (op == "density") opts.Double("-min", &MIN_DENSITY, 0.0, "minimum density");
  else opts.Double("-min", &MIN_RPKM, 0.0, "minimum RPKM");
  long NGPU; opts.Long("--ngpu", &NGPU, 0, "MI355X: number of GPUs the reduction is spread over, by chromosome, RCCL reduce of the result (default: GTX_NGPU or 1)");
  int next_arg = opts.Parse(argc, argv, 2);
  if (NGPU > 0) GtxSetDevices((int)NGPU);
  if (HELP || HELP2 || argc - next_arg < 1) { opts.Usage(PROGRAM, op.c_str(), "[OPTIONS] REFERENCE-REGION-FILE <TEST-REGION-FILE>"); return 1; }
  _MESSAGES_ = VERBOSE;

  if (IS_SORTED && SORTED_BY_STRAND && IGNORE_STRAND) {
    fprintf(stderr, "[Error]: the input is sorted by chromosome/strand/start (i.e. -S and -s are set), therefore the overlap algorithm can only report strand-specific results (i.e. -i cannot be set)!\n");
    return 1;
  }

  char *REF_REG_FILE = argv[next_arg];
  char *TEST_REG_FILE = next_arg + 1 == argc ? NULL : argv[next_arg + 1];
  GenomicRegionSet RefRegSet(REF_REG_FILE, BUFFER_SIZE, VERBOSE, true, true);
  GenomicRegionSet TestRegSet(TEST_REG_FILE, BUFFER_SIZE, VERBOSE, false, true);

  GenomicRegionSetOverlaps *overlaps;
  if (IS_SORTED) overlaps = new SortedGenomicRegionSetOverlaps(&TestRegSet, &RefRegSet, SORTED_BY_STRAND);
  else overlaps = new UnsortedGenomicRegionSetOverlaps(&TestRegSet, &RefRegSet, BIN_BITS);
  unsigned long int *hits = (op == "coverage" || op == "density") ? overlaps->CalcIndexCoverage(MATCH_GAPS, IGNORE_STRAND, MAX_LABEL_VALUE)
                                                                  : overlaps->CountIndexOverlaps(MATCH_GAPS, IGNORE_STRAND, MAX_LABEL_VALUE);

  if (op == "density") {
    for (long int k = 0; k < RefRegSet.n_regions; k++) {
      long int size = (long int)RefRegSet.R[k]->GetSize(!MATCH_GAPS);
      volatile double density = (double)hits[k] / size;
      if (density >= MIN_DENSITY) printf("%s\t%.4e\n", RefRegSet.R[k]->LABEL, density);
    }
  } else if (op == "count" || op == "coverage") {
    for (long int k = 0; k < RefRegSet.n_regions; k++)
      if (hits[k] >= MIN_COUNT) printf("%s\t%lu\n", RefRegSet.R[k]->LABEL, hits[k]);
  } else {
    unsigned long int nreads = 0;
    for (long int k = 0; k < RefRegSet.n_regions; k++) nreads += hits[k];
    if (VERBOSE) fprintf(stderr, "* %lu reads overlap reference regions.\n", nreads);
    volatile double mreads = (double)nreads / 1000000;
    volatile double zero = 0.0;
    for (long int k = 0; k < RefRegSet.n_regions; k++) {
      long int eff_len = (long int)RefRegSet.R[k]->GetSize(!MATCH_GAPS);
      double rpkm = eff_len <= 0 ? zero / zero : (double)1000 * hits[k] / eff_len / mreads;   // the reference's -min is parsed but never applied
      printf("%s\t%.4e\n", RefRegSet.R[k]->LABEL, rpkm);
    }
  }
  GtxMark("output written");
  GtxFinish(0);
  delete[] hits;
  delete overlaps;
  return 0;
}
