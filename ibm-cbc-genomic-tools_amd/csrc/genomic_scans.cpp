// genomic_scans -- MI355X edition of GenomicTools' genomic_scans: `counts` and `peaks`
// (reference driver: gtools/genomic_scans.cpp:73-150 options, :399-436 RunCounts, :209-380 PeakFinder, :449-470).
// Sliding-window read counts over the chromosomes of a genome file; the histogram + window sums
// run on the GPU behind the reference's GenomicRegionSetScanner classes.  `peaks` scans a signal and a control
// read set the same way -- and, when given, a mappability track with the sorted scanner's operator 'p' -- and tests every
// window on the host (tail probabilities: gtx_stats.h, GSL is not linked); the -r reference filter of `counts` is a
// host-side test per reported window.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <list>
#include <random>
#include <string>
#include <vector>

#include "genomic_intervals.h"
#include "gtx_cmdline.h"
#include "gtx_stats.h"

static const char *PROGRAM = "genomic_scans";
static const long int BUFFER_SIZE = 10000;

// ---- peaks ----------------------------------------------------------------------------------------------
static bool P_VERBOSE, P_SORTED, P_IGNORE_STRAND, P_NORM, P_COMPARE, P_PRINT_DETAILS;
static const char *P_GENOME_REG_FILE, *P_METHOD;
static long P_MAX_LABEL_VALUE, P_WIN_SIZE, P_WIN_DIST, P_MIN_READS;
static double P_PVAL_CUTOFF, P_QVAL_CUTOFF;

// p-value threshold at which the estimated false discovery rate drops to the cutoff (genomic_scans.cpp:162-205):
// the control's p-values play the part of one permutation
static double ComputeQValues(const std::vector<double> &pval, const std::vector<double> &pval_rnd, long int n_permutations, double qval_cutoff)
{
  const long n = (long)pval.size();
  if (n == 0) return -1.0;
  std::vector<double> a(pval), b(pval_rnd);
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  std::vector<unsigned long> counts(n, 0);
  long k = 0;
  for (size_t i = 0, j = 0; i < a.size() && j < b.size(); j++) {
    while (i < a.size() && b[j] > a[i]) { i++; k++; }
    if (k < n - 1) counts[k]++;
  }
  std::vector<double> q(n);
  for (long c = 0; c < n; c++) {
    q[c] = (float)counts[c] / n_permutations / (c + 1);                        // single precision there
    if (c + 1 == n) break;
    counts[c + 1] += counts[c];
  }
  float min_q = (float)q[n - 1];
  long p = n - 1;                                                               // walks the sorted p-values from the top
  for (long c = n - 2; c >= 0; c--, p--) {
    if (min_q <= qval_cutoff) return a[p];
    if (q[c] > min_q) q[c] = min_q; else min_q = (float)q[c];
  }
  return -1.0;
}

static int RunPeaks(char *signal_reg_file, char *control_reg_file, char *uniq_reg_file)
{
  const char preprocess = (P_SORTED || uniq_reg_file != NULL) ? '1' : 'c';       // (genomic_scans.cpp:236-237)
  StringLIntMap *bounds = ReadBounds((char *)P_GENOME_REG_FILE, false);
  const unsigned long effective_genome_size = uniq_reg_file == NULL ? CalcBoundSize(bounds) : CalcRegSize(uniq_reg_file);
  fprintf(stderr, "* Effective genome size = %lu\n", effective_genome_size);

  auto make = [&](GenomicRegionSet *set) -> GenomicRegionSetScanner * {
    if (P_SORTED) return new SortedGenomicRegionSetScanner(set, bounds, P_WIN_DIST, P_WIN_SIZE, P_MAX_LABEL_VALUE, P_IGNORE_STRAND, preprocess);
    return new UnsortedGenomicRegionSetScanner(set, bounds, P_WIN_DIST, P_WIN_SIZE, P_MAX_LABEL_VALUE, P_IGNORE_STRAND, preprocess);
  };
  GenomicRegionSet SignalRegSet(signal_reg_file, BUFFER_SIZE, P_VERBOSE, false, true);
  GenomicRegionSetScanner *signal_scanner = make(&SignalRegSet);
  long n_signal_reads = signal_scanner->TotalLabelValue();                     // the whole signal scan runs here, on the GPU
  if (signal_scanner->InputErrorPending()) n_signal_reads = CountGenomicRegions(signal_reg_file, P_MAX_LABEL_VALUE);   // (the reference's own pass, :245)
  const double p_signal = (double)n_signal_reads / effective_genome_size;
  GenomicRegionSet *ControlRegSet = NULL;
  GenomicRegionSetScanner *control_scanner = NULL;
  long n_control_reads = n_signal_reads;
  double p_control = p_signal;
  if (control_reg_file != NULL) {
    ControlRegSet = new GenomicRegionSet(control_reg_file, BUFFER_SIZE, P_VERBOSE, false, true);
    control_scanner = make(ControlRegSet);
    n_control_reads = control_scanner->TotalLabelValue();
    if (control_scanner->InputErrorPending()) n_control_reads = CountGenomicRegions(control_reg_file, P_MAX_LABEL_VALUE);
    p_control = (double)n_control_reads / effective_genome_size;
  }
  const double p_ratio = p_signal / p_control;
  fprintf(stderr, "* Signal input file = %s (reads = %lu; background probability = %.2e)\n", signal_reg_file, n_signal_reads, p_signal);
  fprintf(stderr, "* Control input file = %s (reads = %lu; background probability = %.2e)\n", control_reg_file, n_control_reads, p_control);
  fprintf(stderr, "* Signal/Control background probability = %f\n", p_ratio);

  // the mappability track: always the sorted scanner, operator 'p', label values not used (genomic_scans.cpp:265-267)
  GenomicRegionSet *UniqRegSet = uniq_reg_file == NULL ? NULL : new GenomicRegionSet(uniq_reg_file, BUFFER_SIZE, P_VERBOSE, false, true);
  GenomicRegionSetScanner *uniq_scanner = uniq_reg_file == NULL ? NULL : new SortedGenomicRegionSetScanner(UniqRegSet, bounds, P_WIN_DIST, P_WIN_SIZE, 1, P_IGNORE_STRAND, 'p');

  // without a control the reference draws one Poisson number per window from its clock-seeded generator (:299);
  // here from a generator seeded the same way unless GTX_SEED fixes it
  std::mt19937_64 rng(getenv("GTX_SEED") ? strtoull(getenv("GTX_SEED"), NULL, 10) : (unsigned long long)(getpid() + time(NULL)));
  std::poisson_distribution<long> background(P_WIN_SIZE * p_signal);

  std::vector<double> pval1_list, pval2_list;
  std::vector<GenomicInterval *> interval_list;
  const std::string method = P_METHOD;
  long int v1, v2, v0;
  while ((v1 = signal_scanner->Next()) != -1) {
    v2 = control_scanner ? control_scanner->Next() : background(rng);
    v0 = uniq_scanner == NULL ? P_WIN_SIZE : uniq_scanner->Next();
    v1 = std::min(v1, v0);
    v2 = std::min(v2, v0);
    if (P_NORM) { if (p_ratio < 1.0) v2 = (long int)floor((float)v2 * p_ratio); else v1 = (long int)floor((float)v1 / p_ratio); }
    if (v1 < P_MIN_READS) continue;
    double pval1, pval2;
    if (P_COMPARE) {
      if (method == "binomial") {
        float pp_control = ((float)v2 + 1.0) / (v0 + 1.0);
        pval1 = gtxstats::BinomialQ(v1, std::max((double)pp_control, p_signal), v0 + 1);
        float pp_signal = ((float)v1 + 1.0) / (v0 + 1.0);
        pval2 = gtxstats::BinomialQ(v2, std::max((double)pp_signal, p_control), v0 + 1);
      } else if (method == "poisson") {
        const long pseudo = 5;
        pval1 = gtxstats::PoissonQ(v1 + pseudo, (double)(v2 + pseudo));
        pval2 = gtxstats::PoissonQ(v2 + pseudo, (double)(v1 + pseudo));
      } else if (method == "binomial2") {
        double pp_control = (double)(v2 + 1) / n_control_reads, pp_signal = (double)(v1 + 1) / n_signal_reads;
        pval1 = gtxstats::BinomialQ(v1 + 1, pp_control, n_signal_reads);
        pval2 = gtxstats::BinomialQ(v2 + 1, pp_signal, n_control_reads);
      } else if (method == "cbinomial") {
        pval1 = gtxstats::BinomialQ(v1 + 1, 0.5, v1 + v2 + 2);
        pval2 = gtxstats::BinomialQ(v2 + 1, 0.5, v1 + v2 + 2);
      } else if (method == "normal") {
        double pp_control = (double)(v2 + 1) / n_control_reads, pp_signal = (double)(v1 + 1) / n_signal_reads;
        pval1 = gtxstats::GaussianQ((v1 + 1 - n_signal_reads * pp_control) / sqrt(n_signal_reads * pp_control));
        pval2 = gtxstats::GaussianQ((v2 + 1 - n_control_reads * pp_signal) / sqrt(n_control_reads * pp_signal));
      } else { fprintf(stderr, "Error: unknown probability distribution!\n"); exit(1); }
    } else {
      if (method == "binomial") {
        pval1 = gtxstats::BinomialQ(v1, p_signal, v0 + 1);
        pval2 = gtxstats::BinomialQ(v2, p_control, v0 + 1);
      } else if (method == "poisson") {
        const long pseudo = 5;
        pval1 = gtxstats::PoissonQ(v1 + pseudo, (double)(v2 + pseudo));
        pval2 = gtxstats::PoissonQ(v2 + pseudo, (double)(v1 + pseudo));
      } else { fprintf(stderr, "Error: unknown probability distribution!\n"); exit(1); }
    }
    if (pval1 <= P_PVAL_CUTOFF) {
      interval_list.push_back(signal_scanner->GetInterval());
      pval1_list.push_back(pval1);
      pval2_list.push_back(pval2);
    }
  }
  const double pval_cutoff = ComputeQValues(pval1_list, pval2_list, 1, P_QVAL_CUTOFF);
  for (size_t k = 0; k < interval_list.size(); k++) {
    if (pval1_list[k] <= pval_cutoff) { printf("%.4e\t", pval1_list[k]); interval_list[k]->PrintInterval(); printf("\n"); }
    delete interval_list[k];
  }
  GtxFinish(0);
  delete signal_scanner; delete control_scanner; delete ControlRegSet; delete uniq_scanner; delete UniqRegSet; delete bounds;
  return 0;
}

int main(int argc, char *argv[])
{
  if (argc < 2) {
    fprintf(stderr, "\nUSAGE: \n  %s OPERATION [OPTIONS] INPUT-FILES\n\nOPERATIONS (MI355X path): \n"
                    "  counts     Determines input read counts in sliding windows of reference regions.\n"
                    "  peaks      Scans input reads to identify peaks.\n\n", PROGRAM);
    return 1;
  }
  std::string op = argv[1];
  if (op[0] == '-') op = op.substr(1);
  if (op == "peaks") {
    bool HELP, HELP2;
    gtxhost::Options o;
    o.Flag("--help", &HELP, "help");
    o.Flag("-h", &HELP2, "help");
    o.Flag("-v", &P_VERBOSE, "verbose mode");
    o.Flag("-S", &P_SORTED, "input regions are sorted");
    o.Str("-g", &P_GENOME_REG_FILE, "genome.reg+", "genome region file");
    o.Flag("-i", &P_IGNORE_STRAND, "ignore strand information");
    o.Long("--max-label-value", &P_MAX_LABEL_VALUE, 1, "maximum region label value to be used");
    o.Long("-w", &P_WIN_SIZE, 500, "window size (must be a multiple of window distance)");
    o.Long("-d", &P_WIN_DIST, 25, "window distance");
    o.Long("-min", &P_MIN_READS, 10, "minimum reads in window");
    o.Str("-M", &P_METHOD, "binomial", "method (binomial, poisson)");
    o.Flag("-norm", &P_NORM, "equalize background probabilities");
    o.Flag("-cmp", &P_COMPARE, "compare signal to control window");
    o.Double("-pval", &P_PVAL_CUTOFF, 1.0, "pvalue cutoff");
    o.Double("-qval", &P_QVAL_CUTOFF, 0.05, "qvalue cutoff");
    o.Flag("-D", &P_PRINT_DETAILS, "print details");
    long P_NGPU; o.Long("--ngpu", &P_NGPU, 0, "MI355X: number of GPUs the scans are spread over, by chromosome (default: GTX_NGPU or 1)");
    const int next_arg = o.Parse(argc, argv, 2);
    if (P_NGPU > 0) GtxSetDevices((int)P_NGPU);
    if (HELP || HELP2 || argc - next_arg < 1) { o.Usage(PROGRAM, "peaks", "[OPTIONS] SIGNAL-REG-FILE [CONTROL-REG-FILE [GENOME-UNIQ-REG-FILE]]"); return 1; }
    _MESSAGES_ = P_VERBOSE;
    return RunPeaks(argv[next_arg], next_arg + 1 < argc ? argv[next_arg + 1] : NULL, next_arg + 2 < argc ? argv[next_arg + 2] : NULL);
  }
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
  long NGPU; opts.Long("--ngpu", &NGPU, 0, "MI355X: number of GPUs the scan is spread over, by chromosome (default: GTX_NGPU or 1)");
  int next_arg = opts.Parse(argc, argv, 2);
  if (NGPU > 0) GtxSetDevices((int)NGPU);
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
  if (RefRegSet == NULL) scanner->PrintRemaining(stdout, (long int)MIN_READS);   // (no -r filter: the same lines, formatted in bulk)
  else
  for (long int v = REF_SORTED ? scanner->Next(RefRegSet) : scanner->Next(RefIndex); v != -1; v = REF_SORTED ? scanner->Next(RefRegSet) : scanner->Next(RefIndex)) {
    if (v >= MIN_READS) {
      printf("%ld\t", v);
      scanner->PrintInterval();
      printf("\n");
    }
  }
  GtxFinish(0);
  delete scanner;
  delete bounds;
  delete RefIndex;
  delete RefRegSet;
  return 0;
}
