// permutation_test -- MI355X edition of GenomicTools' category enrichment test
// (reference: gtools/permutation_test.cpp; same command line, same output).
//
// Host side: the table reader (StringSets constructor :120-215), the p-value -> FDR -> adjusted
// p-value arithmetic and the report (:742-812).  Device side (include/gtx_perm.h, libgtx.so): the
// statistics and the permutation loops -- StringSets::Calc*Statistic, RunPermutations and
// RunApproxPermutations are one call each into the C ABI.  There is no CPU implementation of those
// here: without a GPU the tool stops with an error.
//
// -a (p-values from a distribution, FDR from permutations of those): `-S n`, the form the reference's own example uses
// (examples/example06.tcsh), goes by a host-side table of the hypergeometric tail by (category size, k); ratio (without totals),
// t and corr evaluate their normal / Student tails on the device for every (category, permutation) -- the tails are defined in
// include/gtx_perm.h (GSL is not linked); for sum / sens / spec / ratio with totals the reference itself says "not implemented yet".
//
// The seed of the permutations is getpid()+time(NULL) as in the reference (:557) unless the
// environment variable GTX_PERM_SEED gives one; GTX_DEVICE picks the GPU.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "gtx.h"
#include "gtx_cmdline.h"
#include "gtx_perm.h"

static bool VERBOSE, DETAILS, HEADER, NORMALIZE, UNDER, APPROX, PRINT_FDR;
static long N_PERMUTATIONS, MIN_SUPPORT, MAX_SUPPORT;
static double QVAL_CUTOFF_ARG;
static const char *STATISTIC;

// ---- reference tokenizer rules (core.cpp:577-625) -------------------------------------------------
static char *NextToken(char **pbuf, char delim)
{
  char *b = *pbuf;
  while (*b == ' ') b++;
  char *e = b;
  while (*e != 0 && *e != delim) e++;
  if (*e == 0) *pbuf = e; else { *e = 0; *pbuf = e + 1; }
  return b;
}

static int CountTokens(const char *s, char delim)
{
  int n = 0;
  while (*s == ' ') s++;
  while (*s != 0) {
    while (*s != 0 && *s != delim) s++;
    if (*s == delim) s++;
    n++;
    while (*s == ' ') s++;
  }
  return n;
}

static std::vector<char> ReadWholeFile(const char *file, const char *who)
{
  FILE *f = fopen(file, "r");
  if (!f) { fprintf(stderr, "%s can't open file '%s'!\n", who, file); exit(1); }
  std::vector<char> buf;
  char tmp[1 << 16]; size_t got;
  while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  fclose(f);
  buf.push_back(0);
  return buf;
}

// ---- the table ------------------------------------------------------------------------------------
class StringSets
{
 public:
  StringSets(const char *file, const char *vec_file);
  ~StringSets() { gtx_perm_destroy(dev); }

  double *CalcStatistic(int stat);                                              // Calc*Statistic(approx = false)
  double *CalcHyperGeomApprox(const double *k_observed);                        // CalcHyperGeomStatistic(approx = true)
  double *RunPermutations(double *Y, long int n_permutations, int stat);
  double *CalcApprox(int stat, const double *observed);                         // Calc*Statistic(approx = true)
  double *RunApproxPermutations(double *Y, long int n_permutations, int stat);
  void PrintGOGenes(long int c) { for (int64_t z = col_ptr[c]; z < col_ptr[c + 1]; z++) printf("%s ", ROW_LABELS[rows[z]].c_str()); }

  long int n_rows, n_cols, n_values;
  std::vector<std::string> ROW_LABELS, COL_LABELS;
  std::vector<long int> COL_STATS;
  std::vector<int64_t> col_ptr;                  // B[c][1..] of all categories back to back
  std::vector<int32_t> rows;
  bool use_totals;
  std::vector<float> V, Vtotal;
  double Vsum, VsumZ, Vsum2, Vtotal_sum;

 private:
  void Die(int rc, const char *what) { fprintf(stderr, "Error: %s: %s\n", what, gtx_perm_last_error(dev)); (void)rc; exit(1); }
  void BuildHyperTable();
  gtx_perm *dev = nullptr;
  uint64_t seed;
  std::vector<int64_t> tab_ptr; std::vector<double> tab;
};

StringSets::StringSets(const char *file, const char *vec_file)
{
  std::vector<char> text = ReadWholeFile(file, "Error:");
  // complete lines only: a trailing piece without '\n' is not a line (core.cpp:241-259)
  std::vector<char *> lines;
  { char *p = text.data(); for (char *q = p; *q; q++) if (*q == '\n') { *q = 0; lines.push_back(p); p = q + 1; } }
  n_rows = (long)lines.size();
  n_values = 0;

  // pass 1: labels, keys and their support
  ROW_LABELS.resize(n_rows);
  std::vector<char *> value_str(n_rows, nullptr), keys_str(n_rows, nullptr);
  std::map<std::string, std::pair<long, long>> support;                          // key -> (column, occurrences)
  for (long r = 0; r < n_rows; r++) {
    char *inp = lines[r];
    ROW_LABELS[r] = NextToken(&inp, '\t');
    if (!vec_file) value_str[r] = NextToken(&inp, '\t');
    keys_str[r] = inp;
    std::string copy(inp);
    char *q = &copy[0];
    while (*q != 0) support[NextToken(&q, ' ')].second++;
  }
  n_cols = 0;
  if (MAX_SUPPORT == 0) MAX_SUPPORT = n_rows;
  for (auto &kv : support) kv.second.first = (kv.second.second >= MIN_SUPPORT && kv.second.second <= MAX_SUPPORT) ? n_cols++ : -1;
  if (VERBOSE) fprintf(stderr, "* Found %ld rows and %ld columns.\n", n_rows, n_cols);

  // pass 2: values and membership lists
  COL_LABELS.resize(n_cols); COL_STATS.assign(n_cols, 0); col_ptr.assign(n_cols + 1, 0);
  for (auto &kv : support) if (kv.second.first >= 0) { COL_LABELS[kv.second.first] = kv.first; COL_STATS[kv.second.first] = kv.second.second; col_ptr[kv.second.first + 1] = kv.second.second; }
  for (long c = 0; c < n_cols; c++) col_ptr[c + 1] += col_ptr[c];
  rows.resize(col_ptr[n_cols]);
  std::vector<int64_t> fill(col_ptr.begin(), col_ptr.end());
  V.assign(n_rows, 0.0f); Vtotal.assign(n_rows, 1.0f);
  for (long r = 0; r < n_rows; r++) {
    if (!vec_file) {
      char *v_str = value_str[r];
      const int n_tokens = CountTokens(v_str, ' ');
      if (n_tokens == 0 || n_tokens > 3) { fprintf(stderr, "Line %ld: 2nd column should contain 1 or 2 values!\n", r + 1); exit(1); }
      if (r == 0) n_values = n_tokens;
      else if (n_tokens != n_values) { fprintf(stderr, "Line %ld: expected %ld instead of %d tokens in 2nd column!\n", r + 1, n_values, n_tokens); exit(1); }
      V[r] = (float)atof(NextToken(&v_str, ' '));
      Vtotal[r] = n_values == 2 ? (float)atof(NextToken(&v_str, ' ')) : 1;
    }
    char *q = keys_str[r];
    while (*q != 0) {
      auto it = support.find(NextToken(&q, ' '));
      if (it != support.end() && it->second.first >= 0) rows[fill[it->second.first]++] = (int32_t)r;
    }
  }

  // values from a separate file (LoadMatrix, core.cpp:1913-1938): rows = '\n'-separated tokens
  if (vec_file) {
    std::vector<char> vt = ReadWholeFile(vec_file, "<LoadFile>:");
    const long n_vec_rows = CountTokens(vt.data(), '\n');
    char *p = vt.data();
    long n_vec_cols = 0;
    std::vector<char *> vl(n_vec_rows);
    for (long k = 0; k < n_vec_rows; k++) {
      vl[k] = NextToken(&p, '\n');
      const long n = CountTokens(vl[k], ' ');
      if (k == 0) n_vec_cols = n;
      else if (n != n_vec_cols) { fprintf(stderr, "Line %ld: number of columns (%ld) should be equal to %ld!\n%s\n", k + 1, n, n_vec_cols, vl[k]); exit(1); }
    }
    n_values = n_vec_cols;
    if (VERBOSE) fprintf(stderr, "* Found a %ldx%ld matrix.\n", n_vec_rows, n_values);
    if (n_vec_rows != n_rows || n_values > 2) { fprintf(stderr, "Wrong dimensions!\n"); exit(1); }
    auto number = [](char *s) { return strcasecmp(s, "nan") == 0 ? nanf("") : (float)atof(s); };
    for (long r = 0; r < n_rows; r++) {
      char *q = vl[r];
      V[r] = number(NextToken(&q, ' '));
      if (n_values == 2) Vtotal[r] = number(NextToken(&q, ' '));
    }
  }

  if (n_values == 2 && NORMALIZE) { use_totals = false; for (long r = 0; r < n_rows; r++) { V[r] /= Vtotal[r]; Vtotal[r] = 1; } }
  else use_totals = true;
  Vsum = VsumZ = Vsum2 = Vtotal_sum = 0;
  if (!use_totals) for (long r = 0; r < n_rows; r++) { Vsum += V[r]; Vsum2 += V[r] * V[r]; Vtotal_sum += Vtotal[r]; }
  else for (long r = 0; r < n_rows; r++) { Vsum += V[r]; VsumZ += V[r] / Vtotal[r]; Vsum2 += pow((double)(V[r] / Vtotal[r]), 2.0); Vtotal_sum += Vtotal[r]; }
  if (VERBOSE) fprintf(stderr, "* using normalized values = %s\n", use_totals ? "NO" : "YES");

  // hand the table to the GPU
  seed = getenv("GTX_PERM_SEED") ? strtoull(getenv("GTX_PERM_SEED"), NULL, 10) : (uint64_t)(getpid() + time(NULL));
  const int device = getenv("GTX_DEVICE") ? atoi(getenv("GTX_DEVICE")) : 0;
  if (gtx_perm_create(device, &dev) != GTX_OK) { fprintf(stderr, "Error: no usable MI355X device %d (this build has no CPU path)\n", device); exit(1); }
  if (n_rows < 1) { fprintf(stderr, "Error: input file has no rows\n"); exit(1); }
  bool all_one = true;
  for (long r = 0; r < n_rows && all_one; r++) all_one = Vtotal[r] == 1.0f;
  const double sums[4] = {Vsum, VsumZ, Vsum2, Vtotal_sum};
  int rc = gtx_perm_set_table(dev, n_rows, n_cols, col_ptr.data(), rows.data(), V.data(), all_one ? nullptr : Vtotal.data(), sums, use_totals ? GTX_PERM_USE_TOTALS : 0u);
  if (rc != GTX_OK) Die(rc, "gtx_perm_set_table");
}

static int StatId(const char *name)
{
  static const char *names[] = {"sum", "n", "sens", "spec", "ratio", "t", "corr"};
  for (int i = 0; i < 7; i++) if (!strcmp(name, names[i])) return i;
  return -1;
}

double *StringSets::CalcStatistic(int stat)
{
  if (stat == GTX_STAT_CORR && !use_totals) { fprintf(stderr, "Error: this operation is not permitted!\n"); exit(1); }
  double *Y = new double[n_cols + 1];
  int rc = gtx_perm_statistic(dev, stat, UNDER, Y);
  if (rc != GTX_OK) Die(rc, "gtx_perm_statistic");
  return Y;
}

// ---- hypergeometric upper tail (the value gsl_cdf_hypergeometric_Q(k, n1, n2, t) stands for) ------
// P(X > k) for X = positives among t draws without replacement from n1 members + n2 non-members.
// The point mass comes from log-gamma, neighbouring terms from their exact ratio; the tail away from
// the mean is summed directly (that is the small, significant one), the other is its complement.
static double LogChoose(double n, double k) { return lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0); }

static double HyperMass(long k, long n1, long n2, long t)
{
  if (k < 0 || k > n1 || k > t || t - k > n2) return 0.0;
  return exp(LogChoose((double)n1, (double)k) + LogChoose((double)n2, (double)(t - k)) - LogChoose((double)n1 + n2, (double)t));
}

static double HyperUpperTail(long k, long n1, long n2, long t)
{
  if (t > n1 + n2) return NAN;
  if (k >= n1 || k >= t) return 0.0;
  const double eps = 2.220446049250313e-16;
  if ((double)k >= (double)t * n1 / ((double)n1 + n2)) {
    long i = k + 1;
    double term = HyperMass(i, n1, n2, t), sum = term;
    for (; i < t && i < n1; i++) {
      term *= ((double)(n1 - i) / (i + 1.0)) * ((double)(t - i) / (n2 + i + 1.0 - t));
      sum += term;
      if (term / sum < eps) break;
    }
    return sum;
  }
  long i = k;
  double term = HyperMass(i, n1, n2, t), sum = term;
  for (; i > 0; i--) {
    term *= ((double)i / (n1 - i + 1.0)) * ((n2 + i - (double)t) / (t - i + 1.0));
    sum += term;
    if (term / sum < eps) break;
  }
  return 1.0 - sum;
}

void StringSets::BuildHyperTable()
{
  if (!tab_ptr.empty()) return;
  long t = 0;
  for (long r = 0; r < n_rows; r++) t += UNDER ? V[r] < 0 : V[r] > 0;
  tab_ptr.assign(n_cols + 1, 0);
  for (long c = 0; c < n_cols; c++) tab_ptr[c + 1] = tab_ptr[c] + COL_STATS[c] + 1;
  tab.resize(tab_ptr[n_cols] + 1);
  for (long c = 0; c < n_cols; c++) {
    const long n1 = COL_STATS[c];
    for (long k = 0; k <= n1; k++) tab[tab_ptr[c] + k] = k == 0 ? 1.0 : HyperUpperTail(k - 1, n1, n_rows - n1, t);   // :404-405
  }
}

double *StringSets::CalcHyperGeomApprox(const double *k_observed)
{
  BuildHyperTable();
  double *Y = new double[n_cols + 1];
  for (long c = 0; c < n_cols; c++) Y[c] = tab[tab_ptr[c] + (long)k_observed[c]];
  return Y;
}

double *StringSets::RunPermutations(double *Y, long int n_permutations, int stat)
{
  std::vector<uint64_t> counts(n_cols + 1, 0);
  int rc = gtx_perm_count_ge(dev, stat, UNDER, Y, seed, 0, n_permutations, counts.data());
  if (rc != GTX_OK) Die(rc, "gtx_perm_count_ge");
  double *pval = new double[n_cols + 1];
  for (long c = 0; c < n_cols; c++) pval[c] = (double)counts[c] / n_permutations;
  return pval;
}

double *StringSets::CalcApprox(int stat, const double *observed)
{
  if (stat == GTX_STAT_N) return CalcHyperGeomApprox(observed);
  double *P = new double[n_cols + 1];
  int rc = gtx_perm_statistic_approx(dev, stat, UNDER, P);
  if (rc != GTX_OK) Die(rc, "gtx_perm_statistic_approx");
  return P;
}

double *StringSets::RunApproxPermutations(double *Y, long int n_permutations, int stat)
{
  std::vector<uint64_t> hist(n_cols + 1, 0);
  int rc;
  if (stat == GTX_STAT_N) {
    BuildHyperTable();
    rc = gtx_perm_count_rank(dev, UNDER, tab_ptr.data(), tab.data(), Y, seed, 0, n_permutations, hist.data());
  } else rc = gtx_perm_count_rank_approx(dev, stat, UNDER, Y, seed, 0, n_permutations, hist.data());
  if (rc != GTX_OK) Die(rc, "gtx_perm_count_rank");
  // :630-636 (counts are int there)
  std::vector<int> counts(n_cols + 1, 0);
  for (long c = 0; c < n_cols; c++) counts[c] = (int)hist[c];
  double *FDR = new double[n_cols + 1];
  for (long k = 1, c = 0; c < n_cols; c++, k++) {
    FDR[c] = (double)counts[c] / n_permutations / k;
    if (c + 1 < n_cols) counts[c + 1] += counts[c];
  }
  double min_q = FDR[n_cols - 1];
  for (long c = n_cols - 1; c >= 0; c--) { if (FDR[c] > min_q) FDR[c] = min_q; else min_q = FDR[c]; }
  return FDR;
}

// VectorRank / VectorSort (core.cpp:1190-1259): qsort with a comparator that answers "greater ? 1 : -1";
// glibc's merge sort then keeps tied elements in their original order.
static void RankByValue(std::vector<int> &idx, const double *val)
{
  std::vector<int> tmp(idx.size());
  struct Rec {
    static void run(int *a, int *t, long n, const double *v)
    {
      if (n < 2) return;
      const long h = n / 2;
      run(a, t, h, v); run(a + h, t, n - h, v);
      long i = 0, j = h, o = 0;
      while (i < h && j < n) t[o++] = v[a[i]] > v[a[j]] ? a[j++] : a[i++];
      while (i < h) t[o++] = a[i++];
      while (j < n) t[o++] = a[j++];
      memcpy(a, t, sizeof(int) * (size_t)n);
    }
  };
  Rec::run(idx.data(), tmp.data(), (long)idx.size(), val);
}

int main(int argc, char *argv[])
{
  gtxhost::Options opts;
  opts.Flag("-v", &VERBOSE, "verbose mode");
  opts.Long("-kmin", &MIN_SUPPORT, 10, "minimum support per category");
  opts.Long("-kmax", &MAX_SUPPORT, 0, "maximum support per category (default = no maximum)");
  opts.Flag("-norm", &NORMALIZE, "normalize row values (if applicable)");
  opts.Str("-S", &STATISTIC, "sum", "choose statistic [sum|n|sens|spec|ratio|t|corr]");
  opts.Flag("-a", &APPROX, "use a distribution for p-value approximation (not applicable to all statistics)");
  opts.Flag("-u", &UNDER, "find depleted categories (default = enriched)");
  opts.Long("-p", &N_PERMUTATIONS, 100, "number of random permutations");
  opts.Double("-q", &QVAL_CUTOFF_ARG, 1.0, "FDR cutoff");
  opts.Flag("-f", &PRINT_FDR, "print FDR instead of adjusted p-values");
  opts.Flag("-h", &HEADER, "print header");
  opts.Flag("-d", &DETAILS, "print details");
  const int next_arg = opts.Parse(argc, argv, 1);
  if (argc - next_arg < 1) {
    opts.Usage("permutation_test", "[OPTIONS]", "vector(LABEL<tab>DATA<tab>CATEGORIES)\n  permutation_test [OPTIONS] vector(LABEL<tab>CATEGORIES) vector(DATA)");
    return 1;
  }
  const float QVAL_CUTOFF = (float)QVAL_CUTOFF_ARG;                               // a float option there (:42, :70)
  const char *MATRIX_FILE = argv[next_arg], *VECTOR_FILE = next_arg + 1 < argc ? argv[next_arg + 1] : NULL;

  const int stat = StatId(STATISTIC);
  StringSets INPUT(MATRIX_FILE, VECTOR_FILE);
  if (stat < 0) { fprintf(stderr, "Error: unknown statistic '%s'!\n", STATISTIC); return 1; }
  // -a where the reference has no distribution (:364, :387, :475, :502, :514: raised in the first category's turn)
  if (APPROX && INPUT.n_cols > 0 && (stat == GTX_STAT_SUM || stat == GTX_STAT_SENS || stat == GTX_STAT_SPEC || (stat == GTX_STAT_RATIO && INPUT.use_totals))) {
    fprintf(stderr, "Error: not implemented yet!\n");
    return 1;
  }

  const long n_cols = INPUT.n_cols;
  double *VAL = INPUT.CalcStatistic(stat);
  double *PVAL = APPROX ? INPUT.CalcApprox(stat, VAL) : INPUT.RunPermutations(VAL, N_PERMUTATIONS, stat);

  std::vector<int> R(n_cols);
  for (long c = 0; c < n_cols; c++) R[c] = (int)c;
  RankByValue(R, PVAL);
  std::vector<double> SORTED(n_cols + 1);
  for (long c = 0; c < n_cols; c++) SORTED[c] = PVAL[R[c]];

  double *FDR;
  if (n_cols == 0) FDR = new double[1];
  else if (APPROX) FDR = INPUT.RunApproxPermutations(SORTED.data(), N_PERMUTATIONS, stat);
  else {
    FDR = new double[n_cols];
    for (long k = 1, c = 0; c < n_cols; c++, k++) FDR[c] = SORTED[c] * n_cols / k;
    double min_q = FDR[n_cols - 1];
    for (long c = n_cols - 1; c >= 0; c--) { if (FDR[c] > min_q) FDR[c] = min_q; else min_q = FDR[c]; }
  }

  std::vector<double> QVAL(n_cols + 1);
  QVAL[0] = 0;
  for (long c = 1; c < n_cols; c++) { QVAL[c] = (c + 1) * FDR[c] - c * FDR[c - 1]; if (QVAL[c] < QVAL[c - 1]) QVAL[c] = QVAL[c - 1]; if (QVAL[c] > 1) QVAL[c] = 1; }

  if (HEADER) printf("CATEGORY\tCATEGORY-SIZE\tQ-VALUE\tP-VALUE\tSTATISTIC\n");
  for (long c = 0; c < n_cols; c++) {
    if (QVAL[c] > QVAL_CUTOFF) break;
    printf("%s\t%ld\t%.2e\t%.2e\t%f", INPUT.COL_LABELS[R[c]].c_str(), INPUT.COL_STATS[R[c]], PRINT_FDR ? FDR[c] : QVAL[c], SORTED[c], VAL[R[c]]);
    if (DETAILS) { printf("\t"); INPUT.PrintGOGenes(R[c]); }
    printf("\n");
  }
  // everything is written: skip the teardown of the HIP runtime unless a profiler needs the exit handlers
  fflush(stdout); fflush(stderr);
  {
    const char *pre = getenv("LD_PRELOAD");
    if (!(getenv("GTX_FULL_EXIT") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("HSA_TOOLS_LIB") || (pre && strstr(pre, "rocprof")))) _exit(0);
  }
  delete[] VAL; delete[] PVAL; delete[] FDR;
  return 0;
}
