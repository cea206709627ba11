/* perm_oracle.c -- CPU restatement of the reference's category permutation test
 * (gtools/permutation_test.cpp).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it; the product (libgtx.so, the permutation_test binary under
 * ibm-cbc-genomic-tools_amd/csrc) never does.
 *
 * PARITY UNPINNED for the random part.  The reference shuffles with GSL's default generator seeded
 * with getpid()+time(NULL) (permutation_test.cpp:557, core.cpp:4033-4039): its p-values are not
 * reproducible by design, it ships no expected output for this tool, and it cannot be built here
 * (GSL is not installed).  What this file pins instead:
 *   - the deterministic part (input parsing :120-205, the seven statistics :280-545, p-value ->
 *     FDR -> adjusted p-value arithmetic and output :778-812) is restated line by line;
 *   - the permutation source is OUR definition (include/gtx_perm.h: a keyed bijection of the row
 *     ids per permutation number), implemented here and in the HIP kernels identically, so that
 *     device and oracle agree bit for bit for the same seed;
 *   - a second source restates what the reference does -- cumulative gsl_ran_shuffle with MT19937
 *     (GSL's published algorithms, library version unpinned) -- so that tests can check that both
 *     sources estimate the same p-values within sampling error, and `-S n` against the exact
 *     hypergeometric tail.
 * The -a approximation of `-S n` needs gsl_cdf_hypergeometric_Q; porc_hypergeom_Q restates its
 * published algorithm (pmf from log-gamma, tail summed by the term ratio): tolerance parity only.
 * -a for ratio / t / corr needs gsl_cdf_ugaussian_Q / gsl_cdf_tdist_Q: porc_gauss_Q / porc_tdist_Q hold the
 * published definitions (erfc; the regularised incomplete beta function by its continued fraction), checked
 * against scipy in tests/test_perm_oracle.py: tolerance parity only, again.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

/* ---------------------------------------------------------------------------------------------
 * permutation source 1: keyed bijection (specification in include/gtx_perm.h)
 * ------------------------------------------------------------------------------------------- */
#define PERM_ROUNDS 10

static inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
static inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

typedef struct { uint32_t key[PERM_ROUNDS]; uint32_t fy[16]; } perm_keys;

static void perm_keys_init(perm_keys *k, uint64_t seed, uint64_t p)
{
  uint64_t s = mix64(mix64(seed + 0x9E3779B97F4A7C15ull) ^ ((p + 1) * 0xD6E8FEB86659FD93ull));
  for (int i = 0; i < PERM_ROUNDS; i += 2) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = mix64(s);
    k->key[i] = (uint32_t)z; k->key[i + 1] = (uint32_t)(z >> 32);
  }
  for (int i = 0; i < 16; i += 2) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = mix64(s);
    k->fy[i] = (uint32_t)z; k->fy[i + 1] = (uint32_t)(z >> 32);
  }
}

/* a = ceil(sqrt(n)), b = ceil(n / a): the a x b grid that holds [0, n) with the fewest spare cells */
static void perm_grid(uint32_t n, uint32_t *a, uint32_t *b)
{
  uint32_t r = (uint32_t)sqrt((double)n);
  while ((uint64_t)r * r < n) r++;
  while (r > 1 && (uint64_t)(r - 1) * (r - 1) >= n) r--;
  *a = r; *b = (n + r - 1) / r;
}

/* source row of permuted row r: Vperm[r] = V[perm_at(r)] */
static uint32_t perm_at(const perm_keys *k, uint32_t r, uint32_t n)
{
  if (n <= 16) {
    /* exact Fisher-Yates on a nibble-packed identity, randoms fy[i] scaled by multiply-high */
    uint64_t w = 0xFEDCBA9876543210ull;
    for (uint32_t i = n - 1; i > 0; i--) {
      uint32_t j = (uint32_t)(((uint64_t)k->fy[i] * (i + 1)) >> 32);
      uint64_t a = (w >> (4 * i)) & 15, b = (w >> (4 * j)) & 15;
      w ^= ((a ^ b) << (4 * i)) | ((a ^ b) << (4 * j));
    }
    return (uint32_t)((w >> (4 * r)) & 15);
  }
  uint32_t a, b; perm_grid(n, &a, &b);
  uint32_t L = r / b, R = r % b, x;
  do {
    for (int i = 0; i < PERM_ROUNDS; i += 2) {
      L += (uint32_t)(((uint64_t)fmix32(R + k->key[i]) * a) >> 32); if (L >= a) L -= a;
      R += (uint32_t)(((uint64_t)fmix32(L + k->key[i + 1]) * b) >> 32); if (R >= b) R -= b;
    }
    x = L * b + R;
  } while (x >= n);                                   /* cycle walking keeps the map a bijection of [0, n) */
  return x;
}

void porc_permutation(uint64_t seed, int64_t p, int64_t n, int32_t *out)
{
  perm_keys k; perm_keys_init(&k, seed, (uint64_t)p);
  for (int64_t r = 0; r < n; r++) out[r] = (int32_t)perm_at(&k, (uint32_t)r, (uint32_t)n);
}

/* uniformity check of source 1 (tests): chi-square statistics over P permutations of n rows --
 * out[0]: position x value table (df (n-1)^2), out[1]: joint of (pi(0), pi(1)) (df n(n-1)-1) */
void porc_permutation_chi2(uint64_t seed, int64_t P, int64_t n, double *out)
{
  int64_t *pos = calloc((size_t)(n * n), sizeof(int64_t)), *pair = calloc((size_t)(n * n), sizeof(int64_t));
  for (int64_t p = 0; p < P; p++) {
    perm_keys k; perm_keys_init(&k, seed, (uint64_t)p);
    uint32_t v0 = 0, v1 = 0;
    for (int64_t r = 0; r < n; r++) { uint32_t v = perm_at(&k, (uint32_t)r, (uint32_t)n); pos[r * n + v]++; if (r == 0) v0 = v; if (r == 1) v1 = v; }
    pair[v0 * n + v1]++;
  }
  double e = (double)P / n, c = 0;
  for (int64_t i = 0; i < n * n; i++) c += (pos[i] - e) * (pos[i] - e) / e;
  out[0] = c;
  e = (double)P / (n * (n - 1)); c = 0;
  for (int64_t i = 0; i < n; i++) for (int64_t j = 0; j < n; j++) if (i != j) c += (pair[i * n + j] - e) * (pair[i * n + j] - e) / e;
  out[1] = c;
  free(pos); free(pair);
}

/* ---------------------------------------------------------------------------------------------
 * permutation source 2: what the reference does -- gsl_ran_shuffle driven by gsl_rng_mt19937
 * (GSL rng/mt.c, rng/rng.c gsl_rng_uniform_int, randist/shuffle.c; restated from the published
 * algorithms, the library itself is not in this image)
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int mti; } mt_state;

static void mt_seed(mt_state *s, unsigned long seed)
{
  if (seed == 0) seed = 4357;
  s->mt[0] = (uint32_t)(seed & 0xffffffffUL);
  for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->mti = 624;
}

static uint32_t mt_get(mt_state *s)
{
  if (s->mti >= 624) {
    uint32_t *mt = s->mt; int kk;
    for (kk = 0; kk < 624 - 397; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
    for (; kk < 623; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
    uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    s->mti = 0;
  }
  uint32_t k = s->mt[s->mti++];
  k ^= (k >> 11); k ^= (k << 7) & 0x9d2c5680u; k ^= (k << 15) & 0xefc60000u; k ^= (k >> 18);
  return k;
}

static unsigned long mt_uniform_int(mt_state *s, unsigned long n)
{
  const unsigned long range = 0xffffffffUL, scale = range / n;
  unsigned long k;
  do { k = mt_get(s) / scale; } while (k >= n);
  return k;
}

/* x[0..n) shuffled in place (randist/shuffle.c) */
static void mt_shuffle_idx(mt_state *s, int32_t *x, long n)
{
  for (long i = n - 1; i > 0; i--) { long j = (long)mt_uniform_int(s, (unsigned long)(i + 1)); int32_t t = x[i]; x[i] = x[j]; x[j] = t; }
}

/* ---------------------------------------------------------------------------------------------
 * statistics (permutation_test.cpp:280-545).  Table = categories as CSR over the rows, in the
 * order the reference fills B[c][1..] (:176-184).  sums = {Vsum, VsumZ, Vsum2, Vtotal_sum} (:201-204).
 * ------------------------------------------------------------------------------------------- */
enum { STAT_SUM = 0, STAT_N, STAT_SENS, STAT_SPEC, STAT_RATIO, STAT_T, STAT_CORR };

/* The distribution functions of -a for ratio / t / corr: gsl_cdf_ugaussian_Q and gsl_cdf_tdist_Q (permutation_test.cpp:307, :338, :450,
 * :542).  GSL is an un-vendored dependency (gtools/Makefile:15, version unpinned) and absent here; what is restated is the published
 * definition of the two tails: Q(x) = erfc(x / sqrt 2) / 2, and for Student's t with nu degrees of freedom
 * Q(t) = I_x(nu/2, 1/2) / 2 for t >= 0 (1 minus that for t < 0), x = nu / (nu + t^2), with the regularised incomplete beta function
 * by its continued fraction (modified Lentz) on the side of (a + 1) / (a + b + 2) where it converges fast.  GSL evaluates the same
 * function by other routes (a Cornish-Fisher expansion for nu > 30): agreement with it is a matter of tolerance (~1e-10 relative),
 * not of bits -- parity unpinned, like everything of this tool that touches GSL. */
static double beta_cf(double a, double b, double x)
{
  const double tiny = 1e-300;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 100000; m++) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d; h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) <= 2.220446049250313e-16) break;
  }
  return h;
}
/* I_x(a, b) with y = 1 - x given by the caller (no cancellation near x = 1) */
static double ibeta_reg(double a, double b, double x, double y)
{
  if (!(x > 0.0)) return 0.0;
  if (!(y > 0.0)) return 1.0;
  const double lx = x < 0.5 ? log(x) : log1p(-y), ly = y < 0.5 ? log(y) : log1p(-x);
  const double pre = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * lx + b * ly);
  if (x < (a + 1.0) / (a + b + 2.0)) return pre * beta_cf(a, b, x) / a;
  return 1.0 - pre * beta_cf(b, a, y) / b;
}
double porc_tdist_Q(double t, double nu)
{
  if (t != t || !(nu > 0.0)) return NAN;
  if (isinf(t)) return t > 0 ? 0.0 : 1.0;
  const double t2 = t * t, den = nu + t2;
  const double tail = 0.5 * ibeta_reg(0.5 * nu, 0.5, nu / den, t2 / den);
  return t >= 0 ? tail : 1.0 - tail;
}
double porc_gauss_Q(double x) { return 0.5 * erfc(x / M_SQRT2); }

/* (long int)floor(x) of :306 / :337 followed by `df < 0 ? 1.0 : tdist_Q(Y, df)`: a NaN or an out-of-range quotient converts to
 * LONG_MIN on x86-64 (cvttsd2si), i.e. to the `1.0` branch */
static double welch_tail(double y, double s0, long n0, double s1, long n1)
{
  const double dfd = floor(pow(s0 + s1, 2.0) / (pow(s0, 2.0) / (n0 - 1) + pow(s1, 2.0) / (n1 - 1)));
  if (!(dfd >= 0.0) || dfd >= 9223372036854775808.0) return 1.0;
  return porc_tdist_Q(y, (double)(long)dfd);
}

static void statistic_impl(int approx, int stat, int under, int use_totals, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows,
                    const float *V, const float *Vt, const double *sums, double *Y)
{
  const double Vsum = sums[0], VsumZ = sums[1], Vsum2 = sums[2], Vtotal_sum = sums[3];
  long t = 0;
  if (stat == STAT_SPEC) for (int64_t r = 0; r < n_rows; r++) t += under ? V[r] < 0 : V[r] > 0;          /* :343-344 */
  for (int64_t c = 0; c < n_cols; c++) {
    const int32_t *B = rows + col_ptr[c];
    const long nc = (long)(col_ptr[c + 1] - col_ptr[c]);
    switch (stat) {
    case STAT_SUM:                                                                                       /* :487-520 */
      if (use_totals) {
        double y = 0, yt = 0;
        for (long z = 0; z < nc; z++) { y += V[B[z]]; yt += Vt[B[z]]; }
        y /= yt; Y[c] = under ? -y : y;
      } else {
        double y = 0;
        for (long z = 0; z < nc; z++) y += V[B[z]];
        y /= nc; Y[c] = under ? -y : y;
      }
      break;
    case STAT_N: case STAT_SENS: case STAT_SPEC: {                                                       /* :338-413 */
      long k = 0;
      for (long z = 0; z < nc; z++) k += under ? V[B[z]] < 0 : V[B[z]] > 0;
      Y[c] = stat == STAT_N ? (double)k : stat == STAT_SENS ? (double)k / nc : (double)k / t;
      break;
    }
    case STAT_RATIO: case STAT_T:                                                                        /* :281-331, :423-476 */
      if (!use_totals) {
        double mean[2] = {0, 0}, var[2] = {0, 0}; long n[2] = {0, 0};
        for (long z = 0; z < nc; z++) { int32_t r = B[z]; n[1]++; mean[1] += V[r]; var[1] += V[r] * V[r]; }   /* float product, as there */
        n[0] = n_rows - n[1]; mean[0] = Vsum - mean[1]; var[0] = Vsum2 - var[1];
        for (int k = 0; k <= 1; k++) { mean[k] /= n[k]; var[k] = var[k] / n[k] - mean[k] * mean[k]; }
        if (stat == STAT_RATIO) {
          Y[c] = under ? mean[0] / mean[1] : mean[1] / mean[0];
          if (approx) { double m = Vsum / n_rows, v = Vsum2 / n_rows; Y[c] = porc_gauss_Q((m * Y[c] - m) / sqrt(v * pow(Y[c], 2.0) + v)); }   /* :447-451 */
        }
        else {
          double y = (mean[1] - mean[0]) / sqrt(var[1] / n[1] + var[0] / n[0]); Y[c] = under ? -y : y;
          if (approx) Y[c] = welch_tail(Y[c], var[0] / n[0], n[0], var[1] / n[1], n[1]);                                                 /* :305-308 */
        }
      } else if (stat == STAT_RATIO) {
        long n1 = 0; double sum[2] = {0, 0}, total[2] = {0, 0}, mean[2];
        for (long z = 0; z < nc; z++) { int32_t r = B[z]; n1++; sum[1] += V[r]; total[1] += Vt[r]; }
        total[0] = Vtotal_sum - total[1]; sum[0] = Vsum - sum[1];
        for (int k = 0; k <= 1; k++) mean[k] = sum[k] / total[k];
        Y[c] = under ? mean[0] / mean[1] : mean[1] / mean[0];
      } else {
        long n[2] = {0, 0}; double sum[2] = {0, 0}, total[2] = {0, 0}, mean[2], sumZ[2] = {0, 0}, sumqZ[2] = {0, 0}, varZ[2];
        for (long z = 0; z < nc; z++) {
          int32_t r = B[z]; n[1]++; sum[1] += V[r]; total[1] += Vt[r]; sumZ[1] += V[r] / Vt[r];
          sumqZ[1] += pow((double)V[r] / Vt[r], 2.0);
        }
        n[0] = n_rows - n[1]; total[0] = Vtotal_sum - total[1]; sum[0] = Vsum - sum[1]; sumZ[0] = VsumZ - sumZ[1]; sumqZ[0] = Vsum2 - sumqZ[1];
        for (int k = 0; k <= 1; k++) { mean[k] = sum[k] / total[k]; varZ[k] = sumqZ[k] / n[k] - pow((double)sumZ[k] / n[k], 2.0); }
        double y = (mean[1] - mean[0]) / sqrt(varZ[1] / n[1] + varZ[0] / n[0]);
        Y[c] = under ? -y : y;
        if (approx) Y[c] = welch_tail(Y[c], varZ[0] / n[0], n[0], varZ[1] / n[1], n[1]);                                                 /* :336-339 */
      }
      break;
    case STAT_CORR: {                                                                                    /* :527-545, core.cpp:1535-1558 */
      double Ex = 0, Ey = 0, Ex2 = 0, Ey2 = 0, Exy = 0; unsigned long C = 0;
      for (long z = 0; z < nc; z++) {
        double a = V[B[z]], b = Vt[B[z]];
        if (a == a && b == b) { C++; Ex += a; Ex2 += pow(a, 2.0); Ey += b; Ey2 += pow(b, 2.0); Exy += a * b; }
      }
      Ex = Ex / C; Ey = Ey / C; Ex2 = Ex2 / C; Ey2 = Ey2 / C; Exy = Exy / C;
      double y = (Exy - Ex * Ey) / sqrt((Ex2 - pow(Ex, 2.0)) * (Ey2 - pow(Ey, 2.0)));
      y = fabs(y); Y[c] = under ? 1.0 - y : y;
      if (approx) Y[c] = porc_tdist_Q(Y[c] * sqrt((nc - 2) / (1 - pow(Y[c], 2.0))), (double)(nc - 2));                                  /* :542 */
      break;
    }
    }
  }
}

void porc_statistic(int stat, int under, int use_totals, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows,
                    const float *V, const float *Vt, const double *sums, double *Y)
{
  statistic_impl(0, stat, under, use_totals, n_rows, n_cols, col_ptr, rows, V, Vt, sums, Y);
}

/* Calc*Statistic(approx = true) for the statistics that have a distribution there: ratio without totals, t, corr */
void porc_statistic_approx(int stat, int under, int use_totals, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows,
                           const float *V, const float *Vt, const double *sums, double *P)
{
  statistic_impl(1, stat, under, use_totals, n_rows, n_cols, col_ptr, rows, V, Vt, sums, P);
}

/* hypergeometric upper tail P(X > k), X = successes in t draws from n1 successes + n2 failures
 * (what gsl_cdf_hypergeometric_Q(k, n1, n2, t) returns; cdf/hypergeometric.c: the tail on the
 * side of k away from the mean is summed term by term, the other one is its complement) */
static double ln_choose(double n, double k) { return lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1); }
static double hyper_pmf(long k, long n1, long n2, long t)
{
  if (k < 0 || k > n1 || k > t || t - k > n2) return 0;
  return exp(ln_choose(n1, k) + ln_choose(n2, t - k) - ln_choose(n1 + n2, t));
}
double porc_hypergeom_Q(long k, long n1, long n2, long t)
{
  if (t > n1 + n2) return NAN;
  if (k >= n1 || k >= t) return 0.0;
  const double midpoint = (double)t * n1 / ((double)n1 + n2);
  if ((double)k >= midpoint) {
    /* upper tail: sum_{i>k} pmf(i), ratio pmf(i+1)/pmf(i) = (n1-i)(t-i) / ((i+1)(n2-t+i+1)) */
    long i = k + 1; double p = hyper_pmf(i, n1, n2, t), s = p;
    while (i < t && i < n1) {
      p *= ((double)(n1 - i) / (i + 1.0)) * ((double)(t - i) / (n2 + i + 1.0 - t));
      s += p; i++;
      if (p / s < 2.220446049250313e-16) break;
    }
    return s;
  }
  /* lower tail: sum_{i<=k} pmf(i), downwards */
  long i = k; double p = hyper_pmf(i, n1, n2, t), s = p;
  while (i > 0) {
    p *= ((double)i / (n1 - i + 1.0)) * ((n2 + i - (double)t) / (t - i + 1.0));
    s += p; i--;
    if (p / s < 2.220446049250313e-16) break;
  }
  return 1.0 - s;
}

/* table of the approximate p-value of `-S n -a` by k for every category (:404-405):
 * tab[tab_ptr[c] + k], k = 0..n_c; t = rows with V > 0 (V < 0 with -u) */
void porc_hypergeom_table(int under, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const float *V, const int64_t *tab_ptr, double *tab)
{
  long t = 0;
  for (int64_t r = 0; r < n_rows; r++) t += under ? V[r] < 0 : V[r] > 0;
  for (int64_t c = 0; c < n_cols; c++) {
    long n1 = (long)(col_ptr[c + 1] - col_ptr[c]);
    for (long k = 0; k <= n1; k++) tab[tab_ptr[c] + k] = k == 0 ? 1.0 : porc_hypergeom_Q(k - 1, n1, (long)n_rows - n1, t);
  }
}

/* ---------------------------------------------------------------------------------------------
 * the permutation loops
 * ------------------------------------------------------------------------------------------- */
typedef struct { int use_mt; uint64_t seed; mt_state mt; int32_t *cur; } perm_src;

static void apply_perm(perm_src *ps, int64_t p, int64_t n_rows, const float *V, const float *Vt, float *Vp, float *Vtp, int32_t *idx)
{
  if (ps->use_mt) {
    /* cumulative: shuffle the current arrangement (:260-276) */
    mt_shuffle_idx(&ps->mt, ps->cur, n_rows);
    memcpy(idx, ps->cur, sizeof(int32_t) * (size_t)n_rows);
  } else porc_permutation(ps->seed, p, n_rows, idx);
  for (int64_t r = 0; r < n_rows; r++) { Vp[r] = V[idx[r]]; if (Vt) Vtp[r] = Vt[idx[r]]; }
}

/* counts[c] = #{p : Y_p[c] >= Y[c]} over permutations first_perm .. first_perm+n_perm-1 (:555-572).
 * source 0 = keyed bijection, 1 = MT19937 cumulative shuffles (seed = generator seed; first_perm ignored) */
void porc_count_ge(int stat, int under, int use_totals, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows,
                   const float *V, const float *Vt, const double *sums, const double *Y, int source, uint64_t seed,
                   int64_t first_perm, int64_t n_perm, uint64_t *counts)
{
  float *Vp = malloc(sizeof(float) * (size_t)(n_rows + 1)), *Vtp = malloc(sizeof(float) * (size_t)(n_rows + 1));
  int32_t *idx = malloc(sizeof(int32_t) * (size_t)(n_rows + 1));
  double *Yr = malloc(sizeof(double) * (size_t)(n_cols + 1));
  perm_src ps; ps.use_mt = source; ps.seed = seed; ps.cur = NULL;
  if (source) { mt_seed(&ps.mt, (unsigned long)seed); ps.cur = malloc(sizeof(int32_t) * (size_t)(n_rows + 1)); for (int64_t r = 0; r < n_rows; r++) ps.cur[r] = (int32_t)r; }
  for (int64_t c = 0; c < n_cols; c++) counts[c] = 0;
  for (int64_t p = 0; p < n_perm; p++) {
    apply_perm(&ps, first_perm + p, n_rows, V, Vt, Vp, Vtp, idx);
    porc_statistic(stat, under, use_totals, n_rows, n_cols, col_ptr, rows, Vp, Vt ? Vtp : NULL, sums, Yr);
    for (int64_t c = 0; c < n_cols; c++) counts[c] += (Yr[c] >= Y[c]);
  }
  free(Vp); free(Vtp); free(idx); free(Yr); free(ps.cur);
}

static int cmp_double_ref(const void *a, const void *b) { return *(const double *)a > *(const double *)b ? 1 : -1; }   /* core.cpp:1196 */

/* `-S n -a`: counts[z] = number of (permutation, category) pairs whose approximate p-value has its
 * lower bound in the sorted observed p-values at z (:612-627: the sort of Y_random only feeds a
 * two-pointer merge, whose result is this histogram) */
void porc_count_rank(int under, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows, const float *V,
                     const int64_t *tab_ptr, const double *tab, const double *sortedY, int source, uint64_t seed,
                     int64_t first_perm, int64_t n_perm, uint64_t *counts)
{
  float *Vp = malloc(sizeof(float) * (size_t)(n_rows + 1));
  int32_t *idx = malloc(sizeof(int32_t) * (size_t)(n_rows + 1));
  double *Yr = malloc(sizeof(double) * (size_t)(n_cols + 1));
  perm_src ps; ps.use_mt = source; ps.seed = seed; ps.cur = NULL;
  if (source) { mt_seed(&ps.mt, (unsigned long)seed); ps.cur = malloc(sizeof(int32_t) * (size_t)(n_rows + 1)); for (int64_t r = 0; r < n_rows; r++) ps.cur[r] = (int32_t)r; }
  for (int64_t c = 0; c < n_cols; c++) counts[c] = 0;
  for (int64_t p = 0; p < n_perm; p++) {
    apply_perm(&ps, first_perm + p, n_rows, V, NULL, Vp, NULL, idx);
    for (int64_t c = 0; c < n_cols; c++) {
      long k = 0;
      for (int64_t z = col_ptr[c]; z < col_ptr[c + 1]; z++) k += under ? Vp[rows[z]] < 0 : Vp[rows[z]] > 0;
      Yr[c] = tab[tab_ptr[c] + k];
    }
    qsort(Yr, (size_t)n_cols, sizeof(double), cmp_double_ref);
    for (int64_t z = 0, c = 0; z < n_cols && c < n_cols; c++) {
      while (z < n_cols && sortedY[z] < Yr[c]) z++;
      if (z < n_cols) counts[z]++;
    }
  }
  free(Vp); free(idx); free(Yr); free(ps.cur);
}

/* the same for ratio / t / corr with -a (:612-627): the approximate p-values of every permutation, sorted, merged into the observed ones */
void porc_count_rank_approx(int stat, int under, int use_totals, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows,
                            const float *V, const float *Vt, const double *sums, const double *sortedY, int source, uint64_t seed,
                            int64_t first_perm, int64_t n_perm, uint64_t *counts)
{
  float *Vp = malloc(sizeof(float) * (size_t)(n_rows + 1)), *Vtp = malloc(sizeof(float) * (size_t)(n_rows + 1));
  int32_t *idx = malloc(sizeof(int32_t) * (size_t)(n_rows + 1));
  double *Yr = malloc(sizeof(double) * (size_t)(n_cols + 1));
  perm_src ps; ps.use_mt = source; ps.seed = seed; ps.cur = NULL;
  if (source) { mt_seed(&ps.mt, (unsigned long)seed); ps.cur = malloc(sizeof(int32_t) * (size_t)(n_rows + 1)); for (int64_t r = 0; r < n_rows; r++) ps.cur[r] = (int32_t)r; }
  for (int64_t c = 0; c < n_cols; c++) counts[c] = 0;
  for (int64_t p = 0; p < n_perm; p++) {
    apply_perm(&ps, first_perm + p, n_rows, V, Vt, Vp, Vtp, idx);
    statistic_impl(1, stat, under, use_totals, n_rows, n_cols, col_ptr, rows, Vp, Vt ? Vtp : NULL, sums, Yr);
    qsort(Yr, (size_t)n_cols, sizeof(double), cmp_double_ref);
    for (int64_t z = 0, c = 0; z < n_cols && c < n_cols; c++) {
      while (z < n_cols && sortedY[z] < Yr[c]) z++;
      if (z < n_cols) counts[z]++;
    }
  }
  free(Vp); free(Vtp); free(idx); free(Yr); free(ps.cur);
}

/* ---------------------------------------------------------------------------------------------
 * the tool: input (:120-205), p-values -> FDR -> adjusted p-values -> output (:742-812)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  long n_rows, n_cols, n_values; int use_totals;
  char **row_labels, **col_labels; long *col_stats;
  int64_t *col_ptr; int32_t *rows;
  float *V, *Vt; double sums[4];
} perm_table;

static char *next_token_c(char **pbuf, char delim)                                   /* core.cpp:612-625 */
{
  char *b = *pbuf; while (b[0] == ' ') b++;
  int k = 0; while (b[k] != 0 && b[k] != delim) k++;
  if (b[k] == 0) *pbuf = b + k; else { b[k] = 0; *pbuf = b + k + 1; }
  return b;
}
static int count_tokens(const char *s, char delim)                                   /* core.cpp:577-593 */
{
  int k = 0, n = 0; while (s[k] == ' ') k++;
  for (;;) {
    if (s[k] == 0) return n;
    while (s[k] != 0 && s[k] != delim) k++;
    if (s[k] == delim) k++;
    n++;
    while (s[k] == ' ') k++;
    if (s[k] == 0) return n;
  }
}

typedef struct { char *key; long id, count; } key_ent;
static int cmp_key(const void *a, const void *b) { return strcmp(((const key_ent *)a)->key, ((const key_ent *)b)->key); }

/* complete lines ('\n'-terminated; a trailing fragment is not a line, core.cpp:241-259) of a text file */
static char **load_lines(const char *file, long *n_lines, char **storage)
{
  FILE *f = fopen(file, "r");
  if (!f) { fprintf(stderr, "Error: cannot open file '%s'!\n", file); exit(1); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); rewind(f);
  char *buf = malloc((size_t)sz + 1); sz = (long)fread(buf, 1, (size_t)sz, f); buf[sz] = 0; fclose(f);
  long n = 0; for (long i = 0; i < sz; i++) n += buf[i] == '\n';
  char **lines = malloc(sizeof(char *) * (size_t)(n + 1));
  long k = 0; char *p = buf;
  for (long i = 0; i < sz; i++) if (buf[i] == '\n') { buf[i] = 0; lines[k++] = p; p = buf + i + 1; }
  *n_lines = n; *storage = buf;
  return lines;
}

static void table_load(perm_table *T, const char *file, const char *vec_file, long min_support, long max_support, int normalize, int verbose)
{
  char *store; long n_rows; char **lines = load_lines(file, &n_rows, &store);
  T->n_rows = n_rows; T->n_values = 0;
  T->row_labels = malloc(sizeof(char *) * (size_t)(n_rows + 1));
  /* pass 1 (:131-144): keys and their support */
  size_t cap = 1024, nk = 0; key_ent *keys = malloc(sizeof(key_ent) * cap);
  /* a sorted array + bsearch would need re-sorting on insert; collect all occurrences, sort, run-length */
  size_t ocap = 4096, no = 0; char **occ = malloc(sizeof(char *) * ocap);
  char **rest = malloc(sizeof(char *) * (size_t)(n_rows + 1)), **vstr = malloc(sizeof(char *) * (size_t)(n_rows + 1));
  for (long r = 0; r < n_rows; r++) {
    char *inp = lines[r];
    T->row_labels[r] = next_token_c(&inp, '\t');
    vstr[r] = vec_file ? NULL : next_token_c(&inp, '\t');
    rest[r] = inp;
    char *scan = strdup(inp), *q = scan;
    while (q[0] != 0) { char *key = next_token_c(&q, ' '); if (no == ocap) { ocap *= 2; occ = realloc(occ, sizeof(char *) * ocap); } occ[no++] = strdup(key); }
    free(scan);
  }
  key_ent *all = malloc(sizeof(key_ent) * (no + 1));
  for (size_t i = 0; i < no; i++) { all[i].key = occ[i]; all[i].id = 0; all[i].count = 1; }
  qsort(all, no, sizeof(key_ent), cmp_key);                         /* std::map order = byte-wise string order */
  for (size_t i = 0; i < no;) {
    size_t j = i; while (j < no && !strcmp(all[j].key, all[i].key)) j++;
    if (nk == cap) { cap *= 2; keys = realloc(keys, sizeof(key_ent) * cap); }
    keys[nk].key = all[i].key; keys[nk].count = (long)(j - i); keys[nk].id = -1; nk++;
    i = j;
  }
  /* support filter (:147-151) */
  if (max_support == 0) max_support = n_rows;
  long n_cols = 0;
  for (size_t i = 0; i < nk; i++) if (keys[i].count >= min_support && keys[i].count <= max_support) keys[i].id = n_cols++;
  T->n_cols = n_cols;
  if (verbose) fprintf(stderr, "* Found %ld rows and %ld columns.\n", n_rows, n_cols);
  /* pass 2 (:154-186) */
  T->col_labels = calloc((size_t)n_cols + 1, sizeof(char *)); T->col_stats = calloc((size_t)n_cols + 1, sizeof(long));
  T->col_ptr = calloc((size_t)n_cols + 2, sizeof(int64_t));
  for (size_t i = 0; i < nk; i++) if (keys[i].id >= 0) { T->col_ptr[keys[i].id + 1] = keys[i].count; T->col_labels[keys[i].id] = keys[i].key; T->col_stats[keys[i].id] = keys[i].count; }
  for (long c = 0; c < n_cols; c++) T->col_ptr[c + 1] += T->col_ptr[c];
  T->rows = malloc(sizeof(int32_t) * (size_t)(T->col_ptr[n_cols] + 1));
  int64_t *fill = malloc(sizeof(int64_t) * (size_t)(n_cols + 1)); memcpy(fill, T->col_ptr, sizeof(int64_t) * (size_t)(n_cols + 1));
  T->V = malloc(sizeof(float) * (size_t)(n_rows + 1)); T->Vt = malloc(sizeof(float) * (size_t)(n_rows + 1));
  for (long r = 0; r < n_rows; r++) {
    if (!vec_file) {
      char *v = vstr[r];
      int nt = count_tokens(v, ' ');
      if (nt == 0 || nt > 3) { fprintf(stderr, "Line %ld: 2nd column should contain 1 or 2 values!\n", r + 1); exit(1); }
      if (r == 0) T->n_values = nt;
      else if (nt != T->n_values) { fprintf(stderr, "Line %ld: expected %ld instead of %d tokens in 2nd column!\n", r + 1, T->n_values, nt); exit(1); }
      T->V[r] = (float)atof(next_token_c(&v, ' '));
      T->Vt[r] = T->n_values == 2 ? (float)atof(next_token_c(&v, ' ')) : 1;
    }
    char *q = rest[r];
    while (q[0] != 0) {
      char *key = next_token_c(&q, ' ');
      key_ent probe; probe.key = key;
      key_ent *hit = bsearch(&probe, keys, nk, sizeof(key_ent), cmp_key);
      if (hit && hit->id >= 0) T->rows[fill[hit->id]++] = (int32_t)r;
    }
  }
  if (vec_file) {                                                    /* :189-198, core.cpp:1913-1938 */
    /* LoadMatrix: rows are the '\n'-separated tokens of the whole file (a last line without '\n' is a row too) */
    FILE *vf = fopen(vec_file, "r");
    if (!vf) { fprintf(stderr, "<LoadFile>: can't open file '%s'!\n", vec_file); exit(1); }
    fseek(vf, 0, SEEK_END); long vsz = ftell(vf); rewind(vf);
    char *vstore = malloc((size_t)vsz + 1); vsz = (long)fread(vstore, 1, (size_t)vsz, vf); vstore[vsz] = 0; fclose(vf);
    long nv = count_tokens(vstore, '\n');
    char **vl = malloc(sizeof(char *) * (size_t)(nv + 1)), *vp = vstore;
    for (long k = 0; k < nv; k++) vl[k] = next_token_c(&vp, '\n');
    long ncol = 0;
    for (long k = 0; k < nv; k++) { int n = count_tokens(vl[k], ' '); if (k == 0) ncol = n; else if (n != ncol) { fprintf(stderr, "Line %ld: number of columns (%d) should be equal to %ld!\n%s\n", k + 1, n, ncol, vl[k]); exit(1); } }
    T->n_values = ncol;
    if (verbose) fprintf(stderr, "* Found a %ldx%ld matrix.\n", nv, ncol);
    if (nv != n_rows || ncol > 2) { fprintf(stderr, "Wrong dimensions!\n"); exit(1); }
    for (long r = 0; r < n_rows; r++) {
      char *v = vl[r];
      char *a = next_token_c(&v, ' ');
      T->V[r] = !strcasecmp(a, "nan") ? NAN : (float)atof(a);
      T->Vt[r] = 1;                                                  /* (uninitialised there when the file has one column) */
      if (ncol == 2) { char *b = next_token_c(&v, ' '); T->Vt[r] = !strcasecmp(b, "nan") ? NAN : (float)atof(b); }
    }
    free(vl); free(vstore);
  }
  /* normalisation and the permutation-invariant sums (:201-215) */
  if (T->n_values == 2 && normalize) { T->use_totals = 0; for (long r = 0; r < n_rows; r++) { T->V[r] /= T->Vt[r]; T->Vt[r] = 1; } }
  else T->use_totals = 1;
  double Vsum = 0, VsumZ = 0, Vsum2 = 0, Vtotal_sum = 0;
  if (!T->use_totals) for (long r = 0; r < n_rows; r++) { Vsum += T->V[r]; Vsum2 += T->V[r] * T->V[r]; Vtotal_sum += T->Vt[r]; }
  else for (long r = 0; r < n_rows; r++) { Vsum += T->V[r]; VsumZ += T->V[r] / T->Vt[r]; Vsum2 += pow((double)(T->V[r] / T->Vt[r]), 2.0); Vtotal_sum += T->Vt[r]; }
  T->sums[0] = Vsum; T->sums[1] = VsumZ; T->sums[2] = Vsum2; T->sums[3] = Vtotal_sum;
  if (verbose) fprintf(stderr, "* using normalized values = %s\n", T->use_totals ? "NO" : "YES");
  free(fill); free(rest); free(vstr); free(all); free(occ); free(lines); (void)store;
}

/* glibc qsort with the reference's comparator (1 / -1, never 0) is a merge sort that takes the left
 * element unless it compares greater: ties keep their order */
static void merge_sort_idx(int *idx, int *tmp, long n, const double *val)
{
  if (n < 2) return;
  long n1 = n / 2, n2 = n - n1;
  merge_sort_idx(idx, tmp, n1, val); merge_sort_idx(idx + n1, tmp, n2, val);
  long a = 0, b = n1, o = 0;
  while (a < n1 && b < n) tmp[o++] = val[idx[a]] > val[idx[b]] ? idx[b++] : idx[a++];
  while (a < n1) tmp[o++] = idx[a++];
  while (b < n) tmp[o++] = idx[b++];
  memcpy(idx, tmp, sizeof(int) * (size_t)n);
}

int main(int argc, char **argv)
{
  int verbose = 0, normalize = 0, approx = 0, under = 0, print_fdr = 0, header = 0, details = 0;
  long kmin = 10, kmax = 0, n_perm = 100; float qcut = 1.0f; const char *statistic = "sum";
  int a = 1;
  for (; a < argc && argv[a][0] == '-'; a++) {                                        /* core.cpp:2420-2436 */
    const char *o = argv[a];
#define NEEDVAL() do { if (a + 1 >= argc) { fprintf(stderr, "Error: could not set option '%s'!\n", o); return 1; } } while (0)
    if (!strcmp(o, "-v")) verbose = 1;
    else if (!strcmp(o, "-norm")) normalize = 1;
    else if (!strcmp(o, "-a")) approx = 1;
    else if (!strcmp(o, "-u")) under = 1;
    else if (!strcmp(o, "-f")) print_fdr = 1;
    else if (!strcmp(o, "-h")) header = 1;
    else if (!strcmp(o, "-d")) details = 1;
    else if (!strcmp(o, "-kmin")) { NEEDVAL(); kmin = atol(argv[++a]); }
    else if (!strcmp(o, "-kmax")) { NEEDVAL(); kmax = atol(argv[++a]); }
    else if (!strcmp(o, "-p")) { NEEDVAL(); n_perm = atol(argv[++a]); }
    else if (!strcmp(o, "-q")) { NEEDVAL(); qcut = (float)atof(argv[++a]); }
    else if (!strcmp(o, "-S")) { NEEDVAL(); statistic = argv[++a]; }
    else { fprintf(stderr, "Error: unknown option '%s'!\n", o); return 1; }
  }
  if (argc - a < 1) { fprintf(stderr, "\nUSAGE: \n  permutation_test [OPTIONS] vector(LABEL<tab>DATA<tab>CATEGORIES)\n"); return 1; }
  const char *file = argv[a++], *vec_file = a < argc ? argv[a] : NULL;
  /* our additions, through the environment so that the command line stays the reference's:
   * GTX_PERM_SEED (default getpid()+time(NULL), :557), GTX_PERM_SOURCE=mt for the MT19937 shuffles */
  uint64_t seed = getenv("GTX_PERM_SEED") ? strtoull(getenv("GTX_PERM_SEED"), NULL, 10) : (uint64_t)(getpid() + time(NULL));
  int source = getenv("GTX_PERM_SOURCE") && !strcmp(getenv("GTX_PERM_SOURCE"), "mt");

  perm_table T; table_load(&T, file, vec_file, kmin, kmax, normalize, verbose);
  int stat;
  if (!strcmp(statistic, "sum")) stat = STAT_SUM; else if (!strcmp(statistic, "n")) stat = STAT_N; else if (!strcmp(statistic, "sens")) stat = STAT_SENS;
  else if (!strcmp(statistic, "spec")) stat = STAT_SPEC; else if (!strcmp(statistic, "ratio")) stat = STAT_RATIO; else if (!strcmp(statistic, "t")) stat = STAT_T;
  else if (!strcmp(statistic, "corr")) stat = STAT_CORR;
  else { fprintf(stderr, "Error: unknown statistic '%s'!\n", statistic); return 1; }
  if (stat == STAT_CORR && !T.use_totals) { fprintf(stderr, "Error: this operation is not permitted!\n"); return 1; }
  /* -a where the reference has no distribution (:364, :387, :475, :502, :514: raised inside the first category's turn) */
  if (approx && T.n_cols > 0 && (stat == STAT_SUM || stat == STAT_SENS || stat == STAT_SPEC || (stat == STAT_RATIO && T.use_totals))) { fprintf(stderr, "Error: not implemented yet!\n"); return 1; }
  const int by_table = approx && stat == STAT_N;

  const long nc = T.n_cols;
  double *VAL = malloc(sizeof(double) * (size_t)(nc + 1)), *PVAL = malloc(sizeof(double) * (size_t)(nc + 1));
  const float *Vt = T.Vt;
  porc_statistic(stat, under, T.use_totals, T.n_rows, nc, T.col_ptr, T.rows, T.V, Vt, T.sums, VAL);
  int64_t *tab_ptr = NULL; double *tab = NULL;
  if (approx && !by_table) porc_statistic_approx(stat, under, T.use_totals, T.n_rows, nc, T.col_ptr, T.rows, T.V, Vt, T.sums, PVAL);
  else if (approx) {
    tab_ptr = malloc(sizeof(int64_t) * (size_t)(nc + 1)); tab_ptr[0] = 0;
    for (long c = 0; c < nc; c++) tab_ptr[c + 1] = tab_ptr[c] + (T.col_ptr[c + 1] - T.col_ptr[c]) + 1;
    tab = malloc(sizeof(double) * (size_t)(tab_ptr[nc] + 1));
    porc_hypergeom_table(under, T.n_rows, nc, T.col_ptr, T.V, tab_ptr, tab);
    for (long c = 0; c < nc; c++) PVAL[c] = tab[tab_ptr[c] + (long)VAL[c]];
  } else {
    uint64_t *counts = malloc(sizeof(uint64_t) * (size_t)(nc + 1));
    porc_count_ge(stat, under, T.use_totals, T.n_rows, nc, T.col_ptr, T.rows, T.V, Vt, T.sums, VAL, source, seed, 0, n_perm, counts);
    for (long c = 0; c < nc; c++) PVAL[c] = (double)counts[c] / n_perm;
    free(counts);
  }
  /* rank + sort (:778-779) */
  int *R = malloc(sizeof(int) * (size_t)(nc + 1)), *tmp = malloc(sizeof(int) * (size_t)(nc + 1));
  for (long c = 0; c < nc; c++) R[c] = (int)c;
  merge_sort_idx(R, tmp, nc, PVAL);
  double *SP = malloc(sizeof(double) * (size_t)(nc + 1));
  for (long c = 0; c < nc; c++) SP[c] = PVAL[R[c]];
  double *FDR = malloc(sizeof(double) * (size_t)(nc + 1));
  if (nc > 0) {
    if (approx) {                                                                     /* :612-640 */
      uint64_t *counts = malloc(sizeof(uint64_t) * (size_t)(nc + 1));
      if (by_table) porc_count_rank(under, T.n_rows, nc, T.col_ptr, T.rows, T.V, tab_ptr, tab, SP, source, seed, 0, n_perm, counts);
      else porc_count_rank_approx(stat, under, T.use_totals, T.n_rows, nc, T.col_ptr, T.rows, T.V, Vt, T.sums, SP, source, seed, 0, n_perm, counts);
      for (long k = 1, c = 0; c < nc; c++, k++) { FDR[c] = (double)(int)counts[c] / n_perm / k; if (c + 1 < nc) counts[c + 1] += counts[c]; }
      free(counts);
    } else for (long k = 1, c = 0; c < nc; c++, k++) FDR[c] = SP[c] * nc / k;        /* :789-790 */
    double min_q = FDR[nc - 1];
    for (long c = nc - 1; c >= 0; c--) { if (FDR[c] > min_q) FDR[c] = min_q; else min_q = FDR[c]; }
  }
  double *QVAL = malloc(sizeof(double) * (size_t)(nc + 1));
  QVAL[0] = 0;
  for (long c = 1; c < nc; c++) { QVAL[c] = (c + 1) * FDR[c] - c * FDR[c - 1]; if (QVAL[c] < QVAL[c - 1]) QVAL[c] = QVAL[c - 1]; if (QVAL[c] > 1) QVAL[c] = 1; }
  if (header) printf("CATEGORY\tCATEGORY-SIZE\tQ-VALUE\tP-VALUE\tSTATISTIC\n");
  for (long c = 0; c < nc; c++) {
    if (QVAL[c] > qcut) break;
    printf("%s\t%ld\t%.2e\t%.2e\t%f", T.col_labels[R[c]], T.col_stats[R[c]], print_fdr ? FDR[c] : QVAL[c], SP[c], VAL[R[c]]);
    if (details) { printf("\t"); for (int64_t z = T.col_ptr[R[c]]; z < T.col_ptr[R[c] + 1]; z++) printf("%s ", T.row_labels[T.rows[z]]); }
    printf("\n");
  }
  return 0;
}
