// gtx_stats.h -- the three tail probabilities `genomic_scans peaks` needs (reference: GSL's
// gsl_cdf_binomial_Q, gsl_cdf_poisson_Q, gsl_cdf_ugaussian_Q at genomic_scans.cpp:317-356; GSL is not linked).
// Discrete tails are summed term by term from the point mass (saddle-point form) with the exact ratio of neighbouring
// terms, on the side away from the mean; the other side is the complement.  Relative accuracy ~1e-13, i.e.
// the same five digits the tool prints (%.4e) except at rounding boundaries.
#pragma once
#include <math.h>

namespace gtxstats {

static const double kEps = 2.220446049250313e-16;

// Saddle-point form of the point masses (C. Loader, "Fast and accurate computation of binomial probabilities",
// 2000): the large log-gamma terms are never formed, so the relative error stays ~1e-15 for any n.
inline double StirlingError(double n)                  // lgamma(n+1) - [(n + 1/2) log n - n + log(2 pi)/2]
{
  if (n < 16.0) return lgamma(n + 1.0) - ((n + 0.5) * log(n) - n + 0.918938533204672741780329736406);
  const double n2 = n * n;
  return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0 - (1.0 / 1680.0 - (1.0 / 1188.0) / n2) / n2) / n2) / n2) / n;
}

inline double Deviance(double x, double np)             // x log(x / np) + np - x, without cancellation
{
  if (fabs(x - np) < 0.1 * (x + np)) {
    double v = (x - np) / (x + np), s = (x - np) * v, ej = 2.0 * x * v;
    v = v * v;
    for (int j = 1; j < 1000; j++) {
      ej *= v;
      const double s1 = s + ej / (2 * j + 1);
      if (s1 == s) return s1;
      s = s1;
    }
    return s;
  }
  return x * log(x / np) + np - x;
}

inline double BinomialMass(long k, long n, double p)    // 0 < p < 1
{
  const double q = 1.0 - p;
  if (k == 0) return exp(n * log1p(-p));
  if (k == n) return exp(n * log(p));
  const double lc = StirlingError((double)n) - StirlingError((double)k) - StirlingError((double)(n - k)) - Deviance((double)k, n * p) - Deviance((double)(n - k), n * q);
  const double lf = 1.837877066409345483560659472811 + log((double)k) + log1p(-(double)k / n);
  return exp(lc - 0.5 * lf);
}

inline double PoissonMass(long k, double mu)            // mu > 0
{
  if (k == 0) return exp(-mu);
  return exp(-StirlingError((double)k) - Deviance((double)k, mu)) / sqrt(6.283185307179586476925286766559 * k);
}

// P(X > k), X ~ Binomial(n, p)
inline double BinomialQ(long k, double p, long n)
{
  if (p < 0.0 || p > 1.0 || n < 0) return NAN;
  if (k < 0) return 1.0;
  if (k >= n) return 0.0;
  if (p == 0.0) return 0.0;
  if (p == 1.0) return 1.0;
  const double odds = p / (1.0 - p);
  auto mass = [&](long i) { return BinomialMass(i, n, p); };
  if ((double)k + 1.0 > n * p) {                         // upper tail directly
    long i = k + 1;
    double term = mass(i), sum = term;
    for (; i < n; i++) {
      term *= (double)(n - i) / (i + 1.0) * odds;
      sum += term;
      if (term < sum * kEps) break;
    }
    return sum;
  }
  long i = k;                                            // lower tail, downwards
  double term = mass(i), sum = term;
  for (; i > 0; i--) {
    term *= (double)i / (n - i + 1.0) / odds;
    sum += term;
    if (term < sum * kEps) break;
  }
  return 1.0 - sum;
}

// P(X > k), X ~ Poisson(mu)
inline double PoissonQ(long k, double mu)
{
  if (mu < 0.0) return NAN;
  if (k < 0) return 1.0;
  if (mu == 0.0) return 0.0;
  auto mass = [&](long i) { return PoissonMass(i, mu); };
  if ((double)k + 1.0 > mu) {
    long i = k + 1;
    double term = mass(i), sum = term;
    for (;; i++) {
      term *= mu / (i + 1.0);
      sum += term;
      if (term < sum * kEps) break;
    }
    return sum;
  }
  long i = k;
  double term = mass(i), sum = term;
  for (; i > 0; i--) {
    term *= (double)i / mu;
    sum += term;
    if (term < sum * kEps) break;
  }
  return 1.0 - sum;
}

// P(Z > x), Z ~ N(0, 1)
inline double GaussianQ(double x) { return 0.5 * erfc(x / M_SQRT2); }

}  // namespace gtxstats
