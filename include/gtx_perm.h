/* gtx_perm.h -- C ABI of the category permutation test on the MI355X (part of libgtx.so).
 *
 * Replaces the inner loops of the reference's gtools/permutation_test.cpp:
 *   StringSets::Permute            :260-276   (shuffle of the per-row values)
 *   StringSets::Calc*Statistic     :280-545   (per-category statistic over the membership lists B[c][*])
 *   StringSets::RunPermutations    :555-572   (P x { permute, statistic, pval[c] += Y_random[c] >= Y[c] })
 *   StringSets::RunApproxPermutations :606-640 (P x { permute, approximate p-values, merge into the sorted
 *                                               observed ones }) for `-S n`, ratio, t, corr
 * Everything else of that tool (reading the table, p-value -> FDR -> adjusted p-value arithmetic,
 * printing) is host C++ above this boundary (ibm-cbc-genomic-tools_amd/csrc/permutation_test.cpp).
 *
 * Plain pointers and sizes; all pointers are HOST pointers; calls are synchronous.  Every function
 * returns GTX_OK (0) or a negative GTX_E* code from gtx.h; gtx_perm_last_error() has the message.
 *
 * The permutation source.  The reference draws from GSL's generator seeded with getpid()+time(NULL)
 * (:557), so its permutations are not reproducible and not specified.  Here permutation number p of
 * n rows under a 64-bit seed is DEFINED as follows (oracle/perm_oracle.c implements the same thing on
 * the CPU, which is what makes device results checkable bit for bit):
 *
 *   mix64(z):  z = (z ^ z>>30) * 0xBF58476D1CE4E5B9;  z = (z ^ z>>27) * 0x94D049BB133111EB;  return z ^ z>>31
 *   fmix32(h): h ^= h>>16; h *= 0x85ebca6b; h ^= h>>13; h *= 0xc2b2ae35; h ^= h>>16
 *   s = mix64(mix64(seed + G) ^ ((p+1) * 0xD6E8FEB86659FD93)),  G = 0x9E3779B97F4A7C15
 *   thirteen 64-bit words w_i = mix64(s + (i+1)*G): key[2i], key[2i+1] = low, high half of w_i (i < 5),
 *   fy[2i], fy[2i+1] = low, high half of w_(5+i) (i < 8)
 *   n <= 16:  Fisher-Yates on the identity, i = n-1 .. 1, j = (fy[i] * (i+1)) >> 32, swap(i, j)
 *   n  > 16:  a = ceil(sqrt n), b = ceil(n / a); (L, R) = (r / b, r % b); repeat { five times {
 *             L = (L + ((fmix32(R + key[2i]) * a) >> 32)) mod a;  R = (R + ((fmix32(L + key[2i+1]) * b) >> 32)) mod b };
 *             x = L * b + R } until x < n        (an alternating additive Feistel network on the a x b grid that
 *             holds the rows, cycle-walked over its < 2 sqrt(n) spare cells; products are 32 x 32 -> 64 bit)
 *   permuted row r takes the values of source row pi_p(r):  V_p[r] = V[pi_p(r)], Vtotal_p[r] = Vtotal[pi_p(r)]
 *
 * Permutations of different p are independent draws (the reference composes its shuffles; both give
 * uniformly distributed arrangements, which is all the test uses).
 */
#ifndef GTX_PERM_H
#define GTX_PERM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gtx_perm gtx_perm;

/* statistics, in the order of the reference's -S option (permutation_test.cpp:753-760) */
enum {
  GTX_STAT_SUM = 0,    /* CalcSumStatistic          :487-520 */
  GTX_STAT_N = 1,      /* CalcHyperGeomStatistic    :397-413 (approx = false: the count k) */
  GTX_STAT_SENS = 2,   /* CalcSensitivityStatistic  :370-387 */
  GTX_STAT_SPEC = 3,   /* CalcSpecificityStatistic  :341-360 */
  GTX_STAT_RATIO = 4,  /* CalcRatioStatistic        :423-476 */
  GTX_STAT_T = 5,      /* CalcTStatistic            :281-331 */
  GTX_STAT_CORR = 6    /* CalcCorrStatistic         :527-545 + VectorCorr core.cpp:1535-1558 */
};

/* flags of gtx_perm_set_table */
#define GTX_PERM_USE_TOTALS 1u     /* StringSets::use_totals (:207-211) */

int gtx_perm_create(int device, gtx_perm **out);
void gtx_perm_destroy(gtx_perm *p);
const char *gtx_perm_last_error(const gtx_perm *p);

/* The table (StringSets members after the constructor, :120-215):
 *   col_ptr[n_cols+1], rows[col_ptr[n_cols]]  membership lists B[c][1..B[c][0]] of all categories back to back,
 *                                             each in the order the reference fills it (ascending row)
 *   V[n_rows], Vtotal[n_rows]                 per-row values; Vtotal == NULL means all 1
 *   sums[4]                                   Vsum, VsumZ, Vsum2, Vtotal_sum as the constructor computes them
 * n_rows < 2^24 (a 64-permutation slab tile is addressed with 32-bit byte offsets); the table is copied to the device. */
int gtx_perm_set_table(gtx_perm *p, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows,
                       const float *V, const float *Vtotal, const double *sums, uint32_t flags);

/* Y[c] of the table as given (Calc*Statistic(approx = false) on the unpermuted values). `under` = -u. */
int gtx_perm_statistic(gtx_perm *p, int stat, int under, double *Y);

/* counts[c] = #{ q in [first_perm, first_perm + n_perm) : Y_q[c] >= Y[c] }   (RunPermutations :561-566;
 * the caller divides by the number of permutations).  Shards of the permutation range add up, which is
 * how several GPUs split the work. */
int gtx_perm_count_ge(gtx_perm *p, int stat, int under, const double *Y, uint64_t seed, int64_t first_perm, int64_t n_perm,
                      uint64_t *counts);

/* `-S n -a` (RunApproxPermutations :612-627).  tab[tab_ptr[c] + k] = approximate p-value of category c when k of
 * its rows are positive (negative with -u), k = 0 .. n_c; sortedY[n_cols] = the observed p-values, ascending.
 * counts[z] = #{ (q, c) : lower_bound(sortedY, tab[tab_ptr[c] + k_q(c)]) == z } -- the histogram that the
 * reference's sort + two-pointer merge produces (before its running sum, :631-634). */
int gtx_perm_count_rank(gtx_perm *p, int under, const int64_t *tab_ptr, const double *tab, const double *sortedY, uint64_t seed,
                        int64_t first_perm, int64_t n_perm, uint64_t *counts);

/* -a for ratio (without totals), t and corr (Calc*Statistic(approx = true) :305-308, :336-339, :447-451, :542 and the same
 * RunApproxPermutations loop).  The reference turns the statistic into a p-value with gsl_cdf_ugaussian_Q / gsl_cdf_tdist_Q; GSL is
 * not linked here, the two tails are DEFINED as
 *   gauss_Q(x) = erfc(x / sqrt 2) / 2
 *   tdist_Q(t, nu) = I_x(nu/2, 1/2) / 2 for t >= 0, 1 - that for t < 0, x = nu / (nu + t^2); NaN for nu <= 0 or t NaN; 0 / 1 at +-inf
 * with the regularised incomplete beta function I_x(a, b) = x^a (1-x)^b / (a B(a, b)) * cf(a, b, x) by its continued fraction
 * (modified Lentz), evaluated on the side of (a + 1) / (a + b + 2) where it converges fast (the other side through I_x(a, b) =
 * 1 - I_(1-x)(b, a)); ln B by lgamma.  Against the exact function: ~1e-9 relative for nu <= 1e6 (tolerance parity with GSL by
 * construction, like every tail probability of this build).  The degrees of freedom of t are `(long)floor(..)` there: a quotient
 * that is NaN or out of range takes the `df < 0 ? 1.0` branch, as the conversion does on x86-64.
 * gtx_perm_statistic_approx: P[c] = that p-value of the table as given -- computed on the device, so that an observed value and a
 * permutation that reproduces its statistic compare equal.  gtx_perm_count_rank_approx: the histogram of gtx_perm_count_rank over
 * the approximate p-values of the permutations (a NaN p-value has no smaller observed one: it falls at the front).
 * GTX_E_ARG for sum / sens / spec / ratio with totals (the reference: "not implemented yet") and for `n` (by table, above). */
int gtx_perm_statistic_approx(gtx_perm *p, int stat, int under, double *P);
int gtx_perm_count_rank_approx(gtx_perm *p, int stat, int under, const double *sortedY, uint64_t seed, int64_t first_perm, int64_t n_perm,
                               uint64_t *counts);

/* out[r] = pi_q(r), r < n_rows of the current table (tests; the definition above) */
int gtx_perm_permutation(gtx_perm *p, uint64_t seed, int64_t q, int32_t *out);

/* kernel time of the last gtx_perm_count_* call in milliseconds, summed over its batches: apply = writing the
 * permuted value slabs, stat = the gather-sum over the membership lists (HIP events on the launch stream) */
int gtx_perm_last_ms(gtx_perm *p, float *apply_ms, float *stat_ms);

#ifdef __cplusplus
}
#endif
#endif
