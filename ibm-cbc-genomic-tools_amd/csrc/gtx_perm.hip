// gtx_perm.hip -- category permutation test on the MI355X: kernels + C ABI (include/gtx_perm.h).
//
// What the reference does (gtools/permutation_test.cpp:555-572): P times { shuffle the per-row value
// vector, recompute every category's statistic over its membership list, compare with the observed
// one }.  One permutation at a time, one category at a time, one row at a time.
//
// Here the permutation number is the SIMD dimension:
//   perm_apply_kernel   writes a slab Vp[row][q] = V[pi_q(row)] for a batch of permutations q; pi_q is a
//                       keyed bijection evaluated independently per (row, q) -- no shuffle state, no
//                       sequential dependence (definition: include/gtx_perm.h);
//   perm_stat_kernel    one wave = one category x 64 permutations.  The membership list is wave-uniform
//                       (scalar loads); each gathered row is one coalesced 256-byte read of the slab;
//                       every lane accumulates ITS permutation's sums in double, sequentially in list
//                       order -- the reference's summation order, so the statistic is bit-identical to
//                       the CPU restatement -- then finishes the statistic, compares and the wave adds
//                       popcount(ballot) to the category's counter.
// The slab is stored in tiles of 64 permutations (n_rows x 256 B each); every XCD walks its own sequence
// of tiles, all categories of a tile before the next, so a tile is pulled into ONE L2 and re-read there.
// Bound: slab bytes moved = 4 B x (membership entries) x (permutations) -- an HBM/L2 gather stream; the
// double adds are ~1/10 of the FP64 rate at that bandwidth.  No MFMA: the 0/1 membership matrix is ~1%
// dense, a dense contraction would do 100x the work.
//
// Compiled with -ffp-contract=off: the reference's host compiler does not fuse a*b+c on x86-64, and
// results are compared with >=.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>
#include "gtx.h"
#include "gtx_perm.h"
#include "gtx_kernels.h"   // gtx::PerDevice

namespace {

typedef unsigned long long u64;
typedef long long i64;

constexpr int kRounds = 10;

struct PermKeys { uint32_t key[kRounds]; u64 fyw; };   // fyw: the whole n <= 16 permutation, one nibble per row

__host__ __device__ inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__host__ __device__ inline u64 mix64(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__host__ __device__ inline void perm_keys_init(PermKeys &k, u64 seed, u64 q, uint32_t n)
{
  const u64 G = 0x9E3779B97F4A7C15ull;
  u64 s = mix64(mix64(seed + G) ^ ((q + 1) * 0xD6E8FEB86659FD93ull));
  for (int i = 0; i < kRounds; i += 2) { s += G; const u64 z = mix64(s); k.key[i] = (uint32_t)z; k.key[i + 1] = (uint32_t)(z >> 32); }
  k.fyw = 0xFEDCBA9876543210ull;
  if (n <= 16) {
    uint32_t fy[16];
    for (int i = 0; i < 16; i += 2) { s += G; const u64 z = mix64(s); fy[i] = (uint32_t)z; fy[i + 1] = (uint32_t)(z >> 32); }
    u64 w = k.fyw;
    for (uint32_t i = n - 1; i > 0; i--) {
      const uint32_t j = (uint32_t)(((u64)fy[i] * (i + 1)) >> 32);
      const u64 a = (w >> (4 * i)) & 15, b = (w >> (4 * j)) & 15;
      w ^= ((a ^ b) << (4 * i)) | ((a ^ b) << (4 * j));
    }
    k.fyw = w;
  }
}

struct PermGeom { uint32_t n, a, b; };      // n > 16: rows live on an a x b grid, a = ceil(sqrt n), b = ceil(n / a)

inline PermGeom perm_geom(uint32_t n)
{
  PermGeom g; g.n = n;
  uint32_t r = (uint32_t)sqrt((double)n);
  while ((u64)r * r < n) r++;
  while (r > 1 && (u64)(r - 1) * (r - 1) >= n) r--;
  g.a = r; g.b = (n + r - 1) / r;
  return g;
}

// image of grid cell (L, R) = row L * b + R under permutation k
__host__ __device__ inline uint32_t perm_at(const PermKeys &k, const PermGeom &g, uint32_t L, uint32_t R)
{
  if (g.n <= 16) return (uint32_t)((k.fyw >> (4 * (L * g.b + R))) & 15);
  uint32_t x;
  do {
#pragma unroll
    for (int i = 0; i < kRounds; i += 2) {
      L += (uint32_t)(((u64)fmix32(R + k.key[i]) * g.a) >> 32); if (L >= g.a) L -= g.a;
      R += (uint32_t)(((u64)fmix32(L + k.key[i + 1]) * g.b) >> 32); if (R >= g.b) R -= g.b;
    }
    x = L * g.b + R;
  } while (x >= g.n);                       // cycle walking: spare cells (< 2 sqrt n of them) are stepped over
  return x;
}

// ---------------------------------------------------------------------------------------------
// slab writer.  Slab layout: tiles of 64 permutations, tile t = [n_rows][64] floats (one 256-byte line per row),
// element (row r, permutation j) at ((j / 64) * n_rows + r) * 64 + j % 64.  Vp(r, j) = V[pi_(first+j)(r)];
// blockIdx.x = 256 permutations, blockIdx.y = row chunk
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void perm_apply_kernel(const float *__restrict__ V, const float *__restrict__ Vt, float *__restrict__ Vp,
                                                         float *__restrict__ Vtp, PermGeom g, uint32_t rowsPerBlock, u64 seed,
                                                         i64 firstPerm, i64 nPerm)
{
  const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= nPerm) return;
  PermKeys k; perm_keys_init(k, seed, (u64)(firstPerm + j), g.n);
  const uint32_t r0 = blockIdx.y * rowsPerBlock, r1 = (uint32_t)min((u64)g.n, (u64)r0 + rowsPerBlock);
  uint32_t L = r0 / g.b, R = r0 % g.b;            // grid cell of the row, stepped along with it
  for (uint32_t r = r0; r < r1; r++) {
    const uint32_t src = perm_at(k, g, L, R);
    const size_t at = ((size_t)(j >> 6) * g.n + r) * 64 + (j & 63);
    Vp[at] = V[src];
    if (Vtp) Vtp[at] = Vt[src];
    if (++R == g.b) { R = 0; L++; }
  }
}

// The same slab for tables whose grid sides are small (a, b <= 192: up to ~36 k rows).  The round function of perm_at depends on
// the permutation, the round and ONE grid coordinate only -- at most max(a, b) values -- so a block that writes one tile of 64
// permutations first tabulates it: tab[j][i][x] = (fmix32(x + key_j[i]) * side) >> 32, one byte each (10 rounds x <= 192 x 64
// permutations <= 120 KB of LDS), and a row's image is then ten dependent byte reads with an add and a conditional subtract each
// -- against two 32-bit multiplies, a multiply-high and the shifts of fmix32 per round, which is what bounds perm_apply_kernel
// (VALU).  Same arithmetic, same cycle walking: bit-identical images (include/gtx_perm.h).  lane = permutation (a row is one
// coalesced 256-byte store of the tile), a wave takes kTabRows rows at a time so that their chains of LDS reads overlap.
// 20 k rows x 10 k permutations: 0.59 ms against 0.75 for perm_apply_kernel; 2, 4 or 8 rows in flight per wave make no difference.
// Layout: a 32-bit word holds, for ONE coordinate x and a PAIR of rounds (2p, 2p + 1), the entries of lanes j % 32 and j % 32 + 32:
// entry (round i, coordinate x, lane j) at byte ((i / 2 * T + x) * 32 + j % 32) * 4 + 2 (i & 1) + j / 32.  A lane always reads bank
// j % 32 -- the 32 lanes of a half-wave never meet in a bank whatever their coordinates are -- and the address is linear in x: one
// shift-add per lookup.  (Round 3 paired the coordinates 2x, 2x + 1 in a word instead: the same banks, but two more vector
// instructions per lookup for the parity of x -- in a loop that is bound by their number: 0.65 ms.  One byte per entry in plain
// [x][lane] order has the one-instruction address too and meets in banks: 0.59 ms.)
constexpr int kTabMaxSide = 192, kTabRows = 4;

__device__ __forceinline__ uint32_t tab_step(const unsigned char *__restrict__ tab, uint32_t base, uint32_t x)
{
  return tab[(x << 7) + base];                                       // one shift-add (the layout below)
}

__global__ __launch_bounds__(1024) void perm_apply_tab_kernel(const float *__restrict__ V, const float *__restrict__ Vt, float *__restrict__ Vp,
                                                              float *__restrict__ Vtp, PermGeom g, uint32_t rowsPerBlock, u64 seed,
                                                              i64 firstPerm, i64 nPerm, uint32_t T)
{
  extern __shared__ unsigned char tab[];                            // 10 * T * 64 bytes
  __shared__ uint32_t keys[64][kRounds];
  const i64 j0 = (i64)blockIdx.x * 64;
  if (threadIdx.x < 64) {
    PermKeys k; perm_keys_init(k, seed, (u64)(firstPerm + j0 + threadIdx.x), g.n);    // (lanes beyond nPerm: some permutation nobody stores)
#pragma unroll
    for (int i = 0; i < kRounds; i++) keys[threadIdx.x][i] = k.key[i];
  }
  __syncthreads();
  for (uint32_t idx = threadIdx.x; idx < 64u * kRounds * T; idx += blockDim.x) {
    const uint32_t j = idx & 63u, y = idx >> 6, i = y / T, x = y - i * T;
    const uint32_t side = (i & 1) ? g.b : g.a;                        // even rounds move L (argument R < b), odd rounds move R (argument L < a)
    tab[(((i >> 1) * T + x) << 7) + ((j & 31u) << 2) + ((i & 1u) << 1) + (j >> 5)] = (unsigned char)(((u64)fmix32(x + keys[j][i]) * side) >> 32);
  }
  __syncthreads();
  // (the wave's number as a scalar: the rows it takes, their grid cells and the addresses of their lines are then scalar work)
  const uint32_t lane = threadIdx.x & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  uint32_t base[kRounds];
#pragma unroll
  for (int i = 0; i < kRounds; i++) base[i] = (((uint32_t)i >> 1) * T << 7) + ((lane & 31u) << 2) + (((uint32_t)i & 1u) << 1) + (lane >> 5);
  const uint32_t r0 = blockIdx.y * rowsPerBlock, r1 = (uint32_t)min((u64)g.n, (u64)r0 + rowsPerBlock);
  const bool live = j0 + lane < nPerm;
  float *__restrict__ tile = Vp + (size_t)blockIdx.x * g.n * 64, *__restrict__ tileT = Vtp ? Vtp + (size_t)blockIdx.x * g.n * 64 : nullptr;
  // the grid cell of the wave's first row, stepped along with it (scalar work: no division per row)
  const uint32_t stepRows = nw * kTabRows, stepL = stepRows / g.b, stepR = stepRows - stepL * g.b;
  uint32_t cellL = (r0 + wv * kTabRows) / g.b, cellR = (r0 + wv * kTabRows) - cellL * g.b;
  for (uint32_t rb = r0 + wv * kTabRows; rb < r1; rb += stepRows) {
    uint32_t L[kTabRows], R[kTabRows], x[kTabRows];
#pragma unroll
    for (int u = 0; u < kTabRows; u++) {
      // row rb + u = cell (cellL, cellR + u), one wrap at most (b >= 4 > u for n > 16); rows beyond the block's last stand in for row rb
      const bool in = rb + u < r1;
      uint32_t Ru = cellR + (in ? (uint32_t)u : 0u), Lu = cellL;
      if (Ru >= g.b) { Ru -= g.b; Lu++; }
      L[u] = Lu; R[u] = Ru;
    }
    cellR += stepR; cellL += stepL;
    if (cellR >= g.b) { cellR -= g.b; cellL++; }
#pragma unroll
    for (int i = 0; i < kRounds; i += 2) {
#pragma unroll
      for (int u = 0; u < kTabRows; u++) { L[u] += tab_step(tab, base[i], R[u]); L[u] = min(L[u], L[u] - g.a); }
#pragma unroll
      for (int u = 0; u < kTabRows; u++) { R[u] += tab_step(tab, base[i + 1], L[u]); R[u] = min(R[u], R[u] - g.b); }
    }
    bool spare = false;
#pragma unroll
    for (int u = 0; u < kTabRows; u++) { x[u] = L[u] * g.b + R[u]; spare |= x[u] >= g.n; }
    if (__builtin_amdgcn_ballot_w64(spare)) {                        // (one test for the rows of the step: spare cells are < 2 sqrt n of n)
      for (int u = 0; u < kTabRows; u++) {
        while (x[u] >= g.n) {                                         // cycle walking
          for (int i = 0; i < kRounds; i += 2) {
            L[u] += tab_step(tab, base[i], R[u]); L[u] = min(L[u], L[u] - g.a);
            R[u] += tab_step(tab, base[i + 1], L[u]); R[u] = min(R[u], R[u] - g.b);
          }
          x[u] = L[u] * g.b + R[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kTabRows; u++) {
      const uint32_t r = rb + u;
      if (r < r1 && live) {
        tile[(size_t)r * 64 + lane] = V[x[u]];
        if (tileT) tileT[(size_t)r * 64 + lane] = Vt[x[u]];
      }
    }
  }
}

// The slab writer once more, for value vectors that fit the LDS NEXT to the table: with the rounds tabulated, what is left of a row's
// cost is the gather V[image] -- 64 lanes, 64 different cache lines of an 80 KB vector, one line per cycle through the CU's L1:
// ~62 cycles per row and wave, more than the ten table reads and their arithmetic together.  So a block takes HALF a tile -- 32
// permutations: the table is then 12 bytes per (coordinate, permutation) = 384 T bytes (54 KB at 20 k rows) -- and keeps the
// whole value vector in LDS beside it (80 KB): the gather becomes one LDS read per row and lane.  A wave's lanes 0-31 hold one row
// and lanes 32-63 the next (two 128-byte halves of two of the tile's lines per store).  Table layout: word ((i / 4) * T + x) * 32 + j
// holds the entries of rounds 4 (i / 4) .. + 3 for coordinate x and permutation j: a lane reads bank j whatever x is, and the
// address is linear in x.  Same arithmetic, same cycle walking: bit-identical slabs.  20 k rows x 10 k permutations, same box:
// 0.48 ms against 0.60 for the 64-permutation table with the values gathered from memory (eight rows in flight per lane instead
// of four: 0.52).
__global__ __launch_bounds__(1024) void perm_apply_tabv_kernel(const float *__restrict__ V, const float *__restrict__ Vt, float *__restrict__ Vp,
                                                               float *__restrict__ Vtp, PermGeom g, uint32_t rowsPerBlock, u64 seed,
                                                               i64 firstPerm, i64 nPerm, uint32_t T)
{
  extern __shared__ unsigned char tabv[];                           // 3 * T * 128 bytes of table, then n floats
  __shared__ uint32_t keys[32][kRounds];
  constexpr uint32_t kGroups = (kRounds + 3) / 4;
  float *__restrict__ Vs = (float *)(tabv + (size_t)kGroups * T * 128);
  const i64 j0 = (i64)blockIdx.x * 32;
  if (threadIdx.x < 32) {
    PermKeys k; perm_keys_init(k, seed, (u64)(firstPerm + j0 + threadIdx.x), g.n);
#pragma unroll
    for (int i = 0; i < kRounds; i++) keys[threadIdx.x][i] = k.key[i];
  }
  for (uint32_t r = threadIdx.x; r < g.n; r += blockDim.x) Vs[r] = V[r];
  __syncthreads();
  for (uint32_t idx = threadIdx.x; idx < 32u * kRounds * T; idx += blockDim.x) {
    const uint32_t j = idx & 31u, y = idx >> 5, i = y / T, x = y - i * T;
    const uint32_t side = (i & 1) ? g.b : g.a;
    tabv[(((i >> 2) * T + x) << 7) + (j << 2) + (i & 3u)] = (unsigned char)(((u64)fmix32(x + keys[j][i]) * side) >> 32);
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, half = lane >> 5, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  uint32_t base[kRounds];
#pragma unroll
  for (int i = 0; i < kRounds; i++) base[i] = (((uint32_t)i >> 2) * T << 7) + ((lane & 31u) << 2) + ((uint32_t)i & 3u);
  const uint32_t r0 = blockIdx.y * rowsPerBlock, r1 = (uint32_t)min((u64)g.n, (u64)r0 + rowsPerBlock);
  const bool live = j0 + (lane & 31u) < nPerm;
  const size_t tileOff = (size_t)(j0 >> 6) * g.n * 64 + (size_t)(j0 & 32) + (lane & 31u);     // (this lane's column of the 64-permutation tile)
  float *__restrict__ tile = Vp + tileOff, *__restrict__ tileT = Vtp ? Vtp + tileOff : nullptr;
  constexpr uint32_t kRowsPerStep = 2 * kTabRows;                   // rows a wave takes per step: slot u of a lane is row rb + 2 u + half
  const uint32_t stepRows = nw * kRowsPerStep, stepL = stepRows / g.b, stepR = stepRows - stepL * g.b;
  uint32_t cellL = (r0 + wv * kRowsPerStep) / g.b, cellR = (r0 + wv * kRowsPerStep) - cellL * g.b;
  for (uint32_t rb = r0 + wv * kRowsPerStep; rb < r1; rb += stepRows) {
    uint32_t L[kTabRows], R[kTabRows], x[kTabRows];
#pragma unroll
    for (int u = 0; u < kTabRows; u++) {
      const uint32_t off = 2u * (uint32_t)u + half;                  // < 8 <= 2 b: two wraps at most (b >= 4 for n > 16)
      uint32_t Ru = cellR + (rb + off < r1 ? off : 0u), Lu = cellL; // rows beyond the block's last stand in for row rb
      if (Ru >= g.b) { Ru -= g.b; Lu++; }
      if (Ru >= g.b) { Ru -= g.b; Lu++; }
      L[u] = Lu; R[u] = Ru;
    }
    cellR += stepR; cellL += stepL;
    if (cellR >= g.b) { cellR -= g.b; cellL++; }
#pragma unroll
    for (int i = 0; i < kRounds; i += 2) {
#pragma unroll
      for (int u = 0; u < kTabRows; u++) { L[u] += tab_step(tabv, base[i], R[u]); L[u] = min(L[u], L[u] - g.a); }
#pragma unroll
      for (int u = 0; u < kTabRows; u++) { R[u] += tab_step(tabv, base[i + 1], L[u]); R[u] = min(R[u], R[u] - g.b); }
    }
    bool spare = false;
#pragma unroll
    for (int u = 0; u < kTabRows; u++) { x[u] = L[u] * g.b + R[u]; spare |= x[u] >= g.n; }
    if (__builtin_amdgcn_ballot_w64(spare)) {
      for (int u = 0; u < kTabRows; u++) {
        while (x[u] >= g.n) {                                         // cycle walking
          for (int i = 0; i < kRounds; i += 2) {
            L[u] += tab_step(tabv, base[i], R[u]); L[u] = min(L[u], L[u] - g.a);
            R[u] += tab_step(tabv, base[i + 1], L[u]); R[u] = min(R[u], R[u] - g.b);
          }
          x[u] = L[u] * g.b + R[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kTabRows; u++) {
      const uint32_t r = rb + 2u * (uint32_t)u + half;
      if (r < r1 && live) {
        tile[(size_t)r * 64] = Vs[x[u]];
        if (tileT) tileT[(size_t)r * 64] = Vt[x[u]];
      }
    }
  }
}

__global__ void perm_list_kernel(int32_t *__restrict__ out, PermGeom g, u64 seed, i64 q)
{
  PermKeys k; perm_keys_init(k, seed, (u64)q, g.n);
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < g.n; r += gridDim.x * blockDim.x) out[r] = (int32_t)perm_at(k, g, r / g.b, r % g.b);
}

// ---------------------------------------------------------------------------------------------
// per-category accumulators: add() in membership order, finish() = the reference's closing arithmetic
// ---------------------------------------------------------------------------------------------
struct StatConsts { double Vsum, VsumZ, Vsum2, VtotalSum; i64 nRows; i64 tAll; int under; };

// ---- the distribution functions of -a for ratio / t / corr (gsl_cdf_ugaussian_Q, gsl_cdf_tdist_Q at permutation_test.cpp:307, :338, :450,
// :542; GSL is not linked).  Defined in include/gtx_perm.h: Q(x) = erfc(x / sqrt 2) / 2; Student's t with nu degrees of freedom:
// Q(t) = I_x(nu/2, 1/2) / 2 for t >= 0 with x = nu / (nu + t^2), the regularised incomplete beta function by its continued fraction
// (modified Lentz) on the side where it converges fast.  The tests hold a CPU evaluation of the same statements beside it (device
// and host differ in the last bits of log / exp / lgamma / erfc only).
__device__ inline double beta_cf(double a, double b, double x)
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
__device__ inline double ibeta_reg(double a, double b, double x, double y)      // I_x(a, b), y = 1 - x from the caller
{
  if (!(x > 0.0)) return 0.0;
  if (!(y > 0.0)) return 1.0;
  const double lx = x < 0.5 ? log(x) : log1p(-y), ly = y < 0.5 ? log(y) : log1p(-x);
  const double pre = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * lx + b * ly);
  if (x < (a + 1.0) / (a + b + 2.0)) return pre * beta_cf(a, b, x) / a;
  return 1.0 - pre * beta_cf(b, a, y) / b;
}
__device__ inline double tdist_Q(double t, double nu)
{
  if (t != t || !(nu > 0.0)) return __builtin_nan("");
  if (isinf(t)) return t > 0 ? 0.0 : 1.0;
  const double t2 = t * t, den = nu + t2;
  const double tail = 0.5 * ibeta_reg(0.5 * nu, 0.5, nu / den, t2 / den);
  return t >= 0 ? tail : 1.0 - tail;
}
__device__ inline double gauss_Q(double x) { return 0.5 * erfc(x / 1.41421356237309504880); }
// `(long int)floor(..)` of :306 / :337 and `df < 0 ? 1.0 : tdist_Q(Y, df)`: a NaN or an out-of-range quotient converts to LONG_MIN on
// x86-64, i.e. takes the 1.0 branch
__device__ inline double welch_tail(double y, double s0, i64 n0, double s1, i64 n1)
{
  const double a = s0 + s1;
  const double dfd = floor(a * a / (s0 * s0 / (double)(n0 - 1) + s1 * s1 / (double)(n1 - 1)));
  if (!(dfd >= 0.0) || dfd >= 9223372036854775808.0) return 1.0;
  return tdist_Q(y, (double)(i64)dfd);
}

template <int STAT, bool TOTALS> struct Acc;

// (kWords: the leading 8-byte words that make up the running state -- what travels between the row-range launches)
template <bool TOTALS> struct Acc<GTX_STAT_SUM, TOTALS> {                      // :487-520
  static constexpr int kWords = TOTALS ? 2 : 1;
  double y = 0, yt = 0;
  __device__ void add(float v, float vt) { y += v; if (TOTALS) yt += vt; }
  __device__ double finish(i64 nc, const StatConsts &k) const { double r = TOTALS ? y / yt : y / nc; return k.under ? -r : r; }
  __device__ double approx(i64, const StatConsts &) const { return __builtin_nan(""); }      // ("not implemented yet" there: refused by the host)
};

template <int STAT, bool TOTALS> struct AccCount {                              // :341-413
  static constexpr int kWords = 1;
  i64 kk = 0; int under;
  __device__ void add(float v, float) { kk += under ? v < 0 : v > 0; }
  __device__ double finish(i64 nc, const StatConsts &k) const
  {
    return STAT == GTX_STAT_N ? (double)kk : STAT == GTX_STAT_SENS ? (double)kk / nc : (double)kk / k.tAll;
  }
  __device__ double approx(i64, const StatConsts &) const { return __builtin_nan(""); }      // (`n`: by table, MODE_RANK)
};
template <bool TOTALS> struct Acc<GTX_STAT_N, TOTALS> : AccCount<GTX_STAT_N, TOTALS> {};
template <bool TOTALS> struct Acc<GTX_STAT_SENS, TOTALS> : AccCount<GTX_STAT_SENS, TOTALS> {};
template <bool TOTALS> struct Acc<GTX_STAT_SPEC, TOTALS> : AccCount<GTX_STAT_SPEC, TOTALS> {};

template <int STAT> struct AccMoments {                                          // no totals: :288-307, :430-447
  static constexpr int kWords = 2;
  double m1 = 0, v1 = 0;
  __device__ void add(float v, float) { m1 += v; v1 += v * v; }                  // float product, as in the reference
  __device__ double finish(i64 nc, const StatConsts &k) const
  {
    i64 n[2]; double mean[2], var[2];
    n[1] = nc; mean[1] = m1; var[1] = v1;
    n[0] = k.nRows - n[1]; mean[0] = k.Vsum - mean[1]; var[0] = k.Vsum2 - var[1];
    for (int i = 0; i <= 1; i++) { mean[i] /= n[i]; var[i] = var[i] / n[i] - mean[i] * mean[i]; }
    if (STAT == GTX_STAT_RATIO) return k.under ? mean[0] / mean[1] : mean[1] / mean[0];
    const double y = (mean[1] - mean[0]) / sqrt(var[1] / n[1] + var[0] / n[0]);
    return k.under ? -y : y;
  }
  __device__ double approx(i64 nc, const StatConsts &k) const                      // Calc*Statistic(approx = true): :305-308, :447-451
  {
    const double y = finish(nc, k);
    if (STAT == GTX_STAT_RATIO) {
      const double m = k.Vsum / k.nRows, v = k.Vsum2 / k.nRows;
      return gauss_Q((m * y - m) / sqrt(v * (y * y) + v));
    }
    const i64 n1 = nc, n0 = k.nRows - nc;
    double mean0 = (k.Vsum - m1) / n0, mean1 = m1 / n1;
    const double var0 = (k.Vsum2 - v1) / n0 - mean0 * mean0, var1 = v1 / n1 - mean1 * mean1;
    return welch_tail(y, var0 / n0, n0, var1 / n1, n1);
  }
};
template <> struct Acc<GTX_STAT_RATIO, false> : AccMoments<GTX_STAT_RATIO> {};
template <> struct Acc<GTX_STAT_T, false> : AccMoments<GTX_STAT_T> {};

template <> struct Acc<GTX_STAT_RATIO, true> {                                   // :449-470
  static constexpr int kWords = 2;
  double s1 = 0, t1 = 0;
  __device__ void add(float v, float vt) { s1 += v; t1 += vt; }
  __device__ double finish(i64, const StatConsts &k) const
  {
    const double t0 = k.VtotalSum - t1, s0 = k.Vsum - s1;
    const double mean0 = s0 / t0, mean1 = s1 / t1;
    return k.under ? mean0 / mean1 : mean1 / mean0;
  }
  __device__ double approx(i64, const StatConsts &) const { return __builtin_nan(""); }      // ("not implemented yet" there, :475)
};

template <> struct Acc<GTX_STAT_T, true> {                                       // :309-329
  static constexpr int kWords = 4;
  double s1 = 0, t1 = 0, z1 = 0, q1 = 0;
  __device__ void add(float v, float vt)
  {
    s1 += v; t1 += vt; z1 += v / vt;
    const double x = (double)v / vt;               // pow(x, 2.0) there; x*x is its correctly rounded value
    q1 += x * x;
  }
  __device__ double finish(i64 nc, const StatConsts &k) const
  {
    i64 n[2]; double sum[2], total[2], sumZ[2], sumqZ[2], mean[2], varZ[2];
    n[1] = nc; sum[1] = s1; total[1] = t1; sumZ[1] = z1; sumqZ[1] = q1;
    n[0] = k.nRows - n[1]; total[0] = k.VtotalSum - total[1]; sum[0] = k.Vsum - sum[1]; sumZ[0] = k.VsumZ - sumZ[1]; sumqZ[0] = k.Vsum2 - sumqZ[1];
    for (int i = 0; i <= 1; i++) { mean[i] = sum[i] / total[i]; const double a = sumZ[i] / n[i]; varZ[i] = sumqZ[i] / n[i] - a * a; }
    const double y = (mean[1] - mean[0]) / sqrt(varZ[1] / n[1] + varZ[0] / n[0]);
    return k.under ? -y : y;
  }
  __device__ double approx(i64 nc, const StatConsts &k) const                      // :336-339
  {
    const double y = finish(nc, k);
    const i64 n1 = nc, n0 = k.nRows - nc;
    const double a0 = (k.VsumZ - z1) / n0, a1 = z1 / n1;
    const double varZ0 = (k.Vsum2 - q1) / n0 - a0 * a0, varZ1 = q1 / n1 - a1 * a1;
    return welch_tail(y, varZ0 / n0, n0, varZ1 / n1, n1);
  }
};

template <bool TOTALS> struct Acc<GTX_STAT_CORR, TOTALS> {                       // :527-545, core.cpp:1535-1558
  static constexpr int kWords = 6;
  double Ex = 0, Ey = 0, Ex2 = 0, Ey2 = 0, Exy = 0; u64 C = 0;
  __device__ void add(float v, float vt)
  {
    const double a = v, b = vt;
    if (a == a && b == b) { C++; Ex += a; Ex2 += a * a; Ey += b; Ey2 += b * b; Exy += a * b; }
  }
  __device__ double finish(i64, const StatConsts &k) const
  {
    const double ex = Ex / C, ey = Ey / C, ex2 = Ex2 / C, ey2 = Ey2 / C, exy = Exy / C;
    double y = (exy - ex * ey) / sqrt((ex2 - ex * ex) * (ey2 - ey * ey));
    y = fabs(y);
    return k.under ? 1.0 - y : y;
  }
  __device__ double approx(i64 nc, const StatConsts &k) const                      // :542 (n is the list's length there, not VectorCorr's pair count)
  {
    const double y = finish(nc, k);
    return tdist_Q(y * sqrt((double)(nc - 2) / (1 - y * y)), (double)(nc - 2));
  }
};

enum { MODE_STAT = 0, MODE_GE = 1, MODE_RANK = 2 };

struct StatArgs {
  const i64 *colPtr; const int32_t *rows;
  const uint32_t *rows16;         // the same lists as 16-bit row ids, two per word (tables of <= 65536 rows): half the bytes that compete with the slab rows for the L2
  const float *Vp, *Vtp;          // slab(s) in the tile layout above, or (MODE_STAT) the plain value vectors
  i64 nCols, nPerm;               // permutations in this batch
  StatConsts k;
  int approx;                     // MODE_STAT / MODE_RANK of ratio, t, corr: the statistic's approximate p-value (Acc::approx) instead of the statistic
  const double *Yobs;             // MODE_GE
  double *Yout;                   // MODE_STAT
  u64 *counts;
  const i64 *tabPtr; const double *tab; const double *sortedY;   // MODE_RANK
  i64 colBlocks;                  // blocks per slab tile
  // row-range parts (see perm_stat_kernel): this launch adds the members with rows in part `part` of `nParts`
  int part, nParts;
  const i64 *colSplit;            // [(nParts + 1) * nCols]: colSplit[q * nCols + c] = first member of c with row >= part q's first row
  u64 *accBuf;                    // accumulators between the parts: [tile][c][word][lane]
};

constexpr int kWavesPerBlock = 4;

template <int STAT, bool TOTALS, bool HASVT, int MODE, bool ROWS16>
__global__ __launch_bounds__(64 * kWavesPerBlock) void perm_stat_kernel(StatArgs a)
{
  // Workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8), and every XCD has its own
  // 4 MiB L2.  Each XCD therefore walks its own sequence of slab tiles (64 permutations = n_rows x 256 B),
  // all categories of one tile before the next, so that a tile is pulled into ONE L2 and re-read there.
  const i64 xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const i64 tile = xcd + 8 * (idx / a.colBlocks), cb = idx % a.colBlocks;
  if (tile * 64 >= a.nPerm) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const i64 c = cb * kWavesPerBlock + wave;
  if (c >= a.nCols) return;
  const i64 j = tile * 64 + lane;
  const bool valid = j < a.nPerm;
  // Row-range parts.  A 64-permutation tile is n_rows x 256 B; when that exceeds what an XCD's L2 keeps (4 MiB; 20 k rows are
  // 5.1 MB: hit rate 0.69, every tile fetched from the fabric many times over), the rows are cut into nParts ranges of <= ~2.7 MB
  // and each range is ONE launch over all tiles and categories -- the working set of an XCD is then one range of one tile.
  // Membership lists are in ascending row order (permutation_test.cpp:120-186 appends rows as it reads them), so a range is a
  // contiguous piece [zs, ze) of every list; the accumulators travel from launch to launch through accBuf, and a category's
  // members are still added in list order: the sums, hence every statistic and every >= decision, are bit for bit the same.
  const i64 z0 = a.nParts > 1 ? a.colSplit[(i64)a.part * a.nCols + c] : a.colPtr[c];
  const i64 z1 = a.nParts > 1 ? a.colSplit[(i64)(a.part + 1) * a.nCols + c] : a.colPtr[c + 1];
  // byte offset of (row r, this lane) inside the tile: 32 bits (n_rows < 2^24), so that a gather is one
  // v_lshl_add_u32 + one global_load with a scalar base.  MODE_STAT reads the unpermuted vectors: row r at r * 4.
  constexpr int kRowShift = MODE == MODE_STAT ? 2 : 8;
  const uint32_t laneOff = MODE == MODE_STAT ? 0u : (uint32_t)lane * 4u;
  const char *__restrict__ vp = (const char *)(a.Vp + (MODE == MODE_STAT ? 0 : (size_t)tile * a.k.nRows * 64));
  const char *__restrict__ vtp = HASVT ? (const char *)(a.Vtp + (MODE == MODE_STAT ? 0 : (size_t)tile * a.k.nRows * 64)) : nullptr;
  Acc<STAT, TOTALS> acc;
  if constexpr (STAT == GTX_STAT_N || STAT == GTX_STAT_SENS || STAT == GTX_STAT_SPEC) acc.under = a.k.under;
  constexpr int kAccWords = Acc<STAT, TOTALS>::kWords;
  static_assert(sizeof(acc) >= 8 * kAccWords, "the running state is the leading words of the accumulator");
  u64 *accAt = a.nParts > 1 ? a.accBuf + ((size_t)(tile * a.nCols + c) * kAccWords) * 64 + lane : nullptr;
  if (a.nParts > 1 && a.part > 0) {
    u64 w[kAccWords];
#pragma unroll
    for (int f = 0; f < kAccWords; f++) w[f] = __builtin_nontemporal_load(accAt + (size_t)f * 64);   // (streamed once: not to displace the slab rows in the L2)
    __builtin_memcpy((void *)&acc, w, 8 * kAccWords);
  }
  i64 z = z0;
  // one member: its row id from the 32-bit list or from the packed 16-bit one (z is wave-uniform: scalar loads and shifts)
  auto row_at = [&](i64 q) -> uint32_t {
    if constexpr (ROWS16) { const uint32_t w = a.rows16[q >> 1]; return (q & 1) ? w >> 16 : w & 0xffffu; }
    else return (uint32_t)a.rows[q];
  };
  if constexpr (ROWS16) {
    if ((z & 1) && z < z1) {                                        // up to an even member: the packed words are read whole from here on
      const uint32_t off = (row_at(z) << kRowShift) + laneOff;
      acc.add(*(const float *)(vp + off), HASVT ? *(const float *)(vtp + off) : 1.0f);
      z++;
    }
  }
  // 8 gathers in flight, accumulated in list order
  for (; z + 8 <= z1; z += 8) {
    float v[8], vt[8];
    uint32_t r[8];
    if constexpr (ROWS16) {
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t w = a.rows16[(z >> 1) + u]; r[2 * u] = w & 0xffffu; r[2 * u + 1] = w >> 16; }
    } else {
#pragma unroll
      for (int u = 0; u < 8; u++) r[u] = (uint32_t)a.rows[z + u];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t off = (r[u] << kRowShift) + laneOff;
      v[u] = *(const float *)(vp + off); vt[u] = HASVT ? *(const float *)(vtp + off) : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc.add(v[u], vt[u]);
  }
  for (; z < z1; z++) {
    const uint32_t off = (row_at(z) << kRowShift) + laneOff;
    acc.add(*(const float *)(vp + off), HASVT ? *(const float *)(vtp + off) : 1.0f);
  }
  if (a.nParts > 1 && a.part + 1 < a.nParts) {                     // not the last range: hand the accumulators on
    u64 w[kAccWords];
    __builtin_memcpy(w, (const void *)&acc, 8 * kAccWords);
#pragma unroll
    for (int f = 0; f < kAccWords; f++) __builtin_nontemporal_store(w[f], accAt + (size_t)f * 64);
    return;
  }
  const i64 nc = a.colPtr[c + 1] - a.colPtr[c];
  if constexpr (MODE == MODE_RANK) {
    // approximate p-value by table, then its lower bound among the sorted observed ones
    double val;
    if constexpr (STAT == GTX_STAT_N) val = a.tab[a.tabPtr[c] + acc.kk];
    else val = acc.approx(nc, a.k);
    i64 lo = 0, hi = a.nCols;
    while (lo < hi) { const i64 mid = (lo + hi) >> 1; if (a.sortedY[mid] < val) lo = mid + 1; else hi = mid; }
    if (valid && lo < a.nCols) atomicAdd(&a.counts[lo], 1ull);
  } else {
    if constexpr (MODE == MODE_STAT) { if (j == 0) a.Yout[c] = a.approx ? acc.approx(nc, a.k) : acc.finish(nc, a.k); }
    else {
      const double y = acc.finish(nc, a.k);
      const u64 m = __ballot(valid && y >= a.Yobs[c]);
      if (lane == 0 && m) atomicAdd(&a.counts[c], (u64)__popcll(m));
    }
  }
}

template <int STAT, bool TOTALS, bool HASVT>
hipError_t launch_stat_mode(int mode, const StatArgs &a, unsigned grid, hipStream_t st)
{
#define GTX_STAT(M) do { if (a.rows16) perm_stat_kernel<STAT, TOTALS, HASVT, M, true><<<grid, 64 * kWavesPerBlock, 0, st>>>(a); \
                         else perm_stat_kernel<STAT, TOTALS, HASVT, M, false><<<grid, 64 * kWavesPerBlock, 0, st>>>(a); } while (0)
  if (mode == MODE_STAT) GTX_STAT(MODE_STAT);
  else if (mode == MODE_GE) GTX_STAT(MODE_GE);
  else if constexpr (STAT == GTX_STAT_N || STAT == GTX_STAT_T || STAT == GTX_STAT_CORR || (STAT == GTX_STAT_RATIO && !TOTALS)) GTX_STAT(MODE_RANK);
  else return hipErrorInvalidValue;
#undef GTX_STAT
  return hipGetLastError();
}

template <int STAT>
hipError_t launch_stat_t(int mode, bool totals, bool hasVt, const StatArgs &a, unsigned grid, hipStream_t st)
{
  // counting statistics never read the totals; the others read them only under use_totals
  constexpr bool counting = STAT == GTX_STAT_N || STAT == GTX_STAT_SENS || STAT == GTX_STAT_SPEC;
  if (counting) return launch_stat_mode<STAT, false, false>(mode, a, grid, st);
  if (STAT == GTX_STAT_CORR) return hasVt ? launch_stat_mode<STAT, true, true>(mode, a, grid, st) : launch_stat_mode<STAT, true, false>(mode, a, grid, st);
  if (!totals) return launch_stat_mode<STAT, false, false>(mode, a, grid, st);
  // sum over all-ones totals: the total is the member count, exactly -- the no-totals form divides by the same number
  if (STAT == GTX_STAT_SUM && !hasVt) return launch_stat_mode<STAT, false, false>(mode, a, grid, st);
  return hasVt ? launch_stat_mode<STAT, true, true>(mode, a, grid, st) : launch_stat_mode<STAT, true, false>(mode, a, grid, st);
}

hipError_t launch_stat(int stat, int mode, bool totals, bool hasVt, const StatArgs &a, hipStream_t st)
{
  const i64 tiles = (a.nPerm + 63) / 64;
  const i64 blocks = 8 * ((tiles + 7) / 8) * a.colBlocks;             // 8 XCD lanes of ceil(tiles / 8) tiles each
  if (blocks <= 0) return hipSuccess;
  if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)blocks;
  switch (stat) {
  case GTX_STAT_SUM: return launch_stat_t<GTX_STAT_SUM>(mode, totals, hasVt, a, grid, st);
  case GTX_STAT_N: return launch_stat_t<GTX_STAT_N>(mode, totals, hasVt, a, grid, st);
  case GTX_STAT_SENS: return launch_stat_t<GTX_STAT_SENS>(mode, totals, hasVt, a, grid, st);
  case GTX_STAT_SPEC: return launch_stat_t<GTX_STAT_SPEC>(mode, totals, hasVt, a, grid, st);
  case GTX_STAT_RATIO: return launch_stat_t<GTX_STAT_RATIO>(mode, totals, hasVt, a, grid, st);
  case GTX_STAT_T: return launch_stat_t<GTX_STAT_T>(mode, totals, hasVt, a, grid, st);
  case GTX_STAT_CORR: return launch_stat_t<GTX_STAT_CORR>(mode, totals, hasVt, a, grid, st);
  }
  return hipErrorInvalidValue;
}

template <class T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct gtx_perm {
  int device = 0; hipStream_t stream = nullptr;
  std::string err;
  i64 nRows = -1, nCols = 0, nnz = 0;
  bool useTotals = false, hasVt = false;
  double sums[4] = {0, 0, 0, 0};
  i64 tPos = 0, tNeg = 0;
  float *d_V = nullptr, *d_Vt = nullptr; i64 *d_colPtr = nullptr; int32_t *d_rows = nullptr; uint32_t *d_rows16 = nullptr;
  float *d_Vp = nullptr, *d_Vtp = nullptr; size_t capSlab = 0;     // floats per slab
  int nParts = 1; i64 *d_colSplit = nullptr; u64 *d_accBuf = nullptr; size_t capAcc = 0;   // row-range parts of the stat kernel
  bool rowsAscending = true; std::vector<i64> h_colPtr; std::vector<int32_t> h_rows; int splitParts = 0;
  double *d_Y = nullptr; u64 *d_counts = nullptr;
  i64 *d_tabPtr = nullptr; double *d_tab = nullptr, *d_sortedY = nullptr; size_t capTab = 0;
  hipEvent_t ev[3] = {};
  float applyMs = 0, statMs = 0;
  size_t slabBytes = 1ull << 30;                                     // budget per slab batch (GTX_PERM_SLAB_MB)
};

#define PCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return GTX_E_HIP; } } while (0)
static int pfail(gtx_perm *p, int code, const char *msg) { p->err = msg; return code; }

extern "C" {

int gtx_perm_create(int device, gtx_perm **out)
{
  if (!out) return GTX_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GTX_E_HIP;   // no GPU: fail loudly, there is no CPU path
  gtx_perm *p = new gtx_perm;
  p->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete p; return GTX_E_HIP; }
  for (auto &e : p->ev) if (hipEventCreate(&e) != hipSuccess) { delete p; return GTX_E_HIP; }
  if (const char *s = getenv("GTX_PERM_SLAB_MB")) { long v = atol(s); if (v > 0) p->slabBytes = (size_t)v << 20; }
  *out = p;
  return GTX_OK;
}

void gtx_perm_destroy(gtx_perm *p)
{
  if (!p) return;
  (void)hipSetDevice(p->device);
  dfree(p->d_V); dfree(p->d_Vt); dfree(p->d_colPtr); dfree(p->d_rows); dfree(p->d_rows16); dfree(p->d_Vp); dfree(p->d_Vtp);
  dfree(p->d_Y); dfree(p->d_counts); dfree(p->d_tabPtr); dfree(p->d_tab); dfree(p->d_sortedY); dfree(p->d_colSplit); dfree(p->d_accBuf);
  for (auto &e : p->ev) if (e) (void)hipEventDestroy(e);
  delete p;
}

const char *gtx_perm_last_error(const gtx_perm *p) { return p ? p->err.c_str() : "null context"; }

int gtx_perm_set_table(gtx_perm *p, int64_t n_rows, int64_t n_cols, const int64_t *col_ptr, const int32_t *rows, const float *V,
                       const float *Vtotal, const double *sums, uint32_t flags)
{
  if (!p) return GTX_E_ARG;
  if (n_rows < 1 || n_rows >= (1ll << 24) || n_cols < 0 || !col_ptr || !V || !sums) return pfail(p, GTX_E_ARG, "gtx_perm_set_table: bad argument");
  if (col_ptr[0] != 0) return pfail(p, GTX_E_ARG, "gtx_perm_set_table: col_ptr[0] != 0");
  for (int64_t c = 0; c < n_cols; c++) if (col_ptr[c + 1] < col_ptr[c]) return pfail(p, GTX_E_ARG, "gtx_perm_set_table: col_ptr not monotone");
  const i64 nnz = col_ptr[n_cols];
  if (nnz > 0 && !rows) return pfail(p, GTX_E_ARG, "gtx_perm_set_table: rows == NULL");
  for (i64 z = 0; z < nnz; z++) if (rows[z] < 0 || rows[z] >= n_rows) return pfail(p, GTX_E_RANGE, "gtx_perm_set_table: row id out of range");
  PCHK(p, hipSetDevice(p->device));
  p->nRows = -1;
  dfree(p->d_V); dfree(p->d_Vt); dfree(p->d_colPtr); dfree(p->d_rows); dfree(p->d_rows16); dfree(p->d_Y); dfree(p->d_counts);
  dfree(p->d_tab); dfree(p->d_tabPtr); dfree(p->d_sortedY); p->capTab = 0;     // (sized by the table's categories)
  PCHK(p, hipMalloc(&p->d_V, sizeof(float) * n_rows));
  PCHK(p, hipMemcpy(p->d_V, V, sizeof(float) * n_rows, hipMemcpyHostToDevice));
  p->hasVt = Vtotal != nullptr;
  if (Vtotal) { PCHK(p, hipMalloc(&p->d_Vt, sizeof(float) * n_rows)); PCHK(p, hipMemcpy(p->d_Vt, Vtotal, sizeof(float) * n_rows, hipMemcpyHostToDevice)); }
  PCHK(p, hipMalloc(&p->d_colPtr, sizeof(i64) * (n_cols + 1)));
  PCHK(p, hipMemcpy(p->d_colPtr, col_ptr, sizeof(i64) * (n_cols + 1), hipMemcpyHostToDevice));
  PCHK(p, hipMalloc(&p->d_rows, sizeof(int32_t) * (nnz + 1)));
  if (nnz) PCHK(p, hipMemcpy(p->d_rows, rows, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
  if (n_rows <= 65536 && !(getenv("GTX_PERM_ROWS32") && atoi(getenv("GTX_PERM_ROWS32")))) {      // the lists once more, two row ids per word
    std::vector<uint32_t> packed((size_t)(nnz + 1) / 2 + 4, 0u);
    for (i64 z = 0; z < nnz; z++) packed[(size_t)z >> 1] |= (uint32_t)rows[z] << ((z & 1) * 16);
    PCHK(p, hipMalloc(&p->d_rows16, sizeof(uint32_t) * packed.size()));
    PCHK(p, hipMemcpy(p->d_rows16, packed.data(), sizeof(uint32_t) * packed.size(), hipMemcpyHostToDevice));
  }
  PCHK(p, hipMalloc(&p->d_Y, sizeof(double) * (n_cols + 1)));
  PCHK(p, hipMalloc(&p->d_counts, sizeof(u64) * (n_cols + 1)));
  p->tPos = p->tNeg = 0;
  for (int64_t r = 0; r < n_rows; r++) { p->tPos += V[r] > 0; p->tNeg += V[r] < 0; }       // t of :343-344, permutation invariant
  for (int i = 0; i < 4; i++) p->sums[i] = sums[i];
  p->h_colPtr.assign(col_ptr, col_ptr + n_cols + 1); p->h_rows.assign(rows, rows + nnz);
  p->rowsAscending = true; p->splitParts = 0;
  for (int64_t c = 0; c < n_cols && p->rowsAscending; c++)
    for (i64 z = col_ptr[c] + 1; z < col_ptr[c + 1]; z++) if (rows[z] < rows[z - 1]) { p->rowsAscending = false; break; }
  p->useTotals = (flags & GTX_PERM_USE_TOTALS) != 0;
  p->nRows = n_rows; p->nCols = n_cols; p->nnz = nnz;
  return GTX_OK;
}

static StatArgs base_args(gtx_perm *p, int under)
{
  StatArgs a = {};
  a.colPtr = p->d_colPtr; a.rows = p->d_rows; a.rows16 = p->d_rows16; a.nCols = p->nCols;
  a.k.Vsum = p->sums[0]; a.k.VsumZ = p->sums[1]; a.k.Vsum2 = p->sums[2]; a.k.VtotalSum = p->sums[3];
  a.k.nRows = p->nRows; a.k.under = under ? 1 : 0; a.k.tAll = under ? p->tNeg : p->tPos;
  a.colBlocks = (p->nCols + kWavesPerBlock - 1) / kWavesPerBlock;
  return a;
}

static int check_stat(gtx_perm *p, int stat)
{
  if (p->nRows < 0) return pfail(p, GTX_E_STATE, "no table: call gtx_perm_set_table first");
  if (stat < GTX_STAT_SUM || stat > GTX_STAT_CORR) return pfail(p, GTX_E_ARG, "unknown statistic");
  if (stat == GTX_STAT_CORR && !p->useTotals) return pfail(p, GTX_E_ARG, "corr needs use_totals (permutation_test.cpp:532-533)");
  return GTX_OK;
}

int gtx_perm_statistic(gtx_perm *p, int stat, int under, double *Y)
{
  if (!p || !Y) return GTX_E_ARG;
  if (int rc = check_stat(p, stat)) return rc;
  if (p->nCols == 0) return GTX_OK;
  PCHK(p, hipSetDevice(p->device));
  StatArgs a = base_args(p, under);
  a.Vp = p->d_V; a.Vtp = p->d_Vt; a.nPerm = 1; a.Yout = p->d_Y; a.nParts = 1; a.part = 0;
  PCHK(p, launch_stat(stat, MODE_STAT, p->useTotals, p->hasVt, a, p->stream));
  PCHK(p, hipMemcpyAsync(Y, p->d_Y, sizeof(double) * p->nCols, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  return GTX_OK;
}

static int check_approx(gtx_perm *p, int stat)
{
  if (int rc = check_stat(p, stat)) return rc;
  if (stat == GTX_STAT_T || stat == GTX_STAT_CORR || (stat == GTX_STAT_RATIO && !p->useTotals)) return GTX_OK;
  return pfail(p, GTX_E_ARG, "no distribution for this statistic (permutation_test.cpp:364, :387, :475, :502, :514: \"not implemented yet\"; `n` goes by table: gtx_perm_count_rank)");
}

int gtx_perm_statistic_approx(gtx_perm *p, int stat, int under, double *P)
{
  if (!p || !P) return GTX_E_ARG;
  if (int rc = check_approx(p, stat)) return rc;
  if (p->nCols == 0) return GTX_OK;
  PCHK(p, hipSetDevice(p->device));
  StatArgs a = base_args(p, under);
  a.Vp = p->d_V; a.Vtp = p->d_Vt; a.nPerm = 1; a.Yout = p->d_Y; a.nParts = 1; a.part = 0; a.approx = 1;
  PCHK(p, launch_stat(stat, MODE_STAT, p->useTotals, p->hasVt, a, p->stream));
  PCHK(p, hipMemcpyAsync(P, p->d_Y, sizeof(double) * p->nCols, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  return GTX_OK;
}

// permutations per slab batch: a multiple of 64 within the byte budget
static i64 batch_perms(gtx_perm *p, i64 n_perm)
{
  i64 pb = (i64)(p->slabBytes / (sizeof(float) * (size_t)p->nRows)) / 64 * 64;
  if (pb < 64) pb = 64;
  const i64 need = (n_perm + 63) / 64 * 64;
  return pb < need ? pb : need;
}

static int run_batches(gtx_perm *p, int stat, int mode, StatArgs a, bool needVt, uint64_t seed, int64_t first_perm, int64_t n_perm)
{
  const i64 pb = batch_perms(p, n_perm);
  const size_t need = (size_t)pb * (size_t)p->nRows;
  if (need > p->capSlab || (needVt && !p->d_Vtp)) {
    dfree(p->d_Vp); dfree(p->d_Vtp); p->capSlab = 0;
    PCHK(p, hipMalloc(&p->d_Vp, sizeof(float) * need));
    if (p->hasVt) PCHK(p, hipMalloc(&p->d_Vtp, sizeof(float) * need));
    p->capSlab = need;
  }
  // row-range parts of the stat kernel: the rows one launch gathers from must fit an XCD's L2 next to the lists (see perm_stat_kernel)
  int nParts = 1;
  {
    const char *lim = getenv("GTX_PERM_L2_MB");                      // (tests shrink it to cut small tables into many ranges)
    const double budget = (lim ? atof(lim) : 2.75) * 1024 * 1024;
    const double tileBytes = (double)p->nRows * 256.0 * (needVt ? 2 : 1);
    if (p->rowsAscending && budget > 0 && tileBytes > budget) nParts = (int)std::min<double>(16.0, ceil(tileBytes / budget));
  }
  if (nParts > 1) {
    if (p->splitParts != nParts) {
      std::vector<i64> split((size_t)(nParts + 1) * p->nCols);
      for (int q = 0; q <= nParts; q++) {
        const int32_t firstRow = (int32_t)((i64)p->nRows * q / nParts);
        for (i64 c = 0; c < p->nCols; c++)
          split[(size_t)q * p->nCols + c] = q == nParts ? p->h_colPtr[c + 1]
                                                         : std::lower_bound(p->h_rows.begin() + p->h_colPtr[c], p->h_rows.begin() + p->h_colPtr[c + 1], firstRow) - p->h_rows.begin();
      }
      dfree(p->d_colSplit);
      PCHK(p, hipMalloc(&p->d_colSplit, sizeof(i64) * split.size()));
      PCHK(p, hipMemcpy(p->d_colSplit, split.data(), sizeof(i64) * split.size(), hipMemcpyHostToDevice));
      p->splitParts = nParts;
    }
    const int words = stat == GTX_STAT_CORR ? 6 : (stat == GTX_STAT_T && p->useTotals) ? 4 : (stat == GTX_STAT_N || stat == GTX_STAT_SENS || stat == GTX_STAT_SPEC) ? 1
                      : (stat == GTX_STAT_SUM ? 2 : 2);
    const size_t needAcc = (size_t)((pb + 63) / 64) * (size_t)p->nCols * (size_t)words * 64;
    if (needAcc > p->capAcc) { dfree(p->d_accBuf); p->capAcc = 0; PCHK(p, hipMalloc(&p->d_accBuf, sizeof(u64) * needAcc)); p->capAcc = needAcc; }
  }
  const PermGeom g = perm_geom((uint32_t)p->nRows);
  p->applyMs = p->statMs = 0;
  for (i64 done = 0; done < n_perm; done += pb) {
    const i64 cnt = n_perm - done < pb ? n_perm - done : pb;
    PCHK(p, hipEventRecord(p->ev[0], p->stream));
    const bool noTab = getenv("GTX_PERM_NO_TABLE") != nullptr;                            // (the tests compare both writers)
    const uint32_t T = (std::max(g.a, g.b) + 1) & ~1u;                                   // (even: see the table's layout)
    if (!noTab && g.n > 16 && T <= (uint32_t)kTabMaxSide) {
      // one block per tile of 64 permutations and row range (one block per CU at a time: the table takes most of the LDS); the number
      // of row ranges balances full rounds of blocks over the CUs against tabulating once more per range
      static gtx::PerDevice attr;
      PCHK(p, attr.once([] {
        hipError_t e = hipFuncSetAttribute((const void *)perm_apply_tab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        return e != hipSuccess ? e : hipFuncSetAttribute((const void *)perm_apply_tabv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
      }));
      // the value vector in LDS beside a 32-permutation table, where both fit (perm_apply_tabv_kernel)
      const size_t ldsV = (size_t)((kRounds + 3) / 4) * T * 128 + sizeof(float) * (size_t)p->nRows;
      const bool noV = getenv("GTX_PERM_NO_LDS_VALUES") != nullptr;                        // (the tests compare the writers)
      const bool useV = !noV && ldsV <= 154 * 1024;
      const size_t lds = useV ? ldsV : (size_t)64 * kRounds * T;
      const unsigned tiles = useV ? (unsigned)((cnt + 31) / 32) : (unsigned)((cnt + 63) / 64);
      int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
      const unsigned maxChunks = (unsigned)std::max<i64>(1, p->nRows / 1024);
      unsigned chunks = 1; double best = 1e300;
      for (unsigned c = 1; c <= maxChunks && c <= 64; c++) {
        // rounds of blocks over the CUs x (rows of a block + what tabulating costs in rows: ~5.5 T by instruction count)
        const double cost = ceil((double)tiles * c / cus) * ((double)p->nRows / c + (useV ? 2.75 * T + p->nRows / 40.0 : 5.5 * T));   // (staging the values: ~n / 40 rows' worth)
        if (cost < best) { best = cost; chunks = c; }
      }
      if (getenv("GTX_PERM_CHUNKS")) chunks = (unsigned)std::max(1, atoi(getenv("GTX_PERM_CHUNKS")));
      const uint32_t rpbT = (uint32_t)((p->nRows + chunks - 1) / chunks);
      if (useV) perm_apply_tabv_kernel<<<dim3(tiles, (unsigned)((p->nRows + rpbT - 1) / rpbT)), 1024, lds, p->stream>>>(
          p->d_V, needVt ? p->d_Vt : nullptr, p->d_Vp, needVt ? p->d_Vtp : nullptr, g, rpbT, seed, first_perm + done, cnt, T);
      else perm_apply_tab_kernel<<<dim3(tiles, (unsigned)((p->nRows + rpbT - 1) / rpbT)), 1024, lds, p->stream>>>(
          p->d_V, needVt ? p->d_Vt : nullptr, p->d_Vp, needVt ? p->d_Vtp : nullptr, g, rpbT, seed, first_perm + done, cnt, T);
    } else {
      uint32_t rpb = 128;                                              // rows per block; grid.y stays below 65536
      while ((p->nRows + rpb - 1) / rpb > 65535) rpb *= 2;
      dim3 grid((unsigned)((cnt + 255) / 256), (unsigned)((p->nRows + rpb - 1) / rpb));
      perm_apply_kernel<<<grid, 256, 0, p->stream>>>(p->d_V, needVt ? p->d_Vt : nullptr, p->d_Vp, needVt ? p->d_Vtp : nullptr, g, rpb, seed,
                                                      first_perm + done, cnt);
    }
    PCHK(p, hipGetLastError());
    PCHK(p, hipEventRecord(p->ev[1], p->stream));
    a.Vp = p->d_Vp; a.Vtp = needVt ? p->d_Vtp : nullptr; a.nPerm = cnt;
    a.nParts = nParts; a.colSplit = p->d_colSplit; a.accBuf = p->d_accBuf;
    for (int part = 0; part < nParts; part++) { a.part = part; PCHK(p, launch_stat(stat, mode, p->useTotals, needVt, a, p->stream)); }
    PCHK(p, hipEventRecord(p->ev[2], p->stream));
    PCHK(p, hipEventSynchronize(p->ev[2]));
    float t0 = 0, t1 = 0;
    PCHK(p, hipEventElapsedTime(&t0, p->ev[0], p->ev[1]));
    PCHK(p, hipEventElapsedTime(&t1, p->ev[1], p->ev[2]));
    p->applyMs += t0; p->statMs += t1;
  }
  return GTX_OK;
}

// the totals slab is needed only by the statistics that read Vtotal
static bool stat_reads_totals(const gtx_perm *p, int stat)
{
  if (!p->hasVt) return false;
  if (stat == GTX_STAT_N || stat == GTX_STAT_SENS || stat == GTX_STAT_SPEC) return false;
  return stat == GTX_STAT_CORR || p->useTotals;
}

int gtx_perm_count_ge(gtx_perm *p, int stat, int under, const double *Y, uint64_t seed, int64_t first_perm, int64_t n_perm, uint64_t *counts)
{
  if (!p || !counts || !Y) return GTX_E_ARG;
  if (int rc = check_stat(p, stat)) return rc;
  if (first_perm < 0 || n_perm < 0) return pfail(p, GTX_E_ARG, "gtx_perm_count_ge: negative permutation range");
  for (i64 c = 0; c < p->nCols; c++) counts[c] = 0;
  if (p->nCols == 0 || n_perm == 0) return GTX_OK;
  PCHK(p, hipSetDevice(p->device));
  PCHK(p, hipMemcpyAsync(p->d_Y, Y, sizeof(double) * p->nCols, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemsetAsync(p->d_counts, 0, sizeof(u64) * p->nCols, p->stream));
  StatArgs a = base_args(p, under);
  a.Yobs = p->d_Y; a.counts = p->d_counts;
  if (int rc = run_batches(p, stat, MODE_GE, a, stat_reads_totals(p, stat), seed, first_perm, n_perm)) return rc;
  PCHK(p, hipMemcpyAsync(counts, p->d_counts, sizeof(u64) * p->nCols, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  return GTX_OK;
}

int gtx_perm_count_rank(gtx_perm *p, int under, const int64_t *tab_ptr, const double *tab, const double *sortedY, uint64_t seed,
                        int64_t first_perm, int64_t n_perm, uint64_t *counts)
{
  if (!p || !counts || !tab_ptr || !tab || !sortedY) return GTX_E_ARG;
  if (int rc = check_stat(p, GTX_STAT_N)) return rc;
  if (first_perm < 0 || n_perm < 0) return pfail(p, GTX_E_ARG, "gtx_perm_count_rank: negative permutation range");
  for (i64 c = 0; c < p->nCols; c++) counts[c] = 0;
  if (p->nCols == 0 || n_perm == 0) return GTX_OK;
  // every category's table must cover k = 0 .. n_c
  std::vector<i64> colPtr(p->nCols + 1);
  PCHK(p, hipSetDevice(p->device));
  PCHK(p, hipMemcpy(colPtr.data(), p->d_colPtr, sizeof(i64) * (p->nCols + 1), hipMemcpyDeviceToHost));
  for (i64 c = 0; c < p->nCols; c++)
    if (tab_ptr[c] < 0 || tab_ptr[c + 1] - tab_ptr[c] < colPtr[c + 1] - colPtr[c] + 1) return pfail(p, GTX_E_ARG, "gtx_perm_count_rank: table shorter than n_c + 1");
  const size_t nTab = (size_t)tab_ptr[p->nCols];
  if (nTab > p->capTab || !p->d_tabPtr) {
    dfree(p->d_tab); dfree(p->d_tabPtr); dfree(p->d_sortedY); p->capTab = 0;
    PCHK(p, hipMalloc(&p->d_tab, sizeof(double) * (nTab + 1)));
    PCHK(p, hipMalloc(&p->d_tabPtr, sizeof(i64) * (p->nCols + 1)));
    PCHK(p, hipMalloc(&p->d_sortedY, sizeof(double) * (p->nCols + 1)));
    p->capTab = nTab;
  }
  PCHK(p, hipMemcpyAsync(p->d_tab, tab, sizeof(double) * nTab, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemcpyAsync(p->d_tabPtr, tab_ptr, sizeof(i64) * (p->nCols + 1), hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemcpyAsync(p->d_sortedY, sortedY, sizeof(double) * p->nCols, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemsetAsync(p->d_counts, 0, sizeof(u64) * p->nCols, p->stream));
  StatArgs a = base_args(p, under);
  a.counts = p->d_counts; a.tabPtr = p->d_tabPtr; a.tab = p->d_tab; a.sortedY = p->d_sortedY;
  if (int rc = run_batches(p, GTX_STAT_N, MODE_RANK, a, false, seed, first_perm, n_perm)) return rc;
  PCHK(p, hipMemcpyAsync(counts, p->d_counts, sizeof(u64) * p->nCols, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  return GTX_OK;
}

int gtx_perm_count_rank_approx(gtx_perm *p, int stat, int under, const double *sortedY, uint64_t seed, int64_t first_perm, int64_t n_perm,
                               uint64_t *counts)
{
  if (!p || !counts || !sortedY) return GTX_E_ARG;
  if (int rc = check_approx(p, stat)) return rc;
  if (first_perm < 0 || n_perm < 0) return pfail(p, GTX_E_ARG, "gtx_perm_count_rank_approx: negative permutation range");
  for (i64 c = 0; c < p->nCols; c++) counts[c] = 0;
  if (p->nCols == 0 || n_perm == 0) return GTX_OK;
  PCHK(p, hipSetDevice(p->device));
  if (!p->d_sortedY) {
    dfree(p->d_tab); dfree(p->d_tabPtr); p->capTab = 0;
    PCHK(p, hipMalloc(&p->d_sortedY, sizeof(double) * (p->nCols + 1)));
  }
  PCHK(p, hipMemcpyAsync(p->d_sortedY, sortedY, sizeof(double) * p->nCols, hipMemcpyHostToDevice, p->stream));
  PCHK(p, hipMemsetAsync(p->d_counts, 0, sizeof(u64) * p->nCols, p->stream));
  StatArgs a = base_args(p, under);
  a.counts = p->d_counts; a.sortedY = p->d_sortedY; a.approx = 1;
  if (int rc = run_batches(p, stat, MODE_RANK, a, stat_reads_totals(p, stat), seed, first_perm, n_perm)) return rc;
  PCHK(p, hipMemcpyAsync(counts, p->d_counts, sizeof(u64) * p->nCols, hipMemcpyDeviceToHost, p->stream));
  PCHK(p, hipStreamSynchronize(p->stream));
  return GTX_OK;
}

int gtx_perm_permutation(gtx_perm *p, uint64_t seed, int64_t q, int32_t *out)
{
  if (!p || !out || q < 0) return GTX_E_ARG;
  if (p->nRows < 0) return pfail(p, GTX_E_STATE, "no table: call gtx_perm_set_table first");
  PCHK(p, hipSetDevice(p->device));
  int32_t *d = nullptr;
  PCHK(p, hipMalloc(&d, sizeof(int32_t) * p->nRows));
  perm_list_kernel<<<256, 256, 0, p->stream>>>(d, perm_geom((uint32_t)p->nRows), seed, q);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof(int32_t) * p->nRows, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  (void)hipFree(d);
  PCHK(p, e);
  return GTX_OK;
}

int gtx_perm_last_ms(gtx_perm *p, float *apply_ms, float *stat_ms)
{
  if (!p) return GTX_E_ARG;
  if (apply_ms) *apply_ms = p->applyMs;
  if (stat_ms) *stat_ms = p->statMs;
  return GTX_OK;
}

}  // extern "C"
