// gtx_sort.hip -- regions into position order on the device (what bin/sortbed and `genomic_regions gsort` are for
// in the reference -- RunGlobalSort genomic_intervals.cpp:4547-4570 over BinGenomicRegions :6095-6150 with CompareBinnedGenomicRegions
// :6044-6048).  Order: class ascending (the caller folds chromosome rank and, when it sorts by strand, the strand into the class, as
// everywhere in this library), start ascending, stop DESCENDING, and input order among regions that agree in all three -- the order
// the reference's bins + stable list sort produce.
//
// Shape of the work: the order is a permutation of 32-bit ordinals, found by two stable least-significant-digit radix sorts -- first
// by the stop (complemented: descending), then by (class, start) in one 64-bit key whose unused high digits are skipped -- and the
// triples are gathered through it once.  The digit passes are rocPRIM's device radix sort (the library sort of this platform, the
// way a GEMM would be rocBLAS's); building the keys, checking the classes and the gather are kernels of this file.  Nothing here is
// on the counting path: counting reads in no order goes through the partition of gtx_bucket.hip, which needs no order inside a bucket
// and is an order of magnitude cheaper than any sort; this entry point exists to WRITE sorted files (csrc/sortbed.cpp), where the
// text on either side is the bound.
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include "gtx.h"
#include "gtx_internal.h"

namespace {

typedef unsigned long long u64;
typedef long long i64;

// key of the second sort: class in the high word, start (biased to unsigned) in the low one; key of the first: the stop, complemented
__global__ __launch_bounds__(256) void sort_keys_kernel(const int *__restrict__ tri, i64 n, int nClasses, u64 *__restrict__ key, uint32_t *__restrict__ keyStop,
                                                         uint32_t *__restrict__ ord, unsigned *__restrict__ bad)
{
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const int c = __builtin_nontemporal_load(tri + 3 * i), s = __builtin_nontemporal_load(tri + 3 * i + 1), e = __builtin_nontemporal_load(tri + 3 * i + 2);
    if ((unsigned)c >= (unsigned)nClasses) atomicOr(bad, 1u);
    keyStop[i] = ~((uint32_t)e ^ 0x80000000u);
    ord[i] = (uint32_t)i;
    key[i] = ((u64)(uint32_t)c << 32) | ((uint32_t)s ^ 0x80000000u);       // (read again through the first sort's order: sort_rekey_kernel)
  }
}

// the (class, start) keys in the order the first sort left the regions in
__global__ __launch_bounds__(256) void sort_rekey_kernel(const u64 *__restrict__ key, const uint32_t *__restrict__ ord, i64 n, u64 *__restrict__ out)
{
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) out[i] = key[ord[i]];
}

__global__ __launch_bounds__(256) void sort_gather_kernel(const int *__restrict__ tri, const uint32_t *__restrict__ ord, i64 n, int *__restrict__ out)
{
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const i64 q = ord[i];
    const int c = tri[3 * q], s = tri[3 * q + 1], e = tri[3 * q + 2];
    __builtin_nontemporal_store(c, out + 3 * i); __builtin_nontemporal_store(s, out + 3 * i + 1); __builtin_nontemporal_store(e, out + 3 * i + 2);
  }
}

struct Scratch {
  void *p[8] = {};
  ~Scratch() { for (void *q : p) if (q) (void)hipFree(q); }
};

int sort_on_device(gtx_ctx *ctx, const int *d_tri, i64 n, int nClasses, uint32_t *d_order, int *d_sorted)
{
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gtxi_set_error(ctx, hipGetErrorString(e_)); return GTX_E_HIP; } } while (0)
  hipStream_t st = gtxi_stream(ctx);
  SCHK(hipSetDevice(gtxi_device(ctx)));
  // one block of the context's scratch (kept between calls: a call of 100 M reads would otherwise spend more time giving 4 GB back to
  // the driver than sorting), carved into the key, ordinal and digit-pass arrays
  int classBits = 0;
  while (classBits < 31 && (1ll << classBits) < (i64)nClasses) classBits++;
  size_t t1 = 0, t2 = 0;
  SCHK(rocprim::radix_sort_pairs(nullptr, t1, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, 32u, st));
  SCHK(rocprim::radix_sort_pairs(nullptr, t2, (u64 *)nullptr, (u64 *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, 32u + (unsigned)classBits, st));
  const size_t tmpBytes = t1 > t2 ? t1 : t2;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t k8 = up(sizeof(u64) * (size_t)n), k4 = up(sizeof(uint32_t) * (size_t)n);
  char *blk = nullptr;
  { void *q = nullptr; if (int rc = gtxi_scratch(ctx, 2 * k8 + 4 * k4 + up(tmpBytes) + 256, &q)) return rc; blk = (char *)q; }
  u64 *key = (u64 *)blk, *key2 = (u64 *)(blk + k8), *key3 = nullptr;
  uint32_t *ks = (uint32_t *)(blk + 2 * k8), *ks2 = (uint32_t *)(blk + 2 * k8 + k4), *ord = (uint32_t *)(blk + 2 * k8 + 2 * k4), *ord2 = (uint32_t *)(blk + 2 * k8 + 3 * k4);
  void *tmp = blk + 2 * k8 + 4 * k4;
  unsigned *bad = (unsigned *)(blk + 2 * k8 + 4 * k4 + up(tmpBytes));
  SCHK(hipMemsetAsync(bad, 0, sizeof(unsigned), st));
  const unsigned grid = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  sort_keys_kernel<<<grid, 256, 0, st>>>(d_tri, n, nClasses, key, ks, ord, bad);
  SCHK(hipGetLastError());
  SCHK(rocprim::radix_sort_pairs(tmp, t1, ks, ks2, ord, ord2, (size_t)n, 0u, 32u, st));                 // by stop, descending
  sort_rekey_kernel<<<grid, 256, 0, st>>>(key, ord2, n, key2);
  SCHK(hipGetLastError());
  key3 = key;                                                                                            // (its keys are in key2 now)
  SCHK(rocprim::radix_sort_pairs(tmp, t2, key2, key3, ord2, d_order, (size_t)n, 0u, 32u + (unsigned)classBits, st));   // by (class, start); stable
  if (d_sorted) { sort_gather_kernel<<<grid, 256, 0, st>>>(d_tri, d_order, n, d_sorted); SCHK(hipGetLastError()); }
  unsigned h_bad = 0;
  SCHK(hipMemcpyAsync(&h_bad, bad, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  SCHK(hipStreamSynchronize(st));
  if (h_bad) { gtxi_set_error(ctx, "gtx_sort: a class id outside [0, n_classes)"); return GTX_E_RANGE; }
  return GTX_OK;
#undef SCHK
}

}  // namespace

extern "C" {

int gtx_sort_device(gtx_ctx *ctx, const void *d_reads, int64_t n_reads, int32_t n_classes, void *d_order, void *d_sorted)
{
  if (!ctx) return GTX_E_ARG;
  if (n_reads < 0 || n_reads >= (1ll << 32) || n_classes < 1 || (n_reads > 0 && (!d_reads || !d_order))) { gtxi_set_error(ctx, "gtx_sort_device: bad argument (n_reads < 2^32, n_classes >= 1)"); return GTX_E_ARG; }
  if (n_reads == 0) return GTX_OK;
  return sort_on_device(ctx, (const int *)d_reads, n_reads, n_classes, (uint32_t *)d_order, (int *)d_sorted);
}

int gtx_sort(gtx_ctx *ctx, const int32_t *read_triples, int64_t n_reads, int32_t n_classes, uint32_t *order_out, int32_t *sorted_out)
{
  if (!ctx) return GTX_E_ARG;
  if (n_reads < 0 || n_reads >= (1ll << 32) || n_classes < 1 || (n_reads > 0 && (!read_triples || !order_out))) { gtxi_set_error(ctx, "gtx_sort: bad argument (n_reads < 2^32, n_classes >= 1)"); return GTX_E_ARG; }
  if (n_reads == 0) return GTX_OK;
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gtxi_set_error(ctx, hipGetErrorString(e_)); return GTX_E_HIP; } } while (0)
  HCHK(hipSetDevice(gtxi_device(ctx)));
  Scratch s;
  HCHK(hipMalloc(&s.p[0], sizeof(int32_t) * 3 * (size_t)n_reads));
  HCHK(hipMalloc(&s.p[1], sizeof(uint32_t) * (size_t)n_reads));
  if (sorted_out) HCHK(hipMalloc(&s.p[2], sizeof(int32_t) * 3 * (size_t)n_reads));
  hipStream_t st = gtxi_stream(ctx);
  HCHK(hipMemcpyAsync(s.p[0], read_triples, sizeof(int32_t) * 3 * (size_t)n_reads, hipMemcpyHostToDevice, st));
  if (int rc = sort_on_device(ctx, (const int *)s.p[0], n_reads, n_classes, (uint32_t *)s.p[1], (int *)s.p[2])) return rc;
  HCHK(hipMemcpyAsync(order_out, s.p[1], sizeof(uint32_t) * (size_t)n_reads, hipMemcpyDeviceToHost, st));
  if (sorted_out) HCHK(hipMemcpyAsync(sorted_out, s.p[2], sizeof(int32_t) * 3 * (size_t)n_reads, hipMemcpyDeviceToHost, st));
  HCHK(hipStreamSynchronize(st));
  return GTX_OK;
#undef HCHK
}

}  // extern "C"
