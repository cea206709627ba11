// dma_probe.hip -- where do the bytes of global_load_lds_dwordx3 land in LDS?  (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void probe(const int *src, int *out)
{
  __shared__ int lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = -1;
  __syncthreads();
  const char *p = (const char *)src + lane * 12;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p, (__attribute__((address_space(3))) void *)&lds[0], 12, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 768), (__attribute__((address_space(3))) void *)&lds[192], 12, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main()
{
  std::vector<int> h(4096); for (int i = 0; i < 4096; i++) h[i] = i;
  int *d, *o; CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&o, 1024 * 4));
  CK(hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  probe<<<1, 64>>>(d, o); CK(hipDeviceSynchronize());
  std::vector<int> r(1024); CK(hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 384; i++) if (r[i] != i) bad++;
  printf("mismatches against the lane x 12 layout: %d of 384\n", bad);
  for (int i = 0; i < 400; i++) { printf("%d ", r[i]); if (i % 32 == 31) printf("\n"); }
  printf("\n");
  return 0;
}
