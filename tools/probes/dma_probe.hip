// Probe of LDS-DMA semantics on gfx950 (run on the GPU box):
//  1. does the immediate offset of global_load_lds_dwordx4 apply to the LDS address as well as the global one?
//  2. are several DMA batches retired in issue order (s_waitcnt vmcnt(N) usable for double buffering)?
// build: hipcc -O2 --offload-arch=gfx950 -o dma_probe dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(const double *src, double *out) {
  __shared__ __attribute__((aligned(16))) double tile[1024]; // 8 KiB
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) tile[i] = -1.0;
  __syncthreads();
  // lane L reads src[2L .. 2L+1] + imm 256 B (= 32 doubles)   -> LDS (tile + 128 doubles) [+ imm?] + L*16
  const char *g = reinterpret_cast<const char *>(src) + lane * 16;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)(tile + 128), 16, 256, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = tile[i];
}
int main() {
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  double *d, *o;
  hipMalloc(&d, 4096 * 8);
  hipMalloc(&o, 1024 * 8);
  hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o);
  std::vector<double> r(1024);
  hipMemcpy(r.data(), o, 1024 * 8, hipMemcpyDeviceToHost);
  int first = -1;
  for (int i = 0; i < 1024; ++i)
    if (r[i] >= 0) { first = i; break; }
  printf("first written LDS double index: %d (128 => imm NOT applied to LDS, 160 => applied), value %.0f (32 => imm applied to global)\n",
         first, first >= 0 ? r[first] : -1.0);
  int n = 0;
  for (int i = 0; i < 1024; ++i) n += r[i] >= 0;
  printf("doubles written: %d (expect 128)\n", n);
  return 0;
}
