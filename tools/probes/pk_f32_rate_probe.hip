// tools/probes/pk_f32_rate_probe.hip -- issue rate of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 against v_mul_f32 / v_fma_f32 /
// v_fma_f64 on gfx950, with 1, 2 and 4 waves per SIMD and 4, 8 or 16 independent chains per wave.
//   hipcc -O3 --offload-arch=gfx950 -o pk_probe pk_f32_rate_probe.hip && ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2_t __attribute__((ext_vector_type(2)));

template <int OP, int CH>
__global__ void probe(float *out, int iters, float seed) {
  f2_t a[CH];
  double da[CH];
  const f2_t b = {seed, seed * 1.0001f}, c = {1e-3f, 2e-3f};
#pragma unroll
  for (int i = 0; i < CH; ++i) { a[i] = f2_t{(float)threadIdx.x + i, (float)i}; da[i] = (double)threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (OP == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(da[i]) : "v"((double)1.0000001), "v"((double)1e-3));
        if (OP == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da[i]) : "v"((double)1.0000001));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += a[i].x + a[i].y + (float)da[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
void run(const char *name, int waves_per_simd) {
  float *out;
  const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
  hipMalloc(&out, blocks * 64 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP, CH><<<blocks, 64>>>(out, 100, 1.0000001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<OP, CH><<<blocks, 64>>>(out, iters, 1.0000001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 8 * CH * waves_per_simd; // per SIMD
  printf("%-14s chains %2d waves/SIMD %d: %.3f ms, %.2f ns per instruction and SIMD (= %.2f cycles at 2.4 GHz)\n", name, CH, waves_per_simd, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
  hipFree(out);
}
#define ALL(OP, NAME) run<OP, 1>(NAME, 1); run<OP, 2>(NAME, 1); run<OP, 1>(NAME, 2); run<OP, 2>(NAME, 2); run<OP, 4>(NAME, 1); run<OP, 8>(NAME, 1); run<OP, 16>(NAME, 1); run<OP, 8>(NAME, 2); run<OP, 8>(NAME, 4);
int main() {
  ALL(0, "v_pk_mul_f32") ALL(1, "v_pk_add_f32") ALL(2, "v_pk_fma_f32") ALL(3, "v_mul_f32") ALL(4, "v_fma_f32") ALL(5, "v_fma_f64") ALL(6, "v_mul_f64")
  return 0;
}
