// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950 (run on the GPU box): operand / result layout of its four 4x4x4 blocks, and
// whether the matrix pipe runs beside the vector fp64 pipe (VALU-only, MFMA-only and interleaved loops, whole chip).
// build: hipcc -O2 --offload-arch=gfx950 -o mfma_f64_4x4_probe mfma_f64_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(const double *a, const double *b, double *out) {
  const int lane = threadIdx.x;
  double acc = 0.0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], acc, 0, 0, 0);
  out[lane] = acc;
}
template <int MODE>
__global__ __launch_bounds__(64) void rate(double *out, int iters) {
  const int lane = threadIdx.x;
  double v[16], m[8];
  for (int i = 0; i < 16; ++i) v[i] = 1.0 + 1e-9 * (lane + i);
  for (int i = 0; i < 8; ++i) m[i] = 0.0;
  const double x = 1.0 + 1e-12 * lane, y = 1e-13 * (lane + 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = fma(v[i], x, y);   // 16 independent fp64 FMAs
    }
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, m[i], 0, 0, 0);  // 4 independent 4x4x4_4b (256 FMAs each)
    }
    if (MODE & 4) {
      typedef double d4 __attribute__((ext_vector_type(4)));
      d4 q = {m[4], m[5], m[6], m[7]};
      q = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, q, 0, 0, 0);  // one 16x16x4 (1024 FMAs)
      m[4] = q[0], m[5] = q[1], m[6] = q[2], m[7] = q[3];
    }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += m[i];
  out[blockIdx.x * 64 + lane] = s;
}
int main() {
  std::vector<double> A(64), B(64), D(64);
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // decode: set one A and one B element at a time would take 64*64 runs; instead use a random product and test hypotheses
  for (int l = 0; l < 64; ++l) { A[l] = (double)((l * 37 + 11) % 23) - 9.0; B[l] = (double)((l * 53 + 5) % 19) - 7.0; }
  hipMemcpy(da, A.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(db, B.data(), 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dd);
  hipMemcpy(D.data(), dd, 512, hipMemcpyDeviceToHost);
  // hypotheses: lane l: block = l / 16; A element (i = l % 4, k = (l / 4) % 4) or (i = (l/4)%4, k = l%4); same for B; D lane l holds (i, j) = (l%4, (l/4)%4) or swapped
  int found = 0;
  for (int ha = 0; ha < 2; ++ha) for (int hb = 0; hb < 2; ++hb) for (int hd = 0; hd < 2; ++hd) {
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      int blk = l / 16, lo = l % 4, hi = (l / 4) % 4;
      int i = hd ? hi : lo, j = hd ? lo : hi;
      double want = 0;
      for (int k = 0; k < 4; ++k) {
        int la = blk * 16 + (ha ? (i * 4 + k) : (k * 4 + i));   // ha=0: lane = 4k + i; ha=1: lane = 4i + k
        int lb = blk * 16 + (hb ? (j * 4 + k) : (k * 4 + j));
        want += A[la] * B[lb];
      }
      if (D[l] != want) ++bad;
    }
    if (!bad) { printf("layout: A lane-in-block = %s, B lane-in-block = %s, D lane-in-block: (i,j) = %s   CONFIRMED\n", ha ? "4i+k" : "4k+i", hb ? "4j+k" : "4k+j", hd ? "(hi,lo)" : "(lo,hi)"); ++found; }
  }
  if (!found) { printf("no layout hypothesis matched; D:"); for (int l = 0; l < 64; ++l) printf(" %g", D[l]); printf("\n"); }
  // rates: whole chip, 1024 and 2048 one-wave blocks
  double *big; hipMalloc(&big, 4096 * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200000;
  for (int blocks : {1024, 2048}) {
    float ms[8] = {0};
    for (int mode : {1, 2, 3, 4, 5}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 1) rate<1><<<blocks, 64>>>(big, iters);
        if (mode == 2) rate<2><<<blocks, 64>>>(big, iters);
        if (mode == 3) rate<3><<<blocks, 64>>>(big, iters);
        if (mode == 4) rate<4><<<blocks, 64>>>(big, iters);
        if (mode == 5) rate<5><<<blocks, 64>>>(big, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
      }
    }
    const double valu_fma = (double)blocks * 64 * 16 * iters, mf4 = (double)blocks * 4 * 256 * iters, mf16 = (double)blocks * 1024.0 * iters;
    printf("%d waves: VALU only %.1f ms (%.1f TFLOP/s) | 4x4x4 only %.1f ms (%.1f TFLOP/s) | VALU + 4x4x4 %.1f ms | 16x16x4 only %.1f ms (%.1f TFLOP/s) | VALU + 16x16x4 %.1f ms\n",
           blocks, ms[1], 2 * valu_fma / ms[1] * 1e-9, ms[2], 2 * mf4 / ms[2] * 1e-9, ms[3], ms[4], 2 * mf16 / ms[4] * 1e-9, ms[5]);
  }
  return found ? 0 : 1;
}
