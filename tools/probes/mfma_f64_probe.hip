// Probe of v_mfma_f64_16x16x4f64 operand / accumulator layout on gfx950 (run on the GPU box):
//   D[16][16] += A[16][4] * B[4][16]; which (row, k) does lane l supply for A, which (k, col) for B, and which
//   (row, col) does register r of lane l hold for D?  Also: does the instruction honour EXEC (inactive lanes)?
// build: hipcc -O2 --offload-arch=gfx950 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(const double *a, const double *b, double *out, int half) {
  const int lane = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  if (!half || lane < 32) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
int main() {
  // hypothesis: lane l supplies A[i = l%16][k = l/16], B[k = l/16][j = l%16]
  std::vector<double> A(64), B(64), D(256);
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 2048);
  int bad = 0;
  for (int k0 = 0; k0 < 4; ++k0) {
    for (int l = 0; l < 64; ++l) {
      int i = l % 16, k = l / 16;
      A[l] = (k == k0) ? (i + 1) : 0.0;        // A[i][k0] = i+1
      B[l] = (k == k0) ? 100.0 * (i + 1) : 0.0; // B[k0][j] = 100 (j+1)
    }
    hipMemcpy(da, A.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(db, B.data(), 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dd, 0);
    hipMemcpy(D.data(), dd, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        // expected under hypothesis D: lane l reg r = D[i = 4r + l/16][j = l%16] = (i+1) * 100 (j+1)
        int i = 4 * r + l / 16, j = l % 16;
        double want = (i + 1) * 100.0 * (j + 1);
        if (D[l * 4 + r] != want) {
          if (bad < 8) printf("k0=%d lane %d reg %d: got %g want %g\n", k0, l, r, D[l * 4 + r], want);
          ++bad;
        }
      }
  }
  printf("layout hypothesis (A[l%%16][l/16], B[l/16][l%%16], D reg r of lane l = [4r + l/16][l%%16]): %s\n", bad ? "WRONG" : "CONFIRMED");
  if (bad) { // dump one case to decode
    for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, D[l * 4], D[l * 4 + 1], D[l * 4 + 2], D[l * 4 + 3]);
  }
  { // full random check against a host product
    std::vector<double> Am(64), Bm(64);
    for (int l = 0; l < 64; ++l) { Am[l] = (double)((l * 37 + 11) % 23) - 9.0; Bm[l] = (double)((l * 53 + 5) % 19) - 7.0; }
    hipMemcpy(da, Am.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(db, Bm.data(), 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dd, 0);
    hipMemcpy(D.data(), dd, 2048, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        int i = 4 * r + l / 16, j = l % 16;
        double want = 0;
        for (int k = 0; k < 4; ++k) want += Am[16 * k + i] * Bm[16 * k + j]; // A[i][k] from lane 16k+i, B[k][j] from lane 16k+j
        if (D[l * 4 + r] != want) ++bad2;
      }
    printf("random product check: %s\n", bad2 ? "WRONG" : "CONFIRMED");
    bad += bad2;
  }
  // EXEC: only lanes 0..31 execute the instruction
  for (int l = 0; l < 64; ++l) { A[l] = 1.0; B[l] = 1.0; }
  hipMemcpy(da, A.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(db, B.data(), 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dd, 1);
  hipMemcpy(D.data(), dd, 2048, hipMemcpyDeviceToHost);
  printf("half-EXEC: lane0 %g lane31 %g lane32 %g lane63 %g (full sum would be 4)\n", D[0], D[31 * 4], D[32 * 4], D[63 * 4]);
  return bad != 0;
}
