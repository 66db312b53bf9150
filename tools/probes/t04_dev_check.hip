// Device-side check of srt_t04.hpp against numbers given on the command line input file (rows: parmod(10) ps x y z bx by bz)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../stanford_raytracer_amd/csrc/srt_t04.hpp"
__global__ void k(const float *in, float *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *r = in + 17 * i;
  float bx, by, bz;
  srt::t04::t04_s(r, r[10], r[11], r[12], r[13], bx, by, bz);
  out[3 * i] = bx; out[3 * i + 1] = by; out[3 * i + 2] = bz;
}
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "r");
  std::vector<float> in;
  float v;
  while (fscanf(f, "%f", &v) == 1) in.push_back(v);
  int n = in.size() / 17;
  float *di, *dout;
  hipMalloc(&di, in.size() * 4); hipMalloc(&dout, n * 12);
  hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  k<<<(n + 63) / 64, 64>>>(di, dout, n);
  std::vector<float> o(3 * n);
  hipError_t e = hipMemcpy(o.data(), dout, n * 12, hipMemcpyDeviceToHost);
  printf("hip: %s\n", hipGetErrorString(e));
  double worst = 0; int bad = 0;
  for (int i = 0; i < n; ++i) {
    double nrm = 0, d = 0;
    for (int c = 0; c < 3; ++c) { nrm += in[17 * i + 14 + c] * in[17 * i + 14 + c]; double t = o[3 * i + c] - in[17 * i + 14 + c]; d = fmax(d, fabs(t)); }
    double rel = d / sqrt(nrm);
    if (rel > 1e-6) { if (bad < 5) printf("row %d: got %g %g %g want %g %g %g\n", i, o[3*i], o[3*i+1], o[3*i+2], in[17*i+14], in[17*i+15], in[17*i+16]); ++bad; }
    worst = fmax(worst, rel);
  }
  printf("n=%d worst rel %.3g bad %d\n", n, worst, bad);
  return 0;
}
