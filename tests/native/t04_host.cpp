// Host build of stanford_raytracer_amd/csrc/srt_t04.hpp for the CPU tests (tests/test_t04_host.py): the same source the
// device compiles, checked on the CPU against goldens captured from the reference's T04_s / EXTERN.
#include "../../stanford_raytracer_amd/csrc/srt_t04.hpp"
using namespace srt::t04;
extern "C" void t04h_components(const double *in, double *out) {
  const Components c = external_field(T04D_T04_S_A, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9], in[10], in[11],
                                      in[12], in[13]);
  const V3 *v = &c.cf;
  for (int i = 0; i < 11; ++i) {
    out[3 * i] = v[i].x;
    out[3 * i + 1] = v[i].y;
    out[3 * i + 2] = v[i].z;
  }
}
extern "C" void t04h_t04s(const float *parmod, float ps, float x, float y, float z, float *out) { t04_s(parmod, ps, x, y, z, out[0], out[1], out[2]); }
extern "C" const double *t04h_A(void) { return T04D_T04_S_A; }
