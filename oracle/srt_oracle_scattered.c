/* srt_oracle_scattered.c -- TEST INFRASTRUCTURE ONLY.  Placeholder until the kd-tree/MLS restatement lands. */
#include <stdlib.h>
#include "srt_oracle.h"
#include "srt_oracle_internal.h"
void so_scattered_params(struct so_model *m, const double x[3], double qs[4], double Ns[4],
                         double ms[4], double nus[4]) { (void)m; (void)x; (void)qs; (void)Ns; (void)ms; (void)nus; abort(); }
void so_scattered_free(struct so_model *m) { (void)m; }
so_model *so_model_create_scattered_file(const char *ptsfile, int yearday, int msec,
                                         double window_scale, int order, int exact,
                                         double local_window_scale, unsigned perm_seed) {
  (void)ptsfile; (void)yearday; (void)msec; (void)window_scale; (void)order; (void)exact; (void)local_window_scale; (void)perm_seed;
  return NULL;
}
