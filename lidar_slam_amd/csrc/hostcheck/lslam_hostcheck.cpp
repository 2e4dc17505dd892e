// Host build of lslam_solve.h for the CPU unit tests (tests/test_solver_host.py).  Not part of the
// product library: the device code includes the very same header.
#include "../lslam_solve.h"

extern "C" {

// moments: n, sp[3], sq[3], spq[9], sse, lo[3], hi[3]  (24 doubles)
static lslam::Moments unpack(const double* m) {
  lslam::Moments mo;
  mo.n = m[0];
  for (int i = 0; i < 3; ++i) { mo.sp[i] = m[1 + i]; mo.sq[i] = m[4 + i]; mo.lo[i] = m[17 + i]; mo.hi[i] = m[20 + i]; }
  for (int i = 0; i < 9; ++i) mo.spq[i] = m[7 + i];
  mo.sse = m[16];
  return mo;
}

void lslam_hostcheck_step(const double* moments, int planar, double* out_T16) {
  lslam::step_from_moments(unpack(moments), planar != 0, out_T16);
}

double lslam_hostcheck_error(const double* moments, int normalize) {
  return lslam::error_from_moments(unpack(moments), normalize != 0);
}

// feeds a sequence of errors through the stop rules; returns the number of iterations executed
int lslam_hostcheck_stop(const double* errs, int count, double epsilon, double thresh, int max_iters) {
  lslam::StopState s;
  lslam::stop_init(s);
  for (int i = 0; i < count; ++i)
    if (lslam::stop_after_iteration(s, errs[i], epsilon, thresh, max_iters)) break;
  return s.executed;
}

void lslam_hostcheck_matmul4(const double* A, const double* B, double* C) { lslam::matmul4(A, B, C); }
}
