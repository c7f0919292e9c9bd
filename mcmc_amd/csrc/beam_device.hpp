// Device helpers of the beam model shared by the likelihood kernels and the step-wise
// Metropolis-Hastings engine: package (OffsetNormalizedBeamModel.py:123-133) and prior
// (OffsetNormalizedBeamModel.py:93-101 -> BeamModel.py:165-187).
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

constexpr int kBeamP = 9;        // x, y, sigma1, sigma2, angle, amplitude, offset, mu, phi
constexpr double kPi = 3.141592653589793;

struct BeamPriorBox {            // class-level prior state (BeamModel.py:41-43, ProcessBeamData.py:149-151)
  double cmin[2], cmax[2];
  double sig_min, sig_max;
  int has_center, has_sig_min, has_sig_max;
};

// numpy / Python float remainder: result takes the sign of the divisor
__device__ __forceinline__ double py_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}

// package() then unpackage() of a flat vector, OffsetNormalizedBeamModel.py:123-133
__device__ __forceinline__ void beam_package(double* th) {
  th[4] = py_mod(th[4], kPi);
  th[7] = fabs(th[7]);
  th[8] = py_mod(th[8], 2 * kPi);
}

// OffsetNormalizedBeamModel.prior -> GaussianBeamModel2D.prior (packaged parameters)
__device__ __forceinline__ double beam_prior(const double* th, const BeamPriorBox& b) {
  if (th[5] <= 0) return 0.0;                                        // :95
  if (th[7] < -1 || th[7] > 1) return 0.0;                           // :96-97
  if (b.has_center && (th[0] < b.cmin[0] || th[1] < b.cmin[1] || th[0] > b.cmax[0] ||
                       th[1] > b.cmax[1]))
    return 0.0;                                                      // BeamModel.py:171-174
  if (!b.has_sig_min) {
    if (th[2] < 0 || th[3] < 0) return 0.0;                          // :176-177
  } else if (th[2] < b.sig_min || th[3] < b.sig_min) {
    return 0.0;                                                      // :178-179
  }
  if (b.has_sig_max && fmax(th[2], th[3]) > b.sig_max) return 0.0;   // :181-182
  return 1.0;
}

}  // namespace mi355
