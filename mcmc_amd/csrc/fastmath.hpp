// FP64 transcendental kernels for the SED hot path, written for the CDNA4 VALU: straight-line
// FMA chains, no tables, no divergent branches (special cases by select).  Accuracy is measured
// against mpmath in tests/test_fastmath_gpu.py (<= 2 ulp on the domains the models use), which
// keeps lnL within ~1e-14 relative of the reference (budget: 1e-12, BASELINE.json north_star).
//
// Why not OCML here: pow() recomputes ln(base) to double-double precision on every call, but the
// bases of the hot path (nu/nu_b per photometric band, 10, nu_b) are launch constants, so the host
// stores ln(base) as a double-double once (mi355_sed_load) and a power costs one exp.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

constexpr double kLog2e = 0x1.71547652b82fep+0;
constexpr double kLn2Hi = 0x1.62e42fefa39efp-1;    // ln 2 = hi + lo
constexpr double kLn2Lo = 0x1.abc9e3b39803fp-56;
constexpr double kLn10Hi = 0x1.26bb1bbb55516p+1;   // ln 10 = hi + lo
constexpr double kLn10Lo = -0x1.f48ad494ea3e9p-53;
constexpr double kExpMax = 690.0;                  // arguments are clamped here: e^690 ~ 1e299

// expm1(r) for |r| <= 0.355:  r + r^2 * Q(r), Q of degree 10 (Chebyshev-node fit of
// (expm1(r)-r)/r^2, max relative error of the whole expression 2.1e-18).
__device__ __forceinline__ double expm1_reduced(double r) {
  double q = 0x1.1f798f28decdap-29;
  q = fma(q, r, 0x1.af595f03ff6c3p-26);
  q = fma(q, r, 0x1.27e4d82b5badcp-22);
  q = fma(q, r, 0x1.71ddfc8d5d234p-19);
  q = fma(q, r, 0x1.a01a01a7e3268p-16);
  q = fma(q, r, 0x1.a01a01adbaae7p-13);
  q = fma(q, r, 0x1.6c16c16c160e0p-10);
  q = fma(q, r, 0x1.111111110fd70p-7);
  q = fma(q, r, 0x1.5555555555556p-5);
  q = fma(q, r, 0x1.5555555555557p-3);
  q = fma(q, r, 0x1.0000000000000p-1);
  return fma(r * r, q, r);
}

// x = n*ln2 + r, |r| <= ln2/2 (+ |extra|); `extra` is a small correction added to r (the low
// part of a double-double argument), n returned as an integer for ldexp.
__device__ __forceinline__ double reduce_ln2(double x, double extra, int& n) {
  const double fn = rint(x * kLog2e);
  double r = fma(-fn, kLn2Hi, x);
  r = fma(-fn, kLn2Lo, r);
  n = static_cast<int>(fn);
  return r + extra;
}

// e^(x + extra) for x in [-kExpMax, kExpMax] (clamped), |extra| << 1.
__device__ __forceinline__ double exp_dd(double x, double extra) {
  x = fmin(fmax(x, -kExpMax), kExpMax);
  int n;
  const double r = reduce_ln2(x, extra, n);
  return ldexp(1.0 + expm1_reduced(r), n);
}

__device__ __forceinline__ double exp_fast(double x) { return exp_dd(x, 0.0); }

// base^e with ln(base) = lhi + llo known to double-double precision.
__device__ __forceinline__ double pow_prelog(double e, double lhi, double llo) {
  const double p = e * lhi;
  const double pe = fma(e, lhi, -p);       // exact low part of the product
  return exp_dd(p, fma(e, llo, pe));
}

__device__ __forceinline__ double exp10_fast(double x) { return pow_prelog(x, kLn10Hi, kLn10Lo); }

// expm1(x), any finite x (clamped to [-40, kExpMax]: below -40 the result is -1 to the last bit):
//   2^n (1+q) - 1 = 2^n * (q + (1 - 2^-n)),  q = expm1(r); the inner sum is exact-ish (one
//   rounding), so there is no cancellation for any n.
__device__ __forceinline__ double expm1_fast(double x) {
  x = fmin(fmax(x, -40.0), kExpMax);
  int n;
  const double r = reduce_ln2(x, 0.0, n);
  const double q = expm1_reduced(r);
  const double t = 1.0 - ldexp(1.0, -n);
  return ldexp(q + t, n);
}

// a/b by Newton-Raphson on v_rcp_f64 plus one residual correction (Markstein): correctly rounded
// in all but ~1e-3 of cases (then 1 ulp off), for b and a/b in the normal range.  No scaling /
// fix-up for subnormal or overflowing quotients - callers keep operands away from those.
__device__ __forceinline__ double rcp_nr(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  return r;
}

__device__ __forceinline__ double div_rcp(double a, double b, double rb) {   // rb ~ 1/b
  const double q = a * rb;
  const double rem = fma(-q, b, a);
  return fma(rem, rb, q);
}

__device__ __forceinline__ double div_fast(double a, double b) { return div_rcp(a, b, rcp_nr(b)); }

}  // namespace mi355
