// FP64 transcendental kernels for the SED hot path, written for the CDNA4 VALU: straight-line
// FMA chains, no tables, no divergent branches (special cases by select).  Accuracy is measured
// against mpmath in tests/test_fastmath_gpu.py (<= 2.5 ulp on the domains the models use), which
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

// expm1(r) for |r| <= 0.3475:  r + r^2 * Q(r), Q of degree 9 (interpolation of (expm1(r)-r)/r^2
// at the Chebyshev nodes of the interval; max relative error of the whole expression 4.1e-17,
// under half an ulp - one FMA fewer than the degree-10 fit this replaced).
__device__ __forceinline__ double expm1_reduced(double r) {
  double q = 0x1.af39c20098071p-26;
  q = fma(q, r, 0x1.28926932adb78p-22);
  q = fma(q, r, 0x1.71de0d383914ap-19);
  q = fma(q, r, 0x1.a019b84d17a80p-16);
  q = fma(q, r, 0x1.a01a01a7e29cdp-13);
  q = fma(q, r, 0x1.6c16c178bd6bcp-10);
  q = fma(q, r, 0x1.111111111098cp-7);
  q = fma(q, r, 0x1.5555555553ce3p-5);
  q = fma(q, r, 0x1.5555555555556p-3);
  q = fma(q, r, 0x1.0000000000001p-1);
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

// e^(x + extra) for x in [-kExpMax, kExpMax] (clamped), |extra| << 1.  CLAMP = false is for
// callers that have bounded |x| themselves (mi355_sed_load checks the catalogue against the prior
// box, see SedConsts::fast_ok): two instructions fewer per call.
template <bool CLAMP = true>
__device__ __forceinline__ double exp_dd(double x, double extra) {
  if (CLAMP) x = fmin(fmax(x, -kExpMax), kExpMax);
  int n;
  const double r = reduce_ln2(x, extra, n);
  return ldexp(1.0 + expm1_reduced(r), n);
}

__device__ __forceinline__ double exp_fast(double x) { return exp_dd<true>(x, 0.0); }

// base^e with ln(base) = lhi + llo known to double-double precision.
template <bool CLAMP = true>
__device__ __forceinline__ double pow_prelog(double e, double lhi, double llo) {
  const double p = e * lhi;
  const double pe = fma(e, lhi, -p);       // exact low part of the product
  return exp_dd<CLAMP>(p, fma(e, llo, pe));
}

__device__ __forceinline__ double exp10_fast(double x) { return pow_prelog<true>(x, kLn10Hi, kLn10Lo); }

// (2^f - 1)/f for |f| <= 1/2 (+ a small low part): the coefficients of expm1_reduced times ln2^k,
// so that a power whose logarithm is stored in base 2 needs no argument scaling at all.
__device__ __forceinline__ double exp2m1_over_f(double f) {
  double p = 0x1.e9bd283d2763fp-32;
  p = fma(p, f, 0x1.e5eb790ac4b32p-28);
  p = fma(p, f, 0x1.b52507e76b54ep-24);
  p = fma(p, f, 0x1.62bfe3b6ff3bep-20);
  p = fma(p, f, 0x1.ffcbfc621dea4p-17);
  p = fma(p, f, 0x1.43091303a5f47p-13);
  p = fma(p, f, 0x1.5d87fe78a5d91p-10);
  p = fma(p, f, 0x1.3b2ab6fba37e5p-7);
  p = fma(p, f, 0x1.c6b08d704a0c0p-5);
  p = fma(p, f, 0x1.ebfbdff82c591p-3);
  return fma(p, f, 0x1.62e42fefa39efp-1);
}

// base^e with log2(base) = l2hi + l2lo known to double-double precision, |e log2(base)| < 1000
// (the caller bounds it: SedConsts::fast_ok).  Three instructions fewer than pow_prelog.
__device__ __forceinline__ double pow2_prelog(double e, double l2hi, double l2lo) {
  const double p = e * l2hi;
  const double pe = fma(e, l2hi, -p);      // exact low part of the product
  const double fn = rint(p);
  const double f = (p - fn) + fma(e, l2lo, pe);
  return ldexp(fma(exp2m1_over_f(f), f, 1.0), static_cast<int>(fn));
}

// expm1(x), any finite x (clamped to [-40, kExpMax]: below -40 the result is -1 to the last bit):
//   2^n (1+q) - 1 = 2^n * (q + (1 - 2^-n)),  q = expm1(r); the inner sum is exact-ish (one
//   rounding), so there is no cancellation for any n.
template <bool CLAMP = true>
__device__ __forceinline__ double expm1_fast(double x) {
  if (CLAMP) x = fmin(fmax(x, -40.0), kExpMax);
  int n;
  const double r = reduce_ln2(x, 0.0, n);
  const double q = expm1_reduced(r);
  const double t = 1.0 - ldexp(1.0, -n);
  return ldexp(q + t, n);
}

// 2^y - 1 for 0 <= y < 1000 (the Planck argument h nu / kT in units of ln 2; the caller bounds
// it: SedConsts::fast_ok): same composition as expm1_fast, no argument scaling.
__device__ __forceinline__ double exp2m1_fast(double y) {
  const double fn = rint(y);
  const double f = y - fn;                  // exact
  const int n = static_cast<int>(fn);
  const double q = exp2m1_over_f(f) * f;
  const double t = 1.0 - ldexp(1.0, -n);
  return ldexp(q + t, n);
}

// 2^(y + extra), |y| < 1000 (bounded by the caller), |extra| << 1
__device__ __forceinline__ double exp2_fast(double y, double extra) {
  const double fn = rint(y);
  const double f = (y - fn) + extra;
  return ldexp(fma(exp2m1_over_f(f), f, 1.0), static_cast<int>(fn));
}

// 2^y, |y| < 1000 (bounded by the caller).  Not exp2_fast(y, 0.0): without fast-math the
// compiler must keep the addition of a zero low part (it turns -0 into +0).
__device__ __forceinline__ double exp2_fast(double y) {
  const double fn = rint(y);
  const double f = y - fn;                  // exact
  return ldexp(fma(exp2m1_over_f(f), f, 1.0), static_cast<int>(fn));
}

// p z + c as ONE v_fma_f64 with the polynomial constant c in a register pair.  Written out because the
// compiler's own lowering of a Horner step whose constant it keeps in VGPRs across the MH loop is
// v_mov_b64 (a copy of the constant) + v_fmac_f64: two issue slots where one does (20 such copies per
// Box-Muller pair, one per band in the band loop).  Same operation, same bits.
__device__ __forceinline__ double fma_k(double p, double z, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "v"(c));
  return r;
}

// ... and with the constant in a scalar register pair: for kernels that have no vector registers to
// spare for polynomial constants (cl_mhw_kernels.hpp)
__device__ __forceinline__ double fma_ks(double p, double z, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
  return r;
}

// ---- table-driven 2^y (the band loop of the SED fast path) ------------------------------------
// y = k/2048 + f with k = round(2048 y) taken from the low mantissa bits of y + 1.5 2^41 (no
// v_rndne / v_cvt), |f| <= 2^-12:  2^y = 2^(k >> 11) * T[k & 2047] * (1 + f P(f)),  T[i] = 2^(i/2048)
// correctly rounded (16 KB in LDS, exp2_table.inc from tools/gen_exp2_table.py), P the degree-2
// Taylor polynomial of (2^f - 1)/f (remainder (f ln 2)^4/4! < 4e-17), and the power of two added
// to the exponent field (results stay normal: callers bound |y| < 1000).  11 instructions against
// the 16 of exp2_fast's degree-11 chain (13 with a 64-entry table and a degree-4 P) - on this
// kernel every VALU instruction costs one issue slot whatever its type
// (profiles/r1_bench_summary.txt), so the count is what matters; error <= 1.3 ulp
// (tests/test_fastmath_gpu.py).
constexpr int kExp2Bits = 11;
constexpr int kExp2Entries = 1 << kExp2Bits;
static __device__ const double kExp2Table[kExp2Entries] = {
#include "exp2_table.inc"
};

// The 64-row table of the Box-Muller kernels below (log_bm, sincos_bm; tools/gen_bm_table.py):
// row k = (1/c_k, ln c_k, sin(2 pi k/64), cos(2 pi k/64)).
constexpr int kBmEntries = 64;
static __device__ const double kBmTable[4 * kBmEntries] = {
#include "bm_table.inc"
};
// Static LDS of every kernel that stages the tables (18 KB): host code sizes dynamic LDS against it.
constexpr size_t kStaticTabBytes = (kExp2Entries + 4 * kBmEntries) * sizeof(double);

// The block's copy of the tables in LDS: 2^(i/2048), then the Box-Muller rows (bm_of()).  Every
// thread of the block must call this before the block's next __syncthreads(); the pointer is valid
// after that barrier.
__device__ __forceinline__ const double* stage_exp2_table() {
  __shared__ __align__(16) double tab[kExp2Entries + 4 * kBmEntries];
  for (int i = threadIdx.x; i < kExp2Entries; i += blockDim.x) tab[i] = kExp2Table[i];
  for (int i = threadIdx.x; i < 4 * kBmEntries; i += blockDim.x) tab[kExp2Entries + i] = kBmTable[i];
  return tab;
}
__device__ __forceinline__ const double* bm_of(const double* e2t) { return e2t + kExp2Entries; }

// N independent 2^y[i] at once (|y| < 1000, tab = stage_exp2_table()), written stage by stage
// (reductions and table reads of all N first, then the Horner steps of the N chains interleaved)
// so that the LDS latency and the FMA latency of one chain are covered by the others: with 2
// waves per SIMD nothing else hides them.
template <int N>
__device__ __forceinline__ void exp2_tab_n(const double* y, const double* tab, double* out) {
  constexpr double kMagic = 0x1.8p52 / kExp2Entries;      // ulp = 1/2048
  int k[N];
  double f[N], T[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double t = y[i] + kMagic;
    k[i] = __double2loint(t);                             // round(2048 y), two's complement
    f[i] = y[i] - (t - kMagic);                           // exact
    // byte offset of the entry: shift and mask, two instructions (the compiler's own lowering of
    // tab[k & 2047] for several k at once packs halves with v_perm / v_bitop3 and takes more)
    int off;
    asm("v_lshlrev_b32 %0, 3, %1\n\tv_and_b32 %0, 0x3ff8, %0" : "=v"(off) : "v"(k[i]));
    T[i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab) + off);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = fma(0x1.c6b08d704a0c0p-5, f[i], 0x1.ebfbdff82c58fp-3);
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = fma(p[i], f[i], 0x1.62e42fefa39efp-1);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double r = fma(T[i] * f[i], p[i], T[i]);        // T (1 + f P(f)), in [1, 2)
    // (the exponent added into the high word by two 32-bit operations instead of v_ldexp_f64 was measured
    // slower, 26.15 -> 27.1 ms per launch of C2: profiles/history_kernels.md)
    out[i] = ldexp(r, k[i] >> kExp2Bits);                 // v_ashrrev + v_ldexp_f64
  }
}

__device__ __forceinline__ double exp2_tab(double y, const double* tab) {
  double r;
  exp2_tab_n<1>(&y, tab, &r);
  return r;
}

// e^x of the accept test (MCMC.py:264, x = ln alpha < 0) as 2^(x log2 e) through the table: the
// rounding of the product puts |x| 1.1e-16 on top of exp2_tab's error (x = -40: 5e-15 relative, on a
// number that is only compared with a uniform draw); clamped so that the result stays a normal
// number; NaN compares as "reject" like the exp it replaces.
__device__ __forceinline__ double exp_accept(double x, const double* tab) {
  return exp2_tab(fmax(x * kLog2e, -1000.0), tab);
}

// sqrt(x) for Box-Muller's radius, x = -2 ln u in [0, 75]: v_rsq_f64 and one coupled Goldschmidt
// step with a residual correction (< 1 ulp; the compiler's sqrt adds range scaling, a second
// correction and the special-case selects).  x = 0 (u rounds to 1 once in 2^53 draws) is lifted to
// 2^-1000, whose root 2^-500 stands in for 0.
__device__ __forceinline__ double sqrt_fast(double x) {
  x = fmax(x, 0x1.0p-1000);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}

// a/b by Newton-Raphson on v_rcp_f64 plus one residual correction (Markstein): correctly rounded
// in all but ~1e-3 of cases (then 1 ulp off), for b and a/b in the normal range.  No scaling /
// fix-up for subnormal or overflowing quotients - callers keep operands away from those.
// 1/b to ~2^-50: v_rcp_f64 (about 2^-26) plus ONE Newton-Raphson step.  That is all the residual
// correction in div_rcp needs: its result errs by |q| * eps_r^2 before the final rounding.
__device__ __forceinline__ double rcp_nr(double b) {
  double r = __builtin_amdgcn_rcp(b);
  const double e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  return r;
}

// 1/b to the last bit or so (two Newton steps on v_rcp_f64), for b in the normal range: a third
// of the instructions of the IEEE division sequence
__device__ __forceinline__ double rcp_nr2(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  e = fma(-b, r, 1.0);
  return fma(r, e, r);
}

__device__ __forceinline__ double div_rcp(double a, double b, double rb) {   // rb ~ 1/b
  const double q = a * rb;
  const double rem = fma(-q, b, a);
  return fma(rem, rb, q);
}

__device__ __forceinline__ double div_fast(double a, double b) { return div_rcp(a, b, rcp_nr(b)); }

// a/b for the band terms of the SED fast path: the residual correction on the raw v_rcp_f64 (2^-26), no
// Newton step - the quotient errs by |q| 2^-52 before its final rounding (within ~1.5 ulp) where div_fast
// is correctly rounded almost always; two FMAs less per band: 26.15 -> 25.65 ms per 200-step launch of C2
// (tools/ab_bench.sh, one box).  Not for SEDLikelihood2, whose 2 x 2 solve multiplies the band fluxes'
// errors by its condition number.
__device__ __forceinline__ double div_band(double a, double b) {
  return div_rcp(a, b, __builtin_amdgcn_rcp(b));
}

// ---- Box-Muller kernels ------------------------------------------------------------------------
// ln(u) for normal positive u (the Box-Muller radius only needs a few ulp): u = m 2^e with
// m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), ln m = 2s + s z (2/3 + 2/5 z + ... ) with z = s^2.
// SK: the polynomial's constants in scalar instead of vector registers (same operations, same bits)
template <bool SK = false>
__device__ __forceinline__ double log_fast(double u) {
  int e = __builtin_amdgcn_frexp_exp(u);
  double m = __builtin_amdgcn_frexp_mant(u);               // [0.5, 1)
  const bool lo = m < 0x1.6a09e667f3bcdp-1;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = div_fast(f, 2.0 + f);
  const double z = s * s;
  auto hs = [](double p, double zz, double c) { return SK ? fma_ks(p, zz, c) : fma_k(p, zz, c); };
  double r = 2.0 / 23.0;
  r = hs(r, z, 2.0 / 21.0);
  r = hs(r, z, 2.0 / 19.0);
  r = hs(r, z, 2.0 / 17.0);
  r = hs(r, z, 2.0 / 15.0);
  r = hs(r, z, 2.0 / 13.0);
  r = hs(r, z, 2.0 / 11.0);
  r = hs(r, z, 2.0 / 9.0);
  r = hs(r, z, 2.0 / 7.0);
  r = hs(r, z, 2.0 / 5.0);
  r = hs(r, z, 2.0 / 3.0);
  const double lm = fma(s * z, r, s + s);
  const double fe = static_cast<double>(e);
  return fma(fe, kLn2Hi, fma(fe, kLn2Lo, lm));
}

// log2(u), u > 0 and normal: the same reduction, ln(m) scaled to base 2 and added to the exponent
__device__ __forceinline__ double log2_fast(double u) {
  int e = __builtin_amdgcn_frexp_exp(u);
  double m = __builtin_amdgcn_frexp_mant(u);               // [0.5, 1)
  const bool lo = m < 0x1.6a09e667f3bcdp-1;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  return fma(log_fast<>(m), kLog2e, static_cast<double>(e));  // |ln m| <= ln(2)/2: no cancellation
}

// sin(2 pi u), cos(2 pi u) for u in [0, 1): quarter-turn reduction (exact), then the fdlibm
// kernel polynomials on |y| <= pi/4 (error < 2^-57) and a quadrant rotation by selects.
__device__ __forceinline__ void sincos2pi_fast(double u, double* sn, double* cs) {
  const double a = 4.0 * u;
  const double fq = rint(a);
  const double f = a - fq;                                  // exact, |f| <= 1/2
  const int q = static_cast<int>(fq) & 3;
  const double y = fma(f, 0x1.1a62633145c07p-54, f * 0x1.921fb54442d18p+0);   // f * pi/2
  const double z = y * y;
  double rs = 1.58969099521155010221e-10;
  rs = fma_k(rs, z, -2.50507602534068634195e-08);
  rs = fma_k(rs, z, 2.75573137070700676789e-06);
  rs = fma_k(rs, z, -1.98412698298579493134e-04);
  rs = fma_k(rs, z, 8.33333333332248946124e-03);
  rs = fma_k(rs, z, -1.66666666666666324348e-01);
  const double s0 = fma(y * z, rs, y);
  double rc = -1.13596475577881948265e-11;
  rc = fma_k(rc, z, 2.08757232129817482790e-09);
  rc = fma_k(rc, z, -2.75573143513906633035e-07);
  rc = fma_k(rc, z, 2.48015872894767294178e-05);
  rc = fma_k(rc, z, -1.38888888888741095749e-03);
  rc = fma_k(rc, z, 4.16666666666666019037e-02);
  const double c0 = fma(z * z, rc, fma(-0.5, z, 1.0));
  // angle = y + q pi/2
  const double sa = (q & 1) ? c0 : s0;
  const double ca = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -sa : sa;
  *cs = ((q + 1) & 2) ? -ca : ca;
}

// ---- Box-Muller kernels, table-driven -----------------------------------------------------------
// ln(u), u in (0, 1) and normal (the 53-bit uniforms of philox.hpp): u = m 2^e, m in [1/2, 1); the top
// six mantissa bits of m pick c_k; r = m/c_k - 1 is one FMA (1/c_k has 9 bits: the product is rounded
// once, relative error 2^-53), |r| <= 0.0084; ln u = e ln 2 + ln c_k + log1p(r), log1p by its series
// through r^8.  Row 63 is c = 1: the logarithm of u -> 1 keeps its relative accuracy (the only place
// where the sum is small).  18 instructions against the 36 of log_fast (no division), <= 1 ulp
// (tests/test_fastmath_gpu.py).  bm = bm_of(stage_exp2_table()) or kBmTable itself.
__device__ __forceinline__ double log_bm(double u, const double* bm) {
  const int e = __builtin_amdgcn_frexp_exp(u);
  const double m = __builtin_amdgcn_frexp_mant(u);         // [0.5, 1)
  const unsigned off = (static_cast<unsigned>(__double2hiint(m)) >> 9) & 0x7e0u;   // 32-byte rows
  const double2 t = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(bm) + off);
  const double r = fma(m, t.x, -1.0);
  double p = -1.0 / 8.0;
  p = fma(p, r, 1.0 / 7.0);
  p = fma(p, r, -1.0 / 6.0);
  p = fma(p, r, 1.0 / 5.0);
  p = fma(p, r, -1.0 / 4.0);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -1.0 / 2.0);
  const double l = fma(r * r, p, r);
  const double fe = static_cast<double>(e);
  return fma(fe, kLn2Hi, fma(fe, kLn2Lo, t.y + l));
}

// sin(2 pi u), cos(2 pi u), u in [0, 1): u = k/64 + f/64 with k = round(64 u) read off the low mantissa
// bits of 64 u + 1.5 2^52 and |f| <= 1/2 exact; the rotation of (sin, cos)(2 pi k/64) from the table
// by b = f pi/32, |b| <= 0.05 (sine through b^7, cosine - 1 through b^8).  21 instructions against 46
// (no quadrant selects); absolute error <= 2e-16 - a value near zero is NOT relatively accurate, which
// the product with Box-Muller's radius does not need.
__device__ __forceinline__ void sincos_bm(double u, const double* bm, double* sn, double* cs) {
  const double a = 64.0 * u;
  const double t = a + 0x1.8p52;
  const double f = a - (t - 0x1.8p52);                     // exact
  const unsigned off = (static_cast<unsigned>(__double2loint(t)) << 5) & 0x7e0u;
  const double2 sc = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(bm) + off + 16);
  const double b = fma(f, 0x1.1a62633145c07p-58, f * 0x1.921fb54442d18p-4);   // f * pi/32
  const double z = b * b;
  double ps = fma(-1.0 / 5040.0, z, 1.0 / 120.0);
  ps = fma(ps, z, -1.0 / 6.0);
  const double sb = fma(b * z, ps, b);                     // sin b
  double pc = fma(1.0 / 40320.0, z, -1.0 / 720.0);
  pc = fma(pc, z, 1.0 / 24.0);
  const double cm1 = fma(z * z, pc, -0.5 * z);             // cos b - 1
  *sn = fma(sc.y, sb, fma(sc.x, cm1, sc.x));
  *cs = fma(-sc.x, sb, fma(sc.y, cm1, sc.y));
}

}  // namespace mi355
