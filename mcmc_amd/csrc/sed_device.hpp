// Device-side greybody models, Gaussian chi-square, priors and label canonicalisation.
// One lane evaluates one (source, parameter vector); the band table of the lane's source is
// read from LDS (broadcast reads when the lanes of a wave share a source).
//
// Reference arithmetic reproduced, evaluation order included (FP64 parity to 1e-12 depends on
// it): MCMC/submmSED/model.py:80-111 (blackbody/greybody), :130-769 (model classes),
// MCMC/GaussianData.py:31-64 (quadform/chi2), MCMC/submmSED/likelihood.py:35-155.
#pragma once
#include "common.hpp"
#include "fastmath.hpp"

namespace mi355 {

// band table of one source in LDS: [nbm][kBandFields] (common.hpp: BandField)
struct Bands {
  const double* base;
  int nbm;
  int nb;
  // Loop bound of the fast path: nb, or a wave-uniform upper bound of it (wave_max_uniform) so
  // that the band loop is scalar control flow.  Rows nb <= j < nbm of a table are padding with
  // weight 0 and benign values (nu = nu_b, d = 0): they add exactly 0 to a chi-square.
  int nbu;
  const double* e2t;   // stage_exp2_table() of the block: the 2^y kernel of the fast path
  // this source may take the clamp-free fast path (SedConsts::fast_ok and its own bands in range:
  // nb_fast() of the source's entry in the nb table)
  bool fast;
  __device__ __forceinline__ double f(int j, int field) const { return base[j * kBandFields + field]; }
  __device__ __forceinline__ double nu(int j) const { return f(j, kF_NU); }
  __device__ __forceinline__ double hnu(int j) const { return f(j, kF_HNU); }      // h/k * nu
  __device__ __forceinline__ double ratio(int j) const { return f(j, kF_RATIO); }  // nu / nu_b
  __device__ __forceinline__ double d(int j) const { return f(j, kF_D); }
  __device__ __forceinline__ double w(int j) const { return f(j, kF_W); }          // 1/sig^2
  // log2(base) = lhi + llo of the band's power-law base: nu/nu_b for the normalised models,
  // nu for submmModel1/2 (set per catalogue by mi355_sed_load)
  __device__ __forceinline__ double lhi(int j) const { return f(j, kF_LHI); }
  __device__ __forceinline__ double llo(int j) const { return f(j, kF_LLO); }
  // h nu / k in units of ln 2: the Planck argument of the fast path comes out in base 2
  __device__ __forceinline__ double hnu2(int j) const { return f(j, kF_HNU2); }
};

// The nb table on the device: the band count of the source, plus kNbSlow when one of its bands
// leaves the domain of the clamp-free kernels (mi355_sed_load decides, source by source).
constexpr int32_t kNbSlow = 1 << 30;
__device__ __forceinline__ int nb_count(int32_t raw) { return raw & (kNbSlow - 1); }
__device__ __forceinline__ bool nb_fast(int32_t raw, const SedConsts& k) {
  return k.fast_ok != 0.0 && (raw & kNbSlow) == 0;
}
__device__ __forceinline__ Bands make_bands(const double* base, int nbm, int32_t raw,
                                            const double* e2t, const SedConsts& k) {
  return Bands{base, nbm, nb_count(raw), nb_count(raw), e2t, nb_fast(raw, k)};
}

// max of v over the 64 lanes of the wave, as a scalar.  Every lane of the wave must be active.
__device__ __forceinline__ int wave_max_uniform(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

template <int MODEL>
struct ModelTraits;
template <> struct ModelTraits<MI355_SED_M1N> { static constexpr int P = 3; };
template <> struct ModelTraits<MI355_SED_M1N_LOGA> { static constexpr int P = 3; };
template <> struct ModelTraits<MI355_SED_M2N> { static constexpr int P = 6; };
template <> struct ModelTraits<MI355_SED_M2N_LOGA> { static constexpr int P = 6; };
template <> struct ModelTraits<MI355_SED_M1> { static constexpr int P = 2; };
template <> struct ModelTraits<MI355_SED_M2> { static constexpr int P = 4; };
template <> struct ModelTraits<MI355_SED_THICK> { static constexpr int P = 4; };
template <> struct ModelTraits<MI355_SED_THICK_LOGA> { static constexpr int P = 4; };

// ---- priors: model.py prior() classmethods -------------------------------------------------
__device__ __forceinline__ bool t_ok(double T, const SedConsts& k, bool upper) {
  return !(T < k.Tmin || (upper && T > k.Tmax));
}
__device__ __forceinline__ bool b_ok(double b, const SedConsts& k) {
  return !(b < k.bmin || b > k.bmax);
}

template <int MODEL>
__device__ __forceinline__ double sed_prior(const double* th, const SedConsts& k) {
  if constexpr (MODEL == MI355_SED_M1N) {               // model.py:560-570
    return (t_ok(th[2], k, true) && !(th[0] < 0) && b_ok(th[1], k)) ? 1.0 : 0.0;
  } else if constexpr (MODEL == MI355_SED_M1N_LOGA) {   // model.py:759-769
    return (t_ok(th[2], k, true) && b_ok(th[1], k)) ? 1.0 : 0.0;
  } else if constexpr (MODEL == MI355_SED_M2N) {        // model.py:407-425
    return (t_ok(th[2], k, true) && t_ok(th[5], k, true) && !(th[0] < 0) && !(th[3] < 0) &&
            b_ok(th[1], k) && b_ok(th[4], k)) ? 1.0 : 0.0;
  } else if constexpr (MODEL == MI355_SED_M2N_LOGA) {   // model.py:491-508
    return (t_ok(th[2], k, true) && t_ok(th[5], k, true) && b_ok(th[1], k) && b_ok(th[4], k))
               ? 1.0 : 0.0;
  } else if constexpr (MODEL == MI355_SED_M1) {         // model.py:244-255
    return (t_ok(th[1], k, false) && b_ok(th[0], k)) ? 1.0 : 0.0;
  } else if constexpr (MODEL == MI355_SED_M2) {         // model.py:160-174
    return (t_ok(th[1], k, false) && t_ok(th[3], k, false) && b_ok(th[0], k) && b_ok(th[2], k))
               ? 1.0 : 0.0;
  } else {                                              // model.py:674-693, :730-746
    if (MODEL == MI355_SED_THICK && th[0] < 0) return 0.0;
    if (!t_ok(th[2], k, true) || !b_ok(th[1], k) || th[3] < 0) return 0.0;
    return exp(-3.0 * th[3]);
  }
}

// The four normalised greybody models have a pure box prior with the same box for both components
// (value 0 or 1; model.py:407-425, :491-508, :560-570, :759-769): a point that differs from a point
// inside the box in ONE parameter - and by package()'s exchange of whole components - is inside the
// box iff that parameter is inside its own interval.  sed_prior_one tests parameter i = v with the
// comparisons sed_prior makes (NaN passes, as there).
template <int MODEL>
constexpr bool sed_prior_is_box() {
  return MODEL == MI355_SED_M1N || MODEL == MI355_SED_M1N_LOGA || MODEL == MI355_SED_M2N ||
         MODEL == MI355_SED_M2N_LOGA;
}
template <int MODEL>
__device__ __forceinline__ bool sed_prior_one(int i, double v, const SedConsts& k) {
  const int kind = i % 3;                                  // (A or logA, beta, T) per component
  if (kind == 2) return t_ok(v, k, true);
  if (kind == 1) return b_ok(v, k);
  return (MODEL == MI355_SED_M1N_LOGA || MODEL == MI355_SED_M2N_LOGA) ? true : !(v < 0);
}

// ---- package(): T1<=T2 canonicalisation, model.py:192-196, :439-443, :510-514 --------------
template <int MODEL>
__device__ __forceinline__ void sed_package(double* th) {
  if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) {
    if (th[2] > th[5]) {
      double t;
      t = th[0]; th[0] = th[3]; th[3] = t;
      t = th[1]; th[1] = th[4]; th[4] = t;
      t = th[2]; th[2] = th[5]; th[5] = t;
    }
  } else if constexpr (MODEL == MI355_SED_M2) {
    if (th[1] > th[3]) {
      double t;
      t = th[0]; th[0] = th[2]; th[2] = t;
      t = th[1]; th[1] = th[3]; th[3] = t;
    }
  }
}

template <int MODEL>
__device__ double sed_lnlike(const double* th, const Bands& B, const SedConsts& k, double* derived);

// All parameters finite (NaN/inf proposals can only be rejected: MCMC.py:238-243).
template <int P>
__device__ __forceinline__ bool all_finite(const double* th) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) s += fabs(th[i]);
  return s < __builtin_inf();
}

// package() as an involution: the exchange itself, and package() reporting whether it exchanged
// (k_mh_run's in-place step undoes it on a rejected lane).
template <int MODEL>
__device__ __forceinline__ void sed_package_swap(double* th) {
  if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) {
    double t;
    t = th[0]; th[0] = th[3]; th[3] = t;
    t = th[1]; th[1] = th[4]; th[4] = t;
    t = th[2]; th[2] = th[5]; th[5] = t;
  } else if constexpr (MODEL == MI355_SED_M2) {
    double t;
    t = th[0]; th[0] = th[2]; th[2] = t;
    t = th[1]; th[1] = th[3]; th[3] = t;
  }
}
template <int MODEL>
__device__ __forceinline__ bool sed_package_flag(double* th) {
  bool sw = false;
  if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) sw = th[2] > th[5];
  else if constexpr (MODEL == MI355_SED_M2) sw = th[1] > th[3];
  if (__builtin_amdgcn_ballot_w64(sw) != 0) {             // rare: a wave-level branch instead of P selects
    asm volatile("");
    if (sw) sed_package_swap<MODEL>(th);
  }
  return sw;
}

// ---- likelihood, fast path -------------------------------------------------------------------
// like.lnLike(theta) WITHOUT the prior, for theta INSIDE the prior box and finite (the only
// points a Metropolis step can accept): T>=Tmin>0 so every Planck argument is positive, and the
// power-law bases are launch constants whose logarithms the host precomputed.  Same formulas as
// the exact path below, transcendental kernels from fastmath.hpp; differences vs the reference
// stay at the 1e-15 level (tests/test_sed_gpu.py).  The transcendental kernels run without their
// argument clamps: mi355_sed_load only enables this path (SedConsts::fast_ok) when every band of
// the catalogue keeps h nu/kT and (3+beta) ln(base) below 300 over the whole prior box.
template <int MODEL>
__device__ __forceinline__ double sed_lnlike_fast(const double* th, const Bands& B,
                                                  const SedConsts& k, double* derived) {
  if (derived) { derived[0] = __builtin_nan(""); derived[1] = derived[0]; }
  if constexpr (MODEL == MI355_SED_M1N || MODEL == MI355_SED_M1N_LOGA ||
                MODEL == MI355_SED_THICK_LOGA) {
    const double A = (MODEL == MI355_SED_M1N) ? th[0] : exp10_fast(th[0]);
    const double T = th[2], e3 = 3.0 + th[1];
    const double rT = rcp_nr2(T);                          // T >= Tmin > 0 on this path
    // same base-2 form of the band term as the two-component branch below
    const double ae = A * (exp2_tab(k.hk_nub2 * rT, B.e2t) - 1.0);   // y_b >= 0.5: fast_ok
    double chi2 = 0.0;
    for (int j = 0; j < B.nbu; ++j) {
      const double y = B.hnu2(j) * rT;
      const double z[2] = {fma(e3, B.lhi(j), -y), -y};
      double pw[2];
      exp2_tab_n<2>(z, B.e2t, pw);
      const double n = pw[0] * ae, dn = 1.0 - pw[1];
      const double r = B.d(j) - div_band(n, dn);
      chi2 = fma(r * B.w(j), r, chi2);
    }
    return -0.5 * chi2;
  } else if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) {
    const bool lg = (MODEL == MI355_SED_M2N_LOGA);
    const double A1 = lg ? exp10_fast(th[0]) : th[0];
    const double A2 = lg ? exp10_fast(th[3]) : th[3];
    const double T1 = th[2], T2 = th[5];
    const double e1 = 3.0 + th[1], e2 = 3.0 + th[4];
    const double rT1 = rcp_nr2(T1), rT2 = rcp_nr2(T2);     // T >= Tmin > 0 on this path
    // Per band and component, with y = h nu / (k T ln 2) and a = (3+beta) log2(nu/nu_b):
    //   A r^(3+beta) expm1(x_b) / expm1(x) = [A expm1(x_b)] 2^(a-y) / (1 - 2^-y),
    // two plain 2^z kernels and a subtraction instead of a power and an expm1 (1 - 2^-y loses
    // log2(1/x) bits, x >= 0.02 is part of fast_ok); the exponent a - y is one FMA on the high
    // part of log2(nu/nu_b) (its rounding, |a - y| 2^-53 in the result, is the error that
    // matters; tests/parity_margin.py: lnL within 2.5e-14 of the oracle); the two components
    // share one division, (n1 d2 + n2 d1) / (d1 d2) with d in (0, 1].
    // 2^(y_b) - 1 of the normalisation through the table-driven kernel: y_b >= 0.5 over the prior
    // box is part of fast_ok, so the subtraction costs at most one bit
    const double yb[2] = {k.hk_nub2 * rT1, k.hk_nub2 * rT2};
    double eb[2];
    exp2_tab_n<2>(yb, B.e2t, eb);
    const double ae1 = A1 * (eb[0] - 1.0);
    const double ae2 = A2 * (eb[1] - 1.0);
    double chi2 = 0.0;
    int j = 0;
    // two bands at a time (their eight exponentials interleaved), terms added in band order
    for (; j + 1 < B.nbu; j += 2) {
      const double hna = B.hnu2(j), lha = B.lhi(j), hnb = B.hnu2(j + 1), lhb = B.lhi(j + 1);
      const double y1a = hna * rT1, y2a = hna * rT2, y1b = hnb * rT1, y2b = hnb * rT2;
      const double z[8] = {fma(e1, lha, -y1a), -y1a, fma(e2, lha, -y2a), -y2a,
                           fma(e1, lhb, -y1b), -y1b, fma(e2, lhb, -y2b), -y2b};
      double pw[8];
      exp2_tab_n<8>(z, B.e2t, pw);
      const double n1a = pw[0] * ae1, d1a = 1.0 - pw[1], n2a = pw[2] * ae2, d2a = 1.0 - pw[3];
      const double n1b = pw[4] * ae1, d1b = 1.0 - pw[5], n2b = pw[6] * ae2, d2b = 1.0 - pw[7];
      const double ra = B.d(j) - div_band(fma(n1a, d2a, n2a * d1a), d1a * d2a);
      const double rb = B.d(j + 1) - div_band(fma(n1b, d2b, n2b * d1b), d1b * d2b);
      chi2 = fma(ra * B.w(j), ra, chi2);
      chi2 = fma(rb * B.w(j + 1), rb, chi2);
    }
    for (; j < B.nbu; ++j) {
      const double hn = B.hnu2(j), lh = B.lhi(j);
      const double y1 = hn * rT1, y2 = hn * rT2;
      const double z[4] = {fma(e1, lh, -y1), -y1, fma(e2, lh, -y2), -y2};
      double pw[4];
      exp2_tab_n<4>(z, B.e2t, pw);
      const double n1 = pw[0] * ae1, d1 = 1.0 - pw[1];
      const double n2 = pw[2] * ae2, d2 = 1.0 - pw[3];
      const double r = B.d(j) - div_band(fma(n1, d2, n2 * d1), d1 * d2);
      chi2 = fma(r * B.w(j), r, chi2);
    }
    return -0.5 * chi2;
  } else if constexpr (MODEL == MI355_SED_M1) {
    const double T = th[1], e3 = 3.0 + th[0];
    // This model's prior has no upper temperature limit (model.py:244-255) while fast_ok bounds
    // h nu/kT from below with Tmax: beyond it 1 - 2^-y loses log2(T/Tmax) more bits, so those
    // (rare, Rayleigh-Jeans) points take the exact path.
    if (T > k.Tmax) return sed_lnlike<MODEL>(th, B, k, derived);
    const double pre = k.prefac * pow2_prelog(-th[0], k.ln_nub_hi, k.ln_nub_lo);
    const double rT = 1.0 / T;
    double FNiF = 0.0, FNid = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double y = B.hnu2(j) * rT;                      // base is nu itself for this model
      const double g = div_band(pre * exp2_fast(fma(e3, B.lhi(j), -y)), 1.0 - exp2_fast(-y));
      const double gw = g * B.w(j);
      FNiF = fma(gw, g, FNiF);
      FNid = fma(gw, B.d(j), FNid);
    }
    const double amp = FNid / FNiF;
    if (derived) derived[0] = amp;
    return 0.5 * FNid * amp;
  } else if constexpr (MODEL == MI355_SED_M2) {
    // SEDLikelihood2 (likelihood.py:35-74): the two un-normalised greybodies at every band through the
    // table-driven 2^z kernels - prefac nu_b^-beta nu^(3+beta) / expm1(x) = pre 2^(e log2 nu - y) / (1 - 2^-y),
    // as the one-component branch above - then the 2 x 2 normal equations and their Cholesky solve
    // exactly as the exact path does them (same operation order: the system's condition number, up to
    // 1e6 and beyond on real photometry, multiplies whatever the band fluxes err by, so the solve
    // itself adds nothing).  No upper temperature limit in this model's prior either (model.py:160-174):
    // points beyond Tmax take the exact path.
    const double T1 = th[1], T2 = th[3];
    if (T1 > k.Tmax || T2 > k.Tmax) return sed_lnlike<MODEL>(th, B, k, derived);
    const double e1 = 3.0 + th[0], e2 = 3.0 + th[2];
    const double pre1 = k.prefac * pow2_prelog(-th[0], k.ln_nub_hi, k.ln_nub_lo);
    const double pre2 = k.prefac * pow2_prelog(-th[2], k.ln_nub_hi, k.ln_nub_lo);
    const double rT1 = rcp_nr2(T1), rT2 = rcp_nr2(T2);     // T >= Tmin > 0 on this path
    double m00 = 0, m01 = 0, m11 = 0, b0 = 0, b1 = 0;
    auto add = [&](double g1, double g2, double w, double d) {
      const double g1w = g1 * w, g2w = g2 * w;
      m00 = fma(g1w, g1, m00); m01 = fma(g1w, g2, m01); m11 = fma(g2w, g2, m11);
      b0 = fma(g1w, d, b0);    b1 = fma(g2w, d, b1);
    };
    int j = 0;
    // two bands at a time (their eight exponentials interleaved); rows nb <= j < nbu have weight 0
    // and finite fluxes: they add exact zeros
    for (; j + 1 < B.nbu; j += 2) {
      const double hna = B.hnu2(j), lha = B.lhi(j), hnb = B.hnu2(j + 1), lhb = B.lhi(j + 1);
      const double y1a = hna * rT1, y2a = hna * rT2, y1b = hnb * rT1, y2b = hnb * rT2;
      const double z[8] = {fma(e1, lha, -y1a), -y1a, fma(e2, lha, -y2a), -y2a,
                           fma(e1, lhb, -y1b), -y1b, fma(e2, lhb, -y2b), -y2b};
      double pw[8];
      exp2_tab_n<8>(z, B.e2t, pw);
      const double g1a = div_fast(pre1 * pw[0], 1.0 - pw[1]), g2a = div_fast(pre2 * pw[2], 1.0 - pw[3]);
      const double g1b = div_fast(pre1 * pw[4], 1.0 - pw[5]), g2b = div_fast(pre2 * pw[6], 1.0 - pw[7]);
      add(g1a, g2a, B.w(j), B.d(j));
      add(g1b, g2b, B.w(j + 1), B.d(j + 1));
    }
    for (; j < B.nbu; ++j) {
      const double hn = B.hnu2(j), lh = B.lhi(j);
      const double y1 = hn * rT1, y2 = hn * rT2;
      const double z[4] = {fma(e1, lh, -y1), -y1, fma(e2, lh, -y2), -y2};
      double pw[4];
      exp2_tab_n<4>(z, B.e2t, pw);
      add(div_fast(pre1 * pw[0], 1.0 - pw[1]), div_fast(pre2 * pw[2], 1.0 - pw[3]), B.w(j), B.d(j));
    }
    double z0 = __builtin_nan(""), z1 = z0;                // not positive definite: rejected, as the exact path
    if (m00 > 0.0) {
      const double r00 = 1.0 / sqrt(m00);
      const double u01 = m01 * r00;
      const double d11 = m11 - u01 * u01;
      if (d11 > 0.0) {
        const double r11 = 1.0 / sqrt(d11);
        const double y0 = b0 * r00;
        const double y1 = fma(-u01, y0, b1) * r11;
        z1 = y1 * r11;
        z0 = fma(-u01, z1, y0) * r00;
      }
    }
    if (derived) { derived[0] = z0; derived[1] = z1; }
    return 0.5 * (b0 * z0 + b1 * z1);
  } else if constexpr (MODEL == MI355_SED_THICK) {
    // A (1 - e^-tau)/(1 - e^-tau_b) (nu/nu_b)^3 expm1(x_b)/expm1(x), tau = (nu/nu0/nu_rescale)^beta
    // (model.py:649-657).  The optical depth in base 2: log2 tau = beta (log2(nu/nu_b) - c0) with
    // c0 = log2(nu0 nu_rescale/nu_b) - one logarithm per evaluation, log2(nu/nu_b) is the band
    // table's - then 1 - e^-tau through expm1 (no cancellation at small tau; 1 to the last bit
    // beyond tau = 50); the Planck factor as in the one-component branch above with exponent 3.
    // nu0 outside [1e-60, 1e60] (the prior only asks nu0 >= 0) would overflow the 2^z kernels:
    // those points take the exact path.
    const double A = th[0], b = th[1], T = th[2], nu0 = th[3];
    if (!(nu0 > 1.0e-60 && nu0 < 1.0e60)) return sed_lnlike<MODEL>(th, B, k, derived);
    const double c0 = log2_fast(nu0) + k.thick_c;
    const double rT = rcp_nr2(T);
    const double y0[2] = {k.hk_nub2 * rT, -b * c0};
    double e0[2];
    exp2_tab_n<2>(y0, B.e2t, e0);
    const double taub = e0[1];
    const double den = (taub > 50.0) ? 1.0 : -expm1_fast<false>(-taub);
    const double cf = div_fast(A * (e0[0] - 1.0), den);    // A expm1(x_b) / (1 - e^-tau_b)
    double chi2 = 0.0;
    for (int j = 0; j < B.nbu; ++j) {
      const double y = B.hnu2(j) * rT, lh = B.lhi(j);
      const double z[3] = {fma(3.0, lh, -y), -y, b * ((lh - c0) + B.llo(j))};
      double pw[3];
      exp2_tab_n<3>(z, B.e2t, pw);
      const double t1 = (pw[2] > 50.0) ? 1.0 : -expm1_fast<false>(-pw[2]);
      const double r = B.d(j) - div_band((t1 * pw[0]) * cf, 1.0 - pw[1]);
      chi2 = fma(r * B.w(j), r, chi2);
    }
    return -0.5 * chi2;
  } else {
    return sed_lnlike<MODEL>(th, B, k, derived);
  }
}

// ---- likelihood, fast path, lanes = (chain, band) ---------------------------------------------
// The same function as sed_lnlike_fast for a chain that owns G consecutive lanes (G = 8 or 32, a
// power of two >= the source's bands): lane jb of the group evaluates band jb, then every lane
// gathers the band terms and adds them in band order - the very sequence of fused multiply-adds the
// one-lane loop runs, so the value is identical bit for bit.  North_star's primary mapping (SURVEY
// App. C): it shortens a step from nb band evaluations to one, and pays when the chains of the job
// do not fill the machine (tools/ubench_lanemap.hip: 2.4x at nb = 27, 1.6x at nb = 8 up to ~4 000
// chains; a third of the lane = chain rate once every SIMD has a wave of its own).  Every lane of
// the group must call it with the same th (they hold copies of the chain); nbu is the wave-uniform
// bound of sed_lnlike_fast (padding rows add exact zeros there and here).
// The gather goes through LDS: the lane parks its pair in its own 16-byte slot, then reads the
// group's slots in band order (one 128-bit broadcast read per band; four cross-lane permutes per
// band otherwise).  The group is part of one wave, whose LDS operations execute in order: no
// barrier, only a fence that keeps the compiler from moving the reads above the write.
__device__ __forceinline__ void cb_publish(double2* slot, double a, double b) {
  *slot = make_double2(a, b);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double cb_chi2_in_order(double rw, double r, int nbu, double2* slot, int jb) {
  cb_publish(slot, rw, r);
  const double2* grp = slot - jb;
  double chi2 = 0.0;
  for (int j = 0; j < nbu; ++j) {
    const double2 v = grp[j];
    chi2 = fma(v.x, v.y, chi2);
  }
  return chi2;
}

template <int MODEL, int G>
__device__ __forceinline__ double sed_lnlike_fast_cb(const double* th, const Bands& B,
                                                     const SedConsts& k, int jb, double2* slot) {
  const bool mine = jb < B.nbm;                            // rows nb <= jb < nbm: padding, weight 0
  if constexpr (MODEL == MI355_SED_M1N || MODEL == MI355_SED_M1N_LOGA ||
                MODEL == MI355_SED_THICK_LOGA) {
    const double A = (MODEL == MI355_SED_M1N) ? th[0] : exp10_fast(th[0]);
    const double T = th[2], e3 = 3.0 + th[1];
    const double rT = rcp_nr2(T);
    const double ae = A * (exp2_tab(k.hk_nub2 * rT, B.e2t) - 1.0);
    double r = 0.0, rw = 0.0;
    if (mine) {
      const double y = B.hnu2(jb) * rT;
      const double z[2] = {fma(e3, B.lhi(jb), -y), -y};
      double pw[2];
      exp2_tab_n<2>(z, B.e2t, pw);
      const double n = pw[0] * ae, dn = 1.0 - pw[1];
      r = B.d(jb) - div_band(n, dn);
      rw = r * B.w(jb);
    }
    return -0.5 * cb_chi2_in_order(rw, r, B.nbu, slot, jb);
  } else if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) {
    const bool lg = (MODEL == MI355_SED_M2N_LOGA);
    const double A1 = lg ? exp10_fast(th[0]) : th[0];
    const double A2 = lg ? exp10_fast(th[3]) : th[3];
    const double T1 = th[2], T2 = th[5];
    const double e1 = 3.0 + th[1], e2 = 3.0 + th[4];
    const double rT1 = rcp_nr2(T1), rT2 = rcp_nr2(T2);
    const double yb[2] = {k.hk_nub2 * rT1, k.hk_nub2 * rT2};
    double eb[2];
    exp2_tab_n<2>(yb, B.e2t, eb);
    const double ae1 = A1 * (eb[0] - 1.0);
    const double ae2 = A2 * (eb[1] - 1.0);
    double r = 0.0, rw = 0.0;
    if (mine) {
      const double hn = B.hnu2(jb), lh = B.lhi(jb);
      const double y1 = hn * rT1, y2 = hn * rT2;
      const double z[4] = {fma(e1, lh, -y1), -y1, fma(e2, lh, -y2), -y2};
      double pw[4];
      exp2_tab_n<4>(z, B.e2t, pw);
      const double n1 = pw[0] * ae1, d1 = 1.0 - pw[1];
      const double n2 = pw[2] * ae2, d2 = 1.0 - pw[3];
      r = B.d(jb) - div_band(fma(n1, d2, n2 * d1), d1 * d2);
      rw = r * B.w(jb);
    }
    return -0.5 * cb_chi2_in_order(rw, r, B.nbu, slot, jb);
  } else if constexpr (MODEL == MI355_SED_M1) {
    const double T = th[1], e3 = 3.0 + th[0];
    if (T > k.Tmax) return sed_lnlike<MODEL>(th, B, k, nullptr);   // as sed_lnlike_fast: every lane, in full
    const double pre = k.prefac * pow2_prelog(-th[0], k.ln_nub_hi, k.ln_nub_lo);
    const double rT = 1.0 / T;
    double g = 0.0, gw = 0.0;
    if (jb < B.nb) {
      const double y = B.hnu2(jb) * rT;
      g = div_band(pre * exp2_fast(fma(e3, B.lhi(jb), -y)), 1.0 - exp2_fast(-y));
      gw = g * B.w(jb);
    }
    cb_publish(slot, gw, g);
    const double2* grp = slot - jb;
    double FNiF = 0.0, FNid = 0.0;
    for (int j = 0; j < B.nb; ++j) {                       // the group's own count, as the one-lane loop
      const double2 v = grp[j];
      FNiF = fma(v.x, v.y, FNiF);
      FNid = fma(v.x, B.d(j), FNid);
    }
    const double amp = FNid / FNiF;
    return 0.5 * FNid * amp;
  } else if constexpr (MODEL == MI355_SED_THICK) {
    const double A = th[0], b = th[1], T = th[2], nu0 = th[3];
    if (!(nu0 > 1.0e-60 && nu0 < 1.0e60)) return sed_lnlike<MODEL>(th, B, k, nullptr);
    const double c0 = log2_fast(nu0) + k.thick_c;
    const double rT = rcp_nr2(T);
    const double y0[2] = {k.hk_nub2 * rT, -b * c0};
    double e0[2];
    exp2_tab_n<2>(y0, B.e2t, e0);
    const double taub = e0[1];
    const double den = (taub > 50.0) ? 1.0 : -expm1_fast<false>(-taub);
    const double cf = div_fast(A * (e0[0] - 1.0), den);
    double r = 0.0, rw = 0.0;
    if (mine) {
      const double y = B.hnu2(jb) * rT, lh = B.lhi(jb);
      const double z[3] = {fma(3.0, lh, -y), -y, b * ((lh - c0) + B.llo(jb))};
      double pw[3];
      exp2_tab_n<3>(z, B.e2t, pw);
      const double t1 = (pw[2] > 50.0) ? 1.0 : -expm1_fast<false>(-pw[2]);
      r = B.d(jb) - div_band((t1 * pw[0]) * cf, 1.0 - pw[1]);
      rw = r * B.w(jb);
    }
    return -0.5 * cb_chi2_in_order(rw, r, B.nbu, slot, jb);
  } else {
    return sed_lnlike<MODEL>(th, B, k, nullptr);           // SEDLikelihood2: no fast path (host never picks G > 1)
  }
}

// ---- pieces of the incremental sampler (k_mh_run_inc) ------------------------------------------
// The normalised greybody of ONE component at every band of the source, without its amplitude:
//   u[j] = (nu_j/nu_b)^(3+beta) expm1(h nu_b/kT) / expm1(h nu_j/kT)            model.py:107-109
// in the base-2 form of sed_lnlike_fast (same kernels, same domain: fast_ok).  Bands are handled
// in pairs under one wave-uniform branch (the lanes of a wave share the source), so that four
// exponent chains are in flight; u[j] = 0 for j >= nb.
constexpr int kIncBands = 8;            // bands per source the register cache is sized for

__device__ __forceinline__ void sed_unit_terms(double e3, double T, const Bands& B,
                                               const SedConsts& k, double* u) {
  const double rT = rcp_nr2(T);
  const double em = exp2_tab(k.hk_nub2 * rT, B.e2t) - 1.0;          // y_b >= 0.5: fast_ok
#pragma unroll
  for (int q = 0; q < kIncBands; q += 2) {
    if (q < B.nb) {
      const int j1 = (q + 1 < B.nb) ? q + 1 : q;           // odd nb: the last band twice
      const double ya = B.hnu2(q) * rT, yb = B.hnu2(j1) * rT;
      const double z[4] = {fma(e3, B.lhi(q), -ya), -ya, fma(e3, B.lhi(j1), -yb), -yb};
      double pw[4];
      exp2_tab_n<4>(z, B.e2t, pw);
      const double na = pw[0] * em, da = 1.0 - pw[1];
      const double nb_ = pw[2] * em, db = 1.0 - pw[3];
      u[q] = div_band(na, da);
      const double ub = div_band(nb_, db);
      u[q + 1] = (q + 1 < B.nb) ? ub : 0.0;
    } else {
      u[q] = 0.0;
      u[q + 1] = 0.0;
    }
  }
}

// chi2 of the two-component model A_x u_x + A_o u_o against the source's photometry
// (GaussianData.py:47,64); terms beyond nb have u = 0 and get weight 0.  ux / uo are read with the
// given element strides (1: a register array; blockDim.x: a lane's column of the LDS cache, which
// keeps the cached terms out of the register budget).
__device__ __forceinline__ double sed_chi2_two(const Bands& B, double Ax, const double* ux, int sx,
                                               double Ao, const double* uo, int so) {
  double chi2 = 0.0;
#pragma unroll
  for (int q = 0; q < kIncBands; q += 2) {
    if (q < B.nb) {
      const bool two = q + 1 < B.nb;
      const int j1 = two ? q + 1 : q;
      const double ra = B.d(q) - fma(Ax, ux[q * sx], Ao * uo[q * so]);
      const double rb = B.d(j1) - fma(Ax, ux[(q + 1) * sx], Ao * uo[(q + 1) * so]);
      chi2 = fma(ra * B.w(q), ra, chi2);
      chi2 = fma(rb * (two ? B.w(j1) : 0.0), rb, chi2);
    }
  }
  return chi2;
}

// ---- likelihood, exact path --------------------------------------------------------------------
// Returns like.lnLike(theta) WITHOUT the prior for ANY theta, with the reference's IEEE edge
// behaviour (overflow -> inf -> 0 flux, NaN propagation).  derived[0..1] receive the ML amplitudes
// of the marginalised likelihoods (NaN otherwise) when derived != nullptr.
template <int MODEL>
__device__ double sed_lnlike(const double* th, const Bands& B, const SedConsts& k,
                             double* derived) {
  if (derived) { derived[0] = __builtin_nan(""); derived[1] = derived[0]; }

  if constexpr (MODEL == MI355_SED_M1N || MODEL == MI355_SED_M1N_LOGA ||
                MODEL == MI355_SED_THICK_LOGA) {
    // A * (nu/nu_b)^(3+beta) * expm1(x_b)/expm1(x)        model.py:107-109, :550-551
    const double A = (MODEL == MI355_SED_M1N) ? th[0] : pow(10.0, th[0]);   // :755
    const double beta = th[1], T = th[2];
    const double en = expm1(k.hk_nub / T);
    const double e3 = 3.0 + beta;
    double chi2 = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double x = B.hnu(j) / T;
      const double g = pow(B.ratio(j), e3) * en / expm1(x);
      const double r = B.d(j) - A * g;
      chi2 += (r * B.w(j)) * r;                            // GaussianData.py:47,64
    }
    return -0.5 * chi2;                                    // likelihood.py:155
  } else if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) {
    const bool lg = (MODEL == MI355_SED_M2N_LOGA);         // model.py:394-396, :484-489
    const double A1 = lg ? pow(10.0, th[0]) : th[0];
    const double A2 = lg ? pow(10.0, th[3]) : th[3];
    const double T1 = th[2], T2 = th[5];
    const double e1 = 3.0 + th[1], e2 = 3.0 + th[4];
    const double en1 = expm1(k.hk_nub / T1), en2 = expm1(k.hk_nub / T2);
    double chi2 = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double hn = B.hnu(j), rt = B.ratio(j);
      const double g1 = pow(rt, e1) * en1 / expm1(hn / T1);
      const double g2 = pow(rt, e2) * en2 / expm1(hn / T2);
      const double r = B.d(j) - (A1 * g1 + A2 * g2);
      chi2 += (r * B.w(j)) * r;
    }
    return -0.5 * chi2;
  } else if constexpr (MODEL == MI355_SED_M1) {
    // prefac * nu_b^-beta * nu^(3+beta) / expm1(x)         model.py:111, :238-239
    const double beta = th[0], T = th[1];
    const double pre = k.prefac * pow(k.nu_b, -beta);
    const double e3 = 3.0 + beta;
    double FNiF = 0.0, FNid = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double g = pre * pow(B.nu(j), e3) / expm1(B.hnu(j) / T);
      const double gw = g * B.w(j);
      FNiF += gw * g;
      FNid += gw * B.d(j);
    }
    const double amp = FNid / FNiF;                        // likelihood.py:101-110
    if (derived) derived[0] = amp;
    return 0.5 * FNid * amp;
  } else if constexpr (MODEL == MI355_SED_M2) {
    const double pre1 = k.prefac * pow(k.nu_b, -th[0]), pre2 = k.prefac * pow(k.nu_b, -th[2]);
    const double e1 = 3.0 + th[0], e2 = 3.0 + th[2];
    const double T1 = th[1], T2 = th[3];
    double m00 = 0, m01 = 0, m11 = 0, b0 = 0, b1 = 0;
    for (int j = 0; j < B.nb; ++j) {
      const double nu = B.nu(j), hn = B.hnu(j), w = B.w(j), d = B.d(j);
      const double g1 = pre1 * pow(nu, e1) / expm1(hn / T1);
      const double g2 = pre2 * pow(nu, e2) / expm1(hn / T2);
      const double g1w = g1 * w, g2w = g2 * w;
      m00 += g1w * g1; m01 += g1w * g2; m11 += g2w * g2;
      b0 += g1w * d;   b1 += g2w * d;
    }
    // (F^T N^-1 F) z = F^T N^-1 d as linalg.solve(..., sym_pos=True) does it (likelihood.py:63):
    // Cholesky U^T U of the upper triangle, then the two triangular solves - in LAPACK / OpenBLAS
    // operation order (reciprocal pivots, fused multiply-adds in the substitutions; oracle/refcpu.py
    // matches the reference's fixture bit for bit with it).  Not positive definite = LinAlgError
    // there = ZeroPosterior = the point is rejected: NaN here.
    double z0 = __builtin_nan(""), z1 = z0;
    if (m00 > 0.0) {
      const double r00 = 1.0 / sqrt(m00);
      const double u01 = m01 * r00;
      const double d11 = m11 - u01 * u01;
      if (d11 > 0.0) {
        const double r11 = 1.0 / sqrt(d11);
        const double y0 = b0 * r00;
        const double y1 = fma(-u01, y0, b1) * r11;
        z1 = y1 * r11;
        z0 = fma(-u01, z1, y0) * r00;
      }
    }
    if (derived) { derived[0] = z0; derived[1] = z1; }
    return 0.5 * (b0 * z0 + b1 * z1);
  } else {  // MI355_SED_THICK                              model.py:641-657
    const double A = th[0], b = th[1], T = th[2], nu0 = th[3];
    const double en = expm1(k.hk_nub / T);
    const double tau_b = pow(k.nu_b / nu0 / k.nu_rescale, b);
    const double den = 1.0 - exp(-tau_b);
    double chi2 = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double tau = pow(B.nu(j) / nu0 / k.nu_rescale, b);
      const double bb = pow(B.ratio(j), 3.0) * en / expm1(B.hnu(j) / T);   // model.py:92
      const double m = A * (1.0 - exp(-tau)) / den * bb;
      const double r = B.d(j) - m;
      chi2 += (r * B.w(j)) * r;
    }
    return -0.5 * chi2;
  }
}

}  // namespace mi355

namespace mi355 {
// model(*theta).at_nu(nu) for one frequency, exact (OCML) arithmetic in the reference's order.
// out[1] is only written for submmModel2 (two un-normalised components).
template <int MODEL>
__device__ void sed_flux(const double* th, double nu, const SedConsts& k, double* out) {
  auto grey_n = [&](double beta, double T) {             // model.py:107-109
    return pow(nu / k.nu_b, 3.0 + beta) * expm1(k.hk_nub / T) / expm1(k.h_over_k * nu / T);
  };
  auto grey_u = [&](double beta, double T) {             // model.py:111
    return k.prefac * pow(k.nu_b, -beta) * pow(nu, 3.0 + beta) / expm1(k.h_over_k * nu / T);
  };
  if constexpr (MODEL == MI355_SED_M1N) {
    out[0] = th[0] * grey_n(th[1], th[2]);
  } else if constexpr (MODEL == MI355_SED_M1N_LOGA || MODEL == MI355_SED_THICK_LOGA) {
    out[0] = pow(10.0, th[0]) * grey_n(th[1], th[2]);
  } else if constexpr (MODEL == MI355_SED_M2N) {
    out[0] = th[0] * grey_n(th[1], th[2]) + th[3] * grey_n(th[4], th[5]);
  } else if constexpr (MODEL == MI355_SED_M2N_LOGA) {
    out[0] = pow(10.0, th[0]) * grey_n(th[1], th[2]) + pow(10.0, th[3]) * grey_n(th[4], th[5]);
  } else if constexpr (MODEL == MI355_SED_M1) {
    out[0] = grey_u(th[0], th[1]);
  } else if constexpr (MODEL == MI355_SED_M2) {
    out[0] = grey_u(th[0], th[1]);
    out[1] = grey_u(th[2], th[3]);
  } else {                                               // model.py:649-652
    const double tau = pow(nu / th[3] / k.nu_rescale, th[1]);
    const double tau_b = pow(k.nu_b / th[3] / k.nu_rescale, th[1]);
    const double bb = pow(nu / k.nu_b, 3.0) * expm1(k.hk_nub / th[2]) / expm1(k.h_over_k * nu / th[2]);
    out[0] = th[0] * (1.0 - exp(-tau)) / (1.0 - exp(-tau_b)) * bb;
  }
}

// The exact path for a chain that owns G lanes (a source outside the fast domain in a job small enough
// for the (chain, band) mapping): lane jb evaluates band jb through the formulas of sed_lnlike, then
// every lane of the group adds the band terms in band order with the one-lane loop's own multiply and
// add - the same value bit for bit.  SEDLikelihood2 stays lane = chain (the host never picks G > 1).
template <int MODEL, int G>
__device__ __forceinline__ double sed_lnlike_exact_cb(const double* th, const Bands& B, const SedConsts& k,
                                                      int jb, double2* slot) {
  static_assert(MODEL != MI355_SED_M2, "SEDLikelihood2: lane = chain");
  const bool mine = jb < B.nb;
  const double2* grp = slot - jb;
  if constexpr (MODEL == MI355_SED_M1) {
    const double beta = th[0], T = th[1];
    const double pre = k.prefac * pow(k.nu_b, -beta);
    const double e3 = 3.0 + beta;
    double g = 0.0, gw = 0.0;
    if (mine) {
      g = pre * pow(B.nu(jb), e3) / expm1(B.hnu(jb) / T);
      gw = g * B.w(jb);
    }
    cb_publish(slot, gw, g);
    double FNiF = 0.0, FNid = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double2 v = grp[j];
      FNiF += v.x * v.y;
      FNid += v.x * B.d(j);
    }
    const double amp = FNid / FNiF;
    return 0.5 * FNid * amp;
  } else {
    double r = 0.0, rw = 0.0;
    if constexpr (MODEL == MI355_SED_M1N || MODEL == MI355_SED_M1N_LOGA || MODEL == MI355_SED_THICK_LOGA) {
      const double A = (MODEL == MI355_SED_M1N) ? th[0] : pow(10.0, th[0]);
      const double T = th[2], e3 = 3.0 + th[1];
      const double en = expm1(k.hk_nub / T);
      if (mine) {
        const double g = pow(B.ratio(jb), e3) * en / expm1(B.hnu(jb) / T);
        r = B.d(jb) - A * g;
      }
    } else if constexpr (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA) {
      const bool lg = (MODEL == MI355_SED_M2N_LOGA);
      const double A1 = lg ? pow(10.0, th[0]) : th[0];
      const double A2 = lg ? pow(10.0, th[3]) : th[3];
      const double T1 = th[2], T2 = th[5];
      const double e1 = 3.0 + th[1], e2 = 3.0 + th[4];
      const double en1 = expm1(k.hk_nub / T1), en2 = expm1(k.hk_nub / T2);
      if (mine) {
        const double hn = B.hnu(jb), rt = B.ratio(jb);
        const double g1 = pow(rt, e1) * en1 / expm1(hn / T1);
        const double g2 = pow(rt, e2) * en2 / expm1(hn / T2);
        r = B.d(jb) - (A1 * g1 + A2 * g2);
      }
    } else {                                               // MI355_SED_THICK
      const double A = th[0], b = th[1], T = th[2], nu0 = th[3];
      const double en = expm1(k.hk_nub / T);
      const double tau_b = pow(k.nu_b / nu0 / k.nu_rescale, b);
      const double den = 1.0 - exp(-tau_b);
      if (mine) {
        const double tau = pow(B.nu(jb) / nu0 / k.nu_rescale, b);
        const double bb = pow(B.ratio(jb), 3.0) * en / expm1(B.hnu(jb) / T);
        const double m = A * (1.0 - exp(-tau)) / den * bb;
        r = B.d(jb) - m;
      }
    }
    if (mine) rw = r * B.w(jb);
    cb_publish(slot, rw, r);
    double chi2 = 0.0;
    for (int j = 0; j < B.nb; ++j) {
      const double2 v = grp[j];
      chi2 += v.x * v.y;
    }
    return -0.5 * chi2;
  }
}

// What every caller uses: the fast path where it is valid, the exact path elsewhere.
template <int MODEL>
__device__ __forceinline__ double sed_lnlike_auto(const double* th, double prior, const Bands& B,
                                                  const SedConsts& k, double* derived) {
  constexpr int P = ModelTraits<MODEL>::P;
  if (B.fast && prior > 0.0 && all_finite<P>(th)) return sed_lnlike_fast<MODEL>(th, B, k, derived);
  return sed_lnlike<MODEL>(th, B, k, derived);
}
}  // namespace mi355
