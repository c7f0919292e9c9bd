// Kernels of the step-wise Metropolis-Hastings engine (types and helpers: gmh_device.hpp).
#pragma once
#include "gmh_device.hpp"

namespace mi355 {

// getNewParams + package + prior for every chain (gmh_propose_block, gmh_device.hpp); dynamic LDS: z[P][cb] doubles
__global__ __launch_bounds__(256) void k_gmh_propose(GmhArgs a) {
  extern __shared__ double gmh_lds[];
  gmh_propose_block(a, blockIdx.x, gmh_lds);
}

// accept / reject, running moments, trace (gmh_accept_block); dynamic LDS: d[P][cb] doubles + accepted[cb] doubles
__global__ __launch_bounds__(256) void k_gmh_accept(GmhArgs a) {
  extern __shared__ double gmh_lds[];
  gmh_accept_block(a, blockIdx.x, gmh_lds);
}

// The accept of step t and the proposal of step t + 1 in one launch: both work chain by chain on the
// same thread layout, and a step of a small job is bound by its launches (one 64 x 64 beam map, one
// chain - the reference's own use, ProcessBeamData.py:277-279 - took four launches and 15.6 us per step
// under hipGraph replay; two launches - this one and the chi-square, whose finish the accept does
// itself, GmhArgs::lnl_partial - take half).  The same bodies in the same order: the same chain.
// Under graph replay (acc.dyn) the proposal's counters are the accept's plus one step.
__global__ __launch_bounds__(256) void k_gmh_accept_propose(GmhArgs acc, GmhArgs prop) {
  extern __shared__ double gmh_lds[];
  if (acc.dyn) {
    const GmhDyn d = *acc.dyn;
    prop.dyn = nullptr;
    prop.step = d.step + 1;
    prop.nsamp = d.nsamp + 1;
    prop.blk = prop.nblock ? ((d.blk + 1 == prop.nblock) ? 0 : d.blk + 1) : 0;
  }
  gmh_accept_block(acc, blockIdx.x, gmh_lds);
  __syncthreads();                                         // theta of the block's chains is final; LDS is free
  gmh_propose_block(prop, blockIdx.x, gmh_lds);
}

// newMCMC (MCMC.py:343-413): centre = package(mean), stdevs = package(stdev), sigma =
// package(fac*stdev), optional Cholesky factor of the normalised covariance on the sigma>0
// subset (Proposal.py:81-101).  Writes centre -> theta_new, sigma -> sigma, L -> Lout.
__global__ __launch_bounds__(256) void k_gmh_adapt(GmhArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int P = a.pb.P;
  const int64_t N = a.N;
  const double rn = 1.0 / static_cast<double>(a.nsamp);
  double* ctr = a.theta_new;
  double* sd = a.zbuf;
  for (int i = 0; i < P; ++i) {
    ctr[i * N + n] = a.mean[i * N + n];
    sd[i * N + n] = sqrt(a.m2[i * N + n] * rn);
  }
  (void)gmh_package_prior(a.pb, ctr + n, N);               // newStart = package(mean)    :363
  (void)gmh_package_prior(a.pb, sd + n, N);                // stdevs = package(stdev)     :364
  int flag = 0;
  if (a.want_corr) {
    // covar(): <(x_i-m_i)(x_j-m_j)> / (s_i s_j) with the packaged stdevs            :321-333
    double* A = a.Lout;
    for (int i = 0; i < P; ++i)
      for (int j = 0; j < P; ++j) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        const double c = a.comom[(static_cast<int64_t>(hi) * P + lo) * N + n] * rn;
        A[(static_cast<int64_t>(i) * P + j) * N + n] = c / (sd[i * N + n] * sd[j * N + n]);
      }
  }
  for (int i = 0; i < P; ++i) a.sigma[i * N + n] = a.fac * sd[i * N + n];   // :385
  (void)gmh_package_prior(a.pb, a.sigma + n, N);
  if (a.random_restart) {                                  // :366
    const uint64_t sbase = static_cast<uint64_t>(a.stage) << 40;
    for (int i = 0; i < P; ++i)
      ctr[i * N + n] += philox_normal(a.seed, RNG_START, n, sbase + i) * sd[i * N + n];
  }
  if (a.want_corr) {
    // in-place Cholesky (lower) over the parameters with sigma > 0; the strict upper triangle
    // keeps the input so a failed factorisation can be repeated with +0.01 on the diagonal
    double* A = a.Lout;
    for (int attempt = 0; attempt < 2; ++attempt) {
      bool ok = true;
      for (int i = 0; i < P && ok; ++i) {
        if (!(a.sigma[i * N + n] > 0)) continue;
        for (int j = 0; j <= i; ++j) {
          if (!(a.sigma[j * N + n] > 0)) continue;
          double s = (j == i) ? A[(static_cast<int64_t>(i) * P + i) * N + n]
                              : A[(static_cast<int64_t>(j) * P + i) * N + n];   // upper copy
          if (j == i && attempt == 1) s += 0.01;           // MCMC.py:397-399
          for (int k = 0; k < j; ++k)
            if (a.sigma[k * N + n] > 0)
              s -= A[(static_cast<int64_t>(i) * P + k) * N + n] * A[(static_cast<int64_t>(j) * P + k) * N + n];
          if (j == i) {
            if (!(s > 0.0)) { ok = false; break; }         // LinAlgError
            // keep the diagonal input in zbuf-free storage: the diagonal is recomputable
            A[(static_cast<int64_t>(i) * P + i) * N + n] = sqrt(s);
          } else {
            A[(static_cast<int64_t>(i) * P + j) * N + n] = s / A[(static_cast<int64_t>(j) * P + j) * N + n];
          }
        }
      }
      if (ok) break;
      if (attempt == 1) { flag = kGmhFlagCholFailed; break; }
      // restore the diagonal (normalised covariance: c_ii / s_i^2) for the retry
      for (int i = 0; i < P; ++i) {
        const double c = a.comom[(static_cast<int64_t>(i) * P + i) * N + n] * rn;
        A[(static_cast<int64_t>(i) * P + i) * N + n] = c / (sd[i * N + n] * sd[i * N + n]);
      }
    }
    // zero everything that is not the factor (upper triangle, fixed parameters)
    for (int i = 0; i < P; ++i)
      for (int j = 0; j < P; ++j) {
        const bool keep = j <= i && a.sigma[i * N + n] > 0 && a.sigma[j * N + n] > 0;
        if (!keep) A[(static_cast<int64_t>(i) * P + j) * N + n] = 0.0;
      }
  }
  a.flags[n] = flag;
}

// setStart(random=True) (MCMC.py:151-159): theta = package(centre + N(0,1)*sigma) over all
// parameters, redrawn until prior > 0.  centre comes from a.center (init) or theta_new (adapt).
__global__ __launch_bounds__(256) void k_gmh_start(GmhArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int P = a.pb.P;
  const int64_t N = a.N;
  const int PE = (P + 1) & ~1;
  const uint64_t sbase = static_cast<uint64_t>(a.stage) << 40;
  int flag = a.flags[n];
  if (a.sigma_in)
    for (int i = 0; i < P; ++i)
      a.sigma[i * N + n] = a.sigma_per_chain ? a.sigma_in[i * N + n] : a.sigma_in[i];
  double prior = -1.0;
  if (a.randomize) {
    int tries = 0;
    while (!(prior > 0.0)) {
      ++tries;
      if (tries > kGmhStartMaxTries) { flag |= kGmhFlagStartFailed; break; }
      const uint64_t base = sbase + static_cast<uint64_t>(tries) * PE;
      for (int i = 0; i < P; ++i) {
        const double c = a.center ? (a.center_per_chain ? a.center[i * N + n] : a.center[i])
                                  : a.theta_new[i * N + n];
        a.theta[i * N + n] = c + philox_normal(a.seed, RNG_START, n, base + i) * a.sigma[i * N + n];
      }
      prior = gmh_package_prior(a.pb, a.theta + n, N);
    }
  } else {
    for (int i = 0; i < P; ++i)
      a.theta[i * N + n] = a.center ? (a.center_per_chain ? a.center[i * N + n] : a.center[i])
                                    : a.theta_new[i * N + n];
    prior = gmh_package_prior(a.pb, a.theta + n, N);
    if (!(prior > 0.0)) flag |= kGmhFlagStartFailed;       // ZeroPriorError, MCMC.py:174-175
  }
  a.prior_new[n] = prior;
  a.flags[n] = flag;
}

// lnPr[0] = lnLike + log(prior), counters and moments reset (MCMC.py:177-183)
__global__ __launch_bounds__(256) void k_gmh_start_finish(GmhArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int P = a.pb.P;
  const int64_t N = a.N;
  double lp = __builtin_nan("");
  if (!(a.flags[n] & kGmhFlagStartFailed)) {
    lp = a.lnl_new[n];
    const double prior = a.prior_new[n];
    if (prior != 1.0) lp += log(prior);
  }
  a.lnPr[n] = lp;
  a.naccept[n] = 1;
  for (int b = 0; b < a.nblock; ++b) a.accept[static_cast<int64_t>(b) * N + n] = 1;
  for (int i = 0; i < P; ++i) {
    a.mean[i * N + n] = 0.0;
    a.m2[i * N + n] = 0.0;
    if (a.comom)
      for (int j = 0; j <= i; ++j) a.comom[(static_cast<int64_t>(i) * P + j) * N + n] = 0.0;
  }
}

// [N][P] -> [P][N]
__global__ void k_gmh_transpose_np(int64_t N, int P, const double* __restrict__ in,
                                   double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= N * P) return;
  const int64_t n = e % N;
  const int i = static_cast<int>(e / N);
  out[e] = in[n * P + i];
}

// host [N][P][P] (per chain) or [P][P] (shared) -> device [P][P][N]
__global__ void k_gmh_spread_matrix(int64_t N, int P, const double* __restrict__ in, int per_chain,
                                    double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= N * P * P) return;
  const int64_t n = e % N;
  const int64_t ij = e / N;
  out[e] = per_chain ? in[n * P * P + ij] : in[ij];
}

// device [P][P][N] -> [N][P][P]; scale applies (covariance = comoment / nsamp); symmetric fill
// from the lower triangle when `sym`; normalised by stdev_i stdev_j when m2 != null
__global__ void k_gmh_gather_matrix(int64_t N, int P, const double* __restrict__ in, double scale,
                                    int sym, const double* __restrict__ m2, double rn,
                                    double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= N * P * P) return;
  const int64_t n = e / (P * P);
  const int ij = static_cast<int>(e % (P * P));
  int i = ij / P, j = ij % P;
  if (sym && j > i) { const int t = i; i = j; j = t; }
  double v = in[(static_cast<int64_t>(i) * P + j) * N + n] * scale;
  if (m2) v /= sqrt(m2[i * N + n] * rn) * sqrt(m2[j * N + n] * rn);
  out[e] = v;
}

}  // namespace mi355
