// Kernels of the step-wise Metropolis-Hastings engine (types and helpers: gmh_device.hpp).
#pragma once
#include "gmh_device.hpp"

namespace mi355 {

// getNewParams (Proposal.py:127-164) + package + prior for every chain
// dynamic LDS: z[P][cb] doubles
__global__ __launch_bounds__(256) void k_gmh_propose(GmhArgs a) {
  extern __shared__ double gmh_lds[];
  const GmhLane t = gmh_lane(a);
  const int P = a.pb.P;
  const int64_t N = a.N, n = t.n;
  const bool inject = a.inj_n != nullptr;
  if (a.dyn) {
    a.step = a.dyn->step;
    a.blk = a.dyn->blk;
  }
  const uint64_t base = static_cast<uint64_t>(a.step) * a.npn;
  double* z = gmh_lds;
  if (a.mode == GMH_UNCORR) {
    if (t.valid) {
      for (int i0 = 2 * t.grp; i0 < P; i0 += 2 * t.G) {
        const int i1 = i0 + 1;
        const bool in0 = gmh_in_block(a, i0), in1 = i1 < P && gmh_in_block(a, i1);   // :151
        double z0, z1;
        if (inject) {
          z0 = in0 ? a.inj_n[n * P + i0] : 0.0;
          z1 = in1 ? a.inj_n[n * P + i1] : 0.0;
        } else {
          gmh_normal_pair(a, n, in0, in1, base + (in0 ? a.rank_in_block[i0] : 0),
                          base + (in1 ? a.rank_in_block[i1] : 0), z0, z1);
        }
        double v = a.theta[i0 * N + n];
        if (in0) v += z0 * a.sigma[i0 * N + n];            // :156-159
        a.theta_new[i0 * N + n] = v;
        if (i1 < P) {
          v = a.theta[i1 * N + n];
          if (in1) v += z1 * a.sigma[i1 * N + n];
          a.theta_new[i1 * N + n] = v;
        }
      }
    }
  } else {
    // the normals of the step into LDS, then one matrix row per (chain, group) turn
    if (t.valid) {
      if (a.mode == GMH_CORR) {                            // z_j = normal(base + j), all j
        if (inject) {
          for (int j = t.grp; j < P; j += t.G) z[j * t.cb + t.cl] = a.inj_n[n * P + j];
        } else {
          const int odd = static_cast<int>(base & 1);
          for (int q = t.grp; 2 * q - odd < P; q += t.G) {
            const int j0 = 2 * q - odd, j1 = j0 + 1;
            double za, zb;
            philox_normal2(a.seed, RNG_NORMAL, n, (base >> 1) + q, za, zb);
            if (j0 >= 0) z[j0 * t.cb + t.cl] = za;
            if (j1 < P) z[j1 * t.cb + t.cl] = zb;
          }
        }
      } else {                                             // rotated: the block's normals by rank
        for (int j = t.grp; j < P; j += t.G)
          if (gmh_in_block(a, j))
            z[j * t.cb + t.cl] = inject ? a.inj_n[n * P + a.rank_in_block[j]]
                                        : philox_normal(a.seed, RNG_NORMAL, n, base + a.rank_in_block[j]);
      }
    }
    __syncthreads();
    if (t.valid) {
      for (int i = t.grp; i < P; i += t.G) {
        double v = a.theta[i * N + n];
        const double* Li = a.L + static_cast<int64_t>(i) * P * N + n;
        if (a.mode == GMH_CORR) {
          if (gmh_in_block(a, i)) {                        // :153-159
            double s = 0.0;
#pragma unroll 4
            for (int j = 0; j < P; ++j) s += Li[j * N] * z[j * t.cb + t.cl];
            v += s * a.sigma[i * N + n];
          }
        } else {                                           // :145-148, sum() starts from 0
          double s = 0.0;
          for (int j = 0; j < P; ++j)
            if (gmh_in_block(a, j)) s += Li[j * N] * z[j * t.cb + t.cl];
          v += s * a.sigma[i * N + n];
        }
        a.theta_new[i * N + n] = v;
      }
    }
  }
  __syncthreads();                                         // theta_new of the block's chains is complete
  if (t.valid && t.grp == 0)
    a.prior_new[n] = gmh_package_prior(a.pb, a.theta_new + n, N);   // :164, MCMC.py:245
}

// accept / reject (MCMC.py:236-279), running moments (:290-340), trace
// dynamic LDS: d[P][cb] doubles + accepted[cb] doubles
__global__ __launch_bounds__(256) void k_gmh_accept(GmhArgs a) {
  extern __shared__ double gmh_lds[];
  const double* e2t = kExp2Table;   // the accept test's exp as in the fused kernels; one lane per chain
                                    // uses it once, so the table is read where it lies (no LDS copy)
  const GmhLane t = gmh_lane(a);
  const int P = a.pb.P;
  const int64_t N = a.N, n = t.n;
  if (a.dyn) {
    a.step = a.dyn->step;
    a.nsamp = a.dyn->nsamp;
    a.blk = a.dyn->blk;
  }
  double* d = gmh_lds;
  double* dec = gmh_lds + P * t.cb;
  if (t.valid && t.grp == 0) {
    const double prior = a.prior_new[n];
    double v = a.lnl_new[n];
    const double prev = a.lnPr[n];
    bool accepted = false;
    const bool bad = (v != v) || (v == -__builtin_inf());  // :238-243
    if (prior > 0.0 && !bad) {
      if (prior != 1.0) v += log(prior);                   // :250
      const double lnalpha = v - prev;                     // :251-252
      bool reject = false;
      if (!(lnalpha >= 0.0)) {
        const double u = a.inj_u ? a.inj_u[n]
                                 : philox_uniform(a.seed, RNG_UNIFORM, n, static_cast<uint64_t>(a.step));
        reject = !(exp_accept(lnalpha, e2t) > u);          // :264
      }
      accepted = !reject;
    }
    if (accepted) {
      a.lnPr[n] = v;
      a.naccept[n] += 1;                                   // :269
      if (a.nblock) a.accept[static_cast<int64_t>(a.blk) * N + n] += 1;   // :271
    }
    if (a.tr_theta) {
      a.tr_lnPr[n] = accepted ? v : prev;
      a.tr_acc[n] = accepted ? 1 : 0;
    }
    dec[t.cl] = accepted ? 1.0 : 0.0;
    if (n == 0 && a.dyn_next) {                            // next step's counters (graph replay):
      GmhDyn nx;                                           // the other slot, nobody reads it now
      nx.step = a.step + 1;
      nx.nsamp = a.nsamp + 1;
      nx.blk = a.nblock ? ((a.blk + 1 == a.nblock) ? 0 : a.blk + 1) : 0;
      nx.pad = 0;
      *a.dyn_next = nx;
    }
  }
  __syncthreads();
  const double cnt = static_cast<double>(a.nsamp + 1);
  const double rn = rcp_nr(cnt);                           // as the fused kernel
  const double cf = 1.0 - rn;
  if (t.valid) {
    const bool accepted = dec[t.cl] != 0.0;
#pragma unroll 4
    for (int i = t.grp; i < P; i += t.G) {
      double x = a.theta[i * N + n];
      if (accepted) {
        x = a.theta_new[i * N + n];
        a.theta[i * N + n] = x;
      }
      const double mu = a.mean[i * N + n];
      const double dlt = x - mu;
      d[i * t.cb + t.cl] = dlt;
      const double mu1 = fma(dlt, rn, mu);                // the same FMAs as the fused kernel
      a.mean[i * N + n] = mu1;
      a.m2[i * N + n] = fma(dlt, x - mu1, a.m2[i * N + n]);
      if (a.tr_theta) a.tr_theta[i * N + n] = x;
    }
  }
  if (a.comom == nullptr) return;                          // uniform over the grid
  __syncthreads();
  if (!t.valid) return;
  // C_ij += (n-1)/n d_i d_j with d = x - mean_old, j <= i (symmetric by construction); the
  // triangle's entries are dealt round-robin to the groups, eight loads in flight per thread
  const int T = P * (P + 1) / 2;
  constexpr int U = 8;
  for (int e0 = t.grp; e0 < T; e0 += t.G * U) {
    double* ptr[U];
    double inc[U], cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * t.G;
      ptr[u] = nullptr;
      if (e < T) {
        int i, j;
        gmh_tri_decode(e, i, j);
        ptr[u] = a.comom + (static_cast<int64_t>(i) * P + j) * N + n;
        inc[u] = cf * d[i * t.cb + t.cl] * d[j * t.cb + t.cl];
        cur[u] = *ptr[u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ptr[u]) *ptr[u] = cur[u] + inc[u];
  }
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
