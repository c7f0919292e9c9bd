// Step-wise Metropolis-Hastings for any problem kind (SED, binnedCl, BeamFit) and any proposal
// of MCMC/Proposal.py: uncorrelated, Cholesky-correlated (:150-154) and rotated (:145-148), with
// parameter blocks (MCMC.py:229-232).  One MH step = k_gmh_propose -> the problem's batched
// likelihood -> k_gmh_accept; chain state is parameter-major [P][N] in HBM so every access of a
// wave is 512 contiguous bytes.  The SED fused kernel (sed_kernels.hpp) stays the fast path for
// uncorrelated SED proposals; this engine is what serves problems whose likelihood is itself a
// multi-kernel pipeline (Cl: MFMA quadratic forms; beam: per-pixel reductions).
//
// Reference semantics reproduced: MCMC.py:138-187 (setStart), :225-286 (sample), :290-340
// (mean / stdev / covar), :343-413 (newMCMC); Proposal.py:81-101 (Cholesky on the sigma>0
// subset), :127-164 (getNewParams).
#pragma once
#include "beam_device.hpp"
#include "philox.hpp"
#include "sed_device.hpp"

namespace mi355 {

enum GmhKind : int { GMH_SED = 0, GMH_CL = 1, GMH_BEAM = 2 };
enum GmhMode : int { GMH_UNCORR = 0, GMH_CORR = 1, GMH_ROTATE = 2 };
constexpr int kGmhMaxP = 256;
constexpr int kGmhStartMaxTries = 100000;
constexpr int kGmhFlagStartFailed = 1;
constexpr int kGmhFlagCholFailed = 2;

struct GmhProblem {
  int kind;
  int P;
  int sed_model;
  SedConsts k;
  BeamPriorBox box;
  const int32_t* cltype;   // [P] (Cl)
};

template <int M>
__device__ __forceinline__ double gmh_sed_pp(const GmhProblem& pb, double* col, int64_t N) {
  constexpr int P = ModelTraits<M>::P;
  double th[P];
#pragma unroll
  for (int i = 0; i < P; ++i) th[i] = col[i * N];
  sed_package<M>(th);
#pragma unroll
  for (int i = 0; i < P; ++i) col[i * N] = th[i];
  return sed_prior<M>(th, pb.k);
}

// package(th) in place (th(i) = col[i*N]) and return model.prior(*package(th))
__device__ __forceinline__ double gmh_package_prior(const GmhProblem& pb, double* col, int64_t N) {
  if (pb.kind == GMH_CL) {                                 // binnedClModel.py:125-132, package = asarray
    double p = 1.0;
#pragma unroll 8
    for (int b = 0; b < pb.P; ++b)
      if (pb.cltype[b] != 1 && col[b * N] < 0) p = 0.0;
    return p;
  }
  if (pb.kind == GMH_BEAM) {
    double th[kBeamP];
#pragma unroll
    for (int i = 0; i < kBeamP; ++i) th[i] = col[i * N];
    beam_package(th);
    col[4 * N] = th[4];
    col[7 * N] = th[7];
    col[8 * N] = th[8];
    return beam_prior(th, pb.box);
  }
  switch (pb.sed_model) {
    case MI355_SED_M1N: return gmh_sed_pp<MI355_SED_M1N>(pb, col, N);
    case MI355_SED_M1N_LOGA: return gmh_sed_pp<MI355_SED_M1N_LOGA>(pb, col, N);
    case MI355_SED_M2N: return gmh_sed_pp<MI355_SED_M2N>(pb, col, N);
    case MI355_SED_M2N_LOGA: return gmh_sed_pp<MI355_SED_M2N_LOGA>(pb, col, N);
    case MI355_SED_M1: return gmh_sed_pp<MI355_SED_M1>(pb, col, N);
    case MI355_SED_M2: return gmh_sed_pp<MI355_SED_M2>(pb, col, N);
    case MI355_SED_THICK: return gmh_sed_pp<MI355_SED_THICK>(pb, col, N);
    default: return gmh_sed_pp<MI355_SED_THICK_LOGA>(pb, col, N);
  }
}

// Per-step counters kept on the device so that a captured hipGraph of MH steps can be replayed:
// the kernels of a graph have frozen arguments, what changes from step to step lives here.
struct GmhDyn {
  int64_t step;    // global step (Philox position)
  int64_t nsamp;   // samples in the moments before this step
  int32_t blk;     // block of this step
  int32_t pad;
};

struct GmhArgs {
  GmhProblem pb;
  int64_t N;
  double* theta;            // [P][N]
  double* sigma;            // [P][N]
  double* lnPr;             // [N]
  double* mean;             // [P][N]
  double* m2;               // [P][N]
  double* comom;            // [P][P][N] lower triangle used (nullable: covariance not tracked)
  uint32_t* naccept;        // [N]
  uint32_t* accept;         // [nblock][N]
  int32_t* flags;           // [N]
  double* theta_new;        // [P][N]
  double* prior_new;        // [N]
  double* lnl_new;          // [N]
  double* zbuf;             // [P][N] scratch normals (correlated / rotated modes)
  const double* L;          // [P][P][N] sqrtMatrix per chain (modes 1, 2)
  const int32_t* block_of;  // [P]
  const int32_t* rank_in_block;   // [P]
  int nblock;               // 0: doBlock False
  int blk;                  // block of this step
  int mode;
  int npn;                  // normals per step in the Philox stream
  uint64_t seed;
  int64_t step;             // global step (Philox position)
  int64_t nsamp;            // samples in the moments before this step
  int64_t stage;
  const GmhDyn* dyn;        // non-null: step / nsamp / blk come from here (graph replay)
  // injection (parity mode)
  const double* inj_n;      // [P] normals of this step for chain n at inj_n[n*P + .]
  const double* inj_u;      // [N]
  // trace
  double* tr_theta;         // row base [P][N] or null
  double* tr_lnPr;
  uint8_t* tr_acc;
  // start / adapt
  const double* center;     // [P][N] or [P]
  int center_per_chain;
  const double* sigma_in;   // [P][N] or [P]
  int sigma_per_chain;
  int randomize;
  double fac;
  int random_restart;
  int want_corr;            // adapt: build the Cholesky factor
  double* Lout;             // adapt: [P][P][N]
};

__device__ __forceinline__ bool gmh_in_block(const GmhArgs& a, int i) {
  return a.nblock == 0 || a.block_of[i] == a.blk;
}

// getNewParams (Proposal.py:127-164) + package + prior for every chain
__global__ __launch_bounds__(256) void k_gmh_propose(GmhArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int P = a.pb.P;
  const int64_t N = a.N;
  const bool inject = a.inj_n != nullptr;
  if (a.dyn) {
    a.step = a.dyn->step;
    a.blk = a.dyn->blk;
  }
  const uint64_t base = static_cast<uint64_t>(a.step) * a.npn;
  if (a.mode == GMH_UNCORR) {
#pragma unroll 8
    for (int i = 0; i < P; ++i) {
      double v = a.theta[i * N + n];
      if (gmh_in_block(a, i)) {                            // :151, :156-159
        const double z = inject ? a.inj_n[n * P + i]
                                : philox_normal(a.seed, RNG_NORMAL, n, base + a.rank_in_block[i]);
        v += z * a.sigma[i * N + n];
      }
      a.theta_new[i * N + n] = v;
    }
  } else if (a.mode == GMH_CORR) {
    for (int j = 0; j < P; ++j)
      a.zbuf[j * N + n] = inject ? a.inj_n[n * P + j] : philox_normal(a.seed, RNG_NORMAL, n, base + j);
    for (int i = 0; i < P; ++i) {
      double v = a.theta[i * N + n];
      if (gmh_in_block(a, i)) {                            // :153-159
        double s = 0.0;
        for (int j = 0; j < P; ++j) s += a.L[(static_cast<int64_t>(i) * P + j) * N + n] * a.zbuf[j * N + n];
        v += s * a.sigma[i * N + n];
      }
      a.theta_new[i * N + n] = v;
    }
  } else {                                                 // rotated, :145-148
    for (int j = 0; j < P; ++j)
      if (gmh_in_block(a, j))
        a.zbuf[j * N + n] = inject ? a.inj_n[n * P + a.rank_in_block[j]]
                                   : philox_normal(a.seed, RNG_NORMAL, n, base + a.rank_in_block[j]);
    for (int i = 0; i < P; ++i) {
      double s = 0.0;                                      // sum() starts from 0
      for (int j = 0; j < P; ++j)
        if (gmh_in_block(a, j)) s += a.L[(static_cast<int64_t>(i) * P + j) * N + n] * a.zbuf[j * N + n];
      a.theta_new[i * N + n] = a.theta[i * N + n] + s * a.sigma[i * N + n];
    }
  }
  a.prior_new[n] = gmh_package_prior(a.pb, a.theta_new + n, N);   // :164, MCMC.py:245
}

// accept / reject (MCMC.py:236-279), running moments (:290-340), trace
__global__ __launch_bounds__(256) void k_gmh_accept(GmhArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int P = a.pb.P;
  const int64_t N = a.N;
  if (a.dyn) {
    a.step = a.dyn->step;
    a.nsamp = a.dyn->nsamp;
    a.blk = a.dyn->blk;
  }
  const double prior = a.prior_new[n];
  double v = a.lnl_new[n];
  const double prev = a.lnPr[n];
  bool accepted = false;
  const bool bad = (v != v) || (v == -__builtin_inf());    // :238-243
  if (prior > 0.0 && !bad) {
    if (prior != 1.0) v += log(prior);                     // :250
    const double lnalpha = v - prev;                       // :251-252
    bool reject = false;
    if (!(lnalpha >= 0.0)) {
      const double u = a.inj_u ? a.inj_u[n]
                               : philox_uniform(a.seed, RNG_UNIFORM, n, static_cast<uint64_t>(a.step));
      reject = !(exp_fast(lnalpha) > u);                   // :264
    }
    accepted = !reject;
  }
  if (accepted) {
    a.lnPr[n] = v;
    a.naccept[n] += 1;                                     // :269
    if (a.nblock) a.accept[static_cast<int64_t>(a.blk) * N + n] += 1;   // :271
  }
  const double lp = accepted ? v : prev;
  const double cnt = static_cast<double>(a.nsamp + 1);
  const double rn = rcp_nr(cnt);                           // as the fused kernel
  const double cf = 1.0 - rn;
#pragma unroll 4
  for (int i = 0; i < P; ++i) {
    double x = a.theta[i * N + n];
    if (accepted) {
      x = a.theta_new[i * N + n];
      a.theta[i * N + n] = x;
    }
    const double mu = a.mean[i * N + n];
    const double dlt = x - mu;
    if (a.comom) {
      // C_ij += (n-1)/n d_i d_j with d = x - mean_old, j <= i (symmetric by construction);
      // d_j of the earlier parameters is kept in zbuf
      a.zbuf[i * N + n] = dlt;
      for (int j = 0; j <= i; ++j)
        a.comom[(static_cast<int64_t>(i) * P + j) * N + n] += cf * dlt * a.zbuf[j * N + n];
    }
    const double mu1 = mu + dlt * rn;
    a.mean[i * N + n] = mu1;
    a.m2[i * N + n] += dlt * (x - mu1);
    if (a.tr_theta) a.tr_theta[i * N + n] = x;
  }
  if (a.tr_theta) {
    a.tr_lnPr[n] = lp;
    a.tr_acc[n] = accepted ? 1 : 0;
  }
}

// one thread: next step's counters (graph replay)
__global__ void k_gmh_advance(GmhDyn* dyn, int nblock) {
  dyn->step += 1;
  dyn->nsamp += 1;
  if (nblock) dyn->blk = (dyn->blk + 1 == nblock) ? 0 : dyn->blk + 1;
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
