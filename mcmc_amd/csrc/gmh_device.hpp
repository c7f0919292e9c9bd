// Types and device helpers of the step-wise Metropolis-Hastings engine (its kernels are in
// gmh_kernels.hpp; the fused binnedCl kernel, cl_mh_kernels.hpp, shares these).
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
#include "fastmath.hpp"

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
  const double* lnl_partial;   // non-null: lnL[n] = -1/2 sum_j lnl_partial[n][j], j < lnl_nchunk ascending,
  int lnl_nchunk;              // the finish of the beam chi-square (k_beam_finish) folded into the accept step
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
  GmhDyn* dyn_next;         // non-null: k_gmh_accept writes the next step's counters here
  int cb_shift;             // log2(chains per block) of the step kernels
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

// Thread layout of the two step kernels: a block of 256 threads serves cb = 2^cb_shift chains,
// thread = (chain-in-block, group), the G = 256/cb groups share the per-parameter work of a
// chain.  Loads are coalesced over the chains (cb*8 bytes per segment) and a step costs a few
// memory round trips per thread instead of P (or P^2/2 for the co-moments) in sequence.
struct GmhLane {
  int cb, G, cl, grp;
  int64_t n;
  bool valid;
};
__device__ __forceinline__ GmhLane gmh_lane(const GmhArgs& a, int64_t vblock) {
  GmhLane t;
  t.cb = 1 << a.cb_shift;
  t.G = 256 >> a.cb_shift;
  t.cl = threadIdx.x & (t.cb - 1);
  t.grp = threadIdx.x >> a.cb_shift;
  t.n = vblock * t.cb + t.cl;
  t.valid = t.n < a.N;
  return t;
}

// normals of parameters (i0, i0+1) with stream indices x0, x1; one Philox block + Box-Muller
// when the two are the halves of one pair (same values as two philox_normal calls)
__device__ __forceinline__ void gmh_normal_pair(const GmhArgs& a, int64_t n, bool in0, bool in1,
                                                uint64_t x0, uint64_t x1, double& z0, double& z1) {
  z0 = 0.0;
  z1 = 0.0;
  if (in0 && in1 && !(x0 & 1) && x1 == x0 + 1) {
    philox_normal2(a.seed, RNG_NORMAL, n, x0 >> 1, z0, z1);
  } else {
    if (in0) z0 = philox_normal(a.seed, RNG_NORMAL, n, x0);
    if (in1) z1 = philox_normal(a.seed, RNG_NORMAL, n, x1);
  }
}

// (i, j), j <= i, of entry e of the row-major lower triangle
__device__ __forceinline__ void gmh_tri_decode(int e, int& i, int& j) {
  i = static_cast<int>((sqrtf(8.0f * static_cast<float>(e) + 1.0f) - 1.0f) * 0.5f);
  while (i * (i + 1) / 2 > e) --i;
  while ((i + 1) * (i + 2) / 2 <= e) ++i;
  j = e - i * (i + 1) / 2;
}

// C_ij += (n-1)/n d_i d_j, j <= i (MCMC.py:310-340's running covariance, symmetric by construction), for chain
// n: `d` is the chain's column of x - mean_old (LDS, rows `ds` doubles apart); member j0 of the js lanes that
// share the chain takes columns j0, j0 + js, ... of every row (entries j0, j0 + js, ... of the triangle in the
// second form).  Two forms of the same sums:
//  * adds that return nothing (add_noret) - no round trips, which is what a job of few chains waits for
//    (4096 chains x 32 parameters: 54 -> 19 us per step in the fused Cl wave kernel, 87 -> 61 in the step-wise
//    engine); L2 retires about one such add per channel and clock, 2.3e11 entries/s measured;
//  * load / add / store, eight loads in flight per lane - bound by bandwidth, 2.8e11 entries/s at 65 536 chains.
// gmh_comom_atomic picks by the entries of one step (uniform over the grid).
__device__ __forceinline__ bool gmh_comom_atomic(int64_t N, int P) {
  return N * (P * (P + 1) / 2) <= kComomAtomicMaxEntries;
}
__device__ __forceinline__ void gmh_comom_update(double* comom, int64_t N, int64_t n, int P, double cf,
                                                 const double* d, int ds, int j0, int js) {
  if (gmh_comom_atomic(N, P)) {
    double* row = comom + n;
    for (int i = 0; i < P; ++i, row += static_cast<int64_t>(P) * N) {
      const double ci = cf * d[i * ds];
      for (int j = j0; j <= i; j += js) add_noret(row + static_cast<int64_t>(j) * N, ci * d[j * ds]);
    }
    return;
  }
  // Branch-free batches (a branch inside makes the compiler wait for every load and store in turn): entries
  // beyond the triangle go to the chain's entry (0, 1) - the upper triangle is never read - with increment 0.
  const int T = P * (P + 1) / 2;
  if (P < 2) {                                             // (no upper triangle: one entry, one lane)
    if (j0 == 0) comom[n] = comom[n] + cf * d[0] * d[0];
    return;
  }
  double* const spare = comom + N + n;
  constexpr int U = 8;
  for (int e0 = j0; e0 < T; e0 += js * U) {
    double* ptr[U];
    double inc[U], cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * js;
      const bool in = e < T;
      const int ec = in ? e : 0;
      int i = static_cast<int>((sqrtf(8.0f * static_cast<float>(ec) + 1.0f) - 1.0f) * 0.5f);   // gmh_tri_decode
      i -= (i * (i + 1) / 2 > ec) ? 1 : 0;
      i += ((i + 1) * (i + 2) / 2 <= ec) ? 1 : 0;
      const int j = ec - i * (i + 1) / 2;
      double* q = comom + (static_cast<int64_t>(i) * P + j) * N + n;
      ptr[u] = in ? q : spare;
      const double v = cf * d[i * ds] * d[j * ds];
      inc[u] = in ? v : 0.0;
      cur[u] = ldas<1>(ptr[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) stas<1>(ptr[u], cur[u] + inc[u]);
  }
}

// getNewParams (Proposal.py:127-164) + package + prior for every chain
// (the body of k_gmh_propose for block `vblock` of the grid; k_gmh_accept_propose runs it for the next
// step behind the accept of this one)
// LDS: z[P][cb] doubles
__device__ __forceinline__ void gmh_propose_block(GmhArgs a, int64_t vblock, double* gmh_lds) {
  const GmhLane t = gmh_lane(a, vblock);
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
// LDS: d[P][cb] doubles + accepted[cb] doubles
__device__ __forceinline__ void gmh_accept_block(GmhArgs a, int64_t vblock, double* gmh_lds) {
  const double* e2t = kExp2Table;   // the accept test's exp as in the fused kernels; one lane per chain
                                    // uses it once, so the table is read where it lies (no LDS copy)
  const GmhLane t = gmh_lane(a, vblock);
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
    double v;
    if (a.lnl_partial) {                                   // k_beam_finish's sum, NormalizedBeamLikelihood.py:16
      double s = 0.0;
      for (int j = 0; j < a.lnl_nchunk; ++j) s += a.lnl_partial[n * a.lnl_nchunk + j];
      v = -0.5 * s;
    } else {
      v = a.lnl_new[n];
    }
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
  // C_ij += (n-1)/n d_i d_j with d = x - mean_old, the columns dealt round-robin to the chain's groups
  gmh_comom_update(a.comom, N, n, P, cf, d + t.cl, t.cb, t.grp, t.G);
}

}  // namespace mi355
