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
__device__ __forceinline__ GmhLane gmh_lane(const GmhArgs& a) {
  GmhLane t;
  t.cb = 1 << a.cb_shift;
  t.G = 256 >> a.cb_shift;
  t.cl = threadIdx.x & (t.cb - 1);
  t.grp = threadIdx.x >> a.cb_shift;
  t.n = static_cast<int64_t>(blockIdx.x) * t.cb + t.cl;
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

}  // namespace mi355
