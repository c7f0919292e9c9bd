// HIP kernels of the SED path: batched likelihood, chain start / adapt, fused MH steps.
// Mapping: one lane = one (source, chain); a 256-thread block covers 256 consecutive chain
// indices n = s*C + c, stages the band tables of the sources it touches in LDS once and keeps
// every chain's state in registers for all the steps of a launch; HBM is touched only at launch
// entry/exit (parameter-major arrays, so the 64 lanes of a wave read/write 512 contiguous bytes)
// and for the optional trace.
#pragma once
#include "philox.hpp"
#include "sed_device.hpp"

namespace mi355 {

struct MhArgs {
  const double* bands;    // [S][nbm][kBandFields]
  const int32_t* nb;      // [S]
  int nbm;
  int C;
  int64_t N;
  double* theta;          // [P][N]
  double* sigma;          // [P][N]
  double* lnPr;           // [N]
  double* mean;           // [P][N]
  double* m2;             // [P][N]
  uint32_t* naccept;      // [N]
  uint32_t* accept;       // [nblock][N]
  int32_t* flags;         // [N]
  int nblock;
  uint32_t blockmask[kMaxParam];
  int32_t rank_in_block[kMaxParam];
  int npn;
  int64_t nsteps;
  int64_t step_total;
  int64_t stage_step;
  int64_t nsamp0;
  int64_t stage;
  uint64_t seed;
  const double* inj_n;    // [nsteps][N][P] or null
  const double* inj_u;    // [nsteps][N] or null
  int thin;
  double* tr_theta;       // [nkeep][P][N]
  double* tr_lnPr;        // [nkeep][N]
  uint8_t* tr_acc;        // [nkeep][N]
  SedConsts k;
  // start / adapt
  const double* center;   // [P][N] or [P]
  int center_per_chain;
  const double* sigma_in; // [P][N] or [P]
  int sigma_per_chain;
  int randomize;
  int adapt;              // 1: centre and sigma come from the chain's own moments
  double fac;
  int random_restart;
  // extended sampler (k_mh_run<.., EXT = true>): the proposals of Proposal.py:145-154 and the
  // running co-moments of covar() (MCMC.py:310-340), for chains started by mi355_mh_init_problem
  int mode;               // 0 uncorrelated, 1 Cholesky-correlated, 2 rotated (GmhMode)
  const double* L;        // [P][P][N] sqrtMatrix per chain (modes 1, 2)
  double* comom;          // [P][P][N], lower triangle used; null: covariance not tracked
  // k_mh_run: the blocks (of blockDim chains) this launch runs, or null for all of them
  const int32_t* blocklist;
};

constexpr int kStartMaxTries = 100000;
constexpr int kFlagStartFailed = 1;

// Stage the band tables of sources [s_first, s_first+nsrc) into LDS (coalesced).
__device__ __forceinline__ void stage_bands(double* smem, const double* bands, int64_t s_first,
                                            int nsrc, int stride) {
  const double* src = bands + s_first * stride;
  const int total = nsrc * stride;
  for (int i = threadIdx.x; i < total; i += blockDim.x) smem[i] = src[i];
  __syncthreads();
}

// ---- batched like.lnLike ------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(kBlockThreads) void k_sed_lnlike(
    const double* __restrict__ bands, const int32_t* __restrict__ nb, int nbm, int64_t s0,
    int64_t n_eval, int C, const double* __restrict__ theta, double* __restrict__ lnl,
    double* __restrict__ derived, SedConsts k) {
  constexpr int P = ModelTraits<MODEL>::P;
  extern __shared__ double smem[];
  const int stride = kBandFields * nbm;
  const int64_t e0 = static_cast<int64_t>(blockIdx.x) * blockDim.x;
  const int64_t e_last = min(e0 + blockDim.x, n_eval) - 1;
  const int64_t s_first = s0 + e0 / C;
  const int nsrc = static_cast<int>(s0 + e_last / C - s_first) + 1;
  const double* e2t = stage_exp2_table();
  stage_bands(smem, bands, s_first, nsrc, stride);
  const int64_t e = e0 + threadIdx.x;
  if (e >= n_eval) return;
  const int64_t s = s0 + e / C;
  double th[P];
#pragma unroll
  for (int i = 0; i < P; ++i) th[i] = theta[i * n_eval + e];
  const Bands B = make_bands(smem + (s - s_first) * stride, nbm, nb[s], e2t, k);
  double der[2];
  const double v = sed_lnlike_auto<MODEL>(th, sed_prior<MODEL>(th, k), B, k, derived ? der : nullptr);
  lnl[e] = v;
  if (derived) {
    derived[e] = der[0];
    derived[n_eval + e] = der[1];
  }
}

template <int MODEL>
__global__ __launch_bounds__(kBlockThreads) void k_sed_prior_package(
    int64_t n, const double* __restrict__ theta /*[n][P]*/, double* __restrict__ prior,
    double* __restrict__ packaged /*[n][P]*/, SedConsts k) {
  constexpr int P = ModelTraits<MODEL>::P;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double th[P];
#pragma unroll
  for (int i = 0; i < P; ++i) th[i] = theta[e * P + i];
  if (prior) prior[e] = sed_prior<MODEL>(th, k);
  if (packaged) {
    sed_package<MODEL>(th);
#pragma unroll
    for (int i = 0; i < P; ++i) packaged[e * P + i] = th[i];
  }
}

template <int MODEL>
__global__ __launch_bounds__(kBlockThreads) void k_sed_flux(
    int64_t n, const double* __restrict__ theta, int64_t nnu, const double* __restrict__ nu,
    double* __restrict__ out, SedConsts k) {
  constexpr int P = ModelTraits<MODEL>::P;
  constexpr int NC = (MODEL == MI355_SED_M2) ? 2 : 1;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * nnu) return;
  const int64_t it = e / nnu, in = e % nnu;
  double th[P];
#pragma unroll
  for (int i = 0; i < P; ++i) th[i] = theta[it * P + i];
  double f[2] = {0.0, 0.0};
  sed_flux<MODEL>(th, nu[in], k, f);
#pragma unroll
  for (int c = 0; c < NC; ++c) out[e * NC + c] = f[c];
}

// ---- chain start (setStart, MCMC.py:138-187) and restart from moments (newMCMC, :343-413) ---
template <int MODEL>
__global__ __launch_bounds__(kBlockThreads) void k_mh_start(MhArgs a) {
  constexpr int P = ModelTraits<MODEL>::P;
  constexpr int PE = (P + 1) & ~1;   // normals per start draw, rounded up to whole Philox pairs
  extern __shared__ double smem[];
  const int stride = kBandFields * a.nbm;
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * blockDim.x;
  const int64_t n_last = min(n0 + blockDim.x, a.N) - 1;
  const int64_t s_first = n0 / a.C;
  const int nsrc = static_cast<int>(n_last / a.C - s_first) + 1;
  const double* e2t = stage_exp2_table();
  const double* bmt = bm_of(e2t);                        // Box-Muller rows of the same LDS table
  stage_bands(smem, a.bands, s_first, nsrc, stride);
  const int64_t n = n0 + threadIdx.x;
  if (n >= a.N) return;
  const int64_t s = n / a.C;
  const Bands B = make_bands(smem + (s - s_first) * stride, a.nbm, a.nb[s], e2t, a.k);
  const uint64_t sbase = static_cast<uint64_t>(a.stage) << 40;

  double c[P], sg[P];
  if (a.adapt) {
    // newStart = package(mean), stdevs = package(stdev) - the reference canonicalises the
    // stdev vector with the same T1>T2 swap rule (MCMC.py:299-307 -> model.package), kept.
    double sd[P];
    const double rn = 1.0 / static_cast<double>(a.nsamp0);
#pragma unroll
    for (int i = 0; i < P; ++i) {
      c[i] = a.mean[i * a.N + n];
      sd[i] = sqrt(a.m2[i * a.N + n] * rn);
    }
    sed_package<MODEL>(c);
    sed_package<MODEL>(sd);
    if (a.random_restart) {                               // MCMC.py:366
#pragma unroll
      for (int i = 0; i < P; i += 2) {
        double za, zb;
        philox_normal2(a.seed, RNG_START, n, (sbase + i) >> 1, bmt, za, zb);
        c[i] += za * sd[i];
        if (i + 1 < P) c[i + 1] += zb * sd[i + 1];
      }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) sg[i] = a.fac * sd[i];    // MCMC.py:385
    sed_package<MODEL>(sg);
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      c[i] = a.center_per_chain ? a.center[i * a.N + n] : a.center[i];
      sg[i] = a.sigma_per_chain ? a.sigma_in[i * a.N + n] : a.sigma_in[i];
    }
  }

  double th[P];
  double prior;
  int flag = 0;
  if (a.randomize) {                                      // MCMC.py:151-159
    prior = -1.0;
    int tries = 0;
    while (!(prior > 0.0)) {
      ++tries;
      if (tries > kStartMaxTries) { flag = kFlagStartFailed; break; }
      const uint64_t base = sbase + static_cast<uint64_t>(tries) * PE;
#pragma unroll
      for (int i = 0; i < P; i += 2) {
        double za, zb;
        philox_normal2(a.seed, RNG_START, n, (base + i) >> 1, bmt, za, zb);
        th[i] = c[i] + za * sg[i];
        if (i + 1 < P) th[i + 1] = c[i + 1] + zb * sg[i + 1];
      }
      sed_package<MODEL>(th);
      prior = sed_prior<MODEL>(th, a.k);
    }
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) th[i] = c[i];
    sed_package<MODEL>(th);
    prior = sed_prior<MODEL>(th, a.k);
    if (!(prior > 0.0)) flag = kFlagStartFailed;          // ZeroPriorError, MCMC.py:174-175
  }
  double lp = __builtin_nan("");
  if (!flag) {
    lp = sed_lnlike_auto<MODEL>(th, prior, B, a.k, nullptr);
    if (prior != 1.0) lp += log(prior);                   // MCMC.py:177
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    a.theta[i * a.N + n] = th[i];
    a.sigma[i * a.N + n] = sg[i];
    a.mean[i * a.N + n] = 0.0;
    a.m2[i * a.N + n] = 0.0;
  }
  a.lnPr[n] = lp;
  a.naccept[n] = 1;                                       // MCMC.py:179
  for (int b = 0; b < a.nblock; ++b) a.accept[b * a.N + n] = 1;   // :181
  a.flags[n] = flag;
}

// ---- fused propose / evaluate / accept, nsteps per launch (MCMC.py:225-286) ------------------
// FAST: the clamp-free transcendental kernels (catalogue checked at load, SedConsts::fast_ok);
// otherwise the exact OCML path - a separate instantiation so that it does not cost the fast
// one registers.
#ifndef MI355_MH_MIN_WAVES
#define MI355_MH_MIN_WAVES 2
#endif
#ifndef MI355_ABL
#define MI355_ABL 0       // timing ablations of tools/ablate.sh (never set in the product build)
#endif
// EXT: the same sampler with everything the step-wise engine (gmh_kernels.hpp) adds - Cholesky-
// correlated proposals (Proposal.py:150-154: offset = (L z) sigma over the block, P normals per
// step), rotated proposals (:145-148: offset = sum over the block's members j of L[:, j] z_j, every
// parameter moves) and the running co-moments newMCMC's covariance needs (MCMC.py:310-340) - so
// that correlated stages run at the fused kernel's rate (a step of the step-wise engine is three
// launches).  The chain's lower-triangular L sits in the lane's column of an LDS array
// [P(P+1)/2][blockDim] behind the band tables; the co-moments stay in HBM (read-modify-write every
// step: only adaptation stages track them).  Same draws, same arithmetic as k_gmh_propose /
// k_gmh_accept: tests/test_gmh_gpu.py holds the two to identical chains.  A separate instantiation,
// so the uncorrelated production kernel keeps its registers.
// G: lanes per chain.  1: lane = chain (the lane loops over the bands) - the layout that fills the
// machine once there are chains enough for it.  8 / 32: lanes = (chain, band), G lanes hold copies of
// one chain and each evaluates one band (sed_lnlike_fast_cb; lane 0 of the group writes the chain's
// state) - north_star's primary mapping, which the host picks for jobs too small to give every SIMD
// a wave (sed_plan in api.hip), for the plain and the extended sampler alike.  Same draws, same
// arithmetic in the same order: the chains are identical bit for bit under either mapping
// (tests/test_sed_gpu.py, tests/test_gmh_gpu.py).
// SINGLE: one parameter per block with block b = parameter b (the production configuration,
// driver.py:329-332; the launcher checks it) and no sqrt matrix.  The block index is launch-uniform,
// so "which parameter moves" is a scalar branch: the step changes th[blk] in place, evaluates th, and
// a rejected lane puts the one old value back - instead of a masked update of all P parameters into a
// copy and a P-wide select at the accept.  Same arithmetic on the moved parameter (th + z sigma),
// bit-identical chains (tests/test_gmh_gpu.py::test_sed_in_place_instantiation_equals_stepwise_engine and
// the mixed-catalogue / lane-mapping tests, which compare across the instantiations).
// Waves per SIMD the register allocation aims at.  Two (256 VGPRs, no spills) everywhere except in the
// plain lane = chain instantiations on the fast path (in place or not) of the models measured faster with three (168 VGPRs; 80 of them
// spill to scratch in the two-component kernel, and the third wave still wins: 29.7 -> 27.1 ms per
// 200-step launch of C2 - the kernel waits on its own dependency chains, which a third wave fills).
// Measured and left at two: M1N / M1N_LOGA (no change), the extended sampler (50 -> 63 ms).  THICK lost at
// first (9.9 -> 13.4 ms per 200 steps of 20 000 x 64 chains) and gains since the file is compiled without
// machine LICM (10.1 -> 8.9 ms).  Without parameter blocks (six normals a step):
// 11.6 -> 11.2 ms per 200 steps.  SEDLikelihood2 on its fast path (round 3): 8.15 -> 7.76 ms per 200 steps
// of 20 000 x 64 chains.  tools/ab_bench.sh, tools/time_models.py, tools/time_noblock.py.
template <int MODEL, bool FAST, bool EXT, int G>
constexpr int mh_min_waves() {
  return (FAST && !EXT && G == 1 &&
          (MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA || MODEL == MI355_SED_M1 ||
           MODEL == MI355_SED_M2 || MODEL == MI355_SED_THICK || MODEL == MI355_SED_THICK_LOGA))
             ? 3 : MI355_MH_MIN_WAVES;
}
// XM = 1: the extended sampler compiled for ONE proposal - Cholesky-correlated with free-running draws,
// lane = chain, blocks of kBlockThreads lanes (what every stage of sampler() after the first runs unless
// noCorrelations, MCMC.py:389-403; the ESS leg of bench.py).  No run-time proposal mode, rank tables or
// injection in the step loop, the uniform pair serves two steps (the general instantiation draws a
// Philox block per step for it), L z accumulated column by column as the normals arrive: 1137 VALU
// instructions per wave-step against 1390, 42.2 against 51.4 ms per 200 steps of C2.  Same draws, same
// arithmetic as the general instantiation and the step-wise engine (tests/test_gmh_gpu.py compares
// them bit for bit).  Measured on top of it and not kept (tools/ab_ext.sh, round 3): registers for three
// waves per SIMD with the diagonal of L in registers so that three blocks fit a CU's LDS (41.7 - 42.7 ms
// with 36 - 85 spilled VGPRs: at two waves VALUBusy is 92 % already, the kernel is bound by its
// instruction count); one block of 768 lanes per CU (48 ms, average occupancy 2.2 waves per SIMD).
template <int MODEL, bool FAST, bool EXT = false, int G = 1, bool SINGLE = false, int XM = -1>
__global__ __launch_bounds__(kBlockThreads, (mh_min_waves<MODEL, FAST, EXT, G>())) void k_mh_run(MhArgs a) {
  static_assert(G == 1 || MODEL != MI355_SED_M2, "(chain, band) lanes: not for SEDLikelihood2");
  static_assert(!SINGLE || !EXT, "single-parameter blocks: plain sampler");
  static_assert(XM == -1 || ((XM == 1 || XM == 2) && EXT && G == 1), "XM: the Cholesky-correlated extended sampler, lane = chain");
  // XM = 2: XM = 1 with the running co-moments in registers for the launch (loaded before the first
  // step, stored after the last) instead of a read-modify-write of the 21 entries in memory every step:
  // the same additions in the same order on the same values.  Its own instantiation because the 42
  // registers cost the launches without co-moments what the compiler spills for them.
  constexpr bool CMR = (XM == 2);
  constexpr int P = ModelTraits<MODEL>::P;
  constexpr int NTRI = P * (P + 1) / 2;
  extern __shared__ double smem[];
  const int stride = kBandFields * a.nbm;
  // a mixed catalogue runs as two launches over disjoint block lists (LaunchRun): the blocks whose
  // sources all take the fast path, and the rest
  const int64_t bidx = a.blocklist ? a.blocklist[blockIdx.x] : blockIdx.x;
  const int cpb = blockDim.x / G;                          // chains per block
  const int64_t n0 = bidx * cpb;
  const int64_t n_last = min(n0 + cpb, a.N) - 1;
  const int64_t s_first = n0 / a.C;
  const int nsrc = static_cast<int>(n_last / a.C - s_first) + 1;
  const double* e2t = stage_exp2_table();
  const double* bmt = bm_of(e2t);                        // Box-Muller rows of the same LDS table
  stage_bands(smem, a.bands, s_first, nsrc, stride);
  const int64_t n = n0 + threadIdx.x / G;
  const int jb = threadIdx.x % G;                          // the lane's band (G > 1)
  const bool writer = G == 1 || jb == 0;                   // the lane that stores the chain's state
  const bool live = n < a.N;
  const int64_t s = live ? n / a.C : s_first;
  // band count of the lane's source and its maximum over the wave (the lanes of a wave share the
  // source whenever C is a multiple of 64): the band loop of the fast path is scalar control flow
  const int32_t nb_raw = live ? a.nb[s] : 0;
  const int nb_lane = nb_count(nb_raw);
  const int nb_wave = wave_max_uniform(nb_lane);
  if (!live) return;
#if MI355_ABL == 3
  Bands B{smem + (s - s_first) * stride, a.nbm, 0, 0, e2t, FAST};
#else
  // FAST: every source of the block is in the fast domain (the launcher saw to it); otherwise the
  // lane follows its own source, as sed_lnlike_auto does in every other kernel
  Bands B{smem + (s - s_first) * stride, a.nbm, nb_lane, FAST ? nb_wave : nb_lane, e2t,
          FAST || nb_fast(nb_raw, a.k)};
#endif

  double th[P], sg[P], mu[P], m2[P];
  uint32_t accb[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    th[i] = a.theta[i * a.N + n];
    sg[i] = a.sigma[i * a.N + n];
    mu[i] = a.mean[i * a.N + n];
    m2[i] = a.m2[i * a.N + n];
    accb[i] = (i < a.nblock) ? a.accept[i * a.N + n] : 0u;
  }
  double lnPr = a.lnPr[n];
  uint32_t nacc = a.naccept[n];
  // EXT: this chain's sqrt matrix, lower triangle row by row, in the lane's LDS column
  double* Ll = smem + static_cast<size_t>(nsrc) * stride + threadIdx.x;
  // G > 1: the lane's 16-byte slot of the band-term exchange, behind the band tables
  // (behind the sqrt matrices in the extended instantiation)
  double2* cb_slot = reinterpret_cast<double2*>(
                         smem + ((static_cast<size_t>(nsrc) * stride + (EXT ? static_cast<size_t>(NTRI) * blockDim.x : 0) + 1) &
                                 ~static_cast<size_t>(1))) + threadIdx.x;
  // G > 1: the G lanes of a chain draw ahead.  The streams are counter-based, so lane jb computes pair
  // number base + jb of the normal (uniform) stream, parks it in its 16-byte slot behind the exchange
  // slots, and the chain reads the next G pairs with one broadcast each - Philox and Box-Muller, a
  // third of a step's dependent instructions when every lane repeats them, run once per G pairs.
  // Draws are consumed in stream order and the index is wave-uniform, so a refill is a scalar branch;
  // the group is part of one wave, whose LDS operations execute in order (fences as in cb_publish).
  double2* const nrm_slot = cb_slot + blockDim.x;          // (za, zb) of the lane's pair
  double2* const uni_slot = nrm_slot + blockDim.x;         // (ua, ub)
  uint64_t nrm_base = 0, uni_base = 0;
  bool nrm_have = false, uni_have = false;
  auto normal_at = [&](uint64_t idx) -> double {
    const uint64_t pr = idx >> 1;
    if (!nrm_have || pr - nrm_base >= static_cast<uint64_t>(G)) {
      nrm_have = true;
      nrm_base = pr;
      double pa, pb;
      philox_normal2(a.seed, RNG_NORMAL, n, pr + jb, bmt, pa, pb);
      cb_publish(nrm_slot, pa, pb);
    }
    const double2 v = (nrm_slot - jb)[pr - nrm_base];
    return (idx & 1u) ? v.y : v.x;
  };
  auto uniform_at = [&](uint64_t idx) -> double {
    const uint64_t pr = idx >> 1;
    if (!uni_have || pr - uni_base >= static_cast<uint64_t>(G)) {
      uni_have = true;
      uni_base = pr;
      double qa, qb;
      philox_uniform2(a.seed, RNG_UNIFORM, n, pr + jb, qa, qb);
      cb_publish(uni_slot, qa, qb);
    }
    const double2 v = (uni_slot - jb)[pr - uni_base];
    return (idx & 1u) ? v.y : v.x;
  };
  if constexpr (EXT) {
    if (a.mode != 0) {
#pragma unroll
      for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
          Ll[(i * (i + 1) / 2 + j) * (XM >= 1 ? kBlockThreads : static_cast<int>(blockDim.x))] =
              a.L[(static_cast<int64_t>(i) * P + j) * a.N + n];
    }
  }

  double cmr[CMR ? NTRI : 1];                              // running co-moments (XM = 2)
  if constexpr (CMR) {
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) cmr[i * (i + 1) / 2 + j] = a.comom[(static_cast<int64_t>(i) * P + j) * a.N + n];
  }
  const bool inject = a.inj_n != nullptr;
  const uint32_t allmask = (1u << P) - 1u;
  int blk = a.nblock ? static_cast<int>(a.stage_step % a.nblock) : 0;
  int thin_ctr = 0;
  int64_t row = 0;
  double za = 0.0, zb = 0.0, ua = 0.0, ub = 0.0;   // cached Philox pairs (npn == 1)
  bool state_ok = SINGLE && sed_prior_is_box<MODEL>() && sed_prior<MODEL>(th, a.k) > 0.0;   // current state inside the box
  double nsamp = static_cast<double>(a.nsamp0);    // samples in the moments so far (a 64-bit integer costs
                                                   // four instructions per step to convert)

  for (int64_t t = 0; t < a.nsteps; ++t) {
    const uint32_t mask = SINGLE ? 0u : (a.nblock ? a.blockmask[blk] : allmask);
    const uint64_t step = static_cast<uint64_t>(a.step_total + t);
    double nwc[SINGLE ? 1 : P];                            // the proposed point (a copy) ...
    double* const nw = SINGLE ? th : nwc;                  // ... or the state itself, moved in place
    double old = 0.0;                                      // SINGLE: th[blk] before the step
    bool swapped = false;                                  // SINGLE: package() exchanged the components
    bool moved_ok = true;                                  // SINGLE, box prior: the moved parameter is inside its interval
    if constexpr (!SINGLE) {
#pragma unroll
      for (int i = 0; i < P; ++i) nwc[i] = th[i];
    }
    double u;
    if constexpr (SINGLE) {
      double z;
      if (inject) {
        z = a.inj_n[(t * a.N + n) * P + blk];
        u = a.inj_u[t * a.N + n];
      } else if constexpr (G > 1) {
        z = normal_at(step);
        u = uniform_at(step);
      } else {
        if (t == 0 || (step & 1u) == 0) {
          philox_normal2(a.seed, RNG_NORMAL, n, step >> 1, bmt, za, zb);
          philox_uniform2(a.seed, RNG_UNIFORM, n, step >> 1, ua, ub);
        }
        z = (step & 1u) ? zb : za;
        u = (step & 1u) ? ub : ua;
      }
#pragma unroll
      for (int i = 0; i < P; ++i)
        if (blk == i) {                                    // scalar branch: blk is launch-uniform
          asm volatile("");                                // (keeps the optimizer from flattening it into P selects)
          old = th[i];
          th[i] = old + z * sg[i];                         // Proposal.py:151,156-159
          if constexpr (sed_prior_is_box<MODEL>()) moved_ok = sed_prior_one<MODEL>(i, th[i], a.k);
        }
    } else if constexpr (EXT && XM >= 1) {
      // Cholesky-correlated, free-running: normal(step NPN + j) for every j (Proposal.py:150-154), the
      // step's uniform from the pair it shares with its neighbour
      constexpr int NPN = (P + 1) & ~1;
      if (t == 0 || (step & 1u) == 0) philox_uniform2(a.seed, RNG_UNIFORM, n, step >> 1, ua, ub);
      u = (step & 1u) ? ub : ua;
      // L z column by column, a pair of normals at a time (each row's sum still runs over j ascending
      // from 0.0, as k_gmh_propose's): a pair is dead once its two columns are in
      double off[P];
#pragma unroll
      for (int i = 0; i < P; ++i) off[i] = 0.0;
#pragma unroll
      for (int j = 0; j < P; j += 2) {
        double zz[2];
        philox_normal2(a.seed, RNG_NORMAL, n, step * (NPN / 2) + (j >> 1), bmt, zz[0], zz[1]);
#pragma unroll
        for (int q = 0; q < 2 && j + q < P; ++q)
#pragma unroll
          for (int i = j + q; i < P; ++i) off[i] += Ll[(i * (i + 1) / 2 + j + q) * kBlockThreads] * zz[q];
      }
#pragma unroll
      for (int i = 0; i < P; ++i)
        if ((mask >> i) & 1u) nw[i] += off[i] * sg[i];     // Proposal.py:153-159
    } else if (EXT && a.mode != 0) {
      // z_j of the step: correlated - normal(step npn + j) for every j; rotated - normal(step npn +
      // rank_j) for the members j of the block (injected rows: element j / element rank_j)
      double z[P];
      if (inject) {
        const double* zn = a.inj_n + (t * a.N + n) * P;
#pragma unroll
        for (int j = 0; j < P; ++j) z[j] = zn[a.mode == 1 ? j : a.rank_in_block[j]];
        u = a.inj_u[t * a.N + n];
      } else {
        if (a.mode == 1) {                                 // npn is even: whole Philox pairs
#pragma unroll
          for (int j = 0; j < P; j += 2) {
            double p0, p1;
            if constexpr (G > 1) {
              p0 = normal_at(step * a.npn + j);
              p1 = normal_at(step * a.npn + j + 1);
            } else {
              philox_normal2(a.seed, RNG_NORMAL, n, (step * a.npn + j) >> 1, bmt, p0, p1);
            }
            z[j] = p0;
            if (j + 1 < P) z[j + 1] = p1;
          }
        } else {
#pragma unroll
          for (int j = 0; j < P; ++j) {
            const uint64_t idx = step * a.npn + a.rank_in_block[j];
            z[j] = ((mask >> j) & 1u) ? (G > 1 ? normal_at(idx) : philox_normal(a.seed, RNG_NORMAL, n, idx, bmt)) : 0.0;
          }
        }
        u = G > 1 ? uniform_at(step) : philox_uniform(a.seed, RNG_UNIFORM, n, step);
      }
#pragma unroll
      for (int i = 0; i < P; ++i) {
        double sum = 0.0;                                  // row i of L times z, j ascending
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          const double lij = Ll[(i * (i + 1) / 2 + j) * blockDim.x];
          if (a.mode == 1 || ((mask >> j) & 1u)) sum += lij * z[j];
        }
        if (a.mode == 2 || ((mask >> i) & 1u)) nw[i] += sum * sg[i];   // Proposal.py:145-159
      }
    } else if (inject) {
      const double* zn = a.inj_n + (t * a.N + n) * P;
#pragma unroll
      for (int i = 0; i < P; ++i)
        if ((mask >> i) & 1u) nw[i] += zn[i] * sg[i];      // Proposal.py:151,156-159
      u = a.inj_u[t * a.N + n];
    } else {
      if (a.npn == 1) {
        double z;
        if constexpr (G > 1) {
          z = normal_at(step);
          u = uniform_at(step);
        } else {
#if MI355_ABL == 2
          za = 0.3; zb = -0.2; ua = 0.4; ub = 0.6;
#else
          if (t == 0 || (step & 1u) == 0) {
            philox_normal2(a.seed, RNG_NORMAL, n, step >> 1, bmt, za, zb);
            philox_uniform2(a.seed, RNG_UNIFORM, n, step >> 1, ua, ub);
          }
#endif
          z = (step & 1u) ? zb : za;
          u = (step & 1u) ? ub : ua;
        }
#pragma unroll
        for (int i = 0; i < P; ++i)
          if ((mask >> i) & 1u) nw[i] += z * sg[i];
      } else {
#pragma unroll
        for (int i = 0; i < P; ++i)
          if ((mask >> i) & 1u) {
            const uint64_t idx = step * a.npn + a.rank_in_block[i];
            nw[i] += (G > 1 ? normal_at(idx) : philox_normal(a.seed, RNG_NORMAL, n, idx, bmt)) * sg[i];
          }
        u = G > 1 ? uniform_at(step) : philox_uniform(a.seed, RNG_UNIFORM, n, step);
      }
    }
    if constexpr (SINGLE) swapped = sed_package_flag<MODEL>(nw);
    else sed_package<MODEL>(nw);                           // Proposal.py:164
    // The reference evaluates the likelihood before the prior (MCMC.py:236-245); the outcome
    // (reject, keep prev) is the same with the prior tested first, which skips the evaluation.
    // SINGLE, box priors: while every lane's current state is inside the box (checked at launch entry,
    // kept by every accepted step), the proposal is inside iff its one moved parameter is (sed_prior_one:
    // two comparisons behind the scalar branch above instead of the ten of the full prior)
    double prior;
    if (SINGLE && sed_prior_is_box<MODEL>() && __builtin_amdgcn_ballot_w64(!state_ok) == 0)
      prior = moved_ok ? 1.0 : 0.0;
    else
      prior = sed_prior<MODEL>(nw, a.k);
    bool accepted = false;
    double next = lnPr;
    // A non-finite proposal inside the prior box (A = inf / NaN: the box has no upper bound on A
    // and NaN passes every comparison) comes out of the fast kernels as NaN or -inf and is
    // rejected below like any bad likelihood, so the fast instantiation needs no finiteness test.
    if (prior > 0.0 && (FAST || all_finite<P>(nw))) {
      double v;
      if constexpr (G > 1 && FAST) v = sed_lnlike_fast_cb<MODEL, G>(nw, B, a.k, jb, cb_slot);
      else if constexpr (G > 1) v = B.fast ? sed_lnlike_fast_cb<MODEL, G>(nw, B, a.k, jb, cb_slot)
                                           : sed_lnlike_exact_cb<MODEL, G>(nw, B, a.k, jb, cb_slot);
      else v = (FAST || B.fast) ? sed_lnlike_fast<MODEL>(nw, B, a.k, nullptr)
                                : sed_lnlike<MODEL>(nw, B, a.k, nullptr);
      const bool bad = (v != v) || (v == -__builtin_inf());   // MCMC.py:238-243
      if (!bad) {
        if (prior != 1.0) v += log(prior);                 // :250
        const double lnalpha = v - lnPr;                   // :251 (+ lndensityRatio = 0, :252)
#if MI355_ABL == 4
        const bool reject = !(lnalpha >= 0.0) && !(1.0 + lnalpha > u);
#else
        const bool reject = !(lnalpha >= 0.0) && !(exp_accept(lnalpha, e2t) > u);   // :264
#endif
        if (!reject) {
          accepted = true;
          next = v;
        }
      }
    }
    if constexpr (SINGLE) {
      if (accepted) {
        lnPr = next;
        ++nacc;                                            // :269
        state_ok = true;                                   // an accepted point has prior > 0
      }
      if (__builtin_amdgcn_ballot_w64(swapped && !accepted) != 0) {   // rare: a wave-level branch
        asm volatile("");
        if (swapped && !accepted) sed_package_swap<MODEL>(th);        // back to the order the old value belongs to
      }
#pragma unroll
      for (int i = 0; i < P; ++i)
        if (blk == i) {
          asm volatile("");
          th[i] = accepted ? th[i] : old;
          accb[i] += accepted ? 1u : 0u;                   // :271
        }
    } else if (accepted) {
#pragma unroll
      for (int i = 0; i < P; ++i) th[i] = nw[i];
      lnPr = next;
      ++nacc;                                              // :269
#pragma unroll
      for (int b = 0; b < P; ++b) accb[b] += (b == blk && a.nblock) ? 1u : 0u;   // :271
    }
    if (a.nblock) blk = (blk + 1 == a.nblock) ? 0 : blk + 1;   // :278-279

    // running ddof=0 moments over every appended sample (mean/stdev with burn=1, :290-307)
#if MI355_ABL != 1
    nsamp += 1.0;                                          // exact: a count below 2^53
    const double rn = rcp_nr(nsamp);                       // 1/n to 2^-50
    double dl[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const double dlt = th[i] - mu[i];
      dl[i] = dlt;
      mu[i] = fma(dlt, rn, mu[i]);
      m2[i] = fma(dlt, th[i] - mu[i], m2[i]);
    }
    if constexpr (CMR) {
      const double cf = 1.0 - rn;                          // C_ij += (n-1)/n d_i d_j, as k_gmh_accept
#pragma unroll
      for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) cmr[i * (i + 1) / 2 + j] = cmr[i * (i + 1) / 2 + j] + cf * dl[i] * dl[j];
    } else if constexpr (EXT) {
      if (a.comom != nullptr && writer) {                  // C_ij += (n-1)/n d_i d_j, as k_gmh_accept
        const double cf = 1.0 - rn;
        if (a.N * NTRI <= kComomAtomicMaxEntries) {            // few chains: adds that return nothing (gmh_comom_update)
#pragma unroll
          for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
              add_noret(a.comom + (static_cast<int64_t>(i) * P + j) * a.N + n, cf * dl[i] * dl[j]);
        } else {
          double cur[NTRI];
#pragma unroll
          for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
              cur[i * (i + 1) / 2 + j] = a.comom[(static_cast<int64_t>(i) * P + j) * a.N + n];
#pragma unroll
          for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
              a.comom[(static_cast<int64_t>(i) * P + j) * a.N + n] =
                  cur[i * (i + 1) / 2 + j] + cf * dl[i] * dl[j];
        }
      }
    }
#endif
    if (a.thin > 0) {
      if (++thin_ctr == a.thin) {
        thin_ctr = 0;
        if (writer) {
#pragma unroll
          for (int i = 0; i < P; ++i) a.tr_theta[(row * P + i) * a.N + n] = th[i];
          a.tr_lnPr[row * a.N + n] = lnPr;
          a.tr_acc[row * a.N + n] = accepted ? 1 : 0;
        }
        ++row;
      }
    }
  }

  if (!writer) return;
  if constexpr (CMR) {
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) a.comom[(static_cast<int64_t>(i) * P + j) * a.N + n] = cmr[i * (i + 1) / 2 + j];
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    a.theta[i * a.N + n] = th[i];
    a.mean[i * a.N + n] = mu[i];
    a.m2[i * a.N + n] = m2[i];
    if (i < a.nblock) a.accept[i * a.N + n] = accb[i];
  }
  a.lnPr[n] = lnPr;
  a.naccept[n] = nacc;
}

// ---- incremental variant of k_mh_run for the two-component normalised models --------------------
// In single-parameter block mode (the production configuration: doBlock with one parameter per
// block, driver.py:329-332, MCMC.py:230-234) a step moves ONE of (A1, beta1, T1, A2, beta2, T2).
// The band fluxes of a component depend on its own (beta, T) only and enter the model linearly
// through A, so a chain keeps the unit band terms u_c[j] of both components of its CURRENT state
// and a step evaluates only what its parameter touches:
//   A_c         : nothing but the chi2 sum (no transcendental at all),
//   beta_c, T_c : the unit terms of component c (2 nb exponentials instead of 4 nb),
// then chi2 = sum_j w_j (d_j - A_1 u_1j - A_2 u_2j)^2 as always.  The value returned for the
// proposed point is the same lnLike(theta') as a full evaluation up to rounding (the terms are
// the same functions of theta'; only the association of the products differs), so the chain is
// the one k_mh_run produces - tests/test_sed_gpu.py holds it to the same 1e-12 against the oracle
// and the reference's traces.
// The cache lives in LDS, [2 slots][kIncBands][blockDim] doubles in front of the band tables
// (lane-consecutive, conflict-free); a per-lane bit says which slot holds component 0, so that
// package()'s exchange of the components costs one XOR.  The block index is launch-uniform, so
// which parameter moves is a scalar branch.  Requirements (checked by the host, which otherwise
// launches k_mh_run): fast_ok, every nb[s] <= kIncBands, nblock == P with block b = parameter b.
// No counterpart in the reference, which rebuilds model(*params)(data) at every step
// (Likelihood.py:70-75); opt-in through mi355_mh_set_incremental.
template <int MODEL>
__global__ __launch_bounds__(kBlockThreads, 3) void k_mh_run_inc(MhArgs a) {
  static_assert(MODEL == MI355_SED_M2N || MODEL == MI355_SED_M2N_LOGA, "two-component models");
  constexpr int P = 6;
  constexpr bool LG = MODEL == MI355_SED_M2N_LOGA;
  extern __shared__ double smem[];
  const int slot = kIncBands * blockDim.x;                 // doubles per cache slot
  double* tables = smem + 2 * slot;
  const int stride = kBandFields * a.nbm;
  // (a mixed catalogue hands this kernel the list of chain blocks whose sources are all fast)
  const int64_t bidx = a.blocklist ? a.blocklist[blockIdx.x] : blockIdx.x;
  const int64_t n0 = bidx * blockDim.x;
  const int64_t n_last = min(n0 + blockDim.x, a.N) - 1;
  const int64_t s_first = n0 / a.C;
  const int nsrc = static_cast<int>(n_last / a.C - s_first) + 1;
  const double* e2t = stage_exp2_table();
  const double* bmt = bm_of(e2t);                        // Box-Muller rows of the same LDS table
  stage_bands(tables, a.bands, s_first, nsrc, stride);
  const int64_t n = n0 + threadIdx.x;
  if (n >= a.N) return;
  const int64_t s = n / a.C;
  const Bands B = make_bands(tables + (s - s_first) * stride, a.nbm, a.nb[s], e2t, a.k);

  double th[P], mu[P], m2[P];
  uint32_t accb[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    th[i] = a.theta[i * a.N + n];
    mu[i] = a.mean[i * a.N + n];
    m2[i] = a.m2[i * a.N + n];
    accb[i] = a.accept[i * a.N + n];
  }
  double lnPr = a.lnPr[n];
  uint32_t nacc = a.naccept[n];

  // the cache of the current state (rebuilt at every launch entry): amplitudes in registers,
  // unit terms in this lane's column of the two LDS slots
  double Ac[2] = {LG ? exp10_fast(th[0]) : th[0], LG ? exp10_fast(th[3]) : th[3]};
  double* mine = smem + threadIdx.x;
  int flip = 0;                                            // slot of component 0
  {
    double un[kIncBands];
    sed_unit_terms(3.0 + th[1], th[2], B, a.k, un);
#pragma unroll
    for (int j = 0; j < kIncBands; ++j) mine[j * blockDim.x] = un[j];
    sed_unit_terms(3.0 + th[4], th[5], B, a.k, un);
#pragma unroll
    for (int j = 0; j < kIncBands; ++j) mine[slot + j * blockDim.x] = un[j];
  }

  const bool inject = a.inj_n != nullptr;
  int blk = static_cast<int>(a.stage_step % P);
  int thin_ctr = 0;
  int64_t row = 0;
  double za = 0.0, zb = 0.0, ua = 0.0, ub = 0.0;

  for (int64_t t = 0; t < a.nsteps; ++t) {
    const uint64_t step = static_cast<uint64_t>(a.step_total + t);
    const int comp = blk >= 3 ? 1 : 0, kind = blk - 3 * comp;   // 0 amplitude, 1 beta, 2 T
    double z, u;
    if (inject) {
      z = a.inj_n[(t * a.N + n) * P + blk];                  // Proposal.py:151,156-159
      u = a.inj_u[t * a.N + n];
    } else {
      if (t == 0 || (step & 1u) == 0) {
        philox_normal2(a.seed, RNG_NORMAL, n, step >> 1, bmt, za, zb);
        philox_uniform2(a.seed, RNG_UNIFORM, n, step >> 1, ua, ub);
      }
      z = (step & 1u) ? zb : za;
      u = (step & 1u) ? ub : ua;
    }
    const double moved = a.sigma[blk * a.N + n] * z;       // the one width this step needs
    double nw[P];
#pragma unroll
    for (int i = 0; i < P; ++i) nw[i] = (i == blk) ? th[i] + moved : th[i];
    // the moved component as proposed, before package() may exchange the two
    const double pa = comp ? nw[3] : nw[0], pb = comp ? nw[4] : nw[1], pT = comp ? nw[5] : nw[2];
    const bool swapped = nw[2] > nw[5];
    sed_package<MODEL>(nw);                                // Proposal.py:164
    const double prior = sed_prior<MODEL>(nw, a.k);
    bool accepted = false;
    double next = lnPr;
    double un[kIncBands];
    double An = comp ? Ac[1] : Ac[0];
    const double Ao = comp ? Ac[0] : Ac[1];
    double* ux = mine + ((comp ^ flip) ? slot : 0);        // cached terms of the moved component
    const double* uo = mine + ((comp ^ flip) ? 0 : slot);  // ... and of the other one
    if (prior > 0.0) {
      const int bd = static_cast<int>(blockDim.x);
      double v;
      if (kind == 0) {                                     // amplitude step: no transcendental
        An = LG ? exp10_fast(pa) : pa;
        v = -0.5 * sed_chi2_two(B, An, ux, bd, Ao, uo, bd);
      } else {
        sed_unit_terms(3.0 + pb, pT, B, a.k, un);
        v = -0.5 * sed_chi2_two(B, An, un, 1, Ao, uo, bd);
      }
      const bool bad = (v != v) || (v == -__builtin_inf());   // MCMC.py:238-243
      if (!bad) {
        const double lnalpha = v - lnPr;                   // :251
        const bool reject = !(lnalpha >= 0.0) && !(exp_accept(lnalpha, e2t) > u);   // :264
        if (!reject) {
          accepted = true;
          next = v;
        }
      }
    }
    if (accepted) {
#pragma unroll
      for (int i = 0; i < P; ++i) th[i] = nw[i];
      lnPr = next;
      ++nacc;                                              // :269
#pragma unroll
      for (int b = 0; b < P; ++b) accb[b] += (b == blk) ? 1u : 0u;   // :271
      if (kind == 0) {
        if (comp) Ac[1] = An; else Ac[0] = An;
      } else {
#pragma unroll
        for (int j = 0; j < kIncBands; ++j) ux[j * blockDim.x] = un[j];
      }
      if (swapped) {                                       // the cache follows package()
        const double ta = Ac[0]; Ac[0] = Ac[1]; Ac[1] = ta;
        flip ^= 1;
      }
    }
    blk = (blk + 1 == P) ? 0 : blk + 1;                    // :278-279

    const double rn = rcp_nr(static_cast<double>(a.nsamp0 + t + 1));
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const double dlt = th[i] - mu[i];
      mu[i] = fma(dlt, rn, mu[i]);
      m2[i] = fma(dlt, th[i] - mu[i], m2[i]);
    }
    if (a.thin > 0) {
      if (++thin_ctr == a.thin) {
        thin_ctr = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) a.tr_theta[(row * P + i) * a.N + n] = th[i];
        a.tr_lnPr[row * a.N + n] = lnPr;
        a.tr_acc[row * a.N + n] = accepted ? 1 : 0;
        ++row;
      }
    }
  }

#pragma unroll
  for (int i = 0; i < P; ++i) {
    a.theta[i * a.N + n] = th[i];
    a.mean[i * a.N + n] = mu[i];
    a.m2[i * a.N + n] = m2[i];
    a.accept[i * a.N + n] = accb[i];
  }
  a.lnPr[n] = lnPr;
  a.naccept[n] = nacc;
}

// ---- RNG diagnostic -----------------------------------------------------------------------
__global__ void k_rng_dump(uint64_t seed, int64_t step0, int64_t nsteps, int64_t N, int npn,
                           double* normals, double* uniforms) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nsteps * N) return;
  const int64_t t = i / N, n = i % N;
  const uint64_t step = static_cast<uint64_t>(step0 + t);
  for (int r = 0; r < npn; ++r)
    normals[i * npn + r] = philox_normal(seed, RNG_NORMAL, n, step * npn + r);
  uniforms[i] = philox_uniform(seed, RNG_UNIFORM, n, step);
}

__global__ void k_math_probe(int fn, int64_t n, const double* a, const double* b, const double* c,
                             double* out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const double* tab = stage_exp2_table();
  __syncthreads();
  if (i >= n) return;
  double v;
  switch (fn) {
    case 0: v = exp_fast(a[i]); break;
    case 1: v = expm1_fast(a[i]); break;
    case 2: v = pow_prelog(a[i], b[i], c[i]); break;
    case 3: v = div_fast(a[i], b[i]); break;
    case 5: v = log_fast(a[i]); break;
    case 6: { double sn, cs; sincos2pi_fast(a[i], &sn, &cs); v = sn; break; }
    case 7: { double sn, cs; sincos2pi_fast(a[i], &sn, &cs); v = cs; break; }
    case 8: v = pow2_prelog(a[i], b[i], c[i]); break;
    case 9: v = exp2_tab(a[i], tab); break;
    case 10: v = exp2_fast(a[i]); break;
    case 11: v = sqrt_fast(a[i]); break;
    case 12: v = u53(static_cast<uint32_t>(a[i]), static_cast<uint32_t>(b[i])); break;
    case 13: v = log_bm(a[i], bm_of(tab)); break;
    case 14: { double sn, cs; sincos_bm(a[i], bm_of(tab), &sn, &cs); v = sn; break; }
    case 15: { double sn, cs; sincos_bm(a[i], kBmTable, &sn, &cs); v = cs; break; }
    default: v = exp10_fast(a[i]); break;
  }
  out[i] = v;
}

// ---- GaussianData.quadform (GaussianData.py:31-53): one 256-thread block per row ------------
__global__ __launch_bounds__(kBlockThreads) void k_gauss_quadform(
    int64_t n, const double* __restrict__ A, const double* __restrict__ B, int64_t b_stride,
    const double* __restrict__ sig2, int sig_is_scalar, double* __restrict__ out) {
  __shared__ double part[kBlockThreads / 64];
  const double* a = A + static_cast<int64_t>(blockIdx.x) * n;
  const double* b = B ? B + static_cast<int64_t>(blockIdx.x) * b_stride : a;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
    acc = fma(a[i] / sig2[sig_is_scalar ? 0 : i], b[i], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kBlockThreads / 64; ++w) t += part[w];
    out[blockIdx.x] = t;
  }
}

// ---- layout helpers -----------------------------------------------------------------------
// in [rows][cols] -> out [cols][rows]
__global__ void k_transpose(const double* __restrict__ in, double* __restrict__ out, int64_t rows,
                            int64_t cols) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols, c = i % cols;
  out[c * rows + r] = in[i];
}

// ---- Gelman-Rubin partial sums (convergence.py:71-85) ---------------------------------------
// A wave serves kRhatU consecutive sources of one parameter: lanes stride over a source's C chains, the loads of
// the kRhatU sources are issued together (a wave with one 512-byte load in flight per pass left the kernel at
// 4 TB/s), each source's lane-strided partial sums are added on the vector ALU (wave_sum_lane63; the xor tree
// through the LDS crossbar - 24 ds_bpermute per pair of sums - was what bounded the kernel).  Sources beyond S
// are read as source S - 1 and not written.
// LPS lanes per source (64, or 32 / 16 when a source has no more chains than that: two / four sources share a
// load and the last steps of the sum are skipped); a wave serves kRhatU * 64 / LPS sources.
constexpr int kRhatU = 4;
template <int LPS>
__global__ __launch_bounds__(kBlockThreads) void k_rhat_partial(
    int pass, int P, int64_t S, int C, int64_t N, double rn, const double* __restrict__ mean,
    const double* __restrict__ m2, const double* __restrict__ gmean, double* __restrict__ sums) {
  constexpr int SPL = 64 / LPS;                            // sources per load
  const int64_t w = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPS, c0 = lane % LPS;
  const int64_t SG = (S + kRhatU * SPL - 1) / (kRhatU * SPL);
  if (w >= SG * P) return;
  const int p = static_cast<int>(w / SG);
  const int64_t s0 = (w - p * SG) * (kRhatU * SPL) + sub;
  int64_t su[kRhatU];
#pragma unroll
  for (int u = 0; u < kRhatU; ++u) su[u] = (s0 + u * SPL < S) ? s0 + u * SPL : S - 1;
  const double* mp = mean + p * N;
  const double* vp = m2 + p * N;
  double a0[kRhatU], a1[kRhatU], g[kRhatU];
#pragma unroll
  for (int u = 0; u < kRhatU; ++u) {
    a0[u] = 0.0;
    a1[u] = 0.0;
    g[u] = (pass == 0) ? 0.0 : gmean[p * S + su[u]];
  }
  for (int c = c0; c < C; c += LPS) {
    double m[kRhatU], v[kRhatU];
#pragma unroll
    for (int u = 0; u < kRhatU; ++u) m[u] = mp[su[u] * C + c];
    if (pass != 0) {                                       // uniform
#pragma unroll
      for (int u = 0; u < kRhatU; ++u) v[u] = vp[su[u] * C + c];
#pragma unroll
      for (int u = 0; u < kRhatU; ++u) {
        const double dm = m[u] - g[u];
        a0[u] += dm * dm;
        a1[u] += v[u] * rn;
      }
    } else {
#pragma unroll
      for (int u = 0; u < kRhatU; ++u) a0[u] += m[u];
    }
  }
#pragma unroll
  for (int u = 0; u < kRhatU; ++u) {
    a0[u] = wave_sum_last<LPS>(a0[u]);
    if (pass != 0) a1[u] = wave_sum_last<LPS>(a1[u]);
  }
  if (c0 == LPS - 1) {
#pragma unroll
    for (int u = 0; u < kRhatU; ++u) {
      if (s0 + u * SPL < S) {
        sums[p * S + s0 + u * SPL] = a0[u];
        if (pass != 0) sums[P * S + p * S + s0 + u * SPL] = a1[u];
      }
    }
  }
}

__global__ void k_rhat_finish(int64_t n, double Ctot, const double* __restrict__ sums,
                              double* __restrict__ rhat) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double within = sums[n + i] / Ctot;              // mean of the chain variances, :77
  const double between = sums[i] / (Ctot - 1.0);         // ddof=1 variance of the means, :78
  rhat[i] = (within > 0.0) ? sqrt((within + between) / within) : __builtin_nan("");   // :81-85
}

// one thread per parameter; means, vars [nchain][P]
__global__ void k_gelman_rubin(int nchain, int P, const double* __restrict__ means,
                               const double* __restrict__ vars, double* __restrict__ rhat) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double sm = 0.0, sv = 0.0;
  for (int c = 0; c < nchain; ++c) {
    sm += means[c * P + p];
    sv += vars[c * P + p];
  }
  const double g = sm / nchain, within = sv / nchain;
  double ss = 0.0;
  for (int c = 0; c < nchain; ++c) {
    const double dm = means[c * P + p] - g;
    ss += dm * dm;
  }
  const double between = ss / (nchain - 1);
  rhat[p] = (within > 0.0) ? sqrt((within + between) / within) : __builtin_nan("");
}

// ---- effective sample size (no counterpart in the reference: BASELINE.json asks for ESS/s) ----
// Multi-chain estimator of Gelman et al., BDA3 sect. 11.5, on the trace of the last run:
//   n_eff = m n / (1 + 2 sum_t rho_t),  rho_t = 1 - V_t / (2 var+),
//   V_t = sum_j sum_i (x_ij - x_(i-t)j)^2 / (m (n - t)),  var+ = (n-1)/n W + B/n,
// the sum running over t = 1, 2, ... until rho_(t+1) + rho_(t+2) < 0 at an odd t (or the lags
// run out).  One block of 256 threads per (group, parameter): thread = (chain j, lag group g),
// ntc = 2^ntc_shift >= C chain slots, G = 256/ntc lag groups.  The n <= kEssMaxRows trace values
// of every chain sit in LDS; a thread evaluates four consecutive lags per pass over its chain with
// a sliding window (two LDS reads serve four lags), the sum over chains is a wave shuffle
// reduction, and the stopping rule is applied once per batch of 4 G lags.
constexpr int kEssMaxRows = 128;
constexpr int kEssLags = 4;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_ess(int P, int64_t S, int C, int64_t N, int nrows, int ntc_shift,
                                             const double* __restrict__ trace /*[nrows][P][N]*/,
                                             double* __restrict__ ess /*[P][S]*/) {
  extern __shared__ double sh[];                       // x[nrows][ntc] + red[4 waves][kEssLags]
  const int64_t s = blockIdx.x / P;
  const int p = static_cast<int>(blockIdx.x % P);
  const int ntc = 1 << ntc_shift, G = 256 >> ntc_shift;
  const int tid = threadIdx.x, j = tid & (ntc - 1), g = tid >> ntc_shift;
  const int wave = tid >> 6, lane = tid & 63;
  const int wpg = ntc >> 6;                            // waves per lag group
  double* x = sh;
  double* red = sh + static_cast<size_t>(nrows) * ntc;
  const bool live = j < C;
  for (int i = g; i < nrows; i += G)
    x[i * ntc + j] = live ? trace[(static_cast<int64_t>(i) * P + p) * N + s * C + j] : 0.0;
  __syncthreads();
  double sum = 0.0;
  for (int i = 0; i < nrows; ++i) sum += x[i * ntc + j];
  const double mean_j = sum / nrows;
  double ss = 0.0;
  for (int i = 0; i < nrows; ++i) {
    const double dlt = x[i * ntc + j] - mean_j;
    ss += dlt * dlt;
  }
  // sum over the chains (group 0 holds one copy of each)
  auto block_sum = [&](double v) -> double {
    const double w = wave_sum((live && g == 0) ? v : 0.0);
    __syncthreads();
    if (lane == 0) red[wave] = w;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  const double m = static_cast<double>(C), n = static_cast<double>(nrows);
  const double W = block_sum(ss / (n - 1.0)) / m;                  // mean within-chain variance
  const double gmean = block_sum(mean_j) / m;
  const double dm = mean_j - gmean;
  const double B_over_n = (C > 1) ? block_sum(dm * dm) / (m - 1.0) : 0.0;
  const double varplus = (n - 1.0) / n * W + B_over_n;
  double rho_sum = 0.0, prev = 0.0;
  bool have_prev = false, done = !(varplus > 0.0);
  for (int t0 = 1; t0 < nrows && !done; t0 += G * kEssLags) {
    const int tb = t0 + g * kEssLags;                  // lags tb .. tb+3 of this thread's chain
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int i = tb; i < nrows; ++i) {
      const double xi = x[i * ntc + j], a0 = x[(i - tb) * ntc + j];
      const int k = i - tb;                            // wave-uniform
      double d = xi - a0;
      v0 += d * d;
      if (k >= 1) { d = xi - a1; v1 += d * d; }
      if (k >= 2) { d = xi - a2; v2 += d * d; }
      if (k >= 3) { d = xi - a3; v3 += d * d; }
      a3 = a2;
      a2 = a1;
      a1 = a0;
    }
    if (!live) v0 = v1 = v2 = v3 = 0.0;
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    v2 = wave_sum(v2);
    v3 = wave_sum(v3);
    __syncthreads();                                   // red of the previous batch has been read
    if (lane == 0) {
      red[wave * kEssLags + 0] = v0;
      red[wave * kEssLags + 1] = v1;
      red[wave * kEssLags + 2] = v2;
      red[wave * kEssLags + 3] = v3;
    }
    __syncthreads();
    for (int q = 0; q < G * kEssLags; ++q) {           // every thread applies the same rule
      const int t = t0 + q;
      if (t >= nrows) { done = true; break; }
      const int gq = q / kEssLags, u = q % kEssLags;
      double v = 0.0;
      for (int w = 0; w < wpg; ++w) v += red[(gq * wpg + w) * kEssLags + u];
      const double Vt = v / (m * (n - t));
      const double rho = 1.0 - Vt / (2.0 * varplus);
      // pairs (rho_1, rho_2), (rho_3, rho_4), ...: stop at the first pair with a negative sum
      if (!have_prev) {
        prev = rho;
        have_prev = true;
      } else {
        if (prev + rho < 0.0) { done = true; break; }
        rho_sum += prev + rho;
        have_prev = false;
      }
    }
  }
  if (tid == 0) ess[p * S + s] = (varplus > 0.0) ? m * n / (1.0 + 2.0 * rho_sum) : __builtin_nan("");
}

__global__ void k_scale(int64_t n, double f, double* x) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= f;
}

}  // namespace mi355
