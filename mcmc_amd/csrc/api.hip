// C ABI of libmi355mcmc.so (declared in include/mi355mcmc.h): context, SED catalogue,
// batched likelihood, Metropolis-Hastings state machine, Gelman-Rubin.
#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <limits>

#include "common.hpp"
#include "sed_kernels.hpp"

namespace mi355 {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

static const int kModelNParam[MI355_SED_NMODEL] = {3, 3, 6, 6, 2, 4, 4, 4};

// Fixed scratch of mi355_rhat_noalloc_async: [3][MI355_RHAT_NOALLOC_MAX] doubles (24 MB) that
// every GPU the library runs on holds until the process ends.  `scratch` is a host-side stand-in,
// not a device address: host code only ever uses the pointer hipGetSymbolAddress returned for the
// context's device (mi355_ctx::fixed_scratch).
static __device__ __align__(256) double scratch[3 * MI355_RHAT_NOALLOC_MAX];

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr size_t kLdsBig = 128 * 1024;   // dynamic LDS of a block whose kernel has been opted in (launch_lds); 160 KB per CU
constexpr size_t kLdsLimit = 64 * 1024 - kStaticTabBytes;   // dynamic LDS a block may ask for without opt-in (static: the 18 KB of 2^y and Box-Muller tables)

// LDS bytes a `bs`-chain block needs for the band tables of the sources it can touch
static size_t lds_bytes(int bs, int C, int nbm, int64_t nsources) {
  int64_t per_block = (bs % C == 0) ? bs / C : (bs - 1) / C + 2;
  if (per_block > nsources) per_block = nsources;
  return static_cast<size_t>(per_block) * kBandFields * nbm * sizeof(double);
}

// Largest block (256/128/64 chains) whose band tables fit in LDS; 0 if none does.  First within what
// a kernel may ask for without opting in (kLdsLimit: several blocks per CU), then within kLdsBig
// (many bands and few chains per source: 64 sources of 27 bands are 110 KB) - launch_lds opts the
// kernel in.
static int pick_block(int C, int nbm, int64_t nsources, size_t* lds) {
  for (size_t limit : {kLdsLimit, kLdsBig})
    for (int bs = kBlockThreads; bs >= 64; bs >>= 1) {
      *lds = lds_bytes(bs, C, nbm, nsources);
      if (*lds <= limit) return bs;
    }
  return 0;
}

// Launch with `lds` bytes of dynamic LDS, raising the kernel's limit first where it is needed.
template <typename Kern, typename... Args>
static int launch_lds(Kern kern, unsigned grid, int bs, size_t lds, hipStream_t stream, Args... args) {
  if (lds > kLdsLimit)
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(bs), lds, stream, args...);
  return 0;
}

template <template <int> class F, typename... A>
static int dispatch_model(int model, A&&... args) {
  switch (model) {
    case MI355_SED_M1N: return F<MI355_SED_M1N>::run(std::forward<A>(args)...);
    case MI355_SED_M1N_LOGA: return F<MI355_SED_M1N_LOGA>::run(std::forward<A>(args)...);
    case MI355_SED_M2N: return F<MI355_SED_M2N>::run(std::forward<A>(args)...);
    case MI355_SED_M2N_LOGA: return F<MI355_SED_M2N_LOGA>::run(std::forward<A>(args)...);
    case MI355_SED_M1: return F<MI355_SED_M1>::run(std::forward<A>(args)...);
    case MI355_SED_M2: return F<MI355_SED_M2>::run(std::forward<A>(args)...);
    case MI355_SED_THICK: return F<MI355_SED_THICK>::run(std::forward<A>(args)...);
    case MI355_SED_THICK_LOGA: return F<MI355_SED_THICK_LOGA>::run(std::forward<A>(args)...);
  }
  set_error("unknown SED model id %d", model);
  return -1;
}

template <int M>
struct LaunchLnlike {
  static int run(mi355_ctx* c, int64_t s0, int64_t n_eval, int C, const double* theta, double* lnl,
                 double* derived) {
    size_t lds = 0;
    const int bs = pick_block(C, c->nbm, ceil_div(n_eval, C), &lds);
    MI355_REQUIRE(bs > 0, "band tables (nb_max=%d, C=%d) do not fit in LDS", c->nbm, C);
    const unsigned grid = static_cast<unsigned>(ceil_div(n_eval, bs));
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
    { if (launch_lds(k_sed_lnlike<M>, grid, bs, lds, c->stream, c->bands.p,
                       c->nb.p, c->nbm, s0, n_eval, C, theta, lnl, derived, c->kc)) return -2; }
    MI355_HIP(hipGetLastError());
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
    if (!c->capturing) c->ev_valid = true;
    return 0;
  }
};

template <int M>
struct LaunchPriorPackage {
  static int run(mi355_ctx* c, int64_t n, const double* theta, double* prior, double* packaged) {
    const unsigned grid = static_cast<unsigned>(ceil_div(n, kBlockThreads));
    hipLaunchKernelGGL(k_sed_prior_package<M>, dim3(grid), dim3(kBlockThreads), 0, c->stream, n,
                       theta, prior, packaged, c->kc);
    MI355_HIP(hipGetLastError());
    return 0;
  }
};

template <int M>
struct LaunchFlux {
  static int run(mi355_ctx* c, int64_t n, const double* theta, int64_t nnu, const double* nu,
                 double* out) {
    const unsigned grid = static_cast<unsigned>(ceil_div(n * nnu, kBlockThreads));
    hipLaunchKernelGGL(k_sed_flux<M>, dim3(grid), dim3(kBlockThreads), 0, c->stream, n, theta, nnu,
                       nu, out, c->kc);
    MI355_HIP(hipGetLastError());
    return 0;
  }
};

template <int M>
struct LaunchStart {
  static int run(mi355_ctx* c, const MhArgs& a) {
    size_t lds = 0;
    const int bs = pick_block(c->C, c->nbm, c->S, &lds);
    MI355_REQUIRE(bs > 0, "band tables (nb_max=%d, C=%d) do not fit in LDS", c->nbm, c->C);
    const unsigned grid = static_cast<unsigned>(ceil_div(c->N, bs));
    { if (launch_lds(k_mh_start<M>, grid, bs, lds, c->stream, a)) return -2; }
    MI355_HIP(hipGetLastError());
    return 0;
  }
};

// Every source of the catalogue is in the domain of the clamp-free kernels (otherwise the plain,
// extended and incremental samplers - and the (chain, band) lane mapping of the first two - split a
// mixed catalogue into two launches over disjoint lists of chain blocks).
static bool sed_all_fast(const mi355_ctx* c) { return c->kc.fast_ok != 0.0 && c->nslow == 0; }

// Block size and lanes per chain of the next k_mh_run launch (ext: its extended instantiation, which
// keeps a sqrt matrix per lane in LDS).  Lanes per chain: 1 = lane per chain; 8 / 32 = lanes (chain,
// band), picked when the catalogue has sources on the fast path (the chain blocks of the others then
// run lane = chain beside them: LaunchRun) and the job is too small to give every SIMD of the GPU a
// wave of chains (1024 SIMDs x 64 lanes): tools/ubench_lanemap.hip has the crossover there, 3x
// (nb = 27) / 1.6x (nb = 8) below it and a loss of 3x above.
// mi355_mh_set_lane_map overrides (tests hold the two mappings to identical chains).  The block is
// the largest of 256 / 128 / 64 lanes whose band tables, per-lane extras and exchange slots fit.
struct SedPlan {
  int bs = 0;
  size_t lds = 0;
  int G = 1;
};
static int pick_block_x(const mi355_ctx* c, size_t per_lane, size_t fixed, size_t* lds) {
  for (size_t limit : {kLdsLimit, kLdsBig})
    for (int bs = kBlockThreads; bs >= 64; bs >>= 1) {
      *lds = lds_bytes(bs, c->C, c->nbm, c->S) + per_lane * bs + fixed;
      if (*lds <= limit) return bs;
    }
  return 0;
}
static SedPlan sed_plan(const mi355_ctx* c, bool ext) {
  SedPlan p;
  const size_t per_lane = ext ? static_cast<size_t>(c->P * (c->P + 1) / 2) * sizeof(double) : 0;
  p.bs = pick_block_x(c, per_lane, 0, &p.lds);
  if (c->lane_map == 1 || c->kc.fast_ok == 0.0 || c->nslow >= c->S || c->model.model == MI355_SED_M2) return p;
  const int g = c->nb_most <= 8 ? 8 : (c->nb_most <= 32 ? 32 : 0);
  if (g == 0 || c->nb_most < 2) return p;
  if (c->lane_map != 2 && c->N * g > 65536) return p;
  size_t lds = 0;
  const int bs = pick_block_x(c, per_lane + 3 * sizeof(double2), 16, &lds);   // + the exchange slot, two slots of draws
  if (bs > 0) {
    p.bs = bs;
    p.lds = lds;
    p.G = g;
  }
  return p;
}

// Lists of the chain blocks (bs chains each: the kernel's lanes per block over its lanes per chain) whose
// sources are all fast / not all fast.
static int build_block_lists(mi355_ctx* c, int bs) {
  if (c->blk_bs == bs && c->blk_C == c->C) return 0;
  const int64_t nblk = ceil_div(c->N, bs);
  std::vector<int32_t> fast, slow;
  for (int64_t b = 0; b < nblk; ++b) {
    const int64_t s_lo = b * bs / c->C, s_hi = (std::min<int64_t>((b + 1) * bs, c->N) - 1) / c->C;
    bool any = false;
    for (int64_t s = s_lo; s <= s_hi && !any; ++s) any = c->src_slow[s] != 0;
    (any ? slow : fast).push_back(static_cast<int32_t>(b));
  }
  if (c->blk_fast.alloc(std::max<size_t>(1, fast.size())) || c->blk_slow.alloc(std::max<size_t>(1, slow.size())))
    return -2;
  if (!fast.empty())
    MI355_HIP(hipMemcpyAsync(c->blk_fast.p, fast.data(), fast.size() * sizeof(int32_t),
                             hipMemcpyHostToDevice, c->stream));
  if (!slow.empty())
    MI355_HIP(hipMemcpyAsync(c->blk_slow.p, slow.data(), slow.size() * sizeof(int32_t),
                             hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));               // the host vectors go out of scope
  c->nblk_fast = static_cast<int64_t>(fast.size());
  c->nblk_slow = static_cast<int64_t>(slow.size());
  c->blk_bs = bs;
  c->blk_C = c->C;
  return 0;
}

// Whether the fused SED sampler of this context runs as k_mh_run_inc: asked for, a two-component
// normalised model on the clamp-free path, bands within the register cache, one parameter per
// block with block b = parameter b.
static bool incremental_applies(const mi355_ctx* c) {
  if (!c->incremental || !c->mh_ready || gmh_active(c)) return false;
  const int m = c->model.model;
  if (m != MI355_SED_M2N && m != MI355_SED_M2N_LOGA) return false;
  // (a mixed catalogue: the blocks with a source outside the fast domain run the plain per-lane kernel beside it)
  if (c->kc.fast_ok == 0.0 || c->nb_most > kIncBands || c->nblock != c->mhP || c->npn != 1) return false;
  for (int b = 0; b < c->nblock; ++b)
    if (c->blockmask[b] != (1u << b)) return false;
  return true;
}

// One parameter per block, block b = parameter b: the plain sampler then runs its in-place instantiation.
static bool single_blocks(const mi355_ctx* c) {
  if (c->nblock <= 0 || c->npn != 1) return false;
  for (int b = 0; b < c->nblock; ++b)
    if (c->blockmask[b] != (1u << b)) return false;
  return true;
}

template <int M>
struct LaunchRun {
  static int run(mi355_ctx* c, const MhArgs& a) {
    size_t lds = 0;
    const int bs = pick_block(c->C, c->nbm, c->S, &lds);
    MI355_REQUIRE(bs > 0, "band tables (nb_max=%d, C=%d) do not fit in LDS", c->nbm, c->C);
    const unsigned grid = static_cast<unsigned>(ceil_div(c->N, bs));
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
    if constexpr (M == MI355_SED_M2N || M == MI355_SED_M2N_LOGA) {
      const size_t lds_inc = lds + 2 * static_cast<size_t>(kIncBands) * bs * sizeof(double);
      if (incremental_applies(c) && lds_inc <= kLdsLimit) {
        if (sed_all_fast(c)) {
          if (launch_lds(k_mh_run_inc<M>, grid, bs, lds_inc, c->stream, a)) return -2;
        } else {
          // mixed catalogue, as the plain sampler below: the slow blocks through the per-lane kernel on stream2
          if (build_block_lists(c, bs)) return -2;
          MhArgs af = a, as = a;
          af.blocklist = c->blk_fast.p;
          as.blocklist = c->blk_slow.p;
          MI355_HIP(hipEventRecord(c->ev_fork, c->stream));
          MI355_HIP(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
          if (launch_lds((k_mh_run<M, false>), static_cast<unsigned>(c->nblk_slow), bs, lds, c->stream2, as)) return -2;
          MI355_HIP(hipGetLastError());
          MI355_HIP(hipEventRecord(c->ev_join, c->stream2));
          if (c->nblk_fast > 0) {
            if (launch_lds(k_mh_run_inc<M>, static_cast<unsigned>(c->nblk_fast), bs, lds_inc, c->stream, af)) return -2;
          }
          MI355_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
        MI355_HIP(hipGetLastError());
        if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
        if (!c->capturing) c->ev_valid = true;
        return 0;
      }
    }
    if (a.mode != 0 || a.comom != nullptr) {
      // extended sampler: correlated / rotated proposals, co-moments (sed_mh_fused_ok checked it fits);
      // its own block size: the per-lane sqrt matrices count against LDS
      const SedPlan pe = sed_plan(c, true);
      MI355_REQUIRE(pe.bs > 0, "the extended fused sampler does not serve this catalogue (step-wise engine only)");
      const unsigned grid_e = static_cast<unsigned>(ceil_div(c->N, pe.bs / pe.G));
      if (!sed_all_fast(c)) {
        // sources outside the fast domain: the per-lane instantiation for the whole grid (sed_plan keeps
        // lane = chain then), or - mixed catalogue - for the blocks that hold such a source, beside the
        // fast instantiation for all the others, as the plain sampler below
        if (c->kc.fast_ok == 0.0) {
          if (launch_lds((k_mh_run<M, false, true>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2;
        } else {
          // (a small job: both kernels with lanes = (chain, band), the exact path band-parallel as well)
          if (build_block_lists(c, pe.bs / pe.G)) return -2;
          MhArgs af = a, as = a;
          af.blocklist = c->blk_fast.p;
          as.blocklist = c->blk_slow.p;
          MI355_HIP(hipEventRecord(c->ev_fork, c->stream));
          MI355_HIP(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
          const unsigned gs = static_cast<unsigned>(c->nblk_slow);
          if constexpr (M != MI355_SED_M2) {
            if (pe.G == 8)
              { if (launch_lds((k_mh_run<M, false, true, 8>), gs, pe.bs, pe.lds, c->stream2, as)) return -2; }
            else if (pe.G == 32)
              { if (launch_lds((k_mh_run<M, false, true, 32>), gs, pe.bs, pe.lds, c->stream2, as)) return -2; }
            else
              { if (launch_lds((k_mh_run<M, false, true>), gs, pe.bs, pe.lds, c->stream2, as)) return -2; }
          } else {
            { if (launch_lds((k_mh_run<M, false, true>), gs, pe.bs, pe.lds, c->stream2, as)) return -2; }
          }
          MI355_HIP(hipGetLastError());
          MI355_HIP(hipEventRecord(c->ev_join, c->stream2));
          if (c->nblk_fast > 0) {
            const unsigned gf = static_cast<unsigned>(c->nblk_fast);
            if constexpr (M != MI355_SED_M2) {
              if (pe.G == 8)
                { if (launch_lds((k_mh_run<M, true, true, 8>), gf, pe.bs, pe.lds, c->stream, af)) return -2; }
              else if (pe.G == 32)
                { if (launch_lds((k_mh_run<M, true, true, 32>), gf, pe.bs, pe.lds, c->stream, af)) return -2; }
              else
                { if (launch_lds((k_mh_run<M, true, true>), gf, pe.bs, pe.lds, c->stream, af)) return -2; }
            } else {
              { if (launch_lds((k_mh_run<M, true, true>), gf, pe.bs, pe.lds, c->stream, af)) return -2; }
            }
          }
          MI355_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
      } else if (pe.G == 1 && pe.bs == kBlockThreads && a.mode == 1 && a.inj_n == nullptr) {
        // Cholesky-correlated, free-running, lane = chain in full blocks (its LDS columns are
        // kBlockThreads lanes apart): the instantiation compiled for that proposal
        // (with the running co-moments: the instantiation that keeps them in registers)
        if (a.comom != nullptr) {
          if (launch_lds((k_mh_run<M, true, true, 1, false, 2>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2;
        } else {
          if (launch_lds((k_mh_run<M, true, true, 1, false, 1>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2;
        }
      } else if constexpr (M != MI355_SED_M2) {
        if (pe.G == 8)
          { if (launch_lds((k_mh_run<M, true, true, 8>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2; }
        else if (pe.G == 32)
          { if (launch_lds((k_mh_run<M, true, true, 32>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2; }
        else
          { if (launch_lds((k_mh_run<M, true, true>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2; }
      } else {
        { if (launch_lds((k_mh_run<M, true, true>), grid_e, pe.bs, pe.lds, c->stream, a)) return -2; }
      }
    } else if (sed_all_fast(c)) {
      const SedPlan pp = sed_plan(c, false);
      const unsigned grid_g = static_cast<unsigned>(ceil_div(c->N, pp.bs / pp.G));
      if constexpr (M != MI355_SED_M2) {
        if (pp.G == 8 && single_blocks(c))
          { if (launch_lds((k_mh_run<M, true, false, 8, true>), grid_g, pp.bs, pp.lds, c->stream, a)) return -2; }
        else if (pp.G == 32 && single_blocks(c))
          { if (launch_lds((k_mh_run<M, true, false, 32, true>), grid_g, pp.bs, pp.lds, c->stream, a)) return -2; }
        else if (pp.G == 8)
          { if (launch_lds((k_mh_run<M, true, false, 8>), grid_g, pp.bs, pp.lds, c->stream, a)) return -2; }
        else if (pp.G == 32)
          { if (launch_lds((k_mh_run<M, true, false, 32>), grid_g, pp.bs, pp.lds, c->stream, a)) return -2; }
        else if (single_blocks(c))
          { if (launch_lds((k_mh_run<M, true, false, 1, true>), grid, bs, lds, c->stream, a)) return -2; }
        else
          { if (launch_lds((k_mh_run<M, true>), grid, bs, lds, c->stream, a)) return -2; }
      } else {
        { if (launch_lds((k_mh_run<M, true>), grid, bs, lds, c->stream, a)) return -2; }
      }
    } else if (c->kc.fast_ok == 0.0)
      { if (launch_lds((k_mh_run<M, false>), grid, bs, lds, c->stream, a)) return -2; }
    else {
      // mixed catalogue: the blocks with a source outside the fast domain on stream2 (each lane
      // follows its own source there), all the others through the fast instantiation beside them
      // (a small job: both kernels with lanes = (chain, band), the exact path band-parallel as well)
      const SedPlan pp = sed_plan(c, false);
      if (build_block_lists(c, pp.G > 1 ? pp.bs / pp.G : bs)) return -2;
      MhArgs af = a, as = a;
      af.blocklist = c->blk_fast.p;
      as.blocklist = c->blk_slow.p;
      MI355_HIP(hipEventRecord(c->ev_fork, c->stream));
      MI355_HIP(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      const unsigned gs = static_cast<unsigned>(c->nblk_slow);
      bool slow_done = false;
      if constexpr (M != MI355_SED_M2) {
        slow_done = pp.G > 1;
        if (pp.G == 8)
          { if (launch_lds((k_mh_run<M, false, false, 8>), gs, pp.bs, pp.lds, c->stream2, as)) return -2; }
        else if (pp.G == 32)
          { if (launch_lds((k_mh_run<M, false, false, 32>), gs, pp.bs, pp.lds, c->stream2, as)) return -2; }
      }
      if (!slow_done)
        { if (launch_lds((k_mh_run<M, false>), gs, bs, lds, c->stream2, as)) return -2; }
      MI355_HIP(hipGetLastError());
      MI355_HIP(hipEventRecord(c->ev_join, c->stream2));
      if (c->nblk_fast > 0) {
        const unsigned gf = static_cast<unsigned>(c->nblk_fast);
        bool done = false;
        if constexpr (M != MI355_SED_M2) {
          done = pp.G > 1;
          if (pp.G == 8 && single_blocks(c))
            { if (launch_lds((k_mh_run<M, true, false, 8, true>), gf, pp.bs, pp.lds, c->stream, af)) return -2; }
          else if (pp.G == 32 && single_blocks(c))
            { if (launch_lds((k_mh_run<M, true, false, 32, true>), gf, pp.bs, pp.lds, c->stream, af)) return -2; }
          else if (pp.G == 8)
            { if (launch_lds((k_mh_run<M, true, false, 8>), gf, pp.bs, pp.lds, c->stream, af)) return -2; }
          else if (pp.G == 32)
            { if (launch_lds((k_mh_run<M, true, false, 32>), gf, pp.bs, pp.lds, c->stream, af)) return -2; }
        }
        if (done)
          ;
        else if (single_blocks(c))
          { if (launch_lds((k_mh_run<M, true, false, 1, true>), gf, bs, lds, c->stream, af)) return -2; }
        else
          { if (launch_lds((k_mh_run<M, true>), gf, bs, lds, c->stream, af)) return -2; }
      }
      MI355_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    MI355_HIP(hipGetLastError());
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
    if (!c->capturing) c->ev_valid = true;
    return 0;
  }
};

static MhArgs base_args(mi355_ctx* c) {
  MhArgs a{};
  a.bands = c->bands.p;
  a.nb = c->nb.p;
  a.nbm = c->nbm;
  a.C = c->C;
  a.N = c->N;
  a.theta = c->theta.p;
  a.sigma = c->sigma.p;
  a.lnPr = c->lnPr.p;
  a.mean = c->mean.p;
  a.m2 = c->m2.p;
  a.naccept = c->naccept.p;
  a.accept = c->accept.p;
  a.flags = c->flags.p;
  a.nblock = c->nblock;
  for (int i = 0; i < kMaxParam; ++i) {
    a.blockmask[i] = c->blockmask[i];
    a.rank_in_block[i] = c->rank_in_block[i];
  }
  a.npn = c->npn;
  a.step_total = c->step_total;
  a.stage_step = c->stage_step;
  a.nsamp0 = c->nsamp;
  a.stage = c->stage;
  a.seed = c->seed;
  a.k = c->kc;
  return a;
}

static int check_start_flags(mi355_ctx* c) {
  std::vector<int32_t> f(static_cast<size_t>(c->N));
  MI355_HIP(hipMemcpyAsync(f.data(), c->flags.p, f.size() * sizeof(int32_t), hipMemcpyDeviceToHost,
                           c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  int64_t bad = 0, first = -1;
  for (int64_t i = 0; i < c->N; ++i)
    if (f[i] & kFlagStartFailed) {
      if (first < 0) first = i;
      ++bad;
    }
  MI355_REQUIRE(bad == 0,
                "ZeroPriorError: %lld of %lld chains could not be started with prior>0 "
                "(first: chain %lld) [MCMC.py:174-175]",
                static_cast<long long>(bad), static_cast<long long>(c->N),
                static_cast<long long>(first));
  return 0;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

const char* mi355_last_error(void) { return g_err.c_str(); }
int mi355_abi_version(void) { return MI355_ABI_VERSION; }

int mi355_init(int device, uint64_t seed, mi355_ctx** out) {
  MI355_REQUIRE(out != nullptr, "mi355_init: out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("mi355_init: no HIP device available (%s); this library has no CPU fallback",
              hipGetErrorString(e));
    return -2;
  }
  MI355_REQUIRE(device >= 0 && device < ndev, "mi355_init: device %d out of range [0,%d)", device,
                ndev);
  MI355_HIP(hipSetDevice(device));
  mi355_ctx* c = new mi355_ctx();
  c->device = device;
  c->seed = seed;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    // the side stream of mixed catalogues: highest priority, so that it gets a hardware queue of its
    // own (streams of equal priority may share one and serialise) and its few long blocks start first
    int prio_lo = 0, prio_hi = 0;
    e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi);
  }
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("stream/event creation failed: %s", hipGetErrorString(e));
    mi355_free(c);
    return -2;
  }
  // this device's copy of the fixed scratch array (each GPU has its own, at its own address)
  e = hipGetSymbolAddress(reinterpret_cast<void**>(&c->fixed_scratch), HIP_SYMBOL(scratch));
  if (e != hipSuccess || !c->fixed_scratch) {
    set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
    mi355_free(c);
    return -2;
  }
  *out = c;
  return 0;
}

void mi355_free(mi355_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamDestroy(c->stream2);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  cl_destroy(c->cl);
  beam_destroy(c->beam);
  gmh_destroy(c->gmh);
  delete c;
}

double mi355_last_kernel_ms(mi355_ctx* c) {
  if (!c || !c->ev_valid) return -1.0;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
  return static_cast<double>(ms);
}

void* mi355_stream(mi355_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

int mi355_sync(mi355_ctx* c) {
  MI355_REQUIRE(c, "null context");
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_sed_nparam(int32_t model) {
  if (model < 0 || model >= MI355_SED_NMODEL) return -1;
  return kModelNParam[model];
}

int mi355_sed_model_default(int32_t model, mi355_sed_model* out) {
  MI355_REQUIRE(out, "null output");
  MI355_REQUIRE(model >= 0 && model < MI355_SED_NMODEL, "unknown SED model id %d", model);
  out->model = model;
  out->nparam = kModelNParam[model];
  out->h_over_k = 0.04799237;
  out->nu_b = 1000.0;
  out->prefac = 1.0e-9;
  out->Tmin = 3.0;
  out->Tmax = 100.0;
  out->bmin = 0.0;
  out->bmax = 3.0;
  out->nu_rescale = 1000.0;
  return 0;
}

int mi355_sed_load(mi355_ctx* c, const mi355_sed_model* m, int64_t S, int32_t nb_max,
                   const int32_t* nb, const double* nu, const double* d, const double* sig) {
  MI355_REQUIRE(c && m && nb && nu && d && sig, "mi355_sed_load: null argument");
  MI355_REQUIRE(m->model >= 0 && m->model < MI355_SED_NMODEL, "unknown SED model id %d", m->model);
  MI355_REQUIRE(m->nparam == kModelNParam[m->model], "model %d has %d parameters, got nparam=%d",
                m->model, kModelNParam[m->model], m->nparam);
  MI355_REQUIRE(S > 0 && nb_max > 0, "mi355_sed_load: empty catalogue (S=%lld, nb_max=%d)",
                static_cast<long long>(S), nb_max);
  MI355_REQUIRE(static_cast<size_t>(nb_max) * kBandFields * sizeof(double) <= kLdsBig,
                "nb_max=%d too large for the LDS band tables", nb_max);
  for (int64_t s = 0; s < S; ++s)
    MI355_REQUIRE(nb[s] >= 0 && nb[s] <= nb_max, "nb[%lld]=%d outside [0,%d]",
                  static_cast<long long>(s), nb[s], nb_max);
  MI355_HIP(hipSetDevice(c->device));
  c->sed_loaded = false;
  mh_invalidate(c);
  c->model = *m;
  c->P = m->nparam;
  c->S = S;
  c->nbm = nb_max;
  c->nb_most = 0;
  for (int64_t s = 0; s < S; ++s) c->nb_most = std::max<int>(c->nb_most, nb[s]);
  auto dd_log = [](double v, double* hi, double* lo) {   // log2(v) as hi + lo (x87 80-bit log2l)
    const long double l = log2l(static_cast<long double>(v));
    *hi = static_cast<double>(l);
    *lo = static_cast<double>(l - static_cast<long double>(*hi));
  };
  c->kc = SedConsts{m->h_over_k, m->nu_b, m->prefac, m->Tmin, m->Tmax, m->bmin, m->bmax,
                    m->nu_rescale, m->h_over_k * m->nu_b, 0.0, 0.0, 0.0, 0.0, 0.0};
  c->kc.hk_nub2 = static_cast<double>(static_cast<long double>(c->kc.hk_nub) *
                                      1.442695040888963407359924681001892137L);
  dd_log(m->nu_b, &c->kc.ln_nub_hi, &c->kc.ln_nub_lo);
  c->kc.thick_c = static_cast<double>(log2l(static_cast<long double>(m->nu_rescale) /
                                            static_cast<long double>(m->nu_b)));
  const bool base_is_nu = (m->model == MI355_SED_M1 || m->model == MI355_SED_M2);
  // host-side derived band fields (each rounded once, as the reference computes them)
  const size_t stride = static_cast<size_t>(kBandFields) * nb_max;
  std::vector<double> h(static_cast<size_t>(S) * stride);
  for (int64_t s = 0; s < S; ++s) {
    double* o = h.data() + s * stride;
    for (int j = 0; j < nb_max; ++j) {
      const bool live = j < nb[s];
      const double v = live ? nu[s * nb_max + j] : m->nu_b;
      const double sg = live ? sig[s * nb_max + j] : 1.0;
      o[j * kBandFields + kF_NU] = v;
      o[j * kBandFields + kF_HNU] = m->h_over_k * v;       // x = h_over_k*nu/T     model.py:103
      o[j * kBandFields + kF_RATIO] = v / m->nu_b;           // nu/nu_norm            model.py:109
      o[j * kBandFields + kF_D] = live ? d[s * nb_max + j] : 0.0;
      o[j * kBandFields + kF_W] = live ? 1.0 / (sg * sg) : 0.0;   // 1/sig2   GaussianData.py:19,47
      // log2 of the band's power-law base (the rounded double the reference raises to 3+beta)
      dd_log(base_is_nu ? v : v / m->nu_b, &o[j * kBandFields + kF_LHI],
             &o[j * kBandFields + kF_LLO]);
      o[j * kBandFields + kF_HNU2] = static_cast<double>(static_cast<long double>(o[j * kBandFields + kF_HNU]) *
                                              1.442695040888963407359924681001892137L);
      if (live)
        MI355_REQUIRE(v > 0.0 && std::isfinite(v) && std::fabs(o[j * kBandFields + kF_LHI]) < 100.0,
                      "source %lld band %d: frequency %g GHz out of range",
                      static_cast<long long>(s), j, v);
    }
  }
  // The fast kernels run without argument clamps: allow them only if, over the whole prior box,
  // every Planck argument h nu/kT and every power-law exponent (3+beta) log2(base) stays below 300
  // (products of two such exponentials still fit a double); otherwise the exact OCML path runs.
  {
    const double emax = std::fmax(std::fabs(3.0 + m->bmin), std::fabs(3.0 + m->bmax));
    bool ok = m->Tmin > 0.0 && std::isfinite(m->bmin) && std::isfinite(m->bmax) &&
              c->kc.hk_nub / m->Tmin < 300.0 &&
              m->Tmax > 0.0 && c->kc.hk_nub2 / m->Tmax >= 0.5 &&   // 2^(y_b) - 1 by subtraction

              std::fmax(std::fabs(m->bmin), std::fabs(m->bmax)) * std::fabs(c->kc.ln_nub_hi) < 300.0;
    // The band conditions are per source: a source with a band outside them (a radio point, a
    // mid-infrared one) takes the exact path by itself and leaves the rest of the catalogue on the
    // fast one.
    c->src_slow.assign(static_cast<size_t>(S), 0);
    c->nslow = 0;
    for (int64_t s = 0; s < S && ok; ++s)
      for (int j = 0; j < nb[s]; ++j) {
        const double* o = h.data() + s * stride;
        if (!(o[j * kBandFields + kF_HNU] / m->Tmin < 300.0) ||
            !(emax * std::fabs(o[j * kBandFields + kF_LHI]) < 300.0) ||
            !(m->Tmax > 0.0 && o[j * kBandFields + kF_HNU] / m->Tmax >= 0.02)) {   // 1 - 2^-y keeps >= 47 bits
          c->src_slow[s] = 1;
          ++c->nslow;
          break;
        }
      }
    if (c->nslow == S) ok = false;                         // nothing left for the fast kernels
    c->kc.fast_ok = ok ? 1.0 : 0.0;
    if (!ok) {
      c->src_slow.assign(static_cast<size_t>(S), 1);
      c->nslow = S;
    }
  }
  c->blk_bs = c->blk_C = 0;                                // block lists of the previous catalogue
  std::vector<int32_t> nbdev(static_cast<size_t>(S));
  for (int64_t s = 0; s < S; ++s)
    nbdev[s] = nb[s] | ((c->kc.fast_ok != 0.0 && c->src_slow[s]) ? kNbSlow : 0);
  if (c->bands.alloc(h.size())) return -2;
  if (c->nb.alloc(static_cast<size_t>(S))) return -2;
  MI355_HIP(hipMemcpyAsync(c->bands.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice,
                           c->stream));
  MI355_HIP(hipMemcpyAsync(c->nb.p, nbdev.data(), static_cast<size_t>(S) * sizeof(int32_t),
                           hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  c->sed_loaded = true;
  return 0;
}

int mi355_sed_lnlike_dev_async(mi355_ctx* c, int64_t s0, int64_t ns, int32_t C,
                               const double* theta_dev, double* lnL_dev, double* derived_dev) {
  MI355_REQUIRE(c && c->sed_loaded, "mi355_sed_lnlike: no catalogue loaded (call mi355_sed_load)");
  MI355_REQUIRE(theta_dev && lnL_dev, "mi355_sed_lnlike: null buffer");
  MI355_REQUIRE(s0 >= 0 && ns > 0 && s0 + ns <= c->S, "source range [%lld,%lld) outside [0,%lld)",
                static_cast<long long>(s0), static_cast<long long>(s0 + ns),
                static_cast<long long>(c->S));
  MI355_REQUIRE(C > 0, "C must be positive");
  MI355_HIP(hipSetDevice(c->device));
  return dispatch_model<LaunchLnlike>(c->model.model, c, s0, ns * C, static_cast<int>(C), theta_dev,
                                      lnL_dev, derived_dev);
}

int mi355_sed_lnlike(mi355_ctx* c, int64_t s0, int64_t ns, int32_t C, const double* theta,
                     double* lnL, double* derived) {
  MI355_REQUIRE(c && c->sed_loaded, "mi355_sed_lnlike: no catalogue loaded (call mi355_sed_load)");
  MI355_REQUIRE(theta && lnL, "mi355_sed_lnlike: null buffer");
  MI355_REQUIRE(ns > 0 && C > 0, "empty batch");
  MI355_HIP(hipSetDevice(c->device));
  const int64_t n = ns * C;
  const int P = c->P;
  if (c->scratch_a.alloc(static_cast<size_t>(n) * P)) return -2;
  if (c->scratch_b.alloc(static_cast<size_t>(n) * P)) return -2;
  if (c->scratch_c.alloc(static_cast<size_t>(n) * 3)) return -2;
  MI355_HIP(hipMemcpyAsync(c->scratch_a.p, theta, static_cast<size_t>(n) * P * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_transpose, dim3(static_cast<unsigned>(ceil_div(n * P, 256))), dim3(256), 0,
                     c->stream, c->scratch_a.p, c->scratch_b.p, n, static_cast<int64_t>(P));
  MI355_HIP(hipGetLastError());
  double* lnl_d = c->scratch_c.p;
  double* der_d = derived ? c->scratch_c.p + n : nullptr;
  int rc = mi355_sed_lnlike_dev_async(c, s0, ns, C, c->scratch_b.p, lnl_d, der_d);
  if (rc) return rc;
  MI355_HIP(hipMemcpyAsync(lnL, lnl_d, static_cast<size_t>(n) * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  std::vector<double> der_h;
  if (derived) {
    der_h.resize(static_cast<size_t>(n) * 2);
    MI355_HIP(hipMemcpyAsync(der_h.data(), der_d, der_h.size() * sizeof(double),
                             hipMemcpyDeviceToHost, c->stream));
  }
  MI355_HIP(hipStreamSynchronize(c->stream));
  if (derived)
    for (int64_t i = 0; i < n; ++i) {
      derived[2 * i] = der_h[i];
      derived[2 * i + 1] = der_h[n + i];
    }
  return 0;
}

int mi355_sed_flux(mi355_ctx* c, int64_t n, const double* theta, int64_t nnu, const double* nu,
                   double* out) {
  MI355_REQUIRE(c && c->sed_loaded, "mi355_sed_flux: no model loaded");
  MI355_REQUIRE(theta && nu && out && n > 0 && nnu > 0, "mi355_sed_flux: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  const int P = c->P;
  const int nc = (c->model.model == MI355_SED_M2) ? 2 : 1;
  const size_t nt = static_cast<size_t>(n) * P, no = static_cast<size_t>(n) * nnu * nc;
  if (c->scratch_a.alloc(nt + nnu) || c->scratch_b.alloc(no)) return -2;
  MI355_HIP(hipMemcpyAsync(c->scratch_a.p, theta, nt * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(c->scratch_a.p + nt, nu, static_cast<size_t>(nnu) * 8,
                           hipMemcpyHostToDevice, c->stream));
  int rc = dispatch_model<LaunchFlux>(c->model.model, c, n, c->scratch_a.p, nnu,
                                      c->scratch_a.p + nt, c->scratch_b.p);
  if (rc) return rc;
  MI355_HIP(hipMemcpyAsync(out, c->scratch_b.p, no * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_sed_prior_package(mi355_ctx* c, int64_t n, const double* theta, double* prior,
                            double* packaged) {
  MI355_REQUIRE(c && c->sed_loaded, "mi355_sed_prior_package: no model loaded");
  MI355_REQUIRE(theta && n > 0, "empty batch");
  MI355_HIP(hipSetDevice(c->device));
  const int P = c->P;
  if (c->scratch_a.alloc(static_cast<size_t>(n) * P)) return -2;
  if (c->scratch_b.alloc(static_cast<size_t>(n) * P)) return -2;
  if (c->scratch_c.alloc(static_cast<size_t>(n))) return -2;
  MI355_HIP(hipMemcpyAsync(c->scratch_a.p, theta, static_cast<size_t>(n) * P * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
  int rc = dispatch_model<LaunchPriorPackage>(c->model.model, c, n, c->scratch_a.p,
                                              prior ? c->scratch_c.p : nullptr,
                                              packaged ? c->scratch_b.p : nullptr);
  if (rc) return rc;
  if (prior)
    MI355_HIP(hipMemcpyAsync(prior, c->scratch_c.p, static_cast<size_t>(n) * sizeof(double),
                             hipMemcpyDeviceToHost, c->stream));
  if (packaged)
    MI355_HIP(hipMemcpyAsync(packaged, c->scratch_b.p, static_cast<size_t>(n) * P * sizeof(double),
                             hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// upload a host [N][P] (per_chain) or [P] array as device [P][N] / [P]
static int upload_param(mi355_ctx* c, const double* host, int per_chain, DevBuf<double>& tmp,
                        DevBuf<double>& out) {
  const int P = c->mhP;
  if (!per_chain) {
    if (out.alloc(static_cast<size_t>(P))) return -2;
    MI355_HIP(hipMemcpyAsync(out.p, host, P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
  }
  const size_t n = static_cast<size_t>(c->N) * P;
  if (tmp.alloc(n) || out.alloc(n)) return -2;
  MI355_HIP(hipMemcpyAsync(tmp.p, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_transpose, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0,
                     c->stream, tmp.p, out.p, c->N, static_cast<int64_t>(P));
  MI355_HIP(hipGetLastError());
  return 0;
}

int mi355_mh_init(mi355_ctx* c, int32_t C, const double* theta0, int32_t theta0_per_chain,
                  const double* sigma, int32_t sigma_per_chain, const int32_t* param_blocks,
                  int32_t nblock, int32_t randomize_start) {
  MI355_REQUIRE(c && c->sed_loaded, "mi355_mh_init: no catalogue loaded (call mi355_sed_load)");
  MI355_REQUIRE(theta0 && sigma, "mi355_mh_init: null theta0/sigma");
  MI355_REQUIRE(C > 0, "C must be positive");
  MI355_REQUIRE(nblock >= 0 && nblock <= c->P, "nblock=%d outside [0,%d]", nblock, c->P);
  MI355_REQUIRE(nblock == 0 || param_blocks, "param_blocks required when nblock>0");
  MI355_HIP(hipSetDevice(c->device));
  c->mhP = c->P;
  c->mhS = c->S;
  if (c->gmh) gmh_deactivate(c);
  const int P = c->mhP;
  c->mh_ready = false;
  c->C = C;
  c->N = c->mhS * C;
  c->nblock = nblock;
  int maxblk = 1;
  for (int b = 0; b < kMaxParam; ++b) c->blockmask[b] = 0;
  for (int i = 0; i < kMaxParam; ++i) c->rank_in_block[i] = 0;
  if (nblock > 0) {
    int count[kMaxParam] = {0};
    for (int i = 0; i < P; ++i) {
      const int b = param_blocks[i];
      MI355_REQUIRE(b >= -1 && b < nblock, "param_blocks[%d]=%d outside [-1,%d)", i, b, nblock);
      if (b < 0) continue;   // parameter left out of every block (MCMC.py:85-88)
      c->rank_in_block[i] = count[b]++;
      c->blockmask[b] |= 1u << i;
      if (count[b] > maxblk) maxblk = count[b];
    }
    c->npn = maxblk == 1 ? 1 : ((maxblk + 1) & ~1);
  } else {
    for (int i = 0; i < P; ++i) c->rank_in_block[i] = i;
    c->npn = (P + 1) & ~1;
  }
  const size_t NP = static_cast<size_t>(c->N) * P;
  if (c->theta.alloc(NP) || c->sigma.alloc(NP) || c->mean.alloc(NP) || c->m2.alloc(NP)) return -2;
  if (c->lnPr.alloc(c->N) || c->naccept.alloc(c->N) || c->flags.alloc(c->N)) return -2;
  if (c->accept.alloc(static_cast<size_t>(c->N) * (nblock > 0 ? nblock : 1))) return -2;
  if (upload_param(c, theta0, theta0_per_chain, c->scratch_c, c->scratch_a)) return -2;
  if (upload_param(c, sigma, sigma_per_chain, c->scratch_c, c->scratch_b)) return -2;
  c->stage_step = 0;
  c->nsamp = 0;
  c->nkeep = 0;
  MhArgs a = base_args(c);
  a.center = c->scratch_a.p;
  a.center_per_chain = theta0_per_chain;
  a.sigma_in = c->scratch_b.p;
  a.sigma_per_chain = sigma_per_chain;
  a.randomize = randomize_start;
  a.adapt = 0;
  int rc = dispatch_model<LaunchStart>(c->model.model, c, a);
  if (rc) return rc;
  c->stage += 1;
  rc = check_start_flags(c);
  if (rc) return rc;
  c->mh_ready = true;
  return 0;
}

}  // extern "C"

namespace mi355 {
// Whether chains started by mi355_mh_init_problem on the SED catalogue can run in the fused kernel
// (its extended instantiation): parameters within the block-mask tables, LDS for the band tables plus
// the per-lane sqrt matrix.  (Sources outside the fast kernels' domain run in its per-lane instantiation.)
bool sed_mh_fused_ok(const mi355_ctx* c) {
  if (!c || !c->sed_loaded || c->P > kMaxParam) return false;
  return sed_plan(c, true).bs > 0;
}
}  // namespace mi355

extern "C" {

int mi355_mh_run(mi355_ctx* c, int64_t nsteps, int32_t thin, const double* inj_normals,
                 const double* inj_uniforms) {
  MI355_REQUIRE(c && c->mh_ready, "mi355_mh_run: chains not initialised (call mi355_mh_init)");
  if (gmh_active(c)) return gmh_run(c, nsteps, thin, inj_normals, inj_uniforms);
  return sed_mh_run(c, nsteps, thin, inj_normals, inj_uniforms, 0, nullptr, nullptr);
}

}  // extern "C"

namespace mi355 {
// MC_append for every chain in one launch of the fused SED kernel; mode / L / comom: the proposal
// and co-moment state of chains the step-wise engine started (0 / null for mi355_mh_init chains).
int sed_mh_run(mi355_ctx* c, int64_t nsteps, int32_t thin, const double* inj_normals,
               const double* inj_uniforms, int mode, const double* L, double* comom) {
  MI355_REQUIRE(nsteps >= 0 && thin >= 0, "negative nsteps/thin");
  MI355_REQUIRE((inj_normals == nullptr) == (inj_uniforms == nullptr),
                "inj_normals and inj_uniforms must be given together");
  if (nsteps == 0) return 0;
  MI355_HIP(hipSetDevice(c->device));
  const int P = c->mhP;
  MhArgs a = base_args(c);
  a.nsteps = nsteps;
  a.thin = thin;
  a.mode = mode;
  a.L = L;
  a.comom = comom;
  if (inj_normals) {
    const size_t nn = static_cast<size_t>(nsteps) * c->N * P, nu = static_cast<size_t>(nsteps) * c->N;
    if (c->scratch_a.alloc(nn) || c->scratch_b.alloc(nu)) return -2;
    MI355_HIP(hipMemcpyAsync(c->scratch_a.p, inj_normals, nn * sizeof(double), hipMemcpyHostToDevice,
                             c->stream));
    MI355_HIP(hipMemcpyAsync(c->scratch_b.p, inj_uniforms, nu * sizeof(double),
                             hipMemcpyHostToDevice, c->stream));
    a.inj_n = c->scratch_a.p;
    a.inj_u = c->scratch_b.p;
  }
  c->nkeep = thin > 0 ? nsteps / thin : 0;
  if (c->nkeep > 0) {
    const size_t rows = static_cast<size_t>(c->nkeep) * c->N;
    if (c->tr_theta.alloc(rows * P) || c->tr_lnPr.alloc(rows) || c->tr_acc.alloc(rows)) return -2;
    a.tr_theta = c->tr_theta.p;
    a.tr_lnPr = c->tr_lnPr.p;
    a.tr_acc = c->tr_acc.p;
  } else {
    a.thin = 0;
  }
  int rc = dispatch_model<LaunchRun>(c->model.model, c, a);
  if (rc) return rc;
  c->step_total += nsteps;
  c->stage_step += nsteps;
  c->nsamp += nsteps;
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
}  // namespace mi355

extern "C" {

int mi355_mh_set_lane_map(mi355_ctx* c, int32_t mode) {
  MI355_REQUIRE(c, "null context");
  MI355_REQUIRE(mode >= 0 && mode <= 2, "lane map %d: 0 automatic, 1 lane = chain, 2 lanes = (chain, band)", mode);
  c->lane_map = mode;
  return 0;
}

int mi355_mh_lane_group(mi355_ctx* c) {
  if (!c || !c->mh_ready || gmh_active(c)) return -1;
  if (incremental_applies(c)) return 1;
  return sed_plan(c, false).G;
}

int64_t mi355_sed_fast_sources(mi355_ctx* c) {
  if (!c || !c->sed_loaded) return -1;
  return c->S - c->nslow;
}

int mi355_mh_set_incremental(mi355_ctx* c, int32_t on) {
  MI355_REQUIRE(c, "null context");
  c->incremental = on != 0;
  return 0;
}

int mi355_mh_incremental_active(mi355_ctx* c) { return (c && incremental_applies(c)) ? 1 : 0; }

int64_t mi355_mh_stream_pos(mi355_ctx* c, int32_t which) {
  if (!c) return -1;
  return which == 0 ? c->step_total : c->stage;
}

int mi355_mh_set_stream_pos(mi355_ctx* c, int64_t step, int64_t stage) {
  MI355_REQUIRE(c, "null context");
  MI355_REQUIRE(step >= 0 && stage >= 0 && stage < (1 << 23), "stream position out of range");
  c->step_total = step;
  c->stage = stage;
  return 0;
}

int64_t mi355_mh_nsamples(mi355_ctx* c) { return c ? c->nsamp : -1; }
int64_t mi355_mh_ntrace(mi355_ctx* c) { return c ? c->nkeep : -1; }

// device [P][N] -> host [N][P]
static int download_param(mi355_ctx* c, const double* dev, double* host, double scale) {
  const int P = c->mhP;
  const size_t n = static_cast<size_t>(c->N) * P;
  std::vector<double> tmp(n);
  MI355_HIP(hipMemcpyAsync(tmp.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < P; ++i)
    for (int64_t k = 0; k < c->N; ++k) host[k * P + i] = tmp[i * c->N + k] * scale;
  return 0;
}

int mi355_mh_moments(mi355_ctx* c, double* mean, double* var, int64_t* naccept, int64_t* accept,
                     double* lnPr, double* theta) {
  MI355_REQUIRE(c && c->mh_ready, "mi355_mh_moments: chains not initialised");
  MI355_HIP(hipSetDevice(c->device));
  if (mean && download_param(c, c->mean.p, mean, 1.0)) return -2;
  if (var) {
    MI355_REQUIRE(c->nsamp > 0, "mi355_mh_moments: no samples accumulated yet");
    if (download_param(c, c->m2.p, var, 1.0 / static_cast<double>(c->nsamp))) return -2;
  }
  if (theta && download_param(c, c->theta.p, theta, 1.0)) return -2;
  if (lnPr) {
    MI355_HIP(hipMemcpyAsync(lnPr, c->lnPr.p, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MI355_HIP(hipStreamSynchronize(c->stream));
  }
  if (naccept) {
    std::vector<uint32_t> t(static_cast<size_t>(c->N));
    MI355_HIP(hipMemcpyAsync(t.data(), c->naccept.p, t.size() * 4, hipMemcpyDeviceToHost, c->stream));
    MI355_HIP(hipStreamSynchronize(c->stream));
    for (int64_t k = 0; k < c->N; ++k) naccept[k] = t[k];
  }
  if (accept && c->nblock > 0) {
    std::vector<uint32_t> t(static_cast<size_t>(c->N) * c->nblock);
    MI355_HIP(hipMemcpyAsync(t.data(), c->accept.p, t.size() * 4, hipMemcpyDeviceToHost, c->stream));
    MI355_HIP(hipStreamSynchronize(c->stream));
    for (int b = 0; b < c->nblock; ++b)
      for (int64_t k = 0; k < c->N; ++k) accept[k * c->nblock + b] = t[b * c->N + k];
  }
  return 0;
}

int mi355_mh_sigmas(mi355_ctx* c, double* sigma) {
  MI355_REQUIRE(c && c->mh_ready && sigma, "mi355_mh_sigmas: chains not initialised");
  MI355_HIP(hipSetDevice(c->device));
  return download_param(c, c->sigma.p, sigma, 1.0);
}

int mi355_mh_trace(mi355_ctx* c, double* samples, double* lnPr, uint8_t* accepted) {
  MI355_REQUIRE(c && c->mh_ready, "mi355_mh_trace: chains not initialised");
  MI355_REQUIRE(c->nkeep > 0, "mi355_mh_trace: the last mi355_mh_run kept no trace (thin=0)");
  MI355_HIP(hipSetDevice(c->device));
  const int P = c->mhP;
  const size_t rows = static_cast<size_t>(c->nkeep) * c->N;
  if (samples) {
    std::vector<double> tmp(rows * P);
    MI355_HIP(hipMemcpyAsync(tmp.data(), c->tr_theta.p, tmp.size() * sizeof(double),
                             hipMemcpyDeviceToHost, c->stream));
    MI355_HIP(hipStreamSynchronize(c->stream));
    for (int64_t r = 0; r < c->nkeep; ++r)
      for (int i = 0; i < P; ++i)
        for (int64_t k = 0; k < c->N; ++k)
          samples[(r * c->N + k) * P + i] = tmp[(r * P + i) * c->N + k];
  }
  if (lnPr)
    MI355_HIP(hipMemcpyAsync(lnPr, c->tr_lnPr.p, rows * sizeof(double), hipMemcpyDeviceToHost,
                             c->stream));
  if (accepted)
    MI355_HIP(hipMemcpyAsync(accepted, c->tr_acc.p, rows, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_mh_adapt(mi355_ctx* c, double fac, int32_t random_restart) {
  MI355_REQUIRE(c && c->mh_ready, "mi355_mh_adapt: chains not initialised");
  MI355_REQUIRE(c->nsamp > 0, "mi355_mh_adapt: no samples to adapt from");
  if (gmh_active(c)) return gmh_adapt(c, fac, random_restart, 1, 0, -1);
  MI355_HIP(hipSetDevice(c->device));
  if (!(fac > 0.0)) fac = 2.4 / std::sqrt(static_cast<double>(c->mhP));   // MCMC.py:379-380
  MhArgs a = base_args(c);
  a.adapt = 1;
  a.fac = fac;
  a.random_restart = random_restart;
  a.randomize = 1;
  int rc = dispatch_model<LaunchStart>(c->model.model, c, a);
  if (rc) return rc;
  c->stage += 1;
  c->stage_step = 0;
  c->nsamp = 0;
  c->nkeep = 0;
  return check_start_flags(c);
}

int mi355_rng_dump(mi355_ctx* c, int64_t step0, int64_t nsteps, int64_t N, int32_t npn,
                   double* normals, double* uniforms) {
  MI355_REQUIRE(c && normals && uniforms, "mi355_rng_dump: null argument");
  MI355_REQUIRE(nsteps > 0 && N > 0 && npn > 0, "empty request");
  MI355_HIP(hipSetDevice(c->device));
  const size_t n = static_cast<size_t>(nsteps) * N;
  if (c->scratch_a.alloc(n * npn) || c->scratch_b.alloc(n)) return -2;
  hipLaunchKernelGGL(k_rng_dump, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0,
                     c->stream, c->seed, step0, nsteps, N, npn, c->scratch_a.p, c->scratch_b.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(normals, c->scratch_a.p, n * npn * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
  MI355_HIP(hipMemcpyAsync(uniforms, c->scratch_b.p, n * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_gauss_quadform(mi355_ctx* c, int64_t m, int64_t n, const double* A, const double* B,
                         int64_t b_rows, const double* sig2, int64_t nsig, double* out) {
  MI355_REQUIRE(c && A && sig2 && out, "mi355_gauss_quadform: null argument");
  MI355_REQUIRE(m > 0 && n > 0, "mi355_gauss_quadform: empty input");
  MI355_REQUIRE(nsig == n || nsig == 1, "DataSizeError: %lld data points but %lld sigmas "
                "[GaussianData.py:21-26]", static_cast<long long>(n), static_cast<long long>(nsig));
  MI355_REQUIRE(!B || b_rows == m || b_rows == 1, "mi355_gauss_quadform: b_rows must be m or 1");
  MI355_HIP(hipSetDevice(c->device));
  const size_t mn = static_cast<size_t>(m) * n;
  const size_t bn = B ? static_cast<size_t>(b_rows) * n : 0;
  if (c->scratch_a.alloc(mn + bn + nsig) || c->scratch_b.alloc(static_cast<size_t>(m))) return -2;
  double* dA = c->scratch_a.p;
  double* dB = B ? dA + mn : nullptr;
  double* dS = dA + mn + bn;
  MI355_HIP(hipMemcpyAsync(dA, A, mn * 8, hipMemcpyHostToDevice, c->stream));
  if (B) MI355_HIP(hipMemcpyAsync(dB, B, bn * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(dS, sig2, static_cast<size_t>(nsig) * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_gauss_quadform, dim3(static_cast<unsigned>(m)), dim3(kBlockThreads), 0,
                     c->stream, n, dA, dB, static_cast<int64_t>(b_rows == 1 ? 0 : n), dS,
                     nsig == 1 ? 1 : 0, c->scratch_b.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(out, c->scratch_b.p, static_cast<size_t>(m) * 8, hipMemcpyDeviceToHost,
                           c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_math_probe(mi355_ctx* c, int32_t fn, int64_t n, const double* a, const double* b,
                     const double* cc, double* out) {
  MI355_REQUIRE(c && a && out && n > 0, "mi355_math_probe: bad argument");
  MI355_REQUIRE(fn >= 0 && fn <= 15, "mi355_math_probe: unknown function %d", fn);
  MI355_REQUIRE((fn != 2 && fn != 3) || b, "mi355_math_probe: b required");
  MI355_REQUIRE(fn != 2 || cc, "mi355_math_probe: c required");
  MI355_HIP(hipSetDevice(c->device));
  const size_t nn = static_cast<size_t>(n);
  if (c->scratch_a.alloc(nn * 3) || c->scratch_b.alloc(nn)) return -2;
  double* da = c->scratch_a.p;
  double* db = da + nn;
  double* dc = db + nn;
  MI355_HIP(hipMemcpyAsync(da, a, nn * 8, hipMemcpyHostToDevice, c->stream));
  if (b) MI355_HIP(hipMemcpyAsync(db, b, nn * 8, hipMemcpyHostToDevice, c->stream));
  if (cc) MI355_HIP(hipMemcpyAsync(dc, cc, nn * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_math_probe, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0,
                     c->stream, fn, n, da, db, dc, c->scratch_b.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(out, c->scratch_b.p, nn * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_rhat_partial_async(mi355_ctx* c, int32_t pass, const double* gmean_dev, double* sums_dev) {
  MI355_REQUIRE(c && c->mh_ready, "mi355_rhat: chains not initialised");
  MI355_REQUIRE(c->nsamp > 0, "mi355_rhat: no samples accumulated yet");
  MI355_REQUIRE(sums_dev && (pass == 0 || gmean_dev), "mi355_rhat_partial: null buffer");
  MI355_HIP(hipSetDevice(c->device));
  const int lps = c->C <= 16 ? 16 : (c->C <= 32 ? 32 : 64);   // lanes per source
  const int64_t waves = ceil_div(c->mhS, kRhatU * (64 / lps)) * c->mhP;
  const dim3 grid(static_cast<unsigned>(ceil_div(waves * 64, kBlockThreads)));
  const double rn = 1.0 / static_cast<double>(c->nsamp);
  if (lps == 16)
    hipLaunchKernelGGL(k_rhat_partial<16>, grid, dim3(kBlockThreads), 0, c->stream, pass, c->mhP, c->mhS, c->C, c->N,
                       rn, c->mean.p, c->m2.p, gmean_dev, sums_dev);
  else if (lps == 32)
    hipLaunchKernelGGL(k_rhat_partial<32>, grid, dim3(kBlockThreads), 0, c->stream, pass, c->mhP, c->mhS, c->C, c->N,
                       rn, c->mean.p, c->m2.p, gmean_dev, sums_dev);
  else
    hipLaunchKernelGGL(k_rhat_partial<64>, grid, dim3(kBlockThreads), 0, c->stream, pass, c->mhP, c->mhS, c->C, c->N,
                       rn, c->mean.p, c->m2.p, gmean_dev, sums_dev);
  MI355_HIP(hipGetLastError());
  return 0;
}

int mi355_rhat_finish_async(mi355_ctx* c, const double* sums_dev, int64_t Ctot, double* rhat_dev) {
  MI355_REQUIRE(c && c->mh_ready && sums_dev && rhat_dev, "mi355_rhat_finish: bad argument");
  MI355_REQUIRE(Ctot >= 2, "Gelman-Rubin needs at least 2 chains per source (got %lld)",
                static_cast<long long>(Ctot));
  MI355_HIP(hipSetDevice(c->device));
  const int64_t n = c->mhS * c->mhP;
  hipLaunchKernelGGL(k_rhat_finish, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0,
                     c->stream, n, static_cast<double>(Ctot), sums_dev, rhat_dev);
  MI355_HIP(hipGetLastError());
  return 0;
}

int mi355_gelman_rubin(mi355_ctx* c, int32_t nchain, int32_t P, const double* means,
                       const double* vars, double* rhat) {
  MI355_REQUIRE(c && means && vars && rhat, "mi355_gelman_rubin: null argument");
  MI355_REQUIRE(nchain >= 2 && P > 0, "Gelman-Rubin needs at least 2 chains (got %d)", nchain);
  MI355_HIP(hipSetDevice(c->device));
  const size_t n = static_cast<size_t>(nchain) * P;
  if (c->scratch_a.alloc(2 * n + P)) return -2;
  double* dm = c->scratch_a.p;
  double* dv = dm + n;
  double* dr = dv + n;
  MI355_HIP(hipMemcpyAsync(dm, means, n * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(dv, vars, n * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_gelman_rubin, dim3((P + 63) / 64), dim3(64), 0, c->stream, nchain, P, dm, dv, dr);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(rhat, dr, static_cast<size_t>(P) * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_mh_ess(mi355_ctx* c, double* ess) {
  MI355_REQUIRE(c && c->mh_ready && ess, "mi355_mh_ess: bad argument");
  MI355_REQUIRE(c->nkeep >= 4, "mi355_mh_ess: the last mi355_mh_run kept %lld trace rows (need >= 4: "
                "run with thin > 0)", static_cast<long long>(c->nkeep));
  MI355_REQUIRE(c->nkeep <= kEssMaxRows, "mi355_mh_ess: %lld trace rows, limit %d (raise thin)",
                static_cast<long long>(c->nkeep), kEssMaxRows);
  MI355_REQUIRE(c->C <= 256, "mi355_mh_ess: at most 256 chains per group (got %d)", c->C);
  MI355_HIP(hipSetDevice(c->device));
  const int P = c->mhP;
  const int64_t n = c->mhS * P;
  if (c->scratch_a.alloc(static_cast<size_t>(n))) return -2;
  int ntc_shift = 6;                                      // chain slots per block: 64, 128 or 256
  while ((1 << ntc_shift) < c->C) ++ntc_shift;
  const int nt = 1 << ntc_shift;
  const size_t lds = (static_cast<size_t>(c->nkeep) * nt + 4 * kEssLags) * sizeof(double);
  static size_t opted[kMaxDevices] = {};   // per device: the attribute belongs to the device's code object
  size_t& have = opted[c->device % kMaxDevices];
  if (lds > 48 * 1024 && lds > have) {
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ess),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    have = lds;
  }
  MI355_REQUIRE(lds <= 150 * 1024, "mi355_mh_ess: trace tile (%zu B) does not fit in LDS", lds);
  hipLaunchKernelGGL(k_ess, dim3(static_cast<unsigned>(n)), dim3(256), lds, c->stream, P, c->mhS, c->C,
                     c->N, static_cast<int>(c->nkeep), ntc_shift, c->tr_theta.p, c->scratch_a.p);
  MI355_HIP(hipGetLastError());
  std::vector<double> h(static_cast<size_t>(n));
  MI355_HIP(hipMemcpyAsync(h.data(), c->scratch_a.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  for (int p = 0; p < P; ++p)
    for (int64_t s = 0; s < c->mhS; ++s) ess[s * P + p] = h[p * c->mhS + s];
  return 0;
}

int mi355_rhat_noalloc_async(mi355_ctx* c, double* rhat_dev) {
  MI355_REQUIRE(c && c->mh_ready && rhat_dev, "mi355_rhat_noalloc_async: bad argument");
  const int64_t n = c->mhS * c->mhP;
  MI355_REQUIRE(n <= MI355_RHAT_NOALLOC_MAX,
                "mi355_rhat_noalloc_async: %lld sources x %d parameters = %lld entries, the fixed "
                "scratch holds %lld (use mi355_rhat, which allocates)", static_cast<long long>(c->mhS),
                c->mhP, static_cast<long long>(n), static_cast<long long>(MI355_RHAT_NOALLOC_MAX));
  MI355_HIP(hipSetDevice(c->device));
  // every entry of gmean / sums is written by k_rhat_partial before it is read: nothing in the
  // array has to start at zero, whatever the previous call left there
  double* gmean = c->fixed_scratch;
  double* sums = c->fixed_scratch + MI355_RHAT_NOALLOC_MAX;
  int rc = mi355_rhat_partial_async(c, 0, nullptr, gmean);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scale, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0, c->stream,
                     n, 1.0 / static_cast<double>(c->C), gmean);
  MI355_HIP(hipGetLastError());
  rc = mi355_rhat_partial_async(c, 1, gmean, sums);
  if (rc) return rc;
  return mi355_rhat_finish_async(c, sums, c->C, rhat_dev);
}

// ncclAllReduce of the RCCL this process runs with (see the header): bound once
typedef int (*rccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static rccl_allreduce_fn rccl_allreduce() {
  static rccl_allreduce_fn fn = [] {
    void* s = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!s) {
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) s = dlsym(h, "ncclAllReduce");
    }
    return reinterpret_cast<rccl_allreduce_fn>(s);
  }();
  return fn;
}

int mi355_rhat_allreduce_async(mi355_ctx* c, void* rccl_comm, int64_t Ctot, double* rhat_dev) {
  MI355_REQUIRE(c && c->mh_ready && rhat_dev, "mi355_rhat_allreduce: bad argument");
  // (a null communicator: this rank holds every chain - the same kernels, no collective)
  MI355_REQUIRE(rccl_comm || Ctot == c->C, "mi355_rhat_allreduce: null communicator with Ctot=%lld != the %d chains of this rank",
                static_cast<long long>(Ctot), c->C);
  MI355_REQUIRE(Ctot >= 2 && Ctot >= c->C, "mi355_rhat_allreduce: Ctot=%lld chains per source over all ranks (this rank holds %d)",
                static_cast<long long>(Ctot), c->C);
  const rccl_allreduce_fn allreduce = rccl_comm ? rccl_allreduce() : nullptr;
  MI355_REQUIRE(!rccl_comm || allreduce, "mi355_rhat_allreduce: no RCCL in this process (ncclAllReduce not found, librccl.so not loadable)");
  MI355_HIP(hipSetDevice(c->device));
  const int64_t n = c->mhS * c->mhP;
  if (c->scratch_a.alloc(static_cast<size_t>(n) * 3)) return -2;
  double* gmean = c->scratch_a.p;
  double* sums = c->scratch_a.p + n;
  constexpr int kNcclFloat64 = 8, kNcclSum = 0;            // rccl.h: ncclFloat64, ncclSum
  int rc = mi355_rhat_partial_async(c, 0, nullptr, gmean);
  if (rc) return rc;
  int nrc = rccl_comm ? allreduce(gmean, gmean, static_cast<size_t>(n), kNcclFloat64, kNcclSum, rccl_comm, c->stream) : 0;
  MI355_REQUIRE(nrc == 0, "mi355_rhat_allreduce: ncclAllReduce failed with ncclResult_t %d", nrc);
  hipLaunchKernelGGL(k_scale, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0, c->stream,
                     n, 1.0 / static_cast<double>(Ctot), gmean);
  MI355_HIP(hipGetLastError());
  rc = mi355_rhat_partial_async(c, 1, gmean, sums);
  if (rc) return rc;
  if (rccl_comm) nrc = allreduce(sums, sums, static_cast<size_t>(2 * n), kNcclFloat64, kNcclSum, rccl_comm, c->stream);
  MI355_REQUIRE(nrc == 0, "mi355_rhat_allreduce: ncclAllReduce failed with ncclResult_t %d", nrc);
  return mi355_rhat_finish_async(c, sums, Ctot, rhat_dev);
}

int mi355_rhat(mi355_ctx* c, double* rhat) {
  MI355_REQUIRE(c && c->mh_ready && rhat, "mi355_rhat: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  const int64_t n = c->mhS * c->mhP;
  if (c->scratch_a.alloc(static_cast<size_t>(n) * 4)) return -2;
  double* gmean = c->scratch_a.p;
  double* sums = c->scratch_a.p + n;
  double* out = c->scratch_a.p + 3 * n;
  int rc = mi355_rhat_partial_async(c, 0, nullptr, gmean);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scale, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0, c->stream,
                     n, 1.0 / static_cast<double>(c->C), gmean);
  MI355_HIP(hipGetLastError());
  rc = mi355_rhat_partial_async(c, 1, gmean, sums);
  if (rc) return rc;
  rc = mi355_rhat_finish_async(c, sums, c->C, out);
  if (rc) return rc;
  std::vector<double> h(static_cast<size_t>(n));
  MI355_HIP(hipMemcpyAsync(h.data(), out, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  for (int p = 0; p < c->mhP; ++p)
    for (int64_t s = 0; s < c->mhS; ++s) rhat[s * c->mhP + p] = h[p * c->mhS + s];
  return 0;
}

}  // extern "C"
