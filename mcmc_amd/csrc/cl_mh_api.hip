// Host side of the fused binnedCl Metropolis-Hastings samplers (cl_mh_kernels.hpp: a block per 16 chains;
// cl_mhw_kernels.hpp: a wave per 16 chains): plans, launchers, cl_mh_run.  Its own translation unit so that
// it can be compiled without machine-level loop-invariant code motion (Makefile: FLAGS_cl_mh_api) - hoisting
// the step loop's invariants costs these kernels registers they spill: 11-point grid 24.8 -> 22.6 us per
// step of 4096 chains, block kernel 6.55 -> 6.39, nb = 256 15.7 -> 15.4 (one box, round 3) - while the
// batched k_cl_form in cl_api.hip is 1.5 % faster WITH it.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "cl_mhw_kernels.hpp"
#include "cl_problem.hpp"
#include "problems.hpp"

namespace mi355 {

void cl_plan_fused(ClProblem* p) {
  p->mh_nw = 0;
  p->mh_filled = false;
  if (p->data.empty() || p->data.size() > static_cast<size_t>(kClMhMaxData)) return;
  int nw = 0, nbp_max = 0, fout = 9;
  bool block_kernel = true;                        // k_cl_mh_run serves the problem (k_cl_mh_wave: cl_mhw_plan)
  for (size_t i = 0; i < p->data.size(); ++i) {
    ClData* d = p->data[i].get();
    const FormPlan f = plan_for(d);
    if (f.rows_per_chain > kClGrid) block_kernel = false;   // 121 points per chain: not in k_cl_mh_run
    if (d->has_corr) {
      if (nw && nw != form_nw(d->nbp)) block_kernel = false;   // one block size serves every data set
      nw = form_nw(d->nbp);
    }
    if (d->nbp > nbp_max) nbp_max = d->nbp;
    if (f.rows_per_chain > fout) fout = f.rows_per_chain;
    ClMhData& m = p->mh[i];
    m = ClMhData{};
    m.f.nb = d->nb;
    m.f.nbp = d->nbp;
    m.f.obs = d->obs.p;
    m.f.xf = d->xf.p;
    m.f.hasx = d->hasx.p;
    m.f.beam_err = d->beam_err.p;
    m.f.bfac = d->bfac.p;
    m.f.ninv = d->ninv.p;
    m.f.nsym = d->nsym.p;
    m.f.nstride = cl_gstride(d->nbp);
    m.f.var = d->var.p;
    m.f.has_x = d->has_x ? 1 : 0;
    m.f.beam_uncertain = d->beam_uncertain ? 1 : 0;
    m.f.calib = d->calib;
    m.f.mode = f.mode;
    m.f.ncal = f.ncal;
    m.f.nbeam = f.nbeam;
    m.K = d->K.p;
    m.kstride = cl_gstride(d->nbp);
    m.krange = d->krange.p;
    m.kmax = d->kmax;
    m.bp0 = d->bp0.p;
    m.has_corr = d->has_corr ? 1 : 0;
    m.rows_per_chain = f.rows_per_chain;
    m.sym = (d->has_corr && f.mode != CL_ROWS_TRIPLE && (d->nbp >> 4) >= 8) ? 1 : 0;
  }
  if (!nw) nw = 4;
  p->mh_filled = true;
  p->mh_nbp_max = nbp_max;
  p->mh_fout = fout;
  p->mh_tile_doubles = static_cast<int>((form_lds_bytes(nbp_max, 16, nw) + 7) / 8);
  p->mh_nw = block_kernel ? nw : 0;
}

// Plan of the wave-per-16-chains kernel (cl_mhw_kernels.hpp), which takes over from k_cl_mh_run where
// it applies: one data set with correlated errors and a MARGE form (any marginalisation grid, the
// 11 x 11 one included), parameters and bands within its register budget.  Waves per block: as few as
// give every CU a block (a wave never waits for another one, so small jobs spread over the CUs),
// up to kClMhwMaxWaves; N^-1 resident in LDS when it fits beside the waves' own regions.
struct ClMhwPlan {
  int ok = 0, nct_t = 0, ms = 0, sym = 0, res = 0, waves = 1, rep = 1, sb = 1;
  size_t lds = 0;
};
static bool cl_mh_block_ok(const ClProblem* p, int P, int nblock, int npn);   // k_cl_mh_run serves the problem
static ClMhwPlan cl_mhw_plan(const ClProblem* p, int P, int nblock, int npn, int64_t N, int comom = 1, bool force = false) {
  ClMhwPlan r;
  static const int env = [] { const char* e = getenv("MI355_CLMH_WAVE"); return e ? atoi(e) : -1; }();
  if (env == 0 || !p->mh_filled || p->data.size() != 1) return r;   // (MI355_CLMH_WAVE: 0 never, 1 wherever it applies)
  // (force: mi355_mh_set_fused(ctx, 2) - the 121-point grid through this kernel whatever the job size)
  const int mode = (env == 1) ? 1 : -1;
  const ClData* d = p->data[0].get();
  const FormPlan f = plan_for(d);
  const int nct = d->nbp >> 4;
  if (!d->has_corr || f.mode != CL_ROWS_MARGE || P > 4 * kClMhwSlots || nct > 8) return r;
  constexpr int kCUs = 256;
  const int64_t tiles = ceil_div64(N, kClMhChains);        // 16-chain samplers
  const bool grid = f.rows_per_chain > 1;
  // Small jobs without a grid: a wave alone on its SIMD walks a step's whole latency chain, and the
  // block-per-16-chains kernel, which shares a step out over four waves, is faster (6.5 against 10.7 us
  // per step at 4096 chains, nb = 64; the wave kernel wins from 0.75 waves per SIMD up: 11.5 against
  // 17.0 us at 16 384 chains, 31 against 65 at 65 536).  With a grid the replicas fill the SIMDs.
  if (mode != 1 && !grid && tiles < 3 * kCUs && cl_mh_block_ok(p, P, nblock, npn)) return r;
  // The full 11 x 11 grid: a step is 121 forms, throughput work once the chip is full, and the launch-per-
  // step engine's form kernel is the faster one from there (per step, fused against step-wise: 188 / 197 us
  // at 2048 chains, 197 / 259 at 4096, 434 / 473 at 8192, 915 / 803 at 16 384).  The 11-point grids stay
  // fused at every size (25 / 56 us at 4096 chains, 73 / 133 at 16 384, 258 / 364 at 65 536).
  // Below 4096 chains the eight replica waves of a 16-chain tile - one block, one CU - walk 15 forms each
  // on as many CUs as the job has tiles, where the batched form kernel spreads chains x 121 rows over the chip:
  // 188 / 116 us at 512 chains, 189 / 137 at 1024, 191 / 188 at 2048 (with co-moments 222 / 113 ... 223 / 189).
  if (mode != 1 && !force && f.rows_per_chain > kClGrid && (tiles > 2 * kCUs || tiles <= kCUs / 2)) return r;
  // (The running co-moments - every adaptive stage of sampler() - are updated in memory by this kernel, in
  // registers by the block kernel's resident instantiations: gmh_comom_update.  us per step with 32 parameters,
  // wave / block: 24.5 / 24.8 at 8192 chains, 42.8 / 41.0 at 16 384, 56 / 61 at 24 576, 65 / 80 at 32 768;
  // 11-point grid 30 / 46 at 4096 chains, 86 / 130 at 16 384: no rule of their own.)
  r.ms = kClMhwSlots;
  r.nct_t = nct <= 4 ? 4 : 8;
  r.sym = nct >= 8 ? 1 : 0;
  r.sb = cl_mhw_cache_steps(npn);
  if (grid)
    while (r.rep < kClMhwMaxWaves && tiles * r.rep * 2 <= 8 * kCUs && r.rep * 2 <= f.rows_per_chain) r.rep *= 2;   // up to two waves per SIMD
  const size_t per_wave = cl_mhw_wave_doubles(P, nblock, npn, r.sb, comom, grid);
  const size_t xchg = grid ? cl_mhw_xchg_doubles(f.rows_per_chain) : 0;
  // (fewer chains per block, then fewer replicas per chain, before giving up)
  for (; r.rep >= 1; r.rep >>= 1) {
    int tpb = static_cast<int>(std::min<int64_t>(kClMhwMaxWaves / r.rep, std::max<int64_t>(1, ceil_div64(tiles, kCUs))));
    for (; tpb >= 1; --tpb)
      for (int res = 1; res >= 0; --res) {
        if (res && r.nct_t > 4) continue;          // (the resident copy has the 4-tile kernels' compile-time stride)
        const size_t lds = (cl_mhw_shared_doubles(P, r.nct_t, nct, d->kmax, f.nbeam, res) +
                            static_cast<size_t>(tpb) * (r.rep * per_wave + xchg)) * 8;
        if (lds <= kClLdsBudget) {
          r.ok = 1;
          r.res = res;
          r.lds = lds;
          r.waves = tpb * r.rep;
          return r;
        }
      }
  }
  r.rep = 1;
  return r;
}

// Plan of a fused run: LDS bytes and whether the data set lives in LDS for the launch.  Resident
// pays when the grid is at most one block per CU (6.9 against 8.9 us per step at 4096 chains,
// nb = 64); beyond that the streamed kernel wins because two of its blocks share a CU and hide
// one another's latencies (5.2 us per 4096 chains at 65536 chains).
// A single data set keeps its band tables and the compact K in LDS whenever they fit (tables = 1: the
// kernels compiled for its form); otherwise, and for several data sets, the generic kernel reads
// everything from L2.
struct ClMhLds {
  size_t bytes;
  int resident, tables, sb;
};
static ClMhLds cl_mh_plan_lds(const ClProblem* p, int P, int nblock, int npn, int64_t N) {
  const int sb = cl_mh_cache_steps(64 * p->mh_nw, npn);
  ClMhLds r{0, 0, 0, sb};
  constexpr int kCUs = 256;
  auto bytes = [&](size_t extra) {
    return cl_mh_lds_doubles(P, nblock, npn, sb, p->mh_nbp_max, p->mh_fout, p->mh_tile_doubles, extra) * 8;
  };
  if (p->data.size() == 1) {
    const ClMhData& m = p->mh[0];
    const size_t tab = cl_mh_table_doubles(m.f.nbp, m.kmax, m.f.nbeam);
    if (bytes(tab) <= kClLdsBudget) {
      r.tables = 1;
      const size_t mat = cl_mh_matrix_doubles(m.f.nbp, m.has_corr);
      if (ceil_div64(N, kClMhChains) <= kCUs && bytes(tab + mat) <= kClLdsBudget) {
        r.resident = 1;
        r.bytes = bytes(tab + mat);
        return r;
      }
      r.bytes = bytes(tab);
      return r;
    }
  }
  r.bytes = bytes(0);
  return r;
}

static bool cl_mh_block_ok(const ClProblem* p, int P, int nblock, int npn) {
  if (!p->mh_nw) return false;
  if (P > kClMhSlots * (64 * p->mh_nw / kClMhChains)) return false;   // parameters per thread in registers
  return cl_mh_plan_lds(p, P, nblock, npn, 1 << 30).bytes <= kClLdsBudget;
}

int cl_mh_fused_nw(const mi355_ctx* c, int P, int nblock, int npn, int comom) {
  if (!c || !c->cl || !c->cl->binned) return 0;
  // (the wave-per-16-chains kernel: its policy depends on the number of chains of the job)
  if (cl_mhw_plan(c->cl, P, nblock, npn, c->N > 0 ? c->N : int64_t{1} << 30, comom, c->mh_fused == 2).ok) return 1;
  return cl_mh_block_ok(c->cl, P, nblock, npn) ? c->cl->mh_nw : 0;
}

template <int NW, int FORM, bool RES, bool CMR = false>
static int launch_cl_mh(mi355_ctx* c, const ClMhArgs& a, size_t lds) {
  static size_t opted[kMaxDevices] = {};
  size_t& have = opted[c->device % kMaxDevices];
  if (lds > 48 * 1024 && lds > have) {
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cl_mh_run<NW, FORM, RES, CMR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    have = lds;
  }
  ClProblem* p = c->cl;
  if (p->mh_args.alloc(1)) return -2;
  // synchronous copy from this stack frame; the stream is idle here (mi355_mh_run is synchronous)
  MI355_HIP(hipMemcpyAsync(p->mh_args.p, &a, sizeof(ClMhArgs), hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  const unsigned grid = static_cast<unsigned>(ceil_div64(a.g.N, kClMhChains));
  hipLaunchKernelGGL((k_cl_mh_run<NW, FORM, RES, CMR>), dim3(grid), dim3(64 * NW), lds, c->stream,
                     static_cast<const ClMhArgs*>(p->mh_args.p));
  MI355_HIP(hipGetLastError());
  return 0;
}

// One data set: kernels compiled for its form, with its matrices in LDS (resident) or streamed
// from L2; several data sets: the generic kernel.
template <int NW>
static int launch_cl_mh_form(mi355_ctx* c, const ClMhArgs& a, size_t lds) {
  if (a.ndata != 1 || !a.tables) return launch_cl_mh<NW, CLMH_ANY, false>(c, a, lds);
  const bool res = a.resident != 0;
  if constexpr (NW == 4) {
    // resident 4-wave kernels: the running co-moments in registers when the triangle fits
    if (res && a.g.comom != nullptr && a.g.pb.P <= kClMhCmrMaxP) {
      switch (cl_mh_form_kind(a.data[0])) {
        case CLMH_DIAG: return launch_cl_mh<4, CLMH_DIAG, true, true>(c, a, lds);
        case CLMH_TRIPLE: return launch_cl_mh<4, CLMH_TRIPLE, true, true>(c, a, lds);
        case CLMH_SYM: break;
        default: return launch_cl_mh<4, CLMH_MARGE, true, true>(c, a, lds);
      }
    }
  }
  switch (cl_mh_form_kind(a.data[0])) {
    case CLMH_DIAG:
      return res ? launch_cl_mh<NW, CLMH_DIAG, true>(c, a, lds) : launch_cl_mh<NW, CLMH_DIAG, false>(c, a, lds);
    case CLMH_TRIPLE:
      return res ? launch_cl_mh<NW, CLMH_TRIPLE, true>(c, a, lds)
                 : launch_cl_mh<NW, CLMH_TRIPLE, false>(c, a, lds);
    case CLMH_SYM:
      if constexpr (NW == 8)                               // from 8 column tiles up
        return res ? launch_cl_mh<8, CLMH_SYM, true>(c, a, lds) : launch_cl_mh<8, CLMH_SYM, false>(c, a, lds);
      else
        break;
    default:
      return res ? launch_cl_mh<NW, CLMH_MARGE, true>(c, a, lds)
                 : launch_cl_mh<NW, CLMH_MARGE, false>(c, a, lds);
  }
  set_error("cl_mh_run: symmetrised form with a 4-wave block");
  return -1;
}

template <int NCT, int MS, bool SYM, bool RES, bool GRID>
static int launch_cl_mhw(mi355_ctx* c, const ClMhArgs& a, const ClMhwPlan& plan) {
  static size_t opted[kMaxDevices] = {};
  size_t& have = opted[c->device % kMaxDevices];
  if (plan.lds > 48 * 1024 && plan.lds > have) {
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cl_mh_wave<NCT, MS, SYM, RES, GRID>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plan.lds)));
    have = plan.lds;
  }
  ClProblem* p = c->cl;
  if (p->mh_args.alloc(1)) return -2;
  MI355_HIP(hipMemcpyAsync(p->mh_args.p, &a, sizeof(ClMhArgs), hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  const unsigned grid = static_cast<unsigned>(ceil_div64(a.g.N, static_cast<int64_t>(kClMhChains) * (plan.waves / plan.rep)));
  hipLaunchKernelGGL((k_cl_mh_wave<NCT, MS, SYM, RES, GRID>), dim3(grid), dim3(64 * plan.waves), plan.lds, c->stream,
                     static_cast<const ClMhArgs*>(p->mh_args.p));
  MI355_HIP(hipGetLastError());
  return 0;
}

template <int NCT, bool SYM>
static int launch_cl_mhw_form(mi355_ctx* c, const ClMhArgs& a, const ClMhwPlan& plan) {
  constexpr int MS = kClMhwSlots;
  const bool grid = a.data[0].rows_per_chain > 1;
  if constexpr (NCT == 4) {                              // (the resident copy of N^-1: up to 64 bands)
    if (plan.res)
      return grid ? launch_cl_mhw<NCT, MS, SYM, true, true>(c, a, plan) : launch_cl_mhw<NCT, MS, SYM, true, false>(c, a, plan);
  }
  return grid ? launch_cl_mhw<NCT, MS, SYM, false, true>(c, a, plan) : launch_cl_mhw<NCT, MS, SYM, false, false>(c, a, plan);
}

static int cl_mhw_run(mi355_ctx* c, ClMhArgs& a, const ClMhwPlan& plan) {
  a.sb = plan.sb;
  a.rep = plan.rep;
  a.resident = plan.res;
  a.tables = 1;
  if (plan.nct_t == 4) return launch_cl_mhw_form<4, false>(c, a, plan);
  return plan.sym ? launch_cl_mhw_form<8, true>(c, a, plan) : launch_cl_mhw_form<8, false>(c, a, plan);
}

int cl_mh_run(mi355_ctx* c, const GmhArgs& g, int64_t nsteps, int thin, int64_t nkeep) {
  ClProblem* p = c->cl;
  const int nw = cl_mh_fused_nw(c, g.pb.P, g.nblock, g.npn, g.comom != nullptr);
  MI355_REQUIRE(nw > 0, "cl_mh_run: the loaded Cl problem has no fused kernel");
  const ClMhwPlan wplan = cl_mhw_plan(p, g.pb.P, g.nblock, g.npn, g.N, g.comom != nullptr, c->mh_fused == 2);
  if (wplan.ok) {
    ClMhArgs a{};
    a.g = g;
    a.ndata = 1;
    a.data[0] = p->mh[0];
    a.nbp_max = p->mh_nbp_max;
    a.weights = p->weights.p;
    a.margenorm = p->margenorm;
    a.nsteps = nsteps;
    a.thin = thin;
    a.nkeep = nkeep;
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
    const int rc = cl_mhw_run(c, a, wplan);
    if (rc) return rc;
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
    if (!c->capturing) c->ev_valid = true;
    return 0;
  }
  const ClMhLds plan = cl_mh_plan_lds(p, g.pb.P, g.nblock, g.npn, g.N);
  ClMhArgs a{};
  a.g = g;
  a.ndata = static_cast<int>(p->data.size());
  for (int i = 0; i < a.ndata; ++i) a.data[i] = p->mh[i];
  a.nbp_max = p->mh_nbp_max;
  a.fout_per_chain = p->mh_fout;
  a.tile_doubles = p->mh_tile_doubles;
  a.resident = plan.resident;
  a.tables = plan.tables;
  a.sb = plan.sb;
  a.weights = p->weights.p;
  a.margenorm = p->margenorm;
  a.nsteps = nsteps;
  a.thin = thin;
  a.nkeep = nkeep;
  const size_t lds = plan.bytes;
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
  MI355_REQUIRE(p->mh_nw > 0, "cl_mh_run: no fused kernel serves this run");
  const int rc = p->mh_nw == 4 ? launch_cl_mh_form<4>(c, a, lds) : launch_cl_mh_form<8>(c, a, lds);
  if (rc) return rc;
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
  if (!c->capturing) c->ev_valid = true;
  return 0;
}

}  // namespace mi355
