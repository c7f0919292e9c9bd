// Step-wise Metropolis-Hastings over any problem kind: host side of gmh_kernels.hpp and the C ABI
// entry points mi355_mh_init_problem / mi355_mh_set_sqrt_matrix / mi355_mh_adapt_corr /
// mi355_mh_covar / mi355_mh_get_sqrt_matrix.  The shared entry points (mi355_mh_run, _moments,
// _trace, _sigmas, _adapt, mi355_rhat*) in api.hip work on the same chain-state buffers.
#include <cmath>
#include <memory>

#include "gmh_kernels.hpp"
#include "problems.hpp"

namespace mi355 {

struct GmhState {
  bool active = false;
  GmhProblem pb{};
  int mode = GMH_UNCORR;
  bool track_cov = false;           // co-moment buffers exist (mi355_mh_init_problem track_cov=1)
  bool cov_on = false;              // ... and runs update them (mi355_mh_track_cov)
  int64_t cov_nsamp = 0;            // samples of the current stage the co-moments hold
  int nblock_table = 0;             // blocks of the table given at init (doBlock can be toggled)
  DevBuf<double> theta_new, prior_new, lnl_new, zbuf, comom, L, stage_a, stage_b;
  DevBuf<int32_t> block_of, rank_in_block;
  std::vector<int32_t> h_block_of, h_rank;
  int maxblk = 1;
  // hipGraph replay of free-running steps (no trace, no injection)
  DevBuf<GmhDyn> dyn;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int graph_steps = 0;              // MH steps per graph launch
  uint64_t graph_version = 0;       // version the graph was captured for
  uint64_t graph_epoch = 0;         // g_devbuf_epoch at capture: any buffer freed since then may
                                    // be one the graph's kernels point at (Cl / beam scratch grows
                                    // when a host call evaluates a batch larger than N)
  uint64_t version = 1;             // bumped whenever launch arguments may change
  ~GmhState() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
  }
};

constexpr int kGraphSteps = 32;     // MH steps captured per graph (even: two counter slots)
constexpr int64_t kGraphMinSteps = 32;

void gmh_destroy(GmhState* g) { delete g; }
bool gmh_active(const mi355_ctx* c) { return c && c->gmh && c->gmh->active; }
void gmh_deactivate(mi355_ctx* c) {
  if (c->gmh) {
    c->gmh->active = false;
    c->gmh->version += 1;          // a captured graph holds pointers of the old problem
  }
}

static inline unsigned grid_for(int64_t n) { return static_cast<unsigned>(ceil_div64(n, 256)); }

static int problem_lnlike(mi355_ctx* c, const double* theta_pm, double* lnl) {
  GmhState* g = c->gmh;
  switch (g->pb.kind) {
    case GMH_SED:
      return mi355_sed_lnlike_dev_async(c, 0, c->mhS, c->C, theta_pm, lnl, nullptr);
    case GMH_CL:
      return mi355_cl_lnlike_dev_async(c, c->N, theta_pm, lnl);
    default:
      return mi355_beam_lnlike_dev_async(c, 0, c->mhS, c->C, theta_pm, lnl);
  }
}

static GmhArgs base_args(mi355_ctx* c) {
  GmhState* g = c->gmh;
  GmhArgs a{};
  a.pb = g->pb;
  a.N = c->N;
  a.theta = c->theta.p;
  a.sigma = c->sigma.p;
  a.lnPr = c->lnPr.p;
  a.mean = c->mean.p;
  a.m2 = c->m2.p;
  a.comom = g->cov_on ? g->comom.p : nullptr;
  a.naccept = c->naccept.p;
  a.accept = c->accept.p;
  a.flags = c->flags.p;
  a.theta_new = g->theta_new.p;
  a.prior_new = g->prior_new.p;
  a.lnl_new = g->lnl_new.p;
  a.zbuf = g->zbuf.p;
  a.L = g->L.p;
  a.block_of = g->block_of.p;
  a.rank_in_block = g->rank_in_block.p;
  a.nblock = c->nblock;
  a.mode = g->mode;
  a.npn = c->npn;
  a.seed = c->seed;
  a.stage = c->stage;
  // few chains: more groups per chain (shorter per-thread loops); many: wider coalesced segments
  a.cb_shift = c->N <= 2048 ? 3 : (g->pb.P < 128 ? 5 : 4);
  return a;
}

// normals per step the free-running Philox stream reserves for the current proposal
static void set_npn(mi355_ctx* c) {
  GmhState* g = c->gmh;
  const int P = g->pb.P;
  if (g->mode == GMH_CORR || c->nblock == 0) c->npn = (P + 1) & ~1;
  else c->npn = g->maxblk == 1 ? 1 : ((g->maxblk + 1) & ~1);
}

// Block tables of the fused SED kernel (bit masks in the context) from the engine's own, for the
// current doBlock setting; problems with more parameters than the masks hold never run fused.
static void sync_block_masks(mi355_ctx* c) {
  GmhState* g = c->gmh;
  const int P = g->pb.P;
  if (P > kMaxParam) return;
  for (int b = 0; b < kMaxParam; ++b) c->blockmask[b] = 0;
  for (int i = 0; i < kMaxParam; ++i) c->rank_in_block[i] = 0;
  for (int i = 0; i < P; ++i) {
    if (c->nblock > 0) {
      const int b = g->h_block_of[i];
      if (b >= 0 && b < kMaxParam) c->blockmask[b] |= 1u << i;
      c->rank_in_block[i] = g->h_rank[i];
    } else {
      c->rank_in_block[i] = i;
    }
  }
}

// one-launch path of the running sampler: 1 binnedCl kernel, 2 SED kernel, 3 beam kernel, 0 launch per step
static int fused_kind(mi355_ctx* c) {
  GmhState* g = c->gmh;
  if (!c->mh_fused) return 0;
  if (g->pb.kind == GMH_BEAM && beam_mh_fused_ok(c)) return 3;
  if (g->pb.kind == GMH_CL && cl_mh_fused_nw(c, g->pb.P, c->nblock, c->npn, g->cov_on ? 1 : 0) > 0) return 1;
  if (g->pb.kind == GMH_SED && sed_mh_fused_ok(c)) return 2;
  return 0;
}

static int check_flags(mi355_ctx* c, const char* who) {
  std::vector<int32_t> f(static_cast<size_t>(c->N));
  MI355_HIP(hipMemcpyAsync(f.data(), c->flags.p, f.size() * sizeof(int32_t), hipMemcpyDeviceToHost,
                           c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  int64_t bad_start = 0, bad_chol = 0, first = -1;
  for (int64_t i = 0; i < c->N; ++i) {
    if (f[i] && first < 0) first = i;
    if (f[i] & kGmhFlagStartFailed) ++bad_start;
    if (f[i] & kGmhFlagCholFailed) ++bad_chol;
  }
  MI355_REQUIRE(bad_chol == 0,
                "%s: LinAlgError: normalised covariance of %lld chains is not positive definite even "
                "with +0.01 on the diagonal (first: chain %lld) [MCMC.py:397-399]", who,
                static_cast<long long>(bad_chol), static_cast<long long>(first));
  MI355_REQUIRE(bad_start == 0,
                "%s: ZeroPriorError: %lld of %lld chains could not be started with prior>0 "
                "(first: chain %lld) [MCMC.py:174-175]", who, static_cast<long long>(bad_start),
                static_cast<long long>(c->N), static_cast<long long>(first));
  return 0;
}

// start kernel + likelihood at the start + counters; shared by init and adapt
static int finish_start(mi355_ctx* c, GmhArgs& a, const char* who) {
  hipLaunchKernelGGL(k_gmh_start, dim3(grid_for(c->N)), dim3(256), 0, c->stream, a);
  MI355_HIP(hipGetLastError());
  int rc = problem_lnlike(c, c->theta.p, c->gmh->lnl_new.p);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gmh_start_finish, dim3(grid_for(c->N)), dim3(256), 0, c->stream, a);
  MI355_HIP(hipGetLastError());
  c->stage += 1;
  c->stage_step = 0;
  c->nsamp = 0;
  c->nkeep = 0;
  c->gmh->cov_nsamp = 0;
  return check_flags(c, who);
}

// The launches of MH steps on the context stream.  A run is
//   propose(0) | lnl, accept(0) + propose(1) | lnl, accept(1) + propose(2) | ... | lnl, accept(last):
// the accept of a step and the proposal of the next share one launch (k_gmh_accept_propose), and the
// beam chi-square leaves its chunk sums for the accept step to add up (no finish launch).
static unsigned step_grid(mi355_ctx* c, const GmhArgs& a) {
  return static_cast<unsigned>(ceil_div64(c->N, 1 << a.cb_shift));
}
static size_t step_lds(const GmhArgs& a) {
  return static_cast<size_t>(a.pb.P + 1) * (1 << a.cb_shift) * sizeof(double);
}
static int launch_propose(mi355_ctx* c, const GmhArgs& a) {
  hipLaunchKernelGGL(k_gmh_propose, dim3(step_grid(c, a)), dim3(256), step_lds(a), c->stream, a);
  MI355_HIP(hipGetLastError());
  return 0;
}
// likelihood of the proposals in theta_new: into lnl_new, or (beam) as chunk sums the accept step reads
static int launch_lnl(mi355_ctx* c, GmhArgs& a) {
  GmhState* g = c->gmh;
  a.lnl_partial = nullptr;
  a.lnl_nchunk = 0;
  if (g->pb.kind == GMH_BEAM)
    return beam_chi2_partials_async(c, 0, c->mhS, c->C, g->theta_new.p, &a.lnl_partial, &a.lnl_nchunk);
  return problem_lnlike(c, g->theta_new.p, g->lnl_new.p);
}
// accept of `a`'s step, and the proposal of the next (`next` non-null)
static int launch_accept(mi355_ctx* c, const GmhArgs& a, const GmhArgs* next) {
  if (next)
    hipLaunchKernelGGL(k_gmh_accept_propose, dim3(step_grid(c, a)), dim3(256), step_lds(a), c->stream, a, *next);
  else
    hipLaunchKernelGGL(k_gmh_accept, dim3(step_grid(c, a)), dim3(256), step_lds(a), c->stream, a);
  MI355_HIP(hipGetLastError());
  return 0;
}

// (Re)capture kGraphSteps MH steps whose counters come from g->dyn; a replay starts from a pending
// proposal and leaves one.  The caller has run at least one step directly since the last
// (re)initialisation, so every scratch buffer has its final size.
static int build_graph(mi355_ctx* c) {
  GmhState* g = c->gmh;
  if (g->graph_exec && g->graph_version == g->version &&
      g->graph_epoch == g_devbuf_epoch.load(std::memory_order_relaxed))
    return 0;
  if (g->graph_exec) { (void)hipGraphExecDestroy(g->graph_exec); g->graph_exec = nullptr; }
  if (g->graph) { (void)hipGraphDestroy(g->graph); g->graph = nullptr; }
  if (g->dyn.alloc(2)) return -2;
  GmhArgs a = base_args(c);
  MI355_HIP(hipStreamSynchronize(c->stream));
  c->capturing = true;
  hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
  int rc = 0;
  if (e == hipSuccess) {
    // each captured step: likelihood of the pending proposal, accept + proposal of the next step;
    // counters ping-pong between two slots
    for (int t = 0; t < kGraphSteps && rc == 0; ++t) {
      a.dyn = g->dyn.p + (t & 1);
      a.dyn_next = g->dyn.p + ((t & 1) ^ 1);
      rc = launch_lnl(c, a);
      if (rc == 0) {
        GmhArgs nx = a;                                    // (its counters: k_gmh_accept_propose, from a.dyn)
        nx.dyn_next = nullptr;
        rc = launch_accept(c, a, &nx);
      }
    }
    hipGraph_t graph = nullptr;
    const hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
    if (rc == 0 && e2 == hipSuccess) {
      g->graph = graph;
      e = hipGraphInstantiate(&g->graph_exec, g->graph, nullptr, nullptr, 0);
    } else {
      if (graph) (void)hipGraphDestroy(graph);
      if (e2 != hipSuccess) e = e2;
    }
  }
  c->capturing = false;
  if (rc) return rc;
  if (e != hipSuccess || !g->graph_exec) {
    (void)hipGetLastError();
    g->graph_exec = nullptr;
    set_error("hipGraph capture of the MH step failed: %s", hipGetErrorString(e));
    return -2;
  }
  g->graph_steps = kGraphSteps;
  g->graph_version = g->version;
  g->graph_epoch = g_devbuf_epoch.load(std::memory_order_relaxed);
  return 0;
}

int gmh_run(mi355_ctx* c, int64_t nsteps, int32_t thin, const double* inj_n, const double* inj_u) {
  GmhState* g = c->gmh;
  MI355_REQUIRE(nsteps >= 0 && thin >= 0, "negative nsteps/thin");
  MI355_REQUIRE((inj_n == nullptr) == (inj_u == nullptr),
                "inj_normals and inj_uniforms must be given together");
  if (nsteps == 0) return 0;
  MI355_HIP(hipSetDevice(c->device));
  const int P = g->pb.P;
  const int64_t N = c->N;
  if (fused_kind(c) == 2) {
    // SED: the fused kernel's extended instantiation (sed_kernels.hpp) - every proposal mode and
    // the co-moments at the one-launch rate; uploads, trace and counters are sed_mh_run's
    sync_block_masks(c);
    const int rc = sed_mh_run(c, nsteps, thin, inj_n, inj_u, g->mode, g->mode != GMH_UNCORR ? g->L.p : nullptr,
                              g->cov_on ? g->comom.p : nullptr);
    if (rc == 0 && g->cov_on) g->cov_nsamp += nsteps;      // counted like c->nsamp: after the run succeeded
    return rc;
  }
  if (inj_n) {
    const size_t nn = static_cast<size_t>(nsteps) * N * P, nu = static_cast<size_t>(nsteps) * N;
    if (g->stage_a.alloc(nn) || g->stage_b.alloc(nu)) return -2;
    MI355_HIP(hipMemcpyAsync(g->stage_a.p, inj_n, nn * 8, hipMemcpyHostToDevice, c->stream));
    MI355_HIP(hipMemcpyAsync(g->stage_b.p, inj_u, nu * 8, hipMemcpyHostToDevice, c->stream));
  }
  c->nkeep = thin > 0 ? nsteps / thin : 0;
  if (c->nkeep > 0) {
    const size_t rows = static_cast<size_t>(c->nkeep) * N;
    if (c->tr_theta.alloc(rows * P) || c->tr_lnPr.alloc(rows) || c->tr_acc.alloc(rows)) return -2;
  }
  const int fused = fused_kind(c);
  if (fused == 1 || fused == 3) {
    // binnedCl / beam: the whole run in one launch of the fused kernel (cl_mh_kernels.hpp, beam_mh_kernels.hpp)
    GmhArgs a = base_args(c);
    a.blk = c->nblock ? static_cast<int>(c->stage_step % c->nblock) : 0;
    a.step = c->step_total;
    a.nsamp = c->nsamp;
    if (inj_n) {
      a.inj_n = g->stage_a.p;
      a.inj_u = g->stage_b.p;
    }
    if (c->nkeep > 0) {
      a.tr_theta = c->tr_theta.p;
      a.tr_lnPr = c->tr_lnPr.p;
      a.tr_acc = c->tr_acc.p;
    }
    const int rc = fused == 1 ? cl_mh_run(c, a, nsteps, c->nkeep > 0 ? thin : 0, c->nkeep)
                              : beam_mh_run(c, a, nsteps, c->nkeep > 0 ? thin : 0, c->nkeep);
    if (rc) return rc;
    c->step_total += nsteps;
    c->stage_step += nsteps;
    c->nsamp += nsteps;
    if (g->cov_on) g->cov_nsamp += nsteps;
    MI355_HIP(hipStreamSynchronize(c->stream));
    return 0;
  }
  int64_t row = 0;
  // the arguments of step t of this run (counters, injected draws, trace row)
  auto step_args = [&](int64_t t) {
    GmhArgs a = base_args(c);
    a.blk = c->nblock ? static_cast<int>((c->stage_step + t) % c->nblock) : 0;
    a.step = c->step_total + t;
    a.nsamp = c->nsamp + t;
    if (inj_n) {
      a.inj_n = g->stage_a.p + static_cast<size_t>(t) * N * P;
      a.inj_u = g->stage_b.p + static_cast<size_t>(t) * N;
    }
    return a;
  };
  // step t from its pending proposal: likelihood, accept, and the proposal of step t + 1 unless it is the last
  auto run_step = [&](int64_t t) {
    GmhArgs a = step_args(t);
    if (c->nkeep > 0 && (t + 1) % thin == 0 && row < c->nkeep) {
      a.tr_theta = c->tr_theta.p + static_cast<size_t>(row) * P * N;
      a.tr_lnPr = c->tr_lnPr.p + static_cast<size_t>(row) * N;
      a.tr_acc = c->tr_acc.p + static_cast<size_t>(row) * N;
      ++row;
    }
    int rc = launch_lnl(c, a);
    if (rc) return rc;
    if (t + 1 < nsteps) {
      const GmhArgs nx = step_args(t + 1);
      return launch_accept(c, a, &nx);
    }
    return launch_accept(c, a, nullptr);
  };
  {
    const int rc = launch_propose(c, step_args(0));
    if (rc) return rc;
  }
  int64_t t0 = 0;
  if (!inj_n && c->nkeep == 0 && nsteps >= kGraphMinSteps) {
    // free-running production steps: one step directly (sizes every scratch buffer), then
    // replays of a captured graph of kGraphSteps steps, the remainder directly
    int rc = run_step(0);
    if (rc) return rc;
    t0 = 1;
    rc = build_graph(c);
    if (rc) return rc;
    GmhDyn d{c->step_total + 1, c->nsamp + 1,
             c->nblock ? static_cast<int32_t>((c->stage_step + 1) % c->nblock) : 0, 0};
    MI355_HIP(hipMemcpyAsync(g->dyn.p, &d, sizeof(d), hipMemcpyHostToDevice, c->stream));
    MI355_HIP(hipStreamSynchronize(c->stream));   // d lives on this stack frame
    // (a replay ends with the proposal of the step behind it: the last step of the run stays outside, its
    // launch has no proposal to make)
    while (nsteps - t0 > g->graph_steps) {
      MI355_HIP(hipGraphLaunch(g->graph_exec, c->stream));
      t0 += g->graph_steps;
    }
  }
  for (int64_t t = t0; t < nsteps; ++t) {
    const int rc = run_step(t);
    if (rc) return rc;
  }
  c->step_total += nsteps;
  c->stage_step += nsteps;
  c->nsamp += nsteps;
  if (g->cov_on) g->cov_nsamp += nsteps;
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int gmh_adapt(mi355_ctx* c, double fac, int random_restart, int no_correlations, int rotate,
              int do_block) {
  GmhState* g = c->gmh;
  MI355_HIP(hipSetDevice(c->device));
  const int P = g->pb.P;
  if (rotate) no_correlations = 0;                          // MCMC.py:389, :501-502
  const bool corr = !no_correlations;
  MI355_REQUIRE(!corr || g->track_cov,
                "correlated / rotated proposals need the chain covariance: pass track_cov=1 to "
                "mi355_mh_init_problem");
  MI355_REQUIRE(!corr || g->cov_nsamp == c->nsamp,
                "the co-moments hold %lld of the stage's %lld samples (mi355_mh_track_cov was off for "
                "part of it): no covariance to adapt from", static_cast<long long>(g->cov_nsamp),
                static_cast<long long>(c->nsamp));
  if (do_block >= 0) {
    MI355_REQUIRE(do_block == 0 || g->nblock_table > 0,
                  "doBlock=True needs the parameter blocks given to mi355_mh_init_problem");
    c->nblock = do_block ? g->nblock_table : 0;
  }
  MI355_REQUIRE(!rotate || c->nblock > 0, "rotateParams needs doBlock=True (Proposal.py:145-148 "
                "iterates over the block)");
  if (!(fac > 0.0)) fac = 2.4 / std::sqrt(static_cast<double>(P));   // MCMC.py:379-380
  if (corr && g->L.alloc(static_cast<size_t>(P) * P * c->N)) return -2;
  GmhArgs a = base_args(c);
  a.nsamp = c->nsamp;
  a.fac = fac;
  a.random_restart = random_restart;
  a.want_corr = corr ? 1 : 0;
  a.Lout = g->L.p;
  hipLaunchKernelGGL(k_gmh_adapt, dim3(grid_for(c->N)), dim3(256), 0, c->stream, a);
  MI355_HIP(hipGetLastError());
  g->mode = corr ? (rotate ? GMH_ROTATE : GMH_CORR) : GMH_UNCORR;
  g->version += 1;
  set_npn(c);
  a = base_args(c);
  a.randomize = 1;
  a.center = nullptr;                                       // centre is in theta_new
  a.sigma_in = nullptr;
  return finish_start(c, a, "mi355_mh_adapt");
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_mh_init_problem(mi355_ctx* c, int32_t problem, int32_t C, const double* theta0,
                          int32_t theta0_per_chain, const double* sigma, int32_t sigma_per_chain,
                          const int32_t* param_blocks, int32_t nblock, int32_t randomize_start,
                          int32_t track_cov) {
  MI355_REQUIRE(c && theta0 && sigma, "mi355_mh_init_problem: null argument");
  MI355_REQUIRE(C > 0, "C must be positive");
  MI355_HIP(hipSetDevice(c->device));
  if (!c->gmh) c->gmh = new GmhState();
  GmhState* g = c->gmh;
  g->active = false;
  c->mh_ready = false;
  GmhProblem pb{};
  pb.kind = problem;
  int64_t units = 0;
  if (problem == MI355_PROBLEM_SED) {
    MI355_REQUIRE(c->sed_loaded, "mi355_mh_init_problem: no SED catalogue loaded");
    pb.P = c->P;
    pb.sed_model = c->model.model;
    pb.k = c->kc;
    units = c->S;
  } else if (problem == MI355_PROBLEM_CL) {
    int npar = 0;
    MI355_REQUIRE(cl_problem_info(c, &npar, &pb.cltype) == 0,
                  "mi355_mh_init_problem: no binned Cl problem loaded (mi355_cl_load + "
                  "mi355_cl_set_binning)");
    pb.P = npar;
    units = 1;
  } else if (problem == MI355_PROBLEM_BEAM) {
    MI355_REQUIRE(beam_problem_info(c, &units, &pb.box) == 0,
                  "mi355_mh_init_problem: no beam maps loaded (mi355_beam_load)");
    pb.P = kBeamP;
  } else {
    set_error("mi355_mh_init_problem: unknown problem kind %d", problem);
    return -1;
  }
  const int P = pb.P;
  MI355_REQUIRE(P > 0 && P <= kGmhMaxP, "problem has %d parameters (limit %d)", P, kGmhMaxP);
  MI355_REQUIRE(nblock >= 0 && nblock <= P, "nblock=%d outside [0,%d]", nblock, P);
  MI355_REQUIRE(nblock == 0 || param_blocks, "param_blocks required when nblock>0");
  g->pb = pb;
  g->mode = GMH_UNCORR;
  g->track_cov = track_cov != 0;
  g->cov_on = g->track_cov;
  c->mhP = P;
  c->mhS = units;
  c->C = C;
  c->N = units * C;
  c->nblock = nblock;
  g->nblock_table = nblock;
  // block tables; a parameter with block id -1 (or beyond the table) belongs to no block
  g->h_block_of.assign(static_cast<size_t>(P), -1);
  std::vector<int32_t>& rank = g->h_rank;
  rank.assign(static_cast<size_t>(P), 0);
  std::vector<int32_t> count(static_cast<size_t>(P) + 1, 0);
  g->maxblk = 1;
  if (nblock > 0) {
    for (int i = 0; i < P; ++i) {
      const int b = param_blocks[i];
      MI355_REQUIRE(b >= -1 && b < nblock, "param_blocks[%d]=%d outside [-1,%d)", i, b, nblock);
      g->h_block_of[i] = b;
      if (b < 0) continue;
      rank[i] = count[b]++;
      if (count[b] > g->maxblk) g->maxblk = count[b];
    }
  } else {
    for (int i = 0; i < P; ++i) rank[i] = i;
  }
  if (g->block_of.alloc(P) || g->rank_in_block.alloc(P)) return -2;
  MI355_HIP(hipMemcpyAsync(g->block_of.p, g->h_block_of.data(), P * 4, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(g->rank_in_block.p, rank.data(), P * 4, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  set_npn(c);
  const size_t NP = static_cast<size_t>(c->N) * P;
  if (c->theta.alloc(NP) || c->sigma.alloc(NP) || c->mean.alloc(NP) || c->m2.alloc(NP)) return -2;
  if (c->lnPr.alloc(c->N) || c->naccept.alloc(c->N) || c->flags.alloc(c->N)) return -2;
  if (c->accept.alloc(static_cast<size_t>(c->N) * (nblock > 0 ? nblock : 1))) return -2;
  if (g->theta_new.alloc(NP) || g->zbuf.alloc(NP) || g->prior_new.alloc(c->N) || g->lnl_new.alloc(c->N))
    return -2;
  if (g->track_cov && g->comom.alloc(NP * P)) return -2;
  // theta0 / sigma: host [P] or [N][P] -> device [P] or [P][N]
  auto upload = [&](const double* host, int per_chain, DevBuf<double>& stage, DevBuf<double>& out) -> int {
    if (!per_chain) {
      if (out.alloc(static_cast<size_t>(P))) return -2;
      MI355_HIP(hipMemcpyAsync(out.p, host, P * 8, hipMemcpyHostToDevice, c->stream));
      return 0;
    }
    if (stage.alloc(NP) || out.alloc(NP)) return -2;
    MI355_HIP(hipMemcpyAsync(stage.p, host, NP * 8, hipMemcpyHostToDevice, c->stream));
    // [N][P] -> [P][N] is the P x P spread with per_chain layout of a 1 x P "matrix" per chain
    hipLaunchKernelGGL(k_gmh_transpose_np, dim3(grid_for(static_cast<int64_t>(NP))), dim3(256), 0,
                       c->stream, c->N, P, stage.p, out.p);
    MI355_HIP(hipGetLastError());
    return 0;
  };
  if (upload(theta0, theta0_per_chain, c->scratch_c, c->scratch_a)) return -2;
  if (upload(sigma, sigma_per_chain, c->scratch_c, c->scratch_b)) return -2;
  MI355_HIP(hipMemsetAsync(c->flags.p, 0, c->N * sizeof(int32_t), c->stream));
  g->active = true;
  g->version += 1;
  GmhArgs a = base_args(c);
  a.center = c->scratch_a.p;
  a.center_per_chain = theta0_per_chain;
  a.sigma_in = c->scratch_b.p;
  a.sigma_per_chain = sigma_per_chain;
  a.randomize = randomize_start;
  int rc = finish_start(c, a, "mi355_mh_init_problem");
  if (rc) {
    g->active = false;
    return rc;
  }
  c->mh_ready = true;
  return 0;
}

int mi355_mh_set_sqrt_matrix(mi355_ctx* c, const double* L, int32_t per_chain, int32_t rotate) {
  MI355_REQUIRE(gmh_active(c) && c->mh_ready, "mi355_mh_set_sqrt_matrix: no step-wise sampler "
                "(call mi355_mh_init_problem)");
  MI355_HIP(hipSetDevice(c->device));
  GmhState* g = c->gmh;
  const int P = g->pb.P;
  if (!L) {
    MI355_REQUIRE(!rotate, "rotateParams needs a sqrt matrix");
    g->mode = GMH_UNCORR;
    g->version += 1;
    set_npn(c);
    return 0;
  }
  MI355_REQUIRE(!rotate || c->nblock > 0, "rotateParams needs doBlock=True (Proposal.py:145-148)");
  const size_t nin = static_cast<size_t>(P) * P * (per_chain ? c->N : 1);
  if (g->stage_a.alloc(nin) || g->L.alloc(static_cast<size_t>(P) * P * c->N)) return -2;
  MI355_HIP(hipMemcpyAsync(g->stage_a.p, L, nin * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_gmh_spread_matrix, dim3(grid_for(c->N * P * P)), dim3(256), 0, c->stream, c->N,
                     P, g->stage_a.p, per_chain, g->L.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipStreamSynchronize(c->stream));
  g->mode = rotate ? GMH_ROTATE : GMH_CORR;
  g->version += 1;
  set_npn(c);
  return 0;
}

int mi355_mh_adapt_corr(mi355_ctx* c, double fac, int32_t random_restart, int32_t no_correlations,
                        int32_t rotate, int32_t do_block) {
  MI355_REQUIRE(gmh_active(c) && c->mh_ready, "mi355_mh_adapt_corr: no step-wise sampler "
                "(call mi355_mh_init_problem)");
  MI355_REQUIRE(c->nsamp > 0, "mi355_mh_adapt_corr: no samples to adapt from");
  return gmh_adapt(c, fac, random_restart, no_correlations, rotate, do_block);
}

static int gather(mi355_ctx* c, const double* dev, double scale, int sym, const double* m2, double rn,
                  double* host) {
  GmhState* g = c->gmh;
  const int P = g->pb.P;
  const size_t n = static_cast<size_t>(c->N) * P * P;
  if (g->stage_a.alloc(n)) return -2;
  hipLaunchKernelGGL(k_gmh_gather_matrix, dim3(grid_for(static_cast<int64_t>(n))), dim3(256), 0,
                     c->stream, c->N, P, dev, scale, sym, m2, rn, g->stage_a.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(host, g->stage_a.p, n * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_mh_covar(mi355_ctx* c, double* cov, int32_t normalized) {
  MI355_REQUIRE(gmh_active(c) && c->mh_ready && cov, "mi355_mh_covar: no step-wise sampler");
  MI355_REQUIRE(c->gmh->track_cov, "mi355_mh_covar: the sampler was started with track_cov=0");
  MI355_REQUIRE(c->gmh->cov_nsamp == c->nsamp, "mi355_mh_covar: the co-moments hold %lld of the stage's "
                "%lld samples (mi355_mh_track_cov was off for part of it)",
                static_cast<long long>(c->gmh->cov_nsamp), static_cast<long long>(c->nsamp));
  MI355_REQUIRE(c->nsamp > 0, "mi355_mh_covar: no samples accumulated yet");
  MI355_HIP(hipSetDevice(c->device));
  const double rn = 1.0 / static_cast<double>(c->nsamp);
  return gather(c, c->gmh->comom.p, rn, 1, normalized ? c->m2.p : nullptr, rn, cov);
}

int mi355_mh_get_sqrt_matrix(mi355_ctx* c, double* L) {
  MI355_REQUIRE(gmh_active(c) && c->mh_ready && L, "mi355_mh_get_sqrt_matrix: no step-wise sampler");
  MI355_REQUIRE(c->gmh->mode != GMH_UNCORR, "mi355_mh_get_sqrt_matrix: the proposal is uncorrelated");
  MI355_HIP(hipSetDevice(c->device));
  return gather(c, c->gmh->L.p, 1.0, 0, nullptr, 0.0, L);
}

int32_t mi355_mh_proposal_mode(mi355_ctx* c) { return gmh_active(c) ? c->gmh->mode : 0; }

int mi355_mh_track_cov(mi355_ctx* c, int32_t on) {
  MI355_REQUIRE(gmh_active(c) && c->mh_ready, "mi355_mh_track_cov: no step-wise sampler");
  MI355_REQUIRE(!on || c->gmh->track_cov, "mi355_mh_track_cov: the sampler was started with track_cov=0");
  if ((on != 0) != c->gmh->cov_on) c->gmh->version += 1;   // a captured graph has the old setting baked in
  c->gmh->cov_on = on != 0;
  return 0;
}

int mi355_mh_set_fused(mi355_ctx* c, int32_t on) {
  MI355_REQUIRE(c, "null context");
  MI355_REQUIRE(on >= 0 && on <= 2, "mi355_mh_set_fused: %d (0 step-wise, 1 fused where it is the faster path, 2 fused wherever a kernel serves)", on);
  c->mh_fused = on;
  return 0;
}

int mi355_mh_fused_active(mi355_ctx* c) {
  if (!gmh_active(c) || !c->mh_ready) return 0;
  return fused_kind(c) ? 1 : 0;
}

}  // extern "C"
