// C ABI of the binnedCl / ClData band-power likelihood (include/mi355mcmc.h, "binnedCl" block).
#include <cmath>
#include <cstring>
#include <memory>

#include "cl_problem.hpp"
#include "problems.hpp"

namespace mi355 {

void cl_destroy(ClProblem* p) { delete p; }

int cl_problem_info(const mi355_ctx* c, int* npar, const int32_t** cltype_dev) {
  if (!c || !c->cl || !c->cl->binned) return -1;
  *npar = c->cl->npar;
  *cltype_dev = c->cl->cltype.p;
  return 0;
}

template <typename T>
static int upload(mi355_ctx* c, DevBuf<T>& buf, const T* host, size_t n, size_t n_alloc = 0) {
  if (n_alloc < n) n_alloc = n;
  if (buf.alloc(n_alloc)) return -2;
  if (n_alloc > n) MI355_HIP(hipMemsetAsync(buf.p, 0, n_alloc * sizeof(T), c->stream));
  if (n) MI355_HIP(hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return 0;
}

static ClBinning binning_of(const ClProblem* p) {
  return ClBinning{p->nCl, p->npar, p->lmax, p->shapefun.p, p->binof.p};
}

static int build_K(mi355_ctx* c, ClData* d) {
  ClProblem* p = c->cl;
  const size_t nk = static_cast<size_t>(p->npar) * cl_gstride(d->nbp);
  if (d->K.alloc(nk) || d->bp0.alloc(static_cast<size_t>(d->nbp))) return -2;
  MI355_HIP(hipMemsetAsync(d->K.p, 0, nk * sizeof(double), c->stream));
  MI355_HIP(hipMemsetAsync(d->bp0.p, 0, d->nbp * sizeof(double), c->stream));
  hipLaunchKernelGGL(k_cl_build_K, dim3((d->nb + 63) / 64), dim3(64), 0, c->stream, binning_of(p),
                     d->nb, d->nbp, d->wl, d->window.p, d->win_min.p, d->win_max.p,
                     d->has_pol ? 1 : 0, d->inc_pol.p, d->K.p, d->bp0.p);
  MI355_HIP(hipGetLastError());
  // non-zero range of every column tile (16 bands), in k-steps of four bins
  std::vector<double> hK(nk);
  MI355_HIP(hipMemcpyAsync(hK.data(), d->K.p, nk * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  const int nct = d->nbp >> 4, gs = cl_gstride(d->nbp);
  std::vector<int32_t> kr(static_cast<size_t>(2) * nct, 0);
  for (int ct = 0; ct < nct; ++ct) {
    int lo = p->npar, hi = -1;
    for (int b = 0; b < p->npar; ++b)
      for (int i = ct * 16; i < ct * 16 + 16; ++i)
        if (hK[static_cast<size_t>(b) * gs + i] != 0.0) {    // NaN counts as non-zero
          lo = std::min(lo, b);
          hi = std::max(hi, b);
        }
    if (hi >= 0) {
      kr[2 * ct] = lo >> 2;
      kr[2 * ct + 1] = (hi >> 2) + 1;
    }
    d->kmax = std::max(ct == 0 ? 1 : d->kmax, kr[2 * ct + 1] - kr[2 * ct]);
  }
  if (upload(c, d->krange, kr.data(), kr.size())) return -2;
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

template <int RT, int CG, int MODE, int NW, bool SYM = false>
static int launch_form(mi355_ctx* c, const ClFormArgs& a) {
  constexpr int TR = 16 * RT;
  const int rows_used = (a.mode == CL_ROWS_TRIPLE) ? (TR / 3) * 3 : TR;
  const size_t lds = form_lds_bytes(a.nbp, TR, NW);
  // largest dynamic LDS opted in for this instantiation, per device
  static size_t opted[kMaxDevices] = {};
  size_t& have = opted[c->device % kMaxDevices];
  if (lds > 48 * 1024 && lds > have) {
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cl_form<RT, CG, MODE, NW, SYM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    have = lds;
  }
  const unsigned grid = static_cast<unsigned>(ceil_div64(a.rows, rows_used));
  hipLaunchKernelGGL((k_cl_form<RT, CG, MODE, NW, SYM>), dim3(grid), dim3(64 * NW), lds, c->stream, a);
  MI355_HIP(hipGetLastError());
  return 0;
}

template <int RT, int MODE>
static int launch_form_cg(mi355_ctx* c, const ClFormArgs& a) {
  const int nct = a.nbp >> 4;                       // column tiles, shared out over the waves
  if constexpr (MODE != CL_ROWS_TRIPLE) {
    // diagonal forms: block-upper symmetrised matrix, half the MFMA work, from 8 tiles up
    if (nct >= 8) return launch_form<RT, 1, MODE, 8, true>(c, a);
  }
  if (nct <= 4) return launch_form<RT, 1, MODE, 4>(c, a);
  if (nct <= 8) return launch_form<RT, 1, MODE, 8>(c, a);
  if (nct <= 16) return launch_form<RT, 2, MODE, 8>(c, a);
  return launch_form<RT, 4, MODE, 8>(c, a);
}

template <int RT>
static int launch_form_rt(mi355_ctx* c, const ClFormArgs& a) {
  if (a.mode == CL_ROWS_TRIPLE) return launch_form_cg<RT, CL_ROWS_TRIPLE>(c, a);
  if (a.mode == CL_ROWS_RAW) return launch_form_cg<RT, CL_ROWS_RAW>(c, a);
  return launch_form_cg<RT, CL_ROWS_MARGE>(c, a);
}


// chisq rows / G matrices of data set d for the band powers in p->bp (or raw vectors)
static int run_form(mi355_ctx* c, ClData* d, const double* rows_src, int64_t n, int mode, int ncal,
                    int nbeam, int rows_per_chain, double* out) {
  ClFormArgs a{};
  a.bp = rows_src;
  a.n = n;
  a.nb = d->nb;
  a.nbp = d->nbp;
  a.obs = d->obs.p;
  a.xf = d->xf.p;
  a.hasx = d->hasx.p;
  a.beam_err = d->beam_err.p;
  a.bfac = d->bfac.p;
  a.ninv = d->ninv.p;
  a.nsym = d->nsym.p;
  a.nstride = cl_gstride(d->nbp);
  a.var = d->var.p;
  a.has_x = d->has_x ? 1 : 0;
  a.beam_uncertain = d->beam_uncertain ? 1 : 0;
  a.calib = d->calib;
  a.mode = mode;
  a.ncal = ncal;
  a.nbeam = nbeam;
  a.rows = n * rows_per_chain;
  a.out = out;
  MI355_REQUIRE(a.rows < (int64_t{1} << 31), "too many rows (%lld) for one Cl launch",
                static_cast<long long>(a.rows));
  if (!d->has_corr) {
    const int64_t work = (mode == CL_ROWS_TRIPLE) ? n : a.rows;
    hipLaunchKernelGGL(k_cl_form_diag, dim3(static_cast<unsigned>(ceil_div64(work, 256))), dim3(256),
                       0, c->stream, a);
    MI355_HIP(hipGetLastError());
    return 0;
  }
  MI355_REQUIRE(a.rows < (int64_t{1} << 31), "too many rows (%lld) for one Cl launch",
                static_cast<long long>(a.rows));
  auto fits = [&](int TR) { return form_lds_bytes(a.nbp, TR) <= kClLdsBudget; };
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
  int rc;
  // Row tile of 32: two blocks fit in a CU's LDS up to 256 bands, so one block's MFMA phase runs
  // under the other's row staging; measured faster than 64-row tiles (one block per CU, better
  // reuse of the matrix fragments) in every configuration: 1.19 vs 1.52 ms at C4.
  if (fits(32) && a.rows >= 32 * 512) rc = launch_form_rt<2>(c, a);
  else if (fits(16)) rc = launch_form_rt<1>(c, a);
  else {
    set_error("num_points=%d too large for the LDS row tile of the Cl quadratic form", d->nb);
    return -1;
  }
  if (rc) return rc;
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
  if (!c->capturing) c->ev_valid = true;
  return 0;
}

static int run_combine(mi355_ctx* c, ClData* d, const FormPlan& f, int64_t n, const double* form,
                       double* lnl, int accumulate, double* neglnl) {
  ClProblem* p = c->cl;
  ClCombineArgs a{};
  a.form = form;
  a.n = n;
  a.mode = f.mode;
  a.ncal = f.ncal;
  a.nbeam = f.nbeam;
  a.beam_uncertain = d->beam_uncertain ? 1 : 0;
  a.calib = d->calib;
  a.weights = p->weights.p;
  a.margenorm = p->margenorm;
  a.lnl = lnl;
  a.accumulate = accumulate;
  a.neglnl = neglnl;
  hipLaunchKernelGGL(k_cl_combine, dim3(static_cast<unsigned>(ceil_div64(n, 256))), dim3(256), 0,
                     c->stream, a);
  MI355_HIP(hipGetLastError());
  return 0;
}

// -ln L of data set d from the band powers in bp_dev [nbp][n]
static int dataset_from_bp(mi355_ctx* c, ClData* d, const double* bp_dev, int64_t n, double* lnl,
                           int accumulate, double* neglnl) {
  ClProblem* p = c->cl;
  const FormPlan f = plan_for(d);
  const size_t nform = static_cast<size_t>(n) * (f.mode == CL_ROWS_TRIPLE ? 9 : f.rows_per_chain);
  if (p->form.alloc(nform)) return -2;
  int rc = run_form(c, d, bp_dev, n, f.mode, f.ncal, f.nbeam, f.rows_per_chain, p->form.p);
  if (rc) return rc;
  return run_combine(c, d, f, n, p->form.p, lnl, accumulate, neglnl);
}

static int check_ready(mi355_ctx* c, bool need_binning, const char* who) {
  MI355_REQUIRE(c && c->cl && !c->cl->data.empty(), "%s: no Cl data loaded (call mi355_cl_load)", who);
  MI355_REQUIRE(!need_binning || c->cl->binned, "%s: no binning set (call mi355_cl_set_binning)", who);
  return 0;
}

static int check_idata(mi355_ctx* c, int idata, const char* who) {
  MI355_REQUIRE(idata >= 0 && idata < static_cast<int>(c->cl->data.size()),
                "%s: data set %d outside [0,%d)", who, idata, static_cast<int>(c->cl->data.size()));
  return 0;
}

// host [n][cols] -> device band-major [colsp][n]
static int to_bandmajor(mi355_ctx* c, const double* host, int64_t n, int cols, int colsp,
                        DevBuf<double>& stage, DevBuf<double>& out) {
  const size_t ni = static_cast<size_t>(n) * cols, no = static_cast<size_t>(n) * colsp;
  if (stage.alloc(ni) || out.alloc(no)) return -2;
  MI355_HIP(hipMemcpyAsync(stage.p, host, ni * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_cl_to_bandmajor, dim3(static_cast<unsigned>(ceil_div64(no, 256))), dim3(256),
                     0, c->stream, n, cols, colsp, stage.p, out.p);
  MI355_HIP(hipGetLastError());
  return 0;
}

}  // namespace mi355

using namespace mi355;

namespace mi355 {
// BP = q.K + BP0 of n chains into p->bp: k_cl_bandpower stages 16 chains' bin amplitudes in LDS,
// 128 B per bin - beyond the 48 KB a kernel gets unasked (384 bins) the launch raises its limit
// (kClMaxBins bins = 128 KB; the reference has no limit, no data set comes near this one)
constexpr int kClMaxBins = 1024;
static int launch_bandpower(mi355_ctx* c, ClProblem* p, ClData* d, int64_t n, const double* q_pm) {
  const size_t lds = static_cast<size_t>(p->npar) * 16 * sizeof(double);
  if (lds > 48 * 1024)
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cl_bandpower),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(k_cl_bandpower, dim3(static_cast<unsigned>(ceil_div64(n, 16))), dim3(256), lds, c->stream, n,
                     p->npar, d->nb, d->nbp, q_pm, d->K.p, d->krange.p, d->bp0.p, p->bp.p);
  MI355_HIP(hipGetLastError());
  return 0;
}
}  // namespace mi355

extern "C" {

int mi355_cl_load(mi355_ctx* c, int32_t ndata, const mi355_cl_dataset* data) {
  MI355_REQUIRE(c && data && ndata > 0, "mi355_cl_load: bad argument");
  mh_invalidate(c);
  MI355_HIP(hipSetDevice(c->device));
  auto prob = std::make_unique<ClProblem>();
  for (int i = 0; i < ndata; ++i) {
    const mi355_cl_dataset& s = data[i];
    MI355_REQUIRE(s.num_points > 0 && s.win_len > 0, "data set %d: empty (num_points=%d, win_len=%d)",
                  i, s.num_points, s.win_len);
    MI355_REQUIRE(s.obs && s.window && s.win_min && s.win_max, "data set %d: null array", i);
    MI355_REQUIRE(s.N_inv || s.var, "data set %d: neither N_inv nor var given", i);
    MI355_REQUIRE((s.xfactors == nullptr) == (s.has_xfactor == nullptr),
                  "data set %d: xfactors and has_xfactor must be given together", i);
    MI355_REQUIRE(!s.has_pol || s.inc_pol, "data set %d: has_pol without inc_pol", i);
    auto d = std::make_unique<ClData>();
    const int nb = s.num_points, nbp = (nb + 15) & ~15;
    d->nb = nb;
    d->nbp = nbp;
    d->wl = s.win_len;
    d->calib = s.calib_uncertainty;
    d->beam_uncertain = s.beam_err != nullptr;
    d->has_pol = s.has_pol != 0;
    d->has_corr = s.N_inv != nullptr;
    for (int k = 0; k < nb; ++k)
      MI355_REQUIRE(s.win_min[k] >= 0 && s.win_min[k] <= s.win_max[k] + 1,
                    "data set %d band %d: window range [%d,%d] invalid", i, k, s.win_min[k],
                    s.win_max[k]);
    bool any_x = false;
    if (s.has_xfactor)
      for (int k = 0; k < nb; ++k) any_x = any_x || s.has_xfactor[k];
    d->has_x = any_x;
    if (upload(c, d->obs, s.obs, nb, nbp)) return -2;
    std::vector<double> zeros(static_cast<size_t>(nbp), 0.0);
    std::vector<uint8_t> zb(static_cast<size_t>(nbp), 0);
    if (upload(c, d->xf, s.xfactors ? s.xfactors : zeros.data(), nb, nbp)) return -2;
    if (upload(c, d->hasx, s.has_xfactor ? s.has_xfactor : zb.data(), nb, nbp)) return -2;
    if (upload(c, d->beam_err, s.beam_err ? s.beam_err : zeros.data(), nb, nbp)) return -2;
    if (upload(c, d->inc_pol, s.inc_pol ? s.inc_pol : zb.data(), nb, nbp)) return -2;
    if (d->bfac.alloc(static_cast<size_t>(kClGrid) * nbp)) return -2;
    hipLaunchKernelGGL(k_cl_beam_factors, dim3((kClGrid * nbp + 255) / 256), dim3(256), 0, c->stream, nbp,
                       d->beam_err.p, d->bfac.p);
    MI355_HIP(hipGetLastError());
    if (upload(c, d->win_min, s.win_min, nb) || upload(c, d->win_max, s.win_max, nb)) return -2;
    if (upload(c, d->window, s.window, static_cast<size_t>(nb) * 3 * s.win_len)) return -2;
    std::vector<double> pad, sym;   // staging; must outlive the copies below
    if (d->has_corr) {
      const size_t gs = static_cast<size_t>(cl_gstride(nbp));   // row stride in HBM
      pad.assign(static_cast<size_t>(nbp) * gs, 0.0);
      for (int r = 0; r < nb; ++r)
        std::memcpy(&pad[static_cast<size_t>(r) * gs], s.N_inv + static_cast<size_t>(r) * nb,
                    nb * sizeof(double));
      // zero rows beyond the matrix: the form fetches its fragments up to eight k-steps ahead
      if (upload(c, d->ninv, pad.data(), pad.size(), pad.size() + kClFormSlackRows * gs)) return -2;
      // block-upper symmetrised copy for the diagonal forms (k_cl_form SYM): one rounding per
      // off-diagonal entry, none at all when N^-1 is exactly symmetric
      sym.assign(pad.size(), 0.0);
      for (int r = 0; r < nbp; ++r)
        for (int q = 0; q < nbp; ++q) {
          const int I = r >> 4, J = q >> 4;
          if (I == J) sym[static_cast<size_t>(r) * gs + q] = pad[static_cast<size_t>(r) * gs + q];
          else if (I < J)
            sym[static_cast<size_t>(r) * gs + q] =
                pad[static_cast<size_t>(r) * gs + q] + pad[static_cast<size_t>(q) * gs + r];
        }
      if (upload(c, d->nsym, sym.data(), sym.size(), sym.size() + kClFormSlackRows * gs)) return -2;
    } else {
      pad.assign(static_cast<size_t>(nbp), 1.0);
      std::memcpy(pad.data(), s.var, nb * sizeof(double));
      if (upload(c, d->var, pad.data(), pad.size())) return -2;
    }
    MI355_HIP(hipStreamSynchronize(c->stream));
    prob->data.push_back(std::move(d));
  }
  // margeweights = exp(-(step*3/halfsteps)**2/2), margenorm = sum (builtin, in order)   :42-47
  double w[kClGrid], norm = 0.0;
  for (int i = 0; i < kClGrid; ++i) {
    const double step = static_cast<double>(i - kClHalfSteps);
    const double t = step * 3 / static_cast<double>(kClHalfSteps);
    w[i] = std::exp(-(t * t) / 2);
    norm += w[i];
  }
  prob->margenorm = norm;
  if (upload(c, prob->weights, w, kClGrid)) return -2;
  MI355_HIP(hipStreamSynchronize(c->stream));
  cl_destroy(c->cl);
  c->cl = prob.release();
  return 0;
}

int mi355_cl_set_binning(mi355_ctx* c, int32_t nCl, const int32_t* nbins, const int32_t* bins,
                         const double* shapefun, int32_t lmax) {
  if (check_ready(c, false, "mi355_cl_set_binning")) return -1;
  MI355_REQUIRE(nbins && bins && shapefun, "mi355_cl_set_binning: null argument");
  MI355_REQUIRE(nCl >= 1 && nCl <= 3, "Error in shapes of bins, shapefun, giving nCl=%d "
                "[binnedClModel.py:163-164]", nCl);
  MI355_REQUIRE(lmax >= 1, "mi355_cl_set_binning: lmax=%d", lmax);
  mh_invalidate(c);                                     // the parameter vector changes with the bins
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  p->binned = false;
  const int L = lmax + 1;
  std::vector<int32_t> binof(static_cast<size_t>(nCl) * L, -1), cltype;
  int ib = 0, seen_max = 0;
  for (int iCl = 0; iCl < nCl; ++iCl) {
    MI355_REQUIRE(nbins[iCl] > 0, "spectrum %d has no bins", iCl);
    for (int j = 0; j < nbins[iCl]; ++j, ++ib) {
      const int lo = bins[2 * ib], hi = bins[2 * ib + 1];
      MI355_REQUIRE(lo >= 0 && hi >= lo && hi <= lmax, "bin %d = (%d,%d) outside [0,%d]", ib, lo, hi,
                    lmax);
      if (hi > seen_max) seen_max = hi;
      for (int l = lo; l <= hi; ++l) {
        MI355_REQUIRE(binof[iCl * L + l] < 0,
                      "bins %d and %d of spectrum %d overlap at l=%d (bins are inclusive, "
                      "binnedClModel.py:149-150)", binof[iCl * L + l], ib, iCl, l);
        binof[iCl * L + l] = ib;
      }
      cltype.push_back(iCl);
    }
  }
  MI355_REQUIRE(seen_max == lmax, "lmax=%d but the largest binned multipole is %d: pass "
                "shapefun[:, :lmax+1] for that lmax", lmax, seen_max);
  MI355_REQUIRE(ib <= kClMaxBins, "%d bins: the band-power kernel stages at most %d bin amplitudes per "
                "chain in LDS", ib, kClMaxBins);
  p->nCl = nCl;
  p->npar = ib;
  p->lmax = lmax;
  if (upload(c, p->shapefun, shapefun, static_cast<size_t>(nCl) * L)) return -2;
  if (upload(c, p->binof, binof.data(), binof.size())) return -2;
  if (upload(c, p->cltype, cltype.data(), cltype.size())) return -2;
  for (auto& d : p->data)
    if (build_K(c, d.get())) return -2;
  MI355_HIP(hipStreamSynchronize(c->stream));
  p->binned = true;
  cl_plan_fused(p);
  return 0;
}

int32_t mi355_cl_nparam(mi355_ctx* c) { return (c && c->cl && c->cl->binned) ? c->cl->npar : -1; }

int mi355_cl_lnlike_dev_async(mi355_ctx* c, int64_t n, const double* q_dev, double* lnL_dev) {
  if (check_ready(c, true, "mi355_cl_lnlike")) return -1;
  MI355_REQUIRE(n > 0 && q_dev && lnL_dev, "mi355_cl_lnlike: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  int first = 1;
  for (auto& d : p->data) {
    if (p->bp.alloc(static_cast<size_t>(n) * d->nbp)) return -2;
    if (launch_bandpower(c, p, d.get(), n, q_dev)) return -2;
    int rc = dataset_from_bp(c, d.get(), p->bp.p, n, lnL_dev, first ? 0 : 1, nullptr);
    if (rc) return rc;
    first = 0;
  }
  return 0;
}

int mi355_cl_lnlike(mi355_ctx* c, int64_t n, const double* q, double* lnL) {
  if (check_ready(c, true, "mi355_cl_lnlike")) return -1;
  MI355_REQUIRE(n > 0 && q && lnL, "mi355_cl_lnlike: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  if (to_bandmajor(c, q, n, p->npar, p->npar, p->q_host_layout, p->q_pm)) return -2;
  if (p->lnl.alloc(static_cast<size_t>(n))) return -2;
  int rc = mi355_cl_lnlike_dev_async(c, n, p->q_pm.p, p->lnl.p);
  if (rc) return rc;
  MI355_HIP(hipMemcpyAsync(lnL, p->lnl.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_cl_prior(mi355_ctx* c, int64_t n, const double* q, double* prior) {
  if (check_ready(c, true, "mi355_cl_prior")) return -1;
  MI355_REQUIRE(n > 0 && q && prior, "mi355_cl_prior: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  if (upload(c, p->tmp_in, q, static_cast<size_t>(n) * p->npar)) return -2;
  if (p->tmp_out.alloc(static_cast<size_t>(n))) return -2;
  hipLaunchKernelGGL(k_cl_prior, dim3(static_cast<unsigned>(ceil_div64(n, 256))), dim3(256), 0,
                     c->stream, n, p->npar, p->cltype.p, p->tmp_in.p, p->tmp_out.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(prior, p->tmp_out.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_cl_model(mi355_ctx* c, int64_t n, const double* q, double* Cl) {
  if (check_ready(c, true, "mi355_cl_model")) return -1;
  MI355_REQUIRE(n > 0 && q && Cl, "mi355_cl_model: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  const size_t no = static_cast<size_t>(n) * 3 * (p->lmax + 1);
  if (upload(c, p->tmp_in, q, static_cast<size_t>(n) * p->npar) || p->tmp_out.alloc(no)) return -2;
  hipLaunchKernelGGL(k_cl_model, dim3(static_cast<unsigned>(ceil_div64(no, 256))), dim3(256), 0,
                     c->stream, binning_of(p), n, p->tmp_in.p, p->tmp_out.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(Cl, p->tmp_out.p, no * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// band powers of data set d into p->bp ([nbp][n]) from explicit spectra
static int bp_from_cl(mi355_ctx* c, ClData* d, int64_t n, const double* Cl, int rows, int L) {
  ClProblem* p = c->cl;
  MI355_REQUIRE(rows == 1 || rows == 3, "Cl shape wrong in getWinBandpower: %d spectra "
                "[ClData_CosmoMC.py:429-434]", rows);
  MI355_REQUIRE(L > 0, "empty spectrum");
  const size_t ncl = static_cast<size_t>(n) * rows * L;
  if (upload(c, p->tmp_in, Cl, ncl) || p->bp.alloc(static_cast<size_t>(n) * d->nbp)) return -2;
  MI355_HIP(hipMemsetAsync(p->bp.p, 0, static_cast<size_t>(n) * d->nbp * sizeof(double), c->stream));
  hipLaunchKernelGGL(k_cl_bandpower_direct, dim3(static_cast<unsigned>(ceil_div64(n * d->nb * 64, 256))),
                     dim3(256), 0, c->stream, n, rows, L, p->tmp_in.p, d->nb, d->wl, d->window.p,
                     d->win_min.p, d->win_max.p, d->has_pol ? 1 : 0, d->inc_pol.p, p->bp.p);
  MI355_HIP(hipGetLastError());
  return 0;
}

// device [nbp][n] -> host [n][nb]
static int bp_to_host(mi355_ctx* c, ClData* d, int64_t n, double* out) {
  ClProblem* p = c->cl;
  std::vector<double> h(static_cast<size_t>(n) * d->nbp);
  MI355_HIP(hipMemcpyAsync(h.data(), p->bp.p, h.size() * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  for (int k = 0; k < d->nb; ++k)
    for (int64_t i = 0; i < n; ++i) out[i * d->nb + k] = h[static_cast<size_t>(k) * n + i];
  return 0;
}

int mi355_cl_bandpowers(mi355_ctx* c, int32_t idata, int32_t what, int64_t n, const double* in,
                        int32_t rows, int32_t L, double* bp) {
  if (check_ready(c, what == 0, "mi355_cl_bandpowers") || check_idata(c, idata, "mi355_cl_bandpowers"))
    return -1;
  MI355_REQUIRE(n > 0 && in && bp, "mi355_cl_bandpowers: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  ClData* d = p->data[idata].get();
  if (what == 0) {
    if (to_bandmajor(c, in, n, p->npar, p->npar, p->q_host_layout, p->q_pm)) return -2;
    if (p->bp.alloc(static_cast<size_t>(n) * d->nbp)) return -2;
    if (launch_bandpower(c, p, d, n, p->q_pm.p)) return -2;
  } else {
    int rc = bp_from_cl(c, d, n, in, rows, L);
    if (rc) return rc;
  }
  return bp_to_host(c, d, n, bp);
}

int mi355_cl_calc_lnlike(mi355_ctx* c, int32_t idata, int64_t n, const double* Cl, int32_t rows,
                         int32_t L, double* out) {
  if (check_ready(c, false, "mi355_cl_calc_lnlike") || check_idata(c, idata, "mi355_cl_calc_lnlike"))
    return -1;
  MI355_REQUIRE(n > 0 && Cl && out, "mi355_cl_calc_lnlike: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  ClData* d = p->data[idata].get();
  int rc = bp_from_cl(c, d, n, Cl, rows, L);
  if (rc) return rc;
  if (p->lnl.alloc(static_cast<size_t>(n))) return -2;
  rc = dataset_from_bp(c, d, p->bp.p, n, nullptr, 0, p->lnl.p);
  if (rc) return rc;
  MI355_HIP(hipMemcpyAsync(out, p->lnl.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_cl_data_op(mi355_ctx* c, int32_t idata, int32_t op, int64_t n, const double* in,
                     double* out) {
  if (check_ready(c, false, "mi355_cl_data_op") || check_idata(c, idata, "mi355_cl_data_op")) return -1;
  MI355_REQUIRE(n > 0 && in && out, "mi355_cl_data_op: bad argument");
  MI355_REQUIRE(op >= 0 && op <= 4, "mi355_cl_data_op: unknown op %d", op);
  MI355_HIP(hipSetDevice(c->device));
  ClProblem* p = c->cl;
  ClData* d = p->data[idata].get();
  const size_t nv = static_cast<size_t>(n) * d->nb;
  if (op == 0 || op == 1) {
    if (upload(c, p->tmp_in, in, nv) || p->tmp_out.alloc(nv)) return -2;
    const unsigned grid = static_cast<unsigned>(ceil_div64(nv, 256));
    if (op == 0)
      hipLaunchKernelGGL(k_cl_delta, dim3(grid), dim3(256), 0, c->stream, n, d->nb, p->tmp_in.p,
                         d->obs.p, d->xf.p, d->hasx.p, d->has_x ? 1 : 0, p->tmp_out.p);
    else
      hipLaunchKernelGGL(k_cl_linform, dim3(grid), dim3(256), 0, c->stream, n, d->nb, d->nbp,
                         p->tmp_in.p, d->has_corr ? d->ninv.p : nullptr, d->var.p, p->tmp_out.p);
    MI355_HIP(hipGetLastError());
    MI355_HIP(hipMemcpyAsync(out, p->tmp_out.p, nv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MI355_HIP(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (to_bandmajor(c, in, n, d->nb, d->nbp, p->tmp_in, p->bp)) return -2;
  if (p->lnl.alloc(static_cast<size_t>(n))) return -2;
  int rc;
  if (op == 4) {
    rc = dataset_from_bp(c, d, p->bp.p, n, nullptr, 0, p->lnl.p);
  } else {
    // getquadform(vec) on the raw rows, getchisq(BP) on the delta rows (no marginalisation)
    rc = run_form(c, d, p->bp.p, n, op == 2 ? CL_ROWS_RAW : CL_ROWS_MARGE, 1, 1, 1, p->lnl.p);
  }
  if (rc) return rc;
  MI355_HIP(hipMemcpyAsync(out, p->lnl.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
