// C ABI of the BeamFit path (include/mi355mcmc.h, "BeamFit" block).
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

#include "beam_kernels.hpp"
#include "beam_mh_kernels.hpp"
#include "problems.hpp"

namespace mi355 {

struct BeamProblem {
  int64_t M = 0, npix = 0;
  int xy_per_map = 0, sig_kind = 0;
  BeamPriorBox box{};
  DevBuf<double> x, y, d, w, sig;
  DevBuf<double> xyref;            // [1 or M][4] midpoint and half extents of the coordinates' bounding box (beam_fold)
  DevBuf<double> mono;             // one shared grid: [5][mstride] its monomials (k_beam_chi2_cl)
  int64_t mstride = 0;
  double xr = 0.0, yr = 0.0, hx = 0.0, hy = 0.0;   // reference point and half extents of the shared grid
  DevBuf<double> theta_in, theta_pm, partial, lnl, tmp;
};

void beam_destroy(BeamProblem* p) { delete p; }

int beam_problem_info(const mi355_ctx* c, int64_t* M, BeamPriorBox* box) {
  if (!c || !c->beam || c->beam->M <= 0) return -1;
  *M = c->beam->M;
  *box = c->beam->box;
  return 0;
}

static int beam_ready(mi355_ctx* c, const char* who) {
  MI355_REQUIRE(c && c->beam && c->beam->M > 0, "%s: no beam maps loaded (call mi355_beam_load)", who);
  return 0;
}

// k_beam_chi2's pixel chunks: enough blocks to fill 256 CUs several times over, at least 512 pixels each
static void beam_px_plan(int64_t nm, int ngrp, int64_t npix, int64_t* nchunk, int64_t* chunk) {
  const int64_t want = ceil_div64(4096, nm * ngrp);
  int64_t nc = std::max<int64_t>(1, std::min<int64_t>(want, ceil_div64(npix, 512)));
  *chunk = ceil_div64(ceil_div64(npix, nc), 256) * 256;
  *nchunk = ceil_div64(npix, *chunk);
}

static int beam_mf_mode() {
  static const int m = [] { const char* e = getenv("MI355_BEAM_MF"); return e ? atoi(e) : -1; }();
  return m;
}

// [n][9] -> [9][n]
__global__ void k_beam_to_pm(int64_t n, const double* __restrict__ in, double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * kBeamP) return;
  const int i = static_cast<int>(e / n);
  const int64_t k = e % n;
  out[e] = in[k * kBeamP + i];
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_beam_load(mi355_ctx* c, int64_t M, int64_t npix, const double* x, const double* y,
                    int32_t xy_per_map, const double* d, const double* sig, int32_t sig_kind,
                    const mi355_beam_prior* prior) {
  MI355_REQUIRE(c && x && y && d && sig && prior, "mi355_beam_load: null argument");
  MI355_REQUIRE(M > 0 && npix > 0, "mi355_beam_load: empty input (M=%lld, npix=%lld)",
                static_cast<long long>(M), static_cast<long long>(npix));
  MI355_REQUIRE(sig_kind >= 0 && sig_kind <= 2, "mi355_beam_load: sig_kind %d", sig_kind);
  mh_invalidate(c);
  MI355_HIP(hipSetDevice(c->device));
  auto p = std::make_unique<BeamProblem>();
  p->M = M;
  p->npix = npix;
  p->xy_per_map = xy_per_map ? 1 : 0;
  p->sig_kind = sig_kind;
  p->box.cmin[0] = prior->center_min[0];
  p->box.cmin[1] = prior->center_min[1];
  p->box.cmax[0] = prior->center_max[0];
  p->box.cmax[1] = prior->center_max[1];
  p->box.sig_min = prior->sig_min;
  p->box.sig_max = prior->sig_max;
  p->box.has_center = prior->has_center;
  p->box.has_sig_min = prior->has_sig_min;
  p->box.has_sig_max = prior->has_sig_max;
  const size_t nxy = static_cast<size_t>(xy_per_map ? M : 1) * npix;
  const size_t nd = static_cast<size_t>(M) * npix;
  const size_t ns = sig_kind == 2 ? nd : (sig_kind == 1 ? static_cast<size_t>(M) : 1);
  if (p->x.alloc(nxy) || p->y.alloc(nxy) || p->d.alloc(nd) || p->w.alloc(ns) || p->sig.alloc(ns))
    return -2;
  std::vector<double> w(ns);
  for (size_t i = 0; i < ns; ++i) {
    const double s2 = sig[i] * sig[i];                 // GaussianData.py:19
    w[i] = 1.0 / s2;
  }
  // reference point of every coordinate set: the midpoint of its finite extent
  const size_t nref = xy_per_map ? static_cast<size_t>(M) : 1;
  std::vector<double> ref(4 * nref, 0.0);                 // (xr, yr, hx, hy) per coordinate set
  for (size_t r = 0; r < nref; ++r)
    for (int ax = 0; ax < 2; ++ax) {
      const double* v = (ax ? y : x) + r * static_cast<size_t>(npix);
      double lo = INFINITY, hi = -INFINITY;
      for (int64_t i = 0; i < npix; ++i)
        if (std::isfinite(v[i])) { lo = std::fmin(lo, v[i]); hi = std::fmax(hi, v[i]); }
      if (lo <= hi) {
        ref[4 * r + ax] = lo + 0.5 * (hi - lo);
        ref[4 * r + 2 + ax] = 0.5 * (hi - lo);
      }
      if (lo <= hi && r == 0) (ax ? p->hy : p->hx) = 0.5 * (hi - lo);
    }
  if (p->xyref.alloc(ref.size())) return -2;
  MI355_HIP(hipMemcpyAsync(p->xyref.p, ref.data(), ref.size() * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(p->x.p, x, nxy * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(p->y.p, y, nxy * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(p->d.p, d, nd * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(p->w.p, w.data(), ns * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(p->sig.p, sig, ns * 8, hipMemcpyHostToDevice, c->stream));
  if (!xy_per_map) {
    p->mstride = (npix + 15) & ~static_cast<int64_t>(15);
    p->xr = ref[0];
    p->yr = ref[1];
    if (p->mono.alloc(static_cast<size_t>(5 * p->mstride))) return -2;
    hipLaunchKernelGGL(k_beam_mono, dim3(static_cast<unsigned>(ceil_div64(p->mstride, 256))), dim3(256), 0, c->stream,
                       npix, p->mstride, p->x.p, p->y.p, p->xr, p->yr, p->mono.p);
    MI355_HIP(hipGetLastError());
  }
  MI355_HIP(hipStreamSynchronize(c->stream));
  beam_destroy(c->beam);
  c->beam = p.release();
  return 0;
}

int mi355_beam_lnlike_dev_async(mi355_ctx* c, int64_t m0, int64_t nm, int32_t C,
                                const double* theta_dev, double* lnL_dev) {
  MI355_REQUIRE(lnL_dev, "mi355_beam_lnlike: null buffer");
  const double* partial = nullptr;
  int nchunk = 0;
  const int rc = beam_chi2_partials_async(c, m0, nm, C, theta_dev, &partial, &nchunk);
  if (rc) return rc;
  const int64_t n = nm * C;
  hipLaunchKernelGGL(k_beam_finish, dim3(static_cast<unsigned>(ceil_div64(n, 256))), dim3(256), 0,
                     c->stream, n, nchunk, partial, lnL_dev);
  MI355_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace mi355 {
// k_beam_chi2_mf instead of k_beam_chi2 for this job?  (beam_chi2_partials_async has the reasons)
static bool beam_takes_mf(const BeamProblem* p, int64_t n, int C) {
  const int mf_mode = beam_mf_mode();
  const int ngrp16 = (C + 15) / 16;
  return !p->xy_per_map && mf_mode != 0 && (mf_mode == 1 || (n >= 1024 && C * 4 >= ngrp16 * 16 * 3));
}

// ---- the fused sampler (beam_mh_kernels.hpp) ----
constexpr size_t kBeamMhLdsBudget = 150 * 1024;            // of the CU's 160 KB, static tables included
constexpr int kBeamMhMaxChunks = 512;
struct BeamMhPlan {
  bool ok = false;
  int64_t nchunk = 0, chunk = 0;
  int nsub = 1, px_lds = 0;
  size_t lds = 0;
};
static BeamMhPlan beam_mh_plan(const mi355_ctx* c) {
  BeamMhPlan r;
  const BeamProblem* p = c->beam;
  if (!p || p->M <= 0 || c->mhS != p->M || c->C > kBeamCH || c->mhP != kBeamP) return r;
  if (beam_takes_mf(p, c->N, c->C)) return r;              // another summation order: not this kernel's chains
  beam_px_plan(c->mhS, 1, p->npix, &r.nchunk, &r.chunk);
  if (r.nchunk > kBeamMhMaxChunks) return r;
  // Where it is the faster path (tools/scan_beam_mh.py, us per step fused / step-wise): a block walks its map's
  // chunks two at a time on one CU where the engine's chi-square kernel has a CU per chunk, so the fused kernel
  // wins on maps of one or two chunks (1024 pixels, one chain: 6.7 / 10.8 on one map, 6.4 / 13.6 on 256 maps;
  // eight chains: 12.4 / 14.6) and on jobs of some hundred maps, whose launches the engine runs one behind
  // the other (256 maps of 4096 pixels: 14.3 / 28.0 with one chain, 33.6 / 44.1 with eight; of 16 384 pixels:
  // 35.5 / 54.6 ... 89 / 95) - and loses on a few maps of many chunks (one 64 x 64 map, one chain: 14.6 / 11.1;
  // 16 384 pixels: 53 / 13) and from about a thousand maps up (1024 maps x 8 chains: 38 / 25 at 256 pixels).
  // mi355_mh_set_fused(ctx, 2) takes it wherever it serves the job.
  if (c->mh_fused != 2) {
    const bool small_maps = r.nchunk <= 2 && c->mhS <= 512;
    const bool many_maps = c->mhS >= 256 && c->mhS <= 512 && p->npix <= 16384;
    if (!small_maps && !many_maps) return r;
  }
  // (groups of 256 threads per block: with four the kernel's registers spill - 128 per lane - and a round of
  // chunks takes 2.3 x as long as with two)
  static const int max_sub = [] { const char* e = getenv("MI355_BEAM_MH_SUB"); return e ? atoi(e) : 2; }();
  r.nsub = (r.nchunk >= 4 && max_sub >= 4) ? 4 : ((r.nchunk >= 2 && max_sub >= 2) ? 2 : 1);
  const size_t fixed = sizeof(BeamMhLds) + kStaticTabBytes;
  // the map's pixels in LDS when they fit and every map has a CU of its own
  const size_t with_px = beam_mh_lds_doubles(static_cast<int>(r.nchunk), p->npix, p->sig_kind, p->xy_per_map, 1) * sizeof(double);
  r.px_lds = (c->mhS <= 256 && with_px + fixed <= kBeamMhLdsBudget) ? 1 : 0;
  r.lds = beam_mh_lds_doubles(static_cast<int>(r.nchunk), p->npix, p->sig_kind, p->xy_per_map, r.px_lds) * sizeof(double);
  r.ok = r.lds + fixed <= kBeamMhLdsBudget;
  return r;
}
bool beam_mh_fused_ok(const mi355_ctx* c) { return c && c->beam && beam_mh_plan(c).ok; }

template <int NSUB, int NCHT>
static int launch_beam_mh2(mi355_ctx* c, const BeamMhArgs& a, size_t lds) {
  if (lds > 48 * 1024)
    MI355_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_beam_mh<NSUB, NCHT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL((k_beam_mh<NSUB, NCHT>), dim3(static_cast<unsigned>(c->mhS)), dim3(256 * NSUB), lds, c->stream, a);
  MI355_HIP(hipGetLastError());
  return 0;
}
template <int NSUB>
static int launch_beam_mh(mi355_ctx* c, const BeamMhArgs& a, size_t lds) {
  if (c->C == 1) return launch_beam_mh2<NSUB, 1>(c, a, lds);
  if (c->C <= 4) return launch_beam_mh2<NSUB, 4>(c, a, lds);
  return launch_beam_mh2<NSUB, kBeamCH>(c, a, lds);
}

int beam_mh_run(mi355_ctx* c, const GmhArgs& g, int64_t nsteps, int thin, int64_t nkeep) {
  if (beam_ready(c, "mi355_mh_run")) return -1;
  const BeamMhPlan plan = beam_mh_plan(c);
  MI355_REQUIRE(plan.ok, "the fused beam sampler does not serve this job (step-wise engine only)");
  BeamProblem* p = c->beam;
  BeamMhArgs a{};
  a.g = g;
  a.b.x = p->x.p;
  a.b.y = p->y.p;
  a.b.xy_per_map = p->xy_per_map;
  a.b.xyref = p->xyref.p;
  a.b.d = p->d.p;
  a.b.w = p->w.p;
  a.b.w_kind = p->sig_kind;
  a.b.npix = p->npix;
  a.b.m0 = 0;
  a.b.C = c->C;
  a.b.n = c->N;
  a.b.chunk = static_cast<int>(plan.chunk);
  a.b.nchunk = static_cast<int>(plan.nchunk);
  a.nsteps = nsteps;
  a.thin = thin;
  a.nkeep = nkeep;
  a.px_lds = plan.px_lds;
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
  int rc;
  if (plan.nsub == 4) rc = launch_beam_mh<4>(c, a, plan.lds);
  else if (plan.nsub == 2) rc = launch_beam_mh<2>(c, a, plan.lds);
  else rc = launch_beam_mh<1>(c, a, plan.lds);
  if (rc) return rc;
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
  if (!c->capturing) c->ev_valid = true;
  return 0;
}

// The chi-square kernels alone: partial[n][nchunk] (lnL = -1/2 of a row's sum in ascending order:
// k_beam_finish, or the accept step of the Metropolis-Hastings engine itself)
int beam_chi2_partials_async(mi355_ctx* c, int64_t m0, int64_t nm, int32_t C, const double* theta_dev,
                             const double** partial_out, int* nchunk_out) {
  if (beam_ready(c, "mi355_beam_lnlike")) return -1;
  BeamProblem* p = c->beam;
  MI355_REQUIRE(theta_dev, "mi355_beam_lnlike: null buffer");
  MI355_REQUIRE(m0 >= 0 && nm > 0 && m0 + nm <= p->M, "map range [%lld,%lld) outside [0,%lld)",
                static_cast<long long>(m0), static_cast<long long>(m0 + nm),
                static_cast<long long>(p->M));
  MI355_REQUIRE(C > 0, "C must be positive");
  MI355_HIP(hipSetDevice(c->device));
  const int64_t n = nm * C;
  // Many maps on one shared grid with (nearly) full groups of 16 chains: the quadratic form on the matrix
  // cores (k_beam_chi2_mf); otherwise a lane per pixel and eight chains per block (k_beam_chi2: one map
  // and one chain is the reference's own use).  MI355_BEAM_MF: 0 never, 1 wherever the grid is shared.
  // (A value differs between the two kernels in the last bits - 1e-13 relative, summation order - so a
  // job split into launches of very different sizes is not bit-for-bit the job in one launch.)
  const int ngrp16 = (C + 15) / 16;
  if (beam_takes_mf(p, n, C)) {
    const int64_t nunit = nm * ngrp16;
    // pixel runs per unit: about 8192 waves in all (the time is flat from 2048 to 16 384 on C5), runs
    // of at least 256 pixels
    int64_t W = std::max<int64_t>(1, std::min<int64_t>(8192 / nunit, ceil_div64(p->npix, 256)));
    const int64_t run = ceil_div64(ceil_div64(p->npix, W), 16) * 16;
    W = ceil_div64(p->npix, run);
    if (p->partial.alloc(static_cast<size_t>(n) * W)) return -2;
    BeamMfArgs b{};
    b.mono = p->mono.p;
    b.mstride = p->mstride;
    b.x = p->x.p;
    b.y = p->y.p;
    b.d = p->d.p;
    b.w = p->w.p;
    b.w_kind = p->sig_kind;
    b.npix = p->npix;
    b.m0 = m0;
    b.C = C;
    b.ngrp = ngrp16;
    b.n = n;
    b.nunit = nunit;
    b.theta = theta_dev;
    b.xr = p->xr;
    b.yr = p->yr;
    b.hx = p->hx;
    b.hy = p->hy;
    b.W = static_cast<int>(W);
    b.run = run;
    b.partial = p->partial.p;
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k_beam_chi2_mf, dim3(static_cast<unsigned>(ceil_div64(nunit * W, 4))), dim3(256), 0, c->stream, b);
    MI355_HIP(hipGetLastError());
    if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
    if (!c->capturing) c->ev_valid = true;
    *partial_out = p->partial.p;
    *nchunk_out = static_cast<int>(W);
    return 0;
  }
  const int ngrp = (C + kBeamCH - 1) / kBeamCH;
  int64_t nchunk, chunk;
  beam_px_plan(nm, ngrp, p->npix, &nchunk, &chunk);
  if (p->partial.alloc(static_cast<size_t>(n) * nchunk)) return -2;
  BeamArgs a{};
  a.x = p->x.p;
  a.y = p->y.p;
  a.xy_per_map = p->xy_per_map;
  a.xyref = p->xyref.p;
  a.d = p->d.p;
  a.w = p->w.p;
  a.w_kind = p->sig_kind;
  a.npix = p->npix;
  a.C = C;
  a.n = n;
  a.theta = theta_dev;
  a.chunk = static_cast<int>(chunk);
  a.nchunk = static_cast<int>(nchunk);
  a.partial = p->partial.p;
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev0, c->stream));
  // blockIdx.y is limited to 65535: slabs of maps
  for (int64_t s0 = 0; s0 < nm; s0 += 32768) {
    const int64_t ns = std::min<int64_t>(32768, nm - s0);
    BeamArgs b = a;
    b.m0 = m0 + s0;
    b.theta = theta_dev + s0 * C;           // vector e = (m - b.m0)*C + c, stride n between params
    b.partial = p->partial.p + s0 * C * nchunk;
    hipLaunchKernelGGL(k_beam_chi2, dim3(static_cast<unsigned>(nchunk), static_cast<unsigned>(ns), ngrp),
                       dim3(256), 0, c->stream, b);
    MI355_HIP(hipGetLastError());
  }
  if (!c->capturing) MI355_HIP(hipEventRecord(c->ev1, c->stream));
  if (!c->capturing) c->ev_valid = true;
  *partial_out = p->partial.p;
  *nchunk_out = static_cast<int>(nchunk);
  return 0;
}
}  // namespace mi355

extern "C" {

int mi355_beam_lnlike(mi355_ctx* c, int64_t m0, int64_t nm, int32_t C, const double* theta,
                      double* lnL) {
  if (beam_ready(c, "mi355_beam_lnlike")) return -1;
  MI355_REQUIRE(theta && lnL && nm > 0 && C > 0, "mi355_beam_lnlike: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  BeamProblem* p = c->beam;
  const int64_t n = nm * C;
  const size_t nt = static_cast<size_t>(n) * kBeamP;
  if (p->theta_in.alloc(nt) || p->theta_pm.alloc(nt) || p->lnl.alloc(static_cast<size_t>(n))) return -2;
  MI355_HIP(hipMemcpyAsync(p->theta_in.p, theta, nt * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_beam_to_pm, dim3(static_cast<unsigned>(ceil_div64(nt, 256))), dim3(256), 0,
                     c->stream, n, p->theta_in.p, p->theta_pm.p);
  MI355_HIP(hipGetLastError());
  int rc = mi355_beam_lnlike_dev_async(c, m0, nm, C, p->theta_pm.p, p->lnl.p);
  if (rc) return rc;
  MI355_HIP(hipMemcpyAsync(lnL, p->lnl.p, n * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_beam_prior_package(mi355_ctx* c, int64_t n, const double* theta, double* prior,
                             double* packaged) {
  if (beam_ready(c, "mi355_beam_prior_package")) return -1;
  MI355_REQUIRE(theta && n > 0, "mi355_beam_prior_package: empty batch");
  MI355_HIP(hipSetDevice(c->device));
  BeamProblem* p = c->beam;
  const size_t nt = static_cast<size_t>(n) * kBeamP;
  if (p->theta_in.alloc(nt) || p->tmp.alloc(nt + n)) return -2;
  MI355_HIP(hipMemcpyAsync(p->theta_in.p, theta, nt * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_beam_prior_package, dim3(static_cast<unsigned>(ceil_div64(n, 256))), dim3(256),
                     0, c->stream, n, p->theta_in.p, p->box, prior ? p->tmp.p + nt : nullptr,
                     packaged ? p->tmp.p : nullptr);
  MI355_HIP(hipGetLastError());
  if (prior) MI355_HIP(hipMemcpyAsync(prior, p->tmp.p + nt, n * 8, hipMemcpyDeviceToHost, c->stream));
  if (packaged) MI355_HIP(hipMemcpyAsync(packaged, p->tmp.p, nt * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_beam_model(mi355_ctx* c, int64_t n, const double* theta, int64_t npts, const double* x,
                     const double* y, double* out) {
  MI355_REQUIRE(c && theta && x && y && out && n > 0 && npts > 0, "mi355_beam_model: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  const size_t nt = static_cast<size_t>(n) * kBeamP, no = static_cast<size_t>(n) * npts;
  if (c->scratch_a.alloc(nt + 2 * npts) || c->scratch_b.alloc(no)) return -2;
  double* dt = c->scratch_a.p;
  double* dx = dt + nt;
  double* dy = dx + npts;
  MI355_HIP(hipMemcpyAsync(dt, theta, nt * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(dx, x, npts * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(dy, y, npts * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_beam_model, dim3(static_cast<unsigned>(ceil_div64(no, 256))), dim3(256), 0,
                     c->stream, n, dt, npts, dx, dy, c->scratch_b.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(out, c->scratch_b.p, no * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_beam_gauss2d(mi355_ctx* c, int64_t n, const double* x, const double* y, double x0,
                       double y0, double cinv_xx, double cinv_xy, double cinv_yy, double* out) {
  MI355_REQUIRE(c && x && y && out && n > 0, "mi355_beam_gauss2d: bad argument");
  MI355_HIP(hipSetDevice(c->device));
  const size_t np = static_cast<size_t>(n);
  if (c->scratch_a.alloc(2 * np) || c->scratch_b.alloc(np)) return -2;
  double* dx = c->scratch_a.p;
  double* dy = dx + np;
  MI355_HIP(hipMemcpyAsync(dx, x, np * 8, hipMemcpyHostToDevice, c->stream));
  MI355_HIP(hipMemcpyAsync(dy, y, np * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_beam_gauss2d, dim3(static_cast<unsigned>(ceil_div64(n, 256))), dim3(256), 0,
                     c->stream, n, dx, dy, x0, y0, cinv_xx, cinv_xy, cinv_yy, c->scratch_b.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(out, c->scratch_b.p, np * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mi355_beam_stats(mi355_ctx* c, int64_t m, int32_t has_sigcut, double sigcut, int32_t do_abs,
                     double* out) {
  if (beam_ready(c, "mi355_beam_stats")) return -1;
  BeamProblem* p = c->beam;
  MI355_REQUIRE(out && m >= 0 && m < p->M, "mi355_beam_stats: map %lld outside [0,%lld)",
                static_cast<long long>(m), static_cast<long long>(p->M));
  MI355_HIP(hipSetDevice(c->device));
  if (p->tmp.alloc(4)) return -2;
  const double* xs = p->x.p + (p->xy_per_map ? m * p->npix : 0);
  const double* ys = p->y.p + (p->xy_per_map ? m * p->npix : 0);
  const double* sg = p->sig.p + (p->sig_kind == 2 ? m * p->npix : (p->sig_kind == 1 ? m : 0));
  hipLaunchKernelGGL(k_beam_stats, dim3(1), dim3(256), 0, c->stream, p->npix, xs, ys,
                     p->d.p + m * p->npix, sg, p->sig_kind == 2 ? 1 : 0, has_sigcut, sigcut, do_abs,
                     p->tmp.p);
  MI355_HIP(hipGetLastError());
  MI355_HIP(hipMemcpyAsync(out, p->tmp.p, 4 * 8, hipMemcpyDeviceToHost, c->stream));
  MI355_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
