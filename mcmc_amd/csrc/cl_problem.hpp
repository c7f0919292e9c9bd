// Host-side state of the binnedCl / ClData problem, shared by cl_api.hip (the batched likelihood and its C
// ABI) and cl_mh_api.hip (the fused Metropolis-Hastings samplers, a translation unit of their own: they are
// compiled without machine-level loop-invariant code motion, which the batched form kernel does not want).
#pragma once
#include <memory>
#include <vector>

#include "cl_kernels.hpp"
#include "cl_mh_kernels.hpp"

namespace mi355 {

struct ClData {
  int nb = 0, nbp = 0, wl = 0;
  bool has_x = false, has_corr = false, beam_uncertain = false, has_pol = false;
  double calib = 0.0;
  DevBuf<double> obs, xf, beam_err, bfac, ninv, nsym, var, window, K, bp0;
  DevBuf<int32_t> krange;          // [nbp/16][2] non-zero k-step range of K per column tile (cl_bp_tile)
  int kmax = 1;                    // widest of those ranges, at least 1 (rows of the compact LDS copy: 4 * kmax per tile)
  DevBuf<uint8_t> hasx, inc_pol;
  DevBuf<int32_t> win_min, win_max;
};

struct ClProblem {
  std::vector<std::unique_ptr<ClData>> data;
  // binning (binnedClModel class state)
  bool binned = false;
  int nCl = 0, npar = 0, lmax = 0;
  DevBuf<double> shapefun;
  DevBuf<int32_t> binof, cltype;
  DevBuf<double> weights;          // margeweights
  double margenorm = 0.0;
  // per-call scratch
  DevBuf<double> q_host_layout, q_pm, bp, form, lnl, tmp_in, tmp_out;
  // fused Metropolis-Hastings kernel (cl_mh_kernels.hpp), planned by mi355_cl_set_binning
  bool mh_filled = false;          // mh[] describes the loaded data sets (plan_fused)
  int mh_nw = 0;                   // waves per block of k_cl_mh_run; 0: that kernel does not serve the problem
  int mh_nbp_max = 0, mh_fout = 9, mh_tile_doubles = 0;
  ClMhData mh[kClMhMaxData] = {};
  DevBuf<ClMhArgs> mh_args;        // the kernel reads its arguments from device memory
};

// rows of the quadratic-form stage for one data set
struct FormPlan {
  int mode, ncal, nbeam, rows_per_chain;
};

inline FormPlan plan_for(const ClData* d) {
  const bool marge = d->calib > 1.0e-4 || d->beam_uncertain;        // :282, :288
  FormPlan f{CL_ROWS_MARGE, 1, 1, 1};
  if (!marge) return f;
  if (d->has_x) {
    f.ncal = (d->calib == 0.0) ? 1 : kClGrid;   // 11 equal points collapse exactly (k_cl_combine)
    f.nbeam = d->beam_uncertain ? kClGrid : 1;
    f.rows_per_chain = f.ncal * f.nbeam;
  } else {
    f.mode = CL_ROWS_TRIPLE;
    f.rows_per_chain = 3;
  }
  return f;
}

inline size_t form_lds_bytes(int nbp, int TR, int NW = 8) {
  // row tile + reduction scratch (at least 16 rows of it: the row fragments are fetched up to
  // four k-steps past the tile) + row tables
  const size_t red = static_cast<size_t>(NW) * TR * 3 > static_cast<size_t>(16) * TR
                         ? static_cast<size_t>(NW) * TR * 3 : static_cast<size_t>(16) * TR;
  return (static_cast<size_t>(nbp) * TR + red) * sizeof(double) + 2 * TR * sizeof(int);
}

constexpr size_t kClLdsBudget = 144 * 1024;   // dynamic LDS of a block, of the 160 KB per CU (the Cl kernels have no static LDS)

// Waves per block of the form kernel launch_form_cg picks for a data set with nbp padded bands:
// the fused kernel must run the same instantiation to produce the same bits.
inline int form_nw(int nbp) { return (nbp >> 4) <= 4 ? 4 : 8; }

// cl_mh_api.hip: fills p->mh[] for the fused samplers (called when the binning is set)
void cl_plan_fused(ClProblem* p);

}  // namespace mi355
