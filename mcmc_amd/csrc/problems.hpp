// What the step-wise Metropolis-Hastings engine (gmh_api.hip) needs to know about the problem
// kinds that live in other translation units.
#pragma once
#include "beam_device.hpp"
#include "common.hpp"

namespace mi355 {

// binnedCl: number of parameters and the spectrum type of each (device array), cl_api.hip
int cl_problem_info(const mi355_ctx* c, int* npar, const int32_t** cltype_dev);
// binnedCl: the fused Metropolis-Hastings kernels (cl_mh_kernels.hpp, cl_mhw_kernels.hpp; host side in
// cl_mh_api.hip).  cl_mh_fused_nw: > 0 if one of them serves the loaded problem and the job of the context
// (its chain count, `comom`: running co-moments), 0 if it needs the step-wise engine (too many data sets or
// parameters, LDS, an 11 x 11 grid on very many chains).  cl_mh_run:
// nsteps steps of every chain in one launch on the context stream; `a` carries the chain state,
// the counters of the first step and, as bases of the whole run, the injected draws and the trace.
struct GmhArgs;
int cl_mh_fused_nw(const mi355_ctx* c, int P, int nblock, int npn, int comom);
int cl_mh_run(mi355_ctx* c, const GmhArgs& a, int64_t nsteps, int thin, int64_t nkeep);
// BeamFit: number of maps and the prior box, beam_api.hip
int beam_problem_info(const mi355_ctx* c, int64_t* M, BeamPriorBox* box);
// BeamFit: the chi-square kernels of mi355_beam_lnlike_dev_async without their finish: partial sums
// [nm * C][nchunk] on the device (the engine's accept step adds a row up itself: GmhArgs::lnl_partial)
int beam_chi2_partials_async(mi355_ctx* c, int64_t m0, int64_t nm, int32_t C, const double* theta_dev,
                             const double** partial_out, int* nchunk_out);
// BeamFit: the fused Metropolis-Hastings kernel (beam_mh_kernels.hpp), one block per map.  beam_mh_fused_ok:
// it serves the chains of the context (at most 8 chains per map, a job the chi-square of which the step-wise
// engine runs through k_beam_chi2, chunk sums that fit LDS).  beam_mh_run: as cl_mh_run.
bool beam_mh_fused_ok(const mi355_ctx* c);
int beam_mh_run(mi355_ctx* c, const GmhArgs& a, int64_t nsteps, int thin, int64_t nkeep);

}  // namespace mi355
