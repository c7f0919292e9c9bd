/* mi355mcmc.h - C ABI of libmi355mcmc.so: the MI355X (gfx950) implementation of the
 * Metropolis-Hastings likelihood hot path of defjaf/MCMC.
 *
 * The reference is pure Python and has no FFI: its "operator API" is the duck-typed
 * protocol between MCMC.MCMC and its like / model / prop objects.
 * Each entry point below names the reference call (file:line under the reference tree)
 * whose batched equivalent it computes; mcmc_amd/ (Python, ctypes) keeps the reference's
 * class names and method signatures on top of it.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; mi355_last_error() then returns a
 *    thread-local human-readable message.  No entry point ever falls back to a CPU path.
 *  - "host" pointers are caller-owned host memory, "dev" pointers are device memory on the
 *    context's GPU (e.g. torch tensor data_ptr()).  All arrays are FP64 unless typed.
 *  - a context is bound to one GPU and one HIP stream; it is not thread-safe.  Calls are
 *    synchronous on return unless the name ends in _async.
 *  - chain index n = s*C + c (source-major); parameter vectors are "flat" (unpackaged).
 */
#ifndef MI355MCMC_H
#define MI355MCMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 3

typedef struct mi355_ctx mi355_ctx;

/* ---- context ------------------------------------------------------------------------- */
/* seed keys the Philox4x32-10 streams that replace numpy's global MT19937
 * (MCMC/Proposal.py:3,64,151; MCMC/MCMC.py:37,113,264). */
int mi355_init(int device, uint64_t seed, mi355_ctx** out);
void mi355_free(mi355_ctx* ctx);
const char* mi355_last_error(void);
int mi355_abi_version(void);
/* Duration in ms of the main kernel of the last mi355_mh_run / mi355_sed_lnlike* call, measured
 * with HIP events recorded on the context stream around that launch (waits for it to finish).
 * Returns <0 if no such kernel has been launched. */
double mi355_last_kernel_ms(mi355_ctx* ctx);
/* the HIP stream (hipStream_t) all kernels of this context are launched on */
void* mi355_stream(mi355_ctx* ctx);
int mi355_sync(mi355_ctx* ctx);

/* ---- submmSED models: MCMC/submmSED/model.py ------------------------------------------- */
enum mi355_sed_model_id {
  MI355_SED_M1N = 0,        /* submmModel1_normalized            model.py:529-616  (A, beta, T)            */
  MI355_SED_M1N_LOGA = 1,   /* submmModel1_normalized_logA       model.py:748-769  (log10 A, beta, T)      */
  MI355_SED_M2N = 2,        /* submmModel2_normalized            model.py:363-468  (A1,b1,T1,A2,b2,T2)     */
  MI355_SED_M2N_LOGA = 3,   /* submmModel2_normalized_logA       model.py:470-526                           */
  MI355_SED_M1 = 4,         /* submmModel1 + SEDLikelihood1      model.py:218-279, likelihood.py:93-123    */
  MI355_SED_M2 = 5,         /* submmModel2 + SEDLikelihood2      model.py:130-215, likelihood.py:18-88     */
  MI355_SED_THICK = 6,      /* submmModel1_opticallythick        model.py:620-705  (A, beta, T, nu0)       */
  MI355_SED_THICK_LOGA = 7, /* submmModel1_opticallythick_logA   model.py:709-746  (inherits the plain
                               greybody at(): nu0 enters the prior only - reproduced)                      */
  MI355_SED_NMODEL = 8
};

/* Module-level constants of model.py:55-61,638 captured per context ("compile-time" flags
 * of the reference become fields here). */
typedef struct {
  int32_t model;              /* enum mi355_sed_model_id */
  int32_t nparam;             /* must equal the model's parameter count (checked)  */
  double h_over_k;            /* 0.04799237 K/GHz   model.py:55 */
  double nu_b;                /* 1000 GHz           model.py:57 */
  double prefac;              /* 1e-9               model.py:56 */
  double Tmin, Tmax;          /* 3, 100 K           model.py:59 */
  double bmin, bmax;          /* 0, 3               model.py:61 */
  double nu_rescale;          /* 1000               model.py:638 */
} mi355_sed_model;

/* Default constants of the reference for `model`. */
int mi355_sed_model_default(int32_t model, mi355_sed_model* out);
int mi355_sed_nparam(int32_t model);

/* Load a catalogue of S sources (submmData / GaussianData: submmSED/data.py:30-45,
 * GaussianData.py:13-28).  nb[s] <= nb_max photometric bands per source; nu (rest-frame
 * GHz), d (flux) and sig are [S, nb_max] row-major, entries past nb[s] ignored.
 * Upper limits are encoded by the caller as d=0, sig=flux (data.py:233-237). */
int mi355_sed_load(mi355_ctx* ctx, const mi355_sed_model* model, int64_t S, int32_t nb_max,
                   const int32_t* nb, const double* nu, const double* d, const double* sig);

/* How many of the loaded sources evaluate through the clamp-free base-2 kernels (all of them for
 * far-infrared / submillimetre photometry).  A source with a band outside those kernels' domain
 * over the prior box (h nu/k Tmax < 0.02: a radio point; h nu/k Tmin >= 300: a mid-infrared one)
 * evaluates through the exact OCML formulas of model.py instead - by itself: the other sources of
 * the catalogue keep the fast kernels, and mi355_mh_run runs the two groups side by side.  Both
 * paths hold the same 1e-12 against the reference.  -1: no catalogue loaded. */
int64_t mi355_sed_fast_sources(mi355_ctx* ctx);

/* like.lnLike(theta) (Likelihood.py:99-105 -> submmSED/likelihood.py:141-155, :99-110,
 * :35-74) for sources [s0, s0+ns) x C parameter vectors each.
 * theta: host [ns, C, P]; lnL: host [ns, C]; derived (nullable): host [ns, C, 2], the
 * maximum-likelihood amplitudes of SEDLikelihood1/2.getDerived (likelihood.py:77-88,113-123),
 * NaN-filled for the explicit-amplitude models. */
int mi355_sed_lnlike(mi355_ctx* ctx, int64_t s0, int64_t ns, int32_t C, const double* theta,
                     double* lnL, double* derived);
/* Same on device memory, no copies: theta_dev is parameter-major [P, ns*C], lnL_dev [ns*C],
 * derived_dev nullable [2, ns*C].  Asynchronous on the context stream. */
int mi355_sed_lnlike_dev_async(mi355_ctx* ctx, int64_t s0, int64_t ns, int32_t C,
                               const double* theta_dev, double* lnL_dev, double* derived_dev);
/* model(*theta).at_nu(nu) (model.py at_nu/at methods): flux of n parameter vectors (host [n,P])
 * at nnu frequencies (host [nnu], rest-frame GHz) -> out host [n, nnu, ncol], ncol = 2 for
 * submmModel2 (one column per component, model.py:176-182) and 1 otherwise. */
int mi355_sed_flux(mi355_ctx* ctx, int64_t n, const double* theta, int64_t nnu, const double* nu,
                   double* out);
/* model.prior(*theta) (model.py prior classmethods) and model.package(theta)
 * (model.py:192-196,439-443,510-514) for n flat vectors: host [n,P] -> prior[n], packaged[n,P]
 * (either output nullable). */
int mi355_sed_prior_package(mi355_ctx* ctx, int64_t n, const double* theta, double* prior,
                            double* packaged);

/* ---- GaussianData: MCMC/GaussianData.py:31-64 ------------------------------------------ */
/* quadform(A,B) = A^T N^-1 B with N = diag(sig2), for m rows at once: host A [m,n];
 * B host [m,n] (b_rows=m), [n] (b_rows=1, shared by all rows) or null (= A); sig2 host [n]
 * (nsig=n) or scalar (nsig=1); out host [m].  chi2(vec) is quadform(d-vec). */
int mi355_gauss_quadform(mi355_ctx* ctx, int64_t m, int64_t n, const double* A, const double* B,
                         int64_t b_rows, const double* sig2, int64_t nsig, double* out);

/* ---- Metropolis-Hastings: MCMC/MCMC.py, MCMC/Proposal.py -------------------------------- */
/* Create C chains per loaded source (N = S*C chains) = N instances of
 * MCMC.MCMC(like, startProposal=sigma, startParams=theta0, doBlock=nblock>0) (MCMC.py:61-134).
 *  theta0 : host [P] (theta0_per_chain=0) or [S,C,P] (=1)
 *  sigma  : host [P] (sigma_per_chain=0) or [S,C,P] (=1); sigma=0 fixes a parameter
 *  param_blocks: host [P] block id per parameter (model.paramBlocks), nblock = model.nBlock;
 *           nblock=0 means doBlock=False (all parameters move every step)
 *  randomize_start: 1 = setStart(random=True) (MCMC.py:151-159): redraw theta0 + sigma*N(0,1)
 *           over all parameters until prior>0, with the context's Philox start stream;
 *           0 = start exactly at theta0 (the injected-draw parity path; prior must be >0). */
int mi355_mh_init(mi355_ctx* ctx, int32_t C, const double* theta0, int32_t theta0_per_chain,
                  const double* sigma, int32_t sigma_per_chain, const int32_t* param_blocks,
                  int32_t nblock, int32_t randomize_start);

/* MC_append(nsteps) for every chain (MCMC.py:196-221 -> sample() :225-286 ->
 * getNewParams Proposal.py:127-164).  State stays in HBM between calls.
 *  thin : keep every thin-th step in the trace buffer (0 = no trace)
 *  inj_normals  (nullable) host [nsteps, N, P]: the P normals Proposal.py:151 draws per step
 *  inj_uniforms (nullable) host [nsteps, N]   : the uniform of MCMC.py:264 for that step
 *  (both or neither; with them the chain is draw-for-draw the reference's). */
int mi355_mh_run(mi355_ctx* ctx, int64_t nsteps, int32_t thin, const double* inj_normals,
                 const double* inj_uniforms);

/* Incremental evaluation for mi355_mh_run (off by default).  The reference rebuilds
 * model(*params)(data) from scratch at every step (Likelihood.py:70-75 -> model.py:394-396); with
 * one parameter per block (doBlock, MCMC.py:230-234) a step of a two-component normalised model
 * changes one of (A, beta, T) of one component, and the other component's band fluxes - and for
 * an amplitude step every band flux - are those of the current state.  With on=1 the fused SED
 * sampler keeps them in registers and evaluates only what the moved parameter touches; lnLike of
 * the proposed point is the same function of theta' (rounding differs in the last bits), the
 * draws are the same, so the chain is the one the full evaluation produces (held to the same
 * 1e-12 against the reference's traces).  Applies when: submmModel2_normalized[_logA], the
 * catalogue passed the fast-path check of mi355_sed_load, no source has more than 8 bands, mi355_mh_init was given
 * param_blocks = 0..P-1; otherwise mi355_mh_run evaluates in full as before.
 * mi355_mh_incremental_active: 1 if the next mi355_mh_run of this context would use it. */
int mi355_mh_set_incremental(mi355_ctx* ctx, int32_t on);
int mi355_mh_incremental_active(mi355_ctx* ctx);

/* Lane mapping of the fused SED sampler.  lane = chain (a lane walks the nb bands of its source)
 * fills the GPU once the job has chains enough; lanes = (chain, band) - G = 8 or 32 lanes hold one
 * chain and each evaluates one band, the chi-square terms are then added in band order - shortens a
 * step from nb band evaluations to one and is what a small job wants (one source, a handful of
 * chains: the reference's own use).  mode 0 (default): (chain, band) when chains x G <= 65 536 lanes
 * (one wave per SIMD of the GPU), lane = chain beyond; 1 / 2 force either.  The chains are the same
 * bit for bit under both mappings.  (chain, band) applies to catalogues with at most 32 bands per
 * source and at least one source on the fast path (not SEDLikelihood2; the chain blocks of sources
 * outside the fast domain run the exact formulas band-parallel beside the others);
 * mi355_mh_lane_group: the G the next mi355_mh_run of this context would use (1, 8 or 32). */
int mi355_mh_set_lane_map(mi355_ctx* ctx, int32_t mode);
int mi355_mh_lane_group(mi355_ctx* ctx);

/* mean(burn=1), stdev(burn=1)**2 (ddof=0, MCMC.py:290-307) over the steps run since the last
 * init/adapt, naccept (MCMC.py:179,269), per-block accept counts (:181,271) and the current
 * lnPr / position.  Host outputs, all nullable: mean,var,theta [S,C,P]; naccept,lnPr [S,C];
 * accept [S,C,nblock]. */
int mi355_mh_moments(mi355_ctx* ctx, double* mean, double* var, int64_t* naccept,
                     int64_t* accept, double* lnPr, double* theta);
/* Current proposal widths prop.sigmas of every chain (they differ per chain after
 * mi355_mh_adapt): host [S,C,P]. */
int mi355_mh_sigmas(mi355_ctx* ctx, double* sigma);
/* Philox stream position of the context: which=0 the global step counter (normal index =
 * step*npn + rank, uniform index = step), which=1 the stage counter (number of init/adapt calls
 * so far; keys the start-randomisation stream).  Counter-based draws make a resumed run exact:
 * restore both with mi355_mh_set_stream_pos before mi355_mh_init. */
int64_t mi355_mh_stream_pos(mi355_ctx* ctx, int32_t which);
int mi355_mh_set_stream_pos(mi355_ctx* ctx, int64_t step, int64_t stage);
/* Number of steps accumulated in the moments, and kept trace rows. */
int64_t mi355_mh_nsamples(mi355_ctx* ctx);
int64_t mi355_mh_ntrace(mi355_ctx* ctx);
/* Trace of the last mi355_mh_run: samples host [nkeep,S,C,P], lnPr [nkeep,S,C],
 * accepted [nkeep,S,C] (MCMC.samples / .lnPr / .accepted rows, MCMC.py:217-219). */
int mi355_mh_trace(mi355_ctx* ctx, double* samples, double* lnPr, uint8_t* accepted);

/* newMCMC(burn=1, fac, random=random_restart, noCorrelations=True) for every chain
 * (MCMC.py:343-413): restart at package(mean) [+ N(0,1)*stdev], sigma = package(fac*stdev)
 * (fac<=0 means 2.4/sqrt(nparam), :379-380), counters and moments reset, start randomised
 * as in mi355_mh_init(randomize_start=1). */
int mi355_mh_adapt(mi355_ctx* ctx, double fac, int32_t random_restart);

/* Fill host buffers with the context's Philox streams (diagnostic; lets a test replay the
 * free-running draws through the oracle): normals [nsteps, N, npn], uniforms [nsteps, N]
 * for steps [step0, step0+nsteps) of chains [0,N). */
int mi355_rng_dump(mi355_ctx* ctx, int64_t step0, int64_t nsteps, int64_t N, int32_t npn,
                   double* normals, double* uniforms);

/* Evaluate one of the library's own FP64 kernels elementwise on host arrays (diagnostic, used by
 * the accuracy tests): fn 0 exp(a), 1 expm1(a), 2 exp(a*(b+c)) with b+c a double-double
 * logarithm (= base^a), 3 a/b, 4 10^a, 5 ln(a) (a > 0, normal), 6 sin(2 pi a), 7 cos(2 pi a)
 * (a in [0,1)), 8 2^(a*(b+c)) with b+c a double-double base-2 logarithm (= base^a, the kernel
 * the one-off powers of the SED fast path use), 9 2^a by the table-driven kernel of the band and
 * pixel loops, 10 2^a by the plain polynomial kernel (|a| < 1000).  Inputs not used by fn may be
 * null. */
int mi355_math_probe(mi355_ctx* ctx, int32_t fn, int64_t n, const double* a, const double* b,
                     const double* c, double* out);

/* ---- binnedCl / ClData band-power likelihood: MCMC/binnedCl, MCMC/ClData ----------------- */
/* One data set = the attributes of a ClData_CosmoMC object (ClData_CosmoMC.py:85-195 fills them
 * from a .dataset file; the Python mirror class hands them over as arrays).  All host pointers. */
typedef struct {
  int32_t num_points;          /* bands nb                                        :94            */
  int32_t win_len;             /* last dimension of window                        :198-258       */
  const double* obs;           /* [nb]  ln(C+x) where has_xfactor, else C         :377-398       */
  const double* N_inv;         /* [nb,nb] inverse covariance, or NULL: has_corr_errors False     */
  const double* var;           /* [nb] variances, used when N_inv is NULL         :412,420       */
  const double* xfactors;      /* [nb] or NULL (has_xfactors False)                              */
  const uint8_t* has_xfactor;  /* [nb] or NULL                                                   */
  const double* window;        /* [nb, 3, win_len]  TT, TE, EE windows            :422-446       */
  const int32_t* win_min;      /* [nb]                                                           */
  const int32_t* win_max;      /* [nb] inclusive; may exceed the spectrum length  :436           */
  double calib_uncertainty;    /*                                                 :282,299,337   */
  const double* beam_err;      /* [nb] or NULL (beam_uncertain False)             :293,334       */
  int32_t has_pol;             /*                                                 :439           */
  const uint8_t* inc_pol;      /* [nb] or NULL                                                   */
} mi355_cl_dataset;

/* like.data = [ClData, ...]: replaces any data sets loaded before. */
int mi355_cl_load(mi355_ctx* ctx, int32_t ndata, const mi355_cl_dataset* data);
/* binnedClModel.setBinning(bins, shapefun) (binnedClModel.py:145-252): nCl <= 3 spectra (TT, TE,
 * EE), nbins[iCl] bins each, bins host [sum nbins, 2] inclusive (lmin, lmax) in parameter order,
 * shapefun host [nCl, lmax+1] = D_l shape with lmax the largest binned multipole (the caller
 * truncates).  Bins of one spectrum must not overlap.  Builds the band-power
 * matrices K, BP0 of every loaded data set on the device. */
int mi355_cl_set_binning(mi355_ctx* ctx, int32_t nCl, const int32_t* nbins, const int32_t* bins,
                         const double* shapefun, int32_t lmax);
int32_t mi355_cl_nparam(mi355_ctx* ctx);
/* binnedClLikelihood.lnLike(q) (binnedClLikelihood.py:26-32 -> ClData_CosmoMC.calcLnLike :267-315)
 * summed over the loaded data sets, for n parameter vectors: q host [n, npar] -> lnL host [n]. */
int mi355_cl_lnlike(mi355_ctx* ctx, int64_t n, const double* q, double* lnL);
/* Same on device memory: q_dev parameter-major [npar, n], lnL_dev [n]; asynchronous. */
int mi355_cl_lnlike_dev_async(mi355_ctx* ctx, int64_t n, const double* q_dev, double* lnL_dev);
/* binnedClModel.prior(*q) (:125-132): host [n, npar] -> prior [n]. */
int mi355_cl_prior(mi355_ctx* ctx, int64_t n, const double* q, double* prior);
/* binnedClModel(q)() (:62-86): host q [n, npar] -> Cl [n, 3, lmax+1]. */
int mi355_cl_model(mi355_ctx* ctx, int64_t n, const double* q, double* Cl);
/* Band powers of data set idata: from bin amplitudes (what==0, in host [n, npar], through K) or
 * from explicit spectra as getWinBandpower does (what==1, in host [n, rows, L], rows 1 or 3;
 * :422-446) -> bp host [n, nb]. */
int mi355_cl_bandpowers(mi355_ctx* ctx, int32_t idata, int32_t what, int64_t n, const double* in,
                        int32_t rows, int32_t L, double* bp);
/* Methods of ClData_CosmoMC on explicit vectors, data set idata, host [n, nb] in:
 *  op 0: getdelta(BP)   -> out [n, nb]   :377-398
 *  op 1: getlinform(v)  -> out [n, nb]   :407-412
 *  op 2: getquadform(v) -> out [n]       :415-420
 *  op 3: getchisq(BP)   -> out [n]       :400-405
 *  op 4: calcLnLike from band powers (marginalisations included) -> out [n] = -ln L   :267-315 */
int mi355_cl_data_op(mi355_ctx* ctx, int32_t idata, int32_t op, int64_t n, const double* in,
                     double* out);
/* ClData_CosmoMC.calcLnLike(Cl) (:267-315) for explicit spectra: Cl host [n, rows, L] -> out [n]
 * (= MINUS ln L, as the reference returns it). */
int mi355_cl_calc_lnlike(mi355_ctx* ctx, int32_t idata, int64_t n, const double* Cl, int32_t rows,
                         int32_t L, double* out);

/* ---- BeamFit: MCMC/BeamFit ---------------------------------------------------------------- */
/* Class-level prior state of GaussianBeamModel2D (BeamModel.py:41-43) as
 * ProcessBeamData.setup_sampler sets it (:149-151); a has_* flag of 0 means "None". */
typedef struct {
  double center_min[2], center_max[2];
  double sig_min, sig_max;
  int32_t has_center, has_sig_min, has_sig_max;
} mi355_beam_prior;

/* M maps of npix pixels each (BeamData, BeamData.py:11-29 / GaussianData.py:13-28).
 *  x, y : host [npix] shared by all maps (xy_per_map=0) or [M, npix] (=1)
 *  d    : host [M, npix]
 *  sig  : host [1] (sig_kind 0), [M] (1) or [M, npix] (2)
 * Flat parameter vector (OffsetNormalizedBeamModel.unpackage, :123-127):
 *  (x, y, sigma_1, sigma_2, angle, amplitude, offset, mu, phi). */
int mi355_beam_load(mi355_ctx* ctx, int64_t M, int64_t npix, const double* x, const double* y,
                    int32_t xy_per_map, const double* d, const double* sig, int32_t sig_kind,
                    const mi355_beam_prior* prior);
/* NormalizedBeamLikelihood.lnLike(package(theta)) (NormalizedBeamLikelihood.py:9-16 ->
 * OffsetNormalizedBeamModel.at :78-82 -> gauss2d BeamModel.py:288-290 -> GaussianData.chi2) for
 * maps [m0, m0+nm) x C parameter vectors each: theta host [nm, C, 9] -> lnL host [nm, C]. */
int mi355_beam_lnlike(mi355_ctx* ctx, int64_t m0, int64_t nm, int32_t C, const double* theta,
                      double* lnL);
/* Same on device memory: theta_dev parameter-major [9, nm*C], lnL_dev [nm*C]; asynchronous. */
int mi355_beam_lnlike_dev_async(mi355_ctx* ctx, int64_t m0, int64_t nm, int32_t C,
                                const double* theta_dev, double* lnL_dev);
/* model.prior(*package(theta)) (OffsetNormalizedBeamModel.py:93-101, BeamModel.py:165-187) and
 * unpackage(package(theta)) (:123-133) for n flat vectors: host [n,9] -> prior [n], packaged [n,9]
 * (either nullable). */
int mi355_beam_prior_package(mi355_ctx* ctx, int64_t n, const double* theta, double* prior,
                             double* packaged);
/* model(*package(theta)).atxy(x, y) (:84-88): theta host [n,9], x, y host [npts] -> out [n, npts].
 * Needs no maps. */
int mi355_beam_model(mi355_ctx* ctx, int64_t n, const double* theta, int64_t npts, const double* x,
                     const double* y, double* out);
/* gauss2d(x, y, x0, y0, Cinv_xx, Cinv_xy, Cinv_yy) = exp(-(dx^2 Cxx + dy^2 Cyy + 2 dx dy Cxy)/2)
 * (BeamModel.py:288-290; what GaussianBeamModel2D.at / atxy return, :152-160) for n points with the
 * packed inverse covariance given (set_Cinv, :128-136, is scalar work of the caller).  Host
 * pointers x, y [n] -> out [n].  Needs no loaded maps. */
int mi355_beam_gauss2d(mi355_ctx* ctx, int64_t n, const double* x, const double* y, double x0,
                       double y0, double cinv_xx, double cinv_xy, double cinv_yy, double* out);
/* BeamData.stats(sigcut, do_abs) of map m (BeamData.py:35-71): out host [4] = <x>, <y>, var x,
 * var y weighted by the mean-subtracted data (has_sigcut=0 means sigcut=None). */
int mi355_beam_stats(mi355_ctx* ctx, int64_t m, int32_t has_sigcut, double sigcut, int32_t do_abs,
                     double* out);

/* ---- step-wise Metropolis-Hastings over any problem kind ---------------------------------- */
/* mi355_mh_init (above) starts the fused SED sampler: one kernel launch advances a chain many
 * steps, uncorrelated proposals.  mi355_mh_init_problem starts the step-wise sampler instead: one
 * MH step = propose kernel -> the problem's batched likelihood -> accept kernel.  It serves the
 * problems whose likelihood is itself a kernel pipeline (binnedCl, BeamFit) and every proposal of
 * Proposal.py, including the Cholesky-correlated (:150-154) and rotated (:145-148) ones, for all
 * three problem kinds.  After it, mi355_mh_run / _moments / _sigmas / _trace / _adapt / mi355_rhat*
 * act on the step-wise chains (same buffers, same layouts).
 *  problem : MI355_PROBLEM_SED (catalogue of mi355_sed_load: S sources x C chains),
 *            MI355_PROBLEM_CL  (mi355_cl_load + mi355_cl_set_binning: 1 x C chains),
 *            MI355_PROBLEM_BEAM (mi355_beam_load: M maps x C chains)
 *  track_cov : 1 = keep running co-moments so that covar() / correlated restarts need no trace
 *  other arguments as mi355_mh_init; the draw-injection contract of mi355_mh_run is the same, and
 *  in rotated mode the k-th member of the block takes element k of the step's row of normals. */
enum mi355_problem_kind { MI355_PROBLEM_SED = 0, MI355_PROBLEM_CL = 1, MI355_PROBLEM_BEAM = 2 };
int mi355_mh_init_problem(mi355_ctx* ctx, int32_t problem, int32_t C, const double* theta0,
                          int32_t theta0_per_chain, const double* sigma, int32_t sigma_per_chain,
                          const int32_t* param_blocks, int32_t nblock, int32_t randomize_start,
                          int32_t track_cov);
/* One-launch runs for problems started with mi355_mh_init_problem.  A binnedCl problem whose data
 * sets need at most 11 marginalisation points per chain (no marginalisation, analytic, or the
 * 11-point calibration or beam grid) runs mi355_mh_run as ONE kernel launch: every MH step of
 * MCMC.py:225-286 - proposal of any mode, band powers, MFMA quadratic form, marginalisation,
 * accept, moments - inside it, chain state on chip.  A SED problem on a catalogue that passed the
 * fast-path check of mi355_sed_load runs in the fused SED kernel of mi355_mh_init, extended by the
 * correlated and rotated proposals (Proposal.py:145-154) and the co-moments.  Beam chains of at most 8 chains
 * per map run in a fused kernel of their own (a block per map) where that is the faster path: maps of up to
 * 1024 pixels, jobs of some hundred maps.  Same draws, same
 * arithmetic, same chains as the step-wise launches, bit for bit.  on=0 forces the step-wise
 * launches; on=1 (default) takes the one-launch path where it is the faster one (the full 11 x 11 grid
 * runs fused between 2049 and 8192 chains and step-wise outside); on=2 takes it wherever a fused kernel
 * serves the problem.
 * mi355_mh_fused_active: 1 if the next mi355_mh_run of this context takes the one-launch path. */
int mi355_mh_set_fused(mi355_ctx* ctx, int32_t on);
int mi355_mh_fused_active(mi355_ctx* ctx);
/* prop.setNormalizedMatrixFromSqrt(L) + prop.rotateParams = rotate (Proposal.py:106-109, :78):
 * L host [P,P] (per_chain=0) or [S,C,P,P] (=1); NULL returns to uncorrelated proposals. */
int mi355_mh_set_sqrt_matrix(mi355_ctx* ctx, const double* L, int32_t per_chain, int32_t rotate);
/* newMCMC(burn=1, fac, random, noCorrelations, rotateParams, doBlock) for every chain
 * (MCMC.py:343-413): restart at package(mean), sigma = package(fac*stdev), and unless
 * no_correlations the Cholesky factor of the normalised covariance over the parameters with
 * sigma>0 (Proposal.py:81-101; +0.01 on the diagonal if it is not positive definite,
 * MCMC.py:397-399).  do_block: 1 / 0 = doBlock True / False, -1 = keep (MCMC.py:382-383). */
int mi355_mh_adapt_corr(mi355_ctx* ctx, double fac, int32_t random_restart, int32_t no_correlations,
                        int32_t rotate, int32_t do_block);
/* covar(burn=1) of every chain (MCMC.py:310-340) from the running co-moments: host [S,C,P,P];
 * normalized=1 divides by stdev_i*stdev_j. */
int mi355_mh_covar(mi355_ctx* ctx, double* cov, int32_t normalized);
/* Pause / resume the running co-moments of a sampler started with track_cov=1: a production stage
 * that will not be adapted from need not pay for them (P(P+1)/2 read-modify-writes per chain and
 * step).  mi355_mh_covar and correlated restarts need them on for every step of the stage. */
int mi355_mh_track_cov(mi355_ctx* ctx, int32_t on);
/* prop.sqrtMatrix of every chain: host [S,C,P,P]. */
int mi355_mh_get_sqrt_matrix(mi355_ctx* ctx, double* L);
/* 0 uncorrelated, 1 Cholesky-correlated, 2 rotated */
int32_t mi355_mh_proposal_mode(mi355_ctx* ctx);

/* ---- Gelman-Rubin: MCMC/convergence.py:58-87 -------------------------------------------- */
/* Per-source partial sums over this context's C chains, for an all-reduce across GPUs that
 * hold different chains of the same sources (chain-sharded layout).  dev buffers:
 *  pass 0: sums_dev [P, S]    = sum_c mean_c
 *  pass 1: given gmean_dev [P, S] (global mean of chain means), sums_dev [2, P, S] =
 *          (sum_c (mean_c-gmean)^2, sum_c var_c)                                            */
int mi355_rhat_partial_async(mi355_ctx* ctx, int32_t pass, const double* gmean_dev,
                             double* sums_dev);
/* Rhat = sqrt((W+B)/W), W = sum var / Ctot, B = sum (mean-gmean)^2 / (Ctot-1); NaN where W<=0
 * (the reference drops those parameters, convergence.py:81-83).  rhat_dev [P, S]. */
int mi355_rhat_finish_async(mi355_ctx* ctx, const double* sums_dev, int64_t Ctot, double* rhat_dev);
/* gelmanrubin_MCMC(MCMCs) (convergence.py:58-87) from per-chain moments handed over by the host:
 * means, vars host [nchain, P] (vars = stdev**2, ddof=0) -> rhat host [P], NaN where the mean
 * within-chain variance is not positive.  Needs no catalogue. */
int mi355_gelman_rubin(mi355_ctx* ctx, int32_t nchain, int32_t P, const double* means,
                       const double* vars, double* rhat);
/* Effective sample size per (group, parameter) from the trace kept by the last mi355_mh_run
 * (4 <= rows <= 128, C <= 256): the multi-chain variogram estimator of Gelman et al., BDA3
 * sect. 11.5, over the C chains of each source / map.  No counterpart in the reference (it has no
 * ESS or autocorrelation code); BASELINE.json's metric asks for ESS/s.  ess host [S, P], NaN
 * where the parameter did not move. */
int mi355_mh_ess(mi355_ctx* ctx, double* ess);
/* Single-GPU convenience: all C chains are local.  rhat host [S, P]. */
int mi355_rhat(mi355_ctx* ctx, double* rhat);
/* Multi-GPU mi355_rhat behind the header alone: one process per GPU, every rank holds C chains of the
 * same S sources (chain-sharded layout) and calls this with its communicator; Ctot = chains per source
 * over all ranks.  Replaces gelmanrubin_MCMC over the chains of all ranks (convergence.py:58-87,
 * driver.py:247-249).  rccl_comm is the caller's ncclComm_t (RCCL; passed as void* so that this header
 * needs no rccl.h).  Two all-reduces (SUM, FP64) of S*P and 2*S*P doubles on the context stream between
 * the partial-sum kernels; asynchronous; rhat_dev [P, S] on the device (the layout of
 * mi355_rhat_finish_async).  The library does not link RCCL: it binds ncclAllReduce from the RCCL the
 * process has loaded (the caller's own, or torch's), else from librccl.so on the loader path, and fails
 * with an error if there is none.  mcmc_amd.distributed.engine_rhat does the same through
 * torch.distributed.  rccl_comm may be NULL when this rank holds every chain (Ctot = its C): the same
 * four kernels without a collective - mi355_rhat with the result left on the device. */
int mi355_rhat_allreduce_async(mi355_ctx* ctx, void* rccl_comm, int64_t Ctot, double* rhat_dev);
/* mi355_rhat for callers that cannot allocate.  It has no counterpart in the reference; it is for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same
 * kernels as mi355_rhat, but the intermediates (chain-mean sums, centred sums) live in a fixed
 * device array of the library (24 MB per GPU, held until the process ends) and the result stays
 * on the device: rhat_dev [P, S] (the layout of mi355_rhat_finish_async).  The call makes no
 * hipMalloc / hipFree and no host allocation, and is asynchronous on the context stream.
 * Limit: S * P <= MI355_RHAT_NOALLOC_MAX, otherwise an error (mi355_rhat has no such limit).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two contexts, two
 * streams) race - serialise them. */
#define MI355_RHAT_NOALLOC_MAX 1048576
int mi355_rhat_noalloc_async(mi355_ctx* ctx, double* rhat_dev);

#ifdef __cplusplus
}
#endif
#endif /* MI355MCMC_H */
