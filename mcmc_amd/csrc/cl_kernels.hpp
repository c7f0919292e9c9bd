// HIP kernels of the binned-C_l band-power likelihood.
//
// Reference arithmetic: MCMC/binnedCl/binnedClModel.py:62-86 (q_b -> C_l), :125-132 (prior);
// MCMC/ClData/ClData_CosmoMC.py:267-315 (calcLnLike), :317-372 (11x11 numeric calibration/beam
// marginalisation), :377-398 (offset-lognormal delta), :407-420 (linear / quadratic forms),
// :422-446 (window band powers).
//
// Device formulation.  The band powers are linear in the bin amplitudes, BP = q.K + BP0 with K
// built once per (binning, dataset) by k_cl_build_K, so a likelihood evaluation is
//   k_cl_bandpower : BP[band][chain]                                  (small GEMM, VALU)
//   k_cl_form      : rows = (chain x marginalisation point); delta rows generated into LDS,
//                    Y = delta . N^-1 on v_mfma_f64_16x16x4_f64, chi2 = rowsum(Y * delta)
//   k_cl_combine   : log-sum-exp over the 11 (x11) grid / analytic marginalisation, -chi2/2.
// (The header is compiled into two translation units, cl_api.hip and cl_mh_api.hip: its non-template
// kernels are `static`, the one that does not launch them drops them.)
#pragma once
#include "common.hpp"
#include "fastmath.hpp"

namespace mi355 {

// ln for the offset-lognormal deltas in the MFMA kernel's row staging: the library's short
// branch-free kernel (a few ulp; |delta| errors of ~1e-15 against ln(C+x) ~ 8) with the IEEE end
// cases the reference relies on (ln 0 = -inf, :389).
template <bool SK = false>
__device__ __forceinline__ double cl_log(double z) {
  const double l = log_fast<SK>(z);
  return (z > 0.0) ? ((z < __builtin_inf()) ? l : z) : ((z == 0.0) ? -__builtin_inf() : __builtin_nan(""));
}

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kClHalfSteps = 5;          // ClData_CosmoMC.py:40
constexpr int kClGrid = 2 * kClHalfSteps + 1;

enum ClFormMode : int { CL_ROWS_MARGE = 0, CL_ROWS_TRIPLE = 1, CL_ROWS_RAW = 2 };

struct ClFormArgs {
  const double* bp;        // [nbp][n]  band powers, band-major (RAW: the row vectors themselves)
  int64_t n;               // chains
  int nb, nbp;             // bands, padded to a multiple of 16
  const double* obs;       // [nbp]
  const double* xf;        // [nbp] x-factors (0 where none)
  const uint8_t* hasx;     // [nbp]
  const double* beam_err;  // [nbp] (zeros when !beam_uncertain)
  const double* bfac;      // [11][nbp] beam factors 1 + beam_err*ib*3/5, ib = -5..5 (k_cl_beam_factors)
  const double* ninv;      // [nbp + 16][nstride] zero padded (full covariance) ...
  const double* nsym;      // [nbp][nbp] block-upper symmetrised copy: tile (I,J), I<J, holds
                           // N_IJ + N_JI^T, diagonal tiles N_II, tiles I>J unused (k_cl_form SYM)
  int nstride;             // row stride of ninv / nsym in doubles (cl_gstride in HBM, cl_mh_nstride in an LDS copy)
  const double* var;       // ... or [nbp] variances (uncorrelated errors)
  int has_x;               // dataset has_xfactors
  int beam_uncertain;
  double calib;            // calib_uncertainty
  int mode;                // ClFormMode
  int ncal, nbeam;         // marginalisation grid (1 or 11 each); rows per chain = ncal*nbeam
  int64_t rows;            // total rows = n * rows_per_chain (3 per chain in TRIPLE mode)
  double* out;             // MARGE: chisq[rows]; TRIPLE: G[n][9]
};

// Row stride (doubles) of the matrices in HBM (K, N^-1 and its symmetrised copy): nbp + 16, so that
// the rows of a column tile do not all start on the same L2 channel when nbp is a multiple of 256
// doubles (2 KB = 16 cache lines).  Measured: no difference for the kernels here (the L2 hashes its
// channels); kept because it costs nothing.
__host__ __device__ inline int cl_gstride(int nbp) { return nbp + 16; }

// (chain, sub-index) of global row `grow`: sub = marginalisation point (MARGE), row type
// (TRIPLE) or 0 (RAW, where "chain" is the row itself); chain = -1 for rows past the end.
__device__ __forceinline__ void cl_row_decode(const ClFormArgs& a, int64_t grow, int& c, int& sub) {
  if (grow >= a.rows) { c = -1; sub = 0; return; }
  const int per = (a.mode == CL_ROWS_TRIPLE) ? 3 : (a.mode == CL_ROWS_RAW ? 1 : a.ncal * a.nbeam);
  const int r32 = static_cast<int>(grow);                  // rows < 2^31 (run_form checks)
  if (per == 1) { c = r32; sub = 0; return; }              // no 64-bit (or any) division on the hot path
  c = r32 / per;
  sub = r32 - c * per;
}

// Value of element `k` (band) of the row (c, sub): the vector the quadratic form is taken of.
// TAS / IOAS: address space of the data set's tables and matrices / of a.bp and a.out (ldas in
// common.hpp; 0 where the compiler sees the kernel argument itself).
template <int TAS = 0, int IOAS = 0>
__device__ __forceinline__ double cl_row_value(const ClFormArgs& a, int c, int sub, int k) {
  if (c < 0 || k >= a.nb) return 0.0;
  const double bp = ldas<IOAS>(a.bp + static_cast<int64_t>(k) * a.n + c);
  if (a.mode == CL_ROWS_RAW) return bp;                    // getquadform(vec)
  if (a.mode == CL_ROWS_TRIPLE) {
    if (sub == 0) return ldas<TAS>(a.obs + k) - bp;        // getdelta without x-factors, :396
    if (sub == 1) return bp;                               // :289
    return a.beam_uncertain ? ldas<TAS>(a.beam_err + k) * bp : 0.0;    // :293
  }
  double v = bp;
  if (a.nbeam > 1) {                                       // :333-334
    const int ib = sub / a.ncal - kClHalfSteps;
    v *= 1.0 + ldas<TAS>(a.beam_err + k) * ib * 3 / kClHalfSteps;
  }
  if (a.ncal > 1) {                                        // :337-338
    const int ic = sub % a.ncal - kClHalfSteps;
    const double calib = 1.0 + a.calib * ic * 3 / kClHalfSteps;
    v = calib * v;
  }
  if (a.has_x && ldas<TAS>(a.hasx + k)) {                  // :383-392
    double zth = v + ldas<TAS>(a.xf + k);
    if (zth <= 0.0) zth = 0.0;                             // forces -inf rather than NaN, :389
    return ldas<TAS>(a.obs + k) - log(zth);
  }
  return ldas<TAS>(a.obs + k) - v;
}

// Full-covariance forms on the FP64 matrix cores.  NW waves per block (4 or 8); the block owns
// TR = 16*RT rows (all waves share them) and every wave owns the column tiles ct = w, w+4, ...
// of Y = V.N^-1 -> ct = w, w+NW, ..., CG of them per pass.  Eight waves put two on every SIMD, so
// the row staging (a long dependent chain per element: ln) and the MFMA stream of one wave fill
// the stalls of the other.  V is staged band-major in LDS (vt[k][TR]); N^-1
// fragments come straight from L2 (the matrix is shared by every block) and are fetched four
// k-steps ahead of the MFMAs that use them.  MFMA operand maps (cdna_hip_programming.md
// sect. 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D col = l&15, row = (l>>4) + 4*reg.
// MODE (ClFormMode) is a template parameter so that the row-staging loop is specialised; the
// TRIPLE mode keeps three dot products per Y row (NT = 3), the others one.  waves-per-EU >= 2 in
// the launch bounds caps the kernel at 256 registers, which makes the compiler keep the MFMA
// accumulators in VGPRs (no per-iteration VGPR<->AGPR copies).
// SYM (diagonal forms v N^-1 v only, CG = 1): the quadratic form is accumulated over the block
// upper triangle of the symmetrised matrix, sum_J (sum_{I<=J} v_I S_IJ) v_J, which halves the
// MFMA work; column tiles are dealt to the waves in snake order so that every wave gets the same
// number of k-steps.
#ifndef MI355_CL_ABL
#define MI355_CL_ABL 0     // timing ablations (never set in the product build)
#endif
// The whole tile computation is a device function so that the fused Metropolis-Hastings kernel
// (cl_mh_kernels.hpp) runs the very same code on row tiles it generates itself: `a.bp` / `a.out`
// may then point into LDS (a.n = chains of the block), `row0` is the tile's first row and `smem`
// the tile's LDS region (form_lds_bytes in cl_api.hip).  All 64*NW threads of the block must call
// it; it ends without a barrier (the caller synchronises before it touches `smem` again).
// PF: how many k-steps ahead the matrix fragments are fetched (4, or 8 where few waves share a
// SIMD and the matrix comes from L2: the fused kernel's streamed data sets); the matrix carries
// kClFormSlackRows zero rows for the fetches past its end.
constexpr int kClFormSlackRows = 32;
#ifdef MI355_CLMH_PROF
// phase timing of cl_form_tile (block 0, PROF builds only): staging, MFMA loop + epilogue, reduction
static __device__ long long g_clf[4];
static __device__ long long g_clf_t;
#define CLF_TICK(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) { const long long n_ = clock64(); if ((k) >= 0) g_clf[(k) & 3] += n_ - g_clf_t; g_clf_t = n_; } } while (0)
#else
#define CLF_TICK(k) do {} while (0)
#endif
template <int RT, int CG, int MODE, int NW, bool SYM, int PF = 4, int TAS = 0, int IOAS = 0, int MAS = TAS>
__device__ __forceinline__ void cl_form_tile(const ClFormArgs& a, const int64_t row0, double* smem) {
  static_assert(PF == 4 || PF == 8, "fetch distance");
  static_assert(!SYM || (CG == 1 && MODE != CL_ROWS_TRIPLE), "SYM needs CG == 1 and a diagonal form");
  constexpr int TR = 16 * RT;
  constexpr int NTHR = 64 * NW;
  constexpr int NT = (MODE == CL_ROWS_TRIPLE) ? 3 : 1;
  double* vt = smem;                                       // [nbp][TR]
  double* red = smem + static_cast<size_t>(a.nbp) * TR;    // [NW][TR][3]
  int* rowc = reinterpret_cast<int*>(red + (NW * 3 > 16 ? NW * 3 : 16) * TR);   // [TR] chain of each row
  int* rows = rowc + TR;                                   // [TR] sub-index of each row
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the column-tile loops are scalar
  constexpr int rows_used = (MODE == CL_ROWS_TRIPLE) ? (TR / 3) * 3 : TR;

  CLF_TICK(-1);
  if (tid < TR) {
    int c = -1, sub = 0;
    if (tid < rows_used) cl_row_decode(a, row0 + tid, c, sub);
    rowc[tid] = c;
    rows[tid] = sub;
  }
  __syncthreads();
  {
    // every thread owns one row r for the whole staging loop: the per-row factors are hoisted,
    // the per-(beam point, band) factors come from the table built at load time
    const int r = tid % TR;                                // TR is a power of two
    const int c = rowc[r], sub = rows[r];
    constexpr bool marge = MODE == CL_ROWS_MARGE;
    // no marginalisation over one of the two: the centre point of its grid, whose factor is
    // exactly 1 (the beam-factor table always exists; multiplying by 1.0 changes nothing)
    double calib = 1.0;
    const auto bfac_t = as_ptr<TAS>(a.bfac);
    const auto xf_t = as_ptr<TAS>(a.xf);
    const auto obs_t = as_ptr<TAS>(a.obs);
    const auto hasx_t = as_ptr<TAS>(a.hasx);
    const auto bp_t = as_ptr<IOAS>(a.bp);
    auto bf = bfac_t + static_cast<size_t>(kClHalfSteps) * a.nbp;
    if (marge && c >= 0) {
      if (a.ncal > 1) calib = 1.0 + a.calib * (sub % a.ncal - kClHalfSteps) * 3 / kClHalfSteps;   // :337
      if (a.nbeam > 1) bf = bfac_t + static_cast<size_t>(sub / a.ncal) * a.nbp;                   // :334
    }
    const bool has_x = a.has_x != 0;
    constexpr int KS = NTHR / TR;                          // bands between a thread's iterations
    if constexpr (!marge) {
#pragma unroll 4
      for (int k = tid / TR; k < a.nbp; k += KS) vt[k * TR + r] = cl_row_value<TAS, IOAS>(a, c, sub, k);
    } else {
      // Four bands per trip: all operand loads first (the tables are padded to nbp, indices
      // clamped), then the arithmetic, every lane computing the logarithm whether it needs it
      // or not - a branch around it would put a memory round trip in front of every band.
      const int64_t cc = c >= 0 ? c : 0;
      for (int k0 = tid / TR; k0 < a.nbp; k0 += 4 * KS) {
        double bpv[4], bfv[4], xfv[4], obv[4];
        uint8_t hx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = min(k0 + u * KS, a.nbp - 1);
          bpv[u] = bp_t[static_cast<int64_t>(k) * a.n + cc];
          bfv[u] = bf[k];
          xfv[u] = xf_t[k];
          obv[u] = obs_t[k];
          hx[u] = hasx_t[k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * KS;
          double b = bpv[u] * bfv[u];                      // :334
          b = calib * b;                                   // :338
          double zth = b + xfv[u];                         // :383-392
          zth = (zth <= 0.0) ? 0.0 : zth;
#if MI355_CL_ABL == 1
          double lg = zth;
#else
          double lg = cl_log(zth);
#endif
          asm volatile("" : "+v"(lg));                     // keep the logarithm out of a branch
          const double t = (has_x && hx[u]) ? lg : b;
          const double v = (c >= 0 && k < a.nb) ? obv[u] - t : 0.0;
          if (k < a.nbp) vt[k * TR + r] = v;
        }
      }
    }
  }
  __syncthreads();
  CLF_TICK(0);

  const int nct = a.nbp >> 4;
  const int lk = lane >> 4, lc = lane & 15;
  double part[RT][4][NT];                                  // per-lane partial row sums
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) part[rt][i][t] = 0.0;

  const size_t kstride = static_cast<size_t>(4) * a.nstride;
  const auto nmat = as_ptr<MAS>(SYM ? a.nsym : a.ninv);
  for (int pass = 0;; ++pass) {
    int ct0;
    if (SYM) {
      const int base = pass * NW;
      if (base >= nct) break;
      ct0 = (pass & 1) ? base + NW - 1 - w : base + w;     // snake: long and short tiles pair up
      if (ct0 >= nct) continue;
    } else {
      ct0 = w + pass * NW * CG;
      if (ct0 >= nct) break;
    }
    double4_t acc[RT][CG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < CG; ++j) acc[rt][j] = double4_t{0.0, 0.0, 0.0, 0.0};
    // a column tile past the end is computed as a duplicate of the first one and dropped in the
    // epilogue: no branch inside the MFMA loop
    as_ptr_t<MAS, const double> nb_ptr[CG];
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const int ct = (ct0 + NW * j < nct) ? ct0 + NW * j : ct0;
      nb_ptr[j] = nmat + static_cast<size_t>(lk) * a.nstride + ct * 16 + lc;
    }
    const double* va_ptr = vt + lk * TR + lc;
    // Software pipeline: the matrix fragments (L2, ~600 cycles away) are fetched FOUR k-steps
    // ahead into a ring of registers, the row fragments (LDS) one k-step ahead; the scheduling
    // barriers keep the compiler from sinking the fetches next to their use.  The fetches run
    // up to four k-steps past the end: the matrix is allocated with 16 zero rows of slack and
    // the row tile is followed by the reduction scratch in LDS, and nothing read there is used.
    double av[2][RT], bv[PF][CG];
    as_ptr_t<MAS, const double> bp[CG];
#pragma unroll
    for (int j = 0; j < CG; ++j) {
#pragma unroll
      for (int u = 0; u < PF; ++u) bv[u][j] = nb_ptr[j][u * kstride];
      bp[j] = nb_ptr[j] + PF * kstride;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) av[0][rt] = va_ptr[rt * 16];
    const double* ap = va_ptr + 4 * TR;
#if MI355_CL_ABL == 2
    const int ksteps = 4;
#else
    const int ksteps = SYM ? (ct0 + 1) * 4 : a.nbp >> 2;   // a multiple of 4
#endif
    // one k-step: row fragment of the next step, MFMA of this one, refill of the ring slot
    auto kstep = [&](int u, bool refill) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av[(u + 1) & 1][rt] = ap[rt * 16];
      ap += 4 * TR;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < CG; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u & 1][rt], bv[u][j], acc[rt][j], 0, 0, 0);
      if (refill) {
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          bv[u][j] = bp[j][0];
          bp[j] += kstride;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    int ks = 0;
    for (; ks + PF <= ksteps; ks += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) kstep(u, true);
    }
    if constexpr (PF > 4) {
      // ksteps is a multiple of 4: a last half turn of the ring, whose slots 0..3 hold exactly these
      // k-steps (fetched by the last full turn, or by the prologue)
      if (ks < ksteps) {
#pragma unroll
        for (int u = 0; u < 4; ++u) kstep(u, false);
      }
    }
    // epilogue: Y[row][col] * V[row'][col] summed over this wave's columns
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const int col = (ct0 + NW * j) * 16 + lc;
      if (ct0 + NW * j >= nct || col >= a.nb) continue;     // duplicate tile / padded column
      const double* vcol = vt + col * TR;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = rt * 16 + lk + 4 * i;
          const double y = acc[rt][j][i];
          if constexpr (NT == 3) {
            const int g3 = (row / 3) * 3;
            if (row < rows_used) {
#pragma unroll
              for (int t = 0; t < 3; ++t) part[rt][i][t] = fma(y, vcol[g3 + t], part[rt][i][t]);
            }
          } else {
            part[rt][i][0] = fma(y, vcol[row], part[rt][i][0]);
          }
        }
    }
  }
  CLF_TICK(1);
  // reduce over the 16 lanes that share (lane>>4), i.e. over columns
  constexpr int nt = NT;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double v = row16_xor_sum(part[rt][i][t]);    // the xor 1, 2, 4, 8 butterfly, on the vector ALU
        if (lc == 0) red[(w * TR + rt * 16 + lk + 4 * i) * 3 + t] = v;
      }
  __syncthreads();
  const auto out_t = as_ptr<IOAS>(a.out);
  for (int idx = tid; idx < rows_used * nt; idx += NTHR) {
    const int r = idx / nt, t = idx - r * nt;
    const int c = rowc[r];
    if (c < 0) continue;
    double s = red[r * 3 + t];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) s += red[(ww * TR + r) * 3 + t];
    if constexpr (MODE == CL_ROWS_TRIPLE) out_t[static_cast<int64_t>(c) * 9 + rows[r] * 3 + t] = s;   // G[c][type][t]
    else out_t[row0 + r] = s;
  }
  CLF_TICK(2);
}

template <int RT, int CG, int MODE, int NW, bool SYM>
__global__ __launch_bounds__(64 * NW, 2) void k_cl_form(ClFormArgs a) {
  extern __shared__ double smem[];
  constexpr int rows_used = (MODE == CL_ROWS_TRIPLE) ? ((16 * RT) / 3) * 3 : 16 * RT;
  cl_form_tile<RT, CG, MODE, NW, SYM>(a, static_cast<int64_t>(blockIdx.x) * rows_used, smem);
}

// Uncorrelated errors (has_corr_errors False): sum(vec**2/var), vec/var (:412, :420).
// One thread per row (MARGE) or per chain (TRIPLE), bands in ascending order like the
// reference's builtin sum().
template <int TAS = 0, int IOAS = 0>
__device__ __forceinline__ void cl_form_diag_item(const ClFormArgs& a, const int64_t i) {
  if (a.mode == CL_ROWS_TRIPLE) {
    if (i >= a.n) return;
    double g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < a.nb; ++k) {
      double v[3];
      for (int t = 0; t < 3; ++t) v[t] = cl_row_value<TAS, IOAS>(a, static_cast<int>(i), t, k);
      const double var = ldas<TAS>(a.var + k);
      g[0] += v[0] * v[0] / var;                         // getquadform, :420
      for (int t = 1; t < 3; ++t) {
        const double lin = v[t] / var;                   // getlinform, :412
        for (int u = 0; u < 3; ++u) g[t * 3 + u] += v[u] * lin;
      }
    }
    for (int t = 0; t < 9; ++t) stas<IOAS>(a.out + i * 9 + t, g[t]);
  } else {
    if (i >= a.rows) return;
    double s = 0.0;
    int c, sub;
    cl_row_decode(a, i, c, sub);
    for (int k = 0; k < a.nb; ++k) {
      const double v = cl_row_value<TAS, IOAS>(a, c, sub, k);
      s += v * v / ldas<TAS>(a.var + k);
    }
    stas<IOAS>(a.out + i, s);
  }
}

static __global__ __launch_bounds__(256) void k_cl_form_diag(ClFormArgs a) {
  cl_form_diag_item(a, static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x);
}

// 1 + beam_err*ibeam*3/halfsteps for ibeam = -5..5 (ClData_CosmoMC.py:334), once per data set
static __global__ void k_cl_beam_factors(int nbp, const double* __restrict__ beam_err, double* __restrict__ bfac) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kClGrid * nbp) return;
  const int ib = e / nbp - kClHalfSteps, k = e % nbp;
  bfac[e] = 1.0 + beam_err[k] * ib * 3 / kClHalfSteps;
}

// ---- marginalisation and -ln L ------------------------------------------------------------
struct ClCombineArgs {
  const double* form;      // chisq[n*npc] or G[n][9]
  int64_t n;
  int mode, ncal, nbeam;
  int beam_uncertain;
  double calib;
  const double* weights;   // [11] margeweights, :42-47
  double margenorm;
  double* lnl;             // [n]
  int accumulate;          // 0: lnl = -neglnl, 1: lnl += -neglnl (sum over datasets)
  double* neglnl;          // nullable [n]: calcLnLike's own return value
};

// -2 ln( sum_i w_i exp(-(c_i - min)/2) / norm ) + min over an 11-point grid, :340-354
// CAS: address space of the chi-squares c (0: generic, e.g. a local array)
template <int CAS = 0>
__device__ __forceinline__ double cl_marge11(const double* c, int stride, const double* w,
                                             double norm) {
  double cv[kClGrid], wv[kClGrid];
  for (int i = 0; i < kClGrid; ++i) {
    cv[i] = ldas<CAS>(c + i * stride);
    wv[i] = ldas<1>(w + i);                                // margeweights: always in HBM
  }
  double mn = cv[0];
  for (int i = 1; i < kClGrid; ++i) {
    const double v = cv[i];
    if (v < mn) mn = v;                                  // Python min(): first smallest, NaN-blind
  }
  double s = 0.0;
  for (int i = 0; i < kClGrid; ++i) s += wv[i] * exp(-(cv[i] - mn) / 2);
  return -2 * log(s / norm) + mn;
}

// IOAS: address space of a.form and a.lnl (a.neglnl is only given by the batched path: generic)
template <int IOAS = 0>
__device__ __forceinline__ void cl_combine_chain(const ClCombineArgs& a, const int64_t c) {
  if (c >= a.n) return;
  double chisq;
  if (a.mode == CL_ROWS_MARGE) {
    const int npc = a.ncal * a.nbeam;
    const double* f = a.form + c * npc;
    if (npc == 1) {
      chisq = ldas<IOAS>(f);
    } else {
      double cc[kClGrid];
      for (int ib = 0; ib < a.nbeam; ++ib) {
        if (a.ncal > 1) {
          cc[ib] = cl_marge11<IOAS>(f + ib * a.ncal, 1, a.weights, a.margenorm);
        } else {
          // calib_uncertainty == 0: the 11 calibration points are the same number and the
          // weighted mean of exp(0) is exactly 1 (same summation order as margenorm)
          cc[ib] = ldas<IOAS>(f + ib);
        }
      }
      chisq = (a.nbeam > 1) ? cl_marge11<0>(cc, 1, a.weights, a.margenorm) : cc[0];
    }
  } else {                                               // analytic, :286-313
    const double* g = a.form + c * 9;
    chisq = ldas<IOAS>(g);
    const double chi2op = ldas<IOAS>(g + 3), chi2pp = ldas<IOAS>(g + 4);
    const double chi2od = ldas<IOAS>(g + 6), chi2pd = ldas<IOAS>(g + 7), chi2dd = ldas<IOAS>(g + 8);
    double denom = 1.0, wpp = 0.0;
    bool touched = false;
    if (a.calib > 1.0e-4) {
      wpp = 1 / (chi2pp + 1 / (a.calib * a.calib));
      chisq -= wpp * (chi2op * chi2op);
      denom /= wpp * (a.calib * a.calib);
      touched = true;
    }
    if (a.beam_uncertain) {
      const double wdd = 1 / (chi2dd - wpp * (chi2pd * chi2pd) + 1);
      const double t = chi2od - wpp * chi2op * chi2pd;
      chisq -= wdd * (t * t);
      denom /= wdd;
      touched = true;
    }
    if (touched && denom != 1.0) chisq += log(denom);
  }
  const double neg = chisq / 2;                          // :315
  if (a.neglnl) a.neglnl[c] = neg;
  if (a.lnl) stas<IOAS>(a.lnl + c, a.accumulate ? ldas<IOAS>(a.lnl + c) + (-neg) : (0.0 + (-neg)));
}

static __global__ __launch_bounds__(256) void k_cl_combine(ClCombineArgs a) {
  cl_combine_chain(a, static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x);
}

// ---- band powers ----------------------------------------------------------------------------
// BP[i][c] = BP0[i] + sum_b q[b][c] K[b][i] for the 16 chains of a block, on the FP64 matrix cores:
// one 16 x 16 tile of (chains x bands) per wave and pass, accumulator preloaded with BP0, k-steps
// of four bins in ascending order.  q: LDS [>= P][16] (bin-major, chain fastest - the A operand
// layout A[i = l&15][k = l>>4]); K: global [npar][nbp], its fragments B[k = l>>4][j = l&15] are
// row-contiguous 16-wide reads, fetched eight k-steps at a time ahead of the MFMAs that use
// them; out[band * ostride + chain] for chains < nvalid (LDS tile of the fused sampler, or the
// global band-major array of the batched path).  The same function serves k_cl_bandpower and
// k_cl_mh_run, so both engines see the same band powers bit for bit.  All 64*NW threads call it.
// KAS: address space of K and bp0 (HBM, or the LDS copy of a resident data set); OAS: of out.
// KC: K is the compact copy of the fused sampler - per column tile the rows of its non-zero k-steps
// only, [nbp/16][kstride rows][16] with kstride = 4 * (widest k-step range of any tile); row r of
// tile ct is bin 4 * ks0(ct) + r.
template <int NW, int KAS = 0, int OAS = 0, bool KC = false>
__device__ __forceinline__ void cl_bp_tile(const double* q, int P, const double* __restrict__ K,
                                           int kstride, const int32_t* __restrict__ krange,
                                           const double* __restrict__ bp0, int nb, int nbp,
                                           double* out, int64_t ostride, int nvalid) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lk = lane >> 4, lc = lane & 15;
  const int nct = nbp >> 4;
  constexpr int KB = 8;                                    // k-steps per fetch batch
  const auto K_t = as_ptr<KAS>(K);
  const auto bp0_t = as_ptr<KAS>(bp0);
  const auto q_t = as_ptr<3>(q);
  const auto out_t = as_ptr<OAS>(out);
  if constexpr (KC) {
    // compact K: kstride = 4 * kmax rows per tile, zero outside the tile's range and beyond bin P - 1,
    // so the k-loop needs no per-lane predicate: kmax k-steps for every tile (a zero k-step adds
    // exact zeros); krange is the fused sampler's LDS copy
    const int kmax = kstride >> 2;
    const auto kr_t = as_ptr<KAS>(krange);
    for (int ct = w; ct < nct; ct += NW) {
      const int band = ct * 16 + lc;
      const int ks0 = __builtin_amdgcn_readfirstlane(kr_t[2 * ct]);
      const double c0 = (band < nb) ? bp0_t[band] : 0.0;
      double4_t acc = double4_t{c0, c0, c0, c0};
      const auto kp = K_t + (static_cast<size_t>(ct) * kstride + lk) * 16 + lc;
      for (int k0 = 0; k0 < kmax; k0 += KB) {
        double bv[KB], av[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u < kmax) {                             // uniform
            const int bq = 4 * (ks0 + k0 + u) + lk;
            bv[u] = kp[(4 * (k0 + u)) * 16];
            av[u] = q_t[(bq < P ? bq : P - 1) * 16 + lc];  // rows beyond P - 1 meet zeros of K
          }
        }
#pragma unroll
        for (int u = 0; u < KB; ++u)
          if (k0 + u < kmax) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = lk + 4 * i;
        if (row < nvalid) out_t[static_cast<int64_t>(band) * ostride + row] = (band < nb) ? acc[i] : 0.0;
      }
    }
    return;
  }
  for (int ct = w; ct < nct; ct += NW) {
    const int band = ct * 16 + lc;
    const double c0 = (band < nb) ? bp0_t[band] : 0.0;
    double4_t acc = double4_t{c0, c0, c0, c0};
    const auto kp = KC ? K_t + (static_cast<size_t>(ct) * kstride + lk) * 16 + lc
                       : K_t + static_cast<size_t>(lk) * kstride + band;
    // K is banded (a band's window meets a few bins): only the k-steps (groups of four bins) that
    // hold a non-zero of this column tile are multiplied - the others add exact zeros
    const int ks0 = __builtin_amdgcn_readfirstlane(krange[2 * ct]);      // wave-uniform: scalar loop control
    const int ks1 = __builtin_amdgcn_readfirstlane(krange[2 * ct + 1]);
    for (int k0 = ks0; k0 < ks1; k0 += KB) {
      double bv[KB], av[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int b = 4 * (k0 + u) + lk;
        const bool live = b < P && k0 + u < ks1;
        bv[u] = live ? (KC ? kp[(4 * (k0 + u - ks0)) * 16] : kp[static_cast<size_t>(4 * (k0 + u)) * kstride]) : 0.0;
        av[u] = live ? q_t[b * 16 + lc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < KB; ++u)
        if (k0 + u < ks1)                                  // uniform
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lk + 4 * i;
      if (row < nvalid) out_t[static_cast<int64_t>(band) * ostride + row] = (band < nb) ? acc[i] : 0.0;
    }
  }
}

// batched path: q parameter-major [npar][n] -> bp [nbp][n]; a block of 256 threads per 16 chains
static __global__ __launch_bounds__(256) void k_cl_bandpower(int64_t n, int npar, int nb, int nbp,
                                                      const double* __restrict__ q,
                                                      const double* __restrict__ K,
                                                      const int32_t* __restrict__ krange,
                                                      const double* __restrict__ bp0,
                                                      double* __restrict__ bp) {
  extern __shared__ double ql[];                           // [npar][16]
  const int64_t c0 = static_cast<int64_t>(blockIdx.x) * 16;
  const int nvalid = static_cast<int>(n - c0 < 16 ? n - c0 : 16);
  for (int e = threadIdx.x; e < npar * 16; e += 256) {
    const int b = e >> 4, r = e & 15;
    ql[e] = (r < nvalid) ? q[static_cast<int64_t>(b) * n + c0 + r] : 0.0;
  }
  __syncthreads();
  cl_bp_tile<4>(ql, npar, K, cl_gstride(nbp), krange, bp0, nb, nbp, bp + c0, n, nvalid);
}

struct ClBinning {
  int nCl, npar, lmax;     // model lmax (largest binned multipole)
  const double* shapefun;  // [nCl][lmax+1]
  const int32_t* binof;    // [nCl][lmax+1] flat parameter index of the bin holding l, -1 if none
};

__device__ __forceinline__ double cl_ellnorm(int l) {     // binnedClModel.py:216-219
  if (l == 0) return 0.0;
  const double e = static_cast<double>(l);
  return 2 * 3.141592653589793 / (e * (e + 1.0));
}

// K[b][i], BP0[i] of one dataset: one thread per band walks its window (set-up, runs once).
static __global__ void k_cl_build_K(ClBinning m, int nb, int nbp, int wl, const double* __restrict__ window,
                             const int32_t* __restrict__ win_min, const int32_t* __restrict__ win_max,
                             int has_pol, const uint8_t* __restrict__ inc_pol, double* K,
                             double* bp0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  const int lo = win_min[i];
  int maxl = win_max[i] + 1;                              // :436
  if (maxl > m.lmax + 1) maxl = m.lmax + 1;
  if (maxl > wl) maxl = wl;
  double z = 0.0;
  const int ncl_used = (has_pol && inc_pol[i]) ? 3 : 1;   // :437-442
  for (int iCl = 0; iCl < ncl_used && iCl < m.nCl; ++iCl)
    for (int l = lo; l < maxl; ++l) {
      const double t = m.shapefun[iCl * (m.lmax + 1) + l] * cl_ellnorm(l) *
                       window[(static_cast<size_t>(i) * 3 + iCl) * wl + l];
      const int b = m.binof[iCl * (m.lmax + 1) + l];
      if (b < 0) z += t;
      else K[b * cl_gstride(nbp) + i] += t;
    }
  bp0[i] = z;
}

// binnedClModel.__call__ (:62-86): Cl[c][iCl][l] = (shape * q_b) * ellnorm, rows >= nCl zero.
static __global__ __launch_bounds__(256) void k_cl_model(ClBinning m, int64_t n, const double* __restrict__ q
                                                  /*[n][npar]*/, double* __restrict__ out /*[n][3][lmax+1]*/) {
  const int L = m.lmax + 1;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * 3 * L) return;
  const int l = static_cast<int>(e % L);
  const int iCl = static_cast<int>((e / L) % 3);
  const int64_t c = e / (3 * L);
  double v = 0.0;
  if (iCl < m.nCl) {
    v = m.shapefun[iCl * L + l];
    const int b = m.binof[iCl * L + l];
    if (b >= 0) v *= q[c * m.npar + b];
    v *= cl_ellnorm(l);
  }
  out[e] = v;
}

// getWinBandpower (:422-446) for explicit spectra: one wave per (chain, band).
// Cl host layout [n][rows][L]; rows = 1 (TT only, ndim==1 in the reference) or 3.
static __global__ __launch_bounds__(256) void k_cl_bandpower_direct(
    int64_t n, int rows, int L, const double* __restrict__ Cl, int nb, int wl,
    const double* __restrict__ window, const int32_t* __restrict__ win_min,
    const int32_t* __restrict__ win_max, int has_pol, const uint8_t* __restrict__ inc_pol,
    double* __restrict__ bp /*[nbp][n]*/) {
  const int64_t wv = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wv >= n * nb) return;
  const int64_t c = wv / nb;
  const int i = static_cast<int>(wv % nb);
  const int lo = win_min[i];
  int maxl = win_max[i] + 1;
  if (maxl > L) maxl = L;
  if (maxl > wl) maxl = wl;
  const int ncl_used = (has_pol && inc_pol[i] && rows >= 3) ? 3 : 1;
  double s = 0.0;
  for (int iCl = 0; iCl < ncl_used; ++iCl)
    for (int l = lo + lane; l < maxl; l += 64)
      s = fma(Cl[(c * rows + iCl) * L + l], window[(static_cast<size_t>(i) * 3 + iCl) * wl + l], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) bp[static_cast<int64_t>(i) * n + c] = s;
}

// prior (:125-132, positive_corr): 0 if any auto-spectrum bin is negative
static __global__ void k_cl_prior(int64_t n, int npar, const int32_t* __restrict__ cltype,
                           const double* __restrict__ q /*[n][npar]*/, double* __restrict__ prior) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double p = 1.0;
  for (int b = 0; b < npar; ++b)
    if (cltype[b] != 1 && q[c * npar + b] < 0) p = 0.0;
  prior[c] = p;
}

// getdelta (:377-398) of explicit band-power vectors: BP host layout [n][nb] -> delta [n][nb]
static __global__ void k_cl_delta(int64_t n, int nb, const double* __restrict__ BP,
                           const double* __restrict__ obs, const double* __restrict__ xf,
                           const uint8_t* __restrict__ hasx, int has_x, double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * nb) return;
  const int k = static_cast<int>(e % nb);
  const double v = BP[e];
  if (has_x && hasx[k]) {
    double zth = v + xf[k];
    if (zth <= 0.0) zth = 0.0;
    out[e] = obs[k] - log(zth);
  } else {
    out[e] = obs[k] - v;
  }
}

// getlinform (:407-412): out[c][i] = sum_k N^-1[i][k] vec[c][k]  (or vec/var)
static __global__ __launch_bounds__(256) void k_cl_linform(int64_t n, int nb, int nbp,
                                                    const double* __restrict__ vec,
                                                    const double* __restrict__ ninv,
                                                    const double* __restrict__ var,
                                                    double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * nb) return;
  const int i = static_cast<int>(e % nb);
  const int64_t c = e / nb;
  if (!ninv) {
    out[e] = vec[e] / var[i];
    return;
  }
  double s = 0.0;
  for (int k = 0; k < nb; ++k) s = fma(ninv[static_cast<size_t>(i) * cl_gstride(nbp) + k], vec[c * nb + k], s);
  out[e] = s;
}

// [n][cols] -> [colsp][n] with zero padding rows
static __global__ void k_cl_to_bandmajor(int64_t n, int cols, int colsp, const double* __restrict__ in,
                                  double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * colsp) return;
  const int k = static_cast<int>(e / n);
  const int64_t c = e % n;
  out[e] = (k < cols) ? in[c * cols + k] : 0.0;
}

}  // namespace mi355
