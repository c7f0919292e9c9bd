// Fused Metropolis-Hastings over the binned-C_l likelihood: ONE launch advances every chain by
// nsteps steps (propose -> band powers -> delta rows -> MFMA quadratic form -> marginalisation
// -> accept -> moments), where the step-wise engine (gmh_kernels.hpp + cl_kernels.hpp) needs five
// dependent launches per step and is bound by their latency (profiles/r1_clbeam_summary.txt:
// 38 us per step for 4096 chains, nb = 64).
//
// A block of 64*NW threads owns kClMhChains = 16 chains = the 16 rows of one MFMA tile.  Thread
// (chain, group) keeps the position, proposal widths and running moments of its parameters
// (group, group + G, ...) in registers for the whole launch; the proposal of a step goes to LDS
// [P][16] as the A operand of the band-power MFMAs (cl_bp_tile, the function k_cl_bandpower
// runs), the band powers to LDS as the row tile of cl_form_tile (the function k_cl_form runs):
// a chain's log-likelihood comes out of the same instructions in the same order as in the
// step-wise engine.  Every other piece (indices into the Philox streams, proposal arithmetic,
// accept rule, Welford and co-moment updates) restates k_gmh_propose / k_gmh_accept, with the
// free-running draws of several steps generated together into an LDS cache.
// tests/test_cl_mh_fused_gpu.py holds the two engines to bit-identical chains in every proposal
// mode and for every shape of the likelihood; tests/test_gmh_gpu.py runs the reference's own
// chains (G5) through both, draw for draw.
//
// Instantiations: <NW, FORM, RES>.  One data set: compiled for its form (plain / symmetrised
// MFMA form, analytic-marginalisation triple form, uncorrelated errors), with its band tables and
// a compact copy of K (the non-zero k-steps of every column tile) in LDS for the launch; RES = its
// N^-1 too (grids of at most one block per CU: a step then touches L2 / HBM only for traces and
// co-moments), else the matrix is streamed from L2 and two blocks share a CU.  Several data sets,
// or tables that do not fit: the generic kernel (FORM = CLMH_ANY), everything from L2.
//
// Measured (MI355X, 4096 chains, nb = 64, 32 bins; profiles/r2_clmh_summary.txt): 6.5 us per step
// of all chains (step-wise engine: 35); a step is a chain of dependent LDS / MFMA / barrier
// latencies with one wave per SIMD - about 1200 instructions (716 VALU, 19 MFMA) and 15 000 cycles
// per wave and step, the MFMA pipe 8 % busy; halving the chains per block to double the blocks per
// CU was measured and lost (10.9 vs 8.9 us per step at the time: the idle tile rows cost more than
// the overlap wins).
//
// Reference semantics: MCMC.py:225-286 (sample), Proposal.py:127-164 (getNewParams),
// binnedClModel.py:125-132 (prior), ClData_CosmoMC.py:267-315 (calcLnLike), MCMC.py:290-340
// (moments).
#pragma once
#include "cl_kernels.hpp"
#include "gmh_device.hpp"

namespace mi355 {

constexpr int kClMhChains = 16;
constexpr int kClMhMaxData = 4;      // data sets whose descriptors travel in the kernel arguments

struct ClMhData {            // one data set (device pointers)
  ClFormArgs f;              // its tables; bp / n / rows / out are filled in per block
  const double* K;           // [npar][kstride]
  const double* bp0;         // [nbp]
  const int32_t* krange;     // [nbp/16][2]: the k-steps of K that hold a non-zero of each column tile
  int has_corr;              // full covariance (MFMA form) or variances only
  int rows_per_chain;        // 1 / 11 / 121 marginalisation points, or 3 (analytic: TRIPLE)
  int sym;                   // the form of this data set runs on the symmetrised matrix
  int kstride;               // row stride of K in doubles (cl_gstride in HBM); rows per tile of the compact LDS copy
  int kmax;                  // widest non-zero k-step range of a column tile of K (compact copy: 4 * kmax rows per tile)
  int pad;
};

struct ClMhArgs {
  GmhArgs g;                 // chain state and proposal; step / nsamp / blk of the FIRST step;
                             // inj_n / inj_u / tr_* are the bases of the whole run here
  ClMhData data[kClMhMaxData];
  int ndata;
  int nbp_max;               // largest padded band count of the data sets
  int fout_per_chain;        // doubles of form output per chain: max(rows_per_chain, 9)
  int tile_doubles;          // LDS doubles of the form tile region (form_lds_bytes / 8)
  int resident;              // 1: the (single) data set's N^-1 is copied into LDS at launch too
  int tables;                // 1: single data set whose tables and compact K live in LDS (FORM != ANY)
  int sb;                    // steps whose Philox draws one refill of the draw cache covers
  int rep;                   // k_cl_mh_wave: waves (replicas) per 16 chains that share out a marginalisation grid
  const double* weights;     // [11] margeweights
  double margenorm;
  int64_t nsteps;
  int thin;
  int64_t nkeep;
};

// Row stride (doubles) of the LDS copy of N^-1: = 16 (mod 32), so that the four 16-wide rows a
// wave reads per MFMA operand fall on disjoint halves of the banks.
__host__ __device__ inline int cl_mh_nstride(int nbp) { return (nbp % 32 == 0) ? nbp + 16 : nbp; }

// LDS doubles of a single data set's tables: compact K (4 * kmax rows of 16 per column tile), BP0,
// obs, x-factors, beam errors, variances, the beam-factor rows (all 11 when the beam is marginalised
// numerically, else the centre row), has_xfactor bytes
__host__ __device__ inline int cl_mh_bfac_rows(int nbeam) { return nbeam > 1 ? kClGrid : 1; }
__host__ __device__ inline size_t cl_mh_table_doubles(int nbp, int kmax, int nbeam) {
  return static_cast<size_t>(nbp) * 4 * kmax + static_cast<size_t>(5 + cl_mh_bfac_rows(nbeam)) * nbp + nbp / 8 + 1 +
         nbp / 16 + 1;                                     // + the K ranges (int32 pairs)
}
// ... and of the resident copy of its N^-1 (RES kernels, full covariance): kClMhSlackRows zeroed rows
// behind it for the fetches past its end (4 rows per k-step of prefetch depth kClMhResPF)
constexpr int kClMhResPF = 4;
constexpr int kClMhSlackRows = 4 * kClMhResPF;
__host__ __device__ inline size_t cl_mh_matrix_doubles(int nbp, int has_corr) {
  return has_corr ? static_cast<size_t>(nbp + kClMhSlackRows) * cl_mh_nstride(nbp) : 0;
}

// doubles of dynamic LDS the kernel needs (host and device agree through this one function)
__host__ __device__ inline size_t cl_mh_lds_doubles(int P, int nblock, int npn, int sb, int nbp_max,
                                                    int fout_per_chain, int tile_doubles,
                                                    size_t resident_doubles) {
  const size_t CB = kClMhChains;
  size_t n = static_cast<size_t>(tile_doubles) + static_cast<size_t>(nbp_max) * CB + fout_per_chain * CB;
  n += 3 * static_cast<size_t>(P) * CB;                  // thn, dl, z
  n += static_cast<size_t>(sb) * (npn + 1) * CB;         // draw cache: normals, uniforms
  n += 4 * CB;                                           // lnPr, lnl, dec, spare
  n += (3 * static_cast<size_t>(P) + 1) / 2 + 1;         // block_of, rank_in_block, cltype (int32)
  n += CB;                                               // pbad (int32) + naccept (uint32)
  n += (static_cast<size_t>(nblock > 0 ? nblock : 1) * CB + 1) / 2;   // accept counters (uint32)
  return n + resident_doubles;
}

// parameters a thread keeps in registers: thread (chain, group) owns i = group + G * m, m < kClMhSlots
constexpr int kClMhSlots = 4;

// steps per refill of the draw cache: about two Philox blocks per thread and refill
__host__ __device__ inline int cl_mh_cache_steps(int nthreads, int npn) {
  int sb = (2 * nthreads / kClMhChains) / (npn > 0 ? npn : 1);
  return sb < 1 ? 1 : (sb > 32 ? 32 : sb);
}

// Which form a resident (single data set) kernel is compiled for; the generic kernel (several data
// sets, or matrices too large for LDS) dispatches at run time.
enum ClMhForm : int { CLMH_ANY = 0, CLMH_MARGE = 1, CLMH_SYM = 2, CLMH_TRIPLE = 3, CLMH_DIAG = 4 };

__host__ __device__ inline int cl_mh_form_kind(const ClMhData& D) {
  if (!D.has_corr) return CLMH_DIAG;
  if (D.f.mode == CL_ROWS_TRIPLE) return CLMH_TRIPLE;
  return D.sym ? CLMH_SYM : CLMH_MARGE;
}

template <int NW, int FORM, int PF, int TAS, int MAS>
__device__ __forceinline__ void cl_mh_form(const ClMhData& D, const ClFormArgs& f, double* tile) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  const int kind = FORM != CLMH_ANY ? FORM : cl_mh_form_kind(D);
  if (kind == CLMH_DIAG) {
    const int64_t work = (f.mode == CL_ROWS_TRIPLE) ? f.n : f.rows;
    for (int64_t i = tid; i < work; i += NT) cl_form_diag_item<TAS, 3>(f, i);
    return;
  }
  const int rows_used = (kind == CLMH_TRIPLE) ? 15 : 16;
  const int ntiles = static_cast<int>((f.rows + rows_used - 1) / rows_used);
  for (int it = 0; it < ntiles; ++it) {
    if (it) __syncthreads();                             // the previous tile is done with `tile`
    const int64_t row0 = static_cast<int64_t>(it) * rows_used;
    if (kind == CLMH_TRIPLE) cl_form_tile<1, 1, CL_ROWS_TRIPLE, NW, false, PF, TAS, 3, MAS>(f, row0, tile);
    else if (kind == CLMH_SYM) cl_form_tile<1, 1, CL_ROWS_MARGE, NW, true, PF, TAS, 3, MAS>(f, row0, tile);
    else cl_form_tile<1, 1, CL_ROWS_MARGE, NW, false, PF, TAS, 3, MAS>(f, row0, tile);
  }
}

// CMR (resident 4-wave kernels, P <= kClMhCmrMaxP): the running co-moments of the block's chains live in
// registers for the launch - thread (chain, group g) owns rows g and P-1-g of the chain's lower
// triangle (P+1 entries per pair of rows: the triangle dealt evenly over the 16 groups) - instead
// of being read and written in HBM at every step (528 entries per chain at P = 32: the traffic that
// bounded a step at 18 us for 4096 chains).  Same update, entry by entry, as k_gmh_accept.
constexpr int kClMhCmrMaxP = 32;
template <int NW, int FORM, bool RES, bool CMR = false>
__global__ __launch_bounds__(64 * NW, (NW == 4 && RES) ? 1 : 2) void k_cl_mh_run(const ClMhArgs* __restrict__ ap) {
  static_assert(!CMR || (RES && NW == 4), "co-moments in registers: the 512-register kernels");
  // The arguments live in device memory and are read where they are used (scalar loads): passed
  // by value, the ~1 KB of pointers and counters stays in SGPRs for the whole launch and spills.
  const ClMhArgs& a = *ap;
  static_assert(!RES || FORM != CLMH_ANY, "a resident kernel is compiled for its form");
  constexpr int NT = 64 * NW, CB = kClMhChains, G = NT / CB, MS = kClMhSlots;
  constexpr bool TABS = FORM != CLMH_ANY;                // single data set: tables and compact K in LDS
  constexpr int TAS = TABS ? 3 : 1;                      // address space of the tables / K
  constexpr int MAS = RES ? 3 : 1;                       // ... and of N^-1
  extern __shared__ double smem[];
  const GmhArgs& g = a.g;
  const int P = g.pb.P;                                   // <= MS * G (host check)
  const int64_t N = g.N;
  const int tid = threadIdx.x, cl = tid & (CB - 1), grp = tid >> 4;
  const int64_t nbase = static_cast<int64_t>(blockIdx.x) * CB;
  const int64_t n = nbase + cl;
  const bool valid = n < N;
  const int nblk = g.nblock > 0 ? g.nblock : 1;

  // ---- LDS carve-up (cl_mh_lds_doubles) ----
  double* tile = smem;
  double* bp = tile + a.tile_doubles;                    // [nbp_max][CB]
  double* fout = bp + static_cast<size_t>(a.nbp_max) * CB;
  double* thn = fout + a.fout_per_chain * CB;            // [P][CB]: the proposal (A operand of BP)
  double* dl = thn + P * CB;                             // [P][CB]: x - mean, for the co-moments
  double* z = dl + P * CB;                               // [P][CB]: injected normals (CORR / rotated)
  double* zc = z + P * CB;                               // [sb * npn][CB] normals of the cached steps
  double* uc = zc + static_cast<size_t>(a.sb) * g.npn * CB;   // [sb][CB] uniforms
  double* lnpr = uc + static_cast<size_t>(a.sb) * CB;    // [CB] each
  double* lnl = lnpr + CB;
  double* dec = lnl + CB;
  int32_t* block_of = reinterpret_cast<int32_t*>(dec + 2 * CB);
  int32_t* rank_of = block_of + P;
  int32_t* cltype = rank_of + P;
  int32_t* pbad = reinterpret_cast<int32_t*>(dec + 2 * CB + (3 * static_cast<size_t>(P) + 1) / 2 + 1);
  uint32_t* nacc = reinterpret_cast<uint32_t*>(pbad + CB);
  uint32_t* accb = nacc + CB;                            // [nblk][CB]
  double* resid = reinterpret_cast<double*>(pbad) + CB + (static_cast<size_t>(nblk) * CB + 1) / 2;

  for (int i = tid; i < P; i += NT) {
    block_of[i] = g.block_of[i];
    rank_of[i] = g.rank_in_block[i];
    cltype[i] = g.pb.cltype[i];
  }
  // chain state of the thread's parameters, in registers for the whole launch
  double rth[MS], rsg[MS], rmu[MS], rm2[MS], rnew[MS];
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    const int i = grp + G * m;
    const bool on = valid && i < P;
    rth[m] = on ? g.theta[i * N + n] : 0.0;
    rsg[m] = on ? g.sigma[i * N + n] : 0.0;
    rmu[m] = on ? g.mean[i * N + n] : 0.0;
    rm2[m] = on ? g.m2[i * N + n] : 0.0;
    rnew[m] = rth[m];
  }
  // CMR: rows rowA = group (< 16) and rowB = P-1-group (< 32) of the chain's co-moment triangle
  constexpr int CA = CMR ? kClMhCmrMaxP / 2 : 1, CBR = CMR ? kClMhCmrMaxP : 1;
  double cmA[CA], cmB[CBR];
  const int rowA = grp, rowB = P - 1 - grp;
  const bool hasA = CMR && valid && rowA <= rowB;            // the middle row of an odd P is an A row
  const bool hasB = CMR && valid && rowB > rowA;
  if constexpr (CMR) {
#pragma unroll
    for (int j = 0; j < CA; ++j)
      cmA[j] = (hasA && j <= rowA) ? ldas<1>(g.comom + (static_cast<int64_t>(rowA) * P + j) * N + n) : 0.0;
#pragma unroll
    for (int j = 0; j < CBR; ++j)
      cmB[j] = (hasB && j <= rowB) ? ldas<1>(g.comom + (static_cast<int64_t>(rowB) * P + j) * N + n) : 0.0;
  }
  for (int b = grp; b < nblk; b += G)
    accb[b * CB + cl] = (valid && g.nblock) ? g.accept[static_cast<int64_t>(b) * N + n] : 0u;
  if (grp == 0) {
    lnpr[cl] = valid ? g.lnPr[n] : 0.0;
    nacc[cl] = valid ? g.naccept[n] : 0u;
    pbad[cl] = 0;
  }
  // Single data set (FORM != ANY): its band tables and the compact K move to LDS once per launch;
  // RES: N^-1 (the copy its form uses) too, and a step touches HBM / L2 only for the trace and the
  // co-moments.
  ClMhData D0 = a.data[0];
  if constexpr (TABS) {
    const int nbp = D0.f.nbp;
    double* p = resid;
    if constexpr (RES) {
      if (D0.has_corr) {
        const int ns = cl_mh_nstride(nbp), gs = D0.f.nstride;
        const double* src = D0.sym ? D0.f.nsym : D0.f.ninv;
        // the whole copy, padding columns and the kClMhSlackRows slack rows included (zero there: the
        // form fetches its prefetch depth past the end, and what it fetches must not be garbage)
        for (int e = tid; e < (nbp + kClMhSlackRows) * ns; e += NT) {
          const int r = e / ns, cc = e - r * ns;
          p[e] = (r < nbp && cc < nbp) ? src[r * gs + cc] : 0.0;
        }
        D0.f.ninv = D0.f.nsym = p;
        D0.f.nstride = ns;
        p += static_cast<size_t>(nbp + kClMhSlackRows) * ns;   // slack rows: the form fetches ahead
      }
    }
    {
      // compact K: tile ct holds bins 4 ks0(ct) .. 4 ks0(ct) + 4 kmax - 1 (zero beyond its range / P)
      const int rows = 4 * D0.kmax, per = rows * 16;
      for (int e = tid; e < (nbp >> 4) * per; e += NT) {
        const int ct = e / per, r = (e - ct * per) >> 4, cc = e & 15;
        const int ks0 = D0.krange[2 * ct], ks1 = D0.krange[2 * ct + 1];
        const int b = 4 * ks0 + r;
        p[e] = (r < 4 * (ks1 - ks0) && b < P) ? D0.K[static_cast<size_t>(b) * D0.kstride + ct * 16 + cc] : 0.0;
      }
      D0.K = p;
      D0.kstride = rows;
      p += static_cast<size_t>(nbp >> 4) * per;
    }
    const int nbf = cl_mh_bfac_rows(D0.f.nbeam);
    double* t_bp0 = p;
    double* t_obs = p + nbp;
    double* t_xf = p + 2 * nbp;
    double* t_be = p + 3 * nbp;
    double* t_var = p + 4 * nbp;
    double* t_bf = p + 5 * nbp;
    uint8_t* t_hx = reinterpret_cast<uint8_t*>(p + (5 + nbf) * nbp);
    int32_t* t_kr = reinterpret_cast<int32_t*>(p + (5 + nbf) * nbp + nbp / 8 + 1);   // [nbp/16][2]
    for (int e = tid; e < nbp; e += NT) {
      t_bp0[e] = D0.bp0[e];
      t_obs[e] = D0.f.obs[e];
      t_xf[e] = D0.f.xf[e];
      t_be[e] = D0.f.beam_err[e];
      t_var[e] = D0.has_corr ? 1.0 : D0.f.var[e];
      t_hx[e] = D0.f.hasx[e];
    }
    // beam factors: all 11 rows, or the centre row alone placed where row kClHalfSteps is expected
    if (nbf == kClGrid) {
      for (int e = tid; e < kClGrid * nbp; e += NT) t_bf[e] = D0.f.bfac[e];
      D0.f.bfac = t_bf;
    } else {
      for (int e = tid; e < nbp; e += NT) t_bf[e] = D0.f.bfac[kClHalfSteps * nbp + e];
      D0.f.bfac = t_bf - static_cast<ptrdiff_t>(kClHalfSteps) * nbp;
    }
    for (int e = tid; e < (nbp >> 4) * 2; e += NT) t_kr[e] = D0.krange[e];
    D0.krange = t_kr;
    D0.bp0 = t_bp0;
    D0.f.obs = t_obs;
    D0.f.xf = t_xf;
    D0.f.beam_err = t_be;
    D0.f.var = t_var;
    D0.f.hasx = t_hx;
  }
  __syncthreads();

  // per-slot constants of the thread's parameters (block, rank in the block, prior type) in registers:
  // three dependent LDS lookups fewer on the proposal's latency chain
  int rblk[MS], rrank[MS], rtype[MS];
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    const int i = grp + G * m;
    rblk[m] = i < P ? block_of[i] : -1;
    rrank[m] = i < P ? rank_of[i] : 0;
    rtype[m] = i < P ? cltype[i] : 1;
  }
  const bool inject = g.inj_n != nullptr;
  const double* e2t = kExp2Table;                          // accept test: as k_gmh_accept
  int blk = g.blk;

  // getNewParams (Proposal.py:127-164) + prior (binnedClModel.py:125-132) of step t into rnew / thn.
  // Free-running draws: every sb steps all threads together fill the cache with normal(idx), idx in
  // [step * npn, (step + sb) * npn), and uniform(step .. step + sb - 1) of the block's chains - the
  // values philox_normal / philox_uniform return for those indices (one Philox block + Box-Muller
  // serves both halves of a pair).
  int ts = 0;                                              // t % sb, kept without a division
  auto propose = [&](int64_t t) {
    const int64_t step = g.step + t;
    const uint64_t base = static_cast<uint64_t>(step) * g.npn;
    const double* inj = inject ? g.inj_n + (t * N + n) * P : nullptr;   // this chain's row
    auto in_block = [&](int i) { return g.nblock == 0 || block_of[i] == blk; };
    if (!inject && ts == 0) {
      const uint64_t idx0 = base, cnt = static_cast<uint64_t>(a.sb) * g.npn;
      const uint64_t p0 = idx0 >> 1, np = ((idx0 + cnt - 1) >> 1) - p0 + 1;
      const uint64_t s0 = static_cast<uint64_t>(step), q0 = s0 >> 1, nq = ((s0 + a.sb - 1) >> 1) - q0 + 1;
      for (uint64_t it = tid; it < (np + nq) * CB; it += NT) {
        const int c = static_cast<int>(it & (CB - 1));
        const uint64_t q = it >> 4;
        if (q < np) {
          double za, zb;
          philox_normal2(g.seed, RNG_NORMAL, nbase + c, p0 + q, za, zb);
          const uint64_t ia = 2 * (p0 + q) - idx0;         // wraps below idx0: fails the range test
          if (ia < cnt) zc[ia * CB + c] = za;
          if (ia + 1 < cnt) zc[(ia + 1) * CB + c] = zb;
        } else {
          double ua, ub;
          philox_uniform2(g.seed, RNG_UNIFORM, nbase + c, q0 + (q - np), ua, ub);
          const uint64_t ia = 2 * (q0 + (q - np)) - s0;
          if (ia < static_cast<uint64_t>(a.sb)) uc[ia * CB + c] = ua;
          if (ia + 1 < static_cast<uint64_t>(a.sb)) uc[(ia + 1) * CB + c] = ub;
        }
      }
      __syncthreads();
    }
    const double* zrow = zc + static_cast<size_t>(ts) * g.npn * CB + cl;   // normal(base + r) = zrow[r * CB]
    if (g.mode == GMH_UNCORR) {
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        const int i = grp + G * m;
        if (i < P) {
          double v = rth[m];
          if (g.nblock == 0 || rblk[m] == blk) {           // :151, :156-159
            const double zi = valid ? (inject ? ldas<1>(inj + i) : zrow[rrank[m] * CB]) : 0.0;
            v += zi * rsg[m];
          }
          rnew[m] = v;
          thn[i * CB + cl] = v;
          if (rtype[m] != 1 && v < 0) pbad[cl] = 1;
        }
      }
    } else {
      // z_j of the step: CORR normal(base + j) for every j, rotated normal(base + rank_j) for the
      // block's members; injected rows are staged in z, free-running draws are read in the cache
      if (inject) {
        if (valid) {
          for (int j = grp; j < P; j += G)
            if (g.mode == GMH_CORR) z[j * CB + cl] = ldas<1>(inj + j);
            else if (in_block(j)) z[j * CB + cl] = ldas<1>(inj + rank_of[j]);
        }
        __syncthreads();
      }
      auto zval = [&](int j) {
        return inject ? z[j * CB + cl] : (g.mode == GMH_CORR ? zrow[j * CB] : zrow[rank_of[j] * CB]);
      };
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        const int i = grp + G * m;
        if (i < P) {
          double v = rth[m];
          if (valid) {
            const double* Li = g.L + static_cast<int64_t>(i) * P * N + n;
            if (g.mode == GMH_CORR) {
              if (in_block(i)) {                           // :153-159
                double s = 0.0;
#pragma unroll 4
                for (int j = 0; j < P; ++j) s += ldas<1>(Li + j * N) * zval(j);
                v += s * rsg[m];
              }
            } else {                                       // :145-148, sum() starts from 0
              double s = 0.0;
              for (int j = 0; j < P; ++j)
                if (in_block(j)) s += ldas<1>(Li + j * N) * zval(j);
              v += s * rsg[m];
            }
          }
          rnew[m] = v;
          thn[i * CB + cl] = v;
          if (cltype[i] != 1 && v < 0) pbad[cl] = 1;
        }
      }
    }
  };

#ifdef MI355_CLMH_PROF
  long long pt[6] = {0, 0, 0, 0, 0, 0};
  long long pc = clock64();
#define CLMH_TICK(k) do { __syncthreads(); const long long now_ = clock64(); pt[k] += now_ - pc; pc = now_; } while (0)
#else
#define CLMH_TICK(k) do {} while (0)
#endif
  // -ln L of one data set for the proposals in thn -> lnl (accumulated over the data sets)
  auto dataset = [&](const ClMhData& D, int id) {
    // BP[i][c] = BP0[i] + sum_b q[b][c] K[b][i] on the matrix cores (the function k_cl_bandpower runs)
    cl_bp_tile<NW, TAS, 3, TABS>(thn, P, D.K, D.kstride, D.krange, D.bp0, D.f.nb, D.f.nbp, bp, CB, CB);
    __syncthreads();
    CLMH_TICK(1);
    ClFormArgs f = D.f;
    f.bp = bp;
    f.n = CB;
    f.rows = static_cast<int64_t>(CB) * D.rows_per_chain;
    f.out = fout;
    static_assert(4 * kClMhResPF <= kClMhSlackRows, "resident N^-1: slack rows cover the prefetch depth");
    cl_mh_form<NW, FORM, RES ? kClMhResPF : 8, TAS, MAS>(D, f, tile);   // streamed matrices: fetched 8 k-steps ahead
    __syncthreads();
    CLMH_TICK(2);
    if (tid < CB) {
      ClCombineArgs cmb{};
      cmb.form = fout;
      cmb.n = CB;
      cmb.mode = f.mode;
      cmb.ncal = f.ncal;
      cmb.nbeam = f.nbeam;
      cmb.beam_uncertain = f.beam_uncertain;
      cmb.calib = f.calib;
      cmb.weights = a.weights;
      cmb.margenorm = a.margenorm;
      cmb.lnl = lnl;
      cmb.accumulate = id > 0 ? 1 : 0;
      cmb.neglnl = nullptr;
      cl_combine_chain<3>(cmb, tid);
    }
    // (the next data set's band powers go to `bp`, which the combine does not read; its form
    // output is written after the barrier that follows them)
  };

  int thin_ctr = 0;
  int64_t row_next = 0;
  for (int64_t t = 0; t < a.nsteps; ++t) {
    // the proposal follows the previous step's moment update without a barrier: it needs this
    // thread's own registers only
    propose(t);
    __syncthreads();
    CLMH_TICK(0);
    // ---- lnLike(q) summed over the data sets (binnedClLikelihood.py:26-32) ----
    if constexpr (FORM != CLMH_ANY) {
      dataset(D0, 0);
    } else {
      for (int id = 0; id < a.ndata; ++id) dataset(a.data[id], id);
    }

    // ---- accept / reject (MCMC.py:236-279); thread tid < CB is chain cl = tid ----
    bool keep = false;                                     // (t + 1) % thin == 0, rows < nkeep
    if (a.thin > 0 && ++thin_ctr == a.thin) {
      thin_ctr = 0;
      keep = row_next < a.nkeep;
    }
    const int64_t row = row_next;
    if (keep) ++row_next;
    if (tid < CB) {
      bool accepted = false;
      if (valid) {
        const double prior = pbad[cl] ? 0.0 : 1.0;
        const double v = lnl[cl];
        const double prev = lnpr[cl];
        const bool bad = (v != v) || (v == -__builtin_inf());   // :238-243
        if (prior > 0.0 && !bad) {
          const double lnalpha = v - prev;                 // :251-252 (prior is 1: nothing to add)
          bool reject = false;
          if (!(lnalpha >= 0.0)) {
            const double u = g.inj_u ? ldas<1>(g.inj_u + t * N + n)
                                     : uc[ts * CB + cl];                        // = uniform(step)
            reject = !(exp_accept(lnalpha, e2t) > u);      // :264
          }
          accepted = !reject;
        }
        if (accepted) {
          lnpr[cl] = v;
          nacc[cl] += 1;                                   // :269
          if (g.nblock) accb[blk * CB + cl] += 1;          // :271
        }
        if (keep) {
          stas<1>(g.tr_lnPr + row * N + n, accepted ? v : prev);
          stas<1>(g.tr_acc + row * N + n, static_cast<uint8_t>(accepted ? 1 : 0));
        }
      }
      dec[cl] = accepted ? 1.0 : 0.0;
      pbad[cl] = 0;
    }
    __syncthreads();
    CLMH_TICK(3);

    // ---- position, running moments (MCMC.py:290-340), trace: registers only ----
    const double cnt = static_cast<double>(g.nsamp + t + 1);
    const double rn = rcp_nr(cnt);                         // as k_gmh_accept / the fused SED kernel
    const double cf = 1.0 - rn;
    {
      const bool accepted = dec[cl] != 0.0;
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        const int i = grp + G * m;
        if (i < P) {
          const double x = accepted ? rnew[m] : rth[m];
          rth[m] = x;
          const double dlt = x - rmu[m];
          if (g.comom != nullptr) dl[i * CB + cl] = dlt;
          rmu[m] = fma(dlt, rn, rmu[m]);
          rm2[m] = fma(dlt, x - rmu[m], rm2[m]);
          if (keep && valid) stas<1>(g.tr_theta + (row * P + i) * N + n, x);
        }
      }
    }
    if constexpr (CMR) {                                   // (launched with co-moments only)
      __syncthreads();
      // C_ij += (n-1)/n d_i d_j for the thread's two rows: (cf d_i) d_j, as k_gmh_accept
      const double cA = cf * (hasA ? dl[rowA * CB + cl] : 0.0);
      const double cB = cf * (hasB ? dl[rowB * CB + cl] : 0.0);
#pragma unroll
      for (int j = 0; j < CBR; ++j) {
        const double dj = dl[(j < P ? j : 0) * CB + cl];
        if (j < CA) {
          if (hasA && j <= rowA) cmA[j < CA ? j : 0] = cmA[j < CA ? j : 0] + cA * dj;
        }
        if (hasB && j <= rowB) cmB[j] = cmB[j] + cB * dj;
      }
    } else if (g.comom != nullptr) {                       // uniform over the grid
      __syncthreads();
      // C_ij += (n-1)/n d_i d_j, the columns dealt round-robin to the groups (k_gmh_accept)
      if (valid) gmh_comom_update(g.comom, N, n, P, cf, dl + cl, CB, grp, G);
    }
    if (g.nblock) blk = (blk + 1 == g.nblock) ? 0 : blk + 1;   // MCMC.py:278-279
    ts = (ts + 1 == a.sb) ? 0 : ts + 1;
  }
#ifdef MI355_CLMH_PROF
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
    printf("clmh block %d steps %lld: moments+propose %lld bp %lld form %lld accept %lld (ticks per step); "
           "form since load: stage %lld mfma %lld reduce %lld (cumulative ticks, block 0)\n",
           blockIdx.x, (long long)a.nsteps, pt[0] / a.nsteps, pt[1] / a.nsteps, pt[2] / a.nsteps,
           pt[3] / a.nsteps, g_clf[0], g_clf[1], g_clf[2]);
#endif

  // ---- state back to HBM ----
  if constexpr (CMR) {
#pragma unroll
    for (int j = 0; j < CA; ++j)
      if (hasA && j <= rowA) stas<1>(g.comom + (static_cast<int64_t>(rowA) * P + j) * N + n, cmA[j]);
#pragma unroll
    for (int j = 0; j < CBR; ++j)
      if (hasB && j <= rowB) stas<1>(g.comom + (static_cast<int64_t>(rowB) * P + j) * N + n, cmB[j]);
  }
  if (valid) {
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const int i = grp + G * m;
      if (i < P) {
        g.theta[i * N + n] = rth[m];
        g.mean[i * N + n] = rmu[m];
        g.m2[i * N + n] = rm2[m];
      }
    }
    if (g.nblock)
      for (int b = grp; b < nblk; b += G) g.accept[static_cast<int64_t>(b) * N + n] = accb[b * CB + cl];
    if (grp == 0) {
      g.lnPr[n] = lnpr[cl];
      g.naccept[n] = nacc[cl];
    }
  }
}

}  // namespace mi355
