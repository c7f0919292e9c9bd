// Fused Metropolis-Hastings over the binned-C_l likelihood, one WAVE per 16 chains.
//
// k_cl_mh_run (cl_mh_kernels.hpp) gives a block of 4 - 8 waves to the 16 chains of one MFMA row tile:
// every phase of a step is shared out over the waves and closed by a block barrier, and a step is a
// chain of those phases' latencies (15 000 cycles at nb = 64 with the matrix cores 8 % busy,
// profiles/r2_clmh_summary.txt).  Here a wave owns its 16 chains for the whole launch and never
// meets another wave after the tables are staged: no block barrier in the step loop, nothing of a
// step in LDS but the draw cache - the proposal, the band powers, the delta rows and the quadratic
// form live in registers, because the TRANSPOSED products land where the next one needs them:
//
//   lane l = (g = l >> 4, c = l & 15) holds the parameters 4 m + g of chain c (m < MS).  That is the
//   B operand B[k = g][j = c] of k-step m of   BP^T = K^T q^T   (A = K fragment, as cl_bp_tile's B),
//   whose result D[row = g + 4 i][col = c] of column tile ct is band 16 ct + g + 4 i = 4 s + g of
//   chain c with s = 4 ct + i - the B operand of k-step s of   Y^T = N^-T delta^T   (A = the N^-1
//   fragment, as cl_form_tile's B), whose result row g + 4 i of tile ct meets delta of the same band
//   in the same lane: the epilogue's products are register times register.
//
// Same products, same k order, same reduction tree as the batched kernels: the row sums are added per
// "virtual wave" exactly as cl_form_tile's NW waves add theirs (tiles dealt w, w + NW, ... or in
// snake order for the symmetrised matrix; 16-column butterfly, here two shuffles and two register
// adds; virtual waves in ascending order), so a chain's log-likelihood has the bits the step-wise
// engine gives it: tests/test_cl_mh_fused_gpu.py holds the two to identical chains.
// Marginalisation grids (11 or 11 x 11 points per chain, ClData_CosmoMC.py:317-372) are a loop over
// the points in the wave, the chi-squares of a row of the grid in registers.
//
// Serves: one data set with correlated errors whose likelihood is a MARGE form (x-factors; any grid),
// up to 4 MS parameters and 16 NCT bands.  Everything else: k_cl_mh_run / the step-wise engine.
// Reference semantics: MCMC.py:225-286, Proposal.py:127-164, binnedClModel.py:125-132,
// ClData_CosmoMC.py:267-372, MCMC.py:290-340.
#pragma once
#include "cl_mh_kernels.hpp"

namespace mi355 {

constexpr int kClMhwMaxWaves = 8;        // waves (16-chain samplers) per block
constexpr int kClMhwSlots = 8;           // parameters a lane keeps in registers: P <= 4 * kClMhwSlots

// LDS doubles of one wave: the draw cache; the proposal (B operand of the band powers: the k-steps of
// the banded K start at a run-time bin; injected normals are staged in the same words before it is
// written); the per-block accept counters; with co-moments x - mean of the step; with a marginalisation
// grid the marginalised chi-squares of its beam points
__host__ __device__ inline size_t cl_mhw_wave_doubles(int P, int nblock, int npn, int sb, int comom, int grid) {
  return static_cast<size_t>(sb) * (npn + 1) * 16 + static_cast<size_t>(P) * 16 +
         (static_cast<size_t>(nblock > 0 ? nblock : 1) * 16 + 1) / 2 + (comom ? static_cast<size_t>(P) * 16 : 0) +
         (grid ? kClGrid * 16 : 0);
}
// Marginalisation grids on few chains: `rep` waves (replicas) serve the SAME 16 chains - every one of
// them runs the whole sampler on identical state and draws, so they stay identical without talking -
// and share out the grid points of a step, whose chi-squares they exchange through LDS (one block
// barrier per step, two buffers): a job that leaves SIMDs idle at one wave per 16 chains spreads its
// 11 or 121 quadratic forms per chain over them.  Doubles of that exchange, per 16 chains:
__host__ __device__ inline size_t cl_mhw_xchg_doubles(int npoints) { return 2 * static_cast<size_t>(npoints) * 16; }
// ... and of what the waves of a block share: block / rank / type tables; the data set's band tables
// padded to the kernel's 16 NCT bands (BP0, obs, x-factors, beam-factor rows, has_xfactor bytes) so
// that a lane's band 4 s + g is an immediate offset from its own base; the compact K and its k-step
// ranges; N^-1 when resident, [16 NCT][16 NCT + 16] zero padded (a compile-time stride: immediate
// offsets again)
__host__ __device__ inline int cl_mhw_nstride(int nct_t) { return 16 * nct_t + 16; }
__host__ __device__ inline size_t cl_mhw_shared_doubles(int P, int nct_t, int nct, int kmax, int nbeam, int resident) {
  const size_t nbt = 16 * static_cast<size_t>(nct_t);
  return (3 * static_cast<size_t>(P) + 1) / 2 + 1 + (resident ? nbt * cl_mhw_nstride(nct_t) : 0) +
         static_cast<size_t>(nct) * 4 * kmax * 16 + nct + 1 + (3 + cl_mh_bfac_rows(nbeam)) * nbt + nbt / 8 + 1;
}
__host__ __device__ inline int cl_mhw_cache_steps(int npn) { return cl_mh_cache_steps(64, npn); }

__device__ __forceinline__ void cl_mhw_sync() {   // LDS hand-off inside one wave (its LDS operations execute in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// column tile of virtual wave w in pass p of cl_form_tile<.., NW, SYM> (-1: none)
template <int NW, int NCT, bool SYM>
__device__ __forceinline__ constexpr int cl_mhw_tile(int w, int p) {
  const int base = p * NW;
  if (base >= NCT) return -1;
  const int ct = SYM ? ((p & 1) ? base + NW - 1 - w : base + w) : base + w;
  return ct < NCT ? ct : -1;
}

// GRID: the likelihood marginalises numerically (a loop over the grid points, the band powers kept
// beside the delta row of the point); otherwise a chain has one row and delta takes the band powers'
// registers.
template <int NCT, int MS, bool SYM, bool RES, bool GRID>
__global__ __launch_bounds__(64 * kClMhwMaxWaves, 2) void k_cl_mh_wave(const ClMhArgs* __restrict__ ap) {
  static_assert(!SYM || NCT >= 8, "the batched form uses the symmetrised matrix from 8 column tiles up");
  const ClMhArgs& a = *ap;
  constexpr int CB = kClMhChains, G = 4;
  constexpr int NWREF = NCT <= 4 ? 4 : 8;                  // waves of the batched form whose sums are reproduced
  constexpr int NPASS = (NCT + NWREF - 1) / NWREF;
  constexpr int NK = 4 * NCT;                              // k-steps of the form = registers of a delta row
  constexpr int MAS = RES ? 3 : 1;
  extern __shared__ double smem[];
  const GmhArgs& g = a.g;
  const int P = g.pb.P;                                    // <= 4 MS (host check)
  const int64_t N = g.N;
  const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 4, cl = lane & 15;
  const int NT = blockDim.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rep = GRID ? a.rep : 1;                        // replicas per 16 chains (a power of two)
  const int tile = wv / rep, rid = wv - tile * rep;        // the wave's 16 chains within the block; its replica number
  const int64_t nbase = (static_cast<int64_t>(blockIdx.x) * ((NT >> 6) / rep) + tile) * CB;
  const int64_t n = nbase + cl;
  const bool valid = n < N;
  const int nblk = g.nblock > 0 ? g.nblock : 1;

  // ---- what the block shares: staged once, read-only afterwards ----
  int32_t* block_of = reinterpret_cast<int32_t*>(smem);
  int32_t* rank_of = block_of + P;
  int32_t* cltype = rank_of + P;
  for (int i = tid; i < P; i += NT) {
    block_of[i] = g.block_of[i];
    rank_of[i] = g.rank_in_block[i];
    cltype[i] = g.pb.cltype[i];
  }
  ClMhData D0 = a.data[0];
  const int nbp = D0.f.nbp, nb = D0.f.nb, nct = nbp >> 4;
  constexpr int NBT = 16 * NCT, NST = 16 * NCT + 16;       // padded bands; row stride of the resident N^-1
  double* p = smem + (3 * static_cast<size_t>(P) + 1) / 2 + 1;
  if constexpr (RES) {
    const int gs = D0.f.nstride;
    const double* src = SYM ? D0.f.nsym : D0.f.ninv;
    for (int e = tid; e < NBT * NST; e += NT) {
      const int r = e / NST, cc = e - r * NST;
      p[e] = (r < nbp && cc < nbp) ? src[r * gs + cc] : 0.0;
    }
    D0.f.ninv = D0.f.nsym = p;
    p += static_cast<size_t>(NBT) * NST;
  }
  {
    // compact K: tile ct holds bins 4 ks0(ct) .. 4 ks0(ct) + 4 kmax - 1 (zero beyond its range / P)
    const int rows = 4 * D0.kmax, per = rows * 16;
    for (int e = tid; e < nct * per; e += NT) {
      const int ct = e / per, r = (e - ct * per) >> 4, cc = e & 15;
      const int ks0 = D0.krange[2 * ct], ks1 = D0.krange[2 * ct + 1];
      const int b = 4 * ks0 + r;
      p[e] = (r < 4 * (ks1 - ks0) && b < P) ? D0.K[static_cast<size_t>(b) * D0.kstride + ct * 16 + cc] : 0.0;
    }
    D0.K = p;
    D0.kstride = rows;
    p += static_cast<size_t>(nct) * per;
    int32_t* t_kr = reinterpret_cast<int32_t*>(p);
    for (int e = tid; e < nct * 2; e += NT) t_kr[e] = D0.krange[e];
    D0.krange = t_kr;
    p += nct + 1;
  }
  {
    const int nbf = cl_mh_bfac_rows(D0.f.nbeam);
    double* t_bp0 = p;
    double* t_obs = p + NBT;
    double* t_xf = p + 2 * NBT;
    double* t_bf = p + 3 * NBT;
    uint8_t* t_hx = reinterpret_cast<uint8_t*>(p + (3 + nbf) * NBT);
    for (int e = tid; e < NBT; e += NT) {                  // (bands beyond nbp: benign values, masked in the rows)
      const bool in = e < nbp;
      t_bp0[e] = in ? D0.bp0[e] : 0.0;
      t_obs[e] = in ? D0.f.obs[e] : 0.0;
      t_xf[e] = in ? D0.f.xf[e] : 1.0;
      t_hx[e] = in ? D0.f.hasx[e] : 0;
    }
    // beam factors: all 11 rows, or the centre row alone placed where row kClHalfSteps is expected
    for (int e = tid; e < nbf * NBT; e += NT) {
      const int rr = e / NBT, k = e - rr * NBT;
      t_bf[e] = k < nbp ? D0.f.bfac[(nbf == kClGrid ? rr : kClHalfSteps) * nbp + k] : 1.0;
    }
    D0.f.bfac = (nbf == kClGrid) ? t_bf : t_bf - static_cast<ptrdiff_t>(kClHalfSteps) * NBT;
    D0.bp0 = t_bp0;
    D0.f.obs = t_obs;
    D0.f.xf = t_xf;
    D0.f.hasx = t_hx;
  }
  __syncthreads();                                         // the only block barrier of the launch
  if (nbase >= N) return;

  // ---- this wave's LDS ----
  double* wbase = smem + cl_mhw_shared_doubles(P, NCT, nct, D0.kmax, D0.f.nbeam, RES ? 1 : 0) +
                  static_cast<size_t>(wv) * cl_mhw_wave_doubles(P, g.nblock, g.npn, a.sb, g.comom != nullptr, GRID);
  double* zc = wbase;                                      // [sb * npn][CB] normals of the cached steps
  double* uc = zc + static_cast<size_t>(a.sb) * g.npn * CB;   // [sb][CB] uniforms
  double* thn = uc + static_cast<size_t>(a.sb) * CB;       // [P][CB] the proposal ...
  double* z = thn;                                         // ... and, before it, the step's injected normals
  uint32_t* accb = reinterpret_cast<uint32_t*>(thn + static_cast<size_t>(P) * CB);   // [nblk][CB]
  double* dl = thn + static_cast<size_t>(P) * CB + (static_cast<size_t>(nblk) * CB + 1) / 2;   // [P][CB] x - mean (co-moments)
  double* cbeam = dl + (g.comom != nullptr ? static_cast<size_t>(P) * CB : 0) + cl;   // [11][CB], this chain's column (GRID)
  // GRID: the chi-squares of the step's grid points, [2][ncal nbeam][CB] per 16 chains, behind the waves' regions
  const int npts = D0.f.ncal * D0.f.nbeam;
  double* xchg = smem + cl_mhw_shared_doubles(P, NCT, nct, D0.kmax, D0.f.nbeam, RES ? 1 : 0) +
                 static_cast<size_t>(NT >> 6) * cl_mhw_wave_doubles(P, g.nblock, g.npn, a.sb, g.comom != nullptr, GRID) +
                 static_cast<size_t>(tile) * cl_mhw_xchg_doubles(npts) + cl;
  const bool writer = rid == 0;                            // the replica that stores the chains' state

  // ---- chain state: the lane's parameters 4 m + g, the chain's scalars replicated over its 4 lanes ----
  double rth[MS], rsg[MS], rmu[MS], rm2[MS], rnew[MS];
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    const int i = G * m + grp;
    const bool on = valid && i < P;
    rth[m] = on ? g.theta[i * N + n] : 0.0;
    rsg[m] = on ? g.sigma[i * N + n] : 0.0;
    rmu[m] = on ? g.mean[i * N + n] : 0.0;
    rm2[m] = on ? g.m2[i * N + n] : 0.0;
    rnew[m] = rth[m];
  }
  for (int b = grp; b < nblk; b += G) accb[b * CB + cl] = (valid && g.nblock) ? g.accept[static_cast<int64_t>(b) * N + n] : 0u;
  double lnpr = valid ? g.lnPr[n] : 0.0;
  uint32_t nacc = valid ? g.naccept[n] : 0u;
  cl_mhw_sync();

  const bool inject = g.inj_n != nullptr;
  const double* e2t = kExp2Table;                          // accept test: as k_gmh_accept
  const auto Kc = as_ptr<3>(D0.K);
  const auto kr_t = as_ptr<3>(D0.krange);
  // the lane's bands 4 s + g: its own base into every padded table, s an immediate offset
  const auto bp0_l = as_ptr<3>(D0.bp0) + grp;
  const auto obs_l = as_ptr<3>(D0.f.obs) + grp;
  const auto xf_l = as_ptr<3>(D0.f.xf) + grp;
  const auto hx_l = as_ptr<3>(D0.f.hasx) + grp;
  const auto bfac_l = as_ptr<3>(D0.f.bfac) + grp;
  // N^-1: resident, [NBT][NST] in LDS - fragment (k-step s, tile ct) of the lane at the immediate
  // offset (4 s NST + 16 ct) from its base; streamed, [.][nstride] in HBM / L2 - the lane's 32-bit byte
  // offset beside a scalar base per fragment
  const int nstride = RES ? NST : D0.f.nstride;
  const auto nm_l = as_ptr<MAS>(SYM ? D0.f.nsym : D0.f.ninv) + (RES ? grp * NST + cl : 0);
  const uint32_t nm_off = static_cast<uint32_t>(grp * nstride + cl) * 8u;
  const int kmax = D0.kmax;
  const int ncal = D0.f.ncal, nbeam = D0.f.nbeam;
  const bool has_x = D0.f.has_x != 0;
  int ks0[NCT];                                            // first k-step of every tile's bin range
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) ks0[ct] = ct < nct ? __builtin_amdgcn_readfirstlane(kr_t[2 * ct]) : 0;
  int blk = g.blk;
  int ts = 0;                                              // t % sb
  int thin_ctr = 0;
  int64_t row_next = 0;

#ifdef MI355_CLMH_PROF
  long long wt[6] = {0, 0, 0, 0, 0, 0};
  long long wc = clock64();
#define MHW_TICK(k) do { const long long now_ = clock64(); wt[k] += now_ - wc; wc = now_; } while (0)
#else
#define MHW_TICK(k) do {} while (0)
#endif
  for (int64_t t = 0; t < a.nsteps; ++t) {
    MHW_TICK(5);
    // ---- getNewParams (Proposal.py:127-164) + prior (binnedClModel.py:125-132) into rnew ----
    const int64_t step = g.step + t;
    const uint64_t base = static_cast<uint64_t>(step) * g.npn;
    const double* inj = inject ? g.inj_n + (t * N + n) * P : nullptr;
    if (!inject && ts == 0) {
      // refill the draw cache: normal(idx) for idx in [step npn, (step + sb) npn), uniform(step .. step + sb - 1)
      const uint64_t idx0 = base, cnt = static_cast<uint64_t>(a.sb) * g.npn;
      const uint64_t p0 = idx0 >> 1, np = ((idx0 + cnt - 1) >> 1) - p0 + 1;
      const uint64_t s0 = static_cast<uint64_t>(step), q0 = s0 >> 1, nq = ((s0 + a.sb - 1) >> 1) - q0 + 1;
      for (uint64_t it = lane; it < (np + nq) * CB; it += 64) {
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
      cl_mhw_sync();
    }
    const double* zrow = zc + static_cast<size_t>(ts) * g.npn * CB + cl;   // normal(base + r) = zrow[r * CB]
    bool neg = false;                                      // a negative amplitude of a spectrum that forbids it
    if (g.mode == GMH_UNCORR) {
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        const int i = G * m + grp;
        double v = rth[m];
        if (i < P) {
          if (g.nblock == 0 || block_of[i] == blk) {       // :151, :156-159
            const double zi = valid ? (inject ? ldas<1>(inj + i) : zrow[rank_of[i] * CB]) : 0.0;
            v += zi * rsg[m];
          }
          if (cltype[i] != 1 && v < 0) neg = true;
        }
        rnew[m] = v;
      }
    } else {
      if (inject) {
        if (valid) {
          for (int j = grp; j < P; j += G)
            if (g.mode == GMH_CORR) z[j * CB + cl] = ldas<1>(inj + j);
            else if (g.nblock == 0 || block_of[j] == blk) z[j * CB + cl] = ldas<1>(inj + rank_of[j]);
        }
        cl_mhw_sync();
      }
      auto zval = [&](int j) {
        return inject ? z[j * CB + cl] : (g.mode == GMH_CORR ? zrow[j * CB] : zrow[rank_of[j] * CB]);
      };
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        const int i = G * m + grp;
        double v = rth[m];
        if (i < P) {
          if (valid) {
            const double* Li = g.L + static_cast<int64_t>(i) * P * N + n;
            if (g.mode == GMH_CORR) {
              if (g.nblock == 0 || block_of[i] == blk) {   // :153-159
                double s = 0.0;
#pragma unroll 4
                for (int j = 0; j < P; ++j) s += ldas<1>(Li + j * N) * zval(j);
                v += s * rsg[m];
              }
            } else {                                       // :145-148, sum() starts from 0
              double s = 0.0;
              for (int j = 0; j < P; ++j)
                if (g.nblock == 0 || block_of[j] == blk) s += ldas<1>(Li + j * N) * zval(j);
              v += s * rsg[m];
            }
          }
          if (cltype[i] != 1 && v < 0) neg = true;
        }
        rnew[m] = v;
      }
      if (inject) cl_mhw_sync();                           // z is rewritten by the next step
    }
    const bool pbad = ((__builtin_amdgcn_ballot_w64(neg) >> cl) & 0x0001000100010001ull) != 0;

    MHW_TICK(0);
    // ---- band powers, transposed: BP[16 ct + g + 4 i][c] = BP0 + sum_b K[b][band] q[b][c] ----
    // (the proposal goes through LDS: a tile's k-steps start at the run-time bin 4 ks0(ct); kmax k-steps
    // for every tile, rows beyond a tile's range or bin P - 1 are zeros of the compact K, as cl_bp_tile)
#pragma unroll
    for (int m = 0; m < MS; ++m)
      if (G * m + grp < P) thn[(G * m + grp) * CB + cl] = rnew[m];
    cl_mhw_sync();
    // (k-step by k-step over all the tiles at once: NCT independent accumulators, the band-power
    // registers themselves, instead of one tile's dependent chain after the other)
    double bpv[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) bpv[s] = bp0_l[4 * s];    // (the padded table: zeros beyond band nb - 1)
    {
      const auto qp = as_ptr<3>(thn) + cl;
      const auto kp0 = Kc + grp * 16 + cl;
#pragma unroll 2
      for (int u = 0; u < kmax; ++u) {
        double kf[NCT], qf[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int ctc = ct < nct ? ct : 0;               // (tiles beyond the data set's: tile 0 again, dropped)
          const int bq = 4 * (ks0[ct] + u) + grp;
          kf[ct] = kp0[(static_cast<size_t>(ctc) * (4 * kmax) + 4 * u) * 16];
          qf[ct] = qp[(bq < P ? bq : P - 1) * CB];
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          double4_t acc = double4_t{bpv[4 * ct], bpv[4 * ct + 1], bpv[4 * ct + 2], bpv[4 * ct + 3]};
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kf[ct], qf[ct], acc, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) bpv[4 * ct + i] = acc[i];
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        bpv[4 * ct + i] = (ct < nct && ct * 16 + grp + 4 * i < nb) ? bpv[4 * ct + i] : 0.0;

    MHW_TICK(1);
    // ---- chi-square of every marginalisation point (ClData_CosmoMC.py:317-372) ----
    // the row's delta (cl_form_tile's row staging, :333-338, :383-392), band 4 s + g of chain c, four
    // logarithms at a time (the scheduling barriers keep the compiler from interleaving all NK chains,
    // which the registers do not hold); then Y^T = N^-T delta^T tile by tile: the N^-1 fragments of
    // four k-steps fetched while the MFMAs of the four before them run; row sums per virtual wave, then
    // over the virtual waves
    auto point_chisq = [&](int ib, int ic, double (&dv)[NK]) -> double {   // GRID: from bpv; else in place
      double calib = 1.0;
      auto bf = bfac_l + static_cast<size_t>(kClHalfSteps) * NBT;
      if (GRID && ncal > 1) calib = 1.0 + D0.f.calib * (ic - kClHalfSteps) * 3 / kClHalfSteps;   // :337
      if (GRID && nbeam > 1) bf = bfac_l + static_cast<size_t>(ib) * NBT;                         // :334
#pragma unroll
      for (int s0 = 0; s0 < NK; s0 += 4) {
#pragma unroll
        for (int s = s0; s < s0 + 4; ++s) {
          double b = (GRID ? bpv[s] : dv[s]) * bf[4 * s];
          b = calib * b;
          double zth = b + xf_l[4 * s];
          zth = (zth <= 0.0) ? 0.0 : zth;
          const double lg = cl_log<true>(zth);
          const bool lognormal = (hx_l[4 * s] != 0) & has_x;  // (no short circuit: straight-line code)
          double x = obs_l[4 * s] - (lognormal ? lg : b);
          asm("" : "+v"(x));                               // (opaque: the compiler would branch around the row otherwise)
          dv[s] = (valid & (4 * s + grp < nb)) ? x : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      MHW_TICK(2);
      // the tiles in cl_form_tile's order: virtual wave by virtual wave, pass by pass - two virtual
      // waves at a time, their two MFMA chains interleaved (a chain alone waits out the dependent
      // latency of every MFMA).  Rolled loops (one body; nothing of one tile for the compiler to hoist
      // out of the step loop and spill).
      double total = 0.0;
#pragma nounroll
      for (int w = 0; w < NWREF; w += 2) {
        double part[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma nounroll
        for (int ps = 0; ps < NPASS; ++ps) {
          int ct[2];
          bool on[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            ct[h] = cl_mhw_tile<NWREF, NCT, SYM>(w + h, ps);
            on[h] = ct[h] >= 0 && ct[h] < nct;             // uniform
            if (!on[h]) ct[h] = 0;                         // (computed on tile 0, dropped)
          }
          if (!on[0] && !on[1]) continue;
          double4_t acc[2] = {double4_t{0.0, 0.0, 0.0, 0.0}, double4_t{0.0, 0.0, 0.0, 0.0}};
          // k-steps: 4 (ct + 1) of the symmetrised matrix; else all NK - those beyond the padded band
          // count meet delta = 0 (and zero rows of the resident copy; the streamed copy's zero slack row)
          const int nks0 = SYM ? 4 * (ct[0] + 1) : NK, nks1 = SYM ? 4 * (ct[1] + 1) : NK;   // multiples of 4
          const int nksm = nks0 > nks1 ? nks0 : nks1;
          auto frag = [&](int h, int ks) -> double {
            if constexpr (RES) {
              return nm_l[ks * (4 * NST) + ct[h] * 16];
            } else {
              const int r0 = 4 * ks;                       // rows 4 ks + g; beyond the matrix: its first slack row
              const bool in = r0 + 3 < nbp;                // (nbp is a multiple of 16: a k-step is inside or outside)
              const char* base = reinterpret_cast<const char*>(SYM ? D0.f.nsym : D0.f.ninv) +
                                 (static_cast<size_t>(in ? r0 : nbp) * nstride + ct[h] * 16) * 8;
              const uint32_t off = in ? nm_off : static_cast<uint32_t>(cl) * 8u;
              return *(const __attribute__((address_space(1))) double*)((const __attribute__((address_space(1))) char*)base + off);
            }
          };
          double cur[2][4], nxt[2][4];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[h][u] = frag(h, u);
#pragma unroll
          for (int s0 = 0; s0 < NK; s0 += 4) {
            if (s0 < nksm) {                               // uniform (a compile-time truth without SYM)
              if (s0 + 4 < nksm) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                  for (int u = 0; u < 4; ++u) nxt[h][u] = frag(h, s0 + 4 + u);
              }
              __builtin_amdgcn_sched_barrier(0);
              // (a symmetrised tile's k-steps beyond its own 4 (ct + 1) run on zeroed operands only when
              // its partner is longer: SYM tiles take the branch below per half)
              if (!SYM || s0 < nks0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0][u], dv[s0 + u], acc[0], 0, 0, 0);
              }
              if (!SYM || s0 < nks1) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[1][u], dv[s0 + u], acc[1], 0, 0, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[h][u] = nxt[h][u];
            }
          }
          // delta of a tile's own bands (registers 4 ct .. 4 ct + 3, ct uniform: a select chain)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double dct[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              dct[i] = dv[i];
#pragma unroll
              for (int c2 = 1; c2 < NCT; ++c2) dct[i] = (ct[h] == c2) ? dv[4 * c2 + i] : dct[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (on[h] && ct[h] * 16 + grp + 4 * i < nb) part[h][i] = fma(acc[h][i], dct[i], part[h][i]);
          }
        }
        // cl_form_tile's butterfly over the 16 columns lc = g + 4 i: xor 1, 2 across lanes, xor 4, 8 in registers
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double b4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            b4[i] = xor32_sum(xor16_sum(part[h][i]));
          }
          const double tw = (b4[0] + b4[1]) + (b4[2] + b4[3]);
          total = (w + h == 0) ? tw : total + tw;
        }
      }
      MHW_TICK(3);
      return total;
    };
    double chisq = 0.0;
    if constexpr (!GRID) {
      chisq = point_chisq(0, 0, bpv);                      // delta in place of the band powers
    } else {
      // the replicas share out the points (p = ib ncal + ic: replica r takes p = r, r + rep, ...); every
      // chi-square goes to the step's exchange buffer (the chain's four lanes hold the same sums and
      // write the same words), then every replica marginalises over all of them (k_cl_combine, :340-372)
      double* xb = xchg + static_cast<size_t>(t & 1) * npts * CB;
      for (int pt = rid; pt < npts; pt += rep) {
        const int ib = pt / ncal, ic = pt - ib * ncal;
        double dv[NK];
        xb[pt * CB] = point_chisq(ib, ic, dv);
      }
      if (rep > 1) __syncthreads();                        // uniform over the block
      else cl_mhw_sync();
      if (ncal > 1 && nbeam > 1) {
        for (int ib = 0; ib < nbeam; ++ib) cbeam[ib * CB] = cl_marge11<3>(xb + ib * ncal * CB, CB, a.weights, a.margenorm);
        cl_mhw_sync();
        chisq = cl_marge11<3>(cbeam, CB, a.weights, a.margenorm);
        cl_mhw_sync();
      } else {
        chisq = cl_marge11<3>(xb, CB, a.weights, a.margenorm);   // one row or one column of the grid
      }
    }
    MHW_TICK(4);
    const double lnl = 0.0 + (-(chisq / 2));               // :315, as cl_combine_chain

    // ---- accept / reject (MCMC.py:236-279), the same decision in the chain's four lanes ----
    bool keep = false;
    if (a.thin > 0 && ++thin_ctr == a.thin) {
      thin_ctr = 0;
      keep = row_next < a.nkeep;
    }
    const int64_t row = row_next;
    if (keep) ++row_next;
    bool accepted = false;
    if (valid) {
      const double prev = lnpr;
      const bool bad = (lnl != lnl) || (lnl == -__builtin_inf());   // :238-243
      if (!pbad && !bad) {
        const double lnalpha = lnl - prev;                 // :251-252 (prior is 1)
        bool reject = false;
        if (!(lnalpha >= 0.0)) {
          const double u = g.inj_u ? ldas<1>(g.inj_u + t * N + n) : uc[ts * CB + cl];   // = uniform(step)
          reject = !(exp_accept(lnalpha, e2t) > u);        // :264
        }
        accepted = !reject;
      }
      if (accepted) {
        lnpr = lnl;
        nacc += 1;                                         // :269
        if (g.nblock && grp == 0) accb[blk * CB + cl] += 1;   // :271
      }
      if (keep && grp == 0 && writer) {
        stas<1>(g.tr_lnPr + row * N + n, accepted ? lnl : prev);
        stas<1>(g.tr_acc + row * N + n, static_cast<uint8_t>(accepted ? 1 : 0));
      }
    }

    // ---- position, running moments (MCMC.py:290-340), trace ----
    const double cnt = static_cast<double>(g.nsamp + t + 1);
    const double rn = rcp_nr(cnt);
    const double cf = 1.0 - rn;
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const int i = G * m + grp;
      if (i < P) {
        const double x = accepted ? rnew[m] : rth[m];
        rth[m] = x;
        const double dlt = x - rmu[m];
        if (g.comom != nullptr) dl[i * CB + cl] = dlt;
        rmu[m] = fma(dlt, rn, rmu[m]);
        rm2[m] = fma(dlt, x - rmu[m], rm2[m]);
        if (keep && valid && writer) stas<1>(g.tr_theta + (row * P + i) * N + n, x);
      }
    }
    if (g.comom != nullptr) {                              // uniform over the grid
      cl_mhw_sync();
      // C_ij += (n-1)/n d_i d_j (k_gmh_accept), the columns dealt round-robin to the chain's lanes and - every
      // replica holds the same x - mean - to the replicas
      if (valid) gmh_comom_update(g.comom, N, n, P, cf, dl + cl, CB, grp + G * rid, G * rep);
      cl_mhw_sync();                                       // dl is rewritten by the next step
    }
    if (g.nblock) blk = (blk + 1 == g.nblock) ? 0 : blk + 1;   // MCMC.py:278-279
    ts = (ts + 1 == a.sb) ? 0 : ts + 1;
  }

#ifdef MI355_CLMH_PROF
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
    printf("clmhw block %d steps %lld (ticks per step): propose %lld bp %lld rows(ln) %lld form %lld marge %lld accept+moments %lld\n",
           blockIdx.x, (long long)a.nsteps, wt[0] / a.nsteps, wt[1] / a.nsteps, wt[2] / a.nsteps, wt[3] / a.nsteps,
           wt[4] / a.nsteps, wt[5] / a.nsteps);
#endif
  // ---- state back to HBM ----
  cl_mhw_sync();
  if (valid && writer) {
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const int i = G * m + grp;
      if (i < P) {
        g.theta[i * N + n] = rth[m];
        g.mean[i * N + n] = rmu[m];
        g.m2[i * N + n] = rm2[m];
      }
    }
    if (g.nblock)
      for (int b = grp; b < nblk; b += G) g.accept[static_cast<int64_t>(b) * N + n] = accb[b * CB + cl];
    if (grp == 0) {
      g.lnPr[n] = lnpr;
      g.naccept[n] = nacc;
    }
  }
}

}  // namespace mi355
