// Fused Metropolis-Hastings sampler of the beam fit: every MH step of MCMC.py:225-286 on the chains of ONE map
// inside one launch - the reference's own use of BeamFit (ProcessBeamData.py:277-279 -> MCMC.sampler: one map,
// one chain, 1e5 steps), which the launch-per-step engine runs at 11 us per step because a step is two kernel
// launches and a dozen round trips to device memory, whatever the map's size.
//
// One block per map, 256 NSUB threads.  The chains' state (up to kBeamCH = 8 chains of the map: parameters,
// proposal widths, moments, sqrt matrices, co-moments) lives in LDS for the launch, and so do the map's pixels
// when they fit.  A step is the step-wise engine's, operation for operation (gmh_propose_block, k_beam_chi2 +
// the finish in gmh_accept_block): the same Philox draws, the same package / prior, the same constants, the
// same pixel chunks (beam_px_plan) with the same sums inside a chunk (beam_chunk_sums, beam_chunk_wave_sums)
// and over the chunks in ascending order, the same accept test and moments - the chains are the engine's bit
// for bit (tests/test_beam_mh_fused_gpu.py) and the reference's own draw for draw (tests/test_gmh_gpu.py).
// The NSUB groups of 256 threads take the chunks q, q + NSUB, ... of a step side by side.
#pragma once
#include "beam_kernels.hpp"
#include "gmh_device.hpp"

namespace mi355 {

constexpr int kBeamMhMaxSub = 4;
constexpr int kBeamMhU = 2;                 // chunks a group of 256 threads takes between two barriers
constexpr int kBeamTri = kBeamP * (kBeamP + 1) / 2;

struct BeamMhArgs {
  GmhArgs g;                 // chain state and proposal; step / nsamp / blk of the FIRST step; inj_n / inj_u /
                             // tr_* are the bases of the whole run
  BeamArgs b;                // the maps and the pixel chunks of a step (theta / partial unused)
  int64_t nsteps;
  int thin;
  int64_t nkeep;
  int px_lds;                // 1: the block copies its map's pixels into LDS at launch
};

// dynamic LDS doubles: partial sums of a step's chunks, sqrt matrices, co-moments, (pixels)
__host__ __device__ inline size_t beam_mh_lds_doubles(int nchunk, int64_t npix, int w_kind, int xy_per_map, int px_lds) {
  size_t n = static_cast<size_t>(nchunk) * kBeamCH + static_cast<size_t>(kBeamCH) * kBeamP * kBeamP +
             static_cast<size_t>(kBeamCH) * kBeamTri;
  if (px_lds) n += static_cast<size_t>(npix) * (3 + (w_kind == 2 ? 1 : 0));
  return n;
}

struct BeamMhLds {
  double kc[kBeamCH][kBeamNC];
  double fc[kBeamCH][kBeamNF];
  double red[2][kBeamMhMaxSub][kBeamMhU][4][kBeamCH];
  double th[kBeamCH][kBeamP], thn[kBeamCH][kBeamP], sg[kBeamCH][kBeamP], mu[kBeamCH][kBeamP], m2[kBeamCH][kBeamP];
  double z[kBeamCH][kBeamP], dl[kBeamCH][kBeamP];
  double lnpr[kBeamCH], prior[kBeamCH], ref[4];
  double trig[kBeamCH][6];   // the angles of a chain's last proposal and their cos / sin (beam_chain_consts)
  int dec[kBeamCH];
  uint32_t nacc[kBeamCH];
  uint32_t accb[kBeamP][kBeamCH];
  int32_t block_of[kBeamP], rank_of[kBeamP];
};

// The kernel is compiled per class of chain count (NCHT = 1, 4, 8: the pixel loop's instantiations for one, up
// to four, up to eight chains), so that the job of one chain does not carry the registers of the loop for eight.
// The constants (two sincos, the model's divisions and roots) are a function of their own.
template <int NCHT, int AS>
__device__ __forceinline__ void beam_mh_chunk(const BeamPixels& px, int64_t p0, int64_t p1, int nch, int vt,
                                              const double (*kc)[kBeamNC], const double (*fc)[kBeamNF],
                                              const double* e2t, int lane, double* red) {
  double acc[kBeamCH];
  beam_chunk_sums<NCHT, true, AS>(px, p0, p1, nch, vt, kc, fc, e2t, acc);
  beam_chunk_wave_sums(acc, nch, lane, red);
}
// (every pointer is to LDS: an argument on the caller's stack would be a round trip to scratch memory per step)
__device__ __noinline__ double beam_mh_consts(double* thn, bool active, const double* ref, BeamPriorBox box,
                                              double (*kc)[kBeamNC], double (*fc)[kBeamNF], int slot, double* trig) {
  double th9[kBeamP];
  double prior = 0.0;
  if (active) {
#pragma unroll
    for (int i = 0; i < kBeamP; ++i) th9[i] = thn[i];
    beam_package(th9);
    thn[4] = th9[4];
    thn[7] = th9[7];
    thn[8] = th9[8];
    prior = beam_prior(th9, box);
  }
  const double rf[4] = {ref[0], ref[1], ref[2], ref[3]};
  beam_chain_consts(th9, active, rf, kc, fc, slot, trig);
  return prior;
}

template <int NSUB, int NCHT>
__global__ __launch_bounds__(256 * NSUB) void k_beam_mh(BeamMhArgs a) {
  static_assert(NSUB >= 1 && NSUB <= kBeamMhMaxSub, "groups of 256 threads per block");
  __shared__ BeamMhLds S;
  extern __shared__ double beam_mh_dyn[];
  const GmhArgs& g = a.g;
  const BeamArgs& b = a.b;
  constexpr int P = kBeamP;
  const int tid = threadIdx.x, q = tid >> 8, vt = tid & 255, lane = tid & 63, wv = vt >> 6;
  const int64_t N = g.N;
  const int C = b.C;                                       // chains of the map (<= kBeamCH: host check)
  const int64_t ml = blockIdx.x, m = b.m0 + ml;
  const int64_t nb0 = ml * C;                              // first chain of the map
  const double* e2t = stage_exp2_table();                  // valid after the first barrier
  const double* bmt = bm_of(e2t);

  double* part = beam_mh_dyn;                              // [nchunk][kBeamCH]
  double* Lm = part + static_cast<size_t>(b.nchunk) * kBeamCH;   // [kBeamCH][P][P]
  double* cm = Lm + kBeamCH * P * P;                       // [kBeamCH][kBeamTri]
  double* pxl = cm + kBeamCH * kBeamTri;

  // thread (c, i): chain c of the map, parameter i; thread c (< C): chain c; thread (c2, e): entry e of
  // chain c2's co-moment triangle
  const int cc = tid >> 4, ci = tid & 15;
  const bool cp = tid < 16 * kBeamCH && cc < C && ci < P;
  const int64_t ncp = nb0 + cc;
  const bool ch = tid < C;
  const int64_t nch_ = nb0 + tid;
  const int c2 = tid >> 6, e2 = tid & 63;
  const bool ce = g.comom != nullptr && tid < 64 * kBeamCH && c2 < C && e2 < kBeamTri;
  int ei = 0, ej = 0;
  if (ce) gmh_tri_decode(e2, ei, ej);

  if (cp) {
    S.th[cc][ci] = g.theta[ci * N + ncp];
    S.sg[cc][ci] = g.sigma[ci * N + ncp];
    S.mu[cc][ci] = g.mean[ci * N + ncp];
    S.m2[cc][ci] = g.m2[ci * N + ncp];
    if (g.mode != GMH_UNCORR)
      for (int j = 0; j < P; ++j) Lm[(cc * P + ci) * P + j] = g.L[(static_cast<int64_t>(ci) * P + j) * N + ncp];
  }
  if (ch) {
    S.lnpr[tid] = g.lnPr[nch_];
    S.nacc[tid] = g.naccept[nch_];
    for (int k = 0; k < g.nblock; ++k) S.accb[k][tid] = g.accept[static_cast<int64_t>(k) * N + nch_];
  }
  if (ce) cm[c2 * kBeamTri + e2] = g.comom[(static_cast<int64_t>(ei) * P + ej) * N + nb0 + c2];
  if (tid < P) {
    S.block_of[tid] = g.nblock ? g.block_of[tid] : 0;
    S.rank_of[tid] = g.rank_in_block[tid];
  }
  BeamPixels px;
  px.xs = b.x + (b.xy_per_map ? m * b.npix : 0);
  px.ys = b.y + (b.xy_per_map ? m * b.npix : 0);
  px.ds = b.d + m * b.npix;
  px.ws = b.w + (b.w_kind == 2 ? m * b.npix : (b.w_kind == 1 ? m : 0));
  px.w_kind = b.w_kind;
  px.w_shared = (b.w_kind == 2) ? 0.0 : px.ws[0];
  const double* ref = b.xyref + (b.xy_per_map ? m : 0) * 4;
  px.xr = ref[0];
  px.yr = ref[1];
  if (a.px_lds) {
    double* lx = pxl;
    double* ly = lx + b.npix;
    double* ld = ly + b.npix;
    double* lw = ld + b.npix;
    for (int64_t p = tid; p < b.npix; p += 256 * NSUB) {
      lx[p] = px.xs[p];
      ly[p] = px.ys[p];
      ld[p] = px.ds[p];
      if (b.w_kind == 2) lw[p] = px.ws[p];
    }
    px.xs = lx;
    px.ys = ly;
    px.ds = ld;
    if (b.w_kind == 2) px.ws = lw;
  }
  if (tid < 4) S.ref[tid] = ref[tid];
  if (tid < kBeamCH * 6) S.trig[tid / 6][tid % 6] = __builtin_nan("");   // (a NaN angle matches nothing: first call computes)
  __syncthreads();

  const bool inject = g.inj_n != nullptr;
  const int rounds = (b.nchunk + NSUB * kBeamMhU - 1) / (NSUB * kBeamMhU);
  int blk = g.blk;
  int thin_ctr = 0;
  int64_t row_next = 0;

#ifdef MI355_BEAM_MH_PROF
  long long pt[6] = {0, 0, 0, 0, 0, 0};
  long long pc = clock64();
#define BMH_TICK(k) do { const long long now_ = clock64(); pt[k] += now_ - pc; pc = now_; } while (0)
#else
#define BMH_TICK(k) do {} while (0)
#endif
  for (int64_t t = 0; t < a.nsteps; ++t) {
    BMH_TICK(5);
    const int64_t step = g.step + t;
    const uint64_t base = static_cast<uint64_t>(step) * g.npn;
    bool keep = false;
    if (a.thin > 0 && ++thin_ctr == a.thin) {
      thin_ctr = 0;
      keep = row_next < a.nkeep;
    }
    const int64_t row = row_next;
    if (keep) ++row_next;

    // ---- getNewParams (Proposal.py:127-164), as gmh_propose_block ----
    if (g.mode == GMH_UNCORR) {
      if (cp) {
        const bool in = g.nblock == 0 || S.block_of[ci] == blk;                       // :151
        double v = S.th[cc][ci];
        if (in) {
          const double z = inject ? g.inj_n[(t * N + ncp) * P + ci]
                                  : philox_normal(g.seed, RNG_NORMAL, ncp, base + S.rank_of[ci], bmt);
          v += z * S.sg[cc][ci];                                                       // :156-159
        }
        S.thn[cc][ci] = v;
      }
    } else {
      if (cp) {
        if (g.mode == GMH_CORR)                                                        // z_j = normal(base + j), all j
          S.z[cc][ci] = inject ? g.inj_n[(t * N + ncp) * P + ci] : philox_normal(g.seed, RNG_NORMAL, ncp, base + ci, bmt);
        else if (g.nblock == 0 || S.block_of[ci] == blk)                               // rotated: the block's normals by rank
          S.z[cc][ci] = inject ? g.inj_n[(t * N + ncp) * P + S.rank_of[ci]]
                               : philox_normal(g.seed, RNG_NORMAL, ncp, base + S.rank_of[ci], bmt);
      }
      __syncthreads();
      if (cp) {
        double v = S.th[cc][ci];
        const double* Li = Lm + (cc * P + ci) * P;
        if (g.mode == GMH_CORR) {
          if (g.nblock == 0 || S.block_of[ci] == blk) {                                // :153-159
            double s = 0.0;
            for (int j = 0; j < P; ++j) s += Li[j] * S.z[cc][j];
            v += s * S.sg[cc][ci];
          }
        } else {                                                                       // :145-148, sum() starts from 0
          double s = 0.0;
          for (int j = 0; j < P; ++j)
            if (g.nblock == 0 || S.block_of[j] == blk) s += Li[j] * S.z[cc][j];
          v += s * S.sg[cc][ci];
        }
        S.thn[cc][ci] = v;
      }
    }
    __syncthreads();
    BMH_TICK(0);

    // ---- package + prior (gmh_prior_package), the chain's constants (beam_group_consts) ----
    if (tid < kBeamCH) {
      const double prior = beam_mh_consts(S.thn[tid], ch, S.ref, g.pb.box, S.kc, S.fc, tid, S.trig[tid]);
      if (ch) S.prior[tid] = prior;
    }
    __syncthreads();
    BMH_TICK(1);

    // ---- chi-square: the step-wise engine's chunks, NSUB at a time ----
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
      for (int u = 0; u < kBeamMhU; ++u) {
        const int bx = (r * kBeamMhU + u) * NSUB + q;
        if (bx < b.nchunk) {
          const int64_t p0 = static_cast<int64_t>(bx) * b.chunk;
          const int64_t p1 = min(p0 + b.chunk, b.npix);
          if (a.px_lds) beam_mh_chunk<NCHT, 3>(px, p0, p1, C, vt, S.kc, S.fc, e2t, lane, S.red[r & 1][q][u][wv]);
          else beam_mh_chunk<NCHT, 1>(px, p0, p1, C, vt, S.kc, S.fc, e2t, lane, S.red[r & 1][q][u][wv]);
        }
      }
      __syncthreads();
      if (vt < kBeamMhU * kBeamCH) {
        const int u = vt / kBeamCH, c = vt % kBeamCH;
        const int bx = (r * kBeamMhU + u) * NSUB + q;
        if (bx < b.nchunk && c < C) {
          const double (*rd)[kBeamCH] = S.red[r & 1][q][u];
          part[bx * kBeamCH + c] = ((rd[0][c] + rd[1][c]) + rd[2][c]) + rd[3][c];
        }
      }
    }
    __syncthreads();
    BMH_TICK(2);

    // ---- accept / reject (MCMC.py:236-279), as gmh_accept_block ----
    if (ch) {
      double s = 0.0;
      for (int j = 0; j < b.nchunk; ++j) s += part[j * kBeamCH + tid];
      double v = -0.5 * s;                                                             // NormalizedBeamLikelihood.py:16
      const double prior = S.prior[tid];
      const double prev = S.lnpr[tid];
      bool accepted = false;
      const bool bad = (v != v) || (v == -__builtin_inf());                            // :238-243
      if (prior > 0.0 && !bad) {
        if (prior != 1.0) v += log(prior);                                             // :250
        const double lnalpha = v - prev;                                               // :251-252
        bool reject = false;
        if (!(lnalpha >= 0.0)) {
          const double u = g.inj_u ? g.inj_u[t * N + nch_]
                                   : philox_uniform(g.seed, RNG_UNIFORM, nch_, static_cast<uint64_t>(step));
          reject = !(exp_accept(lnalpha, e2t) > u);                                    // :264
        }
        accepted = !reject;
      }
      if (accepted) {
        S.lnpr[tid] = v;
        S.nacc[tid] += 1;                                                              // :269
        if (g.nblock) S.accb[blk][tid] += 1;                                           // :271
      }
      if (keep) {
        g.tr_lnPr[row * N + nch_] = accepted ? v : prev;
        g.tr_acc[row * N + nch_] = accepted ? 1 : 0;
      }
      S.dec[tid] = accepted ? 1 : 0;
    }
    __syncthreads();
    BMH_TICK(3);

    // ---- position, running moments (MCMC.py:290-340), trace ----
    const double cnt = static_cast<double>(g.nsamp + t + 1);
    const double rn = rcp_nr(cnt);
    const double cf = 1.0 - rn;
    if (cp) {
      const double x = S.dec[cc] ? S.thn[cc][ci] : S.th[cc][ci];
      S.th[cc][ci] = x;
      const double mu = S.mu[cc][ci];
      const double dlt = x - mu;
      S.dl[cc][ci] = dlt;
      const double mu1 = fma(dlt, rn, mu);
      S.mu[cc][ci] = mu1;
      S.m2[cc][ci] = fma(dlt, x - mu1, S.m2[cc][ci]);
      if (keep) g.tr_theta[(row * P + ci) * N + ncp] = x;
    }
    if (g.comom != nullptr) {                                                          // uniform over the grid
      __syncthreads();
      if (ce) cm[c2 * kBeamTri + e2] = cm[c2 * kBeamTri + e2] + cf * S.dl[c2][ei] * S.dl[c2][ej];
    }
    if (g.nblock) blk = (blk + 1 == g.nblock) ? 0 : blk + 1;                           // MCMC.py:278-279
  }

#ifdef MI355_BEAM_MH_PROF
  if (tid == 0 && blockIdx.x == 0 && a.nsteps > 100)
    printf("beam mh: steps %lld, cycles per step: propose %lld consts %lld chi2 %lld accept %lld moments %lld\n",
           (long long)a.nsteps, pt[0] / a.nsteps, pt[1] / a.nsteps, pt[2] / a.nsteps, pt[3] / a.nsteps, pt[5] / a.nsteps);
#endif
  // ---- state back to device memory ----
  __syncthreads();
  if (cp) {
    g.theta[ci * N + ncp] = S.th[cc][ci];
    g.mean[ci * N + ncp] = S.mu[cc][ci];
    g.m2[ci * N + ncp] = S.m2[cc][ci];
  }
  if (ch) {
    g.lnPr[nch_] = S.lnpr[tid];
    g.naccept[nch_] = S.nacc[tid];
    for (int k = 0; k < g.nblock; ++k) g.accept[static_cast<int64_t>(k) * N + nch_] = S.accb[k][tid];
  }
  if (ce) g.comom[(static_cast<int64_t>(ei) * P + ej) * N + nb0 + c2] = cm[c2 * kBeamTri + e2];
}

}  // namespace mi355
