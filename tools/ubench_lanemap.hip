// Timing experiment: the two lane mappings of the fused SED
// Metropolis kernel, on the same simplified sampler (submmModel2_normalized, one parameter per step,
// Philox draws, prior, fast-path likelihood, accept, Welford moments):
//   G = 1     lane = chain; the lane loops over the nb bands (the product's k_mh_run)
//   G = 8/32  lane = (chain, band): G lanes per chain, each evaluates one band, chi-square reduced
//             over the G lanes - "tree": xor-shuffle tree (not the band order of the reference's sum);
//             "seq": every lane gathers the G terms and adds them in band order (bit-identical to G=1)
// Not part of the library.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -Iinclude \
//         tools/ubench_lanemap.hip -o /tmp/ubench_lanemap && /tmp/ubench_lanemap
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../mcmc_amd/csrc/common.hpp"
#include "../mcmc_amd/csrc/philox.hpp"
#include "../mcmc_amd/csrc/sed_device.hpp"

using namespace mi355;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

constexpr int P = 6;

struct Args {
  const double* bands;   // [S][nbm][kBandFields]
  int nbm, nb, C;
  int64_t N;             // chains
  double* theta;         // [P][N]
  double* sigma;         // [P][N]
  double* lnPr;          // [N]
  double* mean;          // [P][N]
  double* m2;            // [P][N]
  uint32_t* nacc;        // [N]
  int nsteps;
  uint64_t seed;
  SedConsts k;
};

// chi-square term of band j for the proposed point (the body of sed_lnlike_fast<M2N>'s loop)
__device__ __forceinline__ void band_term(const Bands& B, int j, double e1, double e2, double rT1,
                                          double rT2, double ae1, double ae2, double& r, double& rw) {
  const double hn = B.hnu2(j), lh = B.lhi(j);
  const double y1 = hn * rT1, y2 = hn * rT2;
  const double z[4] = {fma(e1, lh, -y1), -y1, fma(e2, lh, -y2), -y2};
  double pw[4];
  exp2_tab_n<4>(z, B.e2t, pw);
  const double n1 = pw[0] * ae1, d1 = 1.0 - pw[1];
  const double n2 = pw[2] * ae2, d2 = 1.0 - pw[3];
  r = B.d(j) - div_fast(fma(n1, d2, n2 * d1), d1 * d2);
  rw = r * B.w(j);
}

// G lanes per chain; SEQ: in-order sum (G > 1)
template <int G, bool SEQ>
__global__ __launch_bounds__(256) void k_mh(Args a) {
  extern __shared__ double smem[];
  const int stride = kBandFields * a.nbm;
  const int cpb = blockDim.x / G;                                  // chains per block
  const int64_t n0 = static_cast<int64_t>(blockIdx.x) * cpb;
  const int64_t n_last = min(n0 + cpb, a.N) - 1;
  const int64_t s_first = n0 / a.C;
  const int nsrc = static_cast<int>(n_last / a.C - s_first) + 1;
  const double* e2t = stage_exp2_table();
  {
    const double* src = a.bands + s_first * stride;
    for (int i = threadIdx.x; i < nsrc * stride; i += blockDim.x) smem[i] = src[i];
    __syncthreads();
  }
  const int64_t n = n0 + threadIdx.x / G;
  const int jb = threadIdx.x % G;
  const bool live = n < a.N;
  const int64_t nn = live ? n : n_last;                            // dead lanes shadow a live chain
  const int64_t s = nn / a.C;
  Bands B{smem + (s - s_first) * stride, a.nbm, a.nb, a.nb, e2t, true};

  double th[P], sg[P], mu[P], m2[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    th[i] = a.theta[i * a.N + nn];
    sg[i] = a.sigma[i * a.N + nn];
    mu[i] = 0.0;
    m2[i] = 0.0;
  }
  double lnPr = a.lnPr[nn];
  uint32_t nacc = 0;
  int blk = 0;
  double za = 0, zb = 0, ua = 0, ub = 0;
  for (int t = 0; t < a.nsteps; ++t) {
    if ((t & 1) == 0) {
      philox_normal2(a.seed, RNG_NORMAL, nn, t >> 1, za, zb);
      philox_uniform2(a.seed, RNG_UNIFORM, nn, t >> 1, ua, ub);
    }
    const double z = (t & 1) ? zb : za, u = (t & 1) ? ub : ua;
    double nw[P];
#pragma unroll
    for (int i = 0; i < P; ++i) nw[i] = th[i] + ((i == blk) ? z * sg[i] : 0.0);
    sed_package<MI355_SED_M2N>(nw);
    const double prior = sed_prior<MI355_SED_M2N>(nw, a.k);
    bool accepted = false;
    double next = lnPr;
    // G > 1: the lanes of a chain agree on prior (same arithmetic), so the shuffles below are
    // executed by all of them or none
    if (prior > 0.0) {
      double v;
      if constexpr (G == 1) {
        v = sed_lnlike_fast<MI355_SED_M2N>(nw, B, a.k, nullptr);
      } else {
        const double e1 = 3.0 + nw[1], e2 = 3.0 + nw[4];
        const double rT1 = rcp_nr2(nw[2]), rT2 = rcp_nr2(nw[5]);
        const double yb[2] = {a.k.hk_nub2 * rT1, a.k.hk_nub2 * rT2};
        double eb[2];
        exp2_tab_n<2>(yb, e2t, eb);
        const double ae1 = nw[0] * (eb[0] - 1.0), ae2 = nw[3] * (eb[1] - 1.0);
        double r = 0.0, rw = 0.0;
        if (jb < a.nb) band_term(B, jb, e1, e2, rT1, rT2, ae1, ae2, r, rw);
        double chi2;
        if constexpr (SEQ) {
          chi2 = 0.0;
          for (int j = 0; j < a.nb; ++j) chi2 = fma(__shfl(rw, j, G), __shfl(r, j, G), chi2);
        } else {
          chi2 = rw * r;
#pragma unroll
          for (int off = G / 2; off > 0; off >>= 1) chi2 += __shfl_xor(chi2, off, G);
        }
        v = -0.5 * chi2;
      }
      const bool bad = (v != v) || (v == -__builtin_inf());
      if (!bad) {
        const double lnalpha = v - lnPr;
        const bool reject = !(lnalpha >= 0.0) && !(exp_accept(lnalpha, e2t) > u);
        if (!reject) {
          accepted = true;
          next = v;
        }
      }
    }
    if (accepted) {
#pragma unroll
      for (int i = 0; i < P; ++i) th[i] = nw[i];
      lnPr = next;
      ++nacc;
    }
    blk = (blk + 1 == P) ? 0 : blk + 1;
    const double rn = rcp_nr(static_cast<double>(t + 1));
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const double dl = th[i] - mu[i];
      mu[i] = fma(dl, rn, mu[i]);
      m2[i] = fma(dl, th[i] - mu[i], m2[i]);
    }
  }
  if (live && jb == 0) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      a.theta[i * a.N + n] = th[i];
      a.mean[i * a.N + n] = mu[i];
      a.m2[i * a.N + n] = m2[i];
    }
    a.lnPr[n] = lnPr;
    a.nacc[n] = nacc;
  }
}

template <typename T>
static T* dev(size_t n) {
  T* p = nullptr;
  CK(hipMalloc(&p, std::max<size_t>(1, n) * sizeof(T)));
  return p;
}

struct Result {
  double us_per_step;
  std::vector<double> mean0;
  std::vector<uint32_t> nacc;
};

template <int G, bool SEQ>
static Result run(int64_t S, int C, int nb, int nsteps, int reps) {
  const int nbm = nb;
  const int64_t N = S * C;
  const double h_over_k = 0.04799237, nu_b = 1000.0;
  SedConsts k{h_over_k, nu_b, 1e-9, 3.0, 100.0, 0.0, 3.0, 1000.0, h_over_k * nu_b, 0, 0, 1.0, 0, 0};
  k.hk_nub2 = static_cast<double>(static_cast<long double>(k.hk_nub) * 1.442695040888963407359924681001892137L);
  // synthetic two-component sources (the shape of mcmc_amd/synth.py's catalogue, own generator)
  std::vector<double> hb(static_cast<size_t>(S) * nbm * kBandFields);
  srand48(7);
  auto grey = [&](double A, double b, double T, double nu) {
    return A * std::pow(nu / nu_b, 3.0 + b) * std::expm1(h_over_k * nu_b / T) / std::expm1(h_over_k * nu / T);
  };
  std::vector<double> th0(static_cast<size_t>(S) * P);
  for (int64_t s = 0; s < S; ++s) {
    double* t = &th0[s * P];
    t[0] = std::pow(10.0, drand48() - 0.5); t[1] = 1.0 + 1.5 * drand48(); t[2] = 15.0 + 15.0 * drand48();
    t[3] = (0.05 + 0.45 * drand48()) * t[0]; t[4] = 1.0 + 1.5 * drand48(); t[5] = 35.0 + 25.0 * drand48();
    const double zred = 0.3 * drand48();
    for (int j = 0; j < nb; ++j) {
      const double lam = 60.0 * std::pow(1200.0 / 60.0, nb > 1 ? static_cast<double>(j) / (nb - 1) : 0.0);
      const double nu = 299792.458 / lam * (1.0 + zred);
      const double m = grey(t[0], t[1], t[2], nu) + grey(t[3], t[4], t[5], nu);
      const double sig = 0.07 * m, d = m + sig * (drand48() + drand48() + drand48() - 1.5) * 2.0;
      double* o = &hb[(s * nbm + j) * kBandFields];
      o[kF_NU] = nu; o[kF_HNU] = h_over_k * nu; o[kF_RATIO] = nu / nu_b; o[kF_D] = d; o[kF_W] = 1.0 / (sig * sig);
      const long double l = log2l(static_cast<long double>(nu / nu_b));
      o[kF_LHI] = static_cast<double>(l); o[kF_LLO] = static_cast<double>(l - static_cast<long double>(o[kF_LHI]));
      o[kF_HNU2] = static_cast<double>(static_cast<long double>(o[kF_HNU]) * 1.442695040888963407359924681001892137L);
    }
  }
  std::vector<double> hth(static_cast<size_t>(P) * N), hsg(static_cast<size_t>(P) * N), hlp(N, -1e300);
  for (int64_t n = 0; n < N; ++n)
    for (int i = 0; i < P; ++i) {
      hth[i * N + n] = th0[(n / C) * P + i];
      hsg[i * N + n] = 0.03 * th0[(n / C) * P + i];
    }
  Args a{};
  double* bands = dev<double>(hb.size());
  CK(hipMemcpy(bands, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
  a.bands = bands; a.nbm = nbm; a.nb = nb; a.C = C; a.N = N;
  a.theta = dev<double>(P * N); a.sigma = dev<double>(P * N); a.lnPr = dev<double>(N);
  a.mean = dev<double>(P * N); a.m2 = dev<double>(P * N); a.nacc = dev<uint32_t>(N);
  a.nsteps = nsteps; a.seed = 12345; a.k = k;
  CK(hipMemcpy(a.sigma, hsg.data(), hsg.size() * 8, hipMemcpyHostToDevice));
  const int bs = 256, cpb = bs / G;
  const unsigned grid = static_cast<unsigned>((N + cpb - 1) / cpb);
  const int max_src = static_cast<int>(std::min<int64_t>(S, (cpb + C - 1) / C + 1));
  const size_t lds = static_cast<size_t>(max_src) * nbm * kBandFields * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipMemcpy(a.theta, hth.data(), hth.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(a.lnPr, hlp.data(), hlp.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_mh<G, SEQ>), dim3(grid), dim3(bs), lds, 0, a);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best = std::min(best, static_cast<double>(ms));
  }
  Result res;
  res.us_per_step = best * 1e3 / nsteps;
  res.mean0.resize(N); res.nacc.resize(N);
  CK(hipMemcpy(res.mean0.data(), a.mean, N * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(res.nacc.data(), a.nacc, N * 4, hipMemcpyDeviceToHost));
  CK(hipFree(bands)); CK(hipFree(a.theta)); CK(hipFree(a.sigma)); CK(hipFree(a.lnPr));
  CK(hipFree(a.mean)); CK(hipFree(a.m2)); CK(hipFree(a.nacc));
  return res;
}

template <int G>
static void shape(int64_t S, int C, int nb) {
  const int nsteps = 400, reps = 3;
  const Result a = run<1, false>(S, C, nb, nsteps, reps);
  const Result t = run<G, false>(S, C, nb, nsteps, reps);
  const Result q = run<G, true>(S, C, nb, nsteps, reps);
  size_t same_seq = 0, same_tree = 0;
  for (size_t i = 0; i < a.mean0.size(); ++i) {
    same_seq += (a.mean0[i] == q.mean0[i] && a.nacc[i] == q.nacc[i]);
    same_tree += (a.mean0[i] == t.mean0[i] && a.nacc[i] == t.nacc[i]);
  }
  const double N = static_cast<double>(S) * C;
  std::printf("S=%-7lld C=%-3d nb=%-2d G=%-2d | lane=chain %8.2f us/step %9.3e ch-steps/s | (chain,band) tree %8.2f us (x%.2f)  "
              "seq %8.2f us (x%.2f) | chains identical to lane=chain: seq %zu/%zu tree %zu/%zu\n",
              static_cast<long long>(S), C, nb, G, a.us_per_step, N / (a.us_per_step * 1e-6), t.us_per_step,
              a.us_per_step / t.us_per_step, q.us_per_step, a.us_per_step / q.us_per_step, same_seq,
              a.mean0.size(), same_tree, a.mean0.size());
  std::fflush(stdout);
}

int main() {
  shape<32>(1, 1, 27);        // C1-like: Arp220, one chain
  shape<32>(1, 4, 27);
  shape<32>(1, 64, 27);
  shape<32>(16, 16, 27);
  shape<32>(64, 32, 27);      // 2048 chains
  shape<32>(1024, 64, 27);
  shape<8>(1, 4, 8);
  shape<8>(16, 16, 8);
  shape<8>(256, 16, 8);       // 4096 chains
  shape<8>(1024, 16, 8);
  shape<8>(1024, 64, 8);      // 65 536 chains: one wave per SIMD with lane=chain
  shape<8>(8192, 64, 8);
  shape<8>(100000, 64, 8);    // C2
  return 0;
}
