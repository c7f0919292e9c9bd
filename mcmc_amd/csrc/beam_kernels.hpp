// HIP kernels of the 2-D elliptical-Gaussian beam fit.
//
// Reference arithmetic: MCMC/BeamFit/BeamModel.py:55-71 (sigma_1, sigma_2, angle -> sig2_xy, rho),
// :128-136 (packed C^-1), :165-187 (prior), :288-290 (gauss2d);
// MCMC/BeamFit/OffsetNormalizedBeamModel.py:78-82 (at), :93-101 (prior), :123-133
// (package/unpackage), :146-152 (planexy); MCMC/BeamFit/NormalizedBeamLikelihood.py:9-16
// (-chi2/2); MCMC/GaussianData.py:47,64 (chi2 = (d-m)^T N^-1 (d-m), diagonal N).
//
// Mapping: a block owns a chunk of one map's pixels and evaluates it for CH chains of that map
// at once: the pixel values are loaded once per chain group, the per-chain constants (C^-1,
// plane) are computed once per block into LDS and read as broadcasts, and each thread keeps CH
// partial chi-squares in registers.  Partial sums per (map, chunk, chain) are added in a fixed
// order by k_beam_finish, so a result does not depend on scheduling.
#pragma once
#include "beam_device.hpp"
#include "common.hpp"
#include "fastmath.hpp"

namespace mi355 {

constexpr int kBeamCH = 8;       // chains per register group
constexpr int kBeamNC = 12;      // constants per chain in LDS
constexpr int kBeamNF = 9;       // folded constants of the fast pixel loop

// Per-chain constants of the model from PACKAGED parameters.
//  k[0..1] centre, k[2..4] Cxx, Cyy, Cxy, k[5] amplitude, k[6] offset, k[7] plane factor
//  sqrt(1/mu/mu-1), k[8] cos(phi), k[9] sin(phi), k[10] plane flag (0: mu==1, plane is the
//  integer 0), k[11] irregular flag (non-finite constants: use IEEE exp)
__device__ __forceinline__ void beam_consts(const double* th, double* k) {
  const double ang = th[4];
  const double c = cos(ang), s = sin(ang);                           // BeamModel.py:64
  const double s12 = th[2] * th[2], s22 = th[3] * th[3];
  const double sig2x = c * c * s12 + s * s * s22;                    // :68
  const double sig2y = s * s * s12 + c * c * s22;
  const double rt = sqrt(sig2x * sig2y);
  const double rho = c * s * (s22 - s12) / rt;                       // :69
  const double det = sig2x * sig2y * (1.0 - rho * rho);              // :132
  k[0] = th[0];
  k[1] = th[1];
  k[2] = sig2y / det;                                                // :135-136
  k[4] = -rho * rt / det;
  k[3] = sig2x / det;
  k[5] = th[5];
  k[6] = th[6];
  const double mu = th[7];
  k[10] = (mu == 1.0) ? 0.0 : 1.0;                                   // planexy :150
  k[7] = sqrt(1 / mu / mu - 1);                                      // :152
  k[8] = cos(th[8]);
  k[9] = sin(th[8]);
  double chk = 0.0;
  for (int i = 0; i < 7; ++i) chk += fabs(k[i]);
  if (k[10] != 0.0) chk += fabs(k[7]) + fabs(k[8]) + fabs(k[9]);
  k[11] = (chk < __builtin_inf()) ? 0.0 : 1.0;
}

// model value at one point, reference operation order (OffsetNormalizedBeamModel.py:78-82)
template <bool EXACT>
__device__ __forceinline__ double beam_model_at(const double* k, double x, double y) {
  const double dx = x - k[0], dy = y - k[1];
  const double q = dx * dx * k[2] + dy * dy * k[3] + 2 * dx * dy * k[4];   // BeamModel.py:290
  const double g = EXACT ? exp(-0.5 * q) : exp_fast(-0.5 * q);
  double v = g + k[6];
  if (k[10] != 0.0) v += k[7] * (dx * k[8] + dy * k[9]);             // planexy(mu, phi, x-x0, y-y0)
  return k[5] * v;
}

// Folded constants for the pixel loop: the factor -1/2 of the exponent goes into C^-1 and the
// plane factor into its direction cosines, so that a pixel costs 5 FMA-class operations for the
// exponent, 3 for offset + plane and 1 for the residual (the reference's operation order, kept in
// beam_model_at, costs 22).  The reassociation moves a model value by a few ulp, far inside the
// 1e-12 budget of lnL (tests/test_beam_gpu.py).
// The exponent is produced directly in base 2 (the folded C^-1 carries log2 e), so the
// exponential needs no argument scaling: the table-driven 2^y of fastmath.hpp (exp2_tab_n), the
// 8 chains of a pixel interleaved.
//  f[0..1] centre, f[2] -Cxx/2 log2e, f[3] -Cyy/2 log2e, f[4] -Cxy log2e, f[5] offset,
//  f[6..7] plane gradient, f[8] amplitude
__device__ __forceinline__ void beam_fold(const double* k, double* f) {
  f[0] = k[0];
  f[1] = k[1];
  f[2] = -0.5 * k[2] * kLog2e;
  f[3] = -0.5 * k[3] * kLog2e;
  f[4] = -k[4] * kLog2e;
  f[5] = k[6];
  const bool plane = k[10] != 0.0;
  f[6] = plane ? k[7] * k[8] : 0.0;
  f[7] = plane ? k[7] * k[9] : 0.0;
  f[8] = k[5];
}

template <bool WPIX>
__device__ __forceinline__ void beam_pixel_fast(const double (*f)[kBeamNF], const double* e2t,
                                                double x, double y, double dv, double w,
                                                double* acc) {
  constexpr int G = 4;                                     // chains interleaved at a time
#pragma unroll
  for (int c0 = 0; c0 < kBeamCH; c0 += G) {
    double dx[G], dy[G], arg[G], g[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int c = c0 + i;
      dx[i] = x - f[c][0];
      dy[i] = y - f[c][1];
      const double u = fma(f[c][4], dy[i], f[c][2] * dx[i]);
      // the exponent of a positive-definite form is <= 0; the clamp keeps 2^arg a normal number
      // (exp2_tab_n adds the power of two to the exponent field)
      arg[i] = fmax(fma(dx[i], u, (f[c][3] * dy[i]) * dy[i]), -1000.0);
    }
    exp2_tab_n<G>(arg, e2t, g);
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int c = c0 + i;
      const double v = fma(dx[i], f[c][6], fma(dy[i], f[c][7], g[i] + f[c][5]));
      const double r = fma(-f[c][8], v, dv);
      acc[c] = WPIX ? fma(r * w, r, acc[c]) : fma(r, r, acc[c]);
    }
  }
}

struct BeamArgs {
  const double* x;        // [npix] or [M][npix]
  const double* y;
  int xy_per_map;
  const double* d;        // [M][npix]
  const double* w;        // 1/sig^2: [1], [M] or [M][npix]
  int w_kind;             // 0 scalar, 1 per map, 2 per pixel
  int64_t npix;
  int64_t m0;             // first map of this call
  int C;                  // chains per map
  int64_t n;              // nm * C parameter vectors
  const double* theta;    // [9][n] parameter-major, vector e = (m - m0)*C + c
  int chunk;              // pixels per block (multiple of 256)
  int nchunk;
  double* partial;        // [n][nchunk]
};

__global__ __launch_bounds__(256) void k_beam_chi2(BeamArgs a) {
  __shared__ double kc[kBeamCH][kBeamNC];
  __shared__ double fc[kBeamCH][kBeamNF];
  __shared__ double red[4][kBeamCH];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double* e2t = stage_exp2_table();              // valid after the barrier below
  const int64_t ml = blockIdx.y;                       // map within this call
  const int64_t m = a.m0 + ml;
  const int grp = blockIdx.z;                          // chain group
  const int c0 = grp * kBeamCH;
  const int nch = min(kBeamCH, a.C - c0);
  if (tid < kBeamCH) {
    double th[kBeamP];
    double k[kBeamNC];
    if (tid < nch) {
      const int64_t e = ml * a.C + c0 + tid;
#pragma unroll
      for (int i = 0; i < kBeamP; ++i) th[i] = a.theta[i * a.n + e];
      beam_package(th);
      beam_consts(th, k);
    } else {
#pragma unroll
      for (int i = 0; i < kBeamNC; ++i) k[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < kBeamNC; ++i) kc[tid][i] = k[i];
    double f[kBeamNF];
    beam_fold(k, f);
#pragma unroll
    for (int i = 0; i < kBeamNF; ++i) fc[tid][i] = f[i];
  }
  __syncthreads();
  bool irregular = false;
  for (int c = 0; c < nch; ++c) irregular = irregular || kc[c][11] != 0.0;

  const double* xs = a.x + (a.xy_per_map ? m * a.npix : 0);
  const double* ys = a.y + (a.xy_per_map ? m * a.npix : 0);
  const double* ds = a.d + m * a.npix;
  const double* ws = a.w + (a.w_kind == 2 ? m * a.npix : (a.w_kind == 1 ? m : 0));
  const int64_t p0 = static_cast<int64_t>(blockIdx.x) * a.chunk;
  const int64_t p1 = min(p0 + a.chunk, a.npix);
  double acc[kBeamCH];
#pragma unroll
  for (int c = 0; c < kBeamCH; ++c) acc[c] = 0.0;
  const double w_shared = (a.w_kind == 2) ? 0.0 : ws[0];
  if (!irregular && a.w_kind == 2) {
    for (int64_t p = p0 + tid; p < p1; p += 256)
      beam_pixel_fast<true>(fc, e2t, xs[p], ys[p], ds[p], ws[p], acc);   // GaussianData.py:47
  } else if (!irregular) {
    for (int64_t p = p0 + tid; p < p1; p += 256)                    // one weight: taken out of the sum
      beam_pixel_fast<false>(fc, e2t, xs[p], ys[p], ds[p], 0.0, acc);
#pragma unroll
    for (int c = 0; c < kBeamCH; ++c) acc[c] *= w_shared;
  } else {
    for (int64_t p = p0 + tid; p < p1; p += 256) {
      const double x = xs[p], y = ys[p], dv = ds[p];
      const double w = (a.w_kind == 2) ? ws[p] : w_shared;
#pragma unroll
      for (int c = 0; c < kBeamCH; ++c) {
        const double r = dv - beam_model_at<true>(kc[c], x, y);
        acc[c] += (r * w) * r;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kBeamCH; ++c) {
    double v = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wv][c] = v;
  }
  __syncthreads();
  if (tid < nch) {
    const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    const int64_t e = ml * a.C + c0 + tid;
    a.partial[e * a.nchunk + blockIdx.x] = s;
  }
}

// lnL = -chi2/2 with the chunk sums added in ascending order
__global__ void k_beam_finish(int64_t n, int nchunk, const double* __restrict__ partial,
                              double* __restrict__ lnl) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double s = 0.0;
  for (int j = 0; j < nchunk; ++j) s += partial[e * nchunk + j];
  lnl[e] = -0.5 * s;                                                 // NormalizedBeamLikelihood.py:16
}

// prior and package of n flat vectors, host layout [n][9]
__global__ void k_beam_prior_package(int64_t n, const double* __restrict__ theta, BeamPriorBox box,
                                     double* __restrict__ prior, double* __restrict__ packaged) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double th[kBeamP];
#pragma unroll
  for (int i = 0; i < kBeamP; ++i) th[i] = theta[e * kBeamP + i];
  beam_package(th);
  if (prior) prior[e] = beam_prior(th, box);
  if (packaged) {
#pragma unroll
    for (int i = 0; i < kBeamP; ++i) packaged[e * kBeamP + i] = th[i];
  }
}

// model(*package(theta)).atxy(x, y): out [n][npts], exact (OCML) arithmetic
__global__ void k_beam_model(int64_t n, const double* __restrict__ theta, int64_t npts,
                             const double* __restrict__ x, const double* __restrict__ y,
                             double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n * npts) return;
  const int64_t it = e / npts, ip = e % npts;
  double th[kBeamP], k[kBeamNC];
#pragma unroll
  for (int i = 0; i < kBeamP; ++i) th[i] = theta[it * kBeamP + i];
  beam_package(th);
  beam_consts(th, k);
  out[e] = beam_model_at<true>(k, x[ip], y[ip]);
}

// gauss2d (BeamModel.py:288-290) with the packed C^-1 given, in the reference's operation order
__global__ void k_beam_gauss2d(int64_t n, const double* __restrict__ x, const double* __restrict__ y,
                               double x0, double y0, double cxx, double cxy, double cyy,
                               double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const double dx = x[e] - x0, dy = y[e] - y0;
  out[e] = exp(-0.5 * (dx * dx * cxx + dy * dy * cyy + 2 * dx * dy * cxy));
}

// BeamData.stats (BeamData.py:35-71), one block: pass 0 sum(d) -> mean; pass 1 the five
// data-weighted sums over the pixels with d - mean > sigcut*sig (all pixels if !has_cut).
__global__ __launch_bounds__(256) void k_beam_stats(int64_t npix, const double* __restrict__ x,
                                                    const double* __restrict__ y,
                                                    const double* __restrict__ d,
                                                    const double* __restrict__ sig, int sig_per_pixel,
                                                    int has_cut, double sigcut, int do_abs,
                                                    double* __restrict__ out /*[4]*/) {
  __shared__ double sh[5][256];
  __shared__ double mean_s;
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int64_t p = tid; p < npix; p += 256) s += do_abs ? fabs(d[p]) : d[p];
  sh[0][tid] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) sh[0][tid] += sh[0][tid + off];
    __syncthreads();
  }
  if (tid == 0) mean_s = sh[0][0] / static_cast<double>(npix);
  __syncthreads();
  const double mean = mean_s;
  double a[5] = {0, 0, 0, 0, 0};
  for (int64_t p = tid; p < npix; p += 256) {
    const double dd = (do_abs ? fabs(d[p]) : d[p]) - mean;
    if (has_cut && !(dd > sigcut * sig[sig_per_pixel ? p : 0])) continue;
    a[0] += dd;
    a[1] += x[p] * dd;
    a[2] += y[p] * dd;
    a[3] += x[p] * x[p] * dd;
    a[4] += y[p] * y[p] * dd;
  }
  __syncthreads();
  for (int j = 0; j < 5; ++j) sh[j][tid] = a[j];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off)
      for (int j = 0; j < 5; ++j) sh[j][tid] += sh[j][tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    const double norm = sh[0][0];
    const double ex = sh[1][0] / norm, ey = sh[2][0] / norm;
    out[0] = ex;
    out[1] = ey;
    out[2] = sh[3][0] / norm - ex * ex;
    out[3] = sh[4][0] / norm - ey * ey;
  }
}

}  // namespace mi355
