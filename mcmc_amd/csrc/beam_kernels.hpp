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
#include <type_traits>

namespace mi355 {

constexpr int kBeamCH = 8;       // chains per register group
constexpr int kBeamNC = 12;      // constants per chain in LDS

// Per-chain constants of the model from PACKAGED parameters.
//  k[0..1] centre, k[2..4] Cxx, Cyy, Cxy, k[5] amplitude, k[6] offset, k[7] plane factor
//  sqrt(1/mu/mu-1), k[8] cos(phi), k[9] sin(phi), k[10] plane flag (0: mu==1, plane is the
//  integer 0), k[11] irregular flag (non-finite constants: use IEEE exp)
// (cs: cos / sin of the position angle th[4] and of the plane's azimuth th[8] - beam_trig -, split off so that a
// sampler whose step leaves the angles alone can keep them: most of the constants' cost is these four calls)
__device__ __forceinline__ void beam_trig(const double* th, double* cs) {
  cs[0] = cos(th[4]);
  cs[1] = sin(th[4]);
  cs[2] = cos(th[8]);
  cs[3] = sin(th[8]);
}
__device__ __forceinline__ void beam_consts_cs(const double* th, const double* cs, double* k) {
  const double c = cs[0], s = cs[1];                                 // BeamModel.py:64
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
  k[8] = cs[2];
  k[9] = cs[3];
  double chk = 0.0;
  for (int i = 0; i < 7; ++i) chk += fabs(k[i]);
  if (k[10] != 0.0) chk += fabs(k[7]) + fabs(k[8]) + fabs(k[9]);
  k[11] = (chk < __builtin_inf()) ? 0.0 : 1.0;
}
__device__ __forceinline__ void beam_consts(const double* th, double* k) {
  double cs[4];
  beam_trig(th, cs);
  beam_consts_cs(th, cs, k);
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

// Folded constants for the pixel loop.  The exponent -q/2 of the reference (BeamModel.py:290) is a
// quadratic in the pixel's coordinates; written out around a reference point (xr, yr) of the MAP - not
// of the chain - its monomials xs^2, ys^2, xs ys (xs = x - xr) are computed once per pixel for all the
// chains of the block, and a chain's exponent is five FMAs on them, in base 2 (the folded C^-1 carries
// -log2(e)/2, so the table-driven 2^y of fastmath.hpp needs no argument scaling).  Amplitude, offset
// and plane fold into one linear form:  r = (d - L(xs, ys)) - amp 2^arg.  20 VALU instructions per
// (pixel, chain) without a plane (mu = 1: what the reference's own runs keep fixed,
// ProcessBeamData.py:220-221), 22 with one; the reference's operation order, kept in beam_model_at, costs
// 46 FLOP, and the form before this one (differences per chain, exponent from dx, dy) cost 25
// instructions.  The expansion around the map's midpoint costs precision where |xs| is many beam widths:
// the monomial terms reach (extent / sigma)^2 and cancel to the exponent near the chain's centre - an
// absolute error of that size times 2^-53 in the exponent, a few 1e-14 relative in the model value for
// maps up to ~20 widths across, inside the 1e-12 budget of lnL (tests/test_beam_gpu.py, tests/
// test_full_size_gpu.py).
//  f[0..5] the quadratic's coefficients of (xs^2, ys^2, xs ys, xs, ys, 1), f[6] amplitude,
//  f[7..9] L's coefficients of (1, xs, ys)
constexpr int kBeamNF = 10;      // folded constants of the fast pixel loop
__device__ __forceinline__ void beam_fold(const double* k, double xr, double yr, double* f) {
  const double a = -0.5 * k[2] * kLog2e, b = -0.5 * k[3] * kLog2e, c = -k[4] * kLog2e;
  const double ox = k[0] - xr, oy = k[1] - yr;             // the chain's centre from the reference point
  f[0] = a;
  f[1] = b;
  f[2] = c;
  f[3] = -(2.0 * a * ox + c * oy);
  f[4] = -(2.0 * b * oy + c * ox);
  f[5] = fma(a * ox, ox, fma(b * oy, oy, (c * ox) * oy));
  f[6] = k[5];
  const bool plane = k[10] != 0.0;
  const double px = plane ? k[7] * k[8] : 0.0, py = plane ? k[7] * k[9] : 0.0;
  f[7] = k[5] * (k[6] - (px * ox + py * oy));
  f[8] = k[5] * px;
  f[9] = k[5] * py;
}

// NCH: chains of the block that exist (1 .. kBeamCH): a block of one chain - the reference's own use -
// runs one chain's instructions, not eight; a chain's arithmetic does not depend on NCH
// The monomial form is as accurate as its largest term allows: the terms reach `big` = the form's
// coefficients times the grid's half extents and cancel to the exponent near the chain's centre, an
// absolute error of about 3 big 2^-53 in the exponent = relative in the model value.  Chains whose form
// exceeds 1024 on the grid (a map more than ~35 beam widths across: 4e-13) count as irregular - the
// block / wave runs the reference's operation order for them (beam_model_at) - and so do non-finite ones.
__device__ __forceinline__ bool beam_form_ok(const double* f, double hx, double hy) {
  const double big = fabs(f[0]) * hx * hx + fabs(f[1]) * hy * hy + fabs(f[2]) * hx * hy + fabs(f[3]) * hx +
                     fabs(f[4]) * hy + fabs(f[5]);
  return big < 1024.0;                                     // (false for NaN)
}

template <bool WPIX, bool PLANE, int NCH>
__device__ __forceinline__ void beam_pixel_fast(const double (*f)[kBeamNF], const double* e2t,
                                                double xs, double ys, double dv, double w,
                                                double* acc) {
  constexpr int G = NCH < 4 ? NCH : 4;                     // chains interleaved at a time
  const double xx = xs * xs, yy = ys * ys, xy = xs * ys;
#pragma unroll
  for (int c0 = 0; c0 < NCH; c0 += G) {
    double arg[G], g[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int c = c0 + i;
      const double lin = fma(f[c][3], xs, fma(f[c][4], ys, f[c][5]));
      // the exponent of a positive-definite form is <= 0 up to rounding; the clamp keeps 2^arg a
      // normal number (exp2_tab_n adds the power of two to the exponent field)
      arg[i] = fmax(fma(f[c][0], xx, fma(f[c][1], yy, fma(f[c][2], xy, lin))), -1000.0);
    }
    exp2_tab_n<G>(arg, e2t, g);
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int c = c0 + i;
      const double t = PLANE ? dv - fma(f[c][8], xs, fma(f[c][9], ys, f[c][7])) : dv - f[c][7];
      const double r = fma(-f[c][6], g[i], t);
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
  const double* xyref;    // [1][4] or [M][4]: reference point (xr, yr) of every map's coordinates (beam_fold) and
                          // their half extents (hx, hy) around it
  int chunk;              // pixels per block (multiple of 256)
  int nchunk;
  double* partial;        // [n][nchunk]
};

// LDS of a block of k_beam_chi2
struct BeamBlockLds {
  double kc[kBeamCH][kBeamNC];
  double fc[kBeamCH][kBeamNF];
  double red[4][kBeamCH];
};

// The constants of one chain from its parameters `th` (packaged here) into slot `slot` of the block's LDS
// tables; an inactive slot gets the constants of a zero vector (the pixel loop reads whole groups of slots).
// ref: the map's reference point and half extents (xr, yr, hx, hy).
// trig (optional, LDS, 6 doubles of the slot's own): the angles of the last call and their cos / sin, reused
// while the angles are the same bits.
__device__ __forceinline__ void beam_chain_consts(double* th, bool active, const double* ref,
                                                  double (*kc)[kBeamNC], double (*fc)[kBeamNF], int slot,
                                                  double* trig = nullptr) {
  double k[kBeamNC];
  if (active) {
    beam_package(th);
    if (trig == nullptr) {
      beam_consts(th, k);
    } else {
      double cs[4];
      if (__double_as_longlong(th[4]) == __double_as_longlong(trig[0]) &&
          __double_as_longlong(th[8]) == __double_as_longlong(trig[1])) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cs[i] = trig[2 + i];
      } else {
        beam_trig(th, cs);
        trig[0] = th[4];
        trig[1] = th[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) trig[2 + i] = cs[i];
      }
      beam_consts_cs(th, cs, k);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kBeamNC; ++i) k[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < kBeamNC; ++i) kc[slot][i] = k[i];
  double f[kBeamNF];
  beam_fold(k, ref[0], ref[1], f);
  if (active && !beam_form_ok(f, ref[2], ref[3])) k[11] = 1.0;   // (kc[.][11] is written below)
#pragma unroll
  for (int i = 0; i < kBeamNF; ++i) fc[slot][i] = f[i];
  kc[slot][11] = k[11];
}

// the constants of chain group `grp` of map `ml` (of this call) into LDS: threads 0 .. 7; the caller
// puts a barrier behind it
__device__ __forceinline__ void beam_group_consts(const BeamArgs& a, int64_t ml, int grp, BeamBlockLds& L) {
  const int tid = threadIdx.x;
  if (tid >= kBeamCH) return;
  const int64_t m = a.m0 + ml;
  const int c0 = grp * kBeamCH;
  const int nch = min(kBeamCH, a.C - c0);
  const double* ref = a.xyref + (a.xy_per_map ? m : 0) * 4;
  double th[kBeamP];
  if (tid < nch) {
    const int64_t e = ml * a.C + c0 + tid;
#pragma unroll
    for (int i = 0; i < kBeamP; ++i) th[i] = a.theta[i * a.n + e];
  }
  beam_chain_consts(th, tid < nch, ref, L.kc, L.fc, tid);
}

// A thread's partial chi-squares of the pixels p0 + vt, p0 + vt + 256, ... < p1 of one map for the nch chains
// whose constants are in (kc, fc): acc[c] (zeroed here).  vt: the thread's index among the 256 of its chunk.
struct BeamPixels {
  const double* xs;       // the map's coordinates, data and weights (w_kind 2: per pixel, else one value)
  const double* ys;
  const double* ds;
  const double* ws;
  int w_kind;
  double w_shared;        // the one weight of the map (w_kind 0 / 1), read once by the caller
  double xr, yr;          // reference point of the map's coordinates
};
// The same sums in the reference's operation order, for blocks with a chain whose folded form does not apply
// (rare).  beam_chunk_sums_exact_fn: as a function of its own, for callers on whose registers its eight inlined
// exponentials must not weigh (the fused sampler).
__device__ __forceinline__ void beam_chunk_sums_exact(const BeamPixels& px, int64_t p0, int64_t p1, int vt,
                                                      const double (*kc)[kBeamNC], double w_shared, double* acc) {
  for (int64_t p = p0 + vt; p < p1; p += 256) {
    const double x = px.xs[p], y = px.ys[p], dv = px.ds[p];
    const double w = (px.w_kind == 2) ? px.ws[p] : w_shared;
#pragma unroll
    for (int c = 0; c < kBeamCH; ++c) {
      const double r = dv - beam_model_at<true>(kc[c], x, y);
      acc[c] += (r * w) * r;
    }
  }
}
__device__ __noinline__ void beam_chunk_sums_exact_fn(const BeamPixels& px, int64_t p0, int64_t p1, int vt,
                                                      const double (*kc)[kBeamNC], double w_shared, double* acc) {
  beam_chunk_sums_exact(px, p0, p1, vt, kc, w_shared, acc);
}
// NCHT: 0 - the instantiation of the pixel loop (1 / up to 4 / up to 8 chains) is picked here; 1, 4, 8 - by the caller.
// AS: address space of the pixel arrays (0: as the pointers say; 1 global, 3 LDS for a caller whose pointers are
// generic - a flat load counts as an LDS operation too and would be waited for with every read of the constants).
template <int NCHT = 0, bool EXACT_CALL = false, int AS = 0>
__device__ __forceinline__ void beam_chunk_sums(const BeamPixels& px, int64_t p0, int64_t p1, int nch, int vt,
                                                const double (*kc)[kBeamNC], const double (*fc)[kBeamNF],
                                                const double* e2t, double* acc) {
  const double* xs = px.xs;
  const double* ys = px.ys;
  const double* ds = px.ds;
  const double* ws = px.ws;
  const double xr = px.xr, yr = px.yr;
  bool irregular = false, plane = false;
  for (int c = 0; c < nch; ++c) {
    irregular = irregular || kc[c][11] != 0.0;
    plane = plane || kc[c][10] != 0.0;
  }
#pragma unroll
  for (int c = 0; c < kBeamCH; ++c) acc[c] = 0.0;
  const double w_shared = px.w_shared;
  if (!irregular) {
    // instantiations of one loop: per-pixel weights or one weight taken out of the sum
    // (GaussianData.py:47), with or without a plane in any chain of the block, 1 / up to 4 / up to 8 chains
    // (fetching a pixel an iteration ahead was measured and is no faster: two waves per SIMD cover
    // the loads already)
    auto pixels = [&](auto wpix, auto pl, auto nc) {
      constexpr bool WP = decltype(wpix)::value;
      for (int64_t p = p0 + vt; p < p1; p += 256)
        beam_pixel_fast<WP, decltype(pl)::value, decltype(nc)::value>(fc, e2t, ldas<AS>(xs + p) - xr, ldas<AS>(ys + p) - yr,
                                                                     ldas<AS>(ds + p), WP ? ldas<AS>(ws + p) : 0.0, acc);
    };
    using T = std::true_type;
    using F = std::false_type;
    auto by_chains = [&](auto wpix, auto pl) {
      if constexpr (NCHT != 0) pixels(wpix, pl, std::integral_constant<int, NCHT>{});
      else if (nch == 1) pixels(wpix, pl, std::integral_constant<int, 1>{});
      else if (nch <= 4) pixels(wpix, pl, std::integral_constant<int, 4>{});
      else pixels(wpix, pl, std::integral_constant<int, kBeamCH>{});
    };
    if (px.w_kind == 2) { if (plane) by_chains(T{}, T{}); else by_chains(T{}, F{}); }
    else                { if (plane) by_chains(F{}, T{}); else by_chains(F{}, F{}); }
    if (px.w_kind != 2) {
#pragma unroll
      for (int c = 0; c < kBeamCH; ++c) acc[c] *= w_shared;
    }
  } else {
    if constexpr (EXACT_CALL) beam_chunk_sums_exact_fn(px, p0, p1, vt, kc, w_shared, acc);
    else beam_chunk_sums_exact(px, p0, p1, vt, kc, w_shared, acc);
  }
}
// ... their sums over the wave into red[c] (lane 0 writes); the four waves' sums of a chunk are then added
// as ((red0 + red1) + red2) + red3
__device__ __forceinline__ void beam_chunk_wave_sums(const double* acc, int nch, int lane, double* red) {
#pragma unroll
  for (int c = 0; c < kBeamCH; ++c) {
    if (c < nch) {                                         // (uniform: a block of one chain reduces one sum)
      const double v = wave_xor_sum_desc(acc[c]);          // the xor butterfly 32 .. 1, on the vector ALU
      if (lane == 0) red[c] = v;
    }
  }
}

// pixel chunk `bx` of map `ml` for the chains of group `grp` (constants in LDS): every thread's partial
// sums, the block's reduction, partial[e][bx].  Contains two barriers; every thread of the block calls it.
__device__ __forceinline__ void beam_chunk(const BeamArgs& a, int bx, int64_t ml, int grp, BeamBlockLds& L,
                                           const double* e2t) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t m = a.m0 + ml;
  const int c0 = grp * kBeamCH;
  const int nch = min(kBeamCH, a.C - c0);
  BeamPixels px;
  px.xs = a.x + (a.xy_per_map ? m * a.npix : 0);
  px.ys = a.y + (a.xy_per_map ? m * a.npix : 0);
  px.ds = a.d + m * a.npix;
  px.ws = a.w + (a.w_kind == 2 ? m * a.npix : (a.w_kind == 1 ? m : 0));
  px.w_kind = a.w_kind;
  px.w_shared = (a.w_kind == 2) ? 0.0 : px.ws[0];
  px.xr = a.xyref[(a.xy_per_map ? m : 0) * 4];
  px.yr = a.xyref[(a.xy_per_map ? m : 0) * 4 + 1];
  const int64_t p0 = static_cast<int64_t>(bx) * a.chunk;
  const int64_t p1 = min(p0 + a.chunk, a.npix);
  double acc[kBeamCH];
  beam_chunk_sums(px, p0, p1, nch, tid, L.kc, L.fc, e2t, acc);
  beam_chunk_wave_sums(acc, nch, lane, L.red[wv]);
  __syncthreads();
  if (tid < nch) {
    const double s = ((L.red[0][tid] + L.red[1][tid]) + L.red[2][tid]) + L.red[3][tid];
    const int64_t e = ml * a.C + c0 + tid;
    a.partial[e * a.nchunk + bx] = s;
  }
  __syncthreads();                                         // (red is free again)
}

__global__ __launch_bounds__(256, 2) void k_beam_chi2(BeamArgs a) {
  __shared__ BeamBlockLds L;
  const double* e2t = stage_exp2_table();              // valid after the barrier below
  beam_group_consts(a, blockIdx.y, blockIdx.z, L);
  __syncthreads();
  beam_chunk(a, blockIdx.x, blockIdx.y, blockIdx.z, L, e2t);
}

// lnL = -chi2/2 with the chunk sums added in ascending order
__device__ __forceinline__ double beam_finish_one(int64_t e, int nchunk, const double* partial) {
  double s = 0.0;
  for (int j = 0; j < nchunk; ++j) s += partial[e * nchunk + j];
  return -0.5 * s;                                                     // NormalizedBeamLikelihood.py:16
}

// ---- the quadratic form on the matrix cores ------------------------------------------------------
// arg[chain][pixel] = sum_k coef_k[chain] mono_k[pixel] is a [chains x 4] . [4 x pixels] product per tile
// of 16 chains of one map and 16 pixels: ONE v_mfma_f64_16x16x4_f64 with A = the chains' coefficients of
// (xs^2, ys^2, xs ys, xs) - a lane's A entry never changes: lane l holds coefficient l / 16 of chain
// l % 16 -, B = those monomials of the tile's pixels (filled in once per grid by k_beam_mono) and C = the
// constant terms.  The VALU is left with the ys term, the exponential and the residual: 16 instructions
// per (pixel, chain) against 21 in k_beam_chi2, and none of its 128 registers of chain constants.  The result tile gives lane l the chains l / 16 + 4 v, v = 0 .. 3, at pixel
// l % 16: a tile costs a lane three 8-byte loads (its monomial, ys and the datum of its pixel; a lane per
// (chain, four pixels) would fetch 72 bytes and be bound by the vector memory pipe), all through one
// scalar base that the scalar unit moves from tile to tile.  The four chains' constants come from their
// owner lanes by shuffles before the loop.
// One wave = 16 chains of one map (chains beyond C idle: the host picks this kernel for C >= 12 of 16)
// over a run of pixels; a lane's four sums are added over the 16 pixel columns at the end.  Exponents are
// not clamped here: chains whose form is large on the grid (beam_form_ok) count as irregular, and a wave
// with such a chain runs the exact model.
typedef double beam_d4 __attribute__((ext_vector_type(4)));

struct BeamMfArgs {
  const double* mono;     // [5][mstride]: xs, ys, xs^2, ys^2, xs ys; zeros beyond npix (mstride: multiple of 16)
  int64_t mstride;
  const double* x;
  const double* y;
  const double* d;
  const double* w;
  int w_kind;
  int64_t npix;
  int64_t m0;
  int C;
  int ngrp;               // groups of 16 chains per map
  int64_t n;              // nm * C
  int64_t nunit;          // nm * ngrp
  const double* theta;    // [9][n]
  double xr, yr;
  double hx, hy;          // half extents of the grid around (xr, yr)
  int W;                  // waves (pixel runs) per unit
  int64_t run;            // pixels per run (multiple of 16)
  double* partial;        // [n][W]
};

// a double at (wave-uniform base) + (the lane's 32-bit byte offset): a global load with a scalar base
__device__ __forceinline__ double beam_ldu(const double* base, uint32_t off) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off);
}

struct BeamMfChains {     // constants of the lane's four chains l / 16 + 4 v
  double e[4];            // coefficient of ys
  double amp[4];
  double p0[4], px[4], py[4];   // the linear form L
};

// Measured on C5 (2000 maps x 16 chains x 256^2 pixels, tools/prof_beam.sh, profiles/r3_beam_summary.txt):
// 1.63 ms against k_beam_chi2's 2.09 (2.26 before its own monomial form).  The time does not move with
// occupancy (2 to 16 waves per SIMD: 1.71 - 1.62 ms), with the table reads made conflict-free, or with
// the loads fetched further ahead: the FP64 matrix and vector instructions share the SIMD's FP64
// multipliers, and the clock under this instruction mix is 2.03 GHz (GRBM_GUI_ACTIVE), not 2.4 - the
// kernel runs at about 0.8 of the FP64 issue rate the chip sustains.
constexpr int kBeamMfU = 8;                 // tiles per loop iteration: their loads are issued together

// U tiles of 16 pixels from pixel p on (mono_p, d_p, w_p point at it)
template <bool WPIX, bool PLANE, int U>
__device__ __forceinline__ void beam_tiles_mf(const BeamMfChains& ch, const beam_d4& cinit, double acoef,
                                              const double* e2t, const double* mono_p, uint32_t ob,
                                              uint32_t oy, uint32_t ox, const double* d_p, const double* w_p,
                                              uint32_t od, double* acc) {
  double bv[U], ys[U], dv[U], xs[U], wv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    bv[u] = beam_ldu(mono_p + 16 * u, ob);
    ys[u] = beam_ldu(mono_p + 16 * u, oy);
    dv[u] = beam_ldu(d_p + 16 * u, od);
    xs[u] = PLANE ? beam_ldu(mono_p + 16 * u, ox) : 0.0;
    wv[u] = WPIX ? beam_ldu(w_p + 16 * u, od) : 0.0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const beam_d4 q = __builtin_amdgcn_mfma_f64_16x16x4f64(acoef, bv[u], cinit, 0, 0, 0);
    double arg[4], g[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) arg[v] = fma(ch.e[v], ys[u], q[v]);
    exp2_tab_n<4>(arg, e2t, g);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double t = PLANE ? dv[u] - fma(ch.px[v], xs[u], fma(ch.py[v], ys[u], ch.p0[v])) : dv[u] - ch.p0[v];
      const double r = fma(-ch.amp[v], g[v], t);
      acc[v] = WPIX ? fma(r * wv[u], r, acc[v]) : fma(r, r, acc[v]);
    }
  }
}

__global__ __launch_bounds__(256, 3) void k_beam_chi2_mf(BeamMfArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);             // (scalar: so is everything derived from it)
  const double* e2t = stage_exp2_table();
  __syncthreads();
  const int64_t wid = static_cast<int64_t>(blockIdx.x) * 4 + wv;       // wave-uniform
  if (wid >= a.nunit * a.W) return;
  const int64_t unit = wid / a.W;
  const int run_i = static_cast<int>(wid - unit * a.W);
  const int64_t ml = unit / a.ngrp;
  const int cg = static_cast<int>(unit - ml * a.ngrp);
  const int64_t m = a.m0 + ml;
  const int j16 = lane & 15, kq = lane >> 4;
  // the lane's OWN chain cg 16 + j16: its constants are computed here and handed out below
  const int c_own = cg * 16 + j16;
  const int64_t e_own = ml * a.C + (c_own < a.C ? c_own : a.C - 1);
  double k[kBeamNC], f[kBeamNF];
  {
    double th[kBeamP];
#pragma unroll
    for (int i = 0; i < kBeamP; ++i) th[i] = a.theta[i * a.n + e_own];
    beam_package(th);
    beam_consts(th, k);
    beam_fold(k, a.xr, a.yr, f);
  }
  const bool irregular = k[11] != 0.0 || !beam_form_ok(f, a.hx, a.hy);   // (also keeps exp2_tab_n in range: no clamp here)
  const bool wave_irregular = __builtin_amdgcn_ballot_w64(irregular) != 0;
  const bool wave_plane = __builtin_amdgcn_ballot_w64(k[10] != 0.0) != 0;
  const double* dl = a.d + m * a.npix;                                 // wave-uniform
  const double* wl = a.w + (a.w_kind == 2 ? m * a.npix : (a.w_kind == 1 ? m : 0));
  const int64_t p0 = run_i * a.run;
  const int64_t p1 = min(p0 + a.run, a.npix);
  if (wave_irregular) {
    // the exact model, a lane per (own chain, every fourth pixel); summed over the four row groups
    double acc = 0.0;
    for (int64_t p = p0 + kq; p < p1; p += 4) {
      const double w = (a.w_kind == 2) ? wl[p] : wl[0];
      const double r = dl[p] - beam_model_at<true>(k, a.x[p], a.y[p]);
      acc += (r * w) * r;
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (c_own < a.C && kq == 0) a.partial[e_own * a.W + run_i] = acc;
    return;
  }
  // A[i = chain j16][k = kq]: coefficient of xs^2, ys^2, xs ys, xs; the lane's chains kq + 4 v from their owners
  const double acoef = (kq == 0) ? f[0] : (kq == 1) ? f[1] : (kq == 2) ? f[2] : f[3];
  BeamMfChains ch;
  beam_d4 cinit;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int src = kq + 4 * v;                            // lanes 0 .. 15 own chains 0 .. 15 of the group
    cinit[v] = __shfl(f[5], src);
    ch.e[v] = __shfl(f[4], src);
    ch.amp[v] = __shfl(f[6], src);
    ch.p0[v] = __shfl(f[7], src);
    ch.px[v] = __shfl(f[8], src);
    ch.py[v] = __shfl(f[9], src);
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const uint32_t ob = static_cast<uint32_t>((((kq == 3) ? 0 : (kq + 2)) * a.mstride + j16) * 8);   // rows xs^2, ys^2, xs ys, xs
  const uint32_t oy = static_cast<uint32_t>((a.mstride + j16) * 8);
  const uint32_t ox = static_cast<uint32_t>(j16 * 8);
  auto pixels = [&](auto wpix, auto pl) {
    constexpr bool WP = decltype(wpix)::value;
    constexpr bool PL = decltype(pl)::value;
    int64_t p = p0;
    for (; p + 16 * kBeamMfU <= p1; p += 16 * kBeamMfU)
      beam_tiles_mf<WP, PL, kBeamMfU>(ch, cinit, acoef, e2t, a.mono + p, ob, oy, ox, dl + p, wl + p, /* od = */ ox, acc);
    for (; p + 16 <= p1; p += 16)
      beam_tiles_mf<WP, PL, 1>(ch, cinit, acoef, e2t, a.mono + p, ob, oy, ox, dl + p, wl + p, /* od = */ ox, acc);
    // the grid's last pixels (fewer than 16): pixel column j16 while there is one, the form by FMAs
    // (the four chains' coefficients fetched by every lane, before the lanes without a pixel drop out)
    if (p < p1) {
      double cq[4][4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) cq[v][i] = __shfl(f[i], kq + 4 * v);
      if (p + j16 < p1) {
        const int64_t pp = p + j16;
        const double xs = a.mono[pp], ys = a.mono[a.mstride + pp];
        const double xx = a.mono[2 * a.mstride + pp], yy = a.mono[3 * a.mstride + pp], xy = a.mono[4 * a.mstride + pp];
        const double dv = dl[pp];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double lin = fma(cq[v][3], xs, fma(ch.e[v], ys, cinit[v]));
          const double g = exp2_tab(fma(cq[v][0], xx, fma(cq[v][1], yy, fma(cq[v][2], xy, lin))), e2t);
          const double t = PL ? dv - fma(ch.px[v], xs, fma(ch.py[v], ys, ch.p0[v])) : dv - ch.p0[v];
          const double r = fma(-ch.amp[v], g, t);
          acc[v] = WP ? fma(r * wl[pp], r, acc[v]) : fma(r, r, acc[v]);
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if (a.w_kind == 2) { if (wave_plane) pixels(T{}, T{}); else pixels(T{}, F{}); }
  else               { if (wave_plane) pixels(F{}, T{}); else pixels(F{}, F{}); }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    double sum = acc[v];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    sum += __shfl_xor(sum, 8);
    if (a.w_kind != 2) sum *= wl[0];
    const int c = cg * 16 + kq + 4 * v;
    if (j16 == 0 && c < a.C) a.partial[(ml * a.C + c) * a.W + run_i] = sum;
  }
}

// the monomials of a coordinate grid around its reference point, [5][mstride], zeros beyond npix
__global__ void k_beam_mono(int64_t npix, int64_t mstride, const double* __restrict__ x,
                            const double* __restrict__ y, double xr, double yr, double* __restrict__ mono) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= mstride) return;
  const double xs = p < npix ? x[p] - xr : 0.0, ys = p < npix ? y[p] - yr : 0.0;
  mono[p] = xs;
  mono[mstride + p] = ys;
  mono[2 * mstride + p] = xs * xs;
  mono[3 * mstride + p] = ys * ys;
  mono[4 * mstride + p] = xs * ys;
}

__global__ void k_beam_finish(int64_t n, int nchunk, const double* __restrict__ partial,
                              double* __restrict__ lnl) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  lnl[e] = beam_finish_one(e, nchunk, partial);
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
