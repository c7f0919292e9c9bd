// Counter-based RNG for the proposal / accept draws: Philox4x32-10 (Salmon et al., SC11)
// + Box-Muller.  Replaces numpy's global MT19937 (MCMC/Proposal.py:151, MCMC/MCMC.py:264).
// Stream definition (restated bit-for-bit by oracle/refcpu.py philox_normal/philox_uniform):
//   key  = (seed_lo ^ purpose*0x9E3779B9, seed_hi)
//   ctr  = (pair_lo, pair_hi, chain_lo, chain_hi),  pair = idx >> 1
//   words w0..w3 -> u1 = u53(w0,w1), u2 = u53(w2,w3),  u53(hi,lo) = ((hi<<21 | lo>>11)+0.5)*2^-53
//   uniform(idx)  = idx even ? u1 : u2
//   normal(idx)   = sqrt(-2 ln u1) * (idx even ? cos : sin)(2 pi u2)
// ln and sin/cos are the library's own table-driven kernels (fastmath.hpp log_bm, sincos_bm: ln to
// 1 ulp, sine and cosine to 2e-16 absolute); the oracle evaluates the same formula with numpy, so
// device and oracle normals agree to ~1e-15 absolute, far inside the 1e-12 chain tolerances.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fastmath.hpp"

namespace mi355 {

enum : uint32_t { RNG_NORMAL = 0, RNG_UNIFORM = 1, RNG_START = 2 };

struct Philox4 {
  uint32_t w[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32 -> 64 multiply (v_mad_u64_u32) per product instead of a mul_hi / mul_lo pair
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
    const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
    // three-input XOR in one instruction (v_bitop3_b32, truth table 0x96; the compiler leaves two v_xor_b32)
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  // = (double(hi<<21 | lo>>11) + 0.5) * 2^-53 with its one rounding: the low 21 bits and the half
  // are exact in c, the last FMA rounds the exact sum (two 32-bit conversions and two FMAs instead
  // of a 64-bit conversion)
  const double c = fma(static_cast<double>(lo >> 11), 0x1.0p-53, 0x1.0p-54);
  return fma(static_cast<double>(hi), 0x1.0p-32, c);
}

__device__ __forceinline__ Philox4 philox_pair(uint64_t seed, uint32_t purpose, uint64_t chain,
                                               uint64_t pair) {
  const uint32_t k0 = static_cast<uint32_t>(seed) ^ (purpose * 0x9E3779B9u);
  const uint32_t k1 = static_cast<uint32_t>(seed >> 32);
  return philox4x32_10(static_cast<uint32_t>(pair), static_cast<uint32_t>(pair >> 32),
                       static_cast<uint32_t>(chain), static_cast<uint32_t>(chain >> 32), k0, k1);
}

// both uniforms of one pair
__device__ __forceinline__ void philox_uniform2(uint64_t seed, uint32_t purpose, uint64_t chain,
                                                uint64_t pair, double& ua, double& ub) {
  const Philox4 r = philox_pair(seed, purpose, chain, pair);
  ua = u53(r.w[0], r.w[1]);
  ub = u53(r.w[2], r.w[3]);
}

// both normals of one pair (cos branch = even idx, sin branch = odd idx); bm: the Box-Muller table,
// bm_of(stage_exp2_table()) in the kernels that stage it, the table in device memory otherwise
// (same rows, same arithmetic, same normals)
__device__ __forceinline__ void philox_normal2(uint64_t seed, uint32_t purpose, uint64_t chain,
                                               uint64_t pair, const double* bm, double& za, double& zb) {
  double u1, u2;
  philox_uniform2(seed, purpose, chain, pair, u1, u2);
  const double r = sqrt_fast(-2.0 * log_bm(u1, bm));
  double s, c;
  sincos_bm(u2, bm, &s, &c);
  za = r * c;
  zb = r * s;
}
__device__ __forceinline__ void philox_normal2(uint64_t seed, uint32_t purpose, uint64_t chain,
                                               uint64_t pair, double& za, double& zb) {
  philox_normal2(seed, purpose, chain, pair, kBmTable, za, zb);
}

__device__ __forceinline__ double philox_normal(uint64_t seed, uint32_t purpose, uint64_t chain,
                                                uint64_t idx, const double* bm = kBmTable) {
  double za, zb;
  philox_normal2(seed, purpose, chain, idx >> 1, bm, za, zb);
  return (idx & 1) ? zb : za;
}

__device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t purpose, uint64_t chain,
                                                 uint64_t idx) {
  double ua, ub;
  philox_uniform2(seed, purpose, chain, idx >> 1, ua, ub);
  return (idx & 1) ? ub : ua;
}

}  // namespace mi355
