// Internal helpers shared by the HIP translation units of libmi355mcmc.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mi355mcmc.h"

namespace mi355 {

void set_error(const char* fmt, ...);

#define MI355_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      mi355::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                       __LINE__);                                                         \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

#define MI355_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      mi355::set_error(__VA_ARGS__);      \
      return -1;                          \
    }                                     \
  } while (0)

// Bumped whenever a DevBuf that held memory lets go of it (growth or release): a captured hipGraph
// has the old addresses baked into its kernel arguments, so whoever replays one compares the
// epoch it was captured at with the current one and recaptures when they differ (gmh_api.hip).
inline std::atomic<uint64_t> g_devbuf_epoch{0};

// Owning device buffer; freed with the context.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    if (count <= n && p) return 0;
    release();
    if (count == 0) return 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      n = 0;
      set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
      return -2;
    }
    n = count;
    return 0;
  }
  void release() {
    if (p) {
      (void)hipFree(p);
      g_devbuf_epoch.fetch_add(1, std::memory_order_relaxed);
    }
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

constexpr int kMaxParam = 8;     // largest nparam of any SED model is 6
constexpr int kMaxDevices = 64;  // size of per-device caches of launch attributes
// Band table of a source: [nb_max][kBandFields], the fields of one band adjacent (one LDS address
// per band, the rest are immediate offsets; (d, 1/sig^2) and (log2 base, h nu/k log2 e) are
// 16-byte aligned pairs).
constexpr int kBandFields = 8;
enum BandField {
  kF_NU = 0,     // nu
  kF_HNU = 1,    // h/k * nu
  kF_D = 2,      // flux
  kF_W = 3,      // 1/sig^2
  kF_LHI = 4,    // log2(base), high part
  kF_HNU2 = 5,   // h/k * nu * log2(e)
  kF_LLO = 6,    // log2(base), low part
  kF_RATIO = 7   // nu / nu_b
};
constexpr int kBlockThreads = 256;

struct ClProblem;     // cl_api.hip
struct BeamProblem;   // beam_api.hip
struct GmhState;      // gmh_api.hip
void cl_destroy(ClProblem*);
void beam_destroy(BeamProblem*);
void gmh_destroy(GmhState*);

inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Loads / stores through a pointer whose address space the caller knows (AS = 1 global, 3 LDS,
// 0 generic).  A pointer that reaches a kernel through a struct in memory is generic to the
// compiler: it becomes a FLAT access, which counts on both vmcnt and lgkmcnt and can only be waited
// for with both at zero - every prefetch in flight is drained at the next LDS read.  The fused
// kernels name the address space where they know it.
// Pointer type in address space AS, and the cast to it.  Cast the BASE pointer once and index the
// result: every generic -> LDS cast costs a null check (four VALU instructions).
template <int AS, typename T> struct AsPtr { using type = T*; };
template <typename T> struct AsPtr<1, T> { using type = __attribute__((address_space(1))) T*; };
template <typename T> struct AsPtr<3, T> { using type = __attribute__((address_space(3))) T*; };
template <int AS, typename T>
using as_ptr_t = typename AsPtr<AS, T>::type;
template <int AS, typename T>
__device__ __forceinline__ as_ptr_t<AS, T> as_ptr(T* p) {
  return (as_ptr_t<AS, T>)p;
}

template <int AS, typename T>
__device__ __forceinline__ T ldas(const T* p) {
  if constexpr (AS == 3) return *(const __attribute__((address_space(3))) T*)p;
  else if constexpr (AS == 1) return *(const __attribute__((address_space(1))) T*)p;
  else return *p;
}
template <int AS, typename T>
__device__ __forceinline__ void stas(T* p, T v) {
  if constexpr (AS == 3) *(__attribute__((address_space(3))) T*)p = v;
  else if constexpr (AS == 1) *(__attribute__((address_space(1))) T*)p = v;
  else *p = v;
}

// *p += v in global memory without waiting for *p: one global_atomic_add_f64 with no return value.  The add
// unit of L2 rounds like the vector ALU, and the adds a lane sends to one address are applied in program
// order, so a running sum kept this way equals the load / add / store sequence bit for bit - while the lane
// goes on instead of sitting out a memory round trip per batch of entries (the running co-moments:
// P (P + 1) / 2 entries per chain and step).  Only for memory no other lane touches during the launch.
// (Through a global-address-space pointer: the flat form of the instruction also counts as an LDS operation,
// and the lane's next LDS read would wait for it.)
__device__ __forceinline__ void add_noret(double* p, double v) {
  typedef __attribute__((address_space(1))) double gdouble;
  [[clang::atomic(ignore_denormal_mode, no_fine_grained_memory, no_remote_memory)]] {
    (void)__hip_atomic_fetch_add((gdouble*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Sum of x over the 64 lanes of the wave, valid in lane 63 (wave_sum_last<64>): six data-parallel-primitive steps on the vector
// ALU (shifts by 1, 2, 4, 8 lanes within a row of 16, then lane 15 of a row broadcast to the next row, lane 31
// to the upper half) instead of six round trips through the LDS crossbar per value (__shfl_xor is
// ds_bpermute).  A fixed summation tree: the same bits on every run.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);                          // 0.0 where the lane has no source / is masked out
}
// wave_sum_last<W>: the sums of the wave's groups of W = 16, 32 or 64 consecutive lanes, each valid in the last
// lane of its group.
template <int W>
__device__ __forceinline__ double wave_sum_last(double x) {
  static_assert(W == 16 || W == 32 || W == 64, "groups of 16, 32 or 64 lanes");
  x += dpp_shifted<0x111, 0xf>(x);                          // row_shr:1
  x += dpp_shifted<0x112, 0xf>(x);                          // row_shr:2
  x += dpp_shifted<0x114, 0xf>(x);                          // row_shr:4
  x += dpp_shifted<0x118, 0xf>(x);                          // row_shr:8   lane 15 of each row: the row's sum
  if constexpr (W >= 32) x += dpp_shifted<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  if constexpr (W == 64) x += dpp_shifted<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
  return x;
}

// The xor butterflies of __shfl_xor without the LDS crossbar, same partners, same bits:
//  * row16_xor_sum: v + its partners at lane distance 1, 2, 4, 8 - quad permutes, then (every quad / half row
//    being uniform by then) the mirror of a half row and of the row;
//  * xor16_sum / xor32_sum: v + v[lane ^ 16] / v + v[lane ^ 32] by gfx950's row swaps (v_permlane16_swap /
//    v_permlane32_swap exchange the odd rows of one register with the even rows of another).
template <int CTRL>
__device__ __forceinline__ double dpp_perm(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_xor_sum(double v) {
  v += dpp_perm<0xB1>(v);                                  // quad_perm:[1,0,3,2]
  v += dpp_perm<0x4E>(v);                                  // quad_perm:[2,3,0,1]
  v += dpp_perm<0x141>(v);                                 // row_half_mirror
  v += dpp_perm<0x140>(v);                                 // row_mirror
  return v;
}
__device__ __forceinline__ double xor16_sum(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xor32_sum(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// The whole-wave butterfly v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 (every lane ends with the sum) with the
// operands of __shfl_xor's: row swaps for 32 and 16; then rotations within a row by 8, 4, 2, 1 - after the step
// for distance 2 d the values repeat with period 2 d in the row, so the lane d to either side holds what lane ^ d
// holds.
__device__ __forceinline__ double wave_xor_sum_desc(double v) {
  v = xor32_sum(v);
  v = xor16_sum(v);
  v += dpp_perm<0x128>(v);                                 // row_ror:8
  v += dpp_perm<0x124>(v);                                 // row_ror:4
  v += dpp_perm<0x122>(v);                                 // row_ror:2
  v += dpp_perm<0x121>(v);                                 // row_ror:1
  return v;
}

// Co-moment entries of one step (chains x P (P + 1) / 2) up to which the running co-moments are kept with
// add_noret; beyond, with loads and stores (gmh_comom_update in gmh_device.hpp has the measurements).
#ifndef MI355_COMOM_ATOMIC_MAX
#define MI355_COMOM_ATOMIC_MAX (int64_t{1} << 23)
#endif
constexpr int64_t kComomAtomicMaxEntries = MI355_COMOM_ATOMIC_MAX;

struct SedConsts {
  double h_over_k, nu_b, prefac, Tmin, Tmax, bmin, bmax, nu_rescale;
  double hk_nub;   // h_over_k * nu_b, rounded once like model.py:91,108
  double ln_nub_hi, ln_nub_lo;   // log2(nu_b) as a double-double
  double fast_ok;  // 1: every band keeps the fast kernels' arguments in range over the prior box
  double hk_nub2;  // hk_nub * log2(e): the normalisation argument of the fast path in base 2
  double thick_c;  // log2(nu_rescale / nu_b): optically thick model, tau = 2^(beta (log2(nu/nu_b) - log2 nu0 - thick_c))
};

}  // namespace mi355

struct mi355_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket the main kernel of the last call
  bool ev_valid = false;
  bool capturing = false;         // the stream is being captured into a hipGraph: no events, no allocations
  uint64_t seed = 0;

  // --- SED catalogue -----------------------------------------------------------------
  bool sed_loaded = false;
  mi355_sed_model model{};
  mi355::SedConsts kc{};
  int P = 0;
  int64_t S = 0;
  int nbm = 0;
  int nb_most = 0;                // largest nb[s] of the catalogue (<= nbm, the row width)
  mi355::DevBuf<double> bands;   // [S][nbm][kBandFields]
  mi355::DevBuf<int32_t> nb;     // [S]: band count, + kNbSlow for a source outside the fast domain
  // Per-source fast-path domain (mi355_sed_load): src_slow[s] = 1 when a band of source s leaves the
  // domain of the clamp-free kernels; such sources take the exact OCML path, the others keep the fast
  // one.  k_mh_run over a mixed catalogue: two launches over disjoint lists of chain blocks (built
  // when the block size is known), the slow one on stream2 beside the fast one.
  std::vector<uint8_t> src_slow;
  int64_t nslow = 0;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  mi355::DevBuf<int32_t> blk_fast, blk_slow;
  int64_t nblk_fast = 0, nblk_slow = 0;
  int blk_bs = 0, blk_C = 0;      // the (chains per block, chains per source) the lists were built for

  // --- chain state (N = S*C chains, parameter-major so lanes coalesce) --------------
  bool mh_ready = false;
  int C = 0;
  int64_t N = 0;
  int mhP = 0;                    // parameters per chain of the running sampler
  int64_t mhS = 0;                // chain groups (sources / maps; 1 for binnedCl): N = mhS*C
  int nblock = 0;                 // 0: doBlock=False
  uint32_t blockmask[mi355::kMaxParam] = {0};
  int32_t rank_in_block[mi355::kMaxParam] = {0};
  int npn = 1;                    // normals drawn per step in free-running mode
  bool incremental = false;       // mi355_mh_set_incremental: k_mh_run_inc where it applies
  int lane_map = 0;               // mi355_mh_set_lane_map: 0 automatic, 1 lane = chain, 2 lanes = (chain, band)
  int mh_fused = 1;               // mi355_mh_set_fused: step-wise problems run their fused kernel where one exists (2: whatever the job size)
  int64_t step_total = 0;         // Philox stream position (never reset)
  int64_t stage_step = 0;         // steps since the last init/adapt (iBlock = stage_step % nblock)
  int64_t stage = 0;              // number of init/adapt calls (keys the start stream)
  int64_t nsamp = 0;              // samples accumulated in the moments
  mi355::DevBuf<double> theta, sigma, lnPr, mean, m2;
  mi355::DevBuf<uint32_t> naccept, accept;
  mi355::DevBuf<int32_t> flags;   // per-chain status bits (start randomisation failed ...)

  // --- trace of the last run ----------------------------------------------------------
  int64_t nkeep = 0;
  mi355::DevBuf<double> tr_theta, tr_lnPr;
  mi355::DevBuf<uint8_t> tr_acc;

  // --- scratch ------------------------------------------------------------------------
  mi355::DevBuf<double> scratch_a, scratch_b, scratch_c;
  // device address of the library's fixed scratch array on this context's GPU (api.hip:
  // mi355_rhat_noalloc_async); resolved once at mi355_init with the device current
  double* fixed_scratch = nullptr;

  // --- other problem kinds (owned; each lives in its own translation unit) -----------------
  mi355::ClProblem* cl = nullptr;       // binnedCl / ClData band-power likelihood
  mi355::BeamProblem* beam = nullptr;   // BeamFit maps
  mi355::GmhState* gmh = nullptr;       // step-wise Metropolis-Hastings over any problem kind
};

namespace mi355 {
// step-wise sampler hooks (gmh_api.hip) used by the shared mi355_mh_* entry points
bool gmh_active(const mi355_ctx* c);
void gmh_deactivate(mi355_ctx* c);
// Loading data frees what a running sampler (and its captured graph) points at: every load
// entry point ends the current sampler; the caller initialises a new one.
inline void mh_invalidate(mi355_ctx* c) {
  c->mh_ready = false;
  gmh_deactivate(c);
}
int gmh_run(mi355_ctx* c, int64_t nsteps, int32_t thin, const double* inj_n, const double* inj_u);
// fused SED kernel (api.hip), also the one-launch path of step-wise SED chains
bool sed_mh_fused_ok(const mi355_ctx* c);
int sed_mh_run(mi355_ctx* c, int64_t nsteps, int32_t thin, const double* inj_normals,
               const double* inj_uniforms, int mode, const double* L, double* comom);
int gmh_adapt(mi355_ctx* c, double fac, int random_restart, int no_correlations, int rotate,
              int do_block);
}  // namespace mi355
