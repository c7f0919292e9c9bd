// Issue cost of the VALU instructions the SED kernels are made of, on one CU of an MI355X:
// cycles per wave64 instruction with 1 and 2 waves per SIMD (s_memtime around a loop of 8
// independent chains x 16 instructions, 2000 iterations).  Build and run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP2(S) S S
#define REP16(S) REP2(REP2(REP2(REP2(S))))

constexpr int kIters = 2000;
constexpr int kPerIter = 8 * 16;

#define KERNEL(NAME, DECL, BODY, SINK)                                             \
  __global__ void NAME(long long* out, double seed) {                              \
    DECL;                                                                          \
    long long t0 = __builtin_readcyclecounter();                                   \
    for (int it = 0; it < kIters; ++it) {                                          \
      REP16(BODY)                                                                  \
    }                                                                              \
    long long t1 = __builtin_readcyclecounter();                                   \
    SINK;                                                                          \
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
  }

#define DECL_D                                                                          \
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, \
         a6 = seed + 6, a7 = seed + 7;                                                  \
  double c = seed * 0.5, e = seed * 0.25;                                               \
  int ia = threadIdx.x
#define SINK_D                                                    \
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = ia
#define DECL_I                                                                          \
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
           a6 = a0 + 6, a7 = a0 + 7;                                                    \
  unsigned c = (unsigned)seed | 1u
#define SINK_I \
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = c
#define DECL_L                                                                          \
  unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,    \
                     a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                             \
  unsigned c = (unsigned)seed | 1u, e = c + 2
#define SINK_L \
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = c

#define OP_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(e));
#define OP_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a##i));
#define OP_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a##i) : "v"(ia));
#define OP_RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
#define OP_CVT(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia) : "v"(a##i));
#define OP_CMP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a##i), "v"(c) : "vcc");
#define OP_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c) : "vcc");
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define OP_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(c), "v"(e) : "vcc");
#define OP_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a##i) : "v"(c));
#define OP_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));
#define OP_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));

KERNEL(k_fma, DECL_D, REP8(OP_FMA), SINK_D)
KERNEL(k_add, DECL_D, REP8(OP_ADD), SINK_D)
KERNEL(k_mul, DECL_D, REP8(OP_MUL), SINK_D)
KERNEL(k_rndne, DECL_D, REP8(OP_RNDNE), SINK_D)
KERNEL(k_ldexp, DECL_D, REP8(OP_LDEXP), SINK_D)
KERNEL(k_rcp, DECL_D, REP8(OP_RCP), SINK_D)
KERNEL(k_cvt, DECL_D, REP8(OP_CVT), SINK_D)
KERNEL(k_cmp, DECL_D, REP8(OP_CMP), SINK_D)
KERNEL(k_mov64, DECL_D, REP8(OP_MOV64), SINK_D)
KERNEL(k_pkfma32, DECL_D, REP8(OP_PKFMA32), SINK_D)
KERNEL(k_addu, DECL_I, REP8(OP_ADDU), SINK_I)
KERNEL(k_xor, DECL_I, REP8(OP_XOR), SINK_I)
KERNEL(k_cndmask, DECL_I, REP8(OP_CNDMASK), SINK_I)
KERNEL(k_mullo, DECL_I, REP8(OP_MULLO), SINK_I)
KERNEL(k_mulhi, DECL_I, REP8(OP_MULHI), SINK_I)
KERNEL(k_fma32, DECL_I, REP8(OP_FMA32), SINK_I)
KERNEL(k_mad64, DECL_L, REP8(OP_MAD64), SINK_L)

// one dependent chain: latency
#define ONE(X) X(0)
KERNEL(k_fma_dep, DECL_D, ONE(OP_FMA), SINK_D)
KERNEL(k_rndne_dep, DECL_D, ONE(OP_RNDNE), SINK_D)
KERNEL(k_ldexp_dep, DECL_D, ONE(OP_LDEXP), SINK_D)

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));               \
      return 1;                                                         \
    }                                                                   \
  } while (0)

template <typename K>
static int run(const char* name, K kern, long long* dout, int per_iter) {
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int threads = 256 * waves_per_simd;   // one block on one CU: 4 or 8 waves
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dout, 1.0);   // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dout, 1.0);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    std::vector<long long> h(threads / 64);
    CHECK(hipMemcpy(h.data(), dout, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0;
    for (long long v : h) mx = v > mx ? v : mx;
    // readcyclecounter = s_memtime: a constant 100 MHz clock on gfx9, so report ticks per
    // instruction relative to v_fma_f64 rather than absolute cycles
    std::printf("%-12s waves/SIMD=%d  ticks=%lld  ticks/1000 instr=%.2f  kernel %.1f us  ns per "
                "instr per SIMD=%.3f\n", name, waves_per_simd, mx,
                1000.0 * mx / (double(kIters) * per_iter), ms * 1e3,
                ms * 1e6 / (double(kIters) * per_iter * waves_per_simd));
  }
  return 0;
}

int main() {
  long long* dout = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void**>(&dout), 64 * sizeof(long long)));
#define RUN(k) if (run(#k, k, dout, kPerIter)) return 1
  RUN(k_fma); RUN(k_fma);
  RUN(k_add); RUN(k_mul); RUN(k_rndne); RUN(k_ldexp); RUN(k_rcp); RUN(k_cvt); RUN(k_cmp);
  RUN(k_mov64); RUN(k_pkfma32); RUN(k_addu); RUN(k_xor); RUN(k_cndmask); RUN(k_mullo);
  RUN(k_mulhi); RUN(k_fma32); RUN(k_mad64);
  if (run("k_fma_dep", k_fma_dep, dout, 16)) return 1;
  if (run("k_rndne_dep", k_rndne_dep, dout, 16)) return 1;
  if (run("k_ldexp_dep", k_ldexp_dep, dout, 16)) return 1;
  CHECK(hipFree(dout));
  return 0;
}
