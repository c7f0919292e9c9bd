// Latency microbenchmarks on one MI355X CU (one block of 256 threads): dependent LDS reads, dependent
// global (L2-resident) reads, barriers, s_memtime itself.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_lat(const int* __restrict__ gidx, long long* out, int iters) {
  __shared__ int lidx[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lidx[i] = gidx[i];
  __syncthreads();
  long long t0, t1;
  // (a) s_memtime back to back
  t0 = clock64();
  t1 = clock64();
  long long dt_clock = t1 - t0;
  // (b) dependent LDS chain
  int p = threadIdx.x;
  t0 = clock64();
  for (int i = 0; i < iters; ++i) p = lidx[p];
  t1 = clock64();
  long long dt_lds = t1 - t0;
  // (c) dependent global chain (small array: L1/L2 resident)
  int q = threadIdx.x;
  t0 = clock64();
  for (int i = 0; i < iters; ++i) q = gidx[q];
  t1 = clock64();
  long long dt_glb = t1 - t0;
  // (d) barriers
  t0 = clock64();
  for (int i = 0; i < iters; ++i) __syncthreads();
  t1 = clock64();
  long long dt_bar = t1 - t0;
  // (e) barrier + one LDS write/read exchange per iteration
  __shared__ double xch[256];
  double v = threadIdx.x;
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    xch[threadIdx.x] = v;
    __syncthreads();
    v += xch[(threadIdx.x + 64) & 255];
    __syncthreads();
  }
  t1 = clock64();
  long long dt_xch = t1 - t0;
  // (f) dependent FP64 fma chain
  double f = v;
  t0 = clock64();
  for (int i = 0; i < iters; ++i) f = fma(f, 1.0000001, 0.5);
  t1 = clock64();
  long long dt_fma = t1 - t0;
  if (threadIdx.x == 0) {
    out[0] = dt_clock; out[1] = dt_lds; out[2] = dt_glb; out[3] = dt_bar; out[4] = dt_xch; out[5] = dt_fma;
    out[6] = p + q + (long long)v + (long long)f;
  }
}

int main() {
  const int iters = 1000;
  std::vector<int> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (i * 97 + 13) & 4095;
  int* d; long long* o;
  hipMalloc(&d, 4096 * 4); hipMalloc(&o, 64);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(256), 0, 0, d, o, iters);
    hipDeviceSynchronize();
  }
  long long r[8];
  hipMemcpy(r, o, 56, hipMemcpyDeviceToHost);
  printf("clock64 pair %lld ticks; per iteration: LDS dep read %.1f, global dep read %.1f, barrier %.1f, "
         "LDS exchange + 2 barriers %.1f, dependent fma %.1f\n", r[0], r[1] / (double)iters,
         r[2] / (double)iters, r[3] / (double)iters, r[4] / (double)iters, r[5] / (double)iters);
  return 0;
}
