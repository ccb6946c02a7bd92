// march_probe.hip -- how fast does the power-sum march of k_full_kde_chain run ALONE on the card? (diagnostic, not product)
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/march_probe scripts/march_probe.hip && /tmp/march_probe > profiles/r05/march_probe.txt
//
// The kernel is the inner loop of k_full_kde_chain (chm_kernels.h) and nothing else: a thread keeps SPT (pw, u) pairs in registers and walks
// `chunks` chunks of 32 grid points, 5 fp64 instructions per 4 (sample, grid point) pairs + 5 per (sample, chunk); the 32 sums of a chunk are
// folded into one register (32 adds, where the product kernel has its cross-lane exchange).  Launched with W blocks of 256 threads per CU
// (one wave per SIMD each), W = 1, 2, 3: what the march sustains with 1, 2, 3 waves per SIMD, the clock it holds, and the same numbers for a
// stream of independent v_fma_f64 (the rate profiles/r03/issue_cost.txt lists for W = 4 and 8).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LK = 32;

template <int SPT>
__global__ void __launch_bounds__(256, 3) k_march(unsigned long long* out, double* sink, int chunks, double K1, double K2) {
  double pw[SPT], uu[SPT];
#pragma unroll
  for (int j = 0; j < SPT; j++) { pw[j] = 1e-3 * (threadIdx.x + j + 1); uu[j] = 1. - 1e-6 * (threadIdx.x + 3 * j + 1); }
  double tot = 0.;
  const unsigned long long t0 = __builtin_readcyclecounter(), q0 = __builtin_amdgcn_s_memrealtime();
  for (int c = 0; c < chunks; c++) {
    double acc[LK];
#pragma unroll
    for (int j = 0; j < SPT; j++) {
      double q = pw[j];
      const double u = uu[j];
      const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2;
#pragma unroll
      for (int i = 0; i < LK; i += 4) {
        if (j == 0) { acc[i] = q; acc[i + 1] = q * u; acc[i + 2] = q * u2; acc[i + 3] = q * u3; }
        else {
          acc[i] += q;
          acc[i + 1] = __builtin_fma(q, u, acc[i + 1]);
          acc[i + 2] = __builtin_fma(q, u2, acc[i + 2]);
          acc[i + 3] = __builtin_fma(q, u3, acc[i + 3]);
        }
        q *= u4;
      }
      pw[j] = q * K1; uu[j] = u * K2;
      __builtin_amdgcn_sched_barrier(0);
    }
    double v = 0.;
#pragma unroll
    for (int i = 0; i < LK; i++) v += acc[i];
    tot += v;
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), q1 = __builtin_amdgcn_s_memrealtime();
  if (tot == 1.2345e-300) sink[threadIdx.x] = tot;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[2 * w] = t1 - t0; out[2 * w + 1] = q1 - q0;
  }
}

__global__ void __launch_bounds__(256, 3) k_fma(unsigned long long* out, double* sink, int iters, double c1, double c2) {
  double d[16];
#pragma unroll
  for (int j = 0; j < 16; j++) d[j] = 1e-3 * (threadIdx.x + j);
  const unsigned long long t0 = __builtin_readcyclecounter(), q0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int j = 0; j < 16; j++) d[j] = __builtin_fma(d[j], c1, c2);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), q1 = __builtin_amdgcn_s_memrealtime();
  double tot = 0.;
#pragma unroll
  for (int j = 0; j < 16; j++) tot += d[j];
  if (tot == 1.2345e-300) sink[threadIdx.x] = tot;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[2 * w] = t1 - t0; out[2 * w + 1] = q1 - q0;
  }
}

int main(int argc, char** argv) {
  const int chunks = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s, %d CUs; W blocks of 256 threads per CU = W waves per SIMD; %d chunks of 32 grid points per thread\n", prop.name, cus, chunks);
  printf("%-34s %2s %9s %10s %7s %14s %12s\n", "kernel", "W", "ms", "Ginst/s", "GHz", "cyc/inst/SIMD", "Tpair/s");
  unsigned long long* d;
  double* sink;
  const int maxw = cus * 4 * 4;
  CK(hipMalloc(&d, 2 * maxw * sizeof(unsigned long long)));
  CK(hipMalloc(&sink, 256 * sizeof(double)));
  std::vector<unsigned long long> h(2 * maxw);
  std::vector<double> clk(maxw);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, int W, float ms, double inst_per_wave, double pairs_per_lane) {
    const int nw = cus * W * 4;
    CK(hipMemcpy(h.data(), d, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < nw; i++) clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.;
    std::sort(clk.begin(), clk.begin() + nw);
    const double ghz = clk[nw / 2], ginst = nw * inst_per_wave / (ms * 1e-3) / 1e9;
    printf("%-34s %2d %9.3f %10.1f %7.3f %14.3f %12.2f\n", name, W, ms, ginst, ghz, cus * 4 * ghz / ginst, nw * 64. * pairs_per_lane / (ms * 1e-3) / 1e12);
    fflush(stdout);
  };
  for (int W : {1, 2, 3}) {
    float ms;
    // (sample, chunk): 3 + 8 x 5 + 2 = 45 fp64 instructions; + 32 adds per chunk
    hipLaunchKernelGGL(k_march<16>, dim3(cus * W), dim3(256), 0, 0, d, sink, chunks / 10 + 1, 0.999, 0.9999); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_march<16>, dim3(cus * W), dim3(256), 0, 0, d, sink, chunks, 0.999, 0.9999);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    report("march, 16 samples x 32 points", W, ms, (double)chunks * (16 * 45 + 32), (double)chunks * 16 * 32);
  }
  for (int W : {1, 2, 3}) {
    float ms;
    const int iters = chunks * 12;
    hipLaunchKernelGGL(k_fma, dim3(cus * W), dim3(256), 0, 0, d, sink, iters / 10 + 1, 1.0000001, 1e-9); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fma, dim3(cus * W), dim3(256), 0, 0, d, sink, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    report("v_fma_f64, 16 independent chains", W, ms, (double)iters * 64, 0.);
  }
  return 0;
}
