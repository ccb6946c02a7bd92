// mfma_f64_probe.hip -- v_mfma_f64_16x16x4_f64 on gfx950: operand layout, issue rate, and how much VALU work runs beside it (diagnostic, not product).
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/mfma_f64_probe scripts/mfma_f64_probe.hip && /tmp/mfma_f64_probe > profiles/r04/mfma_f64_probe.txt
//
// layout:  D = A (16 x 4) B (4 x 16) + C with A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[4 r + (lane >> 4)][lane & 15] in register r -- checked
//          against a host product.
// rate:    W waves per SIMD, each `iters` passes of 8 MFMAs on 8 independent accumulators -> cycles per MFMA per SIMD.
// beside:  the same with V fp64 FMAs (independent chains) between consecutive MFMAs of a wave -> does the pair take max(matrix, vector) or the sum?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {      // A 16 x 4 row-major, B 4 x 16 row-major, D 16 x 16 row-major
  const int l = threadIdx.x;
  v4d c = {0., 0., 0., 0.};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * r + (l >> 4)) * 16 + (l & 15)] = c[r];
}

template <int V>
__global__ void __launch_bounds__(256) k_rate(double* out, unsigned long long* cyc, int iters, double x, double y) {
  v4d c[8];
  for (int i = 0; i < 8; i++) c[i] = v4d{0., 0., 0., 0.};
  double a = x + threadIdx.x * 1e-6, b = y;
  double f[8];
  for (int i = 0; i < 8; i++) f[i] = 1. + i * 1e-3;
  unsigned long long t0 = __builtin_readcyclecounter(), q0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; v++) f[(i + v) & 7] = __builtin_fma(f[(i + v) & 7], x, y);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), q1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.;
  for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * w] = t1 - t0; cyc[2 * w + 1] = q1 - q0; }
}

template <int V>
static void run_rate(int W, int iters) {
  const int blocks = 256 * W;
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CK(hipMalloc(&cyc, (size_t)blocks * 4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters / 8, 1.0000001, 1e-9);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0000001, 1e-9);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)blocks * 8);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double sc = 0., sr = 0.;
  for (size_t w = 0; w < (size_t)blocks * 4; w++) { sc += (double)h[2 * w]; sr += (double)h[2 * w + 1]; }
  const double ghz = sc / sr * 0.1;                          // shader cycles per 100 MHz tick
  const double n_mfma = (double)blocks * 4 * iters * 8;      // wave-level MFMAs
  const double per_simd = 1024. * ghz * 1e9 * (ms * 1e-3) / n_mfma;
  printf("W=%d V=%d: %.3f ms, clock %.2f GHz, %.1f cycles per MFMA per SIMD (%.1f TFLOP/s fp64 matrix%s), wave-own cycles per MFMA %.1f\n", W, V, ms, ghz, per_simd,
         n_mfma * 2048. / (ms * 1e-3) * 1e-12, V ? ", FMAs beside not counted" : "", sc / ((double)blocks * 4) / (iters * 8.));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  // layout
  std::vector<double> A(64), B(64), D(256), R(256, 0.);
  for (int i = 0; i < 64; i++) { A[i] = std::sin(1. + i); B[i] = std::cos(2. + 3 * i); }
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD;
  CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  double md = 0.;
  for (int i = 0; i < 256; i++) md = std::fmax(md, std::fabs(D[i] - R[i]));
  printf("layout A[i][k] <- lane i + 16 k, B[k][j] <- lane j + 16 k, D[4 r + lane / 16][lane %% 16] <- register r: max |D - A B| = %.3g\n", md);
  for (int W : {1, 2, 4}) run_rate<0>(W, 4000);
  for (int W : {1, 2}) { run_rate<2>(W, 4000); run_rate<4>(W, 4000); run_rate<8>(W, 4000); run_rate<12>(W, 4000); run_rate<16>(W, 4000); }
  return 0;
}
