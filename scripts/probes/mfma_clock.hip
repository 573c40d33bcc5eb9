// Probe: core clock under a chip-wide bf16 MFMA load (clock64 = shader cycles, wall_clock64 = 100 MHz).
// The cycles-per-MFMA column is NOT a clean issue-rate figure: the compiler chains some accumulators (see the ISA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(long long* out, int iters, int nacc) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){(float)out[4 + i], (float)i, (float)threadIdx.x, 1.f};   // distinct, opaque start values (no CSE of the chains)
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[3] = 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int main() {
  long long* o; (void)hipMalloc(&o, 256); (void)hipMemset(o, 0, 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wpc : {4, 8}) {           // waves per CU: 4 = one per SIMD, 8 = two per SIMD
    for (int iters : {2000, 20000, 100000}) {
      const int blocks = 256 * wpc / 4;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, 100, 16);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, iters, 16);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
      const double flops = (double)blocks * 4 * iters * 16 * 16384.0;
      printf("waves/CU %d iters %6d: %.1f us, %.0f TFLOP/s, core clock %.0f MHz (cycles %lld / wall %.1f us), cycles per MFMA per SIMD %.2f\n",
             wpc, iters, ms * 1e3, flops / (ms * 1e-3) / 1e12, h[0] / (h[1] / 100.0), h[0], h[1] / 100.0,
             (double)h[0] / ((double)iters * 16 * (wpc / 4)));
    }
  }
  return 0;
}
