// Probe: where does a one-hot A element land?  B all ones, unit scales; prints the nonzero outputs for a few (lane, byte) positions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(f32x4* out, const int* pos) {
  const int l = threadIdx.x, e = blockIdx.x;
  const int pl = pos[2 * e], pj = pos[2 * e + 1];
  int av[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (l == pl) av[pj >> 2] = 0x38 << (8 * (pj & 3));
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = av[i]; b[i] = 0x38383838; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 127, 0, 127);
  out[e * 64 + l] = acc;
}
int main() {
  const int P[][2] = {{0, 0}, {0, 15}, {0, 16}, {0, 20}, {0, 31}, {16, 0}, {16, 20}, {32, 0}, {32, 20}, {48, 0}, {48, 31}, {5, 3}, {37, 19}};
  const int n = sizeof(P) / sizeof(P[0]);
  int* dp; f32x4* d; hipMalloc(&dp, sizeof(P)); hipMalloc(&d, n * 64 * 16);
  hipMemcpy(dp, P, sizeof(P), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d, dp);
  static float C[32][64][4];
  hipMemcpy(C, d, n * 64 * 16, hipMemcpyDeviceToHost);
  for (int e = 0; e < n; ++e) {
    int cnt = 0; unsigned rows = 0, cols = 0; float v = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (C[e][l][r] != 0.f) { ++cnt; rows |= 1u << (4 * (l >> 4) + r); cols |= 1u << (l & 15); v = C[e][l][r]; }
    printf("A one-hot lane %2d byte %2d: %d nonzero outputs (value %g), rows %04x cols %04x\n", P[e][0], P[e][1], cnt, v, rows, cols);
  }
  return 0;
}
