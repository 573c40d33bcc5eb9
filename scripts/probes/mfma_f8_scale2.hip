// Probe: which operand bytes does the scale of lane (row/col 0, group gs) multiply?  One-hot operand byte at (lane 16 g, byte j) of
// row / column 0, the other operand all ones; experiment (g, j, gs) doubles the scale byte of lane 16 gs on the one-hot side.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out, const int* hot /* [512][8] */, int side) {
  const int l = threadIdx.x, e = blockIdx.x;
  const int g = e >> 7, gs = e & 3;
  i32x8 oh, ones;
  for (int i = 0; i < 8; ++i) { oh[i] = (l == 16 * g) ? hot[e * 8 + i] : 0; ones[i] = 0x38383838; }
  const int sc = (l == 16 * gs) ? 128 : 127;
  f32x4 acc = {0, 0, 0, 0};
  if (side == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(oh, ones, acc, 0, 0, 0, sc, 0, 127);
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, oh, acc, 0, 0, 0, 127, 0, sc);
  if (l == 0) out[e] = acc[0];   // C[row 0][col 0]
}
int main() {
  static int hot[512][8];
  for (int e = 0; e < 512; ++e) { const int j = (e >> 2) & 31; for (int i = 0; i < 8; ++i) hot[e][i] = 0; hot[e][j >> 2] = 0x38 << (8 * (j & 3)); }
  float* d; int* dh; hipMalloc(&d, 512 * 4); hipMalloc(&dh, sizeof(hot));
  hipMemcpy(dh, hot, sizeof(hot), hipMemcpyHostToDevice);
  static float o[512];
  for (int side = 0; side < 2; ++side) {
    hipLaunchKernelGGL(k, dim3(512), dim3(64), 0, 0, d, dh, side);
    hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
    printf("%s side: (lane group g, byte j) -> scale lane group whose doubling doubles the product\n", side ? "B" : "A");
    for (int g = 0; g < 4; ++g) {
      printf("  g=%d: ", g);
      for (int j = 0; j < 32; ++j) {
        int hit = -1, n = 0;
        for (int gs = 0; gs < 4; ++gs) if (o[(g << 7) | (j << 2) | gs] == 2.f) { hit = gs; ++n; }
        printf("%c", n == 1 ? '0' + hit : (n == 0 ? 'x' : '?'));
      }
      printf("   j=0 values [%g %g %g %g]\n", o[(g << 7)], o[(g << 7) | 1], o[(g << 7) | 2], o[(g << 7) | 3]);
    }
  }
  return 0;
}
