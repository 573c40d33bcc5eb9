// Probe: which outputs does the E8M0 scale of ONE lane touch in v_mfma_scale_f32_16x16x128_f8f6f4?  A = B = all 1.0 (C = 128 with
// unit scales); experiment e doubles the scale byte of lane e (A side: e < 64, B side: e - 64) and reports the changed rows / columns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int OPSEL>
__global__ void k(f32x4* c, int byte_pos) {
  const int l = threadIdx.x, e = blockIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
  if (e < 64 && l == e) sa = (sa & ~(0xff << (8 * byte_pos))) | (0x80 << (8 * byte_pos));
  if (e >= 64 && l == e - 64) sb = (sb & ~(0xff << (8 * byte_pos))) | (0x80 << (8 * byte_pos));
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, OPSEL, sa, OPSEL, sb);
  c[e * 64 + l] = acc;
}
int main() {
  f32x4* dc; hipMalloc(&dc, 128 * 64 * 16);
  static float C[128][64][4];
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int byte_pos = cfg == 0 ? 0 : 1, opsel = cfg == 2 ? 1 : 0;
    if (opsel == 0) hipLaunchKernelGGL(k<0>, dim3(128), dim3(64), 0, 0, dc, byte_pos);
    else hipLaunchKernelGGL(k<1>, dim3(128), dim3(64), 0, 0, dc, byte_pos);
    hipMemcpy(C, dc, sizeof(C), hipMemcpyDeviceToHost);
    printf("--- scale byte position %d, op_sel %d\n", byte_pos, opsel);
    for (int e : {0, 1, 5, 16, 17, 33, 63, 64, 65, 80, 127}) {
      // changed entries: C layout col = l&15, row = 4 (l>>4) + r
      int nchg = 0; float val = 0; unsigned rows = 0, cols = 0;
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (C[e][l][r] != 128.f) { ++nchg; val = C[e][l][r]; rows |= 1u << (4 * (l >> 4) + r); cols |= 1u << (l & 15); }
      printf("  %s lane %2d (l&15 = %2d, l>>4 = %d): %3d outputs changed to %g; rows mask %04x cols mask %04x\n", e < 64 ? "A" : "B", e & 63, e & 15, (e & 63) >> 4,
             nchg, val, rows, cols);
    }
  }
  return 0;
}
