// Probe: the LDS-free reduction steps of csrc/hip_common.h (permlane swaps with a forced register copy, DPP row rotations).
#include "../../very-large-scale-face-recognition_amd/csrc/hip_common.h"
#include <cstdio>
using namespace vlsfr;
__global__ void k(float* out) {
  const float v = (float)(threadIdx.x * threadIdx.x % 97);
  out[threadIdx.x * 8 + 0] = v;
  out[threadIdx.x * 8 + 1] = lane_step_sum<32>(v);
  out[threadIdx.x * 8 + 2] = lane_step_sum<16>(v);
  out[threadIdx.x * 8 + 3] = lane_step_sum<8>(v);
  out[threadIdx.x * 8 + 4] = lane_step_sum<4>(v);
  out[threadIdx.x * 8 + 5] = lane_step_sum<2>(v);
  out[threadIdx.x * 8 + 6] = lane_step_sum<1>(v);
  out[threadIdx.x * 8 + 7] = row16_sum(v);
}
int main() {
  float* o; (void)hipMalloc(&o, 64 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float h[64 * 8]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    auto v = [&](int x) { return h[x * 8]; };
    const int row = l & ~15, i = l & 15;
    float rs = 0; for (int x = 0; x < 16; ++x) rs += v(row + x);
    const float exp[7] = {v(l) + v(l ^ 32), v(l) + v(l ^ 16), v(l) + v(row + (i + 8) % 16), v(l) + v(row + (i + 12) % 16),
                          v(l) + v(row + (i + 14) % 16), v(l) + v(row + (i + 15) % 16), rs};   // row_ror:n -> lane i reads lane (i - n) mod 16
    for (int t = 0; t < 7; ++t)
      if (h[l * 8 + 1 + t] != exp[t]) { if (bad < 12) printf("lane %d step %d: got %g expected %g\n", l, t, h[l * 8 + 1 + t], exp[t]); ++bad; }
  }
  printf("mismatches: %d\n", bad);
  return 0;
}
