// Probe: v_permlane16_swap / v_permlane32_swap semantics and the swap(v, v) pair-sum idiom.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  float v = (float)threadIdx.x;
  auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  float sum32 = __builtin_bit_cast(float, s[0]) + __builtin_bit_cast(float, s[1]);
  auto t = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  float sum16 = __builtin_bit_cast(float, t[0]) + __builtin_bit_cast(float, t[1]);
  out[threadIdx.x * 6 + 0] = r[0]; out[threadIdx.x * 6 + 1] = r[1];
  out[threadIdx.x * 6 + 2] = q[0]; out[threadIdx.x * 6 + 3] = q[1];
  out[threadIdx.x * 6 + 4] = (unsigned)sum32; out[threadIdx.x * 6 + 5] = (unsigned)sum16;
}
int main() {
  unsigned* o; (void)hipMalloc(&o, 64 * 6 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  unsigned h[64 * 6]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 5, 16, 21, 32, 37, 48, 53})
    printf("lane %2d: swap16 r0=%u r1=%u | swap32 r0=%u r1=%u | v+v^32=%u (expect %d) v+v^16=%u (expect %d)\n", l, h[l * 6], h[l * 6 + 1],
           h[l * 6 + 2], h[l * 6 + 3], h[l * 6 + 4], l + (l ^ 32), h[l * 6 + 5], l + (l ^ 16));
  return 0;
}
