#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  unsigned* o; hipMalloc(&o, 64 * 2 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  unsigned h[128]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 5, 16, 21, 32, 37, 48, 53}) printf("lane %2d: r0=%u r1=%u\n", l, h[l * 2], h[l * 2 + 1]);
  return 0;
}
