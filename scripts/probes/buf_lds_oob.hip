// Probe: does buffer_load_dwordx4 ... lds write zeros for lanes whose offset is out of the descriptor's range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const float* x, float* out, int nbytes, int soff) {
  __shared__ __attribute__((aligned(16))) float smem[256];
  smem[threadIdx.x] = -7.f; smem[threadIdx.x + 64] = -7.f; smem[threadIdx.x + 128] = -7.f; smem[threadIdx.x + 192] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  // lanes 0..31 in range, 32..47 offset 0x80000000 (out of range), 48..63 straddle / negative
  int vo = threadIdx.x * 16;
  if (threadIdx.x >= 32 && threadIdx.x < 48) vo = 0x80000000;
  if (threadIdx.x >= 48) vo = -16 * (int)(threadIdx.x - 47);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)smem, 16, vo, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = smem[threadIdx.x * 4 + i];
}
int main() {
  const int n = 64 * 4;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = i + 1;
  float *x, *o;
  hipMalloc(&x, n * 4 + 4096); hipMalloc(&o, 256 * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int soff : {0, 16}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o, 32 * 16, soff);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    printf("soffset %d num_records %d bytes\n", soff, 32 * 16);
    for (int l : {0, 1, 30, 31, 32, 40, 47, 48, 63}) printf("  lane %2d: %g %g %g %g\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
  }
  return 0;
}
