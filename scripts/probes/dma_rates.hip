// Probe: LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) throughput per CU, alone and beside the
// convolution kernel's mix (per 16 MFMAs: 8 ds_read_b128 + 4 DMA instructions = the 128x128x64 tile on 4 waves).
// Sources: FOOT bytes per workgroup region of a global buffer (small = L1/L2-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void lds_void_t;

template <int NDMA, int NREAD, int NMFMA>
__global__ __launch_bounds__(256, 2) void k(const char* src, size_t foot, float* out, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float*)smem)[i] = 1.0f;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const uint32_t a128 = base + 32768 + wave * 4096 + lane * 16;      // fragment reads: upper half of the 64 KB
  const char* reg = src + (size_t)(blockIdx.x % 256) * foot;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)reg, 0, (int)foot, 0x00020000);
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(0.001f * (lane + e)); fb[e] = (__bf16)(0.002f * (lane - e)); }
  int off = (wave * 1024 + lane * 16);
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + (wave * NDMA + d) * 1024), 16, off, 0, 0, 0);
      off += 4096;
      if (off >= (int)foot) off -= (int)foot;
    }
    i32x4 v[NREAD > 0 ? NREAD : 1];
#pragma unroll
    for (int r = 0; r < NREAD; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a128), "n"((r & 3) * 1024));
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 15], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");      // one iteration of DMAs stays in flight
#pragma unroll
    for (int r = 0; r < NREAD; ++r) asm volatile("" ::"v"(v[r][0]), "v"(v[r][1]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) out[0] = s;
#endif
}

template <int NDMA, int NREAD, int NMFMA>
void run(const char* name, int wgs_per_cu, size_t foot, const char* src, float* out) {
  auto kern = k<NDMA, NREAD, NMFMA>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int n = 10000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 65536, 0, src, foot, out, 200);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 65536, 0, src, foot, out, n);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double dma = (double)n * NDMA * 1024.0 * 4 * wgs_per_cu / (ms * 1e-3);
  const double tf = (double)n * NMFMA * 4 * wgs_per_cu * 256 * 16384.0 / (ms * 1e-3) / 1e12;
  printf("%-40s WG/CU %d foot %7zu : %8.3f ms | DMA %6.1f GB/s per CU (%5.1f B/clk @2.4GHz, chip %5.2f TB/s) | MFMA %7.1f TFLOP/s (%.2f)\n", name, wgs_per_cu,
         foot, ms, dma / 1e9, dma / 2.4e9, dma * 256 / 1e12, tf, tf / 2500.0);
}

int main() {
  char* src; float* out;
  const size_t total = 256ull << 20;
  (void)hipMalloc(&src, total); (void)hipMemset(src, 1, total); (void)hipMalloc(&out, 4);
  for (int w : {1, 2}) {
    for (size_t foot : {(size_t)16384, (size_t)131072, (size_t)1048576}) {
      run<4, 0, 0>("DMA only (4 per wave-iteration)", w, foot, src, out);
      run<8, 0, 0>("DMA only (8 per wave-iteration)", w, foot, src, out);
    }
    run<0, 8, 16>("reads + mfma (0.5 frag/mfma)", w, 131072, src, out);
    run<4, 8, 16>("conv mix: 4 DMA + 8 reads + 16 mfma", w, 16384, src, out);
    run<4, 8, 16>("conv mix: 4 DMA + 8 reads + 16 mfma", w, 131072, src, out);
    run<4, 8, 16>("conv mix: 4 DMA + 8 reads + 16 mfma", w, 1048576, src, out);
    run<2, 8, 16>("half DMA: 2 DMA + 8 reads + 16 mfma", w, 131072, src, out);
    run<4, 0, 16>("4 DMA + 16 mfma", w, 131072, src, out);
  }
  return 0;
}
