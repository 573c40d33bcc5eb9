// Probe: LDS read instruction rates (ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16) alone and beside a stream of
// independent v_mfma_f32_16x16x32_bf16 — which LDS load per MFMA still lets the matrix pipes run at their rate?
// One workgroup per CU (64 KB of LDS), NW waves; every wave loops N times over: issue NREAD reads, NMFMA MFMAs, lgkmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

template <int KIND, int NREAD, int NMFMA>
__global__ __launch_bounds__(512) void k(float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)smem)[i] = 1.0f;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint32_t a128 = base + ((wave & 3) * 8192) + lane * 16, a64 = base + ((wave & 3) * 8192) + lane * 8;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(0.001f * (lane + e)); fb[e] = (__bf16)(0.002f * (lane - e)); }
  int sink = 0;
  for (int it = 0; it < n; ++it) {
    i32x4 v[NREAD > 0 ? NREAD : 1];
#pragma unroll
    for (int r = 0; r < NREAD; ++r) {
      if constexpr (KIND == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(a128), "n"((r & 7) * 1024));
      else if constexpr (KIND == 1) { i32x2 t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(a64), "n"((r & 15) * 512)); v[r][0] = t[0]; v[r][1] = t[1]; }
      else { i32x2 t; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t) : "v"(a64), "n"((r & 15) * 512)); v[r][0] = t[0]; v[r][1] = t[1]; }
    }
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 15], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < NREAD; ++r) asm volatile("" ::"v"(v[r][0]), "v"(v[r][1]));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f || sink == 77) out[0] = s;
}

template <int KIND, int NREAD, int NMFMA>
void run(const char* name, int nw, float* out) {
  auto kern = k<KIND, NREAD, NMFMA>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int n = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 65536, 0, out, 200);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 65536, 0, out, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes_per_read = KIND == 0 ? 1024.0 : 512.0;
  const double lds = (double)n * NREAD * bytes_per_read * nw / (ms * 1e-3);            // bytes/s per CU
  const double tf = (double)n * NMFMA * nw * 256 * 16384.0 / (ms * 1e-3) / 1e12;       // chip TFLOP/s
  printf("%-34s waves %d reads %2d mfma %2d : %8.3f ms | LDS %7.1f GB/s per CU (%.1f B/clk @2.4GHz) | MFMA %7.1f TFLOP/s (%.2f of 2500)\n", name, nw, NREAD, NMFMA, ms,
         lds / 1e9, lds / 2.4e9, tf, tf / 2500.0);
}

int main() {
  float* out; hipMalloc(&out, 4);
  for (int nw : {4, 8}) {
    run<0, 16, 0>("ds_read_b128 only", nw, out);
    run<1, 16, 0>("ds_read_b64 only", nw, out);
    run<2, 16, 0>("ds_read_b64_tr_b16 only", nw, out);
    run<0, 0, 16>("mfma only", nw, out);
    run<0, 4, 16>("b128 0.25 frag/mfma", nw, out);
    run<0, 8, 16>("b128 0.5 frag/mfma (conv 128x128)", nw, out);
    run<0, 12, 16>("b128 0.75 frag/mfma", nw, out);
    run<0, 16, 16>("b128 1 frag/mfma", nw, out);
    run<2, 16, 16>("tr_b64 0.5 frag/mfma", nw, out);
    run<2, 32, 16>("tr_b64 1 frag/mfma (wgrad, head O)", nw, out);
    run<1, 32, 16>("b64 1 frag/mfma", nw, out);
  }
  return 0;
}
