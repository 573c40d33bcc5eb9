// Probe: streaming read bandwidth (read-only sum) and read+write copy, persistent grid-stride loops.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UF>
__global__ __launch_bounds__(256) void rd(const uint4* x, size_t n16, float* out) {
  float acc = 0.f;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i + (UF - 1) * step < n16; i += UF * step) {
    uint4 v[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) v[u] = x[i + u * step];
#pragma unroll
    for (int u = 0; u < UF; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].y) + __uint_as_float(v[u].z) + __uint_as_float(v[u].w);
  }
  if (acc == 123.456f) out[0] = acc;
}
// block-contiguous variant: each block streams its own contiguous chunk
template <int UF>
__global__ __launch_bounds__(256) void rd_chunk(const uint4* x, size_t n16, size_t per_block, float* out) {
  float acc = 0.f;
  const size_t b0 = (size_t)blockIdx.x * per_block;
  size_t e = b0 + per_block; if (e > n16) e = n16;
  for (size_t i = b0 + threadIdx.x; i + (UF - 1) * 256 < e; i += UF * 256) {
    uint4 v[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < UF; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].y) + __uint_as_float(v[u].z) + __uint_as_float(v[u].w);
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void cp(const uint4* x, uint4* y, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i + 3 * step < n16; i += 4 * step) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x[i + u * step];
#pragma unroll
    for (int u = 0; u < 4; ++u) y[i + u * step] = v[u];
  }
}
int main() {
  const size_t bytes = 822ull << 20;
  uint4 *x, *y; float* o;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&o, 4);
  hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](const char* name, auto launch, double traffic) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.1f us  %.2f TB/s\n", name, ms * 1e3 / 5, traffic * 5 / (ms * 1e-3) / 1e12);
  };
  const size_t n16 = bytes / 16;
  for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
    char nm[96]; snprintf(nm, 96, "read grid-stride UF4 blocks=%d", g);
    time(nm, [&] { hipLaunchKernelGGL(rd<4>, dim3(g), dim3(256), 0, 0, x, n16, o); }, (double)bytes);
  }
  for (int g : {1024, 2048, 4096}) {
    char nm[96]; snprintf(nm, 96, "read grid-stride UF8 blocks=%d", g);
    time(nm, [&] { hipLaunchKernelGGL(rd<8>, dim3(g), dim3(256), 0, 0, x, n16, o); }, (double)bytes);
  }
  for (size_t kb : {64, 256, 1024}) {
    const size_t per = kb * 1024 / 16; const int g = (int)((n16 + per - 1) / per);
    char nm[96]; snprintf(nm, 96, "read block-chunks of %zu KB (blocks=%d) UF4", kb, g);
    time(nm, [&] { hipLaunchKernelGGL(rd_chunk<4>, dim3(g), dim3(256), 0, 0, x, n16, per, o); }, (double)bytes);
  }
  for (int g : {1024, 2048, 4096}) {
    char nm[96]; snprintf(nm, 96, "copy grid-stride blocks=%d", g);
    time(nm, [&] { hipLaunchKernelGGL(cp, dim3(g), dim3(256), 0, 0, x, y, n16); }, 2.0 * bytes);
  }
  return 0;
}
