// Reproducer for DESIGN.md section 8b, finding (1): "with hipMemsetAsync nodes at the head of a captured backbone pass, replays
// gave intermittently wrong BatchNorm statistics".  The shape of the executors' pass, reduced: memset nodes clear accumulator
// regions, a chain of kernels accumulates into them with float64 atomics (every workgroup adds 1.0 per element), a consumer kernel
// folds the accumulators into `out`.  Captured once with stream capture, replayed R times; after every replay `out` must equal
// blocks * chain exactly.  Three variants: memset node / zero-fill kernel node / memset node with the graph's memset destination
// living in a buffer that is freed and re-allocated between capture and replay (what a rebinding bug in the caller would look like).
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/graph_memset_repro.hip -o scripts/probes/graph_memset_repro.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void accumulate(double* acc, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(acc + i, 1.0);
}
__global__ void zero_fill(double* acc, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc[i] = 0.0;
}
__global__ void consume(const double* acc, float* out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (float)acc[i];
}
int run(int variant, int replays) {
  const int n = 96 * 1024, blocks = 224, chain = 40;   // 768 KB of accumulators (the sums region of ir100 is ~0.9 MB)
  hipStream_t st;
  CK(hipStreamCreate(&st));
  double* acc;
  float* out;
  CK(hipMalloc(&acc, n * sizeof(double)));
  CK(hipMalloc(&out, n * sizeof(float)));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  if (variant == 1) hipLaunchKernelGGL(zero_fill, dim3(256), dim3(256), 0, st, acc, n);
  else CK(hipMemsetAsync(acc, 0, n * sizeof(double), st));
  for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(accumulate, dim3(blocks), dim3(256), 0, st, acc, n);
  hipLaunchKernelGGL(consume, dim3(256), dim3(256), 0, st, acc, out, n);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  std::vector<float> h(n);
  int bad_replays = 0;
  for (int r = 0; r < replays; ++r) {
    // dirty the accumulators between replays (what the previous step leaves behind), on the same stream
    hipLaunchKernelGGL(accumulate, dim3(7), dim3(256), 0, st, acc, n);
    CK(hipGraphLaunch(ge, st));
    CK(hipMemcpyAsync(h.data(), out, n * sizeof(float), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i] != (float)(blocks * chain);
    if (bad) {
      if (bad_replays < 3) printf("  variant %d replay %d: %d of %d elements wrong (e.g. %g, expected %d)\n", variant, r, bad, n, h[0], blocks * chain);
      ++bad_replays;
    }
  }
  printf("variant %d (%s): %d of %d replays wrong\n", variant, variant == 1 ? "zero-fill kernel node" : "hipMemsetAsync node", bad_replays, replays);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipFree(acc));
  CK(hipFree(out));
  CK(hipStreamDestroy(st));
  return bad_replays;
}
int main() {
  int bad = 0;
  bad += run(0, 300);
  bad += run(1, 300);
  printf(bad ? "RESULT: wrong replays seen\n" : "RESULT: every replay exact (memset nodes and kernel nodes alike)\n");
  return 0;
}
