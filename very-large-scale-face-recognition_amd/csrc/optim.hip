// Parameter sweeps of the training step on gfx950 (C-ABI section 8 of include/vlsfr.h): one launch
// over every parameter tensor ("multi-tensor"), 16-byte accesses, HBM-bound.
//   vlsfr_sgd_nesterov — torch.optim.SGD(momentum, weight_decay, nesterov) update, which the reference
//                        builds at optim/optimizer.py:148-150 from config/optim_config:9-13
//   vlsfr_ema          — gallery <- m * gallery + (1 - m) * probe, reference ffc.py:139-145
// The chunk table lives in device memory: one row per <= 64K-element chunk holding the (already
// offset) tensor addresses and the element count.
#include "hip_common.h"

using namespace vlsfr;

namespace {

__global__ __launch_bounds__(256) void sgd_kernel(const int64_t* table, float lr, float mu, float wd, int nesterov) {
  const int64_t* row = table + (size_t)blockIdx.x * 4;
  float* p = (float*)row[0];
  const float* g = (const float*)row[1];
  float* buf = (float*)row[2];
  const int n = (int)row[3];
  const int n4 = n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], bv = ((f32x4*)buf)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = gv[j] + wd * pv[j];
      bv[j] = mu * bv[j] + d;
      pv[j] -= lr * (nesterov ? d + mu * bv[j] : bv[j]);
    }
    ((f32x4*)p)[i] = pv;
    ((f32x4*)buf)[i] = bv;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
    const float d = g[i] + wd * p[i];
    const float b = mu * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * (nesterov ? d + mu * b : b);
  }
}

__global__ __launch_bounds__(256) void ema_kernel(const int64_t* table, float m) {
  const int64_t* row = table + (size_t)blockIdx.x * 3;
  float* e = (float*)row[0];
  const float* p = (const float*)row[1];
  const int n = (int)row[2];
  const int n4 = n >> 2;
  const float om = 1.f - m;
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 ev = ((f32x4*)e)[i], pv = ((const f32x4*)p)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) ev[j] = ev[j] * m + pv[j] * om;
    ((f32x4*)e)[i] = ev;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) e[i] = e[i] * m + p[i] * om;
}

// r <- (1 - m)^2 r + (1 - m) d0 + d1: the two running-statistics updates of a step applied in order, from the
// contributions d_k = m * (batch statistic of pass k) that the two concurrent forward passes left in their own buffers
__global__ __launch_bounds__(256) void running_merge_kernel(const int64_t* table, float m) {
  const int64_t* row = table + (size_t)blockIdx.x * 4;
  float* r = (float*)row[0];
  const float* d0 = (const float*)row[1];
  const float* d1 = (const float*)row[2];
  const int n = (int)row[3];
  const float om = 1.f - m;
  for (int i = threadIdx.x; i < n; i += 256) r[i] = om * (om * r[i] + d0[i]) + d1[i];
}

}  // namespace

extern "C" {

int vlsfr_sgd_nesterov(const int64_t* table_dev, int32_t n_chunks, float lr, float momentum, float weight_decay,
                       int32_t nesterov, void* stream) {
  if (!table_dev || n_chunks < 0) return fail(VLSFR_EINVAL, "vlsfr_sgd_nesterov: bad argument");
  if (n_chunks == 0) return VLSFR_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, lr, momentum,
                     weight_decay, nesterov);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_sgd_nesterov");
  return VLSFR_OK;
}

int vlsfr_running_merge(const int64_t* table_dev, int32_t n_chunks, float momentum, void* stream) {
  if (!table_dev || n_chunks < 0) return fail(VLSFR_EINVAL, "vlsfr_running_merge: bad argument");
  if (n_chunks == 0) return VLSFR_OK;
  hipLaunchKernelGGL(running_merge_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, momentum);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_running_merge");
  return VLSFR_OK;
}

int vlsfr_ema(const int64_t* table_dev, int32_t n_chunks, float m, void* stream) {
  if (!table_dev || n_chunks < 0) return fail(VLSFR_EINVAL, "vlsfr_ema: bad argument");
  if (n_chunks == 0) return VLSFR_OK;
  hipLaunchKernelGGL(ema_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, m);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_ema");
  return VLSFR_OK;
}
}
