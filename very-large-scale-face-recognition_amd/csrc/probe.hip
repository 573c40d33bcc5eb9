// Hardware-convention probes: tiny kernels that pin the MFMA fragment maps and the transposed LDS
// read this library's kernels rely on (tests/test_probe_gpu.py checks them with exact integers).
#include "hip_common.h"

using namespace vlsfr;

namespace {

// C[16x16] = A[16x32] * B[32x16].  B sits in LDS row-major [k][n] (row stride rs bytes) and is read
// with ds_read_b64_tr_b16 exactly as head_sweep's second product does; the k index is permuted
// (element q of lane group h <-> k = 16 (q >> 2) + 4h + (q & 3)) on both operands.
__global__ void probe_mfma_tr_kernel(const float* A, const float* B, float* C, int rs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int r16 = lane & 15, h = lane >> 4;
  for (int e = lane; e < 32 * 16; e += 64) {
    const int k = e / 16, n = e % 16;
    *(__bf16*)(smem + k * rs + n * 2) = (__bf16)B[k * 16 + n];
  }
  __syncthreads();
  bf16x8 a;
#pragma unroll
  for (int q = 0; q < 8; ++q) a[q] = (__bf16)A[r16 * 32 + 16 * (q >> 2) + 4 * h + (q & 3)];
  const int trow0 = 4 * h + (r16 >> 2);
  const int tsub = r16 & 3;
  short4v b0 = lds_read_tr16(smem + trow0 * rs + tsub * 8);
  short4v b1 = lds_read_tr16(smem + (trow0 + 16) * rs + tsub * 8);
  short __attribute__((ext_vector_type(8))) bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  bf16x8 b = __builtin_bit_cast(bf16x8, bs);
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = mfma16(a, b, c);
#pragma unroll
  for (int e = 0; e < 4; ++e) C[(4 * h + e) * 16 + r16] = c[e];
}

// C[16x16] = A[16x32] * Bt[16x32]^T with both operands in the natural k order (lane holds 8
// consecutive k): the first product of head_sweep and the GEMM kernels.
__global__ void probe_mfma_nat_kernel(const float* A, const float* Bt, float* C) {
  const int lane = threadIdx.x;
  const int r16 = lane & 15, h = lane >> 4;
  bf16x8 a, b;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a[q] = (__bf16)A[r16 * 32 + 8 * h + q];
    b[q] = (__bf16)Bt[r16 * 32 + 8 * h + q];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = mfma16(a, b, c);
#pragma unroll
  for (int e = 0; e < 4; ++e) C[(4 * h + e) * 16 + r16] = c[e];
}

}  // namespace

extern "C" {

int vlsfr_probe_mfma_tr(const float* A, const float* B, float* C, int32_t row_stride_bytes, void* stream) {
  if (!A || !B || !C || row_stride_bytes < 32 || row_stride_bytes % 8)
    return fail(VLSFR_EINVAL, "vlsfr_probe_mfma_tr: bad argument");
  hipLaunchKernelGGL(probe_mfma_tr_kernel, dim3(1), dim3(64), 32 * row_stride_bytes, (hipStream_t)stream, A, B, C,
                     row_stride_bytes);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_probe_mfma_tr");
  return VLSFR_OK;
}

int vlsfr_probe_mfma_nat(const float* A, const float* Bt, float* C, void* stream) {
  if (!A || !Bt || !C) return fail(VLSFR_EINVAL, "vlsfr_probe_mfma_nat: bad argument");
  hipLaunchKernelGGL(probe_mfma_nat_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bt, C);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_probe_mfma_nat");
  return VLSFR_OK;
}
}
