// Device kernels only the torchvision-style ResNet (reference model/resnet_std.py, `--net_type r50`) needs, besides
// the convolution / BatchNorm kernels it shares with the iResNet path: the 7x7 stride-2 stem's im2col, the 3x3
// stride-2 max-pool (forward and backward) and the backward of the ReLU that follows the residual add.
// Activations NHWC bf16.  All HBM-bound elementwise / small-window kernels.
#include "hip_common.h"

using namespace vlsfr;

namespace {

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ u16 f2bf_rne(float f) {
  const __bf16 b = (__bf16)f;   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(u16, b);
}

inline int grid_for(int64_t items, int cap = 4096) {
  int64_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// fp32 NCHW image [N,3,H,W] -> bf16 rows [N*Ho*Wo][160] of the 7x7 / stride 2 / pad 3 stem (resnet_std.py:127-128),
// k = (r*7 + s)*3 + c for the 147 real taps, zero up to 160.  One thread per (pixel, filter row r): 21 values + pad.
__global__ __launch_bounds__(256) void stem7_im2col_kernel(const float* x, u16* out, int N, int H, int W) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Wo * 8;   // 7 filter rows + one slice for the padding columns
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i & 7);
    const int64_t p = i >> 3;
    const int n = (int)(p / (Ho * Wo));
    const int rem = (int)(p - (int64_t)n * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    u16* dst = out + p * 160;
    if (r == 7) {
      for (int k = 147; k < 160; ++k) dst[k] = 0;
      continue;
    }
    const int hi = ho * 2 + r - 3;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const int wi = wo * 2 + s - 3;
      const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        dst[(r * 7 + s) * 3 + c] = ok ? f2bf_rne(x[(((size_t)n * 3 + c) * H + hi) * W + wi]) : (u16)0;
    }
  }
}

// y[n,ho,wo,c] = max over the 3x3 window (stride 2, pad 1; padding never wins: nn.MaxPool2d pads with -inf)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const u16* x, u16* y, int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, C8 = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 + r - 1;
      if (hi < 0 || hi >= H) continue;
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 + s - 1;
        if (wi < 0 || wi >= W) continue;
        const uint4 v = *(const uint4*)(x + (((size_t)n * H + hi) * W + wi) * C + c8 * 8);
        const u16* e = (const u16*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], bf2f(e[j]));
      }
    }
    uint4 o;
    u16* oe = (u16*)&o;
#pragma unroll
    for (int j = 0; j < 8; ++j) oe[j] = f2bf_rne(m[j]);
    *(uint4*)(y + (((size_t)n * Ho + ho) * Wo + wo) * C + c8 * 8) = o;
  }
}

// dx[n,h,w,c] = sum of dy over the windows whose FIRST maximum (row-major scan of the window, the element
// torch's max_pool2d backward credits) is (h, w).  Gather form: every input position visits the <= 4 windows that
// contain it; no atomics, deterministic.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const u16* dy, const u16* x, const u16* y, u16* dx, int N, int H, int W,
                                                          int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    const float xv = bf2f(x[i]);   // compared as numbers: -0 (what the zero-slope PReLU leaves for negative inputs) ties with +0
    float g = 0.f;
    // windows (ho, wo) with ho*2 - 1 <= h <= ho*2 + 1
    for (int ho = h / 2; ho <= (h + 1) / 2 && ho < Ho; ++ho) {
      if (ho * 2 - 1 > h || ho * 2 + 1 < h) continue;
      for (int wo = w / 2; wo <= (w + 1) / 2 && wo < Wo; ++wo) {
        if (wo * 2 - 1 > w || wo * 2 + 1 < w) continue;
        const size_t yo = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
        if (bf2f(y[yo]) != xv) continue;   // the pooled value is one of the window's elements
        // is (h, w) the first element of the window equal to the maximum?
        bool first = true;
        for (int r = 0; r < 3 && first; ++r) {
          const int hi = ho * 2 + r - 1;
          if (hi < 0 || hi >= H) continue;
          for (int s = 0; s < 3; ++s) {
            const int wi = wo * 2 + s - 1;
            if (wi < 0 || wi >= W) continue;
            if (hi == h && wi == w) {
              r = 3;   // reached ourselves without meeting an earlier maximum
              break;
            }
            if (bf2f(x[(((size_t)n * H + hi) * W + wi) * C + c]) == xv) {
              first = false;
              break;
            }
          }
        }
        if (first) g += bf2f(dy[yo]);
      }
    }
    dx[i] = f2bf_rne(g);
  }
}

// dx = dy where y > 0 else 0 (ReLU after the residual add; y is the block output).  in_nchw: dy and y are in the
// [n][c][hw] flatten order (the last block, whose output feeds fc), dx is written NHWC.
__global__ __launch_bounds__(256) void relu_bwd_kernel(const u16* dy, const u16* y, u16* dx, int64_t total, int C, int HW,
                                                       int in_nchw) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t src = i;
    if (in_nchw) {
      const int c = (int)(i % C);
      const int64_t row = i / C;
      const int64_t n = row / HW;
      const int hw = (int)(row - n * HW);
      src = (n * C + c) * HW + hw;
    }
    dx[i] = bf2f(y[src]) > 0.f ? dy[src] : (u16)0;
  }
}

}  // namespace

extern "C" {

int vlsfr_stem7_im2col(const float* x_nchw, void* out, int32_t N, int32_t H, int32_t W, void* stream) {
  if (!x_nchw || !out || N <= 0 || H < 7 || W < 7) return fail(VLSFR_EINVAL, "vlsfr_stem7_im2col: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(stem7_im2col_kernel, dim3(grid_for((int64_t)N * Ho * Wo * 8)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                     (u16*)out, N, H, W);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_stem7_im2col");
  return VLSFR_OK;
}

int vlsfr_maxpool3x3s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return fail(VLSFR_EINVAL, "vlsfr_maxpool3x3s2_fwd: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)x, (u16*)y, N, H, W, C);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_maxpool3x3s2_fwd");
  return VLSFR_OK;
}

int vlsfr_maxpool3x3s2_bwd(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                           void* stream) {
  if (!dy || !x || !y || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_maxpool3x3s2_bwd: bad argument");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)dy, (const u16*)x, (const u16*)y, (u16*)dx, N, H, W, C);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_maxpool3x3s2_bwd");
  return VLSFR_OK;
}

int vlsfr_relu_bwd_bf16(const void* dy, const void* y, void* dx, int64_t M, int32_t C, int32_t HW, int32_t in_nchw,
                        void* stream) {
  if (!dy || !y || !dx || M <= 0 || C <= 0 || HW <= 0) return fail(VLSFR_EINVAL, "vlsfr_relu_bwd_bf16: bad argument");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(M * C, 8192)), dim3(256), 0, (hipStream_t)stream, (const u16*)dy,
                     (const u16*)y, (u16*)dx, M * C, C, HW, in_nchw);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_relu_bwd_bf16");
  return VLSFR_OK;
}

}  // extern "C"
