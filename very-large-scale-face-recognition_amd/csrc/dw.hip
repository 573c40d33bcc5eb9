// Depthwise convolutions of MobileFaceNet on gfx950 (C-ABI section 5b of include/vlsfr.h): replaces
// nn.Conv2d(groups = channels) of reference model/mobilefacenet_def.py:39 (3x3, stride 1/2, pad 1)
// and :60,88 (7x7 "global" depthwise, valid), forward, input gradient and weight gradient.
// These layers are ~4 % of the MACs and have no channel contraction, so they are HBM-bound VALU
// kernels (not MFMA): NHWC bf16 activations, one thread = 8 channels (16 B) of one output pixel,
// fp32 weights [C][k*k] read as they lie in the parameter.  The forward kernel also accumulates the
// BatchNorm statistics of its output (same replicated accumulators as the dense kernels).
#include "hip_common.h"

using namespace vlsfr;

namespace {

constexpr int REPL = VLSFR_BN_REPL;

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  __bf16 x = (__bf16)a, y = (__bf16)b;
  return (uint32_t)__builtin_bit_cast(u16, x) | ((uint32_t)__builtin_bit_cast(u16, y) << 16);
}

struct DwArgs {
  const u16* in;     // forward: x [N,H,W,C]; dgrad: dy [N,Ho,Wo,C]
  const float* w;    // [C][k*k]
  u16* out;          // forward: y [N,Ho,Wo,C]; dgrad: dx [N,H,W,C]
  int N, H, W, C, Ho, Wo, k, stride, pad;
  int dgrad;         // 0 forward, 1 input gradient
  float* stats;      // forward only: [REPL][2][C] or nullptr
};

// grid-stride over (output position, channel group); cg = C / 8 divides 256 or the tail threads idle
__global__ __launch_bounds__(256) void dw_conv_kernel(DwArgs a) {
  extern __shared__ float sh[];
  const int cg = a.C / 8;
  const int rpb = 256 / cg;
  const int col = threadIdx.x % cg;
  const int rl = threadIdx.x / cg;
  const bool active = rl < rpb;
  const int OH = a.dgrad ? a.H : a.Ho, OW = a.dgrad ? a.W : a.Wo;     // extent of the tensor being written
  const int IH = a.dgrad ? a.Ho : a.H, IW = a.dgrad ? a.Wo : a.W;     // extent of the tensor being read
  const int64_t P = (int64_t)a.N * OH * OW;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (active) {
    for (int64_t p = (int64_t)blockIdx.x * rpb + rl; p < P; p += (int64_t)gridDim.x * rpb) {
      const int n = (int)(p / (OH * OW));
      const int rem = (int)(p - (int64_t)n * OH * OW);
      const int oh = rem / OW, ow = rem - oh * OW;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int r = 0; r < a.k; ++r) {
        int ih;
        if (!a.dgrad) {
          ih = oh * a.stride - a.pad + r;
        } else {
          const int t = oh + a.pad - r;
          if (t % a.stride) continue;
          ih = t / a.stride;
        }
        if (ih < 0 || ih >= IH) continue;
        for (int c = 0; c < a.k; ++c) {
          int iw;
          if (!a.dgrad) {
            iw = ow * a.stride - a.pad + c;
          } else {
            const int t = ow + a.pad - c;
            if (t % a.stride) continue;
            iw = t / a.stride;
          }
          if (iw < 0 || iw >= IW) continue;
          const uint4 v = *(const uint4*)(a.in + (((int64_t)n * IH + ih) * IW + iw) * a.C + col * 8);
          const uint32_t* vw = (const uint32_t*)&v;
          const float* wp = a.w + (size_t)(col * 8) * a.k * a.k + r * a.k + c;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[2 * j] += bf_lo(vw[j]) * wp[(2 * j) * a.k * a.k];
            acc[2 * j + 1] += bf_hi(vw[j]) * wp[(2 * j + 1) * a.k * a.k];
          }
        }
      }
      uint4 o;
      uint32_t* ow32 = (uint32_t*)&o;
#pragma unroll
      for (int j = 0; j < 4; ++j) ow32[j] = pack2(acc[2 * j], acc[2 * j + 1]);
      *(uint4*)(a.out + p * a.C + col * 8) = o;
      if (a.stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float f0 = bf_lo(ow32[j]), f1 = bf_hi(ow32[j]);
          s[2 * j] += f0;
          q[2 * j] += f0 * f0;
          s[2 * j + 1] += f1;
          q[2 * j + 1] += f1 * f1;
        }
      }
    }
  }
  if (a.stats) {
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) sh[i] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&sh[col * 8 + j], s[j]);
        atomicAdd(&sh[a.C + col * 8 + j], q[j]);
      }
    }
    __syncthreads();
    float* dst = a.stats + (size_t)(blockIdx.x % REPL) * 2 * a.C;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) atomicAdd(&dst[i], sh[i]);
  }
}

// dw[c][r][s] += sum over output pixels of dy[n,ho,wo,c] * x[n, ho*st - pad + r, wo*st - pad + s, c]
// grid = (pixel chunks, k*k taps)
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const u16* dy, const u16* x, float* dw, int N, int H, int W,
                                                       int C, int Ho, int Wo, int k, int stride, int pad) {
  extern __shared__ float sh[];
  const int cg = C / 8;
  const int rpb = 256 / cg;
  const int col = threadIdx.x % cg;
  const int rl = threadIdx.x / cg;
  const int tap = blockIdx.y;
  const int r = tap / k, c = tap - r * k;
  const int64_t P = (int64_t)N * Ho * Wo;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (rl < rpb) {
    for (int64_t p = (int64_t)blockIdx.x * rpb + rl; p < P; p += (int64_t)gridDim.x * rpb) {
      const int n = (int)(p / (Ho * Wo));
      const int rem = (int)(p - (int64_t)n * Ho * Wo);
      const int oh = rem / Wo, ow = rem - oh * Wo;
      const int ih = oh * stride - pad + r, iw = ow * stride - pad + c;
      if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
      const uint4 dv = *(const uint4*)(dy + p * C + col * 8);
      const uint4 xv = *(const uint4*)(x + (((int64_t)n * H + ih) * W + iw) * C + col * 8);
      const uint32_t* dw32 = (const uint32_t*)&dv;
      const uint32_t* xw32 = (const uint32_t*)&xv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += bf_lo(dw32[j]) * bf_lo(xw32[j]);
        acc[2 * j + 1] += bf_hi(dw32[j]) * bf_hi(xw32[j]);
      }
    }
  }
  for (int i = threadIdx.x; i < C; i += 256) sh[i] = 0.f;
  __syncthreads();
  if (rl < rpb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&sh[col * 8 + j], acc[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&dw[(size_t)i * k * k + tap], sh[i]);
}

int dw_check(const vlsfr_conv_desc* d, const char* who) {
  if (!d) return fail(VLSFR_EINVAL, "%s: null descriptor", who);
  if (d->Cin != d->Cout || d->Cin % 8 || d->Cin > 2048 || d->Cin <= 0)
    return fail(VLSFR_EINVAL, "%s: depthwise needs Cin == Cout, a multiple of 8, <= 2048", who);
  if (d->R != d->S || d->R < 1 || d->R > 7) return fail(VLSFR_EINVAL, "%s: square filters up to 7x7", who);
  if (d->stride < 1 || d->stride > 2 || d->pad < 0 || d->N <= 0 || d->H <= 0 || d->W <= 0)
    return fail(VLSFR_EINVAL, "%s: bad geometry", who);
  return VLSFR_OK;
}

inline int odim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

int dw_blocks(int64_t P, int C) {
  const int rpb = 256 / (C / 8);
  int64_t b = (P + (int64_t)rpb * 4 - 1) / ((int64_t)rpb * 4);
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

}  // namespace

extern "C" {

int vlsfr_dwconv_fwd(const vlsfr_conv_desc* d, const void* x, const float* w, void* y, float* stats, void* stream) {
  int rc = dw_check(d, "vlsfr_dwconv_fwd");
  if (rc) return rc;
  if (!x || !w || !y) return fail(VLSFR_EINVAL, "vlsfr_dwconv_fwd: null buffer");
  DwArgs a{(const u16*)x, w, (u16*)y, d->N, d->H, d->W, d->Cin, odim(d->H, d->R, d->stride, d->pad),
           odim(d->W, d->S, d->stride, d->pad), d->R, d->stride, d->pad, 0, stats};
  const int64_t P = (int64_t)a.N * a.Ho * a.Wo;
  hipLaunchKernelGGL(dw_conv_kernel, dim3(dw_blocks(P, a.C)), dim3(256), stats ? 2 * a.C * sizeof(float) : 0,
                     (hipStream_t)stream, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_fwd");
  return VLSFR_OK;
}

int vlsfr_dwconv_dgrad(const vlsfr_conv_desc* d, const void* dy, const float* w, void* dx, void* stream) {
  int rc = dw_check(d, "vlsfr_dwconv_dgrad");
  if (rc) return rc;
  if (!dy || !w || !dx) return fail(VLSFR_EINVAL, "vlsfr_dwconv_dgrad: null buffer");
  DwArgs a{(const u16*)dy, w, (u16*)dx, d->N, d->H, d->W, d->Cin, odim(d->H, d->R, d->stride, d->pad),
           odim(d->W, d->S, d->stride, d->pad), d->R, d->stride, d->pad, 1, nullptr};
  const int64_t P = (int64_t)a.N * a.H * a.W;
  hipLaunchKernelGGL(dw_conv_kernel, dim3(dw_blocks(P, a.C)), dim3(256), 0, (hipStream_t)stream, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_dgrad");
  return VLSFR_OK;
}

int vlsfr_dwconv_wgrad(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, void* stream) {
  int rc = dw_check(d, "vlsfr_dwconv_wgrad");
  if (rc) return rc;
  if (!dy || !x || !dw) return fail(VLSFR_EINVAL, "vlsfr_dwconv_wgrad: null buffer");
  const int Ho = odim(d->H, d->R, d->stride, d->pad), Wo = odim(d->W, d->S, d->stride, d->pad);
  const int64_t P = (int64_t)d->N * Ho * Wo;
  int nb = dw_blocks(P, d->Cin);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(dw_wgrad_kernel, dim3(nb, d->R * d->S), dim3(256), d->Cin * sizeof(float), (hipStream_t)stream,
                     (const u16*)dy, (const u16*)x, dw, d->N, d->H, d->W, d->Cin, Ho, Wo, d->R, d->stride, d->pad);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_wgrad");
  return VLSFR_OK;
}

}  // extern "C"
