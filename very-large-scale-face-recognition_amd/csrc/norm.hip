// Normalisation / activation / layout kernels of the backbones on gfx950 (HBM-bound, 16-byte
// vector accesses; C-ABI section 6 of include/vlsfr.h).  Replaces, forward and backward:
//   nn.BatchNorm2d in training mode (batch statistics, eps 1e-5, momentum 0.1; reference
//   model/resnet_arcface.py:35,37,40,75,93), nn.PReLU (:38,76), the residual add (:54),
//   the head BatchNorm1d with frozen weight + F.normalize (:96-98,151), and the input / weight
//   layout conversions (fp32 NCHW image -> bf16 im2col rows; fp32 master weights -> bf16 operands).
// Activations are NHWC bf16 ([M = N*H*W, C], C % 8 == 0); statistics and parameters are fp32.
#include "hip_common.h"

using namespace vlsfr;

namespace {

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(u16, b);
}
struct bf8 {
  uint4 raw;
  __device__ __forceinline__ float get(int j) const {
    const uint32_t w = ((const uint32_t*)&raw)[j >> 1];
    return __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
  }
};
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  uint32_t* w = (uint32_t*)&o;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = (uint32_t)f2bf(v[2 * j]) | ((uint32_t)f2bf(v[2 * j + 1]) << 16);
  return o;
}

// ---------------------------------------------------------------------------------------------
// per-channel sum and sum of squares over the M rows of x[M, C]
// sums: fp32 [2, C], pre-zeroed, accumulated atomically
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const u16* x, int64_t M, int C, float* sums) {
  extern __shared__ float sh[];   // [2, C]
  const int cg = C / 8;           // 16-byte chunk columns
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * C; i += 256) sh[i] = 0.f;
  __syncthreads();
  const int col = tid % cg;
  const int rsub = tid / cg;
  const int rows_per_iter = 256 / cg;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (rsub < rows_per_iter) {
    for (int64_t r = (int64_t)blockIdx.x * rows_per_iter + rsub; r < M; r += (int64_t)gridDim.x * rows_per_iter) {
      bf8 v;
      v.raw = *(const uint4*)(x + r * C + col * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = v.get(j);
        s[j] += f;
        q[j] += f * f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&sh[col * 8 + j], s[j]);
      atomicAdd(&sh[C + col * 8 + j], q[j]);
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) atomicAdd(&sums[i], sh[i]);
}

// ---------------------------------------------------------------------------------------------
// y = prelu(bn(x)) + residual   (each stage optional).  Every block derives scale/shift from the
// raw sums; block 0 also records mean / invstd and updates the running statistics.
// ---------------------------------------------------------------------------------------------
struct BnApplyArgs {
  const u16* x;
  u16* y;
  int64_t M;
  int C, HW;
  const float* sums;      // [2, C]
  const float* gamma;
  const float* beta;
  const float* slope;     // PReLU or nullptr
  const u16* residual;    // or nullptr
  float* save_mean;       // [C]
  float* save_invstd;     // [C]
  float* running_mean;    // or nullptr
  float* running_var;
  float eps, momentum;
  int out_nchw;           // 1: y index = n*(C*HW) + c*HW + hw (the flatten order of the reference's fc input)
};

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a) {
  extern __shared__ float sh[];   // scale[C], shift[C], slope[C]
  float* scale = sh;
  float* shift = sh + a.C;
  float* slp = sh + 2 * a.C;
  const int tid = threadIdx.x;
  const float invM = 1.f / (float)a.M;
  for (int c = tid; c < a.C; c += 256) {
    const float mean = a.sums[c] * invM;
    float var = a.sums[a.C + c] * invM - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float invstd = rsqrtf(var + a.eps);
    const float g = a.gamma ? a.gamma[c] : 1.f;
    const float b = a.beta ? a.beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - mean * g * invstd;
    slp[c] = a.slope ? a.slope[c] : 1.f;
    if (blockIdx.x == 0) {
      a.save_mean[c] = mean;
      a.save_invstd[c] = invstd;
      if (a.running_mean) {
        const float unb = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unb;
      }
    }
  }
  __syncthreads();
  const int cg = a.C / 8;
  const int64_t total = a.M * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cg;
    const int col = (int)(i - r * cg);
    bf8 v, rs;
    v.raw = *(const uint4*)(a.x + r * a.C + col * 8);
    if (a.residual) rs.raw = *(const uint4*)(a.residual + r * a.C + col * 8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = col * 8 + j;
      float z = v.get(j) * scale[c] + shift[c];
      if (a.slope) z = z > 0.f ? z : z * slp[c];
      if (a.residual) z += rs.get(j);
      o[j] = z;
    }
    if (!a.out_nchw) {
      *(uint4*)(a.y + r * a.C + col * 8) = pack8(o);
    } else {
      const int64_t n = r / a.HW;
      const int hw = (int)(r - n * a.HW);
#pragma unroll
      for (int j = 0; j < 8; ++j) a.y[(n * a.C + col * 8 + j) * a.HW + hw] = f2bf(o[j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, pass 1: per-channel sum dz, sum dz*xhat, sum dy*min(z,0) (PReLU slope gradient)
// where z = bn(x), dz = dy * prelu'(z).   red: fp32 [3, C] pre-zeroed.
// ---------------------------------------------------------------------------------------------
struct BnBwdArgs {
  const u16* dy;
  const u16* x;           // BN input (conv output)
  u16* dx;
  int64_t M;
  int C, HW;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* slope;     // or nullptr
  float* red;             // [3, C]
  const u16* dx_add;      // extra gradient added to dx (identity branch) or nullptr
  float* dgamma;          // accumulated (+=) by block 0 of the apply pass; may be nullptr (frozen)
  float* dbeta;
  float* dslope;
  int dy_nchw;            // dy laid out as the flatten order (see bn_apply out_nchw)
};

__device__ __forceinline__ float load_dy(const BnBwdArgs& a, int64_t r, int c) {
  if (!a.dy_nchw) return bf2f(a.dy[r * a.C + c]);
  const int64_t n = r / a.HW;
  const int hw = (int)(r - n * a.HW);
  return bf2f(a.dy[(n * a.C + c) * a.HW + hw]);
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a) {
  extern __shared__ float sh[];   // [3, C]
  const int C = a.C;
  const int cg = C / 8;
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * C; i += 256) sh[i] = 0.f;
  __syncthreads();
  const int col = tid % cg;
  const int rsub = tid / cg;
  const int rows_per_iter = 256 / cg;
  float s0[8], s1[8], s2[8], mu[8], is[8], g[8], b[8], sl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = col * 8 + j;
    s0[j] = s1[j] = s2[j] = 0.f;
    mu[j] = a.mean[c];
    is[j] = a.invstd[c];
    g[j] = a.gamma ? a.gamma[c] : 1.f;
    b[j] = a.beta ? a.beta[c] : 0.f;
    sl[j] = a.slope ? a.slope[c] : 1.f;
  }
  if (rsub < rows_per_iter) {
    for (int64_t r = (int64_t)blockIdx.x * rows_per_iter + rsub; r < a.M; r += (int64_t)gridDim.x * rows_per_iter) {
      bf8 xv, dv;
      xv.raw = *(const uint4*)(a.x + r * C + col * 8);
      if (!a.dy_nchw) dv.raw = *(const uint4*)(a.dy + r * C + col * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xhat = (xv.get(j) - mu[j]) * is[j];
        float dyv = a.dy_nchw ? load_dy(a, r, col * 8 + j) : dv.get(j);
        float dz = dyv;
        if (a.slope) {
          const float z = xhat * g[j] + b[j];
          if (z <= 0.f) {
            s2[j] += dyv * z;
            dz = dyv * sl[j];
          }
        }
        s0[j] += dz;
        s1[j] += dz * xhat;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&sh[col * 8 + j], s0[j]);
      atomicAdd(&sh[C + col * 8 + j], s1[j]);
      if (a.slope) atomicAdd(&sh[2 * C + col * 8 + j], s2[j]);
    }
  }
  __syncthreads();
  const int nred = a.slope ? 3 * C : 2 * C;
  for (int i = tid; i < nred; i += 256) atomicAdd(&a.red[i], sh[i]);
}

// backward, pass 2: dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) (+ dx_add)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
  extern __shared__ float sh[];   // k0[C] = gamma*invstd, k1[C] = mean dz, k2[C] = mean dz*xhat
  const int C = a.C;
  float* k0 = sh;
  float* k1 = sh + C;
  float* k2 = sh + 2 * C;
  const int tid = threadIdx.x;
  const float invM = 1.f / (float)a.M;
  for (int c = tid; c < C; c += 256) {
    const float g = a.gamma ? a.gamma[c] : 1.f;
    k0[c] = g * a.invstd[c];
    k1[c] = a.red[c] * invM;
    k2[c] = a.red[C + c] * invM;
    if (blockIdx.x == 0) {
      if (a.dbeta) a.dbeta[c] += a.red[c];
      if (a.dgamma) a.dgamma[c] += a.red[C + c];
      if (a.dslope && a.slope) a.dslope[c] += a.red[2 * C + c];
    }
  }
  __syncthreads();
  const int cg = C / 8;
  const int64_t total = a.M * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cg;
    const int col = (int)(i - r * cg);
    bf8 xv, dv, av;
    xv.raw = *(const uint4*)(a.x + r * C + col * 8);
    if (!a.dy_nchw) dv.raw = *(const uint4*)(a.dy + r * C + col * 8);
    if (a.dx_add) av.raw = *(const uint4*)(a.dx_add + r * C + col * 8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = col * 8 + j;
      const float xhat = (xv.get(j) - a.mean[c]) * a.invstd[c];
      float dz = a.dy_nchw ? load_dy(a, r, c) : dv.get(j);
      if (a.slope) {
        const float z = xhat * (a.gamma ? a.gamma[c] : 1.f) + (a.beta ? a.beta[c] : 0.f);
        if (z <= 0.f) dz *= a.slope[c];
      }
      float d = k0[c] * (dz - k1[c] - xhat * k2[c]);
      if (a.dx_add) d += av.get(j);
      o[j] = d;
    }
    *(uint4*)(a.dx + r * C + col * 8) = pack8(o);
  }
}

// y = a + b (bf16 tensors): sum of two gradient branches
__global__ __launch_bounds__(256) void add_bf16_kernel(const u16* p, const u16* q, u16* y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    bf8 u, v;
    u.raw = ((const uint4*)p)[i];
    v.raw = ((const uint4*)q)[i];
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = u.get(j) + v.get(j);
    ((uint4*)y)[i] = pack8(o);
  }
}

// ---------------------------------------------------------------------------------------------
// embedding tail: e = normalize(bn1d(fc + bias)), BN1d weight frozen at its stored value
// (resnet_arcface.py:96-98,150-151).  Tiny ([B, D]): one thread per channel, then one block per row.
// ---------------------------------------------------------------------------------------------
struct EmbedArgs {
  const float* fc;        // [B, D] fp32 (split-K accumulated matmul, no bias)
  const float* fc_bias;   // [D]
  const float* gamma;     // features.weight
  const float* beta;      // features.bias
  float* running_mean;
  float* running_var;
  float* z;               // [B, D] BN output (saved)
  float* xhat;            // [B, D] (saved)
  float* invstd;          // [D]
  float* emb;             // [B, D]
  float* inv_norm;        // [B]
  int B, D;
  float eps, momentum;
};

__global__ void embed_bn_kernel(EmbedArgs a) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= a.D) return;
  const float bias = a.fc_bias[d];
  float s = 0.f, q = 0.f;
  for (int b = 0; b < a.B; ++b) {
    const float v = a.fc[(size_t)b * a.D + d] + bias;
    s += v;
    q += v * v;
  }
  const float mean = s / a.B;
  float var = q / a.B - mean * mean;
  var = var > 0.f ? var : 0.f;
  const float is = rsqrtf(var + a.eps);
  a.invstd[d] = is;
  if (a.running_mean) {
    const float unb = a.B > 1 ? var * a.B / (a.B - 1) : var;
    a.running_mean[d] = (1.f - a.momentum) * a.running_mean[d] + a.momentum * mean;
    a.running_var[d] = (1.f - a.momentum) * a.running_var[d] + a.momentum * unb;
  }
  const float g = a.gamma[d], be = a.beta[d];
  for (int b = 0; b < a.B; ++b) {
    const float xh = (a.fc[(size_t)b * a.D + d] + bias - mean) * is;
    a.xhat[(size_t)b * a.D + d] = xh;
    a.z[(size_t)b * a.D + d] = xh * g + be;
  }
}

__global__ __launch_bounds__(256) void embed_norm_kernel(EmbedArgs a) {
  const int b = blockIdx.x;
  __shared__ float red[4];
  float q = 0.f;
  for (int d = threadIdx.x; d < a.D; d += 256) {
    const float v = a.z[(size_t)b * a.D + d];
    q += v * v;
  }
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  const float inv = 1.f / fmaxf(nrm, 1e-12f);            // F.normalize eps
  if (threadIdx.x == 0) a.inv_norm[b] = inv;
  for (int d = threadIdx.x; d < a.D; d += 256) a.emb[(size_t)b * a.D + d] = a.z[(size_t)b * a.D + d] * inv;
}

struct EmbedBwdArgs {
  const float* demb;      // [B, D]
  const float* emb;
  const float* inv_norm;
  const float* xhat;
  const float* invstd;
  const float* gamma;
  float* dz;              // [B, D] scratch
  u16* dfc;               // [B, D] bf16: gradient wrt the fc output, operand of the fc dgrad / wgrad
  float* dbeta;           // features.bias grad (+=)
  float* dfc_bias;        // fc.bias grad (+=)
  int B, D;
};

__global__ __launch_bounds__(256) void embed_norm_bwd_kernel(EmbedBwdArgs a) {
  const int b = blockIdx.x;
  __shared__ float red[4];
  float dot = 0.f;
  for (int d = threadIdx.x; d < a.D; d += 256) dot += a.emb[(size_t)b * a.D + d] * a.demb[(size_t)b * a.D + d];
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  const float inv = a.inv_norm[b];
  for (int d = threadIdx.x; d < a.D; d += 256) {
    const size_t i = (size_t)b * a.D + d;
    a.dz[i] = (a.demb[i] - a.emb[i] * dot) * inv;
  }
}

__global__ void embed_bn_bwd_kernel(EmbedBwdArgs a) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= a.D) return;
  const float g = a.gamma[d];
  float s0 = 0.f, s1 = 0.f;
  for (int b = 0; b < a.B; ++b) {
    const float dzv = a.dz[(size_t)b * a.D + d];
    s0 += dzv;
    s1 += dzv * a.xhat[(size_t)b * a.D + d];
  }
  a.dbeta[d] += s0;
  const float k = g * a.invstd[d];
  const float m0 = g * s0 / a.B, m1 = g * s1 / a.B;
  float sb = 0.f;
  for (int b = 0; b < a.B; ++b) {
    const size_t i = (size_t)b * a.D + d;
    const float dv = a.invstd[d] * (g * a.dz[i] - m0 - a.xhat[i] * m1);
    (void)k;
    a.dfc[i] = f2bf(dv);
    sb += dv;
  }
  a.dfc_bias[d] += sb;
}

// ---------------------------------------------------------------------------------------------
// layout / precision conversions
// ---------------------------------------------------------------------------------------------
// fp32 [rows][K] -> bf16 [rows][Kp] (zero padded) and, optionally, the [R*S][... ] transpose used by dgrad:
// wT[(c)][tap][(row)]  for w[(row)][tap][(c)]   (rows = Cout, C = Cin, taps = R*S)
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* w, u16* wb, u16* wT, int rows, int taps, int C,
                                                          int Kp) {
  const int K = taps * C;
  const int64_t total = (int64_t)rows * Kp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / Kp);
    const int k = (int)(i - (int64_t)row * Kp);
    const float v = k < K ? w[(size_t)row * K + k] : 0.f;
    const u16 b = f2bf(v);
    wb[i] = b;
    if (wT && k < K) {
      const int tap = k / C, c = k - tap * C;
      wT[((size_t)c * taps + tap) * rows + row] = b;
    }
  }
}

// stem: fp32 NCHW image [N,3,H,W] -> bf16 im2col rows [N*H*W][32], k = (r*3 + s)*3 + c, 3x3 pad 1 stride 1
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* x, u16* out, int N, int H, int W) {
  const int64_t P = (int64_t)N * H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int n = (int)(p / (H * W));
    const int rem = (int)(p - (int64_t)n * H * W);
    const int ho = rem / W, wo = rem - ho * W;
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hi = ho + r - 1, wi = wo + s - 1;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
#pragma unroll
          for (int c = 0; c < 3; ++c) v[(r * 3 + s) * 3 + c] = x[(((size_t)n * 3 + c) * H + hi) * W + wi];
        }
      }
    uint4* dst = (uint4*)(out + p * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = pack8(v + 8 * j);
  }
}

// dst[rows][Kdst] += src[rows][Ksrc][:Kdst]   (un-pad the stem weight gradient)
__global__ void unpad_add_kernel(const float* src, float* dst, int rows, int Ksrc, int Kdst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Kdst) return;
  const int r = i / Kdst, k = i - r * Kdst;
  dst[i] += src[r * Ksrc + k];
}

inline int blocks_for(int64_t work_items, int per_block = 256, int cap = 2048) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" {

int vlsfr_bn_stats(const void* x, int64_t M, int32_t C, float* sums, void* stream) {
  if (!x || !sums || M <= 0 || C <= 0 || C % 8 || C > 2048)
    return fail(VLSFR_EINVAL, "vlsfr_bn_stats: need C %% 8 == 0, C <= 2048 (got %d)", C);
  const int rows_per_iter = 256 / (C / 8);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks_for(M, rows_per_iter * 8, 1024)), dim3(256), 2 * C * sizeof(float),
                     (hipStream_t)stream, (const u16*)x, M, C, sums);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_stats");
  return VLSFR_OK;
}

int vlsfr_bn_apply(const void* x, void* y, int64_t M, int32_t C, int32_t HW, const float* sums, const float* gamma,
                   const float* beta, const float* slope, const void* residual, float* save_mean,
                   float* save_invstd, float* running_mean, float* running_var, float eps, float momentum,
                   int32_t out_nchw, void* stream) {
  if (!x || !y || !sums || !save_mean || !save_invstd || M <= 0 || C <= 0 || C % 8 || C > 2048 || HW <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_bn_apply: bad argument");
  BnApplyArgs a{(const u16*)x, (u16*)y, M, C, HW, sums, gamma, beta, slope, (const u16*)residual, save_mean,
                save_invstd, running_mean, running_var, eps, momentum, out_nchw};
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks_for(M * (C / 8), 256 * 4)), dim3(256), 3 * C * sizeof(float),
                     (hipStream_t)stream, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_apply");
  return VLSFR_OK;
}

int vlsfr_bn_backward(const void* dy, const void* x, void* dx, int64_t M, int32_t C, int32_t HW, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, const float* slope, float* red,
                      const void* dx_add, float* dgamma, float* dbeta, float* dslope, int32_t dy_nchw, void* stream) {
  if (!dy || !x || !dx || !mean || !invstd || !red || M <= 0 || C <= 0 || C % 8 || C > 2048 || HW <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_bn_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(red, 0, 3 * C * sizeof(float), st);
  if (e != hipSuccess) return hip_fail(e, "vlsfr_bn_backward: memset");
  BnBwdArgs a{(const u16*)dy, (const u16*)x, (u16*)dx, M, C, HW, mean, invstd, gamma, beta, slope, red,
              (const u16*)dx_add, dgamma, dbeta, dslope, dy_nchw};
  const int rows_per_iter = 256 / (C / 8);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks_for(M, rows_per_iter * 8, 1024)), dim3(256),
                     3 * C * sizeof(float), st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_backward reduce");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(M * (C / 8), 256 * 4)), dim3(256), 3 * C * sizeof(float), st,
                     a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_backward apply");
  return VLSFR_OK;
}

int vlsfr_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
  if (!a || !b || !y || n <= 0 || n % 8) return fail(VLSFR_EINVAL, "vlsfr_add_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(blocks_for(n / 8, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)a, (const u16*)b, (u16*)y, n / 8);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_add_bf16");
  return VLSFR_OK;
}

int vlsfr_embed_fwd(const float* fc, const float* fc_bias, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* z, float* xhat, float* invstd, float* emb,
                    float* inv_norm, int32_t B, int32_t D, float eps, float momentum, void* stream) {
  if (!fc || !fc_bias || !gamma || !beta || !z || !xhat || !invstd || !emb || !inv_norm || B <= 0 || D <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_embed_fwd: bad argument");
  EmbedArgs a{fc, fc_bias, gamma, beta, running_mean, running_var, z, xhat, invstd, emb, inv_norm, B, D, eps, momentum};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_bn_kernel, dim3((D + 63) / 64), dim3(64), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_fwd bn");
  hipLaunchKernelGGL(embed_norm_kernel, dim3(B), dim3(256), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_fwd norm");
  return VLSFR_OK;
}

int vlsfr_embed_bwd(const float* demb, const float* emb, const float* inv_norm, const float* xhat, const float* invstd,
                    const float* gamma, float* dz, void* dfc_bf16, float* dbeta, float* dfc_bias, int32_t B,
                    int32_t D, void* stream) {
  if (!demb || !emb || !inv_norm || !xhat || !invstd || !gamma || !dz || !dfc_bf16 || !dbeta || !dfc_bias || B <= 0 ||
      D <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_embed_bwd: bad argument");
  EmbedBwdArgs a{demb, emb, inv_norm, xhat, invstd, gamma, dz, (u16*)dfc_bf16, dbeta, dfc_bias, B, D};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_norm_bwd_kernel, dim3(B), dim3(256), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_bwd norm");
  hipLaunchKernelGGL(embed_bn_bwd_kernel, dim3((D + 63) / 64), dim3(64), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_bwd bn");
  return VLSFR_OK;
}

int vlsfr_cast_weight(const float* w, void* w_bf16, void* wT_bf16, int32_t rows, int32_t taps, int32_t C, int32_t Kp,
                      void* stream) {
  if (!w || !w_bf16 || rows <= 0 || taps <= 0 || C <= 0 || Kp < taps * C)
    return fail(VLSFR_EINVAL, "vlsfr_cast_weight: bad argument");
  hipLaunchKernelGGL(cast_weight_kernel, dim3(blocks_for((int64_t)rows * Kp, 256 * 4)), dim3(256), 0,
                     (hipStream_t)stream, w, (u16*)w_bf16, (u16*)wT_bf16, rows, taps, C, Kp);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_cast_weight");
  return VLSFR_OK;
}

int vlsfr_stem_im2col(const float* x_nchw, void* out, int32_t N, int32_t H, int32_t W, void* stream) {
  if (!x_nchw || !out || N <= 0 || H <= 0 || W <= 0) return fail(VLSFR_EINVAL, "vlsfr_stem_im2col: bad argument");
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(blocks_for((int64_t)N * H * W, 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, x_nchw, (u16*)out, N, H, W);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_stem_im2col");
  return VLSFR_OK;
}

int vlsfr_unpad_add(const float* src, float* dst, int32_t rows, int32_t Ksrc, int32_t Kdst, void* stream) {
  if (!src || !dst || rows <= 0 || Kdst <= 0 || Ksrc < Kdst) return fail(VLSFR_EINVAL, "vlsfr_unpad_add: bad argument");
  hipLaunchKernelGGL(unpad_add_kernel, dim3((rows * Kdst + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst,
                     rows, Ksrc, Kdst);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_unpad_add");
  return VLSFR_OK;
}

}  // extern "C"
