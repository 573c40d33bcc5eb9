// Depthwise convolutions of MobileFaceNet on gfx950 (C-ABI section 5b of include/vlsfr.h): replaces
// nn.Conv2d(groups = channels) of reference model/mobilefacenet_def.py:39 (3x3, stride 1/2, pad 1)
// and :60,88 (7x7 "global" depthwise, valid), forward, input gradient and weight gradient.
// These layers are ~4 % of the MACs and have no channel contraction, so they are HBM-bound VALU
// kernels (not MFMA): NHWC bf16 activations, one thread = 8 channels (16 B) of one output pixel,
// fp32 weights [C][k*k] read as they lie in the parameter.  The forward kernel also accumulates the
// BatchNorm statistics of its output (same replicated accumulators as the dense kernels).
#include "hip_common.h"

using namespace vlsfr;

namespace vlsfr {
int g_dw_strip = 1;            // "dw_strip": 1 = strip kernels (sliding 3 x 3 window) for the stride-1 depthwise layers, 0 = per-pixel kernels
int g_dw_wgrad_blocks = 256;   // "dw_wgrad_blocks": workgroups of the one-pass depthwise weight gradient (all add into the same 9*C addresses)
}

namespace {

constexpr int REPL = VLSFR_BN_REPL;
}  // namespace
namespace vlsfr {
extern int g_bn_repl;   // csrc/norm.hip
}
namespace {

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
  double* stats;     // forward only: [REPL][2][C] float64 (sum, sum of squares) or nullptr.  Block partials are plain fp32 sums
                     // (a few thousand outputs each), the accumulation across blocks is float64
  int repl = 1;      // replicas in use (vlsfr::g_bn_repl)
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
    double* dst = a.stats + (size_t)(blockIdx.x % a.repl) * 2 * a.C;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) atomicAdd(&dst[i], (double)sh[i]);
  }
}

// "Global" depthwise forward (MobileFaceNet linear7, mobilefacenet_def.py:88: k x k valid filter on a k x k map, one output
// pixel per image): one workgroup per image, the k*k taps dealt to four thread groups of C / 8 lanes, combined through LDS.
// The generic kernel has one thread per (output pixel, 8 channels) — 16 workgroups for a batch of 256 (185 us per launch).
__global__ __launch_bounds__(256) void dw_global_fwd_kernel(DwArgs a) {
  extern __shared__ float sh[];                 // [4][C] partial outputs, then [2][C] statistics
  const int cg = a.C / 8;                       // <= 64 (host-checked)
  const int col = threadIdx.x % cg, grp = threadIdx.x / cg;
  const int ngrp = 256 / cg;                    // >= 4
  const int n = blockIdx.x;
  const int taps = a.k * a.k;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (grp < 4) {
    const u16* in = a.in + (size_t)n * taps * a.C + col * 8;
    for (int t = grp; t < taps; t += 4) {
      const uint4 v = *(const uint4*)(in + (size_t)t * a.C);
      const uint32_t* vw = (const uint32_t*)&v;
      const float* wp = a.w + (size_t)(col * 8) * taps + t;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += bf_lo(vw[j]) * wp[(2 * j) * taps];
        acc[2 * j + 1] += bf_hi(vw[j]) * wp[(2 * j + 1) * taps];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[grp * a.C + col * 8 + j] = acc[j];
  }
  (void)ngrp;
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    const float y = (sh[c] + sh[a.C + c]) + (sh[2 * a.C + c] + sh[3 * a.C + c]);
    const __bf16 yb = (__bf16)y;
    a.out[(size_t)n * a.C + c] = __builtin_bit_cast(u16, yb);
    if (a.stats) {
      const float f = (float)yb;
      double* dst = a.stats + (size_t)(blockIdx.x % a.repl) * 2 * a.C;
      atomicAdd(&dst[c], (double)f);
      atomicAdd(&dst[a.C + c], (double)f * (double)f);
    }
  }
}

// 3x3 / pad 1 specialisation (every depthwise layer of MobileFaceNet but the 7x7 global one): the 72
// weights of the thread's 8 channels live in registers, the nine taps of a pixel are nine predicated
// 16-byte loads issued together (no per-tap branches, no weight loads in the loop); stride and
// direction are compile-time.  DGRAD: dx[ih, iw] = sum_{r,c} dy[(ih + 1 - r) / st, (iw + 1 - c) / st] w[r][c]
// over the taps that divide.
template <int STRIDE, bool DGRAD, bool STATS>
__global__ __launch_bounds__(256) void dw3_kernel(DwArgs a) {
  extern __shared__ float sh[];
  const int cg = a.C / 8;
  const int rpb = 256 / cg;
  const int col = threadIdx.x % cg;
  const int rl = threadIdx.x / cg;
  const bool active = rl < rpb;
  const int OH = DGRAD ? a.H : a.Ho, OW = DGRAD ? a.W : a.Wo;
  const int IH = DGRAD ? a.Ho : a.H, IW = DGRAD ? a.Wo : a.W;
  const int64_t P = (int64_t)a.N * OH * OW;
  float wr[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[t][j] = a.w[(size_t)(col * 8 + j) * 9 + t];
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (active) {
    const u16* in = a.in + col * 8;
    for (int64_t p = (int64_t)blockIdx.x * rpb + rl; p < P; p += (int64_t)gridDim.x * rpb) {
      const int n = (int)(p / (OH * OW));
      const int rem = (int)(p - (int64_t)n * OH * OW);
      const int oh = rem / OW, ow = rem - oh * OW;
      uint4 v[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int ih, iw;
          bool ok;
          if (!DGRAD) {
            ih = oh * STRIDE - 1 + r;
            iw = ow * STRIDE - 1 + c;
            ok = true;
          } else {
            const int th = oh + 1 - r, tw = ow + 1 - c;
            ok = STRIDE == 1 || !((th | tw) & 1);
            ih = STRIDE == 1 ? th : th >> 1;
            iw = STRIDE == 1 ? tw : tw >> 1;
          }
          ok = ok && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
          v[r * 3 + c] = ok ? *(const uint4*)(in + (((int64_t)n * IH + ih) * IW + iw) * a.C) : make_uint4(0, 0, 0, 0);
        }
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const uint32_t* vw = (const uint32_t*)&v[t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += bf_lo(vw[j]) * wr[t][2 * j];
          acc[2 * j + 1] += bf_hi(vw[j]) * wr[t][2 * j + 1];
        }
      }
      uint4 o;
      uint32_t* ow32 = (uint32_t*)&o;
#pragma unroll
      for (int j = 0; j < 4; ++j) ow32[j] = pack2(acc[2 * j], acc[2 * j + 1]);
      *(uint4*)(a.out + p * a.C + col * 8) = o;
      if (STATS) {
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
  if (STATS) {
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
    double* dst = a.stats + (size_t)(blockIdx.x % a.repl) * 2 * a.C;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) atomicAdd(&dst[i], (double)sh[i]);
  }
}

// 3x3 weight gradient in one pass: per output pixel one dy load and nine predicated x loads, 72 per-thread
// accumulators, block reduction in LDS, one atomic per (channel, tap) and block.
template <int STRIDE>
__global__ __launch_bounds__(256) void dw3_wgrad_kernel(const u16* dy, const u16* x, float* dw, float* part, int N, int H, int W, int C,
                                                        int Ho, int Wo) {
  extern __shared__ float sh[];   // [9][C]
  const int cg = C / 8;
  const int rpb = 256 / cg;
  const int col = threadIdx.x % cg;
  const int rl = threadIdx.x / cg;
  const int64_t P = (int64_t)N * Ho * Wo;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  if (rl < rpb) {
    const u16* xin = x + col * 8;
    for (int64_t p = (int64_t)blockIdx.x * rpb + rl; p < P; p += (int64_t)gridDim.x * rpb) {
      const int n = (int)(p / (Ho * Wo));
      const int rem = (int)(p - (int64_t)n * Ho * Wo);
      const int oh = rem / Wo, ow = rem - oh * Wo;
      const uint4 dv = *(const uint4*)(dy + p * C + col * 8);
      uint4 v[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int ih = oh * STRIDE - 1 + r, iw = ow * STRIDE - 1 + c;
          const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
          v[r * 3 + c] = ok ? *(const uint4*)(xin + (((int64_t)n * H + ih) * W + iw) * C) : make_uint4(0, 0, 0, 0);
        }
      const uint32_t* dw32 = (const uint32_t*)&dv;
      float d[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[2 * j] = bf_lo(dw32[j]);
        d[2 * j + 1] = bf_hi(dw32[j]);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const uint32_t* vw = (const uint32_t*)&v[t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[t][2 * j] += d[2 * j] * bf_lo(vw[j]);
          acc[t][2 * j + 1] += d[2 * j + 1] * bf_hi(vw[j]);
        }
      }
    }
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) sh[i] = 0.f;
  __syncthreads();
  // lanes l, l + cg, l + 2 cg, ... of a wave hold the same channel group: sum them in registers first
  // (permlane swaps / DPP rotations), then one LDS atomic per value from the first cg lanes
  const bool pow2 = (cg & (cg - 1)) == 0 && cg <= 32;
  if (pow2) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = rl < rpb ? acc[t][j] : 0.f;
        if (cg <= 32) v = lane_step_sum<32>(v);
        if (cg <= 16) v = lane_step_sum<16>(v);
        if (cg <= 8) v = lane_step_sum<8>(v);
        if (cg <= 4) v = lane_step_sum<4>(v);
        if (cg <= 2) v = lane_step_sum<2>(v);
        if (cg <= 1) v = lane_step_sum<1>(v);
        acc[t][j] = v;
      }
  }
  if (pow2 ? (threadIdx.x & 63) < cg : rl < rpb) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&sh[t * C + col * 8 + j], acc[t][j]);
  }
  __syncthreads();
  if (part) {   // per-block partial sums, summed over the blocks by dw_wgrad_reduce_kernel (no same-address atomics)
    for (int i = threadIdx.x; i < 9 * C; i += 256) part[(size_t)blockIdx.x * 9 * C + i] = sh[i];
    return;
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) {
    const int t = i / C, c = i - t * C;
    atomicAdd(&dw[(size_t)c * 9 + t], sh[i]);
  }
}

// dw[c][t] += sum_b part[b][t][c]: one thread per (tap, channel) and slice of the blocks (blockIdx.y of DW_REDUCE_SLICES:
// a single thread walking 512 partials is 500 dependent-latency loads), coalesced over the channels
constexpr int DW_REDUCE_SLICES = 8;
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* part, int nblk, int C, float* dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * C) return;
  const int per = (nblk + DW_REDUCE_SLICES - 1) / DW_REDUCE_SLICES;
  const int b0 = blockIdx.y * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += part[(size_t)b * 9 * C + i];
    s1 += part[(size_t)(b + 1) * 9 * C + i];
    s2 += part[(size_t)(b + 2) * 9 * C + i];
    s3 += part[(size_t)(b + 3) * 9 * C + i];
  }
  for (; b < b1; ++b) s0 += part[(size_t)b * 9 * C + i];
  if (b1 <= b0) return;
  const int t = i / C, c = i - t * C;
  atomicAdd(&dw[(size_t)c * 9 + t], (s0 + s1) + (s2 + s3));   // 8 slices + the two backward passes of a step share dw
}

// ---- strip kernels (3x3, pad 1, stride 1): a thread walks a run of output pixels along a row with the 3 x 3 input window
// of its 8 channels in registers — 3 new 16-byte loads per output instead of 9 (the per-pixel form is bound by the
// vector-memory path, not HBM: 1.1-1.9 TB/s of tensor bytes), and all index arithmetic per strip, in 32 bits.
constexpr int DW_STRIP = 14;
constexpr int DW_WGRAD_MAX_BLOCKS = 512;   // partial-sum slabs of the depthwise weight gradient (vlsfr_dwconv_wgrad_ws)

__device__ __forceinline__ void dw_unpack(const uint4& v, float (&f)[8]) {
  const uint32_t* w = (const uint32_t*)&v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = bf_lo(w[j]);
    f[2 * j + 1] = bf_hi(w[j]);
  }
}

// forward (DGRAD = false) / input gradient (true: the same stencil with the filter flipped) of a stride-1 layer
template <bool DGRAD, bool STATS>
__global__ __launch_bounds__(256) void dw3_strip_kernel(DwArgs a) {
  extern __shared__ float sh[];
  const int cg = a.C / 8;
  const int rpb = 256 / cg;
  const int col = threadIdx.x % cg;
  const int rl = threadIdx.x / cg;
  const int H = a.H, W = a.W;                      // stride 1, pad 1: input and output extents agree
  const int nseg = (W + DW_STRIP - 1) / DW_STRIP;
  const int nstrip = a.N * H * nseg;
  float wr[3][3][8];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) wr[r][c][j] = a.w[(size_t)(col * 8 + j) * 9 + (DGRAD ? (2 - r) * 3 + (2 - c) : r * 3 + c)];
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (rl < rpb) {
    for (int sid = blockIdx.x * rpb + rl; sid < nstrip; sid += gridDim.x * rpb) {
      const int seg = sid % nseg;
      const int row = sid / nseg;                   // n * H + oh
      const int oh = row % H;
      const int ow0 = seg * DW_STRIP;
      const int len = W - ow0 < DW_STRIP ? W - ow0 : DW_STRIP;
      const bool rv[3] = {oh > 0, true, oh + 1 < H};
      const u16* base = a.in + ((size_t)row * W + ow0) * a.C + col * 8;     // (n, oh, ow0)
      u16* out = a.out + ((size_t)row * W + ow0) * a.C + col * 8;
      const int rstep = W * a.C;
      uint4 m1[3], c0[3], p1[3], nx[3];
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const u16* rp = base + (r - 1) * rstep;
        m1[r] = (rv[r] && ow0 > 0) ? *(const uint4*)(rp - a.C) : z;
        c0[r] = rv[r] ? *(const uint4*)rp : z;
        p1[r] = (rv[r] && ow0 + 1 < W) ? *(const uint4*)(rp + a.C) : z;
      }
      for (int i = 0; i < len; ++i) {
        // the column after next is fetched while this output is computed (two columns of loads in flight per thread)
        const bool cv = ow0 + i + 2 < W && i + 1 < len;
#pragma unroll
        for (int r = 0; r < 3; ++r) nx[r] = (rv[r] && cv) ? *(const uint4*)(base + (r - 1) * rstep + (i + 2) * a.C) : z;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float f0[8], f1[8], f2[8];
          dw_unpack(m1[r], f0);
          dw_unpack(c0[r], f1);
          dw_unpack(p1[r], f2);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += f0[j] * wr[r][0][j] + f1[j] * wr[r][1][j] + f2[j] * wr[r][2][j];
        }
        uint4 o;
        uint32_t* ow32 = (uint32_t*)&o;
#pragma unroll
        for (int j = 0; j < 4; ++j) ow32[j] = pack2(acc[2 * j], acc[2 * j + 1]);
        *(uint4*)(out + (size_t)i * a.C) = o;
        if (STATS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float f0 = bf_lo(ow32[j]), f1 = bf_hi(ow32[j]);
            s[2 * j] += f0;
            q[2 * j] += f0 * f0;
            s[2 * j + 1] += f1;
            q[2 * j + 1] += f1 * f1;
          }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          m1[r] = c0[r];
          c0[r] = p1[r];
          p1[r] = nx[r];
        }
      }
    }
  }
  if (STATS) {
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) sh[i] = 0.f;
    __syncthreads();
    if (rl < rpb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&sh[col * 8 + j], s[j]);
        atomicAdd(&sh[a.C + col * 8 + j], q[j]);
      }
    }
    __syncthreads();
    double* dst = a.stats + (size_t)(blockIdx.x % a.repl) * 2 * a.C;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) atomicAdd(&dst[i], (double)sh[i]);
  }
}

// weight gradient of a stride-1 layer, strip form: per output pixel one dy load and three new x loads
__global__ __launch_bounds__(256) void dw3_wgrad_strip_kernel(const u16* dy, const u16* x, float* dw, float* part, int N, int H, int W, int C) {
  extern __shared__ float sh[];   // [9][C]
  const int cg = C / 8;
  const int rpb = 256 / cg;
  const int col = threadIdx.x % cg;
  const int rl = threadIdx.x / cg;
  const int nseg = (W + DW_STRIP - 1) / DW_STRIP;
  const int nstrip = N * H * nseg;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  if (rl < rpb) {
    for (int sid = blockIdx.x * rpb + rl; sid < nstrip; sid += gridDim.x * rpb) {
      const int seg = sid % nseg;
      const int row = sid / nseg;
      const int oh = row % H;
      const int ow0 = seg * DW_STRIP;
      const int len = W - ow0 < DW_STRIP ? W - ow0 : DW_STRIP;
      const bool rv[3] = {oh > 0, true, oh + 1 < H};
      const size_t off = ((size_t)row * W + ow0) * C + col * 8;
      const u16* base = x + off;
      const u16* dyp = dy + off;
      const int rstep = W * C;
      uint4 m1[3], c0[3], p1[3], nx[3];
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const u16* rp = base + (r - 1) * rstep;
        m1[r] = (rv[r] && ow0 > 0) ? *(const uint4*)(rp - C) : z;
        c0[r] = rv[r] ? *(const uint4*)rp : z;
        p1[r] = (rv[r] && ow0 + 1 < W) ? *(const uint4*)(rp + C) : z;
      }
      uint4 dv = *(const uint4*)dyp, dn = z;
      for (int i = 0; i < len; ++i) {
        const bool cv = ow0 + i + 2 < W && i + 1 < len;
#pragma unroll
        for (int r = 0; r < 3; ++r) nx[r] = (rv[r] && cv) ? *(const uint4*)(base + (r - 1) * rstep + (i + 2) * C) : z;
        if (i + 1 < len) dn = *(const uint4*)(dyp + (size_t)(i + 1) * C);
        float d[8];
        dw_unpack(dv, d);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float f0[8], f1[8], f2[8];
          dw_unpack(m1[r], f0);
          dw_unpack(c0[r], f1);
          dw_unpack(p1[r], f2);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[r * 3 + 0][j] += d[j] * f0[j];
            acc[r * 3 + 1][j] += d[j] * f1[j];
            acc[r * 3 + 2][j] += d[j] * f2[j];
          }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          m1[r] = c0[r];
          c0[r] = p1[r];
          p1[r] = nx[r];
        }
        dv = dn;
      }
    }
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) sh[i] = 0.f;
  __syncthreads();
  const bool pow2 = (cg & (cg - 1)) == 0 && cg <= 32;
  if (pow2) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = rl < rpb ? acc[t][j] : 0.f;
        if (cg <= 32) v = lane_step_sum<32>(v);
        if (cg <= 16) v = lane_step_sum<16>(v);
        if (cg <= 8) v = lane_step_sum<8>(v);
        if (cg <= 4) v = lane_step_sum<4>(v);
        if (cg <= 2) v = lane_step_sum<2>(v);
        if (cg <= 1) v = lane_step_sum<1>(v);
        acc[t][j] = v;
      }
  }
  if (pow2 ? (threadIdx.x & 63) < cg : rl < rpb) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&sh[t * C + col * 8 + j], acc[t][j]);
  }
  __syncthreads();
  if (part) {
    for (int i = threadIdx.x; i < 9 * C; i += 256) part[(size_t)blockIdx.x * 9 * C + i] = sh[i];
    return;
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) {
    const int t = i / C, c = i - t * C;
    atomicAdd(&dw[(size_t)c * 9 + t], sh[i]);
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

// workgroups of the strip kernels: one strip (a run of up to DW_STRIP output pixels of a row) per thread and iteration
// `cap`: the forward kernel with statistics and the weight gradient end in per-block reductions (LDS + 2 C / 9 C global
// adds), so they run with fewer, longer-lived blocks
int dw_strip_blocks(const vlsfr_conv_desc* d, int cap = 4096) {
  const int rpb = 256 / (d->Cin / 8);
  const int64_t nstrip = (int64_t)d->N * d->H * ((d->W + DW_STRIP - 1) / DW_STRIP);
  int64_t b = (nstrip + rpb - 1) / rpb;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

int dw_blocks(int64_t P, int C) {
  const int rpb = 256 / (C / 8);
  int64_t b = (P + (int64_t)rpb * 4 - 1) / ((int64_t)rpb * 4);
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

}  // namespace

extern "C" {

int vlsfr_dwconv_fwd(const vlsfr_conv_desc* d, const void* x, const float* w, void* y, double* stats, void* stream) {
  int rc = dw_check(d, "vlsfr_dwconv_fwd");
  if (rc) return rc;
  if (!x || !w || !y) return fail(VLSFR_EINVAL, "vlsfr_dwconv_fwd: null buffer");
  DwArgs a{(const u16*)x, w, (u16*)y, d->N, d->H, d->W, d->Cin, odim(d->H, d->R, d->stride, d->pad),
           odim(d->W, d->S, d->stride, d->pad), d->R, d->stride, d->pad, 0, stats};
  a.repl = vlsfr::g_bn_repl;
  const int64_t P = (int64_t)a.N * a.Ho * a.Wo;
  const dim3 grid(dw_blocks(P, a.C)), block(256);
  const size_t shb = stats ? 2 * a.C * sizeof(float) : 0;
  hipStream_t st = (hipStream_t)stream;
  if (d->R == 3 && d->pad == 1) {
    if (d->stride == 1 && vlsfr::g_dw_strip) {
      const dim3 sgrid(dw_strip_blocks(d, stats ? 512 : 4096));
      if (stats) hipLaunchKernelGGL((dw3_strip_kernel<false, true>), sgrid, block, shb, st, a);
      else hipLaunchKernelGGL((dw3_strip_kernel<false, false>), sgrid, block, shb, st, a);
    } else if (d->stride == 1) {
      if (stats) hipLaunchKernelGGL((dw3_kernel<1, false, true>), grid, block, shb, st, a);
      else hipLaunchKernelGGL((dw3_kernel<1, false, false>), grid, block, shb, st, a);
    } else {
      if (stats) hipLaunchKernelGGL((dw3_kernel<2, false, true>), grid, block, shb, st, a);
      else hipLaunchKernelGGL((dw3_kernel<2, false, false>), grid, block, shb, st, a);
    }
  } else if (a.Ho == 1 && a.Wo == 1 && d->pad == 0 && d->R == d->H && d->S == d->W && a.C <= 512 && a.C % 8 == 0 && 256 / (a.C / 8) >= 4) {
    hipLaunchKernelGGL(dw_global_fwd_kernel, dim3(a.N), block, 4 * a.C * sizeof(float), st, a);
  } else {
    hipLaunchKernelGGL(dw_conv_kernel, grid, block, shb, st, a);
  }
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
  const dim3 grid(dw_blocks(P, a.C)), block(256);
  if (d->R == 3 && d->pad == 1) {
    if (d->stride == 1 && vlsfr::g_dw_strip)
      hipLaunchKernelGGL((dw3_strip_kernel<true, false>), dim3(dw_strip_blocks(d)), block, 0, (hipStream_t)stream, a);
    else if (d->stride == 1) hipLaunchKernelGGL((dw3_kernel<1, true, false>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dw3_kernel<2, true, false>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(dw_conv_kernel, grid, block, 0, (hipStream_t)stream, a);
  }
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_dgrad");
  return VLSFR_OK;
}

int vlsfr_dwconv_wgrad(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, void* stream) {
  return vlsfr_dwconv_wgrad_ws(d, dy, x, dw, nullptr, 0, stream);
}

size_t vlsfr_dwconv_wgrad_workspace_bytes(const vlsfr_conv_desc* d) {
  if (dw_check(d, "vlsfr_dwconv_wgrad_workspace_bytes") || d->R != 3 || d->pad != 1) return 0;
  return (size_t)DW_WGRAD_MAX_BLOCKS * 9 * d->Cin * sizeof(float);
}

int vlsfr_dwconv_wgrad_ws(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, void* workspace, size_t workspace_bytes,
                          void* stream) {
  int rc = dw_check(d, "vlsfr_dwconv_wgrad");
  if (rc) return rc;
  if (!dy || !x || !dw) return fail(VLSFR_EINVAL, "vlsfr_dwconv_wgrad: null buffer");
  const int Ho = odim(d->H, d->R, d->stride, d->pad), Wo = odim(d->W, d->S, d->stride, d->pad);
  const int64_t P = (int64_t)d->N * Ho * Wo;
  int nb = dw_blocks(P, d->Cin);
  if (d->R == 3 && d->pad == 1) {
    hipStream_t st = (hipStream_t)stream;
    const bool strip = d->stride == 1 && vlsfr::g_dw_strip;
    if (strip) nb = dw_strip_blocks(d, DW_WGRAD_MAX_BLOCKS);
    // with a workspace the blocks leave partial sums and a second kernel adds them up: the one-pass form ends in
    // 9 C atomics per block on the same 9 C addresses, which alone costs ~80 us per launch
    const int cap = workspace ? DW_WGRAD_MAX_BLOCKS : vlsfr::g_dw_wgrad_blocks;
    if (nb > cap) nb = cap;
    float* part = (workspace && workspace_bytes >= (size_t)nb * 9 * d->Cin * sizeof(float)) ? (float*)workspace : nullptr;
    if (!part && nb > vlsfr::g_dw_wgrad_blocks) nb = vlsfr::g_dw_wgrad_blocks;
    const size_t shb = 9 * d->Cin * sizeof(float);
    if (strip)
      hipLaunchKernelGGL(dw3_wgrad_strip_kernel, dim3(nb), dim3(256), shb, st, (const u16*)dy, (const u16*)x, dw, part, d->N, d->H, d->W,
                         d->Cin);
    else if (d->stride == 1)
      hipLaunchKernelGGL((dw3_wgrad_kernel<1>), dim3(nb), dim3(256), shb, st, (const u16*)dy, (const u16*)x, dw, part, d->N, d->H, d->W,
                         d->Cin, Ho, Wo);
    else
      hipLaunchKernelGGL((dw3_wgrad_kernel<2>), dim3(nb), dim3(256), shb, st, (const u16*)dy, (const u16*)x, dw, part, d->N, d->H, d->W,
                         d->Cin, Ho, Wo);
    VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_wgrad");
    if (part) {
      hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((9 * d->Cin + 255) / 256, DW_REDUCE_SLICES), dim3(256), 0, st, part, nb, d->Cin, dw);
      VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_wgrad reduce");
    }
    return VLSFR_OK;
  }
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(dw_wgrad_kernel, dim3(nb, d->R * d->S), dim3(256), d->Cin * sizeof(float), (hipStream_t)stream,
                     (const u16*)dy, (const u16*)x, dw, d->N, d->H, d->W, d->Cin, Ho, Wo, d->R, d->stride, d->pad);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_dwconv_wgrad");
  return VLSFR_OK;
}

}  // extern "C"
