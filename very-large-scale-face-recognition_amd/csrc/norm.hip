// Normalisation / activation / layout kernels of the backbones on gfx950 (HBM-bound, 16-byte
// vector accesses; C-ABI section 6 of include/vlsfr.h).  Replaces, forward and backward:
//   nn.BatchNorm2d in training mode (batch statistics, eps 1e-5, momentum 0.1; reference
//   model/resnet_arcface.py:35,37,40,75,93), nn.PReLU (:38,76), the residual add (:54),
//   the head BatchNorm1d with frozen weight + F.normalize (:96-98,151), and the input / weight
//   layout conversions (fp32 NCHW image -> bf16 im2col rows; fp32 master weights -> bf16 operands).
// Activations are NHWC bf16 ([M = N*H*W, C], C % 8 == 0); statistics and parameters are fp32.
#include "hip_common.h"

using namespace vlsfr;

namespace vlsfr {
// "bn_repl": replicated accumulators in use for the per-channel sums (statistics, backward reductions), <= VLSFR_BN_REPL
// (what the buffers are sized and zeroed for).  Every consumer block folds the replicas itself, so fewer replicas = less
// L2 traffic in front of every streaming loop (32 replicas: 64 KB of sums per block against 32 KB of payload at 256
// channels); more = less same-address contention of the producers' atomics.  8: 108.0 vs 112.5 ms per serial step.
int g_bn_repl = 8;
}

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
// Streaming skeleton shared by the BatchNorm kernels.  A thread owns ONE 16-byte channel group
// (8 channels) for the whole kernel, so per-channel constants sit in registers; a block covers a
// contiguous run of rows (~32 KB of x) with several independent 16-byte loads in flight per thread.
// Reductions: registers -> wave shuffle over the lanes that share a channel group -> LDS ->
// one of VLSFR_REPL replicated fp32 accumulators in global memory (replication keeps the
// same-address atomic contention of thousands of blocks off one cache line; a finalize kernel sums
// the replicas).
// ---------------------------------------------------------------------------------------------
constexpr int REPL = VLSFR_BN_REPL;   // replicas the accumulator buffers are SIZED for; vlsfr::g_bn_repl of them are used
constexpr int UF = 4;   // independent rows (16-byte loads per tensor) in flight per thread in the streaming loops

struct RowMap {
  int cg, col, rl, rpb;
  bool active;
};
__device__ __forceinline__ RowMap row_map(int C) {
  RowMap m;
  m.cg = C / 8;
  m.rpb = 256 / m.cg;
  m.col = threadIdx.x % m.cg;
  m.rl = threadIdx.x / m.cg;
  m.active = m.rl < m.rpb;
  return m;
}

// eight consecutive per-channel constants of a thread's channel group as two 16-byte loads (fill if the array is absent)
__device__ __forceinline__ void load8(const float* arr, int c0, float fill, float (&o)[8]) {
  if (arr) {
    const float4 lo = *(const float4*)(arr + c0), hi = *(const float4*)(arr + c0 + 4);
    o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w;
    o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fill;
  }
}

// sums over the block of per-thread partials v[NV][8] into out[rep][NV][C]
template <int NV>
__device__ __forceinline__ void block_reduce_to_replica(float (&v)[NV][8], const RowMap& m, int C, float* sh,
                                                        float* out, int repl) {
  const int tid = threadIdx.x;
  for (int i = tid; i < NV * C; i += 256) sh[i] = 0.f;
  __syncthreads();
  const bool pow2 = (m.cg & (m.cg - 1)) == 0;
  if (pow2 && m.cg <= 32) {
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = m.active ? v[n][j] : 0.f;   // lanes l, l + cg, l + 2 cg, ... hold the same channel group
        if (m.cg <= 32) t = lane_step_sum<32>(t);
        if (m.cg <= 16) t = lane_step_sum<16>(t);
        if (m.cg <= 8) t = lane_step_sum<8>(t);
        if (m.cg <= 4) t = lane_step_sum<4>(t);
        if (m.cg <= 2) t = lane_step_sum<2>(t);
        if (m.cg <= 1) t = lane_step_sum<1>(t);
        v[n][j] = t;
      }
    if ((tid & 63) < m.cg) {
#pragma unroll
      for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&sh[n * C + m.col * 8 + j], v[n][j]);
    }
  } else if (m.active) {
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&sh[n * C + m.col * 8 + j], v[n][j]);
  }
  __syncthreads();
  float* dst = out + (size_t)(blockIdx.x % repl) * NV * C;
  for (int i = tid; i < NV * C; i += 256) atomicAdd(&dst[i], sh[i]);
}

// ---- BatchNorm statistics (sum, sum of squares per channel) without the E[x^2] - mean^2 cancellation.
// A thread accumulates DEVIATIONS from the first value it sees per channel (fp32: the deviations are of the order of the
// spread, whatever the mean), turns them into (sum x, sum x^2) in float64 — exact to ~1e-16 of the mean^2 term — and from
// there on everything is added in float64: lanes that share a channel group, the block's LDS image, the replicated
// accumulators in global memory (global_atomic_add_f64), the fold in the consumer.  What is left of the cancellation is
// 1e-16 * mean^2 against the variance instead of 1e-7 * mean^2 (|mean| / sigma = 1e3: 1e-10 instead of 0.1).
struct StatAcc {
  float s[8], q[8], p[8];
  int n;
  __device__ __forceinline__ void init() {
    n = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = p[j] = 0.f;
  }
  __device__ __forceinline__ void add(const bf8& v) {
    if (n == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = v.get(j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v.get(j) - p[j];
      s[j] += d;
      q[j] += d * d;
    }
    ++n;
  }
};

// sums of the block into out[rep][2][C] (float64); shd: 2 * C doubles of LDS
__device__ __forceinline__ void block_stats_to_replica(const StatAcc& t, const RowMap& m, int C, double* shd, double* out, int repl) {
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * C; i += 256) shd[i] = 0.0;
  __syncthreads();
  double S[8], Q[8];
  const double n = m.active ? (double)t.n : 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double p = (double)t.p[j], s = m.active ? (double)t.s[j] : 0.0, q = m.active ? (double)t.q[j] : 0.0;
    S[j] = n * p + s;
    Q[j] = q + 2.0 * p * s + n * p * p;
  }
  const bool pow2 = (m.cg & (m.cg - 1)) == 0;
  if (pow2 && m.cg <= 32) {   // lanes l, l + cg, l + 2 cg, ... hold the same channel group
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int o = 32; o >= m.cg; o >>= 1) {
        S[j] += __shfl_xor(S[j], o, 64);
        Q[j] += __shfl_xor(Q[j], o, 64);
      }
    if ((tid & 63) < m.cg) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&shd[m.col * 8 + j], S[j]);
        atomicAdd(&shd[C + m.col * 8 + j], Q[j]);
      }
    }
  } else if (m.active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&shd[m.col * 8 + j], S[j]);
      atomicAdd(&shd[C + m.col * 8 + j], Q[j]);
    }
  }
  __syncthreads();
  double* dst = out + (size_t)(blockIdx.x % repl) * 2 * C;
  for (int i = tid; i < 2 * C; i += 256) atomicAdd(&dst[i], shd[i]);
}

__device__ __forceinline__ int rows_per_block(int C, int rpb) {
  int rb = 32768 / (C * 2);
  if (rb < rpb) rb = rpb;
  return (rb / rpb) * rpb;
}

// ---- statistics of x[M, C]: sums[REPL][2][C] float64 (sum, sum of squares), pre-zeroed
__global__ __launch_bounds__(256) void bn_stats_kernel(const u16* x, int64_t M, int C, int RB, double* sums, int repl) {
  extern __shared__ double shd[];
  const RowMap m = row_map(C);
  StatAcc t;
  t.init();
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int64_t r1 = r0 + RB < M ? r0 + RB : M;
  if (m.active) {
    const int nrow = (int)(r1 - r0);
    const u16* x0 = x + r0 * C + m.col * 8;
    for (int rl = m.rl; rl < nrow; rl += UF * m.rpb) {
      bf8 xv[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u)
        if (rl + u * m.rpb < nrow) xv[u].raw = *(const uint4*)(x0 + (rl + u * m.rpb) * C);
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        if (rl + u * m.rpb >= nrow) break;
        t.add(xv[u]);
      }
    }
  }
  block_stats_to_replica(t, m, C, shd, sums, repl);
}

// ---- y = prelu(bn(x)) + residual.  Every block folds the replicated statistics into
// scale = gamma*invstd / shift = beta - mean*scale itself (a few L2-resident loads per thread), so
// there is no separate finalize launch; block 0 also records mean / invstd and updates the running
// statistics.  Optionally accumulates the statistics of y for the next BatchNorm.
// Row range of a block.  Workgroups go to the 8 XCDs round-robin (id % 8); the LDS-DMA convolutions give every XCD one
// contiguous range of pixel tiles (conv.hip xcd_major_id), so with the same order here a pixel's rows are produced and
// consumed behind the same L2 all along the conv -> BatchNorm -> conv chain (option "bn_xcd", A/B in DESIGN.md section 8).
__device__ __forceinline__ int bn_block_id(int xcd) {
  const int id = blockIdx.x, n = gridDim.x;
  if (!xcd) return id;
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

struct BnApplyArgs {
  const u16* x;
  u16* y;
  int64_t M;
  int C, HW, RB;
  const double* sums;     // [REPL][2][C] statistics of x (float64 sums)
  const float* gamma;
  const float* beta;
  const float* slope;     // PReLU or nullptr
  const u16* residual;    // or nullptr
  float* save_mean;       // [C]
  float* save_invstd;     // [C]
  float* running_mean;    // or nullptr
  float* running_var;
  float eps, momentum;
  double* out_sums;       // [REPL][2][C] statistics of y (float64, pre-zeroed) or nullptr
  int out_nchw;           // 1: y index = n*(C*HW) + c*HW + hw (the flatten order of the reference's fc input)
  int xcd;                // 1: block -> row range in XCD-major order (bn_block_id)
  int repl;               // replicas in use (vlsfr::g_bn_repl)
};

// The per-channel part of bn_apply on its own (vlsfr_bn_finalize): statistics -> mean / invstd (saved), scale / shift, running
// statistics.  For a BatchNorm whose element-wise part runs inside the consumer convolution (vlsfr_conv2d_fwd_bnin).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* sums, int64_t M, int C, int repl, const float* gamma, const float* beta,
                                                          float eps, float momentum, float* save_mean, float* save_invstd, float* scale,
                                                          float* shift, float* running_mean, float* running_var) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    double sd = 0.0, qd = 0.0;
#pragma unroll 8
    for (int r = 0; r < repl; ++r) {
      sd += sums[(size_t)r * 2 * C + c];
      qd += sums[(size_t)r * 2 * C + C + c];
    }
    const double mean_d = sd / (double)M;
    double var_d = qd / (double)M - mean_d * mean_d;
    var_d = var_d > 0.0 ? var_d : 0.0;
    const float mean = (float)mean_d;
    const float var = (float)var_d;
    const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - mean * g * invstd;
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (running_mean) {
      const float unb = M > 1 ? var * (float)M / (float)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
  }
}

// FLAGS (compile time, so that the streaming loop is one straight-line block): 1 PReLU, 2 residual,
// 4 statistics of y, 8 flatten-order (NCHW) output, 16 ReLU after the residual add (with 2; not with 1 / 4)
template <int FLAGS>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a) {
  constexpr bool PRELU = FLAGS & 1, RESID = FLAGS & 2, OSUMS = FLAGS & 4, NCHW = FLAGS & 8, RELU_AFTER = FLAGS & 16;
  extern __shared__ __attribute__((aligned(8))) float sh[];   // scale[C], shift[C]; reused (as 2 C doubles) by the output-statistics reduction
  const int C = a.C;
  const int tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    double sd = 0.0, qd = 0.0;
#pragma unroll 8
    for (int r = 0; r < a.repl; ++r) {
      sd += a.sums[(size_t)r * 2 * C + c];
      qd += a.sums[(size_t)r * 2 * C + C + c];
    }
    const double mean_d = sd / (double)a.M;
    double var_d = qd / (double)a.M - mean_d * mean_d;       // float64 sums: the cancellation costs 1e-16 * mean^2, not 1e-7
    var_d = var_d > 0.0 ? var_d : 0.0;
    const float mean = (float)mean_d;
    const float var = (float)var_d;
    const float invstd = (float)(1.0 / sqrt(var_d + (double)a.eps));
    const float g = a.gamma ? a.gamma[c] : 1.f;
    const float b = a.beta ? a.beta[c] : 0.f;
    sh[c] = g * invstd;
    sh[C + c] = b - mean * g * invstd;
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
  const RowMap m = row_map(C);
  float sc[8], sf[8], sl[8];
  StatAcc osum;
  osum.init();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = m.col * 8 + j;
    sc[j] = sh[c];
    sf[j] = sh[C + c];
  }
  load8(PRELU ? a.slope : nullptr, m.col * 8, 1.f, sl);
  __syncthreads();   // sh is reused below
  const int64_t r0 = (int64_t)bn_block_id(a.xcd) * a.RB;
  const int nrow = (int)((r0 + a.RB < a.M ? r0 + a.RB : a.M) - r0);
  if (m.active) {
    // UF independent rows per iteration (all loads issued before any is consumed); 32-bit offsets from
    // the block's first row
    const u16* x0 = a.x + r0 * C + m.col * 8;
    const u16* q0 = RESID ? a.residual + r0 * C + m.col * 8 : nullptr;
    u16* y0 = a.y + r0 * C + m.col * 8;
    for (int rl = m.rl; rl < nrow; rl += UF * m.rpb) {
      bf8 xv[UF], rs[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u)
        if (rl + u * m.rpb < nrow) {
          xv[u].raw = *(const uint4*)(x0 + (rl + u * m.rpb) * C);
          if (RESID) rs[u].raw = *(const uint4*)(q0 + (rl + u * m.rpb) * C);
        }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        if (rl + u * m.rpb >= nrow) break;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = xv[u].get(j) * sc[j] + sf[j];
          if (PRELU) z = z > 0.f ? z : z * sl[j];
          if (RESID) z += rs[u].get(j);
          if (RELU_AFTER) z = z > 0.f ? z : 0.f;   // torchvision-style block: relu(bn(x) + identity), resnet_std.py:97-105
          o[j] = z;
        }
        const uint4 packed = pack8(o);
        if (!NCHW) {
          *(uint4*)(y0 + (rl + u * m.rpb) * C) = packed;
        } else {
          const int64_t row = r0 + rl + u * m.rpb;
          const int64_t n = row / a.HW;
          const int hw = (int)(row - n * a.HW);
#pragma unroll
          for (int j = 0; j < 8; ++j) a.y[(n * C + m.col * 8 + j) * a.HW + hw] = f2bf(o[j]);
        }
        if (OSUMS) {
          bf8 yv;
          yv.raw = packed;   // statistics of what the next layer will actually read
          osum.add(yv);
        }
      }
    }
  }
  if (OSUMS) {
    __syncthreads();   // (threads without rows skipped the loop: everybody is past the last use of sh as scale / shift)
    block_stats_to_replica(osum, m, C, (double*)sh, a.out_sums, a.repl);
  }
}

// ---------------------------------------------------------------------------------------------
// backward: red[REPL][3][C] = sum dz, sum dz*xhat, sum dy*min(z,0) (PReLU slope gradient) with
// z = bn(x), dz = dy * prelu'(z); then k0 = gamma*invstd, k1 = mean dz, k2 = mean dz*xhat and
// dx = k0 * (dz - k1 - xhat*k2) (+ dx_add)
// ---------------------------------------------------------------------------------------------
struct BnBwdArgs {
  const u16* dy;
  const u16* x;           // BN input (conv output)
  u16* dx;
  int64_t M;
  int C, HW, RB;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* slope;     // or nullptr
  float* red;             // [REPL][3][C] pre-zeroed
  const u16* dx_add;      // extra gradient added to dx (identity branch) or nullptr
  float* dgamma;          // accumulated (+=); may be nullptr (frozen)
  float* dbeta;
  float* dslope;
  int dy_nchw;            // dy laid out as the flatten order (see bn_apply out_nchw)
  int xcd;                // 1: block -> row range in XCD-major order (bn_block_id)
  // apply kernel, FLAGS & 8: dx IS the dY of a following BatchNorm backward (same shape, no PReLU) whose input is nx — its
  // reduction (sum dY, sum dY * xhat) is accumulated here from the rounded dx and one extra read of nx, instead of a
  // separate kernel reading both tensors again
  const u16* nx;
  const float* n_mean;
  const float* n_invstd;
  float* n_red;           // [REPL][3][C], pre-zeroed
  int repl;               // replicas in use (vlsfr::g_bn_repl)
};

__device__ __forceinline__ float load_dy_nchw(const BnBwdArgs& a, int64_t r, int c) {
  const int64_t n = r / a.HW;
  const int hw = (int)(r - n * a.HW);
  return bf2f(a.dy[(n * a.C + c) * a.HW + hw]);
}

// FLAGS: 1 PReLU, 2 dy in flatten (NCHW) order, 4 dx_add (apply kernel only)
template <int FLAGS>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a) {
  constexpr bool PRELU = FLAGS & 1, NCHW = FLAGS & 2;
  extern __shared__ float sh[];
  const int C = a.C;
  const RowMap m = row_map(C);
  // the loop accumulates sum dz, sum dz*x (RAW x) and sum dy*z over z <= 0; sum dz*xhat follows as
  // invstd * (sum dz*x - mean * sum dz) -- fewer per-channel constants in registers.  z = x*zs + zo is only
  // needed for the PReLU sign and slope gradient.
  float v[3][8], zs[8], zo[8], sl[8], c_is[8], c_mean[8];
  {
    float cg_[8], cb_[8];
    load8(a.invstd, m.col * 8, 1.f, c_is);
    load8(a.mean, m.col * 8, 0.f, c_mean);
    load8(a.gamma, m.col * 8, 1.f, cg_);
    load8(a.beta, m.col * 8, 0.f, cb_);
    load8(PRELU ? a.slope : nullptr, m.col * 8, 1.f, sl);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[0][j] = v[1][j] = v[2][j] = 0.f;
      zs[j] = c_is[j] * cg_[j];
      zo[j] = cb_[j] - c_mean[j] * c_is[j] * cg_[j];
    }
  }
  const int64_t r0 = (int64_t)bn_block_id(a.xcd) * a.RB;
  const int nrow = (int)((r0 + a.RB < a.M ? r0 + a.RB : a.M) - r0);
  if (m.active) {
    const u16* x0 = a.x + r0 * C + m.col * 8;
    const u16* d0 = a.dy + r0 * C + m.col * 8;
    for (int rl = m.rl; rl < nrow; rl += UF * m.rpb) {
      bf8 xv[UF], dv[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u)
        if (rl + u * m.rpb < nrow) {
          xv[u].raw = *(const uint4*)(x0 + (rl + u * m.rpb) * C);
          if (!NCHW) dv[u].raw = *(const uint4*)(d0 + (rl + u * m.rpb) * C);
        }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        if (rl + u * m.rpb >= nrow) break;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xf = xv[u].get(j);
          const float dyv = NCHW ? load_dy_nchw(a, r0 + rl + u * m.rpb, m.col * 8 + j) : dv[u].get(j);
          float dz = dyv;
          if (PRELU) {
            const float z = xf * zs[j] + zo[j];
            const bool neg = z <= 0.f;
            v[2][j] += neg ? dyv * z : 0.f;
            dz = neg ? dyv * sl[j] : dyv;
          }
          v[0][j] += dz;
          v[1][j] += dz * (xf - c_mean[j]);   // centred: no cancellation against mean * sum dz when |mean| >> sigma
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[1][j] *= c_is[j];   // sum dz * xhat
  block_reduce_to_replica<3>(v, m, C, sh, a.red, a.repl);
}

template <int FLAGS>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
  constexpr bool PRELU = FLAGS & 1, NCHW = FLAGS & 2, DXADD = FLAGS & 4, NEXT = FLAGS & 8;
  extern __shared__ float sh[];   // k0[C] = gamma*invstd, k1[C] = mean dz, k2[C] = mean dz*xhat
  const int C = a.C;
  const int tid = threadIdx.x;
  const float invM = 1.f / (float)a.M;
  for (int c = tid; c < C; c += 256) {   // fold the replicated reductions (no separate finalize launch)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int r = 0; r < a.repl; ++r) {
      const float* p = a.red + (size_t)r * 3 * C;
      s0 += p[c];
      s1 += p[C + c];
      s2 += p[2 * C + c];
    }
    sh[c] = (a.gamma ? a.gamma[c] : 1.f) * a.invstd[c];
    sh[C + c] = s0 * invM;
    sh[2 * C + c] = s1 * invM;
    if (blockIdx.x == 0) {
      // atomics: the two backward passes of a step may run concurrently on two streams (model/_native.py)
      if (a.dbeta) atomicAdd(&a.dbeta[c], s0);
      if (a.dgamma) atomicAdd(&a.dgamma[c], s1);
      if (a.dslope && a.slope) atomicAdd(&a.dslope[c], s2);
    }
  }
  __syncthreads();
  const RowMap m = row_map(C);
  if (!NEXT && !m.active) return;
  float nv[3][8];   // NEXT: sum dx, sum dx * (nx - mean of nx), unused
  float c_is2[8], c_mean2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) nv[0][j] = nv[1][j] = nv[2][j] = 0.f;
  load8(NEXT ? a.n_invstd : nullptr, m.col * 8, 1.f, c_is2);
  load8(NEXT ? a.n_mean : nullptr, m.col * 8, 0.f, c_mean2);
  if (m.active) {
  // dx = k0*(dz - k1 - xhat*k2) = A*dz + Bx*x + Cc with xhat = x*invstd - mean*invstd folded in;
  // dz = dy * (z <= 0 ? slope : 1): As = A*slope
  float A[8], As[8], Bx[8], Cc[8], zs[8], zo[8];
  {
    float c_is[8], c_mean[8], cg_[8], cb_[8], csl[8];
    load8(a.invstd, m.col * 8, 1.f, c_is);
    load8(a.mean, m.col * 8, 0.f, c_mean);
    load8(a.gamma, m.col * 8, 1.f, cg_);
    load8(a.beta, m.col * 8, 0.f, cb_);
    load8(PRELU ? a.slope : nullptr, m.col * 8, 1.f, csl);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = m.col * 8 + j;
      const float k0 = sh[c], k1 = sh[C + c], k2 = sh[2 * C + c];
      const float is = c_is[j], xo = -c_mean[j] * is, g = cg_[j];
      A[j] = k0;
      As[j] = k0 * csl[j];
      Bx[j] = -k0 * k2 * is;
      Cc[j] = -k0 * (k1 + k2 * xo);
      zs[j] = is * g;
      zo[j] = cb_[j] + xo * g;
    }
  }
  const int64_t r0 = (int64_t)bn_block_id(a.xcd) * a.RB;
  const int nrow = (int)((r0 + a.RB < a.M ? r0 + a.RB : a.M) - r0);
  const u16* x0 = a.x + r0 * C + m.col * 8;
  const u16* d0 = a.dy + r0 * C + m.col * 8;
  const u16* q0 = DXADD ? a.dx_add + r0 * C + m.col * 8 : nullptr;
  const u16* n0 = NEXT ? a.nx + r0 * C + m.col * 8 : nullptr;
  u16* o0 = a.dx + r0 * C + m.col * 8;
  for (int rl = m.rl; rl < nrow; rl += UF * m.rpb) {
    bf8 xv[UF], dv[UF], av[UF], nxv[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u)
      if (rl + u * m.rpb < nrow) {
        xv[u].raw = *(const uint4*)(x0 + (rl + u * m.rpb) * C);
        if (!NCHW) dv[u].raw = *(const uint4*)(d0 + (rl + u * m.rpb) * C);
        if (DXADD) av[u].raw = *(const uint4*)(q0 + (rl + u * m.rpb) * C);
        if (NEXT) nxv[u].raw = *(const uint4*)(n0 + (rl + u * m.rpb) * C);
      }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (rl + u * m.rpb >= nrow) break;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xf = xv[u].get(j);
        const float dyv = NCHW ? load_dy_nchw(a, r0 + rl + u * m.rpb, m.col * 8 + j) : dv[u].get(j);
        float ad = A[j];
        if (PRELU) ad = (xf * zs[j] + zo[j] <= 0.f) ? As[j] : A[j];
        float d = ad * dyv + (Bx[j] * xf + Cc[j]);
        if (DXADD) d += av[u].get(j);
        o[j] = d;
      }
      bf8 ov;
      ov.raw = pack8(o);
      *(uint4*)(o0 + (rl + u * m.rpb) * C) = ov.raw;
      if (NEXT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dz = ov.get(j);            // the value the following BatchNorm reads back
          nv[0][j] += dz;
          nv[1][j] += dz * (nxv[u].get(j) - c_mean2[j]);   // centred, as in bn_bwd_reduce_kernel
        }
      }
    }
  }
  }   // m.active
  if (NEXT) {
#pragma unroll
    for (int j = 0; j < 8; ++j) nv[1][j] *= c_is2[j];
    __syncthreads();   // sh (k0, k1, k2) has been consumed by every thread
    block_reduce_to_replica<3>(nv, m, C, sh, a.n_red, a.repl);
  }
}

// y = a + b (bf16 tensors): sum of two gradient branches
// zero fill as a KERNEL (not hipMemsetAsync): the executors' accumulator regions are cleared at the head of every pass, and a
// pass may be replayed from a HIP graph (NativeBackbone.use_graphs) — with memset nodes in the captured chain the replays
// produced intermittently wrong BatchNorm statistics (scripts/graph_debug.py), with a kernel node they do not
__global__ __launch_bounds__(256) void zero_kernel(uint4* p, int64_t n16, char* tail, int ntail) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

// device-to-device copy as a KERNEL (16-byte granules): the executors hand the embedding to the caller with it, so that a pass
// captured into a HIP graph holds kernel nodes only (no runtime copy nodes: ADVICE r03)
__global__ __launch_bounds__(256) void copy_kernel(const uint4* src, uint4* dst, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

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

// 32 features x 8 row groups per block: a thread walks every 8th row of its feature (coalesced 128-byte rows per
// 32 lanes, four rows in flight), the row groups meet in LDS in a fixed order.  (One thread per feature over all B
// rows paid a memory latency per row: 95 us for a 512 KB tensor.)
__global__ __launch_bounds__(256) void embed_bn_kernel(EmbedArgs a) {
  __shared__ float sh[2][8][32];
  const int f = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + f;
  const bool ok = d < a.D;
  const float bias = ok ? a.fc_bias[d] : 0.f;
  float s = 0.f, q = 0.f;
  if (ok) {
#pragma unroll 4
    for (int b = rg; b < a.B; b += 8) {
      const float v = a.fc[(size_t)b * a.D + d] + bias;
      s += v;
      q += v * v;
    }
  }
  sh[0][rg][f] = s;
  sh[1][rg][f] = q;
  __syncthreads();
  s = q = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    s += sh[0][r][f];
    q += sh[1][r][f];
  }
  if (!ok) return;
  const float mean = s / a.B;
  float var = q / a.B - mean * mean;
  var = var > 0.f ? var : 0.f;
  const float is = rsqrtf(var + a.eps);
  if (rg == 0) {
    a.invstd[d] = is;
    if (a.running_mean) {
      const float unb = a.B > 1 ? var * a.B / (a.B - 1) : var;
      a.running_mean[d] = (1.f - a.momentum) * a.running_mean[d] + a.momentum * mean;
      a.running_var[d] = (1.f - a.momentum) * a.running_var[d] + a.momentum * unb;
    }
  }
  const float g = a.gamma[d], be = a.beta[d];
#pragma unroll 4
  for (int b = rg; b < a.B; b += 8) {
    const size_t i = (size_t)b * a.D + d;
    const float xh = (a.fc[i] + bias - mean) * is;
    a.xhat[i] = xh;
    a.z[i] = xh * g + be;
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
  float* dfc_bias;        // fc.bias grad (+=) or nullptr
  float* dgamma;          // BN weight grad (+=) or nullptr (frozen in the iResNet tail)
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

__global__ __launch_bounds__(256) void embed_bn_bwd_kernel(EmbedBwdArgs a) {   // same decomposition as embed_bn_kernel
  __shared__ float sh[3][8][32];
  const int f = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + f;
  const bool ok = d < a.D;
  float s0 = 0.f, s1 = 0.f;
  if (ok) {
#pragma unroll 4
    for (int b = rg; b < a.B; b += 8) {
      const size_t i = (size_t)b * a.D + d;
      const float dzv = a.dz[i];
      s0 += dzv;
      s1 += dzv * a.xhat[i];
    }
  }
  sh[0][rg][f] = s0;
  sh[1][rg][f] = s1;
  __syncthreads();
  s0 = s1 = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    s0 += sh[0][r][f];
    s1 += sh[1][r][f];
  }
  float sb = 0.f;
  if (ok) {
    const float g = a.gamma[d], is = a.invstd[d];
    if (rg == 0) {
      atomicAdd(&a.dbeta[d], s0);
      if (a.dgamma) atomicAdd(&a.dgamma[d], s1);
    }
    const float m0 = g * s0 / a.B, m1 = g * s1 / a.B;
#pragma unroll 4
    for (int b = rg; b < a.B; b += 8) {
      const size_t i = (size_t)b * a.D + d;
      const float dv = is * (g * a.dz[i] - m0 - a.xhat[i] * m1);
      a.dfc[i] = f2bf(dv);
      sb += dv;
    }
  }
  sh[2][rg][f] = sb;
  __syncthreads();
  if (ok && rg == 0 && a.dfc_bias) {
    sb = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sb += sh[2][r][f];
    atomicAdd(&a.dfc_bias[d], sb);
  }
}

// ---------------------------------------------------------------------------------------------
// layout / precision conversions
// ---------------------------------------------------------------------------------------------
// Weight operand copies (cast_weights_kernel below): fp32 [rows][K] -> bf16 [rows][Kp] (zero padded) and, optionally,
// the transpose used by dgrad, wT[c][tap][row] for w[row][tap][c]  (rows = Cout, C = Cin, taps = R*S)
// stem: fp32 NCHW image [N,3,H,W] -> bf16 im2col rows [N*Ho*Wo][32], k = (r*3 + s)*3 + c, 3x3 pad 1, stride 1 or 2
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* x, u16* out, int N, int H, int W, int stride) {
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int64_t P = (int64_t)N * Ho * Wo;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int n = (int)(p / (Ho * Wo));
    const int rem = (int)(p - (int64_t)n * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hi = ho * stride + r - 1, wi = wo * stride + s - 1;
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

// Loader transform on the device (reference util/lmdb_loader.py:109-127, :206-233): decoded uint8 pixels
// [N][H][W][C] (C = 3: BGR as cv2.imdecode delivers them, C = 1: grey, replicated to three planes) ->
// fp32 [N][3][H][W] = (v - 127.5) * 0.0078125, mirrored left-right where flip[n] != 0 (cv2.flip(img, 1)).
// One thread per 4 output pixels of one plane row: 16-byte stores, the uint8 reads of a row stay in L1/L2.
__global__ __launch_bounds__(256) void faces_normalize_kernel(const uint8_t* raw, const uint8_t* flip, float* out, int N, int H,
                                                              int W, int C) {
  const int W4 = (W + 3) / 4;
  const int64_t total = (int64_t)N * 3 * H * W4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int w4 = (int)(i % W4);
    int64_t r = i / W4;
    const int h = (int)(r % H);
    r /= H;
    const int c = (int)(r % 3);
    const int n = (int)(r / 3);
    const bool fl = flip && flip[n];
    const uint8_t* src = raw + ((size_t)n * H + h) * W * C + (C == 3 ? c : 0);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = w4 * 4 + j;
      const int ws = fl ? W - 1 - w : w;
      v[j] = w < W ? ((float)src[(size_t)ws * C] - 127.5f) * 0.0078125f : 0.f;
    }
    float* dst = out + (((size_t)n * 3 + c) * H + h) * W + w4 * 4;
    if (w4 * 4 + 3 < W && ((W & 3) == 0)) *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
    else
      for (int j = 0; j < 4 && w4 * 4 + j < W; ++j) dst[j] = v[j];
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

int g_bn_xcd = 0;                // "bn_xcd": 1 = XCD-major block order in the BatchNorm kernels (bn_block_id); measured: no change
                                 // (91.7 ms either way at ir100 / batch 256 — lines do not survive in L2 across kernel boundaries)
int g_bn_block_bytes = 65536;   // bytes of x per block (vlsfr_set_option("bn_block_kb", v))

static int bn_geom(int64_t M, int C, int* RB, int* nblk) {
  const int cg = C / 8;
  const int rpb = 256 / cg;
  if (rpb < 1) return -1;
  int rb = g_bn_block_bytes / (C * 2);
  if (rb < rpb) rb = rpb;
  rb = (rb / rpb) * rpb;
  *RB = rb;
  *nblk = (int)((M + rb - 1) / rb);
  return 0;
}

int vlsfr_bn_stats(const void* x, int64_t M, int32_t C, double* sums, void* stream) {
  if (!x || !sums || M <= 0 || C <= 0 || C % 8 || C > 2048)
    return fail(VLSFR_EINVAL, "vlsfr_bn_stats: need C %% 8 == 0, C <= 2048 (got %d)", C);
  int RB, nblk;
  bn_geom(M, C, &RB, &nblk);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(256), 2 * C * sizeof(double), (hipStream_t)stream,
                     (const u16*)x, M, C, RB, sums, vlsfr::g_bn_repl);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_stats");
  return VLSFR_OK;
}

int vlsfr_bn_apply(const void* x, void* y, int64_t M, int32_t C, int32_t HW, const double* sums, const float* gamma,
                   const float* beta, const float* slope, const void* residual, float* save_mean,
                   float* save_invstd, float* running_mean, float* running_var, float eps, float momentum,
                   double* out_sums, int32_t out_nchw, void* stream) {
  if (!x || !y || !sums || !save_mean || !save_invstd || M <= 0 || C <= 0 || C % 8 || C > 2048 || HW <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_bn_apply: bad argument");
  const bool relu_after = (out_nchw & 2) != 0;   // bit 1 of out_nchw: y = relu(bn(x) + residual)
  out_nchw &= 1;
  if (relu_after && (!residual || slope || out_sums))
    return fail(VLSFR_EINVAL, "vlsfr_bn_apply: relu-after-add needs a residual and neither PReLU nor output statistics");
  int RB, nblk;
  bn_geom(M, C, &RB, &nblk);
  BnApplyArgs a{(const u16*)x, (u16*)y, M, C, HW, RB, sums, gamma, beta, slope, (const u16*)residual, save_mean,
                save_invstd, running_mean, running_var, eps, momentum, out_sums, out_nchw, g_bn_xcd && nblk >= 16, vlsfr::g_bn_repl};
  const int flags = (slope ? 1 : 0) | (residual ? 2 : 0) | (out_sums ? 4 : 0) | (out_nchw ? 8 : 0) | (relu_after ? 16 : 0);
  const dim3 grid(nblk), block(256);
  const size_t shb = 2 * C * (out_sums ? sizeof(double) : sizeof(float));
  hipStream_t st = (hipStream_t)stream;
#define VLSFR_CASE(F) case F: hipLaunchKernelGGL(bn_apply_kernel<F>, grid, block, shb, st, a); break;
  switch (flags) {
    VLSFR_CASE(0) VLSFR_CASE(1) VLSFR_CASE(2) VLSFR_CASE(3) VLSFR_CASE(4) VLSFR_CASE(5) VLSFR_CASE(6) VLSFR_CASE(7)
    VLSFR_CASE(8) VLSFR_CASE(9) VLSFR_CASE(10) VLSFR_CASE(11) VLSFR_CASE(12) VLSFR_CASE(13) VLSFR_CASE(14) VLSFR_CASE(15)
    VLSFR_CASE(18) VLSFR_CASE(26)
  }
#undef VLSFR_CASE
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_apply");
  return VLSFR_OK;
}

int vlsfr_bn_finalize(const double* sums, int64_t M, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                      float* save_mean, float* save_invstd, float* scale, float* shift, float* running_mean, float* running_var,
                      void* stream) {
  if (!sums || !save_mean || !save_invstd || !scale || !shift || M <= 0 || C <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_bn_finalize: bad argument");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, M, C, vlsfr::g_bn_repl, gamma, beta,
                     eps, momentum, save_mean, save_invstd, scale, shift, running_mean, running_var);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_finalize");
  return VLSFR_OK;
}

int vlsfr_bn_backward(const void* dy, const void* x, void* dx, int64_t M, int32_t C, int32_t HW, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, const float* slope, float* red,
                      const void* dx_add, float* dgamma, float* dbeta, float* dslope, int32_t dy_nchw, void* stream) {
  return vlsfr_bn_backward_chain(dy, x, dx, M, C, HW, mean, invstd, gamma, beta, slope, red, dx_add, dgamma, dbeta, dslope, dy_nchw, 0,
                                 nullptr, nullptr, nullptr, nullptr, stream);
}

int vlsfr_bn_backward_chain(const void* dy, const void* x, void* dx, int64_t M, int32_t C, int32_t HW, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, const float* slope, float* red,
                            const void* dx_add, float* dgamma, float* dbeta, float* dslope, int32_t dy_nchw, int32_t red_ready,
                            const void* next_x, const float* next_mean, const float* next_invstd, float* next_red, void* stream) {
  if (!dy || !x || !dx || !mean || !invstd || !red || M <= 0 || C <= 0 || C % 8 || C > 2048 || HW <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_bn_backward: bad argument");
  if (next_x && (!next_mean || !next_invstd || !next_red))
    return fail(VLSFR_EINVAL, "vlsfr_bn_backward_chain: next_x needs next_mean, next_invstd and next_red");
  hipStream_t st = (hipStream_t)stream;
  int RB, nblk;
  bn_geom(M, C, &RB, &nblk);
  BnBwdArgs a{(const u16*)dy, (const u16*)x, (u16*)dx, M, C, HW, RB, mean, invstd, gamma, beta, slope, red,
              (const u16*)dx_add, dgamma, dbeta, dslope, dy_nchw, g_bn_xcd && nblk >= 16,
              (const u16*)next_x, next_mean, next_invstd, next_red, vlsfr::g_bn_repl};
  const int rflags = (slope ? 1 : 0) | (dy_nchw ? 2 : 0);
  const int aflags = rflags | (dx_add ? 4 : 0) | (next_x ? 8 : 0);
  const dim3 grid(nblk), block(256);
  const size_t shb = 3 * C * sizeof(float);
  if (!red_ready) {   // red_ready: the producer of dy accumulated this layer's reduction already (next_* of its call)
    switch (rflags) {
      case 0: hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, grid, block, shb, st, a); break;
      case 1: hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, grid, block, shb, st, a); break;
      case 2: hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, grid, block, shb, st, a); break;
      default: hipLaunchKernelGGL(bn_bwd_reduce_kernel<3>, grid, block, shb, st, a); break;
    }
    VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_backward reduce");
  }
#define VLSFR_CASE(F) case F: hipLaunchKernelGGL(bn_bwd_apply_kernel<F>, grid, block, shb, st, a); break;
  switch (aflags) {
    VLSFR_CASE(0) VLSFR_CASE(1) VLSFR_CASE(2) VLSFR_CASE(3) VLSFR_CASE(4) VLSFR_CASE(5) VLSFR_CASE(6) VLSFR_CASE(7)
    VLSFR_CASE(8) VLSFR_CASE(12)          // the chain form: plain BatchNorm, with / without the identity-branch gradient
    default: return fail(VLSFR_EINVAL, "vlsfr_bn_backward_chain: next_x goes with a plain BatchNorm (no PReLU, NHWC dy)");
  }
#undef VLSFR_CASE
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_backward apply");
  return VLSFR_OK;
}

int vlsfr_bn_backward_reduce(const void* dy, const void* x, int64_t M, int32_t C, int32_t HW, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, const float* slope, float* red,
                             void* stream) {
  if (!dy || !x || !mean || !invstd || !red || M <= 0 || C <= 0 || C % 8 || C > 2048 || HW <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_bn_backward_reduce: bad argument");
  int RB, nblk;
  bn_geom(M, C, &RB, &nblk);
  BnBwdArgs a{(const u16*)dy, (const u16*)x, nullptr, M, C, HW, RB, mean, invstd, gamma, beta, slope, red,
              nullptr, nullptr, nullptr, nullptr, 0, g_bn_xcd && nblk >= 16, nullptr, nullptr, nullptr, nullptr, vlsfr::g_bn_repl};
  const dim3 grid(nblk), block(256);
  const size_t shb = 3 * C * sizeof(float);
  if (slope) hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, grid, block, shb, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, grid, block, shb, (hipStream_t)stream, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_bn_backward_reduce");
  return VLSFR_OK;
}

int vlsfr_zero_bytes(void* p, size_t nbytes, void* stream) {
  if (!p || ((uintptr_t)p & 15)) return fail(VLSFR_EINVAL, "vlsfr_zero_bytes: need a 16-byte aligned pointer");
  if (nbytes == 0) return VLSFR_OK;
  const int64_t n16 = (int64_t)(nbytes / 16);
  const int ntail = (int)(nbytes % 16);
  int64_t blocks = (n16 + 256 * 4 - 1) / (256 * 4);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)p, n16, (char*)p + n16 * 16, ntail);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_zero_bytes");
  return VLSFR_OK;
}

int vlsfr_copy_bytes(const void* src, void* dst, size_t nbytes, void* stream) {
  if (!src || !dst || (((uintptr_t)src | (uintptr_t)dst | nbytes) & 15))
    return fail(VLSFR_EINVAL, "vlsfr_copy_bytes: 16-byte aligned pointers and size");
  if (nbytes == 0) return VLSFR_OK;
  const int64_t n16 = (int64_t)(nbytes / 16);
  int64_t blocks = (n16 + 256 * 4 - 1) / (256 * 4);
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_copy_bytes");
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
  hipLaunchKernelGGL(embed_bn_kernel, dim3((D + 31) / 32), dim3(256), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_fwd bn");
  hipLaunchKernelGGL(embed_norm_kernel, dim3(B), dim3(256), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_fwd norm");
  return VLSFR_OK;
}

int vlsfr_embed_bwd(const float* demb, const float* emb, const float* inv_norm, const float* xhat, const float* invstd,
                    const float* gamma, float* dz, void* dfc_bf16, float* dbeta, float* dfc_bias, float* dgamma,
                    int32_t B, int32_t D, void* stream) {
  if (!demb || !emb || !inv_norm || !xhat || !invstd || !gamma || !dz || !dfc_bf16 || !dbeta || B <= 0 || D <= 0)
    return fail(VLSFR_EINVAL, "vlsfr_embed_bwd: bad argument");
  EmbedBwdArgs a{demb, emb, inv_norm, xhat, invstd, gamma, dz, (u16*)dfc_bf16, dbeta, dfc_bias, dgamma, B, D};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_norm_bwd_kernel, dim3(B), dim3(256), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_bwd norm");
  hipLaunchKernelGGL(embed_bn_bwd_kernel, dim3((D + 31) / 32), dim3(256), 0, st, a);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_embed_bwd bn");
  return VLSFR_OK;
}

}  // extern "C"

namespace {
// All weight tensors of a network in one launch (54 per iResNet-50: the per-tensor launches were ~12 us each of
// mostly launch latency): the table travels as the kernel argument, a block finds its tensor by a scan of the
// block prefix.
constexpr int CAST_BATCH = 64;
struct CastBatch {
  vlsfr_cast_entry e[CAST_BATCH];
  int first_block[CAST_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(256) void cast_weights_kernel(CastBatch cb) {
  int t = 0;
  while (t + 1 < cb.n && (int)blockIdx.x >= cb.first_block[t + 1]) ++t;
  const vlsfr_cast_entry& e = cb.e[t];
  const int K = e.taps * e.C;
  u16* wb = (u16*)e.w_bf16;
  u16* wT = (u16*)e.wT_bf16;
  const int blk = (int)blockIdx.x - cb.first_block[t];
  if (e.rows % 64 == 0 && e.C % 64 == 0 && e.Kp == K) {
    // tiled path: one block = 64 rows x 64 channels of one tap.  fp32 rows are read in 256-byte segments, the bf16
    // copy is written in 128-byte segments, and the transposed copy wT[c][tap][row] — 2-byte stores a whole weight
    // row apart in the element-wise kernel this replaces (0.6 TB/s; 1.7 ms per step at ir100) — goes through a
    // padded LDS tile so that its stores are 128-byte row segments too.
    __shared__ u16 tile[64][66];
    const int cblocks = e.C / 64;
    const int cb_i = blk % cblocks;
    const int tap = (blk / cblocks) % e.taps;
    const int rb = blk / (cblocks * e.taps);
    const int r0 = rb * 64, c0 = cb_i * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 rows per pass
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
      const size_t src = (size_t)(r0 + i) * K + (size_t)tap * e.C + c0 + tx;
      const u16 b = f2bf(e.w[src]);
      wb[src] = b;
      tile[i][tx] = b;
    }
    if (wT) {
      __syncthreads();
#pragma unroll 4
      for (int j = ty; j < 64; j += 4)   // channel c0 + j: 64 consecutive rows
        wT[((size_t)(c0 + j) * e.taps + tap) * e.rows + r0 + tx] = tile[tx][j];
    }
    return;
  }
  const int64_t total = (int64_t)e.rows * e.Kp;
  const int64_t i0 = (int64_t)blk * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = i0 + u * 256;
    if (i >= total) break;
    const int row = (int)(i / e.Kp);
    const int k = (int)(i - (int64_t)row * e.Kp);
    const float v = k < K ? e.w[(size_t)row * K + k] : 0.f;
    const u16 b = f2bf(v);
    wb[i] = b;
    if (wT && k < K) {
      const int tap = k / e.C, c = k - tap * e.C;
      wT[((size_t)c * e.taps + tap) * e.rows + row] = b;
    }
  }
}

}  // namespace

extern "C" {

int vlsfr_cast_weight(const float* w, void* w_bf16, void* wT_bf16, int32_t rows, int32_t taps, int32_t C, int32_t Kp,
                      void* stream) {
  if (!w || !w_bf16 || rows <= 0 || taps <= 0 || C <= 0 || Kp < taps * C)
    return fail(VLSFR_EINVAL, "vlsfr_cast_weight: bad argument");
  const vlsfr_cast_entry e{w, w_bf16, wT_bf16, rows, taps, C, Kp};
  return vlsfr_cast_weights(&e, 1, stream);
}

int vlsfr_cast_weights(const vlsfr_cast_entry* entries, int32_t n, void* stream) {
  if (!entries || n <= 0) return fail(VLSFR_EINVAL, "vlsfr_cast_weights: bad argument");
  for (int base = 0; base < n; base += CAST_BATCH) {
    CastBatch cb;
    cb.n = n - base < CAST_BATCH ? n - base : CAST_BATCH;
    int blocks = 0;
    for (int i = 0; i < cb.n; ++i) {
      const vlsfr_cast_entry& e = entries[base + i];
      if (!e.w || !e.w_bf16 || e.rows <= 0 || e.taps <= 0 || e.C <= 0 || e.Kp < e.taps * e.C)
        return fail(VLSFR_EINVAL, "vlsfr_cast_weights: bad entry %d", base + i);
      cb.e[i] = e;
      cb.first_block[i] = blocks;
      if (e.rows % 64 == 0 && e.C % 64 == 0 && e.Kp == e.taps * e.C) blocks += (e.rows / 64) * e.taps * (e.C / 64);   // tiled path
      else blocks += (int)(((int64_t)e.rows * e.Kp + 1023) / 1024);
    }
    cb.first_block[cb.n] = blocks;
    hipLaunchKernelGGL(cast_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cb);
    VLSFR_HIP_CHECK_LAUNCH("vlsfr_cast_weights");
  }
  return VLSFR_OK;
}

int vlsfr_stem_im2col(const float* x_nchw, void* out, int32_t N, int32_t H, int32_t W, int32_t stride, void* stream) {
  if (!x_nchw || !out || N <= 0 || H <= 0 || W <= 0 || stride < 1 || stride > 2)
    return fail(VLSFR_EINVAL, "vlsfr_stem_im2col: bad argument");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(blocks_for((int64_t)N * Ho * Wo, 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, x_nchw, (u16*)out, N, H, W, stride);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_stem_im2col");
  return VLSFR_OK;
}

int vlsfr_faces_normalize(const uint8_t* raw, const uint8_t* flip, float* out, int32_t N, int32_t H, int32_t W, int32_t C,
                          void* stream) {
  if (!raw || !out || N <= 0 || H <= 0 || W <= 0 || (C != 1 && C != 3))
    return fail(VLSFR_EINVAL, "vlsfr_faces_normalize: bad argument (C must be 1 or 3)");
  hipLaunchKernelGGL(faces_normalize_kernel, dim3(blocks_for((int64_t)N * 3 * H * ((W + 3) / 4), 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, raw, flip, out, N, H, W, C);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_faces_normalize");
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
