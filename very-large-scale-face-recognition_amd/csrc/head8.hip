// head_sweep8: the Dynamic-Class-Pool sweep on an fp8 (OCP e4m3) shadow of queue[0], D = 512 — the fp8 slice of
// config C5 (fp8 e4m3 MFMA operands, fp32 accumulate; tolerances cos >= 0.99, loss 5e-2).
//
// Same math and same partial-result format as head_sweep16 (head16.hip; reference ffc.py:195-201 / 248-253 + the
// softmax of F.cross_entropy and its backward): S^T = W . P^T, softmax numerators against the row's fixed reference
// exponent, O += P~ . W.  Both products run on v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 per instruction at twice
// the cycles of the bf16 16x16x32 form: 2x the bf16 rate) and every MFMA operand is 2 KiB of LDS per 128-deep step
// instead of 1 KiB per 32-deep step: half the matrix-pipe time AND half the LDS bytes per FLOP, which is what bounds
// the bf16 sweep (scripts/probes/lds_rates.hip: one LDS fragment per MFMA caps the pipes at 0.6 of their rate).
//
//   * The shadow is stored FRAGMENT-MAJOR, 128 KiB per tile of 128 pool columns (vlsfr_pool_shadow8_build):
//       R part (64 KiB): operand A of the first product.  [mb 8][ks 4][half 2][lane 64][16 B]; lane 16 g + m, bytes
//                        16 half + b  <->  pool column 16 mb + m, feature 128 ks + 32 g + 16 half + b.
//       T part (64 KiB): operand B of the second product (the tile transposed).  [nb 32][half 2][lane 64][16 B]; lane
//                        16 g + n, byte j = 16 half + b  <->  feature 16 nb + n, pool column 16 (j >> 2) + 4 g + (j & 3)
//                        — the order in which a lane of the first product's output holds its 32 columns, so the
//                        softmax numerators become operand A of the second product without leaving their lane.
//     Every fragment read is a linear, bank-conflict-free ds_read_b128 and every LDS-DMA a linear 1-KiB copy; no
//     transposed LDS reads (the bf16 kernel needs ds_read_b64_tr_b16 because it reads ONE row-major image both ways).
//     HBM bytes per column: 1024 (both layouts) = the bf16 shadow's — at batch > 64 the sweep is not HBM-bound.
//   * The tile streams through a 4-slot ring of 32-KiB pieces (RA, RB, TA, TB = the two halves of each part), three
//     pieces in flight behind a counted vmcnt, one raw s_barrier per piece (16 MFMAs per wave).
//   * Operand scaling.  Pool rows and probe rows are unit vectors: both are stored as e4m3(64 x) and the E8M0 block
//     scales of the instruction (2^-6 each) undo it, so the first product returns cosines.  (Probed on this part,
//     scripts/probes/mfma_f8_scale2.hip: the scale byte of lane (row, gs) multiplies k-block gs = bytes 0..15 of lane
//     groups 2 (gs & 1), 2 (gs & 1) + 1 for gs < 2 and bytes 16..31 for gs >= 2 — a block straddles two lanes.)  The
//     numerators 2^(logit - m_ref) span far more than e4m3's 17 binades over a row, so they are scaled per probe row
//     and HALF tile (= two k-blocks): the row's maximum over the 64 columns (two lane-exchange steps) picks an exponent
//     E, the lanes quantise p 2^(134 - E) (< 256), and the two k-blocks carry the scale byte E - 7.  Terms below 2^-16
//     of the largest of their 64 flush to zero (<= 0.1 % of that term in total); the row sum L of the loss stays fp32.
#include "hip_common.h"
#include "head_sweep16.h"

#include <utility>

using namespace vlsfr;

namespace {

constexpr int T8 = SW8_TQ;            // pool columns per tile
constexpr int DP = SW16_D;
constexpr int KTOP = SW16_KTOP;
constexpr int PIECE = 32768;          // bytes per ring slot
constexpr int TILE_BYTES = 4 * PIECE;
constexpr int NSL = 4;
constexpr int NW = 8;
constexpr int NT = 64 * NW;
constexpr int PF = 2;                 // fragment pairs in flight
constexpr float NEG_BIG = -1.0e30f;
constexpr float WSCALE = 64.f;        // stored = e4m3(64 x); E8M0 byte 121 = 2^-6 undoes it
constexpr int SC_UNIT = 121;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

template <int OFF>
__device__ __forceinline__ i32x4 lds_r128(uint32_t addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint32_t lds_r32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ i32x8 join8(i32x4 lo, i32x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

__device__ __forceinline__ int pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return w;
}

__device__ __forceinline__ f32x4 mfma8(i32x8 a, i32x8 b, f32x4 c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}

template <bool TOPK, bool SV>
__global__ __launch_bounds__(NT, 2) void head_sweep8_kernel(Sweep16Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* bits = (uint32_t*)(smem + NSL * PIECE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15;
  const int h = lane >> 4;
  // XCD-aware block order, as head_sweep16_kernel
  const int nrb = a.n_rowblk;
  const int within = blockIdx.x % (8 * nrb);
  const int chunk = (blockIdx.x / (8 * nrb)) * 8 + (within & 7);
  const int rowblk = within >> 3;
  const int64_t c0 = (int64_t)chunk * a.chunk_cols;
  const int64_t c1 = (c0 + a.chunk_cols < a.Q) ? c0 + a.chunk_cols : a.Q;
  const int ncols = c1 > c0 ? (int)(c1 - c0) : 0;
  const int ntiles = (ncols + T8 - 1) / T8;
  const int row_base = rowblk * (16 * NW) + wave * 16;
  const bool wave_active = row_base < a.B;     // wave-uniform

  // ---- special-column bitmap of this chunk: one bit per column, natural column order
  const int nwords = a.chunk_cols / 32;
  for (int i = tid; i < nwords; i += NT) bits[i] = 0u;
  __syncthreads();
  for (int i = tid; i < a.n_special; i += NT) {
    const int64_t c = (int64_t)a.special_col[i] - a.slot_lo;
    if (c >= c0 && c < c1) atomicOr(&bits[(c - c0) >> 5], 1u << ((c - c0) & 31));
  }
  __syncthreads();

  // ---- P fragments (operand B of S^T = W . P^T): lane (probe row r16, group h) holds features 128 ks + 32 h + j
  i32x8 pq[4];
  const int prow = row_base + r16;
  const bool ok_row = wave_active && prow < a.B;
  float nrm = 0.f;
  {
    const f32x4* src = (const f32x4*)(a.p + (size_t)(ok_row ? prow : 0) * DP + 32 * h);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f32x4 v = src[ks * 32 + q];
        if (!ok_row) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        nrm += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        pq[ks][q] = pack_fp8x4(v[0] * WSCALE, v[1] * WSCALE, v[2] * WSCALE, v[3] * WSCALE);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  nrm = lane_step_sum<16>(nrm);
  nrm = lane_step_sum<32>(nrm);
  // fixed reference exponent of the row (head16.hip): every logit lies in [-b, b], b = qscale |p| (SV: 1.4 b)
  const float m_ref = a.qscale * __builtin_sqrtf(nrm) * (SV ? (a.sv_t + a.sv_t - 1.f) : 1.f) - 60.f;
  const float sv_thr = (SV && ok_row) ? a.sv_thr[prow] : 0.f;
  const bool is_out = TOPK && ok_row && a.pool_label[prow] < 0;

  f32x4 oacc[32];
#pragma unroll
  for (int nb = 0; nb < 32; ++nb) oacc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float l_part = 0.f, tk_thr = NEG_BIG;
  const size_t lb = (((size_t)chunk * a.Bp + row_base + r16) * 4 + h) * KTOP;   // this lane's candidate list
  if (TOPK && wave_active) {
#pragma unroll
    for (int k = 0; k < KTOP; ++k) {
      a.topk_val[lb + k] = NEG_BIG;
      a.topk_idx[lb + k] = -1;
    }
  }

  // ---- LDS-DMA: piece p of the chunk = bytes [p, p + 1) * 32 KiB of the chunk's tiles; wave w copies 4 KiB of it
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.w8 + (size_t)(ncols > 0 ? c0 / T8 : 0) * TILE_BYTES), 0, ntiles * TILE_BYTES, 0x00020000);
  const int voff = lane * 16;
  auto issue = [&](int p) {   // pieces past the chunk's last tile are out of range: zeros
    char* dst = smem + (p & 3) * PIECE + wave * 4096;
    const int soff = p * PIECE + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(dst + i * 1024), 16, voff, soff + i * 1024, 0, 0);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem + (uint32_t)(lane * 16);
  const uint32_t bits0 = (uint32_t)(uintptr_t)(lds_void_t*)smem + NSL * PIECE;

  // entry of a stage: this wave's share of piece p has landed, then everybody's; the slot of piece p - 1 is free
  auto stage_begin = [&](int p) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSL - 2) * 4) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(p + NSL - 1);
  };

  issue(0);
  issue(1);
  issue(2);

  int sc_w = SC_UNIT;
  asm volatile("" : "+v"(sc_w));   // one VGPR for the constant scale operand

  for (int t = 0; t < ntiles; ++t) {
    i32x8 pa;
    int sc_lo = 0, sc_hi = 0;
    const int64_t ct = c0 + (int64_t)t * T8;
    // ================= first product + softmax numerators, two stages of 4 column blocks (64 columns):
    // S^T[16 mb + 4 h + e][probe row r16] -> bytes 4 mb + e of operand A of the second product.  Bytes 0..15 of the
    // lanes (this stage's columns when k = 0) are k-blocks 0 and 1 of that instruction, bytes 16..31 blocks 2 and 3,
    // and the scale of block gs comes from lane group gs: one scale per probe row and HALF tile, supplied by lane groups
    // 0, 1 (first half) and 2, 3 (second half).
    static_for<2>([&](auto KS_) {
      constexpr int k = decltype(KS_)::value;
      stage_begin(4 * t + k);
      if (wave_active) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(pq[ks]));
        const uint32_t sb = lds0 + (uint32_t)(((4 * t + k) & 3) * PIECE);
        const uint32_t wb0 = lds_r32(bits0 + 16u * (uint32_t)t + 8u * k), wb1 = lds_r32(bits0 + 16u * (uint32_t)t + 8u * k + 4u);
        f32x4 sacc[4];
        i32x4 wl[PF + 1], wh[PF + 1];
        static_for<PF>([&](auto I) {
          constexpr int i = decltype(I)::value;
          wl[i] = lds_r128<i * 2048>(sb);
          wh[i] = lds_r128<i * 2048 + 1024>(sb);
        });
        static_for<16>([&](auto I) {
          constexpr int i = decltype(I)::value;      // fragment (mb = 4 k + (i >> 2), ks = i & 3)
          if constexpr (i + PF < 16) {
            wl[(i + PF) % (PF + 1)] = lds_r128<(i + PF) * 2048>(sb);
            wh[(i + PF) % (PF + 1)] = lds_r128<(i + PF) * 2048 + 1024>(sb);
          }
          constexpr int later = (15 - i) < PF ? (15 - i) : PF;
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * later) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          constexpr int ml = i >> 2;
          if constexpr ((i & 3) == 0) sacc[ml] = (f32x4){0.f, 0.f, 0.f, 0.f};
          sacc[ml] = mfma8(join8(wl[i % (PF + 1)], wh[i % (PF + 1)]), pq[i & 3], sacc[ml], sc_w, sc_w);
          __builtin_amdgcn_sched_barrier(0);
        });
        // ---- numerators of these 64 columns
        const bool plain = ((wb0 | wb1) == 0u) && (ct + 64 * (k + 1) <= c1);   // wave-uniform
        // masked path without per-element state: the words shifted to this lane group's bits, the columns left in the chunk
        const uint32_t wq0 = wb0 >> (4 * h), wq1 = wb1 >> (4 * h);
        const int64_t left64 = c1 - ct - 64 * k - 4 * h;
        const int left = left64 > 4096 ? 4096 : (int)left64;
        if (TOPK) {
          // hard-negative candidates of an outlier row: the lane's best cosine of these 64 columns against its admission
          // threshold; the insertion (global-memory list) only when something qualifies (rare after the first tiles)
          if (is_out) {
            float best = NEG_BIG;
#pragma unroll
            for (int ml = 0; ml < 4; ++ml)
#pragma unroll
              for (int e = 0; e < 4; ++e) best = fmaxf(best, sacc[ml][e]);
            if (best > tk_thr) {
#pragma unroll
              for (int ml = 0; ml < 4; ++ml) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float c = sacc[ml][e];
                  const bool ok = plain || ((16 * ml + e < left) && !((((ml >> 1) ? wq1 : wq0) >> (16 * (ml & 1) + e)) & 1u));
                  if (ok && c > tk_thr) {
                    float cv = c;
                    int ci = (int)(ct + 64 * k + 16 * ml + 4 * h + e);
#pragma unroll
                    for (int kk = 0; kk < KTOP; ++kk) {
                      const float tv = a.topk_val[lb + kk];
                      const int ti = a.topk_idx[lb + kk];
                      const bool gt = cv > tv;
                      a.topk_val[lb + kk] = gt ? cv : tv;
                      a.topk_idx[lb + kk] = gt ? ci : ti;
                      cv = gt ? tv : cv;
                      ci = gt ? ti : ci;
                    }
                    tk_thr = a.topk_val[lb + KTOP - 1];
                  }
                }
              }
            }
          }
        }
        float mx = 0.f;
#pragma unroll
        for (int ml = 0; ml < 4; ++ml) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float c = sacc[ml][e];
            bool ok = true;
            if (!plain) ok = (16 * ml + e < left) && !((((ml >> 1) ? wq1 : wq0) >> (16 * (ml & 1) + e)) & 1u);
            float fac = 1.f;
            if (SV) {
              if (c > sv_thr) {                                   // ffc.py:122-125
                c = a.sv_t * c + a.sv_t - 1.f;
                fac = a.sv_t;
              }
            }
            float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(c, a.qscale, -m_ref));
            if (!plain) pe = ok ? pe : 0.f;
            l_part += pe;
            if (SV) pe *= fac;
            sacc[ml][e] = pe;
            mx = fmaxf(mx, pe);
          }
        }
        // exponent of the row's largest numerator over the 64 columns -> scale of these two k-blocks
        mx = lane_step_max<16>(mx);
        mx = lane_step_max<32>(mx);
        int E = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
        E = E < 7 ? 7 : E;
        const float mult = __builtin_bit_cast(float, (261 - E) << 23);   // 2^(134 - E): the maximum lands in [128, 256)
        if constexpr (k == 0) sc_lo = E - 7;
        else sc_hi = E - 7;
#pragma unroll
        for (int ml = 0; ml < 4; ++ml)
          pa[4 * k + ml] = pack_fp8x4(sacc[ml][0] * mult, sacc[ml][1] * mult, sacc[ml][2] * mult, sacc[ml][3] * mult);
      }
    });
    const int sc_p = h < 2 ? sc_lo : sc_hi;
    // ================= second product, two stages of 16 feature blocks: O[probe row 4 h + e][feature 16 nb + r16]
    static_for<2>([&](auto KS_) {
      constexpr int k = decltype(KS_)::value;
      stage_begin(4 * t + 2 + k);
      if (wave_active) {
        const uint32_t sb = lds0 + (uint32_t)(((4 * t + 2 + k) & 3) * PIECE);
        i32x4 wl[PF + 1], wh[PF + 1];
        static_for<PF>([&](auto I) {
          constexpr int i = decltype(I)::value;
          wl[i] = lds_r128<i * 2048>(sb);
          wh[i] = lds_r128<i * 2048 + 1024>(sb);
        });
        static_for<16>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if constexpr (i + PF < 16) {
            wl[(i + PF) % (PF + 1)] = lds_r128<(i + PF) * 2048>(sb);
            wh[(i + PF) % (PF + 1)] = lds_r128<(i + PF) * 2048 + 1024>(sb);
          }
          constexpr int later = (15 - i) < PF ? (15 - i) : PF;
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * later) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          constexpr int nb = 16 * k + i;
          oacc[nb] = mfma8(pa, join8(wl[i % (PF + 1)], wh[i % (PF + 1)]), oacc[nb], sc_p, sc_w);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued past the last tile

  // ---- write partials (format of head_sweep16_kernel)
  if (wave_active) {
    float l_row = l_part;
    l_row = lane_step_sum<16>(l_row);
    l_row = lane_step_sum<32>(l_row);
    const size_t pr = (size_t)chunk * a.Bp + prow;
    if (h == 0) {
      a.part_m[pr] = m_ref;
      a.part_l[pr] = l_row;
    }
#pragma unroll
    for (int nb = 0; nb < 32; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t orow = (size_t)chunk * a.Bp + row_base + 4 * h + e;
        a.part_o[orow * DP + nb * 16 + r16] = oacc[nb][e];
      }
  }
#endif
}

// ---- shadow construction: one workgroup per tile of 128 pool rows
constexpr int SROW = DP + 16;   // LDS bytes per pool row (16-byte pad)

__device__ __forceinline__ uint32_t quant4(f32x4 v) {
  return (uint32_t)pack_fp8x4(v[0] * WSCALE, v[1] * WSCALE, v[2] * WSCALE, v[3] * WSCALE);
}

__global__ __launch_bounds__(256) void pool_shadow8_kernel(const float* q0, uint8_t* out, int64_t Q) {
  __shared__ __attribute__((aligned(16))) uint8_t rows[T8 * SROW];
  const int64_t tile = blockIdx.x;
  const int64_t col0 = tile * T8;
  const int tid = threadIdx.x;
  for (int i = tid; i < T8 * (DP / 4); i += 256) {
    const int r = i / (DP / 4), c4 = i - r * (DP / 4);
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (col0 + r < Q) v = ((const f32x4*)(q0 + (size_t)(col0 + r) * DP))[c4];
    *(uint32_t*)(rows + r * SROW + c4 * 4) = quant4(v);
  }
  __syncthreads();
  uint8_t* dst = out + (size_t)tile * TILE_BYTES;
  // R part: chunk id = ((mb 4 + ks) 2 + half) 64 + 16 g + m
  for (int id = tid; id < 4096; id += 256) {
    const int lane = id & 63, half = (id >> 6) & 1, ks = (id >> 7) & 3, mb = id >> 9;
    const int g = lane >> 4, m = lane & 15;
    const uint4 v = *(const uint4*)(rows + (16 * mb + m) * SROW + 128 * ks + 32 * g + 16 * half);
    *(uint4*)(dst + (size_t)id * 16) = v;
  }
  // T part: chunk id = (nb 2 + half) 64 + 16 g + n; byte b <-> j = 16 half + b <-> column 16 (j >> 2) + 4 g + (j & 3)
  for (int id = tid; id < 4096; id += 256) {
    const int lane = id & 63, half = (id >> 6) & 1, nb = id >> 7;
    const int g = lane >> 4, n = lane & 15;
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t x = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = 16 * half + 4 * q + e;
        const int col = 16 * (j >> 2) + 4 * g + (j & 3);
        x |= (uint32_t)rows[col * SROW + 16 * nb + n] << (8 * e);
      }
      w[q] = x;
    }
    *(uint4*)(dst + 65536 + (size_t)id * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// the images of individual columns after queue[0][col] changed (one workgroup of 128 threads per column)
__global__ __launch_bounds__(128) void pool_shadow8_update_kernel(const float* q0, int64_t Q, const int32_t* cols, int32_t n,
                                                                   int32_t slot_lo, uint8_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t row[DP];
  const int64_t c = (int64_t)cols[blockIdx.x] - slot_lo;
  if (c < 0 || c >= Q) return;     // a slot of another rank's shard
  const int tid = threadIdx.x;
  *(uint32_t*)(row + tid * 4) = quant4(((const f32x4*)(q0 + (size_t)c * DP))[tid]);
  __syncthreads();
  uint8_t* dst = out + (size_t)(c / T8) * TILE_BYTES;
  const int cc = (int)(c % T8), mb = cc >> 4, m = cc & 15;
  if (tid < 32) {   // R part: (ks, g, half) = 32 chunks of 16 bytes
    const int half = tid & 1, g = (tid >> 1) & 3, ks = tid >> 3;
    const uint4 v = *(const uint4*)(row + 128 * ks + 32 * g + 16 * half);
    *(uint4*)(dst + (size_t)((((mb * 4 + ks) * 2 + half) * 64) + 16 * g + m) * 16) = v;
  }
  // T part: one byte per feature
  const int g = m >> 2, j = 4 * mb + (m & 3), half = j >> 4, b = j & 15;
  for (int f = tid; f < DP; f += 128) {
    const int nb = f >> 4, nn = f & 15;
    dst[65536 + (size_t)((((nb * 2 + half) * 64) + 16 * g + nn) * 16) + b] = row[f];
  }
}

}  // namespace

namespace vlsfr {

size_t sweep8_lds_bytes(int chunk_cols) { return (size_t)NSL * PIECE + (size_t)(chunk_cols / 32) * 4 + 16; }

int launch_sweep8(const Sweep16Args& a, bool topk, bool sv, hipStream_t st) {
  if (!a.w8) return fail(VLSFR_EINVAL, "head_sweep8: no fp8 shadow");
  if (a.chunk_cols % T8 != 0 || a.chunk_cols / T8 > SW8_MAX_TILES || a.n_chunks % 8 != 0 || a.Bp != a.n_rowblk * 16 * NW)
    return fail(VLSFR_EINVAL, "head_sweep8: inconsistent plan (chunk_cols %d, n_chunks %d, Bp %d)", a.chunk_cols, a.n_chunks, a.Bp);
  const size_t lds = sweep8_lds_bytes(a.chunk_cols);
  const dim3 grid(a.n_chunks * a.n_rowblk);
#define VLSFR_SWEEP8(T, S)                                                                                       \
  do {                                                                                                          \
    auto kern = head_sweep8_kernel<T, S>;                                                                       \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return hip_fail(e, "head_sweep8: hipFuncSetAttribute");                                \
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, a);                                                       \
  } while (0)
  if (topk && sv) VLSFR_SWEEP8(true, true);
  else if (topk) VLSFR_SWEEP8(true, false);
  else if (sv) VLSFR_SWEEP8(false, true);
  else VLSFR_SWEEP8(false, false);
#undef VLSFR_SWEEP8
  VLSFR_HIP_CHECK_LAUNCH("head_sweep8 launch");
  return VLSFR_OK;
}

}  // namespace vlsfr

extern "C" {

size_t vlsfr_pool_shadow8_bytes(int64_t Q) { return Q > 0 ? (size_t)((Q + T8 - 1) / T8) * TILE_BYTES : 0; }

int vlsfr_pool_shadow8_build(const float* queue0, void* shadow8, int64_t Q, int32_t D, void* stream) {
  if (!queue0 || !shadow8 || Q <= 0) return fail(VLSFR_EINVAL, "vlsfr_pool_shadow8_build: null buffer or empty pool");
  if (D != DP) return fail(VLSFR_EINVAL, "vlsfr_pool_shadow8_build: the fp8 sweep covers feat_dim 512 (got %d)", D);
  const int64_t tiles = (Q + T8 - 1) / T8;
  if (tiles > 0x7fffffffLL) return fail(VLSFR_EINVAL, "vlsfr_pool_shadow8_build: pool too large");
  hipLaunchKernelGGL(pool_shadow8_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, queue0, (uint8_t*)shadow8, Q);
  VLSFR_HIP_CHECK_LAUNCH("pool_shadow8 launch");
  return VLSFR_OK;
}

int vlsfr_pool_shadow8_update(const float* queue0, int64_t Q, int32_t D, const int32_t* cols, int32_t n, int32_t slot_lo,
                              void* shadow8, void* stream) {
  if (n <= 0) return VLSFR_OK;
  if (!queue0 || !shadow8 || !cols) return fail(VLSFR_EINVAL, "vlsfr_pool_shadow8_update: null buffer");
  if (D != DP) return fail(VLSFR_EINVAL, "vlsfr_pool_shadow8_update: feat_dim must be 512 (got %d)", D);
  hipLaunchKernelGGL(pool_shadow8_update_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, queue0, Q, cols, n, slot_lo,
                     (uint8_t*)shadow8);
  VLSFR_HIP_CHECK_LAUNCH("pool_shadow8_update launch");
  return VLSFR_OK;
}

}  // extern "C"
