// Fused Dynamic-Class-Pool head for gfx950 (C-ABI sections 3-4 of include/vlsfr.h).
//
// Replaces, for one FFC pass, reference ffc.py:182/195-203 (commit) and ffc.py:240-258 (rollback):
// the pool scatter, both pool contractions F.linear(p, queue[0]) / F.linear(p, weight), the
// mask blend, add_margin (ffc.py:60-138: AM / Arc / SV margins, cross-entropy, hard-negative
// top-k) and — because queue is a buffer and only dL/dp is needed — the whole backward of that
// sub-graph, in ONE sweep over queue[0].
//
// Design:
//   * "special columns" (slots written this pass, ones_idx, label slots: <= 3B of them) are masked
//     out of the sweep and handled exactly in fp32 by head_special/head_finish, per variant
//     (cos_theta1 vs cos_theta2 differ only there).  The rollback pass therefore never
//     mutates the pool at all; the commit pass scatters g afterwards.
//   * head_sweep is a flash-style pass over the remaining columns: S^T = W_tile · P^T on MFMA
//     (pool column on the MFMA row, so the accumulator already is the A operand of the second
//     product), online softmax in base 2 with deferred rescale, O += P~ · W_tile on MFMA with the
//     W tile read back transposed by ds_read_b64_tr_b16.  O/l is the softmax-weighted class
//     vector, i.e. dL/dp up to the target term — no second sweep for the backward.
//   * fp32 pool rows are converted to bf16 on the way into LDS (PRECISE: split into hi + lo bf16
//     and three MFMAs per product ≈ fp32 products, for the fp32-tolerance parity mode).
//   * per-lane top-k lists feed the hard-negative term of outlier rows (label == -1).
#include <cstring>

#include "hip_common.h"
#include "head_sweep16.h"

using namespace vlsfr;

namespace {

constexpr int TQ = 32;        // pool columns per tile
constexpr int ROWS_WG = 64;   // probe rows per workgroup (4 waves x 16)
constexpr int KTOP = 10;      // max hard_neg (ffc.py:48)
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float DEFER_THR = 12.0f;  // deferred-rescale threshold in log2 units

struct SweepArgs {
  const float* p;        // [B, D]
  const float* w0;       // queue[0], [Q, D]
  int64_t Q;
  int32_t B, D;
  int32_t chunk_cols;    // multiple of TQ
  int32_t n_chunks;
  const int32_t* special_col;
  int32_t n_special;
  const int32_t* pool_label;  // [B]; -1 rows take part in top-k
  float qscale;          // scale * log2(e)
  const float* sv_thr;   // SV: per-row hard-example threshold (cos units) or nullptr
  float sv_t;            // SV: mask_svfc (1.2)
  float* part_m;         // [n_chunks, Bp]
  float* part_l;         // [n_chunks, Bp]
  float* part_o;         // [n_chunks, Bp, DP]
  float* topk_val;       // [n_chunks, Bp, 4, KTOP]
  int32_t* topk_idx;
  int32_t Bp;
  int32_t n_rowblk;
  int32_t slot_lo;       // global slot id of local slot 0 (identity-sharded pool); 0 otherwise
};

template <int DP>
__device__ __forceinline__ int swz_off(int row, int chunk16) {
  // LDS image of a W tile: bf16 rows padded by 32 B.  With that stride both the ds_read_b128 row
  // reads of the first product (16 rows x 4 chunks per wave-instruction) and the
  // ds_read_b64_tr_b16 block reads of the second one (8 rows x 32 B per half-wave) are
  // bank-conflict free, and every per-k-step address is lane_base + immediate.
  return row * (DP * 2 + 32) + (chunk16 << 4);
}

template <int DP, bool PRECISE, bool TOPK, bool SV>
__global__ __launch_bounds__(256, 1) void head_sweep_kernel(SweepArgs a) {
  constexpr int TILE_B = TQ * (DP * 2 + 32);   // bytes per padded bf16 tile
  constexpr int KS = DP / 32;                  // k-steps of the first product
  constexpr int NB = DP / 16;                  // 16-column blocks of the second product
  constexpr int GROUPS = TQ * DP / 8;          // 8-float groups per tile
  constexpr int GPT = (GROUPS + 255) / 256;    // groups per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_hi = smem;
  char* tile_lo = smem + TILE_B;               // PRECISE only
  uint32_t* bits = (uint32_t*)(smem + (PRECISE ? 2 : 1) * TILE_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15;
  const int h = lane >> 4;
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs, so ids that are equal
  // mod 8 share an L2.  The row blocks of one column chunk get ids 8 apart inside one group of
  // 8 * n_rowblk consecutive ids: they run at the same time on the same XCD and the chunk is fetched
  // from HBM once instead of once per row block.
  const int nrb = a.n_rowblk;
  const int within = blockIdx.x % (8 * nrb);
  const int chunk = (blockIdx.x / (8 * nrb)) * 8 + (within & 7);
  const int rowblk = within >> 3;
  const int64_t c0 = (int64_t)chunk * a.chunk_cols;
  const int64_t c1 = (c0 + a.chunk_cols < a.Q) ? c0 + a.chunk_cols : a.Q;
  const int ntiles = (int)((c1 - c0 + TQ - 1) / TQ);
  const int row_base = rowblk * ROWS_WG + wave * 16;
  const bool wave_active = row_base < a.B;     // wave-uniform

  // ---- special-column bitmap of this chunk
  const int nwords = a.chunk_cols / 32;
  for (int i = tid; i < nwords; i += 256) bits[i] = 0u;
  __syncthreads();
  for (int i = tid; i < a.n_special; i += 256) {
    int64_t c = (int64_t)a.special_col[i] - a.slot_lo;   // special columns carry global slot ids
    if (c >= c0 && c < c1) atomicOr(&bits[(c - c0) >> 5], 1u << ((c - c0) & 31));
  }

  // ---- P fragments (B operand of S^T = W · P^T): lane holds P[row r16][32ks + 8h + j]
  bf16x8 pf_hi[KS];
  bf16x8 pf_lo[PRECISE ? KS : 1];
  {
    const int prow = row_base + r16;
    const bool ok = wave_active && prow < a.B;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = ks * 32 + h * 8 + j;
        float v = (ok && d < a.D) ? a.p[(size_t)prow * a.D + d] : 0.f;
        __bf16 hi = (__bf16)v;
        pf_hi[ks][j] = hi;
        if constexpr (PRECISE) pf_lo[ks][j] = (__bf16)(v - (float)hi);
      }
    }
  }
  float sv_thr = 0.f;
  bool is_out = false;
  {
    const int prow = row_base + r16;
    if (wave_active && prow < a.B) {
      if (SV) sv_thr = a.sv_thr[prow];
      if (TOPK) is_out = a.pool_label[prow] < 0;
    }
  }

  // ---- accumulators
  f32x4 oacc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) oacc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_ref = NEG_BIG;   // reference exponent of row r16 (log2 units), identical in the 4 h-lanes
  float l_part = 0.f;      // this lane's share of sum 2^(s - m_ref)
  float tk_v[KTOP];
  int tk_i[KTOP];
#pragma unroll
  for (int k = 0; k < KTOP; ++k) {
    tk_v[k] = NEG_BIG;
    tk_i[k] = -1;
  }

  // ---- tile staging registers: thread loads GPT groups of 8 consecutive floats
  f32x4 st[GPT][2];
  auto issue_loads = [&](int t) {
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
      const int e = tid + 256 * u;
      const int row = e / (DP / 8);
      const int c8 = e % (DP / 8);
      const int64_t cg = c0 + (int64_t)t * TQ + row;
      const bool ok = (e < GROUPS) && (cg < c1) && (c8 * 8 < a.D);
      if (ok) {
        const f32x4* src = (const f32x4*)(a.w0 + (size_t)cg * a.D + c8 * 8);
        st[u][0] = src[0];
        st[u][1] = src[1];
      } else {
        st[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        st[u][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
      const int e = tid + 256 * u;
      if (e < GROUPS) {
        const int row = e / (DP / 8);
        const int c8 = e % (DP / 8);
        const float v[8] = {st[u][0][0], st[u][0][1], st[u][0][2], st[u][0][3],
                            st[u][1][0], st[u][1][1], st[u][1][2], st[u][1][3]};
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          hi[j] = (__bf16)v[j];
          if (PRECISE) lo[j] = (__bf16)(v[j] - (float)hi[j]);
        }
        const int off = swz_off<DP>(row, c8);
        *(bf16x8*)(tile_hi + off) = hi;
        if (PRECISE) *(bf16x8*)(tile_lo + off) = lo;
      }
    }
  };

  if (ntiles > 0) issue_loads(0);
  for (int t = 0; t < ntiles; ++t) {
    write_tile();
    __syncthreads();
    if (t + 1 < ntiles) issue_loads(t + 1);
    if (wave_active) {
      // ================= first product: S^T[j][i] for j in 2 blocks of 16 pool columns
      f32x4 sacc[2];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int row = jb * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int off = swz_off<DP>(row, ks * 4 + h);
          bf16x8 wa = *(const bf16x8*)(tile_hi + off);
          acc = mfma16(wa, pf_hi[ks], acc);
          if constexpr (PRECISE) {
            bf16x8 wl = *(const bf16x8*)(tile_lo + off);
            acc = mfma16(wa, pf_lo[ks], acc);
            acc = mfma16(wl, pf_hi[ks], acc);
          }
          if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the ds_read run-ahead (VGPR pressure)
        }
        sacc[jb] = acc;
      }
      // lane (r16, h) now holds cos(p_row r16, pool column tile + 16jb + 4h + e), e = 0..3
      float s[8];
      float av[8];      // SV derivative factor
      bool valid[8];
      float tmax = NEG_BIG;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int jloc = t * TQ + jb * 16 + 4 * h;             // column offset inside the chunk
        const uint32_t word = bits[jloc >> 5];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = jb * 4 + e;
          const int64_t cg = c0 + jloc + e;
          const bool ok = (cg < c1) && !((word >> ((jloc + e) & 31)) & 1u);
          float c = sacc[jb][e];
          if (TOPK) {
            if (ok && is_out && c > tk_v[KTOP - 1]) {           // rare after warm-up
              float cv = c;
              int ci = (int)cg;
#pragma unroll
              for (int k = 0; k < KTOP; ++k) {
                const bool gt = cv > tk_v[k];
                const float tv = tk_v[k];
                const int ti = tk_i[k];
                tk_v[k] = gt ? cv : tv;
                tk_i[k] = gt ? ci : ti;
                cv = gt ? tv : cv;
                ci = gt ? ti : ci;
              }
            }
          }
          float fac = 1.f;
          if (SV) {
            if (c > sv_thr) {                                   // ffc.py:122-125
              c = a.sv_t * c + a.sv_t - 1.f;
              fac = a.sv_t;
            }
          }
          av[q] = fac;
          valid[q] = ok;
          s[q] = ok ? c * a.qscale : NEG_BIG;
          tmax = fmaxf(tmax, s[q]);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      // deferred rescale: move the reference exponent only when the row maximum outgrows it
      const bool grow = tmax > m_ref + DEFER_THR;
      if (__any(grow)) {
        const float m_new = grow ? tmax : m_ref;
        const float alpha = grow ? exp2f(m_ref - m_new) : 1.f;
        m_ref = m_new;
        l_part *= alpha;
        float al[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) al[e] = __shfl(alpha, 4 * h + e, 64);   // O rows are 4h + e
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) oacc[nb][e] *= al[e];
        }
      }
      bf16x8 pa_hi, pa_lo;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float pe = valid[q] ? exp2f(s[q] - m_ref) : 0.f;
        l_part += pe;
        const float pw = SV ? pe * av[q] : pe;
        const __bf16 hi = (__bf16)pw;
        pa_hi[q] = hi;
        if constexpr (PRECISE) pa_lo[q] = (__bf16)(pw - (float)hi);
      }
      // ================= second product: O[i][d] += sum_j P~[i][j] W[j][d]
      // k index of the MFMA: element q of lane group h  <->  tile row 16 (q>>2) + 4h + (q&3),
      // which is exactly how pa_* is laid out; B comes from two transposed 4x16 block reads.
      const int trow0 = 4 * h + (r16 >> 2);            // block row supplied by this lane (first block)
      const int tsub = (r16 & 3);                      // 4-column group inside the 16-column block
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int ch = nb * 2 + (tsub >> 1);
        const int o0 = swz_off<DP>(trow0, ch) + 8 * (tsub & 1);
        const int o1 = swz_off<DP>(trow0 + 16, ch) + 8 * (tsub & 1);
        short4v b0 = lds_read_tr16(tile_hi + o0);
        short4v b1 = lds_read_tr16(tile_hi + o1);
        short __attribute__((ext_vector_type(8))) bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        bf16x8 wb = __builtin_bit_cast(bf16x8, bs);
        oacc[nb] = mfma16(pa_hi, wb, oacc[nb]);
        if constexpr (PRECISE) {
          short4v l0 = lds_read_tr16(tile_lo + o0);
          short4v l1 = lds_read_tr16(tile_lo + o1);
          short __attribute__((ext_vector_type(8))) ls = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
          bf16x8 wl = __builtin_bit_cast(bf16x8, ls);
          oacc[nb] = mfma16(pa_hi, wl, oacc[nb]);
          oacc[nb] = mfma16(pa_lo, wb, oacc[nb]);
        }
        if ((nb & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- write partials
  if (wave_active) {
    float l_row = l_part;
    l_row += __shfl_xor(l_row, 16, 64);
    l_row += __shfl_xor(l_row, 32, 64);
    const int prow = row_base + r16;
    const size_t pr = (size_t)chunk * a.Bp + prow;
    if (h == 0) {
      a.part_m[pr] = m_ref;
      a.part_l[pr] = l_row;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t orow = (size_t)chunk * a.Bp + row_base + 4 * h + e;
        a.part_o[orow * DP + nb * 16 + r16] = oacc[nb][e];
      }
    }
    if (TOPK) {
      const size_t base = (pr * 4 + h) * KTOP;
#pragma unroll
      for (int k = 0; k < KTOP; ++k) {
        a.topk_val[base + k] = tk_v[k];
        a.topk_idx[base + k] = tk_i[k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// head_special: exact fp32 cosines of every probe row against the <= 3B special columns, for both
// variants (cos_theta1: queue[0] after this pass's writes; cos_theta2: mask-blended weight).
// ------------------------------------------------------------------------------------------------
struct SpecialArgs {
  const float* p;
  const float* g;
  const float* queue;   // [2, Q, D]
  int64_t Q;
  int32_t B, D, n_special;
  const int32_t* special_col;
  const int32_t* src1;
  const int32_t* src2;
  float* cos1;          // [B, n_special]
  float* cos2;
  int32_t slot_lo;      // identity-sharded pool: this rank owns global slots [slot_lo, slot_lo + Q)
};

__device__ __forceinline__ const float* special_vec(const float* g, const float* queue, int64_t Q, int D, int col,
                                                    int src) {
  if (src >= 0) return g + (size_t)src * D;
  if (src == -1) return queue + (size_t)col * D;
  return queue + ((size_t)Q + col) * D;
}

constexpr int SPECIAL_ROWS = 128;
__global__ __launch_bounds__(256) void head_special_kernel(SpecialArgs a) {
  const int s = blockIdx.x;
  const int col = a.special_col[s] - a.slot_lo;
  if (col < 0 || col >= a.Q) return;   // owned by another rank
  const int s1 = a.src1[s], s2 = a.src2[s];
  const float* v1 = special_vec(a.g, a.queue, a.Q, a.D, col, s1);
  const float* v2 = special_vec(a.g, a.queue, a.Q, a.D, col, s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blockIdx.y: chunk of SPECIAL_ROWS probe rows (an 8-rank batch has 2048 rows: one block per column would walk them
  // 512 dependent dot products deep)
  const int i_end = ((int)blockIdx.y + 1) * SPECIAL_ROWS < a.B ? ((int)blockIdx.y + 1) * SPECIAL_ROWS : a.B;
  for (int i = (int)blockIdx.y * SPECIAL_ROWS + wave; i < i_end; i += 4) {
    float d1 = 0.f, d2 = 0.f;
    for (int d = lane; d < a.D; d += 64) {
      const float pv = a.p[(size_t)i * a.D + d];
      d1 += pv * v1[d];
      d2 += pv * v2[d];
    }
    d1 = wave_sum(d1);
    d2 = (s1 == s2) ? d1 : wave_sum(d2);
    if (lane == 0) {
      a.cos1[(size_t)i * a.n_special + s] = d1;
      a.cos2[(size_t)i * a.n_special + s] = d2;
    }
  }
}

// Sharded SV: the threshold of a row is known on the rank that owns its label slot; the others report -3e38
// (outlier rows: +3e38 everywhere), so an all-reduce(max) over the ranks yields the global thresholds.
__global__ void head_sv_thr_shard_kernel(const int32_t* pool_label, const int32_t* special_col, int n_special, int B,
                                         const float* cos1, const float* cos2, float margin, int slot_lo, int Qs,
                                         float* thr1, float* thr2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int t = pool_label[i];
  float g1 = 3.0e38f, g2 = 3.0e38f;
  if (t >= 0) {
    g1 = g2 = -3.0e38f;
    if (t >= slot_lo && t < slot_lo + Qs) {
      for (int s = 0; s < n_special; ++s)
        if (special_col[s] == t) {
          g1 = cos1[(size_t)i * n_special + s] - margin;
          g2 = cos2[(size_t)i * n_special + s] - margin;
          break;
        }
    }
  }
  thr1[i] = g1;
  thr2[i] = g2;
}

// SV hard-example thresholds gt - margin per row and variant (ffc.py:121-122)
__global__ void head_sv_thr_kernel(const int32_t* pool_label, const int32_t* special_col, int n_special, int B,
                                   const float* cos1, const float* cos2, float margin, float* thr1, float* thr2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int t = pool_label[i];
  float g1 = 3.0e38f, g2 = 3.0e38f;   // outlier rows: no column is "hard"
  if (t >= 0) {
    for (int s = 0; s < n_special; ++s)
      if (special_col[s] == t) {
        g1 = cos1[(size_t)i * n_special + s] - margin;
        g2 = cos2[(size_t)i * n_special + s] - margin;
        break;
      }
  }
  thr1[i] = g1;
  thr2[i] = g2;
}

// ------------------------------------------------------------------------------------------------
// head_finish: one workgroup per probe row.  Merges the sweep partials with the special columns,
// applies the margin on the target column, and emits the row's loss terms and dL/dp.
// ------------------------------------------------------------------------------------------------
struct FinishArgs {
  const float* g;
  const float* queue;
  int64_t Q;
  int32_t B, D, DP, Bp, n_chunks, n_special;
  const int32_t* pool_label;
  const int32_t* special_col;
  const int32_t* src1;
  const int32_t* src2;
  const float* cos1;
  const float* cos2;
  const float* part_m[2];   // per variant (identical pointers unless SV)
  const float* part_l[2];
  const float* part_o[2];
  const float* topk_val;
  const int32_t* topk_idx;
  int32_t loss_type;        // 0 AM, 1 Arc, 2 SV
  float scale, margin, sv_t;
  int32_t hard_neg, n_pos, n_out;
  float* row_loss;          // [B, 2]
  float* dP;                // [B, D]
};

// sum_s wts[s] * vptr[s][:] over the special columns (weights and class-vector addresses in LDS): wave w takes
// columns w, w + 4, ..., every lane owns features lane + 64 k -- eight independent loads in flight per lane and a
// quarter of the iterations of a "thread per feature, all columns in sequence" loop, which pays one memory latency
// per column (0.77 us per special column and row measured).  The four per-wave sums meet in wsum[4][D] (LDS) and are
// added in a fixed order: deterministic.
__device__ __forceinline__ void combine_rows(const float* const* vptr, const float* wts, int n, int D, float* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float accv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) accv[k] = 0.f;
  for (int s = wave; s < n; s += 4) {
    const float w = wts[s];
    const float* vec = vptr[s];
    float ld[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ld[k] = (lane + 64 * k < D) ? vec[lane + 64 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) accv[k] += w * ld[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (lane + 64 * k < D) wsum[wave * D + lane + 64 * k] = accv[k];
  __syncthreads();
}

// the same over rows base + c * stride (the per-chunk partial O vectors of one probe row)
__device__ __forceinline__ void combine_strided(const float* base, size_t stride, const float* wts, int n, int D, float* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float accv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) accv[k] = 0.f;
  for (int c = wave; c < n; c += 4) {
    const float w = wts[c];
    const float* vec = base + (size_t)c * stride;
    float ld[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ld[k] = (lane + 64 * k < D) ? vec[lane + 64 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) accv[k] += w * ld[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (lane + 64 * k < D) wsum[wave * D + lane + 64 * k] = accv[k];
  __syncthreads();
}

__global__ __launch_bounds__(256) void head_finish_kernel(FinishArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int i = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int D = a.D;
  float* dp = (float*)smem;                     // [D] accumulated gradient of this row
  float* red = dp + ((D + 3) & ~3);             // [16] scratch
  float* wts = red + 16;                        // [max(n_chunks, n_special)] per-chunk / per-special weights
  const int nw_lds = a.n_chunks > a.n_special ? a.n_chunks : a.n_special;
  const float** vptr = (const float**)(wts + ((nw_lds + 1) & ~1));   // [n_special] class-vector address per special column
  float* wsum = (float*)(vptr + a.n_special);                       // [4][D] per-wave partial sums
  const int label = a.pool_label[i];
  const float qs = a.scale * LOG2E;
  for (int d = tid; d < D; d += 256) dp[d] = 0.f;
  __shared__ int sh_idx[4];
  __syncthreads();

  auto block_max = [&](float v) -> float {
    v = wave_max(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
  };
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
  };

  if (label >= 0) {
    // ---------------------------------------------------------------- positive row: margin + CE
    if (tid == 0) sh_idx[0] = -1;
    __syncthreads();
    for (int s = tid; s < a.n_special; s += 256)
      if (a.special_col[s] == label) sh_idx[0] = s;
    __syncthreads();
    const int st = sh_idx[0];   // the label slot is always a special column (vlsfr_dcp_assign)
    __syncthreads();
    const float inv_pos = 1.f / (float)a.n_pos;
    for (int v = 0; v < 2; ++v) {
      const float* cs = (v == 0 ? a.cos1 : a.cos2) + (size_t)i * a.n_special;
      const int32_t* src = (v == 0 ? a.src1 : a.src2);
      const float gt = st >= 0 ? cs[st] : 0.f;
      float tm, dtm;   // modified target cosine and d(tm)/d(gt)
      if (a.loss_type == 0) {
        tm = gt - a.margin;
        dtm = 1.f;
      } else if (a.loss_type == 1) {
        const float sn = sqrtf(1.f - gt * gt);            // no clamp (ffc.py:101)
        tm = gt * cosf(a.margin) - sn * sinf(a.margin);
        dtm = cosf(a.margin) + gt / sn * sinf(a.margin);
      } else {
        tm = (gt > a.margin) ? gt - a.margin : gt;         // ffc.py:123
        dtm = 1.f;
      }
      const float thr = gt - a.margin;                     // SV hard-example threshold
      // --- global exponent
      float mx = NEG_BIG;
      for (int c = tid; c < a.n_chunks; c += 256) mx = fmaxf(mx, a.part_m[v][(size_t)c * a.Bp + i]);
      for (int s = tid; s < a.n_special; s += 256) {
        float c = cs[s];
        if (s == st) c = tm;
        else if (a.loss_type == 2 && c > thr) c = a.sv_t * c + a.sv_t - 1.f;
        mx = fmaxf(mx, c * qs);
      }
      const float M = block_max(mx);
      // --- denominator and per-chunk weights
      float lsum = 0.f;
      for (int c = tid; c < a.n_chunks; c += 256) {
        const float w = exp2f(a.part_m[v][(size_t)c * a.Bp + i] - M);
        wts[c] = w;
        lsum += w * a.part_l[v][(size_t)c * a.Bp + i];
      }
      __syncthreads();
      float zt = 0.f;
      float sp_l = 0.f;
      for (int s = tid; s < a.n_special; s += 256) {
        float c = cs[s];
        if (s == st) c = tm;
        else if (a.loss_type == 2 && c > thr) c = a.sv_t * c + a.sv_t - 1.f;
        sp_l += exp2f(c * qs - M);
      }
      const float L = block_sum(lsum + sp_l);
      zt = tm * a.scale;
      if (tid == 0) a.row_loss[(size_t)i * 2 + v] = (LN2 * (M + log2f(L)) - zt) * inv_pos;
      // --- gradient: swept part
      const float gscale = a.scale * inv_pos / L;
      combine_strided(a.part_o[v] + (size_t)i * a.DP, (size_t)a.Bp * a.DP, wts, a.n_chunks, D, wsum);
      for (int d = tid; d < D; d += 256) dp[d] += gscale * ((wsum[d] + wsum[D + d]) + (wsum[2 * D + d] + wsum[3 * D + d]));
      __syncthreads();
      // --- gradient: special columns (softmax weight times d logit / d cos) and the -1 of the target
      for (int s = tid; s < a.n_special; s += 256) {
        float c = cs[s];
        float fac = 1.f;
        if (s == st) {
          c = tm;
          fac = dtm;
        } else if (a.loss_type == 2 && c > thr) {
          c = a.sv_t * c + a.sv_t - 1.f;
          fac = a.sv_t;
        }
        float w = exp2f(c * qs - M) * gscale * fac;
        if (s == st) w -= a.scale * inv_pos * dtm;
        wts[s] = w;
        vptr[s] = special_vec(a.g, a.queue, a.Q, D, a.special_col[s], src[s]);
      }
      __syncthreads();
      // thread per feature, loop over the special columns: weights and addresses come from LDS, so the
      // global loads of consecutive columns are independent and stay in flight together
      combine_rows(vptr, wts, a.n_special, D, wsum);
      for (int d = tid; d < D; d += 256) dp[d] += (wsum[d] + wsum[D + d]) + (wsum[2 * D + d] + wsum[3 * D + d]);
      __syncthreads();
    }
  } else {
    // ---------------------------------------------------------------- outlier row: hard negatives
    const float inv = 1.f / ((float)a.n_out * (float)a.hard_neg);
    const size_t ncand_sweep = (size_t)a.n_chunks * 4 * KTOP;
    for (int v = 0; v < 2; ++v) {
      const float* cs = (v == 0 ? a.cos1 : a.cos2) + (size_t)i * a.n_special;
      const int32_t* src = (v == 0 ? a.src1 : a.src2);
      float loss = 0.f;
      float last_v = 3.0e38f;
      long long last_k = -1;   // candidates are consumed in (value desc, key asc) order
      for (int k = 0; k < a.hard_neg; ++k) {
        float bv = NEG_BIG;
        long long bk = 0x7fffffffffffffffLL;
        auto consider = [&](float cv, long long key) {
          if (cv <= NEG_BIG) return;
          const bool after = (cv < last_v) || (cv == last_v && key > last_k);
          if (!after) return;
          if (cv > bv || (cv == bv && key < bk)) {
            bv = cv;
            bk = key;
          }
        };
        for (size_t c = tid; c < ncand_sweep; c += 256) {
          const size_t chunk = c / (4 * KTOP), rest = c % (4 * KTOP);
          const size_t off = ((chunk * a.Bp + i) * 4) * KTOP + rest;
          consider(a.topk_val[off], (long long)a.topk_idx[off]);
        }
        for (int s = tid; s < a.n_special; s += 256) consider(cs[s], (long long)a.Q + s);
        // block arg-max on (value, -key)
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const long long ok = __shfl_xor(bk, o, 64);
          if (ov > bv || (ov == bv && ok < bk)) {
            bv = ov;
            bk = ok;
          }
        }
        __shared__ float wv[4];
        __shared__ long long wk[4];
        if (lane == 0) {
          wv[wave] = bv;
          wk[wave] = bk;
        }
        __syncthreads();
        bv = wv[0];
        bk = wk[0];
        for (int w = 1; w < 4; ++w)
          if (wv[w] > bv || (wv[w] == bv && wk[w] < bk)) {
            bv = wv[w];
            bk = wk[w];
          }
        __syncthreads();
        if (bv <= NEG_BIG) break;       // fewer than hard_neg candidates (tiny pools)
        last_v = bv;
        last_k = bk;
        if (bv >= 0.f) {                 // clip(min=0) (ffc.py:89): negative cosines contribute nothing
          loss += bv;
          const float* vec = (bk >= a.Q) ? special_vec(a.g, a.queue, a.Q, D, a.special_col[bk - a.Q], src[bk - a.Q])
                                         : a.queue + (size_t)bk * D;
          for (int d = tid; d < D; d += 256) dp[d] += inv * vec[d];
        }
      }
      if (tid == 0) a.row_loss[(size_t)i * 2 + v] = loss * inv;
      __syncthreads();
    }
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) a.dP[(size_t)i * D + d] = dp[d];
}

// ------------------------------------------------------------------------------------------------
// Identity-sharded pool: every rank sweeps its own slots for ALL rows of the batch and
// emits per row and variant a partial softmax state relative to its local maximum:
//   M (log2 units), L = sum 2^(zz - M), O[D] = sum 2^(zz - M) * dlogit/dcos * w   (unnormalised),
//   T[D] = -scale/n_pos * dtm * w_target and zt = scale * tm if this rank owns the target slot,
//   and for outlier rows its local top-k (value, global slot).
// The ranks then all-reduce max(M), rescale, and sum (O, T, L, zt); see head.py ShardedDcpHead.
// ------------------------------------------------------------------------------------------------
struct ShardFinishArgs {
  FinishArgs f;
  int32_t slot_lo;
  float* out_M;        // [B, 2]
  float* out_L;        // [B, 2]
  float* out_zt;       // [B, 2]
  float* out_O;        // [B, 2, D]
  float* out_T;        // [B, 2, D]
  float* cand_val;     // [B, 2, KTOP]
  int32_t* cand_col;   // [B, 2, KTOP] global slot ids (-1 = none)
  const float* sv_thr; // SV: [2][Bp] GLOBAL hard-example thresholds gt - margin (max over ranks), else nullptr
  float* packed;       // or nullptr.  [B, 2, 2 D + 2] rows (O | T | L | zt): the layout the ranks sum (reduce-scatter), written
                       // here directly instead of out_O / out_T / out_L / out_zt
  int fixed_ref;       // 1: the shadow sweep ran (every chunk's part_m is the row's fixed reference exponent, a function of
                       // the probe row only and so identical on every rank): the state is emitted relative to it, not to
                       // this rank's maximum, and the ranks need no all-reduce(max) before they sum
};

__global__ __launch_bounds__(256) void head_finish_shard_kernel(ShardFinishArgs sa) {
  const FinishArgs& a = sa.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int i = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int D = a.D;
  float* red = (float*)smem;                    // [16]
  float* wts = red + 16;                        // [max(n_chunks, n_special)]
  const int nw_lds = a.n_chunks > a.n_special ? a.n_chunks : a.n_special;
  const float** vptr = (const float**)(wts + ((nw_lds + 1) & ~1));
  float* wsum = (float*)(vptr + a.n_special);   // [4][D] per-wave partial sums   // [n_special]
  int* own = (int*)(wsum + 4 * D);              // [n_special]: indices of the special columns this rank owns, ascending
  const int label = a.pool_label[i];
  const float qs = a.scale * LOG2E;
  const int lo = sa.slot_lo, hi = sa.slot_lo + (int)a.Q;
  __shared__ int sh_idx[1];
  auto owned = [&](int s) { const int c = a.special_col[s]; return c >= lo && c < hi; };
  // Stable compaction of the owned special columns (one wave, ballot prefix): with W ranks only ~1/W of the table is
  // owned here, and every loop below walks the compact list instead of the whole table (the combine of an 8-rank batch
  // read 8x the class vectors it needed, with weight zero).
  __shared__ int sh_nown[1];
  if (wave == 0) {
    int n = 0;
    for (int s0 = 0; s0 < a.n_special; s0 += 64) {
      const int s = s0 + lane;
      const bool o = s < a.n_special && owned(s);
      const unsigned long long m = __ballot(o);
      if (o) own[n + __popcll(m & ((1ull << lane) - 1ull))] = s;
      n += __popcll(m);
    }
    if (lane == 0) sh_nown[0] = n;
  }
  __syncthreads();
  const int n_own = sh_nown[0];
  auto block_max = [&](float v) -> float {
    v = wave_max(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
  };
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
  };
  if (label >= 0) {
    if (tid == 0) sh_idx[0] = -1;
    __syncthreads();
    for (int s = tid; s < a.n_special; s += 256)
      if (a.special_col[s] == label) sh_idx[0] = s;
    __syncthreads();
    const int st_all = sh_idx[0];
    const int st = (st_all >= 0 && label >= lo && label < hi) ? st_all : -1;   // target owned by this rank?
    __syncthreads();
    const float inv_pos = 1.f / (float)a.n_pos;
    for (int v = 0; v < 2; ++v) {
      const float* cs = (v == 0 ? a.cos1 : a.cos2) + (size_t)i * a.n_special;
      const int32_t* src = (v == 0 ? a.src1 : a.src2);
      float tm = 0.f, dtm = 0.f;
      // SV (ffc.py:118-127): hard columns (cos > gt - margin, threshold global over the ranks) become t*cos + t - 1
      const float thr = a.loss_type == 2 ? sa.sv_thr[(size_t)v * a.Bp + i] : 3.0e38f;
      auto cmod = [&](int s) -> float {
        if (s == st) return tm;
        const float c = cs[s];
        return (a.loss_type == 2 && c > thr) ? a.sv_t * c + a.sv_t - 1.f : c;
      };
      auto cfac = [&](int s) -> float {
        if (s == st) return dtm;
        return (a.loss_type == 2 && cs[s] > thr) ? a.sv_t : 1.f;
      };
      if (st >= 0) {
        const float gt = cs[st];
        if (a.loss_type == 0) {
          tm = gt - a.margin;
          dtm = 1.f;
        } else if (a.loss_type == 2) {
          tm = (gt > a.margin) ? gt - a.margin : gt;         // ffc.py:123
          dtm = 1.f;
        } else {
          const float sn = sqrtf(1.f - gt * gt);
          tm = gt * cosf(a.margin) - sn * sinf(a.margin);
          dtm = cosf(a.margin) + gt / sn * sinf(a.margin);
        }
      }
      float mx = NEG_BIG;
      for (int c = tid; c < a.n_chunks; c += 256) mx = fmaxf(mx, a.part_m[v][(size_t)c * a.Bp + i]);
      // fixed_ref: the special columns stay out of the maximum — their terms 2^(z - m_ref) are below 2^60 by the sweep's
      // own bound (head16.hip: m_ref = largest possible logit - 60)
      if (!sa.fixed_ref)
        for (int k = tid; k < n_own; k += 256) mx = fmaxf(mx, cmod(own[k]) * qs);
      const float M = block_max(mx);
      float lsum = 0.f;
      for (int c = tid; c < a.n_chunks; c += 256) {
        const float w = exp2f(a.part_m[v][(size_t)c * a.Bp + i] - M);
        wts[c] = w;
        lsum += w * a.part_l[v][(size_t)c * a.Bp + i];
      }
      for (int k = tid; k < n_own; k += 256) lsum += exp2f(cmod(own[k]) * qs - M);
      const float L = block_sum(lsum);
      float* O = sa.packed ? sa.packed + ((size_t)i * 2 + v) * (2 * D + 2) : sa.out_O + ((size_t)i * 2 + v) * D;
      float* T = sa.packed ? O + D : sa.out_T + ((size_t)i * 2 + v) * D;
      combine_strided(a.part_o[v] + (size_t)i * a.DP, (size_t)a.Bp * a.DP, wts, a.n_chunks, D, wsum);
      for (int d = tid; d < D; d += 256) {
        O[d] = (wsum[d] + wsum[D + d]) + (wsum[2 * D + d] + wsum[3 * D + d]);
        T[d] = 0.f;
      }
      __syncthreads();
      for (int k = tid; k < n_own; k += 256) {
        const int s = own[k];
        wts[k] = exp2f(cmod(s) * qs - M) * cfac(s);
        vptr[k] = special_vec(a.g, a.queue, a.Q, D, a.special_col[s] - lo, src[s]);
      }
      __syncthreads();
      combine_rows(vptr, wts, n_own, D, wsum);
      for (int d = tid; d < D; d += 256) O[d] += (wsum[d] + wsum[D + d]) + (wsum[2 * D + d] + wsum[3 * D + d]);
      __syncthreads();
      if (st >= 0) {
        const float* vec = special_vec(a.g, a.queue, a.Q, D, a.special_col[st] - lo, src[st]);
        const float k = -a.scale * inv_pos * dtm;
        for (int d = tid; d < D; d += 256) T[d] = k * vec[d];
      }
      if (tid == 0) {
        sa.out_M[i * 2 + v] = M;
        const float ztv = st >= 0 ? tm * a.scale : 0.f;
        if (sa.packed) {
          O[2 * D] = L;
          O[2 * D + 1] = ztv;
        } else {
          sa.out_L[i * 2 + v] = L;
          sa.out_zt[i * 2 + v] = ztv;
        }
      }
      __syncthreads();
    }
    for (int e = tid; e < 2 * KTOP; e += 256) {
      sa.cand_val[(size_t)i * 2 * KTOP + e] = NEG_BIG;
      sa.cand_col[(size_t)i * 2 * KTOP + e] = -1;
    }
  } else {
    // outlier row: local top-k candidates (raw cosines) per variant; no softmax state
    const size_t ncand_sweep = (size_t)a.n_chunks * 4 * KTOP;
    for (int v = 0; v < 2; ++v) {
      const float* cs = (v == 0 ? a.cos1 : a.cos2) + (size_t)i * a.n_special;
      float last_v = 3.0e38f;
      long long last_k = -1;
      for (int k = 0; k < KTOP; ++k) {
        float bv = NEG_BIG;
        long long bk = 0x7fffffffffffffffLL;
        auto consider = [&](float cv, long long key) {
          if (cv <= NEG_BIG) return;
          const bool after = (cv < last_v) || (cv == last_v && key > last_k);
          if (!after) return;
          if (cv > bv || (cv == bv && key < bk)) {
            bv = cv;
            bk = key;
          }
        };
        for (size_t c = tid; c < ncand_sweep; c += 256) {
          const size_t chunk = c / (4 * KTOP), rest = c % (4 * KTOP);
          const size_t off = ((chunk * a.Bp + i) * 4) * KTOP + rest;
          if (a.topk_idx[off] >= 0) consider(a.topk_val[off], (long long)a.topk_idx[off] + lo);
        }
        for (int k2 = tid; k2 < n_own; k2 += 256) consider(cs[own[k2]], (long long)a.special_col[own[k2]]);
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const long long ok = __shfl_xor(bk, o, 64);
          if (ov > bv || (ov == bv && ok < bk)) {
            bv = ov;
            bk = ok;
          }
        }
        __shared__ float wv[4];
        __shared__ long long wk[4];
        if (lane == 0) {
          wv[wave] = bv;
          wk[wave] = bk;
        }
        __syncthreads();
        bv = wv[0];
        bk = wk[0];
        for (int w = 1; w < 4; ++w)
          if (wv[w] > bv || (wv[w] == bv && wk[w] < bk)) {
            bv = wv[w];
            bk = wk[w];
          }
        __syncthreads();
        if (tid == 0) {
          sa.cand_val[((size_t)i * 2 + v) * KTOP + k] = bv;
          sa.cand_col[((size_t)i * 2 + v) * KTOP + k] = bv > NEG_BIG ? (int32_t)bk : -1;
        }
        if (bv > NEG_BIG) {
          last_v = bv;
          last_k = bk;
        } else {
          last_v = NEG_BIG;   // exhausted: the remaining entries stay empty
        }
      }
      if (sa.packed) {
        float* row = sa.packed + ((size_t)i * 2 + v) * (2 * D + 2);
        for (int d = tid; d < 2 * D + 2; d += 256) row[d] = 0.f;
        if (tid == 0) sa.out_M[i * 2 + v] = NEG_BIG;
      } else {
        for (int d = tid; d < D; d += 256) {
          sa.out_O[((size_t)i * 2 + v) * D + d] = 0.f;
          sa.out_T[((size_t)i * 2 + v) * D + d] = 0.f;
        }
        if (tid == 0) {
          sa.out_M[i * 2 + v] = NEG_BIG;
          sa.out_L[i * 2 + v] = 0.f;
          sa.out_zt[i * 2 + v] = 0.f;
        }
      }
    }
  }
}

// T[row, v, :] += weight * class vector of the globally selected hard negatives this rank owns
// sel_col: [B, 2, k] global slot ids (or -1), sel_w: [B, 2, k] weights (0 for clipped / empty entries)
__global__ __launch_bounds__(256) void head_outlier_accum_kernel(const float* g, const float* queue, int64_t Q, int D,
                                                                 int slot_lo, const int32_t* special_col,
                                                                 const int32_t* src1, const int32_t* src2,
                                                                 int n_special, const int32_t* sel_col,
                                                                 const float* sel_w, int k, float* T, int tstride) {
  const int row = blockIdx.x, v = blockIdx.y;
  for (int e = 0; e < k; ++e) {
    const int col = sel_col[((size_t)row * 2 + v) * k + e];
    const float w = sel_w[((size_t)row * 2 + v) * k + e];
    if (col < slot_lo || col >= slot_lo + (int)Q || w == 0.f) continue;
    int src = -1;   // swept columns read queue[0]; special columns follow their variant's source
    for (int s = 0; s < n_special; ++s)
      if (special_col[s] == col) {
        src = (v == 0 ? src1 : src2)[s];
        break;
      }
    const float* vec = special_vec(g, queue, Q, D, col - slot_lo, src);
    for (int d = threadIdx.x; d < D; d += 256) T[((size_t)row * 2 + v) * tstride + d] += w * vec[d];
  }
}

// Global top-k of the hard-negative candidates of every outlier row (ffc.py:86-90 over the identity-sharded pool): each
// rank contributed its local top-KTOP (value, global slot) per row and variant; one thread per (row, variant) picks the k
// best of the W * KTOP gathered candidates (value descending, slot ascending on ties: every rank selects the same set),
// clips at zero (ffc.py:89) and emits the weights 1 / (n_out k) of the kept ones and the row's loss term.
__global__ __launch_bounds__(256) void head_topk_merge_kernel(const float* cand_val, const int32_t* cand_col, int W, int B, int k,
                                                              const int32_t* pool_label, float inv, int32_t* sel_col,
                                                              float* sel_w, float* sel_loss) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2 * B) return;
  const int row = idx >> 1;
  const bool is_out = pool_label[row] < 0;
  float last_v = 3.0e38f;
  int last_c = -1;
  float loss = 0.f;
  for (int e = 0; e < k; ++e) {
    float bv = NEG_BIG;
    int bc = 0x7fffffff;
    if (is_out)
      for (int w = 0; w < W; ++w)
        for (int j = 0; j < KTOP; ++j) {
          const size_t o = ((size_t)w * B * 2 + idx) * KTOP + j;
          const float cv = cand_val[o];
          const int cc = cand_col[o];
          if (cv <= NEG_BIG || cc < 0) continue;
          if (!((cv < last_v) || (cv == last_v && cc > last_c))) continue;   // already taken
          if (cv > bv || (cv == bv && cc < bc)) {
            bv = cv;
            bc = cc;
          }
        }
    const bool have = bv > NEG_BIG;
    const bool keep = have && bv >= 0.f;                  // clip(min = 0): negative cosines contribute nothing
    sel_col[(size_t)idx * k + e] = have ? bc : -1;
    sel_w[(size_t)idx * k + e] = keep ? inv : 0.f;
    if (keep) loss += bv;
    if (have) {
      last_v = bv;
      last_c = bc;
    } else {
      last_v = NEG_BIG;
    }
  }
  sel_loss[idx] = loss * inv;
}

// Loss terms and dL/dp rows from the summed state of the identity-sharded head: packed [n, 2, 2 D + 2] (O | T | L | zt)
// after the ranks' reduce-scatter (or all-reduce), Mg [n, 2] the common reference exponents of those rows.
//   positive row:  loss_v = (ln 2 (Mg + log2 L) - zt) / n_pos,  dP += scale / n_pos * O / L + T
//   outlier row:   loss_v = sel_loss,                            dP += T
__global__ __launch_bounds__(256) void head_shard_finish_kernel(const float* packed, const float* Mg, const int32_t* pool_label,
                                                                const float* sel_loss, int D, float scale, float inv_pos,
                                                                float* row_loss, float* dP) {
  const int i = blockIdx.x, tid = threadIdx.x;
  const bool pos = pool_label[i] >= 0;
  const float* r0 = packed + ((size_t)i * 2) * (2 * D + 2);
  const float* r1 = r0 + (2 * D + 2);
  const float L0 = pos ? r0[2 * D] : 1.f, L1 = pos ? r1[2 * D] : 1.f;
  const float k0 = pos ? scale * inv_pos / L0 : 0.f, k1 = pos ? scale * inv_pos / L1 : 0.f;
  for (int d = tid; d < D; d += 256) dP[(size_t)i * D + d] = (k0 * r0[d] + r0[D + d]) + (k1 * r1[d] + r1[D + d]);
  if (tid < 2) {
    const float* r = tid ? r1 : r0;
    const float L = tid ? L1 : L0;
    row_loss[i * 2 + tid] = pos ? (0.6931471805599453f * (Mg[i * 2 + tid] + log2f(L)) - r[2 * D + 1]) * inv_pos
                                : (sel_loss ? sel_loss[i * 2 + tid] : 0.f);
  }
}

__global__ void head_loss_reduce_kernel(const float* row_loss, int n, float* out) {
  // fixed-order sum: the scalar loss is bitwise reproducible run to run
  __shared__ float sh[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += row_loss[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// queue[rows[i], cols[i]] = g[i]; duplicates resolved "highest batch index wins"
__global__ __launch_bounds__(256) void pool_scatter_kernel(float* queue, int64_t Q, int D, const float* g,
                                                           const int32_t* rows, const int32_t* cols, int n,
                                                           int slot_lo, __bf16* shadow) {
  const int i = blockIdx.x;
  const int r = rows[i], c = cols[i] - slot_lo;
  if (c < 0 || c >= Q) return;   // slot owned by another rank (identity-sharded pool)
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  for (int j = i + 1 + threadIdx.x; j < n; j += 256)
    if (rows[j] == r && cols[j] - slot_lo == c) dead = 1;
  __syncthreads();
  if (dead) return;
  float* dst = queue + ((size_t)r * Q + c) * D;
  const float* src = g + (size_t)i * D;
  for (int d = threadIdx.x; d < D; d += 256) dst[d] = src[d];
  if (shadow && r == 0) {   // the bf16 shadow mirrors queue[0] (same round-to-nearest-even cast as the in-kernel conversion)
    __bf16* sd = shadow + (size_t)c * D;
    for (int d = threadIdx.x; d < D; d += 256) sd[d] = (__bf16)src[d];
  }
}

// shadow[q][d] = bf16(queue[0][q][d]): 8 elements per thread, 32 B in / 16 B out
__global__ __launch_bounds__(256) void pool_shadow_kernel(const float* q0, __bf16* shadow, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const f32x4 a = ((const f32x4*)q0)[2 * i], b = ((const f32x4*)q0)[2 * i + 1];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (__bf16)a[j];
      o[4 + j] = (__bf16)b[j];
    }
    ((bf16x8*)shadow)[i] = o;
  }
}

int round_dp(int D) {
  int dp = 32;
  while (dp < D) dp <<= 1;
  return dp;
}

int g_head_variant = -1;   // "head_variant": bf16-shadow sweep geometry (head_sweep16.h); -1 = by batch size
int g_head_dma_spread = 0; // "head_dma_spread": Sweep16Args::dma_spread

struct Plan {
  bool fast;   // shadow sweep: bf16 (head16.hip, variants 0 / 1) or fp8 (head8.hip, variant 2)
  int variant;
  int DP, Bp, n_chunks, chunk_cols, n_rowblk;
  size_t off_m, off_l, off_o, off_tv, off_ti, off_cos1, off_cos2, off_thr, off_rowloss, total;
  int n_sets;
};

int make_plan(const vlsfr_head_cfg* c, Plan* pl) {
  if (!c || c->B <= 0 || c->D <= 0 || c->Q <= 0) return fail(VLSFR_EINVAL, "head: B, D, Q must be positive");
  if (c->D % 16 != 0 || c->D > 512) return fail(VLSFR_EINVAL, "head: feat_dim must be a multiple of 16 and <= 512 (got %d)", c->D);
  if (c->Q > 0x7fffffffLL) return fail(VLSFR_EINVAL, "head: a pool shard holds at most 2^31-1 slots");
  if (c->loss_type < 0 || c->loss_type > 2) return fail(VLSFR_EINVAL, "head: loss_type must be 0 (AM), 1 (Arc) or 2 (SV)");
  if (c->hard_neg < 1 || c->hard_neg > KTOP) return fail(VLSFR_EINVAL, "head: hard_neg must be in [1, 10]");
  pl->DP = round_dp(c->D);
  // The bf16-shadow sweep: D = 512, plain bf16 operands, and a logit range its fixed reference exponent covers
  // (head16.hip: scale <= 64; larger scales keep the online-maximum kernel)
  const bool range_ok = !c->precise && c->D == SW16_D && c->scale > 0.f && c->scale * LOG2E <= 93.f;
  const bool fp8 = c->pool_fp8 != nullptr && range_ok;       // fp8 sweep (head8.hip) when its shadow is given
  pl->fast = (c->pool_bf16 != nullptr && range_ok) || fp8;
  pl->variant = 0;
  if (pl->fast) {
    pl->variant = fp8 ? 2 : g_head_variant >= 0 ? g_head_variant : (c->B > 64 ? 1 : 0);
    const int rows_wg = sweep16_rows_per_wg(pl->variant);
    const int tq = fp8 ? SW8_TQ : SW16_TQ, max_tiles = fp8 ? SW8_MAX_TILES : SW16_MAX_TILES;
    pl->n_rowblk = (c->B + rows_wg - 1) / rows_wg;
    pl->Bp = pl->n_rowblk * rows_wg;
    const int64_t tiles = (c->Q + tq - 1) / tq;
    // one workgroup per CU (the accumulators and the P fragments own the register file): whole rounds of 256
    int nch = c->n_chunks > 0 ? c->n_chunks : (256 + pl->n_rowblk - 1) / pl->n_rowblk;
    if (nch < 8) nch = 8;
    nch = (nch + 7) & ~7;
    while ((tiles + nch - 1) / nch > max_tiles) nch *= 2;
    if (nch > tiles) nch = (int)((tiles + 7) & ~(int64_t)7);
    const int64_t per = (tiles + nch - 1) / nch;
    pl->chunk_cols = (int)per * tq;
    pl->n_chunks = (int)((c->Q + pl->chunk_cols - 1) / pl->chunk_cols);
    pl->n_chunks = (pl->n_chunks + 7) & ~7;
  } else {
  pl->n_rowblk = (c->B + ROWS_WG - 1) / ROWS_WG;
  pl->Bp = pl->n_rowblk * ROWS_WG;
  int64_t tiles = (c->Q + TQ - 1) / TQ;
  // default: ~512 workgroups in the sweep (measured at B = 64 and 256: fewer, larger chunks shrink the
  // partial-state traffic of the finish kernel; below 256 workgroups the sweep loses balance)
  int nch = c->n_chunks > 0 ? c->n_chunks : 512 / pl->n_rowblk;
  if (c->n_chunks <= 0) nch = nch > 256 ? 256 : (nch < 32 ? 32 : nch);
  if (nch > tiles) nch = (int)tiles;
  int64_t per = (tiles + nch - 1) / nch;
  if (per > 1024) {   // bound the special-column bitmap (4 KiB of LDS)
    per = 1024;
    nch = (int)((tiles + per - 1) / per);
  }
  pl->chunk_cols = (int)per * TQ;
  pl->n_chunks = (int)((c->Q + pl->chunk_cols - 1) / pl->chunk_cols);
  pl->n_chunks = (pl->n_chunks + 7) & ~7;   // multiple of 8 for the XCD-aware block order (extra chunks are empty)
  }
  pl->n_sets = (c->loss_type == 2) ? 2 : 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t rowsz = (size_t)pl->n_chunks * pl->Bp;
  pl->off_m = take(rowsz * 4 * pl->n_sets);
  pl->off_l = take(rowsz * 4 * pl->n_sets);
  pl->off_o = take(rowsz * pl->DP * 4 * pl->n_sets);
  pl->off_tv = take(rowsz * 4 * KTOP * 4);
  pl->off_ti = take(rowsz * 4 * KTOP * 4);
  const int Bt = c->n_rows_total > c->B ? c->n_rows_total : c->B;   // rows of the whole (multi-rank) batch
  pl->off_cos1 = take((size_t)c->B * 3 * Bt * 4);
  pl->off_cos2 = take((size_t)c->B * 3 * Bt * 4);
  pl->off_thr = take((size_t)pl->Bp * 4 * 2);
  pl->off_rowloss = take((size_t)c->B * 2 * 4);
  pl->total = off;
  return VLSFR_OK;
}

// fp8 sweep only: the hard-negative candidates of an outlier row were ranked by e4m3 cosines (~3e-3 off), which swaps
// members among near-ties of the top-k (ffc.py:86-90 takes it from exact cosines).  This pass keeps the RS_M best
// candidates of the row by their approximate value, recomputes their cosines in fp32 from the master rows, and leaves
// exactly those — with exact values — in the row's candidate lists (chunk 0's 4 x KTOP entries; every other entry is
// emptied), so head_finish / head_shard_finish select and score the top-k as the fp32 kernels would.
constexpr int RS_M = 32;
__global__ __launch_bounds__(256) void topk_rescore_kernel(const float* p, const float* w0, const int32_t* pool_label, float* topk_val,
                                                           int32_t* topk_idx, int n_chunks, int Bp, int D) {
  const int i = blockIdx.x;
  if (pool_label[i] >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int sel_i[RS_M];
  __shared__ float ex[RS_M];
  __shared__ float wv[4];
  __shared__ long long wk[4];
  const size_t ncand = (size_t)n_chunks * 4 * KTOP;
  float last_v = 3.0e38f;
  long long last_k = -1;
  int nsel = 0;
  for (int k = 0; k < RS_M; ++k) {
    float bv = NEG_BIG;
    long long bk = 0x7fffffffffffffffLL;
    for (size_t c = tid; c < ncand; c += 256) {
      const size_t chunk = c / (4 * KTOP), rest = c % (4 * KTOP);
      const size_t off = ((chunk * Bp + i) * 4) * KTOP + rest;
      const float cv = topk_val[off];
      const long long key = topk_idx[off];
      if (cv <= NEG_BIG || key < 0) continue;
      if (!((cv < last_v) || (cv == last_v && key > last_k))) continue;
      if (cv > bv || (cv == bv && key < bk)) {
        bv = cv;
        bk = key;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const long long ok = __shfl_xor(bk, o, 64);
      if (ov > bv || (ov == bv && ok < bk)) {
        bv = ov;
        bk = ok;
      }
    }
    if (lane == 0) {
      wv[wave] = bv;
      wk[wave] = bk;
    }
    __syncthreads();
    bv = wv[0];
    bk = wk[0];
    for (int w = 1; w < 4; ++w)
      if (wv[w] > bv || (wv[w] == bv && wk[w] < bk)) {
        bv = wv[w];
        bk = wk[w];
      }
    __syncthreads();
    if (bv <= NEG_BIG) break;
    last_v = bv;
    last_k = bk;
    if (tid == 0) sel_i[k] = (int)bk;
    ++nsel;
  }
  __syncthreads();
  for (int m = wave; m < nsel; m += 4) {
    const float* vec = w0 + (size_t)sel_i[m] * D;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += p[(size_t)i * D + d] * vec[d];
    acc = wave_sum(acc);
    if (lane == 0) ex[m] = acc;
  }
  for (size_t c = tid; c < ncand; c += 256) {
    const size_t chunk = c / (4 * KTOP), rest = c % (4 * KTOP);
    const size_t off = ((chunk * Bp + i) * 4) * KTOP + rest;
    topk_val[off] = NEG_BIG;
    topk_idx[off] = -1;
  }
  __syncthreads();
  if (tid < nsel) {
    const size_t off = ((size_t)i * 4) * KTOP + tid;   // chunk 0
    topk_val[off] = ex[tid];
    topk_idx[off] = sel_i[tid];
  }
}

template <int DP, bool PRECISE>
int launch_sweep(const SweepArgs& a, bool topk, bool sv, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)(PRECISE ? 2 : 1) * TQ * (DP * 2 + 32) + (size_t)a.chunk_cols / 8 + 16;
#define VLSFR_SWEEP(T, S)                                                                              \
  do {                                                                                                \
    auto kern = head_sweep_kernel<DP, PRECISE, T, S>;                                                 \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return hip_fail(e, "head_sweep: hipFuncSetAttribute");                       \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);                                            \
  } while (0)
  if (topk && sv) VLSFR_SWEEP(true, true);
  else if (topk) VLSFR_SWEEP(true, false);
  else if (sv) VLSFR_SWEEP(false, true);
  else VLSFR_SWEEP(false, false);
#undef VLSFR_SWEEP
  VLSFR_HIP_CHECK_LAUNCH("head_sweep launch");
  return VLSFR_OK;
}

template <bool PRECISE>
int dispatch_sweep(int DP, const SweepArgs& a, bool topk, bool sv, dim3 grid, hipStream_t st) {
  switch (DP) {
    case 32: return launch_sweep<32, PRECISE>(a, topk, sv, grid, st);
    case 64: return launch_sweep<64, PRECISE>(a, topk, sv, grid, st);
    case 128: return launch_sweep<128, PRECISE>(a, topk, sv, grid, st);
    case 256: return launch_sweep<256, PRECISE>(a, topk, sv, grid, st);
    case 512: return launch_sweep<512, PRECISE>(a, topk, sv, grid, st);
  }
  return fail(VLSFR_EINVAL, "head_sweep: unsupported padded feat_dim %d", DP);
}

// one sweep per set (SV: one per variant) through whichever kernel the plan selected
int run_sweeps(const vlsfr_head_cfg* cfg, const Plan& pl, SweepArgs a, char* ws, const float* thr, bool sv, bool want_topk,
               hipStream_t st) {
  const size_t rowsz = (size_t)pl.n_chunks * pl.Bp;
  const dim3 grid(pl.n_chunks * pl.n_rowblk);
  for (int set = 0; set < pl.n_sets; ++set) {
    a.part_m = (float*)(ws + pl.off_m) + set * rowsz;
    a.part_l = (float*)(ws + pl.off_l) + set * rowsz;
    a.part_o = (float*)(ws + pl.off_o) + set * rowsz * pl.DP;
    a.sv_thr = sv ? thr + (size_t)set * pl.Bp : nullptr;
    const bool topk = want_topk && set == 0;   // top-k uses raw cosines: variant independent
    int rc;
    // bench.py's roofline leg (family 2): both contractions of the sweep, 4 B Q D FLOPs (count only
    // contractions actually executed)
    ProfScope prof(st, 2, 4.0 * (double)a.B * (double)a.Q * (double)a.D);
    if (pl.fast) {
      Sweep16Args f;
      f.p = a.p;
      f.w16 = (const uint16_t*)cfg->pool_bf16;
      f.w8 = (const uint8_t*)cfg->pool_fp8;
      f.Q = a.Q;
      f.B = a.B;
      f.chunk_cols = a.chunk_cols;
      f.n_chunks = a.n_chunks;
      f.special_col = a.special_col;
      f.n_special = a.n_special;
      f.pool_label = a.pool_label;
      f.qscale = a.qscale;
      f.sv_thr = a.sv_thr;
      f.sv_t = a.sv_t;
      f.part_m = a.part_m;
      f.part_l = a.part_l;
      f.part_o = a.part_o;
      f.topk_val = a.topk_val;
      f.topk_idx = a.topk_idx;
      f.Bp = a.Bp;
      f.n_rowblk = a.n_rowblk;
      f.slot_lo = a.slot_lo;
      f.dma_spread = g_head_dma_spread;
      rc = launch_sweep16(f, pl.variant, topk, sv, st);
    } else {
      rc = cfg->precise ? dispatch_sweep<true>(pl.DP, a, topk, sv, grid, st) : dispatch_sweep<false>(pl.DP, a, topk, sv, grid, st);
    }
    if (rc != VLSFR_OK) return rc;
  }
  if (pl.variant == 2 && pl.fast && want_topk) {
    static_assert(RS_M <= 4 * KTOP, "the rescored candidates live in one chunk's lists");
    hipLaunchKernelGGL(topk_rescore_kernel, dim3(a.B), dim3(256), 0, st, a.p, a.w0, a.pool_label, a.topk_val, a.topk_idx, pl.n_chunks,
                       pl.Bp, a.D);
    VLSFR_HIP_CHECK_LAUNCH("topk_rescore launch");
  }
  return VLSFR_OK;
}

}  // namespace

namespace vlsfr {
int head_set_option(const char* name, int32_t value) {
  if (!strcmp(name, "head_dma_spread")) {
    g_head_dma_spread = value != 0;
    return 0;
  }
  if (!strcmp(name, "head_variant")) {
    if (value < -1 || value > 1) return fail(VLSFR_EINVAL, "head_variant must be -1 (auto), 0 or 1");
    g_head_variant = value;
    return VLSFR_OK;
  }
  return 1;   // not a head option
}
}  // namespace vlsfr

extern "C" {

size_t vlsfr_head_cfg_size(void) { return sizeof(vlsfr_head_cfg); }

int vlsfr_pool_shadow_build(const float* queue0, void* shadow_bf16, int64_t Q, int32_t D, void* stream) {
  if (!queue0 || !shadow_bf16 || Q <= 0 || D <= 0 || D % 8 != 0)
    return fail(VLSFR_EINVAL, "vlsfr_pool_shadow_build: bad argument (D must be a multiple of 8)");
  const size_t n8 = (size_t)Q * D / 8;
  const size_t blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(pool_shadow_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                     queue0, (__bf16*)shadow_bf16, n8);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_pool_shadow_build launch");
  return VLSFR_OK;
}

int vlsfr_pool_scatter(float* queue, int64_t Q, int32_t D, const float* g, const int32_t* rows, const int32_t* cols,
                       int32_t n, int32_t slot_lo, void* shadow_bf16, void* stream) {
  if (!queue || !g || !rows || !cols || Q <= 0 || D <= 0 || n < 0)
    return fail(VLSFR_EINVAL, "vlsfr_pool_scatter: bad argument");
  if (n == 0) return VLSFR_OK;
  hipLaunchKernelGGL(pool_scatter_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, queue, Q, D, g, rows, cols, n, slot_lo,
                     (__bf16*)shadow_bf16);
  VLSFR_HIP_CHECK_LAUNCH("vlsfr_pool_scatter launch");
  return VLSFR_OK;
}

size_t vlsfr_head_workspace_bytes(const vlsfr_head_cfg* cfg) {
  Plan pl;
  if (make_plan(cfg, &pl) != VLSFR_OK) return 0;
  return pl.total;
}

int vlsfr_head_fwd_bwd(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                       const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                       const int32_t* src2, int32_t n_special, int32_t n_pos, float* loss_out, float* dP,
                       void* workspace, size_t workspace_bytes, void* stream) {
  Plan pl;
  int rc = make_plan(cfg, &pl);
  if (rc != VLSFR_OK) return rc;
  if (!p || !g || !queue || !pool_label || !loss_out || !dP || !workspace)
    return fail(VLSFR_EINVAL, "vlsfr_head_fwd_bwd: null argument");
  const int Bt = cfg->n_rows_total > cfg->B ? cfg->n_rows_total : cfg->B;
  if (n_special < 0 || n_special > 3 * Bt || (n_special > 0 && (!special_col || !src1 || !src2)))
    return fail(VLSFR_EINVAL, "vlsfr_head_fwd_bwd: bad special-column table");
  if (n_pos < 0 || n_pos > Bt) return fail(VLSFR_EINVAL, "vlsfr_head_fwd_bwd: n_pos out of range");
  if (workspace_bytes < pl.total)
    return fail(VLSFR_EINVAL, "vlsfr_head_fwd_bwd: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int B = cfg->B, D = cfg->D;
  const int n_out = Bt - n_pos;   // counts are over the whole batch: the loss means are global
  float* cos1 = (float*)(ws + pl.off_cos1);
  float* cos2 = (float*)(ws + pl.off_cos2);
  float* thr = (float*)(ws + pl.off_thr);
  float* row_loss = (float*)(ws + pl.off_rowloss);
  const size_t rowsz = (size_t)pl.n_chunks * pl.Bp;

  if (n_special > 0) {
    SpecialArgs sa{p, g, queue, cfg->Q, B, D, n_special, special_col, src1, src2, cos1, cos2, cfg->slot_lo};
    hipLaunchKernelGGL(head_special_kernel, dim3(n_special, (B + SPECIAL_ROWS - 1) / SPECIAL_ROWS), dim3(256), 0, st, sa);
    VLSFR_HIP_CHECK_LAUNCH("head_special launch");
  }
  const bool sv = cfg->loss_type == 2;
  if (sv) {
    hipLaunchKernelGGL(head_sv_thr_kernel, dim3((B + 255) / 256), dim3(256), 0, st, pool_label, special_col,
                       n_special, B, cos1, cos2, cfg->margin, thr, thr + pl.Bp);
    VLSFR_HIP_CHECK_LAUNCH("head_sv_thr launch");
  }
  SweepArgs a;
  a.p = p;
  a.w0 = queue;
  a.Q = cfg->Q;
  a.B = B;
  a.D = D;
  a.chunk_cols = pl.chunk_cols;
  a.n_chunks = pl.n_chunks;
  a.special_col = special_col;
  a.n_special = n_special;
  a.pool_label = pool_label;
  a.qscale = cfg->scale * LOG2E;
  a.sv_t = 1.2f;   // ffc.py:47 mask_svfc
  a.topk_val = (float*)(ws + pl.off_tv);
  a.topk_idx = (int32_t*)(ws + pl.off_ti);
  a.Bp = pl.Bp;
  a.n_rowblk = pl.n_rowblk;
  a.slot_lo = cfg->slot_lo;
  rc = run_sweeps(cfg, pl, a, ws, thr, sv, n_out > 0, st);
  if (rc != VLSFR_OK) return rc;
  FinishArgs f;
  f.g = g;
  f.queue = queue;
  f.Q = cfg->Q;
  f.B = B;
  f.D = D;
  f.DP = pl.DP;
  f.Bp = pl.Bp;
  f.n_chunks = pl.n_chunks;
  f.n_special = n_special;
  f.pool_label = pool_label;
  f.special_col = special_col;
  f.src1 = src1;
  f.src2 = src2;
  f.cos1 = cos1;
  f.cos2 = cos2;
  for (int v = 0; v < 2; ++v) {
    const int set = (pl.n_sets == 2) ? v : 0;
    f.part_m[v] = (float*)(ws + pl.off_m) + set * rowsz;
    f.part_l[v] = (float*)(ws + pl.off_l) + set * rowsz;
    f.part_o[v] = (float*)(ws + pl.off_o) + set * rowsz * pl.DP;
  }
  f.topk_val = (float*)(ws + pl.off_tv);
  f.topk_idx = (int32_t*)(ws + pl.off_ti);
  f.loss_type = cfg->loss_type;
  f.scale = cfg->scale;
  f.margin = cfg->margin;
  f.sv_t = 1.2f;
  f.hard_neg = cfg->hard_neg;
  f.n_pos = n_pos;
  f.n_out = n_out;
  f.row_loss = row_loss;
  f.dP = dP;
  const int nw = pl.n_chunks > n_special ? pl.n_chunks : n_special;
  const size_t lds = ((size_t)((D + 3) & ~3) + 16 + ((nw + 1) & ~1)) * 4 + (size_t)n_special * 8 + (size_t)4 * D * 4 + 16;
  if (lds > 160 * 1024) return fail(VLSFR_EINVAL, "head_finish: too many special columns for one LDS image");
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)head_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "head_finish: hipFuncSetAttribute");
  }
  hipLaunchKernelGGL(head_finish_kernel, dim3(B), dim3(256), lds, st, f);
  VLSFR_HIP_CHECK_LAUNCH("head_finish launch");
  hipLaunchKernelGGL(head_loss_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, 2 * B, loss_out);
  VLSFR_HIP_CHECK_LAUNCH("head_loss_reduce launch");
  return VLSFR_OK;
}

static int shard_partial_impl(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                             const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                             const int32_t* src2, int32_t n_special, int32_t n_pos, float* out_M, float* out_L,
                             float* out_zt, float* out_O, float* out_T, float* cand_val, int32_t* cand_col,
                             void* workspace, size_t workspace_bytes, void* stream, const float* sv_thr,
                             float* packed = nullptr, int32_t* fixed_ref_out = nullptr) {
  Plan pl;
  int rc = make_plan(cfg, &pl);
  if (rc != VLSFR_OK) return rc;
  if (!p || !g || !queue || !pool_label || !out_M || !cand_val || !cand_col || !workspace ||
      (!packed && (!out_L || !out_zt || !out_O || !out_T)))
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_partial: null argument");
  if (fixed_ref_out) *fixed_ref_out = (packed && pl.fast) ? 1 : 0;
  const bool sv = cfg->loss_type == 2;
  if (sv && !sv_thr)
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_partial: SV needs the global thresholds (vlsfr_head_shard_sv_thr + all-reduce(max), then vlsfr_head_shard_partial_sv)");
  const int B = cfg->B, D = cfg->D;
  if (n_special < 0 || n_special > 3 * B || (n_special > 0 && (!special_col || !src1 || !src2)))
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_partial: bad special-column table");
  if (workspace_bytes < pl.total) return fail(VLSFR_EINVAL, "vlsfr_head_shard_partial: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* cos1 = (float*)(ws + pl.off_cos1);
  float* cos2 = (float*)(ws + pl.off_cos2);
  if (n_special > 0) {
    SpecialArgs sa{p, g, queue, cfg->Q, B, D, n_special, special_col, src1, src2, cos1, cos2, cfg->slot_lo};
    hipLaunchKernelGGL(head_special_kernel, dim3(n_special, (B + SPECIAL_ROWS - 1) / SPECIAL_ROWS), dim3(256), 0, st, sa);
    VLSFR_HIP_CHECK_LAUNCH("head_special launch");
  }
  SweepArgs a;
  a.p = p;
  a.w0 = queue;
  a.Q = cfg->Q;
  a.B = B;
  a.D = D;
  a.chunk_cols = pl.chunk_cols;
  a.n_chunks = pl.n_chunks;
  a.special_col = special_col;
  a.n_special = n_special;
  a.pool_label = pool_label;
  a.qscale = cfg->scale * LOG2E;
  a.sv_t = 1.2f;
  float* thr = (float*)(ws + pl.off_thr);
  const size_t rowsz = (size_t)pl.n_chunks * pl.Bp;
  if (sv) {   // caller layout [2][B] -> workspace layout [2][Bp]
    for (int v = 0; v < 2; ++v) {
      hipError_t e = hipMemcpyAsync(thr + (size_t)v * pl.Bp, sv_thr + (size_t)v * B, (size_t)B * sizeof(float),
                                    hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return hip_fail(e, "vlsfr_head_shard_partial_sv: threshold copy");
    }
  }
  a.topk_val = (float*)(ws + pl.off_tv);
  a.topk_idx = (int32_t*)(ws + pl.off_ti);
  a.Bp = pl.Bp;
  a.n_rowblk = pl.n_rowblk;
  a.slot_lo = cfg->slot_lo;
  rc = run_sweeps(cfg, pl, a, ws, thr, sv, n_pos < B, st);
  if (rc != VLSFR_OK) return rc;
  ShardFinishArgs sf;
  FinishArgs& f = sf.f;
  f.g = g;
  f.queue = queue;
  f.Q = cfg->Q;
  f.B = B;
  f.D = D;
  f.DP = pl.DP;
  f.Bp = pl.Bp;
  f.n_chunks = pl.n_chunks;
  f.n_special = n_special;
  f.pool_label = pool_label;
  f.special_col = special_col;
  f.src1 = src1;
  f.src2 = src2;
  f.cos1 = cos1;
  f.cos2 = cos2;
  for (int v = 0; v < 2; ++v) {
    const int set = sv ? v : 0;
    f.part_m[v] = (float*)(ws + pl.off_m) + set * rowsz;
    f.part_l[v] = (float*)(ws + pl.off_l) + set * rowsz;
    f.part_o[v] = (float*)(ws + pl.off_o) + set * rowsz * pl.DP;
  }
  f.topk_val = a.topk_val;
  f.topk_idx = a.topk_idx;
  f.loss_type = cfg->loss_type;
  f.scale = cfg->scale;
  f.margin = cfg->margin;
  f.sv_t = 1.2f;
  f.hard_neg = cfg->hard_neg;
  f.n_pos = n_pos;
  f.n_out = B - n_pos;
  f.row_loss = nullptr;
  f.dP = nullptr;
  sf.slot_lo = cfg->slot_lo;
  sf.out_M = out_M;
  sf.out_L = out_L;
  sf.out_zt = out_zt;
  sf.out_O = out_O;
  sf.out_T = out_T;
  sf.cand_val = cand_val;
  sf.cand_col = cand_col;
  sf.sv_thr = sv ? thr : nullptr;
  sf.packed = packed;
  sf.fixed_ref = (packed && pl.fast) ? 1 : 0;
  const int nw = pl.n_chunks > n_special ? pl.n_chunks : n_special;
  const size_t lds_f = (size_t)(16 + ((nw + 1) & ~1)) * 4 + (size_t)n_special * 8 + (size_t)4 * D * 4 + (size_t)n_special * 4 + 16;
  if (lds_f > 160 * 1024) return fail(VLSFR_EINVAL, "head_finish_shard: too many special columns for one LDS image");
  if (lds_f > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)head_finish_shard_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_f);
    if (e != hipSuccess) return hip_fail(e, "head_finish_shard: hipFuncSetAttribute");
  }
  hipLaunchKernelGGL(head_finish_shard_kernel, dim3(B), dim3(256), lds_f, st, sf);
  VLSFR_HIP_CHECK_LAUNCH("head_finish_shard launch");
  return VLSFR_OK;
}

int vlsfr_head_shard_partial(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                             const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                             const int32_t* src2, int32_t n_special, int32_t n_pos, float* out_M, float* out_L,
                             float* out_zt, float* out_O, float* out_T, float* cand_val, int32_t* cand_col,
                             void* workspace, size_t workspace_bytes, void* stream) {
  return shard_partial_impl(cfg, p, g, queue, pool_label, special_col, src1, src2, n_special, n_pos, out_M, out_L, out_zt,
                            out_O, out_T, cand_val, cand_col, workspace, workspace_bytes, stream, nullptr);
}

int vlsfr_head_shard_partial_sv(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                                const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                                const int32_t* src2, int32_t n_special, int32_t n_pos, const float* sv_thr, float* out_M,
                                float* out_L, float* out_zt, float* out_O, float* out_T, float* cand_val,
                                int32_t* cand_col, void* workspace, size_t workspace_bytes, void* stream) {
  return shard_partial_impl(cfg, p, g, queue, pool_label, special_col, src1, src2, n_special, n_pos, out_M, out_L, out_zt,
                            out_O, out_T, cand_val, cand_col, workspace, workspace_bytes, stream, sv_thr);
}

int vlsfr_head_shard_sv_thr(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                            const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                            const int32_t* src2, int32_t n_special, float* thr_out, void* workspace,
                            size_t workspace_bytes, void* stream) {
  Plan pl;
  int rc = make_plan(cfg, &pl);
  if (rc != VLSFR_OK) return rc;
  if (!p || !g || !queue || !pool_label || !thr_out || !workspace)
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_sv_thr: null argument");
  const int B = cfg->B;
  if (n_special < 0 || n_special > 3 * B || (n_special > 0 && (!special_col || !src1 || !src2)))
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_sv_thr: bad special-column table");
  if (workspace_bytes < pl.total) return fail(VLSFR_EINVAL, "vlsfr_head_shard_sv_thr: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* cos1 = (float*)(ws + pl.off_cos1);
  float* cos2 = (float*)(ws + pl.off_cos2);
  if (n_special > 0) {
    SpecialArgs sa{p, g, queue, cfg->Q, B, cfg->D, n_special, special_col, src1, src2, cos1, cos2, cfg->slot_lo};
    hipLaunchKernelGGL(head_special_kernel, dim3(n_special, (B + SPECIAL_ROWS - 1) / SPECIAL_ROWS), dim3(256), 0, st, sa);
    VLSFR_HIP_CHECK_LAUNCH("head_special launch");
  }
  hipLaunchKernelGGL(head_sv_thr_shard_kernel, dim3((B + 255) / 256), dim3(256), 0, st, pool_label, special_col, n_special, B,
                     cos1, cos2, cfg->margin, cfg->slot_lo, (int32_t)cfg->Q, thr_out, thr_out + B);
  VLSFR_HIP_CHECK_LAUNCH("head_sv_thr_shard launch");
  return VLSFR_OK;
}

int vlsfr_head_outlier_accum(const vlsfr_head_cfg* cfg, const float* g, const float* queue, const int32_t* special_col,
                             const int32_t* src1, const int32_t* src2, int32_t n_special, const int32_t* sel_col,
                             const float* sel_w, int32_t k, float* T, void* stream) {
  return vlsfr_head_outlier_accum_strided(cfg, g, queue, special_col, src1, src2, n_special, sel_col, sel_w, k, T,
                                          cfg ? cfg->D : 0, stream);
}

int vlsfr_head_outlier_accum_strided(const vlsfr_head_cfg* cfg, const float* g, const float* queue, const int32_t* special_col,
                                     const int32_t* src1, const int32_t* src2, int32_t n_special, const int32_t* sel_col,
                                     const float* sel_w, int32_t k, float* T, int32_t t_stride, void* stream) {
  if (!cfg || !g || !queue || !sel_col || !sel_w || !T || k < 1 || k > KTOP || t_stride < cfg->D)
    return fail(VLSFR_EINVAL, "vlsfr_head_outlier_accum: bad argument");
  hipLaunchKernelGGL(head_outlier_accum_kernel, dim3(cfg->B, 2), dim3(256), 0, (hipStream_t)stream, g, queue, cfg->Q,
                     cfg->D, cfg->slot_lo, special_col, src1, src2, n_special, sel_col, sel_w, k, T, t_stride);
  VLSFR_HIP_CHECK_LAUNCH("head_outlier_accum launch");
  return VLSFR_OK;
}

int vlsfr_head_shard_partial_packed(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                                    const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                                    const int32_t* src2, int32_t n_special, int32_t n_pos, const float* sv_thr, float* packed,
                                    float* out_M, float* cand_val, int32_t* cand_col, int32_t* fixed_ref, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  if (!packed || !fixed_ref) return fail(VLSFR_EINVAL, "vlsfr_head_shard_partial_packed: null argument");
  return shard_partial_impl(cfg, p, g, queue, pool_label, special_col, src1, src2, n_special, n_pos, out_M, nullptr, nullptr,
                            nullptr, nullptr, cand_val, cand_col, workspace, workspace_bytes, stream, sv_thr, packed, fixed_ref);
}

int vlsfr_head_shard_topk_merge(const vlsfr_head_cfg* cfg, const float* cand_val, const int32_t* cand_col, int32_t world,
                                const int32_t* pool_label, int32_t n_out, int32_t* sel_col, float* sel_w, float* sel_loss,
                                void* stream) {
  if (!cfg || !cand_val || !cand_col || !pool_label || !sel_col || !sel_w || !sel_loss || world < 1 || n_out < 1 ||
      cfg->hard_neg < 1 || cfg->hard_neg > KTOP)
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_topk_merge: bad argument");
  const float inv = 1.f / ((float)n_out * (float)cfg->hard_neg);
  hipLaunchKernelGGL(head_topk_merge_kernel, dim3((2 * cfg->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, cand_val,
                     cand_col, world, cfg->B, cfg->hard_neg, pool_label, inv, sel_col, sel_w, sel_loss);
  VLSFR_HIP_CHECK_LAUNCH("head_topk_merge launch");
  return VLSFR_OK;
}

int vlsfr_head_shard_finish(const vlsfr_head_cfg* cfg, const float* packed, const float* Mg, const int32_t* pool_label,
                            const float* sel_loss, int32_t n_rows, int32_t n_pos, float* row_loss, float* loss_out, float* dP,
                            void* stream) {
  if (!cfg || !packed || !Mg || !pool_label || !row_loss || !loss_out || !dP || n_rows < 1)
    return fail(VLSFR_EINVAL, "vlsfr_head_shard_finish: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const float inv_pos = 1.f / (float)(n_pos > 0 ? n_pos : 1);
  hipLaunchKernelGGL(head_shard_finish_kernel, dim3(n_rows), dim3(256), 0, st, packed, Mg, pool_label, sel_loss, cfg->D,
                     cfg->scale, inv_pos, row_loss, dP);
  VLSFR_HIP_CHECK_LAUNCH("head_shard_finish launch");
  hipLaunchKernelGGL(head_loss_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, 2 * n_rows, loss_out);
  VLSFR_HIP_CHECK_LAUNCH("head_loss_reduce launch");
  return VLSFR_OK;
}

}  // extern "C"
