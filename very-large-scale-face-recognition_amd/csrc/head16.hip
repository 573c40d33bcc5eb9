// head_sweep16: the Dynamic-Class-Pool sweep over the bf16 SHADOW of queue[0] (D = 512) for gfx950.
//
// Same math as head_sweep_kernel (head.hip; reference ffc.py:195-201 / 248-253 + the softmax of
// F.cross_entropy, ffc.py:83/104/127, and its backward): S^T = W_tile . P^T on MFMA, online softmax in
// base 2 with deferred rescale, O += P~ . W_tile with the SAME LDS image read back transposed.  What
// is different is how the bytes and the registers are spent (the fp32-streaming
// kernel read 1 KiB of LDS per MFMA and converted on the VALU — bound by neither roof):
//   * the pool is streamed as bf16 (half the HBM bytes; the fp32 master in queue[] stays the source of
//     truth for the special columns, the precise mode and the checkpoint) straight into a 4-stage LDS
//     ring by LDS-DMA: one pool row = 1024 B = exactly one `buffer_load_dwordx4 ... lds` wave-instruction,
//     so rows keep the 32-byte pad that makes both the ds_read_b128 row reads and the
//     ds_read_b64_tr_b16 block reads bank-conflict free; rows past the chunk end are out of range for
//     the buffer descriptor and arrive as zeros.  Two tiles (66 KB) are in flight per CU behind a
//     counted s_waitcnt vmcnt, one raw s_barrier per tile.
//   * a wave owns 16 probe rows with P in registers (64 VGPRs) and O in 128 accumulator registers.  Batch <= 64: four
//     waves per workgroup, one per SIMD (HBM-bound: 6.0 TB/s of bf16 pool measured).  Larger batches: EIGHT waves per
//     workgroup = 128 rows on ONE LDS ring, two waves per SIMD, so that the softmax, the LDS latencies and the LDS-DMA
//     issue of one wave run under the MFMAs of its SIMD partner (1104 TFLOP/s = 0.44 of the bf16 MFMA peak at batch
//     256, 10 M x 512 pool; the 4-wave form of the same code: 824).  A 32-rows-per-wave form (RB = 2: every W fragment
//     feeds two MFMAs, half the LDS bytes per MFMA) was built with asm-owned accumulators a[0:255] (hipcc cannot place
//     256 accumulators + 128 registers of P: it spills P) and measured SLOWER (5.45 ms vs 4.98 ms per sweep): with one
//     wave per SIMD nothing hides that wave's own 8 LDS-DMA issues, softmax and LDS waits per tile, which is what
//     the two-waves-per-SIMD form buys; it is not in the tree.
//   * all LDS reads of the loop are inline asm behind counted lgkmcnt waits (the compiler would put a
//     vmcnt(0) in front of any LDS read it can see while a DMA is pending and drain the ring).
//   * hard-negative candidates (rows with label -1, ffc.py:86-90): a lane keeps only the admission
//     threshold of its private top-10 list in a register; the list itself lives in the partial-result
//     buffer (L2-resident) and is touched only on the rare admission.
#include "hip_common.h"
#include "head_sweep16.h"

#include <utility>

using namespace vlsfr;

namespace {

constexpr int TQ = SW16_TQ;
constexpr int DP = SW16_D;
constexpr int KTOP = SW16_KTOP;
constexpr int ROWB = DP * 2 + 32;     // LDS bytes per pool row (32-byte pad, see head.hip swz_off)
constexpr int TILE_B = TQ * ROWB;     // 33 792
constexpr int NS = 4;                 // ring stages
constexpr int KS = DP / 32;           // k-steps of the first product
constexpr int NB = DP / 16;           // 16-column blocks of the second product
constexpr int PF1 = 2;                // k-steps of fragment reads in flight (first product)
constexpr int PF2 = 2;                // column blocks of transposed reads in flight (second product)
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((address_space(3))) void lds_void_t;

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

template <int OFF>
__device__ __forceinline__ bf16x8 lds_r128(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ short4v lds_rtr(uint32_t addr) {
  short4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint32_t lds_r32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int NW, int RB, bool TOPK, bool SV>
__global__ __launch_bounds__(64 * NW, NW / 4) void head_sweep16_kernel(Sweep16Args a) {
  constexpr int NT = 64 * NW;        // threads per workgroup
  constexpr int DR = TQ / NW;        // pool rows one wave fetches per tile (one LDS-DMA each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* bits = (uint32_t*)(smem + NS * TILE_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15;
  const int h = lane >> 4;
  // XCD-aware block order (as head_sweep_kernel): the row blocks of one column chunk get ids 8 apart inside
  // one group of 8 * n_rowblk consecutive ids, so they run at the same time on one XCD and the chunk is
  // fetched from HBM once.
  const int nrb = a.n_rowblk;
  const int within = blockIdx.x % (8 * nrb);
  const int chunk = (blockIdx.x / (8 * nrb)) * 8 + (within & 7);
  const int rowblk = within >> 3;
  const int64_t c0 = (int64_t)chunk * a.chunk_cols;
  const int64_t c1 = (c0 + a.chunk_cols < a.Q) ? c0 + a.chunk_cols : a.Q;
  const int ncols = c1 > c0 ? (int)(c1 - c0) : 0;
  const int ntiles = (ncols + TQ - 1) / TQ;
  const int row_base = rowblk * (16 * RB * NW) + wave * (16 * RB);
  const bool wave_active = row_base < a.B;     // wave-uniform

  // ---- special-column bitmap of this chunk: one 32-bit word per tile (plain LDS ops: no DMA is pending yet)
  const int nwords = a.chunk_cols / TQ;
  for (int i = tid; i < nwords; i += NT) bits[i] = 0u;
  __syncthreads();
  for (int i = tid; i < a.n_special; i += NT) {
    const int64_t c = (int64_t)a.special_col[i] - a.slot_lo;   // special columns carry global slot ids
    if (c >= c0 && c < c1) atomicOr(&bits[(c - c0) >> 5], 1u << ((c - c0) & 31));
  }
  __syncthreads();

  // ---- P fragments (B operand of S^T = W . P^T): lane holds P[row 16 rb + r16][32 ks + 8 h + j]
  bf16x8 pf[RB][KS];
  float sv_thr[RB], m_ref[RB];
  bool is_out[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int prow = row_base + 16 * rb + r16;
    const bool ok = wave_active && prow < a.B;
    const f32x4* src = (const f32x4*)(a.p + (size_t)(ok ? prow : 0) * DP + h * 8);
    float nrm = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 v0 = src[ks * 8], v1 = src[ks * 8 + 1];
      if (!ok) {
        v0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        v1 = v0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pf[rb][ks][j] = (__bf16)v0[j];
        pf[rb][ks][4 + j] = (__bf16)v1[j];
        nrm += v0[j] * v0[j] + v1[j] * v1[j];
      }
      // at most four k-steps of fp32 loads in flight: the accumulators and the bf16 fragments own the register file
      if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // Fixed reference exponent of the row (log2 units).  Pool rows are unit vectors (every row of queue[] is an
    // F.normalize output: ffc.py:30 and the gallery embeddings written at ffc.py:182), so |cos| <= |p| and every
    // logit lies in [-b, b], b = qscale * |p| (SV: up to 1.4 b, ffc.py:124).  With m_ref = b_hi - 60 the terms
    // 2^(logit - m_ref) stay below 2^60 (their sum over a chunk far below the fp32 range) and above 2^-126 for
    // scale <= 64 — the online maximum and the O rescale of the fp32-streaming kernel are not needed.
    nrm = lane_step_sum<16>(nrm);
    nrm = lane_step_sum<32>(nrm);
    m_ref[rb] = a.qscale * __builtin_sqrtf(nrm) * (SV ? (a.sv_t + a.sv_t - 1.f) : 1.f) - 60.f;
    sv_thr[rb] = (SV && ok) ? a.sv_thr[prow] : 0.f;
    is_out[rb] = TOPK && ok && a.pool_label[prow] < 0;
  }

  // ---- accumulators and softmax state
  f32x4 oacc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) oacc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float l_part[RB], tk_thr[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    l_part[rb] = 0.f;      // this lane's share of sum 2^(s - m_ref)
    tk_thr[rb] = NEG_BIG;  // admission threshold of this lane's candidate list
  }
  // candidate lists of this lane: [chunk][row][h][KTOP] in the partial buffers
  auto list_base = [&](int rb) -> size_t {
    return (((size_t)chunk * a.Bp + row_base + 16 * rb + r16) * 4 + h) * KTOP;
  };
  if (TOPK && wave_active) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const size_t lb = list_base(rb);
#pragma unroll
      for (int k = 0; k < KTOP; ++k) {
        a.topk_val[lb + k] = NEG_BIG;
        a.topk_idx[lb + k] = -1;
      }
    }
  }

  // ---- LDS-DMA: wave w fetches rows DR w .. DR w + DR - 1 of a tile, one 1-KiB instruction per pool row
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.w16 + (size_t)(ncols > 0 ? c0 : 0) * DP), 0, ncols * (DP * 2), 0x00020000);
  const int voff = lane * 16;
  auto issue = [&](int t) {
    char* st = smem + (t % NS) * TILE_B + wave * DR * ROWB;
    const int soff = (t * TQ + wave * DR) * (DP * 2);   // rows past the chunk end: out of range -> zeros
#pragma unroll
    for (int i = 0; i < DR; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(st + i * ROWB), 16, voff, soff + i * (DP * 2), 0, 0);
  };
  auto issue_one = [&](int t, int i) {
    char* st = smem + (t % NS) * TILE_B + wave * DR * ROWB;
    const int soff = (t * TQ + wave * DR) * (DP * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(st + i * ROWB), 16, voff, soff + i * (DP * 2), 0, 0);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const uint32_t bits0 = lds0 + NS * TILE_B;
  const uint32_t off1 = (uint32_t)(r16 * ROWB + h * 16);                      // row reads: row r16 (+16 jb), chunk 4 ks + h
  const uint32_t off2 = (uint32_t)((4 * h + (r16 >> 2)) * ROWB + (r16 & 3) * 8);   // transposed reads: block row 4h + q, 4-column group p

#pragma unroll
  for (int t = 0; t < NS - 1; ++t) issue(t);

  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * DR) : "memory");   // this wave's rows of tile t have landed
    __builtin_amdgcn_s_barrier();                                           // everybody's; and tile t - 1 is no longer read
    // the DR LDS-DMA pieces of tile t + NS - 1 (into the slot of tile t - 1): all at once behind the barrier, or ("head_dma_spread",
    // round 4) one every SPREAD_EVERY k-steps of the first product — issued together the 8 waves queue 32 pieces on the CU's one
    // address path right when every wave also starts its fragment reads
    const bool spread = a.dma_spread && wave_active;
    if (!spread) issue(t + NS - 1);
    if (wave_active) {
      // register classes, stated once per tile: the O accumulators own the accumulator file, the P fragments
      // stay in architectural VGPRs (left to itself the allocator spills P and reloads it every k-step at RB = 2)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if constexpr (NW == 4) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+a"(oacc[rb][nb]));
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(pf[rb][ks]));
      }
      const uint32_t sb = lds0 + (uint32_t)((t % NS) * TILE_B);
      const uint32_t a1 = sb + off1, a2 = sb + off2;
      const uint32_t word = lds_r32(bits0 + 4u * (uint32_t)t);
      // ================= first product: S^T[j][i], j = 2 blocks of 16 pool columns, i = RB blocks of 16 probe rows
      f32x4 sacc[2][RB];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) sacc[jb][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 wa[PF1 + 1][2];
      static_for<PF1>([&](auto I) {
        constexpr int ks = decltype(I)::value;
        wa[ks][0] = lds_r128<ks * 64>(a1);
        wa[ks][1] = lds_r128<16 * ROWB + ks * 64>(a1);
      });
      static_for<KS>([&](auto I) {
        constexpr int ks = decltype(I)::value;
        if constexpr (ks % (KS / DR) == 0 && ks / (KS / DR) < DR) {
          if (spread) issue_one(t + NS - 1, ks / (KS / DR));
        }
        if constexpr (ks + PF1 < KS) {
          constexpr int kn = ks + PF1;
          wa[kn % (PF1 + 1)][0] = lds_r128<kn * 64>(a1);
          wa[kn % (PF1 + 1)][1] = lds_r128<16 * ROWB + kn * 64>(a1);
        }
        constexpr int later = (KS - 1 - ks) < PF1 ? (KS - 1 - ks) : PF1;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * later) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) sacc[jb][rb] = mfma16(wa[ks % (PF1 + 1)][jb], pf[rb][ks], sacc[jb][rb]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // transposed reads of the first column blocks of the second product fly under the softmax
      short4v tb[PF2 + 1][2];
      static_for<PF2>([&](auto I) {
        constexpr int nb = decltype(I)::value;
        tb[nb][0] = lds_rtr<nb * 32>(a2);
        tb[nb][1] = lds_rtr<16 * ROWB + nb * 32>(a2);
      });
      // ================= softmax: lane (r16, h) holds cos(p row 16 rb + r16, pool column tile + 16 jb + 4 h + e)
      const int64_t ct = c0 + (int64_t)t * TQ;
      const bool plain = (word == 0u) && (ct + TQ <= c1);   // wave-uniform: no masked column in this tile
      if (TOPK) {
        // hard-negative candidates of outlier rows: one comparison of the lane's best cosine of the tile against its
        // admission threshold; the insertion (global-memory list) runs only when something qualifies — rare after the
        // first tiles — and does not touch the softmax path below
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (is_out[rb]) {
            float best = NEG_BIG;
#pragma unroll
            for (int q = 0; q < 8; ++q) best = fmaxf(best, sacc[q >> 2][rb][q & 3]);
            if (best > tk_thr[rb]) {
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int jl = (q >> 2) * 16 + 4 * h + (q & 3);
                const float c = sacc[q >> 2][rb][q & 3];
                const bool ok = plain || ((ct + jl < c1) && !((word >> jl) & 1u));
                if (ok && c > tk_thr[rb]) {
                  float cv = c;
                  int ci = (int)(ct + jl);
                  const size_t lb = list_base(rb);
#pragma unroll
                  for (int k = 0; k < KTOP; ++k) {
                    const float tv = a.topk_val[lb + k];
                    const int ti = a.topk_idx[lb + k];
                    const bool gt = cv > tv;
                    a.topk_val[lb + k] = gt ? cv : tv;
                    a.topk_idx[lb + k] = gt ? ci : ti;
                    cv = gt ? tv : cv;
                    ci = gt ? ti : ci;
                  }
                  tk_thr[rb] = a.topk_val[lb + KTOP - 1];
                }
              }
            }
          }
        }
      }
      bf16x8 pa[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int jb = q >> 2, e = q & 3;
          float c = sacc[jb][rb][e];
          bool ok = true;
          if (!plain) {
            const int jl = jb * 16 + 4 * h + e;
            ok = (ct + jl < c1) && !((word >> jl) & 1u);
          }
          float fac = 1.f;
          if (SV) {
            if (c > sv_thr[rb]) {                                   // ffc.py:122-125
              c = a.sv_t * c + a.sv_t - 1.f;
              fac = a.sv_t;
            }
          }
          // 2^(logit - m_ref) against the row's FIXED reference exponent: no running maximum, no rescale of O
          float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(c, a.qscale, -m_ref[rb]));
          if (!plain) pe = ok ? pe : 0.f;
          l_part[rb] += pe;
          pa[rb][q] = (__bf16)(SV ? pe * fac : pe);
        }
      }
      // ================= second product: O[i][d] += sum_j P~[i][j] W[j][d]
      // k index of the MFMA: element q of lane group h  <->  tile row 16 (q>>2) + 4h + (q&3), which is how pa is laid
      // out; B comes from two transposed 4x16 block reads (rows 4h + .., and + 16).
      static_for<NB>([&](auto I) {
        constexpr int nb = decltype(I)::value;
        if constexpr (nb + PF2 < NB) {
          constexpr int nn = nb + PF2;
          tb[nn % (PF2 + 1)][0] = lds_rtr<nn * 32>(a2);
          tb[nn % (PF2 + 1)][1] = lds_rtr<16 * ROWB + nn * 32>(a2);
        }
        constexpr int later = (NB - 1 - nb) < PF2 ? (NB - 1 - nb) : PF2;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * later) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const short4v b0 = tb[nb % (PF2 + 1)][0], b1 = tb[nb % (PF2 + 1)][1];
        const short __attribute__((ext_vector_type(8))) bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const bf16x8 wb = __builtin_bit_cast(bf16x8, bs);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) oacc[rb][nb] = mfma16(pa[rb], wb, oacc[rb][nb]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued past the last tile

  // ---- write partials
  if (wave_active) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float l_row = l_part[rb];
      l_row = lane_step_sum<16>(l_row);
      l_row = lane_step_sum<32>(l_row);
      const int prow = row_base + 16 * rb + r16;
      const size_t pr = (size_t)chunk * a.Bp + prow;
      if (h == 0) {
        a.part_m[pr] = m_ref[rb];
        a.part_l[pr] = l_row;
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const size_t orow = (size_t)chunk * a.Bp + row_base + 16 * rb + 4 * h + e;
          a.part_o[orow * DP + nb * 16 + r16] = oacc[rb][nb][e];
        }
    }
  }
}

template <int NW, int RB>
int launch_v(const Sweep16Args& a, bool topk, bool sv, hipStream_t st) {
  const size_t lds = sweep16_lds_bytes(a.chunk_cols);
  const dim3 grid(a.n_chunks * a.n_rowblk);
#define VLSFR_SWEEP16(T, S)                                                                                      \
  do {                                                                                                          \
    auto kern = head_sweep16_kernel<NW, RB, T, S>;                                                              \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return hip_fail(e, "head_sweep16: hipFuncSetAttribute");                               \
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a);                                                  \
  } while (0)
  if (topk && sv) VLSFR_SWEEP16(true, true);
  else if (topk) VLSFR_SWEEP16(true, false);
  else if (sv) VLSFR_SWEEP16(false, true);
  else VLSFR_SWEEP16(false, false);
#undef VLSFR_SWEEP16
  VLSFR_HIP_CHECK_LAUNCH("head_sweep16 launch");
  return VLSFR_OK;
}

}  // namespace

namespace vlsfr {

size_t sweep16_lds_bytes(int chunk_cols) { return (size_t)NS * TILE_B + (size_t)(chunk_cols / TQ) * 4 + 16; }

int sweep16_rows_per_wg(int variant) { return variant == 0 ? 64 : 128; }

int launch_sweep16(const Sweep16Args& a, int variant, bool topk, bool sv, hipStream_t st) {
  if (variant == 2) return launch_sweep8(a, topk, sv, st);
  if (variant < 0 || variant > 1) return fail(VLSFR_EINVAL, "head_sweep16: variant must be 0, 1 or 2");
  if (a.chunk_cols % TQ != 0 || a.chunk_cols / TQ > SW16_MAX_TILES || a.n_chunks % 8 != 0 ||
      a.Bp != a.n_rowblk * sweep16_rows_per_wg(variant))
    return fail(VLSFR_EINVAL, "head_sweep16: inconsistent plan (chunk_cols %d, n_chunks %d, Bp %d)", a.chunk_cols, a.n_chunks, a.Bp);
  if (variant == 0) return launch_v<4, 1>(a, topk, sv, st);
  return launch_v<8, 1>(a, topk, sv, st);
}

}  // namespace vlsfr
