// Interface between head.hip (plan, special columns, finish) and head16.hip (the bf16-shadow sweep).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlsfr {

constexpr int SW16_D = 512;        // feature size the fast sweep is built for (one pool row = one 1-KiB LDS-DMA)
constexpr int SW16_TQ = 32;        // pool columns per tile
constexpr int SW16_KTOP = 10;      // max hard_neg (ffc.py:48)
constexpr int SW16_MAX_TILES = 4096;   // tiles per column chunk (special-column bitmap: one 32-bit word per tile)
constexpr int SW8_TQ = 128;        // fp8 sweep (head8.hip): pool columns per tile
constexpr int SW8_MAX_TILES = 1024;

struct Sweep16Args {
  const float* p;            // [B, 512] fp32 probe embeddings
  const uint16_t* w16;       // bf16 shadow of queue[0]: [Q][512]
  const uint8_t* w8;         // or (variant 2) the fragment-major fp8 shadow: [ceil(Q / 128)][128 KiB] (head8.hip)
  int64_t Q;
  int32_t B;
  int32_t chunk_cols;        // multiple of SW16_TQ, <= SW16_MAX_TILES * SW16_TQ
  int32_t n_chunks;          // multiple of 8
  const int32_t* special_col;
  int32_t n_special;
  const int32_t* pool_label; // [B]; rows with -1 collect hard-negative candidates
  float qscale;              // scale * log2(e)
  const float* sv_thr;       // SV: per-row hard-example threshold (cos units) or nullptr
  float sv_t;                // SV: mask_svfc (1.2)
  float* part_m;             // [n_chunks, Bp]
  float* part_l;             // [n_chunks, Bp]
  float* part_o;             // [n_chunks, Bp, 512]
  float* topk_val;           // [n_chunks, Bp, 4, KTOP]
  int32_t* topk_idx;
  int32_t Bp;                // n_rowblk * rows per workgroup
  int32_t n_rowblk;
  int32_t slot_lo;
  int32_t dma_spread = 0;    // 1: the LDS-DMA pieces of a tile are issued one by one inside the first product ("head_dma_spread")
};

// variant 0: 4 waves x 16 rows (64 probe rows per workgroup, one wave per SIMD) — batch <= 64, HBM-bound;
// variant 1: 8 waves x 16 rows (128 rows per workgroup, two waves per SIMD sharing one LDS ring: the softmax and
//            the LDS latencies of one wave run under the MFMAs of its SIMD partner);
// variant 2: the fp8 sweep of head8.hip (8 waves x 16 rows, tiles of SW8_TQ columns, Sweep16Args::w8);
// grid = n_chunks * n_rowblk workgroups, Bp = n_rowblk * sweep16_rows_per_wg(variant).
int launch_sweep16(const Sweep16Args& a, int variant, bool topk, bool sv, hipStream_t st);
int launch_sweep8(const Sweep16Args& a, bool topk, bool sv, hipStream_t st);
size_t sweep8_lds_bytes(int chunk_cols);
int sweep16_rows_per_wg(int variant);
size_t sweep16_lds_bytes(int chunk_cols);

}  // namespace vlsfr
