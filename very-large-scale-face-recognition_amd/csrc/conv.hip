// Convolution family of the iResNet / MobileFaceNet backbones on gfx950 MFMA
// (C-ABI section 5 of include/vlsfr.h).  Replaces the nn.Conv2d calls of reference
// model/resnet_arcface.py:5-23,36,39,74,120 and nn.Linear at :95 (a 1x1 convolution on a 1x1 map),
// forward, input-gradient and weight-gradient.
//
// Layouts: activations NHWC bf16; weights bf16 [rows][R][S][C] with C fastest ("KRSC" for the forward
// pass, the [Cin][R][S][Cout] transpose for the input gradient); weight gradients fp32 [Cout][R][S][Cin],
// which is exactly the memory of a torch.channels_last OIHW parameter.
//
// conv_igemm_kernel — implicit GEMM, output-channel on the MFMA row and output pixel on the MFMA
//   column, so each lane ends with 4 consecutive channels of one pixel (8-byte NHWC stores).
//   K = (r, s, c) with c fastest; C % 32 == 0, so one 32-deep k-tile never straddles a filter tap and
//   the gather is "one shifted pixel row per k-tile".  Register-staged double-buffered LDS, one
//   barrier per k-tile.  mode 0 gathers for the forward pass, mode 1 for the input gradient
//   (stride-2 taps that do not divide are zero rows).  Optional split-K with fp32 atomic output.
// conv_wgrad_kernel — dW = dY^T · gather(X): the contraction index is the pixel, which is the
//   slow axis of both NHWC operands, so both tiles go to LDS pixel-major as they lie in memory and
//   are read back with ds_read_b64_tr_b16 (hardware transpose).  Split over the pixel range with
//   fp32 atomics into the (pre-zeroed or accumulating) gradient buffer.
#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "hip_common.h"

using namespace vlsfr;

namespace vlsfr {
// ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream around every launch of
// a kernel family while profiling is enabled (ProfScope is declared in hip_common.h; head.hip uses it too).
struct ProfRec {
  hipEvent_t a, b;
  double work;
  int family;   // 0: conv_igemm (forward + input gradient), 1: conv_wgrad, 2: head_sweep, 3: conv_igemm with the fused BN-backward reduction
  bool pooled;  // events taken from g_prof_pool (returned by vlsfr_profile_reset) or created for this bracket (destroyed there)
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;   // forward and backward passes are enqueued from different host threads
// The events of the brackets come from a pool created by vlsfr_profile_enable on ITS caller's thread, before any bracketed
// launch: no hipEventCreate / hipEventDestroy on the launch path (which runs on the caller's thread for forward passes and
// on autograd's for backward passes).  (Round 3 re-ran the rocprofv3 --pmc pass that had ended in a segmentation fault in
// round 2 both ways: it completed with the pool AND with per-launch hipEventCreate, and failed once on an unchanged binary —
// event creation is not what that fault depends on; profiles/r03_pmc_full_pass_status.txt)
// A bracket that finds the pool exhausted is dropped and COUNTED (vlsfr_profile_dropped): a truncated profile is visible.
static std::vector<hipEvent_t> g_prof_pool;
static size_t g_prof_next = 0;   // guarded by g_prof_mu
static long long g_prof_dropped = 0;   // brackets not taken since the last reset (pool exhausted / hipEventCreate failed)
static int prof_pool_reserve(size_t n_events) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  while (g_prof_pool.size() < n_events) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    g_prof_pool.push_back(e);
  }
  return 0;
}
static int g_prof_pool_on = 1;   // "prof_pool": 0 = round 2's per-launch hipEventCreate (kept to reproduce the --pmc failure)
ProfScope::ProfScope(hipStream_t s, int fam, double w) : st(s), on(g_prof_on), family(fam), work(w) {
  if (!on) return;
  if (!g_prof_pool_on) {
    if (hipEventCreate(&a) != hipSuccess) on = false;
    else if (hipEventCreate(&b) != hipSuccess) {
      (void)hipEventDestroy(a);
      on = false;
    }
    if (!on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      ++g_prof_dropped;
    }
  } else {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_next + 2 + 128 > g_prof_pool.size()) {   // pool exhausted (its last 128 events belong to the overhead probe): unbracketed
      on = false;
      ++g_prof_dropped;
      return;
    }
    a = g_prof_pool[g_prof_next];
    b = g_prof_pool[g_prof_next + 1];
    g_prof_next += 2;
  }
  if (on) (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(ProfRec{a, b, work, family, g_prof_pool_on != 0});
}
extern int g_dw_wgrad_blocks;   // csrc/dw.hip
extern int g_dw_strip;          // csrc/dw.hip
extern int g_bn_chain;          // csrc/iresnet.cpp
extern int g_dgrad_bnred;       // csrc/iresnet.cpp
extern int g_wgrad_group;       // csrc/iresnet.cpp
extern int g_bn_repl;           // csrc/norm.hip
int head_set_option(const char* name, int32_t value);   // csrc/head.hip: 0 handled, < 0 error, 1 not a head option
}

namespace {

// vlsfr_set_option switches (A/B and diagnostics; the defaults are what the measurements in DESIGN.md section 8 chose)
#define VLSFR_DEFAULT_CONV_VARIANT 3
int g_use_glds = VLSFR_DEFAULT_CONV_VARIANT;   // "conv_glds": 0 register-staged kernel; LDS-DMA ring: 1 = BK64 x 4 stages, 2 = BK32 x 4,
                                               // 3 = BK64 x 2 (default), 4 = BK32 x 5, 5 = 8-wave tiles, 6 / 7 = BK32 x 3 / x 2, 8 = 256x128 4-wave,
                                               // 9 = ping-pong 8-wave tiles, 10 / 11 = 256x256 / 256x128 8-wave tiles in the standard loop
int g_use_halo = 1;          // "conv_halo": halo-patch kernel for 3x3 / stride-1 layers: 0 never, 1 the 64-channel layers (default: 64 -> 64 at 56 x 56
                             // 111.8 -> 95.1 us forward, 102.6 -> 88.1 us input gradient; serial step 108.4 -> 107.8 ms), 2 all (slower from 128 channels on)
int g_wgrad_glds = 1;        // "wgrad_glds": 1 LDS-DMA ring (conv_wgrad_glds_kernel), 0 register-staged kernel
int g_wgrad_kt = 32;         // "wgrad_kt": pixels per k-tile of the register-staged weight-gradient kernel (32 or 64)
int g_wgrad_slabs = 0;       // "wgrad_slabs": 1 = split-K slices to workspace slabs + ordered reduction (bit-reproducible weight gradients),
                             // 0 = fp32 atomics (default: measured 605 vs 532 TFLOP/s at ir100 / batch 256 — the slab stores are 64-byte
                             // row segments, no cheaper than the atomics they replace, and the reduction is a second launch)
int g_wgrad_round_up = 0;    // "wgrad_round_up": 1 = round the slice count up (may exceed wgrad_target_wgs), the round-1 rule
int g_wgrad_target = 512;    // "wgrad_target_wgs": workgroups the pixel range of the weight gradient is split into (one round of 2 per CU;
                             // fewer splits = fewer fp32 atomics: 384 measured best end to end, 1024 best for the register-staged kernel)
int g_deep_ring = 0;            // "conv_deep_ring": 4-stage LDS ring of the default 128 x 128 / 64 x 128 tiles where a launch has at most one workgroup per CU (1), everywhere (2)
int g_small_tile_wgs = 0;     // "small_tile_wgs": below this many 128 x 128 workgroups a convolution runs on 64 x 128 tiles (0: never)
int g_dgrad_classes = 1;     // "dgrad_classes": stride-2 input gradients as four parity-class launches (ConvArgs::cls)
int g_xcd_map = 1;          // "xcd_map": 1 = XCD-major workgroup order in the LDS-DMA convolution / weight-gradient kernels (xcd_major_id)
int g_conv_p8 = 1;           // "conv_p8": the four-phases-per-k-tile schedule (conv_igemm_p8_kernel) on the full 256-channel one-round tiles
int g_conv_hp8 = 1;          // "conv_hp8": the halo-patch four-phase kernel (conv_igemm_hp8_kernel) on the 3x3 / stride-1 layers whose tiles fill the chip:
                             // 1 = 256-row and 128-row tiles (default), 2 = 256-row tiles only, 3 = 128-row tiles only, 0 = off
int g_conv_hw4 = 1;          // "conv_hw4": the one-wave-per-SIMD software-pipelined form of that kernel (conv_igemm_hw4_kernel) where conv_hp8 applies
int g_conv_bnin = 0;         // "conv_bnin": vlsfr_conv2d_fwd_bnin (BatchNorm / PReLU of the input applied in conv_igemm_hw4_kernel's operand path) offered to the
                             // executors.  OFF by default — measured (scripts/bnin_micro.py, batch 256): bn_apply + conv 71.1 us against 70.4 (no a_out) /
                             // 79.5 us (a_out written) fused on the 256-channel layers, 84.9 against 92.4 / 102.1 on the 128-channel ones; the step
                             // 87.0 against 83.5 ms: with ONE wave per SIMD the transform's ~200 extra instructions per k-tile sit in the MFMA wave's own
                             // issue slots, and bn_apply was half hidden beside the other stream's convolutions anyway
int g_hw4_rounds = 0;        // "hw4_rounds": multi-round launches of conv_igemm_hw4_kernel go out round by round with the tiles dealt evenly (launch_igemm_hw4)
int g_hw4_208 = 0;           // "hw4_208": 256 x 208 tiles (conv_igemm_hw4_kernel<256, 13>) where they need no more rounds than 256 x 224 (1), wherever they fit (2).  OFF: the kernel alone is no faster (57.6 against 58.0 us) and the STEP is 2.5 ms slower — 242 workgroups leave 14 CUs instead of 32 to the other chain's BatchNorm kernels (profiles/r04_hw4_208_tiles_ab.txt)
int g_hw4_64 = 1;            // "hw4_64": the 64-channel 3x3 / stride-1 layers on conv_igemm_hw4_kernel<64, 14> (64 x 896 tiles) instead of conv_igemm_halo_kernel
int g_hw4_red = 1;           // "hw4_red": conv_igemm_hw4_kernel accumulates the BatchNorm-backward reduction in its epilogue when asked to (0: stand-alone kernel)
int g_hp8_fill = 80;         // "hp8_fill": least percentage of the workgroup slots of its rounds (256 per round) that conv_igemm_hp8_kernel must fill
int g_tile256_min = 129;     // "tile256_min": the one-round 8-wave tiles are taken from 256 * this many pixels on, i.e. as soon as the 128 x 128 tiling
                             // (2 cout tiles x P / 128) no longer fits the 512 resident slots: batch 192, 37 632 pixels: 60.7 vs 68.9 us; batch 160
                             // (490 tiles of 128 x 128, one round): 43.8 vs 58.8 us the other way
int g_tile224 = 1;           // "tile224": 256 x 224 tiles where those still make one round of the 256 CUs (ir100 at batch 256: 224 tiles, not 196)
int g_tile256 = 1;           // "tile256": 256 x 256 tiles for the 256-channel layers whose pixel count makes one round of them (run_igemm)
int g_bnred_all = 0;         // "bnred_all": 1 = the fused BatchNorm-backward reduction on every eligible launch (default: where it pays)
int g_conv_dbg = 0;          // "conv_dbg": weight-gradient diagnostics (1 skips the epilogue atomics, 2 the k loop)
long long* g_conv_trace = nullptr;   // device buffer [2][64][8] stamps, set by vlsfr_conv_trace

struct ConvArgs {
  const u16* x;      // gathered activations [Nimg, H, W, C] bf16
  const u16* w;      // [Mrows][R][S][C] bf16
  void* y;           // bf16 [P, Mrows] (splitk == 1) or fp32 [P, Mrows] accumulated atomically
  int Nimg, H, W, C;
  int Ho, Wo;        // output pixels P = Nimg * Ho * Wo
  int Mrows;
  int R, S, stride, pad;
  int mode;          // 0: forward gather, 1: input-gradient gather
  int splitk;
  int out_f32;
  double* stats;     // optional [VLSFR_BN_REPL][2][Mrows] float64 BatchNorm statistics (sum, sum of squares) of the rounded output
  long long* trace;  // diagnostics: per-phase clock stamps of waves 0 and 4 of one workgroup (vlsfr_conv_trace), or nullptr
  int gx = 0, gy = 0, xcd = 0;   // xcd != 0: launched as a 1-D grid of gx * gy * splitk workgroups in XCD-major order (xcd_major_id)
  int tile0 = 0;                 // conv_igemm_hw4_kernel launched round by round ("hw4_rounds"): logical id of this launch's workgroup 0
  // Parity-class launch of a stride-2 input gradient (vlsfr_conv2d_dgrad): the input positions (2 h' + ph, 2 w' + pw) of one
  // parity class see a stride-1 convolution of dY with the 1, 2 or 4 filter taps of matching parity, so each class is run
  // as a forward-mode gather over the dY grid with a subset of a virtual 3 x 3 (or 1 x 1) filter and its rows scattered
  // with stride 2 — the taps a position cannot see are never fetched or multiplied (they were 3/4 of the old kernel's work).
  unsigned tap_mask = 0;          // virtual taps present (bit r' * S + s'); 0: every tap (ordinary launch)
  unsigned long long tap_w = 0;   // 4 bits per virtual tap: the tap of the weight matrix it multiplies
  int cls = 0;                    // 0: dense output rows p; 1 + 2 ph + pw: row of pixel (n, h', w') = (n Hf + 2 h' + ph) Wf + 2 w' + pw
  int Hf = 0, Wf = 0;
  // Input-gradient launches whose output is the dY of a BatchNorm (+ PReLU) backward (vlsfr_conv2d_dgrad_bnred): the epilogue
  // holds the tile of dY it has just rounded, reads the matching tile of that layer's input x once, and accumulates the
  // layer's reduction (sum dz, sum dz * xhat, sum dy * min(z, 0); norm.hip bn_bwd_reduce_kernel) — that kernel, which read
  // dY and x again from HBM, is no longer launched.
  const u16* red_x = nullptr;     // [rows of y][Mrows] bf16
  const float* red_mean = nullptr;
  const float* red_invstd = nullptr;
  const float* red_gamma = nullptr;
  const float* red_beta = nullptr;
  const float* red_slope = nullptr;   // PReLU slopes or nullptr (plain BatchNorm)
  float* red_out = nullptr;       // [VLSFR_BN_REPL][3][Mrows], pre-zeroed
  // Forward launches whose input is a BatchNorm (+ PReLU) of x applied in the operand path (vlsfr_conv2d_fwd_bnin, conv_igemm_hw4_kernel<XF>):
  const float* xf_scale = nullptr;   // [C] gamma * invstd
  const float* xf_shift = nullptr;   // [C] beta - mean * scale
  const float* xf_slope = nullptr;   // [C] PReLU slopes or nullptr
  u16* xf_out = nullptr;             // the transformed input [rows of x][C] bf16 as a by-product, or nullptr
  int repl = 1;                   // replicas of the per-channel accumulators in use (vlsfr::g_bn_repl)
  int dbg = 0;                    // diagnostics ("conv_dbg"): 4 skip the x-tile DMA, 8 skip the reduction arithmetic, 16 skip its atomics
};

// Workgroups are dealt to the 8 XCDs round-robin in dispatch order (id % 8), and each XCD has its own L2.  This maps
// dispatch id -> a logical id such that every XCD owns one CONTIGUOUS range of logical ids (bijective for any n): tiles
// that read the same rows (the cout tiles of a pixel tile and its neighbours; the (tap, channel) tiles of one pixel
// slice of the weight gradient) then run at the same time behind the same L2 instead of being fetched once per XCD.
__device__ __forceinline__ int xcd_major_id(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// LDS image of a k-tile: [rows][BK] bf16, the 16-byte chunks of a row XOR-swizzled with the row so
// that the ds_read_b128 fragment reads (row = lane & 15, chunk = 4 kk + (lane >> 4)) are
// bank-conflict free for every 16-lane service group of the instruction:
//   BK = 32 (64-byte rows):  chunk ^ g[(row >> 2) & 3], g = {0, 2, 3, 1}
//   BK = 64 (128-byte rows): chunk ^ (row & 7)
template <int BK>
__device__ __forceinline__ int swz(int row) {
  if constexpr (BK == 32) return (0x78 >> (2 * ((row >> 2) & 3))) & 3;
  else return row & 7;
}

template <int BM, int BN, int WM, int WN, int MT, int NT, int NW, bool RED = false, bool XWAIT = false, bool PLAIN = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NT], int m0, int p0, int P, int wm, int wn,
                                              int r16, int h, int tid, float* red_lds, const char* x_lds = nullptr);

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int MT = BM / 32;   // 16-row MFMA tiles per wave along output channels
  constexpr int NT = BN / 32;   // along pixels
  constexpr int CPR = BK / 8;   // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;            // rows staged per pass of the 256 threads
  constexpr int ACH = BM / RPP;             // chunks per thread, weight tile
  constexpr int BCH = BN / RPP;             // pixel tile
  constexpr int RSB = BK * 2;               // LDS row bytes
  __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * RSB];
  char* sA = smem;
  char* sB = smem + 2 * BM * RSB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = a.Nimg * a.Ho * a.Wo;
  const int K = a.R * a.S * a.C;
  const int m0 = blockIdx.y * BM;
  const int p0 = blockIdx.x * BN;
  const int nkt = K / BK;
  // split-K range of k-tiles
  const int per = (nkt + a.splitk - 1) / a.splitk;
  const int kt0 = blockIdx.z * per;
  const int kt1 = (kt0 + per < nkt) ? kt0 + per : nkt;
  if (kt0 >= kt1) return;

  // ---- per-thread staging coordinates
  const int srow = tid / CPR;
  const int chunk = tid % CPR;
  int b_n[BCH], b_h[BCH], b_w[BCH];
  bool b_ok[BCH];
#pragma unroll
  for (int u = 0; u < BCH; ++u) {
    const int p = p0 + srow + RPP * u;
    b_ok[u] = p < P;
    const int pp = b_ok[u] ? p : 0;
    const int n = pp / (a.Ho * a.Wo);
    const int rem = pp - n * a.Ho * a.Wo;
    const int ho = rem / a.Wo;
    const int wo = rem - ho * a.Wo;
    b_n[u] = n;
    if (a.mode == 0) {
      b_h[u] = ho * a.stride - a.pad;
      b_w[u] = wo * a.stride - a.pad;
    } else {
      b_h[u] = ho + a.pad;
      b_w[u] = wo + a.pad;
    }
  }
  uint4 ra[ACH], rb[BCH];

  auto issue = [&](int kt) {
    const int k0 = kt * BK;
    const int tap = k0 / a.C;
    const int c0 = k0 - tap * a.C;
    const int r = tap / a.S;
    const int s = tap - r * a.S;
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
      const int m = m0 + srow + RPP * u;
      if (m < a.Mrows) ra[u] = *(const uint4*)(a.w + (size_t)m * K + k0 + chunk * 8);
      else ra[u] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      int hi, wi;
      bool ok = b_ok[u];
      if (a.mode == 0) {
        hi = b_h[u] + r;
        wi = b_w[u] + s;
      } else {
        const int th = b_h[u] - r, tw = b_w[u] - s;
        if (a.stride == 2) {
          ok = ok && !((th | tw) & 1);
          hi = th >> 1;
          wi = tw >> 1;
        } else {
          hi = th;
          wi = tw;
        }
      }
      ok = ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      if (ok) rb[u] = *(const uint4*)(a.x + (((size_t)b_n[u] * a.H + hi) * a.W + wi) * a.C + c0 + chunk * 8);
      else rb[u] = make_uint4(0, 0, 0, 0);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
      const int row = srow + RPP * u;
      *(uint4*)(sA + buf * BM * RSB + row * RSB + ((chunk ^ swz<BK>(row)) << 4)) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int row = srow + RPP * u;
      *(uint4*)(sB + buf * BN * RSB + row * RSB + ((chunk ^ swz<BK>(row)) << 4)) = rb[u];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(kt0);
  stage(0);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int rd_off = r16 * RSB + (((kk * 4 + h) ^ swz<BK>(r16)) << 4);
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[i] = *(const bf16x8*)(sA + buf * BM * RSB + (wm * (BM / 2) + i * 16) * RSB + rd_off);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j] = *(const bf16x8*)(sB + buf * BN * RSB + (wn * (BN / 2) + j * 16) * RSB + rd_off);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < kt1) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue shared with the LDS-DMA kernels (16-byte stores, DPP statistics); smem is free after the loop's last barrier
  conv_epilogue<BM, BN, 2, 2, MT, NT, 4>(a, acc, m0, p0, P, wm, wn, r16, h, tid, (float*)smem);
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_glds_kernel — the same implicit GEMM with a 4-stage LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write) and one raw barrier per 64-deep
// k-tile.  The register-staged kernel above keeps one k-tile in flight per workgroup and is bound by
// memory latency (~10x the MFMA time of a tile); here three tiles (96 KB) are in flight per CU behind
// counted s_waitcnt vmcnt(N), so the MFMAs of tile t overlap the loads of tiles t+1..t+3.
//   * LDS image per stage: A [BM][64] and B [BN][64] bf16, 128-byte rows, 16-byte chunks XOR-swizzled
//     with (row & 7).  An LDS-DMA wave-instruction writes 1 KiB linearly (8 rows x 8 chunks), so the
//     swizzle is applied to the per-lane SOURCE address: lane (r = lane >> 3, c = lane & 7) fetches
//     logical chunk c ^ r of its row.
//   * both operands go through raw buffer descriptors (buffer_load_dwordx4 ... lds): padding and
//     out-of-range rows carry an out-of-range offset, for which the DMA writes zeros.
//   * fragment reads are inline-asm ds_read_b128 behind an explicit lgkmcnt(0): the compiler cannot
//     see that they touch the DMA'd bytes, so it inserts no vmcnt(0) in front of them.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128_asm(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int STRIDE, int BASE, int... I>
__device__ __forceinline__ void lds_read_frags(bf16x8* f, uint32_t addr, std::integer_sequence<int, I...>) {
  ((f[I] = lds_read128_asm<BASE + I * STRIDE>(addr)), ...);
}

// Shared epilogue of the LDS-DMA convolution kernels: lane (r16, h) of wave (wm, wn) holds channels
// m0 + wm*(BM/WM) + 16 i + 4h + e of pixels p0 + wn*(BN/WN) + 16 j + r16.  fp32 output (plain or split-K
// atomics) or bf16 output with the fused BatchNorm statistics; red_lds = BM*WN*2 floats of LDS nobody reads.
// XWAIT (with RED): the caller issued the LDS-DMA of the x tile right before this call instead of under its last k-tile; the
// output stores above go first and the reduction waits for the tile (and a workgroup barrier) itself.
// PLAIN: the caller never asks for fp32 output, split-K or a parity-class launch (run_igemm keeps those off conv_igemm_hw4_kernel):
// those paths are not compiled into it.
template <int BM, int BN, int WM, int WN, int MT, int NT, int NW, bool RED, bool XWAIT, bool PLAIN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NT], int m0, int p0, int P, int wm, int wn,
                                              int r16, int h, int tid, float* red_lds, const char* x_lds) {
  // ---- epilogue: lane holds channels m = .. + 4h + e (e = 0..3) of pixel .. + r16
  const int mw = m0 + wm * (BM / WM) + 4 * h;     // + 16 i
  const int pw = p0 + wn * (BN / WN) + r16;       // + 16 j
  if (!PLAIN && a.out_f32) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int p = pw + j * 16;
      if (p >= P) continue;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = mw + i * 16;
        if (m >= a.Mrows) continue;
        float* dst = (float*)a.y + (size_t)p * a.Mrows + m;
        if (a.splitk > 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[i][j][e]);
        } else {
          *(f32x4*)dst = acc[i][j];
        }
      }
    }
    return;
  }
  // bf16 output.  16-byte stores: rows h and h ^ 1 of the MFMA layout hold channels 4h..4h+3 and 4h+4..4h+7 of the same
  // pixel, so v_permlane16_swap hands the even rows both halves of pixel tile j and the odd rows both
  // halves of tile j + 1 -- 8 dwordx4 stores per lane instead of 16 dwordx2 (the store tail of a workgroup
  // is issue-bound, not bandwidth-bound).
  // An odd NT (the 224-pixel tile: 7 pixel tiles per wave) stores its last tile as it stands, 8 bytes per lane.
  const bool odd = h & 1;
  // output row of the pixel tile this lane stores in pair jp (tile 2 jp + odd): the pixel index itself, or — parity-class
  // launch — the position (n, 2 h' + ph, 2 w' + pw) of pixel (n, h', w') of the class grid, advanced 32 pixels per pair
  // by carries (one division pair per lane)
  int64_t orow[NT / 2 + 1];
  if (!PLAIN && a.cls) {
    const int ph = (a.cls - 1) >> 1, pwc = (a.cls - 1) & 1;
    const int q = pw + (odd ? 16 : 0);
    const int HWc = a.Ho * a.Wo;
    int n = q / HWc;
    const int rem = q - n * HWc;
    int hh = rem / a.Wo;
    int ww = rem - hh * a.Wo;
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
      orow[jp] = ((int64_t)n * a.Hf + 2 * hh + ph) * a.Wf + 2 * ww + pwc;
      ww += 32;
      while (ww >= a.Wo) {
        ww -= a.Wo;
        if (++hh >= a.Ho) {
          hh = 0;
          ++n;
        }
      }
    }
  } else {
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) orow[jp] = pw + (2 * jp + (odd ? 1 : 0)) * 16;
  }
  const int Pst = (a.dbg & 512) ? 0 : P;        // diagnostic 512: every output store masked off
  if (a.dbg & 1024) {                           // diagnostic 1024: every workgroup stores to the first 256 rows (cache-resident lines)
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) orow[jp] &= 255;
  }
  u16* ybase = (u16*)a.y + (mw - 4 * (h & 1));
  // Phase 1 — round, pair, store: every output store of the wave is issued before any of the statistics' arithmetic, so the
  // stores drain (all workgroups of a round write at the same time: the drain is bandwidth-bound) UNDER that arithmetic instead
  // of behind it.  pk[i][j] = the four rounded values of tile (i, j) as two packed pairs (channels e = 0, 1 | 2, 3): what the
  // statistics and the fused reduction below read, so that they see exactly what the consumer will.
  uint32_t pk[MT][NT][2];
  auto store = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = 2 * jp + t;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[i][j][e];
          const uint2 u = __builtin_bit_cast(uint2, o);
          pk[i][j][0] = u.x;
          pk[i][j][1] = u.y;
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[i][2 * jp][0], pk[i][2 * jp + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[i][2 * jp][1], pk[i][2 * jp + 1][1], false, false);
        const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        const int j = 2 * jp + (odd ? 1 : 0);
        const bool okst = FULL || (pw + j * 16 < Pst && mw - 4 * (h & 1) + i * 16 < a.Mrows);
        if (okst) *(uint4*)(ybase + (size_t)orow[jp] * a.Mrows + i * 16) = v;
      }
    }
  };
  const bool full_tile = p0 + BN <= Pst && m0 + BM <= a.Mrows;
  if (full_tile) store(std::true_type{});
  else store(std::false_type{});
  if constexpr (NT % 2 == 1) {   // the unpaired last pixel tile (never a parity-class launch: run_igemm)
    constexpr int j = NT - 1;
    const bool pok = full_tile || pw + j * 16 < P;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool ok = pok && (full_tile || mw + i * 16 < a.Mrows);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[i][j][e];
      const uint2 u = __builtin_bit_cast(uint2, o);
      pk[i][j][0] = u.x;
      pk[i][j][1] = u.y;
      if (ok && pw + j * 16 < Pst) *(uint2*)((u16*)a.y + (size_t)((a.dbg & 1024) ? ((pw + j * 16) & 255) : pw + j * 16) * a.Mrows + mw + i * 16) = u;
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // (phase 2 stays behind the last store)
  // rounded value e of tile (i, j) as a float
  auto rounded = [&](int i, int j, int e) __attribute__((always_inline)) {
    const uint32_t wd = pk[i][j][e >> 1];
    return __uint_as_float((e & 1) ? (wd & 0xffff0000u) : (wd << 16));
  };
  // Per-channel sums -> one of VLSFR_BN_REPL replicated accumulators out[rep][q][Mrows].  A lane's partial for channel
  // 16 i + 4 h + e (of this wave's channel range) is summed over the 16 pixel lanes of its row with DPP adds (no LDS round
  // trips; row16_fold16) and kept by lane r16 = 4 (i & 3) + e; the WN pixel halves of the workgroup meet in the LDS stage the last k-tile
  // did not use (its readers all passed the last barrier), then ONE global atomic per channel, quantity and workgroup.
  constexpr int NR = (MT + 3) / 4;   // rounds of 16 (i, e) values: one value per pixel lane of the row
  auto flush = [&](auto nq_tag, float (&keep)[decltype(nq_tag)::value][NR], float* out) {
    constexpr int NQ = decltype(nq_tag)::value;
    float* red = red_lds;   // [WN][NQ][BM]
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ni = MT - 4 * r < 4 ? MT - 4 * r : 4;
      if (r16 < ni * 4) {
        const int ml = wm * (BM / WM) + (4 * r + (r16 >> 2)) * 16 + 4 * h + (r16 & 3);
#pragma unroll
        for (int q = 0; q < NQ; ++q) red[(wn * NQ + q) * BM + ml] = keep[q][r];
      }
    }
    __syncthreads();
    float* dst = out + (size_t)(blockIdx.x % a.repl) * NQ * a.Mrows;
    for (int i = tid; i < NQ * BM; i += NW * 64) {
      const int k = i / BM, ml = i - k * BM;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) t += red[(w * NQ + k) * BM + ml];
      if (m0 + ml < a.Mrows) atomicAdd(dst + (size_t)k * a.Mrows + m0 + ml, t);
    }
  };
  if (!RED && a.stats) {   // Phase 2 — fused BatchNorm statistics of the rounded output (forward launches)
    // The statistics are taken from the ROUNDED values (what the consumer reads), as sums of DEVIATIONS from a pivot: the value of
    // the first pixel of this wave's pixel range for the channel (lane r16 = 0 of the 16-lane row holds it; one ds_bpermute per
    // channel hands it to the row).  Deviations are of the order of the spread whatever the mean, so fp32 is enough up to the
    // workgroup level; (sum x, sum x^2) are formed and added in float64 there.  One channel tile i at a time (pixel tiles in
    // ascending order, as ever: the same sums bit for bit), folded over the row's 16 pixel lanes as soon as it is complete.
    float keep[3][NR];   // sum of deviations, sum of squared deviations, pivot
#pragma unroll
    for (int r = 0; r < NR; ++r) keep[0][r] = keep[1][r] = keep[2][r] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // (a round = four channel tiles = the 16 (i & 3, e) values of one lane; row16_fold16 leaves the row sum of value k with lane r16 = k)
    auto round_stats = [&](auto rc, auto full_tag) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      constexpr bool FULL = decltype(full_tag)::value;
      float S[16], Q[16], PV[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) S[k] = Q[k] = PV[k] = 0.f;
      static_for<(MT - 4 * r < 4 ? MT - 4 * r : 4)>([&](auto iic) {
        constexpr int ii = decltype(iic)::value, i = 4 * r + ii;
        f32x2 cp2[2], cs2[2], cq2[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          cp2[g][0] = __shfl(rounded(i, 0, 2 * g), (tid & 63) & 48, 64);
          cp2[g][1] = __shfl(rounded(i, 0, 2 * g + 1), (tid & 63) & 48, 64);
          cs2[g] = (f32x2){0.f, 0.f};
          cq2[g] = (f32x2){0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bool ok = FULL || (pw + j * 16 < P && mw + i * 16 < a.Mrows);
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            f32x2 f = (f32x2){rounded(i, j, 2 * g), rounded(i, j, 2 * g + 1)} - cp2[g];
            if (!ok) f = (f32x2){0.f, 0.f};
            cs2[g] += f;
            cq2[g] += f * f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S[ii * 4 + e] = cs2[e >> 1][e & 1];
          Q[ii * 4 + e] = cq2[e >> 1][e & 1];
          PV[ii * 4 + e] = cp2[e >> 1][e & 1];
        }
      });
      keep[0][r] = row16_fold16(S, r16);
      keep[1][r] = row16_fold16(Q, r16);
      float pv = 0.f;   // the row shares one pivot: no sum
#pragma unroll
      for (int k = 0; k < 16; ++k) pv = r16 == k ? PV[k] : pv;
      keep[2][r] = pv;
    };
    if (full_tile) static_for<NR>([&](auto rc) { round_stats(rc, std::true_type{}); });
    else static_for<NR>([&](auto rc) { round_stats(rc, std::false_type{}); });
    float* red = red_lds;   // [WN][3][BM]
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ni = MT - 4 * r < 4 ? MT - 4 * r : 4;
      if (r16 < ni * 4) {
        const int ml = wm * (BM / WM) + (4 * r + (r16 >> 2)) * 16 + 4 * h + (r16 & 3);
#pragma unroll
        for (int q = 0; q < 3; ++q) red[(wn * 3 + q) * BM + ml] = keep[q][r];
      }
    }
    __syncthreads();
    double* dst = a.stats + (size_t)(blockIdx.x % a.repl) * 2 * a.Mrows;
    for (int ml = tid; ml < BM; ml += NW * 64) {
      double S = 0.0, Q = 0.0;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        int nv = P - (p0 + w * (BN / WN));                       // valid pixels of pixel range w
        nv = nv < 0 ? 0 : (nv > BN / WN ? BN / WN : nv);
        const double n = (double)nv, sd = (double)red[(w * 3 + 0) * BM + ml], qd = (double)red[(w * 3 + 1) * BM + ml],
                     pv = (double)red[(w * 3 + 2) * BM + ml];
        S += n * pv + sd;
        Q += qd + 2.0 * pv * sd + n * pv * pv;
      }
      if (m0 + ml < a.Mrows) {
        atomicAdd(dst + m0 + ml, S);
        atomicAdd(dst + (size_t)a.Mrows + m0 + ml, Q);
      }
    }
  }
  if constexpr (RED) {
    // BatchNorm-backward reduction fused into an input-gradient launch (ConvArgs::red_x).  The x tile [BN pixels][BM
    // channels] of this output tile was brought into the LDS stage the last k-tile did not use, by LDS-DMA issued under that
    // k-tile's MFMAs (conv_igemm_glds_kernel: issue_x) — no exposed memory latency, no staging registers; its 16-byte chunks
    // are XOR-swizzled with the pixel row, so the 8-byte reads below (lane = 4 channels of one pixel, the accumulator
    // layout) are bank-conflict free.
    // dz = dy * prelu'(z), z = bn(x) = x * zs + zo; sums of dz, dz * (x - mean) (times invstd = dz * xhat) and, for the
    // PReLU slope gradient, dy * z over z <= 0 — from the ROUNDED dy (what the BatchNorm backward reads back)
    constexpr int XCPR = BM / 8;          // 16-byte chunks per x row
    if constexpr (XWAIT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the x tile (and, older or younger, its output stores)
      __syncthreads();
    }
    const uint32_t x_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)x_lds;
    const bool prelu = a.red_slope != nullptr;
    float keep[3][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) keep[0][r] = keep[1][r] = keep[2][r] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // a round = four channel tiles; per tile the three sums of this lane's 4 channels over its NT pixels, two channels per packed
    // fp32 instruction; then the transposing fold over the row's 16 pixel lanes (row16_fold16)
    auto round_red = [&](auto rc, auto full_tag, auto prelu_tag) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      constexpr bool FULL = decltype(full_tag)::value, PRELU = decltype(prelu_tag)::value;
      float S0[16], S1[16], S2[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) S0[k] = S1[k] = S2[k] = 0.f;
      static_for<(MT - 4 * r < 4 ? MT - 4 * r : 4)>([&](auto iic) {
        constexpr int ii = decltype(iic)::value, i = 4 * r + ii;
        const int mc = mw + i * 16;
        const bool mok = FULL || mc < a.Mrows;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
        const f32x4 c_mean = mok ? *(const f32x4*)(a.red_mean + mc) : zero4;
        const f32x4 c_is = mok ? *(const f32x4*)(a.red_invstd + mc) : zero4;
        f32x4 zs = one4, zo = zero4, sl = one4;
        if (PRELU && mok) {
          const f32x4 g = a.red_gamma ? *(const f32x4*)(a.red_gamma + mc) : one4;
          const f32x4 b = a.red_beta ? *(const f32x4*)(a.red_beta + mc) : zero4;
          sl = *(const f32x4*)(a.red_slope + mc);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            zs[e] = c_is[e] * g[e];
            zo[e] = b[e] - c_mean[e] * zs[e];
          }
        }
        f32x2 s0[2], s1[2], s2[2], mean2[2], zs2[2], zo2[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          s0[g] = s1[g] = s2[g] = (f32x2){0.f, 0.f};
          mean2[g] = (f32x2){c_mean[2 * g], c_mean[2 * g + 1]};
          zs2[g] = (f32x2){zs[2 * g], zs[2 * g + 1]};
          zo2[g] = (f32x2){zo[2 * g], zo[2 * g + 1]};
        }
        const int lchunk = (wm * (BM / WM) + i * 16 + 4 * h) >> 3;     // logical 16-byte chunk of this lane's 4 channels
        uint2 xr[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wn * (BN / WN) + j * 16 + r16;
          const uint32_t addr = x_base + (uint32_t)(row * (BM * 2) + ((lchunk ^ (row & (XCPR - 1))) << 4) + ((h & 1) << 3));
          // (a plain LDS load, NOT an inline-asm ds_read with a wait further down: under register pressure the compiler copies an
          // asm statement's outputs elsewhere right behind the statement — before the data has arrived, which it cannot know)
          xr[j] = *(const __attribute__((address_space(3))) uint2*)(uintptr_t)addr;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bool ok = FULL || (pw + j * 16 < P && mok);
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const uint32_t wd = g ? xr[j].y : xr[j].x;
            const f32x2 x2 = {__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u)};
            f32x2 dy2 = {rounded(i, j, 2 * g), rounded(i, j, 2 * g + 1)};
            if (!ok) dy2 = (f32x2){0.f, 0.f};
            f32x2 dz2 = dy2;
            if constexpr (PRELU) {
              const f32x2 z2 = x2 * zs2[g] + zo2[g];
              const bool n0 = z2[0] <= 0.f, n1 = z2[1] <= 0.f;
              const f32x2 t2 = dy2 * z2;
              s2[g] += (f32x2){n0 ? t2[0] : 0.f, n1 ? t2[1] : 0.f};
              dz2 = dy2 * (f32x2){n0 ? sl[2 * g] : 1.f, n1 ? sl[2 * g + 1] : 1.f};
            }
            s0[g] += dz2;
            s1[g] += dz2 * (x2 - mean2[g]);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S0[ii * 4 + e] = s0[e >> 1][e & 1];
          S1[ii * 4 + e] = s1[e >> 1][e & 1] * c_is[e];
          S2[ii * 4 + e] = s2[e >> 1][e & 1];
        }
      });
      keep[0][r] = row16_fold16(S0, r16);
      keep[1][r] = row16_fold16(S1, r16);
      if constexpr (PRELU) keep[2][r] = row16_fold16(S2, r16);
    };
    if (!(a.dbg & 8)) {
      if (full_tile) {
        if (prelu) static_for<NR>([&](auto rc) { round_red(rc, std::true_type{}, std::true_type{}); });
        else static_for<NR>([&](auto rc) { round_red(rc, std::true_type{}, std::false_type{}); });
      } else {
        if (prelu) static_for<NR>([&](auto rc) { round_red(rc, std::false_type{}, std::true_type{}); });
        else static_for<NR>([&](auto rc) { round_red(rc, std::false_type{}, std::false_type{}); });
      }
    }
    __syncthreads();   // every wave has read its part of the x tile: the sums' scratch below lives in the same LDS stage
    if (!(a.dbg & 16)) flush(std::integral_constant<int, 3>{}, keep, a.red_out);
  }
}

template <int BM, int BN, int BK, int NST, int NW, bool PP = false, bool SWP = false, bool RED = false>
__global__ __launch_bounds__(NW * 64, ((NW == 4 && BM * BN >= 256 * 128) || (NW == 8 && BM * BN <= 128 * 128) || SWP) ? 2 : 1) void conv_igemm_glds_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor builtins exist in the device pass only
  // wave grid WM x WN (NW waves): each wave keeps (BM / WM) x (BN / WN) of the tile; the 8-wave
  // 256 x 128 / 128 x 256 tiles raise the FLOPs per byte a CU has to pull from L2 by a third over
  // the 4-wave 128 x 128 tile (the per-CU load path, not HBM, is what bounds this kernel)
  constexpr int WM = (NW == 8) ? (BM / 64) : 2, WN = NW / WM;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int RSB = BK * 2;           // LDS row bytes
  constexpr int CPR = BK / 8;           // 16-byte chunks per row
  constexpr int RPI = 1024 / RSB;       // rows covered by one 1-KiB LDS-DMA wave-instruction
  constexpr int AI = BM / (NW * RPI);   // LDS-DMA instructions per wave and stage, weight tile
  // pixel tile: BROWS instructions; they divide evenly over the waves (wave w issues BI consecutive ones) or — the
  // 224-pixel tile: 28 instructions on 8 waves — are dealt round-robin (instruction i * NW + w), the last round partial
  constexpr int BROWS = BN / RPI;
  constexpr bool BSPLIT = BROWS % NW != 0;
  constexpr int BI = (BROWS + NW - 1) / NW;
  static_assert(BM % (NW * RPI) == 0 && BN % RPI == 0 && (!BSPLIT || NST == 2), "tile rows per LDS-DMA instruction");
  constexpr int STAGE = (BM + BN) * RSB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int P = a.Nimg * a.Ho * a.Wo;
  const int K = a.R * a.S * a.C;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {   // cout tile fastest, then pixel tile: neighbours in the logical order share activations (and the halo rows)
    const int g = xcd_major_id(blockIdx.x, a.gx * a.gy * a.splitk);
    const int t = g / a.gy;
    by = g - t * a.gy;
    bz = t / a.gx;
    bx = t - bz * a.gx;
  }
  const int m0 = by * BM;
  const int p0 = bx * BN;
  // taps of the filter this launch walks: all R * S of them, or the subset of a parity-class launch
  const int ntap_all = a.R * a.S;
  unsigned long long tap_list = 0;   // 4 bits per entry: virtual tap ids in ascending order
  int ntap = 0;
  for (int t = 0; t < ntap_all; ++t)
    if (!a.tap_mask || ((a.tap_mask >> t) & 1u)) tap_list |= (unsigned long long)t << (4 * ntap++);
  const int nkt = ntap * (a.C / BK);
  const int per = (nkt + a.splitk - 1) / a.splitk;
  const int kt0 = bz * per;
  const int kt1 = (kt0 + per < nkt) ? kt0 + per : nkt;
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  // ---- per-lane gather descriptors
  const int rsub = lane / CPR;
  const int lchunk = (lane % CPR) ^ swz<BK>(rsub);   // logical 16-byte chunk this lane fetches
  // Both operands are fetched through raw buffer descriptors: a lane whose row / tap is padding gets
  // the offset 0x80000000, which is out of range for every tensor this path accepts (< 2 GiB), and the
  // LDS-DMA writes zeros for it (scripts/probes/buf_lds_oob.hip) -- no zero page, 32-bit offsets.
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Mrows * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.Nimg * a.H * a.W * a.C * 2), 0x00020000);
  int a_off[AI];   // byte offset of this lane's chunk in column 0 of its weight row
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + (wave * AI + i) * RPI + rsub;
    a_off[i] = m < a.Mrows ? (m * K + lchunk * 8) * 2 : OOB;
  }
  int b_off[BI];   // byte offset of tap (0,0), channel chunk 0 (may be negative: only used with a valid tap)
  uint32_t b_mask[BI];
  {
    // pixel coordinates: one division for the first row of this lane, the other rows (RPI pixels
    // further each) by carry; tap validity as the outer product of 3 row bits and 3 column bits,
    // both in closed form (filters are at most 3 x 3)
    constexpr int PSTEP = BSPLIT ? NW * RPI : RPI;   // pixels between this lane's rows of instructions i and i + 1
    int p = p0 + (BSPLIT ? wave : wave * BI) * RPI + rsub;
    const int HoWo = a.Ho * a.Wo;
    const int pc = p < P ? p : (P > 0 ? P - 1 : 0);
    int n = pc / HoWo;
    int rem = pc - n * HoWo;
    int ho = rem / a.Wo;
    int wo = rem - ho * a.Wo;
    const uint32_t rbits = (1u << a.R) - 1u, sbits = (1u << a.S) - 1u;
    const bool fwd = a.mode == 0, s2 = a.stride == 2;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      // tap r reads row  fwd: bh + r | input gradient, stride 1: bh - r | stride 2: bh - (r >> 1), only
      // when (ho + pad - r) is even
      const int th = ho + a.pad, tw = wo + a.pad;
      const int bh = fwd ? ho * a.stride - a.pad : (s2 ? th >> 1 : th);
      const int bw = fwd ? wo * a.stride - a.pad : (s2 ? tw >> 1 : tw);
      uint32_t vh = 0, vw = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int dh = fwd ? r : (s2 ? -(r >> 1) : -r);
        const bool par_h = fwd || !s2 || !((th - r) & 1);
        const bool par_w = fwd || !s2 || !((tw - r) & 1);
        vh |= (par_h && (unsigned)(bh + dh) < (unsigned)a.H) ? (1u << r) : 0u;
        vw |= (par_w && (unsigned)(bw + dh) < (unsigned)a.W) ? (1u << r) : 0u;
      }
      vh &= rbits;
      vw &= sbits;
      const uint32_t mask = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << a.S : 0u) | ((vh & 4u) ? vw << (2 * a.S) : 0u);
      b_mask[i] = p < P ? mask : 0u;
      b_off[i] = ((((n * a.H + bh) * a.W + bw) * a.C) + lchunk * 8) * 2;
      // advance PSTEP pixels (a few row carries on the feature maps, PSTEP of them on the 1 x 1 "image" of the FC)
      p += PSTEP;
      wo += PSTEP;
      while (wo >= a.Wo) {
        wo -= a.Wo;
        if (++ho >= a.Ho) {
          ho = 0;
          ++n;
        }
      }
    }
  }

  // k-tile cursor of the NEXT tile to issue, advanced incrementally (no divisions in the loop).
  // Order: channel chunk outer, filter tap inner -- the R*S taps of one BK-channel chunk read the same
  // 128-byte lines of the gathered tensor (shifted by a pixel / a row), so a workgroup's live footprint
  // is one chunk of its pixels (+ halo) and stays in the CU's L1 / the XCD's L2 across the taps
  // (L2 misses per launch fell 3x on the 256-channel 14x14 layer, rocprofv3 TCC_MISS).
  int is_c0 = (kt0 / ntap) * BK;
  int is_j = kt0 % ntap;                     // index into tap_list
  auto issue = [&](int stage) {
    const int is_tap = (int)((tap_list >> (4 * is_j)) & 15u);
    const int is_r = is_tap / a.S, is_s = is_tap - is_r * a.S;
    int toff;
    if (a.mode == 0) toff = (is_r * a.W + is_s) * a.C;
    else if (a.stride == 1) toff = -(is_r * a.W + is_s) * a.C;
    else toff = -((is_r >> 1) * a.W + (is_s >> 1)) * a.C;
    toff += is_c0;
    const int wtap = a.tap_mask ? (int)((a.tap_w >> (4 * is_tap)) & 15u) : is_tap;
    const int is_k0 = wtap * a.C + is_c0;   // column of the [Mrows][R*S*C] weight matrix
    const uint32_t bit = 1u << is_tap;
    char* st = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(st + (wave * AI + i) * 1024), 16, a_off[i], is_k0 * 2, 0,
                                               0);
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int bq = BSPLIT ? i * NW + wave : wave * BI + i;   // instruction of the pixel tile (wave-uniform)
      if (BSPLIT && bq >= BROWS) continue;
      const int off = (b_mask[i] & bit) ? b_off[i] + toff * 2 : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(st + BM * RSB + bq * 1024), 16, off, 0, 0, 0);
    }
    if (++is_j >= ntap) {
      is_j = 0;
      is_c0 += BK;
    }
  };

  // RED: the [BN pixels][BM channels] tile of the BatchNorm input x that matches this output tile -> LDS stage nk % NST
  // (free during the last k-tile), 16-byte chunks XOR-swizzled with the pixel row on the SOURCE side like the operands.
  // Rows are the output rows of the tile: consecutive pixels, or (parity-class launch) the stride-2 positions.
  auto issue_x = [&]() {
    if constexpr (RED) {
      static_assert(NST >= 2 && !PP && !SWP && BM * BN * 2 <= (BM + BN) * BK * 2, "RED: the x tile takes the stage tile nk would have taken (free during the last k-tile)");
      constexpr int XRB = BM * 2;                 // x row bytes in LDS
      constexpr int XCPR = XRB / 16;              // chunks per row
      constexpr int XRPI = 1024 / XRB;            // rows per DMA instruction
      constexpr int XI = BN / XRPI / NW;          // instructions per wave
      const __amdgpu_buffer_rsrc_t rs_rx = __builtin_amdgcn_make_buffer_rsrc(
          (void*)a.red_x, 0, (int)((size_t)(a.cls ? a.Nimg * a.Hf * a.Wf : P) * a.Mrows * 2), 0x00020000);
      char* st = smem + (nk % NST) * STAGE;
      const int rin = lane / XCPR, pc = lane % XCPR;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int r = (wave * XI + i) * XRPI + rin;             // pixel row of the tile
        const int lc = pc ^ (r & (XCPR - 1));
        const int q = p0 + r;
        int64_t orow = q;
        if (a.cls) {
          const int ph = (a.cls - 1) >> 1, pwc = (a.cls - 1) & 1;
          const int HWc = a.Ho * a.Wo;
          const int qc = q < P ? q : 0;
          const int n = qc / HWc;
          const int rem = qc - n * HWc;
          const int hh = rem / a.Wo;
          orow = ((int64_t)n * a.Hf + 2 * hh + ph) * a.Wf + 2 * (rem - hh * a.Wo) + pwc;
        }
        const bool ok = q < P && m0 + lc * 8 < a.Mrows;
        const int off = ok ? (int)((orow * a.Mrows + m0 + lc * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rx, (lds_void_t*)(st + (wave * XI + i) * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int pre = nk < NST - 1 ? nk : NST - 1;
  for (int s = 0; s < pre; ++s) issue(s);

  if constexpr (SWP) {
    // ---- software-pipelined schedule: the fragments of tile t live in registers while its 32 MFMAs run, and
    // the 16 LDS reads of tile t + 1 are interleaved between those MFMAs (two fragment sets, swapped every
    // tile), so a wave no longer pays "issue reads, wait for LDS, then multiply" per k-tile.  With tile t in
    // registers its LDS stage is free as soon as every wave has read it: the DMA of tile t + 2 goes there,
    // still two stages of LDS.  Per tile: vmcnt(0) (tile t + 1 landed) + barrier, DMA issue of tile t + 2,
    // 2 x (16 MFMAs + 8 reads), lgkmcnt(0).
    static_assert(NW == 4 && BK == 64 && NST == 2 && !PP, "software-pipelined variant: 4 waves, 64-deep tiles, 2 stages");
    constexpr int NDMA = AI + BI;
    bf16x8 fA[2][2][MT], fB[2][2][NT];   // [set][k-step][fragment]
    auto rd_base = [&](int t, int kk) -> uint32_t {
      return lds0 + (uint32_t)((t % NST) * STAGE) + (uint32_t)(r16 * RSB + (((kk * 4 + h) ^ swz<BK>(r16)) << 4));
    };
    const uint32_t offA = (uint32_t)(wm * (BM / WM) * RSB), offB = (uint32_t)(wn * (BN / WN) * RSB);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t rd = rd_base(0, kk);
      lds_read_frags<16 * RSB, 0>(fA[0][kk], rd + offA, std::make_integer_sequence<int, MT>{});
      lds_read_frags<16 * RSB, BM * RSB>(fB[0][kk], rd + offB, std::make_integer_sequence<int, NT>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int BPR = NT / MT;   // B fragments fetched per MFMA row (the A fragment of the row comes with them)
    auto tile = [&]<int PAR>(int it, std::integral_constant<int, PAR>) {
      const bool more = it + 1 < nk;
      if (more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slices of tile it + 1 have landed
        __builtin_amdgcn_s_barrier();                         // ... everybody's; and every wave holds tile it in registers
        if (it + 2 < nk) issue(it % NST);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint32_t rd = rd_base(it + 1, kk);
        static_for<MT>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
            fA[PAR ^ 1][kk][I] = lds_read128_asm<I * 16 * RSB>(rd + offA);
            static_for<BPR>([&](auto jc) {
              constexpr int J = I * BPR + decltype(jc)::value;
              fB[PAR ^ 1][kk][J] = lds_read128_asm<BM * RSB + J * 16 * RSB>(rd + offB);
            });
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[I][j] = mfma16(fA[PAR][kk][I], fB[PAR][kk][j], acc[I][j]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int it = 0; it < nk; it += 2) {
      tile(it, std::integral_constant<int, 0>{});
      if (it + 1 < nk) tile(it + 1, std::integral_constant<int, 1>{});
    }
  } else if constexpr (PP) {
    // ---- ping-pong schedule (8 waves = two groups of one wave per SIMD, three 64-deep stages, one
    // workgroup per CU).  Phases alternate: while one group runs the 32 MFMAs of a k-tile from
    // registers, the other reads its fragments of the next tile from LDS and issues its slice of the
    // DMAs two tiles ahead; one workgroup barrier ends every phase.  Group 0 reads tile t in phase 2t
    // and multiplies in 2t + 1, group 1 one phase later.  Tile t + 2 goes into the stage of tile
    // t - 1, whose last readers (group 1, phase 2t - 1) are behind the barrier; a wave's slice of tile
    // t + 1 has landed (counted vmcnt) before the barrier that opens phase 2t + 2.
    static_assert(NW == 8 && BK == 64 && NST == 3, "ping-pong variant: 8 waves, 64-deep tiles, 3 stages");
    constexpr int NDMA = AI + BI;
    const int grp = wave >> 2;
    bf16x8 fa[2][MT], fb[2][NT];
    auto read_frags = [&](int t) {
      const uint32_t sbase = lds0 + (uint32_t)((t % NST) * STAGE);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint32_t rd = sbase + (uint32_t)(r16 * RSB + (((kk * 4 + h) ^ swz<BK>(r16)) << 4));
        lds_read_frags<16 * RSB, 0>(fa[kk], rd + (uint32_t)(wm * (BM / WM) * RSB), std::make_integer_sequence<int, MT>{});
        lds_read_frags<16 * RSB, BM * RSB>(fb[kk], rd + (uint32_t)(wn * (BN / WN) * RSB),
                                           std::make_integer_sequence<int, NT>{});
      }
    };
    auto multiply = [&]() {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[kk][i], fb[kk][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    };
    // tile 0 complete (the slice of tile 1 may still be in flight)
    if (pre >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool tr_on = a.trace && bx == 7 && by == 0 && (wave & 3) == 0;
    long long* tr = a.trace + (wave >> 2) * 64 * 8;
#define VLSFR_STAMP(t, k)                                                     \
  if (tr_on && (t) < 64) {                                                    \
    const long long c_ = (long long)__builtin_readcyclecounter();             \
    if (lane == 0) __builtin_nontemporal_store(c_, tr + (t) * 8 + (k));       \
  }
    if (grp == 0) {
      for (int t = 0; t < nk; ++t) {
        VLSFR_STAMP(t, 0)
        read_frags(t);                                     // phase 2t
        if (t + 2 < nk) issue((t + 2) % NST);
        VLSFR_STAMP(t, 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VLSFR_STAMP(t, 2)
        __builtin_amdgcn_s_barrier();
        VLSFR_STAMP(t, 3)
        multiply();                                        // phase 2t + 1
        VLSFR_STAMP(t, 4)
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VLSFR_STAMP(t, 5)
        __builtin_amdgcn_s_barrier();
      }
    } else {
      __builtin_amdgcn_s_barrier();                        // phase 0: group 0 reads tile 0
      for (int t = 0; t < nk; ++t) {
        VLSFR_STAMP(t, 0)
        read_frags(t);                                     // phase 2t + 1
        if (t + 2 < nk) issue((t + 2) % NST);
        VLSFR_STAMP(t, 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VLSFR_STAMP(t, 2)
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VLSFR_STAMP(t, 3)
        __builtin_amdgcn_s_barrier();
        VLSFR_STAMP(t, 4)
        multiply();                                        // phase 2t + 2
        VLSFR_STAMP(t, 5)
        if (t + 1 < nk) __builtin_amdgcn_s_barrier();
      }
    }
#undef VLSFR_STAMP
  } else
  for (int it = 0; it < nk; ++it) {
    // tiles issued after tile `it` may stay in flight: min(NST - 2, nk - 1 - it) of them, (AI + BI) DMAs each
    const int later = (nk - 1 - it) < (NST - 2) ? (nk - 1 - it) : (NST - 2);
    if (later >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (AI + BI)) : "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AI + BI)) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(a.dbg & 128)) __builtin_amdgcn_s_barrier();                        // dbg 128 (with 32): no barrier (diagnostic)
    // (issued here, ahead of the fragment reads: between the two MFMA blocks — a cheaper issue slot — the DMA has half a
    // k-tile less to land in and the 128 x 128 layers ran 7 - 10 % slower)
    if (it + NST - 1 < nk && !(a.dbg & 32)) issue((it + NST - 1) % NST);   // dbg 32: no DMA after the prologue (diagnostic)
    if constexpr (RED) {
      if (it == nk - 1 && !(a.dbg & 4)) issue_x();       // the stage of tile nk - 2 is free: the x tile of the epilogue's reduction goes there
    }
    const uint32_t sbase = lds0 + (uint32_t)((it % NST) * STAGE);
    // all fragment reads of the tile are issued up front; the MFMAs of k-step kk start as soon as
    // its (MT + NT) reads have returned (counted lgkmcnt), the later reads land underneath them
    constexpr int KK = BK / 32;
    bf16x8 fa[KK][MT], fb[KK][NT];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const uint32_t rd = sbase + (uint32_t)(r16 * RSB + (((kk * 4 + h) ^ swz<BK>(r16)) << 4));
      lds_read_frags<16 * RSB, 0>(fa[kk], rd + (uint32_t)(wm * (BM / WM) * RSB), std::make_integer_sequence<int, MT>{});
      lds_read_frags<16 * RSB, BM * RSB>(fb[kk], rd + (uint32_t)(wn * (BN / WN) * RSB),
                                         std::make_integer_sequence<int, NT>{});
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((KK - 1 - kk) * (MT + NT)) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[kk][i], fb[kk][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (RED) __builtin_amdgcn_s_barrier();   // every wave's slice of the x tile has landed
  if (a.dbg & 64) {                                  // diagnostic: no epilogue (one dword per lane keeps the loop alive)
    if (acc[0][0][0] == 12345.678f) ((float*)a.y)[tid] = acc[0][0][0];
    return;
  }

  conv_epilogue<BM, BN, WM, WN, MT, NT, NW, RED>(a, acc, m0, p0, P, wm, wn, r16, h, tid, (float*)(smem + (nk % NST) * STAGE),
                                                 smem + (nk % NST) * STAGE);
#endif
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_p8_kernel -- the 256 x 256 / 256 x 224 tile with a FOUR-PHASE-PER-K-TILE schedule (the "8-phase" GEMM structure
// of the CDNA4 playbook, two k-tiles per 8 phases): the loop of conv_igemm_glds_kernel issues every fragment read of a k-tile,
// waits, multiplies, and meets one barrier per tile with the next tile's DMA drained (vmcnt 0) — LDS pipe and matrix pipe take
// turns.  Here a k-tile is cut into four phases of 16 MFMAs per wave (one quadrant of the wave's 64 x 128 output x K = 64):
//   phase:   { ds_read of the quadrant's new operand sub-block ; stage ONE half-tile of a later k-tile (2 LDS-DMA instructions) }
//            s_barrier ; lgkmcnt(0) ; setprio(1) 16 x MFMA setprio(0) ; s_barrier
// and the two groups of four waves (one wave per SIMD each) run ONE BARRIER APART, so on every SIMD one wave multiplies while the
// other reads LDS and issues DMA.  The DMA is never drained inside the loop: one counted vmcnt(6) per k-tile leaves the three
// half-tiles staged last in flight across the barriers.
//   LDS: two k-tile buffers of [A half 0 | A half 1 | B half 0 | B half 1], 128-byte rows, chunks XOR-swizzled with (row & 7).
//   The halves are laid out by PHASE OF FIRST USE, not by tile row: A half 0 holds MFMA row tiles i = 0, 1 of every wave
//   (read in phase 1), half 1 tiles i = 2, 3 (phase 2); B half 0 holds pixel tiles j = 0..3 of every wave (phase 1), half 1
//   tiles j = 4.. (phase 3) — so a half is dead for ALL waves two phases after its read and can be restaged then:
//     tile u:  A0 staged in phase 3 of tile u - 2, B0 and A1 in phase 4 of u - 2 (the phase that reads nothing), B1 in phase 2 of
//     u - 1, nothing in phase 1 (which issues 12 of a tile's 22 - 24 reads: a DMA piece issued among them costs 100 - 185 cycles);
//     phase 4 of tile u - 1 waits vmcnt(6): everything but A0 / B0 / A1 of tile u + 1 has landed, i.e. all of tile u, one phase
//     (two barriers) before its first read.  (A counted wait in EVERY phase with the last four half-tiles left in flight — a whole
//     k-tile of latency for each — measured 76.6 us against 63.6 us for this form on the 256 -> 256 14 x 14 layer.)
//   Quadrants: phase 1 acc[0..1][0..3] (A0 B0), 2 acc[2..3][0..3] (A1 B0), 3 acc[2..3][4..] (A1 B1), 4 acc[0..1][4..] (A0 B1):
//   every accumulator sees k in the order of the default kernel (bit-identical results).
// Full tiles only (Mrows % 256 == 0, P % BN == 0), every tap, no split-K: run_igemm falls back to the other kernels otherwise.
// ------------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(512, 1) void conv_igemm_p8_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BK = 64, NW = 8, WM = 4, WN = 2;
  constexpr int MT = 4, NT = BN / WN / 16, NT0 = 4, NT1 = NT - NT0;
  constexpr int RSB = 128;
  constexpr int BH0 = WN * NT0 * 16, BH1 = WN * NT1 * 16;   // rows of the two B halves
  constexpr int BI1 = BH1 / 8;                              // LDS-DMA instructions of B half 1 (16 or 12)
  constexpr int STAGE = (BM + BN) * RSB;
  constexpr int OOB = (int)0x80000000;
  static_assert(BH0 == 128 && BH0 + BH1 == BN && NT1 >= 1 && NT1 <= 4, "pixel halves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
  const int P = a.Nimg * a.Ho * a.Wo;
  const int K = a.R * a.S * a.C;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd) {
    const int g = xcd_major_id(blockIdx.x, a.gx * a.gy);
    bx = g / a.gy;
    by = g - bx * a.gy;
  }
  const int m0 = by * BM, p0 = bx * BN;
  const int ntap = a.R * a.S;
  const int nk = ntap * (a.C / BK);

  // ---- per-lane gather descriptors: instruction i (0..3) of this wave covers LDS rows 8 (8 (i & 1) + wave) .. + 7 of half i >> 1
  const int rsub = lane >> 3;
  const int lchunk = (lane & 7) ^ rsub;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Mrows * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.Nimg * a.H * a.W * a.C * 2), 0x00020000);
  int a_off[4], b_off[4];
  uint32_t b_mask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // A: half hh = i >> 1 holds, for wave row wm', its MFMA row tiles 2 hh, 2 hh + 1
    const int hh = i >> 1;
    const int rem = ((i & 1) * 8 + wave) * 8 + rsub;
    const int m = m0 + (rem >> 5) * 64 + ((hh << 1) | ((rem >> 4) & 1)) * 16 + (rem & 15);
    a_off[i] = m < a.Mrows ? (m * K + lchunk * 8) * 2 : OOB;
  }
  // the weight halves of tile 0 go out before the pixel coordinates (two divisions per row) are worked out
  {
    char* st0 = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(st0 + (i >> 1) * (128 * RSB) + ((i & 1) * 8 + wave) * 1024), 16,
                                               a_off[i], 0, 0, 0);
  }
  {
    const int HoWo = a.Ho * a.Wo;
    const uint32_t rbits = (1u << a.R) - 1u, sbits = (1u << a.S) - 1u;
    const bool fwd = a.mode == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hh = i >> 1;
      const int rem = ((i & 1) * 8 + wave) * 8 + rsub;          // row inside the half
      // B: half hh holds, for wave column wn', its pixel tiles hh * NT0 + 0 .. NTh - 1
      const int nth16 = (hh ? NT1 : NT0) * 16;
      const bool live = hh == 0 || rem < BH1;
      const int wnp = rem / nth16, rr = rem - wnp * nth16;
      const int p = p0 + wnp * (BN / WN) + hh * NT0 * 16 + rr;
      const int pc = (live && p < P) ? p : 0;
      const int n = pc / HoWo;
      const int r2 = pc - n * HoWo;
      const int ho = r2 / a.Wo, wo = r2 - ho * a.Wo;
      const int bh = fwd ? ho * a.stride - a.pad : ho + a.pad;
      const int bw = fwd ? wo * a.stride - a.pad : wo + a.pad;
      uint32_t vh = 0, vw = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int dh = fwd ? r : -r;
        vh |= ((unsigned)(bh + dh) < (unsigned)a.H) ? (1u << r) : 0u;
        vw |= ((unsigned)(bw + dh) < (unsigned)a.W) ? (1u << r) : 0u;
      }
      vh &= rbits;
      vw &= sbits;
      const uint32_t mask = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << a.S : 0u) | ((vh & 4u) ? vw << (2 * a.S) : 0u);
      b_mask[i] = (live && p < P) ? mask : 0u;
      b_off[i] = ((((n * a.H + bh) * a.W + bw) * a.C) + lchunk * 8) * 2;
    }
  }

  // k-tile cursors (channel chunk outer, tap inner, as in conv_igemm_glds_kernel): c1 = tile t + 1, c2 = tile t + 2
  int c1_j = 0, c1_c = 0, c2_j = 0, c2_c = 0;
  auto adv = [&](int& j, int& c) {
    if (++j >= ntap) {
      j = 0;
      c += BK;
    }
  };
  auto stage_a = [&](int hh, int j, int c, int buf) {
    const int k0 = j * a.C + c;
    char* st = smem + buf * STAGE + hh * (128 * RSB);
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(st + (e * 8 + wave) * 1024), 16, a_off[2 * hh + e], k0 * 2, 0, 0);
  };
  auto stage_b = [&](int hh, int j, int c, int buf) {
    const int r = j / a.S, sx = j - r * a.S;
    const int toff = (a.mode == 0 ? (r * a.W + sx) : -(r * a.W + sx)) * a.C + c;
    const uint32_t bit = 1u << j;
    char* st = smem + buf * STAGE + BM * RSB + hh * (BH0 * RSB);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (hh == 1 && e * 8 + wave >= BI1) continue;      // the short half (224-pixel tile): wave-uniform
      const int off = (b_mask[2 * hh + e] & bit) ? b_off[2 * hh + e] + toff * 2 : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(st + (e * 8 + wave) * 1024), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read bases: byte offset of chunk (4 kk + h) ^ (row & 7) of this lane's row, k-steps 0 and 1
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const uint32_t xk0 = (uint32_t)((h ^ (r16 & 7)) << 4), xk1 = (uint32_t)(((4 + h) ^ (r16 & 7)) << 4);
  const uint32_t rowA = lds0 + (uint32_t)((wm * 32 + r16) * RSB);
  const uint32_t rowB0 = lds0 + (uint32_t)(BM * RSB + (wn * NT0 * 16 + r16) * RSB);
  const uint32_t rowB1 = lds0 + (uint32_t)(BM * RSB + (BH0 + wn * NT1 * 16 + r16) * RSB);

  // ---- prologue: tile 0 complete (its A halves were issued above), A0 / B0 of tile 1 in flight
  stage_b(0, 0, 0, 0);
  stage_b(1, 0, 0, 0);
  adv(c1_j, c1_c);
  c2_j = c1_j;
  c2_c = c1_c;
  adv(c2_j, c2_c);
  if (nk > 1) {
    stage_a(0, c1_j, c1_c, 1);
    stage_b(0, c1_j, c1_c, 1);
    stage_a(1, c1_j, c1_c, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier behind the first

  bf16x8 fa[2][4], fb[2][4];   // [k-step][tile]: A sub-blocks 0 (tiles 0, 1) and 1 (tiles 2, 3); the current B sub-block
  auto quad = [&](auto i0_tag, auto j0_tag, auto nj_tag) {
    constexpr int I0 = decltype(i0_tag)::value, J0 = decltype(j0_tag)::value, NJ = decltype(nj_tag)::value;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[I0 + i][J0 + j] = mfma16(fa[kk][I0 + i], fb[kk][j], acc[I0 + i][J0 + j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using I_ = std::integral_constant<int, 0>;
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const uint32_t sb = (uint32_t)(buf * STAGE);
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    // ---- phase 1: A sub-block 0, B sub-block 0
    const uint32_t rA0 = rowA + sb + xk0, rA1 = rowA + sb + xk1;
    fa[0][0] = lds_read128_asm<0>(rA0);
    fa[0][1] = lds_read128_asm<16 * RSB>(rA0);
    fa[1][0] = lds_read128_asm<0>(rA1);
    fa[1][1] = lds_read128_asm<16 * RSB>(rA1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t b = rowB0 + sb + (kk ? xk1 : xk0);
      fb[kk][0] = lds_read128_asm<0>(b);
      fb[kk][1] = lds_read128_asm<16 * RSB>(b);
      fb[kk][2] = lds_read128_asm<32 * RSB>(b);
      fb[kk][3] = lds_read128_asm<48 * RSB>(b);
    }
    const bool dma = !(a.dbg & 32);                      // diagnostic: no DMA after the prologue
    quad(I_{}, I_{}, std::integral_constant<int, NT0>{});   // (nothing staged in the phase that issues 12 of the tile's reads)
    // ---- phase 2: A sub-block 1
    fa[0][2] = lds_read128_asm<128 * RSB>(rA0);
    fa[0][3] = lds_read128_asm<128 * RSB + 16 * RSB>(rA0);
    fa[1][2] = lds_read128_asm<128 * RSB>(rA1);
    fa[1][3] = lds_read128_asm<128 * RSB + 16 * RSB>(rA1);
    if (more1 && dma) stage_b(1, c1_j, c1_c, buf ^ 1);
    quad(std::integral_constant<int, 2>{}, I_{}, std::integral_constant<int, NT0>{});
    // ---- phase 3: B sub-block 1 (over sub-block 0's registers)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t b = rowB1 + sb + (kk ? xk1 : xk0);
      fb[kk][0] = lds_read128_asm<0>(b);
      if constexpr (NT1 > 1) fb[kk][1] = lds_read128_asm<16 * RSB>(b);
      if constexpr (NT1 > 2) fb[kk][2] = lds_read128_asm<32 * RSB>(b);
      if constexpr (NT1 > 3) fb[kk][3] = lds_read128_asm<48 * RSB>(b);
    }
    if (more2 && dma) stage_a(0, c2_j, c2_c, buf);
    quad(std::integral_constant<int, 2>{}, std::integral_constant<int, NT0>{}, std::integral_constant<int, NT1>{});
    // ---- phase 4: no new operand; the k-tile's one counted wait
    if (more2 && dma) {
      stage_b(0, c2_j, c2_c, buf);
      stage_a(1, c2_j, c2_c, buf);                        // A half 1 of this buffer was last read in phase 2
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all of tile t + 1 has landed; A0 / B0 / A1 of tile t + 2 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    quad(I_{}, std::integral_constant<int, NT0>{}, std::integral_constant<int, NT1>{});
    c1_j = c2_j;
    c1_c = c2_c;
    adv(c2_j, c2_c);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // both groups past their last LDS read: the epilogue reuses the ring
  if (a.dbg & 64) {                             // diagnostic: no epilogue
    if (acc[0][0][0] == 12345.678f) ((float*)a.y)[tid] = acc[0][0][0] + acc[3][NT - 1][3] + acc[1][2][1] + acc[2][5][2];
    return;
  }
  conv_epilogue<BM, BN, WM, WN, MT, NT, NW, false>(a, acc, m0, p0, P, wm, wn, r16, h, tid, (float*)smem, smem);
#endif
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_hp8_kernel -- 3x3 / stride 1 / pad 1 convolutions (forward and input gradient) on the four-phase schedule of
// conv_igemm_p8_kernel, with the PIXEL operand held in LDS as one halo'd patch per 64-channel chunk (the idea of
// conv_igemm_halo_kernel below) instead of nine shifted copies staged tap by tap.
// Why: what a CU can pull through its vector L1 into LDS is 64 B / clk.  A 128 x 128 x 64 k-tile costs 32 KiB = 512 clk of that
// path against 512 clk of MFMA per SIMD — the 4-wave kernel is load-path-bound by construction; the 256 x 224 four-phase tile
// costs 60 KiB per 1 792 MFMA clk.  With the patch, a k-tile (one filter tap of one chunk) stages only its weight tile
// (BM x 64: 32 or 16 KiB) plus 1/9 of the next chunk's patch: 36 KiB (BM = 256, 224 pixels) or 23 KiB (BM = 128, 448 pixels)
// per 1 792 MFMA clk, and the DMA pieces a wave issues per k-tile drop from 7.5 to 4.5 / 3.
//   tile: BM output channels (256: waves 4 x 2; 128: waves 2 x 4) x BN = WN * NT * 16 pixels, every wave 64 x (NT * 16).
//   BM = 128 is what the 128-channel 28 x 28 layers needed: their 128 x 128 tiles are bound by the load path (above) and run
//   3.06 rounds on the 512 resident slots; 128 x 448 tiles are 448 workgroups of the four-phase loop.
//   LDS: [A buffer 0 | A buffer 1 | patch 0 | patch 1 | zero row]; A buffers = [half 0 | half 1] by phase of first use as in
//   conv_igemm_p8_kernel; patch = rows q0 .. q0 + PR - 1 of the flattened (n, h, w) tensor, 128 B (64 channels) per row,
//   16-byte chunks XOR-swizzled with the patch row.  q0 = p0 - (W + 1), or p0 - W when every tile starts a line (BN % W == 0:
//   the corner neighbours of the first / last pixel are then never valid) — that is what lets 448 + 2 x 28 rows fit twice.
//   Tap (r, s) of a chunk reads fragment rows lead + d + pixel, d = +-((r - 1) W + (s - 1)); a lane whose pixel has no such
//   neighbour (image border) is pointed at the zero row (one select per fragment, validity bits per pixel from the prologue).
//   Per k-tile u (tap t of chunk c), per wave:  phase 1 reads A half 0 + pixel tiles 0..3; phase 2 reads A half 1 and stages
//   one piece of chunk c + 1's patch (taps 0..7: 8 x 8 = 64 pieces cover 512 rows); phase 3 reads pixel tiles 4.. and stages A
//   half 0 of tile u + 2; phase 4 stages A half 1 of u + 2 and waits (counted) for everything issued before this tap: all of
//   A(u + 1) — and, at a chunk's last tap, which issues no patch piece, the whole next patch.
// The accumulation order (chunk outer, tap inner, two 32-deep steps per k-tile) is conv_igemm_glds_kernel's: bit-identical results.
// ------------------------------------------------------------------------------------------------
// DIAG (diagnostic instantiations): 1 = clock stamps (vlsfr_conv_trace), 2 = no border selects (conv_dbg 256: wrong at image borders, timing only)
template <int BM, int NT, int DIAG = 0>
__global__ __launch_bounds__(512, 1) void conv_igemm_hp8_kernel(ConvArgs a, int PR, int lead) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool TRACE = DIAG == 1;
  constexpr int BK = 64, NW = 8, WM = BM / 64, WN = NW / WM;
  constexpr int MT = 4, NT0 = 4, NT1 = NT - NT0;
  constexpr int BN = WN * NT * 16;
  constexpr int RSB = 128;
  constexpr int AH = BM / 2;                 // rows of an A half
  constexpr int PA = AH / 64;                // LDS-DMA pieces per wave and A half (8 rows each, 8 waves)
  constexpr int ASTAGE = BM * RSB;
  constexpr int OOB = (int)0x80000000;
  static_assert((BM == 256 || BM == 128) && NT1 >= 1 && NT1 <= 4, "tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PATCH = PR * RSB;
  const int NPI = PR >> 3;                    // DMA pieces per patch (<= 64)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave / WN, wn = wave % WN, grp = wave >> 2;
  const int P = a.Nimg * a.H * a.W;           // stride 1: output pixels = input pixels
  const int K = 9 * a.C;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd) {
    const int g = xcd_major_id(blockIdx.x, a.gx * a.gy);
    bx = g / a.gy;
    by = g - bx * a.gy;
  }
  const int m0 = by * BM, p0 = bx * BN;
  const int nchunk = a.C / BK;
  const int nk = 9 * nchunk;
  const bool fwd = a.mode == 0;

  const int rsub = lane >> 3;
  const int lchunk = (lane & 7) ^ rsub;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Mrows * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)P * a.C * 2), 0x00020000);
  // A: piece e of half hh covers rows 8 (8 e + wave) .. + 7 of that half; half hh holds, per wave row, MFMA row tiles 2 hh, 2 hh + 1
  int a_off[2 * PA];
#pragma unroll
  for (int i = 0; i < 2 * PA; ++i) {
    const int hh = i / PA, e = i % PA;
    const int rem = (e * 8 + wave) * 8 + rsub;
    const int m = m0 + (rem >> 5) * 64 + ((hh << 1) | ((rem >> 4) & 1)) * 16 + (rem & 15);
    a_off[i] = m < a.Mrows ? (m * K + lchunk * 8) * 2 : OOB;
  }
  auto stage_a = [&](int hh, int k0, int buf) {
    char* st = smem + buf * ASTAGE + hh * (AH * RSB);
#pragma unroll
    for (int e = 0; e < PA; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(st + (e * 8 + wave) * 1024), 16, a_off[hh * PA + e], k0 * 2, 0, 0);
  };
  // the weight tile of k-tile 0 goes out before anything else is worked out
  stage_a(0, 0, 0);
  stage_a(1, 0, 0);
  // patch: piece x covers rows 8 x .. 8 x + 7; this lane's row of piece 0 and its byte offset in chunk 0
  char* const sP = smem + 2 * ASTAGE;
  const int q0 = p0 - lead;
  const int rowbytes = a.C * 2;
  const int pq = q0 + rsub;                                   // flattened pixel of this lane's row of piece 0 (may be negative)
  const int pq_off = pq * rowbytes + lchunk * 16;            // (only used when the row is valid)
  auto stage_p = [&](int x, int cbyte, int pbuf) {           // x wave-uniform, < NPI
    const int q = pq + x * 8;
    const int off = (unsigned)q < (unsigned)P ? pq_off + x * 8 * rowbytes + cbyte : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(sP + pbuf * PATCH + x * 1024), 16, off, 0, 0, 0);
  };
  for (int x = wave; x < NPI; x += NW) stage_p(x, 0, 0);
  if (tid < 32) ((float*)(sP + 2 * PATCH))[tid] = 0.f;        // the zero row (visible after the first barrier)
  int nprol = 0;                                               // pieces of k-tile 1 in flight behind the prologue's wait
  if (nk > 1) {
    const int k1 = nchunk > 0 ? a.C : 0;                       // k-tile 1 = tap 1 of chunk 0 (nk >= 9 always)
    stage_a(0, k1, 1);
    stage_a(1, k1, 1);
    nprol = 2 * PA;
  }

  // ---- validity bits (bit tap = 3 r + s) of the NT pixels whose fragments this lane reads: one division pair, then carries
  uint32_t fmask[NT];
  {
    const int HW = a.H * a.W;
    int p = p0 + wn * (NT * 16) + r16;
    const int pc = p < P ? p : (P > 0 ? P - 1 : 0);
    int n = pc / HW;
    const int rem = pc - n * HW;
    int ho = rem / a.W;
    int wo = rem - ho * a.W;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      uint32_t vh = 0, vw = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int dd = fwd ? r - 1 : 1 - r;
        vh |= ((unsigned)(ho + dd) < (unsigned)a.H) ? (1u << r) : 0u;
        vw |= ((unsigned)(wo + dd) < (unsigned)a.W) ? (1u << r) : 0u;
      }
      const uint32_t mask = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << 3 : 0u) | ((vh & 4u) ? vw << 6 : 0u);
      fmask[j] = p < P ? mask : 0u;
      p += 16;
      wo += 16;
      while (wo >= a.W) {
        wo -= a.W;
        if (++ho >= a.H) ho = 0;     // (the image index is not needed: validity depends on (ho, wo) only)
      }
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const uint32_t xk0 = (uint32_t)((h ^ (r16 & 7)) << 4), xk1 = (uint32_t)(((4 + h) ^ (r16 & 7)) << 4);
  const uint32_t rowA = lds0 + (uint32_t)((wm * 32 + r16) * RSB);
  const uint32_t ldsP = lds0 + (uint32_t)(2 * ASTAGE);
  const uint32_t zaddr = ldsP + (uint32_t)(2 * PATCH);
  const int rowb0 = lead + wn * (NT * 16) + r16;               // patch row of this lane's pixel of pixel tile 0, tap shift 0

  if (nprol) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier behind the first

  bf16x8 fa[2][4], fb[2][4];
  // TRACE (diagnostic instantiation, vlsfr_conv_trace): clock stamps of waves 0 and 4 of one workgroup, 16 per k-tile:
  // per phase [R section starts, R section issued (and its reads returned: the stamp itself waits on lgkmcnt), barrier passed,
  // MFMAs issued]; layout [2 groups][64 k-tiles][16]
  int tr_u = 0, tr_ph = 0;
  const bool tr_on = TRACE && a.trace && bx == 7 && by == 0 && (wave & 3) == 0;
  long long* const trp = TRACE && a.trace ? a.trace + (wave >> 2) * 64 * 16 : nullptr;
  auto stamp = [&](int k) {
    if constexpr (TRACE) {
      if (tr_on && tr_u < 64) {
        const long long c_ = (long long)__builtin_readcyclecounter();
        if (lane == 0) __builtin_nontemporal_store(c_, trp + tr_u * 16 + tr_ph * 4 + k);
      }
    }
  };
  auto quad = [&](auto i0_tag, auto j0_tag, auto nj_tag) {
    constexpr int I0 = decltype(i0_tag)::value, J0 = decltype(j0_tag)::value, NJ = decltype(nj_tag)::value;
    stamp(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[I0 + i][J0 + j] = mfma16(fa[kk][I0 + i], fb[kk][j], acc[I0 + i][J0 + j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    stamp(3);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);   // the next phase's address arithmetic stays behind the barrier (beside the partner's MFMAs)
    if constexpr (TRACE) {
      if (++tr_ph == 4) {
        tr_ph = 0;
        ++tr_u;
      }
      stamp(0);
    }
  };
  // pixel tiles J0 .. J0 + NJ - 1 of this tap into fb: patch rows rowb + 16 j, or the zero row where the neighbour does not exist
  auto read_b = [&](auto j0_tag, auto nj_tag, uint32_t pbase, uint32_t xb0, uint32_t xb1, uint32_t bit) {
    constexpr int J0 = decltype(j0_tag)::value, NJ = decltype(nj_tag)::value;
    static_for<NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const uint32_t sel = (DIAG == 2 || (fmask[J0 + j] & bit)) ? pbase + (uint32_t)((J0 + j) * 16 * RSB) : zaddr;
      fb[0][j] = lds_read128_asm<0>(sel + xb0);
      fb[1][j] = lds_read128_asm<0>(sel + xb1);
    });
  };
  using I_ = std::integral_constant<int, 0>;
  const bool dma = !(a.dbg & 32);                       // diagnostic: no DMA after the prologue
  stamp(0);
  int t = 0, tr = 0, ts = 0, c = 0;                     // tap (= 3 tr + ts) and chunk of k-tile u
  for (int u = 0; u < nk; ++u) {
    const int buf = u & 1;
    const uint32_t sb = (uint32_t)(buf * ASTAGE);
    const bool more2 = u + 2 < nk;
    // k-tile u + 2: tap t + 2 of this chunk, or tap t - 7 of the next
    const int t2 = t + 2 >= 9 ? t - 7 : t + 2;
    const int k2 = t2 * a.C + (t + 2 >= 9 ? c + 1 : c) * BK;
    const int d = fwd ? (tr - 1) * a.W + (ts - 1) : (1 - tr) * a.W + (1 - ts);
    const int rowb = rowb0 + d;
    const uint32_t key = (uint32_t)(rowb & 7);
    const uint32_t xb0 = ((uint32_t)h ^ key) << 4, xb1 = ((uint32_t)(4 + h) ^ key) << 4;
    const uint32_t pbase = ldsP + (uint32_t)((c & 1) * PATCH) + (uint32_t)(rowb * RSB);
    const uint32_t bit = 1u << t;
    // ---- phase 1: A half 0, pixel tiles 0..3
    const uint32_t rA0 = rowA + sb + xk0, rA1 = rowA + sb + xk1;
    fa[0][0] = lds_read128_asm<0>(rA0);
    fa[0][1] = lds_read128_asm<16 * RSB>(rA0);
    fa[1][0] = lds_read128_asm<0>(rA1);
    fa[1][1] = lds_read128_asm<16 * RSB>(rA1);
    read_b(I_{}, std::integral_constant<int, NT0>{}, pbase, xb0, xb1, bit);
    quad(I_{}, I_{}, std::integral_constant<int, NT0>{});
    // ---- phase 2: A half 1; one piece of the next chunk's patch
    fa[0][2] = lds_read128_asm<AH * RSB>(rA0);
    fa[0][3] = lds_read128_asm<AH * RSB + 16 * RSB>(rA0);
    fa[1][2] = lds_read128_asm<AH * RSB>(rA1);
    fa[1][3] = lds_read128_asm<AH * RSB + 16 * RSB>(rA1);
    const int px = t * NW + wave;
    const bool piece = dma && t < 8 && c + 1 < nchunk && px < NPI;   // wave-uniform
    if (piece) stage_p(px, (c + 1) * (BK * 2), (c + 1) & 1);
    quad(std::integral_constant<int, 2>{}, I_{}, std::integral_constant<int, NT0>{});
    // ---- phase 3: pixel tiles 4.. (over tiles 0..3's registers); A half 0 of k-tile u + 2 (this buffer's half 0 was last read in phase 1)
    read_b(std::integral_constant<int, NT0>{}, std::integral_constant<int, NT1>{}, pbase, xb0, xb1, bit);
    if (more2 && dma) stage_a(0, k2, buf);
    quad(std::integral_constant<int, 2>{}, std::integral_constant<int, NT0>{}, std::integral_constant<int, NT1>{});
    // ---- phase 4: A half 1 of k-tile u + 2; the k-tile's one counted wait: everything issued before this tap has landed
    if (more2 && dma) {
      stage_a(1, k2, buf);
      if (piece) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PA + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    quad(I_{}, std::integral_constant<int, NT0>{}, std::integral_constant<int, NT1>{});
    if (++ts == 3) {
      ts = 0;
      ++tr;
    }
    if (++t == 9) {
      t = tr = ts = 0;
      ++c;
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // both groups past their last LDS read: the epilogue reuses the ring
  if (a.dbg & 64) {                             // diagnostic: no epilogue
    if (acc[0][0][0] == 12345.678f) ((float*)a.y)[tid] = acc[0][0][0] + acc[3][NT - 1][3] + acc[1][2][1] + acc[2][5][2];
    return;
  }
  conv_epilogue<BM, BN, WM, WN, MT, NT, NW, false>(a, acc, m0, p0, P, wm, wn, r16, h, tid, (float*)smem, smem);
#endif
}

// slot of the input transform's micro-operations that gap n of step `step` (0 / 1) of a k-tile carries, or -1 (conv_igemm_hw4_kernel<XF>)
constexpr int xf_seq(int step, int n) {
  return step == 0 ? (n <= 14 ? n : (n >= 37 && n <= 55 ? 15 + (n - 37) : -1)) : (n >= 27 && n <= 52 ? 34 + (n - 27) : -1);
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_hw4_kernel -- the same halo-patch operand path with ONE WAVE PER SIMD and a single software-pipelined instruction
// stream instead of two waves per SIMD trading places at barriers.  What the stamps of conv_igemm_hp8_kernel showed
// (scripts/hp8_trace.py, profiles/r04_hp8_trace.txt): its LDS-read / select sections take 450 - 600 cycles beside MFMA sections of
// 192 - 256, and even the bare four-phase loop pays ~175 cycles per barrier interval, eight times per k-tile — the matrix pipes
// are busy about half the time.  Here:
//   * 4 waves (256 threads, one workgroup per CU, the whole 512-register file per wave): every wave owns 128 channels x (NT * 16)
//     pixels = 8 x NT accumulator tiles — 15 fragment reads per 56 MFMAs instead of 11 per 28, and half as many waves repeat the
//     per-lane border selects and the LDS-DMA address work.
//   * a k-tile (one filter tap of one 64-channel chunk) is two steps of 56 MFMAs (k = 0..31, 32..63).  The fragments of step
//     s + 1 are read from LDS BETWEEN the MFMAs of step s into the other half of a double register buffer (one ds_read, or two
//     VALU, or one LDS-DMA piece per MFMA gap — each gap has 8 issue cycles the MFMA does not hold), the border selects of the
//     next tap are computed in the gaps after them, and the LDS-DMA of the weight tile of k-tile u + 2 and of this tap's piece of
//     the next chunk's patch are issued in the first gaps of the second step.  The order is pinned gap by gap
//     (__builtin_amdgcn_sched_barrier between them): nothing is left to the scheduler's idea of a good interleave.
//   * ONE barrier per k-tile (head of the second step): behind it every wave has finished reading weight buffer u & 1's first
//     half-step... precisely: the reads of step 2u + 1 were issued during step 2u and have returned (lgkmcnt(0)), so buffer u & 1
//     is dead for everyone and takes tile u + 2; the wave's own DMA of the previous k-tile is drained (vmcnt(0): it had a whole
//     k-tile, ~1 800 cycles, to land) in front of the barrier, so behind it tile u + 1 — read from step 2u + 1 on — is complete.
//   LDS: [A buffer 0 | A buffer 1 | patch 0 | patch 1 | zero row], weight rows in natural order (wave row wm reads rows
//   128 wm + 16 i + r16), everything else as in conv_igemm_hp8_kernel.  Same accumulation order: bit-identical results.
// ------------------------------------------------------------------------------------------------
// DIAG = 1 (diagnostic instantiation, vlsfr_conv_trace): wave 0 of one workgroup stamps the shader clock at the head of both steps of
// every k-tile ([64][2]) and the 100 MHz real-time clock around the loop ([128], [129]): cycles per step and the clock the chip holds.
// RED: an input-gradient launch that also accumulates the reduction of the BatchNorm backward reading its output (ConvArgs::red_x,
// conv_epilogue): the [BN pixels][BM channels] tile of that layer's input is fetched by LDS-DMA into the (now free) LDS as soon as
// the loop ends and lands under the output stores.
// XF (forward launches, vlsfr_conv2d_fwd_bnin): the input tensor is the RAW output of the layer in front and the BatchNorm (XF = 1) or
// BatchNorm + PReLU (XF = 2) that stands between the two layers is applied to the patch IN LDS, once per element (the nine taps of
// a chunk read the transformed patch) — what bn_apply did in a pass of its own over HBM.  The wave that fetched a 1 KiB piece owns it:
// a lane's 16 bytes are always the same 8 channels of the chunk (lchunk), so scale / shift / slope sit in registers per chunk; the
// piece fetched in tap t is read back from LDS in tap t + 1 (behind the barrier that drains the DMA), transformed in the MFMA
// gaps of tap t + 2 (micro-operations of 2 - 3 VALU, one per gap) and written back in place — and, for passes that keep
// activations, stored to a_out from the registers (rows of the tile proper only: every row of the tensor is written once).  Pieces
// are fetched in taps 0..5 only, so the last one is back in LDS before the barrier of tap 8, behind which the new patch is first read.
// VLSFR_HW4_REGS (build option, default 224; 0 = no cap): cap of the kernel's architectural VGPRs; the 224 accumulator registers come on
// top.  224 -> 448 of a SIMD's 512 registers, which leaves 64 for one wave of another stream's kernel beside it: the plain and PReLU
// bn_apply kernels of the OTHER chain of a pass (56 / 58 registers) then run on the CUs a convolution occupies instead of only on
// the 32 it leaves idle (same-box A/B, profiles/r04_hw4_register_cap_ab.txt: step -0.2 ms, the kernel alone -0.6 %; the epilogue is
// written to fit — phases, PLAIN, laundered lane coordinates — so that the cap costs no scratch).
#ifndef VLSFR_HW4_REGS
#define VLSFR_HW4_REGS 224
#endif
#if VLSFR_HW4_REGS > 0
#define HW4_REG_ATTR __attribute__((amdgpu_waves_per_eu(1, 2), amdgpu_num_vgpr(VLSFR_HW4_REGS)))
#else
#define HW4_REG_ATTR
#endif
template <int BM, int NT, int PPW, int DIAG = 0, bool RED = false, int XF = 0>
__global__ __launch_bounds__(256, 1) HW4_REG_ATTR void conv_igemm_hw4_kernel(ConvArgs a, int PR, int lead) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool TRACE = DIAG == 1;
  // BM = 64 (the 64-channel layers): one wave row of 64 channels, NT = 14 pixel tiles per wave — the same 56 accumulator tiles and
  // MFMA gaps per step; those layers have ONE 64-channel chunk, so their (large: 896 + 2 W rows) patch is fetched once, in the prologue
  // NT = 13 (the 208-pixel tiles): 64-row waves, 4 x 13 = 52 accumulator tiles — 256 x 208 tiles (four wave rows, one pixel range)
  constexpr int WROWS = (BM >= 128 && NT != 13) ? 128 : 64;
  constexpr int BK = 64, NW = 4, WM = BM / WROWS, WN = NW / WM, MT = WROWS / 16;
  constexpr int BN = WN * NT * 16;
  constexpr int RSB = 128;
  constexpr int ASTAGE = BM * RSB;
  constexpr int PA = BM / 8 / NW;            // LDS-DMA pieces of a weight tile per wave (8 rows each): 8 or 4
  constexpr int NMF = MT * NT;               // MFMAs per step
  constexpr int OOB = (int)0x80000000;
  static_assert((BM == 256 || BM == 128 || BM == 64) && (MT * NT == 56 || MT * NT == 52) && PPW >= 1 && PPW <= 3 && !(XF && RED) && !(XF && BM == 64), "tile");
  constexpr int PT = XF ? 6 : 8;             // taps of a chunk in which pieces of the next patch are fetched
  constexpr int XMP = XF == 2 ? 22 : XF == 1 ? 14 : 0;   // micro-operations of the transform of one piece
  static_assert(XMP * PPW <= 120 && (NMF == 56 || XF == 0), "the transform of a tap's pieces fits its slots (xf_seq is laid out for 56 gaps per step)");
  constexpr int DSTEP = BM == 256 ? 5 : (PPW == 3 ? 7 : 8);   // MFMA gaps between two LDS-DMA pieces of the second step
  static_assert((PA + PPW - 1) * DSTEP < NMF && (PA + PPW - 1) * DSTEP + 6 < NMF && PA + PPW + MT + NT + 5 <= NMF && MT + NT + 4 + 2 * NT + 3 * PPW <= NMF, "the side operations of a step fit its MFMA gaps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PATCH = PR * RSB;
  const int NPI = PR >> 3;                    // DMA pieces per patch (<= PT * NW * PPW where a next patch is fetched in the loop)
  const int NPATCH = a.C > BK ? 2 : 1;        // patch buffers: double-buffered across chunks, or the one patch of a 64-channel layer

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int P = a.Nimg * a.H * a.W;           // stride 1: output pixels = input pixels
  const int K = 9 * a.C;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd) {
    const int g = xcd_major_id(blockIdx.x + a.tile0, a.gx * a.gy);
    bx = g / a.gy;
    by = g - bx * a.gy;
  }
  const int m0 = by * BM, p0 = bx * BN;
  const int nchunk = a.C / BK;
  const int nk = 9 * nchunk;
  const bool fwd = a.mode == 0;
  if constexpr (TRACE) {   // [130]: kernel entry, [131] (below): behind the epilogue — real-time clock, 100 MHz
    if (a.trace && bx == 7 && by == 0 && tid == 0) __builtin_nontemporal_store((long long)__builtin_amdgcn_s_memrealtime(), a.trace + 130);
  }

  const int rsub = lane >> 3;
  const int lchunk = (lane & 7) ^ rsub;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Mrows * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)P * a.C * 2), 0x00020000);
  // weight tile: piece x = e * NW + wave covers rows 8 x .. 8 x + 7 (Mrows % BM == 0: run_igemm); one per-lane offset, the rest scalar
  const int a_off0 = ((m0 + 8 * wave + rsub) * K + lchunk * 8) * 2;
  const int a_estep = NW * 8 * K * 2;
  auto stage_a1 = [&](int e, int k0, int buf, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + buf * ASTAGE + (e * NW + wave) * 1024), 16, voff,
                                             k0 * 2 + e * a_estep, 0, 0);
  };
  for (int e = 0; e < PA; ++e) stage_a1(e, 0, 0, a_off0);
  char* const sP = smem + 2 * ASTAGE;
  const int q0 = p0 - lead;
  const int rowbytes = a.C * 2;
  const int pq = q0 + rsub;
  const int pq_off = pq * rowbytes + lchunk * 16;
  // XF: transform constants of this lane's 8 channels of a chunk, and the transform of one 16-byte piece
  float xs[8], xh[8], xl[8];
  auto xf_load = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (XF != 0) {
      // (inline asm loads: the compiler's own waitcnt insertion would put a vmcnt(0) in front of the first use in EVERY k-tile —
      // a drain of the weight DMA issued half a k-tile earlier; the loads are issued in tap 0 and first used in tap 2, behind two of
      // the loop's own counted waits, and the prologue waits explicitly)
      const int ch = (chunk < nchunk ? chunk : nchunk - 1) * BK + lchunk * 8;
      auto ld4 = [](const float* p) __attribute__((always_inline)) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        return v;
      };
      const f32x4 s0 = ld4(a.xf_scale + ch), s1 = ld4(a.xf_scale + ch + 4);
      const f32x4 h0 = ld4(a.xf_shift + ch), h1 = ld4(a.xf_shift + ch + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xs[e] = s0[e];
        xs[4 + e] = s1[e];
        xh[e] = h0[e];
        xh[4 + e] = h1[e];
      }
      if constexpr (XF == 2) {
        const f32x4 l0 = ld4(a.xf_slope + ch), l1 = ld4(a.xf_slope + ch + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xl[e] = l0[e];
          xl[4 + e] = l1[e];
        }
      }
    }
  };
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  // a_out descriptor: rows outside the tile proper (halo), outside the tensor, and every row when no a_out is wanted get an
  // out-of-range offset — the store is dropped, no branch
  const __amdgpu_buffer_rsrc_t rs_ao =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.xf_out, 0, a.xf_out ? (int)((size_t)P * a.C * 2) : 0, 0x00020000);
  u32x4 xraw[PPW];          // pieces read back from LDS, waiting for / in their transform
  float xf8[8];             // the piece in work
  uint32_t xw_addr[PPW];    // where each goes back to (LDS), and its first pixel row relative to the patch (or far out of range)
  int xw_q8[PPW];
  int xw_soff[PPW];         // ... and its byte offset in x / a_out relative to this lane's pq_off
  // micro-operation m of the transform of piece e (compile-time m): 4 x unpack, 4 x multiply-add, (8 x PReLU), 4 x pack, write, store
  auto xf_micro = [&](auto e_tag, auto m_tag, int chunk) __attribute__((always_inline)) {
    if constexpr (XF != 0) {
      constexpr int e = decltype(e_tag)::value, m = decltype(m_tag)::value;
      constexpr int M_FMA = 4, M_PRELU = 8, M_PACK = XF == 2 ? 16 : 8, M_WRITE = M_PACK + 4, M_STORE = M_WRITE + 1;
      if constexpr (m < M_FMA) {
        const uint32_t w = xraw[e][m];
        xf8[2 * m] = __uint_as_float(w << 16);
        xf8[2 * m + 1] = __uint_as_float(w & 0xffff0000u);
        asm volatile("" : "+v"(xf8[2 * m]), "+v"(xf8[2 * m + 1]));
      } else if constexpr (m < M_PRELU) {
        constexpr int q = 2 * (m - M_FMA);
        xf8[q] = xf8[q] * xs[q] + xh[q];                 // (the expression of bn_apply_kernel)
        xf8[q + 1] = xf8[q + 1] * xs[q + 1] + xh[q + 1];
        asm volatile("" : "+v"(xf8[q]), "+v"(xf8[q + 1]));
      } else if constexpr (m < M_PACK) {                 // XF == 2 only
        constexpr int q = m - M_PRELU;
        const float z = xf8[q];
        xf8[q] = z > 0.f ? z : z * xl[q];
        asm volatile("" : "+v"(xf8[q]));
      } else if constexpr (m < M_WRITE) {
        constexpr int q = m - M_PACK;
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 o;
        o[0] = (__bf16)xf8[2 * q];
        o[1] = (__bf16)xf8[2 * q + 1];
        xraw[e][q] = __builtin_bit_cast(uint32_t, o);
        asm volatile("" : "+v"(xraw[e][q]));
      } else if constexpr (m == M_WRITE) {
        asm volatile("ds_write_b128 %0, %1" ::"v"(xw_addr[e] + (uint32_t)(lane * 16)), "v"(xraw[e]) : "memory");
      } else if constexpr (m == M_STORE) {
        const int q = pq + xw_q8[e];                                         // flattened pixel of this lane's row
        const bool ok = (unsigned)(q - p0) < (unsigned)BN && q < P;
        // (a_out has the layout of x: the byte offset is the one the piece was fetched from)
        const int off = (pq_off + xw_soff[e]) | (ok ? 0 : OOB);
        __builtin_amdgcn_raw_buffer_store_b128(xraw[e], rs_ao, off, 0, 0);
      }
    }
  };
  // The transform's micro-operations go to the gaps that carry little else: gaps 0..14 (one fragment read each) and 37..55 (empty) of
  // a k-tile's first step, gaps 27..52 of its second (behind the DMA pieces and the fragment reads, in front of the read-back of
  // the next pieces) — 60 slots in time order, XPER micro-operations each.
  constexpr int XSLOTS = 60, XTOT = XMP * PPW, XPER = XF ? (XTOT + XSLOTS - 1) / XSLOTS : 1;
  auto xf_slot = [&](auto s_tag, int chunk) __attribute__((always_inline)) {
    constexpr int sidx = decltype(s_tag)::value;
    if constexpr (XF != 0 && sidx >= 0) {
      static_for<XPER>([&](auto kc) {
        constexpr int m = sidx * XPER + decltype(kc)::value;
        if constexpr (m < XTOT) xf_micro(std::integral_constant<int, m / XMP>{}, std::integral_constant<int, m % XMP>{}, chunk);
      });
    }
  };
  // piece x (wave-uniform) of a patch; due == false: nothing to fetch — the instruction still issues (no branch in the stream), with an
  // out-of-range offset (no memory traffic, zeros) into the dump area behind the zero row
  auto stage_p = [&](int x, int cbyte, int pbuf, bool due) {
    const int q = pq + x * 8;
    const int off = (due && (unsigned)q < (unsigned)P) ? pq_off + x * 8 * rowbytes + cbyte : OOB;
    char* dst = due ? sP + pbuf * PATCH + x * 1024 : sP + NPATCH * PATCH + 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)dst, 16, off, 0, 0, 0);
  };
  for (int x = wave; x < NPI; x += NW) stage_p(x, 0, 0, true);
  if (tid < 32) ((float*)(sP + NPATCH * PATCH))[tid] = 0.f;   // the zero row
  if (nk > 1)
    for (int e = 0; e < PA; ++e) stage_a1(e, a.C, 1, a_off0);  // k-tile 1 = tap 1 of chunk 0

  // ---- validity bits (bit tap = 3 r + s) of the NT pixels whose fragments this lane reads
  uint32_t fmask[NT];
  {
    const int HW = a.H * a.W;
    int p = p0 + wn * (NT * 16) + r16;
    const int pc = p < P ? p : (P > 0 ? P - 1 : 0);
    const int n = pc / HW;
    const int rem = pc - n * HW;
    int ho = rem / a.W;
    int wo = rem - ho * a.W;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      uint32_t vh = 0, vw = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int dd = fwd ? r - 1 : 1 - r;
        vh |= ((unsigned)(ho + dd) < (unsigned)a.H) ? (1u << r) : 0u;
        vw |= ((unsigned)(wo + dd) < (unsigned)a.W) ? (1u << r) : 0u;
      }
      const uint32_t mask = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << 3 : 0u) | ((vh & 4u) ? vw << 6 : 0u);
      fmask[j] = p < P ? mask : 0u;
      p += 16;
      wo += 16;
      while (wo >= a.W) {
        wo -= a.W;
        if (++ho >= a.H) ho = 0;
      }
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const uint32_t xk0 = (uint32_t)((h ^ (r16 & 7)) << 4);
  const uint32_t rowA = lds0 + (uint32_t)((wm * WROWS + r16) * RSB) + xk0;      // k-step 0; k-step 1 = ^ 0x40 (the rows are 128-byte aligned)
  const uint32_t ldsP = lds0 + (uint32_t)(2 * ASTAGE);
  const uint32_t zaddr = ldsP + (uint32_t)(NPATCH * PATCH);
  const int rowb0 = lead + wn * (NT * 16) + r16;

  // addresses of this lane's NT pixel-tile fragments (k-step 0) for tap (t, tr, ts) of chunk c: the patch row of the neighbour, or
  // the zero row where the image has none — zx + (valid ? patch address - zx : 0) with the validity bit spread to a mask by one
  // signed bit-field extract: four VALU per fragment, no VCC round trip.  The empty asm statements pin each half to the MFMA gap
  // it is written in (otherwise the compiler sinks all of it to the first use, into one gap).
  uint32_t addrB[NT];
  const int dsgn = fwd ? 1 : -1;
  uint32_t nb_zx = 0;
  int nb_d0 = 0, nb_t = 0, nb_m = 0;
  int nb_ds = 0, nb_ps = 0;                                  // scalar halves of the next tap's common terms
  auto tap_common_s = [&](int tr, int ts, int c) {
    nb_ds = dsgn * ((tr - 1) * a.W + (ts - 1));             // forward: the neighbour (r - 1, s - 1); input gradient: (1 - r, 1 - s)
    nb_ps = (int)ldsP + (c & (NPATCH - 1)) * PATCH - (int)zaddr;   // (one patch buffer: only ever chunk 0, and the last k-tile's idle prefetch stays inside it)
    nb_ds = (std::remove_reference_t<decltype(nb_ds)>)__builtin_amdgcn_readfirstlane((int)nb_ds); nb_ps = (std::remove_reference_t<decltype(nb_ps)>)__builtin_amdgcn_readfirstlane((int)nb_ps); asm volatile("" : "+s"(nb_ds), "+s"(nb_ps));
  };
  auto tap_common_v = [&](int t) {
    const int rowb = rowb0 + nb_ds;
    const uint32_t xb0 = ((uint32_t)h ^ (uint32_t)(rowb & 7)) << 4;
    nb_zx = zaddr + xb0;
    nb_d0 = nb_ps + rowb * RSB;
    nb_t = t;
    asm volatile("" : "+v"(nb_zx), "+v"(nb_d0));
  };
  auto tap_common = [&](int t, int tr, int ts, int c) {
    tap_common_s(tr, ts, c);
    tap_common_v(t);
  };
  auto tap_addr_a = [&](int j) {
    nb_m = __builtin_amdgcn_sbfe((int)fmask[j], nb_t, 1) & (nb_d0 + j * 16 * RSB);
    asm volatile("" : "+v"(nb_m));
  };
  auto tap_addr_b = [&](int j) {
    addrB[j] = nb_zx + (uint32_t)nb_m;
    asm volatile("" : "+v"(addrB[j]));
  };
  tap_common(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    tap_addr_a(j);
    tap_addr_b(j);
  }

  bf16x8 F[2][MT + NT];      // [buffer][8 weight fragments | NT pixel fragments] of one k-step
  auto read_a = [&](auto buf_tag, auto i_tag, uint32_t addr) {
    constexpr int B = decltype(buf_tag)::value, I = decltype(i_tag)::value;
    F[B][I] = lds_read128_asm<I * 16 * RSB>(addr);
  };
  auto read_b = [&](auto buf_tag, auto j_tag, uint32_t addr) {
    constexpr int B = decltype(buf_tag)::value, J = decltype(j_tag)::value;
    F[B][MT + J] = lds_read128_asm<0>(addr);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  // ---- prologue: k-tile 0 and patch 0 complete; fragments of step 0
  // (k-tile 1's weight pieces, issued last, stay in flight: the loop's first barrier drains them; lgkmcnt: the zero row's ds_write)
  if constexpr (XF != 0) {
    // the whole first patch is transformed here, every wave its own pieces (x = wave, wave + 4, ...)
    xf_load(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int x = wave; x < NPI; x += NW) {
      const uint32_t ad = ldsP + (uint32_t)(x * 1024 + lane * 16);
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(xraw[0]) : "v"(ad) : "memory");
      xw_addr[0] = ldsP + (uint32_t)(x * 1024);
      xw_q8[0] = x * 8;
      xw_soff[0] = x * 8 * rowbytes;
      static_for<XMP>([&](auto mc) { xf_micro(std::integral_constant<int, 0>{}, mc, 0); });
    }
    xf_load(1);
  }
  if (nk > 1 && XF == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<MT>([&](auto ic) { read_a(B0{}, ic, rowA); });
  static_for<NT>([&](auto jc) { read_b(B0{}, jc, addrB[decltype(jc)::value]); });

  const bool dma = !(a.dbg & 32);                       // diagnostic: no DMA after the prologue
  const bool tr_on = TRACE && a.trace && bx == 7 && by == 0 && wave == 0;
  if constexpr (TRACE) {
    if (tr_on && lane == 0) __builtin_nontemporal_store((long long)__builtin_amdgcn_s_memrealtime(), a.trace + 128);
  }
  // Loop-carried scalars, advanced INSIDE the MFMA gaps (a wave alone on its SIMD issues in order: thirty scalar instructions at the
  // loop head are thirty instructions during which the matrix pipe has nothing queued; the empty asm statements pin a value to the
  // gap it is computed in): tap / chunk of this k-tile (t, c) and of the next (tn, trn, tsn, cn), this k-tile's weight buffer (sb);
  // for the second step: byte offset of weight piece 0 of k-tile u + 2 (k2s), its per-lane offset (a_voff: out of range when there is
  // no such tile), destination and byte offset of this tap's piece(s) of the next patch.
  int t = 0, c = 0, tn = 1, trn = 0, tsn = 1, cn = 0;
  uint32_t sb = 0;
  int k2s = 0, a_voff = OOB;
  const uint32_t dump = (uint32_t)(uintptr_t)(lds_void_t*)(sP + NPATCH * PATCH + 128);
  uint32_t p_dst[PPW];
  int p_soff[PPW];
  int p_q8[PPW];                 // XF: first row of the piece relative to the patch (far out of range when no piece is due)
  uint32_t xr_addr[PPW];         // XF: the pieces fetched in the previous tap (read back from LDS in this tap's second step) ...
  int xr_q8[PPW];
  int xr_soff[PPW];
#pragma unroll
  for (int e = 0; e < PPW; ++e) {
    p_dst[e] = dump;
    xr_addr[e] = dump;
    xw_addr[e] = dump;
    p_soff[e] = 0;
    p_q8[e] = -(1 << 28);
    xr_q8[e] = -(1 << 28);
    xw_q8[e] = -(1 << 28);
    xw_soff[e] = 0;
    xr_soff[e] = 0;
    if constexpr (XF != 0) asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
                                        : "=v"(xraw[e][0]), "=v"(xraw[e][1]), "=v"(xraw[e][2]), "=v"(xraw[e][3]));
  }
  int s_tmp0 = 0, s_tmp1 = 0, s_px = 0, s_due = 0;
  for (int u = 0; u < nk; ++u) {
    // ---- step 2u (k = 0..31 of this tap, fragments F[0]); reads of step 2u + 1 into F[1], then the next tap's addresses
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (TRACE) {
      if (tr_on && u < 64) {
        const long long c_ = (long long)__builtin_readcyclecounter();
        if (lane == 0) __builtin_nontemporal_store(c_, a.trace + u * 2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NMF>([&](auto nc) {
      constexpr int n = decltype(nc)::value, i = n / NT, j = n % NT;
      constexpr int G = MT + NT;            // first gap behind the fragment reads
      mfma16_agpr(acc[i][j], F[0][i], F[0][MT + j]);
      if constexpr (n < MT) read_a(B1{}, std::integral_constant<int, n>{}, (rowA + sb) ^ 0x40u);
      else if constexpr (n < G) read_b(B1{}, std::integral_constant<int, n - MT>{}, addrB[n - MT] ^ 0x40u);
      else if constexpr (n == G) tap_common_s(trn, tsn, cn);
      else if constexpr (n == G + 1) tap_common_v(tn);
      else if constexpr (n >= G + 2 && n < G + 2 + 2 * NT) {
        if constexpr (((n - (G + 2)) & 1) == 0) tap_addr_a((n - (G + 2)) >> 1);
        else tap_addr_b((n - (G + 2)) >> 1);
      } else if constexpr (n == G + 2 + 2 * NT) {          // weight tile of k-tile u + 2: tap t + 2 of this chunk, or tap t - 7 of the next
        const int wrap = t + 2 >= 9;
        s_tmp0 = (wrap ? t - 7 : t + 2) * a.C;
        s_tmp1 = (c + wrap) * BK;
        s_tmp0 = (std::remove_reference_t<decltype(s_tmp0)>)__builtin_amdgcn_readfirstlane((int)s_tmp0); s_tmp1 = (std::remove_reference_t<decltype(s_tmp1)>)__builtin_amdgcn_readfirstlane((int)s_tmp1); asm volatile("" : "+s"(s_tmp0), "+s"(s_tmp1));
      } else if constexpr (n == G + 3 + 2 * NT) {
        k2s = (s_tmp0 + s_tmp1) * 2;
        a_voff = ((int)(u + 2 < nk) & (int)dma) ? a_off0 : OOB;
        k2s = (std::remove_reference_t<decltype(k2s)>)__builtin_amdgcn_readfirstlane((int)k2s); asm volatile("" : "+s"(k2s), "+v"(a_voff));
      } else if constexpr (n >= G + 4 + 2 * NT && n < G + 4 + 2 * NT + 3 * PPW) {   // this tap's piece(s) of the next chunk's patch, three gaps each
        constexpr int e = (n - (G + 4 + 2 * NT)) / 3, part = (n - (G + 4 + 2 * NT)) % 3;
        if constexpr (part == 0) {
          if constexpr (XF != 0) {        // the piece of the previous tap moves on to its read-back
            xr_addr[e] = p_dst[e];
            xr_q8[e] = p_q8[e];
            xr_soff[e] = p_soff[e];
          }
          s_px = (t * NW + wave) * PPW + e;
          s_due = -((int)dma & (int)(t < PT) & (int)(c + 1 < nchunk) & (int)(s_px < NPI));      // all ones / zero (no branch)
          if constexpr (XF != 0) p_q8[e] = (s_px * 8 & s_due) | (-(1 << 28) & ~s_due);
          s_px = (std::remove_reference_t<decltype(s_px)>)__builtin_amdgcn_readfirstlane((int)s_px); s_due = (std::remove_reference_t<decltype(s_due)>)__builtin_amdgcn_readfirstlane((int)s_due); asm volatile("" : "+s"(s_px), "+s"(s_due));
        } else if constexpr (part == 1) {
          p_dst[e] = dump + ((ldsP + (uint32_t)(((c + 1) & 1) * PATCH + s_px * 1024) - dump) & (uint32_t)s_due);
          p_dst[e] = (std::remove_reference_t<decltype(p_dst[e])>)__builtin_amdgcn_readfirstlane((int)p_dst[e]); asm volatile("" : "+s"(p_dst[e]));
        } else {
          p_soff[e] = s_px * 8 * rowbytes + (c + 1) * (BK * 2);
          // (no pin here: with the a_out store of the transform the compiler derives this sum from a per-lane one, and an "s" constraint then asks for an illegal VGPR -> SGPR copy)
        }
      }
      if constexpr (XF != 0) xf_slot(std::integral_constant<int, xf_seq(0, n)>{}, c + 1);   // the transform of the pieces read back in the previous tap
      __builtin_amdgcn_sched_barrier(0);
    });

    // ---- step 2u + 1 (k = 32..63, fragments F[1]); the k-tile's barrier; DMA of k-tile u + 2 and of the patch piece; reads of step 2u + 2
    // (XF: the a_out stores of the transform count in vmcnt too and are interleaved with the DMA pieces in issue order — some of
    // the transform's slots lie in the second step — so no counted wait can leave "only the stores" in flight: everything is drained,
    // store acknowledgements included.  A first version that left PPW operations in flight raced with the last DMA pieces.)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (TRACE) {
      if (tr_on && u < 64) {
        const long long c_ = (long long)__builtin_readcyclecounter();
        if (lane == 0) __builtin_nontemporal_store(c_, a.trace + u * 2 + 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NMF>([&](auto nc) {
      constexpr int n = decltype(nc)::value, i = n / NT, j = n % NT;
      mfma16_agpr(acc[i][j], F[1][i], F[1][MT + j]);
      // gap n: every DSTEP-th gap issues one LDS-DMA piece (weight pieces first, then the patch piece: spread over the step so that
      // the four waves do not queue 36 pieces on the CU's one address path right behind the barrier), the gaps between them the 15
      // fragment reads of step 2u + 2, then the scalars of the next k-tile
      constexpr int ND = PA + PPW;
      constexpr bool isdma = n % DSTEP == 0 && n / DSTEP < ND;
      constexpr int ndma_before = (n + DSTEP - 1) / DSTEP < ND ? (n + DSTEP - 1) / DSTEP : ND;   // DMA gaps among 0 .. n - 1
      constexpr int rd = n - ndma_before;                                                        // index among the non-DMA gaps
      if constexpr (isdma) {
        constexpr int e = n / DSTEP;
        if constexpr (e < PA) {
          // (no k-tile u + 2: out-of-range offset — no memory traffic, zeros into the buffer nobody reads again)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + __builtin_amdgcn_readfirstlane(sb) + (e * NW + wave) * 1024), 16,
                                                   a_voff, __builtin_amdgcn_readfirstlane(k2s) + e * a_estep, 0, 0);
        } else {
          // rows in front of the tensor or behind it: the offset itself is out of the descriptor's range (negative pq_off wraps)
          // (readfirstlane: the values are uniform; it keeps the compiler from wrapping the instruction in a waterfall loop when its
          // own analysis has lost track of that)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(uintptr_t)__builtin_amdgcn_readfirstlane(p_dst[e - PA]), 16,
                                                   pq_off + p_soff[e - PA], 0, 0, 0);
        }
      } else if constexpr (rd < MT) {
        read_a(B0{}, std::integral_constant<int, rd>{}, rowA + (sb ^ (uint32_t)ASTAGE));
      } else if constexpr (rd < MT + NT) {
        read_b(B0{}, std::integral_constant<int, rd - MT>{}, addrB[rd - MT]);
      }
      if constexpr (XF != 0) {
        // the rest of the transform of the previous tap's pieces (the slots of this step) ...
        xf_slot(std::integral_constant<int, xf_seq(1, n)>{}, c + 1);
        // ... and, behind it (gaps NMF - PPW ..), the read-back of the pieces this wave fetched in the previous tap: landed (the wait at
        // the head of this step), transformed during the next k-tile
        if constexpr (n >= NMF - PPW) {
          constexpr int e = n - (NMF - PPW);
          asm volatile("ds_read_b128 %0, %1" : "=v"(xraw[e]) : "v"(xr_addr[e] + (uint32_t)(lane * 16)) : "memory");
          xw_addr[e] = xr_addr[e];
          xw_q8[e] = xr_q8[e];
          xw_soff[e] = xr_soff[e];
        }
      }
      // behind the last LDS-DMA gap (its destination is this k-tile's buffer sb): the next k-tile becomes this one
      constexpr int ADV = (ND - 1) * DSTEP + 1 > ND + MT + NT ? (ND - 1) * DSTEP + 1 : ND + MT + NT;
      if constexpr (n == ADV) {
        t = tn;
        c = cn;
        sb ^= (uint32_t)ASTAGE;
        t = (std::remove_reference_t<decltype(t)>)__builtin_amdgcn_readfirstlane((int)t); c = (std::remove_reference_t<decltype(c)>)__builtin_amdgcn_readfirstlane((int)c); sb = (std::remove_reference_t<decltype(sb)>)__builtin_amdgcn_readfirstlane((int)sb); asm volatile("" : "+s"(t), "+s"(c), "+s"(sb));
      } else if constexpr (n == ADV + 1) {
        ++tn;
        ++tsn;
        tn = (std::remove_reference_t<decltype(tn)>)__builtin_amdgcn_readfirstlane((int)tn); tsn = (std::remove_reference_t<decltype(tsn)>)__builtin_amdgcn_readfirstlane((int)tsn); asm volatile("" : "+s"(tn), "+s"(tsn));
      } else if constexpr (n == ADV + 2) {
        const int w3 = tsn == 3;
        tsn = w3 ? 0 : tsn;
        trn += w3;
        tsn = (std::remove_reference_t<decltype(tsn)>)__builtin_amdgcn_readfirstlane((int)tsn); trn = (std::remove_reference_t<decltype(trn)>)__builtin_amdgcn_readfirstlane((int)trn); asm volatile("" : "+s"(tsn), "+s"(trn));
      } else if constexpr (n == ADV + 3) {
        const int w9 = tn == 9;
        cn += w9;
        tn = w9 ? 0 : tn;
        tn = (std::remove_reference_t<decltype(tn)>)__builtin_amdgcn_readfirstlane((int)tn); cn = (std::remove_reference_t<decltype(cn)>)__builtin_amdgcn_readfirstlane((int)cn); asm volatile("" : "+s"(tn), "+s"(cn));
      } else if constexpr (n == ADV + 4) {
        const int z = tn == 0;                             // (tap 0 is only ever reached by the wrap)
        trn = z ? 0 : trn;
        tsn = z ? 0 : tsn;
        tsn = (std::remove_reference_t<decltype(tsn)>)__builtin_amdgcn_readfirstlane((int)tsn); trn = (std::remove_reference_t<decltype(trn)>)__builtin_amdgcn_readfirstlane((int)trn); asm volatile("" : "+s"(tsn), "+s"(trn));
      } else if constexpr (XF != 0 && n == ADV + 5) {
        if (t == 0) xf_load(c + 1);                        // a chunk begins: the constants of the patch fetched during it (first used in tap 2)
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  if constexpr (TRACE) {
    if (tr_on && lane == 0) __builtin_nontemporal_store((long long)__builtin_amdgcn_s_memrealtime(), a.trace + 129);
  }
  // (the last step prefetched fragments nobody uses; the s_nops: the MFMAs are inline asm, so the compiler's hazard recogniser does
  // not know that the accumulators it is about to read were written by the matrix pipe)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (a.dbg & 64) {                             // diagnostic: no epilogue
    if (acc[0][0][0] == 12345.678f) ((float*)a.y)[tid] = acc[0][0][0] + acc[3][NT - 1][3] + acc[1][2][1] + acc[MT - 1][5][2];
    return;
  }
  if constexpr (RED) {
    static_assert(BN * BM * 2 <= 160 * 1024, "the x tile fits LDS (launch_igemm_hw4 sizes the allocation for it)");
    constexpr int XRB = BM * 2, XCPR = XRB / 16, XRPI = 1024 / XRB, XI = BN / XRPI / NW;
    static_assert(BN % (XRPI * NW) == 0, "x tile pieces divide over the waves");
    const __amdgpu_buffer_rsrc_t rs_rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.red_x, 0, (int)((size_t)P * a.Mrows * 2), 0x00020000);
    const int rin = lane / XCPR, pc = lane % XCPR;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int r = (wave * XI + i) * XRPI + rin;             // pixel row of the tile
      const int lc = pc ^ (r & (XCPR - 1));
      const int q = p0 + r;
      const int off = q < P ? (q * a.Mrows + m0 + lc * 8) * 2 : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rx, (lds_void_t*)(smem + (wave * XI + i) * 1024), 16, off, 0, 0, 0);
    }
  }
  // (the lane coordinates are laundered through an empty asm: otherwise the compiler keeps per-lane pixel indices of the prologue
  // alive across the whole loop for the partial-tile path of the epilogue — registers the loop has none to spare of)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  conv_epilogue<BM, BN, WM, WN, MT, NT, NW, RED, RED, true>(a, acc, m0, p0, P, wm, wn, tid_e & 15, (tid_e & 63) >> 4, tid_e, (float*)smem, smem);
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores and atomics of this wave have been acknowledged
    if (tr_on && lane == 0) __builtin_nontemporal_store((long long)__builtin_amdgcn_s_memrealtime(), a.trace + 131);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_halo_kernel -- 3x3 / stride 1 / pad 1 convolutions (forward and input gradient) with the
// gathered operand kept as ONE halo'd pixel patch per 64-channel chunk instead of nine shifted copies.
// Written to test whether the per-CU L2->LDS volume (32 KB per 128x128x64 k-tile) bounds the LDS-DMA kernel
// above: with stride 1 the nine taps of a chunk read the same pixels shifted by d = (r-1) W + (s-1) in the
// flattened (n, h, w) index, so the pixel operand is fetched once per chunk.  Result: same
// speed at 0.6x the volume -- the volume is not the bound; kept behind vlsfr_set_option("conv_halo", v).
//   * patch = rows q0 .. q0 + PR - 1 of the flattened tensor, q0 = p0 - (W + 1), PR = 128 + 2W + 2
//     (rounded to 8), 64 channels (128 B) per row, XOR-swizzled with the patch row; double-buffered
//     across chunks, the next chunk's patch arrives in 8 slices behind the weight tiles of taps 0..7.
//   * tap (r, s): fragment rows are the 16 consecutive patch rows starting at (W + 1) + d + pixel, read
//     with the same ds_read_b128; a lane whose pixel has no such neighbour (image border: the shifted
//     row belongs to the next line / image) is pointed at a 128-byte zero row instead -- one v_cndmask
//     per fragment, the 9-bit validity mask per pixel is computed once in the prologue.
//   * per k-tile the DMA traffic drops from 32 KB to 16 KB of weights + 1/9 of a ~20-46 KB patch.
// LDS: [A stage 0][A stage 1][patch 0][patch 1][1 KiB dump: first 128 B = zero row; DMAs past the patch land here]
// ------------------------------------------------------------------------------------------------
template <int BM, int PI>
__global__ __launch_bounds__(256, 2) void conv_igemm_halo_kernel(ConvArgs a, int PR, int npatch) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BN = 128, BK = 64, NW = 4, WM = 2, WN = 2;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int RSB = 128, RPI = 8;
  constexpr int AI = BM / (NW * RPI);
  constexpr int ASTAGE = BM * RSB;
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PATCH = PR * RSB;
  const int NPI = PR / RPI;                    // DMA instructions per patch
  char* sP = smem + 2 * ASTAGE;
  char* sDump = sP + npatch * PATCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int P = a.Nimg * a.H * a.W;            // stride 1: output pixels = input pixels
  const int K = 9 * a.C;
  const int m0 = blockIdx.y * BM;
  const int p0 = blockIdx.x * BN;
  const int nchunk = a.C / BK;
  const int nk = 9 * nchunk;
  const bool fwd = a.mode == 0;

  const int rsub = lane / 8;
  const int lchunk = (lane % 8) ^ rsub;        // swz<64>(row) = row & 7, rows of an instruction start at a multiple of 8
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((size_t)a.Mrows * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)P * a.C * 2), 0x00020000);
  int a_off[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + (wave * AI + i) * RPI + rsub;
    a_off[i] = m < a.Mrows ? (m * K + lchunk * 8) * 2 : OOB;
  }
  const int q0 = p0 - (a.W + 1);               // flattened pixel of patch row 0 (may be negative)
  const int rowbytes = a.C * 2;

  // ---- per-lane validity masks of the NT pixels whose fragments this lane reads (bit tap = r*3 + s)
  uint32_t fmask[NT];
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int p = p0 + wn * (BN / WN) + j * 16 + r16;
      const int pc = p < P ? p : P - 1;
      const int n = pc / HW;
      const int rem = pc - n * HW;
      const int ho = rem / a.W;
      const int wo = rem - ho * a.W;
      uint32_t vh = 0, vw = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int dd = fwd ? r - 1 : 1 - r;
        vh |= ((unsigned)(ho + dd) < (unsigned)a.H) ? (1u << r) : 0u;
        vw |= ((unsigned)(wo + dd) < (unsigned)a.W) ? (1u << r) : 0u;
      }
      const uint32_t mask = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << 3 : 0u) | ((vh & 4u) ? vw << 6 : 0u);
      fmask[j] = p < P ? mask : 0u;
    }
  }
  if (tid < 32) ((float*)sDump)[tid] = 0.f;     // the zero row (visible after the first barrier)

  // ---- DMA issue helpers
  auto issueA = [&](int tap, int chunk, int stage) {
    const int k0 = (tap * a.C + chunk * BK) * 2;
    char* st = smem + stage * ASTAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(st + (wave * AI + i) * 1024), 16, a_off[i], k0, 0, 0);
  };
  // slice `slot` (0..7) of the patch of `chunk` into patch buffer `buf`: PI instructions per wave
  auto issueP = [&](int slot, int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < PI; ++i) {
      const int x = (slot * NW + wave) * PI + i;                 // instruction index in the patch (wave-uniform)
      const int q = q0 + x * RPI + rsub;
      const bool ok = x < NPI && (unsigned)q < (unsigned)P;
      const int off = ok ? q * rowbytes + chunk * (BK * 2) + lchunk * 16 : OOB;
      char* dst = x < NPI ? sP + buf * PATCH + x * 1024 : sDump;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)dst, 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const uint32_t ldsP = lds0 + 2 * ASTAGE;
  const uint32_t zaddr = ldsP + (uint32_t)(npatch * PATCH);      // zero row
  // prologue: weight tile 0 and the whole patch of chunk 0
  issueA(0, 0, 0);
#pragma unroll 1
  for (int t = 0; t < 8; ++t) issueP(t, 0, 0);

  int tap = 0, chunk = 0, tr = 0, ts = 0;
  bool slice_prev = false;                       // a patch slice was issued in the previous iteration
  for (int it = 0; it < nk; ++it) {
    if (slice_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // next weight tile, then (younger, so that it may stay in flight) one slice of the next chunk's patch
    {
      int ntap = tap + 1, nchk = chunk;
      if (ntap == 9) {
        ntap = 0;
        ++nchk;
      }
      if (it + 1 < nk) issueA(ntap, nchk, (it + 1) & 1);
      slice_prev = tap < 8 && chunk + 1 < nchunk;
      if (slice_prev) issueP(tap, chunk + 1, (chunk + 1) & 1);
    }
    // fragment addresses of this tap
    const int d = fwd ? (tr - 1) * a.W + (ts - 1) : (1 - tr) * a.W + (1 - ts);
    const int rowb = (a.W + 1) + d + wn * (BN / WN) + r16;
    const uint32_t pbase = ldsP + (uint32_t)((npatch > 1 ? (chunk & 1) : 0) * PATCH) + (uint32_t)(rowb * RSB);
    const uint32_t abase = lds0 + (uint32_t)((it & 1) * ASTAGE) + (uint32_t)((wm * (BM / WM) + r16) * RSB);
    bool okj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) okj[j] = (fmask[j] >> tap) & 1u;
    bf16x8 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t ra = abase + (uint32_t)((((kk * 4 + h) ^ (r16 & 7))) << 4);
      lds_read_frags<16 * RSB, 0>(fa[kk], ra, std::make_integer_sequence<int, MT>{});
      const uint32_t rb = pbase + (uint32_t)((((kk * 4 + h) ^ (rowb & 7))) << 4);
      [&]<int... J>(std::integer_sequence<int, J...>) {
        ((fb[kk][J] = lds_read128_asm<J * 16 * RSB>(okj[J] ? rb : zaddr - (uint32_t)(J * 16 * RSB))), ...);
      }(std::make_integer_sequence<int, NT>{});
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[kk][i], fb[kk][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++ts == 3) {
      ts = 0;
      ++tr;
    }
    if (++tap == 9) {
      tap = tr = ts = 0;
      ++chunk;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  conv_epilogue<BM, BN, WM, WN, MT, NT, NW>(a, acc, m0, p0, P, wm, wn, r16, h, tid, (float*)(smem + (nk & 1) * ASTAGE));
#endif
}

// ------------------------------------------------------------------------------------------------
#define VLSFR_WGRAD_GROUP_MAX 4
struct WgradArgs {
  const u16* dy;     // [P, Cout] bf16
  const u16* x;      // [Nimg, H, W, C] bf16 (forward input)
  float* dw;         // [Cout][R][S][C] fp32, accumulated atomically
  int Nimg, H, W, C;
  int Ho, Wo, Cout;
  int R, S, stride, pad;
  int splitk;
  int n_coltiles;    // column tiles per tap = ceil(C / BN)
  int dbg;           // diagnostics (vlsfr_set_option conv_dbg): 1 skips the epilogue atomics, 2 the k loop
  float* partial;    // or nullptr: [splitk][Cout][R][S][C] fp32 slabs, slice blockIdx.z written with plain stores and
                     // summed by wgrad_reduce_kernel in a fixed order (deterministic; 1 atomic per element instead of splitk)
  int gx = 0, gy = 0, xcd = 0;   // xcd != 0: 1-D grid of gx * gy * splitk workgroups in XCD-major order, tiles of one pixel slice adjacent
  // Grouped launch (vlsfr_conv2d_wgrad_group): ngroup weight gradients of the SAME descriptor — consecutive layers of a stage — in
  // one launch; problem g > 0 takes (dyg, xg, dwg)[g - 1].  The grid's slice dimension is ngroup * splitk: every problem is cut
  // into ngroup times fewer, longer slices than it would be alone, so the fp32 atomics into its gradient (one 64 KB tile per
  // workgroup, memory-side) and the launch / prologue are paid once per ngroup layers' worth of work.
  int ngroup = 1;
  const u16* dyg[VLSFR_WGRAD_GROUP_MAX - 1] = {};
  const u16* xg[VLSFR_WGRAD_GROUP_MAX - 1] = {};
  float* dwg[VLSFR_WGRAD_GROUP_MAX - 1] = {};
};

template <int BM, int BN, int KT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int MT = BM / 32;
  constexpr int NT = BN / 32;
  constexpr int RSA = BM * 2 + 32;   // LDS row strides (bytes): +32 keeps the transposed reads conflict-free
  constexpr int RSB = BN * 2 + 32;
  constexpr int ACH = KT * (BM / 8) / 256 > 0 ? KT * (BM / 8) / 256 : 1;   // 16-byte chunks per thread (KT pixels per k-tile)
  constexpr int BCH = KT * (BN / 8) / 256 > 0 ? KT * (BN / 8) / 256 : 1;
  __shared__ __attribute__((aligned(16))) char smem[2 * KT * (RSA + RSB)];
  char* sA = smem;
  char* sB = smem + 2 * KT * RSA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = a.Nimg * a.Ho * a.Wo;
  const int m0 = blockIdx.y * BM;
  const int tap = blockIdx.x / a.n_coltiles;
  const int c0 = (blockIdx.x - tap * a.n_coltiles) * BN;
  const int r = tap / a.S, s = tap - r * a.S;
  const int nkt = (P + KT - 1) / KT;
  const int per = (nkt + a.splitk - 1) / a.splitk;
  const int kt0 = blockIdx.z * per;
  const int kt1 = (kt0 + per < nkt) ? kt0 + per : nkt;
  if (kt0 >= kt1) return;

  uint4 ra[ACH], rb[BCH];
  // gathered-operand cursor per staged chunk: pixel coordinates of tile kt0, advanced by 32 pixels per
  // k-tile with carries (no divisions in the loop)
  int g_n[BCH], g_ho[BCH], g_wo[BCH];
#pragma unroll
  for (int u = 0; u < BCH; ++u) {
    const int e = tid + 256 * u;
    const int prow = e / (BN / 8);
    const int64_t p = (int64_t)kt0 * KT + prow;
    const int64_t pc = p < P ? p : (P > 0 ? P - 1 : 0);
    g_n[u] = (int)(pc / (a.Ho * a.Wo));
    const int rem = (int)(pc - (int64_t)g_n[u] * a.Ho * a.Wo);
    g_ho[u] = rem / a.Wo;
    g_wo[u] = rem - g_ho[u] * a.Wo;
  }
  auto issue = [&](int kt) {
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
      const int e = tid + 256 * u;
      const int prow = e / (BM / 8), ch = e % (BM / 8);
      const int p = kt * KT + prow;
      const bool ok = (e < KT * (BM / 8)) && p < P && (m0 + ch * 8) < a.Cout;
      ra[u] = ok ? *(const uint4*)(a.dy + (size_t)p * a.Cout + m0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int e = tid + 256 * u;
      const int prow = e / (BN / 8), ch = e % (BN / 8);
      const int p = kt * KT + prow;
      bool ok = (e < KT * (BN / 8)) && p < P && (c0 + ch * 8) < a.C;
      const int hi = g_ho[u] * a.stride - a.pad + r;
      const int wi = g_wo[u] * a.stride - a.pad + s;
      ok = ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      rb[u] = ok ? *(const uint4*)(a.x + (((size_t)g_n[u] * a.H + hi) * a.W + wi) * a.C + c0 + ch * 8)
                 : make_uint4(0, 0, 0, 0);
      g_wo[u] += KT;
      while (g_wo[u] >= a.Wo) {
        g_wo[u] -= a.Wo;
        if (++g_ho[u] >= a.Ho) {
          g_ho[u] = 0;
          ++g_n[u];
        }
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
      const int e = tid + 256 * u;
      if (e < KT * (BM / 8)) *(uint4*)(sA + buf * KT * RSA + (e / (BM / 8)) * RSA + (e % (BM / 8)) * 16) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int e = tid + 256 * u;
      if (e < KT * (BN / 8)) *(uint4*)(sB + buf * KT * RSB + (e / (BN / 8)) * RSB + (e % (BN / 8)) * 16) = rb[u];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(kt0);
  stage(0);
  __syncthreads();
  // transposed block reads: lane (q = r16 >> 2, p4 = r16 & 3) of group h addresses tile row 4h + q
  // (and + 16), columns 4 p4 .. 4 p4 + 3 of a 16-column block; k order = 16 (e >> 2) + 4h + (e & 3)
  // on both operands.
  const int trow = 4 * h + (r16 >> 2);
  const int tcol = (r16 & 3) * 8;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1);
#pragma unroll
    for (int kk = 0; kk < KT / 32; ++kk) {
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        char* base = sA + buf * KT * RSA + kk * 32 * RSA + (wm * (BM / 2) + i * 16) * 2 + tcol;
        short4v v0 = lds_read_tr16(base + trow * RSA);
        short4v v1 = lds_read_tr16(base + (trow + 16) * RSA);
        short __attribute__((ext_vector_type(8))) vs = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        fa[i] = __builtin_bit_cast(bf16x8, vs);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        char* base = sB + buf * KT * RSB + kk * 32 * RSB + (wn * (BN / 2) + j * 16) * 2 + tcol;
        short4v v0 = lds_read_tr16(base + trow * RSB);
        short4v v1 = lds_read_tr16(base + (trow + 16) * RSB);
        short __attribute__((ext_vector_type(8))) vs = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        fb[j] = __builtin_bit_cast(bf16x8, vs);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < kt1) stage(buf ^ 1);
    __syncthreads();
  }
  // ---- epilogue: D[row = cout 4h + e][col = ci r16]
  const int K = a.R * a.S * a.C;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + wm * (BM / 2) + i * 16 + 4 * h + e;
      if (m >= a.Cout) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = c0 + wn * (BN / 2) + j * 16 + r16;
        if (c < a.C) {
          if (a.partial) a.partial[(size_t)blockIdx.z * a.Cout * K + (size_t)m * K + tap * a.C + c] = acc[i][j][e];
          else atomicAdd(a.dw + (size_t)m * K + tap * a.C + c, acc[i][j][e]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_glds_kernel -- the weight gradient with the same LDS-DMA ring as the forward kernel.
// The register-staged kernel above keeps ONE 32-pixel k-tile in flight per workgroup and waits a full
// memory latency per 16 MFMAs; here KT = 64 pixels per stage, NST stages, both operands through raw
// buffer descriptors (rows past P and padded taps are out-of-range offsets -> zero fill).
//   * LDS image per stage: A [KT pixels][BM channels], B [KT pixels][BN channels] bf16, unpadded rows;
//     the 32-byte blocks (16 channels) of a row are XOR-swizzled with the pixel row so that the
//     ds_read_b64_tr_b16 of a 32-lane group (8 rows x 32 bytes) covers all 64 banks once.  The DMA writes
//     1 KiB linearly, so the swizzle is applied to the per-lane SOURCE chunk.
//   * the gathered operand's pixel coordinates advance by KT pixels per k-tile with precomputed
//     (images, rows, columns) steps and two carries -- no divisions in the loop.
// ------------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ short4v lds_read_tr_asm(uint32_t addr) {
  short4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int ROWB>
__device__ __forceinline__ int wg_swz(int row) {   // 32-byte block permutation of a ROWB-byte row
  if constexpr (ROWB % 256 == 0) return row & 7;   // rows start on the same bank: 8 rows x 8 blocks
  else return (row >> 1) & 3;                      // 128 (mod 256)-byte rows: row parity picks the bank half, 4 blocks each
}

// TPT > 1 (64-channel 3x3 layers): the column tile spans TPT = 3 taps of one filter row, 64 channels each
// (a 64 x 64 tile per tap leaves the MFMAs waiting on LDS: 4 per 8 transposed reads).
template <int BM, int BN, int KT, int NST, int TPT>
__global__ __launch_bounds__(256, 1) void conv_wgrad_glds_kernel(WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MT = BM / 32, NT = BN / 32;
  constexpr int RA = BM * 2, RB = BN * 2;            // row bytes
  constexpr int AI = KT * RA / 4096, BI = KT * RB / 4096;   // 1-KiB DMA wave-instructions per wave and stage
  constexpr int STAGE = KT * (RA + RB);
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, h = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = a.Nimg * a.Ho * a.Wo;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {   // every tile of one pixel slice reads the same dy / x rows: keep a slice's tiles behind one L2
    const int tiles = a.gx * a.gy;
    const int g = xcd_major_id(blockIdx.x, tiles * a.splitk * a.ngroup);
    bz = g / tiles;
    const int t = g - bz * tiles;
    by = t / a.gx;
    bx = t - by * a.gx;
  }
  // grouped launch: slice index -> (problem, slice of that problem)
  const u16* dyp = a.dy;
  const u16* xp = a.x;
  float* dwp = a.dw;
  int grp = 0;
  if (a.ngroup > 1) {
    grp = bz / a.splitk;
    bz -= grp * a.splitk;
    if (grp > 0) {
      dyp = a.dyg[grp - 1];
      xp = a.xg[grp - 1];
      dwp = a.dwg[grp - 1];
    }
  }
  const int m0 = by * BM;
  const int tap = TPT > 1 ? bx * TPT : bx / a.n_coltiles;          // first tap of the tile
  const int c0 = TPT > 1 ? 0 : (bx - tap * a.n_coltiles) * BN;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  const int nkt = (P + KT - 1) / KT;
  const int per = (nkt + a.splitk - 1) / a.splitk;
  const int kt0 = bz * per;
  const int kt1 = (kt0 + per < nkt) ? kt0 + per : nkt;
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc((void*)dyp, 0, (int)((size_t)P * a.Cout * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (int)((size_t)a.Nimg * a.H * a.W * a.C * 2), 0x00020000);

  // ---- A (dy): lane -> (row in instruction, physical chunk); offset linear in the pixel index
  int a_off[AI];
  {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int lin = (wave * AI + i) * 1024 + lane * 16;
      const int row = lin / RA, pc = (lin % RA) / 16;
      const int lc = (((pc >> 1) ^ wg_swz<RA>(row)) << 1) | (pc & 1);
      a_off[i] = (m0 + lc * 8 < a.Cout) ? (row * a.Cout + m0 + lc * 8) * 2 : OOB;
    }
  }
  // ---- B (x, gathered at this workgroup's tap): per-instruction pixel coordinates
  int b_n[BI], b_h[BI], b_w[BI], b_c[BI], b_s[BI];
  const int HoWo = a.Ho * a.Wo;
  const int dN = KT / HoWo, dRem = KT - dN * HoWo, dH = dRem / a.Wo, dW = dRem - dH * a.Wo;   // KT pixels as (images, rows, columns)
  {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int lin = (wave * BI + i) * 1024 + lane * 16;
      const int row = lin / RB, pc = (lin % RB) / 16;
      const int lc = (((pc >> 1) ^ wg_swz<RB>(row)) << 1) | (pc & 1);
      if (TPT > 1) {   // column = 64 * (tap in the row) + channel
        b_s[i] = ts + ((lc * 8) >> 6);
        b_c[i] = ((lc * 8) & 63) * 2;
      } else {
        b_s[i] = ts;
        b_c[i] = (c0 + lc * 8 < a.C) ? (c0 + lc * 8) * 2 : OOB;
      }
      const int64_t p = (int64_t)kt0 * KT + row;     // may be >= P: n >= Nimg marks it
      b_n[i] = (int)(p / HoWo);
      const int rem = (int)(p - (int64_t)b_n[i] * HoWo);
      b_h[i] = rem / a.Wo;
      b_w[i] = rem - b_h[i] * a.Wo;
    }
  }
  int a_soff = kt0 * KT * a.Cout * 2;
  auto issue = [&](int stage) {
    char* st = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_void_t*)(st + (wave * AI + i) * 1024), 16, a_off[i], a_soff, 0, 0);
    a_soff += KT * a.Cout * 2;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int hi = b_h[i] * a.stride - a.pad + tr;
      const int wi = b_w[i] * a.stride - a.pad + b_s[i];
      const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W && b_n[i] < a.Nimg;
      const int off = ((b_n[i] * a.H + hi) * a.W + wi) * a.C * 2 + b_c[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(st + KT * RA + (wave * BI + i) * 1024), 16,
                                               (ok && b_c[i] != OOB) ? off : OOB, 0, 0, 0);
      // advance KT pixels
      b_w[i] += dW;
      const int c1 = b_w[i] >= a.Wo;
      b_w[i] -= c1 ? a.Wo : 0;
      b_h[i] += dH + c1;
      const int c2 = b_h[i] >= a.Ho;
      b_h[i] -= c2 ? a.Ho : 0;
      b_n[i] += dN + c2;
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed block reads: lane (q = r16 >> 2, p4 = r16 & 3) of group h addresses pixel row 4h + q
  // (and + 16), bytes 8 p4 .. 8 p4 + 7 of a 32-byte block; k order = 16 (e >> 2) + 4h + (e & 3) on
  // both operands.
  const int trow = 4 * h + (r16 >> 2);
  const int tcol = (r16 & 3) * 8;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  uint32_t rdA[MT], rdB[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) rdA[i] = lds0 + trow * RA + (((wm * MT + i) ^ wg_swz<RA>(trow)) << 5) + tcol;
#pragma unroll
  for (int j = 0; j < NT; ++j) rdB[j] = lds0 + KT * RA + trow * RB + (((wn * NT + j) ^ wg_swz<RB>(trow)) << 5) + tcol;

  const int pre = nk < NST - 1 ? nk : NST - 1;
  for (int s = 0; s < pre; ++s) issue(s);

  constexpr int KK = KT / 32;
  if (!(a.dbg & 2))
  for (int it = 0; it < nk; ++it) {
    const int later = (nk - 1 - it) < (NST - 2) ? (nk - 1 - it) : (NST - 2);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AI + BI)) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + NST - 1 < nk) issue((it + NST - 1) % NST);
    const uint32_t sb = (uint32_t)((it % NST) * STAGE);
    short4v fa[KK][MT][2], fb[KK][NT][2];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        fa[kk][i][0] = kk == 0 ? lds_read_tr_asm<0>(rdA[i] + sb) : lds_read_tr_asm<32 * RA>(rdA[i] + sb);
        fa[kk][i][1] = kk == 0 ? lds_read_tr_asm<16 * RA>(rdA[i] + sb) : lds_read_tr_asm<48 * RA>(rdA[i] + sb);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        fb[kk][j][0] = kk == 0 ? lds_read_tr_asm<0>(rdB[j] + sb) : lds_read_tr_asm<32 * RB>(rdB[j] + sb);
        fb[kk][j][1] = kk == 0 ? lds_read_tr_asm<16 * RB>(rdB[j] + sb) : lds_read_tr_asm<48 * RB>(rdB[j] + sb);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      // LDS returns in order: once at most (reads of the later k-steps) remain outstanding, this k-step's
      // fragments have landed (the counter saturates at 15)
      constexpr int PER = 2 * (MT + NT);
      const int left = (KK - 1 - kk) * PER;
      if (left >= 15) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
      else if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 va[MT], vb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        short __attribute__((ext_vector_type(8))) vs = {fa[kk][i][0][0], fa[kk][i][0][1], fa[kk][i][0][2], fa[kk][i][0][3],
                                                        fa[kk][i][1][0], fa[kk][i][1][1], fa[kk][i][1][2], fa[kk][i][1][3]};
        va[i] = __builtin_bit_cast(bf16x8, vs);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        short __attribute__((ext_vector_type(8))) vs = {fb[kk][j][0][0], fb[kk][j][0][1], fb[kk][j][0][2], fb[kk][j][0][3],
                                                        fb[kk][j][1][0], fb[kk][j][1][1], fb[kk][j][1][2], fb[kk][j][1][3]};
        vb[j] = __builtin_bit_cast(bf16x8, vs);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(va[i], vb[j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (a.dbg & 1) return;
  // ---- epilogue: D[row = cout 4h + e][col = ci r16]
  const int K = a.R * a.S * a.C;
  if (a.partial) {
    // Deterministic path: this slice's tile goes to its slab with plain stores, summed later in a fixed order
    // (wgrad_reduce_kernel).  The MFMA layout gives a lane 4 rows x 1 column per accumulator register — 64-byte row
    // segments per wave-instruction if stored as it stands (measured 532 vs 605 TFLOP/s for the atomics); the tile is
    // transposed through the LDS the ring no longer needs ([BM][BN] fp32, 16-column groups XOR-swizzled with bit 2 of
    // the row: the four rows a wave-instruction writes land on two bank halves), and written out as whole 512-byte
    // rows, 16 bytes per lane.
    static_assert(BM * BN * 4 <= NST * STAGE, "the transposed tile takes the ring's LDS");
    float* T = (float*)smem;
    __syncthreads();   // every wave has read its last fragments
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = wm * (BM / 2) + i * 16 + 4 * h + e;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = wn * (BN / 2) + j * 16 + r16;
          T[row * BN + (col ^ (((row >> 2) & 1) << 4))] = acc[i][j][e];
        }
      }
    __syncthreads();
    float* slab = a.partial + ((size_t)grp * a.splitk + bz) * a.Cout * K + (size_t)tap * a.C + c0;
    constexpr int QPR = BN / 4;   // 16-byte chunks per row
    for (int idx = tid; idx < BM * QPR; idx += 256) {
      const int row = idx / QPR, q = idx - row * QPR;
      const int m = m0 + row, c = 4 * q;
      if (m < a.Cout && (TPT > 1 || c0 + c < a.C))
        *(f32x4*)(slab + (size_t)m * K + c) = *(const f32x4*)(T + row * BN + ((q ^ (((row >> 2) & 1) << 2)) << 2));
    }
    return;
  }
  // fp32 atomics into the gradient
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + wm * (BM / 2) + i * 16 + 4 * h + e;
      if (m >= a.Cout) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = c0 + wn * (BN / 2) + j * 16 + r16;
        if (TPT > 1 || c < a.C) atomicAdd(dwp + (size_t)m * K + tap * a.C + c, acc[i][j][e]);
      }
    }
  }
#endif
}

inline int out_dim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

// dw[i] += sum_z partial[z][i], z in a fixed order: the split-K weight gradient without splitk atomics per element.
// The final add stays atomic — the two backward passes of a step may run on two streams into the same gradient — but
// with exactly two contributions into a zeroed buffer the result does not depend on their order (a + b == b + a).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* dw, int64_t n4, int64_t n, int splitk) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 s = ((const f32x4*)partial)[i];
    for (int z = 1; z < splitk; ++z) {
      const f32x4 v = *(const f32x4*)(partial + (size_t)z * n + 4 * i);
      s += v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dw + 4 * i + e, s[e]);
  }
}

struct WgradPlan {
  int KT, nkt, BM, BN, grid_x, tiles, splitk, n_coltiles;
  bool row3;
};
WgradPlan wgrad_plan(const vlsfr_conv_desc* d, int splitk, int ngroup = 1) {
  WgradPlan w;
  const int Ho = out_dim(d->H, d->R, d->stride, d->pad), Wo = out_dim(d->W, d->S, d->stride, d->pad);
  const int P = d->N * Ho * Wo;
  w.KT = g_wgrad_glds ? 64 : g_wgrad_kt;
  w.nkt = (P + w.KT - 1) / w.KT;
  const bool wide = d->Cin >= 128;
  w.BM = d->Cout >= 128 ? 128 : 64;
  // 64 -> 64 channel 3x3 layers: one column tile = the three taps of a filter row (LDS-DMA kernel only)
  w.row3 = g_wgrad_glds && d->Cin == 64 && w.BM == 64 && d->R == 3 && d->S == 3;
  w.BN = w.row3 ? 192 : wide ? 128 : 64;
  w.n_coltiles = w.row3 ? 1 : (d->Cin + w.BN - 1) / w.BN;
  w.grid_x = w.row3 ? 3 : w.n_coltiles * d->R * d->S;
  w.tiles = w.grid_x * ((d->Cout + w.BM - 1) / w.BM);
  if (splitk <= 0) {   // at most g_wgrad_target workgroups (never a second round: one more slice than fits costs a third), >= 8 k-tiles each
    // (a grouped launch shares the workgroup budget between its ngroup problems: each gets ngroup times fewer, longer slices)
    splitk = g_wgrad_round_up ? (g_wgrad_target + w.tiles * ngroup - 1) / (w.tiles * ngroup) : g_wgrad_target / (w.tiles * ngroup);
    if (splitk > w.nkt / 8) splitk = w.nkt / 8;
    if (splitk < 1) splitk = 1;
  }
  // no empty slice: every z of the grid owns at least one k-tile (the slab path sums all of them)
  const int per = (w.nkt + splitk - 1) / splitk;
  w.splitk = (w.nkt + per - 1) / per;
  return w;
}

int conv_check(const vlsfr_conv_desc* d, const char* who) {
  if (!d) return fail(VLSFR_EINVAL, "%s: null descriptor", who);
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0)
    return fail(VLSFR_EINVAL, "%s: non-positive dimension", who);
  if (d->Cin % 32 != 0) return fail(VLSFR_EINVAL, "%s: Cin must be a multiple of 32 (got %d)", who, d->Cin);
  if (d->Cout % 8 != 0) return fail(VLSFR_EINVAL, "%s: Cout must be a multiple of 8 (got %d)", who, d->Cout);
  if (!((d->R == 3 && d->S == 3 && d->pad == 1) || (d->R == 1 && d->S == 1 && d->pad == 0)))
    return fail(VLSFR_EINVAL, "%s: only 3x3/pad 1 and 1x1/pad 0 filters are covered", who);
  if (d->stride != 1 && d->stride != 2) return fail(VLSFR_EINVAL, "%s: stride must be 1 or 2", who);
  return VLSFR_OK;
}


template <int BM, int BN, int BK, int NST, int NW = 4, bool PP = false, bool SWP = false, bool RED = false>
int launch_igemm_glds(const ConvArgs& a, int P, hipStream_t st) {
  constexpr int lds = NST * (BM + BN) * BK * 2;
  auto kern = conv_igemm_glds_kernel<BM, BN, BK, NST, NW, PP, SWP, RED>;
  if (int rc = ensure_dynamic_lds((const void*)kern, lds, "conv_igemm_glds")) return rc;
  dim3 grid((P + BN - 1) / BN, (a.Mrows + BM - 1) / BM, a.splitk);
  ConvArgs b = a;
  const size_t nwg = (size_t)grid.x * grid.y * grid.z;
  b.gx = (int)grid.x;
  b.gy = (int)grid.y;
  b.xcd = g_xcd_map && nwg >= 16 && nwg < (1u << 30);
  if (b.xcd) grid = dim3((unsigned)nwg, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, b);
  return VLSFR_OK;
}

template <int BN>
int launch_igemm_p8(const ConvArgs& a, int P, hipStream_t st) {
  constexpr int lds = 2 * (256 + BN) * 128;
  auto kern = conv_igemm_p8_kernel<BN>;
  if (int rc = ensure_dynamic_lds((const void*)kern, lds, "conv_igemm_p8")) return rc;
  dim3 grid(P / BN, a.Mrows / 256, 1);
  ConvArgs b = a;
  const size_t nwg = (size_t)grid.x * grid.y;
  b.gx = (int)grid.x;
  b.gy = (int)grid.y;
  b.xcd = g_xcd_map && nwg >= 16 && nwg < (1u << 30);
  if (b.xcd) grid = dim3((unsigned)nwg, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, b);
  return VLSFR_OK;
}

// LDS of conv_igemm_hp8_kernel: two weight buffers, two patches, the zero row.  lead = halo rows in front of the tile's first pixel
inline int hp8_lead(int BN, int W) { return BN % W == 0 ? W : W + 1; }
inline int hp8_patch_rows(int BN, int W) { return (BN + 2 * hp8_lead(BN, W) + 7) & ~7; }
inline int hp8_lds_bytes(int BM, int BN, int W, int npatch = 2) { return 2 * BM * 128 + npatch * hp8_patch_rows(BN, W) * 128 + 128; }

template <int BM, int NT>
int launch_igemm_hp8(const ConvArgs& a, int P, hipStream_t st) {
  constexpr int BN = (8 / (BM / 64)) * NT * 16;
  const int lead = hp8_lead(BN, a.W), PR = hp8_patch_rows(BN, a.W);
  const int lds = hp8_lds_bytes(BM, BN, a.W);
  auto kern = a.trace ? conv_igemm_hp8_kernel<BM, NT, 1> : (a.dbg & 256) ? conv_igemm_hp8_kernel<BM, NT, 2> : conv_igemm_hp8_kernel<BM, NT, 0>;
  if (int rc = ensure_dynamic_lds((const void*)kern, lds, "conv_igemm_hp8")) return rc;
  dim3 grid((P + BN - 1) / BN, (a.Mrows + BM - 1) / BM, 1);
  ConvArgs b = a;
  const size_t nwg = (size_t)grid.x * grid.y;
  b.gx = (int)grid.x;
  b.gy = (int)grid.y;
  b.xcd = g_xcd_map && nwg >= 16 && nwg < (1u << 30);
  if (b.xcd) grid = dim3((unsigned)nwg, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, b, PR, lead);
  return VLSFR_OK;
}

template <int BM, int NT>
int launch_igemm_hw4(const ConvArgs& a, int P, hipStream_t st, bool red) {
  constexpr int WROWS = (BM >= 128 && NT != 13) ? 128 : 64;
  constexpr int BN = (4 / (BM / WROWS)) * NT * 16;
  const int lead = hp8_lead(BN, a.W), PR = hp8_patch_rows(BN, a.W);
  int lds = hp8_lds_bytes(BM, BN, a.W, a.C > 64 ? 2 : 1) + 1024;   // + the dump area of the LDS-DMA pieces that have nothing to fetch
  if (red && lds < BN * BM * 2) lds = BN * BM * 2;                 // (the x tile of the fused reduction reuses the loop's LDS from offset 0)
  if constexpr (BM == 64) {
    auto kern64 = red ? conv_igemm_hw4_kernel<64, 14, 1, 0, true> : conv_igemm_hw4_kernel<64, 14, 1, 0, false>;
    if (int rc = ensure_dynamic_lds((const void*)kern64, lds, "conv_igemm_hw4")) return rc;
    dim3 grid((P + BN - 1) / BN, a.Mrows / BM, 1);
    ConvArgs b = a;
    const size_t nwg = (size_t)grid.x * grid.y;
    b.gx = (int)grid.x;
    b.gy = (int)grid.y;
    b.xcd = g_xcd_map && nwg >= 16 && nwg < (1u << 30);
    if (b.xcd) grid = dim3((unsigned)nwg, 1, 1);
    hipLaunchKernelGGL(kern64, grid, dim3(256), lds, st, b, PR, lead);
    return VLSFR_OK;
  } else {
  // pieces of the next patch per wave and tap: 8 x 4 x PPW slots per chunk, 6 x 4 x PPW with the input transform (XF)
  constexpr int XPPW = BM == 256 ? 2 : 3;
  if (a.xf_scale && PR / 8 > 24 * XPPW) return fail(VLSFR_EINVAL, "conv_igemm_hw4: patch of %d rows does not fit the transform's piece slots", PR);
  void (*kern)(ConvArgs, int, int);
  if constexpr (NT == 13) {   // 208-pixel tiles: one piece of the next patch per wave and tap (patches of at most 256 rows), no input transform
    if (a.xf_scale || PR > 256) return fail(VLSFR_EINVAL, "conv_igemm_hw4: the 208-pixel tile takes patches of at most 256 rows and no input transform");
    kern = red ? conv_igemm_hw4_kernel<BM, NT, 1, 0, true> : conv_igemm_hw4_kernel<BM, NT, 1, 0, false>;
  } else {
    kern = a.xf_scale ? (a.xf_slope ? conv_igemm_hw4_kernel<BM, NT, XPPW, 0, false, 2> : conv_igemm_hw4_kernel<BM, NT, XPPW, 0, false, 1>)
           : red      ? (PR > 256 ? conv_igemm_hw4_kernel<BM, NT, 2, 0, true> : conv_igemm_hw4_kernel<BM, NT, 1, 0, true>)
           : a.trace  ? (PR > 256 ? conv_igemm_hw4_kernel<BM, NT, 2, 1> : conv_igemm_hw4_kernel<BM, NT, 1, 1>)
                      : (PR > 256 ? conv_igemm_hw4_kernel<BM, NT, 2, 0> : conv_igemm_hw4_kernel<BM, NT, 1, 0>);
  }
  if (int rc = ensure_dynamic_lds((const void*)kern, lds, "conv_igemm_hw4")) return rc;
  dim3 grid((P + BN - 1) / BN, a.Mrows / BM, 1);
  ConvArgs b = a;
  const size_t nwg = (size_t)grid.x * grid.y;
  b.gx = (int)grid.x;
  b.gy = (int)grid.y;
  b.xcd = g_xcd_map && nwg >= 16 && nwg < (1u << 30);
  if (b.xcd) grid = dim3((unsigned)nwg, 1, 1);
  // "hw4_rounds": a launch of two or three rounds of workgroups goes out round by round, the tiles dealt evenly (448 tiles of the
  // 128-channel 28 x 28 layers = 224 + 224 instead of 256 + 192): the same number of rounds, but 32 CUs stay free for the other
  // chain's BatchNorm kernels in every round
  if (b.xcd && g_hw4_rounds && nwg > 256 && nwg <= 3 * 256) {
    const size_t rounds = (nwg + 255) / 256, per = (((nwg + rounds - 1) / rounds) + 7) & ~(size_t)7;
    for (size_t t0 = 0; t0 < nwg; t0 += per) {
      b.tile0 = (int)t0;
      hipLaunchKernelGGL(kern, dim3((unsigned)(nwg - t0 < per ? nwg - t0 : per), 1, 1), dim3(256), lds, st, b, PR, lead);
    }
    return VLSFR_OK;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b, PR, lead);
  return VLSFR_OK;
  }
}

template <int BM, int PI>
int launch_igemm_halo(const ConvArgs& a, int P, hipStream_t st) {
  const int PR = ((128 + 2 * a.W + 2) + 7) & ~7;
  const int npatch = a.C > 64 ? 2 : 1;
  const int lds = 2 * BM * 128 + npatch * PR * 128 + 1024;
  auto kern = conv_igemm_halo_kernel<BM, PI>;
  if (int rc = ensure_dynamic_lds((const void*)kern, lds, "conv_igemm_halo")) return rc;
  dim3 grid((P + 127) / 128, (a.Mrows + BM - 1) / BM, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, PR, npatch);
  return VLSFR_OK;
}

template <int BM, int BN>
void launch_igemm(const ConvArgs& a, int P, hipStream_t st) {
  dim3 grid((P + BN - 1) / BN, (a.Mrows + BM - 1) / BM, a.splitk);
  if (a.C % 64 == 0) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 64>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32>), grid, dim3(256), 0, st, a);
}

// Rows of the halo-patch kernels' tile (256 / 128) for this convolution, or 0: 3x3 / stride 1 / pad 1, 256-row tiles x 224 pixels or
// 128-row tiles x 448 pixels, where those tiles fill at least g_hp8_fill % of the workgroup slots of their rounds
// ... and whether the 256-row tile should be the 208-pixel one (conv_igemm_hw4_kernel<256, 13>: 52 accumulator tiles per wave): where
// that needs no more rounds of workgroups than the 224-pixel tile, every workgroup does 13 / 14 of the work (ir100 at batch 256, the
// 256-channel 14 x 14 layers: 242 workgroups on the 256 CUs instead of 224)
bool hw4_wants_208(int Mrows, int C, int W, int P, bool xf) {
  if (!g_hw4_208 || !g_conv_hw4 || xf || Mrows % 256 != 0 || hp8_patch_rows(208, W) > 256) return false;
  const long t7 = (long)((P + 223) / 224) * (Mrows / 256), t13 = (long)((P + 207) / 208) * (Mrows / 256);
  return g_hw4_208 == 2 || ((t13 + 255) / 256) * 208 < ((t7 + 255) / 256) * 224;
}

int hp8_tile_rows(int Mrows, int C, int H, int W, int Ho, int Wo, int R, int S, int stride, int pad, int P) {
  if (!g_conv_hp8 || !(R == 3 && S == 3 && stride == 1 && pad == 1 && Ho == H && Wo == W && H >= 2 && C % 64 == 0)) return 0;
  // (64 rows: the 64-channel layers, one chunk, one-wave-per-SIMD kernel only — "hw4_64")
  const int bm = (Mrows % 256 == 0 && g_conv_hp8 != 3) ? 256 : ((Mrows == 128 || (Mrows % 128 == 0 && g_conv_hp8 == 3)) && g_conv_hp8 != 2) ? 128
                 : (Mrows == 64 && C == 64 && g_conv_hw4 && g_hw4_64 && g_conv_hp8 == 1) ? 64 : 0;
  if (!bm) return 0;
  const int bn = bm == 256 ? 224 : bm == 128 ? 448 : 896;
  const long tiles = (long)((P + bn - 1) / bn) * (Mrows / bm), rounds = (tiles + 255) / 256;
  const int npatch = C > 64 ? 2 : 1;
  // (64-row tiles: nine k-tiles per tile, so a tile is mostly its patch fetch and its stores; measured at batch 256 they win where
  // there are many rounds of them — 112 x 112: 525 -> 385 us forward, 14 rounds — and not at 56 x 56: 98 -> 100 us, 3.5 rounds)
  if (bm == 64 && g_hp8_fill > 0 && tiles < 8 * 256) return 0;
  if (hp8_lds_bytes(bm, bn, W, npatch) + 1024 <= 160 * 1024 && (bm == 64 || hp8_patch_rows(bn, W) <= 512) && tiles * 100 >= rounds * 256 * g_hp8_fill) return bm;
  return 0;
}

// red_done (optional): set to whether this launch accumulated the BatchNorm-backward reduction of ConvArgs::red_x (only the
// default LDS-DMA tiles carry the RED epilogue; the caller runs the stand-alone reduction otherwise)
int run_igemm(ConvArgs a, hipStream_t st, bool* red_done = nullptr) {
  const int P = a.Nimg * a.Ho * a.Wo;
  a.trace = g_conv_trace;
  a.dbg = g_conv_dbg;
  a.repl = vlsfr::g_bn_repl;
  if (red_done) *red_done = false;
  // ALGORITHMIC FLOPs of the convolution this launch implements (what bench.py's roofline may count):
  // the input gradient of a stride-2 layer visits every INPUT position, but 3/4 of its taps are the
  // zero rows of the dilated dY, so it is priced at the forward's output positions (= the positions of
  // the gathered dY); the 32- / 160-channel 1x1 GEMMs are the stems on im2col rows: K = 27 real taps (3x3x3,
  // iResNet / MobileFaceNet) padded to 32, K = 147 (7x7x3, resnet_std) padded to 160.
  const double alg_pos = a.mode == 1 ? (double)a.Nimg * a.H * a.W : (double)P;
  const double alg_k = a.tap_mask ? (double)__builtin_popcount(a.tap_mask) * a.C   // parity-class launch: the taps it walks
                       : (a.mode == 0 && a.R == 1 && a.S == 1 && (a.C == 32 || a.C == 160)) ? (a.C == 32 ? 27.0 : 147.0)
                                                                                              : (double)a.R * a.S * a.C;
  const double alg_flops = 2.0 * alg_pos * (double)a.Mrows * alg_k;
  // tile choice: the 128x128 tile unless the channel count or the pixel count is small
  const long wg_big = (long)((P + 127) / 128) * ((a.Mrows + 127) / 128) * a.splitk;
  const bool glds_ok = g_use_glds && a.C % 64 == 0 && a.R * a.S <= 9 && (size_t)a.Nimg * a.H * a.W * a.C < (1ull << 30) &&
                       (size_t)a.Mrows * a.R * a.S * a.C < (1ull << 30);
  const bool halo_ok = (g_use_halo == 2 || (g_use_halo == 1 && a.Mrows < 128)) && glds_ok && !a.tap_mask && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1 && a.splitk == 1 &&
                       !a.out_f32 && a.Ho == a.H && a.Wo == a.W && a.W <= 112 && a.H >= 2 && P >= 128;
  if (a.tap_mask && !glds_ok) return fail(VLSFR_EINVAL, "conv_igemm: a parity-class launch needs the LDS-DMA kernel");
  // the default LDS-DMA variant's per-shape choices (decided here so that the timing bracket knows its family)
  const bool variant_default = g_use_glds == VLSFR_DEFAULT_CONV_VARIANT;
  const bool big_tile = a.Mrows >= 128 && wg_big >= g_small_tile_wgs;
  // the halo-patch four-phase kernel: 3x3 / stride 1, 256-row tiles x 224 pixels or 128-row tiles x 448 pixels, where those tiles
  // fill at least g_hp8_fill % of the workgroup slots of their rounds (ir100 at batch 256: 224 tiles on the 256-channel 14 x 14
  // layers, 448 on the 128-channel 28 x 28 layers).  A launch that wants the fused BatchNorm-backward reduction takes the
  // stand-alone reduction kernel instead (red_done stays false), as with the one-round tiles below.
  int hp8_bm = 0;
  if (glds_ok && variant_default && !a.tap_mask && a.splitk == 1 && !a.out_f32 && !(g_bnred_all && !(g_conv_hw4 && g_hw4_red)))
    hp8_bm = hp8_tile_rows(a.Mrows, a.C, a.H, a.W, a.Ho, a.Wo, a.R, a.S, a.stride, a.pad, P);
  if (hp8_bm != 64 && halo_ok) hp8_bm = 0;     // (the 64-row tiles replace conv_igemm_halo_kernel; everything else it still takes keeps it)
  const bool halo_here = halo_ok && !hp8_bm;
  if (a.xf_scale && !(hp8_bm && g_conv_hw4))
    return fail(VLSFR_EINVAL, "conv_igemm: the fused input BatchNorm needs conv_igemm_hw4_kernel (vlsfr_conv2d_fwd_bnin_supported)");
  // (one round of one tile per CU: the tile count over BOTH grid dimensions is bounded by the 256 CUs — 512 output channels at
  // >= 33 024 pixels would be two rounds, the case the 128 x 128 tiles win)
  const long wg_here = (a.Mrows >= 128 && wg_big >= g_small_tile_wgs) ? wg_big : (long)((P + 127) / 128) * ((a.Mrows + 63) / 64) * a.splitk;
  const bool deep_ring = g_deep_ring && (g_deep_ring == 2 || wg_here <= 256);
  const bool tile256_here = !hp8_bm && glds_ok && !halo_here && variant_default && g_tile256 && a.Mrows % 256 == 0 && !a.tap_mask && P >= 256 * g_tile256_min &&
                            (long)(a.Mrows / 256) * ((P + 255) / 256) <= 256 && !(a.red_x && g_tile256 == 2) && !g_bnred_all && a.splitk == 1;
  const bool red_here = !hp8_bm && glds_ok && !halo_here && variant_default && !tile256_here && a.red_x && !a.out_f32 && a.splitk == 1 &&
                        (size_t)(a.cls ? a.Nimg * a.Hf * a.Wf : P) * a.Mrows < (1ull << 30) && (g_bnred_all || (big_tile && !a.cls));
  // the one-wave-per-SIMD kernel carries the reduction itself ("hw4_red", default on): x tile fetched behind the loop
  const bool hw4_red = hp8_bm && g_conv_hw4 && g_hw4_red && a.red_x && !a.cls &&
                       (size_t)P * a.Mrows < (1ull << 30);
  ProfScope prof(st, (red_here || hw4_red) ? 3 : 0, alg_flops);
  if (halo_here) {
    int rc;
    const bool pi2 = (128 + 2 * a.W + 2 + 7) / 8 > 32;
    if (a.Mrows >= 128) rc = pi2 ? launch_igemm_halo<128, 2>(a, P, st) : launch_igemm_halo<128, 1>(a, P, st);
    else rc = pi2 ? launch_igemm_halo<64, 2>(a, P, st) : launch_igemm_halo<64, 1>(a, P, st);
    if (rc != VLSFR_OK) return rc;
  } else if (hp8_bm) {
    const int rc = hp8_bm == 64 ? launch_igemm_hw4<64, 14>(a, P, st, hw4_red)
                   : g_conv_hw4 ? (hp8_bm == 256 ? (hw4_wants_208(a.Mrows, a.C, a.W, P, a.xf_scale != nullptr) ? launch_igemm_hw4<256, 13>(a, P, st, hw4_red)
                                                                                                                  : launch_igemm_hw4<256, 7>(a, P, st, hw4_red))
                                                  : launch_igemm_hw4<128, 7>(a, P, st, hw4_red))
                              : (hp8_bm == 256 ? launch_igemm_hp8<256, 7>(a, P, st) : launch_igemm_hp8<128, 7>(a, P, st));
    if (hw4_red && red_done) *red_done = true;
    if (rc != VLSFR_OK) return rc;
  } else if (glds_ok) {
    int rc;
    // 128 x 128 tiles unless that leaves fewer workgroups than g_small_tile_wgs (small batches: the 256-channel 14 x 14
    // layers of batch 64 make 196 tiles for 256 CUs): then 64 x 128 tiles, twice as many
    const bool big = a.Mrows >= 128 && wg_big >= g_small_tile_wgs;
    if (g_use_glds == 1) rc = big ? launch_igemm_glds<128, 128, 64, 4>(a, P, st) : launch_igemm_glds<64, 128, 64, 4>(a, P, st);
    else if (g_use_glds == 2) rc = big ? launch_igemm_glds<128, 128, 32, 4>(a, P, st) : launch_igemm_glds<64, 128, 32, 4>(a, P, st);
    else if (g_use_glds == 5)   // 8-wave tiles, 3-stage ring, one workgroup per CU
      rc = a.Mrows >= 256 ? launch_igemm_glds<256, 128, 64, 3, 8>(a, P, st)
           : big          ? launch_igemm_glds<128, 256, 64, 3, 8>(a, P, st)
                          : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    else if (g_use_glds == 8)   // 256 x 128 tile, 4 waves of 128 x 64, three 32-deep stages, two workgroups per CU
      rc = a.Mrows >= 256 ? launch_igemm_glds<256, 128, 32, 3, 4>(a, P, st)
           : big          ? launch_igemm_glds<128, 128, 64, 2>(a, P, st)
                          : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    else if (g_use_glds == 9)   // ping-pong 8-wave tiles where the channel count allows
      rc = a.Mrows >= 256 ? launch_igemm_glds<256, 128, 64, 3, 8, true>(a, P, st)
           : big          ? launch_igemm_glds<128, 256, 64, 3, 8, true>(a, P, st)
                          : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    else if (g_use_glds == 10)   // largest tiles the channel count allows, standard 2-stage loop, 8 waves, one workgroup per CU
      rc = a.Mrows >= 256 ? launch_igemm_glds<256, 256, 64, 2, 8>(a, P, st)
           : big          ? launch_igemm_glds<128, 256, 64, 2, 8>(a, P, st)
                          : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    else if (g_use_glds == 11)
      rc = a.Mrows >= 256 ? launch_igemm_glds<256, 128, 64, 2, 8>(a, P, st)
           : big          ? launch_igemm_glds<128, 256, 64, 2, 8>(a, P, st)
                          : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    else if (g_use_glds == 12)   // 128x128 tile on 8 waves (64x32 each): half the per-wave instruction stream, 4 waves per SIMD
      rc = big ? launch_igemm_glds<128, 128, 64, 2, 8>(a, P, st) : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    else if (g_use_glds == 13)   // 64-channel layers on 64 x 256 tiles (half the workgroups, prologue / epilogue amortised)
      rc = big ? launch_igemm_glds<128, 128, 64, 2>(a, P, st) : launch_igemm_glds<64, 256, 64, 2>(a, P, st);
    else if (g_use_glds == 14)   // software-pipelined loop (fragments of tile t + 1 read between the MFMAs of tile t)
      rc = big ? launch_igemm_glds<128, 128, 64, 2, 4, false, true>(a, P, st)
               : launch_igemm_glds<64, 128, 64, 2, 4, false, true>(a, P, st);
    else if (g_use_glds == 6) rc = big ? launch_igemm_glds<128, 128, 32, 3>(a, P, st) : launch_igemm_glds<64, 128, 32, 3>(a, P, st);
    else if (g_use_glds == 7) rc = big ? launch_igemm_glds<128, 128, 32, 2>(a, P, st) : launch_igemm_glds<64, 128, 32, 2>(a, P, st);
    else if (g_use_glds == 4) rc = big ? launch_igemm_glds<128, 128, 32, 5>(a, P, st) : launch_igemm_glds<64, 128, 32, 5>(a, P, st);
    else if (tile256_here) {
      // 256-channel layers with 160 - 256 pixel tiles of 256 (ir100 at batch 256: the 58 forward convolutions at 14 x 14): one
      // 256 x 256 tile per CU, 8 waves, ONE round of 196 tiles instead of 784 tiles of 128 x 128 in 1.53 rounds on 512 slots
      // (scripts/conv_shapes.py: 73.4 vs 81.7 us; at equal fill the two tiles run at the same rate, 995 vs 954 TFLOP/s).
      // "tile256": 0 off, 1 forward + input gradient (the fused BatchNorm-backward reduction has no 256 x 256 form: that
      // launch then takes the stand-alone reduction kernel), 2 forward only (default)
      // 224 pixels per tile where that is still one round: 50 176 pixels = 224 tiles of 224 on 256 CUs instead of 196 of 256,
      // i.e. 0.875 of the work on the critical CU ("tile224", scripts/conv_shapes.py)
      const bool p8_ok = g_conv_p8 && a.R * a.S <= 9 && (a.mode == 0 || a.stride == 1) && !a.out_f32;   // (a.red_x: the caller runs the stand-alone reduction)
      if (g_tile224 && (long)(a.Mrows / 256) * ((P + 223) / 224) <= 256) {
        if (p8_ok && P % 224 == 0) rc = launch_igemm_p8<224>(a, P, st);
        else rc = launch_igemm_glds<256, 224, 64, 2, 8>(a, P, st);
      } else if (p8_ok && P % 256 == 0) rc = launch_igemm_p8<256>(a, P, st);
      else rc = launch_igemm_glds<256, 256, 64, 2, 8>(a, P, st);
    } else if (red_here) {
      // measured per launch at batch 256 (scripts/dgrad_bnred_micro.py): the fused epilogue beats "plain launch + stand-alone
      // reduction kernel" on the 128 x 128 tiles of the stride-1 layers (128 / 256 / 512 channels: 4 - 9 us saved of 20 - 30)
      // and loses on the 64-channel layers (6 - 12 tile rounds per launch: the epilogue is paid per tile) and on the four
      // parity-class launches of a stride-2 layer; "bnred_all" = 1 forces it everywhere (tests)
      if (deep_ring) rc = big ? launch_igemm_glds<128, 128, 64, 4, 4, false, false, true>(a, P, st)
                              : launch_igemm_glds<64, 128, 64, 4, 4, false, false, true>(a, P, st);
      else rc = big ? launch_igemm_glds<128, 128, 64, 2, 4, false, false, true>(a, P, st)
                    : launch_igemm_glds<64, 128, 64, 2, 4, false, false, true>(a, P, st);
      if (red_done) *red_done = true;
    } else if (deep_ring) {
      // at most one workgroup per CU (small batches: 196 tiles on the 256-channel 14 x 14 layers of batch 64, 100 on the
      // 512-channel 7 x 7 ones): no second workgroup hides the LDS-DMA latency, and one k-tile of 32 MFMAs per wave (~0.25 us)
      // is far shorter than it — a 4-stage ring (three k-tiles in flight, 128 KB of LDS) instead of the 2-stage one
      rc = big ? launch_igemm_glds<128, 128, 64, 4>(a, P, st) : launch_igemm_glds<64, 128, 64, 4>(a, P, st);
    } else rc = big ? launch_igemm_glds<128, 128, 64, 2>(a, P, st) : launch_igemm_glds<64, 128, 64, 2>(a, P, st);
    if (rc != VLSFR_OK) return rc;
  } else if (a.Mrows >= 128 && wg_big >= 192) launch_igemm<128, 128>(a, P, st);
  else if (a.Mrows >= 128) launch_igemm<128, 64>(a, P, st);
  else launch_igemm<64, 128>(a, P, st);
  VLSFR_HIP_CHECK_LAUNCH("conv_igemm launch");
  return VLSFR_OK;
}

// the stand-alone reduction for launches that carried no RED epilogue (register-staged / halo / A-B tile variants)
int bn_red_fallback(const vlsfr_conv_desc* d, const void* dx, const vlsfr_bn_red* bn, void* stream) {
  return vlsfr_bn_backward_reduce(dx, bn->x, (int64_t)d->N * d->H * d->W, d->Cin, d->H * d->W, bn->mean, bn->invstd, bn->gamma,
                                  bn->beta, bn->slope, bn->red, stream);
}

}  // namespace

extern "C" {

void vlsfr_profile_enable(int32_t on) {
  if (on && vlsfr::prof_pool_reserve(1 << 16) != 0) on = 0;   // 32 768 brackets (ir100: ~2 700 launches of the timed families per step)
  vlsfr::g_prof_on = on != 0;
}

int vlsfr_set_option(const char* name, int32_t value) {
  if (name && !strcmp(name, "wgrad_round_up")) {
    g_wgrad_round_up = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "small_tile_wgs")) {
    g_small_tile_wgs = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "dgrad_classes")) {
    g_dgrad_classes = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "xcd_map")) {
    g_xcd_map = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_glds")) {
    g_use_glds = value < 0 ? VLSFR_DEFAULT_CONV_VARIANT : value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "dw_wgrad_blocks")) {
    vlsfr::g_dw_wgrad_blocks = value > 0 ? value : 256;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "hw4_rounds")) {
    g_hw4_rounds = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "hw4_208")) {
    g_hw4_208 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_deep_ring")) {
    g_deep_ring = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_halo")) {
    g_use_halo = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "bn_repl")) {
    if (value < 1 || value > VLSFR_BN_REPL) return fail(VLSFR_EINVAL, "vlsfr_set_option: bn_repl must be in [1, %d]", VLSFR_BN_REPL);
    vlsfr::g_bn_repl = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "prof_pool")) {
    vlsfr::g_prof_pool_on = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "tile256")) {
    g_tile256 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_p8")) {
    g_conv_p8 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_hp8")) {
    g_conv_hp8 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_hw4")) {
    g_conv_hw4 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_bnin")) {
    g_conv_bnin = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "hw4_64")) {
    g_hw4_64 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "hw4_red")) {
    g_hw4_red = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "hp8_fill")) {
    g_hp8_fill = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "tile256_min")) {
    g_tile256_min = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "tile224")) {
    g_tile224 = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "bnred_all")) {
    g_bnred_all = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "conv_dbg")) {
    g_conv_dbg = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "wgrad_glds")) {
    g_wgrad_glds = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "wgrad_kt")) {
    g_wgrad_kt = value == 64 ? 64 : 32;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "wgrad_slabs")) {
    g_wgrad_slabs = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "wgrad_target_wgs")) {
    g_wgrad_target = value > 0 ? value : 512;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "bn_chain")) {
    vlsfr::g_bn_chain = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "wgrad_group")) {
    vlsfr::g_wgrad_group = value;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "dgrad_bnred")) {
    vlsfr::g_dgrad_bnred = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "dw_strip")) {
    vlsfr::g_dw_strip = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "bn_xcd")) {
    extern int g_bn_xcd;
    g_bn_xcd = value != 0;
    return VLSFR_OK;
  }
  if (name && !strcmp(name, "bn_block_kb")) {
    extern int g_bn_block_bytes;
    g_bn_block_bytes = value > 0 ? value * 1024 : 65536;
    return VLSFR_OK;
  }
  if (name) {
    const int rc = vlsfr::head_set_option(name, value);   // csrc/head.hip ("head_variant")
    if (rc <= 0) return rc;
  }
  return fail(VLSFR_EINVAL, "vlsfr_set_option: unknown option");
}

int vlsfr_conv_trace(void* device_buffer) {   // diagnostics, see ConvArgs::trace
  g_conv_trace = (long long*)device_buffer;
  return VLSFR_OK;
}

int vlsfr_profile_collect(int32_t family, double* total_ms, double* total_flops, int64_t* launches) {
  if (!total_ms || !total_flops || !launches) return fail(VLSFR_EINVAL, "vlsfr_profile_collect: null argument");
  double ms = 0, fl = 0;
  int64_t n = 0;
  std::lock_guard<std::mutex> lk(vlsfr::g_prof_mu);
  for (auto& r : vlsfr::g_prof) {
    if (r.family != family) continue;
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e, "vlsfr_profile_collect: hipEventSynchronize");
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e, "vlsfr_profile_collect: hipEventElapsedTime");
    ms += t;
    fl += r.work;
    ++n;
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = n;
  return VLSFR_OK;
}

// Mean elapsed time (microseconds) between two events recorded back to back on `stream` with nothing in between: what a
// ProfScope bracket reads on top of the kernel's own duration.  bench.py subtracts it per launch.
double vlsfr_profile_event_overhead_us(void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int N = 64;
  if (vlsfr::prof_pool_reserve(1 << 16) != 0) return 0.0;
  std::lock_guard<std::mutex> lk(vlsfr::g_prof_mu);
  if (vlsfr::g_prof_pool.size() < 2 * (size_t)N) return 0.0;
  hipEvent_t* ev = vlsfr::g_prof_pool.data() + (vlsfr::g_prof_pool.size() - 2 * N);   // the pool's last events: never handed to a bracket while this runs
  for (int i = 0; i < N; ++i) {
    (void)hipEventRecord(ev[2 * i], st);
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  (void)hipStreamSynchronize(st);
  double sum = 0.0;
  for (int i = 0; i < N; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) == hipSuccess) sum += t;
  }
  return sum / N * 1e3;
}

void vlsfr_profile_reset(void) {   // pooled events go back to the pool; events created for one bracket ("prof_pool" = 0) are destroyed
  std::lock_guard<std::mutex> lk(vlsfr::g_prof_mu);
  for (auto& r : vlsfr::g_prof)
    if (!r.pooled) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
  vlsfr::g_prof.clear();
  vlsfr::g_prof_next = 0;
  vlsfr::g_prof_dropped = 0;
}

int64_t vlsfr_profile_dropped(void) {   // brackets that were not taken since the last reset: > 0 = the collected totals undercount
  std::lock_guard<std::mutex> lk(vlsfr::g_prof_mu);
  return (int64_t)vlsfr::g_prof_dropped;
}

int vlsfr_conv2d_fwd(const vlsfr_conv_desc* d, const void* x, const void* w, void* y, int32_t splitk, int32_t out_f32,
                     double* stats, void* stream) {
  int rc = conv_check(d, "vlsfr_conv2d_fwd");
  if (rc) return rc;
  if (!x || !w || !y) return fail(VLSFR_EINVAL, "vlsfr_conv2d_fwd: null buffer");
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && !out_f32) return fail(VLSFR_EINVAL, "vlsfr_conv2d_fwd: split-K needs the fp32 (atomic) output");
  if (stats && out_f32) return fail(VLSFR_EINVAL, "vlsfr_conv2d_fwd: fused statistics need the bf16 output");
  ConvArgs a;
  a.x = (const u16*)x;
  a.w = (const u16*)w;
  a.y = y;
  a.Nimg = d->N;
  a.H = d->H;
  a.W = d->W;
  a.C = d->Cin;
  a.Ho = out_dim(d->H, d->R, d->stride, d->pad);
  a.Wo = out_dim(d->W, d->S, d->stride, d->pad);
  a.Mrows = d->Cout;
  a.R = d->R;
  a.S = d->S;
  a.stride = d->stride;
  a.pad = d->pad;
  a.mode = 0;
  a.splitk = splitk;
  a.out_f32 = out_f32;
  a.stats = stats;
  return run_igemm(a, (hipStream_t)stream);
}

int32_t vlsfr_conv2d_fwd_bnin_supported(const vlsfr_conv_desc* d) {
  if (!d || conv_check(d, "vlsfr_conv2d_fwd_bnin_supported")) return 0;
  if (!(g_use_glds == VLSFR_DEFAULT_CONV_VARIANT && g_conv_hw4 && g_conv_bnin)) return 0;   // (the executors ask here: "conv_bnin" = 0 keeps them on bn_apply)
  const int Ho = out_dim(d->H, d->R, d->stride, d->pad), Wo = out_dim(d->W, d->S, d->stride, d->pad);
  const size_t P = (size_t)d->N * Ho * Wo;
  if (d->Cin % 64 || (size_t)d->N * d->H * d->W * d->Cin >= (1ull << 30) || (size_t)d->Cout * 9 * d->Cin >= (1ull << 30)) return 0;
  const int bm = hp8_tile_rows(d->Cout, d->Cin, d->H, d->W, Ho, Wo, d->R, d->S, d->stride, d->pad, (int)P);
  if (!bm) return 0;
  const int bn = bm == 256 ? 224 : 448;
  return hp8_patch_rows(bn, d->W) / 8 <= 24 * (bm == 256 ? 2 : 3);
}

int vlsfr_conv2d_fwd_bnin(const vlsfr_conv_desc* d, const void* x, const void* w, void* y, double* stats, const vlsfr_bn_in* bn,
                          void* stream) {
  int rc = conv_check(d, "vlsfr_conv2d_fwd_bnin");
  if (rc) return rc;
  if (!x || !w || !y || !bn || !bn->scale || !bn->shift) return fail(VLSFR_EINVAL, "vlsfr_conv2d_fwd_bnin: null buffer");
  if (!vlsfr_conv2d_fwd_bnin_supported(d)) return fail(VLSFR_EINVAL, "vlsfr_conv2d_fwd_bnin: shape not covered (vlsfr_conv2d_fwd_bnin_supported)");
  ConvArgs a;
  a.x = (const u16*)x;
  a.w = (const u16*)w;
  a.y = y;
  a.Nimg = d->N;
  a.H = d->H;
  a.W = d->W;
  a.C = d->Cin;
  a.Ho = d->H;
  a.Wo = d->W;
  a.Mrows = d->Cout;
  a.R = d->R;
  a.S = d->S;
  a.stride = d->stride;
  a.pad = d->pad;
  a.mode = 0;
  a.splitk = 1;
  a.out_f32 = 0;
  a.stats = stats;
  a.xf_scale = bn->scale;
  a.xf_shift = bn->shift;
  a.xf_slope = bn->slope;
  a.xf_out = (u16*)bn->a_out;
  return run_igemm(a, (hipStream_t)stream);
}

int vlsfr_conv2d_dgrad(const vlsfr_conv_desc* d, const void* dy, const void* wT, void* dx, void* stream) {
  return vlsfr_conv2d_dgrad_bnred(d, dy, wT, dx, nullptr, stream);
}

int vlsfr_conv2d_dgrad_bnred(const vlsfr_conv_desc* d, const void* dy, const void* wT, void* dx, const vlsfr_bn_red* bn,
                             void* stream) {
  int rc = conv_check(d, "vlsfr_conv2d_dgrad");
  if (rc) return rc;
  if (!dy || !wT || !dx) return fail(VLSFR_EINVAL, "vlsfr_conv2d_dgrad: null buffer");
  if (d->Cout % 32 != 0) return fail(VLSFR_EINVAL, "vlsfr_conv2d_dgrad: Cout must be a multiple of 32");
  if (bn && (!bn->x || !bn->mean || !bn->invstd || !bn->red || d->Cin % 8))
    return fail(VLSFR_EINVAL, "vlsfr_conv2d_dgrad_bnred: x, mean, invstd and red are required (Cin %% 8 == 0)");
  ConvArgs a;
  if (bn) {
    a.red_x = (const u16*)bn->x;
    a.red_mean = bn->mean;
    a.red_invstd = bn->invstd;
    a.red_gamma = bn->gamma;
    a.red_beta = bn->beta;
    a.red_slope = bn->slope;
    a.red_out = bn->red;
  }
  a.x = (const u16*)dy;
  a.w = (const u16*)wT;
  a.y = dx;
  a.Nimg = d->N;
  a.H = out_dim(d->H, d->R, d->stride, d->pad);   // the gathered tensor is dY
  a.W = out_dim(d->W, d->S, d->stride, d->pad);
  a.C = d->Cout;
  a.Ho = d->H;                                    // one output "pixel" per input position
  a.Wo = d->W;
  a.Mrows = d->Cin;
  a.R = d->R;
  a.S = d->S;
  a.stride = d->stride;
  a.pad = d->pad;
  a.mode = 1;
  a.splitk = 1;
  a.out_f32 = 0;
  a.stats = nullptr;
  // stride 2: one launch per parity class of the input positions, each a forward-mode gather over dY with the taps that
  // class can see (ConvArgs::cls) — 9/4 tap-positions per input position instead of 9
  const int Hd = a.H, Wd = a.W;
  const bool f33 = d->R == 3 && d->S == 3 && d->pad == 1, f11 = d->R == 1 && d->S == 1 && d->pad == 0;
  if (g_dgrad_classes && g_use_glds && d->stride == 2 && (f33 || f11) && d->H == 2 * Hd && d->W == 2 * Wd && d->Cout % 64 == 0 &&
      (size_t)d->N * Hd * Wd * d->Cout < (1ull << 30) && (size_t)d->Cin * d->R * d->S * d->Cout < (1ull << 30) &&
      (size_t)d->N * Hd * Wd >= 128) {
    hipStream_t st = (hipStream_t)stream;
    a.mode = 0;
    a.stride = 1;
    a.Ho = Hd;
    a.Wo = Wd;
    a.Hf = d->H;
    a.Wf = d->W;
    if (f11) {   // only the even positions see dY: the rest of dX is zero
      // (a kernel, not hipMemsetAsync: this call sits inside backbone passes that may be captured into a HIP graph, norm.hip)
      if (int rz = vlsfr_zero_bytes(dx, (size_t)d->N * d->H * d->W * d->Cin * 2, (void*)st)) return rz;
      a.tap_mask = 1u;
      a.tap_w = 0;
      a.cls = 1;
      bool done = false;     // zero rows add nothing to any of the three sums: the class launch's share is the whole reduction
      rc = run_igemm(a, st, &done);
      if (rc || !bn || done) return rc;
      return bn_red_fallback(d, dx, bn, stream);
    }
    bool all_done = true, any_done = false;
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        // virtual tap r' (row offset r' - 1 on the dY grid) <-> filter tap r:  ph = 0: r' = 1 <-> r = 1;
        // ph = 1: r' = 1 <-> r = 2, r' = 2 <-> r = 0   (2 i - 1 + r = 2 h' + ph)
        const int nr = ph ? 2 : 1, ns = pw ? 2 : 1;
        const int vr[2] = {1, 2}, fr1[2] = {2, 0};
        a.tap_mask = 0;
        a.tap_w = 0;
        for (int x = 0; x < nr; ++x)
          for (int y = 0; y < ns; ++y) {
            const int rv = vr[x], sv = vr[y];
            const int rf = ph ? fr1[x] : 1, sf = pw ? fr1[y] : 1;
            a.tap_mask |= 1u << (rv * 3 + sv);
            a.tap_w |= (unsigned long long)(rf * 3 + sf) << (4 * (rv * 3 + sv));
          }
        a.cls = 1 + 2 * ph + pw;
        bool done = false;
        rc = run_igemm(a, st, &done);
        if (rc) return rc;
        all_done = all_done && done;
        any_done = any_done || done;
      }
    if (bn && !all_done) {
      if (any_done) return fail(VLSFR_EINVAL, "vlsfr_conv2d_dgrad_bnred: parity-class launches disagree");
      return bn_red_fallback(d, dx, bn, stream);   // none of the four accumulated (the default for class launches)
    }
    return VLSFR_OK;
  }
  bool done = false;
  rc = run_igemm(a, (hipStream_t)stream, &done);
  if (rc || !bn || done) return rc;
  return bn_red_fallback(d, dx, bn, stream);
}

size_t vlsfr_conv2d_wgrad_workspace_bytes(const vlsfr_conv_desc* d, int32_t splitk) {
  if (conv_check(d, "vlsfr_conv2d_wgrad_workspace_bytes")) return 0;
  const WgradPlan w = wgrad_plan(d, splitk);
  return w.splitk > 1 ? (size_t)w.splitk * d->Cout * d->R * d->S * d->Cin * sizeof(float) : 0;
}

int vlsfr_conv2d_wgrad(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, int32_t splitk,
                       void* stream) {
  return vlsfr_conv2d_wgrad_ws(d, dy, x, dw, splitk, nullptr, 0, stream);
}

int vlsfr_conv2d_wgrad_ws(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, int32_t splitk,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return vlsfr_conv2d_wgrad_group(d, 1, &dy, &x, &dw, splitk, workspace, workspace_bytes, stream);
}

size_t vlsfr_conv2d_wgrad_group_workspace_bytes(const vlsfr_conv_desc* d, int32_t n, int32_t splitk) {
  if (conv_check(d, "vlsfr_conv2d_wgrad_group_workspace_bytes") || n < 1 || n > VLSFR_WGRAD_GROUP_MAX) return 0;
  const WgradPlan w = wgrad_plan(d, splitk, n);
  return w.splitk > 1 ? (size_t)n * w.splitk * d->Cout * d->R * d->S * d->Cin * sizeof(float) : 0;
}

int vlsfr_conv2d_wgrad_group(const vlsfr_conv_desc* d, int32_t n, const void* const* dy, const void* const* x, float* const* dw,
                             int32_t splitk, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = conv_check(d, "vlsfr_conv2d_wgrad");
  if (rc) return rc;
  if (n < 1 || n > VLSFR_WGRAD_GROUP_MAX || !dy || !x || !dw)
    return fail(VLSFR_EINVAL, "vlsfr_conv2d_wgrad_group: 1 .. %d problems", VLSFR_WGRAD_GROUP_MAX);
  for (int g = 0; g < n; ++g)
    if (!dy[g] || !x[g] || !dw[g]) return fail(VLSFR_EINVAL, "vlsfr_conv2d_wgrad: null buffer");
  WgradArgs a;
  a.Nimg = d->N;
  a.H = d->H;
  a.W = d->W;
  a.C = d->Cin;
  a.Ho = out_dim(d->H, d->R, d->stride, d->pad);
  a.Wo = out_dim(d->W, d->S, d->stride, d->pad);
  a.Cout = d->Cout;
  a.R = d->R;
  a.S = d->S;
  a.stride = d->stride;
  a.pad = d->pad;
  const int P = a.Nimg * a.Ho * a.Wo;
  const bool glds = g_wgrad_glds && d->Cin % 8 == 0 && d->Cout % 8 == 0 && (size_t)P * d->Cout < (1ull << 30) &&
                    (size_t)a.Nimg * a.H * a.W * a.C < (1ull << 30);
  if (n > 1 && !glds) {   // the register-staged kernel has no grouped form: one launch per problem
    for (int g = 0; g < n; ++g)
      if ((rc = vlsfr_conv2d_wgrad_group(d, 1, dy + g, x + g, dw + g, splitk, workspace, workspace_bytes, stream))) return rc;
    return VLSFR_OK;
  }
  a.dy = (const u16*)dy[0];
  a.x = (const u16*)x[0];
  a.dw = dw[0];
  a.ngroup = n;
  for (int g = 1; g < n; ++g) {
    a.dyg[g - 1] = (const u16*)dy[g];
    a.xg[g - 1] = (const u16*)x[g];
    a.dwg[g - 1] = dw[g];
  }
  const WgradPlan w = wgrad_plan(d, splitk, n);
  const int KT = w.KT, BM = w.BM, BN = w.BN, grid_x = w.grid_x;
  splitk = w.splitk;
  a.n_coltiles = w.n_coltiles;
  a.splitk = splitk;
  const size_t n_dw = (size_t)d->Cout * d->R * d->S * d->Cin;
  const size_t need = (size_t)n * splitk * n_dw * sizeof(float);
  // slab path: enough workspace, more than one slice, 16-byte granularity of the reduction
  a.partial = (workspace && splitk > 1 && workspace_bytes >= need && n_dw % 4 == 0 && g_wgrad_slabs) ? (float*)workspace : nullptr;
  a.dbg = g_conv_dbg;
  dim3 grid(grid_x, (d->Cout + BM - 1) / BM, splitk * n);
  a.gx = (int)grid.x;
  a.gy = (int)grid.y;
  a.xcd = 0;
  hipStream_t st = (hipStream_t)stream;
  // algorithmic FLOPs (the 32-channel 1x1 case is the stem on im2col rows: 27 real taps)
  const double alg_k = (d->R == 1 && d->S == 1 && (d->Cin == 32 || d->Cin == 160)) ? (d->Cin == 32 ? 27.0 : 147.0)
                                                                                   : (double)d->R * d->S * d->Cin;
  ProfScope prof(st, 1, 2.0 * n * P * (double)d->Cout * alg_k);
  if (glds && g_xcd_map && (size_t)grid.x * grid.y * grid.z >= 16) {
    a.xcd = 1;
    grid = dim3(grid.x * grid.y * grid.z, 1, 1);
  }
#define VLSFR_WGRAD(BM_, BN_)                                                                           \
  do {                                                                                                 \
    if (glds) {                                                                                        \
      constexpr int lds_ = 2 * 64 * (BM_ + BN_) * 2;                                                   \
      auto kern_ = conv_wgrad_glds_kernel<BM_, BN_, 64, 2, 1>;                                            \
      if (int rc_ = ensure_dynamic_lds((const void*)kern_, lds_, "conv_wgrad_glds")) return rc_;       \
      hipLaunchKernelGGL(kern_, grid, dim3(256), lds_, st, a);                                         \
    } else if (KT == 64) hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, 64>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, 32>), grid, dim3(256), 0, st, a);               \
  } while (0)
  if (BN == 192) {
    constexpr int lds_ = 2 * 64 * (64 + 192) * 2;
    auto kern_ = conv_wgrad_glds_kernel<64, 192, 64, 2, 3>;
    if (int rc_ = ensure_dynamic_lds((const void*)kern_, lds_, "conv_wgrad_glds")) return rc_;
    hipLaunchKernelGGL(kern_, grid, dim3(256), lds_, st, a);
  } else if (BM == 128 && BN == 128) VLSFR_WGRAD(128, 128);
  else if (BM == 128) VLSFR_WGRAD(128, 64);
  else if (BN == 128) VLSFR_WGRAD(64, 128);
  else VLSFR_WGRAD(64, 64);
#undef VLSFR_WGRAD
  VLSFR_HIP_CHECK_LAUNCH("conv_wgrad launch");
  if (a.partial) {
    const int64_t n4 = (int64_t)n_dw / 4;
    const int64_t blocks = (n4 + 255) / 256;
    for (int g = 0; g < n; ++g)
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st,
                         a.partial + (size_t)g * splitk * n_dw, dw[g], n4, (int64_t)n_dw, splitk);
    VLSFR_HIP_CHECK_LAUNCH("conv_wgrad reduce launch");
  }
  return VLSFR_OK;
}

}  // extern "C"
