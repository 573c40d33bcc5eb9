// Native executor of the iResNet backbone (C-ABI section 7 of include/vlsfr.h): one call runs the
// whole forward (or backward) pass by enqueueing the gfx950 kernels of conv.hip / norm.hip on the
// caller's stream — no per-operator host round trip.  Architecture and semantics: reference
// model/resnet_arcface.py:26-55 (IBasicBlock: BN -> 3x3 -> BN -> PReLU -> 3x3(stride) -> BN (+ shortcut))
// and :58-152 (stem conv/BN/PReLU at full resolution, 4 stages whose first block is stride 2 with a
// 1x1-s2 + BN shortcut, BN -> flatten -> fc -> BN1d(weight frozen) -> L2 normalise), in training
// mode.  Parameter / buffer order = registration order of the reference module (named_parameters()).
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "common_host.h"

namespace vlsfr {
int g_bn_chain = 1;   // "bn_chain": bn1 backward of block k accumulates the reduction of bn3 of block k - 1 (vlsfr_bn_backward_chain)
int g_dgrad_bnred = 1;   // "dgrad_bnred": the reductions of bn2 / bn1 backward come from the epilogue of the input-gradient
                         // convolution that writes their dY (vlsfr_conv2d_dgrad_bnred) instead of a kernel of their own
}
namespace vlsfr {
int g_wgrad_group = 4;   // "wgrad_group": weight gradients of up to this many consecutive same-shape layers in ONE launch
                         // (vlsfr_conv2d_wgrad_group; 1 = every layer on its own, the round-3 executor)
}
using vlsfr::g_bn_chain;
using vlsfr::g_dgrad_bnred;
using vlsfr::g_wgrad_group;

namespace {

constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOM = 0.1f;

struct Bn {
  int C;
  int p_w, p_b;      // parameter indices (gamma, beta)
  int p_slope;       // PReLU parameter index or -1
  int run;           // index of the (running_mean, running_var) pair
  size_t off_sums;   // ctx: float64 [REPL][2][C] statistics accumulators (zeroed at forward start)
  size_t off_mean;   // ctx: fp32 [C]
  size_t off_invstd; // ctx: fp32 [C]
  size_t off_scale;  // ctx: fp32 [C]
  size_t off_shift;  // ctx: fp32 [C]
  size_t off_red;    // ctx: fp32 [REPL][3][C] backward reductions (zeroed at backward start)
  size_t off_kcoef;  // ctx: fp32 [3][C]
};

struct Conv {
  vlsfr_conv_desc d;
  int p_w;
  size_t off_wb, off_wT;   // wcache (bytes)
};

struct Block {
  int cin, planes, stride, H, W, Ho, Wo;
  Bn bn1, bn2, bn3, bnd;
  Conv conv1, conv2, convd;
  bool has_ds;
  // ctx activations (byte offsets, bf16)
  size_t a1, c1, a2, c2, cs, out;
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr int VLSFR_WGRAD_GROUP = 4;              // the C-ABI's limit of problems per grouped weight-gradient launch
constexpr int WGRAD_RING = VLSFR_WGRAD_GROUP + 2; // dY buffers: one per waiting weight gradient + the one being written + one of slack

}  // namespace

struct vlsfr_iresnet {
  int layers[4];
  int D, B, HW0;
  int n_params = 0, n_bn = 0;
  // stem
  Conv stem;   // 1x1 over the 32-wide im2col rows
  Bn stem_bn;
  size_t off_cols, off_c0, off_a0;
  std::vector<Block> blocks;
  Bn bn_last;
  size_t off_flat;   // bf16 [B, 25088] in the reference's flatten order
  Conv fc;
  int p_fc_b, p_feat_w, p_feat_b, run_feat;
  size_t off_fcout, off_z, off_xhat, off_feat_invstd, off_emb, off_invnorm;
  size_t sums_begin, sums_end, red_begin, red_end;
  size_t ctx_bytes = 0, wcache_bytes = 0, scratch_bytes = 0;
  size_t max_act = 0;   // largest activation tensor in bytes
  size_t wgrad_ws = 0;  // split-K slabs of the largest weight gradient (vlsfr_conv2d_wgrad_ws)

  size_t take_ctx(size_t bytes) {
    size_t o = ctx_bytes;
    ctx_bytes += align_up(bytes);
    return o;
  }
  size_t take_w(size_t bytes) {
    size_t o = wcache_bytes;
    wcache_bytes += align_up(bytes);
    return o;
  }
  Bn make_bn(int C, bool prelu_after_in_order) {
    (void)prelu_after_in_order;
    Bn b;
    b.C = C;
    b.p_w = n_params++;
    b.p_b = n_params++;
    b.p_slope = -1;
    b.run = n_bn++;
    b.off_sums = b.off_mean = b.off_invstd = b.off_scale = b.off_shift = b.off_red = b.off_kcoef = 0;
    return b;
  }
  Conv make_conv(int N, int H, int W, int cin, int cout, int k, int stride) {
    Conv c;
    c.d = vlsfr_conv_desc{N, H, W, cin, cout, k, k, stride, k == 3 ? 1 : 0};
    c.p_w = n_params++;
    const size_t bytes = (size_t)cout * k * k * cin * 2;
    c.off_wb = take_w(bytes);
    c.off_wT = take_w(bytes);
    return c;
  }
  void bn_ctx(Bn& b) {
    b.off_mean = take_ctx((size_t)b.C * 4);
    b.off_invstd = take_ctx((size_t)b.C * 4);
    b.off_scale = take_ctx((size_t)b.C * 4);
    b.off_shift = take_ctx((size_t)b.C * 4);
    b.off_kcoef = take_ctx((size_t)3 * b.C * 4);
  }
  void bn_red(Bn& b) { b.off_red = take_ctx((size_t)VLSFR_BN_REPL * 3 * b.C * 4); }
};

namespace {

using vlsfr::fail;

int build(vlsfr_iresnet* n) {
  const int B = n->B, S = n->HW0;
  // ---- parameter order: registration order of the reference module
  n->stem.d = vlsfr_conv_desc{B, S, S, 32, 64, 1, 1, 1, 0};
  n->stem.p_w = n->n_params++;                      // conv1.weight
  n->stem.off_wb = n->take_w((size_t)64 * 32 * 2);
  n->stem.off_wT = 0;
  n->stem_bn = n->make_bn(64, true);                // bn1.weight, bn1.bias
  n->stem_bn.p_slope = n->n_params++;               // prelu.weight
  int cin = 64, H = S;
  const int planes_of[4] = {64, 128, 256, 512};
  for (int li = 0; li < 4; ++li) {
    for (int bi = 0; bi < n->layers[li]; ++bi) {
      Block b;
      b.cin = cin;
      b.planes = planes_of[li];
      b.stride = bi == 0 ? 2 : 1;
      b.H = b.W = H;
      b.Ho = b.Wo = (H + 2 - 3) / b.stride + 1;
      b.bn1 = n->make_bn(cin, false);
      b.conv1 = n->make_conv(B, H, H, cin, b.planes, 3, 1);
      b.bn2 = n->make_bn(b.planes, true);
      b.bn2.p_slope = n->n_params++;
      b.conv2 = n->make_conv(B, H, H, b.planes, b.planes, 3, b.stride);
      b.bn3 = n->make_bn(b.planes, false);
      b.has_ds = bi == 0;
      if (b.has_ds) {
        b.convd = n->make_conv(B, H, H, cin, b.planes, 1, b.stride);
        b.bnd = n->make_bn(b.planes, false);
      }
      n->blocks.push_back(b);
      cin = b.planes;
      H = b.Ho;
    }
  }
  if (H * H * 512 % 32 != 0) return fail(VLSFR_EINVAL, "iresnet: unsupported image size");
  n->bn_last = n->make_bn(512, false);              // bn2.*
  const int Kfc = 512 * H * H;
  n->fc.d = vlsfr_conv_desc{B, 1, 1, Kfc, n->D, 1, 1, 1, 0};
  n->fc.p_w = n->n_params++;                        // fc.weight
  n->fc.off_wb = n->take_w((size_t)n->D * Kfc * 2);
  n->fc.off_wT = n->take_w((size_t)n->D * Kfc * 2);
  n->p_fc_b = n->n_params++;                        // fc.bias
  n->p_feat_w = n->n_params++;                      // features.weight (frozen)
  n->p_feat_b = n->n_params++;                      // features.bias
  n->run_feat = n->n_bn++;

  // ---- ctx layout: statistics first (one memset clears every sums slot), then activations
  n->sums_begin = n->ctx_bytes;
  auto sums = [&](Bn& b) { b.off_sums = n->take_ctx((size_t)VLSFR_BN_REPL * 2 * b.C * 8); };
  sums(n->stem_bn);
  for (auto& b : n->blocks) {
    sums(b.bn1);
    sums(b.bn2);
    sums(b.bn3);
    if (b.has_ds) sums(b.bnd);
  }
  sums(n->bn_last);
  n->off_fcout = n->take_ctx((size_t)B * n->D * 4);   // zeroed with the sums (split-K accumulates into it)
  n->sums_end = n->ctx_bytes;
  n->red_begin = n->ctx_bytes;
  n->bn_red(n->stem_bn);
  for (auto& b : n->blocks) {
    n->bn_red(b.bn1);
    n->bn_red(b.bn2);
    n->bn_red(b.bn3);
    if (b.has_ds) n->bn_red(b.bnd);
  }
  n->bn_red(n->bn_last);
  n->red_end = n->ctx_bytes;
  n->bn_ctx(n->stem_bn);
  for (auto& b : n->blocks) {
    n->bn_ctx(b.bn1);
    n->bn_ctx(b.bn2);
    n->bn_ctx(b.bn3);
    if (b.has_ds) n->bn_ctx(b.bnd);
  }
  n->bn_ctx(n->bn_last);
  const size_t P0 = (size_t)B * S * S;
  n->off_cols = n->take_ctx(P0 * 32 * 2);
  n->off_c0 = n->take_ctx(P0 * 64 * 2);
  n->off_a0 = n->take_ctx(P0 * 64 * 2);
  n->max_act = P0 * 64 * 2;
  for (auto& b : n->blocks) {
    const size_t pin = (size_t)B * b.H * b.W, pout = (size_t)B * b.Ho * b.Wo;
    b.a1 = n->take_ctx(pin * b.cin * 2);
    b.c1 = n->take_ctx(pin * b.planes * 2);
    b.a2 = n->take_ctx(pin * b.planes * 2);
    b.c2 = n->take_ctx(pout * b.planes * 2);
    b.cs = b.has_ds ? n->take_ctx(pout * b.planes * 2) : 0;
    b.out = n->take_ctx(pout * b.planes * 2);
    if (pin * b.planes * 2 > n->max_act) n->max_act = pin * b.planes * 2;
    if (pin * b.cin * 2 > n->max_act) n->max_act = pin * b.cin * 2;
  }
  n->off_flat = n->take_ctx((size_t)B * Kfc * 2);
  n->off_z = n->take_ctx((size_t)B * n->D * 4);
  n->off_xhat = n->take_ctx((size_t)B * n->D * 4);
  n->off_feat_invstd = n->take_ctx((size_t)n->D * 4);
  n->off_emb = n->take_ctx((size_t)B * n->D * 4);
  n->off_invnorm = n->take_ctx((size_t)B * 4);
  auto ws_of = [&](const Conv& c) {
    for (int g = 1; g <= VLSFR_WGRAD_GROUP; ++g) {
      const size_t w = vlsfr_conv2d_wgrad_group_workspace_bytes(&c.d, g, 0);
      if (w > n->wgrad_ws) n->wgrad_ws = w;
    }
  };
  ws_of(n->stem);
  for (auto& b : n->blocks) {
    ws_of(b.conv1);
    ws_of(b.conv2);
    if (b.has_ds) ws_of(b.convd);
  }
  ws_of(n->fc);
  // scratch: 3 activation-sized gradient buffers + the shortcut tensor + small fp32 scratch + the wgrad slabs + the ring of
  // WGRAD_RING activation-sized buffers that hold the dY of weight gradients waiting for their grouped launch
  n->scratch_bytes = (4 + WGRAD_RING) * align_up(n->max_act) + align_up((size_t)3 * 2048 * 4) + align_up((size_t)64 * 32 * 4) +
                     align_up((size_t)B * n->D * 4) + align_up((size_t)B * n->D * 2) + align_up(n->wgrad_ws);
  return VLSFR_OK;
}

struct Scratch {
  char* g[3];
  char* ring[WGRAD_RING];
  char* idn;
  float* red;
  float* stem_dw;
  float* dz;
  char* dfc;
  void* wgrad_ws;
};

Scratch carve(const vlsfr_iresnet* n, void* scratch) {
  Scratch s;
  char* p = (char*)scratch;
  const size_t a = align_up(n->max_act);
  for (int i = 0; i < 3; ++i) s.g[i] = p + i * a;
  s.idn = p + 3 * a;
  for (int i = 0; i < WGRAD_RING; ++i) s.ring[i] = p + (4 + i) * a;
  p += (4 + WGRAD_RING) * a;
  s.red = (float*)p;
  p += align_up((size_t)3 * 2048 * 4);
  s.stem_dw = (float*)p;
  p += align_up((size_t)64 * 32 * 4);
  s.dz = (float*)p;
  p += align_up((size_t)n->B * n->D * 4);
  s.dfc = p;
  p += align_up((size_t)n->B * n->D * 2);
  s.wgrad_ws = p;
  return s;
}

#define RUN(expr)             \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != VLSFR_OK) return rc__; \
  } while (0)

// Weight gradients waiting for their grouped launch.  Nothing in the backward chain reads a weight gradient, so the executor
// parks (descriptor, dY, saved input) of a layer here and launches up to g_wgrad_group consecutive layers of one shape together
// (vlsfr_conv2d_wgrad_group: fewer, longer pixel slices per layer — the fp32 atomics and the launch are shared).  The dY of a
// waiting entry lives in a ring slot of its own (Scratch::ring) that is not handed out again before the entry has been launched.
struct WgradQueue {
  const vlsfr_iresnet* n;
  float* const* grads;
  const Scratch* sc;
  void* st;
  const Conv* conv[VLSFR_WGRAD_GROUP];
  const void* dy[VLSFR_WGRAD_GROUP];
  const void* x[VLSFR_WGRAD_GROUP];
  int slot[VLSFR_WGRAD_GROUP];
  int count = 0, next = 0;
  int flush() {
    if (!count) return VLSFR_OK;
    float* dw[VLSFR_WGRAD_GROUP];
    for (int i = 0; i < count; ++i) dw[i] = grads[conv[i]->p_w];
    const int rc = vlsfr_conv2d_wgrad_group(&conv[0]->d, count, dy, x, dw, 0, sc->wgrad_ws, n->wgrad_ws, st);
    count = 0;
    return rc;
  }
  // a ring slot for the dY of the next weight gradient (flushes first if the slot still belongs to a waiting entry)
  int acquire(char** out) {
    const int s = next;
    next = (next + 1) % WGRAD_RING;
    for (int i = 0; i < count; ++i)
      if (slot[i] == s) {
        const int rc = flush();
        if (rc) return rc;
        break;
      }
    *out = sc->ring[s];
    cur_slot = s;
    return VLSFR_OK;
  }
  int cur_slot = -1;
  int push(const Conv& c, const void* dyv, const void* xv) {   // dyv: the slot handed out by the last acquire()
    const int limit = g_wgrad_group < 1 ? 1 : (g_wgrad_group > VLSFR_WGRAD_GROUP ? VLSFR_WGRAD_GROUP : g_wgrad_group);
    if (count && (std::memcmp(&conv[0]->d, &c.d, sizeof(vlsfr_conv_desc)) != 0 || count >= limit)) {
      const int rc = flush();
      if (rc) return rc;
    }
    conv[count] = &c;
    dy[count] = dyv;
    x[count] = xv;
    slot[count] = cur_slot;
    ++count;
    return count >= limit ? flush() : VLSFR_OK;
  }
};

// statistics (already accumulated by the producer of x) -> scale/shift, then y = prelu(bn(x)) + residual
int bn_forward(const Bn& b, const void* x, void* y, int64_t M, int HW, const void* residual, double* out_sums,
               int out_nchw, const float* const* params, float* const* running, char* ctx, void* st) {
  float* rm = running ? running[2 * b.run] : nullptr;
  float* rv = running ? running[2 * b.run + 1] : nullptr;
  return vlsfr_bn_apply(x, y, M, b.C, HW, (const double*)(ctx + b.off_sums), params[b.p_w], params[b.p_b],
                        b.p_slope >= 0 ? params[b.p_slope] : nullptr, residual, (float*)(ctx + b.off_mean),
                        (float*)(ctx + b.off_invstd), rm, rv, BN_EPS, BN_MOM, out_sums, out_nchw, st);
}

// red_ready: this layer's reduction was accumulated by the call that produced dy (its `next`); next / next_x: the
// BatchNorm whose dY is this call's dx, and that layer's input (vlsfr_bn_backward_chain)
int bn_backward(const Bn& b, const void* dy, const void* x, void* dx, int64_t M, int HW, const void* dx_add,
                int dy_nchw, const float* const* params, float* const* grads, char* ctx, void* st, int red_ready = 0,
                const Bn* next = nullptr, const void* next_x = nullptr) {
  return vlsfr_bn_backward_chain(dy, x, dx, M, b.C, HW, (const float*)(ctx + b.off_mean), (const float*)(ctx + b.off_invstd),
                                 params[b.p_w], params[b.p_b], b.p_slope >= 0 ? params[b.p_slope] : nullptr,
                                 (float*)(ctx + b.off_red), dx_add, grads[b.p_w], grads[b.p_b],
                                 b.p_slope >= 0 ? grads[b.p_slope] : nullptr, dy_nchw, red_ready, next ? next_x : nullptr,
                                 next ? (const float*)(ctx + next->off_mean) : nullptr,
                                 next ? (const float*)(ctx + next->off_invstd) : nullptr,
                                 next ? (float*)(ctx + next->off_red) : nullptr, st);
}

// IBasicBlock.forward (resnet_arcface.py:44-55) of block k on the activation `cur` (bf16 NHWC; its statistics are already in
// bn1's sums: every BatchNorm's batch statistics are accumulated by the kernel that PRODUCES its input)
// y = conv(bn(x)) (+ PReLU between them): where the convolution kernel can apply the BatchNorm in its operand path
// (vlsfr_conv2d_fwd_bnin: the 3x3 / stride-1 layers on conv_igemm_hw4_kernel) only the per-channel part of the BatchNorm runs as a
// kernel of its own (vlsfr_bn_finalize, one workgroup) and the normalised tensor `a` comes out of the convolution as a by-product —
// or not at all in a pass that keeps no activations (keep == false: the gallery passes); elsewhere bn_apply + conv as before.
int bn_conv_forward(const Bn& bn, const Conv& cv, const char* x, char* a_buf, char* y, double* y_sums, int64_t M, int HW, bool keep,
                    const float* const* params, float* const* running, char* ctx, const char* wc, void* st) {
  if (vlsfr_conv2d_fwd_bnin_supported(&cv.d)) {
    float* rm = running ? running[2 * bn.run] : nullptr;
    float* rv = running ? running[2 * bn.run + 1] : nullptr;
    RUN(vlsfr_bn_finalize((const double*)(ctx + bn.off_sums), M, bn.C, params[bn.p_w], params[bn.p_b], BN_EPS, BN_MOM,
                          (float*)(ctx + bn.off_mean), (float*)(ctx + bn.off_invstd), (float*)(ctx + bn.off_scale),
                          (float*)(ctx + bn.off_shift), rm, rv, st));
    const vlsfr_bn_in in{(const float*)(ctx + bn.off_scale), (const float*)(ctx + bn.off_shift),
                         bn.p_slope >= 0 ? params[bn.p_slope] : nullptr, keep ? a_buf : nullptr};
    return vlsfr_conv2d_fwd_bnin(&cv.d, x, wc + cv.off_wb, y, y_sums, &in, st);
  }
  RUN(bn_forward(bn, x, a_buf, M, HW, nullptr, nullptr, 0, params, running, ctx, st));
  return vlsfr_conv2d_fwd(&cv.d, a_buf, wc + cv.off_wb, y, 1, 0, y_sums, st);
}

int forward_block(const vlsfr_iresnet* n, int k, const char* cur, const float* const* params, float* const* running, char* ctx,
                  const char* wc, const Scratch& sc, void* st, bool keep = true) {
  const int B = n->B;
  auto sums_of = [&](const Bn& b) { return (double*)(ctx + b.off_sums); };
  const Block& b = n->blocks[k];
  const Bn& next_bn = (size_t)k + 1 < n->blocks.size() ? n->blocks[k + 1].bn1 : n->bn_last;
  const int64_t Min = (int64_t)B * b.H * b.W, Mout = (int64_t)B * b.Ho * b.Wo;
  RUN(bn_conv_forward(b.bn1, b.conv1, cur, ctx + b.a1, ctx + b.c1, sums_of(b.bn2), Min, b.H * b.W, keep, params, running, ctx, wc, st));
  RUN(bn_conv_forward(b.bn2, b.conv2, ctx + b.c1, ctx + b.a2, ctx + b.c2, sums_of(b.bn3), Min, b.H * b.W, keep, params, running, ctx, wc, st));
  const void* idn = cur;
  if (b.has_ds) {
    RUN(vlsfr_conv2d_fwd(&b.convd.d, cur, wc + b.convd.off_wb, ctx + b.cs, 1, 0, sums_of(b.bnd), st));
    RUN(bn_forward(b.bnd, ctx + b.cs, sc.idn, Mout, b.Ho * b.Wo, nullptr, nullptr, 0, params, running, ctx, st));
    idn = sc.idn;
  }
  return bn_forward(b.bn3, ctx + b.c2, ctx + b.out, Mout, b.Ho * b.Wo, idn, sums_of(next_bn), 0, params, running, ctx, st);
}

// Backward of block k: dout = gradient of the block output (in sc.g[cur_i]); leaves the gradient of the block input in
// sc.g[(cur_i + 1) % 3].  `chained`: bn3's reduction came with the kernel that wrote dout; chain_prev: accumulate the
// reduction of block k - 1's bn3 while writing this block's input gradient (vlsfr_bn_backward_chain).
int backward_block(const vlsfr_iresnet* n, int k, int cur_i, bool chained, bool chain_prev, const float* const* params,
                   float* const* grads, char* ctx, const char* wc, const Scratch& sc, WgradQueue& wq, void* st) {
  const int B = n->B;
  const Block& b = n->blocks[k];
  const char* x_in = k > 0 ? ctx + n->blocks[k - 1].out : ctx + n->off_a0;
  const int64_t Min = (int64_t)B * b.H * b.W, Mout = (int64_t)B * b.Ho * b.Wo;
  char* t1 = sc.g[(cur_i + 1) % 3];
  char* t2 = sc.g[(cur_i + 2) % 3];
  const char* dout = sc.g[cur_i];
  // The dY of every convolution goes to a ring slot of its own: its weight gradient is launched later, together with its neighbours'
  char* r2 = nullptr;
  char* r1 = nullptr;
  // main branch
  RUN(wq.acquire(&r2));
  RUN(bn_backward(b.bn3, dout, ctx + b.c2, r2, Mout, b.Ho * b.Wo, nullptr, 0, params, grads, ctx, st, chained ? 1 : 0));
  RUN(wq.push(b.conv2, r2, ctx + b.a2));
  // the input-gradient convolutions accumulate the reductions of the BatchNorm backward that reads their output
  const int fused = g_dgrad_bnred ? 1 : 0;
  auto red_of = [&](const Bn& bn, const void* x) {
    return vlsfr_bn_red{x, (const float*)(ctx + bn.off_mean), (const float*)(ctx + bn.off_invstd), params[bn.p_w], params[bn.p_b],
                        bn.p_slope >= 0 ? params[bn.p_slope] : nullptr, (float*)(ctx + bn.off_red)};
  };
  const vlsfr_bn_red red2 = red_of(b.bn2, ctx + b.c1), red1 = red_of(b.bn1, x_in);
  RUN(vlsfr_conv2d_dgrad_bnred(&b.conv2.d, r2, wc + b.conv2.off_wT, t2, fused ? &red2 : nullptr, st));   // d a2
  RUN(wq.acquire(&r1));
  RUN(bn_backward(b.bn2, t2, ctx + b.c1, r1, Min, b.H * b.W, nullptr, 0, params, grads, ctx, st, fused));   // d c1
  RUN(wq.push(b.conv1, r1, ctx + b.a1));
  RUN(vlsfr_conv2d_dgrad_bnred(&b.conv1.d, r1, wc + b.conv1.off_wT, t2, fused ? &red1 : nullptr, st));   // d a1 (in t2)
  const char* add = dout;
  if (b.has_ds) {   // shortcut branch: d cs, then its weight and input gradients
    char* rs = nullptr;
    RUN(wq.acquire(&rs));
    RUN(bn_backward(b.bnd, dout, ctx + b.cs, rs, Mout, b.Ho * b.Wo, nullptr, 0, params, grads, ctx, st));
    RUN(wq.push(b.convd, rs, x_in));
    RUN(vlsfr_conv2d_dgrad(&b.convd.d, rs, wc + b.convd.off_wT, sc.idn, st));
    add = sc.idn;
  }
  // d x_in = bn1 backward of d a1, plus the shortcut gradient; x_in is the output of block k - 1, so this IS the dY of
  // that block's bn3: its reduction is accumulated here (one read of c2 instead of a kernel reading dout and c2)
  return bn_backward(b.bn1, t2, x_in, t1, Min, b.H * b.W, add, 0, params, grads, ctx, st, fused,
                     chain_prev ? &n->blocks[k - 1].bn3 : nullptr, chain_prev ? ctx + n->blocks[k - 1].c2 : nullptr);
}

}  // namespace

extern "C" {

int vlsfr_iresnet_create(const int32_t* layers, int32_t feat_dim, int32_t batch, int32_t image_hw, vlsfr_iresnet** out) {
  if (!layers || !out || feat_dim <= 0 || feat_dim % 8 || batch <= 0 || image_hw <= 0 || image_hw % 16)
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_create: need feat_dim %% 8 == 0 and image size %% 16 == 0");
  for (int i = 0; i < 4; ++i)
    if (layers[i] < 1) return fail(VLSFR_EINVAL, "vlsfr_iresnet_create: every stage needs at least one block");
  vlsfr_iresnet* n = new (std::nothrow) vlsfr_iresnet();
  if (!n) return fail(VLSFR_ENOMEM, "vlsfr_iresnet_create: out of memory");
  std::memcpy(n->layers, layers, sizeof(n->layers));
  n->D = feat_dim;
  n->B = batch;
  n->HW0 = image_hw;
  int rc = build(n);
  if (rc != VLSFR_OK) {
    delete n;
    return rc;
  }
  *out = n;
  return VLSFR_OK;
}

void vlsfr_iresnet_destroy(vlsfr_iresnet* n) { delete n; }
int32_t vlsfr_iresnet_num_params(const vlsfr_iresnet* n) { return n ? n->n_params : -1; }
int32_t vlsfr_iresnet_num_bn(const vlsfr_iresnet* n) { return n ? n->n_bn : -1; }
size_t vlsfr_iresnet_wcache_bytes(const vlsfr_iresnet* n) { return n ? n->wcache_bytes : 0; }
size_t vlsfr_iresnet_ctx_bytes(const vlsfr_iresnet* n) { return n ? n->ctx_bytes : 0; }
size_t vlsfr_iresnet_scratch_bytes(const vlsfr_iresnet* n) { return n ? n->scratch_bytes : 0; }

int vlsfr_iresnet_prepare_weights(const vlsfr_iresnet* n, const float* const* params, void* wcache, void* st) {
  if (!n || !params || !wcache) return fail(VLSFR_EINVAL, "vlsfr_iresnet_prepare_weights: null argument");
  char* wc = (char*)wcache;
  std::vector<vlsfr_cast_entry> tab;
  tab.push_back({params[n->stem.p_w], wc + n->stem.off_wb, nullptr, 64, 1, 27, 32});
  auto add = [&](const Conv& c) {
    tab.push_back({params[c.p_w], wc + c.off_wb, wc + c.off_wT, c.d.Cout, c.d.R * c.d.S, c.d.Cin, c.d.R * c.d.S * c.d.Cin});
  };
  for (const auto& b : n->blocks) {
    add(b.conv1);
    add(b.conv2);
    if (b.has_ds) add(b.convd);
  }
  add(n->fc);
  return vlsfr_cast_weights(tab.data(), (int32_t)tab.size(), st);
}

int vlsfr_iresnet_forward(const vlsfr_iresnet* n, const float* x_nchw, const float* const* params,
                          float* const* running, const void* wcache, void* ctx_v, void* scratch, float* emb_out,
                          void* st) {
  return vlsfr_iresnet_forward_ex(n, x_nchw, params, running, wcache, ctx_v, scratch, emb_out, 1, st);
}

int vlsfr_iresnet_forward_ex(const vlsfr_iresnet* n, const float* x_nchw, const float* const* params,
                             float* const* running, const void* wcache, void* ctx_v, void* scratch, float* emb_out,
                             int32_t keep_activations, void* st) {
  if (!n || !x_nchw || !params || !wcache || !ctx_v || !scratch || !emb_out)
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_forward: null argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  RUN(vlsfr_zero_bytes(ctx + n->sums_begin, n->sums_end - n->sums_begin, st));
  hipError_t e = hipSuccess;
  const int B = n->B, S = n->HW0;
  auto sums_of = [&](const Bn& b) { return (double*)(ctx + b.off_sums); };
  // Every BatchNorm's batch statistics are accumulated by the kernel that PRODUCES its input (conv
  // epilogue or the previous bn_apply), so no tensor is read just to be averaged.
  // stem (resnet_arcface.py:140-142)
  RUN(vlsfr_stem_im2col(x_nchw, ctx + n->off_cols, B, S, S, 1, st));
  RUN(vlsfr_conv2d_fwd(&n->stem.d, ctx + n->off_cols, wc + n->stem.off_wb, ctx + n->off_c0, 1, 0, sums_of(n->stem_bn), st));
  RUN(bn_forward(n->stem_bn, ctx + n->off_c0, ctx + n->off_a0, (int64_t)B * S * S, S * S, nullptr,
                 sums_of(n->blocks[0].bn1), 0, params, running, ctx, st));
  const char* cur = ctx + n->off_a0;
  for (size_t k = 0; k < n->blocks.size(); ++k) {
    RUN(forward_block(n, (int)k, cur, params, running, ctx, wc, sc, st, keep_activations != 0));
    cur = ctx + n->blocks[k].out;
  }
  // bn2 -> flatten -> fc -> features -> normalise (resnet_arcface.py:147-151)
  const Block& last = n->blocks.back();
  const int HWl = last.Ho * last.Wo;
  RUN(bn_forward(n->bn_last, cur, ctx + n->off_flat, (int64_t)B * HWl, HWl, nullptr, nullptr, 1, params, running, ctx,
                 st));
  const int Kfc = n->fc.d.Cin;
  int splitk = (Kfc / 32) / 12;
  if (splitk < 1) splitk = 1;
  if (splitk > 64) splitk = 64;
  RUN(vlsfr_conv2d_fwd(&n->fc.d, ctx + n->off_flat, wc + n->fc.off_wb, ctx + n->off_fcout, splitk, 1, nullptr, st));
  float* rm = running ? running[2 * n->run_feat] : nullptr;
  float* rv = running ? running[2 * n->run_feat + 1] : nullptr;
  RUN(vlsfr_embed_fwd((const float*)(ctx + n->off_fcout), params[n->p_fc_b], params[n->p_feat_w], params[n->p_feat_b],
                      rm, rv, (float*)(ctx + n->off_z), (float*)(ctx + n->off_xhat),
                      (float*)(ctx + n->off_feat_invstd), (float*)(ctx + n->off_emb), (float*)(ctx + n->off_invnorm),
                      B, n->D, BN_EPS, BN_MOM, st));
  (void)e;
  return vlsfr_copy_bytes(ctx + n->off_emb, emb_out, (size_t)B * n->D * 4, st);   // (a kernel: captured passes hold no runtime copy nodes)
}

int vlsfr_iresnet_backward(const vlsfr_iresnet* n, const float* demb, const float* const* params, float* const* grads,
                           const void* wcache, void* ctx_v, void* scratch, void* st) {
  return vlsfr_iresnet_backward_staged(n, demb, params, grads, wcache, ctx_v, scratch, nullptr, st);
}

int vlsfr_iresnet_backward_staged(const vlsfr_iresnet* n, const float* demb, const float* const* params,
                                  float* const* grads, const void* wcache, void* ctx_v, void* scratch,
                                  void* const* stage_events, void* st) {
  if (!n || !demb || !params || !grads || !wcache || !ctx_v || !scratch)
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_backward: null argument");
  // stage_events[k] is recorded on the stream once every parameter gradient of bucket k has been enqueued:
  // 0 = tail (bn2, fc, features), 1..3 = layer4..layer2, 4 = layer1 + stem (backward order)
  auto signal = [&](int k) -> int {
    if (!stage_events || !stage_events[k]) return VLSFR_OK;
    hipError_t ee = hipEventRecord((hipEvent_t)stage_events[k], (hipStream_t)st);
    return ee == hipSuccess ? VLSFR_OK : fail(VLSFR_EHIP, "vlsfr_iresnet_backward: hipEventRecord: %s", hipGetErrorString(ee));
  };
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  const int B = n->B, S = n->HW0;
  const Block& last = n->blocks.back();
  const int HWl = last.Ho * last.Wo;
  RUN(vlsfr_zero_bytes(ctx + n->red_begin, n->red_end - n->red_begin, st));
  // embedding tail, fc
  RUN(vlsfr_embed_bwd(demb, (const float*)(ctx + n->off_emb), (const float*)(ctx + n->off_invnorm),
                      (const float*)(ctx + n->off_xhat), (const float*)(ctx + n->off_feat_invstd),
                      params[n->p_feat_w], sc.dz, sc.dfc, grads[n->p_feat_b], grads[n->p_fc_b], nullptr, B, n->D, st));
  RUN(vlsfr_conv2d_wgrad_ws(&n->fc.d, sc.dfc, ctx + n->off_flat, grads[n->fc.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
  char* dflat = sc.g[0];
  RUN(vlsfr_conv2d_dgrad(&n->fc.d, sc.dfc, wc + n->fc.off_wT, dflat, st));
  char* dcur = sc.g[1];
  RUN(bn_backward(n->bn_last, dflat, ctx + last.out, dcur, (int64_t)B * HWl, HWl, nullptr, 1, params, grads, ctx,
                  st));
  RUN(signal(0));
  // blocks in reverse; dcur rotates through the three gradient buffers
  int cur_i = 1;
  int stage = 4, left = n->layers[3];   // blocks of the current stage still to go
  bool chained = false;                 // bn3 of the current block already has its reduction
  WgradQueue wq{n, grads, &sc, st};
  for (int k = (int)n->blocks.size() - 1; k >= 0; --k) {
    const bool chain_prev = g_bn_chain && k > 0;
    RUN(backward_block(n, k, cur_i, chained, chain_prev, params, grads, ctx, wc, sc, wq, st));
    chained = chain_prev;
    cur_i = (cur_i + 1) % 3;   // t1 is the new dcur
    if (--left == 0 && stage > 1) {   // stage 4, 3, 2 complete -> buckets 1, 2, 3 (stage 1 goes with the stem)
      if (stage_events) RUN(wq.flush());   // somebody waits on the event: the bucket's weight gradients are enqueued in front of it
      RUN(signal(5 - stage));
      --stage;
      left = n->layers[stage - 1];
    }
  }
  RUN(wq.flush());
  // stem
  char* dc0 = sc.g[(cur_i + 1) % 3];
  RUN(bn_backward(n->stem_bn, sc.g[cur_i], ctx + n->off_c0, dc0, (int64_t)B * S * S, S * S, nullptr, 0, params, grads,
                  ctx, st));
  RUN(vlsfr_zero_bytes(sc.stem_dw, 64 * 32 * 4, st));
  RUN(vlsfr_conv2d_wgrad_ws(&n->stem.d, dc0, ctx + n->off_cols, sc.stem_dw, 0, sc.wgrad_ws, n->wgrad_ws, st));
  RUN(vlsfr_unpad_add(sc.stem_dw, grads[n->stem.p_w], 64, 32, 27, st));
  return signal(4);
}

// ---- teacher-forced execution of a range of blocks (parity tests of the executor wiring) --------------------------------
// vlsfr_iresnet_forward_blocks runs IBasicBlock k0 .. k1 - 1 exactly as vlsfr_iresnet_forward does (same kernels, same
// context slots), but from an activation handed in by the caller instead of the output of the layers in front; the
// backward twin starts from a caller-supplied output gradient.  One block does not amplify rounding noise the way 49
// stacked train-mode BatchNorm blocks do, so residual / downsample / PReLU / BatchNorm gradient routing can be held to
// a per-tensor tolerance instead of the whole-network band.
int vlsfr_iresnet_block_info(const vlsfr_iresnet* n, int32_t k, int32_t* info /*[8]*/) {
  if (!n || !info || k < 0 || k >= (int)n->blocks.size()) return fail(VLSFR_EINVAL, "vlsfr_iresnet_block_info: bad argument");
  const Block& b = n->blocks[k];
  const int32_t v[8] = {b.cin, b.planes, b.H, b.Ho, b.stride, b.has_ds ? 1 : 0, b.bn1.p_w, (int32_t)n->blocks.size()};
  std::memcpy(info, v, sizeof(v));
  return VLSFR_OK;
}

int vlsfr_iresnet_forward_blocks(const vlsfr_iresnet* n, int32_t k0, int32_t k1, const void* x_in, const float* const* params,
                                 float* const* running, const void* wcache, void* ctx_v, void* scratch, void* out, void* st) {
  if (!n || !x_in || !params || !wcache || !ctx_v || !scratch || !out || k0 < 0 || k1 <= k0 || k1 > (int)n->blocks.size())
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_forward_blocks: bad argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  hipStream_t s = (hipStream_t)st;
  RUN(vlsfr_zero_bytes(ctx + n->sums_begin, n->sums_end - n->sums_begin, (void*)s));
  hipError_t e = hipSuccess;
  const Block& first = n->blocks[k0];
  char* cur = k0 > 0 ? ctx + n->blocks[k0 - 1].out : ctx + n->off_a0;
  const int64_t Min = (int64_t)n->B * first.H * first.W;
  if (e == hipSuccess) e = hipMemcpyAsync(cur, x_in, (size_t)Min * first.cin * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_forward_blocks: %s", hipGetErrorString(e));
  RUN(vlsfr_bn_stats(cur, Min, first.cin, (double*)(ctx + first.bn1.off_sums), st));
  for (int k = k0; k < k1; ++k) {
    RUN(forward_block(n, k, cur, params, running, ctx, wc, sc, st));
    cur = ctx + n->blocks[k].out;
  }
  const Block& lastb = n->blocks[k1 - 1];
  e = hipMemcpyAsync(out, cur, (size_t)n->B * lastb.Ho * lastb.Wo * lastb.planes * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_forward_blocks: copy: %s", hipGetErrorString(e));
  return VLSFR_OK;
}

int vlsfr_iresnet_backward_blocks(const vlsfr_iresnet* n, int32_t k0, int32_t k1, const void* dout, const float* const* params,
                                  float* const* grads, const void* wcache, void* ctx_v, void* scratch, void* dx, void* st) {
  if (!n || !dout || !params || !grads || !wcache || !ctx_v || !scratch || !dx || k0 < 0 || k1 <= k0 || k1 > (int)n->blocks.size())
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_backward_blocks: bad argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  hipStream_t s = (hipStream_t)st;
  RUN(vlsfr_zero_bytes(ctx + n->red_begin, n->red_end - n->red_begin, (void*)s));
  hipError_t e = hipSuccess;
  const Block& lastb = n->blocks[k1 - 1];
  int cur_i = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(sc.g[cur_i], dout, (size_t)n->B * lastb.Ho * lastb.Wo * lastb.planes * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_backward_blocks: %s", hipGetErrorString(e));
  bool chained = false;
  WgradQueue wq{n, grads, &sc, st};
  for (int k = k1 - 1; k >= k0; --k) {
    const bool chain_prev = g_bn_chain && k > k0;     // inside the range only: block k0 - 1 did not run
    RUN(backward_block(n, k, cur_i, chained, chain_prev, params, grads, ctx, wc, sc, wq, st));
    chained = chain_prev;
    cur_i = (cur_i + 1) % 3;
  }
  RUN(wq.flush());
  const Block& first = n->blocks[k0];
  e = hipMemcpyAsync(dx, sc.g[cur_i], (size_t)n->B * first.H * first.W * first.cin * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_backward_blocks: copy: %s", hipGetErrorString(e));
  return VLSFR_OK;
}

// ---- backward pass with the weight gradients on a second stream -------------------------------------------------
// The input-gradient chain (BatchNorm backward -> dgrad -> BatchNorm backward -> ...) is a strict dependency chain that
// alternates MFMA-bound and HBM-bound kernels; the weight gradient of a layer depends only on that layer's dY and its
// saved input, and nothing in the chain depends on it.  Here every gradient tensor of the chain takes the next slot of
// a ring of OVERLAP_NR buffers instead of rotating through three, and the weight-gradient kernels go to `side_stream`:
// they may lag the chain by up to a block and run under its BatchNorm kernels.  Ordering: event ev_main[s] (recorded on
// the main stream after the kernel that writes slot s) gates the side-stream reader; event ev_side[s] (recorded after the
// reader) gates the next main-stream writer of slot s.  events = [ev_main[0..NR) | ev_side[0..NR) | join].
namespace {
constexpr int OVERLAP_NR = 10;   // a block writes 5 (6 with a shortcut conv) tensors and its input gradient is live to the end
struct Ring {
  char* base;
  size_t stride;
  hipStream_t main, side;
  hipEvent_t* ev_main;
  hipEvent_t* ev_side;
  bool pending[OVERLAP_NR];
  int next;
  hipError_t err;
  char* slot(int s) const { return base + (size_t)s * stride; }
  int acquire() {   // next slot, safe to write from the main stream
    const int s = next;
    next = (next + 1) % OVERLAP_NR;
    if (pending[s]) {
      hipError_t e = hipStreamWaitEvent(main, ev_side[s], 0);
      if (e != hipSuccess) err = e;
      pending[s] = false;
    }
    return s;
  }
  void produced(int s) {   // the kernel writing slot s has been enqueued on the main stream
    hipError_t e = hipEventRecord(ev_main[s], main);
    if (e != hipSuccess) err = e;
  }
  void side_begin(int s) {
    hipError_t e = hipStreamWaitEvent(side, ev_main[s], 0);
    if (e != hipSuccess) err = e;
  }
  void side_end(int s) {
    hipError_t e = hipEventRecord(ev_side[s], side);
    if (e != hipSuccess) err = e;
    pending[s] = true;
  }
};
}  // namespace

size_t vlsfr_iresnet_overlap_ring_bytes(const vlsfr_iresnet* n) { return n ? (size_t)OVERLAP_NR * align_up(n->max_act) : 0; }
int32_t vlsfr_iresnet_overlap_events(void) { return 2 * OVERLAP_NR + 1; }

int vlsfr_iresnet_backward_overlap(const vlsfr_iresnet* n, const float* demb, const float* const* params,
                                   float* const* grads, const void* wcache, void* ctx_v, void* scratch, void* ring_v,
                                   size_t ring_bytes, void* const* stage_events, void* side_stream, void* const* events,
                                   void* st) {
  if (!n || !demb || !params || !grads || !wcache || !ctx_v || !scratch || !ring_v || !side_stream || !events)
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_backward_overlap: null argument");
  if (ring_bytes < vlsfr_iresnet_overlap_ring_bytes(n))
    return fail(VLSFR_EINVAL, "vlsfr_iresnet_backward_overlap: ring of %zu bytes, %zu needed", ring_bytes,
                vlsfr_iresnet_overlap_ring_bytes(n));
  for (int i = 0; i < 2 * OVERLAP_NR + 1; ++i)
    if (!events[i]) return fail(VLSFR_EINVAL, "vlsfr_iresnet_backward_overlap: null event %d", i);
  hipStream_t sm = (hipStream_t)st, ss = (hipStream_t)side_stream;
  Ring R;
  R.base = (char*)ring_v;
  R.stride = align_up(n->max_act);
  R.main = sm;
  R.side = ss;
  R.ev_main = (hipEvent_t*)events;
  R.ev_side = (hipEvent_t*)events + OVERLAP_NR;
  for (bool& p : R.pending) p = false;
  R.next = 0;
  R.err = hipSuccess;
  hipEvent_t ev_join = (hipEvent_t)events[2 * OVERLAP_NR];
  auto join = [&]() -> int {   // the main stream continues once every weight gradient enqueued so far is done
    hipError_t e = hipEventRecord(ev_join, ss);
    if (e == hipSuccess) e = hipStreamWaitEvent(sm, ev_join, 0);
    return e == hipSuccess ? VLSFR_OK : fail(VLSFR_EHIP, "vlsfr_iresnet_backward_overlap: join: %s", hipGetErrorString(e));
  };
  auto signal = [&](int k) -> int {
    if (!stage_events || !stage_events[k]) return VLSFR_OK;
    int rc = join();
    if (rc) return rc;
    hipError_t ee = hipEventRecord((hipEvent_t)stage_events[k], sm);
    return ee == hipSuccess ? VLSFR_OK : fail(VLSFR_EHIP, "vlsfr_iresnet_backward_overlap: hipEventRecord: %s", hipGetErrorString(ee));
  };
  // the weight gradient of `c` from dY in ring slot s and the saved input x, on the side stream
  auto wgrad_side = [&](const Conv& c, int s, const void* x) -> int {
    R.side_begin(s);
    int rc = vlsfr_conv2d_wgrad(&c.d, R.slot(s), x, grads[c.p_w], 0, ss);
    R.side_end(s);
    return rc;
  };
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  const int B = n->B, S = n->HW0;
  const Block& last = n->blocks.back();
  const int HWl = last.Ho * last.Wo;
  RUN(vlsfr_zero_bytes(ctx + n->red_begin, n->red_end - n->red_begin, (void*)sm));
  {   // the side stream starts behind everything the main stream has done so far (forward pass, head, zeroed gradients)
    hipError_t e = hipEventRecord(ev_join, sm);
    if (e == hipSuccess) e = hipStreamWaitEvent(ss, ev_join, 0);
    if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_backward_overlap: fork: %s", hipGetErrorString(e));
  }
  RUN(vlsfr_embed_bwd(demb, (const float*)(ctx + n->off_emb), (const float*)(ctx + n->off_invnorm),
                      (const float*)(ctx + n->off_xhat), (const float*)(ctx + n->off_feat_invstd),
                      params[n->p_feat_w], sc.dz, sc.dfc, grads[n->p_feat_b], grads[n->p_fc_b], nullptr, B, n->D, sm));
  RUN(vlsfr_conv2d_wgrad(&n->fc.d, sc.dfc, ctx + n->off_flat, grads[n->fc.p_w], 0, sm));
  const int s_flat = R.acquire();
  RUN(vlsfr_conv2d_dgrad(&n->fc.d, sc.dfc, wc + n->fc.off_wT, R.slot(s_flat), sm));
  int s_cur = R.acquire();
  RUN(bn_backward(n->bn_last, R.slot(s_flat), ctx + last.out, R.slot(s_cur), (int64_t)B * HWl, HWl, nullptr, 1, params, grads,
                  ctx, sm));
  RUN(signal(0));
  int stage = 4, left = n->layers[3];
  for (int k = (int)n->blocks.size() - 1; k >= 0; --k) {
    const Block& b = n->blocks[k];
    const char* x_in = k > 0 ? ctx + n->blocks[k - 1].out : ctx + n->off_a0;
    const int64_t Min = (int64_t)B * b.H * b.W, Mout = (int64_t)B * b.Ho * b.Wo;
    const char* dout = R.slot(s_cur);
    const int s_c2 = R.acquire();
    RUN(bn_backward(b.bn3, dout, ctx + b.c2, R.slot(s_c2), Mout, b.Ho * b.Wo, nullptr, 0, params, grads, ctx, sm));
    R.produced(s_c2);
    RUN(wgrad_side(b.conv2, s_c2, ctx + b.a2));
    const int s_a2 = R.acquire();
    RUN(vlsfr_conv2d_dgrad(&b.conv2.d, R.slot(s_c2), wc + b.conv2.off_wT, R.slot(s_a2), sm));
    const int s_c1 = R.acquire();
    RUN(bn_backward(b.bn2, R.slot(s_a2), ctx + b.c1, R.slot(s_c1), Min, b.H * b.W, nullptr, 0, params, grads, ctx, sm));
    R.produced(s_c1);
    RUN(wgrad_side(b.conv1, s_c1, ctx + b.a1));
    const int s_a1 = R.acquire();
    RUN(vlsfr_conv2d_dgrad(&b.conv1.d, R.slot(s_c1), wc + b.conv1.off_wT, R.slot(s_a1), sm));
    const char* add = dout;
    if (b.has_ds) {
      const int s_cs = R.acquire();
      RUN(bn_backward(b.bnd, dout, ctx + b.cs, R.slot(s_cs), Mout, b.Ho * b.Wo, nullptr, 0, params, grads, ctx, sm));
      R.produced(s_cs);
      RUN(wgrad_side(b.convd, s_cs, x_in));
      RUN(vlsfr_conv2d_dgrad(&b.convd.d, R.slot(s_cs), wc + b.convd.off_wT, sc.idn, sm));
      add = sc.idn;
    }
    const int s_x = R.acquire();
    RUN(bn_backward(b.bn1, R.slot(s_a1), x_in, R.slot(s_x), Min, b.H * b.W, add, 0, params, grads, ctx, sm));
    s_cur = s_x;
    if (R.err != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_backward_overlap: event: %s", hipGetErrorString(R.err));
    if (--left == 0 && stage > 1) {
      RUN(signal(5 - stage));
      --stage;
      left = n->layers[stage - 1];
    }
  }
  const int s_c0 = R.acquire();
  RUN(bn_backward(n->stem_bn, R.slot(s_cur), ctx + n->off_c0, R.slot(s_c0), (int64_t)B * S * S, S * S, nullptr, 0, params, grads,
                  ctx, sm));
  RUN(vlsfr_zero_bytes(sc.stem_dw, 64 * 32 * 4, (void*)sm));
  RUN(vlsfr_conv2d_wgrad(&n->stem.d, R.slot(s_c0), ctx + n->off_cols, sc.stem_dw, 0, sm));
  RUN(vlsfr_unpad_add(sc.stem_dw, grads[n->stem.p_w], 64, 32, 27, sm));
  RUN(join());
  if (stage_events && stage_events[4]) {
    hipError_t ee = hipEventRecord((hipEvent_t)stage_events[4], sm);
    if (ee != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_iresnet_backward_overlap: hipEventRecord: %s", hipGetErrorString(ee));
  }
  return VLSFR_OK;
}

}  // extern "C"
