// Native executor of the torchvision-style ResNet of the reference (C-ABI section 7c of include/vlsfr.h):
// model/resnet_std.py:55-104 (Bottleneck: 1x1 -> BN -> ReLU -> 3x3(stride) -> BN -> ReLU -> 1x1 -> BN, + shortcut,
// ReLU) and :106-206 (7x7/2 stem, BN, ReLU, 3x3/2 max-pool, four stages, flatten -> fc -> BatchNorm1d -> L2
// normalise), training mode — `--net_type r50`, the reference's default (main.py:152).  Same contract as the iResNet
// executor: one call enqueues a whole forward or backward pass on the caller's stream; parameters in the
// registration order of the reference module; conv weights fp32 in channels_last memory; gradients accumulated.
// ReLU runs through the PReLU path of the BatchNorm kernels with a zero slope.
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "common_host.h"

namespace {

constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOM = 0.1f;

struct Bn {
  int C, p_w, p_b, run;
  size_t off_sums, off_mean, off_invstd, off_red;
};
struct Conv {
  vlsfr_conv_desc d;
  int p_w;
  size_t off_wb, off_wT;
};
struct Block {
  int cin, width, cout, stride, H, Ho;
  Conv conv1, conv2, conv3, convd;
  Bn bn1, bn2, bn3, bnd;
  bool has_ds;
  size_t c1, a1, c2, a2, c3, cs, out;   // ctx offsets (bf16 activations)
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

struct vlsfr_resnet {
  int layers[4];
  int D, B, S;
  int n_params = 0, n_bn = 0;
  Conv stem;
  Bn stem_bn;
  size_t off_cols, off_c0, off_a0, off_m0, off_zero_slope;
  int Hs, Hp;   // stem output and pooled sizes
  std::vector<Block> blocks;
  Conv fc;
  int p_fc_b, p_feat_w, p_feat_b, run_feat;
  size_t off_fcout, off_z, off_xhat, off_feat_invstd, off_emb, off_invnorm;
  size_t sums_begin, sums_end, red_begin, red_end;
  size_t ctx_bytes = 0, wcache_bytes = 0, scratch_bytes = 0, max_act = 0, wgrad_ws = 0;

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
  Bn make_bn(int C) {
    Bn b;
    b.C = C;
    b.p_w = n_params++;
    b.p_b = n_params++;
    b.run = n_bn++;
    b.off_sums = b.off_mean = b.off_invstd = b.off_red = 0;
    return b;
  }
  Conv make_conv(int H, int cin, int cout, int k, int stride) {
    Conv c;
    c.d = vlsfr_conv_desc{B, H, H, cin, cout, k, k, stride, k == 3 ? 1 : 0};
    c.p_w = n_params++;
    const size_t bytes = (size_t)cout * k * k * cin * 2;
    c.off_wb = take_w(bytes);
    c.off_wT = take_w(bytes);
    return c;
  }
};

namespace {

using vlsfr::fail;

int build(vlsfr_resnet* n) {
  const int B = n->B, S = n->S;
  n->Hs = (S + 6 - 7) / 2 + 1;
  n->Hp = (n->Hs + 2 - 3) / 2 + 1;
  // registration order: conv1, bn1, layer1.0.{conv1, bn1, conv2, bn2, conv3, bn3, downsample.0, downsample.1}, ...
  n->stem.d = vlsfr_conv_desc{B, n->Hs, n->Hs, 160, 64, 1, 1, 1, 0};   // GEMM over the im2col rows (147 taps -> 160)
  n->stem.p_w = n->n_params++;
  n->stem.off_wb = n->take_w((size_t)64 * 160 * 2);
  n->stem.off_wT = 0;
  n->stem_bn = n->make_bn(64);
  int cin = 64, H = n->Hp;
  const int planes_of[4] = {64, 128, 256, 512};
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < n->layers[li]; ++bi) {
      Block b;
      b.cin = cin;
      b.width = planes_of[li];
      b.cout = 4 * planes_of[li];
      b.stride = (bi == 0 && li > 0) ? 2 : 1;
      b.H = H;
      b.Ho = (H + 2 - 3) / b.stride + 1;
      b.conv1 = n->make_conv(H, cin, b.width, 1, 1);
      b.bn1 = n->make_bn(b.width);
      b.conv2 = n->make_conv(H, b.width, b.width, 3, b.stride);
      b.bn2 = n->make_bn(b.width);
      b.conv3 = n->make_conv(b.Ho, b.width, b.cout, 1, 1);
      b.bn3 = n->make_bn(b.cout);
      b.has_ds = bi == 0;   // stride != 1 or inplanes != planes * 4 (resnet_std.py:176-180): the first block of every stage
      if (b.has_ds) {
        b.convd = n->make_conv(H, cin, b.cout, 1, b.stride);
        b.bnd = n->make_bn(b.cout);
      }
      n->blocks.push_back(b);
      cin = b.cout;
      H = b.Ho;
    }
  const int Kfc = cin * H * H;   // 2048 * 7 * 7 at 224 x 224
  n->fc.d = vlsfr_conv_desc{B, 1, 1, Kfc, n->D, 1, 1, 1, 0};
  n->fc.p_w = n->n_params++;
  n->fc.off_wb = n->take_w((size_t)n->D * Kfc * 2);
  n->fc.off_wT = n->take_w((size_t)n->D * Kfc * 2);
  n->p_fc_b = n->n_params++;
  n->p_feat_w = n->n_params++;
  n->p_feat_b = n->n_params++;
  n->run_feat = n->n_bn++;

  // ctx: statistics (+ the zero slope vector and the split-K fc accumulator) first, one memset clears them
  n->sums_begin = n->ctx_bytes;
  auto sums = [&](Bn& b) { b.off_sums = n->take_ctx((size_t)VLSFR_BN_REPL * 2 * b.C * 8); };
  auto each_bn = [&](auto&& f) {
    f(n->stem_bn);
    for (auto& b : n->blocks) {
      f(b.bn1);
      f(b.bn2);
      f(b.bn3);
      if (b.has_ds) f(b.bnd);
    }
  };
  each_bn(sums);
  n->off_zero_slope = n->take_ctx((size_t)2048 * 4);
  n->off_fcout = n->take_ctx((size_t)B * n->D * 4);
  n->sums_end = n->ctx_bytes;
  n->red_begin = n->ctx_bytes;
  each_bn([&](Bn& b) { b.off_red = n->take_ctx((size_t)VLSFR_BN_REPL * 3 * b.C * 4); });
  n->red_end = n->ctx_bytes;
  each_bn([&](Bn& b) {
    b.off_mean = n->take_ctx((size_t)b.C * 4);
    b.off_invstd = n->take_ctx((size_t)b.C * 4);
  });
  const size_t Ps = (size_t)B * n->Hs * n->Hs, Pp = (size_t)B * n->Hp * n->Hp;
  n->off_cols = n->take_ctx(Ps * 160 * 2);
  n->off_c0 = n->take_ctx(Ps * 64 * 2);
  n->off_a0 = n->take_ctx(Ps * 64 * 2);
  n->off_m0 = n->take_ctx(Pp * 64 * 2);
  n->max_act = Ps * 64 * 2;
  for (auto& b : n->blocks) {
    const size_t pin = (size_t)B * b.H * b.H, pout = (size_t)B * b.Ho * b.Ho;
    b.c1 = n->take_ctx(pin * b.width * 2);
    b.a1 = n->take_ctx(pin * b.width * 2);
    b.c2 = n->take_ctx(pout * b.width * 2);
    b.a2 = n->take_ctx(pout * b.width * 2);
    b.c3 = n->take_ctx(pout * b.cout * 2);
    b.cs = b.has_ds ? n->take_ctx(pout * b.cout * 2) : 0;
    b.out = n->take_ctx(pout * b.cout * 2);
    const size_t m = pin * (size_t)(b.cin > b.width ? b.cin : b.width) * 2;
    if (m > n->max_act) n->max_act = m;
    if (pout * b.cout * 2 > n->max_act) n->max_act = pout * b.cout * 2;
  }
  n->off_z = n->take_ctx((size_t)B * n->D * 4);
  n->off_xhat = n->take_ctx((size_t)B * n->D * 4);
  n->off_feat_invstd = n->take_ctx((size_t)n->D * 4);
  n->off_emb = n->take_ctx((size_t)B * n->D * 4);
  n->off_invnorm = n->take_ctx((size_t)B * 4);
  // scratch: 4 activation-sized gradient buffers + the shortcut tensor of the forward pass + small fp32 areas
  auto ws_of = [&](const Conv& c) {
    const size_t w = vlsfr_conv2d_wgrad_workspace_bytes(&c.d, 0);   // split-K slabs (vlsfr_conv2d_wgrad_ws)
    if (w > n->wgrad_ws) n->wgrad_ws = w;
  };
  ws_of(n->stem);
  for (auto& b : n->blocks) {
    ws_of(b.conv1);
    ws_of(b.conv2);
    ws_of(b.conv3);
    if (b.has_ds) ws_of(b.convd);
  }
  ws_of(n->fc);
  n->scratch_bytes = 5 * align_up(n->max_act) + align_up((size_t)2048 * 4) + align_up((size_t)64 * 160 * 4) +
                     align_up((size_t)B * n->D * 4) + align_up((size_t)B * n->D * 2) + align_up(n->wgrad_ws);
  return VLSFR_OK;
}

struct Scratch {
  char* g[4];
  char* idn;
  float* dslope;    // sink of the (unused) slope gradient of the ReLU-as-PReLU layers
  float* stem_dw;
  float* dz;
  char* dfc;
  void* wgrad_ws;
};
Scratch carve(const vlsfr_resnet* n, void* scratch) {
  Scratch s;
  char* p = (char*)scratch;
  const size_t a = align_up(n->max_act);
  for (int i = 0; i < 4; ++i) s.g[i] = p + i * a;
  s.idn = p + 4 * a;
  p += 5 * a;
  s.dslope = (float*)p;
  p += align_up((size_t)2048 * 4);
  s.stem_dw = (float*)p;
  p += align_up((size_t)64 * 160 * 4);
  s.dz = (float*)p;
  p += align_up((size_t)n->B * n->D * 4);
  s.dfc = p;
  p += align_up((size_t)n->B * n->D * 2);
  s.wgrad_ws = p;
  return s;
}

#define RUN(expr)                      \
  do {                                 \
    int rc__ = (expr);                 \
    if (rc__ != VLSFR_OK) return rc__; \
  } while (0)

// y = [relu](bn(x)) [+ residual, relu after]: flags bit 0 = ReLU (zero-slope PReLU), bit 1 = ReLU after the add, bit 2 = NCHW output
int bn_forward(const vlsfr_resnet* n, const Bn& b, const void* x, void* y, int64_t M, int HW, int relu, const void* residual,
               int out_flags, const float* const* params, float* const* running, char* ctx, void* st) {
  float* rm = running ? running[2 * b.run] : nullptr;
  float* rv = running ? running[2 * b.run + 1] : nullptr;
  return vlsfr_bn_apply(x, y, M, b.C, HW, (const double*)(ctx + b.off_sums), params[b.p_w], params[b.p_b],
                        relu ? (const float*)(ctx + n->off_zero_slope) : nullptr, residual, (float*)(ctx + b.off_mean),
                        (float*)(ctx + b.off_invstd), rm, rv, BN_EPS, BN_MOM, nullptr, out_flags, st);
}

int bn_backward(const vlsfr_resnet* n, const Bn& b, const void* dy, const void* x, void* dx, int64_t M, int HW, int relu,
                const float* const* params, float* const* grads, char* ctx, float* dslope_sink, void* st) {
  return vlsfr_bn_backward(dy, x, dx, M, b.C, HW, (const float*)(ctx + b.off_mean), (const float*)(ctx + b.off_invstd),
                           params[b.p_w], params[b.p_b], relu ? (const float*)(ctx + n->off_zero_slope) : nullptr,
                           (float*)(ctx + b.off_red), nullptr, grads[b.p_w], grads[b.p_b], relu ? dslope_sink : nullptr, 0, st);
}

}  // namespace

extern "C" {

int vlsfr_resnet_create(const int32_t* layers, int32_t feat_dim, int32_t batch, int32_t image_hw, vlsfr_resnet** out) {
  if (!layers || !out || feat_dim <= 0 || feat_dim % 8 || batch <= 0 || image_hw < 32 || image_hw % 32)
    return fail(VLSFR_EINVAL, "vlsfr_resnet_create: need feat_dim %% 8 == 0 and image size %% 32 == 0");
  for (int i = 0; i < 4; ++i)
    if (layers[i] < 1) return fail(VLSFR_EINVAL, "vlsfr_resnet_create: every stage needs at least one block");
  vlsfr_resnet* n = new (std::nothrow) vlsfr_resnet();
  if (!n) return fail(VLSFR_ENOMEM, "vlsfr_resnet_create: out of memory");
  std::memcpy(n->layers, layers, sizeof(n->layers));
  n->D = feat_dim;
  n->B = batch;
  n->S = image_hw;
  int rc = build(n);
  if (rc != VLSFR_OK) {
    delete n;
    return rc;
  }
  *out = n;
  return VLSFR_OK;
}
void vlsfr_resnet_destroy(vlsfr_resnet* n) { delete n; }
int32_t vlsfr_resnet_num_params(const vlsfr_resnet* n) { return n ? n->n_params : -1; }
int32_t vlsfr_resnet_num_bn(const vlsfr_resnet* n) { return n ? n->n_bn : -1; }
size_t vlsfr_resnet_wcache_bytes(const vlsfr_resnet* n) { return n ? n->wcache_bytes : 0; }
size_t vlsfr_resnet_ctx_bytes(const vlsfr_resnet* n) { return n ? n->ctx_bytes : 0; }
size_t vlsfr_resnet_scratch_bytes(const vlsfr_resnet* n) { return n ? n->scratch_bytes : 0; }

int vlsfr_resnet_prepare_weights(const vlsfr_resnet* n, const float* const* params, void* wcache, void* st) {
  if (!n || !params || !wcache) return fail(VLSFR_EINVAL, "vlsfr_resnet_prepare_weights: null argument");
  char* wc = (char*)wcache;
  std::vector<vlsfr_cast_entry> tab;
  tab.push_back({params[n->stem.p_w], wc + n->stem.off_wb, nullptr, 64, 1, 147, 160});
  auto add = [&](const Conv& c) {
    tab.push_back({params[c.p_w], wc + c.off_wb, wc + c.off_wT, c.d.Cout, c.d.R * c.d.S, c.d.Cin, c.d.R * c.d.S * c.d.Cin});
  };
  for (const auto& b : n->blocks) {
    add(b.conv1);
    add(b.conv2);
    add(b.conv3);
    if (b.has_ds) add(b.convd);
  }
  add(n->fc);
  return vlsfr_cast_weights(tab.data(), (int32_t)tab.size(), st);
}

int vlsfr_resnet_forward(const vlsfr_resnet* n, const float* x_nchw, const float* const* params, float* const* running,
                         const void* wcache, void* ctx_v, void* scratch, float* emb_out, void* st) {
  if (!n || !x_nchw || !params || !wcache || !ctx_v || !scratch || !emb_out)
    return fail(VLSFR_EINVAL, "vlsfr_resnet_forward: null argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  RUN(vlsfr_zero_bytes(ctx + n->sums_begin, n->sums_end - n->sums_begin, st));
  hipError_t e = hipSuccess;
  const int B = n->B;
  auto sums_of = [&](const Bn& b) { return (double*)(ctx + b.off_sums); };
  // stem (resnet_std.py:186-189): 7x7/2 conv -> BN -> ReLU -> 3x3/2 max-pool
  RUN(vlsfr_stem7_im2col(x_nchw, ctx + n->off_cols, B, n->S, n->S, st));
  RUN(vlsfr_conv2d_fwd(&n->stem.d, ctx + n->off_cols, wc + n->stem.off_wb, ctx + n->off_c0, 1, 0, sums_of(n->stem_bn), st));
  RUN(bn_forward(n, n->stem_bn, ctx + n->off_c0, ctx + n->off_a0, (int64_t)B * n->Hs * n->Hs, n->Hs * n->Hs, 1, nullptr, 0,
                 params, running, ctx, st));
  RUN(vlsfr_maxpool3x3s2_fwd(ctx + n->off_a0, ctx + n->off_m0, B, n->Hs, n->Hs, 64, st));
  const char* cur = ctx + n->off_m0;
  for (size_t k = 0; k < n->blocks.size(); ++k) {   // Bottleneck.forward, resnet_std.py:82-104
    const Block& b = n->blocks[k];
    const bool last = k + 1 == n->blocks.size();
    const int64_t Min = (int64_t)B * b.H * b.H, Mout = (int64_t)B * b.Ho * b.Ho;
    RUN(vlsfr_conv2d_fwd(&b.conv1.d, cur, wc + b.conv1.off_wb, ctx + b.c1, 1, 0, sums_of(b.bn1), st));
    RUN(bn_forward(n, b.bn1, ctx + b.c1, ctx + b.a1, Min, b.H * b.H, 1, nullptr, 0, params, running, ctx, st));
    RUN(vlsfr_conv2d_fwd(&b.conv2.d, ctx + b.a1, wc + b.conv2.off_wb, ctx + b.c2, 1, 0, sums_of(b.bn2), st));
    RUN(bn_forward(n, b.bn2, ctx + b.c2, ctx + b.a2, Mout, b.Ho * b.Ho, 1, nullptr, 0, params, running, ctx, st));
    RUN(vlsfr_conv2d_fwd(&b.conv3.d, ctx + b.a2, wc + b.conv3.off_wb, ctx + b.c3, 1, 0, sums_of(b.bn3), st));
    const void* idn = cur;
    if (b.has_ds) {
      RUN(vlsfr_conv2d_fwd(&b.convd.d, cur, wc + b.convd.off_wb, ctx + b.cs, 1, 0, sums_of(b.bnd), st));
      RUN(bn_forward(n, b.bnd, ctx + b.cs, sc.idn, Mout, b.Ho * b.Ho, 0, nullptr, 0, params, running, ctx, st));
      idn = sc.idn;
    }
    // out = relu(bn3(c3) + identity); the last block writes the [n][c][hw] flatten order fc reads (resnet_std.py:199)
    RUN(bn_forward(n, b.bn3, ctx + b.c3, ctx + b.out, Mout, b.Ho * b.Ho, 0, idn, 2 | (last ? 1 : 0), params, running, ctx, st));
    cur = ctx + b.out;
  }
  const int Kfc = n->fc.d.Cin;
  int splitk = (Kfc / 32) / 12;
  if (splitk < 1) splitk = 1;
  if (splitk > 64) splitk = 64;
  RUN(vlsfr_conv2d_fwd(&n->fc.d, cur, wc + n->fc.off_wb, ctx + n->off_fcout, splitk, 1, nullptr, st));
  float* rm = running ? running[2 * n->run_feat] : nullptr;
  float* rv = running ? running[2 * n->run_feat + 1] : nullptr;
  RUN(vlsfr_embed_fwd((const float*)(ctx + n->off_fcout), params[n->p_fc_b], params[n->p_feat_w], params[n->p_feat_b], rm, rv,
                      (float*)(ctx + n->off_z), (float*)(ctx + n->off_xhat), (float*)(ctx + n->off_feat_invstd),
                      (float*)(ctx + n->off_emb), (float*)(ctx + n->off_invnorm), B, n->D, BN_EPS, BN_MOM, st));
  (void)e;
  return vlsfr_copy_bytes(ctx + n->off_emb, emb_out, (size_t)B * n->D * 4, st);   // (a kernel: captured passes hold no runtime copy nodes)
}

int vlsfr_resnet_backward(const vlsfr_resnet* n, const float* demb, const float* const* params, float* const* grads,
                          const void* wcache, void* ctx_v, void* scratch, void* st) {
  if (!n || !demb || !params || !grads || !wcache || !ctx_v || !scratch)
    return fail(VLSFR_EINVAL, "vlsfr_resnet_backward: null argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  const int B = n->B;
  RUN(vlsfr_zero_bytes(ctx + n->red_begin, n->red_end - n->red_begin, st));
  const Block& lastb = n->blocks.back();
  // embedding tail (features.weight is trainable here: resnet_std.py:143 does not freeze it), fc
  RUN(vlsfr_embed_bwd(demb, (const float*)(ctx + n->off_emb), (const float*)(ctx + n->off_invnorm),
                      (const float*)(ctx + n->off_xhat), (const float*)(ctx + n->off_feat_invstd), params[n->p_feat_w], sc.dz,
                      sc.dfc, grads[n->p_feat_b], grads[n->p_fc_b], grads[n->p_feat_w], B, n->D, st));
  RUN(vlsfr_conv2d_wgrad_ws(&n->fc.d, sc.dfc, ctx + lastb.out, grads[n->fc.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
  int cur = 0;   // index of the buffer holding d(out of block k)
  RUN(vlsfr_conv2d_dgrad(&n->fc.d, sc.dfc, wc + n->fc.off_wT, sc.g[cur], st));   // [n][c][hw] order, like the saved output
  for (int k = (int)n->blocks.size() - 1; k >= 0; --k) {
    const Block& b = n->blocks[k];
    const bool last = k + 1 == (int)n->blocks.size();
    const char* x_in = k > 0 ? ctx + n->blocks[k - 1].out : ctx + n->off_m0;
    const int64_t Min = (int64_t)B * b.H * b.H, Mout = (int64_t)B * b.Ho * b.Ho;
    char* t[3];
    for (int i = 0, j = 0; i < 4; ++i)
      if (i != cur) t[j++] = sc.g[i];
    char* dte = t[0];
    // d(bn3 + identity) = d(out) where out > 0
    RUN(vlsfr_relu_bwd_bf16(sc.g[cur], ctx + b.out, dte, Mout, b.cout, b.Ho * b.Ho, last ? 1 : 0, st));
    char* u = sc.g[cur];   // d(out) is consumed: its buffer is free again
    RUN(bn_backward(n, b.bn3, dte, ctx + b.c3, t[1], Mout, b.Ho * b.Ho, 0, params, grads, ctx, sc.dslope, st));       // d c3
    RUN(vlsfr_conv2d_wgrad_ws(&b.conv3.d, t[1], ctx + b.a2, grads[b.conv3.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
    RUN(vlsfr_conv2d_dgrad(&b.conv3.d, t[1], wc + b.conv3.off_wT, t[2], st));                                          // d a2
    RUN(bn_backward(n, b.bn2, t[2], ctx + b.c2, t[1], Mout, b.Ho * b.Ho, 1, params, grads, ctx, sc.dslope, st));        // d c2
    RUN(vlsfr_conv2d_wgrad_ws(&b.conv2.d, t[1], ctx + b.a1, grads[b.conv2.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
    RUN(vlsfr_conv2d_dgrad(&b.conv2.d, t[1], wc + b.conv2.off_wT, t[2], st));                                          // d a1
    RUN(bn_backward(n, b.bn1, t[2], ctx + b.c1, t[1], Min, b.H * b.H, 1, params, grads, ctx, sc.dslope, st));           // d c1
    RUN(vlsfr_conv2d_wgrad_ws(&b.conv1.d, t[1], x_in, grads[b.conv1.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
    RUN(vlsfr_conv2d_dgrad(&b.conv1.d, t[1], wc + b.conv1.off_wT, t[2], st));                                          // d in (main)
    const char* other = dte;                                                                                           // identity branch
    if (b.has_ds) {
      RUN(bn_backward(n, b.bnd, dte, ctx + b.cs, t[1], Mout, b.Ho * b.Ho, 0, params, grads, ctx, sc.dslope, st));       // d cs
      RUN(vlsfr_conv2d_wgrad_ws(&b.convd.d, t[1], x_in, grads[b.convd.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
      RUN(vlsfr_conv2d_dgrad(&b.convd.d, t[1], wc + b.convd.off_wT, u, st));
      other = u;
    }
    RUN(vlsfr_add_bf16(t[2], other, t[2], Min * b.cin, st));
    for (int i = 0; i < 4; ++i)
      if (sc.g[i] == t[2]) cur = i;
  }
  // stem: max-pool, BN + ReLU, 7x7 conv weight gradient
  char* t[3];
  for (int i = 0, j = 0; i < 4; ++i)
    if (i != cur) t[j++] = sc.g[i];
  RUN(vlsfr_maxpool3x3s2_bwd(sc.g[cur], ctx + n->off_a0, ctx + n->off_m0, t[0], B, n->Hs, n->Hs, 64, st));
  RUN(bn_backward(n, n->stem_bn, t[0], ctx + n->off_c0, t[1], (int64_t)B * n->Hs * n->Hs, n->Hs * n->Hs, 1, params, grads, ctx,
                  sc.dslope, st));
  RUN(vlsfr_zero_bytes(sc.stem_dw, 64 * 160 * 4, st));
  RUN(vlsfr_conv2d_wgrad_ws(&n->stem.d, t[1], ctx + n->off_cols, sc.stem_dw, 0, sc.wgrad_ws, n->wgrad_ws, st));
  return vlsfr_unpad_add(sc.stem_dw, grads[n->stem.p_w], 64, 160, 147, st);
}

}  // extern "C"
