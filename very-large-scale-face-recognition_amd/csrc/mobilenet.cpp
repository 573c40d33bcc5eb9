// Native executor of MobileFaceNet (C-ABI section 7b of include/vlsfr.h).  Architecture and
// semantics: reference model/mobilefacenet_def.py:18-25 (bottleneck table t, c, n, s), :27-52
// (BottleNeck: 1x1 -> BN -> PReLU -> depthwise 3x3(stride) -> BN -> PReLU -> 1x1 -> BN, residual when
// stride 1 and equal widths), :55-74 (ConvBlock: conv -> BN [-> PReLU]), :77-123 (stem 3x3 s2,
// depthwise 3x3, 15 bottlenecks, 1x1 128->512, depthwise 7x7 "linear7", 1x1 512->D "linear1",
// flatten, L2 normalise), in training mode.  Parameter / buffer order = registration order of the
// reference module.  Pointwise convolutions run on the MFMA implicit-GEMM kernels, depthwise ones
// on the HBM-bound VALU kernels of dw.hip; every BatchNorm takes its batch statistics from the
// kernel that produced its input.
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "common_host.h"

namespace {

constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOM = 0.1f;
inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

enum Kind { STEM, PW, DW };

// one ConvBlock-like unit: conv -> BN -> (PReLU) (+ residual)
struct Unit {
  Kind kind;
  vlsfr_conv_desc d;      // PW / STEM: dense 1x1 (STEM over the 32-wide im2col rows); DW: depthwise
  int p_w, p_g, p_b, p_slope, run;
  int Ho, Wo;
  int in_unit;            // index of the unit whose output feeds this one (-1: the image)
  int res_unit;           // residual source (output of that unit) or -1
  size_t off_wb, off_wT;  // wcache (PW / STEM)
  size_t c, a;            // ctx: conv output, BN(+PReLU)(+res) output (bf16)
  size_t off_sums, off_mean, off_invstd, off_red;
};

}  // namespace

struct vlsfr_mobilenet {
  int D, B, S;
  int n_params = 0, n_bn = 0;
  std::vector<Unit> units;   // everything up to and including linear7
  // linear1: 1x1 512 -> D on the 1x1 map, then BN over the batch + L2 normalise (embed kernels)
  vlsfr_conv_desc l1d;
  int l1_w, l1_g, l1_b, l1_run;
  size_t l1_wb, l1_wT;
  size_t off_cols, off_fc, off_z, off_xhat, off_invstd, off_emb, off_invnorm, off_zero_bias;
  size_t sums_begin, sums_end, red_begin, red_end;
  size_t ctx_bytes = 0, wcache_bytes = 0, scratch_bytes = 0, max_act = 0, wgrad_ws = 0;
  size_t take_ctx(size_t b) {
    size_t o = ctx_bytes;
    ctx_bytes += align_up(b);
    return o;
  }
  size_t take_w(size_t b) {
    size_t o = wcache_bytes;
    wcache_bytes += align_up(b);
    return o;
  }
};

namespace {
using vlsfr::fail;

#define RUN(expr)                      \
  do {                                 \
    int rc__ = (expr);                 \
    if (rc__ != VLSFR_OK) return rc__; \
  } while (0)

int add_unit(vlsfr_mobilenet* n, Kind kind, int cin, int cout, int k, int stride, int pad, int H, bool prelu, int in_unit,
             int res_unit) {
  Unit u;
  u.kind = kind;
  const int Ho = (H + 2 * pad - k) / stride + 1;
  if (kind == STEM) u.d = vlsfr_conv_desc{n->B, Ho, Ho, 32, cout, 1, 1, 1, 0};       // over im2col rows
  else u.d = vlsfr_conv_desc{n->B, H, H, cin, cout, k, k, stride, pad};
  u.Ho = u.Wo = Ho;
  u.p_w = n->n_params++;
  u.p_g = n->n_params++;
  u.p_b = n->n_params++;
  u.p_slope = prelu ? n->n_params++ : -1;
  u.run = n->n_bn++;
  u.in_unit = in_unit;
  u.res_unit = res_unit;
  u.off_wb = u.off_wT = 0;
  if (kind == STEM) u.off_wb = n->take_w((size_t)cout * 32 * 2);
  if (kind == PW) {
    u.off_wb = n->take_w((size_t)cout * cin * 2);
    u.off_wT = n->take_w((size_t)cout * cin * 2);
  }
  n->units.push_back(u);
  return (int)n->units.size() - 1;
}

int build(vlsfr_mobilenet* n) {
  const int S = n->S;
  int H = S;
  int cur = add_unit(n, STEM, 3, 64, 3, 2, 1, H, true, -1, -1);            // conv1
  H = n->units[cur].Ho;
  cur = add_unit(n, DW, 64, 64, 3, 1, 1, H, true, cur, -1);                // dw_conv1
  const int setting[5][4] = {{2, 64, 5, 2}, {4, 128, 1, 2}, {2, 128, 6, 1}, {4, 128, 1, 2}, {2, 128, 2, 1}};
  int cch = 64;
  for (auto& st : setting) {
    const int t = st[0], c = st[1], reps = st[2], s = st[3];
    for (int i = 0; i < reps; ++i) {
      const int stride = i == 0 ? s : 1;
      const int in = cur;
      const int mid = cch * t;
      int u = add_unit(n, PW, cch, mid, 1, 1, 0, H, true, in, -1);
      u = add_unit(n, DW, mid, mid, 3, stride, 1, H, true, u, -1);
      H = n->units[u].Ho;
      const bool connect = stride == 1 && cch == c;
      cur = add_unit(n, PW, mid, c, 1, 1, 0, H, false, u, connect ? in : -1);
      cch = c;
    }
  }
  cur = add_unit(n, PW, 128, 512, 1, 1, 0, H, true, cur, -1);              // conv2
  if (H != 7) return fail(VLSFR_EINVAL, "mobilefacenet: the 7x7 global depthwise needs a 112x112 input");
  cur = add_unit(n, DW, 512, 512, 7, 1, 0, H, false, cur, -1);             // linear7
  n->l1d = vlsfr_conv_desc{n->B, 1, 1, 512, n->D, 1, 1, 1, 0};             // linear1
  n->l1_w = n->n_params++;
  n->l1_g = n->n_params++;
  n->l1_b = n->n_params++;
  n->l1_run = n->n_bn++;
  n->l1_wb = n->take_w((size_t)n->D * 512 * 2);
  n->l1_wT = n->take_w((size_t)n->D * 512 * 2);

  n->sums_begin = n->ctx_bytes;
  for (auto& u : n->units) u.off_sums = n->take_ctx((size_t)VLSFR_BN_REPL * 2 * u.d.Cout * 8);
  n->off_fc = n->take_ctx((size_t)n->B * n->D * 4);
  n->off_zero_bias = n->take_ctx((size_t)n->D * 4);
  n->sums_end = n->ctx_bytes;
  n->red_begin = n->ctx_bytes;
  for (auto& u : n->units) u.off_red = n->take_ctx((size_t)VLSFR_BN_REPL * 3 * u.d.Cout * 4);
  n->red_end = n->ctx_bytes;
  for (auto& u : n->units) {
    u.off_mean = n->take_ctx((size_t)u.d.Cout * 4);
    u.off_invstd = n->take_ctx((size_t)u.d.Cout * 4);
  }
  n->off_cols = n->take_ctx((size_t)n->B * n->units[0].Ho * n->units[0].Wo * 32 * 2);
  for (auto& u : n->units) {
    const size_t bytes = (size_t)n->B * u.Ho * u.Wo * u.d.Cout * 2;
    u.c = n->take_ctx(bytes);
    u.a = n->take_ctx(bytes);
    if (bytes > n->max_act) n->max_act = bytes;
    const size_t inb = (size_t)n->B * u.d.H * u.d.W * u.d.Cin * 2;
    if (u.kind != STEM && inb > n->max_act) n->max_act = inb;
  }
  n->off_z = n->take_ctx((size_t)n->B * n->D * 4);
  n->off_xhat = n->take_ctx((size_t)n->B * n->D * 4);
  n->off_invstd = n->take_ctx((size_t)n->D * 4);
  n->off_emb = n->take_ctx((size_t)n->B * n->D * 4);
  n->off_invnorm = n->take_ctx((size_t)n->B * 4);
  // scratch: per-unit output gradients are needed until their producers have run (residual fan-out),
  // so keep one gradient buffer per "live" tensor: 4 rotating buffers + small fp32 areas
  for (const auto& u : n->units) {
    // split-K slabs (vlsfr_conv2d_wgrad_ws) / per-block partial sums of the depthwise weight gradient (vlsfr_dwconv_wgrad_ws)
    const size_t w = u.kind != DW ? vlsfr_conv2d_wgrad_workspace_bytes(&u.d, 0) : vlsfr_dwconv_wgrad_workspace_bytes(&u.d);
    if (w > n->wgrad_ws) n->wgrad_ws = w;
  }
  {
    const size_t w = vlsfr_conv2d_wgrad_workspace_bytes(&n->l1d, 0);
    if (w > n->wgrad_ws) n->wgrad_ws = w;
  }
  n->scratch_bytes = 4 * align_up(n->max_act) + align_up((size_t)64 * 32 * 4) + align_up((size_t)n->B * n->D * 4) +
                     align_up((size_t)n->B * n->D * 2) + align_up(n->wgrad_ws);
  return VLSFR_OK;
}

struct Scratch {
  char* g[4];
  float* stem_dw;
  float* dz;
  char* dfc;
  void* wgrad_ws;
};
Scratch carve(const vlsfr_mobilenet* n, void* scratch) {
  Scratch s;
  char* p = (char*)scratch;
  const size_t a = align_up(n->max_act);
  for (int i = 0; i < 4; ++i) s.g[i] = p + i * a;
  p += 4 * a;
  s.stem_dw = (float*)p;
  p += align_up((size_t)64 * 32 * 4);
  s.dz = (float*)p;
  p += align_up((size_t)n->B * n->D * 4);
  s.dfc = p;
  p += align_up((size_t)n->B * n->D * 2);
  s.wgrad_ws = p;
  return s;
}

// conv -> BN -> (PReLU) (+ residual) of unit k (ConvBlock / the three stages of a BottleNeck, mobilefacenet_def.py:27-74)
int forward_unit(const vlsfr_mobilenet* n, int k, const float* x_nchw, const float* const* params, float* const* running,
                 char* ctx, const char* wc, void* st) {
  const Unit& u = n->units[k];
  double* sums = (double*)(ctx + u.off_sums);
  const char* in = u.in_unit >= 0 ? ctx + n->units[u.in_unit].a : nullptr;
  if (u.kind == STEM) {
    RUN(vlsfr_stem_im2col(x_nchw, ctx + n->off_cols, n->B, n->S, n->S, 2, st));
    RUN(vlsfr_conv2d_fwd(&u.d, ctx + n->off_cols, wc + u.off_wb, ctx + u.c, 1, 0, sums, st));
  } else if (u.kind == PW) {
    RUN(vlsfr_conv2d_fwd(&u.d, in, wc + u.off_wb, ctx + u.c, 1, 0, sums, st));
  } else {
    RUN(vlsfr_dwconv_fwd(&u.d, in, params[u.p_w], ctx + u.c, sums, st));
  }
  const int64_t M = (int64_t)n->B * u.Ho * u.Wo;
  const void* res = u.res_unit >= 0 ? ctx + n->units[u.res_unit].a : nullptr;
  return vlsfr_bn_apply(ctx + u.c, ctx + u.a, M, u.d.Cout, u.Ho * u.Wo, sums, params[u.p_g], params[u.p_b],
                        u.p_slope >= 0 ? params[u.p_slope] : nullptr, res, (float*)(ctx + u.off_mean),
                        (float*)(ctx + u.off_invstd), running ? running[2 * u.run] : nullptr,
                        running ? running[2 * u.run + 1] : nullptr, BN_EPS, BN_MOM, nullptr, 0, st);
}

// Backward walk over units k_hi .. k_lo (descending).  In: sc.g[*cur] = d(output of unit k_hi).  Out: sc.g[*cur] =
// d(input of unit k_lo) unless k_lo is the stem; a residual source in front of the range (unit k_lo - 1 feeding a later
// residual add) leaves its extra gradient in sc.g[*pend_buf] with *pend_unit = k_lo - 1.
int backward_units(const vlsfr_mobilenet* n, int k_hi, int k_lo, int* cur_io, int* pend_unit_io, int* pend_buf_io,
                   const float* const* params, float* const* grads, char* ctx, const char* wc, const Scratch& sc, void* st) {
  int cur = *cur_io, pend_unit = *pend_unit_io, pend_buf = *pend_buf_io;
  for (int k = k_hi; k >= k_lo; --k) {
    const Unit& u = n->units[k];
    const int64_t M = (int64_t)n->B * u.Ho * u.Wo;
    // free buffers: any of the 4 that is neither `cur` nor `pend_buf`
    int t1 = -1, t2 = -1;
    for (int i = 0; i < 4; ++i)
      if (i != cur && i != pend_buf) {
        if (t1 < 0) t1 = i;
        else if (t2 < 0) t2 = i;
      }
    if (pend_unit == k) {   // this unit's output also fed a later residual add: sum both gradients
      RUN(vlsfr_add_bf16(sc.g[cur], sc.g[pend_buf], sc.g[cur], M * u.d.Cout, st));
      pend_unit = -1;
      pend_buf = -1;
      t1 = t2 = -1;
      for (int i = 0; i < 4; ++i)
        if (i != cur) {
          if (t1 < 0) t1 = i;
          else if (t2 < 0) t2 = i;
        }
    }
    if (u.res_unit >= 0) {   // out = bn(c) + a[res]: the residual source receives d(out) as is
      pend_unit = u.res_unit;
      pend_buf = cur;        // keep d(out) alive; the BN backward below writes elsewhere
    }
    char* dc = sc.g[t1];
    RUN(vlsfr_bn_backward(sc.g[cur], ctx + u.c, dc, M, u.d.Cout, u.Ho * u.Wo, (const float*)(ctx + u.off_mean),
                          (const float*)(ctx + u.off_invstd), params[u.p_g], params[u.p_b],
                          u.p_slope >= 0 ? params[u.p_slope] : nullptr, (float*)(ctx + u.off_red), nullptr,
                          grads[u.p_g], grads[u.p_b], u.p_slope >= 0 ? grads[u.p_slope] : nullptr, 0, st));
    const char* in = u.in_unit >= 0 ? ctx + n->units[u.in_unit].a : nullptr;
    if (u.kind == STEM) {
      RUN(vlsfr_zero_bytes(sc.stem_dw, 64 * 32 * 4, st));
      RUN(vlsfr_conv2d_wgrad_ws(&u.d, dc, ctx + n->off_cols, sc.stem_dw, 0, sc.wgrad_ws, n->wgrad_ws, st));
      RUN(vlsfr_unpad_add(sc.stem_dw, grads[u.p_w], 64, 32, 27, st));
      break;
    }
    char* din = sc.g[t2];
    if (u.kind == PW) {
      RUN(vlsfr_conv2d_wgrad_ws(&u.d, dc, in, grads[u.p_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
      RUN(vlsfr_conv2d_dgrad(&u.d, dc, wc + u.off_wT, din, st));
    } else {
      RUN(vlsfr_dwconv_wgrad_ws(&u.d, dc, in, grads[u.p_w], sc.wgrad_ws, n->wgrad_ws, st));
      RUN(vlsfr_dwconv_dgrad(&u.d, dc, params[u.p_w], din, st));
    }
    cur = t2;
  }
  *cur_io = cur;
  *pend_unit_io = pend_unit;
  *pend_buf_io = pend_buf;
  return VLSFR_OK;
}

}  // namespace

extern "C" {

int vlsfr_mobilenet_create(int32_t feat_dim, int32_t batch, int32_t image_hw, vlsfr_mobilenet** out) {
  if (!out || feat_dim <= 0 || feat_dim % 8 || batch <= 0 || image_hw != 112)
    return fail(VLSFR_EINVAL, "vlsfr_mobilenet_create: need feat_dim %% 8 == 0 and a 112x112 input");
  vlsfr_mobilenet* n = new (std::nothrow) vlsfr_mobilenet();
  if (!n) return fail(VLSFR_ENOMEM, "vlsfr_mobilenet_create: out of memory");
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
void vlsfr_mobilenet_destroy(vlsfr_mobilenet* n) { delete n; }
int32_t vlsfr_mobilenet_num_params(const vlsfr_mobilenet* n) { return n ? n->n_params : -1; }
int32_t vlsfr_mobilenet_num_bn(const vlsfr_mobilenet* n) { return n ? n->n_bn : -1; }
size_t vlsfr_mobilenet_wcache_bytes(const vlsfr_mobilenet* n) { return n ? n->wcache_bytes : 0; }
size_t vlsfr_mobilenet_ctx_bytes(const vlsfr_mobilenet* n) { return n ? n->ctx_bytes : 0; }
size_t vlsfr_mobilenet_scratch_bytes(const vlsfr_mobilenet* n) { return n ? n->scratch_bytes : 0; }

int vlsfr_mobilenet_prepare_weights(const vlsfr_mobilenet* n, const float* const* params, void* wcache, void* st) {
  if (!n || !params || !wcache) return fail(VLSFR_EINVAL, "vlsfr_mobilenet_prepare_weights: null argument");
  char* wc = (char*)wcache;
  std::vector<vlsfr_cast_entry> tab;
  for (const auto& u : n->units) {
    if (u.kind == STEM) tab.push_back({params[u.p_w], wc + u.off_wb, nullptr, u.d.Cout, 1, 27, 32});
    else if (u.kind == PW) tab.push_back({params[u.p_w], wc + u.off_wb, wc + u.off_wT, u.d.Cout, 1, u.d.Cin, u.d.Cin});
  }
  tab.push_back({params[n->l1_w], wc + n->l1_wb, wc + n->l1_wT, n->D, 1, 512, 512});
  return vlsfr_cast_weights(tab.data(), (int32_t)tab.size(), st);
}

int vlsfr_mobilenet_forward(const vlsfr_mobilenet* n, const float* x_nchw, const float* const* params,
                            float* const* running, const void* wcache, void* ctx_v, void* scratch, float* emb_out,
                            void* st) {
  if (!n || !x_nchw || !params || !wcache || !ctx_v || !scratch || !emb_out)
    return fail(VLSFR_EINVAL, "vlsfr_mobilenet_forward: null argument");
  (void)scratch;
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  RUN(vlsfr_zero_bytes(ctx + n->sums_begin, n->sums_end - n->sums_begin, st));
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < n->units.size(); ++k) RUN(forward_unit(n, (int)k, x_nchw, params, running, ctx, wc, st));
  // linear1 (1x1 on the 1x1 map) -> BN over the batch -> flatten -> normalise (mobilefacenet_def.py:112-114)
  const Unit& l7 = n->units.back();
  RUN(vlsfr_conv2d_fwd(&n->l1d, ctx + l7.a, wc + n->l1_wb, ctx + n->off_fc, 1, 1, nullptr, st));
  RUN(vlsfr_embed_fwd((const float*)(ctx + n->off_fc), (const float*)(ctx + n->off_zero_bias), params[n->l1_g],
                      params[n->l1_b], running ? running[2 * n->l1_run] : nullptr,
                      running ? running[2 * n->l1_run + 1] : nullptr, (float*)(ctx + n->off_z),
                      (float*)(ctx + n->off_xhat), (float*)(ctx + n->off_invstd), (float*)(ctx + n->off_emb),
                      (float*)(ctx + n->off_invnorm), n->B, n->D, BN_EPS, BN_MOM, st));
  (void)e;
  return vlsfr_copy_bytes(ctx + n->off_emb, emb_out, (size_t)n->B * n->D * 4, st);   // (a kernel: captured passes hold no runtime copy nodes)
}

int vlsfr_mobilenet_backward(const vlsfr_mobilenet* n, const float* demb, const float* const* params,
                             float* const* grads, const void* wcache, void* ctx_v, void* scratch, void* st) {
  if (!n || !demb || !params || !grads || !wcache || !ctx_v || !scratch)
    return fail(VLSFR_EINVAL, "vlsfr_mobilenet_backward: null argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  RUN(vlsfr_zero_bytes(ctx + n->red_begin, n->red_end - n->red_begin, st));
  const Unit& l7 = n->units.back();
  RUN(vlsfr_embed_bwd(demb, (const float*)(ctx + n->off_emb), (const float*)(ctx + n->off_invnorm),
                      (const float*)(ctx + n->off_xhat), (const float*)(ctx + n->off_invstd), params[n->l1_g], sc.dz,
                      sc.dfc, grads[n->l1_b], nullptr, grads[n->l1_g], n->B, n->D, st));
  RUN(vlsfr_conv2d_wgrad_ws(&n->l1d, sc.dfc, ctx + l7.a, grads[n->l1_w], 0, sc.wgrad_ws, n->wgrad_ws, st));
  // gradient buffers: `cur` = d(output of unit k); a residual source keeps its extra gradient in
  // `pend` until the walk reaches it (the bottleneck input is consumed 3 units later).
  int cur = 0;
  RUN(vlsfr_conv2d_dgrad(&n->l1d, sc.dfc, wc + n->l1_wT, sc.g[cur], st));
  int pend_unit = -1;
  int pend_buf = -1;
  RUN(backward_units(n, (int)n->units.size() - 1, 0, &cur, &pend_unit, &pend_buf, params, grads, ctx, wc, sc, st));
  return VLSFR_OK;
}

// ---- teacher-forced execution of units u0 .. u1 - 1 (parity tests of the executor wiring; see vlsfr_iresnet_forward_blocks).
// A BottleNeck (mobilefacenet_def.py:27-52) is three consecutive units (1x1, depthwise 3x3, 1x1 linear [+ residual]);
// x_in is the output of unit u0 - 1 (u0 >= 1), bf16 NHWC.
int vlsfr_mobilenet_unit_info(const vlsfr_mobilenet* n, int32_t k, int32_t* info /*[8]*/) {
  if (!n || !info || k < 0 || k >= (int)n->units.size()) return fail(VLSFR_EINVAL, "vlsfr_mobilenet_unit_info: bad argument");
  const Unit& u = n->units[k];
  const int32_t v[8] = {(int32_t)u.kind, u.d.Cin, u.d.Cout, u.d.H, u.Ho, u.res_unit, u.p_w, (int32_t)n->units.size()};
  std::memcpy(info, v, sizeof(v));
  return VLSFR_OK;
}

int vlsfr_mobilenet_forward_units(const vlsfr_mobilenet* n, int32_t u0, int32_t u1, const void* x_in, const float* const* params,
                                  float* const* running, const void* wcache, void* ctx_v, void* scratch, void* out, void* st) {
  if (!n || !x_in || !params || !wcache || !ctx_v || !scratch || !out || u0 < 1 || u1 <= u0 || u1 > (int)n->units.size())
    return fail(VLSFR_EINVAL, "vlsfr_mobilenet_forward_units: bad argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  hipStream_t s = (hipStream_t)st;
  RUN(vlsfr_zero_bytes(ctx + n->sums_begin, n->sums_end - n->sums_begin, (void*)s));
  hipError_t e = hipSuccess;
  const Unit& prev = n->units[u0 - 1];
  if (e == hipSuccess)
    e = hipMemcpyAsync(ctx + prev.a, x_in, (size_t)n->B * prev.Ho * prev.Wo * prev.d.Cout * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_mobilenet_forward_units: %s", hipGetErrorString(e));
  for (int k = u0; k < u1; ++k) {
    if (n->units[k].in_unit < u0 - 1 || (n->units[k].res_unit >= 0 && n->units[k].res_unit < u0 - 1))
      return fail(VLSFR_EINVAL, "vlsfr_mobilenet_forward_units: unit %d reads a tensor in front of the range", k);
    RUN(forward_unit(n, k, nullptr, params, running, ctx, wc, st));
  }
  const Unit& lastu = n->units[u1 - 1];
  e = hipMemcpyAsync(out, ctx + lastu.a, (size_t)n->B * lastu.Ho * lastu.Wo * lastu.d.Cout * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_mobilenet_forward_units: copy: %s", hipGetErrorString(e));
  return VLSFR_OK;
}

int vlsfr_mobilenet_backward_units(const vlsfr_mobilenet* n, int32_t u0, int32_t u1, const void* dout, const float* const* params,
                                   float* const* grads, const void* wcache, void* ctx_v, void* scratch, void* dx, void* st) {
  if (!n || !dout || !params || !grads || !wcache || !ctx_v || !scratch || !dx || u0 < 1 || u1 <= u0 || u1 > (int)n->units.size())
    return fail(VLSFR_EINVAL, "vlsfr_mobilenet_backward_units: bad argument");
  char* ctx = (char*)ctx_v;
  const char* wc = (const char*)wcache;
  Scratch sc = carve(n, scratch);
  hipStream_t s = (hipStream_t)st;
  RUN(vlsfr_zero_bytes(ctx + n->red_begin, n->red_end - n->red_begin, (void*)s));
  hipError_t e = hipSuccess;
  const Unit& lastu = n->units[u1 - 1];
  int cur = 0, pend_unit = -1, pend_buf = -1;
  if (e == hipSuccess)
    e = hipMemcpyAsync(sc.g[cur], dout, (size_t)n->B * lastu.Ho * lastu.Wo * lastu.d.Cout * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_mobilenet_backward_units: %s", hipGetErrorString(e));
  RUN(backward_units(n, u1 - 1, u0, &cur, &pend_unit, &pend_buf, params, grads, ctx, wc, sc, st));
  const Unit& prev = n->units[u0 - 1];
  const int64_t cnt = (int64_t)n->B * prev.Ho * prev.Wo * prev.d.Cout;
  if (pend_unit == u0 - 1) RUN(vlsfr_add_bf16(sc.g[cur], sc.g[pend_buf], sc.g[cur], cnt, st));   // the range's input also fed its residual add
  e = hipMemcpyAsync(dx, sc.g[cur], (size_t)cnt * 2, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "vlsfr_mobilenet_backward_units: copy: %s", hipGetErrorString(e));
  return VLSFR_OK;
}

}  // extern "C"
