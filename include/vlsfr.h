/*
 * vlsfr.h — C-ABI of libvlsfr.so: the MI355X-native (gfx950) FFC training hot path.
 *
 * The reference (sqnkkang/Very-Large-Scale-Face-Recognition) is 100 % Python/PyTorch and has no
 * FFI of its own; its boundary for this path is the Python module API
 * (ffc.FFC / lru.LRU / model.create_net / optim.get_optim_scheduler).  The build's Python mirror
 * of that API (the .py modules of very-large-scale-face-recognition_amd/) binds the entry points below through
 * ctypes on tensor.data_ptr(); INTEGRATION.md shows the stub.  Every entry point names the
 * reference lines it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative VLSFR_E* code on failure;
 *     vlsfr_last_error() returns a thread-local message for the last failure.
 *   - device buffers are caller-owned (PyTorch-allocated); no hidden allocation, no hidden
 *     synchronisation; every device entry point takes an explicit hipStream_t (as void*).
 *   - workspaces are sized by the matching *_workspace_bytes() query.
 *   - plain pointers and sizes only — no torch types.
 */
#ifndef VLSFR_H
#define VLSFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLSFR_OK 0
#define VLSFR_EINVAL (-1)   /* bad argument / shape the kernel does not cover            */
#define VLSFR_ESTATE (-2)   /* call not legal in the handle's current state (assert in ref) */
#define VLSFR_EHIP (-3)     /* a HIP runtime call failed                                 */
#define VLSFR_ENOMEM (-4)

const char* vlsfr_last_error(void);
int vlsfr_version(void);

/* ------------------------------------------------------------------------------------------
 * 1. LRU slot allocator (host).  Replaces lru.py:21-255 (class LRU).
 *    Keys are int64 identity labels, values are int32 pool slots.
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_lru vlsfr_lru;

/* lru.py:27-40 */
int vlsfr_lru_create(int64_t capacity, vlsfr_lru** out);
void vlsfr_lru_destroy(vlsfr_lru* h);
/* lru.py:44-89  LRU.get — committing lookup/insert/evict; *slot receives the value */
int vlsfr_lru_get(vlsfr_lru* h, int64_t key, int32_t* slot);
/* lru.py:157-204 LRU.try_get — same, and pushes an undo record on the op stack */
int vlsfr_lru_try_get(vlsfr_lru* h, int64_t key, int32_t* slot);
/* lru.py:147-151 LRU.view — slot or -1, never reorders */
int vlsfr_lru_view(const vlsfr_lru* h, int64_t key, int32_t* slot);
/* lru.py:145 LRU.__contains__ — returns 1/0 */
int vlsfr_lru_contains(const vlsfr_lru* h, int64_t key);
/* lru.py:210-255 rollback_one_step / rollback_steps — clamped to the stack depth; returns the
 * number of steps actually undone in *undone (may be NULL) */
int vlsfr_lru_rollback(vlsfr_lru* h, int64_t steps, int64_t* undone);
/* lru.py:102-108 state_dict — MRU→LRU (key, slot) pairs; n_max = capacity of the out arrays;
 * *n receives the number of entries */
int vlsfr_lru_state(const vlsfr_lru* h, int64_t* keys, int32_t* slots, int64_t n_max, int64_t* n);
/* lru.py:113-128 restore — requires cur_idx == 0 and n <= capacity and distinct keys
 * (VLSFR_ESTATE otherwise, the reference asserts) */
int vlsfr_lru_restore(vlsfr_lru* h, const int64_t* keys, const int32_t* slots, int64_t n);
/* lru.py:132-141 clear — empties list and map, does NOT reset cur_idx (reference behaviour) */
int vlsfr_lru_clear(vlsfr_lru* h);
/* public fields lru.py:28-30,40: capacity, cur_idx, len(cache), len(op_stack) */
int64_t vlsfr_lru_capacity(const vlsfr_lru* h);
int64_t vlsfr_lru_cur_idx(const vlsfr_lru* h);
int64_t vlsfr_lru_size(const vlsfr_lru* h);
int64_t vlsfr_lru_op_depth(const vlsfr_lru* h);
/* op-stack type names for op_stack[i].op_type: 0 = 'Add', 1 = 'Overflow', 2 = 'Get' */
int vlsfr_lru_op_type(const vlsfr_lru* h, int64_t i);

/* ------------------------------------------------------------------------------------------
 * 2. Dynamic-Class-Pool bookkeeping for one FFC pass (host).
 *    Replaces the Python loops of ffc.py:162-177 / 189-192 (forward_impl) and
 *    ffc.py:214-235 / 242-245 / 256-259 (forward_impl_rollback): gallery labels → (row, slot)
 *    write list, probe labels → pool labels, ones_idx, queue_position_dict toggling.
 *
 *    qp is the reference's queue_position_dict as a uint8[capacity] array owned by the caller.
 *    transactional != 0 selects try_get + old_state semantics; vlsfr_dcp_undo() then restores
 *    qp and rolls the LRU back (ffc.py:256-259).
 *
 *    Besides the reference's outputs the call emits the "special column" table the fused head
 *    kernel consumes: the pool slots whose logits cannot be taken from an
 *    unmodified sweep over queue[0] — slots written in this pass, slots in ones_idx, and the
 *    probe rows' label slots.  For special column s, src1[s] / src2[s] say where the class
 *    vector of the cos_theta1 / cos_theta2 contraction (ffc.py:195,201 / 248,253) comes from:
 *        >= 0 : row i of this pass's gallery embeddings g (last writer wins)
 *        -1   : queue[0][slot] as stored      -2 : queue[1][slot] as stored
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_dcp_plan {
  int32_t n;          /* batch rows B                                                        */
  int32_t n_ones;     /* |ones_idx|                                                          */
  int32_t n_special;  /* number of special columns (<= 3B)                                   */
  int32_t n_pos;      /* probe rows with label != -1                                         */
  int32_t n_undo;     /* transactional: number of qp entries saved                           */
  int32_t steps;      /* transactional: LRU steps to roll back (== n)                        */
} vlsfr_dcp_plan;

int vlsfr_dcp_assign(vlsfr_lru* h, uint8_t* qp, const int64_t* gallery_label,
                     const int64_t* probe_label, int32_t n, int transactional,
                     int32_t* rows /*[n]*/, int32_t* cols /*[n]*/, int32_t* pool_label /*[n]*/,
                     int32_t* ones_idx /*[n]*/, int32_t* special_col /*[3n]*/,
                     int32_t* src1 /*[3n]*/, int32_t* src2 /*[3n]*/,
                     int32_t* undo_slot /*[n]*/, uint8_t* undo_val /*[n]*/,
                     vlsfr_dcp_plan* plan);
int vlsfr_dcp_undo(vlsfr_lru* h, uint8_t* qp, const int32_t* undo_slot, const uint8_t* undo_val,
                   const vlsfr_dcp_plan* plan);

/* ------------------------------------------------------------------------------------------
 * 3. Pool rows (device).  Replaces `self.queue[r, c] = g` (ffc.py:182; the transient write and
 *    restore of ffc.py:240-241,255 are not needed: the rollback pass never mutates the pool).
 *    queue: fp32 [2, Q, D] row-major; duplicate (row, col): the highest batch index wins.
 * ---------------------------------------------------------------------------------------- */
/* cols are global slot ids; with an identity-sharded pool (queue holds slots
 * [slot_lo, slot_lo + Q)) writes to slots of other ranks are skipped */
int vlsfr_pool_scatter(float* queue, int64_t Q, int32_t D, const float* g /*[n, D]*/,
                       const int32_t* rows /*[n] dev*/, const int32_t* cols /*[n] dev*/, int32_t n,
                       int32_t slot_lo, void* shadow_bf16 /* [Q, D] bf16 mirror of queue[0], or NULL */, void* stream);
/* bf16 shadow of queue[0] ([Q, D], round-to-nearest-even), the operand the head sweep streams when
 * vlsfr_head_cfg.pool_bf16 is set: half the HBM bytes of the fp32 pool and no in-kernel conversion.  The fp32
 * master stays the source of truth (special columns, precise mode, checkpoint); the caller rebuilds the shadow
 * whenever it changes queue[0] by other means than vlsfr_pool_scatter (load_state_dict, copy_). */
int vlsfr_pool_shadow_build(const float* queue0 /*[Q, D]*/, void* shadow_bf16, int64_t Q, int32_t D, void* stream);
/* fp8 (e4m3, values x 64) shadow of queue[0] for the fp8 sweep, stored fragment-major in tiles of 128 slots (layout:
 * csrc/head8.hip): vlsfr_pool_shadow8_bytes(Q) bytes.  _update rewrites the images of `n` slots (global ids in `cols`,
 * device; slots outside [slot_lo, slot_lo + Q) are skipped) from the fp32 master after vlsfr_pool_scatter changed them. */
size_t vlsfr_pool_shadow8_bytes(int64_t Q);
int vlsfr_pool_shadow8_build(const float* queue0 /*[Q, 512]*/, void* shadow8, int64_t Q, int32_t D, void* stream);
int vlsfr_pool_shadow8_update(const float* queue0, int64_t Q, int32_t D, const int32_t* cols /*[n] dev*/, int32_t n,
                              int32_t slot_lo, void* shadow8, void* stream);

/* ------------------------------------------------------------------------------------------
 * 4. Fused DCP head (device): loss and dL/dp of one FFC pass in one sweep over queue[0].
 *    Replaces F.linear x2 + mask blend + add_margin x2 and their backward
 *    (ffc.py:195-203 / 248-258, ffc.py:60-138).
 *    loss_type: 0 'AM', 1 'Arc', 2 'SV' (ffc.py:17).  precise != 0 selects split-bf16 (hi + lo)
 *    MFMA products (~fp32 accuracy, parity mode); 0 = plain bf16 MFMA operands, fp32 accumulate.
 *    n_chunks = 0 lets the library choose the column partition.
 *    Device inputs: p, g fp32 [B, D]; queue fp32 [2, Q, D]; pool_label int32 [B] (slot or -1);
 *    special_col/src1/src2 int32 [n_special] from vlsfr_dcp_assign.
 *    Device outputs: loss_out fp32 [1] (= add_margin(cos_theta1) + add_margin(cos_theta2)),
 *    dP fp32 [B, D] (= d loss / d p against the pool as it stands at call time).
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_head_cfg {
  int32_t B;
  int32_t D;
  int64_t Q;
  int32_t loss_type;
  float scale;
  float margin;
  int32_t hard_neg;
  int32_t precise;
  int32_t n_chunks;
  int32_t slot_lo;        /* identity-sharded pool: queue holds global slots [slot_lo, slot_lo + Q); else 0 */
  int32_t n_rows_total;   /* 0 or B: single process.  > B: this call holds B probe rows of a batch of
                             n_rows_total rows spread over ranks (g, the special table, n_pos and the
                             loss normalisers refer to the whole batch; pool_label to these B rows) */
  const void* pool_bf16;  /* NULL, or the bf16 shadow [Q, D] of queue[0] (vlsfr_pool_shadow_build).  With D = 512,
                             precise = 0 and scale <= 64 the sweep then runs the LDS-DMA kernel of csrc/head16.hip
                             on it; it relies on what the reference's pool guarantees — every row of queue[] is an
                             F.normalize output (ffc.py:30,182), so |cos| <= |p| — to fix the softmax reference
                             exponent per row up front instead of tracking a running maximum.  The workspace size
                             depends on this field: query with the cfg the call will use. */
  const void* pool_fp8;   /* NULL, or the fp8 (OCP e4m3) shadow of queue[0] (vlsfr_pool_shadow8_build): the sweep then runs
                             both contractions on v_mfma_scale_f32_16x16x128_f8f6f4 (csrc/head8.hip; config C5's precision:
                             cosines and dL/dp to ~1e-2, loss to ~1e-3).  Same conditions as pool_bf16, which it overrides. */
} vlsfr_head_cfg;

/* sizeof(vlsfr_head_cfg) as this library was compiled: bindings assert their mirror of the struct against it */
size_t vlsfr_head_cfg_size(void);
size_t vlsfr_head_workspace_bytes(const vlsfr_head_cfg* cfg);
/* Identity-sharded pool (the class matrix split by slot range over the ranks of a node): this rank's
 * partial softmax state of one pass for ALL B rows of the (all-gathered) batch over its own slots —
 * per row and variant M (log2 units), L, O[D] (unnormalised), the target terms T[D] / zt if it owns
 * the label slot, and its local top-k hard-negative candidates (value, global slot).  The ranks
 * combine with all-reduce(max) on M and all-reduce(sum) on the rescaled (O, T, L, zt)
 * (head.py ShardedDcpHead).  AM / Arc; SV goes through the two calls below. */
int vlsfr_head_shard_partial(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                             const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                             const int32_t* src2, int32_t n_special, int32_t n_pos, float* out_M /*[B,2]*/,
                             float* out_L /*[B,2]*/, float* out_zt /*[B,2]*/, float* out_O /*[B,2,D]*/,
                             float* out_T /*[B,2,D]*/, float* cand_val /*[B,2,10]*/, int32_t* cand_col /*[B,2,10]*/,
                             void* workspace, size_t workspace_bytes, void* stream);
/* SV under the sharded pool (ffc.py:118-127: columns with cos > gt - margin are "hard" and become t*cos + t - 1):
 * the threshold of a row is known on the rank that owns its label slot.  vlsfr_head_shard_sv_thr writes this rank's
 * view thr_out[2][B] (variant-major; -3e38 where the label slot is another rank's, +3e38 for outlier rows); after an
 * all-reduce(max) over the ranks the result goes to vlsfr_head_shard_partial_sv, otherwise identical to
 * vlsfr_head_shard_partial (one sweep per variant). */
int vlsfr_head_shard_sv_thr(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                            const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                            const int32_t* src2, int32_t n_special, float* thr_out /*[2,B]*/, void* workspace,
                            size_t workspace_bytes, void* stream);
int vlsfr_head_shard_partial_sv(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                                const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                                const int32_t* src2, int32_t n_special, int32_t n_pos, const float* sv_thr /*[2,B] global*/,
                                float* out_M, float* out_L, float* out_zt, float* out_O, float* out_T, float* cand_val,
                                int32_t* cand_col, void* workspace, size_t workspace_bytes, void* stream);
/* T[row, v, :] += sel_w * class vector for the globally selected hard negatives this rank owns */
int vlsfr_head_outlier_accum(const vlsfr_head_cfg* cfg, const float* g, const float* queue,
                             const int32_t* special_col, const int32_t* src1, const int32_t* src2,
                             int32_t n_special, const int32_t* sel_col /*[B,2,k]*/, const float* sel_w /*[B,2,k]*/,
                             int32_t k, float* T /*[B,2,D]*/, void* stream);
int vlsfr_head_outlier_accum_strided(const vlsfr_head_cfg* cfg, const float* g, const float* queue,
                                     const int32_t* special_col, const int32_t* src1, const int32_t* src2,
                                     int32_t n_special, const int32_t* sel_col, const float* sel_w, int32_t k,
                                     float* T /*row (b, v) at T + (2 b + v) * t_stride*/, int32_t t_stride, void* stream);
/* The sharded pass as three launches around the ranks' collectives (head.py ShardedDcpHead):
 *  vlsfr_head_shard_partial_packed — as vlsfr_head_shard_partial(_sv with sv_thr != NULL), but the state goes straight
 *    into the layout the ranks sum: packed [B, 2, 2 D + 2] = (O[D] | T[D] | L | zt) per row and variant; out_M [B, 2] is the
 *    exponent the state is relative to.  *fixed_ref = 1: that exponent is the shadow sweep's fixed reference (a function
 *    of the probe row alone: identical on every rank), so the ranks sum `packed` WITHOUT an all-reduce(max) and without a
 *    rescale; 0 (fp32-pool sweep): out_M is this rank's maximum, rescale by 2^(M - max over ranks) first.
 *  vlsfr_head_shard_topk_merge — hard negatives (only when the batch has outlier rows): the global top-hard_neg of the
 *    gathered candidates [world, B, 2, 10] -> sel_col / sel_w [B, 2, hard_neg] (for vlsfr_head_outlier_accum_strided with
 *    T = packed + D, t_stride = 2 D + 2) and sel_loss [B, 2].
 *  vlsfr_head_shard_finish — after the reduce-scatter (or all-reduce) of `packed`: loss share (deterministic sum) and the
 *    dL/dp rows of the n_rows rows given (their packed rows, exponents Mg [n_rows, 2], labels, sel_loss or NULL);
 *    row_loss: scratch [n_rows, 2]. */
int vlsfr_head_shard_partial_packed(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                                    const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                                    const int32_t* src2, int32_t n_special, int32_t n_pos, const float* sv_thr /*or NULL*/,
                                    float* packed, float* out_M, float* cand_val, int32_t* cand_col, int32_t* fixed_ref,
                                    void* workspace, size_t workspace_bytes, void* stream);
int vlsfr_head_shard_topk_merge(const vlsfr_head_cfg* cfg, const float* cand_val, const int32_t* cand_col, int32_t world,
                                const int32_t* pool_label, int32_t n_out, int32_t* sel_col, float* sel_w, float* sel_loss,
                                void* stream);
int vlsfr_head_shard_finish(const vlsfr_head_cfg* cfg, const float* packed, const float* Mg, const int32_t* pool_label,
                            const float* sel_loss, int32_t n_rows, int32_t n_pos, float* row_loss, float* loss_out, float* dP,
                            void* stream);
int vlsfr_head_fwd_bwd(const vlsfr_head_cfg* cfg, const float* p, const float* g, const float* queue,
                       const int32_t* pool_label, const int32_t* special_col, const int32_t* src1,
                       const int32_t* src2, int32_t n_special, int32_t n_pos, float* loss_out, float* dP,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 0. Hardware-convention probes (device, test support): one-wave kernels that pin the MFMA
 *    16x16x32 bf16 fragment maps and the ds_read_b64_tr_b16 transposed read the kernels use.
 *    A [16,32], B [32,16] / Bt [16,32], C [16,16], all fp32 device buffers.
 * ---------------------------------------------------------------------------------------- */
int vlsfr_probe_mfma_tr(const float* A, const float* B, float* C, int32_t row_stride_bytes, void* stream);
int vlsfr_probe_mfma_nat(const float* A, const float* Bt, float* C, void* stream);

/* ------------------------------------------------------------------------------------------
 * 5. Convolutions on MFMA (device).  Replaces nn.Conv2d forward / input-gradient /
 *    weight-gradient (model/resnet_arcface.py:5-23,36,39,74,120) and nn.Linear (:95, as a 1x1
 *    convolution on a 1x1 map).  Activations NHWC bf16; w bf16 [Cout][R][S][Cin]; wT bf16
 *    [Cin][R][S][Cout]; dw fp32 [Cout][R][S][Cin] (= memory of a channels_last OIHW tensor),
 *    ACCUMULATED atomically (zero it for a fresh gradient).
 *    Covered: 3x3 pad 1 and 1x1 pad 0, stride 1 or 2, Cin % 32 == 0, Cout % 8 == 0.
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_conv_desc {
  int32_t N, H, W;      /* input batch and spatial size */
  int32_t Cin, Cout;
  int32_t R, S, stride, pad;
} vlsfr_conv_desc;

/* y: bf16 [N,Ho,Wo,Cout], or fp32 when out_f32 (required for splitk > 1: accumulated atomically).
 * stats (optional, bf16 output only): FLOAT64 [VLSFR_BN_REPL][2][Cout] accumulators (VLSFR_BN_REPL * 2 * Cout * 8 bytes,
 * 8-byte aligned, pre-zeroed) that receive the per-channel sum / sum of squares of the rounded outputs — the BatchNorm
 * statistics of the following layer, fused into the epilogue (float64 atomics; layout and precision: section 6).  Only the
 * first vlsfr_set_option("bn_repl") replicas (default 8) are written; the consumer folds all VLSFR_BN_REPL of them, so the
 * buffer must be zeroed whole. */
int vlsfr_conv2d_fwd(const vlsfr_conv_desc* d, const void* x, const void* w, void* y, int32_t splitk,
                     int32_t out_f32, double* stats, void* stream);
/* Forward convolution whose INPUT is a BatchNorm (+ PReLU) of x, applied inside the kernel's operand path: y = conv(a), a =
 * prelu(x * scale + shift) (slope NULL: plain BatchNorm), exactly the tensor vlsfr_bn_apply would have written
 * (model/resnet_arcface.py:45-50: bn1 -> conv1, bn2 -> prelu -> conv2) — without the pass over x that writes it and the pass
 * that reads it back.  scale / shift: per input channel, from vlsfr_bn_finalize.  a_out (optional): receives a (bf16, shape of
 * x) as a by-product, for the backward pass (the weight gradient contracts a); NULL in passes that keep no activations.
 * Supported where vlsfr_conv2d_fwd_bnin_supported(d) != 0 (3x3, stride 1, pad 1, Cin % 64 == 0, Cout 128 or a multiple of 256,
 * and enough pixels to fill the chip); elsewhere the caller runs vlsfr_bn_apply + vlsfr_conv2d_fwd. */
typedef struct vlsfr_bn_in {
  const float* scale;
  const float* shift;
  const float* slope;
  void* a_out;
} vlsfr_bn_in;
int32_t vlsfr_conv2d_fwd_bnin_supported(const vlsfr_conv_desc* d);
int vlsfr_conv2d_fwd_bnin(const vlsfr_conv_desc* d, const void* x, const void* w, void* y, double* stats, const vlsfr_bn_in* bn,
                          void* stream);
int vlsfr_conv2d_dgrad(const vlsfr_conv_desc* d, const void* dy, const void* wT, void* dx, void* stream);
/* The same with the reduction of the BatchNorm (+ PReLU) backward that consumes dx as ITS dY fused into the epilogue
 * (resnet_arcface.py:35-38 backward: bn1 behind conv1, bn2 + prelu behind conv2): x is that layer's input (bf16, shape of
 * dx), mean / invstd its saved batch statistics, gamma / beta / slope its parameters (slope NULL: plain BatchNorm; gamma /
 * beta are then not read), red its fp32 [VLSFR_BN_REPL][3][Cin] accumulators (pre-zeroed): sum dz, sum dz * xhat,
 * sum dy * min(z, 0) with z = bn(x), dz = dy * prelu'(z), from the ROUNDED dx.  The later vlsfr_bn_backward_chain call for
 * that layer passes red_ready = 1.  bn == NULL: plain vlsfr_conv2d_dgrad. */
typedef struct vlsfr_bn_red {
  const void* x;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* slope;
  float* red;
} vlsfr_bn_red;
int vlsfr_conv2d_dgrad_bnred(const vlsfr_conv_desc* d, const void* dy, const void* wT, void* dx, const vlsfr_bn_red* bn,
                             void* stream);
/* splitk <= 0: library choice */
int vlsfr_conv2d_wgrad(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, int32_t splitk,
                       void* stream);
/* The same with a workspace of vlsfr_conv2d_wgrad_workspace_bytes(d, splitk) bytes: the split-K slices are written
 * to fp32 slabs with plain stores and summed in a fixed order by a second launch (one atomic per element of dw
 * instead of splitk), which makes the weight gradient bit-reproducible run to run.  Selected by
 * vlsfr_set_option("wgrad_slabs", 1); the default stays the atomic path (measured faster: 605 vs 532 TFLOP/s).
 * workspace NULL / too small, or a single slice: the atomic path as well. */
size_t vlsfr_conv2d_wgrad_workspace_bytes(const vlsfr_conv_desc* d, int32_t splitk);
int vlsfr_conv2d_wgrad_ws(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, int32_t splitk,
                          void* workspace, size_t workspace_bytes, void* stream);
/* n (1 .. 4) weight gradients of the SAME descriptor in one launch — consecutive layers of a stage (the 58 stride-1
 * 256 -> 256 convolutions of iResNet100 share one shape, model/resnet_arcface.py:36,39): dw[g] += wgrad(dy[g], x[g]).  The
 * launch splits every problem into n times fewer, longer pixel slices than it would take alone, so the fp32 atomics into
 * each gradient and the launch / prologue are paid once per n layers' worth of work (84 -> ~65 us per layer at batch 256).
 * Nothing downstream of a weight gradient runs before the optimizer step, so an executor may defer the weight gradient of a
 * layer until it has n of them (the dy and x tensors must stay intact until this call is enqueued).  workspace: as for
 * vlsfr_conv2d_wgrad_ws, vlsfr_conv2d_wgrad_group_workspace_bytes(d, n, splitk) bytes (slab path) or NULL. */
size_t vlsfr_conv2d_wgrad_group_workspace_bytes(const vlsfr_conv_desc* d, int32_t n, int32_t splitk);
int vlsfr_conv2d_wgrad_group(const vlsfr_conv_desc* d, int32_t n, const void* const* dy, const void* const* x, float* const* dw,
                             int32_t splitk, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 5b. Depthwise convolutions (device).  Replaces nn.Conv2d(groups = C) of MobileFaceNet
 *     (mobilefacenet_def.py:39 3x3 stride 1/2 pad 1; :60,88 7x7 valid), forward / input gradient /
 *     weight gradient.  d->Cin == d->Cout == C (C % 8 == 0), square filter <= 7.  x, y, dy, dx:
 *     NHWC bf16; w: fp32 [C][R*S] (the parameter's own memory); dw: fp32 [C][R*S] accumulated (+=);
 *     stats (optional): BatchNorm statistics of y, FLOAT64 [VLSFR_BN_REPL][2][C] (VLSFR_BN_REPL * 2 * C * 8 bytes,
 *     pre-zeroed; as for vlsfr_conv2d_fwd).
 * ---------------------------------------------------------------------------------------- */
int vlsfr_dwconv_fwd(const vlsfr_conv_desc* d, const void* x, const float* w, void* y, double* stats, void* stream);
int vlsfr_dwconv_dgrad(const vlsfr_conv_desc* d, const void* dy, const float* w, void* dx, void* stream);
int vlsfr_dwconv_wgrad(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, void* stream);
/* The same with a workspace of vlsfr_dwconv_wgrad_workspace_bytes(d) bytes: the workgroups leave per-block partial sums
 * there and a second kernel adds them into dw (deterministic per launch; without it every block ends in 9 C atomics on the
 * same addresses, ~80 us per launch whatever the tensor size).  workspace NULL = vlsfr_dwconv_wgrad. */
size_t vlsfr_dwconv_wgrad_workspace_bytes(const vlsfr_conv_desc* d);
int vlsfr_dwconv_wgrad_ws(const vlsfr_conv_desc* d, const void* dy, const void* x, float* dw, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 6. Normalisation / activation / layout kernels (device).  Replaces nn.BatchNorm2d in training
 *    mode (resnet_arcface.py:35,37,40,75,93), nn.PReLU (:38,76), the residual add (:54), the
 *    BatchNorm1d + F.normalize embedding tail (:96-98,151) and the layout/precision conversions.
 *    x, y, dy, dx, residual: bf16 [M, C] (NHWC rows), C % 8 == 0; statistics fp32.
 * ---------------------------------------------------------------------------------------- */
/* Per-channel reductions are kept in VLSFR_BN_REPL replicated accumulators ([VLSFR_BN_REPL][n][C], pre-zeroed by the
 * caller, accumulated atomically; the kernels use the first vlsfr_set_option("bn_repl") of them) and folded by the
 * consumer.  STATISTICS (sum, sum of squares) are FLOAT64: producers accumulate fp32 deviations from a local pivot and
 * add (sum x, sum x^2) in float64, so E[x^2] - mean^2 loses 1e-16 * mean^2, not 1e-7 * mean^2 (PyTorch's BatchNorm2d, the
 * reference's model/resnet_arcface.py:35,37,40, uses Welford merges to the same end).  Backward reductions stay fp32
 * (their sums are centred: dz * (x - mean)). */
#ifndef VLSFR_BN_REPL
#define VLSFR_BN_REPL 32
#endif
int vlsfr_bn_repl(void);   /* the value this library was compiled with */
/* sums [REPL][2][C] float64: sum and sum of squares over the M rows of x */
int vlsfr_bn_stats(const void* x, int64_t M, int32_t C, double* sums, void* stream);
/* y = prelu(bn(x)) + residual from the statistics `sums` of x (every block folds the replicas into
 * scale / shift itself; mean, invstd are saved for the backward pass; running_* get the momentum
 * update with the unbiased variance).  slope / residual / running_* may be NULL.  out_sums
 * (optional): the statistics [REPL][2][C] (float64) of y, for the next BatchNorm.  out_nchw bit 0 writes y in the
 * [n][c][hw] flatten order of the reference's fc input (HW = rows per image); bit 1 applies ReLU AFTER the
 * residual add, y = relu(bn(x) + residual) — the block ending of model/resnet_std.py:97-105 (needs residual,
 * no slope, no out_sums). */
/* The per-channel part of vlsfr_bn_apply alone: mean / invstd (saved for the backward pass), scale = gamma * invstd, shift =
 * beta - mean * scale, and the momentum update of running_* (may be NULL) — for a BatchNorm whose element-wise part runs inside
 * the convolution that consumes it (vlsfr_conv2d_fwd_bnin). */
int vlsfr_bn_finalize(const double* sums, int64_t M, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                      float* save_mean, float* save_invstd, float* scale, float* shift, float* running_mean, float* running_var,
                      void* stream);
int vlsfr_bn_apply(const void* x, void* y, int64_t M, int32_t C, int32_t HW, const double* sums,
                   const float* gamma, const float* beta, const float* slope, const void* residual,
                   float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                   float eps, float momentum, double* out_sums, int32_t out_nchw, void* stream);
/* dx = d(prelu(bn(x)))/dx applied to dy (+ dx_add); dgamma/dbeta/dslope are accumulated (+=);
 * red: fp32 [REPL][3][C] pre-zeroed scratch */
int vlsfr_bn_backward(const void* dy, const void* x, void* dx, int64_t M, int32_t C, int32_t HW,
                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* slope, float* red, const void* dx_add, float* dgamma, float* dbeta,
                      float* dslope, int32_t dy_nchw, void* stream);
/* The same, chained with the BatchNorm backward that consumes dx as ITS dY (iResNet: bn1 of block k writes the output gradient
 * of block k - 1, whose bn3 is next in the chain): with next_x (that layer's input, same shape as dx; a plain BatchNorm)
 * the apply kernel also accumulates that layer's reduction into next_red ([VLSFR_BN_REPL][3][C], zeroed) from the rounded
 * dx; the later call for that layer passes red_ready = 1 and skips its own reduction kernel — one tensor read instead of
 * two, one launch less per block. */
int vlsfr_bn_backward_chain(const void* dy, const void* x, void* dx, int64_t M, int32_t C, int32_t HW, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, const float* slope, float* red,
                            const void* dx_add, float* dgamma, float* dbeta, float* dslope, int32_t dy_nchw,
                            int32_t red_ready, const void* next_x, const float* next_mean, const float* next_invstd,
                            float* next_red, void* stream);
/* Only the reduction of vlsfr_bn_backward (red [VLSFR_BN_REPL][3][C], pre-zeroed: sum dz, sum dz * xhat, sum dy * min(z, 0));
 * what vlsfr_conv2d_dgrad_bnred falls back to when a launch carries no fused epilogue. */
int vlsfr_bn_backward_reduce(const void* dy, const void* x, int64_t M, int32_t C, int32_t HW, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, const float* slope, float* red,
                             void* stream);
int vlsfr_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);
/* Zero fill / device-to-device copy by a kernel launch (16-byte aligned pointers; the copy also a 16-byte multiple): what the
 * executors use for their accumulator regions and to hand out the embedding, so that a pass captured into a HIP graph consists of
 * kernel nodes only. */
int vlsfr_zero_bytes(void* p, size_t nbytes, void* stream);
int vlsfr_copy_bytes(const void* src, void* dst, size_t nbytes, void* stream);
/* e = normalize(bn1d(fc + fc_bias)); all fp32 [B, D] */
int vlsfr_embed_fwd(const float* fc, const float* fc_bias, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* z, float* xhat, float* invstd,
                    float* emb, float* inv_norm, int32_t B, int32_t D, float eps, float momentum,
                    void* stream);
/* dbeta (+=) is required; dfc_bias / dgamma (+=) may be NULL (no bias in front / frozen weight) */
int vlsfr_embed_bwd(const float* demb, const float* emb, const float* inv_norm, const float* xhat,
                    const float* invstd, const float* gamma, float* dz, void* dfc_bf16, float* dbeta,
                    float* dfc_bias, float* dgamma, int32_t B, int32_t D, void* stream);
/* fp32 [rows][taps][C] -> bf16 [rows][Kp] (zero padded to Kp >= taps*C) and optionally the
 * dgrad operand wT bf16 [C][taps][rows] */
int vlsfr_cast_weight(const float* w, void* w_bf16, void* wT_bf16, int32_t rows, int32_t taps, int32_t C,
                      int32_t Kp, void* stream);
/* The same for n tensors in one launch (what the backbone executors use for their weight caches). */
typedef struct vlsfr_cast_entry {
  const float* w;      /* fp32 [rows][taps][C] */
  void* w_bf16;        /* bf16 [rows][Kp], zero padded */
  void* wT_bf16;       /* bf16 [C][taps][rows] or NULL */
  int32_t rows, taps, C, Kp;
} vlsfr_cast_entry;
int vlsfr_cast_weights(const vlsfr_cast_entry* entries, int32_t n, void* stream);
/* fp32 NCHW image [N,3,H,W] -> bf16 im2col rows [N*Ho*Wo][32] of the 3x3 pad-1 stem, stride 1
 * (iResNet, resnet_arcface.py:74) or 2 (MobileFaceNet, mobilefacenet_def.py:80); k=(r*3+s)*3+c */
int vlsfr_stem_im2col(const float* x_nchw, void* out, int32_t N, int32_t H, int32_t W, int32_t stride,
                      void* stream);
int vlsfr_unpad_add(const float* src, float* dst, int32_t rows, int32_t Ksrc, int32_t Kdst, void* stream);
/* The loader's image transform on the device.  Replaces util/lmdb_loader.py:109-127 (MultiLMDBDataset.__getitem__
 * after cv2.imdecode) and :206-233 (PairLMDBDataset): raw = decoded uint8 pixels [N][H][W][C], C = 3 (BGR, the
 * cv2 order) or 1 (grey: replicated to three planes, :112-117); flip = uint8 [N] (non-zero: cv2.flip(img, 1), drawn
 * by the host with p = 0.5, :109) or NULL; out = fp32 [N][3][H][W] = (v - 127.5) * 0.0078125 (exact in fp32). */
int vlsfr_faces_normalize(const uint8_t* raw, const uint8_t* flip, float* out, int32_t N, int32_t H, int32_t W,
                          int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * 7. iResNet backbone executor (host object, device work).  One call = one whole forward or
 *    backward pass of reference model/resnet_arcface.py:58-152 (IResNet.forward, training mode)
 *    enqueued on the caller's stream.  `params` / `grads` are host arrays of device pointers in
 *    the registration order of the reference module (named_parameters(): conv1.weight,
 *    bn1.weight, bn1.bias, prelu.weight, layer1.0.bn1.weight, ... fc.weight, fc.bias,
 *    features.weight, features.bias); `running` holds (running_mean, running_var) device
 *    pointers per BatchNorm in the same order (NULL array = do not update).  Convolution weights
 *    are fp32 in channels_last memory ([Cout][R][S][Cin]); gradients are ACCUMULATED (+=).
 *    wcache: bf16 operand copies (vlsfr_iresnet_prepare_weights, whenever the weights change);
 *    ctx: activations + statistics of one forward pass, consumed by the matching backward;
 *    scratch: transient.  All three are caller-allocated device buffers of the queried sizes.
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_iresnet vlsfr_iresnet;
int vlsfr_iresnet_create(const int32_t* layers /*[4]*/, int32_t feat_dim, int32_t batch, int32_t image_hw,
                         vlsfr_iresnet** out);
void vlsfr_iresnet_destroy(vlsfr_iresnet* n);
int32_t vlsfr_iresnet_num_params(const vlsfr_iresnet* n);
int32_t vlsfr_iresnet_num_bn(const vlsfr_iresnet* n);
size_t vlsfr_iresnet_wcache_bytes(const vlsfr_iresnet* n);
size_t vlsfr_iresnet_ctx_bytes(const vlsfr_iresnet* n);
size_t vlsfr_iresnet_scratch_bytes(const vlsfr_iresnet* n);
int vlsfr_iresnet_prepare_weights(const vlsfr_iresnet* n, const float* const* params, void* wcache, void* stream);
/* x_nchw: fp32 [B,3,HW,HW] (loader contract, util/lmdb_loader.py:127); emb_out: fp32 [B, feat_dim] */
int vlsfr_iresnet_forward(const vlsfr_iresnet* n, const float* x_nchw, const float* const* params,
                          float* const* running, const void* wcache, void* ctx, void* scratch, float* emb_out,
                          void* stream);
/* The same; keep_activations == 0: a pass whose activations nobody will read (the gallery network, no_grad): the executor then
 * skips writing the normalised inputs of the convolutions that apply their BatchNorm themselves (vlsfr_conv2d_fwd_bnin). */
int vlsfr_iresnet_forward_ex(const vlsfr_iresnet* n, const float* x_nchw, const float* const* params, float* const* running,
                             const void* wcache, void* ctx, void* scratch, float* emb_out, int32_t keep_activations, void* stream);
int vlsfr_iresnet_backward(const vlsfr_iresnet* n, const float* demb, const float* const* params,
                           float* const* grads, const void* wcache, void* ctx, void* scratch, void* stream);
/* The same pass, signalling when groups of parameter gradients are complete, so that a multi-GPU caller can
 * start reducing them while the rest of the backward pass still runs (parallel.py): stage_events (or NULL) is an
 * array of 5 events (vlsfr_event_create; entries may be NULL) recorded on `stream` in backward order —
 * [0] bn2 / fc / features, [1] layer4, [2] layer3, [3] layer2, [4] layer1 + stem. */
int vlsfr_iresnet_backward_staged(const vlsfr_iresnet* n, const float* demb, const float* const* params,
                                  float* const* grads, const void* wcache, void* ctx, void* scratch,
                                  void* const* stage_events, void* stream);
/* The same pass with the weight-gradient kernels on `side_stream`: the input-gradient chain alternates MFMA-bound and
 * HBM-bound kernels and nothing in it waits for a weight gradient, so those run beside it (ring of gradient buffers,
 * event-ordered; results identical up to the order of the fp32 atomics).  ring: vlsfr_iresnet_overlap_ring_bytes(n)
 * bytes of device memory; events: vlsfr_iresnet_overlap_events() events from vlsfr_event_create, owned by the caller and
 * reusable across calls issued on the same pair of streams.  On return the main stream is ordered behind every
 * side-stream kernel.  stage_events as in vlsfr_iresnet_backward_staged (may be NULL). */
size_t vlsfr_iresnet_overlap_ring_bytes(const vlsfr_iresnet* n);
int32_t vlsfr_iresnet_overlap_events(void);
int vlsfr_iresnet_backward_overlap(const vlsfr_iresnet* n, const float* demb, const float* const* params,
                                   float* const* grads, const void* wcache, void* ctx, void* scratch, void* ring,
                                   size_t ring_bytes, void* const* stage_events, void* side_stream,
                                   void* const* events, void* stream);
/* Teacher-forced execution of IBasicBlocks k0 .. k1 - 1 (resnet_arcface.py:44-55) for parity tests of the executor's wiring:
 * the same kernels and context slots as vlsfr_iresnet_forward / _backward, but from an activation x_in (bf16 NHWC, the
 * tensor block k0 reads: [B, H, H, cin]) and an output gradient dout (bf16 NHWC [B, Ho, Ho, planes] of block k1 - 1)
 * supplied by the caller.  out / dx: the range's output and input gradient (bf16 NHWC); parameter gradients are
 * accumulated (+=) into grads as in the full pass.  vlsfr_iresnet_block_info: info = {cin, planes, H, Ho, stride,
 * has_downsample, index of bn1.weight in the parameter table, number of blocks}. */
int vlsfr_iresnet_block_info(const vlsfr_iresnet* n, int32_t k, int32_t* info /*[8]*/);
int vlsfr_iresnet_forward_blocks(const vlsfr_iresnet* n, int32_t k0, int32_t k1, const void* x_in, const float* const* params,
                                 float* const* running, const void* wcache, void* ctx, void* scratch, void* out, void* stream);
int vlsfr_iresnet_backward_blocks(const vlsfr_iresnet* n, int32_t k0, int32_t k1, const void* dout, const float* const* params,
                                  float* const* grads, const void* wcache, void* ctx, void* scratch, void* dx, void* stream);

/* ------------------------------------------------------------------------------------------
 * 7b. MobileFaceNet backbone executor: same contract as section 7 for reference
 *     model/mobilefacenet_def.py:77-123 (MobileFaceNet.forward, training mode).  Parameter order =
 *     registration order of the reference module (conv1.conv.weight, conv1.bn.weight, conv1.bn.bias,
 *     conv1.prelu.weight, dw_conv1.*, blocks.0.conv.0.weight, ... linear1.bn.bias); depthwise weights
 *     are used in place as fp32 [C][k*k].
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_mobilenet vlsfr_mobilenet;
int vlsfr_mobilenet_create(int32_t feat_dim, int32_t batch, int32_t image_hw, vlsfr_mobilenet** out);
void vlsfr_mobilenet_destroy(vlsfr_mobilenet* n);
int32_t vlsfr_mobilenet_num_params(const vlsfr_mobilenet* n);
int32_t vlsfr_mobilenet_num_bn(const vlsfr_mobilenet* n);
size_t vlsfr_mobilenet_wcache_bytes(const vlsfr_mobilenet* n);
size_t vlsfr_mobilenet_ctx_bytes(const vlsfr_mobilenet* n);
size_t vlsfr_mobilenet_scratch_bytes(const vlsfr_mobilenet* n);
int vlsfr_mobilenet_prepare_weights(const vlsfr_mobilenet* n, const float* const* params, void* wcache, void* stream);
int vlsfr_mobilenet_forward(const vlsfr_mobilenet* n, const float* x_nchw, const float* const* params,
                            float* const* running, const void* wcache, void* ctx, void* scratch, float* emb_out,
                            void* stream);
int vlsfr_mobilenet_backward(const vlsfr_mobilenet* n, const float* demb, const float* const* params,
                             float* const* grads, const void* wcache, void* ctx, void* scratch, void* stream);
/* Teacher-forced execution of units u0 .. u1 - 1, u0 >= 1 (a unit = conv -> BN [-> PReLU] [+ residual]; a BottleNeck,
 * mobilefacenet_def.py:27-52, is three consecutive units), as vlsfr_iresnet_forward_blocks: x_in = output of unit u0 - 1,
 * dout = gradient of the output of unit u1 - 1, both bf16 NHWC.  vlsfr_mobilenet_unit_info: info = {kind (0 stem,
 * 1 pointwise, 2 depthwise), Cin, Cout, H, Ho, residual source unit or -1, index of the conv weight in the parameter
 * table, number of units}. */
int vlsfr_mobilenet_unit_info(const vlsfr_mobilenet* n, int32_t k, int32_t* info /*[8]*/);
int vlsfr_mobilenet_forward_units(const vlsfr_mobilenet* n, int32_t u0, int32_t u1, const void* x_in, const float* const* params,
                                  float* const* running, const void* wcache, void* ctx, void* scratch, void* out, void* stream);
int vlsfr_mobilenet_backward_units(const vlsfr_mobilenet* n, int32_t u0, int32_t u1, const void* dout, const float* const* params,
                                   float* const* grads, const void* wcache, void* ctx, void* scratch, void* dx, void* stream);

/* ------------------------------------------------------------------------------------------
 * 7c. torchvision-style ResNet executor (Bottleneck): same contract as section 7 for reference
 *     model/resnet_std.py:55-104 (Bottleneck.forward) and :106-206 (ResNet: 7x7/2 stem, BN, ReLU, 3x3/2 max-pool,
 *     four stages, flatten -> fc -> BatchNorm1d -> normalise), training mode; `--net_type r50` = layers {3,4,6,3}
 *     (resnet_std.py:242-251), 224 x 224 input (fc is 2048*7*7 wide, :140).  Parameter order = registration order
 *     (conv1.weight, bn1.*, layer1.0.conv1.weight, layer1.0.bn1.*, conv2, bn2, conv3, bn3, downsample.0.weight,
 *     downsample.1.*, ..., fc.weight, fc.bias, features.weight, features.bias); features.weight is trainable here.
 * ---------------------------------------------------------------------------------------- */
typedef struct vlsfr_resnet vlsfr_resnet;
int vlsfr_resnet_create(const int32_t* layers /*[4]*/, int32_t feat_dim, int32_t batch, int32_t image_hw,
                        vlsfr_resnet** out);
void vlsfr_resnet_destroy(vlsfr_resnet* n);
int32_t vlsfr_resnet_num_params(const vlsfr_resnet* n);
int32_t vlsfr_resnet_num_bn(const vlsfr_resnet* n);
size_t vlsfr_resnet_wcache_bytes(const vlsfr_resnet* n);
size_t vlsfr_resnet_ctx_bytes(const vlsfr_resnet* n);
size_t vlsfr_resnet_scratch_bytes(const vlsfr_resnet* n);
int vlsfr_resnet_prepare_weights(const vlsfr_resnet* n, const float* const* params, void* wcache, void* stream);
int vlsfr_resnet_forward(const vlsfr_resnet* n, const float* x_nchw, const float* const* params,
                         float* const* running, const void* wcache, void* ctx, void* scratch, float* emb_out,
                         void* stream);
int vlsfr_resnet_backward(const vlsfr_resnet* n, const float* demb, const float* const* params,
                          float* const* grads, const void* wcache, void* ctx, void* scratch, void* stream);
/* Its operator kernels beyond sections 5 / 6 (NHWC bf16): the 7x7 / stride 2 / pad 3 stem as im2col rows
 * [N*Ho*Wo][160] (k = (r*7+s)*3+c, 147 taps zero padded; resnet_std.py:127-128), nn.MaxPool2d(3, 2, 1) forward and
 * backward (:131; the gradient goes to the first maximum of a window, as torch's does) and the backward of the
 * ReLU that follows the residual add (:103; in_nchw: dy and y in the [n][c][hw] flatten order, dx NHWC). */
int vlsfr_stem7_im2col(const float* x_nchw, void* out, int32_t N, int32_t H, int32_t W, void* stream);
int vlsfr_maxpool3x3s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int vlsfr_maxpool3x3s2_bwd(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W,
                           int32_t C, void* stream);
int vlsfr_relu_bwd_bf16(const void* dy, const void* y, void* dx, int64_t M, int32_t C, int32_t HW, int32_t in_nchw,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * 8. Parameter sweeps (device), one launch over all tensors.
 *    vlsfr_sgd_nesterov replaces torch.optim.SGD.step as built by optim/optimizer.py:148-150
 *    (momentum buffer starts at zero == torch's first-step clone); vlsfr_ema replaces
 *    FFC._momentum_update_gallery (ffc.py:139-145).
 *    table_dev: int64 device array, one row per chunk (<= 65536 elements, 16-byte aligned starts):
 *      sgd: {param, grad, momentum_buffer, count}      ema: {gallery, probe, count}
 * ---------------------------------------------------------------------------------------- */
int vlsfr_sgd_nesterov(const int64_t* table_dev, int32_t n_chunks, float lr, float momentum,
                       float weight_decay, int32_t nesterov, void* stream);
int vlsfr_ema(const int64_t* table_dev, int32_t n_chunks, float m, void* stream);
/* The two forward passes of a step (ffc.py:264-267) update every BatchNorm's running statistics one after the other
 * (torch.nn.BatchNorm2d, momentum 0.1: r <- (1 - m) r + m s).  When the passes run side by side on their own HIP
 * streams, each is handed a ZEROED table in place of the running buffers, so the unchanged kernels leave d_k = m s_k
 * there; this call then applies both updates in pass order: r <- (1 - m) ((1 - m) r + d_0) + d_1.
 * table_dev rows: {running, d_0, d_1, count}. */
int vlsfr_running_merge(const int64_t* table_dev, int32_t n_chunks, float momentum, void* stream);

/* ------------------------------------------------------------------------------------------
 * 9. Measurement support: while enabled, every launch of the profiled kernel families is bracketed by HIP
 *    events on its own stream — family 0: conv_igemm (forward + input gradient), 1: conv_wgrad, 2: head_sweep,
 *    3: the conv_igemm launches whose epilogue also accumulates a BatchNorm-backward reduction.
 *    vlsfr_profile_collect sums elapsed time, ALGORITHMIC FLOPs (convolutions: 2 * forward output positions *
 *    Cout * R*S*Cin, so a stride-2 input gradient is priced at its forward's positions and the stem at its 27 real
 *    taps; head sweep: 4 * B * Q * D) and launch count.  Used by bench.py's roofline leg.
 * ---------------------------------------------------------------------------------------- */
void vlsfr_profile_enable(int32_t on);
/* tuning switches for A/B measurements: "conv_glds" (1 = LDS-DMA pipelined conv kernel, default;
 * 0 = register-staged kernel) */
int vlsfr_set_option(const char* name, int32_t value);
int vlsfr_profile_collect(int32_t family, double* total_ms, double* total_flops, int64_t* launches);
/* what an empty event bracket reads on `stream` (microseconds, mean of 64): the per-launch overhead contained in the
 * totals above; synchronises the stream (measurement support only) */
double vlsfr_profile_event_overhead_us(void* stream);
/* Diagnostics: device buffer of 2 x 64 x 16 int64 receiving shader-clock stamps of one workgroup of the diagnostic
 * instantiations of the convolution kernels (scripts/hw4_trace.py, hp8_trace.py, conv_trace.py: per phase / per step of a
 * k-tile, and the 100 MHz real-time clock around the loop); nullptr switches the stamps off. */
int vlsfr_conv_trace(void* device_buffer);
void vlsfr_profile_reset(void);
/* brackets NOT taken since the last reset (the 65 536-event pool was exhausted, or an event could not be created): when this
 * is not 0 the totals of vlsfr_profile_collect undercount and the caller should say so */
int64_t vlsfr_profile_dropped(void);

/* ------------------------------------------------------------------------------------------
 * 10. Stream ordering helpers (HIP events without timing) for callers that spread the step over several
 *     streams: the multi-GPU step waits on the per-stage events of vlsfr_iresnet_backward_staged from its
 *     communication stream.  The reference has no counterpart (single default stream).
 * ---------------------------------------------------------------------------------------- */
int vlsfr_event_create(void** event);
void vlsfr_event_destroy(void* event);
int vlsfr_event_record(void* event, void* stream);
int vlsfr_stream_wait_event(void* stream, void* event);

#ifdef __cplusplus
}
#endif
#endif /* VLSFR_H */
