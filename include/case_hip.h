/*
 * case_hip.h -- C ABI of libcase_hip.so, the MI355X (gfx950) kernel library behind the CaSE_RG
 * encoder-decoder hot path.
 *
 * The reference (PengjieRen/CaSE_RG) has no native layer: every operation below replaces an ATen op
 * chain called from the reference's Python modules; the chain is cited as <file>:<line> relative to
 * the reference root.  Host code (case_rg_amd/, Python) binds these entry points with ctypes
 * (case_rg_amd/_abi.py); INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers owned by the caller
 *     (PyTorch caching allocator).  The library never allocates, frees, retains a pointer or
 *     synchronises: every call only enqueues work on the caller's stream, so the whole path is
 *     hipGraph-capturable.
 *   - return 0 on success, a negative CASE_E_* code otherwise; case_last_error() gives the text.
 *     Launch errors are read with hipGetLastError() right after enqueue (no sync).
 *   - dtype arguments are case_dtype_t: activations are f32 (parity mode, exact-f32 MFMA) or bf16
 *     (throughput mode, bf16 MFMA with f32 accumulate); parameters, statistics, losses and
 *     gradients of parameters are always f32.
 *   - masks are uint8 (torch.bool storage), 1 = VALID token unless a name says "pad".
 */
#ifndef CASE_HIP_H
#define CASE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* case_stream_t; /* hipStream_t */

typedef enum { CASE_F32 = 0, CASE_BF16 = 1 } case_dtype_t;

enum {
  CASE_OK = 0,
  CASE_E_ARG = -1,         /* bad shape / stride / null pointer */
  CASE_E_UNSUPPORTED = -2, /* combination not built */
  CASE_E_LAUNCH = -3       /* hipGetLastError() after enqueue */
};

/* ABI generation.  case_version() returns the CASE_ABI_VERSION the library was built from; a binder compares it with the header it was
 * written against (case_rg_amd/_abi.py refuses any other library).  Bumped with every struct or signature change:
 *   100 round 1 | 200 round 2 (case_gemm_dw_bias, decode, optimizer) | 300 round 3 (CaseOptTensor 56 -> 64 bytes, K16 / K17)
 *   400 round 4 (K18 / K19 resident attention, case_attention_bwd scratch = 2 N heads Lq floats, workspace query, reserved CUs)
 *   500 round 5 (K21 case_attention_decode_mqa, K22 case_pointer_attend_decode, K23 case_pointer_head_decode, case_gemm_ln)
 *   600 round 6 (CaseStepState: the per-step scalars -- dropout counter base, Adam step size / bias correction -- may live in caller-owned
 *       DEVICE memory instead of riding in the arguments, so that a hipGraph-captured training step draws new masks and takes the right
 *       Adam step on every replay; `state` members / arguments on every dropout site and on case_optim_adam_ema; case_step_advance;
 *       K8 case_interaction_fwd). */
#define CASE_ABI_VERSION 600
int case_version(void);
/* what the build contains, as a bit mask */
enum {
  CASE_FEAT_GEMM_256 = 1u << 0,        /* 256 x 256 eight-wave GEMM tiling (+ case_gemm_dw_bias) */
  CASE_FEAT_GEMM_SMALL = 1u << 1,      /* 64 x 64 small-problem tiling */
  CASE_FEAT_ENCODER_CHAIN = 1u << 2,   /* K16 case_encoder_chain */
  CASE_FEAT_ATTN_SCORES = 1u << 3,     /* K17 case_attention_scores_* / case_attention_product */
  CASE_FEAT_ATTN_DECODE = 1u << 4,     /* case_attention_decode */
  CASE_FEAT_OPTIM = 1u << 5,           /* K15 case_optim_* */
  CASE_FEAT_ATTN_RESIDENT = 1u << 6,   /* K18 / K19 behind case_attention_fwd / _bwd */
  CASE_FEAT_RESERVED_CUS = 1u << 7,    /* case_set_reserved_cus */
  CASE_FEAT_GEMM_DW_SLABS = 1u << 8,   /* case_gemm_dw_slabs: atomics-free, run-to-run deterministic split-K weight gradients */
  CASE_FEAT_DECODER_CHAIN = 1u << 9,   /* (reserved: K20 case_decoder_chain, retired in round 5 -- never set) */
  CASE_FEAT_ATTN_DECODE_MQA = 1u << 10, /* K21 case_attention_decode_mqa */
  CASE_FEAT_POINTER_DECODE = 1u << 11,  /* K22 case_pointer_attend_decode / case_additive_key_exp */
  CASE_FEAT_POINTER_HEAD = 1u << 12,    /* K23 case_pointer_head_decode */
  CASE_FEAT_GEMM_LN = 1u << 13,         /* case_gemm_ln: LayerNorm prologue of the small-problem GEMM */
  CASE_FEAT_STEP_STATE = 1u << 14,      /* ABI 600: CaseStepState on the dropout sites and the optimizer, case_step_advance */
  CASE_FEAT_INTERACTION = 1u << 15,     /* K8 case_interaction_fwd: the dual co-attention as two kernels */
  CASE_FEAT_ATTN_DECODE_APPEND = 1u << 16, /* case_attention_decode_append: the greedy step's cache append inside the attention launch */
  CASE_FEAT_LINEAR_SKINNY = 1u << 17      /* case_linear_skinny */
};
uint32_t case_abi_features(void);
const char* case_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Device-resident step state (round 6).  A training step of common/CumulativeTrainer.py:52-78 has three kinds of scalars that change
 * from step to step: the position of the dropout counter stream (every nn.Dropout / F.dropout site: common/TransformerEncoder.py:68,72,75
 * ...), the learning rate the scheduler set (CaSE/Run.py:28) and Adam's bias corrections (torch.optim.Adam, CaSE/Run.py:27).  As kernel
 * ARGUMENTS they are frozen into a captured hipGraph -- a replayed step would redraw the same masks.  Every entry point that takes
 * (seed, offset) or the Adam scalars therefore also takes a nullable `const CaseStepState* state`, a DEVICE pointer to 64 caller-owned
 * bytes that the kernels read when they run:
 *   - dropout sites hash (seed, offset + state->rng_base + element index): the caller numbers the sites of ONE step from offset 0 and
 *     moves rng_base by the step's consumption between steps (rng_base must stay even: the kernels hash element pairs);
 *   - case_optim_adam_ema takes step_size / bc2_sqrt from the state instead of the table entries.
 * state == NULL is the round-5 behaviour (arguments only).  The caller writes the struct -- a 64-byte hipMemcpyAsync from pinned memory
 * ahead of the step (exact for any scheduler), or case_step_advance on the stream (no host involvement per step).  The library only
 * reads it; like every pointer here it is never retained.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t rng_base; /* added to the `offset` of every dropout site that is given this state; even */
  float step_size;   /* lr / (1 - beta1^step)   -- as CaseOptTensor.step_size */
  float bc2_sqrt;    /* sqrt(1 - beta2^step)    -- as CaseOptTensor.bc2_sqrt */
  float lr;          /* the learning rate step_size was formed from (read by case_step_advance only) */
  int32_t step;      /* optimizer steps taken, INCLUDING the one these scalars belong to */
  uint64_t reserved[5];
} CaseStepState;     /* 64 bytes */
int case_sizeof_step_state(void);
/* One single-thread launch that moves the state to the next step: step += 1, rng_base += rng_stride (even), and, with the state's lr,
 * step_size = lr / (1 - beta1^step), bc2_sqrt = sqrt(1 - beta2^step), formed in double and rounded to f32 once (what
 * case_rg_amd/optim.py does on the host).  For replay loops that never return to the host; a scheduler that changes lr writes state->lr. */
int case_step_advance(CaseStepState* state, uint64_t rng_stride, double beta1, double beta2, case_stream_t stream);

/* Compute units the persistent kernels (256 x 256 GEMM, K16 .. K19) leave free -- the ONE piece of mutable library configuration:
 * with world_size > 1 RCCL's kernels must be resident beside them for the gradient all-reduce to overlap backward
 * (common/CumulativeTrainer.py:45-47).  0 .. 128, rounded so that the grids stay whole XCD rounds; initial value from the environment
 * variable CASE_RESERVE_CUS (default 0). */
int case_set_reserved_cus(int n);
int case_get_reserved_cus(void);

/* Caller-owned scratch of the entry points that need one, in bytes:
 *   CASE_WS_ATTENTION_SPLITKV  desc = CaseAttnDesc*, arg = ksplit     (workspace of case_attention_fwd_splitkv)
 *   CASE_WS_ATTENTION_BWD      desc = CaseAttnDesc*                   (the `delta` argument of case_attention_bwd)
 *   CASE_WS_OPTIM_SUMSQ        arg = nchunks                          (the `partials` argument of case_optim_sumsq)
 *   CASE_WS_ENCODER_CHAIN_PACK                                        (the packed weights of case_encoder_chain_pack)
 *   CASE_WS_GEMM_DW_SLABS      desc = CaseGemmDesc*                   (the `slabs` argument of case_gemm_dw_slabs; 0 = no split)
 * negative = CASE_E_ARG. */
typedef enum { CASE_WS_ATTENTION_SPLITKV = 1, CASE_WS_ATTENTION_BWD = 2, CASE_WS_OPTIM_SUMSQ = 3, CASE_WS_ENCODER_CHAIN_PACK = 4, CASE_WS_GEMM_DW_SLABS = 5 } case_workspace_kind_t;
int64_t case_workspace_bytes(int32_t kind, const void* desc, int64_t arg);

/* ---------------------------------------------------------------------------------------------
 * K3  strided-batched GEMM on MFMA:   C = epilogue(alpha * op(A) op(B))
 * replaces every nn.Linear / MHA in_proj / out_proj / bmm on the path:
 *   common/TransformerEncoder.py:67,72  common/TransformerDecoder.py:77,81,86
 *   common/TransformerBlock.py:26,28-29  common/Interaction.py:36,50-54  common/BilinearAttention.py:31,34,57
 *   CaSE/Model.py:34,36,43,242  (and their autograd backward)
 * A is M x K, B is N x K ("NT": C = A B^T) unless *_kmajor says the operand is stored K x M / K x N.
 * Batch index b in [0, batch1*batch2): b1 = b / batch2, b2 = b % batch2 (b2 = attention head);
 * operand offset = b1*s?1 + b2*s?2 elements.
 * ------------------------------------------------------------------------------------------- */
enum {
  CASE_EPI_BIAS_COL = 1,   /* + bias_col[n] (f32)                                   */
  CASE_EPI_BIAS_ROW = 2,   /* + bias_row[b*M + m] (f32)  (Interaction rank-1 terms) */
  CASE_EPI_GELU = 4,       /* erf GELU; pre-activation also stored to aux_out if non-null */
  CASE_EPI_RELU = 8,
  CASE_EPI_RESIDUAL = 16,  /* + aux[m, n]                                           */
  CASE_EPI_MUL_DGELU = 32, /* * gelu'(aux[m, n])   (backward through GELU)          */
  CASE_EPI_MUL_DRELU = 64, /* * (aux[m, n] > 0)    (backward through ReLU; aux = activation output) */
  CASE_EPI_ATOMIC = 128,   /* split-K: atomicAdd f32 into pre-zeroed C (out_dtype must be f32) */
  CASE_EPI_DROPOUT = 256   /* * keep(seed, offset + (b*M + m)*N + n) / (1 - drop_p); nn.Dropout / F.dropout sites:
                              TransformerEncoder.py:68,72,75 TransformerDecoder.py:78,82,86,88 TransformerBlock.py:27-28
                              CaSE/Model.py:34.  The same flag regenerates the mask in the backward GEMM. */
};
/* epilogue order: alpha*acc + bias -> GELU|RELU -> MUL_DGELU|MUL_DRELU -> DROPOUT -> + RESIDUAL */

typedef struct {
  int64_t M, N, K;
  int64_t lda, ldb, ldc, ld_aux; /* leading dimensions in elements */
  int64_t batch1, batch2;
  int64_t sa1, sa2, sb1, sb2, sc1, sc2, saux1, saux2;
  int32_t a_kmajor, b_kmajor;
  int32_t in_dtype;  /* dtype of A, B, aux, aux_out */
  int32_t out_dtype; /* dtype of C */
  int32_t epilogue;  /* CASE_EPI_* mask */
  int32_t split_k;   /* >= 1 */
  int32_t tile;      /* 0 = pick the tiling per call (cost model); 128 = the 128x128 tiling; 256 = the 256x256 persistent tiling,
                        64 = the small-problem tiling (whole K panel in LDS) whenever the call is eligible (otherwise 128x128) */
  float alpha;
  float drop_p;
  uint64_t seed, offset;
  const CaseStepState* state; /* nullable: offset += state->rng_base on the device (ABI 600) */
} CaseGemmDesc;

int case_gemm(const CaseGemmDesc* d, const void* A, const void* B, void* C, const float* bias_col,
              const float* bias_row, const void* aux, void* aux_out, case_stream_t stream);
/* LayerNorm as a PROLOGUE of the small-problem GEMM (round 5): C = epilogue(LN(A) B^T), ln_out = LN(A) -- the greedy step's
 * LN1 -> QKV, LN2 -> cross-attention query and LN3 -> feed-forward pairs (common/TransformerDecoder.py:76-89 at one position per sequence)
 * as one launch each.  d describes the GEMM as for case_gemm (bf16, k-contiguous A with K = lda = 512, M and N multiples of 64, unsplit,
 * unbatched; epilogue words BIAS_COL / GELU / RELU / RESIDUAL); gamma / beta f32 [512]; ln_out [M, 512] bf16 or null.  The statistics are
 * the two-pass f32 mean / variance of case_layernorm_fwd.  CASE_E_UNSUPPORTED for anything else (run case_layernorm_fwd + case_gemm). */
int case_gemm_ln(const CaseGemmDesc* d, const void* A, const float* gamma, const float* beta, float eps, void* ln_out, const void* B, void* C,
                 const float* bias_col, const void* aux, case_stream_t stream);

/* case_gemm owns three tilings: 128x128 (every shape / dtype / batch), 256x256 (bf16, M % 256 == N % 256 == 0,
 * K % 64 == 0, unbatched, 16-byte aligned; eight waves, operands by LDS-DMA, persistent: csrc/gemm8w.inc) and 64x64 for small
 * problems (bf16, multiples of 64, <= 10 K tiles per split, the 128x128 grid smaller than the chip: csrc/gemm_small.inc).  CaseGemmDesc.tile selects per call; results
 * of the tilings agree to f32 summation order.  case_gemm_tile_for() returns the tile edge (64, 128 or 256) case_gemm would
 * launch for exactly these arguments (or a negative CASE_E_* code): a pure function of its arguments, no launch, no state --
 * bench.py uses it to attribute each launch to the kernel name rocprofv3 reports. */
/* Weight-gradient GEMM that also produces the bias gradient (replaces a separate pass over dY, case_colsum; reference:
 * autograd of nn.Linear, e.g. common/TransformerBlock.py:13-14): C[M, N] += op(A) op(B) as case_gemm with the bare ATOMIC epilogue,
 * and d_bias[m] += sum_k op(A)[m, k] (f32 [M], pre-zeroed, atomics).  Only for calls the 256x256 tiling takes with a k-major A
 * (case_gemm_tile_for(...) == 256, d->a_kmajor): CASE_E_UNSUPPORTED otherwise, and the caller runs case_gemm + case_colsum. */
int case_gemm_dw_bias(const CaseGemmDesc* d, const void* A, const void* B, void* C, float* d_bias, case_stream_t stream);
/* The same weight-gradient GEMM without atomics on C: every K split stores its partial [M, N] f32 tile into its own slab of the
 * caller's workspace (16-byte aligned, >= case_gemm_dw_slab_bytes(d) bytes, free again when the stream has passed the call) and a
 * second launch adds the slabs to C in split order -- C[M, N] += op(A) op(B), bit-identical from run to run.  For the small,
 * deeply split outputs (a 512 x 512 weight gradient over 122,880 tokens = 64 splits) it is also ~2x faster: 64 MiB of f32 red ops
 * retire at ~1.3 TB/s, plain stores + one read at HBM rate.  d_bias may be NULL; otherwise as in case_gemm_dw_bias (atomics, [M]).
 * Needs the 256x256 tiling, split_k > 1 after clamping, ldc == N; CASE_E_UNSUPPORTED otherwise.  case_gemm_dw_slab_bytes returns
 * 0 when the clamped split is 1. */
int case_gemm_dw_slabs(const CaseGemmDesc* d, const void* A, const void* B, void* C, float* d_bias, void* slabs, int64_t slab_bytes,
                       case_stream_t stream);
int64_t case_gemm_dw_slab_bytes(const CaseGemmDesc* d);

int case_gemm_tile_for(const CaseGemmDesc* d, const void* A, const void* B, const void* C, const float* bias_col,
                       const void* aux, const void* aux_out);

/* ---------------------------------------------------------------------------------------------
 * K1  embedding gather * sqrt(H) + sinusoid position (+ dropout)
 *   common/TransformerSeqEncoderDecoder.py:21,36  common/PositionalEmbedding.py:44-48  CaSE/Model.py:21,67
 * ids int64 [rows]; position of a row = row % seq_len; table f32 [V, H]; pe f32 [>= seq_len, H].
 * bwd: d_table[id] += scale * d_out[row] (f32 atomics; id 0 = padding_idx receives nothing).
 * ------------------------------------------------------------------------------------------- */
int case_embed_pos_fwd(const int64_t* ids, const float* table, const float* pe, void* out, int64_t rows,
                       int64_t seq_len, int64_t H, int64_t vocab, float scale, float drop_p, uint64_t seed,
                       uint64_t offset, const CaseStepState* state, int32_t dtype, case_stream_t stream);
int case_embed_pos_bwd(const int64_t* ids, const void* d_out, float* d_table, int64_t rows, int64_t H,
                       int64_t vocab, float scale, float drop_p, uint64_t seed, uint64_t offset, const CaseStepState* state,
                       int32_t dtype, case_stream_t stream);

/* stand-alone PositionalEmbedding.forward (common/PositionalEmbedding.py:44-48): y = x*scale + pe[row % seq_len]
 * (pe may be null: plain scaling, used as its backward) */
int case_scale_add_rows(const void* x, const float* pe, void* y, int64_t rows, int64_t seq_len, int64_t H, float scale,
                        int32_t dtype, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  LayerNorm (eps inside sqrt, affine), optional fused input add: y = LN(x + x2)
 *   common/TransformerEncoder.py:66,70  common/TransformerDecoder.py:76,80,84  common/TransformerBlock.py:25,28
 *   CaSE/Model.py:69,84,209-210
 * mean / rstd f32 [rows] are saved for backward.  bwd accumulates d_gamma / d_beta with f32 atomics
 * into pre-zeroed buffers; dx is also the gradient of x2 when the add was fused.
 * ------------------------------------------------------------------------------------------- */
int case_layernorm_fwd(const void* x, const void* x2, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype, case_stream_t stream);
/* dx_add (nullable, same layout as dx): a second gradient of the same input -- its residual use, as in
 * x + MHA(LN(x)) of common/TransformerBlock.py:26-27 -- added to dx in the same pass. */
int case_layernorm_bwd(const void* dy, const void* x, const void* x2, const float* gamma, const float* mean,
                       const float* rstd, void* dx, const void* dx_add, float* d_gamma, float* d_beta, int64_t rows,
                       int64_t cols, int32_t dtype, case_stream_t stream);
/* The backward of LN(dropout(x W^T + b) + r) -- out-projection / feed-forward followed by a LayerNorm, common/TransformerEncoder.py:68-69,
 * 72-75, common/TransformerBlock.py:27-28 -- with TWO outputs: dx (the gradient of the LayerNorm's input = of the residual r) and
 * dx_dropped = mask * dx / (1 - p), the gradient of the Linear's pre-dropout output that its dX / dW GEMMs read.  The mask is case_dropout's
 * (element index row * cols + column behind (seed, offset)) applied to the ROUNDED dx: the same bits as case_layernorm_bwd followed by
 * case_dropout, one tensor pass less.  Rows of k x 64 lanes x 16 bytes (k <= 8), no x2 / dx_add; CASE_E_UNSUPPORTED otherwise.
 * `offset` must be EVEN (the mask is drawn per element pair; CASE_E_ARG otherwise -- case_dropout itself accepts odd offsets). */
int case_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                               void* dx_dropped, float* d_gamma, float* d_beta, int64_t rows, int64_t cols, float p, uint64_t seed,
                               uint64_t offset, const CaseStepState* state, int32_t dtype, case_stream_t stream);
/* The backward of LN(G), G = concat5(E, A1, A2) = [E | A1 | A2 | E o A1 | E o A2] with padded rows zeroed (common/Interaction.py:65-72 feeding
 * common/TransformerBlock.py:25), fused with the backward of the concatenation: dE, dA1, dA2 [rows, H] come out directly, dG (5H wide)
 * is never written.  x = G as the forward wrote it; dx_add (nullable, [rows, 5H]) a second gradient of G (its residual use, block :27).
 * dG is rounded to bf16 before the products, as the two-kernel path (case_layernorm_bwd + case_concat5_bwd) stores it.  bf16, H = 512
 * only; CASE_E_UNSUPPORTED otherwise. */
int case_layernorm_bwd_concat5(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dx_add,
                               const void* e, const void* a1, const void* a2, const uint8_t* row_valid, void* de, void* da1, void* da2,
                               float* d_gamma, float* d_beta, int64_t rows, int64_t H, int32_t dtype, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4/K5/K6 (softmax stage) and the masked softmaxes of K7/K8/K10/K11
 *   torch MHA softmax (common/TransformerEncoder.py:67, TransformerDecoder.py:77,81, TransformerBlock.py:26),
 *   common/Interaction.py:43-47, common/BilinearAttention.py:16-19, CaSE/Model.py:34,39
 * x is [outer, inner, R, C] contiguous; softmax over C.  col_valid u8 [outer, C] (or null), row_valid u8
 * [outer, R] (or null), causal: column > row masked.  A row with no admissible column yields exact 0
 * (the reference's NaN -> masked_fill(0) path) and zero gradient.
 * p_out: probabilities (saved for backward); y_out: after dropout (may alias p_out when drop_p == 0).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t outer, inner, R, C;
  int32_t causal;
  int32_t in_dtype, out_dtype;
  float drop_p;
  uint64_t seed, offset;
  const CaseStepState* state; /* nullable: offset += state->rng_base on the device (ABI 600) */
} CaseSoftmaxDesc;

int case_softmax_fwd(const CaseSoftmaxDesc* d, const void* x, const uint8_t* col_valid, const uint8_t* row_valid,
                     void* p_out, void* y_out, case_stream_t stream);
int case_softmax_bwd(const CaseSoftmaxDesc* d, const void* dy, const void* p, void* dx, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4 / K5 / K6  fused multi-head attention (bf16): softmax(scale * Q K^T + masks) V without materialising the scores
 *   nn.MultiheadAttention call sites: common/TransformerEncoder.py:67, common/TransformerBlock.py:26 (self-attention
 *   with key padding, head_dim 64 and 5H/8 = 320), common/TransformerDecoder.py:77 (causal + key padding), :81 (cross
 *   attention over the S = P*Lp token memory).
 * q / k / v point at the first head's columns inside the (packed) projection tensors: element (n, l, head, j) is at
 * base[n*s? + l*ld? + head*head_dim + j].  key_valid u8 [N, Lk] (1 = token) or null; causal masks key > query.
 * out bf16 [N, Lq, heads*head_dim] (row stride ldo, sequence stride so); lse f32 [N, heads, Lq] is saved for backward.
 * Dropout acts on the probabilities with the same counter RNG / element index as case_softmax_fwd on [N, heads, Lq, Lk].
 * Backward: delta f32 [N, heads, Lq] is scratch (rowsum(dO * O), written by the call); dq / dk / dv are bf16 slices of
 * the gradient of the packed projections, addressed with the same strides as q / k / v.
 * case_attention_supported(head_dim) != 0 tells whether a head size has a fused FORWARD (32, 64, 96, 160 = 5 x 256 / 8, 320 = 5 x 512 / 8,
 * 480 = 5 x 768 / 8), case_attention_bwd_supported whether it also has a fused backward (32, 64, 96, 160; round 6 dropped 320 / 480, whose
 * training path is K17's saved-probability form); other sizes use the unfused GEMM + softmax path.  Head sizes above
 * 128 split the head dim over 2-3 waves per block of 32 rows (partial score tiles are exchanged through LDS).
 * case_attention_fwd_splitkv: same result for long memories with few (sequence, head) pairs (cfg 5 cross-attention: 40
 * queries x 20 480 keys): the keys are cut into `ksplit` chunks handled by different workgroups, whose unnormalised partials
 * go through the caller-owned `workspace` (size from case_attention_splitkv_workspace) and a merge kernel.  Not causal.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t N, heads, Lq, Lk, head_dim;
  int64_t ldq, ldk, ldv, sq, sk, sv;
  int64_t ldo, so;
  int32_t causal;
  float scale, drop_p;
  uint64_t seed, offset;
  const CaseStepState* state; /* nullable: offset += state->rng_base on the device (ABI 600) */
} CaseAttnDesc;

int case_attention_supported(int64_t head_dim);
int case_attention_fwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid,
                       void* out, float* lse, case_stream_t stream);
int case_attention_splitkv_workspace(const CaseAttnDesc* d, int32_t ksplit, int64_t* bytes);
int case_attention_fwd_splitkv(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid,
                               void* out, float* lse, void* workspace, int64_t workspace_bytes, int32_t ksplit,
                               case_stream_t stream);
/* Decode step (greedy inference, CaSE/Model.py:94-123 with cached projections): ONE query per sequence (d->Lq == 1) against
 * Lk cached keys / values, no causal mask, no dropout, no LSE (nothing to differentiate).  HBM-bound streaming kernel, one
 * workgroup per (sequence, head); out [N, 1, heads*head_dim].  Rows without a valid key give exact zeros. */
int case_attention_decode_supported(int64_t head_dim);
int case_attention_decode(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid,
                          void* out, case_stream_t stream);
/* The same step WITH the cache append (round 6; common/TransformerDecoder.py:77 at one position: the reference re-projects the whole prefix, the
 * cached form writes position `pos`'s K / V projections into the layer's caches and attends positions <= pos): the keys / values of position
 * `pos` are read from new_k / new_v (row n at n * new_stride elements, head h at column h * head_dim) instead of the caches and stored into
 * k_cache / v_cache at that position by the workgroup that owns the (sequence, head) slice -- one launch instead of a strided copy + the attention.
 * key_valid must mark `pos` as the caller wants it attended.  CASE_FEAT_ATTN_DECODE_APPEND. */
int case_attention_decode_append(const CaseAttnDesc* d, const void* q, void* k_cache, void* v_cache, const void* new_k, const void* new_v,
                                 int64_t new_stride, int64_t pos, const uint8_t* key_valid, void* out, case_stream_t stream);
/* K21, the decode step's cross-attention over a LONG memory with absorbed projections (round 5): multi-query attention of 8 query rows
 * per item against the RAW memory rows [B, S, 512] (bf16), K = V = memory.  Replaces, for the layers of the passage-memory stack,
 * in_proj(q) -> case_attention_decode over the layer's cached K / V projections -> (common/TransformerDecoder.py:81-82 at one position,
 * CaSE/Model.py:94-123): with qp[b, h, :] = log2(e) / sqrt(d) * Wk_h^T q_h the scores q_h . K_h[j] equal qp_h . mem_j up to a term that
 * is constant over j, and sum_j p_hj V_h[j] = Wv_h (sum_j p_hj mem_j) + bv_h -- the caller applies Wv_h (and its bias) to the result.
 * One stream of S x 512 x 2 bytes per item instead of the two cached projections.  qp [B, 8, 512] bf16 (16-byte aligned), out [B, ldo] bf16
 * (head h at columns 512 h .. 512 h + 511), key_valid [B, S] bytes or null; items without a valid key give exact zeros.  nsplit key
 * ranges per item (case_attention_decode_mqa_splits picks one for (B, S); > 1 needs a workspace of case_attention_decode_mqa_workspace
 * bytes, 16-byte aligned).  bf16, 8 heads, width 512 only. */
int64_t case_attention_decode_mqa_workspace(int64_t B, int64_t S, int32_t nsplit);
int case_attention_decode_mqa_splits(int64_t B, int64_t S);
int case_attention_decode_mqa(const void* qp, const void* mem, const uint8_t* key_valid, void* out, int64_t B, int64_t S, int64_t ldo,
                              int32_t nsplit, void* workspace, int64_t workspace_bytes, case_stream_t stream);
int case_attention_bwd_supported(int64_t head_dim);
/* floats of scratch case_attention_bwd needs behind `delta` (2 N heads Lq: the resident single-pass backward of head_dim 64, K19,
 * keeps -lse / scale and -rowsum(dO * O) per query) */
int64_t case_attention_bwd_scratch_floats(const CaseAttnDesc* d);
int case_attention_bwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid,
                       const void* out, const float* lse, const void* dout, float* delta, void* dq, void* dk, void* dv,
                       case_stream_t stream);

/* K17  probabilities of the GEMM -> softmax -> GEMM attention path (the training path of the head_dim 320 blocks,
 * common/TransformerBlock.py:26; F.multi_head_attention_forward's bmm -> softmax -> dropout -> bmm) with the softmax in the score
 * GEMM: one workgroup holds 128 query rows x all Lk keys, so no score tensor reaches memory.
 *   case_attention_scores_fwd: p = softmax(scale q k^T | key_valid) as bf16 [N, heads, Lq, Lk]; with drop_p > 0 also
 *     p_dropped = keep ? p / (1 - drop_p) : 0 (same counter RNG / element index as case_softmax_fwd); replaces case_gemm (f32
 *     scores) + case_softmax_fwd.  Rows without a valid key give exact zeros.
 *   case_attention_scores_bwd: ds = p (g - rowsum(g p)), g = keep ? (dout v^T) / (1 - drop_p) : 0, bf16 [N, heads, Lq, Lk];
 *     replaces case_gemm (dP) + case_softmax_bwd.  dout is addressed with ldo / so, v with ldv / sv of the descriptor.
 * Scope (case_attention_scores_supported(d) != 0): bf16, head_dim a multiple of 64 and >= 128, Lk <= 384 and a multiple of 8, not causal. */
int case_attention_scores_supported(const CaseAttnDesc* d);
int case_attention_scores_fwd(const CaseAttnDesc* d, const void* q, const void* k, const uint8_t* key_valid, void* p,
                              void* p_dropped, case_stream_t stream);
int case_attention_scores_bwd(const CaseAttnDesc* d, const void* dout, const void* v, const void* p, void* ds,
                              case_stream_t stream);
/* The four batched products around those probabilities at head_dim 320 (F.multi_head_attention_forward's second bmm and the three
 * bmm of its backward): c[n, row, head, 0..319] = alpha * sum_k A[n, head][row, k] * b[n, k, head, 0..319], where A is the [L, L] matrix a
 * (a_transposed = 0: O = Pd V, dQ = alpha dS K) or its transpose (a_transposed = 1: dV = Pd^T dO, dK = alpha dS^T Q).  Element
 * (n, head, r, col) of a at a[n sa_seq + head sa_head + r lda + col]; k-row k of b at b[n sb_seq + head sb_head + k ldb + 0..319];
 * output row at c[n sc_seq + head sc_head + row ldc + 0..319].  M output rows, Kc = contraction length (a multiple of 64, >= 128).
 * One workgroup owns 128 rows x all 320 columns (a is read once); replaces case_gemm's batched 128x128 tiling for these shapes. */
typedef struct {
  int64_t N, heads, M, Kc, head_dim;
  int64_t lda, sa_seq, sa_head;
  int64_t ldb, sb_seq, sb_head;
  int64_t ldc, sc_seq, sc_head;
  int32_t a_transposed;
  float alpha;
} CaseAttnProductDesc;
int case_attention_product_supported(const CaseAttnProductDesc* d);
int case_attention_product(const CaseAttnProductDesc* d, const void* a, const void* b, void* c, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * elementwise / small reductions
 * ------------------------------------------------------------------------------------------- */
/* out = a + b  (residual adds: TransformerEncoder.py:68,75 etc.) */
int case_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, case_stream_t stream);
/* out = srcs[0] + ... + srcs[count - 1] (2 <= count <= 8; srcs is a HOST array of device pointers), summed in f32 and rounded once:
 * the gradient of a tensor with several consumers in one pass (what autograd forms with count - 1 binary adds, e.g. the Interaction
 * tensors of common/Interaction.py:32-63, each read by 2-5 products).  n a multiple of 16 bytes' worth of elements, 16-byte aligned. */
int case_add_n(const void* const* srcs, int32_t count, void* out, int64_t n, int32_t dtype, case_stream_t stream);
/* dropout with a counter RNG keyed by (seed, offset + element index); same call regenerates the mask in bwd */
int case_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset, const CaseStepState* state,
                 int32_t dtype, case_stream_t stream);
/* zero rows whose valid flag is 0: TransformerBlock.py:31, Interaction.py:68-70.  y may BE x (round 6): then only the invalid rows are
 * written and the valid ones are not touched at all (inference: a full-length batch costs one flag byte per vector instead of a copy). */
int case_mask_rows(const void* x, const uint8_t* row_valid, void* y, int64_t rows, int64_t cols, int32_t dtype,
                   case_stream_t stream);
/* column sums of a [rows, cols] matrix into f32 [cols] (bias gradients); out must be pre-zeroed */
int case_colsum(const void* x, float* out, int64_t rows, int64_t cols, int32_t dtype, case_stream_t stream);
/* dtype conversion (f32 master weights -> bf16 operands) */
int case_cast(const void* x, void* y, int64_t n, int32_t src_dtype, int32_t dst_dtype, case_stream_t stream);
/* y[r, :] = x[r, :] * w[:]  (Interaction: Ep * w3); bwd dx = dy * w, dw += colsum(dy * x) */
int case_scale_cols(const void* x, const float* w, void* y, int64_t rows, int64_t cols, int32_t dtype,
                    case_stream_t stream);
int case_scale_cols_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, int64_t rows,
                        int64_t cols, int32_t dtype, case_stream_t stream);
/* single-output linear y[r] = x[r, :] . w (+ b[0]): the scorers (CaSE/Model.py:161,201, Masque/Model.py:157) and the
 * rank-1 terms of the Interaction score (common/Interaction.py:36).  y, g f32 [rows]; w f32 [cols].
 * bwd: dx[r, :] = g[r] * w (dx may be null), dw[c] += sum_r g[r] x[r, c], db[0] += sum_r g[r] (dw / db pre-zeroed). */
int case_rowdot_fwd(const void* x, const float* w, const float* b, float* y, int64_t rows, int64_t cols, int32_t dtype,
                    case_stream_t stream);
int case_rowdot_bwd(const float* g, const void* x, const float* w, void* dx, float* dw, float* db, int64_t rows,
                    int64_t cols, int32_t dtype, case_stream_t stream);
/* few-output linear over a column-wise concatenation that is never formed (round 6): the greedy step's mixing logits, Linear(3H, 1 + nmem) on
 * [dec_out | ctx_1 | .. ] (CaSE/Model.py:116, Masque/Model.py:42; common/TransformerSeqEncoderDecoder.py:141-142 of the reference's generic
 * decoder): y[r, o] = sum_k xs[k][r, :] . w[o, off_k .. off_k + widths[k]) + b[o].  xs / widths: HOST arrays of nseg (1..4) device pointers
 * ([rows, widths[k]] contiguous, `dtype`) and their widths; w f32 [nout, sum widths], b f32 [nout] (nullable), y f32 [rows, nout], nout 1..8.
 * Inference only (no backward).  CASE_FEAT_LINEAR_SKINNY. */
int case_linear_skinny(const void* const* xs, const int64_t* widths, int32_t nseg, const float* w, const float* b, float* y, int64_t rows,
                       int32_t nout, int32_t dtype, case_stream_t stream);
/* masked mean over the sequence: common/Utils.py:455-470.  x [n, L, H], valid u8 [n, L] -> out [n, H] */
int case_masked_mean_fwd(const void* x, const uint8_t* valid, void* out, int64_t n, int64_t L, int64_t H,
                         int32_t dtype, case_stream_t stream);
int case_masked_mean_bwd(const void* d_out, const uint8_t* valid, void* dx, int64_t n, int64_t L, int64_t H,
                         int32_t dtype, case_stream_t stream);
/* Highway gate: y = sigmoid(g) * tanh(nl) + (1 - sigmoid(g)) * lin  (common/Highway.py:29-35).
 * gnl is [rows, 3*cols] = [g | nl | lin] (one fused GEMM output). */
int case_highway_gate_fwd(const void* gnl, void* y, int64_t rows, int64_t cols, int32_t dtype, case_stream_t stream);
int case_highway_gate_bwd(const void* dy, const void* gnl, void* d_gnl, int64_t rows, int64_t cols, int32_t dtype,
                          case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8  Interaction feature assembly: common/Interaction.py:65-74
 *   out[r, :] = valid[r] ? [e, a1, a2, e*a1, e*a2] : 0     (each [rows, H] -> [rows, 5H])
 *   bwd returns de, da1, da2.
 *   max over passages for the query side (num_q == 1): x [B, P, L, W] -> out [B, L, W], argmax int32
 * ------------------------------------------------------------------------------------------- */
/* K8 as kernels (round 6): everything between the encoder outputs and the two 5H-wide feature tensors of common/Interaction.py:32-74 in TWO
 * launches -- scores + both masked softmaxes, then the four products and the concatenations -- instead of 16 (two rank-1 projections, the
 * column scaling, U and U^T, two broadcast adds, two softmaxes, four batched products, two concatenations):
 *   U[i, j] = w1 . Eq[j] + w2 . Ep[i] + (w3 o Ep[i]) . Eq[j];  A = softmax_j U;  Bm = softmax_i U  (masked positions and all-masked rows: 0)
 *   A1 = A Eq,  B1 = Bm^T Ep,  A2 = A B1,  B2 = Bm^T A1
 *   g_q_p[n, i, :] = p_valid ? [Ep, A1, A2, Ep o A1, Ep o A2] : 0      g_p_q[n, j, :] = q_valid ? [Eq, B1, B2, Eq o B1, Eq o B2] : 0
 * eq [n / eq_div, Lq, H] (eq_div = P when one query faces P passages: pair n reads query n / eq_div, common/Interaction.py:26-29), ep [n, Lp, H],
 * q_valid [n / eq_div, Lq], p_valid [n, Lp] bytes, w f32 [3H] = dual_att_linear.weight; outputs a [n, Lp, Lq] and bt [n, Lq, Lp] (the two
 * probability matrices, what a backward pass reads), g_q_p [n, Lp, 5H], g_p_q [n, Lq, 5H] (the max over passages of :73-74 is
 * case_max_over_p_fwd on it).  bf16, H = 512, Lq = 64, Lp a multiple of 32 up to 512 (case_interaction_supported); CASE_E_UNSUPPORTED otherwise
 * (run the single launches).  The row term w2 . Ep[i] rides in the product (the query operand is w3 o Eq[j] + w2, rounded to bf16 once). */
typedef struct {
  int64_t n, Lp, Lq, H, eq_div;
  int32_t dtype;
} CaseInteractionDesc;
int case_interaction_supported(const CaseInteractionDesc* d);
int case_interaction_fwd(const CaseInteractionDesc* d, const void* eq, const void* ep, const uint8_t* q_valid, const uint8_t* p_valid, const float* w,
                         void* a, void* bt, void* g_q_p, void* g_p_q, case_stream_t stream);
int case_concat5_fwd(const void* e, const void* a1, const void* a2, const uint8_t* row_valid, void* out,
                     int64_t rows, int64_t H, int32_t dtype, case_stream_t stream);
int case_concat5_bwd(const void* d_out, const void* e, const void* a1, const void* a2, const uint8_t* row_valid,
                     void* de, void* da1, void* da2, int64_t rows, int64_t H, int32_t dtype, case_stream_t stream);
int case_max_over_p_fwd(const void* x, void* out, int32_t* argmax, int64_t B, int64_t P, int64_t inner, int32_t dtype,
                        case_stream_t stream);
int case_max_over_p_bwd(const void* d_out, const int32_t* argmax, void* dx, int64_t B, int64_t P, int64_t inner,
                        int32_t dtype, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7  additive (Bahdanau) attention scores: common/BilinearAttention.py:24-46
 *   s[b, t, j] = sum_h v[h] * tanh(wq[b, t, h] + uh[b, j, h])      (never materialises [B, T, S, H])
 * wq f32 [B, T, H] (small: the query projection is produced in f32), uh [B, S, H] in dtype; v f32 [H];
 * s f32 [B, T, S].
 * bwd: given ds f32 [B, T, S]:  d_wq[b,t,h] = sum_j g, d_uh[b,j,h] = sum_t g, d_v[h] = sum ds*tanh,
 *      g = ds * v[h] * (1 - tanh^2);  d_wq, d_uh are f32 and fully written; d_v is f32 and must be
 *      pre-zeroed (one coalesced atomic per h per workgroup).
 * ------------------------------------------------------------------------------------------- */
int case_additive_scores_fwd(const float* wq, const void* uh, const float* v, float* s, int64_t B, int64_t T,
                             int64_t S, int64_t H, int32_t dtype, case_stream_t stream);
int case_additive_scores_bwd(const float* ds, const float* wq, const void* uh, const float* v, float* d_wq,
                             float* d_uh, float* d_v, int64_t B, int64_t T, int64_t S, int64_t H, int32_t dtype,
                             case_stream_t stream);

/* K22 (round 5), the greedy step's additive attention in one launch (common/BilinearAttention.py:31-59 at T = 1, CaSE/Model.py:79-82):
 *   s_j = v . tanh(wq + uh_j),  p = softmax_j(s | col_valid) (0 where masked, all 0 for an invalid target row),  ctx = sum_j p_j value_j,
 *   copy_j = p_j prior_j / (1e-8 + sum_j p_j prior_j)   (when prior is given)
 * with tanh(a + b) = 1 - 2 / (e^{2a} e^{2b} + 1) and eu = e^{2 uh} CACHED across the steps (case_additive_key_exp: f32 uh -> bf16 eu, both
 * exponents clamped at +-43): one reciprocal per element instead of an exponential and a reciprocal.  wq [B, H] f32 (query projection incl.
 * bias), eu / value [B, S, H] bf16, v [H] f32, col_valid [B, S] / row_valid [B] bytes (nullable), prior [B, S] f32 (nullable, with copy);
 * outputs ctx [B, H] bf16, p [B, S] f32, copy [B, S] f32.  One workgroup per item; H = 512, S <= 28000; CASE_E_UNSUPPORTED otherwise.
 * wq_add [B, H] f32 (nullable, round 6): added to wq -- the step-invariant part of the query projection when the query is [x_t | feature] with a
 * feature that does not change over the steps (CaSE/Model.py:77-78: the answer representation), so that a step projects x_t alone. */
int case_additive_key_exp(const float* uh, void* eu, int64_t n, case_stream_t stream);
/* K23 (round 5), the greedy step's pointer-generator head in one launch (CaSE/Model.py:34-48, :112-117; Masque/Model.py:29-44; argmax as
 * common/Utils.py:156-168, lowest index on ties):  gen = softmax(logits);  pm = softmax(mix_logits);
 *   dist[b, :] = pm[b, 0] gen[b, :] + sum_k scatter(pm[b, 1 + k] copies[k][b, :] over the source tokens);  ids[b] = argmax dist[b, :].
 * logits / gen / dist [B, V] f32 (gen and dist nullable: a greedy step before the last one needs ids only -- 2 x B x V x 4 bytes of writes
 * per step less), mix_logits [B, 1 + nmem] f32, keys [B, S] the sorted (token << 15 | position) keys of
 * case_source_sort over the concatenated source map, copies = nmem device pointers (host array) to [B, lens[k]] f32 weights, sum lens = S,
 * ids [B] int64, top [B] f32 (nullable) = dist[b, ids[b]].  One workgroup per row, the vocabulary row in LDS: V <= 36000, nmem <= 4,
 * S <= 32768; CASE_E_UNSUPPORTED otherwise (run case_softmax_fwd / case_copy_scatter_sorted_fwd / case_row_argmax). */
int case_pointer_head_decode(const float* logits, const float* mix_logits, const uint32_t* keys, const float* const* copies,
                             const int64_t* lens, int32_t nmem, float* gen, float* dist, int64_t* ids, float* top, int64_t B, int64_t V,
                             int64_t S, case_stream_t stream);
int case_pointer_attend_decode(const float* wq, const float* wq_add, const void* eu, const float* v, const void* value, const uint8_t* col_valid,
                               const uint8_t* row_valid, const float* prior, void* ctx, float* p, float* copy, int64_t B, int64_t S, int64_t H,
                               case_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * K11 copy / pointer distribution: CaSE/Model.py:38-48 with common/Utils.py:344-355
 *   dist[b, t, src[b, s]] += w[b, t, s]        (scatter-add instead of the dense one-hot bmm)
 * src int64 [B, S]; w f32 [B, T, S]; dist f32 [B, T, V] (caller zero-fills or pre-loads dist1).
 * bwd: d_w[b, t, s] = d_dist[b, t, src[b, s]].
 * ------------------------------------------------------------------------------------------- */
int case_copy_scatter_fwd(const int64_t* src, const float* w, float* dist, int64_t B, int64_t T, int64_t S,
                          int64_t V, case_stream_t stream);
int case_copy_scatter_bwd(const int64_t* src, const float* d_dist, float* d_w, int64_t B, int64_t T, int64_t S,
                          int64_t V, case_stream_t stream);
/* Sorted form (on-device source_map sort, once per batch; the reference rebuilds a [B, S, V] one-hot per call,
 * common/Utils.py:344-355).  case_source_sort: keys u32 [B, S] = (src[b, s] << 15 | s) ascending per row, ids outside [0, V)
 * last as 0xFFFFFFFF; needs S <= 32768, V <= 131071.  case_copy_scatter_sorted_fwd adds each run of equal tokens to dist in a
 * fixed order with plain stores: no atomics, bit-reproducible.  Same dist contract as case_copy_scatter_fwd; the backward is
 * case_copy_scatter_bwd (a gather, order-free). */
int case_source_sort(const int64_t* src, uint32_t* keys, int64_t B, int64_t S, int64_t V, case_stream_t stream);
int case_copy_scatter_sorted_fwd(const uint32_t* keys, const float* w, float* dist, int64_t B, int64_t T, int64_t S,
                                 int64_t V, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K12 losses
 *   NLL of log(dist + 1e-8) at the target, ignore_index = 0 (CaSE/Model.py:306, Masque/Model.py:239):
 *     per_row[r] = target[r] ? -log(dist[r, target[r]] + 1e-8) : 0 ; bwd writes only the target column
 *     into a pre-zeroed d_dist: d_dist[r, y] = -g_row[r] / (dist[r, y] + 1e-8)
 *   Row argmax with lowest-index tie break (common/Utils.py:167) for greedy decoding (K13).
 * ------------------------------------------------------------------------------------------- */
int case_nll_gather_fwd(const float* dist, const int64_t* target, float* per_row, int64_t rows, int64_t V,
                        case_stream_t stream);
int case_nll_gather_bwd(const float* dist, const int64_t* target, const float* g_row, float* d_dist, int64_t rows,
                        int64_t V, case_stream_t stream);
int case_row_argmax(const float* x, int64_t* idx, float* val, int64_t rows, int64_t cols, int64_t ld,
                    case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K15  optimizer-side multi-tensor kernels: common/CumulativeTrainer.py:70-76 clip_grad_norm_(params, 1) ->
 *   optim.Adam.step() (CaSE/Run.py:27: default betas / eps, no weight decay) -> EMA.update() (common/EMA.py:13-18), plus the
 *   refresh of the bf16 operand copies, in two launches over all parameter tensors.
 * `table` (device): one CaseOptTensor per tensor, f32 p / g / m (exp_avg) / v (exp_avg_sq) / shadow (EMA, may be null),
 *   p_bf16 (may be null), numel, and the tensor's step-dependent scalars step_size = lr / (1 - beta1^step), bc2_sqrt =
 *   sqrt(1 - beta2^step) (torch.optim.Adam keeps one step count per parameter; formed in double on the host and rounded to f32
 *   once, as torch does).  An entry with g == null is a parameter without a gradient this step: only its EMA shadow moves
 *   (common/EMA.py:13-18 touches every trainable parameter).  `chunks` (device): int32 pairs (tensor index, chunk index), one
 *   per workgroup, chunk = case_optim_chunk_elems() elements.
 * case_optim_sumsq: partials f32 [nchunks] (workspace) receives one sum of squares per chunk; sumsq (f32 scalar) their sum,
 *   added by one workgroup in a fixed order: bit-reproducible and identical on every data-parallel rank, like the reference's
 *   clip_grad_norm_.  Pass sumsq to case_optim_adam_ema to apply the clip coefficient min(1, max_norm / (sqrt(sumsq) + 1e-6))
 *   without a host round trip (null: no clipping).  ema_w = 1 - decay (0: no EMA update).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  void* p;
  const void* g;
  void* m;
  void* v;
  void* shadow;
  void* p_bf16;
  int64_t numel;
  float step_size;
  float bc2_sqrt;
} CaseOptTensor;
int case_optim_chunk_elems(void);
int case_sizeof_opt_tensor(void); /* sizeof(CaseOptTensor) as the library was built: a binder checks its own layout against it */
int case_optim_sumsq(const CaseOptTensor* table, const int32_t* chunks, int64_t nchunks, float* partials, float* sumsq,
                     case_stream_t stream);
/* state (nullable, ABI 600): step_size / bc2_sqrt of EVERY entry with a gradient are read from the device struct instead of the table
 * (all such parameters then share one step count -- a captured step keeps its table static). */
int case_optim_adam_ema(const CaseOptTensor* table, const int32_t* chunks, int64_t nchunks, const float* sumsq, float max_norm,
                        double beta1, double beta2, double eps, double ema_w, const CaseStepState* state, case_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K16  the row-local half of an encoder layer in ONE launch (inference form, bf16, d_model = dim_feedforward = 512):
 *   common/TransformerEncoder.py:66-75 around the attention core, and the next layer's :66-67:
 *     y = x_in Wo^T + bo + resid;  s2 = LN2(y);  o = gelu(s2 W1^T + b1) W2^T + b2 + s2;  s' = LN1_next(o);  qkv' = s' Wqkv'^T + bqkv'
 *   variant 0 (full):  x_in = this layer's attention output [rows, 512], resid = its normed input s; writes s_out = s', qkv_out = qkv'
 *   variant 1 (tail):  the last layer: writes s_out = o (the encoder output); LN1_next / QKV arguments may be null
 *   variant 2 (head):  layer 0: x_in = embedding output; writes s_out = LN1(x_in), qkv_out; the layer-stage arguments may be null
 * Weights are passed PRE-PACKED (case_encoder_chain_pack: bf16 row-major [512, 512] x 3 and [1536, 512] -> MFMA fragment order,
 * case_encoder_chain_packed_bytes() bytes; any matrix may be null for the variants that do not read it); biases and LayerNorm
 * parameters f32.  Of the 20 activation passes over HBM that the same chain makes as single GEMM / LayerNorm launches, 6 remain
 * (x_in, resid in; s', qkv' out).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t rows;   /* tokens (sequences x length) */
  int32_t width;  /* 512 */
  int32_t variant;
  float eps_ln2, eps_ln1_next;
} CaseEncoderChainDesc;
int64_t case_encoder_chain_packed_bytes(void);
int case_encoder_chain_pack(const void* wo, const void* w1, const void* w2, const void* wqkv, void* packed, case_stream_t stream);
int case_encoder_chain(const CaseEncoderChainDesc* d, const void* x_in, const void* resid, const void* packed, const float* bo,
                       const float* b1, const float* b2, const float* bqkv, const float* ln2_g, const float* ln2_b,
                       const float* ln1n_g, const float* ln1n_b, void* s_out, void* qkv_out, case_stream_t stream);

/* (K20, case_decoder_chain -- the fused decoder step chain of round 4 -- was retired in round 5: measured slower than the single launches
 * it replaced, 3.93 against 3.43 ms per cached step; the greedy step's time went to K21 - K23 instead.  Its feature bit stays reserved.) */

/* Greedy post-processing on the device (common/Utils.py:200-217 to_sentence): per row of ids [B, T] drop the BOS / PAD ids and
 * everything from the first EOS on; out [B, T] holds the kept ids front-packed (pad behind), len [B] their count.  One host
 * copy of (out, len) replaces the reference's `.item()` per generated token.  Pass -1 for a special id the vocabulary lacks. */
int case_sentence_compact(const int64_t* ids, int64_t* out, int32_t* len, int64_t B, int64_t T, int64_t bos, int64_t pad,
                          int64_t eos, case_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CASE_HIP_H */
