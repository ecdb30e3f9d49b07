/*
 * vyom_hip.h -- C ABI of libvyom_hip.so, the MI355X (gfx950 / CDNA4) kernel library
 * behind vyomai_amd's mirror of the VyomAI layer API.
 *
 * The reference (Ajax0564/VyomAI) has no FFI: its hot path is the Python class API of
 * VyomAI/layers consumed by VyomAI/models.  Each entry point
 * below replaces the torch ops that one reference call site issues; the citation after
 * "replaces:" is path:line in the reference checkout.
 *
 * Conventions (every function):
 *   - plain device pointers + explicit sizes/strides (in ELEMENTS unless noted), no torch types;
 *   - `dtype`: VY_F32 or VY_BF16 -- the storage type of activations/weights; accumulation,
 *     softmax and norm statistics are always fp32;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - no device allocation inside; callers pass workspaces;
 *   - returns 0 on success, a negative vy_status otherwise; vy_last_error() describes it.
 */
#ifndef VYOM_HIP_H
#define VYOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { VY_F32 = 0, VY_BF16 = 1 } vy_dtype;

typedef enum {
  VY_OK = 0,
  VY_ERR_ARG = -1,       /* bad shape / alignment / null pointer */
  VY_ERR_UNSUPPORTED = -2,
  VY_ERR_LAUNCH = -3     /* hipGetLastError() after the launch */
} vy_status;

typedef enum { VY_ACT_NONE = 0, VY_ACT_GELU_ERF = 1, VY_ACT_GELU_TANH = 2 } vy_act;
/* OR-ed into `act` of vy_linear_fwd / vy_linear_dgrad (training): the tensor saved for backward is act'(x W^T + b)
 * instead of the pre-activation -- the forward epilogue has Phi and the Gaussian of the erf GELU at hand anyway, and
 * the dgrad epilogue (dX = (dY W) * act') becomes one multiply per element instead of an erf + exp evaluation.
 * vy_linear_fwd: pre_out receives act' (rounded to the storage type); vy_linear_dgrad: `pre` holds act'. */
#define VY_ACT_SAVE_DERIV 0x100

/* attention mask descriptor bits (vy_attn_fwd / vy_attn_bwd) */
enum {
  VY_MASK_NONE = 0,
  VY_MASK_CAUSAL = 1,   /* key j visible to query i iff j <= i + start_pos              */
  VY_MASK_KEYPAD = 2,   /* uint8 keep[B][S]; 0 = masked with finfo.min like the reference */
  VY_MASK_ADDITIVE = 4  /* generic fp32 additive mask (B,1,Lm,S), Lm in {1, L}            */
};

const char* vy_last_error(void);
int vy_abi_version(void);

/* Launch-sizing hint: the caller is about to run `n` independent launch chains side by side on `n` streams (the
 * training forward as two batch halves, vyomai_amd/ops.py `lanes`), so each launch should size its grid for 1/n of
 * the 256 CUs: a 128-tile launch of 256 x 192 tiles is then a full share, not a half-empty chip.  n = 1 (the default)
 * restores whole-chip sizing.  Process-global, read on the host when a launch is sized; results never depend on it. */
int vy_set_concurrent_chains(int n);

/* Scratch for the launches on `stream` (the library never allocates device memory): mid-size GEMMs (32 < M <= ~2304 rows,
 * fewer than 300 tiles of 128 x 128) split K over workgroups and keep fp32 partial tiles here between their two launches
 * -- 64 KiB per (tile, slice), the widest launch (16 slices of 16 tiles of 320 x 128) needs 42 MiB; the host layer registers 64 MiB.  One workspace per stream (launches on one stream are
 * ordered, so they can share it); ws = NULL or bytes = 0 removes the entry.  Without a workspace such GEMMs run unsplit:
 * results are the same up to the fp32 summation order over K. */
int vy_workspace_set(void* stream, void* ws, int64_t bytes);

/* ------------------------------------------------------------------------------------------
 * vy_linear_fwd:  Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N]) + residual[M,N]
 * replaces: nn.Linear call sites -- AttentionSelfOutput.dense + residual add
 *   (VyomAI/layers/attention.py:69-71), FeedForward.intermediate + GELU and FeedForward.out +
 *   residual (VyomAI/layers/ffn.py:35-39), LMHead.dense/decoder (VyomAI/models/decoder.py:267-275).
 * `pre_out` (nullable, [M,ldy]) receives X.W^T+bias before the activation (saved for backward).
 * bias/residual/pre_out may be NULL.  K % 8 == 0 (bf16) / K % 4 == 0 (f32); rows 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
int vy_linear_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                  const void* residual, int64_t ldr, void* y, int64_t ldy, void* pre_out,
                  int64_t M, int64_t N, int64_t K, int act, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * vy_qkv_rope_fwd: fused Q/K/V projection + bias + rotary embedding + head split.
 * replaces: self.query/key/value (or self.qkv + chunk), rearrange 'b l (h d) -> b h l d',
 *   apply_rotary_pos_emb and the KV-cache slice write
 *   (VyomAI/layers/attention.py:114-126, 607-617; VyomAI/models/decoder.py:91-105;
 *    VyomAI/layers/positional_embeddings.py:155-182; VyomAI/layers/kv_cache.py:355-356).
 * w: packed [(h + 2*hk)*dh, K] rows = [Wq; Wk; Wv]; bias likewise or NULL.
 * cos/sin: fp32 tables [>= pos0+L][dh/2] (angles' cos/sin, half width) or NULL for no RoPE;
 *          row (pos0 + l) is used for token l.
 * q/k/v outputs are (B, heads, *, dh) with element strides {batch, head, token}; writing K/V
 * straight into a static cache is done by pointing k/v at cache[:, :, start_pos] with the
 * cache's strides.
 * ------------------------------------------------------------------------------------------ */
int vy_qkv_rope_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                    const float* cos_tab, const float* sin_tab, int64_t pos0,
                    void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl,
                    void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                    void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                    int64_t B, int64_t L, int64_t K, int h, int hk, int dh, int dtype,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * vy_attn_fwd: softmax(Q K^T / sqrt(dh) + mask) V, flash style, heads merged on output.
 * replaces: repeat_kv + F.scaled_dot_product_attention + rearrange 'b h l d -> b l (h d)'
 *   (VyomAI/layers/attention.py:8-19, 128-132, 205-213, 283-287, 368-377, 619-623;
 *    VyomAI/models/decoder.py:107-111, 190-199).
 * q (B,h,L,dh), k/v (B,hk,S,dh) with element strides {batch, head, token}; kv head of query
 * head i is i / (h/hk).  out is (B, L, h*dh) with strides {o_sb, o_sl}.
 * mask_kind: OR of VY_MASK_*; keypad uint8 [B][S] (stride kp_sb); additive fp32 with strides
 * {am_sb, am_sl} (am_sl = 0 broadcasts one row).  Masked scores take finfo(fp32).min exactly as
 * the reference's (1-mask)*finfo.min does, so a fully masked row averages V.
 * lse (nullable) fp32 [B,h,L]: natural-log sum-exp of the scaled, masked scores (for backward).
 * ------------------------------------------------------------------------------------------ */
int vy_attn_fwd(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl,
                const void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                void* out, int64_t o_sb, int64_t o_sl, float* lse,
                int mask_kind, int64_t start_pos, const uint8_t* keypad, int64_t kp_sb,
                const float* addmask, int64_t am_sb, int64_t am_sl,
                int64_t B, int h, int hk, int64_t L, int64_t S, int dh, float scale, int dtype,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * vy_attn_decode: one query token per sequence against a KV cache (L == 1, no mask).
 * replaces: the L==1 branch of DecoderAttention*.forward (mask=None, VyomAI/models/decoder.py:
 *   355-356, 105-111) reading StaticCacheOne/DynamicCacheOne prefixes (VyomAI/layers/kv_cache.py:
 *   229-236, 358-361).  q (B,h,1,dh) strides {q_sb,q_sh}; k/v cache strides {batch, head, token};
 *   S = start_pos + 1 keys are attended.  out (B,1,h*dh) row stride o_sb.
 * ------------------------------------------------------------------------------------------ */
int vy_attn_decode(const void* q, int64_t q_sb, int64_t q_sh,
                   const void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                   const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                   void* out, int64_t o_sb, int64_t B, int h, int hk, int64_t S, int dh,
                   float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * vy_layernorm_fwd: y = (x - mean) * rstd * gamma + beta over the last dim (biased variance).
 * replaces: nn.LayerNorm in AttentionSelfOutput / FeedForward / LMHead
 *   (VyomAI/layers/attention.py:71, VyomAI/layers/ffn.py:39, VyomAI/models/decoder.py:270).
 * mean/rstd (nullable) fp32 [M] are saved for backward.
 * ------------------------------------------------------------------------------------------ */
int vy_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y,
                     int64_t ldy, float* mean, float* rstd, int64_t M, int64_t N, float eps,
                     int dtype, void* stream);

/* vy_rmsnorm_fwd: y = x * rsqrt(mean(x^2) + eps) * (w_offset + w); Gemma uses w_offset = 1
 * (Examples/paligemma.ipynb cell 11, GemmaRMSNorm).  Statistics in fp32. */
int vy_rmsnorm_fwd(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t M,
                   int64_t N, float eps, float w_offset, int dtype, void* stream);

/* vy_gated_act_fwd: out[m, i] = act(gate_up[m, i]) * gate_up[m, I + i] -- the GeGLU of GemmaMLP
 * (Examples/paligemma.ipynb cell 11: gelu_tanh(gate_proj(x)) * up_proj(x)) after one packed
 * [gate; up] projection. */
int vy_gated_act_fwd(const void* gate_up, int64_t ldg, void* out, int64_t ldo, int64_t M, int64_t I,
                     int act, int dtype, void* stream);

/* vy_rope_fwd: in-place rotary embedding on a (B,heads,L,dh) tensor (strides as above);
 * used when the fused epilogue does not apply.  `inverse` != 0 applies the transpose rotation
 * (the backward of RoPE).  replaces: VyomAI/layers/positional_embeddings.py:155-182. */
int vy_rope_fwd(void* x, int64_t sb, int64_t sh, int64_t sl, const float* cos_tab,
                const float* sin_tab, int64_t pos0, int64_t B, int heads, int64_t L, int dh,
                int inverse, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward / training entry points (bf16 storage, fp32 accumulation).
 * They mirror what autograd derives for the reference modules; the fusion groups follow the
 * author's own fused notebook (Examples/vyom-ai-decoder-fused.ipynb cells 2-7, 11).
 * ------------------------------------------------------------------------------------------ */

/* dX[M,K] = dY[M,N] . W[N,K]  (+ add_to[M,K] + add_to2[M,K]); optional GELU backward: when `pre` is
 * given the result is multiplied by gelu'(pre[M,K]) (pre = saved pre-activation of the consumer of
 * dX).  wt is W transposed, i.e. stored [K,N] row-major (the trainer keeps both copies).  The two
 * addends are the residual-path gradients a layer input collects besides the QKV projection's
 * (add_to2 needs add_to). */
int vy_linear_dgrad(const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const void* pre,
                    int64_t ldpre, int act, const void* add_to, int64_t ldadd, const void* add_to2,
                    int64_t ldadd2, void* dx, int64_t lddx, int64_t M, int64_t N, int64_t K, int dtype,
                    void* stream);

/* dW[N,K] (fp32, accumulate when beta != 0) = alpha * dY[M,N]^T . X[M,K];  db[N] (fp32) likewise
 * alpha * colsum(dY).  alpha_dev: optional DEVICE fp32 scalar (NULL = 1), e.g. the upstream
 * gradient of a loss whose unit gradient is already stored in dY -- no host sync to read it.
 * dtype VY_BF16: MFMA kernels (K, lddy, ldx multiples of 8).  VY_F32: the parity path, plain FMAs
 * (any K; the reference's autograd of nn.Linear in full precision). */
int vy_linear_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw,
                    int64_t lddw, float* db, float beta, const float* alpha_dev, int64_t M, int64_t N,
                    int64_t K, int dtype, void* stream);

/* Several weight gradients in one launch: dW_i += dY_i^T X_i and db_i += column sums of dY_i (db_i may be NULL)
 * for 1..8 GEMMs, accumulating (the beta = 1 form of vy_linear_wgrad).  The four projections of a
 * transformer layer (reference: the autograd of the nn.Linear layers at layers/attention.py:87-95, :57-72,
 * layers/ffn.py:19-40) are small as wgrad problems -- 9 to 36 output tiles of 256 x 256 -- and filling the
 * chip one at a time costs M-splits, i.e. fp32 atomic traffic; together they need a quarter of it. */
typedef struct vy_wgrad_desc {
  const void* dy; int64_t lddy;
  const void* x; int64_t ldx;
  float* dw; int64_t lddw;
  float* db;
  int64_t M, N, K;
} vy_wgrad_desc;
int vy_linear_wgrad_grouped(const vy_wgrad_desc* descs, int32_t n, int dtype, void* stream);

/* LayerNorm backward: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma;
 * dgamma/dbeta fp32 [N] accumulated (beta) from per-block partials in `ws`
 * (ws: fp32, at least 2 * ws_rows * N elements; ws_rows = vy_layernorm_bwd_ws_rows(M)). */
int64_t vy_layernorm_bwd_ws_rows(int64_t M);
int vy_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma,
                     const float* mean, const float* rstd, void* dx, int64_t lddx, float* dgamma,
                     float* dbeta, float beta, float* ws, int64_t M, int64_t N, int dtype,
                     void* stream);

/* Flash attention backward.  dO/O are (B,L,h*dh) (strides {sb,sl}); dq/dk/dv use q/k/v layouts.
 * delta_ws: fp32 [B,h,L] scratch.  dk/dv are (B,hk,S,dh): the n_rep query heads of one kv head
 * are summed in-kernel.  Same mask descriptor as the forward.  cos_tab/sin_tab (nullable): when
 * q/k were rotated by RoPE (row rope_pos0 + token of the tables), the inverse rotation is applied
 * to dq and dk (in the epilogues at dh = 64 bf16, by rotation launches otherwise), so they are
 * gradients w.r.t. the un-rotated projections.
 * bf16: dh = 64 (tuned kernels) or any multiple of 8 up to 256 (general MFMA kernels; SigLIP's 72 runs
 * as 96 columns).  fp32: the parity path (plain FMAs, expf), dh a multiple of 8 up to 256 -- there a
 * row without a visible key is differentiated as the uniform softmax the forward returned for it
 * (reference layers/attention.py:133-137 under autograd); the bf16 kernels give such rows no gradient. */
int vy_attn_bwd(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl,
                const void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                const void* out, const void* dout, int64_t o_sb, int64_t o_sl, const float* lse,
                float* delta_ws,
                void* dq, int64_t dq_sb, int64_t dq_sh, int64_t dq_sl,
                void* dk, int64_t dk_sb, int64_t dk_sh, int64_t dk_sl,
                void* dv, int64_t dv_sb, int64_t dv_sh, int64_t dv_sl,
                int mask_kind, int64_t start_pos, const uint8_t* keypad, int64_t kp_sb,
                const float* cos_tab, const float* sin_tab, int64_t rope_pos0,
                int64_t B, int h, int hk, int64_t L, int64_t S, int dh, float scale, int dtype,
                void* stream);

/* ---- data-parallel gradient exchange (RCCL) --------------------------------------------------------------------
 * Replaces what the reference gets from accelerate / torch DDP (Examples/vyom-ai-decoder_clm.ipynb cell 31,
 * Examples/vyom-ai-accelerate-multimodel-2t4.ipynb cells 1-2): one process per GPU, one communicator per process.
 *   vy_ddp_unique_id   rank 0 fills 128 bytes; the host hands them to every rank (any side channel: torch.distributed's
 *                      store, MPI, a file)
 *   vy_ddp_init        collective; binds the communicator to the calling thread's current HIP device
 *   vy_ddp_all_reduce_async  in-place SUM of `count` elements (VY_F32 | VY_BF16) enqueued on `stream`: the caller orders
 *                      it behind the kernels that wrote the bucket and ahead of its consumers with events / stream waits
 *                      (there is no separate wait call: completion is stream order)
 * RCCL is bound at run time (the copy already in the process, e.g. torch's, else librccl.so): where it is absent these
 * return VY_ERR_UNSUPPORTED.  The number of CUs RCCL's channel kernels occupy beside the backward GEMMs is RCCL's own
 * knob (NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS in the environment of the process), read at vy_ddp_init. */
int vy_ddp_unique_id(void* id128);
int vy_ddp_init(const void* id128, int rank, int world);
int vy_ddp_world(void);   /* 0 before vy_ddp_init */
int vy_ddp_rank(void);    /* -1 before vy_ddp_init */
int vy_ddp_all_reduce_async(void* buf, int64_t count, int dtype, void* stream);
int vy_ddp_destroy(void);

/* Fused AdamW over a flat fp32 parameter arena; also refreshes the bf16 working copy.
 * p, m, v fp32 [n]; g fp32 [n]; p_bf16 (nullable) bf16 [n].  Matches torch.optim.AdamW. */
int vy_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int64_t step,
                  float grad_scale, const float* grad_scale_dev, void* stream);
/* The same with a gate (nullable): one fp32 on the device; when it reads 0 the launch changes nothing.  Data-parallel
 * training with parameters that some ranks' batches do not reach (the reference's multimodal model under
 * find_unused_parameters=True, Examples/vyom-ai-accelerate-multimodel-2t4.ipynb cell 1): whether such a parameter is
 * updated is decided by a flag all-reduced over the ranks, read here on the device. */
int vy_adamw_step_gated(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int64_t step,
                        float grad_scale, const float* grad_scale_dev, const float* gate, void* stream);
/* grad_scale_dev (nullable): one fp32 on the device multiplied into grad_scale by the kernel -- the
 * clip_grad_norm_ coefficient min(1, max_norm / (norm + 1e-6)) of the reference's training loops
 * (Examples/vyomai-fused-kernals-2t4.ipynb cell 0) without a host round trip. */

/* out[0] = sum_i x[i]^2 over an fp32 arena (global gradient norm, same loops).  ws: >= 1024 floats of
 * scratch.  Two launches, fixed summation order (run-to-run identical). */
int vy_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dropout of the two hidden-state sites of a block.
 * replaces: self.dropout(hidden_states) between dense and the residual add in AttentionSelfOutput
 *   (VyomAI/layers/attention.py:55,69-71) and FeedForward (VyomAI/layers/ffn.py:24,37-39), p =
 *   config.hidden_dropout_prob (VyomAI/utils.py:96, default 0.1), active in train().
 * The keep mask is a pure function of (seed, offset, row, column) -- Philox4x32-7 on the counter
 * {column / 8, row, offset}, sixteen bits per element, drop when below round(p * 65536) -- so the forward
 * epilogue, the backward pass and a test all regenerate it; nothing is stored.
 * vy_linear_dropout_fwd: Y = dropout(X . W^T + bias) / (1 - p) + residual   (vy_linear_fwd with the
 *   mask applied in the epilogue, before the residual add).
 * vy_dropout: y = x * keep / (1 - p) as its own pass over an [M, N] view (backward: the same mask on the
 *   incoming gradient; x = ones gives the mask).
 * ------------------------------------------------------------------------------------------ */
int vy_linear_dropout_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                          const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int64_t N,
                          int64_t K, float p_drop, uint64_t seed, uint64_t offset, int dtype, void* stream);
int vy_dropout(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t N, float p_drop,
               uint64_t seed, uint64_t offset, int dtype, void* stream);

/* dx = dy * act'(pre), elementwise over [M,N] row-major views (GELU backward outside a GEMM). */
int vy_act_bwd(const void* dy, int64_t lddy, const void* pre, int64_t ldpre, void* dx, int64_t lddx,
               int64_t M, int64_t N, int act, int dtype, void* stream);

/* Softmax cross-entropy over the vocabulary, ignore_index aware (the CLM loss of
 * Examples/vyom-ai-decoder_clm.ipynb cell 29 after the label shift).
 * fwd: lse[m] = logsumexp(logits[m,:V]); loss_sum += sum_m (lse[m] - logits[m,label]) and
 *      count += #(label != ignore) -- both fp32 scalars on the device, accumulated atomically
 *      (zero them first).
 * bwd: logits[m,:] <- (softmax(logits[m,:]) - onehot(label)) * (*gscale) / (*count), in place
 *      (rows with label == ignore become 0); gscale/count are device pointers: no host sync. */
int vy_xent_fwd(const void* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, float* lse,
                float* loss_sum, float* count, int64_t M, int64_t V, int32_t* err_flag, int dtype, void* stream);
int vy_xent_bwd(void* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, const float* lse,
                const float* gscale, const float* count, int64_t M, int64_t V, int dtype, void* stream);
/* Both in ONE pass over the logits (bf16, V <= 65536): `count` (device scalar: the number of rows
 * with label != ignore) is an INPUT here; lse and loss_sum as in vy_xent_fwd, then the row is
 * overwritten in place as in vy_xent_bwd.  The logits cross HBM once in each direction. */
int vy_xent_fused(void* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, float* lse,
                  float* loss_sum, const float* count, const float* gscale, int64_t M, int64_t V,
                  int32_t* err_flag, int dtype, void* stream);
/* A label that is neither ignore_index nor inside [0, V) (torch.cross_entropy device-asserts on it) is
 * never dereferenced: the row counts as ignored (zero loss, zero gradient) and *err_flag (nullable,
 * device int32) is set to 1 for the host to report. */

/* Token embedding (nn.Embedding: VyomAI/models/decoder.py:287, encoder.py:41, multimodel.py):
 * fwd: out[m,:] = table[ids[m],:] (bf16 or fp32).  Ids outside [0,V) give a zero row and set the
 *      optional device flag *err_flag to 1 (the host cannot check ids without a sync).
 * bwd: dW[ids[m],:] += dOut[m,:] in fp32 (atomics), except for id == padding_idx (pass -1 for none). */
int vy_embedding_fwd(const void* table, int64_t ldt, const int64_t* ids, void* out, int64_t ldo,
                     int64_t M, int64_t d, int64_t V, int32_t* err_flag, int dtype, void* stream);
int vy_embedding_bwd(const void* dout, int64_t lddo, const int64_t* ids, float* dw, int64_t lddw,
                     int64_t padding_idx, int64_t M, int64_t d, int64_t V, int dtype, void* stream);

/* out[c, r] = in[r, c] for a [R,C] matrix (bf16 or f32): keeps W^T copies for dgrad. */
int vy_transpose(const void* in, int64_t ldin, void* out, int64_t ldout, int64_t R, int64_t C,
                 int dtype, void* stream);

/* Many transposes in one launch: descs_dev is a DEVICE array of n descriptors; tile0 is the running
 * sum of ceil(R/64)*ceil(C/64) over the preceding descriptors, tiles_c = ceil(C/64), total_tiles the
 * sum over all.  Used once per training step for every W^T the dgrad GEMMs read. */
typedef struct vy_transpose_desc {
  const void* in; void* out;      /* in: [R,C] row-major (ldin); out: [C,R] (ldout) */
  int64_t ldin, ldout;
  int32_t R, C, tile0, tiles_c;
} vy_transpose_desc;              /* 48 bytes */
int vy_transpose_batched(const vy_transpose_desc* descs_dev, int32_t n, int32_t total_tiles, int dtype,
                         void* stream);

/* Sampling front end of the generate loops.
 * vy_greedy_step replaces reference models/decoder.py:478-507 (topk(k=1), the where() that forces prompt
 *   tokens, the token write, the isin()/or of the EOS bookkeeping and the all() reduction) for one
 *   generated position: for row b, next = argmax_v logits[b,v] (lowest index among equal maxima) unless
 *   text_mask[b,cur_pos] != 0 (the row is still inside its prompt), then next = tokens[b,cur_pos];
 *   tokens[b,cur_pos] = next; eos_reached[b] |= !forced && next in eos_ids[0..n_eos); *not_done (may be
 *   NULL; zeroed by the caller) += number of rows with eos_reached[b] == 0 afterwards.
 * vy_sampling_probs replaces reference logits_processors.py LogitsProcessor.__call__ for the five
 *   processors: probs = softmax(mask(logits) / temperature) in fp32.  top_k > 0 keeps the values >= the
 *   k-th largest (:59-63); 0 < top_p < 1 keeps the descending-sorted prefix up to and including the first
 *   element whose cumulative softmax of the unscaled, top-k-masked logits exceeds top_p (:73-81, :92-102);
 *   masked entries get probability exactly 0 (the reference writes -1e20 into them).  Equal logits at
 *   the nucleus cut are all kept (the reference keeps an implementation-defined subset of a tie). */
int vy_greedy_step(const void* logits, int64_t ldl, int64_t B, int64_t V, int dtype, int64_t* tokens,
                   int64_t ldt, int64_t cur_pos, const uint8_t* text_mask, int64_t ldm,
                   const int64_t* eos_ids, int32_t n_eos, uint8_t* eos_reached, int32_t* not_done, void* stream);
int vy_sampling_probs(const void* logits, int64_t ldl, int64_t B, int64_t V, int dtype, float temperature,
                      int32_t top_k, float top_p, float* probs, int64_t ldp, void* stream);

/* y = x converted between fp32 and bf16 (n elements). src_dtype -> dst_dtype. */
int vy_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Native decode driver: one call runs a whole single-token step (L == 1) of the GPT-style
 * decoder -- every layer's QKV+RoPE (K/V appended to the static cache in place), cached
 * attention, out-projection + residual + LayerNorm, FFN + residual + LayerNorm, then the LM head
 * -- as ~8 launches per layer with no host work in between.
 * replaces: the per-token body of DecoderModel.generate / DecoderModel.forward for seqlen == 1
 *   (VyomAI/models/decoder.py:343-374, 477-488) and DecoderLayer.forward (:222-250).
 * All pointers are device pointers; weights are `dtype` (bf16 or f32) in nn.Linear layout.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void *wqkv, *bqkv;          /* packed [(h+2hk)*dh, d], [(h+2hk)*dh] (bias may be NULL) */
  const void *wo, *bo, *ln1_w, *ln1_b;
  const void *w1, *b1, *w2, *b2, *ln2_w, *ln2_b;
  void *kcache, *vcache;            /* (B, hk, cap, dh) */
  int64_t c_sb, c_sh, c_sl;         /* cache element strides {batch, head, token} */
} vy_decode_layer;

typedef struct {
  int32_t num_layers, B, d, h, hk, dh, ffn, vocab, act, dtype;
  float eps_attn, eps_ffn, eps_head;
  const float *cos_tab, *sin_tab;   /* NULL: no RoPE (additive position encodings) */
  const vy_decode_layer* layers;    /* host array [num_layers] */
  const void *head_wd, *head_bd, *head_ln_w, *head_ln_b, *head_wv, *head_bias;
  void* ws; int64_t ws_bytes;       /* device workspace, >= vy_decode_ws_bytes(...) */
} vy_decode_plan;

int64_t vy_decode_ws_bytes(int32_t B, int32_t d, int32_t h, int32_t hk, int32_t dh, int32_t ffn,
                           int32_t dtype);
/* x: (B, d) input embeddings (+ position info); pos: index of this token (keys 0..pos are
 * attended); hidden_out (nullable): (B, d) last hidden state; logits (nullable): (B, ldv).
 * pos_dev (nullable): device int32 holding the position -- when given, every kernel reads the
 * position on the device and `pos` is ignored, so one captured hipGraph of this call can be
 * replayed for every token (bf16, head_dim 64 models). */
int vy_decoder_step(const vy_decode_plan* plan, const void* x, int64_t pos, const int32_t* pos_dev,
                    void* hidden_out, void* logits, int64_t ldv, void* stream);

/* ------------------------------------------------------------------------------------------
 * vy_gemma_decoder_step: one single-token step of a Gemma-style decoder stack (BASELINE configs[4], the
 * PaliGemma-shape language model; Examples/paligemma.ipynb cells 11-13): per layer
 *   n = RMSNorm(x) ; q,k,v = n Wqkv^T (+b) with RoPE, k/v written into the cache at `pos` ;
 *   x = x + attn(q, K[0..pos], V[0..pos]) Wo^T ; n = RMSNorm(x) ; x = x + (gelu_tanh(n Wg^T) * n Wu^T) Wd^T
 * then the final RMSNorm and the (tied) vocabulary projection.  Same launches as the Python layer
 * (models/paligemma.py GemmaDecoderLayer), one C call per generated token instead of ~150.
 * -------------------------------------------------------------------------------------------- */
typedef struct vy_gemma_layer {
  const void* wqkv; const void* bqkv;      /* packed [q;k;v] projection, bias may be NULL */
  const void* wo; const void* bo;
  const void* ln_in; const void* ln_post;  /* RMSNorm weights (applied as 1 + w) */
  const void* wgu;                         /* packed [gate; up], [2*ffn, d] */
  const void* wdown;                       /* [d, ffn] */
  void* kcache; void* vcache;              /* (B, hk, cap, dh) */
  int64_t c_sb, c_sh, c_sl;
} vy_gemma_layer;
typedef struct vy_gemma_plan {
  int32_t num_layers, B, d, h, hk, dh, ffn, vocab, dtype;
  float eps;
  const float* cos_tab; const float* sin_tab;
  const vy_gemma_layer* layers;
  const void* norm_w; const void* head_w;  /* final RMSNorm, [vocab, d] projection */
  void* ws; int64_t ws_bytes;              /* >= vy_gemma_ws_bytes(...) */
  int32_t flags;                           /* VY_GEMMA_PRESCALED: every layer's wqkv / wgu is already multiplied by
                                              (1 + ln_in) / (1 + ln_post) along K (bf16, B <= 4): the step skips the
                                              RMSNorm launches and scales each product by rsqrt(mean x^2 + eps) */
} vy_gemma_plan;
#define VY_GEMMA_PRESCALED 1
int64_t vy_gemma_ws_bytes(int32_t B, int32_t d, int32_t h, int32_t dh, int32_t ffn, int32_t dtype);
/* x: (B, d) embeddings of the current token (already scaled by sqrt(d)); keys 0..pos are attended;
 * logits: (B, ldv). */
int vy_gemma_decoder_step(const vy_gemma_plan* plan, const void* x, int64_t pos, void* logits, int64_t ldv,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VYOM_HIP_H */
