// Native decode driver (host code): chains the kernel launchers for one single-token step so the
// Python side makes ONE call per generated token.  See include/vyom_hip.h (vy_decoder_step).
#include "vy_common.h"
#include <stdlib.h>

int vy_qkv_rope_fwd_ex(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                       const float* cos_tab, const float* sin_tab, int64_t pos0, const int* pos_dev, void* q,
                       int64_t q_sb, int64_t q_sh, int64_t q_sl, void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                       void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t B, int64_t L, int64_t K, int h,
                       int hk, int dh, int dtype, void* stream);
int vy_attn_decode_ex(const void* q, int64_t q_sb, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_sh,
                      int64_t k_sl, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl, void* out,
                      int64_t o_sb, int64_t B, int h, int hk, int64_t S, const int* pos_dev, int dh, float scale,
                      int dtype, void* stream);

int vy_gemv_norm(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, const void* norm_w, float eps,
                 const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int64_t N, int64_t K, void* stream);
int vy_gemv_qkv_norm(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, const void* norm_w, float eps,
                     void* q, int64_t q_sb, int64_t q_sh, void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl, void* v,
                     int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t B, int64_t K, int h, int hk, int dh,
                     const float* cos_tab, const float* sin_tab, int64_t pos, void* stream);
#define VY_NORM_PRESCALED ((const void*)(intptr_t)-1)
int vy_gemv_gated(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* norm_w, float eps, void* y, int64_t ldy,
                  int64_t M, int64_t I, int64_t K, int act, void* stream);
int vy_rope_qk(void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl, int hq, void* k, int64_t k_sb, int64_t k_sh,
               int64_t k_sl, int hk, const float* cos_tab, const float* sin_tab, int64_t pos0, int64_t B, int64_t L, int dh,
               int dtype, hipStream_t st);
int64_t vy_splitk_ws_floats(int64_t N);
int vy_linear_res_ln_skinny(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                            const void* residual, int64_t ldr, const void* gamma, const void* beta, float eps,
                            void* y, int64_t ldy, float* part, int64_t M, int64_t N, int64_t K, void* stream);

// lean decode kernels (vy_decode.hip)
bool vy_dec_supported(int B, int d, int h, int hk, int dh, int ffn, int dtype);
int vy_dec_qkv(const void* x, const void* w, const void* bias, const float* cos_tab, const float* sin_tab, int64_t pos0,
               const int* pos_dev, void* q, void* k, void* v, int64_t c_sb, int64_t c_sh, int64_t c_sl, int B, int d, int h,
               int hk, hipStream_t st);
int vy_dec_linear_ex(const void* x, int ldx, const void* w, const void* bias, const void* residual, int ldr, const void* ln_g,
                     const void* ln_b, float ln_eps, void* y, int ldy, int B, int N, int K, int act, hipStream_t st);
int vy_dec_linear(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy, int B, int N, int K, int act,
                  hipStream_t st);
int vy_dec_linear_res_ln(const void* x, int ldx, const void* w, const void* bias, const void* residual, int ldr,
                         const void* gamma, const void* beta, float eps, void* y, int ldy, float* part, int B, int N, int K,
                         int act, hipStream_t st);

static int g_decode_lean = -1;
static int g_gemma_lean = -1;
extern "C" int vy_debug_set_gemma_lean(int v) { g_gemma_lean = v; return 0; }   // test aid: 0 = general kernels for B = 1 too
// measurement / test aid, not part of include/vyom_hip.h: 0 = general launchers, 1 = the decode-only kernels
extern "C" int vy_debug_set_decode_lean(int v) { g_decode_lean = v; return 0; }

int vy_dec_gemv1(const void* x, const void* w, const void* bias, const void* residual, void* y, int N, int K, int prescaled,
                 float eps, hipStream_t st);
int vy_dec_gemv1_gated(const void* x, const void* wgu, void* y, int I, int K, int prescaled, float eps, hipStream_t st);
int vy_dec_gemv1_qkv(const void* x, const void* w, const void* bias, void* q, void* k, void* v, int64_t c_sh, int h, int hk,
                     int dh, const float* cos_tab, const float* sin_tab, int64_t pos, int K, int prescaled, float eps,
                     hipStream_t st);

namespace {
inline int64_t esize(int dtype) { return dtype == VY_BF16 ? 2 : 4; }
inline int64_t up(int64_t x) { return (x + 255) / 256 * 256; }
}  // namespace

extern "C" int64_t vy_decode_ws_bytes(int32_t B, int32_t d, int32_t h, int32_t hk, int32_t dh, int32_t ffn,
                                      int32_t dtype) {
  const int64_t e = esize(dtype);
  // q, attn out, s (pre-LN), a (post-LN1), mid (ffn), two ping-pong hidden buffers, split-K partials
  return up(B * (int64_t)h * dh * e) + 5 * up(B * (int64_t)d * e) + up(B * (int64_t)ffn * e) +
         up((vy_splitk_ws_floats(d) > 24 * 32 * (int64_t)d ? vy_splitk_ws_floats(d) : 24 * 32 * (int64_t)d) * 4) + 4096;
}

extern "C" int vy_decoder_step(const vy_decode_plan* p, const void* x, int64_t pos, const int32_t* pos_dev,
                               void* hidden_out, void* logits, int64_t ldv, void* stream) {
  const char* who = "vy_decoder_step";
  if (!p || !x || !p->layers || !p->ws) VY_FAIL(VY_ERR_ARG, "%s: null plan/input/workspace", who);
  if (p->ws_bytes < vy_decode_ws_bytes(p->B, p->d, p->h, p->hk, p->dh, p->ffn, p->dtype))
    VY_FAIL(VY_ERR_ARG, "%s: workspace too small", who);
  const int64_t e = esize(p->dtype);
  const int B = p->B, d = p->d, h = p->h, hk = p->hk, dh = p->dh;
  char* w = (char*)p->ws;
  void* q = w; w += up(B * (int64_t)h * dh * e);
  void* ao = w; w += up(B * (int64_t)d * e);
  void* s = w; w += up(B * (int64_t)d * e);
  void* a = w; w += up(B * (int64_t)d * e);
  void* hb[2];
  hb[0] = w; w += up(B * (int64_t)d * e);
  hb[1] = w; w += up(B * (int64_t)d * e);
  void* mid = w; w += up(B * (int64_t)p->ffn * e);
  float* part = (float*)w;
  // M <= 32 rows in bf16: the two N = d projections (24 workgroups as plain skinny GEMMs) run split-K
  // over ~all CUs, their bias + residual + LayerNorm fused into the kernel that adds the partials
  const bool splitk = p->dtype == VY_BF16 && B <= 32 && d % 32 == 0 && d <= 8192 && p->ffn % 16 == 0;
  const float scale = 1.0f / sqrtf((float)dh);
  // the decode-only kernels of vy_decode.hip (short dependent chains): bf16, 64-wide heads, K of every projection a
  // chunk size they exist for (VY_DECODE_LEAN=0: the general launchers, for A/B runs and the equality test)
  if (g_decode_lean < 0) { const char* e = getenv("VY_DECODE_LEAN"); g_decode_lean = e ? atoi(e) : 1; }
  const bool lean = g_decode_lean && vy_dec_supported(B, d, h, hk, dh, p->ffn, p->dtype);
  hipStream_t hst = (hipStream_t)stream;
  const void* cur = x;
  for (int l = 0; l < p->num_layers; ++l) {
    const vy_decode_layer& L = p->layers[l];
    if (!pos_dev && pos < 0) VY_FAIL(VY_ERR_ARG, "%s: negative position", who);
    // K/V of this token go straight into the cache at index pos (host offset, or the kernel adds
    // *pos_dev itself when the step is being captured into a graph)
    const int64_t hpos = pos_dev ? 0 : pos;
    void* kdst = (char*)L.kcache + hpos * L.c_sl * e;
    void* vdst = (char*)L.vcache + hpos * L.c_sl * e;
    if (lean) {
      int rc = vy_dec_qkv(cur, L.wqkv, L.bqkv, p->cos_tab, p->sin_tab, pos, pos_dev, q, kdst, vdst, L.c_sb, L.c_sh, L.c_sl, B,
                          d, h, hk, hst);
      if (rc) return rc;
      rc = vy_attn_decode_ex(q, (int64_t)h * dh, dh, L.kcache, L.c_sb, L.c_sh, L.c_sl, L.vcache, L.c_sb, L.c_sh,
                             L.c_sl, ao, d, B, h, hk, pos + 1, pos_dev, dh, scale, p->dtype, stream);
      if (rc) return rc;
      // LN1's output feeds FFN1 only (the FFN residual is the layer input): the out-projection keeps whole rows per
      // workgroup (K = d is one chunk: 16-column workgroups, bias + residual in the epilogue) and FFN1 normalises its
      // input rows on the way in -- no split-K partials, no finish + LayerNorm launch: 6 links per layer instead of 7
      static const int lnfold = [] { const char* e = getenv("VY_DEC_LNFOLD"); return e ? atoi(e) : 1; }();
      if (lnfold) {
        rc = vy_dec_linear_ex(ao, d, L.wo, L.bo, cur, d, nullptr, nullptr, 0.f, s, d, B, d, d, VY_ACT_NONE, hst);
        if (rc) return rc;
        rc = vy_dec_linear_ex(s, d, L.w1, L.b1, nullptr, 0, L.ln1_w, L.ln1_b, p->eps_attn, mid, p->ffn, B, p->ffn, d, p->act, hst);
        if (rc) return rc;
      } else {
        rc = vy_dec_linear_res_ln(ao, d, L.wo, L.bo, cur, d, L.ln1_w, L.ln1_b, p->eps_attn, a, d, part, B, d, d, VY_ACT_NONE, hst);
        if (rc) return rc;
        rc = vy_dec_linear(a, d, L.w1, L.b1, mid, p->ffn, B, p->ffn, d, p->act, hst);
        if (rc) return rc;
      }
      void* nxt_l = hb[l & 1];   // FFN residual = the LAYER INPUT (reference models/decoder.py:241-250)
      rc = vy_dec_linear_res_ln(mid, p->ffn, L.w2, L.b2, cur, d, L.ln2_w, L.ln2_b, p->eps_ffn, nxt_l, d, part, B, d, p->ffn,
                                VY_ACT_NONE, hst);
      if (rc) return rc;
      cur = nxt_l;
      continue;
    }
    int rc = vy_qkv_rope_fwd_ex(cur, d, L.wqkv, d, L.bqkv, p->cos_tab, p->sin_tab, pos, pos_dev, q, (int64_t)h * dh,
                                dh, dh, kdst, L.c_sb, L.c_sh, L.c_sl, vdst, L.c_sb, L.c_sh, L.c_sl, B, 1, d, h, hk,
                                dh, p->dtype, stream);
    if (rc) return rc;
    rc = vy_attn_decode_ex(q, (int64_t)h * dh, dh, L.kcache, L.c_sb, L.c_sh, L.c_sl, L.vcache, L.c_sb, L.c_sh,
                           L.c_sl, ao, d, B, h, hk, pos + 1, pos_dev, dh, scale, p->dtype, stream);
    if (rc) return rc;
    if (splitk) {
      rc = vy_linear_res_ln_skinny(ao, d, L.wo, d, L.bo, cur, d, L.ln1_w, L.ln1_b, p->eps_attn, a, d, part, B, d, d, stream);
      if (rc) return rc;
    } else {
      rc = vy_linear_fwd(ao, d, L.wo, d, L.bo, cur, d, s, d, nullptr, B, d, d, VY_ACT_NONE, p->dtype, stream);
      if (rc) return rc;
      rc = vy_layernorm_fwd(s, d, L.ln1_w, L.ln1_b, a, d, nullptr, nullptr, B, d, p->eps_attn, p->dtype, stream);
      if (rc) return rc;
    }
    rc = vy_linear_fwd(a, d, L.w1, d, L.b1, nullptr, 0, mid, p->ffn, nullptr, B, p->ffn, d, p->act, p->dtype, stream);
    if (rc) return rc;
    // FFN residual = the LAYER INPUT (reference models/decoder.py:241-250)
    void* nxt = hb[l & 1];
    if (splitk) {
      rc = vy_linear_res_ln_skinny(mid, p->ffn, L.w2, p->ffn, L.b2, cur, d, L.ln2_w, L.ln2_b, p->eps_ffn, nxt, d, part, B,
                                   d, p->ffn, stream);
      if (rc) return rc;
    } else {
      rc = vy_linear_fwd(mid, p->ffn, L.w2, p->ffn, L.b2, cur, d, s, d, nullptr, B, d, p->ffn, VY_ACT_NONE, p->dtype, stream);
      if (rc) return rc;
      rc = vy_layernorm_fwd(s, d, L.ln2_w, L.ln2_b, nxt, d, nullptr, nullptr, B, d, p->eps_ffn, p->dtype, stream);
      if (rc) return rc;
    }
    cur = nxt;
  }
  if (hidden_out && hidden_out != cur) {
    if (hipMemcpyAsync(hidden_out, cur, B * (int64_t)d * e, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      VY_FAIL(VY_ERR_LAUNCH, "%s: copy of the hidden state failed", who);
  }
  if (logits) {
    int rc;
    if (lean) {   // LayerNorm(gelu(dense)) of the LM head: partial tile + finish
      rc = vy_dec_linear_res_ln(cur, d, p->head_wd, p->head_bd, nullptr, 0, p->head_ln_w, p->head_ln_b, p->eps_head, a, d,
                                part, B, d, d, VY_ACT_GELU_ERF, hst);
      if (rc) return rc;
    } else {
      rc = vy_linear_fwd(cur, d, p->head_wd, d, p->head_bd, nullptr, 0, s, d, nullptr, B, d, d, VY_ACT_GELU_ERF,
                         p->dtype, stream);
      if (rc) return rc;
      rc = vy_layernorm_fwd(s, d, p->head_ln_w, p->head_ln_b, a, d, nullptr, nullptr, B, d, p->eps_head, p->dtype, stream);
      if (rc) return rc;
    }
    rc = vy_linear_fwd(a, d, p->head_wv, d, p->head_bias, nullptr, 0, logits, ldv, nullptr, B, p->vocab, d,
                       VY_ACT_NONE, p->dtype, stream);
    if (rc) return rc;
  }
  return VY_OK;
}

extern "C" int64_t vy_gemma_ws_bytes(int32_t B, int32_t d, int32_t h, int32_t dh, int32_t ffn, int32_t dtype) {
  const int64_t e = esize(dtype);
  // normed input, q, attention output, two hidden buffers (x after attention / layer output), [gate|up], act
  return 4 * up(B * (int64_t)d * e) + 2 * up(B * (int64_t)h * dh * e) + up(2 * B * (int64_t)ffn * e) +
         up(B * (int64_t)ffn * e) + 4096;
}

extern "C" int vy_gemma_decoder_step(const vy_gemma_plan* p, const void* x, int64_t pos, void* logits, int64_t ldv,
                                     void* stream) {
  const char* who = "vy_gemma_decoder_step";
  if (!p || !x || !p->layers || !p->ws || !logits) VY_FAIL(VY_ERR_ARG, "%s: null plan/input/workspace/output", who);
  if (pos < 0) VY_FAIL(VY_ERR_ARG, "%s: negative position", who);
  if (p->ws_bytes < vy_gemma_ws_bytes(p->B, p->d, p->h, p->dh, p->ffn, p->dtype)) VY_FAIL(VY_ERR_ARG, "%s: workspace too small", who);
  const int64_t e = esize(p->dtype);
  const int B = p->B, d = p->d, h = p->h, hk = p->hk, dh = p->dh, ffn = p->ffn;
  char* w = (char*)p->ws;
  void* n = w; w += up(B * (int64_t)d * e);
  void* q = w; w += up(B * (int64_t)h * dh * e);
  void* ao = w; w += up(B * (int64_t)h * dh * e);
  void* x1 = w; w += up(B * (int64_t)d * e);
  void* hb[2];
  hb[0] = w; w += up(B * (int64_t)d * e);
  hb[1] = w; w += up(B * (int64_t)d * e);
  void* gu = w; w += up(2 * B * (int64_t)ffn * e);
  void* act = w;
  const float scale = 1.0f / sqrtf((float)dh);
  const void* cur = x;
  int rc;
  // Single-sequence bf16 decode (B <= 4) is a chain of matrix-vector products: the RMSNorm of each projection's
  // input is folded into the product (every wave redoes the row statistics from the chunks it reads anyway) and
  // the GeGLU into the gate/up product -- 6 launches per layer instead of 9 (VY_GEMMA_FUSED=0: the unfused chain).
  static const int fused_env = [] { const char* e = getenv("VY_GEMMA_FUSED"); return e ? atoi(e) : 1; }();
  const bool fused = fused_env && p->dtype == VY_BF16 && B <= 4 && d % 8 == 0 && ffn % 8 == 0 && ((int64_t)h * dh) % 8 == 0;
  // (folding the RMSNorm into the products as well -- VY_GEMMA_FUSED=2 -- measured SLOWER, 2.05 vs 1.76 ms per token:
  // every one of the N waves of a product redoes the row statistics and the normalisation of its input)
  const bool fold_norm = fused_env >= 2;
  // a plan whose weights already carry the norms' (1 + w) can only run on the chain that skips the RMSNorm launches: on
  // any other chain the factor would be applied twice and the tokens would be wrong without a word
  if ((p->flags & VY_GEMMA_PRESCALED) && !fused)
    VY_FAIL(VY_ERR_ARG, "%s: the plan's weights are pre-scaled by the RMSNorm weights, which needs the fused matrix-vector "
            "chain (bf16, B <= 4, d / ffn / h*dh multiples of 8, VY_GEMMA_FUSED != 0)", who);
  const bool prescaled = fused && (p->flags & VY_GEMMA_PRESCALED);
  static const int rope_env = [] { const char* e = getenv("VY_GEMMA_ROPE_FUSED"); return e ? atoi(e) : 1; }();
  const bool rope_fused = fused && rope_env && p->cos_tab && dh % 4 == 0 && (dh & (dh - 1)) == 0;
  if (g_gemma_lean < 0) { const char* e = getenv("VY_GEMMA_LEAN"); g_gemma_lean = e ? atoi(e) : 1; }
  const bool lean1 = g_gemma_lean && B == 1 && !fold_norm;
  for (int l = 0; l < p->num_layers; ++l) {
    const vy_gemma_layer& L = p->layers[l];
    void* kdst = (char*)L.kcache + pos * L.c_sl * e;
    void* vdst = (char*)L.vcache + pos * L.c_sl * e;
    void* nxt = hb[l & 1];
    if (fused && lean1) {
      // one sequence: the straight-line matrix-vector kernels of vy_decode.hip (norms folded when the plan's weights
      // are pre-scaled, rotary pairs swapped inside the QKV product); any unsupported shape -> the general chain below
      hipStream_t hs = (hipStream_t)stream;
      const void* qin = cur;
      if (!prescaled) {
        if ((rc = vy_rmsnorm_fwd(cur, d, L.ln_in, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
        qin = n;
      }
      rc = vy_dec_gemv1_qkv(qin, L.wqkv, L.bqkv, q, kdst, vdst, L.c_sh, h, hk, dh, rope_fused ? p->cos_tab : nullptr, p->sin_tab,
                            pos, d, prescaled, p->eps, hs);
      if (rc != VY_OK && rc != VY_ERR_UNSUPPORTED) return rc;   // only an unsupported shape falls through to the general chain
      if (rc == VY_OK) {
        if (p->cos_tab && !rope_fused &&
            (rc = vy_rope_qk(q, (int64_t)h * dh, dh, dh, h, kdst, L.c_sb, L.c_sh, L.c_sl, hk, p->cos_tab, p->sin_tab,
                             pos, B, 1, dh, p->dtype, hs))) return rc;
        if ((rc = vy_attn_decode_ex(q, (int64_t)h * dh, dh, L.kcache, L.c_sb, L.c_sh, L.c_sl, L.vcache, L.c_sb, L.c_sh, L.c_sl,
                                    ao, (int64_t)h * dh, B, h, hk, pos + 1, nullptr, dh, scale, p->dtype, stream))) return rc;
        if ((rc = vy_dec_gemv1(ao, L.wo, L.bo, cur, x1, d, h * dh, 0, 0.f, hs))) {
          if (rc != VY_ERR_UNSUPPORTED) return rc;
          if ((rc = vy_gemv_norm(ao, (int64_t)h * dh, L.wo, (int64_t)h * dh, L.bo, nullptr, 0.f, cur, d, x1, d, B, d,
                                 (int64_t)h * dh, stream))) return rc;
        }
        const void* gin = x1;
        if (!prescaled) {
          if ((rc = vy_rmsnorm_fwd(x1, d, L.ln_post, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
          gin = n;
        }
        if ((rc = vy_dec_gemv1_gated(gin, L.wgu, act, ffn, d, prescaled, p->eps, hs))) {
          if (rc != VY_ERR_UNSUPPORTED) return rc;
          if ((rc = vy_gemv_gated(gin, d, L.wgu, d, prescaled ? VY_NORM_PRESCALED : nullptr, p->eps, act, ffn, B, ffn, d,
                                  VY_ACT_GELU_TANH, stream))) return rc;
        }
        if ((rc = vy_dec_gemv1(act, L.wdown, nullptr, x1, nxt, d, ffn, 0, 0.f, hs))) {
          if (rc != VY_ERR_UNSUPPORTED) return rc;
          if ((rc = vy_gemv_norm(act, ffn, L.wdown, ffn, nullptr, nullptr, 0.f, x1, d, nxt, d, B, d, ffn, stream))) return rc;
        }
        cur = nxt;
        continue;
      }
    }
    if (fused) {
      // prescaled: the weights carry the norms' (1 + w), the products scale themselves by the row's rsqrt(mean x^2):
      // no RMSNorm launch; rope_fused: rotary pairs swapped inside the QKV product's workgroups: no RoPE launch
      const void* qin = cur;
      const void* qnorm = prescaled ? VY_NORM_PRESCALED : (fold_norm ? L.ln_in : nullptr);
      if (!fold_norm && !prescaled) {
        if ((rc = vy_rmsnorm_fwd(cur, d, L.ln_in, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
        qin = n;
      }
      if ((rc = vy_gemv_qkv_norm(qin, d, L.wqkv, d, L.bqkv, qnorm, p->eps, q, (int64_t)h * dh, dh, kdst, L.c_sb, L.c_sh,
                                 L.c_sl, vdst, L.c_sb, L.c_sh, L.c_sl, B, d, h, hk, dh, rope_fused ? p->cos_tab : nullptr,
                                 p->sin_tab, pos, stream))) return rc;
      if (p->cos_tab && !rope_fused &&
          (rc = vy_rope_qk(q, (int64_t)h * dh, dh, dh, h, kdst, L.c_sb, L.c_sh, L.c_sl, hk, p->cos_tab, p->sin_tab,
                           pos, B, 1, dh, p->dtype, (hipStream_t)stream))) return rc;
      if ((rc = vy_attn_decode_ex(q, (int64_t)h * dh, dh, L.kcache, L.c_sb, L.c_sh, L.c_sl, L.vcache, L.c_sb, L.c_sh, L.c_sl,
                                  ao, (int64_t)h * dh, B, h, hk, pos + 1, nullptr, dh, scale, p->dtype, stream))) return rc;
      if ((rc = vy_gemv_norm(ao, (int64_t)h * dh, L.wo, (int64_t)h * dh, L.bo, nullptr, 0.f, cur, d, x1, d, B, d,
                             (int64_t)h * dh, stream))) return rc;
      const void* gin = x1;
      const void* gnorm = prescaled ? VY_NORM_PRESCALED : (fold_norm ? L.ln_post : nullptr);
      if (!fold_norm && !prescaled) {
        if ((rc = vy_rmsnorm_fwd(x1, d, L.ln_post, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
        gin = n;
      }
      if ((rc = vy_gemv_gated(gin, d, L.wgu, d, gnorm, p->eps, act, ffn, B, ffn, d, VY_ACT_GELU_TANH, stream))) return rc;
      if ((rc = vy_gemv_norm(act, ffn, L.wdown, ffn, nullptr, nullptr, 0.f, x1, d, nxt, d, B, d, ffn, stream))) return rc;
      cur = nxt;
      continue;
    }
    if ((rc = vy_rmsnorm_fwd(cur, d, L.ln_in, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
    if ((rc = vy_qkv_rope_fwd_ex(n, d, L.wqkv, d, L.bqkv, p->cos_tab, p->sin_tab, pos, nullptr, q, (int64_t)h * dh, dh, dh,
                                 kdst, L.c_sb, L.c_sh, L.c_sl, vdst, L.c_sb, L.c_sh, L.c_sl, B, 1, d, h, hk, dh, p->dtype,
                                 stream))) return rc;
    if ((rc = vy_attn_decode_ex(q, (int64_t)h * dh, dh, L.kcache, L.c_sb, L.c_sh, L.c_sl, L.vcache, L.c_sb, L.c_sh, L.c_sl,
                                ao, (int64_t)h * dh, B, h, hk, pos + 1, nullptr, dh, scale, p->dtype, stream))) return rc;
    if ((rc = vy_linear_fwd(ao, (int64_t)h * dh, L.wo, (int64_t)h * dh, L.bo, cur, d, x1, d, nullptr, B, d, (int64_t)h * dh,
                            VY_ACT_NONE, p->dtype, stream))) return rc;
    if ((rc = vy_rmsnorm_fwd(x1, d, L.ln_post, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
    if ((rc = vy_linear_fwd(n, d, L.wgu, d, nullptr, nullptr, 0, gu, 2 * (int64_t)ffn, nullptr, B, 2 * (int64_t)ffn, d,
                            VY_ACT_NONE, p->dtype, stream))) return rc;
    if ((rc = vy_gated_act_fwd(gu, 2 * (int64_t)ffn, act, ffn, B, ffn, VY_ACT_GELU_TANH, p->dtype, stream))) return rc;
    if ((rc = vy_linear_fwd(act, ffn, L.wdown, ffn, nullptr, x1, d, nxt, d, nullptr, B, d, ffn, VY_ACT_NONE, p->dtype,
                            stream))) return rc;
    cur = nxt;
  }
  if (fused && fold_norm)   // final norm folded into the vocabulary product
    return vy_gemv_norm(cur, d, p->head_w, d, nullptr, p->norm_w, p->eps, nullptr, 0, logits, ldv, B, p->vocab, d, stream);
  if ((rc = vy_rmsnorm_fwd(cur, d, p->norm_w, n, d, B, d, p->eps, 1.0f, p->dtype, stream))) return rc;
  return vy_linear_fwd(n, d, p->head_w, d, nullptr, nullptr, 0, logits, ldv, nullptr, B, p->vocab, d, VY_ACT_NONE,
                       p->dtype, stream);
}
