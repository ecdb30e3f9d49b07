// Sampling front end of the generate loops: the per-token work after the LM head.
//   vy_greedy_step     reference models/decoder.py:478-507 (top-1, prompt forcing, EOS bookkeeping)
//   vy_sampling_probs  reference logits_processors.py:13-16, 59-63, 73-81, 92-102
// One 1024-thread workgroup per row.  Selection (k-th largest value, nucleus cut) is a radix search over
// 4-bit digits of an order-preserving key: every thread keeps the 16 bin counts / masses of its own
// elements in registers and the workgroup adds them in a fixed order -- no atomics, so the result
// does not depend on scheduling.  The row is re-read from L2 for every pass (V = 50265 bf16 = 100 KB).
#include "vy_common.h"
#include <stdlib.h>

namespace {

constexpr int ST = 1024, SW = ST / 64;

__device__ __forceinline__ unsigned fkey(float v) {   // larger float <=> larger key
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
template <typename T> __device__ __forceinline__ float ldf(const T* p, int i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, int i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16>(const bf16* p, int i) { return (float)p[i]; }

// sum of NR per-thread values over the workgroup, in a fixed order; every thread gets the result
template <int NR, typename A>
__device__ __forceinline__ void block_sum(A (&v)[NR], A* red /* [SW][NR] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    A x = v[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (lane == 0) red[wave * NR + r] = x;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    A x = 0;
    for (int w = 0; w < SW; ++w) x += red[w * NR + r];
    v[r] = x;
  }
  __syncthreads();
}
__device__ __forceinline__ float block_max(float m, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  m = vy_wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < SW; ++w) r = fmaxf(r, red[w]);
  __syncthreads();
  return r;
}

template <typename T>
__global__ __launch_bounds__(ST) void greedy_step_kernel(const T* __restrict__ logits, int64_t ldl, int V,
                                                         int64_t* __restrict__ tokens, int64_t ldt, int64_t cur_pos,
                                                         const uint8_t* __restrict__ text_mask, int64_t ldm,
                                                         const int64_t* __restrict__ eos_ids, int n_eos,
                                                         uint8_t* __restrict__ eos_reached, int32_t* __restrict__ not_done) {
  __shared__ float bv[SW];
  __shared__ int bi[SW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool forced = text_mask && text_mask[(int64_t)b * ldm + cur_pos];
  int best_i = 0x7fffffff;
  float best_v = -INFINITY;
  if (!forced) {   // block-uniform
    const T* row = logits + (int64_t)b * ldl;
    auto take = [&](float v, int i) {
      if (v > best_v || (v == best_v && i < best_i) || best_i == 0x7fffffff) { best_v = v; best_i = i; }
    };
    int i0 = 0;
    if constexpr (std::is_same<T, bf16>::value) {
      // 16-byte chunks, all of a thread's chunks requested before the first compare (V = 50265: 7 per thread;
      // one 2-byte load per trip made this kernel 24 us of dependent round trips)
      if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        const int nch = V >> 3;
        constexpr int CPT = 8;
        for (int c0 = 0; c0 < nch; c0 += ST * CPT) {
          bf16x8 ch[CPT];
#pragma unroll
          for (int u = 0; u < CPT; ++u) {
            const int c = c0 + u * ST + tid;
            if (c < nch) ch[u] = *reinterpret_cast<const bf16x8*>(row + (int64_t)c * 8);
          }
#pragma unroll
          for (int u = 0; u < CPT; ++u) {
            const int c = c0 + u * ST + tid;
            if (c < nch) {
#pragma unroll
              for (int e = 0; e < 8; ++e) take((float)ch[u][e], c * 8 + e);
            }
          }
        }
        i0 = nch << 3;
      }
    }
    for (int i = i0 + tid; i < V; i += ST) take(ldf(row, i), i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best_v, o, 64);
      const int oi = __shfl_xor(best_i, o, 64);
      if (oi != 0x7fffffff && (best_i == 0x7fffffff || ov > best_v || (ov == best_v && oi < best_i))) { best_v = ov; best_i = oi; }
    }
    if (lane == 0) { bv[wave] = best_v; bi[wave] = best_i; }
    __syncthreads();
  }
  if (tid == 0) {
    int64_t next;
    if (forced) {
      next = tokens[(int64_t)b * ldt + cur_pos];
    } else {
      float v = bv[0];
      int i = bi[0];
      for (int w = 1; w < SW; ++w)
        if (bi[w] != 0x7fffffff && (i == 0x7fffffff || bv[w] > v || (bv[w] == v && bi[w] < i))) { v = bv[w]; i = bi[w]; }
      next = i == 0x7fffffff ? 0 : i;   // a row of NaNs has no maximum: token 0, not an out-of-range id
      tokens[(int64_t)b * ldt + cur_pos] = next;
    }
    bool eos = eos_reached[b];
    if (!forced)
      for (int e = 0; e < n_eos; ++e) eos |= (next == eos_ids[e]);
    eos_reached[b] = eos;
    if (!eos && not_done) atomicAdd(not_done, 1);
  }
}

// ---- two-stage top-1 for wide vocabularies ---------------------------------------------------------
// One workgroup per row walks 100 KB (V = 50265) to 514 KB (V = 257216, one sequence) of logits: 13 / 35 us of a
// decode step.  Here stage 1 spreads a row over chunks of 8192 logits (one 256-thread workgroup each, every load
// issued before the first compare) and leaves a (value, index) candidate per chunk; stage 2 is greedy_step_kernel's
// bookkeeping on <= 64 candidates per row.  Same order (value descending, index ascending; a row of NaNs -> 0).
constexpr int GP_CHUNK = 8192, GP_MAXCH = 64, GP_MAXB = 256;
__device__ float g_gp_v[GP_MAXB * GP_MAXCH];
__device__ int g_gp_i[GP_MAXB * GP_MAXCH];

__device__ __forceinline__ void gp_take(float v, int i, float& bv, int& bi) {
  if (v > bv || (v == bv && i < bi) || bi == 0x7fffffff) { bv = v; bi = i; }
}
__device__ __forceinline__ void gp_merge(float ov, int oi, float& bv, int& bi) {
  if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
}

template <typename T>
__global__ __launch_bounds__(256) void greedy_part_kernel(const T* __restrict__ logits, int64_t ldl, int V, int nch,
                                                          const uint8_t* __restrict__ text_mask, int64_t ldm, int64_t cur_pos) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (text_mask && text_mask[(int64_t)b * ldm + cur_pos]) return;   // forced position: stage 2 copies the prompt token
  const T* row = logits + (int64_t)b * ldl;
  const int lo = ch * GP_CHUNK, hi = min(V, lo + GP_CHUNK);
  int best_i = 0x7fffffff;
  float best_v = -INFINITY;
  if constexpr (std::is_same<T, bf16>::value) {
    if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
      constexpr int CPT = GP_CHUNK / 8 / 256;   // 4 chunks of 8 per thread
      bf16x8 c8[CPT];
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int e0 = lo + (u * 256 + tid) * 8;
        c8[u] = *reinterpret_cast<const bf16x8*>(row + (e0 + 8 <= V ? e0 : 0));
      }
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int e0 = lo + (u * 256 + tid) * 8;
        if (e0 + 8 <= V) {
#pragma unroll
          for (int e = 0; e < 8; ++e) gp_take((float)c8[u][e], e0 + e, best_v, best_i);
        } else {
          for (int e = e0; e < hi; ++e) gp_take(ldf(row, e), e, best_v, best_i);
        }
      }
    } else {
      for (int i = lo + tid; i < hi; i += 256) gp_take(ldf(row, i), i, best_v, best_i);
    }
  } else {
    for (int i = lo + tid; i < hi; i += 256) gp_take(ldf(row, i), i, best_v, best_i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best_v, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    gp_merge(ov, oi, best_v, best_i);
  }
  if (lane == 0) { bv[wave] = best_v; bi[wave] = best_i; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) gp_merge(bv[w], bi[w], best_v, best_i);
    g_gp_v[b * GP_MAXCH + ch] = best_v;
    g_gp_i[b * GP_MAXCH + ch] = best_i;
  }
}

__global__ __launch_bounds__(64) void greedy_finish_kernel(int nch, int64_t* __restrict__ tokens, int64_t ldt, int64_t cur_pos,
                                                           const uint8_t* __restrict__ text_mask, int64_t ldm,
                                                           const int64_t* __restrict__ eos_ids, int n_eos,
                                                           uint8_t* __restrict__ eos_reached, int32_t* __restrict__ not_done) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const bool forced = text_mask && text_mask[(int64_t)b * ldm + cur_pos];
  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  if (!forced) {
    if (lane < nch) { best_v = g_gp_v[b * GP_MAXCH + lane]; best_i = g_gp_i[b * GP_MAXCH + lane]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best_v, o, 64);
      const int oi = __shfl_xor(best_i, o, 64);
      gp_merge(ov, oi, best_v, best_i);
    }
  }
  if (lane == 0) {
    int64_t next;
    if (forced) {
      next = tokens[(int64_t)b * ldt + cur_pos];
    } else {
      next = best_i == 0x7fffffff ? 0 : best_i;
      tokens[(int64_t)b * ldt + cur_pos] = next;
    }
    bool eos = eos_reached[b];
    if (!forced)
      for (int e = 0; e < n_eos; ++e) eos |= (next == eos_ids[e]);
    eos_reached[b] = eos;
    if (!eos && not_done) atomicAdd(not_done, 1);
  }
}

template <typename T>
__global__ __launch_bounds__(ST) void sampling_probs_kernel(const T* __restrict__ logits, int64_t ldl, int V,
                                                            float inv_t, int top_k, float top_p,
                                                            float* __restrict__ probs, int64_t ldp) {
  __shared__ float fred[SW * 16];
  __shared__ int ired[SW * 16];
  const int tid = threadIdx.x;
  const T* row = logits + (int64_t)blockIdx.x * ldl;
  float* out = probs + (int64_t)blockIdx.x * ldp;

  float m = -INFINITY;
  for (int i = tid; i < V; i += ST) m = fmaxf(m, ldf(row, i));
  m = block_max(m, fred);

  // ---- top-k: key of the k-th largest value (values equal to it are all kept, reference :59-63) ----
  unsigned keep_key = 0;
  if (top_k > 0 && top_k < V) {
    unsigned prefix = 0;
    int kk = top_k;
    for (int shift = 28; shift >= 0; shift -= 4) {
      int cnt[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) cnt[d] = 0;
      for (int i = tid; i < V; i += ST) {
        const unsigned key = fkey(ldf(row, i));
        const bool match = shift == 28 || ((key ^ prefix) >> (shift + 4)) == 0;
        const int dg = (key >> shift) & 15;
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d] += (match && dg == d);
      }
      block_sum<16, int>(cnt, ired);
      int acc = 0, sel = 0;
#pragma unroll
      for (int d = 15; d >= 0; --d) {
        if (acc >= 0) {
          if (acc + cnt[d] >= kk) { sel = d; kk -= acc; acc = -1; }
          else acc += cnt[d];
        }
      }
      prefix |= (unsigned)sel << shift;
    }
    keep_key = prefix;
  }

  // ---- nucleus: the sorted prefix up to and including the first element whose cumulative softmax of the
  //      unscaled (top-k masked) logits exceeds top_p (reference :74-80) ----
  if (top_p > 0.f && top_p < 1.f) {
    float z[1] = {0.f};
    for (int i = tid; i < V; i += ST) {
      const float v = ldf(row, i);
      if (fkey(v) >= keep_key) z[0] += expf(v - m);
    }
    block_sum<1, float>(z, fred);
    const float target = top_p * z[0];
    unsigned prefix = 0;
    float above = 0.f;
    for (int shift = 28; shift >= 0; shift -= 4) {
      float mass[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) mass[d] = 0.f;
      for (int i = tid; i < V; i += ST) {
        const float v = ldf(row, i);
        const unsigned key = fkey(v);
        const bool match = key >= keep_key && (shift == 28 || ((key ^ prefix) >> (shift + 4)) == 0);
        const int dg = (key >> shift) & 15;
        const float e = match ? expf(v - m) : 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) mass[d] += (dg == d) ? e : 0.f;
      }
      block_sum<16, float>(mass, fred);
      int sel = -1, lowest = 0;
      bool have_low = false;
#pragma unroll
      for (int d = 15; d >= 0; --d) {
        if (sel < 0) {
          if (above + mass[d] > target) sel = d;
          else above += mass[d];
        }
        if (mass[d] > 0.f) { lowest = d; have_low = true; }
      }
      // the parent bin crossed the target; if rounding of the finer sums hides that, the crossing
      // element is the last one of the bin
      if (sel < 0) { sel = have_low ? lowest : 0; above -= mass[sel]; }
      prefix |= (unsigned)sel << shift;
    }
    keep_key = prefix > keep_key ? prefix : keep_key;
  }

  float zt[1] = {0.f};
  for (int i = tid; i < V; i += ST) {
    const float v = ldf(row, i);
    if (fkey(v) >= keep_key) zt[0] += expf((v - m) * inv_t);
  }
  block_sum<1, float>(zt, fred);
  const float rz = 1.0f / zt[0];
  for (int i = tid; i < V; i += ST) {
    const float v = ldf(row, i);
    out[i] = fkey(v) >= keep_key ? expf((v - m) * inv_t) * rz : 0.f;
  }
}

}  // namespace

extern "C" int vy_greedy_step(const void* logits, int64_t ldl, int64_t B, int64_t V, int dtype, int64_t* tokens,
                              int64_t ldt, int64_t cur_pos, const uint8_t* text_mask, int64_t ldm,
                              const int64_t* eos_ids, int32_t n_eos, uint8_t* eos_reached, int32_t* not_done,
                              void* stream) {
  if (!logits || !tokens || !eos_reached || B <= 0 || V <= 0 || V > 0x7ffffff0 || cur_pos < 0 || cur_pos >= ldt ||
      n_eos < 0 || (n_eos > 0 && !eos_ids))
    VY_FAIL(VY_ERR_ARG, "vy_greedy_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype != VY_BF16 && dtype != VY_F32) VY_FAIL(VY_ERR_ARG, "vy_greedy_step: bad dtype %d", dtype);
  static const int two_stage = [] { const char* e = getenv("VY_GREEDY_TWO_STAGE"); return e ? atoi(e) : 1; }();
  const int64_t nch = (V + GP_CHUNK - 1) / GP_CHUNK;
  if (two_stage && V >= 2 * GP_CHUNK && nch <= GP_MAXCH && B <= GP_MAXB) {
    const dim3 grid((unsigned)nch, (unsigned)B);
    if (dtype == VY_BF16)
      hipLaunchKernelGGL(greedy_part_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, ldl, (int)V, (int)nch, text_mask,
                         ldm, cur_pos);
    else
      hipLaunchKernelGGL(greedy_part_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, ldl, (int)V, (int)nch,
                         text_mask, ldm, cur_pos);
    hipLaunchKernelGGL(greedy_finish_kernel, dim3((unsigned)B), dim3(64), 0, st, (int)nch, tokens, ldt, cur_pos, text_mask, ldm,
                       eos_ids, (int)n_eos, eos_reached, not_done);
    VY_CHECK_LAUNCH("vy_greedy_step");
    return VY_OK;
  }
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(greedy_step_kernel<bf16>, dim3((unsigned)B), dim3(ST), 0, st, (const bf16*)logits, ldl, (int)V, tokens,
                       ldt, cur_pos, text_mask, ldm, eos_ids, (int)n_eos, eos_reached, not_done);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(greedy_step_kernel<float>, dim3((unsigned)B), dim3(ST), 0, st, (const float*)logits, ldl, (int)V,
                       tokens, ldt, cur_pos, text_mask, ldm, eos_ids, (int)n_eos, eos_reached, not_done);
  else VY_FAIL(VY_ERR_ARG, "vy_greedy_step: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_greedy_step");
  return VY_OK;
}

extern "C" int vy_sampling_probs(const void* logits, int64_t ldl, int64_t B, int64_t V, int dtype, float temperature,
                                 int32_t top_k, float top_p, float* probs, int64_t ldp, void* stream) {
  if (!logits || !probs || B <= 0 || V <= 0 || V > 0x7ffffff0 || ldp < V || ldl < V)
    VY_FAIL(VY_ERR_ARG, "vy_sampling_probs: bad arguments");
  if (!(temperature > 0.f)) VY_FAIL(VY_ERR_ARG, "vy_sampling_probs: temperature must be positive");
  if (top_k < 0 || !(top_p >= 0.f)) VY_FAIL(VY_ERR_ARG, "vy_sampling_probs: top_k and top_p must not be negative");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(sampling_probs_kernel<bf16>, dim3((unsigned)B), dim3(ST), 0, st, (const bf16*)logits, ldl, (int)V,
                       1.0f / temperature, (int)top_k, top_p, probs, ldp);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(sampling_probs_kernel<float>, dim3((unsigned)B), dim3(ST), 0, st, (const float*)logits, ldl, (int)V,
                       1.0f / temperature, (int)top_k, top_p, probs, ldp);
  else VY_FAIL(VY_ERR_ARG, "vy_sampling_probs: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_sampling_probs");
  return VY_OK;
}
