// HBM-bound row kernels of the block: LayerNorm forward/backward, standalone RoPE, casts,
// transposes and the fused AdamW step.  All are one pass over their operands with 16-byte
// accesses per lane; LayerNorm keeps the whole row in registers (one wave per row).
#include "vy_common.h"
#include <stdarg.h>

// ---- error string ------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vy_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vy_last_error(void) { return g_err; }
extern "C" int vy_abi_version(void) { return 5; }

namespace {

template <typename T> struct Chunk;  // one 16-byte chunk = VEC elements
template <> struct Chunk<bf16> {
  static constexpr int VEC = 8;
  typedef bf16x8 Raw;
  static __device__ __forceinline__ void unpack(const Raw& t, float* v) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void load(const bf16* p, float* v) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void store(bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};
template <> struct Chunk<float> {
  static constexpr int VEC = 4;
  typedef f32x4 Raw;
  static __device__ __forceinline__ void unpack(const Raw& t, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  static __device__ __forceinline__ void load(const float* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};

// ---- LayerNorm forward: one wave per row, CH chunks per lane -------------------------------
// mean, then variance about the mean (two reductions on registers): the same two-pass
// formulation as aten's CPU LayerNorm closely enough for 1e-6 agreement.
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                            const T* __restrict__ gamma,
                                                            const T* __restrict__ beta, T* __restrict__ y,
                                                            int64_t ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int64_t M, int N,
                                                            float eps) {
  constexpr int VEC = Chunk<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = N / VEC;
  float v[CH][VEC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      Chunk<T>::load(x + row * ldx + (int64_t)ch * VEC, v[c]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += v[c][e];
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[c][e] = 0.f;
    }
  }
  const float mean = vy_wave_sum(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = v[c][e] - mean; q += d * d; }
    }
  }
  const float var = vy_wave_sum(q) / (float)N;
  const float rstd = rsqrtf(var + eps);
  // rsqrtf is approximate on AMD; one Newton step brings it to fp32 round-off
  const float rstd_r = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float g[VEC], b[VEC], o[VEC];
      Chunk<T>::load(gamma + (int64_t)ch * VEC, g);
      Chunk<T>::load(beta + (int64_t)ch * VEC, b);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = (v[c][e] - mean) * rstd_r * g[e] + b[e];
      Chunk<T>::store(y + row * ldy + (int64_t)ch * VEC, o);
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd_r;
  }
}

// ---- Gemma RMSNorm: y = x * rsqrt(mean(x^2) + eps) * (1 + w), statistics in fp32 -------------
// (Examples/paligemma.ipynb cell 11, GemmaRMSNorm)
template <typename T, int CH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                          const T* __restrict__ w, T* __restrict__ y, int64_t ldy,
                                                          int64_t M, int N, float eps, float w_offset) {
  constexpr int VEC = Chunk<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = N / VEC;
  float v[CH][VEC];
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      Chunk<T>::load(x + row * ldx + (int64_t)ch * VEC, v[c]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) q += v[c][e] * v[c][e];
    }
  }
  const float ms = vy_wave_sum(q) / (float)N + eps;
  float r = rsqrtf(ms);
  r = r * (1.5f - 0.5f * ms * r * r);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float g[VEC], o[VEC];
      Chunk<T>::load(w + (int64_t)ch * VEC, g);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = v[c][e] * r * (w_offset + g[e]);
      Chunk<T>::store(y + row * ldy + (int64_t)ch * VEC, o);
    }
  }
}

// ---- gated activation: out[m, i] = act(gu[m, i]) * gu[m, I + i]  (GeGLU / SwiGLU-style MLPs) ----
template <typename T, int ACT>
__global__ void gated_act_kernel(const T* __restrict__ gu, int64_t ldg, T* __restrict__ out, int64_t ldo,
                                 int64_t M, int I) {
  constexpr int VEC = Chunk<T>::VEC;
  const int nch = I / VEC;
  const int64_t total = M * nch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nch;
    const int c = (int)(i - m * nch) * VEC;
    float a[VEC], b[VEC], o[VEC];
    Chunk<T>::load(gu + m * ldg + c, a);
    Chunk<T>::load(gu + m * ldg + I + c, b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = vy_act_fwd<ACT>(a[e]) * b[e];
    Chunk<T>::store(out + m * ldo + c, o);
  }
}

// ---- LayerNorm backward -------------------------------------------------------------------
// wave w of the grid walks rows w, w+W, ...; a lane always owns the same columns, so dgamma /
// dbeta partials accumulate in registers and are written once per wave into ws[w][N].
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx,
    const T* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    T* __restrict__ dx, int64_t lddx, float* __restrict__ ws, int64_t M, int N, int W) {
  constexpr int VEC = Chunk<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);  // W = 4 * gridDim.x waves walk the rows
  const int nch = N / VEC;
  float g[CH][VEC], dg[CH][VEC], db[CH][VEC];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { dg[c][e] = 0.f; db[c][e] = 0.f; g[c][e] = 0.f; }
    if (ch < nch) Chunk<T>::load(gamma + (int64_t)ch * VEC, g[c]);
  }
  // software pipeline over the rows of this wave: the 16-byte loads (and the row statistics) of
  // the NEXT row are issued before the reductions of the current one, so the HBM latency of a row
  // hides behind the previous row's math instead of serialising with it
  typedef typename Chunk<T>::Raw Raw;
  Raw xr[CH], dr[CH], xn[CH], dn[CH];
  float mu = 0.f, rs = 0.f, mu_n = 0.f, rs_n = 0.f;
  auto fetch = [&](int64_t row, Raw (&xa)[CH], Raw (&da)[CH], float& m_, float& r_) {
    m_ = mean[row]; r_ = rstd[row];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        xa[c] = *reinterpret_cast<const Raw*>(x + row * ldx + (int64_t)ch * VEC);
        da[c] = *reinterpret_cast<const Raw*>(dy + row * lddy + (int64_t)ch * VEC);
      }
    }
  };
  if (wid < M) fetch(wid, xr, dr, mu, rs);
  for (int64_t row = wid; row < M; row += W) {
    const bool more = row + W < M;
    if (more) fetch(row + W, xn, dn, mu_n, rs_n);
    float xh[CH][VEC], gy[CH][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float xv[VEC], dv[VEC];
        Chunk<T>::unpack(xr[c], xv);
        Chunk<T>::unpack(dr[c], dv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[c][e] = (xv[e] - mu) * rs;
          gy[c][e] = dv[e] * g[c][e];
          s1 += gy[c][e];
          s2 += gy[c][e] * xh[c][e];
          dg[c][e] += dv[e] * xh[c][e];
          db[c][e] += dv[e];
        }
      }
    }
    s1 = vy_wave_sum(s1) / (float)N;
    s2 = vy_wave_sum(s2) / (float)N;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rs * (gy[c][e] - s1 - xh[c][e] * s2);
        Chunk<T>::store(dx + row * lddx + (int64_t)ch * VEC, o);
      }
    }
    if (more) {
#pragma unroll
      for (int c = 0; c < CH; ++c) { xr[c] = xn[c]; dr[c] = dn[c]; }
      mu = mu_n; rs = rs_n;
    }
  }
  // the block's 4 waves are summed through LDS first, so one slab row per BLOCK is written
  __shared__ float red[2][3][64 * CH * VEC];
  const int wv = threadIdx.x >> 6;
  if (wv > 0) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        red[0][wv - 1][(c * VEC + e) * 64 + lane] = dg[c][e];
        red[1][wv - 1][(c * VEC + e) * 64 + lane] = db[c][e];
      }
  }
  __syncthreads();
  if (wv != 0) return;
  const int WB = W / 4;  // slab rows = blocks
  float* wg = ws + (int64_t)blockIdx.x * N;
  float* wb = ws + (int64_t)WB * N + (int64_t)blockIdx.x * N;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float a = dg[c][e], b = db[c][e];
#pragma unroll
        for (int w2 = 0; w2 < 3; ++w2) {
          a += red[0][w2][(c * VEC + e) * 64 + lane];
          b += red[1][w2][(c * VEC + e) * 64 + lane];
        }
        wg[ch * VEC + e] = a; wb[ch * VEC + e] = b;
      }
    }
  }
}

// column sums of the [W, N] partial slabs, accumulated into out0/out1 (zeroed by the launcher
// when beta == 0): grid = (column blocks of 64) x (row slices), 4 row groups per block reduced
// through LDS, one fp32 atomic per column per block
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ ws, int W, int N,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              int rows_per_slice) {
  __shared__ float red[2][4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int w0 = blockIdx.y * rows_per_slice, w1 = min(W, w0 + rows_per_slice);
  float a = 0.f, b = 0.f;
  if (n < N) {
    // eight rows of both slabs requested at once (one round trip for the usual 32-row slice; a loop of dependent
    // 4-byte loads made this 3 MB reduction an 11.8 us launch, 25 times per step), summed in row order as before
    for (int wb = w0 + g; wb < w1; wb += 32) {
      float av[8], bv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int w = wb + 4 * i;
        const int wc = w < w1 ? w : wb;
        av[i] = ws[(int64_t)wc * N + n];
        bv[i] = ws[(int64_t)(W + wc) * N + n];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (wb + 4 * i < w1) { a += av[i]; b += bv[i]; }
    }
  }
  red[0][g][c] = a; red[1][g][c] = b;
  __syncthreads();
  if (g == 0 && n < N) {
    a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (out0) atomicAdd(out0 + n, a);
    if (out1) atomicAdd(out1 + n, b);
  }
}

// ---- RoPE (standalone, in place) ------------------------------------------------------------
// thread = (b, head, l, pair i): a = x[i], b = x[i + dh/2]
template <typename T>
__device__ __forceinline__ float rl(float x) { return x; }
template <>
__device__ __forceinline__ float rl<bf16>(float x) { return vy_round_bf16(x); }

template <typename T>
__global__ void rope_kernel(T* __restrict__ x, int64_t sb, int64_t sh, int64_t sl,
                            const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                            int64_t pos0, int64_t B, int heads, int64_t L, int dh, int inverse) {
  const int half = dh >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = B * heads * L * half;
  if (idx >= total) return;
  const int i = (int)(idx % half);
  int64_t r = idx / half;
  const int64_t l = r % L; r /= L;
  const int hd = (int)(r % heads);
  const int64_t b = r / heads;
  T* p = x + b * sb + hd * sh + l * sl;
  const float c = rl<T>(cos_tab[(pos0 + l) * half + i]);
  float s = rl<T>(sin_tab[(pos0 + l) * half + i]);
  if (inverse) s = -s;
  const float a = VyT<T>::ld(p + i), bb = VyT<T>::ld(p + i + half);
  // reference: (q * cos) + (rotate_half(q) * sin), every op rounded to q.dtype
  // (VyomAI/layers/positional_embeddings.py:178-181)
  VyT<T>::st(p + i, rl<T>(a * c) + rl<T>(-bb * s));
  VyT<T>::st(p + i + half, rl<T>(bb * c) + rl<T>(a * s));
}

// q and k in ONE launch (the unfused QKV path of wide heads: Gemma dh = 256, SigLIP dh = 72): blockIdx.y picks
// the tensor; same arithmetic as rope_kernel
template <typename T>
__global__ void rope2_kernel(T* __restrict__ xq, int64_t q_sb, int64_t q_sh, int64_t q_sl, int hq,
                             T* __restrict__ xk, int64_t k_sb, int64_t k_sh, int64_t k_sl, int hk,
                             const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                             int64_t pos0, int64_t B, int64_t L, int dh) {
  const bool isk = blockIdx.y == 1;
  T* x = isk ? xk : xq;
  const int64_t sb = isk ? k_sb : q_sb, sh = isk ? k_sh : q_sh, sl = isk ? k_sl : q_sl;
  const int heads = isk ? hk : hq;
  const int half = dh >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * heads * L * half) return;
  const int i = (int)(idx % half);
  int64_t r = idx / half;
  const int64_t l = r % L; r /= L;
  const int hd = (int)(r % heads);
  const int64_t b = r / heads;
  T* p = x + b * sb + hd * sh + l * sl;
  const float c = rl<T>(cos_tab[(pos0 + l) * half + i]);
  const float s = rl<T>(sin_tab[(pos0 + l) * half + i]);
  const float a = VyT<T>::ld(p + i), bb = VyT<T>::ld(p + i + half);
  VyT<T>::st(p + i, rl<T>(a * c) + rl<T>(-bb * s));
  VyT<T>::st(p + i + half, rl<T>(bb * c) + rl<T>(a * s));
}

// ---- cast / transpose / adamw -----------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) dst[i + e] = (D)(float)src[i + e];
  }
}

template <typename T>
__global__ void transpose_kernel(const T* __restrict__ in, int64_t ldin, T* __restrict__ out, int64_t ldout,
                                 int R, int C) {
  __shared__ T tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = in[(int64_t)r * ldin + c];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) out[(int64_t)c * ldout + r] = tile[tx][i];
  }
}

// all the W^T copies of a model in ONE launch (the per-matrix launches are latency-bound: ~8 us for
// a 1-5 MB matrix, 50 of them per training step): block -> matrix by bisection of the tile prefix
template <typename T>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const vy_transpose_desc* __restrict__ descs, int n) {
  // 64 x 64 tile (tile0 / tiles_c count 64-wide tiles): 16-byte loads along the input rows, 16-byte
  // stores along the output rows, the transposition through LDS (row pitch 66 elements: the eight
  // column reads of a store hit different banks)
  constexpr int VEC = 16 / (int)sizeof(T);   // 8 (bf16) or 4 (fp32)
  constexpr int CPRW = 64 / VEC;              // 16-byte chunks per 64-element row
  __shared__ T tile[64][66];
  int lo = 0, hi = n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const vy_transpose_desc d = descs[lo];
  const int t = b - d.tile0;
  const int c0 = (t % d.tiles_c) * 64, r0 = (t / d.tiles_c) * 64;
  const T* in = (const T*)d.in;
  T* out = (T*)d.out;
  const bool vec_in = (d.ldin % VEC) == 0 && ((uintptr_t)in % 16) == 0;
  const bool vec_out = (d.ldout % VEC) == 0 && ((uintptr_t)out % 16) == 0;
  for (int c = threadIdx.x; c < 64 * CPRW; c += 256) {
    const int r = c / CPRW, cc = (c % CPRW) * VEC;
    const int gr = r0 + r, gc = c0 + cc;
    if (gr >= d.R) continue;
    if (vec_in && gc + VEC <= d.C) {
      typedef typename Chunk<T>::Raw Raw;
      const Raw raw = *reinterpret_cast<const Raw*>(in + (int64_t)gr * d.ldin + gc);
      const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int k = 0; k < VEC; ++k) tile[r][cc + k] = e[k];
    } else {
      for (int k = 0; k < VEC; ++k)
        if (gc + k < d.C) tile[r][cc + k] = in[(int64_t)gr * d.ldin + gc + k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * CPRW; c += 256) {
    const int oc = c / CPRW, rr = (c % CPRW) * VEC;   // output row = input column c0 + oc
    const int gc = c0 + oc, gr = r0 + rr;
    if (gc >= d.C) continue;
    if (vec_out && gr + VEC <= d.R) {
      typedef typename Chunk<T>::Raw Raw;
      Raw raw;
      T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
      for (int k = 0; k < VEC; ++k) e[k] = tile[rr + k][oc];
      *reinterpret_cast<Raw*>(out + (int64_t)gc * d.ldout + gr) = raw;
    } else {
      for (int k = 0; k < VEC; ++k)
        if (gr + k < d.R) out[(int64_t)gc * d.ldout + gr + k] = tile[rr + k][oc];
    }
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, bf16* __restrict__ pb, int64_t n, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                             const float* __restrict__ gscale_dev, const float* __restrict__ gate) {
  if (gate && *gate == 0.0f) return;       // a parameter no rank had a gradient for: left alone (vy_adamw_step_gated)
  if (gscale_dev) gscale *= *gscale_dev;   // e.g. the gradient-clipping coefficient, computed on the device
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      // the optimizer state is touched once per step (3.7 GB per step for 124 M parameters, under the backward pass on a side
      // stream): streamed both ways (nt), so that it does not take the Infinity Cache away from the backward kernels'
      // hand-offs.  The bf16 working copy keeps the default policy: the next forward reads it.
      f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + i));
      const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
      f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + i));
      f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + i));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = gv[e] * gscale;
        pv[e] *= (1.0f - lr * wd);
        mv[e] = b1 * mv[e] + (1.0f - b1) * gg;
        vv[e] = b2 * vv[e] + (1.0f - b2) * gg * gg;
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pv[e] -= (lr / bc1) * (mv[e] / denom);
      }
      __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p + i));
      __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m + i));
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
      if (pb) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)pv[e];
        *reinterpret_cast<bf16x4*>(pb + i) = o;
      }
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) {
        const float gg = g[i + e] * gscale;
        float pv = p[i + e] * (1.0f - lr * wd);
        const float mv = b1 * m[i + e] + (1.0f - b1) * gg;
        const float vv = b2 * v[i + e] + (1.0f - b2) * gg * gg;
        pv -= (lr / bc1) * (mv / (sqrtf(vv) / bc2_sqrt + eps));
        p[i + e] = pv; m[i + e] = mv; v[i + e] = vv;
        if (pb) pb[i + e] = (bf16)pv;
      }
    }
  }
}

// ---- activation backward (elementwise) ------------------------------------------------------
template <typename T, int ACT>
__global__ void act_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ pre, int64_t ldpre,
                               T* __restrict__ dx, int64_t lddx, int64_t M, int N) {
  constexpr int VEC = Chunk<T>::VEC;
  const int nch = N / VEC;
  const int64_t total = M * nch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nch;
    const int c = (int)(i - m * nch) * VEC;
    float a[VEC], b[VEC], o[VEC];
    Chunk<T>::load(dy + m * lddy + c, a);
    Chunk<T>::load(pre + m * ldpre + c, b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      // bf16: the derivative the GEMM epilogues use (vy_act_grad_fast: its error is below bf16 rounding; libm's erff made
      // this 75 MB pass a 94 us launch); fp32 keeps the exact form (the 1e-5 parity path)
      if constexpr (sizeof(T) == 2) o[e] = a[e] * vy_act_grad_fast<ACT>(b[e]);
      else o[e] = a[e] * vy_act_grad<ACT>(b[e]);
    }
    Chunk<T>::store(dx + m * lddx + c, o);
  }
}

// ---- softmax cross-entropy: one workgroup (4 waves) per row ------------------------------------
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? vy_wave_max(v) : vy_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const T* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ labels, int64_t ignore,
                                                       float* __restrict__ lse, float* __restrict__ loss_sum,
                                                       float* __restrict__ count, int V, int* __restrict__ err) {
  constexpr int VEC = Chunk<T>::VEC;
  __shared__ float red[4];
  const int64_t m = blockIdx.x;
  const int64_t label = labels[m];
  // a label outside [0, V) that is not ignore_index (e.g. -100 padding under another ignore_index) would be
  // an out-of-bounds read: the row is treated as ignored and the device error flag raised instead
  const bool oob = label != ignore && (label < 0 || label >= V);
  if (oob && err && threadIdx.x == 0) *err = 1;
  if (label == ignore || oob) { if (threadIdx.x == 0) lse[m] = 0.f; return; }
  const T* row = logits + m * ld;
  const int nch = (V + VEC - 1) / VEC;
  float mx = -INFINITY, sm = 0.f;
  for (int c = threadIdx.x; c < nch; c += blockDim.x) {
    float v[VEC];
    Chunk<T>::load(row + (int64_t)c * VEC, v);  // the padded tail of the row is readable
    float cm = -INFINITY;
#pragma unroll
    for (int e = 0; e < VEC; ++e) if (c * VEC + e < V) cm = fmaxf(cm, v[e]);
    const float nm = fmaxf(mx, cm);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) if (c * VEC + e < V) acc += __expf(v[e] - nm);
    sm = sm * __expf(mx - nm) + acc;
    mx = nm;
  }
  const float gmx = block_reduce(mx, red, true);
  const float gsm = block_reduce(sm * __expf(mx - gmx), red, false);
  if (threadIdx.x == 0) {
    const float l = gmx + __logf(gsm);
    lse[m] = l;
    atomicAdd(loss_sum, l - VyT<T>::ld(row + label));
    atomicAdd(count, 1.0f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(T* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ labels, int64_t ignore,
                                                       const float* __restrict__ lse, const float* __restrict__ gscale,
                                                       const float* __restrict__ count, int V) {
  constexpr int VEC = Chunk<T>::VEC;
  const int64_t m = blockIdx.x;
  const int64_t label = labels[m];
  T* row = logits + m * ld;
  const int nch = (V + VEC - 1) / VEC;
  const bool dead = label == ignore || label < 0 || label >= V;   // out-of-range labels: ignored rows (vy_xent_fwd flags them)
  const float sc = dead ? 0.f : (*gscale) / fmaxf(*count, 1.0f);
  const float l = lse[m];
  for (int c = threadIdx.x; c < nch; c += blockDim.x) {
    float v[VEC], o[VEC];
    Chunk<T>::load(row + (int64_t)c * VEC, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int col = c * VEC + e;
      float g = 0.f;
      if (!dead && col < V) g = (__expf(v[e] - l) - (col == label ? 1.f : 0.f)) * sc;
      o[e] = g;  // pad columns stay zero
    }
    Chunk<T>::store(row + (int64_t)c * VEC, o);
  }
}

// Fused forward + backward: the row (<= 65536 bf16 logits) lives in the registers of a 1024-thread
// workgroup between the two reductions and the gradient store, so the logits cross HBM exactly
// twice (one read, one write) instead of three reads and one write for vy_xent_fwd + vy_xent_bwd.
__global__ __launch_bounds__(1024) void xent_fused_kernel(bf16* __restrict__ logits, int64_t ld,
                                                          const int64_t* __restrict__ labels, int64_t ignore,
                                                          float* __restrict__ lse, float* __restrict__ loss_sum,
                                                          const float* __restrict__ count,
                                                          const float* __restrict__ gscale, int V,
                                                          int* __restrict__ err) {
  constexpr int CPT = 8;  // 16-byte chunks per thread
  __shared__ float red[16];
  const int tid = threadIdx.x;
  const int64_t m = blockIdx.x;
  const int64_t label = labels[m];
  bf16* row = logits + m * ld;
  const int nch = (V + 7) / 8;
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (bf16)0.f;
  const bool oob = label != ignore && (label < 0 || label >= V);   // see xent_fwd_kernel
  if (oob && err && tid == 0) *err = 1;
  if (label == ignore || oob) {
    for (int c = tid; c < nch; c += 1024) *reinterpret_cast<bf16x8*>(row + (int64_t)c * 8) = zero8;
    if (tid == 0) lse[m] = 0.f;
    return;
  }
  const float x_label = (float)row[label];  // read before any thread overwrites the row (barriers below)
  bf16x8 v[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * 1024;
    if (c < nch) v[i] = *reinterpret_cast<const bf16x8*>(row + (int64_t)c * 8);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * 1024;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c * 8 + e < V) mx = fmaxf(mx, (float)v[i][e]);
    }
  }
  const float gmx = block_reduce(mx, red, true);
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * 1024;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c * 8 + e < V) sm += __expf((float)v[i][e] - gmx);
    }
  }
  const float gsm = block_reduce(sm, red, false);
  const float l = gmx + __logf(gsm);
  if (tid == 0) {
    lse[m] = l;
    atomicAdd(loss_sum, l - x_label);
  }
  const float sc = (*gscale) / fmaxf(*count, 1.0f);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * 1024;
    if (c < nch) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = c * 8 + e;
        float g = 0.f;  // pad columns stay zero
        if (col < V) g = (__expf((float)v[i][e] - l) - (col == label ? 1.f : 0.f)) * sc;
        o[e] = (bf16)g;
      }
      *reinterpret_cast<bf16x8*>(row + (int64_t)c * 8) = o;
    }
  }
}

template <typename T>
int ln_fwd_dispatch(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y, int64_t ldy,
                    float* mean, float* rstd, int64_t M, int64_t N, float eps, hipStream_t st) {
  constexpr int VEC = Chunk<T>::VEC;
  if (N % VEC || ldx % VEC || ldy % VEC) VY_FAIL(VY_ERR_ARG, "vy_layernorm_fwd: N/ld must be multiples of %d", VEC);
  const int nch = (int)(N / VEC);
  const dim3 grid((unsigned)vy_cdiv(M, 4)), block(256);
#define LN_GO(CH)                                                                                       \
  hipLaunchKernelGGL((layernorm_fwd_kernel<T, CH>), grid, block, 0, st, (const T*)x, ldx, (const T*)gamma, \
                     (const T*)beta, (T*)y, ldy, mean, rstd, M, (int)N, eps)
  if (nch <= 64) LN_GO(1);
  else if (nch <= 128) LN_GO(2);
  else if (nch <= 256) LN_GO(4);
  else if (nch <= 512) LN_GO(8);
  else if (nch <= 1024) LN_GO(16);
  else VY_FAIL(VY_ERR_UNSUPPORTED, "vy_layernorm_fwd: N=%ld too wide", (long)N);
#undef LN_GO
  VY_CHECK_LAUNCH("vy_layernorm_fwd");
  return VY_OK;
}

template <typename T>
int ln_bwd_dispatch(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma,
                    const float* mean, const float* rstd, void* dx, int64_t lddx, float* dgamma,
                    float* dbeta, float beta, float* ws, int64_t M, int64_t N, hipStream_t st) {
  constexpr int VEC = Chunk<T>::VEC;
  if (N % VEC || ldx % VEC || lddy % VEC || lddx % VEC) VY_FAIL(VY_ERR_ARG, "vy_layernorm_bwd: N/ld must be multiples of %d", VEC);
  const int nch = (int)(N / VEC);
  const int WB = (int)vy_layernorm_bwd_ws_rows(M);  // blocks = slab rows
  const int W = WB * 4;
  const dim3 grid((unsigned)WB), block(256);
#define LN_GO(CH)                                                                                        \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, CH>), grid, block, 0, st, (const T*)dy, lddy, (const T*)x, ldx, \
                     (const T*)gamma, mean, rstd, (T*)dx, lddx, ws, M, (int)N, W)
  if (nch <= 64) LN_GO(1);
  else if (nch <= 128) LN_GO(2);
  else if (nch <= 256) LN_GO(4);
  else VY_FAIL(VY_ERR_UNSUPPORTED, "vy_layernorm_bwd: N=%ld too wide", (long)N);
#undef LN_GO
  VY_CHECK_LAUNCH("vy_layernorm_bwd");
  if (beta == 0.f) {
    if (dgamma && hipMemsetAsync(dgamma, 0, N * sizeof(float), st) != hipSuccess) VY_FAIL(VY_ERR_LAUNCH, "vy_layernorm_bwd: memset failed");
    if (dbeta && hipMemsetAsync(dbeta, 0, N * sizeof(float), st) != hipSuccess) VY_FAIL(VY_ERR_LAUNCH, "vy_layernorm_bwd: memset failed");
  } else if (beta != 1.f) {
    VY_FAIL(VY_ERR_ARG, "vy_layernorm_bwd: beta must be 0 or 1");
  }
  const int slices = WB >= 64 ? 16 : 1;
  const int rps = (int)vy_cdiv(WB, slices);
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)vy_cdiv(N, 64), (unsigned)slices), dim3(256), 0, st, ws, WB,
                     (int)N, dgamma, dbeta, rps);
  VY_CHECK_LAUNCH("vy_layernorm_bwd(colsum)");
  return VY_OK;
}

}  // namespace

extern "C" int vy_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y,
                                int64_t ldy, float* mean, float* rstd, int64_t M, int64_t N, float eps,
                                int dtype, void* stream) {
  if (!x || !gamma || !beta || !y || M <= 0 || N <= 0) VY_FAIL(VY_ERR_ARG, "vy_layernorm_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16) return ln_fwd_dispatch<bf16>(x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, st);
  if (dtype == VY_F32) return ln_fwd_dispatch<float>(x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, st);
  VY_FAIL(VY_ERR_ARG, "vy_layernorm_fwd: bad dtype %d", dtype);
}

// number of workgroups (= partial slab rows) of the LayerNorm backward: 4 waves each, 8 waves per
// CU in flight at M = 16384 -- enough outstanding 16-byte loads to stream dy/x at the HBM rate
extern "C" int vy_rmsnorm_fwd(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t M, int64_t N,
                              float eps, float w_offset, int dtype, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0) VY_FAIL(VY_ERR_ARG, "vy_rmsnorm_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (N % vec || ldx % vec || ldy % vec) VY_FAIL(VY_ERR_ARG, "vy_rmsnorm_fwd: N/ld must be multiples of %d", vec);
  const int nch = (int)(N / vec);
  const dim3 grid((unsigned)vy_cdiv(M, 4)), block(256);
#define RMS_GO(T, CH) hipLaunchKernelGGL((rmsnorm_fwd_kernel<T, CH>), grid, block, 0, st, (const T*)x, ldx, (const T*)w, (T*)y, ldy, M, (int)N, eps, w_offset)
#define RMS_DISPATCH(T)                                  \
  if (nch <= 64) RMS_GO(T, 1);                           \
  else if (nch <= 128) RMS_GO(T, 2);                     \
  else if (nch <= 256) RMS_GO(T, 4);                     \
  else if (nch <= 512) RMS_GO(T, 8);                     \
  else if (nch <= 1024) RMS_GO(T, 16);                   \
  else VY_FAIL(VY_ERR_UNSUPPORTED, "vy_rmsnorm_fwd: N=%ld too wide", (long)N)
  if (dtype == VY_BF16) { RMS_DISPATCH(bf16); }
  else if (dtype == VY_F32) { RMS_DISPATCH(float); }
  else VY_FAIL(VY_ERR_ARG, "vy_rmsnorm_fwd: bad dtype %d", dtype);
#undef RMS_DISPATCH
#undef RMS_GO
  VY_CHECK_LAUNCH("vy_rmsnorm_fwd");
  return VY_OK;
}

extern "C" int vy_gated_act_fwd(const void* gate_up, int64_t ldg, void* out, int64_t ldo, int64_t M, int64_t I, int act,
                                int dtype, void* stream) {
  if (!gate_up || !out || M <= 0 || I <= 0) VY_FAIL(VY_ERR_ARG, "vy_gated_act_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (I % vec || ldg % vec || ldo % vec) VY_FAIL(VY_ERR_ARG, "vy_gated_act_fwd: I/ld must be multiples of %d", vec);
  const int64_t want = vy_cdiv(M * (I / vec), 256);
  const dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
#define GA_GO(T, A) hipLaunchKernelGGL((gated_act_kernel<T, A>), grid, block, 0, st, (const T*)gate_up, ldg, (T*)out, ldo, M, (int)I)
  if (dtype == VY_BF16 && act == VY_ACT_GELU_TANH) GA_GO(bf16, VY_ACT_GELU_TANH);
  else if (dtype == VY_BF16 && act == VY_ACT_GELU_ERF) GA_GO(bf16, VY_ACT_GELU_ERF);
  else if (dtype == VY_F32 && act == VY_ACT_GELU_TANH) GA_GO(float, VY_ACT_GELU_TANH);
  else if (dtype == VY_F32 && act == VY_ACT_GELU_ERF) GA_GO(float, VY_ACT_GELU_ERF);
  else VY_FAIL(VY_ERR_ARG, "vy_gated_act_fwd: unsupported act %d / dtype %d", act, dtype);
#undef GA_GO
  VY_CHECK_LAUNCH("vy_gated_act_fwd");
  return VY_OK;
}

extern "C" int64_t vy_layernorm_bwd_ws_rows(int64_t M) {
  static const int64_t cap = [] { const char* e = getenv("VY_LNBWD_WB"); return e ? (int64_t)atoi(e) : (int64_t)512; }();   // measured at M = 16384, N = 768: 256: 27.8 us, 512: 24.3, 1024: 27.6, 2048: 36.3
  const int64_t b = (M + 3) / 4;  // one wave per row at least
  return b < 1 ? 1 : (b > cap ? cap : b);
}

extern "C" int vy_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma,
                                const float* mean, const float* rstd, void* dx, int64_t lddx, float* dgamma,
                                float* dbeta, float beta, float* ws, int64_t M, int64_t N, int dtype,
                                void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !ws || M <= 0 || N <= 0)
    VY_FAIL(VY_ERR_ARG, "vy_layernorm_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16) return ln_bwd_dispatch<bf16>(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dgamma, dbeta, beta, ws, M, N, st);
  if (dtype == VY_F32) return ln_bwd_dispatch<float>(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dgamma, dbeta, beta, ws, M, N, st);
  VY_FAIL(VY_ERR_ARG, "vy_layernorm_bwd: bad dtype %d", dtype);
}

// internal (vy_qkv_rope_fwd's unfused path): RoPE on q (hq heads) and k (hk heads) in one launch
int vy_rope_qk(void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl, int hq, void* k, int64_t k_sb, int64_t k_sh,
               int64_t k_sl, int hk, const float* cos_tab, const float* sin_tab, int64_t pos0, int64_t B, int64_t L, int dh,
               int dtype, hipStream_t st) {
  if (!q || !k || !cos_tab || !sin_tab || (dh & 1)) VY_FAIL(VY_ERR_ARG, "vy_rope_qk: bad arguments");
  const int64_t total = B * (hq > hk ? hq : hk) * L * (dh / 2);
  const dim3 grid((unsigned)vy_cdiv(total, 256), 2), block(256);
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(rope2_kernel<bf16>, grid, block, 0, st, (bf16*)q, q_sb, q_sh, q_sl, hq, (bf16*)k, k_sb, k_sh, k_sl, hk,
                       cos_tab, sin_tab, pos0, B, L, dh);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(rope2_kernel<float>, grid, block, 0, st, (float*)q, q_sb, q_sh, q_sl, hq, (float*)k, k_sb, k_sh, k_sl,
                       hk, cos_tab, sin_tab, pos0, B, L, dh);
  else VY_FAIL(VY_ERR_ARG, "vy_rope_qk: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_rope_qk");
  return VY_OK;
}

extern "C" int vy_rope_fwd(void* x, int64_t sb, int64_t sh, int64_t sl, const float* cos_tab,
                           const float* sin_tab, int64_t pos0, int64_t B, int heads, int64_t L, int dh,
                           int inverse, int dtype, void* stream) {
  if (!x || !cos_tab || !sin_tab || dh % 2 || B <= 0 || heads <= 0 || L <= 0)
    VY_FAIL(VY_ERR_ARG, "vy_rope_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = B * heads * L * (dh / 2);
  const dim3 grid((unsigned)vy_cdiv(total, 256)), block(256);
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(rope_kernel<bf16>, grid, block, 0, st, (bf16*)x, sb, sh, sl, cos_tab, sin_tab, pos0, B, heads, L, dh, inverse);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(rope_kernel<float>, grid, block, 0, st, (float*)x, sb, sh, sl, cos_tab, sin_tab, pos0, B, heads, L, dh, inverse);
  else VY_FAIL(VY_ERR_ARG, "vy_rope_fwd: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_rope_fwd");
  return VY_OK;
}

extern "C" int vy_act_bwd(const void* dy, int64_t lddy, const void* pre, int64_t ldpre, void* dx, int64_t lddx,
                          int64_t M, int64_t N, int act, int dtype, void* stream) {
  if (!dy || !pre || !dx || M <= 0 || N <= 0) VY_FAIL(VY_ERR_ARG, "vy_act_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (N % vec || lddy % vec || ldpre % vec || lddx % vec) VY_FAIL(VY_ERR_ARG, "vy_act_bwd: N/ld must be multiples of %d", vec);
  const int64_t want = vy_cdiv(M * (N / vec), 256);
  const dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
#define AB_GO(T, A) hipLaunchKernelGGL((act_bwd_kernel<T, A>), grid, block, 0, st, (const T*)dy, lddy, (const T*)pre, ldpre, (T*)dx, lddx, M, (int)N)
  if (dtype == VY_BF16 && act == VY_ACT_GELU_ERF) AB_GO(bf16, VY_ACT_GELU_ERF);
  else if (dtype == VY_BF16 && act == VY_ACT_GELU_TANH) AB_GO(bf16, VY_ACT_GELU_TANH);
  else if (dtype == VY_F32 && act == VY_ACT_GELU_ERF) AB_GO(float, VY_ACT_GELU_ERF);
  else if (dtype == VY_F32 && act == VY_ACT_GELU_TANH) AB_GO(float, VY_ACT_GELU_TANH);
  else VY_FAIL(VY_ERR_ARG, "vy_act_bwd: unsupported act %d / dtype %d", act, dtype);
#undef AB_GO
  VY_CHECK_LAUNCH("vy_act_bwd");
  return VY_OK;
}

extern "C" int vy_xent_fwd(const void* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, float* lse,
                           float* loss_sum, float* count, int64_t M, int64_t V, int32_t* err_flag, int dtype,
                           void* stream) {
  if (!logits || !labels || !lse || !loss_sum || !count || M <= 0 || V <= 0) VY_FAIL(VY_ERR_ARG, "vy_xent_fwd: bad arguments");
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (ld % vec || ld < vy_cdiv(V, vec) * vec) VY_FAIL(VY_ERR_ARG, "vy_xent_fwd: row stride must be a multiple of %d and cover the padded row", vec);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16) hipLaunchKernelGGL(xent_fwd_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, st, (const bf16*)logits, ld, labels, ignore_index, lse, loss_sum, count, (int)V, err_flag);
  else if (dtype == VY_F32) hipLaunchKernelGGL(xent_fwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, (const float*)logits, ld, labels, ignore_index, lse, loss_sum, count, (int)V, err_flag);
  else VY_FAIL(VY_ERR_ARG, "vy_xent_fwd: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_xent_fwd");
  return VY_OK;
}

extern "C" int vy_xent_bwd(void* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, const float* lse,
                           const float* gscale, const float* count, int64_t M, int64_t V, int dtype, void* stream) {
  if (!logits || !labels || !lse || !gscale || !count || M <= 0 || V <= 0) VY_FAIL(VY_ERR_ARG, "vy_xent_bwd: bad arguments");
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (ld % vec || ld < vy_cdiv(V, vec) * vec) VY_FAIL(VY_ERR_ARG, "vy_xent_bwd: row stride must be a multiple of %d and cover the padded row", vec);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16) hipLaunchKernelGGL(xent_bwd_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, st, (bf16*)logits, ld, labels, ignore_index, lse, gscale, count, (int)V);
  else if (dtype == VY_F32) hipLaunchKernelGGL(xent_bwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, (float*)logits, ld, labels, ignore_index, lse, gscale, count, (int)V);
  else VY_FAIL(VY_ERR_ARG, "vy_xent_bwd: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_xent_bwd");
  return VY_OK;
}

extern "C" int vy_xent_fused(void* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, float* lse,
                             float* loss_sum, const float* count, const float* gscale, int64_t M, int64_t V,
                             int32_t* err_flag, int dtype, void* stream) {
  if (!logits || !labels || !lse || !loss_sum || !count || !gscale || M <= 0 || V <= 0) VY_FAIL(VY_ERR_ARG, "vy_xent_fused: bad arguments");
  if (dtype != VY_BF16) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_xent_fused: bf16 only (use vy_xent_fwd + vy_xent_bwd)");
  if (V > 65536) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_xent_fused: V=%ld exceeds the 65536 columns a workgroup keeps in registers (use vy_xent_fwd + vy_xent_bwd)", (long)V);
  if (ld % 8 || ld < vy_cdiv(V, 8) * 8 || (uintptr_t)logits % 16) VY_FAIL(VY_ERR_ARG, "vy_xent_fused: rows must be 16-byte aligned and cover the padded width");
  hipLaunchKernelGGL(xent_fused_kernel, dim3((unsigned)M), dim3(1024), 0, (hipStream_t)stream, (bf16*)logits, ld, labels,
                     ignore_index, lse, loss_sum, count, gscale, (int)V, err_flag);
  VY_CHECK_LAUNCH("vy_xent_fused");
  return VY_OK;
}

extern "C" int vy_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream) {
  if (!src || !dst || n < 0) VY_FAIL(VY_ERR_ARG, "vy_cast: bad arguments");
  if (n == 0) return VY_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = vy_cdiv(n, 1024);
  const dim3 grid((unsigned)(want < 4096 ? want : 4096)), block(256);
  if (src_dtype == VY_F32 && dst_dtype == VY_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, st, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == VY_BF16 && dst_dtype == VY_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, st, (const bf16*)src, (float*)dst, n);
  else VY_FAIL(VY_ERR_ARG, "vy_cast: unsupported %d -> %d", src_dtype, dst_dtype);
  VY_CHECK_LAUNCH("vy_cast");
  return VY_OK;
}

// ---- token embedding: gather and its backward (reference: nn.Embedding in every model) ---------
// fwd: out[m,:] = table[ids[m],:]; one wave per row, 16-byte chunks.  An id outside [0,V) is a
// caller error the host cannot see without a sync: the row is written as zeros and `err` (device
// int, optional) is set, the analogue of aten's device-side assert without killing the context.
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const T* __restrict__ table, int64_t ldt,
                                                            const int64_t* __restrict__ ids, T* __restrict__ out,
                                                            int64_t ldo, int64_t M, int d, int64_t V, int* err) {
  constexpr int VEC = Chunk<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t id = ids[m];
  const bool bad = id < 0 || id >= V;
  if (bad && err && lane == 0) *err = 1;
  for (int c = lane; c * VEC < d; c += 64) {
    float v[VEC];
    if (bad) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    } else {
      Chunk<T>::load(table + id * ldt + (int64_t)c * VEC, v);
    }
    Chunk<T>::store(out + m * ldo + (int64_t)c * VEC, v);
  }
}

// bwd: dW[ids[m],:] += dOut[m,:] (fp32 atomics; rows of repeated ids collide, order not fixed);
// the padding row receives no gradient, as in nn.Embedding(padding_idx=...)
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const T* __restrict__ dout, int64_t lddo,
                                                            const int64_t* __restrict__ ids, float* __restrict__ dw,
                                                            int64_t lddw, int64_t padding_idx, int64_t M, int d, int64_t V) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t id = ids[m];
  if (id == padding_idx || id < 0 || id >= V) return;
  // lane-contiguous: one atomic instruction of the wave covers 64 consecutive floats (whole 128-byte
  // lines at the L2 atomic units), not 64 scattered 32-byte pieces
  float* row = dw + id * lddw;
  const T* src = dout + m * lddo;
  for (int c = lane; c < d; c += 64) atomicAdd(row + c, VyT<T>::ld(src + c));
}

extern "C" int vy_embedding_fwd(const void* table, int64_t ldt, const int64_t* ids, void* out, int64_t ldo,
                                int64_t M, int64_t d, int64_t V, int32_t* err_flag, int dtype, void* stream) {
  if (!table || !ids || !out || M <= 0 || d <= 0 || V <= 0) VY_FAIL(VY_ERR_ARG, "vy_embedding_fwd: bad arguments");
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (d % vec || ldt % vec || ldo % vec) VY_FAIL(VY_ERR_ARG, "vy_embedding_fwd: width and strides must be multiples of %d", vec);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)vy_cdiv(M, 4)), block(256);
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(embedding_fwd_kernel<bf16>, grid, block, 0, st, (const bf16*)table, ldt, ids, (bf16*)out, ldo, M, (int)d, V, err_flag);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(embedding_fwd_kernel<float>, grid, block, 0, st, (const float*)table, ldt, ids, (float*)out, ldo, M, (int)d, V, err_flag);
  else VY_FAIL(VY_ERR_ARG, "vy_embedding_fwd: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_embedding_fwd");
  return VY_OK;
}

extern "C" int vy_embedding_bwd(const void* dout, int64_t lddo, const int64_t* ids, float* dw, int64_t lddw,
                                int64_t padding_idx, int64_t M, int64_t d, int64_t V, int dtype, void* stream) {
  if (!dout || !ids || !dw || M <= 0 || d <= 0 || V <= 0) VY_FAIL(VY_ERR_ARG, "vy_embedding_bwd: bad arguments");
  const int vec = dtype == VY_BF16 ? 8 : 4;
  if (d % vec || lddo % vec) VY_FAIL(VY_ERR_ARG, "vy_embedding_bwd: width and strides must be multiples of %d", vec);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)vy_cdiv(M, 4)), block(256);
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(embedding_bwd_kernel<bf16>, grid, block, 0, st, (const bf16*)dout, lddo, ids, dw, lddw, padding_idx, M, (int)d, V);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(embedding_bwd_kernel<float>, grid, block, 0, st, (const float*)dout, lddo, ids, dw, lddw, padding_idx, M, (int)d, V);
  else VY_FAIL(VY_ERR_ARG, "vy_embedding_bwd: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_embedding_bwd");
  return VY_OK;
}

extern "C" int vy_transpose(const void* in, int64_t ldin, void* out, int64_t ldout, int64_t R, int64_t C,
                            int dtype, void* stream) {
  if (!in || !out || R <= 0 || C <= 0) VY_FAIL(VY_ERR_ARG, "vy_transpose: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)vy_cdiv(C, 32), (unsigned)vy_cdiv(R, 32)), block(256);
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16>, grid, block, 0, st, (const bf16*)in, ldin, (bf16*)out, ldout, (int)R, (int)C);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, st, (const float*)in, ldin, (float*)out, ldout, (int)R, (int)C);
  else VY_FAIL(VY_ERR_ARG, "vy_transpose: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_transpose");
  return VY_OK;
}

extern "C" int vy_transpose_batched(const vy_transpose_desc* descs_dev, int32_t n, int32_t total_tiles, int dtype,
                                    void* stream) {
  if (!descs_dev || n <= 0 || total_tiles <= 0) VY_FAIL(VY_ERR_ARG, "vy_transpose_batched: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16)
    hipLaunchKernelGGL(transpose_batched_kernel<bf16>, dim3((unsigned)total_tiles), dim3(256), 0, st, descs_dev, (int)n);
  else if (dtype == VY_F32)
    hipLaunchKernelGGL(transpose_batched_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, st, descs_dev, (int)n);
  else VY_FAIL(VY_ERR_ARG, "vy_transpose_batched: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_transpose_batched");
  return VY_OK;
}

extern "C" int vy_adamw_step_gated(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                   float grad_scale, const float* grad_scale_dev, const float* gate, void* stream);

extern "C" int vy_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             float grad_scale, const float* grad_scale_dev, void* stream) {
  return vy_adamw_step_gated(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                             grad_scale_dev, nullptr, stream);
}

extern "C" int vy_adamw_step_gated(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                   float grad_scale, const float* grad_scale_dev, const float* gate, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) VY_FAIL(VY_ERR_ARG, "vy_adamw_step: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  const int64_t want = vy_cdiv(n, 1024);
  const dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
  hipLaunchKernelGGL(adamw_kernel, grid, block, 0, st, p, g, m, v, (bf16*)p_bf16, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2s, grad_scale, grad_scale_dev, gate);
  VY_CHECK_LAUNCH("vy_adamw_step");
  return VY_OK;
}

// ---- sum of squares of an fp32 arena (global gradient norm for clip_grad_norm_) ----------------
// Two launches, no atomics: run-to-run identical.  ws: >= 1024 floats.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
  __shared__ float red[4];
  float acc = 0.f;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) acc += x[i + e] * x[i + e];
    }
  }
  acc = vy_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ ws, int nparts, float* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)ws[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(red[0] + red[1] + red[2] + red[3]);
}
extern "C" int vy_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream) {
  if (!x || !out || !ws || n <= 0) VY_FAIL(VY_ERR_ARG, "vy_sumsq: bad arguments");
  if ((uintptr_t)x % 16) VY_FAIL(VY_ERR_ARG, "vy_sumsq: x must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = vy_cdiv(n, 1024 * 8);
  const int nparts = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nparts), dim3(256), 0, st, x, n, ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nparts, out);
  VY_CHECK_LAUNCH("vy_sumsq");
  return VY_OK;
}

// ---- measured ceilings for the roofline report (not part of include/vyom_hip.h) ----------------------------
// The matrix pipe alone: 8 waves per CU, six independent mfma_f32_32x32x16_bf16 chains per wave on lane-dependent
// (non-trivial) operands, no memory traffic.  FLOPs = workgroups * 8 * iters * 6 * 2 * 32 * 32 * 16.
__global__ __launch_bounds__(512) void mfma_peak_kernel(int iters, float* __restrict__ sink) {
  bf16x8 a, b;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (bf16)(0.001f * (float)(((lane * 7 + e * 13 + blockIdx.x) % 97) - 48));
    b[e] = (bf16)(0.002f * (float)(((lane * 11 + e * 5 + threadIdx.x) % 89) - 44));
  }
  f32x16 acc[6];
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 6; ++c) s += acc[c][0] + acc[c][7];
  if (s == 12345.678f) sink[0] = s;
}
extern "C" int vy_debug_mfma_peak(int workgroups, int iters, float* sink, void* stream) {
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, iters, sink);
  VY_CHECK_LAUNCH("vy_debug_mfma_peak");
  return VY_OK;
}
// The HBM ceiling probe: a copy (read n bytes + write n bytes) with eight 16-byte loads in flight per lane before the
// first store, a workgroup walking contiguous 32 KiB pieces (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; the
// one-load-per-iteration loop this replaces reached 4.6-4.7).  MODE bit 0: non-temporal loads, bit 1: non-temporal stores.
template <int MODE>
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
  constexpr int U = 8;
  const int64_t piece = 256 * U;
  const int64_t npieces = n16 / piece;
  for (int64_t p = blockIdx.x; p < npieces; p += gridDim.x) {
    const f32x4* s = src + p * piece + threadIdx.x;
    f32x4* d = dst + p * piece + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = (MODE & 1) ? __builtin_nontemporal_load(s + j * 256) : s[j * 256];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (MODE & 2) __builtin_nontemporal_store(v[j], d + j * 256);
      else d[j * 256] = v[j];
    }
  }
  for (int64_t i = npieces * piece + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int vy_debug_copy(const void* src, void* dst, int64_t bytes, void* stream) {
  if (!src || !dst || bytes % 16) VY_FAIL(VY_ERR_ARG, "vy_debug_copy: bad arguments");
  static const int mode = [] { const char* e = getenv("VY_COPY_MODE"); return e ? atoi(e) : 3; }();
  static const int wgs = [] { const char* e = getenv("VY_COPY_WGS"); return e ? atoi(e) : 256 * 8; }();
  const dim3 g(wgs), b(256);
  hipStream_t st = (hipStream_t)stream;
  const f32x4* s = (const f32x4*)src; f32x4* d = (f32x4*)dst; const int64_t n16 = bytes / 16;
  switch (mode & 3) {
    case 0: hipLaunchKernelGGL(copy16_kernel<0>, g, b, 0, st, s, d, n16); break;
    case 1: hipLaunchKernelGGL(copy16_kernel<1>, g, b, 0, st, s, d, n16); break;
    case 2: hipLaunchKernelGGL(copy16_kernel<2>, g, b, 0, st, s, d, n16); break;
    default: hipLaunchKernelGGL(copy16_kernel<3>, g, b, 0, st, s, d, n16); break;
  }
  VY_CHECK_LAUNCH("vy_debug_copy");
  return VY_OK;
}

// ---- dropout as its own pass: y = x * keep(seed, offset, row, column) / (1 - p) ------------------
// The forward applies the mask inside the GEMM epilogue (vy_linear_dropout_fwd); backward applies the SAME
// mask to the incoming gradient here before the dgrad / wgrad GEMMs (their operands go to LDS by DMA and
// cannot be transformed on the way).  With x = ones this is the mask itself (tests).
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                      int64_t M, int N, VyDrop d) {
  const int nch = (N + 7) / 8;
  const int64_t total = M * nch;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = c / nch;
    const int ch = (int)(c - m * nch);
    uint32_t lots[4];
    vy_drop_lots(d, m, ch, lots);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = ch * 8 + e;
      if (n < N) {
        const float v = VyT<T>::ld(x + m * ldx + n);
        VyT<T>::st(y + m * ldy + n, vy_drop_keep(d, lots, e) ? v * d.scale : 0.f);
      }
    }
  }
}
extern "C" int vy_dropout(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t N, float p_drop,
                          uint64_t seed, uint64_t offset, int dtype, void* stream) {
  if (!x || !y || M <= 0 || N <= 0 || ldx < N || ldy < N) VY_FAIL(VY_ERR_ARG, "vy_dropout: bad arguments");
  if (!(p_drop >= 0.f && p_drop <= 1.f)) VY_FAIL(VY_ERR_ARG, "vy_dropout: p=%g outside [0, 1]", (double)p_drop);
  if (N > INT32_MAX) VY_FAIL(VY_ERR_ARG, "vy_dropout: N too large");
  hipStream_t st = (hipStream_t)stream;
  VyDrop d = vy_make_drop(p_drop, seed, offset);
  if (d.thr == 0) d.scale = 1.0f;   // p == 0: plain copy
  const int64_t want = vy_cdiv(M * vy_cdiv(N, 8), 256);
  const dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
  if (dtype == VY_BF16) hipLaunchKernelGGL(dropout_kernel<bf16>, grid, block, 0, st, (const bf16*)x, ldx, (bf16*)y, ldy, M, (int)N, d);
  else if (dtype == VY_F32) hipLaunchKernelGGL(dropout_kernel<float>, grid, block, 0, st, (const float*)x, ldx, (float*)y, ldy, M, (int)N, d);
  else VY_FAIL(VY_ERR_ARG, "vy_dropout: bad dtype %d", dtype);
  VY_CHECK_LAUNCH("vy_dropout");
  return VY_OK;
}
