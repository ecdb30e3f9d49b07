// Decode-step kernels (bf16, <= 32 sequences, one new token each): the projections, the split-K finish +
// LayerNorm, and single-query attention of reference models/decoder.py:430-514 (generate with a KV cache),
// written for the one thing that bounds a decode step of a d = 768 model on this chip: the LENGTH OF THE
// DEPENDENT CHAIN inside each launch.
//
// Measured (tools/bench_decode_links.py, tools/probe/cold_chain_probe.hip, hipGraph chains): a launch that
// returns at once costs 1.6 us as a chain link; one that streams 24 KiB per workgroup of never-touched memory,
// reduces through LDS and stores costs 2.7 us; the general M <= 32 GEMM kernels of vy_gemm.hip cost 4.3-6.7 us on
// the same shapes.  Their ISA shows why: kernel arguments fetched in three dependent stages, the epilogue operand
// prefetch retired with vmcnt(0) at a loop header BEFORE the weight loads are issued, a fragment re-loaded and
// waited for in the middle of the MFMA chain, 64-bit divisions in the scatter addressing.  Each is one more
// memory round trip (0.3-1 us) in a kernel that is nothing but round trips.  The kernels here have
//   * no loops around memory instructions (the k-steps of a wave are a compile-time count: every load of the
//     launch is issued before the first wait, and hipcc can count its waits instead of draining at a loop header),
//   * one small argument block, no debug knobs, no divisions, the decode case only (one token per row),
//   * wave reductions on the DPP / permlane path instead of ds_bpermute.
// Numerics are those of the general kernels (same k order per wave, same combine order, same rounding points), so
// eager / graph / prefill comparisons in the tests hold unchanged.
#include "vy_common.h"
#include <stdlib.h>
#include <float.h>

namespace {

enum { DEC_STORE = 0, DEC_PART = 1, DEC_QKV = 2 };

// measurement aid (vy_debug_set_decode_stamps, tools/decode_timeline.py): lane 0 of every workgroup's wave 0 writes
// the 100 MHz wall clock (s_memrealtime) at five points of the kernel into stamps[launch][workgroup][8]
struct DecDbg { unsigned long long* buf; int slot; int max_wg; };
#define DEC_STAMP(i)                                                                                          \
  if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); dbg_t[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); }
#define DEC_STAMP_FLUSH()                                                                                     \
  if constexpr (DBG) {                                                                                        \
    if (threadIdx.x == 0 && (int)(blockIdx.x + gridDim.x * blockIdx.y) < dbg.max_wg) {                        \
      for (int i_ = 0; i_ < 8; ++i_)                                                                          \
        dbg.buf[((long long)dbg.slot * dbg.max_wg + blockIdx.x + gridDim.x * blockIdx.y) * 8 + i_] = dbg_t[i_]; \
    }                                                                                                         \
  }

// 16 bytes of a stream too large to stay on chip until it is read again: the non-temporal policy
// (`global_load_dwordx4 ... nt`).  Used for the K/V rows of a LARGE cache only (dec_attn_kernel<..., NT>): at B = 32 the
// caches are 680 MB per step, and streamed with the default policy they push the 170 MB of weights out of the
// Infinity Cache every step (step 514.5 -> 501 us with nt); a small cache (configs[4]: 5.5 MB) is served on chip from one
// step to the next and must keep the default policy (+5 % per token with nt).  The weight tiles keep the default policy
// (nt on them alone: +1 %).
__device__ __forceinline__ bf16x8 dec_load_stream(const bf16* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
}

struct DecGemmArgs {
  const bf16* X; const bf16* W; const bf16* bias;
  int ldx, ldw, M, N;
  bf16* y; int ldy;                    // DEC_STORE: y[M][ldy] = act(x W^T + b)
  float* part;                         // DEC_PART: part[gridDim.y][32][N] (fp32, no bias)
  bf16* q; bf16* k; bf16* v;           // DEC_QKV: q[M][nq]; k / v = cache base (already at the host position)
  long long c_sb, c_sh, c_sl;          //   element strides of the caches (batch, head, token)
  const float* cos_tab; const float* sin_tab; const int* pos_dev;
  int pos0, nq, nkv, rope;
  int wt;                              // write-through stores
  const bf16* residual; int ldr;       // DEC_STORE: y = bf16(act(x W^T + b) + residual)
  const bf16* ln_g; const bf16* ln_b; float ln_eps;   // LNP kernels: the rows of X are LayerNorm'ed on the way in
};

// sum over the 64 lanes, result in every lane: four DPP rotations inside each row of 16 lanes, then the two
// half-swaps (v_permlane16_swap, v_permlane32_swap) -- six VALU instructions, no LDS round trips
__device__ __forceinline__ float dec_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
__device__ __forceinline__ float dec_wave_sum(float v) {
  v = dec_row16_sum(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
  const unsigned w = __builtin_bit_cast(unsigned, v);
  auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
}
// sum over the four 16-lane rows of the wave at each position inside a row (lanes l, l ^ 16, l ^ 32, l ^ 48), in every
// lane: the two half-swaps of dec_wave_sum without the in-row rotations
__device__ __forceinline__ float dec_rows_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
  const unsigned w = __builtin_bit_cast(unsigned, v);
  auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
}
__device__ __forceinline__ float dec_row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ float dec_wave_max(float v) {
  v = dec_row16_max(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
  const unsigned w = __builtin_bit_cast(unsigned, v);
  auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
}

// stores of the step's small outputs.  wt: write-through (sc0 sc1) -- the line does not stay dirty in this XCD's L2,
// so the write-back at the end of the kernel (every launch boundary releases at agent scope) finds nothing to do
template <typename V>
__device__ __forceinline__ void dec_store(void* p, const V& v, int wt) {
  if (wt) {
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<V*>(p) = v;
  }
}

// the value of lane ^ 32 (v_permlane32_swap instead of a ds_bpermute round trip); upper = this lane is >= 32
__device__ __forceinline__ float dec_swap32(float v, bool upper) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // s[0] = {lo, lo}, s[1] = {hi, hi}
  return __builtin_bit_cast(float, upper ? (unsigned)s[0] : (unsigned)s[1]);
}

// ------------------------------------------------------------------------------------------
// 16 output columns x 32 rows per workgroup, the K range of the workgroup (4 * U k-steps of 32) split over the
// 4 waves: wave w takes k-steps w, w + 4, ... -- the four waves together read whole 128-byte lines of every
// weight row.  mfma_f32_16x16x32_bf16 with A = 16 weight rows, B = 16 batch rows (two row blocks); D[n][m]: a
// lane owns batch row lane & 15 (and + 16) and the 4 consecutive columns 4 * (lane >> 4) .. + 3.  With fused RoPE
// a workgroup takes columns {d0 .. d0+7} and {d0+32 .. d0+39} of one 64-wide head: a rotary pair sits in lanes l
// and l ^ 32.  Same arithmetic as gemm_skinny16_bf16_kernel (vy_gemm.hip).
// ------------------------------------------------------------------------------------------
// LNP: the rows of X are normalised on the way in -- y = act(LayerNorm(x) W^T + b) without a LayerNorm launch in front.
// A workgroup holds every element of all 32 rows anyway (its four waves split K), so the row statistics are two
// reductions over values already in registers: over the 4 lane rows of a wave (permlane swaps), then over the 4 waves
// through LDS (one barrier each: mean, then the sum of squared deviations -- the two-pass form of dec_finish_ln_kernel,
// same rsqrt refinement, the normalised value rounded to bf16 exactly where that kernel stores it).  The launch this
// saves is a ~4.3 us link of the step's dependent chain; every workgroup redoes the statistics (25 k elements: noise).
template <int EPI, int ACT, int U, bool DBG, bool LNP = false>
__global__ __launch_bounds__(256) void dec_gemm16_kernel(const DecGemmArgs p, const DecDbg dbg) {
  __shared__ float red[3][8][64];
  __shared__ float lnred[2][4][32];
  unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DEC_STAMP(0)
  if constexpr (DBG) { asm volatile("" ::"s"(p.M), "s"(p.ldw)); }
  DEC_STAMP(5)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int nb = blockIdx.x;
  const bool rope_map = EPI == DEC_QKV && p.rope;
  int col = nb * 16 + r16;
  if (rope_map) col = (nb >> 2) * 64 + (nb & 3) * 8 + (r16 < 8 ? r16 : 24 + r16);
  const int m0 = r16 < p.M ? r16 : p.M - 1, m1 = 16 + r16 < p.M ? 16 + r16 : p.M - 1;
  const int kbase = ((int)blockIdx.y * 4 * U + wave) * 32 + kq * 8;
  const bf16* wp = p.W + (long long)col * p.ldw + kbase;
  const bf16* xp0 = p.X + (long long)m0 * p.ldx + kbase;
  const bf16* xp1 = p.X + (long long)m1 * p.ldx + kbase;
  if constexpr (DBG) { asm volatile("" ::"v"(wp), "v"(xp0), "v"(xp1)); }
  DEC_STAMP(6)
  // this lane's output quad: columns nq0 .. nq0 + 3
  const int nq0 = rope_map ? (nb >> 2) * 64 + (nb & 3) * 8 + 4 * (kq & 1) + (kq >= 2 ? 32 : 0) : nb * 16 + 4 * kq;

  // every load of the launch, oldest first: bias, the weight / activation fragments, then what depends on the
  // position (read on the device under a graph)
  bf16x4 bias4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  if (EPI != DEC_PART && p.bias) bias4 = *reinterpret_cast<const bf16x4*>(p.bias + nq0);
  bf16x8 a[U], b0[U], b1[U];
  bf16x8 lg[LNP ? U : 1], lb[LNP ? U : 1];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    a[u] = *reinterpret_cast<const bf16x8*>(wp + 128 * u);
    b0[u] = *reinterpret_cast<const bf16x8*>(xp0 + 128 * u);
    b1[u] = *reinterpret_cast<const bf16x8*>(xp1 + 128 * u);
    if constexpr (LNP) {
      lg[u] = *reinterpret_cast<const bf16x8*>(p.ln_g + kbase + 128 * u);
      lb[u] = *reinterpret_cast<const bf16x8*>(p.ln_b + kbase + 128 * u);
    }
  }
  bf16x4 res0 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, res1 = res0;
  if constexpr (EPI == DEC_STORE) {
    if (p.residual) {   // (workgroup-uniform; requested with everything else, used by wave 0 at the very end)
      res0 = *reinterpret_cast<const bf16x4*>(p.residual + (long long)m0 * p.ldr + nq0);
      res1 = *reinterpret_cast<const bf16x4*>(p.residual + (long long)m1 * p.ldr + nq0);
    }
  }
  int pos = 0;
  f32x4 cos4 = {1.f, 1.f, 1.f, 1.f}, sin4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == DEC_QKV) {
    pos = p.pos_dev ? *p.pos_dev : 0;   // (the host position is already in the cache pointers)
    if (rope_map) {
      const int dcol = (nb & 3) * 8 + 4 * (kq & 1);
      const long long pp = (long long)(p.pos0 + pos) * 32 + dcol;
      cos4 = *reinterpret_cast<const f32x4*>(p.cos_tab + pp);
      sin4 = *reinterpret_cast<const f32x4*>(p.sin_tab + pp);
    }
  }
  DEC_STAMP(1)
  if constexpr (LNP) {
    const float inv_n = 1.0f / (float)(128 * U);   // K = 4 waves x U k-steps x 32
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0 += (float)b0[u][e]; s1 += (float)b1[u][e]; }
    s0 = dec_rows_sum(s0); s1 = dec_rows_sum(s1);
    if (kq == 0) { lnred[0][wave][r16] = s0; lnred[0][wave][16 + r16] = s1; }
    __syncthreads();
    const float mean0 = (lnred[0][0][r16] + lnred[0][1][r16] + lnred[0][2][r16] + lnred[0][3][r16]) * inv_n;
    const float mean1 = (lnred[0][0][16 + r16] + lnred[0][1][16 + r16] + lnred[0][2][16 + r16] + lnred[0][3][16 + r16]) * inv_n;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d0 = (float)b0[u][e] - mean0, d1 = (float)b1[u][e] - mean1;
        q0 += d0 * d0; q1 += d1 * d1;
      }
    q0 = dec_rows_sum(q0); q1 = dec_rows_sum(q1);
    if (kq == 0) { lnred[1][wave][r16] = q0; lnred[1][wave][16 + r16] = q1; }
    __syncthreads();
    const float var0 = (lnred[1][0][r16] + lnred[1][1][r16] + lnred[1][2][r16] + lnred[1][3][r16]) * inv_n + p.ln_eps;
    const float var1 = (lnred[1][0][16 + r16] + lnred[1][1][16 + r16] + lnred[1][2][16 + r16] + lnred[1][3][16 + r16]) * inv_n + p.ln_eps;
    float r0 = rsqrtf(var0), r1 = rsqrtf(var1);
    r0 = r0 * (1.5f - 0.5f * var0 * r0 * r0);
    r1 = r1 * (1.5f - 0.5f * var1 * r1 * r1);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = (float)lg[u][e], be = (float)lb[u][e];
        b0[u][e] = (bf16)(((float)b0[u][e] - mean0) * r0 * g + be);
        b1[u][e] = (bf16)(((float)b1[u][e] - mean1) * r1 * g + be);
      }
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b0[u], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b1[u], acc1, 0, 0, 0);
  }
  if constexpr (DBG) { asm volatile("" :: "v"(acc0), "v"(acc1)); }
  DEC_STAMP(2)
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[wave - 1][r][lane] = acc0[r]; red[wave - 1][4 + r][lane] = acc1[r]; }
  }
  __syncthreads();
  if (wave != 0) return;
  DEC_STAMP(3)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    acc1[r] += red[0][4 + r][lane] + red[1][4 + r][lane] + red[2][4 + r][lane];
  }
  if constexpr (EPI == DEC_PART) {
    float* dst = p.part + ((long long)blockIdx.y * 32 + r16) * p.N + nq0;
    if (r16 < p.M) dec_store(dst, acc0, p.wt);
    if (16 + r16 < p.M) dec_store(dst + 16ll * p.N, acc1, p.wt);
  } else if constexpr (EPI == DEC_STORE) {
    bf16x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (+ residual in fp32, one rounding: the value dec_finish_ln_kernel normalises)
      o0[i] = (bf16)(vy_act_fwd_fast<ACT>(acc0[i] + (float)bias4[i]) + (float)res0[i]);
      o1[i] = (bf16)(vy_act_fwd_fast<ACT>(acc1[i] + (float)bias4[i]) + (float)res1[i]);
    }
    bf16* dst = p.y + (long long)r16 * p.ldy + nq0;
    if (r16 < p.M) dec_store(dst, o0, p.wt);
    if (16 + r16 < p.M) dec_store(dst + 16ll * p.ldy, o1, p.wt);
  } else {
    // head and offset inside the head of this lane's quad; section q / k / v is workgroup-uniform
    const int hd = nb >> 2;                       // 64-wide heads: 4 workgroups per head under either column map
    const int dcol = nq0 & 63;
    const int hq = p.nq >> 6, hkv = p.nkv >> 6;
    float v0[4], v1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v0[i] = acc0[i] + (float)bias4[i]; v1[i] = acc1[i] + (float)bias4[i]; }
    if (rope_map && hd < hq + hkv) {
      // reference op order (layers/positional_embeddings.py:173-181), every product rounded to bf16; the other
      // half of each rotary pair is in lane ^ 32
      const bool hi_half = kq >= 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float c = vy_round_bf16(cos4[i]), sn = vy_round_bf16(sin4[i]);
        const float o0 = dec_swap32(v0[i], hi_half), o1 = dec_swap32(v1[i], hi_half);
        const float a0 = vy_round_bf16(v0[i]), a1 = vy_round_bf16(v1[i]);
        const float t0 = vy_round_bf16(vy_round_bf16(o0) * sn), t1 = vy_round_bf16(vy_round_bf16(o1) * sn);
        v0[i] = hi_half ? vy_round_bf16(a0 * c) + t0 : vy_round_bf16(a0 * c) - t0;
        v1[i] = hi_half ? vy_round_bf16(a1 * c) + t1 : vy_round_bf16(a1 * c) - t1;
      }
    }
    bf16x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o0[i] = (bf16)v0[i]; o1[i] = (bf16)v1[i]; }
    bf16* base; long long rs;   // destination of row 0 and the row stride
    if (hd < hq) { base = p.q + hd * 64 + dcol; rs = p.nq; }
    else if (hd < hq + hkv) { base = p.k + (hd - hq) * p.c_sh + (long long)pos * p.c_sl + dcol; rs = p.c_sb; }
    else { base = p.v + (hd - hq - hkv) * p.c_sh + (long long)pos * p.c_sl + dcol; rs = p.c_sb; }
    if (r16 < p.M) dec_store(base + r16 * rs, o0, p.wt);
    if (16 + r16 < p.M) dec_store(base + (16 + r16) * rs, o1, p.wt);
  }
  if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  DEC_STAMP(4)
  DEC_STAMP_FLUSH()
}

// ------------------------------------------------------------------------------------------
// Partial tiles of the two N = d projections (out-projection, FFN2) for the split-K finish below: 64 output
// columns x 32 rows x a SHORT K range (U k-steps of 32) per workgroup, one 16-column block per WAVE, all four
// waves on the same K range.  The timeline (tools/decode_timeline.py) shows what a launch of dec_gemm16_kernel
// waits for: its 73 KB per workgroup (49 KB of it the activations, re-read by every workgroup) arrive at the
// ~30 GB/s one CU takes in -- 2.4 us.  Bytes per workgroup for a (c columns, k range) tile are 2k(32 + c) against
// k c of work: wide and short is cheapest, and costs nothing here because these projections are split over K
// anyway.  12 KB of activations (shared by the four waves through L1) + 24 KB of weights per workgroup, no
// cross-wave reduction, no LDS, no barrier.  part[gridDim.y][32][N] as in dec_gemm16_kernel<DEC_PART>.
// ------------------------------------------------------------------------------------------
template <int U, bool DBG>
__global__ __launch_bounds__(256) void dec_gemm64p_kernel(const DecGemmArgs p, const DecDbg dbg) {
  unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DEC_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int col = (int)blockIdx.x * 64 + wave * 16 + r16;
  const int m0 = r16 < p.M ? r16 : p.M - 1, m1 = 16 + r16 < p.M ? 16 + r16 : p.M - 1;
  const int kbase = (int)blockIdx.y * U * 32 + kq * 8;
  const bf16* wp = p.W + (long long)col * p.ldw + kbase;
  const bf16* xp0 = p.X + (long long)m0 * p.ldx + kbase;
  const bf16* xp1 = p.X + (long long)m1 * p.ldx + kbase;
  bf16x8 a[U], b0[U], b1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    a[u] = *reinterpret_cast<const bf16x8*>(wp + 32 * u);
    b0[u] = *reinterpret_cast<const bf16x8*>(xp0 + 32 * u);
    b1[u] = *reinterpret_cast<const bf16x8*>(xp1 + 32 * u);
  }
  DEC_STAMP(1)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b0[u], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b1[u], acc1, 0, 0, 0);
  }
  if constexpr (DBG) { asm volatile("" :: "v"(acc0), "v"(acc1)); }
  DEC_STAMP(2)
  float* dst = p.part + ((long long)blockIdx.y * 32 + r16) * p.N + (int)blockIdx.x * 64 + wave * 16 + 4 * kq;
  if (r16 < p.M) dec_store(dst, acc0, p.wt);
  if (16 + r16 < p.M) dec_store(dst + 16ll * p.N, acc1, p.wt);
  if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  DEC_STAMP(4)
  DEC_STAMP_FLUSH()
}

// ------------------------------------------------------------------------------------------
// y = LayerNorm(bf16(act(sum_k part[k] + bias) + residual)), ONE WORKGROUP PER ROW: every thread requests its
// 4 columns of every partial tile, bias, residual, gamma and beta at once (one memory round trip), then two
// block reductions (the arithmetic of splitk_finish_ln_row_kernel, vy_gemm.hip).
// ------------------------------------------------------------------------------------------
template <int KS, int QPT, int ACT, bool DBG>
__global__ __launch_bounds__(256) void dec_finish_ln_kernel(const float* __restrict__ part, int ksplit, int N,
                                                            const bf16* __restrict__ bias,
                                                            const bf16* __restrict__ residual, int ldr,
                                                            const bf16* __restrict__ gamma,
                                                            const bf16* __restrict__ beta, bf16* __restrict__ y,
                                                            int ldy, float eps, int wt, const DecDbg dbg) {
  __shared__ float red[2][4];
  unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DEC_STAMP(0)
  if constexpr (DBG) { asm volatile("" ::"s"(N), "s"(ksplit)); }
  DEC_STAMP(5)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  const int nq = N >> 2;
  f32x4 pk[QPT][KS];
  bf16x4 b4[QPT], r4[QPT], g4[QPT], be4[QPT];
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + 256 * i;
    const int n = (q < nq ? q : 0) * 4;
#pragma unroll
    for (int k = 0; k < KS; ++k)
      pk[i][k] = *reinterpret_cast<const f32x4*>(part + ((long long)(k < ksplit ? k : 0) * 32 + row) * N + n);
    b4[i] = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    r4[i] = b4[i];
    if (bias) b4[i] = *reinterpret_cast<const bf16x4*>(bias + n);
    if (residual) r4[i] = *reinterpret_cast<const bf16x4*>(residual + (long long)row * ldr + n);
    g4[i] = *reinterpret_cast<const bf16x4*>(gamma + n);
    be4[i] = *reinterpret_cast<const bf16x4*>(beta + n);
  }
  DEC_STAMP(1)
  float v[QPT][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const bool ok = tid + 256 * i < nq;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k < ksplit) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += pk[i][k][e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] += (float)b4[i][e];
      if constexpr (ACT != VY_ACT_NONE) a[e] = vy_act_fwd_fast<ACT>(a[e]);
      a[e] += (float)r4[i][e];
      v[i][e] = ok ? vy_round_bf16(a[e]) : 0.f;
      s += v[i][e];
    }
  }
  DEC_STAMP(2)
  s = dec_wave_sum(s);
  if (lane == 0) red[0][wave] = s;
  __syncthreads();
  const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)N;
  float qv = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    if (tid + 256 * i < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; qv += d * d; }
    }
  }
  qv = dec_wave_sum(qv);
  if (lane == 0) red[1][wave] = qv;
  __syncthreads();
  DEC_STAMP(3)
  const float var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)N;
  const float rstd = rsqrtf(var + eps);
  const float rstd_r = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + 256 * i;
    if (q < nq) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)((v[i][e] - mean) * rstd_r * (float)g4[i][e] + (float)be4[i][e]);
      dec_store(y + (long long)row * ldy + q * 4, o, wt);
    }
  }
  if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  DEC_STAMP(4)
  DEC_STAMP_FLUSH()
}

DecDbg g_dbg{nullptr, 0, 0};
int g_dbg_launches = 0;

int dec_wt() {
  static const int v = [] { const char* e = getenv("VY_DEC_WT"); return e ? atoi(e) : 0; }();
  return v;
}

template <int EPI, int ACT>
int dec_gemm_go(DecGemmArgs a, int K, int chunks, hipStream_t st) {
  a.wt = dec_wt();
  const int u4 = K / chunks / 128;   // k-steps per wave
  const dim3 grid((unsigned)(a.N / 16), (unsigned)chunks), block(256);
  if constexpr (EPI == DEC_STORE) {
    if (a.ln_g) {   // LayerNorm prologue: the workgroup must hold whole rows (one K chunk)
      if (chunks != 1) return 1;
      if (u4 == 6) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 6, false, true>), grid, block, 0, st, a, g_dbg);
      else if (u4 == 8) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 8, false, true>), grid, block, 0, st, a, g_dbg);
      else if (u4 == 4) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 4, false, true>), grid, block, 0, st, a, g_dbg);
      else return 1;
      return 0;
    }
  }
  if (g_dbg.buf && g_dbg.slot < g_dbg_launches) {
    if (u4 == 6) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 6, true>), grid, block, 0, st, a, g_dbg);
    else if (u4 == 8) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 8, true>), grid, block, 0, st, a, g_dbg);
    else if (u4 == 4) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 4, true>), grid, block, 0, st, a, g_dbg);
    else return 1;
    ++g_dbg.slot;
    return 0;
  }
  if (u4 == 6) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 6, false>), grid, block, 0, st, a, g_dbg);
  else if (u4 == 8) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 8, false>), grid, block, 0, st, a, g_dbg);
  else if (u4 == 4) hipLaunchKernelGGL((dec_gemm16_kernel<EPI, ACT, 4, false>), grid, block, 0, st, a, g_dbg);
  else return 1;
  return 0;
}

// partial tiles with 64-column workgroups: k-steps per workgroup U in {6, 8, 4}; returns the number of K chunks or 0
int dec_part64_go(DecGemmArgs a, int K, hipStream_t st) {
  a.wt = dec_wt();
  int u = 0;
  for (int c : {6, 8, 4})
    if (K % (32 * c) == 0 && K / (32 * c) <= 24) { u = c; break; }
  if (!u || a.N % 64) return 0;
  const int chunks = K / (32 * u);
  const dim3 grid((unsigned)(a.N / 64), (unsigned)chunks), block(256);
  const bool dbg_on = g_dbg.buf && g_dbg.slot < g_dbg_launches;
#define P64(UU)                                                                                   \
  do {                                                                                            \
    if (dbg_on) hipLaunchKernelGGL((dec_gemm64p_kernel<UU, true>), grid, block, 0, st, a, g_dbg);  \
    else hipLaunchKernelGGL((dec_gemm64p_kernel<UU, false>), grid, block, 0, st, a, g_dbg);        \
  } while (0)
  if (u == 6) P64(6); else if (u == 8) P64(8); else P64(4);
#undef P64
  if (dbg_on) ++g_dbg.slot;
  return chunks;
}

// ------------------------------------------------------------------------------------------
// Single-sequence matrix-vector products (Gemma-style decode, B = 1: BASELINE configs[4]).  One wave per output
// column, the row of the weight matrix in 16-byte pieces (1 KiB contiguous per wave instruction).  The general
// gemv_bf16_kernel (vy_gemm.hip) serves up to 4 rows with run-time row guards, and its ISA shows what that costs:
// the activation chunk of every k-piece is requested and waited for (vmcnt(0)) in the middle of the dot products,
// one dependent round trip per piece.  Here the row count is 1 and the piece count a template constant: straight-
// line code, the loads of batch b + 1 issued before the dot products of batch b, waits counted by the compiler.
//   NORM: the weights carry the preceding RMSNorm's (1 + w) along K; the wave accumulates sum x^2 from the chunks
//         it holds anyway and scales its result by rsqrt(mean x^2 + eps)              (no RMSNorm launch)
//   GV_QKV: rotary embedding fused: the four waves of a workgroup take the columns {d, d+1, d+dh/2, d+1+dh/2} of one
//         head and swap partners through LDS (rope2_kernel's arithmetic, vy_misc.hip)   (no RoPE launch)
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Single-query attention against the KV cache (reference models/decoder.py:107 / layers/attention.py with L = 1 and
// mask = None): one workgroup per (batch row, query head), every K and V chunk of the row's context requested
// BEFORE the first dot product (NP passes of NW * 64 / (DH / 8) keys: a compile-time count, no loop around the
// loads), plain two-pass softmax over the scores held in registers (no online rescaling: the whole context is
// resident), reductions on the DPP / permlane path, two barriers.  The general attn_rowwise_kernel walks the context in
// dependent trips with ds_bpermute reductions: 11.9 us for the 8 x 300-key rows of a PaliGemma-shape step, 14.7 us for
// 384 rows x 577 keys -- this one is bound by the K/V stream alone.
//   lane = (key group g, 16-byte chunk ch of the head): LPK = DH / 8 lanes per key, KPW = 64 / LPK keys per wave pass.
// ------------------------------------------------------------------------------------------
struct DecAttnArgs {
  const bf16* q; long long q_sb;            // q[b][head * DH + d]
  const bf16* k; const bf16* v; long long c_sb, c_sh, c_sl;
  bf16* out; long long o_sb;                // out[b][head * DH + d]
  const int* pos_dev; int S; int h, hk; float scale;
};

template <int LPK>
__device__ __forceinline__ float dec_group_sum(float v) {   // sum over the LPK lanes of a key group, in every lane
  if constexpr (LPK == 8) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    return v;
  } else {   // 32 lanes: a row of 16, then the neighbouring row
    v = dec_row16_sum(v);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
  }
}
// sum over the lanes that hold the same chunk ch (stride LPK), in every lane
template <int LPK>
__device__ __forceinline__ float dec_stride_sum(float v) {
  if constexpr (LPK == 8) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
  }
  const unsigned w = __builtin_bit_cast(unsigned, v);
  auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
}

// R > 1 (grouped-query attention): one workgroup per (batch row, KV head) serves the R query heads that share the head from
// ONE read of its K and V rows -- with a workgroup per query head the K/V stream (non-temporal: it does not stay on chip)
// was fetched R times, and a GQA step read as many bytes as the full-head model (0.492 against 0.484 ms per token step).
template <int DH, int NW, int NP, bool NT, int R = 1>
__global__ __launch_bounds__(64 * NW) void dec_attn_kernel(const DecAttnArgs p) {
  constexpr int LPK = DH / 8, KPW = 64 / LPK, KPP = NW * KPW;
  __shared__ float red_m[R][NW], red_l[R][NW];
  __shared__ float red_o[R][NW][DH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  // R == 1: blockIdx.x is the query head; R > 1: the KV head, query heads kvh * R .. + R - 1
  const int head0 = R == 1 ? (int)blockIdx.x : (int)blockIdx.x * R;
  const int kvh = R == 1 ? head0 / (p.h / p.hk) : (int)blockIdx.x;
  const int g = lane / LPK, ch = lane % LPK;
  const int S = p.pos_dev ? *p.pos_dev + 1 : p.S;
  const bf16* kb = p.k + (long long)b * p.c_sb + (long long)kvh * p.c_sh + ch * 8;
  const bf16* vb = p.v + (long long)b * p.c_sb + (long long)kvh * p.c_sh + ch * 8;
  bf16x8 q8[R];
#pragma unroll
  for (int r = 0; r < R; ++r) q8[r] = *reinterpret_cast<const bf16x8*>(p.q + (long long)b * p.q_sb + (head0 + r) * DH + ch * 8);
  bf16x8 kr[NP], vr[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int j = i * KPP + wave * KPW + g;
    const long long jo = (long long)(j < S ? j : S - 1) * p.c_sl;
    if constexpr (NT) { kr[i] = dec_load_stream(kb + jo); vr[i] = dec_load_stream(vb + jo); }
    else { kr[i] = *reinterpret_cast<const bf16x8*>(kb + jo); vr[i] = *reinterpret_cast<const bf16x8*>(vb + jo); }
  }
  float sc[R][NP];
  float m[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)q8[r][e];
    m[r] = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf(qf[e], (float)kr[i][e], d);
      d = dec_group_sum<LPK>(d) * p.scale;
      const int j = i * KPP + wave * KPW + g;
      sc[r][i] = j < S ? d : -FLT_MAX;
      m[r] = fmaxf(m[r], sc[r][i]);
    }
    m[r] = dec_wave_max(m[r]);
    if (lane == 0) red_m[r][wave] = m[r];
  }
  __syncthreads();
  float l[R], acc[R][8];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float M = red_m[r][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, red_m[r][w]);
    l[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int j = i * KPP + wave * KPW + g;
      const float ev = j < S ? __expf(sc[r][i] - M) : 0.f;
      l[r] += ev;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[r][e] = fmaf(ev, (float)vr[i][e], acc[r][e]);
    }
    // per wave: the weights of its keys (each key counted once: chunk 0 of its group) and, per chunk, the sum over
    // the wave's key groups
    l[r] = dec_wave_sum(ch == 0 ? l[r] : 0.f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[r][e] = dec_stride_sum<LPK>(acc[r][e]);
    if (lane == 0) red_l[r][wave] = l[r];
    if (g == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red_o[r][wave][ch * 8 + e] = acc[r][e];
    }
  }
  __syncthreads();
  for (int t = tid; t < R * DH; t += 64 * NW) {
    const int r = t / DH, c = t - r * DH;
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { L += red_l[r][w]; o += red_o[r][w][c]; }
    p.out[(long long)b * p.o_sb + (head0 + r) * DH + c] = (bf16)(o / L);
  }
}

template <int DH, int NW, bool NT, int R = 1>
int dec_attn_go(const DecAttnArgs& a, int B, int Smax, hipStream_t st) {
  constexpr int KPP = NW * (64 / (DH / 8));
  const dim3 grid((unsigned)(R == 1 ? a.h : a.hk), (unsigned)B), block(64 * NW);
  const int np = (Smax + KPP - 1) / KPP;
  if (np <= 3) hipLaunchKernelGGL((dec_attn_kernel<DH, NW, 3, NT, R>), grid, block, 0, st, a);
  else if (np <= 5) hipLaunchKernelGGL((dec_attn_kernel<DH, NW, 5, NT, R>), grid, block, 0, st, a);
  else if (np <= 7) hipLaunchKernelGGL((dec_attn_kernel<DH, NW, 7, NT, R>), grid, block, 0, st, a);
  else if (np <= 10) hipLaunchKernelGGL((dec_attn_kernel<DH, NW, 10, NT, R>), grid, block, 0, st, a);
  else if (np <= 12) hipLaunchKernelGGL((dec_attn_kernel<DH, NW, 12, NT, R>), grid, block, 0, st, a);
  else return 1;
  return 0;
}

enum { GV_PLAIN = 0, GV_GATED = 1, GV_QKV = 2 };
struct DecGemvArgs {
  const bf16* x; const bf16* w; const bf16* bias; const bf16* residual; bf16* y;
  int N, K, ldw;
  float eps;
  // GV_QKV
  bf16* q; bf16* k; bf16* v; long long c_sh;   // q [nq]; k / v: cache row of this position, head stride c_sh
  const float* cos_tab; const float* sin_tab; int pos, nq, nkv, dh;
};

template <int EPI, int CH, int NB, bool NORM>
__global__ __launch_bounds__(256) void dec_gemv1_kernel(const DecGemvArgs p) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  __shared__ float rope_x[4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int n = (int)blockIdx.x * 4 + wv;
  const bool rope_on = EPI == GV_QKV && p.cos_tab != nullptr;
  if (rope_on) {
    const int per_head = p.dh >> 2;
    const int hd = (int)blockIdx.x / per_head, pi = (int)blockIdx.x - hd * per_head;
    n = hd * p.dh + 2 * pi + (wv & 1) + (p.dh >> 1) * (wv >> 1);
  }
  const bf16* wp = p.w + (long long)n * p.ldw + lane * 8;
  const bf16* wp2 = p.w + (long long)(p.N + n) * p.ldw + lane * 8;   // GV_GATED: the "up" row
  const bf16* xp = p.x + lane * 8;
  float acc = 0.f, acc2 = 0.f, sq = 0.f;
  bf16x8 wa[2][CH], wb[2][CH], xa[2][CH];
  auto issue = [&](int b, int buf) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int off = (b * CH + u) * 512;
      wa[buf][u] = dec_load_stream(wp + off);   // 5.8 GB of weights per token, each byte once: streamed
      if constexpr (EPI == GV_GATED) wb[buf][u] = dec_load_stream(wp2 + off);
      xa[buf][u] = *reinterpret_cast<const bf16x8*>(xp + off);
    }
  };
  issue(0, 0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) issue(b + 1, (b + 1) & 1);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      union { bf16x8 v; bf16x2_t h[4]; } a, a2, x;
      a.v = wa[b & 1][u];
      x.v = xa[b & 1][u];
      if constexpr (EPI == GV_GATED) a2.v = wb[b & 1][u];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = __builtin_amdgcn_fdot2_f32_bf16(a.h[e], x.h[e], acc, false);
        if constexpr (EPI == GV_GATED) acc2 = __builtin_amdgcn_fdot2_f32_bf16(a2.h[e], x.h[e], acc2, false);
        if constexpr (NORM) sq = __builtin_amdgcn_fdot2_f32_bf16(x.h[e], x.h[e], sq, false);
      }
    }
  }
  acc = dec_wave_sum(acc);
  if constexpr (EPI == GV_GATED) acc2 = dec_wave_sum(acc2);
  if constexpr (NORM) {
    const float ms = dec_wave_sum(sq) / (float)p.K + p.eps;
    const float r = rsqrtf(ms);
    const float rr = r * (1.5f - 0.5f * ms * r * r);
    acc *= rr;
    acc2 *= rr;
  }
  if constexpr (EPI == GV_PLAIN) {
    if (lane == 0) {
      float x = acc;
      if (p.bias) x += (float)p.bias[n];
      if (p.residual) x += (float)p.residual[n];
      p.y[n] = (bf16)x;
    }
  } else if constexpr (EPI == GV_GATED) {
    if (lane == 0) {
      const float gte = vy_round_bf16(acc), up = vy_round_bf16(acc2);
      p.y[n] = (bf16)(vy_gelu_tanh(gte) * up);
    }
  } else {
    float x = acc;
    if (p.bias) x += (float)p.bias[n];
    x = vy_round_bf16(x);   // the projection as the unfused path stores it
    float o = x;
    if (rope_on) {
      if (lane == 0) rope_x[wv] = x;
      __syncthreads();
      if (n < p.nq + p.nkv) {
        const int half = p.dh >> 1;
        const int d = (n & (p.dh - 1)) & (half - 1);
        const float other = rope_x[wv ^ 2];
        const long long pp = (long long)p.pos * half + d;
        const float c = vy_round_bf16(p.cos_tab[pp]), sn = vy_round_bf16(p.sin_tab[pp]);
        o = (wv < 2) ? vy_round_bf16(x * c) + vy_round_bf16(-other * sn) : vy_round_bf16(x * c) + vy_round_bf16(other * sn);
      }
    }
    if (lane == 0) {
      const int dlow = n & (p.dh - 1);
      bf16* dst;
      if (n < p.nq) dst = p.q + n;
      else if (n < p.nq + p.nkv) dst = p.k + (long long)((n - p.nq) / p.dh) * p.c_sh + dlow;
      else dst = p.v + (long long)((n - p.nq - p.nkv) / p.dh) * p.c_sh + dlow;
      *dst = (bf16)o;
    }
  }
}

template <int EPI, bool NORM>
int dec_gemv1_go(const DecGemvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(a.N / 4)), block(256);
  if (a.K == 2048) hipLaunchKernelGGL((dec_gemv1_kernel<EPI, 4, 1, NORM>), grid, block, 0, st, a);
  else if (a.K == 4096) hipLaunchKernelGGL((dec_gemv1_kernel<EPI, 8, 1, NORM>), grid, block, 0, st, a);
  else if (a.K == 8192) hipLaunchKernelGGL((dec_gemv1_kernel<EPI, 8, 2, NORM>), grid, block, 0, st, a);
  else if (a.K == 16384) hipLaunchKernelGGL((dec_gemv1_kernel<EPI, 8, 4, NORM>), grid, block, 0, st, a);
  else return 1;
  return 0;
}

}  // namespace

// K ranges a workgroup of dec_gemm16_kernel can take: 4 waves x {4, 6, 8} k-steps of 32
static int dec_chunks(int64_t K, bool may_split) {
  for (int u : {6, 8, 4}) {
    const int64_t c = 128 * u;
    if (K % c == 0 && (K == c || (may_split && K / c <= 12))) return (int)(K / c);
  }
  return 0;
}

// can the lean kernels run a model of these sizes?  (bf16, <= 32 rows, 64-wide heads, the K of every projection a
// chunk size the straight-line kernels exist for; otherwise the decode driver keeps the general launchers)
bool vy_dec_supported(int B, int d, int h, int hk, int dh, int ffn, int dtype) {
  return dtype == VY_BF16 && B >= 1 && B <= 32 && dh == 64 && h * dh == d && d % 16 == 0 && d % 4 == 0 && d <= 8192 && ffn % 16 == 0 &&
         (h + 2 * hk) * dh % 16 == 0 && dec_chunks(d, false) == 1 && dec_chunks(ffn, true) > 0;
}

int vy_dec_qkv(const void* x, const void* w, const void* bias, const float* cos_tab, const float* sin_tab, int64_t pos0,
               const int* pos_dev, void* q, void* k, void* v, int64_t c_sb, int64_t c_sh, int64_t c_sl, int B, int d, int h,
               int hk, hipStream_t st) {
  DecGemmArgs a{};
  a.X = (const bf16*)x; a.W = (const bf16*)w; a.bias = (const bf16*)bias;
  a.ldx = d; a.ldw = d; a.M = B; a.N = (h + 2 * hk) * 64;
  a.q = (bf16*)q; a.k = (bf16*)k; a.v = (bf16*)v; a.c_sb = c_sb; a.c_sh = c_sh; a.c_sl = c_sl;
  a.cos_tab = cos_tab; a.sin_tab = sin_tab; a.pos_dev = pos_dev; a.pos0 = pos_dev ? 0 : (int)pos0;
  a.nq = h * 64; a.nkv = hk * 64; a.rope = cos_tab != nullptr;
  if (dec_gemm_go<DEC_QKV, VY_ACT_NONE>(a, d, 1, st)) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_dec_qkv: K = %d", d);
  VY_CHECK_LAUNCH("vy_dec_qkv");
  return VY_OK;
}

// y[B][N] = bf16(act(LN(x) W^T + b) + residual), K one chunk.  residual / ln_g / ln_b may be NULL (no residual / x as is)
int vy_dec_linear_ex(const void* x, int ldx, const void* w, const void* bias, const void* residual, int ldr, const void* ln_g,
                     const void* ln_b, float ln_eps, void* y, int ldy, int B, int N, int K, int act, hipStream_t st) {
  DecGemmArgs a{};
  a.X = (const bf16*)x; a.W = (const bf16*)w; a.bias = (const bf16*)bias;
  a.ldx = ldx; a.ldw = K; a.M = B; a.N = N; a.y = (bf16*)y; a.ldy = ldy;
  a.residual = (const bf16*)residual; a.ldr = ldr;
  a.ln_g = (const bf16*)ln_g; a.ln_b = (const bf16*)ln_b; a.ln_eps = ln_eps;
  if (a.ln_g && (K % 8 || ((uintptr_t)ln_g & 15) || ((uintptr_t)ln_b & 15))) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_dec_linear_ex: LayerNorm operands");
  int rc;
  if (act == VY_ACT_GELU_ERF) rc = dec_gemm_go<DEC_STORE, VY_ACT_GELU_ERF>(a, K, 1, st);
  else if (act == VY_ACT_GELU_TANH) rc = dec_gemm_go<DEC_STORE, VY_ACT_GELU_TANH>(a, K, 1, st);
  else rc = dec_gemm_go<DEC_STORE, VY_ACT_NONE>(a, K, 1, st);
  if (rc) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_dec_linear_ex: K = %d", K);
  VY_CHECK_LAUNCH("vy_dec_linear_ex");
  return VY_OK;
}

// y[B][N] = act(x W^T + b), K one chunk
int vy_dec_linear(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy, int B, int N, int K, int act,
                  hipStream_t st) {
  DecGemmArgs a{};
  a.X = (const bf16*)x; a.W = (const bf16*)w; a.bias = (const bf16*)bias;
  a.ldx = ldx; a.ldw = K; a.M = B; a.N = N; a.y = (bf16*)y; a.ldy = ldy;
  int rc;
  if (act == VY_ACT_GELU_ERF) rc = dec_gemm_go<DEC_STORE, VY_ACT_GELU_ERF>(a, K, 1, st);
  else if (act == VY_ACT_GELU_TANH) rc = dec_gemm_go<DEC_STORE, VY_ACT_GELU_TANH>(a, K, 1, st);
  else rc = dec_gemm_go<DEC_STORE, VY_ACT_NONE>(a, K, 1, st);
  if (rc) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_dec_linear: K = %d", K);
  VY_CHECK_LAUNCH("vy_dec_linear");
  return VY_OK;
}

// y = LayerNorm(bf16(act(x W^T + b) + residual)): split-K partial tiles, then one workgroup per row
int vy_dec_linear_res_ln(const void* x, int ldx, const void* w, const void* bias, const void* residual, int ldr,
                         const void* gamma, const void* beta, float eps, void* y, int ldy, float* part, int B, int N, int K,
                         int act, hipStream_t st) {
  const char* who = "vy_dec_linear_res_ln";
  if (N % 16 || N % 4 || N > 8192) VY_FAIL(VY_ERR_UNSUPPORTED, "%s: N = %d", who, N);
  DecGemmArgs a{};
  a.X = (const bf16*)x; a.W = (const bf16*)w; a.ldx = ldx; a.ldw = K; a.M = B; a.N = N; a.part = part;
  static const int p64_env = [] { const char* e = getenv("VY_DEC_P64"); return e ? atoi(e) : 1; }();
  int chunks = p64_env ? dec_part64_go(a, K, st) : 0;
  if (!chunks) {
    chunks = dec_chunks(K, true);
    if (!chunks) VY_FAIL(VY_ERR_UNSUPPORTED, "%s: N = %d, K = %d", who, N, K);
    if (dec_gemm_go<DEC_PART, VY_ACT_NONE>(a, K, chunks, st)) VY_FAIL(VY_ERR_UNSUPPORTED, "%s: K = %d", who, K);
  }
  VY_CHECK_LAUNCH(who);
  const int qpt = (int)vy_cdiv(N / 4, 256);
  const bool dbg_on = g_dbg.buf && g_dbg.slot < g_dbg_launches;
#define FIN(KS, Q, A)                                                                                                  \
  do {                                                                                                                 \
    if (dbg_on)                                                                                                        \
      hipLaunchKernelGGL((dec_finish_ln_kernel<KS, Q, A, true>), dim3((unsigned)B), dim3(256), 0, st, part, chunks, N, \
                         (const bf16*)bias, (const bf16*)residual, ldr, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, ldy, eps, dec_wt(), g_dbg); \
    else                                                                                                               \
      hipLaunchKernelGGL((dec_finish_ln_kernel<KS, Q, A, false>), dim3((unsigned)B), dim3(256), 0, st, part, chunks, N, \
                         (const bf16*)bias, (const bf16*)residual, ldr, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, ldy, eps, dec_wt(), g_dbg); \
  } while (0)
#define FIN_Q(KS, A)                                                                                                   \
  do { if (qpt <= 1) FIN(KS, 1, A); else if (qpt <= 2) FIN(KS, 2, A); else if (qpt <= 4) FIN(KS, 4, A); else FIN(KS, 8, A); } while (0)
#define FIN_K(A)                                                                                                       \
  do { if (chunks <= 1) FIN_Q(1, A); else if (chunks <= 4) FIN_Q(4, A); else if (chunks <= 8) FIN_Q(8, A); else if (chunks <= 16) FIN_Q(16, A); else FIN_Q(24, A); } while (0)
  if (act == VY_ACT_GELU_ERF) FIN_K(VY_ACT_GELU_ERF);
  else if (act == VY_ACT_GELU_TANH) FIN_K(VY_ACT_GELU_TANH);
  else FIN_K(VY_ACT_NONE);
#undef FIN_K
#undef FIN_Q
#undef FIN
  if (dbg_on) ++g_dbg.slot;
  VY_CHECK_LAUNCH(who);
  return VY_OK;
}

// measurement aid, not part of include/vyom_hip.h: the next `launches` launches of the kernels in this file write
// their stamps into buf[launches][max_wg][8] (device memory, 64-bit words); buf = NULL switches it off
extern "C" int vy_debug_set_decode_stamps(void* buf, int launches, int max_wg) {
  g_dbg.buf = (unsigned long long*)buf; g_dbg.slot = 0; g_dbg.max_wg = max_wg; g_dbg_launches = launches;
  return 0;
}

// ---- single-sequence (B = 1) products of the Gemma-style decode step; return VY_ERR_UNSUPPORTED (without an error
// message being fatal: the driver falls back to the general kernels) for shapes the straight-line kernels do not cover
static bool gemv1_ok(const void* x, const void* w, int N, int K) {
  return N % 4 == 0 && (K == 2048 || K == 4096 || K == 8192 || K == 16384) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0);
}
int vy_dec_gemv1(const void* x, const void* w, const void* bias, const void* residual, void* y, int N, int K, int prescaled,
                 float eps, hipStream_t st) {
  if (!gemv1_ok(x, w, N, K)) return VY_ERR_UNSUPPORTED;
  DecGemvArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const bf16*)bias; a.residual = (const bf16*)residual; a.y = (bf16*)y;
  a.N = N; a.K = K; a.ldw = K; a.eps = eps;
  if (prescaled ? dec_gemv1_go<GV_PLAIN, true>(a, st) : dec_gemv1_go<GV_PLAIN, false>(a, st)) return VY_ERR_UNSUPPORTED;
  VY_CHECK_LAUNCH("vy_dec_gemv1");
  return VY_OK;
}
int vy_dec_gemv1_gated(const void* x, const void* wgu, void* y, int I, int K, int prescaled, float eps, hipStream_t st) {
  if (!gemv1_ok(x, wgu, I, K)) return VY_ERR_UNSUPPORTED;
  DecGemvArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wgu; a.y = (bf16*)y; a.N = I; a.K = K; a.ldw = K; a.eps = eps;
  if (prescaled ? dec_gemv1_go<GV_GATED, true>(a, st) : dec_gemv1_go<GV_GATED, false>(a, st)) return VY_ERR_UNSUPPORTED;
  VY_CHECK_LAUNCH("vy_dec_gemv1_gated");
  return VY_OK;
}
int vy_dec_gemv1_qkv(const void* x, const void* w, const void* bias, void* q, void* k, void* v, int64_t c_sh, int h, int hk,
                     int dh, const float* cos_tab, const float* sin_tab, int64_t pos, int K, int prescaled, float eps,
                     hipStream_t st) {
  const int N = (h + 2 * hk) * dh;
  if (!gemv1_ok(x, w, N, K) || dh % 4 || (dh & (dh - 1))) return VY_ERR_UNSUPPORTED;
  DecGemvArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const bf16*)bias; a.N = N; a.K = K; a.ldw = K; a.eps = eps;
  a.q = (bf16*)q; a.k = (bf16*)k; a.v = (bf16*)v; a.c_sh = c_sh; a.cos_tab = cos_tab; a.sin_tab = sin_tab; a.pos = (int)pos;
  a.nq = h * dh; a.nkv = hk * dh; a.dh = dh;
  if (prescaled ? dec_gemv1_go<GV_QKV, true>(a, st) : dec_gemv1_go<GV_QKV, false>(a, st)) return VY_ERR_UNSUPPORTED;
  VY_CHECK_LAUNCH("vy_dec_gemv1_qkv");
  return VY_OK;
}

// single-query attention of the decode steps.  smax: an upper bound of the context length when it is read on the
// device (pos_dev), else S.  Returns VY_ERR_UNSUPPORTED for shapes the resident-context kernel does not cover.
int vy_dec_attn(const void* q, int64_t q_sb, const void* k, const void* v, int64_t c_sb, int64_t c_sh, int64_t c_sl, void* out,
                int64_t o_sb, int B, int h, int hk, int64_t S, int64_t smax, const int* pos_dev, int dh, float scale,
                hipStream_t st) {
  static const int on = [] { const char* e = getenv("VY_DEC_ATTN"); return e ? atoi(e) : 1; }();
  if (!on || (dh != 64 && dh != 256) || hk < 1 || h % hk || S < 1 || c_sl % 8 || c_sh % 8 || c_sb % 8 || q_sb % 8 ||
      ((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16))
    return VY_ERR_UNSUPPORTED;
  DecAttnArgs a{};
  a.q = (const bf16*)q; a.q_sb = q_sb; a.k = (const bf16*)k; a.v = (const bf16*)v; a.c_sb = c_sb; a.c_sh = c_sh; a.c_sl = c_sl;
  a.out = (bf16*)out; a.o_sb = o_sb; a.pos_dev = pos_dev; a.S = (int)S; a.h = h; a.hk = hk; a.scale = scale;
  const int bound = (int)(pos_dev ? smax : S);
  // K/V policy (see dec_load_stream): streamed (nt) unless one layer's rows are under 1 MB.  Measured on the 12-layer
  // d = 768 decoder, whose 170 MB of weights can stay in the Infinity Cache if the K/V stream does not push them out: nt is
  // ahead at every batch (B = 1 / 2 / 4 / 8 / 32: 349 / 360 / 380 / 368 / 485 us against 355 / 363 / 388 / 399 / 515); on
  // configs[4] (0.3 MB per layer, 5.8 GB of weights streaming through anyway) the default policy is 5 % ahead.
  // VY_DEC_ATTN_NT = 0 / 1 forces it
  static const int nt_env = [] { const char* e = getenv("VY_DEC_ATTN_NT"); return e ? atoi(e) : -1; }();
  const int64_t layer_bytes = 2ll * B * hk * (pos_dev ? smax : S) * dh * 2;
  const bool nt = nt_env < 0 ? layer_bytes > (1ll << 20) : nt_env != 0;
  int rc;
  // grouped-query heads of a 64-wide model: the query heads of one KV head in one workgroup (K/V read once, not h / hk times)
  static const int gq_env = [] { const char* e = getenv("VY_DEC_ATTN_GQ"); return e ? atoi(e) : 1; }();
  const int rep = h / hk;
  if (dh == 64 && gq_env && nt && rep == 2) rc = dec_attn_go<64, 8, true, 2>(a, B, bound, st);
  else if (dh == 64 && gq_env && nt && rep == 3) rc = dec_attn_go<64, 8, true, 3>(a, B, bound, st);
  else if (dh == 64 && gq_env && nt && rep == 4) rc = dec_attn_go<64, 8, true, 4>(a, B, bound, st);
  else if (dh == 64) rc = nt ? dec_attn_go<64, 8, true>(a, B, bound, st) : dec_attn_go<64, 8, false>(a, B, bound, st);
  else rc = nt ? dec_attn_go<256, 16, true>(a, B, bound, st) : dec_attn_go<256, 16, false>(a, B, bound, st);
  if (rc) return VY_ERR_UNSUPPORTED;
  VY_CHECK_LAUNCH("vy_dec_attn");
  return VY_OK;
}
