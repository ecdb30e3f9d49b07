// Common device/host helpers for libvyom_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/vyom_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VY_LDS __attribute__((address_space(3)))
#define VY_GLOBAL __attribute__((address_space(1)))

// ---- transposing LDS reads that the compiler's waitcnt pass cannot see ----------------------
// hipcc cannot tell what the ds_read_b64_tr_b16 *builtin* aliases, so whenever LDS-DMA
// (global_load ... lds, counted by vmcnt) is in flight it puts s_waitcnt vmcnt(0) in front of the
// read: the prefetch of the next tile is drained before the current tile's MFMAs even start.
// The asm form is invisible to that pass.  Its result is NOT tracked either: retire the reads with
// vy_lgkm_wait<N>(frags...) (N = LDS reads issued after the ones needed) before the first use.
// A compiler-issued lgkmcnt wait in between only ever over-waits (LDS returns in order).
__device__ __forceinline__ s16x4 vy_lds_tr16(const char* p) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"((unsigned)(uintptr_t)(VY_LDS const char*)p) : "memory");
  return r;
}
// the same with a compile-time byte offset in the instruction's 16-bit offset field: the asm form
// hides the address from the compiler, so it cannot fold constants into that field itself and would
// spend a v_add per read
template <int OFF>
__device__ __forceinline__ s16x4 vy_lds_tr16_off(unsigned lds_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF0, int OFF1>
__device__ __forceinline__ bf16x8 vy_lds_tr16_pair_off(unsigned lds_addr) {
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = vy_lds_tr16_off<OFF0>(lds_addr);
  u.s.b = vy_lds_tr16_off<OFF1>(lds_addr);
  return u.v;
}
// ds_read_b128 in the same hidden form: hipcc retires its own LDS reads with lgkmcnt(0), which also
// waits for the fragments just requested for the NEXT k-step; hidden reads are retired by the caller
// with a counted lgkmcnt, so a k-step's MFMAs wait only for their own operands.
template <int OFF>
__device__ __forceinline__ bf16x8 vy_lds_read128_off(unsigned lds_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ unsigned vy_lds_addr(const char* p) { return (unsigned)(uintptr_t)(VY_LDS const char*)p; }
// rows r and r+8 of a transposed 16-row block -> one MFMA A/B fragment (2 LDS reads, no wait)
__device__ __forceinline__ bf16x8 vy_lds_tr16_pair(const char* p0, const char* p1) {
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = vy_lds_tr16(p0);
  u.s.b = vy_lds_tr16(p1);
  return u.v;
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant
// expression in the body (instruction offset fields, register-array indices)
template <int N, typename F>
__device__ __forceinline__ void vy_static_for(F&& f) {
  if constexpr (N > 0) {
    vy_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <typename T>
__device__ __forceinline__ void vy_tie(T& v) { asm volatile("" : "+v"(v)); }
template <int N, typename... T>
__device__ __forceinline__ void vy_lgkm_wait(T&... v) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  // tie every fragment to this point: the MFMAs that read them cannot be scheduled above the wait
  (vy_tie(v), ...);
}

// host-side error plumbing -------------------------------------------------------------
void vy_set_error(const char* fmt, ...);
#define VY_FAIL(code, ...)      \
  do {                          \
    vy_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)
#define VY_CHECK_LAUNCH(name)                                             \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess)                                                 \
      VY_FAIL(VY_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_));      \
  } while (0)

static inline int64_t vy_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 16 zero bytes: source for out-of-range chunks of LDS-DMA loads (one copy per translation unit:
// the library is built without relocatable device code).
static __device__ __attribute__((aligned(16))) uint32_t vy_zero16[4] = {0, 0, 0, 0};

// device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float vy_gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float vy_gelu_erf_grad(float x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float vy_gelu_tanh(float x) {
  const float c = 0.79788456080286535588f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float vy_gelu_tanh_grad(float x) {
  const float c = 0.79788456080286535588f;
  const float u = c * (x + 0.044715f * x * x * x);
  const float t = tanhf(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
}
// bf16-path GELU: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below a bf16 ulp) --
// one rcp, one exp2, five fma instead of ocml's erff; the exponential exp(-x^2/2) is shared
// with the Gaussian factor of the derivative.
__device__ __forceinline__ void vy_phi_fast(float x, float& cdf, float& gauss) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  gauss = __expf(-ax * ax);  // exp(-x^2/2)
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float erf_abs = 1.0f - poly * t * gauss;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float vy_gelu_erf_fast(float x) {
  float cdf, g;
  vy_phi_fast(x, cdf, g);
  return x * cdf;
}
__device__ __forceinline__ float vy_gelu_erf_grad_fast(float x) {
  float cdf, g;
  vy_phi_fast(x, cdf, g);
  return cdf + x * 0.39894228040143267794f * g;
}

template <int ACT>
__device__ __forceinline__ float vy_act_fwd(float x) {
  if constexpr (ACT == VY_ACT_GELU_ERF) return vy_gelu_erf(x);
  else if constexpr (ACT == VY_ACT_GELU_TANH) return vy_gelu_tanh(x);
  else return x;
}
template <int ACT>
__device__ __forceinline__ float vy_act_grad(float x) {
  if constexpr (ACT == VY_ACT_GELU_ERF) return vy_gelu_erf_grad(x);
  else if constexpr (ACT == VY_ACT_GELU_TANH) return vy_gelu_tanh_grad(x);
  else return 1.0f;
}

__device__ __forceinline__ float vy_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float vy_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// reduced-cost variants for bf16 storage (results are rounded to 8 bits of mantissa anyway)
template <int ACT>
__device__ __forceinline__ float vy_act_fwd_fast(float x) {
  if constexpr (ACT == VY_ACT_GELU_ERF) return vy_gelu_erf_fast(x);
  else return vy_act_fwd<ACT>(x);
}
template <int ACT>
__device__ __forceinline__ float vy_act_grad_fast(float x) {
  if constexpr (ACT == VY_ACT_GELU_ERF) return vy_gelu_erf_grad_fast(x);
  else return vy_act_grad<ACT>(x);
}

// Round an fp32 value to bf16 and back.  Goes through the bit pattern: clang evaluates __bf16
// expressions with excess precision, so a plain (float)(bf16)x round trip may be elided.
__device__ __forceinline__ float vy_round_bf16(float x) {
  const bf16 b = (bf16)x;
  const unsigned short u = __builtin_bit_cast(unsigned short, b);
  return __builtin_bit_cast(float, (unsigned)u << 16);
}

// ---- dropout: counter-based keep mask ----------------------------------------------------------
// nn.Dropout(hidden_dropout_prob) in AttentionSelfOutput / FeedForward (reference layers/attention.py:70,
// layers/ffn.py:38).  The mask is a pure function of (seed, offset, row, column): Philox4x32-7 on the
// counter {column / 8, row, offset} gives eight 16-bit lots for the eight columns of a 16-byte bf16
// chunk; an element is dropped when its lot is below thr = round(p * 65536).  The forward epilogue,
// vy_dropout (backward: the same mask on the incoming gradient) and the tests' mask export all call
// this one function, so nothing has to be stored.
struct VyDrop {
  uint32_t thr;       // 0 = no dropout
  float scale;        // 1 / (1 - p)
  uint32_t seed_lo, seed_hi, off_lo, off_hi;
};
__host__ inline VyDrop vy_make_drop(float p, uint64_t seed, uint64_t offset) {
  VyDrop d{};
  if (p > 0.f) {
    double t = (double)p * 65536.0 + 0.5;
    d.thr = t >= 65536.0 ? 65536u : (uint32_t)t;
    d.scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    d.seed_lo = (uint32_t)seed; d.seed_hi = (uint32_t)(seed >> 32);
    d.off_lo = (uint32_t)offset; d.off_hi = (uint32_t)(offset >> 32);
  }
  return d;
}
__device__ __forceinline__ void vy_philox7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// the eight lots of the chunk holding columns 8 * chunk .. 8 * chunk + 7 of row m
__device__ __forceinline__ void vy_drop_lots(const VyDrop& d, int64_t m, int chunk, uint32_t (&r)[4]) {
  vy_philox7((uint32_t)chunk, (uint32_t)m, d.off_lo, d.off_hi ^ (uint32_t)((uint64_t)m >> 32), d.seed_lo, d.seed_hi, r);
}
__device__ __forceinline__ bool vy_drop_keep(const VyDrop& d, const uint32_t (&r)[4], int e) {   // e = column & 7
  return ((r[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= d.thr;
}

// load/store helpers templated on the storage type
template <typename T> struct VyT;
template <> struct VyT<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct VyT<bf16> {
  static __device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }
};
