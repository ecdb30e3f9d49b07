// GEMM family for the projection / feed-forward path:  Y[M,N] = epi(X[M,K] . W[N,K]^T)
//
// bf16 kernels (the product path): MFMA, fp32 accumulate, LDS-DMA staging, by size of M:
//   * M > ~2304 (training, prefill): 256 x 192 tiles with the X operand in a three-deep ring (gemm_nt_bf16_x3m16_kernel) or
//     256 x 256 two-stage tiles where N >= 3072 (gemm_nt_bf16_m16_kernel), both on mfma_f32_16x16x32_bf16, 8 waves, one
//     workgroup per CU; gemm_nt_bf16_kernel is the same tile family on mfma_f32_32x32x16_bf16 (A/B knob, 32 x 128 fallback);
//   * 32 < M <= ~2304: 128 x 128 tiles (two workgroups per CU) or ONE all-rows tile of 256 / 320 x 128 for 128 < M <= 320,
//     split over K when the tile grid leaves CUs idle (gemm_nt_bf16_m16_splitk_kernel + gemm_splitk_finish_kernel);
//   * M <= 32 (decode): weight-streaming kernels without LDS staging (gemm_skinny*), M <= 4: one wave per output column
//     (gemv_bf16_kernel); the decode driver's straight-line versions are in vy_decode.hip;
//   * both operands are K-contiguous ("NT"), so both MFMA fragments are 16-byte row reads
//     (ds_read_b128) from a [rows][64] bf16 LDS image; the image is XOR-swizzled on the 16-B
//     chunk index with (row>>1)&7, applied on the *source address* of the
//     global_load_lds_dwordx4 (the LDS destination of an LDS-DMA is lane-linear) and again on
//     the read -- conflict-free for the 16-lane groups of ds_read_b128;
//   * the accumulator is kept TRANSPOSED (A operand = W fragment, B operand = X fragment), so a
//     lane owns one output row m and 4 consecutive columns n per register quad: bias is a
//     per-register value, residual / RoPE / head-split / stores are 8-byte row-local accesses,
//     and rotary pairs (d, d+32) sit in the same lane and register of adjacent 32x32 blocks;
//   * blockIdx is remapped so each XCD (private L2) walks a contiguous run of tiles that share
//     the X row panel.
//
// f32 kernel (the 1e-5 parity path): same structure and epilogues on mfma_f32_32x32x2f32, whose
// result is bit-for-bit a k-ordered fmaf chain.
//
// Reference semantics: nn.Linear call sites listed in include/vyom_hip.h.
#include "vy_common.h"

// (vy_misc.hip) RoPE on q and k in one launch
int vy_rope_qk(void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl, int hq, void* k, int64_t k_sb, int64_t k_sh,
               int64_t k_sl, int hk, const float* cos_tab, const float* sin_tab, int64_t pos0, int64_t B, int64_t L, int dh,
               int dtype, hipStream_t st);

#include <stdlib.h>
#include <mutex>

namespace {

// ------------------------------------------------------------------------------------------
// epilogues.  A "quad" is 4 consecutive output columns n0..n0+3 of one row m, held by one lane.
// ------------------------------------------------------------------------------------------

template <typename T> struct Quad;
template <> struct Quad<bf16> {
  static __device__ __forceinline__ void load(const bf16* p, float (&v)[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void store(bf16* p, const float (&v)[4]) {
    bf16x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (bf16)v[e];
    *reinterpret_cast<bf16x4*>(p) = t;
  }
};
template <> struct Quad<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};

template <typename T> __device__ __forceinline__ float round_like(float x);
template <> __device__ __forceinline__ float round_like<bf16>(float x) { return vy_round_bf16(x); }
template <> __device__ __forceinline__ float round_like<float>(float x) { return x; }

// plain epilogue: (+bias) -> [save pre] -> act -> dropout -> (* act'(gradpre)) -> (+residual) -> store
template <typename T>
struct EpiPlain {
  const T* bias;
  const T* residual; int64_t ldr;
  const T* residual2; int64_t ldr2;  // second addend (dgrad: the two residual-path gradients of a layer)
  const T* gradpre;  int64_t ldg;
  T* y;   int64_t ldy;
  T* pre; // same ld as y
  int vec_ok;  // all row strides % 4 == 0 and base pointers 4-element aligned
  VyDrop drop; // dropout on act(x W^T + b) before the residual add (thr == 0: none)
  int wt_store; // large outputs: write-through (sc1) stores from the staged epilogue
  int pre_deriv; // VY_ACT_SAVE_DERIV: `pre` receives act'(x W^T + b) / `gradpre` already holds act' (see include/vyom_hip.h)
};

template <typename T, int ACT, bool GRAD>
__device__ __forceinline__ void epi_plain_quad(const EpiPlain<T>& e, float (&v)[4], int64_t m, int n,
                                               int N) {
  if (n >= N) return;
  const bool full = e.vec_ok && (n + 3 < N);
  if (full) {
    if (e.bias) { float b[4]; Quad<T>::load(e.bias + n, b);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += b[i]; }
    if (e.pre) {
      if (e.pre_deriv) { float dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dv[i] = vy_act_grad<ACT>(round_like<T>(v[i]));
        Quad<T>::store(e.pre + m * e.ldy + n, dv);
      } else Quad<T>::store(e.pre + m * e.ldy + n, v);
    }
    if constexpr (!GRAD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = vy_act_fwd<ACT>(v[i]);
      if (e.drop.thr) {
        uint32_t lots[4];
        vy_drop_lots(e.drop, m, n >> 3, lots);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = vy_drop_keep(e.drop, lots, (n & 7) + i) ? round_like<T>(v[i]) * e.drop.scale : 0.f;
      }
    } else {
      if (e.gradpre) { float g[4]; Quad<T>::load(e.gradpre + m * e.ldg + n, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= e.pre_deriv ? g[i] : vy_act_grad<ACT>(g[i]); }
    }
    if (e.residual) { float r[4]; Quad<T>::load(e.residual + m * e.ldr + n, r);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += r[i]; }
    if (e.residual2) { float r[4]; Quad<T>::load(e.residual2 + m * e.ldr2 + n, r);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += r[i]; }
    Quad<T>::store(e.y + m * e.ldy + n, v);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (n + i >= N) break;
      float x = v[i];
      if (e.bias) x += VyT<T>::ld(e.bias + n + i);
      if (e.pre) VyT<T>::st(e.pre + m * e.ldy + n + i, e.pre_deriv ? vy_act_grad<ACT>(round_like<T>(x)) : x);
      if constexpr (!GRAD) {
        x = vy_act_fwd<ACT>(x);
        if (e.drop.thr) {
          uint32_t lots[4];
          vy_drop_lots(e.drop, m, (n + i) >> 3, lots);
          x = vy_drop_keep(e.drop, lots, (n + i) & 7) ? round_like<T>(x) * e.drop.scale : 0.f;
        }
      }
      else if (e.gradpre) { const float gg = VyT<T>::ld(e.gradpre + m * e.ldg + n + i); x *= e.pre_deriv ? gg : vy_act_grad<ACT>(gg); }
      if (e.residual) x += VyT<T>::ld(e.residual + m * e.ldr + n + i);
      if (e.residual2) x += VyT<T>::ld(e.residual2 + m * e.ldr2 + n + i);
      VyT<T>::st(e.y + m * e.ldy + n + i, x);
    }
  }
}

// QKV epilogue: bias, rotary embedding on the q and k sections, head split + scatter.
template <typename T>
struct EpiQkv {
  const T* bias;
  const float* cos_tab; const float* sin_tab; int64_t pos0;
  T* q; int64_t q_sb, q_sh, q_sl;
  T* k; int64_t k_sb, k_sh, k_sl;
  T* v; int64_t v_sb, v_sh, v_sl;
  int L; int nq; int nkv;  // nq = h*dh, nkv = hk*dh
  int dh;
  int rope;  // 1: fused rotary (requires dh == 64, wave n-tile == one head)
  int vec8;  // dh % 8 == 0 and 16-byte aligned destinations: phase 2 may store 8 elements at once
  const int* pos_dev;  // decode under a hipGraph: token position read on the device (else NULL)
};

// the same for 64-wide heads (the fused-RoPE case): shifts instead of divisions by the runtime head width, and
// the position added once -- this runs once per 16-byte chunk of a phase that is VALU-bound
template <typename T>
__device__ __forceinline__ T* qkv_dest64(const EpiQkv<T>& e, int64_t b, int64_t l, int n, int64_t lkv) {
  const int d = n & 63;
  if (n < e.nq) return e.q + b * e.q_sb + (n >> 6) * e.q_sh + l * e.q_sl + d;
  n -= e.nq;
  if (n < e.nkv) return e.k + b * e.k_sb + (n >> 6) * e.k_sh + lkv * e.k_sl + d;
  n -= e.nkv;
  return e.v + b * e.v_sb + (n >> 6) * e.v_sh + lkv * e.v_sl + d;
}

template <typename T>
__device__ __forceinline__ T* qkv_dest(const EpiQkv<T>& e, int64_t b, int64_t l, int n) {
  // n: column in the packed [q | k | v] output; returns pointer to element (b, head, l, d)
  if (n < e.nq) { const int hd = n / e.dh, d = n - hd * e.dh;
    return e.q + b * e.q_sb + hd * e.q_sh + l * e.q_sl + d; }
  n -= e.nq;
  if (e.pos_dev) l += *e.pos_dev;  // K/V land at the cache slot of the current position
  if (n < e.nkv) { const int hd = n / e.dh, d = n - hd * e.dh;
    return e.k + b * e.k_sb + hd * e.k_sh + l * e.k_sl + d; }
  n -= e.nkv;
  const int hd = n / e.dh, d = n - hd * e.dh;
  return e.v + b * e.v_sb + hd * e.v_sh + l * e.v_sl + d;
}


// lo/hi: the two 32-column blocks of one 64-wide head (d = dlo..dlo+3 and d+32)
template <typename T>
__device__ __forceinline__ void epi_qkv_pair(const EpiQkv<T>& e, float (&lo)[4], float (&hi)[4],
                                             int64_t m, int n_lo, int N) {
  if (n_lo >= N) return;
  const int64_t b = m / e.L, l = m - b * e.L;
  if (e.bias) {
    float b0[4], b1[4];
    Quad<T>::load(e.bias + n_lo, b0); Quad<T>::load(e.bias + n_lo + 32, b1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] += b0[i]; hi[i] += b1[i]; }
  }
  if (e.rope && n_lo < e.nq + e.nkv) {
    // reference: q*cos + rotate_half(q)*sin with cos/sin cast to q.dtype first
    // (VyomAI/layers/positional_embeddings.py:173-181); products are formed on the
    // storage-rounded projection like the reference's separate Linear -> RoPE ops.
    const int d = n_lo & 31;  // head base is a multiple of 64, n_lo is in the low half
    const int64_t pp = (e.pos_dev ? (int64_t)*e.pos_dev : e.pos0) + l;
    const float* cp = e.cos_tab + pp * 32 + d;
    const float* sp = e.sin_tab + pp * 32 + d;
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(cp);
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float c = round_like<T>(c4[i]), s = round_like<T>(s4[i]);
      const float a = round_like<T>(lo[i]), bb = round_like<T>(hi[i]);
      // each product rounded to the storage type before the add, as the reference's
      // (q * cos) + (rotate_half(q) * sin) evaluates it
      lo[i] = round_like<T>(a * c) - round_like<T>(bb * s);
      hi[i] = round_like<T>(bb * c) + round_like<T>(a * s);
    }
  }
  Quad<T>::store(qkv_dest(e, b, l, n_lo), lo);
  Quad<T>::store(qkv_dest(e, b, l, n_lo + 32), hi);
}

template <typename T>
__device__ __forceinline__ void epi_qkv_quad(const EpiQkv<T>& e, float (&v)[4], int64_t m, int n, int N) {
  if (n >= N) return;
  const int64_t b = m / e.L, l = m - b * e.L;
  if (e.bias) { float b0[4]; Quad<T>::load(e.bias + n, b0);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += b0[i]; }
  Quad<T>::store(qkv_dest(e, b, l, n), v);
}

// measurement aid (VY_GEMM_ROT bit 16): workgroup 0 accumulates its shader-clock cycles (s_memtime)
// and 100 MHz ticks (s_memrealtime) here; vy_debug_gemm_clock() reads and clears them
// (tools/gemm_clock.py): cycles / ticks * 100 = the shader clock in MHz the GEMMs really ran at.
__device__ unsigned long long vy_gemm_clk[6];   // cycles, ticks, launches, prologue / k-loop / epilogue cycles
#define VY_CLK_BEGIN(on)                                                         \
  const bool clk_on_ = (on) && blockIdx.x == 0 && threadIdx.x == 0;              \
  unsigned long long clk_c0_ = 0, clk_w0_ = 0, clk_m_ = 0;                       \
  if (clk_on_) { clk_c0_ = clk_m_ = __builtin_readcyclecounter(); clk_w0_ = wall_clock64(); }
#define VY_CLK_MARK(i)                                                           \
  if (clk_on_) {                                                                 \
    const unsigned long long now_ = __builtin_readcyclecounter();                \
    atomicAdd(&vy_gemm_clk[3 + (i)], now_ - clk_m_);                             \
    clk_m_ = now_;                                                               \
  }
#define VY_CLK_END()                                                             \
  if (clk_on_) {                                                                 \
    atomicAdd(&vy_gemm_clk[0], __builtin_readcyclecounter() - clk_c0_);          \
    atomicAdd(&vy_gemm_clk[1], wall_clock64() - clk_w0_);                        \
    atomicAdd(&vy_gemm_clk[2], 1ull);                                            \
  }

// large-M kernel selection knob: VY_GEMM_VARIANT at first use, or vy_debug_set_gemm_variant() (tests and
// same-process A/B timing; not part of include/vyom_hip.h).  -1 = the default selection.
int g_chains = 1;   // vy_set_concurrent_chains: launch chains the caller runs side by side
int g_gemm_variant = -2;
inline bool vy_m16_on() {   // VY_GEMM_M16=0: the two-stage 32 x 32 x 16 kernels instead of the 16 x 16 x 32 defaults
  static const int v = [] { const char* e = getenv("VY_GEMM_M16"); return e ? atoi(e) : 1; }();
  return v != 0;
}
inline int vy_gemm_variant() {
  if (g_gemm_variant == -2) { const char* e = getenv("VY_GEMM_VARIANT"); g_gemm_variant = e ? atoi(e) : -1; }
  return g_gemm_variant;
}

// XCD-aware bijective block remap: blocks b, b+8, ... share an XCD (round-robin dispatch), give
// each XCD a contiguous run of tile ids.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Tile of workgroup `bid`.  Wide launches (>= 8 column tiles): the tile space is cut into 2 column bands x 4 row
// panel groups, one (band, group) per XCD.  With the plain contiguous run an XCD walks ALL column tiles of its
// row panels, i.e. streams the whole weight matrix once per "wave" of 32 tiles -- FFN1 (W = 4.7 MB > the 4 MiB
// L2): 113 MB of weight re-fetch per launch against 30 MB of algorithmic reads (PMC, profiles/r01_gemm_pmc.json).
// Here an XCD keeps ONE band of W (half the matrix) in its L2 for the whole launch and streams its row panels
// past it; every X panel is read by two XCDs instead of one: fetch ~ 2 X + 8 * W/2 = 69 MB instead of 138 MB.
__device__ __forceinline__ void tile_of(int bid, int nwg, int tiles_n, int band_on, int& tile_m, int& tile_n) {
  const int tiles_m = nwg / tiles_n;
  if (band_on && tiles_n >= 8 && (tiles_n & 1) == 0 && (tiles_m & 3) == 0 && tiles_m * tiles_n == nwg) {
    const int xcd = bid & 7, i = bid >> 3;
    const int band = xcd & 1, grp = xcd >> 1;
    const int bw = tiles_n >> 1, pg = tiles_m >> 2;
    tile_m = grp * pg + i / bw;
    tile_n = band * bw + i - (i / bw) * bw;
  } else {
    const int wg = xcd_remap(bid, nwg);
    tile_m = wg / tiles_n;
    tile_n = wg - tile_m * tiles_n;
  }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------
constexpr int BK = 64;            // k elements per stage
constexpr int ROWB = BK * 2;      // bytes per LDS row (128)

// LDS bytes of the staged epilogue: the bf16 tile band (rows padded by 16 B) and, for the QKV epilogue, the
// rotary table of the band's rows (cos | sin, 32 + 32 bf16 = 128 B per row) when both fit in 160 KiB
constexpr int epi_tile_bytes(int BM, int BN, int PASSES) { return (BM / PASSES) * (BN * 2 + 16); }
constexpr bool epi_has_rope_tab(int BM, int BN, int EPI, int PASSES) {
  return EPI == 1 && epi_tile_bytes(BM, BN, PASSES) + (BM / PASSES) * 128 <= 160 * 1024;
}
constexpr int epi_lds_bytes(int BM, int BN, int EPI, int PASSES) {
  return epi_tile_bytes(BM, BN, PASSES) + (epi_has_rope_tab(BM, BN, EPI, PASSES) ? (BM / PASSES) * 128 : 0);
}
constexpr int vy_cmax(int a, int b) { return a > b ? a : b; }

// ---- shared epilogue of the bf16 kernels ------------------------------------------------------
// PASSES > 1: the staged tile does not fit next to a second resident workgroup's LDS, so it goes out in
// PASSES row bands of BM / PASSES rows (each band is owned by whole wave rows: WGM % PASSES == 0).
// the wave's accumulators: 32 x 32 blocks of mfma_f32_32x32x16 (f32x16 each), or -- MF16 -- 16 x 16 blocks of
// mfma_f32_16x16x32 (f32x4 each: a lane owns output row lane & 15 of the block and the 4 consecutive columns
// 4 * (lane >> 4) .. + 3).  Either way a lane holds QUADS of 4 consecutive columns of one row, which is all the
// staged epilogue needs to know.
template <int BM, int BN, int WGM, int WGN, bool MF16> struct AccTile {
  typedef f32x16 T32[BN / (32 * WGN)][BM / (32 * WGM)];
  typedef f32x4 T16[BN / (16 * WGN)][BM / (16 * WGM)];
  typedef std::conditional_t<MF16, T16, T32> type;
};

template <int BM, int BN, int WGM, int WGN, int EPI, int ACT, bool GRAD, int PASSES = 1, bool PF_OK = true, bool MF16 = false>
__device__ __forceinline__ void gemm_epilogue(typename AccTile<BM, BN, WGM, WGN, MF16>::type& acc, char* smem,
                                              int m0, int n0, int M, int N, const EpiPlain<bf16>& ep,
                                              const EpiQkv<bf16>& eq) {
  static_assert(!MF16 || PASSES == 1, "the 16 x 16 accumulator layout is staged in one pass");
  constexpr int NW = WGM * WGN, NT = 64 * NW;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  constexpr int EROW = BN * 2 + 16;
  constexpr int RP = BM / PASSES;   // rows per pass
  static_assert(WGM % PASSES == 0, "a pass must be whole wave rows");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int fr = lane & 31, fh = lane >> 5;
  // Phase 1 (registers -> LDS): a lane owns output row m and, per register quad rg, the 4
  // consecutive columns nb + 8*rg + 4*fh + (0..3): bias / activation are applied there and the
  // bf16 tile is staged in LDS (rows padded by 16 B: 2-way on ds_write_b64 at worst).
  // Phase 2 (LDS -> HBM): the whole workgroup walks the tile in 16-byte chunks, one wave
  // instruction = whole rows, so residual / GELU' / RoPE operands are loaded and the output is
  // stored as full 128-byte lines (a row-per-lane store tail is store-issue bound, ~10x slower).
  constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
  char* et = smem;

  const int my_pass = (wm * 32 * TM) / RP;   // the row band this wave's sub-tile belongs to
  auto stage_quad = [&](int row, int col, const float (&v)[4]) {
    bf16x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (bf16)v[e];
    *reinterpret_cast<bf16x4*>(et + (row - my_pass * RP) * EROW + col * 2) = w;
  };
  // bias of a register quad (depends on the lane half only); loaded at use, not kept live
  const bf16* biasp = EPI == 0 ? ep.bias : eq.bias;
  const bool bias_vec = (reinterpret_cast<uintptr_t>(biasp) & 7) == 0;
  auto bias_quad = [&](int n, float (&bq)[4]) {
    if (biasp && n + 3 < N && bias_vec) {
      const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(biasp + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[e] = (float)b4[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[e] = (biasp && n + e < N) ? (float)biasp[n + e] : 0.f;
    }
  };
  // phase 1 of a pass: every quad of the wave's sub-tile that lies in the pass's row band, + bias, -> LDS
  auto stage_all = [&](int pass) {
    if constexpr (!MF16) {
      if (PASSES == 1 || pass == my_pass) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            float bq[4];
            bias_quad(n0 + wn * 32 * TN + i * 32 + 8 * rg + 4 * fh, bq);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e] + bq[e];
              stage_quad(wm * 32 * TM + j * 32 + fr, wn * 32 * TN + i * 32 + 8 * rg + 4 * fh, v);
            }
          }
      }
    } else {
      constexpr int TM16 = BM / (16 * WGM), TN16 = BN / (16 * WGN);
      const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int i = 0; i < TN16; ++i) {
        float bq[4];
        bias_quad(n0 + wn * 16 * TN16 + 16 * i + 4 * kq, bq);
#pragma unroll
        for (int j = 0; j < TM16; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bq[e];
          stage_quad(wm * 16 * TM16 + 16 * j + r16, wn * 16 * TN16 + 16 * i + 4 * kq, v);
        }
      }
    }
  };

  if constexpr (EPI == 0) {
    // The staged tile holds the PRE-activation (bf16).  The activation is applied HERE, to those bf16
    // values -- what the reference's bf16 Linear -> GELU computes -- and, when the caller wants the
    // pre-activation saved for backward (dual), it is stored from the same chunk: one pass over the
    // tile, no second staging pass, no GELU chains among the accumulator registers.
    // one 16-byte chunk of the staged band.  pf: the chunk's second operand (residual, or the pre-activation of a
    // dgrad) was requested BEFORE the accumulators were staged and is handed in as p8 -- loaded here, inside the
    // loop, every iteration waits a full memory latency for it (the residual epilogue measured 22 k cycles per
    // tile against 12 k without the residual)
    const bf16* pf_ptr = GRAD ? (ep.gradpre ? ep.gradpre : ep.residual) : ep.residual;
    const int64_t pf_ld = GRAD ? (ep.gradpre ? ep.ldg : ep.ldr) : ep.ldr;
    const bool pf_is_gradpre = GRAD && ep.gradpre != nullptr;
    auto do_chunk = [&](bf16* __restrict__ dst, bool dual, int pass, int c, bool pf, const bf16x8& p8) {
        const int row = c / CPR, cc = c - row * CPR;
        const int64_t m = m0 + pass * RP + row;
        const int n = n0 + cc * 8;
        if (m >= M || n >= N) return;
        const bf16x8 sv = *reinterpret_cast<const bf16x8*>(et + row * EROW + cc * 16);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)sv[e];
        if (dual && !ep.pre_deriv) {
          if (ep.vec_ok && n + 8 <= N) {
            *reinterpret_cast<bf16x8*>(ep.pre + m * ep.ldy + n) = sv;
          } else {
            for (int e = 0; e < 8 && n + e < N; ++e) ep.pre[m * ep.ldy + n + e] = sv[e];
          }
        }
        if constexpr (!GRAD && ACT != VY_ACT_NONE) {
          if (dual && ep.pre_deriv) {
            // the derivative is saved instead of the pre-activation: for the erf GELU both come out of one
            // evaluation of Phi and the Gaussian (vy_phi_fast), so the backward epilogue is one multiply per element
            bf16x8 d8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if constexpr (ACT == VY_ACT_GELU_ERF) {
                float cdf, gs;
                vy_phi_fast(v[e], cdf, gs);
                d8[e] = (bf16)(cdf + v[e] * 0.39894228040143267794f * gs);
                v[e] = v[e] * cdf;
              } else {
                d8[e] = (bf16)vy_act_grad_fast<ACT>(v[e]);
                v[e] = vy_act_fwd_fast<ACT>(v[e]);
              }
            }
            if (ep.vec_ok && n + 8 <= N) {
              // read again only by the backward pass, milliseconds later: streamed, so that it does not take the place of
              // the activation written beside it (FFN2's operand) in the Infinity Cache
              __builtin_nontemporal_store(d8, reinterpret_cast<bf16x8*>(ep.pre + m * ep.ldy + n));
            } else {
              for (int e = 0; e < 8 && n + e < N; ++e) ep.pre[m * ep.ldy + n + e] = d8[e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vy_act_fwd_fast<ACT>(v[e]);
          }
        }
        if constexpr (!GRAD) {
          if (ep.drop.thr) {   // n is a multiple of 8: one Philox call covers the chunk
            uint32_t lots[4];
            vy_drop_lots(ep.drop, m, n >> 3, lots);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vy_drop_keep(ep.drop, lots, e) ? v[e] * ep.drop.scale : 0.f;
          }
        }
        if (ep.vec_ok && n + 8 <= N) {
          if constexpr (GRAD) {
            if (ep.gradpre) {
              const bf16x8 g = pf ? p8 : *reinterpret_cast<const bf16x8*>(ep.gradpre + m * ep.ldg + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= ep.pre_deriv ? (float)g[e] : vy_act_grad_fast<ACT>((float)g[e]);
            }
          }
          if (ep.residual) {
            const bf16x8 r = (pf && !pf_is_gradpre) ? p8 : *reinterpret_cast<const bf16x8*>(ep.residual + m * ep.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
          }
          if (ep.residual2) {
            const bf16x8 r = *reinterpret_cast<const bf16x8*>(ep.residual2 + m * ep.ldr2 + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          if (ep.wt_store) {
            // write-through store (sc1): the line does not stay in this XCD's L2, where a 25-100 MB output stream
            // would evict the operand panels the XCD's other tiles are reading (MI355X_MICROARCH.md, store flavours)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + m * ep.ldy + n), "v"(o) : "memory");
          } else {
            *reinterpret_cast<bf16x8*>(dst + m * ep.ldy + n) = o;
          }
        } else {
          for (int e = 0; e < 8 && n + e < N; ++e) {
            float x = v[e];
            if constexpr (GRAD) {
              if (ep.gradpre) { const float gg = (float)ep.gradpre[m * ep.ldg + n + e]; x *= ep.pre_deriv ? gg : vy_act_grad_fast<ACT>(gg); }
            }
            if (ep.residual) x += (float)ep.residual[m * ep.ldr + n + e];
            if (ep.residual2) x += (float)ep.residual2[m * ep.ldr2 + n + e];
            dst[m * ep.ldy + n + e] = (bf16)x;
          }
        }
    };
    constexpr bool PFK = PF_OK && PASSES == 1 && BM * BN == 256 * 192 && (RP * CPR) % NT == 0;   // kernels that prefetch (the 256 x 256 tile has no registers to spare: +5 % on FFN1 with them reserved)
    constexpr int ITERS = PFK ? (RP * CPR) / NT : 1;
    bf16x8 pre8[ITERS];
    const bool pf_on = PFK && pf_ptr != nullptr && ep.vec_ok && m0 + BM <= M && n0 + BN <= N;   // workgroup-uniform
    if constexpr (PFK) {
      if (pf_on) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int c = tid + it * NT;
          const int row = c / CPR, cc = c - row * CPR;
          pre8[it] = *reinterpret_cast<const bf16x8*>(pf_ptr + (int64_t)(m0 + row) * pf_ld + n0 + cc * 8);
        }
      }
    }
    auto flush_plain = [&](bf16* __restrict__ dst, bool final_pass, bool dual, int pass) {
      (void)final_pass;
      if constexpr (PFK) {
        if (pf_on) {
#pragma unroll
          for (int it = 0; it < ITERS; ++it) do_chunk(dst, dual, pass, tid + it * NT, true, pre8[it]);
          return;
        }
      }
      const bf16x8 none{};
      for (int c = tid; c < RP * CPR; c += NT) do_chunk(dst, dual, pass, c, false, none);
    };
    const bool dual = !GRAD && ep.pre != nullptr;   // (the GRAD path never sets ep.pre)
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
      stage_all(pass);
      __syncthreads();
      flush_plain(ep.y, true, dual, pass);
      if (pass + 1 < PASSES) __syncthreads();
    }
  } else {
    // QKV: bias in registers; RoPE and the head-split scatter in phase 2, where a rotary pair
    // (d, d+32) is two 16-byte chunks 64 bytes apart in the same staged row.  The staged values
    // are the bf16-rounded projections, so the rotation reproduces the reference's op order
    // (Linear output in q.dtype, then q*cos + rotate_half(q)*sin with every op rounded:
    // VyomAI/layers/positional_embeddings.py:173-181) exactly.
    // The rotary factors of the band's rows go through LDS: cos | sin of one position, rounded to bf16 once, are
    // 128 bytes next to the staged tile, filled by 2 x 16-byte loads per thread and pass while the accumulators
    // are being staged -- read per chunk from the fp32 tables they were 4 more loads per 16-byte store in a phase
    // that is bound by the CU's vector-memory path (24 k cycles against 13 k for the same tile without RoPE).
    constexpr bool TAB = epi_has_rope_tab(BM, BN, 1, PASSES);
    char* rt = smem + RP * EROW;   // rotary table of the band: [RP][64] bf16
    const bool use_tab = TAB && eq.rope && n0 < eq.nq + eq.nkv;
    // row -> (batch, position) without 64-bit divisions: the tile's first row once, then 32-bit arithmetic
    const unsigned Lu = (unsigned)eq.L;
    const unsigned b0 = (unsigned)m0 / Lu, l0 = (unsigned)m0 - b0 * Lu;
    const int64_t pdev = eq.pos_dev ? (int64_t)*eq.pos_dev : 0;
    const int64_t pbase = eq.pos_dev ? pdev : eq.pos0;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
    constexpr int TIT = (RP * 8 + NT - 1) / NT;   // table chunks (8 bf16) per thread
    f32x4 tv[TAB ? TIT : 1][2];
    if constexpr (TAB) {
      if (use_tab) {
#pragma unroll
        for (int it = 0; it < TIT; ++it) {
          const int idx = tid + it * NT;
          const int row = idx >> 3, ch = idx & 7;
          if (RP * 8 % NT == 0 || idx < RP * 8) {
            const unsigned t = l0 + (unsigned)(pass * RP + row);
            const unsigned l = t - (t / Lu) * Lu;
            const float* src = (ch < 4 ? eq.cos_tab : eq.sin_tab) + (pbase + l) * 32 + (ch & 3) * 8;
            tv[it][0] = *reinterpret_cast<const f32x4*>(src);
            tv[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
          }
        }
      }
    }
    stage_all(pass);
    if constexpr (TAB) {
      if (use_tab) {
#pragma unroll
        for (int it = 0; it < TIT; ++it) {
          const int idx = tid + it * NT;
          if (RP * 8 % NT == 0 || idx < RP * 8) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16)tv[it][0][e]; o[4 + e] = (bf16)tv[it][1][e]; }
            *reinterpret_cast<bf16x8*>(rt + idx * 16) = o;
          }
        }
      }
    }
    __syncthreads();
    for (int c = tid; c < RP * CPR; c += NT) {
      const int row = c / CPR, cc = c - row * CPR;
      const int64_t m = m0 + pass * RP + row;
      const int n = n0 + cc * 8;
      if (m >= M || n >= N) continue;
      const unsigned t = l0 + (unsigned)(pass * RP + row);
      const unsigned tq = t / Lu;
      const int64_t b = b0 + tq, l = t - tq * Lu;
      bf16x8 sv = *reinterpret_cast<const bf16x8*>(et + row * EROW + cc * 16);
      if (eq.rope && n < eq.nq + eq.nkv) {
        // dh == 64 and BN % 64 == 0: the partner chunk (d ^ 32) is in this tile row
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(et + row * EROW + (cc ^ 4) * 16);
        const int d = n & 31;
        const bool hi = (n & 32) != 0;
        float cs[8], sn[8];
        if (TAB && use_tab) {
          const bf16x8 c8 = *reinterpret_cast<const bf16x8*>(rt + row * 128 + (d >> 3) * 16);
          const bf16x8 s8 = *reinterpret_cast<const bf16x8*>(rt + row * 128 + 64 + (d >> 3) * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) { cs[e] = (float)c8[e]; sn[e] = (float)s8[e]; }
        } else {
          const float* cp = eq.cos_tab + (pbase + l) * 32 + d;
          const float* sp = eq.sin_tab + (pbase + l) * 32 + d;
#pragma unroll
          for (int e = 0; e < 8; ++e) { cs[e] = vy_round_bf16(cp[e]); sn[e] = vy_round_bf16(sp[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = (float)sv[e], o = (float)pv[e];
          const float t1 = vy_round_bf16(a * cs[e]);
          const float t2 = vy_round_bf16((hi ? o : -o) * sn[e]);
          sv[e] = (bf16)(t1 + t2);
        }
      }
      if (eq.vec8 && eq.dh == 64) {
        *reinterpret_cast<bf16x8*>(qkv_dest64(eq, b, l, n, l + pdev)) = sv;
      } else if (eq.vec8) {
        *reinterpret_cast<bf16x8*>(qkv_dest(eq, b, l, n)) = sv;
      } else {
        for (int e = 0; e < 8 && n + e < N; ++e) *qkv_dest(eq, b, l, n + e) = sv[e];
      }
    }
    if (pass + 1 < PASSES) __syncthreads();
    }
  }
}

// tile BM x BN x 64 with WGM x WGN waves, each owning a (BM/WGM) x (BN/WGN) sub-tile as
// TN x TM blocks of 32x32.  Shapes in use:
//   256 x 192, 4x2 waves (512 threads, 1 workgroup per CU, 112 KiB LDS): the training shapes.  Loads
//       per FLOP are (BM+BN)/(BM*BN) = 9.1 B/kFLOP -- a 128^2 tile needs 15.6 and is bound by the
//       ~70 GB/s a CU can pull from L2, not by MFMA; 192 columns make N in {768, 2304, 3072} with
//       M = 16384 give 256 / 768 / 1024 tiles, whole multiples of the 256 CUs;
//   128 x 128, 2x2 waves (2 workgroups per CU): mid-size M;   32 x 128, 1x4 waves: skinny M.
template <int BM, int BN, int WGM, int WGN, int EPI, int ACT, bool GRAD>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_bf16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, int tiles_n, EpiPlain<bf16> ep, EpiQkv<bf16> eq, int rot_on) {
  constexpr int NW = WGM * WGN, NT = 64 * NW;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  static_assert(TM * 32 * WGM == BM && TN * 32 * WGN == BN, "tile / wave layout mismatch");
  VY_CLK_BEGIN(rot_on & 16)
  constexpr int PX = BM / 8, PW = BN / 8;               // 1-KiB LDS-DMA pieces (8 rows) per stage
  constexpr int GX = (PX + NW - 1) / NW, GW = (PW + NW - 1) / NW;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int EROW = BN * 2 + 16;                     // epilogue tile row (padded)
  constexpr int LDS_BYTES = vy_cmax(2 * STAGE, epi_lds_bytes(BM, BN, EPI, 1));
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  int tile_m, tile_n;
  tile_of(blockIdx.x, gridDim.x, tiles_n, !(rot_on & 32), tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-lane LDS-DMA source pointers (row clamped for M/N tails, chunk pre-swizzled) ----
  const int lrow = lane >> 3, slot = lane & 7;
  const bf16* xsrc[GX]; int xk[GX];
  const bf16* wsrc[GW]; int wk[GW];
#pragma unroll
  for (int t = 0; t < GX; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gm = m0 + R; gm = gm < M ? gm : M - 1;
    xsrc[t] = X + (int64_t)gm * ldx + g * 8;
    xk[t] = g * 8;
  }
#pragma unroll
  for (int t = 0; t < GW; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gn = n0 + R; gn = gn < N ? gn : N - 1;
    wsrc[t] = W + (int64_t)gn * ldw + g * 8;
    wk[t] = g * 8;
  }
  const bf16* zero = reinterpret_cast<const bf16*>(vy_zero16);
  const bool ktail = (K % BK) != 0;
  const int KT = (K + BK - 1) / BK;

  // optional (VY_GEMM_ROT=1) per-tile rotated k order, an experiment: workgroups that share an operand
  // run in lockstep and request the same not-yet-resident lines at once.  tools/probe/ldsdma_probe
  // shows the L2 merges such requests well (72 GB/s per CU for a 4-way shared stream against 28
  // unshared), and the rotation measured 1-5 % slower -- kept off.
  const int rot = (rot_on & 1) ? (tile_n * 2 + tile_m) % KT : 0;
  auto stage = [&](int kt, int buf) {
    char* xb = smem + buf * STAGE;
    char* wb = xb + BM * ROWB;
    int kr = kt + rot;
    kr = kr >= KT ? kr - KT : kr;
    const int k0 = kr * BK;
#pragma unroll
    for (int t = 0; t < GX; ++t) {
      if (PX % NW == 0 || wave + NW * t < PX) {
        const bf16* s = xsrc[t] + k0;
        if (ktail && k0 + xk[t] >= K) s = zero;
        __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s,
                                         (VY_LDS void*)(xb + (wave + NW * t) * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < GW; ++t) {
      if (PW % NW == 0 || wave + NW * t < PW) {
        const bf16* s = wsrc[t] + k0;
        if (ktail && k0 + wk[t] >= K) s = zero;
        // one row tile (the 32-row decode launches: the vocabulary projection, 77 MB): every weight tile is read by one
        // workgroup once per step -- streamed (aux 2 = nt) so that it does not push the layer weights out of the
        // Infinity Cache (see dec_load_stream, vy_decode.hip)
        constexpr int W_AUX = BM == 32 ? 2 : 0;
        __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s,
                                         (VY_LDS void*)(wb + (wave + NW * t) * 1024), 16, 0, W_AUX);
      }
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets: row = base32 + (lane&31); chunk = 2*ks + (lane>>5), swizzle (row>>1)&7
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  const int xrow_off = (wm * 32 * TM + fr) * ROWB;
  const int wrow_off = (wn * 32 * TN + fr) * ROWB;

  // Fragments are double-buffered in registers: the reads of k-step ks+1 are issued before the MFMAs
  // of k-step ks.  The reads are the hidden asm form (vy_common.h): hipcc would retire its own
  // ds_reads with lgkmcnt(0), i.e. wait for the fragments it has JUST requested as well, exposing
  // the LDS latency on every other k-step; here a k-step waits with a counted lgkmcnt for exactly
  // its own TN + TM reads.  One base address per (buffer, k-step); the 32-row fragment index goes
  // into the instruction's offset field.
  unsigned xa[4], wa[4];  // buffer 0; buffer 1 is + STAGE (an add per stage, not a register-array index)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int coff = (((ks * 2 + fh) ^ fsw) << 4);
    xa[ks] = vy_lds_addr(smem) + xrow_off + coff;
    wa[ks] = vy_lds_addr(smem) + BM * ROWB + wrow_off + coff;
  }
  bf16x8 wf[2][TN], xf[2][TM];
  auto read_frags = [&](unsigned wbase, unsigned xbase, bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    vy_static_for<TN>([&](auto i_c) { constexpr int i = decltype(i_c)::value; w_[i] = vy_lds_read128_off<i * 32 * ROWB>(wbase); });
    vy_static_for<TM>([&](auto j_c) { constexpr int j = decltype(j_c)::value; x_[j] = vy_lds_read128_off<j * 32 * ROWB>(xbase); });
  };
  auto tie_frags = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
#pragma unroll
    for (int i = 0; i < TN; ++i) vy_tie(w_[i]);
#pragma unroll
    for (int j = 0; j < TM; ++j) vy_tie(x_[j]);
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(wa[0], xa[0], wf[0], xf[0]);
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) stage(kt + 1, cur ^ 1);
    const unsigned boff = cur * STAGE;
    unsigned xb4[4], wb4[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { xb4[ks] = xa[ks] + boff; wb4[ks] = wa[ks] + boff; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
        read_frags(wb4[ks + 1], xb4[ks + 1], wf[(ks + 1) & 1], xf[(ks + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN + TM) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      tie_frags(wf[ks & 1], xf[ks & 1]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][i], xf[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < KT) read_frags(wa[0] + (STAGE - boff), xa[0] + (STAGE - boff), wf[0], xf[0]);
  }
  // the epilogue reuses the stage buffers: every LDS read above has been retired (last k-step: lgkmcnt(0))
  gemm_epilogue<BM, BN, WGM, WGN, EPI, ACT, GRAD>(acc, smem, m0, n0, M, N, ep, eq);
  VY_CLK_END()
}

// ------------------------------------------------------------------------------------------
// 256 x BN x 64 tile, 8 waves (4 x 2), with the X operand in a THREE-deep LDS ring and W in two
// buffers (3*32 + 2*BN/8 KiB = 144 KiB at BN = 192).  Why: with two whole stages every workgroup
// barrier drains the memory pipe (all waves issue, then all wait), and the pipe is busy about half
// the time; tools/probe/ldsdma_probe shows free-running waves take in twice as much.  A third whole
// stage does not fit in 160 KiB, but X -- 57 % of the bytes and the operand that streams from
// HBM/MALL rather than from L2 -- does: its loads for k-slice kt+2 stay in flight across the
// barrier that ends slice kt, and the barrier wait is a COUNTED vmcnt (the GX youngest LDS-DMA
// instructions of the wave, slice kt+2 of X, may still be outstanding).
//
// On mfma_f32_16x16x32_bf16 (the 32 x 32 x 16 form of this kernel was removed in round 3 -- identical outputs, slower):
// 4 x 6 blocks of 16 x 16 per wave (96 accumulator registers),
// 24 MFMAs per 32-deep k-step, two k-steps per stage.  Same LDS image, same swizzle (conflict-free for this
// fragment shape too: rows lane & 15, chunk 4 ks + (lane >> 4)), same ring, same epilogue (a lane still owns quads of 4
// consecutive columns of one row).  Why: the chip holds a higher clock on this MFMA shape and the loop needs fewer
// cycles per stage (tools/probe/gemm_loop_probe.hip, QUICK=1: 2250-2270 against 2380; MI355X_MICROARCH.md, DVFS
// give-back item 7).  Per output element the k order is unchanged (one chain over k in steps of 32 instead of 16),
// so results can differ from the 32 x 32 x 16 kernel only in the rounding of the fp32 chain.
// ------------------------------------------------------------------------------------------
template <int BN, int EPI, int ACT, bool GRAD>
__global__ __launch_bounds__(512) void gemm_nt_bf16_x3m16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, int tiles_n, EpiPlain<bf16> ep, EpiQkv<bf16> eq, int knob) {
  constexpr int BM = 256, WGM = 4, WGN = 2, NW = 8;
  constexpr int TM = BM / (16 * WGM), TN = BN / (16 * WGN);   // 4, 6
  constexpr int PX = BM / 8, PW = BN / 8;
  constexpr int GX = PX / NW, GW = PW / NW;
  static_assert(GX * NW == PX && GW * NW == PW, "pieces must divide over the waves");
  constexpr int XT = BM * ROWB, WT = BN * ROWB;
  constexpr int WOFF = 3 * XT;
  constexpr int LDS_BYTES = vy_cmax(3 * XT + 2 * WT, epi_lds_bytes(BM, BN, EPI, 1));
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  int tile_m, tile_n;
  tile_of(blockIdx.x, gridDim.x, tiles_n, !(knob & 32), tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = lane >> 3, slot = lane & 7;
  const bf16* xsrc[GX]; int xk[GX];
  const bf16* wsrc[GW]; int wk[GW];
#pragma unroll
  for (int t = 0; t < GX; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gm = m0 + R; gm = gm < M ? gm : M - 1;
    xsrc[t] = X + (int64_t)gm * ldx + g * 8;
    xk[t] = g * 8;
  }
#pragma unroll
  for (int t = 0; t < GW; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gn = n0 + R; gn = gn < N ? gn : N - 1;
    wsrc[t] = W + (int64_t)gn * ldw + g * 8;
    wk[t] = g * 8;
  }
  const bf16* zero = reinterpret_cast<const bf16*>(vy_zero16);
  const bool ktail = (K % BK) != 0;
  const int KT = (K + BK - 1) / BK;
  auto stage_x = [&](int kt) {
    char* xb = smem + (kt % 3) * XT;
    const int k0 = kt * BK;
#pragma unroll
    for (int t = 0; t < GX; ++t) {
      const bf16* s = xsrc[t] + k0;
      if (ktail && k0 + xk[t] >= K) s = zero;
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s, (VY_LDS void*)(xb + (wave + NW * t) * 1024), 16, 0, 0);
    }
  };
  auto stage_w = [&](int kt) {
    char* wb = smem + WOFF + (kt & 1) * WT;
    const int k0 = kt * BK;
#pragma unroll
    for (int t = 0; t < GW; ++t) {
      const bf16* s = wsrc[t] + k0;
      if (ktail && k0 + wk[t] >= K) s = zero;
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s, (VY_LDS void*)(wb + (wave + NW * t) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, kq = lane >> 4, fsw = (r16 >> 1) & 7;
  unsigned xa[2], wa[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int coff = (((ks * 4 + kq) ^ fsw) << 4);
    xa[ks] = vy_lds_addr(smem) + (wm * 16 * TM + r16) * ROWB + coff;
    wa[ks] = vy_lds_addr(smem) + WOFF + (wn * 16 * TN + r16) * ROWB + coff;
  }
  bf16x8 wf[2][TN], xf[2][TM];
  auto read_frags = [&](unsigned wbase, unsigned xbase, bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    vy_static_for<TN>([&](auto i_c) { constexpr int i = decltype(i_c)::value; w_[i] = vy_lds_read128_off<i * 16 * ROWB>(wbase); });
    vy_static_for<TM>([&](auto j_c) { constexpr int j = decltype(j_c)::value; x_[j] = vy_lds_read128_off<j * 16 * ROWB>(xbase); });
  };
  auto tie_frags = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
#pragma unroll
    for (int i = 0; i < TN; ++i) vy_tie(w_[i]);
#pragma unroll
    for (int j = 0; j < TM; ++j) vy_tie(x_[j]);
  };
  auto mma = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_[i], x_[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  stage_x(0);
  stage_w(0);
  if (KT > 1) stage_x(1);
  if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GX) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(wa[0], xa[0], wf[0], xf[0]);
  int xbuf = 0;   // kt % 3
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) stage_w(kt + 1);
    if (kt + 2 < KT) stage_x(kt + 2);
    const unsigned xo = xbuf * XT, wo = (kt & 1) * WT;
    read_frags(wa[1] + wo, xa[1] + xo, wf[1], xf[1]);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN + TM) : "memory");
    tie_frags(wf[0], xf[0]);
    mma(wf[0], xf[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_frags(wf[1], xf[1]);
    mma(wf[1], xf[1]);
    if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GX) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    xbuf = xbuf == 2 ? 0 : xbuf + 1;
    if (kt + 1 < KT) read_frags(wa[0] + (WT - wo), xa[0] + xbuf * XT, wf[0], xf[0]);
  }
  gemm_epilogue<BM, BN, WGM, WGN, EPI, ACT, GRAD, 1, true, true>(acc, smem, m0, n0, M, N, ep, eq);
}

// ------------------------------------------------------------------------------------------
// BM x BN x 64 tiles, two whole stages, on mfma_f32_16x16x32_bf16.  256 x 256 (8 waves, 128 KiB): the wide launches (FFN1,
// the dgrad of FFN2, the vocabulary projection), where the tile's fewer operand bytes per FLOP matter more than a third X
// stage; 4 x 8 blocks of 16 x 16 per wave (128 accumulator registers), 32 MFMAs per k-step, fragments double-buffered.
// 128 x 128 (4 waves, 64 KiB, two workgroups per CU): mid-size M.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int EPI, int ACT, bool GRAD>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_bf16_m16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, int tiles_n, EpiPlain<bf16> ep, EpiQkv<bf16> eq, int knob) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / (16 * WGM), TN = BN / (16 * WGN);   // 256 x 256, 4 x 2 waves: 4, 8;  128 x 128, 2 x 2 waves: 4, 4
  constexpr int PX = BM / 8, PW = BN / 8;
  constexpr int GX = PX / NW, GW = PW / NW;
  static_assert(GX * NW == PX && GW * NW == PW, "pieces must divide over the waves");
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int LDS_BYTES = vy_cmax(2 * STAGE, epi_lds_bytes(BM, BN, EPI, 1));
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  int tile_m, tile_n;
  tile_of(blockIdx.x, gridDim.x, tiles_n, !(knob & 32), tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = lane >> 3, slot = lane & 7;
  const bf16* xsrc[GX]; int xk[GX];
  const bf16* wsrc[GW]; int wk[GW];
#pragma unroll
  for (int t = 0; t < GX; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gm = m0 + R; gm = gm < M ? gm : M - 1;
    xsrc[t] = X + (int64_t)gm * ldx + g * 8;
    xk[t] = g * 8;
  }
#pragma unroll
  for (int t = 0; t < GW; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gn = n0 + R; gn = gn < N ? gn : N - 1;
    wsrc[t] = W + (int64_t)gn * ldw + g * 8;
    wk[t] = g * 8;
  }
  const bf16* zero = reinterpret_cast<const bf16*>(vy_zero16);
  const bool ktail = (K % BK) != 0;
  const int KT = (K + BK - 1) / BK;
  auto stage = [&](int kt, int buf) {
    char* xb = smem + buf * STAGE;
    char* wb = xb + BM * ROWB;
    const int k0 = kt * BK;
#pragma unroll
    for (int t = 0; t < GX; ++t) {
      const bf16* s = xsrc[t] + k0;
      if (ktail && k0 + xk[t] >= K) s = zero;
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s, (VY_LDS void*)(xb + (wave + NW * t) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < GW; ++t) {
      const bf16* s = wsrc[t] + k0;
      if (ktail && k0 + wk[t] >= K) s = zero;
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s, (VY_LDS void*)(wb + (wave + NW * t) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, kq = lane >> 4, fsw = (r16 >> 1) & 7;
  unsigned xa[2], wa[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int coff = (((ks * 4 + kq) ^ fsw) << 4);
    xa[ks] = vy_lds_addr(smem) + (wm * 16 * TM + r16) * ROWB + coff;
    wa[ks] = vy_lds_addr(smem) + BM * ROWB + (wn * 16 * TN + r16) * ROWB + coff;
  }
  bf16x8 wf[2][TN], xf[2][TM];
  auto read_frags = [&](unsigned wbase, unsigned xbase, bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    vy_static_for<TN>([&](auto i_c) { constexpr int i = decltype(i_c)::value; w_[i] = vy_lds_read128_off<i * 16 * ROWB>(wbase); });
    vy_static_for<TM>([&](auto j_c) { constexpr int j = decltype(j_c)::value; x_[j] = vy_lds_read128_off<j * 16 * ROWB>(xbase); });
  };
  auto tie_frags = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
#pragma unroll
    for (int i = 0; i < TN; ++i) vy_tie(w_[i]);
#pragma unroll
    for (int j = 0; j < TM; ++j) vy_tie(x_[j]);
  };
  auto mma = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_[i], x_[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(wa[0], xa[0], wf[0], xf[0]);
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) stage(kt + 1, cur ^ 1);
    const unsigned boff = cur * STAGE;
    read_frags(wa[1] + boff, xa[1] + boff, wf[1], xf[1]);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN + TM) : "memory");
    tie_frags(wf[0], xf[0]);
    mma(wf[0], xf[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_frags(wf[1], xf[1]);
    mma(wf[1], xf[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 1 < KT) read_frags(wa[0] + (STAGE - boff), xa[0] + (STAGE - boff), wf[0], xf[0]);
  }
  gemm_epilogue<BM, BN, WGM, WGN, EPI, ACT, GRAD, 1, true, true>(acc, smem, m0, n0, M, N, ep, eq);
}

// ------------------------------------------------------------------------------------------
// Split-K for mid-size M (32 < M <= ~2304 rows: a 264-row PaliGemma prefill, the 2112 rows of a captioning decoder).
// A 264 x 2048 output is 48 tiles of 128 x 128 -- 48 of 256 CUs -- and with K = 16384 (Gemma's down-projection) each of
// them walks 256 k-stages alone.  Here the K range is cut into S slices: tiles x S workgroups (~two per CU) each run the
// k-loop of gemm_nt_bf16_m16_kernel<128, 128> over their slice and leave the fp32 accumulators in the workspace AS THEY SIT
// IN THE REGISTERS ([slice][tile][wave][block][lane] x f32x4: every wave-instruction stores 1 KiB contiguously, no staging);
// gemm_splitk_finish_kernel -- one workgroup per tile with the same thread geometry -- adds the S slices in slice order
// (deterministic) and runs the ordinary staged epilogue (bias, activation, residual, saved pre-activation, dgrad operands,
// QKV head split / RoPE).  Workgroup -> (slice, tile): the workgroups of one XCD (b mod 8) work on ONE k-slice where S
// divides 8, so an XCD's L2 sees only its K range of X and W (speed only).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void splitk_block_of(int b, int tiles, int S, int& tile, int& slice) {
  if (S <= 8 && (8 % S) == 0 && ((tiles * S) & 7) == 0) {
    const int xcd = b & 7, j = b >> 3, per = 8 / S;       // `per` XCDs share a slice
    slice = xcd % S;
    tile = j * per + xcd / S;
  } else {
    slice = b % S;
    tile = b / S;
  }
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_bf16_m16_splitk_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, int tiles_n, int tiles, int S, int kt_per, float* __restrict__ ws) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / (16 * WGM), TN = BN / (16 * WGN);
  constexpr int PX = BM / 8, PW = BN / 8;
  constexpr int GX = PX / NW, GW = PW / NW;
  static_assert(GX * NW == PX && GW * NW == PW, "pieces must divide over the waves");
  constexpr int STAGE = (BM + BN) * ROWB;
  // a slice is 2-16 stages, each a full L2 / MALL round trip: where three stages fit in the CU's 160 KiB (the 256-row
  // all-rows tile: 3 x 48 KiB) two of them stay in flight behind the one being multiplied (counted vmcnt), so a short slice
  // is ONE round trip deep instead of one per stage; the other tiles keep two buffers (one stage ahead)
  constexpr int NBUF = (3 * STAGE <= 160 * 1024 && BM * BN > 128 * 128) ? 3 : 2;
  constexpr int PIECES = GX + GW;   // LDS-DMA instructions per wave and stage
  __shared__ __attribute__((aligned(16))) char smem[NBUF * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  int tile, slice;
  splitk_block_of(blockIdx.x, tiles, S, tile, slice);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int KT_all = (K + BK - 1) / BK;
  const int kt0 = slice * kt_per;
  const int kt1 = kt0 + kt_per < KT_all ? kt0 + kt_per : KT_all;

  const int lrow = lane >> 3, slot = lane & 7;
  const bf16* xsrc[GX]; int xk[GX];
  const bf16* wsrc[GW]; int wk[GW];
#pragma unroll
  for (int t = 0; t < GX; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gm = m0 + R; gm = gm < M ? gm : M - 1;
    xsrc[t] = X + (int64_t)gm * ldx + g * 8;
    xk[t] = g * 8;
  }
#pragma unroll
  for (int t = 0; t < GW; ++t) {
    const int R = (wave + NW * t) * 8 + lrow;
    const int g = slot ^ ((R >> 1) & 7);
    int gn = n0 + R; gn = gn < N ? gn : N - 1;
    wsrc[t] = W + (int64_t)gn * ldw + g * 8;
    wk[t] = g * 8;
  }
  const bf16* zero = reinterpret_cast<const bf16*>(vy_zero16);
  const bool ktail = (K % BK) != 0;
  auto stage = [&](int kt, int buf) {
    char* xb = smem + buf * STAGE;
    char* wb = xb + BM * ROWB;
    const int k0 = kt * BK;
#pragma unroll
    for (int t = 0; t < GX; ++t) {
      const bf16* s = xsrc[t] + k0;
      if (ktail && k0 + xk[t] >= K) s = zero;
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s, (VY_LDS void*)(xb + (wave + NW * t) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < GW; ++t) {
      const bf16* s = wsrc[t] + k0;
      if (ktail && k0 + wk[t] >= K) s = zero;
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)s, (VY_LDS void*)(wb + (wave + NW * t) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, kq = lane >> 4, fsw = (r16 >> 1) & 7;
  unsigned xa[2], wa[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int coff = (((ks * 4 + kq) ^ fsw) << 4);
    xa[ks] = vy_lds_addr(smem) + (wm * 16 * TM + r16) * ROWB + coff;
    wa[ks] = vy_lds_addr(smem) + BM * ROWB + (wn * 16 * TN + r16) * ROWB + coff;
  }
  bf16x8 wf[2][TN], xf[2][TM];
  auto read_frags = [&](unsigned wbase, unsigned xbase, bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    vy_static_for<TN>([&](auto i_c) { constexpr int i = decltype(i_c)::value; w_[i] = vy_lds_read128_off<i * 16 * ROWB>(wbase); });
    vy_static_for<TM>([&](auto j_c) { constexpr int j = decltype(j_c)::value; x_[j] = vy_lds_read128_off<j * 16 * ROWB>(xbase); });
  };
  auto tie_frags = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
#pragma unroll
    for (int i = 0; i < TN; ++i) vy_tie(w_[i]);
#pragma unroll
    for (int j = 0; j < TM; ++j) vy_tie(x_[j]);
  };
  auto mma = [&](bf16x8 (&w_)[TN], bf16x8 (&x_)[TM]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_[i], x_[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  if (kt0 < kt1) {   // (workgroup-uniform; an empty slice stores zeros)
    const int n = kt1 - kt0;
    stage(kt0, 0);
    if (NBUF >= 3 && n > 1) stage(kt0 + 1, 1);
    for (int i = 0; i < n; ++i) {
      // stage i has landed for this wave (the younger stage, if any, may still be in flight) ...
      if (NBUF >= 3 && i + 1 < n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... and for every wave once the barrier is passed; every wave has also finished reading the buffer of stage i - 1,
      // which the next request overwrites
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (i + NBUF - 1 < n) stage(kt0 + i + NBUF - 1, (i + NBUF - 1) % NBUF);
      const unsigned boff = (unsigned)(i % NBUF) * STAGE;
      read_frags(wa[0] + boff, xa[0] + boff, wf[0], xf[0]);
      read_frags(wa[1] + boff, xa[1] + boff, wf[1], xf[1]);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN + TM) : "memory");
      tie_frags(wf[0], xf[0]);
      mma(wf[0], xf[0]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tie_frags(wf[1], xf[1]);
      mma(wf[1], xf[1]);
    }
  }
  // the accumulators as they sit in the registers: [slice][tile][wave][i][j][lane] f32x4
  f32x4* out = reinterpret_cast<f32x4*>(ws) + (((int64_t)slice * tiles + tile) * NW + wave) * (TN * TM * 64) + lane;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) out[(i * TM + j) * 64] = acc[i][j];
}

// One 64-thread workgroup per 16-ROW BLOCK of a wave's sub-tile (16 x SN outputs, 4 accumulator quads per lane): all S x 4
// partial reads of a lane are requested at once -- ONE round trip to the partials other XCDs left in HBM / MALL, where a
// whole-sub-tile finish walked S dependent trips (11 us per launch) -- and 16 x as many workgroups share the launch.  The
// block goes through the staged epilogue as a tile of its own (rows of SN = 64 bf16 = one 128-byte line each).
template <int BM, int BN, int WGM, int WGN, int EPI, int ACT, bool GRAD>
__global__ __launch_bounds__(64) void gemm_splitk_finish_kernel(
    const float* __restrict__ ws, int M, int N, int tiles_n, int tiles, int S, EpiPlain<bf16> ep, EpiQkv<bf16> eq) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / (16 * WGM), TN = BN / (16 * WGN);
  constexpr int SM = BM / WGM, SN = BN / WGN;   // the wave's sub-tile
  constexpr int SMAX = 16;
  __shared__ __attribute__((aligned(16))) char smem[epi_lds_bytes(16, SN, EPI, 1)];
  const int lane = threadIdx.x;
  const int tile = blockIdx.x / (NW * TM);
  const int rem = blockIdx.x - tile * (NW * TM);
  const int wave = rem / TM, j = rem - wave * TM;
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int m0 = tile_m * BM + wm * SM + j * 16, n0 = tile_n * BN + wn * SN;
  if (m0 >= M || n0 >= N) return;   // (workgroup-uniform)
  const f32x4* in = reinterpret_cast<const f32x4*>(ws) + ((int64_t)tile * NW + wave) * (TN * TM * 64) + j * 64 + lane;
  const int64_t slice_stride = (int64_t)tiles * NW * (TN * TM * 64);
  f32x4 part[SMAX][TN];
#pragma unroll
  for (int sidx = 0; sidx < SMAX; ++sidx)
    if (sidx < S) {
#pragma unroll
      for (int i = 0; i < TN; ++i) part[sidx][i] = in[sidx * slice_stride + (int64_t)i * TM * 64];
    }
  f32x4 acc[TN][1];
#pragma unroll
  for (int i = 0; i < TN; ++i) acc[i][0] = part[0][i];
#pragma unroll
  for (int sidx = 1; sidx < SMAX; ++sidx)
    if (sidx < S) {
#pragma unroll
      for (int i = 0; i < TN; ++i) acc[i][0] = acc[i][0] + part[sidx][i];   // slice order: deterministic
    }
  gemm_epilogue<16, SN, 1, 1, EPI, ACT, GRAD, 1, true, true>(acc, smem, m0, n0, M, N, ep, eq);
}

// Workspaces for the split-K launches, registered per stream by the caller (the library never allocates device memory):
// vy_workspace_set(stream, ptr, bytes).  Launches on a stream without one simply do not split.
struct WsEntry { hipStream_t st; void* p; int64_t bytes; };
WsEntry g_ws[16];
int g_ws_n = 0;
std::mutex g_ws_mu;
inline float* splitk_ws(hipStream_t st, int64_t need) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (int i = 0; i < g_ws_n; ++i)
    if (g_ws[i].st == st) return g_ws[i].bytes >= need ? (float*)g_ws[i].p : nullptr;
  return nullptr;
}
// slices for a tile grid: `target` workgroups in all (two per CU for the 128 x 128 tiles, one for the all-rows tiles), at least
// two 64-deep stages per slice
inline int splitk_slices(int64_t tiles, int KT, int target) {
  static const int on = [] { const char* e = getenv("VY_SPLITK"); return e ? atoi(e) : 1; }();
  // (measured, same box: 2 stages per slice at least -- configs[4] prefill 7.09 -> 6.88 ms, configs[3] step 19.1 -> 18.5 ms against 4)
  static const int min_stages = [] { const char* e = getenv("VY_SPLITK_MIN_STAGES"); return e ? atoi(e) : 2; }();
  if (!on || tiles * 3 >= target * 2 || g_chains != 1) return 1;
  int S = (int)((target + tiles / 2) / tiles);
  if (S > KT / min_stages) S = KT / min_stages;
  if (S > 16) S = 16;
  if (S < 2) return 1;
  int p2 = 2;
  while (p2 * 2 <= S) p2 *= 2;   // a power of two: one slice per group of XCDs (splitk_block_of)
  S = p2;
  const int per = (KT + S - 1) / S;
  return (KT + per - 1) / per;   // no empty slice
}

// mid-size M launch on BM x BN tiles: split over K when the tile grid leaves CUs idle and the stream has a workspace
template <int BM, int BN, int WGM, int WGN, int EPI, int ACT, bool GRAD>
void launch_mid(const bf16* X, int64_t ldx, const bf16* W, int64_t ldw, int64_t M, int64_t N, int64_t K,
                const EpiPlain<bf16>& ep, const EpiQkv<bf16>& eq, hipStream_t st, int target, int rot) {
  constexpr int NT = 64 * WGM * WGN;
  const int tn = (int)vy_cdiv(N, BN), tm = (int)vy_cdiv(M, BM);
  const int KT = (int)vy_cdiv(K, 64);
  const int tiles = tm * tn;
  const int S = splitk_slices(tiles, KT, target);
  float* ws = S > 1 ? splitk_ws(st, (int64_t)S * tiles * BM * BN * 4) : nullptr;
  if (ws) {
    const int per = (KT + S - 1) / S;
    hipLaunchKernelGGL((gemm_nt_bf16_m16_splitk_kernel<BM, BN, WGM, WGN>), dim3(tiles * S), dim3(NT), 0, st, X, ldx, W, ldw,
                       (int)M, (int)N, (int)K, tn, tiles, S, per, ws);
    constexpr int TMB = BM / (16 * WGM);
    hipLaunchKernelGGL((gemm_splitk_finish_kernel<BM, BN, WGM, WGN, EPI, ACT, GRAD>), dim3(tiles * WGM * WGN * TMB), dim3(64), 0,
                       st, ws, (int)M, (int)N, tn, tiles, S, ep, eq);
  } else {
    hipLaunchKernelGGL((gemm_nt_bf16_m16_kernel<BM, BN, WGM, WGN, EPI, ACT, GRAD>), dim3(tiles), dim3(NT), 0,
                       st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn, ep, eq, rot);
  }
}

// ------------------------------------------------------------------------------------------
// skinny kernel for decode (M <= 32 rows): HBM-bound weight streaming, no LDS staging.
// One workgroup = 32 output columns; its 4 waves split K (k-step s goes to wave s % 4, so the
// four waves together consume whole 128-byte lines of every weight row), W and X fragments are
// loaded straight into MFMA operand registers (each weight byte is used once per workgroup),
// and the four partial 32x32 accumulators are summed through LDS.  With fused RoPE a workgroup
// owns columns {d, d+32} of 16 rotary pairs of one head, so a pair sits in registers r and r+8
// of one lane.
// ------------------------------------------------------------------------------------------
template <int EPI, int ACT>
__global__ __launch_bounds__(256) void gemm_skinny_bf16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, EpiPlain<bf16> ep, EpiQkv<bf16> eq) {
  __shared__ float red[3][16][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int nb = blockIdx.x;
  const bool rope_map = (EPI == 1) && eq.rope;
  int col = nb * 32 + fr;
  if (rope_map) col = (nb >> 1) * 64 + 16 * (nb & 1) + (fr < 16 ? fr : 16 + fr);
  const int wr = col < N ? col : N - 1;
  const int mr = fr < M ? fr : M - 1;
  const bf16* wp = W + (int64_t)wr * ldw + fh * 8;
  const bf16* xp = X + (int64_t)mr * ldx + fh * 8;
  const int nsteps = K >> 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int s = wave;
  // 12 k-steps of this wave per trip (K = 768 in one trip): 24 independent 16-byte loads are in
  // flight before the first MFMA -- the kernel is a latency-bound HBM stream, not compute
  for (; s + 44 < nsteps; s += 48) {
    bf16x8 a[12], b[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      a[u] = *reinterpret_cast<const bf16x8*>(wp + 16 * (s + 4 * u));
      b[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * (s + 4 * u));
    }
    __builtin_amdgcn_sched_barrier(0);  // all 24 loads are issued before the first MFMA waits
#pragma unroll
    for (int u = 0; u < 12; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
  }
  for (; s + 12 < nsteps; s += 16) {
    bf16x8 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const bf16x8*>(wp + 16 * (s + 4 * u));
      b[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * (s + 4 * u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
  }
  for (; s < nsteps; s += 4) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wp + 16 * s);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(xp + 16 * s);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
  const int64_t m = fr;
  if (m >= M) return;
  if constexpr (EPI == 0) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float v[4] = {acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]};
      epi_plain_quad<bf16, ACT, false>(ep, v, m, nb * 32 + 8 * rg + 4 * fh, N);
    }
  } else {
    if (rope_map) {
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        float lo[4] = {acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]};
        float hi[4] = {acc[4 * rg + 8], acc[4 * rg + 9], acc[4 * rg + 10], acc[4 * rg + 11]};
        epi_qkv_pair<bf16>(eq, lo, hi, m, (nb >> 1) * 64 + 16 * (nb & 1) + 8 * rg + 4 * fh, N);
      }
    } else {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4] = {acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]};
        epi_qkv_quad<bf16>(eq, v, m, nb * 32 + 8 * rg + 4 * fh, N);
      }
    }
  }
}

// ---- split-K weight streaming for M <= 32 and narrow N (decode: N = hidden size) --------------
// A skinny GEMM has N/32 workgroups: 24 for N = 768, on a 256-CU chip.  Here grid.y slices of K
// each produce an fp32 partial tile part[ks][32][N] (plain stores: deterministic, no atomics); the
// consumer (splitk_finish_ln_kernel) adds them in slice order with bias and residual.
__global__ __launch_bounds__(256) void gemm_skinny_splitk_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, int steps_per_wg, float* __restrict__ part) {
  __shared__ float red[3][16][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int nb = blockIdx.x, ks = blockIdx.y;
  const int col = nb * 32 + fr;
  const int wr = col < N ? col : N - 1;
  const int mr = fr < M ? fr : M - 1;
  const bf16* wp = W + (int64_t)wr * ldw + fh * 8;
  const bf16* xp = X + (int64_t)mr * ldx + fh * 8;
  const int nsteps = K >> 4;
  const int s_end = min(nsteps, (ks + 1) * steps_per_wg);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int s = ks * steps_per_wg + wave;
  for (; s + 20 < s_end; s += 24) {  // 6 k-steps of this wave per trip: 12 loads in flight
    bf16x8 a[6], b[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      a[u] = *reinterpret_cast<const bf16x8*>(wp + 16 * (s + 4 * u));
      b[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * (s + 4 * u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
  }
  for (; s + 4 < s_end; s += 8) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      a[u] = *reinterpret_cast<const bf16x8*>(wp + 16 * (s + 4 * u));
      b[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * (s + 4 * u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
  }
  for (; s < s_end; s += 4) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wp + 16 * s);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(xp + 16 * s);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
  if (fr >= M) return;
  float* dst = part + ((int64_t)ks * 32 + fr) * N + nb * 32 + 4 * fh;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const f32x4 v = {acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]};
    *reinterpret_cast<f32x4*>(dst + 8 * rg) = v;
  }
}

// y = LayerNorm(bf16(sum_ks part[ks] + bias + residual)): the epilogue of the split-K GEMM above fused
// with the LayerNorm that follows it in the block (AttentionSelfOutput / FeedForward).  One wave per
// row, the same arithmetic as gemm epilogue + layernorm_fwd_kernel (sum rounded to bf16 first).
template <int CH>
__global__ __launch_bounds__(256) void splitk_finish_ln_kernel(const float* __restrict__ part, int ksplit, int M,
                                                               int N, const bf16* __restrict__ bias,
                                                               const bf16* __restrict__ residual, int64_t ldr,
                                                               const bf16* __restrict__ gamma,
                                                               const bf16* __restrict__ beta, bf16* __restrict__ y,
                                                               int64_t ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = N / 8;
  float v[CH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = 0.f;
      // all (<= 8) partial tiles are requested before the first add: one memory latency, not ksplit
      f32x4 p0[8], p1[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float* pp = part + ((int64_t)(k < ksplit ? k : 0) * 32 + row) * N + ch * 8;
        p0[k] = *reinterpret_cast<const f32x4*>(pp);
        p1[k] = *reinterpret_cast<const f32x4*>(pp + 4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < ksplit) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[e] += p0[k][e]; a[4 + e] += p1[k][e]; }
        }
      }
      if (bias) {
        const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(bias + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += (float)b8[e];
      }
      if (residual) {
        const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(residual + (int64_t)row * ldr + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += (float)r8[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[c][e] = vy_round_bf16(a[e]); s += v[c][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  const float mean = vy_wave_sum(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
    }
  }
  const float var = vy_wave_sum(q) / (float)N;
  const float rstd = rsqrtf(var + eps);
  const float rstd_r = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gamma + ch * 8);
      const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(beta + ch * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((v[c][e] - mean) * rstd_r * (float)g8[e] + (float)b8[e]);
      *reinterpret_cast<bf16x8*>(y + (int64_t)row * ldy + ch * 8) = o;
    }
  }
}

// (A one-launch form of the split-K GEMM + combine + LayerNorm -- the workgroup that delivers the last K-slice
// of a tile combines it, the one that finishes the last tile normalises the rows; sc1 write-through hand-offs,
// no spinning -- was built, was bit-identical, and made the decode step 1.5x SLOWER (1.03 vs 0.69 ms): two
// dependent cross-CU hand-offs inside a launch (drain write-through stores, ticket, sc1 loads from beyond L2)
// cost ~15 us against ~1.7 us for a kernel boundary.  On this chip a chain of short dependent phases is
// cheapest as launches in a graph, and each launch is made as short as possible instead.)

// y = LayerNorm(bf16(sum_ks part[ks] + bias + residual)), ONE WORKGROUP PER ROW: every thread requests its
// 4 columns of all partial tiles, bias, residual, gamma and beta at once (one memory round trip), then two
// block reductions.  (splitk_finish_ln_kernel walks a row with one wave: 6.4 us per launch against ~2.5.)
template <int QPT>
__global__ __launch_bounds__(256) void splitk_finish_ln_row_kernel(const float* __restrict__ part, int ksplit, int M,
                                                                   int N, const bf16* __restrict__ bias,
                                                                   const bf16* __restrict__ residual, int64_t ldr,
                                                                   const bf16* __restrict__ gamma,
                                                                   const bf16* __restrict__ beta, bf16* __restrict__ y,
                                                                   int64_t ldy, float eps) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  const int nq = N >> 2;
  f32x4 pk[QPT][8];
  bf16x4 b4[QPT], r4[QPT], g4[QPT], be4[QPT];
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + 256 * i;
    const int n = (q < nq ? q : 0) * 4;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      pk[i][k] = *reinterpret_cast<const f32x4*>(part + ((int64_t)(k < ksplit ? k : 0) * 32 + row) * N + n);
    if (bias) b4[i] = *reinterpret_cast<const bf16x4*>(bias + n);
    if (residual) r4[i] = *reinterpret_cast<const bf16x4*>(residual + (int64_t)row * ldr + n);
    g4[i] = *reinterpret_cast<const bf16x4*>(gamma + n);
    be4[i] = *reinterpret_cast<const bf16x4*>(beta + n);
  }
  float v[QPT][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const bool ok = tid + 256 * i < nq;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < ksplit) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += pk[i][k][e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (bias) a[e] += (float)b4[i][e];
      if (residual) a[e] += (float)r4[i][e];
      v[i][e] = ok ? vy_round_bf16(a[e]) : 0.f;
      s += v[i][e];
    }
  }
  s = vy_wave_sum(s);
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
  qv = vy_wave_sum(qv);
  if (lane == 0) red[1][wave] = qv;
  __syncthreads();
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
      *reinterpret_cast<bf16x4*>(y + (int64_t)row * ldy + q * 4) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// 16 output columns per workgroup (M <= 32): the decode projections with N/32 < ~200 workgroups leave most
// of the 256 CUs without a weight stream, and ONE CU pulls only ~25-40 GB/s (what its L1 keeps in flight
// over a memory latency), so such a launch runs at a quarter of the chip's bandwidth.  Halving the tile
// doubles the CUs that stream: mfma_f32_16x16x32_bf16, A = 16 weight rows x 32 k, B = 16 batch rows x 32 k
// (two row blocks), K split over the workgroup's 4 waves, partial sums through LDS.  D[n][m]: a lane owns
// batch row lane & 15 and the 4 consecutive columns 4 * (lane >> 4) .. + 3.  With fused RoPE a workgroup
// takes columns {d0 .. d0+7} and {d0+32 .. d0+39} of one head: a rotary pair sits in lanes l and l ^ 32.
// ------------------------------------------------------------------------------------------
template <int EPI, int ACT>
__global__ __launch_bounds__(256) void gemm_skinny16_bf16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, int M, int N,
    int K, EpiPlain<bf16> ep, EpiQkv<bf16> eq, int dbg) {
  __shared__ float red[3][8][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int nb = blockIdx.x;
  const bool rope_map = (EPI == 1) && eq.rope;
  int col = nb * 16 + r16;
  if (rope_map) col = (nb >> 2) * 64 + (nb & 3) * 8 + (r16 < 8 ? r16 : 24 + r16);
  const int wr = col < N ? col : N - 1;
  const int m0r = r16 < M ? r16 : M - 1, m1r = 16 + r16 < M ? 16 + r16 : M - 1;
  const bf16* wp = W + (int64_t)wr * ldw + kq * 8;
  const bf16* xp0 = X + (int64_t)m0r * ldx + kq * 8;
  const bf16* xp1 = X + (int64_t)m1r * ldx + kq * 8;
  // measurement knobs (VY_SKINNY_DBG; wrong results): 1 = every lane reads X row 0, 2 = every workgroup reads the
  // weight rows of tile 0, 4 = one k-step only, 8 = no epilogue, 16 = return at once
  if (dbg & 16) return;
  if (dbg & 1) { xp0 = X + kq * 8; xp1 = xp0; }
  if (dbg & 2) wp = W + (int64_t)r16 * ldw + kq * 8;
  const bool two = M > 16;   // (wave-uniform) a second block of batch rows exists
  // the epilogue's operands that do not depend on the product (bias, residual, rotary cos / sin) are requested
  // up front, under the weight stream: loaded after the reduction they are one more dependent memory round
  // trip in a kernel that is nothing but round trips (measured: the epilogue was 2 of this kernel's ~5 us)
  const int nq0 = rope_map ? (nb >> 2) * 64 + (nb & 3) * 8 + 4 * (kq & 1) + (kq >= 2 ? 32 : 0) : nb * 16 + 4 * kq;
  const bf16* biasp = EPI == 0 ? ep.bias : eq.bias;
  const bool fast = nq0 + 3 < N && (EPI == 1 || ep.vec_ok);
  bf16x4 bias4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  if (biasp && fast) bias4 = *reinterpret_cast<const bf16x4*>(biasp + nq0);
  bf16x4 res4[2];
  if (EPI == 0 && fast && ep.residual) {
    res4[0] = *reinterpret_cast<const bf16x4*>(ep.residual + (int64_t)m0r * ep.ldr + nq0);
    res4[1] = *reinterpret_cast<const bf16x4*>(ep.residual + (int64_t)m1r * ep.ldr + nq0);
  }
  f32x4 cos4[2], sin4[2];
  if (rope_map) {
    const int64_t pbase = eq.pos_dev ? (int64_t)*eq.pos_dev : eq.pos0;
    const int dcol = (nb & 3) * 8 + 4 * (kq & 1);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int64_t mm = blk == 0 ? m0r : m1r;
      const int64_t pp = pbase + (mm - (mm / eq.L) * eq.L);
      cos4[blk] = *reinterpret_cast<const f32x4*>(eq.cos_tab + pp * 32 + dcol);
      sin4[blk] = *reinterpret_cast<const f32x4*>(eq.sin_tab + pp * 32 + dcol);
    }
  }
  const int nsteps = (dbg & 4) ? 4 : K >> 5;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int s = wave;
  constexpr int U = 6;   // K = 768: the wave's 6 k-steps in one trip, 18 loads in flight per lane
  for (; s + 4 * (U - 1) < nsteps; s += 4 * U) {
    bf16x8 a[U], b0[U], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = *reinterpret_cast<const bf16x8*>(wp + 32 * (s + 4 * u));
      b0[u] = *reinterpret_cast<const bf16x8*>(xp0 + 32 * (s + 4 * u));
      b1[u] = *reinterpret_cast<const bf16x8*>(xp1 + 32 * (s + 4 * u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b0[u], acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b1[u], acc1, 0, 0, 0);
    }
  }
  for (; s < nsteps; s += 4) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wp + 32 * s);
    const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(xp0 + 32 * s);
    const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(xp1 + 32 * s);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc0, 0, 0, 0);
    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc1, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[wave - 1][r][lane] = acc0[r]; red[wave - 1][4 + r][lane] = acc1[r]; }
  }
  __syncthreads();
  if (wave != 0) return;
  if (dbg & 8) { if (acc0[0] == 12345.678f) ep.y[0] = (bf16)acc1[0]; return; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    acc1[r] += red[0][4 + r][lane] + red[1][4 + r][lane] + red[2][4 + r][lane];
  }
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int64_t m = blk * 16 + r16;
    if (m >= M) continue;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = blk == 0 ? acc0[r] : acc1[r];
    if constexpr (EPI == 0) {
      if (fast && !ep.pre && !ep.residual2 && !ep.drop.thr) {   // epi_plain_quad on the prefetched operands
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = vy_act_fwd<ACT>(v[i] + (float)bias4[i]);
        if (ep.residual) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += (float)res4[blk][i];
        }
        Quad<bf16>::store(ep.y + m * ep.ldy + nq0, v);
      } else {
        epi_plain_quad<bf16, ACT, false>(ep, v, m, nb * 16 + 4 * kq, N);
      }
    } else {
      if (!rope_map) {
        epi_qkv_quad<bf16>(eq, v, m, nb * 16 + 4 * kq, N);
      } else {
        // this lane: columns nq0 .. nq0+3 of the packed [q | k | v] row (the high half of the head for kq >= 2);
        // the other half of each rotary pair is in lane ^ 32
        const bool hi_half = kq >= 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += (float)bias4[i];
        float other[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) other[i] = __shfl_xor(v[i], 32, 64);
        const int64_t b = m / eq.L, l = m - b * eq.L;
        if (nq0 < eq.nq + eq.nkv) {
          // reference op order (positional_embeddings.py:173-181), every product rounded to bf16: epi_qkv_pair
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float c = vy_round_bf16(cos4[blk][i]), sn = vy_round_bf16(sin4[blk][i]);
            const float mine = vy_round_bf16(v[i]), oth = vy_round_bf16(other[i]);
            // low half: a*c - b*s;  high half: b*c + a*s   (a = low element, b = high element)
            v[i] = hi_half ? vy_round_bf16(mine * c) + vy_round_bf16(oth * sn)
                           : vy_round_bf16(mine * c) - vy_round_bf16(oth * sn);
          }
        }
        Quad<bf16>::store(qkv_dest(eq, b, l, nq0), v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// f32 MFMA kernel (parity path): 64x64 tile, BK=16, 4 waves of 32x32, mfma_f32_32x32x2f32
// ------------------------------------------------------------------------------------------
constexpr int FBK = 16;
constexpr int FLD = FBK + 1;  // padded LDS row (floats): conflict-free ds_read_b32 column reads

template <int EPI, int ACT, bool GRAD>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw, int M, int N,
    int K, int tiles_n, EpiPlain<float> ep, EpiQkv<float> eq) {
  __shared__ float xs[64 * FLD];
  __shared__ float ws[64 * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  const int lr = tid >> 2, lc = (tid & 3) * 4;  // loader: row 0..63, 4 floats at column lc
  int gm = m0 + lr; gm = gm < M ? gm : M - 1;
  int gn = n0 + lr; gn = gn < N ? gn : N - 1;
  const float* xp = X + (int64_t)gm * ldx + lc;
  const float* wp = W + (int64_t)gn * ldw + lc;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += FBK) {
    f32x4 xv = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
    if (k0 + lc < K) {  // K % 4 == 0 is required, so a float4 is all-in or all-out
      xv = *reinterpret_cast<const f32x4*>(xp + k0);
      wv = *reinterpret_cast<const f32x4*>(wp + k0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { xs[lr * FLD + lc + e] = xv[e]; ws[lr * FLD + lc + e] = wv[e]; }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FBK / 2; ++kk) {
      const float a = ws[(wn * 32 + fr) * FLD + 2 * kk + fh];
      const float b = xs[(wm * 32 + fr) * FLD + 2 * kk + fh];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int64_t m = m0 + wm * 32 + fr;
  if (m >= M) return;
  if constexpr (EPI == 0) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float v[4] = {acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]};
      epi_plain_quad<float, ACT, GRAD>(ep, v, m, n0 + wn * 32 + 8 * rg + 4 * fh, N);
    }
  } else {
    // rotary pairs (d, d+32) live in the sibling wave here, so the f32 path never fuses RoPE:
    // the host applies vy_rope_fwd afterwards.
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float v[4] = {acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]};
      epi_qkv_quad<float>(eq, v, m, n0 + wn * 32 + 8 * rg + 4 * fh, N);
    }
  }
}

// ------------------------------------------------------------------------------------------
// M <= 4 rows (single-sequence decode): a matrix-vector product is a pure weight stream.  One wave per
// output column reads its weight row in 16-byte pieces -- a wave-instruction = 1 KiB contiguous, eight in
// flight per lane -- against X from L2, v_dot2_f32_bf16 into one fp32 sum per row, wave reduction, scalar
// epilogue.  N/4 workgroups whatever the shape: the 32-column MFMA kernel above has N/32, i.e. 64 for
// a 16384 -> 2048 down-projection, a quarter of the chip.
// ------------------------------------------------------------------------------------------
// NORM: the input rows are RMS-normalised on the way in (x * rsqrt(mean x^2 + eps) * (1 + w), rounded to bf16:
// GemmaRMSNorm, notebook cell 11) -- every wave redoes the row statistics from the chunks it reads anyway
// (the same lane -> chunk map and the same arithmetic as rmsnorm_fwd_kernel: bit-identical), which removes one
// launch per projection from a single-sequence decode step.  GATED: W is the packed [gate; up] matrix
// ([2 I, K]); the wave of output column n reads rows n and I + n and stores act(bf16 gate) * bf16 up
// (GemmaMLP, cell 11; the arithmetic of gated_act_kernel on the two rounded projections).
// NORM: 0 none; 1 RMSNorm of the input folded in (row statistics + per-element normalisation in every wave); 2 the
// weights come pre-multiplied by the norm's (1 + w) along K, so the wave only accumulates sum x^2 beside the product
// and scales its result by rsqrt(mean x^2 + eps) -- no RMSNorm launch, no per-element work.
// EPI == 1 with eq.rope: rotary embedding fused (single-token rows): a workgroup's four waves take the columns
// {d, d+1, d+dh/2, d+1+dh/2} of one head and swap partners through LDS (rope2_kernel's arithmetic).
template <int EPI, int ACT, int MR, int NORM = 0, bool GATED = false>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const bf16* __restrict__ X, int64_t ldx,
                                                        const bf16* __restrict__ W, int64_t ldw, int M, int N, int K,
                                                        EpiPlain<bf16> ep, EpiQkv<bf16> eq,
                                                        const bf16* __restrict__ norm_w, float norm_eps) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  int n = blockIdx.x * 4 + wv;
  const bool rope_on = EPI == 1 && eq.rope;   // (the launcher guarantees N % 4 == 0 and dh % 4 == 0 then)
  if (rope_on) {
    const int per_head = eq.dh >> 2;
    const int hd = (int)blockIdx.x / per_head, pi = (int)blockIdx.x - hd * per_head;
    n = hd * eq.dh + 2 * pi + (wv & 1) + (eq.dh >> 1) * (wv >> 1);
  }
  if (n >= N) return;   // wave-uniform
  const bf16* w = W + (int64_t)n * ldw;
  const bf16* w2 = GATED ? W + (int64_t)(N + n) * ldw : nullptr;
  float acc[MR], acc2[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { acc[m] = 0.f; acc2[m] = 0.f; }
  float rstd[MR], sq[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { rstd[m] = 1.f; sq[m] = 0.f; }
  if constexpr (NORM == 1) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float q = 0.f;
      if (m < M) {
        for (int kk = lane * 8; kk < K; kk += 512) {
          const bf16x8 xv = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * ldx + kk);
#pragma unroll
          for (int e = 0; e < 8; ++e) q += (float)xv[e] * (float)xv[e];
        }
      }
      const float ms = vy_wave_sum(q) / (float)K + norm_eps;
      float r = rsqrtf(ms);
      rstd[m] = r * (1.5f - 0.5f * ms * r * r);
    }
  }
  constexpr int U = 8;
  for (int k0 = lane * 8; k0 < K; k0 += U * 512) {
    bf16x8 wv[U], wv2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + u * 512;
      // (each weight byte once per step: streamed -- MI355X_MICROARCH.md nt-weights; see dec_load_stream, vy_decode.hip)
      wv[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w + (kk < K ? kk : 0)));
      if constexpr (GATED) wv2[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w2 + (kk < K ? kk : 0)));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + u * 512;
      if (kk < K) {
        union { bf16x8 v; bf16x2_t h[4]; } a, a2, b;
        a.v = wv[u];
        if constexpr (GATED) a2.v = wv2[u];
        bf16x8 g8;
        if constexpr (NORM == 1) g8 = *reinterpret_cast<const bf16x8*>(norm_w + kk);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          if (m < M) {
            b.v = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * ldx + kk);
            if constexpr (NORM == 1) {
#pragma unroll
              for (int e = 0; e < 8; ++e) b.v[e] = (bf16)((float)b.v[e] * rstd[m] * (1.0f + (float)g8[e]));
            }
            if constexpr (NORM == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) sq[m] = __builtin_amdgcn_fdot2_f32_bf16(b.h[e], b.h[e], sq[m], false);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[m] = __builtin_amdgcn_fdot2_f32_bf16(a.h[e], b.h[e], acc[m], false);
              if constexpr (GATED) acc2[m] = __builtin_amdgcn_fdot2_f32_bf16(a2.h[e], b.h[e], acc2[m], false);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc[m] = vy_wave_sum(acc[m]);
    if constexpr (GATED) acc2[m] = vy_wave_sum(acc2[m]);
    if constexpr (NORM == 2) {
      const float ms = vy_wave_sum(sq[m]) / (float)K + norm_eps;
      const float r = rsqrtf(ms);
      const float rr = r * (1.5f - 0.5f * ms * r * r);
      acc[m] *= rr;
      if constexpr (GATED) acc2[m] *= rr;
    }
  }
  __shared__ float rope_x[4][MR];
  if constexpr (EPI == 1) {
    if (rope_on) {   // workgroup-uniform: every wave of the workgroup is here
      if (lane < MR) {
        float x = 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) x = lane == m ? acc[m] : x;
        if (eq.bias) x += (float)eq.bias[n];
        rope_x[wv][lane] = vy_round_bf16(x);   // the projection as the unfused path stores it
      }
      __syncthreads();
      if (lane < M && lane < MR) {
        const int64_t m = lane;
        const int half = eq.dh >> 1;
        const int d = (n % eq.dh) & (half - 1);   // index of the rotary pair (dh is a power of two here)
        float o = rope_x[wv][lane];
        if (n < eq.nq + eq.nkv) {
          const float mine = o, other = rope_x[wv ^ 2][lane];
          const int64_t pp = (eq.pos_dev ? (int64_t)*eq.pos_dev : eq.pos0) * half + d;
          const float c = vy_round_bf16(eq.cos_tab[pp]), sn = vy_round_bf16(eq.sin_tab[pp]);
          // rope2_kernel: low half a*c + (-b*s), high half b*c + a*s, every product rounded to bf16
          o = (wv < 2) ? vy_round_bf16(mine * c) + vy_round_bf16(-other * sn)
                       : vy_round_bf16(mine * c) + vy_round_bf16(other * sn);
        }
        *qkv_dest(eq, m, 0, n) = (bf16)o;
      }
      return;
    }
  }
  if (lane < M && lane < MR) {
    float x = 0.f, x2 = 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) { x = lane == m ? acc[m] : x; x2 = lane == m ? acc2[m] : x2; }
    const int64_t m = lane;
    if constexpr (GATED) {
      const float gte = vy_round_bf16(x), up = vy_round_bf16(x2);
      x = vy_act_fwd<ACT>(gte) * up;
      if (ep.residual) x += (float)ep.residual[m * ep.ldr + n];
      ep.y[m * ep.ldy + n] = (bf16)x;
    } else if constexpr (EPI == 0) {
      if (ep.bias) x += (float)ep.bias[n];
      if (ep.pre) ep.pre[m * ep.ldy + n] = (bf16)(ep.pre_deriv ? vy_act_grad<ACT>(vy_round_bf16(x)) : x);
      x = vy_act_fwd<ACT>(x);
      if (ep.drop.thr) {
        uint32_t lots[4];
        vy_drop_lots(ep.drop, m, n >> 3, lots);
        x = vy_drop_keep(ep.drop, lots, n & 7) ? vy_round_bf16(x) * ep.drop.scale : 0.f;
      }
      if (ep.residual) x += (float)ep.residual[m * ep.ldr + n];
      if (ep.residual2) x += (float)ep.residual2[m * ep.ldr2 + n];
      ep.y[m * ep.ldy + n] = (bf16)x;
    } else {   // packed [q | k | v] projection without fused rotary: bias, head split, scatter
      if (eq.bias) x += (float)eq.bias[n];
      const int64_t b = m / eq.L, l = m - b * eq.L;
      *qkv_dest(eq, b, l, n) = (bf16)x;
    }
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int EPI, int ACT, bool GRAD>
int launch_bf16(const bf16* X, int64_t ldx, const bf16* W, int64_t ldw, int64_t M, int64_t N,
                int64_t K, const EpiPlain<bf16>& ep, const EpiQkv<bf16>& eq, hipStream_t st) {
  static const int rot = [] { const char* e = getenv("VY_GEMM_ROT"); return e ? atoi(e) : 0; }();  // rotated k order: measured 1-5 % slower
  static const int mid_tiles = [] { const char* e = getenv("VY_GEMM_MID"); return e ? atoi(e) : 1; }();
  static const int gemv_on = [] { const char* e = getenv("VY_GEMV"); return e ? atoi(e) : 1; }();
  static const int skinny16_on = [] { const char* e = getenv("VY_SKINNY16"); return e ? atoi(e) : 1; }();
  static const int64_t skinny_max_n = [] { const char* e = getenv("VY_SKINNY_MAXN"); return e ? (int64_t)atoll(e) : (int64_t)8192; }();   // wider (the vocabulary): 32 x 128 tiles with X staged once per workgroup -- 15.5 vs 31.1 us at N = 50265
  if (M <= 4 && (EPI == 0 || !eq.rope) && !GRAD && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && gemv_on &&
      ((uintptr_t)X % 16 == 0) && ((uintptr_t)W % 16 == 0)) {
    if constexpr (!GRAD && (EPI == 1 ? ACT == 0 : true))
      hipLaunchKernelGGL((gemv_bf16_kernel<EPI, ACT, 4>), dim3((unsigned)vy_cdiv(N, 4)), dim3(256), 0, st, X, ldx, W, ldw,
                         (int)M, (int)N, (int)K, ep, eq, (const bf16*)nullptr, 0.f);
  } else if (M <= 32 && K % 32 == 0 && !GRAD && N % 16 == 0 && N / 32 < 200 && skinny16_on &&
             (EPI == 0 || !eq.rope || (N % 64 == 0 && eq.dh == 64))) {
    // few 32-column workgroups: 16-column ones, so that twice as many CUs stream weights
    static const int sk_dbg = [] { const char* e = getenv("VY_SKINNY_DBG"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL((gemm_skinny16_bf16_kernel<EPI, ACT>), dim3((unsigned)(N / 16)), dim3(256), 0, st, X, ldx,
                       W, ldw, (int)M, (int)N, (int)K, ep, eq, sk_dbg);
  } else if (M <= 32 && K % 16 == 0 && !GRAD && (EPI == 0 || !eq.rope || N % 64 == 0) && N <= skinny_max_n) {
    hipLaunchKernelGGL((gemm_skinny_bf16_kernel<EPI, ACT>), dim3((unsigned)vy_cdiv(N, 32)), dim3(256), 0, st, X, ldx,
                       W, ldw, (int)M, (int)N, (int)K, ep, eq);
  } else if (M <= 32) {  // skinny fallback: 32 x 128 tiles
    const int tn = (int)vy_cdiv(N, 128), tm = (int)vy_cdiv(M, 32);
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<32, 128, 1, 4, EPI, ACT, GRAD>), dim3(tm * tn), dim3(256), 0,
                       st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn, ep, eq, rot);
  } else if (M <= 1024 || (mid_tiles && vy_cdiv(M, 256) * vy_cdiv(N, 192) * g_chains < 160)) {
    // mid-size M: also whenever the 256 x 192 grid would leave more than a third of the CUs without a tile
    // (M = 2112 rows of a captioning decoder x N = 768: 36 tiles of 256 x 192, 102 of 128 x 128)
    const int tn = (int)vy_cdiv(N, 128), tm = (int)vy_cdiv(M, 128);
    static const int mid16 = [] { const char* e = getenv("VY_GEMM_MID16"); return e ? atoi(e) : 1; }();
    // 128 < M <= 320 (a 256-patch vision tower, a 264-row PaliGemma prefill): ONE row tile holds every row, so each weight
    // element crosses L2 -> LDS once per launch and the 8 rows past 256 do not cost a third row of 128 x 128 tiles
    // (all-rows tiles of 256 or 320 x 128, 8 waves, one workgroup per CU; VY_GEMM_ROWS=0: the 128 x 128 tiles)
    static const int rows_on = [] { const char* e = getenv("VY_GEMM_ROWS"); return e ? atoi(e) : 1; }();
    if (mid16 && rows_on && M > 128 && M <= 320) {
      if (M <= 256) launch_mid<256, 128, 4, 2, EPI, ACT, GRAD>(X, ldx, W, ldw, M, N, K, ep, eq, st, 256, rot);
      else launch_mid<320, 128, 4, 2, EPI, ACT, GRAD>(X, ldx, W, ldw, M, N, K, ep, eq, st, 256, rot);
    } else if (mid16)
      launch_mid<128, 128, 2, 2, EPI, ACT, GRAD>(X, ldx, W, ldw, M, N, K, ep, eq, st, 448, rot);
    else
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2, EPI, ACT, GRAD>), dim3(tm * tn), dim3(256), 0,
                         st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn, ep, eq, rot);
  } else {
    const int tn = (int)vy_cdiv(N, 192), tm = (int)vy_cdiv(M, 256);
    // Large M.  256 x 192 tiles divide N in {768, 2304} x M = 16384 into whole multiples of the 256 CUs (X operand in a
    // three-deep ring: gemm_nt_bf16_x3m16_kernel); 256 x 256 (fewer L2 bytes per FLOP) where N is wide enough for whole
    // rounds anyway (FFN1, the dgrad of FFN2, the vocabulary projection).  Both on mfma_f32_16x16x32_bf16.
    // VY_GEMM_VARIANT (A/B runs in one process, tools/exp_gemm.py): 8 / 9 force the 256 x 256 / 256 x 192 tiles of the
    // two-stage 32 x 32 x 16 kernel, 40 / 41 force the two default kernels whatever N.  The experimental kernels of rounds
    // 1-2 (persistent pipelined epilogue, two 4-wave workgroups per CU, 4- and 5-slot rings of 32-wide slices, the
    // 32 x 32 x 16 X-ring) measured level or behind these on every box and were removed in round 3.
    const int var = vy_gemm_variant();
    const bool wide = (N >= 3072);
    if (var == 8 || (var != 9 && var != 40 && var != 41 && !vy_m16_on() && wide)) {
      const int tn2 = (int)vy_cdiv(N, 256);
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<256, 256, 4, 2, EPI, ACT, GRAD>), dim3(tm * tn2), dim3(512), 0,
                         st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn2, ep, eq, rot);
    } else if (var == 9 || (var != 40 && var != 41 && !vy_m16_on())) {
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<256, 192, 4, 2, EPI, ACT, GRAD>), dim3(tm * tn), dim3(512), 0,
                         st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn, ep, eq, rot);
    } else if (var == 41 || (var != 40 && wide)) {
      const int tn2 = (int)vy_cdiv(N, 256);
      hipLaunchKernelGGL((gemm_nt_bf16_m16_kernel<256, 256, 4, 2, EPI, ACT, GRAD>), dim3(tm * tn2), dim3(512), 0,
                         st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn2, ep, eq, rot);
    } else {
      hipLaunchKernelGGL((gemm_nt_bf16_x3m16_kernel<192, EPI, ACT, GRAD>), dim3(tm * tn), dim3(512), 0,
                         st, X, ldx, W, ldw, (int)M, (int)N, (int)K, tn, ep, eq, rot);
    }
  }
  return 0;
}

template <int EPI, int ACT, bool GRAD>
int launch_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int64_t N,
               int64_t K, const EpiPlain<float>& ep, const EpiQkv<float>& eq, hipStream_t st) {
  const int tn = (int)vy_cdiv(N, 64), tm = (int)vy_cdiv(M, 64);
  hipLaunchKernelGGL((gemm_nt_f32_kernel<EPI, ACT, GRAD>), dim3(tm * tn), dim3(256), 0, st, X, ldx, W,
                     ldw, (int)M, (int)N, (int)K, tn, ep, eq);
  return 0;
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename T>
int check_operands(const char* who, const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M,
                   int64_t N, int64_t K) {
  const int vec = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  if (!x || !w) VY_FAIL(VY_ERR_ARG, "%s: null operand", who);
  if (M <= 0 || N <= 0 || K <= 0) VY_FAIL(VY_ERR_ARG, "%s: empty problem M=%ld N=%ld K=%ld", who, (long)M, (long)N, (long)K);
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) VY_FAIL(VY_ERR_ARG, "%s: dimension overflow", who);
  if (K % vec) VY_FAIL(VY_ERR_ARG, "%s: K=%ld must be a multiple of %d", who, (long)K, vec);
  if (ldx % vec || ldw % vec || !aligned_to(x, 16) || !aligned_to(w, 16))
    VY_FAIL(VY_ERR_ARG, "%s: operand rows must be 16-byte aligned (ldx=%ld ldw=%ld)", who, (long)ldx, (long)ldw);
  if (ldx < K || ldw < K) VY_FAIL(VY_ERR_ARG, "%s: leading dimension smaller than K", who);
  return 0;
}

template <typename T>
int linear_impl(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                const void* residual, int64_t ldr, const void* gradpre, int64_t ldg, void* y,
                int64_t ldy, void* pre_out, int64_t M, int64_t N, int64_t K, int act, bool grad,
                hipStream_t st, const char* who, const void* residual2 = nullptr, int64_t ldr2 = 0,
                VyDrop drop = VyDrop{}) {
  if (int rc = check_operands<T>(who, x, ldx, w, ldw, M, N, K)) return rc;
  if (!y) VY_FAIL(VY_ERR_ARG, "%s: null output", who);
  if (ldy < N) VY_FAIL(VY_ERR_ARG, "%s: ldy < N", who);
  EpiPlain<T> ep;
  ep.pre_deriv = (act & VY_ACT_SAVE_DERIV) ? 1 : 0;
  act &= ~VY_ACT_SAVE_DERIV;
  ep.bias = (const T*)bias; ep.residual = (const T*)residual; ep.ldr = ldr;
  ep.residual2 = (const T*)residual2; ep.ldr2 = ldr2;
  if (residual2 && !residual) VY_FAIL(VY_ERR_ARG, "%s: add_to2 without add_to", who);
  ep.gradpre = (const T*)gradpre; ep.ldg = ldg; ep.y = (T*)y; ep.ldy = ldy; ep.pre = (T*)pre_out;
  ep.drop = drop;
  static const int wt_env = [] { const char* e = getenv("VY_GEMM_WT_STORE"); return e ? atoi(e) : 1; }();
  ep.wt_store = (wt_env && (int64_t)M * N * (int64_t)sizeof(T) >= (16 << 20)) ? 1 : 0;
  const int ve = 16 / (int)sizeof(T);  // elements per 16-byte access (bf16: 8, f32 quads: 4)
  ep.vec_ok = (ldy % ve == 0) && aligned_to(y, 16) && (!bias || aligned_to(bias, 4 * sizeof(T))) &&
              (!residual || (ldr % ve == 0 && aligned_to(residual, 16))) &&
              (!residual2 || (ldr2 % ve == 0 && aligned_to(residual2, 16))) &&
              (!gradpre || (ldg % ve == 0 && aligned_to(gradpre, 16))) &&
              (!pre_out || aligned_to(pre_out, 16));
  EpiQkv<T> eq{};
#define VY_GO(ACT_, GRAD_)                                                                              \
  do {                                                                                                  \
    if constexpr (sizeof(T) == 2)                                                                       \
      launch_bf16<0, ACT_, GRAD_>((const bf16*)x, ldx, (const bf16*)w, ldw, M, N, K,                    \
                                  *reinterpret_cast<EpiPlain<bf16>*>(&ep), *reinterpret_cast<EpiQkv<bf16>*>(&eq), st); \
    else                                                                                                \
      launch_f32<0, ACT_, GRAD_>((const float*)x, ldx, (const float*)w, ldw, M, N, K,                   \
                                 *reinterpret_cast<EpiPlain<float>*>(&ep), *reinterpret_cast<EpiQkv<float>*>(&eq), st); \
  } while (0)
  if (!grad) {
    if (act == VY_ACT_NONE) VY_GO(VY_ACT_NONE, false);
    else if (act == VY_ACT_GELU_ERF) VY_GO(VY_ACT_GELU_ERF, false);
    else if (act == VY_ACT_GELU_TANH) VY_GO(VY_ACT_GELU_TANH, false);
    else VY_FAIL(VY_ERR_ARG, "%s: unknown activation %d", who, act);
  } else {
    if (act == VY_ACT_NONE || !gradpre) VY_GO(VY_ACT_NONE, true);
    else if (act == VY_ACT_GELU_ERF) VY_GO(VY_ACT_GELU_ERF, true);
    else if (act == VY_ACT_GELU_TANH) VY_GO(VY_ACT_GELU_TANH, true);
    else VY_FAIL(VY_ERR_ARG, "%s: unknown activation %d", who, act);
  }
#undef VY_GO
  VY_CHECK_LAUNCH(who);
  return VY_OK;
}

template <typename T>
int qkv_impl(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
             const float* cos_tab, const float* sin_tab, int64_t pos0, void* q, int64_t q_sb,
             int64_t q_sh, int64_t q_sl, void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl, void* v,
             int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t B, int64_t L, int64_t K, int h, int hk,
             int dh, hipStream_t st, const int* pos_dev = nullptr) {
  const char* who = "vy_qkv_rope_fwd";
  const int64_t M = B * L, N = (int64_t)(h + 2 * hk) * dh;
  if (int rc = check_operands<T>(who, x, ldx, w, ldw, M, N, K)) return rc;
  if (!q || !k || !v) VY_FAIL(VY_ERR_ARG, "%s: null output", who);
  if (h <= 0 || hk <= 0 || h % hk) VY_FAIL(VY_ERR_ARG, "%s: h=%d must be a positive multiple of hk=%d", who, h, hk);
  if (dh % 4) VY_FAIL(VY_ERR_ARG, "%s: head_dim %d must be a multiple of 4", who, dh);
  if ((cos_tab == nullptr) != (sin_tab == nullptr)) VY_FAIL(VY_ERR_ARG, "%s: cos/sin must both be given", who);
  const size_t qa = 4 * sizeof(T);
  const int64_t strides[] = {q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl};
  for (int64_t s : strides)
    if (s % 4) VY_FAIL(VY_ERR_ARG, "%s: output strides must be multiples of 4 elements", who);
  if (!aligned_to(q, qa) || !aligned_to(k, qa) || !aligned_to(v, qa) || (bias && !aligned_to(bias, qa)))
    VY_FAIL(VY_ERR_ARG, "%s: outputs/bias must be 4-element aligned", who);
  EpiQkv<T> eq;
  eq.bias = (const T*)bias; eq.cos_tab = cos_tab; eq.sin_tab = sin_tab; eq.pos0 = pos0;
  eq.q = (T*)q; eq.q_sb = q_sb; eq.q_sh = q_sh; eq.q_sl = q_sl;
  eq.k = (T*)k; eq.k_sb = k_sb; eq.k_sh = k_sh; eq.k_sl = k_sl;
  eq.v = (T*)v; eq.v_sb = v_sb; eq.v_sh = v_sh; eq.v_sl = v_sl;
  eq.L = (int)L; eq.nq = h * dh; eq.nkv = hk * dh; eq.dh = dh; eq.pos_dev = pos_dev;
  // fused rotary: bf16 kernel with 64-wide wave tiles and dh == 64 (pairs are register-local)
  const bool fuse = cos_tab && sizeof(T) == 2 && dh == 64;
  eq.rope = fuse ? 1 : 0;
  {
    bool v8 = dh % 8 == 0 && aligned_to(q, 16) && aligned_to(k, 16) && aligned_to(v, 16);
    for (int64_t st_ : strides) v8 = v8 && (st_ % 8 == 0);
    eq.vec8 = v8 ? 1 : 0;
  }
  EpiPlain<T> ep{};
  if constexpr (sizeof(T) == 2)
    launch_bf16<1, VY_ACT_NONE, false>((const bf16*)x, ldx, (const bf16*)w, ldw, M, N, K,
                                       *reinterpret_cast<EpiPlain<bf16>*>(&ep), *reinterpret_cast<EpiQkv<bf16>*>(&eq), st);
  else
    launch_f32<1, VY_ACT_NONE, false>((const float*)x, ldx, (const float*)w, ldw, M, N, K,
                                      *reinterpret_cast<EpiPlain<float>*>(&ep), *reinterpret_cast<EpiQkv<float>*>(&eq), st);
  VY_CHECK_LAUNCH(who);
  if (cos_tab && !fuse) {
    if (pos_dev) VY_FAIL(VY_ERR_UNSUPPORTED, "%s: device-side position needs the fused RoPE path (bf16, dh == 64)", who);
    const int vdt = sizeof(T) == 2 ? VY_BF16 : VY_F32;
    if (int rc = vy_rope_qk(q, q_sb, q_sh, q_sl, h, k, k_sb, k_sh, k_sl, hk, cos_tab, sin_tab, pos0, B, L, dh, vdt,
                            (hipStream_t)st)) return rc;
  }
  return VY_OK;
}

}  // namespace

extern "C" int vy_linear_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                             const void* residual, int64_t ldr, void* y, int64_t ldy, void* pre_out,
                             int64_t M, int64_t N, int64_t K, int act, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16)
    return linear_impl<bf16>(x, ldx, w, ldw, bias, residual, ldr, nullptr, 0, y, ldy, pre_out, M, N, K, act, false, st, "vy_linear_fwd");
  if (dtype == VY_F32)
    return linear_impl<float>(x, ldx, w, ldw, bias, residual, ldr, nullptr, 0, y, ldy, pre_out, M, N, K, act, false, st, "vy_linear_fwd");
  VY_FAIL(VY_ERR_ARG, "vy_linear_fwd: bad dtype %d", dtype);
}

extern "C" int vy_linear_dropout_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                                     const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int64_t N,
                                     int64_t K, float p_drop, uint64_t seed, uint64_t offset, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!(p_drop >= 0.f && p_drop <= 1.f)) VY_FAIL(VY_ERR_ARG, "vy_linear_dropout_fwd: p=%g outside [0, 1]", (double)p_drop);
  const VyDrop d = vy_make_drop(p_drop, seed, offset);
  if (dtype == VY_BF16)
    return linear_impl<bf16>(x, ldx, w, ldw, bias, residual, ldr, nullptr, 0, y, ldy, nullptr, M, N, K, VY_ACT_NONE, false, st, "vy_linear_dropout_fwd", nullptr, 0, d);
  if (dtype == VY_F32)
    return linear_impl<float>(x, ldx, w, ldw, bias, residual, ldr, nullptr, 0, y, ldy, nullptr, M, N, K, VY_ACT_NONE, false, st, "vy_linear_dropout_fwd", nullptr, 0, d);
  VY_FAIL(VY_ERR_ARG, "vy_linear_dropout_fwd: bad dtype %d", dtype);
}

extern "C" int vy_linear_dgrad(const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const void* pre,
                               int64_t ldpre, int act, const void* add_to, int64_t ldadd, const void* add_to2,
                               int64_t ldadd2, void* dx, int64_t lddx, int64_t M, int64_t N, int64_t K, int dtype,
                               void* stream) {
  // dX[M,K] = dY[M,N] . W[N,K] = dY . (W^T)^T with W^T stored [K,N]: an NT GEMM whose
  // "N" is K and whose contraction runs over N.
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16)
    return linear_impl<bf16>(dy, lddy, wt, ldwt, nullptr, add_to, ldadd, pre, ldpre, dx, lddx, nullptr, M, K, N, act, true, st, "vy_linear_dgrad", add_to2, ldadd2);
  if (dtype == VY_F32)
    return linear_impl<float>(dy, lddy, wt, ldwt, nullptr, add_to, ldadd, pre, ldpre, dx, lddx, nullptr, M, K, N, act, true, st, "vy_linear_dgrad", add_to2, ldadd2);
  VY_FAIL(VY_ERR_ARG, "vy_linear_dgrad: bad dtype %d", dtype);
}

extern "C" int vy_qkv_rope_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                               const float* cos_tab, const float* sin_tab, int64_t pos0, void* q,
                               int64_t q_sb, int64_t q_sh, int64_t q_sl, void* k, int64_t k_sb,
                               int64_t k_sh, int64_t k_sl, void* v, int64_t v_sb, int64_t v_sh,
                               int64_t v_sl, int64_t B, int64_t L, int64_t K, int h, int hk, int dh,
                               int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16)
    return qkv_impl<bf16>(x, ldx, w, ldw, bias, cos_tab, sin_tab, pos0, q, q_sb, q_sh, q_sl, k, k_sb, k_sh, k_sl, v, v_sb, v_sh, v_sl, B, L, K, h, hk, dh, st);
  if (dtype == VY_F32)
    return qkv_impl<float>(x, ldx, w, ldw, bias, cos_tab, sin_tab, pos0, q, q_sb, q_sh, q_sl, k, k_sb, k_sh, k_sl, v, v_sb, v_sh, v_sl, B, L, K, h, hk, dh, st);
  VY_FAIL(VY_ERR_ARG, "vy_qkv_rope_fwd: bad dtype %d", dtype);
}

// internal: vy_qkv_rope_fwd with the token position read from device memory (graph replay)
// y = LN(x W^T + bias + residual) for M <= 32 rows (decode), bf16: split-K GEMM + fused finish.
// `part` is an fp32 scratch of vy_splitk_ws_floats(N) elements.  Internal to the decode driver.
int64_t vy_splitk_ws_floats(int64_t N) { return 8 * 32 * N; }

extern "C" int vy_debug_set_gemm_variant(int v) { g_gemm_variant = v; return 0; }
extern "C" int vy_workspace_set(void* stream, void* ws, int64_t bytes) {
  if (bytes < 0 || (ws && ((uintptr_t)ws & 255))) VY_FAIL(VY_ERR_ARG, "vy_workspace_set: bad workspace (256-byte aligned, bytes >= 0)");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < g_ws_n; ++i)
    if (g_ws[i].st == st) {
      if (ws && bytes) { g_ws[i].p = ws; g_ws[i].bytes = bytes; }
      else { g_ws[i] = g_ws[g_ws_n - 1]; --g_ws_n; }
      return VY_OK;
    }
  if (!ws || !bytes) return VY_OK;
  if (g_ws_n == 16) {   // full: the oldest entry goes (its stream's later launches run unsplit -- speed only)
    for (int i = 1; i < 16; ++i) g_ws[i - 1] = g_ws[i];
    --g_ws_n;
  }
  g_ws[g_ws_n++] = WsEntry{st, ws, bytes};
  return VY_OK;
}
extern "C" int vy_set_concurrent_chains(int n) {
  if (n < 1 || n > 8) VY_FAIL(VY_ERR_ARG, "vy_set_concurrent_chains: n = %d (1..8)", n);
  g_chains = n;
  return VY_OK;
}

// measurement aid, not part of include/vyom_hip.h: {shader cycles, 10 ns ticks, launches} since the last call
extern "C" int vy_debug_gemm_clock(unsigned long long* out3) {   // out3: 6 values
  unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(vy_gemm_clk), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(vy_gemm_clk), z, sizeof(z)) != hipSuccess;
}
int vy_linear_res_ln_skinny(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                            const void* residual, int64_t ldr, const void* gamma, const void* beta, float eps,
                            void* y, int64_t ldy, float* part, int64_t M, int64_t N, int64_t K, void* stream) {
  const char* who = "vy_linear_res_ln_skinny";
  if (!x || !w || !gamma || !beta || !y || !part) VY_FAIL(VY_ERR_ARG, "%s: null operand", who);
  if (M < 1 || M > 32 || N % 32 || K % 16 || N > 8192) VY_FAIL(VY_ERR_UNSUPPORTED, "%s: needs M <= 32, N %% 32 == 0 (<= 8192), K %% 16 == 0", who);
  if (ldx % 8 || ldw % 8 || ldy % 8 || (residual && ldr % 8)) VY_FAIL(VY_ERR_ARG, "%s: strides must be multiples of 8", who);
  hipStream_t st = (hipStream_t)stream;
  const int nsteps = (int)(K / 16), nwg = (int)(N / 32);
  int ksplit = (256 + nwg - 1) / nwg;            // ~one workgroup per CU
  if (ksplit > 8) ksplit = 8;
  if (ksplit > nsteps / 8) ksplit = nsteps / 8;  // >= 8 k-steps (two per wave) per workgroup
  if (ksplit < 1) ksplit = 1;
  int spw = (nsteps + ksplit - 1) / ksplit;
  spw = (spw + 3) / 4 * 4;
  ksplit = (nsteps + spw - 1) / spw;
  hipLaunchKernelGGL(gemm_skinny_splitk_kernel, dim3((unsigned)nwg, (unsigned)ksplit), dim3(256), 0, st, (const bf16*)x,
                     ldx, (const bf16*)w, ldw, (int)M, (int)N, (int)K, spw, part);
  VY_CHECK_LAUNCH(who);
  static const int row_fin = [] { const char* e = getenv("VY_DECODE_ROW_FINISH"); return e ? atoi(e) : 1; }();
  if (row_fin && N % 4 == 0) {   // one workgroup per row
    const int qpt = (int)vy_cdiv(N / 4, 256);
#define ROW_GO(Q)                                                                                               \
    hipLaunchKernelGGL((splitk_finish_ln_row_kernel<Q>), dim3((unsigned)M), dim3(256), 0, st, part, ksplit, (int)M, (int)N, \
                       (const bf16*)bias, (const bf16*)residual, ldr, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, ldy, eps)
    if (qpt <= 1) ROW_GO(1);
    else if (qpt <= 2) ROW_GO(2);
    else if (qpt <= 4) ROW_GO(4);
    else ROW_GO(8);
#undef ROW_GO
    VY_CHECK_LAUNCH(who);
    return VY_OK;
  }
  const int nch = (int)(N / 8);
  const dim3 grid((unsigned)((M + 3) / 4)), block(256);
#define FIN_GO(CH)                                                                                               \
  hipLaunchKernelGGL((splitk_finish_ln_kernel<CH>), grid, block, 0, st, part, ksplit, (int)M, (int)N, (const bf16*)bias, \
                     (const bf16*)residual, ldr, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, ldy, eps)
  if (nch <= 64) FIN_GO(1);
  else if (nch <= 128) FIN_GO(2);
  else if (nch <= 256) FIN_GO(4);
  else if (nch <= 512) FIN_GO(8);
  else FIN_GO(16);
#undef FIN_GO
  VY_CHECK_LAUNCH(who);
  return VY_OK;
}

// Single-sequence (M <= 4) bf16 projections with the RMSNorm of their input fused in (internal: the Gemma decode
// driver).  norm_w NULL = no norm.  vy_gemv_gated: w = packed [gate; up] ([2 I, K]) -> y[M, I] = act(gate) * up.
// norm_w of the launchers below: NULL = no norm, VY_NORM_PRESCALED = the weights carry the norm's (1 + w) (mode 2),
// anything else = the RMSNorm weight vector (mode 1)
#define VY_NORM_PRESCALED ((const void*)(intptr_t)-1)
static bool gemv_ok(const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int64_t K) {
  return M >= 1 && M <= 4 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0);
}
int vy_gemv_norm(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, const void* norm_w, float eps,
                 const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int64_t N, int64_t K, void* stream) {
  if (!gemv_ok(x, ldx, w, ldw, M, K) || !x || !w || !y) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_gemv_norm: needs M <= 4 and 16-byte aligned rows");
  EpiPlain<bf16> ep{};
  ep.bias = (const bf16*)bias; ep.residual = (const bf16*)residual; ep.ldr = ldr; ep.y = (bf16*)y; ep.ldy = ldy;
  EpiQkv<bf16> eq{};
  const dim3 grid((unsigned)vy_cdiv(N, 4)), block(256);
  if (norm_w == VY_NORM_PRESCALED)
    hipLaunchKernelGGL((gemv_bf16_kernel<0, VY_ACT_NONE, 4, 2, false>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)M, (int)N, (int)K, ep, eq, (const bf16*)nullptr, eps);
  else if (norm_w)
    hipLaunchKernelGGL((gemv_bf16_kernel<0, VY_ACT_NONE, 4, 1, false>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)M, (int)N, (int)K, ep, eq, (const bf16*)norm_w, eps);
  else
    hipLaunchKernelGGL((gemv_bf16_kernel<0, VY_ACT_NONE, 4, 0, false>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)M, (int)N, (int)K, ep, eq, (const bf16*)nullptr, 0.f);
  VY_CHECK_LAUNCH("vy_gemv_norm");
  return VY_OK;
}
int vy_gemv_qkv_norm(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, const void* norm_w, float eps,
                     void* q, int64_t q_sb, int64_t q_sh, void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl, void* v,
                     int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t B, int64_t K, int h, int hk, int dh,
                     const float* cos_tab, const float* sin_tab, int64_t pos, void* stream) {
  // packed [q | k | v] projection of single-token rows (L = 1); cos_tab != NULL: rotary embedding at position `pos`
  // fused (head widths that are a power of two and a multiple of 4), else applied by vy_rope_qk afterwards
  if (!gemv_ok(x, ldx, w, ldw, B, K)) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_gemv_qkv_norm: needs B <= 4 and 16-byte aligned rows");
  EpiPlain<bf16> ep{};
  EpiQkv<bf16> eq{};
  eq.bias = (const bf16*)bias; eq.q = (bf16*)q; eq.q_sb = q_sb; eq.q_sh = q_sh; eq.q_sl = (int64_t)h * dh;
  eq.k = (bf16*)k; eq.k_sb = k_sb; eq.k_sh = k_sh; eq.k_sl = k_sl;
  eq.v = (bf16*)v; eq.v_sb = v_sb; eq.v_sh = v_sh; eq.v_sl = v_sl;
  eq.L = 1; eq.nq = h * dh; eq.nkv = hk * dh; eq.dh = dh; eq.rope = 0; eq.vec8 = 0; eq.pos_dev = nullptr;
  if (cos_tab) {
    if (dh % 4 || (dh & (dh - 1)) || !sin_tab) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_gemv_qkv_norm: fused rotary needs a power-of-two head width");
    eq.rope = 1; eq.cos_tab = cos_tab; eq.sin_tab = sin_tab; eq.pos0 = pos;
  }
  const int64_t N = (int64_t)(h + 2 * hk) * dh;
  const dim3 grid((unsigned)vy_cdiv(N, 4)), block(256);
  if (norm_w == VY_NORM_PRESCALED)
    hipLaunchKernelGGL((gemv_bf16_kernel<1, VY_ACT_NONE, 4, 2, false>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)B, (int)N, (int)K, ep, eq, (const bf16*)nullptr, eps);
  else if (norm_w)
    hipLaunchKernelGGL((gemv_bf16_kernel<1, VY_ACT_NONE, 4, 1, false>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)B, (int)N, (int)K, ep, eq, (const bf16*)norm_w, eps);
  else
    hipLaunchKernelGGL((gemv_bf16_kernel<1, VY_ACT_NONE, 4, 0, false>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)B, (int)N, (int)K, ep, eq, (const bf16*)nullptr, 0.f);
  VY_CHECK_LAUNCH("vy_gemv_qkv_norm");
  return VY_OK;
}
int vy_gemv_gated(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* norm_w, float eps, void* y, int64_t ldy,
                  int64_t M, int64_t I, int64_t K, int act, void* stream) {
  if (!gemv_ok(x, ldx, w, ldw, M, K)) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_gemv_gated: needs M <= 4 and 16-byte aligned rows");
  if (act != VY_ACT_GELU_TANH) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_gemv_gated: gelu_tanh only");
  EpiPlain<bf16> ep{};
  ep.y = (bf16*)y; ep.ldy = ldy;
  EpiQkv<bf16> eq{};
  const dim3 grid((unsigned)vy_cdiv(I, 4)), block(256);
  if (norm_w == VY_NORM_PRESCALED)
    hipLaunchKernelGGL((gemv_bf16_kernel<0, VY_ACT_GELU_TANH, 4, 2, true>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)M, (int)I, (int)K, ep, eq, (const bf16*)nullptr, eps);
  else if (norm_w)
    hipLaunchKernelGGL((gemv_bf16_kernel<0, VY_ACT_GELU_TANH, 4, 1, true>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)M, (int)I, (int)K, ep, eq, (const bf16*)norm_w, eps);
  else
    hipLaunchKernelGGL((gemv_bf16_kernel<0, VY_ACT_GELU_TANH, 4, 0, true>), grid, block, 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       (const bf16*)w, ldw, (int)M, (int)I, (int)K, ep, eq, (const bf16*)nullptr, 0.f);
  VY_CHECK_LAUNCH("vy_gemv_gated");
  return VY_OK;
}

int vy_qkv_rope_fwd_ex(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                       const float* cos_tab, const float* sin_tab, int64_t pos0, const int* pos_dev, void* q,
                       int64_t q_sb, int64_t q_sh, int64_t q_sl, void* k, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                       void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t B, int64_t L, int64_t K, int h,
                       int hk, int dh, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16)
    return qkv_impl<bf16>(x, ldx, w, ldw, bias, cos_tab, sin_tab, pos0, q, q_sb, q_sh, q_sl, k, k_sb, k_sh, k_sl, v, v_sb, v_sh, v_sl, B, L, K, h, hk, dh, st, pos_dev);
  if (dtype == VY_F32)
    return qkv_impl<float>(x, ldx, w, ldw, bias, cos_tab, sin_tab, pos0, q, q_sb, q_sh, q_sl, k, k_sb, k_sh, k_sl, v, v_sb, v_sh, v_sl, B, L, K, h, hk, dh, st, pos_dev);
  VY_FAIL(VY_ERR_ARG, "vy_qkv_rope_fwd: bad dtype %d", dtype);
}
