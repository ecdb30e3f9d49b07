// Scaled-dot-product attention forward for the VyomAI hot path.
//
// attn_fwd_mfma_kernel (bf16, dh in {64,128}): flash-style, one workgroup = 4 waves = 128 query
// rows of one (batch, head); K/V tiles of 64 keys are staged by LDS-DMA through a 3-deep ring.
//   * "swapped" QK^T: S^T = K.Q^T (A operand = K rows read with ds_read_b128 from an XOR-swizzled
//     [key][d] image, B operand = Q kept in registers), so each lane owns ONE query row and its
//     softmax statistics; the row max/sum need one cross-half exchange (lane ^ 32) only;
//   * P never leaves registers: the S^T accumulator is, after exp2 and a bf16 pack, exactly the
//     B operand of O^T = V^T.P^T (k order permuted identically on both operands);
//   * V^T fragments come from the row-major [key][d] V image through ds_read_b64_tr_b16 (the
//     gfx950 transposing LDS read); the image is swizzled in 64-byte units so the four rows of a
//     transposed read hit distinct banks;
//   * GQA: kv head = q head / (h/hk) -- repeat_kv is never materialised;
//   * masks are descriptors, not tensors: causal offset, key-padding bytes, or a generic additive
//     fp32 mask.  The reference's masked score is finfo.min (the score is absorbed): it weighs
//     exactly 0 next to any visible key, and a row with no visible key averages V over all keys.
//
// attn_rowwise_kernel (f32 / any dh, and the L==1 decode path): one workgroup per query row, a
// single online-softmax pass over K and V with 16-byte coalesced loads, no LDS score buffer.
//
// Reference: F.scaled_dot_product_attention call sites, VyomAI/layers/attention.py:128,209,283,
// 373,619 and VyomAI/models/decoder.py:107,195.
#include "vy_common.h"
#include <float.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnParams {
  const void* q; int64_t q_sb, q_sh, q_sl;
  const void* k; int64_t k_sb, k_sh, k_sl;
  const void* v; int64_t v_sb, v_sh, v_sl;
  void* out; int64_t o_sb, o_sl;
  float* lse;
  int mask_kind; int start_pos;
  const uint8_t* keypad; int64_t kp_sb;
  const float* addmask; int64_t am_sb, am_sl;
  int B, h, hk, L, S;
  float scale;
  const int* pos_dev;  // decode under a hipGraph: S = *pos_dev + 1 (else NULL)
  int diag;            // timing experiments only (VY_ATTN_DIAG): 1 = no output store, 2 = no tile loop
};

// ------------------------------------------------------------------------------------------
// MFMA flash forward (bf16)
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256, DH == 64 ? 3 : 1) void attn_fwd_mfma_kernel(AttnParams p) {
  constexpr int RB = DH * 2;            // bytes per K/V row
  constexpr int TILE = 64 * RB;         // bytes per 64-key tile
  constexpr int NP = TILE / 1024 / 4;   // 1-KiB LDS-DMA pieces per wave per tile
  constexpr int KS = DH / 16;           // k-steps of the QK^T contraction
  constexpr int ND = DH / 32;           // 32-wide d blocks of O^T
  constexpr int NS = 3;                 // K/V tiles in the LDS ring: NS - 1 tiles of loads in flight (4: no gain)
  constexpr int KPW = 256;              // key-padding visibility words (64 keys each): S <= 16384
  // ONE shared object (a second one makes hipcc wait vmcnt(0) before every ds_read while LDS-DMA
  // is in flight): K ring, V ring, then 4 flag words for the block-wide OR below
  __shared__ __attribute__((aligned(16))) char smem[2 * NS * TILE + 64 + KPW * 8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.L + 127) / 128;
  // grid = (h*B, query blocks): the dispatcher walks x first, so ALL the heaviest query blocks
  // (most keys under a causal mask) of every (batch, head) start before any lighter one --
  // longest-processing-time order; with (query block, head, batch) order the last (batch, head)
  // groups still start full-length workgroups at the very end and causal ran as long as full
  const int qb = nqb - 1 - (int)blockIdx.y;
  const int head = (int)blockIdx.x % p.h, b = (int)blockIdx.x / p.h;
  const int kvh = head / (p.h / p.hk);
  const int q0 = qb * 128;
  const int fr = lane & 31, fh = lane >> 5;
  const int qi = q0 + wave * 32 + fr;  // this lane's query row
  const int qrow = qi < p.L ? qi : p.L - 1;

  const bf16* Q = (const bf16*)p.q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh + (int64_t)qrow * p.q_sl;
  const bf16* Kb = (const bf16*)p.k + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
  const bf16* Vb = (const bf16*)p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;

  bf16x8 qf[KS];  // loaded after the first K/V tiles are in flight (below)
  const bool causal = p.mask_kind & VY_MASK_CAUSAL;
  const bool haskp = p.mask_kind & VY_MASK_KEYPAD;
  const bool hasadd = p.mask_kind & VY_MASK_ADDITIVE;
  const uint8_t* kp = haskp ? p.keypad + (int64_t)b * p.kp_sb : nullptr;
  const float* am = hasadd ? p.addmask + (int64_t)b * p.am_sb + (int64_t)qrow * p.am_sl : nullptr;

  // LDS-DMA source geometry: piece pc covers LDS bytes [pc*1024, +1024); lane byte P
  int ld_row[NP], ld_koff[NP], ld_voff[NP];
  // per-lane source pointers of tile 0 (rows clamped to the last key): a full tile adds a wave-uniform offset to them --
  // two 64-bit adds per piece instead of the clamp + 64-bit multiplies by the row strides (12 quarter-rate multiplies per
  // tile and wave in a loop that is bound by vector issue)
  const bf16* ksrc0[NP];
  const bf16* vsrc0[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    const int P = (wave * NP + t) * 1024 + lane * 16;
    const int row = P / RB, off = P % RB;
    const int ksw = (RB == 128) ? ((row >> 1) & 7) : (row & 15);
    const int vsw = (RB == 128) ? (((row >> 1) & 1) << 6) : ((row & 3) << 6);
    ld_row[t] = row;
    ld_koff[t] = ((((off >> 4) ^ ksw) << 4)) >> 1;  // element offset inside the row
    ld_voff[t] = (off ^ vsw) >> 1;
    const int r0 = row < p.S ? row : p.S - 1;
    ksrc0[t] = Kb + (int64_t)r0 * p.k_sl + ld_koff[t];
    vsrc0[t] = Vb + (int64_t)r0 * p.v_sl + ld_voff[t];
  }
  const int64_t k_tile = 64 * p.k_sl, v_tile = 64 * p.v_sl;   // elements per 64-key tile
  auto stage = [&](int tile, int buf) {
    const int k0 = tile * 64;
    if (k0 + 64 <= p.S) {   // wave-uniform: every row of the tile exists
      const int64_t ko = (int64_t)tile * k_tile, vo = (int64_t)tile * v_tile;
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)(ksrc0[t] + ko),
                                         (VY_LDS void*)(smem + buf * TILE + (wave * NP + t) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)(vsrc0[t] + vo),
                                         (VY_LDS void*)(smem + (NS + buf) * TILE + (wave * NP + t) * 1024), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      int kr = k0 + ld_row[t];
      kr = kr < p.S ? kr : p.S - 1;
      const bf16* ks = Kb + (int64_t)kr * p.k_sl + ld_koff[t];
      const bf16* vs = Vb + (int64_t)kr * p.v_sl + ld_voff[t];
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)ks,
                                       (VY_LDS void*)(smem + buf * TILE + (wave * NP + t) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const VY_GLOBAL void*)vs,
                                       (VY_LDS void*)(smem + (NS + buf) * TILE + (wave * NP + t) * 1024), 16, 0, 0);
    }
  };

  // read-side lane constants
  const int k_sw = (RB == 128) ? ((fr >> 1) & 7) : (fr & 15);
  const int li = lane & 15, g16 = (lane >> 4) & 1;
  const int v_sw = (RB == 128) ? (((li >> 3) & 1) << 6) : (((li >> 2) & 3) << 6);
  const int v_lane_row = 4 * fh + (li >> 2);            // + 32kb + 16s + 8u
  const int v_lane_off = 32 * g16 + 8 * (li & 3);       // + 64n, then ^ v_sw
  unsigned k_lds[KS];  // LDS byte address of this lane's K fragment of k-step ks (key row fr, buffer 0)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_lds[ks] = vy_lds_addr(smem) + fr * RB + (((2 * ks + fh) ^ k_sw) << 4);
  unsigned v_lds[ND];  // LDS byte address of V^T fragment column n, row v_lane_row, buffer 0 of the K ring
#pragma unroll
  for (int n = 0; n < ND; ++n) v_lds[n] = vy_lds_addr(smem) + v_lane_row * RB + ((64 * n + v_lane_off) ^ v_sw);

  f32x16 o[ND];
#pragma unroll
  for (int n = 0; n < ND; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
  // running max in scaled log2 units.  Always finite: a masked score is -inf here (it contributes
  // exactly 0, as exp(finfo.min - m) does in the reference whenever the row has a visible key);
  // rows with NO visible key are recognised by l == 0 after the loop and get the reference's
  // uniform average over all keys from the column-sum pass below.
  float m_run = -FLT_MAX, l_run = 0.f;
  const float c = p.scale * LOG2E;

  // pure causal: keys beyond the diagonal of the block's last row are never visible -> skipped
  const int nt_all = (p.S + 63) / 64;
  int nt = nt_all;
  if (causal) {
    const int kv_end = min(p.S, p.start_pos + q0 + 128);
    nt = (kv_end + 63) / 64;
    if (nt < 1) nt = 1;
  }
  const int wave_first = q0 + wave * 32, wave_last = wave_first + 31;

  // key-padding mask -> one 64-bit visibility word per key tile, built once (an ordinary load inside
  // the tile loop would make the compiler drain the LDS-DMA ring with vmcnt(0) every tile)
  unsigned long long* kpbits = reinterpret_cast<unsigned long long*>(smem + 2 * NS * TILE + 64);
  if (haskp) {
    for (int t = wave; t < nt; t += 4) {
      const int kj = t * 64 + lane;
      const bool vis = kj < p.S && kp[kj < p.S ? kj : 0] != 0;
      const unsigned long long bits = __ballot(vis);
      if (lane == 0) kpbits[t] = bits;
    }
    __syncthreads();
  }

  auto compute = [&](int tile, int buf) {
    const int k0 = tile * 64;
    // wave-uniform skip of tiles wholly above this wave's diagonal
    if (causal && k0 > p.start_pos + wave_last) return;
    const char* kb_ = smem + buf * TILE;
    const char* vb_ = smem + (NS + buf) * TILE;
    f32x16 st[2];
    // Hidden (asm) LDS reads retired by COUNTED waits: the 2 KS K fragments are requested first; MFMA i waits for
    // fragment i only (lgkmcnt counts in order) and the V^T fragments are requested two at a time behind each QK^T
    // MFMA, so they land under the remaining MFMAs and the softmax.  (With compiler-visible K reads every QK^T MFMA
    // sat behind an lgkmcnt(0) that also drained the V^T reads just issued.)
    bf16x8 kfr[2 * KS];
    unsigned ka[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = k_lds[ks] + buf * TILE;
    vy_static_for<2 * KS>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      kfr[i] = vy_lds_read128_off<(i / KS) * 32 * RB>(ka[i % KS]);
    });
    unsigned vbase[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) vbase[n] = v_lds[n] + buf * TILE;
    bf16x8 vfr[4 * ND];
    auto vread = [&](auto f_c) {
      constexpr int f = decltype(f_c)::value;
      if constexpr (f < 4 * ND) {
        constexpr int n = f >> 2, kb = (f >> 1) & 1, s_ = f & 1;
        vfr[f] = vy_lds_tr16_pair_off<NS * TILE + (32 * kb + 16 * s_) * RB, NS * TILE + (32 * kb + 16 * s_ + 8) * RB>(vbase[n]);
      }
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[0][r] = 0.f; st[1][r] = 0.f; }
    // V^T requests per QK^T MFMA: 4 ND fragments (two ds_read_b64_tr each) spread over the 2 KS MFMAs
    constexpr int VPM = (4 * ND + 2 * KS - 1) / (2 * KS);
    // (lgkmcnt is a 4-bit counter: where the exact count exceeds 15 the wait is for 15, i.e. stricter -- dh = 128)
    vy_static_for<2 * KS>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      // outstanding before this wait: K fragments i .. 2KS-1 and the 2 VPM i tr reads issued so far
      constexpr int allow = (2 * KS - 1 - i) + 2 * VPM * i;
      if constexpr (allow <= 15) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(allow) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
      vy_tie(kfr[i]);
      st[i / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[i], qf[i % KS], st[i / KS], 0, 0, 0);
      vy_static_for<VPM>([&](auto j_c) { vread(std::integral_constant<int, i * VPM + decltype(j_c)::value>{}); });
    });
    unsigned long long vis = ~0ull;
    if (haskp) vis = kpbits[tile];
    const bool need_mask = hasadd || (k0 + 64 > p.S) || (causal && k0 + 63 > p.start_pos + wave_first) || vis != ~0ull;
    // softmax runs on the RAW scores: max first, then p = exp2(fma(s, c, -m)) -- one FMA and one
    // v_exp_f32 per element (the VALU, not the MFMA pipe, bounds dh=64)
    float tmax = -INFINITY;
    if (need_mask) {
      // key index of register r is k0 + 4fh + kofs, kofs = 32kb + (r&3) + 8(r>>2).  Visibility of
      // the lane's 32 keys as one bit word: padding bits shifted by 4fh, AND kofs <= klim (causal
      // diagonal and end of sequence)
      int klim = p.S - 1 - k0 - 4 * fh;
      if (causal) klim = min(klim, qi + p.start_pos - k0 - 4 * fh);
      unsigned long long lm = klim >= 63 ? ~0ull : (klim < 0 ? 0ull : ((2ull << klim) - 1ull));
      lm &= vis >> (4 * fh);
      const unsigned lmw[2] = {(unsigned)lm, (unsigned)(lm >> 32)};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int bit = (r & 3) + 8 * (r >> 2);
          float t = st[kb][r];
          if (hasadd) {
            // in-range guard only: invisible keys are overwritten below
            const int kj = min(k0 + 4 * fh + 32 * kb + bit, p.S - 1);
            t += am[kj] * (LOG2E / c);
            if (!(t > -FLT_MAX)) t = -INFINITY;  // finfo.min-style additive masks absorb the score
          }
          t = ((lmw[kb] >> bit) & 1u) ? t : -INFINITY;
          st[kb][r] = t;
          tmax = fmaxf(tmax, t);
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[kb][r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float tmax_s = tmax * c;  // -inf stays -inf
    // lazy rescale: the running max is only raised (and O, l rescaled) when some row of the wave
    // would otherwise see p > 2^8; exp2 arguments stay <= 8, harmless for fp32 sums and bf16 P
    if (__any(tmax_s > m_run + 8.f)) {
      const float m_new = fmaxf(m_run, tmax_s);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
    }
    const float neg_m = -m_run;
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(st[kb][r], c, neg_m));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(st[kb][r + 1], c, neg_m));
        st[kb][r] = e0; st[kb][r + 1] = e1;
        rs0 += e0; rs1 += e1;
      }
    l_run += rs0 + rs1;  // per-half partial; halves are summed at the end
    // P fragments: registers 8s..8s+7 of block kb <-> keys 32kb+16s+8(j>>2)+4h+(j&3)
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kb][s][j] = (bf16)st[kb][8 * s + j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int f = 0; f < 4 * ND; ++f) vy_tie(vfr[f]);
#pragma unroll
    for (int f = 0; f < 4 * ND; ++f)
      o[f >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[f], pf[(f >> 1) & 1][f & 1], o[f >> 2], 0, 0, 0);
  };

  // Loads run NS-1 tiles ahead of the MFMAs: a tile is waited for with a COUNTED vmcnt (its own
  // 2*NP LDS-DMA instructions are the oldest outstanding ones of the wave) and one raw s_barrier per
  // tile; nothing in the loop drains the memory pipe, so the L2/HBM latency of a tile is covered
  // by the softmax + MFMA work of the previous tiles instead of being paid once per tile.
  auto wait_tile = [&](int younger) {   // `younger` tiles (2*NP LDS-DMA instructions each) may stay in flight
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // The Q fragments came through ordinary loads: make the compiler retire them HERE (an asm that
  // reads them), before any LDS-DMA is in flight -- otherwise its wait for them sits at the first
  // MFMA inside the loop as vmcnt(0) and drains the ring every tile.
#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_)
    if (s_ < nt) stage(s_, s_);
  // Q after the first K/V tiles are requested: one memory latency for both, not two in a row
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + ks * 16 + fh * 8);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
  for (int t = 0; t < (p.diag == 2 ? 0 : nt); ++t) {
    wait_tile(min(NS - 2, nt - 1 - t));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + NS - 1 < nt) stage(t + NS - 1, (t + NS - 1) % NS);  // the buffer tile t-1 was read from
    compute(t, t % NS);
  }

  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (haskp || hasadd) {
    // Rows without a single visible key: the reference's softmax over [finfo.min, ...] is uniform
    // over EVERY key (causally hidden ones too), i.e. the output is the column mean of V.  Block-wide
    // OR of "one of my rows is such a row" through the flag words, then a cooperative column sum.
    int* flags = reinterpret_cast<int*>(smem + 2 * NS * TILE);
    const bool dead = l_tot == 0.f;
    const bool mine = __any(dead && qi < p.L);
    __builtin_amdgcn_s_barrier();  // every wave is done with the ring
    if (lane == 0) flags[wave] = mine ? 1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if ((flags[0] | flags[1] | flags[2] | flags[3]) != 0) {
      constexpr int NCH = DH / 8, NSL = 256 / NCH;
      float* part = reinterpret_cast<float*>(smem);              // [NSL][DH]
      float* colsum = part + NSL * DH;                           // [DH]
      const int ch = tid % NCH, sl = tid / NCH;
      float a8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[e] = 0.f;
      for (int kj = sl; kj < p.S; kj += NSL) {
        const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)kj * p.v_sl + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] += (float)v8[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) part[sl * DH + ch * 8 + e] = a8[e];
      __syncthreads();
      if (tid < DH) {
        float sacc = 0.f;
        for (int s_ = 0; s_ < NSL; ++s_) sacc += part[s_ * DH + tid];
        colsum[tid] = sacc;
      }
      __syncthreads();
      if (dead) {
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[n][r] = colsum[32 * n + 8 * (r >> 2) + (r & 3) + 4 * fh];
        l_tot = (float)p.S;
        m_run = -FLT_MAX;
      }
    }
  }
  const float inv = 1.0f / l_tot;
  // The wave's 32 x DH output tile goes through LDS so that HBM sees whole 16-byte-per-lane row
  // segments (8 or 16 lanes = one row's DH*2 bytes) instead of 8-byte pieces of 32 different rows per
  // store instruction.  Every wave has left the tile loop once the barrier below is passed; each wave
  // then uses a private slice of the (now idle) K ring.
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  constexpr int OROW = RB + 16;                       // padded row pitch of the staged tile
  char* ot = smem + wave * (32 * OROW);
#pragma unroll
  for (int n = 0; n < ND; ++n)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      bf16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (bf16)(o[n][4 * rg + e] * inv);
      *reinterpret_cast<bf16x4*>(ot + fr * OROW + (32 * n + 8 * rg + 4 * fh) * 2) = w;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: no barrier needed
  constexpr int LPR = RB / 16;                         // lanes per output row
#pragma unroll
  for (int it = 0; it < 32 * LPR / 64; ++it) {
    const int r = (it * 64 + lane) / LPR, ch = (it * 64 + lane) % LPR;
    const int qrow_ = q0 + wave * 32 + r;
    const bf16x8 vv = *reinterpret_cast<const bf16x8*>(ot + r * OROW + ch * 16);
    if (qrow_ < p.L)
      *reinterpret_cast<bf16x8*>((bf16*)p.out + (int64_t)b * p.o_sb + (int64_t)qrow_ * p.o_sl + head * DH + ch * 8) = vv;
  }
  if (qi < p.L && p.lse && fh == 0)
    p.lse[((int64_t)b * p.h + head) * p.L + qi] = (m_run + log2f(l_tot)) * LN2;
}


// ------------------------------------------------------------------------------------------
// MFMA flash forward for the other head widths (bf16): SigLIP's 72 (run as 96: the missing columns are zeros in
// LDS and never stored) and Gemma's 256 -- the PaliGemma-shape prefill (reference Examples/paligemma.ipynb cells
// 9, 12) -- and anything else up to 256 that is a multiple of 8.  Not the tuned structure of attn_fwd_mfma_kernel
// (these launches are 16 x 256 x 256 and 8 x 264 x 264 problems): register-staged K/V tiles of 64 keys, one tile in
// LDS at a time, 64 query rows per workgroup (16 per wave), mfma_f32_16x16x32_bf16 throughout.
//   * swapped QK^T: A = 16 keys, B = the wave's 16 query rows -> a lane owns ONE query row (lane & 15) and, per
//     16-key block, the 4 consecutive keys 4 * (lane >> 4) .. + 3: the softmax of a row is 16 register values and
//     three lane-group exchanges;
//   * P stays in registers as the B operand of O^T = V^T P^T: a k-step of 32 keys takes the lane's 4 + 4 values of
//     two key blocks, i.e. the contraction index is walked in the order the scores already sit in -- and V^T's
//     fragments are read in the same order by two transposing LDS reads (ds_read_b64_tr_b16) of the row-major tile.
// Masks: causal (+ start_pos) and key padding as in attn_fwd_mfma_kernel, rows without a visible key get the
// reference's uniform average; a dense additive mask goes to the row-wise kernel.
// ------------------------------------------------------------------------------------------
template <int DHP>
__global__ __launch_bounds__(256) void attn_fwd_gen_kernel(AttnParams p, int dh) {
  constexpr int PITCH = (DHP + 8) * 2;       // bytes per LDS row (16 B of padding: conflict-free fragment reads)
  constexpr int KS = DHP / 32;               // k-steps of QK^T
  constexpr int NDB = DHP / 16;              // 16-wide d blocks of O^T
  constexpr int CPRW = DHP / 8;              // 16-byte chunks per row
  constexpr int CPT = 64 * CPRW / 256;       // chunks per thread and tile
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * PITCH];
  char* kt = smem;
  char* vt = smem + 64 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  // heads on x (workgroups go to the XCDs round-robin by linear id: every XCD gets whole heads, so a head's K/V stay in
  // one L2 and -- under a causal mask -- every XCD gets the same mix of short and long rows), long rows first
  const int head = (int)blockIdx.x % p.h, b = (int)blockIdx.x / p.h;
  const int kvh = head / (p.h / p.hk);
  const int q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 64;
  const int qi = q0 + wave * 16 + r16;
  const int qrow = qi < p.L ? qi : p.L - 1;
  const bf16* Q = (const bf16*)p.q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh + (int64_t)qrow * p.q_sl;
  const bf16* Kb = (const bf16*)p.k + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
  const bf16* Vb = (const bf16*)p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;
  const bool causal = p.mask_kind & VY_MASK_CAUSAL;
  const bool haskp = p.mask_kind & VY_MASK_KEYPAD;
  const uint8_t* kp = haskp ? p.keypad + (int64_t)b * p.kp_sb : nullptr;
  const bf16x8 zero8 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};

  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = 32 * ks + 8 * kq;
    qf[ks] = d0 < dh ? *reinterpret_cast<const bf16x8*>(Q + d0) : zero8;
  }
  f32x4 o[NDB];
#pragma unroll
  for (int n = 0; n < NDB; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -FLT_MAX, l_run = 0.f;
  const float c = p.scale * LOG2E;
  int nt = (p.S + 63) / 64;
  if (causal) {
    const int kv_end = min(p.S, p.start_pos + q0 + 64);
    nt = max(1, (kv_end + 63) / 64);
  }
  // transposing-read addresses of the V tile: lane j of a 16-lane group supplies row (j >> 2), columns 4 (j & 3) ..
  const unsigned vtr = vy_lds_addr(vt) + (4 * kq + (r16 >> 2)) * PITCH + (4 * (r16 & 3)) * 2;

  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 64;
    // stage the tile: 16-byte chunks, zero beyond the head width and beyond the last key
    bf16x8 kreg[CPT], vreg[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int cidx = tid + 256 * i;
      const int row = cidx / CPRW, ch = cidx - row * CPRW;
      const int kj = k0 + row;
      const bool ok = ch * 8 < dh && kj < p.S;
      const int64_t kjc = kj < p.S ? kj : p.S - 1;
      kreg[i] = ok ? *reinterpret_cast<const bf16x8*>(Kb + kjc * p.k_sl + (ch * 8 < dh ? ch * 8 : 0)) : zero8;
      vreg[i] = ok ? *reinterpret_cast<const bf16x8*>(Vb + kjc * p.v_sl + (ch * 8 < dh ? ch * 8 : 0)) : zero8;
    }
    __syncthreads();   // the previous tile's fragments have been read
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int cidx = tid + 256 * i;
      const int row = cidx / CPRW, ch = cidx - row * CPRW;
      *reinterpret_cast<bf16x8*>(kt + row * PITCH + ch * 16) = kreg[i];
      *reinterpret_cast<bf16x8*>(vt + row * PITCH + ch * 16) = vreg[i];
    }
    __syncthreads();
    // S^T = K Q^T: four 16-key blocks
    f32x4 sc[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      sc[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (16 * blk + r16) * PITCH + (32 * ks + 8 * kq) * 2);
        sc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sc[blk], 0, 0, 0);
      }
    }
    // masks: register r of block blk is key k0 + 16 blk + 4 kq + r
    float tmax = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = k0 + 16 * blk + 4 * kq + r;
        bool vis = kj < p.S;
        if (causal) vis = vis && kj <= qi + p.start_pos;
        if (haskp) vis = vis && kp[kj < p.S ? kj : 0] != 0;
        const float tv = vis ? sc[blk][r] : -INFINITY;
        sc[blk][r] = tv;
        tmax = fmaxf(tmax, tv);
      }
    // the row's keys are spread over the four lane groups (lane >> 4)
    {
      const unsigned u = __builtin_bit_cast(unsigned, tmax);
      auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
      tmax = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
      const unsigned w = __builtin_bit_cast(unsigned, tmax);
      auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
      tmax = fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
    }
    const float m_new = fmaxf(m_run, tmax * c);   // (-inf * c stays -inf; m_run is finite)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int n = 0; n < NDB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[n][r] *= alpha;
    float rs = 0.f;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(sc[blk][r], c, -m_run));
        sc[blk][r] = e;
        rs += e;
      }
    l_run += rs;   // this lane group's keys only; the four groups are added at the end
    // O^T += V^T P^T, k-steps of 32 keys = blocks (2 tt, 2 tt + 1) in the order the scores sit in
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      bf16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pf[r] = (bf16)sc[2 * tt][r]; pf[4 + r] = (bf16)sc[2 * tt + 1][r]; }
      vy_static_for<NDB>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        union { struct { s16x4 a, b; } h; bf16x8 v; } u;
        u.h.a = vy_lds_tr16_off<n * 32>(vtr + (32 * tt) * PITCH);
        u.h.b = vy_lds_tr16_off<n * 32>(vtr + (32 * tt + 16) * PITCH);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        vy_tie(u.v);
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf, o[n], 0, 0, 0);
      });
    }
  }
  // row totals over the four lane groups
  float l_tot = l_run;
  {
    const unsigned u = __builtin_bit_cast(unsigned, l_tot);
    auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    l_tot = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
    const unsigned w = __builtin_bit_cast(unsigned, l_tot);
    auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    l_tot = __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
  }
  if (haskp && l_tot == 0.f) {
    // no visible key: the reference's softmax over [finfo.min, ...] is uniform over EVERY key -> the column mean of
    // V (rare: a plain walk over the keys for this lane's columns d = 16 n + 4 kq + r)
#pragma unroll
    for (int n = 0; n < NDB; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kj = 0; kj < p.S; ++kj) {
#pragma unroll
      for (int n = 0; n < NDB; ++n) {
        const int d0 = 16 * n + 4 * kq;
        if (d0 < dh) {
          const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(Vb + (int64_t)kj * p.v_sl + d0);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[n][r] += (float)v4[r];
        }
      }
    }
    l_tot = (float)p.S;
    m_run = -FLT_MAX;
  }
  const float inv = 1.0f / l_tot;
  if (qi < p.L) {
    bf16* orow = (bf16*)p.out + (int64_t)b * p.o_sb + (int64_t)qi * p.o_sl + head * dh;
#pragma unroll
    for (int n = 0; n < NDB; ++n) {
      const int d0 = 16 * n + 4 * kq;
      if (d0 < dh) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16)(o[n][r] * inv);
        *reinterpret_cast<bf16x4*>(orow + d0) = w;
      }
    }
    if (p.lse && kq == 0) p.lse[((int64_t)b * p.h + head) * p.L + qi] = (m_run + log2f(l_tot)) * LN2;
  }
}

// ------------------------------------------------------------------------------------------
// row-wise kernel: f32 parity path, unusual head dims, and decode (L == 1)
// one workgroup (4 waves) per (b, head, query row).  CPR lanes share one key row (16 B each).
// ------------------------------------------------------------------------------------------
template <typename T> struct RowVec;
template <> struct RowVec<bf16> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void load(const bf16* p, float* v) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
};
template <> struct RowVec<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void load(const float* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
};

// NWV waves per workgroup: 4, or 16 for decode launches with few (batch x head) rows -- an MQA / small-batch
// decode has only B*h workgroups, each walking its keys in dependent trips; four times the waves = a quarter
// of the trips, combined through LDS as before (no workspace, no second launch)
template <typename T, int CPR, int NWV = 4>  // CPR = dh / VEC lanes per key row, power of two <= 64
__global__ __launch_bounds__(64 * NWV) void attn_rowwise_kernel(AttnParams p, int dh) {
  constexpr int VEC = RowVec<T>::VEC;
  constexpr int KPP = 64 / CPR;  // keys per wave pass
  __shared__ float red_m[NWV * KPP], red_l[NWV * KPP];
  __shared__ float red_o[NWV * 64 * VEC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int kvh = head / (p.h / p.hk);
  const int grp = lane / CPR, ch = lane % CPR;
  const T* Q = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh + (int64_t)qi * p.q_sl;
  const T* Kb = (const T*)p.k + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
  const T* Vb = (const T*)p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;
  // head dims whose chunk count is not a power of two (SigLIP: 72 = 9 chunks of 8) run with the
  // next power of two lanes per key; the surplus lanes carry zeros and never store
  const bool ch_ok = ch * VEC < dh;
  const int chl = ch_ok ? ch : 0;  // a valid address for the idle lanes
  float qv[VEC];
  RowVec<T>::load(Q + chl * VEC, qv);
  if (!ch_ok) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) qv[e] = 0.f;
  }
  const bool causal = p.mask_kind & VY_MASK_CAUSAL;
  const bool haskp = p.mask_kind & VY_MASK_KEYPAD;
  const bool hasadd = p.mask_kind & VY_MASK_ADDITIVE;
  const uint8_t* kp = haskp ? p.keypad + (int64_t)b * p.kp_sb : nullptr;
  const float* am = hasadd ? p.addmask + (int64_t)b * p.am_sb + (int64_t)qi * p.am_sl : nullptr;
  // pure causal: invisible keys contribute exactly 0 -> stop at the diagonal.  With key padding
  // every key is walked so a fully masked row reproduces the reference's uniform average.
  int S_eff = p.pos_dev ? *p.pos_dev + 1 : p.S;
  if (causal && !haskp && !hasadd) S_eff = min(S_eff, qi + p.start_pos + 1);

  float m = -FLT_MAX, l = 0.f, acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  // U key groups per trip: their K and V chunks are all requested before the first dot product, AND the
  // next trip's are requested before this trip's arithmetic (raw 16-byte registers, converted at use): a
  // wave keeps up to 4*U 16-byte loads in flight, so the dependent trips of a row overlap instead of each
  // paying a memory latency (decode is a pure HBM stream of the KV cache; S = 577 is 5 trips)
  constexpr int U = 4;
  constexpr int STRIDE = NWV * KPP * U;
  typedef f32x4 raw16;
  auto key_of = [&](int j0, int u, bool& valid) {
    const int j = j0 + u * NWV * KPP + grp;
    valid = j < S_eff;
    return valid ? j : S_eff - 1;
  };
  auto request = [&](int j0, raw16 (&kr)[U], raw16 (&vr)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool ok;
      const int jc = key_of(j0, u, ok);
      kr[u] = *reinterpret_cast<const raw16*>(Kb + (int64_t)jc * p.k_sl + chl * VEC);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool ok;
      const int jc = key_of(j0, u, ok);
      vr[u] = *reinterpret_cast<const raw16*>(Vb + (int64_t)jc * p.v_sl + chl * VEC);
    }
  };
  auto widen = [&](const raw16& r, float (&o)[VEC]) {
    if constexpr (VEC == 8) {
      const bf16x8 t = __builtin_bit_cast(bf16x8, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (float)t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = r[e];
    }
  };
  raw16 kcur[U], vcur[U];
  int j0 = wave * KPP;
  if (j0 < S_eff) request(j0, kcur, vcur);
  for (; j0 < S_eff; j0 += STRIDE) {
    raw16 knext[U], vnext[U];
    const bool more = j0 + STRIDE < S_eff;   // wave-uniform
    if (more) request(j0 + STRIDE, knext, vnext);
    float t[U];
    bool valid[U];
    int jc[U];
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      jc[u] = key_of(j0, u, valid[u]);
      float kv[VEC];
      widen(kcur[u], kv);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) d = fmaf(qv[e], kv[e], d);
#pragma unroll
      for (int o_ = CPR >> 1; o_ > 0; o_ >>= 1) d += __shfl_xor(d, o_, 64);
      float x = d * p.scale;
      if (hasadd) x = x + am[jc[u]];
      if (causal && jc[u] > qi + p.start_pos) x = -FLT_MAX;
      if (haskp && !kp[jc[u]]) x = -FLT_MAX;
      x = fmaxf(x, -FLT_MAX);
      t[u] = x;
      if (valid[u]) mn = fmaxf(mn, x);
    }
    const float a = __expf(m - mn);
    l *= a;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] *= a;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float e_ = valid[u] ? __expf(t[u] - mn) : 0.f;
      l += e_;
      float vv[VEC];
      widen(vcur[u], vv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = fmaf(e_, vv[e], acc[e]);
    }
    m = mn;
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) { kcur[u] = knext[u]; vcur[u] = vnext[u]; }
    }
  }
  // combine the KPP key groups of the wave (lanes with equal ch), then the 4 waves through LDS
#pragma unroll
  for (int o_ = CPR; o_ < 64; o_ <<= 1) {
    const float m2 = __shfl_xor(m, o_, 64), l2 = __shfl_xor(l, o_, 64);
    const float mn = fmaxf(m, m2);
    const float a1 = __expf(m - mn), a2 = __expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = acc[e] * a1 + __shfl_xor(acc[e], o_, 64) * a2;
    m = mn;
  }
  if (grp == 0) {
    red_m[wave * KPP] = m; red_l[wave * KPP] = l;  // one slot per wave (index reuse of KPP stride)
#pragma unroll
    for (int e = 0; e < VEC; ++e) red_o[(wave * 64 + ch) * VEC + e] = acc[e];
  }
  __syncthreads();
  if (wave == 0 && grp == 0 && ch_ok) {
    float M_ = red_m[0];
    for (int w = 1; w < NWV; ++w) M_ = fmaxf(M_, red_m[w * KPP]);
    float Ls = 0.f, out[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = 0.f;
    for (int w = 0; w < NWV; ++w) {
      const float a = __expf(red_m[w * KPP] - M_);
      Ls += red_l[w * KPP] * a;
#pragma unroll
      for (int e = 0; e < VEC; ++e) out[e] += red_o[(w * 64 + ch) * VEC + e] * a;
    }
    T* O = (T*)p.out + (int64_t)b * p.o_sb + (int64_t)qi * p.o_sl + head * dh + ch * VEC;
    const float inv = 1.0f / Ls;
#pragma unroll
    for (int e = 0; e < VEC; ++e) VyT<T>::st(O + e, out[e] * inv);
    if (p.lse && ch == 0) p.lse[((int64_t)b * p.h + head) * p.L + qi] = M_ + __logf(Ls);
  }
}

template <typename T>
int launch_rowwise(const AttnParams& p, int dh, hipStream_t st, const char* who) {
  constexpr int VEC = RowVec<T>::VEC;
  if (dh % VEC) VY_FAIL(VY_ERR_ARG, "%s: head_dim %d not a multiple of %d", who, dh, VEC);
  int cpr = 1;
  while (cpr * VEC < dh) cpr <<= 1;  // lanes per key row, rounded up to a power of two
  const dim3 grid(p.L, p.h, p.B), block(256);
  static const int wide_wg = [] { const char* e = getenv("VY_DECODE_WIDE_WG"); return e ? atoi(e) : 1; }();
  static const int wide_rows = [] { const char* e = getenv("VY_DECODE_WIDE_ROWS"); return e ? atoi(e) : 128; }();
  if (p.L == 1 && wide_wg && (int64_t)p.B * p.h <= wide_rows && p.S > 64 && std::is_same<T, bf16>::value && (cpr == 8 || cpr == 32)) {
    const dim3 block16(1024);
    if (cpr == 8) hipLaunchKernelGGL((attn_rowwise_kernel<T, 8, 16>), grid, block16, 0, st, p, dh);
    else hipLaunchKernelGGL((attn_rowwise_kernel<T, 32, 16>), grid, block16, 0, st, p, dh);
    VY_CHECK_LAUNCH(who);
    return VY_OK;
  }
#define RW_GO(C) hipLaunchKernelGGL((attn_rowwise_kernel<T, C>), grid, block, 0, st, p, dh)
  switch (cpr) {
    case 1: RW_GO(1); break;
    case 2: RW_GO(2); break;
    case 4: RW_GO(4); break;
    case 8: RW_GO(8); break;
    case 16: RW_GO(16); break;
    case 32: RW_GO(32); break;
    case 64: RW_GO(64); break;
    default: VY_FAIL(VY_ERR_UNSUPPORTED, "%s: head_dim %d unsupported (dh/%d must be a power of two <= 64)", who, dh, VEC);
  }
#undef RW_GO
  VY_CHECK_LAUNCH(who);
  return VY_OK;
}

int check_attn(const char* who, const AttnParams& p, int dh, int dtype) {
  if (!p.q || !p.k || !p.v || !p.out) VY_FAIL(VY_ERR_ARG, "%s: null tensor", who);
  if (p.B <= 0 || p.h <= 0 || p.hk <= 0 || p.L <= 0 || p.S <= 0 || dh <= 0)
    VY_FAIL(VY_ERR_ARG, "%s: empty problem B=%d h=%d hk=%d L=%d S=%d dh=%d", who, p.B, p.h, p.hk, p.L, p.S, dh);
  if (p.h % p.hk) VY_FAIL(VY_ERR_ARG, "%s: h=%d is not a multiple of hk=%d", who, p.h, p.hk);
  if ((p.mask_kind & VY_MASK_KEYPAD) && !p.keypad) VY_FAIL(VY_ERR_ARG, "%s: keypad mask requested but NULL", who);
  if ((p.mask_kind & VY_MASK_ADDITIVE) && !p.addmask) VY_FAIL(VY_ERR_ARG, "%s: additive mask requested but NULL", who);
  const int vec = dtype == VY_BF16 ? 8 : 4;
  const int64_t s[] = {p.q_sb, p.q_sh, p.q_sl, p.k_sb, p.k_sh, p.k_sl, p.v_sb, p.v_sh, p.v_sl, p.o_sb, p.o_sl};
  for (int64_t x : s)
    if (x % vec) VY_FAIL(VY_ERR_ARG, "%s: strides must be multiples of %d elements (16 bytes)", who, vec);
  if ((uintptr_t)p.q % 16 || (uintptr_t)p.k % 16 || (uintptr_t)p.v % 16 || (uintptr_t)p.out % 16)
    VY_FAIL(VY_ERR_ARG, "%s: tensors must be 16-byte aligned", who);
  return 0;
}

}  // namespace

extern "C" int vy_attn_fwd(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sl, const void* k,
                           int64_t k_sb, int64_t k_sh, int64_t k_sl, const void* v, int64_t v_sb,
                           int64_t v_sh, int64_t v_sl, void* out, int64_t o_sb, int64_t o_sl, float* lse,
                           int mask_kind, int64_t start_pos, const uint8_t* keypad, int64_t kp_sb,
                           const float* addmask, int64_t am_sb, int64_t am_sl, int64_t B, int h, int hk,
                           int64_t L, int64_t S, int dh, float scale, int dtype, void* stream) {
  const char* who = "vy_attn_fwd";
  AttnParams p;
  p.q = q; p.q_sb = q_sb; p.q_sh = q_sh; p.q_sl = q_sl;
  p.k = k; p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
  p.v = v; p.v_sb = v_sb; p.v_sh = v_sh; p.v_sl = v_sl;
  p.out = out; p.o_sb = o_sb; p.o_sl = o_sl; p.lse = lse;
  p.mask_kind = mask_kind; p.start_pos = (int)start_pos;
  p.keypad = keypad; p.kp_sb = kp_sb; p.addmask = addmask; p.am_sb = am_sb; p.am_sl = am_sl;
  p.B = (int)B; p.h = h; p.hk = hk; p.L = (int)L; p.S = (int)S; p.scale = scale; p.pos_dev = nullptr;
  static const int diag = [] { const char* e = getenv("VY_ATTN_DIAG"); return e ? atoi(e) : 0; }();
  p.diag = diag;
  if (dtype != VY_BF16 && dtype != VY_F32) VY_FAIL(VY_ERR_ARG, "%s: bad dtype %d", who, dtype);
  if (int rc = check_attn(who, p, dh, dtype)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // (the key-padding visibility words of the MFMA kernel cover 16384 keys)
  if (dtype == VY_BF16 && (dh == 64 || dh == 128) && L > 1 && !((mask_kind & VY_MASK_KEYPAD) && S > 16384)) {
    const dim3 grid((unsigned)(h * B), (unsigned)((L + 127) / 128), 1), block(256);
    if (dh == 64) hipLaunchKernelGGL(attn_fwd_mfma_kernel<64>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_mfma_kernel<128>, grid, block, 0, st, p);
    VY_CHECK_LAUNCH(who);
    return VY_OK;
  }
  static const int gen_on = [] { const char* e = getenv("VY_ATTN_GEN"); return e ? atoi(e) : 1; }();
  if (gen_on && dtype == VY_BF16 && L > 1 && dh % 8 == 0 && dh <= 256 && !(mask_kind & VY_MASK_ADDITIVE)) {
    // other head widths (72 -> 96 columns, ... 256): the general MFMA kernel, 64 query rows per workgroup
    const dim3 grid((unsigned)(h * B), (unsigned)((L + 63) / 64), 1), block(256);
    if (dh <= 96) hipLaunchKernelGGL(attn_fwd_gen_kernel<96>, grid, block, 0, st, p, dh);
    else hipLaunchKernelGGL(attn_fwd_gen_kernel<256>, grid, block, 0, st, p, dh);
    VY_CHECK_LAUNCH(who);
    return VY_OK;
  }
  if (dtype == VY_BF16) return launch_rowwise<bf16>(p, dh, st, who);
  return launch_rowwise<float>(p, dh, st, who);
}

int vy_attn_decode_ex(const void* q, int64_t q_sb, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_sh,
                      int64_t k_sl, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl, void* out,
                      int64_t o_sb, int64_t B, int h, int hk, int64_t S, const int* pos_dev, int dh, float scale,
                      int dtype, void* stream);
int vy_dec_attn(const void* q, int64_t q_sb, const void* k, const void* v, int64_t c_sb, int64_t c_sh, int64_t c_sl, void* out,
                int64_t o_sb, int B, int h, int hk, int64_t S, int64_t smax, const int* pos_dev, int dh, float scale,
                hipStream_t st);

extern "C" int vy_attn_decode(const void* q, int64_t q_sb, int64_t q_sh, const void* k, int64_t k_sb,
                              int64_t k_sh, int64_t k_sl, const void* v, int64_t v_sb, int64_t v_sh,
                              int64_t v_sl, void* out, int64_t o_sb, int64_t B, int h, int hk, int64_t S,
                              int dh, float scale, int dtype, void* stream) {
  return vy_attn_decode_ex(q, q_sb, q_sh, k, k_sb, k_sh, k_sl, v, v_sb, v_sh, v_sl, out, o_sb, B, h, hk, S, nullptr,
                           dh, scale, dtype, stream);
}

int vy_attn_decode_ex(const void* q, int64_t q_sb, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_sh,
                      int64_t k_sl, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sl, void* out,
                      int64_t o_sb, int64_t B, int h, int hk, int64_t S, const int* pos_dev, int dh, float scale,
                      int dtype, void* stream) {
  const char* who = "vy_attn_decode";
  // the resident-context kernel of vy_decode.hip (bf16, head widths 64 / 256, contexts up to 640 / 384 keys); under a
  // graph the context length is read on the device and the cache capacity is not known here: S bounds it
  if (dtype == VY_BF16 && q_sh == dh && k_sb == v_sb && k_sh == v_sh && k_sl == v_sl) {
    const int rc = vy_dec_attn(q, q_sb, k, v, k_sb, k_sh, k_sl, out, o_sb, (int)B, h, hk, S, pos_dev ? k_sh / (k_sl ? k_sl : 1) : S,
                               pos_dev, dh, scale, (hipStream_t)stream);
    if (rc != VY_ERR_UNSUPPORTED) return rc;
  }
  AttnParams p{};
  p.q = q; p.q_sb = q_sb; p.q_sh = q_sh; p.q_sl = (int64_t)h * dh;  // single token: unused stride
  p.k = k; p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
  p.v = v; p.v_sb = v_sb; p.v_sh = v_sh; p.v_sl = v_sl;
  p.out = out; p.o_sb = o_sb; p.o_sl = (int64_t)h * dh; p.lse = nullptr;
  p.mask_kind = VY_MASK_NONE; p.start_pos = 0;
  p.B = (int)B; p.h = h; p.hk = hk; p.L = 1; p.S = (int)S; p.scale = scale; p.pos_dev = pos_dev;
  if (dtype != VY_BF16 && dtype != VY_F32) VY_FAIL(VY_ERR_ARG, "%s: bad dtype %d", who, dtype);
  if (int rc = check_attn(who, p, dh, dtype)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VY_BF16) return launch_rowwise<bf16>(p, dh, st, who);
  return launch_rowwise<float>(p, dh, st, who);
}

// diagnostics: how many workgroups of the tuned forward kernel the runtime places on one CU (registers, LDS)
extern "C" int vy_debug_attn_occupancy(int dh) {
  int n = -1;
  hipError_t e = dh == 64 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_mfma_kernel<64>, 256, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_mfma_kernel<128>, 256, 0);
  return e == hipSuccess ? n : -1;
}
