// What does the matrix pipe really deliver?  An MFMA-only loop (v_mfma_f32_32x32x16_bf16, `chains`
// independent accumulators per wave, `waves` waves per SIMD) on every CU, timed with the shader clock
// (s_memtime) and the constant 100 MHz clock (s_memrealtime): reports the shader frequency under
// load, cycles per MFMA per SIMD and the resulting TFLOP/s.  Optional LDS read traffic beside it.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock_probe mfma_clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int CH, int LDSR, int RND = 0, int BAR = 0>
__global__ __launch_bounds__(512) void mfma_loop(int iters, unsigned long long* clk, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63;
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane ^ e); }
  bf16x8 a2 = a, b2 = b;
  if (RND) {  // random operands (toggle power), two alternating fragment sets
    unsigned h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u; a[e] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; b[e] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; a2[e] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; b2[e] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
    }
  }
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime: shader clock
  const unsigned long long w0 = wall_clock64();                 // s_memrealtime: 100 MHz
  for (int it = 0; it < iters; ++it) {
    if (LDSR) {
#pragma unroll
      for (int q = 0; q < LDSR; ++q) {
        bf16x8 t = *reinterpret_cast<const bf16x8*>(smem + ((lane * 16 + q * 1024 + it * 64) & 65520));
        a[0] += t[0];
      }
    }
    if (RND == 1 && (it & 1)) {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[c], 0, 0, 0);
    } else {
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    if (BAR && (it & 3) == 3) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][7];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// what does an (almost) empty loop with one workgroup barrier per iteration cost?
template <int NBAR, int NMFMA>
__global__ __launch_bounds__(512) void barrier_loop(int iters, unsigned long long* clk, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[6];
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane ^ e); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NMFMA; ++q) acc[q % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q % 6], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NBAR; ++q) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 6; ++c) s += acc[c][0];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
// the GEMM's k-loop shape without any memory operation: 4 k-steps x (3 W x 2 X fragments -> 6 MFMAs),
// distinct operand registers, s_setprio around each group, one barrier per 24 MFMAs
template <int PRIO, int DISTINCT>
__global__ __launch_bounds__(512) void gemmlike_loop(int iters, unsigned long long* clk, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 wf[2][3], xf[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[q][i][e] = (__bf16)(float)((lane * 7 + e * 3 + i + q * 5) % 13 - 6);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[q][j][e] = (__bf16)(float)((lane * 5 + e + j * 3 + q) % 11 - 5);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(wf[ks & 1][i]));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(xf[ks & 1][j]));
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[DISTINCT ? (ks & 1) : 0][DISTINCT ? i : 0],
                                                             xf[DISTINCT ? (ks & 1) : 0][DISTINCT ? j : 0], acc[i][j], 0, 0, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += acc[i][0][0] + acc[i][1][5];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int PRIO, int DISTINCT>
void run_gemmlike(const char* what) {
  unsigned long long* clk; float* sink;
  const int nwg = 256, iters = 20000;
  (void)hipMalloc(&clk, 4096 * 8); (void)hipMalloc(&sink, 64);
  gemmlike_loop<PRIO, DISTINCT><<<nwg, 512>>>(iters, clk, sink);
  (void)hipDeviceSynchronize();
  unsigned long long h[4096]; (void)hipMemcpy(h, clk, nwg * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (int i = 0; i < nwg; ++i) cyc += h[i];
  printf("%-44s %7.1f cycles per 24-MFMA stage (ideal 1536)\n", what, cyc / nwg / iters);
  (void)hipFree(clk); (void)hipFree(sink);
}

template <int NBAR, int NMFMA>
void run_bar(int threads, int nwg, const char* what) {
  unsigned long long* clk; float* sink;
  (void)hipMalloc(&clk, 4096 * 8); (void)hipMalloc(&sink, 64);
  const int iters = 20000;
  barrier_loop<NBAR, NMFMA><<<nwg, threads>>>(iters, clk, sink);
  (void)hipDeviceSynchronize();
  unsigned long long h[4096]; (void)hipMemcpy(h, clk, nwg * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (int i = 0; i < nwg; ++i) cyc += h[i];
  printf("%-44s threads %3d, %4d workgroups: %7.1f cycles per iteration\n", what, threads, nwg, cyc / nwg / iters);
  (void)hipFree(clk); (void)hipFree(sink);
}

template <int CH, int LDSR, int RND = 0, int BAR = 0>
void run(int threads, int iters, const char* what) {
  const int nwg = 256;
  unsigned long long* clk; float* sink;
  (void)hipMalloc(&clk, nwg * 16); (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<CH, LDSR, RND, BAR><<<nwg, threads>>>(iters / 10, clk, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<CH, LDSR, RND, BAR><<<nwg, threads>>>(iters, clk, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2 * 256]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0; for (int i = 0; i < nwg; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  cyc /= nwg; wall /= nwg;
  const double mhz = cyc / (wall / 100.0);                  // wall ticks are 10 ns
  const double waves_per_simd = threads / 64 / 4.0;
  const double mfma_per_simd = (double)iters * CH * waves_per_simd;
  const double tf = (double)iters * CH * (threads / 64) * nwg * 32768.0 / (ms * 1e-3) * 1e-12;
  printf("%-34s threads %3d chains %d: %7.3f ms  shader clock %6.0f MHz  %5.1f cycles/MFMA/SIMD  %7.1f TFLOP/s\n",
         what, threads, CH, ms, mhz, cyc / mfma_per_simd, tf);
  hipFree(clk); hipFree(sink);
}

int main() {
  const int iters = 40000;
  run<6, 0>(512, iters, "MFMA only");
  run<6, 0>(256, iters, "MFMA only");
  run<4, 0>(512, iters, "MFMA only");
  run<12, 0>(256, iters, "MFMA only");
  run<6, 5>(512, iters, "MFMA + 5 ds_read_b128 per 6 MFMA");
  run<6, 0>(512, iters * 4, "MFMA only, long");
  run<6, 0, 1>(512, iters * 4, "random operands, long");
  run<6, 0, 1, 1>(512, iters * 4, "random + barrier per 24 MFMA");
  run<6, 0, 0, 1>(512, iters * 4, "constant + barrier per 24 MFMA");
  run<6, 0, 2>(512, iters * 4, "random, same fragments every time");
  run<6, 0, 1>(256, iters * 4, "random operands, 1 wave/SIMD");
  run<2, 0, 1>(512, iters * 8, "random operands, 2 chains");
  run_bar<1, 0>(512, 256, "empty loop, 1 barrier");
  run_bar<1, 0>(256, 256, "empty loop, 1 barrier");
  run_bar<1, 0>(640, 256, "empty loop, 1 barrier");
  run_bar<1, 0>(512, 1, "empty loop, 1 barrier");
  run_bar<2, 0>(512, 256, "empty loop, 2 barriers");
  run_bar<0, 0>(512, 256, "empty loop, no barrier");
  run_bar<1, 24>(512, 256, "24 MFMA per wave + 1 barrier (ideal 1536)");
  run_bar<0, 24>(512, 256, "24 MFMA per wave, no barrier (ideal 1536)");
  run_bar<1, 6>(512, 256, "6 MFMA per wave + 1 barrier (ideal 384)");
  run_gemmlike<0, 0>("GEMM-like, one operand pair, no setprio");
  run_gemmlike<0, 1>("GEMM-like, distinct operands, no setprio");
  run_gemmlike<1, 1>("GEMM-like, distinct operands, setprio");
  return 0;
}
