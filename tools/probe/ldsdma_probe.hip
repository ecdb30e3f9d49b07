// L2 -> LDS ingest probe: how many GB/s can one CU take in through global_load_lds_dwordx4?
//   pattern 0: each wave-instruction reads 1 KiB contiguous
//   pattern 1: each wave-instruction reads 8 rows x 128 B, rows `stride` bytes apart (GEMM k-slice)
//   pattern 2: 4 rows x 256 B
// depth = LDS-DMA instructions kept in flight per wave (counted vmcnt).  Every workgroup sweeps the
// same `span` bytes (L2-resident when span is a few MB), starting at a different offset.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))

template <int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ buf, size_t span, int iters, int pattern, int stride,
                                             int shared_by, float* sink, int lag_iters = 0) {
  __shared__ __attribute__((aligned(16))) char smem[512 / 64 * DEPTH * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned mask = (unsigned)span - 1;  // span is a power of two: no divisions in the loop
  // workgroups in groups of `shared_by` read the same addresses (operand sharing in a GEMM)
  // consecutive workgroup ids land on different XCDs (round-robin): a sharing group must sit on ONE XCD
  const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, group = xcd * 32 + slot / shared_by;
  unsigned row0 = (group * 8 + wave) * 8u * (unsigned)stride * 17u;
  unsigned col = 0;
  // followers of a sharing group trail the leader by lag_iters iterations: they start that many
  // iterations "behind" in the address stream (the leader fetched those lines earlier)
  const int my_lag = ((blockIdx.x >> 3) % shared_by) * lag_iters;
  // patterns 3/4/5: GEMM-like -- the DEPTH instructions of an iteration read DIFFERENT rows at the same
  // column (seg3 = 128 / 256 / 512 bytes per row per iteration), the column advances per iteration
  const int seg3 = pattern == 3 ? 128 : (pattern == 4 ? 256 : 512), lpr3 = seg3 / 16, rpi3 = 64 / lpr3;
  const int rpi = pattern == 1 ? 8 : 4, seg = pattern == 1 ? 128 : 256;   // rows / bytes per row per instruction
  const unsigned lane_off = pattern == 0 ? lane * 16
                          : (pattern == 1 ? (lane >> 3) * stride + (lane & 7) * 16 : (lane >> 4) * stride + (lane & 15) * 16);
  for (int i = -my_lag; i < iters - my_lag; ++i) {
    if (i < 0) {  // followers idle (a dependent ALU spin) while the leader runs ahead
      asm volatile("s_sleep 62");  // ~4000 cycles, about one iteration
      continue;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      unsigned a = (row0 + col + lane_off) & mask;
      if (pattern >= 3) a = (row0 + (unsigned)(d * rpi3 + lane / lpr3) * stride + col + (lane % lpr3) * 16) & mask;
      __builtin_amdgcn_global_load_lds((const GLB void*)(buf + a), (LDS void*)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
      if (pattern == 0) { col += 1024; }
      else if (pattern >= 3) { }
      else {
        col += seg;
        if (col >= (unsigned)stride) { col = 0; row0 += rpi * stride; }
      }
    }
    if (pattern >= 3) {
      col += seg3;
      if (col >= (unsigned)stride) { col = 0; row0 += DEPTH * rpi3 * stride; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (smem[threadIdx.x] == 123 && iters < 0) sink[0] = 1.f;
}

int main(int argc, char** argv) {
  const size_t span = (size_t)(argc > 1 ? atoi(argv[1]) : 4) << 20;
  char* buf; float* sink;
  hipMalloc(&buf, span + (1 << 20)); hipMemset(buf, 1, span + (1 << 20)); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  const int stride = (argc > 2 ? atoi(argv[2]) : 1536);
  for (int pattern : {1, 3, 4, 5})
    for (int shared_by : {1, 4}) {
      const int depth = 7, lag = 0;
      auto launch = [&](int it) { hipLaunchKernelGGL(probe<7>, dim3(256), dim3(512), 0, 0, buf, span, it, pattern, stride, shared_by, sink, lag); };
      launch(20); hipDeviceSynchronize();
      hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * 8 * depth * 1024.0 * iters;
      printf("span %zu MB stride %d pattern %d shared_by %d: %7.1f us  %6.2f TB/s  %6.1f GB/s/CU\n", span >> 20, stride, pattern, shared_by, ms * 1e3,
             bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
    }
  return 0;
}
