// Does a chain link get slower when every launch streams memory that has not been touched for a long time
// (HBM + address-translation misses), as the weights of a decode step are?  Graph-captured chains of one kernel
// that reads `bytes_per_wg` per workgroup from a region that moves by `stride` per link inside a big buffer.
//   hipcc --offload-arch=gfx950 -O3 -o cold_chain_probe cold_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f4;

template <int U>
__global__ __launch_bounds__(256) void stream_link(const f4* __restrict__ w, float* __restrict__ out, const float* __restrict__ in,
                                                   long per_wg_vec) {
  __shared__ float red[256];
  const f4* p = w + (long)blockIdx.x * per_wg_vec + threadIdx.x;
  float acc = in[threadIdx.x];   // dependency on the previous link
  for (long i = 0; i < per_wg_vec; i += 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    acc = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
  }
}

int main() {
  const size_t big = (size_t)3 << 30;   // 3 GiB
  char* buf; float *a, *b;
  if (hipMalloc(&buf, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(buf, 0, big);
  (void)hipMalloc(&a, 1 << 20); (void)hipMalloc(&b, 1 << 20);
  (void)hipMemset(a, 0, 1 << 20); (void)hipMemset(b, 0, 1 << 20);
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int chain = 200;
  for (int wgs : {32, 144, 192, 384}) {
    for (long kb : {0L, 4L, 24L, 96L}) {            // KiB per workgroup
      for (int cold = 0; cold < 2; ++cold) {
        const long per_wg_vec = kb * 1024 / 16;
        const size_t link_bytes = (size_t)wgs * kb * 1024;
        const size_t stride = cold ? ((link_bytes + (8u << 20)) & ~((size_t)(2u << 20) - 1)) + (2u << 20) : 0;   // >= 8 MiB apart
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        size_t off = 0;
        for (int i = 0; i < chain; ++i) {
          stream_link<6><<<wgs, 256, 0, st>>>((const f4*)(buf + off), i & 1 ? a : b, i & 1 ? b : a, per_wg_vec);
          off += stride;
          if (off + link_bytes > big) off = 0;
        }
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%4d WGs x %3ld KiB  %-22s %6.2f us per link  (%.2f TB/s)\n", wgs, kb, cold ? "fresh region per link" : "same region", ms * 1e3 / (5 * chain),
               link_bytes / (ms * 1e-3 / (5 * chain)) * 1e-12);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
      }
    }
  }
  return 0;
}
