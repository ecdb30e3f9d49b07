// What does one link of a decode-step kernel chain cost?  Chains of dependent launches on one stream,
// timed with events (per-launch average), for kernels that do: nothing; one global-memory round trip;
// two / three dependent round trips; a round trip + LDS reduce + store -- at 8, 96, 192 and 384 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int HOPS, bool WORK>
__global__ __launch_bounds__(256) void link(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ idx, int n) {
  if (HOPS == 0) { if (in == nullptr) out[0] = 1.f; return; }
  const int t = blockIdx.x * 256 + threadIdx.x;
  int j = t % n;
  float acc = 0.f;
#pragma unroll
  for (int h = 0; h < HOPS; ++h) {
    const float v = in[j];            // dependent: the next index comes from memory too
    acc += v;
    j = idx[(j + (int)v) % n];
  }
  if (WORK) {
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64) acc = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
  }
  out[t % n] = acc;
}

template <int HOPS, bool WORK>
void run(const char* what, int wgs, float* a, float* b, int* idx, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int chain = 200;
  for (int w = 0; w < 20; ++w) link<HOPS, WORK><<<wgs, 256>>>(a, b, idx, n);
  (void)hipEventRecord(e0);
  for (int i = 0; i < chain; ++i) {
    link<HOPS, WORK><<<wgs, 256>>>(i & 1 ? b : a, i & 1 ? a : b, idx, n);
  }
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // the same chain captured in a graph
  hipStream_t st; (void)hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < chain; ++i) link<HOPS, WORK><<<wgs, 256, 0, st>>>(i & 1 ? b : a, i & 1 ? a : b, idx, n);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
  float msg; (void)hipEventElapsedTime(&msg, e0, e1);
  printf("%-44s %4d WGs: %6.2f us per launch eager, %6.2f us in a graph\n", what, wgs, ms * 1e3 / chain, msg * 1e3 / (5 * chain));
}

// alternate DIFFERENT kernels, as a decode step does (I-cache / kernel-object switches), optionally with a big
// kernel-argument block and LDS
struct Big { float a[60]; };
template <int ID>
__global__ __launch_bounds__(256) void alt_link(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ idx, int n, Big big) {
  __shared__ float red[ID == 0 ? 256 : 2048];
  const int t = blockIdx.x * 256 + threadIdx.x;
  float v = in[t % n] + big.a[ID];
  red[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < 64) v = red[threadIdx.x] + red[threadIdx.x + 64 + ID];
  // some distinct code per ID so the kernels do not share instructions
#pragma unroll
  for (int i = 0; i < 40 + 20 * ID; ++i) v = v * 1.0001f + (float)(i ^ ID);
  out[t % n] = v;
}
void run_alt(int wgs, float* a, float* b, int* idx, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  Big big{};
  const int chain = 200;
  hipStream_t st; (void)hipStreamCreate(&st);
  auto enqueue = [&](hipStream_t s_) {
    for (int i = 0; i < chain; ++i) {
      const float* in = i & 1 ? b : a; float* out = i & 1 ? a : b;
      switch (i % 5) {
        case 0: alt_link<0><<<wgs, 256, 0, s_>>>(in, out, idx, n, big); break;
        case 1: alt_link<1><<<wgs * 2, 256, 0, s_>>>(in, out, idx, n, big); break;
        case 2: alt_link<2><<<wgs, 256, 0, s_>>>(in, out, idx, n, big); break;
        case 3: alt_link<3><<<32, 256, 0, s_>>>(in, out, idx, n, big); break;
        default: alt_link<4><<<wgs, 256, 0, s_>>>(in, out, idx, n, big); break;
      }
    }
  };
  enqueue(st); (void)hipStreamSynchronize(st);
  (void)hipEventRecord(e0, st); enqueue(st); (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  enqueue(st);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
  float msg; (void)hipEventElapsedTime(&msg, e0, e1);
  printf("%-44s %4d WGs: %6.2f us per launch eager, %6.2f us in a graph\n", "5 different kernels alternating", wgs, ms * 1e3 / chain, msg * 1e3 / (5 * chain));
}

int main() {
  const int n = 1 << 20;
  float *a, *b; int* idx;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMalloc(&idx, n * 4);
  std::vector<float> h(n, 0.f); std::vector<int> hi(n);
  unsigned s = 1; for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hi[i] = (int)(s >> 12) % n; }
  (void)hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice);
  for (int wgs : {8, 96, 192, 384}) {
    run<0, false>("empty kernel", wgs, a, b, idx, n);
    run<1, false>("1 memory round trip + store", wgs, a, b, idx, n);
    run<2, false>("2 dependent round trips + store", wgs, a, b, idx, n);
    run<3, false>("3 dependent round trips + store", wgs, a, b, idx, n);
    run<1, true>("1 round trip + LDS reduce + store", wgs, a, b, idx, n);
  }
  run_alt(96, a, b, idx, n);
  run_alt(192, a, b, idx, n);
  return 0;
}
