// Micro-benchmark of the wide kernels' MFMA layer loop in isolation (sdeh_wide.hip: wide_layer): one workgroup per CU, four waves,
// each running `reps` layers of NS4 = 32 k-groups (C = 256) over an LDS plane, weights streamed from a 256 KB L2-resident buffer.
// Prints cycles per k-group against the ideal 256 NT CT.   Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize wide_loop.hip -o wide_loop
#include "../../sde_sampler_amd/csrc/sdeh_wide.hip"
#include <cstdio>
using namespace sdeh;

template <int NT, int CT>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ w, float* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, j = lane & 31;
  constexpr int RS = 32 * CT;
  for (int i = tid; i < 256 * RS; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  unsigned voff[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) voff[k] = (unsigned)(((wv + 4 * k) * 64 + lane) * 16);
  f32x16 acc[NT][CT];
  float s = 0.0f;
  for (int r = 0; r < reps; ++r) {
    WidePre<NT> P;
    wide_prefetch<NT>(P, w, 8 * 256, 32, voff);
    wide_layer<NT, CT>(P, w, 8 * 256, 32, voff, lds + h * RS + j, RS, acc);
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
      for (int c = 0; c < CT; ++c) s += acc[k][c][0];
  }
  out[blockIdx.x * 256 + tid] = s;
}

template <int NT, int CT>
void run(const float* w, float* d) {
  const int reps = 400;
  const size_t lds_bytes = 256 * 32 * CT * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<NT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop_kernel<NT, CT>), dim3(256), dim3(256), lds_bytes, 0, w, d, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((loop_kernel<NT, CT>), dim3(256), dim3(256), lds_bytes, 0, w, d, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 2.4e6 / (reps * 32.0);
  printf("NT=%d CT=%d: %.3f ms, %.0f cycles per k-group (ideal %d) -> %.3f of the MFMA rate\n", NT, CT, ms, cyc, 256 * NT * CT, 256.0 * NT * CT / cyc);
}

int main() {
  float *w, *d;
  hipMalloc(&w, 32 * 8 * 256 * 4 + 65536); hipMemset(w, 0, 32 * 8 * 256 * 4 + 65536);
  hipMalloc(&d, 256 * 256 * 4);
  run<2, 1>(w, d); run<2, 2>(w, d); run<1, 1>(w, d); run<1, 2>(w, d); run<2, 4>(w, d);
  return 0;
}
