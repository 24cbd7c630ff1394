// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 on one wave per SIMD as a function of the number of independent
// accumulators it cycles through (1 = every MFMA depends on the previous one, 2 = the wide kernels at C=256 with one column tile).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a[NACC];
  for (int i = 0; i < NACC; ++i) a[i] = f32x16{};
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += a[i][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%d accumulator(s): %.3f ms for %d MFMAs per wave = %.1f cycles/MFMA @2.4GHz\n", NACC, ms, iters * 8, ms * 2.4e6 / (iters * 8.0));
}

int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  run<1>(d); run<2>(d); run<4>(d); run<8>(d);
  return 0;
}
