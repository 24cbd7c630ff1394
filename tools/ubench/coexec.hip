// Micro-benchmark: do fp32-input MFMA and fp32 VALU work co-execute on one SIMD of gfx950?
// 512-thread workgroups (one per CU): waves 0..3 run an MFMA loop, waves 4..7 a VALU loop (wave w and w+4 share a SIMD).
// Modes: 1 = MFMA waves only, 2 = VALU waves only, 3 = both.  Build: hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>  // 0: v_fma_f32, 1: v_pk_fma_f32, 2: bf16 mfma instead of f32 mfma
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (!(mode & 2)) return;
    if (KIND == 0) {
      float v[8];
      for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
      const float m = 1.0000001f, c = 1e-7f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, c);
      }
      float s = 0; for (int j = 0; j < 8; ++j) s += v[j];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
      f2 v[8];
      for (int j = 0; j < 8; ++j) v[j] = f2{threadIdx.x * 1e-3f + j, 1.0f};
      const f2 m = {1.0000001f, 1.0000001f}, c = {1e-7f, 1e-7f};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], m, c);
      }
      f2 s = {0, 0}; for (int j = 0; j < 8; ++j) s += v[j];
      out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y;
    }
  }
}

template <int KIND>
float run(float* d, int iters, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, iters, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, iters, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int iters = 20000;
  // per wave: 4 MFMA / iter (64 cyc each => 256 cyc/iter ideal) ; 32 VALU / iter
  for (int kind = 0; kind < 2; ++kind) {
    float t1 = kind ? run<1>(d, iters, 1) : run<0>(d, iters, 1);
    float t2 = kind ? run<1>(d, iters, 2) : run<0>(d, iters, 2);
    float t3 = kind ? run<1>(d, iters, 3) : run<0>(d, iters, 3);
    printf("%s: mfma-only %.3f ms (%.1f cyc/mfma @2.4GHz)  valu-only %.3f ms (%.2f cyc/inst @2.4GHz)  both %.3f ms  (sum %.3f, max %.3f)\n",
           kind ? "v_pk_fma_f32" : "v_fma_f32   ", t1, t1 * 2.4e6 / (iters * 4.0), t2, t2 * 2.4e6 / (iters * 32.0), t3, t1 + t2, t1 > t2 ? t1 : t2);
  }
  return 0;
}
