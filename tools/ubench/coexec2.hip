// Micro-benchmark 2: which VALU instruction classes co-execute with fp32-input MFMA on one SIMD of gfx950?
// 512-thread workgroups (one per CU): waves 0..3 run an MFMA loop, waves 4..7 (same SIMDs) a loop of ONE instruction class:
//   0 v_fma_f32   1 v_xor/v_add_u32 (integer)   2 v_mad_u64_u32 (Philox multiply)   3 v_exp_f32 (transcendental)
//   4 v_cndmask / v_med3 (select/clamp)          5 ds_read_b32 (LDS)
// plus SAME-WAVE interleaving: mode 4 = one wave issues the MFMAs with the VALU instructions of class KIND between them.
// Build: hipcc --offload-arch=gfx950 -O3 coexec2.hip -o coexec2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__device__ __forceinline__ void valu_block(float (&v)[8], unsigned (&u)[8], const float* lds) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (KIND == 0) v[j] = __builtin_fmaf(v[j], 1.0000001f, 1e-7f);
    if (KIND == 1) u[j] = (u[j] ^ 0x9E3779B9u) + 0x7F4A7C15u;
    if (KIND == 2) { unsigned long long p = (unsigned long long)u[j] * 0xD2511F53u; u[j] = (unsigned)(p >> 32) ^ (unsigned)p; }
    if (KIND == 3) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.5f;
    if (KIND == 4) v[j] = __builtin_amdgcn_fmed3f(v[j] + 0.0f, -1.0f, 1.0f);
    if (KIND == 5) v[j] = lds[(u[j] & 1023)] ;
  }
}

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  __shared__ float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = i * 1e-3f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  float v[8]; unsigned u[8];
  for (int j = 0; j < 8; ++j) { v[j] = threadIdx.x * 1e-3f + j; u[j] = threadIdx.x * 2654435761u + j; }
  if (mode == 4) {  // same wave: 4 MFMA + 32 VALU per iteration, interleaved
    if (wave >= 4) return;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); valu_block<KIND>(v, u, lds);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0); valu_block<KIND>(v, u, lds);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); valu_block<KIND>(v, u, lds);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0); valu_block<KIND>(v, u, lds);
    }
    float s = a0[0] + a1[1] + a2[2] + a3[3];
    for (int j = 0; j < 8; ++j) s += v[j] + u[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    return;
  }
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
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) valu_block<KIND>(v, u, lds);
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += v[j] + u[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
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

template <int KIND>
void report(float* d, const char* name) {
  const int iters = 20000;
  const float t1 = run<KIND>(d, iters, 1), t2 = run<KIND>(d, iters, 2), t3 = run<KIND>(d, iters, 3), t4 = run<KIND>(d, iters, 4);
  printf("%-14s mfma-only %.3f ms  valu-only %.3f ms (%.2f cyc/inst @2.4GHz)  two waves: %.3f ms  same wave interleaved: %.3f ms   (sum %.3f, max %.3f)\n",
         name, t1, t2, t2 * 2.4e6 / (iters * 32.0), t3, t4, t1 + t2, t1 > t2 ? t1 : t2);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  report<0>(d, "v_fma_f32");
  report<1>(d, "v_xor+v_add");
  report<2>(d, "v_mad_u64_u32");
  report<3>(d, "v_exp_f32+mul");
  report<4>(d, "v_add+v_med3");
  report<5>(d, "ds_read_b32");
  return 0;
}
