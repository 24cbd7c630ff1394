// Micro-benchmark: what does a DEPENDENT packed fp32 instruction cost a wave that is alone on its SIMD?  (The V wave of the trajectory
// kernel runs alone while its M wave waits for x: the mixture loops are chains  t = y - m;  acc = fma(t, t, acc).)
//   mode 0: one chain of v_pk_fma_f32 (every instruction needs the previous result)      mode 1 / 2: two / four interleaved chains
//   mode 3: the logit loop as written  (sub, fma(dep on the sub and on acc two back), two accumulators)
//   mode 4: the same work, subs issued four pairs ahead of their fmas                    mode 5: as 3 with plain v_fma_f32 pairs (unpacked)
// one wave per SIMD (256 threads per workgroup, one workgroup per CU) and two waves per SIMD (512 threads).
// Build: hipcc --offload-arch=gfx950 -O3 valu_dep.hip -o valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
// asm volatile: the instruction sequences below are exactly what executes (nothing hoisted out of the loop, nothing re-ordered)
__device__ __forceinline__ f2 FMA(f2 a, f2 b, f2 c) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f2 SUB(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float FMA1(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float SUB1(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float s) {
  f2 y[8], m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { y[j] = f2{threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f - j}; m[j] = f2{s * j, s + j}; }
  f2 a0 = {0, 0}, a1 = {0, 0}, a2 = {0, 0}, a3 = {0, 0};
  for (int it = 0; it < iters; ++it) {
    FENCE();
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a0 = FMA(a0, y[j & 7], m[j & 7]);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a0 = FMA(a0, y[j], m[j]); a1 = FMA(a1, y[j], m[j]); }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { a0 = FMA(a0, y[j], m[j]); a1 = FMA(a1, y[j], m[j]); a2 = FMA(a2, y[j + 4], m[j]); a3 = FMA(a3, y[j + 4], m[j]); }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f2 t = SUB(y[j], m[j]);
        FENCE();
        if (j & 1) a1 = FMA(t, t, a1); else a0 = FMA(t, t, a0);
        FENCE();
      }
    } else if (MODE == 4) {
      f2 t[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = SUB(y[j], m[j]);
      FENCE();
#pragma unroll
      for (int j = 0; j < 4; ++j) { t[j + 4] = SUB(y[j + 4], m[j + 4]); FENCE(); if (j & 1) a1 = FMA(t[j], t[j], a1); else a0 = FMA(t[j], t[j], a0); FENCE(); }
#pragma unroll
      for (int j = 4; j < 8; ++j) { if (j & 1) a1 = FMA(t[j], t[j], a1); else a0 = FMA(t[j], t[j], a0); }
    } else {
      float b0 = a0.x, b1 = a0.y, b2 = a1.x, b3 = a1.y;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t0 = SUB1(y[j].x, m[j].x), t1 = SUB1(y[j].y, m[j].y);
        FENCE();
        if (j & 1) { b2 = FMA1(t0, t0, b2); b3 = FMA1(t1, t1, b3); } else { b0 = FMA1(t0, t0, b0); b1 = FMA1(t1, t1, b1); }
        FENCE();
      }
      a0 = f2{b0, b1}; a1 = f2{b2, b3};
    }
    FENCE();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a3.y;
}

template <int MODE>
void run(const char* name, int threads, float* d_out) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d_out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  const double inst = 16.0 * (MODE == 5 ? 2 : 1) * iters;  // vector instructions per wave
  printf("%-66s %d waves/SIMD: %.3f ms = %.2f cycles per instruction per SIMD @2.4 GHz\n", name, threads / 256, best,
         best * 1e-3 * 2.4e9 / (inst * (threads / 256)));
}

int main() {
  float* d_out; hipMalloc(&d_out, 256 * 512 * 4);
  for (int th : {256, 512}) {
    run<0>("one chain of v_pk_fma_f32", th, d_out);
    run<1>("two interleaved chains", th, d_out);
    run<2>("four interleaved chains", th, d_out);
    run<3>("logit loop as written (pk_sub, dependent pk_fma, 2 accumulators)", th, d_out);
    run<4>("logit loop, subs four pairs ahead", th, d_out);
    run<5>("logit loop with unpacked v_sub_f32 / v_fma_f32", th, d_out);
  }
  return 0;
}
