// Micro-benchmark for the split-precision ("bf16 x 3") FourierMLP mode, gfx950.
//
// Part A -- co-execution.  512-thread workgroups, one per CU; waves 0..3 loop over a matrix instruction (4 independent accumulators),
// waves 4..7 (same SIMDs) over a vector-ALU instruction class; timed alone and together.  profiles/r01_ubench_coexec.txt found that
// v_mfma_f32_32x32x2_f32 and fp32 vector work take the SUM of their times (one datapath).  Here: v_mfma_f32_32x32x16_bf16 against
// v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_cvt + v_sub (the split itself), and the two matrix instructions against each other.
//
// Part B -- accuracy.  One 64 x 64 layer (the C = 64 hidden layers of models/mlp.py:114-122) on 32 columns: y = W a with
//   fp32   : v_mfma_f32_32x32x2_f32 (bit-wise an fmaf chain: the shipped kernels)
//   3 prod : a = a1 + a2 + a3, W = w1 + w2 + w3 in bf16 pieces; products w1 a1 + w1 a2 + w2 a1         (error ~ 2^-16 per product)
//   6 prod : + w2 a2 + w1 a3 + w3 a1                                                                  (error ~ 2^-24: fp32 level)
// all accumulated in fp32 by v_mfma_f32_32x32x16_bf16, against the float64 product.
//
// Build: hipcc --offload-arch=gfx950 -O3 bf16x3.hip -o bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu_block(float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (KIND == 0) v[j] = __builtin_fmaf(v[j], 1.0000001f, 1e-7f);
    if (KIND == 1) { /* v_pk_fma_f32 on pairs */ }
    if (KIND == 2) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.5f;
    if (KIND == 3) {  // one step of the split: hi = bf16(v) (round to nearest even by integer arithmetic), v <- v - hi
      unsigned u = __float_as_uint(v[j]);
      unsigned r = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
      v[j] = v[j] - __uint_as_float(r) + 1.0f;
    }
  }
  if (KIND == 1) {
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f2 a = {v[j], v[j + 1]}, b = {1.0000001f, 1.0000001f}, c = {1e-7f, 1e-7f}, d;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
      v[j] = d.x; v[j + 1] = d.y;
    }
  }
}

// MM: 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_32x32x2_f32.  mode bit 0: matrix waves run, bit 1: vector waves run,
// mode 4: ONE wave interleaves 4 matrix instructions with 4 vector blocks, mode 8: waves 0..3 bf16 MFMA, waves 4..7 fp32 MFMA
template <int KIND, int MM>
__global__ __launch_bounds__(512) void coexec(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8 xa, xb;
  for (int j = 0; j < 8; ++j) { xa[j] = (__bf16)(0.001f * (threadIdx.x + j)); xb[j] = (__bf16)(1.0f + 0.01f * j); }
  const float fx = threadIdx.x * 1e-3f, fy = 1.0f + threadIdx.x * 1e-4f;
  auto mm = [&](f32x16& acc, bool bf) {
    if (bf) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, acc, 0, 0, 0);
  };
  if (mode == 4) {
    if (wave >= 4) return;
    for (int i = 0; i < iters; ++i) {
      mm(a0, MM == 0); valu_block<KIND>(v);
      mm(a1, MM == 0); valu_block<KIND>(v);
      mm(a2, MM == 0); valu_block<KIND>(v);
      mm(a3, MM == 0); valu_block<KIND>(v);
    }
  } else if (mode == 8) {
    const bool bf = wave < 4;
    for (int i = 0; i < iters; ++i) { mm(a0, bf); mm(a1, bf); mm(a2, bf); mm(a3, bf); }
  } else if (wave < 4) {
    if (!(mode & 1)) return;
    for (int i = 0; i < iters; ++i) { mm(a0, MM == 0); mm(a1, MM == 0); mm(a2, MM == 0); mm(a3, MM == 0); }
  } else {
    if (!(mode & 2)) return;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) valu_block<KIND>(v);
    }
  }
  float s = a0[0] + a1[1] + a2[2] + a3[3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int MM>
float run(float* d, int iters, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((coexec<KIND, MM>), dim3(256), dim3(512), 0, 0, d, iters, mode);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((coexec<KIND, MM>), dim3(256), dim3(512), 0, 0, d, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

template <int KIND, int MM>
void report(float* d, const char* name) {
  const int iters = 20000;
  const float m = run<KIND, MM>(d, iters, 1), v = run<KIND, MM>(d, iters, 2), b = run<KIND, MM>(d, iters, 3), s = run<KIND, MM>(d, iters, 4);
  printf("%-28s %-26s: matrix-only %.3f ms (%.1f cyc/inst @2.4GHz)  vector-only %.3f ms (%.2f cyc/inst)  both %.3f ms (sum %.3f, max %.3f)  "
         "same wave interleaved %.3f ms\n", MM == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", name, m, m * 2.4e6 / (4.0 * iters),
         v, v * 2.4e6 / (32.0 * iters), b, m + v, m > v ? m : v, s);
}

// ---------------------------------------------------------------------------------------------------------------- part B
__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }  // round to nearest even

// one wave: y[64 x 32] = W[64 x 64] a[64 x 32] in the four arithmetic forms; lane (j, h)
__global__ __launch_bounds__(64) void layer(const float* W, const float* a, float* y32, float* y3, float* y6, float* y1) {
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  for (int R = 0; R < 2; ++R) {
    f32x16 c32 = {}, c1 = {}, c3 = {}, c6 = {};
    for (int k = 0; k < 64; k += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(W[(32 * R + j) * 64 + k + h], a[(k + h) * 32 + j], c32, 0, 0, 0);
    for (int k0 = 0; k0 < 64; k0 += 16) {  // 32x32x16: lane (j, h) supplies k = k0 + 8 h .. + 7 of row / column j
      bf16x8 w[3], b[3];
      for (int e = 0; e < 8; ++e) {
        float wv = W[(32 * R + j) * 64 + k0 + 8 * h + e], av = a[(k0 + 8 * h + e) * 32 + j];
        for (int p = 0; p < 3; ++p) {
          w[p][e] = to_bf16(wv); wv -= (float)w[p][e];
          b[p][e] = to_bf16(av); av -= (float)b[p][e];
        }
      }
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[0], c1, 0, 0, 0);
      // smallest products first (the accumulator rounds after every instruction)
      c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], b[0], c6, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[2], c6, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], b[1], c6, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], b[0], c6, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[1], c6, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[0], c6, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], b[0], c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[1], c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], b[0], c3, 0, 0, 0);
    }
    for (int q = 0; q < 16; ++q) {
      const int row = 32 * R + (q & 3) + 8 * (q >> 2) + 4 * h;
      y32[row * 32 + j] = c32[q]; y1[row * 32 + j] = c1[q]; y3[row * 32 + j] = c3[q]; y6[row * 32 + j] = c6[q];
    }
  }
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
  printf("# Part A: co-execution on one SIMD (256 workgroups x 512 threads, 20000 iterations of 4 matrix / 32 vector instructions per wave)\n");
  report<0, 0>(d, "v_fma_f32");
  report<1, 0>(d, "v_pk_fma_f32");
  report<2, 0>(d, "v_exp_f32 (+ v_mul)");
  report<3, 0>(d, "split step (int + 2 add)");
  report<0, 1>(d, "v_fma_f32");
  {
    const int iters = 20000;
    const float mb = run<0, 0>(d, iters, 1), mf = run<0, 1>(d, iters, 1), both = run<0, 0>(d, iters, 8);
    printf("bf16 MFMA waves 0..3 + fp32 MFMA waves 4..7 on the same SIMDs: bf16-only %.3f ms, fp32-only %.3f ms, both %.3f ms (sum %.3f)\n", mb, mf, both,
           mb + mf);
  }
  printf("# Part B: one 64 x 64 layer on 32 columns against float64 (weights ~ U(-1/8, 1/8) as nn.Linear(64, 64) initialises, activations ~ N(0, 1))\n");
  std::vector<float> W(64 * 64), a(64 * 32);
  srand(3);
  auto uni = [] { return rand() / (float)RAND_MAX; };
  for (auto& w : W) w = (2.0f * uni() - 1.0f) * 0.125f;
  for (auto& x : a) { float u1 = uni() * 0.999f + 1e-3f, u2 = uni(); x = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); }
  float *dW, *da, *dy[4];
  hipMalloc(&dW, W.size() * 4); hipMalloc(&da, a.size() * 4);
  for (auto& p : dy) hipMalloc(&p, 64 * 32 * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layer, dim3(1), dim3(64), 0, 0, dW, da, dy[0], dy[1], dy[2], dy[3]);
  hipDeviceSynchronize();
  const char* names[4] = {"fp32 MFMA (shipped)", "bf16 x 3 pieces, 3 products", "bf16 x 3 pieces, 6 products", "plain bf16 (1 product)"};
  for (int f = 0; f < 4; ++f) {
    std::vector<float> y(64 * 32);
    hipMemcpy(y.data(), dy[f], y.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, esum = 0, ymax = 0;
    for (int r = 0; r < 64; ++r)
      for (int c = 0; c < 32; ++c) {
        double ref = 0;
        for (int k = 0; k < 64; ++k) ref += (double)W[r * 64 + k] * (double)a[k * 32 + c];
        const double e = fabs(y[r * 32 + c] - ref);
        emax = e > emax ? e : emax; esum += e; ymax = fabs(ref) > ymax ? fabs(ref) : ymax;
      }
    printf("%-30s: max |err| %.3e  mean |err| %.3e  (max |y| %.3f; fp32 ulp of 1: 1.2e-7)\n", names[f], emax, esum / (64 * 32), ymax);
  }
  return 0;
}
