// Micro-benchmark + layout check: a Gaussian mixture's logit and score contractions as v_mfma_f32_4x4x1_16b_f32 INSIDE the wave that
// owns the trajectories (lane = trajectory, coordinates in registers -- the V wave's T layout):
//   * the instruction is 16 independent 4 x 4 x 1 outer products; B = one value per lane (block b, column j = lane 4 b + j), the result
//     D[i][j] lands in register i of the same lane: with B = x_d of the lane's own trajectory, register i accumulates
//     sum_d A[i] x_d -- four components' logits of the lane's own trajectory.  No layout change, no exchange.
//   * A is wave-uniform table data: CBSZ = 4 broadcasts the A values of block ABID to all 16 blocks, so ONE operand register holds
//     the A values of 16 different instructions (lanes 4 n .. 4 n + 3 = instruction n's four rows): a table is read from LDS exactly once
//     per step (8 ds_read_b128 for 40 x 50), not once per lane.
//   K = 40 components x D = 50 coordinates: 10 x 50 = 500 instructions for the logits, 13 x 40 = 520 for the score; 8 cycles each.
// Checks both contractions against a float64 host evaluation, then times them (one / two waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 40, D = 50, KG = K / 4, DG = (D + 3) / 4;   // 10 component groups, 13 coordinate groups
constexpr int N1 = D * KG, N2 = K * DG;                        // instructions per contraction
constexpr int V1 = (N1 + 15) / 16, V2 = (N2 + 15) / 16;        // operand registers (16 instructions each)
constexpr int Q1 = (V1 + 3) / 4, Q2 = (V2 + 3) / 4;            // ds_read_b128 per contraction

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// img1[q][lane][4]: register v = 4 q + e, lane (b, i): table[4 g + i][d] with n = 16 v + b = d * KG + g
// img2[q][lane][4]: register v, lane (b, i): table[k][4 g' + i] with n = 16 v + b = k * DG + g'
template <bool TIME>
__global__ __launch_bounds__(512) void k(const float* __restrict__ img1, const float* __restrict__ img2, const float* __restrict__ xin,
                                        float* __restrict__ logit_out, float* __restrict__ p_out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* l1 = reinterpret_cast<f32x4*>(lds);
  f32x4* l2 = l1 + Q1 * 64;
  for (int i = threadIdx.x; i < Q1 * 64; i += blockDim.x) l1[i] = reinterpret_cast<const f32x4*>(img1)[i];
  for (int i = threadIdx.x; i < Q2 * 64; i += blockDim.x) l2[i] = reinterpret_cast<const f32x4*>(img2)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float x[D];
#pragma unroll
  for (int d = 0; d < D; ++d) x[d] = xin[(TIME ? (long long)lane : row) * D + d];
  f32x4 lg[KG], P[DG];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < KG; ++g) lg[g] = f32x4{0, 0, 0, 0};
    sfor<Q1>([&](auto Qc) {
      constexpr int q = decltype(Qc)::value;
      const f32x4 a = l1[q * 64 + lane];
      sfor<64>([&](auto Nc) {
        constexpr int n = 64 * q + decltype(Nc)::value;
        if constexpr (n < N1) {
          constexpr int d = n / KG, g = n % KG, e = (n / 16) % 4, b = n % 16;
          lg[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], x[d], lg[g], 4, b, 0);
        }
      });
    });
    // (a stand-in for the softmax: the real kernel exponentiates here)
    float ev[K];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) ev[4 * g + i] = TIME ? lg[g][i] * 1e-3f : lg[g][i];
#pragma unroll
    for (int g = 0; g < DG; ++g) P[g] = f32x4{0, 0, 0, 0};
    sfor<Q2>([&](auto Qc) {
      constexpr int q = decltype(Qc)::value;
      const f32x4 a = l2[q * 64 + lane];
      sfor<64>([&](auto Nc) {
        constexpr int n = 64 * q + decltype(Nc)::value;
        if constexpr (n < N2) {
          constexpr int kk = n / DG, g = n % DG, e = (n / 16) % 4, b = n % 16;
          P[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], ev[kk], P[g], 4, b, 0);
        }
      });
    });
    if (TIME) {
#pragma unroll
      for (int d = 0; d < D; ++d) x[d] += 1e-6f * P[d / 4][d % 4];
    }
  }
  if (!TIME) {
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) logit_out[row * K + 4 * g + i] = lg[g][i];
#pragma unroll
    for (int d = 0; d < D; ++d) p_out[row * D + d] = P[d / 4][d % 4];
  } else if (x[0] == 123.456f) logit_out[row] = x[1];
}

int main() {
  std::vector<float> tab(K * 52, 0.f), x(512 * D);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (int k = 0; k < K; ++k) for (int d = 0; d < D; ++d) tab[k * 52 + d] = rnd() * 4.f;
  for (auto& v : x) v = rnd() * 2.f;
  std::vector<float> img1(Q1 * 64 * 4, 0.f), img2(Q2 * 64 * 4, 0.f);
  for (int n = 0; n < N1; ++n) { const int d = n / KG, g = n % KG, v = n / 16, b = n % 16;
    for (int i = 0; i < 4; ++i) img1[((v / 4) * 64 + 4 * b + i) * 4 + v % 4] = tab[(4 * g + i) * 52 + d]; }
  for (int n = 0; n < N2; ++n) { const int kk = n / DG, g = n % DG, v = n / 16, b = n % 16;
    for (int i = 0; i < 4; ++i) img2[((v / 4) * 64 + 4 * b + i) * 4 + v % 4] = 4 * g + i < D ? tab[kk * 52 + 4 * g + i] : 0.f; }
  float *d1, *d2, *dx, *dl, *dp;
  hipMalloc(&d1, img1.size() * 4); hipMalloc(&d2, img2.size() * 4); hipMalloc(&dx, x.size() * 4);
  hipMalloc(&dl, 256 * 512 * K * 4); hipMalloc(&dp, 256 * 512 * D * 4);
  hipMemcpy(d1, img1.data(), img1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d2, img2.data(), img2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  const size_t sh = (size_t)(Q1 + Q2) * 64 * 16;
  hipLaunchKernelGGL(k<false>, dim3(1), dim3(512), sh, 0, d1, d2, dx, dl, dp, 1);
  std::vector<float> hl(512 * K), hp(512 * D);
  hipMemcpy(hl.data(), dl, hl.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hp.data(), dp, hp.size() * 4, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, m1 = 0, m2 = 0;
  for (int r = 0; r < 512; ++r) {
    double lg[K];
    for (int kk = 0; kk < K; ++kk) { double a = 0; for (int d = 0; d < D; ++d) a += (double)tab[kk * 52 + d] * x[r * D + d]; lg[kk] = a;
      e1 = fmax(e1, fabs(a - hl[r * K + kk])); m1 = fmax(m1, fabs(a)); }
    for (int d = 0; d < D; ++d) { double a = 0; for (int kk = 0; kk < K; ++kk) a += (double)hl[r * K + kk] * tab[kk * 52 + d];
      e2 = fmax(e2, fabs(a - hp[r * D + d])); m2 = fmax(m2, fabs(a)); }
  }
  printf("layout check over 512 trajectories: logits max |err| %.3e (scale %.1f), score contraction max |err| %.3e (scale %.1f)\n", e1, m1, e2, m2);
  for (int th : {256, 512}) {
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k<true>, dim3(256), dim3(th), sh, 0, d1, d2, dx, dl, dp, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (r > 0 && ms < best) best = ms;
    }
    printf("%d waves/SIMD: %.3f ms for %d steps = %.0f cycles per step per SIMD @2.4 GHz (%d + %d matrix instructions per wave and step: %.1f cycles each)\n",
           th / 256, best, iters, best * 1e-3 * 2.4e9 / iters, N1, N2, best * 1e-3 * 2.4e9 / iters / ((N1 + N2) * (th / 256)));
  }
  return 0;
}
