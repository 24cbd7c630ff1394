// Log-domain Sinkhorn sweeps (device side of sdeh_sinkhorn, include/sdeh.h): the [n, m] reductions the reference hands to
// pykeops LazyTensors (eval/sinkhorn.py:112-178) -- the only place it reaches third-party GPU kernels.
//
//   half-iteration:  pot_p[i] = eps (log_w_p[i] - logsumexp_j((pot_q[j] - M_ij) / eps)),   M_ij = ||P_i - Q_j||_p
//   distance:        sum_ij exp((pot_p[i] + pot_q[j] - M_ij) / eps) M_ij, and argmax_j of the plan per row
//
// Never materialises M: a workgroup owns 64 rows of P (one per lane, coordinates in registers) and streams Q through LDS
// in tiles of 256 points; its four waves each take a quarter of every tile (broadcast ds_read_b128: all lanes read the
// same Q_j) and keep an online (max, sum) per lane, merged through LDS at the end.  The j range can additionally be
// split over blockIdx.y (partials combined by sink_finalize_kernel) so that small clouds still fill the chip.
// Distances are formed from coordinate DIFFERENCES on the VALU -- |x|^2 + |y|^2 - 2 x.y on the matrix pipe would lose
// the absolute accuracy that eps = 1e-3 needs ((v - M)/eps amplifies an error in M a thousandfold).
#pragma once
#include "sdeh_common.hpp"

namespace sdeh {

constexpr int kSinkTile = 256;  // Q points per LDS tile
constexpr int kSinkChunk = 8;   // online-softmax chunk

template <int DP>
__device__ __forceinline__ float sink_dist(const float (&x)[DP], const float* __restrict__ q, int pnorm) {
  constexpr int NQ = (DP + 3) / 4;
  const float4* q4 = reinterpret_cast<const float4*>(q);
  float acc = 0.0f;
  if (pnorm == 2) {
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const float4 v = q4[b];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (4 * b + c < DP) { const float t = x[4 * b + c] - e[c]; acc = fmaf(t, t, acc); }
    }
    return __builtin_amdgcn_sqrtf(acc);
  }
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    const float4 v = q4[b];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (4 * b + c < DP) acc += fabsf(x[4 * b + c] - e[c]);
  }
  return acc;
}

// MODE 0: partial logsumexp (part_m, part_s);  MODE 1: per-row transport cost + argmax (pot_p required)
template <int DP, bool PAD, int MODE>
__global__ __launch_bounds__(256) void sink_sweep_kernel(const SinkArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int DS = 4 * ((DP + 3) / 4);  // LDS row stride of a Q point
  if (A.done != nullptr && *A.done != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = PAD ? A.d : DP;
  const long long row = (long long)blockIdx.x * 64 + lane;
  const bool live = row < A.np;
  const long long lrow = live ? row : A.np - 1;
  float x[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) x[c] = (!PAD || c < d) ? A.P[lrow * d + (PAD ? min(c, d - 1) : c)] : 0.0f;
  const float up = MODE == 1 ? A.pot_p[lrow] : 0.0f;

  // this block's share of the Q range, in whole tiles
  const long long tiles = (A.nq + kSinkTile - 1) / kSinkTile;
  const long long t0 = tiles * blockIdx.y / gridDim.y, t1 = tiles * (blockIdx.y + 1) / gridDim.y;
  float* tile = lds;                       // [kSinkTile][DS]
  float* tpot = lds + kSinkTile * DS;      // [kSinkTile]
  float m = -INFINITY, s = 0.0f;           // MODE 0: online logsumexp;  MODE 1: s = sum P M, m = best logit
  int best = 0;
  for (long long t = t0; t < t1; ++t) {
    const long long j0 = t * kSinkTile;
    __syncthreads();
    for (int e = tid; e < kSinkTile * DS; e += 256) {
      const int jj = e / DS, c = e % DS;
      const long long j = j0 + jj;
      tile[e] = (j < A.nq && c < d) ? A.Q[j * d + c] : 0.0f;
    }
    {
      const long long j = j0 + tid;
      tpot[tid] = j < A.nq ? A.pot_q[j] : -INFINITY;  // padding points: logit -inf
    }
    __syncthreads();
    const int jb = wave * (kSinkTile / 4);
#pragma unroll 1
    for (int c0 = 0; c0 < kSinkTile / 4; c0 += kSinkChunk) {
      float lg[kSinkChunk], dist[kSinkChunk];
#pragma unroll
      for (int k = 0; k < kSinkChunk; ++k) {
        const int jj = jb + c0 + k;
        dist[k] = sink_dist<DP>(x, tile + jj * DS, A.pnorm);
        lg[k] = (tpot[jj] - dist[k]) * A.inv_eps;  // (-M_ij + v_j) / eps
      }
      if (MODE == 0) {
        float cm = lg[0];
#pragma unroll
        for (int k = 1; k < kSinkChunk; ++k) cm = fmaxf(cm, lg[k]);
        const float mn = fmaxf(m, cm);
        if (mn > -INFINITY) {  // wave-divergent only on all-padding chunks
          s *= __expf(m - mn);
#pragma unroll
          for (int k = 0; k < kSinkChunk; ++k) s += __expf(lg[k] - mn);
          m = mn;
        }
      } else {
#pragma unroll
        for (int k = 0; k < kSinkChunk; ++k) {
          const float l = fmaf(up, A.inv_eps, lg[k]);  // (-M + u_i + v_j) / eps
          if (lg[k] > m) { m = lg[k]; best = (int)(j0 + jb + c0 + k); }
          s = fmaf(__expf(l), dist[k], s);
        }
      }
    }
  }
  // merge the four waves' per-row results
  __syncthreads();
  float* xm = lds;            // [4][64]
  float* xs = lds + 256;      // [4][64]
  int* xi = reinterpret_cast<int*>(lds + 512);
  xm[wave * 64 + lane] = m;
  xs[wave * 64 + lane] = s;
  if (MODE == 1) xi[wave * 64 + lane] = best;
  __syncthreads();
  if (wave != 0) return;
  if (MODE == 0) {
    float mm = fmaxf(fmaxf(xm[lane], xm[64 + lane]), fmaxf(xm[128 + lane], xm[192 + lane]));
    float ss = 0.0f;
    if (mm > -INFINITY) {
#pragma unroll
      for (int w = 0; w < 4; ++w) ss += xs[w * 64 + lane] * __expf(xm[w * 64 + lane] - mm);
    }
    if (live) {
      A.part_m[(long long)blockIdx.y * A.np + row] = mm;
      A.part_s[(long long)blockIdx.y * A.np + row] = ss;
    }
  } else {
    float ss = 0.0f, bm = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      ss += xs[w * 64 + lane];
      if (xm[w * 64 + lane] > bm) { bm = xm[w * 64 + lane]; bi = xi[w * 64 + lane]; }
    }
    if (live && A.corr != nullptr) A.corr[row] = (long long)bi;
    // block partial of the transport cost (dead lanes contribute 0)
    float v = live ? ss : 0.0f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0 && A.part_s != nullptr) A.part_s[blockIdx.x] = v;
  }
}

template <int DP, bool PAD>
int launch_sink(const SinkArgs& a, int mode, int splits, hipStream_t stream) {
  constexpr int DS = 4 * ((DP + 3) / 4);
  const size_t lds_floats = (size_t)kSinkTile * DS + kSinkTile;
  const size_t lds_bytes = (lds_floats > 768 ? lds_floats : 768) * sizeof(float);
  const dim3 grid((unsigned)((a.np + 63) / 64), (unsigned)(mode == 0 ? splits : 1));
  if (mode == 0) hipLaunchKernelGGL((sink_sweep_kernel<DP, PAD, 0>), grid, dim3(256), lds_bytes, stream, a);
  else hipLaunchKernelGGL((sink_sweep_kernel<DP, PAD, 1>), grid, dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
