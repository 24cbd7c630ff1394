// Helpers shared by the fused training-backward kernels (sdeh_bwdf.hip: teams split by channel tile; sdeh_bwdf2.hip: teams split
// by trajectory tile): LDS geometry, accumulator-layout <-> plane moves, the partial-gradient record of one team.
#pragma once
#include "sdeh_bwd.hpp"

namespace sdeh {

namespace bwdf {

constexpr int C = 64;
constexpr int kMaxLH = 3;  // hidden layers the kernel is instantiated for: 1 .. 3 (num_layers 3 .. 5; the shipped configurations have 2)
constexpr int RSW = 68;             // row stride of the [., 64] weight copies: 4 x odd -> conflict-free ds_read_b128 across 16 rows
constexpr int RS = 36;              // row stride of the exchange planes [row][32 trajectories]
constexpr int PLANE = 64 * RS;
constexpr int TABS = 6 * 64;        // (mu, 1/sigma^2) x {prior, second, target}
template <int OTD> constexpr int rsi() { return OTD == 1 ? 36 : 68; }  // row stride of input_embed.weight [64][d]
template <int OTD, int LH> constexpr int lds_floats() { return 64 * rsi<OTD>() + LH * 64 * RSW + 32 * OTD * RSW + LH * 64 + 64 + TABS + 2 * 4 * PLANE; }
// partial-gradient record of one team
template <int OTD> constexpr int off_whid() { return 64 * 32 * OTD; }
template <int OTD, int LH> constexpr int off_wout() { return off_whid<OTD>() + LH * 4096; }
template <int OTD, int LH> constexpr int off_bhid() { return off_wout<OTD, LH>() + 32 * OTD * 64; }
template <int OTD, int LH> constexpr int off_bout() { return off_bhid<OTD, LH>() + LH * 64; }
template <int OTD, int LH> constexpr int wsize() { return off_bout<OTD, LH>() + 32 * OTD; }

__device__ __forceinline__ int rrow(int q) { return (q & 3) + 8 * (q >> 2); }

// accumulator-layout tile <-> plane [row][trajectory]
__device__ __forceinline__ void plane_put(float* __restrict__ plane, int tile, int j, int h, const f32x16& v) {
  float* __restrict__ p = plane + (32 * tile + 4 * h) * RS + j;
#pragma unroll
  for (int q = 0; q < 16; ++q) p[rrow(q) * RS] = v[q];
}
// transposed read: lane (i, h) gets row 32 tile + i, trajectories 8 c + 4 h .. + 3
__device__ __forceinline__ float4 plane_getT(const float* __restrict__ plane, int tile, int i, int h, int c) {
  return *reinterpret_cast<const float4*>(plane + (32 * tile + i) * RS + 8 * c + 4 * h);
}
// 16 values of a per-row table in accumulator order: p = &table[32 tile + 4 h]
__device__ __forceinline__ f32x16 rows16(const float* __restrict__ p) {
  f32x16 v;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 t = *reinterpret_cast<const float4*>(p + 8 * m);
    v[4 * m] = t.x; v[4 * m + 1] = t.y; v[4 * m + 2] = t.z; v[4 * m + 3] = t.w;
  }
  return v;
}

// out = init + W[rows of this lane's tile][:] . B     (wrow = &W[(32 R + i) * ld + 4 h]; k-group s covers columns 8 s + 4 h .. + 3 of W
// and rows 8 s + 4 h .. + 3 of the B operand, which is read from its plane [row][trajectory] group by group: bcol = &plane[4 h * RS + j]).
// One accumulator: a dependent v_mfma_f32_32x32x2_f32 issues every 69.5 cycles instead of 64.6 (profiles/r02_ubench.txt) -- cheaper
// than a second accumulator's registers and adds.
template <int NG>
__device__ __forceinline__ f32x16 mm_rows(const float* __restrict__ wrow, const float* __restrict__ bcol, int ng, const f32x16& init) {
  f32x16 acc = init;
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    if (s < ng) {
      const float4 w = *reinterpret_cast<const float4*>(wrow + 8 * s);
      const float* __restrict__ bp = bcol + 8 * s * RS;
      const float b0 = bp[0], b1 = bp[RS], b2 = bp[2 * RS], b3 = bp[3 * RS];
      acc = SDEH_MFMA(w.x, b0, acc);
      acc = SDEH_MFMA(w.y, b1, acc);
      acc = SDEH_MFMA(w.z, b2, acc);
      acc = SDEH_MFMA(w.w, b3, acc);
    }
  }
  return acc;
}

// out = W[:, columns of this lane's tile]^T . B     (wcol = &W[(4 h) * LD + 32 R + i]; k-group s covers rows 8 s + 4 h .. + 3 of both)
template <int NG, int LD>
__device__ __forceinline__ f32x16 mm_cols(const float* __restrict__ wcol, const float* __restrict__ bcol, int ng) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    if (s < ng) {
      const float* __restrict__ p = wcol + 8 * s * LD;
      const float* __restrict__ bp = bcol + 8 * s * RS;
      const float w0 = p[0], w1 = p[LD], w2 = p[2 * LD], w3 = p[3 * LD];
      const float b0 = bp[0], b1 = bp[RS], b2 = bp[2 * RS], b3 = bp[3 * RS];
      acc = SDEH_MFMA(w0, b0, acc);
      acc = SDEH_MFMA(w1, b1, acc);
      acc = SDEH_MFMA(w2, b2, acc);
      acc = SDEH_MFMA(w3, b3, acc);
    }
  }
  return acc;
}

// acc0 (+ acc1) += delta[tile R] . a[tile c0 (, c0 + 1)]^T over the 32 trajectories of the planes; bsum += this lane's 16 delta
// values (lane (i, h): row 32 R + i, trajectories with bit 2 == h -- summed over h at the end they are the bias gradient)
template <bool TWO>
__device__ __forceinline__ void dw_acc(const float* __restrict__ dplane, int R, const float* __restrict__ aplane, int c0,
                                       f32x16& acc0, f32x16& acc1, float& bsum, int i, int h) {
  float4 dv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) dv[c] = plane_getT(dplane, R, i, h, c);
#pragma unroll
  for (int c = 0; c < 4; ++c) bsum += (dv[c].x + dv[c].y) + (dv[c].z + dv[c].w);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 a0 = plane_getT(aplane, c0, i, h, c);
    float4 a1 = a0;
    if constexpr (TWO) a1 = plane_getT(aplane, c0 + 1, i, h, c);
    acc0 = SDEH_MFMA(dv[c].x, a0.x, acc0);
    if constexpr (TWO) acc1 = SDEH_MFMA(dv[c].x, a1.x, acc1);
    acc0 = SDEH_MFMA(dv[c].y, a0.y, acc0);
    if constexpr (TWO) acc1 = SDEH_MFMA(dv[c].y, a1.y, acc1);
    acc0 = SDEH_MFMA(dv[c].z, a0.z, acc0);
    if constexpr (TWO) acc1 = SDEH_MFMA(dv[c].z, a1.z, acc1);
    acc0 = SDEH_MFMA(dv[c].w, a0.w, acc0);
    if constexpr (TWO) acc1 = SDEH_MFMA(dv[c].w, a1.w, acc1);
  }
}

// act(z) and act'(z) of one tile
template <int ACT>
__device__ __forceinline__ void act_both(const f32x16& z, f32x16& a, f32x16& g) {
  if constexpr (ACT == SDEH_ACT_GELU_ERF) {
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      f2 av, gv;
      act_gelu2_both(f2{z[q], z[q + 1]}, av, gv);
      a[q] = av.x; a[q + 1] = av.y;
      g[q] = gv.x; g[q + 1] = gv.y;
    }
  } else if constexpr (ACT == SDEH_ACT_SILU) {
#pragma unroll
    for (int q = 0; q < 16; ++q) { float av, gv; act_silu_both(z[q], av, gv); a[q] = av; g[q] = gv; }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = act_grad(z[q], ACT);
    a = z;
    act_tile<ACT>(a);
  }
}

// accumulator tile -> natural [rows][ld] matrix block (row tile R, column tile Cc): coalesced over the lanes
__device__ __forceinline__ void store_tile(float* __restrict__ m, int ld, int R, int Cc, int j, int h, const f32x16& v) {
  float* __restrict__ p = m + (32 * R + 4 * h) * ld + 32 * Cc + j;
#pragma unroll
  for (int q = 0; q < 16; ++q) p[rrow(q) * ld] = v[q];
}

}  // namespace bwdf

}  // namespace sdeh
