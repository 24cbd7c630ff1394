// Helpers of the trajectory-split fused backward kernels (sdeh_bwdf2.hip; sdeh_bridgef.hip): register-fed layers, the chunk ring of the
// weight-gradient products, the paired stage (one transposed / forward layer interleaved with one gradient tile).
#pragma once
#include "sdeh_bwdf.hpp"

namespace sdeh {

namespace bwdf2 {
using namespace bwdf;

// operands of one chunk (8 trajectories: 4 k-steps) of a weight-gradient product; chunk 4 w + c lives in the planes of the team's wave w
struct DwChunk {
  float4 dv, a0;
};

// lane (i, h): delta row 32 R + i, a row 32 c0 + i, trajectories 8 c + 4 h .. + 3 of one wave's tile (wplanes: that wave's D plane, A behind it)
__device__ __forceinline__ void dw_load(const float* __restrict__ planes, int R, int c0, int i, int h, int k, DwChunk& o) {
  const float* __restrict__ Dp = planes + (k >> 2) * 2 * PLANE;
  o.dv = plane_getT(Dp, R, i, h, k & 3);
  o.a0 = plane_getT(Dp + PLANE, c0, i, h, k & 3);
}

// the first NQ registers of an accumulator-layout tile -> plane [row][trajectory]
template <int NQ>
__device__ __forceinline__ void plane_put_n(float* __restrict__ plane, int tile, int j, int h, const f32x16& v) {
  float* __restrict__ p = plane + (32 * tile + 4 * h) * RS + j;
#pragma unroll
  for (int q = 0; q < NQ; ++q) p[rrow(q) * RS] = v[q];
}
// the value of lane j (lower half) in both halves
__device__ __forceinline__ float bcast_lo(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]);
}

__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

// one chunk of a product (4 matrix instructions)
__device__ __forceinline__ void dw_chunk(const DwChunk& o, f32x16& acc) {
  acc = SDEH_MFMA(o.dv.x, o.a0.x, acc);
  acc = SDEH_MFMA(o.dv.y, o.a0.y, acc);
  acc = SDEH_MFMA(o.dv.z, o.a0.z, acc);
  acc = SDEH_MFMA(o.dv.w, o.a0.w, acc);
}

// Forward layer from registers:  o[R] += sum_{s < ng} sum_e W[32 R + i][8 s + 4 h + e] * b[s >> 2][4 (s & 3) + e]   (R < NR)
// wrow = &W[i * ld + 4 h]; the second row tile is 32 ld floats further.  One accumulator per row tile, k order (s, e) as in
// sdeh_bwdf.hip's mm_rows and the forward kernel: the pre-activations are the forward launch's bit for bit.
template <int NG, int NB, int NR>
__device__ __forceinline__ void fwd_rows(const float* __restrict__ wrow, int ld, const f32x16 (&b)[NB], int ng, f32x16 (&o)[NR]) {
  float4 w[2][NR];
#pragma unroll
  for (int R = 0; R < NR; ++R) w[0][R] = *reinterpret_cast<const float4*>(wrow + 32 * R * ld);
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    if (s < ng) {
      if (s + 1 < NG && s + 1 < ng) {
#pragma unroll
        for (int R = 0; R < NR; ++R) w[(s + 1) & 1][R] = *reinterpret_cast<const float4*>(wrow + 32 * R * ld + 8 * (s + 1));
      }
      const f32x16& bt = b[s >> 2];
      const int q0 = 4 * (s & 3);
#pragma unroll
      for (int R = 0; R < NR; ++R) o[R] = SDEH_MFMA(w[s & 1][R].x, bt[q0], o[R]);
#pragma unroll
      for (int R = 0; R < NR; ++R) o[R] = SDEH_MFMA(w[s & 1][R].y, bt[q0 + 1], o[R]);
#pragma unroll
      for (int R = 0; R < NR; ++R) o[R] = SDEH_MFMA(w[s & 1][R].z, bt[q0 + 2], o[R]);
#pragma unroll
      for (int R = 0; R < NR; ++R) o[R] = SDEH_MFMA(w[s & 1][R].w, bt[q0 + 3], o[R]);
      SDEH_FENCE();
    }
  }
}

// One backward stage, entered behind the barrier that made the team's (delta_k, a_k) planes visible: the transposed layer from registers
//     o[R] = sum_{s < ng} sum_e W[(8 s + 4 h + e) * LD + 32 R + i] * b[s >> 2][4 (s & 3) + e]        (R < NR; wcol = &W[4 h * LD + i])
// issued interleaved with the chunks of ONE tile of the weight-gradient product  acc += delta[row tile Rd] a[tile c0]^T  over the
// trajectories of NCH / 4 waves (planes: the first of them; operands one chunk ahead through a ring).  dsum[w] += this lane's delta
// values of wave w's trajectories (bias gradients, d loss / d emb[t]).  Ends with the barrier behind which the planes may be overwritten.
template <int NGC, int LD, int NR, int NB, int NCH>
__device__ __forceinline__ void stage_cols(const float* __restrict__ wcol, const f32x16 (&b)[NB], int ng, f32x16 (&o)[NR],
                                           const float* __restrict__ planes, int Rd, int c0, int i, int h,
                                           f32x16& acc, float (&dsum)[NCH / 4]) {
  constexpr int CPI = NCH / 8;  // chunks per iteration
#pragma unroll
  for (int R = 0; R < NR; ++R)
#pragma unroll
    for (int q = 0; q < 16; ++q) o[R][q] = 0.0f;
  float wc[2][4][NR];
  DwChunk ck[2][CPI];
#pragma unroll
  for (int c = 0; c < CPI; ++c) dw_load(planes, Rd, c0, i, h, c, ck[0][c]);
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int R = 0; R < NR; ++R) wc[0][e][R] = wcol[e * LD + 32 * R];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const bool chain = s < NGC && s < ng;
    if (s + 1 < 8) {
#pragma unroll
      for (int c = 0; c < CPI; ++c) dw_load(planes, Rd, c0, i, h, CPI * (s + 1) + c, ck[(s + 1) & 1][c]);
    }
    if (s + 1 < NGC && s + 1 < ng) {
      const float* __restrict__ p = wcol + 8 * (s + 1) * LD;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int R = 0; R < NR; ++R) wc[(s + 1) & 1][e][R] = p[e * LD + 32 * R];
    }
    if (chain) {
      const f32x16& bt = b[(s >> 2) < NB ? (s >> 2) : 0];
      const int q0 = 4 * (s & 3);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int R = 0; R < NR; ++R) o[R] = SDEH_MFMA(wc[s & 1][e][R], bt[q0 + e], o[R]);
    }
#pragma unroll
    for (int c = 0; c < CPI; ++c) {
      dsum[(CPI * s + c) >> 2] += sum4(ck[s & 1][c].dv);
      dw_chunk(ck[s & 1][c], acc);
    }
    SDEH_FENCE();
  }
  ws_barrier();
}

// o[R] += sum_{s < NG} sum_e W[(8 s + 4 h + e) * LD + 32 R + i] * b[s >> 2][4 (s & 3) + e]   (two row tiles; the input layer on the
// transposed copy of input_embed.weight: the same operands in the same order as fwd_rows on the natural copy)
template <int LD, int NG, int NB>
__device__ __forceinline__ void chain_cols_n(const float* __restrict__ wcol, const f32x16 (&b)[NB], f32x16 (&o)[2]) {
  float wc[2][4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int R = 0; R < 2; ++R) wc[0][e][R] = wcol[e * LD + 32 * R];
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    if (s + 1 < NG) {
      const float* __restrict__ p = wcol + 8 * (s + 1) * LD;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int R = 0; R < 2; ++R) wc[(s + 1) & 1][e][R] = p[e * LD + 32 * R];
    }
    const f32x16& bt = b[(s >> 2) < NB ? (s >> 2) : 0];
    const int q0 = 4 * (s & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int R = 0; R < 2; ++R) o[R] = SDEH_MFMA(wc[s & 1][e][R], bt[q0 + e], o[R]);
    SDEH_FENCE();
  }
}

// The same pairing with a FORWARD layer as the chain:  o[R] = sum_s sum_e W[(32 R + i) * ld + 8 s + 4 h + e] * b[s >> 2][4 (s & 3) + e]
// (wrow = &W[i * ld + 4 h]; two row tiles), next to the chunks of one gradient tile over the team's four waves.
template <int NCH>
__device__ __forceinline__ void stage_rows(const float* __restrict__ wrow, int ld, const f32x16 (&b)[2], f32x16 (&o)[2],
                                           const float* __restrict__ planes, int Rd, int c0, int i, int h,
                                           f32x16& acc, float (&dsum)[NCH / 4]) {
  constexpr int CPI = NCH / 8;
#pragma unroll
  for (int R = 0; R < 2; ++R)
#pragma unroll
    for (int q = 0; q < 16; ++q) o[R][q] = 0.0f;
  float4 w[2][2];
  DwChunk ck[2][CPI];
#pragma unroll
  for (int c = 0; c < CPI; ++c) dw_load(planes, Rd, c0, i, h, c, ck[0][c]);
#pragma unroll
  for (int R = 0; R < 2; ++R) w[0][R] = *reinterpret_cast<const float4*>(wrow + 32 * R * ld);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s + 1 < 8) {
#pragma unroll
      for (int c = 0; c < CPI; ++c) dw_load(planes, Rd, c0, i, h, CPI * (s + 1) + c, ck[(s + 1) & 1][c]);
#pragma unroll
      for (int R = 0; R < 2; ++R) w[(s + 1) & 1][R] = *reinterpret_cast<const float4*>(wrow + 32 * R * ld + 8 * (s + 1));
    }
    const f32x16& bt = b[s >> 2];
    const int q0 = 4 * (s & 3);
#pragma unroll
    for (int R = 0; R < 2; ++R) o[R] = SDEH_MFMA(w[s & 1][R].x, bt[q0], o[R]);
#pragma unroll
    for (int R = 0; R < 2; ++R) o[R] = SDEH_MFMA(w[s & 1][R].y, bt[q0 + 1], o[R]);
#pragma unroll
    for (int R = 0; R < 2; ++R) o[R] = SDEH_MFMA(w[s & 1][R].z, bt[q0 + 2], o[R]);
#pragma unroll
    for (int R = 0; R < 2; ++R) o[R] = SDEH_MFMA(w[s & 1][R].w, bt[q0 + 3], o[R]);
#pragma unroll
    for (int c = 0; c < CPI; ++c) {
      dsum[(CPI * s + c) >> 2] += sum4(ck[s & 1][c].dv);
      dw_chunk(ck[s & 1][c], acc);
    }
    SDEH_FENCE();
  }
  ws_barrier();
}

// Sums over the 32 trajectories (lanes of a half) of the 32 accumulator-layout registers v[R][q], reduce-scatter form: lane (jl, h)
// returns the sum of register q = jl & 15 of tile R = jl >> 4 (its half's row 32 R + 4 h + rrow(q)).  77 vector instructions instead
// of 7 per register: the first exchange (lane ^ 16) pairs the two tiles, the others (lane ^ 15, ^ 7, ^ 3, ^ 1: DPP row_mirror,
// row_half_mirror, quad_perm) halve the register set, every lane sending the half it does not keep.
template <int CTRL>
__device__ __forceinline__ void rs_step(const float* in, float* out, int n, bool up) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (q < n) {
      const float keep = up ? in[q + n] : in[q];
      const float send = up ? in[q] : in[q + n];
      out[q] = keep + dpp_mov<CTRL>(send);
    }
  }
}
__device__ __forceinline__ float reduce_scatter32(const f32x16 (&v)[2], int lane) {
  float w[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {  // even rows of 16 lanes keep tile 0, odd rows tile 1
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0][q]), __float_as_uint(v[1][q]), false, false);
    w[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float a[8], b[4], c[2], e[1];
  rs_step<0x140>(w, a, 8, (lane & 8) != 0);   // row_mirror: lane ^ 15
  rs_step<0x141>(a, b, 4, (lane & 4) != 0);   // row_half_mirror: lane ^ 7
  rs_step<0x1B>(b, c, 2, (lane & 2) != 0);    // quad_perm [3,2,1,0]: lane ^ 3
  rs_step<0xB1>(c, e, 1, (lane & 1) != 0);    // quad_perm [1,0,3,2]: lane ^ 1
  return e[0];
}

// d <= 4 (VIO): the gradients of input_embed.weight [64, d] and out_layer.weight [d, 64] as 4 x 4 BLOCK products (v_mfma_f32_4x4x1_16b_f32:
// 16 independent blocks per instruction, 8 cycles -- the same 32 multiply-adds per cycle as the 32 x 32 tiles, but on tiles without
// padding: a [64, 4] gradient is 16 blocks x 4 x 4, one instruction per trajectory, 32 instructions = 256 cycles per step and wave
// instead of 32 instructions of 64 cycles on a tile that is 7/8 zeros).  A wave contracts over ITS OWN 32 trajectories from its own two
// planes -- no barrier, no partner -- and the team's four partial sums meet once, at the end of the launch.
//   lane l = 4 b + i:  A operand = row value A_b[i], B operand = column value B_b[i];  result register r of lane (b, j) = D_b[r][j]
//   rows from plane R (row index: `own` ? l : l & 3), columns from plane Cp (row index: `own` ? l & 3 : l); sum4 of the A rows -> rsum
typedef float f32x4b __attribute__((ext_vector_type(4)));
template <bool A_OWN>
__device__ __forceinline__ void block_product(const float* __restrict__ Rp, const float* __restrict__ Cp, int lane, f32x4b (&acc)[2], float& rsum) {
  const float* __restrict__ ra = Rp + (A_OWN ? lane : (lane & 3)) * RS;
  const float* __restrict__ rb = Cp + (A_OWN ? (lane & 3) : lane) * RS;
  float4 a[2], b[2];
  a[0] = *reinterpret_cast<const float4*>(ra);
  b[0] = *reinterpret_cast<const float4*>(rb);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c + 1 < 8) {
      a[(c + 1) & 1] = *reinterpret_cast<const float4*>(ra + 4 * (c + 1));
      b[(c + 1) & 1] = *reinterpret_cast<const float4*>(rb + 4 * (c + 1));
    }
    const float4 av = a[c & 1], bv = b[c & 1];
    rsum += (av.x + av.y) + (av.z + av.w);
    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, bv.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, bv.y, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, bv.z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, bv.w, acc[1], 0, 0, 0);
  }
}

// the transposed layer alone (no weight-gradient product next to it):  o[R] = sum_s sum_e W[(8 s + 4 h + e) * LD + 32 R + i] * b[s >> 2][4 (s & 3) + e]
template <int LD>
__device__ __forceinline__ void chain_cols(const float* __restrict__ wcol, const f32x16 (&b)[2], f32x16 (&o)[2]) {
#pragma unroll
  for (int R = 0; R < 2; ++R)
#pragma unroll
    for (int q = 0; q < 16; ++q) o[R][q] = 0.0f;
  float wc[2][4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int R = 0; R < 2; ++R) wc[0][e][R] = wcol[e * LD + 32 * R];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s + 1 < 8) {
      const float* __restrict__ p = wcol + 8 * (s + 1) * LD;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int R = 0; R < 2; ++R) wc[(s + 1) & 1][e][R] = p[e * LD + 32 * R];
    }
    const f32x16& bt = b[s >> 2];
    const int q0 = 4 * (s & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int R = 0; R < 2; ++R) o[R] = SDEH_MFMA(wc[s & 1][e][R], bt[q0 + e], o[R]);
    SDEH_FENCE();
  }
}

}  // namespace bwdf2

}  // namespace sdeh
