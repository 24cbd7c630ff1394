// Fused training backward for SMALL batches, back-propagation through time (method kl / kl_ito): the kernel of sdeh_bwdf.hip on tiles
// of 16 trajectories (v_mfma_f32_16x16x4_f32) instead of 32 (v_mfma_f32_32x32x2_f32).
//
// Why: back-propagation through time is sequential in the T steps; its only parallelism is the batch.  The reference's training
// batches are 512 and 2048 (conf/solver/*.yaml): 16 / 64 tiles of 32 trajectories -- 32 / 128 wavefronts on a chip that holds 1024,
// each running the whole per-step chain of sdeh_bwdf.hip (37 k cycles: 296 matrix instructions of 64 cycles, activation and
// derivative of 48 elements per lane, eight LDS round trips).  Both matrix shapes retire 32 multiply-adds per cycle, so halving the
// tile halves every lane's chain -- 8 accumulator registers per row tile instead of 16 -- and doubles the number of teams.  A team
// is its own workgroup here (a barrier couples only its waves), two of which fit a CU's LDS at d <= 32.  From 16 384 trajectories on
// every SIMD has a wave of the 32-trajectory kernel and that one is used (fewer, longer instructions per trajectory).
// A team is FOUR wavefronts (NW = 4): wave (r, mb) owns the 16 rows 32 r + 16 mb .. of every layer -- one accumulator tile, the chain
// halves once more, and at 252-288 registers two such waves share a SIMD when the batch asks for it.  (NW = 2, two tiles per wave,
// is kept for the comparison.)
//
// Same semantics and the same reference (losses/oc.py:176-222 / 301-334 / 416-446 through models/mlp.py:114-122 and
// models/reparam.py:56-83,131-197) as sdeh_bwdf.hip; same inputs (the coordinate-major planes of sdeh_simulate_fwd_train2), the same
// partial-gradient record per team and the same deterministic sums behind it.  One difference: the re-evaluated pre-activations are
// not bit for bit those of the forward launch (another instruction, another summation tree), so activations with a kink (ReLU) stay
// with the 32-trajectory kernel, whose re-evaluation is bitwise (tests/test_hip_bwd_fused.py::test_relu_units_near_the_kink...).
//
// Layouts (lane = (n, g), n = lane & 15, g = lane >> 4):
//   accumulator tile (16 x 16):  register q <-> row 4 g + q, column n        (two tiles m = 0, 1 make a wave's 32 rows)
//   A operand (16 x 4):          lane holds A[n][g];  B operand (4 x 16): lane holds B[g][n]
//   k order of a 64-deep product: instruction (u, e) covers k = 16 u + 4 g + e -- the A rows are read as one ds_read_b128 per u
//   exchange planes [64 rows][16 trajectories], row stride 20 floats: accumulator-layout stores, B-operand loads and the transposed
//   ds_read_b128 loads of the weight-gradient products are all conflict-free.
#include "sdeh_bwd.hpp"

namespace sdeh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SDEH_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace bwdf16 {

constexpr int RSW = 68;        // row stride of the [., 64] weight copies (as in sdeh_bwdf.hip)
constexpr int RS = 20;         // row stride of the exchange planes [row][16 trajectories]
constexpr int PLANE = 64 * RS;
constexpr int TABS = 6 * 64;   // (mu, 1/sigma^2) x {prior, second, target}
constexpr int TILE = 16;
template <int OTD> constexpr int rsi() { return OTD == 1 ? 36 : 68; }
template <int OTD, int LH> constexpr int lds_floats() { return 64 * rsi<OTD>() + LH * 64 * RSW + 32 * OTD * RSW + LH * 64 + 64 + TABS + 4 * PLANE + 8; }
// partial-gradient record of one team: the layout of sdeh_bwdf.hip (natural matrices)
template <int OTD> constexpr int off_whid() { return 64 * 32 * OTD; }
template <int OTD, int LH> constexpr int off_wout() { return off_whid<OTD>() + LH * 4096; }
template <int OTD, int LH> constexpr int off_bhid() { return off_wout<OTD, LH>() + 32 * OTD * 64; }
template <int OTD, int LH> constexpr int off_bout() { return off_bhid<OTD, LH>() + LH * 64; }

// a wave's MT row tiles (MT = 2: 32 rows, teams of two waves; MT = 1: 16 rows, teams of four) of 16 trajectories:
// element (m, q) <-> row 16 m + 4 g + q of the wave's rows (or coordinates)
template <int MT>
struct Vt {
  f32x4 m[MT];
};
template <int MT>
__device__ __forceinline__ Vt<MT> zero_t() {
  Vt<MT> v;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) v.m[m][q] = 0.0f;
  return v;
}

// accumulator layout -> plane [row][trajectory]; row0 = first row of the wave's tiles
template <int MT>
__device__ __forceinline__ void plane_put(float* __restrict__ plane, int row0, int n, int g, const Vt<MT>& v) {
  float* __restrict__ p = plane + (row0 + 4 * g) * RS + n;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) p[(16 * m + q) * RS] = v.m[m][q];
}
// transposed read: lane (n, g) gets row `row0 + n`, trajectories 4 g .. 4 g + 3
__device__ __forceinline__ float4 plane_getT(const float* __restrict__ plane, int row0, int n, int g) {
  return *reinterpret_cast<const float4*>(plane + (row0 + n) * RS + 4 * g);
}
// values of a per-row table in accumulator order: p = &table[first row of the wave's tiles + 4 g]
template <int MT>
__device__ __forceinline__ Vt<MT> rows_t(const float* __restrict__ p) {
  Vt<MT> v;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float4 t = *reinterpret_cast<const float4*>(p + 16 * m);
    v.m[m][0] = t.x; v.m[m][1] = t.y; v.m[m][2] = t.z; v.m[m][3] = t.w;
  }
  return v;
}

// out = init + W[rows of this wave][:] . B     (wrow = &W[(first row + n) * ld + 4 g], ld = row stride; bcol = &plane[4 g * RS + n];
// k-group u covers columns 16 u + 4 g .. + 3 of W and the same rows of the B plane).  With two row tiles: independent chains.
template <int NU, int MT>
__device__ __forceinline__ Vt<MT> mm_rows(const float* __restrict__ wrow, int ld, const float* __restrict__ bcol, int nu, const Vt<MT>& init) {
  Vt<MT> acc = init;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (u < nu) {
      float4 w[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) w[m] = *reinterpret_cast<const float4*>(wrow + 16 * m * ld + 16 * u);
      const float* __restrict__ bp = bcol + 16 * u * RS;
      const float b0 = bp[0], b1 = bp[RS], b2 = bp[2 * RS], b3 = bp[3 * RS];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m].x, b0, acc.m[m]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m].y, b1, acc.m[m]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m].z, b2, acc.m[m]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m].w, b3, acc.m[m]);
    }
  }
  return acc;
}

// out = W[:, columns of this wave]^T . B     (wcol = &W[(4 g) * LD + first column + n]; k-group u covers rows 16 u + 4 g .. + 3 of both)
template <int NU, int LD, int MT>
__device__ __forceinline__ Vt<MT> mm_cols(const float* __restrict__ wcol, const float* __restrict__ bcol, int nu) {
  Vt<MT> acc = zero_t<MT>();
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (u < nu) {
      const float* __restrict__ p = wcol + 16 * u * LD;
      const float* __restrict__ bp = bcol + 16 * u * RS;
      const float b0 = bp[0], b1 = bp[RS], b2 = bp[2 * RS], b3 = bp[3 * RS];
      float w[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[m][e] = p[e * LD + 16 * m];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m][0], b0, acc.m[m]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m][1], b1, acc.m[m]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m][2], b2, acc.m[m]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc.m[m] = SDEH_MFMA16(w[m][3], b3, acc.m[m]);
    }
  }
  return acc;
}

// acc[m][c] += delta[rows drow0 + 16 m ..][.] . a[rows arow0 + 16 c ..][.]^T over the 16 trajectories of the planes (NC column tiles);
// bsum[m] += this lane's four delta values of row drow0 + 16 m + n (summed over g afterwards: the bias gradient of that row)
template <int NC, int MT>
__device__ __forceinline__ void dw_acc(const float* __restrict__ dplane, int drow0, const float* __restrict__ aplane, int arow0,
                                       f32x4 (&acc)[MT][NC], float (&bsum)[MT], int n, int g) {
  float4 dv[MT], av[NC];
#pragma unroll
  for (int m = 0; m < MT; ++m) dv[m] = plane_getT(dplane, drow0 + 16 * m, n, g);
#pragma unroll
  for (int c = 0; c < NC; ++c) av[c] = plane_getT(aplane, arow0 + 16 * c, n, g);
#pragma unroll
  for (int m = 0; m < MT; ++m) bsum[m] += (dv[m].x + dv[m].y) + (dv[m].z + dv[m].w);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][c] = SDEH_MFMA16(dv[m].x, av[c].x, acc[m][c]);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][c] = SDEH_MFMA16(dv[m].y, av[c].y, acc[m][c]);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][c] = SDEH_MFMA16(dv[m].z, av[c].z, acc[m][c]);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][c] = SDEH_MFMA16(dv[m].w, av[c].w, acc[m][c]);
  }
}

// act(z) and act'(z) of a wave's tiles
template <int ACT, int MT>
__device__ __forceinline__ void act_both(const Vt<MT>& z, Vt<MT>& a, Vt<MT>& gr) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if constexpr (ACT == SDEH_ACT_GELU_ERF) {
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        f2 av, gv;
        act_gelu2_both(f2{z.m[m][q], z.m[m][q + 1]}, av, gv);
        a.m[m][q] = av.x; a.m[m][q + 1] = av.y;
        gr.m[m][q] = gv.x; gr.m[m][q + 1] = gv.y;
      }
    } else if constexpr (ACT == SDEH_ACT_SILU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { float av, gv; act_silu_both(z.m[m][q], av, gv); a.m[m][q] = av; gr.m[m][q] = gv; }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { gr.m[m][q] = act_grad(z.m[m][q], ACT); a.m[m][q] = act_ct<ACT>(z.m[m][q]); }
    }
  }
}

// accumulator tile -> block (rows row0 + 4 g + q, columns col0 + n) of a natural [rows][ld] matrix
__device__ __forceinline__ void store_tile(float* __restrict__ mat, int ld, int row0, int col0, int n, int g, const f32x4& v) {
  float* __restrict__ p = mat + (row0 + 4 * g) * ld + col0 + n;
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q * ld] = v[q];
}

template <int MT>
__device__ __forceinline__ Vt<MT> mul_t(const Vt<MT>& a, const Vt<MT>& b) {
  Vt<MT> v;
#pragma unroll
  for (int m = 0; m < MT; ++m) v.m[m] = a.m[m] * b.m[m];
  return v;
}

}  // namespace bwdf16

// KLB (a Bridge's generative network with method kl): the running cost on the plane cost_in = u + v, lam_in added to the adjoint
// ZIN (round 5): no re-evaluation -- pre-activations and raw network output come from the record of the training forward
// (sdeh_traj_ws.hpp: ZRec; a 16-trajectory tile is one half of a record tile: ONE 16-byte load per lane, layer and row tile, all of a
// step's requested a step ahead).  The forward pass with its layer exchanges is gone: four of a step's nine barriers, the chain of
// dependent matrix instructions behind each of them, and the kink rule (the record IS the forward launch's pre-activation).
template <int OTD, int LH, int NW, bool KLB = false, bool ZIN = false>
__global__ __launch_bounds__(64 * NW) void bwdf16_kernel(const BwdfArgs A) {
  using namespace bwdf16;
  constexpr int RSI = rsi<OTD>(), DPP = 32 * OTD;
  constexpr int MT = NW == 4 ? 1 : 2;        // accumulator tiles (of 16 rows) per wave
  constexpr int NT = 64 * NW;                // threads
  using V = Vt<MT>;
  constexpr int NCI = 2 * OTD;               // coordinate tiles (of 16) of input_embed's columns
  constexpr int NCO = OTD == 2 ? 4 : 2;      // channel tiles of this wave's part of out_layer's gradient
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ Win = lds;
  float* __restrict__ Whid = Win + 64 * RSI;
  float* __restrict__ Wout = Whid + LH * 64 * RSW;
  float* __restrict__ bh = Wout + DPP * RSW;
  float* __restrict__ bo = bh + LH * 64;
  float* __restrict__ tabs = bo + 64;
  float* __restrict__ pl = tabs + TABS;  // planes: A[0], A[1], D[0], D[1]
  float* __restrict__ red = pl + 4 * PLANE;  // NW floats: the waves' parts of d loss / d gamma(t)
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = NW == 4 ? wave >> 1 : wave;     // row tile (of 32) the wave works in
  const int mb = NW == 4 ? wave & 1 : 0;        // ... and, in teams of four, which half of it: rows 32 r + 16 mb ..
  const int mo = 16 * mb;
  const int n = lane & 15, g = lane >> 4;
  const int ct = OTD == 2 ? r : 0;  // coordinate tile (of 32) this wave owns (d <= 32: both waves mirror tile 0)
  const int d = A.d, T = A.n_steps;
  const long long B = A.batch;

  // ---- stage the parameters (natural layouts, zero padding) and the Gaussian tables ----------------------------------------
  for (int idx = tid; idx < 64 * RSI; idx += NT) {
    const int row = idx / RSI, col = idx - row * RSI;
    Win[idx] = col < d ? A.w_in[row * d + col] : 0.0f;
  }
  for (int idx = tid; idx < LH * 64 * RSW; idx += NT) {
    const int l = idx / (64 * RSW), rem = idx - l * 64 * RSW, row = rem / RSW, col = rem - row * RSW;
    Whid[idx] = col < 64 ? A.w_hid[l][row * 64 + col] : 0.0f;
  }
  for (int idx = tid; idx < DPP * RSW; idx += NT) {
    const int row = idx / RSW, col = idx - row * RSW;
    Wout[idx] = (row < d && col < 64) ? A.w_out[row * 64 + col] : 0.0f;
  }
  for (int idx = tid; idx < LH * 64; idx += NT) bh[idx] = A.b_hid[idx >> 6][idx & 63];
  if (tid < 64) bo[tid] = tid < d ? A.b_out[tid] : 0.0f;
  for (int idx = tid; idx < TABS; idx += NT) {  // tabs[(2 k + c) * 64 + coordinate]: c = 0 mean, 1 inverse variance; k = prior, second, target
    const int k = idx / 128, c = (idx >> 6) & 1, cj = idx & 63;
    const int which = k == 0 ? 1 : (k == 1 ? 2 : 0);
    tabs[idx] = cj < d && cj < L.dp ? ws[L.dg[which] + 2 * cj + c] : 0.0f;
  }
  // planes start zeroed: rows / trajectories that are never written must not hold NaNs (they meet zero weights)
  for (int idx = tid; idx < 4 * PLANE; idx += NT) pl[idx] = 0.0f;
  __syncthreads();

  const int act = A.act, ctrl_kind = A.ctrl_kind, flags = A.flags;
  const bool has_score = ctrl_kind != SDEH_CTRL_CLIPPED;
  const bool refc = (flags & SDEH_FLAG_REFERENCE_CTRL) && A.loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool expo = A.loss_kind == SDEH_LOSS_EXPONENTIAL;
  const bool ito = (flags & SDEH_FLAG_ITO) != 0;
  const int cb = 32 * ct + mo + 4 * g;  // first coordinate of this lane's registers: coordinate(m, q) = cb + 16 m + q
  const int n_ku = (d + 15) >> 4;  // k-groups of 16 covering the coordinates
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
  // position of channel 32 r + 16 m + 4 g + q in the workspace's time-embedding rows (stored in the 32 x 32 accumulator order of the
  // forward kernels: [(2 r + h) * 16 + (c & 3) + 4 (c >> 3)], h = (c >> 2) & 1 for c = channel & 31): four consecutive floats per m
  const int emb_pos = (2 * r + (g & 1)) * 16 + 4 * (g >> 1) + 8 * mb;

  // Coordinates >= d of a tile need no masks: their x / sc / xi are loaded or drawn as zeros, the weight copies and Gaussian tables
  // are zero-padded, so every quantity derived from them stays exactly zero.

  // weight-gradient accumulators (row tile m of this wave x column tiles of 16)
  f32x4 dw_in[MT][NCI], dw_hid[LH][MT][4], dw_out[MT][NCO];
  float bs_hid[LH][MT], bs_out[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    bs_out[m] = 0.0f;
#pragma unroll
    for (int c = 0; c < NCI; ++c) dw_in[m][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NCO; ++c) dw_out[m][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int l = 0; l < LH; ++l) {
      bs_hid[l][m] = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) dw_hid[l][m][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  }

  const int n_tiles = A.n_tiles, n_teams = (int)gridDim.x, team_g = (int)blockIdx.x;
  const int n_rounds = (n_tiles + n_teams - 1) / n_teams;
  int par = 0;

  auto load_x = [&](int t, int tile_i) {
    V v;
    const long long rw = (long long)tile_i * TILE + n;
    const float* __restrict__ plane = A.xs + (long long)t * d * B + (rw < B ? rw : B - 1);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) v.m[m][q] = cb + 16 * m + q < d ? plane[(long long)(cb + 16 * m + q) * B] : 0.0f;
    return v;
  };
  auto load_emb = [&](int t) {
    V v;
    const float* __restrict__ p = ws + L.emb + t * 64 + emb_pos;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 e = *reinterpret_cast<const float4*>(p + 8 * m);
      v.m[m][0] = e.x; v.m[m][1] = e.y; v.m[m][2] = e.z; v.m[m][3] = e.w;
    }
    return v;
  };
  // a team without a tile shadows the last one (and contributes zeros)
  auto clamp_tile = [&](int tile_i) { return tile_i < n_tiles ? tile_i : n_tiles - 1; };

  // ZIN: this lane's share of the record of (step, 16-trajectory tile): layers 0 .. LH, then the raw network output of its coordinates
  typedef float f32x4z __attribute__((ext_vector_type(4)));
  struct ZStep { V z[LH + 1]; V nn; };
  const long long zr_tiles = (B + 31) >> 5;
  auto load_zstep = [&](int t, int tile_i, ZStep& zs) {
    if constexpr (ZIN) {
      const float* __restrict__ base = A.zrec + ((long long)t * zr_tiles + (tile_i >> 1)) * ((LH + 1) * 2048 + OTD * 1024);
      unsigned lo = (unsigned)(16 * (tile_i & 1) + n) * 16u;
      asm volatile("" : "+v"(lo));
      const char* __restrict__ pb = reinterpret_cast<const char*>(base) + lo;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int cq = 8 * r + 4 * mb + 4 * m + g;  // channel quad of rows 32 r + 16 mb + 16 m + 4 g ..
#pragma unroll
        for (int k = 0; k <= LH; ++k) zs.z[k].m[m] = __builtin_nontemporal_load(reinterpret_cast<const f32x4z*>(pb + (k * 2048 + cq * 128) * 4));
        const int nq = 4 * mb + 4 * m + g;  // coordinate quad (of this wave's coordinate tile ct) of coordinates cb + 16 m ..
        const bool okq = cb + 16 * m < d;   // (quads beyond d may never have been written: a valid address, zeros by the select)
        const f32x4z v = __builtin_nontemporal_load(reinterpret_cast<const f32x4z*>(pb + ((LH + 1) * 2048 + ct * 1024 + (okq ? nq : 0) * 128) * 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) zs.nn.m[m][q] = cb + 16 * m + q < d ? v[q] : 0.0f;
      }
    }
  };
  V xnext = load_x(T - 1, clamp_tile(team_g)), embnext;
  ZStep znext;
  if constexpr (!ZIN) embnext = load_emb(T - 1);
  load_zstep(T - 1, clamp_tile(team_g), znext);
  for (int round = 0; round < n_rounds; ++round) {
    const int it_tile = team_g + round * n_teams;
    const bool live_item = it_tile < n_tiles;
    const int cur_tile = clamp_tile(it_tile);
    const long long tile = cur_tile;
    const long long row = tile * TILE + n;
    const bool live = live_item && row < B;
    const long long lrow = row < B ? row : B - 1;
    const float wi = live ? A.grad_rnd[lrow] : 0.0f;
    const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);

    auto load8c = [&](const float* __restrict__ plane) {  // 8 coordinates of the own tile from a coordinate-major [d][B] plane
      V v;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) v.m[m][q] = cb + 16 * m + q < d ? plane[(long long)(cb + 16 * m + q) * B + lrow] : 0.0f;
      return v;
    };

    V lam = zero_t<MT>();
    // lambda_T = w_i d(terminal costs)/dx_T  (losses/oc.py:225,337,449-450)
    if (flags & SDEH_FLAG_TERMINAL_SECOND) {
      const V xT = load8c(A.xs + (long long)T * d * B);
      const V smu = rows_t<MT>(tabs + 2 * 64 + cb), sis = rows_t<MT>(tabs + 3 * 64 + cb);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) lam.m[m][q] = wi * (smu.m[m][q] - xT.m[m][q]) * sis.m[m][q];
    }
    if ((flags & SDEH_FLAG_TERMINAL_TARGET) && A.tscore != nullptr) {
      const V st = load8c(A.tscore);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) lam.m[m][q] = fmaf(-wi, st.m[m][q], lam.m[m][q]);
    }

    for (int t = T - 1; t >= 0; --t) {
      const V x = xnext;
      V embv;  // timestep_embed(t) + input bias of this wave's channels
      if constexpr (!ZIN) embv = embnext;
      const ZStep zs = znext;
      if (t > 0) {
        xnext = load_x(t - 1, cur_tile);
        if constexpr (!ZIN) embnext = load_emb(t - 1);
        load_zstep(t - 1, cur_tile, znext);
      } else if (round + 1 < n_rounds) {
        xnext = load_x(T - 1, clamp_tile(it_tile + n_teams));
        if constexpr (!ZIN) embnext = load_emb(T - 1);
        load_zstep(T - 1, clamp_tile(it_tile + n_teams), znext);
      }
      // the step's other inputs: requested first, consumed after the forward pass
      V scv = zero_t<MT>(), xi = zero_t<MT>();
      if (has_score) scv = load8c(A.sc + (long long)t * d * B);
      if (ito) {
        if (A.noise != nullptr) {
          const float* __restrict__ rowp = A.noise + ((long long)t * B + lrow) * d;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) xi.m[m][q] = cb + 16 * m + q < d ? rowp[cb + 16 * m + q] : 0.0f;
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float n4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cb + 16 * m < d) box_muller4(philox_block(A.seed, rng_off, grow, t, (cb + 16 * m) >> 2), n4);
#pragma unroll
            for (int q = 0; q < 4; ++q) xi.m[m][q] = cb + 16 * m + q < d ? n4[q] : 0.0f;
            SDEH_FENCE();
          }
        }
      }
      const float gam0 = has_score ? ws[L.gam + t * L.g] : 0.0f;  // gamma(t) (its first entry), requested early as well
      cfp cf = as_const(ws + L.coef + t * kCoefStride);
      const float sig = cf[CF_SIGMA], wl = cf[CF_W];
      const float c_i = expo ? cf[CF_SBK] : cf[CF_SQDT];
      const float cdt = expo ? cf[CF_B2S2] : cf[CF_DT];
      const float c_u = expo ? cf[CF_B2S2] : sig * cf[CF_DT];
      const float c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], cf[CF_DT], 1.0f);

      // ======================================================================================= forward (re-evaluation at x_t)
      // planes: a_0 = x in A[par], a_k = act(Z_{k-1}) in A[(par + k) & 1] (k = 1 .. LH + 1); the last two stay intact for the backward
      // pass, the earlier ones (and x) are re-published from registers when their weight gradient is due
      float* __restrict__ Ap[2] = {pl + par * PLANE, pl + (1 - par) * PLANE};
      float* __restrict__ Dp[2] = {pl + 2 * PLANE, pl + 3 * PLANE};
      V gr[LH + 1], aown[LH > 1 ? LH - 1 : 1];
      const int bofs = 4 * g * RS + n;  // this lane's column of a plane as an MFMA B operand
      V nn;
      if constexpr (ZIN) {
        // act / act' of every layer from the record (12 values per lane); the planes are laid out as the forward pass leaves them: a_{LH+1}
        // in A[(LH + 1) & 1], a_LH in A[LH & 1], earlier activations kept in registers until their weight gradient is due
        ws_barrier();  // every read of the previous step's planes is done
#pragma unroll
        for (int k = 0; k <= LH; ++k) {
          V ak;
          SDEH_ACT_SWITCH(act, ACT, act_both<ACT, MT>(zs.z[k], ak, gr[k]););
          if (k >= LH - 1) plane_put<MT>(Ap[(k + 1) & 1], 32 * r + mo, n, g, ak);
          else aown[k < LH - 1 ? k : 0] = ak;
        }
        nn = zs.nn;
      } else {
      plane_put<MT>(Ap[0], 32 * ct + mo, n, g, x);
      ws_barrier();
      {
        const V z0 = mm_rows<2 * OTD, MT>(Win + (32 * r + mo + n) * RSI + 4 * g, RSI, Ap[0] + bofs, n_ku, embv);
        V a1;
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT, MT>(z0, a1, gr[0]););
        if constexpr (LH > 1) aown[0] = a1;
        plane_put<MT>(Ap[1], 32 * r + mo, n, g, a1);
      }
      ws_barrier();
#pragma unroll
      for (int l = 0; l < LH; ++l) {  // hidden layer l: Z_{l+1} = W_l a_{l+1} + b_l;  a_{l+2} = act(Z_{l+1})
        const V z = mm_rows<4, MT>(Whid + l * 64 * RSW + (32 * r + mo + n) * RSW + 4 * g, RSW, Ap[(l + 1) & 1] + bofs, 4,
                                rows_t<MT>(bh + l * 64 + 32 * r + mo + 4 * g));
        V an;
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT, MT>(z, an, gr[l + 1]););
        if (l + 2 <= LH - 1) aown[l + 2 <= LH - 1 ? l + 1 : 0] = an;  // a_{l+2} is re-published later iff l + 2 <= LH - 1
        plane_put<MT>(Ap[l & 1], 32 * r + mo, n, g, an);
        ws_barrier();
      }
      nn = mm_rows<4, MT>(Wout + (32 * ct + mo + n) * RSW + 4 * g, RSW, Ap[(LH + 1) & 1] + bofs, 4, rows_t<MT>(bo + cb));
      }

      // ======================================================================================= upstream gradient of the control
      V G, Gc, cvec, dout;
      {
        float gsum = 0.0f;
        V gcoord;
        const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
        V rr = zero_t<MT>();  // reference control sigma * prior.score(x) (solver/oc.py:305-306)
        if (refc) {
          const V pmu = rows_t<MT>(tabs + 0 * 64 + cb), pis = rows_t<MT>(tabs + 1 * 64 + cb);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) rr.m[m][q] = sig * (pmu.m[m][q] - x.m[m][q]) * pis.m[m][q];
        }
        // Bridge, method kl: the control entering the running cost is u + v (a plane, loaded where it is used)
        V cin = zero_t<MT>();
        if constexpr (KLB) cin = load8c(A.cost_in + (long long)t * d * B);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float mfac = 0.0f, csc = 0.0f, keep_s = 0.0f;
            const float s = scv.m[m][q], nv = nn.m[m][q];
            if (has_score) {
              const float gam = A.g == 1 ? gam0 : ws[L.gam + t * L.g + min(cb + 16 * m + q, L.g - 1)];
              mfac = mult * gam;
              csc = clipf(s, A.clip_score);
              keep_s = fabsf(s) <= A.clip_score ? 1.0f : 0.0f;
            }
            const float u = clipf(nv, A.clip_model) + mfac * csc;
            const float gc = wi * fmaf(KLB ? cin.m[m][q] : u - rr.m[m][q], cdt, ito ? c_i * xi.m[m][q] : 0.0f);
            const float gq = fmaf(c_u, lam.m[m][q], gc);
            Gc.m[m][q] = gc;
            G.m[m][q] = gq;
            const float gg = gq * mult * csc;
            gcoord.m[m][q] = gg;
            gsum += gg;
            cvec.m[m][q] = keep_s * mfac * gq;
            dout.m[m][q] = fabsf(nv) <= A.clip_model ? gq : 0.0f;
          }
        if (has_score && live_item) {  // d loss / d gamma(t): summed over the team's trajectories
          if (A.g == 1) {
            gsum = sum_wave(gsum);
            if constexpr (NW == 4) {  // the two waves of a row tile meet in LDS; written behind the next barrier
              if (lane == 0) red[wave] = (OTD == 2 || r == 0) ? gsum : 0.0f;
            } else {
              if (lane == 0) A.gpart[(tile * T + t) * A.gw + r] = (OTD == 2 || r == 0) ? gsum : 0.0f;
            }
          } else if (OTD == 2 || r == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float v = sum_row16(gcoord.m[m][q]);
                if (n == 0) A.gpart[(tile * T + t) * A.gw + cb + 16 * m + q] = v;
              }
          }
        }
      }

      // ======================================================================================= backward + weight gradients
      // delta planes alternate between D[0] and D[1]; each stage: publish delta_k (and a_k unless its plane is still intact), barrier,
      // dW_k += delta_k a_k^T, then the adjoint of the layer below
      plane_put<MT>(Dp[0], 32 * ct + mo, n, g, dout);
      ws_barrier();  // delta_out | a_{LH+1}
      if constexpr (NW == 4) {
        if (has_score && live_item && A.g == 1 && mb == 0 && lane == 0) A.gpart[(tile * T + t) * A.gw + r] = red[wave] + red[wave + 1];
      }
      // out_layer: rows = coordinates.  Two coordinate tiles: this wave's 32 coordinates x all 64 channels; one tile: the 32
      // coordinates x this wave's 32 channels
      dw_acc<NCO, MT>(Dp[0], 32 * ct + mo, Ap[(LH + 1) & 1], OTD == 2 ? 0 : 32 * r, dw_out, bs_out, n, g);
      V dl = mul_t<MT>(mm_cols<2 * OTD, RSW, MT>(Wout + (4 * g) * RSW + 32 * r + mo + n, Dp[0] + bofs, n_ku), gr[LH]);
#pragma unroll
      for (int l = LH - 1; l >= 0; --l) {  // hidden layer l: dl = d loss / d Z_{l+1}
        float* __restrict__ Dl = Dp[(LH - l) & 1];
        float* __restrict__ Al = Ap[(l + 1) & 1];
        plane_put<MT>(Dl, 32 * r + mo, n, g, dl);
        if (l + 1 <= LH - 1) plane_put<MT>(Al, 32 * r + mo, n, g, aown[l + 1 <= LH - 1 ? l : 0]);  // a_{l+1}: its plane was overwritten by a_{l+3}
        ws_barrier();
        dw_acc<4, MT>(Dl, 32 * r + mo, Al, 0, dw_hid[l], bs_hid[l], n, g);
        dl = mul_t<MT>(mm_cols<4, RSW, MT>(Whid + l * 64 * RSW + (4 * g) * RSW + 32 * r + mo + n, Dl + bofs, 4), gr[l]);
      }
      float* __restrict__ Din = Dp[(LH + 1) & 1];
      plane_put<MT>(Din, 32 * r + mo, n, g, dl);
      plane_put<MT>(Ap[0], 32 * ct + mo, n, g, x);
      ws_barrier();  // delta_0 | x
      {
        float esum[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) esum[m] = 0.0f;
        dw_acc<NCI, MT>(Din, 32 * r + mo, Ap[0], 0, dw_in, esum, n, g);
#pragma unroll
        for (int m = 0; m < MT; ++m) {  // d loss / d (time embedding + input bias)[t][32 r + 16 m + n]
          const float e = sum_xor32(sum_xor16(esum[m]));
          if (live_item && g == 0) A.epart[(tile * T + t) * 64 + 32 * r + mo + 16 * m + n] = e;
        }
      }
      // ===================================================================================== adjoint update
      //   lambda_t = c_x lambda_{t+1} + W_in^T delta_0 + (d score term / d x)^T G + direct cost terms
      const V dx = mm_cols<4, RSI, MT>(Win + (4 * g) * RSI + 32 * ct + mo + n, Din + bofs, 4);
      // score terms the reference detaches (reparam.py:58,134,169,188) or obtains by autograd without a graph carry no Jacobian
      const float coef_t = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET ? wl : 0.0f);
      const float coef_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR ? 1.0f - wl : 0.0f;
      const float jac_t = (!has_score || (flags & (SDEH_FLAG_DETACH_SCORE | SDEH_FLAG_TARGET_SCORE_CONST))) ? 0.0f : coef_t;
      const float jac_p = (!has_score || (flags & SDEH_FLAG_DETACH_SCORE)) ? 0.0f : coef_p;
      V vt = zero_t<MT>();
      if (jac_t != 0.0f) {  // closed-form target scores are differentiated through x
        if (A.target.kind == SDEH_DENS_DIAG_GAUSS) {
          const V tis = rows_t<MT>(tabs + 5 * 64 + cb);
#pragma unroll
          for (int m = 0; m < MT; ++m) vt.m[m] = -tis.m[m] * cvec.m[m];
        } else if (A.target.kind == SDEH_DENS_MULTI_WELL) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float y = x.m[m][q] - A.target.p1;
              vt.m[m][q] = (cb + 16 * m + q < A.target.n_comp ? -4.0f * (3.0f * y * y - A.target.p0) : -1.0f) * cvec.m[m][q];
            }
        } else if (A.target.kind == SDEH_DENS_FUNNEL) {
          // s_0 = -x0/var - (d-1)/2 + e^{-x0} sum x_j^2 / 2,  s_j = -x_j e^{-x0}   (coordinate 0 = element (0, 0) of the g = 0 lanes of
          // coordinate tile 0).  The sums run over all coordinates: the four lane groups, and with d > 32 the two waves of the team --
          // they meet in the delta plane that has been free since the last barrier (one more barrier, this configuration only).
          const bool own0 = ct == 0 && mb == 0;  // this wave holds coordinate 0
          float x0 = __shfl(x.m[0][0], n), c0 = __shfl(cvec.m[0][0], n);
          float sq = 0.0f, cx = 0.0f;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bool first = m == 0 && q == 0 && g == 0 && own0;
              sq = fmaf(first ? 0.0f : x.m[m][q], x.m[m][q], sq);
              cx = fmaf(first ? 0.0f : cvec.m[m][q], x.m[m][q], cx);
            }
          sq = sum_xor32(sum_xor16(sq));
          cx = sum_xor32(sum_xor16(cx));
          constexpr int NS = OTD == 2 ? NW : NW / 2;  // waves whose coordinates differ: all of them, or those of row tile 0
          if constexpr (NS > 1) {
            float* __restrict__ sx = Dp[LH & 1];
            const int slot = OTD == 2 ? wave : mb;
            if (g == 0 && (OTD == 2 || r == 0)) {
              sx[(2 * slot) * RS + n] = sq;
              sx[(2 * slot + 1) * RS + n] = cx;
              if (own0) { sx[16 * RS + n] = x0; sx[17 * RS + n] = c0; }
            }
            ws_barrier();
            sq = 0.0f;
            cx = 0.0f;
#pragma unroll
            for (int k = 0; k < NS; ++k) { sq += sx[(2 * k) * RS + n]; cx += sx[(2 * k + 1) * RS + n]; }
            x0 = sx[16 * RS + n];
            c0 = sx[17 * RS + n];
          }
          const float iv = __expf(-x0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) vt.m[m][q] = iv * (c0 * x.m[m][q] - cvec.m[m][q]);
          if (g == 0 && own0) vt.m[0][0] = c0 * (-1.0f / A.target.p0 - 0.5f * iv * sq) + iv * cx;
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) lam.m[m][q] = fmaf(jac_t, vt.m[m][q], fmaf(c_x, lam.m[m][q], dx.m[m][q]));
        if constexpr (KLB) {  // Bridge, method kl: the inference terms' d loss / d x_t
          const V lin = load8c(A.lam_in + (long long)t * d * B);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) lam.m[m][q] += live ? lin.m[m][q] : 0.0f;  // (lanes beyond the batch shadow its last row: nothing from them)
        }
      if (jac_p != 0.0f || refc) {
        const V pis = rows_t<MT>(tabs + 1 * 64 + cb);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = fmaf(-jac_p * pis.m[m][q], cvec.m[m][q], lam.m[m][q]);  // Gaussian prior: J = -1/sigma^2
            if (refc) v = fmaf(sig * pis.m[m][q], Gc.m[m][q], v);              // cost depends on x through sigma * prior.score(x)
            lam.m[m][q] = v;
          }
      }
      par ^= 1;
    }
  }

  // ---- the team's partial gradients (record layout of sdeh_bwdf.hip) --------------------------------------------------------
  float* __restrict__ rec = A.wpart + (long long)team_g * A.wsize;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int c = 0; c < NCI; ++c) store_tile(rec, DPP, 32 * r + mo + 16 * m, 16 * c, n, g, dw_in[m][c]);
#pragma unroll
    for (int l = 0; l < LH; ++l) {
#pragma unroll
      for (int c = 0; c < 4; ++c) store_tile(rec + off_whid<OTD>() + l * 4096, 64, 32 * r + mo + 16 * m, 16 * c, n, g, dw_hid[l][m][c]);
      float b = bs_hid[l][m];
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      if (g == 0) rec[off_bhid<OTD, LH>() + l * 64 + 32 * r + mo + 16 * m + n] = b;
    }
    float b = bs_out[m];
    b += __shfl_xor(b, 16);
    b += __shfl_xor(b, 32);
    if constexpr (OTD == 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c) store_tile(rec + off_wout<OTD, LH>(), 64, 32 * r + mo + 16 * m, 16 * c, n, g, dw_out[m][c]);
      if (g == 0) rec[off_bout<OTD, LH>() + 32 * r + mo + 16 * m + n] = b;
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c) store_tile(rec + off_wout<OTD, LH>(), 64, mo + 16 * m, 32 * r + 16 * c, n, g, dw_out[m][c]);
      if (g == 0 && r == 0) rec[off_bout<OTD, LH>() + mo + 16 * m + n] = b;
    }
  }
}

template <int OTD, int LH, int NW, bool KLB = false, bool ZIN = false>
static int launch_bwdf16_t(const BwdfArgs& a, hipStream_t stream) {
  if constexpr (LH == 2 && NW == 4 && !KLB) {
    if (a.cost_in != nullptr && a.lam_in != nullptr) return launch_bwdf16_t<OTD, LH, NW, true>(a, stream);
  }
  if constexpr (NW == 4 && !KLB && !ZIN) {  // (teams of four read the record; the two-wave comparison form re-evaluates)
    if (a.zrec != nullptr) return launch_bwdf16_t<OTD, LH, NW, false, true>(a, stream);
  }
  if (!KLB && (a.cost_in != nullptr || a.lam_in != nullptr)) return SDEH_ERR_UNSUPPORTED;  // (two hidden layers, four-wave teams, both planes)
  const size_t lds_bytes = (size_t)bwdf16::lds_floats<OTD, LH>() * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf16_kernel<OTD, LH, NW, KLB, ZIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf16_kernel<OTD, LH, NW, KLB, ZIN>), dim3((unsigned)a.n_slots), dim3(64 * NW), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

template <int OTD, int NW>
static int launch_bwdf16_l(const BwdfArgs& a, hipStream_t stream) {
  switch (a.n_hidden) {
    case 1: return launch_bwdf16_t<OTD, 1, NW>(a, stream);
    case 2: return launch_bwdf16_t<OTD, 2, NW>(a, stream);
    case 3: return launch_bwdf16_t<OTD, 3, NW>(a, stream);
    default: return SDEH_ERR_UNSUPPORTED;
  }
}

// Tiles of 16 serve back-propagation through time below 16 384 trajectories, for activations without a kink (header).  Measured
// (profiles/r02_tile16_timing.txt; d = 2, T = 100 / d = 50, T = 200, backward kernel in ms, tiles of 16 against tiles of 32):
// B = 2048: 0.63 / 1.73 and 1.40 / 4.85; 8192: 0.89 / 1.75 and 2.82 / 5.17; 16 384: 1.73 / 1.77 and 5.39 / 5.29 -- from there on
// the 32-trajectory kernel fills the chip as well and needs fewer instructions per trajectory.  SDEH_BWD_TILE=16 | 32 forces either.
int bwdf_tile(long long batch, bool bptt, int act) {
  if (!bptt || act == SDEH_ACT_RELU) return 32;
  const char* force = plan_opt(OPT_BWD_TILE);  // (a plan option: tests switch it)
  if (force != nullptr && force[0] == '1') return 16;
  if (force != nullptr && force[0] == '3') return 32;
  return batch < 16384 ? 16 : 32;
}

// teams (= workgroups = partial records) of a 16-trajectory launch: one per tile up to two per CU, persistent beyond
int bwdf16_slots(long long batch) {
  const long long tiles = (batch + 15) / 16;
  return (int)(tiles < 512 ? tiles : 512);
}

// waves per team: four.  Measured against two (tools/tile16_waves_timing.sh): 0.63 against 0.97 ms at B <= 4096 (d = 2, T = 100: every
// wave has a SIMD of its own), and still 0.89 against 0.99 ms at 8192, where two workgroups of four share a CU -- two waves per SIMD
// (252-288 registers each) fill each other's LDS round trips.  SDEH_BWD_WAVES=2 keeps the two-wave teams reachable (tests).
int bwdf16_waves(long long batch) {
  const char* force = plan_opt(OPT_BWD_WAVES);
  if (force != nullptr && (force[0] == '2' || force[0] == '4')) return force[0] - '0';
  (void)batch;
  return 4;
}

int launch_bwdf16(const BwdfArgs& a, hipStream_t stream) {
  if (a.flags & SDEH_FLAG_CHANGE_SDE_CTRL) return SDEH_ERR_UNSUPPORTED;  // row-parallel modes have (step, tile) items to spread
  if (bwdf16_waves(a.batch) == 4) return a.d <= 32 ? launch_bwdf16_l<1, 4>(a, stream) : launch_bwdf16_l<2, 4>(a, stream);
  return a.d <= 32 ? launch_bwdf16_l<1, 2>(a, stream) : launch_bwdf16_l<2, 2>(a, stream);
}

}  // namespace sdeh
