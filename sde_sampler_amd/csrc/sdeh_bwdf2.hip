// Fused training backward, trajectory-split teams (round 4).  Same contract, inputs, outputs and partial-
// gradient records as sdeh_bwdf.hip (the reference's loss.backward() through losses/oc.py:176-222 / 301-334 / 416-446 and
// models/mlp.py:114-122, solver/base.py:399-407); what changes is who owns what:
//
//   * sdeh_bwdf.hip splits a team's two waves by CHANNEL tile: every layer's input is both waves' output, so the chain of a step
//     (re-evaluation + adjoint) crosses eight workgroup barriers, each followed by an LDS round trip, with one wave per SIMD and
//     nothing to fill the gaps; with d <= 32 both waves mirror the out layer and the input gradient (64 of 296 MFMAs per step).
//   * here a wave owns 32 trajectories and ALL 64 channels.  The accumulator layout of v_mfma_f32_32x32x2_f32 is also its B-operand
//     layout (sdeh_common.hpp), so a layer's activated accumulators feed the next layer's MFMAs straight from registers: the chain
//     of a step -- forward and transposed -- touches LDS only for the weights and crosses NO barrier.  Two row tiles per layer are
//     two independent accumulator chains (no dependent-issue gap), operands are requested one k-group ahead.
//   * the weight gradients contract over trajectories, i.e. need both operands transposed (lane = row): a wave publishes delta_k and
//     a_k of its 32 trajectories to its own two LDS planes, and the team's two waves each take ONE ROW TILE of every gradient over the
//     team's 64 trajectories (half the accumulators per wave: 6-8 tiles, the same record as sdeh_bwdf.hip).  Two barriers per product,
//     both off the critical path of the arithmetic: the operands of a product are loaded into registers between them, and the
//     product's 64 MFMAs are then issued INTERLEAVED with the chain's next transposed layer (four independent accumulators).
//   * no mirror work; the funnel's per-trajectory sums and d loss / d gamma(t) need no cross-wave exchange (a wave has all coordinates).
//   * with two coordinate tiles and two hidden layers the pre-activations are kept instead of (act, act'): act / act' are evaluated
//     again in the backward stage that needs them (~2 k cycles per step) -- the price of holding 64 channels x 3 layers x 32
//     trajectories next to eight gradient tiles in one wave's registers.
//
// Per step and SIMD: 528 (d <= 8) .. 752 (d = 50) matrix instructions for 32 trajectories against 2 x 296 (2 x 392) in sdeh_bwdf.hip.
// One to two hidden layers; three keep sdeh_bwdf.hip.
#include "sdeh_bwdf2.hpp"
#ifdef SDEH_BWDF_PROFILE
#include <cstdio>
#endif

namespace sdeh {

#ifdef SDEH_BWDF_PROFILE
__device__ unsigned long long bwdf2_prof[16];
#define BW2_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define BW2_ADD(k, t0, t1) do { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&bwdf2_prof[k], (t1) - (t0)); } while (0)
#else
#define BW2_T(var) do {} while (0)
#define BW2_ADD(k, t0, t1) do {} while (0)
#endif


// NQ: accumulator registers of a coordinate tile that can hold live coordinates (d <= 8: 4, d <= 16: 8, else 16; two coordinate tiles: 16):
// loads, the elementwise phase and the publishes of x / delta_out loop over those only.  VIO (d <= 4): the input layer, the out layer
// and their transposes run on the vector pipe (2 x 64 weights per coordinate: 32 FMAs per lane and coordinate instead of 8 + 32 + 8 +
// 32 matrix instructions on tiles that are 7/8 padding); their weight gradients stay on the matrix pipe (off the chain).
// BR (Bridge, inference network; row-parallel): the cost's u + v enters the upstream gradient (gextra), the LerpPrior score is evaluated
// here (the Bridge forward keeps no score plane), the divergence term adds its gamma(t) part and S_k (sdeh_bridgef.hip) at every layer.
// BR = 2: also d loss / d x_t of the inference network's terms (W_in^T adj(Z_0) + the score term's Jacobian) as a plane [T, d, B]: with method kl the
// generative network's back-propagation through time adds it to its adjoint at every step.
// KLB (through time, a Bridge's generative network with method kl): the running cost on the plane cost_in = u + v, lam_in added to the adjoint.
// ZIN (round 5): the network is NOT re-evaluated.  The training forward (sdeh_simulate_fwd_train3) kept the
// pre-activations of every layer as a record in this kernel's own register layout (sdeh_traj_ws.hpp: ZRec -- lane (j, h) of the wave that
// owns a tile loads with eight 16-byte loads per layer exactly the 32 values of its accumulator tiles) and the raw network output [T, d, B].
// The reference's autograd keeps the activations as well (losses/oc.py:232-256 -> models/mlp.py:114-122).  Z_k is requested one stage
// ahead of the stage that needs act(Z_k) / act'(Z_k) and activated there: of the network only ONE layer's act' (32 registers) and the next
// layer's Z (32) are alive at any time instead of act / act' of all layers (160), and the 2 Lh C^2 + 2 d C re-evaluated products per
// trajectory-step are gone (1.70 x the algorithmic matrix work in round 4's PMC pass).  ReLU units on their kink take the forward
// launch's side by construction.
template <int OTD, bool BPTT, int LH, bool ZIN, int NQ, bool VIO, bool JAC = false, int BR = 0, bool KLB = false>
__global__ __launch_bounds__(256) void bwdf2_kernel(const BwdfArgs A) {
  using namespace bwdf2;
  static_assert(OTD == 1 || NQ == 16, "two coordinate tiles: all registers live");
  static_assert(!JAC || (VIO && !BPTT), "the Jacobian pass is row-parallel, d <= 4");
  static_assert(!VIO || (OTD == 1 && NQ == 4), "vector-pipe in / out layers: d <= 4");
  static_assert(!BR || (!BPTT && !JAC && !ZIN), "the Bridge form is row-parallel (and re-evaluates the inference network)");
  static_assert(!(ZIN && JAC) || LH == 2, "the Jacobian pass that reads the record: two hidden layers (three buffers)");
  static_assert(!KLB || BPTT, "cost_in / lam_in belong to back-propagation through time");
  constexpr bool WDX = BPTT || BR == 2;  // the chain goes on through the input layer: W_in^T delta_0
  constexpr int RSI = rsi<OTD>(), DPP = 32 * OTD;
  constexpr int NGI = OTD == 2 ? 8 : NQ / 4;              // k-groups of the coordinates
  // two coordinate tiles: x_t is read again where it is needed behind the input layer (the Jacobians of the closed-form scores, the
  // operand of input_embed.weight's gradient) instead of occupying 32 registers for the whole step
  constexpr bool XRELOAD = OTD == 2 && BPTT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ Win = lds;
  float* __restrict__ Whid = Win + 64 * RSI;
  float* __restrict__ Wout = Whid + LH * 64 * RSW;
  float* __restrict__ bh = Wout + DPP * RSW;
  float* __restrict__ bo = bh + LH * 64;
  float* __restrict__ tabs = bo + 64;
  float* __restrict__ planes = tabs + TABS;
  float* __restrict__ vio_in = planes + 2 * 4 * PLANE;  // [4][64]: column i of input_embed.weight in accumulator order (VIO)
  float* __restrict__ vio_out = vio_in + 256;           // [4][64]: row i of out_layer.weight in accumulator order
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  float* __restrict__ Dme = planes + wave * 2 * PLANE;  // wave w: D plane at w * 2 PLANE, A plane behind it
  float* __restrict__ Ame = Dme + PLANE;
  // this wave's tile of every weight gradient: hidden layers and (two coordinate tiles) input_embed / out_layer: tile (wR, wC) of the
  // 2 x 2 over all four waves' trajectories; one coordinate tile: input_embed row tile wC / out_layer channel tile wC over the
  // trajectories of the wave pair wR (the two pairs' sums meet in LDS at the end)
  const int wR = wave >> 1, wC = wave & 1;
  const int d = A.d, T = A.n_steps;
  const long long B = A.batch;

  // ---- stage the parameters (natural layouts, zero padding) and the Gaussian tables ----------------------------------------
  for (int idx = tid; idx < 64 * RSI; idx += 256) {
    const int row = idx / RSI, col = idx - row * RSI;
    Win[idx] = col < d ? A.w_in[row * d + col] : 0.0f;
  }
  for (int idx = tid; idx < LH * 64 * RSW; idx += 256) {
    const int l = idx / (64 * RSW), rem = idx - l * 64 * RSW, row = rem / RSW, col = rem - row * RSW;
    Whid[idx] = col < 64 ? A.w_hid[l][row * 64 + col] : 0.0f;
  }
  for (int idx = tid; idx < DPP * RSW; idx += 256) {
    const int row = idx / RSW, col = idx - row * RSW;
    Wout[idx] = (row < d && col < 64) ? A.w_out[row * 64 + col] : 0.0f;
  }
  if (tid < LH * 64) bh[tid] = A.b_hid[tid >> 6][tid & 63];
  if (tid < 64) bo[tid] = tid < d ? A.b_out[tid] : 0.0f;
  for (int idx = tid; idx < TABS; idx += 256) {  // tabs[(2 k + c) * 64 + coordinate]: c = 0 mean, 1 inverse variance; k = prior, second, target
    const int k = idx / 128, c = (idx >> 6) & 1, cj = idx & 63;
    const int which = k == 0 ? 1 : (k == 1 ? 2 : 0);
    tabs[idx] = cj < d && cj < L.dp ? ws[L.dg[which] + 2 * cj + c] : 0.0f;
  }
  for (int idx = tid; idx < 2 * 4 * PLANE; idx += 256) planes[idx] = 0.0f;  // never-written rows / trajectories must not hold NaNs
  if constexpr (VIO) {
    const int i = tid >> 6, m = tid & 63, ch = 32 * (m >> 5) + rrow(m & 15) + 4 * ((m >> 4) & 1);  // m = (R * 2 + h) * 16 + q
    vio_in[tid] = i < d ? A.w_in[ch * d + i] : 0.0f;
    vio_out[tid] = i < d ? A.w_out[i * 64 + ch] : 0.0f;
  }
  __syncthreads();

#ifdef SDEH_BWDF2_ACT
  const int act = SDEH_BWDF2_ACT, ctrl_kind = A.ctrl_kind, flags = A.flags;
#else
  const int act = A.act, ctrl_kind = A.ctrl_kind, flags = A.flags;
#endif
  const bool has_score = ctrl_kind != SDEH_CTRL_CLIPPED;
  const bool refc = (flags & SDEH_FLAG_REFERENCE_CTRL) && A.loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool expo = A.loss_kind == SDEH_LOSS_EXPONENTIAL;
  const bool ito = (flags & SDEH_FLAG_ITO) != 0;
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);

  f32x16 dw_in, dw_out, dw_hid[LH];
  f32x4b dwi4[2], dwo4[2];  // VIO: the [64, d] / [d, 64] gradients as 4 x 4 blocks over this wave's own trajectories (two chains each)
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dwi4[u][e] = 0.0f; dwo4[u][e] = 0.0f; }
#pragma unroll
  for (int q = 0; q < 16; ++q) { dw_in[q] = 0.0f; dw_out[q] = 0.0f; }
#pragma unroll
  for (int l = 0; l < LH; ++l)
#pragma unroll
    for (int q = 0; q < 16; ++q) dw_hid[l][q] = 0.0f;
  float bs_hid[LH], bs_out = 0.0f;
#pragma unroll
  for (int l = 0; l < LH; ++l) bs_hid[l] = 0.0f;

  // Items: quads of 32-trajectory tiles (through time) or (step, quad), step-major (row-parallel); the workgroup is one team and takes
  // item blockIdx, blockIdx + gridDim, ...; wave w owns tile 4 quad + w (a tile count that is no multiple of four leaves the last
  // quad's spare waves shadowing the last tile with zero weights).
  const int n_tiles = A.n_tiles, n_pairs = (n_tiles + 3) >> 2;
  const int n_teams = (int)gridDim.x, team_g = (int)blockIdx.x;
  const long long n_items = BPTT ? (long long)n_pairs : (long long)n_pairs * T;
  const long long n_rounds = (n_items + n_teams - 1) / n_teams;
  const int d_t = BPTT ? 0 : n_teams / n_pairs, d_pair = BPTT ? n_teams : n_teams % n_pairs;
  int it_t = BPTT ? T - 1 : team_g / n_pairs, it_pair = BPTT ? team_g : team_g % n_pairs;  // the current item
  auto advance = [&](int& t_io, int& pair_io) {
    pair_io += d_pair;
    if constexpr (!BPTT) {
      t_io += d_t;
      if (pair_io >= n_pairs) { pair_io -= n_pairs; t_io += 1; }
    }
  };
  auto item_live = [&](int t_i, int pair_i) { return BPTT ? pair_i < n_pairs : t_i < T; };
  auto clamp_item = [&](int& t_io, int& pair_io) {  // a team without an item shadows the last one (and contributes zeros)
    if (!item_live(t_io, pair_io)) { t_io = T - 1; pair_io = n_pairs - 1; }
  };
  auto tile_of = [&](int pair_i) { const int tl = 4 * pair_i + wave; return tl < n_tiles ? tl : n_tiles - 1; };

  // NQ coordinates (32 ct + 4 h + rrow(q)) of column `col` of a coordinate-major plane [d][B] with a WAVE-UNIFORM start: one scalar
  // base + a 32-bit byte offset per element, the lane's part opaque per call (sdeh_bwdf.hip: 64-bit element addresses become loop
  // invariants that spill); coordinates >= d read a valid element and are zeroed by a select.  Registers >= NQ: zero.
  // The element offsets are formed on the vector pipe from an opaque copy of the row stride: as scalar expressions each of the 16
  // rows of every plane gets its own 64-bit scalar base, a few hundred scalar registers in all -- they spill to vector-register lanes,
  // and every use costs a v_readlane plus the wait states behind it (966 + 553 instructions in the step loop of the d = 50 kernel).
  const unsigned Bu = (unsigned)B;
  auto load_cm = [&](const float* __restrict__ plane_u, unsigned col, int ct) {
    f32x16 v;
    const int cb = 32 * ct + 4 * h;
    unsigned lane_off = ((unsigned)(cb < d ? cb : 0) * Bu + col) * 4u;
    unsigned s1 = Bu * 4u;
    asm volatile("" : "+v"(lane_off), "+v"(s1));
    const unsigned s2 = s1 + s1, s3 = s2 + s1, s8 = s1 << 3;
    const char* __restrict__ pb = reinterpret_cast<const char*>(plane_u);
    unsigned bg = lane_off;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < NQ) {
        const bool ok = cb + rrow(q) < d;
        const unsigned oq = (q & 3) == 0 ? bg : ((q & 3) == 1 ? bg + s1 : ((q & 3) == 2 ? bg + s2 : bg + s3));
        const unsigned off = ok ? oq : lane_off;
        const float val = *reinterpret_cast<const float*>(pb + off);
        v[q] = ok ? val : 0.0f;
        if ((q & 3) == 3) bg += s8;
      } else {
        v[q] = 0.0f;
      }
    }
    return v;
  };
  auto load_x = [&](int t, int tile_i, f32x16 (&xo)[OTD]) {
    const long long rw = (long long)tile_i * 32 + j;
    const unsigned col = (unsigned)(rw < B ? rw : B - 1);
#pragma unroll
    for (int ct = 0; ct < OTD; ++ct) xo[ct] = load_cm(A.xs + (long long)t * d * B, col, ct);
  };
  auto load_emb = [&](int t, f32x16 (&eo)[2]) {  // timestep_embed(t) + input bias, accumulator order: the FIRST addend of the input layer
#pragma unroll
    for (int R = 0; R < 2; ++R) eo[R] = load16(ws + L.emb + t * C + (R * 2 + h) * 16);
  };
  // the per-step scalars travel with x: fetched in front of the previous step's last block of matrix instructions (scalar loads share
  // the LDS counter: requested at the top of a step they hold up the first weight read)
  struct StepCoef { float sig, wl, c_i, cdt, c_u, c_x, gam0; };
  auto load_coef = [&](int t) {
    StepCoef c;
    cfp cf = as_const(ws + L.coef + t * kCoefStride);
    c.sig = cf[CF_SIGMA]; c.wl = cf[CF_W];
    c.c_i = expo ? cf[CF_SBK] : cf[CF_SQDT];
    c.cdt = expo ? cf[CF_B2S2] : cf[CF_DT];
    c.c_u = expo ? cf[CF_B2S2] : c.sig * cf[CF_DT];
    c.c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], cf[CF_DT], 1.0f);
    c.gam0 = has_score ? as_const(ws + L.gam + t * L.g)[0] : 0.0f;  // gamma(t) (its first entry)
    return c;
  };

  // ZIN: layer k of the pre-activation record of (step t, tile): eight 16-byte loads per lane from a wave-uniform base (ZRec layout:
  // [step][tile][layer][quad 8 R + 2 g + h][trajectory j][4]); every word is read once -- non-temporal
  typedef float f32x4z __attribute__((ext_vector_type(4)));
  auto load_z = [&](int t, long long tile_i, int k, f32x16 (&zo)[2]) {
    if constexpr (ZIN) {
      const float* __restrict__ base = A.zrec + ((long long)t * n_tiles + tile_i) * ((LH + 1) * 2048 + OTD * 1024) + k * 2048;
      unsigned lo = (unsigned)(h * 128 + j * 4) * 4u;
      asm volatile("" : "+v"(lo));
      const char* __restrict__ pb = reinterpret_cast<const char*>(base) + lo;
#pragma unroll
      for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4z v = __builtin_nontemporal_load(reinterpret_cast<const f32x4z*>(pb + (R * 1024 + g * 256) * 4));
          zo[R][4 * g] = v[0]; zo[R][4 * g + 1] = v[1]; zo[R][4 * g + 2] = v[2]; zo[R][4 * g + 3] = v[3];
        }
    }
  };

  // ZIN: the raw network output of (step, tile), coordinate tile ct: the record's last part, in the accumulator layout as well (coordinate
  // quads 2 g + h); quads the forward's vector wave did not write (d <= 4: beyond the first) and coordinates >= d: zero by a select
  auto load_nn = [&](int t, long long tile_i, int ct, f32x16& no) {
    if constexpr (ZIN) {
      const float* __restrict__ base = A.zrec + ((long long)t * n_tiles + tile_i) * ((LH + 1) * 2048 + OTD * 1024) + (LH + 1) * 2048 + ct * 1024;
      unsigned lo = (unsigned)(h * 128 + j * 4) * 4u;
      asm volatile("" : "+v"(lo));
      const char* __restrict__ pb = reinterpret_cast<const char*>(base) + lo;
      const int cb = 32 * ct + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (4 * g < NQ) {
          // (a quad beyond d was never written: read quad 0 of this lane half instead -- a valid address -- and select zeros)
          const bool okq = cb + 8 * g < d;
          const f32x4z v = __builtin_nontemporal_load(reinterpret_cast<const f32x4z*>(pb + (okq ? g * 256 * 4 : 0)));
#pragma unroll
          for (int e = 0; e < 4; ++e) no[4 * g + e] = cb + 8 * g + e < d ? v[e] : 0.0f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) no[4 * g + e] = 0.0f;
        }
      }
    }
  };
  // ZIN: two record buffers in flight -- layer k of a step lives in zA when LH - k is even, else in zB, and is requested where the
  // buffer's previous content was activated: Z_LH of the NEXT step in front of this step's last hidden stage, Z_{LH-1} of the next step in
  // front of the in stage, Z_{LH-2} of this step at its top -- every request a whole stage (8-10 k cycles of matrix work) ahead of its use.
  // Two coordinate tiles through time (the kernel's register peak is its two-tile elementwise phase): ONE stage shallower -- Z_LH of the next
  // step in front of the in stage, Z_{LH-1} at the top of the step, the score plane per tile inside the elementwise phase: 412 -> 224 B of
  // scratch per lane, 13.1 -> 11.3 ms at d = 50, B = 65 536, T = 200
  f32x16 xnext[OTD], embnext[2], zA[2], zB[2], zC[2];  // (zC: the Jacobian pass holds all three layers of the next item)
  StepCoef cnext;
  {
    int t0 = it_t, p0 = it_pair;
    clamp_item(t0, p0);
    load_x(t0, tile_of(p0), xnext);
    if constexpr (!ZIN) load_emb(t0, embnext);
    cnext = load_coef(t0);
    load_z(t0, tile_of(p0), LH, zA);
    if constexpr (!(OTD == 2 && BPTT)) load_z(t0, tile_of(p0), LH - 1, zB);  // (two tiles through time: requested at the top of the step)
    if constexpr (ZIN && JAC) load_z(t0, tile_of(p0), 0, zC);
  }
  for (long long round = 0; round < n_rounds; ++round) {
    const bool live_item = item_live(it_t, it_pair);
    int cur_t = it_t, cur_pair = it_pair;
    clamp_item(cur_t, cur_pair);
    const bool live_tile = live_item && 4 * cur_pair + wave < n_tiles;
    const long long tile = tile_of(cur_pair);
    const long long row = tile * 32 + j;
    const bool live = live_tile && row < B;
    const long long lrow = row < B ? row : B - 1;
    const float wi = live ? A.grad_rnd[lrow] : 0.0f;
    const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);

    f32x16 lam[OTD];
#pragma unroll
    for (int ct = 0; ct < OTD; ++ct)
#pragma unroll
      for (int q = 0; q < 16; ++q) lam[ct][q] = 0.0f;
    if constexpr (BPTT) {  // lambda_T = w_i d(terminal costs)/dx_T  (losses/oc.py:225,337,449-450)
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) {
        const int cb = 32 * ct + 4 * h;
        if (flags & SDEH_FLAG_TERMINAL_SECOND) {
          const f32x16 xT = load_cm(A.xs + (long long)T * d * B, (unsigned)lrow, ct);
          const f32x16 smu = rows16(tabs + 2 * 64 + cb), sis = rows16(tabs + 3 * 64 + cb);
#pragma unroll
          for (int q = 0; q < NQ; ++q) lam[ct][q] = wi * (smu[q] - xT[q]) * sis[q];
        }
        if ((flags & SDEH_FLAG_TERMINAL_TARGET) && A.tscore != nullptr) {
          const f32x16 st = load_cm(A.tscore, (unsigned)lrow, ct);
#pragma unroll
          for (int q = 0; q < NQ; ++q) lam[ct][q] = fmaf(-wi, st[q], lam[ct][q]);
        }
      }
    }
    const int t_first = cur_t;
    const int t_last = BPTT ? 0 : t_first;
    advance(it_t, it_pair);  // from here on: the team's NEXT item

    for (int t = t_first; t >= t_last; --t) {
      BW2_T(tp0);
      // The read-only LDS tables through bases the compiler cannot see through: their loads are invariants of the step loop (no store
      // aliases the restrict-qualified tables), and hoisted out of it they occupy -- and spill -- hundreds of registers.
      int opq = 0;
      asm volatile("" : "+v"(opq));
      const float* __restrict__ Win_s = Win + opq;
      const float* __restrict__ Whid_s = Whid + opq;
      const float* __restrict__ Wout_s = Wout + opq;
      const float* __restrict__ bh_s = bh + opq;
      const float* __restrict__ bo_s = bo + opq;
      const float* __restrict__ tabs_s = tabs + opq;
      const float* __restrict__ vin_s = vio_in + opq;
      const float* __restrict__ vout_s = vio_out + opq;
      f32x16 x[OTD], embv[2];
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) x[ct] = xnext[ct];
      if constexpr (!ZIN) { embv[0] = embnext[0]; embv[1] = embnext[1]; }
      const StepCoef cs = cnext;
      const float sig = cs.sig, wl = cs.wl, c_i = cs.c_i, cdt = cs.cdt, c_u = cs.c_u, c_x = cs.c_x, gam0 = cs.gam0;

      // ======================================================================================= forward (re-evaluation at x_t)
      // keep[k] = act'(Z_k)  (k = 0 .. LH);   akeep[k] = a_{k+1} = act(Z_k)  (k < LH)
      // ZIN: no re-evaluation -- zA / zB hold the records in flight, kz = act' of the layer activated last
      f32x16 keep[(ZIN && !JAC) ? 1 : LH + 1][2];
      f32x16 akeep[ZIN ? 1 : LH][2];
      f32x16 cur[2], kz[2];
      f32x16 nn[OTD];
      // the score entering the control: requested in front of the out layer, consumed behind it
      f32x16 scv[OTD];
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) scv[ct][q] = 0.0f;
      if constexpr (ZIN && JAC) {
        // the Jacobian pass on the record: act' of the three layers (requested an item ago) and the raw network output, no matrix work
        load_nn(t, tile, 0, nn[0]);
        f32x16 unused[2];
        SDEH_ACT_SWITCH(act, ACT, {
          act_both<ACT>(zC[0], unused[0], keep[0][0]); act_both<ACT>(zC[1], unused[1], keep[0][1]);
          SDEH_FENCE();
          act_both<ACT>(zB[0], unused[0], keep[1][0]); act_both<ACT>(zB[1], unused[1], keep[1][1]);
          SDEH_FENCE();
          act_both<ACT>(zA[0], unused[0], keep[2][0]); act_both<ACT>(zA[1], unused[1], keep[2][1]);
        });
        SDEH_FENCE();
      } else if constexpr (ZIN) {
        // raw network output and score planes first (their latency hides behind the activation of Z_LH, which arrived a step ago)
#pragma unroll
        for (int ct = 0; ct < OTD; ++ct) load_nn(t, tile, ct, nn[ct]);
        if (has_score && !(OTD == 2 && BPTT)) {  // (two coordinate tiles through time: requested per tile, inside the elementwise phase)
#pragma unroll
          for (int ct = 0; ct < OTD; ++ct) scv[ct] = load_cm(A.sc + (long long)t * d * B, (unsigned)lrow, ct);
        }
        SDEH_FENCE();
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(zA[0], cur[0], kz[0]); SDEH_FENCE(); act_both<ACT>(zA[1], cur[1], kz[1]););
        SDEH_FENCE();
        plane_put(Ame, 0, j, h, cur[0]);  // a_{LH+1}: free since the previous product's second barrier
        plane_put(Ame, 1, j, h, cur[1]);
        SDEH_FENCE();
        if constexpr (LH >= 2) load_z(t, tile, LH - 2, zA);  // consumed two stages on
        if constexpr (OTD == 2 && BPTT) load_z(t, tile, LH - 1, zB);  // (two coordinate tiles through time: shallower prefetch, fewer registers in flight)
        SDEH_FENCE();
      }
      if constexpr (!ZIN) {
        f32x16 z[2] = {embv[0], embv[1]};
        if constexpr (VIO) {
          // emb + sum_i W_in[:, i] x_i in coordinate order: the fmaf chain the matrix instructions of the forward launch evaluate
          // (their other k values meet zero weights)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < d) {
              const float xb = bcast_lo(x[0][i]);
              const f32x16 w0 = load16(vin_s + i * 64 + h * 16), w1 = load16(vin_s + i * 64 + 32 + h * 16);
#pragma unroll
              for (int q = 0; q < 16; ++q) { z[0][q] = fmaf(w0[q], xb, z[0][q]); z[1][q] = fmaf(w1[q], xb, z[1][q]); }
            }
          }
        } else {
          fwd_rows<NGI, OTD, 2>(Win_s + j * RSI + 4 * h, RSI, x, NGI, z);  // all k-groups of the tile class: no branch around matrix instructions
        }
        SDEH_FENCE();
        BW2_T(tpa);
        BW2_ADD(9, tp0, tpa);
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z[0], cur[0], keep[0][0]); SDEH_FENCE(); act_both<ACT>(z[1], cur[1], keep[0][1]););
        SDEH_FENCE();
        akeep[0][0] = cur[0]; akeep[0][1] = cur[1];
        SDEH_FENCE();
        BW2_T(tpb);
        BW2_ADD(10, tpa, tpb);
      }
      BW2_T(tpc);
      if constexpr (!ZIN) {
#pragma unroll
      for (int l = 0; l < LH; ++l) {  // hidden layer l: Z_{l+1} = W_l a_{l+1} + b_l;  a_{l+2} = act(Z_{l+1})
        f32x16 z[2] = {rows16(bh_s + l * 64 + 4 * h), rows16(bh_s + l * 64 + 32 + 4 * h)};
        fwd_rows<8, 2, 2>(Whid_s + l * 64 * RSW + j * RSW + 4 * h, RSW, cur, 8, z);
        SDEH_FENCE();
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z[0], cur[0], keep[l + 1][0]); SDEH_FENCE(); act_both<ACT>(z[1], cur[1], keep[l + 1][1]););
        SDEH_FENCE();
        if (l + 1 < LH) { akeep[l + 1 < LH ? l + 1 : 0][0] = cur[0]; akeep[l + 1 < LH ? l + 1 : 0][1] = cur[1]; }
      }
      }
      BW2_T(tpd);
      BW2_ADD(11, tpc, tpd);
      SDEH_FENCE();
      if constexpr (ZIN) {
      } else if constexpr (BR) {
        if (has_score) {  // LerpPriorCtrl: (1 - t / T) prior.score(x), closed form (models/reparam.py:160-189)
#pragma unroll
          for (int ct = 0; ct < OTD; ++ct) {
            const f32x16 pmu = rows16(tabs_s + 0 * 64 + 32 * ct + 4 * h), pis = rows16(tabs_s + 1 * 64 + 32 * ct + 4 * h);
#pragma unroll
            for (int q = 0; q < NQ; ++q) scv[ct][q] = (1.0f - wl) * ((pmu[q] - x[ct][q]) * pis[q]);
          }
        }
      } else if (has_score) {
#pragma unroll
        for (int ct = 0; ct < OTD; ++ct) scv[ct] = load_cm(A.sc + (long long)t * d * B, (unsigned)lrow, ct);
      }
      SDEH_FENCE();
      if constexpr (ZIN) {
      } else {
      // a_{LH+1} goes to the A plane at once (free since the previous product's second barrier); the out layer reads the registers
      plane_put(Ame, 0, j, h, cur[0]);
      plane_put(Ame, 1, j, h, cur[1]);
      SDEH_FENCE();
      if constexpr (VIO) {
#pragma unroll
        for (int q = 0; q < 16; ++q) nn[0][q] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < d) {
            const f32x16 w0 = load16(vout_s + i * 64 + h * 16), w1 = load16(vout_s + i * 64 + 32 + h * 16);
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { p0 = fmaf(w0[q], cur[0][q], p0); p1 = fmaf(w1[q], cur[1][q], p1); }
            const float tot = sum_xor32(p0 + p1) + bo_s[i];
            nn[0][i] = h == 0 ? tot : 0.0f;  // coordinates 4 .. 7 (the h = 1 half) stay exactly zero
          }
        }
      } else {
#pragma unroll
        for (int ct = 0; ct < OTD; ++ct) nn[ct] = rows16(bo_s + 32 * ct + 4 * h);
        fwd_rows<8, 2, OTD>(Wout_s + j * RSW + 4 * h, RSW, cur, 8, nn);
      }
      }
      SDEH_FENCE();
      BW2_T(tp1);
      BW2_ADD(12, tpd, tp1);
      if constexpr (JAC) {
        // ===================================================================================== Jacobian pass (scan form of BPTT)
        // the raw network output and d nn_k / d x_i of this (step, trajectory): d reverse passes seeded with the unit vectors, no
        // weight gradients, no planes, no barrier.  (The clamp's mask and the score terms are applied by the scan kernel.)
        if (t > t_last) {
          load_x(t - 1, (int)tile, xnext);
          if constexpr (!ZIN) load_emb(t - 1, embnext);
          cnext = load_coef(t - 1);
        } else if (round + 1 < n_rounds) {
          int tn = it_t, pn = it_pair;
          clamp_item(tn, pn);
          load_x(tn, tile_of(pn), xnext);
          if constexpr (!ZIN) load_emb(tn, embnext);
          cnext = load_coef(tn);
          if constexpr (ZIN) { load_z(tn, tile_of(pn), 2, zA); load_z(tn, tile_of(pn), 1, zB); load_z(tn, tile_of(pn), 0, zC); }
        }
        const bool wr = live && h == 0;
        float* __restrict__ nno = A.nn_out + (long long)t * d * B + lrow;
        float* __restrict__ jo = A.jac_out + (long long)t * d * d * B + lrow;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k < d) {
            if (wr) nno[(long long)k * B] = nn[0][k];
            f32x16 dj[2] = {load16(vout_s + k * 64 + h * 16), load16(vout_s + k * 64 + 32 + h * 16)};  // row k of out_layer.weight
#pragma unroll
            for (int l = LH; l >= 0; --l) {
#pragma unroll
              for (int q = 0; q < 16; ++q) { dj[0][q] *= keep[l][0][q]; dj[1][q] *= keep[l][1][q]; }
              if (l > 0) {
                f32x16 dn[2];
                chain_cols<RSW>(Whid_s + (l - 1) * 64 * RSW + 4 * h * RSW + j, dj, dn);
                dj[0] = dn[0]; dj[1] = dn[1];
              }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (i < d) {
                const f32x16 w0 = load16(vin_s + i * 64 + h * 16), w1 = load16(vin_s + i * 64 + 32 + h * 16);
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) { p0 = fmaf(w0[q], dj[0][q], p0); p1 = fmaf(w1[q], dj[1][q], p1); }
                const float tot = sum_xor32(p0 + p1);
                if (wr) jo[((long long)k * d + i) * B] = tot;
              }
            }
          }
        }
        continue;
      }

      // ======================================================================================= upstream gradient of the control
      // (and, through time, everything of the adjoint update that does not need W_in^T delta_0)
      f32x16 dout[OTD];
      {
        float gsum = 0.0f;
        const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
        const float coef_t = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET ? wl : 0.0f);
        const float coef_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR ? 1.0f - wl : 0.0f;
        // score terms the reference detaches (reparam.py:58,134,169,188) or obtains by autograd without a graph carry no Jacobian
        const float jac_t = (!has_score || (flags & (SDEH_FLAG_DETACH_SCORE | SDEH_FLAG_TARGET_SCORE_CONST))) ? 0.0f : coef_t;
        const float jac_p = (!has_score || (flags & SDEH_FLAG_DETACH_SCORE)) ? 0.0f : coef_p;
        // one coordinate tile at a time, the adjoint's update included: nothing but dout and lambda of a tile outlives its turn (the
        // funnel's Jacobian couples all coordinates: one tile only -- launch_bwdf2 leaves funnels with d > 32 to sdeh_bwdf.hip)
#pragma unroll
        for (int ct = 0; ct < OTD; ++ct) {
          SDEH_FENCE();
          const int cb = 32 * ct + 4 * h;
          if constexpr (ZIN && OTD == 2 && BPTT) {
            if (has_score) scv[ct] = load_cm(A.sc + (long long)t * d * B, (unsigned)lrow, ct);
          }
          f32x16 cvec, Gc;
          f32x16 xe;  // x_t of this tile, where the adjoint needs it
          if constexpr (XRELOAD) {
#pragma unroll
            for (int q = 0; q < 16; ++q) xe[q] = 0.0f;
            if (BPTT && (refc || (jac_t != 0.0f && A.target.kind == SDEH_DENS_MULTI_WELL)))
              xe = load_cm(A.xs + (long long)t * d * B, (unsigned)lrow, ct);
          } else {
            xe = x[ct];
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) dout[ct][q] = 0.0f;
          f32x16 xi;
#pragma unroll
          for (int q = 0; q < NQ; ++q) xi[q] = 0.0f;
          if (ito) {
            float n[16];
            if (A.noise != nullptr) {
              const float* __restrict__ rowp = A.noise + ((long long)t * B + lrow) * d;
#pragma unroll
              for (int q = 0; q < NQ; ++q) n[q] = cb + rrow(q) < d ? rowp[cb + rrow(q)] : 0.0f;
            } else {
#pragma unroll
              for (int g4 = 0; g4 < NQ / 4; ++g4) {
                float n4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (cb + 8 * g4 < d) box_muller4(philox_block(A.seed, rng_off, grow, t, (cb + 8 * g4) >> 2), n4);
#pragma unroll
                for (int e = 0; e < 4; ++e) n[4 * g4 + e] = n4[e];
                SDEH_FENCE();
              }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) xi[q] = cb + rrow(q) < d ? n[q] : 0.0f;
          }
          SDEH_FENCE();
          f32x16 rr;  // reference control sigma * prior.score(x) (solver/oc.py:305-306)
#pragma unroll
          for (int q = 0; q < NQ; ++q) rr[q] = 0.0f;
          if (BPTT && refc) {
            const f32x16 pmu = rows16(tabs_s + 0 * 64 + cb), pis = rows16(tabs_s + 1 * 64 + cb);
#pragma unroll
            for (int q = 0; q < NQ; ++q) rr[q] = sig * (pmu[q] - xe[q]) * pis[q];
          }
          SDEH_FENCE();
          // the wave-uniform switches (has_score, ito, per-coordinate gamma) are folded into operands OUTSIDE the element loop: a uniform
          // condition evaluated per element becomes a scalar branch per element, and the accumulators that live across it get copied
          f32x16 mfv;  // mult * gamma(t)[coordinate]; zero without a score term (scv is zero then as well)
          if (!has_score) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mfv[q] = 0.0f;
          } else if (A.g == 1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mfv[q] = mult * gam0;
          } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mfv[q] = mult * ws[L.gam + t * L.g + min(cb + rrow(q), L.g - 1)];
          }
          SDEH_FENCE();
          const float c_ie = ito ? c_i : 0.0f;  // (xi is zero without the Ito term)
          // scan form of back-propagation through time: the upstream gradient of the control comes from the scan kernel
          const bool use_gq = !BPTT && A.gq_in != nullptr;
          f32x16 gql;
#pragma unroll
          for (int q = 0; q < NQ; ++q) gql[q] = 0.0f;
          if (use_gq) {
            gql = load_cm(A.gq_in + (long long)t * d * B, (unsigned)lrow, ct);
            if (!live) {  // lanes beyond the batch shadow its last row: they must contribute nothing (w_i = 0 does that for w_i dB)
#pragma unroll
              for (int q = 0; q < NQ; ++q) gql[q] = 0.0f;
            }
          }
          f32x16 gex, dsv;  // Bridge: u + v of the running cost; d (sigma dt div) / d gamma per coordinate (the score part of the divergence)
#pragma unroll
          for (int q = 0; q < NQ; ++q) { gex[q] = 0.0f; dsv[q] = 0.0f; }
          if constexpr (BR) {
            gex = load_cm(A.gextra + (long long)t * d * B, (unsigned)lrow, ct);
            if (ctrl_kind == SDEH_CTRL_LERP_PRIOR) {
              const f32x16 pis = rows16(tabs_s + 1 * 64 + cb);
              const float cdiv = -(wi * sig * cdt) * mult * (1.0f - wl);
#pragma unroll
              for (int q = 0; q < NQ; ++q) dsv[q] = cdiv * pis[q];
            }
          }
          // Bridge, method kl (through time): the control entering the running cost is u + v (a plane); the inference network's d loss / d x_t
          f32x16 cin, lin;
#pragma unroll
          for (int q = 0; q < NQ; ++q) { cin[q] = 0.0f; lin[q] = 0.0f; }
          if constexpr (KLB) {
            cin = load_cm(A.cost_in + (long long)t * d * B, (unsigned)lrow, ct);
            lin = load_cm(A.lam_in + (long long)t * d * B, (unsigned)lrow, ct);
            if (!live) {  // lanes beyond the batch shadow its last row: nothing from them
#pragma unroll
              for (int q = 0; q < NQ; ++q) lin[q] = 0.0f;
            }
          }
          f32x16 gcoord;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const float mfac = mfv[q];
            const float csc = clipf(scv[ct][q], A.clip_score);
            const float keep_s = fabsf(scv[ct][q]) <= A.clip_score ? 1.0f : 0.0f;
            float gc = wi * c_ie * xi[q];
            if constexpr (BR) gc = fmaf(wi * cdt, gex[q], gc);
            if constexpr (BPTT) {
              const float u = clipf(nn[ct][q], A.clip_model) + mfac * csc;
              gc = wi * fmaf(KLB ? cin[q] : u - rr[q], cdt, c_ie * xi[q]);
            }
            const float gq = BPTT ? fmaf(c_u, lam[ct][q], gc) : (use_gq ? gql[q] : gc);
            Gc[q] = gc;
            float gg = gq * mult * csc;
            if constexpr (BR) gg = fmaf(keep_s, dsv[q], gg);
            gcoord[q] = gg;
            gsum += gg;
            cvec[q] = keep_s * mfac * gq;
            dout[ct][q] = fabsf(nn[ct][q]) <= A.clip_model ? gq : 0.0f;
            if ((q & 3) == 3) SDEH_FENCE();
          }
          SDEH_FENCE();
          if (has_score && live_tile && A.g != 1) {  // d loss / d gamma(t) per coordinate: over the 32 trajectories of this lane half
            float mine = 0.0f;  // lane j < NQ keeps register j's sum: one store per lane half instead of a branch per register
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              const float v = sum_xor16(sum_row16(gcoord[q]));
              mine = j == q ? v : mine;
            }
            if (j < NQ) A.gpart[(tile * T + t) * A.gw + cb + (j & 3) + 8 * (j >> 2)] = mine;
          }
          SDEH_FENCE();
          if constexpr (BR == 2) {  // d (v's score term) / d x_t: the Gaussian prior's J = -1 / sigma^2 through the clamp (lam is zero at the item's start)
            if (jac_p != 0.0f) {
              const f32x16 pis = rows16(tabs_s + 1 * 64 + cb);
#pragma unroll
              for (int q = 0; q < NQ; ++q) lam[ct][q] = -jac_p * pis[q] * cvec[q];
            }
          }
          if constexpr (BPTT) {
            // ===================================================================================== adjoint update, first part
            //   lambda_t = c_x lambda_{t+1} + (d score term / d x)^T G + direct cost terms  [+ W_in^T delta_0 at the end of the step]
            f32x16 vt;
#pragma unroll
            for (int q = 0; q < NQ; ++q) vt[q] = 0.0f;
            if (jac_t != 0.0f) {  // closed-form target scores are differentiated through x
              if (A.target.kind == SDEH_DENS_DIAG_GAUSS) {
                const f32x16 tis = rows16(tabs_s + 5 * 64 + cb);
#pragma unroll
                for (int q = 0; q < NQ; ++q) vt[q] = -tis[q] * cvec[q];
              } else if (A.target.kind == SDEH_DENS_MULTI_WELL) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                  const float y = xe[q] - A.target.p1;
                  vt[q] = (cb + rrow(q) < A.target.n_comp ? -4.0f * (3.0f * y * y - A.target.p0) : -1.0f) * cvec[q];
                }
              } else if (OTD == 1 && A.target.kind == SDEH_DENS_FUNNEL) {
                // s_0 = -x0/var - (d-1)/2 + e^{-x0} sum x_j^2 / 2,  s_j = -x_j e^{-x0}   (coordinate 0 = register 0 of the h = 0 half);
                // the sums run over all coordinates: registers and the two lane halves of this wave
                const float x0 = bcast_lo(x[0][0]), c0 = bcast_lo(cvec[0]);
                float sq = 0.0f, cx = 0.0f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                  const bool first = q == 0 && h == 0;
                  sq = fmaf(first ? 0.0f : x[0][q], x[0][q], sq);
                  cx = fmaf(first ? 0.0f : cvec[q], x[0][q], cx);
                }
                sq = sum_xor32(sq);
                cx = sum_xor32(cx);
                const float iv = __expf(-x0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) vt[q] = iv * (c0 * x[0][q] - cvec[q]);
                if (h == 0) vt[0] = c0 * (-1.0f / A.target.p0 - 0.5f * iv * sq) + iv * cx;
              }
            }
            SDEH_FENCE();
#pragma unroll
            for (int q = 0; q < NQ; ++q) lam[ct][q] = KLB ? fmaf(jac_t, vt[q], fmaf(c_x, lam[ct][q], lin[q])) : fmaf(jac_t, vt[q], c_x * lam[ct][q]);
            SDEH_FENCE();
            if (jac_p != 0.0f || refc) {
              const f32x16 pis = rows16(tabs_s + 1 * 64 + cb);
              const float sig_r = refc ? sig : 0.0f;
#pragma unroll
              for (int q = 0; q < NQ; ++q) {
                const float v = fmaf(-jac_p * pis[q], cvec[q], lam[ct][q]);  // Gaussian prior: J = -1/sigma^2
                lam[ct][q] = fmaf(sig_r * pis[q], Gc[q], v);                // cost depends on x through sigma * prior.score(x)
              }
            }
          }
        }
        if (has_score && live_tile && A.g == 1) {  // summed over the tile's trajectories and all coordinates
          gsum = sum_wave(gsum);
          if (lane < 2) A.gpart[(tile * T + t) * A.gw + lane] = lane == 0 ? gsum : 0.0f;
        }
      }
      SDEH_FENCE();
      BW2_T(tp2);

      // ======================================================================================= backward + weight gradients
      // each product: publish (delta_k, a_k), barrier; then its matrix instructions -- operands a chunk ahead through a register ring
      // -- interleaved with the chain's next transposed layer; a barrier behind them frees the planes
      // Bridge: S_k of the divergence term (sdeh_bridgef.hip), plane [k][64][T B]; requested a stage ahead of its use.  Rows beyond the
      // batch shadow its last row and must contribute nothing.
      f32x16 sv[BR ? 2 : 1];
      auto load_s = [&](int k) {
        if constexpr (BR) {
          const float* __restrict__ base = A.s_in + (long long)k * 64 * ((long long)B * T);
          unsigned ns = Bu * (unsigned)T, off = (unsigned)(4 * h) * ns + (unsigned)t * Bu + (unsigned)lrow;
          asm volatile("" : "+v"(ns), "+v"(off));
#pragma unroll
          for (int R = 0; R < 2; ++R)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float v = base[off + (unsigned)(32 * R + rrow(q)) * ns];
              sv[R][q] = live ? v : 0.0f;
            }
        }
      };
      load_s(LH);
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) plane_put_n<NQ>(Dme, ct, j, h, dout[ct]);
      if constexpr (!VIO) ws_barrier();  // (VIO: the out layer's gradient is a product over this wave's own planes)
      f32x16 dl[2];
      if constexpr (OTD == 2) {
        float ds[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        stage_cols<8, RSW, 2, 2, 16>(Wout_s + 4 * h * RSW + j, dout, 8, dl, planes, wR, wC, j, h, dw_out, ds);
        bs_out += (ds[0] + ds[1]) + (ds[2] + ds[3]);
      } else {
        float ds[2] = {0.0f, 0.0f};
        const float* __restrict__ pp = planes + wR * 4 * PLANE;  // the pair's planes
        if constexpr (VIO) {
          // d out_layer.weight[i][c] += sum_k dout[i][k] a_{LH+1}[c][k]: rows = coordinates (D plane rows 0 .. 3), columns = channels
          block_product<false>(Dme, Ame, lane, dwo4, ds[0]);
          ds[0] = (lane & ~3) == 0 ? ds[0] : 0.0f;  // (every block forms the same row sums: lanes 0 .. 3 keep them -- d loss / d out bias)
#pragma unroll
          for (int q = 0; q < 16; ++q) { dl[0][q] = 0.0f; dl[1][q] = 0.0f; }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < d) {
              const float gb = bcast_lo(dout[0][i]);
              const f32x16 w0 = load16(vout_s + i * 64 + h * 16), w1 = load16(vout_s + i * 64 + 32 + h * 16);
#pragma unroll
              for (int q = 0; q < 16; ++q) { dl[0][q] = fmaf(w0[q], gb, dl[0][q]); dl[1][q] = fmaf(w1[q], gb, dl[1][q]); }
            }
          }
        } else {
          stage_cols<NGI, RSW, 2, 1, 8>(Wout_s + 4 * h * RSW + j, dout, NGI, dl, pp, 0, wC, j, h, dw_out, ds);
        }
        bs_out += ds[0] + ds[1];
      }
      BW2_T(tp3);
      f32x16 xpub[OTD];
#pragma unroll
      for (int l = LH - 1; l >= -1; --l) {
        BW2_T(tq0);
        // dl = d loss / d a_{l+2};  delta = dl . act'(Z_{l+1});  publish with a_{l+1} (l = -1: x)
        if constexpr (ZIN) {
          // kz = act'(Z_{l+1}) from the previous stage; a_{l+1} = act(Z_l) and the next kz from the record requested a stage ago
#pragma unroll
          for (int q = 0; q < 16; ++q) { dl[0][q] *= kz[0][q]; dl[1][q] *= kz[1][q]; }
          SDEH_FENCE();
          if (l >= 0) {
            f32x16 ak[2];
            if (((LH - l) & 1) == 0) {
              SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(zA[0], ak[0], kz[0]); SDEH_FENCE(); act_both<ACT>(zA[1], ak[1], kz[1]););
            } else {
              SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(zB[0], ak[0], kz[0]); SDEH_FENCE(); act_both<ACT>(zB[1], ak[1], kz[1]););
            }
            SDEH_FENCE();
            plane_put(Ame, 0, j, h, ak[0]);
            plane_put(Ame, 1, j, h, ak[1]);
            SDEH_FENCE();
            if (l >= 2) {  // (deeper networks: layer l - 2 takes the buffer just activated)
              if (((LH - l) & 1) == 0) load_z(t, tile, l >= 2 ? l - 2 : 0, zA);
              else load_z(t, tile, l >= 2 ? l - 2 : 0, zB);
            }
            if (l == 0 && !(OTD == 2 && BPTT)) {  // the next step's (or item's) top layer: zA is free (LH even: its Z_0 was just activated; odd: since the top)
              if (t > t_last) {
                load_z(t - 1, tile, LH, zA);
              } else if (round + 1 < n_rounds) {
                int tn = it_t, pn = it_pair;
                clamp_item(tn, pn);
                load_z(tn, tile_of(pn), LH, zA);
              }
            }
          }
        } else {
          if constexpr (BR) {  // adj(Z_k) = act'(Z_k) . d loss / d a_{k+1} + S_k
#pragma unroll
            for (int q = 0; q < 16; ++q) { dl[0][q] = fmaf(dl[0][q], keep[l + 1][0][q], sv[0][q]); dl[1][q] = fmaf(dl[1][q], keep[l + 1][1][q], sv[1][q]); }
            if (l >= 0) load_s(l);
          } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) { dl[0][q] *= keep[l + 1][0][q]; dl[1][q] *= keep[l + 1][1][q]; }
          }
          if (l >= 0) {
            plane_put(Ame, 0, j, h, akeep[l >= 0 ? l : 0][0]);
            plane_put(Ame, 1, j, h, akeep[l >= 0 ? l : 0][1]);
          }
        }
        if constexpr (XRELOAD) {
          if (l == 0) {  // requested in front of the last hidden stage's matrix instructions
#pragma unroll
            for (int ct = 0; ct < OTD; ++ct) xpub[ct] = load_cm(A.xs + (long long)t * d * B, (unsigned)lrow, ct);
          }
        }
        if (l < 0) {
#pragma unroll
          for (int ct = 0; ct < OTD; ++ct) plane_put_n<NQ>(Ame, ct, j, h, XRELOAD ? xpub[ct] : x[ct]);
        }
        SDEH_FENCE();
        plane_put(Dme, 0, j, h, dl[0]);
        plane_put(Dme, 1, j, h, dl[1]);
        SDEH_FENCE();
        BW2_T(tq1);
        if (!(VIO && l < 0)) ws_barrier();  // (VIO: the input layer's gradient is a product over this wave's own planes)
        BW2_T(tq2);
        BW2_ADD(13, tq0, tq1); BW2_ADD(14, tq1, tq2);
        if (l >= 0) {
          float ds[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          f32x16 dn[2];
          stage_cols<8, RSW, 2, 2, 16>(Whid_s + (l >= 0 ? l : 0) * 64 * RSW + 4 * h * RSW + j, dl, 8, dn, planes, wR, wC, j, h,
                                       dw_hid[l >= 0 ? l : 0], ds);
          bs_hid[l >= 0 ? l : 0] += (ds[0] + ds[1]) + (ds[2] + ds[3]);
          dl[0] = dn[0]; dl[1] = dn[1];
        } else {
          // the next step's (or item's) x, time embedding and scalars: requested in front of the step's last block of matrix instructions
          if (t > t_last) {
            load_x(t - 1, (int)tile, xnext);
            if constexpr (!ZIN) load_emb(t - 1, embnext);
            cnext = load_coef(t - 1);
            if constexpr (OTD == 2 && BPTT) load_z(t - 1, tile, LH, zA);
            else load_z(t - 1, tile, LH - 1, zB);
          } else if (round + 1 < n_rounds) {
            int tn = it_t, pn = it_pair;
            clamp_item(tn, pn);
            load_x(tn, tile_of(pn), xnext);
            if constexpr (!ZIN) load_emb(tn, embnext);
            cnext = load_coef(tn);
            if constexpr (OTD == 2 && BPTT) load_z(tn, tile_of(pn), LH, zA);
            else load_z(tn, tile_of(pn), LH - 1, zB);
          }
          f32x16 dx[OTD];
          // d loss / d (time embedding + input bias)[t][row] per tile: the delta row sums of each wave's trajectories
          if constexpr (OTD == 2) {
            float ds[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            stage_cols<(WDX ? 8 : 0), RSI, 2, 2, 16>(Win_s + 4 * h * RSI + j, dl, 8, dx, planes, wR, wC, j, h, dw_in, ds);
            if (live_item && wC == 0) {
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const float e = sum_xor32(ds[w]);
                const long long tl = 4 * (long long)cur_pair + w;
                if (h == 0 && tl < n_tiles) A.epart[(tl * T + t) * 64 + 32 * wR + j] = e;
              }
            }
          } else {
            float ds[2] = {0.0f, 0.0f};
            if constexpr (VIO) {
              // d input_embed.weight[c][i] += sum_k delta_0[c][k] x[i][k]: rows = channels (D plane), columns = coordinates (A plane rows
              // 0 .. 3); the row sums are d loss / d (time embedding + input bias)[t][channel] of this wave's tile -- one 256-byte store
              float esum = 0.0f;
              block_product<true>(Dme, Ame, lane, dwi4, esum);
              if (live_tile) A.epart[(tile * T + t) * 64 + lane] = esum;
            } else {
            stage_cols<(WDX ? NGI == 0 ? 0 : 8 : 0), RSI, 1, 2, 8>(Win_s + 4 * h * RSI + j, dl, 8, dx, planes + wR * 4 * PLANE, wC, 0, j, h, dw_in, ds);
            if (live_item) {
#pragma unroll
              for (int w = 0; w < 2; ++w) {
                const float e = sum_xor32(ds[w]);
                const long long tl = 4 * (long long)cur_pair + 2 * wR + w;
                if (h == 0 && tl < n_tiles) A.epart[(tl * T + t) * 64 + 32 * wC + j] = e;
              }
            }
            }
          }
          if constexpr (WDX) {
            if constexpr (VIO) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (i < d) {
                  const f32x16 w0 = load16(vin_s + i * 64 + h * 16), w1 = load16(vin_s + i * 64 + 32 + h * 16);
                  float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                  for (int q = 0; q < 16; ++q) { p0 = fmaf(w0[q], dl[0][q], p0); p1 = fmaf(w1[q], dl[1][q], p1); }
                  const float tot = sum_xor32(p0 + p1);
                  lam[0][i] += h == 0 ? tot : 0.0f;
                }
              }
            } else {
#pragma unroll
              for (int ct = 0; ct < OTD; ++ct)
#pragma unroll
                for (int q = 0; q < NQ; ++q) lam[ct][q] += dx[ct][q];
            }
          }
          if constexpr (BR == 2) {  // coordinate-major [T][d][B], as load_cm reads
            if (live) {
#pragma unroll
              for (int ct = 0; ct < OTD; ++ct) {
                const int cb = 32 * ct + 4 * h;
                unsigned off = ((unsigned)(cb < d ? cb : 0) * Bu + (unsigned)lrow) * 4u, s1 = Bu * 4u;
                asm volatile("" : "+v"(off), "+v"(s1));
                char* __restrict__ pb = reinterpret_cast<char*>(A.dx_out + (long long)t * d * B);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                  if (cb + rrow(q) < d) *reinterpret_cast<float*>(pb + off + (unsigned)rrow(q) * s1) = lam[ct][q];
              }
            }
          }
        }
      }
      BW2_T(tp4);
      BW2_ADD(0, tp0, tp1); BW2_ADD(1, tp1, tp2); BW2_ADD(2, tp2, tp3); BW2_ADD(3, tp3, tp4); BW2_ADD(7, tp0, tp4); BW2_ADD(8, 0ull, 1ull);
    }
  }

  // ---- the team's partial gradients (the record of sdeh_bwdf.hip) -----------------------------------------------------------
  float* __restrict__ rec = A.wpart + (long long)team_g * A.wsize;
#pragma unroll
  for (int l = 0; l < LH; ++l) {
    store_tile(rec + off_whid<OTD>() + l * 4096, 64, wR, wC, j, h, dw_hid[l]);
    float b = bs_hid[l];
    b += __shfl_xor(b, 32);
    if (h == 0 && wC == 0) rec[off_bhid<OTD, LH>() + l * 64 + 32 * wR + j] = b;
  }
  if constexpr (OTD == 2) {
    store_tile(rec, DPP, wR, wC, j, h, dw_in);
    store_tile(rec + off_wout<OTD, LH>(), 64, wR, wC, j, h, dw_out);
    float b = bs_out;
    b += __shfl_xor(b, 32);
    if (h == 0 && wC == 0) rec[off_bout<OTD, LH>() + 32 * wR + j] = b;
  } else if constexpr (VIO) {
    // the four waves' block sums meet in the (now idle) planes: sIn[wave][channel][4 coordinates], sOut[wave][4 coordinates][channel],
    // the out bias sums behind them; the team writes whole [64][32] / [32][64] record tiles (coordinates >= 4: zeros)
    ws_barrier();
    float* __restrict__ sIn = planes;               // [4][64][4]
    float* __restrict__ sOut = planes + 4 * 256;    // [4][4][64]
    float* __restrict__ sB = planes + 8 * 256;      // [4][4]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // dwi4: register e of lane (b, jj) = gradient of input_embed.weight[4 b + e][jj];  dwo4: register e = out_layer.weight[e][4 b + jj]
      sIn[wave * 256 + (4 * (lane >> 2) + e) * 4 + (lane & 3)] = dwi4[0][e] + dwi4[1][e];
      sOut[wave * 256 + e * 64 + lane] = dwo4[0][e] + dwo4[1][e];
    }
    if (lane < 4) sB[wave * 4 + lane] = bs_out;
    ws_barrier();
    for (int idx = tid; idx < 64 * DPP; idx += 256) {
      const int c = idx / DPP, i = idx - c * DPP;
      rec[idx] = i < 4 ? ((sIn[c * 4 + i] + sIn[256 + c * 4 + i]) + (sIn[512 + c * 4 + i] + sIn[768 + c * 4 + i])) : 0.0f;
    }
    for (int idx = tid; idx < DPP * 64; idx += 256) {
      const int i = idx >> 6, c = idx & 63;
      rec[off_wout<OTD, LH>() + idx] = i < 4 ? ((sOut[i * 64 + c] + sOut[256 + i * 64 + c]) + (sOut[512 + i * 64 + c] + sOut[768 + i * 64 + c])) : 0.0f;
    }
    if (tid < DPP) rec[off_bout<OTD, LH>() + tid] = tid < 4 ? ((sB[tid] + sB[4 + tid]) + (sB[8 + tid] + sB[12 + tid])) : 0.0f;
  } else {
    // input_embed row tile wC / out_layer channel tile wC: the second wave pair hands its sums to the first through the (now idle) planes
    ws_barrier();
    float* __restrict__ xch = planes + wC * 2 * PLANE;  // 2 x 16 x 64 + 64 floats per tile index: inside two planes
    if (wR == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { xch[q * 64 + lane] = dw_in[q]; xch[1024 + q * 64 + lane] = dw_out[q]; }
      xch[2048 + lane] = bs_out;
    }
    ws_barrier();
    if (wR == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { dw_in[q] += xch[q * 64 + lane]; dw_out[q] += xch[1024 + q * 64 + lane]; }
      float b = bs_out + xch[2048 + lane];
      b += __shfl_xor(b, 32);
      store_tile(rec, DPP, wC, 0, j, h, dw_in);
      store_tile(rec + off_wout<OTD, LH>(), 64, 0, wC, j, h, dw_out);
      if (h == 0 && wC == 0) rec[off_bout<OTD, LH>() + j] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Back-propagation through time as a scan (d <= 4; VERDICT r03 next-step 1, the small-batch half).  Through time the adjoint obeys
//     G_t = c_u lambda_{t+1} + gc_t,      lambda_t = c_x lambda_{t+1} + J_t^T (m_t . G_t) + (score / cost terms linear in G_t, gc_t)
// with J_t = d nn / d x at (t, trajectory) and m_t the clamp's mask: the only thing that is sequential is a recursion on d numbers per
// trajectory.  The fused kernels walk it with the whole network in the loop -- at the reference's training batches (512 .. 2048) that is
// one dependent chain of ~6 us per step on a few dozen wavefronts.  Here the network work is ROW-PARALLEL twice (bwdf2_kernel<.., JAC>:
// nn and the d x d Jacobian of every (step, trajectory); afterwards the lv form of the backward with the upstream gradient given), and
// this kernel does the recursion: lane = trajectory, everything of a step in ~60 vector instructions, the next step's operands in flight.
// The elementwise semantics (what is constant, what is differentiated: losses/oc.py:204-225, 319-337, 418-450; models/reparam.py) are
// those of bwdf2_kernel's elementwise phase, statement for statement.
// ---------------------------------------------------------------------------------------------------------------------------
template <int D, bool KLB = false>  // the dimension at compile time: exact loops, no predicated loads (a branch per load breaks the prefetch ring)
__global__ __launch_bounds__(64) void bwdf2_scan_kernel(const BwdfArgs A) {
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  constexpr int d = D;
  const int T = A.n_steps;
  const long long B = A.batch;
  const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
  const bool live = row < B;
  const long long lrow = live ? row : B - 1;
  const float wi = live ? A.grad_rnd[lrow] : 0.0f;
  const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);
  const int ctrl_kind = A.ctrl_kind, flags = A.flags;
  const bool has_score = ctrl_kind != SDEH_CTRL_CLIPPED;
  const bool refc = (flags & SDEH_FLAG_REFERENCE_CTRL) && A.loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool expo = A.loss_kind == SDEH_LOSS_EXPONENTIAL;
  const bool ito = (flags & SDEH_FLAG_ITO) != 0;
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
  // Gaussian tables: (mean, inverse variance) of prior [1], second [2], target [0]
  float pmu[D], pis[D], tis[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const bool in = i < d && i < L.dp;
    pmu[i] = in ? ws[L.dg[1] + 2 * i] : 0.0f;
    pis[i] = in ? ws[L.dg[1] + 2 * i + 1] : 0.0f;
    tis[i] = in ? ws[L.dg[0] + 2 * i + 1] : 0.0f;
  }
  float lam[D];
#pragma unroll
  for (int i = 0; i < D; ++i) lam[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < D; ++i) {  // lambda_T = w_i d(terminal costs)/dx_T
    if (i < d) {
      if (flags & SDEH_FLAG_TERMINAL_SECOND) {
        const float smu = ws[L.dg[2] + 2 * i], sis = ws[L.dg[2] + 2 * i + 1];
        lam[i] = wi * (smu - A.xs[((long long)T * d + i) * B + lrow]) * sis;
      }
      if ((flags & SDEH_FLAG_TERMINAL_TARGET) && A.tscore != nullptr) lam[i] = fmaf(-wi, A.tscore[(long long)i * B + lrow], lam[i]);
    }
  }
  // (the per-step scalars travel with the row as VECTOR loads through an opaque zero offset: scalar loads return out of order, a
  // wait for one is a wait for all of them -- requested at the top of a step they cost their full latency, 1.7 us per step)
  struct Row { float x[D], sc[D], nn[D], J[D * D], cf[8], gam[D], cin[D], lin[D]; };
  int vz = 0;
  asm volatile("" : "+v"(vz));
  const float* __restrict__ scp = has_score ? A.sc : A.xs;
  auto load_row = [&](int t) {
    Row r;
    {
      const float* __restrict__ cp = ws + L.coef + t * kCoefStride + vz;
      r.cf[0] = cp[CF_SIGMA]; r.cf[1] = cp[CF_W]; r.cf[2] = cp[CF_SBK]; r.cf[3] = cp[CF_SQDT];
      r.cf[4] = cp[CF_B2S2]; r.cf[5] = cp[CF_DT]; r.cf[6] = cp[CF_ALPHAK]; r.cf[7] = cp[CF_DRIFT];
      const float* __restrict__ gp = ws + L.gam + t * L.g + vz;
#pragma unroll
      for (int i = 0; i < D; ++i) r.gam[i] = has_score ? gp[A.g == 1 ? 0 : min(i, L.g - 1)] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const bool in = i < d;
      const long long o = ((long long)t * d + (in ? i : 0)) * B + lrow;
      r.x[i] = in ? A.xs[o] : 0.0f;
      const float scl = scp[o];  // (unconditional: a valid plane stands in when there is no score term)
      r.sc[i] = has_score ? scl : 0.0f;
      r.nn[i] = in ? A.nn_out[o] : 0.0f;
      if constexpr (KLB) {  // Bridge, method kl: u + v in the running cost; d loss / d x_t of the inference terms
        r.cin[i] = in ? A.cost_in[o] : 0.0f;
        r.lin[i] = in ? A.lam_in[o] : 0.0f;
      } else {
        r.cin[i] = 0.0f; r.lin[i] = 0.0f;
      }
#pragma unroll
      for (int k = 0; k < D; ++k) r.J[D * k + i] = in && k < d ? A.jac_out[(((long long)t * d + k) * d + i) * B + lrow] : 0.0f;
    }
    return r;
  };
  // the operands of a step do not depend on the recursion: four steps of them in flight (a step is ~60 instructions, a load ~1.5 us)
  constexpr int DEPTH = 4;
  Row ring[DEPTH];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) ring[u] = load_row(T - 1 - u >= 0 ? T - 1 - u : 0);
  for (int t0 = T - 1; t0 >= 0; t0 -= DEPTH) {
#pragma unroll
   for (int u = 0; u < DEPTH; ++u) {
    const int t = t0 - u;
    if (t < 0) break;
    const Row r = ring[u];
    ring[u] = load_row(t - DEPTH >= 0 ? t - DEPTH : 0);
    const float sig = r.cf[0], wl = r.cf[1];
    const float c_i = expo ? r.cf[2] : r.cf[3];
    const float cdt = expo ? r.cf[4] : r.cf[5];
    const float c_u = expo ? r.cf[4] : sig * r.cf[5];
    const float c_x = expo ? r.cf[6] : fmaf(r.cf[7], r.cf[5], 1.0f);
    const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
    const float coef_t = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET ? wl : 0.0f);
    const float coef_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR ? 1.0f - wl : 0.0f;
    const float jac_t = (!has_score || (flags & (SDEH_FLAG_DETACH_SCORE | SDEH_FLAG_TARGET_SCORE_CONST))) ? 0.0f : coef_t;
    const float jac_p = (!has_score || (flags & SDEH_FLAG_DETACH_SCORE)) ? 0.0f : coef_p;
    float xi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) xi[i] = 0.0f;
    if (ito) {
      if (A.noise != nullptr) {
#pragma unroll
        for (int i = 0; i < D; ++i) xi[i] = i < d ? A.noise[((long long)t * B + lrow) * d + i] : 0.0f;
      } else {
        float n4[4];
        box_muller4(philox_block(A.seed, rng_off, grow, t, 0), n4);
#pragma unroll
        for (int i = 0; i < D; ++i) xi[i] = i < d ? n4[i] : 0.0f;
      }
    }
    const float c_ie = ito ? c_i : 0.0f;
    float dout[D], cvec[D], Gc[D], gqv[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float mfac = 0.0f;
      if (has_score) mfac = mult * r.gam[i];
      const float csc = clipf(r.sc[i], A.clip_score);
      const float keep_s = fabsf(r.sc[i]) <= A.clip_score ? 1.0f : 0.0f;
      const float rr = refc ? sig * (pmu[i] - r.x[i]) * pis[i] : 0.0f;  // reference control sigma * prior.score(x)
      const float u = clipf(r.nn[i], A.clip_model) + mfac * csc;
      const float gc = i < d ? wi * fmaf(KLB ? r.cin[i] : u - rr, cdt, c_ie * xi[i]) : 0.0f;
      const float gq = fmaf(c_u, lam[i], gc);
      Gc[i] = gc;
      cvec[i] = keep_s * mfac * gq;
      dout[i] = fabsf(r.nn[i]) <= A.clip_model ? gq : 0.0f;
      gqv[i] = gq;
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < D; ++i) A.gq_out[((long long)t * d + i) * B + lrow] = gqv[i];
    }
    float vt[D];
#pragma unroll
    for (int i = 0; i < D; ++i) vt[i] = 0.0f;
    if (jac_t != 0.0f) {  // closed-form target scores are differentiated through x
      if (A.target.kind == SDEH_DENS_DIAG_GAUSS) {
#pragma unroll
        for (int i = 0; i < D; ++i) vt[i] = -tis[i] * cvec[i];
      } else if (A.target.kind == SDEH_DENS_MULTI_WELL) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const float y = r.x[i] - A.target.p1;
          vt[i] = (i < A.target.n_comp ? -4.0f * (3.0f * y * y - A.target.p0) : -1.0f) * cvec[i];
        }
      } else if (A.target.kind == SDEH_DENS_FUNNEL) {
        float sq = 0.0f, cx = 0.0f;
#pragma unroll
        for (int i = 1; i < D; ++i) { sq = fmaf(r.x[i], r.x[i], sq); cx = fmaf(cvec[i], r.x[i], cx); }
        const float iv = __expf(-r.x[0]), c0 = cvec[0];
#pragma unroll
        for (int i = 1; i < D; ++i) vt[i] = iv * (c0 * r.x[i] - cvec[i]);
        vt[0] = c0 * (-1.0f / A.target.p0 - 0.5f * iv * sq) + iv * cx;
      }
    }
    const float sig_r = refc ? sig : 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float v = fmaf(jac_t, vt[i], c_x * lam[i]);
      if (jac_p != 0.0f || refc) {
        v = fmaf(-jac_p * pis[i], cvec[i], v);  // Gaussian prior: J = -1/sigma^2
        v = fmaf(sig_r * pis[i], Gc[i], v);     // cost depends on x through sigma * prior.score(x)
      }
      float dx = 0.0f;  // W_in^T delta_0 = J^T (clamp mask . G)
#pragma unroll
      for (int k = 0; k < D; ++k) dx = fmaf(r.J[D * k + i], dout[k], dx);
      lam[i] = i < d ? (KLB ? v + dx + r.lin[i] : v + dx) : 0.0f;
    }
   }
  }
}

#ifdef SDEH_BWDF2_SINGLE  // (compile-only aid: ONE instantiation, e.g. '-DSDEH_BWDF2_SINGLE=2, true, 2, true, 16, false' -- register / scratch experiments)
template __global__ void bwdf2_kernel<SDEH_BWDF2_SINGLE>(const BwdfArgs);
#else
int launch_bwdf2_scan(const BwdfArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)((a.batch + 63) / 64));
  if ((a.cost_in != nullptr) != (a.lam_in != nullptr)) return SDEH_ERR_UNSUPPORTED;
  if (a.cost_in != nullptr) {
    switch (a.d) {
      case 1: hipLaunchKernelGGL((bwdf2_scan_kernel<1, true>), grid, dim3(64), 0, stream, a); break;
      case 2: hipLaunchKernelGGL((bwdf2_scan_kernel<2, true>), grid, dim3(64), 0, stream, a); break;
      case 3: hipLaunchKernelGGL((bwdf2_scan_kernel<3, true>), grid, dim3(64), 0, stream, a); break;
      case 4: hipLaunchKernelGGL((bwdf2_scan_kernel<4, true>), grid, dim3(64), 0, stream, a); break;
      default: return SDEH_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
  }
  switch (a.d) {
    case 1: hipLaunchKernelGGL(bwdf2_scan_kernel<1>, grid, dim3(64), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(bwdf2_scan_kernel<2>, grid, dim3(64), 0, stream, a); break;
    case 3: hipLaunchKernelGGL(bwdf2_scan_kernel<3>, grid, dim3(64), 0, stream, a); break;
    case 4: hipLaunchKernelGGL(bwdf2_scan_kernel<4>, grid, dim3(64), 0, stream, a); break;
    default: return SDEH_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

bool bwdf2_scan_fits(int d, int n_hidden) { return d >= 1 && d <= 4 && n_hidden == 2 && bwdf_fits(d, n_hidden); }

template <int LH>
static int launch_bwdf2_jac_t(const BwdfArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)(bwdf::lds_floats<1, LH>() + 512) * sizeof(float);
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf2_kernel<1, false, LH, false, 4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf2_kernel<1, false, LH, false, 4, true, true>), dim3((unsigned)a.n_slots), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

template <int LH>
static int launch_bwdf2_jacz_t(const BwdfArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)(bwdf::lds_floats<1, LH>() + 512) * sizeof(float);
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf2_kernel<1, false, LH, true, 4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf2_kernel<1, false, LH, true, 4, true, true>), dim3((unsigned)a.n_slots), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

int launch_bwdf2_jac(const BwdfArgs& a, hipStream_t stream) {
  if (!bwdf2_scan_fits(a.d, a.n_hidden) || a.nn_out == nullptr || a.jac_out == nullptr) return SDEH_ERR_UNSUPPORTED;
  if (a.zrec != nullptr) return launch_bwdf2_jacz_t<2>(a, stream);  // (the record instead of the re-evaluation)
  return launch_bwdf2_jac_t<2>(a, stream);
}

// the launch that reads the pre-activation record instead of re-evaluating the network (ZIN)
template <int OTD, bool BPTT, int LH, int NQ, bool VIO>
static int launch_bwdf2_z(const BwdfArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)(bwdf::lds_floats<OTD, LH>() + 512) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf2_kernel<OTD, BPTT, LH, true, NQ, VIO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf2_kernel<OTD, BPTT, LH, true, NQ, VIO>), dim3((unsigned)a.n_slots), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

template <int OTD, bool BPTT, int LH, int NQ, bool VIO, bool KLB = false>
static int launch_bwdf2_q(const BwdfArgs& a, hipStream_t stream) {
  if constexpr (BPTT && LH == 2 && !KLB) {
    if (a.cost_in != nullptr && a.lam_in != nullptr) return launch_bwdf2_q<OTD, BPTT, LH, NQ, VIO, true>(a, stream);
  }
  if (!KLB && BPTT && (a.cost_in != nullptr || a.lam_in != nullptr)) return SDEH_ERR_UNSUPPORTED;  // (two hidden layers, both planes)
  if constexpr (!KLB) {
    if (a.zrec != nullptr) return launch_bwdf2_z<OTD, BPTT, LH, NQ, VIO>(a, stream);
  }
  constexpr bool RECOMP = false;  // (the slot of the ZIN parameter: this launch re-evaluates the network)
  const size_t lds_bytes = (size_t)(bwdf::lds_floats<OTD, LH>() + 512) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf2_kernel<OTD, BPTT, LH, RECOMP, NQ, VIO, false, 0, KLB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf2_kernel<OTD, BPTT, LH, RECOMP, NQ, VIO, false, 0, KLB>), dim3((unsigned)a.n_slots), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

template <int OTD, bool BPTT, int LH>
static int launch_bwdf2_t(const BwdfArgs& a, hipStream_t stream) {
  if constexpr (OTD == 2) {
    return launch_bwdf2_q<2, BPTT, LH, 16, false>(a, stream);
  } else if constexpr (LH == 2) {  // the shipped depth: specialised by the number of live coordinates
    const bool no_vio = plan_opt(OPT_BWD_NO_VIO) != nullptr;  // A/B aid (plan option)
    if (a.d <= 4 && !no_vio) return launch_bwdf2_q<1, BPTT, LH, 4, true>(a, stream);
    if (a.d <= 8) return launch_bwdf2_q<1, BPTT, LH, 4, false>(a, stream);
    if (a.d <= 16) return launch_bwdf2_q<1, BPTT, LH, 8, false>(a, stream);
    return launch_bwdf2_q<1, BPTT, LH, 16, false>(a, stream);
  } else {
    return launch_bwdf2_q<1, BPTT, LH, 16, false>(a, stream);
  }
}

template <int OTD, int NQ, bool VIO, int BR>
static int launch_bwdf2_br(const BwdfArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)(bwdf::lds_floats<OTD, 2>() + 512) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf2_kernel<OTD, false, 2, false, NQ, VIO, false, BR>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf2_kernel<OTD, false, 2, false, NQ, VIO, false, BR>), dim3((unsigned)a.n_slots), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// the inference network of a Bridge (row-parallel; two hidden layers): first-order terms + the base chain of the divergence term
int launch_bwdf2_bridge(const BwdfArgs& a, hipStream_t stream) {
  if (a.n_hidden != 2 || a.d > 64 || a.gextra == nullptr || a.s_in == nullptr || !(a.flags & SDEH_FLAG_CHANGE_SDE_CTRL)) return SDEH_ERR_UNSUPPORTED;
  if (a.dx_out != nullptr) {  // method kl: + d loss / d x_t of the inference network's terms
    if (a.d <= 4) return launch_bwdf2_br<1, 4, true, 2>(a, stream);
    if (a.d <= 8) return launch_bwdf2_br<1, 4, false, 2>(a, stream);
    if (a.d <= 16) return launch_bwdf2_br<1, 8, false, 2>(a, stream);
    if (a.d <= 32) return launch_bwdf2_br<1, 16, false, 2>(a, stream);
    return launch_bwdf2_br<2, 16, false, 2>(a, stream);
  }
  if (a.d <= 4) return launch_bwdf2_br<1, 4, true, 1>(a, stream);
  if (a.d <= 8) return launch_bwdf2_br<1, 4, false, 1>(a, stream);
  if (a.d <= 16) return launch_bwdf2_br<1, 8, false, 1>(a, stream);
  if (a.d <= 32) return launch_bwdf2_br<1, 16, false, 1>(a, stream);
  return launch_bwdf2_br<2, 16, false, 1>(a, stream);
}

bool bwdf2_fits(int d, int n_hidden) { return n_hidden >= 1 && n_hidden <= 2 && bwdf_fits(d, n_hidden); }  // (+ launch_bwdf2's own refusals)

// teams (of four 32-trajectory tiles: one workgroup, one per CU) the launch uses
int bwdf2_slots(long long batch, int n_steps, bool bptt) {
  const long long tiles = (batch + 31) / 32, quads = (tiles + 3) / 4;
  const long long items = bptt ? quads : quads * n_steps;
  return (int)(items < 256 ? items : 256);
}

#ifdef SDEH_BWDF_PROFILE
static void bwdf2_prof_dump(hipStream_t stream) {
  (void)hipStreamSynchronize(stream);
  unsigned long long v[16];
  (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(bwdf2_prof), sizeof(v));
  const double n = v[8] ? (double)v[8] : 1.0;
  fprintf(stderr, "bwdf2 phases (cycles per step of block 0 / wave 0, %llu steps): forward %.0f | elementwise %.0f | out stage %.0f | "
          "hidden + in stages %.0f | step %.0f || forward: in mm %.0f, in act %.0f, hidden %.0f, publish + out mm %.0f || stages: act' + publish %.0f, "
          "barrier %.0f\n", v[8], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[7] / n, v[9] / n, v[10] / n, v[11] / n, v[12] / n, v[13] / n, v[14] / n);
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(bwdf2_prof), z, sizeof(z));
}
#endif

int launch_bwdf2(const BwdfArgs& a, hipStream_t stream) {
#ifdef SDEH_BWDF_PROFILE
  struct Dump { hipStream_t s; ~Dump() { bwdf2_prof_dump(s); } } dump{stream};
#endif
  const bool bptt = !(a.flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  if (a.d > 32 && bptt && a.target.kind == SDEH_DENS_FUNNEL) return SDEH_ERR_UNSUPPORTED;  // its Jacobian couples the two coordinate tiles
  if (a.n_hidden == 1) {
    if (a.d <= 32) return bptt ? launch_bwdf2_t<1, true, 1>(a, stream) : launch_bwdf2_t<1, false, 1>(a, stream);
    return bptt ? launch_bwdf2_t<2, true, 1>(a, stream) : launch_bwdf2_t<2, false, 1>(a, stream);
  }
  if (a.n_hidden == 2) {
    if (a.d <= 32) return bptt ? launch_bwdf2_t<1, true, 2>(a, stream) : launch_bwdf2_t<1, false, 2>(a, stream);
    return bptt ? launch_bwdf2_t<2, true, 2>(a, stream) : launch_bwdf2_t<2, false, 2>(a, stream);
  }
  return SDEH_ERR_UNSUPPORTED;
}

#endif  // SDEH_BWDF2_SINGLE

}  // namespace sdeh
