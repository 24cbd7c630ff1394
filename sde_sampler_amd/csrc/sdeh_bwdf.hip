// Fused training backward of the control network: back-propagation through the
// FourierMLP at the stored trajectory AND the weight-gradient contractions in one kernel -- no [C, T*B] planes in HBM.
//
// What the reference does here: loss.backward() through the unrolled Python loop of losses/oc.py:176-222 / 301-334 / 416-446
// (solver/base.py:407), i.e. autograd through T copies of models/mlp.py:114-122 and models/reparam.py:56-83,131-197.  The older
// path of this library (sdeh_bwd.hpp + sdeh_wgrad.hip) writes the pre-activations Z_k and their adjoints as coordinate-major planes
// (10 GB at B = 65 536, T = 100) and contracts them in a second pass; this kernel keeps everything on chip:
//
//   * a TEAM of two wavefronts owns 32 trajectories (one MFMA column tile); wave r computes row tile r (channels 32r .. 32r+31) of
//     every layer, forward (re-evaluation at x_t) and backward (transposed weights).  A layer's input is both tiles, so the waves
//     exchange their halves through LDS planes [row][trajectory] -- one workgroup barrier per layer; the MFMA B operands are read
//     from the planes k-group by k-group (lane (j, h): rows 8 s + 4 h .. + 3 of column j).
//   * the SAME planes are the operands of the weight gradients: dW_k += delta_k a_k^T contracts over the trajectories, i.e. both
//     MFMA operands are the planes read TRANSPOSED (lane = row, 4 consecutive trajectories per ds_read_b128; row stride 36 floats
//     keeps the reads conflict-free).  The [64, 64] accumulators stay in registers for the whole launch (8 tiles per wave) and are
//     written once per team; a deterministic two-pass sum over the teams finishes them (launch_partial_sums).
//   * one natural-layout copy of each weight matrix sits in LDS (row stride 68 floats) and serves both directions: the forward
//     pass reads rows (lane = output row, ds_read_b128 along k), the backward pass reads columns (lane = output column, four
//     ds_read_b32 down k) -- half the LDS of the packed forward + transposed images of the older kernels.
//   * all per-coordinate work (upstream gradient of the control, clip masks, score-term Jacobians, the adjoint lambda_t) runs in
//     the accumulator layout (lane (j, h), register q  <->  coordinate 32 tile + (q & 3) + 8 (q >> 2) + 4 h of trajectory j), wave r
//     owning coordinate tile r (d <= 32: both waves mirror tile 0) -- all 64 lanes work, no T-layout shuffles.
//   * what only the forward launch can know cheaply comes from it (sdeh_simulate_fwd_train2): the combined score entering the control
//     sc (before clip and gamma; for mixture targets this is the expensive part) and d(terminal target cost)/dx_T.  These planes and
//     the trajectory itself are coordinate-major ([T+1][d][B]): the forward kernel (lane = trajectory) stores and this kernel
//     (lane = trajectory, register = coordinate) loads whole 128-byte lines.
//
// Modes: BPTT (method kl / kl_ito: a team walks its 32 trajectories backwards through time carrying lambda_t) and row-parallel
// (lv / lv_traj: the trajectory is a constant of the graph, (step, tile) items are independent).  The semantics (what is constant,
// what is differentiated) are those of sdeh_bwd.hpp, which stays the path for Bridges and for networks this kernel is not compiled
// for (more than three hidden layers).
#include "sdeh_bwdf.hpp"
#ifdef SDEH_BWDF_PROFILE
#include <cstdio>
#endif

namespace sdeh {

// -DSDEH_BWDF_PROFILE: per-phase cycle counts of one wave (block 0, wave 0) in bwdf_prof[] -- a measurement build, never shipped
// (tools/bwdf_phase_profile.sh); phases: 0 forward in-layer, 1 hidden layers, 2 out layer, 3 elementwise, 4 dW_out + adjoint,
// 5 hidden stages, 6 in-layer stage (dW_in, dx, lambda), 7 whole step, 8 steps counted
#ifdef SDEH_BWDF_PROFILE
__device__ unsigned long long bwdf_prof[16];
#define BWDF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define BWDF_DECL(var) unsigned long long var = __builtin_readcyclecounter()
#define BWDF_SET(var) var = __builtin_readcyclecounter()
#define BWDF_ADD(k, t0, t1) do { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&bwdf_prof[k], (t1) - (t0)); } while (0)
#else
#define BWDF_T(var) do {} while (0)
#define BWDF_DECL(var) do {} while (0)
#define BWDF_SET(var) do {} while (0)
#define BWDF_ADD(k, t0, t1) do {} while (0)
#endif



// KLB (through time, a Bridge's generative network with method kl): the running cost on the plane cost_in = u + v, lam_in added to the adjoint
// ZIN (round 5): no re-evaluation -- the pre-activations come from the record the training forward kept (sdeh_traj_ws.hpp: ZRec; wave r
// loads row tile r of a layer with four 16-byte loads per lane, in its accumulator layout), the raw network output from its plane.  A
// layer's record is requested a stage ahead of the stage that activates it (two 16-register buffers); the three layer exchanges of the
// forward pass and their barriers are gone (one barrier at the top of a step keeps the planes' reuse ordered).
template <int OTD, bool BPTT, int LH, bool KLB = false, bool ZIN = false>
__global__ __launch_bounds__(256) void bwdf_kernel(const BwdfArgs A) {
  using namespace bwdf;
  constexpr int RSI = rsi<OTD>(), DPP = 32 * OTD;
  constexpr int NDW = OTD + 2 * LH + (OTD == 2 ? 2 : 1);  // weight-gradient tiles of a wave: input_embed, hidden, out_layer
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ Win = lds;
  float* __restrict__ Whid = Win + 64 * RSI;
  float* __restrict__ Wout = Whid + LH * 64 * RSW;
  float* __restrict__ bh = Wout + DPP * RSW;
  float* __restrict__ bo = bh + LH * 64;
  float* __restrict__ tabs = bo + 64;
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 1, r = wave & 1;
  const int j = lane & 31, h = lane >> 5;
  const int ct = OTD == 2 ? r : 0;  // coordinate tile this wave owns (d <= 32: both waves mirror tile 0)
  float* __restrict__ pl = tabs + TABS + team * 4 * PLANE;  // planes: A[0], A[1], D[0], D[1]
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
  // planes start zeroed: rows / trajectories that are never written must not hold NaNs (they meet zero weights)
  for (int idx = tid; idx < 2 * 4 * PLANE; idx += 256) tabs[TABS + idx] = 0.0f;
  __syncthreads();

  const int act = A.act, ctrl_kind = A.ctrl_kind, flags = A.flags;
  const bool has_score = ctrl_kind != SDEH_CTRL_CLIPPED;
  const bool refc = (flags & SDEH_FLAG_REFERENCE_CTRL) && A.loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool expo = A.loss_kind == SDEH_LOSS_EXPONENTIAL;
  const bool ito = (flags & SDEH_FLAG_ITO) != 0;
  const int cb = 32 * ct + 4 * h;  // first coordinate of this lane's registers: coordinate(q) = cb + rrow(q)
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);

  // Coordinates >= d of a tile need no masks: their x / sc / xi are loaded or drawn as zeros, the weight copies and Gaussian tables
  // are zero-padded, so every quantity derived from them stays exactly zero.

  // weight-gradient accumulators: [0, OTD) input_embed (row tile r x coordinate tiles); then hidden layer l: row tile r x 2;
  // then out_layer: OTD == 2: coordinate tile r x 2 channel tiles; OTD == 1: coordinate tile 0 x channel tile r
  f32x16 dw[NDW];
#pragma unroll
  for (int k = 0; k < NDW; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) dw[k][q] = 0.0f;
  float bs_hid[LH], bs_out = 0.0f;
#pragma unroll
  for (int l = 0; l < LH; ++l) bs_hid[l] = 0.0f;

  const int n_teams = (int)gridDim.x * 2, team_g = (int)blockIdx.x * 2 + team;
  const long long n_items = BPTT ? (long long)A.n_tiles : (long long)A.n_tiles * T;
  const long long n_rounds = (n_items + n_teams - 1) / n_teams;
  int par = 0;

  // Items: tiles (through time) or (step, tile) pairs, step-major (row-parallel); a team takes item team_g, team_g + n_teams, ...
  // The (step, tile) of an item is carried along incrementally: 64-bit divisions per item cost more than the item's bookkeeping.
  const int n_tiles = A.n_tiles;
  const int d_t = BPTT ? 0 : n_teams / n_tiles, d_tile = BPTT ? n_teams : n_teams % n_tiles;
  int it_t = BPTT ? T - 1 : team_g / n_tiles, it_tile = BPTT ? team_g : team_g % n_tiles;  // the current item
  auto advance = [&](int& t_io, int& tile_io) {  // -> the team's next item
    tile_io += d_tile;
    if constexpr (!BPTT) {
      t_io += d_t;
      if (tile_io >= n_tiles) { tile_io -= n_tiles; t_io += 1; }
    }
  };
  auto item_live = [&](int t_i, int tile_i) { return BPTT ? tile_i < n_tiles : t_i < T; };
  // x of the step that comes next -- also across items: in the row-parallel mode an item is a single step
  // 16 coordinates (cb + rrow(q)) of column `col` of a coordinate-major plane [d][B] whose start is WAVE-UNIFORM: one scalar base +
  // a 32-bit byte offset per element (global_load saddr form), the lane's part of it made opaque per call.  Written with 64-bit
  // element addresses (plane[(cb + rrow(q)) * B + col]) hipcc keeps one loop-invariant address PAIR per element and plane alive
  // across the step loop -- they spill, and come back as ~35 dependent scratch reloads per step, each behind its own vmcnt(0).
  // Coordinates >= d read a valid element (clamped) and are zeroed by a select: no branches.
  const unsigned Bu = (unsigned)B;
  auto load_cm16 = [&](const float* __restrict__ plane_u, unsigned col) {
    f32x16 v;
    unsigned lane_off = ((unsigned)(cb < d ? cb : 0) * Bu + col) * 4u;
    asm volatile("" : "+v"(lane_off));
    const char* __restrict__ pb = reinterpret_cast<const char*>(plane_u);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool ok = cb + rrow(q) < d;
      const unsigned off = ok ? lane_off + (unsigned)rrow(q) * Bu * 4u : lane_off;
      const float val = *reinterpret_cast<const float*>(pb + off);
      v[q] = ok ? val : 0.0f;
    }
    return v;
  };
  auto load_x = [&](int t, int tile_i) {
    const long long rw = (long long)tile_i * 32 + j;
    return load_cm16(A.xs + (long long)t * d * B, (unsigned)(rw < B ? rw : B - 1));
  };
  auto clamp_item = [&](int& t_io, int& tile_io) {  // a team without an item shadows the last one (and contributes zeros)
    if (!item_live(t_io, tile_io)) { t_io = BPTT ? T - 1 : T - 1; tile_io = n_tiles - 1; }
  };
  // (the time embedding of the next step travels with it: it must be the FIRST addend of the input layer's sum -- the re-evaluated
  // pre-activations then are bit for bit the forward launch's, so a ReLU unit near its kink is on the same side in both passes --
  // and must not hold up the first MFMA of the step)
  // ZIN: row tile r of layer k of the record of (step, tile); layer k lives in zA when LH - k is even, else in zB (sdeh_bwdf2.hip)
  typedef float f32x4z __attribute__((ext_vector_type(4)));
  auto load_z = [&](int t, long long tile_i, int k, f32x16& zo) {
    if constexpr (ZIN) {
      const float* __restrict__ base = A.zrec + ((long long)t * n_tiles + tile_i) * ((LH + 1) * 2048 + OTD * 1024) + k * 2048 + r * 1024;
      unsigned lo = (unsigned)(h * 128 + j * 4) * 4u;
      asm volatile("" : "+v"(lo));
      const char* __restrict__ pb = reinterpret_cast<const char*>(base) + lo;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4z v = __builtin_nontemporal_load(reinterpret_cast<const f32x4z*>(pb + g * 1024));
        zo[4 * g] = v[0]; zo[4 * g + 1] = v[1]; zo[4 * g + 2] = v[2]; zo[4 * g + 3] = v[3];
      }
    }
  };
  f32x16 xnext, embnext, zA, zB;
  {
    int t0 = it_t, tl0 = it_tile;
    clamp_item(t0, tl0);
    xnext = load_x(t0, tl0);
    if constexpr (!ZIN) embnext = load16(ws + L.emb + t0 * C + (r * 2 + h) * 16);
    load_z(t0, tl0, LH, zA);
    load_z(t0, tl0, LH - 1, zB);
  }
  for (long long round = 0; round < n_rounds; ++round) {
    const bool live_item = item_live(it_t, it_tile);
    int cur_t = it_t, cur_tile = it_tile;
    clamp_item(cur_t, cur_tile);
    const long long tile = cur_tile;
    const long long row = tile * 32 + j;
    const bool live = live_item && row < B;
    const long long lrow = row < B ? row : B - 1;
    const float wi = live ? A.grad_rnd[lrow] : 0.0f;
    const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);

    auto load16c = [&](const float* __restrict__ plane) {  // 16 coordinates of the own tile from a coordinate-major [d][B] plane
      return load_cm16(plane, (unsigned)lrow);
    };
    auto load16r = [&](const float* __restrict__ rowp) {  // ... from a row of a [.., d] tensor (the caller's noise)
      f32x16 v;
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = cb + rrow(q) < d ? rowp[cb + rrow(q)] : 0.0f;
      return v;
    };

    f32x16 lam;
#pragma unroll
    for (int q = 0; q < 16; ++q) lam[q] = 0.0f;
    if constexpr (BPTT) {  // lambda_T = w_i d(terminal costs)/dx_T  (losses/oc.py:225,337,449-450)
      if (flags & SDEH_FLAG_TERMINAL_SECOND) {
        const f32x16 xT = load16c(A.xs + (long long)T * d * B);
        const f32x16 smu = rows16(tabs + 2 * 64 + cb), sis = rows16(tabs + 3 * 64 + cb);
#pragma unroll
        for (int q = 0; q < 16; ++q) lam[q] = wi * (smu[q] - xT[q]) * sis[q];
      }
      if ((flags & SDEH_FLAG_TERMINAL_TARGET) && A.tscore != nullptr) {
        const f32x16 st = load16c(A.tscore);
#pragma unroll
        for (int q = 0; q < 16; ++q) lam[q] = fmaf(-wi, st[q], lam[q]);
      }
    }
    const int t_first = cur_t;
    const int t_last = BPTT ? 0 : t_first;
    advance(it_t, it_tile);  // from here on: the team's NEXT item

    for (int t = t_first; t >= t_last; --t) {
      const f32x16 x = xnext;
      f32x16 embv;  // timestep_embed(t) + input bias of this wave's channels
      if constexpr (!ZIN) embv = embnext;
      int nxt_t = t - 1, nxt_tile = cur_tile;  // the step (or item) that comes next
      bool has_next = true;
      if (t <= t_last) {
        nxt_t = it_t; nxt_tile = it_tile;
        clamp_item(nxt_t, nxt_tile);
        has_next = round + 1 < n_rounds;
      }
      if (has_next) {
        xnext = load_x(nxt_t, nxt_tile);
        if constexpr (!ZIN) embnext = load16(ws + L.emb + nxt_t * C + (r * 2 + h) * 16);
      }
      // the step's other inputs: requested first, consumed after the forward pass
      f32x16 scv, xi;
      if (has_score) scv = load16c(A.sc + (long long)t * d * B);
      if (ito) {
        float n[16];
        if (A.noise != nullptr) {
          const f32x16 nv = load16r(A.noise + ((long long)t * B + lrow) * d);
#pragma unroll
          for (int q = 0; q < 16; ++q) n[q] = nv[q];
        } else {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float n4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cb + 8 * g4 < d) box_muller4(philox_block(A.seed, rng_off, grow, t, (cb + 8 * g4) >> 2), n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) n[4 * g4 + e] = n4[e];
            SDEH_FENCE();
          }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) xi[q] = cb + rrow(q) < d ? n[q] : 0.0f;
      }
      const float gam0 = has_score ? ws[L.gam + t * L.g] : 0.0f;           // gamma(t) (its first entry), requested early as well
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
      BWDF_T(tp0);
      BWDF_DECL(tp1);
      BWDF_DECL(tp2);
      f32x16 g[ZIN ? 1 : LH + 1], aown[LH > 1 ? LH - 1 : 1];
      f32x16 gz;  // ZIN: act'(Z) of the layer activated last
      f32x16 nn;
      const int bofs = 4 * h * RS + j;  // this lane's column of a plane as an MFMA B operand
      if constexpr (ZIN) {
        {  // the raw network output of this wave's coordinate tile: the record's last part (coordinates >= d, unwritten quads: zeros)
          const float* __restrict__ base = A.zrec + ((long long)t * n_tiles + tile) * ((LH + 1) * 2048 + OTD * 1024) + (LH + 1) * 2048 + ct * 1024;
          unsigned lo = (unsigned)(h * 128 + j * 4) * 4u;
          asm volatile("" : "+v"(lo));
          const char* __restrict__ pb = reinterpret_cast<const char*>(base) + lo;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const bool okq = cb + 8 * g < d;
            const f32x4z v = __builtin_nontemporal_load(reinterpret_cast<const f32x4z*>(pb + (okq ? g * 256 * 4 : 0)));
#pragma unroll
            for (int e = 0; e < 4; ++e) nn[4 * g + e] = cb + 8 * g + e < d ? v[e] : 0.0f;
          }
        }
        ws_barrier();  // every read of the previous step's planes is done
        f32x16 atop;
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(zA, atop, gz););
        plane_put(Ap[(LH + 1) & 1], r, j, h, atop);  // a_{LH+1}
        if constexpr (LH >= 2) load_z(t, tile, LH - 2, zA);
      } else {
      plane_put(Ap[0], ct, j, h, x);
      ws_barrier();
      {
        const f32x16 z0 = mm_rows<4 * OTD>(Win + (32 * r + j) * RSI + 4 * h, Ap[0] + bofs, A.n_kg, embv);
        f32x16 a1;
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z0, a1, g[0]););
        if constexpr (LH > 1) aown[0] = a1;
        plane_put(Ap[1], r, j, h, a1);
      }
      ws_barrier();
      BWDF_SET(tp1);
#pragma unroll
      for (int l = 0; l < LH; ++l) {  // hidden layer l: Z_{l+1} = W_l a_{l+1} + b_l;  a_{l+2} = act(Z_{l+1})
        const f32x16 z = mm_rows<8>(Whid + l * 64 * RSW + (32 * r + j) * RSW + 4 * h, Ap[(l + 1) & 1] + bofs, 8, rows16(bh + l * 64 + 32 * r + 4 * h));
        f32x16 an;
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z, an, g[l + 1]););
        if (l + 2 <= LH - 1) aown[l + 2 <= LH - 1 ? l + 1 : 0] = an;  // a_{l+2} is re-published later iff l + 2 <= LH - 1
        plane_put(Ap[l & 1], r, j, h, an);
        ws_barrier();
      }
      BWDF_SET(tp2);
      nn = mm_rows<8>(Wout + (32 * ct + j) * RSW + 4 * h, Ap[(LH + 1) & 1] + bofs, 8, rows16(bo + cb));
      }
      BWDF_T(tp3);

      // ======================================================================================= upstream gradient of the control
      f32x16 G, Gc, cvec, dout;
      {
        float gsum = 0.0f;
        f32x16 gcoord;
        const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
        f32x16 rr;  // reference control sigma * prior.score(x) (solver/oc.py:305-306)
#pragma unroll
        for (int q = 0; q < 16; ++q) rr[q] = 0.0f;
        if (BPTT && refc) {
          const f32x16 pmu = rows16(tabs + 0 * 64 + cb), pis = rows16(tabs + 1 * 64 + cb);
#pragma unroll
          for (int q = 0; q < 16; ++q) rr[q] = sig * (pmu[q] - x[q]) * pis[q];
        }
        // Bridge, method kl: the control entering the running cost is u + v (a plane; loaded where it is used: nothing to hold across the
        // forward pass in the launches without one)
        f32x16 cin;
#pragma unroll
        for (int q = 0; q < 16; ++q) cin[q] = 0.0f;
        if constexpr (KLB) cin = load16c(A.cost_in + (long long)t * d * B);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float mfac = 0.0f, csc = 0.0f, keep_s = 0.0f;
          if (has_score) {
            const float gam = A.g == 1 ? gam0 : ws[L.gam + t * L.g + min(cb + rrow(q), L.g - 1)];
            mfac = mult * gam;
            csc = clipf(scv[q], A.clip_score);
            keep_s = fabsf(scv[q]) <= A.clip_score ? 1.0f : 0.0f;
          }
          float gc = ito ? wi * c_i * xi[q] : 0.0f;
          if constexpr (BPTT) {
            const float u = clipf(nn[q], A.clip_model) + mfac * csc;
            gc = wi * fmaf(KLB ? cin[q] : u - rr[q], cdt, ito ? c_i * xi[q] : 0.0f);
          }
          const float gq = BPTT ? fmaf(c_u, lam[q], gc) : gc;
          Gc[q] = gc;
          G[q] = gq;
          const float gg = gq * mult * csc;
          gcoord[q] = gg;
          gsum += gg;
          cvec[q] = keep_s * mfac * gq;
          dout[q] = fabsf(nn[q]) <= A.clip_model ? gq : 0.0f;
        }
        if (has_score && live_item) {  // d loss / d gamma(t): summed over the team's trajectories
          if (A.g == 1) {
            gsum = sum_wave(gsum);
            if (lane == 0) A.gpart[(tile * T + t) * A.gw + r] = (OTD == 2 || r == 0) ? gsum : 0.0f;
          } else if (OTD == 2 || r == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float v = sum_xor16(sum_row16(gcoord[q]));  // over the 32 trajectories of this lane half
              if (j == 0) A.gpart[(tile * T + t) * A.gw + cb + rrow(q)] = v;
            }
          }
        }
      }

      // ======================================================================================= backward + weight gradients
      // delta planes alternate between D[0] and D[1]; each stage: publish delta_k (and a_k unless its plane is still intact), barrier,
      // dW_k += delta_k a_k^T, then the adjoint of the layer below
      BWDF_T(tp4);
      plane_put(Dp[0], ct, j, h, dout);
      ws_barrier();  // delta_out | a_{LH+1}
      f32x16 dl;
      if constexpr (OTD == 2) dw_acc<true>(Dp[0], r, Ap[(LH + 1) & 1], 0, dw[OTD + 2 * LH], dw[NDW - 1], bs_out, j, h);
      else dw_acc<false>(Dp[0], 0, Ap[(LH + 1) & 1], r, dw[OTD + 2 * LH], dw[OTD + 2 * LH], bs_out, j, h);
      dl = mm_cols<4 * OTD, RSW>(Wout + (4 * h) * RSW + 32 * r + j, Dp[0] + bofs, A.n_kg) * (ZIN ? gz : g[ZIN ? 0 : LH]);
      BWDF_T(tp5);
#pragma unroll
      for (int l = LH - 1; l >= 0; --l) {  // hidden layer l: dl = d loss / d Z_{l+1}
        float* __restrict__ Dl = Dp[(LH - l) & 1];
        float* __restrict__ Al = Ap[(l + 1) & 1];
        plane_put(Dl, r, j, h, dl);
        if constexpr (ZIN) {
          // a_{l+1} = act(Z_l) and act'(Z_l) from the record requested a stage ago
          f32x16 al;
          if (((LH - l) & 1) == 0) { SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(zA, al, gz);); }
          else { SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(zB, al, gz);); }
          plane_put(Al, r, j, h, al);
          if (l >= 2) {
            if (((LH - l) & 1) == 0) load_z(t, tile, l >= 2 ? l - 2 : 0, zA);
            else load_z(t, tile, l >= 2 ? l - 2 : 0, zB);
          }
          if (l == 0 && has_next) load_z(nxt_t, nxt_tile, LH, zA);  // the next step's top layer (zA has been free since its last activation)
        } else {
          if (l + 1 <= LH - 1) plane_put(Al, r, j, h, aown[l + 1 <= LH - 1 ? l : 0]);  // a_{l+1}: its plane was overwritten by a_{l+3}
        }
        ws_barrier();
        dw_acc<true>(Dl, r, Al, 0, dw[OTD + 2 * l], dw[OTD + 2 * l + 1], bs_hid[l], j, h);
        dl = mm_cols<8, RSW>(Whid + l * 64 * RSW + (4 * h) * RSW + 32 * r + j, Dl + bofs, 8) * (ZIN ? gz : g[ZIN ? 0 : l]);
      }
      BWDF_T(tp6);
      float* __restrict__ Din = Dp[(LH + 1) & 1];
      plane_put(Din, r, j, h, dl);
      plane_put(Ap[0], ct, j, h, x);
      if constexpr (ZIN) {
        if (has_next) load_z(nxt_t, nxt_tile, LH - 1, zB);
      }
      ws_barrier();  // delta_0 | x
      {
        float esum = 0.0f;
        dw_acc<(OTD == 2)>(Din, r, Ap[0], 0, dw[0], dw[OTD - 1], esum, j, h);
        esum = sum_xor32(esum);  // d loss / d (time embedding + input bias)[t][32 r + j]
        if (live_item && h == 0) A.epart[(tile * T + t) * 64 + 32 * r + j] = esum;
      }
      if constexpr (BPTT) {
        // ===================================================================================== adjoint update
        //   lambda_t = c_x lambda_{t+1} + W_in^T delta_0 + (d score term / d x)^T G + direct cost terms
        const f32x16 dx = mm_cols<8, RSI>(Win + (4 * h) * RSI + 32 * ct + j, Din + bofs, 8);
        // score terms the reference detaches (reparam.py:58,134,169,188) or obtains by autograd without a graph carry no Jacobian
        const float coef_t = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET ? wl : 0.0f);
        const float coef_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR ? 1.0f - wl : 0.0f;
        const float jac_t = (!has_score || (flags & (SDEH_FLAG_DETACH_SCORE | SDEH_FLAG_TARGET_SCORE_CONST))) ? 0.0f : coef_t;
        const float jac_p = (!has_score || (flags & SDEH_FLAG_DETACH_SCORE)) ? 0.0f : coef_p;
        f32x16 vt;
#pragma unroll
        for (int q = 0; q < 16; ++q) vt[q] = 0.0f;
        if (jac_t != 0.0f) {  // closed-form target scores are differentiated through x
          if (A.target.kind == SDEH_DENS_DIAG_GAUSS) {
            const f32x16 tis = rows16(tabs + 5 * 64 + cb);
#pragma unroll
            for (int q = 0; q < 16; ++q) vt[q] = -tis[q] * cvec[q];
          } else if (A.target.kind == SDEH_DENS_MULTI_WELL) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float y = x[q] - A.target.p1;
              vt[q] = (cb + rrow(q) < A.target.n_comp ? -4.0f * (3.0f * y * y - A.target.p0) : -1.0f) * cvec[q];
            }
          } else if (A.target.kind == SDEH_DENS_FUNNEL) {
            // s_0 = -x0/var - (d-1)/2 + e^{-x0} sum x_j^2 / 2,  s_j = -x_j e^{-x0}   (coordinate 0 = register 0 of the h = 0 half of
            // coordinate tile 0).  The sums run over all coordinates: two lane halves, and with d > 32 the two waves of the team --
            // they meet in the delta plane that has been free since the last barrier (one more barrier, this configuration only).
            const bool tile0 = ct == 0;
            float x0 = __shfl(x[0], j), c0 = __shfl(cvec[0], j);
            float sq = 0.0f, cx = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const bool first = q == 0 && h == 0 && tile0;
              sq = fmaf(first ? 0.0f : x[q], x[q], sq);
              cx = fmaf(first ? 0.0f : cvec[q], x[q], cx);
            }
            sq = sum_xor32(sq);
            cx = sum_xor32(cx);
            if constexpr (OTD == 2) {
              float* __restrict__ sx = Dp[LH & 1];
              if (h == 0) {
                sx[(4 * r) * RS + j] = sq;
                sx[(4 * r + 1) * RS + j] = cx;
                if (tile0) { sx[8 * RS + j] = x0; sx[9 * RS + j] = c0; }
              }
              ws_barrier();
              sq = sx[j] + sx[4 * RS + j];
              cx = sx[RS + j] + sx[5 * RS + j];
              x0 = sx[8 * RS + j];
              c0 = sx[9 * RS + j];
            }
            const float iv = __expf(-x0);
#pragma unroll
            for (int q = 0; q < 16; ++q) vt[q] = iv * (c0 * x[q] - cvec[q]);
            if (h == 0 && tile0) vt[0] = c0 * (-1.0f / A.target.p0 - 0.5f * iv * sq) + iv * cx;
          }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) lam[q] = fmaf(jac_t, vt[q], fmaf(c_x, lam[q], dx[q]));
        if constexpr (KLB) {  // Bridge, method kl: the inference terms' d loss / d x_t
          const f32x16 lin = load16c(A.lam_in + (long long)t * d * B);
#pragma unroll
          for (int q = 0; q < 16; ++q) lam[q] += live ? lin[q] : 0.0f;  // (lanes beyond the batch shadow its last row: nothing from them)
        }
        if (jac_p != 0.0f || refc) {
          const f32x16 pis = rows16(tabs + 1 * 64 + cb);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float v = fmaf(-jac_p * pis[q], cvec[q], lam[q]);  // Gaussian prior: J = -1/sigma^2
            if (refc) v = fmaf(sig * pis[q], Gc[q], v);         // cost depends on x through sigma * prior.score(x)
            lam[q] = v;
          }
        }
      }
      BWDF_T(tp7);
      BWDF_ADD(0, tp0, tp1); BWDF_ADD(1, tp1, tp2); BWDF_ADD(2, tp2, tp3); BWDF_ADD(3, tp3, tp4); BWDF_ADD(4, tp4, tp5);
      BWDF_ADD(5, tp5, tp6); BWDF_ADD(6, tp6, tp7); BWDF_ADD(7, tp0, tp7); BWDF_ADD(8, 0ull, 1ull);
      par ^= 1;
    }
  }

  // ---- the team's partial gradients ---------------------------------------------------------------------------------------
  float* __restrict__ rec = A.wpart + (long long)team_g * A.wsize;
#pragma unroll
  for (int k = 0; k < OTD; ++k) store_tile(rec, DPP, r, k, j, h, dw[k]);
#pragma unroll
  for (int l = 0; l < LH; ++l) {
    store_tile(rec + off_whid<OTD>() + l * 4096, 64, r, 0, j, h, dw[OTD + 2 * l]);
    store_tile(rec + off_whid<OTD>() + l * 4096, 64, r, 1, j, h, dw[OTD + 2 * l + 1]);
    float b = bs_hid[l];
    b += __shfl_xor(b, 32);
    if (h == 0) rec[off_bhid<OTD, LH>() + l * 64 + 32 * r + j] = b;
  }
  {
    float b = bs_out;
    b += __shfl_xor(b, 32);
    if constexpr (OTD == 2) {
      store_tile(rec + off_wout<OTD, LH>(), 64, r, 0, j, h, dw[OTD + 2 * LH]);
      store_tile(rec + off_wout<OTD, LH>(), 64, r, 1, j, h, dw[OTD + 2 * LH + 1]);
      if (h == 0) rec[off_bout<OTD, LH>() + 32 * r + j] = b;
    } else {
      store_tile(rec + off_wout<OTD, LH>(), 64, 0, r, j, h, dw[OTD + 2 * LH]);
      if (h == 0 && r == 0) rec[off_bout<OTD, LH>() + j] = b;
    }
  }
}

template <int OTD, bool BPTT, int LH, bool KLB = false, bool ZIN = false>
static int launch_bwdf_t(const BwdfArgs& a, hipStream_t stream) {
  if constexpr (BPTT && LH == 2 && !KLB) {
    if (a.cost_in != nullptr && a.lam_in != nullptr) return launch_bwdf_t<OTD, BPTT, LH, true>(a, stream);
  }
  if constexpr (!KLB && !ZIN) {
    if (a.zrec != nullptr) return launch_bwdf_t<OTD, BPTT, LH, false, true>(a, stream);
  }
  if (!KLB && (a.cost_in != nullptr || a.lam_in != nullptr)) return SDEH_ERR_UNSUPPORTED;  // (two hidden layers, both planes)
  const size_t lds_bytes = (size_t)bwdf::lds_floats<OTD, LH>() * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwdf_kernel<OTD, BPTT, LH, KLB, ZIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bwdf_kernel<OTD, BPTT, LH, KLB, ZIN>), dim3((unsigned)(a.n_slots / 2)), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// floats of one team's partial-gradient record
int bwdf_wsize(int d, int n_hidden) {
  const int otd = d <= 32 ? 1 : 2;
  return 64 * 32 * otd + n_hidden * 4096 + 32 * otd * 64 + n_hidden * 64 + 32 * otd;
}

// compiled for this shape, and does the LDS image (one natural copy of every weight matrix + 8 exchange planes) fit?
bool bwdf_fits(int d, int n_hidden) {
  if (n_hidden < 1 || n_hidden > bwdf::kMaxLH || d < 1 || d > 64) return false;
  const int otd = d <= 32 ? 1 : 2;
  const long long floats = 64 * (otd == 1 ? 36 : 68) + (long long)n_hidden * 64 * bwdf::RSW + 32 * otd * bwdf::RSW + n_hidden * 64 + 64 + bwdf::TABS +
                           2 * 4 * bwdf::PLANE;
  return floats * 4 <= 160 * 1024;
}

// teams (of 32 trajectories) the launch uses: two per workgroup, one workgroup per CU (the LDS image is ~130-146 KB)
int bwdf_slots(long long batch, int n_steps, bool bptt) {
  const long long tiles = (batch + 31) / 32;
  const long long items = bptt ? tiles : tiles * n_steps;
  const long long wgs = (items + 1) / 2;
  return 2 * (int)(wgs < 256 ? wgs : 256);
}

template <int OTD, bool BPTT>
static int launch_bwdf_l(const BwdfArgs& a, hipStream_t stream) {
  switch (a.n_hidden) {
    case 1: return launch_bwdf_t<OTD, BPTT, 1>(a, stream);
    case 2: return launch_bwdf_t<OTD, BPTT, 2>(a, stream);
    case 3: return launch_bwdf_t<OTD, BPTT, 3>(a, stream);  // (with two coordinate tiles the LDS image is 159.5 of the 160 KB)
    default: return SDEH_ERR_UNSUPPORTED;
  }
}

#ifdef SDEH_BWDF_PROFILE
static void bwdf_prof_dump(hipStream_t stream) {
  (void)hipStreamSynchronize(stream);
  unsigned long long v[16];
  (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(bwdf_prof), sizeof(v));
  const double n = v[8] ? (double)v[8] : 1.0;
  fprintf(stderr, "bwdf phases (cycles per step of block 0 / wave 0, %llu steps): in %.0f | hidden %.0f | out %.0f | elementwise %.0f | dW_out+adj %.0f | "
          "hidden stages %.0f | in stage %.0f | step %.0f\n", v[8], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5] / n, v[6] / n, v[7] / n);
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(bwdf_prof), z, sizeof(z));
}
#endif

int launch_bwdf(const BwdfArgs& a, hipStream_t stream) {
#ifdef SDEH_BWDF_PROFILE
  struct Dump { hipStream_t s; ~Dump() { bwdf_prof_dump(s); } } dump{stream};
#endif
  const bool bptt = !(a.flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  if (a.d <= 32) return bptt ? launch_bwdf_l<1, true>(a, stream) : launch_bwdf_l<1, false>(a, stream);
  return bptt ? launch_bwdf_l<2, true>(a, stream) : launch_bwdf_l<2, false>(a, stream);
}

}  // namespace sdeh
