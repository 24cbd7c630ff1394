// Trajectory kernels for WIDE control networks (FourierMLP with C = 128 / 256 channels, state dimension d <= 256): the
// shape of BASELINE.json configs[4] (Bridge, d = 196, C = 256 -- "MFMA-bound").  Same path as sdeh_traj_ws.hpp (reference
// losses/oc.py:156-230, 286-343, 400-457 with models/mlp.py:85-122, models/reparam.py, eq/sdes.py, distr/*.py) for networks
// whose packed weights (0.9 MB at C = 256, d = 196) are far beyond one CU's LDS.
//
// Design.  The registers are the largest on-chip store (512 KB per CU), the matrix pipe the only unit
// that matters (463 kFLOP per trajectory-step in the network against ~4 kFLOP around it), and the weights are read-only and
// L2-resident.  So:
//   * a workgroup of FOUR waves (one per SIMD) owns CT column tiles of 32 trajectories for all T steps;
//   * CHANNEL SPLIT: wave w owns the output row tiles {w, w + 4, ...} of every layer, for all CT column tiles -- its
//     accumulators are (C / 128) x CT MFMA tiles, nothing else of the network lives in registers;
//   * the A operands (weights) of a wave are its OWN rows: they stream straight from the workspace in L2 as coalesced
//     global_load_dwordx4 (four k-steps per load, 1 KB per wave-instruction, prefetched two groups ahead) -- no LDS staging,
//     nothing to share;
//   * the B operands (a layer's input activations) are what the waves share: [channel][trajectory] planes in LDS, read as
//     ds_read_b32 (lane (j, h) reads row 2 s + h, column j: conflict-free), natural channel order;
//   * after a layer each wave activates its tiles and writes them to the other plane (ping-pong: one workgroup barrier per
//     layer); the state x lives in the same planes ("layer -1") and, for the elementwise work, in the registers of the wave
//     that owns its coordinates' output tiles -- in the accumulator layout, so the network output meets x without any
//     layout change: lane (j, h), register q of tile t <-> coordinate 32 t + (q & 3) + 8 (q >> 2) + 4 h of trajectory j.
//     Four consecutive registers are four consecutive coordinates = one Philox block, so the Gaussian draws are the same
//     stream the narrow kernels consume;
//   * per-trajectory sums over coordinates (running cost, Ito term, log-densities, funnel statistics) are reduced over the
//     lane's registers, the two lane halves, and the four waves (through a few LDS slots, read after the next barrier).
// Everything in the step loop is a run-time loop over k-groups: no dimension is a template parameter except the tile counts.
// No implicit mul+add contraction in this translation unit: whether hipcc fuses `a - b * c` depends on the surrounding code of each
// template instantiation, and the 32- and 64-trajectory instantiations must produce bitwise identical trajectories (every fused
// multiply-add the kernels want is written as fmaf).
#include "sdeh_wide_common.hpp"

namespace sdeh {

// ---------------------------------------------------------------------------------------------------------
// the kernel (no inference control)
// ---------------------------------------------------------------------------------------------------------
// SINGLE: one activation plane instead of two (128 trajectories per workgroup: 4 column tiles x 256 channels x 4 B = 128 KB): a
// layer's output overwrites its input in place, behind an extra barrier.
// GMM: the mixture-target code (tables in LDS, partial logits, responsibilities) is compiled in -- a separate instantiation, so that
// the closed-form targets' kernels carry none of its registers.
template <int OTW, int CT, bool SINGLE, bool GMM>
__global__ __launch_bounds__(256) void traj_wide_kernel(const TrajArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  constexpr int RS = 32 * CT;
  const int d = A.d, OTD = L.otd, C = L.c;
  const int rows = C > 32 * OTD ? C : 32 * OTD;

  WideCtx cx;
  cx.RS = RS; cx.d = d;
  cx.wave = w; cx.lane = lane; cx.j = j; cx.h = h;
  cx.planes = lds; cx.plane_floats = rows * RS;
  cx.scr = lds + (SINGLE ? 1 : 2) * rows * RS;
  float* tabs = cx.scr + kWideSlots * 4 * RS;
  const int tab_stride = 2 * L.dp + 4;
  for (int i = tid; i < 3 * tab_stride; i += 256) {
    const int which = i / tab_stride, o = i % tab_stride;
    tabs[i] = o <= 2 * L.dp ? ws[L.dg[which] + o] : 0.0f;
  }
  cx.tab0 = tabs; cx.tab1 = tabs + tab_stride; cx.tab2 = tabs + 2 * tab_stride;
  float* bias_lds = tabs + 3 * tab_stride;  // hidden biases [n_hidden][C] then the out-layer bias [32 otd] (b_hid and b_out are adjacent)
  for (int i = tid; i < L.n_hidden * C + 32 * OTD; i += 256) bias_lds[i] = ws[L.b_hid + i];
  cx.bias = bias_lds;
  WideGmm gm;  // mixture target: tables and per-trajectory scratch behind the biases
  gm.K = GMM && A.target.kind == SDEH_DENS_GMM ? A.target.n_comp : 0;
  gm.d4 = L.gmm_row;
  {
    float* g0p = bias_lds + ((L.n_hidden * C + 32 * OTD + 3) & ~3);
    float* mu = g0p; float* aa = mu + gm.K * gm.d4; float* ck = aa + gm.K * gm.d4;
    gm.mu = mu; gm.a = aa; gm.ck = ck;
    gm.part = ck + ((gm.K + 3) & ~3); gm.resp = gm.part + 4 * gm.K * RS; gm.lse = gm.resp + gm.K * RS;
    if constexpr (GMM) {
      for (int i = tid; i < gm.K * gm.d4; i += 256) { mu[i] = ws[L.gmm_lg + i]; aa[i] = ws[L.gmm_sc + i]; }
      for (int i = tid; i < gm.K; i += 256) ck[i] = ws[L.gmm_c + i];
    }
  }

  const int nto = (OTD > w ? 1 : 0) + (OTD > w + 4 ? 1 : 0);  // coordinate tiles of this wave: w, w + 4
  const long long row0 = (long long)blockIdx.x * RS;
  const int flags = A.flags, ctrl_kind = A.ctrl_kind, loss_kind = A.loss_kind, act = A.act;
  const DensArgs tgt = A.target;
  const bool lv = flags & SDEH_FLAG_CHANGE_SDE_CTRL;
  const bool need_t = ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool refc = (flags & SDEH_FLAG_REFERENCE_CTRL) && loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool need_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR || refc;
  const bool expo = loss_kind == SDEH_LOSS_EXPONENTIAL;
  const bool vec4 = (d & 3) == 0;  // rows are float4-aligned groups of coordinates

  // ---- x0 -> registers (accumulator layout) and plane 0 ---------------------------------------------------------------
  f32x16 xr[2][CT];
  long long lrow[CT];
  bool live[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const long long r = row0 + 32 * c + j;
    live[c] = r < A.batch;
    lrow[c] = live[c] ? r : A.batch - 1;  // dead lanes shadow the last row and never store
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = 32 * (w + 4 * k) + rho(q, h);
        xr[k][c][q] = (k < nto && cc < d) ? A.x0[lrow[c] * d + cc] : 0.0f;
      }
  if (A.xs != nullptr && !A.seg_continue) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int cc = 32 * (w + 4 * k) + rho(q, h);
          if (k < nto && cc < d && live[c]) A.xs[lrow[c] * d + cc] = xr[k][c][q];
        }
  }
  __syncthreads();  // tables staged
  int p = 0;
  wide_publish<CT>(cx, cx.plane(p), xr, nto);
  if constexpr (GMM) wide_gmm_partials<CT>(cx, gm, xr, nto);
  if (flags & SDEH_FLAG_INIT_LOGP) wide_gauss_quad<CT>(cx, cx.tab2, xr, nto, WSL_LOGP_A);
  __syncthreads();  // x_0 published
  if constexpr (GMM) { if (w == 0) wide_gmm_normalise<CT>(cx, gm); }  // visible to the other waves behind the first layer barrier
  float rnd[CT];  // owned by wave 0, lanes of half 0
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    rnd[c] = 0.0f;
    // Distribution.log_prob = unnorm_log_prob - log_norm_const (distr/base.py:116-119): the constants cancel
    if (flags & SDEH_FLAG_INIT_LOGP) rnd[c] = cx.tab2[2 * L.dp] - 0.5f * wide_slot_sum(cx, WSL_LOGP_A, 32 * c + j);
    if (A.seg_continue) rnd[c] = live[c] ? A.rnd[row0 + 32 * c + j] : 0.0f;  // a later segment of the grid (sdeh_simulate_fwd_steps)
  }
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
  const int wt = w & (L.ot - 1);  // hidden-layer tile of this wave (C = 64: waves 2, 3 double the tiles of waves 0, 1; see wide_mlp)
  unsigned voff_in[OTW];
#pragma unroll
  for (int k = 0; k < OTW; ++k) voff_in[k] = (unsigned)(((wt + 4 * k) * 64 + lane) * 16);
  WidePre<OTW> pre_in;
  wide_prefetch<OTW>(pre_in, ws + L.w_in, L.ot * 256, L.dp8 / 8, voff_in);
  f32x16 emb[OTW];  // FourierMLP.timestep_embed(t_i) + input_embed.bias for this wave's channels (added when layer 0 is activated)
#pragma unroll
  for (int k = 0; k < OTW; ++k) emb[k] = load16(ws + L.emb + A.step0 * C + ((wt + 4 * k) * 2 + h) * 16);

  const int step_end = A.step0 + A.n_steps;
  for (int i = A.step0; i < step_end; ++i) {
    cfp cf = as_const(ws + L.coef + i * kCoefStride);
    const float dt = cf[CF_DT], sqdt = cf[CF_SQDT], sig = cf[CF_SIGMA];
    // funnel statistics of x_i (published with it); read before the planes move on
    float fs[CT], fx0[CT], fiv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      fs[c] = fx0[c] = fiv[c] = 0.0f;
      if (need_t && tgt.kind == SDEH_DENS_FUNNEL) {
        fs[c] = wide_slot_sum(cx, WSL_PRESQ, 32 * c + j);
        fx0[c] = cx.scr[(WSL_X0 * 4) * RS + 32 * c + j];
        fiv[c] = __expf(-fx0[c]);
      }
    }
    // ---- network pass ---------------------------------------------------------------------------------------------
    f32x16 nn[2][CT];
    const int pout = wide_mlp<OTW, CT, false>(cx, ws, L, act, p, SINGLE, nullptr, pre_in, emb, bias_lds, nn, nto);
    p = SINGLE ? 0 : 1 - pout;  // the plane no wave reads any more: the new state goes there

    // ---- elementwise part on this wave's coordinates (reparam.py controls, oc.py cost / update) ---------------------------
    // One (tile, column tile) pair = 16 registers at a time, in stages whose wave-uniform switches (target kind, control kind,
    // noise source, cost form) sit OUTSIDE the 16-element loops -- selected per element they cost ~250 instructions per element
    // (a third of a step); staged, ~40.
    const float c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], dt, 1.0f);
    const float c_u = expo ? cf[CF_B2S2] : sig * dt;
    const float c_n = expo ? cf[CF_SBK] : sig * sqdt;
    const float c_i = expo ? cf[CF_SBK] : sqdt;  // Ito term: sum(g * xi) * c_i
    const float wl = cf[CF_W];
    cfp gam = as_const(ws + L.gam + i * L.g);
    const float g0 = gam[0];
    const float mult = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig;  // Lerp*: ctrl + sde.diff(t) * score
    float costl[CT], itol[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) costl[c] = itol[c] = 0.0f;
    // Everything below that depends on the coordinate index is a cheap function of (w, h, compile-time constants); as loop
    // invariants the compiler would precompute it for all 32 x CT elements of a lane (table addresses, validity masks: hundreds
    // of registers held across the step loop) -- an opaque copy of h per step keeps it recomputed where it is used.
    int hv = h;
    asm volatile("" : "+v"(hv));
    WideScore sq;
    sq.ctrl_kind = ctrl_kind; sq.g = L.g; sq.need_t = need_t; sq.need_p = need_p; sq.tgt = tgt; sq.wl = wl; sq.mult = mult;
    sq.scale_score = A.scale_score; sq.clip_score = A.clip_score; sq.g0 = g0; sq.d = d; sq.gmm = &gm;
    if (tgt.kind == SDEH_DENS_EXTERNAL && need_t) {
      sq.ext = A.ext_score + (long long)(i - A.step0) * A.ext_stride;
#pragma unroll
      for (int c = 0; c < CT; ++c) sq.ext_off[c] = lrow[c] * d;
    }
    auto vtile = [&](f32x16& x, const f32x16& nnv, int t, int c) {
      const int cb = 32 * t + 4 * hv;  // register q <-> coordinate cb + (q & 3) + 8 (q >> 2)
      auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
      float sterm[16], psc[16];
      if constexpr (GMM) {
        // training forward on a mixture target (sdeh_simulate_fwd_train2 on a wide plan): keep the combined score entering the control,
        // row-major [T, B, d] -- the backward (sdeh_wide_bwd.hip) evaluates no mixture
        float scr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) scr[q] = 0.0f;
        wide_score_term16<GMM>(sq, cx, x, cb, c, fs[c], fx0[c], fiv[c], ws + L.gam + i * L.g, sterm, psc, A.sc_out != nullptr ? scr : nullptr);
        if (A.sc_out != nullptr && live[c]) {
          float* __restrict__ sp = A.sc_out + ((long long)i * A.batch + lrow[c]) * d + cb;
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (coord(q) < d) sp[(q & 3) + 8 * (q >> 2)] = scr[q];
        }
      } else {
        wide_score_term16<GMM>(sq, cx, x, cb, c, fs[c], fx0[c], fiv[c], ws + L.gam + i * L.g, sterm, psc);
      }
      SDEH_FENCE();
      // ---- Gaussian draws: register group g4 = coordinates cb + 8 g4 .. + 3 = Philox block (cb + 8 g4) / 4 ------------------------
      float n[16];
      wide_noise16(A.noise != nullptr ? A.noise + ((long long)i * A.batch + lrow[c]) * d : nullptr, vec4, cb, d, A.seed, rng_off,
                   (unsigned long long)(A.row_offset + lrow[c]), i, n);
      // ---- u = clip(nn) + score term; running-cost / Ito partial sums; state update ------------------------------------------
      const f32x16 bo = load16(bias_lds + L.n_hidden * C + (t * 2 + hv) * 16);  // out_layer.bias of these 16 coordinates
      float u[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float uq = clipf(nnv[q] + bo[q], A.clip_model) + sterm[q];
        u[q] = coord(q) < d ? uq : 0.0f;
      }
      if (!refc) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          costl[c] = fmaf(u[q], u[q], costl[c]);
          itol[c] = fmaf(u[q], n[q], itol[c]);
        }
      } else {  // reference_ctrl = sigma prior.score (solver/oc.py:305-306), evaluated at the step's input
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float rs = sig * psc[q];
          const float gm = coord(q) < d ? u[q] - rs : 0.0f;
          costl[c] = lv ? fmaf(gm, u[q] - 0.5f * (rs + u[q]), costl[c]) : fmaf(gm, gm, costl[c]);
          itol[c] = fmaf(gm, n[q], itol[c]);
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float xn = fmaf(c_u, u[q], fmaf(c_n, n[q], c_x * x[q]));
        x[q] = coord(q) < d ? xn : 0.0f;
      }
      if (A.xs != nullptr && live[c]) {
        float* __restrict__ xp = A.xs + ((long long)(i + 1) * A.batch + lrow[c]) * d + cb;
        if (vec4) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            if (cb + 8 * g4 < d) *reinterpret_cast<float4*>(xp + 8 * g4) = float4{x[4 * g4], x[4 * g4 + 1], x[4 * g4 + 2], x[4 * g4 + 3]};
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (coord(q) < d) xp[(q & 3) + 8 * (q >> 2)] = x[q];
        }
      }
      SDEH_FENCE();
    };
    if (nto > 0) {
#pragma unroll
      for (int c = 0; c < CT; ++c) vtile(xr[0][c], nn[0][c], w, c);
    }
    if (nto > 1) {
#pragma unroll
      for (int c = 0; c < CT; ++c) vtile(xr[1][c], nn[1][c], w + 4, c);
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float cs = half_sum(costl[c]), is = half_sum(itol[c]);
      if (h == 0) {
        cx.scr[(WSL_COST * 4 + w) * RS + 32 * c + j] = cs;
        cx.scr[(WSL_ITO * 4 + w) * RS + 32 * c + j] = is;
      }
    }
    if (SINGLE) wide_barrier();  // every wave has finished reading the (only) plane in its out-layer
    // the next step's first operands (input layer A groups, time embedding of step i + 1) travel across the publish barrier
    wide_prefetch<OTW>(pre_in, ws + L.w_in, L.ot * 256, L.dp8 / 8, voff_in);
#pragma unroll
    for (int k = 0; k < OTW; ++k) emb[k] = load16(ws + L.emb + (i + 1 < step_end ? i + 1 : i) * C + ((wt + 4 * k) * 2 + h) * 16);
    wide_publish<CT>(cx, cx.plane(p), xr, nto);
    if constexpr (GMM) wide_gmm_partials<CT>(cx, gm, xr, nto);
    wide_barrier();  // x_{i+1}, its statistics and this step's cost partials are visible
    if constexpr (GMM) { if (w == 0) wide_gmm_normalise<CT>(cx, gm); }
    // ---- running cost (losses/oc.py:204-211, 319-323, 418-431) and Ito term -----------------------------------------------
    if (w == 0) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float cost = wide_slot_sum(cx, WSL_COST, 32 * c + j);
        if (!(refc && lv)) cost *= 0.5f;
        rnd[c] = expo ? fmaf(cf[CF_B2S2], cost, rnd[c]) : fmaf(cost, dt, rnd[c]);
        if (loss_kind == SDEH_LOSS_TIME_REVERSAL && !(flags & SDEH_FLAG_TRAIN)) rnd[c] -= cf[CF_DDIV];
        if (flags & SDEH_FLAG_ITO) rnd[c] = fmaf(wide_slot_sum(cx, WSL_ITO, 32 * c + j), c_i, rnd[c]);
      }
    }
  }

  // the prefetch issued behind the LAST step has no consumer: it must land before its registers are handed to the terminal phase
  // (a late load there showed up as a wrong terminal log-density in a golden test under a loaded GPU, once in a few hundred runs)
#pragma unroll
  for (int g = 0; g < kWidePD; ++g) wide_vmwait<0, OTW>(pre_in.a[g]);

  // ---- terminal costs (oc.py:225, 337, 449-450) -------------------------------------------------------------------------
  if (flags & SDEH_FLAG_TERMINAL_SECOND) wide_gauss_quad<CT>(cx, cx.tab2, xr, nto, WSL_LOGP_A);
  if (flags & SDEH_FLAG_TERMINAL_TARGET) {
    if (tgt.kind == SDEH_DENS_DIAG_GAUSS) wide_gauss_quad<CT>(cx, cx.tab0, xr, nto, WSL_LOGP_B);
    else if (tgt.kind == SDEH_DENS_MULTI_WELL) wide_mwell_sum<CT>(cx, tgt, xr, nto, WSL_LOGP_B);
  }
  __syncthreads();
  if (w == 0 && h == 0) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int col = 32 * c + j;
      float r = rnd[c];
      if (flags & SDEH_FLAG_TERMINAL_SECOND) r += cx.tab2[2 * L.dp] - 0.5f * wide_slot_sum(cx, WSL_LOGP_A, col);
      if (flags & SDEH_FLAG_TERMINAL_TARGET) {
        float lp = 0.0f;
        if (tgt.kind == SDEH_DENS_DIAG_GAUSS) lp = cx.tab0[2 * L.dp] - 0.5f * wide_slot_sum(cx, WSL_LOGP_B, col) + tgt.lnc;
        else if (tgt.kind == SDEH_DENS_MULTI_WELL) lp = -wide_slot_sum(cx, WSL_LOGP_B, col);
        else if (tgt.kind == SDEH_DENS_FUNNEL) {  // distr/funnel.py:54-69 (the statistics of x_T were published with it)
          const float x0v = cx.scr[(WSL_X0 * 4) * RS + col], sq = wide_slot_sum(cx, WSL_PRESQ, col);
          const float first = -0.5f * __logf(6.283185307179586f * tgt.p0) - 0.5f * x0v * x0v / tgt.p0;
          const float other = -(float)(d - 1) * (x0v + 1.8378770664093453f) * 0.5f - 0.5f * sq * __expf(-x0v);
          lp = first + other + tgt.lnc;
        } else if (GMM && tgt.kind == SDEH_DENS_GMM) {
          lp = gm.lse[col] + tgt.lnc;  // normalised with the last publish
        }
        r -= clipf(lp, A.clip_target);
      }
      if (live[c]) A.rnd[row0 + col] = r;
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = 32 * (w + 4 * k) + rho(q, h);
        if (k < nto && cc < d && live[c]) A.xT[lrow[c] * d + cc] = xr[k][c][q];
      }
  if constexpr (GMM) {
    // training forward (method kl): 1[|log rho(x_T)| <= clip_target] target.score(x_T), row-major [B, d] -- d (terminal cost) / d x_T,
    // negated (the responsibilities of x_T were normalised behind the last publish; visible since the barrier above)
    if (A.tsc_out != nullptr && tgt.kind == SDEH_DENS_GMM) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const int cb = 32 * (w + 4 * k) + 4 * h, col = 32 * c + j;
            float sc[16];
            wide_gmm_score16(cx, gm, xr[k][c], cb, col, sc);
            const float lp = gm.lse[col] + tgt.lnc;
            const float keep = fabsf(lp) <= A.clip_target ? 1.0f : 0.0f;
            if (live[c]) {
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                const int cc = cb + (q & 3) + 8 * (q >> 2);
                if (cc < d) A.tsc_out[lrow[c] * d + cc] = keep * sc[q];
              }
            }
          }
        }
    }
  }
}

inline size_t wide_lds_bytes(const WsLayout& L, int ct, int n_planes, int K) {
  const int rows = L.c > 32 * L.otd ? L.c : 32 * L.otd;
  const size_t gmm = K > 0 ? (size_t)2 * K * L.gmm_row + ((K + 3) & ~3) + (size_t)5 * K * 32 * ct + 32 * ct : 0;
  return ((size_t)n_planes * rows * 32 * ct + (size_t)kWideSlots * 4 * 32 * ct + 3 * (2 * L.dp + 4) + L.n_hidden * L.c + 32 * L.otd + 4 + gmm) *
         sizeof(float);
}

template <int OTW, int CT, bool GMM = false>
static int launch_wide_t(const TrajArgs& a, hipStream_t stream) {
  constexpr bool SINGLE = CT > 2;
  const size_t lds_bytes = wide_lds_bytes(a.lay, CT, SINGLE ? 1 : 2, a.target.kind == SDEH_DENS_GMM ? a.target.n_comp : 0);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_wide_kernel<OTW, CT, SINGLE, GMM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  const unsigned grid = (unsigned)((a.batch + 32 * CT - 1) / (32 * CT));
  hipLaunchKernelGGL((traj_wide_kernel<OTW, CT, SINGLE, GMM>), dim3(grid), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// Column tiles per workgroup: 64 trajectories (CT = 2) halve the weight traffic per trajectory and the barriers per MFMA, but
// a launch needs >= 256 workgroups to use every CU: below 16 384 trajectories the 32-trajectory form fills more CUs.
int launch_wide(const TrajArgs& a, hipStream_t stream, int* ct_used) {
  const char* force = plan_opt(OPT_WIDE_CT);  // testing aid: "1" | "2" (a plan option)
  // (128 trajectories per workgroup -- CT = 4, one plane -- would quarter the operand loads per MFMA, but the state and the network
  // output of 128 trajectories do not fit the wave's registers next to the elementwise phase: it spills and measures 36.7 ms against
  // 29.6 ms for CT = 2 at B = 32 768; those instantiations -- 770-1154 spilled registers -- were removed in round 3)
  int ct = a.batch > 32 * 256 ? 2 : 1;
  if (force != nullptr && (force[0] == '1' || force[0] == '2')) ct = force[0] - '0';
  const int K = a.target.kind == SDEH_DENS_GMM ? a.target.n_comp : 0;
  if (K > 0) ct = 1;  // mixture targets: tables, partial logits and responsibilities take the LDS of the second column tile
  if (ct_used != nullptr) *ct_used = ct;
  const int otw = a.lay.c == 64 ? 1 : a.lay.c / 128;  // C = 64 (d > 64): one tile per wave, two waves with hidden tiles
  if (K > 0) return otw == 2 ? launch_wide_t<2, 1, true>(a, stream) : launch_wide_t<1, 1, true>(a, stream);
  if (otw == 2) return ct == 2 ? launch_wide_t<2, 2>(a, stream) : launch_wide_t<2, 1>(a, stream);
  if (otw == 1) return ct == 2 ? launch_wide_t<1, 2>(a, stream) : launch_wide_t<1, 1>(a, stream);
  return SDEH_ERR_UNSUPPORTED;
}

// =========================================================================================================
// Bridge: TimeReversalLoss with an inference control (losses/oc.py:189-202) on wide networks -- BASELINE.json configs[4].
//
//   u = generative_ctrl(s, x);  (div, v) = compute_divx(inference_ctrl, s, x)   [exact divergence, utils/autograd.py:14-22]
//   rnd += sigma div dt;  cost on u + v (kl) / (u + v).(u_sde - (u - v)/2) (lv);  Ito term on u + v;  x driven by u alone.
//
// The reference takes d backward passes through the inference network per step.  Here a workgroup owns ONE column
// tile of 32 trajectories; per step it runs the two network passes channel-split as above (the inference network's pass also
// leaves act'(z_l) of every layer in LDS planes), then the diagonal of the Jacobian, one coordinate per wave at a time:
//     J_jj = sum_ch F[ch] D[ch] G[ch]      with, for an inference network with Lh hidden layers,
//       Lh = 0:  F = W_in[:, j]                            D = act'(z_0)   G = W_out[j, :]
//       Lh = 1:  F = W_1 (act'(z_0) . W_in[:, j])          D = act'(z_1)   G = W_out[j, :]
//       Lh = 2:  F = W_1 (act'(z_0) . W_in[:, j])          D = act'(z_1)   G = W_2^T (act'(z_2) . W_out[j, :])
// i.e. a forward-mode tangent through the first hidden layer meets a reverse-mode adjoint through the second: two C x C products
// per coordinate on the matrix pipe (A operands: the packed W_1 / transposed W_2 streamed from L2, all C/32 row tiles per wave;
// B operands: an act' plane row times one column / row of the in / out layer, formed on the fly), no dependence between them, and
// run-time loops only.  ClippedCtrl's clamp contributes the 0/1 mask d clip(v_j)/d v_j, the LerpPriorCtrl score term its
// closed-form derivative.  Coordinates are dealt to waves in 32 fixed groups (j mod 32) whose sums are kept apart and added in a
// fixed order at the very end, so the result does not depend on how many workgroups (`split`) share a column tile: small batches
// put 2 .. 8 workgroups on one tile (each repeats the cheap network passes and state update bit-identically and takes its share
// of the coordinates), which is what fills the chip at configs[4]'s 4096 trajectories per GPU.
// =========================================================================================================
constexpr int kDivGroups = 32;

// acc[t] += Wp[tile t][:] . (dplane[:, traj] * col[:])  for all OT row tiles (t0 .. t0 + NT - 1) of one hidden layer.
//   wgrp: packed layer (k-groups of OT tiles);  dpl: act' plane + h * 32 + j;  col: LDS row of C floats in B order (+ 4 h)
template <int NT>
__device__ __forceinline__ void wide_tangent_pass(const float* __restrict__ wgrp, int OT, int t0, int NS4, unsigned lane_off,
                                                  const float* __restrict__ dpl, const float* __restrict__ col, f32x16 (&acc)[NT]) {
  f32x4 a[2][NT];  // ring of two k-groups (a group is 4 NT MFMAs = 256 NT cycles: one group of distance hides the L2 latency)
  const int grp_floats = OT * 256;
  auto issue = [&](int S, f32x4 (&av)[NT]) {
    const float* base = wgrp + (long long)(S < NS4 ? S : NS4 - 1) * grp_floats;
#pragma unroll
    for (int k = 0; k < NT; ++k) wide_gload(av[k], lane_off + (unsigned)((t0 + k) * 1024), base);
  };
  auto wait_all = [&](f32x4 (&av)[NT], auto N) {
#pragma unroll
    for (int k = 0; k < NT; k += 2) {
      f32x4 (&pr)[2] = reinterpret_cast<f32x4 (&)[2]>(av[k]);
      wide_vmwait<decltype(N)::value, 2>(pr);
    }
  };
  static_assert(NT % 2 == 0, "tiles are awaited in pairs");
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;
  issue(0, a[0]);
  float b[2][4];
  auto loadB = [&](int S, float (&bv)[4]) {
    const int Sc = S < NS4 ? S : NS4 - 1;
    const float4 cv = *reinterpret_cast<const float4*>(col + 8 * Sc);
    const float* __restrict__ dp = dpl + (8 * Sc) * 32;
    bv[0] = dp[0] * cv.x; bv[1] = dp[64] * cv.y; bv[2] = dp[128] * cv.z; bv[3] = dp[192] * cv.w;
  };
  loadB(0, b[0]);
  for (int S = 0; S < NS4; S += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      issue(S + u + 1, a[(u + 1) % 2]);
      loadB(S + u + 1, b[(u + 1) % 2]);
      wait_all(a[u], std::integral_constant<int, NT>{});  // all but the NT newest loads have landed
      SDEH_FENCE();
      if (S + u < NS4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < NT; ++k) acc[k] = SDEH_MFMA(a[u][k][e], b[u][e], acc[k]);
      }
      SDEH_FENCE();
    }
  }
  wait_all(a[0], std::integral_constant<int, 0>{});  // drain the clamped re-read issued by the last iteration
}

// The same for TWO coordinates at once: one stream of A operands (the packed weights -- each 1 KB operand load costs the issuing wave
// ~35 cycles, tools/ubench/wide_loop.hip) feeds both coordinates' MFMAs, i.e. half the loads per MFMA.  The per-coordinate sequence
// of MFMAs is the one of wide_tangent_pass, so each result is bit-identical to it.
template <int NT>
__device__ __forceinline__ void wide_tangent_pass2(const float* __restrict__ wgrp, int OT, int t0, int NS4, unsigned lane_off,
                                                   const float* __restrict__ dpl, const float* __restrict__ colA,
                                                   const float* __restrict__ colB, f32x16 (&accA)[NT], f32x16 (&accB)[NT]) {
  static_assert(NT == 2, "tiles are awaited in pairs");
  f32x4 a[2][NT];
  const int grp_floats = OT * 256;
  auto issue = [&](int S, f32x4 (&av)[NT]) {
    const float* base = wgrp + (long long)(S < NS4 ? S : NS4 - 1) * grp_floats;
#pragma unroll
    for (int k = 0; k < NT; ++k) wide_gload(av[k], lane_off + (unsigned)((t0 + k) * 1024), base);
  };
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) accA[k][q] = accB[k][q] = 0.0f;
  issue(0, a[0]);
  float bA[2][4], bB[2][4];
  auto loadB = [&](int S, float (&va)[4], float (&vb)[4]) {
    const int Sc = S < NS4 ? S : NS4 - 1;
    const float4 ca = *reinterpret_cast<const float4*>(colA + 8 * Sc);
    const float4 cb = *reinterpret_cast<const float4*>(colB + 8 * Sc);
    const float* __restrict__ dp = dpl + (8 * Sc) * 32;
    const float d0 = dp[0], d1 = dp[64], d2 = dp[128], d3 = dp[192];
    va[0] = d0 * ca.x; va[1] = d1 * ca.y; va[2] = d2 * ca.z; va[3] = d3 * ca.w;
    vb[0] = d0 * cb.x; vb[1] = d1 * cb.y; vb[2] = d2 * cb.z; vb[3] = d3 * cb.w;
  };
  loadB(0, bA[0], bB[0]);
  for (int S = 0; S < NS4; S += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      issue(S + u + 1, a[(u + 1) % 2]);
      loadB(S + u + 1, bA[(u + 1) % 2], bB[(u + 1) % 2]);
      wide_vmwait<NT, 2>(a[u]);  // all but the NT newest loads have landed
      SDEH_FENCE();
      if (S + u < NS4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            accA[k] = SDEH_MFMA(a[u][k][e], bA[u][e], accA[k]);
            accB[k] = SDEH_MFMA(a[u][k][e], bB[u][e], accB[k]);
          }
      }
      SDEH_FENCE();
    }
  }
  wide_vmwait<0, 2>(a[0]);  // drain the clamped re-read issued by the last iteration
}

// GMM: mixture targets (distr/gauss.py:66-140) compiled in.  The divergence leaves no LDS for mixture tables at C = 256, so they are
// read from the workspace (L2: K d floats per trajectory and step next to the d C^2 of the divergence), and the partial logits /
// responsibilities take the act' planes' place between the divergence and the next step's inference pass.
template <int OTW, bool GMM>  // C = 128 OTW
__global__ __launch_bounds__(256) void bridge_wide_kernel(const TrajArgs A, int split, float* __restrict__ divparts) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CT = 1, RS = 32, OT = 4 * OTW, C = 128 * OTW;
  const WsLayout& L = A.lay;
  const WsLayout& L2 = A.lay2;
  const float* __restrict__ ws = A.ws;
  const float* __restrict__ ws2 = A.ws2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int d = A.d, OTD = L.otd;
  const int rows = C > 32 * OTD ? C : 32 * OTD;
  const int tile = (int)blockIdx.x / split, sp = (int)blockIdx.x % split;
  const bool lead = sp == 0;  // the workgroup that owns the tile's outputs (x_T, rnd without the network-divergence part)

  WideCtx cx;
  cx.RS = RS; cx.d = d;
  cx.wave = w; cx.lane = lane; cx.j = j; cx.h = h;
  cx.planes = lds; cx.plane_floats = rows * RS;
  float* dplanes = lds + rows * RS;                       // act'(z_l) of the inference network, l = 0 .. Lh: [Lh + 1][C][32]
  int dpl_floats = (L2.n_hidden + 1) * C * RS;
  if constexpr (GMM) {  // the mixture's partial logits / responsibilities / logsumexp share this region (bridge_wide_lds_bytes)
    const int gmm_floats = A.target.kind == SDEH_DENS_GMM ? 5 * A.target.n_comp * RS + RS : 0;
    dpl_floats = gmm_floats > dpl_floats ? gmm_floats : dpl_floats;
  }
  cx.scr = dplanes + dpl_floats;                          // [kWideSlots][4][32]
  float* divacc = cx.scr + kWideSlots * 4 * RS;           // [4 waves][8 groups][32]: sigma dt mask J_jj accumulated over the steps
  float* cols = divacc + 4 * 8 * RS;                      // [4 waves][2 coordinates][2][C]: column j of W_in / row j of W_out of the coordinates at hand
  float* tabs = cols + 4 * 4 * C;
  const int tab_stride = 2 * L.dp + 4;
  for (int i = tid; i < 3 * tab_stride; i += 256) {
    const int which = i / tab_stride, o = i % tab_stride;
    tabs[i] = o <= 2 * L.dp ? ws[L.dg[which] + o] : 0.0f;
  }
  cx.tab0 = tabs; cx.tab1 = tabs + tab_stride; cx.tab2 = tabs + 2 * tab_stride;
  // biases (hidden layers, then the out layer): read from the workspace (L2) -- the network passes are < 1 % of this kernel, and the
  // LDS they would take is what the second coordinate's columns of the paired divergence pass need at d = 196, C = 256
  const float* bias_u = ws + L.b_hid;    // generative network
  const float* bias_v = ws2 + L2.b_hid;  // inference network
  for (int i = tid; i < 4 * 8 * RS; i += 256) divacc[i] = 0.0f;
  cx.bias = bias_u;
  WideGmm gm{};
  if constexpr (GMM) {
    gm.K = A.target.kind == SDEH_DENS_GMM ? A.target.n_comp : 0;
    gm.d4 = L.gmm_row;
    gm.mu = ws + L.gmm_lg; gm.a = ws + L.gmm_sc; gm.ck = ws + L.gmm_c;
    gm.part = dplanes; gm.resp = dplanes + 4 * gm.K * RS; gm.lse = gm.resp + gm.K * RS;
  }

  const int nto = (OTD > w ? 1 : 0) + (OTD > w + 4 ? 1 : 0);
  const long long row0 = (long long)tile * RS;
  const int flags = A.flags, ctrl_kind = A.ctrl_kind;
  const DensArgs tgt = A.target;
  const bool lv = flags & SDEH_FLAG_CHANGE_SDE_CTRL;
  const bool need_t = ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool inf_lerp = A.inf_kind == SDEH_CTRL_LERP_PRIOR;
  const bool need_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR || inf_lerp;
  const bool vec4 = (d & 3) == 0;
  const int Lh2 = L2.n_hidden;

  f32x16 xr[2][CT];
  const long long r = row0 + j;
  const bool live = r < A.batch;
  const long long lrow = live ? r : A.batch - 1;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int cc = 32 * (w + 4 * k) + rho(q, h);
      xr[k][0][q] = (k < nto && cc < d) ? A.x0[lrow * d + cc] : 0.0f;
      if (A.xs != nullptr && lead && k < nto && cc < d && live && !A.seg_continue) A.xs[lrow * d + cc] = xr[k][0][q];
    }
  __syncthreads();  // tables staged
  wide_publish<CT>(cx, cx.plane(0), xr, nto);
  if (flags & SDEH_FLAG_INIT_LOGP) wide_gauss_quad<CT>(cx, cx.tab2, xr, nto, WSL_LOGP_A);
  __syncthreads();
  float rnd = 0.0f;  // wave 0, lane half 0 of the lead workgroup
  if (flags & SDEH_FLAG_INIT_LOGP) rnd = cx.tab2[2 * L.dp] - 0.5f * wide_slot_sum(cx, WSL_LOGP_A, j);
  if (A.seg_continue) rnd = live ? A.rnd[row0 + j] : 0.0f;  // a later segment of the grid (sdeh_simulate_fwd_steps): rnd carries on
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
  const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);
  unsigned voff_in[OTW];
#pragma unroll
  for (int k = 0; k < OTW; ++k) voff_in[k] = (unsigned)(((w + 4 * k) * 64 + lane) * 16);
  // coordinate groups of this wave: g = gw, gw + TW, ... (gw = global wave index among the TW waves sharing the tile)
  const int TW = 4 * split, gw = sp * 4 + w, ngw = kDivGroups / TW;

  for (int i = A.step0; i < A.step0 + A.n_steps; ++i) {
    cfp cf = as_const(ws + L.coef + i * kCoefStride);
    const float dt = cf[CF_DT], sqdt = cf[CF_SQDT], sig = cf[CF_SIGMA];
    float fs = 0.0f, fx0 = 0.0f, fiv = 0.0f;  // funnel statistics of x_i
    if (need_t && tgt.kind == SDEH_DENS_FUNNEL) {
      fs = wide_slot_sum(cx, WSL_PRESQ, j);
      fx0 = cx.scr[(WSL_X0 * 4) * RS + j];
      fiv = __expf(-fx0);
    }
    // ---- generative network u (plane: x -> ... -> last hidden activation) ------------------------------------------------
    f32x16 nu[2][CT], nv[2][CT];
    {
      WidePre<OTW> pre;
      wide_prefetch<OTW>(pre, ws + L.w_in, OT * 256, L.dp8 / 8, voff_in);
      f32x16 emb[OTW];
#pragma unroll
      for (int k = 0; k < OTW; ++k) emb[k] = load16(ws + L.emb + i * C + ((w + 4 * k) * 2 + h) * 16);
      (void)wide_mlp<OTW, CT, false>(cx, ws, L, A.act, 0, true, nullptr, pre, emb, bias_u, nu, nto);
    }
    wide_barrier();  // every wave is through its out-layer: the plane may take x again
    {
      float* __restrict__ pl = cx.plane(0);
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
#pragma unroll
          for (int q = 0; q < 16; ++q) pl[(32 * (w + 4 * k) + rho(q, h)) * RS + j] = xr[k][0][q];
        }
    }
    wide_barrier();
    // ---- inference network v, keeping act'(z_l) of every layer ----------------------------------------------------------------
    {
      WidePre<OTW> pre;
      wide_prefetch<OTW>(pre, ws2 + L2.w_in, OT * 256, L2.dp8 / 8, voff_in);
      f32x16 emb[OTW];
#pragma unroll
      for (int k = 0; k < OTW; ++k) emb[k] = load16(ws2 + L2.emb + i * C + ((w + 4 * k) * 2 + h) * 16);
      (void)wide_mlp<OTW, CT, true>(cx, ws2, L2, A.inf_act, 0, true, dplanes, pre, emb, bias_v, nv, nto);
    }
    wide_barrier();
    // raw inference-network output (+ bias) -> plane rows [coordinate][trajectory]: the clamp mask of every coordinate
    {
      float* __restrict__ pl = cx.plane(0);
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
          const f32x16 bo = load16(bias_v + L2.n_hidden * C + ((w + 4 * k) * 2 + h) * 16);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            nv[k][0][q] += bo[q];
            pl[(32 * (w + 4 * k) + rho(q, h)) * RS + j] = nv[k][0][q];
          }
        }
    }
    wide_barrier();
    // ---- diagonal of the inference network's Jacobian: this wave's coordinates ------------------------------------------------
    {
      const float sdt = sig * dt;
      float* __restrict__ mycol = cols + w * 4 * C;
      float* __restrict__ mycolB = mycol + 2 * C;  // the second coordinate of a pair
      const float* __restrict__ d0 = dplanes + h * RS + j;
      const unsigned lane_off = (unsigned)(lane * 16);
      for (int gi = 0; gi < ngw; ++gi) {
        const int g = gw + gi * TW;
        float part = 0.0f;
        int jc = g;
        // Coordinates of the group two at a time (hidden layers present): one operand stream for both.  Each coordinate's own sums
        // keep their order, so the result does not change with the pairing.
        for (; Lh2 >= 1 && jc + kDivGroups < d; jc += 2 * kDivGroups) {
          const int jb = jc + kDivGroups;
          {
            const float4 ci = *reinterpret_cast<const float4*>(ws2 + L2.tan_in + jc * C + lane * 4);
            const float4 co = *reinterpret_cast<const float4*>(ws2 + L2.tan_out + jc * C + lane * 4);
            const float4 di = *reinterpret_cast<const float4*>(ws2 + L2.tan_in + jb * C + lane * 4);
            const float4 dq = *reinterpret_cast<const float4*>(ws2 + L2.tan_out + jb * C + lane * 4);
            if (lane * 4 < C) {
              *reinterpret_cast<float4*>(mycol + lane * 4) = ci;
              *reinterpret_cast<float4*>(mycol + C + lane * 4) = co;
              *reinterpret_cast<float4*>(mycolB + lane * 4) = di;
              *reinterpret_cast<float4*>(mycolB + C + lane * 4) = dq;
            }
          }
          float jjA = 0.0f, jjB = 0.0f;
          auto idx = [&](int ot, int q) { return (4 * ot + (q >> 2)) * 8 + (q & 1) * 4 + ((q >> 1) & 1) + 2 * h; };  // B order of channel 32 ot + rho(q, h)
          const float* __restrict__ d1 = dplanes + C * RS;
#pragma unroll
          for (int part_i = 0; part_i < OT / 2; ++part_i) {
            f32x16 FA[2], FB[2];
            wide_tangent_pass2<2>(ws2 + L2.w_hid, OT, 2 * part_i, C / 8, lane_off, d0, mycol + 4 * h, mycolB + 4 * h, FA, FB);
            if (Lh2 == 1) {
#pragma unroll
              for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                  const int ot = 2 * part_i + k;
                  const float dv = d1[(32 * ot + rho(q, h)) * RS + j];
                  jjA = fmaf(FA[k][q] * dv, mycol[C + idx(ot, q)], jjA);
                  jjB = fmaf(FB[k][q] * dv, mycolB[C + idx(ot, q)], jjB);
                }
            } else {
              f32x16 GA[2], GB[2];  // adjoints through the second hidden layer
              wide_tangent_pass2<2>(ws2 + L2.wt_hid + L2.w_hid_stride, OT, 2 * part_i, C / 8, lane_off, d0 + 2 * C * RS,
                                    mycol + C + 4 * h, mycolB + C + 4 * h, GA, GB);
#pragma unroll
              for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                  const int ot = 2 * part_i + k;
                  const float dv = d1[(32 * ot + rho(q, h)) * RS + j];
                  jjA = fmaf(FA[k][q] * dv, GA[k][q], jjA);
                  jjB = fmaf(FB[k][q] * dv, GB[k][q], jjB);
                }
            }
          }
          jjA = half_sum(jjA);
          jjB = half_sum(jjB);
          // d clip(v_j, -m, m) / d v_j = 1 on [-m, m] (torch.clamp's backward), else 0
          const float vA = cx.plane(0)[jc * RS + j], vB = cx.plane(0)[jb * RS + j];
          part += (vA >= -A.inf_clip_model && vA <= A.inf_clip_model) ? jjA : 0.0f;
          part += (vB >= -A.inf_clip_model && vB <= A.inf_clip_model) ? jjB : 0.0f;
        }
        for (; jc < d; jc += kDivGroups) {
          // column jc of W_in and row jc of W_out (B order) -> this wave's LDS rows
          {
            const float4 ci = *reinterpret_cast<const float4*>(ws2 + L2.tan_in + jc * C + lane * 4);
            const float4 co = *reinterpret_cast<const float4*>(ws2 + L2.tan_out + jc * C + lane * 4);
            if (lane * 4 < C) {
              *reinterpret_cast<float4*>(mycol + lane * 4) = ci;
              *reinterpret_cast<float4*>(mycol + C + lane * 4) = co;
            }
          }
          float jj = 0.0f;
          auto idx = [&](int ot, int q) { return (4 * ot + (q >> 2)) * 8 + (q & 1) * 4 + ((q >> 1) & 1) + 2 * h; };  // B order of channel 32 ot + rho(q, h)
          if (Lh2 == 0) {
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
              for (int q = 0; q < 16; ++q)
                jj = fmaf(mycol[idx(ot, q)] * dplanes[(32 * ot + rho(q, h)) * RS + j], mycol[C + idx(ot, q)], jj);
          } else {
            // Channel by channel F meets only its own D and G: the row tiles are processed in two halves, so that the
            // accumulators of F and G together stay at half of the wave's AGPRs
            const float* __restrict__ d1 = dplanes + C * RS;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              constexpr int NH = OT / 2;
              f32x16 F[NH];
              wide_tangent_pass<NH>(ws2 + L2.w_hid, OT, half * NH, C / 8, lane_off, d0, mycol + 4 * h, F);
              if (Lh2 == 1) {
#pragma unroll
                for (int k = 0; k < NH; ++k)
#pragma unroll
                  for (int q = 0; q < 16; ++q) {
                    const int ot = half * NH + k;
                    jj = fmaf(F[k][q] * d1[(32 * ot + rho(q, h)) * RS + j], mycol[C + idx(ot, q)], jj);
                  }
              } else {
                f32x16 G[NH];  // adjoint through the second hidden layer
                wide_tangent_pass<NH>(ws2 + L2.wt_hid + L2.w_hid_stride, OT, half * NH, C / 8, lane_off, d0 + 2 * C * RS,
                                      mycol + C + 4 * h, G);
#pragma unroll
                for (int k = 0; k < NH; ++k)
#pragma unroll
                  for (int q = 0; q < 16; ++q) {
                    const int ot = half * NH + k;
                    jj = fmaf(F[k][q] * d1[(32 * ot + rho(q, h)) * RS + j], G[k][q], jj);
                  }
              }
            }
          }
          jj = half_sum(jj);
          // d clip(v_j, -m, m) / d v_j = 1 on [-m, m] (torch.clamp's backward), else 0
          const float vj = cx.plane(0)[jc * RS + j];
          part += (vj >= -A.inf_clip_model && vj <= A.inf_clip_model) ? jj : 0.0f;
        }
        if (h == 0) divacc[(w * 8 + gi) * RS + j] = fmaf(sdt, part, divacc[(w * 8 + gi) * RS + j]);
      }
    }
    // ---- elementwise part on this wave's coordinates -----------------------------------------------------------------------------
    const float c_x = fmaf(cf[CF_DRIFT], dt, 1.0f), c_u = sig * dt, c_n = sig * sqdt;
    const float wl = cf[CF_W];
    int hv = h;
    asm volatile("" : "+v"(hv));
    WideScore sq;
    sq.ctrl_kind = ctrl_kind; sq.g = L.g; sq.need_t = need_t; sq.need_p = need_p; sq.tgt = tgt; sq.wl = wl;
    sq.mult = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig;
    sq.scale_score = A.scale_score; sq.clip_score = A.clip_score; sq.g0 = as_const(ws + L.gam + i * L.g)[0]; sq.d = d; sq.gmm = &gm;
    if (tgt.kind == SDEH_DENS_EXTERNAL && need_t) {
      sq.ext = A.ext_score + (long long)(i - A.step0) * A.ext_stride;
      sq.ext_off[0] = lrow * d;
    }
    if constexpr (GMM) {
      if (need_t && tgt.kind == SDEH_DENS_GMM) {  // responsibilities of x_i (the act' planes are free: every wave is through its coordinates)
        wide_barrier();
        wide_gmm_partials<CT>(cx, gm, xr, nto);
        wide_barrier();
        if (w == 0) wide_gmm_normalise<CT>(cx, gm);
        wide_barrier();
      }
    }
    const float g20 = as_const(ws2 + L2.gam + i * L2.g)[0];
    float costl = 0.0f, itol = 0.0f, divs = 0.0f;
    auto vtile = [&](f32x16& x, const f32x16& nuv, const f32x16& nvv, int t) {
      const int cb = 32 * t + 4 * hv;
      auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
      float sterm[16], psc[16];
      if constexpr (GMM) {  // training forward: the combined score entering the control, row-major [T, B, d] (as traj_wide_kernel)
        float scr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) scr[q] = 0.0f;
        wide_score_term16<GMM>(sq, cx, x, cb, 0, fs, fx0, fiv, ws + L.gam + i * L.g, sterm, psc, A.sc_out != nullptr ? scr : nullptr);
        if (A.sc_out != nullptr && live && lead) {
          float* __restrict__ sp = A.sc_out + ((long long)i * A.batch + lrow) * d + cb;
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (coord(q) < d) sp[(q & 3) + 8 * (q >> 2)] = scr[q];
        }
      } else {
        wide_score_term16(sq, cx, x, cb, 0, fs, fx0, fiv, ws + L.gam + i * L.g, sterm, psc);
      }
      SDEH_FENCE();
      float n[16];
      wide_noise16(A.noise != nullptr ? A.noise + ((long long)i * A.batch + lrow) * d : nullptr, vec4, cb, d, A.seed, rng_off, grow, i, n);
      const f32x16 bo = load16(bias_u + L.n_hidden * C + (t * 2 + hv) * 16);
      float u[16], v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float uq = clipf(nuv[q] + bo[q], A.clip_model) + sterm[q];
        u[q] = coord(q) < d ? uq : 0.0f;
      }
      if (inf_lerp) {  // LerpPriorCtrl (reparam.py:165-178,149-162): v += sigma [scale clip((1 - t/T) prior_score(x)) gamma(t)]
        const float w1 = 1.0f - wl;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float gq = L2.g == 1 ? g20 : ws2[L2.gam + i * L2.g + coord(q)];
          const float sc = w1 * psc[q];
          const bool inside = sc >= -A.inf_clip_score && sc <= A.inf_clip_score;
          const float vq = clipf(nvv[q], A.inf_clip_model) + sig * ((A.inf_scale_score * clipf(sc, A.inf_clip_score)) * gq);
          const float dsc = inside ? -(w1 * cx.tab1[2 * coord(q) + 1]) : 0.0f;
          const bool valid = coord(q) < d;
          divs = valid ? fmaf(sig * A.inf_scale_score * gq, dsc, divs) : divs;
          v[q] = valid ? vq : 0.0f;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = coord(q) < d ? clipf(nvv[q], A.inf_clip_model) : 0.0f;
      }
      // running cost on gen_plus_inf = u + v, gen_minus_inf = u - v (losses/oc.py:201-211); Ito term on u + v
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float gp = u[q] + v[q];
        costl = lv ? fmaf(gp, u[q] - 0.5f * (u[q] - v[q]), costl) : fmaf(gp, gp, costl);
        itol = fmaf(gp, n[q], itol);
        const float xn = fmaf(c_n, n[q], fmaf(c_u, u[q], c_x * x[q]));
        x[q] = coord(q) < d ? xn : 0.0f;
      }
      if (A.xs != nullptr && live && lead) {
        float* __restrict__ xp = A.xs + ((long long)(i + 1) * A.batch + lrow) * d + cb;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (coord(q) < d) xp[(q & 3) + 8 * (q >> 2)] = x[q];
      }
      if (A.gp != nullptr && live && lead) {  // training: u + v of this step (the inference network's upstream gradient)
        float* __restrict__ gq = A.gp + ((long long)i * A.batch + lrow) * d + cb;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (coord(q) < d) gq[(q & 3) + 8 * (q >> 2)] = u[q] + v[q];
      }
      SDEH_FENCE();
    };
    if (nto > 0) vtile(xr[0][0], nu[0][0], nv[0][0], w);
    if (nto > 1) vtile(xr[1][0], nu[1][0], nv[1][0], w + 4);
    {
      const float cs = half_sum(costl), is = half_sum(itol), ds = half_sum(divs);
      if (h == 0) {
        cx.scr[(WSL_COST * 4 + w) * RS + j] = cs;
        cx.scr[(WSL_ITO * 4 + w) * RS + j] = is;
        cx.scr[(WSL_DIV * 4 + w) * RS + j] = ds;
      }
    }
    wide_barrier();  // everyone has read the clamp-mask rows of the plane
    wide_publish<CT>(cx, cx.plane(0), xr, nto);
    wide_barrier();
    if (w == 0 && lead) {
      float cost = wide_slot_sum(cx, WSL_COST, j);
      if (!lv) cost *= 0.5f;
      rnd = fmaf(sig * wide_slot_sum(cx, WSL_DIV, j), dt, rnd);  // score part of the divergence (losses/oc.py:199-200)
      rnd = fmaf(cost, dt, rnd);
      if (!(flags & SDEH_FLAG_TRAIN)) rnd -= cf[CF_DDIV];
      if (flags & SDEH_FLAG_ITO) rnd = fmaf(wide_slot_sum(cx, WSL_ITO, j), sqdt, rnd);
    }
  }

  // ---- network part of the divergence: this wave's group sums -> divparts[tile][group][trajectory] ---------------------------
  if (h == 0)
    for (int gi = 0; gi < ngw; ++gi) divparts[((long long)tile * kDivGroups + gw + gi * TW) * RS + j] = divacc[(w * 8 + gi) * RS + j];
  if (!lead) return;
  if (flags & SDEH_FLAG_TERMINAL_TARGET) {
    if (tgt.kind == SDEH_DENS_DIAG_GAUSS) wide_gauss_quad<CT>(cx, cx.tab0, xr, nto, WSL_LOGP_B);
    else if (tgt.kind == SDEH_DENS_MULTI_WELL) wide_mwell_sum<CT>(cx, tgt, xr, nto, WSL_LOGP_B);
    if constexpr (GMM) {
      if (tgt.kind == SDEH_DENS_GMM) {  // log-density (and responsibilities) of x_T
        wide_gmm_partials<CT>(cx, gm, xr, nto);
        __syncthreads();
        if (w == 0) wide_gmm_normalise<CT>(cx, gm);
      }
    }
  }
  __syncthreads();
  if (w == 0 && h == 0) {
    float rr = rnd;
    if (flags & SDEH_FLAG_TERMINAL_TARGET) {
      float lp = 0.0f;
      if (tgt.kind == SDEH_DENS_DIAG_GAUSS) lp = cx.tab0[2 * L.dp] - 0.5f * wide_slot_sum(cx, WSL_LOGP_B, j) + tgt.lnc;
      else if (tgt.kind == SDEH_DENS_MULTI_WELL) lp = -wide_slot_sum(cx, WSL_LOGP_B, j);
      else if (tgt.kind == SDEH_DENS_FUNNEL) {
        const float x0v = cx.scr[(WSL_X0 * 4) * RS + j], sqs = wide_slot_sum(cx, WSL_PRESQ, j);
        const float first = -0.5f * __logf(6.283185307179586f * tgt.p0) - 0.5f * x0v * x0v / tgt.p0;
        const float other = -(float)(d - 1) * (x0v + 1.8378770664093453f) * 0.5f - 0.5f * sqs * __expf(-x0v);
        lp = first + other + tgt.lnc;
      } else if (GMM && tgt.kind == SDEH_DENS_GMM) {
        lp = gm.lse[j] + tgt.lnc;
      }
      rr -= clipf(lp, A.clip_target);
    }
    if (live) A.rnd[row0 + j] = rr;
  }
  if constexpr (GMM) {
    // training forward (method kl): 1[|log rho(x_T)| <= clip_target] target.score(x_T), row-major [B, d]
    if (A.tsc_out != nullptr && tgt.kind == SDEH_DENS_GMM && (flags & SDEH_FLAG_TERMINAL_TARGET)) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
          const int cb = 32 * (w + 4 * k) + 4 * h;
          float sc[16];
          wide_gmm_score16(cx, gm, xr[k][0], cb, j, sc);
          const float lp = gm.lse[j] + tgt.lnc;
          const float keep = fabsf(lp) <= A.clip_target ? 1.0f : 0.0f;
          if (live) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int cc = cb + (q & 3) + 8 * (q >> 2);
              if (cc < d) A.tsc_out[lrow * d + cc] = keep * sc[q];
            }
          }
        }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int cc = 32 * (w + 4 * k) + rho(q, h);
      if (k < nto && cc < d && live) A.xT[lrow * d + cc] = xr[k][0][q];
    }
}

// rnd[row] += the 32 group sums of sigma dt div_x(network part), in fixed order (independent of `split`)
__global__ void bridge_wide_finish(float* __restrict__ rnd, const float* __restrict__ divparts, long long batch) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= batch) return;
  const float* p = divparts + (row >> 5) * kDivGroups * 32 + (row & 31);
  float s = 0.0f;
  for (int g = 0; g < kDivGroups; ++g) s += p[g * 32];
  rnd[row] += s;
}

inline size_t bridge_wide_lds_bytes(const WsLayout& L, const WsLayout& L2, int K) {
  const int rows = L.c > 32 * L.otd ? L.c : 32 * L.otd;
  size_t dpl = (size_t)(L2.n_hidden + 1) * L.c * 32;
  const size_t gmm = K > 0 ? (size_t)5 * K * 32 + 32 : 0;  // partial logits + responsibilities + logsumexp share the act' planes
  if (gmm > dpl) dpl = gmm;
  return ((size_t)rows * 32 + dpl + kWideSlots * 4 * 32 + 4 * 8 * 32 + 4 * 4 * L.c + 3 * (2 * L.dp + 4)) * sizeof(float);
}

long long bridge_wide_scratch_floats(long long batch) { return ((batch + 31) / 32) * kDivGroups * 32; }

template <int OTW, bool GMM>
static int launch_bridge_wide_t(const TrajArgs& a, hipStream_t stream, int split, float* scratch) {
  const size_t lds_bytes = bridge_wide_lds_bytes(a.lay, a.lay2, GMM ? a.target.n_comp : 0);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_wide_kernel<OTW, GMM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  const long long tiles = (a.batch + 31) / 32;
  hipLaunchKernelGGL((bridge_wide_kernel<OTW, GMM>), dim3((unsigned)(tiles * split)), dim3(256), lds_bytes, stream, a, split, scratch);
  hipLaunchKernelGGL(bridge_wide_finish, dim3((unsigned)((a.batch + 255) / 256)), dim3(256), 0, stream, a.rnd, scratch, a.batch);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// split: workgroups per column tile of 32 trajectories (1, 2, 4, 8): enough of them to occupy the 256 CUs
int launch_bridge_wide(const TrajArgs& a, hipStream_t stream, int* split_used, float* scratch) {
  const char* force = plan_opt(OPT_WIDE_SPLIT);  // testing aid: "1" | "2" | "4" | "8"
  const long long tiles = (a.batch + 31) / 32;
  int split = 1;
  while (split < 8 && tiles * split < 256) split *= 2;
  if (force != nullptr && (force[0] == '1' || force[0] == '2' || force[0] == '4' || force[0] == '8')) split = force[0] - '0';
  if (split_used != nullptr) *split_used = split;
  const int otw = a.lay.c / 128;
  if (a.target.kind == SDEH_DENS_GMM && a.target.n_comp > 0) {
    if (otw == 2) return launch_bridge_wide_t<2, true>(a, stream, split, scratch);
    if (otw == 1) return launch_bridge_wide_t<1, true>(a, stream, split, scratch);
    return SDEH_ERR_UNSUPPORTED;
  }
  if (otw == 2) return launch_bridge_wide_t<2, false>(a, stream, split, scratch);
  if (otw == 1) return launch_bridge_wide_t<1, false>(a, stream, split, scratch);
  return SDEH_ERR_UNSUPPORTED;
}

}  // namespace sdeh
