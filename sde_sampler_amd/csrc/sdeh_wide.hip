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
#pragma clang fp contract(off)
#include "sdeh_bwd.hpp"

namespace sdeh {

constexpr int kWideSlots = 8;
enum WideSlot { WSL_COST = 0, WSL_ITO = 1, WSL_PRESQ = 2, WSL_X0 = 3, WSL_LOGP_A = 4, WSL_LOGP_B = 5, WSL_DIV = 6 };

// ---------------------------------------------------------------------------------------------------------
// acc[k][c] += W[tile t0 + 4 k][:] . act[:, column tile c]  over NS4 groups of four k-steps.
//   wp   : packed weights of the layer + (t0 * 64 + lane) float4 (tiles of this wave are 4 apart; ntot tiles per group)
//   actl : the input plane + h * RS + j
// A operands are prefetched two groups ahead (global / L2 latency), B operands one group ahead (LDS latency).
// ---------------------------------------------------------------------------------------------------------
template <int NT, int CT>
__device__ __forceinline__ void wide_layer(const float4* __restrict__ wp, int ntot, int NS4, const float* __restrict__ actl,
                                           int RS, f32x16 (&acc)[NT][CT]) {
  float4 a0[NT], a1[NT], a2[NT];
  float b0[4][CT], b1[4][CT], b2[4][CT];
  auto loadA = [&](int S, float4 (&a)[NT]) {
#pragma unroll
    for (int k = 0; k < NT; ++k) a[k] = wp[((long long)S * ntot + 4 * k) * 64];
  };
  auto loadB = [&](int S, float (&b)[4][CT]) {
    const float* __restrict__ ap = actl + (8 * S) * RS;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CT; ++c) b[e][c] = ap[2 * e * RS + 32 * c];
  };
  auto compute = [&](const float4 (&a)[NT], const float (&b)[4][CT]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const float av = e == 0 ? a[k].x : (e == 1 ? a[k].y : (e == 2 ? a[k].z : a[k].w));
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[k][c] = SDEH_MFMA(av, b[e][c], acc[k][c]);
      }
  };
  loadA(0, a0);
  loadB(0, b0);
  if (NS4 > 1) loadA(1, a1);
  for (int S = 0; S < NS4; S += 3) {
    if (S + 2 < NS4) loadA(S + 2, a2);
    if (S + 1 < NS4) loadB(S + 1, b1);
    SDEH_FENCE();
    compute(a0, b0);
    SDEH_FENCE();
    if (S + 1 < NS4) {
      if (S + 3 < NS4) loadA(S + 3, a0);
      if (S + 2 < NS4) loadB(S + 2, b2);
      SDEH_FENCE();
      compute(a1, b1);
      SDEH_FENCE();
      if (S + 2 < NS4) {
        if (S + 4 < NS4) loadA(S + 4, a1);
        if (S + 3 < NS4) loadB(S + 3, b0);
        SDEH_FENCE();
        compute(a2, b2);
        SDEH_FENCE();
      }
    }
  }
}

// accumulators <- a [tiles][(ot * 2 + h) * 16 + q] table in accumulator order (time embedding + input bias, layer biases)
template <int NT, int CT>
__device__ __forceinline__ void wide_init(const float* __restrict__ tab, int t0, int h, f32x16 (&acc)[NT][CT]) {
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const f32x16 v = load16(tab + ((t0 + 4 * k) * 2 + h) * 16);
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[k][c] = v;
  }
}

// out[channel][trajectory] <- act(acc) for this wave's tiles; DSTORE: also act'(acc) into `dout` (Bridge: the inference network)
template <int NT, int CT, bool DSTORE>
__device__ __forceinline__ void wide_act_store(f32x16 (&acc)[NT][CT], int act, float* __restrict__ outl, float* __restrict__ doutl,
                                               int RS, int t0, int h) {
  SDEH_ACT_SWITCH(act, ACT,
    _Pragma("unroll") for (int k = 0; k < NT; ++k)
      _Pragma("unroll") for (int c = 0; c < CT; ++c)
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
          const int o = (32 * (t0 + 4 * k) + rho(q, h)) * RS + 32 * c;
          if constexpr (DSTORE) doutl[o] = act_grad(acc[k][c][q], ACT);
          outl[o] = act_ct<ACT>(acc[k][c][q]);
        });
}

__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32); }

struct WideCtx {
  float* planes;       // LDS: plane p = planes + p * plane_floats ([channel | coordinate][trajectory])
  int plane_floats;
  float* scr;          // [kWideSlots][4][RS]
  const float* tab0;   // LDS copies of the Gaussian tables (target / prior / second): (mu, 1/sigma^2) per coordinate, then the constant
  const float* tab1;
  const float* tab2;
  int RS, d;
  int wave, lane, j, h;
  __device__ __forceinline__ float* plane(int p) const { return planes + p * plane_floats; }
};

// One FourierMLP pass over the workgroup's CT column tiles (models/mlp.py:114-122): input plane cx.plane(p) (the state, rows =
// coordinates) -> out-layer accumulators of this wave's coordinate tiles (before the clamp).  Returns the plane index the LAST
// hidden activation was written to (the out-layer's input); the state may be published into the other one.
// single: only buf[0] exists (Bridge): every layer reads, barrier, writes in place, barrier.
template <int OTW, int CT, bool DSTORE>
__device__ __forceinline__ int wide_mlp(const WideCtx& cx, const float* __restrict__ ws, const WsLayout& L, int act, int step, int p,
                                        bool single, float* __restrict__ dplanes, f32x16 (&out)[2][CT], int nto) {
  const int RS = cx.RS, C = L.c, OT = L.ot, w = cx.wave, h = cx.h, j = cx.j;
  const int plane = C * RS;  // floats per act' plane
  f32x16 acc[OTW][CT];
  wide_init<OTW, CT>(ws + L.emb + step * C, w, h, acc);
  const float4* wbase = reinterpret_cast<const float4*>(ws) + cx.lane;
  wide_layer<OTW, CT>(wbase + L.w_in / 4 + w * 64, OT, L.dp8 / 8, cx.plane(p) + h * RS + j, RS, acc);
  int q = single ? 0 : 1 - p;
  for (int l = 0; l <= L.n_hidden; ++l) {
    if (single) __syncthreads();  // everyone has read the plane that is about to be overwritten
    wide_act_store<OTW, CT, DSTORE>(acc, act, cx.plane(q) + j, DSTORE ? dplanes + l * plane + j : nullptr, RS, w, h);
    __syncthreads();
    if (l == L.n_hidden) break;
    wide_init<OTW, CT>(ws + L.b_hid + l * C, w, h, acc);
    wide_layer<OTW, CT>(wbase + (L.w_hid + l * L.w_hid_stride) / 4 + w * 64, OT, C / 8, cx.plane(q) + h * RS + j, RS, acc);
    if (!single) q = 1 - q;
  }
  // out_layer on this wave's coordinate tiles {w, w + 4}
  if (nto == 2) {
    wide_init<2, CT>(ws + L.b_out, w, h, out);
    wide_layer<2, CT>(wbase + L.w_out / 4 + w * 64, L.otd, C / 8, cx.plane(q) + h * RS + j, RS, out);
  } else if (nto == 1) {
    f32x16 o1[1][CT];
    wide_init<1, CT>(ws + L.b_out, w, h, o1);
    wide_layer<1, CT>(wbase + L.w_out / 4 + w * 64, L.otd, C / 8, cx.plane(q) + h * RS + j, RS, o1);
#pragma unroll
    for (int c = 0; c < CT; ++c) out[0][c] = o1[0][c];
  }
  return q;
}

// state registers -> plane rows (coordinates of this wave's tiles; padded coordinates are zero), plus the per-trajectory
// statistics of the new state that the NEXT step's scores need before any wave may have overwritten the plane: slot WSL_PRESQ =
// sum over this wave's coordinates c >= 1 of x_c^2, slot WSL_X0 = x_0 (funnel: distr/funnel.py:54-80).
template <int CT>
__device__ __forceinline__ void wide_publish(const WideCtx& cx, float* __restrict__ plane, const f32x16 (&xr)[2][CT], int nto) {
  const int RS = cx.RS, w = cx.wave, h = cx.h, j = cx.j;
  float sq[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) sq[c] = 0.0f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k < nto) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float v = xr[k][c][q];
          plane[(32 * (w + 4 * k) + rho(q, h)) * RS + 32 * c + j] = v;
          const bool first = k == 0 && q == 0;  // coordinate 0 sits in wave 0, tile 0, register 0, lane half 0
          sq[c] = (first && w == 0 && h == 0) ? sq[c] : fmaf(v, v, sq[c]);
        }
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const float s = half_sum(sq[c]);
    if (h == 0) {
      cx.scr[(WSL_PRESQ * 4 + w) * RS + 32 * c + j] = s;
      if (w == 0) cx.scr[(WSL_X0 * 4) * RS + 32 * c + j] = xr[0][c][0];
    }
  }
}

__device__ __forceinline__ float wide_slot_sum(const WideCtx& cx, int slot, int col) {
  const float* s = cx.scr + slot * 4 * cx.RS + col;
  return ((s[0] + s[cx.RS]) + s[2 * cx.RS]) + s[3 * cx.RS];
}

// partial sums over this wave's coordinates of the Gaussian quadratic form sum_c (x_c - mu_c)^2 / sigma_c^2 (table `tab`)
template <int CT>
__device__ __forceinline__ void wide_gauss_quad(const WideCtx& cx, const float* tab, const f32x16 (&xr)[2][CT], int nto, int slot) {
  const int w = cx.wave, h = cx.h;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < nto) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int cc = 32 * (w + 4 * k) + rho(q, h);
          const float2 p = *reinterpret_cast<const float2*>(tab + 2 * cc);  // padded coordinates: (0, 0)
          const float t = xr[k][c][q] - p.x;
          acc = fmaf(t * t, p.y, acc);
        }
      }
    acc = half_sum(acc);
    if (h == 0) cx.scr[(slot * 4 + w) * cx.RS + 32 * c + cx.j] = acc;
  }
}

// partial sums of -log density for the double-well family (distr/double_well.py:39-41,165-172)
template <int CT>
__device__ __forceinline__ void wide_mwell_sum(const WideCtx& cx, const DensArgs& D, const f32x16 (&xr)[2][CT], int nto, int slot) {
  const int w = cx.wave, h = cx.h;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < nto) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int cc = 32 * (w + 4 * k) + rho(q, h);
          const float y = xr[k][c][q] - D.p1;
          const float ww = y * y - D.p0;
          const float v = cc < D.n_comp ? ww * ww : 0.5f * y * y;
          acc += cc < cx.d ? v : 0.0f;
        }
      }
    acc = half_sum(acc);
    if (h == 0) cx.scr[(slot * 4 + w) * cx.RS + 32 * c + cx.j] = acc;
  }
}

// per-coordinate target score (closed forms; the funnel's per-trajectory statistics come in through fs / fx0 / fiv)
__device__ __forceinline__ float wide_target_score(const DensArgs& D, const float* tab, int cc, int d, float x, float fs, float fx0,
                                                   float fiv) {
  switch (D.kind) {
    case SDEH_DENS_DIAG_GAUSS: {
      const float2 p = *reinterpret_cast<const float2*>(tab + 2 * cc);
      return (p.x - x) * p.y;
    }
    case SDEH_DENS_MULTI_WELL: {
      const float y = x - D.p1;
      return cc < D.n_comp ? -4.0f * (y * y - D.p0) * y : -y;
    }
    case SDEH_DENS_FUNNEL:
      return cc == 0 ? -fx0 / D.p0 - 0.5f * (float)(d - 1) + 0.5f * fs * fiv : -x * fiv;
    default: return 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// the kernel (no inference control)
// ---------------------------------------------------------------------------------------------------------
template <int OTW, int CT>
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
  cx.scr = lds + 2 * rows * RS;
  float* tabs = cx.scr + kWideSlots * 4 * RS;
  const int tab_stride = 2 * L.dp + 4;
  for (int i = tid; i < 3 * tab_stride; i += 256) {
    const int which = i / tab_stride, o = i % tab_stride;
    tabs[i] = o <= 2 * L.dp ? ws[L.dg[which] + o] : 0.0f;
  }
  cx.tab0 = tabs; cx.tab1 = tabs + tab_stride; cx.tab2 = tabs + 2 * tab_stride;

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
  if (A.xs != nullptr) {
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
  if (flags & SDEH_FLAG_INIT_LOGP) wide_gauss_quad<CT>(cx, cx.tab2, xr, nto, WSL_LOGP_A);
  __syncthreads();  // x_0 published
  float rnd[CT];  // owned by wave 0, lanes of half 0
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    rnd[c] = 0.0f;
    // Distribution.log_prob = unnorm_log_prob - log_norm_const (distr/base.py:116-119): the constants cancel
    if (flags & SDEH_FLAG_INIT_LOGP) rnd[c] = cx.tab2[2 * L.dp] - 0.5f * wide_slot_sum(cx, WSL_LOGP_A, 32 * c + j);
  }
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);

  for (int i = 0; i < A.n_steps; ++i) {
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
    const int pout = wide_mlp<OTW, CT, false>(cx, ws, L, act, i, p, false, nullptr, nn, nto);
    p = 1 - pout;  // the plane no wave reads any more: the new state goes there

    // ---- elementwise part on this wave's coordinates (reparam.py controls, oc.py cost / update) ---------------------------
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
    auto tile = [&](auto KK) {
      constexpr int k = decltype(KK)::value;
      {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const unsigned long long grow = (unsigned long long)(A.row_offset + lrow[c]);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int cbase = 32 * (w + 4 * k) + 8 * g4 + 4 * hv;  // four consecutive coordinates = Philox block cbase / 4
            float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cbase < d) {
              if (A.noise != nullptr) {
                const float* __restrict__ np = A.noise + ((long long)i * A.batch + lrow[c]) * d + cbase;
                if (vec4) {
                  const float4 t4 = *reinterpret_cast<const float4*>(np);
                  n[0] = t4.x; n[1] = t4.y; n[2] = t4.z; n[3] = t4.w;
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) n[e] = cbase + e < d ? np[e] : 0.0f;
                }
              } else {
                box_muller4(philox_block(A.seed, rng_off, grow, i, cbase >> 2), n);
              }
            }
            float xo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = 4 * g4 + e, cc = cbase + e;
              const bool valid = cc < d;
              const float x = xr[k][c][q];
              float sterm = 0.0f, psc = 0.0f;
              if (need_p) {
                const float2 pp = *reinterpret_cast<const float2*>(cx.tab1 + 2 * cc);
                psc = (pp.x - x) * pp.y;
              }
              if (ctrl_kind != SDEH_CTRL_CLIPPED) {
                const float tsc = need_t ? wide_target_score(tgt, cx.tab0, cc, d, x, fs[c], fx0[c], fiv[c]) : 0.0f;
                float sc;
                if (ctrl_kind == SDEH_CTRL_SCORE) sc = tsc;
                else if (ctrl_kind == SDEH_CTRL_LERP) sc = wl < 0.5f ? psc + wl * (tsc - psc) : tsc - (tsc - psc) * (1.0f - wl);
                else if (ctrl_kind == SDEH_CTRL_LERP_TARGET) sc = wl * tsc;
                else sc = (1.0f - wl) * psc;
                const float gq = L.g == 1 ? g0 : ws[L.gam + i * L.g + cc];
                sterm = mult * ((A.scale_score * clipf(sc, A.clip_score)) * gq);
              }
              const float u = valid ? clipf(nn[k][c][q], A.clip_model) + sterm : 0.0f;
              const float rs = refc ? sig * psc : 0.0f;  // reference_ctrl = sigma prior.score (solver/oc.py:305-306)
              const float gm = valid ? u - rs : 0.0f;
              if (!refc) costl[c] = fmaf(u, u, costl[c]);
              else if (lv) costl[c] = fmaf(gm, u - 0.5f * (rs + u), costl[c]);
              else costl[c] = fmaf(gm, gm, costl[c]);
              itol[c] = fmaf(gm, n[e], itol[c]);
              const float xn = valid ? fmaf(c_u, u, fmaf(c_n, n[e], c_x * x)) : 0.0f;
              xr[k][c][q] = xn;
              xo[e] = xn;
            }
            if (A.xs != nullptr && live[c] && cbase < d) {
              float* __restrict__ xp = A.xs + ((long long)(i + 1) * A.batch + lrow[c]) * d + cbase;
              if (vec4) *reinterpret_cast<float4*>(xp) = float4{xo[0], xo[1], xo[2], xo[3]};
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (cbase + e < d) xp[e] = xo[e];
              }
            }
            SDEH_FENCE();
          }
        }
      }
    };
    if (nto > 0) tile(std::integral_constant<int, 0>{});
    if (nto > 1) tile(std::integral_constant<int, 1>{});
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float cs = half_sum(costl[c]), is = half_sum(itol[c]);
      if (h == 0) {
        cx.scr[(WSL_COST * 4 + w) * RS + 32 * c + j] = cs;
        cx.scr[(WSL_ITO * 4 + w) * RS + 32 * c + j] = is;
      }
    }
    wide_publish<CT>(cx, cx.plane(p), xr, nto);
    __syncthreads();  // x_{i+1}, its statistics and this step's cost partials are visible
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
}

inline size_t wide_lds_bytes(const WsLayout& L, int ct, int n_planes) {
  const int rows = L.c > 32 * L.otd ? L.c : 32 * L.otd;
  return ((size_t)n_planes * rows * 32 * ct + (size_t)kWideSlots * 4 * 32 * ct + 3 * (2 * L.dp + 4)) * sizeof(float);
}

template <int OTW, int CT>
static int launch_wide_t(const TrajArgs& a, hipStream_t stream) {
  const size_t lds_bytes = wide_lds_bytes(a.lay, CT, 2);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_wide_kernel<OTW, CT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  const unsigned grid = (unsigned)((a.batch + 32 * CT - 1) / (32 * CT));
  hipLaunchKernelGGL((traj_wide_kernel<OTW, CT>), dim3(grid), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// Column tiles per workgroup: 64 trajectories (CT = 2) halve the weight traffic per trajectory and the barriers per MFMA, but
// a launch needs >= 256 workgroups to use every CU: below 16 384 trajectories the 32-trajectory form fills more CUs.
int launch_wide(const TrajArgs& a, hipStream_t stream, int* ct_used) {
  const char* force = getenv("SDEH_WIDE_CT");  // testing aid: "1" | "2" (read per call)
  int ct = a.batch > 32 * 256 ? 2 : 1;
  if (force != nullptr && (force[0] == '1' || force[0] == '2')) ct = force[0] - '0';
  if (ct_used != nullptr) *ct_used = ct;
  const int otw = a.lay.c / 128;
  if (otw == 2) return ct == 2 ? launch_wide_t<2, 2>(a, stream) : launch_wide_t<2, 1>(a, stream);
  if (otw == 1) return ct == 2 ? launch_wide_t<1, 2>(a, stream) : launch_wide_t<1, 1>(a, stream);
  return SDEH_ERR_UNSUPPORTED;
}

int launch_bridge_wide(const TrajArgs& a, hipStream_t stream, int* split_used) {
  (void)a; (void)stream;
  if (split_used != nullptr) *split_used = 0;
  return SDEH_ERR_UNSUPPORTED;
}

}  // namespace sdeh
