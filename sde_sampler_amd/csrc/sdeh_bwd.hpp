// Backward pass of the control network: log-variance losses (row-parallel) and KL losses
// (back-propagation through time).
//
// With method = "lv" / "lv_traj" the reference detaches the control that drives the SDE (losses/oc.py:60-70), so the
// trajectory x_t is a constant of the autograd graph and
//     d rnd_i / d u_{i,t}  =  dB_{i,t}                 (the (u_sde - u) dt terms vanish identically)
// for all three losses (TimeReversal 204-219, ReferenceSDE 319-331, ExponentialIntegrator 418-443), with
// dB = xi sqrt(dt) (Euler-Maruyama) or sigma beta_k xi (exponential integrator).  The parameter gradient is therefore a
// sum over independent (step, trajectory) rows of  J_theta u(t, x)^T (w_i dB),  w_i = dLoss/drnd_i coming from autograd.
//
// This kernel does the per-row part on the matrix pipe for 64 rows (one step, 64 consecutive trajectories) per wave:
// it re-evaluates the FourierMLP at the stored x_t, forms G = w_i dB (the Gaussian draws are REPLAYED from the Philox
// counters, or read from the given noise), back-propagates through clip / out_layer / activations / hidden layers
// with the transposed packed weights, and writes -- coordinate-major, i.e. coalesced --
//     Zt[k][C][N]  pre-activations of layer k = 0..Lh      Dt[k][C][N]  d loss / d Z_k
//     Dout[d][N]   d loss / d (network output)              Dgam[g][N]   d loss / d gamma(t) contributions per row
// with N = T * B.  The weight gradients are then plain GEMMs over N (Dt[k] . act(Zt[k-1])^T etc.), done by the host
// with library GEMMs, and the time-only sub-networks are differentiated on their [T, .] tables.
//
// BPTT = true (method "kl" / "kl_ito": the SDE is driven by the attached control): a wave keeps its 64 trajectories and
// walks the stored trajectory backwards, carrying the adjoint lambda_t = dLoss/dx_t in registers:
//     G_t       = w_i d cost_t / d u + c_u lambda_{t+1}                       (upstream of the control at step t)
//     lambda_t  = c_x lambda_{t+1} + (du_t/dx_t)^T G_t + w_i d cost_t/dx_t    (du/dx: MLP input gradient through the clip
//                                                                              + the score term's Jacobian)
// with the reference's autograd semantics: the mixture score is obtained with create_graph=False (distr/base.py:130-137,
// models/reparam.py:58-60), i.e. it is a CONSTANT of the graph, whereas the closed-form scores (Gaussian prior/target,
// double wells, funnel) are differentiated through x.  lambda_T = w_i d(terminal costs)/dx_T.
#pragma once
#include "sdeh_traj_ws.hpp"

namespace sdeh {

__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == SDEH_ACT_GELU_ERF) {  // Phi(v) + v phi(v)
    const float z = fminf(fabsf(v) * 0.70710678118654752440f, 6.0f);
    float p = 6.603050149e-07f;
    p = fmaf(p, z, -1.530170759e-05f); p = fmaf(p, z, 1.480139295e-04f); p = fmaf(p, z, -7.626767611e-04f);
    p = fmaf(p, z, 2.001933838e-03f); p = fmaf(p, z, 3.411742314e-04f); p = fmaf(p, z, -2.809073479e-02f);
    p = fmaf(p, z, 1.484803495e-01f); p = fmaf(p, z, 9.184024644e-01f); p = fmaf(p, z, 1.627910815e+00f);
    const float e = __builtin_amdgcn_exp2f(fmaf(-p, z, -1.0f));
    const float Phi = v < 0.0f ? e : 1.0f - e;
    const float phi = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * v * v);
    return fmaf(v, phi, Phi);
  }
  if (act == SDEH_ACT_SILU) {
    const float s = 1.0f / (1.0f + __expf(-v));
    return s * fmaf(v, 1.0f - s, 1.0f);
  }
  return v > 0.0f ? 1.0f : 0.0f;
}

// GELU and its derivative of an element pair from ONE evaluation of the tail Phi(-|v|) (the fused backward kernels: act and act' are
// half of their vector work).  `a` is act_gelu2(v) bit for bit (the same instructions: the re-evaluated activations stay the forward
// kernel's); act' = Phi(v) + v phi(v) takes Phi from that tail instead of act_grad's own degree-9 fit: |error| <= 2.4e-6 (the
// degree-6 fit is tuned for t * tail, its tail alone is off by that much at 0) against 1.4e-7 -- two orders inside the gradients'
// parity bar (2e-4 of each tensor's scale), for 7 instead of 17 instructions per element.  -DSDEH_NO_JOINT_GELU: the separate forms.
__device__ __forceinline__ void act_gelu2_both(f2 v, f2& a, f2& g) {
#if defined(SDEH_NO_JOINT_GELU) || defined(SDEH_NO_PK_GELU)
  a = act_gelu2(v);
  g = f2{act_grad(v.x, SDEH_ACT_GELU_ERF), act_grad(v.y, SDEH_ACT_GELU_ERF)};
#else
  const f2 t = f2{__builtin_amdgcn_fmed3f(fabsf(v.x), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(fabsf(v.y), 0.0f, 6.0f)};
  f2 q = pk_fma_sc(splat(3.3092907814e-05f), t, splat_bits(-7.6922050644e-04f));
  q = pk_fma_sc(q, t, splat_bits(8.0807191412e-03f));
  q = pk_fma_sc(q, t, splat_bits(-5.3412108121e-02f));
  q = pk_fma_sc(q, t, splat_bits(-4.5877097054e-01f));
  q = pk_fma_sc(q, t, splat_bits(-1.1512017029e+00f));
  q = pk_fma_sc(q, t, splat_bits(-9.9999306093e-01f));
  const f2 e = f2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};  // Phi(-|v|)
  f2 relu;  // (see act_gelu2 for the extra input t)
  asm("v_max_f32 %0, 0, %1" : "=v"(relu.x) : "v"(v.x), "v"(t.x));
  asm("v_max_f32 %0, 0, %1" : "=v"(relu.y) : "v"(v.y), "v"(t.y));
  a = pk_fma(-t, e, relu);
  const f2 v2 = v * v;
  const f2 phi = f2{0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * v2.x),
                    0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * v2.y)};
  const f2 Phi = f2{v.x < 0.0f ? e.x : 1.0f - e.x, v.y < 0.0f ? e.y : 1.0f - e.y};
  g = pk_fma(v, phi, Phi);
#endif
}

// SiLU and its derivative from one exponential: act stays act_silu bit for bit (v / (1 + e^-v), IEEE division); the sigmoid for
// act' = s (1 + v (1 - s)) is v_rcp_f32 of the same denominator (1 ulp) instead of a second exponential and division.
__device__ __forceinline__ void act_silu_both(float v, float& a, float& g) {
  const float den = 1.0f + __expf(-v);
  a = v / den;
  const float sg = __builtin_amdgcn_rcpf(den);
  g = sg * fmaf(v, 1.0f - sg, 1.0f);
}

// act(z) in place and act'(z) of one accumulator tile (training backward kernels: joint evaluation for the GELU)
template <int ACT>
__device__ __forceinline__ void act_both_tile(f32x16& z, f32x16& g) {
  if constexpr (ACT == SDEH_ACT_GELU_ERF) {
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      f2 av, gv;
      act_gelu2_both(f2{z[q], z[q + 1]}, av, gv);
      z[q] = av.x; z[q + 1] = av.y;
      g[q] = gv.x; g[q + 1] = gv.y;
    }
  } else if constexpr (ACT == SDEH_ACT_SILU) {
#pragma unroll
    for (int q = 0; q < 16; ++q) { float av, gv; act_silu_both(z[q], av, gv); z[q] = av; g[q] = gv; }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = act_grad(z[q], ACT);
    act_tile<ACT>(z);
  }
}

// second derivative of the activation (Bridge: the divergence's dependence on the base pre-activations)
__device__ __forceinline__ float act_grad2(float v, int act) {
  if (act == SDEH_ACT_GELU_ERF)  // (Phi + v phi)' = phi (2 - v^2)
    return 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * v * v) * (2.0f - v * v);
  if (act == SDEH_ACT_SILU) {  // s' (2 + v (1 - 2 s)),  s = sigmoid(v)
    const float s = 1.0f / (1.0f + __expf(-v));
    return s * (1.0f - s) * fmaf(v, 1.0f - 2.0f * s, 2.0f);
  }
  return 0.0f;
}

// store / load one [C][N] plane in the M layout: register q of lane (j,h), row tile ot, column tile A/B is
// channel 32 ot + rho(q,h) of row n0 + j (+32)
template <int OT, bool HALF = false>
__device__ __forceinline__ void store_plane(float* __restrict__ plane, long long N, long long n0, int nrows, int lane,
                                            const f32x16 (&a)[OT], const f32x16 (&b)[OT]) {
  const int h = lane >> 5, j = lane & 31;
  float* __restrict__ base = plane + n0 + j;
  if (j < nrows) {  // one predicated region per tile (not one per element)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) base[(long long)(32 * ot + rho(q, h)) * N] = a[ot][q];
  }
  if constexpr (!HALF) {
    if (j + 32 < nrows) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) base[(long long)(32 * ot + rho(q, h)) * N + 32] = b[ot][q];
    }
  }
}
template <int OT, bool HALF = false>
__device__ __forceinline__ void load_plane(const float* __restrict__ plane, long long N, long long n0, int nrows, int lane,
                                           f32x16 (&a)[OT], f32x16 (&b)[OT]) {
  const int h = lane >> 5, j = lane & 31;
  const float* __restrict__ base = plane + n0 + j;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      a[ot][q] = 0.0f;
      if constexpr (!HALF) b[ot][q] = 0.0f;
    }
  if (j < nrows) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) a[ot][q] = base[(long long)(32 * ot + rho(q, h)) * N];
  }
  if constexpr (!HALF) {
    if (j + 32 < nrows) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) b[ot][q] = base[(long long)(32 * ot + rho(q, h)) * N + 32];
    }
  }
}

// v = J^T c for the closed-form scores (the reference differentiates these through x; the mixture's autograd score is a
// constant of the graph and contributes nothing)
template <int DP>
__device__ __forceinline__ void target_score_jt(const DensArgs& D, const float* ws, const WsLayout& L, int dreal,
                                                const float (&x)[DP], const float (&c)[DP], float (&v)[DP]) {
  switch (D.kind) {
    case SDEH_DENS_DIAG_GAUSS: {
      cf2p p = as_const2(ws + L.dg[0]);
#pragma unroll
      for (int j = 0; j < DP; ++j) v[j] = -p[j].y * c[j];
      break;
    }
    case SDEH_DENS_MULTI_WELL:
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const float y = x[j] - D.p1;
        const float jac = j < D.n_comp ? -4.0f * (3.0f * y * y - D.p0) : -1.0f;
        v[j] = j < dreal ? jac * c[j] : 0.0f;
      }
      break;
    case SDEH_DENS_FUNNEL: {  // s_0 = -x0/var - (d-1)/2 + e^{-x0} sum x_j^2 / 2,  s_j = -x_j e^{-x0}
      const float iv = __expf(-x[0]);
      float sq = 0.0f, cx = 0.0f;
#pragma unroll
      for (int j = 1; j < DP; ++j) { sq = fmaf(x[j], x[j], sq); cx = fmaf(c[j], x[j], cx); }
      v[0] = c[0] * (-1.0f / D.p0 - 0.5f * iv * sq) + iv * cx;
#pragma unroll
      for (int j = 1; j < DP; ++j) v[j] = iv * (c[0] * x[j] - c[j]);
      break;
    }
    default:  // SDEH_DENS_GMM (constant by the reference's semantics), none
#pragma unroll
      for (int j = 0; j < DP; ++j) v[j] = 0.0f;
  }
}

// One 64-row tile (step t, trajectories i0..i0+63).  BPTT: `lam` holds dLoss/dx_{t+1} on entry and dLoss/dx_t on exit.
// HALF: 32-row tiles (lanes 0..31 = MFMA column tile A only; nrows <= 32) -- half the dependent MFMA chain and half the
// activation work per wave, for back-propagation through time at batches that leave SIMDs idle.
template <int DP, int C, bool PAD, bool BPTT, bool HALF = false>
__device__ __forceinline__ void bwd_tile(const BwdArgs& A, const float* __restrict__ lds, int t, long long i0, int nrows,
                                         int lane, float (&lam)[DP]) {
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int h = lane >> 5;
  const long long B = A.batch;
  const long long N = B * A.n_steps;
  const long long n0 = (long long)t * B + i0;
  const bool live = lane < nrows;
  const long long irow = live ? i0 + lane : B - 1;
  const int d = PAD ? A.d : DP;
  const int act = A.act, ctrl_kind = A.ctrl_kind;
  const bool refc = (A.flags & SDEH_FLAG_REFERENCE_CTRL) && A.loss_kind == SDEH_LOSS_REFERENCE_SDE;

  float x[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    const float v = A.xs[((long long)t * B + irow) * d + (PAD ? min(j, d - 1) : j)];
    x[j] = (!PAD || j < d) ? v : 0.0f;
  }
  cfp cf = as_const(ws + L.coef + t * kCoefStride);

  // The pre-activations Z_k and the raw network output nn either come from the training forward launch (A.nn_in != null: the
  // wave-specialised kernel wrote the planes while it integrated, sdeh_simulate_fwd_train) or are re-evaluated here.
  float nn[DP];
  if (A.nn_in != nullptr) {
    const float* __restrict__ np = A.nn_in + ((long long)t * B + irow) * d;
#pragma unroll
    for (int j = 0; j < DP; ++j) nn[j] = (!PAD || j < d) ? np[PAD ? min(j, d - 1) : j] : 0.0f;
  } else {
    // ---- forward, storing the pre-activations ------------------------------------------------------------------
    f32x16 accA[OT], accB[OT];
    {
      const float* emb = ws + L.emb + t * C;
  #pragma unroll
      for (int ot = 0; ot < OT; ++ot) accA[ot] = accB[ot] = load16(emb + (ot * 2 + h) * 16);
      float xa[R], xb[R];
  #pragma unroll
      for (int r = 0; r < R; ++r) {
        float v0 = x[mdim(r, 0)];
        float v1 = mdim(r, 1) < DP ? x[mdim(r, 1)] : 0.0f;
        swap32(v0, v1);
        xa[r] = v0;
        xb[r] = v1;
      }
      const float* w = lds + L.w_in + lane;
  #pragma unroll
      for (int r = 0; r < R; ++r)
  #pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const float a = w[(r * OT + ot) * 64];
          accA[ot] = SDEH_MFMA(a, xa[r], accA[ot]);
          if constexpr (!HALF) accB[ot] = SDEH_MFMA(a, xb[r], accB[ot]);
          if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
        }
    }
    store_plane<OT, HALF>(A.zt, N, n0, nrows, lane, accA, accB);
    for (int l = 0; l < L.n_hidden; ++l) {
      if constexpr (HALF) activate_one<OT>(accA, act);
      else activate<OT>(accA, accB, act);
      f32x16 nA[OT], nB[OT];
      const float* bias = lds + L.b_hid + l * C;
  #pragma unroll
      for (int ot = 0; ot < OT; ++ot) nA[ot] = nB[ot] = load16(bias + (ot * 2 + h) * 16);
      const float* w = lds + L.w_hid + l * L.w_hid_stride + lane;
  #pragma unroll
      for (int it = 0; it < OT; ++it)
  #pragma unroll
        for (int q = 0; q < 16; ++q)
  #pragma unroll
          for (int ot = 0; ot < OT; ++ot) {
            const float a = w[((it * 16 + q) * OT + ot) * 64];
            nA[ot] = SDEH_MFMA(a, accA[it][q], nA[ot]);
            if constexpr (!HALF) nB[ot] = SDEH_MFMA(a, accB[it][q], nB[ot]);
            if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
          }
  #pragma unroll
      for (int ot = 0; ot < OT; ++ot) { accA[ot] = nA[ot]; accB[ot] = nB[ot]; }
      store_plane<OT, HALF>(A.zt + (long long)(l + 1) * C * N, N, n0, nrows, lane, accA, accB);
    }
    {
      if constexpr (HALF) activate_one<OT>(accA, act);
      else activate<OT>(accA, accB, act);
      f32x16 uA[OTD], uB[OTD];
  #pragma unroll
      for (int tt = 0; tt < OTD; ++tt) uA[tt] = uB[tt] = load16(lds + L.b_out + (tt * 2 + h) * 16);
      const float* w = lds + L.w_out + lane;
  #pragma unroll
      for (int it = 0; it < OT; ++it)
  #pragma unroll
        for (int q = 0; q < 16; ++q)
  #pragma unroll
          for (int tt = 0; tt < OTD; ++tt) {
            const float a = w[((it * 16 + q) * OTD + tt) * 64];
            uA[tt] = SDEH_MFMA(a, accA[it][q], uA[tt]);
            if constexpr (!HALF) uB[tt] = SDEH_MFMA(a, accB[it][q], uB[tt]);
            if (tt == OTD - 1 && (q & 1)) SDEH_FENCE();
          }
  #pragma unroll
      for (int r = 0; r < R; ++r) {
        float v0 = uA[r / 16][r % 16];
        float v1 = uB[r / 16][r % 16];
        swap32(v0, v1);
        nn[mdim(r, 0)] = v0;
        if (mdim(r, 1) < DP) nn[mdim(r, 1)] = v1;
      }
    }
  }
  SDEH_FENCE();

  // ---- score term: sc = (1-w') prior_score + w' target_score (per control kind), S = mult * clip(sc) * gamma ---------
  const float sig = cf[CF_SIGMA], wl = cf[CF_W];
  float sc[DP], mfac[DP];  // mfac_j = mult * scale_score * gamma_j  (d S_j / d clip(sc_j))
  float coef_t = 0.0f, coef_p = 0.0f;
  if (ctrl_kind != SDEH_CTRL_CLIPPED) {
    coef_t = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET ? wl : 0.0f);
    coef_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR ? 1.0f - wl : 0.0f;
    float tsc[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) tsc[j] = 0.0f;
    if (ctrl_kind != SDEH_CTRL_LERP_PRIOR) ws_target_score<DP, DP>(A.target, ws, lds, L, L.gmm_lds, d, x, tsc);
    if (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR) {
      float psc[DP];
      dgauss_score<DP>(ws + L.dg[1], x, psc);
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        if (ctrl_kind == SDEH_CTRL_LERP_PRIOR) sc[j] = (1.0f - wl) * psc[j];
        else sc[j] = wl < 0.5f ? psc[j] + wl * (tsc[j] - psc[j]) : tsc[j] - (tsc[j] - psc[j]) * (1.0f - wl);
      }
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) sc[j] = coef_t * tsc[j];
    }
    cfp gam = as_const(ws + L.gam + t * L.g);
    const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
#pragma unroll
    for (int j = 0; j < DP; ++j) mfac[j] = mult * (L.g == 1 ? gam[0] : gam[j]);
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j) { sc[j] = 0.0f; mfac[j] = 0.0f; }
  }

  // ---- Gaussian draws of this step (replayed) ---------------------------------------------------------------------
  float xi[DP];
  const bool need_xi = (A.flags & SDEH_FLAG_ITO) != 0;  // the log-variance methods always carry the Ito term
  if (need_xi) {
    if (A.noise != nullptr) {
      const float* __restrict__ np = A.noise + ((long long)t * B + irow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j) xi[j] = np[PAD ? min(j, d - 1) : j];
    } else {
      const unsigned long long grow = (unsigned long long)(A.row_offset + irow);
      const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
#pragma unroll
      for (int jb = 0; jb < (DP + 3) / 4; ++jb) {
        float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!PAD || 4 * jb < d) box_muller4(philox_block(A.seed, rng_off, grow, t, jb), n);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * jb + q < DP) xi[4 * jb + q] = n[q];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j) xi[j] = 0.0f;
  }

  // ---- upstream gradient of the control ------------------------------------------------------------------------------
  const bool expo = A.loss_kind == SDEH_LOSS_EXPONENTIAL;
  const float wi = A.grad_rnd[irow];
  const float c_i = expo ? cf[CF_SBK] : cf[CF_SQDT];              // dB = c_i xi
  const float cdt = expo ? cf[CF_B2S2] : cf[CF_DT];               // running cost = cdt * (...)
  const float c_u = expo ? cf[CF_B2S2] : sig * cf[CF_DT];         // x_{t+1} = c_x x_t + c_u u_t + c_n xi_t
  const float c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], cf[CF_DT], 1.0f);
  float G[DP], Gc[DP];  // Gc: the running-cost part  w_i d cost_t / d g  (g = u, or u - reference control)
  cf2p ptab = as_const2(ws + L.dg[1]);
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    if (!BPTT) {  // log-variance: d rnd / d u = dB exactly (+ the Bridge cost's u + v for the inference network)
      Gc[j] = (A.flags & SDEH_FLAG_ITO) ? wi * c_i * xi[j] : 0.0f;
      if (A.gextra != nullptr) Gc[j] = fmaf(wi * cdt, A.gextra[((long long)t * B + irow) * d + (PAD ? min(j, d - 1) : j)], Gc[j]);
      G[j] = Gc[j];
    } else {
      const float u = clipf(nn[j], A.clip_model) + mfac[j] * clipf(sc[j], A.clip_score);
      const float r = refc ? sig * (ptab[j].x - x[j]) * ptab[j].y : 0.0f;
      const float uc = A.cost_ctrl != nullptr ? A.cost_ctrl[((long long)t * B + irow) * d + (PAD ? min(j, d - 1) : j)] : u - r;
      Gc[j] = wi * fmaf(uc, cdt, (A.flags & SDEH_FLAG_ITO) ? c_i * xi[j] : 0.0f);
      G[j] = fmaf(c_u, lam[j], Gc[j]);
    }
    if (PAD) { G[j] = j < d ? G[j] : 0.0f; Gc[j] = j < d ? Gc[j] : 0.0f; }
  }

  // ---- d loss / d gamma(t) -----------------------------------------------------------------------------------------------
  if (ctrl_kind != SDEH_CTRL_CLIPPED) {
    const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
    if (L.g == 1) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < DP; ++j) s = fmaf(G[j], mult * clipf(sc[j], A.clip_score), s);
      if (live) A.dgam[n0 + lane] = s;
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if ((!PAD || j < d) && live) A.dgam[(long long)j * N + n0 + lane] = G[j] * mult * clipf(sc[j], A.clip_score);
    }
  }

  // ---- d loss / d nn through the clip; to the M layout ------------------------------------------------------------
  f32x16 dA[OT], dB[OT];
  {
    float ga[R], gb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j0 = mdim(r, 0), j1 = mdim(r, 1);
      float v0 = fabsf(nn[j0]) <= A.clip_model ? G[j0] : 0.0f;
      float v1 = j1 < DP ? (fabsf(nn[j1]) <= A.clip_model ? G[j1] : 0.0f) : 0.0f;
      swap32(v0, v1);
      ga[r] = v0;
      gb[r] = v1;
      // Dout[d][N]: lane (j,h) holds coordinate mdim(r,h) of rows n0 + j (tile A) and n0 + 32 + j (tile B)
      const int dim = h ? j1 : j0;
      if (dim < d) {
        float* row = A.dout + (long long)dim * N + n0 + (lane & 31);
        if ((lane & 31) < nrows) row[0] = v0;
        if ((lane & 31) + 32 < nrows) row[32] = v1;
      }
    }
    // d a_last = W_out^T d out
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) dA[ot][q] = dB[ot][q] = 0.0f;
    const float* w = lds + L.wt_out + lane;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const float a = w[(r * OT + ot) * 64];
        dA[ot] = SDEH_MFMA(a, ga[r], dA[ot]);
        if constexpr (!HALF) dB[ot] = SDEH_MFMA(a, gb[r], dB[ot]);
        if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
      }
  }
  // ---- back through the layers: dZ_k = dA_k * act'(Z_k);  dA_{k-1} = W_{k-1}^T dZ_k ------------------------------------
  // Z_k comes back from the plane this wave wrote in the forward sweep; the load of Z_{k-1} is issued before the MFMA block of
  // layer k so that its round trip is covered (the time loop of BPTT is one dependent chain)
  f32x16 zA[OT], zB[OT];
  load_plane<OT, HALF>(A.zt + (long long)L.n_hidden * C * N, N, n0, nrows, lane, zA, zB);
  for (int k = L.n_hidden; k >= 0; --k) {
    SDEH_ACT_SWITCH(act, ACT,
      _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
          dA[ot][q] *= act_grad(zA[ot][q], ACT);
          if constexpr (!HALF) dB[ot][q] *= act_grad(zB[ot][q], ACT);
        });
    store_plane<OT, HALF>(A.dt + (long long)k * C * N, N, n0, nrows, lane, dA, dB);
    if (k > 0) {
      load_plane<OT, HALF>(A.zt + (long long)(k - 1) * C * N, N, n0, nrows, lane, zA, zB);
      f32x16 pA[OT], pB[OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) pA[ot][q] = pB[ot][q] = 0.0f;
      const float* w = lds + L.wt_hid + (k - 1) * L.w_hid_stride + lane;
#pragma unroll
      for (int it = 0; it < OT; ++it)
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) {
            const float a = w[((it * 16 + q) * OT + ot) * 64];
            pA[ot] = SDEH_MFMA(a, dA[it][q], pA[ot]);
            if constexpr (!HALF) pB[ot] = SDEH_MFMA(a, dB[it][q], pB[ot]);
            if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
          }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { dA[ot] = pA[ot]; dB[ot] = pB[ot]; }
    }
  }

  if (BPTT || A.dx != nullptr) {
    // ---- adjoint update: lambda_t = c_x lambda_{t+1} + W_in^T dZ_0 + (dS/dx)^T G + direct cost terms -----------------------
    // (row-parallel mode with A.dx: the same quantity without the recursion, written to the plane)
    float dx[DP];
    {
      f32x16 xA[OTD], xB[OTD];
#pragma unroll
      for (int tt = 0; tt < OTD; ++tt)
#pragma unroll
        for (int q = 0; q < 16; ++q) xA[tt][q] = xB[tt][q] = 0.0f;
      const float* w = lds + L.wt_in + lane;
#pragma unroll
      for (int it = 0; it < OT; ++it)
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int tt = 0; tt < OTD; ++tt) {
            const float a = w[((it * 16 + q) * OTD + tt) * 64];
            xA[tt] = SDEH_MFMA(a, dA[it][q], xA[tt]);
            if constexpr (!HALF) xB[tt] = SDEH_MFMA(a, dB[it][q], xB[tt]);
            if (tt == OTD - 1 && (q & 1)) SDEH_FENCE();
          }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float v0 = xA[r / 16][r % 16];
        float v1 = xB[r / 16][r % 16];
        swap32(v0, v1);
        dx[mdim(r, 0)] = v0;
        if (mdim(r, 1) < DP) dx[mdim(r, 1)] = v1;
      }
    }
    float cvec[DP], vt[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) cvec[j] = fabsf(sc[j]) <= A.clip_score ? mfac[j] * G[j] : 0.0f;
    // score terms the reference detaches (reparam.py:58,134,169,188) or obtains by autograd without a graph carry no Jacobian
    const float jac_t = (A.flags & (SDEH_FLAG_DETACH_SCORE | SDEH_FLAG_TARGET_SCORE_CONST)) ? 0.0f : coef_t;
    const float jac_p = (A.flags & SDEH_FLAG_DETACH_SCORE) ? 0.0f : coef_p;
    if (jac_t != 0.0f) target_score_jt<DP>(A.target, ws, L, d, x, cvec, vt);
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      float v = BPTT ? fmaf(c_x, lam[j], dx[j]) : dx[j];
      if (jac_t != 0.0f) v = fmaf(jac_t, vt[j], v);
      if (jac_p != 0.0f) v = fmaf(-jac_p * ptab[j].y, cvec[j], v);   // Gaussian prior: J = -1/sigma^2
      if (refc) v = fmaf(sig * ptab[j].y, Gc[j], v);                   // cost depends on x through sigma * prior.score(x)
      if (BPTT) {
        if (A.lam_extra != nullptr) v += A.lam_extra[((long long)t * B + irow) * d + (PAD ? min(j, d - 1) : j)];
        lam[j] = (!PAD || j < d) ? v : 0.0f;
      } else if (live && (!PAD || j < d)) {
        A.dx[((long long)t * B + irow) * d + j] = v;
      }
    }
  }
}

template <int DP, int C, bool PAD, bool BPTT, bool HALF = false>
__global__ __launch_bounds__(256) void ctrl_bwd_kernel(const BwdArgs A) {
  static_assert(BPTT == HALF, "row-parallel mode uses 64-row tiles, back-propagation through time 32-row tiles");
  constexpr int TR = HALF ? 32 : 64;  // trajectories per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WsLayout& L = A.lay;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const float4* src = reinterpret_cast<const float4*>(A.ws);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < L.lds_floats / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const long long B = A.batch;
  const long long tiles_per_t = (B + 63) / 64;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  float lam[DP];
  if constexpr (!BPTT) {
    if (tile >= tiles_per_t * A.n_steps) return;
    const int t = (int)(tile / tiles_per_t);
    const long long i0 = (tile % tiles_per_t) * 64;
#pragma unroll
    for (int j = 0; j < DP; ++j) lam[j] = 0.0f;
    bwd_tile<DP, C, PAD, false>(A, lds, t, i0, (int)(B - i0 < 64 ? B - i0 : 64), lane, lam);
  } else {
    if (tile >= (B + TR - 1) / TR) return;
    const long long i0 = tile * TR;
    const int nrows = (int)(B - i0 < TR ? B - i0 : TR);
    const long long irow = lane < nrows ? i0 + lane : B - 1;
    const int d = PAD ? A.d : DP;
    // lambda_T = w_i d(terminal costs)/dx_T  (losses/oc.py:225,337,449-450)
    float x[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const float v = A.xs[((long long)A.n_steps * B + irow) * d + (PAD ? min(j, d - 1) : j)];
      x[j] = (!PAD || j < d) ? v : 0.0f;
    }
    const float wi = A.grad_rnd[irow];
#pragma unroll
    for (int j = 0; j < DP; ++j) lam[j] = 0.0f;
    if (A.flags & SDEH_FLAG_TERMINAL_SECOND) {
      float s2[DP];
      dgauss_score<DP>(A.ws + L.dg[2], x, s2);
#pragma unroll
      for (int j = 0; j < DP; ++j) lam[j] = wi * s2[j];
    }
    if (A.flags & SDEH_FLAG_TERMINAL_TARGET) {
      float st[DP];
      ws_target_score<DP, DP>(A.target, A.ws, lds, L, L.gmm_lds, d, x, st);
      float keep = 1.0f;
      if (A.clip_target < 3.0e38f) {
        const float lp = ws_target_logp<DP, DP>(A.target, A.ws, lds, L, L.gmm_lds, d, x);
        keep = fabsf(lp) <= A.clip_target ? 1.0f : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) lam[j] = fmaf(-wi * keep, st[j], lam[j]);
    }
    if (PAD) {
#pragma unroll
      for (int j = 0; j < DP; ++j) lam[j] = j < d ? lam[j] : 0.0f;
    }
    for (int t = A.n_steps - 1; t >= 0; --t) bwd_tile<DP, C, PAD, true, HALF>(A, lds, t, i0, nrows, lane, lam);
  }
}

template <int DP, int C, bool PAD>
int launch_ctrl_bwd(const BwdArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)a.lay.lds_floats * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  const bool bptt = !(a.flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  static bool attr_done[kMaxDevices] = {};  // the raised LDS limit is a per-device function attribute
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ctrl_bwd_kernel<DP, C, PAD, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ctrl_bwd_kernel<DP, C, PAD, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  // Back-propagation through time is one dependent chain per wave and always runs 32-row tiles: half the MFMA chain per wave at
  // small batches (latency), no register spills and the same MFMA work per row at large ones (measured equal or faster than
  // 64-row tiles from B = 2048 to 131 072, tests/perf/bptt_tiles_timing.py history in DESIGN.md 3b).
  const long long tiles = bptt ? (a.batch + 31) / 32 : ((a.batch + 63) / 64) * a.n_steps;
  const dim3 grid((unsigned)((tiles + 3) / 4));
  if (bptt) hipLaunchKernelGGL((ctrl_bwd_kernel<DP, C, PAD, true, true>), grid, dim3(256), lds_bytes, stream, a);
  else hipLaunchKernelGGL((ctrl_bwd_kernel<DP, C, PAD, false>), grid, dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
