// Bridge branch of TimeReversalLoss.simulate (losses/oc.py:176-230 with inference_ctrl != None, lines 189-202):
//   u = generative_ctrl(s, x);  (div, v) = compute_divx(inference_ctrl, s, x)   [exact divergence, utils/autograd.py:14-22]
//   rnd += sigma div dt;  cost on u + v (kl) / (u + v).(u_sde - (u - v)/2) (lv);  Ito term on u + v;  x driven by u alone.
//
// The reference obtains div_x v with d backward passes through the inference network per step.  Here the diagonal of the
// network's Jacobian comes from d forward-mode tangent passes on the matrix pipe: the tangent of input direction e_j
// starts as column j of input_embed.weight (the same for every trajectory), is scaled by act'(z_l) and pushed through the
// hidden layers' MFMAs NEXT TO the base activations (same A operands = packed weights, a second B operand), and is read
// out against row j of out_layer.weight.  ClippedCtrl's clamp contributes the 0/1 mask d clip(v_j)/d v_j, the LerpPriorCtrl
// score term its closed-form derivative.  One wave per 64 trajectories (T layout), both networks' packed weights in LDS.
#pragma once
#include "sdeh_bwd.hpp"

namespace sdeh {

// FourierMLP value and the tangent of input direction e_jt: out = NN(t, x) (T layout), djj = d NN_jt / d x_jt.
// tin / tout: column jt of input_embed.weight / row jt of out_layer.weight in accumulator order (WsLayout::tan_in/out).
// VEC (Hutchinson probe, utils/autograd.py:25-42): the tangent direction is the per-trajectory vector `eps` instead of e_jt
// (seed = W_in eps through the input layer's MFMAs) and the read-out is the whole vector tvec = J eps.
template <int DP, int C, bool VEC = false, bool HALF = false>
__device__ __forceinline__ void mlp_forward_tangent(const float* __restrict__ lds, const WsLayout& L, int act,
                                                    const float* __restrict__ emb_step, const float* __restrict__ tin,
                                                    const float* __restrict__ tout, const float (&x)[DP],
                                                    float (&out)[DP], float& djj, int lane, const float* eps = nullptr,
                                                    float* tvec = nullptr) {
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5;
  f32x16 accA[OT], accB[OT], tA[OT], tB[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    accA[ot] = accB[ot] = load16(emb_step + (ot * 2 + h) * 16);
    if constexpr (VEC) {
#pragma unroll
      for (int q = 0; q < 16; ++q) tA[ot][q] = tB[ot][q] = 0.0f;
    } else {
      tA[ot] = tB[ot] = load16(tin + (ot * 2 + h) * 16);  // d z_0 / d x_jt = W_in[:, jt] for every trajectory
    }
  }
  {
    float xa[R], xb[R], ea[R], eb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v0 = x[mdim(r, 0)];
      float v1 = mdim(r, 1) < DP ? x[mdim(r, 1)] : 0.0f;
      swap32(v0, v1);
      xa[r] = v0;
      xb[r] = v1;
      if constexpr (VEC) {
        float e0 = eps[mdim(r, 0)];
        float e1 = mdim(r, 1) < DP ? eps[mdim(r, 1)] : 0.0f;
        swap32(e0, e1);
        ea[r] = e0;
        eb[r] = e1;
      }
    }
    const float* w = lds + L.w_in + lane;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const float a = w[(r * OT + ot) * 64];
        accA[ot] = SDEH_MFMA(a, xa[r], accA[ot]);
        if constexpr (!HALF) accB[ot] = SDEH_MFMA(a, xb[r], accB[ot]);
        if constexpr (VEC) {
          tA[ot] = SDEH_MFMA(a, ea[r], tA[ot]);
          if constexpr (!HALF) tB[ot] = SDEH_MFMA(a, eb[r], tB[ot]);
        }
        if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
      }
  }
  f32x16 uA[OTD], uB[OTD], tuA[VEC ? OTD : 1], tuB[VEC ? OTD : 1];
  float sA = 0.0f, sB = 0.0f;
  for (int l = 0; l <= L.n_hidden; ++l) {
    // a_l = act(z_l);  d a_l = act'(z_l) d z_l
    SDEH_ACT_SWITCH(act, ACT,
      _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
          tA[ot][q] *= act_grad(accA[ot][q], ACT);
          if constexpr (!HALF) tB[ot][q] *= act_grad(accB[ot][q], ACT);
        });
    if constexpr (HALF) activate_one<OT>(accA, act);
    else activate<OT>(accA, accB, act);
    if (l < L.n_hidden) {
      f32x16 nA[OT], nB[OT], ntA[OT], ntB[OT];
      const float* bias = lds + L.b_hid + l * C;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        nA[ot] = nB[ot] = load16(bias + (ot * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 16; ++q) ntA[ot][q] = ntB[ot][q] = 0.0f;
      }
      const float* w = lds + L.w_hid + l * L.w_hid_stride + lane;
#pragma unroll
      for (int it = 0; it < OT; ++it)
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) {
            const float a = w[((it * 16 + q) * OT + ot) * 64];
            nA[ot] = SDEH_MFMA(a, accA[it][q], nA[ot]);
            ntA[ot] = SDEH_MFMA(a, tA[it][q], ntA[ot]);
            if constexpr (!HALF) {
              nB[ot] = SDEH_MFMA(a, accB[it][q], nB[ot]);
              ntB[ot] = SDEH_MFMA(a, tB[it][q], ntB[ot]);
            }
            if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
          }
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) { accA[ot] = nA[ot]; accB[ot] = nB[ot]; tA[ot] = ntA[ot]; tB[ot] = ntB[ot]; }
    } else {
#pragma unroll
      for (int t = 0; t < OTD; ++t) uA[t] = uB[t] = load16(lds + L.b_out + (t * 2 + h) * 16);
      const float* w = lds + L.w_out + lane;
#pragma unroll
      for (int it = 0; it < OT; ++it)
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int t = 0; t < OTD; ++t) {
            const float a = w[((it * 16 + q) * OTD + t) * 64];
            uA[t] = SDEH_MFMA(a, accA[it][q], uA[t]);
            if constexpr (!HALF) uB[t] = SDEH_MFMA(a, accB[it][q], uB[t]);
            if constexpr (VEC) {
              if (it == 0 && q == 0) {
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) tuA[t][qq] = tuB[t][qq] = 0.0f;
              }
              tuA[t] = SDEH_MFMA(a, tA[it][q], tuA[t]);
              if constexpr (!HALF) tuB[t] = SDEH_MFMA(a, tB[it][q], tuB[t]);
            }
            if (t == OTD - 1 && (q & 1)) SDEH_FENCE();
          }
      if constexpr (!VEC) {
        // read-out of the tangent: row jt of out_layer.weight . d a_last (this lane holds 32 of the 64 channels per tile)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const f32x16 wr = load16(tout + (ot * 2 + h) * 16);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            sA = fmaf(wr[q], tA[ot][q], sA);
            sB = fmaf(wr[q], tB[ot][q], sB);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float v0 = uA[r / 16][r % 16];
    float v1 = uB[r / 16][r % 16];
    swap32(v0, v1);
    out[mdim(r, 0)] = v0;
    if (mdim(r, 1) < DP) out[mdim(r, 1)] = v1;
    if constexpr (VEC) {
      float t0 = tuA[r / 16][r % 16];
      float t1 = tuB[r / 16][r % 16];
      swap32(t0, t1);
      tvec[mdim(r, 0)] = t0;
      if (mdim(r, 1) < DP) tvec[mdim(r, 1)] = t1;
    }
  }
  if constexpr (!VEC) {
    // the other lane half holds the remaining channels of the same trajectory column
    sA = sum_xor32(sA);
    sB = sum_xor32(sB);
    djj = lane < 32 ? sA : sB;  // T layout: lane = trajectory (tile A: 0..31, tile B: 32..63)
  }
}

// ---- 32-row tiles: the activation derivatives stay in registers ------------------------------------------------------
// With 32 trajectories per wave a layer's act'(z_l) is 2 x 16 registers, so the base pass of the inference network can keep
// them for every layer (KC layers) and each coordinate's tangent becomes the bare recursion
//     dz_0 = W_in[:, j];   dz_{l+1} = W_l (act'(z_l) . dz_l);   J_jj = W_out[j, :] (act'(z_Lh) . dz_Lh)
// -- no second evaluation of the base network, of act or act' per coordinate: 1 + d hidden-layer MFMA blocks per step
// instead of 1 + 2 d full passes.  Same operations in the same order as mlp_forward_tangent, hence the same bits.
constexpr int kTanCache = 4;  // pre-activation layers kept (networks with n_hidden <= 3; deeper ones take the generic path)

template <int DP, int C>
__device__ __forceinline__ void mlp_forward_dcache(const float* __restrict__ lds, const WsLayout& L, int act,
                                                   const float* __restrict__ emb_step, const float (&x)[DP],
                                                   float (&out)[DP], int lane, f32x16 (&dc)[kTanCache][C / 32]) {
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5;
  f32x16 accA[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) accA[ot] = load16(emb_step + (ot * 2 + h) * 16);
  {
    float xa[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v0 = x[mdim(r, 0)];
      float v1 = mdim(r, 1) < DP ? x[mdim(r, 1)] : 0.0f;
      swap32(v0, v1);
      xa[r] = v0;
    }
    const float* w = lds + L.w_in + lane;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        accA[ot] = SDEH_MFMA(w[(r * OT + ot) * 64], xa[r], accA[ot]);
        if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
      }
  }
  f32x16 uA[OTD];
#pragma unroll
  for (int l = 0; l < kTanCache; ++l) {
    if (l <= L.n_hidden) {
      SDEH_ACT_SWITCH(act, ACT,
        _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)
          _Pragma("unroll") for (int q = 0; q < 16; ++q) dc[l][ot][q] = act_grad(accA[ot][q], ACT););
      activate_one<OT>(accA, act);
      if (l < L.n_hidden) {
        f32x16 nA[OT];
        const float* bias = lds + L.b_hid + l * C;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) nA[ot] = load16(bias + (ot * 2 + h) * 16);
        const float* w = lds + L.w_hid + l * L.w_hid_stride + lane;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
              nA[ot] = SDEH_MFMA(w[((it * 16 + q) * OT + ot) * 64], accA[it][q], nA[ot]);
              if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
            }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) accA[ot] = nA[ot];
      } else {
#pragma unroll
        for (int t = 0; t < OTD; ++t) uA[t] = load16(lds + L.b_out + (t * 2 + h) * 16);
        const float* w = lds + L.w_out + lane;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int t = 0; t < OTD; ++t) {
              uA[t] = SDEH_MFMA(w[((it * 16 + q) * OTD + t) * 64], accA[it][q], uA[t]);
              if (t == OTD - 1 && (q & 1)) SDEH_FENCE();
            }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float v0 = uA[r / 16][r % 16];
    float v1 = 0.0f;
    swap32(v0, v1);  // lanes 0..31: v0 = coordinate mdim(r,0), v1 = coordinate mdim(r,1) (from lanes 32..63) of trajectory `lane`
    out[mdim(r, 0)] = v0;
    if (mdim(r, 1) < DP) out[mdim(r, 1)] = v1;
  }
}

// J_jj for the coordinate whose tables are tin (column of input_embed.weight) / tout (row of out_layer.weight)
template <int C>
__device__ __forceinline__ float mlp_tangent_cached(const float* __restrict__ lds, const WsLayout& L,
                                                    const float* __restrict__ tin, const float* __restrict__ tout,
                                                    const f32x16 (&dc)[kTanCache][C / 32], int lane) {
  constexpr int OT = C / 32;
  const int h = lane >> 5;
  f32x16 tA[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) tA[ot] = load16(tin + (ot * 2 + h) * 16);
  float s = 0.0f;
#pragma unroll
  for (int l = 0; l < kTanCache; ++l) {
    if (l <= L.n_hidden) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) tA[ot][q] *= dc[l][ot][q];
      if (l < L.n_hidden) {
        f32x16 ntA[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int q = 0; q < 16; ++q) ntA[ot][q] = 0.0f;
        const float* w = lds + L.w_hid + l * L.w_hid_stride + lane;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
              ntA[ot] = SDEH_MFMA(w[((it * 16 + q) * OT + ot) * 64], tA[it][q], ntA[ot]);
              if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
            }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) tA[ot] = ntA[ot];
      } else {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const f32x16 wr = load16(tout + (ot * 2 + h) * 16);
#pragma unroll
          for (int q = 0; q < 16; ++q) s = fmaf(wr[q], tA[ot][q], s);
        }
      }
    }
  }
  return sum_xor32(s);  // the other lane half holds the remaining channels of the same trajectory column
}

template <int DP, int C, bool PAD, bool HALF>
__global__ __launch_bounds__(256) void bridge_kernel(const float* __restrict__ ws, const float* __restrict__ x0,
                                                     const float* __restrict__ noise, float* __restrict__ xT,
                                                     float* __restrict__ rnd_out, float* __restrict__ xs,
                                                     const TrajArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WsLayout& L = A.lay;
  const WsLayout& L2 = A.lay2;
  const float* __restrict__ ws2 = A.ws2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  float* lds2 = lds + L.lds_floats;  // the inference network's packed weights
  {
    const float4* src = reinterpret_cast<const float4*>(ws);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < L.lds_floats / 4; i += 256) dst[i] = src[i];
    const float4* src2 = reinterpret_cast<const float4*>(ws2);
    float4* dst2 = reinterpret_cast<float4*>(lds2);
    for (int i = tid; i < L2.lds_floats / 4; i += 256) dst2[i] = src2[i];
  }
  float* lg_lds = lds + L.lds_floats + L2.lds_floats + tid;  // [K][256] mixture-logit scratch
  __syncthreads();

  // half: a wave carries 32 trajectories (lanes 0..31 = MFMA column tile A): half the dependent MFMA chain per step, for batches
  // that leave SIMDs idle anyway (the step of this kernel is 1 + 2d network passes long, all latency at small batches)
  constexpr int rpw = HALF ? 32 : 64;
  // Coordinate split (small batches, exact divergence): the step is 2 network passes + d tangent recursions in ONE wave, and up to
  // 8192 trajectories there are fewer 32-row tiles than CUs.  With csplit = 4 the workgroup's four waves carry the same 32
  // trajectories: each repeats the inference network's base pass and the state update (identical results) and takes a share of the
  // tangents; wave 0 alone evaluates the generative control (target score + network pass) and therefore takes its tangents last
  // (coordinate jt belongs to wave (jt + 1) & 3).  The per-coordinate diagonal entries and the control meet in LDS, where every wave
  // sums the former in coordinate order -- the sum the single wave forms, bit for bit.  Wave 0 stores.
  const int wave = tid >> 6;
  const int csplit = (HALF && A.csplit > 1 && A.div_noise == nullptr && L2.n_hidden < kTanCache && !A.half) ? A.csplit : 1;  // (cached-tangent branch only)
  const long long wave_row0 = csplit > 1 ? (long long)blockIdx.x * rpw : (long long)blockIdx.x * (4 * rpw) + wave * rpw;
  const long long row = wave_row0 + lane;
  const bool live = lane < rpw && row < A.batch && (csplit == 1 || wave == 0);
  const long long lrow = (lane < rpw && row < A.batch) ? row : A.batch - 1;
  if (wave_row0 >= A.batch) return;  // whole wave (with csplit: whole workgroup) out of range
  float* __restrict__ dj_lds = lds + L.lds_floats + L2.lds_floats + (A.lay.k_max > 0 ? A.lay.k_max : 0) * 256;  // [2][2][d][32]: J_jj | u
  const bool gen_here = csplit == 1 || wave == 0;  // this wave evaluates the generative control

  const int d = PAD ? A.d : DP;
  float x[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    const float v = x0[lrow * d + (PAD ? min(j, d - 1) : j)];
    x[j] = (!PAD || j < d) ? v : 0.0f;
  }
  float rnd = 0.0f;
  if (A.flags & SDEH_FLAG_INIT_LOGP) rnd = dgauss_logp<DP>(ws + L.dg[2], x);
  if (xs != nullptr && live) {
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (!PAD || j < d) xs[lrow * d + j] = x[j];
  }

  const int flags = A.flags, ctrl_kind = A.ctrl_kind, act = A.act;
  const DensArgs tgt = A.target;
  const bool lv = flags & SDEH_FLAG_CHANGE_SDE_CTRL;
  const bool need_t = ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool inf_lerp = A.inf_kind == SDEH_CTRL_LERP_PRIOR;
  const bool need_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR || inf_lerp;

  for (int i = 0; i < A.n_steps; ++i) {
    cfp cf = as_const(ws + L.coef + i * kCoefStride);
    const float dt = cf[CF_DT], sqdt = cf[CF_SQDT], sig = cf[CF_SIGMA];

    // ---- generative control u -----------------------------------------------------------------------------------
    float tsc[DP], psc[DP];
    if (need_t && gen_here) target_score<DP>(tgt, ws, lds, L, 0, d, lg_lds, x, tsc);
    if (need_p) dgauss_score<DP>(ws + L.dg[1], x, psc);
    float u[DP];
    if (gen_here) {
      float sterm[DP];
      ctrl_score_term<DP>(ctrl_kind, A, L, ws, i, cf, sig, tsc, psc, sterm);
      SDEH_FENCE();
      mlp_forward<DP, C, HALF>(lds, L, act, ws + L.emb + i * C, x, u, lane);
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        u[j] = clipf(u[j], A.clip_model) + sterm[j];
        if (PAD) u[j] = j < d ? u[j] : 0.0f;
      }
      if (csplit > 1) {  // for the other waves (read behind the step's barrier)
        float* __restrict__ ub = dj_lds + ((i & 1) * 2 + 1) * d * 32;
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if ((!PAD || j < d) && lane < 32) ub[j * 32 + lane] = u[j];
      }
    }
    SDEH_FENCE();

    // ---- inference control v and its exact divergence ------------------------------------------------------------------
    float v[DP];
    float div = 0.0f;
    float eps2[DP];  // eps_j^2 (Hutchinson) or 1: weight of the score term's diagonal derivative
#pragma unroll
    for (int j = 0; j < DP; ++j) eps2[j] = 1.0f;
    if (A.div_noise != nullptr) {  // Hutchinson estimate eps^T J eps with the given probe vectors (utils/autograd.py:25-42)
      float eps[DP], tv[DP], dummy;
      const float* __restrict__ ep = A.div_noise + ((long long)i * A.batch + lrow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        eps[j] = (!PAD || j < d) ? ep[PAD ? min(j, d - 1) : j] : 0.0f;
        eps2[j] = eps[j] * eps[j];
      }
      mlp_forward_tangent<DP, C, true, HALF>(lds2, L2, A.inf_act, ws2 + L2.emb + i * C, nullptr, nullptr, x, v, dummy, lane, eps, tv);
#pragma unroll
      for (int j = 0; j < DP; ++j)
        div += (v[j] >= -A.inf_clip_model && v[j] <= A.inf_clip_model) ? eps[j] * tv[j] : 0.0f;
    } else if (HALF && L2.n_hidden < kTanCache && !A.half) {  // 32-row tiles: act'(z_l) of the base pass stays in registers
      if constexpr (HALF) {
        f32x16 dc[kTanCache][C / 32];
        mlp_forward_dcache<DP, C>(lds2, L2, A.inf_act, ws2 + L2.emb + i * C, x, v, lane, dc);
        if (csplit > 1) {
          float* __restrict__ dj = dj_lds + (i & 1) * 2 * d * 32;  // two buffers: one barrier per step orders writes and reads
          for (int jt = (wave + 3) & 3; jt < d; jt += 4) {  // wave 1: 0, 4, ..; wave 2: 1, 5, ..; wave 3: 2, ..; wave 0: 3, 7, ..
            const float djj = mlp_tangent_cached<C>(lds2, L2, ws2 + L2.tan_in + jt * C, ws2 + L2.tan_out + jt * C, dc, lane);
            float vj = 0.0f;
#pragma unroll
            for (int k = 0; k < DP; ++k) vj = k == jt ? v[k] : vj;
            if (lane < 32) dj[jt * 32 + lane] = (vj >= -A.inf_clip_model && vj <= A.inf_clip_model) ? djj : 0.0f;
            SDEH_FENCE();
          }
          ws_barrier();
          for (int jt = 0; jt < d; ++jt) div += dj[jt * 32 + (lane & 31)];
          if (wave != 0) {
            const float* __restrict__ ub = dj + d * 32;
#pragma unroll
            for (int j = 0; j < DP; ++j) u[j] = (!PAD || j < d) ? ub[(PAD ? min(j, d - 1) : j) * 32 + (lane & 31)] : 0.0f;
          }
        } else
        for (int jt = 0; jt < d; ++jt) {
          const float djj = mlp_tangent_cached<C>(lds2, L2, ws2 + L2.tan_in + jt * C, ws2 + L2.tan_out + jt * C, dc, lane);
          float vj = 0.0f;
#pragma unroll
          for (int k = 0; k < DP; ++k) vj = k == jt ? v[k] : vj;
          div += (vj >= -A.inf_clip_model && vj <= A.inf_clip_model) ? djj : 0.0f;
          SDEH_FENCE();
        }
      }
    } else
    for (int jt = 0; jt < d; ++jt) {  // one forward-mode tangent per coordinate
      float djj;
      mlp_forward_tangent<DP, C, false, HALF>(lds2, L2, A.inf_act, ws2 + L2.emb + i * C, ws2 + L2.tan_in + jt * C,
                                 ws2 + L2.tan_out + jt * C, x, v, djj, lane);
      float vj = 0.0f;
#pragma unroll
      for (int k = 0; k < DP; ++k) vj = k == jt ? v[k] : vj;
      // d clip(v_j, -m, m) / d v_j = 1 on [-m, m] (torch.clamp's backward), else 0
      div += (vj >= -A.inf_clip_model && vj <= A.inf_clip_model) ? djj : 0.0f;
      SDEH_FENCE();
    }
    if (inf_lerp) {  // LerpPriorCtrl (reparam.py:165-178,149-162): v += sigma [scale clip((1 - t/T) prior_score(x)) gamma(t)]
      const float w1 = 1.0f - cf[CF_W];
      cfp gam = as_const(ws2 + L2.gam + i * L2.g);
      cf2p ptab = as_const2(ws + L.dg[1]);  // (mu, 1/sigma^2): prior_score_j = (mu_j - x_j) / sigma_j^2
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const float g = L2.g == 1 ? gam[0] : gam[j];
        const float sc = w1 * psc[j];
        const bool inside = sc >= -A.inf_clip_score && sc <= A.inf_clip_score;
        v[j] = clipf(v[j], A.inf_clip_model) + sig * ((A.inf_scale_score * clipf(sc, A.inf_clip_score)) * g);
        const float dsc = inside ? -(w1 * ptab[j].y) : 0.0f;
        if (!PAD || j < d) div = fmaf(sig * A.inf_scale_score * g, dsc * eps2[j], div);
        if (PAD) v[j] = j < d ? v[j] : 0.0f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        v[j] = clipf(v[j], A.inf_clip_model);
        if (PAD) v[j] = j < d ? v[j] : 0.0f;
      }
    }
    rnd = fmaf(sig * div, dt, rnd);  // losses/oc.py:199-200

    // ---- running cost on gen_plus_inf = u + v, gen_minus_inf = u - v (losses/oc.py:201-211) -------------------------------
    float gp[DP];
    float cost = 0.0f;
#pragma unroll
    for (int j = 0; j < DP; ++j) gp[j] = u[j] + v[j];
    if (lv) {
#pragma unroll
      for (int j = 0; j < DP; ++j) cost = fmaf(gp[j], u[j] - 0.5f * (u[j] - v[j]), cost);
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) cost = fmaf(gp[j], gp[j], cost);
      cost *= 0.5f;
    }
    rnd = fmaf(cost, dt, rnd);
    if (!(flags & SDEH_FLAG_TRAIN)) rnd -= cf[CF_DDIV];
    if (A.gp != nullptr && live) {
      float* __restrict__ gpp = A.gp + ((long long)i * A.batch + lrow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (!PAD || j < d) gpp[j] = gp[j];
    }

    // ---- Gaussian draw, Euler-Maruyama step driven by u, Ito term on u + v (losses/oc.py:213-219) --------------------------
    SDEH_FENCE();
    float itosum = 0.0f;
    const float c_x = fmaf(cf[CF_DRIFT], dt, 1.0f), c_u = sig * dt, c_n = sig * sqdt;
    const float* __restrict__ np = noise != nullptr ? noise + ((long long)i * A.batch + lrow) * d : nullptr;
    const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);
    const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
#pragma unroll
    for (int jb = 0; jb < (DP + 3) / 4; ++jb) {
      float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (np != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * jb + q < DP) n[q] = np[PAD ? min(4 * jb + q, d - 1) : 4 * jb + q];
      } else if (!PAD || 4 * jb < d) {
        box_muller4(philox_block(A.seed, rng_off, grow, i, jb), n);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 4 * jb + q;
        if (j < DP) {
          itosum = fmaf(gp[j], n[q], itosum);
          x[j] = fmaf(c_n, n[q], fmaf(c_u, u[j], c_x * x[j]));
        }
      }
      SDEH_FENCE();
    }
    if (flags & SDEH_FLAG_ITO) rnd = fmaf(itosum, sqdt, rnd);
    if (PAD) {
#pragma unroll
      for (int j = 0; j < DP; ++j) x[j] = j < d ? x[j] : 0.0f;
    }
    if (xs != nullptr && live) {
      float* __restrict__ xp = xs + ((long long)(i + 1) * A.batch + lrow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (!PAD || j < d) xp[j] = x[j];
    }
  }

  if (flags & SDEH_FLAG_TERMINAL_TARGET) rnd -= clipf(target_logp<DP>(tgt, ws, lds, L, 0, d, lg_lds, x), A.clip_target);
  if (live) {
    rnd_out[row] = rnd;
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (!PAD || j < d) xT[row * d + j] = x[j];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Gradient of the divergence term w.r.t. the inference network (method "lv": the trajectory is constant in the graph).
// loss contribution of row n = (t, i):   sum_j c_j J_jj(x_n),   c_j = w_i sigma_t dt_t 1[|v_nn,j| <= clip_model],
// J_jj = w_out[j] . da_L^j,  da_l^j = act'(z_l) dz_l^j,  dz_{l+1}^j = W_{l+1} da_l^j,  dz_0^j = W_in[:, j].
// Reverse mode over that forward-mode computation, one tangent at a time:
//   adj(da_L^j) = c_j w_out[j];   adj(dz_l^j) = act'(z_l) adj(da_l^j);   adj(da_{l-1}^j) = W_l^T adj(dz_l^j)
//   S_l += act''(z_l) dz_l^j adj(da_l^j)        (how the base pre-activations enter: through act')
// then ONE base chain  adj(z_L) = S_L,  adj(z_{l-1}) = S_{l-1} + act'(z_{l-1}) W_l^T adj(z_l).
// All layer quantities go to coordinate-major planes; the weight gradients are GEMMs over N (losses/_autograd.py).
// ---------------------------------------------------------------------------------------------------------
template <int DP, int C, bool PAD>
__global__ __launch_bounds__(256) void bridge_div_bwd_kernel(const BridgeBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const WsLayout& L = A.lay;
  const WsLayout& L2 = A.lay2;
  const float* __restrict__ ws = A.ws;
  const float* __restrict__ ws2 = A.ws2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  {
    const float4* src = reinterpret_cast<const float4*>(ws2);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < L2.lds_floats / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const long long B = A.batch;
  const long long tiles_per_t = (B + 63) / 64;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  if (tile >= tiles_per_t * A.n_steps) return;
  const int t = (int)(tile / tiles_per_t);
  const long long i0 = (tile % tiles_per_t) * 64;
  const int nrows = (int)(B - i0 < 64 ? B - i0 : 64);
  const long long N = B * A.n_steps, n0 = (long long)t * B + i0;
  const bool live = lane < nrows;
  const long long irow = live ? i0 + lane : B - 1;
  const int d = PAD ? A.d : DP, act = A.act, Lh = L2.n_hidden;
  const long long plane = (long long)C * N, pset = (long long)(Lh + 1) * plane;
  cfp cf = as_const(ws + L.coef + t * kCoefStride);
  const float sig = cf[CF_SIGMA], dt = cf[CF_DT];
  const float c0 = live ? A.grad_rnd[irow] * sig * dt : 0.0f;

  // ---- v_nn = out_layer(act(z_L)) for the clamp mask; x for the score part ------------------------------------------
  float vnn[DP];
  {
    f32x16 aA[OT], aB[OT];
    load_plane<OT>(A.zt + (long long)Lh * plane, N, n0, nrows, lane, aA, aB);
    activate<OT>(aA, aB, act);
    f32x16 uA[OTD], uB[OTD];
#pragma unroll
    for (int tt = 0; tt < OTD; ++tt) uA[tt] = uB[tt] = load16(lds + L2.b_out + (tt * 2 + h) * 16);
    const float* w = lds + L2.w_out + lane;
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int tt = 0; tt < OTD; ++tt) {
          const float a = w[((it * 16 + q) * OTD + tt) * 64];
          uA[tt] = SDEH_MFMA(a, aA[it][q], uA[tt]);
          uB[tt] = SDEH_MFMA(a, aB[it][q], uB[tt]);
          if (tt == OTD - 1 && (q & 1)) SDEH_FENCE();
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v0 = uA[r / 16][r % 16];
      float v1 = uB[r / 16][r % 16];
      swap32(v0, v1);
      vnn[mdim(r, 0)] = v0;
      if (mdim(r, 1) < DP) vnn[mdim(r, 1)] = v1;
    }
  }
  // Hutchinson probe vector of this row (training with div_estimator): ONE tangent in direction eps instead of d unit ones
  const bool vec = A.eps != nullptr;
  float eps[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j)
    eps[j] = (vec && (!PAD || j < d)) ? A.eps[((long long)t * B + irow) * d + (PAD ? min(j, d - 1) : j)] : 1.0f;
  // ---- score part of the divergence: only gamma(t) carries parameters ----------------------------------------------------
  if (A.inf_kind == SDEH_CTRL_LERP_PRIOR) {
    const float w1 = 1.0f - cf[CF_W];
    cf2p ptab = as_const2(ws + L.dg[1]);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const float xj = (!PAD || j < d) ? A.xs[((long long)t * B + irow) * d + (PAD ? min(j, d - 1) : j)] : 0.0f;
      const float sc = w1 * (ptab[j].x - xj) * ptab[j].y;
      const bool inside = sc >= -A.clip_score && sc <= A.clip_score;
      const float dsc = (inside && (!PAD || j < d)) ? -(w1 * ptab[j].y) : 0.0f;
      const float gj = c0 * sig * A.scale_score * dsc * (vec ? eps[j] * eps[j] : 1.0f);  // d (w_i sigma dt div) / d gamma_j
      if (L2.g == 1) s += gj;
      else if ((!PAD || j < d) && live) A.dgam[(long long)j * N + n0 + lane] = gj;
    }
    if (L2.g == 1 && live) A.dgam[n0 + lane] = s;
  }

  // ---- one tangent at a time (exact: d unit directions; Hutchinson: the single direction eps) ---------------------------------
  const int ntan = vec ? 1 : d;
  for (int jt = 0; jt < ntan; ++jt) {
    float* tzj = A.tz + (long long)jt * pset;
    float* taj = A.ta + (long long)jt * pset;
    float* tdj = A.td + (long long)jt * pset;
    f32x16 tA[OT], tB[OT];
    float cA = 0.0f, cB = 0.0f;
    float adj_out[DP];  // Hutchinson: adjoint of the tangent's output vector, c0 eps_j 1[|v_nn,j| <= clip_model]
    if (!vec) {
      float vj = 0.0f;
#pragma unroll
      for (int k = 0; k < DP; ++k) vj = k == jt ? vnn[k] : vj;
      const float c = (vj >= -A.clip_model && vj <= A.clip_model) ? c0 : 0.0f;
      if (live) A.cj[(long long)jt * N + n0 + lane] = c;
      cA = __shfl(c, lane & 31);
      cB = __shfl(c, 32 + (lane & 31));
      // forward: dz_0 = W_in[:, jt];  da_l = act'(z_l) dz_l;  dz_{l+1} = W_{l+1} da_l
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) tA[ot] = tB[ot] = load16(ws2 + L2.tan_in + jt * C + (ot * 2 + h) * 16);
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        adj_out[j] = (vnn[j] >= -A.clip_model && vnn[j] <= A.clip_model && (!PAD || j < d)) ? c0 * eps[j] : 0.0f;
        if (live && (!PAD || j < d)) A.cj[(long long)j * N + n0 + lane] = adj_out[j];
      }
      // forward: dz_0 = W_in eps
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) tA[ot][q] = tB[ot][q] = 0.0f;
      float ea[R], eb[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float e0 = (!PAD || mdim(r, 0) < d) ? eps[mdim(r, 0)] : 0.0f;
        float e1 = (mdim(r, 1) < DP && (!PAD || mdim(r, 1) < d)) ? eps[mdim(r, 1)] : 0.0f;
        swap32(e0, e1);
        ea[r] = e0;
        eb[r] = e1;
      }
      const float* w = lds + L2.w_in + lane;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const float a = w[(r * OT + ot) * 64];
          tA[ot] = SDEH_MFMA(a, ea[r], tA[ot]);
          tB[ot] = SDEH_MFMA(a, eb[r], tB[ot]);
          if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
        }
    }
    for (int l = 0; l <= Lh; ++l) {
      f32x16 zA[OT], zB[OT];
      load_plane<OT>(A.zt + (long long)l * plane, N, n0, nrows, lane, zA, zB);
      store_plane<OT>(tzj + (long long)l * plane, N, n0, nrows, lane, tA, tB);
      SDEH_ACT_SWITCH(act, ACT,
        _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)
          _Pragma("unroll") for (int q = 0; q < 16; ++q) {
            tA[ot][q] *= act_grad(zA[ot][q], ACT);
            tB[ot][q] *= act_grad(zB[ot][q], ACT);
          });
      store_plane<OT>(taj + (long long)l * plane, N, n0, nrows, lane, tA, tB);
      if (l < Lh) {
        f32x16 nA[OT], nB[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int q = 0; q < 16; ++q) nA[ot][q] = nB[ot][q] = 0.0f;
        const float* w = lds + L2.w_hid + l * L2.w_hid_stride + lane;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
              const float a = w[((it * 16 + q) * OT + ot) * 64];
              nA[ot] = SDEH_MFMA(a, tA[it][q], nA[ot]);
              nB[ot] = SDEH_MFMA(a, tB[it][q], nB[ot]);
              if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
            }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) { tA[ot] = nA[ot]; tB[ot] = nB[ot]; }
      }
    }
    // reverse: adj(da_L) = c w_out[jt]   (Hutchinson: W_out^T adj_out)
    f32x16 gA[OT], gB[OT];
    if (!vec) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const f32x16 wr = load16(ws2 + L2.tan_out + jt * C + (ot * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 16; ++q) { gA[ot][q] = cA * wr[q]; gB[ot][q] = cB * wr[q]; }
      }
    } else {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) gA[ot][q] = gB[ot][q] = 0.0f;
      float ga[R], gb[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float v0 = adj_out[mdim(r, 0)];
        float v1 = mdim(r, 1) < DP ? adj_out[mdim(r, 1)] : 0.0f;
        swap32(v0, v1);
        ga[r] = v0;
        gb[r] = v1;
      }
      const float* w = lds + L2.wt_out + lane;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const float a = w[(r * OT + ot) * 64];
          gA[ot] = SDEH_MFMA(a, ga[r], gA[ot]);
          gB[ot] = SDEH_MFMA(a, gb[r], gB[ot]);
          if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
        }
    }
    for (int l = Lh; l >= 0; --l) {
      f32x16 zA[OT], zB[OT], dzA[OT], dzB[OT], sA[OT], sB[OT];
      load_plane<OT>(A.zt + (long long)l * plane, N, n0, nrows, lane, zA, zB);
      load_plane<OT>(tzj + (long long)l * plane, N, n0, nrows, lane, dzA, dzB);
      if (jt > 0) load_plane<OT>(A.d2 + (long long)l * plane, N, n0, nrows, lane, sA, sB);
      SDEH_ACT_SWITCH(act, ACT,
        _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)
          _Pragma("unroll") for (int q = 0; q < 16; ++q) {
            const float s0 = act_grad2(zA[ot][q], ACT) * dzA[ot][q] * gA[ot][q];
            const float s1 = act_grad2(zB[ot][q], ACT) * dzB[ot][q] * gB[ot][q];
            sA[ot][q] = jt > 0 ? sA[ot][q] + s0 : s0;
            sB[ot][q] = jt > 0 ? sB[ot][q] + s1 : s1;
            gA[ot][q] *= act_grad(zA[ot][q], ACT);
            gB[ot][q] *= act_grad(zB[ot][q], ACT);
          });
      store_plane<OT>(A.d2 + (long long)l * plane, N, n0, nrows, lane, sA, sB);
      store_plane<OT>(tdj + (long long)l * plane, N, n0, nrows, lane, gA, gB);
      if (l > 0) {
        f32x16 pA[OT], pB[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int q = 0; q < 16; ++q) pA[ot][q] = pB[ot][q] = 0.0f;
        const float* w = lds + L2.wt_hid + (l - 1) * L2.w_hid_stride + lane;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
              const float a = w[((it * 16 + q) * OT + ot) * 64];
              pA[ot] = SDEH_MFMA(a, gA[it][q], pA[ot]);
              pB[ot] = SDEH_MFMA(a, gB[it][q], pB[ot]);
              if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
            }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) { gA[ot] = pA[ot]; gB[ot] = pB[ot]; }
      }
    }
  }

  // ---- the base chain: adj(z_{l-1}) = S_{l-1} + act'(z_{l-1}) W_l^T adj(z_l) ------------------------------------------------
  f32x16 bA[OT], bB[OT];
  load_plane<OT>(A.d2 + (long long)Lh * plane, N, n0, nrows, lane, bA, bB);
  for (int l = Lh; l >= 1; --l) {
    f32x16 pA[OT], pB[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) pA[ot][q] = pB[ot][q] = 0.0f;
    const float* w = lds + L2.wt_hid + (l - 1) * L2.w_hid_stride + lane;
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const float a = w[((it * 16 + q) * OT + ot) * 64];
          pA[ot] = SDEH_MFMA(a, bA[it][q], pA[ot]);
          pB[ot] = SDEH_MFMA(a, bB[it][q], pB[ot]);
          if (ot == OT - 1 && (q & 1)) SDEH_FENCE();
        }
    f32x16 zA[OT], zB[OT];
    load_plane<OT>(A.zt + (long long)(l - 1) * plane, N, n0, nrows, lane, zA, zB);
    load_plane<OT>(A.d2 + (long long)(l - 1) * plane, N, n0, nrows, lane, bA, bB);
    SDEH_ACT_SWITCH(act, ACT,
      _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
          bA[ot][q] = fmaf(act_grad(zA[ot][q], ACT), pA[ot][q], bA[ot][q]);
          bB[ot][q] = fmaf(act_grad(zB[ot][q], ACT), pB[ot][q], bB[ot][q]);
        });
    store_plane<OT>(A.d2 + (long long)(l - 1) * plane, N, n0, nrows, lane, bA, bB);
  }
  // ---- d / d x_t of the divergence term: W_in^T adj(z_0)  (kl: joins the back-propagation through time) ----------------------
  if (A.dx != nullptr) {
    f32x16 xA[OTD], xB[OTD];
#pragma unroll
    for (int tt = 0; tt < OTD; ++tt)
#pragma unroll
      for (int q = 0; q < 16; ++q) xA[tt][q] = xB[tt][q] = 0.0f;
    const float* w = lds + L2.wt_in + lane;
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int tt = 0; tt < OTD; ++tt) {
          const float a = w[((it * 16 + q) * OTD + tt) * 64];
          xA[tt] = SDEH_MFMA(a, bA[it][q], xA[tt]);
          xB[tt] = SDEH_MFMA(a, bB[it][q], xB[tt]);
          if (tt == OTD - 1 && (q & 1)) SDEH_FENCE();
        }
    float dxv[DP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v0 = xA[r / 16][r % 16];
      float v1 = xB[r / 16][r % 16];
      swap32(v0, v1);
      dxv[mdim(r, 0)] = v0;
      if (mdim(r, 1) < DP) dxv[mdim(r, 1)] = v1;
    }
    if (live) {
      float* __restrict__ row = A.dx + ((long long)t * B + irow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (!PAD || j < d) row[j] += dxv[j];
    }
  }
}

template <int DP, int C, bool PAD>
int launch_bridge_div_bwd(const BridgeBwdArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)a.lay2.lds_floats * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};  // the raised LDS limit is a per-device function attribute
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_div_bwd_kernel<DP, C, PAD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  const long long tiles = ((a.batch + 63) / 64) * a.n_steps;
  hipLaunchKernelGGL((bridge_div_bwd_kernel<DP, C, PAD>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

template <int DP, int C, bool PAD>
int launch_bridge(const TrajArgs& a, hipStream_t stream) {
  const int k_scratch = a.lay.k_max > 0 ? a.lay.k_max : 0;
  const size_t lds_bytes = ((size_t)a.lay.lds_floats + (size_t)a.lay2.lds_floats + (size_t)k_scratch * 256) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};  // the raised LDS limit is a per-device function attribute
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_kernel<DP, C, PAD, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_kernel<DP, C, PAD, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  // 32 trajectories per wave (HALF) while that still leaves a SIMD per wave: the step is a chain of dependent network passes, so
  // small batches are latency-bound and halving the MFMA chain is worth more than filling the lanes
  const char* tiles = plan_opt(OPT_BRIDGE_TILES);  // testing aid, a plan option: "64" = 64-row tiles, "32g" = 32-row, generic tangents
  // exact divergence with act' kept in registers (32-row tiles only): fewer MFMA passes per row than the 64-row tiles at
  // every batch size (d = 2: 6.6 vs 8.0 ms at B = 65 536; d = 10: 13.4 vs 33.3 ms)
  const bool cached = a.div_noise == nullptr && a.lay2.n_hidden < kTanCache;
  if (((a.batch <= 32 * 1024 || cached) && !(tiles && tiles[0] == '6')) || (tiles && tiles[0] == '3')) {
    TrajArgs b = a;
    b.half = tiles && tiles[0] == '3' && tiles[2] == 'g' ? 1 : 0;  // here: 1 = do not keep act' in registers
    // coordinate split (kernel header): while the 32-row tiles are fewer than the CUs.
    // SDEH_BRIDGE_SPLIT=1 | 4 forces it (a plan option: tests compare the two).
    const char* split = plan_opt(OPT_BRIDGE_SPLIT);
    const size_t split_bytes = lds_bytes + (size_t)4 * a.d * 32 * sizeof(float);  // [2 steps][J_jj | u][d][32]
    b.csplit = 1;
    if (cached && !b.half && split_bytes <= 160 * 1024 &&
        (split != nullptr ? split[0] == '4' : (a.batch <= 32 * 256 || (a.batch <= 64 * 256 && a.d >= 10))))  // (two rounds pay from d = 10 on)
      b.csplit = 4;
    const unsigned grid = b.csplit > 1 ? (unsigned)((a.batch + 31) / 32) : (unsigned)((a.batch + 127) / 128);
    hipLaunchKernelGGL((bridge_kernel<DP, C, PAD, true>), dim3(grid), dim3(256), b.csplit > 1 ? split_bytes : lds_bytes, stream, a.ws,
                       a.x0, a.noise, a.xT, a.rnd, a.xs, b);
  } else {
    const unsigned grid = (unsigned)((a.batch + 255) / 256);
    hipLaunchKernelGGL((bridge_kernel<DP, C, PAD, false>), dim3(grid), dim3(256), lds_bytes, stream, a.ws, a.x0, a.noise,
                       a.xT, a.rnd, a.xs, a);
  }
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
