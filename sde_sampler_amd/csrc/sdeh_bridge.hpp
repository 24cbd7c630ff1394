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
template <int DP, int C>
__device__ __forceinline__ void mlp_forward_tangent(const float* __restrict__ lds, const WsLayout& L, int act,
                                                    const float* __restrict__ emb_step, const float* __restrict__ tin,
                                                    const float* __restrict__ tout, const float (&x)[DP],
                                                    float (&out)[DP], float& djj, int lane) {
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5;
  f32x16 accA[OT], accB[OT], tA[OT], tB[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    accA[ot] = accB[ot] = load16(emb_step + (ot * 2 + h) * 16);
    tA[ot] = tB[ot] = load16(tin + (ot * 2 + h) * 16);  // d z_0 / d x_jt = W_in[:, jt] for every trajectory
  }
  {
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
        accB[ot] = SDEH_MFMA(a, xb[r], accB[ot]);
        if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
      }
  }
  f32x16 uA[OTD], uB[OTD];
  float sA = 0.0f, sB = 0.0f;
  for (int l = 0; l <= L.n_hidden; ++l) {
    // a_l = act(z_l);  d a_l = act'(z_l) d z_l
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        tA[ot][q] *= act_grad(accA[ot][q], act);
        tB[ot][q] *= act_grad(accB[ot][q], act);
      }
    activate<OT>(accA, accB, act);
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
            nB[ot] = SDEH_MFMA(a, accB[it][q], nB[ot]);
            ntA[ot] = SDEH_MFMA(a, tA[it][q], ntA[ot]);
            ntB[ot] = SDEH_MFMA(a, tB[it][q], ntB[ot]);
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
            uB[t] = SDEH_MFMA(a, accB[it][q], uB[t]);
            if (t == OTD - 1 && (q & 1)) SDEH_FENCE();
          }
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
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float v0 = uA[r / 16][r % 16];
    float v1 = uB[r / 16][r % 16];
    swap32(v0, v1);
    out[mdim(r, 0)] = v0;
    if (mdim(r, 1) < DP) out[mdim(r, 1)] = v1;
  }
  // the other lane half holds the remaining channels of the same trajectory column
  sA += __shfl_xor(sA, 32);
  sB += __shfl_xor(sB, 32);
  djj = lane < 32 ? sA : sB;  // T layout: lane = trajectory (tile A: 0..31, tile B: 32..63)
}

template <int DP, int C, bool PAD>
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

  const long long row = (long long)blockIdx.x * 256 + tid;
  const bool live = row < A.batch;
  const long long lrow = live ? row : A.batch - 1;
  if ((long long)blockIdx.x * 256 + (tid & ~63) >= A.batch) return;

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
    if (need_t) target_score<DP>(tgt, ws, lds, L, 0, d, lg_lds, x, tsc);
    if (need_p) dgauss_score<DP>(ws + L.dg[1], x, psc);
    float u[DP];
    {
      float sterm[DP];
      ctrl_score_term<DP>(ctrl_kind, A, L, ws, i, cf, sig, tsc, psc, sterm);
      SDEH_FENCE();
      mlp_forward<DP, C>(lds, L, act, ws + L.emb + i * C, x, u, lane);
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        u[j] = clipf(u[j], A.clip_model) + sterm[j];
        if (PAD) u[j] = j < d ? u[j] : 0.0f;
      }
    }
    SDEH_FENCE();

    // ---- inference control v and its exact divergence ------------------------------------------------------------------
    float v[DP];
    float div = 0.0f;
    for (int jt = 0; jt < d; ++jt) {  // one forward-mode tangent per coordinate
      float djj;
      mlp_forward_tangent<DP, C>(lds2, L2, A.inf_act, ws2 + L2.emb + i * C, ws2 + L2.tan_in + jt * C,
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
        if (!PAD || j < d) div = fmaf(sig * A.inf_scale_score * g, dsc, div);
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

    // ---- Gaussian draw, Euler-Maruyama step driven by u, Ito term on u + v (losses/oc.py:213-219) --------------------------
    SDEH_FENCE();
    float itosum = 0.0f;
    const float c_x = fmaf(cf[CF_DRIFT], dt, 1.0f), c_u = sig * dt, c_n = sig * sqdt;
    const float* __restrict__ np = noise != nullptr ? noise + ((long long)i * A.batch + lrow) * d : nullptr;
    const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);
#pragma unroll
    for (int jb = 0; jb < (DP + 3) / 4; ++jb) {
      float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (np != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * jb + q < DP) n[q] = np[PAD ? min(4 * jb + q, d - 1) : 4 * jb + q];
      } else if (!PAD || 4 * jb < d) {
        box_muller4(philox_block(A.seed, A.offset, grow, i, jb), n);
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

template <int DP, int C, bool PAD>
int launch_bridge(const TrajArgs& a, hipStream_t stream) {
  const int k_scratch = a.lay.k_max > 0 ? a.lay.k_max : 0;
  const size_t lds_bytes = ((size_t)a.lay.lds_floats + (size_t)a.lay2.lds_floats + (size_t)k_scratch * 256) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_kernel<DP, C, PAD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  const unsigned grid = (unsigned)((a.batch + 255) / 256);
  hipLaunchKernelGGL((bridge_kernel<DP, C, PAD>), dim3(grid), dim3(256), lds_bytes, stream, a.ws, a.x0, a.noise, a.xT,
                     a.rnd, a.xs, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
