// Persistent Euler-Maruyama trajectory kernel for gfx950 (MI355X).
//
// One launch integrates all T steps: each wavefront owns 64 trajectories whose state x[d], running cost rnd
// and MLP activations never leave registers; the packed network weights sit in LDS for the whole launch;
// HBM traffic is x0 in, x_T + rnd out (plus noise in parity mode / xs when the trajectory is requested).
//
// Replaces the per-step Python loops of the reference (losses/oc.py:176-222, 301-334, 416-446) together with
// everything they call per step (models/reparam.py forward, models/mlp.py:114-122, eq/sdes.py coefficient
// functions, distr/*.py scores, torch.randn_like) and the terminal log-densities (oc.py:225,337,449-450).
#pragma once
#include <type_traits>

#include "sdeh_common.hpp"

namespace sdeh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Tables that are uniform across the wave are read through the constant address space so that hipcc emits
// scalar loads (s_load_dwordx*) and feeds them to the VALU as SGPR operands.
typedef const float __attribute__((address_space(4))) * cfp;
struct alignas(8) F2 {
  float x, y;
};
typedef const F2 __attribute__((address_space(4))) * cf2p;
__device__ __forceinline__ cfp as_const(const float* p) { return (cfp)(unsigned long long)p; }
__device__ __forceinline__ cf2p as_const2(const float* p) { return (cf2p)(unsigned long long)p; }

__device__ __forceinline__ void swap32(float& a, float& b) {
  // lanes 32..63 of `a` <-> lanes 0..31 of `b`
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// Cross-lane sums without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: a full LDS round trip per butterfly stage, ~700
// cycles for a wave-wide sum).  Rows are the 16-lane groups of the DPP hardware.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, false));
}
// every lane: the sum over the 16 lanes of its row (four v_add_f32_dpp)
__device__ __forceinline__ float sum_row16(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
// every lane: v + (v of lane ^ 16)   (v_permlane16_swap exchanges the odd rows of one operand with the even rows of the other)
__device__ __forceinline__ float sum_xor16(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// every lane: v + (v of lane ^ 32)
__device__ __forceinline__ float sum_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_wave(float v) { return sum_xor32(sum_xor16(sum_row16(v))); }

// clamp(v, -m, m) as one v_med3_f32 (m = +inf: identity; NaN -> -m, as fminf(fmaxf(v, -m), m) gives)
__device__ __forceinline__ float clipf(float v, float m) { return __builtin_amdgcn_fmed3f(v, -m, m); }

// GELU(v) = v Phi(v) = max(v, 0) - |v| Phi(-|v|).  Branch-free: Phi(-t) = 2^q(t) for t = min(|v|, 6) with q a degree-6
// polynomial, the weighted minimax fit of log2 Phi(-t) on [0, 6] (weight = d GELU / d q = t Phi(-t) ln 2, so the fit
// error is an ABSOLUTE error of GELU: 5.1e-8, the size of the fp32 rounding of the evaluation itself).  Measured against
// the exact erf form in fp64: max |error| 1.1e-7 max(1, |v|) -- torch's own fp32 erf GELU has 3.5e-7 (tools/gelu_fit.py;
// tests/test_hip_rng_and_stats.py::test_gelu_accuracy).  10 VALU instructions: v_med3(|v|), 6 v_fma, v_exp, v_max, v_fma.
__device__ __forceinline__ float act_gelu(float v) {
  // v_med3_f32: fminf would add a canonicalising v_max_f32 in front
  const float t = __builtin_amdgcn_fmed3f(fabsf(v), 0.0f, 6.0f);
  float q = 3.3092907814e-05f;
  q = fmaf(q, t, -7.6922050644e-04f);
  q = fmaf(q, t, 8.0807191412e-03f);
  q = fmaf(q, t, -5.3412108121e-02f);
  q = fmaf(q, t, -4.5877097054e-01f);
  q = fmaf(q, t, -1.1512017029e+00f);
  q = fmaf(q, t, -9.9999306093e-01f);
  float relu;  // one v_max_f32: written in C (fmaxf, or med3 with +inf) hipcc adds a canonicalising v_max_f32 v, v, v in front
  // (t as an extra input: orders this read of v -- an MFMA result, and the hazard recogniser does not look into assembly -- behind the
  // v_med3 above, which the compiler has given the wait states)
  asm("v_max_f32 %0, 0, %1" : "=v"(relu) : "v"(v), "v"(t));
  return fmaf(-t, __builtin_amdgcn_exp2f(q), relu);
}
__device__ __forceinline__ float act_silu(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float act_relu(float v) { return fmaxf(v, 0.0f); }

template <int N>
__device__ __forceinline__ void activate(f32x16 (&a)[N], f32x16 (&b)[N], int act) {
  if (act == SDEH_ACT_GELU_ERF) {
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) { a[t][q] = act_gelu(a[t][q]); b[t][q] = act_gelu(b[t][q]); }
  } else if (act == SDEH_ACT_SILU) {
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) { a[t][q] = act_silu(a[t][q]); b[t][q] = act_silu(b[t][q]); }
  } else {
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) { a[t][q] = act_relu(a[t][q]); b[t][q] = act_relu(b[t][q]); }
  }
}

// Runs `...` with ACT a compile-time copy of the (wave-uniform) activation id: the dispatch sits OUTSIDE the element loops.
// Selecting per element makes the compiler branch per element, which serialises the polynomial / v_exp chains of 32-64
// independent elements (measured: 2.5x on the activation-heavy generic kernels).
#define SDEH_ACT_SWITCH(act, ACT, ...)                                                            \
  do {                                                                                            \
    if ((act) == SDEH_ACT_GELU_ERF) { constexpr int ACT = SDEH_ACT_GELU_ERF; __VA_ARGS__ }        \
    else if ((act) == SDEH_ACT_SILU) { constexpr int ACT = SDEH_ACT_SILU; __VA_ARGS__ }           \
    else { constexpr int ACT = SDEH_ACT_RELU; __VA_ARGS__ }                                       \
  } while (0)

template <int ACT>
__device__ __forceinline__ float act_ct(float v) {
  return ACT == SDEH_ACT_GELU_ERF ? act_gelu(v) : (ACT == SDEH_ACT_SILU ? act_silu(v) : act_relu(v));
}

template <int N>
__device__ __forceinline__ void activate_one(f32x16 (&a)[N], int act) {
  SDEH_ACT_SWITCH(act, ACT,
    _Pragma("unroll") for (int t = 0; t < N; ++t)
      _Pragma("unroll") for (int q = 0; q < 16; ++q) a[t][q] = act_ct<ACT>(a[t][q]););
}

__device__ __forceinline__ f32x16 load16(const float* p) {
  f32x16 v;
  const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 t = p4[i];
    v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
  }
  return v;
}

#define SDEH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// Scheduling fence: keeps hipcc from hoisting every LDS weight read of an unrolled layer to its top (which
// made >500 registers live and spilled to scratch).  In-order issue still overlaps the next group's ds_reads
// with the MFMAs already queued on the matrix pipe.
#define SDEH_FENCE() __builtin_amdgcn_sched_barrier(0)

// ---------------------------------------------------------------------------------------------------------
// FourierMLP forward for 64 trajectories (models/mlp.py:114-122).  x, out in the T layout.
// ---------------------------------------------------------------------------------------------------------
template <int DP, int C, bool HALF = false>
__device__ __forceinline__ void mlp_forward(const float* __restrict__ lds, const WsLayout& L, int act,
                                            const float* __restrict__ emb_step, const float (&x)[DP],
                                            float (&out)[DP], int lane) {
  // HALF: only lanes 0..31 carry trajectories (column tile A); tile B's MFMAs and activations are skipped (small batches)
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5;
  f32x16 accA[OT], accB[OT];
  // e = input_embed(x) + timestep_embed(t): start the accumulators at (time embedding + input bias)
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) accA[ot] = accB[ot] = load16(emb_step + (ot * 2 + h) * 16);
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
        if constexpr (!HALF) accB[ot] = SDEH_MFMA(a, xb[r], accB[ot]);
        if (ot == OT - 1 && (r & 1)) SDEH_FENCE();
      }
  }
  f32x16 uA[OTD], uB[OTD];
  for (int l = 0; l <= L.n_hidden; ++l) {
    if constexpr (HALF) activate_one<OT>(accA, act);
    else activate<OT>(accA, accB, act);
    if (l < L.n_hidden) {  // hidden layers: e = layer(act(e))
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
    } else {  // out_layer(act(e))
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
            if (t == OTD - 1 && (q & 1)) SDEH_FENCE();
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
}

// ---------------------------------------------------------------------------------------------------------
// densities in the T layout (one trajectory per lane, parameters wave-uniform -> scalar loads)
// ---------------------------------------------------------------------------------------------------------
// GMM (distr/gauss.py:123-140): log p(x) = logsumexp_k( log w~_k + sum_d Normal(mu_kd, sigma_kd).log_prob(x_d) ).
// Returns the log-density; when SCORE, also d/dx = sum_k r_k (mu_k - x)/sigma_k^2 (what the reference obtains
// by autograd, distr/base.py:130-137).  Logits are parked in LDS between the two passes.
template <int DP, bool SCORE>
__device__ __forceinline__ float gmm_eval_smem(const float* ws, const WsLayout& L, int K, float* __restrict__ lg_lds,
                                               const float (&x)[DP], float (&score)[DP]) {
  // rows of (mu_d, mu_d+1, a_d, a_d+1) quads (the layout the wave-specialised kernel reads as float4)
  cfp plg = as_const(ws + L.gmm_lg);
  cfp pc = as_const(ws + L.gmm_c);
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) {
    cfp p = plg + k * L.gmm_row;
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const float t = x[d] - p[4 * (d / 2) + (d & 1)];
      acc = fmaf(t * t, p[4 * (d / 2) + 2 + (d & 1)], acc);
    }
    const float l = pc[k] - acc;
    lg_lds[k * 256] = l;
    m = fmaxf(m, l);
  }
  float z = 0.0f;
  if constexpr (SCORE) {
    cfp psc = as_const(ws + L.gmm_sc);
    float P[DP], Q[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) P[d] = Q[d] = 0.0f;
    for (int k = 0; k < K; ++k) {
      cfp p = psc + k * L.gmm_row;
      const float e = __expf(lg_lds[k * 256] - m);
      z += e;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        P[d] = fmaf(e, p[4 * (d / 2) + (d & 1)], P[d]);
        Q[d] = fmaf(e, p[4 * (d / 2) + 2 + (d & 1)], Q[d]);
      }
    }
    const float iz = 1.0f / z;
#pragma unroll
    for (int d = 0; d < DP; ++d) score[d] = (P[d] - x[d] * Q[d]) * iz;
  } else {
    for (int k = 0; k < K; ++k) z += __expf(lg_lds[k * 256] - m);
  }
  return m + __logf(z);
}

// Same computation with the tables resident in LDS: every lane reads the same address (broadcast ds_read_b128,
// two (k,d) pairs per read), so the loads can be prefetched deep into VGPRs -- scalar loads cannot (SMEM returns
// out of order, ~100 SGPRs, and with one wave per SIMD nothing hides their L2 latency).
template <int N>
__device__ __forceinline__ void load_f4(const float4* __restrict__ p, float4 (&q)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) q[j] = p[j];
}

template <int DP, bool SCORE>
__device__ __forceinline__ float gmm_eval(const float* ws, const float* lds, const WsLayout& L, int gmmv, int K,
                                          float* __restrict__ lg_lds, const float (&x)[DP], float (&score)[DP]) {
  (void)lds; (void)gmmv;  // the single-wave kernel only handles tables in global memory (scalar loads)
  return gmm_eval_smem<DP, SCORE>(ws, L, K, lg_lds, x, score);
}

// Diagonal Gaussian: table [DP][2] = (mu, 1/sigma^2) followed by the scalar  sum_d(-log sigma_d - 0.5 log 2pi).
template <int DP>
__device__ __forceinline__ float dgauss_logp(const float* tab, const float (&x)[DP]) {
  cf2p p = as_const2(tab);
  float acc = 0.0f;
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    const float t = x[d] - p[d].x;
    acc = fmaf(t * t, p[d].y, acc);
  }
  return as_const(tab)[2 * DP] - 0.5f * acc;
}
template <int DP>
__device__ __forceinline__ void dgauss_score(const float* tab, const float (&x)[DP], float (&s)[DP]) {
  cf2p p = as_const2(tab);
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    s[d] = (p[d].x - x[d]) * p[d].y;
  }
}

// MultiWell / DoubleWell (distr/double_well.py:39-45,165-179)
template <int DP>
__device__ __forceinline__ float mwell_logp(const DensArgs& D, int dreal, const float (&x)[DP]) {
  float acc = 0.0f;
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    const float y = x[d] - D.p1;
    const float w = y * y - D.p0;
    const float v = d < D.n_comp ? w * w : 0.5f * y * y;
    acc += d < dreal ? v : 0.0f;
  }
  return -acc;
}
template <int DP>
__device__ __forceinline__ void mwell_score(const DensArgs& D, int dreal, const float (&x)[DP], float (&s)[DP]) {
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    const float y = x[d] - D.p1;
    const float v = d < D.n_comp ? -4.0f * (y * y - D.p0) * y : -y;
    s[d] = d < dreal ? v : 0.0f;
  }
}

// Funnel (distr/funnel.py:54-80); p0 = variance of the first coordinate
template <int DP>
__device__ __forceinline__ float funnel_logp(const DensArgs& D, int dreal, const float (&x)[DP]) {
  const float x0 = x[0];
  float sq = 0.0f;
#pragma unroll
  for (int d = 1; d < DP; ++d) sq = fmaf(x[d], x[d], sq);  // padded coordinates are held at 0
  const float first = -0.5f * __logf(6.283185307179586f * D.p0) - 0.5f * x0 * x0 / D.p0;
  const float other = -(float)(dreal - 1) * (x0 + 1.8378770664093453f) * 0.5f - 0.5f * sq * __expf(-x0);
  return first + other + D.lnc;
}
template <int DP>
__device__ __forceinline__ void funnel_score(const DensArgs& D, int dreal, const float (&x)[DP], float (&s)[DP]) {
  const float x0 = x[0];
  const float iv = __expf(-x0);
  float sq = 0.0f;
#pragma unroll
  for (int d = 1; d < DP; ++d) {
    sq = fmaf(x[d], x[d], sq);
    s[d] = -x[d] * iv;
  }
  s[0] = -x0 / D.p0 - 0.5f * (float)(dreal - 1) + 0.5f * sq * iv;
}

// target.unnorm_log_prob
template <int DP>
__device__ __forceinline__ float target_logp(const DensArgs& D, const float* ws, const float* lds, const WsLayout& L,
                                             int gmmv, int dreal, float* lg_lds, const float (&x)[DP]) {
  float dummy[DP];
  switch (D.kind) {
    case SDEH_DENS_GMM: return gmm_eval<DP, false>(ws, lds, L, gmmv, D.n_comp, lg_lds, x, dummy) + D.lnc;
    case SDEH_DENS_DIAG_GAUSS: return dgauss_logp<DP>(ws + L.dg[0], x) + D.lnc;
    case SDEH_DENS_MULTI_WELL: return mwell_logp<DP>(D, dreal, x);
    case SDEH_DENS_FUNNEL: return funnel_logp<DP>(D, dreal, x);
    default: return 0.0f;
  }
}

template <int DP>
__device__ __forceinline__ void target_score(const DensArgs& D, const float* ws, const float* lds, const WsLayout& L,
                                             int gmmv, int dreal, float* lg_lds, const float (&x)[DP], float (&s)[DP]) {
  switch (D.kind) {
    case SDEH_DENS_GMM: (void)gmm_eval<DP, true>(ws, lds, L, gmmv, D.n_comp, lg_lds, x, s); break;
    case SDEH_DENS_DIAG_GAUSS: dgauss_score<DP>(ws + L.dg[0], x, s); break;
    case SDEH_DENS_MULTI_WELL: mwell_score<DP>(D, dreal, x, s); break;
    case SDEH_DENS_FUNNEL: funnel_score<DP>(D, dreal, x, s); break;
    default:
#pragma unroll
      for (int d = 0; d < DP; ++d) s[d] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Gaussian draws: Philox4x32-10 words -> Box-Muller.  Block j of a step yields coordinates 4j..4j+3.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void box_muller4(const U4& r, float (&n)[4]) {
  // uniform in (0, 1]: the whole 32-bit word scaled by 2^-32 (v_cvt_f32_u32 rounds it to 24 significant bits) plus 2^-33 so
  // that the logarithm never sees 0 -- one conversion and one fma per uniform
  constexpr float S = 2.3283064365386963e-10f, H = 1.1641532182693481e-10f;
  const float u0 = fmaf((float)r.x, S, H);
  const float u1 = fmaf((float)r.y, S, H);
  const float u2 = fmaf((float)r.z, S, H);
  const float u3 = fmaf((float)r.w, S, H);
  // v_log_f32 is log2:  -2 ln u = -2 ln2 log2 u
  const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));  // v_sqrt_f32 (1 ulp; the argument is never denormal)
  const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  // v_sin_f32 / v_cos_f32 take the angle in revolutions
  n[0] = ra * __builtin_amdgcn_cosf(u1);
  n[1] = ra * __builtin_amdgcn_sinf(u1);
  n[2] = rb * __builtin_amdgcn_cosf(u3);
  n[3] = rb * __builtin_amdgcn_sinf(u3);
}

// Counter = (global row, block, step, offset low word); key = (seed low word, seed high ^ offset high); the global
// row index must fit 32 bits (checked by sdeh_simulate_fwd).  With this arrangement the per-lane products of the
// first two rounds depend on the row only (two values hipcc hoists out of the step loop for ALL blocks), the
// other products of those rounds are wave-uniform (SALU), and the key schedule is uniform -- other arrangements
// make hipcc hoist one product per block (13 x 2 registers) and spill them.
__device__ __forceinline__ U4 philox_block(unsigned long long seed, unsigned long long offset,
                                           unsigned long long row, int step, int block) {
  U4 c;
  c.x = (uint32_t)row;
  c.y = (uint32_t)block;
  c.z = (uint32_t)step;
  c.w = (uint32_t)offset;
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32));
}

// ---------------------------------------------------------------------------------------------------------
// score part of the control:  mult * scale_score * clip(score mix, clip_score) * clip(gamma(t), clip_model)
// (reparam.py:56-83 ScoreCtrl, 131-162 LerpCtrl, 166-178 LerpPriorCtrl, 185-197 LerpTargetCtrl); zeros for ClippedCtrl
// ---------------------------------------------------------------------------------------------------------
template <int DP>
__device__ __forceinline__ void ctrl_score_term(int ctrl_kind, const TrajArgs& A, const WsLayout& L, const float* ws,
                                                int i, cfp cf, float sig, const float (&tsc)[DP],
                                                const float (&psc)[DP], float (&sterm)[DP]) {
  if (ctrl_kind != SDEH_CTRL_CLIPPED) {
    const float w = cf[CF_W];
    if (ctrl_kind == SDEH_CTRL_SCORE) {  // reparam.py:56-83
#pragma unroll
      for (int j = 0; j < DP; ++j) sterm[j] = tsc[j];
    } else if (ctrl_kind == SDEH_CTRL_LERP) {  // reparam.py:131-144; torch.lerp's two-sided formula
      if (w < 0.5f) {
#pragma unroll
        for (int j = 0; j < DP; ++j) sterm[j] = psc[j] + w * (tsc[j] - psc[j]);
      } else {
        const float w1 = 1.0f - w;
#pragma unroll
        for (int j = 0; j < DP; ++j) sterm[j] = tsc[j] - (tsc[j] - psc[j]) * w1;
      }
    } else if (ctrl_kind == SDEH_CTRL_LERP_TARGET) {  // reparam.py:185-197
#pragma unroll
      for (int j = 0; j < DP; ++j) sterm[j] = w * tsc[j];
    } else {  // SDEH_CTRL_LERP_PRIOR, reparam.py:166-178
      const float w1 = 1.0f - w;
#pragma unroll
      for (int j = 0; j < DP; ++j) sterm[j] = w1 * psc[j];
    }
    cfp gam = as_const(ws + L.gam + i * L.g);
    // ScoreCtrl: ctrl + score;  Lerp*: ctrl + sde.diff(t) * score   (reparam.py:78-83,149-162)
    const float mult = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig;
    const float g0 = gam[0];
    if (L.g == 1 && A.scale_score == 1.0f && ctrl_kind == SDEH_CTRL_SCORE) {
      // the shipped ScoreCtrl configurations: 1.0 * ((1.0 * clip) * g0) is one multiply, bit-identical
#pragma unroll
      for (int j = 0; j < DP; ++j) sterm[j] = clipf(sterm[j], A.clip_score) * g0;
    } else if (L.g == 1) {
#pragma unroll
      for (int j = 0; j < DP; ++j) sterm[j] = mult * ((A.scale_score * clipf(sterm[j], A.clip_score)) * g0);
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) sterm[j] = mult * ((A.scale_score * clipf(sterm[j], A.clip_score)) * gam[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j) sterm[j] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// the kernel.  PAD = false: d == DP exactly;  PAD = true: d <= DP, coordinates >= d are held at zero.
// LOSS / CTRL / TGT / GMMV / ACT / REFC >= 0 fix the loss kind, control kind, target density kind, GMM table variant,
// activation and presence of a reference control at compile time (the BASELINE configurations get such specialised variants: no dead branches, far
// fewer live registers); -1 leaves the property a wave-uniform run-time switch (the generic variants).
// ---------------------------------------------------------------------------------------------------------
template <int DP, int C, bool PAD, int LOSS, int CTRL, int TGT, int GMMV, int ACT, int REFC>
__global__ __launch_bounds__(256) void traj_kernel(const float* __restrict__ ws, const float* __restrict__ x0,
                                                   const float* __restrict__ noise, float* __restrict__ xT,
                                                   float* __restrict__ rnd_out, float* __restrict__ xs,
                                                   const TrajArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WsLayout& L = A.lay;
  const int tid = threadIdx.x;
  const int lane = tid & 63;

  // stage the packed weights (LDS image) once per workgroup
  {
    const float4* src = reinterpret_cast<const float4*>(ws);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < L.lds_floats / 4; i += 256) dst[i] = src[i];
  }
  float* lg_lds = lds + L.lds_floats + tid;  // [K][256] logit scratch, column = thread
  __syncthreads();

  const long long row = (long long)blockIdx.x * 256 + tid;
  const bool live = row < A.batch;
  const long long lrow = live ? row : A.batch - 1;  // dead lanes shadow the last row and never store
  if ((long long)blockIdx.x * 256 + (tid & ~63) >= A.batch) return;  // whole wave out of range

  const int d = PAD ? A.d : DP;
  float x[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    const float v = x0[lrow * d + (PAD ? min(j, d - 1) : j)];
    x[j] = (!PAD || j < d) ? v : 0.0f;
  }

  float rnd = 0.0f;
  // Distribution.log_prob = unnorm_log_prob - log_norm_const (distr/base.py:116-119): the constants cancel
  if (A.flags & SDEH_FLAG_INIT_LOGP) rnd = dgauss_logp<DP>(ws + L.dg[2], x);

  if (xs != nullptr && live) {
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (!PAD || j < d) xs[lrow * d + j] = x[j];
  }

  const int flags = A.flags;
  const int ctrl_kind = CTRL >= 0 ? CTRL : A.ctrl_kind, loss_kind = LOSS >= 0 ? LOSS : A.loss_kind;
  const int gmmv = GMMV >= 0 ? GMMV : L.gmm_lds, act = ACT >= 0 ? ACT : A.act;
  DensArgs tgt = A.target;
  if (TGT >= 0) tgt.kind = TGT;
  const bool lv = flags & SDEH_FLAG_CHANGE_SDE_CTRL;
  const bool need_t = ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool refc = REFC >= 0 ? REFC != 0 : (flags & SDEH_FLAG_REFERENCE_CTRL) && loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool need_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR || refc;

  for (int i = 0; i < A.n_steps; ++i) {
    cfp cf = as_const(ws + L.coef + i * kCoefStride);
    const float dt = cf[CF_DT], sqdt = cf[CF_SQDT], sig = cf[CF_SIGMA];

    // ---- generative_ctrl(s, x): score term first (it only needs x), then the network ----------------------
    float tsc[DP], psc[DP];
    if (need_t) target_score<DP>(tgt, ws, lds, L, gmmv, d, lg_lds, x, tsc);
    if (need_p) dgauss_score<DP>(ws + L.dg[1], x, psc);
    float sterm[DP];  // mult * scale_score * clip(score) * clip(gamma(t))
    ctrl_score_term<DP>(ctrl_kind, A, L, ws, i, cf, sig, tsc, psc, sterm);
    SDEH_FENCE();

    float u[DP];
    mlp_forward<DP, C>(lds, L, act, ws + L.emb + i * C, x, u, lane);
#pragma unroll
    for (int j = 0; j < DP; ++j) {  // ClippedCtrl (reparam.py:25-36) + score term
      u[j] = clipf(u[j], A.clip_model) + sterm[j];
      if (PAD) u[j] = j < d ? u[j] : 0.0f;
    }

    SDEH_FENCE();
    // ---- running cost (losses/oc.py:204-211, 319-323, 418-431) -----------------------------------------
    float gm[DP];  // the control entering the Ito term (gen_plus_inf / gen_minus_ref)
    float cost = 0.0f;
    if (refc) {
      if (lv) {
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const float r = sig * psc[j];
          gm[j] = u[j] - r;
          cost = fmaf(gm[j], u[j] - 0.5f * (r + u[j]), cost);
        }
      } else {
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          gm[j] = u[j] - sig * psc[j];
          cost = fmaf(gm[j], gm[j], cost);
        }
        cost *= 0.5f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) gm[j] = u[j];
      if (lv) {
#pragma unroll
        for (int j = 0; j < DP; ++j) cost = fmaf(u[j], u[j] - 0.5f * u[j], cost);
      } else {
#pragma unroll
        for (int j = 0; j < DP; ++j) cost = fmaf(u[j], u[j], cost);
        cost *= 0.5f;
      }
    }
    if (loss_kind == SDEH_LOSS_EXPONENTIAL) rnd = fmaf(cf[CF_B2S2], cost, rnd);
    else rnd = fmaf(cost, dt, rnd);
    if (loss_kind == SDEH_LOSS_TIME_REVERSAL && !(flags & SDEH_FLAG_TRAIN)) rnd -= cf[CF_DDIV];

    // ---- Gaussian draw + state update + Ito term, streamed four coordinates at a time -------------------
    SDEH_FENCE();
    float itosum = 0.0f;
    const bool expo = loss_kind == SDEH_LOSS_EXPONENTIAL;
    // exponential integrator (oc.py:428-443):  x <- x a_k + (b_k^2 s^2) u + (s b_k) xi
    // Euler-Maruyama (oc.py:213-219, 325-331): x <- x + (f x + sig u) dt + sig (xi sqrt(dt))
    const float c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], dt, 1.0f);
    const float c_u = expo ? cf[CF_B2S2] : sig * dt;
    const float c_n = expo ? cf[CF_SBK] : sig * sqdt;
    const float c_i = expo ? cf[CF_SBK] : sqdt;  // Ito term: sum(g * xi) * c_i
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
          itosum = fmaf(gm[j], n[q], itosum);
          x[j] = fmaf(c_n, n[q], fmaf(c_u, u[j], c_x * x[j]));
        }
      }
      SDEH_FENCE();
    }
    if (flags & SDEH_FLAG_ITO) rnd = fmaf(itosum, c_i, rnd);
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

  // ---- terminal costs (oc.py:225, 337, 449-450) ----------------------------------------------------------
  if (flags & SDEH_FLAG_TERMINAL_SECOND) rnd += dgauss_logp<DP>(ws + L.dg[2], x);
  if (flags & SDEH_FLAG_TERMINAL_TARGET) rnd -= clipf(target_logp<DP>(tgt, ws, lds, L, gmmv, d, lg_lds, x), A.clip_target);

  if (live) {
    rnd_out[row] = rnd;
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (!PAD || j < d) xT[row * d + j] = x[j];
  }
}

template <int DP, int C, bool PAD, int LOSS, int CTRL, int TGT, int GMMV, int ACT, int REFC>
int launch_traj(const TrajArgs& a, hipStream_t stream) {
  const int k_scratch = a.lay.k_max > 0 ? a.lay.k_max : 0;
  const size_t lds_bytes = ((size_t)a.lay.lds_floats + (size_t)k_scratch * 256) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};  // the raised LDS limit is a per-device function attribute
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  const unsigned grid = (unsigned)((a.batch + 255) / 256);
  hipLaunchKernelGGL((traj_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC>), dim3(grid), dim3(256), lds_bytes, stream, a.ws, a.x0, a.noise,
                     a.xT, a.rnd, a.xs, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
