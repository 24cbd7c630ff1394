// Plain Euler-Maruyama integrator with output-time interpolation: the device side of sdeh_integrate
// (include/sdeh.h), replacing EulerIntegrator.integrate / interpolate (eq/integrator.py:66-77,93-127) for
// LangevinSDE (eq/sdes.py:38-65), bare OU processes and ControlledSDE (eq/sdes.py:272-305).
//
// One trajectory per lane, the state in registers for the whole integration (T layout); the optional control network
// runs on the MFMA path of the single-wave kernel (mlp_forward).  HBM traffic = x_init in, n_out states out: the
// reference writes (and torch.cat's) one [B,d] tensor per step and interpolates afterwards; here only the requested
// output times are ever written.
#pragma once
#include "sdeh_traj_ws.hpp"

namespace sdeh {

// torch.lerp(a, b, w): a + w (b - a) for w < 0.5, else b - (b - a)(1 - w)
__device__ __forceinline__ float lerp_torch(float a, float b, float w) {
  const float diff = b - a;
  return w < 0.5f ? a + w * diff : b - diff * (1.0f - w);
}

// HALF: 32 trajectories per wave (lanes 0..31 = MFMA column tile A), used with a control network at batches that leave SIMDs
// idle: the step is one dependent network pass, so halving the MFMA chain is worth more than filling the lanes.
template <int DP, int C, bool PAD, bool HALF>
__global__ __launch_bounds__(256) void integrate_kernel(const float* __restrict__ ws, const float* __restrict__ x0,
                                                        const float* __restrict__ noise, float* __restrict__ out,
                                                        const TrajArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WsLayout& L = A.lay;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int ctrl_kind = A.ctrl_kind;
  const bool langevin = A.int_kind == SDEH_INT_LANGEVIN;
  const bool net = ctrl_kind != SDEH_CTRL_NONE;

  const int gmmv = L.gmm_lds;  // 1 / 2: mixture tables inside the LDS image (general / shared-scale form); 0: global memory
  if (net || gmmv != 0) {  // packed weights + mixture tables (LDS image)
    const float4* src = reinterpret_cast<const float4*>(ws);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < L.lds_floats / 4; i += 256) dst[i] = src[i];
  }
  float* lg_lds = lds + L.lds_floats + tid;  // [K][256] mixture-logit scratch, column = thread
  __syncthreads();

  constexpr int rpw = HALF ? 32 : 64;  // trajectories per wave
  const long long row = (long long)blockIdx.x * (4 * rpw) + (tid >> 6) * rpw + lane;
  const bool live = lane < rpw && row < A.batch;
  const long long lrow = live ? row : A.batch - 1;
  if ((long long)blockIdx.x * (4 * rpw) + (tid >> 6) * rpw >= A.batch) return;  // whole wave out of range

  const int d = PAD ? A.d : DP;
  float x[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    const float v = x0[lrow * d + (PAD ? min(j, d - 1) : j)];
    x[j] = (!PAD || j < d) ? v : 0.0f;
  }

  const DensArgs tgt = A.target;
  const bool need_t = langevin || ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool need_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR;
  const int* __restrict__ cnt = reinterpret_cast<const int*>(ws + L.out_cnt);
  const float* __restrict__ tso = A.ts_out;
  const size_t plane = (size_t)A.batch * d;

  for (int i = 0; i < A.n_steps; ++i) {
    cfp cf = as_const(ws + L.coef + i * kCoefStride);
    const float s = cf[CF_S], dt = cf[CF_DT], sqdt = cf[CF_SQDT], sig = cf[CF_SIGMA], fco = cf[CF_DRIFT];

    // ---- drift --------------------------------------------------------------------------------------------
    float dr[DP];
    float tsc[DP], psc[DP];
    if (need_t) {
      // mixtures: streamed from LDS with an online softmax when the tables fit, else two passes over scalar loads
      if (tgt.kind == SDEH_DENS_GMM && gmmv != 0) ws_target_score<DP, DP>(tgt, ws, lds, L, gmmv, d, x, tsc);
      else target_score<DP>(tgt, ws, lds, L, 0, d, lg_lds, x, tsc);
    }
    if (need_p) dgauss_score<DP>(ws + L.dg[1], x, psc);
    if (langevin) {  // eq/sdes.py:53-61: clip(target_score(x) * diff_coeff**2 / 2.0, clip_score)
      const float s2 = sig * sig;
#pragma unroll
      for (int j = 0; j < DP; ++j) dr[j] = clipf(tsc[j] * s2 / 2.0f, A.clip_score);
    } else if (net) {  // eq/sdes.py:297-305: sde.drift(t, x) + sde.diff(t, x) * ctrl(t', x)
      float sterm[DP], u[DP];
      ctrl_score_term<DP>(ctrl_kind, A, L, ws, i, cf, sig, tsc, psc, sterm);
      SDEH_FENCE();
      mlp_forward<DP, C, HALF>(lds, L, A.act, ws + L.emb + i * C, x, u, lane);
#pragma unroll
      for (int j = 0; j < DP; ++j) dr[j] = fmaf(sig, clipf(u[j], A.clip_model) + sterm[j], fco * x[j]);
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) dr[j] = fco * x[j];
    }
    SDEH_FENCE();

    // ---- xt = xs + drift (t - s) + diff * (randn sqrt(t - s))   (eq/integrator.py:113-118), streamed ---------------
    const int c0 = cnt[i], c1 = cnt[i + 1];
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
          const float xt = (x[j] + dr[j] * dt) + sig * (n[q] * sqdt);
          if (live && (!PAD || j < d)) {
            // outputs falling into this step: lerp(xs, xt, (t_eval - s) / (t - s))   (eq/integrator.py:66-77)
            for (int jo = c0; jo < c1; ++jo)
              out[(size_t)jo * plane + (size_t)lrow * d + j] = lerp_torch(x[j], xt, (tso[jo] - s) / dt);
          }
          x[j] = (!PAD || j < d) ? xt : 0.0f;
        }
      }
      SDEH_FENCE();
    }
  }
}

template <int DP, int C, bool PAD>
int launch_integrate(const TrajArgs& a, hipStream_t stream) {
  const int k_scratch = (a.lay.k_max > 0 && a.lay.gmm_lds == 0) ? a.lay.k_max : 0;
  const size_t lds_bytes = ((size_t)a.lay.lds_floats + (size_t)k_scratch * 256) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};  // the raised LDS limit is a per-device function attribute
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&integrate_kernel<DP, C, PAD, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&integrate_kernel<DP, C, PAD, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  if (a.ctrl_kind != SDEH_CTRL_NONE && a.batch <= 32 * 1024) {
    const unsigned grid = (unsigned)((a.batch + 127) / 128);
    hipLaunchKernelGGL((integrate_kernel<DP, C, PAD, true>), dim3(grid), dim3(256), lds_bytes, stream, a.ws, a.x0, a.noise,
                       a.xs, a);
  } else {
    const unsigned grid = (unsigned)((a.batch + 255) / 256);
    hipLaunchKernelGGL((integrate_kernel<DP, C, PAD, false>), dim3(grid), dim3(256), lds_bytes, stream, a.ws, a.x0, a.noise,
                       a.xs, a);
  }
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
