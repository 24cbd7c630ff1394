// Per-call preparation kernel: everything that depends on time only (or on parameters only) is computed once
// per simulate() call for all T steps, so the trajectory kernel's inner loop reads small tables:
//   * per-step scalar coefficients of the SDE / exponential integrator   (eq/sdes.py, losses/oc.py:428-431)
//   * FourierMLP.timestep_embed(t) + input_embed.bias for every step      (models/mlp.py:71-82,115-119)
//   * gamma(t) = clip(score_model(t), clip_model) for every step          (models/reparam.py:68-76)
//   * the packed MFMA-operand image of the network weights (LDS image), GMM / Gaussian parameter tables
// The reference recomputes the time embedding for B identical rows at every step (~1/3 of its MLP time).
#include "sdeh_common.hpp"

namespace sdeh {

__device__ __forceinline__ float actf(float v, int act) {
  if (act == SDEH_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == SDEH_ACT_SILU) return v / (1.0f + expf(-v));
  return fmaxf(v, 0.0f);
}

__device__ __forceinline__ int morder(int c) {  // channel -> index in the M-layout ordering
  const int ot = c >> 5, rr = c & 31;
  const int h = (rr >> 2) & 1, q = (rr & 3) + 4 * (rr >> 3);
  return (ot * 2 + h) * 16 + q;
}

// torch.lerp(a, b, w): a + w (b - a) for w < 0.5, else b - (b - a)(1 - w)
__device__ __forceinline__ float lerpf(float a, float b, float w) {
  const float diff = b - a;
  return w < 0.5f ? a + w * diff : b - diff * (1.0f - w);
}

// TimeEmbed.forward for one scalar t, computed cooperatively by the block (models/mlp.py:71-82).
// Result (dim_out values) is left in `res` (shared).  sh_in: 2C, sh_a/sh_b: C floats.
__device__ void time_embed_block(const SdehTimeEmbed& te, int act, float t, float* sh_in, float* sh_a, float* sh_b,
                                 float* res) {
  const int C = te.channels, tid = threadIdx.x, nt = blockDim.x;
  for (int c = tid; c < C; c += nt) {
    const float arg = te.coeff[c] * t + te.phase[c];
    sh_in[c] = sinf(arg);
    sh_in[C + c] = cosf(arg);
  }
  __syncthreads();
  // every output is a dot product of length C or 2C: four adjacent lanes share one (k = part, part + 4, ...) and combine
  // with two shuffles -- all 256 threads work on the 64 outputs of a layer instead of 64 of them (launch latency of the
  // whole prep kernel: 35 -> ~15 us, which is what an evaluation at the reference's batch sizes waits for)
  const int part = tid & 3, grp = tid >> 2, ngrp = nt >> 2;
  auto dot4 = [&](const float* w, const float* v, int n) {
    float acc = 0.0f;
    for (int k = part; k < n; k += 4) acc = fmaf(w[k], v[k], acc);
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    return acc;
  };
  for (int c = grp; c < C; c += ngrp) {
    const float acc = dot4(te.hidden_w[0] + (size_t)c * 2 * C, sh_in, 2 * C);
    if (part == 0) sh_a[c] = actf(acc + te.hidden_b[0][c], act);
  }
  __syncthreads();
  float* cur = sh_a;
  float* nxt = sh_b;
  for (int l = 1; l < te.n_hidden; ++l) {
    for (int c = grp; c < C; c += ngrp) {
      const float acc = dot4(te.hidden_w[l] + (size_t)c * C, cur, C);
      if (part == 0) nxt[c] = actf(acc + te.hidden_b[l][c], act);
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  for (int o = grp; o < te.dim_out; o += ngrp) {
    const float acc = dot4(te.out_w + (size_t)o * C, cur, C);
    if (part == 0) res[o] = acc + te.out_b[o];
  }
  __syncthreads();
}

__device__ void pack_diag_gauss(float* tab, const SdehDensity& D, int dp, int gid, int stride) {
  if (D.kind != SDEH_DENS_DIAG_GAUSS) return;
  for (int j = gid; j < dp; j += stride) {
    const bool ok = j < D.dim;
    const float s = ok ? D.scale[j] : 1.0f;
    tab[2 * j] = ok ? D.loc[j] : 0.0f;
    tab[2 * j + 1] = ok ? 1.0f / (s * s) : 0.0f;
  }
  if (gid == 0) {
    // the normaliser -sum_j (log sigma_j + log(2 pi) / 2), accumulated in double: summed term by term in fp32 it carried up to ~3e-4 of
    // rounding at d = 196 (|c| = 180: 18 ulp) straight into every trajectory's rnd -- the reference forms it as ONE product for an
    // isotropic Gaussian (distr/gauss.py:215-220), i.e. to one ulp
    double c = 0.0;
    for (int j = 0; j < D.dim; ++j) c -= log((double)D.scale[j]) + 0.91893853320467274178;
    tab[2 * dp] = (float)c;
  }
}

constexpr int kPackBlocks = 32;

__global__ __launch_bounds__(256) void prep_kernel(const PrepArgs P) {
  __shared__ float sh_in[512], sh_a[256], sh_b[256], sh_res[256];
  const WsLayout& L = P.lay;
  const SdehProblem& pr = P.prob;
  float* ws = P.ws;
  const int tid = threadIdx.x;
  const int T = P.n_steps;

  if ((int)blockIdx.x < T) {
    // ----------------------------------------------------------------- per-step tables for step i
    const int i = blockIdx.x;
    const float s = P.ts[i], t = P.ts[i + 1];
    if (tid == 0) {
      float* cf = ws + L.coef + i * kCoefStride;
      const float dt = t - s;
      float sigma = 0.0f, drift = 0.0f, ddiv = 0.0f, w = 0.0f;
      // generative=False (SDEH_FLAG_INFERENCE_SDE, sdeh_integrate only): sign -1, VP schedule runs min -> max
      const bool inf = pr.flags & SDEH_FLAG_INFERENCE_SDE;
      const float sgn = inf ? -1.0f : 1.0f;
      if (pr.sde_kind == SDEH_SDE_VP) {  // eq/sdes.py:222-245 (generative: lerp(max, min, t/T), sign +1)
        const float b0 = inf ? pr.vp_beta_min : pr.vp_beta_max, b1 = inf ? pr.vp_beta_max : pr.vp_beta_min;
        const float bs = lerpf(b0, b1, s / pr.terminal_t);
        const float bt = lerpf(b0, b1, t / pr.terminal_t);
        sigma = pr.vp_scale * sqrtf(bs);
        drift = sgn * 0.5f * bs;
        ddiv = sgn * 0.25f * (bt + bs) * dt * (float)pr.base_model.dim;
      } else if (pr.sde_kind == SDEH_SDE_CONST_OU) {  // eq/sdes.py:141-155
        sigma = pr.ou_diff;
        drift = sgn * pr.ou_drift;
        ddiv = sgn * pr.ou_drift * dt * (float)pr.base_model.dim;
      }
      // the control's clock: ControlledSDE hands terminal_t - t to its ctrl when the sde is not generative (sdes.py:301-303)
      if (pr.sde_kind != SDEH_SDE_NONE) w = (inf ? pr.terminal_t - s : s) / pr.terminal_t;
      const float bk = fminf(fmaxf(pr.exp_alpha * sqrtf(dt), 0.0f), 1.0f);  // losses/oc.py:429-430
      cf[CF_S] = s; cf[CF_T] = t; cf[CF_DT] = dt; cf[CF_SQDT] = sqrtf(dt);
      cf[CF_SIGMA] = sigma; cf[CF_DRIFT] = drift; cf[CF_DDIV] = ddiv; cf[CF_W] = w;
      cf[CF_BETAK] = bk; cf[CF_ALPHAK] = sqrtf(1.0f - bk * bk);
      cf[CF_B2S2] = bk * bk * (pr.exp_sigma * pr.exp_sigma);
      cf[CF_SBK] = pr.exp_sigma * bk;
      for (int k = 12; k < kCoefStride; ++k) cf[k] = 0.0f;
      if (P.ts_out != nullptr) {
        // output schedule of EulerIntegrator.integrate (eq/integrator.py:120-122): step i emits the output times in
        // (.., t_i+1 + eps] not emitted earlier; out_cnt[i] = #{ts_out <= timesteps[i] + eps} (0 for i = 0)
        int* cnt = reinterpret_cast<int*>(ws + L.out_cnt);
        const float lim = t + P.eps;
        int lo = 0, hi = P.n_out;  // first index with ts_out > lim
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (P.ts_out[mid] <= lim) lo = mid + 1; else hi = mid;
        }
        cnt[i + 1] = lo;
        if (i == 0) cnt[0] = 0;
      }
    }
    if (pr.ctrl_kind == SDEH_CTRL_NONE) return;  // no network, no gamma
    const int act = pr.base_model.activation;
    const float s_ctrl = (pr.flags & SDEH_FLAG_INFERENCE_SDE) ? pr.terminal_t - s : s;
    // FourierMLP.timestep_embed(s) + input_embed.bias, stored in M order
    time_embed_block(pr.base_model.timestep_embed, act, s_ctrl, sh_in, sh_a, sh_b, sh_res);
    for (int c = tid; c < L.c; c += blockDim.x)
      ws[L.emb + i * L.c + morder(c)] = sh_res[c] + pr.base_model.input_b[c];
    __syncthreads();
    // gamma(s)
    if (pr.ctrl_kind != SDEH_CTRL_CLIPPED) {
      if (pr.score_model.n_hidden > 0) {
        time_embed_block(pr.score_model, act, s_ctrl, sh_in, sh_a, sh_b, sh_res);
        for (int o = tid; o < L.g; o += blockDim.x) {
          float v = o < pr.score_model.dim_out ? sh_res[o] : 0.0f;
          v = fminf(fmaxf(v, -pr.clip_model), pr.clip_model);
          ws[L.gam + i * L.g + o] = v;
        }
      } else {
        for (int o = tid; o < L.g; o += blockDim.x) ws[L.gam + i * L.g + o] = 1.0f;  // score_model=None
      }
    }
    return;
  }

  // ------------------------------------------------------------------- parameter packing
  const int gid = ((int)blockIdx.x - T) * 256 + tid;
  const int stride = kPackBlocks * 256;
  const SdehFourierMLP& net = pr.base_model;
  const int d = net.dim, C = L.c, OT = L.ot, OTD = L.otd;

  if (pr.ctrl_kind != SDEH_CTRL_NONE && L.wide) {
    // wide networks: natural k order, four k-steps per float4 (see WsLayout)
    auto pack = [&](float* dst, const float* W, int ld, int n_rows, int n_cols, int n_tiles, int n_groups, bool transposed) {
      // dst[((S * n_tiles + t) * 64 + lane) * 4 + e] = M[32 t + (lane & 31)][8 S + 2 e + (lane >> 5)],  M = W or W^T
      const int total = n_groups * n_tiles * 256;
      for (int idx = gid; idx < total; idx += stride) {
        const int e = idx & 3, lane = (idx >> 2) & 63, t = (idx >> 8) % n_tiles, S = (idx >> 8) / n_tiles;
        const int row = 32 * t + (lane & 31), col = 8 * S + 2 * e + (lane >> 5);
        float v = 0.0f;
        if (row < n_rows && col < n_cols) v = transposed ? W[(size_t)col * ld + row] : W[(size_t)row * ld + col];
        dst[idx] = v;
      }
    };
    pack(ws + L.w_in, net.input_w, d, C, d, OT, L.dp8 / 8, false);        // input_embed.weight [C, d]
    for (int l = 0; l < L.n_hidden; ++l) {
      pack(ws + L.w_hid + l * L.w_hid_stride, net.hidden_w[l], C, C, C, OT, C / 8, false);  // hidden_layer[l].weight [C, C]
      for (int c = gid; c < C; c += stride) ws[L.b_hid + l * C + morder(c)] = net.hidden_b[l][c];
      if (L.wt_hid >= 0) pack(ws + L.wt_hid + l * L.w_hid_stride, net.hidden_w[l], C, C, C, OT, C / 8, true);
    }
    pack(ws + L.w_out, net.out_w, C, d, C, OTD, C / 8, false);            // out_layer.weight [d, C]
    if (L.wt_out >= 0) {  // training backward: out_layer^T (rows = channels, k = coordinates), input_embed^T (rows = coordinates, k = channels)
      pack(ws + L.wt_out, net.out_w, C, C, d, OT, L.dp8 / 8, true);
      pack(ws + L.wt_in, net.input_w, d, d, C, OTD, C / 8, true);
    }
    for (int c = gid; c < OTD * 32; c += stride) ws[L.b_out + morder(c)] = c < d ? net.out_b[c] : 0.0f;
    if (L.tan_in >= 0) {
      for (int e = gid; e < d * C; e += stride) {
        const int jt = e / C, ch = e % C;
        const int o = (ch >> 3) * 8 + (ch & 1) * 4 + ((ch & 7) >> 1);
        ws[L.tan_in + jt * C + o] = net.input_w[(size_t)ch * d + jt];
        ws[L.tan_out + jt * C + o] = net.out_w[(size_t)jt * C + ch];
      }
    }
  } else if (pr.ctrl_kind != SDEH_CTRL_NONE) {
  for (int e = gid; e < L.r_in * OT * 64; e += stride) {  // input_embed.weight [C, d]
    const int lane = e & 63, ot = (e >> 6) % OT, r = (e >> 6) / OT;
    const int dimidx = mdim(r, lane >> 5);
    ws[L.w_in + e] = dimidx < d ? net.input_w[(size_t)(32 * ot + (lane & 31)) * d + dimidx] : 0.0f;
  }
  for (int l = 0; l < L.n_hidden; ++l) {  // hidden_layer[l].weight [C, C]
    for (int e = gid; e < (C / 2) * OT * 64; e += stride) {
      const int lane = e & 63, ot = (e >> 6) % OT, sidx = (e >> 6) / OT;
      const int chan = mdim(sidx, lane >> 5);
      ws[L.w_hid + l * L.w_hid_stride + e] = net.hidden_w[l][(size_t)(32 * ot + (lane & 31)) * C + chan];
    }
    for (int c = gid; c < C; c += stride) ws[L.b_hid + l * C + morder(c)] = net.hidden_b[l][c];
  }
  for (int e = gid; e < (C / 2) * OTD * 64; e += stride) {  // out_layer.weight [d, C]
    const int lane = e & 63, t = (e >> 6) % OTD, sidx = (e >> 6) / OTD;
    const int chan = mdim(sidx, lane >> 5);
    const int row = 32 * t + (lane & 31);
    ws[L.w_out + e] = row < d ? net.out_w[(size_t)row * C + chan] : 0.0f;
  }
  for (int c = gid; c < OTD * 32; c += stride) ws[L.b_out + morder(c)] = c < d ? net.out_b[c] : 0.0f;
  if (L.w_out4 >= 0) {  // the out layer's 4 x 4 x 1 operand images, one per pass (layout: sdeh_common.hpp, WsLayout::w_out4)
    for (int ps = 0; ps < out4_passes(L.dp); ++ps) {
      const int GP = out4_pass_groups(L.dp, ps), g_first = out4_pass_first(L.dp, ps), n4 = 33 * GP;
      float* img = ws + L.w_out4 + out4_pass_offset(L.dp, ps);
      for (int e = gid; e < out4_pass_floats(L.dp, ps); e += stride) {
        const int q4 = e / 256, ln = (e / 4) % 64, el = e % 4;  // read as b128: register 4 q4 + el of lane ln
        const int n = 8 * (4 * q4 + el) + (ln % 32) / 4, i = ln % 4, hh = ln / 32;
        float v = 0.0f;
        if (n < n4) {
          const int slot = n / GP, row = 4 * (g_first + n % GP) + i;
          if (row < d) v = slot < 32 ? net.out_w[(size_t)row * C + 32 * (slot / 16) + rho(slot % 16, hh)] : (hh == 0 ? net.out_b[row] : 0.0f);
        }
        img[e] = v;
      }
    }
  }
  if (L.tan_in >= 0) {  // forward-mode tangent seeds / read-outs of the Bridge divergence (sdeh_bridge.hpp)
    for (int e = gid; e < L.dp * C; e += stride) {
      const int jt = e / C, c = e % C;
      ws[L.tan_in + jt * C + morder(c)] = jt < d ? net.input_w[(size_t)c * d + jt] : 0.0f;
      ws[L.tan_out + jt * C + morder(c)] = jt < d ? net.out_w[(size_t)jt * C + c] : 0.0f;
    }
  }
  if (L.wt_out >= 0) {  // transposed images for the backward kernel: A operand rows = INPUT channel of the layer
    for (int e = gid; e < L.r_in * OT * 64; e += stride) {  // out_layer^T: k runs over the coordinates
      const int lane = e & 63, ot = (e >> 6) % OT, r = (e >> 6) / OT;
      const int dimidx = mdim(r, lane >> 5);
      ws[L.wt_out + e] = dimidx < d ? net.out_w[(size_t)dimidx * C + 32 * ot + (lane & 31)] : 0.0f;
    }
    for (int e = gid; e < (C / 2) * OTD * 64; e += stride) {  // input_embed^T: rows = coordinates, k over channels
      const int lane = e & 63, t = (e >> 6) % OTD, sidx = (e >> 6) / OTD;
      const int row = 32 * t + (lane & 31);
      ws[L.wt_in + e] = row < d ? net.input_w[(size_t)mdim(sidx, lane >> 5) * d + row] : 0.0f;
    }
    for (int l = 0; l < L.n_hidden; ++l)
      for (int e = gid; e < (C / 2) * OT * 64; e += stride) {  // hidden_layer[l]^T: k runs over the OUTPUT channels
        const int lane = e & 63, ot = (e >> 6) % OT, sidx = (e >> 6) / OT;
        const int cout = mdim(sidx, lane >> 5);
        ws[L.wt_hid + l * L.w_hid_stride + e] = net.hidden_w[l][(size_t)cout * C + 32 * ot + (lane & 31)];
      }
  }

  }  // ctrl_kind != NONE

  // GMM tables (distr/gauss.py:123-135 via torch.distributions.MixtureSameFamily)
  if (pr.target.kind == SDEH_DENS_GMM && L.wide) {
    if (L.k_max > 0) {  // wide kernels: mu[K][d4], a = 1 / (2 sigma^2) [K][d4] (zero beyond d), c[K]
      const SdehDensity& G = pr.target;
      const int K = G.n_components, rs = L.gmm_row;
      for (int e = gid; e < K * rs; e += stride) {
        const int k = e / rs, j = e % rs;
        const bool ok = j < G.dim;
        const float sg = ok ? G.scale[(size_t)k * G.dim + j] : 1.0f;
        ws[L.gmm_lg + e] = ok ? G.loc[(size_t)k * G.dim + j] : 0.0f;
        ws[L.gmm_sc + e] = ok ? 0.5f / (sg * sg) : 0.0f;
      }
      for (int k = gid; k < K; k += stride) {
        float wsum = 0.0f;
        if (G.mixture_weights != nullptr)
          for (int q = 0; q < K; ++q) wsum += G.mixture_weights[q];
        float c = G.mixture_weights != nullptr ? logf(G.mixture_weights[k]) - logf(wsum) : 0.0f;
        for (int j = 0; j < G.dim; ++j) c -= logf(G.scale[(size_t)k * G.dim + j]) + 0.91893853320467274178f;
        ws[L.gmm_c + k] = c;
      }
    }
  } else if (pr.target.kind == SDEH_DENS_GMM) {
    const SdehDensity& G = pr.target;
    const int K = G.n_components;
    const int K2 = L.gmm_rows;  // K rounded up to a multiple of 8; padding rows have logit -inf
    if (L.gmm_lds >= 3) {  // A-operand images of the matrix-pipe mixture (layout: sdeh_common.hpp, WsLayout::gmm_mm1)
      const bool gen = L.gmm_lds == 4;  // per-component scales: two tables per stream (instruction n carries table t = (n / stride) % 2)
      const int K4 = K2 / 4, D4 = (L.dp + 3) / 4, f = gen ? 2 : 1;
      const int n1 = f * L.dp * K4, n2 = f * K2 * D4;
      // t = 0: mu / sigma^2 (shared form: the only table);  general form: logits t = 0: -1 / (2 sigma^2), t = 1: mu / sigma^2;
      // score t = 0: mu / sigma^2, t = 1: 1 / sigma^2;  zero outside the mixture
      auto tval = [&](int k, int j, int which) {  // which: 0 mu / s^2, 1 -1 / (2 s^2), 2 1 / s^2
        if (k >= K || j >= G.dim) return 0.0f;
        const float sg = G.scale[gen ? (size_t)k * G.dim + j : j];
        const float iv = 1.0f / (sg * sg);
        if (which == 0) return gen ? G.loc[(size_t)k * G.dim + j] * iv : G.loc[(size_t)k * G.dim + j] / (sg * sg);
        return which == 1 ? -0.5f * iv : iv;
      };
      for (int e = gid; e < ((n1 + 63) / 64) * 256; e += stride) {
        const int q = e / 256, ln = (e / 4) % 64, el = e % 4;
        const int n = 16 * (4 * q + el) + ln / 4, i = ln % 4;
        float v = 0.0f;
        if (n < n1) {
          const int dt = n / K4, g = n % K4;  // dt = d (shared) or 2 d + t
          v = gen ? tval(4 * g + i, dt / 2, dt % 2 == 0 ? 1 : 0) : tval(4 * g + i, dt, 0);
        }
        ws[L.gmm_mm1 + e] = v;
      }
      for (int e = gid; e < ((n2 + 63) / 64) * 256; e += stride) {
        const int q = e / 256, ln = (e / 4) % 64, el = e % 4;
        const int n = 16 * (4 * q + el) + ln / 4, i = ln % 4;
        float v = 0.0f;
        if (n < n2) {
          const int kt = n / D4, g = n % D4;  // kt = k (shared) or 2 k + t
          v = gen ? tval(kt / 2, 4 * g + i, kt % 2 == 0 ? 0 : 2) : tval(kt, 4 * g + i, 0);
        }
        ws[L.gmm_mm2 + e] = v;
      }
    }
    if (L.gmm_lds == 2 || L.gmm_lds == 3) {  // shared-scale tables (SDEH_DENS_FLAG_SHARED_SCALE); rows cover the first gmm_row coordinates
      const int rs = L.gmm_row, rs_full = 4 * ((L.dp + 3) / 4);
      for (int e = gid; e < K2 * rs; e += stride) {
        const int k = e / rs, j = e % rs;
        const bool ok = j < G.dim && k < K;
        const float mu = ok ? G.loc[(size_t)k * G.dim + j] : 0.0f;
        const float sg = ok ? G.scale[j] : 1.0f;
        ws[L.gmm_lg + e] = ok ? mu * (0.70710678118654752440f / sg) : 0.0f;
        ws[L.gmm_sc + e] = ok ? mu / (sg * sg) : 0.0f;
      }
      for (int j = gid; j < rs_full; j += stride) {  // per-coordinate vectors (component 0 stands for all beyond rs)
        const bool ok = j < G.dim;
        const float sg = ok ? G.scale[j] : 1.0f, mu0 = ok ? G.loc[j] : 0.0f;
        ws[L.gmm_vec + j] = ok ? 0.70710678118654752440f / sg : 0.0f;
        ws[L.gmm_vec + rs_full + j] = ok ? 1.0f / (sg * sg) : 0.0f;
        ws[L.gmm_vec + 2 * rs_full + j] = ok ? mu0 * (0.70710678118654752440f / sg) : 0.0f;
        ws[L.gmm_vec + 3 * rs_full + j] = ok ? mu0 / (sg * sg) : 0.0f;
      }
    } else {
      // general tables: per coordinate pair a quad (mu_d, mu_d+1, a_d, a_d+1) / (mu/s^2 at d, d+1, 1/s^2 at d, d+1)
      const int npair = L.gmm_row / 2;  // coordinates per row (even)
      for (int e = gid; e < K2 * npair; e += stride) {
        const int k = e / npair, j = e % npair;
        const bool ok = j < G.dim && k < K;
        const float mu = ok ? G.loc[(size_t)k * G.dim + j] : 0.0f;
        const float sg = ok ? G.scale[(size_t)k * G.dim + j] : 1.0f;
        const float iv = ok ? 1.0f / (sg * sg) : 0.0f;
        const int o = k * L.gmm_row + 4 * (j / 2) + (j & 1);
        ws[L.gmm_lg + o] = mu;
        ws[L.gmm_lg + o + 2] = 0.5f * iv;
        ws[L.gmm_sc + o] = mu * iv;
        ws[L.gmm_sc + o + 2] = iv;
      }
    }
    for (int k = gid; k < K; k += stride) {
      float wsum = 0.0f;
      if (G.mixture_weights != nullptr)
        for (int q = 0; q < K; ++q) wsum += G.mixture_weights[q];
      float c = G.mixture_weights != nullptr ? logf(G.mixture_weights[k]) - logf(wsum) : 0.0f;
      for (int j = 0; j < G.dim; ++j) c -= logf(G.scale[(size_t)k * G.dim + j]) + 0.91893853320467274178f;
      ws[L.gmm_c + k] = c;
      if (L.gmm_lds >= 3) {  // logit of the product form: c_k - sum_j mu_kj^2 / (2 sigma_j^2) + sum_j x_j mu_kj / sigma_j^2 (- a term common to all k)
        double cc = c;
        for (int j = 0; j < G.dim; ++j) {
          const double mu = G.loc[(size_t)k * G.dim + j], sg = G.scale[L.gmm_lds == 4 ? (size_t)k * G.dim + j : j];
          cc -= 0.5 * mu * mu / (sg * sg);
        }
        ws[L.gmm_cc + k] = (float)cc;
      }
    }
    for (int k = K + gid; k < K2; k += stride) {
      ws[L.gmm_c + k] = -INFINITY;
      if (L.gmm_lds >= 3) ws[L.gmm_cc + k] = -INFINITY;
    }
  }
  pack_diag_gauss(ws + L.dg[0], pr.target, L.dp, gid, stride);
  pack_diag_gauss(ws + L.dg[1], pr.prior, L.dp, gid, stride);
  pack_diag_gauss(ws + L.dg[2], pr.second, L.dp, gid, stride);
}

int launch_prep(const PrepArgs& p, hipStream_t stream) {
  hipLaunchKernelGGL(prep_kernel, dim3(p.n_steps + kPackBlocks), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------
// estimator partial statistics (losses/oc.py:72-123), mergeable across blocks and across ranks
// ---------------------------------------------------------------------------------------------------------
struct Stat {
  float n, mean, m2, mx, s1, s2, nf;
};

__device__ __forceinline__ Stat stat_merge(const Stat& a, const Stat& b) {
  if (b.n == 0.0f) { Stat r = a; r.nf = a.nf + b.nf; return r; }
  if (a.n == 0.0f) { Stat r = b; r.nf = a.nf + b.nf; return r; }
  Stat r;
  r.n = a.n + b.n;
  const float delta = b.mean - a.mean;
  r.mean = a.mean + delta * (b.n / r.n);
  r.m2 = a.m2 + b.m2 + delta * delta * (a.n * b.n / r.n);
  r.mx = fmaxf(a.mx, b.mx);
  const float ea = expf(a.mx - r.mx), eb = expf(b.mx - r.mx);
  r.s1 = a.s1 * ea + b.s1 * eb;
  r.s2 = a.s2 * ea * ea + b.s2 * eb * eb;
  r.nf = a.nf + b.nf;
  return r;
}

__device__ __forceinline__ Stat stat_shfl_xor(const Stat& a, int mask) {
  Stat r;
  r.n = __shfl_xor(a.n, mask); r.mean = __shfl_xor(a.mean, mask); r.m2 = __shfl_xor(a.m2, mask);
  r.mx = __shfl_xor(a.mx, mask); r.s1 = __shfl_xor(a.s1, mask); r.s2 = __shfl_xor(a.s2, mask);
  r.nf = __shfl_xor(a.nf, mask);
  return r;
}

__device__ Stat block_reduce(Stat v) {
  __shared__ Stat sh[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = stat_merge(v, stat_shfl_xor(v, m));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  Stat r = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = stat_merge(r, sh[w]);
  return r;
}

// the values here are of `rnd` (mean, m2) and of `-rnd` (mx, s1, s2)
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ rnd, long long n,
                                                             float max_rnd, float* __restrict__ part) {
  Stat v = {0.f, 0.f, 0.f, -INFINITY, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float r = rnd[i];
    // losses/oc.py:50-58: NaN threshold = keep everything; +INF = keep finite rows; else keep rnd < max_rnd
    const bool keep = (max_rnd != max_rnd) || (max_rnd > 3.0e38f ? fabsf(r) <= 3.4028235e38f : r < max_rnd);
    const float kf = keep ? 1.0f : 0.0f;
    const Stat o = {kf, keep ? r : 0.0f, 0.0f, keep ? -r : -INFINITY, kf, kf, 1.0f - kf};
    v = stat_merge(v, o);
  }
  v = block_reduce(v);
  if (threadIdx.x == 0) {
    float* p = part + (size_t)blockIdx.x * 8;
    p[0] = v.n; p[1] = v.mean; p[2] = v.m2; p[3] = v.mx; p[4] = v.s1; p[5] = v.s2; p[6] = v.nf; p[7] = 0.f;
  }
}

__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ part, int nb,
                                                           float* __restrict__ out) {
  Stat v = {0.f, 0.f, 0.f, -INFINITY, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nb; i += 256) {
    const float* p = part + (size_t)i * 8;
    Stat o = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
    v = stat_merge(v, o);
  }
  v = block_reduce(v);
  if (threadIdx.x == 0) {
    out[0] = v.n; out[1] = -v.mean * v.n; out[2] = v.m2; out[3] = v.mx;
    out[4] = v.s1; out[5] = v.s2; out[6] = v.nf; out[7] = 0.f;
  }
}

int launch_reduce(const float* rnd, long long n, float max_rnd, float* part, int nb, float* out,
                  hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partial_kernel, dim3(nb), dim3(256), 0, stream, rnd, n, max_rnd, part);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, stream, part, nb, out);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// The training loss over the batch and its per-row gradient in one pass behind the reduction (losses/oc.py:72-92: rnd[mask].mean() for
// kl, rnd[mask].var() for lv): from stats = [n, sum(-rnd), M2, ...] of the kept rows
//     loss = mean = -stats[1] / n   |   M2 / (n - 1);      w_i = d loss / d rnd_i = 1 / n   |   2 (rnd_i - mean) / (n - 1),   0 on dropped rows
// -- operation for operation what the framework's masked reductions and their autograd compute (the elementwise chain of
// losses/oc.py::_MaskedMoment: same fp32 values).  stats[7] <- loss; *n_filtered += stats[6] (the reference's running count, on the device).
__global__ __launch_bounds__(256) void moment_weights_kernel(const float* __restrict__ rnd, long long n, float max_rnd, int lv,
                                                             float* __restrict__ stats, float* __restrict__ w,
                                                             long long* __restrict__ n_filtered) {
  const float cnt = stats[0];
  const float mean = -stats[1] / cnt;
  const float kl_w = 1.0f / cnt, den = cnt - 1.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float r = rnd[i];
    const bool keep = (max_rnd != max_rnd) || (max_rnd > 3.0e38f ? fabsf(r) <= 3.4028235e38f : r < max_rnd);
    const float per_row = lv ? 2.0f * (r - mean) / den : kl_w;
    w[i] = keep ? per_row : 0.0f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats[7] = lv ? stats[2] / den : mean;
    if (n_filtered != nullptr) *n_filtered += (long long)stats[6];
  }
}

int launch_loss_moment(const float* rnd, long long n, float max_rnd, int lv, long long* n_filtered, float* part, int nb, float* out, float* w,
                       hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partial_kernel, dim3(nb), dim3(256), 0, stream, rnd, n, max_rnd, part);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, stream, part, nb, out);
  const int nbw = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(moment_weights_kernel, dim3(nbw > 0 ? nbw : 1), dim3(256), 0, stream, rnd, n, max_rnd, lv, out, w, n_filtered);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// Undo of a rejected optimisation step (utils/graphs.py: GraphedTrainStep(guard=True), the device-side form of the reference trainer's
// `if loss_ok and grad_ok`, solver/base.py:409-432): every guarded tensor (parameters, optimizer moments, step counters) is put back from
// its snapshot when *ok == 0 -- ONE launch over a table of (destination, snapshot, 32-bit words) instead of two framework kernels per tensor.
__global__ __launch_bounds__(256) void guard_restore_kernel(const unsigned long long* __restrict__ table, const unsigned char* __restrict__ ok,
                                                            long long* __restrict__ n_skipped) {
  if (ok[0] != 0) return;
  if (n_skipped != nullptr && blockIdx.x == 0 && threadIdx.x == 0) n_skipped[0] += 1;
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(table[3 * blockIdx.x]);
  const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(table[3 * blockIdx.x + 1]);
  const unsigned long long n = table[3 * blockIdx.x + 2];
  for (unsigned long long i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

int launch_guard_restore(const unsigned long long* table, int n_tensors, const unsigned char* ok, long long* n_skipped, hipStream_t stream) {
  hipLaunchKernelGGL(guard_restore_kernel, dim3(n_tensors), dim3(256), 0, stream, table, ok, n_skipped);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// The decision itself: *ok = (loss finite, or |loss| <= max_loss when max_loss >= 0) and every gradient finite; a rejected step's gradients
// are zeroed in place (the optimizer step still runs inside a captured graph and must not see NaN / Inf; its result is undone afterwards).
// One workgroup: the control networks have 4e4 .. 4e5 parameters, a few microseconds of loads.
__global__ __launch_bounds__(1024) void guard_check_kernel(float* __restrict__ g, long long n, const float* __restrict__ value, float max_loss,
                                                           unsigned char* __restrict__ ok) {
  __shared__ int bad_any;
  if (threadIdx.x == 0) {
    const float v = value[0];
    bad_any = max_loss >= 0.0f ? !(fabsf(v) <= max_loss) : !(fabsf(v) <= 3.402823466e38f);
  }
  __syncthreads();
  int bad = 0;
  for (long long i = threadIdx.x; i < n; i += 1024) bad |= !(fabsf(g[i]) <= 3.402823466e38f);
  if (bad) atomicOr(&bad_any, 1);
  __syncthreads();
  const int rejected = bad_any;
  if (threadIdx.x == 0) ok[0] = rejected ? 0 : 1;
  if (rejected)
    for (long long i = threadIdx.x; i < n; i += 1024) g[i] = 0.0f;
}

int launch_guard_check(float* g, long long n, const float* value, float max_loss, unsigned char* ok, hipStream_t stream) {
  hipLaunchKernelGGL(guard_check_kernel, dim3(1), dim3(1024), 0, stream, g, n, value, max_loss, ok);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

__global__ __launch_bounds__(256) void weights_kernel(const float* __restrict__ rnd, long long n,
                                                      const float* __restrict__ mx, float* __restrict__ w) {
  const float m = mx[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    w[i] = expf(-rnd[i] - m);
}

int launch_weights(const float* rnd, long long n, const float* mx, float* w, hipStream_t stream) {
  const int nb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(weights_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, stream, rnd, n, mx, w);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
