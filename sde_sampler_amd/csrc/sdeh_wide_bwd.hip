// Training backward for WIDE control networks (FourierMLP with 128 / 256 channels, or 64 channels at d > 64; d <= 256): what
// `loss.backward()` does in the reference (solver/base.py:407) through the unrolled loops of losses/oc.py:176-222, 301-334, 416-446
// with models/mlp.py:114-122 and models/reparam.py:56-83,131-197 -- for the shapes of BASELINE.json configs[4] (solver=bridge,
// channels = 256: conf/solver/bridge.yaml trains two such networks with loss time_reversal_lv).
//
// Design.  The network's weights (0.9 MB at C = 256, d = 196) exceed a CU's LDS and its weight gradients
// (another 0.9 MB) exceed a CU's registers, so -- unlike the 64-channel fused backward (sdeh_bwdf.hip) -- the work is split where the
// arithmetic intensity allows it:
//   * wide_bwd_kernel: the CHAIN.  A workgroup of four waves owns one column tile of 32 trajectories, channel-split exactly like the
//     forward kernels (sdeh_wide.hip): wave w owns row tiles {w, w + 4} of every layer and coordinate tiles {w, w + 4} of the state.
//     Per (step, tile) it re-evaluates the network at the stored x_t (bit for bit the forward pass: same operand order), keeps
//     act'(z_l) of every layer in LDS planes, forms the upstream gradient of the control in the accumulator layout (clip masks, score
//     term, replayed Philox draws), and walks back through the layers with the TRANSPOSED packed weights streamed from L2 through the
//     same hand-issued operand ring -- a backward layer is the forward layer product on another weight image.  It writes the
//     pre-activations z_l and their adjoints as coordinate-major planes [C][N] (N = T B rows; whole 128-byte lines per store).
//     Row-parallel for the log-variance methods (x_t is a constant of the graph), through time for kl / kl_ito (the adjoint
//     lambda_t = d loss / d x_t stays in the registers of the waves that own its coordinates).
//   * sdeh_weight_grad (sdeh_wgrad.hip, blocked for m, c <= 256): dW_l = sum_n delta_l[:, n] act(z_{l-1}[:, n])^T as a GEMM over N.
//     At C = 256 the planes cost 2 KB per row and layer against 131 kFLOP of contraction (64 FLOP/B: above the HBM ridge), i.e. the
//     plane round trip that dominated the 64-channel design is here a fraction of the matrix time.
//   * wide_bridge_div_bwd_kernel (below): the gradient of the Bridge's divergence term, fused -- its operands exist per (row,
//     coordinate) and can never be written out.
// No implicit contraction in this translation unit (sdeh_wide_common.hpp sets fp contract off): the re-evaluated pre-activations
// must equal the forward launch's bit for bit (a ReLU unit on its kink must take the same side in both passes).
#include "sdeh_wide_common.hpp"

namespace sdeh {

enum WideBwdSlot { WBS_GAM = 0, WBS_CX = 1, WBS_C0 = WSL_DIV, WBS_PRESQ = WSL_PRESQ, WBS_X0 = WSL_X0, WBS_LOGP_A = WSL_LOGP_A,
                   WBS_LOGP_B = WSL_LOGP_B };
static_assert(WSL_PRESQ == 2 && WSL_X0 == 3, "slots 0 / 1 (running cost / Ito partials of the forward kernels) are reused here");

// 16 values of this lane's coordinates (cb + (q & 3) + 8 (q >> 2)) from a row of a row-major [.., d] tensor; zeros beyond d
__device__ __forceinline__ f32x16 wide_row16(const float* __restrict__ rowp, int cb, int d, bool vec4) {
  f32x16 v;
  const float* __restrict__ p = rowp + cb;
  if (vec4) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float4 t4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
      if (cb + 8 * g4 < d) t4 = *reinterpret_cast<const float4*>(p + 8 * g4);
      v[4 * g4] = t4.x; v[4 * g4 + 1] = t4.y; v[4 * g4 + 2] = t4.z; v[4 * g4 + 3] = t4.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = cb + (q & 3) + 8 * (q >> 2) < d ? p[(q & 3) + 8 * (q >> 2)] : 0.0f;
  }
  return v;
}

// this wave's tiles of a layer: z = acc + bias -> HBM plane zt_l [C][N] (column n), act'(z) -> LDS plane dpl [C][32], act(z) -> LDS
// plane out [rows][32]
template <int OTW>
__device__ __forceinline__ void wide_bwd_act_store(f32x16 (&acc)[OTW][1], const f32x16 (&bias)[OTW], int act, float* __restrict__ outl,
                                                   float* __restrict__ dpl, float* __restrict__ zt_col, long long N, bool store, int t0,
                                                   int h) {
  SDEH_ACT_SWITCH(act, ACT,
    _Pragma("unroll") for (int k = 0; k < OTW; ++k) {
      f32x16 z = acc[k][0] + bias[k];
      _Pragma("unroll") for (int q = 0; q < 16; ++q) {
        const int ch = 32 * (t0 + 4 * k) + rho(q, h);
        if (store) zt_col[(long long)ch * N] = z[q];
        if (dpl != nullptr) dpl[ch * 32] = act_grad(z[q], ACT);
      }
      act_tile<ACT>(z);
      _Pragma("unroll") for (int q = 0; q < 16; ++q) outl[(32 * (t0 + 4 * k) + rho(q, h)) * 32] = z[q];
    });
}

// delta_l = acc * act'(z_l) for this wave's tiles -> HBM plane dt_l (column n) and the LDS plane (the next backward layer's input)
// dpl == nullptr (networks whose act' planes do not fit LDS: (n_hidden + 1) C 128 B > ~120 KiB): act'(z_l) is recomputed from the
// pre-activation this lane wrote to zt_l a moment ago (an L2 hit)
template <int OTW>
__device__ __forceinline__ void wide_bwd_delta_store(const f32x16 (&acc)[OTW][1], const float* __restrict__ dpl, const float* __restrict__ zt_col,
                                                     int act, float* __restrict__ outl, float* __restrict__ dt_col, long long N,
                                                     bool store, int t0, int h) {
  if (dpl != nullptr) {
#pragma unroll
    for (int k = 0; k < OTW; ++k)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ch = 32 * (t0 + 4 * k) + rho(q, h);
        const float v = acc[k][0][q] * dpl[ch * 32];
        if (store) dt_col[(long long)ch * N] = v;
        outl[ch * 32] = v;
      }
  } else {
    SDEH_ACT_SWITCH(act, ACT,
      _Pragma("unroll") for (int k = 0; k < OTW; ++k) {
        f32x16 z;
        _Pragma("unroll") for (int q = 0; q < 16; ++q) z[q] = store ? zt_col[(long long)(32 * (t0 + 4 * k) + rho(q, h)) * N] : 0.0f;
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
          const int ch = 32 * (t0 + 4 * k) + rho(q, h);
          const float v = acc[k][0][q] * act_grad(z[q], ACT);
          if (store) dt_col[(long long)ch * N] = v;
          outl[ch * 32] = v;
        }
      });
  }
}

// out[k] = W[coordinate tiles {w, w + 4}][:] . plane   (an out-layer-shaped product: w_out or input_embed^T), nto tiles of this wave
__device__ __forceinline__ void wide_bwd_coord_layer(const float* __restrict__ wl, int otd, int NS4, int w, int lane, int nto,
                                                     const float* __restrict__ actl, f32x16 (&out)[2][1]) {
  unsigned vo[2];
  vo[0] = (unsigned)((w * 64 + lane) * 16);
  vo[1] = nto > 1 ? (unsigned)(((w + 4) * 64 + lane) * 16) : vo[0];
  if (nto == 2) {
    WidePre<2> pre;
    wide_prefetch<2>(pre, wl, otd * 256, NS4, vo);
    wide_layer<2, 1>(pre, wl, otd * 256, NS4, vo, actl, 32, out);
  } else {  // one tile, or none (the wave runs the same stream on tile w % otd and drops the result: no control flow around asm loads)
    WidePre<1> pre;
    unsigned v1[1] = {nto == 1 ? vo[0] : (unsigned)(((w % otd) * 64 + lane) * 16)};
    wide_prefetch<1>(pre, wl, otd * 256, NS4, v1);
    f32x16 o1[1][1];
    wide_layer<1, 1>(pre, wl, otd * 256, NS4, v1, actl, 32, o1);
    out[0][0] = o1[0][0];
  }
}

// ---------------------------------------------------------------------------------------------------------
// The chain kernel.  Semantics (what is a constant of the graph, what is differentiated, which planes are written) are those of
// bwd_tile in sdeh_bwd.hpp, the 64-channel plane kernel; A.nn_in is not used (the wide forward kernels keep no planes).
//   grid: BPTT: one workgroup per column tile of 32 trajectories; row-parallel: n_tiles x step chunks (A.n_steps split into chunks of
//   `spc` steps, blockIdx.x = chunk * n_tiles + tile).
// ---------------------------------------------------------------------------------------------------------
template <int OTW, bool BPTT>
__global__ __launch_bounds__(256) void wide_bwd_kernel(const BwdArgs A, int n_tiles, int spc, int dlds) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RS = 32;
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int d = A.d, OTD = L.otd, C = L.c, OT = L.ot, Lh = L.n_hidden, T = A.n_steps;
  const long long B = A.batch, N = (long long)T * B;
  const int rows = C > 32 * OTD ? C : 32 * OTD;

  WideCtx cx;
  cx.RS = RS; cx.d = d;
  cx.wave = w; cx.lane = lane; cx.j = j; cx.h = h;
  cx.planes = lds; cx.plane_floats = rows * RS;
  float* __restrict__ pl = lds;                              // the one activation / adjoint plane [rows][32]
  float* __restrict__ dplanes = lds + rows * RS;             // act'(z_l), l = 0 .. Lh: [Lh + 1][C][32] (dlds: they fit LDS)
  cx.scr = dplanes + (dlds ? (Lh + 1) * C * RS : 0);         // [kWideSlots][4][32]
  float* tabs = cx.scr + kWideSlots * 4 * RS;
  const int tab_stride = 2 * L.dp + 4;
  for (int i = tid; i < 3 * tab_stride; i += 256) {
    const int which = i / tab_stride, o = i % tab_stride;
    tabs[i] = o <= 2 * L.dp ? ws[L.dg[which] + o] : 0.0f;
  }
  cx.tab0 = tabs; cx.tab1 = tabs + tab_stride; cx.tab2 = tabs + 2 * tab_stride;
  const float* __restrict__ bias_g = ws + L.b_hid;  // hidden biases [Lh][C] then the out-layer bias [32 otd], accumulator order (L2)
  cx.bias = bias_g;

  const int nto = (OTD > w ? 1 : 0) + (OTD > w + 4 ? 1 : 0);
  const int tile = BPTT ? (int)blockIdx.x : (int)blockIdx.x % n_tiles;
  const int chunk = BPTT ? 0 : (int)blockIdx.x / n_tiles;
  const int t_hi = BPTT ? T - 1 : (chunk * spc + spc < T ? chunk * spc + spc : T) - 1;
  const int t_lo = BPTT ? 0 : chunk * spc;
  const long long row0 = (long long)tile * RS;
  const long long r = row0 + j;
  const bool live = r < B;
  const long long lrow = live ? r : B - 1;
  const float wi = live ? A.grad_rnd[lrow] : 0.0f;
  const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);

  const int flags = A.flags, ctrl_kind = A.ctrl_kind, act = A.act;
  const DensArgs tgt = A.target;
  const bool has_score = ctrl_kind != SDEH_CTRL_CLIPPED;
  const bool need_t = ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool refc = (flags & SDEH_FLAG_REFERENCE_CTRL) && A.loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool need_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR || refc;
  const bool expo = A.loss_kind == SDEH_LOSS_EXPONENTIAL;
  const bool ito = (flags & SDEH_FLAG_ITO) != 0;
  const bool vec4 = (d & 3) == 0;
  const bool want_dx = BPTT || A.dx != nullptr;
  const int wh = w & (OT - 1);
  const bool has = w < OT;  // C = 64: waves 2, 3 own no hidden tile (they shadow waves 0, 1 and skip the stores)
  unsigned voff[OTW];
#pragma unroll
  for (int k = 0; k < OTW; ++k) voff[k] = (unsigned)(((wh + 4 * k) * 64 + lane) * 16);

  __syncthreads();  // tables staged

  f32x16 xr[2][1], lam[2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) lam[k][q] = 0.0f;

  auto load_x = [&](int t) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k < nto) xr[k][0] = wide_row16(A.xs + ((long long)t * B + lrow) * d, 32 * (w + 4 * k) + 4 * h, d, vec4);
      else {
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[k][0][q] = 0.0f;
      }
    }
  };

  if constexpr (BPTT) {
    // lambda_T = w_i d(terminal costs)/dx_T  (losses/oc.py:225, 337, 449-450): + second.score(x_T), - 1[|log rho| <= clip] target.score(x_T)
    load_x(T);
    wide_publish<1>(cx, pl, xr, nto);  // the funnel statistics of x_T
    if (flags & SDEH_FLAG_TERMINAL_TARGET) {
      if (tgt.kind == SDEH_DENS_DIAG_GAUSS) wide_gauss_quad<1>(cx, cx.tab0, xr, nto, WBS_LOGP_B);
      else if (tgt.kind == SDEH_DENS_MULTI_WELL) wide_mwell_sum<1>(cx, tgt, xr, nto, WBS_LOGP_B);
    }
    wide_barrier();
    float keep = 1.0f, fs = 0.0f, fx0 = 0.0f, fiv = 0.0f;
    if (flags & SDEH_FLAG_TERMINAL_TARGET) {
      if (tgt.kind == SDEH_DENS_FUNNEL) {
        fs = wide_slot_sum(cx, WBS_PRESQ, j);
        fx0 = cx.scr[(WBS_X0 * 4) * RS + j];
        fiv = __expf(-fx0);
      }
      if (A.clip_target < 3.0e38f) {
        float lp = 0.0f;
        if (tgt.kind == SDEH_DENS_DIAG_GAUSS) lp = cx.tab0[2 * L.dp] - 0.5f * wide_slot_sum(cx, WBS_LOGP_B, j) + tgt.lnc;
        else if (tgt.kind == SDEH_DENS_MULTI_WELL) lp = -wide_slot_sum(cx, WBS_LOGP_B, j);
        else if (tgt.kind == SDEH_DENS_FUNNEL) {
          const float first = -0.5f * __logf(6.283185307179586f * tgt.p0) - 0.5f * fx0 * fx0 / tgt.p0;
          const float other = -(float)(d - 1) * (fx0 + 1.8378770664093453f) * 0.5f - 0.5f * fs * fiv;
          lp = first + other + tgt.lnc;
        }
        keep = fabsf(lp) <= A.clip_target ? 1.0f : 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < nto) {
        const int cb = 32 * (w + 4 * k) + 4 * h;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int cc = cb + (q & 3) + 8 * (q >> 2);
          float v = 0.0f;
          if (flags & SDEH_FLAG_TERMINAL_SECOND) {
            const float2 pp = *reinterpret_cast<const float2*>(cx.tab2 + 2 * cc);
            v = wi * (pp.x - xr[k][0][q]) * pp.y;
          }
          if (flags & SDEH_FLAG_TERMINAL_TARGET)
            v = fmaf(-wi * keep, wide_target_score(tgt, cx.tab0, cc, d, xr[k][0][q], fs, fx0, fiv), v);
          lam[k][q] = cc < d ? v : 0.0f;
        }
      }
    wide_barrier();  // slots read before the first step publishes again
  }

  for (int t = t_hi; t >= t_lo; --t) {
    const long long n = (long long)t * B + lrow;  // this lane's row of the [., N] planes
    cfp cf = as_const(ws + L.coef + t * kCoefStride);
    const float sig = cf[CF_SIGMA], wl = cf[CF_W];
    const float c_i = expo ? cf[CF_SBK] : cf[CF_SQDT];        // dB = c_i xi
    const float cdt = expo ? cf[CF_B2S2] : cf[CF_DT];         // running cost = cdt * (...)
    const float c_u = expo ? cf[CF_B2S2] : sig * cf[CF_DT];   // x_{t+1} = c_x x_t + c_u u_t + c_n xi_t
    const float c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], cf[CF_DT], 1.0f);

    // ======================================================================================= forward (re-evaluation at x_t)
    load_x(t);
    wide_publish<1>(cx, pl, xr, nto);
    wide_barrier();
    float fs = 0.0f, fx0 = 0.0f, fiv = 0.0f;
    if (need_t && tgt.kind == SDEH_DENS_FUNNEL) {
      fs = wide_slot_sum(cx, WBS_PRESQ, j);
      fx0 = cx.scr[(WBS_X0 * 4) * RS + j];
      fiv = __expf(-fx0);
    }
    f32x16 nn[2][1];
    {
      WidePre<OTW> pre_in;
      wide_prefetch<OTW>(pre_in, ws + L.w_in, OT * 256, L.dp8 / 8, voff);
      f32x16 emb[OTW];
#pragma unroll
      for (int k = 0; k < OTW; ++k) emb[k] = load16(ws + L.emb + t * C + ((wh + 4 * k) * 2 + h) * 16);
      f32x16 acc[OTW][1];
      wide_layer<OTW, 1>(pre_in, ws + L.w_in, OT * 256, L.dp8 / 8, voff, pl + h * RS + j, RS, acc);
      for (int l = 0; l <= Lh; ++l) {
        WidePre<OTW> pre_h;
        if (l < Lh) wide_prefetch<OTW>(pre_h, ws + L.w_hid + l * L.w_hid_stride, OT * 256, C / 8, voff);
        f32x16 bias[OTW];
#pragma unroll
        for (int k = 0; k < OTW; ++k) bias[k] = l == 0 ? emb[k] : load16(bias_g + (l - 1) * C + ((wh + 4 * k) * 2 + h) * 16);
        wide_barrier();  // everyone has read the plane that is about to be overwritten
        if (has) wide_bwd_act_store<OTW>(acc, bias, act, pl + j, dlds ? dplanes + l * C * RS + j : nullptr, A.zt + (long long)l * C * N + n, N, live, wh, h);
        wide_barrier();
        if (l == Lh) break;
        wide_layer<OTW, 1>(pre_h, ws + L.w_hid + l * L.w_hid_stride, OT * 256, C / 8, voff, pl + h * RS + j, RS, acc);
      }
      wide_bwd_coord_layer(ws + L.w_out, OTD, C / 8, w, lane, nto, pl + h * RS + j, nn);
    }
    wide_barrier();  // every wave is through its out layer: the plane may take d loss / d (network output)

    // ======================================================================================= upstream gradient of the control
    int hv = h;
    asm volatile("" : "+v"(hv));  // keeps per-element table addresses from being hoisted out of the step loop (sdeh_wide.hip)
    WideScore sq;
    sq.ctrl_kind = ctrl_kind; sq.g = L.g; sq.need_t = need_t; sq.need_p = need_p; sq.tgt = tgt; sq.wl = wl;
    sq.mult = 1.0f; sq.scale_score = A.scale_score; sq.clip_score = A.clip_score; sq.g0 = 0.0f; sq.d = d; sq.gmm = nullptr;
    const float mult = (ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig) * A.scale_score;
    const float gam0 = has_score ? ws[L.gam + t * L.g] : 0.0f;
    f32x16 Gc[2], cvec[2];
    float gsum = 0.0f, cxs = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k < nto) {
        const int ct = w + 4 * k;
        const int cb = 32 * ct + 4 * hv;
        auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
        const f32x16& x = xr[k][0];
        float sc[16], psc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) sc[q] = psc[q] = 0.0f;
        if (has_score || need_p) wide_score_mix16(sq, cx, x, cb, 0, fs, fx0, fiv, sc, psc);
        SDEH_FENCE();
        float xi[16];
        if (ito) {
          wide_noise16(A.noise != nullptr ? A.noise + n * d : nullptr, vec4, cb, d, A.seed, rng_off, grow, t, xi);
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) xi[q] = 0.0f;
        }
        const f32x16 bo = load16(bias_g + Lh * C + (ct * 2 + hv) * 16);
        f32x16 gx, cc_in, dout;
#pragma unroll
        for (int q = 0; q < 16; ++q) gx[q] = cc_in[q] = 0.0f;
        if (!BPTT && A.gextra != nullptr) gx = wide_row16(A.gextra + n * d, cb, d, vec4);
        if (BPTT && A.cost_ctrl != nullptr) cc_in = wide_row16(A.cost_ctrl + n * d, cb, d, vec4);
        float gq16[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const bool valid = coord(q) < d;
          const float nnq = nn[k][0][q] + bo[q];
          float mfac = 0.0f, csc = 0.0f, keep_s = 0.0f;
          if (has_score) {
            mfac = mult * (L.g == 1 ? gam0 : ws[L.gam + t * L.g + coord(q)]);
            csc = clipf(sc[q], A.clip_score);
            keep_s = fabsf(sc[q]) <= A.clip_score ? 1.0f : 0.0f;
          }
          float gc, gq;
          if constexpr (!BPTT) {  // log-variance: d rnd / d u = dB exactly (+ the Bridge cost's u + v for the inference network)
            gc = ito ? wi * c_i * xi[q] : 0.0f;
            if (A.gextra != nullptr) gc = fmaf(wi * cdt, gx[q], gc);
            gq = gc;
          } else {
            const float u = clipf(nnq, A.clip_model) + mfac * csc;
            const float rr = refc ? sig * psc[q] : 0.0f;
            const float uc = A.cost_ctrl != nullptr ? cc_in[q] : u - rr;
            gc = wi * fmaf(uc, cdt, ito ? c_i * xi[q] : 0.0f);
            gq = fmaf(c_u, lam[k][q], gc);
          }
          gc = valid ? gc : 0.0f;
          gq = valid ? gq : 0.0f;
          Gc[k][q] = gc;
          gq16[q] = gq;
          const float gg = has_score ? gq * mult * csc : 0.0f;
          gsum += gg;
          if (has_score && L.g != 1 && valid && live) A.dgam[(long long)coord(q) * N + n] = gg;
          cvec[k][q] = keep_s * mfac * gq;
          dout[q] = fabsf(nnq) <= A.clip_model ? gq : 0.0f;
          if (valid && live) A.dout[(long long)coord(q) * N + n] = dout[q];
          // funnel Jacobian sums (coordinates >= 1): sum_c cvec_c x_c
          const bool first = ct == 0 && q == 0 && hv == 0;
          cxs = first ? cxs : fmaf(cvec[k][q], x[q], cxs);
        }
        // d loss / d (network output) -> plane rows [coordinate][trajectory] (rows of padded coordinates: zeros)
#pragma unroll
        for (int q = 0; q < 16; ++q) pl[(32 * ct + rho(q, hv)) * RS + j] = dout[q];
        SDEH_FENCE();
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) Gc[k][q] = cvec[k][q] = 0.0f;
      }
    }
    {
      const float gs = half_sum(gsum), cs = half_sum(cxs);
      if (h == 0) {
        cx.scr[(WBS_GAM * 4 + w) * RS + j] = gs;
        cx.scr[(WBS_CX * 4 + w) * RS + j] = cs;
        if (w == 0) cx.scr[(WBS_C0 * 4) * RS + j] = cvec[0][0];  // cvec of coordinate 0 (wave 0, tile 0, register 0, lane half 0)
      }
    }
    wide_barrier();  // delta_out and the per-trajectory partial sums are visible
    if (has_score && L.g == 1 && w == 0 && h == 0 && live) A.dgam[n] = wide_slot_sum(cx, WBS_GAM, j);

    // ======================================================================================= back through the layers
    //   delta_Lh = act'(z_Lh) . (W_out^T delta_out);   delta_k = act'(z_k) . (W_hid[k]^T delta_{k+1})
    {
      f32x16 acc[OTW][1];
      {
        WidePre<OTW> pre;
        wide_prefetch<OTW>(pre, ws + L.wt_out, OT * 256, L.dp8 / 8, voff);
        wide_layer<OTW, 1>(pre, ws + L.wt_out, OT * 256, L.dp8 / 8, voff, pl + h * RS + j, RS, acc);
      }
      for (int k = Lh; k >= 0; --k) {
        WidePre<OTW> pre_h;
        if (k > 0) wide_prefetch<OTW>(pre_h, ws + L.wt_hid + (k - 1) * L.w_hid_stride, OT * 256, C / 8, voff);
        wide_barrier();  // everyone has read the plane that is about to be overwritten
        if (has) wide_bwd_delta_store<OTW>(acc, dlds ? dplanes + k * C * RS + j : nullptr, A.zt + (long long)k * C * N + n, act, pl + j,
                                           A.dt + (long long)k * C * N + n, N, live, wh, h);
        wide_barrier();
        if (k == 0) break;
        wide_layer<OTW, 1>(pre_h, ws + L.wt_hid + (k - 1) * L.w_hid_stride, OT * 256, C / 8, voff, pl + h * RS + j, RS, acc);
      }
    }
    if (want_dx) {
      // ===================================================================================== adjoint of the state
      //   lambda_t = c_x lambda_{t+1} + W_in^T delta_0 + (d score term / d x)^T G + direct cost terms   (row-parallel mode with A.dx:
      //   the same quantity without the recursion, written to the plane)
      f32x16 dxa[2][1];
      wide_bwd_coord_layer(ws + L.wt_in, OTD, C / 8, w, lane, nto, pl + h * RS + j, dxa);
      const float coef_t = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET ? wl : 0.0f);
      const float coef_p = ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR ? 1.0f - wl : 0.0f;
      // score terms the reference detaches (reparam.py:58,134,169,188) or obtains by autograd without a graph carry no Jacobian
      const float jac_t = (!has_score || (flags & (SDEH_FLAG_DETACH_SCORE | SDEH_FLAG_TARGET_SCORE_CONST))) ? 0.0f : coef_t;
      const float jac_p = (!has_score || (flags & SDEH_FLAG_DETACH_SCORE)) ? 0.0f : coef_p;
      float fcx = 0.0f, fc0 = 0.0f;
      if (jac_t != 0.0f && tgt.kind == SDEH_DENS_FUNNEL) {
        fcx = wide_slot_sum(cx, WBS_CX, j);
        fc0 = cx.scr[(WBS_C0 * 4) * RS + j];
        if (!(need_t)) { fs = wide_slot_sum(cx, WBS_PRESQ, j); fx0 = cx.scr[(WBS_X0 * 4) * RS + j]; fiv = __expf(-fx0); }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
          const int ct = w + 4 * k;
          const int cb = 32 * ct + 4 * hv;
          auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
          f32x16 le;
#pragma unroll
          for (int q = 0; q < 16; ++q) le[q] = 0.0f;
          if (BPTT && A.lam_extra != nullptr) le = wide_row16(A.lam_extra + n * d, cb, d, vec4);
          float outv[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int cc = coord(q);
            const float x = xr[k][0][q], cv = cvec[k][q];
            float v = BPTT ? fmaf(c_x, lam[k][q], dxa[k][0][q]) : dxa[k][0][q];
            if (jac_t != 0.0f) {  // closed-form target scores are differentiated through x
              float vt = 0.0f;
              if (tgt.kind == SDEH_DENS_DIAG_GAUSS) vt = -cx.tab0[2 * cc + 1] * cv;
              else if (tgt.kind == SDEH_DENS_MULTI_WELL) {
                const float y = x - tgt.p1;
                vt = (cc < tgt.n_comp ? -4.0f * (3.0f * y * y - tgt.p0) : -1.0f) * cv;
              } else if (tgt.kind == SDEH_DENS_FUNNEL) {
                vt = cc == 0 ? fc0 * (-1.0f / tgt.p0 - 0.5f * fiv * fs) + fiv * fcx : fiv * (fc0 * x - cv);
              }
              v = fmaf(jac_t, vt, v);
            }
            if (jac_p != 0.0f || refc) {
              const float pis = cx.tab1[2 * cc + 1];
              v = fmaf(-jac_p * pis, cv, v);                  // Gaussian prior: J = -1/sigma^2
              if (refc) v = fmaf(sig * pis, Gc[k][q], v);     // the cost depends on x through sigma * prior.score(x)
            }
            if (BPTT) {
              v += le[q];
              lam[k][q] = cc < d ? v : 0.0f;
            }
            outv[q] = v;
          }
          if (!BPTT && live) {
            float* __restrict__ xp = A.dx + n * d + cb;
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (coord(q) < d) xp[(q & 3) + 8 * (q >> 2)] = outv[q];
          }
        }
    }
    wide_barrier();  // the plane and the slots are free for the next step's state
  }
}

inline size_t wide_bwd_lds_bytes(const WsLayout& L, bool dlds) {
  const int rows = L.c > 32 * L.otd ? L.c : 32 * L.otd;
  return ((size_t)rows * 32 + (dlds ? (size_t)(L.n_hidden + 1) * L.c * 32 : 0) + kWideSlots * 4 * 32 + 3 * (2 * L.dp + 4)) * sizeof(float);
}

template <int OTW, bool BPTT>
static int launch_wide_bwd_t(const BwdArgs& a, hipStream_t stream) {
  const bool dlds = wide_bwd_lds_bytes(a.lay, true) <= 160 * 1024;  // act' planes in LDS (else recomputed from the zt planes)
  const size_t lds_bytes = wide_bwd_lds_bytes(a.lay, dlds);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_bwd_kernel<OTW, BPTT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  const long long n_tiles = (a.batch + 31) / 32;
  int spc = 1;
  long long grid = n_tiles;
  if (!BPTT) {
    // (step, tile) items are independent: chunks of steps per workgroup so that the launch has a few thousand workgroups at most
    // (the per-workgroup set-up -- tables into LDS -- is amortised) and >= 1024 when the problem allows it
    const long long items = n_tiles * a.n_steps;
    long long want = items < 2048 ? items : 2048;
    long long chunks = (want + n_tiles - 1) / n_tiles;
    if (chunks > a.n_steps) chunks = a.n_steps;
    if (chunks < 1) chunks = 1;
    spc = (int)((a.n_steps + chunks - 1) / chunks);
    chunks = (a.n_steps + spc - 1) / spc;
    grid = n_tiles * chunks;
  }
  hipLaunchKernelGGL((wide_bwd_kernel<OTW, BPTT>), dim3((unsigned)grid), dim3(256), lds_bytes, stream, a, (int)n_tiles, spc, dlds ? 1 : 0);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// SDEH_ERR_UNSUPPORTED: act' planes of (n_hidden + 1) layers beyond 160 KiB of LDS
int launch_wide_bwd(const BwdArgs& a, hipStream_t stream) {
  const bool bptt = !(a.flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  const int otw = a.lay.c == 64 ? 1 : a.lay.c / 128;
  if (otw == 2) return bptt ? launch_wide_bwd_t<2, true>(a, stream) : launch_wide_bwd_t<2, false>(a, stream);
  if (otw == 1) return bptt ? launch_wide_bwd_t<1, true>(a, stream) : launch_wide_bwd_t<1, false>(a, stream);
  return SDEH_ERR_UNSUPPORTED;
}

}  // namespace sdeh
