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

// Column n of rows 32 tile + rho(q, h) of a coordinate-major plane [C][N]: a WAVE-UNIFORM base per (plane, row tile) + a 32-bit byte
// offset per element, (rho(q, h) N + n) 4 < 2^32 (N < 3.3e7 rows), the lane's part made opaque per call -- hipcc otherwise keeps one
// 64-bit address per element and plane as a loop invariant of the step loop (they spill and return as dependent scratch reloads:
// sdeh_bwdf.hip has the measurement).
struct WidePlaneCol {
  char* base;        // plane + 32 tile N floats (uniform)
  unsigned lane;     // (4 h N + n) 4
  unsigned nbytes;   // 4 N (uniform)
  __device__ __forceinline__ float* at(int q) const { return reinterpret_cast<float*>(base + (lane + (unsigned)((q & 3) + 8 * (q >> 2)) * nbytes)); }
};
__device__ __forceinline__ WidePlaneCol wide_plane_col(float* plane_u, int tile, long long N, long long n, int h) {
  WidePlaneCol c;
  c.base = reinterpret_cast<char*>(plane_u + (long long)32 * tile * N);
  c.nbytes = (unsigned)N * 4u;
  c.lane = ((unsigned)(4 * h) * (unsigned)N + (unsigned)n) * 4u;
  asm volatile("" : "+v"(c.lane));
  return c;
}

// this wave's tiles of a layer: z = acc + bias -> HBM plane zt_l [C][N] (column n), act'(z) -> LDS plane dpl [C][32], act(z) -> LDS
// plane out [rows][32]
template <int OTW>
__device__ __forceinline__ void wide_bwd_act_store(f32x16 (&acc)[OTW][1], const f32x16 (&bias)[OTW], int act, float* __restrict__ outl,
                                                   float* __restrict__ dpl, float* __restrict__ zt_l, long long N, long long n, bool store,
                                                   int t0, int h) {
  SDEH_ACT_SWITCH(act, ACT,
    _Pragma("unroll") for (int k = 0; k < OTW; ++k) {
      f32x16 z = acc[k][0] + bias[k];
      const WidePlaneCol col = wide_plane_col(zt_l, t0 + 4 * k, N, n, h);
      if (store) {
        _Pragma("unroll") for (int q = 0; q < 16; ++q) *col.at(q) = z[q];
      }
      if (dpl != nullptr) {
        f32x16 g;
        act_both_tile<ACT>(z, g);
        _Pragma("unroll") for (int q = 0; q < 16; ++q) dpl[(32 * (t0 + 4 * k) + rho(q, h)) * 32] = g[q];
      } else {
        act_tile<ACT>(z);
      }
      _Pragma("unroll") for (int q = 0; q < 16; ++q) outl[(32 * (t0 + 4 * k) + rho(q, h)) * 32] = z[q];
    });
}

// delta_l = acc * act'(z_l) for this wave's tiles -> HBM plane dt_l (column n) and the LDS plane (the next backward layer's input)
// dpl == nullptr (networks whose act' planes do not fit LDS: (n_hidden + 1) C 128 B > ~120 KiB): act'(z_l) is recomputed from the
// pre-activation this lane wrote to zt_l a moment ago (an L2 hit)
template <int OTW>
__device__ __forceinline__ void wide_bwd_delta_store(const f32x16 (&acc)[OTW][1], const float* __restrict__ dpl, float* __restrict__ zt_l,
                                                     int act, float* __restrict__ outl, float* __restrict__ dt_l, long long N, long long n,
                                                     bool store, int t0, int h) {
  if (dpl != nullptr) {
#pragma unroll
    for (int k = 0; k < OTW; ++k) {
      const WidePlaneCol col = wide_plane_col(dt_l, t0 + 4 * k, N, n, h);
      f32x16 v;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ch = 32 * (t0 + 4 * k) + rho(q, h);
        v[q] = acc[k][0][q] * dpl[ch * 32];
        outl[ch * 32] = v[q];
      }
      if (store) {
#pragma unroll
        for (int q = 0; q < 16; ++q) *col.at(q) = v[q];
      }
    }
  } else {
    SDEH_ACT_SWITCH(act, ACT,
      _Pragma("unroll") for (int k = 0; k < OTW; ++k) {
        const WidePlaneCol zc = wide_plane_col(zt_l, t0 + 4 * k, N, n, h);
        const WidePlaneCol col = wide_plane_col(dt_l, t0 + 4 * k, N, n, h);
        f32x16 z;
        _Pragma("unroll") for (int q = 0; q < 16; ++q) z[q] = 0.0f;
        if (store) {
          _Pragma("unroll") for (int q = 0; q < 16; ++q) z[q] = *zc.at(q);
        }
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
          const int ch = 32 * (t0 + 4 * k) + rho(q, h);
          const float v = acc[k][0][q] * act_grad(z[q], ACT);
          if (store) *col.at(q) = v;
          outl[ch * 32] = v;
        }
      });
  }
}

// out[k] = W[coordinate tiles {w, w + 4}][:] . plane   (an out-layer-shaped product: w_out or input_embed^T), nto tiles of this wave
__device__ __forceinline__ void wide_bwd_coord_layer(const float* __restrict__ wl, int otd, int NS4, int w, int lane, int nto,
                                                     const float* __restrict__ actl, f32x16 (&out)[2][1], int rs = 32) {
  unsigned vo[2];
  vo[0] = (unsigned)((w * 64 + lane) * 16);
  vo[1] = nto > 1 ? (unsigned)(((w + 4) * 64 + lane) * 16) : vo[0];
  if (nto == 2) {
    WidePre<2> pre;
    wide_prefetch<2>(pre, wl, otd * 256, NS4, vo);
    wide_layer<2, 1>(pre, wl, otd * 256, NS4, vo, actl, rs, out);
  } else {  // one tile, or none (the wave runs the same stream on tile w % otd and drops the result: no control flow around asm loads)
    WidePre<1> pre;
    unsigned v1[1] = {nto == 1 ? vo[0] : (unsigned)(((w % otd) * 64 + lane) * 16)};
    wide_prefetch<1>(pre, wl, otd * 256, NS4, v1);
    f32x16 o1[1][1];
    wide_layer<1, 1>(pre, wl, otd * 256, NS4, v1, actl, rs, o1);
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
          if (flags & SDEH_FLAG_TERMINAL_TARGET) {
            if (A.tscore_in != nullptr) v = fmaf(-wi, (cc < d ? A.tscore_in[lrow * d + cc] : 0.0f), v);  // (the clip indicator is in the plane)
            else v = fmaf(-wi * keep, wide_target_score(tgt, cx.tab0, cc, d, xr[k][0][q], fs, fx0, fiv), v);
          }
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
    if (A.xt_out != nullptr && live) {  // x_t coordinate-major [d][N]: the input layer's weight-gradient operand (a strided transpose
#pragma unroll                          // of the row-major trajectory costs the framework 5.8 ms per 1.3 GB)
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
          const WidePlaneCol col = wide_plane_col(A.xt_out, w + 4 * k, N, n, h);
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (32 * (w + 4 * k) + rho(q, h) < d) *col.at(q) = xr[k][0][q];
        }
    }
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
        if (has) wide_bwd_act_store<OTW>(acc, bias, act, pl + j, dlds ? dplanes + l * C * RS + j : nullptr, A.zt + (long long)l * C * N, N, n, live, wh, h);
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
        if (A.sc_in != nullptr) {  // mixture target: the forward launch's score plane stands in (no mixture is evaluated here)
          const f32x16 sv = wide_row16(A.sc_in + n * d, cb, d, vec4);
#pragma unroll
          for (int q = 0; q < 16; ++q) sc[q] = sv[q];
          if (need_p) {  // the prior score alone (the reference control of EulerDDS)
            WideScore sp = sq;
            sp.ctrl_kind = SDEH_CTRL_CLIPPED;
            float dummy[16];
            wide_score_mix16(sp, cx, x, cb, 0, fs, fx0, fiv, dummy, psc);
          }
        } else if (has_score || need_p) {
          wide_score_mix16(sq, cx, x, cb, 0, fs, fx0, fiv, sc, psc);
        }
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
        if (has) wide_bwd_delta_store<OTW>(acc, dlds ? dplanes + k * C * RS + j : nullptr, A.zt + (long long)k * C * N, act, pl + j,
                                           A.dt + (long long)k * C * N, N, n, live, wh, h);
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


// =========================================================================================================
// Gradient of the Bridge's divergence term on wide networks (losses/oc.py:189-200 with utils/autograd.py:14-22, create_graph=True):
//     L_div = sum over rows n = (t, i) of  c_n sum_j m_nj J_jj(x_n; theta_v),   c_n = w_i sigma(t) dt,  m_nj = 1[|v_nn,j| <= clip_model]
// for an inference network with Lh = 1 or 2 hidden layers.  With D_l = act'(z_l), P_j = D_0 . W_in[:, j], R_j = D_Lh . W_out[j, :]:
//     Lh = 2:  F_j = W_1 P_j,  G_j = W_2^T R_j,  J_jj = G_j^T D_1 F_j          Lh = 1:  F_j = W_1 P_j,  G_j = R_j := W_out[j, :]
// The per-(row, coordinate) operands (F, G, their adjoints: C floats each, 657 GB per plane at configs[4]'s size) can never be
// written out, so the contraction is fused: the [C, C] gradient of ONE hidden layer stays in the accumulators of a persistent
// workgroup for the whole launch (all 256 AGPRs of its four waves at C = 256 -- which is why the two hidden layers take two
// launches, `side` 0 / 1).  One launch, per (step, tile of 32 trajectories) and coordinate j, all four waves on the same j:
//     S_j  = X_s Q_s                 (side 0: G_j = W_2^T R_j;  side 1: F_j = W_1 P_j)            one [C, C] product, wave w: row tiles {w, w + 4}
//     O_j  = X Q_j                   (side 0 only: F_j, for d L / d D_1 = sum_j c_j F_j G_j)
//     dS_j = c_j D_1 . S_j           -> LDS plane [C][32]
//     dX  += dS_j Q_j^T              (contraction over the 32 trajectories: both operands are planes read transposed; side 0: dW_1,
//                                     side 1: dW_2^T)
//     dQ_j = X^T dS_j                -> d L / d D_q += dQ_j . col_q[j],   d L / d col_q[j] += sum_traj dQ_j . D_q
// (col_q[j] = column j of W_in for side 0, row j of W_out for side 1).  Seven [C, C] products per (row, coordinate) in total against
// the forward's two.  After the coordinates of an item the adjoints of the base pre-activations (how z_l enters through act') close
// the chain: adj z_l = act''(z_l) dD_l + act'(z_l) W_{l+1}^T adj z_{l+1}, written as planes d2 [(Lh+1), C, N] -- the weight gradients
// they imply are the same contraction as the first-order ones (sdeh_weight_grad adds them to the planes of sdeh_ctrl_backward_ex).
// =========================================================================================================
// -DSDEH_WDIV_PROFILE: per-phase cycle counts of one wave (block 0, wave 0) of the divergence backward -- a measurement build, never
// shipped (tools/wdiv_phase_profile.sh).  0 S pass, 1 O pass, 2 elementwise + dS store, 3 barrier A, 4 dX, 5 dQ pass, 6 dQ elementwise,
// 7 barrier B + tables, 8 coordinates counted, 9 item prologue, 10 item epilogue, 11 items
#ifdef SDEH_WDIV_PROFILE
__device__ unsigned long long wdiv_prof[16];
#define WDIV_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define WDIV_ADD(k, t0, t1) do { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&wdiv_prof[k], (t1) - (t0)); } while (0)
#else
#define WDIV_T(var) do {} while (0)
#define WDIV_ADD(k, t0, t1) do {} while (0)
#endif

constexpr int kDivRS = 36;  // row stride of the planes: conflict-free transposed ds_read_b128 (4 x odd), as in sdeh_bwdf.hip


// B-order index of channel ch inside the tangent tables (WsLayout: idx = (ch / 8) * 8 + (ch & 1) * 4 + ((ch & 7) >> 1))
__device__ __forceinline__ int wdiv_bidx(int ch) { return (ch & ~7) + (ch & 1) * 4 + ((ch & 7) >> 1); }

// acc[k] = sum over NS4 k-groups of  Wp[tile tiles[k]][:] . (plane[:, traj] * col[:])   (col == nullptr: the plane alone)
//   wgrp: packed layer (k-groups of OT tiles);  pl: plane + h * kDivRS + j;  col: B-order vector + 4 h;  NS4 a multiple of 4
// The A operands stream from L2 through a ring of four k-groups, three groups (>= 1536 cycles of this wave's MFMAs) ahead of their
// use: with one wave per SIMD nothing else hides the L2 latency (a ring of two measured 3x the MFMA time of the whole kernel).
constexpr int kDivPD = 3;
template <int NT>
__device__ __forceinline__ void wdiv_pass(const float* __restrict__ wgrp, int OT, const int (&tiles)[NT], int NS4, unsigned lane_off,
                                          const float* __restrict__ pl, const float* __restrict__ col, f32x16 (&acc)[NT]) {
  constexpr int U = kDivPD + 1;
  f32x4 a[U][NT];
  const int grp_floats = OT * 256;
  unsigned voff[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) voff[k] = lane_off + (unsigned)(tiles[k] * 1024);
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;
  const int last = NS4 - 1;
#pragma unroll
  for (int g = 0; g < kDivPD; ++g)
#pragma unroll
    for (int k = 0; k < NT; ++k) wide_gload(a[g][k], voff[k], wgrp + (long long)(g < last ? g : last) * grp_floats);
  float b[2][4];
  auto loadB = [&](int S, float (&bv)[4]) {
    const int Sc = S < last ? S : last;
    const float* __restrict__ dp = pl + (8 * Sc) * kDivRS;
    bv[0] = dp[0]; bv[1] = dp[2 * kDivRS]; bv[2] = dp[4 * kDivRS]; bv[3] = dp[6 * kDivRS];
    if (col != nullptr) {
      const float4 cv = *reinterpret_cast<const float4*>(col + 8 * Sc);
      bv[0] *= cv.x; bv[1] *= cv.y; bv[2] *= cv.z; bv[3] *= cv.w;
    }
  };
  loadB(0, b[0]);
  const float* __restrict__ wnext = wgrp + (long long)(kDivPD < last ? kDivPD : last) * grp_floats;  // group S + kDivPD of the stream
  for (int S = 0; S < NS4; S += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      wide_vmwait<(kDivPD - 1) * NT, NT>(a[u]);  // everything older than the kDivPD - 1 newest groups has landed
      SDEH_FENCE();
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[k] = SDEH_MFMA(a[u][k][e], b[u % 2][e], acc[k]);
      SDEH_FENCE();
#pragma unroll
      for (int k = 0; k < NT; ++k) wide_gload(a[(u + kDivPD) % U][k], voff[k], wnext);
      wnext = S + u + kDivPD < last ? wnext + grp_floats : wnext;
      loadB(S + u + 1, b[(u + 1) % 2]);
      SDEH_FENCE();
#pragma unroll
      for (int e = 2; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[k] = SDEH_MFMA(a[u][k][e], b[u % 2][e], acc[k]);
      SDEH_FENCE();
    }
  }
  // the clamped re-reads requested by the last iterations have no consumer: they must land before their registers are reused
#pragma unroll
  for (int u = 0; u < U; ++u) wide_vmwait<0, NT>(a[u]);
}

// LH2: two hidden layers (else one: G_j = W_out[j, :] itself -- no S product, no second side);  SIDE: which hidden layer's gradient
// this launch accumulates (compile-time, so that the per-element loops carry no wave-uniform branches)
template <int OTW, bool LH2, int SIDE>  // C = 128 OTW
__global__ __launch_bounds__(256) void wide_bridge_div_bwd_kernel(const WideDivArgs A, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int OT = 4 * OTW, C = 128 * OTW, RS = kDivRS;
  const WsLayout& L = A.lay;
  const WsLayout& L2 = A.lay2;
  const float* __restrict__ ws = A.ws;
  const float* __restrict__ ws2 = A.ws2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  constexpr int Lh = LH2 ? 2 : 1, side = SIDE;
  const int d = A.d, OTD = L2.otd, T = A.n_steps, act = A.act;
  const long long B = A.batch, N = (long long)T * B;
  constexpr bool direct = !LH2;
  static_assert(LH2 || SIDE == 0, "one hidden layer: a single launch");

  float* __restrict__ Dp = lds;                               // act'(z_l), l = 0 .. Lh: [Lh + 1][C][RS]
  float* __restrict__ dS = Dp + (Lh + 1) * C * RS;             // dS_j / scratch plane [C][RS] (rows of coordinates during the mask pass: 32 OTD <= C)
  float* __restrict__ cols = dS + C * RS;                      // [2 buffers][2][C]: col_s, col_q of the coordinate at hand (B order)
  unsigned* __restrict__ maskw = reinterpret_cast<unsigned*>(cols + 4 * C);  // [8 coordinate tiles][32 trajectories] clip-mask bits
  float* __restrict__ slots = reinterpret_cast<float*>(maskw + 8 * 32);       // [4][32]
  float* __restrict__ stg = slots + 4 * 32;                    // [2][C]: the coordinate's d L / d col_q (and d L / d W_out row, Lh = 1) before
                                                               // they are added to the workgroup's partial tables, coalesced
  float* __restrict__ ptab = stg + 2 * C;                      // prior table (mu, 1/sigma^2) [2 dp]

  for (int i = tid; i < 2 * L.dp; i += 256) ptab[i] = ws[L.dg[1] + i];
  int tiles[OTW];
#pragma unroll
  for (int k = 0; k < OTW; ++k) tiles[k] = w + 4 * k;
  const unsigned lane_off = (unsigned)(lane * 16);
  const int nto = (OTD > w ? 1 : 0) + (OTD > w + 4 ? 1 : 0);
  const bool vec4 = (d & 3) == 0;
  // images: S = X_s Q_s, O = X Q, dQ = X^T dS
  const float* __restrict__ img_s = side == 0 ? ws2 + L2.wt_hid + L2.w_hid_stride : ws2 + L2.w_hid;   // W_2^T | W_1
  const float* __restrict__ img_o = ws2 + L2.w_hid;                                                      // W_1 (side 0: O = F)
  const float* __restrict__ img_b = side == 0 ? ws2 + L2.wt_hid : ws2 + L2.w_hid + L2.w_hid_stride;     // W_1^T | W_2
  const float* __restrict__ tab_s = side == 0 ? ws2 + L2.tan_out : ws2 + L2.tan_in;                      // col_s[j]: row j of W_out | column j of W_in
  const float* __restrict__ tab_q = side == 0 ? ws2 + L2.tan_in : ws2 + L2.tan_out;
  float* __restrict__ Ds = Dp + (side == 0 ? Lh : 0) * C * RS;   // D of the S product's B operand (D_Lh | D_0)
  float* __restrict__ Dq = Dp + (side == 0 ? 0 : Lh) * C * RS;   // D_q (D_0 | D_Lh)
  float* __restrict__ D1 = Dp + (Lh == 2 ? 1 : (side == 0 ? 1 : 0)) * C * RS;  // the middle derivative (Lh = 1: D_1 is also D_Lh)

  f32x16 acc[OTW][OT];  // dX: row tiles {w, w + 4} x all column tiles
#pragma unroll
  for (int k = 0; k < OTW; ++k)
#pragma unroll
    for (int bt = 0; bt < OT; ++bt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][bt][q] = 0.0f;

  float* __restrict__ cpart = A.cpart + (long long)blockIdx.x * d * C;
  float* __restrict__ spart = A.spart != nullptr ? A.spart + (long long)blockIdx.x * d * C : nullptr;
  auto bidx = [&](int ot, int q) { return (4 * ot + (q >> 2)) * 8 + (q & 1) * 4 + ((q >> 1) & 1) + 2 * h; };  // B order of channel 32 ot + rho(q, h)

  // this workgroup's partial tables start at zero (written here, not by a memset node: inside a replayed hipGraph a captured
  // hipMemsetAsync in front of these launches was not reliably re-executed on this stack -- the tables kept accumulating from the
  // second replay on, tests/test_hip_graphs.py)
  for (int i = tid; i < d * C; i += 256) {
    cpart[i] = 0.0f;
    if (spart != nullptr) spart[i] = 0.0f;
  }
  const long long n_items = (long long)n_tiles * T;
  __syncthreads();
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int t = (int)(item / n_tiles), tile = (int)(item % n_tiles);
    const long long row0 = (long long)tile * 32, r = row0 + j;
    const bool live = r < B;
    const long long lrow = live ? r : B - 1;
    const long long n = (long long)t * B + lrow;
    cfp cf = as_const(ws + L.coef + t * kCoefStride);
    const float sig = cf[CF_SIGMA], dt = cf[CF_DT], wl = cf[CF_W];
    const float crow = live ? A.grad_rnd[lrow] * sig * dt : 0.0f;  // c_n
    WDIV_T(ti0);

    // ---- act'(z_l) planes; act(z_Lh) -> scratch plane (the out layer's input) -------------------------------------------------
    SDEH_ACT_SWITCH(act, ACT,
      for (int l = 0; l <= Lh; ++l) {
        _Pragma("unroll") for (int k = 0; k < OTW; ++k) {
          f32x16 z;
          _Pragma("unroll") for (int q = 0; q < 16; ++q) z[q] = live ? A.zt[((long long)l * C + 32 * tiles[k] + rho(q, h)) * N + n] : 0.0f;
          _Pragma("unroll") for (int q = 0; q < 16; ++q) Dp[(l * C + 32 * tiles[k] + rho(q, h)) * RS + j] = act_grad(z[q], ACT);
          if (l == Lh) {
            act_tile<ACT>(z);
            _Pragma("unroll") for (int q = 0; q < 16; ++q) dS[(32 * tiles[k] + rho(q, h)) * RS + j] = z[q];
          }
        }
      });
    if (tid < 8 * 32) maskw[tid] = 0u;
    wide_barrier();
    // ---- v_nn = W_out a_Lh + b_out -> clip-mask bits m_nj (torch.clamp's backward: 1 on [-m, m]) ------------------------------------
    {
      f32x16 vnn[2][1];
      unsigned vo[2];
      vo[0] = (unsigned)((w * 64 + lane) * 16);
      vo[1] = nto > 1 ? (unsigned)(((w + 4) * 64 + lane) * 16) : vo[0];
      const float* __restrict__ wl_out = ws2 + L2.w_out;
      if (nto == 2) {
        WidePre<2> pre;
        wide_prefetch<2>(pre, wl_out, OTD * 256, C / 8, vo);
        wide_layer<2, 1>(pre, wl_out, OTD * 256, C / 8, vo, dS + h * RS + j, RS, vnn);
      } else {
        WidePre<1> pre;
        unsigned v1[1] = {nto == 1 ? vo[0] : (unsigned)(((w % OTD) * 64 + lane) * 16)};
        wide_prefetch<1>(pre, wl_out, OTD * 256, C / 8, v1);
        f32x16 o1[1][1];
        wide_layer<1, 1>(pre, wl_out, OTD * 256, C / 8, v1, dS + h * RS + j, RS, o1);
        vnn[0][0] = o1[0][0];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
          const f32x16 bo = load16(ws2 + L2.b_hid + Lh * C + ((w + 4 * k) * 2 + h) * 16);
          unsigned bits = 0u;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float v = vnn[k][0][q] + bo[q];
            if (v >= -A.clip_model && v <= A.clip_model) bits |= 1u << (4 * h + (q & 3) + 8 * (q >> 2));
          }
          atomicOr(&maskw[(w + 4 * k) * 32 + j], bits);
        }
    }
    // ---- score part of the divergence: only gamma(t) carries parameters (side 0) ---------------------------------------------
    if (side == 0 && A.inf_kind == SDEH_CTRL_LERP_PRIOR) {
      const float w1 = 1.0f - wl;
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < nto) {
          const int cb = 32 * (w + 4 * k) + 4 * h;
          const f32x16 x = wide_row16(A.xs + n * d, cb, d, vec4);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int cc = cb + (q & 3) + 8 * (q >> 2);
            const float mu = ptab[2 * cc], iv = ptab[2 * cc + 1];
            const float sc = w1 * (mu - x[q]) * iv;
            const bool inside = sc >= -A.clip_score && sc <= A.clip_score;
            const float gj = (inside && cc < d) ? crow * sig * A.scale_score * -(w1 * iv) : 0.0f;
            s += gj;
            if (L2.g != 1 && cc < d && live) A.dgam[(long long)cc * N + n] = gj;
          }
        }
      s = half_sum(s);
      if (h == 0) slots[w * 32 + j] = s;
    }
    // the first coordinate's columns
    if (tid < C / 4) {
      *reinterpret_cast<float4*>(cols + tid * 4) = *reinterpret_cast<const float4*>(tab_s + tid * 4);
      *reinterpret_cast<float4*>(cols + C + tid * 4) = *reinterpret_cast<const float4*>(tab_q + tid * 4);
    }
    wide_barrier();
    if (side == 0 && A.inf_kind == SDEH_CTRL_LERP_PRIOR && L2.g == 1 && w == 0 && h == 0 && live)
      A.dgam[n] = ((slots[j] + slots[32 + j]) + slots[64 + j]) + slots[96 + j];

    // ---- the coordinates --------------------------------------------------------------------------------------------------
    f32x16 dDq[OTW], dD1[OTW];
#pragma unroll
    for (int k = 0; k < OTW; ++k)
#pragma unroll
      for (int q = 0; q < 16; ++q) dDq[k][q] = dD1[k][q] = 0.0f;
    WDIV_T(ti1);
    WDIV_ADD(9, ti0, ti1);
    for (int jc = 0; jc < d; ++jc) {
      WDIV_T(tc0);
      const float* __restrict__ cs = cols + (jc & 1) * 2 * C;  // col_s (B order)
      const float* __restrict__ cq = cs + C;                    // col_q
      if (jc + 1 < d && tid < C / 4) {  // stage the next coordinate's columns into the other buffer (visible behind barrier A)
        float* __restrict__ nx = cols + ((jc + 1) & 1) * 2 * C;
        *reinterpret_cast<float4*>(nx + tid * 4) = *reinterpret_cast<const float4*>(tab_s + (long long)(jc + 1) * C + tid * 4);
        *reinterpret_cast<float4*>(nx + C + tid * 4) = *reinterpret_cast<const float4*>(tab_q + (long long)(jc + 1) * C + tid * 4);
      }
      // this coordinate's rows of the partial tables: requested now, added to behind barrier B (one coalesced read-modify-write per
      // thread instead of 32 dependent scattered ones per wave)
      float old_c = 0.0f, old_s = 0.0f;
      if (tid < C) {
        old_c = cpart[(long long)jc * C + tid];
        if constexpr (direct) old_s = spart[(long long)jc * C + tid];
      }
      const float cj = ((maskw[(jc >> 5) * 32 + j] >> (jc & 31)) & 1u) ? crow : 0.0f;
      f32x16 S[OTW], O[OTW];
      if constexpr (!direct) wdiv_pass<OTW>(img_s, OT, tiles, C / 8, lane_off, Ds + h * RS + j, cs + 4 * h, S);
      SDEH_FENCE();
      WDIV_T(tc1);
      if constexpr (side == 0) wdiv_pass<OTW>(img_o, OT, tiles, C / 8, lane_off, Dq + h * RS + j, cq + 4 * h, O);
      SDEH_FENCE();
      WDIV_T(tc2);
      // Element (k, q) of this lane = channel 32 (w + 4 k) + rrow(q) + 4 h: every LDS address below is ONE per-lane base plus a
      // compile-time offset (the DS instructions' immediate).  The bases are made opaque per coordinate: hipcc otherwise precomputes
      // one address register per element and array -- ~600 loop invariants that live in scratch and come back one dependent
      // scratch_load at a time (measured: 40 k cycles per coordinate for this block, 2.4 x its matrix work).
      int e_rs = (32 * w + 4 * h) * RS + j, e_b = 32 * w + 2 * h, e_ch = 32 * w + 4 * h;
      asm volatile("" : "+v"(e_rs), "+v"(e_b), "+v"(e_ch));
      {
        const float* __restrict__ d1e = D1 + e_rs;
        float* __restrict__ dse = dS + e_rs;
        const float* __restrict__ cse = cs + e_b;
        float* __restrict__ stge = stg + C + e_ch;
#pragma unroll
        for (int k = 0; k < OTW; ++k)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int orow = (128 * k + (q & 3) + 8 * (q >> 2)) * RS;
            const int ob = 128 * k + (q >> 2) * 8 + (q & 1) * 4 + ((q >> 1) & 1);
            const float d1 = d1e[orow];
            float sv;
            if constexpr (direct) sv = cse[ob];
            else sv = S[k][q];
            if constexpr (side == 0) dD1[k][q] = fmaf(cj * O[k][q], sv, dD1[k][q]);
            dse[orow] = cj * d1 * sv;
            if constexpr (direct) {  // one hidden layer: d L / d W_out[j, ch] = sum_traj c_j D_1 F_j
              const float v = sum_xor16(sum_row16(cj * d1 * O[k][q]));
              if (j == 0) stge[128 * k + (q & 3) + 8 * (q >> 2)] = v;
            }
          }
      }
      WDIV_T(tc3);
      wide_barrier();  // A: dS_j complete (and the next coordinate's columns staged)
      WDIV_T(tc4);
      // dX += dS_j Q_j^T over the 32 trajectories: lane (m, kk) reads rows transposed, trajectories 8 u + 4 kk .. + 3
      {
        float4 av[OTW][4];
#pragma unroll
        for (int k = 0; k < OTW; ++k)
#pragma unroll
          for (int u = 0; u < 4; ++u) av[k][u] = *reinterpret_cast<const float4*>(dS + (32 * tiles[k] + j) * RS + 8 * u + 4 * h);
#pragma unroll
        for (int bt = 0; bt < OT; ++bt) {
          SDEH_FENCE();  // one column tile's operands at a time (hoisted, the eight tiles' reads are 128 live registers)
          const float cqv = cq[wdiv_bidx(32 * bt + j)];
          float4 bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 t4 = *reinterpret_cast<const float4*>(Dq + (32 * bt + j) * RS + 8 * u + 4 * h);
            bv[u] = float4{t4.x * cqv, t4.y * cqv, t4.z * cqv, t4.w * cqv};
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < OTW; ++k) {
              acc[k][bt] = SDEH_MFMA(av[k][u].x, bv[u].x, acc[k][bt]);
              acc[k][bt] = SDEH_MFMA(av[k][u].y, bv[u].y, acc[k][bt]);
              acc[k][bt] = SDEH_MFMA(av[k][u].z, bv[u].z, acc[k][bt]);
              acc[k][bt] = SDEH_MFMA(av[k][u].w, bv[u].w, acc[k][bt]);
            }
        }
      }
      // dQ_j = X^T dS_j;  d L / d D_q,  d L / d col_q[j]
      WDIV_T(tc5);
      {
        f32x16 dQ[OTW];
        SDEH_FENCE();
        wdiv_pass<OTW>(img_b, OT, tiles, C / 8, lane_off, dS + h * RS + j, nullptr, dQ);
        SDEH_FENCE();
        WDIV_T(tc6);
        WDIV_ADD(5, tc5, tc6);
        int f_rs = (32 * w + 4 * h) * RS + j, f_b = 32 * w + 2 * h, f_ch = 32 * w + 4 * h;
        asm volatile("" : "+v"(f_rs), "+v"(f_b), "+v"(f_ch));
        const float* __restrict__ dqe = Dq + f_rs;
        const float* __restrict__ cqe = cq + f_b;
        float* __restrict__ stgq = stg + f_ch;
#pragma unroll
        for (int k = 0; k < OTW; ++k)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if ((q & 3) == 0) SDEH_FENCE();
            const int orow = (128 * k + (q & 3) + 8 * (q >> 2)) * RS;
            const int ob = 128 * k + (q >> 2) * 8 + (q & 1) * 4 + ((q >> 1) & 1);
            dDq[k][q] = fmaf(dQ[k][q], cqe[ob], dDq[k][q]);
            const float v = sum_xor16(sum_row16(dQ[k][q] * dqe[orow]));
            if (j == 0) stgq[128 * k + (q & 3) + 8 * (q >> 2)] = v;
          }
      }
      WDIV_T(tc7);
      wide_barrier();  // B: everyone is through dS_j; the staged sums are complete
      if (tid < C) {
        cpart[(long long)jc * C + tid] = old_c + stg[tid];
        if constexpr (direct) spart[(long long)jc * C + tid] = old_s + stg[C + tid];
      }
      WDIV_T(tc8);
      WDIV_ADD(0, tc0, tc1); WDIV_ADD(1, tc1, tc2); WDIV_ADD(2, tc2, tc3); WDIV_ADD(3, tc3, tc4); WDIV_ADD(4, tc4, tc5);
      WDIV_ADD(6, tc5, tc7); WDIV_ADD(7, tc7, tc8); WDIV_ADD(8, 0ull, 1ull);
    }

    // ---- adjoints of the base pre-activations ------------------------------------------------------------------------------------
    //   side 1: adj z_Lh = act''(z_Lh) dD_Lh.   side 0: adj z_1 = act''(z_1) dD_1 + act'(z_1) W_2^T adj z_2 (Lh = 2), then
    //   adj z_0 = act''(z_0) dD_0 + act'(z_0) W_1^T adj z_1
    WDIV_T(te0);
    auto load_z = [&](int l, int k) {
      f32x16 z;
#pragma unroll
      for (int q = 0; q < 16; ++q) z[q] = live ? A.zt[((long long)l * C + 32 * tiles[k] + rho(q, h)) * N + n] : 0.0f;
      return z;
    };
    float* __restrict__ dS2 = Dp + Lh * C * RS;  // act'(z_Lh) is no longer needed: adj z_0 goes here (the dS plane is still being read)
    if (side == 1) {
#pragma unroll
      for (int k = 0; k < OTW; ++k) {
        const f32x16 z = load_z(Lh, k);
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (live) A.d2[((long long)Lh * C + 32 * tiles[k] + rho(q, h)) * N + n] = act_grad2(z[q], act) * dDq[k][q];
      }
    } else {
      f32x16 adj[OTW];
      if (Lh == 2) {  // adj z_2 from the side-1 launch -> plane; adj a_1 = W_2^T adj z_2
#pragma unroll
        for (int k = 0; k < OTW; ++k)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int ch = 32 * tiles[k] + rho(q, h);
            dS[ch * RS + j] = live ? A.d2[((long long)2 * C + ch) * N + n] : 0.0f;
          }
        wide_barrier();
        wdiv_pass<OTW>(ws2 + L2.wt_hid + L2.w_hid_stride, OT, tiles, C / 8, lane_off, dS + h * RS + j, nullptr, adj);
        wide_barrier();
      }
      // adj z_1
#pragma unroll
      for (int k = 0; k < OTW; ++k) {
        const f32x16 z = load_z(1, k);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ch = 32 * tiles[k] + rho(q, h);
          float v = act_grad2(z[q], act) * dD1[k][q];
          if (Lh == 2) v = fmaf(Dp[(C + ch) * RS + j], adj[k][q], v);
          if (live) A.d2[((long long)C + ch) * N + n] = v;
          dS[ch * RS + j] = v;
        }
      }
      wide_barrier();
      wdiv_pass<OTW>(ws2 + L2.wt_hid, OT, tiles, C / 8, lane_off, dS + h * RS + j, nullptr, adj);  // adj a_0 = W_1^T adj z_1
#pragma unroll
      for (int k = 0; k < OTW; ++k) {
        const f32x16 z = load_z(0, k);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ch = 32 * tiles[k] + rho(q, h);
          const float v = fmaf(Dp[ch * RS + j], adj[k][q], act_grad2(z[q], act) * dDq[k][q]);
          if (live) A.d2[(long long)ch * N + n] = v;
          if (A.dx != nullptr) dS2[ch * RS + j] = v;
        }
      }
      if (A.dx != nullptr) {  // back-propagation through time (method kl): the divergence term's share of d loss / d x_t = W_in^T adj z_0
        wide_barrier();       // (the mask and the score part carry no d/dx: clamp / clip indicators are constants of the graph)
        f32x16 dxa[2][1];
        wide_bwd_coord_layer(ws2 + L2.wt_in, OTD, C / 8, w, lane, nto, dS2 + h * RS + j, dxa, RS);
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < nto && live) {
            float* __restrict__ xp = A.dx + n * d + 32 * (w + 4 * k) + 4 * h;
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (32 * (w + 4 * k) + 4 * h + (q & 3) + 8 * (q >> 2) < d) xp[(q & 3) + 8 * (q >> 2)] += dxa[k][0][q];
          }
      }
    }
    wide_barrier();  // planes free for the next item
    WDIV_T(te1);
    WDIV_ADD(10, te0, te1); WDIV_ADD(11, 0ull, 1ull);
  }

  // ---- this workgroup's partial of dX ---------------------------------------------------------------------------------------------
  float* __restrict__ xp = A.xpart + (long long)blockIdx.x * C * C;
#pragma unroll
  for (int k = 0; k < OTW; ++k)
#pragma unroll
    for (int bt = 0; bt < OT; ++bt)
#pragma unroll
      for (int q = 0; q < 16; ++q) xp[(32 * tiles[k] + rho(q, h)) * C + 32 * bt + j] = acc[k][bt][q];
}

inline size_t wide_div_lds_bytes(const WsLayout& L, const WsLayout& L2) {
  return ((size_t)(L2.n_hidden + 2) * L2.c * kDivRS + 6 * L2.c + 8 * 32 + 4 * 32 + 2 * L.dp) * sizeof(float);
}

int wide_div_grid(long long batch, int n_steps) {
  const long long items = ((batch + 31) / 32) * n_steps;
  return (int)(items < 256 ? items : 256);  // persistent: one workgroup per CU (the [C, C] accumulators live for the whole launch)
}

// side 1 first (it leaves adj z_Lh in d2), then side 0.  Lh = 1: side 0 only.
#ifdef SDEH_WDIV_PROFILE
static void wdiv_prof_dump(hipStream_t stream, int side) {
  (void)hipStreamSynchronize(stream);
  unsigned long long v[16];
  (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(wdiv_prof), sizeof(v));
  const double n = v[8] ? (double)v[8] : 1.0, ni = v[11] ? (double)v[11] : 1.0;
  fprintf(stderr, "wdiv side %d (cycles per coordinate of block 0 / wave 0, %llu coordinates): S %.0f | O %.0f | elementwise %.0f | barrier A %.0f | "
          "dX %.0f | dQ pass %.0f | dQ pass + elementwise %.0f | barrier B + tables %.0f || per item (%llu): prologue %.0f epilogue %.0f\n",
          side, v[8], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5] / n, v[6] / n, v[7] / n, v[11], v[9] / ni, v[10] / ni);
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(wdiv_prof), z, sizeof(z));
}
#endif

int launch_wide_div_bwd(const WideDivArgs& a, hipStream_t stream) {
#ifdef SDEH_WDIV_PROFILE
  struct Dump { hipStream_t s; int side; ~Dump() { wdiv_prof_dump(s, side); } } dump{stream, a.side};
#endif
  const size_t lds_bytes = wide_div_lds_bytes(a.lay, a.lay2);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  const int otw = a.lay2.c / 128, lh = a.lay2.n_hidden;
  if ((otw != 1 && otw != 2) || (lh != 1 && lh != 2) || (lh == 1 && a.side != 0)) return SDEH_ERR_UNSUPPORTED;
  const int n_tiles = (int)((a.batch + 31) / 32);
  const int grid = wide_div_grid(a.batch, a.n_steps);
  const int variant = (otw - 1) * 3 + (lh == 1 ? 2 : a.side);
  static bool attr_done[kMaxDevices][6] = {};
  bool& attr_set = attr_done[current_device_slot()][variant];
#define SDEH_WDIV_LAUNCH(OTW_, LH2_, SIDE_)                                                                                        \
  do {                                                                                                                             \
    if (!attr_set) {                                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_bridge_div_bwd_kernel<OTW_, LH2_, SIDE_>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                               \
        return SDEH_ERR_HIP;                                                                                                       \
      attr_set = true;                                                                                                             \
    }                                                                                                                              \
    hipLaunchKernelGGL((wide_bridge_div_bwd_kernel<OTW_, LH2_, SIDE_>), dim3(grid), dim3(256), lds_bytes, stream, a, n_tiles);     \
  } while (0)
  switch (variant) {
    case 0: SDEH_WDIV_LAUNCH(1, true, 0); break;
    case 1: SDEH_WDIV_LAUNCH(1, true, 1); break;
    case 2: SDEH_WDIV_LAUNCH(1, false, 0); break;
    case 3: SDEH_WDIV_LAUNCH(2, true, 0); break;
    case 4: SDEH_WDIV_LAUNCH(2, true, 1); break;
    default: SDEH_WDIV_LAUNCH(2, false, 0); break;
  }
#undef SDEH_WDIV_LAUNCH
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
