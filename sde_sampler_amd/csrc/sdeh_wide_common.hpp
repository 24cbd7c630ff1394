// Building blocks of the wide-network kernels (C = 64 with d > 64, 128, 256 channels; d <= 256): the channel-split layer product with
// hand-issued operand streams, activation / publish helpers, per-coordinate score terms and noise in the accumulator layout.
// Shared by the trajectory kernels (sdeh_wide.hip) and the training backward (sdeh_wide_bwd.hip); see sdeh_wide.hip for the design.
#pragma once
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
// What a layer needs before its first MFMA, fetched EARLY (before the previous layer's activation, LDS stores and barrier) so that
// the L2 latency of a layer's first operands is not paid four times per step: its first kWidePD groups of A operands.
constexpr int kWidePD = 3;  // A-operand prefetch distance in k-groups (one group = 4 k-steps = 4 NT CT MFMAs = 256 NT CT cycles)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// The ring of A groups of one layer: entries 0 .. kWidePD-1 are filled by wide_prefetch, the layer loop rotates through all of
// them IN PLACE (an asm load's destination must not be copied before its wait: hipcc believes it is written when the statement ends).
template <int NT>
struct WidePre {
  f32x4 a[kWidePD + 1][NT];
};

// The A-operand stream is issued and awaited by hand.  Left to hipcc, the loads carry a 64-bit multiply-add per address and --
// worse -- the s_waitcnt in front of the first MFMA group of every loop iteration degrades to "all but the newest loads"
// (vmcnt(2) instead of vmcnt(6): the insertion pass merges the loop-entry state, where few loads are younger than the ring's
// oldest entry), i.e. the prefetch distance collapses to one group once per iteration.  Here: scalar base pointer advanced per
// group (saddr form: no VALU in the address path), one 32-bit lane offset per tile, and COUNTED waits -- loads return in order, so
// "at most N younger loads outstanding" is exact; loads hipcc itself issues in between only make the wait stricter.
__device__ __forceinline__ void wide_gload(f32x4& dst, unsigned voff, const float* sbase) {
  // s_nop 4: the base pointer may have been restored from a spill lane by v_readlane (a VALU write of an SGPR) right in front of
  // this statement -- a VMEM read of that SGPR needs 5 wait states, and hipcc's hazard recogniser does not look inside asm
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int N, int NT>
__device__ __forceinline__ void wide_vmwait(f32x4 (&a)[NT]) {
  static_assert(NT == 1 || NT == 2, "tiles per wave");
  if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[0]) : "n"(N));
  else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
}
// wgrp: first group of the layer's packed weights (wave-uniform pointer); voff[k]: byte offset of (tile k of this wave, lane)
template <int NT>
__device__ __forceinline__ void wide_prefetch(WidePre<NT>& P, const float* __restrict__ wgrp, int grp_floats, int NS4,
                                              const unsigned (&voff)[NT]) {
#pragma unroll
  for (int g = 0; g < kWidePD; ++g)
#pragma unroll
    for (int k = 0; k < NT; ++k) wide_gload(P.a[g][k], voff[k], wgrp + (long long)(g < NS4 ? g : NS4 - 1) * grp_floats);
}

// acc[k][c] = sum over NS4 groups of W[tile t0 + 4 k][:] . act[:, column tile c]   (the accumulators start at zero: the layer's
// bias / time embedding is added when the result is activated).  grp_floats: floats per k-group of the packed layer (n_tiles * 256).
template <int NT, int CT>
__device__ __forceinline__ void wide_layer(WidePre<NT>& P, const float* __restrict__ wgrp, int grp_floats, int NS4,
                                           const unsigned (&voff)[NT], const float* __restrict__ actl, int RS,
                                           f32x16 (&acc)[NT][CT]) {
  constexpr int U = kWidePD + 1;  // ring of A groups; the B ring (LDS, distance 1) has two entries: U is even
  static_assert(U % 2 == 0, "the two-entry B ring rotates statically only under an even unroll");
  f32x4 (&a)[U][NT] = P.a;
  float b[2][4][CT];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][c][q] = 0.0f;
  auto loadB = [&](int S, float (&bv)[4][CT]) {
    const float* __restrict__ ap = actl + (8 * S) * RS;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CT; ++c) bv[e][c] = ap[2 * e * RS + 32 * c];
  };
  auto compute = [&](const f32x4 (&av)[NT], const float (&bv)[4][CT], int e0, int e1) {
#pragma unroll
    for (int e = e0; e < e1; ++e)
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[k][c] = SDEH_MFMA(av[k][e], bv[e][c], acc[k][c]);
  };
  const int last = NS4 - 1;
  auto cl = [&](int S) { return S < last ? S : last; };  // loads past the end re-read the last group (harmless)
  loadB(0, b[0]);
  const int main_end = NS4 - NS4 % U;
  const float* __restrict__ wnext = wgrp + (long long)cl(kWidePD) * grp_floats;  // group S + kWidePD of the stream
  for (int S = 0; S < main_end; S += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // The next operands are requested in the MIDDLE of the group's MFMAs: hipcc's wait for the B operands at the top of a loop
      // iteration degrades to lgkmcnt(0) (as its vmcnt would), and by then these reads are half a group (>= 128 cycles) old.
      wide_vmwait<(kWidePD - 1) * NT, NT>(a[u]);  // everything older than the kWidePD - 1 newest groups has landed
      SDEH_FENCE();
      compute(a[u], b[u % 2], 0, 2);
      SDEH_FENCE();
#pragma unroll
      for (int k = 0; k < NT; ++k) wide_gload(a[(u + kWidePD) % U][k], voff[k], wnext);
      wnext = S + u + kWidePD < last ? wnext + grp_floats : wnext;
      loadB(cl(S + u + 1), b[(u + 1) % 2]);
      SDEH_FENCE();
      compute(a[u], b[u % 2], 2, 4);
      SDEH_FENCE();
    }
  }
  // NS4 % U groups left: their A operands are in flight (ring entries 0 ..), b[0] holds the first one's B operands
#pragma unroll
  for (int u = 0; u < U - 1; ++u) {
    if (main_end + u < NS4) {
      if (u + 1 < U - 1) loadB(cl(main_end + u + 1), b[(u + 1) % 2]);
      wide_vmwait<0, NT>(a[u]);
      SDEH_FENCE();
      compute(a[u], b[u % 2], 0, 4);
      SDEH_FENCE();
    }
  }
  // The last iterations of the main loop requested groups past the end (clamped re-reads that nobody consumes).  They MUST have
  // landed before this function returns: the compiler sees their destination registers as dead and hands them to whatever comes
  // next, and a load that lands late overwrites that value (found by the wide random sweep: 64-channel networks at d > 128 -- a
  // short out layer with two tiles per wave -- came out wrong at one step of a launch).  Every ring entry is tied to the wait, so
  // none of them is handed out before it (when NS4 % U != 0 the tail above has already drained the queue: the waits are free).
#pragma unroll
  for (int u = 0; u < U; ++u) wide_vmwait<0, NT>(a[u]);
}

// out[channel][trajectory] <- act(acc + bias) for this wave's tiles (bias: this lane half's 16 values per tile, accumulator order);
// DSTORE: also act'(acc + bias) into `dout` (Bridge: the inference network)
template <int NT, int CT, bool DSTORE>
__device__ __forceinline__ void wide_act_store(f32x16 (&acc)[NT][CT], const f32x16 (&bias)[NT], int act, float* __restrict__ outl,
                                               float* __restrict__ doutl, int RS, int t0, int h) {
  SDEH_ACT_SWITCH(act, ACT,
    _Pragma("unroll") for (int k = 0; k < NT; ++k)
      _Pragma("unroll") for (int c = 0; c < CT; ++c) {
        f32x16 z = acc[k][c] + bias[k];
        if constexpr (DSTORE) {
          _Pragma("unroll") for (int q = 0; q < 16; ++q) doutl[(32 * (t0 + 4 * k) + rho(q, h)) * RS + 32 * c] = act_grad(z[q], ACT);
        }
        act_tile<ACT>(z);
        _Pragma("unroll") for (int q = 0; q < 16; ++q) outl[(32 * (t0 + 4 * k) + rho(q, h)) * RS + 32 * c] = z[q];
      });
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release/acquire fence for ALL address spaces: hipcc
// emits s_waitcnt vmcnt(0) in front of it, i.e. every barrier would wait for the next layer's just-issued operand prefetch (the
// full L2 latency, four times per step: measured 7 ms of a 34 ms launch).  The planes exchanged between the waves live in LDS, so
// lgkmcnt(0) is all the barrier needs; global loads stay in flight across it.
__device__ __forceinline__ void wide_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float half_sum(float v) { return sum_xor32(v); }  // v_permlane32_swap + add (no LDS round trip)

struct WideCtx {
  float* planes;       // LDS: plane p = planes + p * plane_floats ([channel | coordinate][trajectory])
  int plane_floats;
  float* scr;          // [kWideSlots][4][RS]
  const float* tab0;   // LDS copies of the Gaussian tables (target / prior / second): (mu, 1/sigma^2) per coordinate, then the constant
  const float* tab1;
  const float* tab2;
  const float* bias;   // LDS copy of the hidden-layer biases [n_hidden][C] and the out-layer bias [32 otd], accumulator order
  int RS, d;
  int wave, lane, j, h;
  __device__ __forceinline__ float* plane(int p) const { return planes + p * plane_floats; }
};

// One FourierMLP pass over the workgroup's CT column tiles (models/mlp.py:114-122): input plane cx.plane(p) (the state, rows =
// coordinates) -> out-layer accumulators of this wave's coordinate tiles (before the clamp).  Returns the plane index the LAST
// hidden activation was written to (the out-layer's input); the state may be published into the other one.
// single: only buf[0] exists (Bridge): every layer reads, barrier, writes in place, barrier.
template <int OTW, int CT, bool DSTORE>
__device__ __forceinline__ int wide_mlp(const WideCtx& cx, const float* __restrict__ ws, const WsLayout& L, int act, int p,
                                        bool single, float* __restrict__ dplanes, WidePre<OTW>& pre_in,
                                        const f32x16 (&emb)[OTW], const float* __restrict__ bias_lds, f32x16 (&out)[2][CT], int nto) {
  const int RS = cx.RS, C = L.c, OT = L.ot, w = cx.wave, h = cx.h, j = cx.j;
  const int plane = C * RS;  // floats per act' plane
  f32x16 acc[OTW][CT];
  unsigned voff[OTW], voff_o[2];  // byte offsets of (tile w + 4 k, lane) inside a k-group
  const int wh = w & (OT - 1);  // hidden-layer tile of this wave (OT = 2, 4 or 8: waves 2, 3 of a 64-channel network double 0, 1)
#pragma unroll
  for (int k = 0; k < OTW; ++k) voff[k] = (unsigned)(((wh + 4 * k) * 64 + cx.lane) * 16);
  voff_o[0] = (unsigned)((w * 64 + cx.lane) * 16);
  voff_o[1] = nto > 1 ? (unsigned)(((w + 4) * 64 + cx.lane) * 16) : voff_o[0];
  // C = 64 (two row tiles per layer, d > 64): waves 2 and 3 own no hidden tile.  They run the SAME instruction stream on a tile of
  // waves 0 / 1 and only skip the store: the hand-issued operand loads must not sit behind control flow (at a merge point hipcc
  // copies their destination registers -- before the data has landed: measured, every result off by 1e-2).
  const bool has = w < OT;
  wide_layer<OTW, CT>(pre_in, ws + L.w_in, OT * 256, L.dp8 / 8, voff, cx.plane(p) + h * RS + j, RS, acc);
  int q = single ? 0 : 1 - p;
  WidePre<OTW> pre_h;
  WidePre<2> pre_o;
  for (int l = 0; l <= L.n_hidden; ++l) {
    // the next layer's first operands travel while this layer's output is activated, stored and the barrier is crossed
    if (l < L.n_hidden) wide_prefetch<OTW>(pre_h, ws + L.w_hid + l * L.w_hid_stride, OT * 256, C / 8, voff);
    else wide_prefetch<2>(pre_o, ws + L.w_out, L.otd * 256, C / 8, voff_o);
    f32x16 bias[OTW];
#pragma unroll
    for (int k = 0; k < OTW; ++k) bias[k] = l == 0 ? emb[k] : load16(bias_lds + (l - 1) * C + ((wh + 4 * k) * 2 + h) * 16);
    if (single) wide_barrier();  // everyone has read the plane that is about to be overwritten
    if (has) wide_act_store<OTW, CT, DSTORE>(acc, bias, act, cx.plane(q) + j, DSTORE ? dplanes + l * plane + j : nullptr, RS, wh, h);
    wide_barrier();
    if (l == L.n_hidden) break;
    wide_layer<OTW, CT>(pre_h, ws + L.w_hid + l * L.w_hid_stride, OT * 256, C / 8, voff, cx.plane(q) + h * RS + j, RS, acc);
    if (!single) q = 1 - q;
  }
  // out_layer on this wave's coordinate tiles {w, w + 4}; its bias is added by the caller
  if (nto == 2) {
    wide_layer<2, CT>(pre_o, ws + L.w_out, L.otd * 256, C / 8, voff_o, cx.plane(q) + h * RS + j, RS, out);
  } else {  // one tile (or none: the wave still drains its prefetch)
    WidePre<1> p1;
    unsigned v1[1] = {voff_o[0]};
#pragma unroll
    for (int g = 0; g < kWidePD; ++g) {
      wide_vmwait<0, 2>(pre_o.a[g]);  // landed (and tied to the wait) before it is copied
      p1.a[g][0] = pre_o.a[g][0];
    }
    if (nto == 1) {
      f32x16 o1[1][CT];
      wide_layer<1, CT>(p1, ws + L.w_out, L.otd * 256, C / 8, v1, cx.plane(q) + h * RS + j, RS, o1);
#pragma unroll
      for (int c = 0; c < CT; ++c) out[0][c] = o1[0][c];
    }
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
// Mixture targets (distr/gauss.py:123-140 via torch.distributions.MixtureSameFamily) in the accumulator layout.
//   log p(x) = logsumexp_k( c_k - sum_c a_kc (x_c - mu_kc)^2 ),  a = 1 / (2 sigma^2),  c_k = log w~_k - sum_c (log sigma_kc + log sqrt(2 pi))
//   score_c  = sum_k r_k (mu_kc - x_c) 2 a_kc,  r = softmax_k(...)         (what the reference obtains by autograd, distr/base.py:130-137)
// The tables mu[K][d4], a[K][d4] live in LDS.  Each wave forms the partial logits of its own coordinates when it publishes a new
// state ([wave][k][trajectory]); after the publish barrier wave 0 adds the four partials, normalises (online max / sum) and leaves
// the responsibilities r[k][trajectory] for everyone; the elementwise phase then accumulates the score of its coordinates.
// ---------------------------------------------------------------------------------------------------------
struct WideGmm {
  const float* mu;   // LDS [K][d4]
  const float* a;    // LDS [K][d4]
  const float* ck;   // LDS [K]
  float* part;       // LDS [4][K][RS]
  float* resp;       // LDS [K][RS]
  float* lse;        // LDS [RS]: logsumexp of the published state (terminal log-density)
  int K, d4;
};

template <int CT>
__device__ __forceinline__ void wide_gmm_partials(const WideCtx& cx, const WideGmm& G, const f32x16 (&xr)[2][CT], int nto) {
  const int w = cx.wave, h = cx.h, j = cx.j, RS = cx.RS;
  for (int k = 0; k < G.K; ++k) {
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      if (kk < nto) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int cb = 32 * (w + 4 * kk) + 8 * g4 + 4 * h;
          if (cb < cx.d) {  // d4 >= cb + 4: rows are padded to a multiple of four coordinates (mu = 0, a = 0 there; x is 0)
            const float4 m4 = *reinterpret_cast<const float4*>(G.mu + k * G.d4 + cb);
            const float4 a4 = *reinterpret_cast<const float4*>(G.a + k * G.d4 + cb);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              const float t0 = xr[kk][c][4 * g4] - m4.x, t1 = xr[kk][c][4 * g4 + 1] - m4.y;
              const float t2 = xr[kk][c][4 * g4 + 2] - m4.z, t3 = xr[kk][c][4 * g4 + 3] - m4.w;
              acc[c] = fmaf(t0 * t0, a4.x, acc[c]); acc[c] = fmaf(t1 * t1, a4.y, acc[c]);
              acc[c] = fmaf(t2 * t2, a4.z, acc[c]); acc[c] = fmaf(t3 * t3, a4.w, acc[c]);
            }
          }
        }
      }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float v = half_sum(acc[c]);
      if (h == 0) G.part[(w * G.K + k) * RS + 32 * c + j] = v;
    }
  }
}

// wave 0, after the publish barrier: responsibilities and log-density of the published state
template <int CT>
__device__ __forceinline__ void wide_gmm_normalise(const WideCtx& cx, const WideGmm& G) {
  const int RS = cx.RS, j = cx.j, h = cx.h;
  if (h != 0) return;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int col = 32 * c + j;
    float m = -INFINITY;
    for (int k = 0; k < G.K; ++k) {
      const float* pp = G.part + k * RS + col;
      const float l = G.ck[k] - (((pp[0] + pp[G.K * RS]) + pp[2 * G.K * RS]) + pp[3 * G.K * RS]);
      G.resp[k * RS + col] = l;
      m = fmaxf(m, l);
    }
    float z = 0.0f;
    for (int k = 0; k < G.K; ++k) {
      const float e = __expf(G.resp[k * RS + col] - m);
      G.resp[k * RS + col] = e;
      z += e;
    }
    const float iz = 1.0f / z;
    for (int k = 0; k < G.K; ++k) G.resp[k * RS + col] *= iz;
    G.lse[col] = m + __logf(z);
  }
}

// score of the 16 coordinates of one tile x column tile (coordinate base cb = 32 t + 4 h)
__device__ __forceinline__ void wide_gmm_score16(const WideCtx& cx, const WideGmm& G, const f32x16& x, int cb, int col, float (&sc)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) sc[q] = 0.0f;
  for (int k = 0; k < G.K; ++k) {
    const float r = G.resp[k * cx.RS + col];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (cb + 8 * g4 < cx.d) {
        const float4 m4 = *reinterpret_cast<const float4*>(G.mu + k * G.d4 + cb + 8 * g4);
        const float4 a4 = *reinterpret_cast<const float4*>(G.a + k * G.d4 + cb + 8 * g4);
        sc[4 * g4] = fmaf(r * (a4.x + a4.x), m4.x - x[4 * g4], sc[4 * g4]);
        sc[4 * g4 + 1] = fmaf(r * (a4.y + a4.y), m4.y - x[4 * g4 + 1], sc[4 * g4 + 1]);
        sc[4 * g4 + 2] = fmaf(r * (a4.z + a4.z), m4.z - x[4 * g4 + 2], sc[4 * g4 + 2]);
        sc[4 * g4 + 3] = fmaf(r * (a4.w + a4.w), m4.w - x[4 * g4 + 3], sc[4 * g4 + 3]);
      }
    }
  }
}

// Score term of a control on 16 registers (one tile x column tile): mult * scale_score * clip(score mix, clip_score) * gamma
// (reparam.py:56-83 ScoreCtrl, 131-162 LerpCtrl, 166-178 LerpPriorCtrl, 185-197 LerpTargetCtrl); zeros for ClippedCtrl.  The
// wave-uniform switches (target kind, control kind) sit outside the 16-element loops.  psc: prior score (when need_p).
struct WideScore {
  int ctrl_kind, g;       // g: gamma row length (1 or padded d)
  bool need_t, need_p;
  DensArgs tgt;
  float wl, mult, scale_score, clip_score, g0;
  int d;
  const WideGmm* gmm;  // mixture target (null otherwise)
  const float* ext = nullptr;          // SDEH_DENS_EXTERNAL: the caller's score plane of this step, [B, d]
  long long ext_off[4] = {0, 0, 0, 0}; // ... and the row offset (row * d) of each column tile's trajectory of this lane
};
// sc[16]: the combined score entering the control BEFORE clip_score and gamma(t) (target score, lerp of prior and target score, ...);
// psc[16]: the prior score (when need_p).  Not called for ClippedCtrl.
template <bool GMM = false>
__device__ __forceinline__ void wide_score_mix16(const WideScore& S, const WideCtx& cx, const f32x16& x, int cb, int c, float fs,
                                                 float fx0, float fiv, float (&sc)[16], float (&psc)[16]) {
  auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
  if (S.need_p) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float2 pp = *reinterpret_cast<const float2*>(cx.tab1 + 2 * coord(q));
      psc[q] = (pp.x - x[q]) * pp.y;
    }
  }
  if (S.ctrl_kind == SDEH_CTRL_CLIPPED) return;
  if (S.need_t) {
    if (S.tgt.kind == SDEH_DENS_DIAG_GAUSS) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float2 pp = *reinterpret_cast<const float2*>(cx.tab0 + 2 * coord(q));
        sc[q] = (pp.x - x[q]) * pp.y;
      }
    } else if (S.tgt.kind == SDEH_DENS_MULTI_WELL) {  // distr/double_well.py:43-45,174-179
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float y = x[q] - S.tgt.p1;
        sc[q] = coord(q) < S.tgt.n_comp ? -4.0f * (y * y - S.tgt.p0) * y : -y;
      }
    } else if (S.tgt.kind == SDEH_DENS_FUNNEL) {  // distr/funnel.py:71-80
#pragma unroll
      for (int q = 0; q < 16; ++q) sc[q] = -x[q] * fiv;
      const float s0 = -fx0 / S.tgt.p0 - 0.5f * (float)(S.d - 1) + 0.5f * fs * fiv;
      sc[0] = cb == 0 ? s0 : sc[0];
    } else if (GMM && S.tgt.kind == SDEH_DENS_GMM) {
      if constexpr (GMM) wide_gmm_score16(cx, *S.gmm, x, cb, 32 * c + cx.j, sc);
    } else if (S.tgt.kind == SDEH_DENS_EXTERNAL) {  // supplied by the caller (sdeh_simulate_fwd_steps; e.g. the NICE flow's, sdeh_nice.hip)
      const float* __restrict__ er = S.ext + S.ext_off[c] + cb;
#pragma unroll
      for (int q = 0; q < 16; ++q) sc[q] = coord(q) < S.d ? er[(q & 3) + 8 * (q >> 2)] : 0.0f;
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) sc[q] = 0.0f;
    }
  }
  if (S.ctrl_kind == SDEH_CTRL_LERP) {  // reparam.py:131-144; torch.lerp's two-sided formula
    if (S.wl < 0.5f) {
#pragma unroll
      for (int q = 0; q < 16; ++q) sc[q] = psc[q] + S.wl * (sc[q] - psc[q]);
    } else {
      const float w1 = 1.0f - S.wl;
#pragma unroll
      for (int q = 0; q < 16; ++q) sc[q] = sc[q] - (sc[q] - psc[q]) * w1;
    }
  } else if (S.ctrl_kind == SDEH_CTRL_LERP_TARGET) {  // reparam.py:185-197
#pragma unroll
    for (int q = 0; q < 16; ++q) sc[q] = S.wl * sc[q];
  } else if (S.ctrl_kind == SDEH_CTRL_LERP_PRIOR) {  // reparam.py:166-178
    const float w1 = 1.0f - S.wl;
#pragma unroll
    for (int q = 0; q < 16; ++q) sc[q] = w1 * psc[q];
  }
}

template <bool GMM = false>
__device__ __forceinline__ void wide_score_term16(const WideScore& S, const WideCtx& cx, const f32x16& x, int cb, int c, float fs,
                                                  float fx0, float fiv, const float* __restrict__ gam_row, float (&sterm)[16],
                                                  float (&psc)[16], float* __restrict__ sc_raw = nullptr) {
  auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
  float sc[16];
  wide_score_mix16<GMM>(S, cx, x, cb, c, fs, fx0, fiv, sc, psc);
  if (sc_raw != nullptr && S.ctrl_kind != SDEH_CTRL_CLIPPED) {  // training forward on a mixture target: the score before clip / gamma
#pragma unroll
    for (int q = 0; q < 16; ++q) sc_raw[q] = sc[q];
  }
  if (S.ctrl_kind != SDEH_CTRL_CLIPPED) {
    if (S.g == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) sterm[q] = S.mult * ((S.scale_score * clipf(sc[q], S.clip_score)) * S.g0);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) sterm[q] = S.mult * ((S.scale_score * clipf(sc[q], S.clip_score)) * gam_row[coord(q)]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) sterm[q] = 0.0f;
  }
}

// Four Philox blocks -> the 16 standard normals of one tile x column tile (register group g4 = coordinates cb + 8 g4 .. + 3 = block
// (cb + 8 g4) / 4), or the same 16 values read from the caller's noise tensor (parity mode)
__device__ __forceinline__ void wide_noise16(const float* __restrict__ noise_row, bool vec4, int cb, int d, unsigned long long seed,
                                             unsigned long long rng_off, unsigned long long grow, int step, float (&n)[16]) {
  auto coord = [&](int q) { return cb + (q & 3) + 8 * (q >> 2); };
  if (noise_row != nullptr) {
    const float* __restrict__ np = noise_row + cb;
    if (vec4) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float4 t4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
        if (cb + 8 * g4 < d) t4 = *reinterpret_cast<const float4*>(np + 8 * g4);
        n[4 * g4] = t4.x; n[4 * g4 + 1] = t4.y; n[4 * g4 + 2] = t4.z; n[4 * g4 + 3] = t4.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) n[q] = coord(q) < d ? np[(q & 3) + 8 * (q >> 2)] : 0.0f;
    }
  } else {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float n4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (cb + 8 * g4 < d) box_muller4(philox_block(seed, rng_off, grow, step, (cb + 8 * g4) >> 2), n4);
#pragma unroll
      for (int e = 0; e < 4; ++e) n[4 * g4 + e] = n4[e];
      SDEH_FENCE();
    }
  }
}


}  // namespace sdeh
