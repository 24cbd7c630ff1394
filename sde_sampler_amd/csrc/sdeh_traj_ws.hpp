// Wave-specialised persistent trajectory kernel (the default path).
//
// A workgroup owns G groups of 64 trajectories (G = 4: 512 threads, 256 trajectories).  Each group is served by TWO
// wavefronts -- waves g and G + g; a workgroup's waves are dealt to the CU's four SIMDs cyclically, so with G = 4 both
// land on the same SIMD:
//
//   V wave (waves 0..G-1):  owns the state.  T layout (lane = trajectory): target/prior score, score term of the
//                           control, Philox/Box-Muller draws, running cost, EM update, terminal log-densities.
//                           Pure VALU + broadcast LDS reads.
//   M wave (waves G..2G-1): evaluates the FourierMLP for the same 64 trajectories on the matrix pipe
//                           (v_mfma_f32_32x32x2_f32, M layout) with the activation in between.
//
// fp32 MFMA and fp32 VALU instructions share the SIMD's fp32 datapath on gfx950 (profiles/r01_ubench_coexec.txt): two
// waves on one SIMD do not add throughput, they hide each other's latencies (LDS, MFMA result latency, barriers).
// B = 65 536 is exactly one group per SIMD (1024 SIMDs x 64 lanes), so there G = 4 and the pair shares a SIMD.  Smaller
// batches use groups of 32 trajectories (TrajArgs::half: one column tile per M wave, ws_mlp_half) and, when every wave can
// have a SIMD of its own, G = 2 so that the V and the M wave of a group sit on DIFFERENT SIMDs and really run concurrently
// (launch_traj_ws has the table).  Per step the waves exchange x (V -> M) and the network output
// (M -> V) through one [coordinate][trajectory] LDS buffer per group, which also performs the T <-> M layout change
// (no cross-lane shuffles), separated by two workgroup barriers:
//
//   V: score(x), noise            | barrier B | u = clip(nn) + score term, x <- EM(x,u,xi), publish x | barrier A | cost, Ito
//   M: read x, MLP(x), publish nn | barrier B | (prefetch next step's time embedding)                  | barrier A
#pragma once
#include "sdeh_traj.hpp"
#ifdef SDEH_WS_PROFILE
#include <cstdio>
#endif

#ifndef SDEH_MPRIO
#define SDEH_MPRIO 3
#endif
#ifndef SDEH_VPRIO
#define SDEH_VPRIO 0
#endif

namespace sdeh {

// Per-step workgroup barrier that orders LDS traffic only.  __syncthreads() carries a fence for ALL address spaces, i.e. hipcc puts
// s_waitcnt vmcnt(0) in front of it: the M wave's prefetch of the next step's time embedding (a global load issued between the two
// barriers of a step), the V wave's xs / plane stores and parity-mode noise loads would all be drained at every barrier.  What the
// waves exchange lives in LDS (the exchange buffers, the pair mode's activation parking), so lgkmcnt(0) is all a step needs.
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pair-level hand-off without the workgroup barrier: a group's V and M wave only ever wait for EACH OTHER (x from V to M, the
// network output from M to V), but s_barrier makes all groups of the workgroup wait for the slowest pair twice per step.  Each
// group owns two step counters in LDS; the producer bumps its counter after its LDS writes (a wave's LDS operations execute in
// order, so the data is visible before the counter), the consumer polls with s_sleep between the reads (it shares its SIMD with
// the producer at G = 4: a sleeping wave issues nothing).
__device__ __forceinline__ void ws_flag_set(int* flag, int value) {
  // the data stores of this wave have COMPLETED before the flag store is issued (lgkmcnt counts LDS operations): the ordering the
  // consumer relies on is architectural, not a property of the LDS queue being in-order
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  *reinterpret_cast<volatile int*>(flag) = value;
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ws_flag_wait(int* flag, int value) {
  asm volatile("" ::: "memory");
  while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(flag)) < value) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

// -DSDEH_WS_PROFILE: cycle counts of the M and the V wave of group 0 in block 0 (measurement build, tools/ws_phase_profile.sh):
// [0] M: network pass  [1] M: waiting for x  [2] V: score + noise + partial update  [3] V: waiting for the network output
// [4] V: hand-off work (u, x, publish)  [5] V: cost / Ito / stores  [6] steps
#ifdef SDEH_WS_PROFILE
static __device__ unsigned long long ws_prof[8];
#define WS_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define WS_ADD(k, t0, t1) do { if (blockIdx.x == 0 && lane == 0) atomicAdd(&ws_prof[k], (t1) - (t0)); } while (0)
#else
#define WS_T(var) do {} while (0)
#define WS_ADD(k, t0, t1) do {} while (0)
#endif

constexpr int kWsGroups = 4;  // most trajectory groups (of 64) per workgroup; the launcher picks 4 or 2 (blockDim.x = 128 G)

// rows of the exchange buffer: every coordinate an M-layout register can address
template <int DP>
constexpr int xrows() { return mdim(mregs(DP) - 1, 1) + 1; }

// ---------------------------------------------------------------------------------------------------------
// GMM with LDS tables, single pass with an online softmax over chunks of 8 components (no logit scratch).
// SHARED: tables of the shared-scale form (see gmm_eval_lds_shared).  Table rows are padded to a multiple of
// 8 components; padding rows carry logit -inf.
// ---------------------------------------------------------------------------------------------------------
template <int NQ, int ROWS, class F>
__device__ __forceinline__ void stream_rows_n(const float4* __restrict__ tab, F&& f) {
  constexpr int NA = (NQ + 1) / 2, NB = NQ - NA;
  float4 qa[NA], qb[NB > 0 ? NB : 1];
  load_f4<NA>(tab, qa);
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const float4* __restrict__ row = tab + k * NQ;
    if constexpr (NB > 0) load_f4<NB>(row + NA, qb);
    SDEH_FENCE();
    f(k, std::integral_constant<int, 0>{}, qa);
    SDEH_FENCE();
    if (k + 1 < ROWS) load_f4<NA>(tab + (k + 1) * NQ, qa);
    SDEH_FENCE();
    if constexpr (NB > 0) f(k, std::integral_constant<int, NA>{}, qb);
    SDEH_FENCE();
  }
}

// fp32 VALU ops process 16 lanes per cycle on gfx950 (a wave64 v_fma_f32 occupies the SIMD for 4 cycles -- measured:
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.1); the 157 TFLOP/s vector peak is only reached with the packed forms
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two fp32 per lane, same 4 cycles).  The mixture loops therefore work
// on coordinate PAIRS held in 64-bit register pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

// Table layouts (rows of NQ float4, prepared by sdeh_prep.hip):
//   SHARED : logit table  m_kd = mu_kd / (sqrt2 sigma_d), four coordinates per float4; score table mu_kd / sigma_d^2;
//            followed by the vectors 1/(sqrt2 sigma_d) and 1/sigma_d^2.
//   general: logit table (mu_d, mu_d+1, a_d, a_d+1) with a = 1/(2 sigma^2), two coordinates per float4;
//            score table (mu/sigma^2 at d, d+1, 1/sigma^2 at d, d+1).
//   SHARED with NV < DP: the tables cover only the first NV coordinates (SDEH_DENS_FLAG_NVARY: the others are identical in
//            every component); those factor out as one Gaussian -- they cancel in the responsibilities and add
//            (mu_0d - x_d)/sigma_d^2 to the score and -(x_d - mu_0d)^2/(2 sigma_d^2) to the log-density.
// KS (groups of 32 trajectories, where lanes 32..63 of the V wave idle): the component chunks alternate between the lane halves --
// the upper half evaluates the odd chunks for trajectory lane - 32 (it gets that trajectory's table coordinates through
// v_permlane32_swap) and the two online-softmax states (m, z, P, Q) are merged at the end, in both halves alike.  Only where the tables
// cover at most 8 coordinates (the exchange costs a swap per coordinate); the chunk order differs from the unsplit evaluation, so this is
// for the small-batch modes whose results are not bit-identical to the large-batch ones anyway (pair / quad mode).
template <int DP, bool SHARED, bool SCORE, int NV, bool KS = false>
__device__ __forceinline__ float gmm_online(const float* __restrict__ lds, const WsLayout& L, int K,
                                            const float (&x)[DP], float (&score)[DP]) {
  static_assert(NV == DP || (SHARED && NV % 4 == 0 && NV < DP), "NV: DP, or a multiple of 4 below DP (shared scale)");
  if constexpr (KS && (SHARED ? NV : DP) > 8) return gmm_online<DP, SHARED, SCORE, NV, false>(lds, L, K, x, score);
  constexpr int CH = 8;
  constexpr int NP = (NV + 1) / 2;                           // coordinate pairs the tables cover
  constexpr int NQ = SHARED ? (NV + 3) / 4 : (DP + 1) / 2;  // float4 per table row
  constexpr int RSF = 4 * ((DP + 3) / 4);                   // stride of the per-coordinate vectors
  constexpr int NA = (NQ + 1) / 2;
  const int KR = L.gmm_rows;  // multiple of CH
  const float* __restrict__ pc = lds + L.gmm_c;
  const float4* __restrict__ plg = reinterpret_cast<const float4*>(lds + L.gmm_lg);
  const float4* __restrict__ psc = reinterpret_cast<const float4*>(lds + L.gmm_sc);
  const float* __restrict__ vec = lds + L.gmm_vec;  // SHARED: 1/(sqrt2 s_d), 1/s_d^2, mu_0d/(sqrt2 s_d), mu_0d/s_d^2
  // lower lane half's value in every lane (KS: the coordinates of the trajectory both halves work on)
  auto lower = [](float v) {
    if constexpr (KS) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      return __uint_as_float(r[0]);
    } else {
      return v;
    }
  };
  const int hh = KS ? (int)((threadIdx.x & 63) >> 5) : 0;
  f2 y[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float x0 = lower(x[2 * p]), x1 = 2 * p + 1 < DP ? lower(x[2 * p + 1]) : 0.0f;
    y[p] = SHARED ? f2{x0 * vec[2 * p], x1 * vec[2 * p + 1]} : f2{x0, x1};
  }
  float m = -INFINITY, z = 0.0f;
  f2 P[NP], Q[SHARED ? 1 : NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) { P[p] = splat(0.0f); if (!SHARED) Q[p] = splat(0.0f); }
  for (int cp = 0; cp < KR; cp += (KS ? 2 : 1) * CH) {
    // KS: this half's chunk of the pair; the upper half's last chunk may not exist (an odd number of chunks): it re-reads the last
    // one with logits -inf (weights zero)
    const bool have = !KS || cp + hh * CH < KR;
    const int c0 = KS ? (have ? cp + hh * CH : KR - CH) : cp;
    float l[CH];
    {  // the chunk's constants first: a read issued inside the stream would drain the prefetch queue (lgkmcnt(0))
      const float4 c_lo = *reinterpret_cast<const float4*>(pc + c0), c_hi = *reinterpret_cast<const float4*>(pc + c0 + 4);
      l[0] = c_lo.x; l[1] = c_lo.y; l[2] = c_lo.z; l[3] = c_lo.w;
      l[4] = c_hi.x; l[5] = c_hi.y; l[6] = c_hi.z; l[7] = c_hi.w;
      if constexpr (KS) {
#pragma unroll
        for (int k = 0; k < CH; ++k) l[k] = have ? l[k] : -INFINITY;
      }
    }
    SDEH_FENCE();
    f2 acc0 = splat(0.0f), acc1 = splat(0.0f);
    stream_rows_n<NQ, CH>(plg + c0 * NQ, [&](int k, auto J0, const auto& q) {
      constexpr int j0 = decltype(J0)::value;
      constexpr int n = j0 == 0 ? NA : NQ - NA;
#pragma unroll
      for (int j = 0; j < n; ++j) {
        if constexpr (SHARED) {
          const int p = 2 * (j0 + j);
          const f2 t0 = y[p] - f2{q[j].x, q[j].y};
          acc0 = pk_fma(t0, t0, acc0);
          if (p + 1 < NP) {
            const f2 t1 = y[p + 1] - f2{q[j].z, q[j].w};
            acc1 = pk_fma(t1, t1, acc1);
          }
        } else {
          const int p = j0 + j;
          const f2 t = y[p] - f2{q[j].x, q[j].y};
          if (j & 1) acc1 = pk_fma(t * t, f2{q[j].z, q[j].w}, acc1);
          else acc0 = pk_fma(t * t, f2{q[j].z, q[j].w}, acc0);
        }
      }
      if (j0 != 0 || NQ == 1) {  // row complete
        const f2 a = acc0 + acc1;
        l[k] -= a.x + a.y;
        acc0 = acc1 = splat(0.0f);
      }
    });
    float cm = l[0];
#pragma unroll
    for (int k = 1; k < CH; ++k) cm = fmaxf(cm, l[k]);
    // (KS: a half without a chunk sees logits -inf only; the finite floor keeps exp(m - mn) = exp(-inf) = 0 instead of exp(NaN))
    const float mn = KS ? fmaxf(fmaxf(m, cm), -3.0e38f) : fmaxf(m, cm);
    const float resc = __expf(m - mn);  // exp(-inf) = 0 on the first chunk
    m = mn;
    float e[CH];
    z *= resc;
#pragma unroll
    for (int k = 0; k < CH; ++k) { e[k] = __expf(l[k] - m); z += e[k]; }
    if constexpr (SCORE) {
      const f2 r2 = splat(resc);
#pragma unroll
      for (int p = 0; p < NP; ++p) { P[p] *= r2; if (!SHARED) Q[p] *= r2; }
      stream_rows_n<NQ, CH>(psc + c0 * NQ, [&](int k, auto J0, const auto& q) {
        constexpr int j0 = decltype(J0)::value;
        constexpr int n = j0 == 0 ? NA : NQ - NA;
        const f2 e2 = splat(e[k]);
#pragma unroll
        for (int j = 0; j < n; ++j) {
          if constexpr (SHARED) {
            const int p = 2 * (j0 + j);
            P[p] = pk_fma(e2, f2{q[j].x, q[j].y}, P[p]);
            if (p + 1 < NP) P[p + 1] = pk_fma(e2, f2{q[j].z, q[j].w}, P[p + 1]);
          } else {
            const int p = j0 + j;
            P[p] = pk_fma(e2, f2{q[j].x, q[j].y}, P[p]);
            Q[p] = pk_fma(e2, f2{q[j].z, q[j].w}, Q[p]);
          }
        }
      });
    }
  }
  if constexpr (KS) {  // merge the two halves' online-softmax states (both halves end up with the same totals)
    auto both = [](float v, float& lo, float& hi) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      lo = __uint_as_float(r[0]);
      hi = __uint_as_float(r[1]);
    };
    float m0, m1, z0, z1;
    both(m, m0, m1);
    both(z, z0, z1);
    const float mm = fmaxf(m0, m1);
    const float s0 = __expf(m0 - mm), s1 = __expf(m1 - mm);  // (m1 = -inf when the upper half had no component: s1 = 0)
    m = mm;
    z = fmaf(z0, s0, z1 * s1);
    if constexpr (SCORE) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float a0, a1, b0, b1;
        both(P[p].x, a0, a1);
        both(P[p].y, b0, b1);
        P[p] = f2{fmaf(a0, s0, a1 * s1), fmaf(b0, s0, b1 * s1)};
        if constexpr (!SHARED) {
          both(Q[p].x, a0, a1);
          both(Q[p].y, b0, b1);
          Q[p] = f2{fmaf(a0, s0, a1 * s1), fmaf(b0, s0, b1 * s1)};
        }
      }
    }
  }
  if constexpr (SCORE) {
    const float iz = 1.0f / z;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      if constexpr (SHARED) {
        if (d < NV) score[d] = fmaf((d & 1) ? P[d / 2].y : P[d / 2].x, iz, -x[d] * vec[RSF + d]);
        else score[d] = fmaf(-x[d], vec[RSF + d], vec[3 * RSF + d]);
      } else {
        const float Pd = (d & 1) ? P[d / 2].y : P[d / 2].x, Qd = (d & 1) ? Q[d / 2].y : Q[d / 2].x;
        score[d] = (Pd - x[d] * Qd) * iz;
      }
    }
  }
  float logp = m + __logf(z);
  if constexpr (SHARED && NV < DP) {
    float rest = 0.0f;
#pragma unroll
    for (int d = NV; d < DP; ++d) {
      const float t = fmaf(x[d], vec[d], -vec[2 * RSF + d]);
      rest = fmaf(t, t, rest);
    }
    logp -= rest;
  }
  return logp;
}

// ---------------------------------------------------------------------------------------------------------
// The same online softmax with the tables streamed through the SCALAR cache (round 5).  A table entry is the same for all 64
// trajectories of the wave: as broadcast ds_read_b128 it costs the LDS four cycles and 64 x 16 bytes of register-file writes for 16
// useful bytes, per V wave.  Here the rows arrive as s_load_dwordx16 (eight coordinate pairs per instruction) and enter v_pk_add_f32 /
// v_pk_fma_f32 directly as SGPR-pair operands: no vector registers, no LDS cycles, no LDS wait on the V wave's critical path.  Measured
// (profiles/r05_dense_mixture_timing.txt): a two-wave-per-SIMD micro-benchmark is vector-issue bound in either form
// (tools/ubench/smem_stream.hip), the kernel is not -- per-component scales (2000 reads per step) 5.64 -> 4.58 ms, the headline's 4-coordinate
// tables 2.163 -> 2.131 ms, and on 32-lane groups (B = 32 768), where the V wave's waits are exposed, 2.82 -> 2.45 ms.  The tables are read where the prep kernel wrote them (the global copy of the LDS image), in the
// same layout, and every floating-point operation is the one gmm_online performs, in the same order: results are bit-identical.
//
// Stream: a chunk of 8 rows is NQ batches of two s_load_dwordx16 (16 SGPR pairs); batch n + 1 is requested before the vector work
// on batch n (scalar loads return out of order -- the only wait there is is lgkmcnt(0), placed in front of the request), the logit
// stream of a chunk is followed by its score stream and that by the next chunk's logit stream without a bubble.
typedef unsigned long long u64x8 __attribute__((ext_vector_type(8)));  // eight SGPR pairs = one s_load_dwordx16
typedef u64x8 u64x8a __attribute__((aligned(8)));
typedef const u64x8a __attribute__((address_space(4))) * ctab16;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// hipcc unpacks packed fp32 operations with a scalar-register operand into two scalar-operand v_sub / v_fma: written in assembly
__device__ __forceinline__ f2 pk_sub_s(f2 a, unsigned long long s) {  // a - s
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "s"(s));
  return r;
}
__device__ __forceinline__ void pk_fma_acc_s(f2& acc, f2 a, unsigned long long s) {  // acc += a * s
  asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(s));
}
__device__ __forceinline__ void s_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }  // lgkmcnt(0), other counters untouched

template <int DP, bool SHARED, bool SCORE, int NV>
__device__ __forceinline__ float gmm_online_s(const float* __restrict__ ws, const WsLayout& L, const float (&x)[DP],
                                              float (&score)[DP]) {
  static_assert(NV == DP || (SHARED && NV % 4 == 0 && NV < DP), "NV: DP, or a multiple of 4 below DP (shared scale)");
  constexpr int CH = 8;
  constexpr int NP = (NV + 1) / 2;                           // coordinate pairs the tables cover
  constexpr int NQ = SHARED ? (NV + 3) / 4 : (DP + 1) / 2;  // float4 per table row
  constexpr int PR = 2 * NQ;                                 // SGPR pairs per table row
  constexpr int NB = NQ;                                     // batches of 16 SGPR pairs per chunk of 8 rows
  constexpr int NA = (NQ + 1) / 2;                           // (gmm_online's split of a row, kept for its accumulator order)
  constexpr int RSF = 4 * ((DP + 3) / 4);
  constexpr int NS = SCORE ? 2 : 1;                          // streams per chunk
  const int KR = L.gmm_rows;  // multiple of CH
  cfp vec = as_const(ws + L.gmm_vec);
  cfp pc = as_const(ws + L.gmm_c);
  ctab16 glg = (ctab16)(unsigned long long)(ws + L.gmm_lg);
  ctab16 gsc = (ctab16)(unsigned long long)(ws + L.gmm_sc);
  f2 y[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float x0 = x[2 * p], x1 = 2 * p + 1 < DP ? x[2 * p + 1] : 0.0f;
    y[p] = SHARED ? f2{x0 * vec[2 * p], x1 * vec[2 * p + 1]} : f2{x0, x1};
  }
  float m = -INFINITY, z = 0.0f;
  f2 P[NP], Q[SHARED ? 1 : NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) { P[p] = splat(0.0f); if (!SHARED) Q[p] = splat(0.0f); }
  u64x8 buf[2][2];
  buf[0][0] = glg[0];
  buf[0][1] = glg[1];
  for (int cp = 0; cp < KR; cp += CH) {
    ctab16 lg = glg + (cp / CH) * (2 * NB), sc = gsc + (cp / CH) * (2 * NB);
    ctab16 lg_next = glg + (cp + CH < KR ? cp / CH + 1 : 0) * (2 * NB);  // (the last chunk re-requests the first: never used)
    float l[CH];
    {
      cfp c = pc + cp;
#pragma unroll
      for (int k = 0; k < CH; ++k) l[k] = c[k];
    }
    f2 acc0 = splat(0.0f), acc1 = splat(0.0f), tt = splat(0.0f);
    // ---- logit stream
    static_for<NB>([&](auto Bc) {
      constexpr int b = decltype(Bc)::value;
      constexpr int cur = b & 1, nxt = cur ^ 1;
      s_wait_lgkm0();  // batch b has landed
      SDEH_FENCE();
      if constexpr (b + 1 < NB) { buf[nxt][0] = lg[2 * (b + 1)]; buf[nxt][1] = lg[2 * (b + 1) + 1]; }
      else if constexpr (SCORE) { buf[nxt][0] = sc[0]; buf[nxt][1] = sc[1]; }
      else { buf[nxt][0] = lg_next[0]; buf[nxt][1] = lg_next[1]; }
      SDEH_FENCE();
      static_for<16>([&](auto Gc) {
        constexpr int g = 16 * b + decltype(Gc)::value;  // SGPR pair within the chunk
        constexpr int k = g / PR, s = g % PR;
        const unsigned long long q = buf[cur][decltype(Gc)::value / 8][decltype(Gc)::value % 8];
        if constexpr (SHARED) {
          if constexpr (s < NP) {
            const f2 t = pk_sub_s(y[s], q);
            if constexpr (s & 1) acc1 = pk_fma(t, t, acc1);
            else acc0 = pk_fma(t, t, acc0);
          }
        } else {
          constexpr int j = s / 2, jl = j < NA ? j : j - NA;
          if constexpr ((s & 1) == 0) {
            const f2 t = pk_sub_s(y[j], q);
            tt = t * t;
          } else {
            if constexpr (jl & 1) pk_fma_acc_s(acc1, tt, q);
            else pk_fma_acc_s(acc0, tt, q);
          }
        }
        if constexpr (s == PR - 1) {  // row complete
          const f2 a = acc0 + acc1;
          l[k] -= a.x + a.y;
          acc0 = acc1 = splat(0.0f);
        }
      });
      SDEH_FENCE();
    });
    float cm = l[0];
#pragma unroll
    for (int k = 1; k < CH; ++k) cm = fmaxf(cm, l[k]);
    const float mn = fmaxf(m, cm);
    const float resc = __expf(m - mn);  // exp(-inf) = 0 on the first chunk
    m = mn;
    float e[CH];
    z *= resc;
#pragma unroll
    for (int k = 0; k < CH; ++k) { e[k] = __expf(l[k] - m); z += e[k]; }
    if constexpr (SCORE) {
      const f2 r2 = splat(resc);
#pragma unroll
      for (int p = 0; p < NP; ++p) { P[p] *= r2; if (!SHARED) Q[p] *= r2; }
      static_for<NB>([&](auto Bc) {
        constexpr int b = decltype(Bc)::value;
        constexpr int cur = (NB + b) & 1, nxt = cur ^ 1;
        s_wait_lgkm0();
        SDEH_FENCE();
        if constexpr (b + 1 < NB) { buf[nxt][0] = sc[2 * (b + 1)]; buf[nxt][1] = sc[2 * (b + 1) + 1]; }
        else { buf[nxt][0] = lg_next[0]; buf[nxt][1] = lg_next[1]; }
        SDEH_FENCE();
        static_for<16>([&](auto Gc) {
          constexpr int g = 16 * b + decltype(Gc)::value;
          constexpr int k = g / PR, s = g % PR;
          const unsigned long long q = buf[cur][decltype(Gc)::value / 8][decltype(Gc)::value % 8];
          const f2 e2 = splat(e[k]);
          if constexpr (SHARED) {
            if constexpr (s < NP) pk_fma_acc_s(P[s], e2, q);
          } else {
            if constexpr ((s & 1) == 0) pk_fma_acc_s(P[s / 2], e2, q);
            else pk_fma_acc_s(Q[s / 2], e2, q);
          }
        });
        SDEH_FENCE();
      });
    }
    if constexpr ((NS * NB) & 1) {  // an odd number of batches per chunk: the request for the next chunk went to the other buffer
      buf[0][0] = buf[1][0];
      buf[0][1] = buf[1][1];
    }
  }
  s_wait_lgkm0();  // the last (unused) request
  if constexpr (SCORE) {
    const float iz = 1.0f / z;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      if constexpr (SHARED) {
        if (d < NV) score[d] = fmaf((d & 1) ? P[d / 2].y : P[d / 2].x, iz, -x[d] * vec[RSF + d]);
        else score[d] = fmaf(-x[d], vec[RSF + d], vec[3 * RSF + d]);
      } else {
        const float Pd = (d & 1) ? P[d / 2].y : P[d / 2].x, Qd = (d & 1) ? Q[d / 2].y : Q[d / 2].x;
        score[d] = (Pd - x[d] * Qd) * iz;
      }
    }
  }
  float logp = m + __logf(z);
  if constexpr (SHARED && NV < DP) {
    float rest = 0.0f;
#pragma unroll
    for (int d = NV; d < DP; ++d) {
      const float t = fmaf(x[d], vec[d], -vec[2 * RSF + d]);
      rest = fmaf(t, t, rest);
    }
    logp -= rest;
  }
  return logp;
}

// tables through the scalar cache: dense tables only (where the table stream, not the exchange, bounds the V wave), whole waves only
#ifndef SDEH_GMM_SGPR
#define SDEH_GMM_SGPR 1
#endif
#ifndef SDEH_GMM_SGPR_MIN
#define SDEH_GMM_SGPR_MIN 3  // tables over at most this many coordinates keep the LDS form (d = 2: no difference; 4 coordinates: -1.5 %, profiles/r05_dense_mixture_timing.txt)
#endif
template <int DP, bool SHARED, int NV>
constexpr bool gmm_use_sgpr() { return SDEH_GMM_SGPR != 0 && (SHARED ? NV : DP) > SDEH_GMM_SGPR_MIN; }

// ---------------------------------------------------------------------------------------------------------
// The mixture's two contractions on the MATRIX pipe, inside the V wave (round 5; WsLayout::gmm_lds == 3, shared scale, full tables).
//
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4 x 4 x 1 outer products: B is one value per lane (block b, column j = lane 4 b + j)
// and D[i][j] lands in register i of the SAME lane.  With B = x_d of the lane's own trajectory, register i accumulates
// sum_d A[i][d] x_d: four components' logits of the lane's own trajectory -- the T layout on both sides, no exchange, no layout
// change.  A is wave-uniform table data: CBSZ = 4 broadcasts the A values of block ABID to all 16 blocks, so one operand register
// carries the rows of 16 different instructions and a table is read from LDS once per step (8 + 9 ds_read_b128), not once per lane.
//   logits:  l_k = cc_k + sum_d x_d mu_kd / sigma_d^2      (cc_k = c_k - sum_d mu_kd^2 / (2 sigma_d^2); the term -sum_d x_d^2 / (2 sigma_d^2)
//            is common to all components and cancels in the responsibilities)            500 instructions for K = 40, d = 50
//   score:   P_d = sum_k e_k mu_kd / sigma_d^2,  score_d = P_d / z - x_d / sigma_d^2      520 instructions
// 8 cycles each on the fp32 matrix rate (64 FLOP per cycle and SIMD) against ~3100 packed vector instructions of ~5.4 cycles
// (tools/ubench/mfma4x4.hip, valu_dep.hip).  The logits are a PRODUCT form: they carry the rounding of sum |x_d mu_kd| / sigma^2, not of
// the squared distance -- the binding only selects this path where that cannot move a responsibility (SDEH_DENS_FLAG_MM_OK,
// engine._mixture_mm_ok: well-separated components); the terminal log-density stays on the exact form (gmm_online_s).
// The component count is a compile-time constant of the instruction stream: mixtures of 21 .. 40 components (SDEH_MM_K rows).
#ifndef SDEH_MM_K
#define SDEH_MM_K 40
#endif
typedef float mm4 __attribute__((ext_vector_type(4)));
template <int DP>
__device__ __forceinline__ void gmm_mm(const float* __restrict__ ws, const float* __restrict__ lds, const WsLayout& L,
                                       const float (&x)[DP], float (&score)[DP]) {
  constexpr int K4 = SDEH_MM_K / 4, D4 = (DP + 3) / 4;
  constexpr int N1 = DP * K4, N2 = SDEH_MM_K * D4;          // instructions per contraction
  constexpr int Q1 = (N1 + 63) / 64, Q2 = (N2 + 63) / 64;   // ds_read_b128 per contraction (64 instructions each)
  constexpr int RSF = 4 * ((DP + 3) / 4);
  const int lane = threadIdx.x & 63;
  const mm4* __restrict__ a1 = reinterpret_cast<const mm4*>(lds + L.gmm_mm1) + lane;
  const mm4* __restrict__ a2 = reinterpret_cast<const mm4*>(lds + L.gmm_mm2) + lane;
  cfp cc = as_const(ws + L.gmm_cc);
  cfp vec = as_const(ws + L.gmm_vec);
  mm4 lg[K4];
#pragma unroll
  for (int g = 0; g < K4; ++g) lg[g] = mm4{cc[4 * g], cc[4 * g + 1], cc[4 * g + 2], cc[4 * g + 3]};
  static_for<Q1>([&](auto Qc) {
    constexpr int q = decltype(Qc)::value;
    const mm4 a = a1[q * 64];
    static_for<64>([&](auto Nc) {
      constexpr int n = 64 * q + decltype(Nc)::value;
      if constexpr (n < N1) {
        constexpr int d = n / K4, g = n % K4, e = (n / 16) % 4, b = n % 16;
        lg[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], x[d], lg[g], 4, b, 0);
      }
    });
  });
  float m = lg[0][0];
#pragma unroll
  for (int g = 0; g < K4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) m = fmaxf(m, lg[g][i]);
  float z = 0.0f;
#pragma unroll
  for (int g = 0; g < K4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float e = __expf(lg[g][i] - m);  // padding rows: exp(-inf) = 0
      lg[g][i] = e;
      z += e;
    }
  mm4 P[D4];
#pragma unroll
  for (int g = 0; g < D4; ++g) P[g] = mm4{0.0f, 0.0f, 0.0f, 0.0f};
  static_for<Q2>([&](auto Qc) {
    constexpr int q = decltype(Qc)::value;
    const mm4 a = a2[q * 64];
    static_for<64>([&](auto Nc) {
      constexpr int n = 64 * q + decltype(Nc)::value;
      if constexpr (n < N2) {
        constexpr int k = n / D4, g = n % D4, e = (n / 16) % 4, b = n % 16;
        P[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], lg[k / 4][k % 4], P[g], 4, b, 0);
      }
    });
  });
  const float iz = 1.0f / z;
#pragma unroll
  for (int d = 0; d < DP; ++d) score[d] = fmaf(P[d / 4][d % 4], iz, -x[d] * vec[RSF + d]);
}
// Per-component scales (WsLayout::gmm_lds == 4): four contractions, two instruction streams (layout: sdeh_common.hpp)
//   logits:  l_k = cc_k - sum_d x_d^2 / (2 sigma_kd^2) + sum_d x_d mu_kd / sigma_kd^2      (B = x_d^2, then B = x_d: 2 x 500 instructions)
//   score:   P_d = sum_k e_k mu_kd / sigma_kd^2,  Q_d = sum_k e_k / sigma_kd^2,  score_d = (P_d - x_d Q_d) / z      (2 x 520)
template <int DP>
__device__ __forceinline__ void gmm_mm_general(const float* __restrict__ ws, const float* __restrict__ lds, const WsLayout& L,
                                               const float (&x)[DP], float (&score)[DP]) {
  constexpr int K4 = SDEH_MM_K / 4, D4 = (DP + 3) / 4;
  constexpr int N1 = 2 * DP * K4, N2 = 2 * SDEH_MM_K * D4;
  constexpr int Q1 = (N1 + 63) / 64, Q2 = (N2 + 63) / 64;
  const int lane = threadIdx.x & 63;
  const mm4* __restrict__ a1 = reinterpret_cast<const mm4*>(lds + L.gmm_mm1) + lane;
  const mm4* __restrict__ a2 = reinterpret_cast<const mm4*>(lds + L.gmm_mm2) + lane;
  cfp cc = as_const(ws + L.gmm_cc);
  mm4 lg[K4];
#pragma unroll
  for (int g = 0; g < K4; ++g) lg[g] = mm4{cc[4 * g], cc[4 * g + 1], cc[4 * g + 2], cc[4 * g + 3]};
  static_for<Q1>([&](auto Qc) {
    constexpr int q = decltype(Qc)::value;
    const mm4 a = a1[q * 64];
    static_for<64>([&](auto Nc) {
      constexpr int n = 64 * q + decltype(Nc)::value;
      if constexpr (n < N1) {  // n = (2 d + t) K4 + g;  t = 0: A = -1 / (2 sigma^2), B = x_d^2;  t = 1: A = mu / sigma^2, B = x_d
        constexpr int d = n / (2 * K4), t = (n / K4) % 2, g = n % K4, e = (n / 16) % 4, b = n % 16;
        lg[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], t == 0 ? x[d] * x[d] : x[d], lg[g], 4, b, 0);
      }
    });
  });
  float m = lg[0][0];
#pragma unroll
  for (int g = 0; g < K4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) m = fmaxf(m, lg[g][i]);
  float z = 0.0f;
#pragma unroll
  for (int g = 0; g < K4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float e = __expf(lg[g][i] - m);
      lg[g][i] = e;
      z += e;
    }
  mm4 P[D4], Q[D4];
#pragma unroll
  for (int g = 0; g < D4; ++g) P[g] = Q[g] = mm4{0.0f, 0.0f, 0.0f, 0.0f};
  static_for<Q2>([&](auto Qc) {
    constexpr int q = decltype(Qc)::value;
    const mm4 a = a2[q * 64];
    static_for<64>([&](auto Nc) {
      constexpr int n = 64 * q + decltype(Nc)::value;
      if constexpr (n < N2) {  // n = (2 k + t) D4 + g;  t = 0: mu / sigma^2 -> P,  t = 1: 1 / sigma^2 -> Q
        constexpr int k = n / (2 * D4), t = (n / D4) % 2, g = n % D4, e = (n / 16) % 4, b = n % 16;
        if constexpr (t == 0) P[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], lg[k / 4][k % 4], P[g], 4, b, 0);
        else Q[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], lg[k / 4][k % 4], Q[g], 4, b, 0);
      }
    });
  });
  const float iz = 1.0f / z;
#pragma unroll
  for (int d = 0; d < DP; ++d) score[d] = (P[d / 4][d % 4] - x[d] * Q[d / 4][d % 4]) * iz;
}
// which instantiations carry the matrix-pipe mixture (the launcher asks the same question: ws_mm_compiled)
// 0: none, 1: the shared-scale form, 2: + the per-component-scale form (run-time table form only)
template <int DP, bool PAD, int GMMV, int GNV>
constexpr int gmm_mm_compiled() { return ((GMMV == 2 || GMMV < 0) && GNV <= 0 && DP > 8 && !PAD) ? (GMMV < 0 ? 2 : 1) : 0; }

template <int DP, int NV, bool SG = false>
__device__ __forceinline__ float ws_target_logp(const DensArgs& D, const float* ws, const float* lds, const WsLayout& L,
                                                int gmmv, int dreal, const float (&x)[DP]) {
  float dummy[DP];
  switch (D.kind) {
    case SDEH_DENS_GMM:
      if constexpr (SG) {
        if (gmmv == 2 || L.gmm_lds == 3) {  // (gmm_lds == 3: no tables in LDS -- only instantiations with this path are launched)
          if constexpr (gmm_use_sgpr<DP, true, NV>()) return gmm_online_s<DP, true, false, NV>(ws, L, x, dummy) + D.lnc;
        } else {
          if constexpr (gmm_use_sgpr<DP, false, DP>()) return gmm_online_s<DP, false, false, DP>(ws, L, x, dummy) + D.lnc;
        }
      }
      return (gmmv == 2 ? gmm_online<DP, true, false, NV>(lds, L, D.n_comp, x, dummy)
                        : gmm_online<DP, false, false, DP>(lds, L, D.n_comp, x, dummy)) + D.lnc;
    case SDEH_DENS_DIAG_GAUSS: return dgauss_logp<DP>(ws + L.dg[0], x) + D.lnc;
    case SDEH_DENS_MULTI_WELL: return mwell_logp<DP>(D, dreal, x);
    case SDEH_DENS_FUNNEL: return funnel_logp<DP>(D, dreal, x);
    default: return 0.0f;
  }
}

template <int DP, int NV, bool KS = false, bool SG = false, int MM = 0>
__device__ __forceinline__ void ws_target_score(const DensArgs& D, const float* ws, const float* lds, const WsLayout& L,
                                                int gmmv, int dreal, const float (&x)[DP], float (&s)[DP]) {
  switch (D.kind) {
    case SDEH_DENS_GMM:
      if constexpr (MM >= 1 && SG && !KS && NV == DP) {
        if (L.gmm_lds == 3) { gmm_mm<DP>(ws, lds, L, x, s); break; }
      }
      if constexpr (MM >= 2 && SG && !KS && NV == DP) {
        if (L.gmm_lds == 4) { gmm_mm_general<DP>(ws, lds, L, x, s); break; }
      }
      if constexpr (SG && !KS) {
        if (gmmv == 2 || L.gmm_lds == 3) {
          if constexpr (gmm_use_sgpr<DP, true, NV>()) { (void)gmm_online_s<DP, true, true, NV>(ws, L, x, s); break; }
        } else {
          if constexpr (gmm_use_sgpr<DP, false, DP>()) { (void)gmm_online_s<DP, false, true, DP>(ws, L, x, s); break; }
        }
      }
      if (gmmv == 2) (void)gmm_online<DP, true, true, NV, KS>(lds, L, D.n_comp, x, s);
      else (void)gmm_online<DP, false, true, DP, KS>(lds, L, D.n_comp, x, s);
      break;
    case SDEH_DENS_DIAG_GAUSS: dgauss_score<DP>(ws + L.dg[0], x, s); break;
    case SDEH_DENS_MULTI_WELL: mwell_score<DP>(D, dreal, x, s); break;
    case SDEH_DENS_FUNNEL: funnel_score<DP>(D, dreal, x, s); break;
    default:
#pragma unroll
      for (int d = 0; d < DP; ++d) s[d] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// M wave: FourierMLP forward, x read from / output written to the group's exchange buffer [coordinate][64].
//
// The wave's two 32-trajectory column tiles A and B are software-pipelined against each other: while the MFMAs of
// one tile occupy the matrix pipe, the activation of the other tile's previous output issues on the VALU in their
// shadow (one element per k-step, order pinned by scheduling fences):
//   L0(A) | L0(B) + act(A0) | L1(A) + act(B0) | L1(B) + act(A1) | ... | out(A) + act(B_last) | out(B) + publish(A)
// ---------------------------------------------------------------------------------------------------------
// Pre-activation plane store of the training forward: accumulator q of lane (j, h) is channel 32 ot + rho(q, h) of the group's row j
// (+ 32 for column tile B); plane layout [(Lh+1), C, N] coordinate-major, n = step * B + global row.
struct ZStore {
  float* base;       // zt_out + step * B + first row of the group, or null
  long long N;       // T * B
  int rows;          // live rows of the group (<= 64)
};
__device__ __forceinline__ void ws_store_z(const ZStore& Z, int layer, int C, int ot, int col_tile, int lane, const f32x16& v) {
  const int h = lane >> 5, j = (lane & 31) + 32 * col_tile;
  if (j < Z.rows) {
    float* __restrict__ p = Z.base + ((long long)layer * C + 32 * ot) * Z.N + j;
#pragma unroll
    for (int q = 0; q < 16; ++q) p[(long long)rho(q, h) * Z.N] = v[q];
  }
}

// Pre-activation RECORD of the training forward (PLANES == 3, round 5: the fused backward READS the pre-activations instead of
// re-evaluating the network -- the reference's autograd keeps them as well, losses/oc.py:232-256 -> models/mlp.py:114-122):
//     zrec[step][tile of 32 trajectories]{ [layer 0 .. Lh][channel quad cq = 0 .. 15][trajectory j = 0 .. 31][4 channels] ;
//                                          [coordinate tile 0 .. OTD - 1][coordinate quad 0 .. 7][trajectory j][4 coordinates] }
// (the second part: the raw network output, before the clamp -- the backward's clamp mask and, through time, its control)
// Accumulator registers 4 g .. 4 g + 3 of lane (j, h), row tile ot are channels 32 ot + 8 g + 4 h .. + 3 of trajectory j, i.e. quad
// cq = 8 ot + 2 g + h: one 16-byte store per lane, 1 KB contiguous per instruction, and the trajectory-split backward (lane (j, h) of the
// wave that owns the tile) loads exactly what was stored.  Non-temporal: every word is written once and read once, by another kernel.
typedef float f32x4z __attribute__((ext_vector_type(4)));
struct ZRec {
  float* base;   // record of (step, the group's first tile), or null
  int tiles;     // live 32-trajectory tiles of the group (0 .. 2): tiles beyond the batch are not stored
  int lh1;       // layers per tile (Lh + 1); the network output's tiles follow them ("layer" lh1, ot = coordinate tile)
  int stride;    // floats per tile: lh1 * 2048 + 1024 * coordinate tiles
};
__device__ __forceinline__ void ws_store_zrec(const ZRec& Z, int layer, int ot, int col_tile, int lane, const f32x16& v) {
#ifdef SDEH_ZREC_SKIP  // (measurement builds, tools/zrec_fwd_ablation.sh)
  return;
#endif
  if (col_tile < Z.tiles) {
    float* __restrict__ p = Z.base + (long long)col_tile * Z.stride + layer * 2048 + ot * 1024 + (lane >> 5) * 128 + (lane & 31) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#ifdef SDEH_ZREC_NO_NT
      *reinterpret_cast<f32x4z*>(p + g * 256) = f32x4z{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
#else
      __builtin_nontemporal_store(f32x4z{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}, reinterpret_cast<f32x4z*>(p + g * 256));
#endif
    }
  }
}
// ZS (template parameter of the network passes): 0 = nothing stored, 1 = coordinate-major planes (ZStore, round 1), 2 = the record
template <int ZS>
__device__ __forceinline__ void ws_keep_z(const ZStore& Z, const ZRec& Zr, int layer, int C, int ot, int col_tile, int lane, const f32x16& v) {
  if constexpr (ZS == 1) ws_store_z(Z, layer, C, ot, col_tile, lane, v);
  if constexpr (ZS == 2) ws_store_zrec(Zr, layer, ot, col_tile, lane, v);
}

__device__ __forceinline__ float act_apply(float v, int act) {
  return act == SDEH_ACT_GELU_ERF ? act_gelu(v) : (act == SDEH_ACT_SILU ? act_silu(v) : act_relu(v));
}

// Two activations at once: the polynomial and the final product as v_pk_fma_f32 on the element pair (IEEE fma per component: the
// results are those of act_gelu bit for bit).  Written in assembly -- hipcc unpacks packed fp32 ops whose constant operand is a splat
// back into two v_fma_f32.  The coefficients sit in SGPR pairs (one scalar source per instruction; the leading one in a VGPR pair).
// 13 instructions per pair instead of 20.
__device__ __forceinline__ f2 pk_fma_sc(f2 a, f2 b, unsigned long long c2) {
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c2));
  return r;
}
__device__ __forceinline__ unsigned long long splat_bits(float c) {
  const unsigned long long b = __float_as_uint(c);
  return (b << 32) | b;
}
__device__ __forceinline__ f2 act_gelu2(f2 v) {
#ifdef SDEH_NO_PK_GELU
  return f2{act_gelu(v.x), act_gelu(v.y)};
#else
  const f2 t = f2{__builtin_amdgcn_fmed3f(fabsf(v.x), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(fabsf(v.y), 0.0f, 6.0f)};
  f2 q = pk_fma_sc(splat(3.3092907814e-05f), t, splat_bits(-7.6922050644e-04f));
  q = pk_fma_sc(q, t, splat_bits(8.0807191412e-03f));
  q = pk_fma_sc(q, t, splat_bits(-5.3412108121e-02f));
  q = pk_fma_sc(q, t, splat_bits(-4.5877097054e-01f));
  q = pk_fma_sc(q, t, splat_bits(-1.1512017029e+00f));
  q = pk_fma_sc(q, t, splat_bits(-9.9999306093e-01f));
  const f2 e = f2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  // The hazard recogniser does not look into assembly: the statements that READ v (written by MFMAs) take t as an extra input, so
  // they are ordered behind the v_med3 the compiler emitted (with its wait states) for the same registers; and the value handed back
  // to the MFMAs is produced by a compiler-visible instruction.
  f2 relu;
  asm("v_max_f32 %0, 0, %1" : "=v"(relu.x) : "v"(v.x), "v"(t.x));
  asm("v_max_f32 %0, 0, %1" : "=v"(relu.y) : "v"(v.y), "v"(t.y));
  return pk_fma(-t, e, relu);
#endif
}

// one accumulator tile in place (pairs of elements for the GELU)
template <int ACT>
__device__ __forceinline__ void act_tile(f32x16& v) {
  if constexpr (ACT == SDEH_ACT_GELU_ERF) {
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      const f2 r = act_gelu2(f2{v[q], v[q + 1]});
      v[q] = r.x;
      v[q + 1] = r.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = act_ct<ACT>(v[q]);
  }
}

// out[OTO] += W[s][ot] * in-operand(s) for NS k-steps, while activating the NE elements of `side` in place.
// The A operands (packed weights in LDS) are fetched PF k-steps ahead through a rotating register window: the LDS
// queue is shared with the V waves' mixture-table reads, and an un-prefetched weight read in front of every MFMA
// pair left the matrix pipe idle for the whole LDS latency (measured: 25 us per step for 15 us of MFMA work).
template <int NS, int OTO, int NSIDE, class IN>
__device__ __forceinline__ void mfma_stage(const float* __restrict__ w, IN&& in, f32x16 (&out)[OTO],
                                           f32x16 (&side)[NSIDE], bool do_side, int act) {
  constexpr int NE = NSIDE * 16;
  constexpr int PF = NS < 6 ? NS : 6;
  float wq[PF][OTO];
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int ot = 0; ot < OTO; ++ot) wq[s][ot] = w[(s * OTO + ot) * 64];
  SDEH_FENCE();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float a[OTO];
#pragma unroll
    for (int ot = 0; ot < OTO; ++ot) a[ot] = wq[s % PF][ot];
    if (s + PF < NS) {
#pragma unroll
      for (int ot = 0; ot < OTO; ++ot) wq[s % PF][ot] = w[((s + PF) * OTO + ot) * 64];
    }
    const float b = in(s);
#pragma unroll
    for (int ot = 0; ot < OTO; ++ot) out[ot] = SDEH_MFMA(a[ot], b, out[ot]);
    if (do_side) {
      // the NE/2 element pairs (2i, 2i+1) are spread evenly over the NS k-steps
#pragma unroll
      for (int i = s * (NE / 2) / NS; i < (s + 1) * (NE / 2) / NS; ++i) {
        const int e = 2 * i;
        if (act == SDEH_ACT_GELU_ERF) {
          const f2 r = act_gelu2(f2{side[e / 16][e % 16], side[(e + 1) / 16][(e + 1) % 16]});
          side[e / 16][e % 16] = r.x;
          side[(e + 1) / 16][(e + 1) % 16] = r.y;
        } else {
          side[e / 16][e % 16] = act_apply(side[e / 16][e % 16], act);
          side[(e + 1) / 16][(e + 1) % 16] = act_apply(side[(e + 1) / 16][(e + 1) % 16], act);
        }
      }
    }
    SDEH_FENCE();
  }
}

// Small state dimensions (d <= 4): the out layer is a 32-row matrix tile around d live rows -- 64 (32) MFMAs per step and group for
// 2 x 64 useful weights at d = 2.  With `abuf` != nullptr the M wave stops after the last hidden layer and publishes its activated
// outputs [channel][trajectory] instead; the V wave forms the d dot products on the vector pipe (d x 64 FMAs per trajectory).
template <int C>
__device__ __forceinline__ void ws_publish_act(float* __restrict__ abuf, const f32x16 (&a)[C / 32], int col, int h) {
#pragma unroll
  for (int ot = 0; ot < C / 32; ++ot)
#pragma unroll
    for (int q = 0; q < 16; ++q) abuf[(32 * ot + rho(q, h)) * 64 + col] = a[ot][q];
}

// ---------------------------------------------------------------------------------------------------------
// The out layer on v_mfma_f32_4x4x1_16b_f32 (round 5; WsLayout::w_out4).  A [32 x 64] x [64 x 32] tile product spends 64 cycles per
// k-pair on 32 output rows of which d = 50 leaves 14 empty (d = 10: 22 of 32).  Here the output rows come in groups of FOUR: the
// instruction multiplies, block by block (4 lanes = 4 trajectories), a 4-vector of weights with ONE value per lane -- the activation the
// lane already holds in accumulator register (ot, q) -- and adds it to that lane's four partial sums: 8 cycles for 4 rows x 64 lanes,
// the same 64 FLOP per cycle without empty rows: 33 ceil(d / 4) instructions per column tile (d = 50: 429 x 8 = 3.4 k cycles against
// 64 x 64 = 4.1 k; d = 10: 99 x 8 = 0.8 k against 2.0 k; compiled for d <= 16, see ws_out4_compiled).  Lanes j (h = 0) and 32 + j (h = 1) hold DIFFERENT channels of the same
// trajectory, so their weights differ: CBSZ = 3 broadcasts block ABID of each half to the 8 blocks of that half -- one operand register
// carries 8 instructions, 2 x 4 weights each -- and the two halves' partial sums meet in one v_permlane32_swap + add per PAIR of row
// groups (after it the lower half holds the sums of group 2 p, the upper half those of group 2 p + 1: every lane publishes as many
// values as before).  The bias is a 33rd slot against B = 1.  Activation of the other column tile rides in the shadow as in mfma_stage.
template <int DP>
constexpr bool ws_out4_compiled() { return DP > 4 && DP <= 16; }
// d = 50 leaves 14 of 64 rows empty as well (429 x 8 = 3.4 k against 4.1 k cycles per column tile), measured and NOT compiled: with all 13
// accumulator quads alive the M wave spills (2.178 -> 2.175 ms against 2.131 ms without the code); accumulated in two passes of 7 + 6 groups
// (the pass structure below, kOut4Pass) it still spills 232 B and the 26 swaps + adds per tile eat the rest: 2.158 -> 2.153 ms.
template <int G4, int NSIDE>
__device__ __forceinline__ void ws_out4_stage(const mm4* __restrict__ a4, const f32x16 (&in)[2], mm4 (&u)[G4], f32x16 (&side)[NSIDE],
                                              bool do_side, int act) {
  constexpr int N = 33 * G4, NQ = (N + 31) / 32;  // instructions, ds_read_b128 (32 instructions each)
  constexpr int NE2 = NSIDE * 8;                   // element pairs of the side tile, spread over the NQ operand reads
  mm4 aq = a4[0];
  static_for<NQ>([&](auto Qc) {
    constexpr int q4 = decltype(Qc)::value;
    const mm4 a = aq;
    if constexpr (q4 + 1 < NQ) aq = a4[(q4 + 1) * 64];
    static_for<32>([&](auto Nc) {
      constexpr int n = 32 * q4 + decltype(Nc)::value;
      if constexpr (n < N) {
        constexpr int slot = n / G4, g = n % G4, e = (n / 8) % 4, b = n % 8;
        if constexpr (slot < 32) u[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], in[slot / 16][slot % 16], u[g], 3, b, 0);
        else u[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], 1.0f, u[g], 3, b, 0);
      }
    });
    if (do_side) {
#pragma unroll
      for (int i = q4 * NE2 / NQ; i < (q4 + 1) * NE2 / NQ; ++i) {
        const int e = 2 * i;
        if (act == SDEH_ACT_GELU_ERF) {
          const f2 r = act_gelu2(f2{side[e / 16][e % 16], side[(e + 1) / 16][(e + 1) % 16]});
          side[e / 16][e % 16] = r.x;
          side[(e + 1) / 16][(e + 1) % 16] = r.y;
        } else {
          side[e / 16][e % 16] = act_apply(side[e / 16][e % 16], act);
          side[(e + 1) / 16][(e + 1) % 16] = act_apply(side[(e + 1) / 16][(e + 1) % 16], act);
        }
      }
    }
    SDEH_FENCE();
  });
}
// the two halves' partial sums -> the exchange buffer [coordinate][trajectory]; `col` = the lane's trajectory column
template <int DP, int G4>
__device__ __forceinline__ void ws_out4_publish(float* __restrict__ xbuf, mm4 (&u)[G4], int col, int h, int g_first) {
#pragma unroll
  for (int p = 0; p < (G4 + 1) / 2; ++p) {
    const int g0 = 2 * p, g1 = 2 * p + 1 < G4 ? 2 * p + 1 : 2 * p;  // (an unpaired last group meets itself: both halves get its sums)
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(u[g0][i]), __float_as_uint(u[g1][i]), false, false);
      // r[0] = (lower half of g0, lower half of g1), r[1] = (upper half of g0, upper half of g1)
      s[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const int g = g_first + (h ? g1 : g0);
    if (g0 != g1 || h == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * g + i < xrows<DP>()) xbuf[(4 * g + i) * 64 + col] = s[i];
    }
  }
}

template <int DP, int C, int ZS>
__device__ __forceinline__ void ws_mlp(const float* __restrict__ lds, float* __restrict__ xbuf, const WsLayout& L,
                                       int act, const f32x16 (&emb)[C / 32], int lane, const ZStore& Z, const ZRec& Zr, float* __restrict__ abuf = nullptr) {
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5, j = lane & 31;
  f32x16 curA[OT], curB[OT], nxtA[OT], nxtB[OT];
  {  // input layer: e = input_embed(x) + timestep_embed(t)
    float xa[R], xb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* row = xbuf + (h ? mdim(r, 1) : mdim(r, 0)) * 64 + j;
      xa[r] = row[0];
      xb[r] = row[32];
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) curA[ot] = curB[ot] = emb[ot];
    const float* w = lds + L.w_in + lane;
    mfma_stage<R, OT, OT>(w, [&](int s) { return xa[s]; }, curA, curB, false, act);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) ws_keep_z<ZS>(Z, Zr, 0, C, ot, 0, lane, curA[ot]);
    mfma_stage<R, OT, OT>(w, [&](int s) { return xb[s]; }, curB, curA, true, act);  // + act(A0)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) ws_keep_z<ZS>(Z, Zr, 0, C, ot, 1, lane, curB[ot]);
  }
  // invariant at the top of each layer: curA activated, curB not yet
  for (int l = 0; l < L.n_hidden; ++l) {
    const float* bias = lds + L.b_hid + l * C;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) nxtA[ot] = nxtB[ot] = load16(bias + (ot * 2 + h) * 16);
    const float* w = lds + L.w_hid + l * L.w_hid_stride + lane;
    mfma_stage<C / 2, OT, OT>(w, [&](int s) { return curA[s / 16][s % 16]; }, nxtA, curB, true, act);  // + act(B)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) ws_keep_z<ZS>(Z, Zr, l + 1, C, ot, 0, lane, nxtA[ot]);
    mfma_stage<C / 2, OT, OT>(w, [&](int s) { return curB[s / 16][s % 16]; }, nxtB, nxtA, true, act);  // + act(A')
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) { ws_keep_z<ZS>(Z, Zr, l + 1, C, ot, 1, lane, nxtB[ot]); curA[ot] = nxtA[ot]; curB[ot] = nxtB[ot]; }
  }
  if constexpr (DP <= 4) {
    if (abuf != nullptr) {  // out layer on the V wave: activate tile B (tile A is) and publish both
      SDEH_ACT_SWITCH(act, ACTC,
        _Pragma("unroll") for (int ot = 0; ot < OT; ++ot) act_tile<ACTC>(curB[ot]););
      ws_publish_act<C>(abuf, curA, j, h);
      ws_publish_act<C>(abuf, curB, j + 32, h);
      return;
    }
  }
  if constexpr (ZS == 0 && C == 64 && ws_out4_compiled<DP>()) {
    if (L.w_out4 >= 0) {  // out layer on 4 x 4 x 1 matrix instructions (no empty rows); tile B is activated in tile A's shadow
      const mm4* a4 = reinterpret_cast<const mm4*>(lds + L.w_out4) + lane;
      f32x16 none[1];
      // (one pass' partial sums alive at a time: they are published before the next pass starts; tile B is activated in the shadow of
      // tile A's first pass)
      static_for<2 * out4_passes(DP)>([&](auto Pc) {
        constexpr int tile = decltype(Pc)::value / out4_passes(DP), p = decltype(Pc)::value % out4_passes(DP);
        constexpr int GP = out4_pass_groups(DP, p);
        mm4 u4[GP];
#pragma unroll
        for (int g = 0; g < GP; ++g) u4[g] = mm4{0.0f, 0.0f, 0.0f, 0.0f};
        const mm4* ap = a4 + out4_pass_offset(DP, p) / 4;
        if constexpr (tile == 0 && p == 0) ws_out4_stage<GP, OT>(ap, curA, u4, curB, true, act);
        else if constexpr (tile == 0) ws_out4_stage<GP, 1>(ap, curA, u4, none, false, act);
        else ws_out4_stage<GP, 1>(ap, curB, u4, none, false, act);
        ws_out4_publish<DP, GP>(xbuf, u4, j + 32 * tile, h, out4_pass_first(DP, p));
        SDEH_FENCE();
      });
      return;
    }
  }
  {  // out_layer(act(e)); the A tile's result is published while the B tile's MFMAs run
    f32x16 uA[OTD], uB[OTD];
#pragma unroll
    for (int t = 0; t < OTD; ++t) uA[t] = uB[t] = load16(lds + L.b_out + (t * 2 + h) * 16);
    const float* w = lds + L.w_out + lane;
    mfma_stage<C / 2, OTD, OT>(w, [&](int s) { return curA[s / 16][s % 16]; }, uA, curB, true, act);  // + act(B)
    f32x16 none[1];
    mfma_stage<C / 2, OTD, 1>(w, [&](int s) { return curB[s / 16][s % 16]; }, uB, none, false, act);
    if constexpr (ZS == 2) {  // the raw network output joins the record (coordinates >= d: zero weights and biases, exact zeros)
#pragma unroll
      for (int t = 0; t < OTD; ++t) { ws_store_zrec(Zr, Zr.lh1, t, 0, lane, uA[t]); ws_store_zrec(Zr, Zr.lh1, t, 1, lane, uB[t]); }
    }
    // accumulator register r of lane (j,h) is coordinate mdim(r,h) of trajectory j (tile A) / 32+j (tile B)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float* row = xbuf + (h ? mdim(r, 1) : mdim(r, 0)) * 64 + j;
      row[0] = uA[r / 16][r % 16];
      row[32] = uB[r / 16][r % 16];
    }
  }
}

// Single-tile variant for small batches (TrajArgs::half): the group is 32 trajectories = one MFMA column tile, so the chain of
// dependent layers is half as long (the launch is latency-bound: a handful of wavefronts on 1024 SIMDs).
template <int DP, int C, int ZS>
__device__ __forceinline__ void ws_mlp_half(const float* __restrict__ lds, float* __restrict__ xbuf, const WsLayout& L,
                                            int act, const f32x16 (&emb)[C / 32], int lane, const ZStore& Z, const ZRec& Zr, float* __restrict__ abuf = nullptr) {
  constexpr int OT = C / 32, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5, j = lane & 31;
  f32x16 cur[OT], nxt[OT], none[1];
  auto activate_all = [&]() {
    SDEH_ACT_SWITCH(act, ACTC,
      _Pragma("unroll") for (int ot = 0; ot < OT; ++ot) act_tile<ACTC>(cur[ot]););
  };
  {
    float xa[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xa[r] = xbuf[(h ? mdim(r, 1) : mdim(r, 0)) * 64 + j];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) cur[ot] = emb[ot];
    mfma_stage<R, OT, 1>(lds + L.w_in + lane, [&](int s) { return xa[s]; }, cur, none, false, act);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) ws_keep_z<ZS>(Z, Zr, 0, C, ot, 0, lane, cur[ot]);
  }
  for (int l = 0; l < L.n_hidden; ++l) {
    activate_all();
    const float* bias = lds + L.b_hid + l * C;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) nxt[ot] = load16(bias + (ot * 2 + h) * 16);
    mfma_stage<C / 2, OT, 1>(lds + L.w_hid + l * L.w_hid_stride + lane, [&](int s) { return cur[s / 16][s % 16]; }, nxt, none,
                             false, act);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) { cur[ot] = nxt[ot]; ws_keep_z<ZS>(Z, Zr, l + 1, C, ot, 0, lane, cur[ot]); }
  }
  activate_all();
  if constexpr (DP <= 4) {
    if (abuf != nullptr) {  // out layer on the V wave
      ws_publish_act<C>(abuf, cur, j, h);
      return;
    }
  }
  if constexpr (ZS == 0 && C == 64 && ws_out4_compiled<DP>()) {
    if (L.w_out4 >= 0) {
      const mm4* a4 = reinterpret_cast<const mm4*>(lds + L.w_out4) + lane;
      static_for<out4_passes(DP)>([&](auto Pc) {
        constexpr int p = decltype(Pc)::value, GP = out4_pass_groups(DP, p);
        mm4 u4[GP];
#pragma unroll
        for (int g = 0; g < GP; ++g) u4[g] = mm4{0.0f, 0.0f, 0.0f, 0.0f};
        ws_out4_stage<GP, 1>(a4 + out4_pass_offset(DP, p) / 4, cur, u4, none, false, act);
        ws_out4_publish<DP, GP>(xbuf, u4, j, h, out4_pass_first(DP, p));
        SDEH_FENCE();
      });
      return;
    }
  }
  f32x16 u[OTD];
#pragma unroll
  for (int t = 0; t < OTD; ++t) u[t] = load16(lds + L.b_out + (t * 2 + h) * 16);
  mfma_stage<C / 2, OTD, 1>(lds + L.w_out + lane, [&](int s) { return cur[s / 16][s % 16]; }, u, none, false, act);
  if constexpr (ZS == 2) {
#pragma unroll
    for (int t = 0; t < OTD; ++t) ws_store_zrec(Zr, Zr.lh1, t, 0, lane, u[t]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) xbuf[(h ? mdim(r, 1) : mdim(r, 0)) * 64 + j] = u[r / 16][r % 16];
}

// One output tile: out += W[s][tile] * in(s) for NS k-steps; `w` already points at the tile's column of the packed weights,
// WOT = number of output tiles packed per k-step (the stride between k-steps).  Same operand prefetch as mfma_stage.
template <int NS, int WOT, class IN>
__device__ __forceinline__ void mfma_stage_one(const float* __restrict__ w, IN&& in, f32x16& out) {
  constexpr int PF = NS < 6 ? NS : 6;
  float wq[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) wq[s] = w[s * WOT * 64];
  SDEH_FENCE();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float a = wq[s % PF];
    if (s + PF < NS) wq[s % PF] = w[(s + PF) * WOT * 64];
    out = SDEH_MFMA(a, in(s), out);
    SDEH_FENCE();
  }
}

// Pair mode for the smallest batches (TrajArgs::half == 2, C = 64): TWO M waves serve one group of 32 trajectories, wave `mw`
// owning output-channel tile mw of every layer (half the MFMAs and half the activations of the dependent chain per wave).  Each
// layer's input is both tiles, so after activating its tile a wave parks it in `abuf` (double-buffered by layer parity), the
// workgroup barrier makes both tiles visible, and the other tile is read back in the same (register, lane) arrangement.  The
// k-steps of the hidden layers run in the order of ws_mlp_half (tile 0's channels, then tile 1's): their pre-activations are bit-identical
// to it (the fused backward re-evaluates them); the output layer is the sum of two per-wave partial sums (see below).
template <int DP, int C, int ACT, int ZS>
__device__ __forceinline__ void ws_mlp_pair(const float* __restrict__ lds, float* __restrict__ xbuf, float* __restrict__ abuf,
                                            float* __restrict__ xbuf2, const WsLayout& L, const f32x16& emb_mine, int lane, int mw,
                                            int& parity, const ZStore& Z, const ZRec& Zr) {
  static_assert(C == 64, "pair mode splits the two 32-channel tiles of a 64-channel network");
  constexpr int OT = 2, OTD = row_tiles(DP), R = mregs(DP);
  const int h = lane >> 5, j = lane & 31;
  f32x16 mine = emb_mine, other;
  {
    float xa[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xa[r] = xbuf[(h ? mdim(r, 1) : mdim(r, 0)) * 64 + j];
    mfma_stage_one<R, OT>(lds + L.w_in + mw * 64 + lane, [&](int s) { return xa[s]; }, mine);
    ws_keep_z<ZS>(Z, Zr, 0, C, mw, 0, lane, mine);
  }
  auto exchange = [&]() {  // mine <- act(mine); other <- the partner's activated tile
    act_tile<ACT>(mine);
    float* __restrict__ mb = abuf + ((parity * 2 + mw) * 16) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) mb[q * 64] = mine[q];
    ws_barrier();
    const float* __restrict__ ob = abuf + ((parity * 2 + (1 - mw)) * 16) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) other[q] = ob[q * 64];
    parity ^= 1;
  };
  // out += W[:, tile] . [a(tile 0 channels); a(tile 1 channels)], k-steps in channel order
  auto layer = [&](const float* __restrict__ w, auto wot, f32x16& out) {
    constexpr int WOT = decltype(wot)::value;
    if (mw == 0) {
      mfma_stage_one<16, WOT>(w, [&](int s) { return mine[s]; }, out);
      mfma_stage_one<16, WOT>(w + 16 * WOT * 64, [&](int s) { return other[s]; }, out);
    } else {
      mfma_stage_one<16, WOT>(w, [&](int s) { return other[s]; }, out);
      mfma_stage_one<16, WOT>(w + 16 * WOT * 64, [&](int s) { return mine[s]; }, out);
    }
  };
  for (int l = 0; l < L.n_hidden; ++l) {
    exchange();
    f32x16 nxt = load16(lds + L.b_hid + l * C + (mw * 2 + h) * 16);
    layer(lds + L.w_hid + l * L.w_hid_stride + mw * 64 + lane, std::integral_constant<int, OT>{}, nxt);
    mine = nxt;
    ws_keep_z<ZS>(Z, Zr, l + 1, C, mw, 0, lane, mine);
  }
  if constexpr (OTD == 1) {
    // d <= 32, one output tile: no further exchange -- each wave contracts over ITS OWN 32 channels (16 MFMAs; wave 1 used to idle
    // through wave 0's 32) and publishes the partial sum in a buffer of its own; the V wave adds the two (bias in wave 0's).
    act_tile<ACT>(mine);
    float* __restrict__ ob = mw == 0 ? xbuf : xbuf2;
    f32x16 u;
    if (mw == 0) u = load16(lds + L.b_out + h * 16);
    else {
#pragma unroll
      for (int q = 0; q < 16; ++q) u[q] = 0.0f;
    }
    mfma_stage_one<16, 1>(lds + L.w_out + lane + mw * (16 * 64), [&](int s) { return mine[s]; }, u);
#pragma unroll
    for (int r = 0; r < R; ++r) ob[(h ? mdim(r, 1) : mdim(r, 0)) * 64 + j] = u[r];
  } else {  // two output tiles: one each, over all 64 channels (one more exchange; measured equal to partial sums at d = 50)
    exchange();
    f32x16 u = load16(lds + L.b_out + (mw * 2 + h) * 16);
    layer(lds + L.w_out + mw * 64 + lane, std::integral_constant<int, OTD>{}, u);
    if constexpr (ZS == 2) ws_store_zrec(Zr, Zr.lh1, mw, 0, lane, u);  // (the whole sum of coordinate tile mw: it joins the record here)
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r / 16 == mw) xbuf[(h ? mdim(r, 1) : mdim(r, 0)) * 64 + j] = u[r % 16];
  }
}

// Quad mode (TrajArgs::half == 3; C = 64, d <= 32, activations without a kink): FOUR M waves serve one group of 32 trajectories.  Wave mw
// owns output channels 16 mw .. 16 mw + 15 of every layer as two 16 x 16 accumulator tiles (v_mfma_f32_16x16x4_f32: trajectories 0..15 and
// 16..31 -- two independent chains): a hidden layer is 2 x 16 instructions of 32 cycles per wave instead of the pair mode's 32 of 64.
//   lane (n, kk) = (lane & 15, lane >> 4);  accumulator register q <-> row 4 kk + q of the tile, column n
//   A operand: the packed weight image of the 32 x 32 x 2 kernels serves as it is -- instruction t takes the packed k-steps 2t and 2t + 1:
//              lane (n, kk) reads entry (k-step 2t + (kk >> 1), lane half kk & 1) of its row, i.e. k = mdim(2t + (kk >> 1), kk & 1)
//   B operand: a layer's input in natural channel order, plane[k][32 trajectories] in LDS (row stride 48: conflict-free reads); every wave
//              writes its 16 channels, one barrier, every wave reads all 64 (double-buffered by layer parity)
//   output layer: as in the pair mode with one tile -- each wave contracts over its OWN 16 channels straight from its registers (register q
//              of lane (n, kk) is channel 16 mw + 4 kk + q: instruction q's four k) and publishes a partial sum; the V wave adds the four.
// The pre-activations are not bit for bit those of the 32 x 32 x 2 kernels (another instruction), which is why networks with ReLU -- whose
// fused backward re-evaluates them bitwise -- stay with the pair mode.
constexpr int kQuadPRS = 48;
typedef float f32x4q __attribute__((ext_vector_type(4)));
#define SDEH_MFMA16Q(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int ACT>
__device__ __forceinline__ void act_tile4(f32x4q& v) {
  if constexpr (ACT == SDEH_ACT_GELU_ERF) {
    const f2 a = act_gelu2(f2{v[0], v[1]}), b = act_gelu2(f2{v[2], v[3]});
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = act_ct<ACT>(v[q]);
  }
}

template <int DP, int C, int ACT, bool ZQ = false>
__device__ __forceinline__ void ws_mlp_quad(const float* __restrict__ lds, const float* __restrict__ xbuf, float* __restrict__ planes,
                                            float* __restrict__ pout, const WsLayout& L, const f32x4q& emb_mine, int lane, int mw,
                                            int& parity, const ZRec& Zr = ZRec{nullptr, 0, 0}) {
  static_assert(C == 64 && DP <= 32, "quad mode: 64 channels as four 16-row tiles, one 32-coordinate output tile");
  constexpr int OT = 2, R = mregs(DP), XR = xrows<DP>(), PRS = kQuadPRS;
  const int n = lane & 15, kk = lane >> 4;
  // this lane's entry of a packed k-step pair: k-step + (kk >> 1), lane half kk & 1, row 16 mw + n of the weight matrix
  const int aoff = (kk >> 1) * OT * 64 + (mw >> 1) * 64 + 32 * (kk & 1) + 16 * (mw & 1) + n;
  // ... and the channel that entry multiplies: mdim(2t + (kk >> 1), kk & 1) = mdim(2t, 0) + (kk >> 1) + 4 (kk & 1)
  const int koff = (kk >> 1) + 4 * (kk & 1);
  f32x4q acc[2] = {emb_mine, emb_mine};
  // the pre-activation record (ZRec): register q of lane (n, kk) is channel 16 mw + 4 kk + q, i.e. quad 4 mw + kk, of trajectory n / 16 + n
  auto keep_z = [&](int layer) {
    if constexpr (ZQ) {
      if (Zr.tiles > 0) {
        float* __restrict__ p = Zr.base + layer * 2048 + (4 * mw + kk) * 128 + n * 4;
        __builtin_nontemporal_store(acc[0], reinterpret_cast<f32x4q*>(p));
        __builtin_nontemporal_store(acc[1], reinterpret_cast<f32x4q*>(p + 64));
      }
    }
  };
  {  // input layer: k runs over the coordinates, B from the exchange buffer [coordinate][64]
    const float* __restrict__ w = lds + L.w_in + aoff;
#pragma unroll
    for (int t = 0; t < (R + 1) / 2; ++t) {
      const bool in_range = 2 * t + (kk >> 1) < R;  // (an odd R: the pair's second k-step does not exist)
      const float a = w[2 * t * OT * 64];
      const int c = mdim(2 * t, 0) + koff;
      const float b0 = in_range ? xbuf[c * 64 + n] : 0.0f, b1 = in_range ? xbuf[c * 64 + 16 + n] : 0.0f;
      acc[0] = SDEH_MFMA16Q(a, b0, acc[0]);
      acc[1] = SDEH_MFMA16Q(a, b1, acc[1]);
    }
    keep_z(0);
  }
  for (int l = 0; l < L.n_hidden; ++l) {
    act_tile4<ACT>(acc[0]);
    act_tile4<ACT>(acc[1]);
    float* __restrict__ pl = planes + parity * 64 * PRS;
    {
      float* __restrict__ p = pl + (16 * mw + 4 * kk) * PRS + n;
#pragma unroll
      for (int q = 0; q < 4; ++q) { p[q * PRS] = acc[0][q]; p[q * PRS + 16] = acc[1][q]; }
    }
    ws_barrier();
    parity ^= 1;
    const f32x4q bias = *reinterpret_cast<const f32x4q*>(lds + L.b_hid + l * C + ((mw >> 1) * 2 + (kk & 1)) * 16 + 4 * (2 * (mw & 1) + (kk >> 1)));
    acc[0] = bias;
    acc[1] = bias;
    const float* __restrict__ w = lds + L.w_hid + l * L.w_hid_stride + aoff;
    const float* __restrict__ bp = pl + koff * PRS + n;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = w[2 * t * OT * 64];
      const float b0 = bp[mdim(2 * t, 0) * PRS], b1 = bp[mdim(2 * t, 0) * PRS + 16];
      acc[0] = SDEH_MFMA16Q(a, b0, acc[0]);
      acc[1] = SDEH_MFMA16Q(a, b1, acc[1]);
    }
    keep_z(l + 1);
  }
  act_tile4<ACT>(acc[0]);
  act_tile4<ACT>(acc[1]);
  // output layer: partial sums over this wave's 16 channels; instruction q: lane (n, kk) holds channel 16 mw + 4 kk + q = packed k-step
  // 16 (mw >> 1) + q + 4 (2 (mw & 1) + (kk >> 1)), lane half kk & 1
#pragma unroll
  for (int ct = 0; ct < (DP + 15) / 16; ++ct) {
    f32x4q u[2];
    if (mw == 0) {
      u[0] = *reinterpret_cast<const f32x4q*>(lds + L.b_out + (kk & 1) * 16 + 4 * (2 * ct + (kk >> 1)));
      u[1] = u[0];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) u[0][q] = u[1][q] = 0.0f;
    }
    const float* __restrict__ w = lds + L.w_out + (16 * (mw >> 1) + 4 * (2 * (mw & 1) + (kk >> 1))) * 64 + 32 * (kk & 1) + 16 * ct + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a = w[q * 64];
      u[0] = SDEH_MFMA16Q(a, acc[0][q], u[0]);
      u[1] = SDEH_MFMA16Q(a, acc[1][q], u[1]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 16 * ct + 4 * kk + q;
      if (c < XR) { pout[c * 64 + n] = u[0][q]; pout[c * 64 + 16 + n] = u[1][q]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------
// PLANES: the training forward (sdeh_simulate_fwd_train) -- the M waves also store the pre-activation planes, the V wave the raw
// network output; a separate instantiation so that the evaluation kernel carries none of it.
// PLANES: 0 = evaluation; 1 = training forward that keeps the pre-activation planes and raw network outputs (sdeh_simulate_fwd_train);
// 2 = training forward for the fused backward (sdeh_simulate_fwd_train2: coordinate-major trajectory / score planes from the V wave
// only -- the M waves run the evaluation code: with the plane stores compiled in they lose 1 ms per 100 steps at B = 65 536);
// 3 = 2 + the pre-activation RECORD from the M waves (ZRec: 16-byte non-temporal stores, 1 KB per instruction) and the raw network
// output [T, d, B] from the V wave (sdeh_simulate_fwd_train3: the fused backward reads them instead of re-evaluating the network).
template <int DP, int C, bool PAD, int LOSS, int CTRL, int TGT, int GMMV, int ACT, int REFC, int GNV, int PLANES>
__global__ __launch_bounds__(512) void traj_ws_kernel(const float* __restrict__ ws, const float* __restrict__ x0,
                                                      const float* __restrict__ noise, float* __restrict__ xT,
                                                      float* __restrict__ rnd_out, float* __restrict__ xs,
                                                      const TrajArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int XR = xrows<DP>();
  const WsLayout& L = A.lay;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool pair = A.half >= 2;  // one group of 32 trajectories per workgroup: V + two M waves (half == 2) or V + four (half == 3: quad)
  const bool quad = A.half == 3;
  const int n_groups = pair ? 1 : (int)(blockDim.x >> 7);
  const bool is_m = wave >= n_groups;
  const int group = pair ? 0 : (is_m ? wave - n_groups : wave);

  {  // stage the LDS image (packed weights + GMM tables) once per workgroup
    const float4* src = reinterpret_cast<const float4*>(ws);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < L.lds_floats / 4; i += (int)blockDim.x) dst[i] = src[i];
  }
  float* __restrict__ xbuf = lds + L.lds_floats + group * (XR * 64);
  // pair mode: the second M wave's partial network output (behind the exchange buffer and the activation parking); zeroed once --
  // the M waves write columns 0..31 only, the idle lanes of the V wave read the others
  // (quad mode: the exchange planes sit in between, and there are three such buffers)
  float* __restrict__ xbuf2 = xbuf + XR * 64 + (quad ? 2 * 64 * kQuadPRS : 2 * 2 * 16 * 64);
  if (pair)
    for (int i = tid; i < (quad ? 3 : 1) * XR * 64; i += (int)blockDim.x) xbuf2[i] = 0.0f;
  // hand-off counters of the group (behind the exchange buffers): [0] x published by V, [1] network output published by M
  int* hand = reinterpret_cast<int*>(lds + L.lds_floats + kWsGroups * (XR * 64)) + 2 * group;
  const bool fsync = A.flag_sync != 0 && !pair;
  // d <= 4 (launcher: TrajArgs::vout): [64 channels][64 trajectories] per group behind the counters -- the M wave's last activations
  float* __restrict__ abuf = nullptr;
  if constexpr (DP <= 4 && C == 64) {
    if (A.vout && !pair) abuf = lds + L.lds_floats + kWsGroups * (XR * 64) + 2 * kWsGroups + group * (64 * 64);
  }
  if (fsync && tid < 2 * kWsGroups) reinterpret_cast<int*>(lds + L.lds_floats + kWsGroups * (XR * 64))[tid] = 0;

  const int ctrl_kind = CTRL >= 0 ? CTRL : A.ctrl_kind, loss_kind = LOSS >= 0 ? LOSS : A.loss_kind;
  const int gmmv = GMMV >= 0 ? GMMV : L.gmm_lds, act = ACT >= 0 ? ACT : A.act;
  const int flags = A.flags;
  const int d = PAD ? A.d : DP;
  const int n_steps = A.n_steps;

  if (is_m) {
    // =================================================================================== M wave
    constexpr int OT = C / 32;
    const int h = lane >> 5;
#if SDEH_MPRIO > 0
    __builtin_amdgcn_s_setprio(SDEH_MPRIO);
#endif
    __syncthreads();  // LDS image staged
    // the pre-activation record (PLANES == 3): tiles of 32 trajectories, (Lh + 1) layers of 2048 floats per tile and step
    const long long zr_tiles = (A.batch + 31) >> 5;
    const int zr_lh1 = L.n_hidden + 1;
    const int zr_stride = zrec_tile_floats(L.n_hidden, d);
    const long long zr_step = zr_tiles * zr_stride;
    if constexpr (C == 64 && DP <= 32 && PLANES != 1) {
      if (quad) {
        const int mw = wave - 1;
        const int kk = lane >> 4;
        const int epos = ((mw >> 1) * 2 + (kk & 1)) * 16 + 4 * (2 * (mw & 1) + (kk >> 1));  // this lane's four channels in the M-order rows
        float* __restrict__ planes = xbuf + XR * 64;
        float* __restrict__ pout = mw == 0 ? xbuf : xbuf2 + (mw - 1) * XR * 64;
        int parity = 0;
        f32x4q emb1 = *reinterpret_cast<const f32x4q*>(ws + L.emb + epos);
        ws_barrier();  // barrier A: x_0 published
        ZRec Zr{nullptr, 0, zr_lh1, zr_stride};
        if constexpr (PLANES == 3) {
          if (A.zrec != nullptr) { Zr.base = A.zrec + (long long)blockIdx.x * zr_stride; Zr.tiles = 1; }
        }
        for (int i = 0; i < n_steps; ++i) {
          SDEH_ACT_SWITCH(act, ACTC, ws_mlp_quad<DP, C, ACTC, (PLANES == 3)>(lds, xbuf, planes, pout, L, emb1, lane, mw, parity, Zr););
          if constexpr (PLANES == 3) Zr.base += zr_step;
          ws_barrier();  // barrier B: network output published
          if (i + 1 < n_steps) emb1 = *reinterpret_cast<const f32x4q*>(ws + L.emb + (i + 1) * C + epos);
          ws_barrier();  // barrier A: x_{i+1} published
        }
        return;
      }
    }
    if constexpr (C == 64) {
      if (pair) {
        const int mw = wave - 1;
        float* __restrict__ abuf = xbuf + XR * 64;
        int parity = 0;
        f32x16 emb1 = load16(ws + L.emb + (mw * 2 + h) * 16);
        ws_barrier();  // barrier A: x_0 published
        const long long row0 = (long long)blockIdx.x * 32;
        ZStore Z{nullptr, (long long)n_steps * A.batch, A.zt_out == nullptr ? 0 : (int)(A.batch - row0 < 32 ? A.batch - row0 : 32)};
        ZRec Zr{nullptr, 0, zr_lh1, zr_stride};
        if constexpr (PLANES == 3) {
          if (A.zrec != nullptr) { Zr.base = A.zrec + (long long)blockIdx.x * zr_stride; Zr.tiles = 1; }
        }
        for (int i = 0; i < n_steps; ++i) {
          if constexpr (PLANES == 1) Z.base = A.zt_out + (long long)i * A.batch + row0;
          SDEH_ACT_SWITCH(act, ACTC, ws_mlp_pair<DP, C, ACTC, (PLANES == 1 ? 1 : (PLANES == 3 ? 2 : 0))>(lds, xbuf, abuf, abuf + 2 * 2 * 16 * 64, L, emb1, lane, mw, parity, Z, Zr););
          if constexpr (PLANES == 3) Zr.base += zr_step;
          ws_barrier();  // barrier B: network output published
          if (i + 1 < n_steps) emb1 = load16(ws + L.emb + (i + 1) * C + (mw * 2 + h) * 16);
          ws_barrier();  // barrier A: x_{i+1} published
        }
        return;
      }
    }
    f32x16 emb[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) emb[ot] = load16(ws + L.emb + (ot * 2 + h) * 16);
    if (fsync) ws_flag_wait(hand, 1);
    else ws_barrier();  // barrier A: x_0 published
    ZStore Z{nullptr, 0, 0};
    long long row0 = 0;
    if constexpr (PLANES == 1) {
      const int rpg_m = A.half ? 32 : 64;
      row0 = (long long)blockIdx.x * (rpg_m * n_groups) + group * rpg_m;
      Z.N = (long long)n_steps * A.batch;
      Z.rows = (int)(A.batch - row0 < rpg_m ? (A.batch - row0 > 0 ? A.batch - row0 : 0) : rpg_m);
      if (A.zt_out == nullptr) Z.rows = 0;  // fused backward (sdeh_simulate_fwd_train2): no pre-activation planes
    }
    ZRec Zr{nullptr, 0, zr_lh1, zr_stride};
    if constexpr (PLANES == 3) {
      if (A.zrec != nullptr) {
        const int tpg = A.half ? 1 : 2;  // 32-trajectory tiles per group
        const long long tile0 = ((long long)blockIdx.x * n_groups + group) * tpg;
        Zr.base = A.zrec + tile0 * zr_stride;
        Zr.tiles = (int)(zr_tiles - tile0 < tpg ? (zr_tiles - tile0 > 0 ? zr_tiles - tile0 : 0) : tpg);
      }
    }
    for (int i = 0; i < n_steps; ++i) {
      WS_T(tm0);
      if constexpr (PLANES == 1) Z.base = A.zt_out + (long long)i * A.batch + row0;
      // generic variants (ACT < 0): the activation id becomes a compile-time constant of three copies of the network --
      // selecting it per element costs a scalar branch per element pair inside the MFMA stages (1.5x on the whole kernel)
      SDEH_ACT_SWITCH(act, ACTC,
        if (A.half) ws_mlp_half<DP, C, (PLANES == 1 ? 1 : (PLANES == 3 ? 2 : 0))>(lds, xbuf, L, ACTC, emb, lane, Z, Zr, abuf);
        else ws_mlp<DP, C, (PLANES == 1 ? 1 : (PLANES == 3 ? 2 : 0))>(lds, xbuf, L, ACTC, emb, lane, Z, Zr, abuf););
      if constexpr (PLANES == 3) Zr.base += zr_step;
      if (fsync) ws_flag_set(hand + 1, i + 1);
      else ws_barrier();  // barrier B: network output published
      WS_T(tm1);
      if (i + 1 < n_steps) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) emb[ot] = load16(ws + L.emb + (i + 1) * C + (ot * 2 + h) * 16);
      }
      if (fsync) ws_flag_wait(hand, i + 2);
      else ws_barrier();  // barrier A: x_{i+1} published
      WS_T(tm2);
      if (group == 0) { WS_ADD(0, tm0, tm1); WS_ADD(1, tm1, tm2); }
    }
    return;
  }

  // ===================================================================================== V wave
#if SDEH_VPRIO > 0
  __builtin_amdgcn_s_setprio(SDEH_VPRIO);
#endif
  const int rpg = A.half ? 32 : 64;  // trajectories per group
  const long long row = (long long)blockIdx.x * (rpg * n_groups) + group * rpg + lane;
  const bool live = lane < rpg && row < A.batch;
  const long long lrow = live ? row : A.batch - 1;  // dead lanes shadow the last row and never store
  DensArgs tgt = A.target;
  if (TGT >= 0) tgt.kind = TGT;

  float x[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    const float v = x0[lrow * d + (PAD ? min(j, d - 1) : j)];
    x[j] = (!PAD || j < d) ? v : 0.0f;
  }
  // exchange buffer: coordinates >= DP that M-layout registers can address stay zero for the whole launch
#pragma unroll
  for (int j = 0; j < XR; ++j) xbuf[j * 64 + lane] = j < DP ? x[j] : 0.0f;
  if constexpr (DP <= 4) {
    if (abuf != nullptr)  // groups of 32: the M wave writes columns 0..31 only, this wave's idle lanes read the others
      for (int c = 0; c < 64; ++c) abuf[c * 64 + lane] = 0.0f;
  }
  __syncthreads();  // LDS image staged

  float rnd = 0.0f;
  if (flags & SDEH_FLAG_INIT_LOGP) rnd = dgauss_logp<DP>(ws + L.dg[2], x);
  if (xs != nullptr && live) {
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (!PAD || j < d) xs[lrow * d + j] = x[j];
  }
  if constexpr (PLANES >= 2) {
    if (A.xs_cm != nullptr && live) {  // the trajectory for the fused backward, coordinate-major [T+1][d][B]
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (!PAD || j < d) A.xs_cm[(long long)j * A.batch + lrow] = x[j];
    }
  }
  const bool lv = flags & SDEH_FLAG_CHANGE_SDE_CTRL;
  const bool need_t = ctrl_kind == SDEH_CTRL_SCORE || ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool refc = REFC >= 0 ? REFC != 0 : (flags & SDEH_FLAG_REFERENCE_CTRL) && loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);
  // (32-trajectory groups: the row whose noise this lane helps to draw -- lanes 32..63 work for trajectory lane - 32)
  const long long row_n = (long long)blockIdx.x * (rpg * n_groups) + group * rpg + (lane & 31);
  const unsigned long long grow_n = (unsigned long long)(A.row_offset + (row_n < A.batch ? row_n : A.batch - 1));
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
  if (fsync) ws_flag_set(hand, 1);
  else ws_barrier();  // barrier A: x_0 published

  for (int i = 0; i < n_steps; ++i) {
    WS_T(tv0);
    cfp cf = as_const(ws + L.coef + i * kCoefStride);
    const float dt = cf[CF_DT], sqdt = cf[CF_SQDT], sig = cf[CF_SIGMA];

    // pair mode: the M waves exchange activations at n_hidden (d > 32: n_hidden + 1) workgroup barriers per step; this wave joins
    // them, spaced through its own work so that it never arrives late (input layer ~0.4 us, then ~1.1 us per layer)
    constexpr bool kPairSum = row_tiles(DP) == 1;  // the network output arrives as two partial sums (ws_mlp_pair)
    const int n_join = L.n_hidden + (kPairSum ? 0 : 1);
    if (pair && n_join >= 1) ws_barrier();
    // ---- score term of the control (needs x only; runs while the M wave evaluates the network) -----------
    float sterm[DP];
    if (ctrl_kind != SDEH_CTRL_CLIPPED) {
      float tsc[DP], psc[DP];
      if (need_t) {  // pair / quad mode: mixture components split over the two lane halves (gmm_online KS)
        if (pair) ws_target_score<DP, (GNV > 0 ? GNV : DP), true>(tgt, ws, lds, L, gmmv, d, x, tsc);
        else ws_target_score<DP, (GNV > 0 ? GNV : DP), false, true, gmm_mm_compiled<DP, PAD, GMMV, GNV>()>(tgt, ws, lds, L, gmmv, d, x, tsc);
      }
      if (ctrl_kind == SDEH_CTRL_LERP || ctrl_kind == SDEH_CTRL_LERP_PRIOR) dgauss_score<DP>(ws + L.dg[1], x, psc);
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
      if constexpr (PLANES >= 2) {  // the fused backward reads the combined score instead of re-evaluating the densities
        if (A.sc_out != nullptr && live) {  // coordinate-major [T][d][B]: consecutive lanes, consecutive addresses
          float* __restrict__ sp = A.sc_out + (long long)i * d * A.batch + lrow;
#pragma unroll
          for (int j = 0; j < DP; ++j)
            if (!PAD || j < d) sp[(long long)j * A.batch] = sterm[j];
        }
      }
      cfp gam = as_const(ws + L.gam + i * L.g);
      const float mult = ctrl_kind == SDEH_CTRL_SCORE ? 1.0f : sig;  // Lerp*: ctrl + sde.diff(t) * score
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
    SDEH_FENCE();

    if (pair && n_join >= 2) ws_barrier();
    // ---- Gaussian draws (independent of the control) --------------------------------------------------------
    float xi[DP];
    if (noise != nullptr) {
      const float* __restrict__ np = noise + ((long long)i * A.batch + lrow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j) xi[j] = np[PAD ? min(j, d - 1) : j];
    } else if (A.half) {
      // Groups of 32 trajectories leave lanes 32..63 of this wave idle: they draw the ODD Philox blocks of trajectory lane - 32, the
      // lower lanes the even ones, and one v_permlane32_swap per value hands each half the other's (same counters, same values:
      // bit-identical noise for half the Philox / Box-Muller instructions, the largest item of the V wave's step)
      constexpr int NB = (DP + 3) / 4;
      const int hh = lane >> 5;
#pragma unroll
      for (int jp = 0; jp < (NB + 1) / 2; ++jp) {
        const int jb = 2 * jp + hh;
        float n[4];
        box_muller4(philox_block(A.seed, rng_off, grow_n, i, jb), n);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = (!PAD || 4 * jb < d) ? n[q] : 0.0f;
          auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
          if (8 * jp + q < DP) xi[8 * jp + q] = __uint_as_float(r[0]);          // block 2 jp: drawn by the lower lanes
          if (8 * jp + 4 + q < DP) xi[8 * jp + 4 + q] = __uint_as_float(r[1]);  // block 2 jp + 1: drawn by the upper lanes
        }
        SDEH_FENCE();
      }
    } else {
#pragma unroll
      for (int jb = 0; jb < (DP + 3) / 4; ++jb) {
        float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!PAD || 4 * jb < d) {
          box_muller4(philox_block(A.seed, rng_off, grow, i, jb), n);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * jb + q < DP) xi[4 * jb + q] = n[q];
        SDEH_FENCE();
      }
    }

    if (pair)
      for (int k = 3; k <= n_join; ++k) ws_barrier();
    // exponential integrator (oc.py:428-443):  x <- x a_k + (b_k^2 s^2) u + (s b_k) xi
    // Euler-Maruyama (oc.py:213-219, 325-331): x <- x + (f x + sig u) dt + sig (xi sqrt(dt))
    const bool expo = loss_kind == SDEH_LOSS_EXPONENTIAL;
    const float c_x = expo ? cf[CF_ALPHAK] : fmaf(cf[CF_DRIFT], dt, 1.0f);
    const float c_u = expo ? cf[CF_B2S2] : sig * dt;
    const float c_n = expo ? cf[CF_SBK] : sig * sqdt;
    const float c_i = expo ? cf[CF_SBK] : sqdt;  // Ito term: sum(g * xi) * c_i
    float rsub[DP];  // reference_ctrl = sigma(t) * prior.score(x) (solver/oc.py:305-306), evaluated at the step's input
    if (refc) {
      dgauss_score<DP>(ws + L.dg[1], x, rsub);
#pragma unroll
      for (int j = 0; j < DP; ++j) rsub[j] *= sig;
    }
    // the part of the update that does not need the control: x <- c_x x + c_n xi  (the control term is added below)
#pragma unroll
    for (int j = 0; j < DP; ++j) x[j] = fmaf(c_n, xi[j], c_x * x[j]);
    SDEH_FENCE();

    WS_T(tv1);
    if (fsync) ws_flag_wait(hand + 1, i + 1);
    else ws_barrier();  // barrier B: the M wave has published the network output
    WS_T(tv2);
    // ---- u = clip(nn) + score term; publish x_{i+1} first: the M wave is idle until barrier A ------------------------
    float u[DP];
    if (quad) {  // the network output is the sum of the M waves' partial sums (ws_mlp_quad: four; ws_mlp_pair: two); one branch each
#pragma unroll
      for (int j = 0; j < DP; ++j)
        u[j] = ((xbuf[j * 64 + lane] + xbuf2[j * 64 + lane]) + xbuf2[(XR + j) * 64 + lane]) + xbuf2[(2 * XR + j) * 64 + lane];
    } else if (pair && kPairSum) {
#pragma unroll
      for (int j = 0; j < DP; ++j) u[j] = xbuf[j * 64 + lane] + xbuf2[j * 64 + lane];
    } else if (DP <= 4 && abuf != nullptr) {
      if constexpr (DP <= 4) {
        // out_layer(act(e)) on the vector pipe: rows 0..3 of the packed out layer are one broadcast ds_read_b128 per channel (k-step s,
        // lane half hh of the packed image <-> channel mdim(s, hh)); two chains per output
        int woff = L.w_out;
        asm volatile("" : "+v"(woff));  // one base, immediate offsets: nothing per channel to hoist out of the step loop
        const float* __restrict__ ab = abuf + lane;
        float acc[2][DP];
#pragma unroll
        for (int j = 0; j < DP; ++j) { acc[0][j] = lds[L.b_out + j]; acc[1][j] = 0.0f; }
#pragma unroll
        for (int s = 0; s < C / 2; ++s)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float4 wv = *reinterpret_cast<const float4*>(lds + woff + s * 64 + hh * 32);
            const float av = ab[mdim(s, hh) * 64];
            const float wr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int j = 0; j < DP; ++j) acc[hh][j] = fmaf(wr[j], av, acc[hh][j]);
          }
#pragma unroll
        for (int j = 0; j < DP; ++j) u[j] = acc[0][j] + acc[1][j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) u[j] = xbuf[j * 64 + lane];
    }
    if constexpr (PLANES == 3) {
      // the raw network output joins the pre-activation record (ZRec: [coordinate quad][trajectory][4] behind the layers).  The M wave
      // stores it where it forms the whole sum (ws_mlp / ws_mlp_half); here the cases in which only this wave has it: the out layer on
      // the vector pipe (d <= 4) and the pair / quad modes' partial sums (d <= 32) -- 16-byte stores, ceil(d / 4) per lane
      // (rows of the last 32-row tile beyond the batch are written as ZEROS: the record is uninitialised memory otherwise, and a backward
      // launch that reads it weighs those rows by 0 -- 0 x NaN would reach the d gamma / clip partial sums)
      const bool in_tile = lane < rpg && (row >> 5) < (long long)((A.batch + 31) >> 5);
      if (DP <= 32 && A.zrec != nullptr && in_tile && (pair || (DP <= 4 && abuf != nullptr))) {
        const int zr_stride = zrec_tile_floats(L.n_hidden, d);
        float* __restrict__ zp = A.zrec + ((long long)i * ((A.batch + 31) >> 5) + (row >> 5)) * zr_stride + (L.n_hidden + 1) * 2048 + (int)(row & 31) * 4;
#pragma unroll
        for (int g4 = 0; g4 < (DP + 3) / 4; ++g4) {
          f32x4z v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (live && 4 * g4 + e < DP && (!PAD || 4 * g4 + e < d)) ? u[4 * g4 + e < DP ? 4 * g4 + e : 0] : 0.0f;
          if (!PAD || 4 * g4 < d) __builtin_nontemporal_store(v, reinterpret_cast<f32x4z*>(zp + (g4 >> 3) * 1024 + (g4 & 7) * 128));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const float nn = u[j];
      if constexpr (PLANES == 1) {
        if (A.nn_out != nullptr && live && (!PAD || j < d)) A.nn_out[((long long)i * A.batch + lrow) * d + j] = nn;
      }
      u[j] = clipf(nn, A.clip_model) + sterm[j];
      if (PAD) u[j] = j < d ? u[j] : 0.0f;
      x[j] = fmaf(c_u, u[j], x[j]);
      if (PAD) x[j] = j < d ? x[j] : 0.0f;
      xbuf[j * 64 + lane] = x[j];
    }
    if constexpr (PLANES >= 2) {  // Bridge training: the inference pass (sdeh_bridgef.hip) is row-parallel given x_t and u_t
      if (A.u_out != nullptr && live) {
        float* __restrict__ up = A.u_out + (long long)i * d * A.batch + lrow;
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if (!PAD || j < d) up[(long long)j * A.batch] = u[j];
      }
    }
    if (fsync) ws_flag_set(hand, i + 2);
    else ws_barrier();  // barrier A: x_{i+1} published
    WS_T(tv3);
    // ---- running cost (losses/oc.py:204-211, 319-323, 418-431) and Ito term, in the shadow of the next network pass -----
    float cost = 0.0f;
    if (!refc) {
      // u (u - 0.5 u) accumulates to exactly half of what u u does (scaling by 0.5 commutes with every rounding), so the
      // log-variance and the KL form of the cost are the same sum
#pragma unroll
      for (int j = 0; j < DP; ++j) cost = fmaf(u[j], u[j], cost);
      cost *= 0.5f;
    } else if (lv) {
#pragma unroll
      for (int j = 0; j < DP; ++j) cost = fmaf(u[j] - rsub[j], u[j] - 0.5f * (rsub[j] + u[j]), cost);
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j) { const float g = u[j] - rsub[j]; cost = fmaf(g, g, cost); }
      cost *= 0.5f;
    }
    if (expo) rnd = fmaf(cf[CF_B2S2], cost, rnd);
    else rnd = fmaf(cost, dt, rnd);
    if (loss_kind == SDEH_LOSS_TIME_REVERSAL && !(flags & SDEH_FLAG_TRAIN)) rnd -= cf[CF_DDIV];
    if (flags & SDEH_FLAG_ITO) {  // the control entering the Ito term: gen_plus_inf / gen_minus_ref
      float itosum = 0.0f;
      if (refc) {
#pragma unroll
        for (int j = 0; j < DP; ++j) itosum = fmaf(u[j] - rsub[j], xi[j], itosum);
      } else {
#pragma unroll
        for (int j = 0; j < DP; ++j) itosum = fmaf(u[j], xi[j], itosum);
      }
      rnd = fmaf(itosum, c_i, rnd);
    }

    if (xs != nullptr && live) {
      float* __restrict__ xp = xs + ((long long)(i + 1) * A.batch + lrow) * d;
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (!PAD || j < d) xp[j] = x[j];
    }
    if constexpr (PLANES >= 2) {
      if (A.xs_cm != nullptr && live) {
        float* __restrict__ xp = A.xs_cm + (long long)(i + 1) * d * A.batch + lrow;
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if (!PAD || j < d) xp[(long long)j * A.batch] = x[j];
      }
    }
    WS_T(tv4);
    if (group == 0) { WS_ADD(2, tv0, tv1); WS_ADD(3, tv1, tv2); WS_ADD(4, tv2, tv3); WS_ADD(5, tv3, tv4); WS_ADD(6, 0ull, 1ull); }
  }

  // ---- terminal costs (oc.py:225, 337, 449-450) ----------------------------------------------------------
  if (flags & SDEH_FLAG_TERMINAL_SECOND) rnd += dgauss_logp<DP>(ws + L.dg[2], x);
  if (flags & SDEH_FLAG_TERMINAL_TARGET) rnd -= clipf(ws_target_logp<DP, (GNV > 0 ? GNV : DP), true>(tgt, ws, lds, L, gmmv, d, x), A.clip_target);
  if (live) {
    rnd_out[row] = rnd;
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (!PAD || j < d) xT[row * d + j] = x[j];
  }
  if constexpr (PLANES >= 2) {  // d(terminal target cost)/dx_T, negated: the clamp's mask times target.score(x_T)  (oc.py:225)
    if (A.tsc_out != nullptr && (flags & SDEH_FLAG_TERMINAL_TARGET)) {
      float st[DP];
      ws_target_score<DP, (GNV > 0 ? GNV : DP), false, true>(tgt, ws, lds, L, gmmv, d, x, st);
      float keep = 1.0f;
      if (A.clip_target < 3.0e38f)
        keep = fabsf(ws_target_logp<DP, (GNV > 0 ? GNV : DP), true>(tgt, ws, lds, L, gmmv, d, x)) <= A.clip_target ? 1.0f : 0.0f;
      if (live) {
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if (!PAD || j < d) A.tsc_out[(long long)j * A.batch + row] = keep * st[j];
      }
    }
  }
}

template <int DP>
inline size_t ws_lds_bytes(const WsLayout& L) {
  return ((size_t)L.lds_floats + (size_t)kWsGroups * xrows<DP>() * 64 + 2 * kWsGroups) * sizeof(float);  // + the hand-off counters
}
// d <= 4 with the out layer on the V wave: + the activation planes [groups][64][64]
template <int DP>
inline size_t ws_vout_lds_bytes(const WsLayout& L) {
  return ws_lds_bytes<DP>(L) + (size_t)kWsGroups * 64 * 64 * sizeof(float);
}
// pair mode: one exchange buffer + the activation parking (2 parities x 2 tiles x 16 registers x 64 lanes) + the second M wave's
// partial network output
template <int DP>
inline size_t ws_pair_lds_bytes(const WsLayout& L) {
  return ((size_t)L.lds_floats + (size_t)2 * xrows<DP>() * 64 + 2 * 2 * 16 * 64) * sizeof(float);  // x / partial 0 | parking | partial 1
}
// quad mode: x / partial 0 | two exchange planes [64][48] | partials 1..3
template <int DP>
inline size_t ws_quad_lds_bytes(const WsLayout& L) {
  return ((size_t)L.lds_floats + (size_t)4 * xrows<DP>() * 64 + 2 * 64 * kQuadPRS) * sizeof(float);
}

template <int DP, int C, bool PAD, int LOSS, int CTRL, int TGT, int GMMV, int ACT, int REFC, int GNV>
int launch_traj_ws(const TrajArgs& a, hipStream_t stream) {
  size_t lds_bytes = ws_lds_bytes<DP>(a.lay);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  // (pair / quad mode: at least one hidden layer -- its exchange barrier is what separates the M waves' reads of x from the write of the
  // first partial network output into the same buffer)
  const bool pair_fits = C == 64 && a.lay.n_hidden >= 1 && ws_pair_lds_bytes<DP>(a.lay) <= 160 * 1024;
  const int planes = (a.zt_out != nullptr && a.nn_out != nullptr) ? 1 : (a.zrec != nullptr ? 3 :
                     (a.sc_out != nullptr || a.tsc_out != nullptr || a.xs_cm != nullptr ? 2 : 0));
  static bool attr_done[kMaxDevices] = {};  // the raised LDS limit is a per-device function attribute
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return SDEH_ERR_HIP;
    attr_set = true;
  }
  // Placement by batch size (256 CUs x 4 SIMDs; the register budget allows two of these waves per SIMD):
  //   B >  32 768 : groups of 64, G = 4 -- V and M wave of a group share a SIMD (B = 65 536 fills every SIMD exactly)
  //   B <= 32 768 : groups of 32 (one MFMA column tile per M wave: half the dependent chain per step), G = 4, sharing a SIMD
  //   B <= 16 384 : groups of 32, G = 2 -- every wave has a SIMD of its own, V and M really run concurrently; the reference's
  //                 default batch sizes (train 512 / 2048, eval 6000) live here and are latency-bound: 14.5 -> 7.1 us per step
  //   B <=  8 192 : pair mode -- one group of 32 per workgroup served by a V wave and TWO M waves (one output-channel tile each),
  //                 three waves on three SIMDs of a CU: 7.1 -> ~5 us per step
  const char* force = plan_opt(OPT_WS_GROUPS);  // testing aid, a plan option like the other switches: "2" | "4" | "2h" | "4h" | "p" (pair)
  int groups = a.batch <= 2 * 32 * 256 ? 2 : kWsGroups;
  int half = a.batch <= 4 * 32 * 256 ? 1 : 0;
  if (pair_fits && a.batch <= 32 * 256) { groups = 1; half = 2; }
  if (force != nullptr && (force[0] == '2' || force[0] == '4')) { groups = force[0] - '0'; half = force[1] == 'h' ? 1 : 0; }
  if (force != nullptr && force[0] == 'p' && pair_fits) { groups = 1; half = 2; }
  //   B <=  8 192 (16 384 with a closed-form target) : quad mode -- FOUR M waves on 16-row tiles (v_mfma_f32_16x16x4_f32), five waves per group of 32: d <= 32, no plane
  //                 stores, activations without a kink (ReLU: the fused backward re-evaluates the pair mode's pre-activations bitwise)
  const bool quad_fits = C == 64 && DP <= 32 && a.lay.n_hidden >= 1 && planes != 1 && (ACT >= 0 ? ACT : a.act) != SDEH_ACT_RELU &&
                         ws_quad_lds_bytes<DP>(a.lay) <= 160 * 1024;
  const char* quad_env = plan_opt(OPT_WS_QUAD);  // "0": never, "1": whenever it fits (tests); a plan option
  // Measured (tools/quad_threshold_timing.py, us per step quad / otherwise): B = 8192: 2.9 / 4.1 (d = 1), 2.8 / 4.1 (d = 10), 5.4 / 5.9 (d = 2,
  // 40-component mixture); B = 16 384 (two workgroups per CU): 5.5 / 7.0 for the closed-form targets, 10.5 / 7.3 for the mixture (its V
  // waves then compete for the SIMDs); B = 24 576: 8.0 / 7.0.
  const bool light_v = (TGT >= 0 ? TGT : a.target.kind) != SDEH_DENS_GMM;
  if (quad_fits && force == nullptr && (quad_env != nullptr ? quad_env[0] == '1' : (a.batch <= 32 * 256 || (light_v && a.batch <= 64 * 256)))) {
    groups = 1;
    half = 3;
  }
  // matrix-pipe mixture layout (no tables in LDS): whole-wave modes of the instantiations that carry gmm_mm only (the API checks the same)
  if (a.lay.gmm_lds >= 3 && (half >= 2 || gmm_mm_compiled<DP, PAD, GMMV, GNV>() < a.lay.gmm_lds - 2)) return SDEH_ERR_UNSUPPORTED;
  if (half == 2 && ws_pair_lds_bytes<DP>(a.lay) > lds_bytes) lds_bytes = ws_pair_lds_bytes<DP>(a.lay);
  if (half == 3 && ws_quad_lds_bytes<DP>(a.lay) > lds_bytes) lds_bytes = ws_quad_lds_bytes<DP>(a.lay);
  TrajArgs b = a;
  b.half = half;
  // d <= 4, groups of 64 / 32 trajectories: the out layer on the V wave's vector pipe when the activation planes fit (SDEH_WS_VOUT=0: never)
  b.vout = 0;
  if (DP <= 4 && C == 64 && half <= 1 && ws_vout_lds_bytes<DP>(a.lay) <= 160 * 1024) {
    const char* vo = plan_opt(OPT_WS_VOUT);  // A/B aid, a plan option
    if (vo == nullptr || vo[0] != '0') {
      b.vout = 1;
      if (ws_vout_lds_bytes<DP>(a.lay) > lds_bytes) lds_bytes = ws_vout_lds_bytes<DP>(a.lay);
    }
  }
  b.flag_sync = plan_opt(OPT_WS_BARRIER) == nullptr ? 1 : 0;  // A/B aid (a plan option): the workgroup-barrier hand-off
  const int rows = (half ? 32 : 64) * groups;
  const unsigned grid = (unsigned)((a.batch + rows - 1) / rows);
  if (planes == 1)
    hipLaunchKernelGGL((traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 1>), dim3(grid),
                       dim3(half == 3 ? 320 : (half == 2 ? 192 : 128 * groups)), lds_bytes, stream, a.ws, a.x0, a.noise, a.xT, a.rnd, a.xs, b);
  else if (planes == 3)
    hipLaunchKernelGGL((traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 3>), dim3(grid),
                       dim3(half == 3 ? 320 : (half == 2 ? 192 : 128 * groups)), lds_bytes, stream, a.ws, a.x0, a.noise, a.xT, a.rnd, a.xs, b);
  else if (planes == 2)
    hipLaunchKernelGGL((traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 2>), dim3(grid),
                       dim3(half == 3 ? 320 : (half == 2 ? 192 : 128 * groups)), lds_bytes, stream, a.ws, a.x0, a.noise, a.xT, a.rnd, a.xs, b);
  else
    hipLaunchKernelGGL((traj_ws_kernel<DP, C, PAD, LOSS, CTRL, TGT, GMMV, ACT, REFC, GNV, 0>), dim3(grid),
                       dim3(half == 3 ? 320 : (half == 2 ? 192 : 128 * groups)), lds_bytes, stream, a.ws, a.x0, a.noise, a.xT, a.rnd, a.xs, b);
#ifdef SDEH_WS_PROFILE
  {
    (void)hipStreamSynchronize(stream);
    unsigned long long v[8];
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(ws_prof), sizeof(v));
    const double n = v[6] ? (double)v[6] : 1.0;
    fprintf(stderr, "ws phases (cycles per step, block 0 / group 0, %llu steps): M network %.0f | M waits for x %.0f || V score+noise %.0f | "
            "V waits for nn %.0f | V hand-off %.0f | V cost/stores %.0f\n", v[6], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5] / n);
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ws_prof), z, sizeof(z));
  }
#endif
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

}  // namespace sdeh
