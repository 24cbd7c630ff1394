// Bridge training, the divergence term of the 64-channel inference network, FUSED (round 4; VERDICT r03 next-step 3 / missing 5).
// Reference: losses/oc.py:189-202 (running cost + sigma div_x v dt), utils/autograd.py:14-21 (div_x v from d backward passes with
// create_graph=True: loss.backward() differentiates them again).  Loss contribution of row n = (t, i):
//     sum_j c_j J_jj(x_n),    c_j = w_i sigma_t dt_t 1[|nn_j| <= clip_model],    J_jj = W_out[j, :] D_2 W_2 D_1 W_1 D_0 W_in[:, j]
// with D_k = act'(Z_k) of the base pass (two hidden layers: conf/model/base/fouriermlp.yaml).  csrc/sdeh_bridge.hpp's
// bridge_div_bwd_kernel writes every per-(row, coordinate) vector to planes (3 (Lh + 1) C floats per row and coordinate: 126 GB at
// conf/solver/bridge.yaml's size) and leaves the contractions to sdeh_weight_grad.  Here nothing per coordinate leaves the chip.
//
// Per (row, coordinate) the term is evaluated from both ends (the scheme of sdeh_wide_bwd.hip's divergence kernel, §3g):
//     P = D_0 . W_in[:, j]     F = W_1 P        R = D_2 . W_out[j, :]     G = W_2^T R        J_jj = G^T D_1 F
//     U = c D_1 G              V = c D_1 F
//     dW_1 += U P^T            dW_2 += R V^T     dP = W_1^T U              dR = W_2 V
//     d loss / d D_1 += c F G  d loss / d D_0 += dP . W_in[:, j]           d loss / d D_2 += dR . W_out[j, :]
//     d W_in[:, j] += sum_n D_0 dP                d W_out[j, :] += sum_n D_2 dR
// -- six [64, 64] products per (32 rows, coordinate), the minimum.  Trajectory-split teams as in sdeh_bwdf2.hip: a wave owns 32 rows
// and all 64 channels, D_k and d loss / d D_k (6 x 32 registers) stay in its registers over the coordinate loop, F and G come straight
// from registers (the accumulator layout is the B-operand layout); the two weight-gradient products contract over rows, so (U, P) and
// (R, V) are published to the team's LDS planes and each wave accumulates ONE 32 x 32 tile of dW_1 / dW_2 over the team's 128 rows,
// its chunks issued interleaved with dP / dR (stage_cols / stage_rows).  The per-coordinate column / row of the in / out layer is a sum
// over rows = lanes: a reduce-scatter (77 instructions per 32 registers) leaves one channel per lane, added to the wave's own record
// with a no-return atomic (one wave per address: a sequential, deterministic sum).
// After the coordinates the base pass is evaluated once more for Z_k, and S_k = act''(Z_k) . d loss / d D_k goes out as three planes
// [64][T B]: the base chain adj(Z_k) = S_k + D_k W_{k+1}^T adj(Z_{k+1}), its weight gradients and the first-order terms of the
// inference network are sdeh_bwdf2.hip's row-parallel kernel (launch_bwdf2_bridge), which adds S_k where it multiplies by act'.
#include "sdeh_bwdf2.hpp"

namespace sdeh {

template <int OTD, int NQ>
__global__ __launch_bounds__(256) void bridge_divf_kernel(const BwdfArgs A) {
  using namespace bwdf2;
  static_assert(OTD == 1 || NQ == 16, "two coordinate tiles: all registers live");
  constexpr int DPP = 32 * OTD;
  constexpr int NGI = OTD == 2 ? 8 : NQ / 4;  // k-groups of the coordinates
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ WinT = lds;                    // [DPP][RSW]: input_embed.weight transposed (row j = column j)
  float* __restrict__ Whid = WinT + DPP * RSW;       // [2][64][RSW]
  float* __restrict__ Wout = Whid + 2 * 64 * RSW;    // [DPP][RSW]
  float* __restrict__ bh = Wout + DPP * RSW;         // [2][64]
  float* __restrict__ bo = bh + 2 * 64;              // [64]
  float* __restrict__ planes = bo + 64;              // [4 waves][D, A][64][RS]
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  float* __restrict__ Dme = planes + wave * 2 * PLANE;
  float* __restrict__ Ame = Dme + PLANE;
  const int wR = wave >> 1, wC = wave & 1;
  const int d = A.d, T = A.n_steps;
  const long long B = A.batch, N = B * T;

  for (int idx = tid; idx < DPP * RSW; idx += 256) {
    const int row = idx / RSW, col = idx - row * RSW;
    WinT[idx] = (row < d && col < 64) ? A.w_in[col * d + row] : 0.0f;
    Wout[idx] = (row < d && col < 64) ? A.w_out[row * 64 + col] : 0.0f;
  }
  for (int idx = tid; idx < 2 * 64 * RSW; idx += 256) {
    const int l = idx / (64 * RSW), rem = idx - l * 64 * RSW, row = rem / RSW, col = rem - row * RSW;
    Whid[idx] = col < 64 ? A.w_hid[l][row * 64 + col] : 0.0f;
  }
  if (tid < 128) bh[tid] = A.b_hid[tid >> 6][tid & 63];
  if (tid < 64) bo[tid] = tid < d ? A.b_out[tid] : 0.0f;
  for (int idx = tid; idx < 2 * 4 * PLANE; idx += 256) planes[idx] = 0.0f;
  __syncthreads();

#ifdef SDEH_BWDF2_ACT
  const int act = SDEH_BWDF2_ACT;
#else
  const int act = A.act;
#endif

  f32x16 dw_hid[2];
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int q = 0; q < 16; ++q) dw_hid[l][q] = 0.0f;

  // items: (step, quad of 32-row tiles), step-major; the team takes item blockIdx, blockIdx + gridDim, ...; wave w owns tile 4 quad + w
  const int n_tiles = A.n_tiles, n_quads = (n_tiles + 3) >> 2;
  const int n_teams = (int)gridDim.x, team_g = (int)blockIdx.x;
  const long long n_items = (long long)n_quads * T;
  const long long n_rounds = (n_items + n_teams - 1) / n_teams;
  // this wave's record of the per-coordinate columns / rows: lane -> channel 32 (jl >> 4) + 4 h + rrow(jl & 15)
  float* __restrict__ io_rec = A.div_io + ((long long)team_g * 4 + wave) * 2 * DPP * 64 + 32 * (j >> 4) + 4 * h + rrow(j & 15);

  const unsigned Bu = (unsigned)B;
  auto load_cm = [&](const float* __restrict__ plane_u, unsigned col, int ct) {  // (sdeh_bwdf2.hip: coordinate-major column, 32-bit offsets)
    f32x16 v;
    const int cb = 32 * ct + 4 * h;
    unsigned lane_off = ((unsigned)(cb < d ? cb : 0) * Bu + col) * 4u;
    unsigned s1 = Bu * 4u;
    asm volatile("" : "+v"(lane_off), "+v"(s1));
    const unsigned s2 = s1 + s1, s3 = s2 + s1, s8 = s1 << 3;
    const char* __restrict__ pb = reinterpret_cast<const char*>(plane_u);
    unsigned bg = lane_off;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < NQ) {
        const bool ok = cb + rrow(q) < d;
        const unsigned oq = (q & 3) == 0 ? bg : ((q & 3) == 1 ? bg + s1 : ((q & 3) == 2 ? bg + s2 : bg + s3));
        const unsigned off = ok ? oq : lane_off;
        const float val = *reinterpret_cast<const float*>(pb + off);
        v[q] = ok ? val : 0.0f;
        if ((q & 3) == 3) bg += s8;
      } else {
        v[q] = 0.0f;
      }
    }
    return v;
  };

  for (long long round = 0; round < n_rounds; ++round) {
    long long item = round * n_teams + team_g;
    const bool live_item = item < n_items;
    if (!live_item) item = n_items - 1;  // a team without an item shadows the last one with zero weights
    const int t = (int)(item / n_quads), quad = (int)(item - (long long)t * n_quads);
    const bool live_tile = live_item && 4 * quad + wave < n_tiles;
    const long long tile = 4 * quad + wave < n_tiles ? 4 * quad + wave : n_tiles - 1;
    const long long row = tile * 32 + j;
    const bool live = live_tile && row < B;
    const long long lrow = row < B ? row : B - 1;
    cfp cf = as_const(ws + L.coef + t * kCoefStride);
    const float c0 = live ? A.grad_rnd[lrow] * cf[CF_SIGMA] * cf[CF_DT] : 0.0f;

    int opq = 0;  // (read-only LDS tables through bases the compiler cannot see through: sdeh_bwdf2.hip)
    asm volatile("" : "+v"(opq));
    const float* __restrict__ WinT_s = WinT + opq;
    const float* __restrict__ Whid_s = Whid + opq;
    const float* __restrict__ Wout_s = Wout + opq;
    const float* __restrict__ bh_s = bh + opq;
    const float* __restrict__ bo_s = bo + opq;


    // ================================================================================ base pass: D_k = act'(Z_k), the clamp's mask
    // (the input layer as a transposed product on WinT: the k order (s, e) of the forward launch's matrix instructions)
    f32x16 D[3][2];
    unsigned mlo = 0u, mhi = 0u;
    {
      f32x16 cur[2], z[2], x[OTD];
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) x[ct] = load_cm(A.xs + (long long)t * d * B, (unsigned)lrow, ct);
      z[0] = load16(ws + L.emb + t * C + h * 16);  // timestep_embed(t) + input bias: the first addend, as in the forward launch
      z[1] = load16(ws + L.emb + t * C + (2 + h) * 16);
      chain_cols_n<RSW, NGI, OTD>(WinT_s + 4 * h * RSW + j, x, z);
      SDEH_FENCE();
      SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z[0], cur[0], D[0][0]); SDEH_FENCE(); act_both<ACT>(z[1], cur[1], D[0][1]););
      SDEH_FENCE();
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        z[0] = rows16(bh_s + l * 64 + 4 * h); z[1] = rows16(bh_s + l * 64 + 32 + 4 * h);
        fwd_rows<8, 2, 2>(Whid_s + l * 64 * RSW + j * RSW + 4 * h, RSW, cur, 8, z);
        SDEH_FENCE();
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z[0], cur[0], D[l + 1][0]); SDEH_FENCE(); act_both<ACT>(z[1], cur[1], D[l + 1][1]););
        SDEH_FENCE();
      }
      f32x16 nn[OTD];
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) nn[ct] = rows16(bo_s + 32 * ct + 4 * h);
      fwd_rows<8, 2, OTD>(Wout_s + j * RSW + 4 * h, RSW, cur, 8, nn);
      SDEH_FENCE();
      // bit (4 h' + rrow(q)) of m_ct: |nn| <= clip_model at that coordinate; both halves' bits meet through the half swap
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q < NQ) {
          mlo |= fabsf(nn[0][q]) <= A.clip_model ? (1u << (4 * h + rrow(q))) : 0u;
          if constexpr (OTD == 2) mhi |= fabsf(nn[1][q]) <= A.clip_model ? (1u << (4 * h + rrow(q))) : 0u;
        }
      }
      {
        auto r0 = __builtin_amdgcn_permlane32_swap(mlo, mlo, false, false);
        mlo = r0[0] | r0[1];
        if constexpr (OTD == 2) {
          auto r1 = __builtin_amdgcn_permlane32_swap(mhi, mhi, false, false);
          mhi = r1[0] | r1[1];
        }
      }
    }
    f32x16 dD[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int q = 0; q < 16; ++q) dD[k][R][q] = 0.0f;

    // ================================================================================ the coordinates
    for (int jc = 0; jc < d; ++jc) {
      int opj = 0;
      asm volatile("" : "+v"(opj));
      const float* __restrict__ win_p = WinT + opj + jc * RSW + 4 * h;   // W_in[:, jc] in accumulator order: rows16(win_p), rows16(win_p + 32)
      const float* __restrict__ wout_p = Wout + opj + jc * RSW + 4 * h;  // W_out[jc, :]
      const float* __restrict__ Wh = Whid + opj;
      const unsigned mbits = jc < 32 ? mlo : mhi;
      const float c = ((mbits >> (jc & 31)) & 1u) ? c0 : 0.0f;
      f32x16 F[2], G[2];
      {
        f32x16 P[2];
        const f32x16 w0 = rows16(win_p), w1 = rows16(win_p + 32);
#pragma unroll
        for (int q = 0; q < 16; ++q) { P[0][q] = D[0][0][q] * w0[q]; P[1][q] = D[0][1][q] * w1[q]; }
        SDEH_FENCE();
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int q = 0; q < 16; ++q) F[R][q] = 0.0f;
        fwd_rows<8, 2, 2>(Wh + j * RSW + 4 * h, RSW, P, 8, F);  // F = W_1 P
      }
      SDEH_FENCE();
      {
        f32x16 Rv[2];
        const f32x16 w0 = rows16(wout_p), w1 = rows16(wout_p + 32);
#pragma unroll
        for (int q = 0; q < 16; ++q) { Rv[0][q] = D[2][0][q] * w0[q]; Rv[1][q] = D[2][1][q] * w1[q]; }
        plane_put(Dme, 0, j, h, Rv[0]);  // (free since the previous stage's closing barrier)
        plane_put(Dme, 1, j, h, Rv[1]);
        SDEH_FENCE();
        chain_cols<RSW>(Wh + 64 * RSW + 4 * h * RSW + j, Rv, G);  // G = W_2^T R
      }
      SDEH_FENCE();
      // U = c D_1 G (over G), V = c D_1 F (over F)
#pragma unroll
      for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float cd = c * D[1][R][q], f = F[R][q], g = G[R][q];
          dD[1][R][q] = fmaf(c * f, g, dD[1][R][q]);
          G[R][q] = cd * g;
          F[R][q] = cd * f;
        }
      SDEH_FENCE();
      plane_put(Ame, 0, j, h, F[0]);
      plane_put(Ame, 1, j, h, F[1]);
      ws_barrier();
      {  // dR = W_2 V  next to  dW_2 tile (wR, wC) += R V^T
        f32x16 dR[2];
        float ds[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        stage_rows<16>(Wh + 64 * RSW + j * RSW + 4 * h, RSW, F, dR, planes, wR, wC, j, h, dw_hid[1], ds);
        const f32x16 w0 = rows16(wout_p), w1 = rows16(wout_p + 32);
        f32x16 qv[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          dD[2][0][q] = fmaf(dR[0][q], w0[q], dD[2][0][q]);
          dD[2][1][q] = fmaf(dR[1][q], w1[q], dD[2][1][q]);
          qv[0][q] = D[2][0][q] * dR[0][q];
          qv[1][q] = D[2][1][q] * dR[1][q];
        }
        SDEH_FENCE();
        const float rw = reduce_scatter32(qv, lane);
        if (live_tile) unsafeAtomicAdd(io_rec + (DPP + jc) * 64, rw);
      }
      SDEH_FENCE();
      {  // publish (U, P): P = D_0 . W_in[:, jc] once more (32 multiplications instead of 32 registers across the first stage)
        const f32x16 w0 = rows16(win_p), w1 = rows16(win_p + 32);
        f32x16 P[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) { P[0][q] = D[0][0][q] * w0[q]; P[1][q] = D[0][1][q] * w1[q]; }
        plane_put(Ame, 0, j, h, P[0]);
        plane_put(Ame, 1, j, h, P[1]);
      }
      plane_put(Dme, 0, j, h, G[0]);
      plane_put(Dme, 1, j, h, G[1]);
      ws_barrier();
      {  // dP = W_1^T U  next to  dW_1 tile (wR, wC) += U P^T
        f32x16 dP[2];
        float ds[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        stage_cols<8, RSW, 2, 2, 16>(Wh + 4 * h * RSW + j, G, 8, dP, planes, wR, wC, j, h, dw_hid[0], ds);
        const f32x16 w0 = rows16(win_p), w1 = rows16(win_p + 32);
        f32x16 qv[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          dD[0][0][q] = fmaf(dP[0][q], w0[q], dD[0][0][q]);
          dD[0][1][q] = fmaf(dP[1][q], w1[q], dD[0][1][q]);
          qv[0][q] = D[0][0][q] * dP[0][q];
          qv[1][q] = D[0][1][q] * dP[1][q];
        }
        SDEH_FENCE();
        const float col = reduce_scatter32(qv, lane);
        if (live_tile) unsafeAtomicAdd(io_rec + jc * 64, col);
      }
      SDEH_FENCE();
    }

    // ================================================================================ S_k = act''(Z_k) . d loss / d D_k
    {
      f32x16 cur[2], z[2];
      {
        f32x16 xr[OTD];
#pragma unroll
        for (int ct = 0; ct < OTD; ++ct) xr[ct] = load_cm(A.xs + (long long)t * d * B, (unsigned)lrow, ct);
        z[0] = load16(ws + L.emb + t * C + h * 16);
        z[1] = load16(ws + L.emb + t * C + (2 + h) * 16);
        chain_cols_n<RSW, NGI, OTD>(WinT_s + 4 * h * RSW + j, xr, z);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        SDEH_FENCE();
        float* __restrict__ sp = A.s_out + ((long long)k * 64 + 4 * h) * N + (long long)t * B + lrow;
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float s = act_grad2(z[R][q], act) * dD[k][R][q];
            if (live) sp[(long long)(32 * R + rrow(q)) * N] = s;
          }
        if (k < 2) {
          SDEH_ACT_SWITCH(act, ACT, act_tile<ACT>(z[0]); SDEH_FENCE(); act_tile<ACT>(z[1]););
          cur[0] = z[0]; cur[1] = z[1];
          z[0] = rows16(bh_s + k * 64 + 4 * h); z[1] = rows16(bh_s + k * 64 + 32 + 4 * h);
          fwd_rows<8, 2, 2>(Whid_s + k * 64 * RSW + j * RSW + 4 * h, RSW, cur, 8, z);
        }
      }
    }
  }

  float* __restrict__ rec = A.div_hid + (long long)team_g * 2 * 4096;
  store_tile(rec, 64, wR, wC, j, h, dw_hid[0]);
  store_tile(rec + 4096, 64, wR, wC, j, h, dw_hid[1]);
}

template <int OTD, int NQ>
static int launch_divf_t(const BwdfArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)(2 * 32 * OTD * bwdf::RSW + 2 * 64 * bwdf::RSW + 2 * 64 + 64 + 2 * 4 * bwdf::PLANE) * sizeof(float);
  if (lds_bytes > 160 * 1024) return SDEH_ERR_UNSUPPORTED;
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_divf_kernel<OTD, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((bridge_divf_kernel<OTD, NQ>), dim3((unsigned)a.n_slots), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The inference control's share of the FORWARD pass, row-parallel.  The SDE of a Bridge is driven by the generative control alone
// (losses/oc.py:176-217: sde_ctrl; the inference control enters the running cost, the Ito term and the divergence), so given the
// trajectory x_t and the control u_t of a plain launch (sdeh_simulate_fwd_train2u) everything the inference control adds to rnd,
//     drnd[t][i] = sigma div_x v dt + (u . v + |v|^2 / 2) dt + v . dB        (= (u + v)(u - (u - v) / 2) dt - u (u - u / 2) dt + ...)
// is independent per (step, trajectory): csrc/sdeh_bridge.hpp walks the steps with the d tangent passes in the loop (one wave per 32
// trajectories: 32 us per step at d = 10 whatever the batch), here every (step, 32 rows) is an item of its own.  J_jj from both ends
// as above (two products per coordinate, everything in registers, no LDS traffic but the weights, no barrier).  Writes u + v for the
// backward launches.  The waves of a workgroup share the weights and nothing else.
// ---------------------------------------------------------------------------------------------------------------------------
template <int OTD, int NQ>
__global__ __launch_bounds__(256) void bridge_rowsf_kernel(const BwdfArgs A) {
  using namespace bwdf2;
  static_assert(OTD == 1 || NQ == 16, "two coordinate tiles: all registers live");
  constexpr int DPP = 32 * OTD;
  constexpr int NGI = OTD == 2 ? 8 : NQ / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ WinT = lds;
  float* __restrict__ Whid = WinT + DPP * RSW;
  float* __restrict__ Wout = Whid + 2 * 64 * RSW;
  float* __restrict__ bh = Wout + DPP * RSW;
  float* __restrict__ bo = bh + 2 * 64;
  float* __restrict__ tabs = bo + 64;  // [2][64]: prior mean, inverse variance
  const WsLayout& L = A.lay;
  const float* __restrict__ ws = A.ws;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int d = A.d, T = A.n_steps;
  const long long B = A.batch;

  for (int idx = tid; idx < DPP * RSW; idx += 256) {
    const int row = idx / RSW, col = idx - row * RSW;
    WinT[idx] = (row < d && col < 64) ? A.w_in[col * d + row] : 0.0f;
    Wout[idx] = (row < d && col < 64) ? A.w_out[row * 64 + col] : 0.0f;
  }
  for (int idx = tid; idx < 2 * 64 * RSW; idx += 256) {
    const int l = idx / (64 * RSW), rem = idx - l * 64 * RSW, row = rem / RSW, col = rem - row * RSW;
    Whid[idx] = col < 64 ? A.w_hid[l][row * 64 + col] : 0.0f;
  }
  if (tid < 128) {
    bh[tid] = A.b_hid[tid >> 6][tid & 63];
    const int c = tid >> 6, cj = tid & 63;
    tabs[tid] = cj < d && cj < L.dp ? ws[L.dg[1] + 2 * cj + c] : 0.0f;
  }
  if (tid < 64) bo[tid] = tid < d ? A.b_out[tid] : 0.0f;
  __syncthreads();

#ifdef SDEH_BWDF2_ACT
  const int act = SDEH_BWDF2_ACT;
#else
  const int act = A.act;
#endif
  const int ctrl_kind = A.ctrl_kind, flags = A.flags;
  const bool has_score = ctrl_kind == SDEH_CTRL_LERP_PRIOR;
  const bool ito = (flags & SDEH_FLAG_ITO) != 0;
  const unsigned long long rng_off = philox_offset(A.offset, A.rng_dev);
  const int n_tiles = A.n_tiles;
  const long long n_items = (long long)n_tiles * T, n_waves = (long long)gridDim.x * 4;

  const unsigned Bu = (unsigned)B;
  auto cm_offsets = [&](unsigned col, int ct, unsigned (&off)[16]) {  // byte offsets of this lane's coordinates in a [d][B] plane
    const int cb = 32 * ct + 4 * h;
    unsigned lane_off = ((unsigned)(cb < d ? cb : 0) * Bu + col) * 4u;
    unsigned s1 = Bu * 4u;
    asm volatile("" : "+v"(lane_off), "+v"(s1));
    const unsigned s2 = s1 + s1, s3 = s2 + s1, s8 = s1 << 3;
    unsigned bg = lane_off;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool ok = q < NQ && cb + rrow(q) < d;
      const unsigned oq = (q & 3) == 0 ? bg : ((q & 3) == 1 ? bg + s1 : ((q & 3) == 2 ? bg + s2 : bg + s3));
      off[q] = ok ? oq : lane_off;
      if ((q & 3) == 3) bg += s8;
    }
  };

  for (long long item = (long long)blockIdx.x * 4 + wave; item < n_items; item += n_waves) {
    const int t = (int)(item / n_tiles);
    const long long tile = item - (long long)t * n_tiles;
    const long long row = tile * 32 + j;
    const bool live = row < B;
    const long long lrow = live ? row : B - 1;
    cfp cf = as_const(ws + L.coef + t * kCoefStride);
    const float sig = cf[CF_SIGMA], dt = cf[CF_DT], wl = cf[CF_W], sqdt = cf[CF_SQDT];
    const float gam0 = has_score ? as_const(ws + L.gam + t * L.g)[0] : 0.0f;
    const unsigned long long grow = (unsigned long long)(A.row_offset + lrow);

    int opq = 0;
    asm volatile("" : "+v"(opq));
    const float* __restrict__ WinT_s = WinT + opq;
    const float* __restrict__ Whid_s = Whid + opq;
    const float* __restrict__ Wout_s = Wout + opq;
    const float* __restrict__ bh_s = bh + opq;
    const float* __restrict__ bo_s = bo + opq;
    const float* __restrict__ tabs_s = tabs + opq;

    f32x16 x[OTD], D[3][2], nn[OTD];
    {
      const char* __restrict__ xb = reinterpret_cast<const char*>(A.xs + (long long)t * d * B);
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) {
        unsigned off[16];
        cm_offsets((unsigned)lrow, ct, off);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float val = q < NQ ? *reinterpret_cast<const float*>(xb + off[q]) : 0.0f;
          x[ct][q] = (q < NQ && 32 * ct + 4 * h + rrow(q) < d) ? val : 0.0f;
        }
      }
      f32x16 cur[2], z[2];
      z[0] = load16(ws + L.emb + t * C + h * 16);
      z[1] = load16(ws + L.emb + t * C + (2 + h) * 16);
      chain_cols_n<RSW, NGI, OTD>(WinT_s + 4 * h * RSW + j, x, z);
      SDEH_FENCE();
      SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z[0], cur[0], D[0][0]); SDEH_FENCE(); act_both<ACT>(z[1], cur[1], D[0][1]););
      SDEH_FENCE();
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        z[0] = rows16(bh_s + l * 64 + 4 * h); z[1] = rows16(bh_s + l * 64 + 32 + 4 * h);
        fwd_rows<8, 2, 2>(Whid_s + l * 64 * RSW + j * RSW + 4 * h, RSW, cur, 8, z);
        SDEH_FENCE();
        SDEH_ACT_SWITCH(act, ACT, act_both<ACT>(z[0], cur[0], D[l + 1][0]); SDEH_FENCE(); act_both<ACT>(z[1], cur[1], D[l + 1][1]););
        SDEH_FENCE();
      }
#pragma unroll
      for (int ct = 0; ct < OTD; ++ct) nn[ct] = rows16(bo_s + 32 * ct + 4 * h);
      fwd_rows<8, 2, OTD>(Wout_s + j * RSW + 4 * h, RSW, cur, 8, nn);
      SDEH_FENCE();
    }
    unsigned mlo = 0u, mhi = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < NQ) {
        mlo |= fabsf(nn[0][q]) <= A.clip_model ? (1u << (4 * h + rrow(q))) : 0u;
        if constexpr (OTD == 2) mhi |= fabsf(nn[1][q]) <= A.clip_model ? (1u << (4 * h + rrow(q))) : 0u;
      }
    }
    {
      auto r0 = __builtin_amdgcn_permlane32_swap(mlo, mlo, false, false);
      mlo = r0[0] | r0[1];
      if constexpr (OTD == 2) {
        auto r1 = __builtin_amdgcn_permlane32_swap(mhi, mhi, false, false);
        mhi = r1[0] | r1[1];
      }
    }

    // ---- the divergence: sum_j 1[|nn_j| <= clip_model] J_jj, J_jj = (W_2^T (D_2 . W_out[j, :]))^T D_1 (W_1 (D_0 . W_in[:, j])) ---------
    float div = 0.0f;
    for (int jc = 0; jc < d; ++jc) {
      int opj = 0;
      asm volatile("" : "+v"(opj));
      const float* __restrict__ win_p = WinT + opj + jc * RSW + 4 * h;
      const float* __restrict__ wout_p = Wout + opj + jc * RSW + 4 * h;
      const float* __restrict__ Wh = Whid + opj;
      f32x16 F[2], G[2];
      {
        f32x16 P[2];
        const f32x16 w0 = rows16(win_p), w1 = rows16(win_p + 32);
#pragma unroll
        for (int q = 0; q < 16; ++q) { P[0][q] = D[0][0][q] * w0[q]; P[1][q] = D[0][1][q] * w1[q]; }
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int q = 0; q < 16; ++q) F[R][q] = 0.0f;
        fwd_rows<8, 2, 2>(Wh + j * RSW + 4 * h, RSW, P, 8, F);
      }
      SDEH_FENCE();
      {
        f32x16 Rv[2];
        const f32x16 w0 = rows16(wout_p), w1 = rows16(wout_p + 32);
#pragma unroll
        for (int q = 0; q < 16; ++q) { Rv[0][q] = D[2][0][q] * w0[q]; Rv[1][q] = D[2][1][q] * w1[q]; }
        chain_cols<RSW>(Wh + 64 * RSW + 4 * h * RSW + j, Rv, G);
      }
      SDEH_FENCE();
      float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        p0 = fmaf(G[0][q] * D[1][0][q], F[0][q], p0);
        p1 = fmaf(G[1][q] * D[1][1][q], F[1][q], p1);
      }
      const float djj = sum_xor32(p0 + p1);
      const unsigned mbits = jc < 32 ? mlo : mhi;
      div += ((mbits >> (jc & 31)) & 1u) ? djj : 0.0f;
      SDEH_FENCE();
    }

    // ---- v, the score part of the divergence, the cost and Ito terms; u + v ----------------------------------------------------------
    float cost = 0.0f, itosum = 0.0f, dsum = 0.0f;  // this lane's coordinates (the halves meet below)
    const float mult = sig * A.scale_score;
#pragma unroll
    for (int ct = 0; ct < OTD; ++ct) {
      const int cb = 32 * ct + 4 * h;
      unsigned off[16];
      cm_offsets((unsigned)lrow, ct, off);
      const char* __restrict__ ub = reinterpret_cast<const char*>(A.u_in + (long long)t * d * B);
      char* __restrict__ gb = reinterpret_cast<char*>(A.gp_out + (long long)t * d * B);
      f32x16 uu;
#pragma unroll
      for (int q = 0; q < 16; ++q) uu[q] = q < NQ ? *reinterpret_cast<const float*>(ub + off[q]) : 0.0f;
      float n[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) n[q] = 0.0f;
      if (ito) {
        if (A.noise != nullptr) {
          const float* __restrict__ rowp = A.noise + ((long long)t * B + lrow) * d;
#pragma unroll
          for (int q = 0; q < NQ; ++q) n[q] = cb + rrow(q) < d ? rowp[cb + rrow(q)] : 0.0f;
        } else {
#pragma unroll
          for (int g4 = 0; g4 < NQ / 4; ++g4) {
            float n4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cb + 8 * g4 < d) box_muller4(philox_block(A.seed, rng_off, grow, t, (cb + 8 * g4) >> 2), n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) n[4 * g4 + e] = n4[e];
            SDEH_FENCE();
          }
        }
      }
      const f32x16 pmu = rows16(tabs_s + cb), pis = rows16(tabs_s + 64 + cb);
      f32x16 gamv;
      if (!has_score) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) gamv[q] = 0.0f;
      } else if (A.g == 1) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) gamv[q] = gam0;
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) gamv[q] = ws[L.gam + t * L.g + min(cb + rrow(q), L.g - 1)];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool in = cb + rrow(q) < d;
        const float sc = (1.0f - wl) * ((pmu[q] - x[ct][q]) * pis[q]);
        const bool inside = sc >= -A.clip_score && sc <= A.clip_score;
        float v = clipf(nn[ct][q], A.clip_model);
        if (has_score) {
          v += sig * ((A.scale_score * clipf(sc, A.clip_score)) * gamv[q]);
          const float dsc = inside ? -((1.0f - wl) * pis[q]) : 0.0f;
          dsum = fmaf(mult * gamv[q], in ? dsc : 0.0f, dsum);
        }
        v = in ? v : 0.0f;
        const float u = in ? uu[q] : 0.0f;
        const float gpv = u + v;
        cost = fmaf(v, u + 0.5f * v, cost);  // (u + v)(u - (u - v) / 2) - u (u - u / 2)
        itosum = fmaf(v, n[q], itosum);
        if (live && in) *reinterpret_cast<float*>(gb + off[q]) = gpv;
      }
    }
    cost = sum_xor32(cost);
    itosum = sum_xor32(itosum);
    dsum = sum_xor32(dsum);
    float dr = sig * (div + dsum) * dt;
    dr = fmaf(cost, dt, dr);
    if (ito) dr = fmaf(itosum, sqdt, dr);
    if (live && h == 0) A.drnd_out[(long long)t * B + row] = dr;
  }
}

template <int OTD, int NQ>
static int launch_rowsf_t(const BwdfArgs& a, hipStream_t stream) {
  const size_t lds_bytes = (size_t)(2 * 32 * OTD * bwdf::RSW + 2 * 64 * bwdf::RSW + 2 * 64 + 64 + 128) * sizeof(float);
  static bool attr_done[kMaxDevices] = {};
  bool& attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bridge_rowsf_kernel<OTD, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return SDEH_ERR_HIP;
    attr_set = true;
  }
  const long long items = (long long)a.n_tiles * a.n_steps, blocks = (items + 3) / 4;
  hipLaunchKernelGGL((bridge_rowsf_kernel<OTD, NQ>), dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

int launch_bridge_rowsf(const BwdfArgs& a, hipStream_t stream) {
  if (!bridge_divf_fits(a.d, a.n_hidden) || a.u_in == nullptr || a.gp_out == nullptr || a.drnd_out == nullptr) return SDEH_ERR_UNSUPPORTED;
  if (a.d <= 8) return launch_rowsf_t<1, 4>(a, stream);
  if (a.d <= 16) return launch_rowsf_t<1, 8>(a, stream);
  if (a.d <= 32) return launch_rowsf_t<1, 16>(a, stream);
  return launch_rowsf_t<2, 16>(a, stream);
}

__global__ __launch_bounds__(256) void divf_zero_kernel(float* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0.0f;
}

// the per-wave records the divergence kernel adds to (a kernel, not a memset node: csrc/sdeh_wide_bwd.hip's note on graph replays)
int launch_divf_zero(float* p, long long n, hipStream_t stream) {
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(divf_zero_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, p, n);
  return hipGetLastError() == hipSuccess ? SDEH_OK : SDEH_ERR_HIP;
}

bool bridge_divf_fits(int d, int n_hidden) { return d >= 1 && d <= 64 && n_hidden == 2; }

int launch_bridge_divf(const BwdfArgs& a, hipStream_t stream) {
  if (!bridge_divf_fits(a.d, a.n_hidden) || a.s_out == nullptr || a.div_hid == nullptr || a.div_io == nullptr) return SDEH_ERR_UNSUPPORTED;
  if (a.d <= 8) return launch_divf_t<1, 4>(a, stream);
  if (a.d <= 16) return launch_divf_t<1, 8>(a, stream);
  if (a.d <= 32) return launch_divf_t<1, 16>(a, stream);
  return launch_divf_t<2, 16>(a, stream);
}

}  // namespace sdeh
