// Internal (non-ABI) definitions shared by the prep kernel, the trajectory kernels and the host dispatcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdeh.h"

namespace sdeh {

// ---------------------------------------------------------------------------------------------------------
// MFMA register geometry.
//
// A wavefront owns 64 trajectories.  Elementwise / per-trajectory work runs in the "T layout": lane l holds
// every coordinate of trajectory l.  The dense layers run on v_mfma_f32_32x32x2_f32 in the "M layout": the
// wave's 64 trajectories form two 32-column tiles (A: trajectories 0..31, B: 32..63) and lane (j = l&31,
// h = l>>5) holds, for column j of each tile, the rows  rho(q,h) = (q&3) + 8*(q>>2) + 4*h  (q = accumulator
// register 0..15) of every 32-row tile -- the C/D fragment map of the 32x32 MFMA.  Because the B-operand map
// (k = l>>5, column = l&31) is the same as one accumulator register of the D map, a layer's activated output
// registers feed the next layer's MFMAs directly (the k order of the dot product is permuted, which the
// packed weight image accounts for).  T <-> M conversion is one v_permlane32_swap per register pair.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int rho(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }
// coordinate held by M-layout register r (r counts across 32-row tiles) in lane half h
__host__ __device__ constexpr int mdim(int r, int h) { return 32 * (r / 16) + rho(r % 16, h); }
__host__ __device__ constexpr int row_tiles(int n) { return (n + 31) / 32; }
// floats of one 32-trajectory tile of the pre-activation record (sdeh_traj_ws.hpp: ZRec): Lh + 1 layers, then the network output's tiles
__host__ __device__ constexpr int zrec_tile_floats(int n_hidden, int dim) { return (n_hidden + 1) * 2048 + 1024 * ((dim + 31) / 32); }
// number of M-layout registers needed to cover coordinates [0, n)
__host__ __device__ constexpr int mregs(int n) {
  int c = 0;
  for (int r = 0; r < 16 * row_tiles(n); ++r)
    if (mdim(r, 0) < n) ++c;
  return c;
}

constexpr int kCoefStride = 16;
enum CoefSlot {
  CF_S = 0, CF_T = 1, CF_DT = 2, CF_SQDT = 3, CF_SIGMA = 4, CF_DRIFT = 5, CF_DDIV = 6, CF_W = 7,
  CF_BETAK = 8, CF_ALPHAK = 9, CF_B2S2 = 10, CF_SBK = 11
};

// Workspace layout (float offsets from the plan's workspace base).  The first `lds_floats` floats are the
// "LDS image": every workgroup copies them verbatim into LDS.
// the out layer's 4 x 4 x 1 row groups are accumulated in passes (sdeh_traj_ws.hpp: ws_out4_stage); one operand image per pass
constexpr int kOut4Pass = 7;
__host__ __device__ constexpr int out4_passes(int dp) { return ((dp + 3) / 4 + kOut4Pass - 1) / kOut4Pass; }
__host__ __device__ constexpr int out4_pass_groups(int dp, int p) {
  const int g4 = (dp + 3) / 4, np = out4_passes(dp), base = g4 / np, extra = g4 % np;  // (13 -> 7 + 6)
  return base + (p < extra ? 1 : 0);
}
__host__ __device__ constexpr int out4_pass_first(int dp, int p) {
  int g = 0;
  for (int q = 0; q < p; ++q) g += out4_pass_groups(dp, q);
  return g;
}
__host__ __device__ constexpr int out4_pass_floats(int dp, int p) { return ((33 * out4_pass_groups(dp, p) + 31) / 32) * 256; }
__host__ __device__ constexpr int out4_floats(int dp) {
  int f = 0;
  for (int p = 0; p < out4_passes(dp); ++p) f += out4_pass_floats(dp, p);
  return f;
}
__host__ __device__ constexpr int out4_pass_offset(int dp, int p) {
  int f = 0;
  for (int q = 0; q < p; ++q) f += out4_pass_floats(dp, q);
  return f;
}


struct WsLayout {
  int dp, c, ot, otd, r_in, n_hidden, t_max, k_max, g;  // geometry (g = gamma row length: 1 or dp)
  int lds_floats;
  int w_in;       // [r_in][ot][64]
  int w_hid;      // n_hidden x [c/2][ot][64]
  int w_hid_stride;
  int w_out;      // [c/2][otd][64]
  int b_hid;      // n_hidden x [c]   (M order)
  int b_out;      // [otd*32]         (M order)
  // round 5: the out layer as v_mfma_f32_4x4x1_16b_f32 (ws_out4 in sdeh_traj_ws.hpp) for state dimensions that leave >= 8 of the 32-row
  // tile's rows empty and few row groups (d = 5 .. 16): G = ceil(dp / 4) row groups x 33 slots (32 accumulator registers of the activation + the
  // bias against B = 1) = 33 G instructions per column tile; instruction n = slot * G + g; CBSZ = 3 lets one operand register carry 8
  // instructions (ABID = n % 8): register n / 8, lane l: out_w[4 g + (l & 3)][32 (slot / 16) + rho(slot % 16, l >> 5)]  (slot 32: out_b on
  // the lanes l < 32, zero above); per PASS of at most kOut4Pass row groups one such image, g counted within the pass.  -1 when not packed
  // (training launches, images beyond LDS).
  int w_out4;
  int wt_out;     // backward only: transposed out_layer  [r_in][ot][64];  -1 when not packed
  int wt_hid;     // backward only: transposed hidden layers, n_hidden x [c/2][ot][64]
  int wt_in;      // backward only (BPTT): transposed input_embed  [c/2][otd][64]
  // global tables
  int coef;       // [T][16]
  int emb;        // [T][c]  (M order; FourierMLP.timestep_embed(t) + input_embed.bias)
  int gam;        // [T][g]  clip(score_model(t), clip_model)
  int out_cnt;    // sdeh_integrate: [T+1] ints, out_cnt[i] = number of output times emitted by steps < i
  // Bridge (forward-mode tangents of the inference network): column j of input_embed.weight and row j of out_layer.weight,
  // [dp][C] each in accumulator (M) order; -1 when not packed
  int tan_in, tan_out;
  // GMM tables: rows of `gmm_row` floats (dp rounded up to even pairs, so rows are float4-aligned).  They live
  // inside the LDS image when they fit (gmm_lds = 1: broadcast ds_read_b128, deep VGPR prefetch), else in the
  // global part of the workspace (scalar loads).
  int gmm_lds, gmm_row, gmm_rows;  // gmm_rows: table rows = K rounded up to a multiple of 8 (padding: logit -inf)
  int gmm_vec;    // shared-scale form: four vectors of 4*ceil(dp/4) floats: 1/(sqrt2 s_d), 1/s_d^2, mu_0d/(sqrt2 s_d), mu_0d/s_d^2
  int gmm_lg;     // [K][gmm_row/2][2]  (mu, 1/(2 sigma^2))
  int gmm_sc;     // [K][gmm_row/2][2]  (mu/sigma^2, 1/sigma^2)
  int gmm_c;      // [K]  log_softmax(log w)_k - sum_d (log sigma_kd + 0.5 log 2pi)
  // gmm_lds == 3 (round 5, shared scale, full tables): the mixture's two contractions run on the matrix pipe inside the V wave
  // (v_mfma_f32_4x4x1_16b_f32, gmm_mm in sdeh_traj_ws.hpp).  The LDS image holds their A-operand images instead of the tables
  // above, which then live in the global part of the workspace (terminal log-density through the scalar cache):
  //   gmm_mm1 [ceil(dp K4 / 64)][64 lanes][4]: register v = 4 q + e, lane 4 b + i: mu[4 g + i][d] / sigma_d^2, instruction n = 16 v + b = d K4 + g
  //   gmm_mm2 [ceil(K D4 / 64)][64 lanes][4]:  register v, lane 4 b + i: mu[k][4 g + i] / sigma^2,     instruction n = 16 v + b = k D4 + g
  //   gmm_cc  [K rows]: gmm_c[k] - sum_d mu_kd^2 / (2 sigma_d^2)   (-inf for padding rows);  K4 = rows / 4, D4 = ceil(dp / 4)
  // gmm_lds == 4: per-component scales, two tables per stream: mm1 instruction n = (2 d + t) K4 + g with t = 0: -1 / (2 sigma_kd^2) (against x_d^2),
  //   t = 1: mu_kd / sigma_kd^2 (against x_d);  mm2 instruction n = (2 k + t) D4 + g with t = 0: mu / sigma^2 (-> P), t = 1: 1 / sigma^2 (-> Q)
  int gmm_mm1, gmm_mm2, gmm_cc;
  int dg[3];      // diag-gauss tables for target / prior / second: [dp][2] (mu, 1/sigma^2) then 1 float const
  int total;
  // Wide networks (sdeh_wide.hip: C in {128, 256}, d <= 256): no LDS image; the A operands stream from this workspace (L2), packed
  // in NATURAL k order as float4 groups of four k-steps:  pk[((S * n_tiles + t) * 64 + lane) * 4 + e] = W[32 t + (lane & 31)][8 S + 2 e + (lane >> 5)]
  //   w_in  [dp8/8][ot][64] float4 (k = coordinate, zero beyond d)      w_hid  n_hidden x [c/8][ot][64] float4
  //   w_out [c/8][otd][64] float4 (rows = coordinates, zero beyond d)   b_hid / b_out / emb: accumulator (M) order as above
  // Bridge (inference network): tan_in / tan_out [d][c] in the B-operand order of a k-group: idx(ch) = (ch / 8) * 8 + (ch & 1) * 4 + ((ch & 7) >> 1)
  //   (column j of input_embed.weight / row j of out_layer.weight), and wt_hid = the hidden layers transposed, packed like w_hid.
  int wide;       // 1: the layout above
  int dp8;        // d rounded up to a multiple of 8 (k extent of the input layer)
};

struct DensArgs {
  int kind, n_comp;
  float lnc, p0, p1;
};

struct TrajArgs {
  const float* ws;
  WsLayout lay;
  const float* x0;
  const float* noise;
  float* xT;
  float* rnd;
  float* xs;
  long long batch;
  long long row_offset;
  int n_steps, d;
  int loss_kind, ctrl_kind, flags, act;
  float clip_model, clip_score, scale_score, clip_target;
  float exp_sigma;
  DensArgs target, prior, second;
  unsigned long long seed, offset;
  const unsigned long long* rng_dev;  // optional device-resident addend of `offset` (SdehProblem.rng_offset_dev)
  // sdeh_integrate only
  int int_kind, n_out;
  const float* ts_out;
  // Bridge only: the inference control's workspace region (own layout) and attributes
  const float* ws2;
  WsLayout lay2;
  int inf_kind, inf_act;
  float inf_clip_model, inf_clip_score, inf_scale_score;
  float* gp;  // [T, B, d] or null: u + v per step (needed by the backward pass of the inference network)
  int flag_sync;  // wave-specialised kernel: pair-level LDS counters instead of workgroup barriers for the V <-> M hand-off
  int half;   // wave-specialised kernel: a group is 32 trajectories (one MFMA column tile) instead of 64 -- small batches
  int vout;   // wave-specialised kernel, d <= 4: the out layer runs on the V wave's vector pipe (the M wave publishes its last activations)
  int csplit; // Bridge kernel: waves of a workgroup that carry the SAME 32 trajectories and split the coordinates' tangent passes (1 | 4)
  const float* div_noise;  // [T, B, d] or null: Hutchinson probe vectors (training with div_estimator); null = exact divergence
  // training forward (sdeh_simulate_fwd_train): what the backward kernels would otherwise recompute
  float* zt_out;  // [(Lh+1), C, T*B] or null: pre-activations of every layer, coordinate-major
  float* nn_out;  // [T, B, d] or null: raw network output (before the clamp) per step
  // training forward for the fused backward (sdeh_simulate_fwd_train2)
  // (coordinate-major planes: consecutive trajectories at consecutive addresses -- coalesced for both kernels)
  float* xs_cm;   // [T+1, d, B] or null: the trajectory
  float* sc_out;  // [T, d, B] or null: combined score entering the control, before clip_score and gamma(t)
  float* u_out;   // [T, d, B] or null: the control u_t driving the SDE (Bridge training: the inference pass is row-parallel given x_t, u_t)
  float* tsc_out; // [d, B] or null: 1[|log rho(x_T)| <= clip_target] * target.score(x_T)  (d terminal target cost / d x_T, negated)
  // training forward that also keeps the network's pre-activations for the fused backward (sdeh_simulate_fwd_train3)
  float* zrec;    // [T][ceil(B / 32)]{[Lh + 1][16][32][4]; [ceil(d / 32)][8][32][4]} or null: the pre-activation record + raw network output (sdeh_traj_ws.hpp: ZRec)
  // sdeh_simulate_fwd_steps (wide kernels): this launch runs the steps [step0, step0 + n_steps) of the grid the tables were prepared for;
  // table rows, Philox counters and the rows of xs / gp are indexed by the GRID's step
  int step0;
  int seg_continue;        // != 0: not the first segment -- rnd continues from A.rnd, x_0's row of xs is not written
  const float* ext_score;  // SDEH_DENS_EXTERNAL: the target's score at x_t for the steps of this launch
  long long ext_stride;    // floats between consecutive steps of ext_score (0: one [B, d] buffer, one-step segments)
};

// sdeh_bridge_div_backward (sdeh_bridge.hpp): gradient of  sum_n w_i sigma dt div_x v(x_n)  w.r.t. the inference network
struct BridgeBwdArgs {
  const float* ws;   // region 1: per-step coefficients, prior table
  WsLayout lay;
  const float* ws2;  // region 2: the inference network (packed + transposed weights, tangent tables, gamma)
  WsLayout lay2;
  const float* xs;        // [T+1, B, d]
  const float* grad_rnd;  // [B]
  const float* zt;        // [(Lh+1), C, N]  pre-activations of the inference network (from sdeh_ctrl_backward)
  float* tz;              // [d][(Lh+1), C, N]  tangent pre-activations      d z_l / d x_j
  float* ta;              // [d][(Lh+1), C, N]  tangent activations          act'(z_l) d z_l / d x_j
  float* td;              // [d][(Lh+1), C, N]  adjoints of the tangent pre-activations
  float* d2;              // [(Lh+1), C, N]     adjoints of the base pre-activations (second-order path)
  float* cj;              // [d, N]             c_j = w_i sigma dt 1[|v_nn,j| <= clip_model]
  float* dgam;            // [g, N]             d / d gamma(t) of the score part of the divergence
  float* dx;              // [T, B, d] or null: d / d x_t of the divergence term is ADDED to this plane
  const float* eps;       // [T, B, d] or null: Hutchinson probe vectors (then tz/ta/td hold ONE tangent, cj = c0 eps_j m_j)
  long long batch;
  int n_steps, d, inf_kind, act;
  float clip_model, clip_score, scale_score;
};

// sdeh_bridge_div_backward_wide (sdeh_wide_bwd.hip): the divergence term's gradient on wide networks, fused
struct WideDivArgs {
  const float* ws;    // region 1: per-step coefficients, prior table
  WsLayout lay;
  const float* ws2;   // region 2: the inference network (packed + transposed weights, tangent tables, gamma, biases)
  WsLayout lay2;
  const float* xs;        // [T+1, B, d]
  const float* grad_rnd;  // [B]
  const float* zt;        // [(Lh+1), C, N] pre-activations of the inference network (sdeh_ctrl_backward_ex wrote them)
  float* d2;              // [(Lh+1), C, N] adjoints of the base pre-activations through the divergence (side 1 writes plane Lh, side 0 the rest)
  float* dgam;            // [g, N] (side 0)
  float* dx;              // [T, B, d] or null: d loss / d x_t, the divergence term's share ADDED (side 0; method kl)
  float* xpart;           // [grid][C][C] partial of dX
  float* cpart;           // [grid][d][C] partial of d L / d col_q  (zeroed by the kernel)
  float* spart;           // [grid][d][C] partial of d L / d W_out (Lh = 1 only, side 0; zeroed by the kernel)
  long long batch;
  int n_steps, d, inf_kind, act, side;
  float clip_model, clip_score, scale_score;
};

struct BwdArgs {
  const float* ws;
  WsLayout lay;
  const float* xs;        // [T+1, B, d]
  const float* noise;     // [T, B, d] or null (Philox replay)
  const float* grad_rnd;  // [B]
  const float* gextra;    // [T, B, d] or null: additional d rnd_i / d u_{i,t} = cdt * gextra (Bridge: u + v for the inference net)
  const float* cost_ctrl; // [T, B, d] or null (BPTT): the control entering the running cost instead of u (Bridge: u + v)
  const float* lam_extra; // [T, B, d] or null (BPTT): added to d loss / d x_t (Bridge: the inference network's share)
  float* dx;              // [T, B, d] or null (row-parallel): d loss / d x_t through this control, written per row
  float* zt;              // [(Lh+1), C, N]
  float* dt;              // [(Lh+1), C, N]
  float* dout;            // [d, N]
  float* dgam;            // [g, N]
  long long batch, row_offset;
  int n_steps, d;
  int loss_kind, ctrl_kind, flags, act;
  float clip_model, clip_score, scale_score, clip_target;
  DensArgs target;
  unsigned long long seed, offset;
  const unsigned long long* rng_dev;
  const float* nn_in;  // [T, B, d] or null: with it, `zt` already holds the forward launch's pre-activations and is only read
  float* xt_out;       // [d, N] or null (wide plans): x_t coordinate-major, written next to the planes
  const float* sc_in;      // [T, B, d] or null (wide plans, mixture targets): the score entering the control, from the forward launch
  const float* tscore_in;  // [B, d] or null (wide plans, mixture targets, BPTT): 1[|log rho(x_T)| <= clip_target] target.score(x_T)
};

// sdeh_ctrl_backward_fused (sdeh_bwdf.hip): back-propagation + weight gradients in one kernel
struct BwdfArgs {
  const float* ws;  // prep'd workspace: per-step coefficients, time embedding (accumulator order), gamma table, Gaussian tables
  WsLayout lay;
  const float* w_in;      // [64, d]      raw parameters of the FourierMLP (models/mlp.py:85-112)
  const float* w_hid[3];  // [64, 64]  (n_hidden of them)
  const float* b_hid[3];  // [64]
  const float* w_out;     // [d, 64]
  const float* b_out;     // [d]
  const float* xs;        // [T+1, d, B]  (coordinate-major, as sdeh_simulate_fwd_train2 writes it)
  const float* noise;     // [T, B, d] or null (Philox replay)
  const float* grad_rnd;  // [B]
  const float* sc;        // [T, d, B] or null (ClippedCtrl)
  const float* tscore;    // [d, B] or null
  float* wpart;           // [n_slots][wsize]
  float* epart;           // [n_tiles][T][64]
  float* gpart;           // [n_tiles][T][gw]
  long long batch, row_offset;
  int n_steps, d, n_kg;   // n_kg: k-groups (of 4 accumulator registers) covering the d coordinates
  int loss_kind, ctrl_kind, flags, act, g, gw;
  float clip_model, clip_score, scale_score;
  DensArgs target;
  unsigned long long seed, offset;
  const unsigned long long* rng_dev;
  int n_tiles, n_slots, wsize;
  int n_hidden;           // 1 .. 3
  // back-propagation through time as a SCAN (sdeh_bwdf2.hip, d <= 4): a row-parallel pass stores the raw network output and its
  // d x d Jacobian per (step, trajectory), a scan over the steps forms the upstream gradient of the control, a row-parallel pass
  // (the lv form of the backward) turns it into parameter gradients
  float* nn_out;          // [T, d, B] or null
  float* jac_out;         // [T, d, d, B] or null: d nn_k / d x_i at [t][k][i][row]
  float* gq_out;          // [T, d, B] (scan kernel): d loss / d u_t
  const float* gq_in;     // [T, d, B] or null (row-parallel kernel): use this upstream gradient instead of w_i dB
  // Bridge, inference network (sdeh_bridgef.hip + the row-parallel kernel of sdeh_bwdf2.hip): the running cost's u + v and the
  // divergence term  w_i sigma dt sum_j 1[|nn_j| <= clip_model] J_jj
  const float* gextra;    // [T, d, B] or null: u + v (d rnd / d v = (u + v) dt + dB)
  const float* u_in;      // [T, d, B] (bridge_rowsf_kernel): the generative control of the forward launch
  float* gp_out;          // [T, d, B] (bridge_rowsf_kernel): u + v
  float* drnd_out;        // [T, B]    (bridge_rowsf_kernel): what the inference control adds to rnd at step t
  float* dx_out;          // [T, d, B] or null (Bridge row-parallel kernel, method kl): d loss / d x_t of the inference network's terms
  const float* cost_in;   // [T, d, B] or null (through time): the control entering the running cost, u + v, instead of u - reference control
  const float* lam_in;    // [T, d, B] or null (through time): added to the adjoint at every step (the plane above)
  float* s_out;           // [3, 64, T * B] (divergence kernel): act''(Z_k) . d loss / d act'(Z_k), the term the base chain adds at layer k
  const float* s_in;      // the same planes, read by the row-parallel kernel
  float* div_hid;         // [n_slots][2][64][64]: the divergence term's direct gradient of the two hidden weights, per team
  float* div_io;          // [n_slots * 4][2][32 OTD][64] (zeroed by the caller): per wave, columns of input_embed.weight | rows of out_layer.weight
  // the pre-activation record and the raw network output of sdeh_simulate_fwd_train3: the kernels that take them do not re-evaluate the network
  const float* zrec;      // [T][ceil(B / 32)]{[Lh + 1][16][32][4]; [ceil(d / 32)][8][32][4]} or null
};
int launch_bridge_divf(const BwdfArgs& a, hipStream_t stream);  // divergence term of a 64-channel Bridge, two hidden layers (sdeh_bridgef.hip)
bool bridge_divf_fits(int d, int n_hidden);
int launch_divf_zero(float* p, long long n, hipStream_t stream);
int launch_bridge_rowsf(const BwdfArgs& a, hipStream_t stream);  // v, div_x v and their share of rnd for every (step, trajectory), row-parallel
int launch_bwdf2_bridge(const BwdfArgs& a, hipStream_t stream);  // the row-parallel backward with gextra / s_in / the in-kernel prior score
int launch_bwdf(const BwdfArgs& a, hipStream_t stream);
int bwdf_wsize(int d, int n_hidden);                       // floats of one team's partial-gradient record
bool bwdf_fits(int d, int n_hidden);                        // compiled for this shape and its LDS image fits
int bwdf_slots(long long batch, int n_steps, bool bptt);  // teams (partial records) a launch uses
// the same backward on tiles of 16 trajectories (sdeh_bwdf16.hip): back-propagation through time at small batches
int launch_bwdf16(const BwdfArgs& a, hipStream_t stream);
int bwdf_tile(long long batch, bool bptt, int act);  // 16 or 32: trajectories per team of the launch that serves this problem
int bwdf16_slots(long long batch);
// one deterministic sum over chunks (launch_partial_sums_multi, sdeh_wgrad.hip): out[e] = sum_k in[k][e]; mid: ceil(n_chunks / 32) x width floats
struct SumSeg { const float* in; float* mid; float* out; long long n_chunks, width; };
struct SumJob { SumSeg s[5]; int first_block[6]; int n; };
int launch_partial_sums_multi(SumJob job, hipStream_t stream);
int bwdf16_waves(long long batch);  // wavefronts per 16-trajectory team (4; plan option SDEH_BWD_WAVES)
// the same backward with trajectory-split teams (sdeh_bwdf2.hip): a wave owns 32 trajectories and all channels, no barrier in the chain
int launch_bwdf2(const BwdfArgs& a, hipStream_t stream);
bool bwdf2_fits(int d, int n_hidden);                        // one or two hidden layers
int bwdf2_slots(long long batch, int n_steps, bool bptt);  // teams of four 32-trajectory tiles
int launch_bwdf2_jac(const BwdfArgs& a, hipStream_t stream);   // row-parallel: nn_out, jac_out (d <= 4, two hidden layers)
int launch_bwdf2_scan(const BwdfArgs& a, hipStream_t stream);  // the adjoint recursion over the steps: gq_out
bool bwdf2_scan_fits(int d, int n_hidden);

// the NICE flow target (sdeh_nice.hip): log-density and score of a batch of rows
long long nice_work_floats(const SdehNice& nn, long long batch, bool want_score);
int launch_nice_eval(const SdehNice& nn, const float* x, long long batch, float* score, float* logp, float* work, hipStream_t stream);

// effective Philox offset of a launch: by-value part + the optional device-resident counter (hipGraph replays)
__device__ __forceinline__ unsigned long long philox_offset(unsigned long long offset, const unsigned long long* dev) {
  return dev != nullptr ? offset + *dev : offset;
}

struct PrepArgs {
  float* ws;
  WsLayout lay;
  SdehProblem prob;  // by value: holds the device pointers of all parameters
  const float* ts;
  int n_steps;
  const float* ts_out;  // sdeh_integrate: output times (null otherwise)
  int n_out;
  float eps;
};

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Key = 64-bit seed; counter = (row, step, block, offset).
// ---------------------------------------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x;
    const uint64_t p1 = (uint64_t)M1 * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.y = (uint32_t)p1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// sdeh_sinkhorn sweeps (sdeh_sinkhorn.hpp): rows P[np,d] against the cloud Q[nq,d]
struct SinkArgs {
  const float* P;
  const float* Q;
  const float* pot_q;  // [nq]
  const float* pot_p;  // [np]  (distance sweep only)
  float* part_m;       // [splits][np]
  float* part_s;       // [splits][np];  distance sweep: [row blocks]
  long long* corr;     // [np] or null
  const int* done;     // device flag: non-zero = converged, sweeps become no-ops
  long long np, nq;
  int d, pnorm;
  float inv_eps;
};

// Kernel-mode options of the plan whose entry point is executing on this thread (include/sdeh.h: sdeh_plan_set_option).  The launchers
// ask plan_opt(key) where they used to call getenv: nullptr = automatic, else the value string.  Nothing on the launch path reads the
// environment (sdeh_plan_create copies it once).
enum OptKey {
  OPT_LEGACY, OPT_GENERIC_ONLY, OPT_WS_GROUPS, OPT_WS_QUAD, OPT_WS_VOUT, OPT_WS_BARRIER, OPT_BWD_PLANES, OPT_BWD_TILE, OPT_BWD_WAVES,
  OPT_BWD_V1, OPT_BWD_V2, OPT_BWD_NO_VIO, OPT_BWD_SCAN, OPT_BWD_ZREC, OPT_BRIDGE_TILES, OPT_BRIDGE_SPLIT, OPT_WIDE_CT, OPT_WIDE_SPLIT, OPT_GMM_MM, OPT_WS_OUT4, OPT_COUNT
};
struct PlanOptions {
  char v[OPT_COUNT][8];
};
const char* plan_opt(OptKey key);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: the launchers remember it per device ordinal
constexpr int kMaxDevices = 64;
inline int current_device_slot() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 0 && dev < kMaxDevices ? dev : 0;
}

// launchers implemented in the per-DP translation units (sdeh_traj_inst.hip)
typedef int (*TrajLauncher)(const TrajArgs& a, hipStream_t stream);
typedef int (*SinkLauncher)(const SinkArgs& a, int mode, int splits, hipStream_t stream);

}  // namespace sdeh
