/*
 * sdeh.h -- C ABI of libsdeh.so, the MI355X (gfx950) Euler-Maruyama trajectory engine.
 *
 * This is the drop-in boundary for ONE path of juliusberner/sde_sampler: the `simulate()` loops of the
 * optimal-control losses.  The reference has no FFI (it is pure Python; its plugin mechanism is Hydra
 * `_target_` class paths), so every entry point below cites the reference *Python* interface
 * it replaces; INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All `const float*` / `float*` tensor arguments are
 *     DEVICE pointers to contiguous row-major fp32 (the reference's layout: x[B,d], rnd[B,1], xs[T+1,B,d],
 *     nn.Linear.weight[out,in]).  Struct fields that are pointers are device pointers as well; the structs
 *     themselves live in host memory.
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*; NULL = the null stream), never
 *     synchronises, never allocates after sdeh_plan_create, never retains caller buffers.  (One exception: a plan for
 *     channels >= 128 that evaluates a Bridge grows a per-trajectory scratch buffer the first time it sees a larger batch.)
 *   - return value: 0 on success, a negative SdehStatus otherwise (never throws across the ABI);
 *     sdeh_last_error() returns a thread-local message for the last failure.
 *   - parameters are re-read from the given pointers on EVERY call (the reference swaps EMA weights in and
 *     out around evaluation, solver/base.py:335-346, and mutates clip values in place, base.py:586-597).
 */
#ifndef SDEH_H_
#define SDEH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v4 (round 3): sdeh_ctrl_backward_ex gained xt_out / sc_in / tscore_in; wide-network training entry points
 * (sdeh_bridge_div_backward_wide[_sizes]); sdeh_simulate_fwd_aux2.
 * v5 (round 4): sdeh_plan_set_option / sdeh_plan_reserve / SdehPlanDesc.max_batch (kernel-mode options and batch-dependent scratch live in
 * the plan; nothing on the launch path reads the environment or allocates); the 64-channel Bridge as a split -- sdeh_simulate_fwd_train2u,
 * sdeh_bridge_inference_fwd, sdeh_bridge_backward_fused[_sizes], sdeh_ctrl_backward_fused_ex.
 * v6 (round 5): the training forward keeps the network's pre-activations and the fused backward reads them instead of re-evaluating the
 * network -- sdeh_zrec_floats, sdeh_simulate_fwd_train3, sdeh_ctrl_backward_fused_z; plan option SDEH_BWD_ZREC.
 * v7 (round 6): BASELINE configs[4]'s target -- the NICE flow (distr/nice.py) as a row-parallel log-density + score evaluation
 * (SdehNice, sdeh_nice_work_floats, sdeh_nice_eval), a target whose score is SUPPLIED per step (SDEH_DENS_EXTERNAL) and the wide
 * kernels run in segments of the time grid around it (sdeh_simulate_fwd_steps). */
#define SDEH_ABI_VERSION 7
#define SDEH_MAX_HIDDEN 8 /* max entries of any nn.ModuleList of hidden layers */

typedef enum {
  SDEH_OK = 0,
  SDEH_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unknown enum) */
  SDEH_ERR_UNSUPPORTED = -2, /* valid in the reference but not built into this engine (see message) */
  SDEH_ERR_HIP = -3,         /* a HIP runtime call failed */
  SDEH_ERR_CAPACITY = -4     /* problem larger than what the plan was created for */
} SdehStatus;

/* losses/oc.py: TimeReversalLoss (140-278), ReferenceSDELoss (281-391), ExponentialIntegratorSDELoss (394-505) */
typedef enum { SDEH_LOSS_TIME_REVERSAL = 0, SDEH_LOSS_REFERENCE_SDE = 1, SDEH_LOSS_EXPONENTIAL = 2 } SdehLossKind;
/* models/reparam.py: ClippedCtrl 13-36, ScoreCtrl 39-83, LerpCtrl 113-162, LerpTargetCtrl 184-200, LerpPriorCtrl 165-181 */
typedef enum {
  SDEH_CTRL_CLIPPED = 0, SDEH_CTRL_SCORE = 1, SDEH_CTRL_LERP = 2, SDEH_CTRL_LERP_TARGET = 3, SDEH_CTRL_LERP_PRIOR = 4,
  SDEH_CTRL_NONE = 5 /* sdeh_integrate only: ControlledSDE(ctrl=None) / a bare OU / LangevinSDE -- no network */
} SdehCtrlKind;
/* eq/sdes.py: VP 191-269; ConstOU 125-172 (ScaledBM 175-188 == ConstOU with drift_coeff 0); NONE for DDS */
typedef enum { SDEH_SDE_NONE = 0, SDEH_SDE_VP = 1, SDEH_SDE_CONST_OU = 2 } SdehSdeKind;
/* distr/gauss.py GMM 66-155 / Gauss 158-183 / IsotropicGauss 186-242 / delta.py; double_well.py 14-100,103-193; funnel.py */
typedef enum {
  SDEH_DENS_NONE = 0,
  SDEH_DENS_GMM = 1,        /* loc[K,d], scale[K,d], mixture_weights[K] (unnormalised, as given to Categorical) */
  SDEH_DENS_DIAG_GAUSS = 2, /* loc[d], scale[d]   (Gauss / IsotropicGauss / Delta / sde.marginal_distr) */
  SDEH_DENS_MULTI_WELL = 3, /* n_components double wells then (dim-n_components) unit Gaussians at `shift`;
                               DoubleWell == {dim 1, n_components 1} */
  SDEH_DENS_FUNNEL = 4,     /* p0 = variance of the first coordinate (default dim-1) */
  SDEH_DENS_EXTERNAL = 5    /* v7, wide plans through sdeh_simulate_fwd_steps only: the target's score at x_t is SUPPLIED by the caller
                               per step (sdeh_simulate_fwd_steps: `ext_score`), its terminal log-density is subtracted by the caller
                               after the last segment.  What the NICE target of BASELINE configs[4] runs on (sdeh_nice_eval evaluates
                               the flow's score between the segments) */
} SdehDensityKind;
/* activation callable injected by conf/model/base/fouriermlp.yaml:5-6 (default torch.nn.GELU, exact erf) */
typedef enum { SDEH_ACT_GELU_ERF = 0, SDEH_ACT_SILU = 1, SDEH_ACT_RELU = 2,
               SDEH_ACT_IDENTITY = 3 /* sdeh_weight_grad only */ } SdehActivation;

/* flags of simulate(): losses/oc.py:156-166 (train, compute_ito_int, change_sde_ctrl, return_traj) */
enum {
  SDEH_FLAG_TRAIN = 1,           /* TimeReversalLoss only: skip `rnd -= sde.drift_div_int` (oc.py:210-211) */
  SDEH_FLAG_ITO = 2,             /* add the Ito integral  sum(u . dB)            (oc.py:218-219,330-331,440-443) */
  SDEH_FLAG_CHANGE_SDE_CTRL = 4, /* log-variance form of the running cost        (oc.py:204-206,319-321,418-422) */
  SDEH_FLAG_INIT_LOGP = 8,       /* rnd starts at second.log_prob(x0) instead of 0 (oc.py:168-172) */
  SDEH_FLAG_TERMINAL_TARGET = 16,/* subtract clip(target.unnorm_log_prob(x_T), clip_target) in-kernel (oc.py:225) */
  SDEH_FLAG_TERMINAL_SECOND = 32,/* add second.log_prob(x_T) in-kernel (oc.py:337,449-450) */
  SDEH_FLAG_REFERENCE_CTRL = 64, /* ReferenceSDELoss.reference_ctrl = sigma(t) * prior.score(x) (solver/oc.py:305-306) */
  SDEH_FLAG_INFERENCE_SDE = 128, /* sdeh_integrate only: the sde was built with generative=False (eq/sdes.py:76-77: sign -1,
                                    VP schedule min->max) and ControlledSDE evaluates its ctrl at terminal_t - t (sdes.py:301-303) */
  SDEH_FLAG_INFERENCE_CTRL = 256,/* TimeReversalLoss.inference_ctrl is set (Bridge, losses/oc.py:189-202): SdehProblem.inference
                                    describes it; rnd += sigma div_x(v) dt with the EXACT divergence, costs on u + v / u - v */
  /* Backward passes only (sdeh_ctrl_backward[_ex] in back-propagation-through-time mode): which score terms of the control are
   * CONSTANTS of the autograd graph in the reference.  generative_ctrl.detach_score = True detaches x in front of every score
   * (models/reparam.py:58,134,169,188); with detach_score = False a score obtained by autograd (Distribution.score with
   * create_graph=False, distr/base.py:130-137: GMM, also a one-component GMM that the engine evaluates as a Gaussian) is still
   * a constant, closed-form scores are differentiated.  Mixture targets are always treated as constants by the kernel. */
  SDEH_FLAG_DETACH_SCORE = 512,       /* target AND prior score terms of the control carry no d/dx */
  SDEH_FLAG_TARGET_SCORE_CONST = 1024 /* only the target score term carries no d/dx */
};

/* GMM only: scale[k,d] == scale[0,d] for every component k (true for every named mixture of the reference,
 * distr/gauss.py:14-63).  Selects a cheaper evaluation (one table word per (k,d)); results are undefined if the
 * promise is false -- the Python binding checks the tensor once per (tensor, version) before setting it. */
#define SDEH_DENS_FLAG_SHARED_SCALE 1
/* GMM (v6, round 5): the caller vouches that the component logits may be evaluated in PRODUCT form (shared scale:
 * c_k - sum_d mu_kd^2 / (2 sigma_d^2) + sum_d x_d mu_kd / sigma_d^2; per-component scales: additionally - sum_d x_d^2 / (2 sigma_kd^2)), whose fp32 rounding is that of sum_d |x_d mu_kd| / sigma_d^2 rather than
 * of the squared distance -- harmless where no two components that can share a trajectory's weight lie close to each other (the
 * Python binding's rule: engine._mixture_mm_ok).  Lets evaluation launches of 21 .. 40-component mixtures run both mixture
 * contractions (reference: distr/gauss.py:123-140, distr/base.py:130-137) on the matrix pipe; the terminal log-density keeps the exact
 * form.  Plan option SDEH_GMM_MM overrides ("0" never / "1" always). */
#define SDEH_DENS_FLAG_MM_OK 2
/* GMM with SHARED_SCALE only: the caller additionally promises that coordinates >= n are identical in every component
 * (loc[k,d] == loc[0,d] for d >= n), as in the reference's own high-dimensional mixtures, which pad a 2-d mixture with
 * zero means (distr/gauss.py:59-60).  Those coordinates factor out of the mixture as one Gaussian: they cancel in the
 * responsibilities and contribute (loc_d - x_d)/scale_d^2 to the score.  Encoded in bits 8..23 of `flags` as n + 1
 * (0 = no promise). */
#define SDEH_DENS_FLAG_NVARY(n) ((((n) + 1) & 0xFFFF) << 8)
#define SDEH_DENS_FLAG_GET_NVARY(flags) ((((flags) >> 8) & 0xFFFF) - 1) /* -1 when absent */

typedef struct {
  int32_t kind;         /* SdehDensityKind */
  int32_t dim;
  int32_t n_components; /* GMM: K; MULTI_WELL: n_double_wells */
  int32_t flags;        /* SDEH_DENS_FLAG_* : promises of the caller about the parameter values */
  float log_norm_const; /* Distribution.log_norm_const (0 when None) -- added by unnorm_log_prob where the reference does */
  float p0;             /* MULTI_WELL: separation;  FUNNEL: variance of x_0 */
  float p1;             /* MULTI_WELL: shift */
  float p2;
  const float* loc;
  const float* scale;
  const float* mixture_weights; /* GMM only; NULL == single component */
} SdehDensity;

/* models/mlp.py:43-82 TimeEmbed.  hidden_w[0] is [C,2C]; hidden_w[i>0] are [C,C]; out_w is [dim_out,C]. */
typedef struct {
  int32_t channels;
  int32_t n_hidden; /* len(hidden_layer) = num_layers-1 >= 1;  0 == module absent (score_model=None) */
  int32_t dim_out;
  int32_t reserved;
  const float* coeff; /* timestep_coeff [C] */
  const float* phase; /* timestep_phase [C] */
  const float* hidden_w[SDEH_MAX_HIDDEN];
  const float* hidden_b[SDEH_MAX_HIDDEN];
  const float* out_w;
  const float* out_b;
} SdehTimeEmbed;

/* models/mlp.py:85-122 FourierMLP.  input_w [C,d]; hidden_w[i] [C,C] (num_layers-2 of them); out_w [d,C]. */
typedef struct {
  int32_t dim;
  int32_t channels;
  int32_t n_hidden;
  int32_t activation; /* SdehActivation, shared with the time embeddings (same callable in the reference) */
  const float* input_w;
  const float* input_b;
  const float* hidden_w[SDEH_MAX_HIDDEN];
  const float* hidden_b[SDEH_MAX_HIDDEN];
  const float* out_w;
  const float* out_b;
  SdehTimeEmbed timestep_embed; /* FourierMLP.timestep_embed (num_layers=2, dim_out=C) */
} SdehFourierMLP;

/* TimeReversalLoss.inference_ctrl (Bridge): ClippedCtrl or LerpPriorCtrl over its own FourierMLP / TimeEmbed
 * (conf/solver/bridge.yaml, basic_bridge.yaml; built on the solver's generative sde and prior, solver/oc.py:134-140). */
typedef struct {
  int32_t ctrl_kind; /* SDEH_CTRL_CLIPPED or SDEH_CTRL_LERP_PRIOR */
  float clip_model, clip_score, scale_score;
  SdehFourierMLP base_model;
  SdehTimeEmbed score_model;
} SdehInferenceCtrl;

/* One evaluation problem == what the reference loss object + its collaborators hold. */
typedef struct {
  int32_t loss_kind; /* SdehLossKind */
  int32_t ctrl_kind; /* SdehCtrlKind */
  int32_t sde_kind;  /* SdehSdeKind */
  int32_t flags;     /* SDEH_FLAG_* */
  /* generative_ctrl attributes (reparam.py): +INF == None */
  float clip_model, clip_score, scale_score;
  float clip_target; /* solver/oc.py:48-54, +INF == None */
  /* sde (eq/sdes.py); generative=True (sign=+1) unless SDEH_FLAG_INFERENCE_SDE */
  float terminal_t;
  float vp_beta_min, vp_beta_max, vp_scale; /* VP: diff_coeff_sq_min/max, scale_diff_coeff */
  float ou_drift, ou_diff;                  /* ConstOU / ScaledBM: drift_coeff, diff_coeff */
  float exp_alpha, exp_sigma;               /* ExponentialIntegratorSDELoss.alpha / .sigma */
  SdehFourierMLP base_model;                /* generative_ctrl.base_model */
  SdehTimeEmbed score_model;                /* generative_ctrl.score_model (gamma(t); dim_out 1 or d) */
  SdehDensity target;                       /* target (score + terminal log-density) */
  SdehDensity prior;                        /* prior_score of Lerp*Ctrl / reference_ctrl */
  SdehDensity second;                       /* initial_log_prob (DIS) or reference_log_prob (PIS/DDS) density */
  SdehInferenceCtrl inference;              /* read only with SDEH_FLAG_INFERENCE_CTRL */
  /* Optional device-resident Philox offset (NULL = none): one uint64 in device memory that every kernel ADDS to its by-value
   * `offset` argument when it starts.  Lets a captured hipGraph (forward + backward + optimizer step) be replayed with fresh noise:
   * the launch arguments are frozen at capture time, the counter is bumped by a node inside the graph (the reference's
   * torch.randn_like draws advance the generator's device-side offset under graph capture in the same way). */
  const uint64_t* rng_offset_dev;
} SdehProblem;

typedef struct {
  int32_t dim;         /* d <= 64 with channels = 64: every entry point.  d <= 256 with channels = 128 / 256, and 64 < d <= 256 with
                          channels = 64: the "wide" kernels -- sdeh_simulate_fwd[_aux / _train2], sdeh_ctrl_backward[_ex] +
                          sdeh_weight_grad (every loss method, closed-form AND mixture targets; the score planes a mixture's backward
                          needs come from sdeh_simulate_fwd_train2), sdeh_bridge_div_backward_wide (a Bridge needs channels >= 128
                          and an inference network with 1 or 2 hidden layers; exact divergence only) */
  int32_t channels;    /* C: 64, 128 or 256 */
  int32_t max_hidden;  /* largest n_hidden of base_model */
  int32_t max_steps;   /* largest T = len(ts)-1 */
  int32_t max_components; /* largest GMM K (0 if unused) */
  int32_t device;      /* HIP device ordinal */
  int64_t max_batch;   /* ABI v5.  Largest batch of a wide Bridge launch (its divergence needs 32 partial sums per trajectory of plan-
                          owned scratch): reserved by sdeh_plan_create.  0 = none: call sdeh_plan_reserve before the first launch.
                          No stream-ordered entry point allocates. */
} SdehPlanDesc;

typedef struct SdehPlan SdehPlan;

/* ABI version of the loaded library (== SDEH_ABI_VERSION it was built with). */
int32_t sdeh_abi_version(void);
/* Thread-local description of the last non-zero status. */
const char* sdeh_last_error(void);

/* Allocates the plan's device workspace (packed weights, per-step tables, estimator scratch).
 * Replaces: nothing in the reference (loss objects are plain Python, losses/oc.py:13-48). */
int32_t sdeh_plan_create(const SdehPlanDesc* desc, SdehPlan** plan);
void sdeh_plan_destroy(SdehPlan* plan);
/* ABI v5.  Grows the plan's batch-dependent scratch (wide Bridge: desc.max_batch) to `max_batch` trajectories.  Synchronises the
 * device (hipFree / hipMalloc): call it outside stream captures; sdeh_simulate_fwd* never allocates and answers SDEH_ERR_CAPACITY when
 * the reservation is too small. */
int32_t sdeh_plan_reserve(SdehPlan* plan, int64_t max_batch);
/* ABI v5.  Kernel-mode options of a plan (which compiled variant / tiling serves a launch; several of them round differently, see
 * INTEGRATION.md): `value` NULL or "" = automatic choice.  sdeh_plan_create takes the environment variables of the same names as the
 * initial values (a test override, read ONCE there); nothing on the launch path reads the environment.  Names:
 *   SDEH_LEGACY (single-wave trajectory kernel)      SDEH_GENERIC_ONLY ("1" | "2": run-time switched variants only)
 *   SDEH_WS_GROUPS ("2" | "4" | "2h" | "4h" | "p")    SDEH_WS_QUAD ("0" | "1")      SDEH_WS_VOUT ("0")      SDEH_WS_BARRIER
 *   SDEH_BWD_PLANES (plane-writing backward)         SDEH_BWD_TILE ("16" | "32")   SDEH_BWD_WAVES ("2" | "4")
 *   SDEH_BWD_V1 / SDEH_BWD_V2 (channel- / trajectory-split fused backward)        SDEH_BWD_NO_VIO      SDEH_BWD_SCAN ("0" | "1": the
 *   scan form of back-propagation through time, d <= 4)        SDEH_BWD_ZREC ("0": the fused backward ignores the pre-activation record)
 *   SDEH_BRIDGE_TILES ("64" | "32g")   SDEH_BRIDGE_SPLIT ("1" | "4")   SDEH_WIDE_CT ("1" | "2")   SDEH_WIDE_SPLIT ("1" | "2" | "4" | "8")
 *   SDEH_GMM_MM ("0": never / "1": always evaluate an eligible mixture's contractions on the matrix pipe; default: SDEH_DENS_FLAG_MM_OK decides)
 *   SDEH_WS_OUT4 ("0": evaluation launches at d = 5 .. 16 keep the out layer on 32-row matrix tiles instead of 4 x 4 x 1 row groups)
 * Unknown names: SDEH_ERR_INVALID. */
int32_t sdeh_plan_set_option(SdehPlan* plan, const char* name, const char* value);

/* Measurement hooks (reference analogue: the wall-clock `eval/sample_time`, solver/oc.py:88-97).  When enabled,
 * every sdeh_simulate_fwd records HIP events immediately before and after the TRAJECTORY kernel on the stream
 * it is launched on; sdeh_plan_last_kernel_ms waits for the stop event and returns the elapsed time. */
int32_t sdeh_plan_set_timing(SdehPlan* plan, int32_t enable);
int32_t sdeh_plan_last_kernel_ms(SdehPlan* plan, float* ms);
/* ABI v6.  The timed launches are kept in a ring of 128 event pairs: entry `back` launches ago (0 = the newest) with the name of the
 * kernel that served it, so that a run can read its kernel durations AFTERWARDS instead of synchronising after every launch.  Returns 1
 * when there is no such entry. */
int32_t sdeh_plan_timing_entry(SdehPlan* plan, int32_t back, float* ms, char* name, int32_t name_len);
/* Which compiled trajectory kernel served the plan's last sdeh_simulate_fwd* call (e.g. "traj_ws<50_0_pis_gmm4>",
 * "traj_legacy<64_1_g>", "traj_wide<C=256,CT=2>", "bridge_wide<C=256>"); "" before the first call.  The string is owned by the plan. */
const char* sdeh_plan_last_kernel_name(SdehPlan* plan);

/*
 * The hot path.  Replaces {TimeReversalLoss,ReferenceSDELoss,ExponentialIntegratorSDELoss}.simulate
 * (losses/oc.py:156-230, 286-343, 400-457) including, per step, generative_ctrl(t,x) (models/reparam.py,
 * models/mlp.py), sde.diff/drift/drift_div_int (eq/sdes.py), target/prior scores (distr/...py), the Gaussian
 * draw `torch.randn_like(x)` and the EM / exponential-integrator update; and at the end the terminal
 * log-densities.
 *
 *   ts      [n_steps+1]           time grid (utils/common.py:18-55), device
 *   x0      [batch, d]            initial states (prior.sample), device
 *   noise   [n_steps, batch, d]   standard-normal draws to consume INSTEAD of the in-kernel generator
 *                                 (parity mode: reproduces the reference on identical noise), or NULL:
 *                                 in-kernel Philox4x32-10 + Box-Muller: counter (global row, dim/4, step, offset),
 *                                 key (seed)  -> results do not depend on how the batch is sharded.
 *   row_offset                    global index of x0 row 0 (rank * local_batch for sharded runs)
 *   x_T     [batch, d]   out      terminal states ("samples")
 *   rnd     [batch]      out      log Radon-Nikodym derivative per trajectory (reference shape [B,1])
 *   xs      [n_steps+1, batch, d] out or NULL: whole trajectory (return_traj=True)
 */
int32_t sdeh_simulate_fwd(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                          const float* x0, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                          int64_t row_offset, float* x_T, float* rnd, float* xs, void* stream);

/*
 * Backward of the control network for the log-variance losses (method "lv" / "lv_traj"; replaces what
 * `loss.backward()` does through losses/oc.py:60-70,204-219,319-331,418-443 in solver/base.py:399-407).  There the SDE is
 * driven by the DETACHED control, so x_t is constant in the graph and d rnd_i / d u_{i,t} = dB_{i,t}.  For every row
 * n = t*B + i the kernel re-evaluates the FourierMLP at the stored x_t = xs[t,i], forms w_i dB (noise replayed from the
 * same (seed, offset, row_offset) as the forward call, or read from `noise`) and back-propagates through clip,
 * out_layer, activations and hidden layers.  It writes, coordinate-major (N = n_steps*batch):
 *   zt   [(Lh+1), C, N]  pre-activation of layer k          dt   [(Lh+1), C, N]  d loss / d zt
 *   dout [d, N]          d loss / d (out_layer output)      dgam [g, N]          per-row d loss / d gamma(t), g = 1 or d
 * from which the weight gradients are plain GEMMs over N (dW_k = dt[k] . act(zt[k-1])^T, ...): see
 * sde_sampler_amd/losses/_autograd.py.  grad_rnd [batch] = d loss / d rnd.  problem->flags must be those of the
 * forward call.  Without SDEH_FLAG_CHANGE_SDE_CTRL (methods "kl" / "kl_ito") the control drives the SDE, so the kernel
 * runs the discrete adjoint backwards through time instead (one trajectory per lane, d loss / d x_t carried in
 * registers), with the reference's autograd semantics: the mixture score is a constant of the graph
 * (distr/base.py:130-137, create_graph=False), the analytic scores are differentiated.
 */
int32_t sdeh_ctrl_backward(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                           const float* xs, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                           int64_t row_offset, const float* grad_rnd, float* zt, float* dt, float* dout, float* dgam,
                           void* stream);

/*
 * Training a Bridge (TimeReversalLoss with an inference control, losses/oc.py:189-206).  Method "lv" / "lv_traj": the
 * SDE is driven by the detached generative control, so x_t is constant in the graph and, per row n = (t, i),
 *   d rnd / d u = dB                      -> generative network:  sdeh_ctrl_backward on the problem WITHOUT the inference control
 *   d rnd / d v = (u + v) dt + dB         -> inference network:   sdeh_ctrl_backward_ex on a problem whose generative slots
 *                                            hold the inference control, gextra = the (u + v) plane of the forward pass
 *   d rnd / d theta_v  of  sigma div_x v dt -> sdeh_bridge_div_backward (reverse mode over the forward-mode tangents)
 * sdeh_simulate_fwd_aux == sdeh_simulate_fwd that also writes gp[n_steps, batch, d] = u + v per step (Bridge problems only).
 * sdeh_bridge_div_backward: problem with SDEH_FLAG_INFERENCE_CTRL; zt = the inference network's pre-activation planes
 * written by sdeh_ctrl_backward_ex; with N = n_steps*batch, Lh = its hidden layers, it writes
 *   tz, ta, td [d][(Lh+1), C, N]   d z_l / d x_j,  act'(z_l) d z_l / d x_j,  adjoint of d z_l / d x_j
 *   d2 [(Lh+1), C, N]  adjoint of z_l through the divergence       cj [d, N]  w_i sigma dt 1[|v_nn,j| <= clip_model]
 *   dgam [g, N]        d / d gamma(t) of the score part of the divergence
 * from which the parameter gradients are GEMMs over N (sde_sampler_amd/losses/_autograd.py).
 *
 * div_noise [n_steps, batch, d] (both entry points; NULL = exact divergence): probe vectors of the Hutchinson estimator
 * eps^T J eps (TimeReversalLoss.div_estimator = "rademacher" / "gauss", utils/autograd.py:25-42; training only, n_samples = 1;
 * the caller draws them).  With probes the backward works on ONE tangent (tz, ta, td are [1][(Lh+1), C, N]) and
 * cj [d, N] = w_i sigma dt eps_j 1[|v_nn,j| <= clip_model].
 *
 * Method "kl" / "kl_ito" (the generative control drives the SDE: back-propagation through time): the inference network's
 * gradient does not depend on d loss / d x (v does not drive the SDE), so it is computed row-parallel first -- force the
 * row-parallel mode by setting SDEH_FLAG_CHANGE_SDE_CTRL in that problem's flags -- together with its own contribution to
 * d loss / d x_t (dx_out; sdeh_bridge_div_backward ADDS the divergence term's to the same plane); the generative network's
 * back-propagation through time then takes that plane as lam_extra and the (u + v) plane as cost_ctrl.
 *   gextra    [T,B,d] or NULL (row-parallel): extra upstream gradient  w_i cdt gextra
 *   cost_ctrl [T,B,d] or NULL (BPTT): the control entering the running cost instead of u
 *   lam_extra [T,B,d] or NULL (BPTT): added to d loss / d x_t          dx_out [T,B,d] or NULL (row-parallel): written
 *   nn_in     [T,B,d] or NULL: the raw network outputs written by sdeh_simulate_fwd_train; with it `zt` is an INPUT (the forward
 *             launch's pre-activation planes) and the kernel does not re-evaluate the network
 *   xt_out    [d, N] or NULL (plans for channels 128 / 256 or d > 64 only): x_t coordinate-major, written next to the planes -- the
 *             operand of input_embed.weight's gradient in the layout sdeh_weight_grad reads (ABI v4)
 *   sc_in [T,B,d], tscore_in [B,d] or NULL (wide plans with a MIXTURE target only): the planes sdeh_simulate_fwd_train2 wrote -- the
 *             wide backward evaluates no mixture (sc_in whenever the control has a target-score term, tscore_in for methods kl / kl_ito
 *             with SDEH_FLAG_TERMINAL_TARGET).  On a wide plan sdeh_simulate_fwd_train2 keeps everything ROW-major: xs [n_steps+1, batch,
 *             d], sc [n_steps, batch, d], tscore [batch, d] (sc / tscore written for mixture targets only; closed-form targets are
 *             re-evaluated by the backward kernel).
 *
 * sdeh_simulate_fwd_train == sdeh_simulate_fwd for a training step (xs required) that also keeps what the backward needs:
 *   zt [(Lh+1), C, n_steps*batch]  pre-activations of every layer, coordinate-major (n = step * batch + row)
 *   nn [n_steps, batch, d]         network output before the clamp
 * Returns 0 when the planes were written, 1 when the launch was served by a kernel that keeps none (mixture tables too large
 * for LDS, very deep networks): then call the backward with nn_in = NULL.  Not for problems with an inference control.
 */
int32_t sdeh_simulate_fwd_train(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                                const float* x0, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                                int64_t row_offset, float* x_T, float* rnd, float* xs, float* zt, float* nn, void* stream);
int32_t sdeh_simulate_fwd_aux(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                              const float* x0, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                              int64_t row_offset, float* x_T, float* rnd, float* xs, float* gp, const float* div_noise,
                              void* stream);
/* sdeh_simulate_fwd_aux that also keeps, on a WIDE plan with a MIXTURE target, what sdeh_ctrl_backward_ex reads for the generative
 * control of a Bridge (row-major, either may be NULL): sc [n_steps, batch, d] the score entering the generative control,
 * tscore [batch, d] = 1[|target.unnorm_log_prob(x_T)| <= clip_target] target.score(x_T).  Other targets: nothing is written. */
int32_t sdeh_simulate_fwd_aux2(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                               const float* x0, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                               int64_t row_offset, float* x_T, float* rnd, float* xs, float* gp, const float* div_noise,
                               float* sc, float* tscore, void* stream);

/*
 * Fused training backward (csrc/sdeh_bwdf.hip): what `loss.backward()` does in the reference (solver/base.py:407 through the
 * unrolled loops of losses/oc.py:176-222, 301-334, 416-446) for the control's FourierMLP in ONE kernel -- back-propagation at the
 * stored trajectory (through time for methods "kl" / "kl_ito") and the weight-gradient contractions, with no [C, n_steps*batch] plane
 * in device memory.  Compiled for channels = 64, one to three hidden layers (conf/model/base/fouriermlp.yaml: num_layers 4 = two),
 * d <= 64, no inference control: sdeh_ctrl_backward_fused_supported says whether a problem qualifies; sdeh_ctrl_backward_ex + sdeh_weight_grad
 * take the rest.  Through time, batches below 16 384 trajectories run the same kernel on tiles of 16 trajectories (csrc/sdeh_bwdf16.hip:
 * the chain of a step is half as long -- the reference's training batches are 512 and 2048); same arguments, same results up to rounding.
 *
 * sdeh_simulate_fwd_train2 == sdeh_simulate_fwd for a training step that keeps what the fused backward reads, all
 * COORDINATE-MAJOR (consecutive trajectories at consecutive addresses: both kernels move whole cache lines):
 *   xs     [n_steps+1, d, batch]  the trajectory
 *   sc     [n_steps, d, batch]    the score entering the control before clip_score and gamma(t) (models/reparam.py:56-83,131-197:
 *                                 target score, lerp of prior and target score, ...); NULL for ClippedCtrl
 *   tscore [d, batch] or NULL     1[|target.unnorm_log_prob(x_T)| <= clip_target] * target.score(x_T)  (needed by "kl" methods
 *                                 with SDEH_FLAG_TERMINAL_TARGET)
 * Returns 0, or 1 when the launch was served by a kernel that writes none of them (then use the plane-based backward).
 *
 * sdeh_ctrl_backward_fused: xs / sc / tscore as written by sdeh_simulate_fwd_train2 (noise, if given, is the caller's
 * [n_steps, batch, d] tensor); same problem / ts / noise / seed / offset / row_offset as the forward call, grad_rnd [batch] =
 * d loss / d rnd_i.  `scratch` (sdeh_ctrl_backward_fused_sizes) holds per-team partial gradients, summed deterministically (no
 * atomics).  `out` (floats, with P = 32 * ceil(d / 32), gw = 2 if gamma(t) is scalar else 64):
 *   input_embed.weight [64, P] (columns >= d unused) | hidden_layer[l].weight [Lh][64, 64] | out_layer.weight [P, 64] |
 *   hidden_layer[l].bias [Lh][64] | out_layer.bias [P] | d loss / d (timestep_embed(t) + input_embed.bias) [n_steps, 64] |
 *   d loss / d gamma(t) [n_steps, gw]  (scalar gamma: the sum of the two columns; vector gamma: columns < d)
 * The two [n_steps, .] tables are the inputs of sdeh_time_embed_backward.
 */
int32_t sdeh_simulate_fwd_train2(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                                 const float* x0, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                                 int64_t row_offset, float* x_T, float* rnd, float* xs, float* sc, float* tscore, void* stream);
int32_t sdeh_ctrl_backward_fused_supported(const SdehPlan* plan, const SdehProblem* problem);
int32_t sdeh_ctrl_backward_fused_sizes(int32_t dim, int32_t n_hidden, int32_t n_steps, int64_t batch, int32_t gamma_dim, int32_t bptt,
                                       int64_t* scratch_floats, int64_t* out_floats);
int32_t sdeh_ctrl_backward_fused(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                                 const float* xs, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                                 int64_t row_offset, const float* grad_rnd, const float* sc, const float* tscore,
                                 float* scratch, int64_t scratch_floats, float* out, void* stream);
int32_t sdeh_ctrl_backward_fused_ex(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                                    const float* xs, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                                    int64_t row_offset, const float* grad_rnd, const float* sc, const float* tscore,
                                    const float* cost_ctrl, const float* lam_extra, float* scratch, int64_t scratch_floats,
                                    float* out, void* stream);
int32_t sdeh_ctrl_backward_ex(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                              const float* xs, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                              int64_t row_offset, const float* grad_rnd, const float* gextra, const float* cost_ctrl,
                              const float* lam_extra, float* dx_out, float* zt, float* dt, float* dout, float* dgam,
                              const float* nn_in, float* xt_out, const float* sc_in, const float* tscore_in, void* stream);
int32_t sdeh_bridge_div_backward(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps,
                                 const float* xs, int64_t batch, const float* grad_rnd, const float* zt, float* tz,
                                 float* ta, float* td, float* d2, float* cj, float* dgam, float* dx_accum,
                                 const float* div_noise, void* stream);

/* 64-channel Bridge, FUSED (ABI v5; csrc/sdeh_bridgef.hip + the row-parallel kernel of csrc/sdeh_bwdf2.hip): every gradient of the
 * INFERENCE network -- the first-order terms d rnd / d v = (u + v) dt + dB (losses/oc.py:189-202) and the divergence term
 * sigma div_x v dt differentiated once more (utils/autograd.py:14-21 with create_graph=True) -- in three launches, without the
 * per-coordinate planes of sdeh_ctrl_backward_ex + sdeh_bridge_div_backward (3 (Lh + 1) C floats per row and coordinate).
 *   problem:   the inference control as the control of a plain problem (ClippedCtrl / LerpPriorCtrl; SDEH_FLAG_CHANGE_SDE_CTRL set, no
 *              SDEH_FLAG_INFERENCE_*: v does not drive the SDE, every term is row-parallel), two hidden layers of 64 channels, d <= 64
 *   xs        [n_steps + 1, d, batch]  the trajectory, COORDINATE-MAJOR
 *   cost_ctrl [n_steps, d, batch]      u + v entering the running cost (what sdeh_simulate_fwd_aux returns, coordinate-major)
 *   noise / seed / offset / row_offset: as in the forward launch; grad_rnd [batch] = d loss / d rnd_i
 *   dx_out    [n_steps, d, batch] or NULL: d loss / d x_t of the inference control's terms (W_in^T adj(Z_0) + its score term's Jacobian).
 *             Method kl: the generative network's back-propagation through time adds it to its adjoint at every step and takes its running
 *             cost on u + v -- sdeh_ctrl_backward_fused_ex(..., cost_ctrl, lam_extra = dx_out, ...), == sdeh_ctrl_backward_fused otherwise.
 *   scratch:  sdeh_bridge_backward_fused_sizes floats (per-team partial records, three [64, n_steps * batch] planes)
 *   out:      the record of sdeh_ctrl_backward_fused (n_hidden = 2), then the divergence term's DIRECT weight gradients, to be added:
 *             hidden_layer[0..1].weight [2][64, 64] | input_embed.weight^T [P, 64] | out_layer.weight [P, 64]     (P = 32 ceil(d / 32))
 * Deterministic (per-wave sequential sums; fixed-order partial sums). */
/* The FORWARD half of the same split (ABI v5).  The SDE of a Bridge is driven by the generative control alone (losses/oc.py:176-217),
 * so a training forward is (1) the plain problem -- the Bridge's problem without its inference control -- through
 * sdeh_simulate_fwd_train2u == sdeh_simulate_fwd_train2 that also keeps u [n_steps, d, batch], the control driving the SDE, and
 * (2) sdeh_bridge_inference_fwd: for every (step, trajectory) independently what the inference control adds to rnd,
 *     drnd_i = sum_t [ sigma div_x v dt + (u . v + |v|^2 / 2) dt + v . dB ]            (dB only with SDEH_FLAG_ITO; exact divergence)
 * and cost_ctrl = u + v [n_steps, d, batch] for sdeh_bridge_backward_fused.  rnd of the Bridge = rnd of (1) + drnd (another summation
 * order than the step-sequential sdeh_simulate_fwd_aux: equal to fp32 rounding).  Same problem conventions as
 * sdeh_bridge_backward_fused; scratch: sdeh_bridge_inference_fwd_scratch_floats. */
int32_t sdeh_simulate_fwd_train2u(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, const float* x0,
                                  int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                  float* x_T, float* rnd, float* xs, float* sc, float* tscore, float* u, void* stream);
/* ABI v6.  The training forward that also KEEPS THE NETWORK'S PRE-ACTIVATIONS, and the fused backward that reads them.  The reference's
 * autograd keeps every layer's activations between loss(...) and loss.backward() (losses/oc.py:232-256 -> models/mlp.py:114-122 through
 * solver/base.py:399-407); sdeh_simulate_fwd_train2 + sdeh_ctrl_backward_fused re-evaluate the network per step instead (1.7 x the
 * algorithmic matrix work of the backward).  sdeh_simulate_fwd_train3 == sdeh_simulate_fwd_train2u (u may be NULL) that also writes
 *   zrec [n_steps][ceil(batch / 32)] { [n_hidden + 1][16][32][4] ; [ceil(d / 32)][8][32][4] }     (sdeh_zrec_floats floats)
 *        per (step, tile of 32 trajectories): the pre-activations Z_k of the generative network -- layer k, channel quad cq (channels
 *        4 cq .. 4 cq + 3) of trajectory j at [k][cq][j][0..3] -- followed by the raw network output (before the clamp), coordinate quad
 *        cq of trajectory j at [cq / 8][cq % 8][j][0..3].  This is the register layout of the kernels on both sides (16-byte accesses,
 *        1 KB contiguous per instruction); quads that hold only coordinates >= d may be left unwritten.
 * and returns 0 (everything kept) or 1 (served by a kernel that keeps nothing: mixture tables beyond LDS).
 * sdeh_ctrl_backward_fused_z == sdeh_ctrl_backward_fused_ex that is also given zrec (NULL = _ex): the launches that can read the record
 * do not re-evaluate the network (act / act' from the stored Z_k: a ReLU unit on its kink takes the forward launch's side by
 * construction); the others ignore it.  Same scratch / out as sdeh_ctrl_backward_fused. */
int64_t sdeh_zrec_floats(int32_t dim, int32_t n_hidden, int32_t n_steps, int64_t batch);
/* 1 when the launch that serves sdeh_ctrl_backward_fused_z for this problem and batch reads the record (so that the forward only keeps
 * what will be read), 0 when it would be ignored. */
int32_t sdeh_ctrl_backward_fused_reads_zrec(const SdehPlan* plan, const SdehProblem* problem, int64_t batch);
int32_t sdeh_simulate_fwd_train3(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, const float* x0,
                                 int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                 float* x_T, float* rnd, float* xs, float* sc, float* tscore, float* u, float* zrec, void* stream);
int32_t sdeh_ctrl_backward_fused_z(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, const float* xs,
                                   int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                   const float* grad_rnd, const float* sc, const float* tscore, const float* cost_ctrl,
                                   const float* lam_extra, const float* zrec, float* scratch, int64_t scratch_floats, float* out,
                                   void* stream);
int64_t sdeh_bridge_inference_fwd_scratch_floats(int32_t n_steps, int64_t batch);
int32_t sdeh_bridge_inference_fwd(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, const float* xs,
                                  int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset, const float* u,
                                  float* drnd, float* cost_ctrl, float* scratch, int64_t scratch_floats, void* stream);
int32_t sdeh_bridge_backward_fused_sizes(int32_t dim, int32_t n_hidden, int32_t n_steps, int64_t batch, int32_t gamma_dim,
                                         int64_t* scratch_floats, int64_t* out_floats);
int32_t sdeh_bridge_backward_fused(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, const float* xs,
                                   int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                   const float* grad_rnd, const float* cost_ctrl, float* dx_out, float* scratch, int64_t scratch_floats,
                                   float* out, void* stream);
/*
 * Bridge on WIDE networks (channels 128 / 256: conf/solver/bridge.yaml with the channels of BASELINE configs[4]): gradient of the
 * divergence term  sum_n w_i sigma dt sum_j 1[|v_nn,j| <= clip_model] J_jj(x_n; theta_v)  w.r.t. the inference network -- what the
 * reference's autograd does through the d backward passes per step of utils/autograd.py:14-22 (create_graph=True) under
 * losses/oc.py:189-200.  The per-(row, coordinate) tangents cannot be written out at these sizes (the planes tz / ta / td of
 * sdeh_bridge_div_backward would take d (Lh+1) C N floats), so the contraction is fused (csrc/sdeh_wide_bwd.hip): the gradient of one
 * hidden layer stays in the accumulators of persistent workgroups for a whole launch; one launch per hidden layer.  Exact
 * divergence only; inference networks with one or two hidden layers; the first-order terms go through sdeh_ctrl_backward_ex as for
 * 64 channels (the trajectory kernels accept gp for wide plans).
 *   zt   [(Lh+1), C, N]  in : pre-activations of the inference network, as sdeh_ctrl_backward_ex wrote them for the inference problem
 *   d2   [(Lh+1), C, N]  out: adjoints of those pre-activations through the divergence (ADD to the dt planes of the first-order pass:
 *                             the weight gradients of both are one sdeh_weight_grad contraction)
 *   dgam [g, N]          out: d / d gamma(t) of the score part of the divergence (LerpPriorCtrl; NULL for ClippedCtrl)
 *   dx_accum [T, B, d]   in/out or NULL: method "kl" / "kl_ito" -- the divergence term's share of d loss / d x_t (W_in^T adj z_0) is
 *                             ADDED to the plane sdeh_ctrl_backward_ex wrote as dx_out (it becomes lam_extra of the generative BPTT)
 *   out  (floats): d / d input_embed.weight TRANSPOSED [d, C] | d / d out_layer.weight [d, C] | d / d hidden_layer[0].weight [C, C] |
 *                  (two hidden layers:) d / d hidden_layer[1].weight TRANSPOSED [C, C]   -- the tangent streams' direct contributions
 *   scratch: sdeh_bridge_div_backward_wide_sizes floats (per-workgroup partials, summed deterministically: no atomics).
 */
int32_t sdeh_bridge_div_backward_wide_sizes(int32_t dim, int32_t channels, int32_t n_hidden, int32_t n_steps, int64_t batch,
                                            int64_t* scratch_floats, int64_t* out_floats);
int32_t sdeh_bridge_div_backward_wide(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, const float* xs,
                                      int64_t batch, const float* grad_rnd, const float* zt, float* d2, float* dgam, float* dx_accum,
                                      float* scratch, int64_t scratch_floats, float* out, void* stream);

/*
 * Batch reductions of BaseOCLoss.compute_results / compute_loss (losses/oc.py:72-123), as mergeable partial
 * statistics so that ranks can combine them with one tiny collective.
 *   out[0] = n (rows with rnd < max_rnd, or finite rnd when max_rnd = +INF)   out[1] = sum(-rnd)
 *   out[2] = M2 = sum((rnd-mean)^2)     out[3] = m = max(-rnd)
 *   out[4] = sum(exp(-rnd-m))           out[5] = sum(exp(2(-rnd-m)))
 *   out[6] = rows filtered out          out[7] = reserved
 * Filtering (losses/oc.py:50-58): max_rnd = NaN keeps every row (evaluation), +INF keeps finite rows, a finite
 * value keeps rows with rnd < max_rnd.  `out` is a device buffer of 8 floats; `scratch` a device buffer of
 * SDEH_REDUCE_SCRATCH floats owned by the caller (no plan needed).
 */
#define SDEH_REDUCE_SCRATCH 8192
int32_t sdeh_reduce_estimators(const float* rnd, int64_t batch, float max_rnd, float* scratch, float* out,
                               void* stream);

/* ABI v6.  The training loss over the batch AND its per-row gradient, without a host round trip (BaseOCLoss.compute_loss, losses/oc.py:72-92:
 * `rnd[mask].mean()` for method kl / kl_ito, `rnd[mask].var()` for lv, mask = BaseOCLoss.filter, losses/oc.py:50-58): the reduction of
 * sdeh_reduce_estimators followed by one elementwise pass,
 *   out[0..6] as sdeh_reduce_estimators;   out[7] = loss = mean | M2 / (n - 1)
 *   grad_rnd[i] = d loss / d rnd_i = 1 / n | 2 (rnd_i - mean) / (n - 1) on kept rows, 0 on dropped ones  -- what sdeh_ctrl_backward_fused takes
 *   *n_filtered += rows dropped (int64 on the device, NULL to skip): the reference's running `n_filtered`
 * Three launches; the same fp32 values as the framework's masked reductions and their autograd. */
int32_t sdeh_loss_moment(const float* rnd, int64_t batch, float max_rnd, int32_t log_variance, int64_t* n_filtered, float* scratch,
                         float* out, float* grad_rnd, void* stream);

/* ABI v6.  The reference trainer applies an optimisation step only `if loss_ok and grad_ok` (solver/base.py:409-432), a host decision.  A
 * captured step takes it on the device: the step runs on sanitised gradients and is UNDONE when *ok == 0 -- every tensor the optimizer may
 * have changed is put back from its snapshot.  table [n_tensors][3] (device, uint64): destination pointer, snapshot pointer, number of 32-bit
 * words; ok: one byte on the device (torch.bool); n_skipped (nullable): device counter, incremented when the step is undone.  One launch
 * (utils/graphs.py: GraphedTrainStep).
 * sdeh_guard_check takes the decision: *ok = loss_ok and grad_ok, with loss_ok = isfinite(*value) (max_loss < 0) or |*value| <= max_loss
 * (solver/base.py:409-415) and grad_ok = every one of the n gradient entries finite (solver/base.py:416-421, over one flat copy of the
 * gradients); when the step is rejected the n entries are zeroed in place, so that the step that still runs inside the graph sees finite input. */
int32_t sdeh_guard_check(float* grads, int64_t n, const float* value, float max_loss, uint8_t* ok, void* stream);
int32_t sdeh_guard_restore(const uint64_t* table, int32_t n_tensors, const uint8_t* ok, int64_t* n_skipped, void* stream);

/* importance weights exp(-rnd - m) (losses/oc.py:101-103) with a caller-provided global maximum m (device scalar). */
int32_t sdeh_importance_weights(const float* rnd, int64_t batch, const float* log_weight_max, float* weights,
                                void* stream);

/*
 * Plain Euler-Maruyama integration with output-time interpolation.  Replaces EulerIntegrator.integrate
 * (eq/integrator.py:93-127) + interpolate (66-77) for the SDE classes the reference passes to it:
 *   SDEH_INT_LANGEVIN    LangevinSDE (eq/sdes.py:38-65, solver/langevin.py:45-46): drift = clip(target.score(x) *
 *                        diff_coeff^2 / 2, clip_score), diff = diff_coeff.  problem: target, sde_kind CONST_OU with
 *                        ou_diff = diff_coeff, clip_score, ctrl_kind NONE.
 *   SDEH_INT_CONTROLLED  ControlledSDE (eq/sdes.py:272-305) or a bare OU (solver/oc.py:100-110,130-143): drift =
 *                        sde.drift(t,x) + sde.diff(t) * ctrl(t', x), t' = t (generative) or terminal_t - t
 *                        (SDEH_FLAG_INFERENCE_SDE); ctrl_kind NONE = uncontrolled.
 *   timesteps [n_steps+1]  integration grid          ts_out [n_out]  sorted output times inside the grid
 *   x_init [batch, d]      noise [n_steps, batch, d] standard normals (scaled by sqrt(dt) in-kernel) or NULL = Philox
 *   xs_out [n_out, batch, d]  out: for every ts_out[j] in (s - , t + eps] of step (s, t):
 *                             torch.lerp(x_s, x_t, (ts_out[j] - s) / (t - s))   (n_out <= plan max_steps + 1)
 */
typedef enum { SDEH_INT_LANGEVIN = 0, SDEH_INT_CONTROLLED = 1 } SdehIntegrateKind;
int32_t sdeh_integrate(SdehPlan* plan, const SdehProblem* problem, int32_t kind, const float* timesteps,
                       int32_t n_steps, const float* ts_out, int32_t n_out, float eps, const float* x_init,
                       int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                       float* xs_out, void* stream);

/*
 * The NICE flow target of BASELINE configs[4] (reference distr/nice.py:233-298 `Nice` around `NiceModel` 123-231: additive couplings
 * `Coupling` 43-97 -- in_block Linear + ReLU, hidden - 1 mid blocks, out_block Linear --, `Scaling` 100-120, `StandardLogistic` prior
 * 17-40).  One row-parallel evaluation of what the losses ask of a target:
 *   unnorm_log_prob(x) = sum_j -(softplus(z_j) + softplus(-z_j)) + sum_j scale_j + log_norm_const,  z = f(x) * exp(scale)
 *                        (nice.py:176-189, 276-277)
 *   score(x)           = d unnorm_log_prob / d x   (the reference differentiates with autograd, distr/base.py:130-137): the reverse
 *                        pass through the couplings (weights are constants: `requires_grad_(False)`, nice.py:271-273)
 * Every Linear is a [batch, K] x [K, N] product on the fp32 matrix pipe (csrc/sdeh_nice.hip: 64 x 128 tiles staged through LDS, bias /
 * ReLU / ReLU-mask / residual add in the epilogue).  Parameters are read from the given pointers on every call (nn.Linear layout
 * [out, in]); nothing is cached between calls.
 */
#define SDEH_NICE_MAX_COUPLING 8
typedef struct {
  int32_t dim;        /* in_out_dim: even, <= 256 (nice.py: 196) */
  int32_t n_coupling; /* len(model.coupling) <= SDEH_NICE_MAX_COUPLING (train_nice.py: 4) */
  int32_t mid_dim;    /* units of a hidden layer, a multiple of 4 (train_nice.py: 500) */
  int32_t n_mid;      /* len(coupling[i].mid_block) = hidden - 1 <= SDEH_MAX_HIDDEN (train_nice.py: 4) */
  int32_t mask_config[SDEH_NICE_MAX_COUPLING]; /* coupling[i].mask_config != 0: x[:, :, 0] is transformed ("on"), x[:, :, 1] feeds the MLP */
  const float* in_w[SDEH_NICE_MAX_COUPLING];   /* coupling[i].in_block[0].weight [mid_dim, dim / 2] */
  const float* in_b[SDEH_NICE_MAX_COUPLING];   /* [mid_dim] */
  const float* mid_w[SDEH_NICE_MAX_COUPLING][SDEH_MAX_HIDDEN]; /* coupling[i].mid_block[l][0].weight [mid_dim, mid_dim] */
  const float* mid_b[SDEH_NICE_MAX_COUPLING][SDEH_MAX_HIDDEN];
  const float* out_w[SDEH_NICE_MAX_COUPLING];  /* coupling[i].out_block.weight [dim / 2, mid_dim] */
  const float* out_b[SDEH_NICE_MAX_COUPLING];  /* [dim / 2] */
  const float* scale;   /* scaling.scale [dim] (the reference's [1, dim]) */
  float log_norm_const; /* Nice.log_norm_const (nice.py:240, 0.0) */
} SdehNice;
/* floats of caller-owned work memory for a batch: the couplings' hidden activations (kept for the reverse pass when a score is asked
 * for) + the de-interleaved halves of x and of the gradient */
int64_t sdeh_nice_work_floats(const SdehNice* nice, int64_t batch, int32_t want_score);
/* x [batch, dim] -> score [batch, dim] (or NULL: log-density only) and logp [batch] (or NULL) */
int32_t sdeh_nice_eval(const SdehNice* nice, const float* x, int64_t batch, float* score, float* logp, float* work,
                       int64_t work_floats, void* stream);

/*
 * sdeh_simulate_fwd_aux2 in SEGMENTS of the time grid, for a target whose score is not built into the kernels (v7; wide plans).  Runs
 * the Euler-Maruyama steps [step_begin, step_end) of the grid ts[0 .. n_steps]; the per-step tables (coefficients, time embeddings,
 * gamma) are prepared for the WHOLE grid by the segment that starts at step 0 and live in the plan's workspace until the last segment
 * has run (no other launch on this plan in between).  Step indices, Philox counters and the rows of xs / gp are those of the whole grid:
 * a chain of segments draws the noise one launch over all steps would.
 *   x_in  [batch, d]   state at ts[step_begin]                 x_out [batch, d]  state at ts[step_end] (a different buffer than x_in)
 *   rnd   [batch]      in/out: initialised by the segment starting at step 0 (initial log-density term per the problem's flags), read
 *                      and advanced by the others.  The terminal log-density of a built-in target is subtracted by the segment that
 *                      ends at n_steps; with target.kind = SDEH_DENS_EXTERNAL the caller subtracts it (and applies clip_target).
 *   ext_score          SDEH_DENS_EXTERNAL: the target's score at x_t for the steps of this segment, [step_end - step_begin, batch, d]
 *                      starting at this segment's first step (ext_stride = batch * d), or one [batch, d] buffer for a one-step
 *                      segment (ext_stride = 0); NULL otherwise
 *   xs [n_steps + 1, batch, d], gp [n_steps, batch, d]: as sdeh_simulate_fwd_aux2 (rows of this segment's steps are written), or NULL
 * The network part of a Bridge's divergence is accumulated per segment (its partial sums join rnd at the end of each segment): equal to
 * the one-launch result up to the order of that fp32 sum.
 * Training on a supplied target: sdeh_ctrl_backward_ex takes the score ENTERING the control per step as `sc_in` [n_steps, batch, d] (the
 * supplied score after the control's interpolation weight) and, for the methods that back-propagate through time, `tscore_in` [batch, d] =
 * 1[|log rho(x_T)| <= clip_target] score(x_T); the supplied score is a constant of the adjoint recursion (the reference obtains such
 * scores by autograd without a graph: distr/base.py:130-137 under models/reparam.py:56-66, 185-197).
 */
int32_t sdeh_simulate_fwd_steps(SdehPlan* plan, const SdehProblem* problem, const float* ts, int32_t n_steps, int32_t step_begin,
                                int32_t step_end, const float* x_in, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                                int64_t row_offset, float* x_out, float* rnd, float* xs, float* gp, const float* ext_score,
                                int64_t ext_stride, void* stream);

/*
 * Entropy-regularised optimal-transport cost between two point clouds.  Replaces Sinkhorn.compute (eval/sinkhorn.py:
 * 63-178), whose [n, m] reductions the reference delegates to pykeops (pinned `pykeops` in its requirements; absent here):
 *   M_ij = ||x_i - y_j||_p (p = 1 or 2);  u = 0, v = eps log w_y;  repeat (at most max_iters times)
 *     u_i = eps (log w_x_i - logsumexp_j((v_j - M_ij)/eps));   v_j = eps (log w_y_j - logsumexp_i((u_i - M_ij)/eps))
 *   until max|du| < stop_thresh and max|dv| < stop_thresh;   distance = sum_ij exp((u_i + v_j - M_ij)/eps) M_ij.
 * w_x / w_y NULL = the reference's uniform weights (ones(n)/n and ones(m)/m * n/m).  The convergence test runs on the
 * device (no host synchronisation; converged iterations' kernels return immediately).
 *   out[0] = distance   out[1] = iterations run   out[2], out[3] = last max|du|, max|dv|
 *   corr_x_to_y [n] / corr_y_to_x [m]: optional argmax of the transport plan per row / column (NULL to skip)
 *   workspace: sdeh_sinkhorn_workspace_floats(n, m) floats, caller-owned
 */
int64_t sdeh_sinkhorn_workspace_floats(int64_t n, int64_t m);
int32_t sdeh_sinkhorn(const float* x, int64_t n, const float* y, int64_t m, int32_t d, const float* w_x, const float* w_y,
                      int32_t p, float eps, int32_t max_iters, float stop_thresh, float* workspace, float* out,
                      int64_t* corr_x_to_y, int64_t* corr_y_to_x, void* stream);

/*
 * One pass over samples[batch, d] (+ optional importance weights[batch], optional box domain[d, 2]) producing everything
 * get_metrics (eval/metrics.py:70-184) reduces over the batch, as sums that can be merged across ranks:
 *   out[0] = batch   out[1] = sum w   out[2] = sum w^2   out[3] = rows inside the domain (-1 without a domain)
 *   out[4..8)  = sum_i f_k(x_i),  f = square, abs, sum, square_minus_sum (distr/base.py:12-17; each sums over coordinates)
 *   out[8..12) = sum_i w_i f_k(x_i)
 *   out[12 .. 12+d) = per-coordinate mean      out[12+d .. 12+2d) = per-coordinate M2 = sum_i (x_id - mean_d)^2
 * d <= 256.  scratch: sdeh_sample_stats_scratch_floats(d) floats.
 */
int64_t sdeh_sample_stats_scratch_floats(int32_t d);
int32_t sdeh_sample_stats(const float* samples, int64_t batch, int32_t d, const float* weights, const float* domain,
                          float* scratch, float* out, void* stream);

/*
 * Weight-gradient contraction over the N = n_steps * batch rows of the coordinate-major planes that sdeh_ctrl_backward[_ex] /
 * sdeh_bridge_div_backward write -- what the reference's autograd accumulates into Linear.weight.grad / .bias.grad through its
 * T per-step backward calls (models/mlp.py:114-122 under losses/oc.py:232-256):
 *   part_w[k][i][j] = sum_{n in chunk k} D[i][n] * act(Z[j][n])      part_b[k][i] = sum_{n in chunk k} D[i][n]
 * D [m, N] (m <= 256: d loss / d pre-activation of the layer above), Z [c, N] (c <= 256: pre-activations of the layer below;
 * act = SDEH_ACT_IDENTITY when Z already holds the layer input).  chunk: rows per partial, a multiple of 8;
 * n_chunks = ceil(N / chunk);  part_w [n_chunks, mp, cp] and part_b [n_chunks, mp] with mp = 64 ceil(m / 64), cp = 64 ceil(c / 64)
 * (64-channel networks: [n_chunks, 64, 64] / [n_chunks, 64]; rows >= m, columns >= c are zero).
 * The caller sums the partials over k (deterministic: no atomics).  One pass over D and Z per [64, 64] block of the product,
 * activation applied on the fly.
 */
int32_t sdeh_weight_grad(const float* D, int32_t m, const float* Z, int32_t c, int64_t N, int32_t act, int64_t chunk,
                         float* part_w, float* part_b, void* stream);

/*
 * Parameter gradients of a TimeEmbed sub-network (models/mlp.py:43-82: FourierMLP.timestep_embed, the gamma(t) network of the
 * score controls) from the gradient of its [n_steps, dim_out] table -- what the reference's autograd accumulates for these
 * parameters through its T per-step evaluations.  grad_flat (sdeh_time_embed_param_floats(te) floats) receives, in this order:
 *   timestep_phase [C] | for every hidden layer k: weight [C, 2C (k = 0) or C], bias [C] | out_layer.weight [dim_out, C] | bias [dim_out]
 * clip_out: the table entered the loss through clamp(-clip_out, clip_out) (reparam.py: clip(gamma(t)); +INF = no clamp): entries
 * outside carry no gradient.  1 <= te->n_hidden <= 4.  workspace: sdeh_time_embed_workspace_floats(te, n_steps) floats.
 */
int64_t sdeh_time_embed_param_floats(const SdehTimeEmbed* te);
int64_t sdeh_time_embed_workspace_floats(const SdehTimeEmbed* te, int32_t n_steps);
int32_t sdeh_time_embed_backward(const SdehTimeEmbed* te, int32_t activation, const float* ts, int32_t n_steps,
                                 const float* grad_table, float clip_out, float* workspace, float* grad_flat, void* stream);

/*
 * out[i][e] = sum_k part[i][k][e] for part [n_items, n_chunks, width] (the partials of sdeh_weight_grad: width 4096 / 64).
 * Deterministic two-pass sum without atomics or semaphores -- safe to replay inside a captured hipGraph, which the framework's
 * multi-block reduction is not on this stack.  scratch: n_items * ceil(n_chunks / 32) * width floats.
 */
int64_t sdeh_partial_sums_scratch_floats(int64_t n_items, int64_t n_chunks, int64_t width);
int32_t sdeh_partial_sums(const float* part, int64_t n_items, int64_t n_chunks, int64_t width, float* scratch, float* out,
                          void* stream);

/* Philox4x32-10 known-answer hook used by the tests: fills out[4*n] with the generator's raw words for
 * counters (row_offset+i, step, block, offset) and the (seed) key -- the exact stream sdeh_simulate_fwd consumes. */
int32_t sdeh_debug_philox(uint64_t seed, uint64_t offset, int64_t row_offset, int32_t step, int32_t block,
                          int64_t n, uint32_t* out, void* stream);
/* Standard normals exactly as sdeh_simulate_fwd draws them: out[n, d] for one step. */
int32_t sdeh_debug_normals(uint64_t seed, uint64_t offset, int64_t row_offset, int32_t step, int32_t dim,
                           int64_t n, float* out, void* stream);
/* The kernel's branch-free GELU (exact-erf form to fp32 rounding), elementwise on n values. */
int32_t sdeh_debug_gelu(const float* in, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDEH_H_ */
