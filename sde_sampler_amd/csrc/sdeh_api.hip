// Host side of libsdeh.so: the extern "C" entry points of include/sdeh.h, plan/workspace management and the
// dispatch onto the compiled trajectory-kernel variants.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "sdeh_common.hpp"

namespace sdeh {

// ---- launchers defined in other translation units --------------------------------------------------------
int launch_prep(const PrepArgs& p, hipStream_t stream);
int launch_reduce(const float* rnd, long long n, float max_rnd, float* part, int nb, float* out, hipStream_t stream);
int launch_weights(const float* rnd, long long n, const float* mx, float* w, hipStream_t stream);
int launch_loss_moment(const float* rnd, long long n, float max_rnd, int lv, long long* n_filtered, float* part, int nb, float* out, float* w,
                       hipStream_t stream);
int launch_guard_restore(const unsigned long long* table, int n_tensors, const unsigned char* ok, long long* n_skipped, hipStream_t stream);
int launch_guard_check(float* g, long long n, const float* value, float max_loss, unsigned char* ok, hipStream_t stream);
int launch_sink_init(float* u, float* v, float* log_a, float* log_b, const float* w_x, const float* w_y, long long n,
                     long long m, float eps, int* flags, hipStream_t st);
int launch_sink_finalize(const float* pm, const float* ps, int splits, long long np, const float* log_w, float eps, float* pot,
                         int* err_bits, const int* done, hipStream_t st);
int launch_sink_check(int* flags, float thresh, hipStream_t st);
int launch_sink_dist_final(const float* part, int nb, const int* flags, float* out, hipStream_t st);
int launch_weight_grad(const float* D, int m, const float* Z, int c, long long N, int act, long long chunk, float* part_w,
                       float* part_b, hipStream_t stream);
int launch_partial_sums(const float* part, long long n_items, long long n_chunks, long long width, float* scratch, float* out,
                        hipStream_t stream);
int launch_time_embed_bwd(const SdehTimeEmbed& te, int act, const float* ts, int n_steps, const float* gout, float clip,
                          float* workspace, float* grad_flat, hipStream_t stream);
long long time_embed_param_floats(const SdehTimeEmbed& te);
long long time_embed_workspace_floats(const SdehTimeEmbed& te, int n_steps);
int launch_sample_stats(const float* x, const float* w, const float* domain, long long B, int d, float* scratch, int nb,
                        float* out, hipStream_t st);
int launch_wide(const TrajArgs& a, hipStream_t stream, int* ct_used);           // sdeh_wide.hip
int launch_wide_bwd(const BwdArgs& a, hipStream_t stream);                      // sdeh_wide_bwd.hip
int launch_wide_div_bwd(const WideDivArgs& a, hipStream_t stream);              // sdeh_wide_bwd.hip
int wide_div_grid(long long batch, int n_steps);                                // sdeh_wide_bwd.hip
int launch_bridge_wide(const TrajArgs& a, hipStream_t stream, int* split_used, float* scratch);  // sdeh_wide.hip
long long bridge_wide_scratch_floats(long long batch);                                           // sdeh_wide.hip

#define SDEH_DECL(dp, pad, tag, loss, ctrl, tgt, gmm, act, refc, gnv)                           \
  int launch_ws_dp##dp##_p##pad##_##tag(const TrajArgs& a, hipStream_t stream);                 \
  int launch_legacy_dp##dp##_p##pad##_##tag(const TrajArgs& a, hipStream_t stream);             \
  int launch_bwd_dp##dp##_p##pad##_##tag(const BwdArgs& a, hipStream_t stream);                  \
  int launch_int_dp##dp##_p##pad##_##tag(const TrajArgs& a, hipStream_t stream);                  \
  int launch_sink_dp##dp##_p##pad##_##tag(const SinkArgs& a, int mode, int splits, hipStream_t stream); \
  int launch_bridge_dp##dp##_p##pad##_##tag(const TrajArgs& a, hipStream_t stream);              \
  int launch_bridge_bwd_dp##dp##_p##pad##_##tag(const BridgeBwdArgs& a, hipStream_t stream);
#include "sdeh_variants.inc"
#undef SDEH_DECL

struct Variant {
  int dp;
  bool pad;
  int loss, ctrl, tgt, gmm, act, refc;  // -1 = run-time switch
  int gnv;                              // > 0: shared-scale mixture tables cover only the first gnv coordinates
  TrajLauncher fn;         // wave-specialised kernel (sdeh_traj_ws.hpp)
  TrajLauncher fn_legacy;  // single-wave kernel (sdeh_traj.hpp); returns SDEH_ERR_UNSUPPORTED when not compiled in
  int (*fn_bwd)(const BwdArgs&, hipStream_t);  // control-network backward (sdeh_bwd.hpp), generic variants only
  TrajLauncher fn_int;     // plain Euler integrator (sdeh_integrate.hpp), generic variants only
  SinkLauncher fn_sink;    // Sinkhorn sweeps (sdeh_sinkhorn.hpp), generic variants only
  TrajLauncher fn_bridge;  // TimeReversalLoss with an inference control (sdeh_bridge.hpp), generic variants only
  int (*fn_bridge_bwd)(const BridgeBwdArgs&, hipStream_t);  // gradient of the divergence term (sdeh_bridge.hpp)
  const char* name;
};
static const Variant kVariants[] = {
#define SDEH_DECL(dp, pad, tag, loss, ctrl, tgt, gmm, act, refc, gnv) {dp, pad != 0, loss, ctrl, tgt, gmm, act, refc, gnv, &launch_ws_dp##dp##_p##pad##_##tag, &launch_legacy_dp##dp##_p##pad##_##tag, &launch_bwd_dp##dp##_p##pad##_##tag, &launch_int_dp##dp##_p##pad##_##tag, &launch_sink_dp##dp##_p##pad##_##tag, &launch_bridge_dp##dp##_p##pad##_##tag, &launch_bridge_bwd_dp##dp##_p##pad##_##tag, #dp "_" #pad "_" #tag},
#include "sdeh_variants.inc"
#undef SDEH_DECL
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static bool is_generic(const Variant& v) { return v.loss < 0 && v.ctrl < 0 && v.tgt < 0 && v.gmm < 0 && v.act < 0 && v.refc < 0 && v.gnv <= 0; }

// generic variant for dimension d: the exact one when compiled, else the smallest padded one
static const Variant* pick_variant(int d) {
  const Variant* best = nullptr;
  for (const Variant& v : kVariants) {
    if (!is_generic(v)) continue;
    if (!v.pad && v.dp == d) return &v;
    if (v.pad && v.dp >= d && (best == nullptr || v.dp < best->dp)) best = &v;
  }
  return best;
}

// specialised variant (of the plan's dimension class) whose compile-time choices all match the problem, if any
// gmm: 0 = no LDS tables possible, 1 = general, 2 = shared scale; nvary: varying-prefix length promised by the caller (-1 none)
// generic_only: only variants that fix nothing but the table form (tags "g<n>": run-time switches for everything else)
static const Variant* pick_specialised(const Variant* cls, int loss, int ctrl, int tgt, int gmm, int act, int refc, int nvary, bool generic_only) {
  const Variant* best = nullptr;
  int best_fixed = -1;
  for (const Variant& v : kVariants) {
    if (is_generic(v) || v.pad != cls->pad || v.dp != cls->dp) continue;
    if (!((v.loss < 0 || v.loss == loss) && (v.ctrl < 0 || v.ctrl == ctrl) && (v.tgt < 0 || v.tgt == tgt) &&
          (v.gmm < 0 || v.gmm == gmm) && (v.act < 0 || v.act == act) && (v.refc < 0 || v.refc == refc)))
      continue;
    const int fixed = (v.loss >= 0) + (v.ctrl >= 0) + (v.tgt >= 0) + (v.gmm >= 0) + (v.act >= 0) + (v.refc >= 0);
    if (generic_only && fixed > 0) continue;
    if (v.gnv > 0 && !(gmm == 2 && nvary >= 0 && nvary <= v.gnv)) continue;
    // prefer the variant whose mixture tables cover the fewest coordinates, then the one with the most compile-time choices
    const bool fewer = v.gnv > 0 && (best == nullptr || best->gnv <= 0 || v.gnv < best->gnv);
    const bool same = best != nullptr && ((v.gnv > 0) == (best->gnv > 0)) && (v.gnv <= 0 || v.gnv == best->gnv);
    if (best == nullptr || fewer || (same && fixed > best_fixed)) { best = &v; best_fixed = fixed; }
  }
  return best;
}

static int align4(int v) { return (v + 3) & ~3; }
static constexpr int SDEH_MM_ROWS = 40;  // = SDEH_MM_K of sdeh_traj_ws.hpp: component rows of the matrix-pipe mixture's instruction stream

// Workspace layout for one problem geometry.
// gmm_nv: number of leading coordinates the shared-scale mixture tables cover (multiple of 4; 0 = all)
// with_bwd: also pack the transposed weights the backward kernel needs (they join the LDS image)
static WsLayout make_layout(int dp, int c, int n_hidden, int t_max, int k_max, int g, bool shared_scale = false,
                            bool gmm_global = false, int gmm_nv = 0, bool with_bwd = false, bool with_tan = false, bool gmm_mm = false,
                            bool out4 = false) {
  WsLayout L;
  memset(&L, 0, sizeof(L));
  L.dp = dp; L.c = c; L.ot = c / 32; L.otd = row_tiles(dp); L.r_in = mregs(dp);
  L.n_hidden = n_hidden; L.t_max = t_max; L.k_max = k_max; L.g = g;
  int o = 0;
  L.w_in = o; o += L.r_in * L.ot * 64;
  L.w_hid = o; L.w_hid_stride = (c / 2) * L.ot * 64; o += n_hidden * L.w_hid_stride;
  L.w_out = o; o += (c / 2) * L.otd * 64;
  L.b_hid = o; o += n_hidden * c;
  L.b_out = o; o += L.otd * 32;
  o = align4(o);
  L.wt_out = L.wt_hid = L.wt_in = -1;
  if (with_bwd) {
    L.wt_out = o; o += L.r_in * L.ot * 64;
    L.wt_hid = o; o += n_hidden * L.w_hid_stride;
    L.wt_in = o; o += (c / 2) * L.otd * 64;
  }
  L.gmm_row = 2 * ((dp + 1) & ~1);
  // table rows padded to a multiple of 8 (padding rows: logit -inf); the matrix-pipe mixture's instruction stream has SDEH_MM_ROWS rows
  const int k_rows = gmm_mm && k_max <= SDEH_MM_ROWS ? SDEH_MM_ROWS : (k_max + 7) & ~7;
  L.gmm_rows = k_rows;
  // shared-scale tables: one word per (k,d), rows of 4*ceil(dp/4) floats (gmm_nv > 0: of the first gmm_nv coordinates only), then the
  // per-coordinate vectors; general tables: (mu, a) pairs
  const int rs_full = 4 * ((dp + 3) / 4);
  const int rs = gmm_nv > 0 ? gmm_nv : rs_full;
  const int gmm_floats = shared_scale ? 2 * k_rows * rs + 4 * rs_full + align4(k_rows) : 2 * k_rows * L.gmm_row + align4(k_rows);
  // LDS budget of the wave-specialised kernel: image + four [coordinate][64] exchange buffers within 160 KiB
  const size_t xbuf_floats = with_bwd ? 0 : (size_t)4 * (mdim(mregs(dp) - 1, 1) + 1) * 64;
  L.gmm_lds = (k_max > 0 && !gmm_global && ((size_t)(o + gmm_floats) + xbuf_floats) * sizeof(float) <= 160 * 1024) ? 1 : 0;
  L.gmm_mm1 = L.gmm_mm2 = L.gmm_cc = -1;
  // matrix-pipe mixture (full tables; shared scale: form 3, per-component scales: form 4 with twice the instructions): the A-operand
  // images take the tables' place in the LDS image
  const int mm_f = shared_scale ? 1 : 2;
  const int mm1_floats = ((mm_f * dp * (k_rows / 4) + 63) / 64) * 256, mm2_floats = ((mm_f * k_rows * ((dp + 3) / 4) + 63) / 64) * 256;
  const bool mm = gmm_mm && gmm_nv <= 0 && k_max > 0 && !gmm_global && !with_bwd &&
                  ((size_t)(o + mm1_floats + mm2_floats + align4(k_rows)) + xbuf_floats) * sizeof(float) <= 160 * 1024;
  if (mm) {
    L.gmm_lds = shared_scale ? 3 : 4;
    if (shared_scale) L.gmm_row = rs;
    L.gmm_mm1 = o; o += mm1_floats;
    L.gmm_mm2 = o; o += mm2_floats;
    L.gmm_cc = o; o += align4(k_rows);
  } else if (L.gmm_lds && shared_scale) {
    L.gmm_lds = 2;
    L.gmm_row = rs;
    L.gmm_lg = o; o += k_rows * rs;
    L.gmm_sc = o; o += k_rows * rs;
    L.gmm_vec = o; o += 4 * rs_full;
    L.gmm_c = o; o += align4(k_rows);
  } else if (L.gmm_lds) {
    L.gmm_lg = o; o += k_rows * L.gmm_row;
    L.gmm_sc = o; o += k_rows * L.gmm_row;
    L.gmm_c = o; o += align4(k_rows);
  }
  // the out layer's 4 x 4 x 1 operand image (WsLayout::w_out4): whole-wave evaluation launches, when it still fits
  L.w_out4 = -1;
  {
    if (out4 && !with_bwd && dp > 4 && dp <= 16 && c == 64 &&  // = ws_out4_compiled<DP>()
        ((size_t)(o + out4_floats(dp)) + xbuf_floats) * sizeof(float) + 64 <= 160 * 1024) {
      L.w_out4 = o;
      o += out4_floats(dp);
    }
  }
  L.lds_floats = align4(o);
  o = L.lds_floats;
  L.coef = o; o += t_max * kCoefStride;
  L.emb = o; o += t_max * c;
  L.gam = o; o += align4(t_max * g);
  L.out_cnt = o; o += align4(t_max + 1);
  L.tan_in = L.tan_out = -1;
  if (with_tan) {
    L.tan_in = o; o += dp * c;
    L.tan_out = o; o += dp * c;
  }
  if (L.gmm_lds == 3) {  // the shared-scale tables next to the image (64-byte aligned: read as s_load_dwordx16)
    o = (o + 15) & ~15;
    L.gmm_lg = o; o += k_rows * rs;
    L.gmm_sc = o; o += k_rows * rs;
    L.gmm_vec = o; o += 4 * rs_full;
    L.gmm_c = o; o += align4(k_rows);
  } else if (L.gmm_lds == 4) {  // the general tables
    o = (o + 15) & ~15;
    L.gmm_lg = o; o += k_rows * L.gmm_row;
    L.gmm_sc = o; o += k_rows * L.gmm_row;
    L.gmm_c = o; o += align4(k_rows);
  } else if (!L.gmm_lds) {
    L.gmm_lg = o; o += k_rows * L.gmm_row;
    L.gmm_sc = o; o += k_rows * L.gmm_row;
    L.gmm_c = o; o += align4(k_rows);
  }
  for (int i = 0; i < 3; ++i) { L.dg[i] = o; o += align4(2 * dp + 1); }
  L.total = align4(o);
  return L;
}

// Workspace layout of a wide network (C in {128, 256}, d <= 256; sdeh_wide.hip): everything lives in global memory.
// with_bwd: also the transposed images of the training backward (sdeh_wide_bwd.hip): out_layer^T packed like an input layer
// (k = coordinates), the hidden layers transposed, input_embed^T packed like an out layer (rows = coordinates)
static WsLayout make_wide_layout(int d, int c, int n_hidden, int t_max, int g, bool with_tan, int k_max = 0, bool with_bwd = false) {
  WsLayout L;
  memset(&L, 0, sizeof(L));
  L.wide = 1;
  L.otd = row_tiles(d);
  L.dp = 32 * L.otd;  // padded coordinate count of the per-coordinate tables
  L.dp8 = (d + 7) & ~7;
  L.c = c; L.ot = c / 32; L.r_in = 0;
  L.n_hidden = n_hidden; L.t_max = t_max; L.k_max = 0; L.g = g;
  int o = 0;
  L.w_in = o; o += (L.dp8 / 8) * L.ot * 256;
  L.w_hid = o; L.w_hid_stride = (c / 8) * L.ot * 256; o += n_hidden * L.w_hid_stride;
  L.w_out = o; o += (c / 8) * L.otd * 256;
  L.b_hid = o; o += n_hidden * c;
  L.b_out = o; o += L.otd * 32;
  L.wt_out = L.wt_hid = L.wt_in = -1;
  L.tan_in = L.tan_out = -1;
  if (with_tan || with_bwd) { L.wt_hid = o; o += n_hidden * L.w_hid_stride; }
  if (with_tan) {
    L.tan_in = o; o += d * c;
    L.tan_out = o; o += d * c;
  }
  if (with_bwd) {
    L.wt_out = o; o += (L.dp8 / 8) * L.ot * 256;
    L.wt_in = o; o += (c / 8) * L.otd * 256;
  }
  L.lds_floats = 0;
  L.coef = o; o += t_max * kCoefStride;
  L.emb = o; o += t_max * c;
  L.gam = o; o += align4(t_max * g);
  L.out_cnt = o; o += align4(t_max + 1);
  for (int i = 0; i < 3; ++i) { L.dg[i] = o; o += align4(2 * L.dp + 1); }
  // mixture target: mu[K][d4], a = 1/(2 sigma^2) [K][d4] (rows padded to four coordinates with zeros), c[K]
  L.k_max = k_max; L.gmm_lds = 0; L.gmm_row = align4(d); L.gmm_rows = k_max;
  L.gmm_lg = o; o += k_max * L.gmm_row;
  L.gmm_sc = o; o += k_max * L.gmm_row;
  L.gmm_c = o; o += align4(k_max);
  L.total = align4(o);
  return L;
}

}  // namespace sdeh

using namespace sdeh;

struct SdehPlan {
  SdehPlanDesc desc;
  int device;
  const Variant* variant;  // narrow networks (C = 64, d <= 64); null for wide plans
  bool wide;               // C in {128, 256}, d <= 256: the channel-split kernels of sdeh_wide.hip / sdeh_wide_bwd.hip
  float* ws;          // workspace
  size_t ws_floats;
  float* scratch;     // wide Bridge only: per-(column tile, coordinate group) divergence sums (grown on first use at a larger batch)
  size_t scratch_floats;
  bool timing;
  bool timed;
  hipEvent_t ev0, ev1;     // the event pair of the newest timed launch (an entry of the ring below)
  char last_kernel[96];
  // a ring of event pairs: the durations of the last kTimingRing timed launches can be read AFTER a run (sdeh_plan_timing_entry)
  // instead of with one host synchronisation per launch
  static constexpr int kTimingRing = 128;
  hipEvent_t ring0[kTimingRing], ring1[kTimingRing];
  char ring_names[kTimingRing][96];
  int ring_pos, ring_count;
  PlanOptions opts;  // kernel-mode options (sdeh_plan_set_option)
};

// options of the plan whose entry point runs on this thread
static thread_local const PlanOptions* tl_opts = nullptr;
const char* sdeh::plan_opt(OptKey key) { return tl_opts != nullptr && tl_opts->v[key][0] != 0 ? tl_opts->v[key] : nullptr; }
struct OptScope {
  const PlanOptions* prev;
  explicit OptScope(const SdehPlan* plan) : prev(tl_opts) { tl_opts = plan != nullptr ? &plan->opts : nullptr; }
  ~OptScope() { tl_opts = prev; }
};
static const char* const kOptNames[OPT_COUNT] = {
    "SDEH_LEGACY", "SDEH_GENERIC_ONLY", "SDEH_WS_GROUPS", "SDEH_WS_QUAD", "SDEH_WS_VOUT", "SDEH_WS_BARRIER", "SDEH_BWD_PLANES", "SDEH_BWD_TILE",
    "SDEH_BWD_WAVES", "SDEH_BWD_V1", "SDEH_BWD_V2", "SDEH_BWD_NO_VIO", "SDEH_BWD_SCAN", "SDEH_BWD_ZREC", "SDEH_BRIDGE_TILES", "SDEH_BRIDGE_SPLIT", "SDEH_WIDE_CT", "SDEH_WIDE_SPLIT", "SDEH_GMM_MM", "SDEH_WS_OUT4"};
static void opt_store(PlanOptions& o, int key, const char* value) {
  memset(o.v[key], 0, sizeof(o.v[key]));
  if (value != nullptr) strncpy(o.v[key], value, sizeof(o.v[key]) - 1);
}
static int reserve_scratch(SdehPlan* plan, size_t need);
static constexpr int kRedBlocks = SDEH_REDUCE_SCRATCH / 8;

extern "C" {

int32_t sdeh_abi_version(void) { return SDEH_ABI_VERSION; }
const char* sdeh_last_error(void) { return g_err; }

int32_t sdeh_plan_create(const SdehPlanDesc* desc, SdehPlan** out) {
  if (desc == nullptr || out == nullptr) return fail(SDEH_ERR_INVALID, "plan_create: null argument");
  *out = nullptr;
  if (desc->dim < 1) return fail(SDEH_ERR_INVALID, "plan_create: dim=%d", desc->dim);
  // wide-network kernels: 128 / 256 channels at any d <= 256, and 64 channels once the state no longer fits the d <= 64 kernels
  const bool wide = desc->channels == 128 || desc->channels == 256 || (desc->channels == 64 && desc->dim > 64 && desc->dim <= 256);
  if (desc->channels != 64 && !wide)
    return fail(SDEH_ERR_UNSUPPORTED, "plan_create: channels=%d (trajectory kernels are compiled for C = 64, 128 and 256)",
                desc->channels);
  if (desc->max_hidden < 0 || desc->max_hidden > SDEH_MAX_HIDDEN)
    return fail(SDEH_ERR_UNSUPPORTED, "plan_create: %d hidden layers (max %d)", desc->max_hidden, SDEH_MAX_HIDDEN);
  if (desc->max_steps < 1) return fail(SDEH_ERR_INVALID, "plan_create: max_steps=%d", desc->max_steps);
  const int k_max = desc->max_components > 0 ? desc->max_components : 0;
  const Variant* v = nullptr;
  size_t ws_floats = 0;
  if (wide) {
    if (desc->dim > 256) return fail(SDEH_ERR_UNSUPPORTED, "plan_create: dim=%d (the wide-network kernels cover d <= 256)", desc->dim);
    // two regions (generative + inference network of a Bridge), each with the tangent tables / transposed hidden layers
    ws_floats = 2 * (size_t)make_wide_layout(desc->dim, desc->channels, desc->max_hidden, desc->max_steps, 32 * row_tiles(desc->dim), true, k_max, true).total + 64;
  } else {
    v = pick_variant(desc->dim);
    if (v == nullptr)
      return fail(SDEH_ERR_UNSUPPORTED, "plan_create: no trajectory kernel compiled for dim=%d with channels=64", desc->dim);
    WsLayout L = make_layout(v->dp, desc->channels, desc->max_hidden, desc->max_steps, k_max, v->dp);
    // LDS: the wave-specialised kernel needs image + exchange buffers, the single-wave kernel image + logit scratch;
    // a plan is usable when either fits (deep networks fall back to the single-wave kernel at launch time)
    const size_t img = (size_t)L.lds_floats * sizeof(float);
    const size_t lds_ws = img + (size_t)4 * (mdim(mregs(v->dp) - 1, 1) + 1) * 64 * sizeof(float);
    const size_t lds_legacy = img + (size_t)k_max * 256 * sizeof(float);
    if ((lds_ws < lds_legacy ? lds_ws : lds_legacy) > 160 * 1024)
      return fail(SDEH_ERR_UNSUPPORTED, "plan_create: needs %zu B of LDS (> 160 KiB): hidden=%d K=%d",
                  lds_ws < lds_legacy ? lds_ws : lds_legacy, desc->max_hidden, k_max);
    // region 1: the largest layout of any call (backward packs transposed weights too); region 2: the Bridge paths pack a
    // second (inference) network with transposed weights and tangent tables
    ws_floats = (size_t)make_layout(v->dp, desc->channels, desc->max_hidden, desc->max_steps, k_max, v->dp, false, false, 0, true).total +
                (size_t)make_layout(v->dp, desc->channels, desc->max_hidden, desc->max_steps, 0, v->dp, false, true, 0, true, true).total + 64;
  }
  SdehPlan* p = new (std::nothrow) SdehPlan();
  if (p == nullptr) return fail(SDEH_ERR_INVALID, "plan_create: out of host memory");
  p->desc = *desc;
  p->device = desc->device;
  p->variant = v;
  p->wide = wide;
  p->ws_floats = ws_floats;
  p->scratch = nullptr;
  p->scratch_floats = 0;
  p->timing = p->timed = false;
  p->ev0 = p->ev1 = nullptr;
  p->last_kernel[0] = 0;
  for (int i = 0; i < SdehPlan::kTimingRing; ++i) { p->ring0[i] = p->ring1[i] = nullptr; p->ring_names[i][0] = 0; }
  p->ring_pos = -1;
  p->ring_count = 0;
  for (int k = 0; k < OPT_COUNT; ++k) opt_store(p->opts, k, getenv(kOptNames[k]));  // the environment is a test override, read ONCE here
  int prev = 0;
  hipError_t e = hipGetDevice(&prev);
  if (e == hipSuccess) e = hipSetDevice(desc->device);
  if (e == hipSuccess) e = hipMalloc((void**)&p->ws, p->ws_floats * sizeof(float));
  if (e == hipSuccess) e = hipMemset(p->ws, 0, p->ws_floats * sizeof(float));
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    const int rc = fail(SDEH_ERR_HIP, "plan_create: %s", hipGetErrorString(e));
    if (p->ws) (void)hipFree(p->ws);
    delete p;
    return rc;
  }
  if (wide && desc->max_batch > 0) {
    const int rc = reserve_scratch(p, (size_t)bridge_wide_scratch_floats(desc->max_batch));
    if (rc != SDEH_OK) { (void)hipFree(p->ws); delete p; return rc; }
  }
  *out = p;
  return SDEH_OK;
}

// (re)allocates the wide Bridge's divergence scratch: 32 partial sums per trajectory.  Synchronises the device.
static int reserve_scratch(SdehPlan* plan, size_t need) {
  if (need <= plan->scratch_floats) return SDEH_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(plan->device);
  if (e == hipSuccess && plan->scratch != nullptr) e = hipFree(plan->scratch);
  plan->scratch = nullptr;
  plan->scratch_floats = 0;
  if (e == hipSuccess) e = hipMalloc((void**)&plan->scratch, need * sizeof(float));
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return fail(SDEH_ERR_HIP, "plan_reserve: scratch allocation failed: %s", hipGetErrorString(e));
  plan->scratch_floats = need;
  return SDEH_OK;
}

int32_t sdeh_plan_reserve(SdehPlan* plan, int64_t max_batch) {
  if (plan == nullptr || max_batch < 0) return fail(SDEH_ERR_INVALID, "plan_reserve: bad argument");
  if (!plan->wide || max_batch == 0) return SDEH_OK;  // (only the wide Bridge has batch-dependent scratch)
  return reserve_scratch(plan, (size_t)bridge_wide_scratch_floats(max_batch));
}

int32_t sdeh_plan_set_option(SdehPlan* plan, const char* name, const char* value) {
  if (plan == nullptr || name == nullptr) return fail(SDEH_ERR_INVALID, "plan_set_option: null argument");
  for (int k = 0; k < OPT_COUNT; ++k)
    if (strcmp(name, kOptNames[k]) == 0) {
      opt_store(plan->opts, k, value);
      return SDEH_OK;
    }
  return fail(SDEH_ERR_INVALID, "plan_set_option: unknown option %s", name);
}

int32_t sdeh_plan_set_timing(SdehPlan* plan, int32_t enable) {
  if (plan == nullptr) return fail(SDEH_ERR_INVALID, "plan_set_timing: null plan");
  if (enable && plan->ring0[0] == nullptr) {
    for (int i = 0; i < SdehPlan::kTimingRing; ++i)
      if (hipEventCreate(&plan->ring0[i]) != hipSuccess || hipEventCreate(&plan->ring1[i]) != hipSuccess)
        return fail(SDEH_ERR_HIP, "plan_set_timing: hipEventCreate failed");
    plan->ev0 = plan->ring0[0];
    plan->ev1 = plan->ring1[0];
  }
  plan->timing = enable != 0;
  plan->timed = false;
  plan->ring_pos = -1;
  plan->ring_count = 0;
  return SDEH_OK;
}

// the event pair of a timed launch: the next entry of the ring (the previous entry keeps the name of the launch it timed)
static void timing_begin(SdehPlan* plan, hipStream_t st) {
  if (!plan->timing) return;
  if (plan->ring_pos >= 0) snprintf(plan->ring_names[plan->ring_pos], sizeof(plan->ring_names[0]), "%s", plan->last_kernel);
  plan->ring_pos = (plan->ring_pos + 1) % SdehPlan::kTimingRing;
  if (plan->ring_count < SdehPlan::kTimingRing) ++plan->ring_count;
  plan->ev0 = plan->ring0[plan->ring_pos];
  plan->ev1 = plan->ring1[plan->ring_pos];
  (void)hipEventRecord(plan->ev0, st);
}
static void timing_end(SdehPlan* plan, hipStream_t st) {
  if (!plan->timing) return;
  (void)hipEventRecord(plan->ev1, st);
  plan->timed = true;
}

int32_t sdeh_plan_timing_entry(SdehPlan* plan, int32_t back, float* ms, char* name, int32_t name_len) {
  if (plan == nullptr || ms == nullptr || back < 0) return fail(SDEH_ERR_INVALID, "plan_timing_entry: bad argument");
  if (!plan->timed || back >= plan->ring_count) return 1;  // no such entry (yet)
  const int pos = ((plan->ring_pos - back) % SdehPlan::kTimingRing + SdehPlan::kTimingRing) % SdehPlan::kTimingRing;
  hipError_t e = hipEventSynchronize(plan->ring1[pos]);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, plan->ring0[pos], plan->ring1[pos]);
  if (e != hipSuccess) return fail(SDEH_ERR_HIP, "plan_timing_entry: %s", hipGetErrorString(e));
  if (name != nullptr && name_len > 0) snprintf(name, (size_t)name_len, "%s", back == 0 ? plan->last_kernel : plan->ring_names[pos]);
  return SDEH_OK;
}

int32_t sdeh_plan_last_kernel_ms(SdehPlan* plan, float* ms) {
  if (plan == nullptr || ms == nullptr) return fail(SDEH_ERR_INVALID, "plan_last_kernel_ms: null argument");
  if (!plan->timed) return fail(SDEH_ERR_INVALID, "plan_last_kernel_ms: no timed launch yet");
  hipError_t e = hipEventSynchronize(plan->ev1);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, plan->ev0, plan->ev1);
  return e == hipSuccess ? SDEH_OK : fail(SDEH_ERR_HIP, "plan_last_kernel_ms: %s", hipGetErrorString(e));
}

const char* sdeh_plan_last_kernel_name(SdehPlan* plan) { return plan == nullptr ? "" : plan->last_kernel; }

void sdeh_plan_destroy(SdehPlan* plan) {
  if (plan == nullptr) return;
  for (int i = 0; i < SdehPlan::kTimingRing; ++i) {
    if (plan->ring0[i]) (void)hipEventDestroy(plan->ring0[i]);
    if (plan->ring1[i]) (void)hipEventDestroy(plan->ring1[i]);
  }
  if (plan->ws) (void)hipFree(plan->ws);
  if (plan->scratch) (void)hipFree(plan->scratch);
  delete plan;
}

static int check_time_embed(const SdehTimeEmbed& te, int channels, const char* what) {
  if (te.channels != channels) return fail(SDEH_ERR_UNSUPPORTED, "%s: channels=%d != %d", what, te.channels, channels);
  if (te.n_hidden < 1 || te.n_hidden > SDEH_MAX_HIDDEN) return fail(SDEH_ERR_INVALID, "%s: n_hidden=%d", what, te.n_hidden);
  if (te.coeff == nullptr || te.phase == nullptr || te.out_w == nullptr || te.out_b == nullptr)
    return fail(SDEH_ERR_INVALID, "%s: null parameter pointer", what);
  for (int i = 0; i < te.n_hidden; ++i)
    if (te.hidden_w[i] == nullptr || te.hidden_b[i] == nullptr) return fail(SDEH_ERR_INVALID, "%s: null hidden layer %d", what, i);
  return SDEH_OK;
}

static int check_density(const SdehDensity& D, int d, const char* what, bool allow_none) {
  switch (D.kind) {
    case SDEH_DENS_NONE:
      return allow_none ? SDEH_OK : fail(SDEH_ERR_INVALID, "%s density missing", what);
    case SDEH_DENS_GMM:
      if (D.loc == nullptr || D.scale == nullptr || D.n_components < 1) return fail(SDEH_ERR_INVALID, "%s: bad GMM", what);
      break;
    case SDEH_DENS_DIAG_GAUSS:
      if (D.loc == nullptr || D.scale == nullptr) return fail(SDEH_ERR_INVALID, "%s: bad Gaussian", what);
      break;
    case SDEH_DENS_MULTI_WELL:
      if (D.n_components < 1 || D.n_components > d) return fail(SDEH_ERR_INVALID, "%s: n_double_wells=%d", what, D.n_components);
      break;
    case SDEH_DENS_FUNNEL:
      if (!(D.p0 > 0.0f) || d < 2) return fail(SDEH_ERR_INVALID, "%s: bad funnel", what);
      break;
    case SDEH_DENS_EXTERNAL:  // (the score arrives per step through sdeh_simulate_fwd_steps; check_problem: wide plans only)
      break;
    default:
      return fail(SDEH_ERR_INVALID, "%s: unknown density kind %d", what, D.kind);
  }
  if (D.dim != d) return fail(SDEH_ERR_INVALID, "%s: dim=%d != %d", what, D.dim, d);
  return SDEH_OK;
}

// Validation shared by the forward and backward entry points; on success fills the layout and the chosen variant.
struct Checked {
  WsLayout L;
  const Variant* v;
  bool refc;
  int g, k;
};
static int check_problem(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, int64_t batch,
                         int64_t row_offset, bool backward, Checked* out, bool integrate = false) {
  if (plan == nullptr || pr == nullptr || ts == nullptr) return fail(SDEH_ERR_INVALID, "null argument");
  if (batch < 1 || n_steps < 1) return fail(SDEH_ERR_INVALID, "simulate_fwd: batch=%lld n_steps=%d", (long long)batch, n_steps);
  if (row_offset < 0 || (unsigned long long)row_offset + (unsigned long long)batch > 0x100000000ull)
    return fail(SDEH_ERR_INVALID, "simulate_fwd: global row indices must fit 32 bits (row_offset=%lld batch=%lld)",
                (long long)row_offset, (long long)batch);
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  if (d != plan->desc.dim || net.channels != plan->desc.channels)
    return fail(SDEH_ERR_CAPACITY, "simulate_fwd: dim/channels (%d,%d) differ from the plan's (%d,%d)", d, net.channels,
                plan->desc.dim, plan->desc.channels);
  if (net.n_hidden < 0 || net.n_hidden > plan->desc.max_hidden)
    return fail(SDEH_ERR_CAPACITY, "simulate_fwd: %d hidden layers > plan max %d", net.n_hidden, plan->desc.max_hidden);
  if (n_steps > plan->desc.max_steps)
    return fail(SDEH_ERR_CAPACITY, "simulate_fwd: %d steps > plan max %d", n_steps, plan->desc.max_steps);
  if (net.activation < SDEH_ACT_GELU_ERF || net.activation > SDEH_ACT_RELU)
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd: activation %d", net.activation);
  if (net.input_w == nullptr || net.input_b == nullptr || net.out_w == nullptr || net.out_b == nullptr)
    return fail(SDEH_ERR_INVALID, "simulate_fwd: null base_model parameter");
  for (int i = 0; i < net.n_hidden; ++i)
    if (net.hidden_w[i] == nullptr || net.hidden_b[i] == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd: null hidden layer %d", i);
  int rc = check_time_embed(net.timestep_embed, net.channels, "base_model.timestep_embed");
  if (rc != SDEH_OK) return rc;
  if (net.timestep_embed.dim_out != net.channels) return fail(SDEH_ERR_INVALID, "timestep_embed.dim_out != channels");
  if (pr->loss_kind < SDEH_LOSS_TIME_REVERSAL || pr->loss_kind > SDEH_LOSS_EXPONENTIAL)
    return fail(SDEH_ERR_INVALID, "simulate_fwd: loss_kind %d", pr->loss_kind);
  if (pr->ctrl_kind < SDEH_CTRL_CLIPPED || pr->ctrl_kind > SDEH_CTRL_LERP_PRIOR)
    return fail(SDEH_ERR_INVALID, "simulate_fwd: ctrl_kind %d", pr->ctrl_kind);
  if ((pr->flags & SDEH_FLAG_INFERENCE_CTRL) && (backward || integrate || pr->loss_kind != SDEH_LOSS_TIME_REVERSAL))
    return fail(SDEH_ERR_UNSUPPORTED, "an inference control is only evaluated forward, inside TimeReversalLoss (Bridge)");
  if ((pr->flags & SDEH_FLAG_INFERENCE_SDE) && !integrate)
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd: the losses integrate the generative SDE (SDEH_FLAG_INFERENCE_SDE is for sdeh_integrate)");
  if (pr->loss_kind != SDEH_LOSS_EXPONENTIAL && pr->sde_kind == SDEH_SDE_NONE)
    return fail(SDEH_ERR_INVALID, "simulate_fwd: loss kind %d needs an sde", pr->loss_kind);
  const bool lerp_family = pr->ctrl_kind >= SDEH_CTRL_LERP;
  if (lerp_family && pr->sde_kind == SDEH_SDE_NONE) return fail(SDEH_ERR_INVALID, "simulate_fwd: Lerp controls need an sde");
  int g = 1;
  if (pr->ctrl_kind != SDEH_CTRL_CLIPPED && pr->score_model.n_hidden > 0) {
    rc = check_time_embed(pr->score_model, net.channels, "score_model");
    if (rc != SDEH_OK) return rc;
    if (pr->score_model.dim_out != 1 && pr->score_model.dim_out != d)
      return fail(SDEH_ERR_UNSUPPORTED, "score_model.dim_out=%d (1 or dim supported)", pr->score_model.dim_out);
    g = pr->score_model.dim_out == 1 ? 1 : (plan->wide ? 32 * row_tiles(d) : plan->variant->dp);
  }
  const bool need_target_score = pr->ctrl_kind == SDEH_CTRL_SCORE || pr->ctrl_kind == SDEH_CTRL_LERP ||
                                 pr->ctrl_kind == SDEH_CTRL_LERP_TARGET;
  const bool need_target = need_target_score || (pr->flags & SDEH_FLAG_TERMINAL_TARGET);
  rc = check_density(pr->target, d, "target", !need_target);
  if (rc != SDEH_OK) return rc;
  if (pr->target.kind == SDEH_DENS_EXTERNAL && !plan->wide)
    return fail(SDEH_ERR_UNSUPPORTED, "a target whose score is supplied per step (SDEH_DENS_EXTERNAL) runs on the wide kernels (d > 64 or "
                                      "channels >= 128) through sdeh_simulate_fwd_steps");
  if (pr->prior.kind == SDEH_DENS_EXTERNAL || pr->second.kind == SDEH_DENS_EXTERNAL)
    return fail(SDEH_ERR_INVALID, "SDEH_DENS_EXTERNAL is a target kind");
  const bool refc = (pr->flags & SDEH_FLAG_REFERENCE_CTRL) && pr->loss_kind == SDEH_LOSS_REFERENCE_SDE;
  const bool need_prior = pr->ctrl_kind == SDEH_CTRL_LERP || pr->ctrl_kind == SDEH_CTRL_LERP_PRIOR || refc;
  if (need_prior && pr->prior.kind != SDEH_DENS_DIAG_GAUSS)
    return fail(SDEH_ERR_UNSUPPORTED, "prior score: only Gaussian priors are built in (kind %d)", pr->prior.kind);
  rc = check_density(pr->prior, d, "prior", true);
  if (rc != SDEH_OK) return rc;
  const bool need_second = pr->flags & (SDEH_FLAG_INIT_LOGP | SDEH_FLAG_TERMINAL_SECOND);
  if (need_second && pr->second.kind != SDEH_DENS_DIAG_GAUSS)
    return fail(SDEH_ERR_UNSUPPORTED, "initial/reference log-density: only Gaussians are built in (kind %d)", pr->second.kind);
  rc = check_density(pr->second, d, "second", true);
  if (rc != SDEH_OK) return rc;
  const int k = pr->target.kind == SDEH_DENS_GMM ? pr->target.n_components : 0;
  if (k > plan->desc.max_components)
    return fail(SDEH_ERR_CAPACITY, "simulate_fwd: GMM with %d components > plan max %d", k, plan->desc.max_components);

  if (plan->wide) {  // wide networks (sdeh_wide.hip, sdeh_wide_bwd.hip)
    if (integrate)
      return fail(SDEH_ERR_UNSUPPORTED, "networks with %d channels (or d > 64) have no plain integrator (channels = 64 with d <= 64 has)", net.channels);
    if ((pr->flags & SDEH_FLAG_INFERENCE_CTRL) && net.channels < 128)
      return fail(SDEH_ERR_UNSUPPORTED, "Bridge with d > 64 needs channels = 128 or 256 (the wide Bridge kernel splits >= 4 row tiles over its waves)");
    out->L = make_wide_layout(d, net.channels, net.n_hidden, n_steps, g, false, need_target && pr->target.kind == SDEH_DENS_GMM ? k : 0, backward);
    if ((size_t)out->L.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "simulate_fwd: workspace too small");
    out->v = nullptr; out->refc = refc; out->g = g; out->k = k;
    return SDEH_OK;
  }
  const Variant* v = plan->variant;
  const bool shared = pr->target.kind == SDEH_DENS_GMM && (pr->target.flags & SDEH_DENS_FLAG_SHARED_SCALE);
  const int nvary = shared ? SDEH_DENS_FLAG_GET_NVARY(pr->target.flags) : -1;
  const bool force_legacy = plan_opt(OPT_LEGACY) != nullptr;  // A/B aid: single-wave kernel, global tables
  // testing / measurement aid: only the variants with run-time switches for loss / control / target / activation (a plan option);
  // SDEH_GENERIC_ONLY=2 also rules out the ones with reduced mixture tables ("g4")
  const char* gen_only = plan_opt(OPT_GENERIC_ONLY);
  const bool no_spec = gen_only != nullptr;
  // which GMM table form would the layout give?  (0: tables do not fit LDS, 1: general, 2: shared scale)
  // the integrator runs on the single-wave code path of the generic variant (mixture tables in LDS when they fit)
  WsLayout L = make_layout(v->dp, net.channels, net.n_hidden, n_steps, k, g, shared, force_legacy, 0, backward);
  const Variant* sv = (backward || integrate) ? nullptr
                              : pick_specialised(v, pr->loss_kind, pr->ctrl_kind, pr->target.kind, L.gmm_lds,
                                                 net.activation, refc ? 1 : 0, nvary, no_spec);
  if (sv == nullptr && !(backward || integrate) && shared && nvary >= 0 && L.gmm_lds == 0 && !force_legacy) {
    // full tables beyond LDS (d near 64, many components): tables over the varying prefix may still fit
    const Variant* rv = pick_specialised(v, pr->loss_kind, pr->ctrl_kind, pr->target.kind, 2, net.activation, refc ? 1 : 0, nvary, no_spec);
    if (rv != nullptr && rv->gnv > 0 &&
        make_layout(v->dp, net.channels, net.n_hidden, n_steps, k, g, shared, force_legacy, rv->gnv).gmm_lds == 2)
      sv = rv;
  }
  if (gen_only != nullptr && gen_only[0] == '2') sv = nullptr;
  if (sv != nullptr && sv->dp == v->dp) {
    v = sv;
    if (v->gnv > 0) L = make_layout(v->dp, net.channels, net.n_hidden, n_steps, k, g, shared, force_legacy, v->gnv);
  }
  if ((size_t)L.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "simulate_fwd: workspace too small");
  out->L = L;
  out->v = v;
  out->refc = refc;
  out->g = g;
  out->k = k;
  return SDEH_OK;
}

// TimeReversalLoss with an inference control (Bridge): two networks, exact divergence by forward-mode tangents
static int simulate_bridge(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                           int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                           float* x_T, float* rnd, float* xs, void* stream, const Checked& ck, float* gp,
                           const float* div_noise) {
  const SdehInferenceCtrl& inf = pr->inference;
  const SdehFourierMLP& net = pr->base_model;
  const SdehFourierMLP& net2 = inf.base_model;
  const int d = net.dim;
  if (inf.ctrl_kind != SDEH_CTRL_CLIPPED && inf.ctrl_kind != SDEH_CTRL_LERP_PRIOR)
    return fail(SDEH_ERR_UNSUPPORTED, "inference control kind %d: ClippedCtrl and LerpPriorCtrl have a built-in divergence", inf.ctrl_kind);
  if (net2.dim != d || net2.channels != net.channels)
    return fail(SDEH_ERR_INVALID, "inference control: dim/channels (%d,%d) differ from the generative control's (%d,%d)", net2.dim,
                net2.channels, d, net.channels);
  if (net2.n_hidden < 0 || net2.n_hidden > plan->desc.max_hidden)
    return fail(SDEH_ERR_CAPACITY, "inference control: %d hidden layers > plan max %d", net2.n_hidden, plan->desc.max_hidden);
  if (net2.activation < SDEH_ACT_GELU_ERF || net2.activation > SDEH_ACT_RELU) return fail(SDEH_ERR_UNSUPPORTED, "inference control: activation %d", net2.activation);
  if (net2.input_w == nullptr || net2.input_b == nullptr || net2.out_w == nullptr || net2.out_b == nullptr)
    return fail(SDEH_ERR_INVALID, "inference control: null base_model parameter");
  for (int i = 0; i < net2.n_hidden; ++i)
    if (net2.hidden_w[i] == nullptr || net2.hidden_b[i] == nullptr) return fail(SDEH_ERR_INVALID, "inference control: null hidden layer %d", i);
  int rc = check_time_embed(net2.timestep_embed, net2.channels, "inference base_model.timestep_embed");
  if (rc != SDEH_OK) return rc;
  int g2 = 1;
  if (inf.ctrl_kind == SDEH_CTRL_LERP_PRIOR) {
    if (pr->prior.kind != SDEH_DENS_DIAG_GAUSS) return fail(SDEH_ERR_UNSUPPORTED, "LerpPriorCtrl inference control needs a Gaussian prior");
    if (pr->sde_kind == SDEH_SDE_NONE) return fail(SDEH_ERR_INVALID, "Lerp controls need an sde");
    if (inf.score_model.n_hidden > 0) {
      rc = check_time_embed(inf.score_model, net2.channels, "inference score_model");
      if (rc != SDEH_OK) return rc;
      if (inf.score_model.dim_out != 1 && inf.score_model.dim_out != d)
        return fail(SDEH_ERR_UNSUPPORTED, "inference score_model.dim_out=%d (1 or dim supported)", inf.score_model.dim_out);
      g2 = inf.score_model.dim_out == 1 ? 1 : plan->variant->dp;
    }
  }
  const Variant* v = plan->variant;
  const int dp = v->dp;
  // single-wave code path: mixture tables in global memory, both packed networks in LDS
  const WsLayout L1 = make_layout(dp, net.channels, net.n_hidden, n_steps, ck.k, ck.g, false, true);
  const WsLayout L2 = make_layout(dp, net.channels, net2.n_hidden, n_steps, 0, g2, false, true, 0, false, true);
  if ((size_t)L1.total + (size_t)L2.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "simulate_fwd (bridge): workspace too small");
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = L1; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "simulate_fwd (bridge): prep kernel launch failed");
  PrepArgs P2 = P;  // second region: the inference control's network, time embeddings and gamma table
  P2.ws = plan->ws + L1.total; P2.lay = L2;
  P2.prob.ctrl_kind = inf.ctrl_kind; P2.prob.clip_model = inf.clip_model; P2.prob.clip_score = inf.clip_score;
  P2.prob.scale_score = inf.scale_score; P2.prob.base_model = inf.base_model; P2.prob.score_model = inf.score_model;
  P2.prob.target.kind = P2.prob.prior.kind = P2.prob.second.kind = SDEH_DENS_NONE;  // the density tables live in region 1
  rc = launch_prep(P2, st);
  if (rc != SDEH_OK) return fail(rc, "simulate_fwd (bridge): second prep kernel launch failed");
  TrajArgs A{};
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L1; A.ws2 = plan->ws + L1.total; A.lay2 = L2;
  A.x0 = x0; A.noise = noise; A.xT = x_T; A.rnd = rnd; A.xs = xs;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = d;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = net.activation;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score; A.clip_target = pr->clip_target;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.prior = {pr->prior.kind, pr->prior.n_components, pr->prior.log_norm_const, pr->prior.p0, pr->prior.p1};
  A.second = {pr->second.kind, pr->second.n_components, pr->second.log_norm_const, pr->second.p0, pr->second.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  A.inf_kind = inf.ctrl_kind; A.inf_act = net2.activation;
  A.inf_clip_model = inf.clip_model; A.inf_clip_score = inf.clip_score; A.inf_scale_score = inf.scale_score;
  A.gp = gp; A.div_noise = div_noise;
  timing_begin(plan, st);
  rc = v->fn_bridge(A, st);
  timing_end(plan, st);
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), "bridge<%s>", v->name);
  if (rc == SDEH_ERR_UNSUPPORTED)
    return fail(rc, "simulate_fwd (bridge): two packed networks (+ mixture scratch) exceed 160 KiB of LDS at dim=%d", d);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "simulate_fwd (bridge): kernel launch failed");
}

static void fill_traj_args(TrajArgs& A, const SdehProblem* pr, const float* x0, int64_t batch, const float* noise, uint64_t seed,
                           uint64_t offset, int64_t row_offset, float* x_T, float* rnd, float* xs, int32_t n_steps) {
  memset(&A, 0, sizeof(A));
  A.x0 = x0; A.noise = noise; A.xT = x_T; A.rnd = rnd; A.xs = xs;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = pr->base_model.dim;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = pr->base_model.activation;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.clip_target = pr->clip_target; A.exp_sigma = pr->exp_sigma;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.prior = {pr->prior.kind, pr->prior.n_components, pr->prior.log_norm_const, pr->prior.p0, pr->prior.p1};
  A.second = {pr->second.kind, pr->second.n_components, pr->second.log_norm_const, pr->second.p0, pr->second.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
}

// Wide networks (C = 128 / 256, d <= 256): the channel-split kernels of sdeh_wide.hip, with or without an inference control.
// sdeh_simulate_fwd_steps: the steps [begin, end) of the grid; the tables of the whole grid are prepared by the segment that starts at 0
struct Segment {
  int begin, end;
  const float* ext_score;
  long long ext_stride;
};
static int simulate_wide(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0, int64_t batch,
                         const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset, float* x_T, float* rnd, float* xs,
                         void* stream, const Checked& ck, float* gp, const float* div_noise, float* sc_out, float* tsc_out,
                         const Segment* seg = nullptr) {
  if (pr->target.kind == SDEH_DENS_EXTERNAL && seg == nullptr)
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd (wide): a target whose score is supplied per step runs through sdeh_simulate_fwd_steps");
  const bool prep_now = seg == nullptr || seg->begin == 0;
  // (training: the wide forward keeps no network planes -- the caller's backward re-evaluates the network at the stored trajectory,
  // sdeh_wide_bwd.hip; what it keeps for a MIXTURE target is the score entering the control, sc_out [T, B, d], and the terminal target
  // score, tsc_out [B, d] (sdeh_simulate_fwd_train2 on a wide plan; Bridges: sdeh_simulate_fwd_aux2); Hutchinson probes are a
  // 64-channel feature)
  if (div_noise != nullptr)
    return fail(SDEH_ERR_UNSUPPORTED, "wide-network Bridge: the Hutchinson divergence estimators are built for channels = 64 "
                                      "(the exact divergence is built in)");
  if (gp != nullptr && !(pr->flags & SDEH_FLAG_INFERENCE_CTRL)) return fail(SDEH_ERR_INVALID, "simulate_fwd_aux: gp without an inference control");
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  const bool bridge = pr->flags & SDEH_FLAG_INFERENCE_CTRL;
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = ck.L; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  int rc = prep_now ? launch_prep(P, st) : SDEH_OK;
  if (rc != SDEH_OK) return fail(rc, "simulate_fwd (wide): prep kernel launch failed");
  TrajArgs A{};
  fill_traj_args(A, pr, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs, n_steps);
  A.ws = plan->ws; A.lay = ck.L;
  A.sc_out = sc_out; A.tsc_out = tsc_out;  // (only the mixture instantiations write them)
  if (seg != nullptr) {
    A.n_steps = seg->end - seg->begin; A.step0 = seg->begin; A.seg_continue = seg->begin > 0 ? 1 : 0;
    A.ext_score = seg->ext_score; A.ext_stride = seg->ext_stride;
    if (seg->begin > 0) A.flags &= ~SDEH_FLAG_INIT_LOGP;                                                  // rnd carries on
    if (seg->end < n_steps) A.flags &= ~(SDEH_FLAG_TERMINAL_TARGET | SDEH_FLAG_TERMINAL_SECOND);         // terminal terms: last segment
    if (pr->target.kind == SDEH_DENS_EXTERNAL) A.flags &= ~SDEH_FLAG_TERMINAL_TARGET;                     // ... of a supplied target: the caller
  }
  if (bridge) {
    const SdehInferenceCtrl& inf = pr->inference;
    const SdehFourierMLP& net2 = inf.base_model;
    if (inf.ctrl_kind != SDEH_CTRL_CLIPPED && inf.ctrl_kind != SDEH_CTRL_LERP_PRIOR)
      return fail(SDEH_ERR_UNSUPPORTED, "inference control kind %d: ClippedCtrl and LerpPriorCtrl have a built-in divergence", inf.ctrl_kind);
    if (net2.dim != d || net2.channels != net.channels)
      return fail(SDEH_ERR_INVALID, "inference control: dim/channels (%d,%d) differ from the generative control's (%d,%d)", net2.dim,
                  net2.channels, d, net.channels);
    if (net2.n_hidden < 0 || net2.n_hidden > 2)
      return fail(SDEH_ERR_UNSUPPORTED, "wide Bridge: the divergence is built for inference networks with at most 2 hidden layers "
                                        "(num_layers <= 4; got %d hidden)", net2.n_hidden);
    if (net2.n_hidden > plan->desc.max_hidden) return fail(SDEH_ERR_CAPACITY, "inference control: %d hidden layers > plan max %d", net2.n_hidden, plan->desc.max_hidden);
    if (net2.activation < SDEH_ACT_GELU_ERF || net2.activation > SDEH_ACT_RELU) return fail(SDEH_ERR_UNSUPPORTED, "inference control: activation %d", net2.activation);
    if (net2.input_w == nullptr || net2.input_b == nullptr || net2.out_w == nullptr || net2.out_b == nullptr)
      return fail(SDEH_ERR_INVALID, "inference control: null base_model parameter");
    for (int i = 0; i < net2.n_hidden; ++i)
      if (net2.hidden_w[i] == nullptr || net2.hidden_b[i] == nullptr) return fail(SDEH_ERR_INVALID, "inference control: null hidden layer %d", i);
    rc = check_time_embed(net2.timestep_embed, net2.channels, "inference base_model.timestep_embed");
    if (rc != SDEH_OK) return rc;
    int g2 = 1;
    if (inf.ctrl_kind == SDEH_CTRL_LERP_PRIOR) {
      if (pr->prior.kind != SDEH_DENS_DIAG_GAUSS) return fail(SDEH_ERR_UNSUPPORTED, "LerpPriorCtrl inference control needs a Gaussian prior");
      if (pr->sde_kind == SDEH_SDE_NONE) return fail(SDEH_ERR_INVALID, "Lerp controls need an sde");
      if (inf.score_model.n_hidden > 0) {
        rc = check_time_embed(inf.score_model, net2.channels, "inference score_model");
        if (rc != SDEH_OK) return rc;
        if (inf.score_model.dim_out != 1 && inf.score_model.dim_out != d)
          return fail(SDEH_ERR_UNSUPPORTED, "inference score_model.dim_out=%d (1 or dim supported)", inf.score_model.dim_out);
        g2 = inf.score_model.dim_out == 1 ? 1 : 32 * row_tiles(d);
      }
    }
    const WsLayout L2 = make_wide_layout(d, net2.channels, net2.n_hidden, n_steps, g2, true);
    if ((size_t)ck.L.total + (size_t)L2.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "simulate_fwd (wide bridge): workspace too small");
    PrepArgs P2 = P;  // second region: the inference control's network, time embeddings and gamma table
    P2.ws = plan->ws + ck.L.total; P2.lay = L2;
    P2.prob.ctrl_kind = inf.ctrl_kind; P2.prob.clip_model = inf.clip_model; P2.prob.clip_score = inf.clip_score;
    P2.prob.scale_score = inf.scale_score; P2.prob.base_model = inf.base_model; P2.prob.score_model = inf.score_model;
    P2.prob.target.kind = P2.prob.prior.kind = P2.prob.second.kind = SDEH_DENS_NONE;  // the density tables live in region 1
    rc = prep_now ? launch_prep(P2, st) : SDEH_OK;
    if (rc != SDEH_OK) return fail(rc, "simulate_fwd (wide bridge): second prep kernel launch failed");
    A.ws2 = plan->ws + ck.L.total; A.lay2 = L2;
    A.inf_kind = inf.ctrl_kind; A.inf_act = net2.activation;
    A.inf_clip_model = inf.clip_model; A.inf_clip_score = inf.clip_score; A.inf_scale_score = inf.scale_score;
    A.gp = gp;  // [T, B, d] or null: u + v per step (training: the inference network's upstream gradient)
  }
  if (bridge) {
    // 32 group sums per trajectory of the network part of the divergence: reserved by sdeh_plan_create / sdeh_plan_reserve
    const size_t need = (size_t)bridge_wide_scratch_floats(batch);
    if (need > plan->scratch_floats)
      return fail(SDEH_ERR_CAPACITY, "simulate_fwd (wide bridge): the plan's divergence scratch (%zu floats) is smaller than this batch needs "
                                     "(%zu): SdehPlanDesc.max_batch / sdeh_plan_reserve (no stream-ordered call allocates)",
                  plan->scratch_floats, need);
  }
  timing_begin(plan, st);
  int detail = 0;
  rc = bridge ? launch_bridge_wide(A, st, &detail, plan->scratch) : launch_wide(A, st, &detail);
  timing_end(plan, st);
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), bridge ? "bridge_wide<C=%d,split=%d>" : "traj_wide<C=%d,CT=%d>", net.channels, detail);
  if (rc == SDEH_ERR_UNSUPPORTED)
    return fail(rc, "simulate_fwd (wide): the problem needs more than 160 KiB of LDS (d=%d, C=%d, K=%d mixture components)", d, net.channels, ck.k);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "simulate_fwd (wide): kernel launch failed");
}

int32_t sdeh_simulate_fwd(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                          int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                          float* x_T, float* rnd, float* xs, void* stream) {
  return sdeh_simulate_fwd_aux(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs, nullptr, nullptr,
                               stream);
}

static int simulate_impl(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                         int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                         float* x_T, float* rnd, float* xs, float* gp, const float* div_noise, float* zt_out, float* nn_out,
                         bool* planes_written, void* stream, float* sc_out = nullptr, float* tsc_out = nullptr, float* xs_cm = nullptr,
                         float* u_out = nullptr, float* zrec = nullptr) {
  OptScope opt_scope(plan);
  if (planes_written != nullptr) *planes_written = false;
  if (x0 == nullptr || x_T == nullptr || rnd == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd: null argument");
  if ((gp != nullptr || div_noise != nullptr) && (pr == nullptr || !(pr->flags & SDEH_FLAG_INFERENCE_CTRL)))
    return fail(SDEH_ERR_INVALID, "simulate_fwd_aux: gp / div_noise only exist for problems with an inference control");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, row_offset, false, &ck);
  if (rc != SDEH_OK) return rc;
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  if (plan->wide) {
    // sdeh_simulate_fwd_train2 on a wide plan: `xs_cm` is the ROW-major trajectory [T+1, B, d] (the wide backward reads rows), sc /
    // tscore row-major as well and written for mixture targets only; sdeh_simulate_fwd_train's network planes are not produced
    const bool gmm = pr->target.kind == SDEH_DENS_GMM;
    rc = simulate_wide(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs != nullptr ? xs : xs_cm, stream, ck,
                       gp, div_noise, gmm ? sc_out : nullptr, gmm ? tsc_out : nullptr);
    if (rc == SDEH_OK && planes_written != nullptr) *planes_written = xs_cm != nullptr;
    return rc;
  }
  if (pr->flags & SDEH_FLAG_INFERENCE_CTRL)
    return simulate_bridge(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs, stream, ck, gp, div_noise);
  WsLayout L = ck.L;
  const Variant* v = ck.v;
  const bool force_legacy = plan_opt(OPT_LEGACY) != nullptr;
  // The mixture's contractions on the matrix pipe (sdeh_traj_ws.hpp: gmm_mm): mixtures of 21 .. 40 components (fewer rows are padded: the
  // instruction stream is fixed; below half of it the vector pipe is as fast) with tables
  // over all coordinates, where the caller vouches for the product form of the logits (SDEH_DENS_FLAG_MM_OK), on whole-wave launches
  // (the pair / quad modes of small batches read the tables from LDS) of the evaluation kernel.  Plan option SDEH_GMM_MM: "0" never,
  // "1" also without the caller's flag (measurements).
  {
    const char* mo = plan_opt(OPT_GMM_MM);
    const bool want = mo != nullptr ? mo[0] == '1' : (pr->target.flags & SDEH_DENS_FLAG_MM_OK) != 0;
    const bool compiled = (v->gmm == 2 || v->gmm < 0) && v->gnv <= 0 && v->dp > 8 && !v->pad;  // = gmm_mm_compiled<...>()
    const bool training = zt_out != nullptr || nn_out != nullptr || sc_out != nullptr || tsc_out != nullptr || xs_cm != nullptr ||
                          u_out != nullptr || zrec != nullptr;
    const bool general = L.gmm_lds == 1 && v->gmm < 0;  // per-component scales: the run-time switched variants carry that form
    if (want && compiled && !force_legacy && !training && pr->target.kind == SDEH_DENS_GMM && (L.gmm_lds == 2 || general) &&
        ck.k > SDEH_MM_ROWS / 2 && ck.k <= SDEH_MM_ROWS && batch > 32 * 256 && plan_opt(OPT_WS_GROUPS) == nullptr && plan_opt(OPT_WS_QUAD) == nullptr) {
      const WsLayout M = make_layout(v->dp, net.channels, net.n_hidden, n_steps, ck.k, ck.g, !general, false, 0, false, false, true);
      if (M.gmm_lds == (general ? 4 : 3) && (size_t)M.total <= plan->ws_floats) L = M;
    }
    // the out layer on 4 x 4 x 1 matrix instructions (d = 5 .. 16, no empty rows): every evaluation launch carries the operand image (5 KB
    // at most); the whole-wave modes use it, the pair / quad modes of small batches have out layers of their own.  Plan option
    // SDEH_WS_OUT4 = "0" keeps the 32-row tiles
    const char* oo = plan_opt(OPT_WS_OUT4);
    if (!(oo != nullptr && oo[0] == '0') && !force_legacy && !training) {
      const bool shared_l = L.gmm_lds == 2 || L.gmm_lds == 3;
      const WsLayout M = make_layout(v->dp, net.channels, net.n_hidden, n_steps, ck.k, ck.g, shared_l, false, v->gnv > 0 ? v->gnv : 0, false,
                                     false, L.gmm_lds >= 3, true);
      if (M.w_out4 >= 0 && M.gmm_lds == L.gmm_lds && (size_t)M.total <= plan->ws_floats) L = M;
    }
  }

  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws;
  P.lay = L;
  P.prob = *pr;
  P.ts = ts;
  P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "simulate_fwd: prep kernel launch failed: %s", hipGetErrorString(hipGetLastError()));

  TrajArgs A{};
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L;
  A.x0 = x0; A.noise = noise; A.xT = x_T; A.rnd = rnd; A.xs = xs;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = d;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = net.activation;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.clip_target = pr->clip_target; A.exp_sigma = pr->exp_sigma;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.prior = {pr->prior.kind, pr->prior.n_components, pr->prior.log_norm_const, pr->prior.p0, pr->prior.p1};
  A.second = {pr->second.kind, pr->second.n_components, pr->second.log_norm_const, pr->second.p0, pr->second.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  timing_begin(plan, st);
  // The wave-specialised kernel needs GMM tables in LDS; mixtures too large for that use the single-wave kernel
  // with scalar-load tables (also selectable with SDEH_LEGACY=1 for A/B measurements).
  const bool legacy = force_legacy || (pr->target.kind == SDEH_DENS_GMM && L.gmm_lds == 0);
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), "%s<%s%s>", legacy ? "traj_legacy" : "traj_ws", v->name, L.gmm_lds >= 3 ? ",mm" : "");
  if (legacy) {
    rc = v->fn_legacy(A, st);
    if (rc == SDEH_ERR_UNSUPPORTED && v != plan->variant) rc = plan->variant->fn_legacy(A, st);
  } else {
    A.zt_out = zt_out; A.nn_out = nn_out;  // only the wave-specialised kernel writes the training planes
    A.sc_out = sc_out; A.tsc_out = tsc_out; A.xs_cm = xs_cm; A.u_out = u_out;
    A.zrec = zrec;
    rc = v->fn(A, st);
    if (rc == SDEH_OK && planes_written != nullptr)
      *planes_written = (zt_out != nullptr && nn_out != nullptr) || xs_cm != nullptr;
    // image + exchange buffers beyond 160 KiB (deep networks): the single-wave kernel needs less LDS
    if (rc == SDEH_ERR_UNSUPPORTED && pr->target.kind != SDEH_DENS_GMM) {
      A.zt_out = nullptr; A.nn_out = nullptr; A.sc_out = nullptr; A.tsc_out = nullptr; A.xs_cm = nullptr; A.u_out = nullptr;
      A.zrec = nullptr;
      if (planes_written != nullptr) *planes_written = false;
      rc = plan->variant->fn_legacy(A, st);
      snprintf(plan->last_kernel, sizeof(plan->last_kernel), "traj_legacy<%s>", plan->variant->name);
    }
  }
  timing_end(plan, st);
  if (rc != SDEH_OK) return fail(rc, "simulate_fwd: trajectory kernel launch failed (dp=%d)", v->dp);
  return SDEH_OK;
}

int32_t sdeh_simulate_fwd_aux(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                              int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                              float* x_T, float* rnd, float* xs, float* gp, const float* div_noise, void* stream) {
  return simulate_impl(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs, gp, div_noise, nullptr,
                       nullptr, nullptr, stream);
}

int32_t sdeh_simulate_fwd_aux2(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                               int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                               float* x_T, float* rnd, float* xs, float* gp, const float* div_noise, float* sc, float* tscore,
                               void* stream) {
  if ((sc != nullptr || tscore != nullptr) && (plan == nullptr || !plan->wide))
    return fail(SDEH_ERR_INVALID, "simulate_fwd_aux2: sc / tscore are written by the wide-network kernels only (64-channel Bridges: the "
                                  "backward kernels evaluate the mixture themselves)");
  return simulate_impl(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs, gp, div_noise, nullptr,
                       nullptr, nullptr, stream, sc, tscore);
}

int32_t sdeh_simulate_fwd_steps(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, int32_t step_begin,
                                int32_t step_end, const float* x_in, int64_t batch, const float* noise, uint64_t seed, uint64_t offset,
                                int64_t row_offset, float* x_out, float* rnd, float* xs, float* gp, const float* ext_score,
                                int64_t ext_stride, void* stream) {
  OptScope opt_scope(plan);
  if (x_in == nullptr || x_out == nullptr || rnd == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd_steps: null argument");
  if (x_in == x_out) return fail(SDEH_ERR_INVALID, "simulate_fwd_steps: x_out must not be x_in (workgroups that share a tile read x_in)");
  if (gp != nullptr && (pr == nullptr || !(pr->flags & SDEH_FLAG_INFERENCE_CTRL)))
    return fail(SDEH_ERR_INVALID, "simulate_fwd_steps: gp only exists for problems with an inference control");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, row_offset, false, &ck);
  if (rc != SDEH_OK) return rc;
  if (!plan->wide)
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_steps: segments of the grid are a feature of the wide kernels (d > 64 or channels >= 128)");
  if (step_begin < 0 || step_end <= step_begin || step_end > n_steps)
    return fail(SDEH_ERR_INVALID, "simulate_fwd_steps: steps [%d, %d) of a grid of %d", step_begin, step_end, n_steps);
  const bool need_t = pr->ctrl_kind == SDEH_CTRL_SCORE || pr->ctrl_kind == SDEH_CTRL_LERP || pr->ctrl_kind == SDEH_CTRL_LERP_TARGET;
  if (pr->target.kind == SDEH_DENS_EXTERNAL && need_t) {
    if (ext_score == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd_steps: SDEH_DENS_EXTERNAL without ext_score");
    if (ext_stride == 0 ? step_end - step_begin != 1 : ext_stride < batch * pr->base_model.dim)
      return fail(SDEH_ERR_INVALID, "simulate_fwd_steps: ext_stride=%lld (0 for a one-step segment, else >= batch * dim)", (long long)ext_stride);
  }
  const Segment seg{step_begin, step_end, pr->target.kind == SDEH_DENS_EXTERNAL ? ext_score : nullptr, ext_stride};
  return simulate_wide(plan, pr, ts, n_steps, x_in, batch, noise, seed, offset, row_offset, x_out, rnd, xs, stream, ck, gp, nullptr, nullptr,
                       nullptr, &seg);
}

int32_t sdeh_simulate_fwd_train(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                                int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                float* x_T, float* rnd, float* xs, float* zt, float* nn, void* stream) {
  if (xs == nullptr || zt == nullptr || nn == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd_train: null argument");
  if (pr != nullptr && (pr->flags & SDEH_FLAG_INFERENCE_CTRL))
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_train: the Bridge forward keeps no planes (use sdeh_simulate_fwd_aux)");
  bool written = false;
  const int rc = simulate_impl(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, xs, nullptr, nullptr,
                               zt, nn, &written, stream);
  if (rc != SDEH_OK) return rc;
  return written ? SDEH_OK : 1;  // 1: integrated by a kernel that keeps no planes -- the backward re-evaluates the network
}

int32_t sdeh_simulate_fwd_train2(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                                 int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                 float* x_T, float* rnd, float* xs, float* sc, float* tscore, void* stream) {
  if (xs == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd_train2: xs (the coordinate-major trajectory plane) is required");
  if (pr != nullptr && (pr->flags & SDEH_FLAG_INFERENCE_CTRL))
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_train2: the Bridge forward keeps no planes (use sdeh_simulate_fwd_aux)");
  if (pr != nullptr && pr->ctrl_kind != SDEH_CTRL_CLIPPED && sc == nullptr && !(plan != nullptr && plan->wide))
    return fail(SDEH_ERR_INVALID, "simulate_fwd_train2: sc is required for controls with a score term");
  bool written = false;
  const int rc = simulate_impl(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, nullptr, nullptr, nullptr,
                               nullptr, nullptr, &written, stream, sc, tscore, xs);
  if (rc != SDEH_OK) return rc;
  return written ? SDEH_OK : 1;  // 1: integrated by a kernel that writes none of the planes
}

int32_t sdeh_simulate_fwd_train2u(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                                  int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                  float* x_T, float* rnd, float* xs, float* sc, float* tscore, float* u, void* stream) {
  if (xs == nullptr || u == nullptr) return fail(SDEH_ERR_INVALID, "simulate_fwd_train2u: xs and u (coordinate-major planes) are required");
  if (pr != nullptr && (pr->flags & SDEH_FLAG_INFERENCE_CTRL))
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_train2u: the problem WITHOUT its inference control (sdeh_bridge_inference_fwd adds its terms)");
  if (plan != nullptr && plan->wide) return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_train2u: 64-channel plans");
  if (pr != nullptr && pr->ctrl_kind != SDEH_CTRL_CLIPPED && sc == nullptr)
    return fail(SDEH_ERR_INVALID, "simulate_fwd_train2u: sc is required for controls with a score term");
  bool written = false;
  const int rc = simulate_impl(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, nullptr, nullptr, nullptr,
                               nullptr, nullptr, &written, stream, sc, tscore, xs, u);
  if (rc != SDEH_OK) return rc;
  return written ? SDEH_OK : 1;
}

int64_t sdeh_zrec_floats(int32_t dim, int32_t n_hidden, int32_t n_steps, int64_t batch) {
  if (dim < 1 || n_hidden < 0 || n_steps < 1 || batch < 1) return 0;
  return (int64_t)n_steps * ((batch + 31) / 32) * zrec_tile_floats(n_hidden, dim);
}

int32_t sdeh_simulate_fwd_train3(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* x0,
                                 int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                 float* x_T, float* rnd, float* xs, float* sc, float* tscore, float* u, float* zrec, void* stream) {
  if (xs == nullptr || zrec == nullptr)
    return fail(SDEH_ERR_INVALID, "simulate_fwd_train3: xs and zrec are required (sdeh_simulate_fwd_train2[u] keeps the planes without the record)");
  if (pr != nullptr && (pr->flags & SDEH_FLAG_INFERENCE_CTRL))
    return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_train3: the problem WITHOUT its inference control (sdeh_bridge_inference_fwd adds its terms)");
  if (plan != nullptr && plan->wide) return fail(SDEH_ERR_UNSUPPORTED, "simulate_fwd_train3: 64-channel plans");
  if (pr != nullptr && pr->ctrl_kind != SDEH_CTRL_CLIPPED && sc == nullptr)
    return fail(SDEH_ERR_INVALID, "simulate_fwd_train3: sc is required for controls with a score term");
  bool written = false;
  const int rc = simulate_impl(plan, pr, ts, n_steps, x0, batch, noise, seed, offset, row_offset, x_T, rnd, nullptr, nullptr, nullptr,
                               nullptr, nullptr, &written, stream, sc, tscore, xs, u, zrec);
  if (rc != SDEH_OK) return rc;
  return written ? SDEH_OK : 1;  // 1: integrated by a kernel that writes none of the planes
}

int32_t sdeh_ctrl_backward(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                           int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                           const float* grad_rnd, float* zt, float* dt, float* dout, float* dgam, void* stream) {
  return sdeh_ctrl_backward_ex(plan, pr, ts, n_steps, xs, batch, noise, seed, offset, row_offset, grad_rnd, nullptr, nullptr,
                               nullptr, nullptr, zt, dt, dout, dgam, nullptr, nullptr, nullptr, nullptr, stream);
}

int32_t sdeh_bridge_div_backward(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                 int64_t batch, const float* grad_rnd, const float* zt, float* tz, float* ta, float* td,
                                 float* d2, float* cj, float* dgam, float* dx_accum, const float* div_noise, void* stream) {
  OptScope opt_scope(plan);
  if (plan == nullptr || pr == nullptr || ts == nullptr || xs == nullptr || grad_rnd == nullptr || zt == nullptr || tz == nullptr ||
      ta == nullptr || td == nullptr || d2 == nullptr || cj == nullptr)
    return fail(SDEH_ERR_INVALID, "bridge_div_backward: null argument");
  if (!(pr->flags & SDEH_FLAG_INFERENCE_CTRL)) return fail(SDEH_ERR_INVALID, "bridge_div_backward: the problem has no inference control");
  if (plan->wide) return fail(SDEH_ERR_UNSUPPORTED, "bridge_div_backward: wide-network plans take sdeh_bridge_div_backward_wide");
  if (batch < 1 || n_steps < 1 || n_steps > plan->desc.max_steps) return fail(SDEH_ERR_INVALID, "bridge_div_backward: batch=%lld n_steps=%d", (long long)batch, n_steps);
  const SdehInferenceCtrl& inf = pr->inference;
  const SdehFourierMLP& net2 = inf.base_model;
  const int d = net2.dim;
  if (d != plan->desc.dim || net2.channels != plan->desc.channels || net2.n_hidden > plan->desc.max_hidden)
    return fail(SDEH_ERR_CAPACITY, "bridge_div_backward: network geometry differs from the plan's");
  if (inf.ctrl_kind != SDEH_CTRL_CLIPPED && inf.ctrl_kind != SDEH_CTRL_LERP_PRIOR)
    return fail(SDEH_ERR_UNSUPPORTED, "bridge_div_backward: inference control kind %d", inf.ctrl_kind);
  if (inf.ctrl_kind == SDEH_CTRL_LERP_PRIOR && (pr->prior.kind != SDEH_DENS_DIAG_GAUSS || dgam == nullptr))
    return fail(SDEH_ERR_INVALID, "bridge_div_backward: LerpPriorCtrl needs the Gaussian prior and dgam");
  int g2 = 1;
  if (inf.ctrl_kind == SDEH_CTRL_LERP_PRIOR && inf.score_model.n_hidden > 0) g2 = inf.score_model.dim_out == 1 ? 1 : plan->variant->dp;
  const Variant* v = plan->variant;
  // region 1: only the per-step coefficients and the prior table are read (no network: ctrl NONE); region 2: the inference network
  const WsLayout L1 = make_layout(v->dp, net2.channels, 0, n_steps, 0, 1, false, true);
  const WsLayout L2 = make_layout(v->dp, net2.channels, net2.n_hidden, n_steps, 0, g2, false, true, 0, true, true);
  if ((size_t)L1.total + (size_t)L2.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "bridge_div_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = L1; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  P.prob.ctrl_kind = SDEH_CTRL_NONE; P.prob.target.kind = SDEH_DENS_NONE; P.prob.second.kind = SDEH_DENS_NONE;
  P.prob.flags &= ~SDEH_FLAG_INFERENCE_SDE;
  int rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "bridge_div_backward: prep kernel launch failed");
  PrepArgs P2 = P;
  P2.prob = *pr;
  P2.ws = plan->ws + L1.total; P2.lay = L2;
  P2.prob.ctrl_kind = inf.ctrl_kind; P2.prob.clip_model = inf.clip_model; P2.prob.clip_score = inf.clip_score;
  P2.prob.scale_score = inf.scale_score; P2.prob.base_model = inf.base_model; P2.prob.score_model = inf.score_model;
  P2.prob.target.kind = P2.prob.prior.kind = P2.prob.second.kind = SDEH_DENS_NONE;
  rc = launch_prep(P2, st);
  if (rc != SDEH_OK) return fail(rc, "bridge_div_backward: second prep kernel launch failed");
  BridgeBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L1; A.ws2 = plan->ws + L1.total; A.lay2 = L2;
  A.xs = xs; A.grad_rnd = grad_rnd; A.zt = zt; A.tz = tz; A.ta = ta; A.td = td; A.d2 = d2; A.cj = cj; A.dgam = dgam; A.dx = dx_accum; A.eps = div_noise;
  A.batch = batch; A.n_steps = n_steps; A.d = d; A.inf_kind = inf.ctrl_kind; A.act = net2.activation;
  A.clip_model = inf.clip_model; A.clip_score = inf.clip_score; A.scale_score = inf.scale_score;
  rc = v->fn_bridge_bwd(A, st);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "bridge_div_backward: kernel launch failed");
}

int32_t sdeh_ctrl_backward_ex(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                              int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                              const float* grad_rnd, const float* gextra, const float* cost_ctrl, const float* lam_extra,
                              float* dx_out, float* zt, float* dt, float* dout, float* dgam, const float* nn_in, float* xt_out,
                              const float* sc_in, const float* tscore_in, void* stream) {
  OptScope opt_scope(plan);
  if (xs == nullptr || grad_rnd == nullptr || zt == nullptr || dt == nullptr || dout == nullptr)
    return fail(SDEH_ERR_INVALID, "ctrl_backward: null argument");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, row_offset, true, &ck);
  if (rc != SDEH_OK) return rc;
  const bool bptt = !(pr->flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  if (bptt && (gextra != nullptr || dx_out != nullptr))
    return fail(SDEH_ERR_INVALID, "ctrl_backward_ex: gextra / dx_out belong to the row-parallel mode (SDEH_FLAG_CHANGE_SDE_CTRL)");
  if (!bptt && (cost_ctrl != nullptr || lam_extra != nullptr))
    return fail(SDEH_ERR_INVALID, "ctrl_backward_ex: cost_ctrl / lam_extra belong to back-propagation through time");
  if (bptt && (pr->flags & SDEH_FLAG_INIT_LOGP))
    return fail(SDEH_ERR_UNSUPPORTED, "ctrl_backward: an initial log-density term with an attached control is not a "
                                      "configuration the reference produces");
  if (pr->ctrl_kind != SDEH_CTRL_CLIPPED && dgam == nullptr) return fail(SDEH_ERR_INVALID, "ctrl_backward: dgam is null");
  const WsLayout& L = ck.L;
  if ((xt_out != nullptr || sc_in != nullptr || tscore_in != nullptr) && !plan->wide)
    return fail(SDEH_ERR_INVALID, "ctrl_backward_ex: xt_out / sc_in / tscore_in belong to the wide-network kernels");
  if (plan->wide) {  // channel-split chain kernel of sdeh_wide_bwd.hip (same planes; nn_in is not used: the wide forward keeps none)
    const bool planes_stand_in = pr->target.kind == SDEH_DENS_GMM || pr->target.kind == SDEH_DENS_EXTERNAL;
    // (a supplied score is a CONSTANT of the adjoint recursion, like a mixture's: the reference obtains such scores by autograd without a
    // graph -- distr/base.py:130-137 under reparam.py:56-66, 185-197 -- so back-propagation through time needs no derivative of it; the
    // terminal cost's derivative arrives as tscore_in)
    if (planes_stand_in) {  // the chain kernel evaluates no mixture / no supplied target: the forward launch's planes stand in
      const bool ctrl_t = pr->ctrl_kind == SDEH_CTRL_SCORE || pr->ctrl_kind == SDEH_CTRL_LERP || pr->ctrl_kind == SDEH_CTRL_LERP_TARGET;
      if ((ctrl_t && sc_in == nullptr) || (bptt && (pr->flags & SDEH_FLAG_TERMINAL_TARGET) && tscore_in == nullptr))
        return fail(SDEH_ERR_UNSUPPORTED, "ctrl_backward (wide): a mixture target needs the planes of sdeh_simulate_fwd_train2 (sc_in: the score "
                                          "entering the control; tscore_in for methods kl / kl_ito) -- the wide backward evaluates no mixture");
    }
    if ((long long)n_steps * batch >= (1LL << 25))
      return fail(SDEH_ERR_CAPACITY, "ctrl_backward (wide): n_steps * batch = %lld rows: the plane columns are addressed with 32-bit byte "
                                     "offsets per row tile (< 2^25 rows per call; split the batch)", (long long)n_steps * batch);
    hipStream_t stw = (hipStream_t)stream;
    PrepArgs Pw;
    Pw.ws = plan->ws; Pw.lay = L; Pw.prob = *pr; Pw.ts = ts; Pw.n_steps = n_steps;
    Pw.ts_out = nullptr; Pw.n_out = 0; Pw.eps = 0.0f;
    rc = launch_prep(Pw, stw);
    if (rc != SDEH_OK) return fail(rc, "ctrl_backward (wide): prep kernel launch failed");
    BwdArgs Aw;
    memset(&Aw, 0, sizeof(Aw));
    Aw.ws = plan->ws; Aw.lay = L; Aw.xs = xs; Aw.noise = noise; Aw.grad_rnd = grad_rnd; Aw.gextra = gextra;
    Aw.cost_ctrl = cost_ctrl; Aw.lam_extra = lam_extra; Aw.dx = dx_out;
    Aw.zt = zt; Aw.dt = dt; Aw.dout = dout; Aw.dgam = dgam; Aw.nn_in = nullptr; Aw.xt_out = xt_out;
    Aw.sc_in = planes_stand_in ? sc_in : nullptr; Aw.tscore_in = planes_stand_in ? tscore_in : nullptr;
    Aw.batch = batch; Aw.row_offset = row_offset; Aw.n_steps = n_steps; Aw.d = pr->base_model.dim;
    Aw.loss_kind = pr->loss_kind; Aw.ctrl_kind = pr->ctrl_kind; Aw.flags = pr->flags; Aw.act = pr->base_model.activation;
    Aw.clip_model = pr->clip_model; Aw.clip_score = pr->clip_score; Aw.scale_score = pr->scale_score;
    Aw.clip_target = pr->clip_target;
    Aw.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
    Aw.seed = seed; Aw.offset = offset; Aw.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
    timing_begin(plan, stw);
    rc = launch_wide_bwd(Aw, stw);
    timing_end(plan, stw);
    snprintf(plan->last_kernel, sizeof(plan->last_kernel), "bwd_wide<C=%d,%s>", pr->base_model.channels, bptt ? "bptt" : "rows");
    if (rc == SDEH_ERR_UNSUPPORTED)
      return fail(rc, "ctrl_backward (wide): act' planes of %d layers at C=%d exceed 160 KiB of LDS", pr->base_model.n_hidden + 1, pr->base_model.channels);
    return rc == SDEH_OK ? SDEH_OK : fail(rc, "ctrl_backward (wide): kernel launch failed");
  }
  // The backward kernel reads mixture tables from LDS only.  It needs them for the control's target score and -- in
  // back-propagation through time -- for the terminal cost's d/dx_T; ClippedCtrl / LerpPriorCtrl in the row-parallel mode need neither.
  const bool ctrl_uses_target = pr->ctrl_kind != SDEH_CTRL_CLIPPED && pr->ctrl_kind != SDEH_CTRL_LERP_PRIOR;
  if (pr->target.kind == SDEH_DENS_GMM && L.gmm_lds == 0 && (ctrl_uses_target || (bptt && (pr->flags & SDEH_FLAG_TERMINAL_TARGET))))
    return fail(SDEH_ERR_UNSUPPORTED, "ctrl_backward: mixture tables do not fit in LDS next to the transposed weights");
  if (ck.v->fn_bwd == nullptr) return fail(SDEH_ERR_UNSUPPORTED, "ctrl_backward: no backward kernel for this variant");
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = L; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "ctrl_backward: prep kernel launch failed");
  BwdArgs A;
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L; A.xs = xs; A.noise = noise; A.grad_rnd = grad_rnd; A.gextra = gextra;
  A.cost_ctrl = cost_ctrl; A.lam_extra = lam_extra; A.dx = dx_out;
  A.zt = zt; A.dt = dt; A.dout = dout; A.dgam = dgam; A.nn_in = nn_in;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = pr->base_model.dim;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = pr->base_model.activation;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.clip_target = pr->clip_target;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  rc = ck.v->fn_bwd(A, st);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "ctrl_backward: kernel launch failed");
}

// ---------------------------------------------------------------------------------------------------------
// Bridge on wide networks: gradient of the divergence term, fused (sdeh_wide_bwd.hip)
// ---------------------------------------------------------------------------------------------------------
static void wide_div_sizes(int d, int c, int n_hidden, int n_steps, long long batch, long long* grid, long long* xp, long long* cp,
                           long long* sums, long long* out) {
  const long long g = wide_div_grid(batch, n_steps);
  *grid = g;
  *xp = g * c * c;
  *cp = g * d * c;
  *sums = ((g + 31) / 32) * (long long)c * (c > d ? c : d);
  *out = 2LL * d * c + (long long)n_hidden * c * c;
}

int32_t sdeh_bridge_div_backward_wide_sizes(int32_t dim, int32_t channels, int32_t n_hidden, int32_t n_steps, int64_t batch,
                                            int64_t* scratch_floats, int64_t* out_floats) {
  if (dim < 1 || (channels != 128 && channels != 256) || n_hidden < 1 || n_hidden > 2 || n_steps < 1 || batch < 1 ||
      scratch_floats == nullptr || out_floats == nullptr)
    return fail(SDEH_ERR_INVALID, "bridge_div_backward_wide_sizes: bad argument (channels 128 / 256, one or two hidden layers)");
  long long g, xp, cp, sums, out;
  wide_div_sizes(dim, channels, n_hidden, n_steps, batch, &g, &xp, &cp, &sums, &out);
  *scratch_floats = 2 * xp + 3 * cp + sums;
  *out_floats = out;
  return SDEH_OK;
}

int32_t sdeh_bridge_div_backward_wide(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                      int64_t batch, const float* grad_rnd, const float* zt, float* d2, float* dgam, float* dx_accum,
                                      float* scratch, int64_t scratch_floats, float* out, void* stream) {
  OptScope opt_scope(plan);
  if (plan == nullptr || pr == nullptr || ts == nullptr || xs == nullptr || grad_rnd == nullptr || zt == nullptr || d2 == nullptr ||
      scratch == nullptr || out == nullptr)
    return fail(SDEH_ERR_INVALID, "bridge_div_backward_wide: null argument");
  if (!plan->wide || !(pr->flags & SDEH_FLAG_INFERENCE_CTRL))
    return fail(SDEH_ERR_INVALID, "bridge_div_backward_wide: needs a wide plan and a problem with an inference control");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, 0, false, &ck);
  if (rc != SDEH_OK) return rc;
  const SdehInferenceCtrl& inf = pr->inference;
  const SdehFourierMLP& net2 = inf.base_model;
  const int d = pr->base_model.dim, C = net2.channels;
  if (C != pr->base_model.channels || (C != 128 && C != 256))
    return fail(SDEH_ERR_UNSUPPORTED, "bridge_div_backward_wide: both networks need 128 or 256 channels");
  if (net2.n_hidden < 1 || net2.n_hidden > 2)
    return fail(SDEH_ERR_UNSUPPORTED, "bridge_div_backward_wide: built for inference networks with one or two hidden layers (got %d)", net2.n_hidden);
  if (inf.ctrl_kind == SDEH_CTRL_LERP_PRIOR && dgam == nullptr) return fail(SDEH_ERR_INVALID, "bridge_div_backward_wide: dgam is null");
  int g2 = 1;
  if (inf.ctrl_kind == SDEH_CTRL_LERP_PRIOR && inf.score_model.n_hidden > 0) g2 = inf.score_model.dim_out == 1 ? 1 : 32 * row_tiles(d);
  const WsLayout L2 = make_wide_layout(d, C, net2.n_hidden, n_steps, g2, true, 0, true);  // + the transposed images (input_embed^T: d/dx)
  if ((size_t)ck.L.total + (size_t)L2.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "bridge_div_backward_wide: workspace too small");
  long long grid, xp, cp, sums, n_out;
  wide_div_sizes(d, C, net2.n_hidden, n_steps, batch, &grid, &xp, &cp, &sums, &n_out);
  if (scratch_floats < 2 * xp + 3 * cp + sums)
    return fail(SDEH_ERR_CAPACITY, "bridge_div_backward_wide: scratch too small (%lld < %lld floats)", (long long)scratch_floats, 2 * xp + 3 * cp + sums);
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = ck.L; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "bridge_div_backward_wide: prep kernel launch failed");
  PrepArgs P2 = P;
  P2.ws = plan->ws + ck.L.total; P2.lay = L2;
  P2.prob.ctrl_kind = inf.ctrl_kind; P2.prob.clip_model = inf.clip_model; P2.prob.clip_score = inf.clip_score;
  P2.prob.scale_score = inf.scale_score; P2.prob.base_model = inf.base_model; P2.prob.score_model = inf.score_model;
  P2.prob.target.kind = P2.prob.prior.kind = P2.prob.second.kind = SDEH_DENS_NONE;
  rc = launch_prep(P2, st);
  if (rc != SDEH_OK) return fail(rc, "bridge_div_backward_wide: second prep kernel launch failed");
  // scratch: [xpart side 1][xpart side 0][cpart side 1][cpart side 0][spart][sums]
  float* xp1 = scratch; float* xp0 = xp1 + xp; float* cp1 = xp0 + xp; float* cp0 = cp1 + cp; float* sp0 = cp0 + cp; float* sm = sp0 + cp;
  WideDivArgs A;
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = ck.L; A.ws2 = plan->ws + ck.L.total; A.lay2 = L2;
  A.xs = xs; A.grad_rnd = grad_rnd; A.zt = zt; A.d2 = d2; A.dgam = dgam; A.dx = dx_accum;
  A.batch = batch; A.n_steps = n_steps; A.d = d; A.inf_kind = inf.ctrl_kind; A.act = net2.activation;
  A.clip_model = inf.clip_model; A.clip_score = inf.clip_score; A.scale_score = inf.scale_score;
  float* g_in = out; float* g_out = out + (long long)d * C; float* g_hid = g_out + (long long)d * C;
  timing_begin(plan, st);
  if (net2.n_hidden == 2) {  // side 1: d W_2^T, d W_out, adj z_2
    A.side = 1; A.xpart = xp1; A.cpart = cp1; A.spart = nullptr;
    rc = launch_wide_div_bwd(A, st);
    if (rc == SDEH_OK) rc = launch_partial_sums(xp1, 1, grid, (long long)C * C, sm, g_hid + (long long)C * C, st);
    if (rc == SDEH_OK) rc = launch_partial_sums(cp1, 1, grid, (long long)d * C, sm, g_out, st);
  }
  if (rc == SDEH_OK) {  // side 0: d W_1, d W_in^T, (one hidden layer: d W_out), adj z_1, adj z_0, d gamma
    A.side = 0; A.xpart = xp0; A.cpart = cp0; A.spart = net2.n_hidden == 1 ? sp0 : nullptr;
    rc = launch_wide_div_bwd(A, st);
    if (rc == SDEH_OK) rc = launch_partial_sums(xp0, 1, grid, (long long)C * C, sm, g_hid, st);
    if (rc == SDEH_OK) rc = launch_partial_sums(cp0, 1, grid, (long long)d * C, sm, g_in, st);
    if (rc == SDEH_OK && net2.n_hidden == 1) rc = launch_partial_sums(sp0, 1, grid, (long long)d * C, sm, g_out, st);
  }
  timing_end(plan, st);
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), "bridge_div_bwd_wide<C=%d>", C);
  if (rc == SDEH_ERR_UNSUPPORTED) return fail(rc, "bridge_div_backward_wide: planes of %d layers at C=%d exceed 160 KiB of LDS", net2.n_hidden + 2, C);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "bridge_div_backward_wide: kernel launch failed");
}

// ---------------------------------------------------------------------------------------------------------
// Fused training backward (sdeh_bwdf.hip)
// ---------------------------------------------------------------------------------------------------------
int32_t sdeh_ctrl_backward_fused_supported(const SdehPlan* plan, const SdehProblem* pr) {
  OptScope opt_scope(plan);
  if (plan == nullptr || pr == nullptr || plan->wide) return 0;
  const SdehFourierMLP& net = pr->base_model;
  if (net.channels != 64 || !bwdf_fits(net.dim, net.n_hidden)) return 0;
  if (pr->flags & (SDEH_FLAG_INFERENCE_CTRL | SDEH_FLAG_INFERENCE_SDE)) return 0;
  if (plan_opt(OPT_BWD_PLANES) != nullptr) return 0;  // A/B aid: the plane-writing kernels (a plan option)
  const bool bptt = !(pr->flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  if (bptt && (pr->flags & SDEH_FLAG_INIT_LOGP)) return 0;
  // The training forward of the fused path is the wave-specialised kernel.  A launch the single-wave kernel would serve (mixture
  // tables beyond LDS, SDEH_LEGACY) keeps none of its planes: say so BEFORE the caller launches, instead of integrating twice.
  if (plan_opt(OPT_LEGACY) != nullptr) return 0;
  if (pr->target.kind == SDEH_DENS_GMM && plan->variant != nullptr) {
    const bool shared = pr->target.flags & SDEH_DENS_FLAG_SHARED_SCALE;
    if (make_layout(plan->variant->dp, net.channels, net.n_hidden, 1, pr->target.n_components, 1, shared).gmm_lds == 0) return 0;
  }
  return 1;
}

// Which launch serves a fused backward (one place: sdeh_ctrl_backward_fused* and sdeh_ctrl_backward_fused_reads_zrec agree by construction)
struct BwdfChoice {
  int tile;    // trajectories per tile: 16 (sdeh_bwdf16.hip: small batches through time) or 32
  bool scan;   // through time as a scan (d <= 4, sdeh_bwdf2.hip)
  bool v2;     // tiles of 32: trajectory-split teams (sdeh_bwdf2.hip) instead of channel-split ones (sdeh_bwdf.hip)
  bool zin;    // the launch reads the pre-activation record when it is given one
};
static BwdfChoice bwdf_choice(const SdehProblem* pr, long long batch, bool klb, bool have_zrec) {
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  const bool bptt = !(pr->flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  BwdfChoice c;
  c.tile = bwdf_tile(batch, bptt, net.activation);
  // Through time with d <= 4 below the batch that fills the chip with whole-tile items: the SCAN form (sdeh_bwdf2.hip) -- two
  // row-parallel network passes around a recursion on d numbers per trajectory instead of one dependent chain through the network per
  // step.  Plan option SDEH_BWD_SCAN: "0" never, "1" at every batch.
  const char* scan_opt = plan_opt(OPT_BWD_SCAN);
  c.scan = bptt && bwdf2_scan_fits(d, net.n_hidden) && plan_opt(OPT_BWD_V1) == nullptr && plan_opt(OPT_BWD_TILE) == nullptr &&
           (scan_opt != nullptr ? scan_opt[0] == '1' : batch <= 3072) && batch <= 65536;  // (measured, d = 2, T = 100: 0.20 / 0.41 / 0.65 ms at 512 / 2048 / 4096 against 0.59 / 0.59 / 0.60)
  if (c.scan) c.tile = 32;
  // tiles of 32: trajectory-split teams (sdeh_bwdf2.hip; at most as many partial records as the channel-split kernel, whose sizes
  // the scratch was checked against) unless the network has three hidden layers; plan option SDEH_BWD_V1 keeps the channel-split kernel (A/B)
  const bool force_v1 = plan_opt(OPT_BWD_V1) != nullptr, force_v2 = plan_opt(OPT_BWD_V2) != nullptr;
  // (two coordinate tiles through time: the trajectory-split kernel still spills there and loses to the channel-split one -- 22 vs
  // 15 ms at d = 50, B = 65 536; its funnel Jacobian would couple the two tiles)
  // Through time an item is a whole tile: the channel-split kernel gives every 32 trajectories two SIMDs and fills the chip with 512
  // tiles in one round (1.5 ms at d = 2, T = 100), the trajectory-split one gives them one SIMD and needs 1024 (2.2 ms for up to 1024
  // tiles against the channel-split kernel's two rounds = 3.1 ms): the latter from 513 tiles on.  Row-parallel launches always have
  // items to spare.
  const long long n_tiles = (batch + c.tile - 1) / c.tile;
  const bool enough = !bptt || n_tiles > 512 || force_v2;
  const char* zo = plan_opt(OPT_BWD_ZREC);
  // (both tilings of 32 and the four-wave teams of 16 read the record)
  c.zin = (zo == nullptr || zo[0] != '0') && !klb && (c.scan || c.tile == 32 || bwdf16_waves(batch) == 4);
  // Two coordinate tiles through time: re-evaluating, the trajectory-split kernel spills (19.7 ms at d = 50, B = 65 536, T = 200 against the
  // channel-split kernel's 15.2); READING THE RECORD it holds one layer's act' and the records in flight instead of the whole network and
  // wins (11.3 against 12.4 ms, round 5) -- the default whenever the launch is given the record (a funnel's Jacobian couples the two tiles:
  // channel-split).
  const bool v2_two_tiles = (force_v2 || (have_zrec && c.zin)) && pr->target.kind != SDEH_DENS_FUNNEL;
  c.v2 = c.tile == 32 && !force_v1 && enough && bwdf2_fits(d, net.n_hidden) && (d <= 32 || (!bptt) || v2_two_tiles);
  return c;
}

int32_t sdeh_ctrl_backward_fused_reads_zrec(const SdehPlan* plan, const SdehProblem* pr, int64_t batch) {
  OptScope opt_scope(plan);
  if (batch < 1 || !sdeh_ctrl_backward_fused_supported(plan, pr)) return 0;
  return bwdf_choice(pr, batch, false, true).zin ? 1 : 0;
}

static void bwdf_sizes(int d, int n_hidden, int n_steps, long long batch, int g, bool bptt, int tile, long long* wpart, long long* epart,
                       long long* gpart, long long* sums, long long* out) {
  const long long tiles = (batch + tile - 1) / tile, slots = tile == 16 ? bwdf16_slots(batch) : bwdf_slots(batch, n_steps, bptt);
  const long long ws = bwdf_wsize(d, n_hidden);
  const long long gw = g == 1 ? 2 : 64;
  *wpart = slots * ws;
  *epart = tiles * n_steps * 64;
  *gpart = tiles * n_steps * gw;
  *sums = ((slots + 31) / 32) * ws + ((tiles + 31) / 32) * n_steps * (64 + gw);
  *out = ws + (long long)n_steps * (64 + gw);
}

int32_t sdeh_ctrl_backward_fused_sizes(int32_t dim, int32_t n_hidden, int32_t n_steps, int64_t batch, int32_t gamma_dim, int32_t bptt,
                                       int64_t* scratch_floats, int64_t* out_floats) {
  if (!bwdf_fits(dim, n_hidden) || n_steps < 1 || batch < 1 || gamma_dim < 1 || scratch_floats == nullptr || out_floats == nullptr)
    return fail(SDEH_ERR_INVALID, "ctrl_backward_fused_sizes: bad argument");
  long long w, e, g, s, o;
  bwdf_sizes(dim, n_hidden, n_steps, batch, gamma_dim == 1 ? 1 : 64, bptt != 0, 32, &w, &e, &g, &s, &o);
  *scratch_floats = w + e + g + s;
  *out_floats = o;
  // the activation and the plan's SDEH_BWD_TILE option decide between the two tilings as well (sdeh_bwdf16.hip) and are not arguments
  // here: room for either wherever the 16-trajectory kernel can be asked for (through time, up to 65 536 trajectories)
  if (bptt != 0 && batch <= 65536) {
    bwdf_sizes(dim, n_hidden, n_steps, batch, gamma_dim == 1 ? 1 : 64, true, 16, &w, &e, &g, &s, &o);
    if (w + e + g + s > *scratch_floats) *scratch_floats = w + e + g + s;
  }
  // the scan form of back-propagation through time (d <= 4): a row-parallel launch's partial records + the planes nn, J, G
  if (bptt != 0 && bwdf2_scan_fits(dim, n_hidden) && batch <= 65536) {
    bwdf_sizes(dim, n_hidden, n_steps, batch, gamma_dim == 1 ? 1 : 64, false, 32, &w, &e, &g, &s, &o);
    const long long planes = (long long)n_steps * batch * (dim * dim + 2 * dim);
    if (w + e + g + s + planes > *scratch_floats) *scratch_floats = w + e + g + s + planes;
  }
  return SDEH_OK;
}

static int ctrl_backward_fused_impl(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                    int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                    const float* grad_rnd, const float* sc, const float* tscore, const float* cost_ctrl, const float* lam_extra,
                                    float* scratch, int64_t scratch_floats, float* out, void* stream, const float* zrec = nullptr) {
  OptScope opt_scope(plan);
  if (xs == nullptr || grad_rnd == nullptr || scratch == nullptr || out == nullptr)
    return fail(SDEH_ERR_INVALID, "ctrl_backward_fused: null argument");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, row_offset, false, &ck);
  if (rc != SDEH_OK) return rc;
  if (!sdeh_ctrl_backward_fused_supported(plan, pr))
    return fail(SDEH_ERR_UNSUPPORTED, "ctrl_backward_fused: compiled for channels = 64, one to three hidden layers, "
                                      "d <= 64, no inference control (sdeh_ctrl_backward_ex + sdeh_weight_grad take the rest)");
  const bool bptt = !(pr->flags & SDEH_FLAG_CHANGE_SDE_CTRL);
  if ((cost_ctrl != nullptr) != (lam_extra != nullptr))
    return fail(SDEH_ERR_INVALID, "ctrl_backward_fused_ex: cost_ctrl and lam_extra come together (a Bridge's generative network, method kl)");
  if (cost_ctrl != nullptr && pr->base_model.n_hidden != 2)
    return fail(SDEH_ERR_UNSUPPORTED, "ctrl_backward_fused_ex: cost_ctrl / lam_extra are compiled for two hidden layers");
  if ((cost_ctrl != nullptr || lam_extra != nullptr) && !bptt)
    return fail(SDEH_ERR_INVALID, "ctrl_backward_fused_ex: cost_ctrl / lam_extra belong to back-propagation through time (methods kl / kl_ito)");
  // the kernels address the coordinate-major planes [d][B] with 32-bit byte offsets (sdeh_bwdf.hip: load_cm16)
  if ((long long)(pr->base_model.dim <= 32 ? 32 : 64) * batch * 4 >= (1ll << 32))
    return fail(SDEH_ERR_CAPACITY, "ctrl_backward_fused: %lld trajectories: the coordinate-major planes are addressed with 32-bit byte offsets "
                                   "(64 * batch * 4 < 2^32); sdeh_ctrl_backward_ex + sdeh_weight_grad take larger batches", (long long)batch);
  if (pr->ctrl_kind != SDEH_CTRL_CLIPPED && sc == nullptr) return fail(SDEH_ERR_INVALID, "ctrl_backward_fused: sc is null");
  if (bptt && (pr->flags & SDEH_FLAG_TERMINAL_TARGET) && tscore == nullptr)
    return fail(SDEH_ERR_INVALID, "ctrl_backward_fused: tscore is null (back-propagation through time with a terminal target cost)");
  const WsLayout& L = ck.L;
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  long long n_w, n_e, n_g, n_s, n_o;
  const BwdfChoice choice = bwdf_choice(pr, batch, cost_ctrl != nullptr, zrec != nullptr);
  const int tile = choice.tile;
  const bool scan = choice.scan;
  bwdf_sizes(d, net.n_hidden, n_steps, batch, L.g == 1 ? 1 : 64, bptt && !scan, tile, &n_w, &n_e, &n_g, &n_s, &n_o);
  const long long n_planes = scan ? (long long)n_steps * batch * (d * d + 2 * d) : 0;
  if (scratch_floats < n_w + n_e + n_g + n_s + n_planes)
    return fail(SDEH_ERR_CAPACITY, "ctrl_backward_fused: scratch too small (%lld < %lld floats)", (long long)scratch_floats,
                n_w + n_e + n_g + n_s + n_planes);
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = L; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "ctrl_backward_fused: prep kernel launch failed");
  BwdfArgs A;
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L;
  A.w_in = net.input_w; A.w_out = net.out_w; A.b_out = net.out_b;
  for (int l = 0; l < net.n_hidden; ++l) { A.w_hid[l] = net.hidden_w[l]; A.b_hid[l] = net.hidden_b[l]; }
  A.n_hidden = net.n_hidden;
  A.xs = xs; A.noise = noise; A.grad_rnd = grad_rnd; A.sc = sc; A.tscore = tscore;
  A.cost_in = cost_ctrl; A.lam_in = lam_extra;
  // the pre-activation record: the kernels that can read it do not re-evaluate the network (plan option SDEH_BWD_ZREC=0: ignore it)
  if (choice.zin) A.zrec = zrec;
  A.wpart = scratch; A.epart = scratch + n_w; A.gpart = scratch + n_w + n_e;
  float* sums = scratch + n_w + n_e + n_g;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = d; A.n_kg = (d + 7) / 8;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = net.activation;
  A.g = L.g; A.gw = L.g == 1 ? 2 : 64;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  A.n_tiles = (int)((batch + tile - 1) / tile);
  const bool v2 = choice.v2;
  A.n_slots = tile == 16 ? bwdf16_slots(batch) : (v2 ? bwdf2_slots(batch, n_steps, bptt) : bwdf_slots(batch, n_steps, bptt));
  A.wsize = bwdf_wsize(d, net.n_hidden);
  timing_begin(plan, st);
  if (scan) {
    // nn [T, d, B] | J [T, d, d, B] | G [T, d, B] behind the partial records
    float* planes = sums + n_s;
    A.nn_out = planes;
    A.jac_out = planes + (long long)n_steps * batch * d;
    A.gq_out = A.jac_out + (long long)n_steps * batch * d * d;
    A.n_slots = bwdf2_slots(batch, n_steps, false);
    rc = launch_bwdf2_jac(A, st);
    if (rc == SDEH_OK) rc = launch_bwdf2_scan(A, st);
    if (rc == SDEH_OK) {
      BwdfArgs R = A;  // the row-parallel backward (the lv form) with the scan's upstream gradient
      R.flags |= SDEH_FLAG_CHANGE_SDE_CTRL;
      R.gq_in = A.gq_out;
      rc = launch_bwdf2(R, st);
    }
  } else {
    rc = tile == 16 ? launch_bwdf16(A, st) : (v2 ? launch_bwdf2(A, st) : launch_bwdf(A, st));
  }
  timing_end(plan, st);
  const bool zin = A.zrec != nullptr;  // (the launch read the record)
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), "bwd_fused%s<%s,tiles=%d%s%s>", tile == 16 ? "16" : "", scan ? "bptt-scan" : (bptt ? "bptt" : "rows"),
           d <= 32 ? 1 : 2, tile == 16 ? "" : (v2 || scan ? ",traj-split" : ",chan-split"), zin ? ",zrec" : "");
  if (rc != SDEH_OK) return fail(rc, "ctrl_backward_fused: kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  // deterministic sums over the teams / tiles
  float* s1 = sums;
  float* s2 = s1 + ((A.n_slots + 31) / 32) * (long long)A.wsize;
  SumJob job;
  memset(&job, 0, sizeof(job));
  job.s[0] = {A.wpart, s1, out, A.n_slots, A.wsize};
  job.s[1] = {A.epart, s2, out + A.wsize, A.n_tiles, (long long)n_steps * 64};
  job.n = 2;
  if (pr->ctrl_kind != SDEH_CTRL_CLIPPED)
    job.s[job.n++] = {A.gpart, s2 + ((A.n_tiles + 31) / 32) * (long long)n_steps * 64, out + A.wsize + (long long)n_steps * 64, A.n_tiles,
                      (long long)n_steps * A.gw};
  rc = launch_partial_sums_multi(job, st);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "ctrl_backward_fused: partial sums failed");
}

int32_t sdeh_ctrl_backward_fused(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                 int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                 const float* grad_rnd, const float* sc, const float* tscore, float* scratch,
                                 int64_t scratch_floats, float* out, void* stream) {
  return ctrl_backward_fused_impl(plan, pr, ts, n_steps, xs, batch, noise, seed, offset, row_offset, grad_rnd, sc, tscore, nullptr, nullptr,
                                  scratch, scratch_floats, out, stream);
}

int32_t sdeh_ctrl_backward_fused_ex(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                    int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                    const float* grad_rnd, const float* sc, const float* tscore, const float* cost_ctrl,
                                    const float* lam_extra, float* scratch, int64_t scratch_floats, float* out, void* stream) {
  return ctrl_backward_fused_impl(plan, pr, ts, n_steps, xs, batch, noise, seed, offset, row_offset, grad_rnd, sc, tscore, cost_ctrl, lam_extra,
                                  scratch, scratch_floats, out, stream);
}

int32_t sdeh_ctrl_backward_fused_z(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                   int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                   const float* grad_rnd, const float* sc, const float* tscore, const float* cost_ctrl,
                                   const float* lam_extra, const float* zrec, float* scratch, int64_t scratch_floats, float* out,
                                   void* stream) {
  return ctrl_backward_fused_impl(plan, pr, ts, n_steps, xs, batch, noise, seed, offset, row_offset, grad_rnd, sc, tscore, cost_ctrl, lam_extra,
                                  scratch, scratch_floats, out, stream, zrec);
}

// ---------------------------------------------------------------------------------------------------------
// Bridge, 64 channels: every gradient of the INFERENCE network in three launches, no per-coordinate planes (sdeh_bridgef.hip)
// ---------------------------------------------------------------------------------------------------------
static void bridgef_sizes(int d, int n_steps, long long batch, int g, long long* fused, long long* planes, long long* hid, long long* io,
                          long long* sums, long long* out) {
  long long w, e, gp, s, o;
  bwdf_sizes(d, 2, n_steps, batch, g, false, 32, &w, &e, &gp, &s, &o);
  const long long slots = bwdf2_slots(batch, n_steps, false), dpp = d <= 32 ? 32 : 64;
  *fused = w + e + gp + s;
  *planes = 3LL * 64 * n_steps * batch;
  *hid = slots * 2 * 4096;
  *io = slots * 4 * 2 * dpp * 64;
  *sums = ((slots + 31) / 32) * 2 * 4096 + ((slots * 4 + 31) / 32) * 2 * dpp * 64;
  *out = o + 2 * 4096 + 2 * dpp * 64;
}

int64_t sdeh_bridge_inference_fwd_scratch_floats(int32_t n_steps, int64_t batch) {
  if (n_steps < 1 || batch < 1) return 0;
  return (int64_t)n_steps * batch + (int64_t)((n_steps + 31) / 32) * batch;
}

int32_t sdeh_bridge_inference_fwd(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                  int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset, const float* u,
                                  float* drnd, float* cost_ctrl, float* scratch, int64_t scratch_floats, void* stream) {
  OptScope opt_scope(plan);
  if (xs == nullptr || u == nullptr || drnd == nullptr || cost_ctrl == nullptr || scratch == nullptr)
    return fail(SDEH_ERR_INVALID, "bridge_inference_fwd: null argument");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, row_offset, false, &ck);
  if (rc != SDEH_OK) return rc;
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  if (plan->wide || net.channels != 64 || !bridge_divf_fits(d, net.n_hidden) || (pr->flags & (SDEH_FLAG_INFERENCE_CTRL | SDEH_FLAG_INFERENCE_SDE)) ||
      (pr->ctrl_kind != SDEH_CTRL_CLIPPED && pr->ctrl_kind != SDEH_CTRL_LERP_PRIOR))
    return fail(SDEH_ERR_UNSUPPORTED, "bridge_inference_fwd: the inference control as the control of a plain problem (ClippedCtrl / LerpPriorCtrl, "
                                      "64 channels, two hidden layers, d <= 64)");
  if ((long long)(d <= 32 ? 32 : 64) * batch * 4 >= (1ll << 32))
    return fail(SDEH_ERR_CAPACITY, "bridge_inference_fwd: %lld trajectories: the planes are addressed with 32-bit byte offsets", (long long)batch);
  if (scratch_floats < sdeh_bridge_inference_fwd_scratch_floats(n_steps, batch)) return fail(SDEH_ERR_CAPACITY, "bridge_inference_fwd: scratch too small");
  const WsLayout& L = ck.L;
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = L; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "bridge_inference_fwd: prep kernel launch failed");
  BwdfArgs A;
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L;
  A.w_in = net.input_w; A.w_out = net.out_w; A.b_out = net.out_b;
  for (int l = 0; l < net.n_hidden; ++l) { A.w_hid[l] = net.hidden_w[l]; A.b_hid[l] = net.hidden_b[l]; }
  A.n_hidden = net.n_hidden;
  A.xs = xs; A.noise = noise; A.u_in = u; A.gp_out = cost_ctrl; A.drnd_out = scratch;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = d;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = net.activation;
  A.g = L.g; A.gw = L.g == 1 ? 2 : 64;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  A.n_tiles = (int)((batch + 31) / 32);
  timing_begin(plan, st);
  rc = launch_bridge_rowsf(A, st);
  timing_end(plan, st);
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), "bridge_rows_fwd<tiles=%d>", d <= 32 ? 1 : 2);
  if (rc != SDEH_OK) return fail(rc, "bridge_inference_fwd: kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  rc = launch_partial_sums(scratch, 1, n_steps, batch, scratch + (long long)n_steps * batch, drnd, st);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "bridge_inference_fwd: partial sums failed");
}

int32_t sdeh_bridge_backward_fused_sizes(int32_t dim, int32_t n_hidden, int32_t n_steps, int64_t batch, int32_t gamma_dim,
                                         int64_t* scratch_floats, int64_t* out_floats) {
  if (!bridge_divf_fits(dim, n_hidden) || !bwdf2_fits(dim, n_hidden) || n_steps < 1 || batch < 1 || gamma_dim < 1 || scratch_floats == nullptr ||
      out_floats == nullptr)
    return fail(SDEH_ERR_UNSUPPORTED, "bridge_backward_fused_sizes: compiled for d <= 64 and two hidden layers of 64 channels");
  long long f, p, hd, io, sm, o;
  bridgef_sizes(dim, n_steps, batch, gamma_dim == 1 ? 1 : 64, &f, &p, &hd, &io, &sm, &o);
  *scratch_floats = f + p + hd + io + sm;
  *out_floats = o;
  return SDEH_OK;
}

int32_t sdeh_bridge_backward_fused(SdehPlan* plan, const SdehProblem* pr, const float* ts, int32_t n_steps, const float* xs,
                                   int64_t batch, const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset,
                                   const float* grad_rnd, const float* cost_ctrl, float* dx_out, float* scratch, int64_t scratch_floats,
                                   float* out, void* stream) {
  OptScope opt_scope(plan);
  if (xs == nullptr || grad_rnd == nullptr || cost_ctrl == nullptr || scratch == nullptr || out == nullptr)
    return fail(SDEH_ERR_INVALID, "bridge_backward_fused: null argument");
  Checked ck;
  int rc = check_problem(plan, pr, ts, n_steps, batch, row_offset, false, &ck);
  if (rc != SDEH_OK) return rc;
  const SdehFourierMLP& net = pr->base_model;
  const int d = net.dim;
  if (plan->wide || net.channels != 64 || !bridge_divf_fits(d, net.n_hidden) || !bwdf2_fits(d, net.n_hidden) ||
      (pr->flags & (SDEH_FLAG_INFERENCE_CTRL | SDEH_FLAG_INFERENCE_SDE)) || !(pr->flags & SDEH_FLAG_CHANGE_SDE_CTRL) ||
      (pr->ctrl_kind != SDEH_CTRL_CLIPPED && pr->ctrl_kind != SDEH_CTRL_LERP_PRIOR))
    return fail(SDEH_ERR_UNSUPPORTED, "bridge_backward_fused: the inference control as a plain row-parallel problem (ClippedCtrl / LerpPriorCtrl, "
                                      "64 channels, two hidden layers, d <= 64); sdeh_ctrl_backward_ex + sdeh_bridge_div_backward take the rest");
  // 32-bit byte offsets: the coordinate-major planes [d][B] per step, the S planes [64][T B] per layer
  if ((long long)(d <= 32 ? 32 : 64) * batch * 4 >= (1ll << 32) || 64LL * n_steps * batch * 4 >= (1ll << 32))
    return fail(SDEH_ERR_CAPACITY, "bridge_backward_fused: %lld x %d rows: the planes are addressed with 32-bit byte offsets", (long long)batch, n_steps);
  const WsLayout& L = ck.L;
  long long n_f, n_p, n_h, n_io, n_sm, n_o;
  bridgef_sizes(d, n_steps, batch, L.g == 1 ? 1 : 64, &n_f, &n_p, &n_h, &n_io, &n_sm, &n_o);
  if (scratch_floats < n_f + n_p + n_h + n_io + n_sm)
    return fail(SDEH_ERR_CAPACITY, "bridge_backward_fused: scratch too small (%lld < %lld floats)", (long long)scratch_floats, n_f + n_p + n_h + n_io + n_sm);
  long long n_w, n_e, n_g, n_s, n_o1;
  bwdf_sizes(d, 2, n_steps, batch, L.g == 1 ? 1 : 64, false, 32, &n_w, &n_e, &n_g, &n_s, &n_o1);
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = L; P.prob = *pr; P.ts = ts; P.n_steps = n_steps;
  P.ts_out = nullptr; P.n_out = 0; P.eps = 0.0f;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "bridge_backward_fused: prep kernel launch failed");
  BwdfArgs A;
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = L;
  A.w_in = net.input_w; A.w_out = net.out_w; A.b_out = net.out_b;
  for (int l = 0; l < net.n_hidden; ++l) { A.w_hid[l] = net.hidden_w[l]; A.b_hid[l] = net.hidden_b[l]; }
  A.n_hidden = net.n_hidden;
  A.xs = xs; A.noise = noise; A.grad_rnd = grad_rnd; A.gextra = cost_ctrl; A.dx_out = dx_out;
  A.wpart = scratch; A.epart = scratch + n_w; A.gpart = scratch + n_w + n_e;
  float* sums = scratch + n_w + n_e + n_g;
  float* planes = scratch + n_f;
  A.s_out = planes; A.s_in = planes;
  A.div_hid = planes + n_p; A.div_io = A.div_hid + n_h;
  float* sums2 = A.div_io + n_io;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = d; A.n_kg = (d + 7) / 8;
  A.loss_kind = pr->loss_kind; A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags; A.act = net.activation;
  A.g = L.g; A.gw = L.g == 1 ? 2 : 64;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  A.n_tiles = (int)((batch + 31) / 32);
  A.n_slots = bwdf2_slots(batch, n_steps, false);
  A.wsize = bwdf_wsize(d, net.n_hidden);
  const long long dpp = d <= 32 ? 32 : 64;
  timing_begin(plan, st);
  rc = launch_divf_zero(A.div_io, n_io, st);
  if (rc == SDEH_OK) rc = launch_bridge_divf(A, st);
  if (rc == SDEH_OK) rc = launch_bwdf2_bridge(A, st);
  timing_end(plan, st);
  snprintf(plan->last_kernel, sizeof(plan->last_kernel), "bridge_bwd_fused<tiles=%d>", d <= 32 ? 1 : 2);
  if (rc != SDEH_OK) return fail(rc, "bridge_backward_fused: kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  float* s1 = sums;
  float* s2 = s1 + ((A.n_slots + 31) / 32) * (long long)A.wsize;
  float* o2 = out + n_o1;
  SumJob job;
  memset(&job, 0, sizeof(job));
  job.s[0] = {A.wpart, s1, out, A.n_slots, A.wsize};
  job.s[1] = {A.epart, s2, out + A.wsize, A.n_tiles, (long long)n_steps * 64};
  job.s[2] = {A.div_hid, sums2, o2, A.n_slots, 2 * 4096};
  job.s[3] = {A.div_io, sums2 + ((A.n_slots + 31) / 32) * 2 * 4096, o2 + 2 * 4096, (long long)A.n_slots * 4, 2 * dpp * 64};
  job.n = 4;
  if (pr->ctrl_kind != SDEH_CTRL_CLIPPED)
    job.s[job.n++] = {A.gpart, s2 + ((A.n_tiles + 31) / 32) * (long long)n_steps * 64, out + A.wsize + (long long)n_steps * 64, A.n_tiles,
                      (long long)n_steps * A.gw};
  rc = launch_partial_sums_multi(job, st);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "bridge_backward_fused: partial sums failed");
}

int32_t sdeh_integrate(SdehPlan* plan, const SdehProblem* pr, int32_t kind, const float* timesteps, int32_t n_steps,
                       const float* ts_out, int32_t n_out, float eps, const float* x_init, int64_t batch,
                       const float* noise, uint64_t seed, uint64_t offset, int64_t row_offset, float* xs_out,
                       void* stream) {
  OptScope opt_scope(plan);
  if (plan == nullptr || pr == nullptr || timesteps == nullptr || ts_out == nullptr || x_init == nullptr || xs_out == nullptr)
    return fail(SDEH_ERR_INVALID, "integrate: null argument");
  if (kind != SDEH_INT_LANGEVIN && kind != SDEH_INT_CONTROLLED) return fail(SDEH_ERR_INVALID, "integrate: kind %d", kind);
  if (plan->wide) return fail(SDEH_ERR_UNSUPPORTED, "integrate: the plain integrator is compiled for channels = 64 plans (d <= 64)");
  if (n_out < 1 || n_out > plan->desc.max_steps + 1)
    return fail(SDEH_ERR_CAPACITY, "integrate: %d output times (plan allows %d)", n_out, plan->desc.max_steps + 1);
  if (pr->sde_kind == SDEH_SDE_NONE) return fail(SDEH_ERR_INVALID, "integrate: needs an sde");
  if (kind == SDEH_INT_LANGEVIN && pr->ctrl_kind != SDEH_CTRL_NONE)
    return fail(SDEH_ERR_INVALID, "integrate: LangevinSDE has no control (ctrl_kind must be SDEH_CTRL_NONE)");
  const int d = pr->base_model.dim;
  Checked ck;
  int rc;
  if (pr->ctrl_kind == SDEH_CTRL_NONE) {
    if (batch < 1 || n_steps < 1) return fail(SDEH_ERR_INVALID, "integrate: batch=%lld n_steps=%d", (long long)batch, n_steps);
    if (row_offset < 0 || (unsigned long long)row_offset + (unsigned long long)batch > 0x100000000ull)
      return fail(SDEH_ERR_INVALID, "integrate: global row indices must fit 32 bits");
    if (d != plan->desc.dim) return fail(SDEH_ERR_CAPACITY, "integrate: dim %d differs from the plan's %d", d, plan->desc.dim);
    if (n_steps > plan->desc.max_steps) return fail(SDEH_ERR_CAPACITY, "integrate: %d steps > plan max %d", n_steps, plan->desc.max_steps);
    rc = check_density(pr->target, d, "target", kind != SDEH_INT_LANGEVIN);
    if (rc != SDEH_OK) return rc;
    const int k = pr->target.kind == SDEH_DENS_GMM ? pr->target.n_components : 0;
    if (k > plan->desc.max_components) return fail(SDEH_ERR_CAPACITY, "integrate: GMM with %d components > plan max %d", k, plan->desc.max_components);
    ck.v = plan->variant;
    ck.refc = false;
    ck.L = make_layout(ck.v->dp, plan->desc.channels, 0, n_steps, k, 1,
                       k > 0 && (pr->target.flags & SDEH_DENS_FLAG_SHARED_SCALE), false);
    if ((size_t)ck.L.total > plan->ws_floats) return fail(SDEH_ERR_CAPACITY, "integrate: workspace too small");
  } else {
    SdehProblem q = *pr;  // the loss-specific fields are not used by the integrator
    q.loss_kind = SDEH_LOSS_TIME_REVERSAL;
    q.flags &= SDEH_FLAG_INFERENCE_SDE;
    rc = check_problem(plan, &q, timesteps, n_steps, batch, row_offset, false, &ck, true);
    if (rc != SDEH_OK) return rc;
  }
  hipStream_t st = (hipStream_t)stream;
  PrepArgs P;
  P.ws = plan->ws; P.lay = ck.L; P.prob = *pr; P.ts = timesteps; P.n_steps = n_steps;
  P.prob.flags &= SDEH_FLAG_INFERENCE_SDE;
  P.ts_out = ts_out; P.n_out = n_out; P.eps = eps;
  rc = launch_prep(P, st);
  if (rc != SDEH_OK) return fail(rc, "integrate: prep kernel launch failed");
  TrajArgs A{};
  memset(&A, 0, sizeof(A));
  A.ws = plan->ws; A.lay = ck.L;
  A.x0 = x_init; A.noise = noise; A.xs = xs_out;
  A.batch = batch; A.row_offset = row_offset; A.n_steps = n_steps; A.d = d;
  A.ctrl_kind = pr->ctrl_kind; A.flags = pr->flags & SDEH_FLAG_INFERENCE_SDE; A.act = pr->base_model.activation;
  A.clip_model = pr->clip_model; A.clip_score = pr->clip_score; A.scale_score = pr->scale_score;
  A.target = {pr->target.kind, pr->target.n_components, pr->target.log_norm_const, pr->target.p0, pr->target.p1};
  A.prior = {pr->prior.kind, pr->prior.n_components, pr->prior.log_norm_const, pr->prior.p0, pr->prior.p1};
  A.seed = seed; A.offset = offset; A.rng_dev = reinterpret_cast<const unsigned long long*>(pr->rng_offset_dev);
  A.int_kind = kind; A.n_out = n_out; A.ts_out = ts_out;
  timing_begin(plan, st);
  rc = plan->variant->fn_int(A, st);
  timing_end(plan, st);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "integrate: kernel launch failed (dp=%d)", plan->variant->dp);
}

static constexpr int kSinkMaxSplits = 64;
static constexpr int kStatBlocks = 256;

// ---- the NICE flow target (BASELINE configs[4]; reference distr/nice.py) ---------------------------------------------------------------
static int check_nice(const SdehNice* nn) {
  if (nn == nullptr) return fail(SDEH_ERR_INVALID, "nice: null description");
  if (nn->dim < 2 || (nn->dim & 1) || nn->dim > 256) return fail(SDEH_ERR_INVALID, "nice: dim=%d (even, <= 256)", nn->dim);
  if (nn->n_coupling < 1 || nn->n_coupling > SDEH_NICE_MAX_COUPLING) return fail(SDEH_ERR_INVALID, "nice: %d couplings", nn->n_coupling);
  if (nn->mid_dim < 4 || (nn->mid_dim & 3)) return fail(SDEH_ERR_UNSUPPORTED, "nice: mid_dim=%d (a multiple of 4)", nn->mid_dim);
  if (nn->n_mid < 0 || nn->n_mid > SDEH_MAX_HIDDEN) return fail(SDEH_ERR_INVALID, "nice: %d mid blocks", nn->n_mid);
  if (nn->scale == nullptr) return fail(SDEH_ERR_INVALID, "nice: null scale");
  for (int c = 0; c < nn->n_coupling; ++c) {
    if (nn->in_w[c] == nullptr || nn->in_b[c] == nullptr || nn->out_w[c] == nullptr || nn->out_b[c] == nullptr)
      return fail(SDEH_ERR_INVALID, "nice: null parameter of coupling %d", c);
    for (int l = 0; l < nn->n_mid; ++l)
      if (nn->mid_w[c][l] == nullptr || nn->mid_b[c][l] == nullptr) return fail(SDEH_ERR_INVALID, "nice: null mid block %d of coupling %d", l, c);
  }
  return SDEH_OK;
}

int64_t sdeh_nice_work_floats(const SdehNice* nice, int64_t batch, int32_t want_score) {
  if (check_nice(nice) != SDEH_OK || batch < 1) return -1;
  return nice_work_floats(*nice, batch, want_score != 0);
}

int32_t sdeh_nice_eval(const SdehNice* nice, const float* x, int64_t batch, float* score, float* logp, float* work, int64_t work_floats,
                       void* stream) {
  int rc = check_nice(nice);
  if (rc != SDEH_OK) return rc;
  if (x == nullptr || work == nullptr || batch < 1 || (score == nullptr && logp == nullptr)) return fail(SDEH_ERR_INVALID, "nice_eval: null argument");
  if (work_floats < nice_work_floats(*nice, batch, score != nullptr))
    return fail(SDEH_ERR_CAPACITY, "nice_eval: work memory of %lld floats < sdeh_nice_work_floats = %lld", (long long)work_floats,
                nice_work_floats(*nice, batch, score != nullptr));
  if ((reinterpret_cast<unsigned long long>(work) & 15) != 0) return fail(SDEH_ERR_INVALID, "nice_eval: work memory must be 16-byte aligned");
  rc = launch_nice_eval(*nice, x, batch, score, logp, work, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "nice_eval: kernel launch failed");
}

int64_t sdeh_sinkhorn_workspace_floats(int64_t n, int64_t m) {
  if (n < 1 || m < 1) return 0;
  const int64_t mx = n > m ? n : m;
  return 2 * (n + m) + 2 * (int64_t)kSinkMaxSplits * mx + (mx + 63) / 64 + 16;
}

int32_t sdeh_sinkhorn(const float* x, int64_t n, const float* y, int64_t m, int32_t d, const float* w_x, const float* w_y,
                      int32_t p, float eps, int32_t max_iters, float stop_thresh, float* workspace, float* out,
                      int64_t* corr_x_to_y, int64_t* corr_y_to_x, void* stream) {
  if (x == nullptr || y == nullptr || workspace == nullptr || out == nullptr) return fail(SDEH_ERR_INVALID, "sinkhorn: null argument");
  if (n < 1 || m < 1 || d < 1) return fail(SDEH_ERR_INVALID, "sinkhorn: n=%lld m=%lld d=%d", (long long)n, (long long)m, d);
  if (p != 1 && p != 2) return fail(SDEH_ERR_UNSUPPORTED, "sinkhorn: p=%d (1 and 2 are built in)", p);
  if (!(eps > 0.0f) || max_iters < 1) return fail(SDEH_ERR_INVALID, "sinkhorn: eps=%g max_iters=%d", eps, max_iters);
  if ((w_x == nullptr) != (w_y == nullptr)) return fail(SDEH_ERR_INVALID, "sinkhorn: give both weight vectors or neither");
  const Variant* v = pick_variant(d);
  if (v == nullptr) return fail(SDEH_ERR_UNSUPPORTED, "sinkhorn: no kernel compiled for dim=%d", d);
  hipStream_t st = (hipStream_t)stream;
  const int64_t mx = n > m ? n : m;
  float* u = workspace;
  float* vv = u + n;
  float* log_a = vv + m;
  float* log_b = log_a + n;
  float* part_m = log_b + m;
  float* part_s = part_m + (int64_t)kSinkMaxSplits * mx;
  float* dpart = part_s + (int64_t)kSinkMaxSplits * mx;
  int* flags = reinterpret_cast<int*>(dpart + (mx + 63) / 64);
  int rc = launch_sink_init(u, vv, log_a, log_b, w_x, w_y, n, m, eps, flags, st);
  if (rc != SDEH_OK) return fail(rc, "sinkhorn: init launch failed");
  SinkArgs A;
  memset(&A, 0, sizeof(A));
  A.d = d; A.pnorm = p; A.inv_eps = (float)(1.0 / (double)eps); A.done = flags;
  A.part_m = part_m; A.part_s = part_s;
  auto splits_for = [](int64_t np, int64_t nq) {
    const int64_t rb = (np + 63) / 64, tiles = (nq + 255) / 256;
    int64_t s = (1024 + rb - 1) / rb;  // aim at >= 1024 workgroups
    if (s > tiles) s = tiles;
    if (s > kSinkMaxSplits) s = kSinkMaxSplits;
    return (int)(s < 1 ? 1 : s);
  };
  const int su = splits_for(n, m), sv = splits_for(m, n);
  for (int it = 0; it < max_iters; ++it) {
    // u_i = eps (log a_i - LSE_j((v_j - M_ij)/eps))          eval/sinkhorn.py:153-155
    A.P = x; A.np = n; A.Q = y; A.nq = m; A.pot_q = vv;
    rc = v->fn_sink(A, 0, su, st);
    if (rc == SDEH_OK) rc = launch_sink_finalize(part_m, part_s, su, n, log_a, eps, u, flags + 2, flags, st);
    // v_j = eps (log b_j - LSE_i((u_i - M_ij)/eps))          eval/sinkhorn.py:157-159
    A.P = y; A.np = m; A.Q = x; A.nq = n; A.pot_q = u;
    if (rc == SDEH_OK) rc = v->fn_sink(A, 0, sv, st);
    if (rc == SDEH_OK) rc = launch_sink_finalize(part_m, part_s, sv, m, log_b, eps, vv, flags + 3, flags, st);
    if (rc == SDEH_OK) rc = launch_sink_check(flags, stop_thresh, st);
    if (rc != SDEH_OK) return fail(rc, "sinkhorn: iteration %d launch failed", it);
  }
  // distance = sum_ij P_ij M_ij (+ argmax of the plan)            eval/sinkhorn.py:169-178
  A.done = nullptr; A.part_m = nullptr;
  A.P = x; A.np = n; A.Q = y; A.nq = m; A.pot_q = vv; A.pot_p = u; A.part_s = dpart;
  A.corr = reinterpret_cast<long long*>(corr_x_to_y);
  rc = v->fn_sink(A, 1, 1, st);
  if (rc == SDEH_OK) rc = launch_sink_dist_final(dpart, (int)((n + 63) / 64), flags, out, st);
  if (rc == SDEH_OK && corr_y_to_x != nullptr) {
    A.P = y; A.np = m; A.Q = x; A.nq = n; A.pot_q = u; A.pot_p = vv; A.part_s = nullptr;
    A.corr = reinterpret_cast<long long*>(corr_y_to_x);
    rc = v->fn_sink(A, 1, 1, st);
  }
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "sinkhorn: distance launch failed");
}

int64_t sdeh_sample_stats_scratch_floats(int32_t d) { return d < 1 ? 0 : (int64_t)kStatBlocks * (12 + 3 * (int64_t)d); }

int32_t sdeh_sample_stats(const float* samples, int64_t batch, int32_t d, const float* weights, const float* domain,
                          float* scratch, float* out, void* stream) {
  if (samples == nullptr || scratch == nullptr || out == nullptr || batch < 1 || d < 1)
    return fail(SDEH_ERR_INVALID, "sample_stats: bad argument");
  if (d > 256) return fail(SDEH_ERR_UNSUPPORTED, "sample_stats: d=%d > 256", d);
  int64_t nb = (batch + 255) / 256;
  if (nb > kStatBlocks) nb = kStatBlocks;
  const int rc = launch_sample_stats(samples, weights, domain, batch, d, scratch, (int)nb, out, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "sample_stats: launch failed");
}

int32_t sdeh_weight_grad(const float* D, int32_t m, const float* Z, int32_t c, int64_t N, int32_t act, int64_t chunk,
                         float* part_w, float* part_b, void* stream) {
  if (D == nullptr || Z == nullptr || part_w == nullptr || part_b == nullptr || N < 1)
    return fail(SDEH_ERR_INVALID, "weight_grad: bad argument");
  if (m < 1 || m > 256 || c < 1 || c > 256) return fail(SDEH_ERR_UNSUPPORTED, "weight_grad: m=%d c=%d (1..256)", m, c);
  if (chunk < 8 || (chunk & 7) != 0) return fail(SDEH_ERR_INVALID, "weight_grad: chunk=%lld must be a positive multiple of 8", (long long)chunk);
  const int rc = launch_weight_grad(D, m, Z, c, N, act, chunk, part_w, part_b, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "weight_grad: launch failed (act=%d)", act);
}

int64_t sdeh_time_embed_param_floats(const SdehTimeEmbed* te) { return te == nullptr || te->n_hidden < 1 ? 0 : time_embed_param_floats(*te); }
int64_t sdeh_time_embed_workspace_floats(const SdehTimeEmbed* te, int32_t n_steps) {
  return te == nullptr || te->n_hidden < 1 || n_steps < 1 ? 0 : time_embed_workspace_floats(*te, n_steps);
}

int32_t sdeh_time_embed_backward(const SdehTimeEmbed* te, int32_t activation, const float* ts, int32_t n_steps,
                                 const float* grad_table, float clip_out, float* workspace, float* grad_flat, void* stream) {
  if (te == nullptr || ts == nullptr || grad_table == nullptr || workspace == nullptr || grad_flat == nullptr || n_steps < 1)
    return fail(SDEH_ERR_INVALID, "time_embed_backward: bad argument");
  if (te->channels < 1 || te->dim_out < 1 || te->coeff == nullptr || te->phase == nullptr || te->out_w == nullptr ||
      te->out_b == nullptr)
    return fail(SDEH_ERR_INVALID, "time_embed_backward: null parameter pointer");
  for (int k = 0; k < te->n_hidden && k < SDEH_MAX_HIDDEN; ++k)
    if (te->hidden_w[k] == nullptr || te->hidden_b[k] == nullptr)
      return fail(SDEH_ERR_INVALID, "time_embed_backward: null hidden layer %d", k);
  if (activation < SDEH_ACT_GELU_ERF || activation > SDEH_ACT_RELU) return fail(SDEH_ERR_INVALID, "time_embed_backward: activation");
  const int rc = launch_time_embed_bwd(*te, activation, ts, n_steps, grad_table, clip_out, workspace, grad_flat, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "time_embed_backward: n_hidden=%d channels=%d not supported by this kernel", te->n_hidden, te->channels);
}

int64_t sdeh_partial_sums_scratch_floats(int64_t n_items, int64_t n_chunks, int64_t width) {
  return n_items * ((n_chunks + 31) / 32) * width;
}

int32_t sdeh_partial_sums(const float* part, int64_t n_items, int64_t n_chunks, int64_t width, float* scratch, float* out,
                          void* stream) {
  if (part == nullptr || scratch == nullptr || out == nullptr || n_items < 1 || n_chunks < 1 || width < 1 || n_items > 65535 ||
      (n_chunks + 31) / 32 > 65535)
    return fail(SDEH_ERR_INVALID, "partial_sums: bad argument");
  const int rc = launch_partial_sums(part, n_items, n_chunks, width, scratch, out, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "partial_sums: launch failed");
}

int32_t sdeh_reduce_estimators(const float* rnd, int64_t batch, float max_rnd, float* scratch, float* out, void* stream) {
  if (rnd == nullptr || out == nullptr || scratch == nullptr || batch < 1)
    return fail(SDEH_ERR_INVALID, "reduce_estimators: bad argument");
  long long nb = (batch + 255) / 256;
  if (nb > kRedBlocks) nb = kRedBlocks;
  const int rc = launch_reduce(rnd, batch, max_rnd, scratch, (int)nb, out, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "reduce_estimators: launch failed");
}

int32_t sdeh_loss_moment(const float* rnd, int64_t batch, float max_rnd, int32_t log_variance, int64_t* n_filtered, float* scratch,
                         float* out, float* grad_rnd, void* stream) {
  if (rnd == nullptr || out == nullptr || scratch == nullptr || grad_rnd == nullptr || batch < 1)
    return fail(SDEH_ERR_INVALID, "loss_moment: bad argument");
  long long nb = (batch + 255) / 256;
  if (nb > kRedBlocks) nb = kRedBlocks;
  const int rc = launch_loss_moment(rnd, batch, max_rnd, log_variance != 0, reinterpret_cast<long long*>(n_filtered), scratch, (int)nb, out,
                                    grad_rnd, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "loss_moment: launch failed");
}

int32_t sdeh_guard_restore(const uint64_t* table, int32_t n_tensors, const uint8_t* ok, int64_t* n_skipped, void* stream) {
  if (table == nullptr || ok == nullptr || n_tensors < 1) return fail(SDEH_ERR_INVALID, "guard_restore: bad argument");
  const int rc = launch_guard_restore(reinterpret_cast<const unsigned long long*>(table), n_tensors, ok, reinterpret_cast<long long*>(n_skipped),
                                      (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "guard_restore: launch failed");
}

int32_t sdeh_guard_check(float* grads, int64_t n, const float* value, float max_loss, uint8_t* ok, void* stream) {
  if (grads == nullptr || value == nullptr || ok == nullptr || n < 1) return fail(SDEH_ERR_INVALID, "guard_check: bad argument");
  const int rc = launch_guard_check(grads, n, value, max_loss, ok, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "guard_check: launch failed");
}

int32_t sdeh_importance_weights(const float* rnd, int64_t batch, const float* log_weight_max, float* weights,
                                void* stream) {
  if (rnd == nullptr || log_weight_max == nullptr || weights == nullptr || batch < 1)
    return fail(SDEH_ERR_INVALID, "importance_weights: bad argument");
  const int rc = launch_weights(rnd, batch, log_weight_max, weights, (hipStream_t)stream);
  return rc == SDEH_OK ? SDEH_OK : fail(rc, "importance_weights: launch failed");
}

}  // extern "C"
