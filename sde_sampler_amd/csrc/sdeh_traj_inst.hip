// One translation unit per compiled trajectory-kernel variant:
//   hipcc -DSDEH_DP=<n> -DSDEH_PAD=<0|1> [-DSDEH_SPEC="loss,ctrl,target,gmm,act,refc" -DSDEH_GNV=<n> -DSDEH_SPECNAME=<tag>] -c sdeh_traj_inst.hip
// The kernels keep x[d] in registers, so d is a compile-time constant.  PAD=0 variants require d == DP; PAD=1
// variants accept any d <= DP (coordinates >= d are held at zero).  SDEH_SPEC additionally fixes the loss / control /
// target / GMM-table / activation kinds (see sdeh_variants.inc); without it they stay run-time switches.
// Every variant gets the wave-specialised kernel; the generic ones also carry the single-wave ("legacy") kernel,
// which is the fallback for mixtures whose tables do not fit in LDS.
#include "sdeh_bwd.hpp"
#include "sdeh_integrate.hpp"
#include "sdeh_sinkhorn.hpp"
#include "sdeh_bridge.hpp"

#ifndef SDEH_DP
#error "compile with -DSDEH_DP=<state dimension>"
#endif
#ifndef SDEH_PAD
#define SDEH_PAD 0
#endif
#ifndef SDEH_SPEC
#define SDEH_SPEC -1, -1, -1, -1, -1, -1
#define SDEH_SPECNAME g
#define SDEH_GENERIC 1
#endif
#ifndef SDEH_GENERIC
#define SDEH_GENERIC 0
#endif
#ifndef SDEH_GNV
#define SDEH_GNV -1
#endif

#define SDEH_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define SDEH_CAT(a, b, c, d, e, f) SDEH_CAT2(a, b, c, d, e, f)

namespace sdeh {
int SDEH_CAT(launch_ws_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const TrajArgs& a, hipStream_t stream) {
  return launch_traj_ws<SDEH_DP, 64, (SDEH_PAD != 0), SDEH_SPEC, SDEH_GNV>(a, stream);
}
int SDEH_CAT(launch_legacy_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const TrajArgs& a, hipStream_t stream) {
#if SDEH_GENERIC
  return launch_traj<SDEH_DP, 64, (SDEH_PAD != 0), SDEH_SPEC>(a, stream);
#else
  (void)a; (void)stream;
  return SDEH_ERR_UNSUPPORTED;
#endif
}
int SDEH_CAT(launch_bwd_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const BwdArgs& a, hipStream_t stream) {
#if SDEH_GENERIC
  return launch_ctrl_bwd<SDEH_DP, 64, (SDEH_PAD != 0)>(a, stream);
#else
  (void)a; (void)stream;
  return SDEH_ERR_UNSUPPORTED;
#endif
}
int SDEH_CAT(launch_int_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const TrajArgs& a, hipStream_t stream) {
#if SDEH_GENERIC
  return launch_integrate<SDEH_DP, 64, (SDEH_PAD != 0)>(a, stream);
#else
  (void)a; (void)stream;
  return SDEH_ERR_UNSUPPORTED;
#endif
}
int SDEH_CAT(launch_bridge_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const TrajArgs& a, hipStream_t stream) {
#if SDEH_GENERIC
  return launch_bridge<SDEH_DP, 64, (SDEH_PAD != 0)>(a, stream);
#else
  (void)a; (void)stream;
  return SDEH_ERR_UNSUPPORTED;
#endif
}
int SDEH_CAT(launch_bridge_bwd_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const BridgeBwdArgs& a, hipStream_t stream) {
#if SDEH_GENERIC
  return launch_bridge_div_bwd<SDEH_DP, 64, (SDEH_PAD != 0)>(a, stream);
#else
  (void)a; (void)stream;
  return SDEH_ERR_UNSUPPORTED;
#endif
}
int SDEH_CAT(launch_sink_dp, SDEH_DP, _p, SDEH_PAD, _, SDEH_SPECNAME)(const SinkArgs& a, int mode, int splits, hipStream_t stream) {
#if SDEH_GENERIC
  return launch_sink<SDEH_DP, (SDEH_PAD != 0)>(a, mode, splits, stream);
#else
  (void)a; (void)mode; (void)splits; (void)stream;
  return SDEH_ERR_UNSUPPORTED;
#endif
}
}  // namespace sdeh
