// One translation unit per compiled state dimension:
//   hipcc -DSDEH_DP=<n> -DSDEH_PAD=<0|1> [-DSDEH_C=<channels>] -c sdeh_traj_inst.hip
// The trajectory kernel keeps x[d] in registers, so d is a compile-time constant.  PAD=0 variants require
// d == DP; PAD=1 variants accept any d <= DP (coordinates >= d are held at zero).  The dispatcher
// (sdeh_api.hip) picks the exact variant when one was compiled, else the smallest padded one.
#include "sdeh_traj.hpp"

#ifndef SDEH_DP
#error "compile with -DSDEH_DP=<state dimension>"
#endif
#ifndef SDEH_PAD
#define SDEH_PAD 0
#endif
#ifndef SDEH_C
#define SDEH_C 64
#endif

#define SDEH_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define SDEH_CAT(a, b, c, d, e, f) SDEH_CAT2(a, b, c, d, e, f)

namespace sdeh {
int SDEH_CAT(launch_traj_dp, SDEH_DP, _c, SDEH_C, _p, SDEH_PAD)(const TrajArgs& a, hipStream_t stream) {
  return launch_traj<SDEH_DP, SDEH_C, (SDEH_PAD != 0)>(a, stream);
}
}  // namespace sdeh
