"""ctypes binding of libsdeh.so (the C ABI declared in include/sdeh.h).

There is deliberately NO fallback here: if the HIP library is missing or fails to load, importing the loss
classes still works (so that CPU-only tooling can introspect them) but the first call into the engine raises
``SdehLibraryError``.  Nothing in this package imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

SDEH_ABI_VERSION = 7
SDEH_NICE_MAX_COUPLING = 8
SDEH_MAX_HIDDEN = 8
SDEH_REDUCE_SCRATCH = 8192

# enums (include/sdeh.h)
LOSS_TIME_REVERSAL, LOSS_REFERENCE_SDE, LOSS_EXPONENTIAL = 0, 1, 2
CTRL_CLIPPED, CTRL_SCORE, CTRL_LERP, CTRL_LERP_TARGET, CTRL_LERP_PRIOR, CTRL_NONE = 0, 1, 2, 3, 4, 5
SDE_NONE, SDE_VP, SDE_CONST_OU = 0, 1, 2
DENS_NONE, DENS_GMM, DENS_DIAG_GAUSS, DENS_MULTI_WELL, DENS_FUNNEL = 0, 1, 2, 3, 4
DENS_EXTERNAL = 5  # the target's score is supplied per step (sdeh_simulate_fwd_steps): the NICE flow of BASELINE configs[4]
ACT_GELU_ERF, ACT_SILU, ACT_RELU, ACT_IDENTITY = 0, 1, 2, 3
FLAG_TRAIN, FLAG_ITO, FLAG_CHANGE_SDE_CTRL, FLAG_INIT_LOGP = 1, 2, 4, 8
FLAG_TERMINAL_TARGET, FLAG_TERMINAL_SECOND, FLAG_REFERENCE_CTRL = 16, 32, 64
FLAG_INFERENCE_SDE, FLAG_INFERENCE_CTRL = 128, 256
FLAG_DETACH_SCORE, FLAG_TARGET_SCORE_CONST = 512, 1024
INT_LANGEVIN, INT_CONTROLLED = 0, 1
DENS_FLAG_SHARED_SCALE = 1
DENS_FLAG_MM_OK = 2  # the mixture's logits may be evaluated in product form (engine._mixture_mm_ok)

STATUS = {0: "SDEH_OK", -1: "SDEH_ERR_INVALID", -2: "SDEH_ERR_UNSUPPORTED", -3: "SDEH_ERR_HIP", -4: "SDEH_ERR_CAPACITY"}

fp = C.c_void_p  # device pointers travel as integers


class SdehDensity(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dim", C.c_int32), ("n_components", C.c_int32), ("flags", C.c_int32),
        ("log_norm_const", C.c_float), ("p0", C.c_float), ("p1", C.c_float), ("p2", C.c_float),
        ("loc", fp), ("scale", fp), ("mixture_weights", fp),
    ]


class SdehTimeEmbed(C.Structure):
    _fields_ = [
        ("channels", C.c_int32), ("n_hidden", C.c_int32), ("dim_out", C.c_int32), ("reserved", C.c_int32),
        ("coeff", fp), ("phase", fp),
        ("hidden_w", fp * SDEH_MAX_HIDDEN), ("hidden_b", fp * SDEH_MAX_HIDDEN),
        ("out_w", fp), ("out_b", fp),
    ]


class SdehFourierMLP(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("channels", C.c_int32), ("n_hidden", C.c_int32), ("activation", C.c_int32),
        ("input_w", fp), ("input_b", fp),
        ("hidden_w", fp * SDEH_MAX_HIDDEN), ("hidden_b", fp * SDEH_MAX_HIDDEN),
        ("out_w", fp), ("out_b", fp),
        ("timestep_embed", SdehTimeEmbed),
    ]


class SdehInferenceCtrl(C.Structure):
    _fields_ = [
        ("ctrl_kind", C.c_int32), ("clip_model", C.c_float), ("clip_score", C.c_float), ("scale_score", C.c_float),
        ("base_model", SdehFourierMLP), ("score_model", SdehTimeEmbed),
    ]


class SdehProblem(C.Structure):
    _fields_ = [
        ("loss_kind", C.c_int32), ("ctrl_kind", C.c_int32), ("sde_kind", C.c_int32), ("flags", C.c_int32),
        ("clip_model", C.c_float), ("clip_score", C.c_float), ("scale_score", C.c_float), ("clip_target", C.c_float),
        ("terminal_t", C.c_float),
        ("vp_beta_min", C.c_float), ("vp_beta_max", C.c_float), ("vp_scale", C.c_float),
        ("ou_drift", C.c_float), ("ou_diff", C.c_float),
        ("exp_alpha", C.c_float), ("exp_sigma", C.c_float),
        ("base_model", SdehFourierMLP), ("score_model", SdehTimeEmbed),
        ("target", SdehDensity), ("prior", SdehDensity), ("second", SdehDensity),
        ("inference", SdehInferenceCtrl),
        ("rng_offset_dev", C.c_void_p),
    ]


class SdehNice(C.Structure):
    """distr/nice.py NiceModel: additive couplings (in_block / mid_block / out_block), scaling, logistic prior."""
    _fields_ = [
        ("dim", C.c_int32), ("n_coupling", C.c_int32), ("mid_dim", C.c_int32), ("n_mid", C.c_int32),
        ("mask_config", C.c_int32 * SDEH_NICE_MAX_COUPLING),
        ("in_w", fp * SDEH_NICE_MAX_COUPLING), ("in_b", fp * SDEH_NICE_MAX_COUPLING),
        ("mid_w", (fp * SDEH_MAX_HIDDEN) * SDEH_NICE_MAX_COUPLING), ("mid_b", (fp * SDEH_MAX_HIDDEN) * SDEH_NICE_MAX_COUPLING),
        ("out_w", fp * SDEH_NICE_MAX_COUPLING), ("out_b", fp * SDEH_NICE_MAX_COUPLING),
        ("scale", fp), ("log_norm_const", C.c_float),
    ]


class SdehPlanDesc(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("channels", C.c_int32), ("max_hidden", C.c_int32), ("max_steps", C.c_int32),
        ("max_components", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int64),
    ]


class SdehLibraryError(RuntimeError):
    """libsdeh.so is missing / not loadable / reports an ABI mismatch."""


class SdehError(RuntimeError):
    """A libsdeh call returned a negative status."""

    def __init__(self, status: int, message: str):
        self.status = status
        super().__init__(f"{STATUS.get(status, status)}: {message}")


class SdehUnsupported(SdehError, NotImplementedError):
    """The configuration is valid in the reference but not built into the HIP engine."""


# kernel-mode options of a plan (include/sdeh.h: sdeh_plan_set_option); the environment variables of the same names are the test
# override -- the binding copies them into the plan before a launch whenever they changed, the library itself never reads them on
# the launch path
PLAN_OPTIONS = ("SDEH_LEGACY", "SDEH_GENERIC_ONLY", "SDEH_WS_GROUPS", "SDEH_WS_QUAD", "SDEH_WS_VOUT", "SDEH_WS_BARRIER", "SDEH_BWD_PLANES",
                "SDEH_BWD_TILE", "SDEH_BWD_WAVES", "SDEH_BWD_V1", "SDEH_BWD_V2", "SDEH_BWD_NO_VIO", "SDEH_BWD_SCAN", "SDEH_BWD_ZREC", "SDEH_BRIDGE_TILES", "SDEH_BRIDGE_SPLIT",
                "SDEH_WIDE_CT", "SDEH_WIDE_SPLIT", "SDEH_GMM_MM", "SDEH_WS_OUT4")

# every symbol include/sdeh.h declares, with its prototype
PROTOTYPES = {
    "sdeh_abi_version": (C.c_int32, []),
    "sdeh_last_error": (C.c_char_p, []),
    "sdeh_plan_create": (C.c_int32, [C.POINTER(SdehPlanDesc), C.POINTER(C.c_void_p)]),
    "sdeh_plan_destroy": (None, [C.c_void_p]),
    "sdeh_plan_reserve": (C.c_int32, [C.c_void_p, C.c_int64]),
    "sdeh_plan_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "sdeh_plan_set_timing": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sdeh_plan_last_kernel_ms": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float)]),
    "sdeh_plan_last_kernel_name": (C.c_char_p, [C.c_void_p]),
    "sdeh_simulate_fwd": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp,
                                      C.c_uint64, C.c_uint64, C.c_int64, fp, fp, fp, C.c_void_p]),
    "sdeh_ctrl_backward": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64,
                                       C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_simulate_fwd_aux": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp,
                                          C.c_uint64, C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_simulate_fwd_aux2": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp,
                                           C.c_uint64, C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_simulate_fwd_train": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp,
                                            C.c_uint64, C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_simulate_fwd_train2": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp,
                                             C.c_uint64, C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_ctrl_backward_fused_supported": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem)]),
    "sdeh_ctrl_backward_fused_sizes": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sdeh_ctrl_backward_fused": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64,
                                             C.c_uint64, C.c_int64, fp, fp, fp, fp, C.c_int64, fp, C.c_void_p]),
    "sdeh_ctrl_backward_ex": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64,
                                          C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_bridge_div_backward": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, fp, fp, fp,
                                             fp, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_bridge_div_backward_wide_sizes": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int64),
                                                        C.POINTER(C.c_int64)]),
    "sdeh_bridge_div_backward_wide": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, fp, fp, fp, fp, fp,
                                                  C.c_int64, fp, C.c_void_p]),
    "sdeh_simulate_fwd_train2u": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64, C.c_uint64,
                                              C.c_int64, fp, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_bridge_inference_fwd_scratch_floats": (C.c_int64, [C.c_int32, C.c_int64]),
    "sdeh_bridge_inference_fwd": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64, C.c_uint64,
                                              C.c_int64, fp, fp, fp, fp, C.c_int64, C.c_void_p]),
    "sdeh_bridge_backward_fused_sizes": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_int64),
                                                     C.POINTER(C.c_int64)]),
    "sdeh_bridge_backward_fused": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64, C.c_uint64,
                                               C.c_int64, fp, fp, fp, fp, C.c_int64, fp, C.c_void_p]),
    "sdeh_ctrl_backward_fused_ex": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64, C.c_uint64,
                                                C.c_int64, fp, fp, fp, fp, fp, fp, C.c_int64, fp, C.c_void_p]),
    "sdeh_loss_moment": (C.c_int32, [fp, C.c_int64, C.c_float, C.c_int32, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_guard_check": (C.c_int32, [fp, C.c_int64, fp, C.c_float, fp, C.c_void_p]),
    "sdeh_guard_restore": (C.c_int32, [fp, C.c_int32, fp, fp, C.c_void_p]),
    "sdeh_plan_timing_entry": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.c_char_p, C.c_int32]),
    "sdeh_zrec_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "sdeh_ctrl_backward_fused_reads_zrec": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), C.c_int64]),
    "sdeh_simulate_fwd_train3": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64, C.c_uint64,
                                             C.c_int64, fp, fp, fp, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_ctrl_backward_fused_z": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, fp, C.c_int64, fp, C.c_uint64, C.c_uint64,
                                               C.c_int64, fp, fp, fp, fp, fp, fp, fp, C.c_int64, fp, C.c_void_p]),
    "sdeh_integrate": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), C.c_int32, fp, C.c_int32, fp, C.c_int32, C.c_float,
                                   fp, C.c_int64, fp, C.c_uint64, C.c_uint64, C.c_int64, fp, C.c_void_p]),
    "sdeh_sinkhorn_workspace_floats": (C.c_int64, [C.c_int64, C.c_int64]),
    "sdeh_nice_work_floats": (C.c_int64, [C.POINTER(SdehNice), C.c_int64, C.c_int32]),
    "sdeh_nice_eval": (C.c_int32, [C.POINTER(SdehNice), fp, C.c_int64, fp, fp, fp, C.c_int64, C.c_void_p]),
    "sdeh_simulate_fwd_steps": (C.c_int32, [C.c_void_p, C.POINTER(SdehProblem), fp, C.c_int32, C.c_int32, C.c_int32, fp, C.c_int64, fp,
                                            C.c_uint64, C.c_uint64, C.c_int64, fp, fp, fp, fp, fp, C.c_int64, C.c_void_p]),
    "sdeh_sinkhorn": (C.c_int32, [fp, C.c_int64, fp, C.c_int64, C.c_int32, fp, fp, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                  fp, fp, fp, fp, C.c_void_p]),
    "sdeh_sample_stats_scratch_floats": (C.c_int64, [C.c_int32]),
    "sdeh_sample_stats": (C.c_int32, [fp, C.c_int64, C.c_int32, fp, fp, fp, fp, C.c_void_p]),
    "sdeh_time_embed_param_floats": (C.c_int64, [C.POINTER(SdehTimeEmbed)]),
    "sdeh_time_embed_workspace_floats": (C.c_int64, [C.POINTER(SdehTimeEmbed), C.c_int32]),
    "sdeh_time_embed_backward": (C.c_int32, [C.POINTER(SdehTimeEmbed), C.c_int32, fp, C.c_int32, fp, C.c_float, fp, fp, C.c_void_p]),
    "sdeh_partial_sums_scratch_floats": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "sdeh_partial_sums": (C.c_int32, [fp, C.c_int64, C.c_int64, C.c_int64, fp, fp, C.c_void_p]),
    "sdeh_weight_grad": (C.c_int32, [fp, C.c_int32, fp, C.c_int32, C.c_int64, C.c_int32, C.c_int64, fp, fp, C.c_void_p]),
    "sdeh_reduce_estimators": (C.c_int32, [fp, C.c_int64, C.c_float, fp, fp, C.c_void_p]),
    "sdeh_importance_weights": (C.c_int32, [fp, C.c_int64, fp, fp, C.c_void_p]),
    "sdeh_debug_philox": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, fp, C.c_void_p]),
    "sdeh_debug_normals": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, fp, C.c_void_p]),
    "sdeh_debug_gelu": (C.c_int32, [fp, C.c_int64, fp, C.c_void_p]),
}

_LIB = None


def library_path() -> Path:
    env = os.environ.get("SDEH_LIBRARY")
    return Path(env) if env else Path(__file__).resolve().parent / "libsdeh.so"


def load():
    """Loads libsdeh.so once (after torch, so that it binds to the HIP runtime torch already loaded)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (must come first: one HIP runtime per process)

    path = library_path()
    if not path.exists():
        raise SdehLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sde_sampler_amd/csrc`).  There is no CPU fallback for the trajectory engine.")
    try:
        lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    except OSError as exc:  # pragma: no cover
        raise SdehLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise SdehLibraryError(f"{path} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.sdeh_abi_version() != SDEH_ABI_VERSION:
        raise SdehLibraryError(f"ABI mismatch: library {lib.sdeh_abi_version()} != binding {SDEH_ABI_VERSION}")
    _LIB = lib
    return lib


def check(status: int):
    if status == 0:
        return
    msg = load().sdeh_last_error().decode(errors="replace")
    if status == -2:
        raise SdehUnsupported(status, msg)
    raise SdehError(status, msg)
