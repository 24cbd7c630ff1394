"""MI355X-native Euler-Maruyama trajectory engine behind sde_sampler's loss plugin surface.

Public entry points
  sde_sampler_amd.losses.oc.{TimeReversalLoss, ReferenceSDELoss, ExponentialIntegratorSDELoss}
      drop-in replacements for sde_sampler.losses.oc.* (Hydra `_target_` swap)
  sde_sampler_amd.eq.integrator.EulerIntegrator
      drop-in replacement for sde_sampler.eq.integrator.EulerIntegrator (LangevinSDE / OU / ControlledSDE)
  sde_sampler_amd.problems.build / baseline_spec
      plain-data problem construction (stand-in for the Hydra config tree)
  include/sdeh.h + sde_sampler_amd/libsdeh.so
      the C ABI underneath (ctypes binding: sde_sampler_amd/_lib.py)
"""
from ._lib import SdehError, SdehLibraryError, SdehUnsupported  # noqa: F401

__version__ = "0.1.0"
