"""vgpa_amd -- MI355X-native free-energy / gradient evaluation for the Variational
Gaussian Process Approximation of SDE smoothing.

Layout (mirrors the reference package names so call sites read the same):
  vgpa_amd.core        varFreeEnergy, gradL_Ab, fwd_ode, bwd_ode, likelihood, kl0, gradEsde_mS
  vgpa_amd.dynamics    sys_ornstein_uhlenbeck, sys_double_well, sys_lorenz_63,
                       sys_lorenz_63_ut, sys_lorenz_96_ut  (``energy_mode``)
  vgpa_amd.auxiliary   numerical (mytrapz, ut_approx), momGauss, optimize (SCG)
  vgpa_amd.engine      device-resident batched evaluation
  vgpa_amd.batch       batched problem API and multi-GPU sharding
The arithmetic runs in hand-written gfx950 HIP kernels (``csrc/``) reached through
the C ABI of ``include/vgpa.h``; importing this package loads ``lib/libvgpa_hip.so``
and fails if it has not been built.
"""
from . import _lib  # noqa: F401  (loads the shared library, raises if missing)

__version__ = "0.1.0"
