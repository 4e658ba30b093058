"""vgpa_amd.compat: the reference's module names resolve to this package (no GPU
needed: nothing is evaluated)."""
import os
import sys

import pytest

REF = "/root/reference"


def _cleanup(saved):
    import vgpa_amd.compat as compat
    compat.uninstall()
    for k in [k for k in sys.modules if k.split(".")[0] in ("core", "dynamics", "auxiliary")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_aliases_resolve_to_this_package():
    import vgpa_amd.compat as compat
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("core", "dynamics", "auxiliary")}
    try:
        names = compat.install()
        assert len(names) == 17
        import core.fwd_ode
        import dynamics.sys_double_well as dw
        from auxiliary.momGauss import momGauss
        from core.gradEsde_mS import gradEsde_mS
        assert core.fwd_ode.__name__ == "vgpa_amd.core.fwd_ode"
        assert dw.energy_mode.__module__ == "vgpa_amd.dynamics.sys_double_well"
        assert momGauss.__module__ == "vgpa_amd.auxiliary.momGauss"
        assert gradEsde_mS.__module__ == "vgpa_amd.core.gradEsde_mS"
    finally:
        _cleanup(saved)
    assert "core.fwd_ode" not in sys.modules or sys.modules["core.fwd_ode"].__name__ != "vgpa_amd.core.fwd_ode"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")
def test_reference_smoothing_binds_to_this_package():
    """core/smoothing.py of the reference, imported unmodified after install(), picks up
    this package's varFreeEnergy, gradL_Ab and optim_SCG through its own import lines."""
    import vgpa_amd.compat as compat
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("core", "dynamics", "auxiliary")}
    path = list(sys.path)
    try:
        compat.install(reference_root=REF)
        from core import smoothing
        assert smoothing.__file__.startswith(REF)
        assert smoothing.varFreeEnergy.__module__ == "vgpa_amd.core.varFreeEnergy"
        assert smoothing.gradL_Ab.__module__ == "vgpa_amd.core.gradL_Ab"
        assert smoothing.optim_SCG.__module__ == "vgpa_amd.auxiliary.optimize"
    finally:
        _cleanup(saved)
        sys.path[:] = path
