"""The C-ABI library loads and exports every symbol include/vgpa.h declares;
argument validation that needs no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vgpa.h")).read()
    names = set(re.findall(r"\b(vgpa_[a-z0-9_]+)\s*\(", txt))
    names |= set(re.findall(r"VGPA_DECL_ESDE\((vgpa_[a-z0-9_]+)\)", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from vgpa_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.lib.vgpa_version() == _lib.ABI_VERSION


def test_header_constants_match_binding():
    from vgpa_amd import _lib
    txt = open(os.path.join(ROOT, "include", "vgpa.h")).read()
    defs = dict(re.findall(r"#define\s+(VGPA_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", txt))
    assert int(defs["VGPA_ABI_VERSION"]) == _lib.ABI_VERSION
    assert int(defs["VGPA_MAX_D"]) == _lib.MAX_D
    for k, v in _lib.METHODS.items():
        assert int(defs["VGPA_" + k]) == v
    for k, v in _lib.SYSTEMS.items():
        assert int(defs["VGPA_SYS_" + k.upper()]) == v
    assert int(defs["VGPA_GRAD_OUTER"]) == _lib.GRAD_OUTER
    assert int(defs["VGPA_GRAD_SIGINV_DIAG"]) == _lib.GRAD_SIGINV_DIAG
    assert int(defs["VGPA_SCAN_SYMMETRIC"]) == _lib.SCAN_SYMMETRIC


def test_bad_arguments_are_rejected_without_a_device():
    from vgpa_amd import _lib
    one = ctypes.c_void_p(8)   # never dereferenced: validation fails first
    with pytest.raises(ValueError, match="Unknown method of integration"):
        _lib.call("vgpa_fwd_scan", 7, 1, 4, 3, one, one, one, one, one, 0, 0.01, 0, one, one, None)
    with pytest.raises(ValueError, match="Unknown method of integration"):
        _lib.call("vgpa_bwd_scan", -1, 1, 4, 3, one, one, one, one, None, None, None, 0, 0.01, 0, None, None, one, one, None)
    with pytest.raises(_lib.VgpaError, match="null pointer"):
        _lib.call("vgpa_fwd_scan", 0, 1, 4, 3, None, one, one, one, one, 0, 0.01, 0, one, one, None)
    with pytest.raises(_lib.VgpaError, match="exceeds VGPA_MAX_D"):
        _lib.call("vgpa_fwd_scan", 0, 1, 4, 200, one, one, one, one, one, 0, 0.01, 0, one, one, None)
    with pytest.raises(_lib.VgpaError, match="Lorenz-63 needs D == 3"):
        _lib.call("vgpa_esde_l63", 1, 4, 5, one, one, one, one, one, 0, one, 0, one, one, one, one, one,
                  None, None, None, None)


def test_method_names_follow_the_reference():
    from vgpa_amd import engine
    assert engine.method_id("euler") == 0 and engine.method_id("Heun") == 1
    assert engine.method_id("rk2") == 2 and engine.method_id("RK4") == 3
    with pytest.raises(ValueError, match="Unknown method of integration"):
        engine.method_id("leapfrog")


def test_trapz_weights_match_oracle():
    from vgpa_amd import engine
    from oracle import vgpa_oracle as orc
    for N, idx in ((37, None), (37, [3, 4, 11, 30]), (10, [1, 2, 3, 9]), (10, [5]), (5, [0, 1])):
        assert np.array_equal(engine.trapz_weights(N, idx), orc.trapz_weights(N, idx))


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; import vgpa_amd, vgpa_amd.engine, vgpa_amd.core.varFreeEnergy, vgpa_amd.core.gradL_Ab, "
            "vgpa_amd.dynamics.sys_lorenz_96_ut; "
            "bad=[m for m in sys.modules if m.startswith('oracle')]; assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
