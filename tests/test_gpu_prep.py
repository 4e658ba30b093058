"""Input preparation on the device, through the C-ABI (vgpa_sde_path, vgpa_init_ab0) and the
modules with the reference's signatures, against vectors recorded from the reference
(tests/golden/prep.npz).  Bit-exact: seeded ``numpy.random`` draws in the reference's order,
products and sums rounded separately in the reference's order."""
import contextlib
import io

import numpy as np
import pytest

from conftest import load_golden
from oracle import vgpa_oracle as orc

pytestmark = pytest.mark.gpu


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


def test_system_paths_reproduce_the_reference_bit_for_bit():
    from vgpa_amd.dynamics import (sys_double_well, sys_lorenz_63, sys_lorenz_63_ut, sys_lorenz_96_ut,
                                   sys_ornstein_uhlenbeck)
    g = load_golden("prep.npz")
    T = g["ou_T"]
    with _quiet():
        np.random.seed(101)
        assert np.array_equal(sys_ornstein_uhlenbeck.system_path(T, float(g["ou_sig"]), float(g["ou_theta"])), g["ou_X"])
        np.random.seed(102)
        state = np.random.get_state()[1].copy()
        assert np.array_equal(sys_double_well.system_path(T, float(g["dw_sig"]), float(g["dw_theta"])), g["dw_X"])
        assert np.array_equal(np.random.get_state()[1], state)      # the caller's generator is restored
        np.random.seed(103)
        assert np.array_equal(sys_lorenz_63.system_path(T, g["l63_sig"], g["l63_theta"]), g["l63_X"])
        np.random.seed(104)
        assert np.array_equal(sys_lorenz_63_ut.system_path(T, g["l63_sig"], g["l63_theta"]), g["l63_ut_X"])
        np.random.seed(105)
        assert np.array_equal(sys_lorenz_96_ut.system_path(T, 40, g["l96_sig"], 8.0), g["l96_X"])
        np.random.seed(106)
        assert np.array_equal(sys_lorenz_96_ut.system_path(T, 6, 2.5, 8.0), g["l96s_X"])
    with pytest.raises(ValueError):
        sys_lorenz_63_ut.system_path(T, np.ones((3, 3)), g["l63_theta"])      # not diagonal


def test_batched_paths_vs_oracle():
    """P paths with their own drift parameter, noise scale and random numbers; D = 70 spans two
    components per lane."""
    import torch
    from vgpa_amd import engine, _lib
    rng = np.random.default_rng(5)
    for D, P, N in ((40, 5, 30), (70, 3, 12)):
        thetas = 6.0 + 3.0 * rng.random((P, 1))
        kd = 0.1 + 0.2 * rng.random((P, D))
        x0 = thetas + 1e-3 * rng.standard_normal((P, D))
        ek = rng.standard_normal((P, N, D))
        d = engine.to_dev
        X = engine.sde_paths(_lib.SYS_L96_UT, d(x0), d(thetas), d(kd), d(ek), 0.01, burn=300, dtau=1e-3).cpu().numpy()
        for p in range(P):
            ref = orc.sde_path("l96_ut", x0[p], float(thetas[p, 0]), kd[p], ek[p], 0.01, burn=300, dtau=1e-3)
            assert np.array_equal(X[p], ref), (D, p)
    assert torch.cuda.is_available()


def test_starting_values_reproduce_the_reference():
    from vgpa_amd import engine
    from vgpa_amd.auxiliary.initialize_Ab0 import initialize_Ab0, initialize_Ab0_batched
    g = load_golden("prep.npz")
    T = g["obs_T"]
    sde = dict(Sig=g["l63_sig"], N=len(T), D=3, Tw=T, obsX=g["obs1_X"], obsY=g["obs1_Y"], dt=0.01)
    assert np.array_equal(initialize_Ab0(g["ab0_S0"], sde), g["ab0_vec"])
    sde2 = dict(sde, Sig=np.diag([1.0, 2.0, 3.0]))
    A0, b0 = initialize_Ab0_batched(g["ab0_S0"], [sde, sde2])
    assert A0.shape == (2, len(T), 3, 3) and b0.shape == (2, len(T), 3)
    for p, s in enumerate((sde, sde2)):
        vec = np.concatenate((A0[p].cpu().numpy().ravel(), b0[p].cpu().numpy().ravel()))[:, None]
        assert np.array_equal(vec, initialize_Ab0(g["ab0_S0"], s))
    assert np.array_equal(vec[: 3 * 3, 0], np.diag(0.5 * (np.array([1.0, 2.0, 3.0]) / np.array([0.25, 0.5, 0.75]))).ravel())
    assert engine.device().type == "cuda"


def test_bad_arguments_are_rejected():
    from vgpa_amd import engine, _lib
    from vgpa_amd._lib import VgpaError
    d = engine.to_dev
    ek = d(np.zeros((1, 4, 3)))
    with pytest.raises((VgpaError, ValueError)):
        engine.sde_paths(_lib.SYS_OU, d(np.zeros((1, 3))), d(np.ones(1)), d(np.ones(3)), ek, 0.01)      # OU is 1-D
    with pytest.raises((VgpaError, ValueError)):
        engine.sde_paths(99, d(np.zeros((1, 3))), d(np.ones(1)), d(np.ones(3)), ek, 0.01)
