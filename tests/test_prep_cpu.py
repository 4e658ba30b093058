"""Input preparation (sample paths, observations, starting values) on the CPU: the oracle's
restatements and the host-side modules against vectors recorded from the reference
(tests/golden/prep.npz, written by oracle/make_golden.py).  Everything here is bit-exact:
the same random numbers in the same order, the same operations in the same order."""
import numpy as np

from conftest import load_golden
from oracle import vgpa_oracle as orc


def _kdiag(sig, dt, D):
    K = np.sqrt(np.asarray(sig, dtype=np.float64) * dt)
    return np.full(D, float(K)) if K.ndim == 0 else np.array(np.diagonal(K))


def test_oracle_paths_reproduce_the_reference_bit_for_bit():
    g = load_golden("prep.npz")
    T = g["ou_T"]
    dt, N = np.diff(T)[0], len(T)
    np.random.seed(101)
    X = orc.sde_path("ou", np.zeros(1), float(g["ou_theta"]), _kdiag(g["ou_sig"], dt, 1), np.random.randn(N, 1), dt)
    assert np.array_equal(X, g["ou_X"])
    # double well: its own generator seeded with 6771 (sys_double_well.py:52-68)
    sig, th = float(g["dw_sig"]), float(g["dw_theta"])
    np.random.seed(6771)
    sign = +1.0 if np.random.rand(1) > 0.5 else -1.0
    x0 = sign * th + np.sqrt(0.5 * sig * dt) * np.random.randn(1)
    X = orc.sde_path("dw", x0, th, _kdiag(sig, dt, 1), np.random.randn(N, 1), dt)
    assert np.array_equal(X, g["dw_X"])
    np.random.seed(103)
    add0 = np.random.standard_normal(3)
    X = orc.sde_path("l63", np.ones(3), g["l63_theta"], _kdiag(g["l63_sig"], dt, 3), np.random.randn(N, 3), dt,
                     burn=8000, dtau=1e-3, add0=add0)
    assert np.array_equal(X, g["l63_X"])
    np.random.seed(104)
    X = orc.sde_path("l63_ut", np.ones(3), g["l63_theta"], _kdiag(g["l63_sig"], dt, 3), np.random.randn(N, 3), dt,
                     burn=8000, dtau=1e-3)
    assert np.array_equal(X, g["l63_ut_X"])
    for seed, key, D, sig in ((105, "l96_X", 40, g["l96_sig"]), (106, "l96s_X", 6, 2.5)):
        x0 = np.full(D, 8.0)
        x0[D // 2] += 1e-3
        np.random.seed(seed)
        X = orc.sde_path("l96_ut", x0, 8.0, _kdiag(sig, dt, D), np.random.randn(N, D), dt, burn=50000, dtau=1e-3)
        assert np.array_equal(X, g[key]), key


def test_collect_obs_reproduces_the_reference():
    from vgpa_amd.auxiliary.collect_obs import collect_obs
    g = load_golden("prep.npz")
    np.random.seed(108)
    oX, oY = collect_obs(g["obs_path"], g["obs_T"], 4, 2.0)
    assert np.array_equal(oX, g["obs1_X"]) and np.array_equal(oY, g["obs1_Y"])
    np.random.seed(109)
    oX, oY = collect_obs(g["obs_path"], g["obs_T"], 2, np.array([1.0, 2.0]), [0, 2])
    assert np.array_equal(oX, g["obs2_X"]) and np.array_equal(oY, g["obs2_Y"])
    import pytest
    with pytest.raises(ValueError):
        collect_obs(g["obs_path"], g["obs_T"], 1000, 2.0)


def test_starting_values_host_parts_and_oracle():
    from vgpa_amd.auxiliary.initialize_Ab0 import _spline_means, initialize_Ab0
    g = load_golden("prep.npz")
    T = g["obs_T"]
    sde = dict(Sig=g["l63_sig"], N=len(T), D=3, Tw=T, obsX=g["obs1_X"], obsY=g["obs1_Y"], dt=0.01)
    A0, b0 = orc.init_ab0(_spline_means(sde), np.diagonal(g["l63_sig"]), np.diagonal(g["ab0_S0"]), 0.01)
    vec = np.concatenate((A0.ravel(), b0.ravel()))[:, None]
    assert np.array_equal(vec, g["ab0_vec"])
    # D = 1 is a host formula in the package too
    sde1 = dict(Sig=0.8, N=len(T), D=1, Tw=T, obsX=g["obs1_X"], obsY=g["obs1_Y"][:, :1], dt=0.01)
    assert np.array_equal(initialize_Ab0(0.3, sde1), g["ab0_1d_vec"])
