import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def relerr(x, y):
    """max-norm relative error max|x-y| / max|y| (entries cross zero, so no
    element-wise ratio)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    assert x.size == y.size, (x.shape, y.shape)
    x = x.reshape(y.shape)
    den = np.max(np.abs(y)) if y.size else 0.0
    num = np.max(np.abs(x - y)) if y.size else 0.0
    return num / den if den > 0 else num


def sde_from_golden(g, method="EULER"):
    """Rebuild the reference's ``sde_struct`` dictionary from a fixture."""
    D = int(g["D"])
    scal = (lambda a: float(a)) if D == 1 else (lambda a: np.array(a))
    theta = g["theta"]
    theta = float(theta) if theta.ndim == 0 else np.array(theta)
    mu0 = float(g["mu0"]) if g["mu0"].ndim == 0 else np.array(g["mu0"])
    return dict(Sig=scal(g["Sig"]), theta=theta, Rig=scal(g["Rig"]), D=D, N=int(g["N"]),
                H=scal(g["H"]), obsX=np.array(g["obsX"]), obsY=np.array(g["obsY"]),
                px0=dict(mu0=mu0, tau0=scal(g["tau0"])), dt=float(g["dt"]),
                ode_method=method, checkGradf=False)


def initial_moments(g):
    D = int(g["D"])
    m0 = np.array(g["m0"])
    S0 = float(g["S0"]) if D == 1 else np.array(g["S0"])
    return m0, S0


@pytest.fixture(scope="session")
def golden():
    return load_golden
