"""Per-output error of the Lorenz-96 D=40 energy kernel against the oracle (diagnostic).
Run with VGPA_L96_WAVE=1 to select the wavefront-per-step kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402,F401
from oracle import vgpa_oracle as orc  # noqa: E402
from vgpa_amd.dynamics import sys_lorenz_96_ut  # noqa: E402

D, N = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 9
rng = np.random.default_rng(300 + D)
A = 6.0 * np.eye(D)[None] + rng.standard_normal((N, D, D))
b = 4.0 * rng.standard_normal((N, D))
m = 3.0 * rng.standard_normal((N, D))
X = rng.standard_normal((N, D, D))
S = 0.05 * np.eye(D)[None] + np.einsum("nij,nkj->nik", X, X) / D
sig = 1.0 + rng.random(D)
sde = dict(N=N, D=D, dt=0.01, Sig=np.diag(sig), theta=7.5, obsX=np.array([2, 3, 7]))
out = sys_lorenz_96_ut.energy_mode(A, b, m, S, sde)
ref = orc.energy_mode("l96_ut", A, b, m, S, sde)
names = ["Esde", "Ef", "Edf", "dEsde_dm", "dEsde_dS", "dEsde_dth", "dEsde_dSig"]
for n, v, r in zip(names, out, ref[:7]):
    v, r = np.asarray(v), np.asarray(r)
    print(f"{n:12s} relerr {np.max(np.abs(v - r)) / max(np.max(np.abs(r)), 1e-300):.3e}  nan={int(np.isnan(v).sum())}")
print("dEsde_dS bitwise symmetric:", np.array_equal(out[4], np.swapaxes(out[4], 1, 2)))
