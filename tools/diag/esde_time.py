"""HIP-event time of the Lorenz-96 D=40 energy kernel alone, plus its error against the oracle on a
small case.  usage: esde_time.py [problems] [reps]   (VGPA_L96_VARIANT selects an experimental build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vgpa_amd import engine, _lib
import bench
from oracle import vgpa_oracle as orc
from vgpa_amd.dynamics import sys_lorenz_96_ut

dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
D, N, dt = 40, 2002, 0.005
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
ts = []
for _ in range(reps + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = ts[1:]
print(f"variant {os.environ.get('VGPA_L96_VARIANT', '0')}: esde min {min(ts):.3f} ms avg {sum(ts) / len(ts):.3f} ms "
      f"frac {P * N * 39688 / min(ts) / 1e6 / 8000:.3f}  status {int(en['status'].max())}")
# accuracy on the oracle test case
rng = np.random.default_rng(300 + D)
n = 9
A1 = 6.0 * np.eye(D)[None] + rng.standard_normal((n, D, D))
b1 = 4.0 * rng.standard_normal((n, D))
m1 = 3.0 * rng.standard_normal((n, D))
X = rng.standard_normal((n, D, D))
S1 = 0.05 * np.eye(D)[None] + np.einsum("nij,nkj->nik", X, X) / D
sig = 1.0 + rng.random(D)
sde1 = dict(N=n, D=D, dt=0.01, Sig=np.diag(sig), theta=7.5, obsX=np.array([2, 3, 7]))
ref = orc.energy_mode("l96_ut", A1, b1, m1, S1, sde1)
d = engine.to_dev
en1 = engine.esde(_lib.SYS_L96_UT, d(A1[None]), d(b1[None]), d(m1[None]), d(S1[None]), d(1.0 / sig), d(np.array([7.5])),
                  False, False)
got = dict(Esde_t=en1["Esde_t"][0].cpu().numpy(), Ef=en1["Ef"][0].cpu().numpy(), dEdm=en1["dEdm"][0].cpu().numpy(),
           dEdS=en1["dEdS"][0].cpu().numpy())
want = dict(Esde_t=ref[7]["Esde_t"], Ef=ref[1], dEdm=ref[3], dEdS=ref[4])
errs = [f"{k} {np.max(np.abs(got[k] - want[k])) / np.max(np.abs(want[k])):.1e}" for k in got]
print("  errors:", " ".join(errs), " symmetric:", np.array_equal(got["dEdS"], np.swapaxes(got["dEdS"], 1, 2)),
      " status", int(en1["status"].max()))
