"""Lorenz-63 scans and whole evaluation at several batch sizes (HIP events).
usage: small_scan_times.py [P ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vgpa_amd import engine, _lib

dev = torch.device("cuda", 0)
D, N, dt = 3, 2002, 0.01
Ps = [int(a) for a in sys.argv[1:]] or [512, 4096]


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return out, min(ts)


for P in Ps:
    gen = torch.Generator(device=dev).manual_seed(1)
    A = 0.3 * torch.randn((P, N, D, D), generator=gen, device=dev, dtype=torch.float64) + 20.0 * torch.eye(D, device=dev, dtype=torch.float64)
    b = 20.0 * torch.randn((P, N, D), generator=gen, device=dev, dtype=torch.float64)
    rng = np.random.default_rng(1)
    M = int(np.floor((N - 1) * dt * 5))
    obsX = np.linspace(0, N, M + 2, dtype=int)[1:-1]
    mu0 = np.array([1.0, -1.0, 20.0])
    sde = dict(Sig=10.0 * np.eye(D), theta=np.array([10.0, 28.0, 8.0 / 3.0]), Rig=2.0 * np.eye(D), D=D, N=N, H=np.eye(D),
               obsX=obsX, obsY=mu0[None] + 2.0 * rng.standard_normal((M, D)), px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)),
               dt=dt, ode_method=os.environ.get("METHOD", "EULER"))
    for name, sid in (("l63", _lib.SYS_L63), ("l63_ut", _lib.SYS_L63_UT)):
        prob = engine.make_problem(sid, sde, mu0 + 0.1, 0.25 * np.eye(D), P=P)
        (m, S), tf = timed(lambda: engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=prob.sym_fwd))
        Eobs, dEobs_dm, jumpS = engine.obs_terms(prob, m, S)
        en, te = timed(lambda: engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False))
        sym = prob.sym_jump and engine.esde_output_is_symmetric(prob.system, D)
        (lam, Psi), tb = timed(lambda: engine.bwd_scan(prob.method, A, en["dEdm"], en["dEdS"], dEobs_dm, prob.dt,
                                                       obs_flag=prob.obs_flag, jumpS=jumpS, symmetric=sym))
        _, tall = timed(lambda: engine.free_energy_and_gradient(prob, A, b))
        print(f"P={P:5d} {name:7s} {sde['ode_method']:5s} fwd {tf:7.3f} ms  esde {te:7.3f}  bwd {tb:7.3f} (sym {int(sym)})  F+grad {tall:7.3f} ms", flush=True)
