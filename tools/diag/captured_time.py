"""One evaluation eagerly and as a replayed HIP graph (engine.CapturedEvaluation), wall time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vgpa_amd import engine, _lib

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)


def case(name, system, D, N, dt, theta, Sig):
    M = 40
    obsX = np.linspace(0, N, M + 2, dtype=int)[1:-1]
    if D == 1:
        sde = dict(Sig=Sig, theta=theta, Rig=0.04, D=1, N=N, H=1.0, obsX=obsX, obsY=rng.standard_normal((M, 1)),
                   px0=dict(mu0=0.0, tau0=0.5), dt=dt, ode_method="EULER")
        m0, S0 = np.array([0.1]), 0.25
        A = torch.full((1, N), 0.5 * Sig / 0.25, device=dev, dtype=torch.float64)
        b = torch.randn((1, N), device=dev, dtype=torch.float64)
    else:
        sde = dict(Sig=Sig * np.eye(D), theta=theta, Rig=2.0 * np.eye(D), D=D, N=N, H=np.eye(D), obsX=obsX,
                   obsY=rng.standard_normal((M, D)), px0=dict(mu0=np.zeros(D), tau0=0.5 * np.eye(D)), dt=dt, ode_method="EULER")
        m0, S0 = rng.standard_normal(D), 0.25 * np.eye(D)
        A = 0.5 * Sig / 0.25 * torch.eye(D, device=dev, dtype=torch.float64) + 0.1 * torch.randn((1, N, D, D), device=dev, dtype=torch.float64)
        b = 5.0 * torch.randn((1, N, D), device=dev, dtype=torch.float64)
    prob = engine.make_problem(system, sde, m0, S0, P=1)
    cap = engine.CapturedEvaluation(prob, A, b)

    def wall(fn, reps=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    te = wall(lambda: engine.free_energy_and_gradient(prob, A, b))
    tg = wall(cap.replay)
    print(f"{name:22s} N={N:5d}: eager {te:7.3f} ms, graph replay {tg:7.3f} ms per evaluation")


case("double-well", _lib.SYS_DW, 1, 5001, 0.01, 1.0, 0.8)
case("Ornstein-Uhlenbeck", _lib.SYS_OU, 1, 1001, 0.01, 2.0, 1.0)
case("Lorenz-63 unscented", _lib.SYS_L63_UT, 3, 2002, 0.01, (10.0, 28.0, 8.0 / 3.0), 10.0)
case("Lorenz-96 D=40", _lib.SYS_L96_UT, 40, 2002, 0.005, 8.0, 4.0)
