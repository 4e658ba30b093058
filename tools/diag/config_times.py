"""Wall time of one free-energy + gradient evaluation for the configurations other than the
headline one (Lorenz-63 grid, double-well, Ornstein-Uhlenbeck, Lorenz-96 with one problem and
with the higher-order integrators), per kernel with HIP events.  The scans also print their
algorithmic HBM rate (bytes every unit must move / time) as a fraction of the 8 TB/s peak."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vgpa_amd import engine, _lib

dev = torch.device("cuda", 0)


def run(name, system, D, N, dt, P, theta, Sig, method="EULER"):
    rng = np.random.default_rng(0)
    M = 40
    obsX = np.linspace(0, N, M + 2, dtype=int)[1:-1]
    if D == 1:
        sde = dict(Sig=Sig, theta=theta, Rig=0.04, D=1, N=N, H=1.0, obsX=obsX, obsY=rng.standard_normal((M, 1)),
                   px0=dict(mu0=0.0, tau0=0.5), dt=dt, ode_method=method)
        m0, S0 = np.array([0.1]), 0.25
        A = torch.full((P, N), 0.5 * Sig / 0.25, device=dev, dtype=torch.float64) + 0.1 * torch.randn((P, N), device=dev, dtype=torch.float64)
        b = torch.randn((P, N), device=dev, dtype=torch.float64)
        th = None
    else:
        sde = dict(Sig=Sig * np.eye(D), theta=theta, Rig=2.0 * np.eye(D), D=D, N=N, H=np.eye(D), obsX=obsX,
                   obsY=rng.standard_normal((M, D)), px0=dict(mu0=np.zeros(D), tau0=0.5 * np.eye(D)), dt=dt, ode_method=method)
        m0, S0 = rng.standard_normal(D), 0.25 * np.eye(D)
        A = 0.5 * Sig / 0.25 * torch.eye(D, device=dev, dtype=torch.float64) + 0.1 * torch.randn((P, N, D, D), device=dev, dtype=torch.float64)
        b = 5.0 * torch.randn((P, N, D), device=dev, dtype=torch.float64)
        th = np.tile(np.asarray(theta, dtype=float)[None, :], (P, 1)) * (1.0 + 0.05 * rng.standard_normal((P, len(theta))))
    prob = engine.make_problem(system, sde, m0, S0, P=P, theta=th)

    def ev(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        return out, e0.elapsed_time(e1)

    (m, S), t_f = ev(lambda: engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=prob.sym_fwd))
    (Eobs, dm, jS), t_o = ev(lambda: engine.obs_terms(prob, m, S))
    en, t_e = ev(lambda: engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, D == 1, True))
    sym = prob.sym_jump and engine.esde_output_is_symmetric(prob.system, D)
    (lam, Psi), t_b = ev(lambda: engine.bwd_scan(prob.method, A, en["dEdm"], en["dEdS"], dm, prob.dt, obs_flag=prob.obs_flag, jumpS=jS, symmetric=sym))
    (gA, gb), t_g = ev(lambda: engine.grad_assemble(prob, A, b, m, S, lam, Psi, en["Ef"], en["Edf"]))
    st, t_all = ev(lambda: engine.gradient(prob, A, b, engine.free_energy(prob, A, b, want_hyper=True, want_edf=D == 1)))
    units = P * N
    # forward: A, b in, S, m out; backward: A, dEdS, dEdm, jump-m in, Psi, lam out
    f_f = units * 8 * (2 * D * D + 2 * D) / (t_f * 1e-3) / 8e12
    f_b = units * 8 * (3 * D * D + 3 * D) / (t_b * 1e-3) / 8e12
    print(f"{name:28s} P={P:5d} N={N:5d}: fwd {t_f:7.3f} ({f_f:5.3f} of HBM peak) obs {t_o:6.3f} esde {t_e:7.3f} "
          f"bwd {t_b:7.3f} ({f_b:5.3f}) grad {t_g:7.3f} | "
          f"F+grad {t_all:8.3f} ms = {units / t_all / 1e3:8.2f} M units/s; status max {int(en['status'].max())}")


for P in (512, 4096):
    run("L63 analytic grid", _lib.SYS_L63, 3, 2002, 0.01, P, (10.0, 28.0, 8.0 / 3.0), 10.0)
    run("L63 unscented grid", _lib.SYS_L63_UT, 3, 2002, 0.01, P, (10.0, 28.0, 8.0 / 3.0), 10.0)
for P in (1, 4096):
    run("double-well", _lib.SYS_DW, 1, 5001, 0.01, P, 1.0, 0.8)
    run("Ornstein-Uhlenbeck", _lib.SYS_OU, 1, 1001, 0.01, P, 2.0, 1.0)
run("L63 unscented RK4", _lib.SYS_L63_UT, 3, 2002, 0.01, 512, (10.0, 28.0, 8.0 / 3.0), 10.0, "RK4")


# Lorenz-96, D = 40: one problem (the reference's own run) and the higher-order integrators
def run_l96(P, N, method):
    import bench
    A, b, sde, m0, S0, thetas = bench.synth_problem(40, N, 0.005, P, 0, dev)
    sde["ode_method"] = method
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)

    def one():
        return engine.free_energy_and_gradient(prob, A, b)
    one(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        one()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print(f"L96 D=40 {method:5s} device-resident  P={P:5d} N={N:5d}: F+grad {t:8.3f} ms = {P * N / t / 1e3:8.2f} M units/s")
    return A, b, sde, m0, S0


for method in ("EULER", "HEUN", "RK2", "RK4"):
    run_l96(512, 2002, method)
A, b, sde, m0, S0 = run_l96(1, 2002, "EULER")

# the same single problem through the reference's call signatures: host arrays in, host arrays out
from vgpa_amd.core.varFreeEnergy import varFreeEnergy
from vgpa_amd.core.gradL_Ab import gradL_Ab
from vgpa_amd.dynamics import sys_lorenz_96_ut
Ab = np.concatenate([A[0].reshape(-1).cpu().numpy(), b[0].reshape(-1).cpu().numpy()])
mParam = dict(m0=m0, S0=S0, fun=sys_lorenz_96_ut.energy_mode, sde_struct=sde)
varFreeEnergy(Ab, mParam); gradL_Ab(Ab, mParam)
t0 = time.perf_counter()
for _ in range(3):
    F = varFreeEnergy(Ab, mParam)
    g = gradL_Ab(Ab, mParam)
t = (time.perf_counter() - t0) / 3 * 1e3
print(f"L96 D=40 EULER varFreeEnergy + gradL_Ab on host arrays (PCIe both ways) P=    1 N= 2002: {t:8.3f} ms = "
      f"{2002 / t / 1e3:8.3f} M units/s")
