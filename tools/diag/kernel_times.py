"""Time each kernel of one free-energy + gradient evaluation with HIP events.
usage: kernel_times.py [problems] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import engine, _lib
import bench

dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D, N, dt = 40, 2002, 0.005
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return out, min(ts), sum(ts) / len(ts)


units = P * N
(m, S), t, ta = timed(lambda: engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True))
print(f"fwd_scan   min {t:8.3f} ms  avg {ta:8.3f}  {units * 26240 / t / 1e6:8.1f} GB/s")
(Eobs, dEobs_dm, jumpS), t, ta = timed(lambda: engine.obs_terms(prob, m, S))
print(f"obs_terms  min {t:8.3f} ms  avg {ta:8.3f}")
en, t, ta = timed(lambda: engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False))
print(f"esde       min {t:8.3f} ms  avg {ta:8.3f}  {units * 39688 / t / 1e6:8.1f} GB/s  status max {int(en['status'].max())}")
(lam, Psi), t, ta = timed(lambda: engine.bwd_scan(prob.method, A, en["dEdm"], en["dEdS"], dEobs_dm, prob.dt,
                                                  obs_flag=prob.obs_flag, jumpS=jumpS, symmetric=True))
print(f"bwd_scan   min {t:8.3f} ms  avg {ta:8.3f}  {units * 39360 / t / 1e6:8.1f} GB/s")
(gLA, gLb), t, ta = timed(lambda: engine.grad_assemble(prob, A, b, m, S, lam, Psi, en["Ef"], None))
print(f"grad       min {t:8.3f} ms  avg {ta:8.3f}  {units * 53120 / t / 1e6:8.1f} GB/s")
(l2, P2, fA, fb), t, ta = timed(lambda: engine.bwd_scan_grad(prob, A, b, m, S, en["dEdm"], en["dEdS"], dEobs_dm, en["Ef"], None,
                                                           jumpS=jumpS, symmetric=True, state_symmetric=True))
print(f"bwd+grad   min {t:8.3f} ms  avg {ta:8.3f}  {units * 66240 / t / 1e6:8.1f} GB/s  (fused)")
print("checksum", float(en["Esde_t"].sum()), float(gLA.sum()), float(lam.sum()), float((fA - gLA).abs().max()))
