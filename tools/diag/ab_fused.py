"""A/B timing of library builds in ONE process: the forward scan, the energy kernel and the fused backward scan +
gradient of the benchmark problem are launched with every library in turn, several rounds, and the median of each
is reported (separate processes on the same box differ by more than most changes do).
usage: ab_fused.py lib1.so lib2.so ...   (paths; 'main' = the built library)"""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import _lib
import bench
from vgpa_amd import engine

paths = [(_lib.LIB_PATH if a == "main" else os.path.abspath(a)) for a in sys.argv[1:]] or [_lib.LIB_PATH]
libs = []
for pth in paths:
    L = ctypes.CDLL(pth)
    for name, args in _lib.SIGNATURES.items():
        fn = getattr(L, name); fn.restype = ctypes.c_int; fn.argtypes = args
    L.vgpa_last_error_string.restype = ctypes.c_char_p
    libs.append(L)
dev = torch.device("cuda", 0)
D, N, dt, P = 40, 2002, 0.005, 512
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
Eobs, dEobs_dm, jumpS = engine.obs_terms(prob, m, S)
en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
out = (torch.empty_like(A), torch.empty_like(b))


def one(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


calls = {
    "forward": lambda: engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True),
    "energy": lambda: engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False),
    "backward+gradient": lambda: engine.bwd_scan_grad(prob, A, b, m, S, en["dEdm"], en["dEdS"], dEobs_dm, en["Ef"], None,
                                                      jumpS=jumpS, symmetric=True, state_symmetric=True, out=out),
}
times = {(i, k): [] for i in range(len(libs)) for k in calls}
for rnd in range(6):
    for i, L in enumerate(libs):
        _lib.lib = L
        for k, fn in calls.items():
            for rep in range(4):
                t = one(fn)
                if rnd > 0:
                    times[(i, k)].append(t)
for i, pth in enumerate(paths):
    print(os.path.basename(pth).ljust(28), "  ".join(f"{k} {statistics.median(times[(i, k)]):7.3f} (min {min(times[(i, k)]):7.3f})" for k in calls), flush=True)
