import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import _lib
dbg = ctypes.CDLL(os.path.join(ROOT, "tools/diag/libvgpa_stamps.so"))
# point the binding at the diagnostic library
for name, args in _lib.SIGNATURES.items():
    fn = getattr(dbg, name); fn.restype = ctypes.c_int; fn.argtypes = args
_lib.lib = dbg
from vgpa_amd import engine
import bench
dev = torch.device("cuda", 0)
WANT_EDF = '--edf' in sys.argv
D, N, dt, P = 16, 2002, 0.005, 64   # the workgroup-per-step kernel runs D = 16 (D = 40: l96_wave_stamps.py)
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
for _ in range(3):
    en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, WANT_EDF, False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
dbg.vgpa_debug_l96_stamps.argtypes = [ctypes.c_void_p]
print("rc", dbg.vgpa_debug_l96_stamps(buf))
t = np.array(list(buf)[:9], dtype=np.int64)
names = ["load", "ef(/edf)", "cholesky", "zero-upper", "G+resid", "reduce|W", "dm,V", "dS+store"]
d = np.diff(t)
for n, v in zip(names, d): print(f"{n:12s} {v:8d} ticks")
print("total", t[-1] - t[0])
