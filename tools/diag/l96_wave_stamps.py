"""Per-phase cycle stamps of the wavefront-per-step Lorenz-96 kernel (needs libvgpa_stamps.so,
see README.md; run with VGPA_L96_WAVE=1)."""
import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import _lib
dbg = ctypes.CDLL(os.path.join(ROOT, "tools/diag/libvgpa_stamps.so"))
for name, args in _lib.SIGNATURES.items():
    fn = getattr(dbg, name); fn.restype = ctypes.c_int; fn.argtypes = args
_lib.lib = dbg
from vgpa_amd import engine
import bench
dev = torch.device("cuda", 0)
D, N, dt, P = 40, 2002, 0.005, int(sys.argv[1]) if len(sys.argv) > 1 else 64
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
for _ in range(3):
    en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
dbg.vgpa_debug_l96_wave_stamps.argtypes = [ctypes.c_void_p]
print("rc", dbg.vgpa_debug_l96_wave_stamps(buf))
t = np.array([buf[i] for i in (16, 17, 18, 19, 21, 22, 23, 25, 26, 31)], dtype=np.int64)
names = ["loads", "cholesky", "ef,base", "G+residuals", "reduce", "W", "dS+dm mfma", "dS store", "stores land"]
for n, v in zip(names, np.diff(t)): print(f"{n:12s} {v:8d} ticks")
print("total", t[-1] - t[0])
c = np.array([buf[17], buf[27], buf[28], buf[29], buf[30], buf[18]], dtype=np.int64)
for n, v in zip(["diag0", "panel0+trail0", "diag1", "panel1+trail1", "diag2"], np.diff(c)): print(f"  chol {n:14s} {v:8d}")
