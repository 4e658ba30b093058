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
D, N, dt, P = 40, 2002, 0.005, int(sys.argv[1]) if len(sys.argv) > 1 else 512
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
for _ in range(2):
    m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
dbg.vgpa_debug_m40_stamps.argtypes = [ctypes.c_void_p]
print("rc", dbg.vgpa_debug_m40_stamps(buf))
allv = np.array(list(buf), dtype=np.int64).reshape(8, 16)[:4]
base = allv[allv > 0].min()
names = ["matrix 0", "matrix 1", "matrix 2", "memory"]
# forward scan, steps t = 1000 and t = 1001 of workgroup 7, in clock cycles.  Stamp 0: after the first barrier of the step,
# 1: the work between the barriers done, 2: after the second barrier, 3: end of the step's work,
# 4: the matrix products done (matrix wavefronts) / A(t+1) committed to LDS (memory wavefront)
for w in range(4):
    print(f"{names[w]:9s} t=1000:", [int(x - base) if x > 0 else -1 for x in allv[w, 0:5]], " t=1001:", [int(x - base) if x > 0 else -1 for x in allv[w, 8:13]])
