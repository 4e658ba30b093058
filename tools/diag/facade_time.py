"""Where the time of one varFreeEnergy + gradL_Ab pair on host arrays goes (Lorenz-96 D=40, one problem)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vgpa_amd import engine, _lib
from vgpa_amd.core import _facade
from vgpa_amd.core.varFreeEnergy import varFreeEnergy, _publish
from vgpa_amd.core.gradL_Ab import gradL_Ab
from vgpa_amd.dynamics import sys_lorenz_96_ut
import bench

dev = torch.device("cuda", 0)
D, N = 40, 2002
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, 0.005, 1, 0, dev)
Ab = np.concatenate([A[0].reshape(-1).cpu().numpy(), b[0].reshape(-1).cpu().numpy()])
mParam = dict(m0=m0, S0=S0, fun=sys_lorenz_96_ut.energy_mode, sde_struct=sde)


def tm(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


prob, t = tm(lambda: engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=1)); print(f"make_problem      {t:7.3f} ms")
Ah, bh = _facade.unpack_Ab(Ab, N, D)
(Ad, bd), t = tm(lambda: (_facade.dev_batch1(Ah, D, "m"), _facade.dev_batch1(bh, D, "v"))); print(f"upload A, b       {t:7.3f} ms")
st, t = tm(lambda: engine.free_energy(prob, Ad, bd)); print(f"free_energy       {t:7.3f} ms")
_, t = tm(lambda: _publish(dict(), st, D)); print(f"publish (6 D2H)   {t:7.3f} ms")
_, t = tm(lambda: float(st.F[0].item())); print(f"F.item            {t:7.3f} ms")
_, t = tm(lambda: varFreeEnergy(Ab, mParam)); print(f"varFreeEnergy     {t:7.3f} ms")
_, t = tm(lambda: gradL_Ab(Ab, mParam)); print(f"gradL_Ab          {t:7.3f} ms")
_, t = tm(lambda: gradL_Ab(Ab, mParam, True)); print(f"gradL_Ab evalF    {t:7.3f} ms")
