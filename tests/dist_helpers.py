"""Test-side pieces for the sharding tests: an ops object that evaluates one time
window with the NumPy oracle (so that the window logic of vgpa_amd.batch can run on
the CPU), and the worker functions the gloo tests spawn."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import vgpa_oracle as orc   # noqa: E402
from vgpa_amd import batch              # noqa: E402


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))


class OracleWindowOps:
    """The oracle on one rank's rows, one problem (leading axis of length 1)."""
    device = torch.device("cpu")

    def __init__(self, system, sde, m0, S0, window, A_rows, b_rows):
        self.system, self.window = system, window
        self.sde, self.w = batch.local_window_sde(sde, window)
        self.full_sde = sde
        self.D = int(sde["D"])
        self.A, self.b = np.asarray(A_rows), np.asarray(b_rows)
        self.m0, self.S0 = np.asarray(m0, dtype=np.float64), np.asarray(S0, dtype=np.float64)
        assert self.A.shape[0] == window.rows

    def state_shapes(self):
        D = self.D
        return [(1,), (1,)] if D == 1 else [(1, D), (1, D, D)]

    def initial_state(self):
        return _t(self.m0).reshape(self.state_shapes()[0]), _t(self.S0).reshape(self.state_shapes()[1])

    def _row(self, t, matrix):
        a = t.numpy()[0]
        if self.D == 1:
            return a.reshape(1, 1) if not matrix else a.reshape(1, 1)
        return a

    def fwd(self, m_first, S_first):
        s = self.sde
        m1 = m_first.numpy().reshape(self.m0.shape) if self.D > 1 else m_first.numpy().reshape(-1)[0]
        S1 = S_first.numpy().reshape(self.S0.shape) if self.D > 1 else S_first.numpy().reshape(-1)[0]
        m, S = orc.fwd_scan(self.A, self.b, m1, S1, s["Sig"], s["dt"], s["ode_method"])
        if self.D == 1:
            m, S = m.reshape(-1), S.reshape(-1)
        return _t(m)[None], _t(S)[None]

    def _np(self, t):
        a = t.numpy()[0]
        return a.reshape(-1, 1) if self.D == 1 else a

    def energy_and_obs(self, m, S):
        mn, Sn = self._np(m), self._np(S)
        Eobs, dEobs_dm, dEobs_dS = orc.likelihood_gauss(mn, Sn, self.sde)
        # only per-step outputs are used; a single dummy segment keeps the oracle's own
        # time integrals (which this test does not read) well defined in empty windows
        out = orc.energy_mode(self.system, self.A, self.b, mn, Sn, dict(self.sde, obsX=np.array([self.window.rows])))
        Esde_t = np.asarray(out[7]["Esde_t"]).reshape(-1)
        dt = self.sde["dt"]
        Esde = dt * float(np.dot(self.w, Esde_t))
        # window shares of the integrals behind dEsde/dtheta and dEsde/dSigma
        tr_th = dt * self.w.dot(np.asarray(out[7]["dth_t"]).reshape(self.window.rows, -1))
        tr_sig = (dt * self.w.dot(np.asarray(out[7]["dSig_t"]).reshape(self.window.rows, -1))
                  if out[7]["dSig_t"] is not None else np.zeros(0))
        return dict(Eobs=float(np.asarray(Eobs).reshape(-1)[0]), dEobs_dm=dEobs_dm, dEobs_dS=dEobs_dS, Esde=Esde,
                    Ef=out[1], Edf=out[2], dEdm=out[3], dEdS=out[4], m=mn, S=Sn, tr_th=tr_th, tr_sig=tr_sig)

    def bwd(self, en, lam_last, Psi_last):
        s = self.sde
        ll = None if lam_last is None else lam_last.numpy()[0]
        pl = None if Psi_last is None else Psi_last.numpy()[0]
        lam, Psi = orc.bwd_scan(self.A, en["dEdm"], en["dEdS"], en["dEobs_dm"], en["dEobs_dS"], s["dt"],
                                s["ode_method"], lam_last=ll, Psi_last=pl)
        if self.D == 1:
            lam, Psi = lam.reshape(-1), Psi.reshape(-1)
        return _t(lam)[None], _t(Psi)[None]

    def partial_sums(self, en):
        self.n_th = en["tr_th"].size
        return _t(np.concatenate(([en["Esde"], en["Eobs"]], en["tr_th"], en["tr_sig"])))[None]

    def finish(self, sums):
        s = sums.numpy()[0]
        Esde, Eobs, tr_th, tr_sig = s[0], s[1], s[2:2 + self.n_th], s[2 + self.n_th:]
        KL0 = float(np.asarray(orc.kl0_gauss(self.m0, self.S0, self.full_sde)).reshape(-1)[0])
        sde = self.full_sde
        if self.D == 1:                                   # sys_ornstein_uhlenbeck.py:153-156, sys_double_well.py:181-184
            SigInv = 1.0 / float(np.asarray(sde["Sig"]).reshape(-1)[0])
            dth = (1.0 if self.system == "ou" else 4.0) * SigInv * tr_th
            dSig = np.array([-SigInv * Esde])
        else:                                             # sys_lorenz_96_ut.py:250-256
            SigInv = np.linalg.inv(sde["Sig"])
            dth = np.diag(SigInv) * tr_th
            dSig = -0.5 * SigInv.dot(np.diag(tr_sig)).dot(SigInv)
        return dict(F=torch.tensor([KL0 + Eobs + Esde], dtype=torch.float64), Esde=_t([Esde]), Eobs=_t([Eobs]),
                    dth=_t(dth)[None], dSig=_t(dSig)[None])

    def grad(self, en, m, S, lam, Psi):
        s = self.sde
        g = orc.grad_L_Ab(self.A, self.b, en["m"], en["S"], self._np(lam), self._np(Psi), en["Ef"], en["Edf"],
                          s["Sig"], s["dt"]).reshape(-1)
        rows, D = self.window.rows, self.D
        if D == 1:
            return _t(g[:rows])[None], _t(g[rows:])[None]
        return _t(g[: rows * D * D].reshape(rows, D, D))[None], _t(g[rows * D * D:].reshape(rows, D))[None]


def window_inputs(A, b, win):
    sl = win.local_slice()
    return A[sl], b[sl]


def run_time_sharded(system, sde, m0, S0, A, b, comm):
    win = batch.time_window(int(sde["N"]), comm.rank, comm.world)
    Aw, bw = window_inputs(A, b, win)
    ops = OracleWindowOps(system, sde, m0, S0, win, Aw, bw)
    return batch.free_energy_time_sharded([ops], comm)[0]


# -- workers for torch.multiprocessing.spawn (gloo, CPU) ---------------------
def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def worker_time_sharded(rank, world, port, fixture, system, method, outdir, comm_class="DistComm"):
    dist = _init(rank, world, port)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import load_golden, sde_from_golden, initial_moments
        g = load_golden(fixture + ".npz")
        sde = sde_from_golden(g, method)
        m0, S0 = initial_moments(g)
        A, b = orc.unpack_Ab(g["Ab"], int(g["N"]), int(g["D"]))
        res = run_time_sharded(system, sde, m0, S0, A, b, getattr(batch, comm_class)())
        np.savez(os.path.join(outdir, f"r{rank}.npz"), F=res.F.numpy(), gLA=res.gLA.numpy(), gLb=res.gLb.numpy(),
                 m=res.m.numpy(), lam=res.lam.numpy(), t0=res.window.t0, t1=res.window.t1,
                 dth=res.dth.numpy(), dSig=res.dSig.numpy())
    finally:
        dist.destroy_process_group()


def worker_problem_shard(rank, world, port, P_total, outdir, comm_class="DistComm"):
    dist = _init(rank, world, port)
    try:
        comm = getattr(batch, comm_class)()

        def evaluate(lo, hi):
            # a stand-in evaluation whose value identifies the problem
            F = torch.tensor([1000.0 + 3.0 * p for p in range(lo, hi)], dtype=torch.float64)
            return F, dict(count=hi - lo)

        F_all, extras, (lo, hi) = batch.evaluate_problem_shard(evaluate, P_total, comm)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), F=F_all.numpy(), lo=lo, hi=hi, count=extras["count"])
    finally:
        dist.destroy_process_group()


def worker_device_windows(rank, world, port, fixture, method, micro, outdir):
    """Two or more PROCESSES that share cuda:0: DeviceWindowOps (the HIP kernels) on each rank's window,
    messages staged through the host over gloo."""
    dist = _init(rank, world, port)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import load_golden, sde_from_golden, initial_moments
        from vgpa_amd import engine, _lib
        torch.cuda.set_device(0)
        g = load_golden(fixture + ".npz")
        sde = sde_from_golden(g, method)
        m0, S0 = initial_moments(g)
        N, D, P = int(g["N"]), int(g["D"]), 4
        An, bn = orc.unpack_Ab(g["Ab"], N, D)
        rng = np.random.default_rng(3)                     # the same problems on every rank
        A = engine.to_dev(An[None] + 0.05 * rng.standard_normal((P, N, D, D)))
        b = engine.to_dev(bn[None] + 0.5 * rng.standard_normal((P, N, D)))
        win = batch.time_window(N, rank, world)
        sl = win.local_slice()
        ops = []
        for c in range(micro):
            lo, hi = batch.shard_bounds(P, c, micro)
            ops.append(batch.DeviceWindowOps(_lib.SYS_L96_UT, sde, m0, S0, win, A[lo:hi, sl].contiguous(),
                                             b[lo:hi, sl].contiguous(), P=hi - lo))
        res = batch.free_energy_time_sharded(ops, batch.HostStagedComm())
        cat = lambda k, dim=0: torch.cat([getattr(r, k) for r in res], dim=dim).cpu().numpy()   # noqa: E731
        np.savez(os.path.join(outdir, f"r{rank}.npz"), F=cat("F"), gLA=cat("gLA"), gLb=cat("gLb"), dth=cat("dth"),
                 dSig=cat("dSig"), t0=win.t0, t1=win.t1)
    finally:
        dist.destroy_process_group()
