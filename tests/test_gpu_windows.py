"""Time-window sharding with the HIP kernels: several windows rehearsed as threads on
one GPU (vgpa_amd.batch.ThreadComm) must reproduce the unsharded device evaluation
and the golden vectors; the terminal-state input of the backward scan is checked on
its own against the oracle."""
import threading

import numpy as np
import pytest

from conftest import load_golden, relerr, sde_from_golden, initial_moments
from oracle import vgpa_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-9
SYSTEMS = dict(ou=("SYS_OU", "sys_ou.npz"), dw=("SYS_DW", "sys_dw.npz"), l63=("SYS_L63", "sys_l63.npz"),
               l63_ut=("SYS_L63_UT", "sys_l63_ut.npz"), l96_d6=("SYS_L96_UT", "sys_l96_ut_d6.npz"),
               l96_d40=("SYS_L96_UT", "sys_l96_ut_d40.npz"))


def _run_windows(system_id, sde, m0, S0, A, b, world, micro=1):
    """A, b: device tensors (P,N,...).  Returns per-rank lists of WindowResult."""
    from vgpa_amd import batch
    P = A.shape[0]
    comms = batch.ThreadComm.world_of(world)
    out, errs = [None] * world, []

    def run(r):
        try:
            win = batch.time_window(int(sde["N"]), r, world)
            sl = win.local_slice()
            ops = []
            for c in range(micro):
                lo, hi = batch.shard_bounds(P, c, micro)
                ops.append(batch.DeviceWindowOps(system_id, sde, m0, S0, win, A[lo:hi, sl], b[lo:hi, sl], P=hi - lo))
            out[r] = batch.free_energy_time_sharded(ops, comms[r])
        except Exception as e:   # pragma: no cover
            errs.append(e)
            comms[r]._sh["barrier"].abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    return out


@pytest.mark.parametrize("method", ["EULER", "HEUN", "RK2", "RK4"])
@pytest.mark.parametrize("name,world", [("ou", 2), ("dw", 3), ("l63", 3), ("l63_ut", 2), ("l96_d6", 4), ("l96_d40", 2)])
def test_time_windows_on_device_vs_golden(name, world, method):
    import torch
    from vgpa_amd import engine, _lib
    sysname, fname = SYSTEMS[name]
    g = load_golden(fname)
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    N, D = int(g["N"]), int(g["D"])
    An, bn = orc.unpack_Ab(g["Ab"], N, D)
    A = engine.to_dev(An.reshape((1, N) if D == 1 else (1, N, D, D)))
    b = engine.to_dev(bn.reshape((1, N) if D == 1 else (1, N, D)))
    out = _run_windows(getattr(_lib, sysname), sde, m0, S0, A, b, world)
    Fref = float(g[f"{method}.F"])
    for r in range(world):
        assert abs(float(out[r][0].F.item()) - Fref) <= TOL * abs(Fref)
        # the hyper-parameter gradients are part of the same all-reduce (every rank has them)
        assert relerr(out[r][0].dth.cpu().numpy().reshape(-1), np.asarray(g[f"{method}.dEsde_dth"]).reshape(-1)) <= TOL
        assert relerr(out[r][0].dSig.cpu().numpy().reshape(-1), np.asarray(g[f"{method}.dEsde_dSig"]).reshape(-1)) <= TOL
    gLA = torch.cat([out[r][0].gLA for r in range(world)], dim=1).cpu().numpy()
    gLb = torch.cat([out[r][0].gLb for r in range(world)], dim=1).cpu().numpy()
    grad = np.concatenate((gLA.reshape(-1), gLb.reshape(-1)))
    assert relerr(grad, g[f"{method}.grad"].reshape(-1)) <= TOL
    m = torch.cat([out[r][0].m for r in range(world)], dim=1).cpu().numpy()
    lam = torch.cat([out[r][0].lam for r in range(world)], dim=1).cpu().numpy()
    if f"{method}.m" in g:   # the large fixture keeps intermediates for one method only
        assert relerr(m.reshape(-1), g[f"{method}.m"].reshape(-1)) <= TOL
        assert relerr(lam.reshape(-1), g[f"{method}.lam"].reshape(-1)) <= TOL


def test_time_windows_batched_microbatches_match_unsharded_device():
    """Four L96 problems, three windows, two micro-batches: F and gradients equal the
    one-call device evaluation to rounding (only the energy sum is re-associated)."""
    import torch
    from vgpa_amd import engine, _lib
    g = load_golden("sys_l96_ut_d40.npz")
    sde = sde_from_golden(g, "EULER")
    m0, S0 = initial_moments(g)
    N, D, P = int(g["N"]), 40, 4
    An, bn = orc.unpack_Ab(g["Ab"], N, D)
    rng = np.random.default_rng(3)
    A = engine.to_dev(An[None] + 0.05 * rng.standard_normal((P, N, D, D)))
    b = engine.to_dev(bn[None] + 0.5 * rng.standard_normal((P, N, D)))
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P)
    st = engine.free_energy(prob, A, b, want_edf=False, want_hyper=True)
    gLA, gLb = engine.gradient(prob, A, b, st)
    out = _run_windows(_lib.SYS_L96_UT, sde, m0, S0, A, b, world=3, micro=2)
    for r in range(3):
        F = torch.cat([res.F for res in out[r]])
        assert relerr(F.cpu().numpy(), st.F.cpu().numpy()) <= 1e-12
        dth = torch.cat([res.dth for res in out[r]])
        dSig = torch.cat([res.dSig for res in out[r]])
        assert relerr(dth.cpu().numpy(), st.dth.cpu().numpy()) <= 1e-12
        assert relerr(dSig.cpu().numpy(), st.dSig.cpu().numpy()) <= 1e-12
    got_A = torch.cat([torch.cat([res.gLA for res in out[r]], dim=0) for r in range(3)], dim=1)
    got_b = torch.cat([torch.cat([res.gLb for res in out[r]], dim=0) for r in range(3)], dim=1)
    assert relerr(got_A.cpu().numpy(), gLA.cpu().numpy()) <= 1e-12
    assert relerr(got_b.cpu().numpy(), gLb.cpu().numpy()) <= 1e-12


@pytest.mark.parametrize("D", [1, 3, 16, 40])
@pytest.mark.parametrize("method", ["EULER", "RK4"])
def test_bwd_scan_terminal_state_vs_oracle(D, method):
    from vgpa_amd import engine
    rng = np.random.default_rng(7 + D)
    N, dt = 19, 0.01
    if D == 1:
        A = 2.0 + 0.5 * rng.standard_normal((N, 1))
        gm, gS, jm, jS = (rng.standard_normal((N, 1)) for _ in range(4))
        lamT, PsiT = rng.standard_normal(1), rng.standard_normal(1)
        shp_v = shp_m = (1, N)
    else:
        A = 4.0 * np.eye(D)[None] + 0.5 * rng.standard_normal((N, D, D))
        gm, gS = rng.standard_normal((N, D)), rng.standard_normal((N, D, D))
        jm, jS = rng.standard_normal((N, D)), rng.standard_normal((N, D, D))
        lamT, PsiT = rng.standard_normal(D), rng.standard_normal((D, D))
        shp_v, shp_m = (1, N, D), (1, N, D, D)
    lr, Pr = orc.bwd_scan(A, gm, gS, jm, jS, dt, method, lam_last=lamT, Psi_last=PsiT)
    lam, Psi = engine.bwd_scan(engine.method_id(method), engine.to_dev(A.reshape(shp_m)), engine.to_dev(gm.reshape(shp_v)),
                               engine.to_dev(gS.reshape(shp_m)), engine.to_dev(jm.reshape(shp_v)), dt,
                               dEobs_dS=engine.to_dev(jS.reshape(shp_m)),
                               lam_last=engine.to_dev(lamT.reshape((1,) + lamT.shape if D > 1 else (1,))),
                               Psi_last=engine.to_dev(PsiT.reshape((1,) + PsiT.shape if D > 1 else (1,))))
    assert relerr(lam.cpu().numpy().reshape(-1), lr.reshape(-1)) <= TOL
    assert relerr(Psi.cpu().numpy().reshape(-1), Pr.reshape(-1)) <= TOL


def test_time_windows_across_processes_sharing_the_gpu(tmp_path):
    """Two processes, one window each, both on cuda:0: the device ops of every rank with a real
    torch.distributed process group between them (gloo, messages staged through the host by
    batch.HostStagedComm) against the one-call device evaluation in this process."""
    import socket
    import torch
    import torch.multiprocessing as mp
    import dist_helpers
    from vgpa_amd import engine, _lib
    g = load_golden("sys_l96_ut_d40.npz")
    sde = sde_from_golden(g, "EULER")
    m0, S0 = initial_moments(g)
    N, D, P, world = int(g["N"]), 40, 4, 2
    An, bn = orc.unpack_Ab(g["Ab"], N, D)
    rng = np.random.default_rng(3)
    A = engine.to_dev(An[None] + 0.05 * rng.standard_normal((P, N, D, D)))
    b = engine.to_dev(bn[None] + 0.5 * rng.standard_normal((P, N, D)))
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P)
    st = engine.free_energy(prob, A, b, want_edf=False, want_hyper=True)
    gLA, gLb = engine.gradient(prob, A, b, st)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(dist_helpers.worker_device_windows, args=(world, port, "sys_l96_ut_d40", "EULER", 2, str(tmp_path)),
             nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in range(world):
        assert relerr(parts[r]["F"], st.F.cpu().numpy()) <= 1e-12
        assert relerr(parts[r]["dth"], st.dth.cpu().numpy()) <= 1e-12
        assert relerr(parts[r]["dSig"], st.dSig.cpu().numpy()) <= 1e-12
    assert [int(parts[r]["t0"]) for r in range(world)] == [0, int(parts[0]["t1"])]
    assert relerr(np.concatenate([parts[r]["gLA"] for r in range(world)], axis=1), gLA.cpu().numpy()) <= 1e-12
    assert relerr(np.concatenate([parts[r]["gLb"] for r in range(world)], axis=1), gLb.cpu().numpy()) <= 1e-12
