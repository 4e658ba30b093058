"""Sharding over ranks, on the CPU: index arithmetic, the problem-sharded gather and
the time-window hand-offs of vgpa_amd.batch.  The kernels are replaced by the NumPy
oracle (tests/dist_helpers.py), the transport is gloo with two processes, or threads
for larger worlds; the result must equal the unsharded oracle evaluation."""
import socket
import threading

import numpy as np
import pytest
import torch

from conftest import load_golden, sde_from_golden, initial_moments, relerr
from oracle import vgpa_oracle as orc
from vgpa_amd import batch
import dist_helpers as dh

TOL = 1e-11   # the window sums are re-associated; everything else is the same arithmetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 5, 16, 2002):
        for world in (1, 2, 3, 8):
            got = [batch.shard_bounds(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            for (_, hi), (lo, _) in zip(got, got[1:]):
                assert hi == lo
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        batch.shard_bounds(4, 2, 2)


def test_time_window_halo_and_rows():
    wins = [batch.time_window(10, r, 3) for r in range(3)]
    assert [(w.t0, w.t1) for w in wins] == [(0, 4), (4, 7), (7, 10)]
    assert [w.halo for w in wins] == [1, 1, 0]
    assert [w.rows for w in wins] == [5, 4, 3]
    assert wins[1].local_slice() == slice(4, 8)
    with pytest.raises(ValueError):
        batch.time_window(2, 0, 3)


def test_local_window_sde_keeps_global_segmentation():
    g = load_golden("sys_l63.npz")
    sde = sde_from_golden(g, "EULER")
    N = int(sde["N"])
    w_global = orc.trapz_weights(N, sde["obsX"])
    tot, seen = 0.0, 0
    for r in range(3):
        win = batch.time_window(N, r, 3)
        loc, w = batch.local_window_sde(sde, win)
        assert loc["N"] == win.rows and w.shape == (win.rows,)
        assert np.array_equal(w[: win.owned], w_global[win.t0:win.t1])
        assert np.all(w[win.owned:] == 0)
        assert np.all((loc["obsX"] >= 0) & (loc["obsX"] < win.owned))
        assert loc["obsY"].shape[0] == loc["obsX"].size
        seen += loc["obsX"].size
        tot += w.sum()
    assert seen == np.asarray(sde["obsX"]).size
    assert tot == w_global.sum()


def _threads_time_sharded(system, fixture, method, world):
    g = load_golden(fixture + ".npz")
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    A, b = orc.unpack_Ab(g["Ab"], int(g["N"]), int(g["D"]))
    comms = batch.ThreadComm.world_of(world)
    out, errs = [None] * world, []

    def run(r):
        try:
            out[r] = dh.run_time_sharded(system, sde, m0, S0, A, b, comms[r])
        except Exception as e:   # pragma: no cover
            errs.append(e)
            comms[r]._sh["barrier"].abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    ref = orc.evaluate(system, g["Ab"], m0, S0, sde)
    return out, ref, int(g["N"]), int(g["D"])


def _check_against_unsharded(parts, ref, N, D):
    gLA = np.concatenate([np.asarray(p["gLA"])[0] for p in parts])
    gLb = np.concatenate([np.asarray(p["gLb"])[0] for p in parts])
    grad = np.concatenate((gLA.reshape(-1), gLb.reshape(-1)))
    for p in parts:
        assert relerr(float(np.asarray(p["F"]).reshape(-1)[0]), ref["F"]) < TOL
        # the hyper-parameter gradients travel in the same all-reduce as the energies
        assert relerr(np.asarray(p["dth"]).reshape(-1), np.asarray(ref["dEsde_dth"]).reshape(-1)) < TOL
        assert relerr(np.asarray(p["dSig"]).reshape(-1), np.asarray(ref["dEsde_dSig"]).reshape(-1)) < TOL
    assert relerr(grad, ref["grad"].reshape(-1)) < TOL
    m = np.concatenate([np.asarray(p["m"])[0] for p in parts])
    lam = np.concatenate([np.asarray(p["lam"])[0] for p in parts])
    assert relerr(m.reshape(-1), ref["m"].reshape(-1)) < TOL
    assert relerr(lam.reshape(-1), ref["lam"].reshape(-1)) < TOL


@pytest.mark.parametrize("method", ["EULER", "HEUN", "RK2", "RK4"])
@pytest.mark.parametrize("system,fixture,world", [("l63", "sys_l63", 3), ("ou", "sys_ou", 2), ("l96_ut", "sys_l96_ut_d6", 4)])
def test_time_windows_in_threads_equal_unsharded(system, fixture, world, method):
    out, ref, N, D = _threads_time_sharded(system, fixture, method, world)
    parts = [dict(F=o.F.numpy(), gLA=o.gLA.numpy(), gLb=o.gLb.numpy(), m=o.m.numpy(), lam=o.lam.numpy(),
                  dth=o.dth.numpy(), dSig=o.dSig.numpy()) for o in out]
    _check_against_unsharded(parts, ref, N, D)


@pytest.mark.parametrize("method", ["EULER", "RK4"])
def test_time_windows_gloo_world2(tmp_path, method):
    import torch.multiprocessing as mp
    mp.spawn(dh.worker_time_sharded, args=(2, _free_port(), "sys_l63_ut", "l63_ut", method, str(tmp_path)),
             nprocs=2, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
    g = load_golden("sys_l63_ut.npz")
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    ref = orc.evaluate("l63_ut", g["Ab"], m0, S0, sde)
    assert int(parts[0]["t1"]) == int(parts[1]["t0"])
    _check_against_unsharded(parts, ref, int(g["N"]), int(g["D"]))


def test_problem_shard_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    P_total = 5
    mp.spawn(dh.worker_problem_shard, args=(2, _free_port(), P_total, str(tmp_path)), nprocs=2, join=True)
    want = np.array([1000.0 + 3.0 * p for p in range(P_total)])
    bounds = []
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["F"], want)          # every rank sees the whole grid, in problem order
        assert int(z["count"]) == int(z["hi"]) - int(z["lo"])
        bounds.append((int(z["lo"]), int(z["hi"])))
    assert bounds == [(0, 3), (3, 5)]


def test_host_staged_communicator_gloo_world2(tmp_path):
    """batch.HostStagedComm (device tensors through host copies; here the tensors are host tensors
    already) carries the same messages as DistComm: time windows and the gather of F."""
    import torch.multiprocessing as mp
    (tmp_path / "w").mkdir()
    (tmp_path / "p").mkdir()
    mp.spawn(dh.worker_time_sharded, args=(2, _free_port(), "sys_l63_ut", "l63_ut", "HEUN", str(tmp_path / "w"),
                                           "HostStagedComm"), nprocs=2, join=True)
    parts = [np.load(tmp_path / "w" / f"r{r}.npz") for r in range(2)]
    g = load_golden("sys_l63_ut.npz")
    sde = sde_from_golden(g, "HEUN")
    m0, S0 = initial_moments(g)
    ref = orc.evaluate("l63_ut", g["Ab"], m0, S0, sde)
    _check_against_unsharded(parts, ref, int(g["N"]), int(g["D"]))
    mp.spawn(dh.worker_problem_shard, args=(2, _free_port(), 5, str(tmp_path / "p"), "HostStagedComm"), nprocs=2, join=True)
    want = np.array([1000.0 + 3.0 * p for p in range(5)])
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / "p" / f"r{r}.npz")["F"], want)


def test_gather_problem_values_single_rank_is_identity():
    x = torch.arange(4, dtype=torch.float64)
    assert batch.gather_problem_values(x, 4, batch.SingleComm()) is x
