"""A fixed slice of the randomised parity sweep (tests/fuzz_parity.py): 600 configurations drawn from
seeded generators -- all five systems, four integrators, D from 1 to 60, N from 1 to 300, 1 to 170
problems (both sides of the thresholds at which the library changes kernels), dense and diagonal
operators, observation indices that touch each other and the ends -- each compared with the oracle
at 1e-9 on F, the gradient and every intermediate."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first_seed", [1, 5001])
def test_random_configurations_vs_oracle(first_seed):
    import fuzz_parity
    n, skipped, failures, worst = fuzz_parity.sweep(first_seed, count=300, verbose=False)
    assert n == 300 and skipped < 30
    assert not failures, "\n".join(failures[:10])
    assert worst <= fuzz_parity.TOL


def test_random_energy_steps_with_cholesky_fallback_vs_oracle():
    """energy_mode on covariances a third of which are not positive definite, at every kernel family of
    the unscented energies (Lorenz-63; Lorenz-96 generic D, D = 16, D = 40)."""
    import fuzz_parity
    failures, worst = fuzz_parity.sweep_energy(1, 120, verbose=False)
    assert not failures, "\n".join(failures[:10])
    assert worst <= fuzz_parity.TOL


def test_random_time_window_cuts_vs_unsharded_device():
    """Random configurations cut into 2 to 5 time windows (ranks as threads on the one GPU), 1 to 3
    micro-batches of problems: F, the hyper-gradients, the gradients and the scan outputs against the
    one-call device evaluation.  Window edges fall on, before and after observation rows; windows of a
    single row occur."""
    import numpy as np
    import torch
    import fuzz_parity
    from test_gpu_windows import _run_windows
    from vgpa_amd import engine
    done = 0
    for seed in range(7001, 7400):
        rng = np.random.default_rng(seed)
        cfg = fuzz_parity.draw(rng)
        if cfg["P"] > 17:
            continue
        world = int(rng.integers(2, 6))
        if cfg["N"] < world:
            continue
        system, D, P, N, sde = cfg["system"], cfg["D"], cfg["P"], cfg["N"], cfg["sde"]
        micro = int(rng.integers(1, min(3, P) + 1))
        sid = fuzz_parity.SYS[system]
        shape_A = (P, N) if D == 1 else (P, N, D, D)
        shape_b = (P, N) if D == 1 else (P, N, D)
        A, b = engine.to_dev(cfg["A"].reshape(shape_A)), engine.to_dev(cfg["b"].reshape(shape_b))
        # every problem takes the drift parameters of sde_struct here (the problems differ in A, b)
        prob = engine.make_problem(sid, sde, cfg["m0"], cfg["S0"], P=P)
        st = engine.free_energy(prob, A, b, want_edf=True, want_hyper=True)
        gLA, gLb = engine.gradient(prob, A, b, st)
        out = _run_windows(sid, sde, cfg["m0"], cfg["S0"], A, b, world, micro=micro)
        tag = f"seed {seed}: {system} {cfg['method']} D={D} N={N} P={P} world={world} micro={micro}"
        rel = fuzz_parity.relerr
        for r in range(world):
            F = torch.cat([res.F for res in out[r]])
            assert rel(F.cpu().numpy(), st.F.cpu().numpy()) <= 1e-11, tag
            if out[r][0].dth is not None and st.dth is not None:
                assert rel(torch.cat([res.dth for res in out[r]]).cpu().numpy(), st.dth.cpu().numpy()) <= 1e-10, tag
                assert rel(torch.cat([res.dSig for res in out[r]]).cpu().numpy(), st.dSig.cpu().numpy()) <= 1e-10, tag
        cat = lambda k: torch.cat([torch.cat([getattr(res, k) for res in out[r]], dim=0) for r in range(world)], dim=1)   # noqa: E731
        assert rel(cat("gLA").cpu().numpy(), gLA.cpu().numpy()) <= 1e-11, tag
        assert rel(cat("gLb").cpu().numpy(), gLb.cpu().numpy()) <= 1e-11, tag
        assert rel(cat("m").cpu().numpy(), st.m.cpu().numpy()) <= 1e-12, tag
        assert rel(cat("lam").cpu().numpy(), st.lam.cpu().numpy()) <= 1e-11, tag
        done += 1
        if done == 60:
            break
    assert done == 60
