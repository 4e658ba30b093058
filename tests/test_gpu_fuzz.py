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
