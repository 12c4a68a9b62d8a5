"""Helpers shared by tools/gen_golden.py and tests/: which activation elements a fixture keeps.

TEST INFRASTRUCTURE (see oracle/lft_oracle.py header)."""
import numpy as np

N_SUB = 2048


def sub_indices(size: int) -> np.ndarray:
    """Seeded, sorted sample of flat indices into an activation of ``size`` elements."""
    if size <= N_SUB:
        return np.arange(size)
    rng = np.random.Generator(np.random.PCG64([7, size]))
    return np.sort(rng.choice(size, N_SUB, replace=False))


def stats(a: np.ndarray) -> np.ndarray:
    d = a.astype(np.float64).ravel()
    return np.array([d.size, d.sum(), np.abs(d).sum(), (d * d).sum()])
