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


KINK_TAGS = ["conv0", "conv2", "conv4"] + [f"{k}{l}" for l in range(4) for k in ("ang", "spa")] + ["up"]


def kink_compare(g, tag: str, positive: np.ndarray):
    """Compare an implementation's branch decisions (bool array `positive`, flat, in the reference's layout of pre-activation
    `tag`) with a train_kink_*.npz fixture `g`.  Returns (agree_elsewhere, flips): agree_elsewhere says that every unit NOT
    listed as near-zero took the reference's branch (SHA-256 of the masked sign bitmap; on a mismatch the popcount table
    names the first differing chunk in the assertion message); flips = [(flat index, reference z)] of the listed near-zero
    units whose decision differs from the reference's."""
    import hashlib
    near = g[f"kink_{tag}_near_idx"].astype(np.int64)
    z = g[f"kink_{tag}_near_z"]
    bits = positive.astype(bool).copy()
    mine_near = bits[near]
    bits[near] = False
    ok = hashlib.sha256(np.packbits(bits).tobytes()).digest() == g[f"kink_{tag}_sha256"].tobytes()
    where = ""
    if not ok:
        chunk = int(g["kink_chunk"])
        pad = (-bits.size) % chunk
        pc = np.concatenate([bits, np.zeros(pad, dtype=bool)]).reshape(-1, chunk).sum(1)
        bad = np.nonzero(pc != g[f"kink_{tag}_popcount"].astype(np.int64))[0]
        where = f"{tag}: {bad.size} of {pc.size} chunks of {chunk} units differ in popcount, first at chunk {int(bad[0]) if bad.size else -1}"
    flips = [(int(i), float(v)) for i, v, m in zip(near, z, mine_near) if bool(m) != (v > 0)]
    return ok, where, flips
