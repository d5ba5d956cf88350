"""TEST INFRASTRUCTURE (parity oracle; never imported by deepof_amd).

Numpy restatement of the noise stream behind ``dof_step_begin`` (include/deepof_hip.h): Philox-4x32-10
(Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the generator behind
``torch.randn`` on accelerators, which the reference draws its reparameterisation noise
(/root/reference/deepof/clustering/models_new.py:1741) and Monte-Carlo samples (losses.py:536) from) with
key = seed, counter = (element // 4, buffer index, call index, 0), and Box-Muller on 24-bit uniforms.

Pinned by the Random123 known-answer vectors in tests/test_oracle_golden.py::test_philox_known_answers.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (n, 4) uint32, key: (2,) ints -> (n, 4) uint32."""
    c = [np.asarray(counter[:, i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0, p1 & MASK, n2, p0 & MASK]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def normal_fill(n: int, seed: int, buffer_index: int, call_index: int) -> np.ndarray:
    """The n draws dof_step_begin writes into noise buffer ``buffer_index`` on its ``call_index``-th call."""
    quads = (n + 3) // 4
    ctr = np.zeros((quads, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(quads, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = buffer_index
    ctr[:, 2] = call_index
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    out = np.empty((quads, 4), dtype=np.float32)
    two_pi = np.float32(6.283185307179586)
    for h in range(2):
        u1 = ((r[:, 2 * h] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
        u2 = ((r[:, 2 * h + 1] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
        rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        out[:, 2 * h] = rad * np.cos(two_pi * u2)
        out[:, 2 * h + 1] = rad * np.sin(two_pi * u2)
    return out.reshape(-1)[:n]
