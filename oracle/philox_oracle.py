"""CPU oracle of the in-kernel noise generator (csrc/dsd_kernels.hpp::philox_normal).  TEST INFRASTRUCTURE ONLY.

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the counter-based generator torch's CUDA
`randn` is built on) keyed by the 64-bit seed, counter = (element index lo, hi, p_sample call index, 0); the first two output
words feed a Box-Muller transform in float32.  It replaces `noise_like` (usr/diff/shallow_diffusion_tts.py:38-41) when the
caller supplies no explicit noise; the reference draws from torch's global generator there, so only the DISTRIBUTION is shared
with the reference - this oracle pins the exact stream of OUR generator (known-answer vectors of the Philox paper below)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c, k):
    """c: 4 arrays of uint32 (counter words), k: 2 uint32 scalars/arrays.  Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint32) for v in c]
    k0, k1 = np.uint32(k[0]), np.uint32(k[1])
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def philox_normal(seed: int, step: int, n: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    c = ((idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32), np.full(n, step, dtype=np.uint32), np.zeros(n, dtype=np.uint32))
    x0, x1, _, _ = philox4x32_10(c, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u1 = ((x0 >> np.uint32(8)).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
    u2 = (x1 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.sqrt(np.float32(-2) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32)
