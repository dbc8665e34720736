"""Philox4x32-10 counter-based RNG and the keyed standard-normal draw used by
the reparameterisation sampler.  numpy restatement (oracle; test infrastructure).

Algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as
1, 2, 3" (SC'11), Philox-4x32 with 10 rounds.  The reference only *names* the
reparameterisation trick (``README.md:51`` -> refs [10][11][12]); it contains no
sampler, so the keying convention below is this build's definition:

    counter = (block, sample, stream, 0)   key = (seed_lo, seed_hi)

``block`` indexes groups of four consecutive parameters, ``sample`` the Monte
Carlo draw, ``stream`` separates independent uses (0 = weights, 1 = scale
parameter, ...).  Nothing depends on the rank or the GPU count, so every rank
draws identical noise.

Each 4x32 output block gives two Box-Muller pairs, i.e. four normals:
    u = (x + 0.5) * 2**-32           (never 0 or 1)
    z0 = sqrt(-2 ln u0) cos(2 pi u1),  z1 = sqrt(-2 ln u0) sin(2 pi u1)
    z2 = sqrt(-2 ln u2) cos(2 pi u3),  z3 = sqrt(-2 ln u2) sin(2 pi u3)
all in float64.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: uint32 array [..., 2] (broadcastable).

    Returns uint32 array [..., 4].
    """
    counter = np.asarray(counter, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    c0, c1, c2, c3 = (counter[..., i].astype(np.uint64) for i in range(4))
    k0 = np.broadcast_to(key[..., 0], c0.shape).astype(np.uint32)
    k1 = np.broadcast_to(key[..., 1], c0.shape).astype(np.uint32)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = PHILOX_M0 * c0
            p1 = PHILOX_M1 * c2
            hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
            n0 = hi1 ^ c1 ^ k0.astype(np.uint64)
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1.astype(np.uint64)
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = (k0 + PHILOX_W0).astype(np.uint32)
            k1 = (k1 + PHILOX_W1).astype(np.uint32)
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _box_muller4(words):
    """uint32 [..., 4] -> float64 normals [..., 4]."""
    u = (words.astype(np.float64) + 0.5) * (2.0 ** -32)
    r0 = np.sqrt(-2.0 * np.log(u[..., 0]))
    r1 = np.sqrt(-2.0 * np.log(u[..., 2]))
    t0 = 2.0 * np.pi * u[..., 1]
    t1 = 2.0 * np.pi * u[..., 3]
    return np.stack([r0 * np.cos(t0), r0 * np.sin(t0),
                     r1 * np.cos(t1), r1 * np.sin(t1)], axis=-1)


def normal_draws(seed, n_samples, n_params, stream=0, step=0):
    """Standard normals eps[s, d], float64, shape [n_samples, n_params].

    eps[s, 4*b + j] is output word-pair j of Philox(counter=(b, s, stream, step),
    key=(seed & 0xffffffff, seed >> 32)).
    """
    n_blocks = (n_params + 3) // 4
    b = np.arange(n_blocks, dtype=np.uint32)
    s = np.arange(n_samples, dtype=np.uint32)
    ctr = np.zeros((n_samples, n_blocks, 4), dtype=np.uint32)
    ctr[..., 0] = b[None, :]
    ctr[..., 1] = s[:, None]
    ctr[..., 2] = np.uint32(stream)
    ctr[..., 3] = np.uint32(step & 0xFFFFFFFF)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    z = _box_muller4(philox4x32_10(ctr, key))          # [S, n_blocks, 4]
    return z.reshape(n_samples, n_blocks * 4)[:, :n_params].copy()


def uniform_words(seed, n_samples, n_params, stream=0, step=0):
    """Raw uint32 words laid out exactly like ``normal_draws`` (for device checks)."""
    n_blocks = (n_params + 3) // 4
    ctr = np.zeros((n_samples, n_blocks, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(n_blocks, dtype=np.uint32)[None, :]
    ctr[..., 1] = np.arange(n_samples, dtype=np.uint32)[:, None]
    ctr[..., 2] = np.uint32(stream)
    ctr[..., 3] = np.uint32(step & 0xFFFFFFFF)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key).reshape(n_samples, n_blocks * 4)[:, :n_params].copy()
