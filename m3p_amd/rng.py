"""NumPy twin of the counter-based dropout RNG in csrc/common.hpp (m3p_hash32 /
m3p_keep): the keep mask of every dropout site is a pure function of
(stream seed, linear element index), so tests can hand the oracle the exact mask a
kernel used and compare outputs / gradients element-wise with dropout switched on."""
import numpy as np


def hash32(idx, seed):
    with np.errstate(over='ignore'):
        h = (np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
        h = h.astype(np.uint32)
        h ^= h >> np.uint32(16)
        h = (h.astype(np.uint64) * np.uint64(0x21f0aaad) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        h ^= h >> np.uint32(15)
        h = (h.astype(np.uint64) * np.uint64(0x735a2d97) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        h ^= h >> np.uint32(15)
    return h


def keep_mask(n_elems, seed, p, shape=None):
    """Boolean keep mask for elements 0..n_elems-1 of a dropout stream."""
    thresh = int(round(float(p) * (1 << 24)))
    k = (hash32(np.arange(n_elems, dtype=np.uint64), seed) >> np.uint32(8)) >= np.uint32(thresh)
    return k.reshape(shape) if shape is not None else k


def stream_seed(base_seed, step, site):
    """Per-(optimizer step, dropout site) 32-bit stream key; `site` enumerates the dropout
    call sites of the model (layer * 8 + k).  Plain integer mixing, identical on host and in tests."""
    x = (int(base_seed) * 0x9E3779B97F4A7C15 + int(step) * 0xBF58476D1CE4E5B9 + int(site) * 0x94D049BB133111EB) & (2 ** 64 - 1)
    x ^= x >> 31
    x = (x * 0xD6E8FEB86659FD93) & (2 ** 64 - 1)
    x ^= x >> 32
    return int(x & 0xFFFFFFFF)
