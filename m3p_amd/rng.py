"""NumPy twin of the counter-based dropout RNG in csrc/common.hpp (m3p_hash32 /
m3p_keep): the keep mask of every dropout site is a pure function of
(stream seed, linear element index), so tests can hand the oracle the exact mask a
kernel used and compare outputs / gradients element-wise with dropout switched on."""
import numpy as np


def hash32(idx, seed):
    """m3p_hash32 of csrc/common.hpp (round 6: add the seed, three rounds of xor-shift-16 + 24-bit multiply-add, xor-shift-16)."""
    m32, m24, s16 = np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFF), np.uint64(16)
    h = (np.asarray(idx, dtype=np.uint64) + np.uint64(int(seed) & 0xFFFFFFFF)) & m32
    for k in (0x9E3779, 0x85EBCB, 0xC2B2AF):
        h ^= h >> s16
        h = (h + (h & m24) * np.uint64(k)) & m32
    h ^= h >> s16
    return h.astype(np.uint32)


def keep_mask(n_elems, seed, p, shape=None):
    """Boolean keep mask for elements 0..n_elems-1 of a dropout stream: element i takes the low (i even) or high (i odd)
    16 bits of hash32(i >> 1) and is kept iff they are >= thresh24 >> 8, thresh24 = round(p * 2^24) (csrc/common.hpp:
    one hash serves two elements)."""
    thresh16 = int(round(float(p) * (1 << 24))) >> 8
    idx = np.arange(n_elems, dtype=np.uint64)
    h = hash32(idx >> np.uint64(1), seed)
    half = np.where((idx & np.uint64(1)) != 0, h >> np.uint32(16), h & np.uint32(0xFFFF))
    k = half >= np.uint32(thresh16)
    return k.reshape(shape) if shape is not None else k


def stream_seed(base_seed, step, site):
    """Per-(optimizer step, dropout site) 32-bit stream key; `site` enumerates the dropout
    call sites of the model (layer * 8 + k).  Plain integer mixing, identical on host and in tests."""
    x = (int(base_seed) * 0x9E3779B97F4A7C15 + int(step) * 0xBF58476D1CE4E5B9 + int(site) * 0x94D049BB133111EB) & (2 ** 64 - 1)
    x ^= x >> 31
    x = (x * 0xD6E8FEB86659FD93) & (2 ** 64 - 1)
    x ^= x >> 32
    return int(x & 0xFFFFFFFF)
