"""Counter-based random numbers shared by the host driver and the device sampler.

Philox4x32-10 (Salmon, Moraes, Dror & Shaw, SC'11).  The device kernels draw the stretch factors, partner indices
and acceptance uniforms themselves (``csrc/lcf_device.h``); the host only needs the per-step red/blue colouring,
which it generates here with the same generator so that every rank of a multi-GPU run derives the identical split
from ``(seed, step)`` without communicating.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32(c0, c1, c2, c3, key0, key1):
    """Vectorised Philox4x32-10.  Counter words are array-likes of equal shape; returns four uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & _MASK for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(key0) & _MASK, np.uint64(key1) & _MASK
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        c = [((p1 >> _S32) ^ c[1] ^ k0) & _MASK, p1 & _MASK, ((p0 >> _S32) ^ c[3] ^ k1) & _MASK, p0 & _MASK]
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return tuple(x.astype(np.uint32) for x in c)


def split_keys(seed, step, nwalkers):
    """Unique 64-bit sort keys for the colouring of ``step``: 50 random bits, then the walker id in the low 14 bits
    (the same key the device kernel ``k_make_perm`` sorts)."""
    if nwalkers > 16384:
        raise ValueError('at most 16384 walkers')
    wid = np.arange(nwalkers, dtype=np.uint64)
    r0, r1, _, _ = philox4x32(wid, np.uint64(step & 0xFFFFFFFF), np.uint64(2), np.uint64(7), seed & 0xFFFFFFFF,
                              (seed >> 32) & 0xFFFFFFFF)
    h = (r0.astype(np.uint64) << _S32) | r1.astype(np.uint64)
    return (h & ~np.uint64(0x3FFF)) | wid


def split_permutations(seed, first_step, nsteps, nwalkers):
    """``perm[k]`` = walker ids ordered by their key for step ``first_step + k``; the first half is colour 0.
    Equivalent to emcee's ``randomize_split`` (a fresh uniformly random balanced colouring every step)."""
    out = np.empty((nsteps, nwalkers), dtype=np.int32)
    for k in range(nsteps):
        out[k] = np.argsort(split_keys(seed, first_step + k, nwalkers), kind='stable')
    return out
