"""CPU oracle of the device-side data generator.

TEST INFRASTRUCTURE ONLY (import rule of oracle/dccn_oracle.py).  The *deterministic* stages of the
generator (bits -> constellation -> resource grid -> IFFT -> cyclic prefix; static multipath taps ->
'same' FIR; power normalisation + noise scaling) are pinned by the host substrate dl_ofdm_amd/ofdm.py
and radio.py, which are themselves bit-pinned to the reference by tests/test_golden_substrate.py --
tests feed the same bits / tap draws / noise draws to both.  What this file adds is the restatement of
the generator's *random streams*: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as
1, 2, 3", SC'11 -- the counter-based generator TensorFlow's and cuRAND's device RNGs also use) with
the known-answer vectors of the Random123 distribution, and the uniform / Box-Muller transforms.
The reference draws from numpy's Mersenne Twister on the host (dev/py/util.py:25-29,
radio.py:359,513-526); a device generator cannot reproduce that sequence, so stream parity is
statistical by construction and the tests say so.
"""
from __future__ import annotations

import numpy as np

PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
STREAM_BITS, STREAM_TAPS, STREAM_NOISE, STREAM_DOPPLER = 0, 1, 2, 3

# Random123 kat_vectors, philox4x32-10: (counter, key) -> output
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """ctr uint32 [..., 4], key (k0, k1) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c[0]
        p1 = np.uint64(PHILOX_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(PHILOX_W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def counters(index: np.ndarray, stream: int, offset: int) -> np.ndarray:
    """the generator's counter layout: (index low, index high, stream id, batch offset)"""
    idx = np.asarray(index, dtype=np.uint64)
    out = np.empty(idx.shape + (4,), dtype=np.uint32)
    out[..., 0] = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out[..., 1] = (idx >> np.uint64(32)).astype(np.uint32)
    out[..., 2] = np.uint32(stream)
    out[..., 3] = np.uint32(offset)
    return out


def key_of(seed: int):
    return (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def uniform01(w: np.ndarray) -> np.ndarray:
    """uint32 -> float32 in (0, 1): ((w >> 8) + 0.5) * 2^-24"""
    return ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def box_muller(w0: np.ndarray, w1: np.ndarray):
    u1, u2 = uniform01(w0), uniform01(w1)
    r = np.sqrt(np.float32(-2.0) * np.log(u1))
    th = np.float32(2.0 * np.pi) * u2
    return (r * np.cos(th)).astype(np.float32), (r * np.sin(th)).astype(np.float32)


def bits(seed: int, offset: int, n_frames: int, D: int, nbits: int) -> np.ndarray:
    """label bits [n, D, nbits]: bit j of cell (frame, d) = bit j of word 0 of counter frame*D + d"""
    idx = np.arange(n_frames * D, dtype=np.uint64)
    w = philox4x32_10(counters(idx, STREAM_BITS, offset), key_of(seed))[:, 0]
    return ((w[:, None] >> np.arange(nbits, dtype=np.uint32)[None, :]) & np.uint32(1)).astype(np.int32).reshape(
        n_frames, D, nbits)


def tap_normals(seed: int, offset: int, n_frames: int, n_taps: int) -> np.ndarray:
    """standard-normal pairs [n, n_taps, 2] of the static tap draw (host: np.random.normal per frame)"""
    idx = np.arange(n_frames * n_taps, dtype=np.uint64)
    w = philox4x32_10(counters(idx, STREAM_TAPS, offset), key_of(seed))
    z0, z1 = box_muller(w[:, 0], w[:, 1])
    return np.stack([z0, z1], -1).reshape(n_frames, n_taps, 2)


def noise_normals(seed: int, offset: int, n_pairs: int) -> np.ndarray:
    """standard-normal IQ pairs [n_pairs, 2] of the AWGN stage (host: np.random.randn)"""
    idx = np.arange(n_pairs, dtype=np.uint64)
    w = philox4x32_10(counters(idx, STREAM_NOISE, offset), key_of(seed))
    z0, z1 = box_muller(w[:, 0], w[:, 1])
    return np.stack([z0, z1], -1)


def doppler_thetas(seed: int, offset: int, n_frames: int, n_taps: int, ss: int = 48) -> np.ndarray:
    """uniform phases [n, 2, ss, n_taps] in (0, 2 pi) of the Jakes taps (host: two np.random.uniform draws per frame)"""
    idx = np.arange(n_frames * 2 * ss * n_taps, dtype=np.uint64)
    w = philox4x32_10(counters(idx, STREAM_DOPPLER, offset), key_of(seed))[:, 0]
    return (np.float32(2.0 * np.pi) * uniform01(w)).reshape(n_frames, 2, ss, n_taps)
