"""Oracle (test infrastructure): numpy restatement of the device Latin-hypercube generator
(pinns-tf2.0_amd/csrc/kernels_sampling.h): Philox4x32-10 uniforms + 6-round Feistel permutation with cycle walking.

This is NOT a restatement of reference code: the reference samples on the host with pyDOE's `lhs`
(1d-burgers/burgersutil.py:122), which tests/ref_shims/pyDOE.py and utils/sampling.py restate and pin.  The device
generator produces the same *kind* of design (one point per stratum in every dimension); this module lets the
tests check the device integers bit for bit and the stratification property independently."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u64(a):
    return np.asarray(a, dtype=np.uint64)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (_u64(c) & M32 for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def mix(h):
    h = _u64(h) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def half_bits(n):
    bits = 1
    while bits < 62 and (1 << bits) < n:
        bits += 1
    return (bits + 1) // 2


def permute(i, n, k0, k1):
    hb = np.uint64(half_bits(n))
    mask = (np.uint64(1) << hb) - np.uint64(1)
    v = _u64(i).copy()
    todo = np.ones(v.shape, dtype=bool)
    while todo.any():
        w = v[todo]
        L, R = w >> hb, w & mask
        for r in range(6):
            key = np.uint64(k1 if r & 1 else k0)
            f = mix(R ^ key ^ np.uint64((0x9E3779B9 * (r + 1)) & 0xFFFFFFFF)) & mask
            L, R = R, L ^ f
        w = (L << hb) | R
        v[todo] = w
        todo[todo] = w >= np.uint64(n)
    return v


def lhs_points(n, seed, lb, ub, first=0, count=None):
    """Points [first, first+count) of the n-point design, float64 [count, 2]."""
    count = n - first if count is None else count
    i = np.arange(first, first + count, dtype=np.uint64)
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c0, c1, c2, c3 = philox4x32_10(i & M32, i >> np.uint64(32), np.zeros_like(i), np.full_like(i, 0x4C485321), lo, hi)
    inv = 1.0 / 9007199254740992.0
    u0 = (((c0 << np.uint64(32)) | c1) >> np.uint64(11)).astype(np.float64) * inv
    u1 = (((c2 << np.uint64(32)) | c3) >> np.uint64(11)).astype(np.float64) * inv
    p0 = permute(i, n, lo ^ 0x243F6A88, hi ^ 0x85A308D3)
    p1 = permute(i, n, lo ^ 0x13198A2E, hi ^ 0x03707344)
    lb, ub = np.asarray(lb, float), np.asarray(ub, float)
    x = lb[0] + (ub[0] - lb[0]) * ((p0.astype(np.float64) + u0) / float(n))
    t = lb[1] + (ub[1] - lb[1]) * ((p1.astype(np.float64) + u1) / float(n))
    return np.stack([x, t], axis=1), (p0, p1)
