"""numpy restatement of the engine's counter-based dropout RNG (TEST INFRASTRUCTURE ONLY: csrc/b2s_common.h: b2s_hash32 / b2s_keep /
make_drop).  The reference draws its masks from torch's global generator (F.dropout, transformer/modules.py:14-17); the engine's masks are a
pure function of (seed, op id, element index), so they cannot be compared with the reference's -- what is pinned is the function itself
(tests/test_gpu_ops.py: device mask == this restatement, bit for bit) and its statistics (tests/test_host_logic.py)."""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _u(x):
    return x & _M32


def _hash32(x):
    x = _u(np.uint64(x))
    x ^= x >> np.uint64(16); x = _u(x * np.uint64(0x7feb352d)); x ^= x >> np.uint64(15); x = _u(x * np.uint64(0x846ca68b)); x ^= x >> np.uint64(16)
    return x


def rand32(idx, key):
    """32 pseudo-random bits of element idx: lowbias32(idx * golden + key)."""
    return _hash32(_u(np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(key)))


def drop_key(seed, op_id):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    a = _hash32(np.uint64(((seed >> 32) + 0x51ed270b) & 0xFFFFFFFF))
    return int(_hash32(np.uint64(seed & 0xFFFFFFFF) ^ a ^ np.uint64((op_id * 0x85ebca6b + 0x1234567) & 0xFFFFFFFF)))


def drop_thresh(p):
    t = float(np.float32(p)) * 4294967296.0
    return 0xFFFFFFFF if t >= 4294967295.0 else int(t)


def keep_mask(p, seed, op_id, n):
    """bool[n]: element i is kept (drop if hash < p * 2^32)."""
    if p <= 0:
        return np.ones(n, dtype=bool)
    idx = np.arange(n, dtype=np.uint64)
    return rand32(idx, drop_key(seed, op_id)) >= np.uint64(drop_thresh(p))
