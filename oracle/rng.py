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


_WC = (np.uint64(0x85EBCA6B), np.uint64(0xC2B2AE35))


def keep_mask_attn(p, seed, op_id, rows, Lk, row0=0):
    """bool[rows, Lk]: the TRAINING kernels' attention-weight masks (csrc/b2s_common.h: b2s_wseed / b2s_wmix / b2s_keep_w, csrc/drop_sites.h:
    B2S_DROP_ATTN) for weight rows row0 .. row0 + rows - 1: a row has a seed drawn with the full hash, the four keys 4 kq .. 4 kq + 3 share one
    mixing step and take their 16-bit fields from two multiplies of it; a key is kept when its field, read as a signed 16-bit number, is
    >= ((p * 2^32) >> 16) - 32768."""
    if p <= 0:
        return np.ones((rows, Lk), dtype=bool)
    hq = (Lk + 3) // 4
    key = drop_key(seed, op_id)
    ts = (drop_thresh(p) >> 16) - 32768
    sd = rand32(np.arange(row0, row0 + rows, dtype=np.uint64), key)                       # seed(row) = hash32(row * golden + key)
    x = _u(sd[:, None] + np.arange(hq, dtype=np.uint64)[None, :] * np.uint64(0x9E3779B1))
    y = x ^ (x >> np.uint64(16))
    out = np.empty((rows, 4 * hq), dtype=bool)
    for j in (0, 1):
        w = _u(y * _WC[j])
        for half in (0, 1):
            f = ((w >> np.uint64(16 * half)) & np.uint64(0xFFFF)).astype(np.int64)
            out[:, 2 * j + half::4] = np.where(f >= 32768, f - 65536, f) >= ts
    return out[:, :Lk]


def _salt(rule, t):
    """Frame salt of the decode loop (csrc/drop_sites.h: B2S_SALT_*), XORed into the key of frame t."""
    t = np.uint64(int(t))
    if rule == 1:
        return int(_hash32(_u(t * np.uint64(2246822519) + np.uint64(3266489917))))
    if rule == 2:
        return int(_hash32(_u(t + np.uint64(0x9e3779b9))))
    if rule == 3:
        return int(_hash32(_u(t * np.uint64(2654435761) + np.uint64(77))))
    return 0


class DeviceMasks:
    """The masks the HIP engine draws at every dropout site of the model path (include/b2s_hip.h: b2s_dropout_site), for
    oracle.b2s_oracle.device_masks.

    seeds: {"encoder": s, "decoder": s, "postnet": s} -- the `seed` argument of the engine's segment calls (decode loop: the seed
    of b2s_decode_begin under "decoder").  site_info(site, layer, decode) -> (op_id, kind, salt_rule): the library's own table,
    queried by the test through the C ABI (nothing about op ids is restated here).  The index rules (kind: 0 = flat element index,
    1 = attention weights -- row seeds and key quads in the training kernels, keep_mask_attn) and the hash are the documented convention of that query.  decode: the autoregressive loop -- row r of a [B, rows, C] tensor (or query row r of
    attention weights) was drawn in frame r + frame_offset with a frame-salted key and the per-frame index rules.
    overrides: {(site, layer): op_id} -- deliberately wrong ids, for the test that shows the comparison notices."""

    def __init__(self, seeds, site_info, decode=False, overrides=None, ragged_lengths=None):
        """ragged_lengths: {"decoder": target_lengths} when the engine ran that segment on ragged rows (include/b2s_hip.h: b2s_decoder_compact_rows):
        the element index of its row sites then counts ragged rows -- frame t of utterance b is row sum(lengths[:b]) + t; padded frames
        (t >= length) do not exist there and keep everything (nothing downstream of the masked heads reads them)."""
        self.seeds, self.site_info, self.decode, self.overrides = seeds, site_info, decode, overrides or {}
        self.ragged = {k: [int(x) for x in v] for k, v in (ragged_lengths or {}).items()}
        self.calls = []

    def keep(self, site, layer, shape, p, frame_offset=0):
        assert site is not None, "dropout call without a site name under device_masks"
        op, kind, salt = self.site_info(site, layer, self.decode)
        op = self.overrides.get((site, layer), op)
        seed = self.seeds[site.split(".")[0]]
        key = drop_key(seed, op)
        th = np.uint64(drop_thresh(p))
        self.calls.append((site, layer, tuple(shape)))
        n = int(np.prod(shape))
        if not self.decode:
            assert n < 2 ** 32, "element index would wrap"
            if kind == 1:                           # weights [B, H, Lq, Lk] of the training kernels: row seeds + key quads
                return keep_mask_attn(p, seed, op, int(np.prod(shape[:-1])), shape[-1]).reshape(shape)
            lens = self.ragged.get(site.split(".")[0])
            if lens is not None and len(shape) == 3:            # rows [B, T, C] of a ragged segment: idx = (offset[b] + t) * C + c for t < length[b]
                B, T, Cc = shape
                assert len(lens) == B
                out = np.ones(shape, dtype=bool)
                off = 0
                for b, n_b in enumerate(lens):
                    idx = (np.arange(off, off + n_b, dtype=np.uint64)[:, None] * np.uint64(Cc) + np.arange(Cc, dtype=np.uint64)[None, :])
                    out[b, :n_b, :] = rand32(idx, key) >= th
                    off += n_b
                return out
            return (rand32(np.arange(n, dtype=np.uint64), key) >= th).reshape(shape)
        out = np.empty(shape, dtype=bool)
        if kind == 0:                               # rows [B, R, C]: frame r + offset, idx = b * C + c
            B, R, C = shape
            idx = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(C) + np.arange(C, dtype=np.uint64)[None, :])
            for r in range(R):
                out[:, r, :] = rand32(idx, key ^ _salt(salt, r + frame_offset)) >= th
        else:                                       # weights [B, H, Lq, Lk]: frame q, idx = (b * H + h) * 4096 + k
            B, H, Lq, Lk = shape
            zh = (np.arange(B * H, dtype=np.uint64) * np.uint64(4096)).reshape(B, H, 1)
            idx = zh + np.arange(Lk, dtype=np.uint64)[None, None, :]
            for q in range(Lq):
                out[:, :, q, :] = rand32(idx, key ^ _salt(salt, q + frame_offset)) >= th
        return out
