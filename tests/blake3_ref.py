"""Plain BLAKE3 (hash mode, 32-byte output) in pure Python, written from the BLAKE3 specification
(sections 2.2-2.6) — an independent checker for the chained-mode planner.  Test infrastructure."""
import struct

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
CHUNK_START, CHUNK_END, PARENT, ROOT = 1, 2, 4, 8
M32 = 0xFFFFFFFF


def _rotr(x, r):
    return ((x >> r) | (x << (32 - r))) & M32


def compress(cv, m, t, b, d):
    v = list(cv) + IV[:4] + [t & M32, (t >> 32) & M32, b, d]
    m = list(m)

    def g(a, b_, c, d_, x, y):
        v[a] = (v[a] + v[b_] + x) & M32; v[d_] = _rotr(v[d_] ^ v[a], 16)
        v[c] = (v[c] + v[d_]) & M32; v[b_] = _rotr(v[b_] ^ v[c], 12)
        v[a] = (v[a] + v[b_] + y) & M32; v[d_] = _rotr(v[d_] ^ v[a], 8)
        v[c] = (v[c] + v[d_]) & M32; v[b_] = _rotr(v[b_] ^ v[c], 7)
    for _ in range(7):
        g(0, 4, 8, 12, m[0], m[1]); g(1, 5, 9, 13, m[2], m[3]); g(2, 6, 10, 14, m[4], m[5]); g(3, 7, 11, 15, m[6], m[7])
        g(0, 5, 10, 15, m[8], m[9]); g(1, 6, 11, 12, m[10], m[11]); g(2, 7, 8, 13, m[12], m[13]); g(3, 4, 9, 14, m[14], m[15])
        m = [m[PERM[i]] for i in range(16)]
    return [v[i] ^ v[i + 8] for i in range(8)] + [v[i + 8] ^ cv[i] for i in range(8)]


def _words(block):
    return list(struct.unpack("<16I", block.ljust(64, b"\0")))


def chunk_cv(data, counter, root):
    blocks = [data[i:i + 64] for i in range(0, len(data), 64)] or [b""]
    cv = IV
    for i, blk in enumerate(blocks):
        d = (CHUNK_START if i == 0 else 0) | (CHUNK_END if i == len(blocks) - 1 else 0)
        if root and i == len(blocks) - 1:
            d |= ROOT
        cv = compress(cv, _words(blk), counter, len(blk), d)[:8]
    return cv


def _subtree(data, first_chunk, root):
    if len(data) <= 1024:
        return chunk_cv(data, first_chunk, root)
    nchunks = (len(data) + 1023) // 1024
    k = 1
    while k * 2 < nchunks:
        k *= 2
    left = _subtree(data[:k * 1024], first_chunk, False)
    right = _subtree(data[k * 1024:], first_chunk + k, False)
    return compress(IV, left + right, 0, 64, PARENT | (ROOT if root else 0))[:8]


def hash_words(data):
    return _subtree(bytes(data), 0, True)


def blake3(data):
    return struct.pack("<8I", *hash_words(data))


# ---- the same hash, vectorised over chunks with numpy (for preimages of 1 GiB: BASELINE config 5) ----------------
def _compress_np(cv, m, t0, t1, b, d):
    """cv [n, 8], m [n, 16] uint32; t0, t1, b, d scalars or [n] arrays -> first 8 output words [n, 8]"""
    import numpy as np
    n = cv.shape[0]
    col = lambda x: np.full(n, x, np.uint32) if np.isscalar(x) else np.asarray(x, dtype=np.uint32)
    v = [cv[:, i].copy() for i in range(8)] + [np.full(n, IV[i], np.uint32) for i in range(4)] + [col(t0), col(t1), col(b), col(d)]
    msg = [m[:, i] for i in range(16)]

    def rot(x, r):
        return (x >> np.uint32(r)) | (x << np.uint32(32 - r))

    def g(a, b_, c, d_, x, y):
        v[a] = v[a] + v[b_] + x; v[d_] = rot(v[d_] ^ v[a], 16)
        v[c] = v[c] + v[d_]; v[b_] = rot(v[b_] ^ v[c], 12)
        v[a] = v[a] + v[b_] + y; v[d_] = rot(v[d_] ^ v[a], 8)
        v[c] = v[c] + v[d_]; v[b_] = rot(v[b_] ^ v[c], 7)
    for _ in range(7):
        g(0, 4, 8, 12, msg[0], msg[1]); g(1, 5, 9, 13, msg[2], msg[3]); g(2, 6, 10, 14, msg[4], msg[5]); g(3, 7, 11, 15, msg[6], msg[7])
        g(0, 5, 10, 15, msg[8], msg[9]); g(1, 6, 11, 12, msg[10], msg[11]); g(2, 7, 8, 13, msg[12], msg[13]); g(3, 4, 9, 14, msg[14], msg[15])
        msg = [msg[PERM[j]] for j in range(16)]
    return np.stack([v[i] ^ v[i + 8] for i in range(8)], axis=1)


def chunk_cvs_np(data):
    """data: uint8 array whose length is a multiple of 1024 (>= 2048) -> chaining values of its chunks [n, 8] uint32"""
    import numpy as np
    assert data.dtype == np.uint8 and data.size % 1024 == 0 and data.size >= 2048
    n = data.size // 1024
    words = data.view("<u4").reshape(n, 16, 16)
    cv = np.tile(np.array(IV, dtype=np.uint32), (n, 1))
    idx = np.arange(n, dtype=np.uint64)
    t0, t1 = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    for j in range(16):
        d = (CHUNK_START if j == 0 else 0) | (CHUNK_END if j == 15 else 0)
        cv = _compress_np(cv, words[:, j, :], t0, t1, 64, d)
    return cv


def tree_levels_np(cvs):
    """levels of a COMPLETE tree (number of chunks a power of two >= 2): [cvs, parents, ..., root[1, 8]]"""
    import numpy as np
    n = cvs.shape[0]
    assert n >= 2 and n & (n - 1) == 0
    levels = [cvs]
    ivs = np.array(IV, dtype=np.uint32)
    while levels[-1].shape[0] > 1:
        cur = levels[-1]
        k = cur.shape[0] // 2
        m = cur.reshape(k, 16)
        levels.append(_compress_np(np.tile(ivs, (k, 1)), m, 0, 0, 64, PARENT | (ROOT if k == 1 else 0)))
    return levels
