"""Deterministic synthetic inputs for the BASELINE.json configs.

All generators are seeded with the reference's LCG (test/utils.ts:4-21: a=1664525,
c=1013904223, m=2^32; `next()` returns the updated seed) and return packed u32 records, the
batch fast-path input format:

  compression record (28 words): h[8] m[16] t[2] b d
  nova record        (32 words): n_blocks block_count h[8] chunk_idx_low chunk_idx_high
                                 leaf_depth total_depth depth m[16] b
"""
import numpy as np

IV = np.array([0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
               0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19], dtype=np.uint32)

COMPRESSION_KEYS = (("h", 8), ("m", 16), ("t", 2), ("b", 1), ("d", 1))
NOVA_KEYS = (("n_blocks", 1), ("block_count", 1), ("h", 8), ("chunk_idx_low", 1), ("chunk_idx_high", 1),
             ("leaf_depth", 1), ("total_depth", 1), ("depth", 1), ("m", 16), ("b", 1))


class LCG:
    """Scalar LCG, identical stream to test/utils.ts LCG."""

    def __init__(self, seed):
        self.seed = int(seed)

    def next(self):
        self.seed = (1664525 * self.seed + 1013904223) % 4294967296
        return self.seed


def _lcg_columns(seeds, ndraws):
    """Vectorised: column j holds draw j of LCG(seeds[i]) for every instance i."""
    s = np.asarray(seeds, dtype=np.uint64)
    out = np.empty((s.shape[0], ndraws), dtype=np.uint32)
    for j in range(ndraws):
        s = (np.uint64(1664525) * s + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        out[:, j] = s.astype(np.uint32)
    return out


def lcg_stream(seed, nwords):
    """The first nwords draws of LCG(seed) as a uint32 array (config 4/5 preimage = their little-endian bytes,
    SURVEY.md 8(d) item 4).  Jump-ahead doubling: x[k+L] = a^L x[k] + c (a^L - 1)/(a - 1) mod 2^32, so the array
    doubles in ~log2(n) vector steps instead of n scalar draws."""
    out = np.empty(max(int(nwords), 1), dtype=np.uint32)
    out[0] = (1664525 * int(seed) + 1013904223) % 4294967296
    a, c, have = 1664525, 1013904223, 1                 # x -> a*x + c advances `have` draws
    while have < nwords:
        k = min(have, nwords - have)
        out[have:have + k] = out[:k] * np.uint32(a) + np.uint32(c)          # uint32 arithmetic wraps mod 2^32
        a, c = (a * a) % 4294967296, (a * c + c) % 4294967296
        have += k
    return out[:nwords]


def lcg_preimage(nbytes, seed=1):
    """nbytes of the little-endian byte stream of LCG(seed) (uint8 array)."""
    return lcg_stream(seed, (int(nbytes) + 3) // 4).view(np.uint8)[:int(nbytes)]


def gen_random_chunk(lcg, b=64, d=0, t0=0, t1=0, h=None):
    """test/utils.ts:34-56 genRandomChunk -> compression record (28 u32)."""
    assert b % 4 == 0 and b <= 64
    lcg.next()                                   # the reference burns one draw (utils.ts:45)
    nw = (b + 3) // 4
    m = [lcg.next() for _ in range(nw)] + [0] * (16 - b // 4)
    hh = IV if h is None else np.asarray(h, dtype=np.uint32)
    return np.array(list(hh) + m + [t0, t1, b, d], dtype=np.uint32)


def config1_cases():
    """Config 1 + the reference's own unit-test stream (test/blake3_hash.test.ts:30-59):
    one full block with h=IV,b=64,d=0,t=0, then 5 follow-on cases continuing the same LCG."""
    lcg = LCG(6429)
    recs = [gen_random_chunk(lcg)]
    for _ in range(5):
        b = (lcg.next() % 16) * 4
        t0 = lcg.next()
        t1 = lcg.next()
        recs.append(gen_random_chunk(lcg, b, 3, t0, t1))
    return np.stack(recs)


def config2_compression(n, first=0):
    """Config 2: instance i uses LCG(6429+i): h=8 draws, m=16, t0, t1, b=draw%65, d=draw%16."""
    c = _lcg_columns(np.arange(first, first + n, dtype=np.uint64) + np.uint64(6429), 28)
    rec = c.copy()
    rec[:, 26] = c[:, 26] % np.uint32(65)
    rec[:, 27] = c[:, 27] % np.uint32(16)
    return np.ascontiguousarray(rec)


def config3_nova(n, first=0):
    """Config 3: valid nova steps, 3/4 leaf steps and 1/4 parent steps.  Draw order of
    LCG(6429+i): n_blocks, kind, leaf_depth, h[8], m[16], chunk_idx_low, b, block_count, depth."""
    c = _lcg_columns(np.arange(first, first + n, dtype=np.uint64) + np.uint64(6429), 31).astype(np.uint64)
    n_blocks = 1 + c[:, 0] % 16
    parent = (c[:, 1] % 4) == 0
    ld = 1 + c[:, 2] % 32
    ld = np.where(parent & (ld < 2), 2, ld)
    h = c[:, 3:11]
    m = c[:, 11:27].copy()
    cil = c[:, 27]
    b = np.where(parent, 64, c[:, 28] % 65)
    block_count = np.where(parent, n_blocks, c[:, 29] % n_blocks)
    depth = np.where(parent, c[:, 30] % np.maximum(ld - 1, 1), ld - 1)
    m[parent, 8:] = 0
    rec = np.zeros((n, 32), dtype=np.uint32)
    rec[:, 0] = n_blocks
    rec[:, 1] = block_count
    rec[:, 2:10] = h
    rec[:, 10] = cil
    rec[:, 11] = 0
    rec[:, 12] = ld
    rec[:, 13] = ld
    rec[:, 14] = depth
    rec[:, 15:31] = m
    rec[:, 31] = b
    return rec


def record_to_input(rec, keys):
    """u32 record -> circom input object (what generate_witness.js reads from input.json)."""
    out, pos = {}, 0
    for k, n in keys:
        vals = [int(x) for x in rec[pos:pos + n]]
        out[k] = vals[0] if n == 1 else vals
        pos += n
    return out


def input_to_values(inp, keys):
    """circom input object -> flat list of python ints in record order (may be non-canonical)."""
    out = []
    for k, n in keys:
        v = inp[k]
        flat = []

        def fill(x):
            if isinstance(x, (list, tuple)):
                for y in x:
                    fill(y)
            else:
                flat.append(int(x, 0) if isinstance(x, str) else int(x))
        fill(v)
        assert len(flat) == n, (k, len(flat), n)
        out += flat
    return out
