"""iden3 .r1cs reader and a plain-integer evaluator of  <A,z> * <B,z> - <C,z> = 0  — the checker of the on-device
constraint check (csrc/b3w_r1cs.hip).  Test infrastructure."""
import gzip
import os
import struct

import b3w_testlib as T

BUILTIN = os.path.join(T.PKG_DIR, "constraints", "blake3_compression.r1cs.gz")
BUILTIN_NOVA_O1 = os.path.join(T.PKG_DIR, "constraints", "blake3_nova_bn254_o1.r1cs.gz")
BUILTIN_NOVA_O2 = {"nova_bn254": os.path.join(T.PKG_DIR, "constraints", "blake3_nova_bn254.r1cs.gz"),
                   "nova_vesta": os.path.join(T.PKG_DIR, "constraints", "blake3_nova_vesta.r1cs.gz")}


def read_image(path=BUILTIN):
    raw = open(path, "rb").read()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


def parse(img):
    """-> dict(prime, n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, constraints=[(A, B, C)] with {wire: coef}, wire2label)"""
    assert img[:4] == b"r1cs" and struct.unpack_from("<I", img, 4)[0] == 1
    nsec = struct.unpack_from("<I", img, 8)[0]
    pos, sec = 12, {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", img, pos)
        sec[typ] = (pos + 12, size)
        pos += 12 + size
    assert pos == len(img)
    h, _ = sec[1]
    n8 = struct.unpack_from("<I", img, h)[0]
    assert n8 == 32
    prime = int.from_bytes(img[h + 4:h + 36], "little")
    nw, po, pi, pr, nl, m = struct.unpack_from("<IIIIQI", img, h + 36)
    c, clen = sec[2]
    end = c + clen
    cons = []
    for _ in range(m):
        abc = []
        for _ in range(3):
            n = struct.unpack_from("<I", img, c)[0]
            c += 4
            lc = {}
            for _ in range(n):
                w = struct.unpack_from("<I", img, c)[0]
                lc[w] = int.from_bytes(img[c + 4:c + 36], "little")
                c += 36
            abc.append(lc)
        cons.append(tuple(abc))
    assert c == end
    w2l = list(struct.unpack_from(f"<{nw}Q", img, sec[3][0])) if 3 in sec else None
    return dict(prime=prime, n_wires=nw, n_pub_out=po, n_pub_in=pi, n_prv_in=pr, n_labels=nl, constraints=cons, wire2label=w2l)


def body_to_ints(body):
    b = bytes(body)
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def violated(sys_, z, rows=None):
    """indices of the violated constraints (all of them, or only of `rows`)"""
    p, cons = sys_["prime"], sys_["constraints"]
    out = []
    wild = {i for i, v in enumerate(z) if v >= p}
    for k in (range(len(cons)) if rows is None else rows):
        a, b, c = cons[k]
        dot = lambda lc: sum(f * z[w] for w, f in lc.items()) % p
        if (dot(a) * dot(b) - dot(c)) % p or (wild and (wild & (a.keys() | b.keys() | c.keys()))):
            out.append(k)
    return out


def rows_of_wire(sys_):
    idx = {}
    for k, (a, b, c) in enumerate(sys_["constraints"]):
        for w in a.keys() | b.keys() | c.keys():
            idx.setdefault(w, []).append(k)
    return idx


def write_image(prime, nwires, constraints, npubout=0, npubin=0, nprvin=0):
    """iden3 .r1cs v1 image of `constraints` = [(A, B, C)] with A, B, C = {wire: coefficient}, or a list of (wire, coefficient)
    pairs where a test wants a wire repeated inside one list (test systems)."""
    import struct
    hdr = struct.pack("<I", 32) + prime.to_bytes(32, "little") + struct.pack("<IIIIQI", nwires, npubout, npubin, nprvin, nwires, len(constraints))
    body = bytearray()
    for parts in constraints:
        for lc in parts:
            body += struct.pack("<I", len(lc))
            for w, c in (sorted(lc.items()) if isinstance(lc, dict) else lc):
                body += struct.pack("<I", w) + (c % prime).to_bytes(32, "little")
    wmap = b"".join(struct.pack("<Q", i) for i in range(nwires))
    out = bytearray(b"r1cs" + struct.pack("<II", 1, 3))
    for typ, sec in ((1, hdr), (2, bytes(body)), (3, wmap)):
        out += struct.pack("<IQ", typ, len(sec)) + sec
    return bytes(out)
