"""Value-level model of the reference circuits' signals ("atoms"), build-container tool only.

Used by tools/recover_layout.py to fingerprint witness slots of the committed circom WASMs
against named quantities, and by tools/gen_golden.py to produce probe inputs.  Pure Python
ints, full field semantics (values are canonical representatives in [0,p); `>>`/`&` act on
the representative, as the circom WASM runtime does).

Follows (reference file:line, relative to /root/reference):
  circuits/blake3_common.circom:15-26   Blake3Permute
  circuits/blake3_common.circom:142-203 ToBits / Bits33 / Bits34
  circuits/blake3_compression.circom:72-100  HalfFunG
  circuits/blake3_compression.circom:128-161 SingleRound
  circuits/blake3_compression.circom:171-228 Blake3Compression
  circuits/blake3_nova.circom:13-267    Blake3Nova and helpers
  circomlib 2.0.5 comparators/gates/bitify (not vendored; yarn.lock:1243) restated from
  the published templates.

ATOM NUMBERING (shared contract with oracle/b3w_oracle.c and csrc/b3w_atoms.h):
  0            ONE
  1..8         H[8]      compression chaining input
  9..24        M[16]     compression message
  25,26        T[2]
  27           B
  28           D
  29..44       O[16]     compression output words
  45+8k+j      half-G k=(r*8+g)*2+hf, j: 0 S1(add1.inp) 1 A(add1.out_word) 2 S3(add3.inp)
               3 C(add3.out_word) 4 D2(rxor2.out_word) 5 DI(rxor2 input word v[d])
               6 B4(rxor4.out_word) 7 BI(rxor4 input word v[b])         k in 0..111
  941..        nova atoms, see NOVA_NAMES
"""

BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
VESTA_Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
PALLAS_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
SIGMA = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
G_IDX = [(0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15),
         (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14)]

A_ONE, A_H, A_M, A_T, A_B, A_D, A_O, A_HG, A_NV = 0, 1, 9, 25, 27, 28, 29, 45, 941
HG_S1, HG_A, HG_S3, HG_C, HG_D2, HG_DI, HG_B4, HG_BI = range(8)
N_COMP_ATOMS = A_NV

NOVA_NARROW = (["n_blocks", "block_count"] + [f"h[{i}]" for i in range(8)] +
               ["chunk_idx_low", "chunk_idx_high", "leaf_depth", "total_depth", "depth"] +
               [f"m[{i}]" for i in range(16)] + ["b", "block_count_out", "depth_out", "is_root",
                "is_parent", "cp_in1", "cp_n2b_in", "ed_in1", "ed_n2b_in", "ed_out", "not_root",
                "not_parent", "e0", "e1", "is_last_block", "first", "ur_tmp", "ur_flag",
                "chunk_idx", "dl", "cdd_out", "decr_depth"] +
               [f"tmp_down[{i}]" for i in range(16)] + [f"m_is_parent[{i}]" for i in range(16)] +
               [f"tmp_is_par[{i}]" for i in range(16)] + [f"tmpIV[{i}]" for i in range(8)] +
               [f"eq_out[{i}]" for i in range(64)] + [f"bit_at_depth[{i}]" for i in range(64)])
NOVA_WIDE = (["root_inv", "e0_inv", "e1_inv", "root_isz_in", "e0_isz_in", "e1_isz_in", "e1_in1"] +
             [f"eq_inv[{i}]" for i in range(64)] + [f"eq_isz_in[{i}]" for i in range(64)] +
             [f"eq_in1[{i}]" for i in range(64)])
NOVA_NAMES = NOVA_NARROW + NOVA_WIDE
A_NVW = A_NV + len(NOVA_NARROW)
N_NOVA_ATOMS = A_NV + len(NOVA_NAMES)


def comp_atom_names():
    n = ["ONE"] + [f"H[{i}]" for i in range(8)] + [f"M[{i}]" for i in range(16)] + ["T[0]", "T[1]", "B", "D"]
    n += [f"O[{i}]" for i in range(16)]
    for k in range(112):
        r, g, hf = k // 16, (k // 2) % 8, k % 2
        for j in ("S1", "A", "S3", "C", "D2", "DI", "B4", "BI"):
            n.append(f"r{r}.g{g}.h{hf + 1}.{j}")
    return n


def atom_names(circuit):
    n = comp_atom_names()
    if circuit != "compression":
        n += NOVA_NAMES
    return n


class AssertFailed(Exception):
    pass


def _tobits_check(x, n, what):
    if x >> n:
        raise AssertFailed(what)


def eval_compression(p, h, m, t, b, d, atoms):
    """Fills atoms[0..940]; returns out[16]. All inputs are field elements in [0,p)."""
    atoms[A_ONE] = 1
    for i in range(8):
        atoms[A_H + i] = h[i]
    for i in range(16):
        atoms[A_M + i] = m[i]
    atoms[A_T], atoms[A_T + 1], atoms[A_B], atoms[A_D] = t[0], t[1], b, d
    v = list(h) + IV[:4] + [t[0], t[1], b, d]
    msg = list(m)
    for r in range(7):
        for g in range(8):
            a_, b_, c_, d_ = G_IDX[g]
            for hf in range(2):
                k = (r * 8 + g) * 2 + hf
                base = A_HG + 8 * k
                xy = msg[2 * g + hf]
                s1 = (v[a_] + v[b_] + xy) % p
                _tobits_check(s1, 34, f"Bits34 r{r} g{g} h{hf}")
                A = s1 & 0xFFFFFFFF
                DI = v[d_]
                _tobits_check(DI, 32, f"ToBits rxor2 r{r} g{g} h{hf}")
                x = DI ^ A
                R1, R2 = (16, 12) if hf == 0 else (8, 7)
                D2 = ((x >> R1) | (x << (32 - R1))) & 0xFFFFFFFF
                s3 = (v[c_] + D2) % p
                _tobits_check(s3, 33, f"Bits33 r{r} g{g} h{hf}")
                C = s3 & 0xFFFFFFFF
                BI = v[b_]
                _tobits_check(BI, 32, f"ToBits rxor4 r{r} g{g} h{hf}")
                y = BI ^ C
                B4 = ((y >> R2) | (y << (32 - R2))) & 0xFFFFFFFF
                atoms[base:base + 8] = [s1, A, s3, C, D2, DI, B4, BI]
                v[a_], v[b_], v[c_], v[d_] = A, B4, C, D2
        msg = [msg[SIGMA[j]] for j in range(16)]
    out = []
    for k in range(16):
        x = v[k]
        y = v[k + 8] if k < 8 else h[k - 8]
        _tobits_check(x, 32, f"outXor[{k}].x")
        _tobits_check(y, 32, f"outXor[{k}].y")
        out.append(x ^ y)
        atoms[A_O + k] = x ^ y
    return out


def _inv(x, p):
    x %= p
    return pow(x, p - 2, p) if x else 0


def eval_nova(p, inp, atoms):
    """inp: dict with n_blocks, block_count, h[8], chunk_idx_low, chunk_idx_high, leaf_depth,
    total_depth, depth, m[16], b (field elements). Fills atoms[0..N_NOVA_ATOMS)."""
    nv = {}
    f = lambda x: x % p
    n_blocks, block_count, h = inp["n_blocks"], inp["block_count"], inp["h"]
    cil, cih = inp["chunk_idx_low"], inp["chunk_idx_high"]
    leaf_depth, total_depth, depth, m, b = inp["leaf_depth"], inp["total_depth"], inp["depth"], inp["m"], inp["b"]
    nv.update(n_blocks=n_blocks, block_count=block_count, chunk_idx_low=cil, chunk_idx_high=cih,
              leaf_depth=leaf_depth, total_depth=total_depth, depth=depth, b=b)
    for i in range(8):
        nv[f"h[{i}]"] = h[i]
    for i in range(16):
        nv[f"m[{i}]"] = m[i]
    # check_depth (blake3_nova.circom:13-45 minus the Num2Bits(8) pair absent from the WASMs)
    nv["root_isz_in"] = f(0 - depth)
    nv["root_inv"] = _inv(nv["root_isz_in"], p)
    is_root = f(1 - nv["root_isz_in"] * nv["root_inv"])
    nv["is_root"] = is_root
    nv["cp_in1"] = f(leaf_depth - 1)
    nv["cp_n2b_in"] = f(depth + 256 - nv["cp_in1"])
    _tobits_check(nv["cp_n2b_in"], 9, "check_parent.n2b")
    is_parent = 1 - ((nv["cp_n2b_in"] >> 8) & 1)
    nv["is_parent"] = is_parent
    nv["ed_in1"] = f(depth + 1)
    nv["ed_n2b_in"] = f(leaf_depth + 256 - nv["ed_in1"])
    _tobits_check(nv["ed_n2b_in"], 9, "exceed_depth.lt.n2b")
    nv["ed_out"] = 1 - ((nv["ed_n2b_in"] >> 8) & 1)
    if nv["ed_out"] != 0:
        raise AssertFailed("exceed_depth.out === 0")
    # comp_d (Blake3GetFlag :122-167)
    nv["not_root"], nv["not_parent"] = 1 - is_root, 1 - is_parent
    nv["e0_isz_in"] = f(0 - block_count)
    nv["e0_inv"] = _inv(nv["e0_isz_in"], p)
    nv["e0"] = f(1 - nv["e0_isz_in"] * nv["e0_inv"])
    nv["e1_in1"] = f(n_blocks - 1)
    nv["e1_isz_in"] = f(nv["e1_in1"] - block_count)
    nv["e1_inv"] = _inv(nv["e1_isz_in"], p)
    nv["e1"] = f(1 - nv["e1_isz_in"] * nv["e1_inv"])
    nv["is_last_block"] = nv["e1"] * nv["not_parent"]
    nv["first"] = nv["e0"] * nv["not_parent"]
    nv["ur_tmp"] = is_parent + nv["e1"] - is_parent * nv["e1"]
    nv["ur_flag"] = nv["ur_tmp"] * is_root
    dflag = f(0 + nv["first"] + 2 * nv["is_last_block"] + 8 * nv["ur_flag"] + 4 * is_parent)
    # final_m (:86-120) / down_left_path (:47-84)
    chunk_idx = f(cil + cih * (1 << 32))
    nv["chunk_idx"] = chunk_idx
    _tobits_check(chunk_idx, 65, "down_left_path.n2b")
    bad = 0
    for i in range(64):
        nv[f"eq_in1[{i}]"] = f(total_depth - i - 2)
        nv[f"eq_isz_in[{i}]"] = f(nv[f"eq_in1[{i}]"] - depth)
        nv[f"eq_inv[{i}]"] = _inv(nv[f"eq_isz_in[{i}]"], p)
        nv[f"eq_out[{i}]"] = f(1 - nv[f"eq_isz_in[{i}]"] * nv[f"eq_inv[{i}]"])
        bad = f(bad + (1 - ((chunk_idx >> i) & 1)) * nv[f"eq_out[{i}]"])
        nv[f"bit_at_depth[{i}]"] = bad
    dl = f((1 - is_parent) + is_parent * bad)
    nv["dl"] = dl
    if f(dl * (1 - dl)) != 0:
        raise AssertFailed("down_left_path.out boolean")
    out_m = []
    for i in range(16):
        if i < 8:
            td = f(h[i] * dl)
            mp = f(m[i] * (1 - dl) + td)
        else:
            td = f(h[i - 8] * (1 - dl))
            mp = f(m[i - 8] * dl + td)
        tp = f(mp * is_parent)
        nv[f"tmp_down[{i}]"], nv[f"m_is_parent[{i}]"], nv[f"tmp_is_par[{i}]"] = td, mp, tp
        out_m.append(f(m[i] * (1 - is_parent) + tp))
    hc = []
    for i in range(8):
        nv[f"tmpIV[{i}]"] = f(IV[i] * is_parent)
        hc.append(f(h[i] * (1 - is_parent) + nv[f"tmpIV[{i}]"]))
    t = [f(cil * (1 - is_parent)), f(cih * (1 - is_parent))]
    out = eval_compression(p, hc, out_m, t, b, dflag, atoms)
    nv["block_count_out"] = f(block_count + (1 - is_parent))
    nv["cdd_out"] = nv["is_last_block"] + is_parent - nv["is_last_block"] * is_parent
    nv["decr_depth"] = f(nv["cdd_out"] * (1 - is_root))
    nv["depth_out"] = f(depth - nv["decr_depth"])
    for i, name in enumerate(NOVA_NAMES):
        atoms[A_NV + i] = nv[name]
    return out


def lcg_stream(seed):
    """test/utils.ts:4-21 LCG (a=1664525, c=1013904223, m=2^32)."""
    while True:
        seed = (1664525 * seed + 1013904223) % 4294967296
        yield seed
