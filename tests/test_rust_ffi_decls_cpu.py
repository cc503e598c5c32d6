"""integrations/rust/b3wit_ffi.rs cannot be compiled here (no Rust toolchain), but its `extern "C"` declarations can be held
against include/b3wit.h: every function the stub binds exists in the header with the same parameters IN THE SAME ORDER — same
names (a swap of two u64 parameters would otherwise go unseen), the same integer widths, pointer-ness and constness — and the
same return type.  (the stub replaces
circom_scotia::calculate_witness at rust_fold/src/blake3_circuit.rs:305.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_args(s):
    out, depth, cur = [], 0, ""
    s = s.replace("->", "\u2192")                           # (a return arrow is not a closing bracket)
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return [a.replace("\u2192", "->") for a in out]


def _c_decls():
    text = open(os.path.join(ROOT, "include", "b3wit.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(b3w_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        decls[name] = (ret, [] if args.strip() in ("", "void") else _split_args(args))
    return decls


INT = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "int": "i32",
       "uint8_t": "u8", "char": "c_char", "uint16_t": "u16"}


def _c_fnptr_typedefs():
    """typedef void|int32_t (*name)(args); -> {name: (return shape, [arg shapes])}"""
    text = open(os.path.join(ROOT, "include", "b3wit.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return {m.group(2): (_c_shape(m.group(1)), [_c_shape(a) for a in _split_args(m.group(3))])
            for m in re.finditer(r"typedef\s+(void|int32_t)\s*\(\s*\*\s*(b3w_[a-z0-9_]+)\s*\)\s*\(([^;]*)\)\s*;", text)}


def _c_shape(decl):
    """C parameter or return type -> canonical shape string, e.g. '*const u8', '*mut *mut opaque', 'u32', 'fnptr'."""
    d = decl.strip()
    if "(*" in d or re.match(r"b3w_[a-z0-9_]*(consumer|_fn)\b", d):
        return "fnptr"
    d = re.sub(r"\[[^\]]*\]", "*", d)                        # uint8_t out[76] -> pointer
    stars = d.count("*")
    const = bool(re.search(r"\bconst\b", d))
    base = re.sub(r"\bconst\b|\bstruct\b|\*", " ", d).split()
    base = base[0] if base else "void"
    if stars == 0:
        return INT.get(base, base)
    kind = INT.get(base, "opaque")                          # struct handles, void, hipStream_t-as-void* are opaque
    return ("*mut " * (stars - 1)) + ("*const " if const else "*mut ") + kind


def _rs_shape(t):
    t = t.strip()
    if t.startswith("Option<extern"):
        return "fnptr"
    t = t.replace("c_void", "opaque")
    return t


def test_every_bound_function_matches_the_header():
    src = open(os.path.join(ROOT, "integrations", "rust", "b3wit_ffi.rs")).read()
    c = _c_decls()
    bound = re.findall(r"\bfn\s+(b3w_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([A-Za-z0-9_]+))?\s*;", src, flags=re.S)
    assert len(bound) >= 28 and {"b3w_calc_witness", "b3w_create", "b3w_chain_run_leaves", "b3w_chain_run_parents_sharded", "b3w_chain_allgather_hout",
                                 "b3w_comm_create", "b3w_comm_create_host", "b3w_comm_create_external", "b3w_chain_shard"} <= {b[0] for b in bound}
    for name, args, ret in bound:
        assert name in c, f"{name} is not declared in include/b3wit.h"
        c_ret, c_args = c[name]
        assert _c_shape(c_ret) == (ret or "void"), (name, c_ret, ret)
        rs_named = [a.split(":", 1) for a in _split_args(" ".join(args.split()))]
        rs_args = [a[1] for a in rs_named]
        assert len(rs_args) == len(c_args), (name, rs_args, c_args)
        # parameter names, position by position (C: the last identifier of the declarator)
        c_names = [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", re.sub(r"\[[^\]]*\]", "", ca))[-1] for ca in c_args]
        assert [a[0].strip() for a in rs_named] == c_names, (name, [a[0].strip() for a in rs_named], c_names)
        for k, (ra, ca) in enumerate(zip(rs_args, c_args)):
            want, got = _c_shape(ca), _rs_shape(ra)
            if want == "fnptr":                                       # the callback's own parameters and return type too
                cb_ret, cb = _c_fnptr_typedefs()[ca.split()[0]]
                inner = re.search(r"fn\((.*)\)", ra).group(1)
                rs_cb = [_rs_shape(a) for a in _split_args(inner)]
                assert [x.count("*") for x in cb] == [x.count("*") for x in rs_cb] and \
                       [x.split()[-1] for x in cb] == [x.split()[-1] for x in rs_cb], (name, cb, rs_cb)
                rs_ret = re.search(r"\)\s*->\s*([A-Za-z0-9_]+)\s*>", ra)
                assert (rs_ret.group(1) if rs_ret else "void") == cb_ret, (name, cb_ret, ra)
            if want.endswith("opaque") and got.endswith("opaque"):       # handles: constness of an opaque handle is advisory
                assert want.count("*") == got.count("*"), (name, k, ca, ra)
            else:
                assert want == got, (name, k, ca, ra)


def test_the_callback_type_aliases_match_the_header():
    """the typed consumer callback of the chained-pass wrapper (BatchConsumer) and the caller's collective (AllgatherFn): same
    parameter names, order and shapes as b3w_batch_consumer / b3w_allgather_fn"""
    src = open(os.path.join(ROOT, "integrations", "rust", "b3wit_ffi.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "b3wit.h")).read(), flags=re.S)
    for rs_name, c_name in (("BatchConsumer", "b3w_batch_consumer"), ("AllgatherFn", "b3w_allgather_fn")):
        m = re.search(rf"pub type {rs_name} = extern \"C\" fn\((.*?)\)\s*(?:->\s*([a-z0-9]+))?;", src, flags=re.S)
        assert m, rs_name
        rs = [a.split(":", 1) for a in _split_args(" ".join(m.group(1).split()))]
        c = re.search(rf"typedef\s+(void|int32_t)\s*\(\s*\*\s*{c_name}\s*\)\s*\(([^;]*)\)\s*;", hdr)
        c_args = _split_args(c.group(2))
        assert [a[0].strip() for a in rs] == [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", ca)[-1] for ca in c_args], rs_name
        assert [_rs_shape(a[1]) for a in rs] == [_c_shape(ca) for ca in c_args], rs_name
        assert (m.group(2) or "void") == _c_shape(c.group(1)), rs_name


def test_the_wrappers_are_the_call_site_and_the_chained_pass():
    """what cannot be compiled here stays declarations + two safe wrappers: the reference's call site (Calculator, for
    rust_fold/src/blake3_circuit.rs:305) and the chained pass that replaces its loop (Fold, for rust_fold/src/main.rs:166-179)"""
    src = open(os.path.join(ROOT, "integrations", "rust", "b3wit_ffi.rs")).read()
    wrappers = set(re.findall(r"\bpub fn (?!b3w_)([a-z0-9_]+)", src))
    assert wrappers == {"fnv1a64", "new", "calculate_witness", "commit_only", "fold_shaped", "violations", "run", "commitments"}, wrappers
    assert "impl Drop for Fold" in src and "impl Drop for Calculator" in src
    assert len(src.splitlines()) <= 210          # (r06: + the private `steps()` helper that sizes violations() / commitments() from b3w_chain_info)
    assert "fn steps(&self)" in src and "pub fn violations(&mut self) ->" in src and "pub fn commitments(&mut self) ->" in src


def test_constants_match_the_header():
    src = open(os.path.join(ROOT, "integrations", "rust", "b3wit_ffi.rs")).read()
    hdr = open(os.path.join(ROOT, "include", "b3wit.h")).read()
    ids = dict(re.findall(r"#define\s+(B3W_CIRCUIT_[A-Z0-9_]+)\s+(\d+)", hdr))
    ids.update(dict(re.findall(r"#define\s+(B3W_CURVE_[A-Z0-9_]+)\s+(\d+)", hdr)))
    for rs_name, c_name in (("CIRCUIT_NOVA_BN254", "B3W_CIRCUIT_NOVA_BN254"), ("CIRCUIT_NOVA_VESTA", "B3W_CIRCUIT_NOVA_VESTA"),
                            ("CURVE_BN254_G1", "B3W_CURVE_BN254_G1"), ("CURVE_PALLAS", "B3W_CURVE_PALLAS")):
        m = re.search(rf"pub const {rs_name}: i32 = (\d+);", src)
        assert m and c_name in ids and m.group(1) == ids[c_name], (rs_name, ids.get(c_name))
