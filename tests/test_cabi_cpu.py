"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/b3wit.h declares, refuses to compute without a HIP device (no CPU fallback), and the host
logic mirrored from the reference's witness_calculator.js behaves like it."""
import ctypes, os, re
import numpy as np
import pytest
import b3w_testlib as T


def test_library_exports_every_declared_symbol():
    m = T.pkg()
    L = m.lib()
    hdr = open(os.path.join(T.ROOT, "include", "b3wit.h")).read()
    declared = set(re.findall(r"\b(b3w_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    assert declared == set(m.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.b3w_abi_version() >> 16 == 1


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = T.pkg()
    with pytest.raises(m.B3WError) as e:
        m.Context("compression", 0)
    assert e.value.status == m.B3W_E_NO_DEVICE
    with pytest.raises(m.B3WError):
        m.builder("nova_vesta")


def test_identify_wasm_rejects_unknown():
    m = T.pkg()
    assert m.lib().b3w_identify_wasm(b"\x00asm\x01\x00\x00\x00", 8) == -1
    assert m.lib().b3w_identify_wasm(b"", 0) == -1


def test_identify_wasm_knows_reference_binaries():
    ref = "/root/reference/build"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present (GPU box)")
    m = T.pkg()
    for rel, cid in (("blake3_compression/blake3_compression_js/blake3_compression.wasm", 0),
                     ("blake3_nova_js/blake3_nova.wasm", 1), ("blake3_nova_pasta_js/blake3_nova_pasta.wasm", 2),
                     ("blake3_nova/blake3_nova_js/blake3_nova.wasm", 3),
                     ("blake3_nova_pasta/blake3_nova_pasta_js/blake3_nova_pasta.wasm", 3)):
        code = open(os.path.join(ref, rel), "rb").read()
        assert m.lib().b3w_identify_wasm(code, len(code)) == cid


def test_fnv_hash_matches_reference_definition():
    # witness_calculator.js:325-337; known FNV-1a-64 vectors
    m = T.pkg()
    assert m.fnv_hash("") == 0xCBF29CE484222325
    assert m.fnv_hash("a") == 0xAF63DC4C8601EC8C
    assert m.fnv_hash("foobar") == 0x85944171F73967E8


def test_flat_array():
    m = T.pkg()
    assert m.flat_array([[1, 2], [3, [4, 5]], 6]) == [1, 2, 3, 4, 5, 6]
    assert m.flat_array(7) == [7]


def test_lcg_stream_matches_reference_fixture_inputs():
    # first outputs for seed 6429 after the burned draw are m[0..3] of testInp
    W = T.workloads()
    rec = W.config1_cases()[0]
    assert list(rec[8:12]) == [1774135639, 3474699978, 2774906785, 410807436]
    assert list(rec[:8]) == list(W.IV) and list(rec[24:]) == [0, 0, 64, 0]
    g = T.golden("compression")
    assert W.record_to_input(rec, W.COMPRESSION_KEYS) == g["cases"][0]["input"]


def test_workload_generators_shapes_and_validity():
    W = T.workloads()
    c2 = W.config2_compression(64)
    assert c2.shape == (64, 28) and c2.dtype == np.uint32
    assert (c2[:, 26] <= 64).all() and (c2[:, 27] < 16).all()
    assert (W.config2_compression(8, first=3) == c2[3:11]).all()
    c3 = W.config3_nova(512)
    assert c3.shape == (512, 32)
    depth, leaf = c3[:, 14].astype(np.int64), c3[:, 12].astype(np.int64)
    assert (depth < leaf).all()
    parent = depth < leaf - 1
    assert 0.15 < parent.mean() < 0.35
    assert (c3[parent, 23:31] == 0).all() and (c3[parent, 31] == 64).all()
    assert (c3[~parent, 1] < c3[~parent, 0]).all()
    # every generated step is accepted by the oracle
    bad, _ = T.oracle_batch_u32("nova_bn254", c3[:64])
    assert bad == 0
