import b3w_testlib as T
import blake3_ref as B


def test_blake3_reference_known_answers():
    # official test vector for the empty input, and the digests quoted in the reference's own tests
    # (rust_fold/src/main.rs:505-519 comments: 1024 and 68 zero bytes)
    assert B.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert B.blake3(bytes(68)).hex() == "155e0c74d6aa369966999c8a972e3d92e6266656fd74087fa46531db452965f5"
    assert B.blake3(bytes(1024)).hex() == "d6fd9de5bccf223f523b316c9cd1cf9a9d87ea42473d68e011dad13f09bf8917"


def test_blake3_reference_agrees_with_wasm_single_block():
    # [0u8;4]: h_out recorded from the reference's nova WASMs (tests/golden)
    want = [0x3bd02bec, 0x5f936bf8, 0xad714da3, 0x9f04bb7e, 0x7df8101f, 0x15523e34, 0xe6f9d811, 0xcd205662]
    assert B.hash_words(bytes(4)) == want
