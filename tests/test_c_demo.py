"""The C-ABI from plain C (examples/c/b3wit_demo.c): compiles against include/b3wit.h with gcc alone; without a GPU it
must fail loudly (status 101), on a GPU its .wtns must equal the reference's committed witness byte for byte."""
import os, shutil, subprocess
import pytest
import b3w_testlib as T

SRC = os.path.join(T.ROOT, "examples", "c", "b3wit_demo.c")
LIB = os.path.join(T.PKG_DIR, "libb3wit.so")


def _build(tmp_path):
    exe = str(tmp_path / "b3wit_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(T.ROOT, "include"), "-o", exe, SRC, "-ldl"])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_c_demo_builds_and_refuses_without_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, LIB, str(tmp_path / "o.wtns")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "status 101" in r.stderr and "no CPU path" in r.stderr
    assert not os.path.exists(tmp_path / "o.wtns")


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_c_demo_reproduces_the_reference_fixture(tmp_path):
    exe = _build(tmp_path)
    out = tmp_path / "o.wtns"
    r = subprocess.run([exe, LIB, str(out), "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert out.read_bytes() == T.golden_image("reference_testInp_witness.wtns.gz")
    assert "batch: 300 witnesses, 0 not ok" in r.stdout
