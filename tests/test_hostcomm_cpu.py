"""csrc/b3w_hostcomm.cpp — the shared-memory all-gather behind b3w_comm_create_host (the transport that lets the native
exchange of the sharded chained pass run with several ranks on one GPU) — between forked processes under ASan + UBSan, no GPU:
message sizes around the slot size, a rank that never comes (everybody fails within the timeout, nothing left in /dev/shm), a
stale segment under the job's name."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hot-proofs-blake3-circom_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("hostcomm") / "harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", CSRC,
           os.path.join(ROOT, "tests", "hostcomm_harness.cpp"), os.path.join(CSRC, "b3w_hostcomm.cpp"), "-o", str(out), "-lrt", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


@pytest.mark.parametrize("nranks,slot", [(1, 128), (2, 4096), (3, 256), (5, 65536)])
def test_allgather_between_processes(harness, nranks, slot):
    r = subprocess.run([harness, "run", str(nranks), str(slot)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and f"{nranks} ranks, 0 failed" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_a_missing_rank_fails_everybody_within_the_timeout(harness):
    r = subprocess.run([harness, "missing", "3"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.count("failed as it should") == 2, (r.stdout, r.stderr[-2000:])


def test_a_stale_segment_under_the_name_is_replaced(harness):
    r = subprocess.run([harness, "stale", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "3 ranks, 0 failed" in r.stdout, (r.stdout, r.stderr[-2000:])
