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


def test_ranks_that_close_and_reopen_under_the_same_name(harness):
    """open has two phases (barrier, rank 0 unlinks the name, barrier), so no rank comes out of it while the name is still
    linked — five rounds of open / gather / close / open again under one name, the ranks returning at different times."""
    r = subprocess.run([harness, "reopen", "4", "512"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "4 ranks, 0 failed" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_eight_ranks_as_threads_of_one_process(harness):
    r = subprocess.run([harness, "threads", "8", "1024"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "8 ranks, 0 failed" in r.stdout, (r.stdout, r.stderr[-2000:])
