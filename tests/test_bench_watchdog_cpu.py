"""bench.py's launch watchdog (VERDICT r02 next #2b): a rank that never reaches the rendezvous must not cost the driver its whole
time limit.  Here, on a box without a GPU, both ranks hang before they have even looked for one: the parent ends them after
--launch-timeout, exits with 124 and leaves no process behind."""
import os
import subprocess
import sys
import time

import b3w_testlib as T


def _alive(pid):
    try:
        os.kill(pid, 0)
        return True
    except ProcessLookupError:
        return False
    except PermissionError:
        return True


def test_hung_ranks_are_ended_by_the_watchdog():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(B3W_BENCH_TEST_HANG_RANK="0,1")
    t0 = time.monotonic()
    # (--launch-timeout 0: no watchdog — this test is about the parent being ended from outside)
    p = subprocess.Popen([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--launch-timeout", "0"], env=env, cwd=T.ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(3.0)                                         # both children are up and asleep
    kids = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 2, kids
    p.terminate()                                           # what a driver's timeout does to the parent
    p.wait(timeout=30)
    assert p.returncode != 0
    time.sleep(0.5)
    assert not any(_alive(int(k)) for k in kids), "the launcher left ranks behind"
    assert time.monotonic() - t0 < 60


def test_watchdog_fires_when_one_rank_never_reaches_the_rendezvous():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # rank 1 hangs after importing torch (B3W_BENCH_TEST_HANG_AT=rendezvous); rank 0 stops at "needs a HIP device" on this box, which
    # is a failure of its own — so hang it too, a little later, and let the watchdog be the one that ends the run
    env.update(B3W_BENCH_TEST_HANG_RANK="0,1", B3W_BENCH_TEST_HANG_AT="imported")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--launch-timeout", "5"], env=env, cwd=T.ROOT,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "have not passed rendezvous after 5 s" in r.stderr and "ranks [0, 1]" in r.stderr
    assert r.stdout.strip() == ""
    assert time.monotonic() - t0 < 150
