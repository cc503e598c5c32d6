"""bench.py's own launcher on a box without a GPU: `--gpus N` must start N ranks (fresh children, before any HIP call)
and hand their failure back instead of hanging — here every rank stops at "needs a HIP device"."""
import os
import subprocess
import sys

import b3w_testlib as T


def test_gpus_n_spawns_n_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box behaviour")
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=T.ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    # (the launcher ends the other ranks as soon as one has failed: one to three of them get to say why)
    assert 1 <= r.stderr.count("needs a HIP device") <= 3, r.stderr[-1500:]
    assert r.stdout.strip() == ""
