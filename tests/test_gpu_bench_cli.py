"""bench.py's command line on a GPU: the default contract line and the chained-pass modes, on tiny workloads
(the numbers are not looked at; the line must carry every field of the contract and the run must verify itself)."""
import json
import os
import subprocess
import sys

import pytest

import b3w_testlib as T

pytestmark = pytest.mark.gpu


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=T.ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_default_line_has_the_contract_fields():
    d = _bench("--batch", "512", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["vs_baseline"] is None and d["dtype"] == "u32" and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["bound"] == "hbm"
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and d["config"]["verified_on_device"] is True


@pytest.mark.parametrize("consumer", ["none", "commit", "commit-only"])
def test_chain_workload_with_each_consumer(consumer):
    d = _bench("--workload", "chain", "--preimage-mib", "0.25", "--steps", "1", "--warmup", "1", "--consumer", consumer)
    assert d["scaling"] == "strong" and d["value"] > 0 and d["config"]["n_chunks"] == 256 and d["config"]["path_len"] == 8
    assert ("no bodies" in d["config"]["consumer"]) == (consumer == "commit-only")
