"""tools/wasm_baseline.py — the reference's WASM witness generator on all cores, the CPU number bench.py prints beside the GPU's
(`cpu_baseline.reference_wasm`).  Here: the recorded result has the fields bench.py reads; bench.py's leg quotes it when there is no
reference checkout and runs the tool live when there is one (build container only: a short run of one circuit, two workers)."""
import importlib.util
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_recorded_baseline_has_what_bench_reads():
    doc = json.load(open(os.path.join(ROOT, "profiles", "wasm_baseline.json")))
    assert set(doc["circuits"]) >= {"compression", "nova_vesta", "nova_bn254"}
    for c, r in doc["circuits"].items():
        assert r["unit"] == "witnesses/s" and r["cores"] >= 1 and r["value"] > 1 and abs(r["per_core"] * r["cores"] - r["value"]) < 1e-6 * r["value"]
        assert r["seconds"] >= 25 and r["witnesses"] == sum(r["per_worker"]) and "calculateWTNSBin" in r["sample"] and r["wasm"].endswith(".wasm")


def test_bench_quotes_the_record_without_a_reference_and_says_so(tmp_path):
    bench = _load("bench_for_test", os.path.join(ROOT, "bench.py"))
    r = bench.reference_wasm("compression", str(tmp_path / "no_such_checkout"), 10.0)
    rec = json.load(open(os.path.join(ROOT, "profiles", "wasm_baseline.json")))["circuits"]["compression"]
    assert r["measured_here"] is False and r["value"] == rec["value"] and r["cores"] == rec["cores"] and "no reference checkout" in r["where"]
    assert bench.reference_wasm("nova_bn254_o1", str(tmp_path), 10.0) is None          # no committed WASM run recorded for the circomkit build


@pytest.mark.skipif(not (os.path.isdir(REF) and shutil.which("node")), reason="needs the reference checkout and node (build container)")
def test_live_leg_runs_the_reference_wasm():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    WB = _load("wasm_baseline_for_test", os.path.join(ROOT, "tools", "wasm_baseline.py"))
    assert WB.available(REF)
    r = WB.measure("compression", REF, 1.0, workers=2, n_inputs=8)
    assert r["cores"] == 2 and r["witnesses"] >= 2 and r["wtns_bytes"] == 771052 and 1 < r["per_core"] < 1000
    bench = _load("bench_for_test2", os.path.join(ROOT, "bench.py"))
    live = bench.reference_wasm("compression", REF, 1.0)
    assert live["measured_here"] is True and live["value"] > 1 and "this host" in live["where"]
