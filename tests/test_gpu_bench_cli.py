"""bench.py's command line on a GPU: the default contract line and the chained-pass modes, on tiny workloads
(the numbers are not looked at; the line must carry every field of the contract and the run must verify itself)."""
import json
import os
import subprocess
import sys

import pytest

import b3w_testlib as T

pytestmark = pytest.mark.gpu


def _bench(*args, env=None):
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=T.ROOT,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                    # ONE JSON line, from rank 0
    return json.loads(lines[0])


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
    assert d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["reference_wasm"]["measured_here"] is False
    # value, ms_per_step and the launch count hang together
    c = d["config"]
    # what placement cost this process (round-3 verdict #6): seconds and GiB walked by the search, the real-kernel check, the limit
    # (r05: bench.py sets none of the allocator's knobs — the library's own defaults: 5 s)
    assert c["placement_search_limit_s"] == 5.0 and c["placement_search_s"] >= 0 and c["placement_alloc_s"] >= c["placement_search_s"]
    assert set(c["placement_search_breakdown_s"]) == {"hipMemCreate", "map", "probes", "release"}
    # r05: the same kernel on a plain hipMalloc buffer and the pure-store ceilings of both buffers, measured in the same run
    r = d["roofline"]
    assert r["plain"]["kernel_ms"] > 0 and abs(r["plain"]["frac"] - r["plain"]["achieved"] / r["peak"]) < 1e-9
    # r06: the plain buffer's launch shape is autotuned ON the plain buffer and named; the ceilings include the paced shapes of the sweep
    assert isinstance(r["plain"]["kernel_variant"], str) and r["plain"]["kernel_variant"].split()[0] in ("fused", "sweep", "fill-ordered")
    assert set(r["store_ceiling"]["placed"]) == set(r["store_ceiling"]["plain"]) == {"streams_w4", "streams_w8", "fill", "paced_persistent_w4x512",
                                                                                      "paced_persistent_w8x512", "paced_streams_w8",
                                                                                      "paced_region_fill_sleep", "paced_region_fill_valu"}
    assert min(list(r["store_ceiling"]["placed"].values()) + list(r["store_ceiling"]["plain"].values())) > 100
    assert abs(r["of_measured_ceiling"] - r["achieved"] / max(r["store_ceiling"]["placed"].values())) < 1e-9
    # the reference WASM's all-core rate: recorded in the build container (the reference cannot travel to this box), and the line says so
    w = d["cpu_baseline"]["reference_wasm"]
    assert w["cores"] >= 1 and w["value"] > 1 and w["unit"] == "witnesses/s" and "wasm_baseline.json" in w["where"] and "calculateWTNSBin" in w["sample"]
    assert c["placement_search_gib_walked"] >= 0 and c["placement_search_timeouts"] in (0, 1) and c["exchange_impl"] == "none"
    assert c["witnesses_per_step"] == 512 * c["launches_per_step"]
    assert abs(d["value"] - c["witnesses_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["launches_timed"] == 3 * c["launches_per_step"] and d["roofline"]["kernel_ms"] * c["launches_per_step"] <= d["ms_per_step"] * 1.001
    # r05: the HBM traffic of a launch is measured by the invocation itself (two rocprofv3 --pmc child passes before it touches the GPU)
    tm = r["traffic_measured"]
    assert tm is not None and "measured by this invocation" in r["traffic_source"], r["traffic_source"]
    assert abs(r["traffic"] - (tm["write_bytes"] + tm["fetch_bytes_x2"])) < 1.0 and tm["write_bytes"] > 100 * tm["fetch_bytes_x2"] >= 0
    assert 0.995 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.03, (r["traffic"], r["algorithmic_bytes_per_launch"])


def test_nova_line_is_labelled_config3():
    d = _bench("--circuit", "nova_vesta", "--batch", "512", "--steps", "2", "--warmup", "1", "--inner", "2", "--cpu-seconds", "0", "--placement", "plain", "--traffic", "quoted")
    assert d["roofline"]["traffic_measured"] is None and "--traffic quoted" in (d["roofline"]["traffic_source"] or "--traffic quoted")
    assert d["config"]["workload"].startswith("config3") and "Vesta" in d["config"]["workload"] and d["config"]["launches_per_step"] == 2


def test_gpus_2_launches_itself():
    """`python bench.py --gpus 2` (the driver's command shape, no launcher): the script spawns one child per rank before
    touching the GPU.  On the one-GPU test box the two ranks share the card and rehearse the exchange over gloo."""
    d = _bench("--gpus", "2", "--batch", "256", "--steps", "2", "--warmup", "1", "--inner", "3", env={"B3W_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "2 ranks" in d["config"]["exchange"] and len(d["config"]["placement_per_rank"]) == 2
    assert d["config"]["witnesses_per_step"] == 2 * 256 * 3 and "cpu_baseline" not in d
    # (the default exchange mode; the other two: test_gpus_2_exchange_modes)
    c, r = d["config"], d["roofline"]
    assert c["exchange_mode"] == "every" and c["exchange"].startswith("--exchange every")
    assert len(r["kernel_ms_per_rank"]) == 2 and r["kernel_ms_min"] <= r["kernel_ms_max"] == r["kernel_ms"]


@pytest.mark.parametrize("mode,impl", [("last", "native"), ("none", "torch")])
def test_gpus_2_exchange_modes(mode, impl):
    """--exchange splits an N > 1 number into kernel and collective: all three modes run, verify themselves, and say what they did;
    every rank's own kernel time and the collective's own time are in the line.  --exchange-impl native: the gather goes through the
    C-ABI's b3w_comm (here its host shared-memory transport: two ranks on one GPU) instead of torch.distributed."""
    d = _bench("--gpus", "2", "--batch", "256", "--steps", "2", "--warmup", "1", "--inner", "3", "--exchange", mode, "--placement", "plain",
               "--exchange-impl", impl, env={"B3W_DIST_BACKEND": "gloo"})
    c, r = d["config"], d["roofline"]
    assert d["n_gpus"] == 2 and c["exchange_mode"] == mode and c["exchange"].startswith(f"--exchange {mode}")
    assert c["exchange_impl"] == ("native b3w_comm (host)" if impl == "native" else "torch.distributed")
    assert len(c["exchange_ms_per_rank"]["public_outputs"]) == 2 and min(c["exchange_ms_per_rank"]["public_outputs"]) > 0
    assert len(r["kernel_ms_per_rank"]) == 2 and r["kernel_ms_min"] <= r["kernel_ms_max"] == r["kernel_ms"]
    assert len(c["devices_per_rank"]) == 2 and all(x.startswith("cuda:") for x in c["devices_per_rank"])


def test_a_rank_hung_before_the_rendezvous_fails_the_run_quickly():
    """one rank never reaches init_process_group (GPU initialised, then stuck): the launcher's watchdog ends the run with status
    124 well inside the driver's time limit instead of waiting for it (--launch-timeout 8: seconds from the first rank's torch import)"""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--batch", "256", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "8", "--placement", "plain"], capture_output=True, text=True, timeout=400, cwd=T.ROOT,
                       env=dict(os.environ, B3W_DIST_BACKEND="gloo", B3W_BENCH_TEST_HANG_RANK="1", B3W_BENCH_TEST_HANG_AT="rendezvous"))
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "ranks [0, 1] have not passed rendezvous" in r.stderr or "ranks [1] have not passed rendezvous" in r.stderr
    assert r.stdout.strip() == "" and time.monotonic() - t0 < 150


def test_gpus_2_chain_launches_itself():
    """config 4's shape on two ranks with the NATIVE exchange (b3w_chain_run_parents_sharded + b3w_chain_allgather_hout through a
    host-transport b3w_comm); the torch path: test_gpus_2_chain_without_the_h_out_gather"""
    d = _bench("--gpus", "2", "--workload", "chain", "--preimage-mib", "0.25", "--steps", "1", "--warmup", "1", "--exchange-impl", "native",
               "--placement", "plain", env={"B3W_DIST_BACKEND": "gloo"})
    assert d["config"]["exchange_impl"] == "native b3w_comm (host)"
    ex = d["config"]["exchange_ms_per_rank"]
    assert len(ex["chunk_cvs"]) == 2 and len(ex["h_out"]) == 2 and min(ex["chunk_cvs"] + ex["h_out"]) > 0
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["n_chunks"] == 256
    assert "2 ranks" in d["config"]["exchange"] and len(d["config"]["placement_per_rank"]) == 2
    # 256 chunks x 16 leaf steps + 8 parent steps per chunk path, summed over both ranks
    assert "-> 6144 nova steps" in d["config"]["workload"]
    # BASELINE config 4's exchange, inside the timed pass: every step's h_out and the chunk chaining values
    assert d["config"]["exchange"].startswith("all_gather of 4096 x 8 u32 h_out (+ 2048 x 8 of the parent steps) + all_gather of 256 x 8 u32 chunk chaining values")
    assert len(d["config"]["pass_ms_per_rank"]) == 2


def test_gpus_2_chain_without_the_h_out_gather():
    d = _bench("--gpus", "2", "--workload", "chain", "--preimage-mib", "0.25", "--steps", "1", "--warmup", "1", "--exchange", "none",
               "--placement", "plain", env={"B3W_DIST_BACKEND": "gloo"})
    assert d["config"]["exchange"].startswith("--exchange none: all_gather of 256 x 8 u32 chunk chaining values") and d["config"]["exchange_mode"] == "none"
    ex = d["config"]["exchange_ms_per_rank"]
    assert d["config"]["exchange_impl"] == "torch.distributed" and min(ex["chunk_cvs"]) > 0 and ex["h_out"] == [0.0, 0.0]


@pytest.mark.parametrize("consumer", ["none", "commit-only", "check+commit"])
def test_chain_workload_with_each_consumer(consumer):
    """the chain line's roofline is its dominant kernel's (round-3 verdict #3): HBM bytes written (+ read back by the check) without
    a commit consumer, the commit kernel's field multiplications against the multiplication-only ceiling with one; cpu_baseline on
    the N = 1 line.  (commit and check alone: tools/jobs/r04 runs; here one of each roofline kind)"""
    # (placement: the default, placed ring only for the plain pass — the allocator's search and claim check are seconds per process)
    d = _bench("--workload", "chain", "--preimage-mib", "0.25", "--steps", "1", "--warmup", "1", "--consumer", consumer, "--cpu-seconds", "1",
               *([] if consumer == "none" else ["--placement", "plain"]))
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["frac"] > 0.001, r
    if consumer == "none":
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["algorithmic_bytes_per_step"] == 745312 + 128
    else:
        assert r["bound"] == "valu" and r["unit"] == "G field mul/s" and 100 < r["peak"] < 300
        assert 1000 < r["point_additions_per_step"] < 5000 and r["field_multiplications_per_step"] > 10 * r["point_additions_per_step"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and "first_pass_s" in d["config"]
    assert d["scaling"] == "strong" and d["value"] > 0 and d["config"]["n_chunks"] == 256 and d["config"]["path_len"] == 8
    assert ("no bodies" in d["config"]["consumer"]) == (consumer == "commit-only")
    assert ("beside the bodies" in d["config"]["consumer"]) == (consumer == "check+commit")
    assert ("constraint check" in d["config"]["consumer"]) == ("check" in consumer)
    if "check" in consumer:
        assert "inside the timed pass: 0 of 23744 constraints violated by any of 6144 step witnesses" in d["config"]["verification"]
    assert d["config"]["commit_overlap"] == ("auto" if consumer == "check+commit" else None)
