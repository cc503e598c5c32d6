"""single_latency.py — latency of ONE witness through the WitnessCalculator mirror (calculateWTNSBin): canonical
inputs (batch kernel with n = 1) and non-canonical ones (exact field-element kernel)."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import b3w_testlib as T
m = importlib.import_module("hot-proofs-blake3-circom_amd")
for circuit in ("compression", "nova_vesta"):
    g = T.golden(circuit)
    ok = [c for c in g["cases"] if "error" not in c]
    canon = next(c for c in ok if T.is_canonical_u32(circuit, c["input"]))
    wild = next((c for c in ok if not T.is_canonical_u32(circuit, c["input"])), None)
    wc = m.builder(circuit)
    for name, case in (("canonical", canon), ("field-element", wild)):
        if case is None: continue
        for _ in range(3): img = wc.calculateWTNSBin(case["input"], 0)
        assert T.sha256(img) == case["wtns_sha256"]
        t0 = time.perf_counter(); k = 20
        for _ in range(k): wc.calculateWTNSBin(case["input"], 0)
        dt = (time.perf_counter() - t0) / k
        t0 = time.perf_counter()
        for _ in range(k): wc.calculateBinWitness(case["input"], 0)
        db = (time.perf_counter() - t0) / k
        print(f"{circuit} {name}: calculateWTNSBin {dt*1e3:.2f} ms, calculateBinWitness {db*1e3:.2f} ms per witness")
