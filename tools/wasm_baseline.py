#!/usr/bin/env python3
"""The reference's WASM witness generator on all host cores — the CPU number BASELINE.json asks to have beside the GPU's
(/root/reference/test/witness_gen.test.ts:43-50 is the only timing the reference has).

Writes the inputs of BASELINE configs 2 (blake3_compression, LCG(6429 + i)) and 3 (blake3_nova over Vesta and BN254) as circom input
objects, runs tools/wasm_baseline.js — one `node` process per core looping the reference's own calculateWTNSBin — for each circuit,
and records the result in profiles/wasm_baseline.json, which bench.py quotes (`cpu_baseline.reference_wasm`) when the reference is
not present where it runs.  The reference is loaded from --reference-dir at run time; nothing of it is copied.

  python tools/wasm_baseline.py [--reference-dir /root/reference] [--seconds 30] [--workers N] [--circuits compression,nova_vesta,nova_bn254]
  python tools/wasm_baseline.py --print-only ...      (no file written: bench.py's live leg)
"""
import argparse, importlib, json, os, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the committed builds of the reference, relative to the reference checkout
WASM = {"compression": "build/blake3_compression/blake3_compression_js/blake3_compression.wasm",
        "nova_vesta": "build/blake3_nova_pasta_js/blake3_nova_pasta.wasm",
        "nova_bn254": "build/blake3_nova_js/blake3_nova.wasm"}


def available(reference_dir):
    return (shutil.which("node") is not None and os.path.exists(os.path.join(reference_dir, "blake3_nova_js", "witness_calculator.js"))
            and any(os.path.exists(os.path.join(reference_dir, w)) for w in WASM.values()))


def measure(circuit, reference_dir, seconds, workers=0, n_inputs=256):
    """One circuit: the JSON object tools/wasm_baseline.js prints, plus what the sample was."""
    W = importlib.import_module("hot-proofs-blake3-circom_amd.workloads")
    recs = W.config2_compression(n_inputs) if circuit == "compression" else W.config3_nova(n_inputs)
    keys = W.COMPRESSION_KEYS if circuit == "compression" else W.NOVA_KEYS
    with tempfile.TemporaryDirectory(prefix="b3w_wasm_") as d:
        inp = os.path.join(d, "inputs.json")
        json.dump([W.record_to_input(r, keys) for r in recs], open(inp, "w"))
        cmd = ["node", os.path.join(ROOT, "tools", "wasm_baseline.js"), "--reference-dir", reference_dir, "--wasm", os.path.join(reference_dir, WASM[circuit]),
               "--inputs", inp, "--seconds", str(seconds)] + (["--workers", str(workers)] if workers else [])
        out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, text=True).stdout
    doc = json.loads(out.strip().splitlines()[-1])
    doc["wasm"] = WASM[circuit]
    doc["circuit"] = circuit
    doc["sample"] = (f"{doc['witnesses']} witnesses in {doc['seconds']:.1f} s: {doc['cores']} node processes, each looping calculateWTNSBin over the first "
                     f"{n_inputs} inputs of BASELINE config {'2' if circuit == 'compression' else '3'} (LCG(6429 + i)), a different input every call")
    return doc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference-dir", default="/root/reference")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--workers", type=int, default=0, help="node processes (0 = one per usable core)")
    ap.add_argument("--circuits", default="compression,nova_vesta,nova_bn254")
    ap.add_argument("--print-only", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "wasm_baseline.json"))
    args = ap.parse_args()
    if not available(args.reference_dir):
        raise SystemExit(f"wasm_baseline: no node, or no reference checkout at {args.reference_dir}")
    doc = {"measured": time.strftime("%Y-%m-%d"), "where": "build container (no GPU here; the reference cannot travel to the GPU box)",
           "tool": "tools/wasm_baseline.py -> tools/wasm_baseline.js", "seconds_per_circuit": args.seconds, "circuits": {}}
    for c in args.circuits.split(","):
        doc["circuits"][c] = measure(c, args.reference_dir, args.seconds, args.workers)
        r = doc["circuits"][c]
        print(f"{c}: {r['value']:.1f} witnesses/s on {r['cores']} cores ({r['per_core']:.2f} per core), {r['cpu']}, node {r['node']}", file=sys.stderr)
    if args.print_only:
        print(json.dumps(doc))
    else:
        json.dump(doc, open(args.out, "w"), indent=1)
        print(f"wrote {args.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
