#!/usr/bin/env python3
"""scale_day1.py — first contact with a real multi-GPU node: the runs the driver's SCALE step makes, in the order that finds a plumbing
fault soonest, every line held against what has never run anywhere yet (an RCCL communicator of more than one rank, xGMI).

For N in --gpus-list (default 1 2 4 8):
  batch     python bench.py --gpus N                                      (config 2: weak scaling, all-gather of the public outputs)
  chain     python bench.py --gpus N --workload chain --preimage-mib 1    --exchange-impl native   (config 4)
  stream    python bench.py --gpus N --workload chain --preimage-mib M    --exchange-impl native   (config 5: M = 1024, --quick: 64)
and per run, from the JSON line: n_gpus = N; one device per rank and all distinct under RCCL (bench.py refuses otherwise: checked again here);
every rank's body-buffer placement; kernel_ms_per_rank and exchange_ms_per_rank complete (N values each) and the spread between ranks;
the cross-rank checks bench.py makes at the end of every N > 1 run (batch: each rank recomputes rows of every other rank's gathered
block; chain: every rank's position-weighted checksum of the gathered h_out and the root) — a failed one ends bench.py with a
non-zero status, which ends this script.  Prints ONE SCALE-shaped JSON object per (workload, N) and a table at the end; the first
failure stops the run with the child's stderr tail.

A box with ONE GPU rehearses the plumbing: B3W_DIST_BACKEND=gloo python tools/scale_day1.py --gpus-list 1 2 4 --quick --rehearsal
(several ranks share the card over the host transport; the rates are not scaling numbers and the script says so).
"""
import argparse, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, timeout):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(f"scale_day1: {' '.join(cmd)} ended with status {r.returncode}\n---- stderr tail ----\n{r.stderr[-3000:]}\n")
        raise SystemExit(1)
    return json.loads(lines[-1]), time.time() - t0


def check(line, n, what, rehearsal):
    problems = []
    cfg, roof = line["config"], line.get("roofline", {})
    if line["n_gpus"] != n:
        problems.append(f"n_gpus {line['n_gpus']} != {n}")
    devs = cfg.get("devices_per_rank", [])
    if len(devs) != n:
        problems.append(f"devices_per_rank has {len(devs)} entries")
    idents = [d.split("=", 1)[1] if "=" in d else d for d in devs]
    if n > 1 and not rehearsal and len(set(idents)) != n:
        problems.append(f"ranks share a GPU: {devs}")
    places = cfg.get("placement_per_rank") or [cfg.get("placement")]
    if len(places) != n:
        problems.append(f"placement_per_rank has {len(places)} entries")
    if any(p not in ("mixed", "interleaved") for p in places):
        note = f"a rank's body buffer is plain: {places} (placement_search_s / _timeouts in the line say why)"
        if rehearsal:                                          # (several ranks searching ONE card's memory at once: expected, not a fault)
            sys.stderr.write(f"scale_day1: note, {what} at {n}: {note}\n")
        else:
            problems.append(note)
    km = roof.get("kernel_ms_per_rank") or cfg.get("pass_ms_per_rank") or []          # (chain lines: the pass of every rank)
    if isinstance(km, dict):
        km = km.get("all") or [km.get("min"), km.get("max")]
    if len(km) != n:
        problems.append(f"{len(km)} per-rank kernel / pass times for {n} ranks")
    if n > 1:
        ex = cfg.get("exchange_ms_per_rank")
        if not ex:
            problems.append("no exchange_ms_per_rank")
        impl = cfg.get("exchange_impl", "")
        if what != "batch" and "native" not in impl:
            problems.append(f"exchange_impl is {impl!r}, asked for native")
        if not rehearsal and what != "batch" and "rccl" not in impl:
            problems.append(f"the native exchange did not run over RCCL: {impl!r}")
        if cfg.get("comm_size") not in (None, n):
            problems.append(f"communicator size {cfg.get('comm_size')} != {n}")
    return problems, dict(workload=what, n_gpus=n, metric=line["metric"], value=line["value"], unit=line["unit"], ms_per_step=line["ms_per_step"],
                          scaling=line.get("scaling"), kernel_ms_per_rank=km, exchange_ms_per_rank=cfg.get("exchange_ms_per_rank"),
                          exchange_impl=cfg.get("exchange_impl"), comm_size=cfg.get("comm_size"), devices_per_rank=devs, placement_per_rank=places,
                          roofline_frac=roof.get("frac"), first_pass_s=cfg.get("first_pass_s"), problems=problems)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus-list", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--quick", action="store_true", help="64 MiB instead of 1 GiB for the streamed pass, fewer steps")
    ap.add_argument("--rehearsal", action="store_true", help="ranks may share a GPU (B3W_DIST_BACKEND=gloo): plumbing only, no RCCL, no rates")
    ap.add_argument("--timeout", type=float, default=900.0)
    ap.add_argument("--out", default=None, help="also write the objects to this file, one per line")
    args = ap.parse_args()
    if args.rehearsal and os.environ.get("B3W_DIST_BACKEND") != "gloo":
        raise SystemExit("scale_day1: --rehearsal wants B3W_DIST_BACKEND=gloo (several ranks on one card)")
    steps = ["--steps", "3", "--warmup", "1"] if args.quick else []
    plans = [("batch", steps), ("chain", ["--workload", "chain", "--preimage-mib", "1", "--exchange-impl", "native"] + steps),
             ("stream", ["--workload", "chain", "--preimage-mib", "64" if args.quick else "1024", "--exchange-impl", "native"] + steps)]
    rows, failed = [], False
    sink = open(args.out, "w") if args.out else None
    for n in args.gpus_list:
        for what, extra in plans:
            line, secs = run_bench(["--gpus", str(n), "--cpu-seconds", "0"] + extra, args.timeout)
            problems, row = check(line, n, what, args.rehearsal)
            row["wall_s"] = round(secs, 1)
            row["rehearsal"] = args.rehearsal
            rows.append(row)
            print(json.dumps(row), flush=True)
            if sink:
                sink.write(json.dumps(row) + "\n"); sink.flush()
            if problems:
                failed = True
                sys.stderr.write(f"scale_day1: {what} at {n} GPUs: " + "; ".join(problems) + "\n")
    base = {r["workload"]: r["value"] for r in rows if r["n_gpus"] == args.gpus_list[0]}
    sys.stderr.write(f"\n{'workload':8s} {'N':>2s} {'value':>14s} {'x of N=' + str(args.gpus_list[0]):>10s} {'kernel ms min..max':>20s}  exchange\n")
    for r in rows:
        km = [k for k in (r["kernel_ms_per_rank"] or []) if isinstance(k, (int, float))]
        sys.stderr.write(f"{r['workload']:8s} {r['n_gpus']:2d} {r['value']:14.0f} {r['value'] / base[r['workload']]:10.2f} "
                         f"{(min(km) if km else 0):9.3f}..{(max(km) if km else 0):<9.3f}  {r['exchange_impl']}\n")
    if args.rehearsal:
        sys.stderr.write("(rehearsal: the ranks shared one GPU over the host transport — these are not scaling numbers)\n")
    raise SystemExit(1 if failed else 0)


if __name__ == "__main__":
    main()
