#!/usr/bin/env python3
"""power_cases.py — the workload side of the power measurement (round 6; DESIGN.md §8d "one budget" was a hypothesis until now).

Runs each kernel of the fold-shaped pass for `seconds` of steady state and writes, per case, the CLOCK_MONOTONIC window it ran in
(the clock tools/ubench/smi_sampler.cpp stamps its samples with) and the kernel's own rate, so that tools/ubench/power_report.py can
cut the sampler's CSV by case.  The sampler is started by tools/ubench/power_run.sh BEFORE this process touches the GPU.

  idle            nothing enqueued
  witness         nova/Vesta witness kernel, n steps per launch, back to back (HBM-write bound)
  commit          commitments from the records alone (VALU bound)
  both            the two on two streams, commitments enqueued first, the pair joined per iteration (as the chain's GATED mode)
  walk            constraint check (walk + deferred kernel) over the bodies the witness kernel left (HBM-read / issue bound)
  compression     the headline kernel: 4 096 compression witnesses per launch
  stores          the store-only body-stream kernel over the same buffer (no ALU work at all)
  fpmul_1chain / fpmul_2chains   tools/ubench/fpmul29_peak loop: nothing but the commit kernel's 9 x 29-bit Montgomery multiplication, one / two
                  independent chains per lane — the "VALU ceiling" the commit lines are priced against

  python tools/ubench/power_cases.py out.json [seconds=4] [n=65536]
"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out_path = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
t_import = time.monotonic()
import numpy as np
import torch

m = importlib.import_module("hot-proofs-blake3-circom_amd")
K = importlib.import_module("hot-proofs-blake3-circom_amd.synthetic_key")
dev = torch.device("cuda", 0)
props = torch.cuda.get_device_properties(0)
bdf = None
try:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
        bdf = buf.value.decode()
except OSError:
    pass
t_gpu_first_touch = time.monotonic()

ctx = m.Context("nova_vesta", 0)
key = m.CommitKey(ctx, "vesta", K.generators("vesta", ctx.witness_size, seed=b"bench"), fold=True)
recs = m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
bodies = ctx.alloc_bodies(n * ctx.body_bytes)
d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
d_st2 = torch.zeros(n, dtype=torch.int32, device=dev)
d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
viol = torch.zeros(n, dtype=torch.int32, device=dev)
r1cs = m.R1cs(ctx)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

cctx = m.Context("compression", 0)
nc = 4096
crecs = torch.from_numpy(m.workloads.config2_compression(nc).view(np.int32)).to(dev)
cbodies = cctx.alloc_bodies(nc * cctx.body_bytes)


def witness(s):
    ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s.cuda_stream)


def commit(s):
    key.commit_records_device(d_recs.data_ptr(), n, d_pts.data_ptr(), d_st2.data_ptr(), 0, s.cuda_stream)


def walk(s):
    r1cs.check_device(bodies.ptr, n, 0, viol.data_ptr(), 0, s.cuda_stream)


def compression(s):
    for _ in range(16):
        cctx.run_device(crecs.data_ptr(), nc, cbodies.ptr, 0, 0, 0, s.cuda_stream)


def stores(s):
    assert m.lib().b3w_place_store_launch(ctypes.c_void_p(bodies.ptr), ctypes.c_uint64(ctx.body_bytes), ctypes.c_uint32(n),
                                          ctypes.c_uint32(ctx.body_bytes), ctypes.c_int(1), ctypes.c_void_p(s.cuda_stream)) == 0


def both(s):
    e0 = torch.cuda.Event()
    e0.record(sa)
    sb.wait_event(e0)
    commit(sb)
    witness(sa)
    e1 = torch.cuda.Event()
    e1.record(sb)
    sa.wait_event(e1)


def run_case(name, fn, steps_per_iter):
    """enqueue at most two iterations ahead of the device, for `seconds`; returns the window and the rate"""
    if fn is None:
        t0 = time.monotonic(); time.sleep(seconds); t1 = time.monotonic()
        return dict(name=name, t0=t0, t1=t1, iters=0)
    for _ in range(2):
        fn(sa)
    torch.cuda.synchronize()
    evs = []
    t0 = time.monotonic()
    iters = 0
    while time.monotonic() - t0 < seconds:
        fn(sa)
        e = torch.cuda.Event()
        e.record(sa)
        evs.append(e)
        iters += 1
        if len(evs) > 2:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    t1 = time.monotonic()
    ms = (t1 - t0) * 1e3 / iters
    return dict(name=name, t0=t0, t1=t1, iters=iters, ms_per_iter=ms, steps_per_iter=steps_per_iter,
                m_steps_per_s=steps_per_iter / ms / 1e3)


def run_binary(name, argv):
    """another process's kernels for `seconds` (the field-multiplication loop of tools/ubench/fpmul29_peak.hip): the window it ran in"""
    import subprocess
    t0 = time.monotonic()
    r = subprocess.run(argv, capture_output=True, text=True, timeout=seconds + 60)
    t1 = time.monotonic()
    rates = [float(ln.split(":")[1].split()[0]) for ln in r.stdout.splitlines() if "G field mul/s" in ln]
    return dict(name=name, t0=t0 + 0.5, t1=t1, iters=len(rates), g_field_mul_per_s=(sorted(rates)[len(rates) // 2] if rates else None), rc=r.returncode)


witness(sa); torch.cuda.synchronize()                         # the walk case reads valid bodies
cases = []
plan = [("idle", None, 0), ("witness", witness, n), ("commit", commit, n), ("both", both, n), ("walk", walk, n),
        ("compression", compression, 16 * nc), ("stores", stores, n), ("idle_after", None, 0)]
for name, fn, steps in plan:
    cases.append(run_case(name, fn, steps))
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in cases[-1].items() if k != "name"}, flush=True)
    time.sleep(1.0)                                           # a visible gap between the cases in the power trace
fp = os.path.join(ROOT, "tools", "ubench", "fpmul29_peak")
if os.path.exists(fp):                                        # the commit kernel's VALU ceiling: is it arithmetic or power?
    for name, chains in (("fpmul_1chain", "1"), ("fpmul_2chains", "2")):
        cases.append(run_binary(name, [fp, "loop", str(seconds), chains]))
        print(name, {k: v for k, v in cases[-1].items() if k != "name"}, flush=True)
        time.sleep(1.0)
json.dump(dict(device=props.name, bdf=bdf, n=n, seconds=seconds, t_import=t_import, t_gpu_first_touch=t_gpu_first_touch,
               placement=bodies.placement, key_window=key.window, cases=cases), open(out_path, "w"), indent=1)
