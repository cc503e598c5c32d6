"""lds_pad_small.py — does limiting the workgroups per CU (unused dynamic LDS) help mid-size batches?  The batch curve
(profiles/r02/batch_curve.json) has 512 witnesses SLOWER than 256 with one body per wave: if the dispatcher piles several
waves on some CUs while others idle, a pad that lets only k workgroups fit per CU spreads them evenly.
One subprocess per pad (B3W_LDS_PAD is read once per process)."""
import json, os, subprocess, sys
CHILD = r'''
import importlib, os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
s = torch.cuda.current_stream().cuda_stream
out = {}
for circuit in ("compression", "nova_vesta"):
    ctx = m.Context(circuit, 0)
    nmax = 2048
    recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    d_st = torch.zeros(nmax, dtype=torch.int32, device="cuda")
    buf = ctx.alloc_bodies(nmax * ctx.body_bytes)
    for n in (256, 384, 512, 768, 1024, 1536, 2048):
        for _ in range(3):
            ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), s)
        ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), s, 40) for _ in range(3))
        out[f"{circuit}:{n}"] = n / ms / 1e3
    buf.free(); ctx.close()
print(json.dumps(out))
'''
rows = {}
for variant in ("1", "2"):
    for pad in ("0", "36000", "49000", "60000"):
        env = dict(os.environ, B3W_LDS_PAD=pad, B3W_VARIANT=variant, B3W_PLACE_CHECK="0")
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
        if r.returncode:
            print("failed", variant, pad, r.stderr[-500:]); continue
        rows[f"variant {variant} pad {pad}"] = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"variant {variant} pad {pad:>6}: " + "  ".join(f"{k.split(':')[0][:4]}{k.split(':')[1]:>5}={v:5.2f}" for k, v in rows[f"variant {variant} pad {pad}"].items()), flush=True)
json.dump(rows, open(os.path.join(os.getcwd(), "gpurun_out", "lds_pad_small.json"), "w"), indent=1)
