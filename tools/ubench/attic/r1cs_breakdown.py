"""r1cs_breakdown.py — where does the constraint-check kernel spend its time?  Times b3w_r1cs_check_device on 4 096 valid
blake3_compression bodies with sub-systems of the derived R1CS: all rows, only booleanity rows, only xor rows, only the linear
(recomposition) rows, everything but the linear rows."""
import importlib, os, struct, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import r1cs_ref as R
m = importlib.import_module("hot-proofs-blake3-circom_amd")
img = R.read_image()
sys_ = R.parse(img)
P = sys_["prime"]


def image(cons):
    lc = lambda d: struct.pack("<I", len(d)) + b"".join(struct.pack("<I", w) + f.to_bytes(32, "little") for w, f in d.items())
    header = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIIIQI", sys_["n_wires"], 16, 0, 28, sys_["n_labels"], len(cons))
    body = b"".join(lc(a) + lc(b) + lc(c) for a, b, c in cons)
    blob = b"r1cs" + struct.pack("<II", 1, 2)
    for typ, sec in ((1, header), (2, body)):
        blob += struct.pack("<IQ", typ, len(sec)) + sec
    return blob


cons = sys_["constraints"]
kind = lambda r: "linear" if not r[0] else "bool" if not r[2] else "xor"
subsets = {"all": cons, "bool only": [r for r in cons if kind(r) == "bool"], "xor only": [r for r in cons if kind(r) == "xor"],
           "linear only": [r for r in cons if kind(r) == "linear"], "no linear": [r for r in cons if kind(r) != "linear"]}
ctx = m.Context("compression", 0)
n = 4096
recs = m.workloads.config2_compression(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
bodies = ctx.alloc_bodies(n * ctx.body_bytes)
s = torch.cuda.current_stream().cuda_stream
ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, 0, 0, s)
viol = torch.zeros(n, dtype=torch.int32, device="cuda")
for name, sub in subsets.items():
    r = m.R1cs(ctx, image(sub))
    for _ in range(2):
        r.check_device(bodies.ptr, n, 0, viol.data_ptr(), 0, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r.check_device(bodies.ptr, n, 0, viol.data_ptr(), 0, s)
    e1.record()
    torch.cuda.synchronize()
    assert int(viol.abs().sum().item()) == 0
    print(f"{name:12s} rows {len(sub):6d} terms {sum(len(a) + len(b) + len(c) for a, b, c in sub):7d}: {e0.elapsed_time(e1) / 5:.2f} ms", flush=True)
    r.close()
