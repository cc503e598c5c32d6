"""walk_ablate.py build <mask> [<mask> ...]   (build container or GPU box: hipcc only)
   walk_ablate.py time  <circuit> <mask> [<mask> ...]   (GPU box)
Where a constraint check's time goes, by leaving parts of the walk kernel OUT (wrong verdicts, right amount of everything else):
`build` compiles csrc/b3w_r1cs_walk.hip with -DB3W_WALK_ABLATE=<mask> (see the kernel: 1 truth-table runs, 2 verdicts, 4 entries,
8 exports, 32 scratch blocks, 2048 wide records, 4096 the records' stores) and links it with the product's other objects into hot-proofs-blake3-circom_amd/build/ablate/;
`time` loads each of those libraries in a child process and times 20 checks of 4 096 bodies with HIP events (walk + deferred kernel;
a variant that makes no wide records shortens the deferred kernel too); `one <circuit>` is what tools/jobs/r05/s26_ab.sh runs under
rocprofv3 --kernel-trace, kernel by kernel (tools/ubench/walk_trace_median.py: medians of the last 20 launches — between two
processes on one box the walk kernel differs by +-5 us, on the circomkit build by up to 25: compare medians of alternating runs)."""
import importlib, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "hot-proofs-blake3-circom_amd")
OUT = os.path.join(PKG, "build", "ablate")


def lib_of(mask):
    return os.path.join(OUT, f"libb3wit_a{mask}.so")


def build(masks):
    sys.path.insert(0, ROOT)
    b = importlib.import_module("hot-proofs-blake3-circom_amd.build")
    b.build_lib()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", f) for f in sorted(os.listdir(os.path.join(PKG, "build"))) if f.endswith(".o") and f != "b3w_r1cs_walk.hip.o"]
    for m in masks:
        obj = os.path.join(OUT, f"walk_a{m}.o")
        extra = [x for x in os.environ.get("B3W_ABLATE_FLAGS", "").split() if x]
        subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", f"-DB3W_WALK_ABLATE={m}", *extra, "-c",
                               os.path.join(PKG, "csrc", "b3w_r1cs_walk.hip"), "-o", obj])
        subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib_of(m), *objs, obj, "-ldl", "-lrt", "-lpthread"])
        print("built", lib_of(m), flush=True)


def time_one(circuit, n=4096, reps=20):
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    m = importlib.import_module("hot-proofs-blake3-circom_amd")
    s = torch.cuda.current_stream().cuda_stream
    ctx = m.Context(circuit, 0)
    r = m.R1cs(ctx)
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device="cuda")
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
    viol = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(8):
        r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"{circuit} {os.path.basename(os.environ.get('B3WIT_LIB', 'product'))}: median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f}  (walk + deferred, {n} bodies; violations {int((viol != 0).sum())})", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build([int(x) for x in sys.argv[2:]])
    elif sys.argv[1] == "one":
        time_one(sys.argv[2])
    else:
        for rnd in range(2):                                  # (twice round the variants: the chip's clock drifts over a minute)
            for mk in sys.argv[3:]:
                env = dict(os.environ)
                if mk != "product":
                    env["B3WIT_LIB"] = lib_of(int(mk))
                subprocess.run([sys.executable, os.path.abspath(__file__), "one", sys.argv[2]], env=env, timeout=300)
