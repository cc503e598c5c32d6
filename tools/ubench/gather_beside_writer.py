#!/usr/bin/env python3
"""gather_beside_writer.py — does a kernel that gathers random 64-byte table entries and computes on them (the commit kernels' shape:
768 threads a CU, one gather per ~2 000 vector instructions) keep its speed beside a writer that saturates HBM, when the table fits L2 /
the Infinity Cache / neither?  (r05: would a commit formulation whose tables stay in L2 overlap with the witness kernel?  No: the kernel
takes 13.5 ms alone whatever its table and 20-22 ms beside the writer whatever its table, the writer 7.0 -> 9-10.8 ms, the shader clock
2.39 -> 2.14-2.19 GHz: profiles/r05/overlap/gather_beside_writer.log.)"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
m = importlib.import_module("hot-proofs-blake3-circom_amd")
L = m.lib()
vp = ctypes.c_void_p
L.b3w_place_gather_launch.restype = ctypes.c_int
L.b3w_place_gather_launch.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
L.b3w_place_store_launch.restype = ctypes.c_int
L.b3w_place_store_launch.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, vp]
dev = torch.device("cuda", 0)
ctx = m.Context("nova_vesta", 0)
n = 65536
bodies = ctx.alloc_bodies(n * ctx.body_bytes)
sink = torch.zeros(8, dtype=torch.int32, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ITERS, VALU, GRID = 26, 2000, 256 * 3 * 16                    # 16 rounds of three 256-thread workgroups a CU, 26 gathers a lane, 2 000 multiply-adds a gather


def writer(s):
    assert L.b3w_place_store_launch(bodies.ptr, ctx.body_bytes, n, ctx.body_bytes, 202, s.cuda_stream) == 0      # two persistent store waves a CU


def timed(fa, fb=None, reps=3):
    out = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        e[0].record(sa); sb.wait_event(e[0])
        if fb is not None:
            e[3].record(sb); fb(sb); e[4].record(sb)
        e[1].record(sa); fa(sa); e[2].record(sa)
        if fb is not None:
            sa.wait_event(e[4])
        e[5].record(sa)
        torch.cuda.synchronize()
        out.append((e[0].elapsed_time(e[5]), e[1].elapsed_time(e[2]), e[3].elapsed_time(e[4]) if fb is not None else 0.0))
    out = out[1:]
    return tuple(sorted(x[i] for x in out)[len(out) // 2] for i in range(3))


print(f"writer alone: {timed(writer)[0]:.3f} ms (two persistent store waves a CU over {n} nova bodies)")
for mb in (0.25, 16, 14336):
    nb = int(mb * (1 << 20))
    table = torch.randint(0, 2**31 - 1, (nb // 4,), dtype=torch.int32, device=dev)

    def gather(s, t=table, b=nb):
        assert L.b3w_place_gather_launch(t.data_ptr(), b, ITERS, VALU, GRID, sink.data_ptr(), s.cuda_stream) == 0
    def clock():                                               # GHz of the last gather launch: shader cycles per 100 MHz tick around its loop
        q = sink.cpu().numpy().view(np.uint64)
        return float(q[1]) / max(float(q[2]), 1.0) * 0.1
    alone = timed(gather)[0]
    ck_alone = clock()
    w, a, g = timed(writer, gather)
    print(f"table {mb:8.2f} MiB: gather kernel alone {alone:7.3f} ms at {ck_alone:.2f} GHz; beside the writer: pair {w:7.3f} ms, writer {a:7.3f} ms, "
          f"gather kernel {g:7.3f} ms at {clock():.2f} GHz")
    del table

# the same with the gather kernel's arithmetic cut to a tenth (memory-latency-bound instead of VALU-bound) and to nothing but the gathers
for valu in (200, 0):
    VALU = valu
    for mb in (0.25, 14336):
        nb = int(mb * (1 << 20))
        table = torch.randint(0, 2**31 - 1, (nb // 4,), dtype=torch.int32, device=dev)

        def gather(s, t=table, b=nb, v=valu):
            assert L.b3w_place_gather_launch(t.data_ptr(), b, ITERS * (10 if v else 100), v, GRID, sink.data_ptr(), s.cuda_stream) == 0
        alone = timed(gather)[0]
        w, a, g = timed(writer, gather)
        print(f"{valu:4d} multiply-adds a gather, table {mb:8.2f} MiB: alone {alone:7.3f} ms; beside the writer: pair {w:7.3f} ms, writer {a:7.3f} ms, gather kernel {g:7.3f} ms")
        del table
