"""files/s of the streaming .wtns writer (b3w_batch_write_wtns_ex) next to the D2H copy alone: the hand-off to
`snarkjs groth16 prove` (test/witness_gen.test.ts:47-49), one 771 052-byte file per witness.
    python tools/ubench/wtns_writer_rate.py [n_witnesses] [target_dir]
Target: tmpfs (/dev/shm) by default — a disk would measure the disk.  Writes n files, deletes them, per thread count."""
import ctypes, importlib, os, shutil, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

m = importlib.import_module("hot-proofs-blake3-circom_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
base = sys.argv[2] if len(sys.argv) > 2 else ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
ctx = m.Context("compression", 0)
b = m.Batch(ctx, n)
b.run(m.workloads.config2_compression(n))
body = ctx.body_bytes
print(f"{n} blake3_compression witnesses, {body + 76} bytes per file, target {base} "
      f"({os.cpu_count()} cpus, affinity {len(os.sched_getaffinity(0))})", flush=True)

# D2H alone: the same two-buffer copy without the files (pinned destination, whole batch)
ptr, pitch = b.device_ptr()
host = torch.empty(min(n, 1024) * body, dtype=torch.uint8).pin_memory()
L = ctypes.CDLL("libamdhip64.so")
L.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
k = min(n, 1024)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert L.hipMemcpy(host.data_ptr(), ptr, k * body, 2) == 0
    dt = time.perf_counter() - t0
print(f"D2H alone (one hipMemcpy of {k} bodies into pinned memory): {k * body / dt / 1e9:.1f} GB/s = {k / dt / 1e3:.1f} k witnesses/s", flush=True)

# the filesystem alone: the same files written from a buffer that is already in host memory (no GPU, no copy) by T Python threads
# (os.writev releases the GIL); what the writer can reach at most on this target
import threading
blob = bytes(body)
hdr = ctx.wtns_header()
for threads in (1, 4, 16, 32):
    d = tempfile.mkdtemp(prefix="b3w_raw_", dir=base)
    try:
        def work(t):
            for i in range(t, n, threads):
                fd = os.open(os.path.join(d, f"r{i}.wtns"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                os.writev(fd, [hdr, blob])
                os.close(fd)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t0
        print(f"filesystem alone, {threads:2d} threads: {n / dt / 1e3:7.2f} k files/s = {n * (len(blob) + 76) / dt / 1e9:5.1f} GB/s", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)

for threads in (1, 2, 4, 8, 16, 32):
    d = tempfile.mkdtemp(prefix="b3w_wtns_", dir=base)
    try:
        best = None
        for rep in range(2):
            for f in os.listdir(d):
                os.unlink(os.path.join(d, f))
            t0 = time.perf_counter()
            wrote = b.write_wtns(d, "w", threads=threads)
            dt = time.perf_counter() - t0
            assert wrote == n
            best = dt if best is None else min(best, dt)
        sz = os.path.getsize(os.path.join(d, "w0.wtns"))
        print(f"{threads:2d} writer threads: {n / best / 1e3:7.2f} k files/s = {n * sz / best / 1e9:5.1f} GB/s", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
