"""r1cs_fuzz.py <out.npz> [bodies per circuit] — verdicts of the constraint check on tampered witnesses of all four circuits, for a
comparison ACROSS formulations: run it once per B3W_R1CS_GATHER setting (unset = walk, 4 = stream, 1 = gather; the switch is read
once per process) and compare the files (tools/ubench/r1cs_fuzz_compare.py).  Every body of a batch of valid witnesses gets one to
three slots overwritten: a neighbour's value, 0, 1, 2, p - 1, p, 2^32, 2^63, 2^64 - 1, a field inverse of another body, or random
bits of random width — seeded, so every run tampers alike."""
import importlib, os, random, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
out_path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
s = torch.cuda.current_stream().cuda_stream
res = {}
for circuit in ("compression", "nova_bn254_o1", "nova_bn254", "nova_vesta"):
    ctx = m.Context(circuit, 0)
    r = m.R1cs(ctx)
    p = int(ctx.prime)
    recs = m.workloads.config2_compression(n, first=9) if circuit == "compression" else m.workloads.config3_nova(n, first=9)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device="cuda")
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
    torch.cuda.synchronize()
    host = bodies.cpu().numpy().reshape(n, ctx.witness_size, 32)
    rng = random.Random(1234)
    wide = [w for w in range(ctx.witness_size) if int.from_bytes(host[0, w].tobytes(), "little") >= 1 << 64]
    log = []
    for b in range(1, n):                                    # body 0 stays valid
        for _ in range(rng.choice((1, 1, 2, 3))):
            w = rng.randrange(ctx.witness_size) if not wide or rng.random() < 0.8 else rng.choice(wide)
            kind = rng.randrange(11)
            if kind == 0: val = int.from_bytes(host[b, (w + 1) % ctx.witness_size].tobytes(), "little")
            elif kind == 1: val = 0
            elif kind == 2: val = 1
            elif kind == 3: val = 2
            elif kind == 4: val = p - 1 - rng.randrange(3)
            elif kind == 5: val = p + rng.randrange(2)
            elif kind == 6: val = 1 << 32
            elif kind == 7: val = 1 << 63
            elif kind == 8: val = (1 << 64) - 1
            elif kind == 9 and wide: val = int.from_bytes(host[rng.randrange(n), rng.choice(wide)].tobytes(), "little")
            else: val = rng.getrandbits(rng.choice((8, 33, 64, 128, 250, 256)))
            host[b, w] = np.frombuffer(val.to_bytes(32, "little"), dtype=np.uint8)
            log.append((b, w))
    d = torch.from_numpy(host.reshape(n, -1)).cuda()
    viol = torch.full((n,), 99, dtype=torch.int32, device="cuda")
    first = torch.zeros((n,), dtype=torch.int32, device="cuda")
    r.check_device(d.data_ptr(), n, 0, viol.data_ptr(), first.data_ptr(), s)
    torch.cuda.synchronize()
    res[circuit + "_viol"] = viol.cpu().numpy().view(np.uint32)
    res[circuit + "_first"] = first.cpu().numpy().view(np.uint32)
    nz = int(np.count_nonzero(res[circuit + "_viol"]))
    print(f"{circuit}: {n} bodies, {len(log)} slots overwritten, {nz} bodies with violated rows, body 0: {int(res[circuit + '_viol'][0])}", flush=True)
    r.close(); ctx.close()
np.savez(out_path, **res)
