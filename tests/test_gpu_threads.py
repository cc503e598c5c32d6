"""include/b3wit.h: "ctx objects are not thread-safe, distinct ctxs are".  Four host threads, each with its own contexts and its
own stream, run batches (sliced and whole-body launch shapes), placed allocations, the constraint check, the tamper check and
single witnesses through the calculator surface at the same time (ctypes drops the GIL inside the library); every result is
compared with the oracle."""
import threading

import numpy as np
import pytest

import b3w_testlib as T

pytestmark = pytest.mark.gpu


def test_distinct_contexts_on_concurrent_host_threads():
    import torch
    m = T.pkg()
    W = T.workloads()
    dev = torch.device("cuda:0")
    jobs = [("compression", 300, 11), ("nova_vesta", 200, 12), ("compression", 3000, 13), ("nova_bn254_o1", 150, 14)]
    wants = {}
    for circuit, n, first in jobs:
        recs = W.config2_compression(n, first=first) if circuit == "compression" else W.config3_nova(n, first=first)
        bad, bodies = T.oracle_batch_u32(circuit, recs[:24])
        assert bad == 0
        wants[(circuit, n, first)] = (recs, bodies.copy())
    errors = []
    barrier = threading.Barrier(len(jobs))

    def work(circuit, n, first):
        try:
            recs, want = wants[(circuit, n, first)]
            stream = torch.cuda.Stream()
            ctx = m.Context(circuit, 0)
            r1cs = m.R1cs(ctx)
            d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
            d_st = torch.zeros(n, dtype=torch.int32, device=dev)
            viol = torch.zeros(n, dtype=torch.int32, device=dev)
            mm = torch.zeros(n, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                                 # (the tensors above were made on torch's default stream)
            barrier.wait()
            for rnd in range(6):
                buf = ctx.alloc_bodies(n * ctx.body_bytes)          # the placement allocator is shared by all contexts
                for k in (n, 7, 1):
                    ctx.run_device(d_recs.data_ptr(), k, buf.ptr, 0, 0, d_st.data_ptr(), stream.cuda_stream)
                ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), stream.cuda_stream)
                r1cs.check_device(buf.ptr, n, 0, viol.data_ptr(), 0, stream.cuda_stream)
                ctx.verify_device(buf.ptr, n, 0, mm.data_ptr(), stream.cuda_stream)
                stream.synchronize()
                with torch.cuda.stream(stream):                      # (torch's fill on the same stream as the kernels: no race of the test's own)
                    got = torch.full((24, ctx.body_bytes), 9, dtype=torch.uint8, device=dev)     # the same kernels into torch memory, to look at
                ctx.run_device(d_recs.data_ptr(), 24, got.data_ptr(), 0, 0, d_st.data_ptr(), stream.cuda_stream)
                stream.synchronize()
                assert np.array_equal(got.cpu().numpy().reshape(24, -1), want), (circuit, rnd, "bodies")
                assert int(d_st.abs().sum().item()) == 0 and int(viol.abs().sum().item()) == 0 and int(mm.abs().sum().item()) == 0, (circuit, rnd)
                buf.free()
            wc = m.WitnessCalculator(ctx)
            keys = W.COMPRESSION_KEYS if circuit == "compression" else W.NOVA_KEYS
            for i in range(4):
                body = wc.calculateBinWitness(W.record_to_input(recs[i], keys), 0)
                assert np.array_equal(np.asarray(body), want[i]), (circuit, "single", i)
            r1cs.close(); ctx.close()
        except BaseException as e:                                   # noqa: BLE001 — reported by the main thread
            errors.append((circuit, n, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
