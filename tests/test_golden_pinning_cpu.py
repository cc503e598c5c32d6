"""Do the committed fixtures pin the witness LAYOUT, slot by slot?  The CPU restatement places its signals with the product's own
layout files (tests/b3w_testlib.py), so a layout that swapped two slots would pass every fixture on which the two hold equal values.
Here: the slots are grouped by their values over all of a build's successful reference-made cases (tests/golden/<build>.json: the
sha256 of the reference WASM's whole witness per case — checked on the way); then several hundred FURTHER inputs go through the
restatement alone, and none of them may separate two slots that no fixture separates.  What stays together then is together for
every input the restatement can produce of these families: the same signal twice (a gadget's output bits and the next one's input
bits), where a swap changes nothing.  tools/gen_golden.py (pin_slots) chose the `pin_*` cases for exactly this, from the
reference's own witnesses."""
import os
import random
import sys

import numpy as np
import pytest

import b3w_testlib as T

sys.path.insert(0, os.path.join(T.ROOT, "tools"))
import recover_layout as RL      # (probe generators only: pure Python, no reference needed)


def _slot_keys(body, nwit):
    w = np.frombuffer(np.ascontiguousarray(body).tobytes(), dtype="<u8").reshape(nwit, 4)
    return w[:, 0] * np.uint64(0x9E3779B97F4A7C15) ^ w[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) ^ w[:, 2] * np.uint64(0x165667B19E3779F9) ^ w[:, 3] * np.uint64(0x27D4EB2F165667C5)


def _refine(gid, body, nwit):
    pair = np.stack([gid.astype(np.uint64), _slot_keys(body, nwit)], axis=1)
    _, new = np.unique(pair, axis=0, return_inverse=True)
    return new.reshape(-1)


@pytest.mark.parametrize("circuit", T.CIRCUITS)
def test_no_further_input_separates_slots_the_fixtures_leave_together(circuit):
    nwit = T.NWIT[circuit]
    g = T.golden(circuit)
    gid = np.zeros(nwit, dtype=np.int64)
    used = 0
    for case in g["cases"]:
        if "error" in case:
            continue
        rc, body, err = T.oracle_witness(circuit, T.normalize_input(circuit, case["input"]))
        assert rc == 0 and T.sha256(body) == case["body_sha256"], case["name"]      # the reference WASM's witness, byte for byte
        gid = _refine(gid, body, nwit)
        used += 1
    groups = int(gid.max()) + 1
    assert used >= 60 and sum(c["name"].startswith("pin_") for c in g["cases"]) >= 10
    # further inputs, none of them a fixture: the configuration streams far behind the fixtures' indices, and seeded probes
    w = T.workloads()
    rng = random.Random(20260105)
    if circuit == "compression":
        recs = w.config2_compression(1200)[900:]
        extra = [w.record_to_input(r, w.COMPRESSION_KEYS) for r in recs] + [RL.compression_probe(rng) for _ in range(200)]
    else:
        recs = w.config3_nova(1200)[900:]
        extra = [w.record_to_input(r, w.NOVA_KEYS) for r in recs] + [RL.nova_probe(rng) for _ in range(300)]
        extra += [RL.nova_probe(rng, directed=(i, bit)) for i in range(64) for bit in (0, 1)]
    split_by = []
    for k, inp in enumerate(extra):
        rc, body, err = T.oracle_witness(circuit, T.normalize_input(circuit, inp))
        if rc != 0:
            continue
        new = _refine(gid, body, nwit)
        if int(new.max()) + 1 != groups:
            split_by.append((k, int(new.max()) + 1 - groups))
    assert not split_by, f"{circuit}: inputs {split_by[:8]} separate slots that no reference-made fixture separates ({groups} groups)"
    # (for the record: how much of a witness is the same signal more than once)
    assert 0.55 < groups / nwit < 0.75
