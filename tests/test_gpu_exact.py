"""The exact (field-element) device kernel: every input the WASM accepts or rejects, canonical or not,
must come out like the oracle says (which is pinned against the WASM): random "wild" probes."""
import random
import numpy as np
import pytest
import b3w_testlib as T

pytestmark = pytest.mark.gpu


def _wild_inputs(circuit, n, seed):
    rng = random.Random(seed)
    p = T.PRIME[circuit]
    W = T.workloads()
    out = []
    for i in range(n):
        if circuit == "compression":
            inp = W.record_to_input(W.config2_compression(1, first=1000 + i)[0], W.COMPRESSION_KEYS)
            for j in range(16):
                r = rng.random()
                if r < 0.25:
                    inp["m"][j] = -rng.randint(1, 5000)
                elif r < 0.4:
                    inp["m"][j] = (1 << 32) + rng.getrandbits(33)          # may or may not overflow Bits34
            if rng.random() < 0.15:
                inp["h"][rng.randrange(8)] = (1 << 32) + rng.getrandbits(8)   # rejected somewhere
            if rng.random() < 0.1:
                inp["t"][rng.randrange(2)] = p - rng.randint(1, 9)
        else:
            inp = W.record_to_input(W.config3_nova(1, first=2000 + i)[0], W.NOVA_KEYS)
            parent = inp["depth"] < inp["leaf_depth"] - 1
            if rng.random() < 0.5:
                inp["n_blocks"] = rng.getrandbits(250)
            if rng.random() < 0.5:
                inp["block_count"] = rng.getrandbits(250) if rng.random() < 0.6 else inp["n_blocks"] - 1
            if rng.random() < 0.3:
                inp["total_depth"] = rng.getrandbits(250)
            if rng.random() < 0.3:
                x = rng.getrandbits(249)
                inp["leaf_depth"] = x + (inp["leaf_depth"] - inp["depth"]); inp["depth"] = x
            if parent:
                inp["m"] = inp["m"][:8] + [rng.getrandbits(252) for _ in range(8)]
                if rng.random() < 0.5:
                    inp["h"][rng.randrange(8)] = -rng.randint(1, 99)
                if rng.random() < 0.5:
                    inp["chunk_idx_low"] = rng.getrandbits(64)
            else:
                for j in range(16):
                    if rng.random() < 0.2:
                        inp["m"][j] = -rng.randint(1, 5000)
                if rng.random() < 0.1:
                    inp["chunk_idx_high"] = 1 << 32                       # rejected: t[1] feeds a 32-bit decomposition
            if rng.random() < 0.08:
                inp["depth"] = inp["leaf_depth"] + rng.randint(0, 3)        # rejected by CheckDepth
        out.append(inp)
    return out


@pytest.mark.parametrize("circuit", T.CIRCUITS)
def test_exact_kernel_matches_oracle_on_wild_inputs(circuit):
    m = T.pkg()
    wc = m.builder(circuit)
    nok = nrej = 0
    for inp in _wild_inputs(circuit, 160, 77):
        rc, want, _ = T.oracle_witness(circuit, T.normalize_input(circuit, inp))
        if rc == 0:
            got = wc.calculateBinWitness(inp, 0)
            assert np.array_equal(got, want), inp
            nok += 1
        else:
            with pytest.raises(m.B3WError, match="Assert Failed") as e:
                wc.calculateBinWitness(inp, 0)
            assert e.value.status == m.B3W_E_ASSERT_FAILED
            nrej += 1
    assert nok >= 50 and nrej >= 10, (nok, nrej)
