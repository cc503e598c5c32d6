"""The Node.js side of the drop-in boundary: N-API addon + witness_calculator.js shim
(hot-proofs-blake3-circom_amd/js).  CPU tests: the addon loads, identifies the reference's circuit
binaries and refuses to compute without a device.  GPU tests: the reference's CLI flow
(generate_witness.js -> calculateWTNSBin) reproduces the golden .wtns files byte for byte."""
import json, os, shutil, subprocess
import pytest
import b3w_testlib as T

JS = os.path.join(T.PKG_DIR, "js")
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node not installed")
REF = "/root/reference"


def _node(script, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([NODE, "-e", script, *args], capture_output=True, text=True, cwd=T.ROOT, env=e, timeout=300)


@needs_node
def test_addon_loads_and_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert os.path.exists(os.path.join(JS, "b3wit_napi.node")), "run __graft_entry__.build()"
    r = _node("""
      const b = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const nat = require('./hot-proofs-blake3-circom_amd/js/b3wit_napi.node');
      console.log('abi', nat.abiVersion() >> 16);
      (async () => {
        await b('compression').then(() => console.log('UNEXPECTED')).catch(e => console.log('rejected', e.status, e.message));
        await b(Buffer.from('not a circuit'), {wasmFallback: false}).catch(e => console.log('junk', e.message));
        const say = console.log; console.log = () => {};          // the loader prints the CompileError, like the reference's
        const msg = await b(Buffer.from('not a circuit')).then(() => 'UNEXPECTED', e => e.message);
        console.log = say;
        console.log('fallback', msg);
      })();
    """)
    assert r.returncode == 0, r.stderr
    assert "abi 1" in r.stdout and "rejected 101" in r.stdout and "no CPU path" in r.stdout
    assert "junk b3wit: not one of the reference's committed BLAKE3 circuits" in r.stdout and "UNEXPECTED" not in r.stdout
    assert "fallback CompileError" in r.stdout            # unknown bytes go to the generic WebAssembly loader


@needs_node
def test_generic_wasm_fallback_reproduces_the_reference_loader():
    """Unknown circom WASM runs through js/wasm_fallback.js.  Here the reference's own circuit binaries are forced down
    that road (options.forceWasm) and compared with the goldens the reference loader produced: .wtns images, error text,
    and — on ONE calculator, in sequence — the console lines and the never-cleared error trace
    (tests/golden/nova_vesta.sequence.json)."""
    if not os.path.isdir(REF):
        pytest.skip("reference checkout not present (GPU box): the circuit binaries live there")
    g = T.golden("compression")
    ok = [c for c in g["cases"] if "error" not in c][:2] + [c for c in g["cases"] if c["name"] in ("m0_neg1",)]
    bad = [c for c in g["cases"] if "error" in c][:2]
    seq = json.load(open(os.path.join(T.GOLD, "nova_vesta.sequence.json")))["steps"]
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const crypto = require('crypto'), fs = require('fs');
      const sha = (b) => crypto.createHash('sha256').update(b).digest('hex');
      (async () => {
        const [ok, bad, seq] = JSON.parse(process.argv[1]);
        const out = {ok: [], bad: [], seq: []};
        const code = fs.readFileSync(process.argv[2]);
        let wc = await builder(code, {forceWasm: true});
        out.fields = [wc.version, wc.n32, wc.witnessSize, wc.prime.toString(), wc.circom_version(), wc.constructor.name];
        for (const c of ok) out.ok.push(sha(await wc.calculateWTNSBin(c.input, 0)));
        out.bin = sha(await wc.calculateBinWitness(ok[0].input, 0));
        out.w = (await wc.calculateWitness(ok[0].input, 0)).slice(0, 16).map(String);
        for (const c of bad) { wc = await builder(code, {forceWasm: true}); try { await wc.calculateWitness(c.input, 0); out.bad.push('NOERR'); } catch (e) { out.bad.push(e.message); } }
        const i = Object.assign({}, ok[0].input); delete i.b;
        try { await wc.calculateWitness(i, 0); } catch (e) { out.missing = e.message; }
        try { await wc.calculateWitness(Object.assign({zz: 1}, ok[0].input), 0); } catch (e) { out.unknown = e.message; }
        // one calculator, a sequence of good and rejected nova steps: console lines and error text per call
        const nova = await builder(fs.readFileSync(process.argv[3]), {forceWasm: true});
        const real = console.log;
        for (const s of seq) {
          const logs = [];
          console.log = (...a) => logs.push(a.join(' '));
          let err = null, h = null;
          try { h = sha(await nova.calculateBinWitness(s.input, 0)); } catch (e) { err = e.message; }
          console.log = real;
          out.seq.push({logs, error: err, body_sha256: h});
        }
        console.log(JSON.stringify(out));
      })().catch(e => { console.error(e); process.exit(1); });
    """, json.dumps([ok, bad, [{"input": s["input"]} for s in seq]]),
        os.path.join(REF, "build/blake3_compression/blake3_compression_js/blake3_compression.wasm"),
        os.path.join(REF, "build/blake3_nova_pasta_js/blake3_nova_pasta.wasm"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["fields"] == [2, 8, 24093, str(T.BN254_R), 2, "GenericWitnessCalculator"]
    assert out["ok"] == [c["wtns_sha256"] for c in ok] and out["bin"] == ok[0]["body_sha256"] and out["w"] == ok[0]["first16"]
    assert out["bad"] == [c["error"] for c in bad]
    assert out["missing"] == "Not all inputs have been set. Only 27 out of 28" and out["unknown"] == "Too many values for input signal zz\n"
    for got, want in zip(out["seq"], seq):
        assert got["logs"] == want["logs"] and got["error"] == want["error"] and got["body_sha256"] == want["body_sha256"], want["name"]


@needs_node
def test_addon_identifies_reference_binaries_and_module_swap():
    if not os.path.isdir(REF):
        pytest.skip("reference checkout not present (GPU box)")
    r = _node("""
      const nat = require('./hot-proofs-blake3-circom_amd/js/b3wit_napi.node');
      const fs = require('fs');
      for (const f of process.argv.slice(1)) console.log(nat.identifyWasm(fs.readFileSync(f)));
    """, *[os.path.join(REF, "build", p) for p in (
        "blake3_compression/blake3_compression_js/blake3_compression.wasm", "blake3_nova_js/blake3_nova.wasm",
        "blake3_nova_pasta_js/blake3_nova_pasta.wasm", "blake3_nova/blake3_nova_js/blake3_nova.wasm")])
    assert r.stdout.split() == ["0", "1", "2", "3"], r.stderr
    # the reference's own generate_witness.js, unchanged, resolves its require("./witness_calculator.js")
    # to the shim under `node -r register.js` (here without a GPU it must fail loudly, not fall back to WASM)
    import torch
    if torch.cuda.is_available():
        return
    inp = os.path.join(T.ROOT, "gpurun_out", "_swap_in.json")
    os.makedirs(os.path.dirname(inp), exist_ok=True)
    json.dump(T.golden("compression")["cases"][0]["input"], open(inp, "w"))
    gw = os.path.join(REF, "build/blake3_compression/blake3_compression_js/generate_witness.js")
    r = subprocess.run([NODE, "-r", os.path.join(JS, "register.js"), gw,
                        os.path.join(REF, "build/blake3_compression/blake3_compression_js/blake3_compression.wasm"),
                        inp, inp + ".wtns"], capture_output=True, text=True, timeout=120)
    assert "no HIP device" in (r.stderr + r.stdout)
    assert not os.path.exists(inp + ".wtns")


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("circuit", ["compression", "nova_vesta", "nova_bn254"])
def test_generate_witness_cli_reproduces_goldens(circuit, tmp_path):
    g = T.golden(circuit)
    names = [f for f in os.listdir(T.GOLD) if f.startswith(circuit + ".") and f.endswith(".wtns.gz")]
    for f in names:
        case = next(c for c in g["cases"] if c["name"] == f[len(circuit) + 1:-len(".wtns.gz")])
        inp, out = tmp_path / "in.json", tmp_path / "out.wtns"
        inp.write_text(json.dumps(case["input"]))
        r = subprocess.run([NODE, os.path.join(JS, "b3wit_cli.js"), circuit, str(inp), str(out)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert out.read_bytes() == T.golden_image(f)


@needs_node
@pytest.mark.gpu
def test_js_surface_errors_and_batch(tmp_path):
    g = T.golden("compression")
    cases = [c for c in g["cases"] if "error" not in c and T.is_canonical_u32("compression", c["input"])][:8]
    (tmp_path / "cases.json").write_text(json.dumps(cases))
    neg = next(c for c in g["cases"] if c["name"] == "m0_neg1")
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const crypto = require('crypto'), fs = require('fs');
      (async () => {
        const cases = JSON.parse(fs.readFileSync(process.argv[1]));
        const wc = await builder('compression');
        const out = {fields: [wc.version, wc.n32, wc.witnessSize, wc.prime.toString(), wc.circom_version()], sha: [], first16: null, errs: []};
        for (const c of cases) out.sha.push(crypto.createHash('sha256').update(await wc.calculateWTNSBin(c.input, 0)).digest('hex'));
        const w = await wc.calculateWitness(cases[0].input, 0);
        out.first16 = w.slice(0, 16).map(x => x.toString()); out.len = w.length;
        out.bin = crypto.createHash('sha256').update(await wc.calculateBinWitness(cases[0].input, 0)).digest('hex');
        const base = cases[0].input;
        const tryErr = async (inp) => { try { await wc.calculateWitness(inp); out.errs.push('NOERR'); } catch (e) { out.errs.push(e.message); } };
        let i = Object.assign({}, base); delete i.b; await tryErr(i);
        i = Object.assign({}, base, {m: base.m.slice(0, 15)}); await tryErr(i);
        i = Object.assign({}, base, {m: base.m.concat([1])}); await tryErr(i);
        i = Object.assign({}, base, {zz: 1}); await tryErr(i);
        i = Object.assign({}, base, {h: ['4294967296'].concat(base.h.slice(1))}); await tryErr(i);
        // a negative message word is accepted by the circuit (field-element path)
        out.neg = crypto.createHash('sha256').update(await wc.calculateWTNSBin(JSON.parse(process.argv[2]), 0)).digest('hex');
        // batch extension on packed records
        const recs = new Uint32Array(28 * cases.length);
        cases.forEach((c, k) => { const v = [].concat(c.input.h, c.input.m, c.input.t, [c.input.b, c.input.d]); v.forEach((x, j) => recs[28 * k + j] = Number(x)); });
        const b = await wc.calculateWitnessBatch(recs);
        out.batch = {n: b.n, status: Array.from(b.status), pub0: Array.from(b.publicOutputs.slice(0, 16)).map(String),
                     body3: crypto.createHash('sha256').update(b.fetch(3)).digest('hex'),
                     files: b.writeWtns(process.argv[3], 'js_'),
                     verify: Array.from(b.verify()),
                     file5: crypto.createHash('sha256').update(fs.readFileSync(process.argv[3] + '/js_5.wtns')).digest('hex')};
        // constraint check of the batch on the device (the derived blake3_compression system), clean and with a tampered input
        out.r1cs = wc.loadR1cs();
        const cc = b.checkConstraints();
        out.cc = [Array.from(cc.violations), Array.from(cc.first)];
        // the calculator owns one device batch: a later run replaces it, and the older result must refuse to act on it
        const b2 = await wc.calculateWitnessBatch(recs.slice(0, 56));
        out.stale = [];
        for (const f of [() => b.verify(), () => b.fetch(0), () => b.writeWtns(process.argv[3], 'stale_'), () => b.commit()])
          try { f(); out.stale.push('NOERR'); } catch (e) { out.stale.push(e.message); }
        out.v2 = Array.from(b2.verify());
        try { b2.writeWtns(process.argv[3], 'wrap_', 0xFFFFFFFF, 2); out.wrap = 'NOERR'; } catch (e) { out.wrap = 'refused'; }
        out.none = b2.writeWtns(process.argv[3], 'none_', 2, 0);
        console.log(JSON.stringify(out));
      })().catch(e => { console.error(e); process.exit(1); });
    """, str(tmp_path / "cases.json"), json.dumps(neg["input"]), str(tmp_path))
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["fields"] == [2, 8, 24093, str(T.BN254_R), 2]
    assert out["sha"] == [c["wtns_sha256"] for c in cases]
    assert out["len"] == 24093 and out["first16"] == cases[0]["first16"] and out["bin"] == cases[0]["body_sha256"]
    assert out["errs"][0] == "Not all inputs have been set. Only 27 out of 28"
    assert out["errs"][1] == "Not enough values for input signal m\n"
    assert out["errs"][2] == "Too many values for input signal m\n"
    assert out["errs"][3] == "Too many values for input signal zz\n"
    assert out["errs"][4] == next(c for c in g["cases"] if c["name"] == "err_h0_2p32")["error"].replace(
        "ToBits_3 line: 153\nError in template XorWord2_39 line: 66", "ToBits_3 line: 153\nError in template XorWord2_39 line: 66")
    assert out["neg"] == neg["wtns_sha256"]
    assert out["batch"]["n"] == 8 and out["batch"]["status"] == [0] * 8
    assert out["batch"]["pub0"] == [str(x) for x in cases[0]["first16"][1:]] + [out["batch"]["pub0"][15]]
    assert out["batch"]["body3"] == cases[3]["body_sha256"]
    assert out["batch"]["verify"] == [0] * 8
    assert out["batch"]["files"] == 8 and out["batch"]["file5"] == cases[5]["wtns_sha256"]
    assert out["r1cs"] == {"nConstraints": 24544, "nWires": 24093, "nTerms": 117760}
    assert out["cc"] == [[0] * 8, [0xFFFFFFFF] * 8]
    assert out["stale"] == ["stale batch result: a later calculateWitnessBatch on this calculator replaced it"] * 4
    assert out["v2"] == [0, 0] and out["wrap"] == "refused" and out["none"] == 0
    assert not [f for f in os.listdir(tmp_path) if f.startswith(("stale_", "wrap_", "none_"))]


@needs_node
@pytest.mark.gpu
def test_nova_sequence_on_one_calculator_logs_and_sticky_errors():
    """tests/golden/nova_vesta.sequence.json: ten witnesses on ONE reference calculator (tools/gen_sequence_golden.js) —
    console.log lines ("D_FLAGS:  0": on success and on asserts past the first component, not on CheckDepth asserts)
    and, with options.strictErrorParity, the reference's accumulating error text, call for call."""
    seq = json.load(open(os.path.join(T.GOLD, "nova_vesta.sequence.json")))
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const crypto = require('crypto'), fs = require('fs');
      (async () => {
        const seq = JSON.parse(fs.readFileSync(process.argv[1]));
        const out = {};
        for (const mode of ['strict', 'default', 'quiet']) {
          const wc = await builder('nova_vesta', mode === 'strict' ? {strictErrorParity: true} : mode === 'quiet' ? {logDFlags: false} : undefined);
          const real = console.log, res = [];
          for (const s of seq.steps) {
            const logs = []; console.log = (...a) => logs.push(a.join(' '));
            let err = null, sha = null;
            try { sha = crypto.createHash('sha256').update(await wc.calculateBinWitness(s.input, 0)).digest('hex'); } catch (e) { err = e.message; }
            console.log = real;
            res.push({logs, error: err, body_sha256: sha});
          }
          out[mode] = res;
        }
        console.log(JSON.stringify(out));
      })().catch(e => { console.error(e); process.exit(1); });
    """, os.path.join(T.GOLD, "nova_vesta.sequence.json"))
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    own = {c["name"]: c.get("error") for c in T.golden("nova_vesta")["cases"]}
    for k, s in enumerate(seq["steps"]):
        strict, default, quiet = out["strict"][k], out["default"][k], out["quiet"][k]
        assert strict["logs"] == s["logs"] and default["logs"] == s["logs"] and quiet["logs"] == [], (s["name"], strict["logs"])
        assert strict["body_sha256"] == s["body_sha256"] == default["body_sha256"], s["name"]
        assert strict["error"] == s["error"], (s["name"], strict["error"])
        if s["error"] is not None and s["name"] in own:       # default mode: only this call's own trace
            assert default["error"] == own[s["name"]], s["name"]


@needs_node
@pytest.mark.gpu
def test_js_fold_preimage_matches_blake3_and_python_driver(tmp_path):
    """wc.foldPreimage (N-API chainFold over the native b3w_chain_* driver): the root equals BLAKE3(preimage)
    (independent pure-Python BLAKE3), every step verifies, and counts / public outputs equal the Python driver's."""
    import numpy as np, torch, blake3_ref
    m = T.pkg()
    shapes = {"complete": 16 * 1024, "ragged": 5 * 1024 + 100, "tiny": 4}
    for name, nbytes in shapes.items():
        data = ((np.arange(nbytes, dtype=np.uint64) * 2654435761 + 7) % 251).astype(np.uint8)
        (tmp_path / (name + ".bin")).write_bytes(data.tobytes())
    import ec_ref as E
    gens_bytes = E.points_to_bytes(E.random_points("vesta", T.NWIT["nova_vesta"]))
    (tmp_path / "gens.bin").write_bytes(gens_bytes)
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const fs = require('fs'), crypto = require('crypto');
      (async () => {
        const wc = await builder('nova_vesta', {logDFlags: false});
        const out = {};
        for (const name of ['complete', 'ragged', 'tiny']) {
          const r = await wc.foldPreimage(fs.readFileSync(process.argv[1] + '/' + name + '.bin'), {batchSteps: 64});
          out[name] = {nLeaf: r.nLeafSteps, nPar: r.nParentSteps, nChunks: r.nChunks, pathLen: r.pathLen, hash: r.hash, placement: r.placement,
                       bad: Array.from(r.status).filter(x => x !== 0).length,
                       pub: crypto.createHash('sha256').update(Buffer.from(r.publicOutputs.buffer, r.publicOutputs.byteOffset, r.publicOutputs.byteLength)).digest('hex')};
        }
        // every step witness checked against the step circuit (derived system of this build) while it sits in the ring
        wc.loadR1cs();
        const cc = await wc.foldPreimage(fs.readFileSync(process.argv[1] + '/ragged.bin'), {batchSteps: 64, checkConstraints: true});
        out.checked = {n: cc.violations.length, bad: Array.from(cc.violations).filter(x => x !== 0).length, hash: cc.hash};
        // commitments only (setCommitKey, then foldPreimage with commitOnly): one point per step, same root
        wc.setCommitKey('vesta', new Uint8Array(fs.readFileSync(process.argv[1] + '/gens.bin')), 0, 12);
        const c = await wc.foldPreimage(fs.readFileSync(process.argv[1] + '/complete.bin'), {batchSteps: 64, commitOnly: true});
        out.commitOnly = {hash: c.hash, n: c.commitments.length / 64, bad: Array.from(c.status).filter(x => x !== 0).length,
                          points: crypto.createHash('sha256').update(Buffer.from(c.commitments.buffer, c.commitments.byteOffset, c.commitments.byteLength)).digest('hex')};
        console.log(JSON.stringify(out));
      })().catch(e => { console.error(e); process.exit(1); });
    """, str(tmp_path))
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    ctx = m.Context("nova_vesta", 0)
    import hashlib
    key = m.CommitKey(ctx, "vesta", gens_bytes, 0, 12)
    data = np.frombuffer((tmp_path / "complete.bin").read_bytes(), dtype=np.uint8).copy()
    pts = torch.zeros((16 * 16 + 16 * 4, 64), dtype=torch.uint8, device="cuda:0")
    py = m.chain.fold_witnesses(ctx, data, batch_steps=64, commit_only=(key, pts))
    torch.cuda.synchronize()
    chk = out.pop("checked")
    assert chk == {"n": out["ragged"]["nLeaf"] + out["ragged"]["nPar"], "bad": 0, "hash": out["ragged"]["hash"]}
    co = out.pop("commitOnly")
    assert co["hash"] == out["complete"]["hash"] and co["n"] == pts.shape[0] and co["bad"] == 0
    assert co["points"] == hashlib.sha256(pts.cpu().numpy().tobytes()).hexdigest() and int(pts.max(dim=1).values.min().item()) > 0
    key.close()
    for name, nbytes in shapes.items():
        data = np.frombuffer((tmp_path / (name + ".bin")).read_bytes(), dtype=np.uint8).copy()
        assert out[name]["hash"] == blake3_ref.blake3(data.tobytes()).hex(), name
        assert out[name]["bad"] == 0
        py = m.chain.fold_witnesses(ctx, data, batch_steps=64)
        torch.cuda.synchronize()
        assert (out[name]["nLeaf"], out[name]["nPar"]) == (py["n_leaf_steps"], py["n_parent_steps"]), name
        assert out[name]["pub"] == hashlib.sha256(py["public"].cpu().numpy().tobytes()).hexdigest(), name
    # 6 chunks: paths of length 3 for the leading four chunks, 2 for chunks 4 and 5 (b3w_chain_plan_parents_device)
    assert out["complete"]["nPar"] == 16 * 4 and out["ragged"]["nPar"] == 4 * 3 + 2 * 2 and out["tiny"]["nLeaf"] == 1
    ctx.close()


@needs_node
@pytest.mark.gpu
def test_js_rccl_exchange_single_rank():
    """commUniqueId / wc.joinRanks / batch.allgatherPublic over the C-ABI's RCCL exchange (one rank on the test box)."""
    g = T.golden("compression")
    cases = [c for c in g["cases"] if "error" not in c and T.is_canonical_u32("compression", c["input"])][:4]
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      (async () => {
        const cases = JSON.parse(process.argv[1]);
        const wc = await builder('compression');
        const id = builder.commUniqueId();
        wc.joinRanks(id, 0, 1);
        const recs = new Uint32Array(28 * cases.length);
        cases.forEach((c, k) => { const v = [].concat(c.input.h, c.input.m, c.input.t, [c.input.b, c.input.d]); v.forEach((x, j) => recs[28 * k + j] = Number(x)); });
        const b = await wc.calculateWitnessBatch(recs);
        const all = b.allgatherPublic();
        // sharded chained mode through the same communicator (one rank: its shard is everything)
        const nv = await builder('nova_vesta', {logDFlags: false});
        nv.joinRanks(builder.commUniqueId(), 0, 1);
        const pre = Buffer.alloc(8 * 1024); for (let i = 0; i < pre.length; i++) pre[i] = (i * 7 + 3) % 251;
        const f = await nv.foldPreimage(pre, {batchSteps: 64});
        const solo = await (await builder('nova_vesta', {logDFlags: false})).foldPreimage(pre, {batchSteps: 64});
        // the gathered h_out (8 words per step, global step order) = words 2..9 of this rank's own public outputs when it is alone
        const nl = f.nLeafSteps, np_ = f.nParentSteps;
        let hout = f.hOutAll.length === 8 * nl && f.hOutParentsAll.length === 8 * np_ && solo.hOutAll === undefined;
        for (let s = 0; s < nl + np_ && hout; s++)
          for (let j = 0; j < 8; j++) hout = hout && (s < nl ? f.hOutAll[8 * s + j] : f.hOutParentsAll[8 * (s - nl) + j]) === f.publicOutputs[15 * s + 2 + j];
        const fold = {hash: f.hash, same: f.hash === solo.hash && Array.from(f.publicOutputs).join() === Array.from(solo.publicOutputs).join(),
                      first: f.firstChunk, local: f.nChunksLocal, par: f.nParentSteps, bad: Array.from(f.status).filter(x => x !== 0).length, hout};
        console.log(JSON.stringify({idLen: id.length, n: b.n, same: Array.from(all).join() === Array.from(b.publicOutputs).join(), first: Array.from(all.slice(0, 15)).map(String), fold}));
      })().catch(e => { console.error(e); process.exit(1); });
    """, json.dumps(cases))
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["idLen"] == 128 and out["n"] == 4 and out["same"] is True
    assert out["first"] == [str(x) for x in cases[0]["first16"][1:]]
    import blake3_ref
    pre = bytes((i * 7 + 3) % 251 for i in range(8 * 1024))
    assert out["fold"] == {"hash": blake3_ref.blake3(pre).hex(), "same": True, "first": 0, "local": 8, "par": 24, "bad": 0, "hout": True}


@needs_node
@pytest.mark.gpu
def test_js_two_node_processes_join_ranks_over_host_shared_memory(tmp_path):
    """wc.joinRanksHost -> foldPreimage().hOutAll with TWO Node processes on the one GPU (b3w_comm_create_host): both get the
    h_out of every step of BOTH ranks, equal to what one rank computes for the whole preimage; ragged shards (6 chunks, a
    partial last one), then a one-chunk preimage (rank 1 has no chunk)."""
    import uuid
    import numpy as np, torch
    m = T.pkg()
    shapes = {"ragged": 5 * 1024 + 100, "tiny": 700}
    for name, nbytes in shapes.items():
        (tmp_path / (name + ".bin")).write_bytes(m.workloads.lcg_preimage(nbytes, seed=1).tobytes())
    script = """
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const fs = require('fs');
      (async () => {
        const [dir, name, rank] = [process.argv[1], process.argv[2], Number(process.argv[3])];
        const wc = await builder('nova_vesta', {logDFlags: false});
        wc.joinRanksHost(name, rank, 2);
        const out = {};
        for (const shape of ['ragged', 'tiny']) {
          const r = await wc.foldPreimage(fs.readFileSync(dir + '/' + shape + '.bin'), {batchSteps: 64});
          out[shape] = {first: r.firstChunk, local: r.nChunksLocal, hash: r.hash, bad: Array.from(r.status).filter(x => x !== 0).length,
                        hOutAll: Array.from(r.hOutAll), hOutParentsAll: Array.from(r.hOutParentsAll)};
        }
        console.log(JSON.stringify(out));
      })().catch(e => { console.error(e); process.exit(1); });
    """
    name = "/b3w_js_" + uuid.uuid4().hex[:16]
    env = dict(os.environ, B3W_PLACEMENT="plain", B3W_HOSTCOMM_TIMEOUT_S="90")
    procs = [subprocess.Popen([NODE, "-e", script, str(tmp_path), name, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              cwd=T.ROOT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    res = [json.loads(o[0].strip().splitlines()[-1]) for o in outs]
    ctx = m.Context("nova_vesta", 0)
    import blake3_ref
    for shape, nbytes in shapes.items():
        data = m.workloads.lcg_preimage(nbytes, seed=1)
        one = m.chain.fold_witnesses(ctx, data, batch_steps=64)
        torch.cuda.synchronize()
        leaf = one["h_out_all"].cpu().numpy().view(np.uint32).reshape(-1).tolist()
        par = one["h_out_parents_all"].cpu().numpy().view(np.uint32).reshape(-1).tolist()
        for r in range(2):
            got = res[r][shape]
            assert got["hash"] == blake3_ref.blake3(data.tobytes()).hex() and got["bad"] == 0, (shape, r)
            assert got["hOutAll"] == leaf and got["hOutParentsAll"] == par, (shape, r)
    assert [(res[r]["ragged"]["first"], res[r]["ragged"]["local"]) for r in range(2)] == [(0, 3), (3, 3)]
    assert [(res[r]["tiny"]["first"], res[r]["tiny"]["local"]) for r in range(2)] == [(0, 1), (1, 0)]
    ctx.close()


@needs_node
@pytest.mark.gpu
def test_js_batch_commit_matches_plain_integer_group_law(tmp_path):
    """wc.setCommitKey / batch.commit() from Node against tests/ec_ref.py."""
    import numpy as np, ec_ref as E
    g = T.golden("compression")
    cases = [c for c in g["cases"] if "error" not in c and T.is_canonical_u32("compression", c["input"])][:2]
    gens = E.random_points("bn254_g1", T.NWIT["compression"] - 17)
    (tmp_path / "gens.bin").write_bytes(E.points_to_bytes(gens))
    import subprocess, sys
    fk = subprocess.run([sys.executable, os.path.join(T.ROOT, "tools", "fold_key.py"), "compression", "bn254_g1", str(tmp_path / "gens.bin"),
                         str(tmp_path / "folded"), "--first-slot", "17"], capture_output=True, text=True, cwd=T.ROOT, timeout=600)
    assert fk.returncode == 0 and "'folded_slots': 7" in fk.stdout, (fk.stdout[-500:], fk.stderr[-1500:])
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const fs = require('fs');
      (async () => {
        const cases = JSON.parse(process.argv[1]);
        const wc = await builder('compression');
        wc.setCommitKey('bn254_g1', new Uint8Array(fs.readFileSync(process.argv[2])), 17);
        const recs = new Uint32Array(28 * cases.length);
        cases.forEach((c, k) => { const v = [].concat(c.input.h, c.input.m, c.input.t, [c.input.b, c.input.d]); v.forEach((x, j) => recs[28 * k + j] = Number(x)); });
        const b = await wc.calculateWitnessBatch(recs);
        const c = b.commit();
        const r = wc.commitRecords(recs);                       // the same points without the witnesses
        // a FOLDED key (tools/fold_key.py wrote the folded generators and the mask): same points again
        wc.setCommitKey('bn254_g1', new Uint8Array(fs.readFileSync(process.argv[3] + '.gens')), 17, 0, new Uint8Array(fs.readFileSync(process.argv[3] + '.mask')));
        const b2 = await wc.calculateWitnessBatch(recs);
        const f = b2.commit();
        const fr = wc.commitRecords(recs);
        console.log(JSON.stringify({points: Buffer.from(c.points).toString('hex'), status: Array.from(c.status),
                                    rpoints: Buffer.from(r.points).toString('hex'), rstatus: Array.from(r.status),
                                    fpoints: Buffer.from(f.points).toString('hex'), frpoints: Buffer.from(fr.points).toString('hex'),
                                    rpub: Array.from(r.publicOutputs), pub: Array.from(b.publicOutputs)}));
      })().catch(e => { console.error(e); process.exit(1); });
    """, json.dumps(cases), str(tmp_path / "gens.bin"), str(tmp_path / "folded"))
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["status"] == [0, 0]
    assert out["fpoints"] == out["points"] and out["frpoints"] == out["points"], "the folded key gives other points"
    assert out["rpoints"] == out["points"] and out["rstatus"] == [0, 0] and out["rpub"] == out["pub"] and len(out["pub"]) == 32
    W = T.workloads()
    recs = np.array([[int(x) for x in (c["input"]["h"] + c["input"]["m"] + c["input"]["t"] + [c["input"]["b"], c["input"]["d"]])] for c in cases], dtype=np.uint32)
    _, bodies = T.oracle_batch_u32("compression", recs)
    pts = bytes.fromhex(out["points"])
    for i in range(2):
        vals = [int.from_bytes(bodies[i, 32 * s: 32 * s + 32].tobytes(), "little") for s in range(17, T.NWIT["compression"])]
        assert E.point_from_bytes(pts[64 * i: 64 * i + 64]) == E.commit(vals, gens, "bn254_g1"), i


@needs_node
@pytest.mark.gpu
def test_cli_fold_prints_the_blake3_hash(tmp_path):
    import blake3_ref
    pre = bytes((i * 13 + 5) % 256 for i in range(3 * 1024 + 77))
    (tmp_path / "pre.bin").write_bytes(pre)
    r = subprocess.run([NODE, os.path.join(JS, "b3wit_cli.js"), "--fold", "nova_bn254", str(tmp_path / "pre.bin"), str(tmp_path / "pub.json")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "blake3 = " + blake3_ref.blake3(pre).hex() in r.stdout and "(0 rejected)" in r.stdout
    pub = json.loads((tmp_path / "pub.json").read_text())
    assert pub["nLeafSteps"] == 3 * 16 + 2 and len(pub["publicOutputs"]) == 15 * (pub["nLeafSteps"] + pub["nParentSteps"])


def _toy_circom_wasm():
    """A circom-ABI witness generator that is none of the reference's: one input `a`, witness [1, a, a * a mod p] over p = 2^31 - 1
    (one 32-bit limb).  Hand-assembled WebAssembly with the export set the loaders drive (witness_calculator.js:109-169 of the
    reference): no imports, two globals (the shared read/write word, the input)."""
    def leb(n):
        out = bytearray()
        while True:
            b = n & 0x7F
            n >>= 7
            out.append(b | (0x80 if n else 0))
            if not n:
                return bytes(out)

    def sleb(n):
        out = bytearray()
        while True:
            b = n & 0x7F
            n >>= 7
            done = (n == 0 and not b & 0x40) or (n == -1 and b & 0x40)
            out.append(b | (0 if done else 0x80))
            if done:
                return bytes(out)

    def vec(items):
        return leb(len(items)) + b"".join(items)

    def section(sid, body):
        return bytes([sid]) + leb(len(body)) + body
    I32, I64 = 0x7F, 0x7E
    types = [([], [I32]), ([], []), ([I32], [I32]), ([I32, I32], []), ([I32, I32], [I32]), ([I32], []), ([I32, I32, I32], [])]
    P = 0x7FFFFFFF
    GET_S, SET_S, GET_A, SET_A = b"\x23\x00", b"\x24\x00", b"\x23\x01", b"\x24\x01"
    c32 = lambda v: b"\x41" + sleb(v)
    funcs = [  # (export name, type index, body)
        ("getVersion", 0, c32(2)), ("getFieldNumLen32", 0, c32(1)), ("getRawPrime", 1, c32(P) + SET_S),
        ("readSharedRWMemory", 2, GET_S), ("writeSharedRWMemory", 3, b"\x20\x01" + SET_S), ("getWitnessSize", 0, c32(3)),
        ("getInputSize", 0, c32(1)), ("getInputSignalSize", 4, c32(1)), ("init", 5, c32(0) + SET_A), ("setInputSignal", 6, GET_S + SET_A),
        ("getWitness", 5,
         b"\x20\x00\x45\x04\x40" + c32(1) + SET_S + b"\x05"                     # if (i == 0) shared = 1 else
         + b"\x20\x00" + c32(1) + b"\x46\x04\x40" + GET_A + SET_S + b"\x05"         #   if (i == 1) shared = a else
         + GET_A + b"\xAD" + GET_A + b"\xAD\x7E\x42" + sleb(P) + b"\x82\xA7" + SET_S  #     shared = (u64 a * u64 a) % p
         + b"\x0B\x0B"),
        ("getMessageChar", 0, c32(0)),
    ]
    mod = b"\x00asm\x01\x00\x00\x00"
    mod += section(1, vec([b"\x60" + vec([bytes([t]) for t in a]) + vec([bytes([t]) for t in r]) for a, r in types]))
    mod += section(3, vec([leb(t) for _, t, _ in funcs]))
    mod += section(6, vec([bytes([I32, 1]) + c32(0) + b"\x0B"] * 2))              # two mutable i32 globals
    mod += section(7, vec([leb(len(n)) + n.encode() + b"\x00" + leb(i) for i, (n, _, _) in enumerate(funcs)]))
    mod += section(10, vec([leb(len(b) + 2) + b"\x00" + b + b"\x0B" for _, _, b in funcs]))
    return mod


@needs_node
def test_toy_circom_wasm_assembles_and_runs_through_the_fallback_loader(tmp_path):
    """(CPU) the hand-assembled module is valid WebAssembly with the circom export set; the shim's fallback computes its witness."""
    w = tmp_path / "toy.wasm"
    w.write_bytes(_toy_circom_wasm())
    r = _node("""
      const wf = require('./hot-proofs-blake3-circom_amd/js/wasm_fallback.js');
      (async () => {
        const wc = await wf(require('fs').readFileSync(process.argv[1]));
        console.log(wc.prime.toString(), wc.witnessSize, (await wc.calculateWitness({a: 123456}, 0)).join(','));
      })().catch(e => { console.log('ERR', e.message); });
    """, str(w))
    assert r.stdout.split() == ["2147483647", "3", f"1,123456,{123456 * 123456 % 0x7FFFFFFF}"], (r.stdout, r.stderr)


@needs_node
@pytest.mark.gpu
def test_builder_takes_unknown_wasm_bytes_and_circuit_names_in_one_process(tmp_path):
    """r04 verdict, weak 1(iii): on the GPU box builder() only ever saw circuit NAMES (the reference's .wasm files cannot travel).
    Here one Node process hands it (a) the bytes of a circom witness generator that is not one of the four committed circuits — a
    hand-assembled toy module: sha256 identification says "unknown", the generic loader computes its witness; (b) the same bytes
    with wasmFallback: false and bytes that are no WebAssembly at all — refused / CompileError; (c) a circuit name — the GPU path,
    whose .wtns image is the reference's golden one."""
    import hashlib
    g = T.golden("compression")
    case = next(c for c in g["cases"] if "error" not in c)
    w, inp = tmp_path / "toy.wasm", tmp_path / "in.json"
    w.write_bytes(_toy_circom_wasm())
    inp.write_text(json.dumps(case["input"]))
    r = _node("""
      const builder = require('./hot-proofs-blake3-circom_amd/js/witness_calculator.js');
      const nat = require('./hot-proofs-blake3-circom_amd/js/b3wit_napi.node');
      const fs = require('fs'), crypto = require('crypto');
      (async () => {
        const toy = fs.readFileSync(process.argv[1]);
        console.log('identify', nat.identifyWasm(toy));
        const t = await builder(toy);
        console.log('toy', t.constructor.name, t.witnessSize, (await t.calculateWitness({a: 77777}, 0)).join(','));
        const img = await t.calculateWTNSBin({a: 5}, 0);
        console.log('toywtns', img.length, Buffer.from(img.slice(0, 4)).toString());
        await builder(toy, {wasmFallback: false}).then(() => console.log('UNEXPECTED'), e => console.log('refused', e.message));
        const say = console.log; console.log = () => {};
        const msg = await builder(Buffer.from('not a circuit')).then(() => 'UNEXPECTED', e => e.message);
        console.log = say;
        console.log('junk', msg.split('\\n')[0]);
        const wc = await builder('compression');
        const out = await wc.calculateWTNSBin(JSON.parse(fs.readFileSync(process.argv[2], 'utf8')), 0);
        console.log('gpu', wc.constructor.name, crypto.createHash('sha256').update(out).digest('hex'));
      })().catch(e => { console.log('ERR', e.message); });
    """, str(w), str(inp))
    assert r.returncode == 0, r.stderr
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())
    assert lines["identify"] == "-1"
    assert lines["toy"] == f"GenericWitnessCalculator 3 1,77777,{77777 * 77777 % 0x7FFFFFFF}"
    assert lines["toywtns"] == f"{4 * (11 + 1 + 3)} wtns"
    assert "not one of the reference's committed BLAKE3 circuits" in lines["refused"] and "UNEXPECTED" not in r.stdout
    assert lines["junk"].startswith("CompileError")
    assert lines["gpu"] == "WitnessCalculator " + case["wtns_sha256"], r.stdout
