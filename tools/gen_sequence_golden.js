// Build-container-only tool (like tools/wasm_oracle.js): runs a SEQUENCE of witnesses on ONE calculator of the
// reference (its own witness_calculator.js + committed WASM, loaded from /root/reference at run time) and records,
// per call, the console.log lines and the error message.  Pins two loader behaviours the per-case goldens cannot:
// the "D_FLAGS:  0" line the nova circuits log (circuits/blake3_nova.circom:166 via writeBufferMessage,
// witness_calculator.js:44-58) and the never-cleared errStr (witness_calculator.js:16,41).
//   node tools/gen_sequence_golden.js > tests/golden/nova_vesta.sequence.json
"use strict";
const fs = require("fs"), path = require("path");
const refDir = process.env.B3W_REFERENCE_DIR || "/root/reference";
const builder = require(path.join(refDir, "blake3_nova_js/witness_calculator.js"));
async function main() {
  const root = path.dirname(__dirname);
  const gold = JSON.parse(fs.readFileSync(path.join(root, "tests/golden/nova_vesta.json")));
  const byName = {};
  for (const c of gold.cases) byName[c.name] = c;
  const code = fs.readFileSync(path.join(refDir, "build/blake3_nova_pasta_js/blake3_nova_pasta.wasm"));
  const wc = await builder(code);
  const steps = ["config3_0", "err_depth_ge_leaf", "config3_1", "err_b_2p32", "err_h0_neg1", "config3_2", "err_leaf_far",
                 "err_cil_2p32", "config3_3"].map((n) => ({ name: n, input: byName[n].input }));
  steps.push({ name: "cih_2p40", input: Object.assign({}, byName["config3_0"].input, { chunk_idx_high: "1099511627776" }) });
  const real = console.log;
  const out = [];
  for (const s of steps) {
    const logs = [];
    console.log = (...a) => logs.push(a.join(" "));
    let err = null, sha = null;
    try {
      const body = await wc.calculateBinWitness(s.input, 0);
      sha = require("crypto").createHash("sha256").update(body).digest("hex");
    } catch (e) { err = e.message; }
    console.log = real;
    out.push({ name: s.name, input: s.input, logs, error: err, body_sha256: sha });
  }
  console.log(JSON.stringify({ circuit: "nova_vesta", generated_by: "tools/gen_sequence_golden.js (reference WASM, one calculator)", steps: out }, null, 1));
}
main().catch((e) => { console.error(e); process.exit(1); });
