// Build-container-only tool: drives the reference's committed circom WASM through the
// reference's own witness_calculator.js (loaded from --reference-dir at run time; nothing
// from the reference is copied into this repo) and dumps witness bodies for a JSON list
// of inputs.  Used to (1) recover slot layouts, (2) generate tests/golden fixtures,
// (3) validate the oracle/ C restatement.  Never runs on the GPU box.
//
//   node tools/wasm_oracle.js <circuit.wasm> <inputs.json> <out.bin> [start] [count]
//
// inputs.json : JSON array of circom input objects (numbers or decimal strings)
// out.bin     : count x (witnessSize*32) bytes, calculateBinWitness bodies back to back;
//               an instance that throws gets an all-zero body and an entry in out.bin.err.json
"use strict";
const fs = require("fs");
const path = require("path");
const refDir = process.env.B3W_REFERENCE_DIR || "/root/reference";
const builder = require(path.join(refDir, "blake3_nova_js/witness_calculator.js"));

async function main() {
  const [wasmPath, inPath, outPath, startS, countS] = process.argv.slice(2);
  const inputs = JSON.parse(fs.readFileSync(inPath, "utf8"));
  const start = startS ? parseInt(startS) : 0;
  const count = countS ? Math.min(parseInt(countS), inputs.length - start) : inputs.length - start;
  const realLog = console.log;
  const logs = [];
  console.log = (...a) => { logs.push(a.join(" ")); };      // nova circuits log "D_FLAGS:  0"
  const code = fs.readFileSync(wasmPath);
  let wc = await builder(code);
  const bodyLen = wc.witnessSize * wc.n32 * 4;
  const fd = fs.openSync(outPath, "w");
  const errs = {};
  const zero = Buffer.alloc(bodyLen);
  const nlogs = [];
  for (let i = 0; i < count; i++) {
    logs.length = 0;
    try {
      const body = await wc.calculateBinWitness(inputs[start + i], 0);
      fs.writeSync(fd, Buffer.from(body.buffer, body.byteOffset, body.byteLength));
    } catch (e) {
      errs[start + i] = String(e.message);
      fs.writeSync(fd, zero);
      wc = await builder(code);   // fresh instance: the reference never clears its errStr (witness_calculator.js:16,41)
    }
    nlogs.push(logs.length);
  }
  fs.closeSync(fd);
  fs.writeFileSync(outPath + ".err.json", JSON.stringify({ errors: errs, nlogs,
    witnessSize: wc.witnessSize, prime: wc.prime.toString(), version: wc.version }));
  console.log = realLog;
}
main().catch(e => { console.error(e); process.exit(1); });
