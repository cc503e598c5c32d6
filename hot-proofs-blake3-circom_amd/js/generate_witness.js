// generate_witness.js — same CLI as the reference's (blake3_nova_js/generate_witness.js):
//   node generate_witness.js <file.wasm | circuit-name> <input.json> <output.wtns>
// against the native calculator.  (The reference's own copy also works unchanged: see register.js.)
"use strict";
const wc = require("./witness_calculator.js");
const { readFileSync, writeFileSync, existsSync } = require("fs");

if (process.argv.length != 5) {
  console.log("Usage: node generate_witness.js <file.wasm> <input.json> <output.wtns>");
} else {
  const input = JSON.parse(readFileSync(process.argv[3], "utf8"));
  const code = existsSync(process.argv[2]) ? readFileSync(process.argv[2]) : process.argv[2];
  wc(code).then(async (witnessCalculator) => {
    const buff = await witnessCalculator.calculateWTNSBin(input, 0);
    writeFileSync(process.argv[4], buff);
  }).catch((e) => { console.error(e.message); process.exit(1); });
}
