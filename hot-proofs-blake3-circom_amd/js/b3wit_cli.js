#!/usr/bin/env node
// b3wit_cli.js — command-line front end of the native calculator, argument-compatible with the
// reference's CLI (blake3_nova_js/generate_witness.js: <circuit.wasm> <input.json> <output.wtns>).
//   node b3wit_cli.js <circuit.wasm | circuit-name> <input.json> <output.wtns>
//   node b3wit_cli.js --batch <circuit> <records.json> <out-dir> [prefix]     many witnesses, streamed to files
// (The reference's own generate_witness.js also runs unchanged on top of this calculator: see register.js.)
"use strict";
const fs = require("fs");
const builder = require("./witness_calculator.js");

async function single(circuitArg, inputPath, outPath) {
  const code = fs.existsSync(circuitArg) ? fs.readFileSync(circuitArg) : circuitArg;
  const calc = await builder(code);
  const input = JSON.parse(fs.readFileSync(inputPath, "utf8"));
  fs.writeFileSync(outPath, await calc.calculateWTNSBin(input, 0));
}

async function batch(circuitArg, recordsPath, outDir, prefix) {
  const calc = await builder(circuitArg);
  const rows = JSON.parse(fs.readFileSync(recordsPath, "utf8"));      // array of u32 records
  const flat = Uint32Array.from([].concat(...rows));
  const res = await calc.calculateWitnessBatch(flat);
  fs.mkdirSync(outDir, { recursive: true });
  console.log(`${res.writeWtns(outDir, prefix || "witness_")} .wtns files written to ${outDir}`);
}

const argv = process.argv.slice(2);
let job;
if (argv[0] === "--batch" && argv.length >= 4) job = batch(argv[1], argv[2], argv[3], argv[4]);
else if (argv.length === 3) job = single(argv[0], argv[1], argv[2]);
else {
  console.log("Usage: node b3wit_cli.js <file.wasm|circuit> <input.json> <output.wtns>\n       node b3wit_cli.js --batch <circuit> <records.json> <out-dir> [prefix]");
  process.exit(2);
}
job.catch((e) => { console.error(e.message); process.exit(1); });
