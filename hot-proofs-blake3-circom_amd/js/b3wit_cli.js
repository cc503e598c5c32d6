#!/usr/bin/env node
// b3wit_cli.js — command-line front end of the native calculator, argument-compatible with the
// reference's CLI (blake3_nova_js/generate_witness.js: <circuit.wasm> <input.json> <output.wtns>).
//   node b3wit_cli.js <circuit.wasm | circuit-name> <input.json> <output.wtns>
//   node b3wit_cli.js --batch <circuit> <records.json> <out-dir> [prefix]     many witnesses, streamed to files
//   node b3wit_cli.js --fold <nova-circuit> <preimage-file> [public.json]     chained mode: every step witness of every
//                                                                             chunk path; prints BLAKE3(preimage)
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

async function fold(circuitArg, preimagePath, outPath) {
  const calc = await builder(circuitArg, { logDFlags: false });
  const t0 = process.hrtime.bigint();
  const r = await calc.foldPreimage(fs.readFileSync(preimagePath));
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  const bad = r.status.reduce((a, x) => a + (x !== 0 ? 1 : 0), 0);
  console.log(`${r.nLeafSteps} leaf + ${r.nParentSteps} parent step witnesses over ${r.nChunks} chunks in ${ms.toFixed(1)} ms (${bad} rejected), blake3 = ${r.hash}`);
  if (outPath) fs.writeFileSync(outPath, JSON.stringify({ hash: r.hash, nLeafSteps: r.nLeafSteps, nParentSteps: r.nParentSteps, publicOutputs: Array.from(r.publicOutputs) }));
  if (bad) process.exit(1);
}

const argv = process.argv.slice(2);
let job;
if (argv[0] === "--batch" && argv.length >= 4) job = batch(argv[1], argv[2], argv[3], argv[4]);
else if (argv[0] === "--fold" && argv.length >= 3) job = fold(argv[1], argv[2], argv[3]);
else if (argv.length === 3) job = single(argv[0], argv[1], argv[2]);
else {
  console.log("Usage: node b3wit_cli.js <file.wasm|circuit> <input.json> <output.wtns>\n       node b3wit_cli.js --batch <circuit> <records.json> <out-dir> [prefix]\n       node b3wit_cli.js --fold <nova-circuit> <preimage-file> [public.json]");
  process.exit(2);
}
job.catch((e) => { console.error(e.message); process.exit(1); });
