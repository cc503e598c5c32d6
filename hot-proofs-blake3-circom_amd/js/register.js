// register.js — module-path swap so the reference's OWN drivers run unchanged against this
// calculator:   node -r <repo>/hot-proofs-blake3-circom_amd/js/register.js \
//                    <reference>/build/blake3_compression/blake3_compression_js/generate_witness.js \
//                    circuit.wasm input.json out.wtns
// Every `require("./witness_calculator.js")` (generate_witness.js:1) and
// `require(".../witness_calculator")` then resolves to js/witness_calculator.js.
"use strict";
const Module = require("module");
const path = require("path");
const shim = path.join(__dirname, "witness_calculator.js");
const orig = Module._resolveFilename;
Module._resolveFilename = function (request, parent, ...rest) {
  if (/(^|[\\/])witness_calculator(\.js)?$/.test(request) && !(parent && parent.filename === shim)) return shim;
  return orig.call(this, request, parent, ...rest);
};
