// Build-container-only tool (like tools/wasm_oracle.js, tools/gen_sequence_golden.js): runs the REFERENCE's loader
// (blake3_nova_js/witness_calculator.js + the committed WASMs, loaded from /root/reference at run time) on inputs whose
// KEY ORDER matters, and records what it does: console lines, error text, sha256 of the body.
//
// What is pinned (witness_calculator.js:131-169): the keys are walked in Object.keys order; per key — size check, then
// setInputSignal per value — and the circuit RUNS inside the call that sets the last missing input.  So
//   * a fault of a key BEFORE the completing one wins over anything the circuit would say,
//   * the circuit's assert (or its D_FLAGS log line) comes BEFORE anything is known about keys behind the completing one,
//   * a trailing unknown key with no values ([]) is not a fault at all.
// Each case is a list of [key, value] PAIRS (an object's key order would not survive every JSON tool); "calls" cases are
// several inputs on ONE calculator, for the never-cleared errStr (witness_calculator.js:16,41).
//
//   node tools/gen_order_golden.js > tests/golden/order.json
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const refDir = process.env.B3W_REFERENCE_DIR || "/root/reference";
const builder = require(path.join(refDir, "blake3_nova_js/witness_calculator.js"));
const root = path.dirname(__dirname);

const WASM = {
  compression: "build/blake3_compression/blake3_compression_js/blake3_compression.wasm",
  nova_bn254: "build/blake3_nova_js/blake3_nova.wasm",
  nova_vesta: "build/blake3_nova_pasta_js/blake3_nova_pasta.wasm",
  nova_bn254_o1: "build/blake3_nova/blake3_nova_js/blake3_nova.wasm",
};
const P32 = "4294967296", P34 = "17179869184";

function pairsOf(input, order) {
  const keys = order || Object.keys(input);
  return keys.map((k) => [k, JSON.parse(JSON.stringify(input[k]))]);
}
const set = (pairs, k, f) => pairs.map(([kk, v]) => (kk === k ? [kk, f(JSON.parse(JSON.stringify(v)))] : [kk, v]));
const drop = (pairs, k) => pairs.filter(([kk]) => kk !== k);
const first = (pairs, k) => [pairs.find(([kk]) => kk === k)].concat(drop(pairs, k));
const last = (pairs, k) => drop(pairs, k).concat([pairs.find(([kk]) => kk === k)]);
const toObj = (pairs) => { const o = {}; for (const [k, v] of pairs) o[k] = v; return o; };

function compressionCases(gold) {
  const by = {};
  for (const c of gold.cases) by[c.name] = c;
  const ok = pairsOf(by.config2_0.input);                              // h m t b d
  const m34 = set(ok, "m", (m) => { m[0] = P34; return m; });           // asserts in Bits34
  const h32 = set(ok, "h", (h) => { h[0] = P32; return h; });           // asserts in ToBits (feed-forward XOR)
  const c = [];
  c.push(["order_valid", ok]);
  c.push(["order_valid_reversed_keys", ok.slice().reverse()]);
  c.push(["order_valid_m_last", last(ok, "m")]);
  c.push(["order_assert_then_unknown", m34.concat([["zz", 1]])]);
  c.push(["order_unknown_then_assert", [["zz", 1]].concat(m34)]);
  c.push(["order_unknown_middle_then_assert", m34.slice(0, 2).concat([["zz", 1]], m34.slice(2))]);
  c.push(["order_valid_then_unknown", ok.concat([["zz", 1]])]);
  c.push(["order_unknown_then_valid", [["zz", 1]].concat(ok)]);
  c.push(["order_valid_then_unknown_empty", ok.concat([["zz", []]])]);
  c.push(["order_unknown_empty_then_valid", [["zz", []]].concat(ok)]);
  c.push(["order_valid_then_two_unknown", ok.concat([["zz", []], ["yy", [1, 2]]])]);
  c.push(["order_short_key_then_assert", first(set(m34, "h", (h) => h.slice(0, 7)), "h")]);
  c.push(["order_assert_value_then_short_last_key", last(set(h32, "d", () => []), "d")]);
  c.push(["order_assert_value_then_long_last_key", last(set(h32, "d", (d) => [d, d]), "d")]);
  c.push(["order_assert_value_missing_key", drop(m34, "b")]);
  c.push(["order_assert_value_missing_key_then_unknown", drop(m34, "b").concat([["zz", 1]])]);
  c.push(["order_assert_in_completing_key", last(m34, "m").concat([["zz", 1]])]);
  c.push(["order_two_asserting_values_then_unknown", set(m34, "h", (h) => { h[0] = P32; return h; }).concat([["zz", 1]])]);
  return c;
}

function novaCases(gold) {
  const by = {};
  for (const c of gold.cases) by[c.name] = c;
  const ok = pairsOf(by.config3_0.input);
  const par = pairsOf(by.config3_3.input);
  const depth = pairsOf(by.err_depth_ge_leaf.input);                    // rejected by CheckDepth: no log line
  const b32 = pairsOf(by.err_b_2p32.input);                             // asserts past GetFlag: log line, then the assert
  const c = [];
  c.push(["order_valid", ok]);
  c.push(["order_valid_reversed_keys", ok.slice().reverse()]);
  c.push(["order_valid_then_unknown", ok.concat([["zz", 1]])]);
  c.push(["order_parent_then_unknown", par.concat([["zz", [1, 2, 3]]])]);
  c.push(["order_unknown_then_valid", [["zz", 1]].concat(ok)]);
  c.push(["order_valid_then_unknown_empty", ok.concat([["zz", []]])]);
  c.push(["order_unknown_empty_then_valid", [["zz", []]].concat(ok)]);
  c.push(["order_checkdepth_then_unknown", depth.concat([["zz", 1]])]);
  c.push(["order_unknown_then_checkdepth", [["zz", 1]].concat(depth)]);
  c.push(["order_assert_then_unknown", b32.concat([["zz", 1]])]);
  c.push(["order_unknown_then_assert", [["zz", 1]].concat(b32)]);
  c.push(["order_short_key_then_checkdepth", first(set(depth, "m", (m) => m.slice(0, 15)), "m")]);
  c.push(["order_checkdepth_value_then_long_last_key", last(set(depth, "b", (b) => [b, b]), "b")]);
  c.push(["order_checkdepth_value_missing_key_then_unknown", drop(depth, "h").concat([["zz", 1]])]);
  c.push(["order_assert_in_completing_key", last(b32, "b").concat([["zz", 1]])]);
  return c;
}

// several inputs on ONE calculator: the error text of call k carries the traces of every earlier assert on it
function callSequences(circuit, cases) {
  const by = {};
  for (const [n, p] of cases) by[n] = p;
  if (circuit === "compression")
    return [["calls_two_faults", ["order_assert_then_unknown", "order_valid_then_unknown", "order_two_asserting_values_then_unknown",
                                  "order_unknown_then_assert", "order_valid", "order_assert_in_completing_key"].map((n) => [n, by[n]])]];
  return [["calls_two_faults", ["order_assert_then_unknown", "order_valid_then_unknown", "order_checkdepth_then_unknown",
                                "order_unknown_then_assert", "order_valid", "order_assert_in_completing_key",
                                "order_checkdepth_value_missing_key_then_unknown"].map((n) => [n, by[n]])]];
}

async function runOne(wc, pairs) {
  const real = console.log, logs = [];
  console.log = (...a) => logs.push(a.join(" "));
  let error = null, sha = null;
  try {
    const body = await wc.calculateBinWitness(toObj(pairs), 0);
    sha = crypto.createHash("sha256").update(body).digest("hex");
  } catch (e) { error = e.message; }
  console.log = real;
  return { logs, error, body_sha256: sha };
}

async function main() {
  const out = { generated_by: "tools/gen_order_golden.js: the reference's witness_calculator.js over its committed WASMs", circuits: {} };
  for (const circuit of Object.keys(WASM)) {
    const gold = JSON.parse(fs.readFileSync(path.join(root, "tests/golden", circuit + ".json")));
    const code = fs.readFileSync(path.join(refDir, WASM[circuit]));
    const cases = circuit === "compression" ? compressionCases(gold) : novaCases(gold);
    const rec = { cases: [], calls: [] };
    for (const [name, pairs] of cases) {
      const wc = await builder(code);                                   // a fresh calculator per case: only this call's trace
      rec.cases.push(Object.assign({ name, pairs }, await runOne(wc, pairs)));
    }
    for (const [name, seq] of callSequences(circuit, cases)) {
      const wc = await builder(code);
      const steps = [];
      for (const [n, pairs] of seq) steps.push(Object.assign({ name: n, pairs }, await runOne(wc, pairs)));
      rec.calls.push({ name, steps });
    }
    out.circuits[circuit] = rec;
  }
  console.log(JSON.stringify(out, null, 1));
}
main().catch((e) => { console.error(e); process.exit(1); });
