// Build-container-only tool: the reference's WASM witness generator timed on ALL host cores (SURVEY.md 8(d)(i); the only timing
// the reference itself has is the print at test/witness_gen.test.ts:43-50).  One `node` worker PROCESS per core; each loads the
// circuit's committed .wasm through the reference's own witness_calculator.js (from --reference-dir at run time: nothing of the
// reference is copied into this repository) and loops calculateWTNSBin over the config's inputs, a fresh input every call, for
// `seconds` after `warmup` calls.  Prints ONE JSON object: aggregate and per-core witnesses/s, core count, CPU model, node version.
//
//   node tools/wasm_baseline.js --reference-dir /root/reference --wasm <circuit.wasm> --inputs <inputs.json> [--seconds 30]
//                               [--workers N] [--warmup 5]
// Driven by tools/wasm_baseline.py (which writes the inputs of BASELINE configs 2 and 3 and records the result under profiles/).
"use strict";
const fs = require("fs");
const os = require("os");
const path = require("path");
const { fork } = require("child_process");

function arg(name, dflt) {
  const i = process.argv.indexOf("--" + name);
  return i >= 0 && i + 1 < process.argv.length ? process.argv[i + 1] : dflt;
}

async function worker() {
  const refDir = arg("reference-dir", "/root/reference");
  const builder = require(path.join(refDir, "blake3_nova_js/witness_calculator.js"));
  const inputs = JSON.parse(fs.readFileSync(arg("inputs"), "utf8"));
  const seconds = parseFloat(arg("seconds", "30")), warmup = parseInt(arg("warmup", "5"));
  const index = parseInt(arg("worker-index", "0")), nworkers = parseInt(arg("workers", "1"));
  console.log = () => {};                                  // the nova circuits log "D_FLAGS:  0" on every call
  const wc = await builder(fs.readFileSync(arg("wasm")));
  let k = (index * Math.ceil(inputs.length / nworkers)) % inputs.length, bytes = 0;
  for (let i = 0; i < warmup; i++) { await wc.calculateWTNSBin(inputs[k], 0); k = (k + 1) % inputs.length; }
  process.send({ ready: true });
  await new Promise((resolve) => process.once("message", resolve));      // all workers start their timed loop together
  const t0 = process.hrtime.bigint();
  let n = 0, t1 = t0;
  const limit = BigInt(Math.round(seconds * 1e9));
  while (t1 - t0 < limit) {
    const img = await wc.calculateWTNSBin(inputs[k], 0);
    bytes = img.length;
    k = (k + 1) % inputs.length;
    n++;
    t1 = process.hrtime.bigint();
  }
  process.send({ done: true, n, seconds: Number(t1 - t0) / 1e9, wtns_bytes: bytes });
  process.exit(0);
}

function cpuModel() {
  const c = os.cpus();
  return c.length ? c[0].model : "unknown";
}

function usableCores() {
  // a container's share (cgroup v2 cpu.max) and the affinity mask bound the worker count, not os.cpus() alone
  let n = os.cpus().length;
  try {
    const [q, per] = fs.readFileSync("/sys/fs/cgroup/cpu.max", "utf8").trim().split(/\s+/);
    if (q !== "max") n = Math.max(1, Math.min(n, Math.ceil(parseFloat(q) / parseFloat(per))));
  } catch (e) { /* no cgroup v2 */ }
  try {
    const m = /Cpus_allowed_list:\s*(\S+)/.exec(fs.readFileSync("/proc/self/status", "utf8"));
    if (m) {
      let a = 0;
      for (const part of m[1].split(",")) { const [lo, hi] = part.split("-").map(Number); a += (hi === undefined ? lo : hi) - lo + 1; }
      if (a > 0) n = Math.min(n, a);
    }
  } catch (e) { /* not Linux */ }
  return n;
}

async function main() {
  if (process.argv.includes("--worker-index")) return worker();
  const nworkers = parseInt(arg("workers", String(usableCores())));
  const kids = [];
  let ready = 0;
  const results = [];
  await new Promise((resolve, reject) => {
    for (let i = 0; i < nworkers; i++) {
      const kid = fork(__filename, [...process.argv.slice(2), "--workers", String(nworkers), "--worker-index", String(i)]);
      kids.push(kid);
      kid.on("message", (m) => {
        if (m.ready && ++ready === nworkers) for (const q of kids) q.send({ go: true });
        if (m.done) { results.push(m); if (results.length === nworkers) resolve(); }
      });
      kid.on("exit", (code) => { if (code !== 0) reject(new Error("worker " + i + " exited with " + code)); });
    }
  });
  const total = results.reduce((a, r) => a + r.n, 0), wall = Math.max(...results.map((r) => r.seconds));
  const out = {
    value: total / wall, unit: "witnesses/s", per_core: total / wall / nworkers, cores: nworkers, cpu: cpuModel(), node: process.version,
    witnesses: total, seconds: wall, per_worker: results.map((r) => r.n), wtns_bytes: results[0].wtns_bytes,
    wasm: arg("wasm"), surface: "calculateWTNSBin of the reference's blake3_nova_js/witness_calculator.js, a fresh input every call",
  };
  process.stdout.write(JSON.stringify(out) + "\n");
}
main().catch((e) => { console.error(e); process.exit(1); });
