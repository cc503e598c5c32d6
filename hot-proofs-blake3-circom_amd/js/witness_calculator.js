// witness_calculator.js — drop-in for the circom-emitted loader of the reference
// (blake3_nova_js/witness_calculator.js and its copies under build/**).  Same surface:
//
//   const builder = require("./witness_calculator.js");
//   const wc = await builder(codeOfTheCircuitWasm, options);
//   wc.version, wc.n32, wc.prime (BigInt), wc.witnessSize, wc.sanityCheck, wc.circom_version()
//   await wc.calculateWitness(input, sanityCheck)     -> BigInt[]
//   await wc.calculateBinWitness(input, sanityCheck)  -> Uint8Array (witnessSize*32, no header)
//   await wc.calculateWTNSBin(input, sanityCheck)     -> Uint8Array (.wtns v2 image)
//
// `code` is the circuit's .wasm bytes exactly as the reference's callers pass them
// (generate_witness.js:9-10, snarkjs wtns calculate, circomkit).  It is only hashed: the four
// committed circuits are recognised by sha256 and computed by the MI355X kernels behind the
// N-API addon (b3wit_napi.node -> libb3wit.so).  A circuit name ("compression", "nova_bn254",
// "nova_vesta", "nova_bn254_o1") is accepted in place of the bytes.  Bytes of any OTHER circom witness generator
// run through a generic WebAssembly loader written for this shim (wasm_fallback.js: the same calculator surface over the
// module's own exports), so swapping the reference loader for this file loses nothing for circuits outside this build's
// scope; options.wasmFallback === false refuses them instead.  The four BLAKE3 circuits never take that road (no CPU path
// for them) unless options.forceWasm asks for it (parity tests of the loader itself).
//
// Extensions the reference does not have: wc.calculateWitnessBatch(records) on packed u32 records.
"use strict";
const path = require("path");

const CIRCUITS = ["compression", "nova_bn254", "nova_vesta", "nova_bn254_o1"];
let addon = null;
function native() {
  if (!addon) addon = require(path.join(__dirname, "b3wit_napi.node"));
  return addon;
}

module.exports = async function builder(code, options) {
  options = options || {};
  if (options.forceWasm && typeof code !== "string") return require("./wasm_fallback.js")(code, options);
  const nat = native();
  let circuit;
  if (typeof code === "string") {
    circuit = CIRCUITS.indexOf(code);
  } else {
    circuit = nat.identifyWasm(code);
  }
  if (circuit < 0) {
    if (typeof code !== "string" && options.wasmFallback !== false) return require("./wasm_fallback.js")(code, options);
    throw new Error("b3wit: not one of the reference's committed BLAKE3 circuits" +
                    (typeof code === "string" ? "" : " (and options.wasmFallback is false)"));
  }
  const device = options.device === undefined ? parseInt(process.env.B3WIT_DEVICE || "0") : options.device;
  const handle = nat.create(circuit, device);
  // witness_calculator.js:66-75 — `sanityCheck = options`, always truthy for the default {}
  return new WitnessCalculator(handle, options, CIRCUITS[circuit]);
};
module.exports.CIRCUITS = CIRCUITS;
// Multi-GPU (one Node process per GPU, B3WIT_DEVICE = rank): rank 0 calls commUniqueId() and hands the 128 bytes to
// the other ranks; every rank then calls wc.joinRanks(id, rank, nranks).  After that a batch's
// allgatherPublic() returns the public outputs (h_out ...) of every rank's batch — RCCL over xGMI.
module.exports.commUniqueId = () => native().commUniqueId();

class WitnessCalculator {
  constructor(handle, sanityCheck, circuitName) {
    this.instance = handle;                 // opaque, as in the reference
    const info = native().info(handle);
    this.version = info.version[0];
    this.n32 = info.n32;
    this.prime = info.prime;
    this.witnessSize = info.witnessSize;
    this.inputSize = info.inputSize;
    this.publicWords = info.publicWords;
    this.sanityCheck = sanityCheck;
    this.circuit = circuitName;
    // The nova circuits log "D_FLAGS:  0" through console.log once per witness (circuits/blake3_nova.circom:166 via
    // writeBufferMessage, witness_calculator.js:44-58) — also when a later assert rejects the input, but not when
    // the first component (Blake3NovaTreePath_CheckDepth, blake3_nova.circom:201) already does.  Reproduced by
    // default; options.logDFlags === false silences it.
    this._logDFlags = circuitName !== "compression" && !(sanityCheck && sanityCheck.logDFlags === false);
    // options.strictErrorParity reproduces the loader's never-cleared errStr (witness_calculator.js:16,41): every
    // assert trace thrown on this calculator is prefixed by the traces of all earlier failures on it.
    this._strictErr = !!(sanityCheck && sanityCheck.strictErrorParity);
    this._errStr = "";
  }

  circom_version() {
    return this.version;
  }

  // witness_calculator.js:131-169, in the reference's EVALUATION ORDER: the keys are walked as Object.keys gives them; per key the
  // name is hashed (:138-140) and its size checked (:142-151), then its values are normalised and set one by one (:152-163) —
  // and the circuit runs INSIDE the setInputSignal call that sets the last missing input.  So an assert (or the D_FLAGS log line)
  // comes before anything is known about the keys behind the completing one, and a fault of an earlier key comes before the
  // circuit has run at all: { ...valid, m[0] = 2^34, zz: 1 } is "Assert Failed" in Bits34, { zz: 1, ...the same } is
  // "Too many values for input signal zz", and a valid nova step followed by an unknown key logs its line and THEN throws
  // (tests/golden/order.json, made by the reference loader).
  async _doCalculateWitness(input, sanityCheck, asWtns) {   // (asWtns: extension — the .wtns image instead of the bare body)
    const nat = native();
    const keys = Object.keys(input);
    const hashes = [], counts = [], values = [];
    let inputCounter = 0;
    let body = null;
    keys.forEach((k) => {
      const h = fnvHash(k);
      const hMSB = parseInt(h.slice(0, 8), 16);
      const hLSB = parseInt(h.slice(8, 16), 16);
      const fArr = flatArray(input[k]);
      const signalSize = nat.inputSignalSize(this.instance, hMSB, hLSB);
      if (signalSize < 0) {
        throw new Error(`Signal ${k} not found\n`);
      }
      if (fArr.length < signalSize) {
        throw new Error(`Not enough values for input signal ${k}\n`);
      }
      if (fArr.length > signalSize) {
        throw new Error(`Too many values for input signal ${k}\n`);
      }
      hashes.push(hMSB, hLSB);
      counts.push(fArr.length);
      for (let i = 0; i < fArr.length; i++) {
        values.push(normalize(fArr[i], this.prime));
        inputCounter++;
      }
      // the last missing input has just been set: the reference's circuit runs here, before the next key is looked at
      if (body === null && fArr.length > 0 && inputCounter === this.inputSize) {
        body = this._run(hashes, counts, values, asWtns);
      }
    });
    if (inputCounter < this.inputSize) {
      throw new Error(`Not all inputs have been set. Only ${inputCounter} out of ${this.inputSize}`);
    }
    return body;
  }

  // the device call for a complete set of inputs: throws the circuit's assert with the reference's text, logs what the circuit logs
  _run(hashes, counts, values, asWtns) {
    const nat = native();
    const vals = new Uint8Array(32 * values.length);
    const mask = BigInt(0xffffffff), s32 = BigInt(32);
    const dv = new DataView(vals.buffer);
    for (let i = 0; i < values.length; i++) {
      let v = values[i];
      for (let j = 0; j < 8; j++) {
        dv.setUint32(32 * i + 4 * j, Number(v & mask), true);
        v >>= s32;
      }
    }
    let body;
    try {
      body = nat.calcWitness(this.instance, Uint32Array.from(hashes), Uint32Array.from(counts), vals, asWtns === true);
    } catch (err) {
      if (err.status !== 4) throw err;
      const head = "Assert Failed.\n";
      const trace = err.message.startsWith(head) ? err.message.slice(head.length) : err.message;
      if (this._logDFlags && !/^Error in template (Num2Bits_\d+ line: \d+\nError in template (LessThan|GreaterEqThan)_\d+ line: \d+\nError in template )?Blake3NovaTreePath_CheckDepth_/.test(trace)) {
        console.log("D_FLAGS:  0");
      }
      const prior = this._strictErr ? this._errStr : "";
      this._errStr += trace;
      throw new Error("Error: " + head + prior + trace);
    }
    if (this._logDFlags) console.log("D_FLAGS:  0");
    return body;
  }

  async calculateWitness(input, sanityCheck) {
    const body = await this._doCalculateWitness(input, sanityCheck);
    const w = new Array(this.witnessSize);
    // 98 % of a witness of these circuits are bits and almost all the rest 32-bit words: those elements cost one lookup or one
    // BigInt(number) instead of eight shift-or steps (10.7 ms -> 0.6 ms per 24 093-element witness)
    const u32 = body.byteOffset % 4 === 0 ? new Uint32Array(body.buffer, body.byteOffset, body.byteLength >> 2)
                                          : new Uint32Array(body.slice().buffer);
    const s32 = BigInt(32), B0 = BigInt(0), B1 = BigInt(1);
    for (let i = 0, o = 0; i < this.witnessSize; i++, o += 8) {
      if ((u32[o + 1] | u32[o + 2] | u32[o + 3] | u32[o + 4] | u32[o + 5] | u32[o + 6] | u32[o + 7]) === 0) {
        const lo = u32[o];
        w[i] = lo === 0 ? B0 : lo === 1 ? B1 : BigInt(lo);
      } else {
        let v = B0;
        for (let j = 7; j >= 0; j--) v = (v << s32) | BigInt(u32[o + j]);
        w[i] = v;
      }
    }
    return w;
  }

  async calculateBinWitness(input, sanityCheck) {
    return await this._doCalculateWitness(input, sanityCheck);
  }

  async calculateWTNSBin(input, sanityCheck) {
    return await this._doCalculateWitness(input, sanityCheck, true);
  }

  // ---- extension: packed u32 records (h m t b d | nova step record), many witnesses per call.
  // Returns { n, publicOutputs: Uint32Array, status: Int32Array, fetch(i) -> Uint8Array body,
  //           writeWtns(dir, prefix) -> number of files }.
  async calculateWitnessBatch(records) {
    const nat = native();
    const r = nat.batchRun(this.instance, records);
    // The calculator owns ONE device batch: the next calculateWitnessBatch replaces it.  r.generation ties these
    // methods to the run that made r; on a replaced batch they throw ("stale batch result") instead of touching
    // the newer one with this one's sizes.
    const g = r.generation;
    r.fetch = (i) => nat.batchFetch(this.instance, i, g);
    // on-device check of every body of the batch: Uint32Array of mismatch counts (0 = valid witness)
    r.verify = () => nat.batchVerify(this.instance, g);
    // stream every witness of the batch to <dir>/<prefix><index>.wtns (same bytes as calculateWTNSBin)
    r.writeWtns = (dir, prefix, first, count) =>
      nat.batchWriteWtns(this.instance, first || 0, count === undefined ? r.n - (first || 0) : count, dir, prefix || "witness_", g);
    // every rank's public outputs of this batch (needs wc.joinRanks first; all ranks run the same batch size)
    r.allgatherPublic = () => nat.batchAllgatherPublic(this.instance, g);
    // Pedersen commitments of the batch's witnesses on the device (after wc.setCommitKey): { points, status }
    r.commit = () => nat.batchCommit(this.instance, g);
    // rank-1 constraint check of every witness of the batch on the device (after wc.loadR1cs): { violations, first }
    r.checkConstraints = () => nat.batchR1csCheck(this.instance, g);
    r.placement = nat.batchPlacement(this.instance);   // "mixed": the body buffer alternates two classes of HBM
    return r;
  }

  // ---- extension: the constraint system batch.checkConstraints() evaluates (what circom_tester's expectPass does with a
  // witness, test/blake3_hash.test.ts:36).  image: the bytes of an iden3 .r1cs file (.gz accepted) over this circuit's field
  // and witness size; omitted: the system this package derives for the circuit (all four committed builds).
  loadR1cs(image) {
    if (image === undefined) {
      const builtin = {compression: "blake3_compression.r1cs.gz", nova_bn254_o1: "blake3_nova_bn254_o1.r1cs.gz",
                       nova_bn254: "blake3_nova_bn254.r1cs.gz", nova_vesta: "blake3_nova_vesta.r1cs.gz"}[this.circuit];
      if (!builtin) throw new Error("no derived constraint system for " + this.circuit + ": pass the circuit's .r1cs");
      image = require("fs").readFileSync(path.join(__dirname, "..", "constraints", builtin));
    }
    if (image[0] === 0x1f && image[1] === 0x8b) image = require("zlib").gunzipSync(image);
    return native().r1csLoad(this.instance, new Uint8Array(image.buffer, image.byteOffset, image.length));
  }

  // ---- extension: commitment key for batch.commit().  curve: "bn254_g1" | "pallas" (the group over the --prime vesta circuit's scalar field); generators: Uint8Array with one
  // affine point per committed slot (x then y, 32-byte little-endian each), slots firstSlot .. witnessSize - 1;
  // windowBits: 12 | 16 | 18 (table size against speed, see include/b3wit.h), default automatic.
  // folded (optional Uint8Array, one byte per committed slot): `generators` are FOLDED ones (include/b3wit.h "FOLDED keys": slots
  // the circuit's constraints express through others are folded into those slots' generators — same points, fewer additions);
  // tools/fold_key.py derives both arrays from a circuit and a key.
  setCommitKey(curve, generators, firstSlot, windowBits, folded) {
    const id = {bn254_g1: 0, pallas: 1, vesta: 1}[curve];      // "vesta": older name of the Pallas curve id (after the circuit's prime)
    if (id === undefined) throw new Error("curve: bn254_g1 or pallas");
    native().commitKey(this.instance, id, firstSlot || 0, generators, windowBits || 0, folded || null);
  }

  // ---- extension: commitments of the witnesses of `records` (Uint32Array of whole input records, as for
  // calculateWitnessBatch) without producing the witnesses: { points, publicOutputs, status }
  commitRecords(records) {
    return native().commitRecords(this.instance, records);
  }

  // ---- extension: this calculator's GPU joins an RCCL communicator (see commUniqueId above)
  joinRanks(id, rank, nranks) {
    native().commCreate(this.instance, id, rank, nranks);
    this.rank = rank;
    this.nranks = nranks;
  }

  // ---- extension: the same over the host's shared memory instead of RCCL (b3w_comm_create_host): several Node processes on
  // ONE GPU, or a host without librccl.  name = "/unique-to-the-job", the same string on every rank.
  joinRanksHost(name, rank, nranks) {
    native().commCreateHost(this.instance, name, rank, nranks);
    this.rank = rank;
    this.nranks = nranks;
  }

  // ---- extension (nova circuits): chained mode.  preimage -> the step witnesses of every chunk path, what
  // rust_fold/src/main.rs:41-203 folds one step at a time.  Returns { nLeafSteps, nParentSteps, nChunks, pathLen,
  // placement, publicOutputs: Uint32Array (15 words per step: n_blocks_out block_count_out h_out[8] ...),
  // status: Int32Array, root: Uint32Array(8) = BLAKE3(preimage) as little-endian words, hash: hex string }.
  // After wc.joinRanks every rank passes the same preimage and gets the steps of ITS chunk range (firstChunk,
  // nChunksLocal); the chunk chaining values are all-gathered over RCCL so that each rank knows the whole tree, and so is every
  // step's h_out (the running chaining value the fold consumes, rust_fold/src/blake3_circuit.rs:111-123): r.hOutAll (8 words per
  // leaf step of ALL ranks, global step order: row 16 c + 15 is chunk c's chaining value), r.hOutParentsAll.
  async foldPreimage(preimage, opts) {
    opts = opts || {};
    // opts.commitOnly (after setCommitKey): r.commitments = one 64-byte point per step, no witness bodies written
    // opts.checkConstraints (after loadR1cs): r.violations = per step, the constraints of the step circuit its witness violates
    // (0 everywhere = every step of the fold is a valid witness), checked on the device while the bodies sit in the ring
    const r = native().chainFold(this.instance, preimage, opts.batchSteps || 16384, opts.ring || 2, opts.withParents !== false,
                                 !!opts.commitOnly, !!opts.checkConstraints);
    const b = Buffer.alloc(32);
    r.root.forEach((w, i) => b.writeUInt32LE(w, 4 * i));
    r.hash = b.toString("hex");
    return r;
  }
}

// helpers with the reference's semantics (witness_calculator.js:303-337)
function flatArray(a) {
  const res = [];
  (function fill(x) {
    if (Array.isArray(x)) for (let i = 0; i < x.length; i++) fill(x[i]);
    else res.push(x);
  })(a);
  return res;
}

function normalize(n, prime) {
  let res = BigInt(n) % prime;
  if (res < 0) res += prime;
  return res;
}

function fnvHash(str) {
  const m64 = (BigInt(1) << BigInt(64)) - BigInt(1);
  let hash = BigInt("0xCBF29CE484222325");
  for (let i = 0; i < str.length; i++) {
    hash ^= BigInt(str.charCodeAt(i));
    hash = (hash * BigInt(0x100000001B3)) & m64;
  }
  return hash.toString(16).padStart(16, "0");
}
