// wasm_fallback.js — generic calculator for circom-compiled WebAssembly the native path does not know.
//
// The shim (witness_calculator.js) computes the reference's four committed BLAKE3 circuits on the GPU and recognises
// them by the sha256 of their bytes.  Any OTHER circom 2.x witness generator handed to builder(code, options) is run the
// way the reference's loader runs it (blake3_nova_js/witness_calculator.js:1-274): instantiate the module with the four
// `runtime` callbacks, then drive its exports (init / getInputSignalSize / writeSharedRWMemory / setInputSignal /
// getWitness / readSharedRWMemory / getRawPrime / getMessageChar ...).  Same surface, same argument meaning, same error
// text — so a caller that swaps the reference loader for the shim loses nothing for circuits outside this build's scope.
// Written against the circom 2.1 WASM ABI as the reference's loader uses it; this is host-side JavaScript only, no
// part of the GPU product path (the four BLAKE3 circuits never come through here unless options.forceWasm is set,
// which the parity tests use to compare this loader with the reference's goldens).
"use strict";

const CIRCOM_ERRORS = ["Signal not found.\n", "Too many signals set.\n", "Signal already set.\n", "Assert Failed.\n",
                       "Not enough memory.\n", "Input signal array access exceeds the size.\n"];

// little-endian u32 limbs (as the shared read/write window hands them out) <-> BigInt
function limbsToBigInt(readLimb, n32) {
  let v = BigInt(0);
  for (let j = n32 - 1; j >= 0; j--) v = (v << BigInt(32)) | BigInt(readLimb(j) >>> 0);
  return v;
}

class GenericWitnessCalculator {
  constructor(instance, sanityCheck) {
    const x = instance.exports;
    this.instance = instance;
    this.version = x.getVersion();
    this.n32 = x.getFieldNumLen32();
    x.getRawPrime();
    this.prime = limbsToBigInt((j) => x.readSharedRWMemory(j), this.n32);
    this.witnessSize = x.getWitnessSize();
    this.sanityCheck = sanityCheck;
  }

  circom_version() {
    return this.instance.exports.getVersion();
  }

  // one pass over the input object: name hash, size check, flatten, reduce mod p, hand over limb by limb (WC:131-169)
  async _doCalculateWitness(input, sanityCheck) {
    const x = this.instance.exports;
    x.init(this.sanityCheck || sanityCheck ? 1 : 0);
    let fed = 0;
    const mask = BigInt(0xffffffff), s32 = BigInt(32);
    for (const name of Object.keys(input)) {
      const h = fnv1a64(name);
      const hi = Number(h >> s32), lo = Number(h & mask);
      const vals = flatten(input[name]);
      const want = x.getInputSignalSize(hi, lo);
      if (want < 0) throw new Error(`Signal ${name} not found\n`);
      if (vals.length < want) throw new Error(`Not enough values for input signal ${name}\n`);
      if (vals.length > want) throw new Error(`Too many values for input signal ${name}\n`);
      for (let i = 0; i < vals.length; i++) {
        let v = BigInt(vals[i]) % this.prime;
        if (v < 0) v += this.prime;
        for (let j = 0; j < this.n32; j++) {
          x.writeSharedRWMemory(j, Number(v & mask));
          v >>= s32;
        }
        try {
          x.setInputSignal(hi, lo, i);
        } catch (err) {
          throw new Error(err);                    // "Error: Assert Failed.\n..." — the text callers match on
        }
        fed++;
      }
    }
    const total = x.getInputSize();
    if (fed < total) throw new Error(`Not all inputs have been set. Only ${fed} out of ${total}`);
  }

  async calculateWitness(input, sanityCheck) {
    await this._doCalculateWitness(input, sanityCheck);
    const x = this.instance.exports;
    const w = new Array(this.witnessSize);
    for (let i = 0; i < this.witnessSize; i++) {
      x.getWitness(i);
      w[i] = limbsToBigInt((j) => x.readSharedRWMemory(j), this.n32);
    }
    return w;
  }

  _fill(words, at) {
    const x = this.instance.exports;
    for (let i = 0; i < this.witnessSize; i++) {
      x.getWitness(i);
      for (let j = 0; j < this.n32; j++) words[at++] = x.readSharedRWMemory(j);
    }
  }

  async calculateBinWitness(input, sanityCheck) {
    const words = new Uint32Array(this.witnessSize * this.n32);
    await this._doCalculateWitness(input, sanityCheck);
    this._fill(words, 0);
    return new Uint8Array(words.buffer);
  }

  // .wtns v2 image: "wtns" | 2 | 2 sections | (1, len 8 + n8: n8, prime, nWitness) | (2, len n8 * nWitness: the elements)
  async calculateWTNSBin(input, sanityCheck) {
    const n32 = this.n32, n8 = 4 * n32;
    const words = new Uint32Array(this.witnessSize * n32 + n32 + 11);
    await this._doCalculateWitness(input, sanityCheck);
    const x = this.instance.exports;
    words[0] = 0x736e7477;                         // "wtns"
    words[1] = 2; words[2] = 2;
    words[3] = 1; words[4] = 8 + n8; words[5] = 0;
    words[6] = n8;
    x.getRawPrime();
    for (let j = 0; j < n32; j++) words[7 + j] = x.readSharedRWMemory(j);
    let at = 7 + n32;
    words[at++] = this.witnessSize;
    words[at++] = 2;
    const len = n8 * this.witnessSize;             // the reference's hex-slice arithmetic gives (len, 0) below 2^32 too
    words[at++] = len >>> 0; words[at++] = Math.floor(len / 4294967296);
    this._fill(words, at);
    return new Uint8Array(words.buffer);
  }
}

function flatten(a) {
  const out = [];
  const walk = (v) => { if (Array.isArray(v)) v.forEach(walk); else out.push(v); };
  walk(a);
  return out;
}

function fnv1a64(s) {
  const m64 = (BigInt(1) << BigInt(64)) - BigInt(1), prime = BigInt("0x100000001b3");
  let h = BigInt("0xcbf29ce484222325");
  for (let i = 0; i < s.length; i++) h = ((h ^ BigInt(s.charCodeAt(i))) * prime) & m64;
  return h;
}

// builder(code, options) for arbitrary circom WASM (WC:1-106)
module.exports = async function genericBuilder(code, options) {
  options = options || {};
  let mod;
  try {
    mod = await WebAssembly.compile(code);
  } catch (err) {
    console.log(err);
    console.log("\nTry to run circom --c in order to generate c++ code instead\n");
    throw new Error(err);
  }
  let instance = null;
  let trace = "";                                  // circuit error lines; like the reference's, never cleared
  let line = "";                                   // pieces of one log() call until its "\n"
  const message = () => {
    let s = "";
    for (let c = instance.exports.getMessageChar(); c !== 0; c = instance.exports.getMessageChar()) s += String.fromCharCode(c);
    return s;
  };
  const piece = (s) => { line = line === "" ? s : line + " " + s; };
  instance = await WebAssembly.instantiate(mod, {
    runtime: {
      exceptionHandler(codeNo) {
        throw new Error((CIRCOM_ERRORS[codeNo - 1] || "Unknown error.\n") + trace);
      },
      printErrorMessage() { trace += message() + "\n"; },
      writeBufferMessage() {
        const s = message();
        if (s === "\n") { console.log(line); line = ""; } else piece(s);
      },
      showSharedRWMemory() {
        const n32 = instance.exports.getFieldNumLen32();
        piece(limbsToBigInt((j) => instance.exports.readSharedRWMemory(j), n32).toString());
      },
    },
  });
  return new GenericWitnessCalculator(instance, options);
};
module.exports.GenericWitnessCalculator = GenericWitnessCalculator;
