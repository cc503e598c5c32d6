//! Source-only FFI stub (no Rust toolchain in the build image: not compiled here; its extern declarations are held
//! against include/b3wit.h by tests/test_rust_ffi_decls_cpu.py).
//! Drop-in for `circom_scotia::calculate_witness(&cfg, input, true)` at
//! rust_fold/src/blake3_circuit.rs:305 of banyancomputer/hot-proofs-blake3-circom, over the C-ABI of
//! libb3wit.so (include/b3wit.h).  The stale `override_h_to_IV` input (blake3_circuit.rs:260-265,285) must
//! not be passed: no committed circuit declares it.
use std::ffi::c_void;
use std::os::raw::c_char;

#[link(name = "b3wit")]
extern "C" {
    fn b3w_create(circuit: i32, device: i32, out: *mut *mut c_void) -> i32;
    fn b3w_destroy(ctx: *mut c_void);
    fn b3w_info(ctx: *const c_void, n32: *mut u32, prime_le: *mut u8, witness_size: *mut u32,
                input_size: *mut u32, version: *mut u32) -> i32;
    fn b3w_calc_witness(ctx: *mut c_void, name_hashes: *const u64, counts: *const u32,
                        values_le32: *const u8, nkeys: u32, out_body: *mut u8) -> i32;
    fn b3w_last_error(ctx: *const c_void, buf: *mut c_char, len: usize) -> i32;
}

pub const CIRCUIT_NOVA_BN254: i32 = 1; // build/blake3_nova_js/blake3_nova.wasm (rust_fold/src/main.rs:29)
pub const CIRCUIT_NOVA_VESTA: i32 = 2; // build/blake3_nova_pasta_js/blake3_nova_pasta.wasm (main.rs:364)
pub const CURVE_BN254_G1: i32 = 0;     // the group arecibo's Bn256Engine commits in (scalar field = the bn128 circuits' prime)
pub const CURVE_PALLAS: i32 = 1;       // PallasEngine (rust_fold/src/main.rs:366): scalar field = circom's "--prime vesta"

/// FNV-1a 64 of the signal name, as witness_calculator.js:325-337 / circom's WASM runtime key inputs.
pub fn fnv1a64(name: &str) -> u64 {
    let mut h: u64 = 0xCBF29CE484222325;
    for b in name.bytes() {
        h ^= b as u64;
        h = h.wrapping_mul(0x100000001B3);
    }
    h
}

pub struct Calculator { ctx: *mut c_void, witness_size: usize }

impl Calculator {
    pub fn new(circuit: i32, device: i32) -> Result<Self, i32> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { b3w_create(circuit, device, &mut ctx) };
        if rc != 0 { return Err(rc); }
        let mut nwit = 0u32;
        unsafe { b3w_info(ctx, std::ptr::null_mut(), std::ptr::null_mut(), &mut nwit, std::ptr::null_mut(), std::ptr::null_mut()) };
        Ok(Calculator { ctx, witness_size: nwit as usize })
    }

    /// `input`: (signal name, values as 32-byte little-endian canonical field elements), any key order.
    /// Returns the witness as `witness_size` 32-byte little-endian elements (F::from_repr each).
    pub fn calculate_witness(&mut self, input: &[(String, Vec<[u8; 32]>)]) -> Result<Vec<[u8; 32]>, String> {
        let hashes: Vec<u64> = input.iter().map(|(k, _)| fnv1a64(k)).collect();
        let counts: Vec<u32> = input.iter().map(|(_, v)| v.len() as u32).collect();
        let values: Vec<u8> = input.iter().flat_map(|(_, v)| v.iter().flat_map(|x| x.iter().copied())).collect();
        let mut body = vec![0u8; self.witness_size * 32];
        let rc = unsafe {
            b3w_calc_witness(self.ctx, hashes.as_ptr(), counts.as_ptr(), values.as_ptr(), input.len() as u32, body.as_mut_ptr())
        };
        if rc != 0 {
            let mut buf = vec![0 as c_char; 1024];
            unsafe { b3w_last_error(self.ctx, buf.as_mut_ptr(), buf.len()) };
            let msg = unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
            return Err(format!("status {}: {}", rc, msg));
        }
        Ok(body.chunks_exact(32).map(|c| { let mut a = [0u8; 32]; a.copy_from_slice(c); a }).collect())
    }
}

impl Drop for Calculator {
    fn drop(&mut self) { unsafe { b3w_destroy(self.ctx) } }
}

// ---- the chained pass: what replaces the loop at rust_fold/src/main.rs:166-179 (prove_step per step, each calling
// calculate_witness) with ONE pass that produces every step witness of a chunk range on the device and hands each batch to a
// consumer while it sits in HBM.  Parameter names, order and types of every declaration below are held against include/b3wit.h
// by tests/test_rust_ffi_decls_cpu.py; none of this file can be compiled in the build image (no Rust toolchain).

/// b3w_batch_consumer: (user, d_bodies, pitch, first_step, count, stream).  `d_bodies` is a DEVICE pointer to `count` witness
/// bodies `pitch` bytes apart (steps first_step .. first_step + count of this rank's pass); the callee must enqueue its work
/// on `stream` (a hipStream_t) — the buffer is overwritten `ring` batches later.
pub type BatchConsumer = extern "C" fn(user: *mut c_void, d_bodies: *const u8, pitch: u64, first_step: u64, count: u32, stream: *mut c_void);
/// b3w_allgather_fn: the caller's own collective for b3w_comm_create_external (device pointers; 0 = done).
pub type AllgatherFn = extern "C" fn(user: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: u64, stream: *mut c_void) -> i32;

#[link(name = "b3wit")]
extern "C" {
    pub fn b3w_chain_create(ctx: *mut c_void, preimage_len: u64, first_chunk: u64, n_chunks_local: u32, batch_steps: u32,
                            ring: u32, with_parents: i32, out: *mut *mut c_void) -> i32;
    pub fn b3w_chain_destroy(chain: *mut c_void);
    pub fn b3w_chain_run_leaves(chain: *mut c_void, host_preimage: *const u8,
                                consumer: Option<extern "C" fn(*mut c_void, *const u8, u64, u64, u32, *mut c_void)>,
                                user: *mut c_void, stream: *mut c_void) -> i32;
    pub fn b3w_chain_run_parents(chain: *mut c_void, d_all_chunk_cvs: *const u32,
                                 consumer: Option<extern "C" fn(*mut c_void, *const u8, u64, u64, u32, *mut c_void)>,
                                 user: *mut c_void, stream: *mut c_void) -> i32;
    pub fn b3w_chain_info(chain: *const c_void, n_leaf_steps: *mut u64, n_parent_steps: *mut u64, n_chunks: *mut u64, path_len: *mut u32,
                          placement: *mut i32) -> i32;
    pub fn b3w_chain_outputs(chain: *mut c_void, host_public: *mut u32, host_status: *mut i32, host_root: *mut u32,
                             stream: *mut c_void) -> i32;
    // sharded over ranks (one process per GPU): this rank's chunk range, the two exchanges (chunk chaining values; every step's
    // h_out = z_{i+1} of blake3_circuit.rs:111-123 in global step order on every rank), the communicator's three transports
    pub fn b3w_chain_shard(n_chunks: u64, rank: i32, nranks: i32, first_chunk: *mut u64, n_chunks_local: *mut u32);
    pub fn b3w_chain_num_chunks(preimage_len: u64) -> u64;
    pub fn b3w_chain_num_leaf_steps(preimage_len: u64) -> u64;
    pub fn b3w_chain_parent_row(chunk: u64, n_chunks: u64) -> u64;
    pub fn b3w_chain_run_parents_sharded(chain: *mut c_void, comm: *mut c_void,
                                         consumer: Option<extern "C" fn(*mut c_void, *const u8, u64, u64, u32, *mut c_void)>,
                                         user: *mut c_void, stream: *mut c_void) -> i32;
    pub fn b3w_chain_allgather_hout(chain: *mut c_void, comm: *mut c_void, d_leaf_hout: *mut u32, d_parent_hout: *mut u32,
                                    stream: *mut c_void) -> i32;
    pub fn b3w_chain_allgather_hout_host(chain: *mut c_void, comm: *mut c_void, host_leaf_hout: *mut u32, host_parent_hout: *mut u32,
                                         stream: *mut c_void) -> i32;
    pub fn b3w_comm_unique_id(id: *mut u8) -> i32;
    pub fn b3w_comm_create(ctx: *mut c_void, id: *const u8, rank: i32, nranks: i32, out: *mut *mut c_void) -> i32;
    pub fn b3w_comm_create_host(ctx: *mut c_void, name: *const c_char, rank: i32, nranks: i32, out: *mut *mut c_void) -> i32;
    pub fn b3w_comm_create_external(ctx: *mut c_void, rank: i32, nranks: i32,
                                    allgather: Option<extern "C" fn(*mut c_void, *const c_void, *mut c_void, u64, *mut c_void) -> i32>,
                                    user: *mut c_void, out: *mut *mut c_void) -> i32;
    pub fn b3w_comm_destroy(comm: *mut c_void);
    // the witness commitment (main.rs:166-179 -> prove_step commits to W) and the constraint check (utils.rs:17-88) as consumers
    pub fn b3w_commit_key_create_ex(ctx: *mut c_void, curve: i32, first_slot: u32, host_generators: *const u8, window_bits: u32,
                                    out: *mut *mut c_void) -> i32;
    pub fn b3w_commit_key_destroy(key: *mut c_void);
    pub fn b3w_commit_records(ctx: *mut c_void, key: *const c_void, host_records: *const u32, n: u32, host_points: *mut u8,
                              host_public: *mut u32, host_status: *mut i32) -> i32;
    pub fn b3w_chain_commit_only(chain: *mut c_void, key: *const c_void, d_points: *mut u8) -> i32;
    pub fn b3w_chain_commitments(chain: *mut c_void, host_points: *mut u8, stream: *mut c_void) -> i32;
    pub fn b3w_chain_commit_from_records(chain: *mut c_void, key: *const c_void, d_points: *mut u8) -> i32;
    pub fn b3w_chain_commit_overlap(chain: *mut c_void, mode: i32) -> i32;
    pub fn b3w_chain_check_constraints(chain: *mut c_void, r1cs: *const c_void) -> i32;
    pub fn b3w_chain_violations(chain: *mut c_void, host_violations: *mut u32, stream: *mut c_void) -> i32;
}

/// What one pass leaves on the host: 15 public-output words per step (leaf steps in (chunk, block) order, then the parent steps
/// in (chunk, height) order: z_{i+1} of step i = words 0..15 of row i), a status per step, BLAKE3(preimage) as 8 words.
pub struct FoldOutputs { pub public: Vec<u32>, pub status: Vec<i32>, pub root: [u32; 8], pub n_leaf_steps: u64, pub n_parent_steps: u64 }

/// Safe wrapper of the chained pass for ONE rank: create -> run_leaves -> run_parents -> outputs.
pub struct Fold { chain: *mut c_void }

impl Fold {
    /// `preimage_len` bytes, all chunks on this rank; bodies go through `ring` buffers of `batch_steps` step witnesses.
    pub fn new(calc: &Calculator, preimage_len: u64, batch_steps: u32, ring: u32) -> Result<Self, i32> {
        let mut chain = std::ptr::null_mut();
        let n = unsafe { b3w_chain_num_chunks(preimage_len) };
        let rc = unsafe { b3w_chain_create(calc.ctx, preimage_len, 0, n as u32, batch_steps, ring, 1, &mut chain) };
        if rc != 0 { Err(rc) } else { Ok(Fold { chain }) }
    }

    /// One commitment per step instead of witness bodies (`key` from b3w_commit_key_create_ex on the same Calculator).
    pub fn commit_only(&mut self, key: *const c_void) -> Result<(), i32> {
        let rc = unsafe { b3w_chain_commit_only(self.chain, key, std::ptr::null_mut()) };
        if rc != 0 { Err(rc) } else { Ok(()) }
    }

    /// The fold-shaped pass: bodies are written and handed on as usual, and every step's commitment is computed from its record
    /// beside them (`overlap`: -1 auto, 0 serial, 1 free, 2 gated — b3wit.h, b3w_chain_commit_overlap).  `r1cs` (from
    /// b3w_r1cs_create on the same Calculator, or null) adds the constraint check of every step witness in front of the consumer.
    pub fn fold_shaped(&mut self, key: *const c_void, r1cs: *const c_void, overlap: i32) -> Result<(), i32> {
        let mut rc = unsafe { b3w_chain_check_constraints(self.chain, r1cs) };
        if rc == 0 { rc = unsafe { b3w_chain_commit_overlap(self.chain, overlap) }; }
        if rc == 0 { rc = unsafe { b3w_chain_commit_from_records(self.chain, key, std::ptr::null_mut()) }; }
        if rc != 0 { Err(rc) } else { Ok(()) }
    }

    /// After a pass with `fold_shaped(_, r1cs, _)`: violated constraints per step (0 = the step satisfies its circuit), leaf steps then
    /// parent steps: n_leaf + n_parent words, sized HERE from b3w_chain_info (a caller's smaller count would be a heap overflow).
    pub fn violations(&mut self) -> Result<Vec<u32>, i32> {
        let mut v = vec![0u32; self.steps()?];
        let rc = unsafe { b3w_chain_violations(self.chain, v.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 { Err(rc) } else { Ok(v) }
    }

    fn steps(&self) -> Result<usize, i32> {                      // n_leaf + n_parent: what the per-step outputs of the chain are sized by
        let (mut nl, mut np, mut nc, mut pl, mut place) = (0u64, 0u64, 0u64, 0u32, 0i32);
        let rc = unsafe { b3w_chain_info(self.chain, &mut nl, &mut np, &mut nc, &mut pl, &mut place) };
        if rc != 0 { Err(rc) } else { Ok((nl + np) as usize) }
    }

    /// The whole pass over `preimage`.  `consumer` (with its `user` pointer) sees every batch of step witnesses on the device.
    pub fn run(&mut self, preimage: &[u8], consumer: Option<BatchConsumer>, user: *mut c_void) -> Result<FoldOutputs, i32> {
        let stream = std::ptr::null_mut();                       // the null stream; b3w_chain_outputs waits for it
        let mut rc = unsafe { b3w_chain_run_leaves(self.chain, preimage.as_ptr(), consumer, user, stream) };
        if rc == 0 { rc = unsafe { b3w_chain_run_parents(self.chain, std::ptr::null(), consumer, user, stream) }; }
        if rc != 0 { return Err(rc); }
        let (mut nl, mut np, mut nc, mut pl, mut place) = (0u64, 0u64, 0u64, 0u32, 0i32);
        unsafe { b3w_chain_info(self.chain, &mut nl, &mut np, &mut nc, &mut pl, &mut place) };
        let rows = (nl + np) as usize;
        let mut out = FoldOutputs { public: vec![0u32; rows * 15], status: vec![0i32; rows], root: [0u32; 8], n_leaf_steps: nl, n_parent_steps: np };
        rc = unsafe { b3w_chain_outputs(self.chain, out.public.as_mut_ptr(), out.status.as_mut_ptr(), out.root.as_mut_ptr(), stream) };
        if rc != 0 { Err(rc) } else { Ok(out) }
    }

    /// After `commit_only` + `run`: the points, 64 bytes (x, y little-endian) per step (n_leaf + n_parent of them, from b3w_chain_info).
    pub fn commitments(&mut self) -> Result<Vec<u8>, i32> {
        let mut pts = vec![0u8; self.steps()? * 64];
        let rc = unsafe { b3w_chain_commitments(self.chain, pts.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 { Err(rc) } else { Ok(pts) }
    }
}

impl Drop for Fold {
    fn drop(&mut self) { unsafe { b3w_chain_destroy(self.chain) } }
}
