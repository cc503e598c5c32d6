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

// ---- declarations only (no wrappers: nothing here can be compiled or run in the build image, so the untested surface stays
// at the one call the reference makes).  The chained pass a fold driver would stream (main.rs:41-203 as one pass) and the
// witness commitment (main.rs:166-179 -> prove_step commits to W); parameter names, order and types are held against
// include/b3wit.h by tests/test_rust_ffi_decls_cpu.py.
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
    pub fn b3w_commit_key_create_ex(ctx: *mut c_void, curve: i32, first_slot: u32, host_generators: *const u8, window_bits: u32,
                                    out: *mut *mut c_void) -> i32;
    pub fn b3w_commit_key_destroy(key: *mut c_void);
    pub fn b3w_commit_records(ctx: *mut c_void, key: *const c_void, host_records: *const u32, n: u32, host_points: *mut u8,
                              host_public: *mut u32, host_status: *mut i32) -> i32;
}
