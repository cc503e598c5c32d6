//! Source-only FFI stub (no Rust toolchain in the build image; not compiled or tested here).
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
