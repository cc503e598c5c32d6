"""hot-proofs-blake3-circom_amd — MI355X-native batched witness generator for the BLAKE3 circom
circuits of banyancomputer/hot-proofs-blake3-circom.

This module is the Python binding over the C-ABI in include/b3wit.h (libb3wit.so, HIP kernels
for gfx950).  It mirrors the reference's witness-calculator surface
(blake3_nova_js/witness_calculator.js: builder -> WitnessCalculator.calculateWitness /
calculateBinWitness / calculateWTNSBin, same argument meaning and error text) so the parity
tests read like the reference's, and adds the batch API the device path is built for.  The
Node.js drop-in (N-API addon + witness_calculator.js shim) lives in js/.

There is no CPU fallback: if libb3wit.so is missing or no HIP device is present every compute
entry point raises.
"""
import ctypes
import os
import re

import numpy as np

from . import workloads  # noqa: F401


def __getattr__(name):
    if name in ("sharding", "chain"):  # import torch.distributed: load on demand
        import importlib
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("B3WIT_LIB") or os.path.join(PKG_DIR, "libb3wit.so")   # same override as the N-API addon

CIRCUITS = ("compression", "nova_bn254", "nova_vesta", "nova_bn254_o1")
CIRCUIT_ID = {c: i for i, c in enumerate(CIRCUITS)}

B3W_OK = 0
PLACEMENT_NAMES = {0: "plain", 1: "mixed", 2: "interleaved"}          # B3W_PLACEMENT_* of include/b3wit.h
B3W_E_ASSERT_FAILED = 4
B3W_E_NO_DEVICE = 101
B3W_E_DOMAIN = 103

# the strings witness_calculator.js:21-37 attaches to the circom exception codes
_CIRCOM_ERR = {1: "Signal not found.\n", 2: "Too many signals set.\n", 3: "Signal already set.\n",
               4: "Assert Failed.\n", 5: "Not enough memory.\n", 6: "Input signal array access exceeds the size.\n"}


class B3WError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


_lib = None


def lib():
    """The C-ABI library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise B3WError(B3W_E_NO_DEVICE, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; a process must hold ONE
    # HIP runtime, so when torch is installed let it load first and libb3wit.so binds to that copy by soname.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, i32, u64, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_size_t
    sig = {
        "b3w_abi_version": (u32, []),
        "b3w_identify_wasm": (i32, [vp, sz]),
        "b3w_create": (i32, [i32, i32, ctypes.POINTER(vp)]),
        "b3w_destroy": (None, [vp]),
        "b3w_info": (i32, [vp, vp, vp, vp, vp, vp]),
        "b3w_input_signal_size": (i32, [vp, u64]),
        "b3w_calc_witness": (i32, [vp, vp, vp, vp, u32, vp]),
        "b3w_write_wtns_header": (i32, [vp, vp]),
        "b3w_last_error": (i32, [vp, ctypes.c_char_p, sz]),
        "b3w_batch_run_device": (i32, [vp, vp, u32, vp, u64, vp, vp, vp]),
        "b3w_public_words": (u32, [vp]),
        "b3w_batch_alloc": (i32, [vp, u32, u64, ctypes.POINTER(vp)]),
        "b3w_batch_free": (None, [vp]),
        "b3w_batch_run": (i32, [vp, vp, u32, vp]),
        "b3w_batch_outputs": (i32, [vp, vp, vp]),
        "b3w_batch_fetch": (i32, [vp, u32, vp]),
        "b3w_batch_device_ptr": (vp, [vp, ctypes.POINTER(u64)]),
        "b3w_batch_time_device": (i32, [vp, vp, u32, vp, u64, vp, vp, vp, u32, ctypes.POINTER(ctypes.c_float)]),
        "b3w_batch_verify_device": (i32, [vp, vp, u32, u64, vp, vp]),
        "b3w_batch_verify": (i32, [vp, vp]),
        "b3w_r1cs_create": (i32, [vp, vp, sz, ctypes.POINTER(vp)]),
        "b3w_r1cs_info": (i32, [vp, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u64), ctypes.POINTER(u32),
                                ctypes.POINTER(u32), ctypes.POINTER(u32)]),
        "b3w_r1cs_is_tiled": (i32, [vp]),
        "b3w_r1cs_destroy": (None, [vp]),
        "b3w_r1cs_check_device": (i32, [vp, vp, vp, u32, u64, vp, vp, vp]),
        "b3w_batch_r1cs_check": (i32, [vp, vp, vp, vp]),
        "b3w_r1cs_consumer": (None, [vp, vp, u64, u64, u32, vp]),
        "b3w_batch_write_wtns": (i32, [vp, u32, u32, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(u32)]),
        "b3w_batch_write_wtns_ex": (i32, [vp, u32, u32, ctypes.c_char_p, ctypes.c_char_p, u32, ctypes.POINTER(u32)]),
        "b3w_batch_autotune_device": (i32, [vp, vp, u32, vp, u64, vp, vp, vp, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_float)]),
        "b3w_bodies_alloc": (i32, [vp, u64, ctypes.POINTER(vp), ctypes.POINTER(i32)]),
        "b3w_bodies_free": (i32, [vp, vp]),
        "b3w_bodies_trim": (None, []),
        "b3w_ctx_trim": (i32, [vp]),
        "b3w_bodies_configure": (None, [ctypes.c_int64, ctypes.c_int64]),
        "b3w_bodies_stats": (i32, [vp, vp]),
        "b3w_bodies_search_stats": (i32, [vp, ctypes.POINTER(ctypes.c_double)]),
        "b3w_bodies_search_limit": (None, [ctypes.c_double]),
        "b3w_bodies_search_breakdown": (i32, [vp, ctypes.POINTER(ctypes.c_double)]),
        "b3w_bodies_store_rate": (i32, [vp, vp, u32, u64, i32, u32, vp, ctypes.POINTER(ctypes.c_double)]),
        "b3w_batch_placement": (i32, [vp]),
        "b3w_chain_num_chunks": (u64, [u64]),
        "b3w_chain_num_leaf_steps": (u64, [u64]),
        "b3w_chain_path_len": (u32, [u64, u64]),
        "b3w_chain_num_parent_steps": (u64, [u64, u64, u64]),
        "b3w_chain_parent_row": (u64, [u64, u64]),
        "b3w_chain_path_provable": (i32, [u64, u64]),
        "b3w_chain_plan_leaves_device": (i32, [vp, vp, u64, u64, u32, vp, vp, vp]),
        "b3w_chain_tree_device": (i32, [vp, vp, u64, vp, vp]),
        "b3w_chain_plan_parents_device": (i32, [vp, vp, u64, u64, u64, u32, vp, vp]),
        "b3w_commit_key_create": (i32, [vp, i32, u32, vp, ctypes.POINTER(vp)]),
        "b3w_commit_key_create_ex": (i32, [vp, i32, u32, vp, u32, ctypes.POINTER(vp)]),
        "b3w_commit_key_create_folded": (i32, [vp, i32, u32, vp, vp, u32, ctypes.POINTER(vp)]),
        "b3w_slot_widths": (i32, [vp, vp]),
        "b3w_commit_key_window": (u32, [vp]),
        "b3w_commit_key_destroy": (None, [vp]),
        "b3w_commit_key_count": (i32, [vp, i32]),
        "b3w_commit_key_counts": (i32, [vp, ctypes.POINTER(u64)]),
        "b3w_commit_records_device": (i32, [vp, vp, vp, u32, vp, vp, vp, vp]),
        "b3w_commit_records": (i32, [vp, vp, vp, u32, vp, vp, vp]),
        "b3w_chain_commit_only": (i32, [vp, vp, vp]),
        "b3w_chain_commit_from_records": (i32, [vp, vp, vp]),
        "b3w_chain_commit_overlap": (i32, [vp, i32]),
        "b3w_chain_violations_device": (vp, [vp]),
        "b3w_chain_commitments": (i32, [vp, vp, vp]),
        "b3w_chain_check_constraints": (i32, [vp, vp]),
        "b3w_chain_violations": (i32, [vp, vp, vp]),
        "b3w_batch_commit_device": (i32, [vp, vp, vp, u32, u64, vp, vp, vp]),
        "b3w_batch_commit": (i32, [vp, vp, vp, vp]),
        "b3w_commit_consumer": (None, [vp, vp, u64, u64, u32, vp]),
        "b3w_comm_unique_id": (i32, [vp]),
        "b3w_comm_create": (i32, [vp, vp, i32, i32, ctypes.POINTER(vp)]),
        "b3w_comm_create_host": (i32, [vp, ctypes.c_char_p, i32, i32, ctypes.POINTER(vp)]),
        "b3w_comm_create_external": (i32, [vp, i32, i32, vp, vp, ctypes.POINTER(vp)]),
        "b3w_comm_rank": (i32, [vp]),
        "b3w_comm_size": (i32, [vp]),
        "b3w_comm_destroy": (None, [vp]),
        "b3w_comm_allgather": (i32, [vp, vp, vp, u64, vp]),
        "b3w_batch_allgather_public": (i32, [vp, vp, vp]),
        "b3w_chain_create": (i32, [vp, u64, u64, u32, u32, u32, i32, ctypes.POINTER(vp)]),
        "b3w_chain_destroy": (None, [vp]),
        "b3w_chain_run_leaves": (i32, [vp, vp, vp, vp, vp]),
        "b3w_chain_run_parents": (i32, [vp, vp, vp, vp, vp]),
        "b3w_chain_shard": (None, [u64, i32, i32, ctypes.POINTER(u64), ctypes.POINTER(u32)]),
        "b3w_chain_run_parents_sharded": (i32, [vp, vp, vp, vp, vp]),
        "b3w_chain_allgather_hout": (i32, [vp, vp, vp, vp, vp]),
        "b3w_chain_allgather_hout_host": (i32, [vp, vp, vp, vp, vp]),
        "b3w_chain_exchange_ms": (i32, [vp, ctypes.POINTER(ctypes.c_float)]),
        "b3w_chain_info": (i32, [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u32), ctypes.POINTER(i32)]),
        "b3w_chain_outputs": (i32, [vp, vp, vp, vp, vp]),
        "b3w_chain_records": (vp, [vp]),
        "b3w_chain_public": (vp, [vp]),
        "b3w_chain_status": (vp, [vp]),
        "b3w_chain_local_cvs": (vp, [vp]),
        "b3w_chain_root": (vp, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


EXPORTED_SYMBOLS = ("b3w_abi_version", "b3w_identify_wasm", "b3w_create", "b3w_destroy", "b3w_info",
                    "b3w_input_signal_size", "b3w_calc_witness", "b3w_write_wtns_header", "b3w_last_error",
                    "b3w_batch_run_device", "b3w_public_words", "b3w_batch_alloc", "b3w_batch_free",
                    "b3w_batch_run", "b3w_batch_outputs", "b3w_batch_fetch", "b3w_batch_device_ptr",
                    "b3w_batch_time_device", "b3w_batch_verify_device", "b3w_batch_verify",
                    "b3w_r1cs_create", "b3w_r1cs_info", "b3w_r1cs_is_tiled", "b3w_r1cs_destroy", "b3w_r1cs_check_device", "b3w_batch_r1cs_check", "b3w_r1cs_consumer", "b3w_batch_write_wtns", "b3w_batch_write_wtns_ex", "b3w_batch_autotune_device", "b3w_bodies_alloc", "b3w_bodies_free", "b3w_bodies_trim", "b3w_ctx_trim", "b3w_bodies_configure", "b3w_bodies_stats", "b3w_bodies_search_stats", "b3w_bodies_search_limit", "b3w_bodies_search_breakdown", "b3w_bodies_store_rate", "b3w_batch_placement", "b3w_chain_num_chunks", "b3w_chain_num_leaf_steps", "b3w_chain_path_len",
                    "b3w_chain_num_parent_steps", "b3w_chain_parent_row", "b3w_chain_path_provable",
                    "b3w_chain_plan_leaves_device", "b3w_chain_tree_device", "b3w_chain_plan_parents_device",
                    "b3w_commit_key_create", "b3w_commit_key_create_ex", "b3w_commit_key_create_folded", "b3w_slot_widths", "b3w_commit_key_window", "b3w_commit_key_destroy", "b3w_commit_key_count", "b3w_commit_key_counts", "b3w_commit_records_device", "b3w_commit_records", "b3w_chain_commit_only", "b3w_chain_commit_from_records", "b3w_chain_commit_overlap", "b3w_chain_violations_device", "b3w_chain_commitments", "b3w_chain_check_constraints", "b3w_chain_violations", "b3w_batch_commit_device", "b3w_batch_commit", "b3w_commit_consumer",
                    "b3w_comm_unique_id", "b3w_comm_create", "b3w_comm_create_host", "b3w_comm_create_external", "b3w_comm_rank", "b3w_comm_size", "b3w_comm_destroy", "b3w_comm_allgather", "b3w_batch_allgather_public",
                    "b3w_chain_create", "b3w_chain_destroy", "b3w_chain_run_leaves", "b3w_chain_run_parents", "b3w_chain_shard", "b3w_chain_run_parents_sharded", "b3w_chain_allgather_hout", "b3w_chain_allgather_hout_host", "b3w_chain_exchange_ms", "b3w_chain_info",
                    "b3w_chain_outputs", "b3w_chain_records", "b3w_chain_public", "b3w_chain_status", "b3w_chain_local_cvs", "b3w_chain_root")


class graph_capture:
    """`with graph_capture(g, stream=side): <library launches>` — torch.cuda.graph(g, stream=side) for launches of this library.
    What every entry point that takes a stream enqueues is capturable (tests/test_gpu_graph_capture.py: batches, constraint
    checks, a whole chained pass); what is NOT is a finaliser: a Context, Chain, R1cs or CommitKey that Python's cyclic collector
    frees while the capture is open releases device memory and streams — calls the capture's global mode forbids, and the
    process aborts.  So: garbage is collected before the capture opens and the collector stays off until it has closed (objects
    you drop by hand inside the block are your own affair: keep them alive until the block ends)."""

    def __init__(self, graph, stream=None, **kw):
        import torch
        self._inner = torch.cuda.graph(graph, stream=stream, **kw)
        self._gc_was_on = False

    def __enter__(self):
        import gc
        gc.collect()
        self._gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._inner.__enter__()
        except BaseException:
            if self._gc_was_on:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self._inner.__exit__(*exc)
        finally:
            if self._gc_was_on:
                gc.enable()


def fnv_hash(name):
    """witness_calculator.js:325-337 fnvHash (FNV-1a 64) as an int."""
    h = 0xCBF29CE484222325
    for ch in name:
        h ^= ord(ch)
        h = (h * 0x100000001B3) % (1 << 64)
    return h


def flat_array(a):
    """witness_calculator.js:303-317 flatArray."""
    out = []

    def fill(x):
        if isinstance(x, (list, tuple)):
            for y in x:
                fill(y)
        else:
            out.append(x)
    fill(a)
    return out


def _to_int(v):
    if isinstance(v, str):
        return int(v, 0)          # BigInt("0x..") / BigInt("123") / BigInt("-5")
    return int(v)


class Context:
    """Owns a b3w_ctx (one circuit on one device)."""

    def __init__(self, circuit, device=0):
        self.circuit = circuit if isinstance(circuit, str) else CIRCUITS[circuit]
        self._lib = lib()
        h = ctypes.c_void_p()
        rc = self._lib.b3w_create(CIRCUIT_ID[self.circuit], int(device), ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_create({self.circuit}, device={device}) failed with status {rc}"
                               + (" (no HIP device: this library has no CPU path)" if rc == B3W_E_NO_DEVICE else ""))
        self.handle = h
        n32, nwit, nin = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        prime = (ctypes.c_uint8 * 32)()
        ver = (ctypes.c_uint32 * 3)()
        self._lib.b3w_info(h, ctypes.byref(n32), prime, ctypes.byref(nwit), ctypes.byref(nin), ver)
        self.n32, self.witness_size, self.input_size = n32.value, nwit.value, nin.value
        self.prime = int.from_bytes(bytes(prime), "little")
        self.version = tuple(ver)
        self.public_words = self._lib.b3w_public_words(h)
        self.body_bytes = self.witness_size * 32

    def close(self):
        if getattr(self, "handle", None):
            for h in self.__dict__.pop("_chain_cache", {}).values():     # chain.fold_witnesses' native objects
                self._lib.b3w_chain_destroy(h)
            self._lib.b3w_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        buf = ctypes.create_string_buffer(512)
        self._lib.b3w_last_error(self.handle, buf, 512)
        return buf.value.decode()

    def wtns_header(self):
        out = (ctypes.c_uint8 * 76)()
        self._lib.b3w_write_wtns_header(self.handle, out)
        return bytes(out)

    def input_signal_size(self, name):
        return self._lib.b3w_input_signal_size(self.handle, fnv_hash(name))

    # -- batch on caller-owned device memory (torch tensors or raw pointers) -----------------
    def run_device(self, d_records, n, d_bodies, pitch=0, d_public=0, d_status=0, stream=0):
        rc = self._lib.b3w_batch_run_device(self.handle, d_records, n, d_bodies, pitch, d_public or None,
                                            d_status or None, stream or None)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_run_device: status {rc}: {self.last_error()}")

    def verify_device(self, d_bodies, n, pitch, d_mismatch, stream=0):
        """On-device check of n bodies: d_mismatch[i] = differing 16-byte units (0 = valid witness)."""
        rc = self._lib.b3w_batch_verify_device(self.handle, d_bodies, n, pitch, d_mismatch, stream or None)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_verify_device: status {rc}: {self.last_error()}")

    def autotune_device(self, d_records, n, d_bodies, pitch=0, d_public=0, d_status=0, stream=0):
        """Pick the fastest bit-identical kernel variant for these buffers; returns (variant, ms)."""
        v, ms = ctypes.c_int32(), ctypes.c_float()
        rc = self._lib.b3w_batch_autotune_device(self.handle, d_records, n, d_bodies, pitch, d_public or None,
                                                 d_status or None, stream or None, ctypes.byref(v), ctypes.byref(ms))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_autotune_device: status {rc}: {self.last_error()}")
        return v.value, ms.value

    def slot_widths(self):
        """Bits each witness slot can hold (1, 32, 64 or 256), as the commitment kernel cuts it into virtual slots."""
        out = np.zeros(self.witness_size, dtype=np.uint16)
        rc = lib().b3w_slot_widths(self.handle, out.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_slot_widths: status {rc}: {self.last_error()}")
        return out

    def alloc_bodies(self, nbytes):
        """Device buffer for bodies, placed over two classes of HBM when possible (b3w_bodies_alloc)."""
        return BodyBuffer(self, nbytes)

    def trim(self):
        """Release what this context holds beyond its tables: chain.fold_witnesses' cached chain objects and the ring buffers
        kept from destroyed chains (b3w_ctx_trim).  Follow with lib().b3w_bodies_trim() to hand the memory to the driver."""
        for h in self.__dict__.pop("_chain_cache", {}).values():
            self._lib.b3w_chain_destroy(h)
        self._lib.b3w_ctx_trim(self.handle)

    def bodies_stats(self):
        """Placement allocator of this context's device (b3w_bodies_stats): arena / used-up address space, pooled and live bytes."""
        out = (ctypes.c_uint64 * 6)()
        self._lib.b3w_bodies_stats(self.handle, out)
        return dict(arena_bytes=out[0], arena_used=out[1], pooled_bytes=out[2], live_bytes=out[3], live_buffers=out[4], handles_created=out[5])

    def placement_cost(self):
        """What placement has cost on this context's device so far (b3w_bodies_search_stats)."""
        out = (ctypes.c_double * 5)()
        self._lib.b3w_bodies_search_stats(self.handle, out)
        bd = (ctypes.c_double * 4)()
        self._lib.b3w_bodies_search_breakdown(self.handle, bd)
        return dict(search_s=out[0], search_gib_walked=out[1], search_timeouts=int(out[2]), search_limit_s=out[3], check_s=out[4],
                    create_s=bd[0], map_s=bd[1], probe_s=bd[2], release_s=bd[3])

    def store_rate(self, d_bodies, n, pitch=0, shape=0, iters=20, stream=0):
        """GB/s of kernels that do nothing but the witness kernels' stores into n bodies of this buffer (b3w_bodies_store_rate):
        shape 0 / 1 = body streams (one wave per 4 / 8 bodies), 2 = the runtime's fill shape.  Overwrites the bodies."""
        g = ctypes.c_double()
        rc = self._lib.b3w_bodies_store_rate(self.handle, d_bodies, n, pitch, shape, iters, stream or None, ctypes.byref(g))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_bodies_store_rate: status {rc}: {self.last_error()}")
        return g.value

    def time_device(self, d_records, n, d_bodies, pitch, d_public, d_status, stream, iters):
        ms = ctypes.c_float()
        rc = self._lib.b3w_batch_time_device(self.handle, d_records, n, d_bodies, pitch, d_public or None,
                                             d_status or None, stream or None, iters, ctypes.byref(ms))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_time_device: status {rc}: {self.last_error()}")
        return ms.value


class BodyBuffer:
    """A linear device buffer for witness bodies from b3w_bodies_alloc: `ptr` (int), `nbytes`, and
    `placement` ("mixed" = its 256 MiB pieces alternate between two classes of HBM and one real launch was >= 10 % faster than
    into plain buffers; "interleaved" = they alternate, but this box's plain buffers were as fast: no speed claim; "plain")."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p, pl = ctypes.c_void_p(), ctypes.c_int32()
        rc = lib().b3w_bodies_alloc(ctx.handle, self.nbytes, ctypes.byref(p), ctypes.byref(pl))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_bodies_alloc({nbytes}): status {rc}: {ctx.last_error()}")
        self.ptr = p.value
        self.placement = PLACEMENT_NAMES.get(pl.value, "plain")

    def data_ptr(self):
        return self.ptr

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def tensor(self):
        """Zero-copy 1-D uint8 torch view of the buffer (valid until free())."""
        import torch
        return torch.as_tensor(self, device=torch.device("cuda", torch.cuda.current_device()))

    def free(self):
        if getattr(self, "ptr", None):
            lib().b3w_bodies_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)


class Comm:
    """A native communicator of the C-ABI (b3w_comm_*), for hosts without torch.distributed and for
    chain.fold_witnesses(comm=...).  Three transports behind the same handle:
      Comm(ctx, uid, rank, nranks)           RCCL over xGMI, one process per GPU (librccl loaded at run time); rank 0 makes the id
                                             (Comm.unique_id()) and hands its 128 bytes to the other ranks
      Comm.host(ctx, name, rank, nranks)     the processes of one host through POSIX shared memory (b3w_comm_create_host): several
                                             ranks on ONE GPU, or no RCCL; `name` = "/unique-to-the-job", the same on every rank
      Comm.external(ctx, rank, nranks, fn)   the caller's collective: fn(d_send, d_recv, bytes_per_rank, stream) with device
                                             pointers as ints (sharding.torch_allgather builds one over torch.distributed)"""

    @staticmethod
    def unique_id():
        uid = (ctypes.c_uint8 * 128)()
        rc = lib().b3w_comm_unique_id(uid)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_comm_unique_id: status {rc}")
        return bytes(uid)

    def __init__(self, ctx, uid, rank, nranks, _handle=None):
        self.ctx, self.rank, self.nranks = ctx, int(rank), int(nranks)
        self.transport, self._cb, self._cb_error = "rccl", None, None
        if _handle is not None:
            self.handle = _handle
            return
        h = ctypes.c_void_p()
        rc = lib().b3w_comm_create(ctx.handle, (ctypes.c_uint8 * 128).from_buffer_copy(uid), self.rank, self.nranks, ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_comm_create: status {rc}: {ctx.last_error()}")
        self.handle = h

    @classmethod
    def host(cls, ctx, name, rank, nranks):
        h = ctypes.c_void_p()
        rc = lib().b3w_comm_create_host(ctx.handle, name.encode(), int(rank), int(nranks), ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_comm_create_host({name}): status {rc}: {ctx.last_error()}")
        c = cls(ctx, None, rank, nranks, _handle=h)
        c.transport = "host"
        return c

    @classmethod
    def external(cls, ctx, rank, nranks, fn):
        c = cls(ctx, None, rank, nranks, _handle=ctypes.c_void_p())
        c.transport = "external"

        def _cb(user, d_send, d_recv, nbytes, stream):
            try:
                fn(d_send or 0, d_recv or 0, nbytes, stream or 0)
                return 0
            except BaseException as e:          # never unwind through the C frames: report, fail the native call
                c._cb_error = e
                return -1
        c._cb = _ALLGATHER_FN(_cb)                # (kept alive with the object)
        h = ctypes.c_void_p()
        rc = lib().b3w_comm_create_external(ctx.handle, int(rank), int(nranks), ctypes.cast(c._cb, ctypes.c_void_p), None, ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_comm_create_external: status {rc}: {ctx.last_error()}")
        c.handle = h
        return c

    def allgather(self, d_send, d_recv, bytes_per_rank, stream=0):
        rc = lib().b3w_comm_allgather(self.handle, d_send, d_recv, bytes_per_rank, stream or None)
        if rc != B3W_OK:
            err, self._cb_error = self._cb_error, None
            raise B3WError(rc, f"b3w_comm_allgather: status {rc}: {self.ctx.last_error()}" + (f" ({err!r})" if err else ""))

    def close(self):
        if getattr(self, "handle", None):
            lib().b3w_comm_destroy(self.handle)
            self.handle = None


R1CS_DIR = os.path.join(PKG_DIR, "constraints")
# derived from the circuit text by tools/gen_r1cs.py: the two builds whose simplification (aliases and constants only) is
# restated exactly, and the two O2 builds, whose wires are found by aligning the reference's O2 witnesses with its O1
# witnesses on the same inputs and whose system is the O1 system with the missing wires eliminated
BUILTIN_R1CS = {"compression": "blake3_compression.r1cs.gz", "nova_bn254_o1": "blake3_nova_bn254_o1.r1cs.gz",
                "nova_bn254": "blake3_nova_bn254.r1cs.gz", "nova_vesta": "blake3_nova_vesta.r1cs.gz"}


def read_r1cs_image(image):
    """bytes of an .r1cs image from bytes, a path (.r1cs or .r1cs.gz) or the name of a built-in file under constraints/."""
    if isinstance(image, (str, os.PathLike)):
        path = image if os.path.exists(image) else os.path.join(R1CS_DIR, image)
        raw = open(path, "rb").read()
        if raw[:2] == b"\x1f\x8b":
            import gzip
            raw = gzip.decompress(raw)
        return raw
    return bytes(image)


class R1cs:
    """A rank-1 constraint system on the device (b3w_r1cs_create) for on-device satisfaction checks of witness bodies
    — the counterpart of circom_tester's expectPass (test/blake3_hash.test.ts:36) / synthesize_with_vec's constraints
    (rust_fold/src/utils.rs:17-88).  `image`: bytes of an iden3 .r1cs file, or a path to one (.r1cs or .r1cs.gz);
    None = the system this package derives for the circuit (all four committed builds; the reference ships no .r1cs)."""

    def __init__(self, ctx, image=None):
        self.ctx = ctx
        if image is None:
            if ctx.circuit not in BUILTIN_R1CS:
                raise B3WError(100, f"no derived constraint system for {ctx.circuit}: pass the circuit's .r1cs")
            image = os.path.join(R1CS_DIR, BUILTIN_R1CS[ctx.circuit])
        if isinstance(image, (str, os.PathLike)):
            raw = open(image, "rb").read()
            if raw[:2] == b"\x1f\x8b":
                import gzip
                raw = gzip.decompress(raw)
            image = raw
        buf = bytes(image)
        h = ctypes.c_void_p()
        rc = lib().b3w_r1cs_create(ctx.handle, buf, len(buf), ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_r1cs_create: status {rc}: {ctx.last_error()}")
        self.handle = h
        m, nw, nt, po, pi, pr = (ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32(),
                                 ctypes.c_uint32())
        lib().b3w_r1cs_info(h, ctypes.byref(m), ctypes.byref(nw), ctypes.byref(nt), ctypes.byref(po), ctypes.byref(pi), ctypes.byref(pr))
        self.n_constraints, self.n_wires, self.n_terms = m.value, nw.value, nt.value
        self.n_pub_out, self.n_pub_in, self.n_prv_in = po.value, pi.value, pr.value
        self.tiled = bool(lib().b3w_r1cs_is_tiled(h))      # False: rows not local enough, the gather kernel serves this system

    def check_device(self, d_bodies, n, pitch, d_violations, d_first=0, stream=0):
        """n bodies in HBM -> d_violations[i] = violated constraints of body i (0 = valid), d_first[i] = lowest one."""
        rc = lib().b3w_r1cs_check_device(self.ctx.handle, self.handle, d_bodies, n, pitch, d_violations, d_first or None, stream or None)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_r1cs_check_device: status {rc}: {self.ctx.last_error()}")

    def close(self):
        if getattr(self, "handle", None):
            lib().b3w_r1cs_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CommitKey:
    """Commitment key on the device (b3w_commit_key_create): `generators` = bytes, one affine point (x, y: 32-byte
    little-endian each, standard form) per committed slot, i.e. witness_size - first_slot of them; curve "bn254_g1"
    or "pallas" (the group whose scalar field is the --prime vesta circuit's field; "vesta" is accepted as its older name); window = 12 | 16 | 18 bits per table window (0: automatic, see include/b3wit.h)."""
    CURVES = {"bn254_g1": 0, "pallas": 1, "vesta": 1}     # "vesta" = older name of the Pallas curve id (after the circuit's prime)

    def __init__(self, ctx, curve, generators, first_slot=0, window=0, fold=None):
        """fold: an .r1cs image (bytes / path), or True for the circuit's built-in one — the slots that the circuit's linear
        constraints express through others (every 32-bit word through its bits) are folded into those slots' generators
        (fold.py) and not committed by themselves: the same point for every witness, half the point additions."""
        self.ctx = ctx
        buf = bytes(generators)
        nslots = ctx.witness_size - first_slot
        if len(buf) != 64 * nslots:
            raise B3WError(100, "generators: 64 bytes per committed slot")
        self.folded_slots, mask = 0, None
        if fold is not None and fold is not False:
            from . import fold as _fold
            image = read_r1cs_image(BUILTIN_R1CS[ctx.circuit]) if fold is True else read_r1cs_image(fold)
            buf, mask, self.fold_stats = _fold.fold_generators(image, ctx.slot_widths(), first_slot, buf, curve)
            self.folded_slots = sum(1 for v in mask if v)          # folded away, or cut down to one bit
        h = ctypes.c_void_p()
        rc = lib().b3w_commit_key_create_folded(ctx.handle, self.CURVES[curve], first_slot, buf, bytes(mask) if mask is not None else None, window,
                                                ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_commit_key_create_folded: status {rc}: {ctx.last_error()}")
        self.handle = h
        self.window = lib().b3w_commit_key_window(h)

    def commit_device(self, d_bodies, n, pitch, d_points, d_status=0, stream=0):
        """n bodies in HBM -> n affine points (64 bytes each) in HBM."""
        rc = lib().b3w_batch_commit_device(self.ctx.handle, self.handle, d_bodies, n, pitch, d_points, d_status or None, stream or None)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_commit_device: status {rc}: {self.ctx.last_error()}")

    def commit_records_device(self, d_records, n, d_points, d_status, d_public=0, stream=0):
        """n input records in HBM -> n affine points, without the witness bodies (b3w_commit_records_device)."""
        rc = lib().b3w_commit_records_device(self.ctx.handle, self.handle, d_records, n, d_points, d_public or None, d_status, stream or None)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_commit_records_device: status {rc}: {self.ctx.last_error()}")

    def commit_records(self, records):
        """records: uint32 [n, 28 | 32] numpy array -> (points uint8 [n, 64], public uint32 [n, 16 | 15], status int32 [n])."""
        recs = np.ascontiguousarray(records, dtype=np.uint32)
        n = recs.shape[0]
        pts = np.zeros((n, 64), dtype=np.uint8)
        pub = np.zeros((n, self.ctx.public_words), dtype=np.uint32)
        st = np.zeros(n, dtype=np.int32)
        rc = lib().b3w_commit_records(self.ctx.handle, self.handle, recs.ctypes.data, n, pts.ctypes.data, pub.ctypes.data, st.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_commit_records: status {rc}: {self.ctx.last_error()}")
        return pts, pub, st

    def count(self, on=True):
        """start (and reset) / stop counting the point additions of this key's commit launches (b3w_commit_key_count)"""
        lib().b3w_commit_key_count(self.handle, 1 if on else 0)

    def counts(self):
        """(mixed point additions, witnesses) since count() — 10 field multiplications per addition"""
        out = (ctypes.c_uint64 * 2)()
        lib().b3w_commit_key_counts(self.handle, out)
        return int(out[0]), int(out[1])

    def close(self):
        if getattr(self, "handle", None):
            lib().b3w_commit_key_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChainPlanner:
    """Chained-mode step-input planner over torch device tensors (b3w_chain_*): the device counterpart
    of rust_fold's format_input / update_for_step / hash_with_path for ALL chunks of a preimage."""

    def __init__(self, ctx):
        self.ctx = ctx

    def _chk(self, rc, what):
        if rc != B3W_OK:
            raise B3WError(rc, f"{what}: status {rc}: {self.ctx.last_error()}")

    def parent_rows(self, n_chunks, first_chunk=0, n_chunks_local=None):
        """Where each local chunk's parent steps sit among the parent records: (row of its first parent step, path length,
        provable) per chunk — b3w_chain_parent_row / path_len / path_provable.  Complete trees: (c * P, P, True)."""
        L = lib()
        nl = n_chunks - first_chunk if n_chunks_local is None else n_chunks_local
        r0 = L.b3w_chain_parent_row(first_chunk, n_chunks)
        return [(L.b3w_chain_parent_row(c, n_chunks) - r0, L.b3w_chain_path_len(c, n_chunks), bool(L.b3w_chain_path_provable(c, n_chunks)))
                for c in range(first_chunk, first_chunk + nl)]

    def plan(self, d_preimage, with_parents=True, first_chunk=0, n_chunks_local=None, stream=0):
        """d_preimage: uint8 CUDA tensor holding the WHOLE preimage.  Returns a dict with
        records (int32 [n_steps, 32]: leaf steps of the local chunks, then their parent steps — see parent_rows),
        n_leaf_steps, chunk_cvs (all chunks), root (8 words), n_chunks, path_len (the longest path)."""
        import torch
        L = lib()
        ln = d_preimage.numel()
        dev = d_preimage.device
        n = L.b3w_chain_num_chunks(ln)
        nl = n - first_chunk if n_chunks_local is None else n_chunks_local
        last_local = first_chunk + nl == n
        last_blocks = (max(ln - (n - 1) * 1024, 1) + 63) // 64
        n_leaf = nl * 16 - ((16 - last_blocks) if last_local else 0)
        P = L.b3w_chain_path_len(0, n)                     # the longest path (chunk 0's); every path in a complete tree
        complete = (n & (n - 1)) == 0
        n_par = L.b3w_chain_num_parent_steps(ln, first_chunk, nl) if with_parents else 0
        recs = torch.zeros((nl * 16 + n_par, 32), dtype=torch.int32, device=dev)
        levels = torch.zeros(((2 * n + 64) * 8,), dtype=torch.int32, device=dev)
        root = torch.zeros(8, dtype=torch.int32, device=dev)
        s = stream or None
        # chunk CVs of ALL chunks are needed for the tree (a multi-GPU job all-gathers them instead)
        allrecs = recs if (first_chunk == 0 and nl == n) else None
        if allrecs is None:
            tmp = torch.zeros((n * 16, 32), dtype=torch.int32, device=dev)
            self._chk(L.b3w_chain_plan_leaves_device(self.ctx.handle, d_preimage.data_ptr(), ln, 0, n, tmp.data_ptr(),
                                                     levels.data_ptr(), s), "plan_leaves")
            recs[:nl * 16] = tmp[first_chunk * 16:(first_chunk + nl) * 16]
        else:
            self._chk(L.b3w_chain_plan_leaves_device(self.ctx.handle, d_preimage.data_ptr(), ln, 0, n, recs.data_ptr(),
                                                     levels.data_ptr(), s), "plan_leaves")
        self._chk(L.b3w_chain_tree_device(self.ctx.handle, levels.data_ptr(), n, root.data_ptr(), s), "tree")
        if n_par:
            self._chk(L.b3w_chain_plan_parents_device(self.ctx.handle, levels.data_ptr(), n, ln, first_chunk, nl,
                                                      recs[nl * 16:].data_ptr(), s), "plan_parents")
        # drop the unused record slots of a partial last chunk
        if last_local and last_blocks < 16:
            keep = torch.ones(recs.shape[0], dtype=torch.bool, device=dev)
            keep[(nl - 1) * 16 + last_blocks: nl * 16] = False
            recs = recs[keep].contiguous()
        return dict(records=recs, n_leaf_steps=n_leaf, n_parent_steps=n_par, chunk_cvs=levels[:n * 8].view(n, 8),
                    root=root, n_chunks=n, path_len=P, complete=complete, last_blocks=last_blocks)


class Batch:
    """Library-owned device buffers for up to `capacity` witnesses (b3w_batch_*)."""

    def __init__(self, ctx, capacity, pitch=0):
        self.ctx = ctx
        h = ctypes.c_void_p()
        rc = lib().b3w_batch_alloc(ctx.handle, capacity, pitch, ctypes.byref(h))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_alloc: status {rc}: {ctx.last_error()}")
        self.handle, self.capacity, self.n = h, capacity, 0

    def run(self, records):
        records = np.ascontiguousarray(records, dtype=np.uint32)
        assert records.ndim == 2 and records.shape[1] == self.ctx.input_size
        rc = lib().b3w_batch_run(self.handle, records.ctypes.data, records.shape[0], None)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_run: status {rc}: {self.ctx.last_error()}")
        self.n = records.shape[0]

    def outputs(self):
        pub = np.zeros((self.n, self.ctx.public_words), dtype=np.uint32)
        st = np.zeros(self.n, dtype=np.int32)
        rc = lib().b3w_batch_outputs(self.handle, pub.ctypes.data, st.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_outputs: status {rc}")
        return pub, st

    def fetch(self, index):
        body = np.zeros(self.ctx.body_bytes, dtype=np.uint8)
        rc = lib().b3w_batch_fetch(self.handle, index, body.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_fetch: status {rc}")
        return body

    def verify(self):
        """On-device check of the last run's bodies; returns per-witness mismatch counts (0 = valid witness)."""
        mm = np.zeros(self.n, dtype=np.uint32)
        rc = lib().b3w_batch_verify(self.handle, mm.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_verify: status {rc}: {self.ctx.last_error()}")
        return mm

    def r1cs_check(self, r1cs):
        """Constraint check of the last run's bodies: (violations uint32 [n], first violated constraint uint32 [n])."""
        viol = np.zeros(self.n, dtype=np.uint32)
        first = np.zeros(self.n, dtype=np.uint32)
        rc = lib().b3w_batch_r1cs_check(self.handle, r1cs.handle, viol.ctypes.data, first.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_r1cs_check: status {rc}: {self.ctx.last_error()}")
        return viol, first

    def write_wtns(self, directory, prefix="witness_", first=0, count=None, threads=0):
        """Stream witnesses [first, first+count) to <directory>/<prefix><index>.wtns with `threads` writer threads (0 = the
        library's choice); returns files written."""
        count = self.n - first if count is None else count
        wr = ctypes.c_uint32()
        rc = lib().b3w_batch_write_wtns_ex(self.handle, first, count, str(directory).encode(), prefix.encode(), threads, ctypes.byref(wr))
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_write_wtns: status {rc}: {self.ctx.last_error()}")
        return wr.value

    def commit(self, key):
        """Pedersen commitments of the last run's witnesses: (points uint8 [n, 64], status int32 [n])."""
        pts = np.zeros((self.n, 64), dtype=np.uint8)
        st = np.zeros(self.n, dtype=np.int32)
        rc = lib().b3w_batch_commit(self.handle, key.handle, pts.ctypes.data, st.ctypes.data)
        if rc != B3W_OK:
            raise B3WError(rc, f"b3w_batch_commit: status {rc}: {self.ctx.last_error()}")
        return pts, st

    @property
    def placement(self):
        return PLACEMENT_NAMES.get(lib().b3w_batch_placement(self.handle), "plain")

    def device_ptr(self):
        pitch = ctypes.c_uint64()
        p = lib().b3w_batch_device_ptr(self.handle, ctypes.byref(pitch))
        return p, pitch.value

    def close(self):
        if self.handle:
            lib().b3w_batch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WitnessCalculator:
    """Mirror of `class WitnessCalculator` (witness_calculator.js:108-274) over the C-ABI.

    Same fields (version, n32, prime, witnessSize), same methods, same error text, same EVALUATION ORDER (below).  The
    reference methods are `async` but resolve synchronously; these are plain calls.

    `sanity_check` is the reference's `options` object (witness_calculator.js:66-75); a dict may carry this mirror's own
    switches, named as in the Node shim: `logDFlags` (False: the nova circuits' "D_FLAGS:  0" line is not printed), `log`
    (a callable that receives the line instead of `print` — what console.log is to the reference), `strictErrorParity`
    (the loader's never-cleared errStr, witness_calculator.js:16,41: an assert's text carries the traces of every earlier
    assert on this calculator)."""

    _CHECKDEPTH_FIRST = re.compile(r"^Error in template (Num2Bits_\d+ line: \d+\nError in template (LessThan|GreaterEqThan)_\d+ "
                                   r"line: \d+\nError in template )?Blake3NovaTreePath_CheckDepth_")

    def __init__(self, ctx, sanity_check=True):
        self.instance = ctx
        self.version = ctx.version[0]
        self.n32 = ctx.n32
        self.prime = ctx.prime
        self.witnessSize = ctx.witness_size
        self.sanityCheck = sanity_check
        opts = sanity_check if isinstance(sanity_check, dict) else {}
        # circuits/blake3_nova.circom:166 logs once per witness through writeBufferMessage (witness_calculator.js:44-58) — also
        # when a LATER assert rejects the input, not when the first component (CheckDepth, blake3_nova.circom:201) already does
        self._log_dflags = ctx.circuit != "compression" and opts.get("logDFlags", True) is not False
        self._log = opts.get("log", print)
        self._strict_err = bool(opts.get("strictErrorParity"))
        self._err_str = ""

    def circom_version(self):
        return self.version

    def _do_calculate_witness(self, inp):
        """_doCalculateWitness (witness_calculator.js:131-169): returns the body (uint8 array).

        In the reference's order: keys as the mapping yields them; per key the size check (:142-151), then its values are set
        one by one (:152-163) and the circuit runs INSIDE the call that sets the last missing input — so its assert, or its log
        line, comes before anything is known about the keys behind the completing one, and a fault of an earlier key before the
        circuit has run (tests/golden/order.json, made by the reference loader)."""
        ctx = self.instance
        hashes, counts, vals = [], [], []
        body = None
        for k in inp.keys():
            farr = flat_array(inp[k])
            size = ctx.input_signal_size(k)
            if len(farr) < size:
                raise B3WError(6, f"Not enough values for input signal {k}\n")
            if len(farr) > size:
                raise B3WError(2, f"Too many values for input signal {k}\n")
            hashes.append(fnv_hash(k))
            counts.append(len(farr))
            vals += [(_to_int(v) % self.prime) for v in farr]          # normalize(), :319-323
            if body is None and farr and len(vals) == ctx.input_size:
                body = self._run(hashes, counts, vals)                 # the last missing input has just been set
        if len(vals) < ctx.input_size:
            raise B3WError(104, f"Not all inputs have been set. Only {len(vals)} out of {ctx.input_size}")
        return body

    def _run(self, hashes, counts, vals):
        ctx = self.instance
        h = np.array(hashes, dtype=np.uint64)
        c = np.array(counts, dtype=np.uint32)
        v = np.frombuffer(b"".join(x.to_bytes(32, "little") for x in vals), dtype=np.uint8)
        body = np.zeros(ctx.body_bytes, dtype=np.uint8)
        rc = lib().b3w_calc_witness(ctx.handle, h.ctypes.data, c.ctypes.data, v.ctypes.data, len(hashes), body.ctypes.data)
        if rc != B3W_OK:
            if rc in _CIRCOM_ERR:
                tail = ctx.last_error()            # "Assert Failed.\n" + the circom trace lines
                head = _CIRCOM_ERR[rc]
                trace = tail[len(head):] if tail.startswith(head) else tail
                if rc == B3W_E_ASSERT_FAILED:
                    if self._log_dflags and not self._CHECKDEPTH_FIRST.match(trace):
                        self._log("D_FLAGS:  0")
                    prior = self._err_str if self._strict_err else ""
                    self._err_str += trace
                    raise B3WError(rc, "Error: " + head + prior + trace)
                raise B3WError(rc, "Error: " + head + trace)
            raise B3WError(rc, ctx.last_error())
        if self._log_dflags:
            self._log("D_FLAGS:  0")
        return body

    def calculateWitness(self, inp, sanity_check=0):
        body = self._do_calculate_witness(inp)
        b = body.tobytes()
        return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(self.witnessSize)]

    def calculateBinWitness(self, inp, sanity_check=0):
        return self._do_calculate_witness(inp)

    def calculateWTNSBin(self, inp, sanity_check=0):
        body = self._do_calculate_witness(inp)
        return np.frombuffer(self.instance.wtns_header() + body.tobytes(), dtype=np.uint8)


def builder(code, options=None, device=0):
    """Mirror of `module.exports = async function builder(code, options)` (witness_calculator.js:1-106).
    `code` = the circuit's .wasm bytes (identified by sha256), or a circuit name."""
    if isinstance(code, str):
        circuit = code
    else:
        buf = bytes(code)
        cid = lib().b3w_identify_wasm(buf, len(buf))
        if cid < 0:
            raise B3WError(100, "unknown circuit binary: not one of the reference's committed WASMs")
        circuit = CIRCUITS[cid]
    return WitnessCalculator(Context(circuit, device), True if options is None else options)
