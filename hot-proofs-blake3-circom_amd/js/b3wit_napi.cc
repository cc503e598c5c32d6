// b3wit_napi.cc — Node.js N-API addon over the C-ABI of libb3wit.so (include/b3wit.h).
// Thin marshalling only: every function below forwards to one b3w_* entry point.  The JS shim
// js/witness_calculator.js builds the reference's `builder -> WitnessCalculator` surface
// (blake3_nova_js/witness_calculator.js) on top of it.
//
// Build (no node-gyp needed; headers ship in /usr/include/node):
//   g++ -O2 -std=c++17 -fPIC -shared -I/usr/include/node -Iinclude -o b3wit_napi.node b3wit_napi.cc -ldl
// libb3wit.so is dlopen'ed from the package directory (one level above this file) on first use.
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "b3wit.h"

namespace {

struct Api {
  void *so = nullptr;
  decltype(&b3w_abi_version) abi_version;
  decltype(&b3w_identify_wasm) identify_wasm;
  decltype(&b3w_create) create;
  decltype(&b3w_destroy) destroy;
  decltype(&b3w_info) info;
  decltype(&b3w_input_signal_size) input_signal_size;
  decltype(&b3w_calc_witness) calc_witness;
  decltype(&b3w_write_wtns_header) write_wtns_header;
  decltype(&b3w_last_error) last_error;
  decltype(&b3w_public_words) public_words;
  decltype(&b3w_batch_alloc) batch_alloc;
  decltype(&b3w_batch_free) batch_free;
  decltype(&b3w_batch_run) batch_run;
  decltype(&b3w_batch_outputs) batch_outputs;
  decltype(&b3w_batch_fetch) batch_fetch;
  decltype(&b3w_batch_write_wtns) batch_write_wtns;
  decltype(&b3w_batch_verify) batch_verify;
  decltype(&b3w_batch_placement) batch_placement;
  decltype(&b3w_bodies_trim) bodies_trim;
  decltype(&b3w_chain_create) chain_create;
  decltype(&b3w_chain_destroy) chain_destroy;
  decltype(&b3w_chain_run_leaves) chain_run_leaves;
  decltype(&b3w_chain_run_parents) chain_run_parents;
  decltype(&b3w_chain_info) chain_info;
  decltype(&b3w_chain_outputs) chain_outputs;
  decltype(&b3w_commit_key_create_ex) commit_key_create_ex;
  decltype(&b3w_commit_key_create_folded) commit_key_create_folded;
  decltype(&b3w_commit_key_destroy) commit_key_destroy;
  decltype(&b3w_commit_records) commit_records;
  decltype(&b3w_chain_commit_only) chain_commit_only;
  decltype(&b3w_chain_commitments) chain_commitments;
  decltype(&b3w_batch_commit) batch_commit;
  decltype(&b3w_chain_shard) chain_shard;
  decltype(&b3w_chain_run_parents_sharded) chain_run_parents_sharded;
  decltype(&b3w_chain_allgather_hout_host) chain_allgather_hout_host;
  decltype(&b3w_chain_num_leaf_steps) chain_num_leaf_steps;
  decltype(&b3w_chain_parent_row) chain_parent_row;
  decltype(&b3w_comm_unique_id) comm_unique_id;
  decltype(&b3w_comm_create) comm_create;
  decltype(&b3w_comm_create_host) comm_create_host;
  decltype(&b3w_comm_destroy) comm_destroy;
  decltype(&b3w_batch_allgather_public) batch_allgather_public;
  decltype(&b3w_r1cs_create) r1cs_create;
  decltype(&b3w_r1cs_info) r1cs_info;
  decltype(&b3w_r1cs_destroy) r1cs_destroy;
  decltype(&b3w_batch_r1cs_check) batch_r1cs_check;
  decltype(&b3w_chain_check_constraints) chain_check_constraints;
  decltype(&b3w_chain_violations) chain_violations;
  std::string err;
} api;

bool load_api() {
  if (api.so) return true;
  Dl_info di;
  std::string dir = ".";
  if (dladdr((void *)&load_api, &di) && di.dli_fname) {
    dir = di.dli_fname;
    size_t p = dir.rfind('/');
    dir = p == std::string::npos ? "." : dir.substr(0, p);
  }
  const char *env = getenv("B3WIT_LIB");
  const std::string path = env ? env : dir + "/../libb3wit.so";
  void *so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!so) { api.err = std::string("cannot load ") + path + ": " + dlerror(); return false; }
#define SYM(name)                                                                     \
  api.name = (decltype(api.name))dlsym(so, "b3w_" #name);                             \
  if (!api.name) { api.err = "libb3wit.so lacks b3w_" #name; dlclose(so); return false; }
  SYM(abi_version) SYM(identify_wasm) SYM(create) SYM(destroy) SYM(info) SYM(input_signal_size) SYM(calc_witness)
  SYM(write_wtns_header) SYM(last_error) SYM(public_words) SYM(batch_alloc) SYM(batch_free) SYM(batch_run)
  SYM(batch_outputs) SYM(batch_fetch) SYM(batch_write_wtns) SYM(batch_verify) SYM(batch_placement) SYM(bodies_trim)
  SYM(chain_create) SYM(chain_destroy) SYM(chain_run_leaves) SYM(chain_run_parents) SYM(chain_info) SYM(chain_outputs)
  SYM(commit_key_create_ex) SYM(commit_key_create_folded) SYM(commit_key_destroy) SYM(commit_records) SYM(chain_commit_only) SYM(chain_commitments) SYM(batch_commit) SYM(chain_shard) SYM(chain_run_parents_sharded) SYM(chain_allgather_hout_host) SYM(chain_num_leaf_steps) SYM(chain_parent_row) SYM(comm_unique_id) SYM(comm_create) SYM(comm_create_host) SYM(comm_destroy) SYM(batch_allgather_public)
  SYM(r1cs_create) SYM(r1cs_info) SYM(r1cs_destroy) SYM(batch_r1cs_check) SYM(chain_check_constraints) SYM(chain_violations)
#undef SYM
  api.so = so;
  return true;
}

struct Handle {
  b3w_ctx *ctx = nullptr;
  b3w_batch *batch = nullptr;
  uint32_t batch_cap = 0, batch_n = 0;
  uint32_t generation = 0;          // bumped by every batchRun: the handle owns ONE batch, a later run replaces it
  b3w_comm *comm = nullptr;
  int32_t rank = 0, nranks = 1;
  b3w_commit_key *key = nullptr;
  b3w_r1cs *r1cs = nullptr;
};

#define NAPI_OK(call)                                                   \
  do {                                                                  \
    if ((call) != napi_ok) {                                            \
      napi_throw_error(env, nullptr, "b3wit_napi: N-API call failed: " #call); \
      return nullptr;                                                   \
    }                                                                   \
  } while (0)

// Methods of a batchRun result pass the generation they were created with: acting on the batch of a LATER run with
// the sizes of an earlier one would read or write past buffers, so stale results are refused.
bool fresh_result(napi_env env, Handle *h, size_t argc, napi_value *argv, size_t idx) {
  if (argc <= idx) return true;
  napi_valuetype t;
  if (napi_typeof(env, argv[idx], &t) != napi_ok || t != napi_number) return true;
  uint32_t g = 0;
  if (napi_get_value_uint32(env, argv[idx], &g) == napi_ok && g == h->generation) return true;
  napi_throw_error(env, nullptr, "stale batch result: a later calculateWitnessBatch on this calculator replaced it");
  return false;
}

napi_value throw_status(napi_env env, Handle *h, int32_t rc, const char *what) {
  char msg[640], tail[512] = "";
  if (h && h->ctx) api.last_error(h->ctx, tail, sizeof tail);
  snprintf(msg, sizeof msg, "%s", tail[0] ? tail : what);
  napi_value err, code, m;
  napi_create_string_utf8(env, msg, NAPI_AUTO_LENGTH, &m);
  napi_create_error(env, nullptr, m, &err);
  napi_create_int32(env, rc, &code);
  napi_set_named_property(env, err, "status", code);
  napi_throw(env, err);
  return nullptr;
}

Handle *get_handle(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "expected a b3wit handle"); return nullptr; }
  return (Handle *)p;
}

void finalize_handle(napi_env, void *data, void *) {
  Handle *h = (Handle *)data;
  if (h->key) api.commit_key_destroy(h->key);
  if (h->r1cs) api.r1cs_destroy(h->r1cs);
  if (h->comm) api.comm_destroy(h->comm);
  if (h->batch) api.batch_free(h->batch);
  if (h->ctx) api.destroy(h->ctx);
  delete h;
}

// identifyWasm(Buffer|TypedArray|ArrayBuffer) -> circuit id or -1
napi_value IdentifyWasm(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!load_api()) { napi_throw_error(env, nullptr, api.err.c_str()); return nullptr; }
  void *data = nullptr; size_t len = 0;
  bool is = false;
  napi_is_buffer(env, argv[0], &is);
  if (is) NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &len));
  else {
    napi_is_typedarray(env, argv[0], &is);
    if (is) {
      napi_typedarray_type t; napi_value ab; size_t off;
      NAPI_OK(napi_get_typedarray_info(env, argv[0], &t, &len, &data, &ab, &off));
    } else {
      napi_is_arraybuffer(env, argv[0], &is);
      if (!is) { napi_throw_type_error(env, nullptr, "identifyWasm: expected Buffer / Uint8Array / ArrayBuffer"); return nullptr; }
      NAPI_OK(napi_get_arraybuffer_info(env, argv[0], &data, &len));
    }
  }
  napi_value out;
  NAPI_OK(napi_create_int32(env, api.identify_wasm((const uint8_t *)data, len), &out));
  return out;
}

// create(circuitId, device) -> handle
napi_value Create(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!load_api()) { napi_throw_error(env, nullptr, api.err.c_str()); return nullptr; }
  int32_t circuit = -1, device = 0;
  NAPI_OK(napi_get_value_int32(env, argv[0], &circuit));
  if (argc > 1) napi_get_value_int32(env, argv[1], &device);
  Handle *h = new Handle;
  const int32_t rc = api.create(circuit, device, &h->ctx);
  if (rc != B3W_OK) {
    delete h;
    return throw_status(env, nullptr, rc, rc == B3W_E_NO_DEVICE
        ? "b3wit: no HIP device available (this addon has no CPU path)" : "b3wit: b3w_create failed");
  }
  napi_value out;
  NAPI_OK(napi_create_external(env, h, finalize_handle, nullptr, &out));
  return out;
}

// info(handle) -> { n32, prime (BigInt), witnessSize, inputSize, version:[maj,min,patch], publicWords }
napi_value Info(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  uint32_t n32, nwit, nin, ver[3];
  uint64_t prime[4];
  api.info(h->ctx, &n32, (uint8_t *)prime, &nwit, &nin, ver);
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
  napi_create_uint32(env, n32, &v); napi_set_named_property(env, o, "n32", v);
  napi_create_bigint_words(env, 0, 4, prime, &v); napi_set_named_property(env, o, "prime", v);
  napi_create_uint32(env, nwit, &v); napi_set_named_property(env, o, "witnessSize", v);
  napi_create_uint32(env, nin, &v); napi_set_named_property(env, o, "inputSize", v);
  napi_create_uint32(env, api.public_words(h->ctx), &v); napi_set_named_property(env, o, "publicWords", v);
  napi_value arr;
  napi_create_array_with_length(env, 3, &arr);
  for (int i = 0; i < 3; i++) { napi_create_uint32(env, ver[i], &v); napi_set_element(env, arr, i, v); }
  napi_set_named_property(env, o, "version", arr);
  return o;
}

// inputSignalSize(handle, hMSB, hLSB) — the (hMSB,hLSB) pair the reference passes to getInputSignalSize
napi_value InputSignalSize(napi_env env, napi_callback_info info) {
  size_t argc = 3; napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  uint32_t hi = 0, lo = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[1], &hi));
  NAPI_OK(napi_get_value_uint32(env, argv[2], &lo));
  napi_value out;
  NAPI_OK(napi_create_int32(env, api.input_signal_size(h->ctx, ((uint64_t)hi << 32) | lo), &out));
  return out;
}

// calcWitness(handle, hashes: Uint32Array [hMSB,hLSB]*, counts: Uint32Array, values: Uint8Array(32*sum)[, asWtns: boolean])
//   -> Uint8Array body, or the whole .wtns image (76-byte preamble + body, witness_calculator.js:208-272) when asWtns is true:
//   the body then lands behind the preamble straight from the device, no second 771 KB copy in JS
napi_value CalcWitness(napi_env env, napi_callback_info info) {
  size_t argc = 5; napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  napi_typedarray_type t; napi_value ab; size_t off;
  void *ph, *pc, *pv; size_t nh, nc, nv;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &t, &nh, &ph, &ab, &off));
  if (t != napi_uint32_array) { napi_throw_type_error(env, nullptr, "hashes: Uint32Array"); return nullptr; }
  NAPI_OK(napi_get_typedarray_info(env, argv[2], &t, &nc, &pc, &ab, &off));
  if (t != napi_uint32_array || nh != 2 * nc) { napi_throw_type_error(env, nullptr, "counts: Uint32Array, hashes twice as long"); return nullptr; }
  NAPI_OK(napi_get_typedarray_info(env, argv[3], &t, &nv, &pv, &ab, &off));
  if (t != napi_uint8_array) { napi_throw_type_error(env, nullptr, "values: Uint8Array"); return nullptr; }
  std::vector<uint64_t> hashes(nc);
  uint64_t total = 0;
  for (size_t i = 0; i < nc; i++) {
    hashes[i] = ((uint64_t)((uint32_t *)ph)[2 * i] << 32) | ((uint32_t *)ph)[2 * i + 1];
    total += ((uint32_t *)pc)[i];
  }
  if (nv < 32 * total) { napi_throw_range_error(env, nullptr, "values shorter than 32*sum(counts)"); return nullptr; }
  uint32_t nwit = 0;
  api.info(h->ctx, nullptr, nullptr, &nwit, nullptr, nullptr);
  bool as_wtns = false;
  if (argc >= 5) {
    napi_valuetype vt;
    NAPI_OK(napi_typeof(env, argv[4], &vt));
    if (vt == napi_boolean) NAPI_OK(napi_get_value_bool(env, argv[4], &as_wtns));
  }
  const size_t head = as_wtns ? 76 : 0;
  void *buf = nullptr; napi_value abuf, out;
  NAPI_OK(napi_create_arraybuffer(env, head + (size_t)nwit * 32, &buf, &abuf));
  const int32_t rc = api.calc_witness(h->ctx, hashes.data(), (const uint32_t *)pc, (const uint8_t *)pv, (uint32_t)nc, (uint8_t *)buf + head);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_calc_witness failed");
  if (as_wtns) api.write_wtns_header(h->ctx, (uint8_t *)buf);
  NAPI_OK(napi_create_typedarray(env, napi_uint8_array, head + (size_t)nwit * 32, abuf, 0, &out));
  return out;
}

napi_value WtnsHeader(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  void *p = nullptr; napi_value abuf, out;
  NAPI_OK(napi_create_arraybuffer(env, 76, &p, &abuf));
  api.write_wtns_header(h->ctx, (uint8_t *)p);
  NAPI_OK(napi_create_typedarray(env, napi_uint8_array, 76, abuf, 0, &out));
  return out;
}

// batchRun(handle, records: Uint32Array(n*inputSize)) -> { n, publicOutputs: Uint32Array, status: Int32Array }
napi_value BatchRun(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  napi_typedarray_type t; napi_value ab; size_t off, len; void *p;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &t, &len, &p, &ab, &off));
  if (t != napi_uint32_array) { napi_throw_type_error(env, nullptr, "records: Uint32Array"); return nullptr; }
  uint32_t nin = 0;
  api.info(h->ctx, nullptr, nullptr, nullptr, &nin, nullptr);
  if (len == 0 || len % nin) { napi_throw_range_error(env, nullptr, "records length must be a positive multiple of the input size"); return nullptr; }
  const uint32_t n = (uint32_t)(len / nin);
  if (!h->batch || h->batch_cap < n) {
    if (h->batch) { api.batch_free(h->batch); h->batch = nullptr; }
    const int32_t rc = api.batch_alloc(h->ctx, n, 0, &h->batch);
    if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_alloc failed");
    h->batch_cap = n;
  }
  int32_t rc = api.batch_run(h->batch, (const uint32_t *)p, n, nullptr);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_run failed");
  h->batch_n = n;
  h->generation++;
  const uint32_t npub = api.public_words(h->ctx);
  void *pp, *ps; napi_value abp, abs_, o, v;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * npub * 4, &pp, &abp));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 4, &ps, &abs_));
  rc = api.batch_outputs(h->batch, (uint32_t *)pp, (int32_t *)ps);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_outputs failed");
  NAPI_OK(napi_create_object(env, &o));
  napi_create_uint32(env, n, &v); napi_set_named_property(env, o, "n", v);
  napi_create_uint32(env, h->generation, &v); napi_set_named_property(env, o, "generation", v);
  napi_create_typedarray(env, napi_uint32_array, (size_t)n * npub, abp, 0, &v); napi_set_named_property(env, o, "publicOutputs", v);
  napi_create_typedarray(env, napi_int32_array, n, abs_, 0, &v); napi_set_named_property(env, o, "status", v);
  return o;
}

// batchFetch(handle, index[, generation]) -> Uint8Array body of witness `index` of the last batchRun
napi_value BatchFetch(napi_env env, napi_callback_info info) {
  size_t argc = 3; napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h || !fresh_result(env, h, argc, argv, 2)) return nullptr;
  uint32_t idx = 0, nwit = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[1], &idx));
  if (!h->batch) { napi_throw_error(env, nullptr, "batchFetch before batchRun"); return nullptr; }
  api.info(h->ctx, nullptr, nullptr, &nwit, nullptr, nullptr);
  void *body; napi_value abuf, out;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)nwit * 32, &body, &abuf));
  const int32_t rc = api.batch_fetch(h->batch, idx, (uint8_t *)body);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_fetch failed");
  NAPI_OK(napi_create_typedarray(env, napi_uint8_array, (size_t)nwit * 32, abuf, 0, &out));
  return out;
}

// batchWriteWtns(handle, first, count, dir, prefix[, generation]) -> number of .wtns files written (streamed D2H)
napi_value BatchWriteWtns(napi_env env, napi_callback_info info) {
  size_t argc = 6; napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h || !fresh_result(env, h, argc, argv, 5)) return nullptr;
  if (!h->batch) { napi_throw_error(env, nullptr, "batchWriteWtns before batchRun"); return nullptr; }
  uint32_t first = 0, count = 0;
  NAPI_OK(napi_get_value_uint32(env, argv[1], &first));
  NAPI_OK(napi_get_value_uint32(env, argv[2], &count));
  char dir[1024], prefix[256];
  size_t n1 = 0, n2 = 0;
  NAPI_OK(napi_get_value_string_utf8(env, argv[3], dir, sizeof dir, &n1));
  NAPI_OK(napi_get_value_string_utf8(env, argv[4], prefix, sizeof prefix, &n2));
  uint32_t written = 0;
  const int32_t rc = api.batch_write_wtns(h->batch, first, count, dir, prefix, &written);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_write_wtns failed");
  napi_value out;
  NAPI_OK(napi_create_uint32(env, written, &out));
  return out;
}

// batchVerify(handle[, generation]) -> Uint32Array of per-witness mismatch counts of the last batchRun (0 = valid witness).
// The result is sized from the handle's own batch, never from a caller-supplied count: b3w_batch_verify writes n * 4 bytes.
napi_value BatchVerify(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h || !fresh_result(env, h, argc, argv, 1)) return nullptr;
  if (!h->batch || !h->batch_n) { napi_throw_error(env, nullptr, "batchVerify before batchRun"); return nullptr; }
  const uint32_t n = h->batch_n;
  void *p; napi_value ab, out;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 4, &p, &ab));
  const int32_t rc = api.batch_verify(h->batch, (uint32_t *)p);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_verify failed");
  NAPI_OK(napi_create_typedarray(env, napi_uint32_array, n, ab, 0, &out));
  return out;
}

static const char *placement_name(int32_t p) { return p == B3W_PLACEMENT_MIXED ? "mixed" : p == B3W_PLACEMENT_INTERLEAVED ? "interleaved" : "plain"; }

// batchPlacement(handle) -> "mixed" | "interleaved" | "plain": where the body buffer of the last batchRun lives (b3w_bodies_alloc)
napi_value BatchPlacement(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  napi_value out;
  NAPI_OK(napi_create_string_utf8(env, placement_name(h->batch ? api.batch_placement(h->batch) : B3W_PLACEMENT_PLAIN), NAPI_AUTO_LENGTH, &out));
  return out;
}

// chainFold(handle, preimage: Buffer|Uint8Array, batchSteps, ring, withParents)
//   -> { nLeafSteps, nParentSteps, nChunks, pathLen, placement, publicOutputs: Uint32Array(steps*15), status: Int32Array, root: Uint32Array(8)
//        [, hOutAll: Uint32Array(all ranks' leaf steps * 8), hOutParentsAll: Uint32Array(all ranks' parent steps * 8) after commCreate] }
// The whole chained-mode pass of b3wit.h (b3w_chain_*) over one preimage on this handle's device.
napi_value ChainFold(napi_env env, napi_callback_info info) {
  size_t argc = 7; napi_value argv[7];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  void *data = nullptr; size_t len = 0;
  bool is = false;
  napi_is_buffer(env, argv[1], &is);
  if (is) NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &len));
  else {
    napi_typedarray_type t; napi_value ab; size_t off;
    NAPI_OK(napi_get_typedarray_info(env, argv[1], &t, &len, &data, &ab, &off));
    if (t != napi_uint8_array) { napi_throw_type_error(env, nullptr, "preimage: Buffer or Uint8Array"); return nullptr; }
  }
  if (!len) { napi_throw_range_error(env, nullptr, "preimage must not be empty"); return nullptr; }
  uint32_t batch_steps = 16384, ring = 2;
  bool with_parents = true;
  if (argc > 2) napi_get_value_uint32(env, argv[2], &batch_steps);
  if (argc > 3) napi_get_value_uint32(env, argv[3], &ring);
  if (argc > 4) napi_get_value_bool(env, argv[4], &with_parents);
  bool commit_only = false;                              // one commitment per step instead of the witness bodies (needs commitKey)
  if (argc > 5) napi_get_value_bool(env, argv[5], &commit_only);
  if (commit_only && !h->key) { napi_throw_error(env, nullptr, "commitOnly needs setCommitKey first"); return nullptr; }
  bool check = false;                                    // constraint check of every step witness in the ring (needs loadR1cs)
  if (argc > 6) napi_get_value_bool(env, argv[6], &check);
  if (check && (!h->r1cs || commit_only)) { napi_throw_error(env, nullptr, "checkConstraints needs loadR1cs first and witness bodies (no commitOnly)"); return nullptr; }
  const uint64_t nchunks = (len + 1023) / 1024;
  if (nchunks > 0xFFFFFFFFull) { napi_throw_range_error(env, nullptr, "preimage too large for one pass"); return nullptr; }
  b3w_chain *c = nullptr;
  uint64_t first = 0; uint32_t count = (uint32_t)nchunks;
  if (h->comm) api.chain_shard(nchunks, h->rank, h->nranks, &first, &count);     // after commCreate: this rank's share of the chunks
  int32_t rc = api.chain_create(h->ctx, len, first, count, batch_steps, ring, with_parents ? 1 : 0, &c);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_chain_create failed");
  if (commit_only) rc = api.chain_commit_only(c, h->key, nullptr);
  if (rc == B3W_OK && check) rc = api.chain_check_constraints(c, h->r1cs);
  if (rc == B3W_OK) rc = api.chain_run_leaves(c, (const uint8_t *)data, nullptr, nullptr, nullptr);
  if (rc == B3W_OK) rc = h->comm ? api.chain_run_parents_sharded(c, h->comm, nullptr, nullptr, nullptr)
                                 : api.chain_run_parents(c, nullptr, nullptr, nullptr, nullptr);
  uint64_t nleaf = 0, npar = 0, nch = 0; uint32_t plen = 0; int32_t placement = 0;
  api.chain_info(c, &nleaf, &npar, &nch, &plen, &placement);
  const uint64_t rows = nleaf + npar;
  void *pp = nullptr, *ps = nullptr, *pr = nullptr; napi_value abp, abs_, abr, o, v;
  if (rc == B3W_OK && (napi_create_arraybuffer(env, rows * 15 * 4, &pp, &abp) != napi_ok || napi_create_arraybuffer(env, rows * 4, &ps, &abs_) != napi_ok ||
                       napi_create_arraybuffer(env, 32, &pr, &abr) != napi_ok)) {
    api.chain_destroy(c);
    napi_throw_error(env, nullptr, "b3wit_napi: cannot allocate the result arrays");
    return nullptr;
  }
  if (rc == B3W_OK) rc = api.chain_outputs(c, (uint32_t *)pp, (int32_t *)ps, (uint32_t *)pr, nullptr);
  void *pc = nullptr; napi_value abc;
  if (rc == B3W_OK && commit_only) {
    if (napi_create_arraybuffer(env, rows * 64, &pc, &abc) != napi_ok) { api.chain_destroy(c); napi_throw_error(env, nullptr, "b3wit_napi: cannot allocate the points"); return nullptr; }
    rc = api.chain_commitments(c, (uint8_t *)pc, nullptr);
  }
  void *pv = nullptr; napi_value abv;
  if (rc == B3W_OK && check) {
    if (napi_create_arraybuffer(env, rows * 4, &pv, &abv) != napi_ok) { api.chain_destroy(c); napi_throw_error(env, nullptr, "b3wit_napi: cannot allocate the violation counts"); return nullptr; }
    rc = api.chain_violations(c, (uint32_t *)pv, nullptr);
  }
  // after joinRanks: the fold's exchange — every step's h_out of EVERY rank, global step order (b3w_chain_allgather_hout)
  void *phl = nullptr, *php = nullptr; napi_value abhl, abhp;
  const uint64_t hl_rows = api.chain_num_leaf_steps(len), hp_rows = with_parents ? api.chain_parent_row(nchunks, nchunks) : 0;
  if (rc == B3W_OK && h->comm) {
    if (napi_create_arraybuffer(env, hl_rows * 32, &phl, &abhl) != napi_ok || napi_create_arraybuffer(env, hp_rows * 32, &php, &abhp) != napi_ok) {
      api.chain_destroy(c); napi_throw_error(env, nullptr, "b3wit_napi: cannot allocate the gathered h_out"); return nullptr;
    }
    rc = api.chain_allgather_hout_host(c, h->comm, (uint32_t *)phl, hp_rows ? (uint32_t *)php : nullptr, nullptr);
  }
  api.chain_destroy(c);
  if (rc != B3W_OK) return throw_status(env, h, rc, "chained pass failed");
  NAPI_OK(napi_create_object(env, &o));
  if (h->comm) {
    napi_create_typedarray(env, napi_uint32_array, hl_rows * 8, abhl, 0, &v); napi_set_named_property(env, o, "hOutAll", v);
    napi_create_typedarray(env, napi_uint32_array, hp_rows * 8, abhp, 0, &v); napi_set_named_property(env, o, "hOutParentsAll", v);
  }
  napi_create_double(env, (double)nleaf, &v); napi_set_named_property(env, o, "nLeafSteps", v);
  napi_create_double(env, (double)npar, &v); napi_set_named_property(env, o, "nParentSteps", v);
  napi_create_double(env, (double)nch, &v); napi_set_named_property(env, o, "nChunks", v);
  napi_create_double(env, (double)first, &v); napi_set_named_property(env, o, "firstChunk", v);
  napi_create_uint32(env, count, &v); napi_set_named_property(env, o, "nChunksLocal", v);
  napi_create_uint32(env, plen, &v); napi_set_named_property(env, o, "pathLen", v);
  napi_create_string_utf8(env, placement_name(placement), NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "placement", v);
  napi_create_typedarray(env, napi_uint32_array, rows * 15, abp, 0, &v); napi_set_named_property(env, o, "publicOutputs", v);
  napi_create_typedarray(env, napi_int32_array, rows, abs_, 0, &v); napi_set_named_property(env, o, "status", v);
  napi_create_typedarray(env, napi_uint32_array, 8, abr, 0, &v); napi_set_named_property(env, o, "root", v);
  if (commit_only) { napi_create_typedarray(env, napi_uint8_array, rows * 64, abc, 0, &v); napi_set_named_property(env, o, "commitments", v); }
  if (check) { napi_create_typedarray(env, napi_uint32_array, rows, abv, 0, &v); napi_set_named_property(env, o, "violations", v); }
  return o;
}

// commitKey(handle, curve: 0 = BN254 G1 | 1 = Vesta, firstSlot, generators: Uint8Array((witnessSize - firstSlot) * 64)
//           [, windowBits: 0 (automatic) | 12 | 16 | 18[, folded: Uint8Array(witnessSize - firstSlot)]])
// installs the commitment key of this handle (tables on the device).  folded: include/b3wit.h "FOLDED keys" — the generators are
// then the folded ones and folded[k] says what became of slot k (0 kept, 1 folded away, 0x80 | i only bit i of the word);
// tools/fold_key.py writes both files for a circuit and a key.
napi_value CommitKey(napi_env env, napi_callback_info info) {
  size_t argc = 6; napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  int32_t curve = 0; uint32_t first = 0, nwit = 0;
  NAPI_OK(napi_get_value_int32(env, argv[1], &curve));
  NAPI_OK(napi_get_value_uint32(env, argv[2], &first));
  napi_typedarray_type t; napi_value ab; size_t off, len; void *p;
  NAPI_OK(napi_get_typedarray_info(env, argv[3], &t, &len, &p, &ab, &off));
  api.info(h->ctx, nullptr, nullptr, &nwit, nullptr, nullptr);
  if (t != napi_uint8_array || first >= nwit || len != (size_t)(nwit - first) * 64) {
    napi_throw_type_error(env, nullptr, "generators: Uint8Array with 64 bytes (x, y little-endian) per committed slot");
    return nullptr;
  }
  if (h->key) { api.commit_key_destroy(h->key); h->key = nullptr; }
  uint32_t window = 0;
  if (argc > 4) {
    napi_valuetype vt;
    NAPI_OK(napi_typeof(env, argv[4], &vt));
    if (vt == napi_number) NAPI_OK(napi_get_value_uint32(env, argv[4], &window));
  }
  const uint8_t *folded = nullptr;
  if (argc > 5) {
    napi_valuetype vt;
    NAPI_OK(napi_typeof(env, argv[5], &vt));
    if (vt == napi_object) {
      napi_typedarray_type ft; napi_value fab; size_t foff, flen; void *fp;
      NAPI_OK(napi_get_typedarray_info(env, argv[5], &ft, &flen, &fp, &fab, &foff));
      if (ft != napi_uint8_array || flen != (size_t)(nwit - first)) { napi_throw_type_error(env, nullptr, "folded: Uint8Array with one byte per committed slot"); return nullptr; }
      folded = (const uint8_t *)fp;
    }
  }
  const int32_t rc = api.commit_key_create_folded(h->ctx, curve, first, (const uint8_t *)p, folded, window, &h->key);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_commit_key_create_folded failed");
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

// batchCommit(handle[, generation]) -> { points: Uint8Array(n * 64), status: Int32Array(n) }: Pedersen commitments of the last batchRun
napi_value BatchCommit(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h || !fresh_result(env, h, argc, argv, 1)) return nullptr;
  if (!h->batch || !h->key || !h->batch_n) { napi_throw_error(env, nullptr, "batchCommit needs commitKey and a batchRun"); return nullptr; }
  void *pp, *ps; napi_value abp, abs_, o, v;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)h->batch_n * 64, &pp, &abp));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)h->batch_n * 4, &ps, &abs_));
  const int32_t rc = api.batch_commit(h->batch, h->key, (uint8_t *)pp, (int32_t *)ps);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_commit failed");
  NAPI_OK(napi_create_object(env, &o));
  napi_create_typedarray(env, napi_uint8_array, (size_t)h->batch_n * 64, abp, 0, &v); napi_set_named_property(env, o, "points", v);
  napi_create_typedarray(env, napi_int32_array, h->batch_n, abs_, 0, &v); napi_set_named_property(env, o, "status", v);
  return o;
}

// r1csLoad(handle, image: Uint8Array of an iden3 .r1cs file) -> { nConstraints, nWires, nTerms }: the constraint system
// batchR1csCheck evaluates (b3w_r1cs_create)
napi_value R1csLoad(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  napi_typedarray_type t; napi_value ab; size_t off, len; void *p;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &t, &len, &p, &ab, &off));
  if (t != napi_uint8_array) { napi_throw_type_error(env, nullptr, "image: Uint8Array / Buffer with the bytes of a .r1cs file"); return nullptr; }
  if (h->r1cs) { api.r1cs_destroy(h->r1cs); h->r1cs = nullptr; }
  const int32_t rc = api.r1cs_create(h->ctx, (const uint8_t *)p, len, &h->r1cs);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_r1cs_create failed");
  uint32_t m = 0, nw = 0; uint64_t nt = 0;
  api.r1cs_info(h->r1cs, &m, &nw, &nt, nullptr, nullptr, nullptr);
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
  napi_create_uint32(env, m, &v); napi_set_named_property(env, o, "nConstraints", v);
  napi_create_uint32(env, nw, &v); napi_set_named_property(env, o, "nWires", v);
  napi_create_double(env, (double)nt, &v); napi_set_named_property(env, o, "nTerms", v);
  return o;
}

// batchR1csCheck(handle[, generation]) -> { violations: Uint32Array(n), first: Uint32Array(n) }: A z * B z - C z = 0 for every
// constraint and every witness of the last batchRun, on the device (0 violations = a valid witness; first = 0xFFFFFFFF then)
napi_value BatchR1csCheck(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h || !fresh_result(env, h, argc, argv, 1)) return nullptr;
  if (!h->batch || !h->r1cs || !h->batch_n) { napi_throw_error(env, nullptr, "batchR1csCheck needs r1csLoad and a batchRun"); return nullptr; }
  void *pv, *pf; napi_value abv, abf, o, v;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)h->batch_n * 4, &pv, &abv));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)h->batch_n * 4, &pf, &abf));
  const int32_t rc = api.batch_r1cs_check(h->batch, h->r1cs, (uint32_t *)pv, (uint32_t *)pf);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_r1cs_check failed");
  NAPI_OK(napi_create_object(env, &o));
  napi_create_typedarray(env, napi_uint32_array, h->batch_n, abv, 0, &v); napi_set_named_property(env, o, "violations", v);
  napi_create_typedarray(env, napi_uint32_array, h->batch_n, abf, 0, &v); napi_set_named_property(env, o, "first", v);
  return o;
}

// commitRecords(handle, records: Uint32Array(n * inputSize)) -> { points: Uint8Array(n * 64), publicOutputs: Uint32Array(n * 16 | 15),
// status: Int32Array(n) }: the commitments of the witnesses of these records, without the witnesses (b3w_commit_records)
napi_value CommitRecords(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  if (!h->key) { napi_throw_error(env, nullptr, "commitRecords needs commitKey"); return nullptr; }
  uint32_t nin = 0;
  api.info(h->ctx, nullptr, nullptr, nullptr, &nin, nullptr);
  napi_typedarray_type t; napi_value ab; size_t off, len; void *p;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &t, &len, &p, &ab, &off));
  if (t != napi_uint32_array || nin == 0 || len == 0 || len % nin) { napi_throw_type_error(env, nullptr, "records: Uint32Array of whole input records"); return nullptr; }
  const uint32_t n = (uint32_t)(len / nin), npub = api.public_words(h->ctx);
  void *pp, *pq, *ps; napi_value abp, abq, abs_, o, v;
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 64, &pp, &abp));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * npub * 4, &pq, &abq));
  NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 4, &ps, &abs_));
  const int32_t rc = api.commit_records(h->ctx, h->key, (const uint32_t *)p, n, (uint8_t *)pp, (uint32_t *)pq, (int32_t *)ps);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_commit_records failed");
  NAPI_OK(napi_create_object(env, &o));
  napi_create_typedarray(env, napi_uint8_array, (size_t)n * 64, abp, 0, &v); napi_set_named_property(env, o, "points", v);
  napi_create_typedarray(env, napi_uint32_array, (size_t)n * npub, abq, 0, &v); napi_set_named_property(env, o, "publicOutputs", v);
  napi_create_typedarray(env, napi_int32_array, n, abs_, 0, &v); napi_set_named_property(env, o, "status", v);
  return o;
}

// commUniqueId() -> Uint8Array(128): rank 0 creates it and hands it to the other ranks (file, socket, env)
napi_value CommUniqueId(napi_env env, napi_callback_info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.err.c_str()); return nullptr; }
  void *p; napi_value ab, out;
  NAPI_OK(napi_create_arraybuffer(env, B3W_COMM_ID_BYTES, &p, &ab));
  const int32_t rc = api.comm_unique_id((uint8_t *)p);
  if (rc != B3W_OK) return throw_status(env, nullptr, rc, "b3wit: librccl not available");
  NAPI_OK(napi_create_typedarray(env, napi_uint8_array, B3W_COMM_ID_BYTES, ab, 0, &out));
  return out;
}

// commCreate(handle, id: Uint8Array(128), rank, nranks): this handle's device joins the RCCL communicator
napi_value CommCreate(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  napi_typedarray_type t; napi_value ab; size_t off, len; void *p;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &t, &len, &p, &ab, &off));
  if (t != napi_uint8_array || len != B3W_COMM_ID_BYTES) { napi_throw_type_error(env, nullptr, "id: Uint8Array(128) from commUniqueId()"); return nullptr; }
  int32_t rank = 0, nranks = 1;
  NAPI_OK(napi_get_value_int32(env, argv[2], &rank));
  NAPI_OK(napi_get_value_int32(env, argv[3], &nranks));
  if (h->comm) { api.comm_destroy(h->comm); h->comm = nullptr; }
  const int32_t rc = api.comm_create(h->ctx, (const uint8_t *)p, rank, nranks, &h->comm);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_comm_create failed");
  h->rank = rank;
  h->nranks = nranks;
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

// commCreateHost(handle, name: "/unique-to-the-job", rank, nranks): the same communicator over the host's shared memory
// (b3w_comm_create_host: several ranks on one GPU, or no RCCL)
napi_value CommCreateHost(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h) return nullptr;
  char name[256]; size_t nlen = 0;
  NAPI_OK(napi_get_value_string_utf8(env, argv[1], name, sizeof name, &nlen));
  int32_t rank = 0, nranks = 1;
  NAPI_OK(napi_get_value_int32(env, argv[2], &rank));
  NAPI_OK(napi_get_value_int32(env, argv[3], &nranks));
  if (h->comm) { api.comm_destroy(h->comm); h->comm = nullptr; }
  const int32_t rc = api.comm_create_host(h->ctx, name, rank, nranks, &h->comm);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_comm_create_host failed");
  h->rank = rank;
  h->nranks = nranks;
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

// batchAllgatherPublic(handle[, generation]) -> Uint32Array(nranks * n * publicWords): the last batchRun's public outputs of every rank
napi_value BatchAllgatherPublic(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  Handle *h = get_handle(env, argv[0]);
  if (!h || !fresh_result(env, h, argc, argv, 1)) return nullptr;
  if (!h->batch || !h->comm || !h->batch_n) { napi_throw_error(env, nullptr, "batchAllgatherPublic needs commCreate and a batchRun"); return nullptr; }
  const size_t words = (size_t)h->nranks * h->batch_n * api.public_words(h->ctx);
  void *p; napi_value ab, out;
  NAPI_OK(napi_create_arraybuffer(env, words * 4, &p, &ab));
  const int32_t rc = api.batch_allgather_public(h->batch, h->comm, (uint32_t *)p);
  if (rc != B3W_OK) return throw_status(env, h, rc, "b3w_batch_allgather_public failed");
  NAPI_OK(napi_create_typedarray(env, napi_uint32_array, words, ab, 0, &out));
  return out;
}

napi_value AbiVersion(napi_env env, napi_callback_info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.err.c_str()); return nullptr; }
  napi_value out;
  napi_create_uint32(env, api.abi_version(), &out);
  return out;
}

napi_value Init(napi_env env, napi_value exports) {
  const napi_property_descriptor props[] = {
      {"abiVersion", nullptr, AbiVersion, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"identifyWasm", nullptr, IdentifyWasm, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"create", nullptr, Create, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"info", nullptr, Info, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"inputSignalSize", nullptr, InputSignalSize, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"calcWitness", nullptr, CalcWitness, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"wtnsHeader", nullptr, WtnsHeader, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchRun", nullptr, BatchRun, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchFetch", nullptr, BatchFetch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchWriteWtns", nullptr, BatchWriteWtns, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchVerify", nullptr, BatchVerify, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchPlacement", nullptr, BatchPlacement, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"chainFold", nullptr, ChainFold, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"commitKey", nullptr, CommitKey, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchCommit", nullptr, BatchCommit, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"r1csLoad", nullptr, R1csLoad, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchR1csCheck", nullptr, BatchR1csCheck, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"commitRecords", nullptr, CommitRecords, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"commUniqueId", nullptr, CommUniqueId, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"commCreate", nullptr, CommCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"commCreateHost", nullptr, CommCreateHost, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"batchAllgatherPublic", nullptr, BatchAllgatherPublic, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
