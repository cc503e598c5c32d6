/* b3wit_demo.c — the C-ABI from plain C, no Python, no Node: what a native caller of the reference's witness
 * calculator (e.g. a port of generate_witness.js:10-18, or the Rust fold driver through FFI) does.
 *
 *   gcc -O2 -Iinclude -o b3wit_demo examples/c/b3wit_demo.c -ldl
 *   ./b3wit_demo <libb3wit.so> <out.wtns>          one blake3_compression witness (the reference's testInp) -> .wtns
 *   ./b3wit_demo <libb3wit.so> <out.wtns> <n>      additionally a batch of n witnesses, checked on the device
 *
 * Inputs = build/blake3_compression/testInp (h = IV, m = LCG(6429) words, b = 64, d = 0, t = [0, 0];
 * test/witness_gen.test.ts:26,36), so out.wtns must equal the reference's committed witness.wtns byte for byte. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "b3wit.h"

static uint64_t fnv1a64(const char *s) {            /* witness_calculator.js:325-337 */
  uint64_t h = 0xCBF29CE484222325ull;
  for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ull; }
  return h;
}

#define LOAD(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(so, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <libb3wit.so> <out.wtns> [batch]\n", argv[0]); return 2; }
  void *so = dlopen(argv[1], RTLD_NOW);
  if (!so) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  LOAD(b3w_create) LOAD(b3w_destroy) LOAD(b3w_info) LOAD(b3w_calc_witness) LOAD(b3w_write_wtns_header) LOAD(b3w_last_error)
  LOAD(b3w_batch_alloc) LOAD(b3w_batch_run) LOAD(b3w_batch_verify) LOAD(b3w_batch_outputs) LOAD(b3w_batch_free) LOAD(b3w_batch_placement)

  b3w_ctx *ctx = NULL;
  int32_t rc = p_b3w_create(B3W_CIRCUIT_COMPRESSION_BN254, 0, &ctx);
  if (rc) { fprintf(stderr, "b3w_create: status %d%s\n", rc, rc == B3W_E_NO_DEVICE ? " (no HIP device; there is no CPU path)" : ""); return 1; }
  uint32_t nwit = 0, nin = 0;
  p_b3w_info(ctx, NULL, NULL, &nwit, &nin, NULL);

  /* the reference's testInp: LCG(6429), one draw burned, 16 message words (test/utils.ts:4-21,34-56) */
  uint32_t rec[28], seed = 6429;
  const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
  memcpy(rec, IV, 32);
  seed = 1664525u * seed + 1013904223u;
  for (int i = 0; i < 16; i++) { seed = 1664525u * seed + 1013904223u; rec[8 + i] = seed; }
  rec[24] = 0; rec[25] = 0; rec[26] = 64; rec[27] = 0;

  /* inputs as the loader passes them: name hash, count, values as 32-byte little-endian field elements */
  const char *names[5] = {"h", "m", "t", "b", "d"};
  const uint32_t counts[5] = {8, 16, 2, 1, 1};
  uint64_t hashes[5];
  uint8_t values[28 * 32];
  memset(values, 0, sizeof values);
  for (int k = 0; k < 5; k++) hashes[k] = fnv1a64(names[k]);
  for (int i = 0; i < 28; i++) memcpy(values + 32 * i, &rec[i], 4);
  uint8_t *body = malloc((size_t)nwit * 32);
  rc = p_b3w_calc_witness(ctx, hashes, counts, values, 5, body);
  if (rc) { char msg[512]; p_b3w_last_error(ctx, msg, sizeof msg); fprintf(stderr, "b3w_calc_witness: status %d: %s\n", rc, msg); return 1; }
  uint8_t hdr[76];
  p_b3w_write_wtns_header(ctx, hdr);
  FILE *f = fopen(argv[2], "wb");
  if (!f || fwrite(hdr, 1, 76, f) != 76 || fwrite(body, 1, (size_t)nwit * 32, f) != (size_t)nwit * 32) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  fclose(f);
  printf("witness: %u signals, out[0] = %u, wrote %s\n", nwit, *(uint32_t *)(body + 32), argv[2]);

  if (argc > 3) {
    const uint32_t n = (uint32_t)atoi(argv[3]);
    uint32_t *recs = malloc((size_t)n * 28 * 4), *pub = malloc((size_t)n * 16 * 4), *mm = malloc((size_t)n * 4);
    int32_t *st = malloc((size_t)n * 4);
    for (uint32_t i = 0; i < n; i++) {                 /* instance i from LCG(6429 + i) */
      uint32_t s = 6429 + i, *r = recs + (size_t)i * 28;
      for (int j = 0; j < 26; j++) { s = 1664525u * s + 1013904223u; r[j] = s; }
      s = 1664525u * s + 1013904223u; r[26] = s % 65;
      s = 1664525u * s + 1013904223u; r[27] = s % 16;
    }
    b3w_batch *b = NULL;
    rc = p_b3w_batch_alloc(ctx, n, 0, &b);
    if (!rc) rc = p_b3w_batch_run(b, recs, n, NULL);
    if (!rc) rc = p_b3w_batch_outputs(b, pub, st);
    if (!rc) rc = p_b3w_batch_verify(b, mm);
    if (rc) { char msg[512]; p_b3w_last_error(ctx, msg, sizeof msg); fprintf(stderr, "batch: status %d: %s\n", rc, msg); return 1; }
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n; i++) bad += (st[i] != 0) + (mm[i] != 0);
    printf("batch: %u witnesses, %u not ok, placement %s, out[0] of the last = %u\n", n, bad,
           p_b3w_batch_placement(b) == B3W_PLACEMENT_MIXED ? "mixed" : "plain", pub[(size_t)(n - 1) * 16]);
    p_b3w_batch_free(b);
    if (bad) return 1;
  }
  p_b3w_destroy(ctx);
  return 0;
}
