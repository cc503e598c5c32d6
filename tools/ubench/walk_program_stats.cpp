// walk_program_stats.cpp — prints the walk program's per-unit counts of a derived system (no GPU; `<unit>`: that unit's general rows, entry by entry):
//   g++ -std=c++17 -O1 -I hot-proofs-blake3-circom_amd/csrc tools/ubench/walk_program_stats.cpp hot-proofs-blake3-circom_amd/csrc/b3w_r1cs_host.cpp -o /tmp/wps
//   zcat hot-proofs-blake3-circom_amd/constraints/blake3_nova_vesta.r1cs.gz > /tmp/x.r1cs && /tmp/wps /tmp/x.r1cs 23291 vesta [<unit>]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <string>
#include <vector>
#include "b3w_r1cs_host.h"
int main(int argc, char **argv) {
  if (argc < 4) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> img;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) img.insert(img.end(), buf, buf + n);
  fclose(f);
  static const uint64_t BN[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static const uint64_t VE[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};
  B3wR1csHost H;
  if (!b3w_r1cs_host_build(img.data(), img.size(), (const uint8_t *)(!strcmp(argv[3], "vesta") ? VE : BN), (uint32_t)atoi(argv[2]), &H)) { fprintf(stderr, "%s\n", H.error.c_str()); return 1; }
  printf("walk %d units %u exp_slots %u max_gen %u max_ent %u max_runs %u\n", (int)H.walk, H.wunits, H.wexp_slots, H.wmax_gen, H.wmax_ent, H.wmax_runs);
  {
    const size_t xw = (H.wexp_slots >> 6) + 1u, gr2 = (H.wmax_gen + 1u) & ~1u;      // (walk_smem in b3w_r1cs_walk.hip)
    const size_t smem = 8u * (2u * (size_t)B3W_R1CS_TILE + H.wexp_slots + 36u + xw + 12u * (size_t)H.wmax_gen + 20u + ((H.ncoef + 1u) & ~1u) + 16u * (size_t)H.wunits +
                              (size_t)H.wstatic_words * H.wunits) + 4u * (4u * gr2 + (size_t)H.wunits * B3W_WT_WORDS + 4u + 9u + 8u) + 32u;
    printf("ncoef %u static_words %u: LDS of a workgroup %zu B (two to a CU: 81 920 each)\n", H.ncoef, H.wstatic_words, smem);
  }
  {
    std::vector<uint8_t> used(H.ncoef, 0);
    size_t others = 0;
    for (uint32_t u = 0; u < H.wunits; u++) {
      const uint32_t *w = &H.wtile[(size_t)u * B3W_WT_WORDS];
      for (uint32_t e = w[B3W_WT_ENT_RUNS]; e < w[B3W_WT_ENT_N]; e++) {
        const uint32_t ew = H.went_w[w[B3W_WT_ENT_OFF] + e], em = H.went_m[w[B3W_WT_ENT_OFF] + e];
        if ((em & 4u) || (ew >> 31)) continue;
        used[ew >> 16] = 1; others++;
      }
    }
    size_t nu = 0; for (uint8_t x : used) nu += x;
    printf("terms with a coefficient from the table: %zu, distinct coefficients %zu of %u\n", others, nu, H.ncoef);
  }
  printf("unit src  gen  ent(runs)  ttruns  exp  rows\n");
  unsigned tg = 0, te = 0, tr = 0, tx = 0;
  for (uint32_t u = 0; u < H.wunits; u++) {
    const uint32_t *w = &H.wtile[(size_t)u * B3W_WT_WORDS];
    printf("%3u %3u %5u %5u(%4u) %5u %5u %5u\n", u, w[B3W_WT_SRC], w[B3W_WT_GEN_N], w[B3W_WT_ENT_N], w[B3W_WT_ENT_RUNS], w[B3W_WT_RUN_N], w[B3W_WT_EXP_N], w[B3W_WT_NROWS]);
    tg += w[B3W_WT_GEN_N]; te += w[B3W_WT_ENT_N]; tr += w[B3W_WT_RUN_N]; tx += w[B3W_WT_EXP_N];
  }
  printf("sum     %5u %5u       %5u %5u\n", tg, te, tr, tx);
  if (argc > 4) {                                            // one unit's general rows, entry by entry
    const uint32_t u = (uint32_t)atoi(argv[4]);
    const uint32_t *w = &H.wtile[(size_t)u * B3W_WT_WORDS];
    std::vector<std::vector<std::string>> rows(w[B3W_WT_GEN_N]);
    for (uint32_t e = 0; e < w[B3W_WT_ENT_N]; e++) {
      const uint32_t ew = H.went_w[w[B3W_WT_ENT_OFF] + e], em = H.went_m[w[B3W_WT_ENT_OFF] + e];
      if (em & 4u) continue;
      char b[96];
      const char part = "ABC"[em & 3u];
      if (e < w[B3W_WT_ENT_RUNS]) snprintf(b, sizeof b, "%c:run(%s%u len %u <<%u%s)", part, (ew & 0xFFFF) < 1024 ? "e" : "x", (ew & 0xFFFF) < 1024 ? ew & 0xFFFF : (ew & 0xFFFF) - 1024, (ew >> 16) & 0x7F, (ew >> 23) & 0x3F, (ew >> 29) & 1 ? " neg" : "");
      else if (ew >> 31) snprintf(b, sizeof b, "%c:%s2^%u*%s%u", part, (ew >> 22) & 1 ? "-" : "+", (ew >> 16) & 63, (ew & 0xFFFF) < 1024 ? "e" : "x", (ew & 0xFFFF) < 1024 ? ew & 0xFFFF : (ew & 0xFFFF) - 1024);
      else snprintf(b, sizeof b, "%c:c%u*%s%u", part, ew >> 16, (ew & 0xFFFF) < 1024 ? "e" : "x", (ew & 0xFFFF) < 1024 ? ew & 0xFFFF : (ew & 0xFFFF) - 1024);
      rows[em >> 8].push_back(b);
    }
    for (size_t g = 0; g < rows.size(); g++) { printf("row %3zu:", g); for (auto &x : rows[g]) printf(" %s", x.c_str()); printf("\n"); }
  }
  return 0;
}
