// b3w_r1cs_defs.h — what the constraint check's host code (b3w_r1cs_host.cpp, no HIP) and its kernels share
#pragma once
#include <stdint.h>

struct B3wField {
  uint32_t p[8];      // modulus of the circuit's field, little-endian limbs
  uint32_t r2[8];     // 2^512 mod p
  uint32_t inv;       // -p^-1 mod 2^32
};
#define B3W_R1CS_TILE 1024u
#define B3W_R1CS_NOT_SMALL ((long long)0x8000000000000000ull)
