// b3w_kernels.h — host-visible launch entry of b3w_kernels.hip
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

// number of tuning variants of the compression kernel (witnesses per wave, store flavour)
#define B3W_NUM_VARIANTS 8

extern "C" int b3w_launch_batch(int kind, int variant, const uint32_t *d_recs, uint32_t n, uint8_t *d_out,
                                uint64_t pitch, const uint32_t *d_table, uint32_t nwit, uint32_t *d_pub,
                                int32_t *d_status, const void *d_aux, hipStream_t stream);
