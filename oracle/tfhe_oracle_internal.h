/* internal helpers shared between the oracle's translation units (test infrastructure) */
#ifndef TFHE_ORACLE_INTERNAL_H
#define TFHE_ORACLE_INTERNAL_H
#include <stdint.h>
#include "tfhe_oracle.h"
uint32_t orc_log2_u32(uint32_t x);
void orc_ext_product_exact(uint64_t *acc, const uint64_t *ct1, const uint64_t *ggsw, uint32_t k,
                           uint32_t N, uint32_t base_log, uint32_t level, int64_t *digit_buf,
                           uint64_t *states);
void orc_ext_product_fft(uint64_t *acc, const uint64_t *ct1, const double *ggsw_f, uint32_t k,
                         uint32_t N, uint32_t base_log, uint32_t level, uint64_t *states,
                         int64_t *digits, double *fbuf, double *outbuf);
void orc_ggsw_encrypt(orc_rng *r, uint64_t *ggsw, uint64_t cleartext, const uint64_t *glwe_sk,
                      uint32_t k, uint32_t N, uint32_t base_log, uint32_t level,
                      uint32_t noise_bound_log2);
#endif
