/*
 * kernels.h - device entry points shared between the .hip translation units
 * and the host side of the C-ABI (host_api.hip).
 */
#ifndef LDA_KERNELS_H
#define LDA_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* checksum_kernels.hip */
extern "C" __global__ void
lda_crc32_batch_kernel(uint64_t n_chunks, const uint8_t *base,
		       const uint64_t *offsets, const uint64_t *nbytes,
		       const uint32_t *init, uint32_t *out,
		       const uint32_t *g_tables, const uint32_t *xpow8);
extern "C" __global__ void
lda_adler32_batch_kernel(uint64_t n_chunks, const uint8_t *base,
			 const uint64_t *offsets, const uint64_t *nbytes,
			 const uint32_t *init, uint32_t *out);

/* CRC constant tables, generated on the host at first use (host_api.hip) */
#define LDA_CRC_TABLE_WORDS (17 * 256)
#define LDA_CRC_XPOW_WORDS 1024

#endif /* LDA_KERNELS_H */
