/*
 * kernels.h - device entry points shared between the .hip translation units
 * and the host side of the C-ABI (host_*.hip).
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

/* inflate_kernel.hip */
extern "C" __global__ void
lda_inflate_batch_kernel(uint64_t n_chunks, int format, uint32_t lpw,
			 const uint8_t *in_base,
			 const uint64_t *in_offsets, const uint64_t *in_nbytes,
			 uint8_t *out_base, const uint64_t *out_offsets,
			 const uint64_t *out_avail, int32_t *results,
			 uint64_t *actual_in, uint64_t *actual_out);
extern "C" __global__ void
lda_inflate_wave_kernel(uint64_t n_chunks, int format, uint32_t *tokscratch,
			uint32_t *next_stream, const uint32_t *order,
			const uint8_t *in_base,
			const uint64_t *in_offsets, const uint64_t *in_nbytes,
			uint8_t *out_base, const uint64_t *out_offsets,
			const uint64_t *out_avail, int32_t *results,
			uint64_t *actual_in, uint64_t *actual_out);
extern "C" __global__ void
lda_inflate_order_kernel(uint64_t n, const uint64_t *in_nbytes, const uint64_t *out_avail,
			 uint32_t *order);
extern "C" size_t lda_inflate_tokcap(void);
extern "C" size_t lda_inflate_window_bytes(void);
extern "C" __global__ void
lda_inflate_finalize_kernel(uint64_t n_chunks, int format, int exact_fill,
			    const uint8_t *in_base, const uint64_t *in_offsets,
			    const uint64_t *out_avail, const uint32_t *sums,
			    int32_t *results, uint64_t *actual_in,
			    uint64_t *actual_out);

/* deflate_kernel.hip */
#define LDA_DEFLATE_THREADS 1024	/* one workgroup (16 waves) per buffer */
extern "C" __global__ void
lda_deflate_batch_kernel(uint64_t n_chunks, int format, int level,
			 uint32_t depth, uint32_t nice, uint32_t mode,
			 const uint8_t *in_base, const uint64_t *in_offsets,
			 const uint64_t *in_nbytes, uint8_t *out_base,
			 const uint64_t *out_offsets, const uint64_t *out_avail,
			 uint64_t *out_nbytes, const uint32_t *sums,
			 uint64_t *seq_scratch, const uint32_t *seg_info,
			 uint32_t *next_chunk);
/* same body with the min-cost parse compiled in: levels 10-12 */
extern "C" __global__ void
lda_deflate_opt_kernel(uint64_t n_chunks, int format, int level,
		       uint32_t depth, uint32_t nice, uint32_t mode,
		       const uint8_t *in_base, const uint64_t *in_offsets,
		       const uint64_t *in_nbytes, uint8_t *out_base,
		       const uint64_t *out_offsets, const uint64_t *out_avail,
		       uint64_t *out_nbytes, const uint32_t *sums,
		       uint64_t *seq_scratch, const uint32_t *seg_info,
		       uint32_t *next_chunk);
/* deflate_small.hip: buffers of at most lda_deflate_small_max() bytes */
#define LDA_DEFLATE_SMALL_THREADS 256
extern "C" __global__ void
lda_deflate_small_kernel(uint64_t n_chunks, int format, int level,
			 uint32_t depth, uint32_t nice, uint32_t mode,
			 const uint8_t *in_base, const uint64_t *in_offsets,
			 const uint64_t *in_nbytes, uint8_t *out_base,
			 const uint64_t *out_offsets, const uint64_t *out_avail,
			 uint64_t *out_nbytes, const uint32_t *sums,
			 uint64_t *seq_scratch, const uint32_t *seg_info,
			 uint32_t *next_chunk);
extern "C" size_t lda_deflate_small_lds_bytes(void);
extern "C" size_t lda_deflate_small_max(void);
extern "C" size_t lda_deflate_small_wgs(void);	/* workgroups per CU it is built for */
extern "C" size_t lda_deflate_tile(void);	/* positions per tile (dictionary granularity) */
extern "C" size_t lda_deflate_lds_bytes(void);
extern "C" size_t lda_deflate_seq_words(void);	/* u64 words of HBM scratch per workgroup (token list, saved histograms) */

extern "C" size_t lda_inflate_lds_per_stream(void);
extern "C" size_t lda_inflate_lds_shared(void);

/* compact_kernels.hip */
extern "C" __global__ void
lda_scan_local_kernel(uint64_t n, const uint64_t *sizes, uint64_t *offsets,
		      uint64_t *block_sums);
extern "C" __global__ void
lda_scan_blocks_kernel(uint64_t nblocks, uint64_t *block_sums);
extern "C" __global__ void
lda_compact_copy_kernel(uint64_t n, const uint8_t *in_base,
			const uint64_t *in_offsets, const uint64_t *sizes,
			uint8_t *out_base, uint64_t *offsets,
			const uint64_t *block_sums);

/* selfcheck_kernels.hip: the hardware behaviours the kernels rely on, checked
 * per device (counters: [0] lanes, [1] order mismatches, [2] same-instruction
 * conflicts seen, [3] loads, [4] stale loads) */
extern "C" __global__ void
lda_selfcheck_lds_order_kernel(uint64_t *counters);
extern "C" __global__ void
lda_selfcheck_visibility_kernel(uint8_t *buf, uint64_t *counters);
extern "C" size_t lda_selfcheck_region_bytes(void);

/* CRC constant tables, generated on the host at first use (host_context.hip) */
#define LDA_CRC_TABLE_WORDS (17 * 256)
#define LDA_CRC_XPOW_WORDS 1024

#endif /* LDA_KERNELS_H */
