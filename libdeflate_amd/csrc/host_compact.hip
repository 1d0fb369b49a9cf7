/*
 * host_compact.hip - C-ABI of the device-side output compaction
 * (compact_kernels.hip): prefix sum of the per-chunk sizes + one gather copy.
 */
#include "host_common.h"
#include "kernels.h"

using namespace lda;

#define LDA_SCAN_BLOCK 2048	/* compact_kernels.hip: SCAN_BLOCK */

extern "C" LIBDEFLATEAPI size_t
libdeflate_amd_compact_offsets_len(size_t n_chunks)
{
	return n_chunks + 2 + (n_chunks + LDA_SCAN_BLOCK - 1) / LDA_SCAN_BLOCK;
}

extern "C" LIBDEFLATEAPI int
libdeflate_amd_compact_batch(size_t n, const void *d_in,
			     const uint64_t *d_in_offsets,
			     const uint64_t *d_nbytes, void *d_out,
			     uint64_t *d_out_offsets, void *stream)
{
	DeviceCtx *c = device_ctx();
	hipStream_t st = (hipStream_t)stream;

	if (!c)
		return LIBDEFLATE_AMD_NO_DEVICE;
	if (!d_out_offsets || (n && (!d_in || !d_in_offsets || !d_nbytes || !d_out))) {
		set_error("compact_batch: NULL argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	if (n == 0) {
		LDA_HIP_TRY(hipMemsetAsync(d_out_offsets, 0, 8, st), LIBDEFLATE_AMD_NO_DEVICE);
		return LIBDEFLATE_AMD_OK;
	}
	const size_t nblocks = (n + LDA_SCAN_BLOCK - 1) / LDA_SCAN_BLOCK;
	uint64_t *block_sums = d_out_offsets + n + 1;	/* nblocks + 1 entries */
	hipLaunchKernelGGL(lda_scan_local_kernel, dim3((unsigned)nblocks), dim3(256),
			   0, st, (uint64_t)n, d_nbytes, d_out_offsets, block_sums);
	hipLaunchKernelGGL(lda_scan_blocks_kernel, dim3(1), dim3(1024), 0, st,
			   (uint64_t)nblocks, block_sums);
	size_t grid = (size_t)c->num_cus * 8;
	if (grid > n)
		grid = n;
	hipLaunchKernelGGL(lda_compact_copy_kernel, dim3((unsigned)grid), dim3(256),
			   0, st, (uint64_t)n, (const uint8_t *)d_in, d_in_offsets,
			   d_nbytes, (uint8_t *)d_out, d_out_offsets, block_sums);
	LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	return LIBDEFLATE_AMD_OK;
}
