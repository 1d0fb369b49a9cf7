/*
 * host_checksum.hip - C-ABI for CRC-32 / Adler-32 (single buffer and batch).
 * Reference interface replaced: libdeflate.h:335-336, :345-346.
 */
#include "host_common.h"
#include "kernels.h"

using namespace lda;

static unsigned checksum_grid(const DeviceCtx *c, size_t n_chunks)
{
	size_t blocks = (n_chunks + 3) / 4;	/* 4 waves (chunks) per block */
	size_t cap = (size_t)c->num_cus * 8;

	if (blocks > cap)
		blocks = cap;
	return blocks ? (unsigned)blocks : 1u;
}

extern "C" LIBDEFLATEAPI int
libdeflate_amd_crc32_batch(size_t n_chunks, const void *d_in,
			   const uint64_t *d_offsets, const uint64_t *d_nbytes,
			   const uint32_t *d_init, uint32_t *d_out, void *stream)
{
	DeviceCtx *c = device_ctx();

	if (!c)
		return LIBDEFLATE_AMD_NO_DEVICE;
	if (n_chunks == 0)
		return LIBDEFLATE_AMD_OK;
	if (!d_in || !d_offsets || !d_nbytes || !d_out) {
		set_error("crc32_batch: NULL argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	hipLaunchKernelGGL(lda_crc32_batch_kernel, dim3(checksum_grid(c, n_chunks)),
			   dim3(256), 0, (hipStream_t)stream, (uint64_t)n_chunks,
			   (const uint8_t *)d_in, d_offsets, d_nbytes, d_init,
			   d_out, c->d_crc_tables, c->d_crc_xpow8);
	LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	return LIBDEFLATE_AMD_OK;
}

extern "C" LIBDEFLATEAPI int
libdeflate_amd_adler32_batch(size_t n_chunks, const void *d_in,
			     const uint64_t *d_offsets,
			     const uint64_t *d_nbytes, const uint32_t *d_init,
			     uint32_t *d_out, void *stream)
{
	DeviceCtx *c = device_ctx();

	if (!c)
		return LIBDEFLATE_AMD_NO_DEVICE;
	if (n_chunks == 0)
		return LIBDEFLATE_AMD_OK;
	if (!d_in || !d_offsets || !d_nbytes || !d_out) {
		set_error("adler32_batch: NULL argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	hipLaunchKernelGGL(lda_adler32_batch_kernel,
			   dim3(checksum_grid(c, n_chunks)), dim3(256), 0,
			   (hipStream_t)stream, (uint64_t)n_chunks,
			   (const uint8_t *)d_in, d_offsets, d_nbytes, d_init,
			   d_out);
	LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	return LIBDEFLATE_AMD_OK;
}

/* one host buffer -> batch of one through the staging area */
static uint32_t checksum_host(bool crc, uint32_t init, const void *buf,
			      size_t len)
{
	DeviceCtx *c = device_ctx();

	if (!c)
		die(crc ? "libdeflate_crc32" : "libdeflate_adler32");
	std::lock_guard<std::mutex> lk(c->stage_mu);
	/* layout: [offset u64][nbytes u64][init u32][out u32][pad][data] */
	size_t hdr = 64;
	uint8_t *st = (uint8_t *)stage_reserve(c, hdr + len + 16);
	if (!st)
		die("checksum staging");
	struct { uint64_t off, n; uint32_t init, out; } h = { hdr, len, init, 0 };
	if (hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess ||
	    (len && hipMemcpy(st + hdr, buf, len, hipMemcpyHostToDevice) !=
			    hipSuccess))
		die("checksum H2D copy");
	int rc = crc ?
		libdeflate_amd_crc32_batch(1, st, (uint64_t *)st,
					   (uint64_t *)(st + 8),
					   (uint32_t *)(st + 16),
					   (uint32_t *)(st + 20), nullptr) :
		libdeflate_amd_adler32_batch(1, st, (uint64_t *)st,
					     (uint64_t *)(st + 8),
					     (uint32_t *)(st + 16),
					     (uint32_t *)(st + 20), nullptr);
	uint32_t out = 0;
	if (rc != LIBDEFLATE_AMD_OK ||
	    hipMemcpy(&out, st + 20, 4, hipMemcpyDeviceToHost) != hipSuccess)
		die("checksum kernel");
	return out;
}

/* libdeflate.h:345-346 / lib/crc32.c:256-262 */
extern "C" LIBDEFLATEAPI uint32_t
libdeflate_crc32(uint32_t crc, const void *buffer, size_t len)
{
	if (buffer == NULL)	/* "Return initial value." */
		return 0;
	return checksum_host(true, crc, buffer, len);
}

/* libdeflate.h:335-336 / lib/adler32.c:156-162 */
extern "C" LIBDEFLATEAPI uint32_t
libdeflate_adler32(uint32_t adler, const void *buffer, size_t len)
{
	if (buffer == NULL)
		return 1;
	return checksum_host(false, adler, buffer, len);
}
