/*
 * host_compress.hip - C-ABI of the compressor: object lifetime, level table,
 * compress_bound, the device batch entry point, the host-pointer batch and
 * the three single-buffer libdeflate_*_compress calls as batches of one.
 *
 * Reference interfaces replaced: libdeflate.h:59-67 (alloc), :85-152
 * (compress + bounds), :159-160 (free).
 */
#include <stdio.h>
#include <stdlib.h>
#include <new>
#include <vector>

#include "host_objects.h"
#include "kernels.h"

namespace lda {
bool pick_allocator(const struct libdeflate_options *options,
		    malloc_func_t *m, free_func_t *f);
}
using namespace lda;

/*
 * level -> (search depth, nice length, parse mode), the policy table of
 * lib/deflate_compress.c:3927-3979.  Level 1 maps onto the same hash-chain
 * kernel with a 2-deep search (the reference's level 1 probes a 2-way
 * bucket, lib/ht_matchfinder.h:50-55).  Levels 10-12 (near-optimal parsing,
 * SURVEY.md §8(f) "next") currently run the deepest lazy2 search.
 */
struct level_cfg { uint32_t depth, nice, mode; };
static const level_cfg k_levels[13] = {
	{ 0, 0, 0 },		/* 0: stored */
	{ 2, 32, 0 },		/* 1 */
	{ 6, 10, 0 },		/* 2: greedy */
	{ 12, 14, 0 },
	{ 16, 30, 0 },
	{ 16, 30, 1 },		/* 5: lazy */
	{ 35, 65, 1 },
	{ 100, 130, 1 },
	{ 300, 258, 2 },	/* 8: lazy2 */
	{ 600, 258, 2 },
	{ 1000, 258, 2 },	/* 10-12: see above */
	{ 1500, 258, 2 },
	{ 2000, 258, 2 },
};

extern "C" LIBDEFLATEAPI struct libdeflate_compressor *
libdeflate_alloc_compressor_ex(int level, const struct libdeflate_options *options)
{
	malloc_func_t m;
	free_func_t f;

	if (!pick_allocator(options, &m, &f))
		return NULL;
	if (level == -1)	/* libdeflate.h:47 */
		level = 6;
	if (level < 0 || level > 12)
		return NULL;
	if (!device_ctx()) {
		fprintf(stderr, "libdeflate_amd: alloc_compressor: no usable "
			"gfx950 device (%s); no CPU fallback\n",
			libdeflate_amd_last_error());
		return NULL;
	}
	void *mem = m(sizeof(struct libdeflate_compressor));
	if (!mem)
		return NULL;
	struct libdeflate_compressor *c = new (mem) libdeflate_compressor();
	c->free_func = f;
	c->level = level;
	return c;
}

extern "C" LIBDEFLATEAPI struct libdeflate_compressor *
libdeflate_alloc_compressor(int level)
{
	return libdeflate_alloc_compressor_ex(level, NULL);
}

extern "C" LIBDEFLATEAPI void
libdeflate_free_compressor(struct libdeflate_compressor *c)
{
	if (!c)
		return;
	c->scratch.release();
	c->stage.release();
	free_func_t f = c->free_func;
	c->~libdeflate_compressor();
	f(c);
}

/* lib/deflate_compress.c:4087-4135: 5 bytes per 5000-byte worst-case block */
extern "C" LIBDEFLATEAPI size_t
libdeflate_deflate_compress_bound(struct libdeflate_compressor *c, size_t n)
{
	(void)c;
	size_t blocks = (n + 4999) / 5000;
	if (blocks < 1)
		blocks = 1;
	return 5 * blocks + n;
}

/* lib/zlib_compress.c:76-82 */
extern "C" LIBDEFLATEAPI size_t
libdeflate_zlib_compress_bound(struct libdeflate_compressor *c, size_t n)
{
	return 6 + libdeflate_deflate_compress_bound(c, n);
}

/* lib/gzip_compress.c:84-90 */
extern "C" LIBDEFLATEAPI size_t
libdeflate_gzip_compress_bound(struct libdeflate_compressor *c, size_t n)
{
	return 18 + libdeflate_deflate_compress_bound(c, n);
}

extern "C" LIBDEFLATEAPI int
libdeflate_amd_compress_batch(struct libdeflate_compressor *c, int format,
			      size_t n, const void *d_in,
			      const uint64_t *d_in_offsets,
			      const uint64_t *d_in_nbytes, void *d_out,
			      const uint64_t *d_out_offsets,
			      const uint64_t *d_out_avail,
			      uint64_t *d_out_nbytes, void *stream)
{
	DeviceCtx *ctx = device_ctx();
	hipStream_t st = (hipStream_t)stream;

	if (!ctx)
		return LIBDEFLATE_AMD_NO_DEVICE;
	if (n == 0)
		return LIBDEFLATE_AMD_OK;
	if (!c || !d_in || !d_in_offsets || !d_in_nbytes || !d_out ||
	    !d_out_offsets || !d_out_avail || !d_out_nbytes ||
	    format < LIBDEFLATE_AMD_DEFLATE || format > LIBDEFLATE_AMD_GZIP) {
		set_error("compress_batch: bad argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	/* scratch: [match lists: u64 x words x grid][sums u32 x n] */
	size_t grid = n < (size_t)ctx->num_cus ? n : (size_t)ctx->num_cus;
	size_t seq_bytes = grid * lda_deflate_seq_words() * 8;
	uint8_t *scr = (uint8_t *)c->scratch.reserve(seq_bytes + n * 4);
	if (!scr)
		return LIBDEFLATE_AMD_OOM;
	uint32_t *sums = NULL;
	if (format != LIBDEFLATE_AMD_DEFLATE) {
		sums = (uint32_t *)(scr + seq_bytes);
		int rc = format == LIBDEFLATE_AMD_GZIP ?
			libdeflate_amd_crc32_batch(n, d_in, d_in_offsets,
						   d_in_nbytes, NULL, sums, stream) :
			libdeflate_amd_adler32_batch(n, d_in, d_in_offsets,
						     d_in_nbytes, NULL, sums, stream);
		if (rc != LIBDEFLATE_AMD_OK)
			return rc;
	}
	static bool attr_set[16];
	size_t lds = lda_deflate_lds_bytes();
	if (!attr_set[ctx->device]) {
		LDA_HIP_TRY(hipFuncSetAttribute(
				(const void *)lda_deflate_batch_kernel,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)lds), LIBDEFLATE_AMD_NO_DEVICE);
		attr_set[ctx->device] = true;
	}
	const level_cfg &lv = k_levels[c->level];
	hipLaunchKernelGGL(lda_deflate_batch_kernel, dim3((unsigned)grid),
			   dim3(LDA_DEFLATE_THREADS), lds, st, (uint64_t)n, format, c->level,
			   lv.depth, lv.nice, lv.mode, (const uint8_t *)d_in,
			   d_in_offsets, d_in_nbytes, (uint8_t *)d_out,
			   d_out_offsets, d_out_avail, d_out_nbytes, sums,
			   (uint64_t *)scr);
	LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	return LIBDEFLATE_AMD_OK;
}

extern "C" LIBDEFLATEAPI int
libdeflate_amd_compress_batch_host(struct libdeflate_compressor *c, int format,
				   size_t n, const void *const *in,
				   const size_t *in_nbytes, void *const *out,
				   const size_t *out_avail, size_t *out_nbytes)
{
	if (!device_ctx())
		return LIBDEFLATE_AMD_NO_DEVICE;
	if (n == 0)
		return LIBDEFLATE_AMD_OK;
	if (!c || !in || !in_nbytes || !out || !out_avail || !out_nbytes) {
		set_error("compress_batch_host: NULL argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	std::vector<uint64_t> desc(5 * n);
	uint64_t *in_off = &desc[0], *in_n = &desc[n], *out_off = &desc[2 * n],
		 *out_av = &desc[3 * n];
	size_t pos = align_up(5 * n * 8, 64);
	for (size_t i = 0; i < n; i++) {
		in_off[i] = pos;
		in_n[i] = in_nbytes[i];
		pos = align_up(pos + in_nbytes[i] + 16, 16);
	}
	for (size_t i = 0; i < n; i++) {
		out_off[i] = pos;
		out_av[i] = out_avail[i];
		pos = align_up(pos + out_avail[i] + 16, 16);
	}
	uint8_t *st = (uint8_t *)c->stage.reserve(pos + 64);
	if (!st)
		return LIBDEFLATE_AMD_OOM;
	LDA_HIP_TRY(hipMemcpy(st, desc.data(), 4 * n * 8, hipMemcpyHostToDevice),
		    LIBDEFLATE_AMD_NO_DEVICE);
	for (size_t i = 0; i < n; i++)
		if (in_nbytes[i])
			LDA_HIP_TRY(hipMemcpy(st + in_off[i], in[i], in_nbytes[i],
					      hipMemcpyHostToDevice),
				    LIBDEFLATE_AMD_NO_DEVICE);
	uint64_t *d_desc = (uint64_t *)st;
	int rc = libdeflate_amd_compress_batch(c, format, n, st, d_desc,
					       d_desc + n, st, d_desc + 2 * n,
					       d_desc + 3 * n, d_desc + 4 * n,
					       NULL);
	if (rc != LIBDEFLATE_AMD_OK)
		return rc;
	LDA_HIP_TRY(hipDeviceSynchronize(), LIBDEFLATE_AMD_NO_DEVICE);
	LDA_HIP_TRY(hipMemcpy(&desc[4 * n], d_desc + 4 * n, n * 8,
			      hipMemcpyDeviceToHost), LIBDEFLATE_AMD_NO_DEVICE);
	for (size_t i = 0; i < n; i++) {
		out_nbytes[i] = desc[4 * n + i];
		if (out_nbytes[i])
			LDA_HIP_TRY(hipMemcpy(out[i], st + out_off[i],
					      out_nbytes[i],
					      hipMemcpyDeviceToHost),
				    LIBDEFLATE_AMD_NO_DEVICE);
	}
	return LIBDEFLATE_AMD_OK;
}

static size_t compress_one(struct libdeflate_compressor *c, int format,
			   const void *in, size_t in_nbytes, void *out,
			   size_t out_avail)
{
	const void *ins[1] = { in };
	void *outs[1] = { out };
	size_t got = 0;
	int rc = libdeflate_amd_compress_batch_host(c, format, 1, ins, &in_nbytes,
						    outs, &out_avail, &got);

	if (rc != LIBDEFLATE_AMD_OK)
		die_no_device("libdeflate_*_compress");
	return got;
}

/* libdeflate.h:85-88 */
extern "C" LIBDEFLATEAPI size_t
libdeflate_deflate_compress(struct libdeflate_compressor *c, const void *in,
			    size_t in_nbytes, void *out, size_t out_avail)
{
	return compress_one(c, LIBDEFLATE_AMD_DEFLATE, in, in_nbytes, out,
			    out_avail);
}

/* libdeflate.h:122-125 */
extern "C" LIBDEFLATEAPI size_t
libdeflate_zlib_compress(struct libdeflate_compressor *c, const void *in,
			 size_t in_nbytes, void *out, size_t out_avail)
{
	return compress_one(c, LIBDEFLATE_AMD_ZLIB, in, in_nbytes, out, out_avail);
}

/* libdeflate.h:140-143 */
extern "C" LIBDEFLATEAPI size_t
libdeflate_gzip_compress(struct libdeflate_compressor *c, const void *in,
			 size_t in_nbytes, void *out, size_t out_avail)
{
	return compress_one(c, LIBDEFLATE_AMD_GZIP, in, in_nbytes, out, out_avail);
}
