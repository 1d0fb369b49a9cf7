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
#include <string.h>
#include <atomic>
#include <new>
#include <algorithm>
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
 * bucket, lib/ht_matchfinder.h:50-55).  Levels 10-12 (mode 3) choose the
 * tokens by a min-cost parse over the chain search's results instead of the
 * lazy rule (deflate_kernel.hip, "min-cost parse"); the reference's binary
 * tree match finder (lib/bt_matchfinder.h) has no counterpart.
 */
struct level_cfg { uint32_t depth, nice, mode; };	/* mode: 0 greedy, 1 lazy, 2 lazy2, 3 min-cost */
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
	/* 10-12: min-cost parse over every position's deepest match.  The chains
	 * of a 64 KiB buffer are exhausted at ~600 steps (13 hash bits, 24 K
	 * window: 600, 1000 and 2000 measured the same time and, to 0.02 %, the
	 * same bytes - round 5's 600 / 1000 / 2000 were one level three times),
	 * so the ladder is 150 / 300 / all of the chain: 38.4 / 45.6 / 53.0 ms
	 * per 4096 x 64 KiB at 1.0062 / 1.0082 / 1.0074 x the reference's size
	 * at the same level (the mix of tests/datagen.py; round 6) */
	{ 150, 258, 3 },
	{ 300, 258, 3 },
	{ 2000, 258, 3 },
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
	c->malloc_func = m;
	c->level = level;
	c->device = 0;
	(void)hipGetDevice(&c->device);	/* (device_ctx() above has seen it work) */
	for (int k = 0; k < LDA_MAX_SHARDS; k++)
		c->shard[k] = NULL;
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
	for (int k = 0; k < LDA_MAX_SHARDS; k++)
		libdeflate_free_compressor(c->shard[k]);
	DeviceGuard on(c->device);
	c->scratch.release();
	c->stage.release();
	c->pinned.release();
	c->meta.release();
	c->streams.release();
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

static int
compress_batch_impl(struct libdeflate_compressor *c, int format, size_t n,
		    const void *d_in, const uint64_t *d_in_offsets,
		    const uint64_t *d_in_nbytes, void *d_out,
		    const uint64_t *d_out_offsets, const uint64_t *d_out_avail,
		    uint64_t *d_out_nbytes, void *stream,
		    const uint32_t *d_seg_info, size_t max_in_nbytes = SIZE_MAX)
{
	if (!c) {
		set_error("compress_batch: bad argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	DeviceGuard on(c->device);
	if (!on.ok())
		return LIBDEFLATE_AMD_NO_DEVICE;
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
	/* buffers of at most 4 KiB (filesystem blocks): the 256-thread kernel,
	 * several workgroups per CU (deflate_small.hip); levels 10-12 keep the
	 * big one (their parse wants its LDS) */
	const bool small = max_in_nbytes <= lda_deflate_small_max() &&
			   c->level <= 9 && !d_seg_info && !env_cfg().no_small;
	/* scratch: [token lists: u64 x words x grid][chunk counter][sums u32 x n] */
	size_t grid_max = (size_t)ctx->num_cus * (small ? lda_deflate_small_wgs() : 1);
	size_t grid = n < grid_max ? n : grid_max;
	size_t seq_bytes = grid * lda_deflate_seq_words() * 8;
	uint8_t *scr = (uint8_t *)c->scratch.reserve(seq_bytes + 16 + n * 4);
	if (!scr)
		return LIBDEFLATE_AMD_OOM;
	uint32_t *next_chunk = (uint32_t *)(scr + seq_bytes);
	LDA_HIP_TRY(hipMemsetAsync(next_chunk, 0, 16, st), LIBDEFLATE_AMD_NO_DEVICE);
	uint32_t *sums = NULL;
	if (format != LIBDEFLATE_AMD_DEFLATE) {
		sums = (uint32_t *)(scr + seq_bytes + 16);
		int rc = format == LIBDEFLATE_AMD_GZIP ?
			libdeflate_amd_crc32_batch(n, d_in, d_in_offsets,
						   d_in_nbytes, NULL, sums, stream) :
			libdeflate_amd_adler32_batch(n, d_in, d_in_offsets,
						     d_in_nbytes, NULL, sums, stream);
		if (rc != LIBDEFLATE_AMD_OK)
			return rc;
	}
	const size_t lds = small ? lda_deflate_small_lds_bytes() : lda_deflate_lds_bytes();
	if (!ctx->deflate_attr_set.load(std::memory_order_acquire)) {
		LDA_HIP_TRY(hipFuncSetAttribute(
				(const void *)lda_deflate_small_kernel,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)lda_deflate_small_lds_bytes()), LIBDEFLATE_AMD_NO_DEVICE);
		LDA_HIP_TRY(hipFuncSetAttribute(
				(const void *)lda_deflate_batch_kernel,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)lda_deflate_lds_bytes()), LIBDEFLATE_AMD_NO_DEVICE);
		LDA_HIP_TRY(hipFuncSetAttribute(
				(const void *)lda_deflate_opt_kernel,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)lda_deflate_lds_bytes()), LIBDEFLATE_AMD_NO_DEVICE);
		ctx->deflate_attr_set.store(true, std::memory_order_release);
	}
	const level_cfg &lv = k_levels[c->level];
	hipLaunchKernelGGL(small ? lda_deflate_small_kernel :
			   lv.mode == 3 ? lda_deflate_opt_kernel :
					  lda_deflate_batch_kernel, dim3((unsigned)grid),
			   dim3(small ? LDA_DEFLATE_SMALL_THREADS : LDA_DEFLATE_THREADS),
			   lds, st, (uint64_t)n, format, c->level,
			   lv.depth, lv.nice, lv.mode, (const uint8_t *)d_in,
			   d_in_offsets, d_in_nbytes, (uint8_t *)d_out,
			   d_out_offsets, d_out_avail, d_out_nbytes, sums,
			   (uint64_t *)scr, d_seg_info, next_chunk);
	LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	return LIBDEFLATE_AMD_OK;
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
	return compress_batch_impl(c, format, n, d_in, d_in_offsets, d_in_nbytes,
				   d_out, d_out_offsets, d_out_avail,
				   d_out_nbytes, stream, NULL);
}

extern "C" LIBDEFLATEAPI int
libdeflate_amd_compress_batch_bounded(struct libdeflate_compressor *c, int format,
				      size_t n, const void *d_in,
				      const uint64_t *d_in_offsets,
				      const uint64_t *d_in_nbytes, void *d_out,
				      const uint64_t *d_out_offsets,
				      const uint64_t *d_out_avail,
				      uint64_t *d_out_nbytes, size_t max_in_nbytes,
				      void *stream)
{
	return compress_batch_impl(c, format, n, d_in, d_in_offsets, d_in_nbytes,
				   d_out, d_out_offsets, d_out_avail,
				   d_out_nbytes, stream, NULL, max_in_nbytes);
}

static int compress_batch_host_body(struct libdeflate_compressor *c, int format,
				   size_t n, const void *const *in,
				   const size_t *in_nbytes, void *const *out,
				   const size_t *out_avail, size_t *out_nbytes);

extern "C" LIBDEFLATEAPI int
libdeflate_amd_compress_batch_host(struct libdeflate_compressor *c, int format,
				   size_t n, const void *const *in,
				   const size_t *in_nbytes, void *const *out,
				   const size_t *out_avail, size_t *out_nbytes)
{
	return no_unwind("compress_batch_host", (int)LIBDEFLATE_AMD_OOM, [&]() {
		if (!c || !in_nbytes || n == 0)
			return compress_batch_host_body(c, format, n, in, in_nbytes, out, out_avail,
							out_nbytes);
		/* several GPUs (LDA_DEVICES, host_fanout.hip): contiguous shards,
		 * an object and a host thread per device, results in place */
		size_t bounds[LDA_MAX_SHARDS + 1];
		int devs[LDA_MAX_SHARDS];
		const size_t shards = fanout_plan(c->device, n, in_nbytes, bounds, devs);
		fanout_note(shards);
		if (shards < 2)
			return compress_batch_host_body(c, format, n, in, in_nbytes, out, out_avail,
							out_nbytes);
		if (!in || !out || !out_avail || !out_nbytes) {
			set_error("compress_batch_host: NULL argument");
			return (int)LIBDEFLATE_AMD_BAD_ARG;
		}
		for (size_t k = 1; k < shards; k++) {
			if (c->shard[k])
				continue;
			DeviceGuard on(devs[k]);
			struct libdeflate_options o = {};
			o.sizeof_options = sizeof(o);
			o.malloc_func = c->malloc_func;
			o.free_func = c->free_func;
			if (on.ok())
				c->shard[k] = libdeflate_alloc_compressor_ex(c->level, &o);
			if (!c->shard[k]) {
				/* a device that cannot take its shard (out of memory,
				 * refused by the self-check, busy): the batch stays on
				 * the object's own device rather than fail - the reason
				 * stays in libdeflate_amd_last_error() */
				fanout_note(1);
				return compress_batch_host_body(c, format, n, in, in_nbytes, out, out_avail,
							out_nbytes);
			}
		}
		return fanout_run(shards, [&](size_t k) {
			const size_t lo = bounds[k], cnt = bounds[k + 1] - lo;
			return compress_batch_host_body(k ? c->shard[k] : c, format, cnt, in + lo,
							in_nbytes + lo, out + lo, out_avail + lo,
							out_nbytes + lo);
		});
	});
}

static int compress_batch_host_body(struct libdeflate_compressor *c, int format,
				   size_t n, const void *const *in,
				   const size_t *in_nbytes, void *const *out,
				   const size_t *out_avail, size_t *out_nbytes)
{
	if (n == 0)
		return device_ctx() ? LIBDEFLATE_AMD_OK : LIBDEFLATE_AMD_NO_DEVICE;
	if (!c || !in || !in_nbytes || !out || !out_avail || !out_nbytes) {
		set_error("compress_batch_host: NULL argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	DeviceGuard on(c->device);
	if (!on.ok() || !device_ctx())
		return LIBDEFLATE_AMD_NO_DEVICE;
	/* The batch goes through in SLICES (up to 8, >= 64 MiB of input each - a
	 * slice has to fill the GPU several times over, or its kernel's tail
	 * costs more than the overlap gives: 8 MiB slices measured slower than
	 * no slices): the
	 * kernels of slice k run on the object's compute stream while the host
	 * packs and sends slice k + 1 and unpacks slice k - 1 on its copy stream.
	 * device layout: [in_off in_n out_off out_av out_n][cmp_off of every
	 * slice] [inputs] [output slots] [compacted outputs, slice after slice];
	 * everything 16-byte aligned */
	enum { MAX_SLICES = 8 };
	size_t bounds[MAX_SLICES + 1];
	const size_t ns = slice_by_bytes(n, in_nbytes, MAX_SLICES, (size_t)64 << 20, bounds);
	size_t cmp_len[MAX_SLICES], cmp_pos[MAX_SLICES], ncmp = 0;
	for (size_t k = 0; k < ns; k++) {
		cmp_pos[k] = ncmp;
		cmp_len[k] = libdeflate_amd_compact_offsets_len(bounds[k + 1] - bounds[k]);
		ncmp += cmp_len[k];
	}
	std::vector<uint64_t> desc(5 * n + ncmp);
	uint64_t *in_off = &desc[0], *in_n = &desc[n], *out_off = &desc[2 * n],
		 *out_av = &desc[3 * n], *out_n = &desc[4 * n], *cmp_off = &desc[5 * n];
	size_t pos = align_up(desc.size() * 8, 64);
	for (size_t i = 0; i < n; i++) {
		in_off[i] = pos;
		in_n[i] = in_nbytes[i];
		pos = align_up(pos + in_nbytes[i] + 16, 16);
	}
	size_t total_avail = 0;
	size_t avail_before[MAX_SLICES + 1];
	for (size_t k = 0, i = 0; k < ns; k++) {
		avail_before[k] = total_avail;
		for (; i < bounds[k + 1]; i++) {
			out_off[i] = pos;
			out_av[i] = out_avail[i];
			pos = align_up(pos + out_avail[i] + 16, 16);
			total_avail += out_avail[i];
		}
	}
	const size_t cmp_at = pos;
	uint8_t *st = (uint8_t *)c->stage.reserve(cmp_at + total_avail + 64 * ns + 64);
	if (!st)
		return LIBDEFLATE_AMD_OOM;
	if (!c->streams.ensure())
		return LIBDEFLATE_AMD_NO_DEVICE;
	size_t max_in = 0;
	for (size_t i = 0; i < n; i++)
		max_in = in_nbytes[i] > max_in ? in_nbytes[i] : max_in;
	/* the kernels' scratch for the largest launch up front: growing it between
	 * two slices would free memory a running kernel uses (and synchronise).
	 * Sized by the launches that are made (compress_batch_impl(): one token
	 * list per workgroup, the grid never larger than the slice) - a one-shot
	 * call on a small buffer reserves one list, not a device's worth */
	{
		size_t max_nk = 0;
		for (size_t k = 0; k < ns; k++)
			max_nk = bounds[k + 1] - bounds[k] > max_nk ? bounds[k + 1] - bounds[k] : max_nk;
		const bool small = max_in <= lda_deflate_small_max() && c->level <= 9 &&
				   !env_cfg().no_small;
		const size_t grid_max = (size_t)device_ctx()->num_cus * (small ? lda_deflate_small_wgs() : 1);
		const size_t grid = max_nk < grid_max ? max_nk : grid_max;
		if (!c->scratch.reserve(grid * lda_deflate_seq_words() * 8 + 16 + max_nk * 4))
			return LIBDEFLATE_AMD_OOM;
	}
	/* what comes back per slice (sizes, compaction offsets) lands in pinned
	 * memory, so the copies are asynchronous and the host is free to pack the
	 * next slice while this one's kernels run */
	uint64_t *h_back = (uint64_t *)c->meta.ensure((n + ncmp) * 8);
	if (!h_back)
		return LIBDEFLATE_AMD_OOM;
	out_n = h_back;
	cmp_off = h_back + n;
	hipStream_t s_copy = c->streams.copy, s_comp = c->streams.comp;
	LDA_HIP_TRY(hipMemcpyAsync(st, desc.data(), 4 * n * 8, hipMemcpyHostToDevice,
				   s_copy), LIBDEFLATE_AMD_NO_DEVICE);
	uint64_t *d_desc = (uint64_t *)st;
	hipEvent_t ev_done[MAX_SLICES] = {};
	int rc = LIBDEFLATE_AMD_OK;
	auto cleanup = [&]() {
		(void)hipStreamSynchronize(s_comp);
		(void)hipStreamSynchronize(s_copy);
		for (size_t k = 0; k < ns; k++)
			if (ev_done[k])
				(void)hipEventDestroy(ev_done[k]);
	};
	/* slice k's streams are complete on the device: sizes to the caller,
	 * bytes through the pinned pair */
	auto drain = [&](size_t k) -> int {
		const size_t lo = bounds[k], nk = bounds[k + 1] - lo;
		LDA_HIP_TRY(hipEventSynchronize(ev_done[k]), LIBDEFLATE_AMD_NO_DEVICE);
		for (size_t i = lo; i < lo + nk; i++)
			out_nbytes[i] = out_n[i];
		return copy_out_packed(&c->pinned, st + cmp_at + avail_before[k] + 64 * k, nk,
				       out + lo, out_n + lo, cmp_off + cmp_pos[k], s_copy);
	};
	for (size_t k = 0; k < ns && rc == LIBDEFLATE_AMD_OK; k++) {
		const size_t lo = bounds[k], nk = bounds[k + 1] - lo;
		/* (returns when the slice is on the device) */
		rc = copy_in_packed(&c->pinned, st, nk, in + lo, in_nbytes + lo, in_off + lo, s_copy);
		if (rc != LIBDEFLATE_AMD_OK)
			break;
		rc = libdeflate_amd_compress_batch_bounded(
			c, format, nk, st, d_desc + lo, d_desc + n + lo, st, d_desc + 2 * n + lo,
			d_desc + 3 * n + lo, d_desc + 4 * n + lo, max_in, s_comp);
		if (rc != LIBDEFLATE_AMD_OK)
			break;
		rc = libdeflate_amd_compact_batch(nk, st, d_desc + 2 * n + lo, d_desc + 4 * n + lo,
						  st + cmp_at + avail_before[k] + 64 * k,
						  d_desc + 5 * n + cmp_pos[k], s_comp);
		if (rc != LIBDEFLATE_AMD_OK)
			break;
		if (hipMemcpyAsync(out_n + lo, d_desc + 4 * n + lo, nk * 8, hipMemcpyDeviceToHost,
				   s_comp) != hipSuccess ||
		    hipMemcpyAsync(cmp_off + cmp_pos[k], d_desc + 5 * n + cmp_pos[k], (nk + 1) * 8,
				   hipMemcpyDeviceToHost, s_comp) != hipSuccess ||
		    hipEventCreateWithFlags(&ev_done[k], hipEventDisableTiming) != hipSuccess ||
		    hipEventRecord(ev_done[k], s_comp) != hipSuccess) {
			set_error("compress_batch_host: %s", hipGetErrorString(hipGetLastError()));
			rc = LIBDEFLATE_AMD_NO_DEVICE;
			break;
		}
		if (k)
			rc = drain(k - 1);
	}
	if (rc == LIBDEFLATE_AMD_OK)
		rc = drain(ns - 1);
	cleanup();
	return rc;
}

/*
 * One LARGE buffer: the input is cut into sub-ranges
 * of LDA_SEG_BYTES that are compressed side by side, one workgroup each.
 * Every sub-range but the first is given the tail of its predecessor as a
 * dictionary (whole tiles that only prime the hash chains), every one but
 * the last ends byte aligned (non-final block + empty stored block), so the
 * pieces concatenate into ONE raw DEFLATE stream; the container header /
 * footer are written on the host, the checksum is combined from the
 * per-piece checksums:
 *   crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B)
 * (programs/gzip.c:149-185 hands the whole file to one compress call; this
 * is what makes that call use the whole GPU.)
 */
#define LDA_SEG_BYTES 65536u
#define LDA_LARGE_MIN (2 * LDA_SEG_BYTES)

/* multiply two reflected polynomials mod the CRC-32 polynomial (bit 31 = x^0) */
static uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
	for (uint32_t m = 0x80000000u; m; m >>= 1) {
		if (a & m)
			p ^= b;
		b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
	}
	return p;
}

/* x^(8 len) mod P: what appending len bytes multiplies a CRC by */
uint32_t lda::crc32_shift(uint64_t len)
{
	uint32_t xp = 0x80000000u;	/* 1 */
	uint32_t base = 0x00800000u;	/* x^8 */
	for (uint64_t k = len; k; k >>= 1) {
		if (k & 1)
			xp = crc_mulmod(xp, base);
		base = crc_mulmod(base, base);
	}
	return xp;
}

uint32_t lda::crc32_concat_shift(uint32_t crc_a, uint32_t crc_b, uint32_t shift_b)
{
	return crc_mulmod(crc_a, shift_b) ^ crc_b;
}

uint32_t lda::crc32_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b)
{
	return crc32_concat_shift(crc_a, crc_b, crc32_shift(len_b));
}

uint32_t lda::adler32_concat(uint32_t ad_a, uint32_t ad_b, uint64_t len_b)
{
	const uint32_t M = 65521;
	uint32_t a1 = ad_a & 0xFFFF, b1 = ad_a >> 16;
	uint32_t a2 = ad_b & 0xFFFF, b2 = ad_b >> 16;
	uint32_t rem = (uint32_t)(len_b % M);
	uint32_t a = (a1 + a2 + M - 1) % M;
	uint32_t b = (uint32_t)(((uint64_t)rem * a1 + b1 + b2 + M - rem) % M);
	return (b << 16) | a;
}

/* a device failure inside a single-buffer call: reported, and the call returns
 * 0 like any other "could not produce the stream" (libdeflate.h:73-74); the
 * reason is in libdeflate_amd_last_error() */
static size_t large_fail(const char *what)
{
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		set_error("%s: %s", what, hipGetErrorString(e));
	complain("libdeflate_*_compress", LIBDEFLATE_AMD_NO_DEVICE);
	return 0;
}

/*
 * The segments go through in SLICES of up to 32 MiB of input on the object's
 * two streams, like the host-pointer batches: while the kernels of slice k
 * (deflate of its segments, their checksums, the compaction of their streams)
 * run on the compute stream, the host packs slice k + 1 into the pinned
 * staging and sends it, and brings the compacted streams of slice k - 1 back.
 * Nothing runs on the null stream and nothing waits for the whole device.
 */
static size_t compress_large(struct libdeflate_compressor *c, int format,
			     const uint8_t *in, size_t n, uint8_t *out,
			     size_t out_avail)
{
	/* sub-ranges of 64 KiB; an input that would not fill the CUs with those
	 * is cut finer (more blocks and sync markers: ~1 % larger at 16 KiB) */
	size_t S = LDA_SEG_BYTES;
	if (env_cfg().seg_bytes)
		S = env_cfg().seg_bytes;
	else if (n <= ((size_t)4 << 20))
		S = 16384;
	else if (n <= ((size_t)8 << 20))
		S = 32768;
	const size_t tile = lda_deflate_tile();
	/* usable window is 32 KiB minus two tiles (the kernel inserts one tile
	 * ahead) and the lookahead; whole tiles */
	const size_t D = (32768 - 2 * tile - 272) / tile * tile;
	const size_t nseg = (n + S - 1) / S;
	const size_t slot = align_up(libdeflate_deflate_compress_bound(c, S) + 32, 16);
	const uint32_t hdr = format == LIBDEFLATE_AMD_GZIP ? 10 :
			     format == LIBDEFLATE_AMD_ZLIB ? 2 : 0;
	const uint32_t ftr = format == LIBDEFLATE_AMD_GZIP ? 8 :
			     format == LIBDEFLATE_AMD_ZLIB ? 4 : 0;

	if (out_avail <= hdr + ftr)
		return 0;
	DeviceCtx *ctx = device_ctx();
	if (!ctx || !c->streams.ensure())
		return large_fail("streams");
	hipStream_t s_copy = c->streams.copy, s_comp = c->streams.comp;
	const size_t per_slice = std::max<size_t>(1, ((size_t)32 << 20) / S);
	const size_t ns = (nseg + per_slice - 1) / per_slice;
	/* device layout: [7 u64 rows: in_off in_n out_off out_av out_n piece_off
	 * piece_n][seg_info u32][sums u32][compaction offsets of every slice]
	 * [input][slots][the slots' used parts, slice after slice] */
	std::vector<size_t> cmp_pos(ns + 1);
	cmp_pos[0] = 0;
	for (size_t k = 0; k < ns; k++)
		cmp_pos[k + 1] = cmp_pos[k] + libdeflate_amd_compact_offsets_len(
			std::min(per_slice, nseg - k * per_slice));
	const size_t ncmp = cmp_pos[ns];
	const size_t desc_bytes = align_up(nseg * (7 * 8 + 4 + 4) + 64, 64);
	const size_t cmp_at = desc_bytes, in_at = align_up(cmp_at + ncmp * 8 + 64, 64);
	const size_t out_at = align_up(in_at + n + 64, 64);
	const size_t pk_at = align_up(out_at + nseg * slot + 64, 64);
	uint8_t *st = (uint8_t *)c->stage.reserve(pk_at + nseg * slot + 64);
	if (!st) {
		complain("libdeflate_*_compress (device memory)", LIBDEFLATE_AMD_OOM);
		return 0;
	}
	{	/* the kernels' scratch for the largest launch, before any is queued */
		const size_t g = std::min<size_t>(std::min(per_slice, nseg), (size_t)ctx->num_cus);
		if (!c->scratch.reserve(g * lda_deflate_seq_words() * 8 + 16 + std::min(per_slice, nseg) * 4)) {
			complain("libdeflate_*_compress (device memory)", LIBDEFLATE_AMD_OOM);
			return 0;
		}
	}
	std::vector<uint64_t> d64(7 * nseg);
	std::vector<uint32_t> d32(nseg);
	uint64_t *in_off = &d64[0], *in_n = &d64[nseg], *out_off = &d64[2 * nseg],
		 *out_av = &d64[3 * nseg], *pc_off = &d64[5 * nseg], *pc_n = &d64[6 * nseg];
	for (size_t i = 0; i < nseg; i++) {
		const size_t dict = i ? std::min(D, i * S) / tile * tile : 0;
		const size_t len = i + 1 < nseg ? S : n - i * S;
		in_off[i] = in_at + i * S - dict;
		in_n[i] = dict + len;
		out_off[i] = out_at + i * slot;
		out_av[i] = slot;
		pc_off[i] = in_at + i * S;
		pc_n[i] = len;
		d32[i] = (uint32_t)dict | (i + 1 == nseg ? 0x80000000u : 0);
	}
	uint64_t *d_desc = (uint64_t *)st;
	uint32_t *d_seg = (uint32_t *)(st + 7 * 8 * nseg);
	uint32_t *d_sums = d_seg + nseg;
	uint64_t *d_cmp = (uint64_t *)(st + cmp_at);
	/* what comes back per slice, in pinned memory: sizes, sums, the total */
	uint64_t *h_back = (uint64_t *)c->meta.ensure(nseg * 12 + ns * 8 + 64);
	if (!h_back)
		return large_fail("pinned memory");
	uint64_t *h_out_n = h_back, *h_tot = h_back + nseg;
	uint32_t *h_sums = (uint32_t *)(h_back + nseg + ns);
	if (hipMemcpyAsync(d_desc, d64.data(), 7 * 8 * nseg, hipMemcpyHostToDevice, s_copy) != hipSuccess ||
	    hipMemcpyAsync(d_seg, d32.data(), 4 * nseg, hipMemcpyHostToDevice, s_copy) != hipSuccess)
		return large_fail("copy in");
	std::vector<hipEvent_t> ev_done(ns, nullptr);
	/* A call of one slice (up to 32 MiB) takes the copy helpers' own pieces:
	 * all copy threads on its input AND on its output (16 MiB: 10.9 -> 12.2
	 * GB/s, round 6).  A call of several slices keeps pieces of 1 MiB, with
	 * which a slice's output - a third of its input - is copied by the calling
	 * thread alone: with the threads on it too, four processes of six ran
	 * 256 MiB calls in 12.3 ms instead of 10.9 (`profiles/r06_ab.md` run 35;
	 * the calling thread is what queues the next slice's kernels). */
	const size_t piece = ns > 1 ? (size_t)1 << 20 : 0;
	size_t total = hdr;	/* bytes of the output so far */
	bool fits = true, failed = false;
	auto cleanup = [&]() {
		(void)hipStreamSynchronize(s_comp);
		(void)hipStreamSynchronize(s_copy);
		for (size_t k = 0; k < ns; k++)
			if (ev_done[k])
				(void)hipEventDestroy(ev_done[k]);
	};
	auto drain = [&](size_t k) -> bool {
		const size_t lo = k * per_slice, nk = std::min(per_slice, nseg - lo);
		if (hipEventSynchronize(ev_done[k]) != hipSuccess)
			return false;
		for (size_t i = lo; i < lo + nk; i++)
			if (h_out_n[i] == 0)
				fits = false;	/* a segment did not fit its slot: cannot happen within the bound */
		const size_t tk = (size_t)h_tot[k];
		if (!fits || total + tk + ftr > out_avail) {
			fits = false;
			return true;
		}
		if (tk && span_out(&c->pinned, st, pk_at + lo * slot, out + total, tk, s_copy, piece) != LIBDEFLATE_AMD_OK)
			return false;
		total += tk;
		return true;
	};
	for (size_t k = 0; k < ns && fits && !failed; k++) {
		const size_t lo = k * per_slice, nk = std::min(per_slice, nseg - lo);
		const size_t a = lo * S, b = std::min(n, (lo + nk) * S);
		/* (returns when the slice - and, the first time, the descriptors -
		 * are on the device) */
		if (span_in(&c->pinned, st, in_at + a, in + a, b - a, s_copy, piece) != LIBDEFLATE_AMD_OK) {
			failed = true;
			break;
		}
		int rc = compress_batch_impl(c, LIBDEFLATE_AMD_DEFLATE, nk, st, d_desc + lo,
					     d_desc + nseg + lo, st, d_desc + 2 * nseg + lo,
					     d_desc + 3 * nseg + lo, d_desc + 4 * nseg + lo, s_comp,
					     d_seg + lo);
		if (rc == LIBDEFLATE_AMD_OK && ftr)
			rc = format == LIBDEFLATE_AMD_GZIP ?
				libdeflate_amd_crc32_batch(nk, st, d_desc + 5 * nseg + lo,
							   d_desc + 6 * nseg + lo, NULL, d_sums + lo, s_comp) :
				libdeflate_amd_adler32_batch(nk, st, d_desc + 5 * nseg + lo,
							     d_desc + 6 * nseg + lo, NULL, d_sums + lo, s_comp);
		if (rc == LIBDEFLATE_AMD_OK)
			rc = libdeflate_amd_compact_batch(nk, st, d_desc + 2 * nseg + lo,
							  d_desc + 4 * nseg + lo, st + pk_at + lo * slot,
							  d_cmp + cmp_pos[k], s_comp);
		if (rc != LIBDEFLATE_AMD_OK ||
		    hipMemcpyAsync(h_out_n + lo, d_desc + 4 * nseg + lo, nk * 8, hipMemcpyDeviceToHost,
				   s_comp) != hipSuccess ||
		    hipMemcpyAsync(h_tot + k, d_cmp + cmp_pos[k] + nk, 8, hipMemcpyDeviceToHost,
				   s_comp) != hipSuccess ||
		    (ftr && hipMemcpyAsync(h_sums + lo, d_sums + lo, nk * 4, hipMemcpyDeviceToHost,
					   s_comp) != hipSuccess) ||
		    hipEventCreateWithFlags(&ev_done[k], hipEventDisableTiming) != hipSuccess ||
		    hipEventRecord(ev_done[k], s_comp) != hipSuccess) {
			failed = true;
			break;
		}
		if (k && !drain(k - 1))
			failed = true;
	}
	if (!failed && fits && !drain(ns - 1))
		failed = true;
	cleanup();
	if (failed)
		return large_fail("segmented compress");
	if (!fits || total + ftr > out_avail)
		return 0;
	const size_t at = total;
	total += ftr;
	if (format == LIBDEFLATE_AMD_GZIP) {
		/* lib/gzip_compress.c:44-79 */
		uint32_t crc = 0;
		const uint32_t shS = crc32_shift(S);	/* every piece but the last is S bytes */
		for (size_t i = 0; i < nseg; i++)
			crc = !i ? h_sums[0] :
			      pc_n[i] == S ? crc32_concat_shift(crc, h_sums[i], shS) :
					     crc32_concat(crc, h_sums[i], pc_n[i]);
		const uint8_t xfl = c->level < 2 ? 4 : c->level >= 8 ? 2 : 0;
		const uint8_t h[10] = { 0x1F, 0x8B, 8, 0, 0, 0, 0, 0, xfl, 0xFF };
		memcpy(out, h, 10);
		uint32_t isize = (uint32_t)n;
		for (int k = 0; k < 4; k++) {
			out[at + k] = (uint8_t)(crc >> (8 * k));
			out[at + 4 + k] = (uint8_t)(isize >> (8 * k));
		}
	} else if (format == LIBDEFLATE_AMD_ZLIB) {
		/* lib/zlib_compress.c:45-72 */
		uint32_t ad = 1;
		for (size_t i = 0; i < nseg; i++)
			ad = i ? adler32_concat(ad, h_sums[i], pc_n[i]) : h_sums[0];
		uint32_t fl = c->level < 2 ? 0 : c->level < 6 ? 1 : c->level < 8 ? 2 : 3;
		uint32_t hw = (0x78u << 8) | (fl << 6);
		hw |= 31 - (hw % 31);
		out[0] = (uint8_t)(hw >> 8);
		out[1] = (uint8_t)hw;
		for (int k = 0; k < 4; k++)
			out[at + k] = (uint8_t)(ad >> (8 * (3 - k)));
	}
	return total;
}

static size_t compress_one(struct libdeflate_compressor *c, int format,
			   const void *in, size_t in_nbytes, void *out,
			   size_t out_avail)
{
	if (!c)
		return 0;
	DeviceGuard on(c->device);
	if (!on.ok()) {
		complain("libdeflate_*_compress", LIBDEFLATE_AMD_NO_DEVICE);
		return 0;
	}
	/* (inputs of 4 GiB and more always take this path: the kernels index a
	 * chunk with 32 bits, the segments are 64 KiB each) */
	if (in_nbytes >= LDA_LARGE_MIN && c->level > 0 &&
	    (!env_cfg().no_segments || in_nbytes >= 0xFFFF0000u))
		return no_unwind("libdeflate_*_compress", (size_t)0, [&]() {
			return compress_large(c, format, (const uint8_t *)in, in_nbytes,
					      (uint8_t *)out, out_avail);
		});
	const void *ins[1] = { in };
	void *outs[1] = { out };
	size_t got = 0;
	int rc = libdeflate_amd_compress_batch_host(c, format, 1, ins, &in_nbytes,
						    outs, &out_avail, &got);

	if (rc != LIBDEFLATE_AMD_OK) {
		complain("libdeflate_*_compress", rc);
		return 0;
	}
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
