/*
 * host_decompress.hip - C-ABI of the decompressor: object lifetime, the
 * device batch entry point, the host-pointer batch, and the nine
 * single-buffer libdeflate_*_decompress[_ex] calls as batches of one.
 *
 * Reference interfaces replaced: libdeflate.h:181-188 (alloc), :242-315
 * (decompress), :322-323 (free), :363-365 (allocator).
 */
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <new>
#include <string.h>
#include <vector>

#include "host_objects.h"
#include "kernels.h"

namespace lda {

malloc_func_t g_malloc = nullptr;
free_func_t g_free = nullptr;

void *DevBuf::reserve(size_t n)
{
	if (n <= cap)
		return p;
	release();
	size_t want = align_up(n + n / 8 + 4096, 4096);
	hipError_t e = hipMalloc(&p, want);
	if (e != hipSuccess) {
		p = nullptr;
		set_error("hipMalloc(%zu): %s", want, hipGetErrorString(e));
		return nullptr;
	}
	cap = want;
	return p;
}

void DevBuf::release()
{
	if (p)
		(void)hipFree(p);
	p = nullptr;
	cap = 0;
}

/* resolve the allocator the way lib/deflate_compress.c:3910-3917 does */
bool pick_allocator(const struct libdeflate_options *options,
		    malloc_func_t *m, free_func_t *f)
{
	*m = g_malloc ? g_malloc : malloc;
	*f = g_free ? g_free : free;
	if (options) {
		/* lib/deflate_compress.c:3885-3886 */
		if (options->sizeof_options != sizeof(*options))
			return false;
		if (options->malloc_func)
			*m = options->malloc_func;
		if (options->free_func)
			*f = options->free_func;
	}
	return true;
}

} /* namespace lda */

using namespace lda;

/* device bytes behind d->tokens for a batch of n streams: the token rows of
 * every wave of the grid, 16 bytes of counters */
/* waves (= streams in flight) per CU: what the kernel's LDS leaves room for, at
 * most LDA_INFLATE_WAVES_PER_CU (16: four per SIMD at 128 VGPRs) */
static size_t inflate_wave_lds(void)
{
	return lda_inflate_lds_per_stream() + lda_inflate_lds_shared() + lda_inflate_window_bytes();
}

static size_t inflate_waves_per_cu(void)
{
	const size_t fit = 163840 / inflate_wave_lds(), want = (size_t)env_cfg().inflate_waves_per_cu;

	return want < fit ? want : fit;
}

static size_t inflate_tokens_bytes(size_t n, int num_cus)
{
	size_t grid = (size_t)num_cus * inflate_waves_per_cu();

	if (grid > n)
		grid = n;
	return grid * lda_inflate_tokcap() * 4 + 16;
}

/* lib/utils.c:61-67 */
extern "C" LIBDEFLATEAPI void
libdeflate_set_memory_allocator(void *(*malloc_func)(size_t),
				void (*free_func)(void *))
{
	g_malloc = malloc_func;
	g_free = free_func;
}

extern "C" LIBDEFLATEAPI struct libdeflate_decompressor *
libdeflate_alloc_decompressor_ex(const struct libdeflate_options *options)
{
	malloc_func_t m;
	free_func_t f;

	if (!pick_allocator(options, &m, &f))
		return NULL;
	if (!device_ctx()) {
		fprintf(stderr, "libdeflate_amd: alloc_decompressor: no usable "
			"gfx950 device (%s); no CPU fallback\n",
			libdeflate_amd_last_error());
		return NULL;
	}
	void *mem = m(sizeof(struct libdeflate_decompressor));
	if (!mem)
		return NULL;
	struct libdeflate_decompressor *d =
		new (mem) libdeflate_decompressor();
	d->free_func = f;
	d->malloc_func = m;
	d->device = 0;
	(void)hipGetDevice(&d->device);	/* (device_ctx() above has seen it work) */
	for (int k = 0; k < LDA_MAX_SHARDS; k++)
		d->shard[k] = NULL;
	return d;
}

extern "C" LIBDEFLATEAPI struct libdeflate_decompressor *
libdeflate_alloc_decompressor(void)
{
	return libdeflate_alloc_decompressor_ex(NULL);
}

extern "C" LIBDEFLATEAPI void
libdeflate_free_decompressor(struct libdeflate_decompressor *d)
{
	if (!d)
		return;
	for (int k = 0; k < LDA_MAX_SHARDS; k++)
		libdeflate_free_decompressor(d->shard[k]);
	DeviceGuard on(d->device);
	d->scratch.release();
	d->stage.release();
	d->tokens.release();
	d->sin.release();
	d->squeue.release();
	d->schunks.release();
	d->srepair.release();
	d->swin.release();
	d->shdr.release();
	d->shint.release();
	d->ssym.release();
	d->sout.release();
	d->pinned.release();
	d->meta.release();
	d->streams.release();
	free_func_t f = d->free_func;
	d->~libdeflate_decompressor();
	f(d);
}

/* ------------------------------------------------------------------ */

extern "C" LIBDEFLATEAPI int
libdeflate_amd_decompress_batch(struct libdeflate_decompressor *d, int format,
				size_t n, const void *d_in,
				const uint64_t *d_in_offsets,
				const uint64_t *d_in_nbytes, void *d_out,
				const uint64_t *d_out_offsets,
				const uint64_t *d_out_avail, int32_t *d_results,
				uint64_t *d_actual_in, uint64_t *d_actual_out,
				void *stream)
{
	if (!d) {
		set_error("decompress_batch: bad argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	DeviceGuard on(d->device);
	if (!on.ok())
		return LIBDEFLATE_AMD_NO_DEVICE;
	DeviceCtx *c = device_ctx();
	hipStream_t st = (hipStream_t)stream;

	if (!c)
		return LIBDEFLATE_AMD_NO_DEVICE;
	if (n == 0)
		return LIBDEFLATE_AMD_OK;
	if (!d || !d_in || !d_in_offsets || !d_in_nbytes || !d_out ||
	    !d_out_offsets || !d_out_avail || !d_results ||
	    format < LIBDEFLATE_AMD_DEFLATE || format > LIBDEFLATE_AMD_GZIP) {
		set_error("decompress_batch: bad argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	/* scratch: [sums u32 x n][actual_in u64 x n][actual_out u64 x n][order u32 x n] */
	size_t sums_bytes = align_up(n * 4, 16);
	uint8_t *s = (uint8_t *)d->scratch.reserve(sums_bytes + 16 * n + 4 * n + 16);
	if (!s)
		return LIBDEFLATE_AMD_OOM;
	uint32_t *sums = (uint32_t *)s;
	uint64_t *ain = d_actual_in ? d_actual_in : (uint64_t *)(s + sums_bytes);
	uint64_t *aout = d_actual_out ? d_actual_out :
			 (uint64_t *)(s + sums_bytes + 8 * n);
	int exact_fill = d_actual_out == NULL;

	/* streams per wave: a stream's decode is one serial dependence chain, so
	 * per-stream speed is best with few streams per wave (less waiting for
	 * the slowest lane, fewer divergent paths); lpw only grows once the
	 * batch no longer fits in flight (measured best: 2 at 4096 streams, 8 at
	 * 65536 on 256 CUs) */
	/* (as many streams as a workgroup's LDS holds tables for) */
	uint32_t lpw_max = 64;
	while (lda_inflate_lds_per_stream() * lpw_max + lda_inflate_lds_shared() > 163840)
		lpw_max >>= 1;
	uint32_t lpw = 2;
	while (lpw < lpw_max && (size_t)lpw * 32 * (size_t)c->num_cus < n)
		lpw <<= 1;
	if (env_cfg().inflate_lpw)	/* tuning aid */
		lpw = (uint32_t)env_cfg().inflate_lpw;
	if (lpw > lpw_max)
		lpw = lpw_max;
	/* wave per stream with sub-block parallel token decoding (the default):
	 * one stream keeps all 64 lanes of its wave busy, so even a batch that
	 * is small next to the machine (4096 streams on 1024 SIMDs) runs at the
	 * rate of a huge one.  LDA_INFLATE_PAR=0 selects lane-per-stream. */
	const bool par = env_cfg().inflate_par;
	if (!c->inflate_attr_set.load(std::memory_order_acquire)) {
		LDA_HIP_TRY(hipFuncSetAttribute(
				(const void *)lda_inflate_batch_kernel,
				hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)(lda_inflate_lds_per_stream() * lpw_max +
				      lda_inflate_lds_shared())),
			    LIBDEFLATE_AMD_NO_DEVICE);
		c->inflate_attr_set.store(true, std::memory_order_release);
	}
	if (par) {
		const size_t grid_max = (size_t)c->num_cus * inflate_waves_per_cu();
		size_t grid = grid_max < n ? grid_max : n;
		/* token rows of every wave (lda_inflate_tokcap() words: one row
		 * of 64 tokens per parse step of a round), then the counter the
		 * waves take their second and later streams from */
		const size_t tok_bytes = grid * lda_inflate_tokcap() * 4;
		uint32_t *tok = (uint32_t *)d->tokens.reserve(inflate_tokens_bytes(n, c->num_cus));
		if (!tok)
			return LIBDEFLATE_AMD_OOM;
		uint32_t *next = (uint32_t *)((uint8_t *)tok + tok_bytes);
		LDA_HIP_TRY(hipMemsetAsync(next, 0, 16, st), LIBDEFLATE_AMD_NO_DEVICE);
		/* a batch of more streams than wave slots is handed out costliest
		 * stream first (a batch that fits the grid starts all at once) */
		uint32_t *order = NULL;
		if (n > grid && n < 0xFFFFFFFFull) {
			order = (uint32_t *)(s + sums_bytes + 16 * n);
			hipLaunchKernelGGL(lda_inflate_order_kernel, dim3(1), dim3(1024), 0, st,
					   (uint64_t)n, d_in_nbytes, d_out_avail, order);
		}
		size_t lds = inflate_wave_lds();
		hipLaunchKernelGGL(lda_inflate_wave_kernel, dim3((unsigned)grid),
				   dim3(64), lds, st, (uint64_t)n, format, tok, next,
				   (const uint32_t *)order,
				   (const uint8_t *)d_in, d_in_offsets, d_in_nbytes,
				   (uint8_t *)d_out, d_out_offsets, d_out_avail,
				   d_results, ain, aout);
	} else {
		size_t lds = lda_inflate_lds_per_stream() * lpw +
			     lda_inflate_lds_shared();
		hipLaunchKernelGGL(lda_inflate_batch_kernel,
				   dim3((unsigned)((n + lpw - 1) / lpw)), dim3(64),
				   lds, st, (uint64_t)n, format, lpw,
				   (const uint8_t *)d_in, d_in_offsets, d_in_nbytes,
				   (uint8_t *)d_out, d_out_offsets, d_out_avail,
				   d_results, ain, aout);
	}
	LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	if (format != LIBDEFLATE_AMD_DEFLATE) {
		int rc = format == LIBDEFLATE_AMD_GZIP ?
			libdeflate_amd_crc32_batch(n, d_out, d_out_offsets, aout,
						   NULL, sums, stream) :
			libdeflate_amd_adler32_batch(n, d_out, d_out_offsets,
						     aout, NULL, sums, stream);
		if (rc != LIBDEFLATE_AMD_OK)
			return rc;
	}
	if (format != LIBDEFLATE_AMD_DEFLATE || exact_fill) {
		hipLaunchKernelGGL(lda_inflate_finalize_kernel,
				   dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
				   st, (uint64_t)n, format, exact_fill,
				   (const uint8_t *)d_in, d_in_offsets, d_out_avail,
				   sums, d_results, ain, aout);
		LDA_HIP_TRY(hipGetLastError(), LIBDEFLATE_AMD_NO_DEVICE);
	}
	return LIBDEFLATE_AMD_OK;
}

static int decompress_batch_host_body(struct libdeflate_decompressor *d,
				     int format, size_t n,
				     const void *const *in,
				     const size_t *in_nbytes, void *const *out,
				     const size_t *out_avail, int32_t *results,
				     size_t *actual_in, size_t *actual_out);

extern "C" LIBDEFLATEAPI int
libdeflate_amd_decompress_batch_host(struct libdeflate_decompressor *d,
				     int format, size_t n,
				     const void *const *in,
				     const size_t *in_nbytes, void *const *out,
				     const size_t *out_avail, int32_t *results,
				     size_t *actual_in, size_t *actual_out)
{
	return no_unwind("decompress_batch_host", (int)LIBDEFLATE_AMD_OOM, [&]() {
		if (!d || !out_avail || n == 0)
			return decompress_batch_host_body(d, format, n, in, in_nbytes, out, out_avail,
							  results, actual_in, actual_out);
		/* several GPUs (LDA_DEVICES, host_fanout.hip): shards of about equal
		 * OUTPUT (what a stream costs to decode), an object and a host thread
		 * per device, results in place */
		size_t bounds[LDA_MAX_SHARDS + 1];
		int devs[LDA_MAX_SHARDS];
		const size_t shards = fanout_plan(d->device, n, out_avail, bounds, devs);
		fanout_note(shards);
		if (shards < 2)
			return decompress_batch_host_body(d, format, n, in, in_nbytes, out, out_avail,
							  results, actual_in, actual_out);
		if (!in || !in_nbytes || !out || !results) {
			set_error("decompress_batch_host: NULL argument");
			return (int)LIBDEFLATE_AMD_BAD_ARG;
		}
		for (size_t k = 1; k < shards; k++) {
			if (d->shard[k])
				continue;
			DeviceGuard on(devs[k]);
			struct libdeflate_options o = {};
			o.sizeof_options = sizeof(o);
			o.malloc_func = d->malloc_func;
			o.free_func = d->free_func;
			if (on.ok())
				d->shard[k] = libdeflate_alloc_decompressor_ex(&o);
			if (!d->shard[k]) {
				/* a device that cannot take its shard (out of memory,
				 * refused by the self-check, busy): the batch stays on
				 * the object's own device rather than fail - the reason
				 * stays in libdeflate_amd_last_error() */
				fanout_note(1);
				return decompress_batch_host_body(d, format, n, in, in_nbytes, out, out_avail,
							  results, actual_in, actual_out);
			}
		}
		return fanout_run(shards, [&](size_t k) {
			const size_t lo = bounds[k], cnt = bounds[k + 1] - lo;
			return decompress_batch_host_body(k ? d->shard[k] : d, format, cnt, in + lo,
							  in_nbytes + lo, out + lo, out_avail + lo,
							  results + lo, actual_in ? actual_in + lo : NULL,
							  actual_out ? actual_out + lo : NULL);
		});
	});
}

static int decompress_batch_host_body(struct libdeflate_decompressor *d,
				     int format, size_t n,
				     const void *const *in,
				     const size_t *in_nbytes, void *const *out,
				     const size_t *out_avail, int32_t *results,
				     size_t *actual_in, size_t *actual_out)
{
	if (n == 0)
		return device_ctx() ? LIBDEFLATE_AMD_OK : LIBDEFLATE_AMD_NO_DEVICE;
	if (!d || !in || !in_nbytes || !out || !out_avail || !results) {
		set_error("decompress_batch_host: NULL argument");
		return LIBDEFLATE_AMD_BAD_ARG;
	}
	DeviceGuard on(d->device);
	if (!on.ok() || !device_ctx())
		return LIBDEFLATE_AMD_NO_DEVICE;
	/* In slices like libdeflate_amd_compress_batch_host(): the kernels of
	 * slice k (compute stream) run while the host packs and sends slice k + 1
	 * and unpacks slice k - 1 (copy stream).  Slices of at least 256 MiB of
	 * output: a launch of few streams is latency bound (one wave per stream),
	 * so small slices cost more kernel time than their overlap saves.
	 * staging layout: 6 u64 arrays + results, then inputs, then outputs */
	enum { MAX_SLICES = 8 };
	size_t bounds[MAX_SLICES + 1];
	const size_t ns = slice_by_bytes(n, out_avail, MAX_SLICES, (size_t)256 << 20, bounds);
	std::vector<uint64_t> desc(4 * n);
	uint64_t *in_off = &desc[0], *in_n = &desc[n], *out_off = &desc[2 * n],
		 *out_av = &desc[3 * n];
	size_t desc_bytes = align_up(6 * n * 8 + n * 4, 64);
	size_t pos = desc_bytes;
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
	uint8_t *st = (uint8_t *)d->stage.reserve(pos + 64);
	if (!st)
		return LIBDEFLATE_AMD_OOM;
	if (!d->streams.ensure())
		return LIBDEFLATE_AMD_NO_DEVICE;
	/* scratch of the largest launch up front (see the compress side) */
	if (!d->scratch.reserve(align_up(n * 4, 16) + 16 * n + 4 * n + 16))
		return LIBDEFLATE_AMD_OOM;
	{
		if (env_cfg().inflate_par &&
		    !d->tokens.reserve(inflate_tokens_bytes(n, device_ctx()->num_cus)))
			return LIBDEFLATE_AMD_OOM;
	}
	/* per-chunk read-backs land in pinned memory (asynchronous for real) and
	 * go to the caller's arrays in drain() */
	uint64_t *h_back = (uint64_t *)d->meta.ensure(n * (8 + 8 + 4));
	if (!h_back)
		return LIBDEFLATE_AMD_OOM;
	uint64_t *h_ain = h_back, *h_aout = h_back + n;
	int32_t *h_res = (int32_t *)(h_back + 2 * n);
	hipStream_t s_copy = d->streams.copy, s_comp = d->streams.comp;
	LDA_HIP_TRY(hipMemcpyAsync(st, desc.data(), 4 * n * 8, hipMemcpyHostToDevice,
				   s_copy), LIBDEFLATE_AMD_NO_DEVICE);
	uint64_t *d_desc = (uint64_t *)st;
	int32_t *d_res = (int32_t *)(st + 6 * n * 8);
	std::vector<uint64_t> nout(n);
	hipEvent_t ev_done[MAX_SLICES] = {};
	int rc = LIBDEFLATE_AMD_OK;
	auto cleanup = [&]() {
		(void)hipStreamSynchronize(s_comp);
		(void)hipStreamSynchronize(s_copy);
		for (size_t k = 0; k < ns; k++)
			if (ev_done[k])
				(void)hipEventDestroy(ev_done[k]);
	};
	auto drain = [&](size_t k) -> int {
		const size_t lo = bounds[k], nk = bounds[k + 1] - lo;
		LDA_HIP_TRY(hipEventSynchronize(ev_done[k]), LIBDEFLATE_AMD_NO_DEVICE);
		/* bytes to bring back per chunk: the produced ones of successful
		 * chunks (output is undefined on failure, libdeflate.h:216-217) */
		for (size_t i = lo; i < lo + nk; i++) {
			results[i] = h_res[i];
			const bool ok = results[i] == LIBDEFLATE_SUCCESS;
			if (actual_in)
				actual_in[i] = ok ? h_ain[i] : 0;
			if (actual_out)
				actual_out[i] = ok ? h_aout[i] : 0;
			nout[i] = !ok ? 0 : actual_out ? h_aout[i] : out_avail[i];
		}
		return copy_out_packed(&d->pinned, st, nk, out + lo, nout.data() + lo,
				       out_off + lo, s_copy);
	};
	for (size_t k = 0; k < ns && rc == LIBDEFLATE_AMD_OK; k++) {
		const size_t lo = bounds[k], nk = bounds[k + 1] - lo;
		rc = copy_in_packed(&d->pinned, st, nk, in + lo, in_nbytes + lo, in_off + lo, s_copy);
		if (rc != LIBDEFLATE_AMD_OK)
			break;
		rc = libdeflate_amd_decompress_batch(
			d, format, nk, st, d_desc + lo, d_desc + n + lo, st, d_desc + 2 * n + lo,
			d_desc + 3 * n + lo, d_res + lo, d_desc + 4 * n + lo,
			actual_out ? d_desc + 5 * n + lo : NULL, s_comp);
		if (rc != LIBDEFLATE_AMD_OK)
			break;
		if (hipMemcpyAsync(h_ain + lo, d_desc + 4 * n + lo, nk * 8,
				   hipMemcpyDeviceToHost, s_comp) != hipSuccess ||
		    (actual_out &&
		     hipMemcpyAsync(h_aout + lo, d_desc + 5 * n + lo, nk * 8,
				    hipMemcpyDeviceToHost, s_comp) != hipSuccess) ||
		    hipMemcpyAsync(h_res + lo, d_res + lo, nk * 4, hipMemcpyDeviceToHost,
				   s_comp) != hipSuccess ||
		    hipEventCreateWithFlags(&ev_done[k], hipEventDisableTiming) != hipSuccess ||
		    hipEventRecord(ev_done[k], s_comp) != hipSuccess) {
			set_error("decompress_batch_host: %s", hipGetErrorString(hipGetLastError()));
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

/* ---- the single-buffer calls: batches of one ---- */

static enum libdeflate_result
decompress_one(struct libdeflate_decompressor *d, int format, const void *in,
	       size_t in_nbytes, void *out, size_t out_avail,
	       size_t *actual_in_ret, size_t *actual_out_ret)
{
	const void *ins[1] = { in };
	void *outs[1] = { out };
	int32_t res = LIBDEFLATE_BAD_DATA;
	size_t ain = 0, aout = 0;
	if (!d)
		return LIBDEFLATE_BAD_DATA;
	/* (a single stream never fans out: it is one device's work) */
	DeviceGuard on(d->device);
	if (!on.ok()) {
		complain("libdeflate_*_decompress", LIBDEFLATE_AMD_NO_DEVICE);
		return LIBDEFLATE_BAD_DATA;	/* see below: a library-side failure */
	}
	/* a large stream: many waves (host_stream.hip); it answers only for what
	 * it decoded cleanly, everything else goes on to the sequential kernel */
	if (d && in && out &&
	    no_unwind("libdeflate_*_decompress (many waves)", false, [&]() {
		    return decompress_stream_parallel(d, format, (const uint8_t *)in, in_nbytes,
						      (uint8_t *)out, out_avail,
						      actual_out_ret == NULL, &res, &ain, &aout);
	    })) {
		if (res == LIBDEFLATE_SUCCESS) {
			if (actual_in_ret)
				*actual_in_ret = ain;
			if (actual_out_ret)
				*actual_out_ret = aout;
		}
		return (enum libdeflate_result)res;
	}
	int rc = libdeflate_amd_decompress_batch_host(
		d, format, 1, ins, &in_nbytes, outs, &out_avail, &res, &ain,
		actual_out_ret ? &aout : NULL);

	if (rc != LIBDEFLATE_AMD_OK) {
		/* the reference never aborts: a library-side failure comes back
		 * through the result (the reason is in libdeflate_amd_last_error).
		 * Always BAD_DATA, also when device or pinned memory ran out:
		 * INSUFFICIENT_SPACE means "the output buffer was too small" to
		 * the reference's callers, who answer it with a larger buffer -
		 * which makes a memory shortage worse on every retry */
		complain("libdeflate_*_decompress", rc);
		return LIBDEFLATE_BAD_DATA;
	}
	if (res == LIBDEFLATE_SUCCESS) {
		if (actual_in_ret)
			*actual_in_ret = ain;
		if (actual_out_ret)
			*actual_out_ret = aout;
	}
	return (enum libdeflate_result)res;
}

#define DEFINE_DECOMPRESS(name, fmt)                                          \
	extern "C" LIBDEFLATEAPI enum libdeflate_result                       \
	libdeflate_##name##_decompress_ex(struct libdeflate_decompressor *d, \
					  const void *in, size_t in_nbytes,   \
					  void *out, size_t out_avail,        \
					  size_t *actual_in_ret,              \
					  size_t *actual_out_ret)             \
	{                                                                     \
		return decompress_one(d, fmt, in, in_nbytes, out, out_avail,  \
				      actual_in_ret, actual_out_ret);         \
	}                                                                     \
	extern "C" LIBDEFLATEAPI enum libdeflate_result                       \
	libdeflate_##name##_decompress(struct libdeflate_decompressor *d,     \
				       const void *in, size_t in_nbytes,      \
				       void *out, size_t out_avail,           \
				       size_t *actual_out_ret)                \
	{                                                                     \
		return decompress_one(d, fmt, in, in_nbytes, out, out_avail,  \
				      NULL, actual_out_ret);                  \
	}

DEFINE_DECOMPRESS(deflate, LIBDEFLATE_AMD_DEFLATE)
DEFINE_DECOMPRESS(zlib, LIBDEFLATE_AMD_ZLIB)
DEFINE_DECOMPRESS(gzip, LIBDEFLATE_AMD_GZIP)

/*
 * A gzip buffer made of SEVERAL members (what `cat a.gz b.gz`, pigz -i and
 * the BGZF container of BAM / tabix files produce).  libdeflate_gzip_decompress
 * decodes the first member only, by design (lib/gzip_decompress.c:103-131);
 * its caller loops, as programs/gzip.c:236-299 does.  This is that loop behind
 * one call - and where the members say how long they are (the "BC" extra
 * subfield of BGZF: total member size - 1) the whole file is indexed from the
 * headers alone, the output offsets follow from the ISIZE footers, and ALL
 * members go to the device as ONE batch.  Anything else (plain concatenations,
 * or what follows the indexed part of a file) is decoded member after member:
 * a member's end is only known once it has been decoded.
 */
static enum libdeflate_result
decompress_members_body(struct libdeflate_decompressor *d,
			const void *in_, size_t in_nbytes,
			void *out_, size_t out_avail,
			size_t *actual_in_ret,
			size_t *actual_out_ret,
			size_t *members_ret);

extern "C" LIBDEFLATEAPI enum libdeflate_result
libdeflate_amd_gzip_decompress_members(struct libdeflate_decompressor *d,
				       const void *in_, size_t in_nbytes,
				       void *out_, size_t out_avail,
				       size_t *actual_in_ret,
				       size_t *actual_out_ret,
				       size_t *members_ret)
{
	/* (a library-side failure is BAD_DATA, see decompress_one()) */
	return no_unwind("libdeflate_amd_gzip_decompress_members", LIBDEFLATE_BAD_DATA, [&]() {
		return decompress_members_body(d, in_, in_nbytes, out_, out_avail, actual_in_ret,
					       actual_out_ret, members_ret);
	});
}

static enum libdeflate_result
decompress_members_body(struct libdeflate_decompressor *d,
			const void *in_, size_t in_nbytes,
			void *out_, size_t out_avail,
			size_t *actual_in_ret,
			size_t *actual_out_ret,
			size_t *members_ret)
{
	const uint8_t *in = (const uint8_t *)in_;
	uint8_t *out = (uint8_t *)out_;
	std::vector<size_t> off, len, osz;
	size_t pos = 0, total = 0;
	bool indexed = true;

	while (pos < in_nbytes && indexed) {
		const uint8_t *p = in + pos;
		const size_t left = in_nbytes - pos;
		size_t bsize = 0;

		if (left < 18 + 6 || p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 ||
		    !(p[3] & 4)) {
			indexed = false;
			break;
		}
		const size_t xlen = p[10] | ((size_t)p[11] << 8);
		if (12 + xlen + 8 > left) {
			indexed = false;
			break;
		}
		for (size_t x = 12; x + 4 <= 12 + xlen;) {
			const size_t slen = p[x + 2] | ((size_t)p[x + 3] << 8);
			if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen)
				bsize = (size_t)(p[x + 4] | ((size_t)p[x + 5] << 8)) + 1;
			x += 4 + slen;
		}
		if (bsize < 12 + xlen + 8 || bsize > left) {
			indexed = false;
			break;
		}
		const size_t isize = p[bsize - 4] | ((size_t)p[bsize - 3] << 8) |
				     ((size_t)p[bsize - 2] << 16) | ((size_t)p[bsize - 1] << 24);
		off.push_back(pos);
		len.push_back(bsize);
		osz.push_back(isize);
		total += isize;
		pos += bsize;
	}
	size_t ipos = 0, opos = 0, members = 0;
	/* The indexed PREFIX goes to the device as one batch; whatever follows it
	 * (a plain gzip member appended to a BGZF file, say) continues member
	 * after member from where the index ends - not from the start. */
	if (!off.empty()) {
		if (total > out_avail)
			return LIBDEFLATE_INSUFFICIENT_SPACE;
		const size_t n = off.size();
		std::vector<const void *> ins(n);
		std::vector<void *> outs(n);
		std::vector<int32_t> res(n);
		std::vector<size_t> ain(n);
		size_t o = 0;
		for (size_t i = 0; i < n; i++) {
			ins[i] = in + off[i];
			outs[i] = out + o;
			o += osz[i];
		}
		/* exact fill: a member whose ISIZE lies comes back SHORT_OUTPUT /
		 * INSUFFICIENT_SPACE and fails the call */
		int rc = libdeflate_amd_decompress_batch_host(
			d, LIBDEFLATE_AMD_GZIP, n, ins.data(), len.data(), outs.data(),
			osz.data(), res.data(), ain.data(), NULL);
		if (rc != LIBDEFLATE_AMD_OK) {
			complain("libdeflate_amd_gzip_decompress_members", rc);
			return LIBDEFLATE_BAD_DATA;	/* see decompress_one() */
		}
		for (size_t i = 0; i < n; i++) {
			if (res[i] != LIBDEFLATE_SUCCESS)
				return (enum libdeflate_result)res[i];
			if (ain[i] != len[i])	/* the member is shorter than it says */
				return LIBDEFLATE_BAD_DATA;
		}
		ipos = pos;
		opos = total;
		members = n;
	}
	/* member after member (programs/gzip.c:236-299) */
	while (ipos < in_nbytes || members == 0) {
		size_t ain = 0, aout = 0;
		enum libdeflate_result r = decompress_one(
			d, LIBDEFLATE_AMD_GZIP, in + ipos, in_nbytes - ipos, out + opos,
			out_avail - opos, &ain, &aout);
		if (r != LIBDEFLATE_SUCCESS)
			return r;
		if (ain == 0 || ain > in_nbytes - ipos)
			return LIBDEFLATE_BAD_DATA;
		ipos += ain;
		opos += aout;
		members++;
	}
	if (actual_in_ret)
		*actual_in_ret = ipos;
	if (actual_out_ret)
		*actual_out_ret = opos;
	if (members_ret)
		*members_ret = members;
	return LIBDEFLATE_SUCCESS;
}
