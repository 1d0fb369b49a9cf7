/*
 * host_context.hip - per-device context, constant-table generation, errors.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "host_common.h"
#include "kernels.h"

namespace lda {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

[[noreturn]] void die(const char *what)
{
	fprintf(stderr,
		"libdeflate_amd: %s failed on the gfx950 device (%s) and this "
		"call has no error return. This library has no CPU fallback.\n",
		what, g_err);
	abort();
}

void complain(const char *what, int status)
{
	fprintf(stderr, "libdeflate_amd: %s: status %d (%s)\n", what, status, g_err);
}

static EnvCfg read_env()
{
	EnvCfg c;
	c.no_small = getenv("LDA_NO_SMALL") != nullptr;
	c.no_segments = getenv("LDA_NO_SEGMENTS") != nullptr;
	if (const char *e = getenv("LDA_INFLATE_LPW")) {
		int v = atoi(e);
		if (v >= 1 && v <= 64)
			c.inflate_lpw = v;
	}
	if (const char *e = getenv("LDA_INFLATE_PAR"))
		c.inflate_par = atoi(e) != 0;
	if (const char *e = getenv("LDA_HOST_THREADS")) {
		int v = atoi(e);
		if (v >= 1 && v <= 16)
			c.host_threads = v;
	}
	if (const char *e = getenv("LDA_INFLATE_WAVES_PER_CU")) {
		int v = atoi(e);
		if (v >= 1 && v <= 16)
			c.inflate_waves_per_cu = v;
	}
	if (const char *e = getenv("LDA_SEG_BYTES")) {
		size_t v = (size_t)strtoull(e, nullptr, 0);
		if (v >= 8192 && v <= 65536 && v % 4096 == 0)
			c.seg_bytes = v;
	}
	c.no_stream_par = getenv("LDA_NO_STREAM_PAR") != nullptr;
	if (const char *e = getenv("LDA_STREAM_PAR_MIN"))
		c.stream_par_min = (size_t)strtoull(e, nullptr, 0);
	if (const char *e = getenv("LDA_STREAM_WINDOW")) {
		c.stream_window = (size_t)strtoull(e, nullptr, 0);
		if (c.stream_window && c.stream_window < 32768)
			c.stream_window = 32768;
	}
	if (const char *e = getenv("LDA_STREAM_CHUNK"))
		c.stream_chunk = (size_t)strtoull(e, nullptr, 0);
	if (const char *e = getenv("LDA_DEVICES")) {
		if (!strcmp(e, "all")) {
			c.devices = -1;
		} else {
			int v = atoi(e);
			if (v >= 1 && v <= 16)
				c.devices = v;
		}
	}
	c.fanout_oversub = getenv("LDA_FANOUT_OVERSUB") != nullptr;
	c.no_selfcheck = getenv("LDA_NO_SELFCHECK") != nullptr;
	return c;
}

static EnvCfg g_env = read_env();	/* once, when the library is loaded */

const EnvCfg &env_cfg()
{
	return g_env;
}

bool PinnedPair::ensure(size_t want)
{
	if (want > LDA_PINNED_SLICE)
		want = LDA_PINNED_SLICE;
	if (want < LDA_PINNED_MIN)
		want = LDA_PINNED_MIN;
	if (cap >= want)
		return true;
	/* grow: round up to a power of two so that a sequence of slowly
	 * growing batches does not re-pin every time */
	size_t ncap = LDA_PINNED_MIN;
	while (ncap < want)
		ncap <<= 1;
	uint8_t *nb[2] = { nullptr, nullptr };
	for (int b = 0; b < 2; b++) {
		hipError_t e = hipHostMalloc((void **)&nb[b], ncap, hipHostMallocDefault);
		if (e == hipSuccess && !ev[b])
			e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
		if (e != hipSuccess) {
			set_error("pinned staging (%zu bytes): %s", ncap, hipGetErrorString(e));
			for (int k = 0; k <= b; k++)
				if (nb[k])
					(void)hipHostFree(nb[k]);
			return false;	/* what was there stays usable */
		}
	}
	for (int b = 0; b < 2; b++) {
		if (buf[b])
			(void)hipHostFree(buf[b]);
		buf[b] = nb[b];
	}
	cap = ncap;
	return true;
}

void *PinnedBuf::ensure(size_t want)
{
	if (want <= cap)
		return p;
	size_t ncap = 4096;
	while (ncap < want)
		ncap <<= 1;
	void *np = nullptr;
	hipError_t e = hipHostMalloc(&np, ncap, hipHostMallocDefault);
	if (e != hipSuccess) {
		set_error("pinned read-back buffer (%zu bytes): %s", ncap, hipGetErrorString(e));
		return nullptr;
	}
	release();
	p = np;
	cap = ncap;
	return p;
}

void PinnedBuf::release()
{
	if (p)
		(void)hipHostFree(p);
	p = nullptr;
	cap = 0;
}

bool StreamPair::ensure()
{
	if (copy && comp && mark && mark2)
		return true;
	if ((!copy && hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) != hipSuccess) ||
	    (!comp && hipStreamCreateWithFlags(&comp, hipStreamNonBlocking) != hipSuccess) ||
	    (!mark && hipEventCreateWithFlags(&mark, hipEventDisableTiming) != hipSuccess) ||
	    (!mark2 && hipEventCreateWithFlags(&mark2, hipEventDisableTiming) != hipSuccess)) {
		set_error("hipStreamCreate: %s", hipGetErrorString(hipGetLastError()));
		return false;
	}
	return true;
}

void StreamPair::release()
{
	if (copy)
		(void)hipStreamDestroy(copy);
	if (comp)
		(void)hipStreamDestroy(comp);
	if (mark)
		(void)hipEventDestroy(mark);
	if (mark2)
		(void)hipEventDestroy(mark2);
	copy = comp = nullptr;
	mark = mark2 = nullptr;
}

size_t slice_by_bytes(size_t n, const size_t *nbytes, size_t max_slices,
		      size_t min_bytes, size_t *bounds)
{
	size_t total = 0;
	for (size_t i = 0; i < n; i++)
		total += nbytes[i];
	size_t k = min_bytes ? total / min_bytes : max_slices;
	if (k > max_slices)
		k = max_slices;
	if (k > n)
		k = n;
	if (k < 1)
		k = 1;
	bounds[0] = 0;
	size_t done = 0, i = 0;
	for (size_t s = 1; s < k; s++) {
		const size_t want = total / k * s;
		while (i < n && done < want)
			done += nbytes[i++];
		if (i <= bounds[s - 1])		/* never an empty slice */
			i = bounds[s - 1] + 1;
		if (i > n - (k - s))
			i = n - (k - s);
		bounds[s] = i;
	}
	bounds[k] = n;
	return k;
}

void PinnedPair::release()
{
	for (int b = 0; b < 2; b++) {
		if (buf[b])
			(void)hipHostFree(buf[b]);
		if (ev[b])
			(void)hipEventDestroy(ev[b]);
		buf[b] = nullptr;
		ev[b] = nullptr;
	}
	cap = 0;
}

/*
 * The host side of the host-pointer batches is memcpy between the caller's
 * (pageable) buffers and the pinned staging: one thread moves ~12 GB/s, PCIe
 * 55.  Large slices are packed / unpacked by a few threads side by side
 * (LDA_HOST_THREADS, default 4; 1 = the calling thread alone): workers of a
 * small pool that is started on first use and shared by all objects (a thread
 * per slice and call cost 20-30 us each to create - as much as the copy of a
 * 4 MiB slice takes).  The workers are detached and never joined: they sleep
 * on a condition variable between calls; a forked child starts its own (see
 * host_pool()).
 */
namespace {

struct HostPool {
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::function<void()>> q;
	unsigned nthreads = 0;

	void run()
	{
		for (;;) {
			std::function<void()> f;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return !q.empty(); });
				f = std::move(q.front());
				q.pop_front();
			}
			f();
		}
	}
	/* workers that exist after trying for `want` (a thread that cannot be
	 * created - EAGAIN under a thread or cgroup limit - is simply not there) */
	unsigned ensure(unsigned want)
	{
		std::lock_guard<std::mutex> lk(mu);
		while (nthreads < want) {
			try {
				std::thread(&HostPool::run, this).detach();
			} catch (...) {
				break;
			}
			nthreads++;
		}
		return nthreads;
	}
	void submit(std::function<void()> f)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			q.push_back(std::move(f));
		}
		cv.notify_one();
	}
};

/* The pool is never freed: its workers outlive static destruction, and the
 * library is linked -z nodelete (Makefile), so a dlclose() by a plugin host
 * cannot unmap the text they sleep in.  A forked child has none of the
 * parent's threads: its fork handler only drops the pointer (nothing that
 * allocates or locks may run there - the child of a multi-threaded process
 * is restricted to async-signal-safe calls), and the child's first use builds
 * a pool of its own. */
std::atomic<HostPool *> g_pool{nullptr};
std::mutex g_pool_mu;
std::once_flag g_pool_once;

HostPool *host_pool()
{
	HostPool *p = g_pool.load(std::memory_order_acquire);
	if (p)
		return p;
	std::call_once(g_pool_once, [] {
		(void)pthread_atfork(nullptr, nullptr,
				     [] {
					     g_pool.store(nullptr, std::memory_order_relaxed);
					     /* (a parent thread may have held it at the fork) */
					     new (&g_pool_mu) std::mutex;
				     });
	});
	std::lock_guard<std::mutex> lk(g_pool_mu);
	p = g_pool.load(std::memory_order_acquire);
	if (!p) {
		p = new HostPool;
		g_pool.store(p, std::memory_order_release);
	}
	return p;
}

struct Join {
	std::mutex m;
	std::condition_variable cv;
	size_t left = 0;
};

} /* namespace */

template <typename F> static void for_chunks_parallel(size_t lo, size_t hi, uint64_t bytes, F fn)
{
	size_t nt = (size_t)env_cfg().host_threads;
	if (bytes < ((uint64_t)2 << 20) || hi - lo < 2 * nt || nt < 2) {
		for (size_t k = lo; k < hi; k++)
			fn(k);
		return;
	}
	HostPool *pool = nullptr;
	try {
		pool = host_pool();
		if (pool->ensure((unsigned)nt - 1) == 0)
			pool = nullptr;
	} catch (...) {		/* nothing may unwind through the extern "C" entry points */
		pool = nullptr;
	}
	if (!pool) {
		for (size_t k = lo; k < hi; k++)
			fn(k);
		return;
	}
	const size_t per = (hi - lo + nt - 1) / nt;
	const size_t own_hi = lo + per < hi ? lo + per : hi;
	auto join = std::make_shared<Join>();
	size_t queued = 0;
	for (size_t t = 1; t < nt; t++) {
		const size_t a = lo + t * per, b = a + per < hi ? a + per : hi;
		if (a >= hi)
			break;
		{
			std::lock_guard<std::mutex> lk(join->m);
			join->left++;
		}
		try {
			pool->submit([=]() {
				for (size_t k = a; k < b; k++)
					fn(k);
				std::lock_guard<std::mutex> lk(join->m);
				if (--join->left == 0)
					join->cv.notify_one();
			});
			queued++;
		} catch (...) {
			{
				std::lock_guard<std::mutex> lk(join->m);
				join->left--;
			}
			for (size_t k = a; k < hi; k++)	/* the calling thread takes the rest */
				fn(k);
			break;
		}
	}
	for (size_t k = lo; k < own_hi; k++)
		fn(k);
	if (queued) {
		std::unique_lock<std::mutex> lk(join->m);
		join->cv.wait(lk, [&] { return join->left == 0; });
	}
}

int copy_in_packed(PinnedPair *pp, uint8_t *d_base, size_t n,
		   const void *const *in, const size_t *in_nbytes,
		   const uint64_t *off, hipStream_t st)
{
	{	/* pinned staging sized by the batch's span (two buffers alternate) */
		uint64_t span = n ? off[n - 1] + in_nbytes[n - 1] - off[0] : 0;
		if (!pp->ensure(span))
			return LIBDEFLATE_AMD_OOM;
	}
	/* a span goes through in at least four pieces, so that the DMA of one
	 * pinned buffer runs beside the packing of the other */
	const size_t lim = n > 1 ? std::min<size_t>(pp->cap, std::max<size_t>(
		(size_t)2 << 20, (size_t)(off[n - 1] + in_nbytes[n - 1] - off[0]) / 4)) : pp->cap;
	int b = 0;
	bool used[2] = { false, false };
	for (size_t i = 0; i < n;) {
		if (in_nbytes[i] > pp->cap) {	/* one huge chunk: straight from the caller */
			LDA_HIP_TRY(hipMemcpyAsync(d_base + off[i], in[i], in_nbytes[i],
						   hipMemcpyHostToDevice, st),
				    LIBDEFLATE_AMD_NO_DEVICE);
			i++;
			continue;
		}
		const uint64_t s0 = off[i];
		size_t j = i;
		while (j < n && (j == i || off[j] + in_nbytes[j] - s0 <= lim))
			j++;
		if (used[b])
			LDA_HIP_TRY(hipEventSynchronize(pp->ev[b]), LIBDEFLATE_AMD_NO_DEVICE);
		const uint64_t span = off[j - 1] + in_nbytes[j - 1] - s0;
		{
			uint8_t *dst = pp->buf[b];
			for_chunks_parallel(i, j, span, [=](size_t k) {
				if (in_nbytes[k])
					memcpy(dst + (off[k] - s0), in[k], in_nbytes[k]);
			});
		}
		if (span)
			LDA_HIP_TRY(hipMemcpyAsync(d_base + s0, pp->buf[b], span,
						   hipMemcpyHostToDevice, st),
				    LIBDEFLATE_AMD_NO_DEVICE);
		LDA_HIP_TRY(hipEventRecord(pp->ev[b], st), LIBDEFLATE_AMD_NO_DEVICE);
		used[b] = true;
		b ^= 1;
		i = j;
	}
	/* "returns when the slice is on the device": the kernels that read it run
	 * on ANOTHER stream, so everything queued on this one has to have landed -
	 * the packed slices (whose pinned buffers the caller's next step reuses),
	 * the chunks copied straight from the caller's memory, and whatever the
	 * caller queued in front (its descriptor upload) */
	LDA_HIP_TRY(hipStreamSynchronize(st), LIBDEFLATE_AMD_NO_DEVICE);
	return LIBDEFLATE_AMD_OK;
}

int copy_out_packed(PinnedPair *pp, const uint8_t *d_base, size_t n,
		    void *const *out, const uint64_t *nbytes,
		    const uint64_t *off, hipStream_t st)
{
	uint64_t span_all = 0;
	{
		uint64_t lo = 0, hi = 0;
		bool any = false;
		for (size_t i = 0; i < n; i++)
			if (nbytes[i]) {
				if (!any)
					lo = off[i];
				any = true;
				hi = off[i] + nbytes[i];
			}
		if (!pp->ensure(hi - lo))
			return LIBDEFLATE_AMD_OOM;
		span_all = hi - lo;
	}
	const size_t lim = n > 1 ? std::min<size_t>(pp->cap, std::max<size_t>((size_t)2 << 20,
										    (size_t)span_all / 4)) : pp->cap;
	struct slice { size_t i, j; uint64_t s0, span; int b; };
	auto next_slice = [&](size_t i, int b, slice *s) {
		while (i < n && nbytes[i] == 0)
			i++;
		s->i = s->j = i;
		s->b = b;
		s->span = 0;
		if (i >= n)
			return;
		s->s0 = off[i];
		size_t j = i;
		if (nbytes[i] > pp->cap) {
			s->j = i + 1;
			s->span = nbytes[i];
			return;
		}
		while (j < n && (j == i || off[j] + nbytes[j] - s->s0 <= lim)) {
			if (nbytes[j])
				s->span = off[j] + nbytes[j] - s->s0;
			j++;
		}
		s->j = j;
	};
	auto issue = [&](const slice &s) -> int {
		if (s.i >= n)
			return LIBDEFLATE_AMD_OK;
		if (s.span > pp->cap)	/* one huge chunk: straight to the caller */
			LDA_HIP_TRY(hipMemcpyAsync(out[s.i], d_base + s.s0, s.span,
						   hipMemcpyDeviceToHost, st),
				    LIBDEFLATE_AMD_NO_DEVICE);
		else
			LDA_HIP_TRY(hipMemcpyAsync(pp->buf[s.b], d_base + s.s0, s.span,
						   hipMemcpyDeviceToHost, st),
				    LIBDEFLATE_AMD_NO_DEVICE);
		LDA_HIP_TRY(hipEventRecord(pp->ev[s.b], st), LIBDEFLATE_AMD_NO_DEVICE);
		return LIBDEFLATE_AMD_OK;
	};
	slice cur, nxt;
	next_slice(0, 0, &cur);
	int rc = issue(cur);
	while (rc == LIBDEFLATE_AMD_OK && cur.i < n) {
		next_slice(cur.j, cur.b ^ 1, &nxt);
		rc = issue(nxt);	/* in flight while this one is unpacked */
		if (rc != LIBDEFLATE_AMD_OK)
			break;
		LDA_HIP_TRY(hipEventSynchronize(pp->ev[cur.b]), LIBDEFLATE_AMD_NO_DEVICE);
		if (cur.span <= pp->cap) {
			const uint8_t *src = pp->buf[cur.b];
			const uint64_t s0 = cur.s0;
			for_chunks_parallel(cur.i, cur.j, cur.span, [=](size_t k) {
				if (nbytes[k])
					memcpy(out[k], src + (off[k] - s0), nbytes[k]);
			});
		}
		cur = nxt;
	}
	return rc;
}

/* one contiguous host range <-> a device range through a pinned pair, cut into
 * pieces so that the packing threads share the memcpy and the DMA of one pinned
 * buffer runs beside the memcpy of the other: a slice is a quarter of the range
 * (at least 2 MiB) and has sixteen pieces of 64 KiB .. 1 MiB.  (Pieces of 1 MiB
 * whatever the range until round 6: a slice of 4 MiB - a quarter of a 16 MiB
 * stream - then had four pieces, fewer than for_chunks_parallel() asks for
 * before it wakes the threads, and ONE thread copied every slice: 0.7 ms of a
 * 2.4 ms call.) */
static size_t span_piece(size_t n)
{
	size_t slice = std::max<size_t>((size_t)2 << 20, n / 4), p = (size_t)64 << 10;
	while (p < ((size_t)1 << 20) && 16 * p < slice)
		p *= 2;
	return p;
}

int span_in(PinnedPair *pp, uint8_t *d_base, uint64_t d_off, const uint8_t *src, size_t n,
	    hipStream_t st, size_t piece)
{
	const size_t P = piece ? piece : span_piece(n), np = (n + P - 1) / P;
	std::vector<const void *> ins(np);
	std::vector<size_t> nb(np);
	std::vector<uint64_t> off(np);
	for (size_t i = 0; i < np; i++) {
		ins[i] = src + i * P;
		nb[i] = i + 1 < np ? P : n - i * P;
		off[i] = d_off + i * P;
	}
	return copy_in_packed(pp, d_base, np, ins.data(), nb.data(), off.data(), st);
}

int span_out(PinnedPair *pp, const uint8_t *d_base, uint64_t d_off, uint8_t *dst, size_t n,
	     hipStream_t st, size_t piece)
{
	const size_t P = piece ? piece : span_piece(n), np = (n + P - 1) / P;
	std::vector<void *> outs(np);
	std::vector<uint64_t> nb(np), off(np);
	for (size_t i = 0; i < np; i++) {
		outs[i] = dst + i * P;
		nb[i] = i + 1 < np ? P : n - i * P;
		off[i] = d_off + i * P;
	}
	return copy_out_packed(pp, d_base, np, outs.data(), nb.data(), off.data(), st);
}

#define MAX_DEVICES 16
static DeviceCtx g_ctx[MAX_DEVICES];
static std::mutex g_ctx_mu;

/* CRC-32 tables: see checksum_kernels.hip header for their meaning */
static void gen_crc_tables(uint32_t *tab /*17*256*/, uint32_t *xpow /*1024*/)
{
	const uint32_t poly = 0xEDB88320u;
	uint32_t t0[256];

	for (uint32_t b = 0; b < 256; b++) {
		uint32_t r = b;
		for (int k = 0; k < 8; k++)
			r = (r >> 1) ^ (poly & (0u - (r & 1u)));
		t0[b] = r;
	}
	/* S_k[b] = register after byte b and then (15-k)+1008 zero bytes */
	for (uint32_t b = 0; b < 256; b++) {
		uint32_t r = t0[b];
		int zeros = 0;
		for (int k = 15; k >= 0; k--) {
			int want = (15 - k) + 1008;
			for (; zeros < want; zeros++)
				r = t0[r & 0xFF] ^ (r >> 8);
			tab[k * 256 + b] = r;
		}
	}
	memcpy(tab + 16 * 256, t0, sizeof(t0));
	/* xpow[d] = x^(8d) mod P, x^0 at bit 31 */
	uint32_t x = 0x80000000u;
	for (int d = 0; d < 1024; d++) {
		xpow[d] = x;
		for (int k = 0; k < 8; k++)
			x = (x >> 1) ^ (poly & (0u - (x & 1u)));
	}
}

bool device_selfcheck(int num_cus, uint64_t out[5])
{
	const unsigned blocks = (unsigned)(num_cus > 0 ? num_cus : 256);
	const unsigned waves = 16 * blocks;
	const size_t region = lda_selfcheck_region_bytes();
	uint8_t *buf = nullptr;
	uint64_t *d_cnt = nullptr;

	memset(out, 0, 5 * sizeof(uint64_t));
	LDA_HIP_TRY(hipMalloc((void **)&buf, (size_t)waves * region + 64), false);
	d_cnt = (uint64_t *)(buf + (size_t)waves * region);
	hipError_t e = hipMemset(buf, 0, (size_t)waves * region + 64);
	if (e == hipSuccess) {
		/* one workgroup of 16 waves per CU hammering one LDS; 16 waves per
		 * CU storing and loading their own regions */
		hipLaunchKernelGGL(lda_selfcheck_lds_order_kernel, dim3(blocks), dim3(1024), 0, 0,
				   d_cnt);
		hipLaunchKernelGGL(lda_selfcheck_visibility_kernel, dim3(waves), dim3(64), 0, 0,
				   buf, d_cnt);
		e = hipGetLastError();
	}
	if (e == hipSuccess)
		e = hipMemcpy(out, d_cnt, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost);
	(void)hipFree(buf);
	if (e != hipSuccess) {
		set_error("hardware self-check: %s", hipGetErrorString(e));
		return false;
	}
	if (out[0] == 0 || out[3] == 0 || out[1] || out[4]) {
		set_error("hardware self-check failed: %llu of %llu LDS-atomic lanes out of lane "
			  "order, %llu of %llu loads did not see the wave's own store - this "
			  "device does not behave as the kernels need (deflate_kernel.hip "
			  "insert_tile(), inflate_kernel.hip par_round())",
			  (unsigned long long)out[1], (unsigned long long)out[0],
			  (unsigned long long)out[4], (unsigned long long)out[3]);
		return false;
	}
	return true;
}

DeviceGuard::DeviceGuard(int device)
{
	hipError_t e = hipGetDevice(&prev);

	if (e == hipSuccess && prev == device)
		return;
	e = hipSetDevice(device);
	if (e != hipSuccess) {
		set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
		good = false;
		return;
	}
	switched = true;
}

DeviceGuard::~DeviceGuard()
{
	if (switched && prev >= 0)
		(void)hipSetDevice(prev);
}

DeviceCtx *device_ctx()
{
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);

	if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
		set_error("hipGetDevice: %s", hipGetErrorString(e));
		return nullptr;
	}
	std::lock_guard<std::mutex> lk(g_ctx_mu);
	DeviceCtx *c = &g_ctx[dev];
	if (c->device == dev)
		return c;

	hipDeviceProp_t prop;
	LDA_HIP_TRY(hipGetDeviceProperties(&prop, dev), nullptr);
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		set_error("device %d is %s, this build targets gfx950 only",
			  dev, prop.gcnArchName);
		return nullptr;
	}
	static uint32_t tab[LDA_CRC_TABLE_WORDS], xpow[LDA_CRC_XPOW_WORDS];
	gen_crc_tables(tab, xpow);
	/* one allocation for both tables: a failure half way leaves nothing
	 * behind, and the next call tries again from the start */
	uint32_t *d_tab = nullptr;
	LDA_HIP_TRY(hipMalloc((void **)&d_tab, sizeof(tab) + sizeof(xpow)), nullptr);
	hipError_t ce = hipMemcpy(d_tab, tab, sizeof(tab), hipMemcpyHostToDevice);
	if (ce == hipSuccess)
		ce = hipMemcpy(d_tab + LDA_CRC_TABLE_WORDS, xpow, sizeof(xpow),
			       hipMemcpyHostToDevice);
	if (ce != hipSuccess) {
		(void)hipFree(d_tab);
		set_error("hipMemcpy(CRC tables): %s", hipGetErrorString(ce));
		return nullptr;
	}
	/* the hardware behaves as the kernels need?  (once per device; a device
	 * that does not is refused: the allocators return NULL with the reason
	 * in libdeflate_amd_last_error()) */
	if (!env_cfg().no_selfcheck) {
		uint64_t sc[5];
		if (!device_selfcheck(prop.multiProcessorCount, sc)) {
			(void)hipFree(d_tab);
			return nullptr;
		}
	}
	c->d_crc_tables = d_tab;
	c->d_crc_xpow8 = d_tab + LDA_CRC_TABLE_WORDS;
	c->num_cus = prop.multiProcessorCount;
	c->device = dev;
	return c;
}

void *stage_reserve(DeviceCtx *ctx, size_t nbytes)
{
	if (nbytes <= ctx->stage_cap)
		return ctx->d_stage;
	size_t cap = align_up(nbytes + nbytes / 4 + 4096, 4096);
	void *p = nullptr;
	if (ctx->d_stage)
		(void)hipFree(ctx->d_stage);
	ctx->d_stage = nullptr;
	ctx->stage_cap = 0;
	hipError_t e = hipMalloc(&p, cap);
	if (e != hipSuccess) {
		set_error("hipMalloc(%zu): %s", cap, hipGetErrorString(e));
		return nullptr;
	}
	ctx->d_stage = p;
	ctx->stage_cap = cap;
	return p;
}

} /* namespace lda */

/* runs the hardware self-check again on the calling thread's current device */
extern "C" LIBDEFLATEAPI int libdeflate_amd_selfcheck(uint64_t *out /* [5] */)
{
	lda::DeviceCtx *c = lda::device_ctx();
	uint64_t tmp[5];

	if (!c)
		return LIBDEFLATE_AMD_NO_DEVICE;
	const bool ok = lda::device_selfcheck(c->num_cus, out ? out : tmp);
	return ok ? LIBDEFLATE_AMD_OK : LIBDEFLATE_AMD_NO_DEVICE;
}

extern "C" LIBDEFLATEAPI int libdeflate_amd_device_ready(void)
{
	return lda::device_ctx() ? LIBDEFLATE_AMD_OK : LIBDEFLATE_AMD_NO_DEVICE;
}

/* the tuning switches are read when the library is loaded; a process that
 * changes them afterwards (the tests do) asks for a re-read.  Not thread safe
 * against calls in flight. */
extern "C" LIBDEFLATEAPI void libdeflate_amd_reload_env(void)
{
	lda::g_env = lda::read_env();
}

extern "C" LIBDEFLATEAPI const char *libdeflate_amd_last_error(void)
{
	return lda::g_err;
}
