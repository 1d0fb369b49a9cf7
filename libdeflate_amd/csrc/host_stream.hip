/*
 * host_stream.hip - ONE large stream on many waves: the host side of
 * inflate_stream.hip (find -> plan -> count -> chain -> decode -> window ->
 * resolve -> checksum; see that file's header for what each step is).
 *
 * Entered from the single-buffer libdeflate_{deflate,zlib,gzip}_decompress[_ex]
 * calls (host_decompress.hip) for streams of LDA_STREAM_PAR_MIN bytes and up -
 * what programs/gzip.c:187-303 and programs/benchmark.c:543-544 hand to the
 * library.  It only ever ANSWERS for a clean success (or a footer that does not
 * match a cleanly decoded stream): anything else - an invalid header, a chain
 * that does not close, an output that does not fit or does not fill - is left
 * to the sequential kernel, which follows the reference's result codes bit for
 * bit, so the codes cannot depend on which path ran.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <unordered_map>
#include <vector>

#include "host_objects.h"
#include "kernels.h"
#include "stream_kernels.h"

namespace lda {

static thread_local uint64_t g_stats[LIBDEFLATE_AMD_STREAM_STATS];

#define ST_TRY(expr)                                                          \
	do {                                                                  \
		hipError_t e_ = (expr);                                       \
		if (e_ != hipSuccess) {                                       \
			set_error("%s: %s", #expr, hipGetErrorString(e_));    \
			return false;                                         \
		}                                                             \
	} while (0)

enum { WHY_OK = 0, WHY_DISABLED, WHY_HEADER, WHY_CHAIN, WHY_ERRCHUNK, WHY_NOFINAL,
       WHY_SPACE, WHY_FILL, WHY_DECODE, WHY_DEVICE, WHY_REPAIRS };

struct planned {
	lda_stream_chunk c;
	uint64_t at;	/* nominal start: hdr_bit (HEADER) or target_bit (WARM) */
};

/* gzip / zlib container: offset of the raw stream and the footer's size; false
 * if the sequential path should look at it (lib/gzip_decompress.c:45-107,
 * lib/zlib_decompress.c:45-72) */
static bool container(int format, const uint8_t *in, size_t n, size_t *hdr, size_t *ftr)
{
	*hdr = *ftr = 0;
	if (format == LIBDEFLATE_AMD_DEFLATE)
		return true;
	if (format == LIBDEFLATE_AMD_ZLIB) {
		if (n < 6)
			return false;
		const uint32_t h = ((uint32_t)in[0] << 8) | in[1];
		if (h % 31 || ((h >> 8) & 0xF) != 8 || (h >> 12) > 7 || ((h >> 5) & 1))
			return false;
		*hdr = 2;
		*ftr = 4;
		return true;
	}
	if (n < 18 || in[0] != 0x1F || in[1] != 0x8B || in[2] != 8 || (in[3] & 0xE0))
		return false;
	const uint32_t flg = in[3];
	size_t p = 10;
	if (flg & 0x04) {
		const size_t xlen = in[p] | ((size_t)in[p + 1] << 8);
		p += 2;
		if (n - p < xlen + 8)
			return false;
		p += xlen;
	}
	for (int k = 0; k < 2; k++)
		if (flg & (k ? 0x10 : 0x08)) {
			while (in[p++] != 0 && p != n)
				;
			if (n - p < 8)
				return false;
		}
	if (flg & 0x02) {
		p += 2;
		if (n - p < 8)
			return false;
	}
	*hdr = p;
	*ftr = 8;
	return true;
}

static bool launch_count(hipStream_t st, uint32_t n, const lda_stream_chunk *d_chunks,
			 lda_stream_res *d_res, const uint8_t *d_raw, uint64_t raw_n,
			 const uint8_t *d_hlens, const uint32_t *d_hinfo, uint16_t *d_hints)
{
	hipLaunchKernelGGL(lda_stream_count_kernel, dim3(n), dim3(64), lda_stream_chunk_lds(),
			   st, n, d_chunks, d_res, d_raw, raw_n, (uint32_t *)NULL, d_hlens, d_hinfo,
			   d_hints);
	ST_TRY(hipGetLastError());
	return true;
}

/*
 * true: *res (and on success *ain / *aout, the output in `out`) are final.
 * false: not answered here - the caller takes the sequential path (the reason
 * is in the stats; a device failure is also in last_error).
 */
bool decompress_stream_parallel(struct libdeflate_decompressor *d, int format,
				const uint8_t *in, size_t in_nbytes, uint8_t *out,
				size_t out_avail, bool exact_fill, int32_t *res,
				size_t *ain, size_t *aout)
{
	uint64_t *S = g_stats;
	memset(g_stats, 0, sizeof(g_stats));
	/* host-side phase clock: S[8..13] = microseconds of copy in, find, count +
	 * chain, decode + window + resolve, checksum, copy out */
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](int slot) {
		const auto now = std::chrono::steady_clock::now();
		S[slot] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(now - t_last).count();
		t_last = now;
	};
	/* LDA_STREAM_DEBUG: where the host's time goes inside a phase (stderr) */
	const bool debug = getenv("LDA_STREAM_DEBUG") != nullptr;
	auto dbg = [&](const char *what) {
		if (debug)
			fprintf(stderr, "  %-28s +%lld us\n", what,
				(long long)std::chrono::duration_cast<std::chrono::microseconds>(
					std::chrono::steady_clock::now() - t_last).count());
	};
	DeviceCtx *ctx = device_ctx();
	const EnvCfg &env = env_cfg();
	if (!ctx || env.no_stream_par || in_nbytes < env.stream_par_min) {
		S[1] = WHY_DISABLED;
		return false;
	}
	size_t hdr, ftr;
	if (!container(format, in, in_nbytes, &hdr, &ftr)) {
		S[1] = WHY_HEADER;
		return false;
	}
	const uint64_t raw_n = in_nbytes - hdr - ftr;
	const uint64_t raw_bits = 8 * raw_n;
	if (raw_n < 8) {
		S[1] = WHY_HEADER;
		return false;
	}
	S[1] = WHY_DEVICE;	/* until something better is known */
	if (!d->streams.ensure())
		return false;
	hipStream_t s_copy = d->streams.copy, s_comp = d->streams.comp;
	/* The small transfers of every phase (descriptors down, results back) go
	 * through pinned memory: from and to pageable memory each of them would
	 * be staged by the runtime and cost a host-blocking round trip of its own.
	 * pin_phase() sizes the arena (nothing may be in flight), up() / back()
	 * queue a copy, pin_sync() waits and delivers what came back. */
	uint8_t *pin_base = nullptr;
	size_t pin_used = 0;
	struct pending_t { void *dst; const void *src; size_t n; };
	std::vector<pending_t> pin_pending;
	auto pin_phase = [&](size_t need) -> bool {
		pin_base = (uint8_t *)d->meta.ensure(need + 1024);
		pin_used = 0;
		return pin_base != nullptr;
	};
	auto up = [&](void *dev, const void *src, size_t n) -> hipError_t {
		uint8_t *q = pin_base + pin_used;
		pin_used += align_up(n, 64);
		memcpy(q, src, n);
		return hipMemcpyAsync(dev, q, n, hipMemcpyHostToDevice, s_comp);
	};
	auto back = [&](void *dst, const void *dev, size_t n) -> hipError_t {
		uint8_t *q = pin_base + pin_used;
		pin_used += align_up(n, 64);
		pin_pending.push_back({ dst, q, n });
		return hipMemcpyAsync(q, dev, n, hipMemcpyDeviceToHost, s_comp);
	};
	auto pin_sync = [&]() -> hipError_t {
		const hipError_t e = hipStreamSynchronize(s_comp);
		for (const pending_t &c : pin_pending)
			memcpy(c.dst, c.src, c.n);
		pin_pending.clear();
		pin_used = 0;
		return e;
	};

	/* the host has the stream too: bits of it, for what it can decide itself */
	const uint8_t *raw = in + hdr;
	auto peek = [&](uint64_t bit, unsigned n) -> uint32_t {	/* n <= 24; zeros past the end */
		uint32_t v = 0;
		const uint64_t b0 = bit >> 3;
		if (b0 + 4 <= raw_n)
			memcpy(&v, raw + b0, 4);	/* (little-endian host, as the HIP runtime's) */
		else
			for (unsigned k = 0; k < 4; k++)
				if (b0 + k < raw_n)
					v |= (uint32_t)raw[b0 + k] << (8 * k);
		return (v >> (bit & 7)) & ((1u << n) - 1);
	};
	/*
	 * A RUN OF STORED BLOCKS from the block boundary `p` on, walked by the
	 * host (5 header bytes per block of up to 65535: lib/decompress_template.h:
	 * 247-285): the finder does not look for stored blocks, and a chunk that
	 * walked such a run alone copied it alone - a level-0 file at one wave's
	 * speed.  Every stored block (several small ones together, up to `group`
	 * bits of input) becomes a chunk of its own whose result is known without
	 * a count pass; the decode pass copies them side by side.  Stops in front
	 * of the first block that is not stored, is not wholly inside the first
	 * `dev_bytes` of the stream (what the kernels can read), or is invalid
	 * (the kernels - and after them the sequential path - say what the
	 * reference says about that one).  Returns the bit it stopped at.
	 */
	auto walk_stored = [&](uint64_t p, uint64_t dev_bytes, uint64_t group,
			       std::vector<lda_stream_chunk> &oc, std::vector<lda_stream_res> &orr,
			       bool *fin_ret) -> uint64_t {
		*fin_ret = false;
		for (;;) {
			lda_stream_chunk c = {};
			lda_stream_res r = {};
			c.kind = LDA_CHUNK_HEADER;
			c.hdr_bit = c.start_bit = c.target_bit = p;
			r.start_bit = p;
			uint64_t q = p;
			bool fin = false;
			while (q + 3 <= raw_bits && !fin && q - p < group) {
				const uint32_t h = peek(q, 3);
				if ((h >> 1) != 0)
					break;
				const uint64_t bp = (q + 3 + 7) >> 3;
				if (bp + 4 > raw_n)
					break;
				const uint32_t len = raw[bp] | ((uint32_t)raw[bp + 1] << 8);
				const uint32_t nlen = raw[bp + 2] | ((uint32_t)raw[bp + 3] << 8);
				if (len != (nlen ^ 0xFFFFu) || bp + 4 + len > raw_n || bp + 4 + len > dev_bytes)
					break;
				r.nout += len;
				q = 8 * (bp + 4 + len);
				fin = h & 1;
			}
			if (q == p)
				return p;
			c.limit_bit = q;
			r.end_bit = r.end_hdr_bit = q;
			r.status = fin ? LDA_STREAM_FINAL : LDA_STREAM_OK;
			r.flags = LDA_RES_BOUNDARY;
			oc.push_back(c);
			orr.push_back(r);
			S[15]++;
			p = q;
			if (fin) {
				*fin_ret = true;
				return p;
			}
		}
	};

	/*
	 * Is the dynamic block whose header starts at bit `hb` one whose parses
	 * do not fall in step - literal codewords of (nearly) ONE length, the
	 * block a compressor writes over incompressible bytes inside a stream of
	 * other data?  The host reads the header itself (the precode and the
	 * literal lengths, lib/deflate_decompress.c:1227-1359 restated for the
	 * first 256 symbols; a few hundred bits): returns the longest literal
	 * codeword when the literals of one length fill 98 % of the code
	 * space, 0 otherwise (or when the header is not a
	 * valid dynamic one: the kernels say what is wrong with it).  Only the
	 * plan depends on the answer, never the result.
	 */
	auto one_length_code = [&](uint64_t hb, uint64_t *first_token) -> uint32_t {
		static const uint8_t perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		uint64_t p = hb;
		if (p + 17 + 19 * 3 > raw_bits || ((peek(p, 3) >> 1) & 3) != 2)
			return 0;
		const uint32_t nl = 257 + peek(p + 3, 5), nd = 1 + peek(p + 8, 5), nc = 4 + peek(p + 13, 4);
		p += 17;
		uint8_t pl[19] = { 0 };
		for (uint32_t i = 0; i < nc; i++, p += 3)
			pl[perm[i]] = (uint8_t)peek(p, 3);
		uint32_t cnt[8] = { 0 }, first[8] = { 0 }, base[8] = { 0 };
		uint8_t sorted[19];
		uint32_t ns = 0;
		for (uint32_t len = 1; len < 8; len++)
			for (uint32_t sy = 0; sy < 19; sy++)
				if (pl[sy] == len) {
					sorted[ns++] = (uint8_t)sy;
					cnt[len]++;
				}
		for (uint32_t len = 1, code = 0; len < 8; len++) {
			code = (code + cnt[len - 1]) << 1;
			first[len] = code;
			base[len] = len > 1 ? base[len - 1] + cnt[len - 1] : 0;
		}
		/* (all lengths are read, the offsets' too: behind them is the block's
		 * first token, where the first of the exact starts lies) */
		uint8_t lens[320 + 138];
		uint32_t n = 0;
		const uint32_t want = nl < 256 ? nl : 256, total = nl + nd;
		/* (an ordinary block is told apart after two dozen lengths: under
		 * three quarters of the literals seen so far on one length - a stream
		 * of 1 GiB has 3600 blocks, and parsing every header in full was 2 ms
		 * of its call) */
		uint32_t seen[16] = { 0 }, nz = 0, most = 0;
		while (n < total) {
			if (n < want && nz >= 24 && 4 * most < 3 * nz)
				return 0;
			if (p + 32 > raw_bits)
				return 0;
			uint32_t c = 0, sy = 99;
			const uint32_t v7 = peek(p, 7);
			for (uint32_t len = 1; len < 8; len++) {
				c = (c << 1) | ((v7 >> (len - 1)) & 1);
				if (c - first[len] < cnt[len]) {
					sy = sorted[base[len] + c - first[len]];
					p += len;
					break;
				}
			}
			if (sy < 16) {
				lens[n++] = (uint8_t)sy;
				if (sy && n <= want) {
					nz++;
					most = std::max(most, ++seen[sy]);
				}
			} else if (sy == 16) {
				if (!n)
					return 0;
				const uint32_t r = 3 + peek(p, 2);
				p += 2;
				for (uint32_t k = 0; k < r; k++, n++)
					lens[n] = lens[n - 1];
				if (lens[n - 1] && n <= want) {
					nz += r;
					seen[lens[n - 1]] += r;
					most = std::max(most, seen[lens[n - 1]]);
				}
			} else if (sy == 17 || sy == 18) {
				const uint32_t r = sy == 17 ? 3 + peek(p, 3) : 11 + peek(p, 7);
				p += sy == 17 ? 3 : 7;
				for (uint32_t k = 0; k < r; k++)
					lens[n++] = 0;
			} else {
				return 0;
			}
		}
		uint32_t hist[16] = { 0 }, hi = 0, top = 1;
		for (uint32_t sy = 0; sy < want; sy++)
			hist[lens[sy]]++;
		for (uint32_t len = 1; len < 16; len++) {
			if (hist[len])
				hi = len;
			if (hist[len] > hist[top])
				top = len;
		}
		/* 98 % of the code space at one length.  (Two parses that are d bits
		 * apart drift by a bit where one of them meets a codeword of another
		 * length: with a share p of those they meet after ~10 / 2p tokens -
		 * at 5 % well inside the 1 KiB warm-up, which then costs a ninth of
		 * what the exact starts cost; at 0.5 % - 256 literals of 8 bits and
		 * what a compressor squeezes in beside them - in a thousand.) */
		if (n != total || p >= raw_bits)
			return 0;
		*first_token = p;
		return hist[top] >= 32 && hi <= 11 && 50 * hist[top] >= 49 * (1u << top) ? hi : 0;
	};

	/*
	 * The input is taken in WINDOWS (4 to 16 MiB of it, then four times as much
	 * each time, up to all of it): copied to the device, searched for block starts,
	 * planned, counted and chained - and when the chain reaches the stream's
	 * final block inside a window, the rest of the input is never touched.  The
	 * reference's callers hand the decompressor everything that is left of a
	 * file (programs/gzip.c:236-299 loops over the members of a .gz that way):
	 * without windows every call on a multi-member file would copy and search
	 * the whole remainder.  A window that ends before the final block hands its
	 * last accepted state (position, governing header) to the next one.
	 */
	/* (the first window by the output space: a stream rarely takes more input
	 * than half of what it produces, so one that fits the caller's buffer
	 * usually ends inside a window of out_avail / 2) */
	const size_t W0 = env.stream_window ? env.stream_window :
			  std::min<size_t>(std::max<size_t>(out_avail / 2, (size_t)4 << 20), (size_t)16 << 20);
	const size_t in_at = 64;
	/* device memory follows the windows, not the caller's buffer: the input
	 * copy grows with them (a regrown buffer is filled again from the start:
	 * a quarter more bytes copied at worst), the finder's queues are sized by
	 * the window searched */
	uint8_t *sin = nullptr, *d_raw = nullptr;
	uint64_t *d_queue = nullptr, *d_cand = nullptr;
	/* The headers the finder accepts are parsed ONCE each, by a wave of their
	 * own, beside the host's planning (lda_stream_hdr_cache_kernel on the copy
	 * stream, which has nothing to do then): slot i holds the code lengths of
	 * candidate i of the current window, and every chunk at or inside that
	 * block takes them from there instead of parsing the header again (one
	 * lane's loop: 40 us of every chunk of the count pass and of the decode
	 * pass).  hdr_bit -> slot + 1: */
	uint8_t *d_hlens = (uint8_t *)d->shdr.reserve((size_t)LDA_STREAM_HDR_SLOTS * (320 + 16) + 64);
	if (!d_hlens)
		return false;
	uint32_t *d_hinfo = (uint32_t *)(d_hlens + (size_t)LDA_STREAM_HDR_SLOTS * 320);
	uint16_t *d_hints = nullptr;	/* the last window's rows of lane starts (chunk.hint) */
	std::unordered_map<uint64_t, uint32_t> hdr_slot;
	auto cache_of = [&](uint64_t hdr_bit) -> uint32_t {
		const auto it = hdr_slot.find(hdr_bit);
		return it == hdr_slot.end() ? 0 : it->second;
	};
	uint32_t *d_cnt = nullptr;	/* [0] queue, [1] candidates, [2] error flag */
	uint32_t qcap = 0, ccap = 0;

	std::vector<lda_stream_chunk> acc;	/* accepted chunks, exact starts */
	std::vector<lda_stream_res> accr;
	/* where the next window's first chunk starts */
	lda_stream_chunk carry = {};
	carry.kind = LDA_CHUNK_HEADER;
	/* the state carried in lies inside a static block (carry.hdr_bit ==
	 * LDA_HDR_STATIC): is that block the stream's last?  (the chunks planned
	 * under the static codes cannot know, see stream_kernels.h) */
	bool carry_gf = false;
	size_t copied = 0;	/* bytes of the caller's buffer on the device */
	uint64_t dev_n = 0;	/* raw bytes the kernels may read (the last window's) */
	bool final_seen = false;
	for (size_t W = W0; !final_seen; W = W < ((size_t)1 << 40) ? W * 4 : W) {
		/* (the parsed headers are the current window's) */
		hdr_slot.clear();
		for (lda_stream_chunk &c : acc)
			c.hdr_cache = c.hint = 0;
		d_hints = nullptr;
		bool cache_queued = false;
		/* ---- this window's input ---- */
		const size_t upto = std::min<size_t>(in_nbytes, std::max(copied, hdr) + W);
		if (!sin || in_at + upto + 64 > d->sin.cap) {
			sin = (uint8_t *)d->sin.reserve(in_at + std::min<size_t>(in_nbytes, 2 * upto) + 64);
			if (!sin)
				return false;
			d_raw = sin + in_at + hdr;
			copied = 0;
		}
		if (upto > copied) {
			if (span_in(&d->pinned, sin, in_at + copied, in + copied, upto - copied, s_copy) !=
			    LIBDEFLATE_AMD_OK)
				return false;
			copied = upto;
		}
		const bool whole = copied == in_nbytes;
		/* raw bytes the kernels may read; chunks of a partial window end a few
		 * KiB in front of that (a round stages up to 3 KiB ahead, a header 704
		 * bytes) */
		/* (never the footer: a partial window that ends inside it would let a
		 * run of stored blocks be walked past the end of the raw stream, which
		 * the reference - it hands the decoder in_nbytes - hdr - ftr bytes -
		 * rejects) */
		const uint64_t win_n = whole ? raw_n : std::min<uint64_t>(copied - hdr, raw_n);
		dev_n = win_n;
		const uint64_t R1 = whole ? raw_bits : 8 * (win_n > 8192 ? win_n - 8192 : 0);
		S[14]++;
		/* stored blocks at the carried-in boundary: the host's */
		if (carry.kind == LDA_CHUNK_HEADER) {
			const uint64_t group = 8 * (uint64_t)(env.stream_chunk ? env.stream_chunk : 16384);
			bool fin = false;
			const uint64_t q = walk_stored(carry.start_bit, win_n, group, acc, accr, &fin);
			if (fin) {
				final_seen = true;
				break;
			}
			if (q != carry.start_bit) {
				carry = lda_stream_chunk();
				carry.kind = LDA_CHUNK_HEADER;
				carry.hdr_bit = carry.start_bit = carry.target_bit = q;
			}
		}
		if (R1 <= carry.start_bit + 4096 && !whole)
			continue;
		lap(8);
		/* ---- block starts in [carry, R1) ---- */
		std::vector<uint64_t> cands;
		const uint64_t fb0 = carry.start_bit & ~(uint64_t)7;
		const uint64_t nbits = R1 > 80 ? R1 - 80 : 0;	/* a header needs its bits */
		{
			/* about one offset in 500 passes the first filter on compressed
			 * data and a real block is rarely under a few hundred bits; a
			 * queue that overflows only loses entry points */
			const uint64_t span = nbits > fb0 ? nbits - fb0 : 0;
			qcap = (uint32_t)std::min<uint64_t>(span / 128 + 4096, 1u << 28);
			ccap = (uint32_t)std::min<uint64_t>(span / 512 + 4096, 1u << 26);
			uint8_t *sq = (uint8_t *)d->squeue.reserve(((size_t)qcap + ccap) * 8 + 128);
			if (!sq)
				return false;
			d_queue = (uint64_t *)sq;
			d_cand = d_queue + qcap;
			d_cnt = (uint32_t *)(d_cand + ccap);
		}
		ST_TRY(hipMemsetAsync(d_cnt, 0, 16, s_comp));
		if (nbits > fb0) {
			for (uint64_t b0 = fb0; b0 < nbits; b0 += 1ull << 31) {
				const uint64_t nb = std::min<uint64_t>(nbits - b0, 1ull << 31);
				/* (a workgroup of four waves takes 16 windows of 4 x 64 bytes) */
				hipLaunchKernelGGL(lda_stream_find_a_kernel, dim3((unsigned)((nb + 32767) / 32768)),
						   dim3(256), 0, s_comp, d_raw, win_n, b0, nbits, d_queue, d_cnt, qcap);
			}
			hipLaunchKernelGGL(lda_stream_find_b_kernel,
					   dim3(std::min<unsigned>((qcap + 63) / 64, 8u * (unsigned)ctx->num_cus)),
					   dim3(64), lda_stream_find_b_lds(),
					   s_comp, d_raw, win_n, d_queue, d_cnt, qcap, d_cand, d_cnt + 1, ccap);
			ST_TRY(hipGetLastError());
			ST_TRY(hipEventRecord(d->streams.mark, s_comp));
			ST_TRY(hipStreamWaitEvent(s_copy, d->streams.mark, 0));
			hipLaunchKernelGGL(lda_stream_hdr_cache_kernel, dim3(8u * (unsigned)ctx->num_cus), dim3(64),
					   lda_stream_hdr_cache_lds(), s_copy, d_raw, win_n, d_cand, d_cnt + 1,
					   std::min<uint32_t>(ccap, LDA_STREAM_HDR_SLOTS), d_hlens, d_hinfo);
			ST_TRY(hipGetLastError());
			ST_TRY(hipEventRecord(d->streams.mark2, s_copy));
			cache_queued = true;
			/* the counts and the first candidates in one round trip (a stream
			 * of a few MiB has a few hundred) */
			const uint32_t first = std::min<uint32_t>(ccap, 2048);
			uint32_t cnt[2];
			cands.resize(first);
			if (!pin_phase(64 + (size_t)first * 8))
				return false;
			ST_TRY(back(cnt, d_cnt, 8));
			ST_TRY(back(cands.data(), d_cand, (size_t)first * 8));
			ST_TRY(pin_sync());
			const uint32_t nc = std::min(cnt[1], ccap);
			cands.resize(nc);
			if (nc > first) {
				if (!pin_phase((size_t)(nc - first) * 8))
					return false;
				ST_TRY(back(cands.data() + first, d_cand + first, (size_t)(nc - first) * 8));
				ST_TRY(pin_sync());
			}
			for (uint32_t i = 0; i < nc && i < LDA_STREAM_HDR_SLOTS; i++)
				hdr_slot.emplace(cands[i], i + 1);
			std::sort(cands.begin(), cands.end());
			S[2] += cnt[0];
			S[3] += nc;
		}
		lap(9);

		/* ---- plan ----
		 * chunks of a few KiB of input: up to about two thousand for a window
		 * that has them (a wave slot each on 256 CUs, one launch of the decode
		 * pass), never under 2 KiB (the warm-up in front of an inner chunk is
		 * 1 KiB) */
		uint64_t T = env.stream_chunk ? (uint64_t)env.stream_chunk :
						(R1 - carry.start_bit) / 8 / 1536;
		T = 8 * std::min<uint64_t>(std::max<uint64_t>(T, 2048), 65536);
		const uint64_t OV = 8192, HDRSAFE = 4608;
		std::vector<planned> plan;
		plan.reserve(4096 + cands.size() * 8);
		uint32_t nexact = 0;	/* chunks planned at exact starts (blocks of one codeword length) */
		/* a block (or, for the carried-in state, what is left of one): its
		 * first chunk, then inner chunks up to `next` */
		/* (`inner`: the block's tables are known without looking - from its
		 * header at first.hdr_bit, or the static codes', `under` =
		 * LDA_HDR_STATIC) */
		auto add_block = [&](const lda_stream_chunk &first, uint64_t next, bool inner,
				     uint64_t under) {
			planned p = {};
			p.c = first;
			p.at = first.start_bit;
			plan.push_back(p);
			if (!inner)
				return;
			const uint64_t start = first.start_bit;
			/* a block of one codeword length (one_length_code()): no warm-up
			 * falls in step there, and a count pass that walks on such a
			 * block is slow (ten parses per piece, par_phase_starts()).  Its
			 * inner chunks are small - 2 KiB of input - and start EXACTLY,
			 * at every bit a literal that overhangs the planned start can
			 * end at: P, P + 1, .. P + longest literal codeword.  One of them
			 * is the true parse; the chain finds it by its key, in the first
			 * count pass.  (A match across P is not covered: a repair.) */
			uint64_t tok0 = 0;
			const uint32_t hi = first.kind == LDA_CHUNK_HEADER && !env.stream_chunk ?
						    one_length_code(first.hdr_bit, &tok0) : 0;
			/* (the starts of one position are counted together by ONE wave,
			 * phase_count() of inflate_stream.hip, when the chunk is one
			 * round of input: 1.5 x TN <= 64 pieces of 384 bits) */
			const uint64_t TN = 15360;
			if (hi && nexact + ((next - start) / TN + 1) * (hi + 1) <= 65536) {
				/* (the first of them at the block's first token - the host has
				 * read the header to its end -, so that the chunk that reads
				 * the header holds no tokens: as the only chunk of the block
				 * counted alone it was the only one the decode pass had no
				 * starts for, 1.1 M cycles against 0.4.  The other positions
				 * stay where they were.) */
				const uint64_t P1 = start + TN;
				const bool more = P1 + TN / 2 <= next;
				const uint64_t lim0 = more ? P1 : next;
				if (tok0 > start && tok0 + hi + 2 < lim0 && lim0 - tok0 <= 24000)
					for (uint32_t j = 0; j <= hi; j++) {
						planned q = {};
						q.c.kind = LDA_CHUNK_EXACT;
						q.c.hdr_bit = under;
						q.c.start_bit = q.c.target_bit = tok0 + j;
						q.c.phases = j ? ~0u : hi + 1;
						q.at = tok0;
						plan.push_back(q);
						nexact++;
					}
				for (uint64_t P = start + TN; P + TN / 2 <= next; P += TN)
					for (uint32_t j = 0; j <= hi; j++) {
						planned q = {};
						q.c.kind = LDA_CHUNK_EXACT;
						q.c.hdr_bit = under;
						q.c.start_bit = q.c.target_bit = P + j;
						q.c.phases = j ? ~0u : hi + 1;
						q.at = P;
						plan.push_back(q);
						nexact++;
					}
				return;
			}
			const uint64_t safe = first.kind == LDA_CHUNK_HEADER ? start + HDRSAFE : start;
			/* (the block in equal parts of at most T: with steps of T and what
			 * is left added to the last, a block's last chunk was up to 1.5 T -
			 * and the count and decode launches last as long as their longest
			 * chunk) */
			/* (not the window's last block: a chunk that starts within T of
			 * the end of the input runs its last rounds through the
			 * sequential tail code, and one that close to the end was the
			 * slowest chunk of the count launch by a factor of two) */
			const uint64_t blen = next > start ? next - start : 0;
			const uint64_t nparts = std::max<uint64_t>(1, (blen + T - 1) / T);
			const uint64_t step = next == R1 ? T : std::max<uint64_t>(T / 2, blen / nparts);
			for (uint64_t P = start + step; P + step / 2 <= next; P += step) {
				uint64_t ws = P > OV ? P - OV : 0;
				if (ws < safe)
					ws = safe;
				if (ws + OV / 4 > P)
					continue;
				planned q = {};
				q.c.kind = LDA_CHUNK_WARM;
				q.c.hdr_bit = under;
				q.c.start_bit = ws;
				q.c.target_bit = P;
				q.at = P;
				plan.push_back(q);
			}
		};
		{
			/* block starts: the carried-in state, then the candidates behind
			 * it.  A block's inner chunks end at the next candidate whatever
			 * becomes of it; a small block close behind a chunk start gets no
			 * chunk of its own (the chunk in front of it walks through), so
			 * chunk starts are at least T / 8 apart however small the blocks
			 * are */
			std::vector<uint64_t> cs;
			bool carry_dynamic = carry.kind != LDA_CHUNK_HEADER;	/* inside a Huffman block */
			for (uint64_t c : cands) {
				if (c == carry.start_bit && carry.kind == LDA_CHUNK_HEADER)
					carry_dynamic = true;
				else if (c > carry.start_bit)
					cs.push_back(c);
			}
			uint64_t last_at = carry.start_bit;
			/* a STATIC block at the carried-in state (the host sees the
			 * header, or the state says so): chunks under the static
			 * codes up to the next candidate.  They stop at the block's
			 * end; what follows there is found by the chain (repairs). */
			const bool carry_static =
				carry.kind == LDA_CHUNK_HEADER ? !carry_dynamic && (peek(carry.start_bit, 3) >> 1) == 1 :
								 carry.hdr_bit == LDA_HDR_STATIC;
			add_block(carry, cs.empty() ? R1 : cs[0], carry_dynamic || carry_static,
				  carry_static ? LDA_HDR_STATIC : carry.hdr_bit);
			for (size_t i = 0; i < cs.size(); i++) {
				const uint64_t next = i + 1 < cs.size() ? cs[i + 1] : R1;
				if (cs[i] - last_at < T / 8 && next - cs[i] < T / 2)
					continue;
				lda_stream_chunk b = {};
				b.kind = LDA_CHUNK_HEADER;
				b.hdr_bit = b.start_bit = b.target_bit = cs[i];
				add_block(b, next, true, b.hdr_bit);
				last_at = cs[i];
			}
		}
		/* (a chunk ends at the next planned start: the starts of one planned
		 * position - see add_block() - share theirs) */
		for (size_t i = plan.size(), nxt_at = R1; i-- > 0;) {
			if (i + 1 < plan.size() && plan[i + 1].at != plan[i].at)
				nxt_at = plan[i + 1].at;
			plan[i].c.limit_bit = nxt_at;
		}
		if (!hdr_slot.empty())
			for (planned &q : plan)
				q.c.hdr_cache = cache_of(q.c.hdr_bit);
		const uint32_t np = (uint32_t)plan.size();
		S[4] += np;
		dbg("planned");
		if (cache_queued)
			ST_TRY(hipStreamWaitEvent(s_comp, d->streams.mark2, 0));

		/* ---- count ---- */
		const size_t res_at = align_up((size_t)np * sizeof(lda_stream_chunk) + 64, 64);
		uint8_t *sch = (uint8_t *)d->schunks.reserve(
			res_at + align_up((size_t)np * sizeof(lda_stream_res) + 64, 64));
		if (!sch)
			return false;
		lda_stream_chunk *d_chunks = (lda_stream_chunk *)sch;
		lda_stream_res *d_res = (lda_stream_res *)(sch + res_at);
		std::vector<lda_stream_chunk> hc(np);
		for (uint32_t i = 0; i < np; i++)
			hc[i] = plan[i].c;
		std::vector<lda_stream_res> hr(np);
		if (!pin_phase((size_t)np * (sizeof(lda_stream_chunk) + sizeof(lda_stream_res)) + 256))
			return false;
		ST_TRY(up(d_chunks, hc.data(), (size_t)np * sizeof(lda_stream_chunk)));
		/* (the chunks counted together leave the decode pass their lanes' starts:
		 * a row of 64 per chunk of this launch, see phase_count()) */
		if (nexact) {
			d_hints = (uint16_t *)d->shint.reserve((size_t)np * 128 + 64);
			if (!d_hints)
				return false;
		}
		if (!launch_count(s_comp, np, d_chunks, d_res, d_raw, win_n, d_hlens, d_hinfo,
				  nexact ? d_hints : nullptr))
			return false;
		ST_TRY(back(hr.data(), d_res, (size_t)np * sizeof(lda_stream_res)));
		dbg("count queued");
		ST_TRY(pin_sync());
		dbg("counted");

		/* ---- chain ----
		 * Every counted chunk is a pool entry keyed by its exact start state.
		 * The walk from the window's first chunk follows end state -> start
		 * state; where an end state has no chunk starting there (a block the
		 * finder does not look for, a false candidate, a warm-up that did not
		 * fall in step) a REPAIR chunk is counted from that state up to the
		 * next planned start.  Repairs are made for every open end in the pool
		 * at once, one launch per round, so the number of host round trips is
		 * the longest run of consecutive breaks, not the number of breaks. */
		{
			typedef std::pair<uint64_t, uint64_t> key_t;	/* (start_bit * 2 + boundary, header) */
			/* (hashed: a window of blocks of one codeword length has ten
			 * thousand entries, and an ordered map's inserts were a third of
			 * its count phase) */
			struct key_hash {
				size_t operator()(const key_t &k) const {
					uint64_t h = k.first * 0x9E3779B97F4A7C15ull ^ (k.second + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
					return (size_t)(h ^ (h >> 29));
				}
			};
			std::unordered_map<key_t, uint32_t, key_hash> by_start;
			by_start.reserve((size_t)np / 2 + 4096);
			std::vector<lda_stream_chunk> pc(hc);
			std::vector<lda_stream_res> pr(hr);
			auto start_key = [&](uint32_t i) -> key_t {
				if (pc[i].kind == LDA_CHUNK_HEADER)
					return key_t(pc[i].hdr_bit * 2 + 1, pc[i].hdr_bit);
				return key_t(pr[i].start_bit * 2, pc[i].hdr_bit);
			};
			auto end_key = [&](uint32_t i) -> key_t {
				const bool bnd = pr[i].flags & LDA_RES_BOUNDARY;
				return key_t(pr[i].end_bit * 2 + (bnd ? 1 : 0), bnd ? pr[i].end_bit : pr[i].end_hdr_bit);
			};
			/* The K exact starts of one planned position (phases != 0) are
			 * consecutive entries at consecutive bits: they are found by
			 * position, not through the map (four fifths of a window's
			 * entries where it has such blocks: their inserts were 0.2 ms of
			 * the 16 MiB mix's count phase). */
			struct pgroup { uint64_t P, hdr; uint32_t first, K; };
			std::vector<pgroup> groups;	/* sorted by P */
			auto pool_add = [&](uint32_t i) {
				if (pc[i].phases == ~0u)
					return;
				if (pc[i].phases) {
					const pgroup g = { pc[i].start_bit, pc[i].hdr_bit, i, pc[i].phases };
					groups.insert(std::upper_bound(groups.begin(), groups.end(), g.P,
								       [](uint64_t v, const pgroup &a) { return v < a.P; }), g);
					return;
				}
				if (pc[i].kind == LDA_CHUNK_HEADER || pr[i].status != LDA_STREAM_ERR)
					by_start.emplace(start_key(i), i);
			};
			auto pool_find = [&](const key_t &k) -> int64_t {
				const auto it = by_start.find(k);
				if (it != by_start.end())
					return it->second;
				if ((k.first & 1) || groups.empty())
					return -1;	/* (a start at a block boundary is a header chunk's) */
				const uint64_t e = k.first >> 1;
				auto g = std::upper_bound(groups.begin(), groups.end(), e,
							  [](uint64_t v, const pgroup &a) { return v < a.P; });
				if (g == groups.begin())
					return -1;
				--g;
				if (e - g->P >= g->K || g->hdr != k.second)
					return -1;
				const uint32_t idx = g->first + (uint32_t)(e - g->P);
				return pr[idx].status != LDA_STREAM_ERR ? (int64_t)idx : -1;
			};
			for (uint32_t i = 0; i < np; i++)
				pool_add(i);
			std::vector<uint64_t> ats(np);
			for (uint32_t i = 0; i < np; i++)
				ats[i] = plan[i].at;
			dbg("pool built");
			std::vector<uint32_t> path;
			/* how many repairs in a row led to an entry (0: planned): a
			 * repair behind a repair reaches twice as far as the one before
			 * it - a block whose parse never falls in step (codewords of one
			 * length) is walked in a few long strides, not chunk by chunk */
			std::vector<uint8_t> depth(np, 0);
			uint32_t repairs = 0, first_open = 0;
			/* phase candidates (see the repairs): planned starts that have theirs,
			 * and how many there are (they do not count as repairs for the limit) */
			const uint32_t PHASES = 10;
			std::vector<uint8_t> phased(np + 1, 0);
			uint32_t ncand = 0;
			const uint32_t max_repairs = 64 + 2 * np;
			const uint64_t sgroup = 8 * (uint64_t)(env.stream_chunk ? env.stream_chunk : 16384);
			bool closed = false;	/* the walk ended: final block, or the window's end */
			bool gf_end = false;	/* the walk's end lies in the stream's final (static) block */
			/* chunks planned under the static codes: not a header, not a real block */
			auto under_static = [&](uint32_t i) {
				return pc[i].kind != LDA_CHUNK_HEADER && pc[i].hdr_bit == LDA_HDR_STATIC;
			};
			for (int round = 0; round < 16 && !closed; round++) {
				path.clear();
				uint32_t cur = 0;
				bool gf = carry_gf;
				for (;;) {
					if (pr[cur].status == LDA_STREAM_ERR) {
						if (whole) {
							S[1] = WHY_ERRCHUNK;
							return false;
						}
						closed = true;	/* (it may only have run out of window) */
						break;
					}
					path.push_back(cur);
					if (pr[cur].status == LDA_STREAM_FINAL) {
						closed = final_seen = true;
						break;
					}
					/* is the block the walk stands in the stream's last?  A
					 * chunk that read the header says so itself; one under
					 * the static codes inherits it - and when it stopped at
					 * its block's end, that was the end of the stream */
					const bool bnd_end = pr[cur].flags & LDA_RES_BOUNDARY;
					if (!under_static(cur))
						gf = pr[cur].flags & LDA_RES_GOV_FINAL;
					else if (bnd_end && gf) {
						/* (its status stays OK: that is what the decode
						 * pass will report for it too) */
						closed = final_seen = true;
						break;
					}
					if (bnd_end)
						gf = false;
					gf_end = gf;
					if (pr[cur].end_bit >= R1 || path.size() > pc.size()) {
						if (whole) {
							S[1] = WHY_NOFINAL;	/* ran out of input without a final block */
							return false;
						}
						closed = true;
						break;
					}
					int64_t nxt = pool_find(end_key(cur));
					if (nxt < 0 && bnd_end) {
						/* a run of stored blocks behind this boundary: the
						 * host's (no count pass, no round trip) */
						std::vector<lda_stream_chunk> oc;
						std::vector<lda_stream_res> orr;
						bool fin = false;
						const uint64_t stop = std::min<uint64_t>(win_n, (R1 + 7) / 8 + 65536 + 16);
						(void)walk_stored(pr[cur].end_bit, stop, sgroup, oc, orr, &fin);
						for (size_t k = 0; k < oc.size(); k++) {
							pc.push_back(oc[k]);
							pr.push_back(orr[k]);
							depth.push_back(0);
							pool_add((uint32_t)pc.size() - 1);
						}
						nxt = pool_find(end_key(cur));
					}
					if (nxt < 0)
						break;
					cur = (uint32_t)nxt;
				}
				if (closed)
					break;
				if (debug) {
					const uint32_t e = path.back();
					dbg("walked");
					fprintf(stderr, "round %d: walk of %zu stops after chunk %u (kind %u hdr %llu start %llu) "
						"end %llu bnd %u endhdr %llu status %u nout %llu\n", round, path.size(), e,
						pc[e].kind, (unsigned long long)pc[e].hdr_bit,
						(unsigned long long)pr[e].start_bit, (unsigned long long)pr[e].end_bit,
						pr[e].flags & 1, (unsigned long long)pr[e].end_hdr_bit, pr[e].status,
						(unsigned long long)pr[e].nout);
				}
				/* repairs for every open end (entries added in earlier rounds
				 * were looked at then: start at first_open) */
				std::vector<lda_stream_chunk> rc;
				std::vector<uint8_t> rdepth;
				const uint32_t npool = (uint32_t)pc.size();
				std::unordered_map<key_t, int, key_hash> asked;
				for (uint32_t i = first_open; i < npool; i++) {
					if (pr[i].status != LDA_STREAM_OK || pr[i].end_bit >= R1)
						continue;
					const key_t k = end_key(i);
					if (pool_find(k) >= 0 || asked.count(k))
						continue;
					asked[k] = 1;
					const bool bnd = pr[i].flags & LDA_RES_BOUNDARY;
					lda_stream_chunk c = {};
					c.kind = bnd ? LDA_CHUNK_HEADER : LDA_CHUNK_EXACT;
					c.hdr_bit = bnd ? pr[i].end_bit : pr[i].end_hdr_bit;
					c.start_bit = c.target_bit = pr[i].end_bit;
					/* up to the next planned start - or, behind a repair,
					 * twice as many planned starts further than that one */
					const uint32_t dp = std::min<uint32_t>((uint32_t)depth[i] + 1, 12);
					size_t nx = (size_t)(std::upper_bound(ats.begin(), ats.end(), pr[i].end_bit) -
							     ats.begin());
					const size_t nx0 = nx;	/* the next planned start behind this end */
					nx += ((size_t)1 << (dp - 1)) - 1;
					c.limit_bit = nx >= ats.size() ? R1 : ats[nx];
					rc.push_back(c);
					rdepth.push_back((uint8_t)dp);
					/* PHASE CANDIDATES.  The planned chunk q that should have
					 * gone on from this end did not (its warm-up ended on
					 * another token boundary) - and its own end is open too
					 * or it failed (a parse out of step meets an end-of-block
					 * codeword sooner or later and reads a header that is
					 * none): a code whose parses do not fall in step
					 * (codewords of nearly one length: a dynamic block over
					 * incompressible bytes).  Repairs alone would walk
					 * such a block one stride per round trip.  But the end of
					 * whatever comes to the next planned start P from the
					 * true parse is the first token boundary at or behind P:
					 * a literal's codeword is at most a dozen bits, so one of
					 * the chunks that start EXACTLY at P, P + 1, .. P + K - 1
					 * is the true parse, and the chain finds it by its key.
					 * Every open end of the block asks for the K starts at
					 * its own next planned start, in this same round. */
					const size_t nq = nx0 ? nx0 - 1 : 0;
					if (!bnd && i < np && nx0 >= 1 && nx0 < np && nq != i && nq < np &&
					    plan[nq].c.kind == LDA_CHUNK_WARM &&
					    plan[nq].c.hdr_bit == pr[i].end_hdr_bit &&
					    (pr[nq].status != LDA_STREAM_OK ||
					     pool_find(end_key((uint32_t)nq)) < 0) &&
					    plan[nx0].c.kind == LDA_CHUNK_WARM &&
					    plan[nx0].c.hdr_bit == pr[i].end_hdr_bit && !phased[nx0] &&
					    ncand + PHASES <= 4096) {
						phased[nx0] = 1;
						const uint64_t lim2 = nx0 + 1 < np ? ats[nx0 + 1] : R1;
						/* (one wave for all of them when they are one round
						 * of input: phase_count() of inflate_stream.hip) */
						const bool together = pr[i].end_hdr_bit != LDA_HDR_STATIC &&
								      lim2 > ats[nx0] + PHASES &&
								      lim2 - ats[nx0] <= 24000;
						for (uint32_t j = 0; j < PHASES && ats[nx0] + j < lim2; j++) {
							lda_stream_chunk k2 = {};
							k2.kind = LDA_CHUNK_EXACT;
							k2.hdr_bit = pr[i].end_hdr_bit;
							k2.start_bit = k2.target_bit = ats[nx0] + j;
							k2.limit_bit = lim2;
							k2.phases = !together ? 0 : j ? ~0u : PHASES;
							rc.push_back(k2);
							rdepth.push_back(0);
							ncand++;
						}
					}
				}
				/* the open end of the walk is always among them (round 0 looks at
				 * all entries; later rounds at the new ones, and the walk can only
				 * have stopped at a new one) */
				first_open = npool;
				repairs += (uint32_t)rc.size();
				S[5] += (uint32_t)rc.size();
				if (rc.empty() || repairs > max_repairs + ncand) {
					S[1] = rc.empty() ? WHY_CHAIN : WHY_REPAIRS;
					return false;
				}
				const uint32_t nr = (uint32_t)rc.size();
				if (!hdr_slot.empty())
					for (lda_stream_chunk &c : rc)
						c.hdr_cache = cache_of(c.hdr_bit);
				uint8_t *rp = (uint8_t *)d->srepair.reserve(
					(size_t)nr * (sizeof(lda_stream_chunk) + sizeof(lda_stream_res)) + 128);
				if (!rp)
					return false;
				lda_stream_chunk *d_rc = (lda_stream_chunk *)rp;
				lda_stream_res *d_rr = (lda_stream_res *)(rp + align_up((size_t)nr * sizeof(lda_stream_chunk), 64));
				std::vector<lda_stream_res> rr(nr);
				if (!pin_phase((size_t)nr * (sizeof(lda_stream_chunk) + sizeof(lda_stream_res)) + 256))
					return false;
				ST_TRY(up(d_rc, rc.data(), (size_t)nr * sizeof(lda_stream_chunk)));
				if (!launch_count(s_comp, nr, d_rc, d_rr, d_raw, win_n, d_hlens, d_hinfo, nullptr))
					return false;
				ST_TRY(back(rr.data(), d_rr, (size_t)nr * sizeof(lda_stream_res)));
				dbg("repairs queued");
				ST_TRY(pin_sync());
				dbg("repairs counted");
				for (uint32_t i = 0; i < nr; i++) {
					pc.push_back(rc[i]);
					pr.push_back(rr[i]);
					depth.push_back(rdepth[i]);
				}
				for (uint32_t i = 0; i < nr; i++)
					pool_add(npool + i);
			}
			if (!closed) {
				S[1] = WHY_CHAIN;
				return false;
			}
			for (uint32_t i : path) {
				lda_stream_chunk c = pc[i];
				/* (one of the starts counted together in the window's first
				 * launch: row i holds where its parse entered the pieces) */
				c.hint = pc[i].phases && i < np && d_hints ? i + 1 : 0;
				c.phases = 0;
				if (c.kind == LDA_CHUNK_WARM) {
					c.kind = LDA_CHUNK_EXACT;
					c.start_bit = pr[i].start_bit;
				}
				acc.push_back(c);
				accr.push_back(pr[i]);
			}
			if (!path.empty() && !final_seen) {
				/* the next window goes on from where this one's chain ends */
				const lda_stream_res &e = pr[path.back()];
				const bool bnd = e.flags & LDA_RES_BOUNDARY;
				carry = lda_stream_chunk();
				carry.kind = bnd ? LDA_CHUNK_HEADER : LDA_CHUNK_EXACT;
				carry.hdr_bit = bnd ? e.end_bit : e.end_hdr_bit;
				carry.start_bit = carry.target_bit = e.end_bit;
				carry_gf = !bnd && gf_end;
			}
		}
		lap(10);
		if (whole && !final_seen) {
			S[1] = WHY_NOFINAL;
			return false;
		}
		/* an output buffer that is already too small is known now: no further
		 * window is copied and searched for a call that cannot succeed here
		 * (programs/gzip.c retries with a larger buffer: every attempt would
		 * pay for all windows) */
		{
			uint64_t sofar = 0;
			for (const lda_stream_res &r : accr)
				sofar += r.nout;
			if (sofar > out_avail) {
				S[1] = WHY_SPACE;
				return false;
			}
		}
	}
	const uint32_t na = (uint32_t)acc.size();
	S[6] = na;
	std::vector<uint64_t> offs(na + 1);
	uint64_t total = 0;
	for (uint32_t i = 0; i < na; i++) {
		offs[i] = total;
		acc[i].out_off = total;
		total += accr[i].nout;
	}
	offs[na] = total;
	const uint64_t end_bit = accr[na - 1].end_bit;
	if (end_bit > raw_bits) {
		/* a chain that closes beyond the raw stream (inside the footer): the
		 * sequential kernel decides what the reference would say, and the
		 * footer is never read at f = in + hdr + consumed past the buffer */
		S[1] = WHY_NOFINAL;
		return false;
	}
	if (total > out_avail) {
		S[1] = WHY_SPACE;
		return false;
	}
	if (exact_fill && total != out_avail) {
		S[1] = WHY_FILL;
		return false;
	}

	/* ---- decode -> window -> resolve -> checksum, one round trip ----
	 * Everything is queued on the compute stream; what comes back (the decode
	 * pass's per-chunk results, the error flag, the pieces' checksums) is
	 * looked at after ONE synchronisation, and the output is on its way to the
	 * caller meanwhile: the copy stream waits for the resolve pass, not for the
	 * host.  (Output is undefined on failure, libdeflate.h:216-217; when the
	 * sequential kernel has to decide it writes the buffer again.) */
	const size_t consumed = (size_t)((end_bit + 7) / 8);
	int32_t result = LIBDEFLATE_SUCCESS;
	uint32_t sum = format == LIBDEFLATE_AMD_GZIP ? 0u : 1u;
	if (total) {
		S[1] = WHY_DEVICE;
		if (!d_cnt) {
			/* (no window went through the block finder: every block was
			 * the host's) */
			uint8_t *sq = (uint8_t *)d->squeue.reserve(128);
			if (!sq)
				return false;
			d_cnt = (uint32_t *)sq;
		}
		/* [2]: the error flag of the window / resolve kernels */
		ST_TRY(hipMemsetAsync(d_cnt, 0, 16, s_comp));
		const size_t BATCH = 4096;	/* decode waves per launch (their token scratch: 48 KiB each) */
		uint16_t *d_sym = (uint16_t *)d->ssym.reserve((size_t)total * 2 + 64);
		uint8_t *d_out = (uint8_t *)d->sout.reserve((size_t)total + 64);
		uint32_t *d_tok = (uint32_t *)d->tokens.reserve(
			std::min<size_t>(na, BATCH) * lda_stream_tokcap() * 4 + 64);
		if (!d_sym || !d_out || !d_tok)
			return false;
		/* (the accepted chain may hold more chunks than were planned) */
		const size_t res2_at = align_up((size_t)na * sizeof(lda_stream_chunk) + 64, 64);
		const size_t off2_at = res2_at + align_up((size_t)na * sizeof(lda_stream_res) + 64, 64);
		uint8_t *sch = (uint8_t *)d->schunks.reserve(off2_at + ((size_t)na + 2) * 8 + 64);
		if (!sch)
			return false;
		lda_stream_chunk *d_chunks = (lda_stream_chunk *)sch;
		lda_stream_res *d_res = (lda_stream_res *)(sch + res2_at);
		uint64_t *d_off = (uint64_t *)(sch + off2_at);
		/* pieces of the output for the checksum kernels */
		const uint64_t piece = std::max<uint64_t>(65536, align_up(total / 4096, 4096));
		const size_t npc = ftr ? (size_t)((total + piece - 1) / piece) : 0;
		std::vector<uint64_t> po(2 * npc);
		for (size_t i = 0; i < npc; i++) {
			po[i] = i * piece;
			po[npc + i] = std::min<uint64_t>(piece, total - i * piece);
		}
		uint8_t *scr = npc ? (uint8_t *)d->scratch.reserve(npc * 20 + 64) : nullptr;
		if (npc && !scr)
			return false;
		uint64_t *d_po = (uint64_t *)scr;
		uint32_t *d_sums = (uint32_t *)(scr + npc * 16);
		if (!pin_phase((size_t)na * (sizeof(lda_stream_chunk) + sizeof(lda_stream_res) + 8) +
			       npc * 20 + 1024))
			return false;
		ST_TRY(up(d_chunks, acc.data(), (size_t)na * sizeof(lda_stream_chunk)));
		ST_TRY(up(d_off, offs.data(), ((size_t)na + 1) * 8));
		if (npc)
			ST_TRY(up(d_po, po.data(), npc * 16));
		for (size_t lo = 0; lo < na; lo += BATCH) {
			const uint32_t nk = (uint32_t)std::min<size_t>(BATCH, na - lo);
			hipLaunchKernelGGL(lda_stream_decode_kernel, dim3(nk), dim3(64),
					   lda_stream_chunk_lds(), s_comp, nk, d_chunks + lo,
					   d_res + lo, d_raw, dev_n, d_sym, d_tok, d_hlens, d_hinfo, d_hints);
		}
		{
			/* the window chain: groups of chunks side by side with a symbolic
			 * window, the groups' windows composed by a prefix scan (log2
			 * launches), the groups again from their real windows.  Up to 512
			 * groups - two workgroups of 64 KiB LDS per CU - of at least four
			 * chunks (round 5: sqrt(chunks) / 2 groups, because a serial link
			 * step per group had to be paid) */
			uint32_t per_group = 4;
			while ((uint64_t)per_group * 512 < na)
				per_group++;
			const uint32_t groups = (na + per_group - 1) / per_group;
			uint8_t *gw = (uint8_t *)d->swin.reserve((size_t)groups * 65536 * 2 + 64);
			if (!gw)
				return false;
			uint16_t *d_gwin = (uint16_t *)gw;
			uint16_t *d_gwin2 = d_gwin + (size_t)groups * 32768;
			const uint16_t *d_fwin = d_gwin;
			if (!ctx->stream_attr_set.load(std::memory_order_acquire)) {
				ST_TRY(hipFuncSetAttribute((const void *)lda_stream_window_kernel,
							   hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
				ST_TRY(hipFuncSetAttribute((const void *)lda_stream_window_scan_kernel,
							   hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
				ctx->stream_attr_set.store(true, std::memory_order_release);
			}
			if (groups > 1) {
				hipLaunchKernelGGL(lda_stream_window_kernel, dim3(groups), dim3(1024), 65536,
						   s_comp, na, per_group, 0u, d_off, d_sym, d_out, d_gwin,
						   d_fwin, d_cnt + 2);
				/* (the window behind the last group is nobody's) */
				uint16_t *src = d_gwin, *dst = d_gwin2;
				for (uint32_t h = 1; h < groups - 1; h *= 2) {
					hipLaunchKernelGGL(lda_stream_window_scan_kernel, dim3(groups - 1), dim3(1024),
							   65536, s_comp, groups - 1, h, src, dst);
					std::swap(src, dst);
				}
				d_fwin = src;
			}
			hipLaunchKernelGGL(lda_stream_window_kernel, dim3(groups), dim3(1024), 65536, s_comp,
					   na, per_group, 2u, d_off, d_sym, d_out, d_gwin, d_fwin, d_cnt + 2);
		}
		{
			uint64_t longest = 0;
			for (uint32_t i = 0; i < na; i++)
				longest = std::max(longest, accr[i].nout);
			if (longest > 32768) {
				const unsigned gx = (unsigned)std::min<uint64_t>(
					(longest - 32768 + 2047) / 2048, 64);
				/* (grid.y <= 65535: a window of small blocks can have more
				 * chunks than that) */
				for (uint32_t c0 = 0; c0 < na; c0 += 32768)
					hipLaunchKernelGGL(lda_stream_resolve_kernel,
							   dim3(gx, std::min<uint32_t>(32768, na - c0)), dim3(256),
							   0, s_comp, na, c0, d_off, d_sym, d_out, d_cnt + 2);
			}
		}
		ST_TRY(hipGetLastError());
		ST_TRY(hipEventRecord(d->streams.mark, s_comp));	/* the bytes are final */
		if (npc) {
			const int rc = format == LIBDEFLATE_AMD_GZIP ?
				libdeflate_amd_crc32_batch(npc, d_out, d_po, d_po + npc, NULL, d_sums, s_comp) :
				libdeflate_amd_adler32_batch(npc, d_out, d_po, d_po + npc, NULL, d_sums, s_comp);
			if (rc != LIBDEFLATE_AMD_OK)
				return false;
		}
		std::vector<lda_stream_res> dr(na);
		std::vector<uint32_t> sums(npc);
		uint32_t err = 0;
		ST_TRY(back(dr.data(), d_res, (size_t)na * sizeof(lda_stream_res)));
		ST_TRY(back(&err, d_cnt + 2, 4));
		if (npc)
			ST_TRY(back(sums.data(), d_sums, npc * 4));
		lap(11);
		if (debug) {
			ST_TRY(hipEventSynchronize(d->streams.mark));
			dbg("decode .. resolve kernels");
		}
		/* the output, beside the checksum kernels and the read-backs */
		ST_TRY(hipStreamWaitEvent(s_copy, d->streams.mark, 0));
		if (span_out(&d->pinned, d_out, 0, out, (size_t)total, s_copy) != LIBDEFLATE_AMD_OK)
			return false;
		lap(13);
		ST_TRY(pin_sync());
		bool same = err == 0;
		for (uint32_t i = 0; i < na && same; i++)
			same = dr[i].end_bit == accr[i].end_bit && dr[i].nout == accr[i].nout &&
			       dr[i].status == accr[i].status && !(dr[i].flags & LDA_RES_BAD_DIST);
		if (!same) {
			S[1] = WHY_DECODE;
			return false;
		}
		const uint32_t shp = format == LIBDEFLATE_AMD_GZIP ? crc32_shift(piece) : 0;
		for (size_t i = 0; i < npc; i++)
			sum = i == 0 ? sums[0] :
			      format != LIBDEFLATE_AMD_GZIP ? adler32_concat(sum, sums[i], po[npc + i]) :
			      po[npc + i] == piece ? crc32_concat_shift(sum, sums[i], shp) :
						     crc32_concat(sum, sums[i], po[npc + i]);
	}
	if (ftr) {
		const uint8_t *f = in + hdr + consumed;
		if (format == LIBDEFLATE_AMD_GZIP) {
			const uint32_t want = f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) |
					      ((uint32_t)f[3] << 24);
			const uint32_t isize = f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) |
					       ((uint32_t)f[7] << 24);
			if (want != sum || isize != (uint32_t)total)
				result = LIBDEFLATE_BAD_DATA;
		} else {
			const uint32_t want = ((uint32_t)f[0] << 24) | ((uint32_t)f[1] << 16) |
					      ((uint32_t)f[2] << 8) | f[3];
			if (want != sum)
				result = LIBDEFLATE_BAD_DATA;
		}
	}
	lap(12);
	*res = result;
	if (result == LIBDEFLATE_SUCCESS) {
		*ain = hdr + consumed + ftr;
		*aout = (size_t)total;
	}
	S[0] = 1;
	S[1] = WHY_OK;
	S[7] = total;
	return true;
}

} /* namespace lda */

extern "C" LIBDEFLATEAPI void libdeflate_amd_stream_stats(uint64_t *out)
{
	memcpy(out, lda::g_stats, sizeof(lda::g_stats));
}
