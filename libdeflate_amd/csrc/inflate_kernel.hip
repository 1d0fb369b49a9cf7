/*
 * inflate_kernel.hip - batched raw DEFLATE / zlib / gzip decoding for gfx950.
 *
 * Replaces, for a batch of independent streams resident in HBM:
 *   libdeflate_deflate_decompress_ex  lib/deflate_decompress.c:1133-1142,
 *                                     lib/decompress_template.h:44-772
 *   build_decode_table                lib/deflate_decompress.c:721-1004
 *   gzip / zlib header parsing        lib/gzip_decompress.c:45-107,
 *                                     lib/zlib_decompress.c:45-72
 * (footer verification runs in lda_inflate_finalize_kernel after the batched
 * checksum kernel.)
 *
 * Two mappings share the sequential decoder in inflate_block():
 *
 * WAVE PER STREAM (lda_inflate_wave_kernel, the default).  Inside a Huffman
 * block the 64 lanes parse 64 consecutive 384-bit pieces of the input at
 * once: a parse started at an arbitrary bit falls in step with the true one
 * within a few dozen bits, so a few passes in which every lane restarts
 * where its left neighbour ended give the exact token boundaries and, as a
 * by-product, the tokens; these are then executed byte-parallel ("sub-block
 * parallel token decoding" below).  Headers, stored blocks, the last bytes of a stream and every error
 * path run on lane 0 through the sequential code, so result codes are the
 * same in both mappings.
 *
 * LANE PER STREAM (lda_inflate_batch_kernel, LDA_INFLATE_PAR=0).  Huffman
 * decoding is a serial dependence chain per stream, so here a stream keeps
 * exactly one lane busy.  A wave carries `lpw` streams (lanes 0..lpw-1);
 * every wave instruction advances all of them:
 *
 *   - per-stream state lives in registers (64-bit bit buffer, one 8-byte
 *     word of input prefetched ahead, output cursor) and 2112 bytes of LDS
 *     (9-bit litlen table, 7-bit offset table, 16-bit entries; canonical
 *     first-code/count arrays for the rare longer codewords - no subtables);
 *   - each round every lane decodes one token and writes it: literals are
 *     byte stores, matches are copied by the lane itself with 8-byte
 *     loads/stores from its own earlier output (same-lane program order makes
 *     the read-after-write safe without fences);
 *   - block headers are parsed by the lanes that need one, in lock step;
 *     the litlen/offset tables of ONE stream are then built by ALL 64 lanes
 *     (ballot-ranked counting sort of the code lengths, then each lane
 *     canonically decodes its own table indices);
 *   - the host picks lpw (streams per wave) so that the whole batch is in
 *     flight at once: 2 while the batch fits, doubling from there (2 at 4096
 *     streams, 8 at 65 536 on 256 CUs; host_decompress.hip).
 *
 * Result codes follow the reference bit for bit, including the implicit
 * zero-padding rule: the reference fails when a refill would need a 9th
 * zero byte (lib/deflate_decompress.c:236-254); with whole-byte refills to
 * 56..63 bits that is exactly "bits consumed > 8*in_nbytes + 8 at a refill
 * point", which is what is tested here at the same program points.
 */
#include "device_common.h"
#include "kernels.h"

#ifndef PAR_PRIO
#define PAR_PRIO 1	/* wave-per-stream: issue priority by the share of the input still ahead */
#endif
/*
 * Primary table bits.  The tables' LDS holds LIT_TB / OFF_TB bits; a block
 * uses that many or the _MIN ones (inflate_block() decides per block: a table
 * of 10 bits sends a wave down the long-codeword path in one step of five
 * instead of one of two on text, but costs twice the entries to fill - a
 * small block keeps 9 / 7).  The parse (par_decode()) takes the masks as
 * wave-uniform values; the long-codeword walk starts at the _MIN length and a
 * block with the larger table has that length's limit out of reach.
 */
#ifndef LIT_TB
#define LIT_TB 10
#endif
#ifndef LIT_TB_MIN
#define LIT_TB_MIN 9
#endif
#ifndef OFF_TB
#define OFF_TB 8
#endif
#ifndef OFF_TB_MIN
#define OFF_TB_MIN 7
#endif
#define TB_BIG_BLOCK 8192u	/* bytes of input: see inflate_block() */

/* 16-bit table entry: [3:0] codeword length (0 = longer than the table),
 * [15:14] kind, [13:4] payload (literal byte / length index / offset index /
 * precode symbol) */
#define K_LIT (0u << 14)
#define K_LEN (1u << 14)
#define K_EOB (2u << 14)
#define ENTRY(kind, payload, len) ((u16)((kind) | ((payload) << 4) | (len)))

struct canon16 {
	u16 count[16];
	u16 first[16];
	u16 index[16];
};

struct stream_lds {
	union {
		u16 lit_tab[1 << LIT_TB];
		u8 lens[288 + 32 + 138 + 6];	/* during header parsing */
	};
	union {
		u16 off_tab[1 << OFF_TB];
		u16 pre_tab[128];		/* during header parsing */
	};
	struct canon16 lit, off;
	u16 lit_sorted[288];
	u16 off_sorted[32];
	u8 in_ring[128 + 8];	/* this stream's next input bytes (+8 mirror) */
} __attribute__((aligned(16)));	/* what follows it in LDS (mirror, copy scratch) takes 16-byte accesses */

__constant__ u8 c_pre_perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4,
				   12, 3, 13, 2, 14, 1, 15 };

/* per-workgroup tables for the fast loop: symbol -> base | extra bits << 16 */
struct shared_lds {
	u32 len_tab[32];
	u32 dist_tab[32];
};

/* LDS-resident: pointers into the workgroup's LDS carry the address space,
 * and the launch's dynamic LDS is the only LDS of these kernels, so it starts
 * at LDS address 0 and member offsets become instruction immediates */
#ifdef __HIP_DEVICE_COMPILE__
#define AS3 __attribute__((address_space(3)))
#else
#define AS3	/* the host pass only parses the device code */
#endif
typedef AS3 struct stream_lds slds_t;
typedef AS3 struct shared_lds shlds_t;
typedef AS3 struct canon16 lcanon_t;
typedef AS3 u8 lu8;
typedef AS3 u16 lu16;
typedef AS3 u32 lu32;
typedef AS3 u64 lu64;

enum { ST_HDR = 0, ST_PRETAB, ST_LENS, ST_TABLES, ST_TOK, ST_STORED, ST_DONE };

/* length / offset symbol -> base and extra-bit count, computed instead of
 * looked up (values of lib/deflate_decompress.c:555-588, :615-628; symbols
 * 286/287 and 30/31 alias their neighbours exactly as there) */
static __device__ __forceinline__ void len_sym(u32 s, u32 *base, u32 *xb)
{
	/* straight-line selects: as if / else the parse loop ran both sides
	 * behind EXEC masks on every token */
	const u32 e = (s - 4) >> 2;	/* (garbage below 8: not selected) */
	const u32 mid = 3 + ((4 | (s & 3)) << (e & 7));
	const bool lo = s < 8, hi = s >= 28;

	*xb = lo || hi ? 0 : e;
	*base = hi ? 258 : lo ? 3 + s : mid;
}

static __device__ __forceinline__ void off_sym(u32 d, u32 *base, u32 *xb)
{
	if (d > 29)
		d = 29;
	const u32 e = (d - 2) >> 1;	/* (garbage below 4: not selected) */
	const u32 big = 1 + ((2 | (d & 1)) << (e & 15));
	const bool lo = d < 4;

	*xb = lo ? 0 : e;
	*base = lo ? 1 + d : big;
}

/*
 * 8 input bytes at stream position pos; bytes past the end read as zero.
 * Two ALIGNED 8-byte loads and a funnel shift: an aligned word that contains
 * at least one valid byte can always be read safely, and there is no
 * per-byte tail loop in the instruction stream.
 */
static __device__ __forceinline__ u64 load_in(const u8 *inp, u64 in_n, u64 pos)
{
	if (pos >= in_n)
		return 0;
	uintptr_t a = (uintptr_t)(inp + pos);
	const u64 *w = (const u64 *)(a & ~(uintptr_t)7);
	u32 sh = (u32)(a & 7) * 8;
	u64 avail = in_n - pos;		/* valid bytes from pos on */
	u64 lo = w[0];
	u64 v = lo >> sh;
	/* the next word is needed (and valid) only if bytes beyond this one
	 * are both wanted and inside the buffer */
	if (sh && avail > 8 - (sh >> 3))
		v |= w[1] << (64 - sh);
	if (avail < 8)
		v &= (1ull << (8 * avail)) - 1;
	return v;
}

/* canonical bit-serial decode for codewords longer than the table */
static __device__ u32
decode_long(const lcanon_t *cn, const lu16 *sorted, u64 bits, u32 *len_ret)
{
	u32 code = 0;
	for (u32 l = 1; l <= 15; l++) {
		code = (code << 1) | (u32)((bits >> (l - 1)) & 1);
		u32 rel = code - cn->first[l];
		if (rel < cn->count[l]) {
			*len_ret = l;
			return sorted[cn->index[l] + rel];
		}
	}
	*len_ret = 15;	/* unreachable for a complete code */
	return 256;
}

/*
 * Per-lane (serial) build of the 7-bit precode table from 19 lengths.
 * Validity rules of lib/deflate_decompress.c:804-853.  Returns false if the
 * code is invalid.
 */
static __device__ bool build_precode(lu16 *tab, const u8 *plens)
{
	u32 cnt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	for (u32 s = 0; s < 19; s++)
		cnt[plens[s]]++;
	cnt[0] = 0;
	u32 maxlen = 7;
	while (maxlen > 1 && cnt[maxlen] == 0)
		maxlen--;
	u32 used = 0, next[8], code = 0;
	for (u32 l = 1; l <= 7; l++) {
		next[l] = code;
		code = (code + cnt[l]) << 1;
		if (l <= maxlen)
			used = (used << 1) + cnt[l];
	}
	if (used > (1u << maxlen))
		return false;
	if (used < (1u << maxlen)) {
		u32 sym = 0;
		if (used != 0) {
			if (used != (1u << (maxlen - 1)) || cnt[1] != 1)
				return false;
			for (u32 s = 0; s < 19; s++)
				if (plens[s] == 1)
					sym = s;
		}
		for (u32 i = 0; i < 128; i++)
			tab[i] = ENTRY(0, sym, 1);
		return true;
	}
	for (u32 s = 0; s < 19; s++) {
		u32 l = plens[s];
		if (!l)
			continue;
		u32 rev = __brev(next[l]++) >> (32 - l);
		for (u32 i = rev; i < 128; i += 1u << l)
			tab[i] = ENTRY(0, s, l);
	}
	return true;
}

/*
 * The same by all 64 lanes, from the 19 lengths in LDS (the wave-per-stream
 * kernel: the serial build above indexes small private arrays with run-time
 * values, which compile into select chains - ten thousand instructions on
 * the one lane that holds the stream, 5 % of a 64 KiB stream's).  Lane s is
 * symbol s: the counts per length are ballots, a symbol's codeword is the
 * first codeword of its length plus its rank among the symbols of that
 * length, and it fills its own 2^(7 - len) table entries.  Same validity
 * rules, same tables.  Returns false (uniformly) if the code is invalid.
 */
static __device__ bool build_precode_coop(lu16 *tab, const lu8 *plens, u32 lane)
{
	const u32 l = lane < 19 ? plens[lane] : 0;
	u64 m[8];
	u32 c[8];
#pragma unroll
	for (u32 L = 1; L < 8; L++) {
		m[L] = __ballot(l == L);
		c[L] = (u32)__builtin_popcountll(m[L]);
	}
	u32 maxlen = 7;
	while (maxlen > 1 && c[maxlen] == 0)
		maxlen--;
	u32 used = 0, next[8], code = 0;
#pragma unroll
	for (u32 L = 1; L < 8; L++) {
		next[L] = code;
		code = (code + c[L]) << 1;
		if (L <= maxlen)
			used = (used << 1) + c[L];
	}
	if (used > (1u << maxlen))
		return false;
	if (used < (1u << maxlen)) {
		u32 sym = 0;
		if (used != 0) {
			if (used != (1u << (maxlen - 1)) || c[1] != 1)
				return false;
			sym = (u32)__builtin_ctzll(m[1]);	/* the one symbol of length 1 */
		}
		tab[lane] = ENTRY(0, sym, 1);
		tab[64 + lane] = ENTRY(0, sym, 1);
		return true;
	}
	u32 cw = 0;
	const u64 lt = (1ull << lane) - 1;
#pragma unroll
	for (u32 L = 1; L < 8; L++)
		cw = l == L ? next[L] + (u32)__builtin_popcountll(m[L] & lt) : cw;
	if (l) {
		const u32 rev = __brev(cw) >> (32 - l);
		for (u32 i = rev; i < 128; i += 1u << l)
			tab[i] = ENTRY(0, lane, l);
	}
	return true;
}

/*
 * Cooperative (all 64 lanes) build of one stream's litlen or offset table
 * from lens[0..n).  Counting and the (len, sym) sort are done with ballots,
 * then every lane canonically decodes its own table indices.  Returns false
 * (uniformly) when the code is invalid.
 */
static __device__ bool
build_table_coop(const lu8 *lens, u32 n, u32 tb, bool is_litlen,
		 lcanon_t *cn, lu16 *sorted, u32 lane, u32 *single_ret)
{
	u32 c[16];
#pragma unroll
	for (u32 l = 0; l < 16; l++)
		c[l] = 0;
	for (u32 s0 = 0; s0 < n; s0 += 64) {
		u32 s = s0 + lane;
		u32 l = s < n ? lens[s] : 0;
#pragma unroll
		for (u32 L = 1; L < 16; L++)
			c[L] += __builtin_popcountll(__ballot(l == L));
	}
	u32 maxlen = 15;
	while (maxlen > 1 && c[maxlen] == 0)
		maxlen--;
	u32 used = 0, code = 0, idx = 0;
	u32 first[16], index[16];
#pragma unroll
	for (u32 l = 1; l < 16; l++) {
		first[l] = code;
		index[l] = idx;
		code = (code + c[l]) << 1;
		idx += c[l];
		if (l <= maxlen)
			used = (used << 1) + c[l];
	}
	if (lane == 0) {
#pragma unroll
		for (u32 l = 1; l < 16; l++) {
			cn->count[l] = (u16)c[l];
			cn->first[l] = (u16)first[l];
			cn->index[l] = (u16)index[l];
		}
	}
	/* sorted[] by (len, sym) */
	u32 run[16];
#pragma unroll
	for (u32 l = 1; l < 16; l++)
		run[l] = index[l];
	for (u32 s0 = 0; s0 < n; s0 += 64) {
		u32 s = s0 + lane;
		u32 l = s < n ? lens[s] : 0;
#pragma unroll
		for (u32 L = 1; L < 16; L++) {
			u64 m = __ballot(l == L);
			if (l == L)
				sorted[run[L] + __builtin_popcountll(m & ((1ull << lane) - 1))] = (u16)s;
			run[L] += __builtin_popcountll(m);
		}
	}
	wave_sync();
	*single_ret = 0xFFFFFFFFu;
	if (used > (1u << maxlen))
		return false;
	if (used < (1u << maxlen)) {
		if (used == 0) {
			*single_ret = 0;
		} else {
			if (used != (1u << (maxlen - 1)) || c[1] != 1)
				return false;
			*single_ret = sorted[0];
		}
	}
	(void)tb;
	(void)is_litlen;
	return true;
}

static __device__ __forceinline__ u16 lit_entry(u32 sym, u32 len)
{
	if (sym < 256)
		return ENTRY(K_LIT, sym, len);
	if (sym == 256)
		return ENTRY(K_EOB, 0, len);
	return ENTRY(K_LEN, sym - 257, len);
}

/* fill a table: each lane canonically decodes its own indices */
static __device__ void
fill_table(lu16 *tab, u32 tb, bool is_litlen, const lcanon_t *cn,
	   const lu16 *sorted, u32 single, u32 lane)
{
	for (u32 e = lane; e < (1u << tb); e += 64) {
		u16 entry = 0;
		if (single != 0xFFFFFFFFu) {
			entry = is_litlen ? lit_entry(single, 1) : ENTRY(0, single, 1);
		} else {
			u32 code = 0;
			for (u32 l = 1; l <= tb; l++) {
				code = (code << 1) | ((e >> (l - 1)) & 1);
				u32 rel = code - cn->first[l];
				if (rel < cn->count[l]) {
					u32 sym = sorted[cn->index[l] + rel];
					entry = is_litlen ? lit_entry(sym, l) : ENTRY(0, sym, l);
					break;
				}
			}
		}
		tab[e] = entry;
	}
}

/*
 * Copy a match inside one lane's output.  Global round trips are the cost
 * here (a load the next store depends on), so the bytes are moved in 8-byte
 * words with all loads of a pass issued before its stores:
 *   - dist >= 8: passes of up to 4 words, never wider than the distance, so
 *     a pass never reads what it writes; the last word may overrun the match
 *     by up to 7 bytes (overwritten by the following tokens) when the output
 *     buffer has room, else the tail goes bytewise;
 *   - dist < 8 (runs): the period is expanded in registers, no reloads.
 * Same-lane program order makes reading back earlier stores safe.
 */
static __device__ __forceinline__ u64 ld8(const u8 *p)
{
	u64 v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

static __device__ __forceinline__ void st8(u8 *p, u64 v)
{
	__builtin_memcpy(p, &v, 8);
}

static __device__ void
copy_match(u8 *outp, u64 out_pos, u64 out_avail, u32 dist, u32 length)
{
	u8 *dst = outp + out_pos;
	const u8 *src = dst - dist;
	u32 nwords = (length + 7) >> 3;
	bool room = out_pos + 8ull * nwords <= out_avail;

	if (dist >= 8 && room) {
		u32 maxw = dist >> 3;	/* words per pass that cannot overlap */
		if (maxw > 4)
			maxw = 4;
		u32 w = 0;
		while (w < nwords) {
			u32 cnt = nwords - w < maxw ? nwords - w : maxw;
			u64 v0 = ld8(src + 8 * w), v1 = 0, v2 = 0, v3 = 0;
			if (cnt > 1)
				v1 = ld8(src + 8 * w + 8);
			if (cnt > 2)
				v2 = ld8(src + 8 * w + 16);
			if (cnt > 3)
				v3 = ld8(src + 8 * w + 24);
			st8(dst + 8 * w, v0);
			if (cnt > 1)
				st8(dst + 8 * w + 8, v1);
			if (cnt > 2)
				st8(dst + 8 * w + 16, v2);
			if (cnt > 3)
				st8(dst + 8 * w + 24, v3);
			w += cnt;
		}
		return;
	}
	if (dist < 8 && room && dist <= out_pos) {
		/* period 'dist' (1..7): expand in registers */
		u64 pat = 0;
		for (u32 k = 0; k < dist; k++)
			pat |= (u64)src[k] << (8 * k);
		u32 ph = 0;	/* phase = (8*w) % dist */
		for (u32 w = 0; w < nwords; w++) {
			u64 v = 0;
			u32 q = ph;
			for (u32 j = 0; j < 8; j++) {
				v |= ((pat >> (8 * q)) & 0xFF) << (8 * j);
				q = q + 1 == dist ? 0 : q + 1;
			}
			st8(dst + 8 * w, v);
			ph = q;
		}
		return;
	}
	/* near the end of the output buffer: exact, bytewise */
	for (u32 k = 0; k < length; k++)
		dst[k] = src[k];
}

/*
 * Input staging: the lane keeps the 64-byte block it is reading and the next
 * one in a 128-byte LDS ring, so bit-buffer refills are LDS reads (lgkmcnt)
 * and never wait behind the output stores in the vector-memory queue; a new
 * block is fetched from HBM once per 64 bytes consumed.
 */
static __device__ __forceinline__ void
ring_fill(lu8 *ring, const u8 *inp, u64 in_n, u64 at)
{
	u32 slot = (u32)at & 64;

	if (at + 72 <= in_n) {
		/* common case: nine aligned words, all loads in flight together,
		 * then funnel-shifted into place */
		uintptr_t a = (uintptr_t)(inp + at);
		const u64 *w = (const u64 *)(a & ~(uintptr_t)7);
		u32 sh = (u32)(a & 7) * 8;
		u64 v[9];
#pragma unroll
		for (u32 k = 0; k < 9; k++)
			v[k] = w[k];
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			u64 x = sh ? (v[k] >> sh) | (v[k + 1] << (64 - sh)) : v[k];
			__builtin_memcpy(ring + slot + 8 * k, &x, 8);
			if (slot == 0 && k == 0)
				__builtin_memcpy(ring + 128, &x, 8);
		}
		return;
	}
#pragma unroll 1
	for (u32 k = 0; k < 8; k++) {
		u64 x = load_in(inp, in_n, at + 8 * k);
		__builtin_memcpy(ring + slot + 8 * k, &x, 8);
		if (slot == 0 && k == 0)
			__builtin_memcpy(ring + 128, &x, 8);
	}
}

#define ENSURE_INPUT()                                                        \
	do {                                                                  \
		while (filled < rpos + 64) {                                  \
			ring_fill(S->in_ring, inp, in_n, filled);             \
			filled += 64;                                         \
		}                                                             \
	} while (0)
#define REFILL()                                                              \
	do {                                                                  \
		u64 w_;                                                       \
		__builtin_memcpy(&w_, S->in_ring + ((u32)rpos & 127), 8);     \
		bitbuf |= w_ << bitcnt;                                       \
		rpos += (63 - bitcnt) >> 3;                                   \
		bitcnt |= 56;                                                 \
	} while (0)
#define CONSUME(k)                                                            \
	do {                                                                  \
		bitbuf >>= (k);                                               \
		bitcnt -= (k);                                                \
	} while (0)
#define CONSUMED() (8 * rpos - bitcnt)
#define FLUSH_PENDING()                                                       \
	do {                                                                  \
		if (pend_n) {                                                 \
			st8(pend_dst, pv0);                                   \
			if (pend_n > 1)                                       \
				st8(pend_dst + 8, pv1);                       \
			if (pend_n > 2)                                       \
				st8(pend_dst + 16, pv2);                      \
			if (pend_n > 3)                                       \
				st8(pend_dst + 24, pv3);                      \
			/* the copied bytes are now known: refresh the history */ \
			if (pend_len <= 8) {                                  \
				hist = pend_len == 8 ? pv0 :                  \
				       (hist >> (8 * pend_len)) |             \
				       (pv0 << (8 * (8 - pend_len)));         \
				hist_n = hist_n + pend_len > 8 ? 8 : hist_n + pend_len; \
			} else {                                              \
				u32 o_ = pend_len - 8, j_ = o_ >> 3, sh_ = (o_ & 7) * 8; \
				u64 a_ = j_ == 0 ? pv0 : j_ == 1 ? pv1 : j_ == 2 ? pv2 : pv3; \
				u64 b_ = j_ == 0 ? pv1 : j_ == 1 ? pv2 : pv3; \
				hist = sh_ ? (a_ >> sh_) | (b_ << (64 - sh_)) : a_; \
				hist_n = 8;                                   \
			}                                                     \
			pend_n = 0;                                           \
		}                                                             \
	} while (0)

/* ---------------- sub-block parallel token decoding ----------------
 *
 * A Huffman-coded DEFLATE block can only be parsed from its first bit, but a
 * parse started at an arbitrary bit falls in step with the true parse after a
 * few dozen bits (measured on level-6 streams of the benchmark data: median
 * 52 bits, 99 % within 450).  One round gives each of the 64 lanes PAR_CB
 * bits of the block:
 *
 *   sync   every lane parses from its (guessed) start to the end of its
 *          piece and reports where its last token ended, how many tokens and
 *          output bytes it saw; lane i + 1 then restarts from lane i's end.
 *          Lane 0 starts at a known token boundary, so after k passes lanes
 *          0..k-1 are exact; measured: three or four passes settle all 64
 *          (the last ones for two to five lanes).  Every parse writes its
 *          tokens (literal byte, or length and distance) as it goes, into
 *          lane-interleaved rows of the wave's scratch in HBM - row k holds
 *          the k-th token of every lane, one coalesced 256-byte store per
 *          step - so the tokens of a lane's last parse, the one from its
 *          exact start, are simply there when the passes end: there is no
 *          separate emit parse;
 *   copy   the tokens are executed in groups of up to 256 tokens / 1 KiB of
 *          output (tok_fetch maps the group's tokens, in stream order, back
 *          to rows), byte-parallel and 64 bytes at a time in output order
 *          (see the copy phase in par_round): the bytes meet in a 4 KiB LDS
 *          mirror of the recent output, which also serves nearly every
 *          match source, and go to HBM in whole words once per group.  A
 *          wave's vector memory operations reach its L1 in issue order, so
 *          loads see earlier stores of other lanes without waiting for write
 *          acknowledgements; only the compiler has to be kept from
 *          reordering them (wave_sync).
 *
 * The kernel is bound by VALU issue (a wave64 instruction occupies its SIMD
 * for four cycles; with four waves per SIMD the measured VALU busy share is
 * about 85 % while the waves are resident), not by memory or LDS latency:
 * what pays is fewer instructions per token and per output byte, and that is
 * what the choices above are for (no second parse, no pointer-doubling passes
 * over the group, word-wide output stores, address-space-qualified LDS
 * pointers so that offsets fold into the instructions).
 *
 * A round is abandoned, and the sequential decoder takes the same bits, when
 * it meets anything that decoder has a rule for: it needs the whole input
 * span plus 64 bytes inside the buffer and the produced bytes inside the
 * output buffer, and every distance inside the bytes already produced.
 */
#ifndef PAR_CB
#define PAR_CB 384u		/* input bits per lane and round: the span of 64 lanes must fit PAR_STAGE_BYTES */
#endif
#define PAR_LANECAP (PAR_CB / 2)	/* tokens one lane may find in its piece (2 bits each) */
/* where token k of lane l of a round lives in the wave's token scratch */
#ifdef TOK_LANE_MAJOR
#define TOK_AT(k, l) ((l) * PAR_LANECAP + (k))
#else
#define TOK_AT(k, l) ((k) * 64 + (l))
#endif
#ifndef PAR_TAIL
#define PAR_TAIL 8u		/* input bytes a round needs in front of it */
#endif
#define PAR_SCRATCH (64u * PAR_LANECAP)	/* u32 words per wave */
#define PAR_MAP_BYTES (256u + 128u)	/* tok_fetch: marks + tbase table */
enum { PAR_STOP = 0, PAR_OK = 1, PAR_EOB = 2 };

struct par_bits {
	u64 buf;
	u64 nxt;	/* the 8 bytes at nb, loaded one token ahead */
	u32 nb;		/* next input byte to load (offset in the staged span) */
	u32 cnt;
};

/* 'inp' is the round's input span staged in LDS (8-byte aligned, PAR_SPAN
 * bytes); b->nb and all bit positions of a round are relative to it */
#define PAR_SPAN (64u * PAR_CB / 8 + 80)
/* the staged span shares its LDS with the copy phase's scratch (a group's
 * tokens and its byte -> token map): whichever is larger */
#define PAR_MAP_CLEAR 2176u	/* bytes of the byte -> token map: what three 16-byte stores per lane clear */
#define PAR_COPY_BYTES (256u * 4 + PAR_MAP_CLEAR)
#define PAR_STAGE_BYTES (PAR_SPAN > PAR_COPY_BYTES ? (PAR_SPAN + 15u) & ~15u : PAR_COPY_BYTES)

static __device__ __forceinline__ u64 pb_load(const lu8 *inp, u32 nb)
{
	const lu32 *w = (const lu32 *)inp;
	u32 i = nb >> 2, sh = nb & 3;
	if (i + 2 >= PAR_SPAN / 4)	/* stopped lanes only; keeps reads inside */
		i = PAR_SPAN / 4 - 3;
	u32 a = w[i], c = w[i + 1], d = w[i + 2];
	return ((u64)__builtin_amdgcn_alignbyte(d, c, sh) << 32) |
	       __builtin_amdgcn_alignbyte(c, a, sh);
}

/* The bytes a refill adds start at nb whatever the bit count is, so they are
 * loaded right after the previous refill moved nb: the LDS round trip runs
 * beside the decode of the token in between instead of in front of the next
 * one (the refill rule itself is decompress_template.h's REFILL_BITS). */
static __device__ __forceinline__ void pb_refill(struct par_bits *b, const lu8 *inp)
{
	b->buf |= b->nxt << b->cnt;
	b->nb += (63 - b->cnt) >> 3;
	b->cnt |= 56;
	b->nxt = pb_load(inp, b->nb);
}

static __device__ __forceinline__ void pb_init(struct par_bits *b, const lu8 *inp, u32 pos)
{
	b->nb = pos >> 3;
	b->buf = 0;
	b->cnt = 0;
	b->nxt = pb_load(inp, b->nb);
	pb_refill(b, inp);
	b->buf >>= pos & 7;
	b->cnt -= pos & 7;
}

#define PB_POS(b) (8 * (b).nb - (b).cnt)

/* one token at the head of b->buf (>= 56 bits): kind, literal byte or
 * (length, distance), bits used; long codewords take the bit-serial path */
struct par_token {
	u32 kind, lit, length, dist, used;
	u32 e1;		/* the litlen table entry behind the first codeword */
};

/*
 * Codewords longer than the primary tables.  All lanes of a wave parse the
 * same block, so everything about the lengths beyond the table is
 * wave-uniform and fetched once per round (par_long_init).  Left-justified
 * to 16 bits, canonical codewords grow with their length (deflate's
 * canonical code, RFC 1951 3.2.2), so the length of a long codeword is the
 * number of per-length upper limits it reaches: one compare and one
 * conditional add per length, which also accumulate what turns the codeword
 * into its rank among the sorted symbols.
 */
struct par_long {
	u32 limit[16];	/* (first + count) << (16 - l): end of length l */
	u32 inc[16];	/* 1 | (adj[l + 1] - adj[l]) << 16 */
	u32 acc0;	/* from | adj[from] << 16; adj[l] = index[l] - first[l] */
};

/* `from` (wave-uniform) is FROM or FROM + 1: the first length beyond the
 * block's table; par_long_decode<FROM> walks from FROM either way */
template <u32 FROM> static __device__ __forceinline__ void
par_long_init(struct par_long *pl, const lcanon_t *cn, u32 from)
{
	u32 adj[16];
	const bool up = from > FROM;
#pragma unroll
	for (u32 l = 1; l < 16; l++) {
		if (l >= FROM) {
			const u32 first = bcast_first(cn->first[l]);
			const u32 count = bcast_first(cn->count[l]);
			const u32 index = bcast_first(cn->index[l]);
			pl->limit[l] = (first + count) << (16 - l);
			adj[l] = (index - first) & 0xFFFF;
		}
	}
	/* (a left-justified codeword is below 1 << 16: that limit is never reached) */
	pl->limit[FROM] = up ? 0xFFFFFFFFu : pl->limit[FROM];
	pl->acc0 = up ? (FROM + 1) | (adj[FROM + 1] << 16) : FROM | (adj[FROM] << 16);
#pragma unroll
	for (u32 l = 1; l < 15; l++)
		if (l >= FROM)
			pl->inc[l] = 1 + (((adj[l + 1] - adj[l]) & 0xFFFF) << 16);
}

template <u32 FROM, u32 MASK> static __device__ __forceinline__ u32
par_long_decode(const struct par_long *pl, const lu16 *sorted, u64 bits, u32 *len_ret)
{
	const u32 rev = __brev((u32)bits) >> 16;	/* first code bit on top */
	u32 acc = pl->acc0;
#pragma unroll
	for (u32 l = FROM; l < 15; l++)
		acc += rev >= pl->limit[l] ? pl->inc[l] : 0;
	const u32 len = acc & 15;
	*len_ret = len;
	/* lanes that are not on a long codeword compute nonsense: keep the
	 * read inside the stream's tables */
	return sorted[((acc >> 16) + (rev >> (16 - len))) & MASK];
}

/* (`act`: the lanes whose result is used.  The long-codeword paths run when ANY
 * lane needs them; a lane that has finished its piece still decodes - garbage -
 * and must not send the wave down them) */
static __device__ __forceinline__ struct par_token
par_decode(const slds_t *S, const shlds_t *SH,
	   const struct par_long *pll, const struct par_long *plo, u64 buf, bool act = true,
	   u32 lmask = (1u << LIT_TB) - 1, u32 omask = (1u << OFF_TB) - 1)
{
	struct par_token t;
	u32 e = S->lit_tab[(u32)buf & lmask];
	u32 cl = e & 15, kind = e & 0xC000, pay = (e >> 4) & 0x3FF;

	if (__ballot(cl == 0) & __ballot(act)) {
		u32 l2;
		u32 sym = par_long_decode<LIT_TB_MIN + 1, 511>(pll, S->lit_sorted, buf, &l2);
		if (cl == 0) {
			cl = l2;
			kind = sym < 256 ? K_LIT : sym == 256 ? K_EOB : K_LEN;
			pay = sym < 256 ? sym : sym - 257;
		}
	}
	/* base and extra-bit count of the length / distance symbol are computed:
	 * a handful of ALU operations where a table would put another LDS round
	 * trip on the token-to-token dependence chain */
	u64 bb = buf >> cl;
	/* what follows a literal is looked up now, beside the offset codeword of
	 * a match and not behind it: a step is two dependent LDS round trips
	 * for every kind of token (the callers pair two literals) */
	t.e1 = S->lit_tab[(u32)bb & lmask];
	u32 lbase, xb;
	len_sym(pay & 31, &lbase, &xb);
	t.length = lbase + ((u32)bb & ((1u << xb) - 1));
	bb >>= xb;
	u32 e2 = S->off_tab[(u32)bb & omask];
	u32 ol = e2 & 15, osym = e2 >> 4;
	if (__ballot(kind == K_LEN) & __ballot(ol == 0) & __ballot(act)) {
		u32 l2;
		u32 sym = par_long_decode<OFF_TB_MIN + 1, 31>(plo, S->off_sorted, bb, &l2);
		if (ol == 0) {
			ol = l2;
			osym = sym;
		}
	}
	u32 dbase, dxb;
	off_sym(osym & 31, &dbase, &dxb);
	t.dist = dbase + ((u32)(bb >> ol) & ((1u << dxb) - 1));
	t.kind = kind;
	t.lit = pay & 0xFF;
	t.used = cl + (kind == K_LEN ? xb + ol + dxb : 0);
	return t;
}

/*
 * The wave's most recent PAR_RW output bytes are mirrored in an LDS ring
 * (byte at output offset p lives at win[p % PAR_RW]): nearly every match
 * reads from there, an LDS round trip instead of an HBM one on the dependent
 * path of the copy phase.
 */
#ifndef PAR_RW
#define PAR_RW 2048u
#endif
#ifndef PAR_GBYTES
#define PAR_GBYTES 1088u	/* output bytes resolved per group (>= 4 x 258; the mirror holds the group and 256 more) */
#endif

static __device__ __forceinline__ u64 shfl_up64(u64 v)
{
	/* DPP wave_shr:1 (lane 0 keeps its own value) */
	u32 lo = __builtin_amdgcn_update_dpp((u32)v, (u32)v, 0x138, 0xF, 0xF, false);
	u32 hi = __builtin_amdgcn_update_dpp((u32)(v >> 32), (u32)(v >> 32), 0x138, 0xF, 0xF, false);
	return ((u64)hi << 32) | lo;
}

static __device__ __forceinline__ u64 readlane64(u64 v, u32 l)
{
	return ((u64)bcast_lane((u32)(v >> 32), l) << 32) | bcast_lane((u32)v, l);
}

typedef __attribute__((address_space(1))) u8 gu8;	/* output bytes in HBM */

/* bit `lane` of a wave-uniform 64-bit mask: the mask IS a lane predicate */
static __device__ __forceinline__ bool lane_bit(u64 uniform_mask)
{
	return __builtin_amdgcn_inverse_ballot_w64(uniform_mask);
}

/*
 * Output bytes one lane of the wave stored are read back by another lane
 * (match sources older than the LDS mirror; the sequential decoder's matches
 * behind a parallel round or a stored block).  Between such a store and such
 * a load the wave waits until every earlier vector-memory operation has
 * completed - s_waitcnt vmcnt(0): on gfx9 a store leaves the counter when the
 * L2 has it, and the wave's L1 is write-through - which is what the
 * compiler's memory model emits for a workgroup-scope release / acquire pair
 * (AMDGPUUsage, memory model gfx942, non-tgsplit mode).  Nothing here rests
 * on how the L1 treats a store that is still in flight.
 */
static __device__ __forceinline__ void global_stores_visible(void)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/*
 * A round whose passes do not converge.  The passes rest on a parse started
 * at a wrong bit falling in step with the true one within a few tokens; a
 * code whose codewords all have (nearly) one length never does - a dynamic
 * block over incompressible bytes: 256 literals of 8 bits, a few of 7 or 9 -
 * and then a pass makes exactly one more lane exact: 64 passes per round.
 * But such a parse cannot be far off either: a lane's true start lies in the
 * first PAR_PHASES bits of its piece (a literal overhangs the piece before by
 * less than its own length).  So, once, every lane behind the first lane
 * with a known start parses its piece from EACH of those starts (nothing
 * kept but where the parse ends, 6 bits per start in one 64-bit word), and
 * the chain - my end is the next lane's start - is followed through all
 * lanes on the scalar unit.  Returns this lane's start on that chain
 * (`cur` where the chain does not reach: a token that overhangs further, an
 * end of block); the caller's next pass parses from there and the passes'
 * own check (every start equals the end before it) decides as ever, so the
 * result does not depend on any of this.  Ten parses instead of sixty-four.
 * f: the first lane whose start changed, start_f its (exact) new start;
 * cend: where this lane's piece ends for the caller (a limit may cut it).
 */
#define PAR_PHASES 9u
#ifndef PAR_PHASE_MIN
#define PAR_PHASE_MIN 16u	/* lanes still out of step after the second parse that trigger it */
#endif
static __device__ u32
par_phase_starts(const slds_t *S, const shlds_t *SH, const struct par_long *pll,
		 const struct par_long *plo, const lu8 *span, u32 bpos0, u32 cb, u32 cend,
		 u32 lane, u32 NL, u32 f, u32 start_f, u32 cur,
		 u32 lmask = (1u << LIT_TB) - 1, u32 omask = (1u << OFF_TB) - 1)
{
	const u32 ps = bpos0 + lane * cb, pe = ps + cb;
	u32 o = start_f - (bpos0 + f * cb);	/* wave-uniform */
	if (o >= PAR_PHASES)
		return cur;
	u64 ends = ~0ull;	/* 63 = no usable end */
	const bool mine = lane >= f && lane < NL && cend == pe;
	for (u32 ph = 0; ph < PAR_PHASES; ph++) {
		struct par_bits b;
		bool stop = false;
		pb_init(&b, span, ps + ph);
		bool run = mine && PB_POS(b) < pe;	/* (tested at the end of the body: see par_round()) */
		while (__ballot(run)) {
			pb_refill(&b, span);
			const struct par_token t = par_decode(S, SH, pll, plo, b.buf, run, lmask, omask);
			const u32 e1 = t.e1;
			const bool two = t.kind == K_LIT && PB_POS(b) + t.used < pe &&
					 (e1 & 0xC000) == K_LIT && (e1 & 15) != 0;
			if (run) {
				u32 used = t.used + (two ? e1 & 15 : 0);
				if (t.kind == K_EOB) {
					stop = true;
					run = false;
				}
				b.buf >>= used;
				b.cnt -= used;
			}
			run = run && PB_POS(b) < pe;
		}
		const u32 over = PB_POS(b) - pe;
		const u64 code = mine && !stop && PB_POS(b) >= pe && over < PAR_PHASES ? over : 63;
		ends = (ends & ~(63ull << (6 * ph))) | (code << (6 * ph));
	}
	u32 res = cur;
	for (u32 i = f; i + 1 < NL; i++) {
		const u64 e = readlane64(ends, i);
		const u32 nx = (u32)(e >> (6 * o)) & 63;
		if (nx >= PAR_PHASES)
			break;
		o = nx;
		if (lane == i + 1)
			res = ps + o;
	}
	return res;
}

/*
 * Output positions [flushed, end) are in the LDS mirror and not yet in
 * memory: store the whole 16-byte units among them (units of the POSITION, the
 * alignment the mirror can be read with: a group of 1 KiB leaves in one store
 * per lane) and return the new 'flushed'.  The rest, less than a unit, waits
 * for the next group or the end of the round.
 */
static __device__ __forceinline__ u64
flush_ring(gu8 *gout, const lu8 *win, u64 flushed, u64 end, u32 lane)
{
	u64 a = (flushed + 15) & ~(u64)15;
	if (a > end)
		return flushed;
	if (flushed + lane < a)		/* up to 15 bytes in front of the first unit */
		gout[flushed + lane] = win[(u32)(flushed + lane) & (PAR_RW - 1)];
	const u64 e = end & ~(u64)15;
	if (e <= a)
		return a;
	const u32 nq = (u32)(e - a) >> 4;
	gu8 *dst = gout + a;
	const u32 a32 = (u32)a;
	for (u32 q = lane; q < nq; q += 64) {
		const uint4 v = *(const AS3 uint4 *)(win + ((a32 + 16 * q) & (PAR_RW - 1)));
		__builtin_memcpy(dst + 16 * q, &v, 16);
	}
	return e;
}

/*
 * Tokens g + 4 * lane .. + 3 of the round, in stream order, from the
 * lane-interleaved rows.  Lane l owns tokens [tbase, tbase + cnt); every
 * owner whose range starts inside the group drops its number at that token,
 * a running maximum spreads it (the tokens before the first mark belong to
 * the owner of token g).  Words past `total` read as 0.
 */
static __device__ __forceinline__ uint4
tok_fetch(const u32 *__restrict__ rows, lu8 *mk, const lu16 *tb, u32 tbase,
	  u32 cnt, u32 g, u32 total, u32 lane)
{
	((lu32 *)mk)[lane] = 0;
	const u64 holds = __ballot(cnt != 0 && tbase <= g && g - tbase < cnt);
	const u32 first = holds ? (u32)__builtin_ctzll(holds) + 1 : 1;
	wave_sync();
	if (cnt != 0 && tbase - g - 1 < 255u)	/* g < tbase < g + 256 */
		mk[tbase - g] = (u8)(lane + 1);
	wave_sync();
	const u32 m4 = ((const lu32 *)mk)[lane];
	u32 o[4];
	o[0] = m4 & 0xFF;
	o[1] = (m4 >> 8) & 0xFF;
	o[2] = (m4 >> 16) & 0xFF;
	o[3] = m4 >> 24;
	o[1] = o[1] > o[0] ? o[1] : o[0];
	o[2] = o[2] > o[1] ? o[2] : o[1];
	o[3] = o[3] > o[2] ? o[3] : o[2];
	/* exclusive running maximum of the lanes' last values */
	u32 c = o[3];
#define DPP_MAX(ctrl, rm, bc)                                                  \
	do {                                                                   \
		u32 t_ = __builtin_amdgcn_update_dpp(0, c, ctrl, rm, 0xF, bc);  \
		c = c > t_ ? c : t_;                                           \
	} while (0)
	DPP_MAX(0x111, 0xF, true);
	DPP_MAX(0x112, 0xF, true);
	DPP_MAX(0x114, 0xF, true);
	DPP_MAX(0x118, 0xF, true);
	DPP_MAX(0x142, 0xA, false);
	DPP_MAX(0x143, 0xC, false);
#undef DPP_MAX
	c = __builtin_amdgcn_update_dpp(0, c, 0x138, 0xF, 0xF, true);	/* wave_shr:1, lane 0 gets 0 */
	c = c > first ? c : first;
	/* the four owners' bases first (one LDS round trip for all four), then
	 * four unconditional loads: a token past `total` reads row 0 of its owner
	 * and is masked (as `i < total ? rows[..] : 0` each load sat in an EXEC
	 * section of its own behind its own LDS wait) */
	u32 w[4], own[4], base[4];
#pragma unroll
	for (u32 j = 0; j < 4; j++) {
		own[j] = (o[j] > c ? o[j] : c) - 1;
		base[j] = tb[own[j]];
	}
#pragma unroll
	for (u32 j = 0; j < 4; j++) {
		const u32 i = g + 4 * lane + j;
		const u32 row = i < total ? i - base[j] : 0;
		const u32 v = rows[TOK_AT(row, own[j])];
		w[j] = i < total ? v : 0;
	}
	return make_uint4(w[0], w[1], w[2], w[3]);
}

/*
 * One round.  bpos0: bit position (in inp) of the next token; out0: bytes
 * produced so far.  Returns PAR_STOP with the decoder's position unchanged,
 * or PAR_OK / PAR_EOB with *bpos_ret / *out_ret advanced (PAR_EOB: the
 * end-of-block symbol was consumed).  After PAR_STOP the sequential decoder
 * takes the same bits; the only case in which the round has already written
 * output by then is a match distance that reaches back before the stream,
 * found while its group is executed - the bytes written before it are the
 * ones the sequential decoder writes again before it reports the error.
 */
static __device__ u32
par_round(const u8 *inp, u64 in_n, u8 *outp, u64 out_avail,
	  const slds_t *S, const shlds_t *SH,
	  u32 *__restrict__ tok, lu8 *win, lu8 *stage, u64 ring_lo, u32 lane,
	  u64 bpos_abs, u64 out0, u64 *bpos_ret, u64 *out_ret, u32 ltb, u32 otb)
{
	const u32 lmask = (1u << ltb) - 1, omask = (1u << otb) - 1;	/* wave-uniform */
	/* lanes in this round: one PAR_CB-bit chunk each, up to the end of the
	 * input.  Bytes past the end are staged as zeros, exactly the implicit
	 * padding of the sequential decoder; a round whose exact parse ends
	 * beyond the input is abandoned below and left to that decoder (it is
	 * the one that knows the overread rules). */
	const u64 byte0 = bpos_abs >> 3;
	/* (rounds run up to the last bytes of the input: what is left behind them
	 * goes token by token through lane 0 - with a margin of 64 bytes that was
	 * some fifty tokens per stream, 2 % of a 64 KiB stream's instructions) */
	if (byte0 + PAR_TAIL > in_n)
		return PAR_STOP;
	u32 *__restrict__ tokS = tok;	/* [PAR_LANECAP][64]: row k holds every lane's k-th token */
	/* long pieces need fewer rounds and fewer passes per round (a parse
	 * falls in step within ~50 bits); when the input left would not fill
	 * the 64 lanes with them, shorter pieces keep more lanes busy */
	const u32 cb = in_n - byte0 >= 64 * (PAR_CB / 8) ? PAR_CB : 256u;
	const u64 room = (in_n - byte0 + cb / 8 - 1) / (cb / 8);
	const u32 NL = room < 64 ? (u32)room : 64;
	/* stage the span: 8-byte words, unaligned in HBM, aligned in LDS */
	{
		const u32 nw = (NL * (cb / 8) + 80) / 8;
		for (u32 w = lane; w < nw; w += 64) {
			const u64 pos = byte0 + 8 * w;
			*(lu64 *)(stage + 8 * w) = pos + 8 <= in_n ? ld8(inp + pos) :
						  load_in(inp, in_n, pos);
		}
		/* the staged words have arrived, and with them every store the
		 * wave issued before this round (the rounds before, the sequential
		 * decoder, a stored block): what this round reads back from the
		 * output below `out0` is in memory */
		global_stores_visible();
	}
	const lu8 *span = stage;	/* the parse reads the staged copy */
	const u32 bpos0 = (u32)bpos_abs & 7;	/* positions relative to the span */
	struct par_long pll, plo;
	par_long_init<LIT_TB_MIN + 1>(&pll, &S->lit, ltb + 1);
	par_long_init<OFF_TB_MIN + 1>(&plo, &S->off, otb + 1);
	const u32 cend = bpos0 + (lane + 1) * cb;
	u32 start = bpos0 + lane * cb, end = 0;
	u32 nbytes = 0, ntok = 0;
	bool eob = false, dirty = lane < NL;
	u32 K = NL - 1;		/* last lane of the round */
	bool has_eob = false;

	PROF_SEC_DECL;
	/* ---- sync passes ---- */
	for (u32 pass = 0; pass < 64; pass++) {
		struct par_bits b;
		pb_init(&b, span, start);
		if (dirty) {
			nbytes = 0;
			ntok = 0;
			eob = false;
		}
		/* (a lane runs while its position is inside its piece: tested
		 * where the position moves, at the end of the body, not at its
		 * top - that form went round once more, a whole step, only to
		 * find every lane at its end) */
		bool run = dirty && PB_POS(b) < cend;
		if (pass == 0) {
			/* The first pass is a guess for every lane but lane 0, and
			 * every lane parses again in the second: all it has to find
			 * is where each lane's parse ENDS.  Its loop keeps nothing
			 * else - no token words, lengths, distances, counts (the
			 * compiler drops what computes them) - and lane 0 records
			 * its tokens in the second pass with everybody else. */
			while (__ballot(run)) {
				PROF_SEC_ADD(1, 1);
				pb_refill(&b, span);
				const struct par_token t = par_decode(S, SH, &pll, &plo, b.buf, run, lmask, omask);
				const u32 e1 = t.e1;
				const bool two = t.kind == K_LIT && PB_POS(b) + t.used < cend &&
						 (e1 & 0xC000) == K_LIT && (e1 & 15) != 0;
				const u32 used = run ? t.used + (two ? e1 & 15 : 0) : 0;
				b.buf >>= used;
				b.cnt -= used;
				eob = eob || (run && t.kind == K_EOB);
				run = run && t.kind != K_EOB && PB_POS(b) < cend;
			}
		} else
		while (__ballot(run)) {
			PROF_SEC_ADD(1, 1);
			pb_refill(&b, span);
			struct par_token t = par_decode(S, SH, &pll, &plo, b.buf, run, lmask, omask);
			/* A literal takes a second one with it when that one starts
			 * inside the piece and its codeword is in the table: a pass
			 * lasts as long as its lane with the most tokens, and those
			 * are the lanes full of literals. */
			const u32 e1 = t.e1;
			const bool two = t.kind == K_LIT && PB_POS(b) + t.used < cend &&
					 (e1 & 0xC000) == K_LIT && (e1 & 15) != 0;
			if (run) {
				u32 used = t.used;
				if (t.kind == K_EOB) {
					eob = true;
					run = false;
				} else {
					/* row ntok of the lane-interleaved list: the 64
					 * lanes of an iteration write one 256-byte row */
					if (ntok < PAR_LANECAP)
						tokS[TOK_AT(ntok, lane)] = t.kind == K_LEN ?
							0x80000000u | t.length | (t.dist << 9) : t.lit;
					nbytes += t.kind == K_LEN ? t.length : 1;
					ntok++;
					if (two) {
						if (ntok < PAR_LANECAP)
							tokS[TOK_AT(ntok, lane)] = (e1 >> 4) & 0xFF;
						nbytes++;
						ntok++;
						used += e1 & 15;
					}
				}
				b.buf >>= used;
				b.cnt -= used;
			}
			run = run && PB_POS(b) < cend;
		}
		if (dirty)
			end = PB_POS(b);
		/* DPP wave_shr:1 (lane 0 keeps its own value) */
		u32 ns = __builtin_amdgcn_update_dpp(end, end, 0x138, 0xF, 0xF, false);
		if (lane == 0)
			ns = bpos0;
		dirty = (ns != start || pass == 0) && lane < NL;
		/* (the block ends in lane 0's piece - a stream of tiny blocks,
		 * programs/test_slow_decompression.c: only lane 0 parses again, to
		 * record its tokens; nothing behind it belongs to the round) */
		if (pass == 0 && bcast_lane(eob ? 1u : 0u, 0))
			dirty = lane == 0;
		const u64 dm = __ballot(dirty), em = __ballot(eob);
		const u64 exact = dm ? (1ull << __builtin_ctzll(dm)) - 1 : ~0ull;
		if (em & exact) {	/* end of block on the exact prefix */
			K = (u32)__builtin_ctzll(em & exact);
			has_eob = true;
			break;
		}
		if (!dm)
			break;
		if (pass == 1 && (u32)__builtin_popcountll(dm) >= PAR_PHASE_MIN) {
			/* the passes are not converging: see par_phase_starts() */
			const u32 f = (u32)__builtin_ctzll(dm);
			const u32 g = par_phase_starts(S, SH, &pll, &plo, span, bpos0, cb, cend, lane,
						       NL, f, bcast_lane(ns, f), ns, lmask, omask);
			if (lane > f && lane < NL) {
				ns = g;
				dirty = ns != start;
			}
		}
		start = ns;
	}
	PROF_SEC(0);
	/* ---- counts -> offsets; clip the round to the token scratch ---- */
	bool valid = lane <= K;
	u32 tcnt = valid ? ntok : 0;
	u32 tbase = wave_scan_incl(tcnt) - tcnt;
	{
		u64 vm = __ballot(valid);
		const u64 over = __ballot(lane <= K && ntok > PAR_LANECAP);
		if (over)	/* a lane whose row list overflowed, and all after it */
			vm &= (1ull << __builtin_ctzll(over)) - 1;
		const u32 nv = __builtin_popcountll(vm);	/* a prefix of lanes */
		if (nv == 0)
			return PAR_STOP;
		if (nv - 1 < K) {
			K = nv - 1;
			has_eob = false;
		}
		valid = lane <= K;
	}
	const u32 bcnt = valid ? nbytes : 0;
	const u32 obase = wave_scan_incl(bcnt) - bcnt;
	const u32 total_tok = bcast_lane(tbase + tcnt, K);
	const u64 total_bytes = bcast_lane(obase + bcnt, K);
	if (total_bytes > out_avail - out0)
		return PAR_STOP;
	const u64 end_bits = bcast_lane(end, K) - bpos0 + bpos_abs;
	if (end_bits > 8 * in_n)
		return PAR_STOP;

	/* Every lane's last parse started at its exact position, so the rows it
	 * wrote then are its tokens: no further parse.  Token i of the round
	 * (stream order) is row i - tbase[l] of the lane l whose range holds i;
	 * the copy phase finds l per group (tok_fetch). */
	lu8 *mk = stage + PAR_STAGE_BYTES;		/* [256] group token -> lane + 1 */
	lu16 *tb = (lu16 *)(mk + 256);			/* [64] tbase per lane */
	const u32 own_cnt = valid ? tcnt : 0;
	tb[lane] = (u16)tbase;
	wave_sync();
	/* ---- execute the tokens: up to 256 tokens / PAR_GBYTES bytes a group ----
	 * The copies of a group are resolved per output BYTE, not per token, 64
	 * bytes (a slot) at a time and in output order: byte b is a literal, or
	 * a copy of the byte dist before it.  That byte is final - in the LDS
	 * mirror of the recent output, or in the output itself when it is further
	 * back than the mirror reaches - unless it lies in the same slot; copies
	 * inside a slot (runs, short periods) are settled by pointer jumping over
	 * the 64 lanes, whatever the shape of the dependencies.  The bytes meet
	 * in the mirror; the output is written from there in whole words once
	 * per group (flush_ring). */
	{
		lu32 *tk = (lu32 *)stage;			/* [256] the group's tokens */
		/* (a group's tokens are numbered 0 .. 255 and its first byte starts
		 * token 0, so a cleared entry and "token 0" say the same) */
		lu8 *R = (lu8 *)((lu32 *)stage + 256);		/* [PAR_GBYTES] byte -> token of the group */
		gu8 *gout = (gu8 *)outp;
		u64 gbase = out0;
		u64 flushed = out0;	/* output below this is in memory */
		u64 safe_hi = out0;	/* ... and below this its stores have been waited for */
		u32 g = 0;		/* a multiple of 4: 16-byte token loads */
		/* four consecutive tokens per lane; the next group's are requested
		 * as soon as this group's extent is known, so their trip to the
		 * scratch runs beside the group's LDS work */
		uint4 tq_next = tok_fetch(tokS, mk, tb, tbase, own_cnt, 0, total_tok, lane);
		while (g < total_tok) {
			const u32 ti0 = g + 4 * lane;
			const uint4 tq = tq_next;
			const u32 tw4[4] = { tq.x, tq.y, tq.z, tq.w };
			u32 len4[4], lsum = 0;
#pragma unroll
			for (u32 j = 0; j < 4; j++) {
				len4[j] = ti0 + j >= total_tok ? 0 :
					  (tw4[j] >> 31) ? (tw4[j] & 0x1FF) : 1;
				lsum += len4[j];
			}
			const u32 incl0 = wave_scan_incl(lsum);
			/* the lanes whose tokens fit: a prefix */
			const bool fits = ti0 < total_tok && incl0 <= PAR_GBYTES;
			const u32 cnt = __builtin_popcountll(__ballot(fits));
			const u32 gtot = bcast_lane(incl0, cnt - 1);
			if (g + 4 * cnt < total_tok)
				tq_next = tok_fetch(tokS, mk, tb, tbase, own_cnt,
						    g + 4 * cnt, total_tok, lane);
			/* byte -> token: every token drops its number at its first
			 * byte, a running maximum over the bytes spreads it */
			/* (the whole map is cleared with three 16-byte stores per lane
			 * - 2 x 1024 + 128 bytes - instead of a loop of 2-byte stores
			 * over the group's bytes: 12 rounds of 8 instructions) */
			{
				static_assert(PAR_GBYTES <= PAR_MAP_CLEAR && PAR_MAP_CLEAR == 2176 &&
					      PAR_GBYTES + 256 <= PAR_RW && PAR_GBYTES >= 4 * 258,
					      "two full wave stores and one of eight lanes");
				const uint4 z = make_uint4(0, 0, 0, 0);
				AS3 uint4 *R16 = (AS3 uint4 *)R;
				R16[lane] = z;
				R16[64 + lane] = z;
				if (lane < 8)
					R16[128 + lane] = z;
			}
			wave_sync();
			if (lane < cnt) {
				u32 o = incl0 - lsum;
				*(AS3 uint4 *)&tk[4 * lane] = tq;	/* the lane's four token words */
#pragma unroll
				for (u32 j = 0; j < 4; j++) {
					if (len4[j])
						R[o] = (u8)(4 * lane + j);
					o += len4[j];
				}
			}
			wave_sync();
			/* a source `rel` bytes before the group is still in the mirror
			 * when the group's own bytes have not overwritten it and the
			 * mirror has been kept that far back */
			const u32 gb = (u32)gbase;
			u32 ring_rel = PAR_RW - gtot;
			if (gbase - ring_lo < ring_rel)
				ring_rel = (u32)(gbase - ring_lo);
			/* sources further back come from the output itself: base + a
			 * non-negative 32-bit lane offset (distances are <= 32768) */
			const gu8 *gfar = (const gu8 *)((uintptr_t)gout + gbase - 32768);
			/* a distance that reaches back before the stream: possible only
			 * in the first 32 KiB */
			const u32 back_max = gbase < 32768 ? (u32)gbase : 32768u;
			u64 badm = 0;	/* lanes whose distance reaches back before the stream */
			/* The slots (64 bytes each) are resolved in order, so the
			 * source of a byte is final in the mirror when its slot is
			 * reached, unless it lies in the same slot.  The token lookup
			 * of SB slots is done together (its LDS reads and the reads
			 * from the output are independent of the mirror); then every
			 * slot reads its sources, settles the copies inside itself
			 * (pointer jumping over the 64 lanes, only when there are
			 * any) and writes its bytes. */
			enum { SB = 4 };
			u32 carry = 0;
			PROF_SEC(2);
			PROF_SEC_ADD(6, 1);
			for (u32 s0 = 0; s0 < gtot; s0 += 64 * SB) {
				u32 own[SB], vfar[SB];
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					/* (an index past the group reads its last entry and is
					 * masked: as `bi < gtot ? R[bi] : 0` each of the four
					 * reads sat in an EXEC section of its own with its own
					 * wait) */
					const u32 bi = s0 + 64 * k + lane;
					const u32 r = R[bi < gtot ? bi : gtot - 1];
					own[k] = bi < gtot ? r : 0;
				}
#define DPP_MAX(k, ctrl, rm, bc)                                               \
	do {                                                                   \
		u32 t_ = __builtin_amdgcn_update_dpp(0, own[k], ctrl, rm, 0xF, bc); \
		own[k] = own[k] > t_ ? own[k] : t_;                            \
	} while (0)
#pragma unroll
				for (u32 k = 0; k < SB; k++)
					DPP_MAX(k, 0x111, 0xF, true);
#pragma unroll
				for (u32 k = 0; k < SB; k++)
					DPP_MAX(k, 0x112, 0xF, true);
#pragma unroll
				for (u32 k = 0; k < SB; k++)
					DPP_MAX(k, 0x114, 0xF, true);
#pragma unroll
				for (u32 k = 0; k < SB; k++)
					DPP_MAX(k, 0x118, 0xF, true);
#pragma unroll
				for (u32 k = 0; k < SB; k++)
					DPP_MAX(k, 0x142, 0xA, false);
#pragma unroll
				for (u32 k = 0; k < SB; k++)
					DPP_MAX(k, 0x143, 0xC, false);
#undef DPP_MAX
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					own[k] = own[k] > carry ? own[k] : carry;
					carry = bcast_lane(own[k], 63);
				}
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					/* (the group's first byte starts token 0 and the running
					 * maximum carries on) */
					const u32 bi = s0 + 64 * k + lane;
					const u32 w = tk[own[k]];	/* token word */
					own[k] = bi < gtot ? w : 0;
				}
				/* bytes whose source is older than the mirror.  The tests are
				 * one ballot per compare, combined on the scalar unit (a
				 * ballot of a compound predicate goes through a 0 / 1 detour
				 * in a vector register), and the four slots share ONE
				 * section: nothing of it runs when the batch has no such byte */
				u64 mfar[SB], anyfar = 0, anyneed = 0;
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					const u32 bi = s0 + 64 * k + lane, tw = own[k];
					const u32 dist = (tw >> 9) & 0xFFFF;
					const u32 back = dist - bi;	/* bytes in front of the group (when dist > bi) */
					const u64 before = __ballot((s32)tw < 0) & __ballot(dist > bi);
					const u64 toofar = before & __ballot(back > back_max);
					mfar[k] = before & ~toofar & __ballot(back > ring_rel);
					badm |= toofar;
					anyfar |= mfar[k];
					/* a source at or above safe_hi was stored by this wave
					 * after its last wait: see below */
					anyneed |= mfar[k] & __ballot(back <= (u32)(gbase - safe_hi));
					vfar[k] = 0x100;	/* not a byte: no far source */
				}
				if (anyfar) {
					/* The source may be a byte another lane of this wave
					 * stored earlier in THIS round (flush_ring); everything
					 * below safe_hi was stored before a wait.  Only a source
					 * at or above it - rare: it must be older than the mirror
					 * and younger than the last wait - makes the wave wait for
					 * its stores (and for the token rows requested ahead)
					 * before it loads. */
					if (anyneed) {
						global_stores_visible();
						safe_hi = flushed;
					}
#pragma unroll
					for (u32 k = 0; k < SB; k++) {
						const u32 bi = s0 + 64 * k + lane;
						const u32 dist = (own[k] >> 9) & 0xFFFF;
						if (lane_bit(mfar[k]))
							vfar[k] = gfar[bi + 32768u - dist];
					}
				}
				/* copies inside a slot: where each lane's byte finally comes
				 * from.  That depends on the tokens alone, so the SB slots'
				 * pointer chains are jumped together (their latencies
				 * overlap) before the slots' bytes are settled in order. */
				u32 root[SB];
				bool any_intra = false;
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					const u32 tw = own[k], dist = (tw >> 9) & 0xFFFF;
					const bool intra = (tw >> 31) && dist <= lane;
					root[k] = intra ? lane - dist : lane;
					any_intra |= intra;
				}
				if (__ballot(any_intra)) {
					for (;;) {
						/* the four permutes go out together, one wait (as
						 * `pp = permute; ch |= pp != root` per slot each
						 * compare waited for its own permute) */
						u32 pp[SB];
#pragma unroll
						for (u32 k = 0; k < SB; k++)
							pp[k] = (u32)__builtin_amdgcn_ds_bpermute(
									(int)(root[k] << 2), (int)root[k]);
						asm volatile("" :: "v"(pp[0]), "v"(pp[1]), "v"(pp[2]), "v"(pp[3]));
						u64 chm = 0;
#pragma unroll
						for (u32 k = 0; k < SB; k++) {
							chm |= __ballot(pp[k] != root[k]);
							root[k] = pp[k];
						}
						if (!chm)
							break;
					}
				}
#if defined(LDA_PROFILE) && defined(LDA_PROFILE_COUNTS)
				/* the wait for the far sources on its own (the profile build
				 * only: it also waits for the token rows requested ahead) */
				PROF_SEC(3);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				PROF_SEC(4);
				PROF_SEC_ADD(7, 1);
#endif
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					const u32 bi = s0 + 64 * k + lane, tw = own[k];
					const u32 dist = (tw >> 9) & 0xFFFF;
					const bool match = (tw >> 31) != 0;	/* false past gtot */
					/* every lane reads the mirror (the index is always
					 * inside it); matches from outside the slot use it */
					u32 wv = win[(gb + bi - dist) & (PAR_RW - 1)];
					/* (kept out of the branches below: as part of one the
					 * read, and its wait, sat inside an EXEC section) */
					asm volatile("" : "+v"(wv));
					u32 v = match ? wv : tw & 0xFF;
					v = vfar[k] < 0x100 ? vfar[k] : v;
					if (__ballot(root[k] != lane))
						v = (u32)__builtin_amdgcn_ds_bpermute((int)(root[k] << 2), (int)v);
					if (bi < gtot)
						win[(gb + bi) & (PAR_RW - 1)] = (u8)v;
				}
			}
			wave_sync();
			PROF_SEC(3);
			if (badm) {
				/* Invalid stream.  The sequential decoder takes the
				 * round's bits again and reports it at the token where
				 * it belongs; what the earlier groups wrote is what it
				 * will write again, and the mirror is not used across
				 * an abandoned round. */
				return PAR_STOP;
			}
			gbase += gtot;
			flushed = flush_ring(gout, win, flushed, gbase, lane);
			PROF_SEC(5);
			g += 4 * cnt;
		}
		/* the last bytes (less than a word) */
		if (flushed + lane < gbase)
			gout[flushed + lane] = win[(u32)(flushed + lane) & (PAR_RW - 1)];
		wave_sync();
	}
	PROF_SEC_FLUSH8(16);
	*bpos_ret = end_bits;
	*out_ret = out0 + total_bytes;
	return has_eob ? PAR_EOB : PAR_OK;
}

/*
 * One workgroup's worth of streams: streams blk * lpw .. blk * lpw + lpw - 1
 * on lanes 0..lpw-1.  With par != 0 (lpw == 1: a wave per stream) the token
 * phase of a block runs sub-block parallel rounds (par_round above) wherever
 * the stream is far enough from both buffer ends; everything else - headers,
 * stored blocks, the last bytes of a stream, every error path - is the
 * sequential decoder below, so result codes do not depend on the mode.
 */
static __device__ __forceinline__ void
inflate_block(u64 blk, lu8 *lds_raw, u32 par, u32 *__restrict__ tok,
	      u64 n_chunks, int format, u32 lpw,
	      const u8 *__restrict__ in_base,
	      const u64 *__restrict__ in_offsets,
	      const u64 *__restrict__ in_nbytes,
	      u8 *__restrict__ out_base,
	      const u64 *__restrict__ out_offsets,
	      const u64 *__restrict__ out_avail_arr,
	      s32 *__restrict__ results,
	      u64 *__restrict__ actual_in,	/* incl. container header */
	      u64 *__restrict__ actual_out)
{
	slds_t *SL = (slds_t *)lds_raw;
	const u32 lane = threadIdx.x;
	const u64 c = blk * lpw + lane;
	const bool owner = lane < lpw && c < n_chunks;
	slds_t *S = &SL[lane < lpw ? lane : 0];
	shlds_t *SH = (shlds_t *)&SL[lpw];
	if (lane < 32) {
		u32 b, x;
		len_sym(lane, &b, &x);
		SH->len_tab[lane] = b | (x << 16);
		off_sym(lane, &b, &x);
		SH->dist_tab[lane] = b | (x << 16);
	}
	wave_sync();
	PROF_DECL;
	PROF_START();
#ifdef LDA_PROFILE
	unsigned long long pc_reg = 0, pc_pipe = 0, pc_slow = 0, pc_slowbytes = 0;
	unsigned long long pa_dec = 0, pa_flush = 0, pa_out = 0, pa_rounds = 0, pt_ = 0, pt2_ = 0;
#define SEG_T0() do { pt_ = __builtin_readcyclecounter(); } while (0)
#define SEG_ADD(acc) do { unsigned long long n_ = __builtin_readcyclecounter(); acc += n_ - pt_; pt_ = n_; } while (0)
#else
#define SEG_T0() do { } while (0)
#define SEG_ADD(acc) do { } while (0)
#endif

	const u8 *inp = in_base;
	u64 in_n = 0, out_avail = 0;
	u8 *outp = out_base;
	u32 hdr = 0;
	s32 result = LDA_SUCCESS;
	u32 state = ST_DONE;

	if (owner) {
		inp = in_base + in_offsets[c];
		in_n = in_nbytes[c];
		outp = out_base + out_offsets[c];
		out_avail = out_avail_arr[c];
		state = ST_HDR;
		/* ---- container header ---- */
		if (format == LDA_FMT_ZLIB) {
			/* lib/zlib_decompress.c:45-72 */
			if (in_n < 6) {
				result = LDA_BAD_DATA;
			} else {
				u32 h = ((u32)inp[0] << 8) | inp[1];
				if (h % 31 || ((h >> 8) & 0xF) != 8 || (h >> 12) > 7 ||
				    ((h >> 5) & 1))
					result = LDA_BAD_DATA;
				hdr = 2;
				in_n -= 6;
			}
		} else if (format == LDA_FMT_GZIP) {
			/* lib/gzip_decompress.c:45-107 */
			if (in_n < 18) {
				result = LDA_BAD_DATA;
			} else if (inp[0] != 0x1F || inp[1] != 0x8B || inp[2] != 8 ||
				   (inp[3] & 0xE0)) {
				result = LDA_BAD_DATA;
			} else {
				u32 flg = inp[3];
				u64 p = 10, end = in_n;
				if (flg & 0x04) {
					u32 xlen = inp[p] | ((u32)inp[p + 1] << 8);
					p += 2;
					if (end - p < (u64)xlen + 8)
						result = LDA_BAD_DATA;
					p += xlen;
				}
				if (result == LDA_SUCCESS && (flg & 0x08)) {
					while (inp[p++] != 0 && p != end)
						;
					if (end - p < 8)
						result = LDA_BAD_DATA;
				}
				if (result == LDA_SUCCESS && (flg & 0x10)) {
					while (inp[p++] != 0 && p != end)
						;
					if (end - p < 8)
						result = LDA_BAD_DATA;
				}
				if (result == LDA_SUCCESS && (flg & 0x02)) {
					p += 2;
					if (end - p < 8)
						result = LDA_BAD_DATA;
				}
				if (result == LDA_SUCCESS) {
					hdr = (u32)p;
					in_n = end - 8 - p;
				}
			}
		}
		if (result != LDA_SUCCESS)
			state = ST_DONE;
		else
			inp += hdr;
	}

	u64 bitbuf = 0, rpos = 0, out_pos = 0, filled = 0;
	u32 bitcnt = 0, final_block = 0, nlit = 0, noff = 0;
	u64 stored_left = 0;
	bool static_loaded = false;	/* the stream's tables are the static codes' */
	/* the current block has the larger tables (see LIT_TB); where its header
	 * began (bits, modulo 2^32: only differences below TB_BIG_BLOCK matter) */
	bool tb_big = false;
	u32 blk_at = 0;
#define ltb (tb_big ? LIT_TB : LIT_TB_MIN)
#define otb (tb_big ? OFF_TB : OFF_TB_MIN)
	u64 pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0;	/* loaded, not yet stored */
	u8 *pend_dst = outp;
	u32 pend_n = 0, pend_len = 0;
	/* the lane's last hist_n (<= 8) output bytes, newest in the top byte:
	 * matches at distance <= hist_n never touch memory for their source */
	u64 hist = 0;
	u32 hist_n = 0;
	const u64 limit_bits = 8 * in_n + 8;	/* see header comment */
	(void)limit_bits;

	while (__ballot(state != ST_DONE)) {
		/* ------------ block headers (lanes that need one) ------------ */
		if (state == ST_HDR) {
			/* decompress_template.h:72-83 */
			ENSURE_INPUT();
			REFILL();
			if (CONSUMED() > limit_bits) {
				result = LDA_BAD_DATA;
				state = ST_DONE;
			} else {
				final_block = (u32)bitbuf & 1;
				u32 btype = ((u32)bitbuf >> 1) & 3;
				/* the larger tables where they will be used enough to pay
				 * for their entries: a dynamic block with input ahead of
				 * it, behind a block that was not a small one */
				{
					const u64 at = CONSUMED();
					const bool was_small = at != 0 && (u32)at - blk_at < 8 * TB_BIG_BLOCK;
					tb_big = btype == 2 && !was_small && in_n - (at >> 3) >= TB_BIG_BLOCK;
					blk_at = (u32)at;
				}
				if (btype == 0) {
					/* stored: decompress_template.h:247-285 */
					CONSUME(3);
					u64 pos = (CONSUMED() + 7) / 8;
					if (pos > in_n || in_n - pos < 4) {
						result = LDA_BAD_DATA;
						state = ST_DONE;
					} else {
						u32 len = inp[pos] | ((u32)inp[pos + 1] << 8);
						u32 nlen = inp[pos + 2] | ((u32)inp[pos + 3] << 8);
						pos += 4;
						if (len != (nlen ^ 0xFFFF)) {
							result = LDA_BAD_DATA;
							state = ST_DONE;
						} else if (len > out_avail - out_pos) {
							result = LDA_INSUFFICIENT_SPACE;
							state = ST_DONE;
						} else if (len > in_n - pos) {
							result = LDA_BAD_DATA;
							state = ST_DONE;
						} else {
							rpos = pos;
							stored_left = len;
							state = ST_STORED;
						}
					}
				} else if (btype == 3) {
					result = LDA_BAD_DATA;
					state = ST_DONE;
				} else if (btype == 1) {
					/* static codes: decompress_template.h:313-326; the
					 * tables of the block before are kept when that one
					 * was static too (static_codes_loaded, :303-311 - a
					 * stream of tiny static blocks is the case of
					 * programs/test_slow_decompression.c:18-30) */
					CONSUME(3);
					if (static_loaded) {
						state = ST_TOK;
					} else {
						for (u32 i = 0; i < 320; i++)
							S->lens[i] = i < 144 ? 8 : i < 256 ? 9 :
								     i < 280 ? 7 : i < 288 ? 8 : 5;
						nlit = 288;
						noff = 32;
						/* (no static codeword is longer than 9 / 5 bits:
						 * tb_big is false) */
						state = ST_TABLES;
						static_loaded = true;	/* taken back if the build fails (ST_TABLES) */
					}
				} else {
					/* dynamic header: decompress_template.h:85-245 (its
					 * scratch shares the tables' LDS) */
					/* (the 19 precode lengths go to LDS scratch behind
					 * lens[]: a private array indexed by c_pre_perm[i]
					 * compiles into a 19-way select chain per store) */
					lu8 *plens = (lu8 *)S->lit_tab + 480;
					static_loaded = false;

					nlit = 257 + (((u32)bitbuf >> 3) & 31);
					noff = 1 + (((u32)bitbuf >> 8) & 31);
					u32 npre = 4 + (((u32)bitbuf >> 13) & 15);
					for (u32 i = 0; i < 5; i++)
						((lu32 *)plens)[i] = 0;
					plens[c_pre_perm[0]] = ((u32)bitbuf >> 17) & 7;
					CONSUME(20);
					ENSURE_INPUT();
					REFILL();
					if (CONSUMED() > limit_bits) {
						result = LDA_BAD_DATA;
						state = ST_DONE;
					} else {
						for (u32 i = 1; i < npre; i++) {
							plens[c_pre_perm[i]] = (u32)bitbuf & 7;
							CONSUME(3);
						}
						state = ST_PRETAB;
					}
				}
			}
		}
		/* ------------ precode tables: one stream at a time, all lanes ------------ */
		{
			u64 need = __ballot(state == ST_PRETAB);
			while (need) {
				const u32 who = (u32)__builtin_ctzll(need);
				need &= need - 1;
				slds_t *T = &SL[who];
				wave_sync();
				const bool ok = build_precode_coop(T->pre_tab, (lu8 *)T->lit_tab + 480, lane);
				wave_sync();
				if (lane == who) {
					if (ok) {
						state = ST_LENS;
					} else {
						result = LDA_BAD_DATA;
						state = ST_DONE;
					}
				}
			}
		}
		/* ------------ code length runs (lanes whose precode is built) ------------ */
		if (state == ST_LENS) {
			/* code length runs (:150-245); bitcnt tracks the
			 * reference's bitsleft: both were topped up at
			 * the same points */
			u32 i = 0, total = nlit + noff, bad = 0;
			do {
				if (bitcnt < 14) {
					ENSURE_INPUT();
					REFILL();
					if (CONSUMED() > limit_bits) {
						bad = 1;
						break;
					}
				}
				u32 e = S->pre_tab[(u32)bitbuf & 127];
				CONSUME(e & 15);
				u32 presym = e >> 4;
				if (presym < 16) {
					S->lens[i++] = (u8)presym;
					continue;
				}
				u32 rep, val = 0;
				if (presym == 16) {
					if (i == 0) {
						bad = 1;
						break;
					}
					val = S->lens[i - 1];
					rep = 3 + ((u32)bitbuf & 3);
					CONSUME(2);
				} else if (presym == 17) {
					rep = 3 + ((u32)bitbuf & 7);
					CONSUME(3);
				} else {
					rep = 11 + ((u32)bitbuf & 127);
					CONSUME(7);
				}
				for (u32 k = 0; k < rep; k++)
					S->lens[i + k] = (u8)val;
				i += rep;
			} while (i < total);
			if (bad || i != total) {
				result = LDA_BAD_DATA;
				state = ST_DONE;
			} else {
				state = ST_TABLES;
			}
		}
		PROF_MARK(0);

		/* ------------ decode tables: one stream at a time, all lanes ------------ */
		{
			u64 need = __ballot(state == ST_TABLES);
			while (need) {
				u32 who = (u32)__builtin_ctzll(need);
				need &= need - 1;
				slds_t *T = &SL[who];
				u32 t_nlit = bcast_lane(nlit, who);
				u32 t_noff = bcast_lane(noff, who);
				const bool t_big = bcast_lane(tb_big ? 1u : 0u, who) != 0;
				const u32 t_ltb = t_big ? LIT_TB : LIT_TB_MIN, t_otb = t_big ? OFF_TB : OFF_TB_MIN;
				u32 s_lit, s_off;
				wave_sync();
				/* both sorts read lens[] before any table overwrites it;
				 * offset first as in the reference (:331-332) */
				bool ok = build_table_coop(T->lens + t_nlit, t_noff, t_otb,
							   false, &T->off, T->off_sorted,
							   lane, &s_off);
				ok = build_table_coop(T->lens, t_nlit, t_ltb, true, &T->lit,
						      T->lit_sorted, lane, &s_lit) && ok;
				wave_sync();
				if (ok) {
					fill_table(T->off_tab, t_otb, false, &T->off,
						   T->off_sorted, s_off, lane);
					fill_table(T->lit_tab, t_ltb, true, &T->lit,
						   T->lit_sorted, s_lit, lane);
				}
				wave_sync();
				if (lane == who) {
					if (ok) {
						state = ST_TOK;
					} else {
						result = LDA_BAD_DATA;
						state = ST_DONE;
						/* (the tables are no code's now; the stream ends
						 * here, but the flag must not outlive them) */
						static_loaded = false;
					}
				}
			}
		}
		PROF_MARK(1);

		/* ------------ stored blocks ------------ */
		if (par) {
			/* wave per stream: lane 0 holds the stream, all 64 lanes copy
			 * (8 bytes each, 512 per step; the odd tail byte by byte) */
			if (bcast_first(state == ST_STORED ? 1u : 0u)) {
				const u8 *src = (const u8 *)bcast64((u64)(uintptr_t)(inp + rpos));
				u8 *dst = (u8 *)bcast64((u64)(uintptr_t)(outp + out_pos));
				const u64 len = bcast64(stored_left);
				for (u64 k = 8 * (u64)lane; k + 8 <= len; k += 512) {
					u64 v;
					__builtin_memcpy(&v, src + k, 8);
					__builtin_memcpy(dst + k, &v, 8);
				}
				if ((len & ~7ull) + lane < len)
					dst[(len & ~7ull) + lane] = src[(len & ~7ull) + lane];
				global_stores_visible();	/* later matches read these bytes */
				if (state == ST_STORED) {
					/* the register history describes the bytes before
					 * the stored block: a short-distance match of the
					 * next block must not be served from it */
					if (stored_left)
						hist_n = 0;
					out_pos += stored_left;
					rpos += stored_left;
					bitbuf = 0;
					bitcnt = 0;
					filled = rpos & ~(u64)63;	/* restart the input ring */
					state = final_block ? ST_DONE : ST_HDR;
				}
			}
		} else if (state == ST_STORED) {	/* lane per stream: the lane copies its bytes */
			const u8 *src = inp + rpos;
			u8 *dst = outp + out_pos;
			u64 k = 0;
			for (; k + 8 <= stored_left; k += 8) {
				u64 v;
				__builtin_memcpy(&v, src + k, 8);
				__builtin_memcpy(dst + k, &v, 8);
			}
			for (; k < stored_left; k++)
				dst[k] = src[k];
			if (stored_left)
				hist_n = 0;	/* see the wave mode above */
			out_pos += stored_left;
			rpos += stored_left;
			bitbuf = 0;
			bitcnt = 0;
			filled = rpos & ~(u64)63;	/* restart the input ring */
			state = final_block ? ST_DONE : ST_HDR;
		}

		/* ------------ sub-block parallel rounds (wave per stream) ------------ */
		if (par) {
			/* the ring mirrors the output only while parallel rounds
			 * follow one another */
			u64 ring_lo = ~0ull;
			for (;;) {
				if (!bcast_first(state == ST_TOK ? 1u : 0u))
					break;
				if (lane == 0)
					FLUSH_PENDING();
				const u64 bpos0 = bcast64(CONSUMED());
				const u64 o0 = bcast64(out_pos);
				const u8 *inp0 = (const u8 *)bcast64((u64)(uintptr_t)inp);
				u8 *outp0 = (u8 *)bcast64((u64)(uintptr_t)outp);
				u64 nb = 0, no = 0;
				if (ring_lo == ~0ull)
					ring_lo = o0;
#if PAR_PRIO
				{
					/* A launch of one stream per wave slot lasts as long
					 * as its slowest stream.  The waves of a SIMD share its
					 * issue slots; the one with most of its input still
					 * ahead gets them first, so the streams of a CU finish
					 * closer together (s_setprio takes an immediate). */
					const u64 inn = bcast64(in_n), left = inn - (bpos0 >> 3);
#if PAR_PRIO == 2
					const u32 q = left > 18432 ? 3 : left > 10240 ? 2 : left > 4096 ? 1 : 0;
#else
					const u32 q = inn ? (u32)(4 * left / inn) : 0;
#endif
					if (q >= 3)
						__builtin_amdgcn_s_setprio(3);
					else if (q == 2)
						__builtin_amdgcn_s_setprio(2);
					else if (q == 1)
						__builtin_amdgcn_s_setprio(1);
					else
						__builtin_amdgcn_s_setprio(0);
				}
#endif
				u32 pr = par_round(inp0, bcast64(in_n), outp0,
						   bcast64(out_avail), &SL[0], SH, tok,
						   (lu8 *)(SH + 1), (lu8 *)(SH + 1) + PAR_RW,
						   ring_lo, lane, bpos0, o0,
						   &nb, &no, bcast_first(ltb), bcast_first(otb));

				PROF_COUNT(12 + pr, 1);
				if (pr == PAR_STOP)
					break;
				if (lane == 0) {
					/* back to the sequential decoder's state */
					rpos = nb >> 3;
					bitbuf = 0;
					bitcnt = 0;
					filled = rpos & ~(u64)63;
					ENSURE_INPUT();
					REFILL();
					CONSUME((u32)nb & 7);
					out_pos = no;
					hist_n = 0;
					if (pr == PAR_EOB)
						state = final_block ? ST_DONE : ST_HDR;
				}
				if (pr == PAR_EOB)
					break;
			}
			/* lane 0's sequential decoder may copy from what all lanes
			 * stored in the rounds above (once per block; the staged
			 * input of a following round waits the same way) */
			global_stores_visible();
		}

		/* ------------ fast token loop ------------
		 * While every active stream has >= 16 input bytes and >= 272
		 * output bytes left, none of the end-of-buffer rules can fire
		 * (the reference's fastloop, decompress_template.h:344-671, rests
		 * on the same argument), so the per-token checks reduce to
		 * "distance <= bytes written".  Anything unusual - a codeword
		 * longer than the table, a bad distance, a stream near one of its
		 * ends, a new block - leaves to the general loop below, which
		 * re-reads the same token with all checks. */
		for (;;) {
			bool tok = state == ST_TOK;
			bool elig = tok && rpos + 16 < in_n && out_pos + 272 <= out_avail;
			if (!__ballot(tok) || __ballot(tok != elig) ||
			    __ballot(state == ST_HDR || state == ST_TABLES))
				break;
			bool punt = false;
			if (tok) {
				if (filled < rpos + 64)
					ENSURE_INPUT();
				REFILL();
				u32 e = S->lit_tab[(u32)bitbuf & ((1u << ltb) - 1)];
				FLUSH_PENDING();
				u32 cl = e & 15;
				if (cl == 0) {
					punt = true;
				} else if ((e & 0xC000) == K_LIT) {
					u32 pay = (e >> 4) & 0xFF;
					CONSUME(cl);
					u32 e1 = S->lit_tab[(u32)bitbuf & ((1u << ltb) - 1)];
					if ((e1 & 15) && (e1 & 0xC000) == K_LIT) {
						u16 two = (u16)(pay | (((e1 >> 4) & 0xFF) << 8));
						__builtin_memcpy(outp + out_pos, &two, 2);
						out_pos += 2;
						CONSUME(e1 & 15);
						hist = (hist >> 16) | ((u64)two << 48);
						hist_n = hist_n + 2 > 8 ? 8 : hist_n + 2;
					} else {
						outp[out_pos++] = (u8)pay;
						hist = (hist >> 8) | ((u64)pay << 56);
						hist_n = hist_n + 1 > 8 ? 8 : hist_n + 1;
					}
				} else if ((e & 0xC000) == K_EOB) {
					CONSUME(cl);
					state = final_block ? ST_DONE : ST_HDR;
				} else {
					/* match: decode on a copy of the bit buffer, commit
					 * only if the distance is valid */
					u64 bb = bitbuf >> cl;
					u32 lt = SH->len_tab[(e >> 4) & 31];
					u32 xb = lt >> 16;
					u32 length = (lt & 0xFFFF) + ((u32)bb & ((1u << xb) - 1));
					bb >>= xb;
					u32 used = cl + xb;
					u32 e2 = S->off_tab[(u32)bb & ((1u << otb) - 1)];
					u32 ol = e2 & 15;
					u32 dt = SH->dist_tab[(e2 >> 4) & 31];
					bb >>= ol;
					xb = dt >> 16;
					u32 dist = (dt & 0xFFFF) + ((u32)bb & ((1u << xb) - 1));
					bb >>= xb;
					used += ol + xb;
					if (ol == 0 || dist > out_pos) {
						punt = true;
					} else {
						bitbuf = bb;
						bitcnt -= used;
						if (dist <= hist_n && length <= 8) {
							u64 pat = hist >> (8 * (8 - dist));
							u32 sh = 8 * dist;
							if (sh < 64)
								pat |= pat << sh;
							sh *= 2;
							if (sh < 64)
								pat |= pat << sh;
							sh *= 2;
							if (sh < 64)
								pat |= pat << sh;
							st8(outp + out_pos, pat);
							hist = length == 8 ? pat :
							       (hist >> (8 * length)) |
							       (pat << (8 * (8 - length)));
							hist_n = hist_n + length > 8 ? 8 : hist_n + length;
						} else {
							u32 nwords = (length + 7) >> 3;
							if (nwords <= 4 && nwords <= (dist >> 3)) {
								const u8 *src = outp + out_pos - dist;
								pend_dst = outp + out_pos;
								pend_n = nwords;
								pend_len = length;
								pv0 = ld8(src);
								if (nwords > 1)
									pv1 = ld8(src + 8);
								if (nwords > 2)
									pv2 = ld8(src + 16);
								if (nwords > 3)
									pv3 = ld8(src + 24);
							} else {
								hist_n = 0;
								copy_match(outp, out_pos, out_avail, dist, length);
							}
						}
						out_pos += length;
					}
				}
			}
			if (__ballot(punt))
				break;
		}

		/* ------------ tokens: one per lane per round (all checks) ------------ */
		for (u32 round = 0; round < 4; round++) {
			u64 tk = __ballot(state == ST_TOK);
			if (!tk)
				break;
			/* leave as soon as someone needs a header or tables */
			if (__ballot(state == ST_HDR || state == ST_TABLES))
				break;
#ifdef LDA_PROFILE
			{	/* whole-iteration time, all iterations of this wave */
				unsigned long long n2_ = __builtin_readcyclecounter();
				if (pa_out)
					pa_flush += n2_ - pt2_;
				pt2_ = n2_;
				pa_out++;
			}
#endif
			if (state != ST_TOK)
				continue;
			/* generic_loop: decompress_template.h:680-738 */
			SEG_T0();
			if (filled < rpos + 64)
				ENSURE_INPUT();
			REFILL();
			/* "consumed > 8*in_n + 8" with 56..63 bits buffered is
			 * exactly "rpos > in_n + 8" */
			if (rpos > in_n + 8) {
				result = LDA_BAD_DATA;
				state = ST_DONE;
				continue;
			}
			u32 e = S->lit_tab[(u32)bitbuf & ((1u << ltb) - 1)];
			FLUSH_PENDING();
			u32 cl = e & 15, kind = e & 0xC000, pay = (e >> 4) & 0x3FF;
			if (cl == 0) {
				u32 sym = decode_long(&S->lit, S->lit_sorted, bitbuf, &cl);
				kind = sym < 256 ? K_LIT : sym == 256 ? K_EOB : K_LEN;
				pay = sym < 256 ? sym : sym - 257;
			}
			CONSUME(cl);
			SEG_ADD(pa_dec);
#ifdef LDA_PROFILE
			pa_rounds++;
#endif
			if (kind == K_LIT) {
				if (out_pos == out_avail) {
					result = LDA_INSUFFICIENT_SPACE;
					state = ST_DONE;
					continue;
				}
				/* a second literal in the same round when no end-of-input
				 * rule can interfere (>= 16 input bytes left) and the
				 * 41+ bits still buffered hold its whole codeword */
				u32 e1 = S->lit_tab[(u32)bitbuf & ((1u << ltb) - 1)];
				if ((e1 & 0xC00F) > K_LIT && (e1 & 0xC000) == K_LIT &&
				    rpos + 16 < in_n && out_pos + 1 < out_avail) {
					u16 two = (u16)(pay | (((e1 >> 4) & 0xFF) << 8));
					__builtin_memcpy(outp + out_pos, &two, 2);
					out_pos += 2;
					CONSUME(e1 & 15);
					hist = (hist >> 16) | ((u64)two << 48);
					hist_n = hist_n + 2 > 8 ? 8 : hist_n + 2;
				} else {
					outp[out_pos++] = (u8)pay;
					hist = (hist >> 8) | ((u64)pay << 56);
					hist_n = hist_n + 1 > 8 ? 8 : hist_n + 1;
				}
				continue;
			}
			if (kind == K_EOB) {
				state = final_block ? ST_DONE : ST_HDR;
				continue;
			}
			u32 base, xb;
			len_sym(pay, &base, &xb);
			u32 length = base + ((u32)bitbuf & ((1u << xb) - 1));
			CONSUME(xb);
			if (length > out_avail - out_pos) {
				result = LDA_INSUFFICIENT_SPACE;
				state = ST_DONE;
				continue;
			}
			u32 e2 = S->off_tab[(u32)bitbuf & ((1u << otb) - 1)];
			u32 ol = e2 & 15, osym = e2 >> 4;
			if (ol == 0)
				osym = decode_long(&S->off, S->off_sorted, bitbuf, &ol);
			CONSUME(ol);
			off_sym(osym, &base, &xb);
			u32 dist = base + ((u32)bitbuf & ((1u << xb) - 1));
			CONSUME(xb);
			if (dist > out_pos) {
				result = LDA_BAD_DATA;
				state = ST_DONE;
				continue;
			}
			/* the lane copies from its own earlier output; short
			 * non-overlapping copies are split: loads now, stores at the
			 * next token, so the HBM/L2 round trip overlaps the decode */
			if (dist <= hist_n && length <= 8 && out_pos + 8 <= out_avail) {
				/* source entirely in the register history: expand the
				 * period in registers, one 8-byte store (the bytes past
				 * 'length' are overwritten by the following tokens) */
				u64 pat = hist >> (8 * (8 - dist));
				u32 sh = 8 * dist;
				if (sh < 64)
					pat |= pat << sh;
				sh *= 2;
				if (sh < 64)
					pat |= pat << sh;
				sh *= 2;
				if (sh < 64)
					pat |= pat << sh;
				st8(outp + out_pos, pat);
#ifdef LDA_PROFILE
				pc_reg++;
#endif
				hist = length == 8 ? pat :
				       (hist >> (8 * length)) | (pat << (8 * (8 - length)));
				hist_n = hist_n + length > 8 ? 8 : hist_n + length;
			} else {
				u32 nwords = (length + 7) >> 3;
				u32 maxw = dist >> 3;
				if (nwords <= 4 && nwords <= maxw &&
				    out_pos + 8ull * nwords <= out_avail) {
					const u8 *src = outp + out_pos - dist;
					pend_dst = outp + out_pos;
					pend_n = nwords;
					pend_len = length;
#ifdef LDA_PROFILE
					pc_pipe++;
#endif
					pv0 = ld8(src);
					if (nwords > 1)
						pv1 = ld8(src + 8);
					if (nwords > 2)
						pv2 = ld8(src + 16);
					if (nwords > 3)
						pv3 = ld8(src + 24);
				} else {
#ifdef LDA_PROFILE
					pc_slow++; pc_slowbytes += length;
#endif
					hist_n = 0;
					copy_match(outp, out_pos, out_avail, dist, length);
				}
			}
			out_pos += length;
		}
		PROF_MARK(2);
	}

	FLUSH_PENDING();
#ifdef LDA_PROFILE
	if (lane == 0) {
		atomicAdd(&lda_prof[4], pa_dec);
		atomicAdd(&lda_prof[5], pa_flush);
		atomicAdd(&lda_prof[6], pa_rounds);
		atomicAdd(&lda_prof[7], pa_out);
	}
	{
		atomicAdd(&lda_prof[8], pc_reg);
		atomicAdd(&lda_prof[9], pc_pipe);
		atomicAdd(&lda_prof[10], pc_slow);
		atomicAdd(&lda_prof[11], pc_slowbytes);
	}
#endif
	if (owner) {
		if (result == LDA_SUCCESS) {
			/* epilogue: decompress_template.h:740-771 */
			u64 ain = (CONSUMED() + 7) / 8;
			if (ain > in_n)
				result = LDA_BAD_DATA;
			else
				actual_in[c] = hdr + ain;
		}
		results[c] = result;
		actual_out[c] = result == LDA_SUCCESS ? out_pos : 0;
		if (result != LDA_SUCCESS)
			actual_in[c] = 0;
	}
}
#undef ltb
#undef otb

/* inflate_stream.hip includes this file for its device functions only */
#ifndef LDA_INFLATE_DEVICE_ONLY
extern "C" __global__ void __launch_bounds__(64, 4)
lda_inflate_batch_kernel(u64 n_chunks, int format, u32 lpw,
			 const u8 *__restrict__ in_base,
			 const u64 *__restrict__ in_offsets,
			 const u64 *__restrict__ in_nbytes,
			 u8 *__restrict__ out_base,
			 const u64 *__restrict__ out_offsets,
			 const u64 *__restrict__ out_avail_arr,
			 s32 *__restrict__ results,
			 u64 *__restrict__ actual_in,
			 u64 *__restrict__ actual_out)
{
	lu8 *lds_raw = (lu8 *)(uintptr_t)0;

	inflate_block(blockIdx.x, lds_raw, 0, NULL, n_chunks, format, lpw, in_base,
		      in_offsets, in_nbytes, out_base, out_offsets, out_avail_arr,
		      results, actual_in, actual_out);
}

/*
 * Wave per stream with sub-block parallel token decoding; persistent grid
 * (each wave owns PAR_SCRATCH words of the token scratch and walks the
 * streams first, first + gridDim.x, ...).
 */
#ifndef PAR_WAVES_PER_SIMD
#define PAR_WAVES_PER_SIMD 4	/* 16 streams in flight per CU: occupancy hides the LDS chains */
#endif
/*
 * Costliest first: the streams of a batch in the order of what they cost to
 * decode (the classic longest-processing-time rule: a batch larger than the
 * grid is handed out in this order, so that the last waves do not start a long
 * stream when the others are about to finish).  The cost of a stream
 * follows its compressed size (measured on the benchmark kinds, 64 KiB out:
 * 0.45 + 0.040 ms per KiB in) - except where that is all but the output's
 * size: stored blocks, copied at memory speed.  One workgroup: a 256-bucket
 * counting sort by cost / 512 (everything from 128 KiB up shares the first
 * bucket); the order inside a bucket is whatever the atomics give - it only
 * decides which wave decodes which stream.
 */
static __device__ __forceinline__ u32
order_bucket(const u64 *__restrict__ in_nbytes, const u64 *__restrict__ out_avail, u64 i)
{
	u64 b = in_nbytes[i];
	if (out_avail && b + (b >> 5) >= out_avail[i])
		b >>= 4;	/* (nearly) incompressible: stored */
	b >>= 9;
	return 255 - (b < 255 ? (u32)b : 255u);
}

extern "C" __global__ void __launch_bounds__(1024)
lda_inflate_order_kernel(u64 n, const u64 *__restrict__ in_nbytes,
			 const u64 *__restrict__ out_avail,
			 u32 *__restrict__ order)
{
	__shared__ u32 cnt[256], at[256];
	const u32 tid = threadIdx.x;

	if (tid < 256)
		cnt[tid] = 0;
	__syncthreads();
	for (u64 i = tid; i < n; i += 1024)
		atomicAdd(&cnt[order_bucket(in_nbytes, out_avail, i)], 1u);
	__syncthreads();
	if (tid < 64) {		/* exclusive prefix over the 256 buckets, one wave */
		u32 v[4], s = 0;
#pragma unroll
		for (u32 k = 0; k < 4; k++) {
			v[k] = cnt[4 * tid + k];
			s += v[k];
		}
		u32 base = wave_scan_incl(s) - s;
#pragma unroll
		for (u32 k = 0; k < 4; k++) {
			at[4 * tid + k] = base;
			base += v[k];
		}
	}
	__syncthreads();
	for (u64 i = tid; i < n; i += 1024)
		order[atomicAdd(&at[order_bucket(in_nbytes, out_avail, i)], 1u)] = (u32)i;
}

extern "C" __global__ void __launch_bounds__(64, PAR_WAVES_PER_SIMD)
lda_inflate_wave_kernel(u64 n_chunks, int format, u32 *__restrict__ tokscratch,
			u32 *__restrict__ next_stream,
			const u32 *__restrict__ order,
			const u8 *__restrict__ in_base,
			const u64 *__restrict__ in_offsets,
			const u64 *__restrict__ in_nbytes,
			u8 *__restrict__ out_base,
			const u64 *__restrict__ out_offsets,
			const u64 *__restrict__ out_avail_arr,
			s32 *__restrict__ results,
			u64 *__restrict__ actual_in,
			u64 *__restrict__ actual_out)
{
	lu8 *lds_raw = (lu8 *)(uintptr_t)0;
	u32 *tok = tokscratch + (size_t)blockIdx.x * PAR_SCRATCH;

	/* Which streams share a CU follows from the dispatch order (workgroup i
	 * goes to XCD i mod 8, then CU by CU), so a batch whose content is
	 * periodic in the stream index - every 8th buffer of the same kind, say -
	 * would put all its slow streams on the same CUs.  The streams are taken
	 * in a scrambled order instead: a bijection of the grid that folds the
	 * high index bits into the low ones (an invertible xor-shift on the next
	 * power of two, walked along its cycle until it lands inside the grid). */
	u32 first = blockIdx.x;
	if (gridDim.x > 1) {
		const u32 m = (2u << (31 - __builtin_clz(gridDim.x - 1))) - 1;
		do
			first = (first ^ (first >> 3) ^ (first >> 6) ^ (first >> 9)) & m;
		while (first >= gridDim.x);
	}
	/* the first stream of a wave is fixed by that order; the streams beyond
	 * the grid are handed out as the waves become free (their cost depends
	 * on their content) */
	for (u64 blk = first; blk < n_chunks;) {
		/* (order: see lda_inflate_order_kernel(); NULL = index order) */
		inflate_block(order ? order[blk] : blk, lds_raw, 1, tok, n_chunks, format, 1, in_base,
			      in_offsets, in_nbytes, out_base, out_offsets,
			      out_avail_arr, results, actual_in, actual_out);
		wave_sync();
		u32 nx = 0;
		if (lane_id() == 0)
			nx = atomicAdd(next_stream, 1u);
		blk = (u64)gridDim.x + bcast_first(nx);
	}
}

/* u32 words of token scratch per wave: the lane-interleaved rows of the
 * parse and the list in stream order */
extern "C" size_t lda_inflate_tokcap(void)
{
	return PAR_SCRATCH;
}

extern "C" size_t lda_inflate_window_bytes(void)
{
	return PAR_RW + PAR_STAGE_BYTES + PAR_MAP_BYTES;	/* output mirror, staged input span / copy scratch, token map */
}

/* host helper: LDS bytes per stream */
extern "C" size_t lda_inflate_lds_per_stream(void)
{
	return sizeof(struct stream_lds);
}

extern "C" size_t lda_inflate_lds_shared(void)
{
	return sizeof(struct shared_lds);
}

/*
 * After the batched checksum of the produced bytes: compare with the footer
 * and apply the exact-fill rule.  One thread per stream.
 * lib/gzip_decompress.c:109-131, lib/zlib_decompress.c:74-95,
 * lib/decompress_template.h:765-770.
 */
extern "C" __global__ void
lda_inflate_finalize_kernel(u64 n_chunks, int format, int exact_fill,
			    const u8 *__restrict__ in_base,
			    const u64 *__restrict__ in_offsets,
			    const u64 *__restrict__ out_avail,
			    const u32 *__restrict__ sums,
			    s32 *__restrict__ results,
			    u64 *__restrict__ actual_in,
			    u64 *__restrict__ actual_out)
{
	u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;

	if (c >= n_chunks)
		return;
	s32 r = results[c];
	if (r != LDA_SUCCESS)
		return;
	/* SHORT_OUTPUT is reported by the raw decoder, before the wrappers
	 * look at the footer (gzip_decompress.c:103-109) */
	if (exact_fill && actual_out[c] != out_avail[c]) {
		results[c] = LDA_SHORT_OUTPUT;
		return;
	}
	if (format == LDA_FMT_DEFLATE)
		return;
	const u8 *p = in_base + in_offsets[c] + actual_in[c];
	if (format == LDA_FMT_ZLIB) {
		u32 want = ((u32)p[0] << 24) | ((u32)p[1] << 16) |
			   ((u32)p[2] << 8) | p[3];
		if (want != sums[c]) {
			results[c] = LDA_BAD_DATA;
			return;
		}
		actual_in[c] += 4;
	} else {
		u32 want = p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) |
			   ((u32)p[3] << 24);
		u32 isize = p[4] | ((u32)p[5] << 8) | ((u32)p[6] << 16) |
			    ((u32)p[7] << 24);
		if (want != sums[c] || isize != (u32)actual_out[c]) {
			results[c] = LDA_BAD_DATA;
			return;
		}
		actual_in[c] += 8;
	}
}

LDA_PROF_DEFINE_READER(libdeflate_amd_profile_read_inflate)
#endif /* LDA_INFLATE_DEVICE_ONLY */
