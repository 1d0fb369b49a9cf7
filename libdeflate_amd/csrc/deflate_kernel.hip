/*
 * deflate_kernel.hip - batched DEFLATE / zlib / gzip compression for gfx950.
 *
 * Replaces, for a batch of independent buffers resident in HBM:
 *   libdeflate_deflate_compress      lib/deflate_compress.c:4030-4072
 *   greedy / lazy / lazy2 parsers    lib/deflate_compress.c:2528-2834
 *   hash-chain match finder          lib/hc_matchfinder.h:182-399
 *   Huffman code construction        lib/deflate_compress.c:759-1396
 *   block flush / bit output         lib/deflate_compress.c:1482-2038
 *   gzip / zlib framing              lib/gzip_compress.c:31-82, lib/zlib_compress.c:31-74
 *
 * This is NOT the reference's algorithm transliterated: the reference walks
 * the buffer one position at a time.  Here ONE 1024-thread workgroup (16
 * waves) owns a buffer and keeps the whole LZ77 window state in the CU's
 * 160 KiB LDS (struct deflate_lds below):
 *
 *   in[]    32 KiB ring of input bytes           (coalesced 16 B loads)
 *   prev[]  32 Ki x u16 ring: previous position with the same 4-byte hash
 *   head[]  most recent position per hash bucket; head3[] for 3-byte matches
 *   M[]     per-position best (length, distance) of the current tile
 *
 * and advances in TILES of 4096 positions:
 *
 *   S1  every lane hashes its positions; each wave bitonic-sorts 64
 *       (hash, position) keys with DPP exchanges so equal hashes become
 *       neighbours: that yields the in-order "previous occurrence" links
 *       inside the group without any serial insertion;
 *   S2  wave 0 threads the groups through head[] in position order (first /
 *       last of each hash run only), wave 1 does the same for head3[]; the
 *       other waves already search the part of the tile published so far;
 *   S3  ALL positions of the tile search their chain in parallel (depth and
 *       nice length per level as lib/deflate_compress.c:3927-3979), in
 *       alternating "walk" passes (8 chain steps, filtered on the byte at the
 *       best length so far) and "evaluate" passes (8-byte-at-a-time
 *       extension of the queued hits); waves claim work dynamically;
 *   S4  the greedy / lazy / lazy2 choice is a pure function of the
 *       per-position results (rules of deflate_compress.c:2573-2575,
 *       2712-2755): "next token start" is a forest over the positions, so
 *       the parse is pointer doubling per wave segment plus a short chain
 *       across segments; matches are appended to a per-workgroup list in HBM
 *       and the symbol histogram is built in the same pass;
 *   S5  at block end (content-driven split, deflate_compress.c:2143-2256, or
 *       64 Ki positions): length-limited canonical Huffman codes, exact cost
 *       of dynamic / static / stored, header;
 *   S6  tokens are encoded position-parallel in windows of 4096 positions:
 *       every lane looks up its codeword(s), a workgroup prefix sum of the
 *       bit lengths gives the bit offset, ds_or packs the bits into an LDS
 *       staging buffer that is written to HBM with coalesced stores.
 *
 * HBM traffic beyond input-once / output-once: the match list (8 B per match,
 * written in S4, read in S6) and literals of a block that have left the LDS
 * ring by the time the block is flushed.  The compressed bytes differ from
 * the reference's (libdeflate.h:76-83 leaves them unpinned); validity, round
 * trip, compress_bound and ratio-vs-reference are what the tests check.
 */
#include <stddef.h>
#include "device_common.h"
#include "kernels.h"

#define NT LDA_DEFLATE_THREADS
#define NWAVES (NT / 64)
#define TILE 4096
#define RING 32768u
#define RMASK (RING - 1)
#define LOOKAHEAD 272u
#ifndef HASH_BITS
#define HASH_BITS 13
#endif
#define HASH3_BITS 12
/*
 * The matches of the current block live in HBM (8 bytes each, one list per
 * workgroup): nothing of a block has to stay in LDS until the block is
 * written, so a block can be as long as a whole 64 KiB buffer, like the
 * reference's for homogeneous data.  There is no adaptive block splitting
 * (lib/deflate_compress.c:2092-2218) yet: blocks end at tile boundaries once
 * they reach MAX_BLOCK_LEN, which bounds the cost of content that changes
 * inside a buffer.
 */
#define MAX_BLOCK_LEN 65536u
#define SEQ_TILE_MAX (TILE / 3 + 40)	/* > new matches per tile (min match 3) */
#define SEQ_GCAP (MAX_BLOCK_LEN / 3 + 2 * TILE)
#define SEQ_STRIDE (SEQ_GCAP + (TILE + 8) / 2 + 320 + 256)	/* u64 words of HBM scratch per workgroup */
#define EWIN TILE		/* encode window (positions) */
#ifndef S3_WALK
#define S3_WALK 8		/* chain steps per walk pass (a lane stalls while its 4-entry hit queue is full) */
#endif
#ifndef S3_EVMIN
#define S3_EVMIN 1u		/* lanes with a queued hit that trigger an evaluate round */
#endif
#ifndef S3_TAIL
#define S3_TAIL 256u		/* positions at the end of a tile searched with reduced depth */
#endif
#ifndef S3_CLAIM
#define S3_CLAIM 24u		/* finished lanes that trigger a claim pass */
#endif
#define STG_WORDS ((TILE + 8) / 2 - 8)	/* staging: STG_WORDS + 8 words = sizeof nxtA */

#define M_FIRST 0x10000u
#define M_LAST 0x20000u
#define M_VALID 0x40000u

struct deflate_lds {
	u8 in[RING + 32];
	u16 prev[RING];
	u16 head[1u << HASH_BITS];
	u16 head3[1u << HASH3_BITS];	/* last position per 3-byte hash (no chain) */
	u32 M[TILE + 8];	/* tile scratch; encode: KD[EWIN] + staging */
	u8 mark[TILE + 8];	/* 1 = literal chosen at this position */
	u32 freq[320];		/* litlen 0..287, offset 288..319 */
	union {
		struct {	/* live only while a block is being finished */
			u8 lens[320];
			u16 codes[320];	/* bit-reversed codewords */
			u16 sorted[288];
			u32 hw[288];	/* Huffman build scratch */
			u16 pre_items[320 + 8];	/* precode symbol | extra << 5 */
			u32 pre_freq[19];
			u8 pre_lens[20];
			u16 pre_codes[20];
		};
		u16 nxtB[TILE + 8];	/* live only during the token choice */
	};
	u16 nxtA[TILE + 8] __attribute__((aligned(16)));	/* S6: bit staging */
	u32 scan[2][NWAVES + 1];
	u32 carry[6];		/* staging bytes kept between blocks */
	u32 obs[1][10];		/* block-split observations of the block before this tile */
	u32 vars[20];
};

/* LDS-resident: every pointer into the block carries the address space, and
 * the block itself sits at LDS address 0 (no static LDS in this kernel), so
 * member offsets become instruction immediates */
#ifdef __HIP_DEVICE_COMPILE__
#define AS3 __attribute__((address_space(3)))
#else
#define AS3	/* the host pass only parses the device code */
#endif
typedef AS3 struct deflate_lds lds_t;

static_assert(sizeof(struct deflate_lds) <= 163840, "LDS of one CU");
static_assert(offsetof(struct deflate_lds, in) == 0, "ld32/ld64 assume in[] at LDS offset 0");
#define PREV_OFF ((u32)offsetof(struct deflate_lds, prev))

enum {
	V_NSEQ = 0, V_ENTRY, V_WALKPOS_LO, V_SPILL, V_NPRE, V_TMP0, V_TMP1,
	V_TMP2, V_TMP3, V_CTR, V_MINLEN, V_SPILL1, V_SEQCNT, V_SEQCNT1, V_READY,
	V_SPLIT, V_NSEQ_PRE, V_WPOS_PRE, V_FIT
};

struct level_params {
	u32 depth;
	u32 nice;
	u32 mode;	/* 0 greedy, 1 lazy, 2 lazy2 */
};

/* ---------------- small helpers ---------------- */

/*
 * Unaligned reads from the input ring as ALIGNED dword reads + a byte funnel
 * shift (v_alignbyte): misaligned ds_read_b32/b64 are legal on gfx950 but
 * measured ~10x slower than aligned ones.  The ring is mirrored for 32 bytes
 * past its end so idx+1/idx+2 never wrap.
 */
/*
 * The kernel has no static LDS, so the dynamic LDS block (struct deflate_lds)
 * starts at LDS address 0 (checked once per workgroup): reads of the hot
 * arrays go through address-space-3 pointers built from constant offsets, so
 * the array base folds into the instruction's immediate offset instead of
 * costing a VALU add of a link-time symbol per access.
 */
#define LDS32(byteoff) (*(const __attribute__((address_space(3))) u32 *)(uintptr_t)(byteoff))
#define LDS16(byteoff) (*(const __attribute__((address_space(3))) u16 *)(uintptr_t)(byteoff))

static __device__ __forceinline__ u32 ld32(const AS3 u8 *ring, u32 pos)
{
	(void)ring;	/* in[] is the first member: byte offset 0 */
	u32 o = pos & RMASK, i = o & ~3u;
	return __builtin_amdgcn_alignbyte(LDS32(i + 4), LDS32(i), o & 3);
}

static __device__ __forceinline__ u64 ld64(const AS3 u8 *ring, u32 pos)
{
	(void)ring;
	u32 o = pos & RMASK, i = o & ~3u;
	u32 a = LDS32(i), b = LDS32(i + 4), c = LDS32(i + 8);
	u32 lo = __builtin_amdgcn_alignbyte(b, a, o & 3);
	u32 hi = __builtin_amdgcn_alignbyte(c, b, o & 3);
	return ((u64)hi << 32) | lo;
}

static __device__ __forceinline__ u32 hash3(u32 w)
{
	/* 3-byte hash for length-3 matches, lib/hc_matchfinder.h:228-231 role */
	return ((w & 0xFFFFFFu) * 0x1E35A7BDu) >> (32 - HASH3_BITS);
}

static __device__ __forceinline__ u32 hash4(u32 w)
{
	/* multiplicative hash, lib/matchfinder_common.h:168-172 */
	return (w * 0x1E35A7BDu) >> (32 - HASH_BITS);
}

/*
 * Length-3 match for a position whose chain search found nothing >= 4:
 * the short distances 1..8 are compared in registers (covers strided binary
 * records), then the single-slot 3-byte hash candidate.  Distance limits for
 * length 3 as lib/deflate_compress.c:2573-2575 / :2666-2668.
 */
struct deflate_lds;
static __device__ u32
find_len3(const lds_t *L, u32 p, u32 cur, u32 c3_16, u32 dmax,
	  u32 dlim, u32 *best);

/* workgroup exclusive scan of one value per thread; returns the exclusive
 * prefix and writes the total to *total.  Two barriers. */
static __device__ u32 block_scan(lds_t *L, u32 v, u32 *total)
{
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	u32 incl = wave_scan_incl(v);

	if (lane == 63)
		L->scan[0][wave] = incl;
	__syncthreads();
	u32 base = 0, tot = 0;
#pragma unroll
	for (u32 w = 0; w < NWAVES; w++) {
		u32 s = L->scan[0][w];
		if (w < wave)
			base += s;
		tot += s;
	}
	__syncthreads();
	*total = tot;
	return base + incl - v;
}

/* the same with ONE barrier: the partial sums alternate between two arrays
 * (*tog flips per call, uniformly), so a fast thread's next call cannot
 * overwrite what a slow thread still reads.  Every second call reuses an
 * array, and the barrier of the call in between orders that. */
static __device__ u32 block_scan1(lds_t *L, u32 v, u32 *total,
				  u32 *tog)
{
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	u32 incl = wave_scan_incl(v);
	u32 *sc = L->scan[*tog];

	*tog ^= 1;
	if (lane == 63)
		sc[wave] = incl;
	__syncthreads();
	u32 base = 0, tot = 0;
#pragma unroll
	for (u32 w = 0; w < NWAVES; w++) {
		u32 s = sc[w];
		if (w < wave)
			base += s;
		tot += s;
	}
	*total = tot;
	return base + incl - v;
}

/* length slot / extra bits (lib/deflate_compress.c:237-308 tables, computed) */
static __device__ __forceinline__ void
length_code(u32 len, u32 *slot, u32 *xbits, u32 *xval)
{
	u32 l = len - 3;
	if (l < 8) {
		*slot = l; *xbits = 0; *xval = 0;
	} else if (len == 258) {
		*slot = 28; *xbits = 0; *xval = 0;
	} else {
		u32 hb = 31 - __builtin_clz(l);
		*xbits = hb - 2;
		*slot = 4 * (hb - 1) + ((l >> (hb - 2)) & 3);
		*xval = l & ((1u << (hb - 2)) - 1);
	}
}

static __device__ __forceinline__ void
dist_code(u32 dist, u32 *slot, u32 *xbits, u32 *xval)
{
	u32 d = dist - 1;
	if (d < 4) {
		*slot = d; *xbits = 0; *xval = 0;
	} else {
		u32 hb = 31 - __builtin_clz(d);
		*xbits = hb - 1;
		*slot = 2 * hb + ((d >> (hb - 1)) & 1);
		*xval = d & ((1u << (hb - 1)) - 1);
	}
}

/*
 * Minimum useful match length from the number of distinct literals in use:
 * with few distinct literals a literal is so cheap that short matches lose.
 * Policy of lib/deflate_compress.c:2295-2327 (table restated as thresholds).
 */
static __device__ u32 choose_min_len(u32 used_literals, u32 depth)
{
	u32 m = used_literals >= 80 ? 3 : used_literals >= 45 ? 4 :
		used_literals >= 16 ? 5 : used_literals >= 10 ? 6 :
		used_literals >= 8 ? 7 : used_literals >= 6 ? 8 : 9;
	if (depth < 16) {
		u32 cap = depth < 5 ? 4 : depth < 10 ? 5 : 7;
		if (m > cap)
			m = cap;
	}
	return m;
}

static __device__ u32
find_len3(const lds_t *L, u32 p, u32 cur, u32 c3_16, u32 dmax,
	  u32 dlim, u32 *best)
{
	if (dmax > dlim)
		dmax = dlim;
	u32 want = cur & 0xFFFFFFu;
	if (p >= 8) {
		u64 w8 = ld64(L->in, p - 8);
#pragma unroll
		for (u32 d = 1; d <= 8; d++) {
			u64 t = w8 >> (8 * (8 - d));
			if (d < 3)
				t |= (u64)cur << (8 * d);
			if (d <= dmax && ((u32)t & 0xFFFFFFu) == want) {
				*best = 3;
				return d;
			}
		}
	}
	u32 d = (p - c3_16) & 0xFFFF;
	if (d && d <= dmax && ((ld32(L->in, p - d) ^ cur) & 0xFFFFFFu) == 0) {
		*best = 3;
		return d;
	}
	return 0;
}

/*
 * How far the parse advances from a position whose best matches (len |
 * dist << 16; 0 = none) at p, p+1, p+2 are m0, m1, m2:
 *   1 = literal, 2 = two literals (lazy2 deferral), else the match length.
 * Greedy / lazy / lazy2 rules: lib/deflate_compress.c:2573-2575, 2681,
 * 2712-2725, 2742-2755.
 */
static __device__ __forceinline__ u32
token_step(u32 m0, u32 m1, u32 m2, u32 mode, u32 nice)
{
	u32 l0 = m0 & 0xFFFF;

	if (l0 == 0)
		return 1;
	if (mode >= 1 && l0 < nice) {
		s32 b0 = 31 - __builtin_clz(m0 >> 16);
		u32 l1 = m1 & 0xFFFF;
		if (l1 >= l0 &&
		    4 * (s32)(l1 - l0) + (b0 - (s32)(31 - __builtin_clz(m1 >> 16))) > 2)
			return 1;
		if (mode >= 2) {
			u32 l2 = m2 & 0xFFFF;
			if (l2 >= l0 &&
			    4 * (s32)(l2 - l0) +
			    (b0 - (s32)(31 - __builtin_clz(m2 >> 16))) > 6)
				return 2;
		}
	}
	return l0;
}

/* ---------------- min-cost parse (levels 10-12) ---------------- */

/*
 * The reference's levels 10-12 (lib/deflate_compress.c:3327-3849) collect all
 * matches per position with a binary-tree finder, run a backward min-cost DP
 * over a whole block and re-cost it several times.  Restated for the tile
 * pipeline:
 *   - the candidates of a position are all the lengths 3..L of its best
 *     (longest, then nearest) match from the chain search;
 *   - symbol prices are -log2 of the frequencies of the block so far, in
 *     1/16 bit.  The first tile of a block has no history: it is parsed
 *     lazily into the histogram first (a dry run that is rolled back), and if
 *     pure literals priced by the tile's own byte statistics would be
 *     cheaper than that parse, literal prices come from the byte statistics
 *     and match prices from flat defaults (the role of the reference's
 *     default-cost tables, :2986-3102);
 *   - the DP runs backwards, per wave over 256 positions plus 64 positions
 *     of warm-up beyond them (a min-cost parse forgets where it started
 *     within a few tokens, like a Huffman parse re-synchronises), with the
 *     cost-to-go of the positions ahead held in registers: no memory traffic
 *     inside the recurrence (opt_parse_wave());
 *   - the chosen lengths replace the match lengths in M[], and the ordinary
 *     token walk (S4, greedy rule) follows them.
 */
#define OPT_SEG 256
#define OPT_WARM 64
#define OPT_BIG 0x40000000u
#ifndef OPT_FIT_NUM
#define OPT_FIT_NUM 7u	/* the block's literal statistics "fit" a tile up to 7/4 of */
#define OPT_FIT_DEN 4u	/* the tile's own literal-only estimate */
#endif
/* price tables (u16, 1/16 bit) and the byte histogram live in the block-end
 * scratch, which is dead until S4 uses nxtB */
#define OPT_LIT(L) ((AS3 u16 *)(L)->sorted)	/* [256] by literal */
#define OPT_LEN(L) ((AS3 u16 *)(L)->codes)	/* [259] by length, extra bits included */
#define OPT_OFF(L) ((AS3 u16 *)(L)->pre_items)	/* [30] by offset slot, extra bits included */
#define OPT_HIST(L) ((AS3 u32 *)(L)->hw)	/* [256] bytes of the tile */

static __device__ __forceinline__ u32 opt_price(u32 f, float lg_total, float maxbits)
{
	float b = lg_total - __log2f((float)f + 0.4f);
	b = fminf(fmaxf(b, 1.0f), maxbits);
	return (u32)(b * 16.0f + 0.5f);
}

/* Prices from freq[]; with try_flat (freq[] = lazy parse of this tile alone)
 * the literal-only estimate decides between them and the flat start.
 * Without try_flat (freq[] = the block so far) the return value tells whether
 * the block's literal statistics fit this tile's bytes at all: 0 = they do;
 * 1 = poorly: the tile is better parsed by the lazy rule than with prices
 * that describe other data; 2 = not at all: the content has changed, the
 * block should end here whatever the observation classes of the split
 * heuristic say (they cannot tell a 16-letter alphabet from text once both
 * are mostly literals).
 * Whole workgroup; ends with a barrier. */
static __device__ u32
opt_build_costs(lds_t *L, u32 tid, bool try_flat, bool check_fit, u32 t, u32 tn, u32 *bsave)
{
	u32 *osave = bsave + 256;
	AS3 u16 *lit = OPT_LIT(L), *len = OPT_LEN(L), *off = OPT_OFF(L);
	AS3 u32 *o0 = OPT_HIST(L);
	u32 tl, to;
	(void)block_scan(L, tid < 286 ? L->freq[tid] : 0, &tl);
	(void)block_scan(L, tid >= 288 && tid < 318 ? L->freq[tid] : 0, &to);
	const float lgl = __log2f((float)tl + 1.0f), lgo = __log2f((float)to + 1.0f);
	u32 est = 0, lsl = 0, lxb = 0, lxv = 0;
	if (tid < 256) {
		u32 f = L->freq[tid], c = opt_price(f, lgl, 14.0f);
		lit[tid] = (u16)c;
		est = f * c;
	} else if (tid < 512) {
		u32 l = tid - 253;	/* 3..258 */
		length_code(l, &lsl, &lxb, &lxv);
		u32 f = L->freq[257 + lsl], c = opt_price(f, lgl, 14.0f) + 16 * lxb;
		len[l] = (u16)c;
		if (lxv == 0)
			est = f * c;
	} else if (tid < 542) {
		u32 sl = tid - 512, xb = sl < 4 ? 0 : (sl >> 1) - 1;
		u32 f = L->freq[288 + sl], c = opt_price(f, lgo, 12.0f) + 16 * xb;
		off[sl] = (u16)c;
		est = f * c;
	}
	if (try_flat) {
		u32 el, e0;
		(void)block_scan(L, est, &el);
		for (u32 i = tid; i < 256; i += NT)
			o0[i] = 0;
		__syncthreads();
		for (u32 i = tid; i < tn; i += NT)
			atomicAdd((u32 *)&o0[L->in[(t + i) & RMASK]], 1u);
		__syncthreads();
		u32 cf = 0, e = 0;
		if (tid < 256) {
			u32 f = o0[tid];
			cf = opt_price(f, __log2f((float)tn + 1.0f), 14.0f);
			e = f * cf;
			bsave[tid] = 0;	/* the block's bytes start with this tile */
			osave[tid] = f;
		}
		(void)block_scan(L, e, &e0);
		if (e0 < el) {
			if (tid < 256)
				lit[tid] = (u16)cf;
			else if (tid < 512)
				len[tid - 253] = (u16)(16 * (7 + lxb));
			else if (tid < 542) {
				u32 sl = tid - 512;
				off[sl] = (u16)(16 * (5 + (sl < 4 ? 0 : (sl >> 1) - 1)));
			}
		}
		__syncthreads();
		return 0;
	}
	if (!check_fit) {	/* prices only (second pass over a first tile) */
		__syncthreads();
		return 0;
	}
	/* fit: the tile's bytes priced as literals of this block vs by their own
	 * statistics (both without the share of the matches) */
	u32 tlit, e_blk, e_own;
	(void)block_scan(L, tid < 256 ? L->freq[tid] : 0, &tlit);
	for (u32 i = tid; i < 256; i += NT)
		o0[i] = 0;
	__syncthreads();
	for (u32 i = tid; i < tn; i += NT)
		atomicAdd((u32 *)&o0[L->in[(t + i) & RMASK]], 1u);
	__syncthreads();
	u32 eb = 0, eo = 0, bb = 0, tb, tv2;
	if (tid < 256) {
		u32 f = o0[tid];
		eb = f * opt_price(L->freq[tid], __log2f((float)tlit + 1.0f), 14.0f);
		eo = f * opt_price(f, __log2f((float)tn + 1.0f), 14.0f);
		/* bytes of the block so far (the previous tile joins them now) */
		bb = bsave[tid] + osave[tid];
		bsave[tid] = bb;
		osave[tid] = f;
	}
	(void)block_scan(L, eb, &e_blk);
	(void)block_scan(L, eo, &e_own);
	/* total variation between the byte distributions of this tile and of
	 * the block: homogeneous data stays below 0.5 (drifting binary counters
	 * reach it), a change of content is 0.7 and up */
	(void)block_scan(L, bb, &tb);
	u32 dv = 0;
	if (tid < 256 && tb) {
		float d = (float)o0[tid] / (float)tn - (float)bb / (float)tb;
		dv = (u32)(fabsf(d) * 65536.0f);
	}
	(void)block_scan(L, dv, &tv2);	/* 2 TV in 1/65536 */
	__syncthreads();
	if (tv2 > (u32)(2 * 0.6f * 65536.0f))
		return 2;
#ifdef LDA_DEBUG_SPLIT
	if (tid == 0)
		L->vars[V_TMP3] = 100 * e_blk / (e_own ? e_own : 1);
#endif
	return OPT_FIT_DEN * e_blk <= OPT_FIT_NUM * e_own ? 0 : 2 * e_blk <= 5 * e_own ? 1 : 2;
}

/* minimum over the wave, wave-uniform */
static __device__ __forceinline__ u32 wave_min_u32(u32 v)
{
	u32 o;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x111, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x112, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x114, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x118, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x142, 0xA, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x143, 0xC, 0xF, false);
	v = o < v ? o : v;
	return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

/* minimum over lanes 0..15, wave-uniform */
static __device__ __forceinline__ u32 row0_min_u32(u32 v)
{
	u32 o;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x111, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x112, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x114, 0xF, 0xF, false);
	v = o < v ? o : v;
	o = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, v, 0x118, 0xF, 0xF, false);
	v = o < v ? o : v;
	return (u32)__builtin_amdgcn_readlane((int)v, 15);
}

/*
 * One wave's part of the min-cost parse: chosen length (1 = literal) for the
 * tile-relative positions [lo, hi) into ch16[position + 4]; positions up to
 * 'e' are parsed as warm-up.  Backwards, one position p per step.  The stage
 * is bound by VALU issue (16 waves x 4 cycles per instruction), so the step
 * is built to need few vector instructions:
 *   - c(p) = min(c(p+1) + literal, best match candidate) runs on the scalar
 *     unit; the costs are packed as cost << 9 so that adding the packed
 *     length price (price << 9 | length) and taking the minimum yields the
 *     cost and the length together;
 *   - the costs of the 64 positions p+3.. sit in w0 (lane j = position
 *     p + 3 + j): the candidates of a match of up to 66 bytes are one add,
 *     one select and a DPP reduction (over one row of 16 lanes when the
 *     match is no longer than 18); w0 slides by one lane per step, and
 *     c(p+2) enters at lane 0;
 *   - longer matches are rare: the costs further ahead are kept as
 *     snapshots of w0 taken every 64 steps (ws1..ws4; at step s of a group
 *     lane j of ws_k is length s + 3 + j + 64 (k - 1)) and looked at only
 *     then, with the length prices read from LDS.
 */
static __device__ void
opt_parse_wave(lds_t *L, AS3 u16 *ch16, u32 t, s32 lo_, s32 hi_, s32 e_, u32 lane)
{
	AS3 u16 *lit = OPT_LIT(L), *len = OPT_LEN(L), *off = OPT_OFF(L);
	/* wave-uniform by construction; tell the compiler, or the step loop
	 * is compiled as a divergent one */
	const s32 lo = __builtin_amdgcn_readfirstlane(lo_);
	const s32 hi = __builtin_amdgcn_readfirstlane(hi_);
	const s32 e = __builtin_amdgcn_readfirstlane(e_);
	const u32 lcp0 = ((u32)len[3 + lane] << 9) | (3 + lane);
	const bool lane0 = lane == 0;
	u32 w0 = 0, ws1 = 0, ws2 = 0, ws3 = 0, ws4 = 0;
	const u32 nsteps = (u32)(e - lo);
	u32 ch = 0, c1 = 0, c2 = 0;	/* c1 = c(p+1), c2 = c(p+2) */
	/* groups of 64 steps: position data in, snapshots rotated, choices out */
	for (u32 g0 = 0; g0 < nsteps; g0 += 64) {
		const u32 cnt = (u32)__builtin_amdgcn_readfirstlane(
			(int)(nsteps - g0 < 64 ? nsteps - g0 : 64));
		const s32 ptop = e - 1 - (s32)g0;	/* lane j = position ptop - j */
		u32 pk = 0;
		{
			s32 pj = ptop - (s32)lane;
			if (pj >= lo) {
				u32 m = L->M[pj + 4], lm = m & 0xFFFF, oc = 0;
				if (lm >= 3) {
					u32 ds, xb, xv;
					dist_code(m >> 16, &ds, &xb, &xv);
					oc = off[ds];
				} else {
					lm = 0;
				}
				u32 lc = lit[L->in[(t + (u32)pj) & RMASK]];
				pk = lm | (oc << 9) | (lc << 18);
			}
		}
		ws4 = ws3;
		ws3 = ws2;
		ws2 = ws1;
		ws1 = w0;
		for (u32 sl = 0; sl < cnt; sl++) {
			const u32 q = (u32)__builtin_amdgcn_readlane((int)pk, sl);
			const u32 lm = q & 511, oc = (q >> 9) & 511, lc = q >> 18;
			u32 best = c1 + (lc << 9) + 1;
			if (lm >= 3) {
				u32 cand = lane + 3 <= lm ? w0 + lcp0 : OPT_BIG;
				u32 mn;
				if (lm <= 18) {
					mn = row0_min_u32(cand);
				} else {
					if (lm > 66) {
						/* the asm statement keeps this a branch instead
						 * of predicated instructions on every step */
						u32 ln = lane;	/* opaque: no address induction
								 * variable in the common path */
						asm volatile("; long match" : "+v"(ln));
						const u32 l1 = sl + 3 + ln;
						u32 x1 = l1 <= lm ? ws1 + (((u32)len[l1] << 9) | l1) : OPT_BIG;
						u32 x2 = l1 + 64 <= lm ? ws2 + (((u32)len[l1 + 64] << 9) | (l1 + 64)) : OPT_BIG;
						u32 x3 = l1 + 128 <= lm ? ws3 + (((u32)len[l1 + 128] << 9) | (l1 + 128)) : OPT_BIG;
						u32 l4 = l1 + 192 <= 258 ? l1 + 192 : 258;
						u32 x4 = l1 + 192 <= lm ? ws4 + (((u32)len[l4] << 9) | l4) : OPT_BIG;
						x1 = x2 < x1 ? x2 : x1;
						x3 = x4 < x3 ? x4 : x3;
						cand = x1 < cand ? x1 : cand;
						cand = x3 < cand ? x3 : cand;
					}
					mn = wave_min_u32(cand);
				}
				mn += oc << 9;
				best = mn < best ? mn : best;
			}
			{	/* ch[lane sl] = chosen length (lane select through m0: one
				 * SGPR operand per VALU instruction); the s_nop covers the
				 * lane-select hazard the compiler cannot see in the asm */
				const u32 cl = best & 511;
				asm("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0"
				    : "+v"(ch) : "s"(cl), "s"(sl) : "m0");
			}
			/* slide: every cost moves one lane up, c(p+2) enters at lane 0
			 * (wave_ror:1; every lane has a source, 'old' is unused) */
			const u32 r0 = __builtin_amdgcn_update_dpp(w0, w0, 0x13C, 0xF, 0xF, false);
			w0 = lane0 ? c2 : r0;
			c2 = c1;
			c1 = best & ~511u;
		}
		{
			s32 pj = ptop - (s32)lane;
			if (lane < cnt && pj < hi)	/* pj >= lo: lane < cnt */
				ch16[pj + 4] = (u16)ch;
		}
	}
}

/* ---------------- Huffman code construction (wave 0) ---------------- */

/*
 * Length-limited canonical code for freq[0..n) -> lens[], codes[] (codewords
 * bit-reversed, ready for LSB-first output).  Called by ONE wave.
 *   - rank sort by (freq, sym): by the whole workgroup beforehand
 *     (presorted) or by this wave;
 *   - optimal tree by the in-place two-queue method on lane 0;
 *   - depth clamp with Kraft repair (what lib/deflate_compress.c:1022-1091
 *     achieves with its length-count shuffle);
 *   - fewer than two used symbols -> two 1-bit codewords
 *     (lib/deflate_compress.c:1369-1378).
 */
template <int N> struct huff_scratch {
	u32 A[N];	/* leaf weights, later hop pointers */
	u32 NW[N];	/* internal node weights, later depths */
	u32 P[N];	/* parent of each internal node */
	u32 cntI[40];	/* internal nodes per depth */
	u32 cnt[40];	/* leaves per depth (code lengths) */
	u32 start[16];	/* first index in sorted[] for each length */
	u32 nc[16];	/* next canonical codeword per length */
};

template <int N> static __device__ void
make_code(const u32 *freq, u32 n, u32 maxlen, u8 *lens, u16 *codes,
	  u16 *sorted, huff_scratch<N> *H, u32 used, bool presorted, u32 lane)
{
	for (u32 s = lane; s < n; s += 64)
		lens[s] = 0;
	if (!presorted) {
		/* rank sort of the used symbols by (freq, sym), one wave */
		used = 0;
		for (u32 s0 = 0; s0 < n; s0 += 64) {
			u32 s = s0 + lane;
			u32 f = s < n ? freq[s] : 0;
			used += __builtin_popcountll(__ballot(f != 0));
		}
		for (u32 s = lane; s < n; s += 64) {
			u32 f = freq[s];
			if (!f)
				continue;
			u32 key = (f << 9) | s, rank = 0;	/* freq < 2^22 */
			for (u32 t = 0; t < n; t++) {
				u32 ft = freq[t];
				rank += (ft != 0) & (((ft << 9) | t) < key);
			}
			sorted[rank] = (u16)s;
		}
	}
	wave_sync();
	const u32 m = used;
	if (m < 2) {
		if (lane == 0) {
			u32 s = m ? sorted[0] : 0;
			u32 other = s ? 0 : 1;
			lens[s] = 1;
			lens[other] = 1;
			for (u32 d = 0; d < 16; d++)
				H->cnt[d] = 0;
			H->cnt[1] = 2;
		}
		wave_sync();
	} else {
		for (u32 i = lane; i < m; i += 64)
			H->A[i] = freq[sorted[i]];
		if (lane < 40)
			H->cntI[lane] = 0;
		wave_sync();
		if (lane == 0) {
			/* two-queue merge: leaves A[] (ascending), nodes NW[] in
			 * creation order (ascending too); heads cached in registers
			 * (a variant that also prefetched the following entries had
			 * more instructions on this single-lane path and was slower) */
			u32 leaf = 0, node = 0;
			u32 wl = H->A[0], wn = 0xFFFFFFFFu;
			for (u32 k = 0; k + 1 < m; k++) {
				u32 w;
				if (leaf < m && wl <= wn) {
					w = wl;
					leaf++;
					wl = leaf < m ? H->A[leaf] : 0xFFFFFFFFu;
				} else {
					w = wn;
					H->P[node] = k;
					node++;
					wn = node < k ? H->NW[node] : 0xFFFFFFFFu;
				}
				if (leaf < m && wl <= wn) {
					w += wl;
					leaf++;
					wl = leaf < m ? H->A[leaf] : 0xFFFFFFFFu;
				} else {
					w += wn;
					H->P[node] = k;
					node++;
					wn = node < k ? H->NW[node] : 0xFFFFFFFFu;
				}
				H->NW[k] = w;
				if (node == k)
					wn = w;	/* the new node is the only one queued */
			}
		}
		wave_sync();
		/* depth of every internal node by pointer jumping (root = m-2) */
		{
			const u32 root = m - 2;
			enum { NJ = (N + 63) / 64 };	/* internal nodes per lane */
			u32 dd[NJ], hh[NJ];
#pragma unroll
			for (u32 j = 0; j < NJ; j++) {
				u32 k = lane + 64 * j;
				dd[j] = (k < root) ? 1 : 0;
				hh[j] = (k < root) ? H->P[k] : root;
			}
			wave_sync();
#pragma unroll
			for (u32 j = 0; j < NJ; j++) {
				u32 k = lane + 64 * j;
				if (k <= root) {
					H->NW[k] = dd[j];
					H->A[k] = hh[j];
				}
			}
			wave_sync();
			for (u32 r = 0; r < 6; r++) {	/* depth < 64 */
#pragma unroll
				for (u32 j = 0; j < NJ; j++) {
					u32 k = lane + 64 * j;
					if (k <= root) {
						u32 h = H->A[k];
						dd[j] = H->NW[k] + H->NW[h];
						hh[j] = H->A[h];
					}
				}
				wave_sync();
#pragma unroll
				for (u32 j = 0; j < NJ; j++) {
					u32 k = lane + 64 * j;
					if (k <= root) {
						H->NW[k] = dd[j];
						H->A[k] = hh[j];
					}
				}
				wave_sync();
			}
#pragma unroll
			for (u32 j = 0; j < NJ; j++) {
				u32 k = lane + 64 * j;
				if (k <= root)
					atomicAdd((u32 *)&H->cntI[dd[j] < 39 ? dd[j] : 39], 1u);
			}
			wave_sync();
			/* leaves at depth d = 2 * internal(d-1) - internal(d) */
			if (lane < 40)
				H->cnt[lane] = lane ? 2 * H->cntI[lane - 1] - H->cntI[lane] : 0;
			wave_sync();
		}
		if (lane == 0) {
			/* clamp to maxlen, repair Kraft sum (zlib-style) */
			u32 over = 0;
			for (u32 d = maxlen + 1; d < 40; d++) {
				over += H->cnt[d];
				H->cnt[maxlen] += H->cnt[d];
				H->cnt[d] = 0;
			}
			if (over) {
				u32 kraft = 0;
				for (u32 d = 1; d <= maxlen; d++)
					kraft += H->cnt[d] << (maxlen - d);
				while (kraft > (1u << maxlen)) {
					u32 d = maxlen - 1;
					while (H->cnt[d] == 0)
						d--;
					H->cnt[d]--;
					H->cnt[d + 1] += 2;
					H->cnt[maxlen]--;
					kraft -= 1;
				}
			}
			/* rarest symbols get the longest codewords */
			u32 at = 0;
			for (u32 d = maxlen; d >= 1; d--) {
				H->start[d] = at;
				at += H->cnt[d];
			}
		}
		wave_sync();
		for (u32 i = lane; i < m; i += 64) {
			u32 d = 1;
			for (u32 q = 2; q <= maxlen; q++)
				if (H->cnt[q] && i >= H->start[q] &&
				    i < H->start[q] + H->cnt[q])
					d = q;
			lens[sorted[i]] = (u8)d;
		}
		wave_sync();
	}
	/* canonical codewords, bit-reversed: codes of one length go to the
	 * symbols in increasing symbol order -> ballot ranks */
	if (lane == 0) {
		u32 code = 0;
		H->nc[0] = 0;
		for (u32 d = 1; d < 16; d++) {
			code = (code + (d > 1 ? H->cnt[d - 1] : 0)) << 1;
			H->nc[d] = code;
		}
	}
	wave_sync();
	{
		u32 run[16];
#pragma unroll
		for (u32 d = 1; d < 16; d++)
			run[d] = H->nc[d];
		for (u32 s0 = 0; s0 < n; s0 += 64) {
			u32 s = s0 + lane;
			u32 l = s < n ? lens[s] : 0;
			u32 mycode = 0;
#pragma unroll
			for (u32 d = 1; d < 16; d++) {
				u64 mm = __ballot(l == d);
				if (l == d)
					mycode = run[d] + __builtin_popcountll(mm & ((1ull << lane) - 1));
				run[d] += __builtin_popcountll(mm);
			}
			if (s < n)
				codes[s] = l ? (u16)(__brev(mycode) >> (32 - l)) : 0;
		}
	}
	wave_sync();
}

/* ---------------- bit output through the LDS staging area ---------------- */

struct outstate {
	u8 *out;		/* output slot of this buffer */
	u64 avail;
	u64 sg;			/* global byte offset (relative to out, may be
				 * negative via wrap) of staging word 0; 16-aligned
				 * as an absolute address */
	u64 bits;		/* bits produced so far, relative to out[0] */
};

static __device__ __forceinline__ u32 *stg_of(lds_t *L)
{
	return (u32 *)L->nxtA;
}

/* OR 'nbits' (<= 57) bits of 'code' at absolute bit position 'bitpos' */
static __device__ __forceinline__ void
stg_put(lds_t *L, const struct outstate *os, u64 bitpos, u64 code,
	u32 nbits)
{
	if (!nbits)
		return;
	u64 rel = bitpos - 8 * os->sg;	/* sg <= bitpos/8 by construction */
	u32 w = (u32)(rel >> 5), s = (u32)rel & 31;
	u32 *stg = stg_of(L);
	u64 lo = code << s;
	atomicOr((u32 *)&stg[w], (u32)lo);
	if (s + nbits > 32)
		atomicOr((u32 *)&stg[w + 1], (u32)(lo >> 32));
	if (s + nbits > 64)
		atomicOr((u32 *)&stg[w + 2], (u32)(code >> (64 - s)));
}

/*
 * Write the completed bytes of the staging area to HBM and slide the rest to
 * the front.  Whole workgroup; 'final' also writes the last partial unit.
 */
static __device__ __forceinline__ void
stg_flush(lds_t *L, struct outstate *os, bool final)
{
	u32 *stg = stg_of(L);
	u8 *stgb = (u8 *)stg;
	const u32 tid = threadIdx.x;
	u64 done_bytes = final ? (os->bits + 7) / 8 : os->bits / 8;
	u64 rel_end = done_bytes - os->sg;	/* staging bytes that are final */
	s64 first = -(s64)os->sg;		/* staging index of out[0] if sg<0 */
	u32 start = first > 0 ? (u32)first : 0;
	u32 units = final ? (u32)((rel_end + 15) / 16) : (u32)(rel_end / 16);

	__syncthreads();
	/* 16-byte units: unit u covers staging bytes [16u, 16u+16) */
	for (u32 u = tid; u < units; u += NT) {
		u32 b0 = u * 16, b1 = b0 + 16;
		u8 *g = os->out + (s64)(os->sg + b0);
		if (b0 >= start && b1 <= rel_end) {
			*(uint4 *)g = *(const uint4 *)(stgb + b0);
		} else {
			for (u32 b = b0 < start ? start : b0; b < b1 && b < rel_end; b++)
				g[b - b0] = stgb[b];
		}
	}
	/* slide the unfinished tail to the front; thread i both clears word i
	 * and (for the few tail words) rewrites it, so no barrier in between */
	u32 keep_from = units * 16;
	u32 total_words = (u32)((os->bits - 8 * os->sg + 31) / 32) + 1;
	u32 keep_words = final ? 0 : total_words - keep_from / 4;
	u32 v = 0;
	if (tid < keep_words && keep_from / 4 + tid < STG_WORDS + 8)
		v = stg[keep_from / 4 + tid];
	__syncthreads();
	for (u32 i = tid; i < STG_WORDS + 8; i += NT)
		stg[i] = 0;
	if (tid < keep_words)
		stg[tid] = v;
	os->sg += keep_from;
	/* callers put a barrier before the next stg_put by another thread */
}

/* bring back the few unfinished bytes saved in carry[] (the staging area
 * shares LDS with the tile scratch and is clobbered between blocks) */
static __device__ __forceinline__ void stg_restore(lds_t *L)
{
	u32 *stg = stg_of(L);

	__syncthreads();
	for (u32 i = threadIdx.x; i < STG_WORDS + 8; i += NT)
		stg[i] = i < 6 ? L->carry[i] : 0;
	__syncthreads();
}

static __device__ __forceinline__ void stg_save(lds_t *L, struct outstate *os)
{
	stg_flush(L, os, false);
	if (threadIdx.x < 6)
		L->carry[threadIdx.x] = stg_of(L)[threadIdx.x];
	__syncthreads();
}

#ifdef LDA_DEBUG_SPLIT	/* per-tile trace of the block-split inputs of buffer 0 (debug builds) */
static __device__ u32 lda_dbg[2048];
extern "C" __attribute__((visibility("default"))) void libdeflate_amd_debug_read(u32 *out)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lda_dbg), sizeof(lda_dbg));
}
#define DBG(tile, k, v) do { if ((tile) < 250) lda_dbg[8 * (tile) + (k)] = (v); } while (0)
#else
#define DBG(tile, k, v) do { } while (0)
#endif

/* ---------------- the kernel ---------------- */

/*
 * The body is compiled twice: OPT = false for levels 0-9 (the min-cost parse
 * and its re-parse loop compile away, so the lazy levels keep their register
 * allocation), OPT = true for levels 10-12.
 */
template <bool OPT> static __device__ __forceinline__ void
deflate_batch_body(u8 *lds_raw, u64 n_chunks, int format, int level, u32 depth,
		   u32 nice, u32 mode,
		   const u8 *__restrict__ in_base,
		   const u64 *__restrict__ in_offsets,
		   const u64 *__restrict__ in_nbytes,
		   u8 *__restrict__ out_base,
		   const u64 *__restrict__ out_offsets,
		   const u64 *__restrict__ out_avail_arr,
		   u64 *__restrict__ out_nbytes,
		   const u32 *__restrict__ sums,
		   u64 *__restrict__ seq_scratch,
		   const u32 *__restrict__ seg_info,
		   u32 *__restrict__ next_chunk)
{
	lds_t *L = (lds_t *)(uintptr_t)0;
	const u32 tid = threadIdx.x;
	if ((u32)(uintptr_t)(__attribute__((address_space(3))) u8 *)lds_raw != 0)
		__builtin_trap();	/* see LDS32(): the dynamic LDS block must start at 0 */
	/* block-relative position | length << 32 | distance << 41 */
	u64 *__restrict__ seqg = seq_scratch + (size_t)blockIdx.x * SEQ_STRIDE;
	/* levels 10-12: the search results of a block's first tile, kept while
	 * that tile is parsed more than once */
	u32 *__restrict__ msave = (u32 *)(seqg + SEQ_GCAP);
	/* the block histogram before the current tile [320] and the tile's own
	 * share [320], for a block that ends in front of the tile */
	u32 *__restrict__ fsave = msave + TILE + 8;
	/* levels 10-12: byte histogram of the block before the previous tile
	 * [256] and of the previous tile [256] */
	u32 *__restrict__ bsave = fsave + 640;
	u32 tog = 0;		/* which scan[] array the next single-barrier scan uses */
	PROF_DECL;

	/* Buffers are handed out dynamically (one global counter): their cost
	 * depends on their content, and a fixed stride gives every workgroup
	 * the same kind of buffer whenever the batch is periodic. */
	for (;;) {
		__syncthreads();
		if (tid == 0)
			L->vars[V_TMP0] = atomicAdd(next_chunk, 1u);
		__syncthreads();
		const u64 c = L->vars[V_TMP0];
		if (c >= n_chunks)
			break;
		const u8 *inp = in_base + in_offsets[c];
		const u64 n64 = in_nbytes[c];
		struct outstate os;
		os.out = out_base + out_offsets[c];
		os.avail = out_avail_arr[c];
		const u32 hdr_bytes = format == LDA_FMT_GZIP ? 10 :
				      format == LDA_FMT_ZLIB ? 2 : 0;
		const u32 ftr_bytes = format == LDA_FMT_GZIP ? 8 :
				      format == LDA_FMT_ZLIB ? 4 : 0;
		bool overflow = false;

		/* the reference refuses outright when the container cannot fit
		 * (gzip_compress.c:41-42, zlib_compress.c:42-43) */
		if (os.avail <= hdr_bytes + ftr_bytes && (hdr_bytes + ftr_bytes))
			overflow = true;
		if (n64 > 0xFFFFFF00u)	/* positions are 32-bit here */
			overflow = true;
		const u32 n = (u32)n64;

		/* ---- per-buffer init ---- */
		__syncthreads();
		PROF_START();
		for (u32 i = tid; i < (1u << HASH_BITS) / 2; i += NT)
			((u32 *)L->head)[i] = 0x80008000u;
		for (u32 i = tid; i < (1u << HASH3_BITS) / 2; i += NT)
			((u32 *)L->head3)[i] = 0x80008000u;
		for (u32 i = tid; i < 320; i += NT)
			L->freq[i] = 0;
		if (tid < 10)
			L->obs[0][tid] = 0;
		for (u32 i = tid; i < STG_WORDS + 8; i += NT)
			stg_of(L)[i] = 0;
		if (tid < 8)
			L->M[tid] = 0;
		if (tid == 0) {
			L->vars[V_NSEQ] = 0;
			L->vars[V_ENTRY] = 0;
		}
		os.sg = (u64)(0 - ((uintptr_t)os.out & 15));
		os.bits = 0;
		__syncthreads();

		/* container header through the staging area */
		if (!overflow && hdr_bytes) {
			if (tid == 0) {
				if (format == LDA_FMT_GZIP) {
					/* gzip_compress.c:44-64: XFL 4 fastest, 2 best */
					u32 xfl = level < 2 ? 4 : level >= 8 ? 2 : 0;
					stg_put(L, &os, 0, 0x00088B1Full, 32);
					stg_put(L, &os, 32, 0, 32);	/* MTIME */
					stg_put(L, &os, 64, xfl | (0xFFu << 8), 16);
				} else {
					/* zlib_compress.c:45-60 */
					u32 fl = level < 2 ? 0 : level < 6 ? 1 :
						 level < 8 ? 2 : 3;
					u32 h = (0x78u << 8) | (fl << 6);
					h |= 31 - (h % 31);
					stg_put(L, &os, 0, ((h & 0xFF) << 8) | (h >> 8), 16);
				}
			}
			os.bits = 8 * hdr_bytes;
		}
		__syncthreads();
		stg_save(L, &os);

		/* Segment mode (one large buffer cut into sub-ranges that are
		 * compressed side by side): the first dict_len bytes (whole tiles)
		 * of this "chunk" are the tail of the previous sub-range; they only
		 * prime the hash chains.  Every segment but the last ends with a
		 * non-final block and an empty stored block, so the segments'
		 * outputs are byte aligned and concatenate into one stream
		 * (lib/deflate_compress.c:1839-1847 is the same alignment rule). */
		const u32 sinfo = seg_info ? seg_info[c] : 0x80000000u;
		const u32 dict_len = sinfo & 0x7FFFFFFFu;
		const bool seg_last = sinfo >> 31;
		u32 loaded = 0;		/* input bytes present in the ring */
		u32 block_start = dict_len;
		u32 walkpos = dict_len;	/* absolute position the parse has reached */
		const bool aligned_in = ((uintptr_t)inp & 15) == 0;

		/* level 0 and tiny inputs: stored blocks only
		 * (deflate_compress.c:3925-3931, 2392-2443) */
		const bool stored_only = level == 0 ||
			n - dict_len <= (u32)(55 - 4 * (level > 12 ? 12 : level));
		u32 num_tiles = (n + TILE - 1) / TILE;
		if (num_tiles == 0)
			num_tiles = 1;

		for (u32 tile = 0; tile < num_tiles && !overflow; tile++) {
			const u32 t = tile * TILE;
			const u32 tend = t + TILE < n ? t + TILE : n;
			const bool last_tile = tile + 1 == num_tiles;
			const bool prime = t < dict_len;	/* dictionary tile (whole tiles) */
			/* the thread index is made opaque once per tile: otherwise every
			 * per-lane address in this loop body is computed before the loop
			 * and kept alive (in scratch) across it */
			u32 tid_opaque = threadIdx.x;
			asm volatile("" : "+v"(tid_opaque));
			const u32 tid = tid_opaque, lane = tid & 63, wave = tid >> 6;

			PROF_MARK(0);
			/* ---- S0: stage input up to tend + LOOKAHEAD ---- */
			u32 want = tend + LOOKAHEAD < n ? tend + LOOKAHEAD : n;
			if (aligned_in) {
				u32 from = loaded & ~15u;
				for (u32 p = from + tid * 16; p < want; p += NT * 16) {
					uint4 v = *(const uint4 *)(inp + p);
					*(uint4 *)&L->in[p & RMASK] = v;
					if ((p & RMASK) < 32)
						*(uint4 *)&L->in[RING + (p & RMASK)] = v;
				}
			} else {
				for (u32 p = loaded + tid; p < want; p += NT) {
					u8 b = inp[p];
					L->in[p & RMASK] = b;
					if ((p & RMASK) < 32)
						L->in[RING + (p & RMASK)] = b;
				}
			}
			loaded = want;
			for (u32 i = tid; i < TILE + 8; i += NT)
				L->mark[i] = 0;
			if (tid < 4)
				L->M[TILE + 4 + tid] = 0;
			__syncthreads();
			if (!prime && !stored_only) {
				/* minimum match length from the distinct bytes of this
				 * tile's input (calculate_min_match_len,
				 * deflate_compress.c:2329-2353, which the reference applies
				 * to the first 4096 bytes and then refreshes per block from
				 * the literals used; with blocks as long as a buffer the
				 * per-tile estimate is what follows content changes) */
				u32 *seen = L->M + 16;
				for (u32 i = tid; i < 256; i += NT)
					seen[i] = 0;
				__syncthreads();
				u32 lim = want - t < 4096 ? want - t : 4096;
				for (u32 i = tid; i < lim; i += NT)
					seen[L->in[(t + i) & RMASK]] = 1;
				__syncthreads();
				u32 c1 = tid < 256 ? seen[tid] : 0, tot1;
				(void)block_scan(L, c1, &tot1);
				if (tid == 0)
					L->vars[V_MINLEN] = n - dict_len < 512 ? 3 :
							    choose_min_len(tot1, depth);
				__syncthreads();
			} else if (tile == 0 && tid == 0) {
				L->vars[V_MINLEN] = 3;	/* dictionary tiles: not used */
			}

			PROF_MARK(1);
			if (!stored_only) {
				/* ---- S1: hash + in-wave sort -> local links ---- */
#pragma unroll
				for (u32 gi = 0; gi < TILE / 64 / NWAVES; gi++) {
					u32 g = wave + NWAVES * gi;
					u32 gb = t + g * 64;
					u32 p = gb + lane;
					bool valid = p + 4 <= n;
					u32 w4 = ld32(L->in, p);
					u32 h = valid ? hash4(w4) : (0x4000u | lane);
					u32 key = (h << 6) | lane;
					/* 3-byte hash rides along in M (bits 19..30) for S2;
					 * written before the sort scatters the hash4 part */
					u32 h3v = p + 3 <= n ? (hash3(w4) | 0x1000u) : 0;
#define SORT_STEP(K, J)                                                        \
	do {                                                                   \
		u32 other = lane_xor<J>(key);                                  \
		bool up = (lane & (K)) == 0, lower = (lane & (J)) == 0;        \
		u32 mn = key < other ? key : other;                            \
		u32 mx = key < other ? other : key;                            \
		key = (lower == up) ? mn : mx;                                 \
	} while (0)
					SORT_STEP(2, 1);
					SORT_STEP(4, 2); SORT_STEP(4, 1);
					SORT_STEP(8, 4); SORT_STEP(8, 2); SORT_STEP(8, 1);
					SORT_STEP(16, 8); SORT_STEP(16, 4); SORT_STEP(16, 2);
					SORT_STEP(16, 1);
					SORT_STEP(32, 16); SORT_STEP(32, 8); SORT_STEP(32, 4);
					SORT_STEP(32, 2); SORT_STEP(32, 1);
					SORT_STEP(64, 32); SORT_STEP(64, 16); SORT_STEP(64, 8);
					SORT_STEP(64, 4); SORT_STEP(64, 2); SORT_STEP(64, 1);
#undef SORT_STEP
					/* neighbours: DPP wave shifts (lane 0 / 63 keep their own) */
					u32 left = __builtin_amdgcn_update_dpp(key, key, 0x138, 0xF, 0xF, false);
					u32 right = __builtin_amdgcn_update_dpp(key, key, 0x130, 0xF, 0xF, false);
					bool first = lane == 0 || (left >> 6) != (key >> 6);
					bool last = lane == 63 || (right >> 6) != (key >> 6);
					u32 orig = key & 63, hh = key >> 6;
					bool isv = hh < 0x4000u;
					if (isv && !first)
						L->prev[(gb + orig) & RMASK] =
							(u16)(gb + (left & 63));
					L->nxtB[4 + g * 64 + lane] = (u16)h3v;
					L->M[4 + g * 64 + orig] = hh |
						(first ? M_FIRST : 0) |
						(last ? M_LAST : 0) |
						(isv ? M_VALID : 0);
				}
				if (tid == NT - 1) {
					L->vars[V_CTR] = 0;
					L->vars[V_READY] = 0;
				}
				__syncthreads();

				PROF_MARK(2);
				/* ---- S2: thread groups through head[] in order ----
				 * Only wave 0 (hash chains) and wave 1 (3-byte table) do
				 * this; chains only ever look backwards, so the other waves
				 * start searching at once and wave 0 publishes how far the
				 * chains are complete (V_READY) after every batch of groups;
				 * claims beyond that wait. */
				/* A wave's LDS operations execute in issue order, so the
				 * read-old-head / write-new-head pairs of all groups are
				 * issued back to back (no wait in between); the values read
				 * are stored to prev[] afterwards. */
				if (wave == 0) {
					const u32 ngroups = (tend - t + 63) / 64;
					enum { GB = 8 };	/* groups in flight */
					for (u32 g0 = 0; g0 < ngroups; g0 += GB) {
						u32 v[GB];
#pragma unroll
						for (u32 k = 0; k < GB; k++)
							v[k] = g0 + k < ngroups ?
								L->M[4 + (g0 + k) * 64 + lane] : 0;
#pragma unroll
						for (u32 k = 0; k < GB; k++) {
							u32 m = v[k];
							u32 h = m & ((1u << HASH_BITS) - 1);
							u32 old = 0xFFFFFFFFu;
							if (m & M_VALID) {
								if (m & M_FIRST)
									old = L->head[h];
								if (m & M_LAST)
									L->head[h] = (u16)(t + (g0 + k) * 64 + lane);
							}
							v[k] = old;
						}
#pragma unroll
						for (u32 k = 0; k < GB; k++)
							if (v[k] != 0xFFFFFFFFu)
								L->prev[(t + (g0 + k) * 64 + lane) & RMASK] =
									(u16)v[k];
						/* LDS writes of a wave land in issue order */
						if (lane == 0)
							*(volatile u32 *)&L->vars[V_READY] =
								g0 + GB < ngroups ? (g0 + GB) * 64 : TILE;
					}
					if (lane == 0)
						*(volatile u32 *)&L->vars[V_READY] = TILE;
				} else if (wave == 1) {
					/* 3-byte table, same order, on its own wave: candidate
					 * = last position of an EARLIER group with this hash */
					const u32 ngroups = (tend - t + 63) / 64;
					enum { GB = 16 };
					for (u32 g0 = 0; g0 < ngroups; g0 += GB) {
						u32 v[GB];
#pragma unroll
						for (u32 k = 0; k < GB; k++)
							v[k] = g0 + k < ngroups ?
								L->nxtB[4 + (g0 + k) * 64 + lane] : 0;
#pragma unroll
						for (u32 k = 0; k < GB; k++) {
							u32 h3v = v[k], c3 = 0;
							if (h3v) {
								c3 = L->head3[h3v & 0xFFF];
								L->head3[h3v & 0xFFF] =
									(u16)(t + (g0 + k) * 64 + lane);
							}
							v[k] = c3;
						}
#pragma unroll
						for (u32 k = 0; k < GB; k++)
							if (g0 + k < ngroups)
								L->nxtA[4 + (g0 + k) * 64 + lane] = (u16)v[k];
					}
				}
				if (prime) {	/* chains primed; nothing to search or emit */
					__syncthreads();
					continue;
				}

				PROF_MARK(3);
				/* ---- S3: all positions search their chain ----
				 * Chain lengths differ wildly between positions, so lanes
				 * do not own fixed positions: a lane claims the next
				 * unsearched position from a workgroup counter when it
				 * finishes one.  The kernel is instruction-issue bound, so
				 * the search is split into two kinds of wave-uniform passes:
				 *   walk      S3_WALK chain steps per lane, nothing but the
				 *             link chase and the 4-byte compare; hits are
				 *             queued (<= 4 distances packed in a u64);
				 *   evaluate  every lane pops its oldest (closest) hit and
				 *             measures it: 8 bytes against the cached bytes
				 *             p+4..p+11, longer ones by the whole wave, 256
				 *             bytes per pass.
				 * Positions that end without a match >= 4 get their
				 * length-3 probe in a separate position-parallel pass. */
				{
					s32 lo = (s32)(t + TILE + LOOKAHEAD) - (s32)RING;
					const u32 lo_pos = lo > 0 ? (u32)lo : 0;
					const u32 min_len = L->vars[V_MINLEN];
					const u32 drain = depth < 8 ? depth : depth >> 3 > 8 ? depth >> 3 : 8;
					u32 my_i = 0xFFFFFFFFu, p = 0, cur = 0, c16 = 0, dmaxp = 0,
					    maxlen = 0, dep = 0, best = 3, bestd = 0, dprev = 0,
					    cnt = 0, boff = 0, curb = 0;
					u64 nxt8 = 0, q = 0;
					bool have = false, fin = true, ended = false;
					PROF_SEC_DECL;

					for (;;) {
						PROF_SEC(3);
						u64 mh = __ballot(have);
						u32 nf = __builtin_popcountll(__ballot(fin));
						if (!mh && !nf)
							break;
						if (nf && (nf >= S3_CLAIM || !mh)) {
							/* one counter update per wave; ranks by ballot */
							const u64 fm = __ballot(fin);
							u32 cbase = 0;
							if (lane == 0)
								cbase = atomicAdd((u32 *)&L->vars[V_CTR], nf);
							cbase = bcast_first(cbase);
							{	/* chains complete up to the claimed ones? */
								const u32 need = cbase + nf < TILE ? cbase + nf : TILE;
								if (cbase < TILE)
									while (*(volatile u32 *)&L->vars[V_READY] < need)
										__builtin_amdgcn_s_sleep(4);
							}
							if (fin) {
								if (my_i < TILE)
									L->M[4 + my_i] =
										best >= 4 && best >= min_len ?
										(best | (bestd << 16)) : 0;
								fin = false;
								best = 3;
								my_i = cbase + __builtin_popcountll(fm & ((1ull << lane) - 1));
								if (my_i < TILE) {
									p = t + my_i;
									if (p + 4 <= n) {
										cur = ld32(L->in, p);
										curb = cur;
										boff = 0;
										nxt8 = ld64(L->in, p + 4);
										c16 = LDS16(PREV_OFF + 2 * (p & RMASK));
										maxlen = n - p < 258 ? n - p : 258;
										dmaxp = p - lo_pos;
										/* the last positions of a tile are claimed when
										 * the other waves are about to run dry: they
										 * search less deep (by position, so the output
										 * does not depend on timing) */
										dep = my_i >= TILE - S3_TAIL ? drain : depth;
										bestd = 0;
										dprev = 0;
										cnt = 0;
										ended = false;
										have = true;
									} else {
										fin = true;	/* M = 0; len 3 later */
									}
								}
							}
							PROF_SEC(0);
							continue;
						}
						/* walk */
#pragma unroll
						for (int s = 0; s < S3_WALK; s++) {
							u32 d = (p - c16) & 0xFFFF;
							bool chain = dep && d > dprev && d <= dmaxp;
							bool stall = ended || cnt >= 4;
							bool ok = !stall && chain;
							u32 cp = p - d;
							u32 w = ld32(L->in, cp + boff);
							u32 c16n = LDS16(PREV_OFF + 2 * (cp & RMASK));
							bool hit = ok && w == curb;
							c16 = ok ? c16n : c16;
							dprev = ok ? d : dprev;
							dep -= ok ? 1 : 0;
							ended = ended || (!stall && !chain);
							q = hit ? ((q << 16) | d) : q;
							cnt += hit ? 1 : 0;
						}
						if (!have)
							cnt = 0;
						PROF_SEC(1);
						/* evaluate: a round when enough lanes hold a hit, or
						 * when no lane can walk any further */
						bool done = false;
						u32 nev = __builtin_popcountll(__ballot(cnt > 0));
						u32 nwalk = __builtin_popcountll(__ballot(have && !ended && cnt < 4));
						while (nev && (nev >= S3_EVMIN || !nwalk)) {
							bool ev = cnt > 0;
							u32 d = (u32)(q >> (16 * ((cnt - 1) & 3))) & 0xFFFF;
							cnt -= ev ? 1 : 0;
							u32 cp = p - d;
							u64 x = nxt8 ^ ld64(L->in, cp + 4);
							u32 len = 4 + ((u32)__builtin_ctzll(x | (1ull << 63)) >> 3);
							ev = ev && ld32(L->in, cp) == cur;
							bool more = ev && x == 0 && 12 < maxlen;
							/* bytes 12..27 lane by lane; only longer matches
							 * go to the wave */
							if (__ballot(more)) {
								u64 y = ld64(L->in, p + 12) ^ ld64(L->in, cp + 12);
								if (more) {
									len = 12 + ((u32)__builtin_ctzll(y | (1ull << 63)) >> 3);
									more = y == 0 && 20 < maxlen;
								}
								if (__ballot(more)) {
									u64 z = ld64(L->in, p + 20) ^ ld64(L->in, cp + 20);
									if (more) {
										len = 20 + ((u32)__builtin_ctzll(z | (1ull << 63)) >> 3);
										more = z == 0 && 28 < maxlen;
									}
								}
							}
							for (u64 mm = __ballot(more); mm; mm &= mm - 1) {
								u32 src = (u32)__builtin_ctzll(mm);
								u32 bp = bcast_lane(p, src);
								u32 bc = bcast_lane(cp, src);
								u32 bmax = bcast_lane(maxlen, src);
								u32 off = 28 + 4 * lane;
								u32 x4 = off < bmax ?
									(ld32(L->in, bp + off) ^
									 ld32(L->in, bc + off)) : 1;
								u64 ne = __ballot(x4 != 0);
								u32 tot = bmax;	/* 28 + 256 >= 258 */
								if (ne) {
									u32 kk = (u32)__builtin_ctzll(ne);
									u32 xk = bcast_lane(x4, kk);
									u32 o = 28 + 4 * kk;
									if (o < bmax)
										tot = o + ((u32)__builtin_ctz(xk) >> 3);
								}
								if (lane == src)
									len = tot;
							}
							if (ev) {
								if (len > maxlen)
									len = maxlen;
								if (len > best) {
									best = len;
									bestd = d;
									if (len >= nice || len >= maxlen) {
										done = true;
										cnt = 0;
									} else {
										boff = len - 3;
										curb = ld32(L->in, p + boff);
									}
								}
							}
							if (have && (done || (ended && cnt == 0))) {
								have = false;
								fin = true;
								done = false;
							}
							nev = __builtin_popcountll(__ballot(cnt > 0));
							nwalk = __builtin_popcountll(__ballot(have && !ended && cnt < 4));
						}
						if (have && ended && cnt == 0) {
							have = false;
							fin = true;
						}
						PROF_SEC(2);
					}
					PROF_SEC_FLUSH(17);
				}
				__syncthreads();
				PROF_MARK(16);
				/* length-3 matches for the positions left without a match */
				if (L->vars[V_MINLEN] <= 3) {
					s32 lo = (s32)(t + TILE + LOOKAHEAD) - (s32)RING;
					const u32 lo_pos = lo > 0 ? (u32)lo : 0;
					const u32 dlim3 = mode ? 8192u : 4096u;
					for (u32 i = tid; i < TILE; i += NT) {
						u32 p = t + i, b3 = 0;
						if (p + 3 <= n && L->M[4 + i] == 0) {
							u32 bd = find_len3(L, p, ld32(L->in, p),
									   L->nxtA[4 + i], p - lo_pos,
									   dlim3, &b3);
							if (bd)
								L->M[4 + i] = 3 | (bd << 16);
						}
					}
				}
				__syncthreads();

				PROF_MARK(4);
				/* the block as it is before this tile's tokens: if the tile
				 * turns out to be of different content, the block ends in
				 * front of it (see "block end?") */
				if (tid < 320)
					fsave[tid] = L->freq[tid];
				if (tid == 0) {
					L->vars[V_NSEQ_PRE] = L->vars[V_NSEQ];
					L->vars[V_WPOS_PRE] = walkpos;
				}
				/* levels 10-12 (mode 3): min-cost parse, see opt_parse_wave().
				 * A block's first tile is parsed lazily as a dry run (stage 0),
				 * rolled back, and parsed again with prices from that run
				 * (stage 1; level 11 and up once more with the prices of stage
				 * 1); later tiles take their prices from the block so far. */
				const bool opt = OPT && mode == 3;
				const bool opt_first = opt && walkpos == block_start;
				const u32 opt_last = level >= 11 ? 2 : 1;
				u32 opt_stage = 0;
				const u32 ent0 = L->vars[V_ENTRY], nseq0 = L->vars[V_NSEQ];
				if (opt_first) {
					for (u32 i = tid; i < TILE + 8; i += NT)
						msave[i] = L->M[i];
					if (tid == 0)
						L->vars[V_FIT] = 0;
				} else if (opt) {
					/* a tile the block's statistics do not describe keeps
					 * the lazy parse (stage 0, final) */
					const u32 fit = opt_build_costs(L, tid, false, true, t, tend - t, bsave);
					opt_stage = fit ? 0 : 1;
					if (tid == 0) {
						L->vars[V_FIT] = fit;
						if (c == 0) {
							DBG(tile, 6, 100 + fit);
							DBG(tile, 7, L->vars[V_TMP3]);
						}
					}
				}
				for (;;) {
				/* opaque again: see the top of the tile loop */
				u32 tid_opaque2 = threadIdx.x;
				asm volatile("" : "+v"(tid_opaque2));
				const u32 tid = tid_opaque2, lane = tid & 63, wave = tid >> 6;
				const u32 s4mode = opt ? (opt_stage ? 0 : 2) : mode;
				if (opt && opt_stage) {
					const s32 ent = (s32)ent0;
					const s32 lim = last_tile ? (s32)(tend - t) : (s32)TILE - 2;
					const s32 pend = (s32)(tend - t);
					s32 lo = (s32)(OPT_SEG * wave), hi = lo + OPT_SEG;
					if (wave == 0 && ent < 0)
						lo = ent;
					if (hi > lim)
						hi = lim;
					s32 e = hi + OPT_WARM;
					if (e > pend)
						e = pend;
					if (lo < hi)
						opt_parse_wave(L, L->nxtA, t, lo, hi, e, lane);
					__syncthreads();
					for (s32 i = (s32)tid + (ent < 0 ? ent : 0); i < lim; i += NT) {
						u32 c = L->nxtA[i + 4], m = L->M[i + 4];
						L->M[i + 4] = c >= 3 ? c | (m & 0xFFFF0000u) : 0;
					}
					__syncthreads();
				}
				/* ---- S4: token choice by pointer jumping ----
				 * step(p) is a pure function of M[p..p+2]; the chosen
				 * tokens are the positions reachable from the entry point
				 * by p -> p + step(p); idx = p + 4.  Each wave owns a
				 * segment of 128 idx (the last one 132) and works without
				 * workgroup barriers: 8 doubling rounds give, for every idx,
				 * the first position its path reaches outside the segment;
				 * after ONE barrier every wave chains the segment exits up
				 * to its own entry point and marks its part of the path by
				 * replaying the doubling steps from the largest down. */
				{
					const s32 entry = (s32)L->vars[V_ENTRY];
					const s32 limit = last_tile ? (s32)(tend - t) :
							  (s32)TILE - 2;
					const u32 lim_idx = (u32)(limit + 4);
					enum { SEG = TILE / NWAVES, SL = SEG / 64,
					       JR = SEG == 128 ? 7 : 8 };	/* 2^JR >= SEG */
					/* wave w owns idx [4 + SEG w, 4 + SEG (w + 1)); the
					 * carried-in idx 2, 3 can only be the entry itself */
					const u32 seg_lo = 4 + SEG * wave, seg_hi = seg_lo + SEG;
					u16 *J = L->nxtA, *Jn = L->nxtB;
					u32 jh[JR][SL], jc[SL];	/* J^(2^r) of the own idx */
					u32 m0[SL];		/* M of the own idx */
#pragma unroll
					for (u32 k = 0; k < SL; k++) {
						u32 idx = seg_lo + lane + 64 * k;
						u32 p = idx - 4;
						u32 nx = p;
						m0[k] = L->M[idx];
						if ((s32)p < limit)
							nx = p + token_step(m0[k], L->M[idx + 1],
									    L->M[idx + 2], s4mode, nice);
						jc[k] = nx + 4;	/* stored as idx */
						J[idx] = (u16)jc[k];
					}
					wave_sync();
#pragma unroll
					for (u32 r = 0; r < JR; r++) {
#pragma unroll
						for (u32 k = 0; k < SL; k++) {
							u32 idx = seg_lo + lane + 64 * k;
							u32 q = jc[k];
							jh[r][k] = q;
							if (q < seg_hi && q < lim_idx)
								jc[k] = J[q];
							Jn[idx] = (u16)jc[k];
						}
						wave_sync();
						u16 *tmp = J; J = Jn; Jn = tmp;
					}
					__syncthreads();
					PROF_MARK(12);
					u32 e = (u32)(entry + 4);
					const u32 seq0 = L->vars[V_NSEQ];
					u32 npre = 0;
					for (u32 pre = 0; pre < 2; pre++) {	/* idx 2, 3 */
						if (e < 4 && e < lim_idx) {
							u32 mm = L->M[e];
							u32 st = token_step(mm, L->M[e + 1], L->M[e + 2],
									    s4mode, nice);
							u32 l0 = mm & 0xFFFF;
							bool ism = st == l0 && l0;
							if (tid == 0) {
								s32 np = (s32)e - 4 + (s32)st;
								u32 pos = (u32)((s32)t + (s32)e - 4);
								if (e + st >= lim_idx) {
									L->vars[V_WALKPOS_LO] = (u32)((s32)t + np);
									L->vars[V_ENTRY] = (u32)(np - (s32)TILE);
								}
								if (ism) {
									u32 sl, xb, xv;
									seqg[seq0 + npre] = (pos - block_start) |
										((u64)l0 << 32) | ((u64)(mm >> 16) << 41);
									length_code(l0, &sl, &xb, &xv);
									atomicAdd((u32 *)&L->freq[257 + sl], 1u);
									dist_code(mm >> 16, &sl, &xb, &xv);
									atomicAdd((u32 *)&L->freq[288 + sl], 1u);
								} else {
									atomicAdd((u32 *)&L->freq[L->in[pos & RMASK]], 1u);
									if (st == 2)
										atomicAdd((u32 *)&L->freq[L->in[(pos + 1) & RMASK]], 1u);
								}
							}
							npre += ism ? 1 : 0;
							e += st;
						}
					}
					for (u32 sgm = 0; sgm < wave; sgm++)
						if (e < 4 + SEG * (sgm + 1) && e < lim_idx)
							e = J[e];
					bool mk[SL];
#pragma unroll
					for (u32 k = 0; k < SL; k++) {
						u32 idx = seg_lo + lane + 64 * k;
						mk[k] = idx == e && e < lim_idx;
					}
#pragma unroll
					for (s32 r = JR - 1; r >= 0; r--) {
#pragma unroll
						for (u32 k = 0; k < SL; k++) {
							u32 q = jh[r][k];
							if (mk[k] && q < seg_hi && q < lim_idx)
								L->mark[q] = 1;
						}
						wave_sync();
#pragma unroll
						for (u32 k = 0; k < SL; k++) {
							u32 idx = seg_lo + lane + 64 * k;
							mk[k] = mk[k] || L->mark[idx];
						}
					}
					PROF_MARK(13);
					/* emit: the lanes on the path classify their token,
					 * count it for the block's Huffman codes and append
					 * the matches to seq[] in position order (ballot ranks
					 * inside the wave, one workgroup scan across waves) */
					bool ism[SL];
#pragma unroll
					for (u32 k = 0; k < SL; k++) {
						u32 idx = seg_lo + lane + 64 * k;
						u32 st = jh[0][k] - idx, l0 = m0[k] & 0xFFFF;
						ism[k] = mk[k] && st == l0 && l0;
						if (mk[k]) {
							u32 pos = t + idx - 4;
							if (idx + st >= lim_idx) {
								s32 np = (s32)idx - 4 + (s32)st;
								L->vars[V_WALKPOS_LO] = (u32)((s32)t + np);
								L->vars[V_ENTRY] = (u32)(np - (s32)TILE);
							}
							if (ism[k]) {
								u32 sl, xb, xv;
								length_code(l0, &sl, &xb, &xv);
								atomicAdd((u32 *)&L->freq[257 + sl], 1u);
								dist_code(m0[k] >> 16, &sl, &xb, &xv);
								atomicAdd((u32 *)&L->freq[288 + sl], 1u);
							} else {
								atomicAdd((u32 *)&L->freq[L->in[pos & RMASK]], 1u);
								if (st == 2)
									atomicAdd((u32 *)&L->freq[L->in[(pos + 1) & RMASK]], 1u);
							}
						}
					}
					if (tid == 0 && entry >= limit) {
						L->vars[V_WALKPOS_LO] = (u32)((s32)t + entry);
						L->vars[V_ENTRY] = (u32)(entry - (s32)TILE);
					}
					u64 bal[SL];
					u32 cw = 0;
#pragma unroll
					for (u32 k = 0; k < SL; k++) {
						bal[k] = __ballot(ism[k]);
						cw += __builtin_popcountll(bal[k]);
					}
					u32 *sc = L->scan[tog];
					tog ^= 1;
					if (lane == 0)
						sc[wave] = cw;
					__syncthreads();
					u32 base = seq0 + npre, tot = 0;
#pragma unroll
					for (u32 w = 0; w < NWAVES; w++) {
						u32 c = sc[w];
						if (w < wave)
							base += c;
						tot += c;
					}
					const u64 lt = (1ull << lane) - 1;
#pragma unroll
					for (u32 k = 0; k < SL; k++) {
						if (ism[k]) {
							u32 idx = seg_lo + lane + 64 * k;
							u32 at = base + __builtin_popcountll(bal[k] & lt);
							seqg[at] = (t + idx - 4 - block_start) |
								   ((u64)(m0[k] & 0xFFFF) << 32) |
								   ((u64)(m0[k] >> 16) << 41);
						}
						base += __builtin_popcountll(bal[k]);
					}
					if (tid == 0)
						L->vars[V_NSEQ] = seq0 + npre + tot;
					/* the match list is read back by other waves at the
					 * end of the block */
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				}
				__syncthreads();
				if (!opt_first || opt_stage == opt_last)
					break;
				/* prices from this parse, then undo it */
				opt_build_costs(L, tid, opt_stage == 0, false, t, tend - t, bsave);
				for (u32 i = tid; i < 320; i += NT)
					L->freq[i] = 0;
				for (u32 i = tid; i < TILE + 8; i += NT) {
					L->mark[i] = 0;
					L->M[i] = msave[i];
				}
				if (tid == 0) {
					L->vars[V_ENTRY] = ent0;
					L->vars[V_NSEQ] = nseq0;
				}
				__syncthreads();
				opt_stage++;
				}
				walkpos = L->vars[V_WALKPOS_LO];
				PROF_MARK(5);

				/* carry the last 4 match entries to the front for the
				 * positions the walk deferred */
				if (tid < 4)
					L->M[tid] = L->M[TILE + tid];
				/* block split observations (see "block end?" below), by the
				 * last wave while the others wait at the barrier */
				if (wave == NWAVES - 1) {
					/* class of literal sy = lane + 64 j is 2 j + (lane & 1);
					 * matches: length slots 0..5 (3..8) / 6..28 */
					u32 onow[10];
#pragma unroll
					for (u32 j = 0; j < 4; j++) {
						u32 f = L->freq[lane + 64 * j];
						onow[2 * j] = wave_sum(lane & 1 ? 0 : f);
						onow[2 * j + 1] = wave_sum(lane & 1 ? f : 0);
					}
					{
						u32 f = lane < 29 ? L->freq[257 + lane] : 0;
						onow[8] = wave_sum(lane < 6 ? f : 0);
						onow[9] = wave_sum(lane < 6 ? 0 : f);
					}
					u32 nprev = 0, nnew = 0;
					u64 delta = 0;
#pragma unroll
					for (u32 i = 0; i < 10; i++) {
						nprev += L->obs[0][i];
						nnew += onow[i] - L->obs[0][i];
					}
#pragma unroll
					for (u32 i = 0; i < 10; i++) {
						u64 a = (u64)(onow[i] - L->obs[0][i]) * nprev;
						u64 e = (u64)L->obs[0][i] * nnew;
						delta += a > e ? a - e : e - a;
					}
					/* cutoff 200/512 of the mass as :2179-2193; blocks below
					 * the minimum length of :2204 are never cut */
					bool sp = nprev && walkpos - block_start >= 5000 &&
						  delta >= (u64)nnew * 200 / 512 * nprev;
					if (OPT && mode == 3 && L->vars[V_FIT] == 2 &&
					    walkpos - block_start >= 5000)
						sp = true;	/* see opt_build_costs() */
					wave_sync();
#pragma unroll
					for (u32 i = 0; i < 10; i++)
						if (lane == i)
							L->obs[0][i] = sp ? 0 : onow[i];
					/* 2: the part before this tile is a block of its own
					 * (>= the minimum block length of :2204) */
					if (lane == 0)
						L->vars[V_SPLIT] = !sp ? 0 :
							L->vars[V_WPOS_PRE] - block_start >= 5000 ? 2 : 1;
					if (lane == 0 && c == 0) {
						DBG(tile, 0, nprev);
						DBG(tile, 1, nnew);
						DBG(tile, 2, (u32)(delta / (nprev ? nprev : 1)));
						DBG(tile, 3, sp);
						DBG(tile, 4, walkpos - block_start);
						DBG(tile, 5, onow[8] + onow[9]);
					}
				}
			} else {
				if (prime)
					continue;
				walkpos = tend;
			}
			__syncthreads();

			PROF_MARK(6);
			/* ---- block end? ----
			 * The reference ends a block when the kind of symbols changes
			 * (lib/deflate_compress.c:2092-2218): ten observation classes
			 * (literals by their top two bits and low bit, matches shorter /
			 * not shorter than 9), and a split when the distribution of the
			 * new observations is far from the block's.  Here the classes
			 * are sums over the block histogram, "new" is what this tile
			 * added, and the decision is taken per tile.  A tile that differs
			 * ends the block IN FRONT of itself: the block is written from
			 * the histogram and the match count saved before the tile, and
			 * the tile's tokens (already in the match list) become the start
			 * of the next block.  Only when that would leave a block shorter
			 * than the minimum, the block ends after the tile. */
			const u32 splitv = !stored_only && !last_tile ? L->vars[V_SPLIT] : 0;
			bool end_block = last_tile || splitv ||
				(!stored_only && L->vars[V_NSEQ] + SEQ_TILE_MAX > SEQ_GCAP) ||
				walkpos - block_start > MAX_BLOCK_LEN;
			if (!end_block)
				continue;
			if (tid < 10)
				L->obs[0][tid] = 0;

			const bool retro = splitv == 2;
			const u32 bstart = block_start;
			const u32 bend = last_tile ? n : retro ? L->vars[V_WPOS_PRE] : walkpos;
			const u32 blen = bend - bstart;
			const u32 nseq_all = stored_only ? 0 : L->vars[V_NSEQ];
			const u32 nseq = retro ? L->vars[V_NSEQ_PRE] : nseq_all;
			const u32 is_final = last_tile && seg_last ? 1 : 0;
			if (retro) {
				if (tid < 320) {
					u32 f = L->freq[tid], fp = fsave[tid];
					fsave[320 + tid] = f - fp;
					L->freq[tid] = fp;
				}
				__syncthreads();
			}

			/* ---- S5: codes, costs, block type ---- */
			u32 btype = 0;	/* 0 stored, 1 static, 2 dynamic */
			if (!stored_only) {
				if (tid == 0)
					L->freq[256]++;
				__syncthreads();
				/* rank sort of both alphabets by the whole workgroup:
				 * thread (s, part) counts the keys below key(s) in one
				 * third of the litlen alphabet; M[] is free scratch here */
				{
					u32 *rk = L->M;			/* [320] ranks */
					u32 *usedv = L->M + 320;	/* [2] used counts */
					u16 *sortedO = (u16 *)(L->M + 324);	/* [32] */
					for (u32 i = tid; i < 324; i += NT)
						L->M[i] = 0;
					__syncthreads();
					if (tid < 864) {
						u32 sidx = tid / 3, part = tid % 3;
						u32 f = L->freq[sidx];
						if (f) {
							u32 key = (f << 9) | sidx, r = 0;
							for (u32 q = part * 96; q < part * 96 + 96; q++) {
								u32 ft = L->freq[q];
								r += (ft != 0) & (((ft << 9) | q) < key);
							}
							atomicAdd(&rk[sidx], r);
							if (part == 0)
								atomicAdd(&usedv[0], 1u);
						}
					} else if (tid < 896) {
						u32 sidx = tid - 864;
						u32 f = L->freq[288 + sidx];
						if (f) {
							u32 key = (f << 9) | sidx, r = 0;
							for (u32 q = 0; q < 32; q++) {
								u32 ft = L->freq[288 + q];
								r += (ft != 0) & (((ft << 9) | q) < key);
							}
							rk[288 + sidx] = r;
							atomicAdd(&usedv[1], 1u);
						}
					}
					__syncthreads();
					if (tid < 288 && L->freq[tid])
						L->sorted[rk[tid]] = (u16)tid;
					else if (tid >= 288 && tid < 320 && L->freq[tid])
						sortedO[rk[tid]] = (u16)(tid - 288);
					__syncthreads();
					PROF_MARK(10);
					/* the two trees are built side by side on two waves */
					if (wave == 0)
						make_code(L->freq, 288, 15, L->lens, L->codes,
							  L->sorted, (huff_scratch<288> *)(L->M + 512),
							  usedv[0], true, lane);
					else if (wave == 1)
						make_code(L->freq + 288, 32, 15, L->lens + 288,
							  L->codes + 288, sortedO,
							  (huff_scratch<32> *)L->hw,
							  usedv[1], true, lane);
				}
				__syncthreads();
				PROF_MARK(11);
				/* precode items: run-length coding of the code lengths
				 * (deflate_compress.c:1482-1557 semantics), one thread per
				 * length, then one thread per run */
				{
					u32 *starts = L->M;		/* [<= 321] run start indices */
					if (tid == 0) {
						L->vars[V_TMP1] = 257;
						L->vars[V_TMP2] = 1;
					}
					if (tid < 19)
						L->pre_freq[tid] = 0;
					__syncthreads();
					if (tid < 288 && tid >= 257 && L->lens[tid])
						atomicMax((u32 *)&L->vars[V_TMP1], tid + 1);
					if (tid >= 288 && tid < 320 && L->lens[tid])
						atomicMax((u32 *)&L->vars[V_TMP2], tid - 288 + 1);
					__syncthreads();
					const u32 nlit = L->vars[V_TMP1], noff = L->vars[V_TMP2];
					const u32 total = nlit + noff;
					u32 v = 0, isst = 0;
					if (tid < total) {
						v = L->lens[tid < nlit ? tid : 288 + (tid - nlit)];
						u32 pv = 0xFF;
						if (tid)
							pv = L->lens[tid - 1 < nlit ? tid - 1 :
								     288 + (tid - 1 - nlit)];
						isst = pv != v;
					}
					u32 nruns;
					u32 ridx = block_scan(L, isst, &nruns);
					if (isst)
						starts[ridx] = tid;
					if (tid == 0)
						starts[nruns] = total;
					__syncthreads();
					/* thread r < nruns owns run r */
					u32 rv = 0, rlen = 0, nitems = 0;
					if (tid < nruns) {
						u32 st = starts[tid];
						rlen = starts[tid + 1] - st;
						rv = L->lens[st < nlit ? st : 288 + (st - nlit)];
						if (rv == 0) {
							u32 full = rlen / 138, rem = rlen % 138;
							nitems = full + (rem >= 3 ? 1 : rem);
						} else if (rlen >= 4) {
							u32 l1 = rlen - 1;
							nitems = 1 + l1 / 6 + (l1 % 6 >= 3 ? 1 : l1 % 6);
						} else {
							nitems = rlen;
						}
					}
					u32 ni;
					u32 at = block_scan(L, nitems, &ni);
					if (tid < nruns) {
						u32 left = rlen;
						if (rv == 0) {
							while (left >= 11) {
								u32 r = left > 138 ? 138 : left;
								L->pre_items[at++] = 18 | ((r - 11) << 5);
								atomicAdd((u32 *)&L->pre_freq[18], 1u);
								left -= r;
							}
							if (left >= 3) {
								L->pre_items[at++] = 17 | ((left - 3) << 5);
								atomicAdd((u32 *)&L->pre_freq[17], 1u);
								left = 0;
							}
						} else if (left >= 4) {
							L->pre_items[at++] = (u16)rv;
							left--;
							u32 n16 = 0;
							while (left >= 3) {
								u32 r = left > 6 ? 6 : left;
								L->pre_items[at++] = 16 | ((r - 3) << 5);
								n16++;
								left -= r;
							}
							atomicAdd((u32 *)&L->pre_freq[16], n16);
							atomicAdd((u32 *)&L->pre_freq[rv], 1u);
						}
						if (left)
							atomicAdd((u32 *)&L->pre_freq[rv], left);
						while (left) {
							L->pre_items[at++] = (u16)rv;
							left--;
						}
					}
					if (tid == 0)
						L->vars[V_NPRE] = ni;
				}
				__syncthreads();
				PROF_MARK(21);
				if (wave == 0)
					make_code(L->pre_freq, 19, 7, L->pre_lens, L->pre_codes,
						  L->sorted, (huff_scratch<32> *)L->hw,
						  0, false, lane);
				__syncthreads();
				PROF_MARK(22);
				/* exact costs (deflate_compress.c:1747-1808) */
				u32 dyn = 0, stat = 0;
				if (tid < 320) {
					u32 f = L->freq[tid];
					u32 xb = 0, sl = 8;
					if (tid < 288) {
						sl = tid < 144 ? 8 : tid < 256 ? 9 : tid < 280 ? 7 : 8;
						if (tid >= 265 && tid < 285)
							xb = (tid - 261) >> 2;
					} else {
						u32 ds = tid - 288;
						sl = 5;
						if (ds >= 4)
							xb = (ds >> 1) - 1;
					}
					dyn = f * (L->lens[tid] + xb);
					stat = f * (sl + xb);
				}
				if (tid < 19) {
					u32 xb = tid == 16 ? 2 : tid == 17 ? 3 : tid == 18 ? 7 : 0;
					dyn += L->pre_freq[tid] * (L->pre_lens[tid] + xb);
				}
				u32 dyn_tot, stat_tot;
				(void)block_scan(L, dyn, &dyn_tot);
				(void)block_scan(L, stat, &stat_tot);
				static const u8 perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10,
							     5, 11, 4, 12, 3, 13, 2, 14,
							     1, 15 };
				u32 nexp = 19;
				while (nexp > 4 && L->pre_lens[perm[nexp - 1]] == 0)
					nexp--;
				u32 cost_dyn = 3 + 5 + 5 + 4 + 3 * nexp + dyn_tot;
				u32 cost_stat = 3 + stat_tot;
				/* stored: align + (LEN,NLEN) per <= 65535 piece */
				u32 pieces = blen ? (blen + 65534) / 65535 : 1;
				u32 pad = (u32)((0 - (os.bits + 3)) & 7);
				u64 cost_stored = 3 + pad + 32 + 8ull * blen +
						  (u64)(pieces - 1) * 40;
				u64 best = cost_stored;
				btype = 0;
				if (cost_stat < best) {
					best = cost_stat;
					btype = 1;
				}
				if (cost_dyn < best) {
					best = cost_dyn;
					btype = 2;
				}
				if ((os.bits + best + 7) / 8 + ftr_bytes > os.avail)
					overflow = true;
				L->vars[V_TMP3] = nexp;
			} else {
				u32 pieces = blen ? (blen + 65534) / 65535 : 1;
				u64 cost = (u64)pieces * 40 + 8ull * blen;
				if ((os.bits + cost + 7) / 8 + ftr_bytes > os.avail)
					overflow = true;
			}
			if (overflow)
				break;

			PROF_MARK(7);
			/* ---- S6: emit ---- */
			stg_restore(L);
			if (btype == 0) {
				/* stored pieces: header by thread 0, bytes as 8-bit
				 * "codes" through the same staging path */
				u32 done = 0;
				do {
					u32 piece = blen - done > 65535 ? 65535 : blen - done;
					u32 fin = (is_final && done + piece == blen) ? 1 : 0;
					u32 pad = (u32)((0 - (os.bits + 3)) & 7);
					if (tid == 0) {
						stg_put(L, &os, os.bits, fin, 3);
						u64 b = os.bits + 3 + pad;
						stg_put(L, &os, b, piece | ((u64)(piece ^ 0xFFFF) << 16), 32);
					}
					os.bits += 3 + pad + 32;
					__syncthreads();
					for (u32 w0 = 0; w0 < piece; w0 += 2048) {
						u32 cnt = piece - w0 < 2048 ? piece - w0 : 2048;
						stg_flush(L, &os, false);
						__syncthreads();
						for (u32 j = tid; j < cnt; j += NT) {
							u32 pos = bstart + done + w0 + j;
							stg_put(L, &os, os.bits + 8ull * j, inp[pos], 8);
						}
						os.bits += 8ull * cnt;
						__syncthreads();
					}
					done += piece;
				} while (done < blen);
				stg_flush(L, &os, false);
			} else {
				if (btype == 1) {
					/* static codes: lens fixed, canonical codewords */
					__syncthreads();
					for (u32 s = tid; s < 320; s += NT)
						L->lens[s] = s < 144 ? 8 : s < 256 ? 9 :
							     s < 280 ? 7 : s < 288 ? 8 : 5;
					__syncthreads();
					if (tid == 0) {
						u32 nc[16] = { 0 }, bl[16] = { 0 };
						for (u32 s = 0; s < 288; s++)
							bl[L->lens[s]]++;
						u32 code = 0;
						for (u32 d = 1; d < 16; d++) {
							code = (code + bl[d - 1]) << 1;
							nc[d] = code;
						}
						for (u32 s = 0; s < 288; s++) {
							u32 l = L->lens[s];
							L->codes[s] = (u16)(__brev(nc[l]++) >> (32 - l));
						}
						for (u32 s = 0; s < 32; s++)
							L->codes[288 + s] = (u16)(__brev(s) >> 27);
					}
					__syncthreads();
				}
				/* block header: thread 0 the fixed fields, threads
				 * 1..nexp the precode lengths, then one thread per
				 * precode item; bit offsets by a workgroup scan */
				{
					u64 hcode = 0;
					u32 hbits = 0;
					const u32 nexp = btype == 2 ? L->vars[V_TMP3] : 0;
					const u32 ni = btype == 2 ? L->vars[V_NPRE] : 0;
					if (tid == 0) {
						hcode = is_final | (btype << 1);
						hbits = 3;
						if (btype == 2) {
							u32 nlit = L->vars[V_TMP1], noff = L->vars[V_TMP2];
							hcode |= (u64)((nlit - 257) | ((noff - 1) << 5) |
								       ((nexp - 4) << 10)) << 3;
							hbits = 17;
						}
					} else if (tid <= nexp) {
						static const u8 perm2[19] = { 16, 17, 18, 0, 8, 7,
							9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
						hcode = L->pre_lens[perm2[tid - 1]];
						hbits = 3;
					} else if (tid <= nexp + ni) {
						u32 it = L->pre_items[tid - nexp - 1];
						u32 sym = it & 31, ex = it >> 5;
						u32 l = L->pre_lens[sym];
						u32 xb = sym == 16 ? 2 : sym == 17 ? 3 :
							 sym == 18 ? 7 : 0;
						hcode = L->pre_codes[sym] | ((u64)ex << l);
						hbits = l + xb;
					}
					u32 htot;
					u32 hoff = block_scan(L, hbits, &htot);
					stg_put(L, &os, os.bits + hoff, hcode, hbits);
					os.bits += htot;
				}
				stg_flush(L, &os, false);

				PROF_MARK(9);
				/* tokens, EWIN positions at a time.  Four barriers per
				 * window: after the match scatter, in the bit-offset scan,
				 * and two in the staging flush.  KD[] (what starts at each
				 * position: 0 literal, len | dist << 16 match, ~0 covered)
				 * is re-initialised for the next window by the thread that
				 * just consumed the entry; the covered prefix and the match
				 * count travel in two alternating pairs of LDS words. */
				u32 *KD = L->M;
				u32 seq_lo = 0;
				for (u32 i = tid; i < EWIN; i += NT)
					KD[i] = 0;
				if (tid == 0) {
					L->vars[V_SPILL] = 0;
					L->vars[V_SPILL1] = 0;
					L->vars[V_SEQCNT] = 0;
					L->vars[V_SEQCNT1] = 0;
				}
				__syncthreads();
				u32 wpar = 0;
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				for (u32 w0 = bstart; w0 < bend; w0 += EWIN, wpar ^= 1) {
					u32 wend = w0 + EWIN < bend ? w0 + EWIN : bend;
					const u32 v_spill_next = wpar ? V_SPILL : V_SPILL1;
					const u32 v_cnt = wpar ? V_SEQCNT1 : V_SEQCNT;
					/* this thread's input bytes: from the LDS ring while it
					 * still holds them, else from HBM (a block may be longer
					 * than the ring); in flight during the scatter */
					u8 litb[(EWIN + NT - 1) / NT];
#pragma unroll
					for (u32 k = 0; k < (EWIN + NT - 1) / NT; k++) {
						u32 pos = w0 + tid * ((EWIN + NT - 1) / NT) + k;
						litb[k] = pos >= wend ? 0 :
							  pos + RING >= loaded + 32 ? L->in[pos & RMASK] : inp[pos];
					}
					/* matches that start in this window (the list is
					 * position-sorted and holds < NT of them per window) */
					{
						u32 cw = 0;
						for (u32 sidx = seq_lo + tid; ; sidx += NT) {
							bool mine = false;
							if (sidx < nseq) {
								u64 sq = seqg[sidx];
								u32 pos = bstart + (u32)sq;
								if (pos < wend) {
									u32 len = (u32)(sq >> 32) & 0x1FF;
									u32 q = pos - w0;
									mine = true;
									KD[q] = len | ((u32)(sq >> 41) << 16);
									for (u32 j = 1; j < len && q + j < EWIN; j++)
										KD[q + j] = 0xFFFFFFFFu;
									if (pos + len > w0 + EWIN)
										atomicMax((u32 *)&L->vars[v_spill_next],
											  pos + len - (w0 + EWIN));
								}
							}
							const u64 mm = __ballot(mine);
							cw += __builtin_popcountll(mm);
							if (mm != ~0ull)	/* the list is position-sorted */
								break;
						}
						if (lane == 0 && cw)
							atomicAdd((u32 *)&L->vars[v_cnt], cw);
					}
					__syncthreads();
					seq_lo += L->vars[v_cnt];
					const u32 spill_next = L->vars[v_spill_next];
					/* each thread: EPT consecutive positions */
					enum { EPT = (EWIN + NT - 1) / NT };
					u64 code[EPT];
					u32 nb[EPT];
#pragma unroll
					for (u32 k = 0; k < EPT; k++) {
						u32 q = tid * EPT + k;
						u32 pos = w0 + q;
						code[k] = 0;
						nb[k] = 0;
						u32 kd = KD[q];
						KD[q] = q < spill_next ? 0xFFFFFFFFu : 0;
						if (pos >= wend)
							continue;
						if (kd == 0xFFFFFFFFu)
							continue;
						if (kd == 0) {
							u32 b = litb[k];
							code[k] = L->codes[b];
							nb[k] = L->lens[b];
						} else {
							u32 len = kd & 0xFFFF, dist = kd >> 16;
							u32 sl, xb, xv, ds, dxb, dxv;
							length_code(len, &sl, &xb, &xv);
							dist_code(dist, &ds, &dxb, &dxv);
							u32 ll = L->lens[257 + sl];
							u32 dl = L->lens[288 + ds];
							u64 v = L->codes[257 + sl];
							u32 sh = ll;
							v |= (u64)xv << sh;
							sh += xb;
							v |= (u64)L->codes[288 + ds] << sh;
							sh += dl;
							v |= (u64)dxv << sh;
							sh += dxb;
							code[k] = v;
							nb[k] = sh;
						}
					}
					u32 tot, mine = 0;
#pragma unroll
					for (u32 k = 0; k < EPT; k++)
						mine += nb[k];
					u32 off = block_scan1(L, mine, &tot, &tog);
					/* everyone has read this window's words: clear them for
					 * the window after the next one */
					if (tid == 0) {
						L->vars[v_cnt] = 0;
						L->vars[wpar ? V_SPILL1 : V_SPILL] = 0;
					}
#pragma unroll
					for (u32 k = 0; k < EPT; k++) {
						stg_put(L, &os, os.bits + off, code[k], nb[k]);
						off += nb[k];
					}
					os.bits += tot;
					stg_flush(L, &os, false);
				}
				__syncthreads();
				/* end of block */
				if (tid == 0)
					stg_put(L, &os, os.bits, L->codes[256], L->lens[256]);
				os.bits += L->lens[256];
				__syncthreads();
			}

			/* keep the unfinished staging bytes across the next tiles
			 * (M is reused as tile scratch) */
			stg_save(L, &os);
			PROF_MARK(8);
			if (!stored_only) {
				/* recalculate_min_match_len (deflate_compress.c:2359-2378):
				 * literals used more than total/1024 times in this block */
				u32 lf = tid < 256 ? L->freq[tid] : 0, lit_total;
				(void)block_scan(L, lf, &lit_total);
				u32 usedf = (tid < 256 && lf > (lit_total >> 10)) ? 1 : 0, nused;
				(void)block_scan(L, usedf, &nused);
				if (tid == 0)
					L->vars[V_MINLEN] = choose_min_len(nused, depth);
				__syncthreads();
			}
			if (OPT && mode == 3 && tid < 256)
				bsave[tid] = 0;	/* the previous tile's bytes are added by the next one */
			if (retro) {
				/* the last tile's tokens open the next block: its matches
				 * move to the front of the list (positions are relative to
				 * the block start), its histogram becomes the block's */
				const u32 cnt = nseq_all - nseq;	/* <= SEQ_TILE_MAX < 2 NT */
				u64 e0 = 0, e1 = 0;
				if (tid < cnt)
					e0 = seqg[nseq + tid];
				if (tid + NT < cnt)
					e1 = seqg[nseq + NT + tid];
				__syncthreads();
				if (tid < cnt)
					seqg[tid] = e0 - blen;
				if (tid + NT < cnt)
					seqg[NT + tid] = e1 - blen;
				if (tid < 320)
					L->freq[tid] = fsave[320 + tid];
				if (tid == 0)
					L->vars[V_NSEQ] = cnt;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			} else {
				for (u32 i = tid; i < 320; i += NT)
					L->freq[i] = 0;
				if (tid == 0)
					L->vars[V_NSEQ] = 0;
			}
			/* restore what S1..S4 expect in M[0..3]: the deferred
			 * entries were consumed only if the walk passed them; the
			 * encode pass clobbered them, so re-derive from nothing:
			 * deferred positions are re-evaluated as "no match" */
			if (tid < 4)
				L->M[tid] = 0;
			block_start = bend;
			__syncthreads();
		}

		/* ---- finish the stream ---- */
		__syncthreads();
		if (!overflow && !seg_last &&
		    (os.bits + 3 + 7) / 8 + 4 > os.avail)
			overflow = true;
		if (!overflow) {
			stg_restore(L);
			if (!seg_last) {
				/* empty stored block: BFINAL 0, BTYPE 00, pad, LEN 0, NLEN ~0 */
				u64 fb = 8 * ((os.bits + 3 + 7) / 8);
				if (tid == 0)
					stg_put(L, &os, fb, 0xFFFF0000ull, 32);
				os.bits = fb + 32;
				__syncthreads();
			}
			if (ftr_bytes) {
				/* gzip_compress.c:73-79 / zlib_compress.c:66-72 */
				u32 sum = sums ? sums[c] : 0;
				u64 fb = 8 * ((os.bits + 7) / 8);
				if (tid == 0) {
					if (format == LDA_FMT_GZIP) {
						stg_put(L, &os, fb, sum, 32);
						stg_put(L, &os, fb + 32, n, 32);
					} else {
						stg_put(L, &os, fb, __builtin_bswap32(sum), 32);
					}
				}
				os.bits = fb + 8 * ftr_bytes;
				__syncthreads();
			}
			stg_flush(L, &os, true);
			if (tid == 0)
				out_nbytes[c] = (os.bits + 7) / 8;
		} else if (tid == 0) {
			out_nbytes[c] = 0;
		}
		__syncthreads();
	}
}

/* host helper: dynamic LDS size the kernel needs */
#define DEFLATE_KERNEL_PARAMS                                                  \
	u64 n_chunks, int format, int level, u32 depth, u32 nice, u32 mode,    \
	const u8 *__restrict__ in_base, const u64 *__restrict__ in_offsets,    \
	const u64 *__restrict__ in_nbytes, u8 *__restrict__ out_base,          \
	const u64 *__restrict__ out_offsets,                                   \
	const u64 *__restrict__ out_avail_arr, u64 *__restrict__ out_nbytes,   \
	const u32 *__restrict__ sums, u64 *__restrict__ seq_scratch,           \
	const u32 *__restrict__ seg_info, u32 *__restrict__ next_chunk
#define DEFLATE_KERNEL_ARGS                                                    \
	lds_raw, n_chunks, format, level, depth, nice, mode, in_base,          \
	in_offsets, in_nbytes, out_base, out_offsets, out_avail_arr,           \
	out_nbytes, sums, seq_scratch, seg_info, next_chunk

extern "C" __global__ void __launch_bounds__(NT)
lda_deflate_batch_kernel(DEFLATE_KERNEL_PARAMS)
{
	extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
	deflate_batch_body<false>(DEFLATE_KERNEL_ARGS);
}

/* levels 10-12 */
extern "C" __global__ void __launch_bounds__(NT)
lda_deflate_opt_kernel(DEFLATE_KERNEL_PARAMS)
{
	extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
	deflate_batch_body<true>(DEFLATE_KERNEL_ARGS);
}

extern "C" size_t lda_deflate_lds_bytes(void)
{
	return sizeof(struct deflate_lds);
}

extern "C" size_t lda_deflate_tile(void)
{
	return TILE;
}

extern "C" size_t lda_deflate_seq_words(void)
{
	return SEQ_STRIDE;
}

LDA_PROF_DEFINE_READER(libdeflate_amd_profile_read_deflate)
