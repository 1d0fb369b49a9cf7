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
 * and advances in TILES of 4096 positions (each stage is a function below;
 * deflate_batch_body() is the schedule: two tiles are in flight - the final
 * parse, the token emission and the split statistics of tile k run in the
 * same phase ("phase X") as the shallow search and the first parse of tile
 * k + 1 and the chain insertion of tile k + 2, handing over through LDS
 * counters and flags; see the comment at the tile loop and DESIGN.md 3.3):
 *
 *   S0  input up to the end of the tile after the next (it joins the chains
 *       while the next one is searched);
 *   S1+S2  chain insertion WITHOUT a sort (insert_tile): a masked LDS
 *       exchange per position; conflicting lanes of one LDS atomic are
 *       served in lane order, so 64 consecutive positions per instruction
 *       get what a serial insertion loop would have returned.  One wave, two
 *       tiles ahead of the parse, beside round A;
 *   S3  progressive search: round A - every position measures its two
 *       nearest chain members (round_a); then, after a first parse, round B -
 *       only the positions that parse visited are searched to the full depth
 *       (depth / nice length per level as lib/deflate_compress.c:3927-3979),
 *       in packed generations of growing quanta of chain steps
 *       (build_worklist, search_queue); the last generation - few items, long
 *       chains - runs inside phase X beside the next tile's round A
 *       (rb_batch).  Levels 10-12 search every position (search_items);
 *   S4  the greedy / lazy / lazy2 choice is a pure function of the
 *       per-position results (rules of deflate_compress.c:2573-2575,
 *       2712-2755): steps position-parallel (stage_steps), the path by one
 *       wave of 64 speculative segment walkers (parse_tile); the tokens go to
 *       a per-workgroup list in HBM, the symbol histogram is built in the
 *       same pass;
 *   S5  at block end (content-driven split, deflate_compress.c:2143-2256, or
 *       64 Ki positions): length-limited canonical Huffman codes, exact cost
 *       of dynamic / static / stored, header;
 *   S6  the tokens are encoded one per thread: a workgroup prefix sum of the
 *       bit lengths gives the bit offset, ds_or packs the bits into an LDS
 *       staging buffer that is written to HBM with coalesced stores.
 *
 * HBM traffic beyond input-once / output-once: the token list (4 B per token,
 * written in S4, read in S6) and the 3-byte-table candidates (2 B per
 * position, from the inserting wave to the next tile's round A).  The
 * compressed bytes differ from the reference's (libdeflate.h:76-83 leaves
 * them unpinned); validity, round trip, compress_bound and
 * ratio-vs-reference are what the tests check.
 *
 * deflate_small.hip compiles this file a second time (LDA_SMALL: 256 threads,
 * a 4 KiB ring, tiles of 2048) for batches of buffers of at most 4096 bytes.
 */
#include <stddef.h>
#include "device_common.h"
#include "kernels.h"


#ifndef NT
#define NT LDA_DEFLATE_THREADS
#endif
/* sections written for one element per thread of a 1024-thread workgroup run
 * VPT consecutive elements per thread in a smaller one */
#define VPT (1024 / NT)
#define NWAVES (NT / 64)
#ifndef TILE
#define TILE 4096
#endif
#ifndef RING
#define RING 32768u
#endif
#define RMASK (RING - 1)
#define LOOKAHEAD 272u
#ifndef HASH_BITS
#define HASH_BITS 13
#endif
#ifndef HASH3_BITS
#define HASH3_BITS 12
#endif
/*
 * Blocks end where the content changes (the observation classes of
 * lib/deflate_compress.c:2092-2218, evaluated per tile, see "block end?") or
 * once they reach MAX_BLOCK_LEN: the tokens of the current block live in HBM
 * (see TOK_MATCH below), so a block can be as long as a whole 64 KiB buffer,
 * like the reference's for homogeneous data.
 */
#define MAX_BLOCK_LEN 65536u
/*
 * The tokens of the current block live in HBM, one u32 each, in position
 * order (one list per workgroup): a literal is its byte; a match is
 * TOK_MATCH | (length - 3) | (distance - 1) << 8.  Nothing of a block has to
 * stay in LDS until the block is written, so a block can be as long as a
 * whole 64 KiB buffer, and the encode pass runs over tokens (all lanes busy),
 * not over positions (most of them inside a match).
 */
#define TOK_MATCH 0x80000000u
#define TOK_TILE_MAX (TILE + 8)		/* >= new tokens per tile */
#define TOK_CAP (MAX_BLOCK_LEN + 2 * TILE + 64)	/* u32 entries */
#define SEQ_GCAP (TOK_CAP / 2)			/* the same in u64 words */
#define SEQ_STRIDE (SEQ_GCAP + (TILE + 8) / 2 + 320 + 256 + (TILE + 8) / 2)	/* u64 words of HBM scratch per workgroup */
#define S3_WALK 8		/* chain steps per walk pass (a lane stalls while its 4-entry hit queue is full) */
#define S3_EVMIN 1u		/* lanes with a queued hit that trigger an evaluate round */
#define S3_TAIL 256u		/* positions at the end of a tile searched with reduced depth */
#define S3_CLAIM 24u		/* finished lanes that trigger a claim pass */
#define S3_RA_DEPTH 2u		/* chain members the shallow pass measures at every position */
#define S6_ALWAYS_FLUSH 0
#define S3_HALF_SHIFT 1		/* the look-ahead positions are searched to depth >> this (the reference: 1) */
#define S3_HALF 1		/* 0: the lazy rule's look-ahead positions are not searched deeper */
#define P1_PASSES 0xFFFFFFFFu	/* passes of the first (worklist) parse */
/* The last generation of round B runs inside phase X, beside the next tile's
 * shallow search (rb_last_gen() and the schedule), and round A measures its two
 * candidates side by side (match_length2_p()).  Neither in the small-buffer
 * kernel: its four waves have no eight to spare, and several workgroups share a
 * CU there and hide each other's LDS round trips already (the longer code
 * measured 12 % slower). */
#ifdef LDA_SMALL
#define RB_DEFER 0
#define RA_PAIR 0
#else
#define RB_DEFER 1
#define RA_PAIR 1
#endif
#define RB_TAIL_WAVES 8u	/* waves 1..8 take a deferred generation: at most 512 items */
static_assert(!RB_DEFER || RB_TAIL_WAVES + 3 <= NWAVES,
	      "wave 0 parses, the last two waves insert: the tail waves lie between them");
#define S3_ROUNDS 1u		/* deepening rounds (parse -> search what it visits) per tile */
#ifndef WQ_CAP
#define WQ_CAP 2048u		/* round B items per round; the rest waits for the next round */
#endif
#define WQ_SEG (WQ_CAP / NWAVES)
#define RA_PREF_TAIL NWAVES
#define ABSH(o, pos) (pos)
#define WQ_LIMIT WQ_CAP		/* items taken per round (<= WQ_CAP) */
/* nxtA / nxtB: two scratch arrays of NXT_ELEMS u16 (round B lists of WQ_CAP
 * u32 each, the bit staging area, block-end tables) */
#define NXT_ELEMS (2 * WQ_CAP + 8)
#define STG_WORDS (NXT_ELEMS / 2 - 8)	/* staging: STG_WORDS + 8 words = sizeof nxtA */

#define M_VALID 0x40000u

struct deflate_lds {
	u8 in[RING + 32];
	u16 prev[RING];
	u16 head[1u << HASH_BITS];
	u16 head3[(1u << HASH3_BITS) + 2];	/* last position per 3-byte hash (no chain); + a dummy slot */
	u32 M[TILE + 8];	/* tile scratch: best length | distance << 16 per position; block end: Huffman scratch */
	u64 dhalf[TILE / 64];	/* search depth class a position has had (DC_*), as two */
	u64 dfull[TILE / 64];	/* bitmaps: class >= DC_HALF, class == DC_FULL */
	u8 seen8[256];		/* distinct bytes of a tile (minimum match length) */
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
		u16 nxtB[NXT_ELEMS];	/* live only during the token choice */
	};
	u16 nxtA[NXT_ELEMS] __attribute__((aligned(16)));	/* S6: bit staging */
	u32 scan[2][NWAVES + 1];
	u32 carry[6];		/* staging bytes kept between blocks */
	u32 obs[1][10];		/* block-split observations of the block before this tile */
	u32 vars[24] __attribute__((aligned(16)));
	u64 pm[TILE / 64];	/* parse: token starts of the tile, one bit per position */
	u64 lit1[TILE / 64];	/* step of the position is 1 (a literal) */
	u64 lit2[TILE / 64];	/* step is 2 (two literals, lazy2 deferral) */
	u64 pmA[TILE / 64];	/* the same three for the FIRST parse of the next tile */
	u64 lit1A[TILE / 64];	/* (the one that finds what round B searches), which */
	u64 lit2A[TILE / 64];	/* runs beside the final parse of the current tile */
	u32 qn[4];		/* round B: item counts of three generations in rotation */
	u32 gbase[TILE / 64];	/* emit: first token-list index of each group of 64 positions */
	u32 rdy[TILE / 64 + 1];	/* round A: the iteration that last finished this group of nxt (+ 1) */
	u8 lslot[256];		/* length - 3 -> length slot (length_code()), built once per workgroup */
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

#ifdef LDA_SMALL
static_assert(sizeof(struct deflate_lds) <= 163840 / SMALL_WGS, "SMALL_WGS workgroups per CU");
static_assert(RING >= TILE && NXT_ELEMS * 2 >= 3600, "the whole buffer is resident; block-end tables fit nxtB");
#else
static_assert(sizeof(struct deflate_lds) <= 163840, "LDS of one CU");
static_assert(NXT_ELEMS == TILE + 8, "levels 10-12 keep one u16 per position in nxtA");
#endif
static_assert(offsetof(struct deflate_lds, in) == 0, "ld32/ld64 assume in[] at LDS offset 0");
#define PREV_OFF ((u32)offsetof(struct deflate_lds, prev))

enum {
	/* (the first four are what every tile reads after phase X: one 16-byte read) */
	V_NSEQ = 0, V_ENTRY, V_WALKPOS_LO, V_SPLIT, V_TMP0, V_TMP1, V_TMP2,
	V_TMP3, V_CTR, V_MINLEN, V_NPRE, V_NSEQ_PRE, V_WPOS_PRE, V_FIT, V_UNUSED0,
	V_CTR2, V_PFLAG, V_CTR3, V_STDONE, V_EMDONE, V_TAILDONE, V_CTR4, V_ST2DONE,
	V_COUNT
};
static_assert(V_COUNT <= sizeof(((struct deflate_lds *)0)->vars) / sizeof(u32), "vars[] holds them all");

/* depth classes of the progressive search (done[]): what a position has been
 * searched with so far */
#define DC_SHALLOW 0	/* the first, shallow pass over all positions */
#define DC_HALF 1	/* depth / 2: consulted as lazy look-ahead */
#define DC_FULL 2	/* full depth: a token starts here */

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
	return __builtin_amdgcn_alignbyte(LDS32(i + 4), LDS32(i), ABSH(o, pos));
}

static __device__ __forceinline__ u64 ld64(const AS3 u8 *ring, u32 pos)
{
	(void)ring;
	u32 o = pos & RMASK, i = o & ~3u;
	u32 a = LDS32(i), b = LDS32(i + 4), c = LDS32(i + 8);
	u32 lo = __builtin_amdgcn_alignbyte(b, a, ABSH(o, pos));
	u32 hi = __builtin_amdgcn_alignbyte(c, b, ABSH(o, pos));
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

/* bit `lane` of a wave-uniform 64-bit mask: the mask IS a lane predicate, one
 * v_cndmask instead of a 64-bit shift per lane */
static __device__ __forceinline__ bool lane_bit(u64 uniform_mask)
{
	return __builtin_amdgcn_inverse_ballot_w64(uniform_mask);
}

/* number of set bits of a wave-uniform mask below this lane */
static __device__ __forceinline__ u32 rank_below(u64 uniform_mask)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(uniform_mask >> 32),
					 __builtin_amdgcn_mbcnt_lo((u32)uniform_mask, 0));
}

/* workgroup exclusive scan of one value per thread; returns the exclusive
 * prefix and writes the total to *total.  Two barriers. */
static __device__ u32 block_scan(lds_t *L, u32 v, u32 *total)
{
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	u32 incl = wave_scan_incl(v);

	if (lane == 63)
		L->scan[0][wave] = incl;
	__syncthreads();
	/* the waves' sums one per lane, a wave scan over them (a loop over the
	 * sixteen words is six times the instructions, on every wave's path) */
	const u32 sw = lane < NWAVES ? L->scan[0][lane] : 0;
	const u32 iw = wave_scan_incl(sw);
	const u32 base = bcast_lane(iw - sw, wave);
	__syncthreads();
	*total = bcast_lane(iw, NWAVES - 1);
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
	const u32 sw = lane < NWAVES ? sc[lane] : 0;
	const u32 iw = wave_scan_incl(sw);
	*total = bcast_lane(iw, NWAVES - 1);
	return bcast_lane(iw - sw, wave) + incl - v;
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

/* buffers under 512 bytes: 3, as the reference; without the 3-byte table
 * (level 1) never under 4 */
static __device__ __forceinline__ u32
min_len_policy(u32 distinct, u32 nbytes, u32 depth, bool use3)
{
	const u32 ml = nbytes < 512 ? 3 : choose_min_len(distinct, depth);

	return use3 || ml > 4 ? ml : 4;
}

static __device__ u32
find_len3(const lds_t *L, u32 p, u32 cur, u32 c3_16, u32 dmax,
	  u32 dlim, u32 *best)
{
	if (dmax > dlim)
		dmax = dlim;
	if (p >= 8) {
		/* the eight nearest distances without a branch: the three bytes at
		 * p - d for d = 8..1 are byte windows of (p-8..p-5, p-4..p-1, cur);
		 * v_perm_b32 cuts each out with a zero byte on top (selector 0x0C),
		 * so a window is one instruction and its test one compare with the
		 * position's own three bytes; the nearest distance that matches wins
		 * (a chain of selects, farthest first) and is valid iff it is within
		 * dmax - every other match is farther */
		const u64 w8 = ld64(L->in, p - 8);
		const u32 A = (u32)w8, B = (u32)(w8 >> 32);
		const u32 cur3 = cur & 0xFFFFFFu;
		u32 bd = 0;
#define LEN3_TRY(d, hi, lo, sel) bd = __builtin_amdgcn_perm(hi, lo, sel) == cur3 ? (d) : bd
		LEN3_TRY(8, A, A, 0x0C020100u);
		LEN3_TRY(7, A, A, 0x0C030201u);
		LEN3_TRY(6, B, A, 0x0C040302u);
		LEN3_TRY(5, B, A, 0x0C050403u);
		LEN3_TRY(4, B, B, 0x0C020100u);
		LEN3_TRY(3, B, B, 0x0C030201u);
		LEN3_TRY(2, cur, B, 0x0C040302u);
		LEN3_TRY(1, cur, B, 0x0C050403u);
#undef LEN3_TRY
		if (bd && bd <= dmax) {
			*best = 3;
			return bd;
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
	/* without a branch: every lane evaluates both look-aheads and selects
	 * (as nested ifs this compiled into a dozen EXEC-mask sections per call -
	 * 72 scalar instructions and as many branches around 50 vector ones, on
	 * the two step passes every position goes through).  log2 of a distance:
	 * 31 - clz; a position without a match has length 0 and loses every
	 * comparison whatever its "distance" gives. */
	const u32 l0 = m0 & 0xFFFF, l1 = m1 & 0xFFFF, l2 = m2 & 0xFFFF;
	if (mode == 0)		/* greedy (wave-uniform): nothing to look ahead at */
		return l0 ? l0 : 1;
	const s32 z0 = (s32)__builtin_clz((m0 >> 16) | 1), z1 = (s32)__builtin_clz((m1 >> 16) | 1),
		  z2 = (s32)__builtin_clz((m2 >> 16) | 1);
	/* b0 - b1 = z1 - z0 */
	const bool lazy = mode >= 1 && l0 < nice;
	const bool c1 = lazy && l1 >= l0 && 4 * (s32)(l1 - l0) + (z1 - z0) > 2;
	const bool c2 = lazy && mode >= 2 && l2 >= l0 && 4 * (s32)(l2 - l0) + (z2 - z0) > 6;
	u32 st = c2 ? 2 : l0;
	st = c1 ? 1 : st;
	return l0 == 0 ? 1 : st;
}

#include "deflate_opt.h"
#include "deflate_huffman.h"

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

/* ---------------- chain construction without a sort ---------------- */

/*
 * Masked exchange of one u16 half of an LDS dword, returning the old dword
 * (ds_mskor_rtn_b32: MEM = (MEM & ~mask) | val).  Lanes of ONE instruction
 * that hit the same address are served in ascending lane order on gfx950
 * (measured: tools/hwtest_lds_order.hip, run by tests/test_hw_gpu.py), and a
 * wave's LDS instructions execute in issue order.  So when the lanes of a
 * wave are consecutive positions, "exchange my position into head[hash]"
 * returns to every lane what a serial insertion loop would have found there:
 * the previous position with its hash, whether that is in the same group of
 * 64 or an earlier one.  That replaces the per-group sort and the ordered
 * threading of the groups (lib/hc_matchfinder.h:360-399 is the serial loop).
 * The asm has no wait: the caller waits once for a batch (lds_wait8).
 */
static __device__ __forceinline__ u32 lds_mskor_rtn(u32 byteaddr, u32 mask, u32 val)
{
	u32 old;
	asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3"
		     : "=v"(old) : "v"(byteaddr), "v"(mask), "v"(val) : "memory");
	return old;
}

/* wait for the LDS results above; the operands tie the uses to the wait */
static __device__ __forceinline__ void lds_wait8(u32 *o)
{
	asm volatile("s_waitcnt lgkmcnt(0)"
		     : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]),
		       "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]) :: "memory");
}

#define HEAD_OFF ((u32)offsetof(struct deflate_lds, head))
#define HEAD3_OFF ((u32)offsetof(struct deflate_lds, head3))

/* hash word of a position in M[]: 4-byte hash | M_VALID | 3-byte hash << 19 | valid3 << 31 */
#define MH_H3_SHIFT 19
#define MH_V3 0x80000000u

/*
 * ONE wave inserts the positions [t, tend) into the hash chains (head[] /
 * prev[], masked exchange, see above), another one into the single-slot
 * 3-byte table (plain read + write: duplicates inside a group of 64 all get
 * the slot's content from before the group, which only hides length-3
 * candidates 9..63 bytes back; the distances 1..8 are probed in registers
 * anyway), leaving the 3-byte candidate of every position in c3[] (HBM
 * scratch: it is consumed a tile later).  Position order, 8 groups of 64 in
 * flight (16 measured slower).  The LDS pipeline is not what bounds it
 * (tools/hwtest_lds_atomic_rate.hip: with 16 waves issuing, a 64-lane
 * ds_mskor_rtn_b32 on random addresses completes every 11 cycles CU-wide, a
 * 32-bit exchange or a plain read / write every 6; one wave alone gets one
 * through every ~50 cycles, its own address arithmetic included): one wave
 * inserting a tile is 64 dependent batches of its own instruction stream,
 * ~22 K cycles.  Spreading the buckets over 16 waves by hash made every wave
 * run that whole stream (slower); moving parts of it beside the single-wave
 * parse phases made those 1.5x longer (the waves share the LDS queue) and the
 * kernel 5 % slower; preparing the whole next tile (insertion AND shallow
 * search) beside the first parse, with the results waiting in HBM, cost more
 * at the top of the tile than the overlap saved (13.3 vs 12.4 ms).  So the
 * insertion runs on one wave, one tile AHEAD, beside the shallow search of
 * the current tile, the phase with the most independent work per wave.
 */
static __device__ __forceinline__ void
insert_tile(lds_t *L, u32 t, u32 tend, u32 n, u32 lane)
{
	const u32 ngroups = (tend - t + 63) / 64;

	/* Nothing in a batch is conditional: the input words of all 8 groups are
	 * read first (any position is a valid ring address), then the 8
	 * exchanges go out back to back (a lane without a position exchanges
	 * nothing: mask 0), one wait, then the 8 prev[] entries - written for
	 * every lane of the tile's groups: a slot past the buffer's end is never
	 * read, and what it aliases in the ring lies before every window.  (With
	 * the loads inside per-group conditionals every group waited for its own
	 * LDS round trip: 340 cycles per group instead of 100.) */
	for (u32 g0 = 0; g0 < ngroups; g0 += 8) {
		u32 o[8], sh[8], wv[8];
#pragma unroll
		for (u32 k = 0; k < 8; k++)
			wv[k] = ld32(L->in, t + (g0 + k) * 64 + lane);
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			const u32 p = t + (g0 + k) * 64 + lane;
			const u32 h = hash4(wv[k]);
			const bool ok = g0 + k < ngroups && p + 4 <= n;
			sh[k] = (h & 1) << 4;
			o[k] = lds_mskor_rtn(HEAD_OFF + ((h >> 1) << 2),
					     ok ? 0xFFFFu << sh[k] : 0,
					     ok ? (p & 0xFFFF) << sh[k] : 0);
		}
		lds_wait8(o);
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			const u32 i = (g0 + k) * 64 + lane;
			L->prev[(t + i) & RMASK] = (u16)(o[k] >> sh[k]);
		}
	}
}

/* the 3-byte table's share of the insertion, on a wave of its own */
static __device__ __forceinline__ void
insert_tile3(lds_t *L, u16 *__restrict__ c3, u32 t, u32 tend, u32 n, u32 lane)
{
	const u32 ngroups = (tend - t + 63) / 64;
	AS3 u16 *const h3tab = (AS3 u16 *)L->head3;

	/* branch-free like insert_tile(): a lane without a position reads and
	 * rewrites a dummy slot behind the table */
	for (u32 g0 = 0; g0 < ngroups; g0 += 8) {
		u32 wv[8], v[8];
#pragma unroll
		for (u32 k = 0; k < 8; k++)
			wv[k] = ld32(L->in, t + (g0 + k) * 64 + lane);
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			const u32 p = t + (g0 + k) * 64 + lane;
			const bool ok = g0 + k < ngroups && p + 3 <= n;
			const u32 slot = ok ? hash3(wv[k]) : (1u << HASH3_BITS);
			/* (in program order: a later group sees this group's entries) */
			v[k] = h3tab[slot];
			h3tab[slot] = (u16)p;
			v[k] = ok ? v[k] : 0x8000;
		}
#pragma unroll
		for (u32 k = 0; k < 8; k++)
			c3[4 + (g0 + k) * 64 + lane] = (u16)v[k];
	}
}

/* a claim from an LDS counter without a lane-0 branch: EXEC is narrowed to
 * lane 0 around one returning add.  claim_issue() only sends it (the result
 * register is valid in lane 0 once the LDS queue has drained to it),
 * claim_get() waits and makes the value wave-uniform.  Both must run with all
 * lanes active.  (As `if (lane == 0) atomicAdd(...)` the compiler's atomic
 * optimizer wraps every such add in a dozen instructions of lane counting.) */
static __device__ __forceinline__ u32 claim_issue(u32 lds_byteaddr)
{
	u32 r;
	u64 save;
	asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1"
		     : "=&v"(r), "=&s"(save) : "v"(lds_byteaddr), "v"(1u) : "memory");
	return r;
}

static __device__ __forceinline__ u32 claim_get(u32 r)
{
	u32 g;
	asm volatile("s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, %1"
		     : "=s"(g) : "v"(r) : "memory");
	return g;
}

/* the same without a result: one add by lane 0 */
static __device__ __forceinline__ void lds_add_lane0(u32 lds_byteaddr, u32 v)
{
	u64 save;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
		     : "=&s"(save) : "v"(lds_byteaddr), "v"(v) : "memory");
}
#define VAR_ADDR(idx) ((u32)offsetof(struct deflate_lds, vars) + 4 * (idx))

/* ---------------- match measurement ---------------- */

/* raw aligned words of the ring (see ld32()): `i` is a multiple of 4 below RING,
 * `k` <= 28 bytes further lies inside the ring or its 32-byte mirror */
#define RAW(i, k) LDS32((i) + (k))
#define AB(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
/* the words of a stage are all in registers here: without it the compiler moves
 * a load that only one branch below uses into that branch, behind the wait for
 * the others (a second LDS round trip) */
#define HAVE4(a, b, c, d) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d))
static __device__ __forceinline__ u64 mk64(u32 lo, u32 hi)
{
	return ((u64)hi << 32) | lo;
}

/*
 * Length of the match of position p against cp (both absolute), for the
 * lanes with ev set; every lane of the wave must call it (matches longer
 * than 28 bytes are measured by the whole wave, 256 bytes per pass).  cur =
 * bytes p..p+3, nxt8 = bytes p+4..p+11.  0 when the first four bytes differ.
 * Every stage's raw words are requested before any of them is used: one LDS
 * round trip per stage.
 */
static __device__ __forceinline__ u32
match_length(const lds_t *L, bool ev, u32 p, u32 cp, u32 cur, u64 nxt8,
	       u32 maxlen, u32 lane)
{
	(void)L;
	const u32 op = p & RMASK, ip = op & ~3u, sp = op & 3;
	const u32 oc = cp & RMASK, ic = oc & ~3u, sc = oc & 3;
	const u32 b0 = RAW(ic, 0), b1 = RAW(ic, 4), b2 = RAW(ic, 8), b3 = RAW(ic, 12);
	const u64 x = nxt8 ^ mk64(AB(b2, b1, sc), AB(b3, b2, sc));
	u32 len = 4 + ((u32)__builtin_ctzll(x | (1ull << 63)) >> 3);
	ev = ev && AB(b1, b0, sc) == cur;
	bool more = ev && x == 0 && 12 < maxlen;
	if (__ballot(more)) {
		const u32 p3 = RAW(ip, 12), p4 = RAW(ip, 16), p5 = RAW(ip, 20);
		const u32 b4 = RAW(ic, 16), b5 = RAW(ic, 20);
		HAVE4(p3, p5, b4, b5);
		const u64 y = mk64(AB(p4, p3, sp), AB(p5, p4, sp)) ^ mk64(AB(b4, b3, sc), AB(b5, b4, sc));
		if (more) {
			len = 12 + ((u32)__builtin_ctzll(y | (1ull << 63)) >> 3);
			more = y == 0 && 20 < maxlen;
		}
		if (__ballot(more)) {
			const u32 p6 = RAW(ip, 24), p7 = RAW(ip, 28);
			const u32 b6 = RAW(ic, 24), b7 = RAW(ic, 28);
			HAVE4(p6, p7, b6, b7);
			const u64 z = mk64(AB(p6, p5, sp), AB(p7, p6, sp)) ^
				      mk64(AB(b6, b5, sc), AB(b7, b6, sc));
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
		u32 x4 = 1;
		if (off < bmax) {
			const u32 o1 = (bp + off) & RMASK, i1 = o1 & ~3u;
			const u32 o2 = (bc + off) & RMASK, i2 = o2 & ~3u;
			const u32 w0 = RAW(i1, 0), w1 = RAW(i1, 4), v0 = RAW(i2, 0), v1 = RAW(i2, 4);
			x4 = AB(w1, w0, o1 & 3) ^ AB(v1, v0, o2 & 3);
		}
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
	if (len > maxlen)
		len = maxlen;
	return ev ? len : 0;
}

/*
 * The same for TWO candidates of one position, the loads of each stage issued
 * together: half the LDS round trips of two calls (the shallow search measures
 * the two nearest chain members of every position, and a wave's time there is
 * the sum of its dependent LDS waits).  `hook` runs between the request of the
 * first stage's words and their use (round A requests the next group's first
 * loads there).  Same results as two match_length().
 */
template <class F> static __device__ __forceinline__ void
match_length2_p(const lds_t *L, bool ev1, bool ev2, u32 p, u32 cp1, u32 cp2, u32 cur,
		u64 nxt8, u32 maxlen, u32 lane, u32 *len1o, u32 *len2o, F &&hook)
{
	(void)L;
	const u32 op = p & RMASK, ip = op & ~3u, sp = op & 3;
	const u32 o1 = cp1 & RMASK, i1 = o1 & ~3u, s1 = o1 & 3;
	const u32 o2 = cp2 & RMASK, i2 = o2 & ~3u, s2 = o2 & 3;
	const u32 b0 = RAW(i1, 0), b1 = RAW(i1, 4), b2 = RAW(i1, 8), b3 = RAW(i1, 12);
	const u32 e0 = RAW(i2, 0), e1 = RAW(i2, 4), e2 = RAW(i2, 8), e3 = RAW(i2, 12);
	hook();
	const u64 x1 = nxt8 ^ mk64(AB(b2, b1, s1), AB(b3, b2, s1));
	const u64 x2 = nxt8 ^ mk64(AB(e2, e1, s2), AB(e3, e2, s2));
	u32 len1 = 4 + ((u32)__builtin_ctzll(x1 | (1ull << 63)) >> 3);
	u32 len2 = 4 + ((u32)__builtin_ctzll(x2 | (1ull << 63)) >> 3);
	ev1 = ev1 && AB(b1, b0, s1) == cur;
	ev2 = ev2 && AB(e1, e0, s2) == cur;
	bool m1 = ev1 && x1 == 0 && 12 < maxlen, m2 = ev2 && x2 == 0 && 12 < maxlen;
	if (__ballot(m1 || m2)) {
		const u32 p3 = RAW(ip, 12), p4 = RAW(ip, 16), p5 = RAW(ip, 20);
		const u32 b4 = RAW(i1, 16), b5 = RAW(i1, 20);
		const u32 e4 = RAW(i2, 16), e5 = RAW(i2, 20);
		HAVE4(p3, p5, b4, b5);
		HAVE4(p4, e4, e5, e5);
		const u64 pw = mk64(AB(p4, p3, sp), AB(p5, p4, sp));
		const u64 y1 = pw ^ mk64(AB(b4, b3, s1), AB(b5, b4, s1));
		const u64 y2 = pw ^ mk64(AB(e4, e3, s2), AB(e5, e4, s2));
		if (m1) {
			len1 = 12 + ((u32)__builtin_ctzll(y1 | (1ull << 63)) >> 3);
			m1 = y1 == 0 && 20 < maxlen;
		}
		if (m2) {
			len2 = 12 + ((u32)__builtin_ctzll(y2 | (1ull << 63)) >> 3);
			m2 = y2 == 0 && 20 < maxlen;
		}
		if (__ballot(m1 || m2)) {
			const u32 p6 = RAW(ip, 24), p7 = RAW(ip, 28);
			const u32 b6 = RAW(i1, 24), b7 = RAW(i1, 28);
			const u32 e6 = RAW(i2, 24), e7 = RAW(i2, 28);
			HAVE4(p6, p7, b6, b7);
			HAVE4(e6, e7, e7, e7);
			const u64 pz = mk64(AB(p6, p5, sp), AB(p7, p6, sp));
			const u64 z1 = pz ^ mk64(AB(b6, b5, s1), AB(b7, b6, s1));
			const u64 z2 = pz ^ mk64(AB(e6, e5, s2), AB(e7, e6, s2));
			if (m1) {
				len1 = 20 + ((u32)__builtin_ctzll(z1 | (1ull << 63)) >> 3);
				m1 = z1 == 0 && 28 < maxlen;
			}
			if (m2) {
				len2 = 20 + ((u32)__builtin_ctzll(z2 | (1ull << 63)) >> 3);
				m2 = z2 == 0 && 28 < maxlen;
			}
		}
	}
#pragma unroll
	for (u32 c = 0; c < 2; c++) {
		const u32 cp = c ? cp2 : cp1;
		for (u64 mm = __ballot(c ? m2 : m1); mm; mm &= mm - 1) {
			u32 src = (u32)__builtin_ctzll(mm);
			u32 bp = bcast_lane(p, src);
			u32 bc = bcast_lane(cp, src);
			u32 bmax = bcast_lane(maxlen, src);
			u32 off = 28 + 4 * lane;
			u32 x4 = 1;
			if (off < bmax) {
				const u32 oa = (bp + off) & RMASK, ia = oa & ~3u;
				const u32 ob = (bc + off) & RMASK, ib = ob & ~3u;
				const u32 w0 = RAW(ia, 0), w1 = RAW(ia, 4), v0 = RAW(ib, 0), v1 = RAW(ib, 4);
				x4 = AB(w1, w0, oa & 3) ^ AB(v1, v0, ob & 3);
			}
			u64 ne = __ballot(x4 != 0);
			u32 tot = bmax;	/* 28 + 256 >= 258 */
			if (ne) {
				u32 kk = (u32)__builtin_ctzll(ne);
				u32 xk = bcast_lane(x4, kk);
				u32 o = 28 + 4 * kk;
				if (o < bmax)
					tot = o + ((u32)__builtin_ctz(xk) >> 3);
			}
			if (lane == src) {
				if (c)
					len2 = tot;
				else
					len1 = tot;
			}
		}
	}
	if (len1 > maxlen)
		len1 = maxlen;
	if (len2 > maxlen)
		len2 = maxlen;
	*len1o = ev1 ? len1 : 0;
	*len2o = ev2 ? len2 : 0;
}

/*
 * Minimum match length from the distinct bytes of a tile's input
 * (calculate_min_match_len, lib/deflate_compress.c:2329-2353, which the
 * reference applies to the first 4096 bytes and then refreshes per block from
 * the literals used; with blocks as long as a buffer the per-tile estimate is
 * what follows content changes).  ONE wave: it runs beside the other waves'
 * work, a tile ahead.  t0 is a tile start (16-byte aligned in the ring).
 */
static __device__ __forceinline__ u32
wave_distinct_bytes(lds_t *L, u32 t0, u32 lim, u32 lane)
{
	AS3 u32 *const s32p = (AS3 u32 *)L->seen8;

	s32p[lane] = 0;
	wave_sync();
#pragma unroll
	for (u32 j = 0; j < TILE / 1024; j++) {
		const u32 off = (TILE / 64) * lane + 16 * j;
		const uint4 v = *(const AS3 uint4 *)&L->in[(t0 + off) & RMASK];
		const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (u32 i = 0; i < 16; i++)
			if (off + i < lim)
				L->seen8[(w[i >> 2] >> (8 * (i & 3))) & 0xFF] = 1;
	}
	wave_sync();
	return wave_sum((u32)__builtin_popcount(s32p[lane] & 0x01010101u));
}

/* stage the input bytes [loaded, want) into the ring (whole workgroup; the
 * first 32 bytes of the ring are mirrored past its end for ld32 / ld64) */
static __device__ __forceinline__ void
stage_input(lds_t *L, const u8 *__restrict__ inp, u32 loaded, u32 want,
	    bool aligned_in, u32 tid)
{
	if (aligned_in) {
		const u32 from = loaded & ~15u;
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
}

/* ---------------- progressive search ---------------- */

/*
 * The reference searches only where a token can start and where the lazy
 * rule looks ahead (lib/deflate_compress.c:2604-2808); the positions inside
 * its matches are merely inserted.  On text that is a third of the positions
 * and a seventh of the chain steps of "search everything".  The parse is
 * serial, so the positions it will visit are not known in advance; they are
 * found by iteration instead:
 *   round A   EVERY position gets a shallow search (the `ra_depth` nearest
 *             chain members, measured in full) + the length-3 probes;
 *   parse     the lazy / greedy rule over the results so far (parse_tile());
 *   round B   the token starts of that parse are searched to the full depth,
 *             the positions its lazy rule consulted to half of it
 *             (:2712-2721), each position at most once per class; then the
 *             parse is repeated, and so on until the parse visits nothing
 *             new (or a round limit).  At the fixed point the result IS a
 *             serial lazy parse with full-depth searches at what it visits.
 * Measured on the 64 KiB mix (CPU model of this scheme, tools/models/sim_mf.c):
 * 3.9 chain steps per position instead of 15.0 at level 6 on text, output
 * 0.03 % smaller than the serial parse after two rounds.
 */
/*
 * round_a() for the usual case (the two nearest chain members of every
 * position) with the LDS round trips off the dependent chain, see RA_PREF.
 * A group's first loads - the chain link of each position, its input words
 * p .. p + 15 and its 3-byte-table candidate - are in registers when its turn
 * comes: requested by the iteration before (RA_PREF >= 2: the next group is
 * claimed while this one's second chain link is on its way), or at the end of
 * the previous iteration for the last RA_PREF_TAIL groups of a tile, which
 * are not claimed ahead (a wave holding a group it has not started would
 * keep a free wave from taking it).
 */
struct ra_first {
	u32 c16, a0, a1, a2, a3, c3v;
};

static __device__ __forceinline__ struct ra_first
ra_first_loads(const u16 *__restrict__ c3, u32 t, u32 g, u32 lane, bool want3)
{
	struct ra_first f;
	const u32 i = 64 * g + lane, p = t + i;
	const u32 ip = (p & RMASK) & ~3u;

	/* (whole words, the 16-bit entry is cut out where it is used: a 16-bit
	 * value carried around the loop is masked where it is loaded, and the
	 * mask would wait for the load) */
	f.c16 = LDS32(PREV_OFF + ((2 * (p & RMASK)) & ~3u));
	f.a0 = RAW(ip, 0);
	f.a1 = RAW(ip, 4);
	f.a2 = RAW(ip, 8);
	f.a3 = RAW(ip, 12);
	/* (a group past the tile is only ever loaded for, never searched: the
	 * counter ran out while the claim was on its way.  The load itself is
	 * unconditional - of an entry inside the tile - so that the waits for
	 * this load and for the one before it can be told apart) */
	(void)want3;
	f.c3v = *(const u32 *)((const u8 *)c3 + ((2 * (4 + (g < TILE / 64 ? i : lane))) & ~3u));
	return f;
}

/* one group of 64 positions; PRE: the next group is claimed ahead (see above).
 * On return `g` / `f` are the next group and its first loads. */
template <bool PRE> static __device__ __forceinline__ void
ra_group(lds_t *L, AS3 u32 *Mo, const u16 *__restrict__ c3, u32 t, u32 tend, u32 n,
	 u32 lo_pos, u32 min_len, u32 done_class, u32 nice, u32 dlim3,
	 u32 tag, u32 lane, u32 &g, const struct ra_first &f, struct ra_first &fnext)
{
	const u32 ctr = (u32)offsetof(struct deflate_lds, vars) + 4 * V_CTR;
	const bool want3 = min_len <= 3;
	const u32 i = 64 * g + lane, p = t + i;
	const u32 sp = p & 3;
	const u32 cur = AB(f.a1, f.a0, sp);
	const u64 nxt8 = mk64(AB(f.a2, f.a1, sp), AB(f.a3, f.a2, sp));
	const u32 hsh = 16 * (lane & 1);	/* p and i have the lane's parity */
	const u32 c3v = want3 ? (f.c3v >> hsh) & 0xFFFF : 0;
	const bool act0 = p < tend && p + 4 <= n;
	const u32 maxlen = n - p < 258 ? n - p : 258;
	const u32 dmaxp = p - lo_pos;
	const u32 nic = nice < maxlen ? nice : maxlen;
	u32 best = 3, bestd = 0;

	const u32 d1 = (p - (f.c16 >> hsh)) & 0xFFFF;
	const bool act1 = act0 && d1 > 0 && d1 <= dmaxp && 3 < nic;
	const u32 cp1 = p - d1;
	const u32 c2 = LDS16(PREV_OFF + 2 * (cp1 & RMASK));
	/* the next group: claimed now, its loads requested below */
	u32 rn = 0, gn = 0;
	if (PRE)
		rn = claim_issue(ctr);
	const u32 d2 = (p - c2) & 0xFFFF;
	const bool ch2 = act1 && d2 > d1 && d2 <= dmaxp;
	const u32 cp2 = p - d2;
	if (PRE)
		gn = claim_get(rn);
	const u32 c3n = LDS16(PREV_OFF + 2 * (cp2 & RMASK));
	u32 len1, len2;
	match_length2_p(L, act1, ch2, p, cp1, cp2, cur, nxt8, maxlen, lane,
			&len1, &len2, [&]() {
		if (PRE)
			fnext = ra_first_loads(c3, t, gn, lane, want3);
	});
	if (len1 > best) {
		best = len1;
		bestd = d1;
	}
	const bool act2 = ch2 && best < nic;
	if (act2 && len2 > best) {
		best = len2;
		bestd = d2;
	}
	u32 m = best >= 4 && best >= min_len ? best | (bestd << 16) : 0;
	if (m == 0 && want3 && p < tend && p + 3 <= n) {
		u32 b3 = 0;
		u32 bd = find_len3(L, p, cur, c3v, dmaxp, dlim3, &b3);
		if (bd)
			m = 3 | (bd << 16);
	}
	Mo[4 + i] = m;
	/* a chain that ended inside the shallow pass has been searched in full */
	const u32 dn = (p - c3n) & 0xFFFF;
	const u32 dcl = act2 && dn > d2 && dn <= dmaxp && best < nic ?
			done_class : DC_FULL;
	L->dhalf[g] = __ballot(dcl >= DC_HALF);
	L->dfull[g] = __ballot(dcl == DC_FULL);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	*(volatile AS3 u32 *)&L->rdy[g] = tag;
	if (PRE) {
		g = gn;
	} else {
		g = claim_get(claim_issue(ctr));
		if (g < TILE / 64)
			fnext = ra_first_loads(c3, t, g, lane, want3);
	}
}

static __device__ __forceinline__ void
round_a_p(lds_t *L, AS3 u32 *Mo, const u16 *__restrict__ c3, u32 t, u32 tend, u32 n,
	  u32 lo_pos, u32 min_len, u32 done_class, u32 nice, u32 dlim3,
	  u32 tag, u32 tid)
{
	const u32 lane = tid & 63;
	const u32 ctr = (u32)offsetof(struct deflate_lds, vars) + 4 * V_CTR;
	u32 g = claim_get(claim_issue(ctr));
	struct ra_first f = { 0, 0, 0, 0, 0, 0 };

	if (g < TILE / 64)
		f = ra_first_loads(c3, t, g, lane, min_len <= 3);
	/* two groups per round, the two sets of first loads changing roles: a
	 * copy at the end of the body would have to wait for the loads */
	struct ra_first f2 = { 0, 0, 0, 0, 0, 0 };
#pragma unroll 1
	while (g + RA_PREF_TAIL < TILE / 64) {
		ra_group<true>(L, Mo, c3, t, tend, n, lo_pos, min_len, done_class, nice, dlim3,
			       tag, lane, g, f, f2);
		if (!(g + RA_PREF_TAIL < TILE / 64)) {
			f = f2;
			break;
		}
		ra_group<true>(L, Mo, c3, t, tend, n, lo_pos, min_len, done_class, nice, dlim3,
			       tag, lane, g, f2, f);
	}
#pragma unroll 1
	while (g < TILE / 64)
		ra_group<false>(L, Mo, c3, t, tend, n, lo_pos, min_len, done_class, nice, dlim3,
				tag, lane, g, f, f);
}

static __device__ __forceinline__ void
round_a(lds_t *L, AS3 u32 *Mo, const u16 *__restrict__ c3, u32 t, u32 tend, u32 n,
	u32 lo_pos, u32 min_len, u32 ra_depth, u32 done_class, u32 nice, u32 dlim3,
	u32 tag, u32 tid)
{
	const u32 lane = tid & 63;
	if (ra_depth == 2 && RA_PAIR) {
		round_a_p(L, Mo, c3, t, tend, n, lo_pos, min_len, done_class, nice, dlim3,
			  tag, tid);
		return;
	}

	/* groups of 64 positions are taken from a counter: the two waves that
	 * insert the next tile meanwhile join in when they are done */
#pragma unroll 1
	for (;;) {
		u32 g = 0;
		/* (ONE lane-0 section per round: see the note at the end of the body) */
		if (lane == 0)
			g = atomicAdd((u32 *)&L->vars[V_CTR], 1u);
		g = bcast_first(g);
		if (g >= TILE / 64)
			break;
		const u32 i = 64 * g + lane, p = t + i;
		const u32 c3v = min_len <= 3 ? c3[4 + i] : 0;
		bool act = p < tend && p + 4 <= n;
		const u32 cur = ld32(L->in, p);
		const u64 nxt8 = ld64(L->in, p + 4);
		const u32 maxlen = n - p < 258 ? n - p : 258;
		const u32 dmaxp = p - lo_pos;
		const u32 nic = nice < maxlen ? nice : maxlen;
		u32 c16 = LDS16(PREV_OFF + 2 * (p & RMASK));
		u32 best = 3, bestd = 0, dprev = 0;

		for (u32 s = 0; s < ra_depth; s++) {
			u32 d = (p - c16) & 0xFFFF;
			act = act && d > dprev && d <= dmaxp && best < nic;
			if (!__ballot(act))
				break;
			u32 cp = p - d;
			u32 len = match_length(L, act, p, cp, cur, nxt8, maxlen, lane);
			if (len > best) {
				best = len;
				bestd = d;
			}
			if (act) {
				c16 = LDS16(PREV_OFF + 2 * (cp & RMASK));
				dprev = d;
			}
		}
		u32 m = best >= 4 && best >= min_len ? best | (bestd << 16) : 0;
		if (m == 0 && min_len <= 3 && p < tend && p + 3 <= n) {
			u32 b3 = 0;
			u32 bd = find_len3(L, p, cur, c3v, dmaxp, dlim3, &b3);
			if (bd)
				m = 3 | (bd << 16);
		}
		Mo[4 + i] = m;
		/* a chain that ended inside the shallow pass has been searched in full */
		const u32 dn = (p - c16) & 0xFFFF;
		const u32 dcl = act && dn > dprev && dn <= dmaxp && best < nic ?
				done_class : DC_FULL;
		/* every lane stores the same words: a second `if (lane == 0)` at
		 * the end of this loop's body gets threaded into the one at its
		 * top by the compiler, lane 0 then runs a round ahead of the other
		 * lanes and the wave hangs in the counter's read-first-lane */
		L->dhalf[g] = __ballot(dcl >= DC_HALF);
		L->dfull[g] = __ballot(dcl == DC_FULL);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		/* the group's results are in place: whoever computes the steps of
		 * the next tile's first parse may take it (every lane stores) */
		*(volatile AS3 u32 *)&L->rdy[g] = tag;
	}
}

/*
 * The parse, part 1 (whole workgroup, position-parallel): the step of every
 * position of the tile, token_step(q): 1 = literal, 2 = two literals, else
 * the match length - kept as two bitmaps (step 1, step 2) per 64 positions;
 * the length of a match step is in M[].  The caller puts a barrier after it.
 */
static __device__ __forceinline__ void
stage_steps(lds_t *L, s32 limit, u32 mode, u32 nice, u32 tid)
{
	const u32 lane = tid & 63, wave = tid >> 6;

#pragma unroll
	for (u32 k = 0; k < TILE / NT; k++) {
		const u32 g = wave * (TILE / NT) + k, q = 64 * g + lane, idx = q + 4;
		u32 st = 1;
		if ((s32)q < limit)
			st = token_step(L->M[idx], L->M[idx + 1], L->M[idx + 2], mode, nice);
		const u64 b1 = __ballot(st == 1), b2 = __ballot(st == 2);
		if (lane == 0) {
			L->lit1[g] = b1;
			L->lit2[g] = b2;
		}
	}
}

/* the same for groups of 64 positions claimed from a counter (V_CTR3), into
 * the bitmaps of the next tile's first parse; the groups done are counted in
 * V_STDONE.  Ms = the tile's search results; a group is taken as soon as round
 * A has finished it and the next one (rdy[] == tag: the step of a group's last
 * positions reads the first entries of the next group), so the steps of most
 * groups are computed while the last groups of round A are still searched. */
static __device__ __forceinline__ void
stage_steps_claimed(lds_t *L, const AS3 u32 *Ms, s32 limit, u32 mode, u32 nice, u32 tag,
		    u32 lane)
{
	bool had = false;

#pragma unroll 1
	for (;;) {
		if (had)
			lds_add_lane0(VAR_ADDR(V_STDONE), 1u);
		const u32 g = claim_get(claim_issue(VAR_ADDR(V_CTR3)));
		if (g >= TILE / 64)
			break;
		had = true;
		/* round A may still be working on the tile's last groups: a group's
		 * steps need its own results and the first two of the next group */
		while (*(volatile AS3 u32 *)&L->rdy[g] != tag ||
		       (g + 1 < TILE / 64 && *(volatile AS3 u32 *)&L->rdy[g + 1] != tag))
			__builtin_amdgcn_s_sleep(2);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const u32 q = 64 * g + lane, idx = q + 4;
		u32 st = 1;
		if ((s32)q < limit)
			st = token_step(Ms[idx], Ms[idx + 1], Ms[idx + 2], mode, nice);
		/* (every lane stores the same words: no second lane-0 section) */
		L->lit1A[g] = __ballot(st == 1);
		L->lit2A[g] = __ballot(st == 2);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	}
}

/* the steps of the FINAL parse of the current tile by claimed groups (V_CTR4,
 * counted in V_ST2DONE): what stage_steps() does, for the waves that ran a
 * deferred generation of round B inside phase X (M[] is complete when they
 * all are through) */
static __device__ __forceinline__ void
stage_steps_cur_claimed(lds_t *L, s32 limit, u32 mode, u32 nice, u32 lane)
{
	bool had = false;

#pragma unroll 1
	for (;;) {
		if (had)
			lds_add_lane0(VAR_ADDR(V_ST2DONE), 1u);
		const u32 g = claim_get(claim_issue(VAR_ADDR(V_CTR4)));
		if (g >= TILE / 64)
			break;
		had = true;
		const u32 q = 64 * g + lane, idx = q + 4;
		u32 st = 1;
		if ((s32)q < limit)
			st = token_step(L->M[idx], L->M[idx + 1], L->M[idx + 2], mode, nice);
		/* (every lane stores the same words: no second lane-0 section) */
		L->lit1[g] = __ballot(st == 1);
		L->lit2[g] = __ballot(st == 2);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	}
}

/* wait until an LDS word written by other waves reaches a value */
static __device__ __forceinline__ void wait_lds_eq(lds_t *L, u32 idx, u32 val)
{
	while (*(volatile AS3 u32 *)&L->vars[idx] != val)
		__builtin_amdgcn_s_sleep(2);
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/*
 * Part 2, one wave: lane l walks the positions [64 l, 64 l + 64) by
 * p -> p + step(p) (a run of literals in one go, through the bitmap), first
 * from a guessed start (its first position), then - a parse started anywhere
 * falls in step with the true one within a few tokens - from where the path
 * of the lane before it really arrives, until it meets its own earlier
 * path; repeated until no lane's arrival point changes (lane 0 starts at the
 * true entry, so lane k is exact after k passes at the latest; measured 2-3
 * passes).  Result: pm[l] = token starts in lane l's 64 positions; returns
 * where the path leaves [0, limit).  entry >= 0.
 */
static __device__ __forceinline__ s32
parse_tile(const AS3 u32 *Ms, const AS3 u64 *lit1p, const AS3 u64 *lit2p, AS3 u64 *pmp,
	   u32 lane, s32 entry, s32 limit, u64 *mask_out, u32 max_pass = 0xFFFFFFFFu)
{
	const s32 seg_lo = 64 * (s32)lane;
	const s32 hi = seg_lo + 64 < limit ? seg_lo + 64 : limit;
	const u64 lit = lane < TILE / 64 ? lit1p[lane] : 0;
	const u64 two = lane < TILE / 64 ? lit2p[lane] : 0;
	s32 ein = lane == 0 ? entry : seg_lo;
	s32 q = ein, ex = ein;
	u64 mask = 0, nm = 0;
	bool walking = q < hi;

	/* ---- the first pass (every lane from its guess, nothing to meet) ----
	 * A stretch of the path: lands at `rel`, a run of literals, then the
	 * token at r2 = rel + run (or the segment ends inside the run: the
	 * stretch then ends with its last literal).  Its bits are rel .. r2 =
	 * 2 * (1 << r2) - (1 << rel); the stretches of a lane are disjoint, so
	 * all of them are 2 * ends - starts. */
	{
		const u64 nlit = ~lit;
		const s32 hirel = hi - seg_lo;	/* <= 64; positions hirel .. are not mine */
		const AS3 u16 *const ml = (const AS3 u16 *)(Ms + 4 + seg_lo);	/* the lane's match lengths */
		u64 starts = 0, ends = 0;
		/* (a lane is walking while rel < hirel; one without positions to
		 * walk starts at or past hirel) */
		s32 rel = q - seg_lo;
		while (__ballot(rel < hirel)) {
			if (rel < hirel) {
				const u64 x = nlit >> (u32)rel;	/* zeros come in: at most 64 - rel literals */
				const u32 left = (u32)(hirel - rel);	/* > 0 */
				u32 run = (u32)__builtin_ctzll(x | (1ull << 63));
				run = x ? run : 64;
				run = run < left ? run : left;
				const u32 r2 = (u32)rel + run;	/* <= 64 */
				const bool tok = run < left;	/* a token that is not a literal starts at r2 */
				const u32 last = tok ? r2 : r2 - 1;
				starts |= 1ull << (u32)rel;
				ends |= 1ull << last;
				/* (read past the segment where it ends with a literal: the
				 * entry exists, M[] has 8 more than a tile's) */
				const u32 len = ml[2 * r2];
				const u32 st = (u32)(two >> (r2 & 63)) & 1 ? 2 : len;
				rel = (s32)(tok ? r2 + st : r2);
			}
		}
		/* (a lane that had nothing to walk keeps ex = ein and an empty mask) */
		if (q < hi) {
			mask = 2 * ends - starts;
			ex = seg_lo + rel;
		}
		walking = false;
	}
	/* (after the first pass above no lane is walking: the loop goes straight
	 * to the exchange) */
	for (;;) {
		while (__ballot(walking)) {
			if (walking) {
				/* a run of literals, then one more token */
				const u32 rel = (u32)(q - seg_lo);
				/* (the shift brings in zeros, so only a segment of 64
				 * literals walked from its first position has no end) */
				const u64 inv = ~(lit >> rel);
				u32 run = inv ? (u32)__builtin_ctzll(inv) : 64;
				if (run > (u32)(hi - q))
					run = (u32)(hi - q);
				u64 add = run >= 64 ? ~0ull : (((1ull << run) - 1) << rel);
				q += (s32)run;
				if (q < hi) {
					const u32 r2 = (u32)(q - seg_lo);
					add |= 1ull << r2;
					q += (two >> r2) & 1 ? 2 : (s32)(Ms[4 + q] & 0xFFFF);
				}
				const u64 hit = add & mask;
				if (hit) {	/* met my earlier path */
					const u64 below = (hit & (0 - hit)) - 1;
					mask = nm | (add & below) | (mask & ~below);
					walking = false;
				} else {
					nm |= add;
					if (q >= hi) {
						mask = nm;
						ex = q;
						walking = false;
					}
				}
			}
		}
		/* where the path of the lane before me leaves its positions (the
		 * lanes past the last position have nothing to walk: the path's
		 * exit is read from the last lane that has) */
		s32 pe = (s32)__builtin_amdgcn_update_dpp((u32)ex, (u32)ex, 0x138, 0xF, 0xF, false);
		const s32 e = lane == 0 ? entry : pe;
		const bool redo = e != ein && seg_lo < limit;
		if (!__ballot(redo) || --max_pass == 0)
			break;
		nm = 0;
		if (redo) {
			ein = e;
			if (e >= hi) {	/* the path jumps over my positions */
				mask = 0;
				ex = e;
			} else {
				q = e;
				walking = true;
			}
		}
	}
	if (lane < TILE / 64)
		pmp[lane] = mask;
	*mask_out = mask;
	return (s32)bcast_lane((u32)ex, limit > 0 ? (u32)(limit - 1) >> 6 : 0);
}

/* the carried-in idx 2, 3 (the last two positions of the tile before, which
 * its parse deferred) can only be the entry itself */
static __device__ __forceinline__ u32
entry_skip(const AS3 u32 *Ms, s32 entry, u32 lim_idx, u32 mode, u32 nice)
{
	u32 e = (u32)(entry + 4);

	for (u32 pre = 0; pre < 2; pre++)
		if (e < 4 && e < lim_idx)
			e += token_step(Ms[e], Ms[e + 1], Ms[e + 2], mode, nice);
	return e;
}

/*
 * The final parse of a tile (ONE wave): the tokens that start in the deferred
 * positions idx 2, 3 are appended by lane 0, parse_tile() finds the path, and
 * a wave scan over the per-group token counts (one token per token start, two
 * where the step is "two literals") leaves in gbase[g] where the tokens of
 * group g go in the block's token list - after that the groups can be emitted
 * in any order by any wave (emit_groups()).
 */
static __device__ __forceinline__ void
parse_and_base(lds_t *L, u32 *__restrict__ tokg, u32 t, s32 limit, u32 mode,
	       u32 nice, u32 lane)
{
	const s32 entry = (s32)bcast_first(L->vars[V_ENTRY]);
	const u32 lim_idx = (u32)(limit + 4);
	const u32 seq0 = bcast_first(L->vars[V_NSEQ]);
	u32 e = (u32)(entry + 4), npre = 0;

	for (u32 pre = 0; pre < 2; pre++) {	/* idx 2, 3 */
		if (e < 4 && e < lim_idx) {
			const u32 mm = L->M[e];
			const u32 st = token_step(mm, L->M[e + 1], L->M[e + 2], mode, nice);
			const u32 l0 = mm & 0xFFFF;
			const bool ism = st == l0 && l0;
			const u32 pos = (u32)((s32)t + (s32)e - 4);
			if (lane == 0) {
				if (ism) {
					u32 sl, xb, xv;
					tokg[seq0 + npre] = TOK_MATCH | (l0 - 3) |
							    (((mm >> 16) - 1) << 8);
					length_code(l0, &sl, &xb, &xv);
					atomicAdd((u32 *)&L->freq[257 + sl], 1u);
					dist_code(mm >> 16, &sl, &xb, &xv);
					atomicAdd((u32 *)&L->freq[288 + sl], 1u);
				} else {
					const u32 b0 = L->in[pos & RMASK];
					tokg[seq0 + npre] = b0;
					atomicAdd((u32 *)&L->freq[b0], 1u);
					if (st == 2) {
						const u32 b1 = L->in[(pos + 1) & RMASK];
						tokg[seq0 + npre + 1] = b1;
						atomicAdd((u32 *)&L->freq[b1], 1u);
					}
				}
			}
			npre += !ism && st == 2 ? 2 : 1;
			e += st;
		}
	}
	u64 mask;
	const s32 px = parse_tile((const AS3 u32 *)L->M, (const AS3 u64 *)L->lit1,
				  (const AS3 u64 *)L->lit2, (AS3 u64 *)L->pm, lane,
				  (s32)e - 4, limit, &mask);
	/* (a tile shorter than 64 groups: the lanes past it hold an empty mask) */
	const u64 two = lane < TILE / 64 ? mask & L->lit2[lane] : 0;
	const u32 cnt = (u32)__builtin_popcountll(mask) + (u32)__builtin_popcountll(two);
	const u32 incl = wave_scan_incl(cnt);
	if (lane < TILE / 64)
		L->gbase[lane] = seq0 + npre + incl - cnt;
	if (lane == 63) {
		L->vars[V_NSEQ] = seq0 + npre + incl;
		/* where the path leaves the tile */
		L->vars[V_WALKPOS_LO] = (u32)((s32)t + px);
		L->vars[V_ENTRY] = (u32)(px - (s32)TILE);
	}
}

/*
 * Emit: the lanes on the path classify their token, count it for the block's
 * Huffman codes and append it to the block's token list in position order
 * (ballot ranks inside the group, gbase[] across groups).  Groups of 64
 * positions are claimed from a counter (V_CTR2, zero on entry) by whatever
 * waves call this.
 */
static __device__ __forceinline__ void
emit_groups(lds_t *L, u32 *__restrict__ tokg, u32 t, u32 lane)
{
	const u64 lt = (1ull << lane) - 1;
	bool had = false;

#pragma unroll 1
	for (;;) {
		u32 g = 0;
		/* (the group finished in the round before is counted here, in the
		 * loop's one lane-0 section: see round_a(); its histogram atomics
		 * are LDS operations of this wave and complete after the wait) */
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (had)
			lds_add_lane0(VAR_ADDR(V_EMDONE), 1u);
		g = claim_get(claim_issue(VAR_ADDR(V_CTR2)));
		if (g >= TILE / 64)
			break;
		had = true;
		const u64 pmk = L->pm[g];
		if (!pmk)
			continue;
		const u64 two = pmk & L->lit2[g];
		if ((pmk >> lane) & 1) {
			/* One path for both kinds of token (as two branches the wave ran
			 * both, one after the other, with the mask juggling in between):
			 * the first list entry and histogram symbol are the match's
			 * (entry, length slot from the table) or the literal's, the
			 * second symbol is the match's offset slot or the second literal
			 * of a "two literals" step; both literal bytes are read whatever
			 * the token is (the ring is mirrored past its end). */
			const u32 q = 64 * g + lane, idx = q + 4;
			const u32 m0 = L->M[idx];
			const u32 l0 = m0 & 0xFFFF, dist = m0 >> 16;
			const u32 st = (L->lit1[g] >> lane) & 1 ? 1 : (two >> lane) & 1 ? 2 : l0;
			const u32 o = (t + q) & RMASK;
			const u32 b0 = L->in[o], b1 = L->in[o + 1];
			const u32 lsl = L->lslot[(l0 - 3) & 255];
			const u32 at = L->gbase[g] + (u32)__builtin_popcountll(pmk & lt) +
				       (u32)__builtin_popcountll(two & lt);
			const bool ism = st == l0 && l0;
			u32 dsl, xb, xv;
			dist_code(dist, &dsl, &xb, &xv);
			tokg[at] = ism ? TOK_MATCH | (l0 - 3) | ((dist - 1) << 8) : b0;
			atomicAdd((u32 *)&L->freq[ism ? 257 + lsl : b0], 1u);
			if (ism || st == 2)
				atomicAdd((u32 *)&L->freq[ism ? 288 + dsl : b1], 1u);
			if (!ism && st == 2)
				tokg[at + 1] = b1;
		}
	}
	/* the token list is read back by other waves at the end of the block */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

/*
 * What the parse in pm[] visited and has not been searched deeply enough:
 * token starts -> DC_FULL, the positions the lazy rule looked at -> DC_HALF
 * (lazy2's second look-ahead as well: a superset of what the rule reads).
 * Positions whose match already reaches the nice length are left alone.
 * Builds the worklist W[] (position | class << 12; search_queue() has the
 * layout), returns its length (at most WQ_CAP: what does not fit is asked
 * for again by the next parse).
 * Whole workgroup, two barriers.
 */
static __device__ __forceinline__ u32
build_worklist(lds_t *L, AS3 u32 *W, const AS3 u32 *src, u32 mode, u32 nice, u32 limit, u32 tid)
{
	const u32 lane = tid & 63, wave = tid >> 6;
	/* `src`: the tile's shallow results still wait in MX - every thread moves
	 * the entries of its own positions to M[] while it looks at them (a copy
	 * pass of its own in front of this function cost a barrier and a second
	 * read of every entry); what it needs of its neighbours' it reads from
	 * MX, which nothing overwrites before this function's first barrier */
	const AS3 u32 *Ms = src ? src : (const AS3 u32 *)L->M;

	/* parse-based: the positions the lazy rule looked at are the one (lazy2:
	 * two) after a token start that holds a match shorter than the nice
	 * length.  A wave owns four consecutive groups of 64, so the bits that
	 * spill into the next group travel in a register; only the spill from
	 * the group before the wave's first is recomputed from that group's last
	 * two positions. */
	u64 spill = 0;
	if (mode >= 1 && S3_HALF && wave) {
		const u32 g0 = wave * (TILE / NT), q1 = 64 * g0 - 1;
		const u64 pmask = L->pmA[g0 - 1];
		const u32 l1 = Ms[4 + q1] & 0xFFFF, l2 = Ms[4 + q1 - 1] & 0xFFFF;
		if ((pmask >> 63) && l1 >= 3 && l1 < nice)
			spill = mode >= 2 ? 3 : 1;
		if (mode >= 2 && ((pmask >> 62) & 1) && l2 >= 3 && l2 < nice)
			spill |= 1;
	}
	/* ranks by a scan, not by atomics: which items fall under the limit must
	 * not depend on timing.  (One round B per tile - S3_ROUNDS is 1 - so the
	 * classes of the items taken need not be recorded: the next shallow
	 * search rewrites dhalf[] / dfull[].)  The phase is instruction bound -
	 * 16 waves, ~100 instructions per group of 64 positions - so the masks
	 * stay wave-uniform words and every per-lane test is one v_cndmask. */
	static_assert(S3_ROUNDS <= 1, "build_worklist() does not record the classes it hands out");
	u32 item[TILE / NT];
	u64 bal[TILE / NT];
	u64 tm_[TILE / NT], dh_[TILE / NT], df_[TILE / NT];
	u32 l0_[TILE / NT];
	u32 cw = 0;
#pragma unroll
	for (u32 k = 0; k < TILE / NT; k++) {
		const u32 g = wave * (TILE / NT) + k, q = 64 * g + lane;
		tm_[k] = L->pmA[g];
		dh_[k] = L->dhalf[g];
		df_[k] = L->dfull[g];
		const u32 mq = Ms[4 + q];
		if (src)
			L->M[4 + q] = mq;
		l0_[k] = mq & 0xFFFF;
	}
#pragma unroll
	for (u32 k = 0; k < TILE / NT; k++) {
		const u32 g = wave * (TILE / NT) + k, q = 64 * g + lane;
		const u64 tmask = bcast64(tm_[k]);
		const bool below = l0_[k] < nice;
		const u64 b = mode >= 1 && S3_HALF ?
			__ballot(lane_bit(tmask) && l0_[k] >= 3 && below) : 0;
		const u64 hmask = (b << 1) | (mode >= 2 ? b << 2 : 0) | spill;
		spill = (b >> 63) | (mode >= 2 ? b >> 62 : 0);
		/* wanted: full depth where a token starts and the position has not
		 * had it; half depth where the lazy rule looked and the position
		 * has had only the shallow pass */
		const u64 wantf = tmask & ~bcast64(df_[k]);
		const u64 wanth = hmask & ~tmask & ~bcast64(dh_[k]);
		const bool isf = lane_bit(wantf);
		bal[k] = __ballot((isf || lane_bit(wanth)) && below);
		item[k] = q | ((isf ? DC_FULL : DC_HALF) << 12);
		cw += (u32)__builtin_popcountll(bal[k]);
	}
	if (lane == 0)
		L->scan[0][wave] = cw;
	__syncthreads();
	/* the waves' counts, one per lane, and a wave scan over them (a loop over
	 * the sixteen words was a third of this function's instructions) */
	const u32 cnt_w = lane < NWAVES ? L->scan[0][lane] : 0;
	const u32 incl_w = wave_scan_incl(cnt_w);
	u32 base = bcast_lane(incl_w - cnt_w, wave);
	const u32 wc = bcast_lane(incl_w, NWAVES - 1);
#pragma unroll
	for (u32 k = 0; k < TILE / NT; k++) {
		const u32 j = base + rank_below(bal[k]);
		if (lane_bit(bal[k]) && j < limit)
			W[j] = item[k];
		base += (u32)__builtin_popcountll(bal[k]);
	}
	if (tid == 0)
		L->qn[0] = 0;
	__syncthreads();
	return bcast_first(wc < limit ? wc : limit);
}

/*
 * Deep search of a list of positions (round B; levels 10-12: all positions
 * of the tile).  Chain lengths differ wildly between positions, so lanes do
 * not own fixed items: a lane claims the next unsearched item from a
 * workgroup counter when it finishes one.  The kernel is instruction-issue
 * bound, so the search is split into two kinds of wave-uniform passes:
 *   walk      S3_WALK chain steps per lane, nothing but the link chase and a
 *             4-byte compare at the offset where a longer match must differ
 *             (the zlib scan_end idea); hits are queued (<= 4 distances
 *             packed in a u64);
 *   evaluate  every lane pops its oldest (closest) hit and measures it
 *             (match_length()).
 * A position that was searched before (round A) starts from the match it
 * already has, so the filter rejects whatever cannot beat it; its M[] entry
 * is only rewritten when a longer match turns up.
 * W == NULL: the items are the positions 0..cnt-1 themselves, searched from
 * scratch at full depth (the last S3_TAIL positions of a tile - claimed when
 * the other waves are about to run dry - at a reduced depth, chosen by
 * position so that the output does not depend on timing).
 * Whole workgroup; V_CTR must be 0 on entry (barrier in between).
 */
static __device__ __forceinline__ void
search_items(lds_t *L, AS3 u32 *Mo, u32 t, u32 n, u32 lo_pos, u32 min_len, u32 depth,
	     u32 nice, const u16 *W, u32 cnt_items, u32 nwaves, u32 tid)
{
	const u32 lane = tid & 63;
	/* a short list is searched by few waves: a lane must get several items,
	 * or every wave ends up waiting for the deepest of its 64 first ones */
	if ((tid >> 6) >= nwaves)
		return;
	const u32 drain = depth < 8 ? depth : depth >> 3 > 8 ? depth >> 3 : 8;
	const u32 half = depth >> S3_HALF_SHIFT ? depth >> S3_HALF_SHIFT : 1;
	u32 my_i = 0xFFFFFFFFu, p = 0, cur = 0, c16 = 0, dmaxp = 0,
	    maxlen = 0, dep = 0, best = 3, bestd = 0, dprev = 0,
	    cnt = 0, boff = 0, curb = 0, best0 = 3;
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
			if (fin) {
				if (my_i < TILE && (!W || best > best0))
					Mo[4 + my_i] = best >= 4 && best >= min_len ?
						(best | (bestd << 16)) : 0;
				fin = false;
				my_i = 0xFFFFFFFFu;
				u32 it = cbase + __builtin_popcountll(fm & ((1ull << lane) - 1));
				if (it < cnt_items) {
					u32 item = W ? W[it] : it;
					my_i = item & 0xFFF;
					p = t + my_i;
					if (p + 4 <= n) {
						cur = ld32(L->in, p);
						nxt8 = ld64(L->in, p + 4);
						c16 = LDS16(PREV_OFF + 2 * (p & RMASK));
						maxlen = n - p < 258 ? n - p : 258;
						dmaxp = p - lo_pos;
						best = 3;
						bestd = 0;
						if (W) {
							dep = (item >> 12) == DC_FULL ? depth : half;
							u32 m = Mo[4 + my_i], l0 = m & 0xFFFF;
							if (l0 >= 4) {
								best = l0;
								bestd = m >> 16;
							}
						} else {
							dep = my_i >= TILE - S3_TAIL ? drain : depth;
						}
						best0 = best;
						boff = best - 3;
						curb = boff ? ld32(L->in, p + boff) : cur;
						dprev = 0;
						cnt = 0;
						ended = false;
						have = true;
						if (best >= maxlen) {	/* nothing longer exists */
							have = false;
							fin = true;
						}
					} else {
						fin = true;	/* no 4-byte match can start here */
						best = best0 = 3;
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
		/* evaluate: a round when enough lanes hold a hit, or when no lane
		 * can walk any further */
		bool done = false;
		u32 nev = __builtin_popcountll(__ballot(cnt > 0));
		u32 nwalk = __builtin_popcountll(__ballot(have && !ended && cnt < 4));
		while (nev && (nev >= S3_EVMIN || !nwalk)) {
			bool ev = cnt > 0;
			u32 d = (u32)(q >> (16 * ((cnt - 1) & 3))) & 0xFFFF;
			cnt -= ev ? 1 : 0;
			u32 len = match_length(L, ev, p, p - d, cur, nxt8, maxlen, lane);
			if (ev && len > best) {
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

/*
 * Round B: deep search of the worklist.  Chain depths differ wildly between
 * positions (1 to `depth` steps), and a wave pays for its deepest lane, so
 * the items are not walked to the end by the lane that takes them: the search
 * proceeds in GENERATIONS of a few walk passes (8 chain steps each) per item.
 * An item that is not finished after its quantum goes to the next
 * generation's list with the place it reached (the last chain member
 * visited, 16 bits), and the next generation takes the survivors packed 64 to
 * a wave again: every walk pass runs with (nearly) all lanes on useful steps.
 * Two lists alternate (WA, WB); one barrier per generation.
 * List entry: tile position | depth class << 12 | last visited << 16.
 * A position starts from the match it already has (M[]), so the filter
 * (4 bytes at the offset where a longer match must differ) rejects whatever
 * cannot beat it; M[] is rewritten only when a longer match turns up.
 * Whole workgroup; ends with a barrier.
 */
#ifdef LDA_PROFILE
#define PROF_GEN(g) do { if (threadIdx.x == 0) { unsigned long long n_ = __builtin_readcyclecounter(); \
	atomicAdd(&lda_prof[35 + ((g) < 3 ? (g) : 3)], n_ - pg_); pg_ = n_; } } while (0)
#else
#define PROF_GEN(g) do { } while (0)
#endif
/* the search depth round B works with: see the comment inside */
static __device__ __forceinline__ u32 rb_trim_depth(u32 depth)
{
	const u32 npass = depth >= 256 ? 4 : depth >= 32 ? 2 : 1;	/* walk passes per generation */
	const u32 quantum = 8 * npass;
	/* a depth a few steps past the end of a generation (level 6: 35 = 16 +
	 * 16 + 3, level 7: 100 = 16 + 32 + 48 + 4) would cost a whole round of
	 * batches - or a third pass for every batch of the last generation - for
	 * those few steps: up to an eighth of the depth is given up instead */
	for (u32 g = 1, end = quantum; end < depth; g++, end += quantum * g)
		if (depth - end <= depth / 8)
			depth = end;
	/* and the same for a pass of 8 steps (level 6: 35 -> 32, two passes in
	 * the second generation instead of three) */
	if (depth >= 16 && (depth & 7) <= depth / 8)
		depth &= ~7u;
	return depth;
}

/* generation `gen`: chain steps already walked, walk passes, steps of this one */
static __device__ __forceinline__ void
rb_gen_params(u32 depth, u32 gen, u32 *before, u32 *npass_g)
{
	const u32 npass = depth >= 256 ? 4 : depth >= 32 ? 2 : 1;
	const u32 quantum = 8 * npass;
	/* the survivors of a generation are the deep chains: later
	 * generations walk longer before they repack */
	*npass_g = npass * (gen + 1);
	*before = quantum * (gen * (gen + 1) / 2);
}

/*
 * One batch of a generation: lane's item `e` (see search_queue() for the
 * layout; `have` = the lane has one).  `depth` is the trimmed one.  Returns
 * whether the item goes on to the next generation, *next = its entry there.
 */
static __device__ __forceinline__ bool
rb_batch(lds_t *L, u32 t, u32 n, u32 lo_pos, u32 min_len, u32 depth, u32 nice,
	 u32 e, bool have, u32 gen, u32 lane, u32 *next)
{
	const u32 half = depth >> S3_HALF_SHIFT ? depth >> S3_HALF_SHIFT : 1;
	u32 before, npass_g;
	rb_gen_params(depth, gen, &before, &npass_g);
	const u32 qg = 8 * npass_g;
	const u32 i = e & 0xFFF, p = t + i;
	const u32 cdepth = ((e >> 12) & 3) == DC_FULL ? depth : half;
	u32 dep = before < cdepth ? cdepth - before : 0;
	dep = dep < qg ? dep : qg;
	/* (the chain link and the match the position has are requested first:
	 * they head the dependent loads) */
	u32 dprev = gen ? (p - (e >> 16)) & 0xFFFF : 0;
	u32 c16 = LDS16(PREV_OFF + 2 * ((p - dprev) & RMASK));
	const u32 m = L->M[4 + i], l0 = m & 0xFFFF;
	const u32 cur = ld32(L->in, p);
	const u64 nxt8 = ld64(L->in, p + 4);
	const u32 maxlen = n - p < 258 ? n - p : 258;
	const u32 dmaxp = p - lo_pos;
	const u32 nic = nice < maxlen ? nice : maxlen;
	u32 best = l0 >= 4 ? l0 : 3, bestd = l0 >= 4 ? m >> 16 : 0;
	const u32 best0 = best;
	u32 boff = best - 3;
	u32 curb = ld32(L->in, p + boff);
	bool act = have && p + 4 <= n && best < nic && dep;
	u64 endm = 0;	/* lanes whose chain has ended (wave-uniform mask) */
	for (u32 ps = 0; ps < npass_g; ps++) {
		u64 h1 = 0, h2 = 0;	/* lanes with >= 1 / 2 queued hits */
		u32 qh = 0, nok = 0;
		const u32 dep0 = dep;
		const u64 actm = __ballot(act);
#pragma unroll
		for (int s = 0; s < 8; s++) {
			/* (16-bit instructions of gfx9 clear the upper half of their
			 * result: one instruction each for the distance and for the
			 * chain entry's byte offset, 2 x (c16 mod 32768)) */
			u32 d, c16x2;
			asm("v_sub_u16 %0, %1, %2" : "=v"(d) : "v"(p), "v"(c16));
			if (RING == 32768u)
				asm("v_lshlrev_b16 %0, 1, %1" : "=v"(c16x2) : "v"(c16));
			else
				c16x2 = 2 * (c16 & RMASK);
			/* (a lane that stops - a full queue, the end of its chain or of
			 * its budget - stays stopped for the rest of the pass, so a lane
			 * that reaches step s has walked s steps of it; one ballot per
			 * compare: the masks are combined on the scalar unit) */
			const u64 chainm = __ballot(dep0 > (u32)s) & __ballot(d > dprev) &
					   __ballot(d <= dmaxp);
			const u64 stallm = endm | h2;
			const u64 okm = actm & ~stallm & chainm;
			const u32 w = ld32(L->in, c16 + boff);
			const u32 c16n = LDS16(PREV_OFF + c16x2);
			const u64 hitm = okm & __ballot(w == curb);
			const bool ok = lane_bit(okm);
			c16 = ok ? c16n : c16;
			dprev = ok ? d : dprev;
			nok = ok ? (u32)s + 1 : nok;
			endm |= actm & ~stallm & ~chainm;
			qh = lane_bit(hitm) ? ((qh << 16) | d) : qh;
			h2 |= h1 & hitm;
			h1 |= hitm;
			PROF_COUNT(20, __builtin_popcountll(okm));
			PROF_COUNT(17, __builtin_popcountll(hitm));
		}
		PROF_COUNT(13, __builtin_popcountll(__ballot(have)));
		PROF_COUNT(21, (h1 != 0) + (h2 != 0));
		dep = dep0 - nok;
		/* evaluate: every lane pops its oldest (closest) hit */
		if (h1) {
			const bool ev = lane_bit(h1);
			const u32 d = lane_bit(h2) ? qh >> 16 : qh & 0xFFFF;
			const u32 len = match_length(L, ev, p, p - d, cur, nxt8, maxlen, lane);
			if (ev && len > best) {
				best = len;
				bestd = d;
			}
		}
		if (h2) {
			const bool ev = lane_bit(h2);
			const u32 d = qh & 0xFFFF;
			const u32 len = match_length(L, ev, p, p - d, cur, nxt8, maxlen, lane);
			if (ev && len > best) {
				best = len;
				bestd = d;
			}
		}
		if (best >= nic)
			act = false;
		if (npass_g > 1) {
			if (!__ballot(act && !lane_bit(endm) && dep))
				break;
			boff = best - 3;
			curb = ld32(L->in, p + boff);
		}
	}
	const bool ended = lane_bit(endm);
	if (best > best0)
		L->M[4 + i] = best >= min_len ? best | (bestd << 16) : 0;
	*next = (e & 0x3FFF) | (((p - dprev) & 0xFFFF) << 16);
	return act && !ended && before + qg < cdepth;
}

/*
 * `defer`: a generation that no item can outlive (the search depth ends with
 * it) and that fits RB_TAIL_WAVES batches is NOT run here: its list stays in
 * *tail_list (*tail_n items, generation *tail_gen) and the caller runs it
 * inside phase X (rb_batch() by the waves that loaded the items).  Never the
 * first generation: that one is the bulk of the work and fills every wave.
 */
static __device__ __forceinline__ void
search_queue(lds_t *L, u32 t, u32 n, u32 lo_pos, u32 min_len, u32 depth,
	     u32 nice, AS3 u32 *WA, AS3 u32 *WB, u32 wc, u32 tid,
	     bool defer, u32 *tail_n, u32 *tail_gen, AS3 u32 **tail_list)
{
	const u32 lane = tid & 63, wave = tid >> 6;
	depth = rb_trim_depth(depth);
	const u64 lt = (1ull << lane) - 1;
	u32 ncur = wc;
#ifdef LDA_PROFILE
	unsigned long long pg_ = __builtin_readcyclecounter();
#endif

	*tail_n = 0;
	PROF_COUNT(15, wc);
	PROF_COUNT(18, 1);
	for (u32 gen = 0;; gen++) {
		PROF_COUNT(16, 1);
		PROF_COUNT(14, (ncur + 63) / 64);
		PROF_COUNT(19, ncur);
		AS3 u32 *cur_l = gen & 1 ? WB : WA, *nxt_l = gen & 1 ? WA : WB;
		AS3 u32 *ctr = &L->qn[gen % 3];
		u32 before, npass_g;
		rb_gen_params(depth, gen, &before, &npass_g);
		if (RB_DEFER && defer && gen && before + 8 * npass_g >= depth &&
		    ncur <= 64 * RB_TAIL_WAVES) {
			*tail_n = ncur;
			*tail_gen = gen;
			*tail_list = cur_l;
			return;
		}
		if (tid == 0)
			L->qn[(gen + 1) % 3] = 0;
		for (u32 base = 64 * wave; base < ncur; base += 64 * NWAVES) {
			const bool have = base + lane < ncur;
			const u32 e = have ? cur_l[base + lane] : 0;
			u32 nx;
			const bool surv = rb_batch(L, t, n, lo_pos, min_len, depth, nice, e, have,
						   gen, lane, &nx);
			const u64 b = __ballot(surv);
			u32 at = 0;
			if (lane == 0 && b)
				at = atomicAdd((u32 *)ctr, (u32)__builtin_popcountll(b));
			at = bcast_first(at);
			if (surv)
				nxt_l[at + __builtin_popcountll(b & lt)] = nx;
		}
		__syncthreads();
		PROF_GEN(gen);
		ncur = bcast_first(*(volatile AS3 u32 *)ctr);
		if (!ncur)
			break;
	}
}

/*
 * Block split observations of the tile just emitted (ONE wave; see "block
 * end?" in the schedule): the reference ends a block when the kind of symbols
 * changes (lib/deflate_compress.c:2092-2218) - ten observation classes
 * (literals by their top two bits and low bit, matches shorter / not shorter
 * than 9) and a split when the distribution of the new observations is far
 * from the block's.  Here the classes are sums over the block histogram and
 * "new" is what this tile added.  Leaves V_SPLIT: 0 no split, 2 the part
 * before this tile is a block of its own, 1 the block ends after the tile.
 */
static __device__ __forceinline__ void
split_stats(lds_t *L, u32 walkpos, u32 block_start, bool fit_split, u32 lane)
{
	/* The wave that computes this is the last thing of phase X (every token
	 * of the tile has to be out first) with fifteen waves at the barrier: its
	 * instruction count is on the tile's path.  Class sums: the 320 counts
	 * added into ten LDS words (class of literal sy = lane + 64 j is 2 j +
	 * (lane & 1); matches: length slots 0..5 (3..8) / 6..28); the rest on
	 * lanes 0..9, one class each.  The products fit 32 bits: a tile adds at
	 * most TOK_TILE_MAX < 2^13 observations to at most TOK_CAP < 2^17. */
	static_assert(TOK_TILE_MAX < (1u << 13) && TOK_CAP < (1u << 17), "32-bit products");
	AS3 u32 *const acc = (AS3 u32 *)&L->scan[1][0];	/* (no block scan runs beside this) */
	if (lane < 10)
		acc[lane] = 0;
	wave_sync();
#pragma unroll
	for (u32 j = 0; j < 4; j++)
		atomicAdd((u32 *)&acc[2 * j + (lane & 1)], L->freq[lane + 64 * j]);
	if (lane < 29)
		atomicAdd((u32 *)&acc[8 + (lane >= 6)], L->freq[257 + lane]);
	wave_sync();
	const u32 now = lane < 10 ? acc[lane] : 0;
	const u32 prev = lane < 10 ? L->obs[0][lane] : 0;
	const u32 d = now - prev;
	const u32 nprev = row16_sum(prev), nnew = row16_sum(d);
	const u32 a = d * nprev, e = prev * nnew;
	const u32 df = a > e ? a - e : e - a;
	const u64 delta = ((u64)row16_sum(df >> 16) << 16) + row16_sum(df & 0xFFFF);
	bool sp = nprev && walkpos - block_start >= 5000 &&
		  delta >= (u64)nnew * 200 / 512 * nprev;
	if (fit_split && walkpos - block_start >= 5000)
		sp = true;	/* see opt_build_costs() */
	wave_sync();
	if (lane < 10)
		L->obs[0][lane] = sp ? 0 : now;
	if (lane == 0)
		L->vars[V_SPLIT] = !sp ? 0 :
			L->vars[V_WPOS_PRE] - block_start >= 5000 ? 2 : 1;
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
	u64 *__restrict__ seqg = seq_scratch + (size_t)blockIdx.x * SEQ_STRIDE;
	u32 *__restrict__ tokg = (u32 *)seqg;	/* the block's tokens, see TOK_MATCH */
	/* levels 10-12: the search results of a block's first tile, kept while
	 * that tile is parsed more than once */
	u32 *__restrict__ msave = (u32 *)(seqg + SEQ_GCAP);
	/* the block histogram before the current tile [320] and the tile's own
	 * share [320], for a block that ends in front of the tile */
	u32 *__restrict__ fsave = msave + TILE + 8;
	/* levels 10-12: byte histogram of the block before the previous tile
	 * [256] and of the previous tile [256] */
	u32 *__restrict__ bsave = fsave + 640;
	/* 3-byte-table candidate of every position of the current and of the
	 * next tile (two halves, by tile parity): written by the inserting wave
	 * one tile ahead, read by the shallow search */
	u16 *__restrict__ c3g = (u16 *)(bsave + 512);
	u32 tog = 0;		/* which scan[] array the next single-barrier scan uses */
	PROF_DECL;

	for (u32 i = tid; i < 256; i += NT) {
		u32 sl, xb, xv;
		length_code(i + 3, &sl, &xb, &xv);
		L->lslot[i] = (u8)sl;
	}
	/* Buffers are handed out dynamically (one global counter): their cost
	 * depends on their content, and a fixed stride gives every workgroup
	 * the same kind of buffer whenever the batch is periodic. */
	for (;;) {
		__syncthreads();
		if (tid == 0)
			L->vars[V_TMP0] = atomicAdd(next_chunk, 1u);
		__syncthreads();
		/* (workgroup-uniform values read back from LDS are made scalar: what
		 * derives from them - descriptors, addresses, loop bounds - then lives
		 * in SGPRs instead of vector registers) */
		const u64 c = bcast_first(L->vars[V_TMP0]);
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
#ifdef LDA_SMALL
		if (n64 > RING)		/* the caller's size bound was wrong */
			overflow = true;
#endif
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
		if (tid < TILE / 64 + 1)
			L->rdy[tid] = 0;
		for (u32 i = tid; i < STG_WORDS + 8; i += NT)
			stg_of(L)[i] = 0;
		if (tid < 8)
			L->M[tid] = 0;
		if (tid == 0) {
			L->vars[V_NSEQ] = 0;
			L->vars[V_ENTRY] = 0;
			L->vars[V_PFLAG] = 0;
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

		const bool ra_all = depth <= 4;	/* a search of a few steps is done in full by the first pass */
		const bool optm = OPT && mode == 3;
		const u32 rounds = optm || ra_all ? 0 : S3_ROUNDS;
		const u32 dlim3 = mode ? 8192u : 4096u;
		const u32 ra_depth = ra_all ? depth : S3_RA_DEPTH;
		const u32 ra_class = ra_all ? DC_FULL :
				     ra_depth >= (depth >> S3_HALF_SHIFT) ? DC_HALF : DC_SHALLOW;
		/* one batch of 64 per wave: the lazy levels' first generation of
		 * round B is one round of batches (what is cut off: look-ahead
		 * positions at the end of a tile, +0.0x % size); lazy2 has twice the
		 * look-ahead positions and keeps them all */
		const u32 wq_limit = mode == 1 && depth < 100 && WQ_LIMIT > 64 * NWAVES ?
				     64 * NWAVES : WQ_LIMIT;
		/* level 1: matches of 4 bytes and up like the reference's
		 * (lib/ht_matchfinder.h:50-55: HT_MATCHFINDER_MIN_MATCH_LEN 4), so no
		 * 3-byte table; one block per 64 KiB without split statistics
		 * (deflate_compress_fastest(), lib/deflate_compress.c:2451-2523, ends
		 * blocks by length only) */
		const bool use3 = level >= 2;
		/* split statistics are kept where a block can end before the buffer
		 * does: not at level 1 (blocks end by length only), not in a buffer
		 * shorter than the minimum block length of two blocks' first
		 * (split_stats(): no cut below 5000 bytes; every 4 KiB buffer) */
		const bool splits = use3 && n - dict_len >= 5000;
		bool mx_pending = false;	/* the next tile's search results wait in MX */
		u32 tail_n = 0, tail_gen = 0, tail_e = 0;	/* round B's deferred last generation */
		u32 ml_cur = 3, ml_nxt = 3;	/* minimum match length of tile cur / nxt */
		u32 carryv = 0;			/* M[TILE + tid] of the tile before (tid < 4) */

		/*
		 * The schedule.  Iteration `it` FINISHES tile cur = it - 1 (deep
		 * search of what a first parse visits, final parse, tokens, block
		 * end) and PREPARES the tiles after it: the shallow search of tile
		 * nxt = it and the chain insertion of tile it + 1 run in the same
		 * phase ("phase X") as the final parse and the token emission of
		 * tile cur - the parse is one wave's serial walk and the emission
		 * starts when it ends, the insertion is one wave's serial instruction
		 * stream, and the shallow search is what fills the other waves'
		 * issue slots meanwhile; the step bitmaps and the FIRST parse of
		 * tile nxt follow inside the same phase as soon as its groups are
		 * searched, and the split statistics of cur as soon as its tokens
		 * are out; the last generation of tile cur's deep search, when it
		 * was deferred (search_queue()), runs at the head of the phase on
		 * waves 1 .. RB_TAIL_WAVES, which then compute the final parse's
		 * steps - wave 0 waits for those instead of starting at once.  The
		 * search results of tile nxt land in MX (the LDS of
		 * the round-B lists and the bit staging area, both idle in that
		 * phase) and move to M[] at the top of the next iteration.  The
		 * hand-overs inside the phase are LDS words polled with s_sleep
		 * (V_PFLAG, rdy[], V_STDONE, V_EMDONE, V_TAILDONE, V_ST2DONE): every wait is for work that
		 * some wave is already doing or will do without waiting itself.
		 * Iteration 0 has no cur: it inserts tile 0 and runs phase X for
		 * tile 0 alone; dictionary tiles (segment mode) are only inserted.
		 */
		for (u32 it = 0; it <= num_tiles && !overflow; it++) {
			const bool have_cur = it >= 1;
			const u32 tile = it - 1;	/* cur (meaningless in iteration 0) */
			const u32 t = have_cur ? tile * TILE : 0;
			const u32 tend = t + TILE < n ? t + TILE : n;
			const bool last_tile = it == num_tiles;
			const bool cur_real = have_cur && t >= dict_len;	/* not a dictionary tile */
			const u32 tn = it * TILE;	/* nxt */
			const u32 tnend = tn + TILE < n ? tn + TILE : n;
			const bool nxt_real = it < num_tiles && tn >= dict_len;
			const bool have_ins = it + 1 < num_tiles;
			/* the thread index is made opaque once per tile: otherwise every
			 * per-lane address in this loop body is computed before the loop
			 * and kept alive (in scratch) across it */
			u32 tid_opaque = threadIdx.x;
			asm volatile("" : "+v"(tid_opaque));
			const u32 tid = tid_opaque, lane = tid & 63, wave = tid >> 6;

			PROF_MARK(0);
			if (stored_only) {
				if (!cur_real)	/* iteration 0, dictionary tiles */
					continue;
				walkpos = tend;
			} else {
				AS3 u32 *const MX = (AS3 u32 *)L->nxtB;
#ifdef LDA_SMALL
				const u32 lo_cur = 0, lo_nxt = 0;	/* the whole buffer is resident */
#else
				const s32 lo_s = (s32)(t + 2 * TILE + LOOKAHEAD) - (s32)RING;
				const u32 lo_cur = lo_s > 0 ? (u32)lo_s : 0;
				const s32 lo_n = (s32)(tn + 2 * TILE + LOOKAHEAD) - (s32)RING;
				const u32 lo_nxt = lo_n > 0 ? (u32)lo_n : 0;
#endif
				u16 *c3nxt = c3g + (it & 1) * (TILE + 8);
				u16 *c3ins = c3g + ((it + 1) & 1) * (TILE + 8);
				const s32 limit = last_tile ? (s32)(tend - t) : (s32)TILE - 2;

				/* (with a round B ahead, build_worklist() moves the results
				 * and its first barrier is this block's) */
				const bool wl_copies = cur_real && !optm && rounds && mx_pending;
				if (cur_real) {
					/* ---- tile cur: its shallow results into M[] ---- */
					if (mx_pending && !wl_copies)
						for (u32 i = tid; i < TILE; i += NT)
							L->M[4 + i] = MX[4 + i];
					if (tid < 4) {
						L->M[tid] = carryv;
						L->M[TILE + 4 + tid] = 0;
					}
					mx_pending = false;
					ml_cur = ml_nxt;
					/* the block as it is before this tile's tokens: if the tile
					 * turns out to be of different content, the block ends in
					 * front of it (see "block end?") */
					for (u32 i = tid; i < 320; i += NT)
						fsave[i] = L->freq[i];
					if (tid == 0) {
						L->vars[V_NSEQ_PRE] = L->vars[V_NSEQ];
						L->vars[V_WPOS_PRE] = walkpos;
					}
					if (!wl_copies)
						__syncthreads();
					PROF_MARK(6);
				}
				/* the input of the tile that S0 stages below is requested now:
				 * its trip from HBM runs beside round B */
				const u32 s0_want0 = (it + 2) * TILE + LOOKAHEAD;
				const u32 s0_want = s0_want0 < n ? s0_want0 : n;
				const u32 s0_p = (loaded & ~15u) + tid * 16;
				const bool s0_pre = aligned_in && s0_p < s0_want;
				uint4 s0_v = make_uint4(0, 0, 0, 0);
				if (s0_pre)
					s0_v = *(const uint4 *)(inp + s0_p);
				tail_n = 0;
				if (cur_real && !optm) {
					/* ---- S3 round B: the positions the first parse visited
					 * (pmA: it ran beside phase X of the iteration before) are
					 * searched deeper ("progressive search") ---- */
					if (rounds) {
						const u32 wc = build_worklist(L, (AS3 u32 *)L->nxtB,
									      wl_copies ? (const AS3 u32 *)MX : NULL,
									      mode, nice, wq_limit, tid);
						PROF_MARK(39);
						if (wc) {
							AS3 u32 *tl = (AS3 u32 *)L->nxtB;
							search_queue(L, t, n, lo_cur, ml_cur, depth, nice,
								     (AS3 u32 *)L->nxtB, (AS3 u32 *)L->nxtA, wc, tid,
								     nxt_real, &tail_n, &tail_gen, &tl);
							/* a deferred last generation: its items leave the
							 * list (the next tile's search results take the
							 * list's LDS in phase X) for a register of the
							 * waves that will walk them */
							/* (0 = no item: an entry carries its depth class,
							 * DC_HALF or DC_FULL, in bits 12-13) */
							if (tail_n && wave >= 1 && wave <= RB_TAIL_WAVES) {
								const u32 k = 64 * (wave - 1) + lane;
								tail_e = k < tail_n ? tl[k] : 0;
							}
						}
						PROF_MARK(3);
					}
					/* steps of the final parse (it runs in phase X; behind a
					 * deferred generation they are computed there as well) */
					/* (no barrier here: S0 below writes what neither the steps
					 * nor a tail wave's item load read - the ring's oldest tile,
					 * which round B's last barrier has released, counters, and
					 * of MX only entries behind the lists - and ends with one) */
					if (!tail_n)
						stage_steps(L, limit, mode, nice, tid);
				}
				if (cur_real && optm) {
					/* levels 10-12 (mode 3): min-cost parse, see opt_parse_wave().
					 * A block's first tile is parsed lazily as a dry run (stage 0),
					 * rolled back, and parsed again with prices from that run
					 * (stage 1; level 11 and up once more with the prices of stage
					 * 1); later tiles take their prices from the block so far.
					 * Parse and emission run here, before phase X: the price
					 * tables and the chosen lengths use the LDS of MX. */
					const bool opt_first = walkpos == block_start;
					const u32 opt_last = level >= 11 ? 2 : 1;
					u32 opt_stage = 0;
					const u32 ent0 = L->vars[V_ENTRY], nseq0 = L->vars[V_NSEQ];
					if (opt_first) {
						for (u32 i = tid; i < TILE + 8; i += NT)
							msave[i] = L->M[i];
						if (tid == 0)
							L->vars[V_FIT] = 0;
					} else {
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
						const u32 s4mode = opt_stage ? 0 : 2;
						if (opt_stage) {
							const s32 ent = (s32)ent0;
							const s32 pend = (s32)(tend - t);
							s32 lo = (s32)(OPT_SEG * wave), hi = lo + OPT_SEG;
							if (wave == 0 && ent < 0)
								lo = ent;
							if (hi > limit)
								hi = limit;
							s32 e = hi + OPT_WARM;
							if (e > pend)
								e = pend;
							if (lo < hi)
								opt_parse_wave(L, L->nxtA, t, lo, hi, e, lane);
							__syncthreads();
							for (s32 i = (s32)tid + (ent < 0 ? ent : 0); i < limit; i += NT) {
								u32 c = L->nxtA[i + 4], m = L->M[i + 4];
								L->M[i + 4] = c >= 3 ? c | (m & 0xFFFF0000u) : 0;
							}
							__syncthreads();
						}
						stage_steps(L, limit, s4mode, nice, tid);
						if (tid == 0)
							L->vars[V_CTR2] = 0;
						__syncthreads();
						if (wave == 0)
							parse_and_base(L, tokg, t, limit, s4mode, nice, lane);
						__syncthreads();
						PROF_MARK(12);
						emit_groups(L, tokg, t, lane);
						__syncthreads();
						PROF_MARK(5);
						if (!opt_first || opt_stage == opt_last)
							break;
						/* prices from this parse, then undo it */
						opt_build_costs(L, tid, opt_stage == 0, false, t, tend - t, bsave);
						for (u32 i = tid; i < 320; i += NT)
							L->freq[i] = 0;
						for (u32 i = tid; i < TILE + 8; i += NT)
							L->M[i] = msave[i];
						if (tid == 0) {
							L->vars[V_ENTRY] = ent0;
							L->vars[V_NSEQ] = nseq0;
						}
						__syncthreads();
						opt_stage++;
					}
				}

				/* ---- S0: input up to the end of tile it + 1 (+ LOOKAHEAD): what
				 * the shallow search of tile nxt reads and what the insertion
				 * of tile it + 1 hashes ---- */
				{
					const u32 want = s0_want;
					if (s0_pre) {
						*(uint4 *)&L->in[s0_p & RMASK] = s0_v;
						if ((s0_p & RMASK) < 32)
							*(uint4 *)&L->in[RING + (s0_p & RMASK)] = s0_v;
					}
					if (aligned_in) {	/* (more than one unit per thread: first tile only) */
						if ((loaded & ~15u) + NT * 16 < want)
							stage_input(L, inp, (loaded & ~15u) + NT * 16, want, true, tid);
					} else {
						stage_input(L, inp, loaded, want, false, tid);
					}
					loaded = want > loaded ? want : loaded;
					if (tid == 0) {
						L->vars[V_CTR] = 0;
						L->vars[V_CTR2] = 0;
						L->vars[V_CTR3] = 0;
						L->vars[V_STDONE] = 0;
						L->vars[V_EMDONE] = 0;
						L->vars[V_TAILDONE] = 0;
						L->vars[V_CTR4] = 0;
						L->vars[V_ST2DONE] = 0;
					}
					/* what the first parse of tile nxt reads around its search
					 * results: the entries its predecessor's walk deferred, and
					 * no matches past the end */
					if (tid < 4 && nxt_real) {
						AS3 u32 *const Mo = cur_real ? MX : (AS3 u32 *)L->M;
						/* (behind a deferred generation of round B the
						 * entries may still change: wave 0 copies them after
						 * its final parse) */
						if (!tail_n)
							Mo[tid] = cur_real ? L->M[TILE + tid] : 0;
						Mo[TILE + 4 + tid] = 0;
					}
					/* (the minimum match length of tile nxt was estimated a
					 * tile ahead, beside phase X) */
					if (nxt_real && it)
						ml_nxt = bcast_first(L->vars[V_MINLEN]);
					__syncthreads();
				}
				PROF_MARK(1);
				if (it == 0) {
					/* the first tile joins the chains up front */
					if (wave == NWAVES - 1)
						insert_tile(L, 0, tnend, n, lane);
					if (wave == NWAVES - 2 && use3) {
						insert_tile3(L, c3nxt, 0, tnend, n, lane);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					}
					if (wave == NWAVES - 3 && nxt_real) {
						const u32 nd = wave_distinct_bytes(L, 0, loaded < TILE ? loaded : TILE, lane);
						if (lane == 0)
							L->vars[V_MINLEN] = min_len_policy(nd, n - dict_len, depth, use3);
					}
					__syncthreads();
					if (nxt_real)
						ml_nxt = bcast_first(L->vars[V_MINLEN]);
					PROF_MARK(2);
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

				/* ---- phase X ---- */
				{
					const bool do_p2 = cur_real && !optm;
					AS3 u32 *const Mo = cur_real ? MX : (AS3 u32 *)L->M;
					PROF_WDECL;
					PROF_W0();
					if (wave == NWAVES - 1) {
						/* the chain insertion of tile it + 1: one wave's serial
						 * instruction stream, the longest item of this phase -
						 * it gets the issue slots of its SIMD first */
						__builtin_amdgcn_s_setprio(3);
						if (have_ins)
							insert_tile(L, tnend, tnend + TILE < n ? tnend + TILE : n,
								    n, lane);
						__builtin_amdgcn_s_setprio(0);
						PROF_W(24);
					} else if (wave == NWAVES - 2) {
						__builtin_amdgcn_s_setprio(2);
						if (have_ins && use3) {
							insert_tile3(L, c3ins, tnend,
								     tnend + TILE < n ? tnend + TILE : n, n, lane);
							__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						}
						__builtin_amdgcn_s_setprio(0);
						/* the minimum match length of the tile after nxt */
						if (have_ins && tnend >= dict_len) {
							const u32 nd = wave_distinct_bytes(L, tnend,
								loaded - tnend < TILE ? loaded - tnend : TILE, lane);
							if (lane == 0)
								L->vars[V_MINLEN] = min_len_policy(nd, n - dict_len, depth, use3);
						}
						PROF_W(25);
					} else if (wave == 0 && do_p2) {
						/* ---- S4: the final parse of tile cur ----
						 * step(p) is a pure function of M[p..p+2]; the chosen
						 * tokens are the positions reachable from the entry
						 * point by p -> p + step(p); idx = p + 4 */
						if (tail_n)	/* its steps follow round B's last generation */
							wait_lds_eq(L, V_ST2DONE, TILE / 64);
						__builtin_amdgcn_s_setprio(3);
						parse_and_base(L, tokg, t, limit, mode, nice, lane);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						if (lane == 0)
							*(volatile AS3 u32 *)&L->vars[V_PFLAG] = it;
						__builtin_amdgcn_s_setprio(0);
						/* (the entries the next tile's first parse takes over
						 * from this one may have changed until now) */
						if (tail_n && nxt_real && lane < 4)
							Mo[lane] = L->M[TILE + lane];
						PROF_W(26);
					} else if (RB_DEFER && tail_n && wave <= RB_TAIL_WAVES) {
						/* ---- round B's last generation, deferred (waves 1 ..
						 * RB_TAIL_WAVES; wave 0 stays free to start the final
						 * parse the moment its steps are there: with a batch of
						 * its own it measured 0.6 % slower): beside the shallow
						 * search of tile nxt.  Phase X has staged the input of
						 * the tile after nxt and inserts it meanwhile, over the
						 * oldest tile of the window round B had: these items
						 * search with the window of tile nxt ---- */
						__builtin_amdgcn_s_setprio(2);
						if (64 * (wave - 1) < tail_n) {
							u32 nx;
							(void)rb_batch(L, t, n, lo_nxt, ml_cur, rb_trim_depth(depth), nice,
								       tail_e, tail_e != 0, tail_gen, lane, &nx);
						}
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						if (lane == 0)
							atomicAdd((u32 *)&L->vars[V_TAILDONE], 1u);
						/* (all of them share the steps: a tail of one batch
						 * would leave one wave with the whole tile's - measured
						 * 6 % slower than no deferral at all) */
						wait_lds_eq(L, V_TAILDONE, RB_TAIL_WAVES);
						stage_steps_cur_claimed(L, limit, mode, nice, lane);
						__builtin_amdgcn_s_setprio(0);
					}
					/* ---- S3 round A: tile nxt ---- */
					if (nxt_real) {
						if (optm) {
							/* the min-cost parse prices every position: all of
							 * them are searched to the full depth */
							if (wave < NWAVES - 2)
								search_items(L, Mo, tn, n, lo_nxt, ml_nxt, depth, nice,
									     NULL, TILE, NWAVES - 2, tid);
						} else {
							round_a(L, Mo, c3nxt, tn, tnend, n, lo_nxt, ml_nxt, ra_depth,
								ra_class, nice, dlim3, it + 1, tid);
						}
					}
					if (wave == 1)
						PROF_W(27);
					if (nxt_real && rounds) {
						/* ---- the first parse of tile nxt: steps by all waves as
						 * soon as every group of round A is through, the walk by
						 * wave 0 (its entry point is where the final parse of
						 * tile cur, which it ran itself, left) ---- */
						const s32 lim_n = it + 1 == num_tiles ? (s32)(tnend - tn) : (s32)TILE - 2;
						stage_steps_claimed(L, Mo, lim_n, mode, nice, it + 1, lane);
						if (wave == 0) {
							u64 mk;
							wait_lds_eq(L, V_STDONE, TILE / 64);
							const u32 e = entry_skip(Mo, (s32)L->vars[V_ENTRY],
										 (u32)(lim_n + 4), mode, nice);
							(void)parse_tile(Mo, (const AS3 u64 *)L->lit1A,
									 (const AS3 u64 *)L->lit2A, (AS3 u64 *)L->pmA,
									 lane, (s32)e - 4, lim_n, &mk, P1_PASSES);
						}
						if (wave == 1)
							PROF_W(33);
						if (wave == 0)
							PROF_W(34);
					}
					if (do_p2) {
						/* ---- the tokens of tile cur, as soon as its parse is
						 * through ---- */
						while (*(volatile AS3 u32 *)&L->vars[V_PFLAG] != it)
							__builtin_amdgcn_s_sleep(4);
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
						if (wave == 1)
							PROF_W(28);
						emit_groups(L, tokg, t, lane);
						if (wave == 1)
							PROF_W(29);
						/* block split observations of tile cur, by the last wave
						 * once every group is emitted (wave 0 is still walking
						 * the next tile's first parse) */
						if (wave == NWAVES - 1 && splits && !last_tile) {
							wait_lds_eq(L, V_EMDONE, TILE / 64);
							split_stats(L, bcast_first(L->vars[V_WALKPOS_LO]), block_start, false, lane);
						}
					}
					__syncthreads();
					if (wave == 1)
						PROF_W(30);
					if (wave == 0)
						PROF_W(31);
					if (optm && nxt_real) {
						PROF_MARK(16);
						/* length-3 matches for the positions left without a match */
						if (ml_nxt <= 3) {
							for (u32 i = tid; i < TILE; i += NT) {
								u32 p = tn + i, b3 = 0;
								if (p + 3 <= n && Mo[4 + i] == 0) {
									u32 bd = find_len3(L, p, ld32(L->in, p),
											   c3nxt[4 + i], p - lo_nxt,
											   dlim3, &b3);
									if (bd)
										Mo[4 + i] = 3 | (bd << 16);
								}
							}
						}
						__syncthreads();
					}
					mx_pending = cur_real && nxt_real;
				}
				PROF_MARK(4);
				if (!cur_real)
					continue;	/* iteration 0, dictionary tiles: nothing to emit */

				/* (levels 0-9: walkpos, the split decision and the token count
				 * come with one read, see "block end?") */
				if (OPT && mode == 3)
					walkpos = bcast_first(L->vars[V_WALKPOS_LO]);
				/* the last 4 match entries go to the front of the next tile's,
				 * for the positions the walk deferred */
				if (tid < 4)
					carryv = L->M[TILE + tid];
				/* block split observations (see "block end?" below), by the
				 * last wave while the others wait at the barrier */
				if (splits && wave == NWAVES - 1 && optm) {
					/* (levels 0-9: done in phase X, beside the first parse of
					 * the next tile) */
					split_stats(L, walkpos, block_start, L->vars[V_FIT] == 2, lane);
				}
			}
			/* (levels 0-9: the split decision was written inside phase X, in
			 * front of its closing barrier) */
			if (OPT && mode == 3)
				__syncthreads();
			PROF_MARK(32);

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
			u32 splitv, nseq_now;
			if (OPT && mode == 3) {
				splitv = splits && !stored_only && !last_tile ?
					 bcast_first(L->vars[V_SPLIT]) : 0;
				nseq_now = stored_only ? 0 : bcast_first(L->vars[V_NSEQ]);
			} else {
				/* (three dependent LDS round trips were 3 % of a tile) */
				const uint4 pv = *(const AS3 uint4 *)&L->vars[V_NSEQ];
				static_assert(V_NSEQ == 0 && V_WALKPOS_LO == 2 && V_SPLIT == 3, "one 16-byte read");
				if (!stored_only)
					walkpos = bcast_first(pv.z);
				splitv = splits && !stored_only && !last_tile ? bcast_first(pv.w) : 0;
				nseq_now = stored_only ? 0 : bcast_first(pv.x);
			}
			bool end_block = last_tile || splitv ||
				(!stored_only && nseq_now + 2 * TOK_TILE_MAX > TOK_CAP) ||
				walkpos - block_start > MAX_BLOCK_LEN;
			if (!end_block)
				continue;
			if (tid < 10)
				L->obs[0][tid] = 0;

			const bool retro = splitv == 2;
			const u32 bstart = block_start;
			const u32 bend = last_tile ? n : retro ? bcast_first(L->vars[V_WPOS_PRE]) : walkpos;
			const u32 blen = bend - bstart;
			const u32 nseq_all = nseq_now;
			const u32 nseq = retro ? bcast_first(L->vars[V_NSEQ_PRE]) : nseq_all;
			const u32 is_final = last_tile && seg_last ? 1 : 0;
			if (retro) {
				for (u32 i = tid; i < 320; i += NT) {
					u32 f = L->freq[i], fp = fsave[i];
					fsave[320 + i] = f - fp;
					L->freq[i] = fp;
				}
				__syncthreads();
			}

			/* ---- S5: codes, costs, block type ---- */
			u32 btype = 0;	/* 0 stored, 1 static, 2 dynamic */
			if (!stored_only) {
				/* the block-end tables and the bit staging area are MX's
				 * LDS: the next tile's search results wait in HBM meanwhile */
				if (mx_pending) {
					const AS3 u32 *MXs = (const AS3 u32 *)L->nxtB;
					for (u32 i = tid; i < TILE; i += NT)
						msave[4 + i] = MXs[4 + i];
				}
				if (tid == 0)
					L->freq[256]++;
				__syncthreads();
				/* rank sort of both alphabets by the whole workgroup: the
				 * used symbols are collected first (their keys, freq << 9 |
				 * symbol, in any order), then every key counts the keys
				 * below it - a block uses a third of the litlen alphabet,
				 * a small one a fifth; M[] is free scratch here */
				{
					u32 *keys = L->M;		/* [288] litlen, [288, 320) offset keys */
					u32 *usedv = L->M + 320;	/* [2] used counts */
					u16 *sortedO = (u16 *)(L->M + 324);	/* [32] */
					if (tid < 2)
						usedv[tid] = 0;
					__syncthreads();
					for (u32 vt = tid; vt < 320; vt += NT) {
						const u32 f = L->freq[vt];
						if (f) {
							if (vt < 288)
								keys[atomicAdd(&usedv[0], 1u)] = (f << 9) | vt;
							else
								keys[288 + atomicAdd(&usedv[1], 1u)] =
									(f << 9) | (vt - 288);
						}
					}
					__syncthreads();
					{
						const u32 m1 = usedv[0], m2 = usedv[1];
						for (u32 i = tid; i < m1 + m2; i += NT) {
							const bool lit = i < m1;
							const u32 lo = lit ? 0 : 288, m = lit ? m1 : m2;
							const u32 key = keys[lit ? i : 288 + i - m1];
							u32 r = 0;
							for (u32 q = 0; q < m; q++)
								r += keys[lo + q] < key;
							if (lit)
								L->sorted[r] = (u16)(key & 511);
							else
								sortedO[r] = (u16)(key & 511);
						}
					}
					__syncthreads();
					PROF_MARK(10);
					/* the two trees are built side by side on two waves */
					if (wave == 0)
						make_code(L->freq, 288, 15, L->lens, L->codes,
							  L->sorted, HUFF_LITLEN(L),
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
					for (u32 vt = tid; vt < 320; vt += NT) {
						if (vt < 288 && vt >= 257 && L->lens[vt])
							atomicMax((u32 *)&L->vars[V_TMP1], vt + 1);
						if (vt >= 288 && L->lens[vt])
							atomicMax((u32 *)&L->vars[V_TMP2], vt - 288 + 1);
					}
					__syncthreads();
					const u32 nlit = L->vars[V_TMP1], noff = L->vars[V_TMP2];
					const u32 total = nlit + noff;
					/* element e of the concatenated lengths: thread tid owns
					 * the VPT consecutive elements from tid * VPT */
					u32 isst[VPT], nst = 0;
#pragma unroll
					for (u32 j = 0; j < VPT; j++) {
						const u32 e = tid * VPT + j;
						isst[j] = 0;
						if (e < total) {
							u32 v = L->lens[e < nlit ? e : 288 + (e - nlit)];
							u32 pv = 0xFF;
							if (e)
								pv = L->lens[e - 1 < nlit ? e - 1 :
									     288 + (e - 1 - nlit)];
							isst[j] = pv != v;
						}
						nst += isst[j];
					}
					u32 nruns;
					u32 ridx = block_scan(L, nst, &nruns);
#pragma unroll
					for (u32 j = 0; j < VPT; j++) {
						if (isst[j])
							starts[ridx] = tid * VPT + j;
						ridx += isst[j];
					}
					if (tid == 0)
						starts[nruns] = total;
					__syncthreads();
					/* run r: thread tid owns the runs from tid * VPT */
					u32 rv[VPT], rlen[VPT], nitems[VPT], nit = 0;
#pragma unroll
					for (u32 j = 0; j < VPT; j++) {
						const u32 r = tid * VPT + j;
						rv[j] = rlen[j] = nitems[j] = 0;
						if (r < nruns) {
							u32 st = starts[r];
							rlen[j] = starts[r + 1] - st;
							rv[j] = L->lens[st < nlit ? st : 288 + (st - nlit)];
							if (rv[j] == 0) {
								u32 full = rlen[j] / 138, rem = rlen[j] % 138;
								nitems[j] = full + (rem >= 3 ? 1 : rem);
							} else if (rlen[j] >= 4) {
								u32 l1 = rlen[j] - 1;
								nitems[j] = 1 + l1 / 6 + (l1 % 6 >= 3 ? 1 : l1 % 6);
							} else {
								nitems[j] = rlen[j];
							}
						}
						nit += nitems[j];
					}
					u32 ni;
					u32 at = block_scan(L, nit, &ni);
#pragma unroll
					for (u32 j = 0; j < VPT; j++) {
						if (tid * VPT + j < nruns) {
							u32 left = rlen[j];
							const u32 rvj = rv[j];
							if (rvj == 0) {
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
								L->pre_items[at++] = (u16)rvj;
								left--;
								u32 n16 = 0;
								while (left >= 3) {
									u32 r = left > 6 ? 6 : left;
									L->pre_items[at++] = 16 | ((r - 3) << 5);
									n16++;
									left -= r;
								}
								atomicAdd((u32 *)&L->pre_freq[16], n16);
								atomicAdd((u32 *)&L->pre_freq[rvj], 1u);
							}
							if (left)
								atomicAdd((u32 *)&L->pre_freq[rvj], left);
							while (left) {
								L->pre_items[at++] = (u16)rvj;
								left--;
							}
						}
					}
					if (tid == 0)
						L->vars[V_NPRE] = ni;
				}
				__syncthreads();
				PROF_MARK(23);
				if (wave == 0)
					make_code(L->pre_freq, 19, 7, L->pre_lens, L->pre_codes,
						  L->sorted, (huff_scratch<32> *)L->hw,
						  0, false, lane);
				__syncthreads();
				PROF_MARK(22);
				/* exact costs (deflate_compress.c:1747-1808) */
				u32 dyn = 0, stat = 0;
				for (u32 vt = tid; vt < 320; vt += NT) {
					u32 f = L->freq[vt];
					u32 xb = 0, sl = 8;
					if (vt < 288) {
						sl = vt < 144 ? 8 : vt < 256 ? 9 : vt < 280 ? 7 : 8;
						if (vt >= 265 && vt < 285)
							xb = (vt - 261) >> 2;
					} else {
						u32 ds = vt - 288;
						sl = 5;
						if (ds >= 4)
							xb = (ds >> 1) - 1;
					}
					dyn += f * (L->lens[vt] + xb);
					stat += f * (sl + xb);
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
					u64 hcode[VPT];
					u32 hbits[VPT], hsum = 0;
					const u32 nexp = btype == 2 ? L->vars[V_TMP3] : 0;
					const u32 ni = btype == 2 ? L->vars[V_NPRE] : 0;
#pragma unroll
					for (u32 j = 0; j < VPT; j++) {
						const u32 vt = tid * VPT + j;	/* header item */
						hcode[j] = 0;
						hbits[j] = 0;
						if (vt == 0) {
							hcode[j] = is_final | (btype << 1);
							hbits[j] = 3;
							if (btype == 2) {
								u32 nlit = L->vars[V_TMP1], noff = L->vars[V_TMP2];
								hcode[j] |= (u64)((nlit - 257) | ((noff - 1) << 5) |
										  ((nexp - 4) << 10)) << 3;
								hbits[j] = 17;
							}
						} else if (vt <= nexp) {
							static const u8 perm2[19] = { 16, 17, 18, 0, 8, 7,
								9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
							hcode[j] = L->pre_lens[perm2[vt - 1]];
							hbits[j] = 3;
						} else if (vt <= nexp + ni) {
							u32 it = L->pre_items[vt - nexp - 1];
							u32 sym = it & 31, ex = it >> 5;
							u32 l = L->pre_lens[sym];
							u32 xb = sym == 16 ? 2 : sym == 17 ? 3 :
								 sym == 18 ? 7 : 0;
							hcode[j] = L->pre_codes[sym] | ((u64)ex << l);
							hbits[j] = l + xb;
						}
						hsum += hbits[j];
					}
					u32 htot;
					u32 hoff = block_scan(L, hsum, &htot);
#pragma unroll
					for (u32 j = 0; j < VPT; j++) {
						stg_put(L, &os, os.bits + hoff, hcode[j], hbits[j]);
						hoff += hbits[j];
					}
					os.bits += htot;
				}
				stg_flush(L, &os, false);

				PROF_MARK(9);
				/* tokens, NT at a time: one token per thread, a workgroup
				 * prefix sum of the bit lengths (single-barrier scan),
				 * ds_or into the staging area.  The staging area is only
				 * written out when another window might not fit (a token is
				 * at most 48 bits: 6 KiB per window in the worst case, a
				 * fifth of that on text). */
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				/* (the next window's token is requested before this one is
				 * encoded: the list is in HBM) */
				u32 tok_nxt = tid < nseq ? tokg[tid] : 0;
				for (u32 b0 = 0; b0 < nseq; b0 += NT) {
					u64 code = 0;
					u32 nb = 0;
					const u32 tok = tok_nxt;
					if (b0 + NT + tid < nseq)
						tok_nxt = tokg[b0 + NT + tid];
					if (b0 + tid < nseq) {
						if (tok & TOK_MATCH) {
							const u32 len = (tok & 0xFF) + 3, dist = ((tok >> 8) & 0x7FFF) + 1;
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
							code = v;
							nb = sh;
						} else {
							code = L->codes[tok];
							nb = L->lens[tok];
						}
					}
					u32 tot;
					u32 off = block_scan1(L, nb, &tot, &tog);
					stg_put(L, &os, os.bits + off, code, nb);
					os.bits += tot;
					/* room for one more window of 48-bit tokens? */
					if (S6_ALWAYS_FLUSH || os.bits - 8 * os.sg + 48 * NT + 64 > 32 * STG_WORDS)
						stg_flush(L, &os, false);
				}
				stg_flush(L, &os, false);
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
			if (OPT && mode == 3 && tid < 256)
				bsave[tid] = 0;	/* the previous tile's bytes are added by the next one */
			if (retro) {
				/* the last tile's tokens open the next block: they move to
				 * the front of the list, its histogram becomes the block's */
				const u32 cnt = nseq_all - nseq;	/* <= TOK_TILE_MAX */
				enum { RPT = (TOK_TILE_MAX + NT - 1) / NT };
				u32 mv[RPT];
#pragma unroll
				for (u32 k = 0; k < RPT; k++)
					mv[k] = tid + NT * k < cnt ? tokg[nseq + tid + NT * k] : 0;
				__syncthreads();
#pragma unroll
				for (u32 k = 0; k < RPT; k++)
					if (tid + NT * k < cnt)
						tokg[tid + NT * k] = mv[k];
				for (u32 i = tid; i < 320; i += NT)
					L->freq[i] = fsave[320 + i];
				if (tid == 0)
					L->vars[V_NSEQ] = cnt;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			} else {
				for (u32 i = tid; i < 320; i += NT)
					L->freq[i] = 0;
				if (tid == 0)
					L->vars[V_NSEQ] = 0;
			}
			/* the positions the walk deferred into the next tile are
			 * re-evaluated as "no match" after a block end */
			carryv = 0;
			if (mx_pending) {
				for (u32 i = tid; i < TILE; i += NT)
					L->M[4 + i] = msave[4 + i];
				mx_pending = false;
			}
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

#ifdef LDA_SMALL
/* buffers of at most RING bytes, 256 threads, SMALL_WGS workgroups per CU
 * (= waves per SIMD: four leave 128 VGPRs each): deflate_small.hip */
extern "C" __global__ void __launch_bounds__(NT, SMALL_WGS)
lda_deflate_small_kernel(DEFLATE_KERNEL_PARAMS)
{
	extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
	deflate_batch_body<false>(DEFLATE_KERNEL_ARGS);
}

extern "C" size_t lda_deflate_small_lds_bytes(void)
{
	return sizeof(struct deflate_lds);
}

extern "C" size_t lda_deflate_small_max(void)
{
	return RING;
}

extern "C" size_t lda_deflate_small_wgs(void)
{
	return SMALL_WGS;
}

LDA_PROF_DEFINE_READER(libdeflate_amd_profile_read_deflate_small)
#else
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

#endif /* LDA_SMALL */
