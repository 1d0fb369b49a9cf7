/*
 * inflate_stream.hip - ONE large DEFLATE stream decoded by many waves.
 *
 * The reference decodes a stream front to back, one token after the other
 * (lib/decompress_template.h:344-671), and that is what the batch kernels of
 * inflate_kernel.hip do per stream: a stream of 16 MiB keeps one wave busy and
 * 255 CUs idle.  This is what programs/gzip.c:187-303 and the default 1 MiB
 * chunks of programs/benchmark.c:543-544 hand to libdeflate_*_decompress, so
 * the single-buffer calls get a second path (host_stream.hip) built from these
 * kernels:
 *
 *   find     every bit offset of the compressed stream is tried as the start
 *            of a dynamic Huffman block (BTYPE, HLIT/HDIST, a complete precode:
 *            lda_stream_find_a_kernel; then the code lengths decoded in full
 *            and both codes checked: lda_stream_find_b_kernel).  What passes
 *            is, with overwhelming probability, a block boundary; the chain
 *            check below does not depend on it.
 *   plan     (host) the stream is cut into chunks of a few KiB of input.  A
 *            chunk starts at a found block header, or INSIDE a block: a parse
 *            started at an arbitrary bit falls in step with the true one
 *            within a few dozen bits (the property the rounds of
 *            inflate_kernel.hip rest on), so a chunk that knows its block's
 *            header parses 1 KiB of warm-up in front of its nominal start,
 *            and the first token boundary at or after that start is its start.
 *   count    lda_stream_count_kernel: a wave per chunk parses its chunk up to
 *            the first token (or block) boundary at or after the next chunk's
 *            nominal start (runs of stored blocks are walked through) and
 *            reports where it started, where it ended, under which block
 *            header, and how many bytes it makes.
 *   chain    (host) chunk 0 starts at the stream's first bit and is exact.
 *            A chunk is accepted iff its start - position, governing header,
 *            boundary kind - is exactly the end of an accepted chunk, so by
 *            induction every accepted chunk is the reference's parse.  Where
 *            the chain breaks (a block the finder does not look for - stored,
 *            static -, a false candidate, a warm-up that did not fall in
 *            step) a repair chunk is counted from the exact end state - for
 *            every open end at once, one launch per round; too many repairs,
 *            any error, or an output that does not fit send the whole stream
 *            to the sequential kernel, which owns the result codes.  The input
 *            goes through find .. chain in WINDOWS of growing size, so a
 *            stream that ends early never has the rest of the buffer touched.
 *   decode   lda_stream_decode_kernel: the same parse again (tokens are
 *            cheaper to decode twice than to keep), now executed: 16-bit
 *            symbols, a byte or - for a match source in front of the chunk's
 *            first byte - a MARKER 0x8000 | index into the 32 KiB in front of
 *            the chunk.
 *   window   lda_stream_window_kernel (+ lda_stream_window_scan_kernel): the
 *            last 32 KiB of every chunk settled against the 32 KiB in front
 *            of it - a chain through all chunks, run as a two-level scan over
 *            groups of chunks with the window as a ring in LDS (see "markers
 *            -> bytes" below).
 *   resolve  lda_stream_resolve_kernel: every other symbol, all in parallel.
 *
 * Not a restatement of anything in the reference; the validity rules of the
 * headers are those of lib/deflate_decompress.c:721-1004 and
 * lib/decompress_template.h:85-245 through the functions of inflate_kernel.hip.
 */
#define LDA_INFLATE_DEVICE_ONLY
#ifndef STREAM_PAR_CB
#define STREAM_PAR_CB 384u	/* this file's rounds keep their own piece length (LDS per chunk wave) */
#endif
#undef PAR_CB
#define PAR_CB STREAM_PAR_CB
/* (this file's kernels build a block's tables once per CHUNK of a few KiB: they
 * keep the 9 / 7-bit tables for every block, and the 4 Ki-entry mirror) */
#define LIT_TB 9
#define LIT_TB_MIN 9
#define OFF_TB 7
#define OFF_TB_MIN 7
#define PAR_RW 4096u
#include "inflate_kernel.hip"
#include "stream_kernels.h"

typedef AS3 u16 lsym;

#define SM_COUNT 1
#define SM_MARK 2
/* LDS of a wave of the block finder's second stage */
#define FIND_B_TAB_BYTES 16384u		/* 64 precode tables of 128 u16 */
#define FIND_B_LDS (FIND_B_TAB_BYTES + 64 * 72)	/* + a row of input per lane */

/* LDS of one chunk wave: tables, shared tables, 16-bit output mirror, stage +
 * token map (the layout of lda_inflate_wave_kernel with a 16-bit mirror) */
#define STREAM_LDS (sizeof(struct stream_lds) + sizeof(struct shared_lds) + \
		    2 * PAR_RW + PAR_STAGE_BYTES + PAR_MAP_BYTES)

/* positions [flushed, end) are in the mirror and not yet in memory: store
 * the whole pairs among them (a 4-byte word of two symbols) */
static __device__ __forceinline__ u64
flush_ring16(u16 *__restrict__ sym, const lsym *win, u64 flushed, u64 end, u32 lane)
{
	u64 a = (flushed + 1) & ~(u64)1;
	if (a > end)
		return flushed;
	if (flushed < a && lane == 0)
		sym[flushed] = win[(u32)flushed & (PAR_RW - 1)];
	const u64 e = end & ~(u64)1;
	if (e <= a)
		return a;
	const u32 nw = (u32)(e - a) >> 1;
	u32 *dst = (u32 *)(sym + a);
	const u32 a32 = (u32)a;
	for (u32 w = lane; w < nw; w += 64)
		dst[w] = *(const lu32 *)(win + ((a32 + 2 * w) & (PAR_RW - 1)));
	return e;
}

/*
 * One round of a chunk: par_round() of inflate_kernel.hip with
 *   - a LIMIT: the round (and with it the chunk) ends at the first token
 *     boundary at or after bit `limit_abs`;
 *   - WARM rounds, whose lane 0 starts at a guess like every other lane and
 *     which run over end-of-block symbols (they only look for the boundary);
 *   - MODE SM_COUNT: no tokens kept, nothing executed, bytes counted;
 *   - MODE SM_MARK: tokens executed into 16-bit symbols at absolute output
 *     positions; a source in front of `chunk_abs` becomes a marker.
 * *bad_ret is set when a distance reaches back before the stream's first byte.
 */
template <int MODE> static __device__ u32
chunk_round(const u8 *inp, u64 in_n, const slds_t *S, const shlds_t *SH,
	    u32 *__restrict__ tok, lsym *win, lu8 *stage, u64 ring_lo, u32 lane,
	    u64 bpos_abs, u64 limit_abs, bool warm, u16 *__restrict__ sym,
	    u64 out0, u64 chunk_abs, u64 *bpos_ret, u64 *out_ret, u32 *bad_ret,
	    const u16 *__restrict__ hint = NULL)
{
	const u64 byte0 = bpos_abs >> 3;
	if (byte0 + 64 > in_n)
		return PAR_STOP;
	u32 *__restrict__ tokS = tok;
	const u32 cb = in_n - byte0 >= 64 * (PAR_CB / 8) ? PAR_CB : 256u;
	const u64 room = (in_n - byte0 + cb / 8 - 1) / (cb / 8);
	u32 NL = room < 64 ? (u32)room : 64;
	const u32 bpos0 = (u32)bpos_abs & 7;
	/* the limit, relative to the staged span; lanes whose piece starts at or
	 * after it have nothing to do */
	u32 lim = 0x7FFFFFFFu;
	if (limit_abs - 8 * byte0 < 0x40000000ull) {
		lim = (u32)(limit_abs - 8 * byte0);
		const u32 need = lim > bpos0 ? (lim - bpos0 + cb - 1) / cb : 1;
		NL = need < NL ? need : NL;
	}
	{
		const u32 nw = (NL * (cb / 8) + 80) / 8;
		for (u32 w = lane; w < nw; w += 64) {
			const u64 pos = byte0 + 8 * w;
			*(lu64 *)(stage + 8 * w) = pos + 8 <= in_n ? ld8(inp + pos) :
						  load_in(inp, in_n, pos);
		}
		global_stores_visible();	/* as par_round(): the stores of the rounds before */
	}
	const lu8 *span = stage;
	struct par_long pll, plo;
	par_long_init<LIT_TB + 1>(&pll, &S->lit, LIT_TB + 1);
	par_long_init<OFF_TB + 1>(&plo, &S->off, OFF_TB + 1);
	u32 cend = bpos0 + (lane + 1) * cb;
	cend = cend < lim ? cend : lim;
	u32 start = bpos0 + lane * cb, end = 0;
	/* `hint`: token boundaries the count pass met near the lanes' pieces, in
	 * bits from the round's first (phase_count(): a chunk of a block of one
	 * codeword length).  A lane that starts at one need not guess - and where
	 * guesses do not fall in step that is ten parses of the round's twelve.
	 * Only where the passes start: what they accept is decided as ever. */
	if (hint != NULL && lane != 0) {
		const u32 h = hint[lane];
		if (h != 0xFFFFu && bpos0 + h < lim)
			start = bpos0 + h;
	}
	u32 nbytes = 0, ntok = 0;
	bool eob = false, dirty = lane < NL;
	u32 K = NL - 1;
	bool has_eob = false;
	u32 nphase = 0;

	for (u32 pass = 0; pass < 64; pass++) {
		struct par_bits b;
		bool run = dirty;
		const bool keep = MODE == SM_MARK && (pass != 0 || lane == 0);
		pb_init(&b, span, start);
		if (dirty) {
			nbytes = 0;
			ntok = 0;
			eob = false;
		}
		while (__ballot(run)) {
			run = run && PB_POS(b) < cend;
			pb_refill(&b, span);
			struct par_token t = par_decode(S, SH, &pll, &plo, b.buf);
			const u32 e1 = t.e1;
			const bool two = t.kind == K_LIT && PB_POS(b) + t.used < cend &&
					 (e1 & 0xC000) == K_LIT && (e1 & 15) != 0;
			if (run) {
				u32 used = t.used;
				if (t.kind == K_EOB) {
					if (!warm) {
						eob = true;
						run = false;
					}
				} else {
					if (keep && ntok < PAR_LANECAP)
						tokS[TOK_AT(ntok, lane)] = t.kind == K_LEN ?
							0x80000000u | t.length | (t.dist << 9) : t.lit;
					nbytes += t.kind == K_LEN ? t.length : 1;
					ntok++;
					if (two) {
						if (keep && ntok < PAR_LANECAP)
							tokS[TOK_AT(ntok, lane)] = (e1 >> 4) & 0xFF;
						nbytes++;
						ntok++;
						used += e1 & 15;
					}
				}
				b.buf >>= used;
				b.cnt -= used;
			}
		}
		if (dirty)
			end = PB_POS(b);
		u32 ns = __builtin_amdgcn_update_dpp(end, end, 0x138, 0xF, 0xF, false);
		if (lane == 0)
			ns = bpos0;
		dirty = (ns != start || (pass == 0 && lane != 0)) && lane < NL;
		const u64 dm = __ballot(dirty), em = __ballot(eob);
		const u64 exact = dm ? (1ull << __builtin_ctzll(dm)) - 1 : ~0ull;
		if (em & exact) {
			K = (u32)__builtin_ctzll(em & exact);
			has_eob = true;
			break;
		}
		if (!dm)
			break;
		if (pass >= 1 && nphase < 6 && (u32)__builtin_popcountll(dm) >= PAR_PHASE_MIN) {
			/* the passes are not converging (a code of nearly one codeword
			 * length): par_phase_starts() of inflate_kernel.hip.  Here it
			 * runs again where its chain broke (a match across a piece's
			 * start: the lane behind it starts beyond the phases) once
			 * that lane's start lies within them - a chunk is as slow as
			 * its slowest round, and a block over incompressible bytes
			 * has a match every few hundred literals: one pass per lane
			 * behind the first of them made such a chunk three times as
			 * slow as its neighbours. */
			const u32 f = (u32)__builtin_ctzll(dm);
			const u32 sf = bcast_lane(ns, f);
			if (sf - (bpos0 + f * cb) < PAR_PHASES) {
				const u32 g = par_phase_starts(S, SH, &pll, &plo, span, bpos0, cb, cend,
							       lane, NL, f, sf, ns);
				nphase++;
				if (lane > f && lane < NL) {
					ns = g;
					dirty = ns != start;
				}
			}
		}
		start = ns;
	}
	bool valid = lane <= K;
	u32 tcnt = valid ? ntok : 0;
	u32 tbase = wave_scan_incl(tcnt) - tcnt;
	{
		u64 vm = __ballot(valid);
		const u64 over = __ballot(lane <= K && ntok > PAR_LANECAP);
		if (over)
			vm &= (1ull << __builtin_ctzll(over)) - 1;
		const u32 nv = __builtin_popcountll(vm);
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
	const u64 end_bits = (u64)bcast_lane(end, K) + 8 * byte0;
	if (end_bits > 8 * in_n)
		return PAR_STOP;
	*bpos_ret = end_bits;
	*out_ret = out0 + total_bytes;
	if (MODE != SM_MARK || warm)
		return has_eob ? PAR_EOB : PAR_OK;

	lu8 *mk = stage + PAR_STAGE_BYTES;
	lu16 *tb = (lu16 *)(mk + 256);
	const u32 own_cnt = valid ? tcnt : 0;
	tb[lane] = (u16)tbase;
	wave_sync();
	{
		lu32 *tk = (lu32 *)stage;
		lu16 *R = (lu16 *)((lu32 *)stage + 256);
		u64 gbase = out0;
		u64 flushed = out0;
		u64 safe_hi = out0;	/* symbols below this were stored before a wait (stage_input) */
		u32 g = 0;
		bool bad = false;
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
			const bool fits = ti0 < total_tok && incl0 <= PAR_GBYTES;
			const u32 cnt = __builtin_popcountll(__ballot(fits));
			const u32 gtot = bcast_lane(incl0, cnt - 1);
			if (g + 4 * cnt < total_tok)
				tq_next = tok_fetch(tokS, mk, tb, tbase, own_cnt,
						    g + 4 * cnt, total_tok, lane);
			for (u32 b0 = lane; b0 < gtot; b0 += 64)
				R[b0] = 0;
			wave_sync();
			if (lane < cnt) {
				u32 o = incl0 - lsum;
#pragma unroll
				for (u32 j = 0; j < 4; j++) {
					tk[4 * lane + j] = tw4[j];
					if (len4[j])
						R[o] = (u16)(4 * lane + j + 1);
					o += len4[j];
				}
			}
			wave_sync();
			const u32 gb = (u32)gbase;
			u32 ring_rel = PAR_RW - gtot;
			if (gbase - ring_lo < ring_rel)
				ring_rel = (u32)(gbase - ring_lo);
			/* bytes of this chunk in front of the group (a source further
			 * back is in front of the chunk: a marker), and of the stream */
			const u64 in_chunk64 = gbase - chunk_abs;
			const u32 in_chunk = in_chunk64 < 0x10000 ? (u32)in_chunk64 : 0x10000u;
			const u32 in_stream = gbase < 0x10000 ? (u32)gbase : 0x10000u;
			enum { SB = 4 };
			u32 carry = 0;
			for (u32 s0 = 0; s0 < gtot; s0 += 64 * SB) {
				u32 own[SB], vfar[SB];
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					const u32 bi = s0 + 64 * k + lane;
					own[k] = bi < gtot ? R[bi] : 0;
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
					const u32 bi = s0 + 64 * k + lane;
					own[k] = bi < gtot ? tk[own[k] - 1] : 0;
				}
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					const u32 bi = s0 + 64 * k + lane, tw = own[k];
					const u32 dist = (tw >> 9) & 0xFFFF;
					const bool before = (tw >> 31) && dist > bi;
					const u32 back = dist - bi;	/* bytes in front of the group */
					const bool outside = before && back > in_chunk;
					const bool far = before && !outside && back > ring_rel;
					bad |= before && back > in_stream;
					vfar[k] = 0x10000;	/* not a symbol: no far source */
					if (outside)
						vfar[k] = 0x8000u | (32768u - (back - in_chunk));
					if (__ballot(far)) {
						/* a symbol stored in this round, after the last
						 * wait: see par_round() */
						if (__ballot(far && back <= (u32)(gbase - safe_hi))) {
							global_stores_visible();
							safe_hi = flushed;
						}
						if (far)
							vfar[k] = sym[gbase - back];
					}
				}
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
						bool ch = false;
#pragma unroll
						for (u32 k = 0; k < SB; k++) {
							const u32 pp = (u32)__builtin_amdgcn_ds_bpermute(
									(int)(root[k] << 2), (int)root[k]);
							ch |= pp != root[k];
							root[k] = pp;
						}
						if (!__ballot(ch))
							break;
					}
				}
#pragma unroll
				for (u32 k = 0; k < SB; k++) {
					const u32 bi = s0 + 64 * k + lane, tw = own[k];
					const u32 dist = (tw >> 9) & 0xFFFF;
					const bool match = (tw >> 31) != 0;
					const u32 wv = win[(gb + bi - dist) & (PAR_RW - 1)];
					u32 v = match ? wv : tw & 0xFF;
					v = vfar[k] < 0x10000 ? vfar[k] : v;
					if (__ballot(root[k] != lane))
						v = (u32)__builtin_amdgcn_ds_bpermute((int)(root[k] << 2), (int)v);
					if (bi < gtot)
						win[(gb + bi) & (PAR_RW - 1)] = (u16)v;
				}
			}
			wave_sync();
			gbase += gtot;
			flushed = flush_ring16(sym, win, flushed, gbase, lane);
			g += 4 * cnt;
		}
		if (flushed < gbase && lane == 0)
			sym[flushed] = win[(u32)flushed & (PAR_RW - 1)];
		wave_sync();
		if (__ballot(bad))
			*bad_ret = 1;
	}
	return has_eob ? PAR_EOB : PAR_OK;
}

/* stage `nbytes` (a multiple of 8) of input from byte0 on, zeros past the end */
static __device__ __forceinline__ void
stage_input(lu8 *stage, const u8 *inp, u64 in_n, u64 byte0, u32 nbytes, u32 lane)
{
	for (u32 w = lane; w < nbytes / 8; w += 64) {
		const u64 pos = byte0 + 8 * w;
		*(lu64 *)(stage + 8 * w) = pos + 8 <= in_n ? ld8(inp + pos) :
					  load_in(inp, in_n, pos);
	}
	/* ... and every store of the rounds before has completed (par_round()) */
	global_stores_visible();
}

/*
 * The code lengths of a dynamic block header (lib/decompress_template.h:
 * 150-245): ONE lane's loop, from the staged copy; b stands at the header's
 * three first bits (b->buf holds BFINAL / BTYPE / HLIT .. at its low end).
 * Leaves the lengths in S->lens[0 .. nlit + noff) and b behind the header.
 * false: a repeat without a previous length, or one that overruns the counts.
 */
static __device__ __forceinline__ bool
dynamic_lens(slds_t *S, const lu8 *stage, struct par_bits *bp, u32 *nlit_ret, u32 *noff_ret)
{
	struct par_bits b = *bp;
	u8 plens[19];
	const u32 nlit = 257 + (((u32)b.buf >> 3) & 31);
	const u32 noff = 1 + (((u32)b.buf >> 8) & 31);
	const u32 npre = 4 + (((u32)b.buf >> 13) & 15);
	for (u32 i = 0; i < 19; i++)
		plens[i] = 0;
	plens[c_pre_perm[0]] = ((u32)b.buf >> 17) & 7;
	b.buf >>= 20;
	b.cnt -= 20;
	pb_refill(&b, stage);
	for (u32 i = 1; i < npre; i++) {
		plens[c_pre_perm[i]] = (u32)b.buf & 7;
		b.buf >>= 3;
		b.cnt -= 3;
	}
	bool ok = build_precode(S->pre_tab, plens);
	u32 i = 0;
	const u32 total = nlit + noff;
	while (ok && i < total) {
		if (b.cnt < 14)
			pb_refill(&b, stage);
		const u32 e = S->pre_tab[(u32)b.buf & 127];
		b.buf >>= e & 15;
		b.cnt -= e & 15;
		const u32 presym = e >> 4;
		if (presym < 16) {
			S->lens[i++] = (u8)presym;
			continue;
		}
		u32 rep, val = 0;
		if (presym == 16) {
			if (i == 0) {
				ok = false;
				break;
			}
			val = S->lens[i - 1];
			rep = 3 + ((u32)b.buf & 3);
			b.buf >>= 2;
			b.cnt -= 2;
		} else if (presym == 17) {
			rep = 3 + ((u32)b.buf & 7);
			b.buf >>= 3;
			b.cnt -= 3;
		} else {
			rep = 11 + ((u32)b.buf & 127);
			b.buf >>= 7;
			b.cnt -= 7;
		}
		for (u32 k = 0; k < rep; k++)
			S->lens[i + k] = (u8)val;
		i += rep;
	}
	*bp = b;
	*nlit_ret = nlit;
	*noff_ret = noff;
	return ok && i == total;
}

/* a header the host had parsed ahead of the chunk kernels (lda_stream_hdr_cache_kernel):
 * its code lengths and {HLIT + 257, HDIST + 1, bits up to the first token, BFINAL} */
#define HDRC_LENS 320u
struct hdr_cached {
	const u8 *lens;		/* NULL: none, the chunk parses the header itself */
	const u32 *info;
};

/*
 * A block header at bit `pos` (lib/decompress_template.h:72-245, :313-326):
 * lane 0 parses it from a staged copy, all lanes build the tables.
 * Returns 0: Huffman block, tables ready, *pos_ret = first token;
 *         1: stored block, *pos_ret = bit after the 3 header bits;
 *         2: not a valid header.
 */
#define HDR_STAGE 704u	/* 17 + 57 + 316 x 14 bits at most, + slack */
static __device__ u32
chunk_header(const u8 *inp, u64 in_n, slds_t *S, lu8 *stage, u32 lane, u64 pos,
	     u32 *final_ret, u64 *pos_ret, bool *static_ret,
	     struct hdr_cached hc = { NULL, NULL })
{
	/* pos == LDA_HDR_STATIC: no header to read, the static codes' tables */
	const bool forced = pos == LDA_HDR_STATIC;
	/* a dynamic header parsed ahead: its lengths come from memory (the parse
	 * is one lane's loop of ~300 dependent LDS round trips: 40 us of every
	 * chunk of the count pass and of the decode pass, and a block's chunks -
	 * two dozen - all did it) */
	const bool cached = !forced && hc.lens != NULL && hc.info[0] != 0;
	if (forced)
		pos = 0;
	else if (!cached)
		stage_input(stage, inp, in_n, pos >> 3, HDR_STAGE, lane);
	u32 r = 2, fin = 0, nlit = 0, noff = 0, used = 0, stat = 0;
	if (cached) {
		nlit = hc.info[0];
		noff = hc.info[1];
		used = hc.info[2];
		fin = hc.info[3];
		r = 0;
		for (u32 w = lane; w < HDRC_LENS / 4; w += 64)
			((lu32 *)S->lens)[w] = ((const u32 *)hc.lens)[w];
	} else {
	if (lane == 0) {
		struct par_bits b;
		const u32 p0 = (u32)pos & 7;
		if (!forced)
			pb_init(&b, stage, p0);
		else
			b.buf = 2;	/* BTYPE 01, not final, nothing consumed */
		fin = (u32)b.buf & 1;
		const u32 btype = ((u32)b.buf >> 1) & 3;
		stat = btype == 1;
		if (btype == 0) {
			r = 1;
			used = 3;
		} else if (btype == 1) {
			for (u32 i = 0; i < 320; i++)
				S->lens[i] = i < 144 ? 8 : i < 256 ? 9 :
					     i < 280 ? 7 : i < 288 ? 8 : 5;
			nlit = 288;
			noff = 32;
			r = 0;
			used = forced ? 0 : 3;
		} else if (btype == 2) {
			if (dynamic_lens(S, stage, &b, &nlit, &noff)) {
				r = 0;
				used = PB_POS(b) - p0;
			}
		}
	}
	}
	r = bcast_first(r);
	fin = bcast_first(fin);
	nlit = bcast_first(nlit);
	noff = bcast_first(noff);
	used = bcast_first(used);
	*static_ret = bcast_first(stat) != 0;
	wave_sync();
	if (r == 0) {
		u32 s_lit, s_off;
		bool ok = build_table_coop(S->lens + nlit, noff, OFF_TB, false, &S->off,
					   S->off_sorted, lane, &s_off);
		ok = build_table_coop(S->lens, nlit, LIT_TB, true, &S->lit,
				      S->lit_sorted, lane, &s_lit) && ok;
		wave_sync();
		if (ok) {
			fill_table(S->off_tab, OFF_TB, false, &S->off, S->off_sorted, s_off, lane);
			fill_table(S->lit_tab, LIT_TB, true, &S->lit, S->lit_sorted, s_lit, lane);
		} else {
			r = 2;
		}
		wave_sync();
	}
	if (!forced && pos + used > 8 * in_n)
		r = 2;
	*final_ret = fin;
	*pos_ret = pos + used;
	return r;
}

/*
 * Tokens one at a time (lane 0 decodes, the wave copies): what is left of a
 * stream when fewer than 64 bytes of input remain, and whatever a round
 * could not take.  Decodes from a staged window until the end of the block
 * (PAR_EOB), the limit or the end of the window (PAR_OK: the caller goes on),
 * or an error (PAR_STOP).
 */
template <int MODE> static __device__ u32
chunk_seq(const u8 *inp, u64 in_n, const slds_t *S, lu8 *stage, u32 lane,
	  u64 pos, u64 limit, u16 *__restrict__ sym, u64 out0, u64 chunk_abs,
	  u64 *pos_ret, u64 *out_ret, u32 *bad_ret)
{
	const u64 byte0 = pos >> 3;
	const u32 winbytes = 1024;
	stage_input(stage, inp, in_n, byte0, winbytes, lane);
	struct par_bits b;
	pb_init(&b, stage, (u32)pos & 7);
	u64 out = out0;
	u32 ret = PAR_OK;
	for (;;) {
		/* lane 0's token, wave-uniform */
		u32 kind = 0, lit = 0, length = 0, dist = 0, stop = 0;
		if (lane == 0) {
			if (PB_POS(b) > 8 * (winbytes - 32) || 8 * byte0 + PB_POS(b) >= limit) {
				stop = 1;
			} else {
				pb_refill(&b, stage);
				u32 e = S->lit_tab[(u32)b.buf & ((1u << LIT_TB) - 1)];
				u32 cl = e & 15, pay = (e >> 4) & 0x3FF;
				kind = e & 0xC000;
				if (cl == 0) {
					u32 sy = decode_long(&S->lit, S->lit_sorted, b.buf, &cl);
					kind = sy < 256 ? K_LIT : sy == 256 ? K_EOB : K_LEN;
					pay = sy < 256 ? sy : sy - 257;
				}
				b.buf >>= cl;
				b.cnt -= cl;
				lit = pay & 0xFF;
				if (kind == K_LEN) {
					u32 base, xb;
					len_sym(pay & 31, &base, &xb);
					length = base + ((u32)b.buf & ((1u << xb) - 1));
					b.buf >>= xb;
					b.cnt -= xb;
					u32 e2 = S->off_tab[(u32)b.buf & ((1u << OFF_TB) - 1)];
					u32 ol = e2 & 15, osym = e2 >> 4;
					if (ol == 0)
						osym = decode_long(&S->off, S->off_sorted, b.buf, &ol);
					b.buf >>= ol;
					b.cnt -= ol;
					off_sym(osym & 31, &base, &xb);
					dist = base + ((u32)b.buf & ((1u << xb) - 1));
					b.buf >>= xb;
					b.cnt -= xb;
				}
				if (8 * byte0 + PB_POS(b) > 8 * in_n)
					stop = 2;	/* the token runs past the input */
			}
		}
		stop = bcast_first(stop);
		if (stop == 2) {
			ret = PAR_STOP;
			break;
		}
		if (stop)
			break;
		kind = bcast_first(kind);
		if (kind == K_EOB) {
			ret = PAR_EOB;
			break;
		}
		if (kind == K_LIT) {
			if (MODE == SM_MARK && lane == 0)
				sym[out] = (u16)lit;
			out++;
			continue;
		}
		length = bcast_first(length);
		dist = bcast_first(dist);
		if (MODE == SM_MARK && dist > out) {
			*bad_ret = 1;
		} else if (MODE == SM_MARK) {
			__threadfence();	/* the symbols stored so far, by any lane */
			/* byte k of the match is the byte (k mod dist) of the `dist`
			 * bytes in front of it: all sources are older than the match */
			const u64 in_chunk = out - chunk_abs;
			for (u32 k0 = 0; k0 < length; k0 += 64) {
				const u32 k = k0 + lane;
				if (k < length) {
					const u32 back = dist - k % dist;	/* 1..dist */
					u16 v;
					if (back > in_chunk)
						v = (u16)(0x8000u | (32768u - (back - (u32)in_chunk)));
					else
						v = sym[out - back];
					sym[out + k] = v;
				}
			}
		}
		out += length;
	}
	*pos_ret = 8 * byte0 + bcast_first(PB_POS(b));
	*out_ret = out;
	if (MODE == SM_MARK)
		__threadfence();
	return ret;
}

static __device__ __forceinline__ struct hdr_cached
hdr_of(const struct lda_stream_chunk *cd, const u8 *hdr_lens, const u32 *hdr_info)
{
	const u32 c = cd->hdr_cache;	/* slot + 1 of the header at cd->hdr_bit; 0: none */
	struct hdr_cached h = { NULL, NULL };
	if (c && hdr_lens) {
		h.lens = hdr_lens + (size_t)(c - 1) * HDRC_LENS;
		h.info = hdr_info + 4 * (size_t)(c - 1);
	}
	return h;
}

/* one chunk on one wave */
template <int MODE> static __device__ void
chunk_run(const struct lda_stream_chunk *__restrict__ cd,
	  struct lda_stream_res *__restrict__ rs, const u8 *__restrict__ inp,
	  u64 in_n, u16 *__restrict__ sym, u32 *__restrict__ tok,
	  const u8 *__restrict__ hdr_lens, const u32 *__restrict__ hdr_info,
	  const u16 *__restrict__ hints = NULL)
{
	const u32 lane = threadIdx.x;
	slds_t *S = (slds_t *)(lu8 *)(uintptr_t)0;
	shlds_t *SH = (shlds_t *)(S + 1);
	lsym *win = (lsym *)(SH + 1);
	lu8 *stage = (lu8 *)(win + PAR_RW);
	const u32 kind = cd->kind;
	const u64 limit = cd->limit_bit;
	const u64 chunk_abs = MODE == SM_MARK ? cd->out_off : 0;
	u64 hdr = cd->hdr_bit, pos = hdr, out = chunk_abs;
	u64 start_exact = cd->start_bit;
	u32 final_blk = 0, status = LDA_STREAM_OK, bad = 0, at_boundary = 0;
	bool in_block = false, first = cd->kind == LDA_CHUNK_HEADER;	/* its own header is not a stop */
	bool gov_static = false;	/* the block pos lies in is a static one */
	/* a chunk planned under the static codes stops at its block's end: it
	 * cannot know whether that block is the stream's last (stream_kernels.h) */
	const bool stop_at_eob = kind != LDA_CHUNK_HEADER && hdr == LDA_HDR_STATIC;
	u64 ring_lo = out;

	if (kind != LDA_CHUNK_HEADER) {
		/* inside a block: its tables, then the start */
		u64 p2;
		const u32 r = chunk_header(inp, in_n, S, stage, lane, hdr, &final_blk, &p2, &gov_static,
					   hdr_of(cd, hdr_lens, hdr_info));
		if (r != 0 || (!stop_at_eob && cd->start_bit < p2)) {
			status = LDA_STREAM_ERR;
		} else {
			pos = cd->start_bit;
			in_block = true;
			if (kind == LDA_CHUNK_WARM) {
				/* the first token boundary at or after target_bit */
				const u64 target = cd->target_bit;
				while (pos < target && status == LDA_STREAM_OK) {
					u64 np = pos, no = 0;
					const u32 pr = chunk_round<SM_COUNT>(
						inp, in_n, S, SH, tok, win, stage, 0, lane, pos,
						target, true, sym, 0, 0, &np, &no, &bad);
					if (pr == PAR_STOP || np <= pos)
						status = LDA_STREAM_ERR;
					pos = np;
				}
				start_exact = pos;
			}
		}
	}
	while (status == LDA_STREAM_OK) {
		if (!in_block) {
			if (pos >= limit && !first) {
				/* (also in the middle of a run of stored blocks: the host
				 * walks such a run itself - it has the input - and gives
				 * every stored block a chunk of its own, host_stream.hip) */
				at_boundary = 1;
				break;
			}
			const bool own = first && pos == cd->hdr_bit;
			first = false;
			u64 p2;
			hdr = pos;
			/* (the chunk's own header may have been parsed ahead; what it
			 * walks into behind it was not) */
			const u32 r = chunk_header(inp, in_n, S, stage, lane, pos, &final_blk, &p2,
						   &gov_static,
						   own ? hdr_of(cd, hdr_lens, hdr_info) : hdr_cached{ NULL, NULL });
			if (r == 2) {
				status = LDA_STREAM_ERR;
				break;
			}
			if (r == 1) {
				/* stored: lib/decompress_template.h:247-285 */
				const u64 bp = (p2 + 7) >> 3;
				if (bp + 4 > in_n) {
					status = LDA_STREAM_ERR;
					break;
				}
				const u32 len = inp[bp] | ((u32)inp[bp + 1] << 8);
				const u32 nlen = inp[bp + 2] | ((u32)inp[bp + 3] << 8);
				if (len != (nlen ^ 0xFFFFu) || len > in_n - (bp + 4)) {
					status = LDA_STREAM_ERR;
					break;
				}
				if (MODE == SM_MARK) {
					/* 8 bytes per lane and step (a stream of stored blocks
					 * is copied at the rate of this loop) */
					const u8 *src = inp + bp + 4;
					u16 *dst = sym + out;
					for (u32 k = 8 * lane; k < len; k += 512) {
						if (k + 8 <= len) {
							const u64 v = ld8(src + k);
#pragma unroll
							for (u32 j = 0; j < 8; j++)
								dst[k + j] = (u16)((v >> (8 * j)) & 0xFF);
						} else {
							for (u32 j = k; j < len; j++)
								dst[j] = src[j];
						}
					}
					__threadfence();
				}
				out += len;
				ring_lo = out;
				pos = 8 * (bp + 4 + len);
				if (final_blk) {
					status = LDA_STREAM_FINAL;
					break;
				}
				continue;
			}
			pos = p2;
			in_block = true;
		}
		/* tokens of the block, up to its end or the limit */
		if (pos >= limit)
			break;
		u64 np = pos, no = out;
		u32 pr = chunk_round<MODE>(inp, in_n, S, SH, tok, win, stage, ring_lo, lane,
					   pos, limit, false, sym, out, chunk_abs, &np, &no, &bad,
					   MODE == SM_MARK && hints != NULL && cd->hint &&
					   pos == cd->start_bit ? hints + (size_t)(cd->hint - 1) * 64 : NULL);
		if (pr == PAR_STOP) {
			pr = chunk_seq<MODE>(inp, in_n, S, stage, lane, pos, limit, sym, out,
					     chunk_abs, &np, &no, &bad);
			ring_lo = no;	/* the mirror does not hold these bytes */
			if (pr == PAR_STOP || (pr == PAR_OK && np <= pos && np < limit)) {
				status = LDA_STREAM_ERR;
				break;
			}
		}
		pos = np;
		out = no;
		if (pr == PAR_EOB) {
			in_block = false;
			if (stop_at_eob) {	/* the host knows what follows */
				at_boundary = 1;
				break;
			}
			if (final_blk) {
				status = LDA_STREAM_FINAL;
				at_boundary = 1;
				break;
			}
		}
	}
	if (lane == 0) {
		rs->start_bit = start_exact;
		rs->end_bit = pos;
		rs->end_hdr_bit = in_block ? (gov_static ? LDA_HDR_STATIC : hdr) : pos;
		rs->nout = out - chunk_abs;
		rs->status = status;
		rs->flags = (at_boundary || !in_block ? LDA_RES_BOUNDARY : 0) |
			    (bad ? LDA_RES_BAD_DIST : 0) |
			    (in_block && final_blk && !stop_at_eob ? LDA_RES_GOV_FINAL : 0);
	}
}

/*
 * K EXACT chunks that start at consecutive bits P, P + 1, .. P + K - 1 of one
 * block and end at the same limit, counted TOGETHER by one wave (the host
 * plans them where no parse falls in step: a block of one codeword length,
 * host_stream.hip).  The true parse crosses P at one of those bits; as K
 * separate chunks every one would parse the same input, ten times over.  Here
 * every lane parses its piece of the input once from each of its first K bits
 * (as par_phase_starts() does for a round that does not converge) and keeps
 * where the parse leaves the piece and how many bytes it makes; then lane j
 * follows the chain "my end is the next piece's start" from bit j of piece 0
 * through the pieces, which is chunk j's answer: K answers for K parses per
 * piece instead of K x (K + 3).  A chain that enters a piece behind its first
 * K bits (a match across the piece's start) has the rest of that piece parsed
 * for it.  A chain that meets an end of block in front of the limit reports
 * LDA_STREAM_ERR: the host's chain treats that start as not counted and
 * asks for it on its own if it ever arrives there.  What is
 * reported as counted is exactly what chunk_run() would report - the decode
 * pass runs chunk_run() on every accepted chunk and the host compares.
 * false: not taken (a static block, more than one round of input, the last
 * bytes of the stream): the caller counts the K chunks one after the other.
 */
#define PHASE_MAX 12u
static __device__ bool
phase_count(const struct lda_stream_chunk *__restrict__ cd, struct lda_stream_res *__restrict__ rs,
	    u32 K, const u8 *__restrict__ inp, u64 in_n,
	    const u8 *__restrict__ hdr_lens, const u32 *__restrict__ hdr_info,
	    u16 *__restrict__ hints /* [K][64] or NULL */)
{
	const u32 lane = threadIdx.x;
	slds_t *S = (slds_t *)(lu8 *)(uintptr_t)0;
	shlds_t *SH = (shlds_t *)(S + 1);
	lu8 *scr = (lu8 *)(SH + 1);		/* the mirror's LDS: nothing is executed here */
	lu8 *stage = scr + 2 * PAR_RW;
	lu32 *nb = (lu32 *)scr;			/* [PHASE_MAX][64] bytes a piece makes from a start */
	lu32 *endpos = nb + PHASE_MAX * 64;	/* [PHASE_MAX][64] where a chain ends (bit 31: at an end of block) */
	lu8 *code = (lu8 *)(endpos + PHASE_MAX * 64);	/* [PHASE_MAX][64] 0 goes on at endpos, 62 ends there, 63 fails */
	static_assert(PHASE_MAX * 64 * 9 <= 2 * PAR_RW, "the tables fit the mirror");
	const u64 hdr = cd->hdr_bit, P = cd->start_bit, limit = cd->limit_bit;
	const u64 byte0 = P >> 3;
	const u32 bpos0 = (u32)P & 7, cb = PAR_CB;
	if (K < 1 || K > PHASE_MAX || hdr == LDA_HDR_STATIC || limit <= P + K ||
	    limit - 8 * byte0 > 64 * cb - 64 || in_n < byte0 + 64 * (PAR_CB / 8))
		return false;
	u32 final_blk = 0;
	u64 p2;
	bool gov_static = false;
	if (chunk_header(inp, in_n, S, stage, lane, hdr, &final_blk, &p2, &gov_static,
			 hdr_of(cd, hdr_lens, hdr_info)) != 0 || P < p2)
		return false;
	const u32 lim = (u32)(limit - 8 * byte0);
	const u32 NL = (lim - bpos0 + cb - 1) / cb;	/* 1 .. 64 */
	stage_input(stage, inp, in_n, byte0, PAR_SPAN & ~7u, lane);
	struct par_long pll, plo;
	par_long_init<LIT_TB + 1>(&pll, &S->lit, LIT_TB + 1);
	par_long_init<OFF_TB + 1>(&plo, &S->off, OFF_TB + 1);
	const u32 ps = bpos0 + lane * cb, pe = ps + cb;
	const u32 cend = pe < lim ? pe : lim;	/* the last piece ends at the limit */
	const bool mine = lane < NL;
	for (u32 ph = 0; ph < K; ph++) {
		struct par_bits b;
		u32 nbytes = 0;
		bool eob = false;
		pb_init(&b, stage, ps + ph);
		bool run = mine && PB_POS(b) < cend;
		while (__ballot(run)) {
			pb_refill(&b, stage);
			const struct par_token t = par_decode(S, SH, &pll, &plo, b.buf, run);
			const u32 e1 = t.e1;
			const bool two = t.kind == K_LIT && PB_POS(b) + t.used < cend &&
					 (e1 & 0xC000) == K_LIT && (e1 & 15) != 0;
			if (run) {
				u32 used = t.used;
				if (t.kind == K_EOB) {
					eob = true;
					run = false;
				} else {
					nbytes += t.kind == K_LEN ? t.length : 1;
					if (two) {
						nbytes++;
						used += e1 & 15;
					}
				}
				b.buf >>= used;
				b.cnt -= used;
			}
			run = run && PB_POS(b) < cend;
		}
		const u32 pos = PB_POS(b);
		u32 c = 0;	/* the chain goes on at `pos` */
		if (eob) {
			/* the chunk ends with its block when that is the stream's last or
			 * ends at or behind the limit; else it would go on into the next */
			c = final_blk || pos >= lim ? 62 : 63;
		} else if (lane == NL - 1) {
			c = 62;
		}
		if (mine) {
			nb[ph * 64 + lane] = nbytes;
			endpos[ph * 64 + lane] = pos | (eob ? 0x80000000u : 0);
			code[ph * 64 + lane] = (u8)c;
		}
	}
	wave_sync();
	/* lane j follows chunk j's parse through the table; where it enters a
	 * piece behind its first K bits (a match across the piece's start) the
	 * rest of that piece is parsed for it - all lanes that are in that
	 * position at once, so the passes are as many as the longest run of such
	 * entries on one chain */
	bool active = lane < K;
	u32 pos = bpos0 + lane, st = LDA_STREAM_ERR, endv = 0;
	u64 total = 0;
	for (u32 guard = 0; guard < 64 && __ballot(active); guard++) {
		bool need = false, last = false;
		u32 pcend = 0;
		if (active) {
			for (;;) {
				const u32 i = (pos - bpos0) / cb, o = pos - bpos0 - i * cb;
				if (i >= NL) {
					active = false;
					break;
				}
				/* (for the decode pass: where this chunk's parse enters
				 * piece i, in bits from the chunk's start) */
				if (hints != NULL)
					hints[lane * 64 + i] = (u16)(pos - bpos0 - lane);
				if (o >= K) {
					need = true;
					last = i == NL - 1;
					pcend = bpos0 + (i + 1) * cb;
					pcend = pcend < lim ? pcend : lim;
					break;
				}
				const u32 c = code[o * 64 + i], e = endpos[o * 64 + i];
				total += nb[o * 64 + i];
				if (c == 62) {
					endv = e;
					st = LDA_STREAM_OK;
				}
				if (c >= 62) {
					active = false;
					break;
				}
				pos = e;
			}
		}
		if (!__ballot(need))
			break;
		struct par_bits b;
		u32 nbytes = 0;
		bool eob = false;
		pb_init(&b, stage, need ? pos : 0);
		bool run = need && PB_POS(b) < pcend;
		while (__ballot(run)) {
			pb_refill(&b, stage);
			const struct par_token t = par_decode(S, SH, &pll, &plo, b.buf, run);
			const u32 e1 = t.e1;
			const bool two = t.kind == K_LIT && PB_POS(b) + t.used < pcend &&
					 (e1 & 0xC000) == K_LIT && (e1 & 15) != 0;
			if (run) {
				u32 used = t.used;
				if (t.kind == K_EOB) {
					eob = true;
					run = false;
				} else {
					nbytes += t.kind == K_LEN ? t.length : 1;
					if (two) {
						nbytes++;
						used += e1 & 15;
					}
				}
				b.buf >>= used;
				b.cnt -= used;
			}
			run = run && PB_POS(b) < pcend;
		}
		if (need) {
			const u32 np = PB_POS(b);
			total += nbytes;
			if (eob ? final_blk || np >= lim : last) {
				endv = np | (eob ? 0x80000000u : 0);
				st = LDA_STREAM_OK;
				active = false;
			} else if (eob) {
				active = false;
			} else {
				pos = np;
			}
		}
	}
	if (lane < K) {
		const bool eobt = endv >> 31;
		const u64 end_bit = 8 * byte0 + (endv & 0x7FFFFFFFu);
		if (st == LDA_STREAM_OK && end_bit > 8 * in_n)
			st = LDA_STREAM_ERR;
		struct lda_stream_res *r = rs + lane;
		r->start_bit = P + lane;
		r->end_bit = st == LDA_STREAM_OK ? end_bit : P + lane;
		r->end_hdr_bit = st != LDA_STREAM_OK ? hdr : eobt ? end_bit : hdr;
		r->nout = st == LDA_STREAM_OK ? total : 0;
		r->status = st != LDA_STREAM_OK ? LDA_STREAM_ERR :
			    eobt && final_blk ? LDA_STREAM_FINAL : LDA_STREAM_OK;
		r->flags = st != LDA_STREAM_OK ? 0 :
			   (eobt ? LDA_RES_BOUNDARY : 0) | (!eobt && final_blk ? LDA_RES_GOV_FINAL : 0);
	}
	return true;
}

extern "C" __global__ void __launch_bounds__(64)
lda_stream_count_kernel(u32 nchunks, const struct lda_stream_chunk *chunks,
			struct lda_stream_res *res, const u8 *inp, u64 in_n,
			u32 *tokscratch, const u8 *hdr_lens, const u32 *hdr_info, u16 *hints)
{
	if (blockIdx.x >= nchunks)
		return;
	/* `phases`: 0 a chunk of its own; K the first of K exact starts counted
	 * together (phase_count()); ~0 one of the K - 1 behind it */
	const u32 K = chunks[blockIdx.x].phases;
	if (K == ~0u)
		return;
	if (K != 0 && K <= nchunks - blockIdx.x) {
		u16 *hrow = hints ? hints + (size_t)blockIdx.x * 64 : NULL;
		if (hrow != NULL) {
			for (u32 r = 0; r < K; r++)
				hrow[r * 64 + threadIdx.x] = 0xFFFFu;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		if (phase_count(chunks + blockIdx.x, res + blockIdx.x, K, inp, in_n, hdr_lens, hdr_info,
				hrow))
			return;
		for (u32 j = 0; j < K; j++)
			chunk_run<SM_COUNT>(chunks + blockIdx.x + j, res + blockIdx.x + j, inp, in_n,
					    NULL, tokscratch, hdr_lens, hdr_info);
		return;
	}
	chunk_run<SM_COUNT>(chunks + blockIdx.x, res + blockIdx.x, inp, in_n, NULL,
			    tokscratch, hdr_lens, hdr_info);	/* (keeps no tokens) */
}

extern "C" __global__ void __launch_bounds__(64)
lda_stream_decode_kernel(u32 nchunks, const struct lda_stream_chunk *chunks,
			 struct lda_stream_res *res, const u8 *inp, u64 in_n,
			 u16 *sym, u32 *tokscratch, const u8 *hdr_lens, const u32 *hdr_info,
			 const u16 *hints)
{
	if (blockIdx.x < nchunks)
		chunk_run<SM_MARK>(chunks + blockIdx.x, res + blockIdx.x, inp, in_n, sym,
				   tokscratch + (size_t)blockIdx.x * PAR_SCRATCH, hdr_lens, hdr_info,
				   hints);
}

/*
 * The headers the finder accepted, parsed ONCE each (a wave per header: lane 0
 * decodes the code lengths, all lanes store them) beside the host's planning,
 * for every chunk that starts at or inside the block: slot i belongs to
 * cand[i].  A header this kernel cannot parse leaves HLIT + 257 = 0 in its
 * slot: the chunks parse it themselves and say what is wrong with it.
 */
extern "C" __global__ void __launch_bounds__(64)
lda_stream_hdr_cache_kernel(const u8 *__restrict__ inp, u64 in_n,
			    const u64 *__restrict__ cand, const u32 *__restrict__ ncand,
			    u32 nslots, u8 *__restrict__ hdr_lens, u32 *__restrict__ hdr_info)
{
	const u32 lane = threadIdx.x;
	slds_t *S = (slds_t *)(lu8 *)(uintptr_t)0;
	lu8 *stage = (lu8 *)(S + 1);
	u32 nc = *ncand;
	nc = nc < nslots ? nc : nslots;
	for (u32 idx = blockIdx.x; idx < nc; idx += gridDim.x) {
		const u64 p = cand[idx];
		wave_sync();	/* (the LDS of the header before) */
		stage_input(stage, inp, in_n, p >> 3, HDR_STAGE, lane);
		u32 ok = 0, nlit = 0, noff = 0, used = 0, fin = 0;
		if (lane == 0) {
			struct par_bits b;
			const u32 p0 = (u32)p & 7;
			pb_init(&b, stage, p0);
			fin = (u32)b.buf & 1;
			if ((((u32)b.buf >> 1) & 3) == 2 && dynamic_lens(S, stage, &b, &nlit, &noff)) {
				ok = 1;
				used = PB_POS(b) - p0;
			}
		}
		ok = bcast_first(ok);
		nlit = bcast_first(nlit);
		noff = bcast_first(noff);
		used = bcast_first(used);
		fin = bcast_first(fin);
		wave_sync();
		if (!ok || p + used > 8 * in_n)
			nlit = 0;
		for (u32 w = lane; w < HDRC_LENS / 4; w += 64)
			((u32 *)(hdr_lens + (size_t)idx * HDRC_LENS))[w] = ((const lu32 *)S->lens)[w];
		if (lane == 0) {
			u32 *inf = hdr_info + 4 * (size_t)idx;
			inf[0] = nlit;
			inf[1] = noff;
			inf[2] = used;
			inf[3] = fin;
		}
	}
}

extern "C" size_t lda_stream_hdr_cache_lds(void)
{
	return sizeof(struct stream_lds) + HDR_STAGE;
}

extern "C" size_t lda_stream_chunk_lds(void)
{
	return STREAM_LDS;
}

extern "C" size_t lda_stream_tokcap(void)
{
	return PAR_SCRATCH;
}

extern "C" size_t lda_stream_find_b_lds(void)
{
	return FIND_B_LDS;
}

/* ---------------- the block finder ---------------- */

/* 64 bits of input at bit position p (zeros past the end) and the 64 after */
static __device__ __forceinline__ void
bits128(const u8 *inp, u64 in_n, u64 p, u64 *x0, u64 *x1)
{
	const u64 lo = load_in(inp, in_n, p >> 3), hi = load_in(inp, in_n, (p >> 3) + 8);
	const u32 sh = (u32)p & 7;
	*x0 = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
	*x1 = hi >> sh;
}

/*
 * Stage A: BTYPE = dynamic, HLIT <= 29, HDIST <= 29, and the precode lengths
 * form a COMPLETE code (the compressors emit nothing else; incomplete ones
 * are legal, rare, and found by the chain as ordinary misses).  About one
 * offset in five hundred passes on compressed data.  A wave takes 64 bytes =
 * 512 bit offsets: every lane tests the eight offsets of its byte for the
 * cheap part (one in nine passes), the survivors are queued in LDS and the
 * Kraft sum of the precode is then evaluated with all lanes busy.
 * (a launch covers at most 2^31 offsets from bit0 on: the threads of a grid
 * are counted in 32 bits)
 */
static __device__ __forceinline__ bool precode_complete(u64 x0, u64 x1)
{
	const u32 h = (u32)x0;
	const u32 npre = 4 + ((h >> 13) & 15);
	u64 f = (x0 >> 17) | (x1 << 47);
	f &= (1ull << (3 * npre)) - 1;
	u32 kraft = 0;
#pragma unroll
	for (u32 i = 0; i < 19; i++)
		kraft += (128u >> ((u32)(f >> (3 * i)) & 7)) & 127;
	return kraft == 128;
}

#define FIND_A_WINDOWS 16	/* 2048-offset windows per workgroup */
extern "C" __global__ void __launch_bounds__(256)
lda_stream_find_a_kernel(const u8 *__restrict__ inp, u64 in_n, u64 bit0, u64 nbits,
			 u64 *__restrict__ queue, u32 *__restrict__ qcount, u32 qcap)
{
	__shared__ u16 wq[4][512];
	__shared__ u64 found[512];	/* the workgroup's survivors: ONE atomic on the global counter */
	__shared__ u32 nfound, gbase;
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0)
		nfound = 0;
	__syncthreads();
	for (u32 win = 0; win < FIND_A_WINDOWS; win++) {
		const u64 byte = (bit0 >> 3) +
				 (((u64)blockIdx.x * FIND_A_WINDOWS + win) * 4 + wave) * 64 + lane;
		if (8 * (byte - lane) >= nbits)
			break;		/* (uniform per wave) */
		const u64 w0 = load_in(inp, in_n, byte);	/* the 17 header bits of its 8 offsets */
		u32 nq = 0;
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			const u32 h = (u32)(w0 >> k);	/* k + 17 <= 64 */
			const bool ok = 8 * byte + k < nbits && (h & 6) == 4 &&
					((h >> 3) & 31) <= 29 && ((h >> 8) & 31) <= 29;
			const u64 m = __ballot(ok);
			if (ok)
				wq[wave][nq + __builtin_popcountll(m & ((1ull << lane) - 1))] =
					(u16)(8 * lane + k);
			nq += __builtin_popcountll(m);
		}
		wave_sync();
		for (u32 q0 = 0; q0 < nq; q0 += 64) {
			bool pass = false;
			u64 p = 0;
			if (q0 + lane < nq) {
				p = 8 * (byte - lane) + wq[wave][q0 + lane];
				u64 x0, x1;
				bits128(inp, in_n, p, &x0, &x1);
				pass = precode_complete(x0, x1);
			}
			const u64 m = __ballot(pass);
			if (m) {
				u32 base = 0;
				if (lane == (u32)__builtin_ctzll(m))
					base = atomicAdd(&nfound, (u32)__builtin_popcountll(m));
				base = bcast_lane(base, (u32)__builtin_ctzll(m));
				if (pass) {
					const u32 at = base + __builtin_popcountll(m & ((1ull << lane) - 1));
					if (at < 512)
						found[at] = p;
				}
			}
		}
		wave_sync();
	}
	__syncthreads();
	const u32 nf = nfound < 512 ? nfound : 512;
	if (threadIdx.x == 0)
		gbase = nf ? atomicAdd(qcount, nf) : 0;
	__syncthreads();
	for (u32 i = threadIdx.x; i < nf; i += 256)
		if (gbase + i < qcap)
			queue[gbase + i] = found[i];
}

/*
 * Stage B, one thread per survivor: the code lengths decoded in full
 * (decompress_template.h:150-245) and both codes checked for what
 * build_decode_table() accepts and a compressor produces: the litlen code
 * complete with an end-of-block symbol; the offset code complete, or a single
 * 1-bit codeword, or empty.
 */
static __device__ void
find_b_one(const u8 *__restrict__ inp, u64 in_n, u64 p, lu16 *tab,
	   u64 *__restrict__ cand, u32 *__restrict__ ncand, u32 ccap);

extern "C" __global__ void __launch_bounds__(64)
lda_stream_find_b_kernel(const u8 *__restrict__ inp, u64 in_n,
			 const u64 *__restrict__ queue, const u32 *__restrict__ qcount,
			 u32 qcap, u64 *__restrict__ cand, u32 *__restrict__ ncand, u32 ccap)
{
	u32 nq = *qcount;
	nq = nq < qcap ? nq : qcap;
	lu16 *tab = (lu16 *)(uintptr_t)(threadIdx.x * 256u);
	/* (a fixed grid walks the queue: its length is only known on the device) */
	for (u32 idx = blockIdx.x * 64 + threadIdx.x; idx < nq; idx += gridDim.x * 64)
		find_b_one(inp, in_n, queue[idx], tab, cand, ncand, ccap);
}

static __device__ void
find_b_one(const u8 *__restrict__ inp, u64 in_n, u64 p, lu16 *tab,
	   u64 *__restrict__ cand, u32 *__restrict__ ncand, u32 ccap)
{
	u64 x0, x1;
	bits128(inp, in_n, p, &x0, &x1);
	const u32 h = (u32)x0;
	const u32 nlit = 257 + ((h >> 3) & 31), noff = 1 + ((h >> 8) & 31);
	const u32 npre = 4 + ((h >> 13) & 15);
	u64 f = (x0 >> 17) | (x1 << 47);
	f &= (1ull << (3 * npre)) - 1;
	/* the compressors send no trailing zero lengths (HCLEN, HLIT and HDIST are
	 * the smallest that do: lib/deflate_compress.c:1575-1586, :1618-1624;
	 * zlib's max_code); a header that does is, among incompressible bytes,
	 * nearly always a false one */
	if (npre > 4 && ((f >> (3 * (npre - 1))) & 7) == 0)
		return;
	/* The precode's decode table, from registers only (field i of f is the
	 * length of symbol c_pre_perm[i]; the first stage has checked that the
	 * code is complete): lengths counted in 5-bit fields, first codewords per
	 * length in 8-bit fields, codewords handed out in symbol order.  (The
	 * general build_precode() indexes small arrays with run-time values: they
	 * live in scratch memory, a trip to memory per access.) */
	{
		u64 cntp = 0, nextp = 0;
#pragma unroll
		for (u32 i = 0; i < 19; i++)
			cntp += 1ull << (5 * ((u32)(f >> (3 * i)) & 7));
		u32 code = 0;
#pragma unroll
		for (u32 l = 1; l <= 7; l++) {
			nextp |= (u64)code << (8 * l);
			code = (code + ((u32)(cntp >> (5 * l)) & 31)) << 1;
		}
		/* position of symbol s in the permuted header order */
		static constexpr u8 inv[19] = { 3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12,
						14, 16, 18, 0, 1, 2 };
#pragma unroll
		for (u32 sy = 0; sy < 19; sy++) {
			const u32 l = (u32)(f >> (3 * inv[sy])) & 7;
			if (l) {
				const u32 c = (u32)(nextp >> (8 * l)) & 0xFF;
				nextp += 1ull << (8 * l);
				const u32 rev = __brev(c) >> (32 - l);
				for (u32 e = rev; e < 128; e += 1u << l)
					tab[e] = ENTRY(0, sy, l);
			}
		}
	}
	u64 bp = p + 17 + 3 * npre;	/* next bit to read */
	u64 buf = 0;
	u32 cnt = 0;
	u32 i = 0, prev = 0, kl = 0, ko = 0, n_o = 0, eob_len = 0, last_lit = 0, last_off = 0;
	const u32 total = nlit + noff;
	bool ok = true;
	/* the lane's next 72 bytes of input sit in its LDS row (nine loads in
	 * flight at once, then ~60 code lengths without touching memory: a refill
	 * from memory per four code lengths made this kernel wait 150 us for a
	 * handful of candidates) */
	lu64 *row = (lu64 *)(uintptr_t)(FIND_B_TAB_BYTES + threadIdx.x * 72u);
	u64 row_byte = ~0ull;
	while (i < total) {
		if (cnt < 14) {
			u64 r = (bp >> 3) - row_byte;
			if (row_byte == ~0ull || r > 56) {
				row_byte = bp >> 3;
#pragma unroll
				for (u32 k = 0; k < 9; k++)
					row[k] = load_in(inp, in_n, row_byte + 8 * k);
				r = 0;
			}
			const u32 w = (u32)r >> 3, sh = 8 * ((u32)r & 7) + ((u32)bp & 7);
			const u64 a = row[w], b = row[w + 1];
			buf = sh ? (a >> sh) | (b << (64 - sh)) : a;
			cnt = 56;	/* at least 57 valid bits */
		}
		const u32 e = tab[(u32)buf & 127];
		u32 used = e & 15;
		const u32 presym = e >> 4;
		u32 rep = 1, val = presym;
		if (presym >= 16) {
			if (presym == 16) {
				if (i == 0) {
					ok = false;
					break;
				}
				val = prev;
				rep = 3 + ((u32)(buf >> used) & 3);
				used += 2;
			} else if (presym == 17) {
				val = 0;
				rep = 3 + ((u32)(buf >> used) & 7);
				used += 3;
			} else {
				val = 0;
				rep = 11 + ((u32)(buf >> used) & 127);
				used += 7;
			}
		}
		buf >>= used;
		cnt -= used;
		bp += used;
		if (i + rep > total) {
			ok = false;
			break;
		}
		if (val) {
			const u32 w = 32768u >> val;
			const u32 n1 = i < nlit ? (rep < nlit - i ? rep : nlit - i) : 0;
			kl += n1 * w;
			ko += (rep - n1) * w;
			n_o += rep - n1;
			if (i <= 256 && 256 < i + n1)
				eob_len = val;
			if (i + n1 == nlit && n1)
				last_lit = val;
			if (i + rep == total && rep > n1)
				last_off = val;
			/* an over-subscribed code cannot recover: most false survivors
			 * of the first stage end here within a dozen lengths */
			if (kl > 32768 || ko > 32768) {
				ok = false;
				break;
			}
		}
		prev = val;
		i += rep;
	}
	if (!ok || bp > 8 * in_n)
		return;
	if (kl != 32768 || eob_len == 0)
		return;
	if ((nlit > 257 && last_lit == 0) || (noff > 1 && last_off == 0))
		return;		/* trailing zero lengths: see above */
	if (!(ko == 32768 || (ko == 16384 && n_o == 1) || ko == 0))
		return;
	const u32 at = atomicAdd(ncand, 1u);
	if (at < ccap)
		cand[at] = p;
}

/* ---------------- markers -> bytes ---------------- */

/*
 * The window chain.  The last min(length, 32 KiB) symbols of chunk c - its
 * TAIL - settled against the 32 KiB in front of the chunk are the window of
 * chunk c + 1: a chain through all chunks, 32 Ki symbols per step.  It is cut
 * into GROUPS of consecutive chunks and run as a two-level scan:
 *
 *   phase 0  every group but the last, side by side: the chain through the
 *            group's chunks with a SYMBOLIC window - entry i of the window in
 *            front of the group is the marker 0x8000 | i - leaves the window
 *            behind the group in terms of the window in front of it;
 *   phase 1  lda_stream_window_scan_kernel: those symbolic windows composed by
 *            a prefix scan over the groups (log2(groups) launches);
 *   phase 2  every group side by side again, now from the real window in
 *            front of it: the tails' bytes go to the output.
 *
 * The window is a ring in LDS indexed by the absolute output position (mod
 * 32 Ki), 16 bits per entry; a step reads what its markers point at, then -
 * behind a barrier - overwrites the ring with its own tail.
 */
extern "C" __global__ void __launch_bounds__(1024)
lda_stream_window_kernel(u32 nchunks, u32 per_group, u32 phase,
			 const u64 *__restrict__ out_off /*[nchunks + 1]*/,
			 const u16 *__restrict__ sym, u8 *__restrict__ out,
			 u16 *__restrict__ gwin /*[groups][32768]: phase 0 out */,
			 const u16 *__restrict__ fwin /*[groups][32768]: phase 2 in (the scan's result) */,
			 u32 *__restrict__ err)
{
	lu16 *W = (lu16 *)(uintptr_t)0;		/* [32768], the launch's dynamic LDS */
	const u32 tid = threadIdx.x, g = blockIdx.x;
	const u32 c0 = g * per_group;
	u32 c1 = c0 + per_group;
	c1 = c1 < nchunks ? c1 : nchunks;
	if (c0 >= nchunks || (phase == 0 && c1 >= nchunks))
		return;		/* nobody needs the window behind the last group */
	const u64 s0 = out_off[c0];
	for (u32 i = tid; i < 32768; i += 1024) {
		u32 v = 0x8000u | i;
		if (phase != 0)
			v = g ? fwin[(size_t)(g - 1) * 32768 + i] : 0;
		/* (what is still a marker behind the scan points in front of the
		 * stream's first byte: caught, with its position, where it is used) */
		if (phase != 0 && (v & 0x8000u))
			v = 0;
		W[((u32)s0 + i) & 32767] = (u16)v;
	}
	__syncthreads();
	u32 bad = 0;
	/* 32 symbols per thread at most, held two to a register (symbols 2 tid +
	 * 2048 k and the one after it); the loops run in blocks of 4 registers and
	 * skip the blocks a short tail does not reach (the branch is uniform) */
	u32 nx[16];
#define TAIL_BLOCKS(n_, ...)                                                   \
	_Pragma("unroll") for (u32 kb = 0; kb < 4; kb++)                       \
		if (kb * 8192u < (n_)) {                                       \
			_Pragma("unroll") for (u32 k = 4 * kb; k < 4 * kb + 4; k++) { __VA_ARGS__ } \
		}
	auto load_tail = [&](u32 c) {
		const u64 s = out_off[c], e = out_off[c + 1];
		const u64 t0 = e - s > 32768 ? e - 32768 : s;
		const u32 n = (u32)(e - t0);
		TAIL_BLOCKS(n, {
			const u32 i = 2 * tid + 2048 * k;
			u32 v2 = 0;
			if (i + 1 < n)
				__builtin_memcpy(&v2, sym + t0 + i, 4);
			else if (i < n)
				v2 = sym[t0 + i];
			nx[k] = v2;
		})
	};
	load_tail(c0);
	for (u32 c = c0; c < c1; c++) {
		const u64 s = out_off[c], e = out_off[c + 1];
		const u64 t0 = e - s > 32768 ? e - 32768 : s;
		const u32 n = (u32)(e - t0);
		u32 v[16];
		TAIL_BLOCKS(n, { v[k] = nx[k]; })
		if (c + 1 < c1)
			load_tail(c + 1);
		/* a marker w of this chunk is the position s - 32768 + w; the ones
		 * below `lowest` lie before the stream's first byte */
		const u32 lowest = s < 32768 ? 32768 - (u32)s : 0;
		TAIL_BLOCKS(n, {
			const u32 i = 2 * tid + 2048 * k;
			u32 a = v[k] & 0xFFFF, b = v[k] >> 16;
			if (i < n && (a & 0x8000)) {
				const u32 w = a & 0x7FFF;
				if (phase != 0 && w < lowest) {
					bad = 1;
					a = 0;
				} else {
					a = W[((u32)s + w) & 32767];
				}
			}
			if (i + 1 < n && (b & 0x8000)) {
				const u32 w = b & 0x7FFF;
				if (phase != 0 && w < lowest) {
					bad = 1;
					b = 0;
				} else {
					b = W[((u32)s + w) & 32767];
				}
			}
			v[k] = a | (b << 16);
		})
		__syncthreads();
		TAIL_BLOCKS(n, {
			const u32 i = 2 * tid + 2048 * k;
			if (i < n) {
				W[((u32)t0 + i) & 32767] = (u16)v[k];
				if (phase != 0)
					out[t0 + i] = (u8)v[k];
			}
			if (i + 1 < n) {
				W[((u32)t0 + i + 1) & 32767] = (u16)(v[k] >> 16);
				if (phase != 0)
					out[t0 + i + 1] = (u8)(v[k] >> 16);
			}
		})
		__syncthreads();
	}
#undef TAIL_BLOCKS
	if (phase == 0) {
		const u64 e1 = out_off[c1];
		for (u32 i = tid; i < 32768; i += 1024)
			gwin[(size_t)g * 32768 + i] = W[((u32)e1 + i) & 32767];
	}
	if (bad)
		*err = 1;
}

/*
 * phase 1: the symbolic windows behind the groups, composed.  The window
 * behind group g in terms of the window in front of group g (what phase 0
 * leaves) and the window behind group g - h in terms of the one in front of
 * group g - 2h + 1 give the window behind g in terms of the one in front of
 * g - 2h + 1: an entry that is a byte stays, a marker w becomes entry w of the
 * other window.  Composition is associative, so the chain through all groups
 * is a prefix scan: ceil(log2(groups)) launches with h = 1, 2, 4 .., every
 * group a workgroup that holds the window it looks things up in in LDS
 * (Hillis / Steele; the arrays alternate).  Round 5 walked the groups one
 * after the other in one workgroup: 2.3 us per group, 173 us of a 16 MiB
 * call - and the price of a group there kept the groups few and the group
 * passes long (20 chunks each).  What is still a marker at the end points in
 * front of the stream.
 */
extern "C" __global__ void __launch_bounds__(1024)
lda_stream_window_scan_kernel(u32 n, u32 h, const u16 *__restrict__ src, u16 *__restrict__ dst)
{
	lu16 *A = (lu16 *)(uintptr_t)0;		/* [32768], the launch's dynamic LDS */
	const u32 tid = threadIdx.x, g = blockIdx.x;
	if (g >= n)
		return;
	const uint4 *sb = (const uint4 *)(src + (size_t)g * 32768);
	uint4 *db = (uint4 *)(dst + (size_t)g * 32768);
	if (g < h) {	/* already in terms of the window in front of the stream */
		for (u32 q = tid; q < 4096; q += 1024)
			db[q] = sb[q];
		return;
	}
	const uint4 *sa = (const uint4 *)(src + (size_t)(g - h) * 32768);
	uint4 vb[4];
#pragma unroll
	for (u32 k = 0; k < 4; k++) {
		vb[k] = sb[tid + 1024 * k];
		((AS3 uint4 *)A)[tid + 1024 * k] = sa[tid + 1024 * k];
	}
	__syncthreads();
#pragma unroll
	for (u32 k = 0; k < 4; k++) {
		u32 w[4] = { vb[k].x, vb[k].y, vb[k].z, vb[k].w };
#pragma unroll
		for (u32 j = 0; j < 4; j++) {
			u32 lo = w[j] & 0xFFFF, hi = w[j] >> 16;
			if (lo & 0x8000)
				lo = A[lo & 0x7FFF];
			if (hi & 0x8000)
				hi = A[hi & 0x7FFF];
			w[j] = lo | (hi << 16);
		}
		db[tid + 1024 * k] = make_uint4(w[0], w[1], w[2], w[3]);
	}
}

/* everything in front of a chunk's tail, 8 symbols per thread */
extern "C" __global__ void __launch_bounds__(256)
lda_stream_resolve_kernel(u32 nchunks, u32 chunk0, const u64 *__restrict__ out_off,
			  const u16 *__restrict__ sym, u8 *__restrict__ out,
			  u32 *__restrict__ err)
{
	/* (grid.y is limited to 65535: the host launches batches of chunks) */
	const u32 c = chunk0 + blockIdx.y;
	if (c >= nchunks)
		return;
	const u64 s = out_off[c], e = out_off[c + 1];
	if (e - s <= 32768)
		return;		/* all of it is tail */
	const u64 t0 = e - 32768;
	const u32 lowest = s < 32768 ? 32768 - (u32)s : 0;
	const u8 *win = out + s - 32768;	/* (never read below `lowest`) */
	for (u64 i = s + 8 * ((u64)blockIdx.x * 256 + threadIdx.x); i < t0;
	     i += 8ull * 256 * gridDim.x) {
		u64 r = 0;
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			if (i + k < t0) {
				u32 b = sym[i + k];
				if (b & 0x8000) {
					const u32 w = b & 0x7FFF;
					if (w < lowest) {
						*err = 1;
						b = 0;
					} else {
						b = win[w];
					}
				}
				r |= (u64)(b & 0xFF) << (8 * k);
			}
		}
		if (i + 8 <= t0) {
			__builtin_memcpy(out + i, &r, 8);
		} else {
			for (u32 k = 0; i + k < t0; k++)
				out[i + k] = (u8)(r >> (8 * k));
		}
	}
}
