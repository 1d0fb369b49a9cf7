/*
 * deflate_opt.h - the min-cost parse of levels 10-12 (a part of
 * deflate_kernel.hip, included by it: the stage between the chain search and
 * the token walk that lda_deflate_opt_kernel compiles in).  Restates
 * lib/deflate_compress.c:3327-3849 for the tile pipeline; see the comment below.
 */
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
#define OPT_FIT_NUM 7u	/* the block's literal statistics "fit" a tile up to 7/4 of */
#define OPT_FIT_DEN 4u	/* the tile's own literal-only estimate */
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
			{	/* ch[lane sl] = chosen length.  v_writelane_b32 takes one
				 * SGPR, so the lane select travels in m0 - saved and put back
				 * inside the statement (m0 is a reserved register: it may not
				 * simply be declared clobbered); the s_nop covers the
				 * lane-select hazard the compiler cannot see in the asm */
				const u32 cl = best & 511;
				u32 m0save;
				asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\t"
				    "v_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
				    : "+v"(ch), "=&s"(m0save) : "s"(cl), "s"(sl));
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
