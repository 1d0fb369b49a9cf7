/*
 * oracle_deflate.c - CPU restatement of the reference compressor's policy.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What is restated (each function cites the lines it follows):
 *   level table, passthrough size          lib/deflate_compress.c:3925-3979
 *   hash-chain match finder (hash3 + hash4 chains, depth, nice length)
 *                                          lib/hc_matchfinder.h:182-399
 *   greedy / lazy / lazy2 parsers          lib/deflate_compress.c:2528-2808
 *   minimum match length heuristic         lib/deflate_compress.c:2295-2378
 *   block split heuristic                  lib/deflate_compress.c:2092-2218
 *   length-limited Huffman codes           lib/deflate_compress.c:759-1396
 *   precode RLE + exact block costs        lib/deflate_compress.c:1482-1631,
 *                                          :1747-1808
 *   stored blocks, compress_bound          lib/deflate_compress.c:2392-2443,
 *                                          :4087-4135
 *   gzip / zlib framing                    lib/gzip_compress.c:31-90,
 *                                          lib/zlib_compress.c:31-82
 *
 * The compressed bytes are NOT expected to equal the reference's
 * (libdeflate.h:76-83): data structures are the plainest possible (full
 * 32-bit position arrays instead of the sliding s16 tables, a textbook
 * heap-free Huffman build with a Kraft repair instead of the reference's
 * packed sort), levels 10-12 fall back to the level-9 parser.  tests/
 * check that streams round-trip through oracle_inflate and oracle/_ref and
 * that sizes track the reference's within a few percent.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define WSIZE 32768
#define MINB 5000		/* MIN_BLOCK_LENGTH, deflate_compress.c:66 */
#define SOFTMAX 300000		/* SOFT_MAX_BLOCK_LENGTH, :81 */
#define SEQMAX 50000		/* SEQ_STORE_LENGTH, :93 */

struct seq { uint32_t litrun, len, off; };

struct bitw { uint8_t *out; size_t cap, pos; uint64_t acc; unsigned n; int ovf; };

static void putbits(struct bitw *w, uint32_t v, unsigned n)
{
	w->acc |= (uint64_t)v << w->n;
	w->n += n;
	while (w->n >= 8) {
		if (w->pos < w->cap)
			w->out[w->pos++] = (uint8_t)w->acc;
		else
			w->ovf = 1;
		w->acc >>= 8;
		w->n -= 8;
	}
}

static void alignbyte(struct bitw *w)
{
	if (w->n)
		putbits(w, 0, 8 - w->n);
}

/* ---- tables: length / offset slots (lib/deflate_compress.c:237-308) ---- */
static const uint16_t LBASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19,
	23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t LXB[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
	3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t OBASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65,
	97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
	8193, 12289, 16385, 24577 };
static const uint8_t OXB[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
	7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

static unsigned lslot(unsigned len)
{
	unsigned s = 28;
	while (LBASE[s] > len)
		s--;
	return s;
}

static unsigned oslot(unsigned off)
{
	unsigned s = 29;
	while (OBASE[s] > off)
		s--;
	return s;
}

/* ---- Huffman: lengths limited to maxlen, canonical bit-reversed codes ---- */
static void make_code(const uint32_t *freq, unsigned n, unsigned maxlen,
		      uint8_t *lens, uint16_t *codes)
{
	unsigned sym[288], m = 0, i, j;
	uint32_t w[2 * 288];
	int parent[2 * 288];
	unsigned depth[2 * 288], cnt[64] = { 0 };

	memset(lens, 0, n);
	for (i = 0; i < n; i++)
		if (freq[i])
			sym[m++] = i;
	/* sort by (freq, sym): lib/deflate_compress.c:846-906 ordering */
	for (i = 1; i < m; i++) {
		unsigned s = sym[i];
		for (j = i; j && (freq[sym[j - 1]] > freq[s] ||
				  (freq[sym[j - 1]] == freq[s] && sym[j - 1] > s)); j--)
			sym[j] = sym[j - 1];
		sym[j] = s;
	}
	if (m < 2) {
		/* lib/deflate_compress.c:1369-1378 */
		unsigned s = m ? sym[0] : 0;
		lens[s] = 1;
		lens[s ? 0 : 1] = 1;
	} else {
		/* two-queue tree build (:939-995) */
		unsigned leaf = 0, node = m, nn = m, a, b;
		for (i = 0; i < m; i++)
			w[i] = freq[sym[i]];
		while (leaf < m || nn - node > 1) {
			if (leaf < m && (node >= nn || w[leaf] <= w[node]))
				a = leaf++;
			else
				a = node++;
			if (leaf < m && (node >= nn || w[leaf] <= w[node]))
				b = leaf++;
			else
				b = node++;
			w[nn] = w[a] + w[b];
			parent[a] = parent[b] = (int)nn;
			nn++;
		}
		depth[nn - 1] = 0;
		for (i = nn - 1; i-- > 0;)
			depth[i] = depth[parent[i]] + 1;
		for (i = 0; i < m; i++)
			cnt[depth[i] < 63 ? depth[i] : 63]++;
		/* clamp + Kraft repair (what :1022-1091 achieves) */
		{
			unsigned over = 0, d;
			uint32_t kraft = 0;
			for (d = maxlen + 1; d < 64; d++) {
				over += cnt[d];
				cnt[maxlen] += cnt[d];
				cnt[d] = 0;
			}
			if (over) {
				for (d = 1; d <= maxlen; d++)
					kraft += cnt[d] << (maxlen - d);
				while (kraft > (1u << maxlen)) {
					d = maxlen - 1;
					while (cnt[d] == 0)
						d--;
					cnt[d]--;
					cnt[d + 1] += 2;
					cnt[maxlen]--;
					kraft--;
				}
			}
			i = 0;
			for (d = maxlen; d >= 1; d--)
				for (j = 0; j < cnt[d]; j++)
					lens[sym[i++]] = (uint8_t)d;
		}
	}
	/* canonical codewords (:1177-1216), bit-reversed (:1103-1152) */
	{
		unsigned bl[16] = { 0 }, next[16], code = 0, d;
		for (i = 0; i < n; i++)
			bl[lens[i]]++;
		bl[0] = 0;
		for (d = 1; d < 16; d++) {
			code = (code + bl[d - 1]) << 1;
			next[d] = code;
		}
		for (i = 0; i < n; i++) {
			unsigned l = lens[i], c, r = 0;
			if (!l) {
				codes[i] = 0;
				continue;
			}
			c = next[l]++;
			for (d = 0; d < l; d++)
				r |= ((c >> d) & 1) << (l - 1 - d);
			codes[i] = (uint16_t)r;
		}
	}
}

/* ---- hash-chain match finder (lib/hc_matchfinder.h) ---- */
struct mf {
	int32_t *head3, *head4, *prev;	/* absolute positions, -1 = none */
	size_t ins;			/* positions < ins are inserted */
};

static uint32_t ld32(const uint8_t *p)
{
	return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
}

static unsigned h3(const uint8_t *p)
{
	return (unsigned)(((ld32(p) & 0xFFFFFF) * 0x1E35A7BDu) >> 17);	/* 15 bits */
}

static unsigned h4(const uint8_t *p)
{
	return (unsigned)((ld32(p) * 0x1E35A7BDu) >> 16);		/* 16 bits */
}

static void mf_insert(struct mf *m, const uint8_t *in, size_t n, size_t pos)
{
	if (pos < m->ins)
		return;
	m->ins = pos + 1;
	if (pos + 5 > n)	/* hc_matchfinder.h:214-215, :373-374 */
		return;
	m->head3[h3(in + pos)] = (int32_t)pos;
	m->prev[pos] = m->head4[h4(in + pos)];
	m->head4[h4(in + pos)] = (int32_t)pos;
}

/* longest match at pos beating best_len; inserts pos (hc_matchfinder.h:182-338) */
static unsigned mf_longest(struct mf *m, const uint8_t *in, size_t n, size_t pos,
			   unsigned best_len, unsigned max_len, unsigned nice,
			   unsigned depth, unsigned *off_ret)
{
	int32_t c3, c4;
	size_t best_pos = 0;
	unsigned len;
	int found = 0;

	if (max_len < 5)
		return best_len;
	c3 = m->head3[h3(in + pos)];
	c4 = m->head4[h4(in + pos)];
	if (pos < m->ins) {	/* already inserted by an earlier look-ahead */
		if (c4 == (int32_t)pos)
			c4 = m->prev[pos];
		if (c3 == (int32_t)pos)
			c3 = -1;
	}
	mf_insert(m, in, n, pos);
	if (best_len < 3 && c3 >= 0 && pos - (size_t)c3 <= WSIZE &&
	    !memcmp(in + c3, in + pos, 3)) {
		best_len = 3;
		best_pos = (size_t)c3;
		found = 1;
	}
	while (c4 >= 0 && pos - (size_t)c4 <= WSIZE && depth--) {
		const uint8_t *a = in + pos, *b = in + c4;
		if (ld32(a) == ld32(b) &&
		    (best_len < 4 || a[best_len] == b[best_len])) {
			len = 4;
			while (len < max_len && a[len] == b[len])
				len++;
			if (len > best_len) {
				best_len = len;
				best_pos = (size_t)c4;
				found = 1;
				if (len >= nice)
					break;
			}
		}
		c4 = m->prev[c4];
	}
	if (found)
		*off_ret = (unsigned)(pos - best_pos);
	return best_len;
}

static unsigned bsr(unsigned v)
{
	unsigned r = 0;
	while (v >>= 1)
		r++;
	return r;
}

/* lib/deflate_compress.c:2295-2327 */
static unsigned choose_min_len(unsigned used, unsigned depth)
{
	static const uint8_t t[] = { 9, 9, 9, 9, 9, 9, 8, 8, 7, 7, 6, 6, 6, 6, 6, 6 };
	unsigned m = used >= 80 ? 3 : used >= 45 ? 4 : used >= 16 ? 5 : t[used];
	if (depth < 16) {
		unsigned cap = depth < 5 ? 4 : depth < 10 ? 5 : 7;
		if (m > cap)
			m = cap;
	}
	return m;
}

/* block split statistics, lib/deflate_compress.c:2092-2218 */
struct split { uint32_t nw[10], ob[10], nnw, nob; };

static void observe(struct split *s, int type)
{
	s->nw[type]++;
	s->nnw++;
}

static int should_end(struct split *s, size_t blen, size_t remaining)
{
	uint32_t total = 0, cutoff;
	int i;

	if (s->nnw < 512 || blen < MINB || remaining < MINB)
		return 0;
	if (s->nob > 0) {
		for (i = 0; i < 10; i++) {
			uint32_t e = s->ob[i] * s->nnw, a = s->nw[i] * s->nob;
			total += a > e ? a - e : e - a;
		}
		cutoff = s->nnw * 200 / 512 * s->nob;
		if (blen < 10000 && s->nob + s->nnw < 8192)
			cutoff += (uint64_t)cutoff * (8192 - (s->nob + s->nnw)) / 8192;
		if (total + (uint32_t)(blen / 4096) * s->nob >= cutoff)
			return 1;
	}
	for (i = 0; i < 10; i++) {
		s->ob[i] += s->nw[i];
		s->nw[i] = 0;
	}
	s->nob += s->nnw;
	s->nnw = 0;
	return 0;
}

/* ---- block output (lib/deflate_compress.c:1706-2038) ---- */
static void flush_block(struct bitw *w, const uint8_t *blk, size_t blen,
			const struct seq *sq, size_t nseq, uint32_t *fl, uint32_t *fo,
			int final)
{
	static const uint8_t perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4,
					  12, 3, 13, 2, 14, 1, 15 };
	uint8_t ll[288], ol[32], pl[19], sl[288], so[32];
	uint16_t lc[288], oc[32], pc[19], slc[288], soc[32];
	uint32_t pf[19] = { 0 }, sfl[288], sfo[32];
	uint16_t items[320];
	unsigned nlit = 288, noff = 32, nitems = 0, nexp = 19, i;
	uint64_t dyn, stat, stored;
	uint8_t all[320];

	fl[256]++;
	make_code(fl, 288, 14, ll, lc);	/* litlen limit 14, :117 */
	make_code(fo, 32, 15, ol, oc);
	for (i = 0; i < 288; i++)
		sfl[i] = 1u << (9 - (i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8));
	for (i = 0; i < 32; i++)
		sfo[i] = 1;
	make_code(sfl, 288, 15, sl, slc);	/* static codes, :1432-1450 */
	make_code(sfo, 32, 15, so, soc);
	while (nlit > 257 && !ll[nlit - 1])
		nlit--;
	while (noff > 1 && !ol[noff - 1])
		noff--;
	memcpy(all, ll, nlit);
	memcpy(all + nlit, ol, noff);
	/* RLE into precode items (:1482-1557) */
	for (i = 0; i < nlit + noff;) {
		unsigned v = all[i], run = 1, left;
		while (i + run < nlit + noff && all[i + run] == v)
			run++;
		left = run;
		if (v == 0) {
			while (left >= 11) {
				unsigned r = left > 138 ? 138 : left;
				items[nitems++] = (uint16_t)(18 | ((r - 11) << 5));
				pf[18]++;
				left -= r;
			}
			if (left >= 3) {
				items[nitems++] = (uint16_t)(17 | ((left - 3) << 5));
				pf[17]++;
				left = 0;
			}
		} else if (left >= 4) {
			items[nitems++] = (uint16_t)v;
			pf[v]++;
			left--;
			while (left >= 3) {
				unsigned r = left > 6 ? 6 : left;
				items[nitems++] = (uint16_t)(16 | ((r - 3) << 5));
				pf[16]++;
				left -= r;
			}
		}
		while (left--) {
			items[nitems++] = (uint16_t)v;
			pf[v]++;
		}
		i += run;
	}
	make_code(pf, 19, 7, pl, pc);
	while (nexp > 4 && !pl[perm[nexp - 1]])
		nexp--;
	/* exact costs (:1747-1808) */
	dyn = 3 + 5 + 5 + 4 + 3 * nexp;
	stat = 3;
	for (i = 0; i < 19; i++)
		dyn += (uint64_t)pf[i] * (pl[i] + (i == 16 ? 2 : i == 17 ? 3 : i == 18 ? 7 : 0));
	for (i = 0; i < 288; i++) {
		unsigned xb = i >= 257 ? LXB[i - 257 > 28 ? 28 : i - 257] : 0;
		dyn += (uint64_t)fl[i] * (ll[i] + xb);
		stat += (uint64_t)fl[i] * (sl[i] + xb);
	}
	for (i = 0; i < 30; i++) {
		dyn += (uint64_t)fo[i] * (ol[i] + OXB[i]);
		stat += (uint64_t)fo[i] * (so[i] + OXB[i]);
	}
	stored = ((0 - (w->n + 3)) & 7) + 3 + 32 + 8 * (uint64_t)blen +
		 (blen ? (blen - 1) / 65535 : 0) * 40;
	if (stored <= stat && stored <= dyn) {
		size_t done = 0;
		do {
			size_t piece = blen - done > 65535 ? 65535 : blen - done;
			putbits(w, (final && done + piece == blen) ? 1 : 0, 1);
			putbits(w, 0, 2);
			alignbyte(w);
			putbits(w, (uint32_t)piece, 16);
			putbits(w, (uint32_t)piece ^ 0xFFFF, 16);
			for (i = 0; i < piece; i++)
				putbits(w, blk[done + i], 8);
			done += piece;
		} while (done < blen);
		return;
	}
	{
		const uint8_t *L = ll, *O = ol;
		const uint16_t *LC = lc, *OC = oc;
		const uint8_t *p = blk;
		size_t k;

		putbits(w, (uint32_t)final, 1);
		if (stat <= dyn) {
			putbits(w, 1, 2);
			L = sl; O = so; LC = slc; OC = soc;
		} else {
			putbits(w, 2, 2);
			putbits(w, nlit - 257, 5);
			putbits(w, noff - 1, 5);
			putbits(w, nexp - 4, 4);
			for (i = 0; i < nexp; i++)
				putbits(w, pl[perm[i]], 3);
			for (i = 0; i < nitems; i++) {
				unsigned s = items[i] & 31, ex = items[i] >> 5;
				putbits(w, pc[s], pl[s]);
				if (s >= 16)
					putbits(w, ex, s == 16 ? 2 : s == 17 ? 3 : 7);
			}
		}
		for (k = 0; k <= nseq; k++) {
			uint32_t r;
			for (r = 0; r < sq[k].litrun; r++, p++)
				putbits(w, LC[*p], L[*p]);
			if (k < nseq) {
				unsigned ls = lslot(sq[k].len), os = oslot(sq[k].off);
				putbits(w, LC[257 + ls], L[257 + ls]);
				putbits(w, sq[k].len - LBASE[ls], LXB[ls]);
				putbits(w, OC[os], O[os]);
				putbits(w, sq[k].off - OBASE[os], OXB[os]);
				p += sq[k].len;
			}
		}
		putbits(w, LC[256], L[256]);
	}
}

size_t oracle_deflate_compress_bound(size_t n)
{
	size_t b = (n + 4999) / 5000;	/* deflate_compress.c:4124-4134 */
	return 5 * (b ? b : 1) + n;
}

size_t oracle_zlib_compress_bound(size_t n)
{
	return 6 + oracle_deflate_compress_bound(n);
}

size_t oracle_gzip_compress_bound(size_t n)
{
	return 18 + oracle_deflate_compress_bound(n);
}

size_t oracle_deflate_compress(int level, const void *in_, size_t n, void *out,
			       size_t avail)
{
	/* level table, lib/deflate_compress.c:3927-3979 */
	static const struct { unsigned depth, nice, mode; } LV[10] = {
		{ 0, 0, 0 }, { 2, 32, 0 }, { 6, 10, 0 }, { 12, 14, 0 },
		{ 16, 30, 0 }, { 16, 30, 1 }, { 35, 65, 1 }, { 100, 130, 1 },
		{ 300, 258, 2 }, { 600, 258, 2 } };
	const uint8_t *in = (const uint8_t *)in_;
	struct bitw w = { (uint8_t *)out, avail, 0, 0, 0, 0 };
	struct mf m;
	struct seq *sq;
	size_t pos = 0;
	unsigned depth, nice, mode;

	if (level < 0)
		level = 6;
	if (level > 9)
		level = 9;
	depth = LV[level].depth;
	nice = LV[level].nice;
	mode = LV[level].mode;
	if (level == 0 || n <= (size_t)(55 - 4 * level)) {
		/* deflate_compress_none, :2392-2443 */
		size_t done = 0;
		do {
			size_t piece = n - done > 65535 ? 65535 : n - done;
			unsigned i;
			putbits(&w, done + piece == n, 1);
			putbits(&w, 0, 2);
			alignbyte(&w);
			putbits(&w, (uint32_t)piece, 16);
			putbits(&w, (uint32_t)piece ^ 0xFFFF, 16);
			for (i = 0; i < piece; i++)
				putbits(&w, in[done + i], 8);
			done += piece;
		} while (done < n);
		return w.ovf ? 0 : w.pos;
	}
	m.head3 = (int32_t *)malloc(sizeof(int32_t) * (1 << 15));
	m.head4 = (int32_t *)malloc(sizeof(int32_t) * (1 << 16));
	m.prev = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
	sq = (struct seq *)malloc(sizeof(*sq) * (SEQMAX + 2));
	memset(m.head3, 0xFF, sizeof(int32_t) * (1 << 15));
	memset(m.head4, 0xFF, sizeof(int32_t) * (1 << 16));
	m.ins = 0;

	while (pos < n && !w.ovf) {
		size_t bstart = pos, bmax, nseq = 0, recalc;
		uint32_t fl[288] = { 0 }, fo[32] = { 0 };
		struct split sp;
		unsigned min_len;

		memset(&sp, 0, sizeof(sp));
		bmax = n - bstart < SOFTMAX + MINB ? n : bstart + SOFTMAX;
		recalc = bstart + (n - bstart < 10000 ? n - bstart : 10000);
		sq[0].litrun = 0;
		{	/* calculate_min_match_len, :2329-2353 */
			size_t span = bmax - bstart, i;
			uint8_t used[256] = { 0 };
			unsigned cnt = 0;
			if (span < 512) {
				min_len = 3;
			} else {
				for (i = 0; i < (span < 4096 ? span : 4096); i++)
					used[in[bstart + i]] = 1;
				for (i = 0; i < 256; i++)
					cnt += used[i];
				min_len = choose_min_len(cnt, depth);
			}
		}
		while (pos < bmax && nseq < SEQMAX) {
			unsigned max_len = n - pos < 258 ? (unsigned)(n - pos) : 258;
			unsigned nl = nice < max_len ? nice : max_len;
			unsigned off = 0, len, i;

			if (mode && pos >= recalc) {
				/* recalculate_min_match_len, :2359-2378 */
				uint32_t tot = 0, cut;
				unsigned cnt = 0;
				for (i = 0; i < 256; i++)
					tot += fl[i];
				cut = tot >> 10;
				for (i = 0; i < 256; i++)
					cnt += fl[i] > cut;
				min_len = choose_min_len(cnt, depth);
				recalc += (n - recalc < pos - bstart) ? n - recalc : pos - bstart;
			}
			len = mf_longest(&m, in, n, pos, min_len - 1, max_len, nl, depth, &off);
			if (len < min_len || !off ||
			    (len == 3 && off > (mode ? 8192u : 4096u))) {
				fl[in[pos]]++;
				observe(&sp, ((in[pos] >> 5) & 6) | (in[pos] & 1));
				sq[nseq].litrun++;
				pos++;
			} else {
				/* lazy evaluation, :2712-2755 */
				for (;;) {
					unsigned noff = 0, nlen, k, look = mode == 2 ? 2 : mode;
					int deferred = 0;
					if (len >= nl)
						break;
					for (k = 1; k <= look && pos + k < n; k++) {
						unsigned ml = n - (pos + k) < 258 ?
							(unsigned)(n - (pos + k)) : 258;
						unsigned nn = nice < ml ? nice : ml;
						nlen = mf_longest(&m, in, n, pos + k, len - 1, ml, nn,
								  depth >> k, &noff);
						if (nlen >= len && noff &&
						    4 * (int)(nlen - len) +
						    ((int)bsr(off) - (int)bsr(noff)) > (k == 1 ? 2 : 6)) {
							for (i = 0; i < k; i++) {
								fl[in[pos + i]]++;
								observe(&sp, ((in[pos + i] >> 5) & 6) |
									     (in[pos + i] & 1));
								sq[nseq].litrun++;
							}
							pos += k;
							len = nlen;
							off = noff;
							deferred = 1;
							break;
						}
					}
					if (!deferred)
						break;
				}
				fl[257 + lslot(len)]++;
				fo[oslot(off)]++;
				observe(&sp, 8 + (len >= 9));
				sq[nseq].len = len;
				sq[nseq].off = off;
				nseq++;
				sq[nseq].litrun = 0;
				/* skip_bytes: insert the covered positions */
				for (i = 1; i < len; i++)
					mf_insert(&m, in, n, pos + i);
				pos += len;
			}
			if (level >= 2 && should_end(&sp, pos - bstart, n - pos))
				break;
		}
		flush_block(&w, in + bstart, pos - bstart, sq, nseq, fl, fo, pos == n);
	}
	if (w.n)
		putbits(&w, 0, 8 - w.n);
	free(m.head3);
	free(m.head4);
	free(m.prev);
	free(sq);
	return w.ovf ? 0 : w.pos;
}

size_t oracle_zlib_compress(int level, const void *in, size_t n, void *out_,
			    size_t avail)
{
	uint8_t *out = (uint8_t *)out_;
	unsigned fl = level < 2 ? 0 : level < 6 ? 1 : level < 8 ? 2 : 3;
	unsigned h = (0x78u << 8) | (fl << 6);
	uint32_t a;
	size_t c;

	if (avail <= 6)		/* zlib_compress.c:42-43 */
		return 0;
	h |= 31 - (h % 31);
	out[0] = (uint8_t)(h >> 8);
	out[1] = (uint8_t)h;
	c = oracle_deflate_compress(level, in, n, out + 2, avail - 6);
	if (!c)
		return 0;
	a = oracle_adler32(1, n ? in : "", n);
	out[2 + c] = (uint8_t)(a >> 24);
	out[3 + c] = (uint8_t)(a >> 16);
	out[4 + c] = (uint8_t)(a >> 8);
	out[5 + c] = (uint8_t)a;
	return c + 6;
}

size_t oracle_gzip_compress(int level, const void *in, size_t n, void *out_,
			    size_t avail)
{
	uint8_t *out = (uint8_t *)out_;
	uint32_t crc;
	size_t c;
	int i;

	if (avail <= 18)	/* gzip_compress.c:41-42 */
		return 0;
	memcpy(out, "\x1f\x8b\x08\x00\x00\x00\x00\x00", 8);
	out[8] = level < 2 ? 4 : level >= 8 ? 2 : 0;	/* XFL, :55-61 */
	out[9] = 0xFF;
	c = oracle_deflate_compress(level, in, n, out + 10, avail - 18);
	if (!c)
		return 0;
	crc = oracle_crc32(0, n ? in : "", n);
	for (i = 0; i < 4; i++) {
		out[10 + c + i] = (uint8_t)(crc >> (8 * i));
		out[14 + c + i] = (uint8_t)((uint32_t)n >> (8 * i));
	}
	return c + 18;
}
