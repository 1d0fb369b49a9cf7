/* CPU model of the compress kernel's progressive search (design aid, not
 * product code): chain steps per position and output size of "search
 * everything" against round A + parse + round B.
 * cc -O2 -o sim_mf sim_mf.c -lm ; ./sim_mf <file of 64 KiB chunks> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define CH 65536
static uint8_t buf[CH + 512];
static int n;

static int W = 32768;

static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

static int bsr(unsigned v) { return 31 - __builtin_clz(v); }

/* ---- cost ---- */
static int len_sym[259], len_xb[259];
static int off_slot(int d) { d--; if (d < 4) return d; int hb = bsr(d); return 2 * hb + ((d >> (hb - 1)) & 1); }
static int off_xb(int s) { return s < 4 ? 0 : (s >> 1) - 1; }
static void init_tabs(void)
{
	static const int lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35,
				       43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
	static const int lx[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
	for (int s = 0; s < 29; s++)
		for (int l = lbase[s]; l <= (s == 28 ? 258 : lbase[s] + (1 << lx[s]) - 1) && l <= 258; l++) {
			len_sym[l] = 257 + s; len_xb[l] = lx[s];
		}
	len_sym[258] = 285; len_xb[258] = 0;
}
static uint64_t huff_cost(const uint32_t *f, int nsym)
{
	/* optimal (unlimited) Huffman cost = sum of internal node weights */
	uint64_t w[700]; int cnt = 0;
	for (int s = 0; s < nsym; s++) if (f[s]) w[cnt++] = f[s];
	if (cnt == 0) return 0;
	if (cnt == 1) return w[0];
	uint64_t total = 0;
	while (cnt > 1) {
		int a = 0, b = 1;
		if (w[b] < w[a]) { a = 1; b = 0; }
		for (int i = 2; i < cnt; i++) {
			if (w[i] < w[a]) { b = a; a = i; } else if (w[i] < w[b]) b = i;
		}
		uint64_t s = w[a] + w[b];
		total += s;
		if (a > b) { int t = a; a = b; b = t; }
		w[a] = s; w[b] = w[cnt - 1]; cnt--;
	}
	return total;
}
static uint32_t fl[288], fo[32];
static uint64_t xbits;
static long ntok, nmatch;
static void tok_reset(void) { memset(fl, 0, sizeof fl); memset(fo, 0, sizeof fo); xbits = 0; }
static void tok_lit(int b) { fl[b]++; ntok++; }
static void tok_match(int len, int d) { fl[len_sym[len]]++; int s = off_slot(d); fo[s]++; xbits += len_xb[len] + off_xb(s); ntok++; nmatch++; }
static uint64_t tok_cost(void) { fl[256] = 1; uint64_t c = huff_cost(fl, 288) + huff_cost(fo, 32) + xbits + 600; uint64_t st = 8ull * n + 40; return c < st ? c : st; }

/* ---- matchfinder: chains on K-byte hash + optional single-slot tables ---- */
static int K = 4, HB = 16;
static int32_t headK[1 << 16], prevK[CH], head4s[1 << 16], head3s[1 << 15];
static long steps;	/* chain steps taken */
static long searched;	/* positions searched */
static int min_len = 3;

static uint32_t hashK(const uint8_t *p)
{
	if (K == 4) return (ld32(p) * 0x1E35A7BDu) >> (32 - HB);
	uint64_t v = ld64(p) & ((1ull << (8 * K)) - 1);
	return (uint32_t)((v * 0x9E3779B185EBCA87ull) >> (64 - HB));
}
static uint32_t hash4s(const uint8_t *p) { return (ld32(p) * 0x1E35A7BDu) >> 16; }
static uint32_t hash3s(const uint8_t *p) { return ((ld32(p) & 0xFFFFFF) * 0x1E35A7BDu) >> 17; }

static void mf_build(void)
{
	/* all links up front (what the GPU does per tile): prevK[p] = previous
	 * position with the same K-hash; p4[p], p3[p] = single-slot candidates */
}
static int32_t p4[CH], p3[CH];
static void mf_build_all(void)
{
	memset(headK, -1, sizeof headK); memset(head4s, -1, sizeof head4s); memset(head3s, -1, sizeof head3s);
	for (int i = 0; i < n; i++) {
		prevK[i] = -1; p4[i] = -1; p3[i] = -1;
		if (i + K <= n) { uint32_t h = hashK(buf + i); prevK[i] = headK[h]; headK[h] = i; }
		if (i + 4 <= n) { uint32_t h = hash4s(buf + i); p4[i] = head4s[h]; head4s[h] = i; }
		if (i + 3 <= n) { uint32_t h = hash3s(buf + i); p3[i] = head3s[h]; head3s[h] = i; }
	}
}
static int extend(int a, int b, int maxl)
{
	int l = 0;
	while (l < maxl && buf[a + l] == buf[b + l]) l++;
	return l;
}
/* longest match at p with chain depth D; returns len (>=3 or 0), *dist */
static int longest(int p, int D, int nice, int *dist, int mode3dist)
{
	int maxl = n - p < 258 ? n - p : 258;
	int best = 2, bd = 0;
	searched++;
	if (maxl < 3) return 0;
	if (nice > maxl) nice = maxl;
	/* length-3 / 4 single slots (only matter if chain is on K >= 5 or for len 3) */
	int c = prevK[p], d = D;
	while (c >= 0 && p - c <= W && d-- > 0) {
		steps++;
		if (buf[c + best] == buf[p + best] || best < 3) {
			int l = extend(c, p, maxl);
			if (l > best) { best = l; bd = p - c; if (l >= nice) break; }
		}
		c = prevK[c];
	}
	if (K > 4 && best < K && p4[p] >= 0 && p - p4[p] <= W) {
		steps++;
		int l = extend(p4[p], p, maxl);
		if (l > best && l >= 4) { best = l; bd = p - p4[p]; }
	}
	if (!getenv("NO3") && best < 4 && p3[p] >= 0 && p - p3[p] <= W) {
		int l = extend(p3[p], p, maxl);
		if (l >= 3 && l > best) { best = l; bd = p - p3[p]; }
	}
	if (getenv("NO3") && best < 4) { for (int d = 1; d <= (getenv("P16")?atoi(getenv("P16")):8) && d <= p; d++) if (buf[p-d]==buf[p] && buf[p-d+1]==buf[p+1] && buf[p-d+2]==buf[p+2] && maxl>=3) { best = 3; bd = d; break; } }
	if (best < 3) return 0;
	if (best == 3 && bd > mode3dist) return 0;
	if (best < min_len) return 0;
	*dist = bd;
	return best;
}

static int choose_min_len(int used, int depth)
{
	int m = used >= 80 ? 3 : used >= 45 ? 4 : used >= 16 ? 5 : used >= 10 ? 6 : used >= 8 ? 7 : used >= 6 ? 8 : 9;
	if (depth < 16) { int cap = depth < 5 ? 4 : depth < 10 ? 5 : 7; if (m > cap) m = cap; }
	return m;
}
static void set_min_len(int depth)
{
	int seen[256] = { 0 }, u = 0;
	for (int i = 0; i < 4096 && i < n; i++) if (!seen[buf[i]]) { seen[buf[i]] = 1; u++; }
	min_len = n < 512 ? 3 : choose_min_len(u, depth);
}

/* sequential lazy parse as the reference (depth D at token start, D/2 look-ahead) */
static void parse_seq_lazy(int D, int nice, int halve)
{
	int p = 0;
	while (p < n) {
		int d0, l0 = longest(p, D, nice, &d0, 8192);
		if (!l0) { tok_lit(buf[p]); p++; continue; }
		for (;;) {
			if (l0 >= nice) break;
			if (p + 1 >= n) break;
			int d1, l1 = longest(p + 1, halve ? (D / 2 > 0 ? D / 2 : 1) : D, nice, &d1, 8192);
			if (l1 >= l0 && 4 * (l1 - l0) + (bsr(d0) - bsr(d1)) > 2) {
				tok_lit(buf[p]); p++; l0 = l1; d0 = d1; continue;
			}
			break;
		}
		tok_match(l0, d0); p += l0;
	}
}

/* ---- row-hash match finder (mode 4): NROWS rows of ENT recent positions with
 * one-byte tags, all of a tile inserted before it is searched (what a GPU
 * kernel without chains would do): every position looks at the entries of its
 * row whose tag matches, newest first, at most CAND of them ---- */
static int NROWS = 1536, ENT = 15, CAND = 15, TILE_A = 4096, AHEAD = 1;
static uint16_t rpos[8192][256];
static uint8_t rtag[8192][256];
static uint32_t rcnt[8192];
static long rcands;
static void row_reset(void) { memset(rcnt, 0, sizeof rcnt); }
static void row_insert(int p)
{
	if (p + 4 > n) return;
	/* (the row from the TOP bits of the product - the low ones depend on the
	 * low input bytes only -, the tag from a second product) */
	uint32_t hv = ld32(buf + p) * 0x9E3779B1u;
	uint32_t row = (uint32_t)(((uint64_t)(hv >> 8) * NROWS) >> 24), slot = rcnt[row] % ENT;
	rcnt[row]++;
	rpos[row][slot] = (uint16_t)p;
	rtag[row][slot] = (uint8_t)((ld32(buf + p) * 0x85EBCA6Bu) >> 24);
}
static int row_longest(int p, int nice, int *dist, int mode3dist)
{
	int maxl = n - p < 258 ? n - p : 258;
	int best = 2, bd = 0;
	searched++;
	if (maxl < 3) return 0;
	if (nice > maxl) nice = maxl;
	if (p + 4 <= n) {
		uint32_t hv = ld32(buf + p) * 0x9E3779B1u;
		uint32_t row = (uint32_t)(((uint64_t)(hv >> 8) * NROWS) >> 24);
		uint8_t tag = (uint8_t)((ld32(buf + p) * 0x85EBCA6Bu) >> 24);
		int have = rcnt[row] < (uint32_t)ENT ? (int)rcnt[row] : ENT, tried = 0;
		for (int a = 0; a < have && tried < CAND; a++) {
			uint32_t slot = (rcnt[row] - 1 - a) % ENT;
			if (rtag[row][slot] != tag) continue;
			int c = rpos[row][slot];
			if (c >= p || p - c > W) continue;
			tried++; rcands++; steps++;
			if (buf[c + best] == buf[p + best] || best < 3) {
				int l = extend(c, p, maxl);
				if (l > best && l >= 4) { best = l; bd = p - c; if (l >= nice) break; }
			}
		}
	}
	if (best < 4 && p3[p] >= 0 && p - p3[p] <= W) {
		int l = extend(p3[p], p, maxl);
		if (l >= 3 && l > best) { best = l; bd = p - p3[p]; }
	}
	if (best < 3) return 0;
	if (best == 3 && bd > mode3dist) return 0;
	if (best < min_len) return 0;
	*dist = bd;
	return best;
}

/* all-position search then the token_step rule (the current GPU scheme) */
static int Mlen[CH + 4], Mdist[CH + 4];
static int use_rows;
static void parse_allpos(int D, int nice)
{
	if (use_rows) {
		row_reset();
		int inserted = 0;
		for (int i = 0; i < n; i++) {
			/* the tile of i (and AHEAD - 1 more) is in the rows before i is searched */
			int upto = (i / TILE_A + AHEAD) * TILE_A;
			if (upto > n) upto = n;
			for (; inserted < upto; inserted++) row_insert(inserted);
			int d = 0; Mlen[i] = row_longest(i, nice, &d, 8192); Mdist[i] = d;
		}
	} else
	for (int i = 0; i < n; i++) { int d = 0; Mlen[i] = longest(i, D, nice, &d, 8192); Mdist[i] = d; }
	Mlen[n] = Mlen[n + 1] = 0;
	int p = 0;
	while (p < n) {
		int l0 = Mlen[p];
		if (!l0) { tok_lit(buf[p]); p++; continue; }
		if (l0 < nice) {
			int l1 = Mlen[p + 1];
			if (l1 >= l0 && 4 * (l1 - l0) + (bsr(Mdist[p]) - bsr(Mdist[p + 1])) > 2) { tok_lit(buf[p]); p++; continue; }
		}
		tok_match(l0, Mdist[p]); p += l0;
	}
}

/* progressive deepening: all positions at depth d0, then rounds of
 * parse -> deepen what the parse visited (token starts to D, look-ahead to D/2) */
static int donedepth[CH + 4];
static long stepsat[CH + 4];
static int R = 3, D0 = 4;
static long rounds_used, deepened_total;
static void search_to(int p, int depth, int nice)
{
	if (donedepth[p] >= depth) return;
	long before = steps; int d = 0;
	Mlen[p] = longest(p, depth, nice, &d, 8192); Mdist[p] = d;
	long used = steps - before;
	/* incremental cost: the walk resumes where it stopped */
	steps = before + (used - stepsat[p]);
	stepsat[p] = used; donedepth[p] = depth;
}
static void parse_prog(int D, int nice)
{
	static int want[CH + 4];
	for (int i = 0; i < n; i++) { donedepth[i] = 0; stepsat[i] = 0; search_to(i, D0, nice); }
	Mlen[n] = Mlen[n + 1] = 0;
	for (int r = 0; ; r++) {
		int changed = 0;
		int p = 0;
		int final = r == R;
		memset(want, 0, sizeof(int) * (n + 2));
		while (p < n) {
			if (want[p] < D) want[p] = D;
			int l0 = Mlen[p];
			if (!l0) { if (final) tok_lit(buf[p]); p++; continue; }
			if (l0 < nice && p + 1 < n) {
				if (want[p + 1] < D / 2) want[p + 1] = D / 2;
				int l1 = Mlen[p + 1];
				if (l1 >= l0 && 4 * (l1 - l0) + (bsr(Mdist[p]) - bsr(Mdist[p + 1])) > 2) { if (final) tok_lit(buf[p]); p++; continue; }
			}
			if (final) tok_match(l0, Mdist[p]);
			p += l0;
		}
		if (final) break;
		for (int i = 0; i < n; i++)
			if (want[i] > donedepth[i]) { search_to(i, want[i], nice); changed++; }
		deepened_total += changed;
		rounds_used++;
		if (!changed) { r = R - 1; }
	}
}

int main(int argc, char **argv)
{
	FILE *f = fopen(argv[1], "rb");
	int mode = argc > 2 ? atoi(argv[2]) : 0;
	int D = argc > 3 ? atoi(argv[3]) : 35;
	int nice = argc > 4 ? atoi(argv[4]) : 65;
	K = argc > 5 ? atoi(argv[5]) : 4;
	W = argc > 6 ? atoi(argv[6]) : 32768;
	HB = argc > 7 ? atoi(argv[7]) : 16;
	D0 = argc > 8 ? atoi(argv[8]) : 4; R = argc > 9 ? atoi(argv[9]) : 3;
	if (getenv("NROWS")) NROWS = atoi(getenv("NROWS"));
	if (getenv("ENT")) ENT = atoi(getenv("ENT"));
	if (getenv("CAND")) CAND = atoi(getenv("CAND"));
	if (getenv("AHEAD")) AHEAD = atoi(getenv("AHEAD"));
	init_tabs();
	uint64_t total = 0, per[8] = { 0 };
	int idx = 0;
	while ((n = fread(buf, 1, CH, f)) > 0) {
		memset(buf + n, 0, 300);
		tok_reset();
		set_min_len(D);
		mf_build_all();
		if (mode == 0) parse_seq_lazy(D, nice, 1);
		else if (mode == 1) parse_allpos(D, nice);
		else if (mode == 2) parse_seq_lazy(D, nice, 0);
		else if (mode == 3) parse_prog(D, nice);
		else if (mode == 4) { use_rows = 1; parse_allpos(D, nice); }
		uint64_t c = (tok_cost() + 7) / 8;
		total += c; per[idx & 7] += c; idx++;
	}
	printf("mode %d D %d nice %d K %d W %d HB %d: bytes %llu  steps/pos %.2f searched/pos %.3f tok %ld match %ld | text %llu bin %llu low %llu\n",
	       mode, D, nice, K, W, HB, (unsigned long long)total, (double)steps / (idx * (double)CH),
	       (double)searched / (idx * (double)CH), ntok / idx, nmatch / idx,
	       (unsigned long long)per[0] / 8, (unsigned long long)per[5] / 8, (unsigned long long)per[6] / 8);
	if (mode == 3) printf("   rounds/chunk %.2f deepened/pos %.3f\n", (double)rounds_used / idx, (double)deepened_total / (idx * (double)CH));
	return 0;
}
