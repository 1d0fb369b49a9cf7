/*
 * deflate_huffman.h - length-limited canonical Huffman codes for a block (a part
 * of deflate_kernel.hip, included by it): make_code(), what
 * lib/deflate_compress.c:759-1396 (deflate_make_huffman_code and its helpers)
 * do, by one wave.
 */
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
/* W: the type of a weight.  A block of the small-buffer kernel has at most
 * 4097 tokens: its weights (and everything else these arrays hold: node
 * indices, depths) fit 16 bits, and the scratch of the litlen tree fits the
 * 4 KiB of the bit staging area, idle while the codes are built - M[] of a
 * tile of 1024 positions has no room for it */
template <int N, typename W = u32> struct huff_scratch {
	W A[N];	/* leaf weights, later hop pointers */
	W NW[N];	/* internal node weights, later depths */
	W P[N];	/* parent of each internal node */
	u32 cntI[40];	/* internal nodes per depth */
	u32 cnt[40];	/* leaves per depth (code lengths) */
	u32 start[16];	/* first index in sorted[] for each length */
	u32 nc[16];	/* next canonical codeword per length */
	u16 S[2 * N];	/* merge rounds: the items of a round in merged order */
};

/* a block end builds its codes in M[]: keys and sorted symbols in the first
 * 2 KiB, the litlen tree's scratch behind them - or, where M[] is a tile of
 * 1024 positions (small-buffer kernel), in the bit staging area with 16-bit
 * entries (a build that kept it in M[] wrote over the histogram) */
#ifdef LDA_SMALL
typedef huff_scratch<288, u16> huff_litlen_t;
#define HUFF_LITLEN(L) ((huff_litlen_t *)(L)->nxtA)
static_assert(sizeof(((struct deflate_lds *)0)->nxtA) >= sizeof(huff_litlen_t) &&
	      sizeof(((struct deflate_lds *)0)->M) >= 2048 && RING + 1 < 65536,
	      "the staging area holds the litlen tree's scratch, M[] the keys");
#else
typedef huff_scratch<288, u32> huff_litlen_t;
#define HUFF_LITLEN(L) ((huff_litlen_t *)((L)->M + 512))
static_assert(sizeof(((struct deflate_lds *)0)->M) >= 2048 + sizeof(huff_litlen_t),
	      "M[] holds the block-end scratch");
#endif

template <int N, typename W> static __device__ void
make_code(const u32 *freq, u32 n, u32 maxlen, u8 *lens, u16 *codes,
	  u16 *sorted, huff_scratch<N, W> *H, u32 used, bool presorted, u32 lane)
{
	PROF_DECL;
	PROF_START();
	/* a serial stretch (the merge is one lane): where other workgroups share
	 * the CU (small-buffer kernel) it gets its SIMD's issue slots first */
	__builtin_amdgcn_s_setprio(3);
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
		if (N == 288) PROF_MARK(13);
		/* The two-queue merge (leaves A[] ascending, nodes NW[] in creation
		 * order, ascending too) in ROUNDS by the whole wave.  The next node
		 * to be created weighs T = the sum of the two smallest items; every
		 * node created from now on weighs at least T, and every node that
		 * exists weighs at most T (the sums never decrease), so all items of
		 * at most T - the leaves up to T and all queued nodes - are consumed
		 * before any new node is, in merged order, two by two: that is one
		 * round.  An odd item out waits for the next round (where it is one of
		 * the two smallest).  The serial loop below does the same merges one
		 * at a time: 375 cycles each on one lane, 45 K cycles for a block's
		 * litlen tree; a round is ~250 wave instructions and a block takes
		 * 9-15 of them.  Weights that grow like Fibonacci numbers give one
		 * merge per round: after MERGE_ROUNDS rounds the serial loop takes
		 * over from where the rounds are. */
		u32 leaf = 0, node = 0, made = 0;	/* consumed leaves / nodes, created nodes */
#define MERGE_ROUNDS 40u
		if (m >= 24) {
			const u32 INF = 0x7FFFFFFFu;
			for (u32 round = 0; round < MERGE_ROUNDS && made + 1 < m; round++) {
				const u32 wl0 = leaf < m ? H->A[leaf] : INF;
				const u32 wl1 = leaf + 1 < m ? H->A[leaf + 1] : INF;
				const u32 wn0 = node < made ? H->NW[node] : INF;
				const u32 wn1 = node + 1 < made ? H->NW[node + 1] : INF;
				u32 T = wl0 + wl1;	/* (INF + INF does not wrap) */
				T = wl0 + wn0 < T ? wl0 + wn0 : T;
				T = wn0 + wn1 < T ? wn0 + wn1 : T;
				/* leaves of at most T: a prefix of what is left */
				u32 cl = 0;
				for (u32 b0 = leaf; b0 < m; b0 += 64) {
					const u32 i = b0 + lane;
					const u32 c = (u32)__builtin_popcountll(__ballot(i < m && H->A[i] <= T));
					cl += c;
					if (c < 64)
						break;
				}
				u32 cn = made - node;
				if ((cl + cn) & 1) {
					/* the last item of the merged order stays (a node
					 * follows a leaf of the same weight) */
					if (cn && (cl == 0 || H->NW[node + cn - 1] >= H->A[leaf + cl - 1]))
						cn--;
					else
						cl--;
				}
				const u32 tot = cl + cn;
				for (u32 k = lane; k < tot; k += 64)
					H->S[k] = 0xFFFF;
				wave_sync();
				/* a leaf's place: its index + the nodes that weigh less (the
				 * first 64 queued nodes sit in a register, one per lane, and
				 * are read lane by lane: no LDS round trip per node) */
				const u32 wnode = lane < cn ? H->NW[node + lane] : 0;
				const u32 cn64 = cn < 64 ? cn : 64;
				for (u32 i0 = 0; i0 < cl; i0 += 64) {
					const u32 i = i0 + lane;
					const u32 wv = i < cl ? H->A[leaf + i] : 0;
					u32 r = i;
					for (u32 j = 0; j < cn64; j++)
						r += bcast_lane(wnode, j) < wv;
					for (u32 j = 64; j < cn; j++)
						r += H->NW[node + j] < wv;
					if (i < cl)
						H->S[r] = (u16)i;
				}
				wave_sync();
				/* the nodes take the places left, in order */
				u32 nfree = 0;
				for (u32 k0 = 0; k0 < tot; k0 += 64) {
					const u32 k = k0 + lane;
					const bool fr = k < tot && H->S[k] == 0xFFFF;
					const u64 mk = __ballot(fr);
					if (fr)
						H->S[k] = (u16)(0x8000u | (nfree + rank_below(mk)));
					nfree += (u32)__builtin_popcountll(mk);
				}
				wave_sync();
				for (u32 q = lane; q < tot / 2; q += 64) {
					const u32 a = H->S[2 * q], b = H->S[2 * q + 1];
					const u32 wa = a & 0x8000 ? H->NW[node + (a & 0x7FFF)] : H->A[leaf + a];
					const u32 wb = b & 0x8000 ? H->NW[node + (b & 0x7FFF)] : H->A[leaf + b];
					if (a & 0x8000)
						H->P[node + (a & 0x7FFF)] = made + q;
					if (b & 0x8000)
						H->P[node + (b & 0x7FFF)] = made + q;
					H->NW[made + q] = wa + wb;
				}
				wave_sync();
				leaf += cl;
				node += cn;
				made += tot / 2;
			}
		}
		if (lane == 0 && made + 1 < m) {
			/* one merge at a time; heads cached in registers (a variant that
			 * also prefetched the following entries had more instructions on
			 * this single-lane path and was slower) */
			u32 wl = leaf < m ? H->A[leaf] : 0xFFFFFFFFu;
			u32 wn = node < made ? H->NW[node] : 0xFFFFFFFFu;
			for (u32 k = made; k + 1 < m; k++) {
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
		if (N == 288) PROF_MARK(14);
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
		if (N == 288) PROF_MARK(15);
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
	if (N == 288) PROF_MARK(17);
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
	__builtin_amdgcn_s_setprio(0);
}
