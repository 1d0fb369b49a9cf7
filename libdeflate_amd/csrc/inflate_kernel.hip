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
 * Mapping: ONE 64-lane wavefront per stream, one wave per workgroup.
 *
 *   - the compressed bytes are staged through a 2 KiB LDS ring, filled 1 KiB
 *     at a time with coalesced, 16-byte-aligned loads by all 64 lanes;
 *   - lane 0 is the entropy decoder: 64-bit bit buffer, one LDS table look-up
 *     per symbol (10-bit litlen table, 8-bit offset table, canonical
 *     bit-serial fallback for the rare longer codewords - no subtables);
 *     it emits up to 64 tokens {literal | (length, distance)} into LDS;
 *   - all 64 lanes then apply the batch: wave prefix-sum of token lengths
 *     gives every token its output position, literals are scattered in one
 *     step, matches are copied 64 bytes per step;
 *   - output goes through an LDS window ring (the most recent W bytes) that
 *     is drained to HBM in coalesced 16-byte stores; back-references that
 *     reach behind the window are read back from HBM (already drained);
 *   - decode tables are built by all lanes (each lane canonically decodes its
 *     own table indices), only the code-length run decoding is serial.
 *
 * Result codes follow the reference bit for bit, including the implicit
 * zero-padding rule: the reference fails when a refill would need a 9th
 * zero byte (lib/deflate_decompress.c:236-254); with whole-byte refills to
 * 56..63 bits that is exactly "bits consumed > 8*in_nbytes + 8 at a refill
 * point", which is what is tested here at the same program points.
 */
#include "device_common.h"
#include "kernels.h"

#define LIT_TB 10		/* litlen primary table bits */
#define OFF_TB 8		/* offset primary table bits */
#define IN_RING 2048u
#define IN_HALF 1024u
#define WBITS 13		/* output window ring: 8 KiB */
#define WSIZE (1u << WBITS)
#define WMASK (WSIZE - 1)
#define OUT_CAP 2048u		/* max bytes one token batch may produce */
#define BATCH 64

/* table entry: [3:0] codeword len (0 = long codeword, use canonical path)
 *              [7:4] extra bits   [8] literal   [9] end of block
 *              [31:16] literal value / length base / offset base */
#define E_LIT 0x100u
#define E_EOB 0x200u

struct canon {
	u16 count[16];
	u16 first[16];	/* first codeword of each length */
	u16 index[16];	/* index into sorted[] of first symbol of each length */
};

struct inflate_lds {
	u32 lit_tab[1 << LIT_TB];
	u32 off_tab[1 << OFF_TB];
	u32 pre_tab[128];
	u32 tok[BATCH];
	struct canon lit, off, pre;
	u16 lit_sorted[288];
	u16 off_sorted[32];
	u16 pre_sorted[20];
	u8 lens[288 + 32 + 138 + 6];
	u8 pre_lens[20];
	u8 in_ring[IN_RING + 16];
	u8 win[WSIZE];
};

/* lib/deflate_decompress.c:555-588 (285..287 -> 258) and :615-628 */
__constant__ u16 c_len_base[31] = {
	3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
	59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 258, 258 };
__constant__ u8 c_len_extra[31] = {
	0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
	4, 5, 5, 5, 5, 0, 0, 0 };
__constant__ u16 c_off_base[32] = {
	1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385,
	513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385,
	24577, 24577, 24577 };
__constant__ u8 c_off_extra[32] = {
	0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10,
	10, 11, 11, 12, 12, 13, 13, 13, 13 };
__constant__ u8 c_pre_perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4,
				   12, 3, 13, 2, 14, 1, 15 };

enum { KIND_LITLEN = 0, KIND_OFFSET = 1, KIND_PRECODE = 2 };

static __device__ __forceinline__ u32 make_entry(int kind, u32 sym, u32 len)
{
	if (kind == KIND_LITLEN) {
		if (sym < 256)
			return (sym << 16) | E_LIT | len;
		if (sym == 256)
			return E_EOB | len;
		return ((u32)c_len_base[sym - 257] << 16) |
		       ((u32)c_len_extra[sym - 257] << 4) | len;
	}
	if (kind == KIND_OFFSET)
		return ((u32)c_off_base[sym] << 16) |
		       ((u32)c_off_extra[sym] << 4) | len;
	return (sym << 16) | len;
}

/*
 * Build one decode table from lens[0..n).  Whole wave.  Returns false when
 * the code is invalid by the rules of lib/deflate_decompress.c:804-853.
 * Counting/sorting is done by lane 0 (<= 288 steps); the table itself is
 * filled by all lanes, each canonically decoding its own indices.
 */
static __device__ bool
build_table(int kind, const u8 *lens, u32 n, u32 tb, u32 *tab,
	    struct canon *cn, u16 *sorted, u32 lane)
{
	u32 status = 0;	/* 0 ok-complete, 1 invalid, 2 incomplete(single) */
	u32 single_entry = 0;

	if (lane < 16)
		cn->count[lane] = 0;
	wave_sync();
	if (lane == 0) {
		u32 maxlen = (kind == KIND_PRECODE) ? 7 : 15;
		u32 used = 0, idx = 0;

		for (u32 s = 0; s < n; s++)
			cn->count[lens[s]]++;
		cn->count[0] = 0;
		while (maxlen > 1 && cn->count[maxlen] == 0)
			maxlen--;
		u32 code = 0;
		for (u32 l = 1; l <= 15; l++) {
			u32 c = cn->count[l];
			cn->first[l] = (u16)code;
			cn->index[l] = (u16)idx;
			code = (code + c) << 1;
			idx += c;
			if (l <= maxlen)
				used = (used << 1) + c;
		}
		/* sorted[] by (len, sym): offsets via index[] copy */
		u16 next[16];
		for (u32 l = 1; l <= 15; l++)
			next[l] = cn->index[l];
		for (u32 s = 0; s < n; s++) {
			u32 l = lens[s];
			if (l)
				sorted[next[l]++] = (u16)s;
		}
		if (used > (1u << maxlen)) {
			status = 1;
		} else if (used < (1u << maxlen)) {
			u32 sym;
			if (used == 0) {
				sym = 0;
				status = 2;
			} else if (used != (1u << (maxlen - 1)) ||
				   cn->count[1] != 1) {
				status = 1;
				sym = 0;
			} else {
				sym = sorted[0];
				status = 2;
			}
			single_entry = make_entry(kind, sym, 1);
		}
	}
	status = bcast_first(status);
	single_entry = bcast_first(single_entry);
	if (status == 1)
		return false;
	wave_sync();
	if (status == 2) {
		for (u32 e = lane; e < (1u << tb); e += 64)
			tab[e] = single_entry;
		wave_sync();
		return true;
	}
	for (u32 e = lane; e < (1u << tb); e += 64) {
		u32 code = 0, entry = 0;
		for (u32 l = 1; l <= tb; l++) {
			code = (code << 1) | ((e >> (l - 1)) & 1);
			u32 rel = code - cn->first[l];
			if (rel < cn->count[l]) {
				entry = make_entry(kind, sorted[cn->index[l] + rel], l);
				break;
			}
		}
		tab[e] = entry;	/* 0 -> codeword longer than tb bits */
	}
	wave_sync();
	return true;
}

/* canonical bit-serial decode for codewords longer than the table */
static __device__ __forceinline__ u32
decode_long(int kind, const struct canon *cn, const u16 *sorted, u64 bits)
{
	u32 code = 0;
	for (u32 l = 1; l <= 15; l++) {
		code = (code << 1) | (u32)((bits >> (l - 1)) & 1);
		u32 rel = code - cn->first[l];
		if (rel < cn->count[l])
			return make_entry(kind, sorted[cn->index[l] + rel], l);
	}
	return E_EOB | 15;	/* unreachable for a complete code */
}

struct instream {
	const u8 *base_al;	/* 16-byte aligned address at/below the stream */
	u32 shift;		/* stream byte 0 is at base_al[shift] */
	u64 n;			/* stream length in bytes */
	s32 tag[2];		/* which 1 KiB block each ring half holds */
};

/* whole wave: make the ring hold 1 KiB block 'blk' of the virtual stream */
static __device__ __forceinline__ void
load_in_block(struct inflate_lds *L, struct instream *in, u32 blk, u32 lane)
{
	u32 slot = blk & 1;
	u64 v = (u64)blk * IN_HALF + lane * 16;	/* virtual position */
	u64 end = in->shift + in->n;		/* virtual end of stream */
	uint4 w = make_uint4(0, 0, 0, 0);

	if (v < end)
		w = *(const uint4 *)(in->base_al + v);
	if (v + 16 > end) {	/* zero the bytes past the end of the stream */
		u32 keep = v < end ? (u32)(end - v) : 0;
		u32 ww[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
		for (int i = 0; i < 4; i++) {
			u32 kb = keep > (u32)(4 * i) ? keep - 4 * i : 0;
			if (kb < 4)
				ww[i] &= kb ? (0xFFFFFFFFu >> (32 - 8 * kb)) : 0;
		}
		w = make_uint4(ww[0], ww[1], ww[2], ww[3]);
	}
	*(uint4 *)&L->in_ring[slot * IN_HALF + lane * 16] = w;
	if (slot == 0 && lane == 0)	/* mirror for reads that wrap */
		*(uint4 *)&L->in_ring[IN_RING] = w;
	in->tag[slot] = (s32)blk;
}

static __device__ __forceinline__ void
ensure_input(struct inflate_lds *L, struct instream *in, u64 vpos, u32 lane)
{
	u32 blk = (u32)(vpos / IN_HALF);
	bool any = false;
	if (in->tag[blk & 1] != (s32)blk) {
		load_in_block(L, in, blk, lane);
		any = true;
	}
	if (in->tag[(blk + 1) & 1] != (s32)(blk + 1)) {
		load_in_block(L, in, blk + 1, lane);
		any = true;
	}
	if (any)
		wave_sync();
}

static __device__ __forceinline__ u64 ring_load64(const struct inflate_lds *L,
						  u64 vpos)
{
	u64 v;
	__builtin_memcpy(&v, &L->in_ring[vpos & (IN_RING - 1)], 8);
	return v;
}

/* bit reader state, meaningful in lane 0 */
struct bitreader {
	u64 buf;
	u32 cnt;	/* valid bits in buf (<= 63) */
	u64 vpos;	/* virtual position of the next byte to load */
};

#define BR_REFILL(L, br)                                                     \
	do {                                                                 \
		(br).buf |= ring_load64(L, (br).vpos) << (br).cnt;           \
		(br).vpos += (63 - (br).cnt) >> 3;                           \
		(br).cnt |= 56;                                              \
	} while (0)
#define BR_CONSUME(br, k)                                                    \
	do {                                                                 \
		(br).buf >>= (k);                                            \
		(br).cnt -= (k);                                             \
	} while (0)
/* bits consumed from the stream so far */
#define BR_CONSUMED(br, in) (8 * ((br).vpos - (in).shift) - (br).cnt)

/* drain window bytes [from, to) to the output buffer; whole wave */
static __device__ void
flush_window(const struct inflate_lds *L, u8 *outp, u64 from, u64 to, u32 lane)
{
	if (from >= to)
		return;
	u64 a = from;
	/* head up to the first 16-byte aligned global address */
	u64 head_end = from + ((0 - (uintptr_t)(outp + from)) & 15);
	if (head_end > to)
		head_end = to;
	if (a + lane < head_end)
		outp[a + lane] = L->win[(a + lane) & WMASK];
	a = head_end;
	if (((uintptr_t)outp & 15) == 0) {
		/* ring index and global address are congruent mod 16 */
		for (; a + 1024 <= to; a += 1024) {
			u64 p = a + lane * 16;
			*(uint4 *)(outp + p) = *(const uint4 *)&L->win[p & WMASK];
		}
		u64 p = a + lane * 16;
		if (p + 16 <= to)
			*(uint4 *)(outp + p) = *(const uint4 *)&L->win[p & WMASK];
		a += ((to - a) / 16) * 16;
	}
	for (u64 p = a + lane; p < to; p += 64)
		outp[p] = L->win[p & WMASK];
}

extern "C" __global__ void __launch_bounds__(64)
lda_inflate_batch_kernel(u64 n_chunks, int format,
			 const u8 *__restrict__ in_base,
			 const u64 *__restrict__ in_offsets,
			 const u64 *__restrict__ in_nbytes,
			 u8 *__restrict__ out_base,
			 const u64 *__restrict__ out_offsets,
			 const u64 *__restrict__ out_avail_arr,
			 s32 *__restrict__ results,
			 u64 *__restrict__ actual_in,	/* stream-relative, incl. header */
			 u64 *__restrict__ actual_out)
{
	__shared__ struct inflate_lds Ls;
	struct inflate_lds *L = &Ls;
	const u32 lane = threadIdx.x;
	const u64 c = blockIdx.x;
	PROF_DECL;
	PROF_START();

	if (c >= n_chunks)
		return;

	const u8 *inp = in_base + in_offsets[c];
	u64 in_n = in_nbytes[c];
	u8 *outp = out_base + out_offsets[c];
	const u64 out_avail = out_avail_arr[c];
	u32 hdr = 0;		/* container header bytes before the deflate data */
	s32 result = LDA_SUCCESS;

	/* ---- container header (uniform scalar code, few bytes) ---- */
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
	if (result != LDA_SUCCESS) {
		if (lane == 0) {
			results[c] = result;
			actual_in[c] = 0;
			actual_out[c] = 0;
		}
		return;
	}
	inp += hdr;

	struct instream in;
	in.base_al = (const u8 *)((uintptr_t)inp & ~(uintptr_t)15);
	in.shift = (u32)((uintptr_t)inp & 15);
	in.n = in_n;
	in.tag[0] = in.tag[1] = -1;

	struct bitreader br;
	br.buf = 0;
	br.cnt = 0;
	br.vpos = in.shift;

	const u64 limit_bits = 8 * in_n + 8;	/* see header comment */
	u64 out_pos = 0;	/* bytes produced */
	u64 flushed = 0;	/* bytes drained to HBM */
	u32 final_block = 0;

	do {
		/* ---------------- block header (lane 0 reads) ---------------- */
		ensure_input(L, &in, bcast64(br.vpos), lane);
		u32 btype = 0, nlit = 0, noff = 0, err = 0;
		u64 stored_pos = 0;
		u32 stored_len = 0;
		if (lane == 0) {
			BR_REFILL(L, br);
			if (BR_CONSUMED(br, in) > limit_bits)
				err = LDA_BAD_DATA;
			final_block = (u32)br.buf & 1;
			btype = ((u32)br.buf >> 1) & 3;
			if (!err && btype == 0) {
				/* stored: decompress_template.h:247-285 */
				BR_CONSUME(br, 3);
				u64 cons = BR_CONSUMED(br, in);
				u64 pos = (cons + 7) / 8;
				if (pos > in_n || in_n - pos < 4) {
					err = LDA_BAD_DATA;
				} else {
					u32 len = inp[pos] | ((u32)inp[pos + 1] << 8);
					u32 nlen = inp[pos + 2] | ((u32)inp[pos + 3] << 8);
					pos += 4;
					if (len != (nlen ^ 0xFFFF))
						err = LDA_BAD_DATA;
					else if (len > out_avail - out_pos)
						err = LDA_INSUFFICIENT_SPACE;
					else if (len > in_n - pos)
						err = LDA_BAD_DATA;
					stored_pos = pos;
					stored_len = len;
				}
			} else if (!err && btype == 3) {
				err = LDA_BAD_DATA;
			} else if (!err && btype == 2) {
				/* dynamic header: decompress_template.h:85-146 */
				nlit = 257 + (((u32)br.buf >> 3) & 31);
				noff = 1 + (((u32)br.buf >> 8) & 31);
				u32 npre = 4 + (((u32)br.buf >> 13) & 15);
				for (u32 i = 0; i < 19; i++)
					L->pre_lens[i] = 0;
				L->pre_lens[c_pre_perm[0]] = ((u32)br.buf >> 17) & 7;
				BR_CONSUME(br, 20);
				BR_REFILL(L, br);
				if (BR_CONSUMED(br, in) > limit_bits)
					err = LDA_BAD_DATA;
				for (u32 i = 1; i < npre; i++) {
					L->pre_lens[c_pre_perm[i]] = (u32)br.buf & 7;
					BR_CONSUME(br, 3);
				}
			} else if (!err) {
				BR_CONSUME(br, 3);	/* static */
			}
		}
		err = bcast_first(err);
		btype = bcast_first(btype);
		final_block = bcast_first(final_block);
		if (err) {
			result = (s32)err;
			break;
		}

		if (btype == 0) {
			/* copy the stored bytes through the window, wave-wide */
			stored_len = bcast_first(stored_len);
			u64 sp = bcast64(stored_pos);
			u32 done = 0;
			while (done < stored_len) {
				u32 piece = stored_len - done;
				if (piece > OUT_CAP)
					piece = OUT_CAP;
				if (out_pos - flushed > WSIZE - OUT_CAP) {
					flush_window(L, outp, flushed, out_pos, lane);
					flushed = out_pos;
				}
				for (u32 j = lane; j < piece; j += 64)
					L->win[(out_pos + j) & WMASK] = inp[sp + done + j];
				wave_sync();
				out_pos += piece;
				done += piece;
			}
			/* restart the bit reader after the stored bytes */
			br.buf = 0;
			br.cnt = 0;
			br.vpos = in.shift + sp + stored_len;
			continue;
		}

		if (btype == 2) {
			nlit = bcast_first(nlit);
			noff = bcast_first(noff);
			wave_sync();
			if (!build_table(KIND_PRECODE, L->pre_lens, 19, 7,
					 L->pre_tab, &L->pre, L->pre_sorted,
					 lane)) {
				result = LDA_BAD_DATA;
				break;
			}
			/* code length runs: decompress_template.h:150-245.
			 * The reference refills only when fewer than 14 bits are
			 * left; 'loaded' reproduces its refill points. */
			u32 herr = 0;
			if (lane == 0) {
				u32 i = 0, total = nlit + noff;
				/* br.cnt equals the reference's bitsleft here: both
				 * were topped up at the same points, and a top-up only
				 * depends on the consumed bit count */
				do {
					if (br.cnt < 14) {
						BR_REFILL(L, br);
						if (BR_CONSUMED(br, in) > limit_bits) {
							herr = LDA_BAD_DATA;
							break;
						}
					}
					u32 e = L->pre_tab[(u32)br.buf & 127];
					BR_CONSUME(br, e & 15);
					u32 presym = e >> 16;
					if (presym < 16) {
						L->lens[i++] = (u8)presym;
						continue;
					}
					u32 rep, val = 0;
					if (presym == 16) {
						if (i == 0) {
							herr = LDA_BAD_DATA;
							break;
						}
						val = L->lens[i - 1];
						rep = 3 + ((u32)br.buf & 3);
						BR_CONSUME(br, 2);
					} else if (presym == 17) {
						rep = 3 + ((u32)br.buf & 7);
						BR_CONSUME(br, 3);
					} else {
						rep = 11 + ((u32)br.buf & 127);
						BR_CONSUME(br, 7);
					}
					for (u32 k = 0; k < rep; k++)
						L->lens[i + k] = (u8)val;
					i += rep;
				} while (i < total);
				if (!herr && i != total)
					herr = LDA_BAD_DATA;
			}
			herr = bcast_first(herr);
			if (herr) {
				result = (s32)herr;
				break;
			}
		} else {
			/* static codes: decompress_template.h:313-326 */
			for (u32 i = lane; i < 320; i += 64)
				L->lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 :
					     i < 288 ? 8 : 5;
			nlit = 288;
			noff = 32;
		}
		wave_sync();
		if (!build_table(KIND_OFFSET, L->lens + nlit, noff, OFF_TB,
				 L->off_tab, &L->off, L->off_sorted, lane) ||
		    !build_table(KIND_LITLEN, L->lens, nlit, LIT_TB, L->lit_tab,
				 &L->lit, L->lit_sorted, lane)) {
			result = LDA_BAD_DATA;
			break;
		}

		PROF_MARK(0);	/* header + tables */
		/* ---------------- token batches ---------------- */
		u32 eob = 0;
		while (!eob) {
			PROF_MARK(3);
			ensure_input(L, &in, bcast64(br.vpos), lane);
			if (out_pos - flushed > WSIZE - OUT_CAP) {
				flush_window(L, outp, flushed, out_pos, lane);
				flushed = out_pos;
				/* make the drained bytes visible to later far reads */
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			}
			PROF_MARK(1);	/* input staging + flush */
			u32 ntok = 0, berr = 0, produced = 0;
			if (lane == 0) {
				u64 opos = out_pos;
				/* generic_loop: decompress_template.h:680-738 */
				while (ntok < BATCH && produced < OUT_CAP - 258) {
					BR_REFILL(L, br);
					if (BR_CONSUMED(br, in) > limit_bits) {
						berr = LDA_BAD_DATA;
						break;
					}
					u32 e = L->lit_tab[(u32)br.buf & ((1u << LIT_TB) - 1)];
					if ((e & 15) == 0)
						e = decode_long(KIND_LITLEN, &L->lit,
								L->lit_sorted, br.buf);
					if (e & E_LIT) {
						BR_CONSUME(br, e & 15);
						if (opos == out_avail) {
							berr = LDA_INSUFFICIENT_SPACE;
							break;
						}
						L->tok[ntok++] = e & 0x00FF0000u;
						opos++;
						produced++;
						continue;
					}
					if (e & E_EOB) {
						BR_CONSUME(br, e & 15);
						eob = 1;
						break;
					}
					u32 cl = e & 15, xb = (e >> 4) & 15;
					u32 length = (e >> 16) +
						(((u32)(br.buf >> cl)) & ((1u << xb) - 1));
					BR_CONSUME(br, cl + xb);
					if (length > out_avail - opos) {
						berr = LDA_INSUFFICIENT_SPACE;
						break;
					}
					u32 e2 = L->off_tab[(u32)br.buf & ((1u << OFF_TB) - 1)];
					if ((e2 & 15) == 0)
						e2 = decode_long(KIND_OFFSET, &L->off,
								 L->off_sorted, br.buf);
					cl = e2 & 15;
					xb = (e2 >> 4) & 15;
					u32 dist = (e2 >> 16) +
						(((u32)(br.buf >> cl)) & ((1u << xb) - 1));
					BR_CONSUME(br, cl + xb);
					if (dist > opos) {
						berr = LDA_BAD_DATA;
						break;
					}
					L->tok[ntok++] = (length << 16) | dist;
					opos += length;
					produced += length;
				}
			}
			ntok = bcast_first(ntok);
			berr = bcast_first(berr);
			eob = bcast_first(eob);
			produced = bcast_first(produced);
			if (berr) {
				result = (s32)berr;
				break;
			}
			wave_sync();

			PROF_MARK(2);	/* serial decode */
			/* ---- apply the batch with all lanes ---- */
			u32 t = lane < ntok ? L->tok[lane] : 0;
			u32 dist = t & 0xFFFF;
			u32 len = lane < ntok ? (dist ? (t >> 16) : 1) : 0;
			u32 incl = wave_scan_incl(len);
			u64 dst = out_pos + (incl - len);
			/* positions >= lds_lo are guaranteed to be in the window */
			s64 lds_lo = (s64)(out_pos + produced) - (s64)WSIZE;

			if (lane < ntok && dist == 0)
				L->win[dst & WMASK] = (u8)(t >> 16);
			u64 mm = __ballot(lane < ntok && dist != 0);
			while (mm) {
				u32 tl = (u32)__builtin_ctzll(mm);
				mm &= mm - 1;
				u32 mlen = bcast_lane(len, tl);
				u32 mdist = bcast_lane(dist, tl);
				u64 mdst = ((u64)bcast_lane((u32)(dst >> 32), tl) << 32) |
					   bcast_lane((u32)dst, tl);
				u64 src0 = mdst - mdist;
				wave_sync();
				if (mdist >= 64) {
					for (u32 j = lane; j < mlen; j += 64) {
						u64 sp = src0 + j;
						u8 b = ((s64)sp >= lds_lo) ?
							L->win[sp & WMASK] : outp[sp];
						L->win[(mdst + j) & WMASK] = b;
						if (mdist < mlen)
							wave_sync();
					}
				} else {
					/* periodic fill: byte j repeats byte j % dist of
					 * the 'dist' bytes that precede the match */
					u32 inv = (0x100000u + mdist - 1) / mdist;
					for (u32 j = lane; j < mlen; j += 64) {
						u32 q = (j * inv) >> 20;
						u64 sp = src0 + (j - q * mdist);
						u8 b = ((s64)sp >= lds_lo) ?
							L->win[sp & WMASK] : outp[sp];
						L->win[(mdst + j) & WMASK] = b;
					}
				}
			}
			wave_sync();
			out_pos += produced;
		}
		if (result != LDA_SUCCESS)
			break;
	} while (!final_block);

	if (result == LDA_SUCCESS) {
		flush_window(L, outp, flushed, out_pos, lane);
		/* epilogue: decompress_template.h:740-771 */
		u32 e = 0;
		u64 ain = 0;
		if (lane == 0) {
			u64 cons = BR_CONSUMED(br, in);
			ain = (cons + 7) / 8;
			if (ain > in_n)
				e = LDA_BAD_DATA;
		}
		e = bcast_first(e);
		if (e)
			result = (s32)e;
		else if (lane == 0)
			actual_in[c] = hdr + ain;
	}
	if (lane == 0) {
		results[c] = result;
		actual_out[c] = result == LDA_SUCCESS ? out_pos : 0;
		if (result != LDA_SUCCESS)
			actual_in[c] = 0;
	}
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
