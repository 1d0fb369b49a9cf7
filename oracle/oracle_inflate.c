/*
 * oracle_inflate.c - raw DEFLATE / zlib / gzip decoding restated for the CPU.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What is restated: the OBSERVABLE behaviour of lib/decompress_template.h:44-772
 * (bytes produced, enum libdeflate_result, actual_in / actual_out), of the code
 * validity rules in lib/deflate_decompress.c:721-1004, and of the wrappers in
 * lib/gzip_decompress.c:31-144 and lib/zlib_decompress.c:31-104.
 *
 * What is deliberately different: symbols are decoded one bit at a time from
 * canonical (count[], first-code) arrays instead of lookup tables, and there
 * is a single decode loop.  The reference has a "fastloop" and a "generic
 * loop"; the fastloop only runs while >= 299 output and >= 25 input bytes
 * remain (lib/deflate_decompress.c:280-297), i.e. where neither
 * INSUFFICIENT_SPACE nor the overread BAD_DATA can fire, so its results are
 * those of the generic loop (decompress_template.h:680-738) which is what is
 * restated here.
 *
 * The one piece of hidden state that IS observable is how many bits the
 * reference has *loaded* (it fails when a refill would need a 9th implicit
 * zero byte, lib/deflate_decompress.c:236-254).  It is modelled exactly by
 * 'loaded' below and refilled at the same program points as the reference.
 */
#include <string.h>
#include "oracle.h"

#define MAXBITS 15

struct bits {
	const uint8_t *in;
	size_t in_n;
	uint64_t consumed;	/* bits consumed so far */
	uint64_t loaded;	/* bits the reference would have in/through its
				 * bit buffer; always a multiple of 8 */
};

/* One bit of the stream; bits past the end read as zero
 * (lib/deflate_decompress.c:230-235 "leaving the bits zeroed"). */
static unsigned peekbit(const struct bits *b, uint64_t pos)
{
	size_t byte = (size_t)(pos >> 3);

	if (byte >= b->in_n)
		return 0;
	return (b->in[byte] >> (pos & 7)) & 1;
}

static uint32_t getbits(struct bits *b, unsigned n)
{
	uint32_t v = 0;
	unsigned i;

	for (i = 0; i < n; i++)
		v |= (uint32_t)peekbit(b, b->consumed + i) << i;
	b->consumed += n;
	return v;
}

static unsigned bitsleft(const struct bits *b)
{
	return (unsigned)(b->loaded - b->consumed);
}

/*
 * REFILL_BITS() (lib/deflate_decompress.c:236-254, branchless form :206-212):
 * top the buffer up to 56..63 bits in whole bytes.  Once the real input is
 * exhausted each further byte counts as an overread; more than 8 of them is
 * BAD_DATA.  Returns 0 on that failure.
 */
static int refill(struct bits *b)
{
	unsigned left = bitsleft(b);

	if (left < 56)
		b->loaded = b->consumed + 56 + (left & 7);
	if (b->loaded > 8 * (uint64_t)b->in_n + 64)
		return 0;
	return 1;
}

/* canonical code in "puff" style: count per length + symbols sorted by
 * (length, symbol) - the ordering lib/deflate_decompress.c:783-784 produces */
struct code {
	uint16_t count[MAXBITS + 1];
	uint16_t sym[288];
	int single;	/* -1: normal; else every codeword decodes (1 bit) to it */
};

/*
 * Validity rules of build_decode_table (lib/deflate_decompress.c:804-853):
 * over-subscribed -> invalid; incomplete -> valid only if there are no
 * codewords at all (then symbol 0, 1 bit) or exactly one codeword and it has
 * length 1 (then that symbol, 1 bit, for both bit values).
 */
static int build_code(struct code *c, const uint8_t *lens, unsigned n)
{
	unsigned offs[MAXBITS + 2];
	unsigned sym, len;
	uint32_t used = 0;
	unsigned maxlen = MAXBITS;

	memset(c->count, 0, sizeof(c->count));
	for (sym = 0; sym < n; sym++)
		c->count[lens[sym]]++;
	while (maxlen > 1 && c->count[maxlen] == 0)
		maxlen--;
	for (len = 1; len <= maxlen; len++)
		used = (used << 1) + c->count[len];
	offs[1] = 0;
	for (len = 1; len < MAXBITS; len++)
		offs[len + 1] = offs[len] + c->count[len];
	for (sym = 0; sym < n; sym++)
		if (lens[sym])
			c->sym[offs[lens[sym]]++] = (uint16_t)sym;
	c->single = -1;
	if (used > (1u << maxlen))
		return 0;
	if (used < (1u << maxlen)) {
		if (used == 0)
			c->single = 0;
		else if (used != (1u << (maxlen - 1)) || c->count[1] != 1)
			return 0;
		else
			c->single = c->sym[0];
	}
	return 1;
}

static unsigned decode_sym(struct bits *b, const struct code *c)
{
	int code = 0, first = 0, index = 0;
	unsigned len;

	if (c->single >= 0) {
		b->consumed += 1;
		return (unsigned)c->single;
	}
	for (len = 1; len <= MAXBITS; len++) {
		int count = c->count[len];

		code |= (int)peekbit(b, b->consumed + len - 1);
		if (code - count < first) {
			b->consumed += len;
			return c->sym[index + (code - first)];
		}
		index += count;
		first += count;
		first <<= 1;
		code <<= 1;
	}
	/* unreachable for a complete code */
	b->consumed += MAXBITS;
	return 0;
}

/* lib/deflate_decompress.c:555-588: symbols 285..287 all mean length 258 */
static const uint16_t len_base[31] = {
	3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
	59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 258, 258 };
static const uint8_t len_extra[31] = {
	0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
	4, 5, 5, 5, 5, 0, 0, 0 };
/* lib/deflate_decompress.c:615-628: symbols 30, 31 alias symbol 29 */
static const uint16_t off_base[32] = {
	1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385,
	513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385,
	24577, 24577, 24577 };
static const uint8_t off_extra[32] = {
	0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10,
	10, 11, 11, 12, 12, 13, 13, 13, 13 };

/*
 * Block map (test infrastructure for the parallel single-stream decoder's
 * block finder): while a trace is installed, every block header the decoder
 * reaches is recorded - bit position of its BFINAL bit, BTYPE, bytes of
 * output before it.
 */
static __thread struct oracle_block *trace_blocks;
static __thread size_t trace_cap, trace_n;

size_t oracle_deflate_block_map(const void *in, size_t in_nbytes, void *out,
				size_t out_avail, struct oracle_block *blocks,
				size_t cap, int *result)
{
	int r;

	trace_blocks = blocks;
	trace_cap = cap;
	trace_n = 0;
	r = oracle_deflate_decompress(in, in_nbytes, out, out_avail, NULL,
				      &(size_t){ 0 });
	trace_blocks = NULL;
	if (result)
		*result = r;
	return trace_n;
}

int oracle_deflate_decompress(const void *in_, size_t in_nbytes,
			      void *out_, size_t out_avail,
			      size_t *actual_in, size_t *actual_out)
{
	static const uint8_t perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
					  11, 4, 12, 3, 13, 2, 14, 1, 15 };
	struct bits b = { (const uint8_t *)in_, in_nbytes, 0, 0 };
	uint8_t *out = (uint8_t *)out_;
	size_t outpos = 0;
	struct code pre, lit, off;
	uint8_t lens[288 + 32 + 138];
	int final;

	do {
		unsigned type;

		/* next_block: decompress_template.h:72-83 */
		if (!refill(&b))
			return ORACLE_BAD_DATA;
		if (trace_blocks) {
			if (trace_n < trace_cap) {
				trace_blocks[trace_n].bit = b.consumed;
				trace_blocks[trace_n].out_pos = outpos;
			}
		}
		final = (int)getbits(&b, 1);
		type = getbits(&b, 2);
		if (trace_blocks) {
			if (trace_n < trace_cap)
				trace_blocks[trace_n].type = type | (final ? 4u : 0u);
			trace_n++;
		}

		if (type == 0) {
			/* stored: decompress_template.h:247-285 */
			size_t pos;
			unsigned len, nlen;

			/* "SAFETY_CHECK(overread_count <= (bitsleft >> 3))" */
			{
				uint64_t over = b.loaded / 8 > b.in_n ?
					b.loaded / 8 - b.in_n : 0;
				if (over > bitsleft(&b) / 8)
					return ORACLE_BAD_DATA;
			}
			pos = (size_t)((b.consumed + 7) / 8);
			if (b.in_n - pos < 4)
				return ORACLE_BAD_DATA;
			len = b.in[pos] | (b.in[pos + 1] << 8);
			nlen = b.in[pos + 2] | (b.in[pos + 3] << 8);
			pos += 4;
			if (len != (nlen ^ 0xFFFF))
				return ORACLE_BAD_DATA;
			if (len > out_avail - outpos)
				return ORACLE_INSUFFICIENT_SPACE;
			if (len > b.in_n - pos)
				return ORACLE_BAD_DATA;
			memcpy(out + outpos, b.in + pos, len);
			outpos += len;
			pos += len;
			b.consumed = b.loaded = 8 * (uint64_t)pos;
			continue;
		}
		if (type == 3)
			return ORACLE_BAD_DATA; /* decompress_template.h:290 */

		if (type == 2) {
			/* dynamic header: decompress_template.h:85-245 */
			unsigned nlit = 257 + getbits(&b, 5);
			unsigned noff = 1 + getbits(&b, 5);
			unsigned npre = 4 + getbits(&b, 4);
			uint8_t prelens[19];
			unsigned i;

			memset(prelens, 0, sizeof(prelens));
			/* 64-bit build: first len merged, then REFILL_BITS()
			 * (:120-131) */
			prelens[perm[0]] = (uint8_t)getbits(&b, 3);
			if (!refill(&b))
				return ORACLE_BAD_DATA;
			for (i = 1; i < npre; i++)
				prelens[perm[i]] = (uint8_t)getbits(&b, 3);
			if (!build_code(&pre, prelens, 19))
				return ORACLE_BAD_DATA;
			i = 0;
			do {
				unsigned presym, rep;

				if (bitsleft(&b) < 7 + 7 && !refill(&b))
					return ORACLE_BAD_DATA;
				presym = decode_sym(&b, &pre);
				if (presym < 16) {
					lens[i++] = (uint8_t)presym;
				} else if (presym == 16) {
					if (i == 0)
						return ORACLE_BAD_DATA;
					rep = 3 + getbits(&b, 2);
					memset(&lens[i], lens[i - 1], rep);
					i += rep;
				} else if (presym == 17) {
					rep = 3 + getbits(&b, 3);
					memset(&lens[i], 0, rep);
					i += rep;
				} else {
					rep = 11 + getbits(&b, 7);
					memset(&lens[i], 0, rep);
					i += rep;
				}
			} while (i < nlit + noff);
			if (i != nlit + noff)
				return ORACLE_BAD_DATA;
			/* offset table first (:331-332) - order only matters
			 * for which failure is reported, and both are BAD_DATA */
			if (!build_code(&off, lens + nlit, noff))
				return ORACLE_BAD_DATA;
			if (!build_code(&lit, lens, nlit))
				return ORACLE_BAD_DATA;
		} else {
			/* static codes: decompress_template.h:313-326 */
			unsigned i;

			for (i = 0; i < 144; i++) lens[i] = 8;
			for (; i < 256; i++) lens[i] = 9;
			for (; i < 280; i++) lens[i] = 7;
			for (; i < 288; i++) lens[i] = 8;
			for (; i < 320; i++) lens[i] = 5;
			build_code(&off, lens + 288, 32);
			build_code(&lit, lens, 288);
		}

		/* generic_loop: decompress_template.h:680-738 */
		for (;;) {
			unsigned sym, length, offset;

			if (!refill(&b))
				return ORACLE_BAD_DATA;
			sym = decode_sym(&b, &lit);
			if (sym < 256) {
				if (outpos == out_avail)
					return ORACLE_INSUFFICIENT_SPACE;
				out[outpos++] = (uint8_t)sym;
				continue;
			}
			if (sym == 256)
				break;
			length = len_base[sym - 257] +
				 getbits(&b, len_extra[sym - 257]);
			if (length > out_avail - outpos)
				return ORACLE_INSUFFICIENT_SPACE;
			sym = decode_sym(&b, &off);
			offset = off_base[sym] + getbits(&b, off_extra[sym]);
			if (offset > outpos)
				return ORACLE_BAD_DATA;
			while (length--) {
				out[outpos] = out[outpos - offset];
				outpos++;
			}
		}
	} while (!final);

	/* block_done epilogue: decompress_template.h:740-771 */
	{
		uint64_t over = b.loaded / 8 > b.in_n ? b.loaded / 8 - b.in_n : 0;

		if (over > bitsleft(&b) / 8)
			return ORACLE_BAD_DATA;
	}
	if (actual_in)
		*actual_in = (size_t)((b.consumed + 7) / 8);
	if (actual_out)
		*actual_out = outpos;
	else if (outpos != out_avail)
		return ORACLE_SHORT_OUTPUT;
	return ORACLE_SUCCESS;
}

/* lib/zlib_decompress.c:31-95 */
int oracle_zlib_decompress(const void *in_, size_t in_nbytes,
			   void *out, size_t out_avail,
			   size_t *actual_in, size_t *actual_out)
{
	const uint8_t *in = (const uint8_t *)in_;
	unsigned hdr;
	size_t ain, aout;
	int r;

	if (in_nbytes < 6)
		return ORACLE_BAD_DATA;
	hdr = (in[0] << 8) | in[1];
	if (hdr % 31)
		return ORACLE_BAD_DATA;
	if (((hdr >> 8) & 0xF) != 8)
		return ORACLE_BAD_DATA;
	if ((hdr >> 12) > 7)
		return ORACLE_BAD_DATA;
	if ((hdr >> 5) & 1)
		return ORACLE_BAD_DATA;
	r = oracle_deflate_decompress(in + 2, in_nbytes - 6, out, out_avail,
				      &ain, actual_out ? &aout : NULL);
	if (r != ORACLE_SUCCESS)
		return r;
	if (actual_out)
		*actual_out = aout;
	else
		aout = out_avail;
	in += 2 + ain;
	if (oracle_adler32(1, out, aout) !=
	    (((uint32_t)in[0] << 24) | (in[1] << 16) | (in[2] << 8) | in[3]))
		return ORACLE_BAD_DATA;
	if (actual_in)
		*actual_in = 2 + ain + 4;
	return ORACLE_SUCCESS;
}

/* lib/gzip_decompress.c:31-131 */
int oracle_gzip_decompress(const void *in_, size_t in_nbytes,
			   void *out, size_t out_avail,
			   size_t *actual_in, size_t *actual_out)
{
	const uint8_t *in = (const uint8_t *)in_;
	const uint8_t *p = in, *end = in + in_nbytes;
	unsigned flg;
	size_t ain, aout;
	int r;

	if (in_nbytes < 18)
		return ORACLE_BAD_DATA;
	if (*p++ != 0x1F || *p++ != 0x8B || *p++ != 8)
		return ORACLE_BAD_DATA;
	flg = *p++;
	p += 6;		/* MTIME, XFL, OS */
	if (flg & 0xE0)
		return ORACLE_BAD_DATA;
	if (flg & 0x04) {	/* FEXTRA */
		unsigned xlen = p[0] | (p[1] << 8);

		p += 2;
		if ((size_t)(end - p) < (size_t)xlen + 8)
			return ORACLE_BAD_DATA;
		p += xlen;
	}
	if (flg & 0x08) {	/* FNAME */
		while (*p++ != 0 && p != end)
			;
		if (end - p < 8)
			return ORACLE_BAD_DATA;
	}
	if (flg & 0x10) {	/* FCOMMENT */
		while (*p++ != 0 && p != end)
			;
		if (end - p < 8)
			return ORACLE_BAD_DATA;
	}
	if (flg & 0x02) {	/* FHCRC */
		p += 2;
		if (end - p < 8)
			return ORACLE_BAD_DATA;
	}
	r = oracle_deflate_decompress(p, (size_t)(end - 8 - p), out, out_avail,
				      &ain, actual_out ? &aout : NULL);
	if (r != ORACLE_SUCCESS)
		return r;
	if (actual_out)
		*actual_out = aout;
	else
		aout = out_avail;
	p += ain;
	if (oracle_crc32(0, out, aout) !=
	    ((uint32_t)p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)))
		return ORACLE_BAD_DATA;
	p += 4;
	if ((uint32_t)aout !=
	    ((uint32_t)p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)))
		return ORACLE_BAD_DATA;
	p += 4;
	if (actual_in)
		*actual_in = (size_t)(p - in);
	return ORACLE_SUCCESS;
}
