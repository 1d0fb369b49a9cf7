/*
 * oracle.h - CPU restatement of the reference's whole-buffer DEFLATE path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so (or oracle/_ref/libdeflate_ref.so).  The product library
 * (libdeflate_amd/csrc -> libdeflate_amd.so) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here
 * against (a) the hand-assembled streams of the reference's own unit tests
 * (programs/test_incomplete_codes.c, test_invalid_streams.c, test_overread.c,
 * test_trailing_bytes.c, test_checksums.c semantics), (b) the committed golden
 * fixtures in tests/golden/ produced by the real reference (oracle/_ref) with
 * oracle/make_golden.py, and (c) live against oracle/_ref when it is built.
 *
 * Each function cites the reference file:line whose behaviour it restates.
 * The code is a from-scratch restatement (canonical-code decoder in the
 * count/first-code style, bitwise CRC, plain Adler), not a copy: it keeps the
 * reference's observable semantics (bytes, result codes, actual_in/out) and
 * none of its table-driven fast paths.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as enum libdeflate_result (libdeflate.h:194-209) */
enum {
	ORACLE_SUCCESS = 0,
	ORACLE_BAD_DATA = 1,
	ORACLE_SHORT_OUTPUT = 2,
	ORACLE_INSUFFICIENT_SPACE = 3,
};

/* formats */
enum { ORACLE_FMT_DEFLATE = 0, ORACLE_FMT_ZLIB = 1, ORACLE_FMT_GZIP = 2 };

uint32_t oracle_crc32(uint32_t crc, const void *buf, size_t len);
uint32_t oracle_adler32(uint32_t adler, const void *buf, size_t len);

/*
 * Raw DEFLATE / zlib / gzip decode with the reference's result-code and
 * actual_in/actual_out semantics.  actual_in / actual_out may be NULL.
 */
int oracle_deflate_decompress(const void *in, size_t in_nbytes,
			      void *out, size_t out_avail,
			      size_t *actual_in, size_t *actual_out);
int oracle_zlib_decompress(const void *in, size_t in_nbytes,
			   void *out, size_t out_avail,
			   size_t *actual_in, size_t *actual_out);
int oracle_gzip_decompress(const void *in, size_t in_nbytes,
			   void *out, size_t out_avail,
			   size_t *actual_in, size_t *actual_out);

/*
 * The block structure of a raw DEFLATE stream, for testing the product's
 * block finder: decodes like oracle_deflate_decompress() and records every
 * block header reached (bit position of BFINAL, BTYPE | 4 * BFINAL, output
 * bytes before it).  Returns the number of blocks (may exceed cap; only cap
 * are stored); *result = the decoder's result code.
 */
struct oracle_block {
	uint64_t bit;
	uint64_t out_pos;
	uint32_t type;
	uint32_t pad;
};
size_t oracle_deflate_block_map(const void *in, size_t in_nbytes, void *out,
				size_t out_avail, struct oracle_block *blocks,
				size_t cap, int *result);

/* compress bounds (pure functions of n) */
size_t oracle_deflate_compress_bound(size_t n);
size_t oracle_zlib_compress_bound(size_t n);
size_t oracle_gzip_compress_bound(size_t n);

/*
 * Restatement of the reference compressor policy (levels 0..9): greedy / lazy /
 * lazy2 hash-chain parse, length-limited Huffman codes, exact block-cost
 * choice.  Output bytes are NOT expected to equal the reference's
 * (libdeflate.h:76-83 leaves them unpinned); tests check that it round-trips
 * and that its ratio tracks oracle/_ref.  Returns bytes written, 0 if it
 * does not fit.
 */
size_t oracle_deflate_compress(int level, const void *in, size_t in_nbytes,
			       void *out, size_t out_avail);
size_t oracle_zlib_compress(int level, const void *in, size_t in_nbytes,
			    void *out, size_t out_avail);
size_t oracle_gzip_compress(int level, const void *in, size_t in_nbytes,
			    void *out, size_t out_avail);

#ifdef __cplusplus
}
#endif
#endif /* ORACLE_H */
