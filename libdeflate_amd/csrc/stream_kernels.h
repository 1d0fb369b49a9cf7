/*
 * stream_kernels.h - what inflate_stream.hip (the kernels that decode ONE
 * large stream on many waves) shares with host_stream.hip.
 */
#ifndef LDA_STREAM_KERNELS_H
#define LDA_STREAM_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* how a chunk starts */
#define LDA_CHUNK_HEADER 0u	/* at the block header at hdr_bit */
#define LDA_CHUNK_WARM 1u	/* inside the block of hdr_bit: parse from start_bit (a guess),
				 * the first token boundary >= target_bit is the start */
#define LDA_CHUNK_EXACT 2u	/* inside the block of hdr_bit, at the token boundary start_bit */

/* hdr_bit of a chunk that starts inside a STATIC block: the static codes need
 * no header, so such a chunk can be planned without knowing where its block
 * began (a stream of static blocks has no header the finder could find).  It
 * does not know either whether its block is the stream's last: it stops at
 * the block's end-of-block symbol, a boundary like any other, and the host -
 * which follows the chain from a chunk that did read the header - knows
 * whether the stream ends there. */
#define LDA_HDR_STATIC (~(uint64_t)1)

/* All positions are bit offsets into the raw DEFLATE stream.  A chunk ends at
 * the first token boundary (the end of a block counts) at or after limit_bit,
 * or with the stream's final block. */
struct lda_stream_chunk {
	uint64_t hdr_bit;
	uint64_t start_bit;
	uint64_t target_bit;
	uint64_t limit_bit;
	uint64_t out_off;	/* decode pass: absolute output position of the chunk's first byte */
	uint32_t kind;
	uint32_t phases;	/* count pass: 0, or K on the first of K EXACT chunks at consecutive
				 * bits with one limit that are counted together, ~0 on the others */
	uint32_t hdr_cache;	/* 0, or 1 + the slot of lda_stream_hdr_cache_kernel that holds the
				 * code lengths of the header at hdr_bit */
	uint32_t hint;		/* decode pass: 0, or 1 + the row of token boundaries the count pass
				 * left for this chunk (phase_count(): starts for the lanes' parses) */
};

#define LDA_STREAM_OK 0u	/* stopped at the limit */
#define LDA_STREAM_FINAL 1u	/* the final block ended */
#define LDA_STREAM_ERR 2u	/* not decodable from here (or a garbage start) */
#define LDA_RES_BOUNDARY 1u	/* end_bit is the first bit of a block header */
#define LDA_RES_BAD_DIST 2u	/* a distance reaches back before the stream */
#define LDA_RES_GOV_FINAL 4u	/* end_bit lies inside the stream's final block (as far as the
				 * chunk knows: it read that block's header) */

struct lda_stream_res {
	uint64_t start_bit;	/* where the chunk really started (WARM: found) */
	uint64_t end_bit;
	uint64_t end_hdr_bit;	/* header of the block end_bit lies in (= end_bit at a boundary;
				 * LDA_HDR_STATIC inside a static block, wherever it began) */
	uint64_t nout;		/* bytes the chunk produces */
	uint32_t status;
	uint32_t flags;
};

extern "C" __global__ void
lda_stream_count_kernel(uint32_t nchunks, const struct lda_stream_chunk *chunks,
			struct lda_stream_res *res, const uint8_t *inp, uint64_t in_n,
			uint32_t *tokscratch, const uint8_t *hdr_lens, const uint32_t *hdr_info,
			uint16_t *hints);
extern "C" __global__ void
lda_stream_decode_kernel(uint32_t nchunks, const struct lda_stream_chunk *chunks,
			 struct lda_stream_res *res, const uint8_t *inp, uint64_t in_n,
			 uint16_t *sym, uint32_t *tokscratch, const uint8_t *hdr_lens,
			 const uint32_t *hdr_info, const uint16_t *hints);
extern "C" __global__ void
lda_stream_hdr_cache_kernel(const uint8_t *inp, uint64_t in_n, const uint64_t *cand,
			    const uint32_t *ncand, uint32_t nslots, uint8_t *hdr_lens,
			    uint32_t *hdr_info);
extern "C" size_t lda_stream_hdr_cache_lds(void);
#define LDA_STREAM_HDR_SLOTS 8192u	/* headers of a window parsed ahead (320 + 16 bytes each) */
extern "C" __global__ void
lda_stream_find_a_kernel(const uint8_t *inp, uint64_t in_n, uint64_t bit0, uint64_t nbits,
			 uint64_t *queue, uint32_t *qcount, uint32_t qcap);
extern "C" __global__ void
lda_stream_find_b_kernel(const uint8_t *inp, uint64_t in_n, const uint64_t *queue,
			 const uint32_t *qcount, uint32_t qcap, uint64_t *cand,
			 uint32_t *ncand, uint32_t ccap);
extern "C" __global__ void
lda_stream_window_kernel(uint32_t nchunks, uint32_t per_group, uint32_t phase,
			 const uint64_t *out_off, const uint16_t *sym, uint8_t *out,
			 uint16_t *gwin, const uint16_t *fwin, uint32_t *err);
extern "C" __global__ void
lda_stream_window_scan_kernel(uint32_t n, uint32_t h, const uint16_t *src, uint16_t *dst);
extern "C" __global__ void
lda_stream_resolve_kernel(uint32_t nchunks, uint32_t chunk0, const uint64_t *out_off,
			  const uint16_t *sym, uint8_t *out, uint32_t *err);
extern "C" size_t lda_stream_chunk_lds(void);
extern "C" size_t lda_stream_find_b_lds(void);
extern "C" size_t lda_stream_tokcap(void);	/* u32 words of token scratch per decode wave */

#endif /* LDA_STREAM_KERNELS_H */
