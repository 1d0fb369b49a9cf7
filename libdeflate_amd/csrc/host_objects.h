/*
 * host_objects.h - the two opaque objects of the C-ABI.
 *
 * Each is ONE allocation through the selected allocator (per-object options >
 * libdeflate_set_memory_allocator > malloc), like the reference
 * (lib/deflate_compress.c:3910-3917, programs/test_custom_malloc.c:53-77).
 * Device memory hangs off the object and never goes through that allocator.
 */
#ifndef LDA_HOST_OBJECTS_H
#define LDA_HOST_OBJECTS_H

#include "host_common.h"

namespace lda {

typedef void *(*malloc_func_t)(size_t);
typedef void (*free_func_t)(void *);

extern malloc_func_t g_malloc;	/* libdeflate_set_memory_allocator */
extern free_func_t g_free;

/* grow-only device buffer owned by an object */
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	void *reserve(size_t n);	/* nullptr + error on failure */
	void release();
};

} /* namespace lda */

struct libdeflate_decompressor {
	lda::free_func_t free_func;
	lda::DevBuf scratch;	/* per-chunk u32 sums + u64 actual_in/out */
	lda::DevBuf stage;	/* host-pointer entry points */
	lda::DevBuf tokens;	/* per-wave token scratch of the wave-per-stream kernel */
	/* one large stream on many waves (host_stream.hip): input + finder
	 * queues, chunk descriptors / results, 16-bit symbols, output bytes */
	lda::DevBuf sin, squeue, schunks, srepair, ssym, sout, swin;
	lda::PinnedPair pinned;	/* host-pointer entry points */
	lda::PinnedBuf meta;	/* host-pointer entry points: per-chunk read-backs */
	lda::StreamPair streams;	/* host-pointer entry points: transfers / kernels */
};

struct libdeflate_compressor {
	lda::free_func_t free_func;
	int level;
	lda::DevBuf scratch;	/* parse/encode workspace + per-chunk sums */
	lda::DevBuf stage;
	lda::PinnedPair pinned;	/* host-pointer entry points */
	lda::PinnedBuf meta;	/* host-pointer entry points: per-chunk read-backs */
	lda::StreamPair streams;	/* host-pointer entry points: transfers / kernels */
};

namespace lda {
/* host_stream.hip: true = answered (result, sizes, output); false = the
 * caller takes the sequential path */
bool decompress_stream_parallel(struct libdeflate_decompressor *d, int format,
				const uint8_t *in, size_t in_nbytes, uint8_t *out,
				size_t out_avail, bool exact_fill, int32_t *res,
				size_t *ain, size_t *aout);
}

#endif /* LDA_HOST_OBJECTS_H */
