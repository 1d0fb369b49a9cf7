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
#include <functional>

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

#define LDA_MAX_SHARDS 16	/* devices one host-pointer batch is spread over */

struct libdeflate_decompressor {
	lda::free_func_t free_func;
	lda::malloc_func_t malloc_func;
	int device;		/* the device the object lives on (current at allocation) */
	/* host-pointer batches over several GPUs (LDA_DEVICES): one more object
	 * per further device, built on first use, freed with this one */
	struct libdeflate_decompressor *shard[LDA_MAX_SHARDS];
	lda::DevBuf scratch;	/* per-chunk u32 sums + u64 actual_in/out */
	lda::DevBuf stage;	/* host-pointer entry points */
	lda::DevBuf tokens;	/* per-wave token scratch of the wave-per-stream kernel */
	/* one large stream on many waves (host_stream.hip): input + finder
	 * queues, chunk descriptors / results, 16-bit symbols, output bytes */
	lda::DevBuf sin, squeue, schunks, srepair, ssym, sout, swin, shdr, shint;
	lda::PinnedPair pinned;	/* host-pointer entry points */
	lda::PinnedBuf meta;	/* host-pointer entry points: per-chunk read-backs */
	lda::StreamPair streams;	/* host-pointer entry points: transfers / kernels */
};

struct libdeflate_compressor {
	lda::free_func_t free_func;
	lda::malloc_func_t malloc_func;
	int device;		/* the device the object lives on (current at allocation) */
	struct libdeflate_compressor *shard[LDA_MAX_SHARDS];	/* see libdeflate_decompressor */
	int level;
	lda::DevBuf scratch;	/* parse/encode workspace + per-chunk sums */
	lda::DevBuf stage;
	lda::PinnedPair pinned;	/* host-pointer entry points */
	lda::PinnedBuf meta;	/* host-pointer entry points: per-chunk read-backs */
	lda::StreamPair streams;	/* host-pointer entry points: transfers / kernels */
};

namespace lda {
/* host_fanout.hip: the devices a host-pointer batch of n chunks is spread over
 * (1 = the object's own device only), and the plan: shard k takes the chunks
 * [bounds[k], bounds[k+1]) on device devs[k] */
size_t fanout_plan(int own_device, size_t n, const size_t *nbytes, size_t *bounds, int *devs);
void fanout_note(size_t shards);	/* what libdeflate_amd_last_fanout() reports */
/* run fn(k) for k = 1 .. shards-1 on threads of their own and fn(0) on the
 * calling one; returns the first non-OK status */
int fanout_run(size_t shards, const std::function<int(size_t)> &fn);

/* host_stream.hip: true = answered (result, sizes, output); false = the
 * caller takes the sequential path */
bool decompress_stream_parallel(struct libdeflate_decompressor *d, int format,
				const uint8_t *in, size_t in_nbytes, uint8_t *out,
				size_t out_avail, bool exact_fill, int32_t *res,
				size_t *ain, size_t *aout);
}

#endif /* LDA_HOST_OBJECTS_H */
