/*
 * libdeflate_amd.h - C-ABI of the MI355X-native whole-buffer DEFLATE engine.
 *
 * Two groups of entry points, all `extern "C"`, plain pointers and sizes:
 *
 *  (1) The 21 `libdeflate_*` symbols of the reference's public header
 *      (/root/reference/libdeflate.h, v1.25) with identical signatures,
 *      argument meaning, return conventions and error codes, so that a
 *      program written against libdeflate links against libdeflate_amd.so
 *      unchanged.  Each declaration cites the reference line it replaces.
 *      `in`/`out` are HOST pointers, exactly as in the reference; every call
 *      is executed as a batch of one on the GPU (H2D, kernels, D2H).  There
 *      is NO CPU fallback: without a usable gfx950 device the allocators
 *      return NULL and the checksum calls abort() with a message on stderr.
 *
 *  (2) The additive batch extension `libdeflate_amd_*`: the same operations
 *      over N independent chunks that are ALREADY RESIDENT IN HBM, described
 *      by offset/size arrays (also in HBM).  This is the hot path bench.py
 *      measures; one 64-lane wavefront (decode, checksums) or one workgroup
 *      (LZ77 parse) per chunk.  `stream` is a hipStream_t passed as void*
 *      (NULL = the default stream); the calls enqueue work and return.
 *
 * Result conventions are the reference's: compress -> bytes written, 0 when
 * the output does not fit (libdeflate.h:73-74); decompress ->
 * enum libdeflate_result (libdeflate.h:194-209).
 */
#ifndef LIBDEFLATE_AMD_H
#define LIBDEFLATE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIBDEFLATE_VERSION_MAJOR	1	/* libdeflate.h:15 */
#define LIBDEFLATE_VERSION_MINOR	25	/* libdeflate.h:16 */
#define LIBDEFLATE_VERSION_STRING	"1.25"	/* libdeflate.h:17 */
#define LIBDEFLATE_AMD_VERSION_STRING	"0.1-gfx950"

#ifndef LIBDEFLATEAPI
#  define LIBDEFLATEAPI __attribute__((visibility("default")))
#endif

struct libdeflate_compressor;	/* opaque; host object + device scratch */
struct libdeflate_decompressor;	/* opaque */

/* libdeflate.h:379-406 - per-object allocator override */
struct libdeflate_options {
	size_t sizeof_options;		/* must equal sizeof(struct) */
	void *(*malloc_func)(size_t);	/* NULL -> global / malloc */
	void (*free_func)(void *);
};

/* libdeflate.h:194-209 */
enum libdeflate_result {
	LIBDEFLATE_SUCCESS = 0,
	LIBDEFLATE_BAD_DATA = 1,
	LIBDEFLATE_SHORT_OUTPUT = 2,
	LIBDEFLATE_INSUFFICIENT_SPACE = 3,
};

/* ------------------------------------------------------------------ */
/* (1) drop-in single-buffer API (host pointers)                       */
/* ------------------------------------------------------------------ */

/* libdeflate.h:59-60; level 0..12, -1 = 6; NULL on bad level / OOM / no GPU */
LIBDEFLATEAPI struct libdeflate_compressor *
libdeflate_alloc_compressor(int compression_level);

/* libdeflate.h:65-67 */
LIBDEFLATEAPI struct libdeflate_compressor *
libdeflate_alloc_compressor_ex(int compression_level,
			       const struct libdeflate_options *options);

/* libdeflate.h:85-88 */
LIBDEFLATEAPI size_t
libdeflate_deflate_compress(struct libdeflate_compressor *compressor,
			    const void *in, size_t in_nbytes,
			    void *out, size_t out_nbytes_avail);

/* libdeflate.h:114-116; compressor may be NULL */
LIBDEFLATEAPI size_t
libdeflate_deflate_compress_bound(struct libdeflate_compressor *compressor,
				  size_t in_nbytes);

/* libdeflate.h:122-125 */
LIBDEFLATEAPI size_t
libdeflate_zlib_compress(struct libdeflate_compressor *compressor,
			 const void *in, size_t in_nbytes,
			 void *out, size_t out_nbytes_avail);

/* libdeflate.h:132-134 */
LIBDEFLATEAPI size_t
libdeflate_zlib_compress_bound(struct libdeflate_compressor *compressor,
			       size_t in_nbytes);

/* libdeflate.h:140-143 */
LIBDEFLATEAPI size_t
libdeflate_gzip_compress(struct libdeflate_compressor *compressor,
			 const void *in, size_t in_nbytes,
			 void *out, size_t out_nbytes_avail);

/* libdeflate.h:150-152 */
LIBDEFLATEAPI size_t
libdeflate_gzip_compress_bound(struct libdeflate_compressor *compressor,
			       size_t in_nbytes);

/* libdeflate.h:159-160; NULL is a no-op */
LIBDEFLATEAPI void
libdeflate_free_compressor(struct libdeflate_compressor *compressor);

/* libdeflate.h:181-182 */
LIBDEFLATEAPI struct libdeflate_decompressor *
libdeflate_alloc_decompressor(void);

/* libdeflate.h:187-188 */
LIBDEFLATEAPI struct libdeflate_decompressor *
libdeflate_alloc_decompressor_ex(const struct libdeflate_options *options);

/* libdeflate.h:242-246 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_deflate_decompress(struct libdeflate_decompressor *decompressor,
			      const void *in, size_t in_nbytes,
			      void *out, size_t out_nbytes_avail,
			      size_t *actual_out_nbytes_ret);

/* libdeflate.h:254-259 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_deflate_decompress_ex(struct libdeflate_decompressor *decompressor,
				 const void *in, size_t in_nbytes,
				 void *out, size_t out_nbytes_avail,
				 size_t *actual_in_nbytes_ret,
				 size_t *actual_out_nbytes_ret);

/* libdeflate.h:269-273 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_zlib_decompress(struct libdeflate_decompressor *decompressor,
			   const void *in, size_t in_nbytes,
			   void *out, size_t out_nbytes_avail,
			   size_t *actual_out_nbytes_ret);

/* libdeflate.h:282-287 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_zlib_decompress_ex(struct libdeflate_decompressor *decompressor,
			      const void *in, size_t in_nbytes,
			      void *out, size_t out_nbytes_avail,
			      size_t *actual_in_nbytes_ret,
			      size_t *actual_out_nbytes_ret);

/* libdeflate.h:297-301 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_gzip_decompress(struct libdeflate_decompressor *decompressor,
			   const void *in, size_t in_nbytes,
			   void *out, size_t out_nbytes_avail,
			   size_t *actual_out_nbytes_ret);

/* libdeflate.h:310-315 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_gzip_decompress_ex(struct libdeflate_decompressor *decompressor,
			      const void *in, size_t in_nbytes,
			      void *out, size_t out_nbytes_avail,
			      size_t *actual_in_nbytes_ret,
			      size_t *actual_out_nbytes_ret);

/* libdeflate.h:322-323; NULL is a no-op */
LIBDEFLATEAPI void
libdeflate_free_decompressor(struct libdeflate_decompressor *decompressor);

/* libdeflate.h:335-336; initial value 1; buffer == NULL -> 1 */
LIBDEFLATEAPI uint32_t
libdeflate_adler32(uint32_t adler, const void *buffer, size_t len);

/* libdeflate.h:345-346; initial value 0; buffer == NULL -> 0 */
LIBDEFLATEAPI uint32_t
libdeflate_crc32(uint32_t crc, const void *buffer, size_t len);

/* libdeflate.h:363-365 */
LIBDEFLATEAPI void
libdeflate_set_memory_allocator(void *(*malloc_func)(size_t),
				void (*free_func)(void *));

/* ------------------------------------------------------------------ */
/* (2) batch extension: N independent chunks resident in HBM            */
/* ------------------------------------------------------------------ */

enum libdeflate_amd_format {
	LIBDEFLATE_AMD_DEFLATE = 0,	/* raw DEFLATE     */
	LIBDEFLATE_AMD_ZLIB = 1,	/* + 2 B header, Adler-32 footer (BE) */
	LIBDEFLATE_AMD_GZIP = 2,	/* + 10 B header, CRC-32 + ISIZE (LE) */
};

/* status of the library itself (not of a stream) */
enum libdeflate_amd_status {
	LIBDEFLATE_AMD_OK = 0,
	LIBDEFLATE_AMD_NO_DEVICE = -1,	/* no gfx950 device / HIP runtime error */
	LIBDEFLATE_AMD_BAD_ARG = -2,
	LIBDEFLATE_AMD_OOM = -3,	/* hipMalloc of scratch failed */
};

/* 0 when a usable device is present; never falls back to the CPU */
LIBDEFLATEAPI int
libdeflate_amd_device_ready(void);
/* human-readable reason for the last non-OK status (thread-local) */
LIBDEFLATEAPI const char *
libdeflate_amd_last_error(void);
/* The tuning switches of INTEGRATION.md (LDA_* environment variables) are
 * read ONCE, when the library is loaded - not per call.  A process that
 * changes them afterwards calls this to have them read again (not to be
 * called while batches are in flight). */
LIBDEFLATEAPI void
libdeflate_amd_reload_env(void);

/*
 * Devices.  An object (compressor / decompressor) belongs to the HIP device
 * that was current when it was allocated; every call that takes the object
 * runs there, whatever device the calling thread has current, and puts the
 * caller's device back (libdeflate.h has no notion of a device).  The
 * device-pointer batch calls expect their buffers and their stream on the
 * object's device.
 *
 * The host-pointer batch calls (libdeflate_amd_*_batch_host) can use every GPU
 * of the node from ONE object: with LDA_DEVICES=all (or =N) in the environment
 * a batch is cut into contiguous shards of about equal byte counts, shard k
 * runs on visible device (own + k) through an object and a host thread of its
 * own, and every result lands where the caller asked for it - host order, no
 * gather.  Unset (the default) or one visible GPU: one shard, the
 * single-device path.  Returns the shards the calling thread's last
 * host-pointer batch was spread over.
 */
LIBDEFLATEAPI size_t
libdeflate_amd_last_fanout(void);

/*
 * Two hardware behaviours are checked on every device before its first use
 * (~1 ms, once per device and process; LDA_NO_SELFCHECK skips it): the lane
 * order of conflicting LDS atomics (the compress kernel relies on it beyond
 * what the ISA manual states) and - a belt: the kernels wait for their stores
 * first, which is architected - that a wave's global store is visible to
 * another lane's plain load behind s_waitcnt vmcnt(0).  A device that
 * deviates is refused - the allocators return NULL and
 * libdeflate_amd_last_error() says why.  This runs the check again on the
 * current device: out[0..4] = LDS-atomic lanes checked, of them out of order,
 * same-instruction conflicts among them, loads checked, of them stale.
 * Returns LIBDEFLATE_AMD_OK when the device passes.
 */
LIBDEFLATEAPI int
libdeflate_amd_selfcheck(uint64_t *out /* [5], may be NULL */);

/*
 * Chunk i of a batch occupies bytes [offsets[i], offsets[i] + nbytes[i]) of a
 * base buffer.  `d_` pointers are device pointers.  Offsets/sizes are u64 so
 * the same descriptors serve 4 KiB filesystem blocks and multi-GiB buffers.
 *
 * Compress: for each chunk writes a complete stream of `format` into its
 * output slot and the stream size into d_out_nbytes[i] (0 = did not fit in
 * d_out_avail[i], same rule as libdeflate.h:73-74).
 * Batch counterpart of libdeflate_{deflate,zlib,gzip}_compress
 * (libdeflate.h:85-88,122-125,140-143).
 */
LIBDEFLATEAPI int
libdeflate_amd_compress_batch(struct libdeflate_compressor *compressor,
			      int format, size_t n_chunks,
			      const void *d_in, const uint64_t *d_in_offsets,
			      const uint64_t *d_in_nbytes,
			      void *d_out, const uint64_t *d_out_offsets,
			      const uint64_t *d_out_avail,
			      uint64_t *d_out_nbytes, void *stream);

/*
 * The same with an upper bound of the chunk sizes stated by the caller (the
 * sizes themselves live in HBM, where the host cannot see them): batches of
 * small chunks - at most 4096 bytes, the filesystem-block shape - run on a
 * kernel that keeps three chunks per CU in flight instead of one.  The bound
 * is a promise, and it is only CHECKED where it selects that kernel (bound <=
 * 4096, levels 0-9): there a chunk larger than the bound reports 0, like one
 * that does not fit its slot.  With any other bound, or at levels 10-12, the
 * ordinary kernel runs and takes chunks of any size.
 */
LIBDEFLATEAPI int
libdeflate_amd_compress_batch_bounded(struct libdeflate_compressor *compressor,
				      int format, size_t n_chunks,
				      const void *d_in, const uint64_t *d_in_offsets,
				      const uint64_t *d_in_nbytes,
				      void *d_out, const uint64_t *d_out_offsets,
				      const uint64_t *d_out_avail,
				      uint64_t *d_out_nbytes, size_t max_in_nbytes,
				      void *stream);

/*
 * Decompress: d_results[i] receives the enum libdeflate_result of chunk i.
 * d_actual_in / d_actual_out may be NULL; a NULL d_actual_out has the
 * reference's meaning (the stream must fill d_out_avail[i] exactly, else
 * SHORT_OUTPUT, decompress_template.h:765-770).  A failed chunk never
 * affects its neighbours.  Batch counterpart of
 * libdeflate_{deflate,zlib,gzip}_decompress_ex (libdeflate.h:254-259,
 * 282-287,310-315).
 */
LIBDEFLATEAPI int
libdeflate_amd_decompress_batch(struct libdeflate_decompressor *decompressor,
				int format, size_t n_chunks,
				const void *d_in, const uint64_t *d_in_offsets,
				const uint64_t *d_in_nbytes,
				void *d_out, const uint64_t *d_out_offsets,
				const uint64_t *d_out_avail,
				int32_t *d_results,
				uint64_t *d_actual_in, uint64_t *d_actual_out,
				void *stream);

/*
 * Checksums of N chunks.  d_init may be NULL (CRC: 0, Adler: 1), otherwise
 * d_init[i] is the running value to continue from (libdeflate.h:326-346).
 */
LIBDEFLATEAPI int
libdeflate_amd_crc32_batch(size_t n_chunks, const void *d_in,
			   const uint64_t *d_offsets, const uint64_t *d_nbytes,
			   const uint32_t *d_init, uint32_t *d_out,
			   void *stream);
LIBDEFLATEAPI int
libdeflate_amd_adler32_batch(size_t n_chunks, const void *d_in,
			     const uint64_t *d_offsets,
			     const uint64_t *d_nbytes,
			     const uint32_t *d_init, uint32_t *d_out,
			     void *stream);

/*
 * Compaction of a batch's ragged outputs (what a caller of the reference does
 * with the sizes libdeflate_*_compress returns: write them back to back,
 * programs/gzip.c:149-185).  Copies d_nbytes[i] bytes from d_in +
 * d_in_offsets[i] to d_out + d_out_offsets[i], where d_out_offsets[0..n) is
 * the exclusive prefix sum of d_nbytes (computed here) and d_out_offsets[n]
 * the total.  d_out_offsets must have room for
 * libdeflate_amd_compact_offsets_len(n) entries (the tail is scan scratch).
 * d_out must not overlap the inputs.
 */
LIBDEFLATEAPI size_t
libdeflate_amd_compact_offsets_len(size_t n_chunks);
LIBDEFLATEAPI int
libdeflate_amd_compact_batch(size_t n_chunks, const void *d_in,
			     const uint64_t *d_in_offsets,
			     const uint64_t *d_nbytes, void *d_out,
			     uint64_t *d_out_offsets, void *stream);

/*
 * Objects and streams: the batch calls only ENQUEUE work on `stream`.  A
 * compressor / decompressor owns device scratch (match lists, token scratch,
 * per-chunk sums, work counters) that a batch uses until it completes, so an
 * object may have batches in flight on ONE stream at a time (they are ordered
 * by the stream); use one object per stream for concurrent batches, exactly
 * as the reference asks for one object per thread (libdeflate.h:56-57,
 * :178-179).  Chunks of 4 GiB and more are not supported by the batch
 * kernels (positions are 32-bit): such a chunk reports 0 / BAD_DATA.  The
 * single-buffer libdeflate_*_compress calls take inputs of any size (they cut
 * them into 64 KiB segments that run as one batch and stitch the streams);
 * the single-buffer decompress calls are limited to streams of less than
 * 4 GiB each.
 */

/*
 * Convenience forms taking HOST arrays of per-chunk host pointers (what a
 * cgo/JNI/ctypes caller holding ordinary buffers has).  They pack the batch
 * into pinned host memory, move it as a few large DMA transfers (not one copy
 * per chunk), run the device batch above, compact the outputs on the device
 * and bring them back the same way; blocking.
 * results/actual_* have the meaning above.
 */
LIBDEFLATEAPI int
libdeflate_amd_compress_batch_host(struct libdeflate_compressor *compressor,
				   int format, size_t n_chunks,
				   const void *const *in,
				   const size_t *in_nbytes,
				   void *const *out, const size_t *out_avail,
				   size_t *out_nbytes);
LIBDEFLATEAPI int
libdeflate_amd_decompress_batch_host(struct libdeflate_decompressor *d,
				     int format, size_t n_chunks,
				     const void *const *in,
				     const size_t *in_nbytes,
				     void *const *out, const size_t *out_avail,
				     int32_t *results, size_t *actual_in,
				     size_t *actual_out /* NULL allowed */);

/*
 * What the last single-buffer libdeflate_*_decompress[_ex] call of this thread
 * did with its stream (diagnostics; tests use it to see that a large stream
 * really went over many waves): [0] 1 = decoded on the many-wave path, 0 = on
 * one wave; [1] why not (0 none, 1 switched off or too small, 2 container
 * header, 3 chain, 4 a chunk failed, 5 no final block, 6 output does not fit,
 * 7 output does not fill, 8 decode pass disagrees, 9 device, 10 too many
 * repairs); [2] bit offsets that passed the first filter of the block finder;
 * [3] block starts found; [4] chunks planned; [5] repairs; [6] chunks decoded;
 * [7] bytes produced; [8..13] host-side microseconds of: copy in, block finder,
 * count pass + chain, queueing the decode / window / resolve / checksum kernels,
 * footer check, output copy (which waits for those kernels); [14] input windows;
 * [15] chunks the host made itself from runs of stored blocks (no count pass).
 */
#define LIBDEFLATE_AMD_STREAM_STATS 16
LIBDEFLATEAPI void
libdeflate_amd_stream_stats(uint64_t *out /* [16] */);

/*
 * A gzip buffer of SEVERAL members (concatenated .gz files, pigz -i, BGZF).
 * libdeflate_gzip_decompress decodes the first member only
 * (lib/gzip_decompress.c:103-131, libdeflate.h:289-296); its caller loops, as
 * programs/gzip.c:236-299 does.  This is that loop in one call: members that
 * state their size (BGZF "BC" extra subfield) are indexed from their headers
 * and decoded as ONE device batch, anything else member after member.  `out`
 * receives the members' outputs back to back; fails with the first member's
 * non-success result.  The three result pointers may be NULL.
 */
LIBDEFLATEAPI enum libdeflate_result
libdeflate_amd_gzip_decompress_members(struct libdeflate_decompressor *d,
				       const void *in, size_t in_nbytes,
				       void *out, size_t out_nbytes_avail,
				       size_t *actual_in_nbytes_ret,
				       size_t *actual_out_nbytes_ret,
				       size_t *members_ret);

#ifdef __cplusplus
}
#endif
#endif /* LIBDEFLATE_AMD_H */
