/*
 * host_common.h - host-side plumbing shared by the C-ABI translation units:
 * per-device context (constant tables, staging buffer), error reporting.
 */
#ifndef LDA_HOST_COMMON_H
#define LDA_HOST_COMMON_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <exception>
#include <mutex>

#include "../../include/libdeflate_amd.h"

namespace lda {

void set_error(const char *fmt, ...);

#define LDA_HIP_TRY(expr, failret)                                           \
	do {                                                                 \
		hipError_t e_ = (expr);                                      \
		if (e_ != hipSuccess) {                                      \
			lda::set_error("%s: %s", #expr, hipGetErrorString(e_)); \
			return failret;                                      \
		}                                                            \
	} while (0)

struct DeviceCtx {
	int device = -1;
	int num_cus = 0;
	uint32_t *d_crc_tables = nullptr;	/* 17*256 words */
	uint32_t *d_crc_xpow8 = nullptr;	/* 1024 words */
	/* staging area used by the host-pointer entry points */
	std::mutex stage_mu;
	void *d_stage = nullptr;
	size_t stage_cap = 0;
	/* the kernels' dynamic-LDS limits are raised once per device (setting
	 * them again is harmless, so a race between two first calls only repeats
	 * it); part of the context so that no table is indexed by a device id */
	std::atomic<bool> deflate_attr_set{false}, inflate_attr_set{false}, stream_attr_set{false};
};

/* context of the calling thread's current device; nullptr (+error) if none */
DeviceCtx *device_ctx();
/* the hardware self-check on the current device (selfcheck_kernels.hip):
 * out[0..4] = lanes, order mismatches, conflicts seen, loads, stale loads;
 * true = the device behaves as the kernels need */
bool device_selfcheck(int num_cus, uint64_t out[5]);

/*
 * An object belongs to the device that was current when it was allocated (its
 * scratch, staging and streams live there).  Every entry point that takes an
 * object makes that device current for its duration and puts the caller's
 * back: a caller that drives several GPUs from one thread - or calls with
 * another device current - launches where the object's memory is, not where
 * the thread happens to point (libdeflate.h has no notion of a device: the
 * drop-in must not grow one).  ok() == false: the device could not be made
 * current (the reason is in libdeflate_amd_last_error()).
 */
struct DeviceGuard {
	int prev = -1;
	bool switched = false, good = true;
	explicit DeviceGuard(int device);
	~DeviceGuard();
	bool ok() const { return good; }
	DeviceGuard(const DeviceGuard &) = delete;
	DeviceGuard &operator=(const DeviceGuard &) = delete;
};

/* the environment switches (tuning aids), read ONCE per
 * process at the first use - not per call */
struct EnvCfg {
	bool no_small = false;		/* LDA_NO_SMALL */
	bool no_segments = false;	/* LDA_NO_SEGMENTS */
	int inflate_lpw = 0;		/* LDA_INFLATE_LPW (0 = automatic) */
	bool inflate_par = true;	/* LDA_INFLATE_PAR */
	int inflate_waves_per_cu = 16;	/* LDA_INFLATE_WAVES_PER_CU */
	int host_threads = 4;		/* LDA_HOST_THREADS: packing threads of the host-pointer batches */
	size_t seg_bytes = 0;		/* LDA_SEG_BYTES: sub-range of the segmented single-buffer compress (0 = by size) */
	bool no_stream_par = false;	/* LDA_NO_STREAM_PAR: single streams stay on one wave */
	size_t stream_par_min = 16384;	/* LDA_STREAM_PAR_MIN: smallest stream (bytes in) for the many-wave path */
	size_t stream_window = 0;	/* LDA_STREAM_WINDOW: first input window of that path (0 = 4 MiB) */
	size_t stream_chunk = 0;	/* LDA_STREAM_CHUNK: input bytes per chunk of that path (0 = by size) */
	int devices = 1;		/* LDA_DEVICES: GPUs a host-pointer batch is spread over (N, or "all" = -1) */
	bool no_selfcheck = false;	/* LDA_NO_SELFCHECK: skip the per-device hardware self-check */
	bool fanout_oversub = false;	/* LDA_FANOUT_OVERSUB: more shards than visible devices (a test aid: several shards share a GPU) */
};
const EnvCfg &env_cfg();

/* grow-only device staging; caller holds ctx->stage_mu */
void *stage_reserve(DeviceCtx *ctx, size_t nbytes);

/* abort with a message: ONLY where the reference API has no error channel
 * (libdeflate_crc32 / libdeflate_adler32); prints the real HIP error */
[[noreturn]] void die(const char *what);

/* a failure inside a single-buffer libdeflate_* call that does have an error
 * return: one line on stderr (the error is also in libdeflate_amd_last_error) */
void complain(const char *what, int status);

/*
 * Two pinned host buffers the host-pointer batch entry points pack into /
 * unpack from while the other one is in flight: a batch crosses PCIe as a few
 * large DMA transfers instead of one blocking copy per chunk.
 */
struct PinnedPair {
	uint8_t *buf[2] = { nullptr, nullptr };
	hipEvent_t ev[2] = { nullptr, nullptr };
	size_t cap = 0;
	/* grow-only, sized to what the batch needs (a call on a few KiB pins a
	 * few KiB: 64 KiB at least, LDA_PINNED_SLICE at most per buffer); false
	 * + error if pinned memory is unavailable */
	bool ensure(size_t want);
	void release();
};
/* a small grow-only pinned buffer: what a batch reads back per chunk (sizes,
 * offsets, result codes).  An asynchronous copy to pageable memory is staged
 * by the runtime and blocks the host until the stream gets there; to pinned
 * memory it is a real asynchronous copy and the host looks after the event */
struct PinnedBuf {
	void *p = nullptr;
	size_t cap = 0;
	void *ensure(size_t want);	/* nullptr + error on failure */
	void release();
};
#define LDA_PINNED_SLICE ((size_t)32 << 20)
#define LDA_PINNED_MIN ((size_t)64 << 10)

/*
 * Two streams of an object for its host-pointer batch entry points: the batch
 * goes through in slices, the transfers of a slice on `copy`, its kernels on
 * `comp`, so that the kernels of slice k run while the host packs and sends
 * slice k + 1 and unpacks slice k - 1.
 */
struct StreamPair {
	hipStream_t copy = nullptr, comp = nullptr;
	hipEvent_t mark = nullptr;	/* a point on `comp` that `copy` may wait for */
	hipEvent_t mark2 = nullptr;	/* and one on `copy` that `comp` may wait for */
	bool ensure();
	void release();
};

/* cut n chunks into at most `max_slices` consecutive slices of about equal
 * byte counts (each at least `min_bytes`, as far as the total allows);
 * bounds[0..k] with bounds[0] = 0, bounds[k] = n; returns k >= 1 */
size_t slice_by_bytes(size_t n, const size_t *nbytes, size_t max_slices,
		      size_t min_bytes, size_t *bounds);

/* host chunks -> device, packed at d_base + off[i] (off ascending); blocking
 * only on its own pinned buffers */
int copy_in_packed(PinnedPair *pp, uint8_t *d_base, size_t n,
		   const void *const *in, const size_t *in_nbytes,
		   const uint64_t *off, hipStream_t st);
/* device bytes at d_base + off[i] (off ascending) -> out[i], nbytes[i] each
 * (0 = skip); the stream must already hold the work that produces them */
int copy_out_packed(PinnedPair *pp, const uint8_t *d_base, size_t n,
		    void *const *out, const uint64_t *nbytes,
		    const uint64_t *off, hipStream_t st);

/* one contiguous host range <-> device range, in pieces the copy threads share
 * (`piece` bytes each; 0: sixteen to a slice, see host_context.hip) */
int span_in(PinnedPair *pp, uint8_t *d_base, uint64_t d_off, const uint8_t *src, size_t n,
	    hipStream_t st, size_t piece = 0);
int span_out(PinnedPair *pp, const uint8_t *d_base, uint64_t d_off, uint8_t *dst, size_t n,
	     hipStream_t st, size_t piece = 0);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/*
 * Nothing may unwind through an extern "C" entry point: the reference is C and
 * its callers have no handlers, an exception that reaches them ends the
 * process.  The entry points whose bodies use the standard library's
 * containers or threads run them through this: a failed host allocation
 * becomes the call's failure value, with the reason in
 * libdeflate_amd_last_error().
 */
template <typename R, typename F> static inline R no_unwind(const char *what, R failed, F body)
{
	try {
		return body();
	} catch (const std::exception &e) {
		set_error("%s: %s", what, e.what());
	} catch (...) {
		set_error("%s: exception", what);
	}
	return failed;
}

/* checksum of A || B from the checksums of A and B and the length of B
 * (host_compress.hip) */
uint32_t crc32_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);
/* the same with x^(8 len_b) mod P computed once for many pieces of one length */
uint32_t crc32_shift(uint64_t len);
uint32_t crc32_concat_shift(uint32_t crc_a, uint32_t crc_b, uint32_t shift_b);
uint32_t adler32_concat(uint32_t ad_a, uint32_t ad_b, uint64_t len_b);

} /* namespace lda */

#endif /* LDA_HOST_COMMON_H */
