/*
 * host_common.h - host-side plumbing shared by the C-ABI translation units:
 * per-device context (constant tables, staging buffer), error reporting.
 */
#ifndef LDA_HOST_COMMON_H
#define LDA_HOST_COMMON_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
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
};

/* context of the calling thread's current device; nullptr (+error) if none */
DeviceCtx *device_ctx();

/* grow-only device staging; caller holds ctx->stage_mu */
void *stage_reserve(DeviceCtx *ctx, size_t nbytes);

/* abort with a message: used where the reference API has no error channel */
[[noreturn]] void die_no_device(const char *what);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} /* namespace lda */

#endif /* LDA_HOST_COMMON_H */
