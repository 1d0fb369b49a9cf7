/*
 * host_context.hip - per-device context, constant-table generation, errors.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "host_common.h"
#include "kernels.h"

namespace lda {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

[[noreturn]] void die_no_device(const char *what)
{
	fprintf(stderr,
		"libdeflate_amd: %s needs a gfx950 (MI355X) device and none is "
		"usable (%s). This library has no CPU fallback.\n", what, g_err);
	abort();
}

#define MAX_DEVICES 16
static DeviceCtx g_ctx[MAX_DEVICES];
static std::mutex g_ctx_mu;

/* CRC-32 tables: see checksum_kernels.hip header for their meaning */
static void gen_crc_tables(uint32_t *tab /*17*256*/, uint32_t *xpow /*1024*/)
{
	const uint32_t poly = 0xEDB88320u;
	uint32_t t0[256];

	for (uint32_t b = 0; b < 256; b++) {
		uint32_t r = b;
		for (int k = 0; k < 8; k++)
			r = (r >> 1) ^ (poly & (0u - (r & 1u)));
		t0[b] = r;
	}
	/* S_k[b] = register after byte b and then (15-k)+1008 zero bytes */
	for (uint32_t b = 0; b < 256; b++) {
		uint32_t r = t0[b];
		int zeros = 0;
		for (int k = 15; k >= 0; k--) {
			int want = (15 - k) + 1008;
			for (; zeros < want; zeros++)
				r = t0[r & 0xFF] ^ (r >> 8);
			tab[k * 256 + b] = r;
		}
	}
	memcpy(tab + 16 * 256, t0, sizeof(t0));
	/* xpow[d] = x^(8d) mod P, x^0 at bit 31 */
	uint32_t x = 0x80000000u;
	for (int d = 0; d < 1024; d++) {
		xpow[d] = x;
		for (int k = 0; k < 8; k++)
			x = (x >> 1) ^ (poly & (0u - (x & 1u)));
	}
}

DeviceCtx *device_ctx()
{
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);

	if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
		set_error("hipGetDevice: %s", hipGetErrorString(e));
		return nullptr;
	}
	std::lock_guard<std::mutex> lk(g_ctx_mu);
	DeviceCtx *c = &g_ctx[dev];
	if (c->device == dev)
		return c;

	hipDeviceProp_t prop;
	LDA_HIP_TRY(hipGetDeviceProperties(&prop, dev), nullptr);
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		set_error("device %d is %s, this build targets gfx950 only",
			  dev, prop.gcnArchName);
		return nullptr;
	}
	static uint32_t tab[LDA_CRC_TABLE_WORDS], xpow[LDA_CRC_XPOW_WORDS];
	gen_crc_tables(tab, xpow);
	LDA_HIP_TRY(hipMalloc((void **)&c->d_crc_tables, sizeof(tab)), nullptr);
	LDA_HIP_TRY(hipMalloc((void **)&c->d_crc_xpow8, sizeof(xpow)), nullptr);
	LDA_HIP_TRY(hipMemcpy(c->d_crc_tables, tab, sizeof(tab),
			      hipMemcpyHostToDevice), nullptr);
	LDA_HIP_TRY(hipMemcpy(c->d_crc_xpow8, xpow, sizeof(xpow),
			      hipMemcpyHostToDevice), nullptr);
	c->num_cus = prop.multiProcessorCount;
	c->device = dev;
	return c;
}

void *stage_reserve(DeviceCtx *ctx, size_t nbytes)
{
	if (nbytes <= ctx->stage_cap)
		return ctx->d_stage;
	size_t cap = align_up(nbytes + nbytes / 4 + 4096, 4096);
	void *p = nullptr;
	if (ctx->d_stage)
		(void)hipFree(ctx->d_stage);
	ctx->d_stage = nullptr;
	ctx->stage_cap = 0;
	hipError_t e = hipMalloc(&p, cap);
	if (e != hipSuccess) {
		set_error("hipMalloc(%zu): %s", cap, hipGetErrorString(e));
		return nullptr;
	}
	ctx->d_stage = p;
	ctx->stage_cap = cap;
	return p;
}

} /* namespace lda */

extern "C" LIBDEFLATEAPI int libdeflate_amd_device_ready(void)
{
	return lda::device_ctx() ? LIBDEFLATE_AMD_OK : LIBDEFLATE_AMD_NO_DEVICE;
}

extern "C" LIBDEFLATEAPI const char *libdeflate_amd_last_error(void)
{
	return lda::g_err;
}
