/*
 * host_fanout.hip - one host-pointer batch over several GPUs of a node, for
 * callers without Python (the chunks of a batch are
 * independent, so the batch is partitioned and nothing else).
 *
 * libdeflate_amd_{compress,decompress}_batch_host take host pointers and
 * return host results, so a C / cgo / JNI caller that owns ONE compressor
 * object can still use every GPU of its node: with LDA_DEVICES=all (or =N) in
 * the environment the batch is cut into contiguous shards of about equal byte
 * counts, shard k runs on device k of the visible ones (the object's own
 * device first) through an object of its own on a host thread of its own, and
 * every result lands where the caller asked for it - the outputs are in host
 * order already, no gather and no collective.  The per-device objects are
 * built on first use and freed with the caller's.  On a box with one visible
 * GPU (or with LDA_DEVICES unset) the plan is one shard and the call is the
 * single-device path, untouched.
 *
 * What torch.distributed does for the Python callers (libdeflate_amd/shard.py:
 * one process per GPU, RCCL for the final gather of sizes) this does inside
 * one process with threads; it is the multi-device entry point the C-ABI did
 * not have.
 */
#include <new>
#include <stdio.h>
#include <thread>

#include "host_objects.h"

namespace lda {

static thread_local size_t t_last_fanout = 1;

void fanout_note(size_t shards)
{
	t_last_fanout = shards;
}

size_t fanout_plan(int own_device, size_t n, const size_t *nbytes, size_t *bounds, int *devs)
{
	const EnvCfg &env = env_cfg();
	int want = env.devices, count = 0;

	bounds[0] = 0;
	bounds[1] = n;
	devs[0] = own_device;
	if (want == 1 || n < 2)
		return 1;
	if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
		return 1;
	if (want < 0)
		want = count;
	if (want > count && !env.fanout_oversub)
		want = count;
	if (want > LDA_MAX_SHARDS)
		want = LDA_MAX_SHARDS;
	if (want < 2)
		return 1;
	/* shards of at least 1 MiB of input: a device that gets less costs more
	 * to reach than it gives */
	const size_t k = slice_by_bytes(n, nbytes, (size_t)want, (size_t)1 << 20, bounds);
	for (size_t i = 0; i < k; i++)
		devs[i] = (own_device + (int)i) % count;
	return k;
}

int fanout_run(size_t shards, const std::function<int(size_t)> &fn)
{
	/* Nothing may unwind past a joinable std::thread (that is std::terminate),
	 * and the error text is per thread: every shard - the calling thread's
	 * included - runs inside no_unwind, leaves its status and its message in
	 * its own slot, and the first failing shard's message becomes the
	 * caller's libdeflate_amd_last_error() after ALL threads were joined. */
	struct Slot {
		int rc = (int)LIBDEFLATE_AMD_OK;
		char msg[256] = { 0 };
	};
	Slot *slot = new (std::nothrow) Slot[shards];
	if (!slot) {
		set_error("fan-out: out of host memory");
		return (int)LIBDEFLATE_AMD_OOM;
	}
	auto run = [slot, &fn](size_t k) {
		slot[k].rc = no_unwind("fan-out shard", (int)LIBDEFLATE_AMD_OOM,
				       [&]() { return fn(k); });
		if (slot[k].rc != LIBDEFLATE_AMD_OK) {
			const char *m = libdeflate_amd_last_error();
			snprintf(slot[k].msg, sizeof(slot[k].msg), "%s", m ? m : "");
		}
	};
	std::thread *th = new (std::nothrow) std::thread[shards];
	size_t started = 0;	/* threads th[1..started] run shards 1..started */

	if (th) {
		for (size_t k = 1; k < shards; k++) {
			try {
				th[k] = std::thread(run, k);
				started = k;
			} catch (...) {
				break;	/* no thread to be had: the calling one takes the rest */
			}
		}
	}
	run(0);
	for (size_t k = started + 1; k < shards; k++)
		run(k);
	for (size_t k = 1; k <= started; k++)
		th[k].join();
	delete[] th;
	int rc = (int)LIBDEFLATE_AMD_OK;
	for (size_t k = 0; k < shards; k++)
		if (slot[k].rc != LIBDEFLATE_AMD_OK) {
			rc = slot[k].rc;
			set_error("shard %zu of %zu: %s", k, shards, slot[k].msg);
			break;
		}
	delete[] slot;
	return rc;
}

} /* namespace lda */

/* shards the calling thread's last host-pointer batch was spread over */
extern "C" LIBDEFLATEAPI size_t libdeflate_amd_last_fanout(void)
{
	return lda::t_last_fanout;
}
