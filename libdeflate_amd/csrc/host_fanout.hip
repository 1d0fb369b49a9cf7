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
#include <thread>
#include <vector>

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
	std::vector<int> rc(shards, (int)LIBDEFLATE_AMD_OK);
	std::vector<std::thread> th;
	size_t started = 1;

	th.reserve(shards);
	for (size_t k = 1; k < shards; k++) {
		try {
			th.emplace_back([&rc, &fn, k] {
				rc[k] = no_unwind("fan-out shard", (int)LIBDEFLATE_AMD_OOM,
						  [&]() { return fn(k); });
			});
			started++;
		} catch (...) {
			break;	/* no thread to be had: the calling one takes the rest */
		}
	}
	rc[0] = fn(0);
	for (size_t k = started; k < shards; k++)
		rc[k] = fn(k);
	for (std::thread &t : th)
		t.join();
	for (size_t k = 0; k < shards; k++)
		if (rc[k] != LIBDEFLATE_AMD_OK)
			return rc[k];
	return LIBDEFLATE_AMD_OK;
}

} /* namespace lda */

/* shards the calling thread's last host-pointer batch was spread over */
extern "C" LIBDEFLATEAPI size_t libdeflate_amd_last_fanout(void)
{
	return lda::t_last_fanout;
}
