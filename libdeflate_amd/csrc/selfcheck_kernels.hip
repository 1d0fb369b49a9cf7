/*
 * selfcheck_kernels.hip - the two hardware behaviours the product kernels
 * rely on beyond what the ISA manual promises, checked on every device the
 * library is about to use (device_ctx(), once per device, ~1 ms):
 *
 *  1. lane order of conflicting LDS atomics.  insert_tile() (deflate_kernel.hip)
 *     inserts 64 consecutive positions into the hash chains with ONE
 *     ds_mskor_rtn_b32 and needs every lane to get what a serial insertion
 *     loop would have returned: lanes of one instruction that hit the same
 *     address must be served in ascending lane order.  A device that serves
 *     them otherwise would still produce valid streams (candidates are
 *     byte-verified) but worse ones.
 *  2. a wave's global store is visible to a following load of another lane
 *     of the same wave.  par_round() (inflate_kernel.hip) reads match sources
 *     older than its LDS mirror from the output in HBM, where another lane of
 *     the same wave may have stored them earlier in the round.  Since round 6
 *     the kernel waits for its outstanding stores first (s_waitcnt vmcnt(0):
 *     global_stores_visible(), the compiler's workgroup-scope release /
 *     acquire sequence), so this is architected behaviour; the probe is kept
 *     as a belt and uses the product's own kind of access - plain
 *     (non-volatile) global stores and loads, no sc0 / sc1 bits in the ISA
 *     (tests/test_abi.py checks the encoding), separated by exactly that
 *     sequence.  A device on which such a load could return the old byte
 *     would corrupt raw DEFLATE output silently and is refused.  (The form
 *     without the wait, which the kernel used until round 5, is still
 *     measured at length by tools/hwtest_global_visibility.hip.)
 *
 * Both are measured at length by tools/hwtest_lds_order.hip and
 * tools/hwtest_global_visibility.hip (tests/test_hw_gpu.py); these are their
 * short forms, verified on the device itself so that only two counters come
 * back.  A deviation makes the allocators refuse the device.
 */
#include "device_common.h"
#include "kernels.h"

#define SC_TRIALS 16
#define SC_TAB_BITS 8

/* the hash of (wave, trial, lane): trial classes of 1 / 2 / 8 / 64 / 256
 * distinct buckets, like the long test */
static __device__ __forceinline__ u32 sc_hash(u32 gw, u32 t, u32 lane)
{
	u32 x = (gw * SC_TRIALS + t) * 64 + lane;
	x ^= x >> 16;
	x *= 0x7FEB352Du;
	x ^= x >> 15;
	x *= 0x846CA68Bu;
	x ^= x >> 16;
	const u32 cls = t % 5;
	const u32 k = cls == 0 ? 1 : cls == 1 ? 2 : cls == 2 ? 8 : cls == 3 ? 64 : 256;
	return x % k;
}

extern "C" __global__ void __launch_bounds__(1024)
lda_selfcheck_lds_order_kernel(u64 *__restrict__ counters /* [0] lanes, [1] mismatches, [2] conflicts */)
{
	__shared__ u32 tab[16][1 << (SC_TAB_BITS - 1)];
	__shared__ u16 want_tab[16][1 << SC_TAB_BITS];
	__shared__ u16 got[16][SC_TRIALS][64];
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const u32 gw = blockIdx.x * 16 + wave;

	for (u32 i = lane; i < (1u << (SC_TAB_BITS - 1)); i += 64)
		tab[wave][i] = 0x80008000u;
	for (u32 i = lane; i < (1u << SC_TAB_BITS); i += 64)
		want_tab[wave][i] = 0x8000;
	wave_sync();
	for (u32 t = 0; t < SC_TRIALS; t++) {
		const u32 h = sc_hash(gw, t, lane);
		const u32 pos = (t * 64 + lane) & 0x7FFF;
		const u32 sh = 16 * (h & 1);
		const u32 addr = (u32)(uintptr_t)(__attribute__((address_space(3))) u32 *)&tab[wave][h >> 1];
		u32 old;
		asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
			     : "=v"(old) : "v"(addr), "v"(0xFFFFu << sh), "v"(pos << sh) : "memory");
		got[wave][t][lane] = (u16)((old >> sh) & 0xFFFF);
	}
	wave_sync();
	/* the serial insertion loop, by lane 0 */
	if (lane == 0) {
		u32 bad = 0, conf = 0;
		for (u32 t = 0; t < SC_TRIALS; t++)
			for (u32 l = 0; l < 64; l++) {
				const u32 h = sc_hash(gw, t, l);
				const u32 pos = (t * 64 + l) & 0x7FFF;
				const u32 want = want_tab[wave][h];
				conf += want != 0x8000 && (want >> 6) == (pos >> 6);
				bad += got[wave][t][l] != want;
				want_tab[wave][h] = (u16)pos;
			}
		atomicAdd((unsigned long long *)&counters[0], (unsigned long long)SC_TRIALS * 64);
		if (bad)
			atomicAdd((unsigned long long *)&counters[1], (unsigned long long)bad);
		atomicAdd((unsigned long long *)&counters[2], (unsigned long long)conf);
	}
}

#define SC_ROUNDS 256
#define SC_REGION 8192	/* bytes per wave */

typedef __attribute__((address_space(1))) u8 sc_gu8;
typedef __attribute__((address_space(1))) u32 sc_gu32;

extern "C" __global__ void __launch_bounds__(64, 4)
lda_selfcheck_visibility_kernel(u8 *__restrict__ buf,
				u64 *__restrict__ counters /* [3] loads, [4] stale */)
{
	const u32 lane = threadIdx.x & 63;
	/* plain global accesses, like gout[] / gfar[] of par_round(): the
	 * compiler must not see through the store -> load pairs (it cannot: the
	 * loading lane differs from the storing one) and must not reorder them
	 * (the fences) */
	sc_gu8 *r = (sc_gu8 *)(buf + (size_t)blockIdx.x * SC_REGION);
	sc_gu32 *rw = (sc_gu32 *)r;
	unsigned long long bad = 0;
	u32 x = 0x9E3779B9u * (blockIdx.x + 1);

	for (u32 it = 0; it < SC_ROUNDS; it++) {
		x = x * 1664525u + 1013904223u;
		const u32 base = (x >> 8) % (SC_REGION - 1024);
		const u32 perm = ((x >> 3) | 1) & 63;	/* odd multiplier: a permutation of the lanes */
		const u32 val = it * 64 + lane;
		const u32 src = (lane * perm) & 63;
		/* the product's form: store, wait for the stores, load */
		r[base + lane] = (u8)val;
		const u32 wb = ((base + 512) & ~3u) / 4;
		rw[wb + lane] = val ^ 0xA5A5A5A5u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const u8 gotb = r[base + src];
		const u32 gotw = rw[wb + src];
		bad += gotb != (u8)(it * 64 + src);
		bad += gotw != ((it * 64 + src) ^ 0xA5A5A5A5u);
		wave_sync();
	}
	bad = wave_sum64(bad);
	if (lane == 0) {
		atomicAdd((unsigned long long *)&counters[3], 2ull * SC_ROUNDS * 64);
		if (bad)
			atomicAdd((unsigned long long *)&counters[4], bad);
	}
}

extern "C" size_t lda_selfcheck_region_bytes(void)
{
	return SC_REGION;
}
