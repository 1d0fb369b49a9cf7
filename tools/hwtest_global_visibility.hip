/*
 * hwtest_global_visibility.hip - does a global-memory LOAD of a wave see what
 * ANOTHER LANE of the same wave STORED earlier in program order?
 *
 * The copy phase of inflate_kernel.hip (par_round) reads match sources that
 * lie further back than its LDS mirror from the output in HBM; such a byte
 * may have been stored earlier in the round by a different lane of the same
 * wave (flush_ring).  Two forms are measured, both with the product's own
 * kind of access - PLAIN global stores and loads through address-space-1
 * pointers (no volatile: a volatile access compiles to sc0 sc1, system scope,
 * and would bypass the very L1 behaviour in question; tests/test_abi.py
 * checks that neither this file's nor the product's loads carry sc bits):
 *
 *   waited   store; s_waitcnt vmcnt(0); load - what the product does since
 *            round 6 (global_stores_visible(): the compiler's workgroup-scope
 *            release / acquire sequence on gfx9).  Architected; a stale load
 *            here fails the test.
 *   nowait   store; wavefront-scope fence (no instruction on gfx9); load -
 *            what the product did until round 5, resting on a wave's vector
 *            memory operations being performed in issue order with a
 *            write-through L1 updated on a hit.  Reported; nothing depends on
 *            it any more.
 *
 * Every wave repeatedly stores a pattern with one lane permutation and loads
 * it back with another (byte and dword granularity, lines that are and are
 * not resident in the L1, 16 waves per CU hammering their own regions).
 * Build: hipcc --offload-arch=gfx950 -O2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ROUNDS 4096
#define REGION 8192	/* bytes per wave */

typedef __attribute__((address_space(1))) uint8_t gu8;
typedef __attribute__((address_space(1))) uint32_t gu32;

static __device__ __forceinline__ void fence_wave(void)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool WAIT> static __device__ __forceinline__ void fence_form(void)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	if (WAIT)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool WAIT> static __device__ unsigned long long
vis_rounds(gu8 *r, uint32_t seed)
{
	gu32 *rw = (gu32 *)r;
	unsigned long long bad = 0;
	uint32_t x = seed;

	for (uint32_t it = 0; it < ROUNDS; it++) {
		x = x * 1664525u + 1013904223u;
		const uint32_t base = (x >> 8) % (REGION - 1024);
		const uint32_t perm = ((x >> 3) | 1) & 63;	/* odd multiplier: a permutation of the lanes */
		const uint32_t lane = threadIdx.x & 63;
		const uint32_t val = it * 64 + lane;
		/* bytes: lane l stores byte l of a 64-byte slot, lane (l * perm) & 63 reads it */
		r[base + lane] = (uint8_t)val;
		fence_form<WAIT>();
		const uint32_t src = (lane * perm) & 63;
		const uint8_t got = r[base + src];
		bad += got != (uint8_t)(it * 64 + src);
		/* dwords */
		const uint32_t wb = ((base + 512) & ~3u) / 4;
		rw[wb + lane] = val ^ 0xA5A5A5A5u;
		fence_form<WAIT>();
		const uint32_t gw = rw[wb + src];
		bad += gw != ((it * 64 + src) ^ 0xA5A5A5A5u);
		fence_wave();
	}
	return bad;
}

__global__ void __launch_bounds__(64, 4)
vis_kernel(uint8_t *__restrict__ buf, unsigned long long *__restrict__ stale)
{
	gu8 *r = (gu8 *)(buf + (size_t)blockIdx.x * REGION);
	const uint32_t seed = 0x9E3779B9u * (blockIdx.x + 1);
	const unsigned long long w = vis_rounds<true>(r, seed);
	const unsigned long long n = vis_rounds<false>(r, seed ^ 0x55555555u);

	if (w)
		atomicAdd(&stale[0], w);
	if (n)
		atomicAdd(&stale[1], n);
}

int main(void)
{
	const int waves = 256 * 16;	/* 16 per CU on 256 CUs */
	uint8_t *buf;
	unsigned long long *d_stale, stale[2] = { 0, 0 };

	if (hipMalloc(&buf, (size_t)waves * REGION) != hipSuccess ||
	    hipMalloc(&d_stale, 16) != hipSuccess)
		return 2;
	hipMemset(buf, 0, (size_t)waves * REGION);
	hipMemset(d_stale, 0, 16);
	hipLaunchKernelGGL(vis_kernel, dim3(waves), dim3(64), 0, 0, buf, d_stale);
	if (hipDeviceSynchronize() != hipSuccess)
		return 2;
	hipMemcpy(stale, d_stale, 16, hipMemcpyDeviceToHost);
	printf("{\"loads\": %llu, \"stale\": %llu, \"store_then_load_visible\": %s, "
	       "\"loads_without_wait\": %llu, \"stale_without_wait\": %llu}\n",
	       2ull * waves * ROUNDS * 64, stale[0], stale[0] ? "false" : "true",
	       2ull * waves * ROUNDS * 64, stale[1]);
	return stale[0] ? 1 : 0;
}
