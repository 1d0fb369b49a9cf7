/*
 * hwtest_global_visibility.hip - does a global-memory LOAD of a wave see what
 * ANOTHER LANE of the same wave STORED earlier in program order, without a
 * wait in between?
 *
 * The copy phase of inflate_kernel.hip (par_round) reads match sources that
 * lie further back than its LDS mirror from the output in HBM; such a byte
 * may have been stored a moment ago by a different lane of the same wave
 * (flush_ring).  Only a compiler barrier separates the two: correctness rests
 * on a wave's vector-memory operations being performed in issue order with
 * the store visible to a following load of the same CU (write-through L1
 * updated on a hit).  That is how gfx9 behaves but it is not a documented
 * contract, hence this program: every wave repeatedly stores a pattern with
 * one lane permutation and immediately loads it back with another, the way
 * the kernel does (byte and dword granularity, lines that are and are not
 * resident in the L1, 16 waves per CU hammering their own regions), and
 * counts stale reads.  Build: hipcc --offload-arch=gfx950 -O2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ROUNDS 4096
#define REGION 8192	/* bytes per wave */

__global__ void __launch_bounds__(64, 4)
vis_kernel(uint8_t *__restrict__ buf, unsigned long long *__restrict__ stale)
{
	const uint32_t lane = threadIdx.x & 63;
	volatile uint8_t *r = buf + (size_t)blockIdx.x * REGION;
	volatile uint32_t *rw = (volatile uint32_t *)r;
	unsigned long long bad = 0;
	uint32_t x = 0x9E3779B9u * (blockIdx.x + 1);

	for (uint32_t it = 0; it < ROUNDS; it++) {
		x = x * 1664525u + 1013904223u;
		const uint32_t base = (x >> 8) % (REGION - 1024);
		const uint32_t perm = ((x >> 3) | 1) & 63;	/* odd multiplier: a permutation of the lanes */
		const uint32_t val = it * 64 + lane;
		/* bytes: lane l stores byte l of a 64-byte slot, lane (l * perm) & 63 reads it */
		r[base + lane] = (uint8_t)val;
		__builtin_amdgcn_wave_barrier();	/* compiler barrier only, like wave_sync()'s */
		const uint32_t src = (lane * perm) & 63;
		const uint8_t got = r[base + src];
		bad += got != (uint8_t)(it * 64 + src);
		/* dwords, 16-byte stores side by side with byte loads elsewhere */
		const uint32_t wb = ((base + 512) & ~3u) / 4;
		rw[wb + lane] = val ^ 0xA5A5A5A5u;
		__builtin_amdgcn_wave_barrier();
		const uint32_t gw = rw[wb + src];
		bad += gw != ((it * 64 + src) ^ 0xA5A5A5A5u);
	}
	if (bad)
		atomicAdd(stale, bad);
}

int main(void)
{
	const int waves = 256 * 16;	/* 16 per CU on 256 CUs */
	uint8_t *buf;
	unsigned long long *d_stale, stale = 0;

	if (hipMalloc(&buf, (size_t)waves * REGION) != hipSuccess ||
	    hipMalloc(&d_stale, 8) != hipSuccess)
		return 2;
	hipMemset(buf, 0, (size_t)waves * REGION);
	hipMemset(d_stale, 0, 8);
	hipLaunchKernelGGL(vis_kernel, dim3(waves), dim3(64), 0, 0, buf, d_stale);
	if (hipDeviceSynchronize() != hipSuccess)
		return 2;
	hipMemcpy(&stale, d_stale, 8, hipMemcpyDeviceToHost);
	printf("{\"loads\": %llu, \"stale\": %llu, \"store_then_load_visible\": %s}\n",
	       2ull * waves * ROUNDS * 64, stale, stale ? "false" : "true");
	return stale ? 1 : 0;
}
