/*
 * hwtest_lds_order.hip - does one LDS read-modify-write instruction whose lanes
 * hit the SAME address process its lanes in ascending lane order on gfx950?
 *
 * The chain insertion of deflate_kernel.hip wants, for 64 consecutive
 * positions at once, "previous position with my hash" = what a serial
 * insertion loop would return.  ds_mskor_rtn_b32 (masked exchange of one u16
 * half of a dword, returning the old dword) gives exactly that IF conflicting
 * lanes are served in lane order.  That order is not documented, so this
 * program measures it: random hash patterns with heavy duplication, many
 * trials, 16 waves per workgroup all hammering the same LDS, result compared
 * with a serial computation.  Build: hipcc --offload-arch=gfx950 -O2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define TRIALS 256
#define TAB_BITS 8	/* per-wave table: 256 u16 buckets = 128 dwords */

__global__ void __launch_bounds__(1024)
order_kernel(const uint32_t *__restrict__ hashes /* [waves][TRIALS][64] */,
	     uint32_t *__restrict__ got /* same shape: returned previous pos */)
{
	__shared__ uint32_t tab[16][1 << (TAB_BITS - 1)];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t gw = blockIdx.x * 16 + wave;

	for (uint32_t i = lane; i < (1u << (TAB_BITS - 1)); i += 64)
		tab[wave][i] = 0x80008000u;
	__builtin_amdgcn_wave_barrier();
	for (uint32_t t = 0; t < TRIALS; t++) {
		const uint32_t h = hashes[((size_t)gw * TRIALS + t) * 64 + lane];
		const uint32_t pos = (t * 64 + lane) & 0x7FFF;
		const uint32_t sh = 16 * (h & 1);
		const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&tab[wave][h >> 1];
		uint32_t old;
		asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
			     : "=v"(old) : "v"(addr), "v"(0xFFFFu << sh), "v"(pos << sh) : "memory");
		got[((size_t)gw * TRIALS + t) * 64 + lane] = (old >> sh) & 0xFFFF;
	}
}

int main(void)
{
	const int blocks = 512, waves = blocks * 16;
	const size_t n = (size_t)waves * TRIALS * 64;
	std::vector<uint32_t> h(n), g(n);
	uint64_t s = 0x9E3779B97F4A7C15ull;
	for (size_t i = 0; i < n; i++) {
		s ^= s << 13; s ^= s >> 7; s ^= s << 17;
		/* trial classes: all lanes equal / 2 values / 8 / 64 / 256 buckets */
		uint32_t cls = (uint32_t)((i / 64) % 5);
		uint32_t k = cls == 0 ? 1 : cls == 1 ? 2 : cls == 2 ? 8 : cls == 3 ? 64 : 256;
		h[i] = (uint32_t)(s >> 33) % k;
	}
	uint32_t *dh, *dg;
	if (hipMalloc(&dh, n * 4) != hipSuccess || hipMalloc(&dg, n * 4) != hipSuccess)
		return 2;
	hipMemcpy(dh, h.data(), n * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(order_kernel, dim3(blocks), dim3(1024), 0, 0, dh, dg);
	if (hipDeviceSynchronize() != hipSuccess)
		return 2;
	hipMemcpy(g.data(), dg, n * 4, hipMemcpyDeviceToHost);
	size_t bad = 0, conflicts = 0;
	for (int w = 0; w < waves; w++) {
		uint32_t tab[1 << TAB_BITS];
		for (auto &x : tab)
			x = 0x8000;
		for (int t = 0; t < TRIALS; t++)
			for (int l = 0; l < 64; l++) {
				size_t i = ((size_t)w * TRIALS + t) * 64 + l;
				uint32_t pos = (t * 64 + l) & 0x7FFF;
				uint32_t want = tab[h[i]];
				if (want != 0x8000 && (want >> 6) == (pos >> 6))
					conflicts++;
				if (g[i] != want && bad++ < 10)
					printf("mismatch wave %d trial %d lane %d: got %u want %u\n",
					       w, t, l, g[i], want);
				tab[h[i]] = pos;
			}
	}
	printf("{\"lanes\": %zu, \"same_instruction_conflicts\": %zu, \"mismatches\": %zu, "
	       "\"lane_order\": %s}\n", n, conflicts, bad, bad ? "false" : "true");
	return bad ? 1 : 0;
}
