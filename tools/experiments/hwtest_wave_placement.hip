/*
 * hwtest_wave_placement.hip - where do the waves of a one-wave-per-workgroup
 * launch land, and does a wave know it?
 *
 * lda_inflate_wave_kernel (inflate_kernel.hip, PLACEMENT) deals the streams of
 * a batch of one stream per wave slot to the SIMDs by cost: every wave reads
 * its place from HW_REG_HW_ID / HW_REG_XCC_ID and takes its stream
 * accordingly.  Correctness does not depend on it (entries are claimed with
 * atomics, leftovers go through a cursor); the BALANCE does: it assumes that
 * the key built from those registers tells the SIMDs apart (1024 distinct keys
 * on 256 CUs) and that a launch of 16 waves per CU with the kernel's footprint
 * (10 KiB of LDS per wave, at most 128 VGPRs) puts four waves on every SIMD.
 * This program launches such a grid with the same key function and reports
 * what it sees.  Build: hipcc --offload-arch=gfx950 -O2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define KEYS 4096u

static __device__ __forceinline__ uint32_t place_key(void)
{
	/* (the same as inflate_kernel.hip) */
	const uint32_t hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
	const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
	return ((xcc & 7) << 9) | (((hw >> 13) & 3) << 7) | (((hw >> 12) & 1) << 6) |
	       (((hw >> 8) & 15) << 2) | ((hw >> 4) & 3);
}

__global__ void __launch_bounds__(64, 4)
place_kernel(uint32_t *__restrict__ arrivals, uint32_t *__restrict__ raw, uint64_t spin)
{
	extern __shared__ uint8_t lds[];
	if (threadIdx.x == 0) {
		const uint32_t key = place_key();
		atomicAdd(&arrivals[key], 1u);
		raw[blockIdx.x] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
		lds[0] = (uint8_t)key;
	}
	/* stay resident until the whole grid has been dispatched */
	const uint64_t t0 = __builtin_readcyclecounter();
	while (__builtin_readcyclecounter() - t0 < spin)
		__builtin_amdgcn_s_sleep(8);
}

int main(void)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, 0) != hipSuccess)
		return 2;
	const unsigned cus = (unsigned)prop.multiProcessorCount, grid = 16 * cus;
	uint32_t *d_arr, *d_raw;
	static uint32_t arr[KEYS];

	if (hipMalloc(&d_arr, KEYS * 4) != hipSuccess || hipMalloc(&d_raw, grid * 4) != hipSuccess)
		return 2;
	hipMemset(d_arr, 0, KEYS * 4);
	hipFuncSetAttribute((const void *)place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 10240);
	hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(64), 10240, 0, d_arr, d_raw, 4000000ull);
	if (hipDeviceSynchronize() != hipSuccess)
		return 2;
	hipMemcpy(arr, d_arr, sizeof(arr), hipMemcpyDeviceToHost);
	unsigned keys = 0, hist[8] = { 0 }, mx = 0;
	for (unsigned k = 0; k < KEYS; k++) {
		if (!arr[k])
			continue;
		keys++;
		hist[arr[k] < 7 ? arr[k] : 7]++;
		mx = arr[k] > mx ? arr[k] : mx;
	}
	printf("{\"cus\": %u, \"waves\": %u, \"distinct_simd_keys\": %u, \"expected_keys\": %u, "
	       "\"keys_with_4_waves\": %u, \"max_waves_per_key\": %u, "
	       "\"waves_per_key_histogram\": [%u, %u, %u, %u, %u, %u, %u, %u]}\n",
	       cus, grid, keys, 4 * cus, hist[4], mx, hist[0], hist[1], hist[2], hist[3], hist[4],
	       hist[5], hist[6], hist[7]);
	return 0;
}
