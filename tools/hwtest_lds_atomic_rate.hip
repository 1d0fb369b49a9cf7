/* hwtest_lds_atomic_rate.hip - cycles per wave instruction of LDS atomics with
 * return (random addresses, 64 active lanes), one wave alone on a CU and 16
 * waves of one workgroup together.  Design input for the chain insertion of
 * deflate_kernel.hip.  hipcc --offload-arch=gfx950 -O2. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define N 512
template <int OP> __global__ void __launch_bounds__(1024)
rate_kernel(uint64_t *out, uint32_t seed)
{
	__shared__ uint32_t tab[8192];
	const uint32_t lane = threadIdx.x & 63;
	for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x)
		tab[i] = i;
	__syncthreads();
	uint32_t x = seed * 2654435761u + threadIdx.x * 40503u, acc = 0;
	uint64_t t0 = __builtin_readcyclecounter();
	for (int i = 0; i < N; i += 8) {
		uint32_t o[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			x = x * 1664525u + 1013904223u;
			uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&tab[(x >> 12) & 4095];
			uint32_t v = x >> 3;
			if (OP == 0) asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "=v"(o[k]) : "v"(a), "v"(0xFFFFu), "v"(v & 0xFFFF) : "memory");
			if (OP == 1) asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2" : "=v"(o[k]) : "v"(a), "v"(v) : "memory");
			if (OP == 2) asm volatile("ds_max_rtn_u32 %0, %1, %2" : "=v"(o[k]) : "v"(a), "v"(v) : "memory");
			if (OP == 3) asm volatile("ds_add_u32 %1, %2" : "=v"(o[k]) : "v"(a), "v"(v) : "memory");
			if (OP == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(o[k]) : "v"(a) : "memory");
			if (OP == 5) asm volatile("ds_write_b32 %1, %2" : "=v"(o[k]) : "v"(a), "v"(v) : "memory");
			if (OP == 6) asm volatile("ds_read_u16 %0, %1\n\tds_write_b16 %1, %2" : "=v"(o[k]) : "v"(a), "v"(v) : "memory");
		}
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]) :: "memory");
#pragma unroll
		for (int k = 0; k < 8; k++)
			acc += o[k];
	}
	uint64_t t1 = __builtin_readcyclecounter();
	if (lane == 0)
		out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (t1 - t0) | ((uint64_t)(acc & 1) << 63);
}

template <int OP> static void run(const char *name)
{
	uint64_t *d, h[16];
	hipMalloc(&d, 4096 * 8);
	for (int waves = 1; waves <= 16; waves *= 4) {
		hipLaunchKernelGGL(rate_kernel<OP>, dim3(1), dim3(64 * waves), 0, 0, d, 7u);
		hipLaunchKernelGGL(rate_kernel<OP>, dim3(1), dim3(64 * waves), 0, 0, d, 9u);
		hipDeviceSynchronize();
		hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		double mx = 0;
		for (int w = 0; w < waves; w++) {
			double c = (double)(h[w] & ~(1ull << 63));
			if (c > mx) mx = c;
		}
		printf("%-18s %2d waves: %.1f cycles per wave-instruction (per wave), %.1f per instruction CU-wide\n",
		       name, waves, mx / N, mx / N / waves);
	}
}

int main(void)
{
	run<0>("ds_mskor_rtn_b32");
	run<1>("ds_wrxchg_rtn_b32");
	run<2>("ds_max_rtn_u32");
	run<3>("ds_add_u32 (no rtn)");
	run<4>("ds_read_b32");
	run<5>("ds_write_b32");
	run<6>("read_u16+write_b16");
	return 0;
}
