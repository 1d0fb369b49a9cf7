/*
 * compact_kernels.hip - device-side compaction of a batch's ragged outputs.
 *
 * A compress batch leaves stream i in a slot sized by compress_bound; callers
 * that ship the bytes on (the multi-GPU payload gather, the host-pointer
 * batch entry points, a file writer) want them back to back, the way the
 * reference's callers concatenate the return values of
 * libdeflate_*_compress (lib/deflate_compress.c:4064-4068 returns the size
 * that makes that possible; programs/gzip.c:149-185 writes exactly that many
 * bytes).  Three launches, all HBM-bound:
 *
 *   lda_scan_local_kernel    exclusive prefix sum of the sizes inside blocks
 *                            of 2048 chunks + the block totals
 *   lda_scan_blocks_kernel   exclusive prefix sum of the block totals (one
 *                            workgroup), grand total
 *   lda_compact_copy_kernel  one 256-thread workgroup per chunk: final offset
 *                            = local prefix + block prefix, then a copy with
 *                            16-byte stores to the aligned part of the
 *                            destination
 */
#include "device_common.h"
#include "kernels.h"

#define SCAN_THREADS 256
#define SCAN_PER_THREAD 8
#define SCAN_BLOCK (SCAN_THREADS * SCAN_PER_THREAD)

static __device__ __forceinline__ u64 wave_scan_incl64(u64 v)
{
	const u32 lane = lane_id();
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		u32 lo = __shfl_up((u32)v, off, 64);
		u32 hi = __shfl_up((u32)(v >> 32), off, 64);
		if (lane >= (u32)off)
			v += ((u64)hi << 32) | lo;
	}
	return v;
}

extern "C" __global__ void __launch_bounds__(SCAN_THREADS)
lda_scan_local_kernel(u64 n, const u64 *__restrict__ sizes,
		      u64 *__restrict__ offsets, u64 *__restrict__ block_sums)
{
	__shared__ u64 wsum[SCAN_THREADS / 64];
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const u64 base = (u64)blockIdx.x * SCAN_BLOCK + (u64)tid * SCAN_PER_THREAD;
	u64 v[SCAN_PER_THREAD], mine = 0;

#pragma unroll
	for (int k = 0; k < SCAN_PER_THREAD; k++) {
		v[k] = base + k < n ? sizes[base + k] : 0;
		mine += v[k];
	}
	u64 incl = wave_scan_incl64(mine);
	if (lane == 63)
		wsum[wave] = incl;
	__syncthreads();
	u64 pre = incl - mine, tot = 0;
#pragma unroll
	for (u32 w = 0; w < SCAN_THREADS / 64; w++) {
		u64 s = wsum[w];
		if (w < wave)
			pre += s;
		tot += s;
	}
#pragma unroll
	for (int k = 0; k < SCAN_PER_THREAD; k++) {
		if (base + k < n)
			offsets[base + k] = pre;
		pre += v[k];
	}
	if (tid == 0)
		block_sums[blockIdx.x] = tot;
}

/* in place: block_sums[b] := sum of the totals before block b;
 * block_sums[nblocks] := grand total.  One workgroup. */
extern "C" __global__ void __launch_bounds__(1024)
lda_scan_blocks_kernel(u64 nblocks, u64 *__restrict__ block_sums)
{
	__shared__ u64 wsum[16];
	__shared__ u64 carry_s;
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

	if (tid == 0)
		carry_s = 0;
	__syncthreads();
	for (u64 b0 = 0; b0 < nblocks; b0 += 1024) {
		u64 mine = b0 + tid < nblocks ? block_sums[b0 + tid] : 0;
		u64 incl = wave_scan_incl64(mine);
		if (lane == 63)
			wsum[wave] = incl;
		__syncthreads();
		u64 pre = carry_s + incl - mine, tot = 0;
#pragma unroll
		for (u32 w = 0; w < 16; w++) {
			u64 s = wsum[w];
			if (w < wave)
				pre += s;
			tot += s;
		}
		if (b0 + tid < nblocks)
			block_sums[b0 + tid] = pre;
		__syncthreads();
		if (tid == 0)
			carry_s += tot;
		__syncthreads();
	}
	if (tid == 0)
		block_sums[nblocks] = carry_s;
}

extern "C" __global__ void __launch_bounds__(256)
lda_compact_copy_kernel(u64 n, const u8 *__restrict__ in_base,
			const u64 *__restrict__ in_offsets,
			const u64 *__restrict__ sizes, u8 *__restrict__ out_base,
			u64 *__restrict__ offsets,
			const u64 *__restrict__ block_sums)
{
	const u32 tid = threadIdx.x;

	for (u64 c = blockIdx.x; c < n; c += gridDim.x) {
		const u64 start = offsets[c] + block_sums[c / SCAN_BLOCK];
		const u64 len = sizes[c];
		const u8 *src = in_base + in_offsets[c];
		u8 *dst = out_base + start;

		__syncthreads();	/* every thread has read offsets[c] */
		if (tid == 0) {
			offsets[c] = start;
			if (c == n - 1)
				offsets[n] = block_sums[(n + SCAN_BLOCK - 1) / SCAN_BLOCK];
		}
		/* head: up to the first 16-byte boundary of the destination */
		u64 head = (0 - (uintptr_t)dst) & 15;
		if (head > len)
			head = len;
		if (tid < head)
			dst[tid] = src[tid];
		const u64 body = (len - head) & ~(u64)15;
		for (u64 k = head + 16 * (u64)tid; k < head + body; k += 16 * 256) {
			uint4 v;
			__builtin_memcpy(&v, src + k, 16);	/* source may be unaligned */
			*(uint4 *)(dst + k) = v;
		}
		const u64 tail = head + body;
		if (tail + tid < len)
			dst[tail + tid] = src[tail + tid];
	}
}
