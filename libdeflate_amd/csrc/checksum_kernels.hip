/*
 * checksum_kernels.hip - batched CRC-32 and Adler-32 for gfx950.
 *
 * Replaces, for a batch of independent chunks resident in HBM:
 *   libdeflate_crc32    lib/crc32.c:256-262 (slice-by-8 core :176-204, and the
 *                       fold-by-x^d structure of scripts/gen-crc32-consts.py:65-87)
 *   libdeflate_adler32  lib/adler32.c:156-162 (ADLER32_CHUNK :75-103)
 *
 * Mapping: one 64-lane wavefront per chunk, 4 chunks per 256-thread
 * workgroup, persistent grid striding over the batch.  Both kernels stream
 * the chunk once with coalesced 16-byte-per-lane loads (1 KiB per wave
 * instruction) and are HBM-bound by construction; algorithmic traffic is
 * exactly the chunk bytes + 4.
 *
 * CRC-32 without carry-less multiply (CDNA4 has none): a *lane-strided
 * slice-by-16*.  Lane l owns the 16-byte blocks at row*1024 + 16*l.  Its
 * 32-bit register is carried from one of its blocks to the next - a distance
 * of 1024 bytes - by 16 table look-ups S_k[byte k], where S_k[b] is the
 * register left by byte b followed by (15-k)+1008 zero bytes.  The tables
 * (16 KiB) sit in LDS and are shared by the 4 waves of the workgroup.  The
 * ragged head (to reach 16-byte alignment), the last full row and the tail
 * are folded with a per-lane bytewise pass plus one GF(2) multiplication by
 * x^(8*bytes-after-my-block) and a wave XOR reduction.
 */
#include "device_common.h"
#include "kernels.h"

#define CRC_POLY 0xEDB88320u
#define ROW 1024u	/* bytes per wave row: 64 lanes x 16 B */

/*
 * a(x) * b(x) mod P in the reflected representation used by the register
 * (x^0 at bit 31).  Branch-free 32-step shift-and-add.
 */
static __device__ __forceinline__ u32 gf2_mulmod(u32 a, u32 b)
{
	u32 p = 0;
#pragma unroll 8
	for (int i = 31; i >= 0; i--) {
		p ^= b & (0u - ((a >> i) & 1u));
		b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
	}
	return p;
}

struct crc_lds {
	u32 stride[16][256];	/* S_k, see header comment */
	u32 t0[256];		/* classic byte table */
};

/*
 * CRC register after the bytes [0, len) of 'p' (len <= 1024) starting from
 * register 'reg'.  Lane l folds bytes [16l, 16l+16) bytewise, multiplies by
 * x^(8 * bytes-after-its-block) and the wave XORs the pieces.
 * 'blk' is the lane's block already loaded when have_blk is set (full row).
 */
static __device__ __forceinline__ u32
crc_row_generic(const crc_lds &L, const u32 *__restrict__ xpow8,
		const u8 *__restrict__ p, u32 len, u32 reg, u32 lane)
{
	u32 start = lane * 16;
	u32 r = (lane == 0) ? reg : 0;
	u32 part = 0;

	if (start < len) {
		u32 m = len - start;
		if (m > 16)
			m = 16;
		for (u32 k = 0; k < m; k++)
			r = L.t0[(r ^ p[start + k]) & 0xFF] ^ (r >> 8);
		part = gf2_mulmod(xpow8[len - start - m], r);
	} else if (lane == 0) {
		part = r;	/* len == 0: register passes through */
	}
	return wave_xor(part);
}

/* same, for a full 1 KiB row whose 16-byte blocks are already in registers */
static __device__ __forceinline__ u32
crc_row_full_final(const crc_lds &L, const u32 *__restrict__ xpow8,
		   uint4 v, u32 reg_lane, u32 lane)
{
	u32 w[4] = { v.x ^ reg_lane, v.y, v.z, v.w };
	u32 r = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) {
#pragma unroll
		for (int j = 0; j < 4; j++)
			r = L.t0[(r ^ (w[i] >> (8 * j))) & 0xFF] ^ (r >> 8);
	}
	return wave_xor(gf2_mulmod(xpow8[(63 - lane) * 16], r));
}

extern "C" __global__ void __launch_bounds__(256)
lda_crc32_batch_kernel(u64 n_chunks, const u8 *__restrict__ base,
		       const u64 *__restrict__ offsets,
		       const u64 *__restrict__ nbytes,
		       const u32 *__restrict__ init,
		       u32 *__restrict__ out,
		       const u32 *__restrict__ g_tables /* 17*256 */,
		       const u32 *__restrict__ xpow8 /* 1024 */)
{
	__shared__ crc_lds L;
	const u32 lane = threadIdx.x & 63;
	const u32 wave = threadIdx.x >> 6;

	for (u32 i = threadIdx.x; i < 17 * 256; i += 256)
		((u32 *)&L)[i] = g_tables[i];
	__syncthreads();

	for (u64 c = (u64)blockIdx.x * 4 + wave; c < n_chunks;
	     c += (u64)gridDim.x * 4) {
		const u8 *p = base + offsets[c];
		u64 n = nbytes[c];
		u32 reg = ~(init ? init[c] : 0u);

		/* head: up to 15 bytes so that full rows are 16-byte aligned */
		u32 head = (u32)((0 - (uintptr_t)p) & 15);
		if (head > n)
			head = (u32)n;
		if (head) {
			reg = crc_row_generic(L, xpow8, p, head, reg, lane);
			p += head;
			n -= head;
		}
		u64 rows = n / ROW;
		if (rows) {
			const uint4 *q = (const uint4 *)p + lane;
			u32 r = (lane == 0) ? reg : 0;
			uint4 v = q[0];
			for (u64 row = 0; row + 1 < rows; row++) {
				uint4 nv = q[(row + 1) * 64];
				u32 w0 = v.x ^ r, w1 = v.y, w2 = v.z, w3 = v.w;
				r = L.stride[0][w0 & 0xFF] ^
				    L.stride[1][(w0 >> 8) & 0xFF] ^
				    L.stride[2][(w0 >> 16) & 0xFF] ^
				    L.stride[3][w0 >> 24] ^
				    L.stride[4][w1 & 0xFF] ^
				    L.stride[5][(w1 >> 8) & 0xFF] ^
				    L.stride[6][(w1 >> 16) & 0xFF] ^
				    L.stride[7][w1 >> 24] ^
				    L.stride[8][w2 & 0xFF] ^
				    L.stride[9][(w2 >> 8) & 0xFF] ^
				    L.stride[10][(w2 >> 16) & 0xFF] ^
				    L.stride[11][w2 >> 24] ^
				    L.stride[12][w3 & 0xFF] ^
				    L.stride[13][(w3 >> 8) & 0xFF] ^
				    L.stride[14][(w3 >> 16) & 0xFF] ^
				    L.stride[15][w3 >> 24];
				v = nv;
			}
			reg = crc_row_full_final(L, xpow8, v, r, lane);
			p += rows * ROW;
			n -= rows * ROW;
		}
		if (n)
			reg = crc_row_generic(L, xpow8, p, (u32)n, reg, lane);
		if (lane == 0)
			out[c] = ~reg;
	}
}

/* ------------------------------------------------------------------ */
/* Adler-32                                                             */
/* ------------------------------------------------------------------ */

#define ADLER_MOD 65521u
#define ADLER_SEG (1u << 24)	/* per-segment sums stay below 2^57 */

static __device__ __forceinline__ u32 dot4(u32 bytes, u32 weights, u32 acc)
{
	return __builtin_amdgcn_udot4(bytes, weights, acc, false);
}

/*
 * s1 += sum b_i ; s2 += sum (n - i) * b_i + n * s1_before, per segment.
 * Each lane accumulates S = sum of its bytes and W = sum (n - i) * b_i in
 * 64 bits; one wave reduction and two 64-bit modulo operations per segment.
 */
extern "C" __global__ void __launch_bounds__(256)
lda_adler32_batch_kernel(u64 n_chunks, const u8 *__restrict__ base,
			 const u64 *__restrict__ offsets,
			 const u64 *__restrict__ nbytes,
			 const u32 *__restrict__ init,
			 u32 *__restrict__ out)
{
	const u32 lane = threadIdx.x & 63;
	const u32 wave = threadIdx.x >> 6;

	for (u64 c = (u64)blockIdx.x * 4 + wave; c < n_chunks;
	     c += (u64)gridDim.x * 4) {
		const u8 *p = base + offsets[c];
		u64 left = nbytes[c];
		u32 a0 = init ? init[c] : 1u;
		u64 s1 = a0 & 0xFFFF, s2 = a0 >> 16;

		/* lib/adler32.c:105-119: an empty buffer returns the value as is */
		while (left) {
			u32 n = left > ADLER_SEG ? ADLER_SEG : (u32)left;
			u64 S = 0, W = 0;
			u32 pos = 0;
			u32 head = (u32)((0 - (uintptr_t)p) & 15);

			if (head > n)
				head = n;
			if (lane < head) {
				u32 b = p[lane];
				S += b;
				W += (u64)(n - lane) * b;
			}
			pos = head;
			/* full 1 KiB rows, 16 B per lane */
			u32 rows = (n - pos) / ROW;
			const uint4 *q = (const uint4 *)(p + pos) + lane;
			for (u32 row = 0; row < rows; row++) {
				uint4 v = q[(u64)row * 64];
				u32 o = pos + row * ROW + lane * 16;
				u32 bs = dot4(v.x, 0x01010101u, 0);
				bs = dot4(v.y, 0x01010101u, bs);
				bs = dot4(v.z, 0x01010101u, bs);
				bs = dot4(v.w, 0x01010101u, bs);
				u32 ks = dot4(v.x, 0x03020100u, 0);
				ks = dot4(v.y, 0x07060504u, ks);
				ks = dot4(v.z, 0x0B0A0908u, ks);
				ks = dot4(v.w, 0x0F0E0D0Cu, ks);
				S += bs;
				W += (u64)(n - o) * bs - ks;
			}
			pos += rows * ROW;
			/* tail < 1 KiB: bytes strided across lanes */
			for (u32 i = pos + lane; i < n; i += 64) {
				u32 b = p[i];
				S += b;
				W += (u64)(n - i) * b;
			}
			S = wave_sum64(S);
			W = wave_sum64(W);
			s2 = (s2 + (u64)n % ADLER_MOD * s1 + W % ADLER_MOD) % ADLER_MOD;
			s1 = (s1 + S) % ADLER_MOD;
			p += n;
			left -= n;
		}
		if (lane == 0)
			out[c] = (u32)((s2 << 16) | s1);
	}
}
