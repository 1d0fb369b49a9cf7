/*
 * device_common.h - small wave64 helpers shared by the gfx950 kernels.
 * CDNA4 only: a wavefront is 64 lanes; no warp-size abstraction.
 */
#ifndef LDA_DEVICE_COMMON_H
#define LDA_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define LDA_WAVE 64

/*
 * __ballot() of the HIP headers takes an int: a predicate that is not one plain
 * compare is turned into 0 / 1 in a vector register and compared again (two
 * vector instructions per use, in loops whose cost is their instruction count);
 * the builtin takes the predicate as it is.  Where a predicate is a conjunction
 * of compares, one ballot per compare and the masks combined on the scalar
 * unit is cheaper still (rb_batch() in deflate_kernel.hip).
 */
#ifndef BALLOT_BUILTIN
#define BALLOT_BUILTIN 1
#endif
#if BALLOT_BUILTIN
#define __ballot(p) __builtin_amdgcn_ballot_w64((bool)(p))
#endif

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t s32;
typedef int64_t s64;

/* result codes: libdeflate.h:194-209 */
#define LDA_SUCCESS 0
#define LDA_BAD_DATA 1
#define LDA_SHORT_OUTPUT 2
#define LDA_INSUFFICIENT_SPACE 3

#define LDA_FMT_DEFLATE 0
#define LDA_FMT_ZLIB 1
#define LDA_FMT_GZIP 2

static __device__ __forceinline__ u32 lane_id(void)
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}

/* value of 'v' in the first active lane, as a wave-uniform (SGPR) value */
static __device__ __forceinline__ u32 bcast_first(u32 v)
{
	return __builtin_amdgcn_readfirstlane(v);
}

/* 64-bit value of the first active lane */
static __device__ __forceinline__ u64 bcast64(u64 v)
{
	/* the builtin returns int: without the casts the low half would be
	 * sign-extended over the high half */
	return ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32) |
	       (u32)__builtin_amdgcn_readfirstlane((u32)v);
}

static __device__ __forceinline__ u32 bcast_lane(u32 v, u32 lane)
{
	return __builtin_amdgcn_readlane(v, lane);
}

/*
 * v from lane (lane ^ J), J a power of two known at compile time.  Within a
 * row of 16 lanes this is one or two DPP moves (quad_perm for 1 and 2;
 * reversals of 8 and 16 lanes compose to 4 and 8), across rows of a 32-lane
 * half a ds_swizzle; only J = 32 needs the LDS permute.
 */
template <u32 J> static __device__ __forceinline__ u32 lane_xor(u32 v)
{
	if (J == 1)
		return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
	if (J == 2)
		return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
	if (J == 4) {	/* reverse 8, then reverse 4: xor 7 ^ 3 */
		u32 t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
		return __builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, false);
	}
	if (J == 8) {	/* reverse 16, then reverse 8: xor 15 ^ 7 */
		u32 t = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
		return __builtin_amdgcn_update_dpp(0, t, 0x141, 0xF, 0xF, false);
	}
	if (J == 16)
		return __builtin_amdgcn_ds_swizzle(v, 0x401F);
	return __shfl_xor(v, 32, 64);
}

static __device__ __forceinline__ u32 wave_xor(u32 v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v ^= __shfl_xor(v, off, 64);
	return v;
}

static __device__ __forceinline__ u64 wave_sum64(u64 v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		u32 lo = __shfl_xor((u32)v, off, 64);
		u32 hi = __shfl_xor((u32)(v >> 32), off, 64);
		v += ((u64)hi << 32) | lo;
	}
	return v;
}

/* inclusive prefix sum across the 64 lanes: DPP row shifts inside the rows of
 * 16, then row_bcast15 / row_bcast31 carry the row totals (no LDS permutes) */
static __device__ __forceinline__ u32 wave_scan_incl(u32 v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);	/* row_shr:1 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);	/* row_shr:2 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);	/* row_shr:4 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);	/* row_shr:8 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);	/* row_bcast15 -> rows 1, 3 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);	/* row_bcast31 -> rows 2, 3 */
	return v;
}

/* sum over lanes 0..15 (the first row), wave-uniform; the caller's lanes 16..63
 * take no part */
static __device__ __forceinline__ u32 row16_sum(u32 v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);	/* row_shr:1 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);	/* row_shr:2 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);	/* row_shr:4 */
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);	/* row_shr:8 */
	return (u32)__builtin_amdgcn_readlane((int)v, 15);
}

/* sum over the wave, wave-uniform: the last lane of the DPP scan */
static __device__ __forceinline__ u32 wave_sum(u32 v)
{
	return (u32)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63);
}

static __device__ __forceinline__ u32 wave_max(u32 v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		u32 t = __shfl_xor(v, off, 64);
		v = t > v ? t : v;
	}
	return v;
}

/* make earlier LDS/global writes of this wave visible to its other lanes */
static __device__ __forceinline__ void wave_sync(void)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/*
 * Optional phase profiling (make PROFILE=1 builds libdeflate_amd_prof.so):
 * thread 0 of each workgroup accumulates s_memtime deltas per phase.
 */
#ifdef LDA_PROFILE
#define LDA_PROF_SLOTS 40
static __device__ unsigned long long lda_prof[LDA_PROF_SLOTS];	/* per TU */
/* exported reader for this translation unit's counters (reads and resets) */
#define LDA_PROF_DEFINE_READER(name)                                          \
	extern "C" __attribute__((visibility("default"))) void name(           \
		unsigned long long *out)                                      \
	{                                                                     \
		unsigned long long z[LDA_PROF_SLOTS] = { 0 };                 \
		(void)hipDeviceSynchronize();                                 \
		(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lda_prof), sizeof(z)); \
		(void)hipMemcpyToSymbol(HIP_SYMBOL(lda_prof), z, sizeof(z));  \
	}
#define PROF_DECL unsigned long long prof_t_ = 0
#define PROF_START() do { if (threadIdx.x == 0) prof_t_ = __builtin_readcyclecounter(); } while (0)
#define PROF_MARK(slot) do { if (threadIdx.x == 0) { \
		unsigned long long n_ = __builtin_readcyclecounter(); \
		atomicAdd(&lda_prof[slot], n_ - prof_t_); prof_t_ = n_; } } while (0)
/* section timers of any wave (its lane 0): PROF_W0 starts, PROF_W adds the
 * time since the last PROF_W0 / PROF_W of this wave to a slot */
#define PROF_WDECL unsigned long long prof_w_ = 0
#define PROF_W0() do { prof_w_ = __builtin_readcyclecounter(); } while (0)
#define PROF_W(slot) do { unsigned long long n_ = __builtin_readcyclecounter(); \
		if ((threadIdx.x & 63) == 0) atomicAdd(&lda_prof[slot], n_ - prof_w_); \
		prof_w_ = n_; } while (0)
#ifdef LDA_PROFILE_COUNTS	/* event counters distort the phase times */
#define PROF_COUNT(slot, v) do { unsigned long long v_ = (v); if (threadIdx.x == 0) \
		atomicAdd(&lda_prof[slot], v_); } while (0)
/* section timers kept in registers, flushed once per tile by PROF_SEC_FLUSH */
#define PROF_SEC_DECL unsigned long long sec_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, sec_t_ = __builtin_readcyclecounter()
#define PROF_SEC(i) do { unsigned long long n_ = __builtin_readcyclecounter(); \
		sec_[i] += n_ - sec_t_; sec_t_ = n_; } while (0)
#define PROF_SEC_FLUSH(base) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 4; i_++) \
		atomicAdd(&lda_prof[(base) + i_], sec_[i_]); } while (0)
#define PROF_SEC_FLUSH8(base) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) \
		atomicAdd(&lda_prof[(base) + i_], sec_[i_]); } while (0)
#define PROF_SEC_ADD(i, v) do { sec_[i] += (v); } while (0)
#else
#define PROF_COUNT(slot, v) do { } while (0)
#define PROF_SEC_DECL
#define PROF_SEC(i) do { } while (0)
#define PROF_SEC_FLUSH(base) do { } while (0)
#define PROF_SEC_FLUSH8(base) do { } while (0)
#define PROF_SEC_ADD(i, v) do { } while (0)
#endif
#else
#define PROF_COUNT(slot, v) do { } while (0)
#define PROF_SEC_DECL
#define PROF_SEC(i) do { } while (0)
#define PROF_SEC_FLUSH(base) do { } while (0)
#define PROF_SEC_FLUSH8(base) do { } while (0)
#define PROF_SEC_ADD(i, v) do { } while (0)
#define PROF_DECL
#define PROF_WDECL
#define PROF_W0() do { } while (0)
#define PROF_W(slot) do { } while (0)
#define PROF_START() do { } while (0)
#define PROF_MARK(slot) do { } while (0)
#define LDA_PROF_DEFINE_READER(name)
#endif

#endif /* LDA_DEVICE_COMMON_H */
