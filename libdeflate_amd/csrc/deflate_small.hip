/*
 * deflate_small.hip - the compress kernel for SMALL buffers (filesystem blocks,
 * BASELINE configs[4]: 4 KiB): the same tile pipeline as deflate_kernel.hip,
 * compiled a second time with a 256-thread workgroup, a 4 KiB ring (window =
 * the whole buffer) and tiles of 2048 positions: 38.6 KiB of LDS instead of
 * 159 and 128 VGPRs, so FOUR workgroups share a CU, and the phases of one
 * buffer that keep a single wave or a single lane busy - the chain insertion,
 * the parses, what is left of the Huffman merge - run beside the wide phases
 * of the three others.  A 1024-thread workgroup per 4 KiB buffer spent 112 us
 * per buffer and CU that way.  The tile size is what sets the LDS (M[] and
 * the round-B lists / next tile's results are per tile): one tile of 4096
 * (52.9 KiB, three workgroups per CU) is 12 % slower per batch although a
 * buffer then has no second tile's barriers and hand-overs to pay for
 * (measured, 262 144 x 4 KiB: 54.8 ms against 48.3; two tiles at three
 * workgroups per CU: 61.4).  Levels 0-9; the caller states the size bound
 * (libdeflate_amd_compress_batch_bounded).
 */
#define LDA_SMALL 1
#define NT 256
#ifndef SMALL_TILE
#define SMALL_TILE 2048
#endif
#ifndef SMALL_HASH3_BITS
#define SMALL_HASH3_BITS 10
#endif
#ifndef SMALL_WGS
#define SMALL_WGS 4	/* workgroups per CU (LDS and registers are checked against it) */
#endif
#define TILE SMALL_TILE
#define RING 4096u
#ifndef SMALL_HASH_BITS
#define SMALL_HASH_BITS 11
#endif
#define HASH_BITS SMALL_HASH_BITS
#define HASH3_BITS SMALL_HASH3_BITS
#define WQ_CAP 1024u
#include "deflate_kernel.hip"
