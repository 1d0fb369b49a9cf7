/*
 * deflate_small.hip - the compress kernel for SMALL buffers (filesystem blocks,
 * BASELINE configs[4]: 4 KiB): the same tile pipeline as deflate_kernel.hip,
 * compiled a second time with a 256-thread workgroup and the LDS state of ONE
 * tile (a buffer of at most 4096 bytes is one tile; window = the whole
 * buffer), 52 KiB instead of 159: three workgroups share a CU, and the
 * phases of one buffer that keep a single wave or a single lane busy - the
 * chain insertion of its only tile, the parse, the Huffman merge - run beside
 * the wide phases of the two others.  A 1024-thread workgroup per 4 KiB
 * buffer spent 112 us per buffer and CU that way.  Levels 0-9; the caller
 * states the size bound (libdeflate_amd_compress_batch_bounded).
 */
#define LDA_SMALL 1
#define NT 256
#define TILE 4096
#define RING 4096u
#define HASH_BITS 11
#define HASH3_BITS 11
#define WQ_CAP 1024u
#include "deflate_kernel.hip"
