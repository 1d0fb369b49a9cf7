/*
 * oracle_checksum.c - CRC-32 and Adler-32 restated in their simplest form.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "oracle.h"

/*
 * CRC-32, reflected polynomial 0xEDB88320, ~crc on entry and exit, NULL -> 0.
 * Restates: lib/crc32.c:256-262 (API, inversion, NULL rule) and the bytewise
 * definition lib/crc32.c:211-219 / scripts/gen-crc32-consts.py:49-63.
 * Done one bit at a time on purpose: no tables to get wrong.
 */
uint32_t oracle_crc32(uint32_t crc, const void *buf, size_t len)
{
	const uint8_t *p = (const uint8_t *)buf;
	size_t i;
	int k;

	if (p == NULL)
		return 0;
	crc = ~crc;
	for (i = 0; i < len; i++) {
		crc ^= p[i];
		for (k = 0; k < 8; k++)
			crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
	}
	return ~crc;
}

/*
 * Adler-32: s1 = low half, s2 = high half, both mod 65521; NULL -> 1.
 * Restates lib/adler32.c:156-162 (API) and :105-119 (generic loop).  The
 * reference reduces every <= 5552 bytes (lib/adler32.c:54); reducing after
 * every byte with 64-bit sums gives the same residues, including for the
 * "unreduced-looking" initial halves up to 65535 that
 * programs/test_checksums.c:49-60 feeds in.
 */
uint32_t oracle_adler32(uint32_t adler, const void *buf, size_t len)
{
	const uint8_t *p = (const uint8_t *)buf;
	uint64_t s1 = adler & 0xFFFF, s2 = adler >> 16;
	size_t i;

	if (p == NULL)
		return 1;
	for (i = 0; i < len; i++) {
		s1 = (s1 + p[i]) % 65521;
		s2 = (s2 + s1) % 65521;
	}
	/*
	 * With len == 0 the reference returns (s2 << 16) | s1 without reducing
	 * (its loop body never runs), so do not reduce here either.
	 */
	return (uint32_t)((s2 << 16) | s1);
}
