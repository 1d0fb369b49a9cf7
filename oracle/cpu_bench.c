/*
 * cpu_bench.c - times the REAL reference (oracle/_ref/libdeflate_ref.so) on
 * the host cores of the box.  TEST / MEASUREMENT INFRASTRUCTURE ONLY: this is
 * bench.py's cpu_baseline leg, never part of the product.
 *
 * Convention of SURVEY.md 8(d) "CPU baseline": one compressor + one
 * decompressor per thread (libdeflate.h:56-57), chunks statically partitioned
 * over T threads, wall clock by clock_gettime(CLOCK_MONOTONIC)
 * (programs/test_util.c:143-164), best of `passes` after one warm-up pass,
 * output buffers sized by *_compress_bound, MB = 1e6 bytes
 * (programs/test_util.c:197-200).
 *
 *   cpu_bench FILE CHUNK COUNT FMT LEVEL THREADS PASSES [MODE]
 *     FILE   chunks of CHUNK bytes, concatenated; COUNT may exceed what the
 *            file holds, chunk i is then file chunk i mod (file chunks)
 *     FMT    deflate | zlib | gzip
 *     MODE   rt (compress then decompress, default) | c | d
 * prints one JSON object.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "libdeflate.h"

struct job {
	const uint8_t *data;
	size_t chunk, lo, hi, total, fchunks;
	int fmt, level;
	uint8_t *comp;		/* hi - lo slots of `bound` bytes */
	size_t *csize;
	size_t bound;
	uint8_t *back;
	double t_comp, t_dec;	/* seconds of this pass */
	size_t cbytes;
	int failed;
	int mode;		/* 0 rt, 1 compress only, 2 decompress only */
	struct libdeflate_compressor *c;
	struct libdeflate_decompressor *d;
};

static pthread_barrier_t bar;

static double now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static size_t do_compress(struct job *j, const void *in, size_t n, void *out)
{
	switch (j->fmt) {
	case 0: return libdeflate_deflate_compress(j->c, in, n, out, j->bound);
	case 1: return libdeflate_zlib_compress(j->c, in, n, out, j->bound);
	default: return libdeflate_gzip_compress(j->c, in, n, out, j->bound);
	}
}

static int do_decompress(struct job *j, const void *in, size_t n, void *out,
			 size_t avail)
{
	/* exact-fill mode, as config 4 asks (actual_out_nbytes_ret = NULL) */
	switch (j->fmt) {
	case 0: return libdeflate_deflate_decompress(j->d, in, n, out, avail, NULL);
	case 1: return libdeflate_zlib_decompress(j->d, in, n, out, avail, NULL);
	default: return libdeflate_gzip_decompress(j->d, in, n, out, avail, NULL);
	}
}

static void *worker(void *arg)
{
	struct job *j = arg;
	int pass, passes = j->failed;	/* passes smuggled in; reset below */

	j->failed = 0;
	for (pass = 0; pass <= passes; pass++) {
		double t0, t1, t2;
		size_t i, cb = 0;

		pthread_barrier_wait(&bar);
		t0 = now();
		if (j->mode != 2 || pass == 0) {
			for (i = j->lo; i < j->hi; i++) {
				size_t off = (i % j->fchunks) * j->chunk;
				size_t n = j->chunk;
				size_t k = do_compress(j, j->data + off, n,
						       j->comp + (i - j->lo) * j->bound);
				if (!k)
					j->failed++;
				j->csize[i - j->lo] = k;
				cb += k;
			}
			j->cbytes = cb;
		}
		t1 = now();
		if (j->mode != 1) {
			for (i = j->lo; i < j->hi; i++) {
				size_t off = (i % j->fchunks) * j->chunk;
				size_t n = j->chunk;
				if (do_decompress(j, j->comp + (i - j->lo) * j->bound,
						  j->csize[i - j->lo], j->back, n))
					j->failed++;
				/* the bytes are compared in the warm-up pass only:
				 * the timed passes time the decompressor alone */
				else if (pass == 0 && memcmp(j->back, j->data + off, n))
					j->failed++;
			}
		}
		t2 = now();
		pthread_barrier_wait(&bar);
		j->t_comp = t1 - t0;
		j->t_dec = t2 - t1;
		pthread_barrier_wait(&bar);	/* main reads between these two */
	}
	return NULL;
}

int main(int argc, char **argv)
{
	if (argc < 8) {
		fprintf(stderr, "usage: cpu_bench FILE CHUNK COUNT FMT LEVEL THREADS PASSES [rt|c|d]\n");
		return 2;
	}
	const size_t chunk = strtoull(argv[2], NULL, 10);
	const size_t count = strtoull(argv[3], NULL, 10);
	const int fmt = !strcmp(argv[4], "deflate") ? 0 : !strcmp(argv[4], "zlib") ? 1 : 2;
	const int level = atoi(argv[5]);
	int T = atoi(argv[6]);
	const int passes = atoi(argv[7]);
	const int mode = argc > 8 ? (!strcmp(argv[8], "c") ? 1 : !strcmp(argv[8], "d") ? 2 : 0) : 0;
	FILE *f = fopen(argv[1], "rb");
	size_t total, i;
	uint8_t *data;

	if (!f || !chunk || !count || T < 1 || passes < 1)
		return 2;
	if ((size_t)T > count)
		T = (int)count;
	fseek(f, 0, SEEK_END);
	total = (size_t)ftell(f) / chunk * chunk;	/* whole chunks only */
	fseek(f, 0, SEEK_SET);
	data = malloc(total + 1);
	if (!total || fread(data, 1, total, f) != total) {
		fprintf(stderr, "cpu_bench: %s holds no whole chunk\n", argv[1]);
		return 2;
	}
	fclose(f);

	struct job *jobs = calloc(T, sizeof(*jobs));
	pthread_t *th = calloc(T, sizeof(*th));
	pthread_barrier_init(&bar, NULL, T + 1);
	for (i = 0; i < (size_t)T; i++) {
		struct job *j = &jobs[i];

		j->data = data;
		j->chunk = chunk;
		j->total = total;
		j->fchunks = total / chunk;
		j->lo = i * count / T;
		j->hi = (i + 1) * count / T;
		j->fmt = fmt;
		j->level = level;
		j->mode = mode;
		j->c = libdeflate_alloc_compressor(level);
		j->d = libdeflate_alloc_decompressor();
		j->bound = fmt == 0 ? libdeflate_deflate_compress_bound(j->c, chunk) :
			   fmt == 1 ? libdeflate_zlib_compress_bound(j->c, chunk) :
				      libdeflate_gzip_compress_bound(j->c, chunk);
		j->comp = malloc((j->hi - j->lo) * j->bound + 1);
		j->csize = calloc(j->hi - j->lo + 1, sizeof(size_t));
		j->back = malloc(chunk + 1);
		j->failed = passes;
		if (!j->c || !j->d || !j->comp || !j->back)
			return 3;
		pthread_create(&th[i], NULL, worker, j);
	}
	double best_rt = 0, best_c = 0, best_d = 0;
	size_t cbytes = 0;
	int failed = 0, pass;
	for (pass = 0; pass <= passes; pass++) {
		double t0, t1, mc = 0, md = 0;

		pthread_barrier_wait(&bar);
		t0 = now();
		pthread_barrier_wait(&bar);
		t1 = now();
		for (i = 0; i < (size_t)T; i++) {
			if (jobs[i].t_comp > mc)
				mc = jobs[i].t_comp;
			if (jobs[i].t_dec > md)
				md = jobs[i].t_dec;
		}
		if (pass) {	/* pass 0 is the warm-up */
			if (!best_rt || t1 - t0 < best_rt)
				best_rt = t1 - t0;
			if (!best_c || mc < best_c)
				best_c = mc;
			if (!best_d || md < best_d)
				best_d = md;
		}
		pthread_barrier_wait(&bar);
	}
	for (i = 0; i < (size_t)T; i++) {
		pthread_join(th[i], NULL);
		failed += jobs[i].failed;
		cbytes += jobs[i].cbytes;
	}
	printf("{\"threads\": %d, \"chunks\": %zu, \"chunk_bytes\": %zu, \"bytes\": %zu, "
	       "\"compressed_bytes\": %zu, \"passes\": %d, \"failed\": %d, "
	       "\"wall_s\": %.6f, \"compress_s\": %.6f, \"decompress_s\": %.6f, "
	       "\"MBps\": %.1f, \"compress_MBps\": %.1f, \"decompress_MBps\": %.1f}\n",
	       T, count, chunk, count * chunk, cbytes, passes, failed, best_rt, best_c, best_d,
	       count * chunk / best_rt / 1e6, best_c > 0 ? count * chunk / best_c / 1e6 : 0.0,
	       best_d > 0 ? count * chunk / best_d / 1e6 : 0.0);
	return failed ? 1 : 0;
}
