/*
 * mi355x_batch.h - batch mode (-B) for the patched copy of the reference's
 * benchmark harness: the per-chunk loop of programs/benchmark.c:430-538 with
 * the chunks of a file handed to the engine TOGETHER where it has a batch
 * entry (mi355x), chunk by chunk otherwise (libdeflate, libz), so any pair of
 * engines can be crossed.  Same checks (every chunk decompresses to the
 * original), same report lines.  TEST / MEASUREMENT INFRASTRUCTURE.
 */
static int
do_benchmark_batch(struct file_stream *in, u32 chunk_size, bool allow_expansion,
		   struct compressor *compressor,
		   struct decompressor *decompressor)
{
	size_t cap = 0, n = 0, i;
	u8 **orig = NULL;
	size_t *orig_n = NULL;
	u64 total_uncompressed_size = 0, total_compressed_size = 0;
	u64 start_time, compress_time, decompress_time;
	int ret = -1;

	for (;;) {
		u8 *buf = xmalloc(chunk_size);
		ssize_t r;

		if (!buf)
			return -1;
		r = xread(in, buf, chunk_size);
		if (r <= 0) {
			free(buf);
			if (r < 0)
				return r;
			break;
		}
		if (n == cap) {
			cap = cap ? 2 * cap : 64;
			orig = realloc(orig, cap * sizeof(*orig));
			orig_n = realloc(orig_n, cap * sizeof(*orig_n));
			if (!orig || !orig_n)
				return -1;
		}
		orig[n] = buf;
		orig_n[n++] = r;
		total_uncompressed_size += r;
	}
	if (n == 0) {
		printf("\tFile was empty.\n");
		return 0;
	}
	void **comp = xmalloc(n * sizeof(*comp));
	void **back = xmalloc(n * sizeof(*back));
	size_t *avail = xmalloc(n * sizeof(*avail));
	size_t *comp_n = xmalloc(n * sizeof(*comp_n));
	int *results = xmalloc(n * sizeof(*results));
	if (!comp || !back || !avail || !comp_n || !results)
		return -1;
	for (i = 0; i < n; i++) {
		avail[i] = allow_expansion ? compress_bound(compressor, orig_n[i]) :
					     orig_n[i] - 1;
		comp[i] = xmalloc(avail[i] + 1);
		back[i] = xmalloc(orig_n[i]);
		if (!comp[i] || !back[i])
			return -1;
	}

	start_time = timer_ticks();
	if (compressor->engine->compress_batch) {
		if (!compressor->engine->compress_batch(compressor, n,
				(const void *const *)orig, orig_n, comp, avail, comp_n))
			goto out;
	} else {
		for (i = 0; i < n; i++)
			comp_n[i] = do_compress(compressor, orig[i], orig_n[i],
						comp[i], avail[i]);
	}
	compress_time = timer_ticks() - start_time;

	/* chunks that did not fit stay uncompressed, like the per-chunk loop */
	size_t m = 0;
	const void **din = xmalloc(n * sizeof(*din));
	size_t *din_n = xmalloc(n * sizeof(*din_n));
	void **dout = xmalloc(n * sizeof(*dout));
	size_t *dout_n = xmalloc(n * sizeof(*dout_n));
	size_t *idx = xmalloc(n * sizeof(*idx));
	if (!din || !din_n || !dout || !dout_n || !idx)
		return -1;
	for (i = 0; i < n; i++) {
		if (comp_n[i] == 0) {
			if (allow_expansion) {
				msg("%"TS": bug in compress_bound()", in->name);
				goto out;
			}
			total_compressed_size += orig_n[i];
			continue;
		}
		total_compressed_size += comp_n[i];
		din[m] = comp[i];
		din_n[m] = comp_n[i];
		dout[m] = back[i];
		dout_n[m] = orig_n[i];
		idx[m++] = i;
	}
	start_time = timer_ticks();
	if (decompressor->engine->decompress_batch) {
		if (m && !decompressor->engine->decompress_batch(decompressor, m, din,
				din_n, dout, dout_n, results))
			goto out;
	} else {
		for (i = 0; i < m; i++)
			results[i] = !do_decompress(decompressor, din[i], din_n[i],
						    dout[i], dout_n[i]);
	}
	decompress_time = timer_ticks() - start_time;
	for (i = 0; i < m; i++) {
		if (results[i]) {
			msg("%"TS": failed to decompress data (chunk %zu)", in->name, idx[i]);
			goto out;
		}
		if (memcmp(orig[idx[i]], dout[i], dout_n[i]) != 0) {
			msg("%"TS": data did not decompress to original (chunk %zu)",
			    in->name, idx[i]);
			goto out;
		}
	}
	if (compress_time == 0)
		compress_time = 1;
	if (decompress_time == 0)
		decompress_time = 1;
	printf("\tBatch mode: %zu chunks per engine call\n", n);
	printf("\tCompressed %"PRIu64 " => %"PRIu64" bytes (%u.%03u%%)\n",
	       total_uncompressed_size, total_compressed_size,
	       (unsigned int)(total_compressed_size * 100 / total_uncompressed_size),
	       (unsigned int)(total_compressed_size * 100000 /
				total_uncompressed_size % 1000));
	printf("\tCompression time: %"PRIu64" ms (%"PRIu64" MB/s)\n",
	       timer_ticks_to_ms(compress_time),
	       timer_MB_per_s(total_uncompressed_size, compress_time));
	printf("\tDecompression time: %"PRIu64" ms (%"PRIu64" MB/s)\n",
	       timer_ticks_to_ms(decompress_time),
	       timer_MB_per_s(total_uncompressed_size, decompress_time));
	ret = 0;
out:
	for (i = 0; i < n; i++) {
		free(orig[i]);
		free(comp[i]);
		free(back[i]);
	}
	free(orig); free(orig_n); free(comp); free(back); free(avail);
	free(comp_n); free(results); free(din); free(din_n); free(dout);
	free(dout_n); free(idx);
	return ret;
}
