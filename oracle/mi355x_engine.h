/*
 * mi355x_engine.h - a third engine for the reference's benchmark harness
 * (programs/benchmark.c: `struct engine` :51-64, registry :300-303).
 * TEST / MEASUREMENT INFRASTRUCTURE: included by the patched COPY of
 * benchmark.c that oracle/Makefile builds under oracle/_ref/reftests/
 * (oracle/benchmark_mi355x.patch; the reference tree is never touched).
 *
 * The harness itself links the REFERENCE library (its "libdeflate" engine),
 * and libdeflate_amd.so exports the same 21 symbols, so this engine reaches
 * the GPU library through dlopen(RTLD_LOCAL) + dlsym: both implementations
 * live in one process and `-C mi355x -D libdeflate` / `-C libdeflate -D
 * mi355x` cross-check each other chunk by chunk.  The library is looked up in
 * $LIBDEFLATE_AMD_LIB, then next to the executable's usual place
 * (../../../libdeflate_amd/libdeflate_amd.so), then by its plain name.
 */
#include <dlfcn.h>

static struct {
	void *h;
	void *(*alloc_c)(int);
	void (*free_c)(void *);
	void *(*alloc_d)(void);
	void (*free_d)(void *);
	size_t (*bound[3])(void *, size_t);
	size_t (*compress[3])(void *, const void *, size_t, void *, size_t);
	int (*decompress[3])(void *, const void *, size_t, void *, size_t, size_t *);
	int (*compress_batch_host)(void *, int, size_t, const void *const *,
				   const size_t *, void *const *, const size_t *,
				   size_t *);
	int (*decompress_batch_host)(void *, int, size_t, const void *const *,
				     const size_t *, void *const *, const size_t *,
				     int32_t *, size_t *, size_t *);
	const char *(*last_error)(void);
} amd;

static bool
mi355x_load(void)
{
	static const char *const fmt[3] = { "deflate", "zlib", "gzip" };
	char name[128], path[4096];
	const char *env = getenv("LIBDEFLATE_AMD_LIB");
	int i;

	if (amd.h)
		return true;
	if (env)
		amd.h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
	if (!amd.h) {
		ssize_t n = readlink("/proc/self/exe", path, sizeof(path) - 64);
		if (n > 0) {
			path[n] = 0;
			char *slash = strrchr(path, '/');
			if (slash) {
				strcpy(slash, "/../../../libdeflate_amd/libdeflate_amd.so");
				amd.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
			}
		}
	}
	if (!amd.h)
		amd.h = dlopen("libdeflate_amd.so", RTLD_NOW | RTLD_LOCAL);
	if (!amd.h) {
		msg("mi355x engine: cannot load libdeflate_amd.so (%s)", dlerror());
		return false;
	}
#define SYM(field, sym) do { *(void **)&amd.field = dlsym(amd.h, sym);        \
		if (!amd.field) { msg("mi355x engine: no %s", sym); return false; } } while (0)
	SYM(alloc_c, "libdeflate_alloc_compressor");
	SYM(free_c, "libdeflate_free_compressor");
	SYM(alloc_d, "libdeflate_alloc_decompressor");
	SYM(free_d, "libdeflate_free_decompressor");
	for (i = 0; i < 3; i++) {
		snprintf(name, sizeof(name), "libdeflate_%s_compress_bound", fmt[i]);
		SYM(bound[i], name);
		snprintf(name, sizeof(name), "libdeflate_%s_compress", fmt[i]);
		SYM(compress[i], name);
		snprintf(name, sizeof(name), "libdeflate_%s_decompress", fmt[i]);
		SYM(decompress[i], name);
	}
	SYM(compress_batch_host, "libdeflate_amd_compress_batch_host");
	SYM(decompress_batch_host, "libdeflate_amd_decompress_batch_host");
	SYM(last_error, "libdeflate_amd_last_error");
#undef SYM
	return true;
}

/* enum format of the harness -> enum libdeflate_amd_format / table index */
static int
mi355x_fmt(enum format f)
{
	return f == ZLIB_FORMAT ? 1 : f == GZIP_FORMAT ? 2 : 0;
}

static bool
mi355x_engine_init_compressor(struct compressor *c)
{
	if (!mi355x_load())
		return false;
	c->private = amd.alloc_c(c->level);
	if (!c->private)
		msg("mi355x engine: %s", amd.last_error());
	return c->private != NULL;
}

static size_t
mi355x_engine_compress_bound(struct compressor *c, size_t in_nbytes)
{
	return amd.bound[mi355x_fmt(c->format)](c->private, in_nbytes);
}

static size_t
mi355x_engine_compress(struct compressor *c, const void *in, size_t in_nbytes,
		       void *out, size_t out_nbytes_avail)
{
	return amd.compress[mi355x_fmt(c->format)](c->private, in, in_nbytes,
						   out, out_nbytes_avail);
}

static void
mi355x_engine_destroy_compressor(struct compressor *c)
{
	if (amd.h)
		amd.free_c(c->private);
}

static bool
mi355x_engine_init_decompressor(struct decompressor *d)
{
	if (!mi355x_load())
		return false;
	d->private = amd.alloc_d();
	if (!d->private)
		msg("mi355x engine: %s", amd.last_error());
	return d->private != NULL;
}

static bool
mi355x_engine_decompress(struct decompressor *d, const void *in,
			 size_t in_nbytes, void *out, size_t out_nbytes)
{
	return !amd.decompress[mi355x_fmt(d->format)](d->private, in, in_nbytes,
						      out, out_nbytes, NULL);
}

static void
mi355x_engine_destroy_decompressor(struct decompressor *d)
{
	if (amd.h)
		amd.free_d(d->private);
}

static bool
mi355x_engine_compress_batch(struct compressor *c, size_t n,
			     const void *const *in, const size_t *in_nbytes,
			     void *const *out, const size_t *out_avail,
			     size_t *out_nbytes)
{
	int rc = amd.compress_batch_host(c->private, mi355x_fmt(c->format), n, in,
					 in_nbytes, out, out_avail, out_nbytes);
	if (rc)
		msg("mi355x engine: compress_batch_host: %d (%s)", rc, amd.last_error());
	return rc == 0;
}

static bool
mi355x_engine_decompress_batch(struct decompressor *d, size_t n,
			       const void *const *in, const size_t *in_nbytes,
			       void *const *out, const size_t *out_nbytes,
			       int *results)
{
	/* exact-fill mode (actual_out = NULL), like the harness's own calls */
	int32_t *res = xmalloc(n * sizeof(*res));
	size_t *ain = xmalloc(n * sizeof(*ain));
	size_t i;
	int rc;

	if (!res || !ain)
		return false;
	rc = amd.decompress_batch_host(d->private, mi355x_fmt(d->format), n, in,
				       in_nbytes, out, out_nbytes, res, ain, NULL);
	if (rc)
		msg("mi355x engine: decompress_batch_host: %d (%s)", rc, amd.last_error());
	for (i = 0; i < n; i++)
		results[i] = res[i];
	free(res);
	free(ain);
	return rc == 0;
}

static const struct engine mi355x_engine = {
	.name			= T("mi355x"),

	.init_compressor	= mi355x_engine_init_compressor,
	.compress_bound		= mi355x_engine_compress_bound,
	.compress		= mi355x_engine_compress,
	.destroy_compressor	= mi355x_engine_destroy_compressor,

	.init_decompressor	= mi355x_engine_init_decompressor,
	.decompress		= mi355x_engine_decompress,
	.destroy_decompressor	= mi355x_engine_destroy_decompressor,

	.compress_batch		= mi355x_engine_compress_batch,
	.decompress_batch	= mi355x_engine_decompress_batch,
};
