/* placeholder replaced below */
#include "oracle.h"
size_t oracle_deflate_compress_bound(size_t n){ size_t b=(n+4999)/5000; if(b<1)b=1; return 5*b+n; }
size_t oracle_zlib_compress_bound(size_t n){ return 6+oracle_deflate_compress_bound(n); }
size_t oracle_gzip_compress_bound(size_t n){ return 18+oracle_deflate_compress_bound(n); }
size_t oracle_deflate_compress(int l,const void*i,size_t n,void*o,size_t a){(void)l;(void)i;(void)n;(void)o;(void)a;return 0;}
size_t oracle_zlib_compress(int l,const void*i,size_t n,void*o,size_t a){(void)l;(void)i;(void)n;(void)o;(void)a;return 0;}
size_t oracle_gzip_compress(int l,const void*i,size_t n,void*o,size_t a){(void)l;(void)i;(void)n;(void)o;(void)a;return 0;}
