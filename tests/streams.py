"""Hand-assembled DEFLATE streams and randomized stream cases used by both the
oracle tests (CPU) and the GPU parity tests.

The hand-assembled ones rebuild, with our own bit writer, the vectors that the
reference's unit tests pin:
  programs/test_incomplete_codes.c:71-372, test_invalid_streams.c:58-121,
  test_overread.c:15-91, test_trailing_bytes.c:41-150.
"""
import random
import zlib

from tests import datagen


class BitWriter:
    """LSB-first bit packer (the job of put_bits/flush_bits in
    programs/test_util.c:210-237)."""

    def __init__(self):
        self.acc = 0
        self.n = 0
        self.out = bytearray()

    def put(self, value, nbits):
        self.acc |= (value & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def put_code(self, code, nbits):
        """Huffman codewords go MSB-first."""
        for i in range(nbits - 1, -1, -1):
            self.put((code >> i) & 1, 1)

    def finish(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
            self.acc = 0
            self.n = 0
        return bytes(self.out)


PERM = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def _dyn_header(w, final, nlit, noff, pre_lens):
    """BFINAL, BTYPE=2, HLIT/HDIST/HCLEN and the precode lengths."""
    w.put(final, 1)
    w.put(2, 2)
    w.put(nlit - 257, 5)
    w.put(noff - 1, 5)
    nexp = 19
    while nexp > 4 and pre_lens.get(PERM[nexp - 1], 0) == 0:
        nexp -= 1
    w.put(nexp - 4, 4)
    for i in range(nexp):
        w.put(pre_lens.get(PERM[i], 0), 3)


def incomplete_empty_offset_code():
    """test_incomplete_codes.c:75-147: litlen code {A:1,B:2,EOB:2 bits}, the
    offset code has no codewords at all (one length-0 entry); data "ABAA"."""
    w = BitWriter()
    # precode: sym0 -> 1 bit '0'?  use lens: 0:1? need lens {0,1,2}: three syms
    # precode lens: presym 0:2 bits, 1:2 bits, 2:2 bits, 18:2 bits (complete)
    pre = {0: 2, 1: 2, 2: 2, 18: 2}
    # canonical precode codewords (by len then symbol): 0->00 1->01 2->10 18->11
    code = {0: 0, 1: 1, 2: 2, 18: 3}
    _dyn_header(w, 1, 257, 1, pre)

    def lens_run_zeros(n):
        while n:
            r = min(n, 138)
            assert r >= 11
            w.put_code(code[18], 2)
            w.put(r - 11, 7)
            n -= r
    # litlen lens: 0..64 zero (65 syms), 'A'(65)=1, 'B'(66)=2, 67..255 zero
    # (189 syms), 256=2
    lens_run_zeros(65)
    w.put_code(code[1], 2)
    w.put_code(code[2], 2)
    n = 189
    w.put_code(code[18], 2); w.put(138 - 11, 7); n -= 138
    w.put_code(code[18], 2); w.put(n - 11, 7)
    w.put_code(code[2], 2)
    # offset lens: single 0
    w.put_code(code[0], 2)
    # litlen canonical: A:'0', B:'10', EOB:'11'
    for sym in "ABAA":
        if sym == "A":
            w.put_code(0, 1)
        else:
            w.put_code(2, 2)
    w.put_code(3, 2)
    return w.finish(), b"ABAA"


def incomplete_singleton_litlen():
    """test_incomplete_codes.c:179-210: the only litlen codeword is EOB with
    length 1 (incomplete but accepted); output is empty."""
    w = BitWriter()
    pre = {0: 1, 1: 2, 18: 2}
    code = {0: (0, 1), 1: (2, 2), 18: (3, 2)}
    _dyn_header(w, 1, 257, 1, pre)
    n = 256
    w.put_code(*code[18]); w.put(138 - 11, 7); n -= 138
    w.put_code(*code[18]); w.put(n - 11, 7)
    w.put_code(*code[1])        # EOB len 1
    w.put_code(*code[0])        # offset sym0 len 0
    w.put_code(0, 1)            # EOB
    return w.finish(), b""


def incomplete_singleton_offset(sym_nonzero):
    """test_incomplete_codes.c:217-368: the offset code has exactly one
    codeword of length 1 (symbol 0, or a non-zero symbol); the data is a
    literal followed by a match that uses it."""
    w = BitWriter()
    pre = {0: 1, 1: 2, 2: 3, 18: 3}
    # canonical: 0:'0', 1:'10', 2:'110', 18:'111'
    code = {0: (0, 1), 1: (2, 2), 2: (6, 3), 18: (7, 3)}
    noff = 2 if sym_nonzero else 1
    # litlen: 254 -> len 2? keep simple: lits 0xfe(254):2, 0xff(255):2,
    # EOB(256):2, len sym 257 (length 3): 2  -> complete (4 x 2 bits)
    _dyn_header(w, 1, 258, noff, pre)
    n = 254
    w.put_code(*code[18]); w.put(138 - 11, 7); n -= 138
    w.put_code(*code[18]); w.put(n - 11, 7)
    for _ in range(4):
        w.put_code(*code[2])
    if sym_nonzero:
        w.put_code(*code[0])    # offset sym 0: unused
        w.put_code(*code[1])    # offset sym 1: len 1
    else:
        w.put_code(*code[1])    # offset sym 0: len 1
    # litlen canonical (all len 2): 254:'00' 255:'01' 256:'10' 257:'11'
    if sym_nonzero:
        # "fe ff" then match len 3 offset 2 -> fe ff fe ff fe
        w.put_code(0, 2); w.put_code(1, 2)
        w.put_code(3, 2); w.put_code(0, 1)
        want = bytes([0xFE, 0xFF, 0xFE, 0xFF, 0xFE])
    else:
        # "ff" then match len 3 offset 1 -> ff ff ff ff
        w.put_code(1, 2)
        w.put_code(3, 2); w.put_code(0, 1)
        want = bytes([0xFF] * 4)
    w.put_code(2, 2)
    return w.finish(), want


def too_many_codeword_lengths():
    """test_invalid_streams.c:65-118: the length runs overshoot
    HLIT+HDIST -> BAD_DATA."""
    w = BitWriter()
    pre = {0: 1, 18: 1}
    _dyn_header(w, 1, 257, 1, pre)
    # canonical: 0:'0', 18:'1'; 258 lengths wanted, give 138+138 zeros = 276
    w.put_code(1, 1); w.put(127, 7)
    w.put_code(1, 1); w.put(127, 7)
    w.put(0, 16)
    return w.finish()


def overread_stream():
    """test_overread.c:15-68: litlen code where the all-zero-bits codeword is
    a literal, stream truncated right after the header: the implicit zero bits
    decode as an endless run of literals -> must be BAD_DATA."""
    w = BitWriter()
    pre = {1: 1, 18: 1}
    # canonical: 1:'0', 18:'1'
    _dyn_header(w, 1, 257, 1, pre)
    # litlen: sym 0 len 1, syms 1..255 zero, sym 256 len 1 -> '0'=lit 0,'1'=EOB
    w.put_code(0, 1)
    n = 255
    w.put_code(1, 1); w.put(138 - 11, 7); n -= 138
    w.put_code(1, 1); w.put(n - 11, 7)
    w.put_code(0, 1)
    # offset: one symbol with len 1
    w.put_code(0, 1)
    return w.finish()


def trailing_bytes_input():
    """test_trailing_bytes.c:74-75"""
    return bytes(((i % 123) + (i % 1023)) & 0xFF for i in range(32768))


def litrunlen_input():
    """test_litrunlen_overflow.c:35-40 style: 125500 bytes, no matches."""
    out = bytearray(125500)
    v = 0
    for i in range(len(out)):
        out[i] = (v >> 8) & 0xFF if i & 1 else v & 0xFF
        if i & 1:
            v = (v * 31 + 7) & 0xFFFF
    return bytes(out)


def _zcompress(fmt, level, data):
    wbits = {"deflate": -15, "zlib": 15, "gzip": 31}[fmt]
    c = zlib.compressobj(level, zlib.DEFLATED, wbits)
    return c.compress(data) + c.flush()


def random_cases(seed, count, compress=None, sizes=None):
    """Randomized (fmt, stream, out_avail, want_actual_out, tag) cases:
    valid streams from `compress(fmt, level, data)` (default: Python's zlib),
    plus truncated / bit-flipped / short-output / oversized-output / trailing
    garbage variants."""
    rng = random.Random(seed)
    compress = compress or _zcompress
    sizes = sizes or [0, 1, 5, 31, 32, 100, 1000, 5000, 20000, 70000]
    cases = []
    for it in range(count):
        n = rng.choice(sizes)
        kind = rng.randrange(5)
        if kind == 0:
            data = datagen.random_chunk(n, seed * 7919 + it)
        elif kind == 1:
            data = datagen.lowentropy_chunk(n, seed * 7919 + it)
        elif kind == 2:
            data = (b"hello world, " * (n // 13 + 1))[:n]
        elif kind == 3:
            data = datagen.binary_chunk(n, seed * 7919 + it)
        else:
            data = datagen.text_chunk(n, seed * 7919 + it)
        fmt = rng.choice(["deflate", "zlib", "gzip"])
        level = rng.choice([0, 1, 3, 6, 9])
        comp = compress(fmt, level, data)
        mode = rng.randrange(7)
        avail = n
        if mode == 1 and comp:
            comp = comp[:rng.randrange(len(comp))]
        elif mode == 2 and comp:
            b = bytearray(comp)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            comp = bytes(b)
        elif mode == 3:
            avail = max(0, n - rng.randrange(0, 300))
        elif mode == 4:
            avail = n + rng.randrange(0, 300)
        elif mode == 5:
            comp = comp + bytes(rng.randrange(256)
                                for _ in range(rng.randrange(20)))
        for want in (True, False):
            cases.append((fmt, comp, avail, want, f"it{it}/m{mode}/l{level}"))
    return cases


def garbage_cases(seed, count):
    """Streams of random bytes: exercises every header/validity failure."""
    rng = random.Random(seed)
    cases = []
    for it in range(count):
        n = rng.choice([0, 1, 2, 5, 6, 17, 18, 19, 40, 200, 2000])
        s = bytes(rng.randrange(256) for _ in range(n))
        if rng.randrange(3) == 0 and n >= 3:
            # bias towards a dynamic block header
            s = bytes([(s[0] & 0xF8) | 0x05]) + s[1:]
        fmt = rng.choice(["deflate", "deflate", "zlib", "gzip"])
        if fmt == "zlib" and n >= 2 and rng.randrange(2):
            s = b"\x78\x9c" + s[2:]
        if fmt == "gzip" and n >= 10 and rng.randrange(2):
            s = b"\x1f\x8b\x08" + bytes([rng.choice([0, 4, 8, 16, 2, 28])]) + s[4:]
        avail = rng.choice([0, 10, 1000, 70000])
        cases.append((fmt, s, avail, True, f"garbage{it}"))
    return cases


# ---- static-Huffman helpers (RFC 1951 3.2.6) ----

def _static_lit(w, sym):
    if sym < 144:
        w.put_code(0x30 + sym, 8)
    elif sym < 256:
        w.put_code(0x190 + sym - 144, 9)
    elif sym < 280:
        w.put_code(sym - 256, 7)
    else:
        w.put_code(0xC0 + sym - 280, 8)


_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
          59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LXB = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4,
        5, 5, 5, 5, 0]


def _static_match(w, length, dist):
    s = max(i for i in range(29) if _LBASE[i] <= length)
    _static_lit(w, 257 + s)
    w.put(length - _LBASE[s], _LXB[s])
    d = dist - 1
    if d < 4:
        slot, xb, xv = d, 0, 0
    else:
        hb = d.bit_length() - 1
        slot, xb = 2 * hb + ((d >> (hb - 1)) & 1), hb - 1
        xv = d & ((1 << xb) - 1)
    w.put_code(slot, 5)
    w.put(xv, xb)


def stored_then_match_streams():
    """[Huffman block][non-empty stored block][Huffman block that opens with a
    short-distance match]: the decoder must take the match's source from the
    stored bytes, not from what preceded them.  -> list of (stream, expected).
    Sizes under and over 64 input bytes, stored lengths 1..300."""
    out = []
    for head, stored, (mlen, mdist), tail in [
            (b"abcdefgh", b"XYZ", (3, 1), b""),
            (b"abcdefgh", b"XYZ", (8, 3), b"!"),
            (b"q", b"Z", (5, 1), b"end"),
            (b"hello, hello", b"0123456789" * 30, (7, 2), b"tail" * 20),
            (b"", b"AB", (6, 2), b""),
            (b"12345678", bytes(range(65, 73)), (8, 8), b"zz"),
            (b"12345678", bytes(range(65, 75)), (4, 7), b"")]:
        w = BitWriter()
        w.put(0, 1); w.put(1, 2)                 # static block, not final
        for b in head:
            _static_lit(w, b)
        _static_lit(w, 256)
        w.put(0, 1); w.put(0, 2)                 # stored block
        if w.n:
            w.put(0, 8 - w.n)
        w.put(len(stored), 16); w.put(len(stored) ^ 0xFFFF, 16)
        for b in stored:
            w.put(b, 8)
        w.put(1, 1); w.put(1, 2)                 # final static block
        _static_match(w, mlen, mdist)
        for b in tail:
            _static_lit(w, b)
        _static_lit(w, 256)
        sofar = bytearray(head + stored)
        for _ in range(mlen):
            sofar.append(sofar[-mdist])
        out.append((w.finish(), bytes(sofar) + tail))
    return out


def static_dynamic_static_streams():
    """[static][empty stored][dynamic (zlib)][empty stored][static, final]: the
    decoder keeps the static tables across the static blocks that follow one
    another (decompress_template.h:303-311) - and must build them again after
    a dynamic block has used the tables' memory.  The last block copies from
    both earlier ones.  -> list of (stream, expected)."""
    import zlib
    out = []
    for head, mid, (mlen, mdist), tail in [
            (b"static one. ", b"the quick brown fox jumps over the lazy dog. " * 40, (9, 5), b" end"),
            (b"A" * 70, bytes(range(256)) * 3 + b"abcabcabc" * 50, (200, 300), b""),
            (b"", b"0123456789" * 200, (12, 1999), b"zz" * 40)]:
        w = BitWriter()
        w.put(0, 1); w.put(1, 2)                 # static block, not final
        for b in head:
            _static_lit(w, b)
        _static_lit(w, 256)
        w.put(0, 1); w.put(0, 2)                 # empty stored block: byte boundary
        if w.n:
            w.put(0, 8 - w.n)
        w.put(0, 16); w.put(0xFFFF, 16)
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        for b in co.compress(mid) + co.flush(zlib.Z_FULL_FLUSH):
            w.put(b, 8)                          # dynamic block(s) + empty stored
        w.put(1, 1); w.put(1, 2)                 # final static block
        sofar = bytearray(head + mid)
        _static_match(w, mlen, mdist)
        for _ in range(mlen):
            sofar.append(sofar[-mdist])
        for b in tail:
            _static_lit(w, b)
        _static_lit(w, 256)
        out.append((w.finish(), bytes(sofar) + tail))
    return out


def gzip_optional_field_streams():
    """VALID gzip members that carry FEXTRA / FNAME / FCOMMENT / FHCRC in every
    combination (lib/gzip_decompress.c:69-100 skips them) -> (stream, data)."""
    import struct
    out = []
    for flg in range(0, 32, 2):        # FHCRC 2, FEXTRA 4, FNAME 8, FCOMMENT 16
        data = datagen.text_chunk(300 + 37 * flg, 0x0E1100F0 + flg)
        body = _zcompress("deflate", 6, data)
        h = bytearray(b"\x1f\x8b\x08" + bytes([flg]) + b"\x12\x34\x56\x78\x02\x03")
        if flg & 4:
            extra = bytes(range(flg + 3))
            h += struct.pack("<H", len(extra)) + extra
        if flg & 8:
            h += b"file-name.txt\x00"
        if flg & 16:
            h += b"a comment\x00"
        if flg & 2:
            h += struct.pack("<H", zlib.crc32(bytes(h)) & 0xFFFF)
        s = bytes(h) + body + struct.pack("<II", zlib.crc32(data), len(data))
        out.append((s, data))
    return out


def parallel_round_streams():
    """Long raw DEFLATE streams aimed at the wave-per-stream rounds:
    -> list of (name, stream, expected).
      huff2 / huff3   Huffman-only coding of 2- and 3-symbol data: 1- and
                      2-bit literals, more tokens per 384-bit piece than a
                      lane may record (the round is clipped or abandoned);
      periods         stretches of period p between stretches of noise, p
                      from 1 to 200: long matches whose distance is below,
                      at and above the 64-byte slot size (copies inside a
                      slot, across slots, runs);
      far             matches that reach 5..30 KiB back, past the LDS mirror;
      flushed*        (round 6) hand-built static blocks whose matches take their
                      sources from just behind the 4 KiB LDS mirror, 2.9..11 KiB
                      back in a sweep: bytes the wave stored to the output
                      earlier in the SAME round (the round's first group, the
                      group before the last wait, ...) and now loads back -
                      the case global_stores_visible() exists for."""
    import random
    import zlib
    out = []
    rng = random.Random(0x0E110031)
    for nsym in (2, 3):
        data = bytes(rng.choice(b"ab" if nsym == 2 else b"abc") for _ in range(60000))
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
        out.append((f"huff{nsym}", co.compress(data) + co.flush(), data))
    parts = []
    for p in (1, 2, 3, 5, 7, 31, 63, 64, 65, 127, 200) * 3:
        unit = bytes(rng.randrange(256) for _ in range(p))
        parts.append((unit * (2500 // p + 1))[:rng.randrange(900, 2500)])
        parts.append(bytes(rng.randrange(256) for _ in range(rng.randrange(300, 700))))
    data = b"".join(parts)
    co = zlib.compressobj(9, zlib.DEFLATED, -15)
    out.append(("periods", co.compress(data) + co.flush(), data))
    base = bytes(rng.randrange(256) for _ in range(30000))
    pieces = [base]
    for i in range(400):
        at = rng.randrange(0, 30000 - 300)
        pieces.append(base[at:at + rng.randrange(8, 200)])
        pieces.append(bytes(rng.randrange(256) for _ in range(rng.randrange(0, 40))))
    data = b"".join(pieces)
    co = zlib.compressobj(9, zlib.DEFLATED, -15)
    out.append(("far", co.compress(data) + co.flush(), data))
    for v, (lo, hi, step, lits) in enumerate(((2900, 11000, 37, 3), (3900, 4400, 1, 0),
                                              (3000, 9000, 211, 40))):
        w = BitWriter()
        w.put(1, 1)
        w.put(1, 2)                     # final, static
        data = bytearray(rng.randrange(256) for _ in range(12000))
        for b in data:
            _static_lit(w, b)
        d = lo
        while len(data) < 150000:
            length = rng.choice((3, 4, 9, 30, 64, 65, 130, 258))
            _static_match(w, length, d)
            for k in range(length):
                data.append(data[-d])
            for _ in range(rng.randrange(0, lits + 1)):
                b = rng.randrange(256)
                _static_lit(w, b)
                data.append(b)
            d = lo if d + step > hi else d + step
        _static_lit(w, 256)
        out.append((f"flushed{v}", w.finish(), bytes(data)))
    return out


def bad_distance_streams():
    """A dynamic-Huffman stream long enough for a parallel round whose first
    kilobytes hold a match reaching back before the start of the output:
    -> list of streams (all invalid).  Built by compressing with a preset
    dictionary and dropping the dictionary (the distances into it dangle)."""
    import random
    import zlib
    rng = random.Random(0x0E110032)
    out = []
    for at in (0, 700, 5000):
        words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(3, 9)))
                 for _ in range(300)]
        text = b" ".join(rng.choice(words) for _ in range(12000))
        zdict = text[20000:52768] if at == 0 else bytes(rng.randrange(256) for _ in range(32768))
        body = text[:at] + zdict[1000:1400] + text[at:]
        co = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=zdict)
        out.append(co.compress(body) + co.flush())
    return out


def _take_bytes(gen_block, nbytes):
    """The first nbytes of an endless sequence of blocks: what the reference's
    generators leave when put_bits() runs out of buffer
    (programs/test_util.c:210-228, test_slow_decompression.c:24-27)."""
    w = BitWriter()
    while len(w.out) < nbytes:
        gen_block(w)
    return bytes(w.out[:nbytes])


def empty_static_blocks(nbytes=4096):
    """test_slow_decompression.c:18-30: BFINAL=0, static Huffman, end of block,
    over and over; the stream runs into the end of its buffer."""
    def block(w):
        w.put(0, 1)
        w.put(1, 2)
        w.put(0, 7)     # litlen symbol 256
    return _take_bytes(block, nbytes)


def empty_dynamic_blocks(nbytes=4096):
    """test_slow_decompression.c:32-108: the smallest dynamic block - litlen
    code {256: 1 bit}, offset code {0: 1 bit}, precode {1: 1 bit, 18: 1 bit} -
    holding nothing but its end-of-block symbol, over and over."""
    def block(w):
        w.put(0, 1)         # BFINAL
        w.put(2, 2)         # dynamic
        w.put(0, 5)         # 257 litlen symbols
        w.put(0, 5)         # 1 offset symbol
        w.put(14, 4)        # 18 explicit precode lengths
        for _ in range(2):
            w.put(0, 3)     # presym 16, 17
        w.put(1, 3)         # presym 18: 1 bit
        for _ in range(14):
            w.put(0, 3)
        w.put(1, 3)         # presym 1: 1 bit
        for _ in range(2):
            w.put(1, 1)     # presym 18 ...
            w.put(117, 7)   # ... 128 zeros
        w.put(0, 1)         # presym 1 (litlen 256)
        w.put(0, 1)         # presym 1 (offset 0)
        w.put(0, 1)         # end of block
    return _take_bytes(block, nbytes)


def damage(stream, cut=None, flip=None):
    """A variant of a valid stream: its first `cut` bytes, or bit 4 of byte
    `flip` inverted (tests/golden/golden_64k.json stores each stream once)."""
    s = bytearray(stream)
    if flip is not None:
        s[flip] ^= 0x10
    return bytes(s if cut is None else s[:cut])


def golden_64k_cases(path):
    """(fmt, stream, avail, want, tag, expected-dict) per variant of
    tests/golden/golden_64k.json."""
    import base64
    import json
    with open(path) as f:
        g = json.load(f)
    for st in g["streams"]:
        z = base64.b64decode(st["stream_b64"])
        for v in st["variants"]:
            yield (st["fmt"], damage(z, v["cut"], v["flip"]), v["avail"],
                   v["want_out"], f'{st["tag"]}/{v["name"]}', v)
