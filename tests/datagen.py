"""Seeded synthetic chunk generators.  Shared by tests and
bench.py so that parity and timing run on the same kind of bytes."""
import numpy as np

_WORDS_CACHE = {}


def _vocab(seed=0x0E11):
    if seed not in _WORDS_CACHE:
        rng = np.random.default_rng(seed)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        p = 1.0 / np.arange(1, 27) ** 0.9
        p /= p.sum()
        words = []
        for _ in range(8192):
            n = int(rng.integers(2, 13))
            words.append(bytes(rng.choice(letters, size=n, p=p)))
        _WORDS_CACHE[seed] = words
    return _WORDS_CACHE[seed]


def text_chunk(n, seed):
    """enwik-style text: Zipf word choice, XML-ish tags, newlines."""
    rng = np.random.default_rng(seed)
    words = _vocab()
    ranks = rng.zipf(1.1, size=n // 3 + 16)
    ranks = (ranks - 1) % len(words)
    out = bytearray()
    col = 0
    i = 0
    next_tag = int(rng.integers(100, 300))
    while len(out) < n:
        w = words[int(ranks[i])]
        i += 1
        out += w
        col += len(w) + 1
        if len(out) >= next_tag:
            t = words[int(ranks[i]) % 64]
            out += b" <" + t + b">" + words[int(ranks[i + 1])] + b"</" + t + b">"
            i += 2
            next_tag = len(out) + int(rng.integers(100, 300))
        if col > 72:
            out += b"\n"
            col = 0
        else:
            out += b" "
    return bytes(out[:n])


def binary_chunk(n, seed):
    """little-endian u32 counters + small-range noise."""
    rng = np.random.default_rng(seed)
    m = n // 4 + 1
    base = np.arange(m, dtype=np.uint32) * np.uint32(rng.integers(1, 9))
    noise = rng.integers(0, 4, size=m, dtype=np.uint32)
    v = (base + noise + np.uint32(rng.integers(0, 1 << 20))).astype("<u4")
    return v.tobytes()[:n]


def lowentropy_chunk(n, seed):
    """16-symbol alphabet, geometric distribution."""
    rng = np.random.default_rng(seed)
    g = np.minimum(rng.geometric(0.35, size=n) - 1, 15).astype(np.uint8)
    return (g + 0x41).tobytes()


def random_chunk(n, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()


def zero_chunk(n, seed):
    return bytes(n)


MIX64K = [text_chunk] * 5 + [binary_chunk, lowentropy_chunk, random_chunk]
MIX4K = [text_chunk] * 3 + [binary_chunk] * 2 + [zero_chunk, lowentropy_chunk,
                                                 random_chunk]


def chunk(idx, n, base_seed, mix=MIX64K):
    """chunk `idx` of a batch: kind by idx mod 8, seed = base + idx."""
    return mix[idx % 8](n, base_seed + idx)


def batch(count, n, base_seed, mix=MIX64K, distinct=None):
    """`count` chunks of n bytes as one bytes object.  `distinct` bounds how
    many different chunks are generated (the rest repeat), which keeps big
    batches cheap to build on the host."""
    distinct = count if distinct is None else min(distinct, count)
    uniq = [chunk(i, n, base_seed, mix) for i in range(distinct)]
    return [uniq[i % distinct] for i in range(count)]
