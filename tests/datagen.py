"""Seeded synthetic chunk generators.  Shared by tests and
bench.py so that parity and timing run on the same kind of bytes."""
import numpy as np

_VOCAB = {}


def _vocab(seed=0x0E11):
    """8192 words (2-12 letters, skewed letter frequencies) + 64 tag names,
    as a padded [N, 16] uint8 table with lengths."""
    if seed not in _VOCAB:
        rng = np.random.default_rng(seed)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        p = 1.0 / np.arange(1, 27) ** 0.9
        p /= p.sum()
        nw = 8192
        lens = rng.integers(2, 13, size=nw)
        tab = rng.choice(letters, size=(nw, 16), p=p)
        # 128 markup tokens: "<tag>" and "</tag>"
        tags = []
        for i in range(64):
            w = bytes(tab[i, :lens[i]])
            tags += [b"<" + w[:6] + b">", b"</" + w[:6] + b">"]
        ttab = np.zeros((128, 16), dtype=np.uint8)
        tlen = np.zeros(128, dtype=np.int64)
        for i, t in enumerate(tags):
            ttab[i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
            tlen[i] = len(t)
        _VOCAB[seed] = (np.vstack([tab, ttab]), np.concatenate([lens, tlen]), nw)
    return _VOCAB[seed]


def text_chunk(n, seed):
    """enwik-style text: Zipf(1.2) word choice (zlib -6 ratio ~0.33) over an 8192-word vocabulary,
    XML-ish tags every ~30 words, a newline every ~12 words.  Vectorised."""
    if n == 0:
        return b""
    rng = np.random.default_rng(seed)
    tab, lens, nw = _vocab()
    m = n // 3 + 16                      # words are >= 2 letters + separator
    idx = (rng.zipf(1.2, size=m) - 1) % nw
    tagpos = rng.random(m) < 1.0 / 30
    idx = np.where(tagpos, nw + rng.integers(0, 128, size=m), idx)
    sep = np.where(rng.random(m) < 1.0 / 12, 10, 32).astype(np.uint8)
    rows = tab[idx].copy()
    wl = lens[idx]
    rows[np.arange(m), wl] = sep
    mask = np.arange(16)[None, :] <= wl[:, None]
    return rows[mask].tobytes()[:n]


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
