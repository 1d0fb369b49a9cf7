"""CPU model of the wave's piece-parallel parse in inflate_kernel.hip (design
aid, not product code): how many sync passes / wave steps does a round need?
  python tools/models/sim_par.py [chunk kind 0-7]"""
import os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import datagen

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LXB   = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DXB   = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]

class Bits:
    def __init__(s, data): s.d = data; s.n = len(data) * 8
    def get(s, pos, k):
        v = 0
        for i in range(k):
            p = pos + i
            b = (s.d[p >> 3] >> (p & 7)) & 1 if p < s.n else 0
            v |= b << i
        return v

def build(lens):
    # canonical code -> dict (len, code) -> sym
    cnt = [0] * 16
    for l in lens: cnt[l] += 1
    cnt[0] = 0
    nxt = [0] * 16; code = 0
    for l in range(1, 16):
        code = (code + cnt[l - 1]) << 1; nxt[l] = code
    tab = {}
    for s, l in enumerate(lens):
        if l:
            tab[(l, nxt[l])] = s; nxt[l] += 1
    return tab

def decode(bits, pos, tab):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | bits.get(pos + l - 1, 1)
        if (l, code) in tab: return tab[(l, code)], l
    return 256, 15  # garbage

def token(bits, pos, lit, off):
    sym, l = decode(bits, pos, lit)
    pos += l
    if sym < 256: return pos, 0
    if sym == 256: return pos, 1
    s = min(sym - 257, 28)
    pos += LXB[s]
    d, l = decode(bits, pos, off)
    pos += l + DXB[min(d, 29)]
    return pos, 0

def blocks(data):
    """yield (first token bit, lit tab, off tab, end bit after EOB) for dynamic blocks"""
    bits = Bits(data); pos = 0
    while True:
        final = bits.get(pos, 1); typ = bits.get(pos + 1, 2); pos += 3
        if typ != 2:
            return    # stored / static block: not modelled
        nlit = 257 + bits.get(pos, 5); noff = 1 + bits.get(pos + 5, 5); npre = 4 + bits.get(pos + 10, 4); pos += 14
        perm = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
        pl = [0] * 19
        for i in range(npre): pl[perm[i]] = bits.get(pos, 3); pos += 3
        pt = build(pl); lens = []
        while len(lens) < nlit + noff:
            s, l = decode(bits, pos, pt); pos += l
            if s < 16: lens.append(s)
            elif s == 16: r = 3 + bits.get(pos, 2); pos += 2; lens += [lens[-1]] * r
            elif s == 17: r = 3 + bits.get(pos, 3); pos += 3; lens += [0] * r
            else: r = 11 + bits.get(pos, 7); pos += 7; lens += [0] * r
        lit = build(lens[:nlit]); off = build(lens[nlit:])
        start = pos
        while True:
            pos, e = token(bits, pos, lit, off)
            if e: break
        yield bits, start, lit, off, pos
        if final: return

def sim_round(bits, bpos0, lit, off, CB=384, NL=64):
    start = [bpos0 + l * CB for l in range(NL)]
    end = [0] * NL; ntok = [0] * NL; eob = [False] * NL
    dirty = [True] * NL
    passes = 0; iters = 0; log = []
    while True:
        passes += 1
        mx = 0
        for l in range(NL):
            if not dirty[l]: continue
            pos = start[l]; cend = bpos0 + (l + 1) * CB; n = 0; eob[l] = False; it = 0
            while pos < cend:
                pos, e = token(bits, pos, lit, off); it += 1
                if e: eob[l] = True; break
                n += 1
            it += 0 if eob[l] else 1   # the iteration that notices pos >= cend
            end[l] = pos; ntok[l] = n; mx = max(mx, it)
        iters += mx
        log.append((sum(dirty), mx))
        ns = [bpos0] + end[:-1]
        dirty = [ns[l] != start[l] for l in range(NL)]
        start = ns
        fd = dirty.index(True) if True in dirty else NL
        ex = [l for l in range(fd) if eob[l]]
        if ex: return passes, iters, log, end[ex[0]], True
        if fd == NL: return passes, iters, log, end[NL - 1], False

if __name__ == "__main__":
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    raw = datagen.chunk(kind, 65536, 0x0E110004)
    co = zlib.compressobj(6, zlib.DEFLATED, -15); data = co.compress(raw) + co.flush()
    print("compressed", len(data))
    for bits, start, lit, off, endpos in blocks(data):
        pos = start
        print("block", start, endpos)
        while pos < endpos and (pos >> 3) + 64 <= len(data):
            p, it, log, pos2, e = sim_round(bits, pos, lit, off)
            print(f"  round at {pos}: passes {p} iterations {it} eob {e} {log[:8]}{'...' if len(log) > 8 else ''}")
            pos = pos2
            if e: break
