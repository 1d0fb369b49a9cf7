"""Parse a raw DEFLATE stream and report its structure (blocks, literals,
matches, header bits) - used to compare our streams with the reference's."""
import sys

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258,258,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0,0,0]
OFF_BASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577,24577,24577]
OFF_EXTRA = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13,13,13]
PERM = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]


class Bits:
    def __init__(s, data):
        s.d = data; s.pos = 0
    def get(s, n):
        v = 0
        for i in range(n):
            v |= ((s.d[s.pos >> 3] >> (s.pos & 7)) & 1) << i
            s.pos += 1
        return v


def mkdec(lens):
    codes = {}
    code = 0
    for l in range(1, 16):
        for sym, sl in enumerate(lens):
            if sl == l:
                codes[(l, code)] = sym
                code += 1
        code <<= 1
    def dec(b):
        c = 0
        for l in range(1, 16):
            c = (c << 1) | b.get(1)
            if (l, c) in codes:
                return codes[(l, c)]
        raise ValueError("bad code")
    return dec


def stats(data):
    b = Bits(data)
    out = bytearray()
    blocks = []
    while True:
        final = b.get(1); typ = b.get(2)
        st = {"type": typ, "start": len(out), "lits": 0, "matches": 0, "mlen": 0,
              "hdr_bits": 0, "len3": 0, "dist_hist": [0]*5}
        p0 = b.pos
        if typ == 0:
            b.pos = (b.pos + 7) & ~7
            ln = b.get(16); b.get(16)
            for _ in range(ln):
                out.append(b.get(8))
            st["lits"] = ln
        else:
            if typ == 2:
                nl = 257 + b.get(5); nd = 1 + b.get(5); nc = 4 + b.get(4)
                pl = [0]*19
                for i in range(nc):
                    pl[PERM[i]] = b.get(3)
                pd = mkdec(pl)
                lens = []
                while len(lens) < nl + nd:
                    s = pd(b)
                    if s < 16: lens.append(s)
                    elif s == 16: lens += [lens[-1]] * (3 + b.get(2))
                    elif s == 17: lens += [0] * (3 + b.get(3))
                    else: lens += [0] * (11 + b.get(7))
                ld, dd = mkdec(lens[:nl]), mkdec(lens[nl:nl+nd])
            else:
                ld = mkdec([8]*144 + [9]*112 + [7]*24 + [8]*8); dd = mkdec([5]*32)
            st["hdr_bits"] = b.pos - p0
            while True:
                s = ld(b)
                if s < 256:
                    out.append(s); st["lits"] += 1
                elif s == 256:
                    break
                else:
                    ln = LEN_BASE[s-257] + b.get(LEN_EXTRA[s-257])
                    ds = dd(b)
                    dist = OFF_BASE[ds] + b.get(OFF_EXTRA[ds])
                    for _ in range(ln):
                        out.append(out[-dist])
                    st["matches"] += 1; st["mlen"] += ln
                    st["len3"] += ln == 3
                    st["dist_hist"][0 if dist < 64 else 1 if dist < 1024 else 2 if dist < 8192 else 3 if dist < 24576 else 4] += 1
        st["bits"] = b.pos - p0 + 3
        st["len"] = len(out) - st["start"]
        blocks.append(st)
        if final:
            break
    return bytes(out), blocks


def summarize(name, data):
    out, blocks = stats(data)
    tl = sum(x["lits"] for x in blocks); tm = sum(x["matches"] for x in blocks)
    ml = sum(x["mlen"] for x in blocks); hb = sum(x["hdr_bits"] for x in blocks)
    dh = [sum(x["dist_hist"][i] for x in blocks) for i in range(5)]
    print(f"{name}: {len(data)} B, blocks {len(blocks)} types {[x['type'] for x in blocks]}, "
          f"lits {tl}, matches {tm} (len3 {sum(x['len3'] for x in blocks)}), avg mlen {ml/max(tm,1):.2f}, "
          f"hdr {hb//8} B, dist<64/1K/8K/24K/32K {dh}")
    return out
