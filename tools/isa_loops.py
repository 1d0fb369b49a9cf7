#!/usr/bin/env python3
"""isa_loops.py FILE.s KERNEL [--spills] - list the backward-branch loops of one
kernel in `hipcc -S [-gline-tables-only]` output with their instruction mix (a
static count: the body between the branch target and the branch), the SGPR
spill traffic inside (v_readlane / v_writelane that the source did not ask
for issue on the VALU) and, when the file carries .loc directives, the source
lines the body comes from.  A tuning aid, not part of the product."""
import re
import sys
from collections import Counter


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels, insts, locs = {}, [], []
    cur = 0
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.strip()
        m = re.match(r"\.loc\s+\d+\s+(\d+)", t)
        if m:
            cur = int(m.group(1))
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t)
        locs.append(cur)
    loops = []
    for i, t in enumerate(insts):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            lab = m.group(1) or m.group(2)
            j = labels.get(lab)
            if j is not None and j <= i:
                loops.append((j, i, lab))
    tot_r = sum(1 for t in insts if t.startswith("v_readlane"))
    tot_w = sum(1 for t in insts if t.startswith("v_writelane"))
    print(f"{kern}: {len(insts)} instructions, {len(loops)} loops, "
          f"{tot_r} v_readlane, {tot_w} v_writelane, "
          f"{sum(1 for t in insts if t.startswith('scratch_'))} scratch ops")
    for j, i, lab in sorted(loops):
        seg = insts[j:i + 1]
        cat = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "dpp": 0,
               "rdl": 0, "wrl": 0, "scr": 0}
        for t in seg:
            op = t.split()[0]
            if op.startswith("v_"):
                cat["valu"] += 1
                if "dpp" in t:
                    cat["dpp"] += 1
                if op.startswith("v_readlane"):
                    cat["rdl"] += 1
                if op.startswith("v_writelane"):
                    cat["wrl"] += 1
            elif op.startswith("ds_"):
                cat["lds"] += 1
            elif op.startswith("scratch_"):
                cat["scr"] += 1
                cat["vmem"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                cat["vmem"] += 1
            elif op.startswith("s_waitcnt"):
                cat["wait"] += 1
            elif op.startswith("s_"):
                cat["salu"] += 1
        lc = Counter(x for x in locs[j:i + 1] if x)
        src = ""
        if lc:
            lo, hi = min(lc), max(lc)
            top = ",".join(str(k) for k, _ in lc.most_common(3))
            src = f" src {lo}-{hi} (most: {top})"
        print(f"  {lab:12s} [{j:5d}..{i:5d}] n={len(seg):5d} " +
              " ".join(f"{k}={v}" for k, v in cat.items()) + src)


if __name__ == "__main__":
    main()
