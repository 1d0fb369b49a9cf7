#!/usr/bin/env python3
"""isa_loops.py FILE.s KERNEL - list the backward-branch loops of one kernel in
hipcc -S output with their instruction mix (a static count: the body between
the branch target and the branch).  A tuning aid, not part of the product."""
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels, insts = {}, []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t)
    loops = []
    for i, t in enumerate(insts):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            lab = m.group(1) or m.group(2)
            j = labels.get(lab)
            if j is not None and j <= i:
                loops.append((j, i, lab))
    print(f"{kern}: {len(insts)} instructions, {len(loops)} loops")
    for j, i, lab in sorted(loops):
        seg = insts[j:i + 1]
        cat = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "dpp": 0}
        for t in seg:
            op = t.split()[0]
            if op.startswith("v_"):
                cat["valu"] += 1
                if "dpp" in t:
                    cat["dpp"] += 1
            elif op.startswith("ds_"):
                cat["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cat["vmem"] += 1
            elif op.startswith("s_waitcnt"):
                cat["wait"] += 1
            elif op.startswith("s_"):
                cat["salu"] += 1
        print(f"  {lab:12s} [{j:5d}..{i:5d}] n={len(seg):5d} " +
              " ".join(f"{k}={v}" for k, v in cat.items()))


if __name__ == "__main__":
    main()
